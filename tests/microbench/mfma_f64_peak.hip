// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (the guide lists no fp64 MFMA peak).
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, unsigned long long* cyc) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = double4_t{ 0, 0, 0, 0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC> void run(int blocks, const char* label) {
    double* out; unsigned long long* cyc; hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, 8);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 100, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4 * blocks;
    printf("%-28s blocks %5d: %8.3f ms  %7.2f TFLOP/s   cycles(counter)/MFMA/wave %.1f\n", label, blocks, ms, flops / ms / 1e9, (double)c / ((double)NACC * iters));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(256, "1 acc (dependent chain)");
    run<2>(256, "2 acc");
    run<4>(256, "4 acc, 1 wave/SIMD");
    run<4>(512, "4 acc, 2 waves/SIMD");
    run<8>(1024, "8 acc, 4 waves/SIMD");
    // one CU alone (the latency path's regime: does the clock / the matrix pipe behave differently when 255 CUs idle?)
    run<1>(1, "ONE block: 1 acc (dependent)");
    run<4>(1, "ONE block: 4 acc, 1 wave/SIMD");
    run<4>(2, "TWO blocks: 4 acc");
    run<8>(4, "FOUR blocks: 8 acc");
    return 0;
}
