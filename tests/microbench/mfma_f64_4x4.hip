// mfma_f64_4x4.hip — v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64: issue interval, dependent latency, one wave and four per SIMD.
// A 16x16x4 rank-4 tile update is four 4x4x4_4b instructions (A block abid broadcast to the four blocks), same operand registers.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_4x4 mfma_f64_4x4.hip && ./mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(1024) k(double* out, int iters, unsigned long long* cyc) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    double4_t acc[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // 8 independent 4x4x4 chains
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[i], 0, 0, 0);
        } else if (MODE == 1) {     // one dependent 4x4x4 chain (8 per iteration)
#pragma unroll
            for (int i = 0; i < 8; i++) d[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[0], 0, 0, 0);
        } else if (MODE == 2) {     // a 16x16x16 tile product as 16 instructions: four strips (abid 0..3, cbsz 2), four k-slices each
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                d[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[0], 2, 0, 0);
                d[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[1], 2, 1, 0);
                d[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[2], 2, 2, 0);
                d[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[3], 2, 3, 0);
            }
        } else if (MODE == 3) {     // the same product as four dependent 16x16x4
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 4) {     // two tile products interleaved, 16x16x4
#pragma unroll
            for (int kk = 0; kk < 4; kk++) { acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[1], 0, 0, 0); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += d[i];
    s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(int threads, int per_iter, double fma_per_instr, const char* label) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    int iters = 20000;
    k<MODE><<<1, threads>>>(out, 100, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<1, threads>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int wps = threads / 256; if (wps < 1) wps = 1;
    double ns_instr_simd = ms * 1e6 / ((double)per_iter * iters) / wps;
    printf("%-44s %4d thr (%d wave/SIMD): %7.1f ticks/instr/wave  %6.2f ns/instr/SIMD  %6.2f FMA/ns/SIMD\n", label, threads, wps,
           (double)c / ((double)per_iter * iters), ns_instr_simd, fma_per_instr / ns_instr_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>(64, 8, 256, "4x4x4_4b, 8 independent chains");
    run<0>(1024, 8, 256, "4x4x4_4b, 8 independent chains");
    run<1>(64, 8, 256, "4x4x4_4b, dependent chain");
    run<2>(64, 16, 256, "16x16x16 product = 16 x 4x4x4_4b (bcast A)");
    run<2>(1024, 16, 256, "16x16x16 product = 16 x 4x4x4_4b (bcast A)");
    run<3>(64, 4, 1024, "16x16x16 product = 4 dependent 16x16x4");
    run<3>(1024, 4, 1024, "16x16x16 product = 4 dependent 16x16x4");
    run<4>(64, 8, 1024, "two products interleaved, 16x16x4");
    run<4>(1024, 8, 1024, "two products interleaved, 16x16x4");
    return 0;
}
