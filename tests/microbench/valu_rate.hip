// valu_rate.hip — issue interval of the instructions the Cholesky pivot wave is made of, for ONE wave on a SIMD and for several.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, int iters, unsigned long long* cyc) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
    double b = 1.0000001, c = 1e-9;
    int ia[8]; for (int i = 0; i < 8; i++) ia[i] = threadIdx.x + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_fma(a[i], b, c);                                    // v_fma_f64, 8 independent chains
            if (OP == 1) a[i] = a[i] * b;                                                      // v_mul_f64
            if (OP == 2) a[i] = a[i] + c;                                                      // v_add_f64
            if (OP == 3) ia[i] = __builtin_amdgcn_mov_dpp(ia[i], 0x150 + 3, 0xf, 0xf, false);  // v_mov_b32_dpp row_newbcast
            if (OP == 4) a[i] = __builtin_amdgcn_update_dpp(0.0, a[i], 0x150 + 3, 0xf, 0xf, false);   // v_mov_b64_dpp
            if (OP == 5) { float f = (float)a[i]; f = __builtin_fmaf(f, 1.0001f, 1e-6f); a[i] = f; }
            if (OP == 6) ia[i] = __builtin_amdgcn_readlane(ia[i], 5) + ia[i];                   // v_readlane + v_add
            if (OP == 7) a[i] = __builtin_fma(a[(i + 1) & 7], b, a[i]);                          // fma with all-VGPR operands
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i] + ia[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(int threads, const char* label) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    int iters = 20000;
    k<OP><<<1, threads>>>(out, 100, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<1, threads>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int wps = threads / 256; if (wps < 1) wps = 1;
    printf("%-34s %4d threads (%d wave/SIMD): %6.2f ticks per instruction per wave, %6.2f ns per instruction per SIMD\n", label, threads, wps,
           (double)c / (8.0 * iters), ms * 1e6 / (8.0 * iters) / wps);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>(64, "v_fma_f64 (imm/sgpr operands)"); run<0>(256, "v_fma_f64"); run<0>(512, "v_fma_f64"); run<0>(1024, "v_fma_f64");
    run<7>(64, "v_fma_f64 (three VGPR operands)"); run<7>(1024, "v_fma_f64 (three VGPR operands)");
    run<1>(64, "v_mul_f64"); run<1>(1024, "v_mul_f64");
    run<2>(64, "v_add_f64"); run<2>(1024, "v_add_f64");
    run<3>(64, "v_mov_b32_dpp"); run<3>(1024, "v_mov_b32_dpp");
    run<4>(64, "v_mov_b64_dpp"); run<4>(1024, "v_mov_b64_dpp");
    run<5>(64, "cvt + v_fma_f32 + cvt"); 
    run<6>(64, "v_readlane + v_add_u32"); run<6>(1024, "v_readlane + v_add_u32");
    return 0;
}
