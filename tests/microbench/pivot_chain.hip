// pivot_chain.hip — what bounds one column of the 16x16 diagonal-tile factorisation (rr3_pivot_factor, swf_chol_rr.h)?
// One wavefront, the tile in LDS, REP tiles back to back; variants switch parts of the column off.  Prints core-clock cycles per column.
//   hipcc --offload-arch=gfx950 -O3 -o pivot_chain pivot_chain.hip && ./pivot_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_d(double v, int idx_bytes) {
    int lo = __builtin_amdgcn_ds_bpermute(idx_bytes, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(idx_bytes, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int N> __device__ __forceinline__ double bc(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + N, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// VAR bit 0: no LDS publishes; bit 1: one Newton step only; bit 2: no register update (chain only); bit 3: rsq replaced by a multiply (chain without the transcendental)
template <int VAR>
__global__ void __launch_bounds__(64) k_pivot(const double* Din, double* out, unsigned long long* cyc, int rep) {
#pragma clang fp contract(off)
    __shared__ double D[16][17];
    __shared__ double colb[256];
    __shared__ double ipb[16];
    __shared__ unsigned prog;
    int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    for (int e = lane; e < 256; e += 64) D[e >> 4][e & 15] = Din[e];
    __syncthreads();
    int bidx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bidx[r] = (r * 16 + li) * 4;
    const int pidx = (li & 3) * 4 + (li >> 2);
    double keep = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rep; it++) {
        double A_[4];
#pragma unroll
        for (int q = 0; q < 4; q++) A_[q] = D[lk + 4 * q][li];
        double rowA = bperm_d(A_[0], bidx[0]);
        double dp = readlane_d(A_[0], 0);
        double rowPre = bperm_d(A_[0], bidx[1]);
#pragma unroll
        for (int c = 0; c < 16; c++) {
            double y = (VAR & 8) ? dp * 0.01 : __builtin_amdgcn_rsq(dp), h = 0.5 * dp;
            y = y * __builtin_fma(-(h * y), y, 1.5);
            double f = 1.0, ip = y;
            if (!(VAR & 2)) { f = __builtin_fma(-(h * y), y, 1.5); ip = y * f; }
            const double row0 = (li > c) ? rowA : 0.0;
            const double sA = (VAR & 2) ? (row0 * y) * ip : ((row0 * y) * f) * ip;
            if (!(VAR & 1)) {
                colb[c * 16 + pidx] = rowA; ipb[c] = ip;
                asm volatile("" ::: "memory");
                __hip_atomic_store(&prog, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            }
            if (c == 15) { keep += ip; break; }
            const int c1 = c + 1;
            const double x = readlane_d(A_[c1 >> 2], (c1 & 3) * 16 + c);
            const double rowNext = __builtin_fma(-x, sA, rowPre);
            const double dpNext = readlane_d(rowNext, c1);
            if (!(VAR & 4)) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (4 * q + 3 <= c) continue;
                    double col;
                    switch (c) {
#define C(K) case K: col = bc<K>(A_[q]); break;
                        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) default: col = 0;
#undef C
                    }
                    A_[q] = __builtin_fma(-col, sA, A_[q]);
                }
            }
            if (c + 2 < 16) rowPre = bperm_d(A_[(c + 2) >> 2], bidx[(c + 2) & 3]);
            rowA = rowNext; dp = dpNext;
        }
        keep += A_[0] + A_[1] + A_[2] + A_[3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = keep;
    if (lane == 0) *cyc = t1 - t0;
}
template <int VAR> static void run(const double* dD, double* dO, unsigned long long* dC, const char* what) {
    const int rep = 2000;
    hipLaunchKernelGGL(k_pivot<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dC, 10);
    hipLaunchKernelGGL(k_pivot<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dC, rep);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx950: convert with the measured launch length instead of trusting a constant
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_pivot<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dC, rep); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f memtime ticks / column   %7.1f ns / column (events)\n", what, (double)c / rep / 16, ms * 1e6 / rep / 16);
}
int main() {
    std::vector<double> D(256);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) D[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *dD, *dO; unsigned long long* dC;
    hipMalloc(&dD, 2048); hipMalloc(&dO, 512); hipMalloc(&dC, 8);
    hipMemcpy(dD, D.data(), 2048, hipMemcpyHostToDevice);
    run<0>(dD, dO, dC, "full column (as rr3_pivot_factor)");
    run<1>(dD, dO, dC, "no LDS publishes");
    run<2>(dD, dO, dC, "one Newton step");
    run<4>(dD, dO, dC, "chain only (no register update)");
    run<5>(dD, dO, dC, "chain only, no publishes");
    run<7>(dD, dO, dC, "chain only, no publishes, one Newton step");
    run<13>(dD, dO, dC, "chain only, no publishes, multiply instead of v_rsq");
    return 0;
}
