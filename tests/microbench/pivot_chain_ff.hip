// pivot_chain_ff.hip — the diagonal-tile factorisation WITHOUT a reciprocal square root on its dependency chain.
// (round 5; compare with pivot_chain.hip = rr3_pivot_factor of rounds 3-4: 211 cycles per column, of which the v_rsq_f64 +
// two Newton steps + three scalings are ~150.)
//
// Scaled fraction-free elimination: the tile is carried as M^(c) = s_c A^(c) (A^(c) = the Schur complement after c columns, s_c > 0):
//     M^(c+1) = (dp M^(c) - col row^T) 2^-e        dp = M^(c)_cc = m 2^e, m in [1, 2)
// so s_(c+1) = s_c m stays within [1, 2^16) over a tile, nothing is divided and the chain of a column is
//     v_readlane (next pivot) -> two SALU bit operations (2^-e, m) -> v_mul (row 2^-e) -> v_mul (x row) -> v_fma.
// The Cholesky column is recovered off the chain: L[r][c] = M^(c)[r][c] rsqrt(s_c dp_c), the rsqrt of column c running in the
// shadow of column c+1.  Prints core-clock ticks and ns per column and the residual of L L^T = A.
//   hipcc --offload-arch=gfx950 -O3 -o pivot_chain_ff pivot_chain_ff.hip && ./pivot_chain_ff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
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
__device__ __forceinline__ double bcn(double v, int c) {
    switch (c) {
#define C(K) case K: return bc<K>(v);
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) default: return 0;
#undef C
    }
}
__device__ __forceinline__ double rsqrt_nr(double x) {
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rsq(x), h = 0.5 * x;
    y = y * __builtin_fma(-(h * y), y, 1.5);
    y = y * __builtin_fma(-(h * y), y, 1.5);
    return y;
}
// VAR bit 0: no LDS publishes; bit 2: chain only (no register update); bit 4: no rho (the rsqrt in the shadow) at all
template <int VAR>
__global__ void __launch_bounds__(64) k_pivot_ff(const double* Din, double* out, double* Lout, unsigned long long* cyc, int rep) {
#pragma clang fp contract(off)
    __shared__ double D[16][17];
    __shared__ double colb[256];
    __shared__ double dsb[16], sgb[16], ipb[16];
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
        double s = 1.0, tprev = 1.0, tl = 1.0;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            // dp = m 2^e: 2^-e and m by integer operations on the (wave-uniform) high word
            const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(dp)), lo = __builtin_amdgcn_readfirstlane(__double2loint(dp));
            const double sg = __hiloint2double(0x7fe00000 - (hi & 0x7ff00000), 0);
            const double dpS = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
            const double rowS = (VAR & 32) ? rowA * sg : ((li > c) ? rowA * sg : 0.0);
            const double t = s * dp;                       // s_c dp_c: L[:, c] = M[:, c] rsqrt(t)
            s = s * dpS;
            if (!(VAR & 1)) {
                if (VAR & 32) {
                    if (lane < 16) colb[c * 16 + pidx] = rowA;
                    asm volatile("" ::: "memory");
                    if (lane == 0) __hip_atomic_store((unsigned long long*)(sgb + c), (unsigned long long)__double_as_longlong(sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                } else {
                colb[c * 16 + pidx] = rowA; dsb[c] = dpS;
                asm volatile("" ::: "memory");
                __hip_atomic_store((unsigned long long*)(sgb + c), (unsigned long long)__double_as_longlong(sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                }
            }
            if (!(VAR & 16) && !(VAR & 32) && c > 0) {                    // the rsqrt of the previous column, in the shadow of this one
                const double rho = rsqrt_nr(tprev);
                if (!(VAR & 1)) ipb[c - 1] = rho; else keep += rho;
            }
            if (VAR & 32) tl = (li == c) ? t : tl;
            tprev = t;
            if (c == 15) { keep += dpS; break; }
            const int c1 = c + 1;
            const double x = readlane_d(A_[c1 >> 2], (c1 & 3) * 16 + c);
            const double rowNext = __builtin_fma(dpS, rowPre, -(x * rowS));
            const double dpNext = readlane_d(rowNext, c1);
            if (!(VAR & 4)) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (4 * q + 3 <= c) continue;
                    const double col = bcn(A_[q], c);
                    A_[q] = __builtin_fma(dpS, A_[q], -(col * rowS));
                }
            }
            if (c + 2 < 16) rowPre = bperm_d(A_[(c + 2) >> 2], bidx[(c + 2) & 3]);
            rowA = rowNext; dp = dpNext;
        }
        if (VAR & 32) { const double rho = rsqrt_nr(tl); if (lane < 16) ipb[li] = rho; }
        else if (!(VAR & 16)) { const double rho = rsqrt_nr(tprev); if (!(VAR & 1)) ipb[15] = rho; else keep += rho; }
        keep += A_[0] + A_[1] + A_[2] + A_[3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = keep;
    if (lane == 0) *cyc = t1 - t0;
    if (VAR == 0 || VAR == 32) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) { int r = lk + 4 * q; Lout[r * 16 + li] = (li <= r) ? colb[li * 16 + (r & 3) * 4 + (r >> 2)] * ipb[li] : 0.0; }
    }
}
// rows in registers: R[r] = row r as a vector over li (the four 16-lane rows carry the same values); the multiplier of row r is lane c of
// R[r] itself (64-bit DPP row_newbcast), so a column needs no LDS crossbar and no cross-row traffic at all.
// VAR bit 0: fused v_fmac_f64_dpp through inline asm (s_nop for the VALU -> DPP hazard inside the asm); else v_mov_b64_dpp by the compiler
template <int VAR, int C> struct Col {
    static __device__ __forceinline__ void go(double (&R)[16], double* colb, double* sgb, double& s, double& tl, int lane, int li) {
#pragma clang fp contract(off)
        const double dp = readlane_d(R[C], C);
        const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(dp)), lo = __builtin_amdgcn_readfirstlane(__double2loint(dp));
        const double sg = __hiloint2double(0x7fe00000 - (hi & 0x7ff00000), 0);
        const double dpS = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
        if (lane < 16) colb[C * 16 + li] = R[C];
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store((unsigned long long*)(sgb + C), (unsigned long long)__double_as_longlong(sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        const double t = s * dp;
        s = s * dpS;
        tl = (li == C) ? t : tl;
        const double nrs = R[C] * -sg;
#pragma unroll
        for (int r = C + 1; r < 16; r++) {
            if (VAR & 1) {
                double T = dpS * R[r];
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(T) : "v"(R[r]), "v"(nrs), "n"(C));
                R[r] = T;
            } else {
                const double m = __builtin_amdgcn_update_dpp(0.0, R[r], 0x150 + C, 0xf, 0xf, false);
                R[r] = __builtin_fma(dpS, R[r], m * nrs);
            }
        }
        if constexpr (C < 15) Col<VAR, C + 1>::go(R, colb, sgb, s, tl, lane, li);
    }
};
template <int VAR>
__global__ void __launch_bounds__(64) k_pivot_rr(const double* Din, double* out, double* Lout, unsigned long long* cyc, int rep) {
#pragma clang fp contract(off)
    __shared__ double D[16][17];
    __shared__ double colb[256];
    __shared__ double sgb[16], ipb[16];
    int lane = threadIdx.x, li = lane & 15;
    for (int e = lane; e < 256; e += 64) D[e >> 4][e & 15] = Din[e];
    __syncthreads();
    double keep = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rep; it++) {
        double R[16];
#pragma unroll
        for (int r = 0; r < 16; r++) R[r] = D[r][li];
        double s = 1.0, tl = 1.0;
        Col<VAR, 0>::go(R, colb, sgb, s, tl, lane, li);
        const double rho = rsqrt_nr(tl);
        if (lane < 16) ipb[li] = rho;
        keep += R[15] + rho;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = keep;
    if (lane == 0) *cyc = t1 - t0;
    __syncthreads();
    for (int e = lane; e < 256; e += 64) { int r = e >> 4, c = e & 15; Lout[e] = (c <= r) ? colb[c * 16 + r] * ipb[c] : 0.0; }
}
template <int VAR> static void run_rr(const double* dD, double* dO, double* dL, unsigned long long* dC, const char* what) {
    const int rep = 2000;
    hipLaunchKernelGGL(k_pivot_rr<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, 10);
    hipLaunchKernelGGL(k_pivot_rr<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, rep);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_pivot_rr<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, rep); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f memtime ticks / column   %7.1f ns / column (events)\n", what, (double)c / rep / 16, ms * 1e6 / rep / 16);
}
static double resid(const std::vector<double>& D, const std::vector<double>& L) {
    double worst = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) {
        long double a = 0; for (int k = 0; k <= j; k++) a += (long double)L[i * 16 + k] * L[j * 16 + k];
        double rel = fabs((double)(a - D[i * 16 + j])) / sqrt(D[i * 16 + i] * D[j * 16 + j]);
        if (rel > worst) worst = rel;
    }
    return worst;
}
template <int VAR> static void run(const double* dD, double* dO, double* dL, unsigned long long* dC, const char* what) {
    const int rep = 2000;
    hipLaunchKernelGGL(k_pivot_ff<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, 10);
    hipLaunchKernelGGL(k_pivot_ff<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, rep);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_pivot_ff<VAR>, dim3(1), dim3(64), 0, 0, dD, dO, dL, dC, rep); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f memtime ticks / column   %7.1f ns / column (events)\n", what, (double)c / rep / 16, ms * 1e6 / rep / 16);
}
int main() {
    std::vector<double> D(256), L(256);
    // a badly scaled SPD tile: diagonal entries from 1e8 down to 1e-2
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double si = pow(10.0, 4.0 - i / 3.0), sj = pow(10.0, 4.0 - j / 3.0);
        D[i * 16 + j] = si * sj * ((i == j ? 1.2 : 0.0) + 1.0 / (1 + i + j));
    }
    double *dD, *dO, *dL; unsigned long long* dC;
    hipMalloc(&dD, 2048); hipMalloc(&dO, 512); hipMalloc(&dL, 2048); hipMalloc(&dC, 8);
    hipMemcpy(dD, D.data(), 2048, hipMemcpyHostToDevice);
    run<0>(dD, dO, dL, dC, "fraction-free column, full");
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) {
        long double a = 0; for (int k = 0; k <= j; k++) a += (long double)L[i * 16 + k] * L[j * 16 + k];
        double rel = fabs((double)(a - D[i * 16 + j])) / sqrt(D[i * 16 + i] * D[j * 16 + j]);
        if (rel > worst) worst = rel;
    }
    printf("max |L L^T - A|_ij / sqrt(A_ii A_jj) = %.3e\n", worst);
    run<32>(dD, dO, dL, dC, "fraction-free, single-lane publishes, no masks, rho at end");
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
    printf("   residual %.3e\n", resid(D, L));
    run_rr<0>(dD, dO, dL, dC, "rows in registers, v_mov_b64_dpp");
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
    printf("   residual %.3e\n", resid(D, L));
    run_rr<1>(dD, dO, dL, dC, "rows in registers, fused v_fmac_f64_dpp");
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
    printf("   residual %.3e\n", resid(D, L));
    run<1>(dD, dO, dL, dC, "no LDS publishes");
    run<16>(dD, dO, dL, dC, "no rsqrt in the shadow");
    run<4>(dD, dO, dL, dC, "chain only (no register update)");
    run<5>(dD, dO, dL, dC, "chain only, no publishes");
    run<21>(dD, dO, dL, dC, "chain only, no publishes, no rsqrt");
    return 0;
}
