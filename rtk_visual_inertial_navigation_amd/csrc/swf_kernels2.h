// swf_kernels2.h — reduced solve, back-substitution and the on-device trust-region control.
//
// The dogleg loop restates the public Ceres 2.x TrustRegionMinimizer + DoglegStrategy
// (TRADITIONAL_DOGLEG) that the reference drives through ceres::Solve
// (R/swf/swf_image.cpp:198-251, options R/swf/swf.cpp:25-30; constants SURVEY.md App. C).
// Accept/reject, radius, mu and convergence are decided per window on the device; the host
// only enqueues a fixed sequence of launches and never reads anything back mid-solve.
#pragma once
#include "swf_kernels.h"

// =========================================================================================
// Dense reduced solve, one workgroup per window (256 threads when n_red < 256, else 1024).
// Blocked left-looking Cholesky S = L L^T in the predefined elimination order, 32-wide block
// columns, one matrix row per thread:
//   * the O(n^3) part — block column minus (already factored columns)^2 — runs on the fp64
//     matrix cores (v_mfma_f64_16x16x4_f64): each wave owns 64 rows x 32 columns = 4x2 tiles;
//     both operands come straight from the TRANSPOSED working factor (Lt[k][r] = L[r][k],
//     leading dim n+1), 16 contiguous doubles per k, so no LDS staging is needed;
//   * the tiles are transposed through LDS into row-per-lane registers for the 32x32 diagonal
//     factorisation (inside wave 0, column broadcast through LDS) and the triangular solve;
//   * rhs rides along as row n of the matrix, so the forward solve y = L^-1 rhs falls out of
//     the factorisation; only the backward solve L^T z = y is done separately.
// f64 MFMA layouts: A[i][k]: lane = i + 16k;  B[k][j]: lane = j + 16k;
//                   D: lane l, reg q -> D[row = (l>>4) + 4q][col = l&15].
// =========================================================================================
#define CH_NB 32
#ifndef CH_OCC
#define CH_OCC 2
#endif
#ifndef CH_KU
#define CH_KU 4
#endif
// broadcast of lane N of each 16-lane row to the whole row (DPP row_newbcast, gfx90a+)
__device__ __forceinline__ double row_newbcast_d(double v, int n) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (n) {
#define NB_CASE(K) case K: lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + K, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + K, 0xf, 0xf, false); break;
        NB_CASE(0) NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5) NB_CASE(6) NB_CASE(7)
        NB_CASE(8) NB_CASE(9) NB_CASE(10) NB_CASE(11) NB_CASE(12) NB_CASE(13) NB_CASE(14) NB_CASE(15)
#undef NB_CASE
    }
    return __hiloint2double(hi, lo);
}
// gather: every lane reads v from the lane whose index * 4 is idx_bytes
__device__ __forceinline__ double bperm_d(double v, int idx_bytes) {
    int lo = __builtin_amdgcn_ds_bpermute(idx_bytes, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(idx_bytes, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
#ifdef SWF_PROFILE_CHOL
#ifndef SWF_PROFILE_CHOL_STEP
#define SWF_PROFILE_CHOL_STEP 2
#endif
__device__ unsigned long long g_chol_stamps[64];
#define CHSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_chol_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define CHACC(i, t0) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_chol_stamps[i] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#define CHSTAMP2(i) do { if (blockIdx.x == 0 && threadIdx.x == 128) g_chol_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define CHACC2(i, t0) do { if (blockIdx.x == 0 && threadIdx.x == 128) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_chol_stamps[i] += t_ - (t0); (t0) = t_; } } while (0)
#else
#define CHSTAMP(i)
#define CHACC(i, t0)
#define CHSTAMP2(i)
#define CHACC2(i, t0)
#endif
// per-wave time stamps of ONE step of k_chol_rr4 (-DSWF_PROFILE_CHOLW): plain stores, no read-modify-write, the step chosen at run time
// (swf_debug_chol_wstep).  slot [wave][k]; wave 0: 0 pivot start, 1 pivot end, 2 past B_j, 3 past C_j; wave 1: 1 inverse end;
// tile waves: 0 past B_j, 1 panel done, 2 past C_j, 3 mask read, 4 diagonal terms done, 5 trailing done, 6 at B_j+1, 7 past B_j+1
#ifdef SWF_PROFILE_CHOLW
__device__ unsigned long long g_chol_pst[16];          // stamps inside the panel phase of the profiled step, by the wave that owns the next pivot tile
#ifdef SWF_PROFILE_CHOLC
#define PST(on, k) do { if ((on) && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_chol_pst[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PST(on, k)
#endif
__device__ unsigned long long g_chol_cst[2 * 16];      // per-column stamps of the pivot wave [0..15] and the inverse wave [16..31] in the profiled step
#ifdef SWF_PROFILE_CHOLC      // per-column and in-panel stamps cost ~100 cycles each: only on request
#define CST(on, k) do { if ((on) && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_chol_cst[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CST(on, k)
#endif
__device__ unsigned long long g_chol_wst[16 * 8];
__device__ int g_chol_wstep;
#define WST(step, k) do { if (blockIdx.x == 0 && (step) == g_chol_wstep && (threadIdx.x & 63) == 0) g_chol_wst[(threadIdx.x >> 6) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WST(step, k)
#define CST(on, k)
#define PST(on, k)
#endif
template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? CH_OCC : 4) k_chol_solve(DevBatch B) {
    __shared__ double Xs[2][64][CH_NB + 1];     // tile -> row layout, two waves at a time
    __shared__ double Dinv[CH_NB];              // 1 / L[c][c] of the current diagonal block
    __shared__ double D[CH_NB][CH_NB + 1];      // diagonal block (lower)
    __shared__ double zs[1024];
    __shared__ double part[CH_NB][NT / 32 + 1];
    __shared__ int fail;
    int w = blockIdx.x;
    WinState& s = B.ws[w];
    if (!s.need_lin || s.lin_fail) return;
    const WinRec& W = B.win[w];
    int n = W.n_red, ld = n + 1, tid = threadIdx.x;
    if (n <= 0) return;
    const double* S = B.S + W.S_base;
    double* Lt = B.L + W.Lt_base;
    const double* rhs = B.rhs + W.loc_base + W.n_e;
    int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    if (tid == 0) fail = 0;
#ifdef SWF_PROFILE_CHOL
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 64; i++) g_chol_stamps[i] = 0;
    unsigned long long tph = 0;
#endif
    __syncthreads();
    CHSTAMP(0);
    for (int J0 = 0; J0 < n; J0 += CH_NB) {
#ifdef SWF_PROFILE_CHOL
        tph = __builtin_amdgcn_s_memtime();
#endif
        int nb = (n - J0) < CH_NB ? (n - J0) : CH_NB;
        int r = J0 + tid;                       // my row (r == n : augmented rhs row)
        bool active = r <= n;
        int rbase = J0 + 64 * wv;
        bool wave_active = rbase <= n;
        double acc[CH_NB];
        if (wave_active) {
            double4_t T[4][2];
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        int rr = rbase + 16 * ti + lk + 4 * q, c = 16 * tj + li;
                        double v = 0;
                        if (rr <= n && c < nb) v = (rr < n) ? S[(size_t)rr * n + J0 + c] : rhs[J0 + c];
                        T[ti][tj][q] = v;
                    }
            bool vb0 = li < nb, vb1 = (16 + li) < nb;
            bool va[4];
#pragma unroll
            for (int ti = 0; ti < 4; ti++) va[ti] = (rbase + 16 * ti + li) <= n;
            // J0 is a multiple of 32: 4*KU columns of k per trip, all operand loads issued up front
            constexpr int KU = (NT == 256) ? CH_KU : 1;
            for (int k0 = 0; k0 < J0; k0 += 4 * KU) {
                double a[KU][4], b0[KU], b1[KU];
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    const double* Lk = Lt + (size_t)(k0 + 4 * u + lk) * ld;
                    b0[u] = vb0 ? Lk[J0 + li] : 0.0; b1[u] = vb1 ? Lk[J0 + 16 + li] : 0.0;
#pragma unroll
                    for (int ti = 0; ti < 4; ti++) a[u][ti] = va[ti] ? -Lk[rbase + 16 * ti + li] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < KU; u++)
#pragma unroll
                    for (int ti = 0; ti < 4; ti++) {
                        T[ti][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], b0[u], T[ti][0], 0, 0, 0);
                        T[ti][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], b1[u], T[ti][1], 0, 0, 0);
                    }
            }
#ifdef SWF_PROFILE_CHOL
            CHACC(8, tph); tph = __builtin_amdgcn_s_memtime();
#endif
            // tile layout -> one row per lane, four waves per round through LDS
#pragma unroll 1
            for (int rd = 0; rd < NT / 128; rd++) {
                if ((wv >> 1) == rd) {
#pragma unroll
                    for (int ti = 0; ti < 4; ti++)
#pragma unroll
                        for (int tj = 0; tj < 2; tj++)
#pragma unroll
                            for (int q = 0; q < 4; q++) Xs[wv & 1][16 * ti + lk + 4 * q][16 * tj + li] = T[ti][tj][q];
                }
                __syncthreads();
                if ((wv >> 1) == rd) {
#pragma unroll
                    for (int c = 0; c < CH_NB; c++) acc[c] = Xs[wv & 1][lane][c];
                }
                __syncthreads();
            }
        } else {
#pragma unroll 1
            for (int rd = 0; rd < NT / 128; rd++) { __syncthreads(); __syncthreads(); }
#pragma unroll
            for (int c = 0; c < CH_NB; c++) acc[c] = 0.0;
        }
        // factor the diagonal block inside wave 0 (lanes 0..nb-1 hold rows J0..J0+nb-1);
        // each finished column is broadcast to the other lanes through LDS
#ifdef SWF_PROFILE_CHOL
        CHACC(9, tph); tph = __builtin_amdgcn_s_memtime();
#endif
        if (tid < 64) {
            // 32x32 diagonal block: right-looking Cholesky in registers inside wave 0 (lane = row).
            // Column entries are broadcast with v_readlane (VALU -> SGPR), so the 32-step
            // dependency chain never waits on an LDS round trip.
            bool bad = false;
            bool inblk = lane < nb;
#pragma unroll
            for (int c = 0; c < CH_NB; c++) {
                double dpiv = readlane_d(acc[c], c);
                if (c < nb) {
                    if (!(dpiv > 0.0)) bad = true;
                    // one reciprocal square root per column instead of sqrt + a division per row
                    double ipiv = 1.0 / sqrt(dpiv);
                    if (inblk) acc[c] = (lane == c) ? dpiv * ipiv : (lane > c ? acc[c] * ipiv : acc[c]);
                    if (lane == c) Dinv[c] = ipiv;
                }
#pragma unroll
                for (int c2 = c + 1; c2 < CH_NB; c2++) {
                    double l2 = readlane_d(acc[c], c2);       // L[c2][c]
                    if (c2 < nb && inblk && lane >= c2) acc[c2] -= acc[c] * l2;
                }
            }
            if (inblk) {
#pragma unroll
                for (int c = 0; c < CH_NB; c++) {
                    D[lane][c] = (c <= lane) ? acc[c] : 0.0;
                    if (c <= lane && c < nb) Lt[(size_t)(J0 + c) * ld + J0 + lane] = acc[c];
                }
            }
            if (bad && lane == 0) fail = 1;
        }
        __syncthreads();
#ifdef SWF_PROFILE_CHOL
        CHACC(10, tph); tph = __builtin_amdgcn_s_memtime();
#endif
        if (fail) { if (tid == 0) { s.lin_fail = 1; s.chol_fail = 1; } return; }
        // triangular solve for the rows below the block: x L_d^T = acc
        if (active && r >= J0 + nb) {
#pragma unroll
            for (int c = 0; c < CH_NB; c++) {
                if (c < nb) {
                    double v = acc[c];
#pragma unroll
                    for (int k = 0; k < c; k++) v -= acc[k] * D[c][k];
                    v = v * Dinv[c];
                    acc[c] = v;
                    Lt[(size_t)(J0 + c) * ld + r] = v;
                }
            }
        }
        __syncthreads();
#ifdef SWF_PROFILE_CHOL
        CHACC(11, tph);
#endif
    }
    CHSTAMP(1);
    // backward solve L^T z = y, y = row n of L (Lt[k][n]); blocks from the bottom up
    for (int e = tid; e < n; e += NT) zs[e] = 0.0;
    __syncthreads();
    int nblk = (n + CH_NB - 1) / CH_NB;
    for (int jb = nblk - 1; jb >= 0; jb--) {
        int J0 = jb * CH_NB;
        int nb = (n - J0) < CH_NB ? (n - J0) : CH_NB;
        // partial dots over already solved z_r, r >= J0+nb
        {
            constexpr int PT = NT / 32;
            int kk = tid / PT, pt = tid % PT;
            double a = 0;
            if (kk < nb) for (int r = J0 + nb + pt; r < n; r += PT) a += Lt[(size_t)(J0 + kk) * ld + r] * zs[r];
            part[kk][pt] = a;
        }
        for (int e = tid; e < CH_NB * CH_NB; e += NT) {
            int c = e / CH_NB, k = e % CH_NB;      // D[c][k] = L[J0+c][J0+k], c >= k
            D[c][k] = (c < nb && k <= c) ? Lt[(size_t)(J0 + k) * ld + J0 + c] : 0.0;
        }
        if (tid < nb) Dinv[tid] = 1.0 / Lt[(size_t)(J0 + tid) * ld + J0 + tid];
        __syncthreads();
        if (tid < 64) {
            double b = 0;
            if (lane < nb) {
                double a = 0;
                for (int pt = 0; pt < NT / 32; pt++) a += part[lane][pt];
                b = Lt[(size_t)(J0 + lane) * ld + n] - a;
            }
#pragma unroll
            for (int c = CH_NB - 1; c >= 0; c--) {
                double bc = readlane_d(b, c);
                if (c < nb) {
                    double zc = bc * Dinv[c];
                    if (lane == c) b = zc;
                    else if (lane < c) b -= D[c][lane] * zc;
                }
            }
            if (lane < nb) zs[J0 + lane] = b;
        }
        __syncthreads();
    }
    double* y = B.y + W.loc_base + W.n_e;
    for (int e = tid; e < n; e += NT) y[e] = zs[e];
    CHSTAMP(2);
}

// Factor and invert one 16x16 SPD tile with the 64 lanes of one wavefront (shared by k_chol_col and k_chol_big; k_chol_rr2, its first user, left the tree in round 6).
// D: the full symmetric tile in LDS, overwritten by L (lower, zeros above); LiJ: receives L^-1 (lower).  Returns
// true if a pivot was not positive.
__device__ __forceinline__ bool chol_pivot_tile(double (*D)[17], double (*LiJ)[17], int li, int lk, double* ipb /* LDS, 16 doubles */) {
    // Factor and invert the 16x16 tile with all 64 lanes: the tile A and the running inverse R (starts as I)
    // live in the MFMA C-layout (lane (li, lk), reg q <-> row lk+4q, column li; A is kept fully symmetric).
    // Column c:  ip = 1/sqrt(A_cc);  column c of A reaches every lane of its row by one DPP row_newbcast,
    // row c of A / R reaches every row by one ds_bpermute;  then, for rows r > c,
    //   A[r][:] -= A[r][c] A[c][:] / A_cc      (right-looking Cholesky update)
    //   R[r][:] -= A[r][c] R[c][:] / A_cc      (forward substitution of L X = I, same broadcasts)
    // Column c of L = A[:][c] ip_c and row c of X = R[c][:] ip_c are never read again inside the loop, so their scalings
    // are deferred to one pass at the end (ip_c parked in LDS): the wave is bound by its own instruction issue
    // (~50 instructions per column, measured ~8 cycles each), and the per-column scaling with its exec-mask
    // bookkeeping was a quarter of them.  Same multiplications, same operands: bit-identical to scaling in place.
    double A_[4], R_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { A_[q] = D[lk + 4 * q][li]; R_[q] = (lk + 4 * q == li) ? 1.0 : 0.0; }
    bool bad = false;
    int bidx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bidx[r] = (r * 16 + li) * 4;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const int cq = c >> 2, cr = c & 3;
        double dp = readlane_d(A_[cq], cr * 16 + c);
        if (!(dp > 0.0)) bad = true;
        double ip = rsqrt_nr(dp);
        double ip2 = ip * ip;
        ipb[c] = ip;                                      // every lane writes the same value
        double rowA = bperm_d(A_[cq], bidx[cr]);          // A[c][li]
        double rowR = bperm_d(R_[cq], bidx[cr]);          // R[c][li]
        double sA = (li > c) ? rowA * ip2 : 0.0;          // columns <= c of A are final (L) already
        double sR = rowR * ip2;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (4 * q + 3 <= c) continue;                 // every row of this register is final already
            double col = row_newbcast_d(A_[q], c);        // A[lk+4q][c]
            if (lk + 4 * q > c) {
                A_[q] = __builtin_fma(-col, sA, A_[q]);
                R_[q] = __builtin_fma(-col, sR, R_[q]);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const double ipc = ipb[li];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int r = lk + 4 * q;
        D[r][li] = (li <= r) ? A_[q] * ipc : 0.0;         // column li of L
        LiJ[r][li] = (li <= r) ? R_[q] * ipb[r] : 0.0;    // row r of X = L^-1
    }
    return bad;
}

#include "swf_chol_rr4.h"

// =========================================================================================
// k_chol_big — the pivot / panel / look-ahead pipeline of round 1's k_chol_rr2 for 256 < n_red <= 640, where the factor
// (n^2/2 doubles, up to 1 MB) no longer fits the register file.  The tiles live in HBM/L2 in the window's L buffer
// (row-major, ld = n, the reduced rhs as row n) and are streamed through registers step by step:
//   wave 0        pivot wave: chol_pivot_tile on the published diagonal tile, Linv_jj kept in LDS for the panel and
//                 stored to the window's Linv slab for the backward pass
//   waves 1..15   own the tiles, owner(I, J) = 1 + (I + J) mod 15.  Ownership is static down to the lane (lane
//                 (li, lk), reg q <-> row lk+4q, column li), so every global read-after-write is same-thread.
// Step j:  B_j  [panel column j: tile -> LDS -> X = tile Linv^T on MFMA -> LDS (operands) + HBM (final L)]  C_j
//          [look-ahead: tile (j+1, j+1) updated first and published]  A_{j+1}  [remaining trailing tiles: load,
//          4 MFMAs with the panel operands from LDS, store; the next tile's loads are issued before the MFMAs].
// Step 0 reads S (lower; diagonal tiles mirrored), later steps read the L buffer.  Right-looking backward pass.
// =========================================================================================
#define CB_NMAX 640                     // largest reduced system of the tiled kernels (hs_row reaches ~620 in a live window)
#define CB_MAXT (CB_NMAX / 16 + 1)      // tile rows: 40 of the matrix + the rhs row
#define CC_NMAX 512                     // k_chol_col keeps two panels in LDS: up to 512 dimensions
#define CC_MAXT (CC_NMAX / 16 + 1)
template <bool BACK_ONLY>
__global__ void __launch_bounds__(1024) k_chol_big(DevBatch B) {
    __shared__ double Pn[CB_MAXT][16][17];     // panel of the current column, tile row I -> L_Ij (89 KB)
    __shared__ double Lic[16][17];             // Linv_jj of the current column
    __shared__ double ipiv[16];                // 1/sqrt(pivot) of the tile being factored (chol_pivot_tile scratch)
    __shared__ double Dt[2][16][17];           // published diagonal tiles, double-buffered
    __shared__ double zs[CB_NMAX + 16];
    __shared__ double yv[CB_NMAX + 16];
    __shared__ int fail;
    int w = blockIdx.x;
    WinState& st = B.ws[w];
    if (!st.need_lin || st.lin_fail) return;
    const WinRec& W = B.win[w];
    int n = W.n_red, tid = threadIdx.x;
    if (n <= B.rr_nmax) return;                    // the register-resident kernel's windows: the kernel is chosen per WINDOW, so a
                                                   // window's arithmetic does not depend on what else is in the batch
    int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int Tc = (n + 15) >> 4, Tr = Tc + 1;
    const double* S = B.S + W.S_base;
    double* Lw = B.L + W.Lt_base;
    double* LinvG = B.Linv + (size_t)w * (CB_MAXT - 1) * 256;
    if (tid == 0) fail = 0;
    for (int e = tid; e < CB_NMAX + 16; e += 1024) yv[e] = 0.0;     // padded entries must be exact zeros (they meet identity rows of Linv)
    // tile I/O in the C-layout.  first = true: from S (lower, mirrored inside diagonal tiles, rhs = row n of S)
    // interior tiles (all 16 rows inside the matrix, not a mirrored first read): scalar tile base + per-lane offset
    const int lane_off = lk * n + li;
    auto load_tile = [&](int I, int J, bool first) {
        double4_t v;
        const double* src = first ? S : Lw;
        if (16 * I + 16 <= n && !(first && I == J)) {
            const double* t0 = src + (size_t)(16 * I) * n + 16 * J + lane_off;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = t0[(size_t)(4 * q) * n];
            return v;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * I + lk + 4 * q, c = 16 * J + li;
            bool rhs_el = I == Tc && lk + 4 * q == 0;
            int rr = rhs_el ? n : r;
            bool inside = c < n && (rhs_el || (r < n && (c <= r || I == J)));
            int rc = rr < n ? rr : (rhs_el ? n : n - 1), cc = c < n ? c : n - 1;
            int off = (first && cc > rc) ? cc * n + rc : rc * n + cc;
            double x = src[off];
            v[q] = inside ? x : ((I < Tc && r == c) ? 1.0 : 0.0);
        }
        return v;
    };
    auto store_tile = [&](int I, int J, double4_t v, bool lower_only) {
        if (16 * I + 16 <= n && !lower_only) {
            double* t0 = Lw + (size_t)(16 * I) * n + 16 * J + lane_off;
#pragma unroll
            for (int q = 0; q < 4; q++) t0[(size_t)(4 * q) * n] = v[q];
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * I + lk + 4 * q, c = 16 * J + li;
            bool rhs_el = I == Tc && lk + 4 * q == 0;
            if (rhs_el) { if (c < n) Lw[(size_t)n * n + c] = v[q]; }
            else if (I < Tc && r < n && c < n) Lw[(size_t)r * n + c] = (lower_only && c > r) ? 0.0 : v[q];
        }
    };
    if (wv == 0) {
        // =============================== pivot wave ===============================
#ifdef SWF_PROFILE_CHOL
        if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 64; i++) g_chol_stamps[i] = 0;
        unsigned long long tq = 0;
#endif
        CHSTAMP(0);
        if (!BACK_ONLY) __syncthreads();                   // A_0: tile (0,0) published
        CHSTAMP(3);
        for (int j = 0; j < (BACK_ONLY ? 0 : Tc); j++) {
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            bool bad = chol_pivot_tile(Dt[j & 1], Lic, li, lk, ipiv);
#pragma unroll
            for (int q = 0; q < 4; q++) LinvG[(size_t)j * 256 + (lk + 4 * q) * 16 + li] = Lic[lk + 4 * q][li];
            if (bad && lane == 0) fail = 1;
            CHACC(9, tq);
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                               // B_j
            CHACC(8, tq);
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            if (fail) { if (tid == 0) { st.lin_fail = 1; st.chol_fail = 1; } return; }
            __syncthreads();                               // C_j
            CHACC(10, tq);
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                               // A_{j+1}
            CHACC(11, tq);
        }
        CHSTAMP(1);
        __syncthreads();                                   // E: yv ready
        CHSTAMP(4);
        // backward solve y = L^-T z, right-looking; Linv_JJ comes back from the slab
        for (int J = Tc - 1; J >= 0; J--) {
            double p = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) p += LinvG[(size_t)J * 256 + (lk + 4 * q) * 16 + li] * yv[16 * J + lk + 4 * q];
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (lk == 0) zs[16 * J + li] = p;
            __syncthreads();                               // X_J: y_J published
            __syncthreads();                               // Y_J: row J applied to the pending blocks
        }
        CHSTAMP(2);
        return;
    }
    // =============================== tile waves ===============================
    int kq = wv - 1;                                       // owns the tiles with (I + J) mod 15 == kq
    // first row I >= J of column J owned by this wave (then I + 15, I + 30)
    auto first_row = [&](int J) { int d = (kq - 2 * J) % 15; if (d < 0) d += 15; return J + d; };
    if (!BACK_ONLY && first_row(0) == 0) {                 // owner of (0,0): publish it
        double4_t v = load_tile(0, 0, true);
#pragma unroll
        for (int q = 0; q < 4; q++) Dt[0][lk + 4 * q][li] = v[q];
    }
    if (!BACK_ONLY) __syncthreads();                       // A_0
    for (int j = 0; j < (BACK_ONLY ? 0 : Tc); j++) {
        bool first = j == 0;
        __syncthreads();                                   // B_j: L_jj, Linv_jj ready
        if (fail) return;
        // ---- panel of column j: tiles (I, j), I > j, and the final L_jj
        for (int I = first_row(j); I < Tr; I += 15) {
            if (I == j) {
                double4_t v;
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = Dt[j & 1][lk + 4 * q][li];
                store_tile(j, j, v, true);
                continue;
            }
            double4_t v = load_tile(I, j, first);
#pragma unroll
            for (int q = 0; q < 4; q++) Pn[I][lk + 4 * q][li] = v[q];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            double4_t X = { 0, 0, 0, 0 };
#pragma unroll
            for (int kk = 0; kk < 4; kk++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Pn[I][li][lk + 4 * kk], Lic[li][lk + 4 * kk], X, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; q++) Pn[I][lk + 4 * q][li] = X[q];
            store_tile(I, j, X, false);
        }
        __syncthreads();                                   // C_j: panel published
        // ---- look-ahead: the next diagonal tile first, published for the pivot wave
        if (j + 1 < Tc && first_row(j + 1) == j + 1) {
            double4_t v = load_tile(j + 1, j + 1, first);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) v = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn[j + 1][li][lk + 4 * kk], Pn[j + 1][li][lk + 4 * kk], v, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) Dt[(j + 1) & 1][lk + 4 * q][li] = v[q];
            store_tile(j + 1, j + 1, v, false);
        }
        __syncthreads();                                   // A_{j+1}
        // ---- remaining trailing tiles (I, J), J > j, I >= J (the rhs row included), in groups of four: independent
        //      accumulators (a dependent fp64 MFMA chain costs 184 cycles per link) and the next group's loads in flight
        {
            int J = j + 1, I = J < Tc ? first_row(J) : Tr;
            auto settle = [&]() {                                   // move to the next owned tile, skipping the look-ahead tile
                while (J < Tc && (I >= Tr || (I == j + 1 && J == j + 1))) {
                    if (I >= Tr) { J++; I = J < Tc ? first_row(J) : Tr; } else I += 15;
                }
            };
            settle();
            int Ic[4], Jc[4], nc = 0;
            double4_t vc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { vc[t] = double4_t{ 0, 0, 0, 0 }; Ic[t] = 0; Jc[t] = 0; }
#pragma unroll
            for (int t = 0; t < 4; t++) if (J < Tc) { Ic[t] = I; Jc[t] = J; vc[t] = load_tile(I, J, first); nc = t + 1; I += 15; settle(); }
            while (nc) {
                int In[4], Jn[4], nn = 0;
                double4_t vn[4];
#pragma unroll
                for (int t = 0; t < 4; t++) { vn[t] = double4_t{ 0, 0, 0, 0 }; In[t] = 0; Jn[t] = 0; }
#pragma unroll
                for (int t = 0; t < 4; t++) if (J < Tc) { In[t] = I; Jn[t] = J; vn[t] = load_tile(I, J, first); nn = t + 1; I += 15; settle(); }
#pragma unroll
                for (int kk = 0; kk < 4; kk++)
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (t < nc) vc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn[Ic[t]][li][lk + 4 * kk], Pn[Jc[t]][li][lk + 4 * kk], vc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; t++) if (t < nc) store_tile(Ic[t], Jc[t], vc[t], false);
#pragma unroll
                for (int t = 0; t < 4; t++) { Ic[t] = In[t]; Jc[t] = Jn[t]; vc[t] = vn[t]; }
                nc = nn;
            }
        }
    }
    // z = L^-1 rhs sits in row n of the L buffer: this wave's tiles of the rhs row -> yv
    for (int J = 0; J < Tc; J++)
        if ((Tc + J) % 15 == kq && lk == 0 && 16 * J + li < n) yv[16 * J + li] = Lw[(size_t)n * n + 16 * J + li];
    for (int e = tid - 64; e < CB_NMAX + 16; e += 960) zs[e] = 0.0;
    __syncthreads();                                       // E
    // tiles (J, J') of row J, J' < J, owned by this wave: J' = (kq - J) mod 15, + 15, + 30.  The tiles of row J - 1 are
    // requested before row J is applied (L is final: the loads do not depend on y), so no step waits for its loads
    double4_t t[3], tn[3];
    int Jp[3], Jpn[3], nt_ = 0, ntn = 0;
#pragma unroll
    for (int u = 0; u < 3; u++) { t[u] = double4_t{ 0, 0, 0, 0 }; tn[u] = t[u]; Jp[u] = 0; Jpn[u] = 0; }
    auto fetch_row = [&](int J, double4_t* tv, int* jp, int& cnt) {
        cnt = 0;
        if (J < 0) return;
        int d = (kq - J) % 15; if (d < 0) d += 15;
#pragma unroll
        for (int u = 0; u < 3; u++) { int Jq = d + 15 * u; if (Jq < J) { jp[u] = Jq; tv[u] = load_tile(J, Jq, false); cnt = u + 1; } }
    };
    fetch_row(Tc - 1, t, Jp, nt_);
    for (int J = Tc - 1; J >= 0; J--) {
        fetch_row(J - 1, tn, Jpn, ntn);
        __syncthreads();                                   // X_J: y_J published
#pragma unroll
        for (int u = 0; u < 3; u++) if (u < nt_) {
            double p = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) p += t[u][q] * zs[16 * J + lk + 4 * q];
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (lk == 0) yv[16 * Jp[u] + li] -= p;
        }
        __syncthreads();                                   // Y_J
#pragma unroll
        for (int u = 0; u < 3; u++) { t[u] = tn[u]; Jp[u] = Jpn[u]; }
        nt_ = ntn;
    }
    double* y = B.y + W.loc_base + W.n_e;
    for (int e = tid - 64; e < n; e += 960) y[e] = zs[e];
}

// k_chol_col — two tile columns (j, j + 1) of the k_chol_big factorisation per launch, spread over CC_NB workgroups per
// window, for the latency path (a single cfg5-class window would otherwise factor on one CU): every workgroup re-factors
// the two 16x16 pivot tiles and re-forms both panels (cheap, and bit-identical in each; the tiles of column j + 1 take
// column j's update in registers), then applies both rank-16 updates to its share of the trailing tiles.  The trailing
// matrix lives in a working copy (B.Wk) and the finished columns go to the L buffer, so no workgroup overwrites what
// another still reads; k_chol_big<true> does the backward substitution.  Every tile sees the same MFMA sequence as in
// k_chol_big: the two paths give bit-identical factors.
#define CC_NB 32                        // workgroups per window, at most (the engine divides the chip by the window count)
#define CC_NT 512
__global__ void __launch_bounds__(CC_NT) k_chol_col(DevBatch B, int j) {
    __shared__ double Pn[CC_MAXT][16][17];      // panel of column j
    __shared__ double Pn2[CC_MAXT][16][17];     // panel of column j + 1 (2 x 72 KB)
    __shared__ double Lic[16][17];
    __shared__ double ipiv[16];
    __shared__ double Dt[16][17];
    __shared__ int fail;
    const int w = blockIdx.x, g = blockIdx.y, nbw = gridDim.y;      // nbw workgroups share the window (ownership only: the arithmetic of a tile does not depend on it)
    WinState& st = B.ws[w];
    if (!st.need_lin || st.lin_fail) return;
    const WinRec& W = B.win[w];
    const int n = W.n_red, tid = threadIdx.x;
    if (n <= B.rr_nmax) return;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int Tc = (n + 15) >> 4, Tr = Tc + 1;
    if (j >= Tc) return;
    const bool first = j == 0;
    const double* S = B.S + W.S_base;
    double* Lw = B.L + W.Lt_base;
    double* Wk = B.Wk + W.Lt_base;
    double* LinvG = B.Linv + (size_t)w * (CB_MAXT - 1) * 256;
    if (tid == 0) fail = 0;
    const int lane_off = lk * n + li;
    auto load_tile = [&](int I, int J) {
        double4_t v;
        const double* src = first ? S : Wk;
        if (16 * I + 16 <= n && !(first && I == J)) {
            const double* t0 = src + (size_t)(16 * I) * n + 16 * J + lane_off;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = t0[(size_t)(4 * q) * n];
            return v;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * I + lk + 4 * q, c = 16 * J + li;
            bool rhs_el = I == Tc && lk + 4 * q == 0;
            int rr = rhs_el ? n : r;
            bool inside = c < n && (rhs_el || (r < n && (c <= r || I == J)));
            int rc = rr < n ? rr : (rhs_el ? n : n - 1), cc = c < n ? c : n - 1;
            int off = (first && cc > rc) ? cc * n + rc : rc * n + cc;
            double x = src[off];
            v[q] = inside ? x : ((I < Tc && r == c) ? 1.0 : 0.0);
        }
        return v;
    };
    auto store_tile = [&](double* dst, int I, int J, double4_t v, bool lower_only) {
        if (16 * I + 16 <= n && !lower_only) {
            double* t0 = dst + (size_t)(16 * I) * n + 16 * J + lane_off;
#pragma unroll
            for (int q = 0; q < 4; q++) t0[(size_t)(4 * q) * n] = v[q];
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * I + lk + 4 * q, c = 16 * J + li;
            bool rhs_el = I == Tc && lk + 4 * q == 0;
            if (rhs_el) { if (c < n) dst[(size_t)n * n + c] = v[q]; }
            else if (I < Tc && r < n && c < n) dst[(size_t)r * n + c] = (lower_only && c > r) ? 0.0 : v[q];
        }
    };
    // ---- the loads that do not depend on the pivots go out first: this wave's tiles of columns j and j + 1 and its first
    //      trailing tiles
    constexpr int NW = CC_NT / 64;                         // waves per workgroup
    constexpr int PPW = (CC_MAXT - 1 + NW - 1) / NW;        // column tiles per wave, at most
    const bool two = j + 1 < Tc;                           // this launch also finishes column j + 1
    const int jt = two ? j + 2 : j + 1;                    // first trailing column
    double4_t pv[PPW], pw[PPW];
#pragma unroll
    for (int u = 0; u < PPW; u++) {
        int I = j + 1 + wv + NW * u;
        pv[u] = double4_t{ 0, 0, 0, 0 }; pw[u] = pv[u];
        if (I < Tr) { pv[u] = load_tile(I, j); if (two) pw[u] = load_tile(I, j + 1); }
    }
    // trailing tiles (I, J), J >= jt, I >= J (the rhs row included): tile (I, J) belongs to wave (I + 33 J) mod (NW nbw) of
    // the window, so a wave owns at most one tile per column and finds it without walking the others; four in flight
    const int NWT = NW * nbw;
    const int me = g * NW + wv;
    int tJ = jt;                                           // column cursor
    auto next_owned = [&](int& I, int& J) {                // advance to this wave's next tile; false when exhausted
        while (tJ < Tc) {
            int c = tJ++;
            int r = (me - 33 * c) % NWT; if (r < 0) r += NWT;
            if (r >= c && r < Tr) { I = r; J = c; return true; }
        }
        return false;
    };
    int Ia[4], Ja[4], na = 0;
    double4_t va[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { Ia[u] = 0; Ja[u] = 0; va[u] = double4_t{ 0, 0, 0, 0 }; }
#pragma unroll
    for (int u = 0; u < 4; u++) if (na == u && next_owned(Ia[u], Ja[u])) { va[u] = load_tile(Ia[u], Ja[u]); na = u + 1; }
    __syncthreads();                                       // fail = 0 visible
    // pivot of column c from the tile in Dt (every workgroup, wave 0); workgroup 0 keeps Linv_cc and L_cc
    auto pivot = [&](int c) {
        bool bad = chol_pivot_tile(Dt, Lic, li, lk, ipiv);
        if (bad && lane == 0) fail = 1;
        if (g == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) LinvG[(size_t)c * 256 + (lk + 4 * q) * 16 + li] = Lic[lk + 4 * q][li];
            double4_t l;
#pragma unroll
            for (int q = 0; q < 4; q++) l[q] = Dt[lk + 4 * q][li];
            store_tile(Lw, c, c, l, true);
        }
    };
    // panel tile I of column c: X = A Linv_cc^T, left in Pc[I]; stored to L by workgroup I mod nbw
    auto panel = [&](double (*Pc)[16][17], int I, int c, double4_t a) {
#pragma unroll
        for (int q = 0; q < 4; q++) Pc[I][lk + 4 * q][li] = a[q];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double4_t X = { 0, 0, 0, 0 };
#pragma unroll
        for (int kk = 0; kk < 4; kk++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[I][li][lk + 4 * kk], Lic[li][lk + 4 * kk], X, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; q++) Pc[I][lk + 4 * q][li] = X[q];
        if (I % nbw == g) store_tile(Lw, I, c, X, false);
    };
    // ---- column j
    if (wv == 0) {
        double4_t v = load_tile(j, j);
#pragma unroll
        for (int q = 0; q < 4; q++) Dt[lk + 4 * q][li] = v[q];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        pivot(j);
    }
    __syncthreads();
    if (fail) { if (g == 0 && tid == 0) { st.lin_fail = 1; st.chol_fail = 1; } return; }
#pragma unroll
    for (int u = 0; u < PPW; u++) { int I = j + 1 + wv + NW * u; if (I < Tr) panel(Pn, I, j, pv[u]); }
    __syncthreads();
    if (two) {
        // ---- column j + 1: its tiles take the update of column j (the same four MFMAs as any trailing tile), the diagonal
        //      one goes to the pivot wave, the others become the panel of column j + 1
#pragma unroll
        for (int u = 0; u < PPW; u++) {
            int I = j + 1 + wv + NW * u;
            if (I < Tr) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) pw[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn[I][li][lk + 4 * kk], Pn[j + 1][li][lk + 4 * kk], pw[u], 0, 0, 0);
                if (I == j + 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) Dt[lk + 4 * q][li] = pw[u][q];
                }
            }
        }
        __syncthreads();                                   // the updated diagonal tile is in Dt (written by wave 0 itself: I = j + 1)
        if (wv == 0) pivot(j + 1);
        __syncthreads();
        if (fail) { if (g == 0 && tid == 0) { st.lin_fail = 1; st.chol_fail = 1; } return; }
#pragma unroll
        for (int u = 0; u < PPW; u++) { int I = j + 1 + wv + NW * u; if (I > j + 1 && I < Tr) panel(Pn2, I, j + 1, pw[u]); }
        __syncthreads();
    }
    // ---- trailing updates (both columns' rank-16 updates, column j first: the order of k_chol_big), groups of four
    //      independent accumulators, the next group's loads in flight
    while (na) {
        int In[4], Jn[4], nn = 0;
        double4_t vn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { In[u] = 0; Jn[u] = 0; vn[u] = double4_t{ 0, 0, 0, 0 }; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (nn == u && next_owned(In[u], Jn[u])) { vn[u] = load_tile(In[u], Jn[u]); nn = u + 1; }
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (u < na) va[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn[Ia[u]][li][lk + 4 * kk], Pn[Ja[u]][li][lk + 4 * kk], va[u], 0, 0, 0);
        if (two) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (u < na) va[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn2[Ia[u]][li][lk + 4 * kk], Pn2[Ja[u]][li][lk + 4 * kk], va[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (u < na) store_tile(Wk, Ia[u], Ja[u], va[u], false);
#pragma unroll
        for (int u = 0; u < 4; u++) { Ia[u] = In[u]; Ja[u] = Jn[u]; va[u] = vn[u]; }
        na = nn;
    }
}

// =========================================================================================
// Back-substitution of the eliminated blocks: y_e = Einv (g_e - H_ef y_f)
// =========================================================================================
__device__ __forceinline__ void d_backsub_lm(const DevBatch& B, const DevOpt& O, int bid, double (*sc)[256]) {
    // One lane per observation (a block = up to 256 consecutive observations = whole landmarks of one window, host-built table), so every
    // lane of the Jacobian loads is live — 16 lanes per landmark left 6 of them idle at the mean track length of 10.  Two things share the
    // observation's Jacobians here:
    //   back-substitution  t = g_l - sum_o W_o^T y_pose(o),  W_o^T y = Jl_o^T (Jp_o y)   (W is not stored)
    //   Cauchy point       aux_o = |Jp_o v_pose + Jl_o v_l|^2,  v = D^-2 g            (the projection part of |J D^-2 g|^2)
    // The per-observation terms of t are staged in LDS and one lane per landmark adds its track in observation order.
    // Loads by dependency level, each level issued as a whole (round 3 walked them one after the other — record -> window -> flags ->
    // landmark -> offsets -> values, ~8 exposed round trips of ~1.2 us on the latency path): (1) the block record; (2) every index and
    // constant that hangs off it (clamped addresses: unconditional loads); (3) the window's flags, the state values, y and D^-2 g at
    // the pose; (4) D^-2 g at the landmark.
    const int4 rec = ((const int4*)B.lmb_rec)[bid];
    const int o0 = rec.x, cnt = rec.y, L0 = rec.z, nlm = rec.w, tid = threadIdx.x;
    const int nl = B.n_lm, n = B.n_proj;
    const int win = B.lm_win[L0];
    const bool ho = tid < cnt, hl = tid < nlm;
    const int o = ho ? o0 + tid : o0, L = hl ? L0 + tid : L0;
    const int plm = B.p_lm[o], lp = B.p_lpose[o], xpo = B.p_xpose[o], xex = B.p_xex[o], xlm = B.p_xlm[o], pfr = B.p_fr[o], pll = B.p_llm[o];
    const double u0 = B.p_uv[2 * o], u1 = B.p_uv[2 * o + 1];
    const int lloc = B.lm_loc[L], lob = B.lm_obs0[L], loe = B.lm_obs0[L + 1];
    const double g0 = B.lm_g[0 * nl + L], g1 = B.lm_g[1 * nl + L], g2 = B.lm_g[2 * nl + L];
    const double e00 = B.lm_Einv[0 * nl + L], e10 = B.lm_Einv[1 * nl + L], e20 = B.lm_Einv[2 * nl + L];
    const double e11 = B.lm_Einv[3 * nl + L], e21 = B.lm_Einv[4 * nl + L], e22 = B.lm_Einv[5 * nl + L];
    const WinState& s = B.ws[win];
    const WinRec& W = B.win[win];
    const int need = s.need_lin, failed = s.lin_fail;
    const int loc = B.lm_loc[plm];
    double P[7], E[7], X[3], yp[6], vp[6];
#pragma unroll
    for (int k = 0; k < 7; k++) { P[k] = B.x[xpo + k]; E[k] = B.x[xex + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) X[k] = B.x[xlm + k];
#pragma unroll
    for (int k = 0; k < 6; k++) { const int q = lp >= 0 ? lp + k : 0; yp[k] = B.y[q]; vp[k] = B.vc[q]; }
    double vl0 = 0, vl1 = 0, vl2 = 0;
    { const int q = loc >= 0 ? loc : 0; vl0 = B.vc[q]; vl1 = B.vc[q + 1]; vl2 = B.vc[q + 2]; }
    if (loc < 0) { vl0 = 0; vl1 = 0; vl2 = 0; }
    if (!need) return;                                         // uniform: the block belongs to one window
    double c0 = 0, c1 = 0, c2 = 0, aux = 0;
    if (ho) {
        // the observation's Jacobian, re-derived at the linearisation point from its inputs (pose, extrinsic and landmark sit in cache
        // for the whole track; 16 B of image coordinates per observation) rather than read back: 144 B per observation less traffic
        double kj[20];
#pragma unroll
        for (int k = 0; k < 20; k++) kj[k] = 0.0;
        proj_core<true, false>(B, o, W, P, E, X, u0, u1, lp >= 0, pll >= 0, kj);
        double jl0 = kj[14], jl1 = kj[15], jl2 = kj[16], jl3 = kj[17], jl4 = kj[18], jl5 = kj[19];
        double w0 = 0, w1 = 0;                                     // Jp y
        double a0 = jl0 * vl0 + jl1 * vl1 + jl2 * vl2, a1 = jl3 * vl0 + jl4 * vl1 + jl5 * vl2;   // J v
        if (lp >= 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double ja = kj[i], jb = kj[6 + i];
                w0 += ja * yp[i]; w1 += jb * yp[i];
                a0 += ja * vp[i]; a1 += jb * vp[i];
            }
        }
        aux = a0 * a0 + a1 * a1;
        if (pfr >= 0) { c0 = -(jl0 * w0 + jl3 * w1); c1 = -(jl1 * w0 + jl4 * w1); c2 = -(jl2 * w0 + jl5 * w1); }
    }
    (void)n;
    sc[0][tid] = c0; sc[1][tid] = c1; sc[2][tid] = c2;
    {   // the block's share of |J D^-2 g|^2 (k_dogleg adds the window's blocks in order)
        __shared__ double asum[16];
        aux = block_sum(aux, asum);                      // (its barriers also order the staging above before the reads below)
        if (tid == 0) B.p_apart[bid] = aux;
    }
    if (!hl || failed || lloc < 0) return;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int q = lob - o0, qe = loe - o0; q < qe; q++) { t0 += sc[0][q]; t1 += sc[1][q]; t2 += sc[2][q]; }
    t0 += g0; t1 += g1; t2 += g2;
    B.y[lloc] = e00 * t0 + e10 * t1 + e20 * t2;
    B.y[lloc + 1] = e10 * t0 + e11 * t1 + e21 * t2;
    B.y[lloc + 2] = e20 * t0 + e21 * t1 + e22 * t2;
}
__device__ __forceinline__ void d_backsub_clique(const DevBatch& B, int bid) {
    // 16 lanes per clique with an eliminated block (d_e <= 9).  The strip product M_ef y_f is split by column over the
    // lanes (lane s takes columns s, s+16, ...: coalesced rows of the strip), the 16
    // partials of each row are added by the fixed butterfly, then lane a holds t_a = g_e[a] - (M_ef y_f)_a and the
    // Einv product gathers the t's with 16-wide shuffles.
    int q = (bid * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15, lane = threadIdx.x & 63;
    bool valid = q < B.n_cle;
    const Clique& C = B.cle_rec[valid ? q : B.n_cle - 1];
    const WinState& s = B.ws[C.win];
    bool act = valid && s.need_lin && !s.lin_fail;
    int de = C.d_e, df = C.d_f;
    const double* E = B.cE + C.e_off;
    const double* M = E + de * de;
    double part[9];
#pragma unroll
    for (int a = 0; a < 9; a++) part[a] = 0;
    // column c of the strip -> local index of its variable: host-built table at the clique's vector slot (cv_loc[v_off + c]; round 3
    // searched the member records with 16-wide shuffles per column chunk — 12 us of the latency path's back-substitution launch were this
    // loop).  Same products in the same order (a lane's columns ascending).
    const int* cl = B.cv_loc + C.v_off;
    for (int c = sub; c < (act ? df : 0); c += 16) {
        const double yv = B.y[cl[c]];
#pragma unroll
        for (int a = 0; a < 9; a++) if (a < de) part[a] += M[a * df + c] * yv;
    }
    double t = 0;
#pragma unroll
    for (int a = 0; a < 9; a++) { double pa = grp16_sum(part[a]); if (sub == a) t = pa; }
    if (act && sub < de) t = M[de * df + sub] - t;
    double acc = 0;
#pragma unroll
    for (int b = 0; b < 9; b++) {
        double tb = __shfl(t, (lane & ~15) + b, 64);
        if (act && sub < de && b < de) acc += E[sub * de + b] * tb;
    }
    if (act && sub < de) B.y[C.e_loc + sub] = acc;
}


// =========================================================================================
// Fused launches.  Kernels that are mutually independent at the same point of the iteration
// (no LDS, 256 threads) run as segments of ONE grid: the block id selects the segment.  On the
// single-window latency path this removes ~12 dependent dispatches per iteration; in a batch
// it lets the small segments fill the tail of the large ones.
// =========================================================================================
struct Segs { int e[8]; };     // exclusive end block of segment k (cumulative)

// Jacobian/residual evaluation of the one-lane-per-factor families: projection + scalar GNSS/prior factors
// FS (Jacobian evaluations): the projection segment runs one workgroup per frame-sum block and leaves the per-frame partial sums of
// Jp^T Jp | Jp^T r next to the Jacobians (d_eval_proj_fs).
// IMU (latency path, few windows): the IMU factors ride along as a fourth segment (8 factors per workgroup) instead of their own launch
// behind this one — the two evaluations are independent, and on the latency path a launch costs its whole dependent-load chain
// (one window: 7.6 + 11.8 us as two kernels).  Large batches keep k_eval_imu apart: its LDS and registers would cost the HBM-bound
// segments their occupancy.  Same device functions, same results.
template <bool JAC, bool FS = false, bool IMU = false>
__global__ void __launch_bounds__(256) k_eval_ps(DevBatch B, Segs S) {
    // one LDS buffer for whichever segment the block runs: the prior's staging vectors, or the frame sums' staging tile (+ its frame offsets)
    constexpr int SM_PRIOR = 2 * PRIOR_LDS_DIM + 16, SM_FS = FS ? FS_BLK * FS_HALF + 168 / 2 + 1 : 1;
    __shared__ double sm[SM_PRIOR > SM_FS ? SM_PRIOR : SM_FS];
    int bid = blockIdx.x;
    if (bid < S.e[0]) { static_assert(FS && JAC, "the projection segment evaluates Jacobians by frame-sum block"); d_eval_proj_fs(B, bid, (double (*)[FS_HALF])sm, (int*)(sm + FS_BLK * FS_HALF)); }
    else if (bid < S.e[1]) d_eval_scalar<JAC>(B, bid - S.e[0]);
    else if (!IMU || bid < S.e[2]) d_eval_prior<JAC>(B, bid - S.e[1], sm);              // one workgroup per prior (segment empty for large priors)
    else d_eval_imu<JAC>(B, bid - S.e[2]);
}
// after the reduced solve: back-substitution of the eliminated blocks, and |J D^-2 g|^2 for the Cauchy point.
// PART 0: every segment in one grid (latency path).  Large batches launch the landmark segment (PART 1, the
// HBM-bound one) apart from the rest (PART 2) so that it keeps its own, lower register count and full occupancy.
template <int PART>
__global__ void __launch_bounds__(256) k_post_chol(DevBatch B, DevOpt O, Segs S) {
    int bid = blockIdx.x + (PART == 2 ? S.e[0] : 0);
    __shared__ double sm_bs[PART == 2 ? 1 : 3][256];
    __shared__ double sm_pv[PART == 1 ? 1 : PRB_MAX]; __shared__ int sm_pl[PART == 1 ? 1 : PRB_MAX]; __shared__ double sm_pw[4];
    if (PART != 2 && bid < S.e[0]) { d_backsub_lm(B, O, bid, sm_bs); return; }   // also the projection part of |J D^-2 g|^2
    if (PART == 1) return;
    if (bid < S.e[1]) d_backsub_clique(B, bid - S.e[0]);
    else if (bid < S.e[3]) d_jtimes_scalar<0>(B, O, bid - S.e[2]);
    else if (bid < S.e[4]) d_jtimes_imu<0>(B, O, bid - S.e[3]);
    else d_jtimes_prior<0>(B, O, bid - S.e[4], sm_pv, sm_pl, sm_pw);
}
// after the dogleg step: the candidate residuals (costs) of every factor family.  The model cost change needs no pass over the
// Jacobians any more: k_dogleg gets it from vectors alone (see there).
// PART 0: every segment in one grid (latency path).  Large batches launch the projection segment (PART 1, the HBM-bound one,
// at its own register count) apart from the small latency-bound families (PART 2).
template <bool WITH_IMU, int PART>
__global__ void __launch_bounds__(256) k_post_dogleg(DevBatch B, DevOpt O, Segs S) {
    __shared__ double sm_prior[PART == 1 ? 1 : 2 * PRIOR_LDS_DIM + 16];
    int bid = blockIdx.x;
    if (PART == 1) { d_eval_proj_cost(B, bid); return; }       // grid = S.e[0] = the frame-sum blocks
    if (PART == 2) bid += S.e[0];                              // skip the projection segment
    if (PART == 0 && bid < S.e[0]) d_eval_proj_cost(B, bid);
    else if (bid < S.e[1]) d_eval_scalar<false>(B, bid - S.e[0]);
    else if (bid < S.e[2]) d_eval_prior<false>(B, bid - S.e[1], sm_prior);
    else if (WITH_IMU) d_eval_imu<false>(B, bid - S.e[2]);    // candidate IMU residuals (8 factors per workgroup), small batches only
}

// =========================================================================================
// per-window control kernels (one 256-thread workgroup per window)
// =========================================================================================
// A window's cost = (its frame-sum blocks' projection costs, added in block order) + (its generic factors' costs: per-thread strided
// sums in index order, wave butterfly, waves in order).  The SAME formula in k_dogleg, k_decide and k_finalize, in every launch shape.
// The block values travel through LDS: one load per lane, all in flight at once (a sequential loop over global memory would be a chain of
// round trips), staged BEFORE a block reduction and added, in block order, by every thread AFTER it (the reduction's barriers order the two).
#define WP_LDS 256              // block values staged per window (a cfg3 window has 12 of each; beyond: read in place, in order)
__device__ __forceinline__ void win_part_stage(const double* part, int q0, int q1, double* st) {
    for (int q = q0 + (int)threadIdx.x; q < q1 && q - q0 < WP_LDS; q += (int)blockDim.x) st[q - q0] = part[q];
}
__device__ __forceinline__ double win_part_sum(const double* part, int q0, int q1, const double* st) {
    double a = 0;
    const int n = q1 - q0;
    for (int q = 0; q < n && q < WP_LDS; q++) a += st[q];
    for (int q = WP_LDS; q < n; q++) a += part[q0 + q];
    return a;
}
// cost = ((projection blocks + prior chunks) + generic factors)
__device__ __forceinline__ double win_cost_sum(const DevBatch& B, const WinRec& W, double* red, double* st /* 2 WP_LDS */) {
    win_part_stage(B.p_cpart, W.fsb0, W.fsb1, st);
    win_part_stage(B.pr_cpart, W.pch0, W.pch1, st + WP_LDS);
    double a = 0;
    for (int i = W.gf0 + threadIdx.x; i < W.gf1; i += blockDim.x) a += B.g_cost[i];
    a = block_sum(a, red);
    return (win_part_sum(B.p_cpart, W.fsb0, W.fsb1, st) + win_part_sum(B.pr_cpart, W.pch0, W.pch1, st + WP_LDS)) + a;
}
// || x ||_2 over variable blocks (ambient coordinates)
__device__ __forceinline__ double win_x_norm(const DevBatch& B, const WinRec& W, const double* x, double* red) {
    double a = 0;
    for (int b = W.blk_base + threadIdx.x; b < W.blk_base + W.n_blk; b += blockDim.x) {
        if (B.blk_loc[b] < 0) continue;
        for (int k = 0; k < B.blk_gs[b]; k++) { double v = x[B.blk_xoff[b] + k]; a += v * v; }
    }
    return sqrt(block_sum(a, red));
}
// gradient_max_norm = || x - Plus(x, -g) ||_inf (TrustRegionMinimizer::EvaluateGradientAndJacobian)
__device__ __forceinline__ double win_gmax_part(const DevBatch& B, const WinRec& W) {
    double m = 0;
    for (int b = W.blk_base + threadIdx.x; b < W.blk_base + W.n_blk; b += blockDim.x) {
        int lo = B.blk_loc[b];
        if (lo < 0) continue;
        const double* x = B.x + B.blk_xoff[b];
        if (B.blk_gs[b] == 7) {
            double d[6], o[7];
            for (int k = 0; k < 6; k++) d[k] = -B.g[lo + k];
            pose_plus(x, d, o);
            for (int k = 0; k < 7; k++) { double v = fabs(x[k] - o[k]); m = v > m ? v : m; }
        } else {
            for (int k = 0; k < B.blk_gs[b]; k++) { double v = fabs(B.g[lo + k]); m = v > m ? v : m; }
        }
    }
    return m;
}
__device__ __forceinline__ double win_gmax(const DevBatch& B, const WinRec& W, double* red) { return block_max(win_gmax_part(B, W), red); }

__global__ void __launch_bounds__(256) k_init(DevBatch B, DevOpt O) {
    __shared__ double red[16];
    int w = blockIdx.x;
    const WinRec& W = B.win[w];
    double xa = 0;
    for (int i = W.x_base + threadIdx.x; i < W.x_base + W.x_n; i += blockDim.x) { const double xv = B.x[i]; if (B.x_var[i]) xa += xv * xv; }
    double xn = sqrt(block_sum(xa, red));
    for (int k = threadIdx.x; k < B.max_iter_trace * (int)(sizeof(swf_iteration) / 8); k += blockDim.x)
        ((double*)(B.trace + (size_t)w * B.max_iter_trace))[k] = 0.0;
    if (threadIdx.x == 0) {
        WinState& s = B.ws[w];
        // Levenberg-Marquardt: the damping of the linear solve is D^2 / radius, i.e. mu = 1 / radius on the clamped diagonal
        s.radius = O.r0; s.mu = (O.step_mode == SWF_ASSEMBLE_ELIMINATE_ONLY) ? 0.0 : (O.strategy == SWF_LEVENBERG_MARQUARDT ? 1.0 / O.r0 : O.min_mu);
        s.lm_dec = 2.0;
        s.x_cost = 0; s.x_norm = xn; s.alpha = 0; s.dogleg_step_norm = 0; s.step_norm = 0; s.gmax = 0;
        s.jg_sq = 0; s.initial_cost = 0;
        s.status = SWF_RUNNING; s.iter = 0; s.need_lin = 1; s.reuse = 0; s.eval_cand = 0; s.lin_fail = 0; s.chol_fail = 0;
        s.invalid_run = 0; s.nsucc = 0; s.nunsucc = 0;
    }
}

// DoglegStrategy::ComputeStep (+ the bookkeeping of FinalizeIterationAndCheckIfMinimizerCanContinue)
#ifdef SWF_PROFILE_DOG
__device__ unsigned long long g_dog_stamps[16];
#define DST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_dog_stamps[i] += t_ - td_; td_ = t_; } } while (0)
#else
#define DST(i)
#endif
// threads of the per-window control kernels k_dogleg / k_decide: their loops over the window's dimensions and cost terms are chains of
// dependent loads, so a window's latency falls with the thread count (one window: k_dogleg 20.7 -> ? us); the same count for every
// batch size, because the order of the strided partial sums depends on it
#define CTL_NT 256
// Latency notes (one window: 46 k cycles in round 3, most of them chains of dependent loads — block table -> offsets -> values, three
// blocks per thread one after the other, twice): the passes below run over FLAT host-built tables instead.  loc2x[i] = ambient
// coordinate of local dimension i (-1 for the six dimensions of a pose block, which a thread per pose block handles with Plus);
// every load of a pass depends on nothing but the window record, so a pass is one round trip.  Constant blocks are never written:
// xc holds their values since the upload (swf_batch_upload_state / _reset_state copy x to xc).
// DU / GU: strided elements per thread whose loads are issued up front and kept in registers (the latency path takes a cfg3 window
// whole: 256 registers, one workgroup per CU; batches take a quarter and walk the rest in the loops behind — two workgroups per CU and
// more: 512 windows 33.5 -> ? us).  The sums run in index order either way: same bits.
// The body of k_dogleg as a device function, so that the latency path can run it at the head of its candidate evaluation (k_step_eval:
// EVERY workgroup of that grid forms the step redundantly — same loads, same sums in the same order, same bits — and goes on to evaluate
// its share of the factors at the candidate it holds in LDS; the launch of k_dogleg, its dependent round trips and the candidate's trip
// through HBM disappear).  sin: the window's state as the previous kernels left it (read only); sout: where the LEAD workgroup's thread 0
// leaves the new state (== sin for the in-place form of the k_dogleg launch; another buffer in k_step_eval, whose other workgroups are
// still reading sin).  Only the lead writes global memory (state, trace, step, candidate).  xcl != nullptr: the candidate's coordinates,
// window-relative, for this workgroup's evaluation (filled here: x, then the candidate over it; constant blocks keep x's values;
// x_n <= XU * 256).  W: the window's record — in k_step_eval a kernel argument, so that every load below but the pose threads' second
// level depends on nothing but the kernel's arguments: ONE round trip ahead of the sums (the record and the state were one of their own).
// Returns 1 (uniform over the workgroup) if a candidate was formed.
template <int DU, int GU, int XU = 1>
__device__ __forceinline__ int d_dogleg(const DevBatch& B, const DevOpt& O, const int w, const WinRec& W, const WinState* sin, WinState* sout, const bool lead, double* xcl, double* red, double* pst /* LDS, 4 WP_LDS doubles */) {
#ifdef SWF_PROFILE_DOG
    unsigned long long td_ = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 16; i++) g_dog_stamps[i] = 0;
#endif
    const int tid = threadIdx.x;
    // every thread carries the window's state (uniform values) through the same bookkeeping; the lead's thread 0 stores it
    WinState t = *sin;
    const bool wr = lead && tid == 0;
    swf_iteration* tr = B.trace + (size_t)w * B.max_iter_trace;
    const double* g = B.g + W.loc_base; const double* dg = B.diag + W.loc_base; const double* y = B.y + W.loc_base;
    const int* l2x = B.loc2x + W.loc_base;
    double* step = B.step + W.loc_base;
    const int n = W.n_loc, xb = W.x_base;
    // one pass, one barrier pair: cost, |J D^-2 g|^2, gradient max-norm (fresh linearisation only) and the scalars of the
    // scaled problem |g/d|^2, |d y|^2, (g/d).(-d y).
    // EVERY load of the kernel is issued here, before the first value is used: the first GU / DU strided elements per thread of the
    // cost arrays and of g, diag, y, loc2x stay in registers (a cfg3 window: all of them) and serve the step pass below as well, which
    // then touches no memory but its stores.  Two exposed round trips (the tables, then x through loc2x) instead of eight.
    double v[6] = { 0, 0, 0, 0, 0, 0 };
    DST(0);
    // the pose blocks (the window's first blocks; constant ones have no local dimensions): a thread each
    int p_lo = -1, p_xo = 0;
    if (tid < W.n_pose_blk) { p_lo = B.blk_loc[W.blk_base + tid]; p_xo = B.blk_xoff[W.blk_base + tid]; }
    double cvg[GU], avg[GU], gv[DU], yv[DU], dgv[DU], xv[DU];
    int lxv[DU];
    // the projection factors' share of the cost and of |J D^-2 g|^2: the blocks' partial sums (left by the evaluation and by k_post_chol)
    // (none of these loads waits for the window's state: whether the linearisation is fresh decides only what is added up)
    win_part_stage(B.p_cpart, W.fsb0, W.fsb1, pst); win_part_stage(B.p_apart, W.lmb0, W.lmb1, pst + WP_LDS);
    win_part_stage(B.pr_cpart, W.pch0, W.pch1, pst + 2 * WP_LDS); win_part_stage(B.pr_apart, W.pch0, W.pch1, pst + 3 * WP_LDS);
    double xin[XU];
    if (xcl) {
#pragma unroll
        for (int u = 0; u < XU; u++) { int i = tid + u * CTL_NT; xin[u] = i < W.x_n ? B.x[xb + i] : 0.0; }
    }
#pragma unroll
    for (int u = 0; u < GU; u++) { int i = W.gf0 + tid + u * CTL_NT; bool ok = i < W.gf1; cvg[u] = ok ? B.g_cost[i] : 0.0; avg[u] = ok ? B.g_aux[i] : 0.0; }
#pragma unroll
    for (int u = 0; u < DU; u++) { int i = tid + u * CTL_NT; bool ok = i < n; gv[u] = ok ? g[i] : 0.0; yv[u] = ok ? y[i] : 0.0; dgv[u] = ok ? dg[i] : 1.0; lxv[u] = ok ? l2x[i] : -1; }
    double xp[7] = { 0, 0, 0, 0, 0, 0, 1 }, pg[6] = { 0, 0, 0, 0, 0, 0 }, pd[6] = { 1, 1, 1, 1, 1, 1 }, py[6] = { 0, 0, 0, 0, 0, 0 };
    if (p_lo >= 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) xp[k] = B.x[p_xo + k];
#pragma unroll
        for (int k = 0; k < 6; k++) { pg[k] = B.g[p_lo + k]; pd[k] = B.diag[p_lo + k]; py[k] = B.y[p_lo + k]; }
    }
    if (!xcl) {
#pragma unroll
        for (int u = 0; u < DU; u++) xv[u] = lxv[u] >= 0 ? B.x[lxv[u]] : 0.0;
    } else {
        // x into LDS (the candidate is formed over it); this thread's coordinates come back from there behind the first reduction
#pragma unroll
        for (int u = 0; u < XU; u++) { int i = tid + u * CTL_NT; if (i < W.x_n) xcl[i] = xin[u]; }
    }
    if (t.status != SWF_RUNNING) { if (wr && sout != sin) *sout = t; return 0; }          // (uniform)
    const int fresh = t.need_lin;
    double x_cost = t.x_cost, gmax = t.gmax, jg_sq = t.jg_sq;
    const bool want_gmax = fresh && !t.lin_fail;
    if (fresh) {
        // (the generic factors: index order per thread)
#pragma unroll
        for (int u = 0; u < GU; u++) { v[0] += cvg[u]; v[1] += avg[u]; }
        for (int base = W.gf0 + tid + GU * CTL_NT; base < W.gf1; base += GU * CTL_NT) {
            double cv[GU], av[GU];
#pragma unroll
            for (int u = 0; u < GU; u++) { int i = base + u * CTL_NT; bool ok = i < W.gf1; cv[u] = ok ? B.g_cost[i] : 0.0; av[u] = ok ? B.g_aux[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < GU; u++) { v[0] += cv[u]; v[1] += av[u]; }
        }
    }
    DST(1);
    if (p_lo >= 0 && want_gmax) {
        // gradient_max_norm = || x - Plus(x, -g) ||_inf (TrustRegionMinimizer::EvaluateGradientAndJacobian)
        double d[6], o[7];
#pragma unroll
        for (int k = 0; k < 6; k++) d[k] = -pg[k];
        pose_plus(xp, d, o);
#pragma unroll
        for (int k = 0; k < 7; k++) { double q = fabs(xp[k] - o[k]); v[5] = q > v[5] ? q : v[5]; }
    }
    DST(2);
#pragma unroll
    for (int u = 0; u < DU; u++) {
        const int i = tid + u * CTL_NT;
        if (i < n) {
            double gi = gv[u], yi = yv[u];
            double dc = damp_diag(O, dgv[u], B.jsc + W.loc_base + i, false);      // (the damping diagonal: with Jacobi scaling, LM's effective one)
            double ir = rsqrt_nr(dc);                     // 1 / sqrt(d): no IEEE sqrt / division expansions in these loops
            double gs = gi * ir;
            v[2] += gs * gs; v[3] += dc * yi * yi; v[4] += -gi * yi;
            if (want_gmax && lxv[u] >= 0) { double q = fabs(gi); v[5] = q > v[5] ? q : v[5]; }       // (Plus of a vector block is x + delta)
        }
    }
    for (int i = tid + DU * CTL_NT; i < n; i += CTL_NT) {       // windows of more than DU * 256 local dimensions
        double gi = g[i], yi = y[i];
        double dc = damp_diag(O, dg[i], B.jsc + W.loc_base + i, false);
        double ir = rsqrt_nr(dc);
        double gs = gi * ir;
        v[2] += gs * gs; v[3] += dc * yi * yi; v[4] += -gi * yi;
        if (want_gmax && l2x[i] >= 0) { double q = fabs(gi); v[5] = q > v[5] ? q : v[5]; }
    }
    DST(3);
    block_reduce<5, 1>(v, red);
    DST(4);
    if (xcl) {
#pragma unroll
        for (int u = 0; u < DU; u++) xv[u] = lxv[u] >= 0 ? xcl[lxv[u] - xb] : 0.0;
    }
    if (fresh) {
        x_cost = (win_part_sum(B.p_cpart, W.fsb0, W.fsb1, pst) + win_part_sum(B.pr_cpart, W.pch0, W.pch1, pst + 2 * WP_LDS)) + v[0];
        if (!t.lin_fail) { jg_sq = (win_part_sum(B.p_apart, W.lmb0, W.lmb1, pst + WP_LDS) + win_part_sum(B.pr_apart, W.pch0, W.pch1, pst + 3 * WP_LDS)) + v[1]; gmax = v[5]; }
    }
    const double gsq = v[2], ynn = v[3], gdot = v[4];
    // the bookkeeping of FinalizeIterationAndCheckIfMinimizerCanContinue, by every thread on its copy of the state (uniform)
    int go = 0;
    {
        int it = t.iter;
        if (fresh) {
            t.x_cost = x_cost; t.gmax = gmax; t.jg_sq = jg_sq;
            if (wr) {
                if (it == 0) { tr[0].cost = x_cost; tr[0].trust_region_radius = t.radius; tr[0].step_is_valid = 1; tr[0].step_is_successful = 1; }
                else tr[it].cost = x_cost;
                tr[it].gradient_max_norm = gmax;
            }
            if (it == 0) t.initial_cost = x_cost;
            t.need_lin = 0;
        }
        if (it >= O.max_iter) t.status = SWF_NO_CONVERGENCE;
        else if (gmax <= O.gtol && !t.lin_fail) t.status = SWF_CONVERGED_GRADIENT;
        else if (t.radius < O.min_r) t.status = SWF_RADIUS_TOO_SMALL;
        else {
            it = ++t.iter;
            swf_iteration& rec = tr[it < B.max_iter_trace ? it : B.max_iter_trace - 1];
            if (wr) rec.gradient_max_norm = gmax;
            if (t.lin_fail) {
                // Gauss-Newton solve failed: DoglegStrategy raises mu; HandleInvalidStep
                if (wr) { rec.step_is_valid = 0; rec.cost = t.x_cost; rec.trust_region_radius = t.radius; }
                t.lin_fail = 0; t.chol_fail = 0; t.eval_cand = 0;
                t.reuse = 0; t.need_lin = 1;
                if (O.strategy == SWF_LEVENBERG_MARQUARDT) {
                    // LevenbergMarquardtStrategy::StepIsInvalid = StepRejected(0)
                    if (++t.invalid_run >= 5) t.status = SWF_LINEAR_SOLVER_FAILURE;
                    else { t.radius /= t.lm_dec; t.lm_dec *= 2.0; t.mu = 1.0 / t.radius; if (wr) rec.trust_region_radius = t.radius; }
                } else {
                    t.mu *= O.mu_inc;
                    if (++t.invalid_run >= 5 || t.mu >= O.max_mu) t.status = SWF_LINEAR_SOLVER_FAILURE;
                }
            } else {
                if (!t.reuse) t.alpha = gsq / jg_sq;
                t.reuse = 1;
                go = 1;
            }
        }
    }
    DST(5);
    if (!go) { if (wr) *sout = t; return 0; }
    // ComputeTraditionalDoglegStep in the scaled space, un-scaled on the fly
    double gnorm = sqrt(gsq), gnn = sqrt(ynn), alpha = t.alpha, radius = t.radius;
    const double mu_s = t.mu;
    int mode; double c1 = 0, c2 = 0;          // step = c1 * g / dclamp + c2 * y
    double dnorm;
    if (O.strategy == SWF_LEVENBERG_MARQUARDT) { mode = 0; c1 = 0; c2 = -1.0; dnorm = gnn; }      // the damped step itself
    else if (gnn <= radius) { mode = 0; c1 = 0; c2 = -1.0; dnorm = gnn; }
    else if (gnorm * alpha >= radius) { mode = 1; c1 = -(radius / gnorm); c2 = 0; dnorm = radius; }
    else {
        double b_dot_a = -alpha * gdot;
        double a_sq = pow(alpha * gnorm, 2.0);
        double bma_sq = a_sq - 2 * b_dot_a + pow(gnn, 2);
        double cc = b_dot_a - a_sq;
        double dd = sqrt(cc * cc + bma_sq * (pow(radius, 2.0) - a_sq));
        double beta = (cc <= 0) ? (dd - cc) / bma_sq : (radius * radius - a_sq) / (dd + cc);
        mode = 2; c1 = -alpha * (1.0 - beta); c2 = -beta; dnorm = -1.0;
    }
    // Model cost change of the step s = c1 v + c2 y (v = D^-2 g, the Cauchy direction; y = the damped solution) WITHOUT touching a
    // Jacobian: ceres evaluates -(J s).(r + J s / 2) = -(g.s + s^T H s / 2) with H = J^T J; here
    //   v^T H v = |J v|^2 = jg_sq   (formed once per linearisation, with the back-substitution pass),
    //   H y = g - mu D^2 y          (y solves the damped system)  =>  v^T H y = |g / D|^2 - mu g.y,   y^T H y = g.y - mu |D y|^2,
    // and g.v = |g / D|^2: every term is one of the sums of the pass above.  (The identity holds to the backward error of the linear
    // solve, 1e-12 of the terms; round 1 re-read every Jacobian for this — 73 + 35 us per iteration of the cfg4 batch.)
    // An invalid step — model cost change not positive, TrustRegionMinimizer::HandleInvalidStep — is handled HERE, before any candidate
    // exists, so that no kernel behind this one sees a candidate of such a step: the speculative flow keeps the Jacobians of x in place
    // for the re-linearisation that follows.
    double mcc;
    {
        const double gy = -gdot, mu = mu_s;
        const double sHs = c1 * c1 * jg_sq + 2.0 * c1 * c2 * (gsq - mu * gy) + c2 * c2 * (gy - mu * ynn);
        mcc = -(c1 * gsq + c2 * gy + 0.5 * sHs);
    }
    t.model_cost_change = mcc;
    if (!(mcc > 0.0)) {
        swf_iteration& rec = tr[t.iter < B.max_iter_trace ? t.iter : B.max_iter_trace - 1];
        t.eval_cand = 0;
        if (wr) { rec.model_cost_change = mcc; rec.step_is_valid = 0; rec.cost = t.x_cost; rec.trust_region_radius = t.radius; }
        t.reuse = 0; t.need_lin = 1;
        if (++t.invalid_run >= 5) t.status = SWF_LINEAR_SOLVER_FAILURE;
        else if (O.strategy == SWF_LEVENBERG_MARQUARDT) { t.radius /= t.lm_dec; t.lm_dec *= 2.0; t.mu = 1.0 / t.radius; if (wr) rec.trust_region_radius = t.radius; }
        else t.mu *= O.mu_inc;
        if (wr) *sout = t;
        return 0;
    }
    // the step and the candidate = Plus(x, step) in one pass over the local dimensions, from the registers of the pass above; the pose
    // threads form their six step entries from the same operands (the same bits as step[]) and apply PoseLocalParameterization::Plus
    double a_s = 0, a_n = 0;
#pragma unroll
    for (int u = 0; u < DU; u++) {
        const int i = tid + u * CTL_NT;
        if (i < n) {
            double dc = clampd(dgv[u], O.min_diag, O.max_diag);
            double ir = rsqrt_nr(dc);
            // scaled step c1 g / sqrt(d) + c2 sqrt(d) y, then un-scaled (/ sqrt(d))
            double sc = c1 * (gv[u] * ir) + c2 * (dc * ir * yv[u]);
            a_s += sc * sc;
            const double st = sc * ir;
            if (lead) step[i] = st;
            if (lxv[u] >= 0) { const double x0 = xv[u], xn = x0 + st; if (lead) B.xc[lxv[u]] = xn; if (xcl) xcl[lxv[u] - xb] = xn; const double dv = x0 - xn; a_n += dv * dv; }
        }
    }
    for (int i = tid + DU * CTL_NT; i < n; i += CTL_NT) {
        const int a = l2x[i];
        double dc = clampd(dg[i], O.min_diag, O.max_diag);
        double ir = rsqrt_nr(dc);
        double sc = c1 * (g[i] * ir) + c2 * (dc * ir * y[i]);
        a_s += sc * sc;
        const double st = sc * ir;
        if (lead) step[i] = st;
        if (a >= 0) { const double x0 = B.x[a], xn = x0 + st; if (lead) B.xc[a] = xn; if (xcl) xcl[a - xb] = xn; const double dv = x0 - xn; a_n += dv * dv; }
    }
    DST(6);
    if (p_lo >= 0) {
        double d[6], o[7];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double dc = clampd(pd[k], O.min_diag, O.max_diag);
            double ir = rsqrt_nr(dc);
            double sc = c1 * (pg[k] * ir) + c2 * (dc * ir * py[k]);
            d[k] = sc * ir;
        }
        pose_plus(xp, d, o);
#pragma unroll
        for (int k = 0; k < 7; k++) { if (lead) B.xc[p_xo + k] = o[k]; if (xcl) xcl[p_xo + k - xb] = o[k]; const double dv = xp[k] - o[k]; a_n += dv * dv; }
    }
    DST(7);
    if (lead) {           // (the step's norms go into the state only: the other workgroups of a fused grid leave them out)
        double r2[2] = { a_s, a_n };
        block_reduce<2, 0>(r2, red);
        if (mode == 2) dnorm = sqrt(r2[0]);
        t.dogleg_step_norm = dnorm; t.step_norm = sqrt(r2[1]); t.eval_cand = 1;
        if (wr) *sout = t;
    }
    DST(8);
    return 1;
}
template <int DU, int GU>
__global__ void __launch_bounds__(CTL_NT) k_dogleg(DevBatch B, DevOpt O) {
    __shared__ double red[16 * 6];
    __shared__ double pst[4 * WP_LDS];
    WinState* s = B.ws + blockIdx.x;
    (void)d_dogleg<DU, GU>(B, O, (int)blockIdx.x, B.win[blockIdx.x], s, s, true, nullptr, red, pst);
}

// Latency path of ONE window in the speculative dogleg flow: k_dogleg and the Jacobian evaluation at its candidate as ONE grid.
// Every workgroup of k_eval_ps's grid (frame-sum blocks of projection observations | scalar factors | priors | IMU factors) first forms
// the dogleg step for itself — d_dogleg: the same loads, sums and bits in every workgroup; workgroup 0 alone writes the state (into
// ws_out: the others are still reading B.ws), the trace, the step and the candidate — keeps the candidate in LDS and evaluates its
// factors there.  What it replaces: a launch of one workgroup (12.8 us for a cfg3 window: dispatch, two dependent round trips, a
// reduction, the candidate written to HBM) in front of a launch that began by reading that candidate back.
#define XCL_MAX 4096            // ambient coordinates of a window the fused form takes (32 KB of LDS)
// JAC = false: the two-pass flows (Levenberg-Marquardt; windows with composite factors, whose re-linearisation a rejected candidate would
// have to undo): the same grid with the COST-ONLY evaluation of the candidate behind the step (k_dogleg + k_post_dogleg in one launch).
template <bool IMU, bool JAC = true>
__global__ void __launch_bounds__(256) k_step_eval(DevBatch B, DevOpt O, Segs S, WinState* ws_out, WinRec W, WinState* ws_clr) {
    constexpr int SM_PRIOR = 2 * PRIOR_LDS_DIM + 16, SM_FS = JAC ? FS_BLK * FS_HALF + 168 / 2 + 1 : 1;
    __shared__ double sm[SM_PRIOR > SM_FS ? SM_PRIOR : SM_FS];
    __shared__ double red[16 * 6];
    __shared__ double pst[4 * WP_LDS];
    __shared__ double xcl[XCL_MAX];
    const int bid = blockIdx.x;
    static_assert(XCL_MAX == 16 * CTL_NT, "d_dogleg stages x through XU = 16 registers per thread");
    // (ws_clr: the buffer the elimination grid behind k_decide will leave the state in — k_decide_lm_clique; nothing reads it during this
    // grid, and its failure flags have to be down before that grid's workgroups may raise them)
    if (bid == 0 && threadIdx.x == 0 && ws_clr) { ws_clr->lin_fail = 0; ws_clr->chol_fail = 0; }
    const int go = d_dogleg<16, 4, 16>(B, O, 0, W, B.ws, ws_out, bid == 0, xcl, red, pst);
    __syncthreads();
    if (!go) return;
    DevBatch E = B;
    E.xc = xcl - W.x_base; E.spec = 2;                  // evaluate at the LDS candidate, ungated (the state this grid reads still says "no candidate")
    if (bid < S.e[0]) { if (JAC) d_eval_proj_fs(E, bid, (double (*)[FS_HALF])sm, (int*)(sm + FS_BLK * FS_HALF)); else d_eval_proj_cost(E, bid); }
    else if (bid < S.e[1]) d_eval_scalar<JAC>(E, bid - S.e[0]);
    else if (!IMU || bid < S.e[2]) d_eval_prior<JAC>(E, bid - S.e[1], sm);
    else d_eval_imu<JAC>(E, bid - S.e[2]);
}

// acceptance test + trust-region update (TrustRegionMinimizer::Minimize loop body,
// DoglegStrategy::StepAccepted / StepRejected / StepIsInvalid)
// every field of the window's state but the failure flags (which the kernels of a linearisation set on their own: see k_decide_lm_clique)
__device__ __forceinline__ void ws_store_nofail(WinState* d, const WinState& t) {
    d->radius = t.radius; d->mu = t.mu; d->x_cost = t.x_cost; d->x_norm = t.x_norm; d->alpha = t.alpha; d->dogleg_step_norm = t.dogleg_step_norm;
    d->step_norm = t.step_norm; d->gmax = t.gmax; d->jg_sq = t.jg_sq; d->initial_cost = t.initial_cost; d->lm_dec = t.lm_dec;
    d->model_cost_change = t.model_cost_change; d->status = t.status; d->iter = t.iter; d->need_lin = t.need_lin; d->reuse = t.reuse;
    d->eval_cand = t.eval_cand; d->invalid_run = t.invalid_run; d->nsucc = t.nsucc; d->nunsucc = t.nunsucc;
}
// The body of k_decide as a device function (the latency path runs it at the head of the elimination grid: k_decide_lm_clique).  Threads
// [0, CTL_NT) of the workgroup carry the sums (the same strided order whatever the workgroup's size: the other waves add zeros); EVERY
// thread takes the decision on its copy t of the state (uniform values, same bits); the LEAD workgroup alone writes: the trace, an
// accepted candidate into x, and the new state into sout (every field but the failure flags; sout == sin for the in-place launch).
// t: the state after the decision (t.x_norm is refreshed in the lead only: nothing in the same grid reads it).
__device__ __forceinline__ void d_decide(const DevBatch& B, const DevOpt& O, const int w, const WinState* sin, WinState* sout, const bool lead,
                                         double* red /* 32 */, double* pst /* 2 WP_LDS */, WinState& t) {
    const int tid = threadIdx.x;
    t = *sin;
    const bool wr = lead && tid == 0;
    if (t.status != SWF_RUNNING || !t.eval_cand) { if (wr && sout != sin) ws_store_nofail(sout, t); return; }
    const WinRec& W = B.win[w];
    double ca[2] = { 0.0, 0.0 };
    // every load up front, as in k_dogleg: the candidate costs, and — speculatively, in the lead — the candidate itself with the flags of
    // its coordinates (the first XU strided coordinates per thread: a cfg3 window's 1447 are all of them), so that an accepted step is
    // stored from registers
    constexpr int XU = 8;
    const bool st_ = tid < CTL_NT;                            // the threads of the strided sums
    double cvg[4], xcv[XU]; unsigned char xfl[XU];
    if (st_) { win_part_stage(B.p_cpart, W.fsb0, W.fsb1, pst); win_part_stage(B.pr_cpart, W.pch0, W.pch1, pst + WP_LDS); }      // (staged by CTL_NT threads: blockDim may be larger)
#pragma unroll
    for (int u = 0; u < 4; u++) { int i = W.gf0 + tid + u * CTL_NT; cvg[u] = (st_ && i < W.gf1) ? B.g_cost[i] : 0.0; }
#pragma unroll
    for (int u = 0; u < XU; u++) { int i = W.x_base + tid + u * CTL_NT; bool ok = lead && st_ && i < W.x_base + W.x_n; xcv[u] = ok ? B.xc[i] : 0.0; xfl[u] = ok ? B.x_var[i] : (unsigned char)0; }
    if (st_) {
        // candidate cost: the formula of win_cost_sum (the generic factors' costs in index order per thread; the blocks' values behind the reduction)
#pragma unroll
        for (int u = 0; u < 4; u++) ca[0] += cvg[u];
        for (int base = W.gf0 + tid + 4 * CTL_NT; base < W.gf1; base += 4 * CTL_NT) {
            double cv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { int i = base + u * CTL_NT; cv[u] = i < W.gf1 ? B.g_cost[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) ca[0] += cv[u];
        }
    }
    block_reduce<2, 0>(ca, red);
    double cand = (win_part_sum(B.p_cpart, W.fsb0, W.fsb1, pst) + win_part_sum(B.pr_cpart, W.pch0, W.pch1, pst + WP_LDS)) + ca[0];
    const double model_cost_change = t.model_cost_change;
    if (!(cand == cand) || cand > 1.7976931348623157e308) cand = 1.7976931348623157e308;
    int accept = 0;
    {
        swf_iteration* tr = B.trace + (size_t)w * B.max_iter_trace;
        swf_iteration& rec = tr[t.iter < B.max_iter_trace ? t.iter : B.max_iter_trace - 1];
        swf_iteration r_{};                                   // the record's fields, written by the lead's thread 0 below
        t.eval_cand = 0;
        int fld = 0;                                          // bit 0: step_norm / cost_change, bit 1: relative_decrease / step_is_successful
        if (!(model_cost_change > 0.0)) {
            r_.step_is_valid = 0; r_.cost = t.x_cost; r_.trust_region_radius = t.radius;
            t.reuse = 0; t.need_lin = 1;
            if (++t.invalid_run >= 5) t.status = SWF_LINEAR_SOLVER_FAILURE;
            else if (O.strategy == SWF_LEVENBERG_MARQUARDT) { t.radius /= t.lm_dec; t.lm_dec *= 2.0; t.mu = 1.0 / t.radius; r_.trust_region_radius = t.radius; }
            else t.mu *= O.mu_inc;
        } else {
            r_.step_is_valid = 1; t.invalid_run = 0;
            r_.step_norm = t.step_norm;
            r_.cost_change = t.x_cost - cand;
            fld = 1;
            if (t.step_norm <= O.ptol * (t.x_norm + O.ptol)) {
                r_.cost = t.x_cost; r_.trust_region_radius = t.radius; t.status = SWF_CONVERGED_PARAMETER;
            } else if (fabs(r_.cost_change) <= O.ftol * t.x_cost) {
                r_.cost = t.x_cost; r_.trust_region_radius = t.radius; t.status = SWF_CONVERGED_FUNCTION;
            } else {
                fld = 3;
                r_.relative_decrease = r_.cost_change / model_cost_change;
                if (r_.relative_decrease > O.min_rel_dec) {
                    accept = 1;
                    r_.step_is_successful = 1; r_.cost = cand; t.x_cost = cand; t.nsucc++;
                    if (O.strategy == SWF_LEVENBERG_MARQUARDT) {
                        // LevenbergMarquardtStrategy::StepAccepted
                        double q = 2.0 * r_.relative_decrease - 1.0, f = 1.0 - q * q * q;
                        t.radius = t.radius / (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
                        t.radius = t.radius < O.max_r ? t.radius : O.max_r;
                        t.lm_dec = 2.0; t.mu = 1.0 / t.radius;
                    } else {
                        if (r_.relative_decrease < 0.25) t.radius *= 0.5;
                        if (r_.relative_decrease > 0.75) t.radius = t.radius > 3.0 * t.dogleg_step_norm ? t.radius : 3.0 * t.dogleg_step_norm;
                        double m2 = 2.0 * t.mu / O.mu_inc;
                        t.mu = O.min_mu > m2 ? O.min_mu : m2;
                    }
                    t.reuse = 0; t.need_lin = 1;
                } else {
                    r_.step_is_successful = 0; r_.cost = t.x_cost; t.nunsucc++;
                    if (O.strategy == SWF_LEVENBERG_MARQUARDT) {
                        // LevenbergMarquardtStrategy::StepRejected: the same Jacobian is damped harder — the window re-linearises at
                        // the unchanged point (bit-identical Jacobians) so that the assembly picks up the new mu
                        t.radius /= t.lm_dec; t.lm_dec *= 2.0; t.mu = 1.0 / t.radius; t.reuse = 0; t.need_lin = 1;
                    } else { t.radius *= 0.5; t.reuse = 1; }
                }
                r_.trust_region_radius = t.radius;
            }
        }
        if (wr) {
            // (the fields k_decide has always written, no others: the rest of the row belongs to k_dogleg)
            rec.model_cost_change = model_cost_change;
            rec.step_is_valid = r_.step_is_valid; rec.cost = r_.cost; rec.trust_region_radius = r_.trust_region_radius;
            if (fld & 1) { rec.step_norm = r_.step_norm; rec.cost_change = r_.cost_change; }
            if (fld & 2) { rec.relative_decrease = r_.relative_decrease; rec.step_is_successful = r_.step_is_successful; }
        }
    }
    if (lead && accept) {
        // x <- candidate and || x || over the variable blocks in one pass (x_var: host-built flag per ambient coordinate)
        double a = 0;
        if (st_) {
#pragma unroll
            for (int u = 0; u < XU; u++) { int i = W.x_base + tid + u * CTL_NT; if (i < W.x_base + W.x_n) { B.x[i] = xcv[u]; if (xfl[u]) a += xcv[u] * xcv[u]; } }
            for (int i = W.x_base + tid + XU * CTL_NT; i < W.x_base + W.x_n; i += CTL_NT) { const double xv = B.xc[i]; B.x[i] = xv; if (B.x_var[i]) a += xv * xv; }
        }
        t.x_norm = sqrt(block_sum(a, red));
    }
    if (wr) ws_store_nofail(sout, t);
}
__global__ void __launch_bounds__(CTL_NT) k_decide(DevBatch B, DevOpt O) {
    __shared__ double red[16 * 2];
    __shared__ double pst[2 * WP_LDS];
    WinState* s = B.ws + blockIdx.x;
    WinState t;
    d_decide(B, O, (int)blockIdx.x, s, s, true, red, pst, t);
}

// Latency path of ONE window: k_decide at the head of the elimination grid (k_lm_clique).  Every workgroup takes the accept / reject
// decision for itself (d_decide: same loads, sums, bits), keeps the window's state AFTER the decision in LDS and runs its share of
// the landmark Schur product / its clique against that copy; workgroup (0, 0) alone commits — the trace, an accepted candidate into x,
// the new state into ws_out (a THIRD buffer: the other workgroups are still reading B.ws, and k_step_eval's lead cleared ws_out's
// failure flags two launches ago so that a workgroup of this grid may raise them without racing the lead's store, which leaves
// those two fields alone).  A rejected step (or a finished window) makes every workgroup return right behind the decision.
// What it replaces: a launch of one workgroup between two grids (5.9 us of a cfg3 window's iteration).
template <int NCW, int TPW, int TW, int LDR>
__global__ void __launch_bounds__(LS_NT(NCW, TW)) k_decide_lm_clique(DevBatch B, DevOpt O, int qpb, int lp, int kms, int s_direct, int n_parts, WinState* ws_out) {
    __shared__ double red[16 * 2];
    __shared__ double pst[2 * WP_LDS];
    __shared__ WinState s_loc;
    WinState t;
    const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
    d_decide(B, O, 0, B.ws, ws_out, lead, red, pst, t);
    if (threadIdx.x == 0) s_loc = t;
    __syncthreads();
    if (!t.need_lin || t.status != SWF_RUNNING) return;       // (uniform over the grid)
    DevBatch E = B;
    E.ws = &s_loc;                                             // window 0's state = the copy behind the decision (reads of need_lin / mu / iter, a raised lin_fail)
    if ((int)blockIdx.y < n_parts) d_lm_schur<NCW, TPW, TW, LDR, true>(E, O, qpb, lp, kms, s_direct, (int)blockIdx.x, (int)blockIdx.y);
    else {
        constexpr int CLQ_NW = 4;                              // (as in k_lm_clique: waves 4 .. 15 end here, the four survivors synchronise among themselves)
        if (threadIdx.x >= CLQ_NW * 64) return;
        d_clique_elim<64, 64, 9, 2, 8, 4, CLQ_NW>(E, O, ((int)blockIdx.y - n_parts) * (int)gridDim.x + (int)blockIdx.x);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_loc.lin_fail) ws_out->lin_fail = 1;
}

// after the last slot: fold the final linearisation (cost, gradient norm) into the trace
__global__ void __launch_bounds__(256) k_finalize(DevBatch B, DevOpt O, WinState* ws_primary) {
    __shared__ double red[16];
    __shared__ double pst[2 * WP_LDS];
    int w = blockIdx.x, tid = threadIdx.x;
    // (the latency path's fused step kernel leaves a window's state in the other of two buffers at every iteration: the solve ends in the primary one)
    if (ws_primary != B.ws) { if (tid == 0) ws_primary[w] = B.ws[w]; __syncthreads(); }
    WinState& s = ws_primary[w];
    const WinRec& W = B.win[w];
    swf_iteration* tr = B.trace + (size_t)w * B.max_iter_trace;
    if (s.need_lin && (s.status == SWF_RUNNING)) {
        double x_cost = win_cost_sum(B, W, red, pst);
        double gmax = win_gmax(B, W, red);
        if (tid == 0) {
            s.x_cost = x_cost; s.gmax = gmax; s.need_lin = 0;
            int it = s.iter < B.max_iter_trace ? s.iter : B.max_iter_trace - 1;
            tr[it].cost = x_cost; tr[it].gradient_max_norm = gmax;
            if (s.iter == 0) { s.initial_cost = x_cost; tr[0].trust_region_radius = s.radius; tr[0].step_is_valid = 1; tr[0].step_is_successful = 1; }
        }
    }
    __syncthreads();
    if (tid == 0 && s.status == SWF_RUNNING) {
        if (O.step_mode == SWF_ASSEMBLE_ELIMINATE_ONLY) s.status = s.lin_fail ? SWF_LINEAR_SOLVER_FAILURE : SWF_ASSEMBLED_ONLY;
        else if (s.gmax <= O.gtol) s.status = SWF_CONVERGED_GRADIENT;
        else s.status = SWF_NO_CONVERGENCE;
    }
}

// device-side state restore (bench: same inputs every step, already resident in HBM)
__global__ void k_copy(double* dst, const double* src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
