// swf_kernels3.h — marginalisation consumer: what the reference does with the export of an
// is_optimize = false solve,
//   SWFOptimization::UpdateSchur              R/swf/swf_gnss.cpp:25-61      A = S_nn - S_nm pinv(S_mm) S_mn, b likewise
//   MarginalizationInfo::setmarginalizeinfo   R/factor/marginalization_factor.cpp:449-488 (Sqrt = true)
//                                             J = sqrt(Lambda+) V^T,  r0 = Lambda+^-1/2 V^T b,  eigenvalues <= eps dropped
// re-designed around what the solve already left on the device.  With S = L L^T in elimination order (parameter_head
// last), the Schur complement onto the trailing n states IS L_nn L_nn^T, and b = A y_n with y = S^-1 rhs — no m x m
// eigen-decomposition (the reference's pseudo-inverse equals the inverse whenever S_mm is positive definite, which
// the successful Cholesky certifies; a failed factorisation is reported, not papered over).  The eigen square root of
// A = G G^T, G = L_nn, comes from a ONE-SIDED (Hestenes) Jacobi on the columns of G held in LDS: it never forms A
// for the iteration, works to high relative accuracy, and parallelises as n/2 independent column pairs per step;
// the orthogonalised columns are the rows of J (G W = U Sigma), no eigenvector matrix is carried along.
#pragma once
#include "swf_dev.h"

#define MG_MAXN 140                       // largest tail whose M is LDS-resident (153 KB)
#define MG_BIGN 640                       // largest tail of the eigen form (= the largest reduced system, CB_NMAX): above MG_MAXN, M lives in a per-window HBM / L2 scratch
#define MG_NT 1024                        // 64 sixteen-lane groups = 64 column pairs per step
#define MG_LDS_DOUBLES 19600              // 153 KB: M for n <= 140, M and V together for n <= 98
#define Mc(c, r) Mm[(c) * n + (r)]
// form: 0 = eigen square root (the reference's prior), 1 = Cholesky square root J = L_nn^T, r0 = L_nn^T y_n (same quadratic)
// ldn = leading dimension of the per-window output slabs (>= every window's tail dimension)
// GM = false: tails up to MG_MAXN, M in LDS (and every Cholesky-form request); GM = true: eigen form for MG_MAXN < n <= MG_BIGN with M
// in Mscr (same algorithm, same rotation order).  A batch launches both; each instantiation skips the other's windows.
// ---------------------------------------------------------------------------------------------------------------------
// Rank-deficient tails.  The reference pseudo-inverts only S_mm (UpdateSchur) and lets the eigen square root drop the null
// directions of A (setmarginalizeinfo): a marginal that is singular on the kept states — an unobservable camera extrinsic, a
// yaw nobody measured — is business as usual there.  The Cholesky of ALL of S fails on such a window (in its last n columns),
// and L_nn with it.  This kernel steps in for exactly those windows (WinState::chol_fail): a right-looking Cholesky of the
// first m columns only, in the window's (now free) L buffer, leaves the Schur complement A and the reduced right-hand side b
// in the trailing block; a diagonally pivoted outer-product Cholesky of A then yields rows v_r with sum_r v_r v_r^T = A up to
// the pivots it drops (rank-revealing: it stops at pivots below 1e-14 of the largest).  k_marginalize takes M = [v_r] and b
// from here instead of L_nn / y_n.
// A singular S_mm — a marginalised state nothing constrains, or a direction of the marginalised block only a combination of
// which is observed — is business as usual in the reference too: UpdateSchur PSEUDO-inverts S_mm (eigenvalues <= 1e-8 dropped,
// R/swf/swf_gnss.cpp:44-51; the same in MarginalizationInfo::marginalize, R/factor/marginalization_factor.cpp:260-377).  For a
// positive semi-definite S the generalised Schur complement does not depend on which generalised inverse is taken, and a
// Cholesky that SKIPS a numerically null pivot (the column below it is null as well) computes it: a pivot <= max(eps, 1e-13 of
// the block's largest diagonal entry) among the first m columns drops its column; a pivot below minus a thousand times that
// (an indefinite matrix, or a NaN) is a real failure (rank -1).
// One 1024-thread workgroup per window.  Round 4: both halves are BLOCKED (16 columns / pivots at a time) — round 3 applied every
// column and every pivot as a rank-one update of the whole trailing matrix in global memory (one CU streaming 1.5 MB 177 times, then
// 0.55 MB 263 times: 9.1 ms at cfg5 size; the path is not rare there: 6 of 128 marginalisation windows took it):
//   * the first m columns: the panel (rows below the block, 16 columns) is factored in LDS, null pivots dropping their columns as before,
//     and applied to the trailing matrix once, as a rank-16 update;
//   * the pivoted Cholesky of A is LAZY inside a block (LAPACK's dpstrf): the diagonal is kept up to date pivot by pivot (that is all the
//     pivot choice needs), the row of a chosen pivot is formed from the untouched trailing matrix minus the block's earlier rows (in
//     LDS), and the trailing matrix takes the block's 16 rows at once.
// Same pivots, same drops, same stopping rule; the sums run in a different order (tolerances of the parity tests unchanged).
// ---------------------------------------------------------------------------------------------------------------------
#define RS_NB 16
// Diagonally pivoted Cholesky of the n x n matrix A (full symmetric, ld = n; read only), 16 pivots at a time, by one 1024-thread
// workgroup: rows v_r of V (ld = n) with sum_r v_r v_r^T = A up to the pivots it drops (it stops at pivots below 1e-14 of the largest;
// the remaining rows are zero).  LEFT-LOOKING by blocks: a block starts from a POOL of candidates, the (up to) 24 largest entries of
// the running diagonal; their rows of the Schur complement, A[p, :] - sum_r v_r[:] v_r[p] over all rows so far, are formed in LDS in one
// pass over V; the block's (up to) 16 pivots are then taken from the pool in the order the running diagonal dictates (it is kept up to
// date pivot by pivot: that is all the choice needs), each pivot row = its pool row minus the block's earlier rows — LDS work behind
// ONE barrier, no trip to L2 per pivot, no trailing matrix to update (round 3: a rank-one update of the trailing matrix per pivot;
// an earlier form of this round: a lazily updated trailing matrix, one L2 round trip per pivot + 0.5 MB streamed per block).  With 24
// candidates for 16 places the pivots are those of full diagonal pivoting on the cfg5 prior (same rotation counts in the sweeps).
//   !INPLACE: V in global memory (rows) with VT its transpose (the pool pass reads v_r[p] for 24 fixed p and all r: contiguous in VT),
//             Rb = LDS pool rows (24 x n), stg = LDS staging of the pool's VT rows (24 x rc doubles, rc = rows of V per pass)
//   INPLACE:  V itself is LDS (the small tails): pool rows and pivot rows are built in its rows, everything is read from there
// dg = n doubles of LDS: the running diagonal, -1e300 for an eliminated (or dropped) index.
// Two callers: the rank-deficient tails (k_marg_rescue), and the PRECONDITIONER of the Jacobi sweeps (k_marg_pchol, k_marginalize):
// the one-sided Jacobi on the columns of a pivoted Cholesky factor (Veselic / Hari) converges in about half the sweeps the columns of
// the unpivoted L_nn need (cfg5's 263-dimension prior: 8 against 16).
#define RS_POOL 24                        // candidate rows held per block (16 of them at most become pivots)
#define RS_GONE (-1e300)
template <bool INPLACE>
// drop_abs: the stopping rule.  The rank-deficient tails pass 0: a pivot below 1e-14 of the largest ends the factorisation (rank-revealing).
// The preconditioner of a HEALTHY window (its Cholesky went through: A is numerically definite) passes eps / (16 n), eps = the caller's
// eigenvalue threshold (the reference's 1e-8, R/factor/marginalization_factor.cpp:463-470): there the relative cut alone would drop
// whole directions the reference keeps — with diag(A) ~ 1e10 it sits at 1e-4, far above eps — whereas a factorisation that ends at a
// pivot p leaves a remainder of trace <= (n - r) p, i.e. below eps / 16 in every direction: nothing the eps test would have kept is lost.
__device__ __forceinline__ void d_pivoted_chol(const double* A, double* V, double* VT, double* Rb, double* dg, double* stg, const int rc, const int n, const double drop_abs = 0.0) {
    __shared__ int cid[RS_POOL];                          // the block's candidates (indices)
    __shared__ double cdg[RS_POOL];                       // their running diagonal entries; -2 once taken, -1 for an empty slot
    __shared__ int ncand_s;
    __shared__ double d0_s;
    const int tid = threadIdx.x;
    __syncthreads();
    for (int e = tid; e < n * n; e += 1024) V[e] = 0.0;
    for (int i = tid; i < n; i += 1024) dg[i] = A[(size_t)i * n + i];
    if (tid == 0) d0_s = -1.0;
    __syncthreads();
    // thread (g, i): group 0 owns index i = tid (n <= 640) through the pivot steps; up to three groups share the passes over j (ranks)
    // and over the rows of V (pool rows), a third of the rows each, their partial sums added in group order (deterministic)
    const int ng = min(3, 1024 / n), g = tid / n, i = tid - g * n;
    const bool act = g == 0, gact = g < ng;
#ifdef SWF_PROFILE_CHOL
    unsigned long long pc_t[6] = {0, 0, 0, 0, 0, 0}, pc_l = __builtin_amdgcn_s_memtime();
#define PCACC(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pc_t[k] += t_ - pc_l; pc_l = t_; } while (0)
#else
#define PCACC(k)
#endif
    int r0 = 0;                                             // rows of V written
    bool stop = false;
    while (r0 < n && !stop) {
        double* R = INPLACE ? V + (size_t)r0 * n : Rb;        // (a compile-time choice: the pointer keeps its address space, LDS either way)
        // ---- the block's candidate pool: the largest entries of the running diagonal, by rank (ties: the smaller index first)
        if (tid < RS_POOL) { cid[tid] = -1; cdg[tid] = -1.0; }
        if (tid == 0) ncand_s = 0;
        __syncthreads();
        int myslot = -1;
        const int pool = INPLACE ? min(RS_POOL, n - r0) : RS_POOL;
        int* rk = INPLACE ? (int*)R : (int*)stg;            // n integers of LDS that are free right now
        if (act) rk[i] = 0;
        __syncthreads();
        if (gact && dg[i] > 0.5 * RS_GONE) {
            const double ki = dg[i];
            int rank = 0;
            const int j0 = g * n / ng, j1 = (g + 1) * n / ng;
#pragma unroll 8
            for (int j = j0; j < j1; j++) { const double kj = dg[j]; rank += (kj > ki) || (kj == ki && j < i); }
            atomicAdd(&rk[i], rank);
        }
        __syncthreads();
        if (act && dg[i] > 0.5 * RS_GONE) {
            const int rank = rk[i];
            if (rank < pool) { myslot = rank; cid[rank] = i; cdg[rank] = dg[i]; atomicAdd(&ncand_s, 1); }
        }
        __syncthreads();
        if (INPLACE && act && 2 * i < n + 1) R[i] = 0.0;      // (the rank counters sat in a row of V)
        PCACC(0);
        const int ncand = ncand_s;                          // (ranks 0 .. ncand - 1 are all present)
        if (ncand == 0) break;
        if (d0_s < 0.0) { __syncthreads(); if (tid == 0) d0_s = cdg[0]; __syncthreads(); }
        const double d0 = d0_s;
        const double thr = drop_abs > 0.0 ? fmin(1e-14 * d0, drop_abs) : 1e-14 * d0;
        // ---- the pool's rows of the Schur complement: thread (g, i) forms entry i of all of them over every ng-th row of V
        {
            double acc[RS_POOL];
#pragma unroll
            for (int c = 0; c < RS_POOL; c++) acc[c] = 0.0;
            if (INPLACE) {
                int pc[RS_POOL];
#pragma unroll
                for (int c = 0; c < RS_POOL; c++) pc[c] = c < ncand ? cid[c] : 0;
                if (gact) for (int r = g; r < r0; r += ng) {
                    const double* row = V + (size_t)r * n;
                    const double vi = row[i];
#pragma unroll
                    for (int c = 0; c < RS_POOL; c++) acc[c] -= vi * row[pc[c]];
                }
            } else {
                for (int rb = 0; rb < r0; rb += rc) {
                    const int nr_ = min(rc, r0 - rb);
                    // stg[rr][c] = v_(rb + rr)[cid[c]], read along the rows of VT
                    __syncthreads();
                    for (int e = tid; e < RS_POOL * nr_; e += 1024) { const int c = e / nr_, rr = e - c * nr_; stg[rr * RS_POOL + c] = c < ncand ? __hip_atomic_load(VT + (size_t)cid[c] * n + rb + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }      // (past this CU's L1: the lines of a row of VT fill up block by block, written by other waves)
                    __syncthreads();
                    if (gact) {
                        // (eight rows of V on their way at a time: a row is one L2 round trip)
#pragma unroll 8
                        for (int rr = g; rr < nr_; rr += ng) {
                            const double vi = V[(size_t)(rb + rr) * n + i];
                            const double* sp = stg + rr * RS_POOL;
#pragma unroll
                            for (int c = 0; c < RS_POOL; c++) acc[c] -= vi * sp[c];
                        }
                    }
                }
            }
            for (int gg = 0; gg < ng; gg++) {
                if (g == gg) {
#pragma unroll
                    for (int c = 0; c < RS_POOL; c++) if (c < ncand) R[(size_t)c * n + i] = (gg == 0 ? A[(size_t)cid[c] * n + i] : R[(size_t)c * n + i]) + acc[c];      // (A is symmetric: row p = column p, read along the row)
                }
                if (gg + 1 < ng) __syncthreads();
            }
        }
        __syncthreads();
        PCACC(1);
        unsigned long long o_lo = 0, o_hi = 0;              // slot taken at each step, a byte each (uniform)
        auto slot_of = [&](int t) { return (int)(((t < 8 ? o_lo : o_hi) >> (8 * (t & 7))) & 255ull); };
        int nbk = 0;
        for (int sblk = 0; sblk < RS_NB && sblk < ncand; sblk++) {
            // the largest running diagonal entry among the candidates not taken yet (every thread, redundantly: no broadcast step)
            // (every wave, redundantly — no broadcast step — but lane-parallel: most of the 16 waves have no index to work on, and what
            // they issue here competes with the waves that do.  Lane c holds candidate c; the first lane holding the maximum wins.)
            const int ln = tid & 63;
            const double cv = ln < RS_POOL ? cdg[ln] : -3.0;
            const double m16 = grp16_max(cv);
            const double bv = fmax(rows_lane(m16, 0), rows_lane(m16, 16));
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(cv == bv);
            int bc = hit ? __builtin_ctzll(hit) : -1;
            const int ok = (int)(bc >= 0 && bv > thr && bv > 0.0);      // uniform
            if (!ok) { if (sblk == 0) stop = true; break; }      // numerically zero; the largest of all: rank reached
            const int p = __builtin_amdgcn_readfirstlane(cid[bc]); const double isq = rsqrt_nr(bv);
            // row of the pivot: its pool row minus the block's earlier rows (all requests before the first use); zero at eliminated indices
            if (act) {
                double v = R[(size_t)bc * n + i];
                double xi[RS_NB], xp[RS_NB];
#pragma unroll
                for (int t = 0; t < RS_NB; t++) { const double* row = R + (size_t)(t < sblk ? slot_of(t) : bc) * n; xi[t] = row[i]; xp[t] = row[p]; }
#pragma unroll
                for (int t = 0; t < RS_NB; t++) v -= t < sblk ? xi[t] * xp[t] : 0.0;
                const double di = dg[i];
                const bool gone = !(di > 0.5 * RS_GONE);
                v = gone ? 0.0 : v * isq;
                R[(size_t)bc * n + i] = v;
                const double d = di - v * v;
                if (i == p) { dg[i] = RS_GONE; cdg[bc] = -2.0; }
                else if (!gone) { dg[i] = d; if (myslot >= 0) cdg[myslot] = d; }
            }
            if (sblk < 8) o_lo |= (unsigned long long)bc << (8 * sblk); else o_hi |= (unsigned long long)bc << (8 * (sblk - 8));
            nbk = sblk + 1;
            __syncthreads();
        }
        PCACC(2);
        // pool members that are numerically null by now (a running diagonal only shrinks) are dropped for good
        if (act && myslot >= 0) { const double di = dg[i]; if (di > 0.5 * RS_GONE && !(di > thr)) dg[i] = RS_GONE; }
        if (INPLACE) {
            // the block's rows to the front of the pool's slots, in pivot order (row swaps inside LDS: thread i moves column i)
            for (int t = 0; t < nbk; t++) {
                const int sl = slot_of(t);                      // uniform
                if (sl != t) {
                    if (act) { const double x = R[(size_t)t * n + i]; R[(size_t)t * n + i] = R[(size_t)sl * n + i]; R[(size_t)sl * n + i] = x; }
                    // whatever pivot row sat in slot t now sits in slot sl
                    for (int u = t + 1; u < nbk; u++) if (slot_of(u) == t) {
                        if (u < 8) o_lo = (o_lo & ~(255ull << (8 * u))) | ((unsigned long long)sl << (8 * u));
                        else o_hi = (o_hi & ~(255ull << (8 * (u - 8)))) | ((unsigned long long)sl << (8 * (u - 8)));
                    }
                    if (t < 8) o_lo = (o_lo & ~(255ull << (8 * t))) | ((unsigned long long)t << (8 * t));
                    else o_hi = (o_hi & ~(255ull << (8 * (t - 8)))) | ((unsigned long long)t << (8 * (t - 8)));
                }
            }
            __syncthreads();
            for (int e = tid; e < (ncand - nbk) * n; e += 1024) R[(size_t)nbk * n + e] = 0.0;      // the other slots back to zero
        } else if (act) {
            // the block's rows to V and, transposed, to VT (thread i: 16 consecutive entries of its row of VT)
            for (int t = 0; t < nbk; t++) { const double v = R[(size_t)slot_of(t) * n + i]; V[(size_t)(r0 + t) * n + i] = v; VT[(size_t)i * n + r0 + t] = v; }
        }
        r0 += nbk;
        if (nbk == 0) stop = true;
        __threadfence_block();
        __syncthreads();
        PCACC(3);
    }
    __syncthreads();
#ifdef SWF_PROFILE_CHOL
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 5; k++) g_chol_stamps[40 + k] = pc_t[k];
#endif
}

__global__ void __launch_bounds__(1024) k_marg_rescue(DevBatch B, const int* tail_dim, int ldn, double* resM, double* resb, int* res_ok, int force, double eps, double* vt_scr, int rc) {
    extern __shared__ double rs_lds[];                  // panel: (nr + 1) x 16 | later: block rows 16 x n, diagonal n, flags n
    __shared__ double red_v[16];
    __shared__ double piv_s; __shared__ int bad_s;
    int w = blockIdx.x, tid = threadIdx.x;
    const WinRec& W = B.win[w];
    const WinState& s = B.ws[w];
    if (tid == 0) res_ok[w] = 0;
    if (!s.chol_fail && !(force && !s.lin_fail)) return;      // force: testing aid (SWF_FORCE_MARG_RESCUE), every healthy window takes this path
    const int n = tail_dim[w], nr = W.n_red, m = nr - n;
    if (n <= 0 || n > ldn || m < 0) return;
    double* Wk = B.L + W.Lt_base;                       // (nr + 1) rows x nr columns, row-major; row nr carries the right-hand side
    const double* S = B.S + W.S_base;
    for (size_t e = tid; e < (size_t)(nr + 1) * nr; e += 1024) Wk[e] = S[e];
    // scale of the marginalised block: its largest diagonal entry
    {
        double dm = 0.0;
        for (int j = tid; j < m; j += 1024) dm = fmax(dm, S[(size_t)j * nr + j]);
        dm = wave_max(dm);
        if ((tid & 63) == 0) red_v[tid >> 6] = dm;
        if (tid == 0) bad_s = 0;
        __syncthreads();
        if (tid == 0) { double v = red_v[0]; for (int q = 1; q < 16; q++) v = fmax(v, red_v[q]); piv_s = v; }
    }
    __syncthreads();
    const double tol = fmax(eps, 1e-13 * piv_s);
    __syncthreads();
    // ---- the first m columns, 16 at a time.  Panel P[r][c] = Wk[jb + r][jb + c], r = 0 .. nr - jb (the right-hand-side row included)
    double (*P)[RS_NB + 1] = (double (*)[RS_NB + 1])rs_lds;
    for (int jb = 0; jb < m; jb += RS_NB) {
        const int wd = min(RS_NB, m - jb), rows = nr + 1 - jb;
        // (columns beyond a short last block are zero: the rank-16 update below runs over all sixteen)
        for (int e = tid; e < rows * RS_NB; e += 1024) { int r = e / RS_NB, c = e - r * RS_NB; P[r][c] = (c < wd && (r >= c || jb + r == nr)) ? Wk[(size_t)(jb + r) * nr + jb + c] : 0.0; }
        __syncthreads();
        for (int c = 0; c < wd; c++) {
            const double piv = P[c][c];                    // uniform
            if (!(piv > tol)) {
                if (!(piv >= -1e3 * tol)) { if (tid == 0) bad_s = 1; }      // indefinite (or NaN): a real failure
                // a null direction of S_mm: the pseudo-inverse drops it (its column takes no part in any update)
                __syncthreads();
                for (int r = tid; r < rows; r += 1024) P[r][c] = 0.0;
                __syncthreads();
                continue;
            }
            const double id = 1.0 / sqrt(piv);
            __syncthreads();
            for (int r = c + tid; r < rows; r += 1024) P[r][c] = r == c ? sqrt(piv) : P[r][c] * id;
            __syncthreads();
            // the rest of the panel: columns c' in (c, wd), rows r >= c'
            const int rem = wd - 1 - c;
            for (int e = tid; e < rem * rows; e += 1024) {
                int cc = c + 1 + e / rows, r = e - (e / rows) * rows;
                if (r >= cc) P[r][cc] -= P[r][c] * P[cc][c];
            }
            __syncthreads();
        }
        if (bad_s) return;
        // trailing update (lower triangle and the right-hand-side row): Wk[i][k] -= sum_c P[i][c] P[k][c], i in [jb + wd, nr], k in [jb + wd, min(i, nr - 1)]
        {
            const int t0 = jb + wd, nt = nr - t0;          // trailing columns t0 .. nr - 1 (nt of them), rows t0 .. nr
            // a thread takes one row and every 32nd group of four columns: its row of the panel stays in registers
            const int kg = (nt + 3) >> 2;                   // groups of four columns
            for (int i = t0 + (tid >> 5); i <= nr; i += 32) {
                double pi[RS_NB];
#pragma unroll
                for (int c = 0; c < RS_NB; c++) pi[c] = c < wd ? P[i - jb][c] : 0.0;
                for (int g = tid & 31; g < kg; g += 32) {
                    const int k0 = t0 + 4 * g;
                    if (k0 > i) break;
                    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
                    for (int c = 0; c < RS_NB; c++) {
                        const double x = pi[c];
                        a0 += x * P[k0 - jb][c];
                        a1 += x * P[min(k0 + 1, nr) - jb][c];
                        a2 += x * P[min(k0 + 2, nr) - jb][c];
                        a3 += x * P[min(k0 + 3, nr) - jb][c];
                    }
                    double* row = Wk + (size_t)i * nr;
                    if (k0 <= i && k0 < nr) row[k0] -= a0;
                    if (k0 + 1 <= i && k0 + 1 < nr) row[k0 + 1] -= a1;
                    if (k0 + 2 <= i && k0 + 2 < nr) row[k0 + 2] -= a2;
                    if (k0 + 3 <= i && k0 + 3 < nr) row[k0 + 3] -= a3;
                }
            }
        }
        __syncthreads();
    }
    // A = trailing n x n block (lower), b = right-hand-side row; mirror A into the scratch as a full symmetric matrix
    double* M = resM + (size_t)w * ldn * ldn;           // first the working copy of A (full symmetric, ld = n), finally the rows v_r
    for (int e = tid; e < n * n; e += 1024) { int i = e / n, k = e - i * n; M[e] = i >= k ? Wk[(size_t)(m + i) * nr + m + k] : Wk[(size_t)(m + k) * nr + m + i]; }
    for (int i = tid; i < n; i += 1024) resb[(size_t)w * ldn + i] = Wk[(size_t)nr * nr + m + i];
    __syncthreads();
    // ---- pivoted Cholesky of A, 16 pivots at a time.  V = Wk (n x n, ld = n; Wk has (nr + 1) * nr >= n * n doubles): row r of the result.
    // LDS: the pool's rows 24 x n | dg[n] the running diagonal | 24 x rc staging; vt_scr = the window's (still unused) J slab for V^T
    double* V = Wk;
    d_pivoted_chol<false>(M, V, vt_scr + (size_t)w * ldn * ldn, rs_lds, rs_lds + (size_t)RS_POOL * n, rs_lds + (size_t)(RS_POOL + 1) * n, rc, n);
    __syncthreads();
    for (int e = tid; e < n * n; e += 1024) M[e] = V[e];
    if (tid == 0) res_ok[w] = 1;
}

template <bool GM>
__global__ void __launch_bounds__(MG_NT) k_marginalize(DevBatch B, const int* tail_dim, double eps, int form, int ldn,
                                                        double* outA, double* outb, double* outJ, double* outr0,
                                                        double* outw, int* outrank, double* Mscr,
                                                        const double* resM, const double* resb, const int* res_ok, int force,
                                                        int phase, int* bj_ok) {
    // phase 0: everything (GM = false).  GM = true runs as phase 1 (set-up: M, A, b, V = I), the block-Jacobi sweeps of k_marg_bj
    // over many workgroups, then phase 2 (eigenvalues, square root, sorted write-out).
    __shared__ double lds[GM ? 16 : MG_LDS_DOUBLES];  // M (n x n, column c contiguous: row c of L_nn) | V (n x n) if both fit
    __shared__ double lam[GM ? MG_BIGN : MG_MAXN + 4];
    __shared__ double bv[GM ? MG_BIGN : MG_MAXN + 4];
    __shared__ double pc_dg[GM ? 1 : MG_MAXN + 4];      // d_pivoted_chol's running diagonal
    __shared__ int nrot;
    __shared__ unsigned long long crit_sh[2];
    int w = blockIdx.x, tid = threadIdx.x;
    const WinRec& W = B.win[w];
    const WinState& s = B.ws[w];
    int n = tail_dim[w], nr = W.n_red, m = nr - n;
    size_t o2 = (size_t)w * ldn * ldn, o1 = (size_t)w * ldn;
    if (GM != (form == 0 && n > MG_MAXN)) return;     // the other instantiation's window
    const bool rescued = ((s.lin_fail && s.chol_fail) || (force && !s.lin_fail)) && form == 0 && res_ok[w];       // k_marg_rescue supplied M (M^T M = A) and b
    if (GM && phase == 1 && tid == 0) bj_ok[w] = 0;
    if (n <= 0 || n > ldn || m < 0 || (s.lin_fail && !rescued) || (form == 0 && n > MG_BIGN)) { if (tid == 0) outrank[w] = -1; return; }
    double* Mm = GM ? Mscr + o2 : lds;
    const double* L = B.L + W.Lt_base;                // row-major lower, ld = n_red (k_chol_rr4 with export_full / k_chol_big)
    const double* y = B.y + W.loc_base + W.n_e + m;   // tail of the solution of S y = rhs
    if (form == 1) {
        // Cholesky square root, straight from the L buffer (no LDS residency, any tail up to ldn):
        //   A = L_nn L_nn^T, b = A y_n = L_nn (L_nn^T y_n), J = L_nn^T, r0 = L_nn^T y_n;  J^T J = A, J^T r0 = b
        const double* Ln = L + (size_t)m * nr + m;     // L_nn[i][r] = Ln[i * nr + r]
        // J = L_nn^T first; A is then formed from J's rows, J[r][i] J[r][j] with the lanes over j: coalesced, where the same products read
        // from the L buffer stride by n_red (the 263-dimension tail of the stress window: 2.7 ms that way).  Same terms in the same order.
        // (the reads run along the rows of L_nn.  A = J^T J follows in k_marg_gram, over the chip: this workgroup alone spent 1.4 ms on it
        // at the 263-dimension tail)
        double* Jw = outJ + o2;
        for (int e = tid; e < n * n; e += MG_NT) { int j = e / n, i = e - j * n; Jw[(size_t)i * n + j] = (j >= i) ? Ln[(size_t)j * nr + i] : 0.0; }
        for (int r = tid; r < n; r += MG_NT) {
            double a = 0;
            for (int c = r; c < n; c++) a += Ln[(size_t)c * nr + r] * y[c];
            outr0[o1 + r] = a; outw[o1 + r] = Ln[(size_t)r * nr + r] * Ln[(size_t)r * nr + r];
        }
        __syncthreads();
        for (int i = tid; i < n; i += MG_NT) {
            double a = 0;
            for (int r = 0; r <= i; r++) a += Ln[(size_t)i * nr + r] * outr0[o1 + r];
            outb[o1 + i] = a;
        }
        if (tid == 0) outrank[w] = n;
        return;
    }
    if (phase == 2) {
        for (int i = tid; i < n; i += MG_NT) bv[i] = outb[o1 + i];
        __syncthreads();
    } else {
    // G (column c contiguous) with G G^T = A: the columns of L_nn, or of the transposed rank-revealing factor of k_marg_rescue
    // (row r of resM is v_r, sum_r v_r v_r^T = A).  The loads run along the rows of the source (coalesced).
    if (rescued) {
        const double* Rm = resM + o2;
        for (int e = tid; e < n * n; e += MG_NT) Mm[e] = Rm[e];
    } else
        for (int e = tid; e < n * n; e += MG_NT) { int r = e / n, c = e - r * n; Mc(c, r) = (c <= r) ? L[(size_t)(m + r) * nr + m + c] : 0.0; }
    __syncthreads();
    // A = G G^T (= L_nn L_nn^T, the marginal information of the tail)
    if (!(GM && phase == 1))                              // (large tails: k_marg_gram, over the chip)
    for (int e = tid; e < n * n; e += MG_NT) {
        int i = e / n, j = e - i * n, k = rescued ? n - 1 : (i < j ? i : j);
        double a = 0;
        for (int r = 0; r <= k; r++) a += Mc(r, i) * Mc(r, j);
        outA[o2 + e] = a;
    }
    if (rescued) {
        for (int i = tid; i < n; i += MG_NT) { double a = resb[o1 + i]; bv[i] = a; outb[o1 + i] = a; }
    } else {
        // b = A y_n evaluated as L_nn (L_nn^T y_n), the same two triangular products as the Cholesky form (bit-identical b)
        for (int r = tid; r < n; r += MG_NT) { double a = 0; for (int c = r; c < n; c++) a += Mc(r, c) * y[c]; lam[r] = a; }
        __syncthreads();
        for (int i = tid; i < n; i += MG_NT) {
            double a = 0;
            for (int r = 0; r <= i; r++) a += Mc(r, i) * lam[r];
            bv[i] = a; outb[o1 + i] = a;
        }
    }
    __syncthreads();
    }
    // ---- one-sided Jacobi on the columns of G (Veselic / Hari: the Cholesky factor, not A, is what gets rotated).  Plane rotations
    // from the right, G <- G W, until all columns are mutually orthogonal: G W = U Sigma, so A = G G^T = (U Sigma)(U Sigma)^T and the
    // rows of J = Sigma U^T are the final columns themselves — no eigenvector matrix to accumulate, nothing divided by a small sigma,
    // and J^T J = A holds as long as every rotation is a rotation.  The implicit Gram matrix G^T G = L^T L is one LR step closer to
    // diagonal than A (round 3 rotated the columns of L^T, Gram matrix A, and carried V: 16 sweeps at the 263-dimension tail of cfg5
    // against the sweeps this order needs, at twice the traffic per rotation).
    if (!GM && !rescued && phase != 2) {
        // preconditioner: G <- the pivoted Cholesky factor of A (d_pivoted_chol; the rows are built in place, G lives in LDS)
        __threadfence_block();
        d_pivoted_chol<true>(outA + o2, lds, nullptr, nullptr, pc_dg, nullptr, 0, n, eps / (16.0 * n));
    }
    if (phase == 1) { if (tid == 0) bj_ok[w] = rescued ? 2 : 1; return; }      // k_marg_gram, k_marg_pchol (1 only) and the sweeps of k_marg_bj follow
    int grp = tid >> 4, sub = tid & 15;
    int ne = (n + 1) & ~1;                            // even number of players in the round-robin (a bye if n is odd)
    int sweeps_done = 0;
    for (int sweep = 0; sweep < (phase == 2 ? 0 : 40); sweep++) {
        sweeps_done = sweep + 1;
        if (tid == 0) { nrot = 0; crit_sh[0] = 0; crit_sh[1] = 0; }
        double mc2 = 0.0, ms2 = 0.0;
        __syncthreads();
        for (int st = 0; st < ne - 1; st++) {
            // circle method: player ne-1 is fixed, the others rotate; group k plays pair k (and k + 64) of this step
            for (int pr = grp; pr < ne / 2; pr += MG_NT / 16) {
                int p, q;
                if (pr == 0) { p = ne - 1; q = st; }
                else { p = st + pr; if (p >= ne - 1) p -= ne - 1; q = st - pr; if (q < 0) q += ne - 1; }
                if (p > q) { int t = p; p = q; q = t; }
                if (q >= n) continue;                 // the bye (odd n)
                double al = 0, be = 0, ga = 0;
                for (int r = sub; r < n; r += 16) { double a = Mc(p, r), b2 = Mc(q, r); al += a * a; be += b2 * b2; ga += a * b2; }
                al = grp16_sum(al); be = grp16_sum(be); ga = grp16_sum(ga);
                // (the second test keeps zeta^2 finite when a column is numerically null — rank-deficient tails, k_marg_rescue)
                if (ga * ga > 1e-30 * (al * be) && fabs(ga) > 1e-140 * (al + be)) {
                    // rotation from v_rcp / v_rsq + Newton steps: this scalar chain is the critical path of a step
                    double zeta = (be - al) * (0.5 * rcp_nr(ga));
                    double hz = 1.0 + zeta * zeta;
                    double t = (zeta >= 0 ? 1.0 : -1.0) * rcp_nr(fabs(zeta) + hz * rsqrt_nr(hz));
                    double c = rsqrt_nr(1.0 + t * t), sn = c * t;
                    mc2 = fmax(mc2, ga * ga * __builtin_amdgcn_rcp(al * be)); ms2 = fmax(ms2, sn * sn);
                    for (int r = sub; r < n; r += 16) { double a = Mc(p, r), b2 = Mc(q, r); Mc(p, r) = c * a - sn * b2; Mc(q, r) = sn * a + c * b2; }
                    if (sub == 0) atomicAdd(&nrot, 1);
                }
            }
            __syncthreads();
        }
        // (a sweep of tiny rotations leaves nothing for the next one to rotate: see k_marg_bj_crit)
        if (sub == 0 && ms2 > 0.0) { atomicMax(&crit_sh[0], (unsigned long long)__double_as_longlong(mc2)); atomicMax(&crit_sh[1], (unsigned long long)__double_as_longlong(ms2)); }
        __syncthreads();
        int done = nrot == 0 || (double)n * (double)n * __longlong_as_double((long long)crit_sh[0]) * __longlong_as_double((long long)crit_sh[1]) <= 1e-30;
        __syncthreads();
        if (done) break;
    }
    // eigenvalues = squared column norms; J = the columns as rows (row k = sigma_k u_k^T), r0 = Sigma^-1 U^T b = (column . b) / lambda;
    // rows ordered by ascending eigenvalue as Eigen returns them; eigenvalues <= eps dropped (null row, null r0) as the reference does.
    for (int c = grp; c < n; c += MG_NT / 16) {
        double a = 0;
        for (int r = sub; r < n; r += 16) a += Mc(c, r) * Mc(c, r);
        a = grp16_sum(a);
        if (sub == 0) lam[c] = a;
    }
    __syncthreads();
    int rank = 0;
    for (int c = 0; c < n; c++) rank += lam[c] > eps;                 // (cheap, every thread)
#ifdef MG_DEBUG_SWEEPS
    if (tid == 0) outrank[w] = sweeps_done;
#else
    if (tid == 0) outrank[w] = rank;
#endif
    for (int c = grp; c < n; c += MG_NT / 16) {
        const double lc = lam[c];
        const bool keep = lc > eps;
        int pos = 0;
        for (int k = 0; k < n; k++) pos += (lam[k] < lc) || (lam[k] == lc && k < c);
        double dotb = 0;
        for (int j = sub; j < n; j += 16) { const double v = Mc(c, j); outJ[o2 + (size_t)pos * n + j] = keep ? v : 0.0; dotb += v * bv[j]; }
        dotb = grp16_sum(dotb);
        if (sub == 0) { outr0[o1 + pos] = keep ? dotb / lc : 0.0; outw[o1 + pos] = lc; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_marg_bj — the Jacobi sweeps of the large tails (MG_MAXN < n <= MG_BIGN) as a BLOCK one-sided Jacobi over many workgroups.
// The single-workgroup loop above streams M and V (2 n^2 doubles) through one CU's L2 port at every one of its n - 1 steps per sweep:
// 184 ms for the 263-dimension prior of the cfg5 window.  Here the columns are cut into blocks of BS; a launch is one BLOCK step:
// workgroup g loads the columns of two blocks (M and V, 2 BS columns each) into LDS, orthogonalises every pair (p in the first block,
// q in the second) in BS inner steps of BS disjoint pairs (one wavefront per pair), and writes the columns back.  The block pairs of
// a step are disjoint (circle method), so a sweep is nb - 1 cross launches plus one launch for the pairs inside the blocks; every
// column pair meets exactly once per sweep, as in the cyclic order.  Sweeps end when one reports no rotation (rot[sweep] == 0):
// the remaining launches of the fixed schedule return at once.  Same rotation formulas and thresholds as k_marginalize.
// ---------------------------------------------------------------------------------------------------------------------
#define MG_SWEEPS 32
// A = M^T M of the large tails, one thread per entry over as many workgroups as it takes (the single workgroup of the set-up phase
// spent 2.6 ms on the 263-dimension tail here).  Full-length sums: the zeros of a triangular M add exact zeros, so the entries are
// those of the triangular loops above, bit for bit.
__global__ void __launch_bounds__(256) k_marg_gram(const int* tail_dim, int ldn, const double* Mscr, double* outA, const int* bj_ok) {
    const int w = blockIdx.y;
    if (bj_ok && !bj_ok[w]) return;                       // (no flags: the Cholesky form, every window)
    const int n = tail_dim[w], e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const size_t o2 = (size_t)w * ldn * ldn;
    const double* Mm = Mscr + o2;
    const int i = e / n, j = e - i * n;
    double a = 0;
    for (int r = 0; r < n; r++) a += Mc(r, i) * Mc(r, j);
    outA[o2 + e] = a;
}
// The Jacobi preconditioner of the large tails: G <- the pivoted Cholesky factor of A = G G^T (columns v_r, original row order), one
// workgroup per window, in the window's Mscr slab (ld = n; k_marg_gram is done with the old G).  Windows whose G already is one
// (bj_ok == 2: k_marg_rescue supplied it) pass.
__global__ void __launch_bounds__(1024) k_marg_pchol(const int* tail_dim, int ldn, const double* outA, double* vt_scr, double* Mscr, const int* bj_ok, int rc, double eps) {
    extern __shared__ double rs_lds[];                  // the pool's rows 24 x n | diagonal n | staging 24 x rc
    const int w = blockIdx.x;
    if (bj_ok[w] != 1) return;
    const int n = tail_dim[w];
    const size_t o2 = (size_t)w * ldn * ldn;
    d_pivoted_chol<false>(outA + o2, Mscr + o2, vt_scr + o2, rs_lds, rs_lds + (size_t)RS_POOL * n, rs_lds + (size_t)(RS_POOL + 1) * n, rc, n, eps / (16.0 * n));
}
template <int BS, int LDM, int NR>       // NR = rows per lane the launch's largest tail needs (n <= 64 NR <= LDM)
__global__ void __launch_bounds__(1024) k_marg_bj(const int* tail_dim, int ldn, double* Mscr, int* rot, unsigned long long* crit, const int* bj_ok, int sweep, int bstep) {
    static_assert(64 * NR <= LDM && 2 * BS <= 16, "a column's LDS row holds all 64 NR lanes' rows (zero past n); one wave per column");
    __shared__ double Ml[2 * BS][LDM];
    __shared__ double nrm[2 * BS];                        // squared norms of the columns, kept up to date rotation by rotation
    __shared__ int nrot;
    __shared__ unsigned long long crit_s[2];              // largest cos^2 between two rotated columns, largest sin^2 of a rotation (bit patterns)
    const int w = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (!bj_ok[w]) return;
    const int n = tail_dim[w];
    if (n > LDM || n > 64 * NR) return;
    if ((BS == 4) != (n > 576)) return;                    // the block size is a property of the WINDOW (8 columns up to 576 dimensions, 4 above): a batch launches both classes, and a window's rotation sequence does not depend on its neighbours
    int* rw = rot + (size_t)w * MG_SWEEPS;
    if (sweep > 0 && rw[sweep - 1] == 0) return;         // converged
    const int nb = (n + BS - 1) / BS, nbe = (nb + 1) & ~1;
    int P, Q;
    if (bstep < 0) { P = 2 * g; Q = 2 * g + 1; }          // the pairs inside blocks 2g and 2g + 1
    else {
        // circle method over the blocks: block nbe - 1 is fixed, the others rotate
        if (bstep >= nbe - 1 || g >= nbe / 2) return;     // (the schedule and the grid are the batch's largest window's: this one's sweep is over / has fewer block pairs)
        if (g == 0) { P = nbe - 1; Q = bstep; }
        else { P = bstep + g; if (P >= nbe - 1) P -= nbe - 1; Q = bstep - g; if (Q < 0) Q += nbe - 1; }
        if (P > Q) { int t = P; P = Q; Q = t; }
        if (Q >= nb) return;                              // the bye (odd number of blocks)
    }
    if (P >= nb) return;
    const size_t o2 = (size_t)w * ldn * ldn;
    double* Mm = Mscr + o2;
    auto gcol = [&](int lc) { int c = lc < BS ? P * BS + lc : Q * BS + (lc - BS); return (lc >= BS && Q >= nb) ? n : c; };
#ifdef SWF_PROFILE_CHOL
#define MGACC(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && tid == 0 && sweep == 1 && bstep == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_chol_stamps[i] += t_ - tacc; tacc = t_; } } while (0)
#define MGSTAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && tid == 0 && sweep == 1 && bstep == 0) g_chol_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MGSTAMP(i)
#define MGACC(i)
#endif
    unsigned long long tacc = 0; (void)tacc;
    MGSTAMP(50);
    // wave lc loads column lc: every request on its way before the first value is used (a launch is a few microseconds of work), zeros
    // past row n (the steps then need no row masks), and the column's squared norm — the only full-length sums of the launch besides
    // the one inner product per pair: a plane rotation by tan t moves t * <p, q> from one squared norm to the other.
    const bool mine = wv < 2 * BS && gcol(wv) < n;        // this wave's column exists
    double mp[NR];
    double al = 0.0;
#pragma unroll
    for (int k = 0; k < NR; k++) { const int r = lane + 64 * k; mp[k] = (mine && r < n) ? Mm[(size_t)gcol(wv) * n + r] : 0.0; }
    if (wv < 2 * BS) {
#pragma unroll
        for (int k = 0; k < NR; k++) { Ml[wv][lane + 64 * k] = mp[k]; al += mp[k] * mp[k]; }
        al = rows4_sum(grp16_sum(al));
        if (lane == 0) nrm[wv] = al;
    }
    if (tid == 0) { nrot = 0; crit_s[0] = 0; crit_s[1] = 0; }
    double mc2 = 0.0, ms2 = 0.0;
    int myrot = 0;
    __syncthreads();
    MGSTAMP(51);
#ifdef SWF_PROFILE_CHOL
    if (blockIdx.x == 1 && blockIdx.y == 0 && tid == 0 && sweep == 1 && bstep == 0) { for (int i = 54; i < 60; i++) g_chol_stamps[i] = 0; }
    tacc = __builtin_amdgcn_s_memtime();
#endif
    // one inner step = BS disjoint pairs, one wavefront each (8 of the 16 waves: two per SIMD).  A step is bound by the INSTRUCTIONS the
    // two waves of a SIMD issue, not by their latency (two half-wave pairs per wavefront were slower; so were three full-length sums per
    // pair), so the step is kept short: one inner product, the norms carried along, no row masks, and the rotation angle from the raw
    // v_rcp / v_rsq estimates — a rotation by a slightly inexact angle is still exactly a rotation as long as c^2 + s^2 = 1, which the
    // one Newton-refined rsqrt for c (s = c t) keeps to rounding; the angle only has to shrink the off-diagonal term.
    if (bstep >= 0) {
        // cross launches (all but one of a sweep): wave wv plays column p = wv of block P against every column of block Q in turn, so
        // its p stays in REGISTERS for the BS steps (it is the column the wave loaded) and only the q columns travel through LDS.
        const bool act = wv < BS && mine;
        for (int st = 0; st < BS; st++) {
            const int q = BS + ((wv + st) % BS);
            if (act && gcol(q) < n) {
                double mb[NR];
#pragma unroll
                for (int k = 0; k < NR; k++) mb[k] = Ml[q][lane + 64 * k];
                const double be = nrm[q];
                MGACC(54);
                double ga = 0;
#pragma unroll
                for (int k = 0; k < NR; k++) ga += mp[k] * mb[k];
                ga = rows4_sum(grp16_sum(ga));
                MGACC(55);
                // (no rotation: c = 1, s = t = 0 through the same instructions — the waves of a step stay in step)
                const bool go = ga * ga > 1e-30 * (al * be) && fabs(ga) > 1e-140 * (al + be);
                const double zeta = (be - al) * (0.5 * __builtin_amdgcn_rcp(ga));
                const double hz = 1.0 + zeta * zeta;
                const double t = go ? (zeta >= 0 ? 1.0 : -1.0) * __builtin_amdgcn_rcp(fabs(zeta) + hz * __builtin_amdgcn_rsq(hz)) : 0.0;
                const double c = rsqrt_nr(1.0 + t * t), sn = c * t;
                if (go) { mc2 = fmax(mc2, ga * ga * __builtin_amdgcn_rcp(al * be)); ms2 = fmax(ms2, sn * sn); myrot++; }
                MGACC(56);
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    const double a = mp[k], b2 = mb[k];
                    mp[k] = c * a - sn * b2;
                    Ml[q][lane + 64 * k] = sn * a + c * b2;
                }
                const double tg = t * ga;
                al -= tg;
                if (lane == 0) nrm[q] = be + tg;
                MGACC(57);
            }
            __syncthreads();
            MGACC(58);
        }
        if (lane == 0 && myrot) atomicAdd(&nrot, myrot);
    } else {
        // the launch of the pairs inside the blocks: two independent round-robins of BS players (blocks P and Q), waves 0 .. BS/2 - 1 play
        // in P, the others in Q; both columns of a pair change hands every step, so both travel through LDS.
        if (wv < 2 * BS) {
#pragma unroll
            for (int k = 0; k < NR; k++) mp[k] = 0.0;
        }
        for (int st = 0; st < BS - 1; st++) {
            if (wv < BS) {
                int p, q;
                const int blk = wv >= BS / 2, pr = wv - blk * (BS / 2);
                if (pr == 0) { p = BS - 1; q = st; }
                else { p = st + pr; if (p >= BS - 1) p -= BS - 1; q = st - pr; if (q < 0) q += BS - 1; }
                p += blk * BS; q += blk * BS;
                if (p > q) { int t = p; p = q; q = t; }
                if (gcol(p) < n && gcol(q) < n) {
                    double ma[NR], mb[NR];
#pragma unroll
                    for (int k = 0; k < NR; k++) { ma[k] = Ml[p][lane + 64 * k]; mb[k] = Ml[q][lane + 64 * k]; }
                    const double ap = nrm[p], be = nrm[q];
                    double ga = 0;
#pragma unroll
                    for (int k = 0; k < NR; k++) ga += ma[k] * mb[k];
                    ga = rows4_sum(grp16_sum(ga));
                    if (ga * ga > 1e-30 * (ap * be) && fabs(ga) > 1e-140 * (ap + be)) {
                        const double zeta = (be - ap) * (0.5 * __builtin_amdgcn_rcp(ga));
                        const double hz = 1.0 + zeta * zeta;
                        const double t = (zeta >= 0 ? 1.0 : -1.0) * __builtin_amdgcn_rcp(fabs(zeta) + hz * __builtin_amdgcn_rsq(hz));
                        const double c = rsqrt_nr(1.0 + t * t), sn = c * t;
                        mc2 = fmax(mc2, ga * ga * __builtin_amdgcn_rcp(ap * be)); ms2 = fmax(ms2, sn * sn);
#pragma unroll
                        for (int k = 0; k < NR; k++) {
                            Ml[p][lane + 64 * k] = c * ma[k] - sn * mb[k]; Ml[q][lane + 64 * k] = sn * ma[k] + c * mb[k];
                        }
                        if (lane == 0) { nrm[p] = ap - t * ga; nrm[q] = be + t * ga; }
                        myrot++;
                    }
                }
            }
            __syncthreads();
        }
        if (lane == 0 && myrot) atomicAdd(&nrot, myrot);
    }
    if (lane == 0 && ms2 > 0.0) { atomicMax(&crit_s[0], (unsigned long long)__double_as_longlong(mc2)); atomicMax(&crit_s[1], (unsigned long long)__double_as_longlong(ms2)); }
    __syncthreads();
    MGSTAMP(52);
    // write-back: a cross launch's p columns straight from the registers, everything else from LDS; nothing if nothing rotated
    if (nrot && mine) {
        double* col = Mm + (size_t)gcol(wv) * n;
        if (bstep >= 0 && wv < BS) {
#pragma unroll
            for (int k = 0; k < NR; k++) { const int r = lane + 64 * k; if (r < n) col[r] = mp[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < NR; k++) { const int r = lane + 64 * k; if (r < n) col[r] = Ml[wv][r]; }
        }
    }
    if (tid == 0 && nrot) { atomicAdd(&rw[sweep], nrot); atomicMax(&crit[2 * w], crit_s[0]); atomicMax(&crit[2 * w + 1], crit_s[1]); }
    MGSTAMP(53);
}

// End of a sweep: the sweep after one whose rotations were all tiny finds nothing to rotate, and need not be run to know it.  A rotation
// by an angle of sine s changes the inner product of either of its columns with a third column r by at most |s| times the cosine that
// r makes with the other one; with every cosine met in the sweep <= c_max and every sine <= s_max, the cosines the sweep leaves behind
// are below n s_max c_max.  Below the rotation threshold (1e-15) the sweep is recorded as the converged one (rot = 0): the launches
// of the next sweep would all return at once.  At cfg5's 263-dimension tail this is the ninth sweep (of 34 launches) not run.
__global__ void k_marg_bj_crit(const int* tail_dim, int nw, int* rot, unsigned long long* crit, const int* bj_ok, int sweep) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw || !bj_ok[w]) return;
    const double c2 = __longlong_as_double((long long)crit[2 * w]), s2 = __longlong_as_double((long long)crit[2 * w + 1]);
    const double n = (double)tail_dim[w];
    int* rw = rot + (size_t)w * MG_SWEEPS;
    if (rw[sweep] != 0 && n * n * c2 * s2 <= 1e-30) rw[sweep] = 0;
    crit[2 * w] = 0; crit[2 * w + 1] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Ambiguity covariance hand-off: what the integer-ambiguity consumer reads after a solve,
//   SWFOptimization::UpdateSchurHessianOnly  R/swf/swf_gnss.cpp:65-94    A = A3 A3^T, A3 = trailing n x n block of lhs_out2
//   SWFOptimization::LambdaSearch            R/swf/swf_lambda.cpp:94-99  Qy = A^-1 (the float ambiguities' covariance)
// A = L_nn L_nn^T, Qy = L_nn^-T L_nn^-1: one thread per column solves L z = e_j forwards and L^T q = z backwards (no explicit
// inverse of A; error eps * cond(L) instead of eps * cond(A)).  X = per-window scratch (column j contiguous).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tail_cov(DevBatch B, const int* tail_dim, int ldn, double* outA, double* outQ, double* X, int* outrank) {
    int w = blockIdx.x, tid = threadIdx.x;
    const WinRec& W = B.win[w];
    const WinState& s = B.ws[w];
    int n = tail_dim[w], nr = W.n_red, m = nr - n;
    size_t o2 = (size_t)w * ldn * ldn;
    if (n <= 0 || n > ldn || m < 0 || s.lin_fail) { if (tid == 0) outrank[w] = -1; return; }
    const double* Ln = B.L + W.Lt_base + (size_t)m * nr + m;     // L_nn[i][r] = Ln[i * nr + r]
    for (int e = tid; e < n * n; e += blockDim.x) {
        int i = e / n, j = e - i * n, k = i < j ? i : j;
        double a = 0;
        for (int r = 0; r <= k; r++) a += Ln[(size_t)i * nr + r] * Ln[(size_t)j * nr + r];
        outA[o2 + e] = a;
    }
    for (int j = tid; j < n; j += blockDim.x) {
        double* z = X + o2 + (size_t)j * n;
        for (int i = 0; i < j; i++) z[i] = 0.0;
        for (int i = j; i < n; i++) {                 // L z = e_j
            double a = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; k++) a -= Ln[(size_t)i * nr + k] * z[k];
            z[i] = a / Ln[(size_t)i * nr + i];
        }
        for (int i = n - 1; i >= 0; i--) {            // L^T q = z, in place
            double a = z[i];
            for (int k = i + 1; k < n; k++) a -= Ln[(size_t)k * nr + i] * z[k];
            z[i] = a / Ln[(size_t)i * nr + i];
        }
        for (int i = 0; i < n; i++) outQ[o2 + (size_t)i * n + j] = z[i];
    }
    if (tid == 0) outrank[w] = n;
}
