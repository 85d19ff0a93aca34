// swf_chol_rr4.h — k_chol_rr4 (round 5): the register-resident tiled Cholesky of the reduced system, n_red <= 240, rebuilt around
// three measurements of this round (tests/microbench/valu_rate.hip, mfma_f64_4x4.hip, pivot_chain_ff.hip; tools/prof/chol_wprof.sh):
//   * a lone wave issues ONE fp64 VALU instruction per ~8.5 cycles whatever the dependencies: the pivot wave of k_chol_rr3 was bound by
//     its ~26 instructions per column (v_rsq_f64 + two Newton steps + three scalings are 12 of them), not by a latency chain;
//   * v_mfma_f64_16x16x4_f64 issues every 64 cycles from one wave, dependent or not (the 105 cycles of rounds 3-4 were accumulator
//     copies in the micro-benchmark's loop): four panel MFMAs are 256 cycles, and ONE wave saturates a SIMD's matrix pipe;
//   * per-wave stamps inside k_chol_rr3: the inverse wave finished 1.8 k cycles behind the pivot wave, a tile wave spent 1.3 - 3.2 k
//     cycles per step between the barriers on ten statically unrolled slots x three phases of mostly skipped code, the read-modify-
//     write of the pending diagonal tiles in LDS, and three LDS round trips.
// What changed against k_chol_rr3 (same mathematics, same elimination order, same tile layout U = L^T in the MFMA accumulator layout):
//   * FRACTION-FREE PIVOT.  The diagonal tile is carried as M^(c) = s_c A^(c) (A^(c) = the Schur complement after c columns):
//         M^(c+1) = (dp M^(c) - col row^T) 2^-e,   dp = M^(c)_cc = m 2^e, m in [1, 2)        (exponent arithmetic on scalar registers)
//     so nothing is divided and no square root is taken while the columns are eliminated: 19 instructions per column instead of 26
//     for the pivot wave, and the inverse wave (same row operations on the identity) keeps up with it.  The Cholesky scaling
//     rho_c = 1 / sqrt(s_c dp_c) of ALL sixteen columns is one lane-parallel rsqrt at the end of the tile (s_c by a DPP prefix
//     product); the panel product applies it to its A operand.  L[r][c] = M^(c)[r][c] rho_c, Linv[r][:] = R^(r)[r][:] rho_r.
//   * ROW OWNERSHIP.  A tile wave owns whole tile rows — rows (t + 1, Tc - 1 - t) for wave t, the right-hand-side row for the last
//     wave: at most 15 tiles, row A at acc[J], row B at acc[14 - J], every register index static once the step loop is unrolled.
//     The pending diagonal tile of a row lives in its owner's registers (its terms are the owner's own panel tiles: no LDS
//     read-modify-write, no hand-off); a trailing update takes one operand from LDS and the other from the owner's registers;
//     a step is straight-line code: 2 panel products, then (Tc - 1 - j) x 2 guarded updates.
//   * 12 waves (768 threads, 168 registers): pivot + inverse on one SIMD, eight tile waves on the other three.
// Determinism: as before, every output element is formed by one lane from the same operands in the same order.
#pragma once
#include <utility>
#include <type_traits>

#define R4_NT 768
// transpose a 16x16 tile held in the accumulator layout through a wave-private [16][17] LDS scratch.  No s_waitcnt between the
// writes and the reads: the LDS executes one wave's instructions in order, so only the compiler has to keep them in place.
__device__ __forceinline__ void rr4_transpose(double4_t& a, double* X, int li, int lk) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; q++) X[(lk + 4 * q) * 17 + li] = a[q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = X[li * 17 + lk + 4 * q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// NS = tiles per tile wave = the most tile columns the instance takes: 14 (n_red <= 224: no register spills under the 168-register
// budget of twelve waves), 15 (224 < n_red <= 240) or 16 (240 < n_red <= 256, round 5: a 20-frame window of the reference's topology with 31 .. 46
// ambiguities; a few accumulator registers spill).  A launch of each covers a batch; every window belongs to exactly one.

// pivot wave: fraction-free elimination of the published diagonal tile D (full symmetric), lane (li, lk), register q <-> M[lk+4q][li].
// Column c: row c of M (the chain's copy, brought up to date one column ahead as in rr3_pivot_factor) is published for the
// inverse wave — its own element c is the pivot — and flagged; rows r > c: M[r][:] <- m M[r][:] - M[r][c] (M[c][:] 2^-e).
// Finished rows and columns are left to rot: nothing reads them again.  Both triangles stay bit-symmetric (a (b 2^-e) == b (a 2^-e)).
__device__ __forceinline__ bool rr4_pivot(double (*D)[17], double* colb, unsigned* flagb, int li, int lk, bool prof = false) {
#pragma clang fp contract(off)
    double A_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) A_[q] = D[lk + 4 * q][li];
    int bidx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bidx[r] = (r * 16 + li) * 4;
    const int pidx = (li & 3) * 4 + (li >> 2);            // row r of the tile sits at 4 (r mod 4) + r / 4: the inverse wave's lanes read their four rows as one run
    double rowA = bperm_d(A_[0], bidx[0]);                // M[0][li]
    double rowPre = bperm_d(A_[0], bidx[1]);              // row 1, before step 0
    int bad = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        CST(prof, c);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(rowA), c), lo = __builtin_amdgcn_readlane(__double2loint(rowA), c);
        bad |= (hi < 0x00100000) | (hi >= 0x7fd00000);    // pivot <= 0, subnormal, huge, Inf or NaN
        asm volatile("" ::: "memory");
        colb[c * 16 + pidx] = rowA;                       // (the four rows of lanes write the same values)
        asm volatile("" ::: "memory");
        __hip_atomic_store(flagb, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // "c + 1 columns published": releases the inverse wave (the LDS keeps a wave's operations in order)
        asm volatile("" ::: "memory");
        if (c == 15) break;
        const double sg = __hiloint2double(0x7fe00000 - (hi & 0x7ff00000), 0);               // 2^-e
        const double dpS = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);             // m
        const double rowS = rowA * sg;
        const double x = readlane_d(rowA, c + 1);                                            // M[c][c+1] == M[c+1][c]
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (4 * q + 3 <= c) continue;                 // every row of this register is finished
            const double col = row_newbcast_d(A_[q], c);  // M[lk+4q][c]
            A_[q] = __builtin_fma(dpS, A_[q], -(col * rowS));
        }
        // row c+2 after step c goes on its way across the lanes BEFORE the row fetched a column ago is used: the LDS crossbar's
        // ~130 cycles pass behind this column's arithmetic instead of in front of the next one's
        double rowPreN = 0;
        if (c + 2 < 16) rowPreN = bperm_d(A_[(c + 2) >> 2], bidx[(c + 2) & 3]);
        asm volatile("" ::: "memory");
        rowA = __builtin_fma(dpS, rowPre, -(x * rowS));                                      // row c+1 after step c (what the register copy has become above)
        rowPre = rowPreN;
    }
    return bad != 0;
}

// the Cholesky scaling of the tile, lane-parallel, by the PIVOT wave once its columns are through: lane li takes column li.
// rho_li = 1 / sqrt(s_li dp_li), s_li = prod_{k < li} m_k (the scale the eliminations have put on M^(li)): an inclusive DPP scan of m
// over the 16-lane row, shifted by one lane.  What the tile waves need is the ROW scaling of the unit-triangular solve,
// U = D^-1/2 (Ltilde^-1 A):  rsd_li = 1 / sqrt(d_li) = rho_li s_li (d_li = dp_li / s_li), stored at 4 (li mod 4) + li / 4 so that a lane
// reads its four rows as one run.  L_jj (export only) = published rows x rho.
__device__ __forceinline__ void rr4_rho(double* rsdJ, const double* colb, double (*Dl)[17], bool want_L, int li, int lk) {
#pragma clang fp contract(off)
    asm volatile("" ::: "memory");
    const double dpl = colb[li * 16 + (li & 3) * 4 + (li >> 2)];
    const double ml = __hiloint2double((__double2hiint(dpl) & 0x000fffff) | 0x3ff00000, __double2loint(dpl));
    double v = ml;
    v = v * __builtin_amdgcn_update_dpp(1.0, v, 0x111, 0xf, 0xf, false);          // row_shr:1 (lanes without a source keep 1.0)
    v = v * __builtin_amdgcn_update_dpp(1.0, v, 0x112, 0xf, 0xf, false);
    v = v * __builtin_amdgcn_update_dpp(1.0, v, 0x114, 0xf, 0xf, false);
    v = v * __builtin_amdgcn_update_dpp(1.0, v, 0x118, 0xf, 0xf, false);
    const double sl = __builtin_amdgcn_update_dpp(1.0, v, 0x111, 0xf, 0xf, false);
    const double rho = rsqrt_nr(sl * dpl);
    rsdJ[(li & 3) * 4 + (li >> 2)] = rho * sl;            // (the four rows of lanes write the same values)
    if (want_L) {
#pragma unroll
        for (int q = 0; q < 4; q++) { const int r = lk + 4 * q; Dl[r][li] = (li <= r) ? colb[li * 16 + lk * 4 + q] * rho : 0.0; }
    }
    asm volatile("" ::: "memory");
}

// transform wave (round 5, third form of the "inverse wave"): what the panel product needs of the diagonal tile, without an inverse.
// A streamed 16 x 16 inverse is a chain of fifteen columns at ~35 instructions each, and the wave finished 2.1 - 3.5 k cycles behind
// the pivot wave whatever was tried (own SIMD, prefetched columns, one progress counter).  The unit-triangular solve
// Utilde = Ltilde^-1 A (L_jj = Ltilde D^1/2) of a tile held as four 4-row blocks is four BLOCK GAUSS TRANSFORMS applied in place,
//     T <- T + G_k T_k,   G_k = [ 0 ; -(Wtilde_k + I) ; -Ltilde_{>k,k} Wtilde_k ]  (16 x 4),   Wtilde_k = inverse of the k-th unit-lower 4 x 4 block
// — one MFMA each with G_k as the A operand and register k of the tile as the B operand, the same four dependent MFMAs the product
// with the inverse took.  (T holds -A: row block k becomes +Utilde_k, the blocks below stay negated residuals.)  G_k needs columns
// 4k .. 4k+3 of the elimination only — the multipliers M[i][c] / dp_c, scale-free — so this wave works a BLOCK behind the pivot
// wave instead of a chain behind it: ~60 instructions per block, and only the last block's are left when the pivot wave is through.
// The square roots stay out of it: the tile waves scale the rows of Utilde by rsd (rr4_rho).  It has to be the UNIT-lower factor:
// Wtilde + I is exact (a 2 on the diagonal), whereas with the Cholesky-scaled blocks W + I rounds W away where 1 / L_cc is small and
// T_k - (W + I) T_k cancels (tests/perf/block_transform_accuracy.py: 1e-11 against 1e-16 on badly scaled tiles; measured on the
// device as L L^T = S to 4e-11 instead of 1e-12).
// Linv_jj, which the backward substitution multiplies by, is the same four transforms applied to -I by a tile wave after the loop.
__device__ __forceinline__ int rr4_progress(const unsigned* flagb) {
    return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flagb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void rr4_gops(double (*GoJ)[64], const double* colb, unsigned* flagb, int* failp, int lane, int li, int lk, bool prof = false) {
#pragma clang fp contract(off)
    const int pli = (li & 3) * 4 + (li >> 2);             // where row li sits in a published row
    int avail = 0;
    const double d0 = lk == 0 ? 1.0 : 0.0, d1 = lk == 1 ? 1.0 : 0.0, d2 = lk == 2 ? 1.0 : 0.0, d3 = lk == 3 ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c0 = 4 * k;
        for (int spin = 0; spin < (1 << 22) && avail < c0 + 4; spin++) { avail = rr4_progress(flagb); if (avail < c0 + 4) __builtin_amdgcn_s_sleep(0); }
        if (avail < c0 + 4 && lane == 0) *failp = 1;      // a stalled pivot wave: the window reports a linear-solver failure (never a silently wrong factor)
        asm volatile("" ::: "memory");
        CST(prof, 16 + k);
        // position of row c0 + u in a published row: 4 ((c0 + u) mod 4) + (c0 + u) / 4 = 4 u + k
        const double dpo = colb[li * 16 + pli];                                   // the pivot of column li (used by lanes c0 <= li < c0 + 4 only)
        double Mi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) Mi[u] = colb[(c0 + u) * 16 + pli];           // M[li][c0+u] (by symmetry: element li of published row c0+u)
        const double m10 = colb[(c0 + 0) * 16 + 4 * 1 + k], m20 = colb[(c0 + 0) * 16 + 4 * 2 + k], m30 = colb[(c0 + 0) * 16 + 4 * 3 + k];
        const double m21 = colb[(c0 + 1) * 16 + 4 * 2 + k], m31 = colb[(c0 + 1) * 16 + 4 * 3 + k], m32 = colb[(c0 + 2) * 16 + 4 * 3 + k];
        asm volatile("" ::: "memory");
        // 1 / pivot, every lane for its own column; the block's four to every lane (DPP row broadcast)
        const double idpo = rcp_nr(dpo);
        const double i0 = row_newbcast_d(idpo, c0), i1 = row_newbcast_d(idpo, c0 + 1), i2 = row_newbcast_d(idpo, c0 + 2), i3 = row_newbcast_d(idpo, c0 + 3);
        // the unit-lower multipliers inside the block, and column lk of the inverse of the block (forward substitution on e_lk)
        const double l10 = m10 * i0, l20 = m20 * i0, l30 = m30 * i0, l21 = m21 * i1, l31 = m31 * i1, l32 = m32 * i2;
        const double x0 = d0;
        const double x1 = __builtin_fma(-l10, x0, d1);
        const double x2 = __builtin_fma(-l21, x1, __builtin_fma(-l20, x0, d2));
        const double x3 = __builtin_fma(-l32, x2, __builtin_fma(-l31, x1, __builtin_fma(-l30, x0, d3)));
        // rows below the block: -sum_u Ltilde[li][c0+u] x_u
        double below = (Mi[0] * i0) * x0;
        below = __builtin_fma(Mi[1] * i1, x1, below);
        below = __builtin_fma(Mi[2] * i2, x2, below);
        below = __builtin_fma(Mi[3] * i3, x3, below);
        // rows of the block: -(Wtilde + I)[li - c0][lk]
        const int u = li - c0;
        const double xs = u == 0 ? x0 : u == 1 ? x1 : u == 2 ? x2 : x3;
        const double blk = xs + (u == lk ? 1.0 : 0.0);
        const double val = li < c0 ? 0.0 : li < c0 + 4 ? -blk : -below;
        GoJ[k][lane] = val;
    }
    CST(prof, 31);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flagb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // back to "nothing published" (avail == 16 was seen; the pivot wave starts the next tile two barriers from here)
    asm volatile("" ::: "memory");
}

// one tile of S into the (row-major) accumulator layout, lane (li, lk), register q <-> S[16 I + lk + 4q][16 J + li]; tile row Tc is the
// right-hand side (row n of the S storage) in its first row.  The row part of the addressing is worked out once per tile ROW
// (rr4_rowaddr: element offset of the lane's four rows, 0 and a cleared validity bit for a row outside the matrix), a tile adds its
// column: ~14 instructions per tile for interior and edge tiles alike (the per-element clamping of k_chol_rr3's edge path cost the
// owners of the last matrix row and of the right-hand side 5 - 7 k cycles, and everybody else that long at the first barrier).
struct rr4_rowaddr { int ro[4]; unsigned rv; };
__device__ __forceinline__ rr4_rowaddr rr4_row(int n, int Tc, int I, int lk) {
    rr4_rowaddr R; R.rv = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 16 * I + lk + 4 * q;
        const bool rhs_el = I == Tc && lk + 4 * q == 0;
        const bool valid = rhs_el || (I < Tc && r < n);
        R.ro[q] = valid ? (rhs_el ? n : r) * n : 0;
        R.rv |= valid ? 1u << q : 0u;
    }
    return R;
}
// (the loads only: what lies outside the matrix is cleared by rr4_tile_mask when the values are first touched, so that no load of
// the wave waits behind another tile's select)
__device__ __forceinline__ double4_t rr4_load_tile(const double* S, int n, const rr4_rowaddr& R, int J, int li) {
    double4_t a;
    const int c = 16 * J + li;
    const int cc = c < n ? c : 0;
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = S[R.ro[q] + cc];
    return a;
}
// negated, and zero outside the matrix
__device__ __forceinline__ void rr4_tile_mask(double4_t& a, int n, const rr4_rowaddr& R, int J, int li) {
    const bool cv = 16 * J + li < n;
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = (cv && ((R.rv >> q) & 1u)) ? -a[q] : 0.0;
}
// a diagonal tile of S, full symmetric (S is stored lower), identity on the padding, NEGATED (the tile waves keep -A)
__device__ __forceinline__ double4_t rr4_load_diag(const double* S, int n, int J, int li, int lk) {
    double4_t a;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 16 * J + lk + 4 * q, c = 16 * J + li;
        const int rc = r < n ? r : n - 1, cc = c < n ? c : n - 1;
        double v = S[(cc > rc) ? cc * n + rc : rc * n + cc];
        v = (r < n && c < n) ? v : (r == c ? 1.0 : 0.0);
        a[q] = -v;
    }
    return a;
}

// panel product of one tile of the step: T <- U_jI = D^-1/2 Ltilde_jj^-1 A_jI (T holds -A_jI; g = the four block Gauss transforms of the
// diagonal tile, rs = the lane's four row scalings), published to Pn[I].  The row's pending diagonal tile (Dg[I] in LDS, negated,
// touched by the row's owner only) takes its term here when it is the NEXT pivot tile and goes to Dt for the pivot wave; the other
// rows' after the barrier (rr4_diag_term).  Returns 0 for a tile that is still exactly zero: it stays zero, is not published, and
// every product with it is skipped (the step's mask nzm).
__device__ __forceinline__ void rr4_solve(double4_t& T, const double (&g)[4], const double (&rs)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) { const double b = T[k]; T = __builtin_amdgcn_mfma_f64_16x16x4f64(g[k], b, T, 0, 0, 0); }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) T[kk] *= rs[kk];
}
__device__ __forceinline__ unsigned rr4_panel(double4_t& T, const double (&g)[4], const double (&rs)[4], int I, int j, int Tc, double (*Pn)[4][64], double4_t d /* Dg[j+1], fetched ahead of the barrier */,
                                          double (*DtN)[17], unsigned* nzmj, int lane, int li, int lk, bool prof = false) {
    const bool crit = I == j + 1 && I < Tc;
    PST(prof && crit, 2);
    const bool tnz = __ballot((T[0] != 0.0) | (T[1] != 0.0) | (T[2] != 0.0) | (T[3] != 0.0)) != 0ull;
    if (!tnz) {
        if (crit) {
#pragma unroll
            for (int q = 0; q < 4; q++) DtN[lk + 4 * q][li] = -d[q];          // the next diagonal tile takes nothing from this step
        }
        return 0u;
    }
    if (lane == 0) atomicOr(nzmj, 1u << I);
    rr4_solve(T, g, rs);
    PST(prof && crit, 3);
    if (crit) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(T[kk], T[kk], d, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) DtN[lk + 4 * q][li] = -d[q];
    }
    PST(prof && crit, 4);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) Pn[I][kk][lane] = T[kk];
    PST(prof && crit, 5);
    return 1u;
}
// a later diagonal tile takes the step's term from its row's panel tile X (the row's owner is the only wave that touches Dg[I])
__device__ __forceinline__ void rr4_diag_term(const double4_t& X, int I, double (*Dg)[4][64], int lane) {
    double4_t d;
#pragma unroll
    for (int q = 0; q < 4; q++) d[q] = Dg[I][q][lane];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(X[kk], X[kk], d, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) Dg[I][q][lane] = d[q];
}

template <class F, int... Js>
__device__ __forceinline__ void rr4_steps(F& f, std::integer_sequence<int, Js...>) { (void)(f(std::integral_constant<int, Js>{}) && ...); }

template <int R4_NS>
__global__ void __launch_bounds__(R4_NT) k_chol_rr4(DevBatch B, int export_full) {
    __shared__ double Pn[17][4][64];           // published panel tiles of the step, registers as they are: Pn[I][kk][lane]; the transposition scratch before / after the loop
    __shared__ double Dg[16][4][64];           // the rows' pending diagonal tiles, negated, accumulator layout as it is: Dg[I][q][lane] (owner-private)
    __shared__ double Li[16][16][17];          // Linv_jj of every step (the transforms applied to -I once the loop is through: the backward pass multiplies by them)
    __shared__ double Gop[16][4][64];          // the four block Gauss transforms of every diagonal tile, A-operand layout: Gop[j][k][lane]
    __shared__ double rsd[16][16];             // 1 / sqrt(d) of every column, permuted (rr4_rho)
    __shared__ double Dt[2][16][17];           // published diagonal tiles, double-buffered
    __shared__ double Dl[16][17];              // L_jj on its way to HBM (export only)
    __shared__ double colb[16 * 16];           // pivot pair: row c of the tile being factored, per column
    __shared__ unsigned flagb[16];             //             "column c is published"
    __shared__ double zs[256];
    __shared__ double yv[256];
    __shared__ int wsimd[16];
    __shared__ unsigned nzm[2];                // bit I: panel tile (I, j) of the step is not all zero (double-buffered by step parity)
    __shared__ int fail;
    const int w = blockIdx.x;
    WinState& st = B.ws[w];
    if (!st.need_lin || st.lin_fail) return;
    const WinRec& W = B.win[w];
    const int n = W.n_red, tid = threadIdx.x;
    if (n <= 0 || n > 256 || n > B.rr_nmax) return;      // larger windows of a mixed batch belong to k_chol_big (launched next to this one)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int Tc = (n + 15) >> 4;
    if (R4_NS == 14 ? Tc > 14 : Tc != R4_NS) return;      // another instance's window
    // Tile rows -> tile waves (at most nine: three on each SIMD the pivot wave does not use), heaviest first: the right-hand-side row,
    // the k longest matrix rows one each, the other m = Tc - 1 - k rows in pairs (a, m + 1 - a) (a middle row alone).  The trailing
    // work of row I is I (I - 1) / 2 tile updates: dealt over the three SIMDs in snake order, the matrix pipes carry 159 / 150 / 146
    // of a cfg3 window's 455 updates (pairs (t + 1, Tc - 1 - t) in wave order: 150 / 204 / 101, and the steps waited for the 204).
    int gk = 0;
    for (int kk = Tc - 1 < 8 ? Tc - 1 : 8; kk >= 0; kk--) if (kk + ((Tc - kk) >> 1) + 1 <= 9) { gk = kk; break; }
    const int gm = Tc - 1 - gk, NW = 1 + gk + ((gm + 1) >> 1);      // rows left for the pairs; groups = tile waves
    // first tile row / column whose factor is written out: everything, or the parameter_head tail block, or nothing
    const int ef = export_full ? 0 : (W.tail_dim > 0 ? (n - W.tail_dim) >> 4 : Tc);
    double* Lrm = B.L + W.Lt_base;
    CHSTAMP(0);
    if (tid == 0) { fail = 0; nzm[0] = 0u; nzm[1] = 0u; }
    if (tid < 16) flagb[tid] = 0u;
    for (int e = tid; e < 256; e += R4_NT) zs[e] = 0.0;
    // Roles by SIMD (HW_ID): wave 0 is the pivot wave; its SIMD takes the inverse wave and no matrix-core work (fp64 MFMAs and fp64 VALU
    // instructions of one SIMD do not overlap); the tile waves come from the other SIMDs, the rest leave.
    if (lane == 0) wsimd[wv] = (int)__builtin_amdgcn_s_getreg(2308);       // HW_REG_HW_ID, SIMD_ID (bits 5:4)
    __syncthreads();                                       // R
    int role, tw;                                          // role 0 pivot, 1 inverse, 2 tile wave tw, 3 none
    {
        const int nwv = R4_NT / 64;
        const int ps = wsimd[0];
        const unsigned all = (1u << nwv) - 2u;                                                                   // waves 1 .. nwv-1
        const unsigned onp = (unsigned)__ballot(lane >= 1 && lane < nwv && wsimd[lane & 15] == ps) & all;        // other waves on the pivot's SIMD
        const unsigned off = all & ~onp;
        // the inverse wave: on the pivot wave's SIMD (no matrix-core work there: next to two tile waves it took 8.5 k cycles per tile
        // in the update-heavy steps, 4.9 k alone, 5.1 k beside the pivot wave, which as the older wave keeps its own pace)
        const unsigned rbit = onp ? (onp & (0u - onp)) : off ? (1u << (31 - __clz(off))) : 2u;
        unsigned rest = onp & ~rbit, tiles = off & ~rbit;
        for (int need = NW - __popc(tiles); need > 0 && rest; need--) { unsigned b = rest & (0u - rest); tiles |= b; rest &= ~b; }
        const unsigned me = 1u << wv;
        // group of this wave: snake order over (slot on the SIMD, SIMD) when the waves sit three to a SIMD, else plain wave order
        unsigned sm[4], simds = 0u;
        bool three = __popc(tiles) == 9;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            sm[x] = (unsigned)__ballot(lane < nwv && wsimd[lane & 15] == x) & tiles;
            if (sm[x]) { simds |= 1u << x; three = three && __popc(sm[x]) == 3; }
        }
        int g = __popc(tiles & (me - 1u));
        if (three && (tiles & me)) {
            const int mys = wsimd[wv], sr = __popc(simds & ((1u << mys) - 1u)), t = __popc(sm[mys] & (me - 1u));
            g = t == 0 ? sr : t == 1 ? 5 - sr : 6 + sr;
        }
        role = wv == 0 ? 0 : (rbit & me) ? 1 : ((tiles & me) && g < NW) ? 2 : 3;
        tw = g;
    }
    if (role == 3) return;
    if (role == 0) {
        // =============================== pivot wave ===============================
        __syncthreads();                                   // A_0: tile (0,0) published
        CHSTAMP(3);
        for (int j = 0; j < Tc; j++) {
            WST(j, 0);
#ifdef SWF_PROFILE_CHOLW
            const bool bad = rr4_pivot(Dt[j & 1], colb, flagb, li, lk, j == g_chol_wstep);
#else
            const bool bad = rr4_pivot(Dt[j & 1], colb, flagb, li, lk);
#endif
            WST(j, 1);
            rr4_rho(rsd[j], colb, Dl, j >= ef, li, lk);
            if (bad && lane == 0) fail = 1;
            __syncthreads();                               // B_j
#ifdef SWF_PROFILE_CHOL
            if (blockIdx.x == 0 && lane == 0 && j < 15) g_chol_stamps[49 + j] = __builtin_amdgcn_s_memtime();
#endif
            WST(j, 2);
            if (fail) { if (tid == 0) { st.lin_fail = 1; st.chol_fail = 1; } return; }
            if (lane == 0) nzm[(j + 1) & 1] = 0u;          // the next step's mask (its last readers left before B_j)
            __syncthreads();                               // C_j
            WST(j, 3);
        }
        CHSTAMP(1);
        __syncthreads();                                   // E: L exported, yv ready
        CHSTAMP(4);
        // backward solve y = L^-T z, right-looking: yv holds z; once y_J is known the owner of tile row J subtracts
        // L_{J,J'}^T y_J from the pending blocks J' < J.  Here: y_J = Linv_JJ^T yv_J on all 64 lanes
        for (int J = Tc - 1; J >= 0; J--) {
            double p = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) p += Li[J][lk + 4 * q][li] * yv[16 * J + lk + 4 * q];
            atomicAdd(&zs[16 * J + li], p);                // ds_add_f64: the four row groups of lanes add into the (zeroed) slot
            __syncthreads();                               // X_J: y_J published
            __syncthreads();                               // Y_J: row J applied to the pending blocks
        }
        CHSTAMP(2);
        return;
    }
    if (role == 1) {
        // =============================== inverse wave ===============================
        __syncthreads();                                   // A_0
        for (int j = 0; j < Tc; j++) {
#ifdef SWF_PROFILE_CHOLW
            rr4_gops(Gop[j], colb, flagb, &fail, lane, li, lk, j == g_chol_wstep);
#else
            rr4_gops(Gop[j], colb, flagb, &fail, lane, li, lk);
#endif
            WST(j, 1);
            __syncthreads();                               // B_j
            if (fail) return;
            if (j >= ef) {
                // L_jj to HBM (nothing overwrites Dl before the next tile's end)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int r = 16 * j + lk + 4 * q, c = 16 * j + li;
                    if (r < n && c <= r) Lrm[(size_t)r * n + c] = Dl[lk + 4 * q][li];
                }
            }
            __syncthreads();                               // C_j
        }
        __syncthreads();                                   // E
        for (int J = Tc - 1; J >= 0; J--) { __syncthreads(); __syncthreads(); }
        return;
    }
    // =============================== tile waves ===============================
    const double* S = B.S + W.S_base;
    double* Xs = &Pn[0][0][0] + tw * 272;                  // wave-private transposition scratch
    const bool isrhs = tw == 0;
    const int pa_ = tw - gk;                               // pair number (1 ..) behind the right-hand-side row and the gk single rows
    // row A: tiles (Ia, J), J < Ia, at acc[J]; row B: tiles (Ib, J), J < Ib, at acc[NS - 1 - J]
    const int Ia = isrhs ? Tc : tw <= gk ? Tc - tw : pa_ <= (gm >> 1) ? pa_ : (gm + 1) >> 1;
    const int Ib = (!isrhs && tw > gk && pa_ <= (gm >> 1)) ? gm + 1 - pa_ : 0;
    const unsigned rmA = (1u << Ia) - 1u, rmB = (1u << Ib) - 1u;      // the columns of rows A and B
    double4_t acc[R4_NS];
    CHSTAMP2(16);
    WST(-1, 0);
    // tiles no block pair of the window reaches are zero in S (and stay out of the load: a third of a cfg3 window's tiles — the
    // speed-bias columns couple to few poses before the elimination fills them in): one bit per tile from the host's symbolic phase
    unsigned tzA = ~0u, tzB = ~0u;                         // bit J: tile (Ia, J) / (Ib, J) is loaded
    if (B.s_tnz) {
        const unsigned* tm = B.s_tnz + (size_t)w * 4;
        const unsigned long long m0 = (unsigned long long)tm[0] | ((unsigned long long)tm[1] << 32), m1 = (unsigned long long)tm[2] | ((unsigned long long)tm[3] << 32);
        if (!isrhs) { const int t0 = Ia * (Ia - 1) / 2; tzA = (unsigned)(t0 < 64 ? (m0 >> t0) | (t0 ? m1 << (64 - t0) : 0ull) : m1 >> (t0 - 64)); }
        if (Ib) { const int t0 = Ib * (Ib - 1) / 2; tzB = (unsigned)(t0 < 64 ? (m0 >> t0) | (t0 ? m1 << (64 - t0) : 0ull) : m1 >> (t0 - 64)); }
    }
    const rr4_rowaddr rwA = rr4_row(n, Tc, Ia, lk), rwB = rr4_row(n, Tc, Ib ? Ib : Ia, lk);
#pragma unroll
    for (int J = 0; J < R4_NS; J++) {
        // (rows A and B never share a slot: Ia + Ib <= Tc <= 15)
        acc[J] = double4_t{ 0, 0, 0, 0 };
        if (J < Ia) { if ((tzA >> J) & 1u) acc[J] = rr4_load_tile(S, n, rwA, J, li); }
        else if (R4_NS - 1 - J < Ib) { if ((tzB >> (R4_NS - 1 - J)) & 1u) acc[J] = rr4_load_tile(S, n, rwB, R4_NS - 1 - J, li); }
    }
    if (!isrhs) {
        const double4_t d = rr4_load_diag(S, n, Ia, li, lk);
#pragma unroll
        for (int q = 0; q < 4; q++) Dg[Ia][q][lane] = d[q];
    }
    if (Ib) {
        const double4_t d = rr4_load_diag(S, n, Ib, li, lk);
#pragma unroll
        for (int q = 0; q < 4; q++) Dg[Ib][q][lane] = d[q];
    }
    if (isrhs) {
        const double4_t d0 = rr4_load_diag(S, n, 0, li, lk);
#pragma unroll
        for (int q = 0; q < 4; q++) Dt[0][lk + 4 * q][li] = -d0[q];
    }
    CHSTAMP2(17);
    WST(-1, 1);
    // negated, into the transposed layout
    // (the LDS port is what bounds this phase: eight waves x 14 tiles x 2 KB written and read back = 3.6 k cycles at 128 B per cycle;
    // batching several tiles per round trip was measured and is no faster)
#pragma unroll
    for (int J = 0; J < R4_NS; J++)
        if (J < Ia ? ((tzA >> J) & 1u) != 0u : (R4_NS - 1 - J < Ib && ((tzB >> (R4_NS - 1 - J)) & 1u) != 0u)) {
            if (J < Ia) rr4_tile_mask(acc[J], n, rwA, J, li); else rr4_tile_mask(acc[J], n, rwB, R4_NS - 1 - J, li);
            rr4_transpose(acc[J], Xs, li, lk);
        }
    CHSTAMP2(18);
    WST(-1, 2);
    __syncthreads();                                       // A_0
    WST(-1, 3);
    CHSTAMP2(19);
    // one step, j a compile-time constant (every register index below is static); false ends the factorisation (j == Tc, or a failed pivot)
    auto step = [&](auto jc) -> bool {
        constexpr int j = decltype(jc)::value;
        if (j >= Tc) return false;
        // the next pivot tile's pending value is final since this wave's own update of step j-1: fetched ahead of the barrier by its row's owner
        double4_t dn = { 0, 0, 0, 0 };
        if (j + 1 < Tc && (Ia == j + 1 || Ib == j + 1)) {
#pragma unroll
            for (int q = 0; q < 4; q++) dn[q] = Dg[j + 1][q][lane];
        }
        WST(j - 1, 6);
        __syncthreads();                                   // B_j: Linv_jj ready; trailing updates of step j-1 done
        WST(j - 1, 7);
        WST(j, 0);
        if (fail) return false;
#ifdef SWF_PROFILE_CHOLW
        const bool prof = j == g_chol_wstep && (Ia == j + 1 || Ib == j + 1);
#else
        const bool prof = false;
#endif
        PST(prof, 0);
        const bool pa = j < Ia, pb = j < Ib;
        unsigned nza = 0u, nzb = 0u;                       // this step's panel tiles are not zero (wave-uniform)
        if (pa || pb) {
            // the diagonal tile's block Gauss transforms (A operands) and the lane's row scalings
            double aop[4], rs4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) aop[k] = Gop[j][k][lane];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) rs4[kk] = rsd[j][lk * 4 + kk];
            PST(prof, 1);
            // the row that holds the next pivot tile first
            if (pb && Ib == j + 1) {
                nzb = rr4_panel(acc[R4_NS - 1 - j], aop, rs4, Ib, j, Tc, Pn, dn, Dt[(j + 1) & 1], &nzm[j & 1], lane, li, lk, prof);
                if (pa) nza = rr4_panel(acc[j], aop, rs4, Ia, j, Tc, Pn, dn, Dt[(j + 1) & 1], &nzm[j & 1], lane, li, lk, prof);
            } else {
                if (pa) nza = rr4_panel(acc[j], aop, rs4, Ia, j, Tc, Pn, dn, Dt[(j + 1) & 1], &nzm[j & 1], lane, li, lk, prof);
                if (pb) nzb = rr4_panel(acc[R4_NS - 1 - j], aop, rs4, Ib, j, Tc, Pn, dn, Dt[(j + 1) & 1], &nzm[j & 1], lane, li, lk, prof);
            }
        }
        WST(j, 1);
        PST(prof, 6);
        __syncthreads();                                   // C_j: panel and diagonal tile j+1 published
        PST(prof, 7);
        WST(j, 2);
        const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)nzm[j & 1]);
        WST(j, 3);
        // the rows' pending diagonal tiles take their term of this step (the next pivot tile took its own inside the panel phase)
        if (nza && !isrhs && Ia > j + 1) rr4_diag_term(acc[j], Ia, Dg, lane);
        if (nzb && Ib > j + 1) rr4_diag_term(acc[R4_NS - 1 - j], Ib, Dg, lane);
        WST(j, 4);
        // trailing updates -A_IJ += U_jJ^T U_jI: U_jJ from the published panel, U_jI from this wave's own registers.
        // ma / mb: the columns J > j whose tile of row A / row B takes a term (scalar bit masks: one s_bitcmp per test)
        const unsigned ma = nza ? (m & rmA) : 0u, mb = nzb ? (m & rmB) : 0u, mab = ma | mb;
#pragma unroll
        for (int J = j + 1; J < R4_NS; J++) {
            if (!((mab >> J) & 1u)) continue;
            double pj[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) pj[kk] = Pn[J][kk][lane];
            if ((ma >> J) & 1u) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[kk], acc[j][kk], acc[J], 0, 0, 0);
            }
            if ((mb >> J) & 1u) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc[R4_NS - 1 - J] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[kk], acc[R4_NS - 1 - j][kk], acc[R4_NS - 1 - J], 0, 0, 0);
            }
        }
        WST(j, 5);
        return true;
    };
    rr4_steps(step, std::make_integer_sequence<int, R4_NS>{});
    if (fail) return;
    CHSTAMP2(20);
    // Linv_JJ for the backward substitution: the transforms of tile J applied to -I (accumulator layout = the row layout Li is read in)
    for (int J = tw; J < Tc; J += NW) {
        double g[4], rs4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = Gop[J][k][lane];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) rs4[kk] = rsd[J][lk * 4 + kk];
        double4_t T;
#pragma unroll
        for (int q = 0; q < 4; q++) T[q] = (lk + 4 * q == li) ? -1.0 : 0.0;
        rr4_solve(T, g, rs4);
#pragma unroll
        for (int q = 0; q < 4; q++) Li[J][lk + 4 * q][li] = T[q];
    }
    // back to the row layout (lane (li, lk), register q <-> L[lk+4q][li]); export L where it is read, y = L^-1 rhs from the rhs tile row
#pragma unroll
    for (int J = 0; J < R4_NS; J++) {
        const bool a = J < Ia, b = R4_NS - 1 - J < Ib;
        if (!(a || b)) continue;
        rr4_transpose(acc[J], Xs, li, lk);
        const int I = a ? Ia : Ib, Jc = a ? J : R4_NS - 1 - J;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 16 * I + lk + 4 * q, c = 16 * Jc + li;
            if (I < Tc) { if (Jc >= ef && r < n) Lrm[(size_t)r * n + c] = acc[J][q]; }
            else if (lk + 4 * q == 0) yv[c] = acc[J][q];
        }
    }
    CHSTAMP2(21);
    __syncthreads();                                       // E
    for (int J = Tc - 1; J >= 0; J--) {
        __syncthreads();                                   // X_J: y_J published
        // tiles (J, J') of row J, J' < J: yv_J' -= L_{J,J'}^T y_J — all of them in the row's owner
        if (!isrhs && (Ia == J || Ib == J)) {
            double zq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) zq[q] = zs[16 * J + lk + 4 * q];
            const bool a = Ia == J;
#pragma unroll
            for (int Jp = 0; Jp < R4_NS - 1; Jp++) {
                if (Jp >= J) continue;
                double p = 0;
                if (a) {
#pragma unroll
                    for (int q = 0; q < 4; q++) p += acc[Jp][q] * zq[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) p += acc[R4_NS - 1 - Jp][q] * zq[q];
                }
                atomicAdd(&yv[16 * Jp + li], -p);          // ds_add_f64 (the tiles of row J update distinct blocks J')
            }
        }
        __syncthreads();                                   // Y_J
    }
    double* y = B.y + W.loc_base + W.n_e;
    for (int e = tw * 64 + lane; e < n; e += NW * 64) y[e] = zs[e];
}
