// swf_chol_rr.h — k_chol_rr3: the register-resident tiled Cholesky of the reduced system, n_red <= 240 (included by swf_kernels2.h).
//
// One workgroup of 16 waves per window; the whole factor lives in the tile waves' registers.  Against k_chol_rr2 (kept for A/B
// runs, SWF_CHOL_RR2=1) the critical path — pivot tile j -> panel tile (j+1, j) -> diagonal tile j+1 -> pivot tile j+1 — is shorter:
//   * the tiles are held TRANSPOSED (the upper factor U = L^T in the MFMA accumulator layout: lane (li, lk), register q <-> U[lk+4q][li]).
//     In that layout a tile's registers ARE the B operand of  U_jI = Linv_jj A_jI  (panel) and both operands of the trailing update
//     A_JI -= U_jJ^T U_jI, so the panel needs no LDS round trip before its MFMAs and publishes its registers as they are;
//   * the wave that owns panel tile (j+1, j) also owns diagonal tile (j+1, j+1): it updates and publishes it straight from its own
//     registers, inside the panel phase — two workgroup barriers per step instead of three;
//   * the pivot is two waves on a SIMD of their own: wave 0 factors the diagonal tile (the 16-step rsqrt chain and nothing else),
//     the first other wave of its SIMD runs the forward substitution L X = I one column behind it, fed through LDS (the row of A
//     and 1/sqrt(pivot) per column; the latter doubles as the "column ready" flag); the SIMD's remaining waves own no tiles,
//     because fp64 MFMAs and fp64 VALU work of one SIMD do not overlap (tests/microbench/pivot_chain.hip);
//   * the factor is written to HBM only where somebody reads it: the parameter_head tail block (marginalisation / covariance
//     hand-off) on solve paths, everything in ASSEMBLE_ELIMINATE_ONLY mode (swf_batch_export_reduced).
// Plain dense Cholesky of S in the predefined elimination order, only tiled; same MFMA term order as k_chol_rr2.
#pragma once

#define RR3_NS 10        // off-diagonal tiles per tile wave (120 of a 240-dimension system, rhs row included, over 12 or more waves)
#define RR3_MINTW 12     // tile waves the roles guarantee

// transpose a 16x16 tile held in the accumulator layout through a wave-private [16][17] LDS scratch.  No s_waitcnt between the
// writes and the reads: the LDS executes one wave's instructions in order, so only the compiler has to keep them in place.
__device__ __forceinline__ void rr3_transpose(double4_t& a, double* X, int li, int lk) {
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

// wave 1 waits for column c of the pivot wave: 1/sqrt(pivot) is written last (the LDS keeps a wave's operations in order) into a slot
// that is zero between tiles; any non-zero bit pattern (NaN of a broken pivot included) releases the wait
__device__ __forceinline__ double rr3_wait_ip(const double* slot) {
    asm volatile("" ::: "memory");
    double v = 0.0;
    for (int spin = 0; spin < (1 << 22); spin++) {
        v = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (__builtin_amdgcn_readfirstlane(__double2hiint(v)) != 0) break;
        __builtin_amdgcn_s_sleep(0);
    }
    asm volatile("" ::: "memory");
    return v;
}

// wave 0: right-looking Cholesky of the published diagonal tile D (full symmetric), lane (li, lk), register q <-> A[lk+4q][li].
// Column c: ip = 1/sqrt(A_cc); row c of A (= column c, A is symmetric) is handed to wave 1 with ip; then
// A[r][:] -= A[r][c] A[c][:] / A_cc.  Finished rows are never read again, so the register update runs unmasked over the rows (it
// only dirties the upper triangle of finished rows).  The 16-step dependency chain is kept as short as the arithmetic allows:
//   * row c+1 is fetched across the lanes (ds_bpermute) one column AHEAD, before step c's update, and brought up to date by the
//     one FMA the register copy gets too (same operands: the same bits), so neither the LDS round trip nor the update of the
//     four registers sits between one pivot and the next: chain = rsqrt, scaled row, that FMA, v_readlane of the next pivot;
//   * 1/sqrt is v_rsq_f64 + two Newton steps (rsqrt_nr) opened up: the scaled row sA = A[c][:] / A_cc leaves two operations
//     after the last Newton factor f, sA = (row y1 f) (y1 f), instead of four (ip, ip^2, row ip^2, mask).
// Column scalings are deferred (as in chol_pivot_tile).
__device__ __forceinline__ bool rr3_pivot_factor(double (*D)[17], double* colb, double* ipb, double (*Dl)[17], bool want_L, int li, int lk) {
#pragma clang fp contract(off)
    double A_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) A_[q] = D[lk + 4 * q][li];
    bool bad = false;
    int bidx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bidx[r] = (r * 16 + li) * 4;
    const int pidx = (li & 3) * 4 + (li >> 2);            // row r of the tile sits at 4 (r mod 4) + r / 4: wave 1's lanes read their four rows as one run
    double rowA = bperm_d(A_[0], bidx[0]);                // A[0][li]
    double dp = readlane_d(A_[0], 0);
    double rowPre = bperm_d(A_[0], bidx[1]);              // row 1, before step 0
#pragma unroll
    for (int c = 0; c < 16; c++) {
        if (!(dp > 0.0)) bad = true;
        double y = __builtin_amdgcn_rsq(dp), h = 0.5 * dp;
        y = y * __builtin_fma(-(h * y), y, 1.5);
        const double f = __builtin_fma(-(h * y), y, 1.5);
        const double ip = y * f;
        const double row0 = (li > c) ? rowA : 0.0;        // columns <= c of A are final (L) already
        const double sA = ((row0 * y) * f) * ip;
        asm volatile("" ::: "memory");
        colb[c * 16 + pidx] = rowA;                       // (the four rows of lanes write the same values)
        asm volatile("" ::: "memory");
        __hip_atomic_store((unsigned long long*)(ipb + c), (unsigned long long)__double_as_longlong(ip), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ... and this releases wave 1 (the LDS keeps a wave's operations in order)
        asm volatile("" ::: "memory");
        if (c == 15) break;
        const int c1 = c + 1;
        const double x = readlane_d(A_[c1 >> 2], (c1 & 3) * 16 + c);          // A[c+1][c]: final since step c-1
        const double rowNext = __builtin_fma(-x, sA, rowPre);                // row c+1 after step c (what the register copy becomes below)
        const double dpNext = readlane_d(rowNext, c1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (4 * q + 3 <= c) continue;                 // every row of this register is final already
            double col = row_newbcast_d(A_[q], c);        // A[lk+4q][c]
            A_[q] = __builtin_fma(-col, sA, A_[q]);
        }
        if (c + 2 < 16) rowPre = bperm_d(A_[(c + 2) >> 2], bidx[(c + 2) & 3]);      // row c+2 after step c, for the column after next
        rowA = rowNext; dp = dpNext;
    }
    if (want_L) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const double ipc = ipb[li];
#pragma unroll
        for (int q = 0; q < 4; q++) { int r = lk + 4 * q; Dl[r][li] = (li <= r) ? A_[q] * ipc : 0.0; }
    }
    return bad;
}

// wave 1: X = L^-1 by forward substitution, column by column behind wave 0:  R[r][:] -= A[r][c] R[c][:] / A_cc for r > c.
// One LDS round trip per column: the poll of 1/sqrt(pivot) and the reads of the column go out together (the column was written
// first, so a released poll means the reads behind it saw it); row c+1 of R is fetched across the lanes a column ahead and
// brought up to date by the FMA its register copy gets too, as in the pivot wave.
__device__ __forceinline__ void rr3_pivot_inverse(double (*LiJ)[17], const double* colb, double* ipb, int li, int lk) {
#pragma clang fp contract(off)
    double R_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) R_[q] = (lk + 4 * q == li) ? 1.0 : 0.0;
    int bidx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bidx[r] = (r * 16 + li) * 4;
    double rowR = li == 0 ? 1.0 : 0.0;                    // R[0][li]
    double rowPre = li == 1 ? 1.0 : 0.0;                  // row 1, before step 0
#pragma unroll
    for (int c = 0; c < 15; c++) {
        double ip, colv[4];
        asm volatile("" ::: "memory");
        for (int spin = 0; spin < (1 << 22); spin++) {
            asm volatile("" ::: "memory");
            ip = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(ipb + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
#pragma unroll
            for (int q = 0; q < 4; q++) colv[q] = colb[c * 16 + lk * 4 + q];      // A[lk+4q][c]
            asm volatile("" ::: "memory");
            if (__builtin_amdgcn_readfirstlane(__double2hiint(ip)) != 0) break;
            __builtin_amdgcn_s_sleep(0);
        }
        asm volatile("" ::: "memory");
        const double ip2 = ip * ip;
        const double sR = rowR * ip2;
        const int c1 = c + 1;
        const double x = readlane_d(colv[c1 >> 2], (c1 & 3) * 16);            // A[c+1][c]
        const double rowNext = __builtin_fma(-x, sR, rowPre);                // row c+1 of R after step c
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (4 * q + 3 <= c) continue;
            if (lk + 4 * q > c) R_[q] = __builtin_fma(-colv[q], sR, R_[q]);
        }
        if (c + 2 < 16) rowPre = bperm_d(R_[(c + 2) >> 2], bidx[(c + 2) & 3]);      // row c+2 after step c
        rowR = rowNext;
    }
    (void)rr3_wait_ip(ipb + 15);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int r = lk + 4 * q;
        LiJ[r][li] = (li <= r) ? R_[q] * ipb[r] : 0.0;    // row r of X = L^-1
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) < 16) __hip_atomic_store((unsigned long long*)(ipb + (threadIdx.x & 63)), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // slots back to "not yet" (wave 0 starts the next tile two barriers from here)
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(1024) k_chol_rr3(DevBatch B, int export_full) {
    __shared__ double Pn[16][4][64];           // published panel tiles, registers as they are: Pn[I][kk][lane]; the transposition scratch before / after the loop
    __shared__ double Dg[16][4][64];           // the diagonal tiles still to be factored, accumulator layout as it is: Dg[J][q][lane]
    __shared__ double Li[16][16][17];          // Linv_jj of every step (the backward pass multiplies by them again)
    __shared__ double Dt[2][16][17];           // published diagonal tiles, double-buffered
    __shared__ double Dl[16][17];              // L_jj on its way to HBM (export only)
    __shared__ double colb[16 * 16];           // pivot pair: row c of the tile being factored, per column
    __shared__ double dinv[16];                //             1/sqrt(pivot), per column
    __shared__ double zs[256];
    __shared__ double yv[256];
    __shared__ int wsimd[16];
    __shared__ unsigned nzm[2];                // bit I: panel tile (I, j) of the step is not all zero (double-buffered by step parity)
    __shared__ int fail;
    int w = blockIdx.x;
    WinState& st = B.ws[w];
    if (!st.need_lin || st.lin_fail) return;
    const WinRec& W = B.win[w];
    int n = W.n_red, tid = threadIdx.x;
    if (n <= 0 || n > 240) return;                 // larger windows of a mixed batch belong to k_chol_big (launched next to this one)
    int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int Tc = (n + 15) >> 4, Tr = Tc + 1;
    // first tile row / column whose factor is written out: everything, or the parameter_head tail block, or nothing
    const int ef = export_full ? 0 : (W.tail_dim > 0 ? (n - W.tail_dim) >> 4 : Tc);
    double* Lrm = B.L + W.Lt_base;
#ifdef SWF_PROFILE_CHOL
    if (blockIdx.x == 0 && tid == 0) for (int i = 8; i < 12; i++) g_chol_stamps[i] = 0;
    unsigned long long tq = 0;
#endif
    CHSTAMP(0);
    if (tid == 0) { fail = 0; nzm[0] = 0u; nzm[1] = 0u; }
    if (tid < 16) dinv[tid] = 0.0;
    for (int e = tid; e < 256; e += 1024) zs[e] = 0.0;
    // Roles by SIMD.  fp64 MFMAs and fp64 VALU instructions of one SIMD do not overlap, so every trailing-update MFMA issued on the
    // pivot wave's SIMD lengthens the one dependency chain the whole factorisation waits for (tests/microbench/pivot_chain.hip: 211
    // cycles per column alone, ~350 next to three tile waves).  Wave 0 is the pivot wave; the first other wave on its SIMD takes
    // the inverse (light VALU work); the SIMD's remaining waves leave at once unless fewer than 12 waves sit on the other SIMDs.
    if (lane == 0) wsimd[wv] = (int)__builtin_amdgcn_s_getreg(2308);       // HW_REG_HW_ID, SIMD_ID (bits 5:4)
    __syncthreads();                                       // R
    int role, tw, ntw;                                     // role 0 pivot factor, 1 pivot inverse, 2 tile wave (tw of ntw)
    {
        const int ps = wsimd[0];
        const unsigned onp = (unsigned)__ballot(lane >= 1 && lane < 16 && wsimd[lane & 15] == ps) & 0xfffeu;      // other waves on the pivot's SIMD
        const unsigned rbit = onp ? (onp & (0u - onp)) : 2u;                                                     // the inverse wave
        unsigned rest = onp & ~rbit, tiles = 0xfffeu & ~onp & ~rbit;
        for (int need = RR3_MINTW - __popc(tiles); need > 0 && rest; need--) { unsigned b = rest & (0u - rest); tiles |= b; rest &= ~b; }
        const unsigned me = 1u << wv;
        role = wv == 0 ? 0 : (rbit & me) ? 1 : (tiles & me) ? 2 : 3;
        tw = __popc(tiles & (me - 1u)); ntw = __popc(tiles);
    }
    if (role == 3) return;
    if (role == 0) {
        // =============================== pivot wave: the factor of the diagonal tiles ===============================
        __syncthreads();                                   // A_0: tile (0,0) published
        CHSTAMP(3);
        for (int j = 0; j < Tc; j++) {
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            WST(j, 0);
            bool bad = rr3_pivot_factor(Dt[j & 1], colb, dinv, Dl, j >= ef, li, lk);
            WST(j, 1);
            if (bad && lane == 0) fail = 1;
            CHACC(9, tq);
#ifdef SWF_PROFILE_CHOL
            if (blockIdx.x == 0 && lane == 0 && j == SWF_PROFILE_CHOL_STEP) { g_chol_stamps[48] = tq; g_chol_stamps[32] = __builtin_amdgcn_s_memtime(); }
#endif
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                               // B_j
            CHACC(8, tq);
#ifdef SWF_PROFILE_CHOL
            if (blockIdx.x == 0 && lane == 0 && j < 15) g_chol_stamps[49 + j] = __builtin_amdgcn_s_memtime();     // the pivot wave past B_j
#endif
#ifdef SWF_PROFILE_CHOL
            tq = __builtin_amdgcn_s_memtime();
#endif
            WST(j, 2);
            if (fail) { if (tid == 0) { st.lin_fail = 1; st.chol_fail = 1; } return; }
            if (lane == 0) nzm[(j + 1) & 1] = 0u;          // the next step's mask (its last readers left before B_j)
            __syncthreads();                               // C_j
            WST(j, 3);
            CHACC(10, tq);
        }
        CHSTAMP(1);
        __syncthreads();                                   // E: L exported, yv ready
        CHSTAMP(4);
        // backward solve y = L^-T z, right-looking: yv holds z; once y_J is known the tiles of row J subtract
        // L_{J,J'}^T y_J from the pending blocks J' < J.  Here: y_J = Linv_JJ^T yv_J on all 64 lanes
        for (int J = Tc - 1; J >= 0; J--) {
            double p = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) p += Li[J][lk + 4 * q][li] * yv[16 * J + lk + 4 * q];
            atomicAdd(&zs[16 * J + li], p);                // ds_add_f64: the four row groups of lanes add into the (zeroed) slot, no cross-lane round trips
            __syncthreads();                               // X_J: y_J published
            __syncthreads();                               // Y_J: row J applied to the pending blocks
        }
        CHSTAMP(2);
        return;
    }
    if (role == 1) {
        // =============================== pivot wave: the inverse of the diagonal tiles ===============================
        __syncthreads();                                   // A_0
        for (int j = 0; j < Tc; j++) {
            rr3_pivot_inverse(Li[j], colb, dinv, li, lk);
            WST(j, 1);
#ifdef SWF_PROFILE_CHOL
            if (blockIdx.x == 0 && lane == 0 && j == SWF_PROFILE_CHOL_STEP) g_chol_stamps[32 + wv] = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                               // B_j
            if (fail) return;
            if (j >= ef) {
                // L_jj to HBM (wave 0 left it in Dl; nothing overwrites it before C_j)
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
    double* Xs = &Pn[0][0][0] + tw * 272;                  // wave-private transposition scratch (up to 14 x 272 doubles of the panel buffer)
    // this wave's off-diagonal tiles (the rhs tile row Tc included): tile e of the column-order list belongs to tile wave e mod ntw, slot e / ntw,
    // so the panel tiles of a column and the trailing tiles of every step spread evenly over the waves
    int sI[RR3_NS], sJ[RR3_NS];
    {
        int mI = -1, mJ = -1;                               // lane s works out slot s
        if (lane < RR3_NS) {
            int e = lane * ntw + tw, J = 0;
            while (J < Tc && (J + 1) * (Tr - 1) - (J + 1) * J / 2 <= e) J++;
            if (J < Tc) { mI = J + 1 + e - (J * (Tr - 1) - J * (J - 1) / 2); mJ = J; }
        }
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) { sI[s] = __builtin_amdgcn_readlane(mI, s); sJ[s] = __builtin_amdgcn_readlane(mJ, s); }
    }
    CHSTAMP2(16);
    double4_t acc[RR3_NS];
#ifdef SWF_PROFILE_CHOL
    unsigned long long tq2 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 128) for (int i = 22; i < 30; i++) g_chol_stamps[i] = 0;
#endif
    // all loads of the wave's tiles are issued branch-free (clamped addresses, values selected afterwards): S is stored lower;
    // the extra tile row Tc carries the right-hand side in its first row
    // (the tile waves keep -A: every update is then a plain accumulate)
    int lpart[4];
#pragma unroll
    for (int q = 0; q < 4; q++) lpart[q] = (lk + 4 * q) * n + li;
#pragma unroll
    for (int s = 0; s < RR3_NS; s++) {
        const int I = sI[s], J = sJ[s];
        if (I >= 0 && 16 * I + 16 <= n) {
            // tile fully inside the matrix (its columns end before its rows begin): one add per element
            const double* St = S + (16 * I * n + 16 * J);
#pragma unroll
            for (int q = 0; q < 4; q++) acc[s][q] = St[lpart[q]];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int r = 16 * I + lk + 4 * q, c = 16 * J + li;
                bool rhs_el = I == Tc && lk + 4 * q == 0;                 // row n of the S storage = reduced rhs
                int rr = rhs_el ? n : r;
                bool inside = I >= 0 && c < n && (rhs_el || r < n);
                int rc = rr < n ? rr : (rhs_el ? n : n - 1), cc = c < n ? c : n - 1;
                if (I < 0) { rc = 0; cc = 0; }
                double v = S[rc * n + cc];                                // 32-bit element offset from the one S base (rc >= cc: I > J)
                acc[s][q] = inside ? v : 0.0;
            }
        }
    }
    asm volatile("" ::: "memory");                          // (every load above is on its way before the first value is touched)
    // the diagonal tiles go to LDS, fully symmetric (the pivot needs both triangles), identity on the padding; tile J by tile wave J mod ntw
    for (int J = tw; J < Tc; J += ntw) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * J + lk + 4 * q, c = 16 * J + li;
            int rc = r < n ? r : n - 1, cc = c < n ? c : n - 1;
            double v = S[(cc > rc) ? cc * n + rc : rc * n + cc];
            v = (r < n && c < n) ? v : (r == c ? 1.0 : 0.0);
            Dg[J][q][lane] = -v;
            if (J == 0) Dt[0][lk + 4 * q][li] = v;
        }
    }
    CHSTAMP2(17);
    // negated, into the transposed layout
#pragma unroll
    for (int s = 0; s < RR3_NS; s++)
        if (sI[s] >= 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) acc[s][q] = -acc[s][q];
            rr3_transpose(acc[s], Xs, li, lk);
        }
    CHSTAMP2(18);
    __syncthreads();                                       // A_0
    CHSTAMP2(19);
    for (int j = 0; j < Tc; j++) {
        // the pending value of diagonal tile j+1 is final since the panel phase of step j-1 (its last term, from tile (j+1, j-1)): fetch it ahead of the barrier
        bool pair = false;
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) pair = pair || (sJ[s] == j && sI[s] == j + 1 && j + 1 < Tc);
        double4_t dn = { 0, 0, 0, 0 };
        if (pair) {
#pragma unroll
            for (int q = 0; q < 4; q++) dn[q] = Dg[j + 1][q][lane];
        }
#ifdef SWF_PROFILE_CHOL
        if (blockIdx.x == 0 && lane == 0 && j == SWF_PROFILE_CHOL_STEP) g_chol_stamps[32 + wv] = __builtin_amdgcn_s_memtime();
#endif
        CHACC2(28, tq2);
        WST(j - 1, 6);
        __syncthreads();                                   // B_j: Linv_jj ready; trailing updates of step j-1 done
        WST(j - 1, 7);
        WST(j, 0);
        CHACC2(26, tq2);
        if (fail) return;
        // panel row j of U: U_jI = Linv_jj A_jI.  Tile (j+1, j) first, with the diagonal tile j+1 right behind it (published for the pivot).
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) {
            if (sJ[s] != j) continue;
            int I = sI[s];
            asm volatile("" : "+s"(I));                    // (addresses from the scalar tile index at the point of use: no per-slot address registers across the loop)
            // A tile of S that is still exactly zero (no coupling, no fill yet: the banded speed-bias part of a cfg3 window leaves 14 of
            // its 105 tiles like that) stays zero through the panel product and contributes nothing below: it is neither multiplied
            // nor published, and the trailing updates skip every product one of whose operands is such a tile (the step's mask, nzm) —
            // 455 -> 314 tile updates for a cfg3 window, most of them in the first steps, where the updates and not the pivot set the pace.
            const bool tnz = __ballot((acc[s][0] != 0.0) | (acc[s][1] != 0.0) | (acc[s][2] != 0.0) | (acc[s][3] != 0.0)) != 0ull;
            if (!tnz) {
                if (I == j + 1 && j + 1 < Tc) {
#pragma unroll
                    for (int q = 0; q < 4; q++) Dt[(j + 1) & 1][lk + 4 * q][li] = -dn[q];       // diagonal tile j+1 takes nothing from this step
                }
                continue;
            }
            if (lane == 0) atomicOr(&nzm[j & 1], 1u << I);
            double4_t X = { 0, 0, 0, 0 };
#pragma unroll
            for (int kk = 0; kk < 4; kk++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[j][li][lk + 4 * kk], acc[s][kk], X, 0, 0, 0);       // U_jI = Linv (A_jI) = (-Linv) (-A_jI)
            acc[s] = X;
            if (I == j + 1 && j + 1 < Tc) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) dn = __builtin_amdgcn_mfma_f64_16x16x4f64(X[kk], X[kk], dn, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) Dt[(j + 1) & 1][lk + 4 * q][li] = -dn[q];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) Pn[I][kk][lane] = X[kk];
            if (I == j + 2 && I < Tc) {
                // diagonal tile j+2 takes its term here, ahead of the barrier: the next step's (j+2, j+1) owner fetches it before B_j+1
                int I2 = j + 2;
                asm volatile("" : "+s"(I2));
                double4_t d;
#pragma unroll
                for (int q = 0; q < 4; q++) d[q] = Dg[I2][q][lane];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(X[kk], X[kk], d, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) Dg[I2][q][lane] = d[q];
            }
        }
        CHACC2(27, tq2);
        WST(j, 1);
        __syncthreads();                                   // C_j: panel and diagonal tile j+1 published
        WST(j, 2);
        // trailing updates (overlap with the pivot pair's work on tile j+1): the later diagonal tiles take their term of this step
        // from the wave that holds it in registers, -A_II += U_jI^T U_jI; the off-diagonal tiles -A_JI += U_jJ^T U_jI from the panel
        CHACC2(22, tq2);
        const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)nzm[j & 1]);
        CHACC2(23, tq2);
        WST(j, 3);
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) {
            if (sJ[s] != j) continue;
            int I = sI[s];
            if (I <= j + 2 || I >= Tc || !((m >> I) & 1u)) continue;
            asm volatile("" : "+s"(I));
            double4_t d;
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] = Dg[I][q][lane];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[s][kk], acc[s][kk], d, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) Dg[I][q][lane] = d[q];
        }
        CHACC2(24, tq2);
        WST(j, 4);
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) {
            int J = sJ[s], I = sI[s];
            if (J <= j || I < 0 || !((m >> I) & (m >> J) & 1u)) continue;
            asm volatile("" : "+s"(I), "+s"(J));           // (the panel addresses are one scalar add away: not worth a register pair per slot across the loop)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pn[J][kk][lane], Pn[I][kk][lane], acc[s], 0, 0, 0);
        }
        CHACC2(25, tq2);
        WST(j, 5);
    }
    CHSTAMP2(20);
    // back to the row layout (lane (li, lk), register q <-> L[lk+4q][li]); export L where it is read, y = L^-1 rhs from the rhs tile row
#pragma unroll
    for (int s = 0; s < RR3_NS; s++)
        if (sI[s] >= 0) rr3_transpose(acc[s], Xs, li, lk);
#pragma unroll
    for (int s = 0; s < RR3_NS; s++) {
        int I = sI[s], J = sJ[s];
        if (I < 0) continue;
        asm volatile("" : "+s"(I), "+s"(J));               // (recompute the indices here rather than carry the load phase's through the loop)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int r = 16 * I + lk + 4 * q, c = 16 * J + li;
            if (I < Tc) { if (J >= ef && r < n) Lrm[(size_t)r * n + c] = acc[s][q]; }
            else if (lk + 4 * q == 0) yv[c] = acc[s][q];
        }
    }
    CHSTAMP2(21);
    __syncthreads();                                       // E
    for (int J = Tc - 1; J >= 0; J--) {
        __syncthreads();                                   // X_J: y_J published
#pragma unroll
        for (int s = 0; s < RR3_NS; s++) {
            if (sI[s] != J) continue;                      // tiles (J, J') of row J, J' < J: yv_J' -= L_{J,J'}^T y_J
            double p = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) p += acc[s][q] * zs[16 * J + lk + 4 * q];
            atomicAdd(&yv[16 * sJ[s] + li], -p);           // ds_add_f64 (the tiles of row J update distinct blocks J')
        }
        __syncthreads();                                   // Y_J
    }
    double* y = B.y + W.loc_base + W.n_e;
    for (int e = tw * 64 + lane; e < n; e += ntw * 64) y[e] = zs[e];
}
