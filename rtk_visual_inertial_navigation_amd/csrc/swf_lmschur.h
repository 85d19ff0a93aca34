// swf_lmschur.h — landmark Schur elimination (the bulk of elimination group 0), one fused kernel per linearisation.
//
//   H_ll = sum Jl^T Jl + mu clamp(diag) = L L^T,  C = L^-T (upper triangular),  Einv = C C^T
//   Z_o  = Jp_o^T (Jl_o C)          (6 x 3 per observation),   h_l = C^T g_l
//   P    = sum_l Z_l Z_l^T          (the landmark part of the reduced camera matrix: a SYRK, fp64 matrix cores)
//   q    = sum_l Z_l h_l            (= sum_l Y_l g_l, the landmark part of the reduced right-hand side)
//
// In-tree analogue of this arithmetic: MarginalizationInfo::marginalize, R/factor/marginalization_factor.cpp:260-377;
// in Ceres it is SchurEliminator::Eliminate (E^T E inverted by InvertPSDMatrix = Cholesky, F^T E (E^T E)^-1 E^T F).
//
// Unit of work: the WAVE TASK = one producer wavefront's four 16-lane groups = four landmarks (one group each, whatever the track
// length: lane j takes observations j, j + 16, j + 32, j + 48 in as many rounds as the task's longest track needs) = 12 columns =
// three k-steps of the K-packed panel
//   Zp[column][row],  column = 3 * (landmark slot) + coordinate,  row = 6 * frame + i     (k-major, LDR doubles per column)
// which is zero where a landmark does not see a frame.  One workgroup per (window, 1..16 of the window's 16 landmark parts)
// runs a RING of four panel buffers in LDS between two kinds of wavefronts that never meet at a barrier:
//   producers  4 teams of TW waves; team t fills buffer t with chunks t, t + 4, ... (a chunk = TW wave tasks): DPP butterflies
//              for H_ll / g_l, division-free 3x3 Cholesky inverse, Z straight into the panel.  The panel is kept zero
//              incrementally: every lane remembers where it wrote last time and clears exactly that before it writes (a wave
//              owns a fixed column range of its team's buffer, so old and new positions never cross waves).  Loads of the next
//              task are in flight while the current one computes.
//   consumers  NCW waves own TPW 16x16 tiles of the lower triangle of P each (diagonal tiles first in the tile list) and run
//              plain v_mfma_f64_16x16x4_f64 over the k-steps of each chunk: all four k-slots carry data, the operand address of
//              a k-step is tile base + an immediate.  Which wave tasks a tile needs is STATIC (some landmark of the task is seen
//              from the tile's row frames AND its column frames): host-built bit masks, one word per (chunk, wave) — the host
//              also sorts each window's landmarks by tile footprint so that the landmarks of a task share theirs (cfg3: 3.5 k
//              MFMAs per window against 4.4 k with one MFMA per (tile, landmark)).  The wave of diagonal tile (t, t) multiplies
//              its A operands with h on the side: q.
//   hand-off   two LDS counters per buffer: `ready` (producer waves that finished writing) and `done` (consumer waves that
//              finished reading); LDS operations of a wave execute in order, so a counter bump behind the data accesses is all
//              the ordering there is.  Producers run up to four chunks ahead.
// The packing of landmarks into wave tasks, and therefore every sum, is the same in all size classes: a window gets
// bit-identical results whichever class its batch selects.
// f64 MFMA layouts: A[i][k]: lane = i + 16k; B[k][j]: lane = j + 16k; D: lane l, reg q -> row (l>>4)+4q, col l&15.
// Per observation the kernel reads 160 B (Jp, Jl, r) and writes nothing; per landmark it writes Einv, g_l (back-substitution).
#pragma once
typedef double double4_t __attribute__((ext_vector_type(4)));
#define GEMM_SPLIT 16                         // fixed landmark split: partial products P_0..P_15, summed in order (by k_lm_schur
                                              // itself when one block covers them all, else by k_assemble)
#ifdef SWF_PROFILE_GEMM
__device__ unsigned long long g_gemm_stamps[16];
#define GSTAMP_ACC(i, t0) do { if (bx_ == 0 && by_ == 0 && (threadIdx.x & 63) == 0) g_gemm_stamps[i] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#define GNOW() __builtin_amdgcn_s_memtime()
#else
#define GSTAMP_ACC(i, t0)
#define GNOW() 0ULL
#endif
#define LS_NB 4                               // ring buffers = producer teams
#define LS_NT(NCW, TW) ((LS_NB * (TW) + (NCW)) * 64)
#define LS_MAXF 64                            // observing frames per window (64-bit frame masks)
#define LS_SPIN_MAX (1 << 22)                 // bound of the hand-off polls (a broken table must not hang the device)
// size classes (LDR = doubles per panel column, = 16 mod 32: the four k-slots of an operand read fall on distinct banks):
//   <8, 2, 2,  80>   <= 10 frames  (<= 10 tiles)            <8, 5, 2, 144>   <= 21 frames (<= 36 tiles)
//   <12, 6, 1, 272>  <= 42 frames  (<= 136 tiles, 2 launches of 72)    <12, 6, 1, 400>  <= 64 frames (<= 300 tiles, 5 launches)
// GEMM = false: the elimination alone (g_l, diag, Einv for a cost / gradient pass: the solve's last linearisation, whose system
// is never solved): producer waves only, no panel, under its own kernel name.
struct LsRec { int L, loc, o0, o1, col, info; };            // o0 .. o1: the landmark's observations (<= 64); col = first column of the landmark within its wave's twelve (0, 3, 6, 9)
struct LsDat { double jl[6], rr[2]; int f; };
__device__ __forceinline__ void ls_wait(const unsigned* flag, unsigned target, WinState& s) {
    asm volatile("" ::: "memory");
    for (int spin = 0; ; spin++) {
        unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
        if (spin >= LS_SPIN_MAX) { s.lin_fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ls_signal(unsigned* flag) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// (the kernel's body as a device function of the block coordinates: k_lm_schur below, and — latency path — the landmark workgroups of the
// fused elimination grid k_lm_clique in swf_kernels.h)
template <int NCW, int TPW, int TW, int LDR, bool GEMM>
__device__ __forceinline__ void d_lm_schur(const DevBatch& B, const DevOpt& O, int qpb, int lp, int kms, int s_direct, const int bx_, const int by_) {
    constexpr int NPW = GEMM ? LS_NB * TW : 4;                         // producer waves
    constexpr int NCOL = 12 * TW, NG = TW;                             // columns / wave tasks per chunk
    constexpr int PANEL = NCOL * LDR;                                  // doubles per buffer
    constexpr int KSB = 4 * LDR * 8;                                   // bytes per k-step
    static_assert(!GEMM || LS_NB * PANEL < 0xffff, "panel offsets are kept in 16 bits");
    __shared__ double Zp[GEMM ? LS_NB * PANEL + LS_NB * NCOL : 1];     // four panels | h_l at the landmark's columns, per buffer
    __shared__ unsigned flg[2 * LS_NB];                                // ready[4] | done[4]
    __shared__ int freds[LS_MAXF];                                     // first reduced row of every frame's pose (s_direct write-out)
    double* const hb = Zp + (GEMM ? LS_NB * PANEL : 0);
    const int tile_base = lp * NCW * TPW;              // launch lp covers the tile-list entries [tile_base, tile_base + NCW TPW)
    // a block covers qpb consecutive landmark parts of its window (qpb = 1, 2, 4, 8 or 16; a single window spreads over 16
    // workgroups, large batches use 16 so the ring fills once per block).  Every part still gets its own partial product,
    // so the result does not depend on qpb.
    int w = bx_, sp0 = by_ * qpb;
    const bool outs = tile_base == 0;                  // launches of further tile ranges only add their tiles of P: they may run next to the first
    WinState& s = B.ws[w];
    if (!s.need_lin) return;
    const WinRec& W = B.win[w];
    int nF = W.nF, m = 6 * nF, nt = (m + 15) / 16, ntl = nt * (nt + 1) / 2;
    int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool gemm = GEMM && m > 0;
    int blk = w * GEMM_SPLIT + sp0;
    const int t0 = B.sch_c0[blk], t1 = B.sch_c0[blk + qpb];            // wave tasks of this block (even numbers: a chunk never straddles parts)
    if (GEMM) {
        // the panels start all-zero (from then on the producers keep them so)
        for (int e = tid; e < LS_NB * PANEL / 2; e += LS_NT(NCW, TW)) ((double2*)Zp)[e] = double2{ 0.0, 0.0 };
        for (int e = tid; e < LS_NB * NCOL; e += LS_NT(NCW, TW)) hb[e] = 0.0;
        if (tid < 2 * LS_NB) flg[tid] = 0u;
        if (s_direct && tid >= NPW * 64 && tid - NPW * 64 < nF) freds[tid - NPW * 64] = B.fr_red[W.fr_base + tid - NPW * 64];
        __syncthreads();
    }
    if (GEMM && wv >= NPW) {
        // =========================== consumer waves: P += Z Z^T on the matrix cores ===========================
        if (!gemm) return;
        int cw = wv - NPW;
        // the lower-triangle tiles as a list: the nt diagonal tiles first, then (tr > tc) row by row; entry e of the list
        // belongs to wave (e - tile_base) % NCW, slot (e - tile_base) / NCW.  Recomputed at write-out (a register pair per slot spilled).
        auto tile_rc = [&](int e, int& tr, int& tc) {
            if (e < nt) { tr = tc = e; return; }
            int u = e - nt; tr = 1;
            while (tr * (tr + 1) / 2 <= u) tr++;
            tc = u - tr * (tr - 1) / 2;
        };
        constexpr bool CAN_FOLD = TPW <= 5 && NCW == 8;     // these blocks have the registers for the folded product
        constexpr int NDS = LDR <= 144 ? 1 : 2;             // slots that can hold a diagonal tile (the list's first nt entries over NCW waves)
        double4_t acc[TPW], tot[CAN_FOLD ? TPW : 1];
        double qa[NDS], qt[NDS];
        unsigned aoff[TPW], boff[TPW];                      // LDS byte offsets of the lane's A / B operand at k-step 0, buffer 0
        const unsigned laneoff = (unsigned)(lk * LDR + li) * 8u;
#pragma unroll
        for (int sl = 0; sl < TPW; sl++) {
            if (CAN_FOLD) tot[sl] = double4_t{ 0, 0, 0, 0 };
            int e = tile_base + cw + sl * NCW;
            int tr = 0, tc = 0;
            if (e < ntl) tile_rc(e, tr, tc);
            aoff[sl] = laneoff + (unsigned)tr * 128u; boff[sl] = laneoff + (unsigned)tc * 128u;
            acc[sl] = double4_t{ 0, 0, 0, 0 };
        }
#pragma unroll
        for (int d = 0; d < NDS; d++) { qa[d] = 0.0; qt[d] = 0.0; }
        const unsigned hoff = (unsigned)(LS_NB * PANEL * 8) + (unsigned)lk * 8u;            // h of column 4 j + lk: behind the panels
        const char* lds = (const char*)Zp;
        unsigned long long tg = GNOW(); (void)tg;
        // tile masks: one word per (chunk, launch, wave) = the wave's TPW slots x 4 bits (bit g of slot s: the tile needs the three
        // k-steps of wave task g of this chunk)
        const int k0 = t0 / TW, k1 = t1 / TW;                                               // chunks of this block
        // (lane j holds the word of chunk k0 + 64 i + j; the next 64 are requested when a 64-block begins)
        const int* km = B.sch_km + (size_t)lp * NCW + cw;
        auto load_masks = [&](int kb) { return kb + lane < k1 ? (unsigned)km[(size_t)(kb + lane) * kms] : 0u; };
        unsigned mcur = load_masks(k0), mnxt = 0u;
        for (int sq = 0; sq < qpb; sq++) {
        const int kq1 = B.sch_c0[blk + sq + 1] / TW;
        for (int k = (sq == 0 ? k0 : B.sch_c0[blk + sq] / TW); k < kq1; k++) {
            const int r = k - k0, buf = r & (LS_NB - 1), use = r / LS_NB;
            tg = GNOW();
            if ((r & 63) == 0) { if (r) mcur = mnxt; mnxt = load_masks(k + 64); }
#ifdef SWF_LS_NOMFMA
            const unsigned wm = 0u;                        // experiment: consumers idle
#else
            const unsigned wm = (unsigned)__builtin_amdgcn_readlane((int)mcur, r & 63);
#endif
            ls_wait(&flg[buf], (unsigned)(TW * (use + 1)), s);                              // chunk k is in its buffer
            if (cw == 0) GSTAMP_ACC(10, tg);
            tg = GNOW();
            const unsigned pb = (unsigned)buf * (unsigned)(PANEL * 8);
            const unsigned hbo = hoff + (unsigned)buf * (unsigned)(NCOL * 8);
#pragma unroll
            for (int sl = 0; sl < TPW; sl++) {
                // 3 NG bits per slot: bit 3 g + j = k-step j of wave task g carries a landmark this tile needs (round 3 kept one bit per
                // task; a k-step whose landmarks do not touch the tile multiplies zeros)
                const unsigned mg = (wm >> (3 * NG * sl)) & ((1u << (3 * NG)) - 1u);
                if (!mg) continue;
                const bool dg = sl < NDS && tile_base + cw + sl * NCW < nt;   // diagonal tile: B operand = A operand, and the q side product
                const unsigned ao = aoff[sl] + pb, bo = boff[sl] + pb;
                double4_t c_ = acc[sl];
                if (sl < NDS && dg) {
                    double q_ = qa[sl < NDS ? sl : 0];
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (!(mg & (7u << (3 * g)))) continue;
                        // (the task's operands are requested together; the mask decides which k-steps are multiplied)
                        double a0 = *(const double*)(lds + ao + (unsigned)((3 * g + 0) * KSB)), a1 = *(const double*)(lds + ao + (unsigned)((3 * g + 1) * KSB)),
                               a2 = *(const double*)(lds + ao + (unsigned)((3 * g + 2) * KSB));
                        double h0 = *(const double*)(lds + hbo + (unsigned)((3 * g + 0) * 32)), h1 = *(const double*)(lds + hbo + (unsigned)((3 * g + 1) * 32)),
                               h2 = *(const double*)(lds + hbo + (unsigned)((3 * g + 2) * 32));
                        if (mg & (1u << (3 * g + 0))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, c_, 0, 0, 0);
                        if (mg & (1u << (3 * g + 1))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, c_, 0, 0, 0);
                        if (mg & (1u << (3 * g + 2))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, c_, 0, 0, 0);
                        q_ = __builtin_fma(a0, h0, q_); q_ = __builtin_fma(a1, h1, q_); q_ = __builtin_fma(a2, h2, q_);      // (a skipped k-step's rows are zero)
                    }
                    qa[sl < NDS ? sl : 0] = q_;
                } else {
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (!(mg & (7u << (3 * g)))) continue;
                        double a0 = *(const double*)(lds + ao + (unsigned)((3 * g + 0) * KSB)), a1 = *(const double*)(lds + ao + (unsigned)((3 * g + 1) * KSB)),
                               a2 = *(const double*)(lds + ao + (unsigned)((3 * g + 2) * KSB));
                        double b0 = *(const double*)(lds + bo + (unsigned)((3 * g + 0) * KSB)), b1 = *(const double*)(lds + bo + (unsigned)((3 * g + 1) * KSB)),
                               b2 = *(const double*)(lds + bo + (unsigned)((3 * g + 2) * KSB));
                        if (mg & (1u << (3 * g + 0))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c_, 0, 0, 0);
                        if (mg & (1u << (3 * g + 1))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c_, 0, 0, 0);
                        if (mg & (1u << (3 * g + 2))) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, c_, 0, 0, 0);
                    }
                }
                acc[sl] = c_;
            }
            ls_signal(&flg[LS_NB + buf]);                   // this wave is through with the buffer
            if (cw == 0) GSTAMP_ACC(9, tg);
        }
        // end of a part.  A block that covers all GEMM_SPLIT parts folds them in registers in exactly the order
        // k_assemble adds partials, ((P0 + P1) + P2) + ... with every Pq summed from zero, and writes ONE product
        // (slot 0; k_assemble is told to read one partial): same bits, 1/GEMM_SPLIT of the P traffic.  Otherwise
        // each part's partial product is flushed to its own slot.  q likewise (its k-slot partials first, butterfly).
        bool fold = CAN_FOLD && qpb == GEMM_SPLIT;
        tg = GNOW();
        if (outs) {
            double* Q = B.lmq + (size_t)6 * W.fr_base * GEMM_SPLIT + (size_t)(fold ? 0 : sp0 + sq) * m;
#pragma unroll
            for (int d = 0; d < NDS; d++) {
                double v = qa[d];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                qa[d] = 0.0;
                if (fold) { qt[d] = sq == 0 ? v : qt[d] + v; v = qt[d]; }
                int e = cw + d * NCW, row = e * 16 + li;
                if ((!fold || sq + 1 == qpb) && e < nt && lk == 0 && row < m) Q[row] = v;
            }
        }
        if (fold) {
#pragma unroll
            for (int sl = 0; sl < TPW; sl++) { tot[CAN_FOLD ? sl : 0] = sq == 0 ? acc[sl] : tot[CAN_FOLD ? sl : 0] + acc[sl]; acc[sl] = double4_t{ 0, 0, 0, 0 }; }
            if (sq + 1 < qpb) continue;
        }
        double* P = B.P + W.P_base * GEMM_SPLIT + (size_t)(fold ? 0 : sp0 + sq) * m * m;
        if (fold && s_direct) {
            // large batches: the folded product goes straight to where it ends up, S_pp = -P in the reduced system's own order
            // (k_assemble_flat then adds the few other contributions on top and never touches a frame pair that has none — 171 of the
            // 190 pose pairs of a cfg3 window).  -P + c == c - P bit for bit, so the result is that of the P route.
            double* Sw = B.S + W.S_base; const int nr = W.n_red; const int* fred = freds;
#pragma unroll
            for (int sl = 0; sl < TPW; sl++) {
                int e = tile_base + cw + sl * NCW;
                if (e < ntl) {
                    int tr, tc; tile_rc(e, tr, tc);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        int rr_ = tr * 16 + lk + 4 * q, cc = tc * 16 + li;
                        if (rr_ < m && cc < m && rr_ >= cc) {
                            int fa = rr_ / 6, fb = cc / 6;
                            int row = fred[fa] + (rr_ - 6 * fa), col = fred[fb] + (cc - 6 * fb);
                            if (row < col) { int tt = row; row = col; col = tt; }
                            Sw[(size_t)row * nr + col] = -tot[CAN_FOLD ? sl : 0][q];
                        }
                    }
                }
                acc[sl] = double4_t{ 0, 0, 0, 0 };
            }
            if (cw == 0) GSTAMP_ACC(11, tg);
            continue;
        }
#pragma unroll
        for (int sl = 0; sl < TPW; sl++) {
            int e = tile_base + cw + sl * NCW;
            if (e < ntl) {
                int tr, tc; tile_rc(e, tr, tc);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int rr_ = tr * 16 + lk + 4 * q, cc = tc * 16 + li;
                    if (rr_ < m && cc < m) P[(size_t)rr_ * m + cc] = fold ? tot[CAN_FOLD ? sl : 0][q] : acc[sl][q];
                }
            }
            acc[sl] = double4_t{ 0, 0, 0, 0 };
        }
        }
        return;
    }
    // =========================== producer waves: eliminate the landmarks, wave task by wave task ===========================
    // One 16-lane group per landmark (tracks of 17..32 / 33..64 observations take 2 / 4 adjacent groups and merge their sums
    // with one / two more butterfly steps).  The host-built record of (task, group) makes the addressing one level deep.
    // With the ring (GEMM): wave pw belongs to team pw / TW and takes the tasks (k0 + team + 4 i) TW + pw % TW of the block;
    // without: wave pw takes the tasks t0 + pw + 4 i.
    {
    // The arithmetic below is written with explicit fma() under contract(off): the kernel is instantiated per size class,
    // and a window must get bit-identical panels whichever instantiation its batch selects.
#pragma clang fp contract(off)
    const int grp = lane >> 4, sub = lane & 15;
    const int n = B.n_proj, nl = B.n_lm;
    const double mu = s.mu;
#ifdef SWF_PROFILE_GEMM
    if (bx_ == 0 && by_ == 0 && tid == 0) for (int i = 0; i < 16; i++) g_gemm_stamps[i] = 0;
#endif
    unsigned long long tg = GNOW(), tall = tg; (void)tg; (void)tall;
    const int team = GEMM ? wv / TW : 0, mi = GEMM ? wv % TW : 0;
    const int tstep = GEMM ? LS_NB * TW : 4;
    int task = GEMM ? (t0 / TW + team) * TW + mi : t0 + wv;
    auto load_rec = [&](int t) {
        LsRec r; r.L = -1; r.loc = -1; r.o0 = r.o1 = 0; r.col = 0; r.info = 0;
        if (t < t1) {
            const int4* q = (const int4*)(B.sch_rec + ((size_t)t * 4 + grp) * 8);
            int4 a = q[0], b = q[1];
            r.L = a.x; r.loc = a.y; r.o0 = a.z; r.o1 = a.w; r.col = b.x; r.info = b.z;
        }
        return r;
    };
    // round rd of a track: lane `sub` takes observation o0 + 16 rd + sub (a landmark is ONE 16-lane group whatever its track length;
    // tracks beyond 16 observations take further rounds of the same lanes: tile-footprint order puts them into the same tasks)
    auto load_dat = [&](const LsRec& r, LsDat& d, int rd) {
#pragma unroll
        for (int k = 0; k < 6; k++) d.jl[k] = 0.0;
        d.rr[0] = d.rr[1] = 0.0; d.f = -1;
        int o = r.o0 + 16 * rd + sub;
        if (r.L >= 0 && r.loc >= 0 && o < r.o1) {
            const double* pl = B.p_Jl + o;
#pragma unroll
            for (int k = 0; k < 6; k++) d.jl[k] = pl[(size_t)k * n];
            d.rr[0] = B.p_r[o]; d.rr[1] = B.p_r[(size_t)n + o];
            d.f = B.p_fr[o];
        }
    };
    unsigned oldp01 = 0xffffffffu, oldp23 = 0xffffffffu;    // where this lane wrote its Z blocks (one per round) last time: four 16-bit offsets in doubles (0xffff = nowhere)
    int nold = 0;                                           // rounds the wave wrote last time (uniform)
    int use = 0;
    // one wave task: `cur` (Jl, r, frame of round 0) was requested a task ago; the next task's and the record after that are requested
    // first thing, then this task's Jp (it lands during the sums and the inverse)
    auto run_task = [&](const LsRec& rc, LsDat& cur, const LsRec& rn, LsDat& nxt, LsRec& rnn) {
        tg = GNOW();
        // land this task's data here (everything requested a task ago), then send the next requests on their way
#pragma unroll
        for (int k = 0; k < 6; k++) asm volatile("" : "+v"(cur.jl[k]));
        asm volatile("" : "+v"(cur.rr[0]), "+v"(cur.rr[1]), "+v"(cur.f));
        load_dat(rn, nxt, 0);
        rnn = load_rec(task + 2 * tstep);
        const int L = rc.L, loc = rc.loc, o = rc.o0 + sub, kobs = rc.o1 - rc.o0;
#ifdef SWF_LS_NOPROD
        const bool act = false, has = false;               // experiment: producers idle
#else
        const bool act = L >= 0 && loc >= 0, has = act && o < rc.o1;
#endif
        // rounds of this wave (uniform): 1 unless some track of the task has more than 16 observations
        const int nrd = !__any(act && kobs > 16) ? 1 : !__any(act && kobs > 32) ? 2 : !__any(act && kobs > 48) ? 3 : 4;
        const double* a = cur.jl;
        double jp[12];
#pragma unroll
        for (int k = 0; k < 12; k++) jp[k] = 0.0;
        if (GEMM && has) {
            // the translation half of Jp is -Jl bit for bit (k_eval_ps does not store it next to a variable landmark): six loads, not twelve
            const double* pj = B.p_Jp + o;
#pragma unroll
            for (int k = 0; k < 3; k++) { jp[k] = -a[k]; jp[6 + k] = -a[3 + k]; jp[3 + k] = pj[(size_t)(3 + k) * n]; jp[9 + k] = pj[(size_t)(9 + k) * n]; }
        }
#define FMA2(x0, y0, x1, y1) __builtin_fma(x0, y0, (x1) * (y1))
#define FMA3(x0, y0, x1, y1, x2, y2) __builtin_fma(x0, y0, __builtin_fma(x1, y1, (x2) * (y2)))
        double h00 = FMA2(a[0], a[0], a[3], a[3]), h10 = FMA2(a[1], a[0], a[4], a[3]), h20 = FMA2(a[2], a[0], a[5], a[3]);
        double h11 = FMA2(a[1], a[1], a[4], a[4]), h21 = FMA2(a[2], a[1], a[5], a[4]), h22 = FMA2(a[2], a[2], a[5], a[5]);
        double g0 = FMA2(a[0], cur.rr[0], a[3], cur.rr[1]), g1 = FMA2(a[1], cur.rr[0], a[4], cur.rr[1]), g2 = FMA2(a[2], cur.rr[0], a[5], cur.rr[1]);
        for (int rd = 1; rd < nrd; rd++) {                  // further rounds: each lane adds its later observations to its own sums, in round order
            LsDat d2;
            load_dat(rc, d2, rd);
            const double* b = d2.jl;
            h00 += FMA2(b[0], b[0], b[3], b[3]); h10 += FMA2(b[1], b[0], b[4], b[3]); h20 += FMA2(b[2], b[0], b[5], b[3]);
            h11 += FMA2(b[1], b[1], b[4], b[4]); h21 += FMA2(b[2], b[1], b[5], b[4]); h22 += FMA2(b[2], b[2], b[5], b[5]);
            g0 += FMA2(b[0], d2.rr[0], b[3], d2.rr[1]); g1 += FMA2(b[1], d2.rr[0], b[4], d2.rr[1]); g2 += FMA2(b[2], d2.rr[0], b[5], d2.rr[1]);
        }
        h00 = grp16_sum(h00); h10 = grp16_sum(h10); h20 = grp16_sum(h20); h11 = grp16_sum(h11); h21 = grp16_sum(h21); h22 = grp16_sum(h22);
        g0 = grp16_sum(g0); g1 = grp16_sum(g1); g2 = grp16_sum(g2);
        if (wv == 0) GSTAMP_ACC(0, tg);
        tg = GNOW();
        const bool lead = act && sub == 0;
        if (lead && outs) {
            B.g[loc] = g0; B.g[loc + 1] = g1; B.g[loc + 2] = g2;
            B.diag[loc] = h00; B.diag[loc + 1] = h11; B.diag[loc + 2] = h22;
            // D^-2 g for the Cauchy direction: reciprocal by v_rcp_f64 + Newton (2 ulp) instead of three IEEE divisions (~12 instructions each, issued by the whole wave)
            B.vc[loc] = g0 * rcp_nr(clampd(h00, O.min_diag, O.max_diag)); B.vc[loc + 1] = g1 * rcp_nr(clampd(h11, O.min_diag, O.max_diag)); B.vc[loc + 2] = g2 * rcp_nr(clampd(h22, O.min_diag, O.max_diag));
        }
        double i00 = 0, i11 = 0, i22 = 0, i10 = 0, i20 = 0, i21 = 0;
        if (act) {
            const bool jfirst = s.iter == 0;
            h00 = __builtin_fma(mu, damp_diag(O, h00, B.jsc + loc, jfirst), h00);
            h11 = __builtin_fma(mu, damp_diag(O, h11, B.jsc + loc + 1, jfirst), h11);
            h22 = __builtin_fma(mu, damp_diag(O, h22, B.jsc + loc + 2, jfirst), h22);
            // Cholesky inverse of the 3x3 (ceres InvertPSDMatrix), division-free: the reciprocal pivots
            // come from v_rsq_f64 + Newton steps (the IEEE fp64 sqrt/div expansions are instruction-bound)
            i00 = rsqrt_nr(h00);
            double l10 = h10 * i00, l20 = h20 * i00;
            double d11 = __builtin_fma(-l10, l10, h11);
            i11 = rsqrt_nr(d11);
            double l21 = __builtin_fma(-l20, l10, h21) * i11;
            double d22 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, h22));
            i22 = rsqrt_nr(d22);
            bool bad = !(h00 > 0.0) || !(d11 > 0.0) || !(d22 > 0.0);
            if (bad) { if (lead && outs) s.lin_fail = 1; i00 = i11 = i22 = 0.0; }
            // L^-1 = [i00 0 0; i10 i11 0; i20 i21 i22];  C = L^-T
            i10 = -l10 * i00 * i11;
            i21 = -l21 * i11 * i22;
            i20 = -FMA2(l20, i00, l21, i10) * i22;
            if (lead && outs) {
                double e00 = FMA3(i00, i00, i10, i10, i20, i20), e10 = FMA2(i10, i11, i20, i21), e20 = i20 * i22;
                double e11 = FMA2(i11, i11, i21, i21), e21 = i21 * i22, e22 = i22 * i22;
                B.lm_Einv[0 * nl + L] = e00; B.lm_Einv[1 * nl + L] = e10; B.lm_Einv[2 * nl + L] = e20;
                B.lm_Einv[3 * nl + L] = e11; B.lm_Einv[4 * nl + L] = e21; B.lm_Einv[5 * nl + L] = e22;
                B.lm_g[0 * nl + L] = g0; B.lm_g[1 * nl + L] = g1; B.lm_g[2 * nl + L] = g2;
            }
        }
        if (gemm) {
            if (wv == 0) GSTAMP_ACC(1, tg);
            tg = GNOW();
            ls_wait(&flg[LS_NB + team], (unsigned)(NCW * use), s);               // every consumer wave is through with the buffer's previous chunk
            if (wv == 0) GSTAMP_ACC(2, tg);
            tg = GNOW();
            // the panel stays zero outside the live Z blocks: clear what this lane wrote here last time, then write
            for (int rd = 0; rd < nold; rd++) {
                const unsigned op = ((rd < 2 ? oldp01 : oldp23) >> (16 * (rd & 1))) & 0xffffu;
                if (op != 0xffffu) {
                    double* zo = Zp + op;
#pragma unroll
                    for (int cc = 0; cc < 3; cc++)
#pragma unroll
                        for (int i = 0; i < 6; i++) zo[cc * LDR + i] = 0.0;
                }
            }
            oldp01 = oldp23 = 0xffffffffu;
            // per observation: Z = Jp^T (Jl C), column by column straight into the panel
            auto write_z = [&](const double* jl_, const double* jp_, int f_) {
                const int np = team * PANEL + (12 * mi + rc.col) * LDR + 6 * f_;
                double* zn = Zp + np;
                {
                    const double q0 = jl_[0] * i00, q1 = jl_[3] * i00;
#pragma unroll
                    for (int i = 0; i < 6; i++) zn[i] = FMA2(jp_[i], q0, jp_[6 + i], q1);
                }
                {
                    const double q0 = FMA2(jl_[0], i10, jl_[1], i11), q1 = FMA2(jl_[3], i10, jl_[4], i11);
#pragma unroll
                    for (int i = 0; i < 6; i++) zn[LDR + i] = FMA2(jp_[i], q0, jp_[6 + i], q1);
                }
                {
                    const double q0 = FMA3(jl_[0], i20, jl_[1], i21, jl_[2], i22), q1 = FMA3(jl_[3], i20, jl_[4], i21, jl_[5], i22);
#pragma unroll
                    for (int i = 0; i < 6; i++) zn[2 * LDR + i] = FMA2(jp_[i], q0, jp_[6 + i], q1);
                }
                return np;
            };
            if (has && cur.f >= 0) oldp01 = (oldp01 & 0xffff0000u) | (unsigned)write_z(a, jp, cur.f);
            for (int rd = 1; rd < nrd; rd++) {              // later rounds: Jl, frame and Jp of the lane's next observation come from L2 again
                LsDat d2;
                load_dat(rc, d2, rd);
                const int o2 = rc.o0 + 16 * rd + sub;
                double jq[12];
#pragma unroll
                for (int k = 0; k < 12; k++) jq[k] = 0.0;
                const bool has2 = act && o2 < rc.o1;
                if (has2) {
                    const double* pj = B.p_Jp + o2;
#pragma unroll
                    for (int k = 0; k < 3; k++) { jq[k] = -d2.jl[k]; jq[6 + k] = -d2.jl[3 + k]; jq[3 + k] = pj[(size_t)(3 + k) * n]; jq[9 + k] = pj[(size_t)(9 + k) * n]; }
                }
                unsigned np = 0xffffu;
                if (has2 && d2.f >= 0) np = (unsigned)write_z(d2.jl, jq, d2.f);
                if (rd == 1) oldp01 = (oldp01 & 0xffffu) | (np << 16); else if (rd == 2) oldp23 = (oldp23 & 0xffff0000u) | np; else oldp23 = (oldp23 & 0xffffu) | (np << 16);
            }
            nold = nrd;
            if (lead) {
                // h = C^T g_l = L^-1 g_l
                double* hq = hb + team * NCOL + 12 * mi + rc.col;
                hq[0] = i00 * g0; hq[1] = FMA2(i10, g0, i11, g1); hq[2] = FMA3(i20, g0, i21, g1, i22, g2);
            }
            ls_signal(&flg[team]);
            use++;
            if (wv == 0) GSTAMP_ACC(3, tg);
        }
#undef FMA2
#undef FMA3
        task += tstep;
    };
    if (!GEMM || gemm) {
        LsRec rc = load_rec(task), rn = load_rec(task + tstep), rnn;
        LsDat da, db;
        load_dat(rc, da, 0);
        // two tasks per round, the data sets swapping roles (no register moves of loads in flight)
        while (task < t1) {
            run_task(rc, da, rn, db, rnn);
            rc = rn; rn = rnn;
            if (task >= t1) break;
            run_task(rc, db, rn, da, rnn);
            rc = rn; rn = rnn;
        }
    }
    if (wv == 0) GSTAMP_ACC(6, tall);
    }
}
template <int NCW, int TPW, int TW, int LDR, bool GEMM>
__global__ void __launch_bounds__(GEMM ? LS_NT(NCW, TW) : 256) k_lm_schur(DevBatch B, DevOpt O, int qpb, int lp, int kms, int s_direct) {
    d_lm_schur<NCW, TPW, TW, LDR, GEMM>(B, O, qpb, lp, kms, s_direct, (int)blockIdx.x, (int)blockIdx.y);
}
