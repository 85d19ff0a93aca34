// swf_dev.h — device-side data model of the batch engine (gfx950 / CDNA4 only).
//
// One batch = W independent sliding windows laid out back to back in HBM.  Everything that
// depends only on the windows' STRUCTURE (who observes what, elimination order, which
// reduced block pairs are non-zero) is built once on the host ("symbolic phase") into flat
// index arrays; per solve only the parameter blocks change.  All accumulation orders are
// fixed by these arrays — no floating-point atomics anywhere — so results are run-to-run
// bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/swf_types.h"

#define SWF_WAVE 64

// ---- per-window static record -----------------------------------------------------------
struct WinRec {
    int x_base, x_n;             // ambient state range
    int blk_base, n_blk;         // block table range
    int loc_base, n_loc, n_e, n_red;   // local vector: [eliminated dims | reduced dims]
    long long S_base;            // offset of this window's n_red x n_red reduced matrix S
    long long Lt_base;           // offset of the (n_red+1)^2 transposed Cholesky factor (+ rhs row)
    int proj0, proj1;            // projection observations (sorted by landmark, then frame)
    int lm0, lm1;                // landmark records
    int fr_base, nF;             // observing frames (pose blocks that carry observations)
    long long P_base;            // (6 nF)^2 landmark Schur product
    int gf0, gf1;                // generic (non-projection) factors
    int cl0, cl1;                // cliques
    int pair0, pair1;            // reduced block pairs
    int fsb0, fsb1;              // frame-sum blocks of this window
    int tail_dim;                // dimensions of the parameter_head tail (last rows of the reduced system): the block of L its consumers read
    int n_pose_blk;              // the window's first n_pose_blk blocks are its pose blocks (7 -> 6, PoseLocalParameterization)
    int lmb0, lmb1;              // landmark back-substitution blocks of this window (DevBatch::lmb_rec)
    int pch0, pch1;              // row chunks of this window's priors (DevBatch::pch_q / pch_r0)
    double proj_sqrt_info, proj_loss_a;
    double pbg[3], gw[3], base[3];
};

// ---- per-window mutable solver state (lives on the device for the whole solve) ----------
struct WinState {
    double radius, mu, x_cost, x_norm, alpha, dogleg_step_norm, step_norm, gmax;
    double jg_sq, initial_cost;
    double lm_dec;               // LevenbergMarquardtStrategy::decrease_factor_ (SWF_LEVENBERG_MARQUARDT only)
    double model_cost_change;    // of the step k_dogleg proposed (from vectors alone, see k_dogleg)
    int status, iter, need_lin, reuse, eval_cand, lin_fail, invalid_run, nsucc, nunsucc;
    int chol_fail;               // lin_fail came from the dense factorisation (S itself is valid): the marginalisation consumer can still work
};

// ---- generic factor (everything except projection) --------------------------------------
enum { GF_IMU = 1, GF_CP = 2, GF_PR = 3, GF_DOP = 4, GF_SP = 5, GF_PRIOR = 6, GF_SPR = 7, GF_SCP = 8, GF_FIX = 9, GF_IDP = 10,
       GF_PROJX = 11 };      // GF_PROJX: a world-point projection factor on the generic clique path (variable extrinsic, or a landmark outside group 0)
struct GFac {
    int type, win, nres, nslot;
    int slot0;                   // into slot arrays
    int roff;                    // into g_r
    int data;                    // index into the type's data array (record index)
    int clique;                  // owning clique
    int jld;                     // column stride of this factor's Jacobian blocks in g_J (= rows of the clique's dense column-major matrix)
    int pad;                     // GF_PROJX: the caller's projection-factor index (row order of the Jacobian export)
};

// ---- clique: a group-0 block (or none) + the factors touching it + its reduced neighbours
struct Clique {
    int win;
    int e_loc, d_e;              // local offset / size of the eliminated block (d_e = 0: none)
    int d_f;                     // sum of member local sizes
    int fac0, fac1;              // into clique factor list (generic factor ids)
    int mem0, mem1;              // into member arrays (loc offset, size, column)
    long long C_off;             // d_f x d_f Schur'd block (static for priors)
    int v_off;                   // d_f vectors: graw, dgraw, cs
    int e_off;                   // d_e*d_e Einv + d_e*d_f strip + d_e g_e   (offset into e-buffer)
    int is_static;               // 1: prior clique, C/dgraw precomputed; only graw changes
    int n_rows;                  // total residual rows of the clique's factors
    int j_off, r_off;            // the clique's dense column-major Jacobian [d_e + d_f][n_rows] in g_J, and its residual rows in g_r
    int pad0, pad1;
};

// ---- reduced block pair (a >= b in elimination order): one wave assembles S[a,b] --------
struct Pair {
    int win;
    int ra, rb, la, lb;          // reduced offsets and sizes
    int fa, fb;                  // frame slots (>= 0 if pose carries observations) else -1
    int c0, c1;                  // contribution list range
    int is_diag;
    int loc_a;                   // local offset of block a (for g/diag/rhs of diagonal pairs)
    int fsb0, fsb1;              // the window's frame-sum blocks (diagonal pairs of observing poses)
    int n, m;                    // the window's n_red and 6 * nF, and its S / P slabs: the record is self-contained,
    long long S_base, P_base;    // no load of the pair's data depends on a second record
    long long q_base;            // the window's landmark rhs partials in DevBatch::lmq (GEMM_SPLIT vectors of m doubles)
};

// ---- assembly program of a window (k_assemble_flat): one thread per entry of the reduced system that receives anything, one
// per reduced dimension for the vectors.  The tables hold window-RELATIVE offsets, so windows of identical structure (a batch
// of one configuration) share one copy, which then lives in L2.
struct AsmWin {
    int se0, ne;                 // S-entry range in the shared tables
    int ve0, nv;                 // vector-entry range
    int n_red, m;                // reduced dimension; 6 nF
    int loc_base;                // first local dimension of the window's REDUCED part (g / diag / vc / rhs / jsc)
    int fs_base;                 // first frame-sum row of the window (fs_part rows of 27 doubles)
    int v_base;                  // first clique-vector slot of the window (cv_graw / cv_dgraw / cv_cs)
    int win;
    long long C_base, S_base, P_base, q_base;
};
#define AS_NC(c) ((c) & 4095u)
#define AS_NP(c) (((c) >> 12) & 31u)
#define AS_NH(c) (((c) >> 17) & 4095u)
#define AS_DIAG(c) (((c) >> 29) & 1u)
#define AS_SOLD(c) (((c) >> 30) & 1u)

struct DevBatch {
    int n_win;
    int n_x, n_loc_total;
    int max_iter_trace;
    // state
    double* x; double* xc; double* x0;
    // local-space vectors
    double* g; double* diag; double* rhs; double* y; double* step;
    double* vc;                                // D^-2 g = g / clamp(diag): written next to g / diag by their producers (Cauchy direction)
    // reduced matrices
    double* S; double* L;
    double* Wk;                                // working copy of the trailing matrix for k_chol_col (latency path, max n_red > 240), laid out like L
    double* Linv;                              // inverse diagonal tiles of k_chol_big: [window][32][16][16] (allocated when max n_red > 240)
    // tables
    const WinRec* win; WinState* ws; swf_iteration* trace;
    const int* blk_xoff; const int* blk_loc; const int* blk_gs;
    const int* loc2x;                          // per local dimension: its ambient coordinate (absolute), -1 for the dimensions of a pose block
    const unsigned char* x_var;                // per ambient coordinate: 1 if its block is variable
    // projection observations (SoA outputs, stride n_proj)
    int n_proj;
    const int* p_win; const int* p_xpose; const int* p_xex; const int* p_xlm;
    const int* p_lpose; const int* p_llm; const int* p_fr; const int* p_lm;
    const double* p_uv;
    double* p_r; double* p_Jp; double* p_Jl;
    // block partial sums (round 6): the cost of a frame-sum block's observations (written by every projection evaluation, one value per
    // block) and |J D^-2 g|^2 of a landmark back-substitution block's observations (k_post_chol).  The per-window control kernels add
    // a dozen block values in block order instead of reading 16 B per observation; per-observation costs are not stored any more.
    double* p_cpart; double* p_apart;
    // two-level per-frame sums: blocks of <= 256 observations of one window
    int n_fsb; const int* fsb_win; const int* fsb_obs0; const int* fsb_perm; const int* fsb_foff; const int* fsb_foff0; const int* fsb_out0;
    double* fs_part;
    double* jsc;                 // [n_loc_total] Jacobi scaling of the solve's first linearisation, (1 + sqrt(diag))^2 (Solver::Options::jacobi_scaling)
    // landmarks
    int n_lm;
    const int* lm_win; const int* lm_obs0; const int* lm_loc; const int* lm_col;
    double* lm_Einv; double* lm_g;             // SoA stride n_lm: 6 / 3
    double* P;                                 // landmark Schur product, GEMM_SPLIT partials per window
    double* lmq;                               // landmark part of the reduced rhs, sum_l Y_l g_l per pose row: GEMM_SPLIT partial vectors of 6 nF per window (at 6 fr_base GEMM_SPLIT)
    const int* lmb_rec; int n_lmb;             // landmark back-substitution blocks: {first observation, observations (<= 256), first landmark, landmarks}, whole landmarks of one window
    const int* sch_c0; const int* sch_rec;     // k_lm_schur chunk table: chunks of block (window, split); 8-int record per (chunk, group)
    const int* sch_km;                         // k_lm_schur tile masks: word (chunk, launch, consumer wave) = 3 TW bits per tile slot of the wave, bit 3 g + j = k-step j of wave task g is needed
    const unsigned long long* lm_fmask;        // frames (slots < 64) each landmark is observed in
    // frames
    int n_fr;
    const int* fr_obs0; const int* fr_obs;     // CSR of observations per frame
    const int* fr_red;                         // per frame slot: offset of its pose block in the reduced system
    // generic factors
    int n_gf;
    const GFac* gf;
    const int* s_x; const int* s_loc; const int* s_ls; const int* s_joff; const int* s_ccol;   // per slot: Jacobian block offset in g_J (column stride: GFac.jld)
    double* g_r; double* g_J; double* g_cost; double* g_aux;
    const double* imu_pre; const double* cp_dat; const double* pr_dat; const double* dop_dat; const double* sp_w;
    const double* gx_dat;            // records of the rover-only / fixed-integer scalar factors (GFac.data = offset in doubles)
    int n_imu; const int* imu_gf;              // generic-factor ids by kernel
    int n_idp; const int* idp_gf;              // two-row projection factors of the generic path: inverse-depth (GF_IDP) and world-point (GF_PROJX) ones (also members of sc_gf for the J v products)
    int n_sc;  const int* sc_gf;
    int n_prior; const int* prior_gf;
    // priors.  A prior of more than PRIOR_SPLIT_DIM rows is evaluated by one workgroup per chunk of PRIOR_CHUNK rows (its n x n record is
    // n^2 doubles through ONE compute unit otherwise: 553 KB, 19 us per pass, for the 263-dimension marginalisation prior of BASELINE
    // config 5); smaller ones are one chunk.  pch_q / pch_r0: prior and first row of every chunk; pr_cpart / pr_apart: the chunk's cost and
    // its share of |J D^-2 g|^2, which the per-window control kernels add in chunk order (a prior's generic cost slot stays zero).
    int n_pch; const int* pch_q; const int* pch_r0; const int* prior_nch;
    double* pr_cpart; double* pr_apart;
    const int* prior_dim; const long long* prior_Joff; const int* prior_roff; const int* prior_x0off;
    const double* prior_J; const double* prior_r0; const double* prior_x0;
    const double* prior_Jt;          // the same records transposed (element (k, c) at c * n + k): the J v products read these, lanes over rows
    const int* prior_colloc;         // per prior column (at prior_roff + c): reduced-local index of the column's variable, -1 if constant
    const int* prior_colcc;          // likewise: the column's position in the prior clique's vectors, -1 if constant
    const int* s_pcol; const int* s_pxo;   // prior slots only: first column of the block / its offset in the record's x0
    // cliques
    int n_cl;
    const Clique* cl;
    const int* cl_fac; const int* cl_frow;     // factor ids and their first row in the clique Jacobian
    const int* cm_loc; const int* cm_ls; const int* cm_col;
    double* C; double* cv_graw; double* cv_dgraw; double* cv_cs; double* cE;
    const int* cv_loc;                         // per clique vector slot (v_off + c): local index of the variable behind column c of the clique's reduced part
    int n_clc[5]; const Clique* clc_rec[5];    // non-static cliques by size class (copies of the records: no index indirection); 3 = k_clique_big, 4 = k_clique_tall
    int n_cle; const Clique* cle_rec;          // cliques with an eliminated block (back-substitution), likewise
    // pairs
    int n_pair;
    const long long* pc_coff; const int* pc_cld; const int* pc_voff;
    int n_pd, n_po; const Pair* pair_d; const Pair* pair_o;   // diagonal / off-diagonal pair records (the latter sorted by size)
    // assembly programs (k_assemble_flat)
    const AsmWin* asw; int as_max_ne, as_max_nv;
    const int* as_dst; const unsigned* as_cnt; const int* as_src0; const int* as_aux; const int* as_src;
    const int* av_loc; const int* av_red; const unsigned* av_cnt; const int* av_src0; const int* av_i; const int* av_src;
    const unsigned* s_tnz;                     // per window, 4 words: bit I (I - 1) / 2 + J = some block pair reaches tile (I, J), I > J, of the reduced matrix (k_chol_rr4 loads only those; n_red <= 256)
    int spec;                    // this launch evaluates Jacobians at the CANDIDATE of the windows with a proposed step (swf_kernels.h: eval_gate / eval_src); 0 in the batch the engine keeps
    int rr_nmax;                 // reduced systems up to this size take the register-resident Cholesky (256: k_chol_rr4); the streamed kernels take the rest
};

// ------------------------------------------------------------------ device math
// quaternions (x, y, z, w), Hamilton product — same formulas as Eigen, which the reference
// uses (R/factor/*.cpp); see oracle/swf_oracle.c for the CPU restatement.
__device__ __forceinline__ void qmul(const double* a, const double* b, double* o) {
    double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
__device__ __forceinline__ void qinv(const double* q, double* o) {
    double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2;
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void qrot(const double* q, const double* v, double* o) {
    double uv[3], t[3];
    cross3(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(q, uv, t);
    o[0] = v[0] + q[3] * uv[0] + t[0];
    o[1] = v[1] + q[3] * uv[1] + t[1];
    o[2] = v[2] + q[3] * uv[2] + t[2];
}
__device__ __forceinline__ void q2R(const double* q, double* R) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void skew3(const double* v, double* S) {
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}
__device__ __forceinline__ void mat3mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3T(const double* A, double* T) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
}
__device__ __forceinline__ void mat3vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void qleft_br(const double* q, double* M) {
    skew3(q, M);
    M[0] += q[3]; M[4] += q[3]; M[8] += q[3];
}
__device__ __forceinline__ void qleft_qright_br(const double* a, const double* b, double* M) {
    double L[9], Rr[9], S[9];
    qleft_br(a, L);
    skew3(b, S);
#pragma unroll
    for (int i = 0; i < 9; i++) Rr[i] = -S[i];
    Rr[0] += b[3]; Rr[4] += b[3]; Rr[8] += b[3];
    mat3mul(L, Rr, M);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[i * 3 + j] += -a[i] * b[j];
}
// PoseLocalParameterization::Plus (R/factor/pose_local_parameterization.cpp:5-19)
__device__ __forceinline__ void pose_plus(const double* x, const double* d, double* o) {
    o[0] = x[0] + d[0]; o[1] = x[1] + d[1]; o[2] = x[2] + d[2];
    double dq[4] = { d[3] / 2, d[4] / 2, d[5] / 2, 1.0 }, q[4];
    qmul(x + 3, dq, q);
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[3] = q[0] / n; o[4] = q[1] / n; o[5] = q[2] / n; o[6] = q[3] / n;
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// broadcast a double from a compile-time-constant lane through SGPRs (v_readlane_b32 x2)
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(x) without the IEEE sqrt/div expansions: v_rsq_f64 (measured 2^-24 relative) + 2 Newton steps
// = 2.2 ulp worst case (a third step gives 1.9; tests/microbench/rsq_f64_accuracy.hip).  Explicit fma under
// contract(off): every inlined copy rounds identically.
__device__ __forceinline__ double rsqrt_nr(double x) {
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rsq(x), h = 0.5 * x;
    y = y * __builtin_fma(-(h * y), y, 1.5);
    y = y * __builtin_fma(-(h * y), y, 1.5);
    return y;
}
// 1/x: v_rcp_f64 + 2 Newton steps (same accuracy class as rsqrt_nr)
__device__ __forceinline__ double rcp_nr(double x) {
#pragma clang fp contract(off)
    double r = __builtin_amdgcn_rcp(x);
    r = r * __builtin_fma(-x, r, 2.0);
    r = r * __builtin_fma(-x, r, 2.0);
    return r;
}
// value of lane (i + N) mod 16 of the same 16-lane row, through the DPP row-rotate path (no LDS crossbar).
// On a value that is already symmetric under the coarser exchanges this equals the xor-N butterfly partner.
template <int N>
__device__ __forceinline__ double row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 16 lanes of a row: the butterfly 8, 4, 2, 1 (fixed order => deterministic)
__device__ __forceinline__ double grp16_sum(double v) {
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return v;
}
// maximum over the 16-lane row, every lane of the row ends with it (DPP row rotations)
__device__ __forceinline__ double grp16_max(double v) {
    v = fmax(v, row_ror<8>(v)); v = fmax(v, row_ror<4>(v)); v = fmax(v, row_ror<2>(v)); v = fmax(v, row_ror<1>(v));
    return v;
}
// the value lane L holds, as a wave-uniform number (v_readlane)
__device__ __forceinline__ double rows_lane(double v, int L) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}
// sum over the four 16-lane rows of a value that is already uniform inside each row (after grp16_sum): v_readlane of lanes 0, 16, 32
// and 48, no LDS crossbar; the result is wave-uniform.  ((row 0 + row 1) + (row 2 + row 3))
__device__ __forceinline__ double rows4_sum(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double a0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double a1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double a2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double a3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (a0 + a1) + (a2 + a3);
}
// wave-level all-reduce (butterfly 32, 16, 8, 4, 2, 1: every lane ends with the total)
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64);
    return grp16_sum(v);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
// block-level sum for blockDim.x = multiple of 64 (<= 1024); scratch >= 16 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wid] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; i++) s += scratch[i];
    return s;
}
// several block-level reductions behind ONE barrier pair: v[0..NS) are summed, v[NS..NS+NM) maximised; every
// thread gets all results.  Same order as block_sum / block_max (wave butterfly, then waves in order).
// scratch >= 16 * (NS + NM) doubles of LDS.
template <int NS, int NM>
__device__ __forceinline__ void block_reduce(double* v, double* scratch) {
#pragma unroll
    for (int k = 0; k < NS; k++) v[k] = wave_sum(v[k]);
#pragma unroll
    for (int k = 0; k < NM; k++) v[NS + k] = wave_max(v[NS + k]);
    int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NS + NM; k++) scratch[k * 16 + wid] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; k++) { double s = 0; for (int i = 0; i < nw; i++) s += scratch[k * 16 + i]; v[k] = s; }
#pragma unroll
    for (int k = 0; k < NM; k++) { double s = scratch[(NS + k) * 16]; for (int i = 1; i < nw; i++) s = scratch[(NS + k) * 16 + i] > s ? scratch[(NS + k) * 16 + i] : s; v[NS + k] = s; }
}
__device__ __forceinline__ double block_max(double v, double* scratch) {
    v = wave_max(v);
    int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wid] = v;
    __syncthreads();
    double s = scratch[0];
    for (int i = 1; i < nw; i++) s = scratch[i] > s ? scratch[i] : s;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inverse-depth projection factors, one evaluation:
//   kind 0 ProjectionTwoFrameOneCamFactor (R/factor/projection_factor.cpp:179-256), 1 ProjectionTwoFrameTwoCamFactor (:77-166),
//   kind 2 ProjectionOneFrameTwoCamFactor (:269-329; Pi = Pj = identity pose, no lever arm).  e2 = the camera projected into
//   (= ex for kind 0).  J (if jac): d r / d pose_i (2x6 local, row-major) | pose_j | ex | ex2 | lambda (2) = 50 doubles.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void red_out(const double* red, const double* Mx, double sgn, double* Jd, int col) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Jd[i * 6 + col + j] = sgn * (red[i * 3] * Mx[j] + red[i * 3 + 1] * Mx[3 + j] + red[i * 3 + 2] * Mx[6 + j]);
}

__device__ __forceinline__ void d_idepth_eval(int kind, const double* Pi, const double* Pj, const double* ex, const double* e2, double inv_dep,
                                              const double* pts, double si, const double* pbg, double* r, double* Jq, bool jac) {
    const double* pts_i = pts; const double* pts_j = pts + 3;
    const double lever[3] = { kind == 2 ? 0.0 : pbg[0], kind == 2 ? 0.0 : pbg[1], kind == 2 ? 0.0 : pbg[2] };
    double pci[3] = { pts_i[0] / inv_dep, pts_i[1] / inv_dep, pts_i[2] / inv_dep }, pimu_i[3], pimu_j[3], t[3], pcj[3], qi[4], w[3];
    qrot(ex + 3, pci, pimu_i);
#pragma unroll
    for (int k = 0; k < 3; k++) pimu_i[k] += ex[k] - lever[k];
    if (kind == 2) { pimu_j[0] = pimu_i[0]; pimu_j[1] = pimu_i[1]; pimu_j[2] = pimu_i[2]; }
    else {
        qrot(Pi + 3, pimu_i, w);
#pragma unroll
        for (int k = 0; k < 3; k++) w[k] += Pi[k] - Pj[k];
        qinv(Pj + 3, qi); qrot(qi, w, pimu_j);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = pimu_j[k] + lever[k] - e2[k];
    qinv(e2 + 3, qi); qrot(qi, t, pcj);
    const double dep = pcj[2];
    r[0] = si * (pcj[0] / dep - pts_j[0]);
    r[1] = si * (pcj[1] / dep - pts_j[1]);
    if (!jac) return;
    double red[6] = { si * (1. / dep), 0, si * (-pcj[0] / (dep * dep)), 0, si * (1. / dep), si * (-pcj[1] / (dep * dep)) };
    double Ri[9], Rj[9], ric[9], ric2[9], ric2T[9], RjT[9], Am[9], Bm[9], Cm[9], S[9], M[9];
    q2R(ex + 3, ric); q2R(e2 + 3, ric2); mat3T(ric2, ric2T);
    q2R(Pi + 3, Ri); q2R(Pj + 3, Rj);                   // identities for kind 2
    mat3T(Rj, RjT);
    mat3mul(ric2T, RjT, Am); mat3mul(Am, Ri, Bm); mat3mul(Bm, ric, Cm);
    for (int k = 0; k < 50; k++) Jq[k] = 0.0;
    if (kind != 2) {
        skew3(pimu_i, S); mat3mul(Bm, S, M);
        red_out(red, Am, 1.0, Jq, 0); red_out(red, M, -1.0, Jq, 3);
        skew3(pimu_j, S); mat3mul(ric2T, S, M);
        red_out(red, Am, -1.0, Jq + 12, 0); red_out(red, M, 1.0, Jq + 12, 3);
    }
    if (kind == 0) {
        double T1[9], tmp[3], v[3], w2[3], u[3], S2[9], S3[9], L[9];
#pragma unroll
        for (int k = 0; k < 9; k++) T1[k] = Bm[k] - ric2T[k];
        mat3vec(Cm, pci, tmp);
        skew3(pci, S); mat3mul(Cm, S, L); skew3(tmp, S2);
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = ex[k] - pbg[k];
        mat3vec(Ri, v, w2);
#pragma unroll
        for (int k = 0; k < 3; k++) w2[k] += Pi[k] - Pj[k];
        mat3vec(RjT, w2, u);
#pragma unroll
        for (int k = 0; k < 3; k++) u[k] += pbg[k] - ex[k];
        mat3vec(ric2T, u, v); skew3(v, S3);
#pragma unroll
        for (int k = 0; k < 9; k++) M[k] = -L[k] + S2[k] + S3[k];
        red_out(red, T1, 1.0, Jq + 24, 0); red_out(red, M, 1.0, Jq + 24, 3);
    } else {
        skew3(pci, S); mat3mul(Cm, S, M);
        red_out(red, Bm, 1.0, Jq + 24, 0); red_out(red, M, -1.0, Jq + 24, 3);
        skew3(pcj, S);
        red_out(red, ric2T, -1.0, Jq + 36, 0); red_out(red, S, 1.0, Jq + 36, 3);
    }
    double v3[3];
    mat3vec(Cm, pts_i, v3);
#pragma unroll
    for (int i = 0; i < 2; i++) Jq[48 + i] = (red[i * 3] * v3[0] + red[i * 3 + 1] * v3[1] + red[i * 3 + 2] * v3[2]) * -1.0 / (inv_dep * inv_dep);
}
