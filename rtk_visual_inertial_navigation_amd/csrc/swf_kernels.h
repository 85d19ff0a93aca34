// swf_kernels.h — HIP kernels of the sliding-window Gauss-Newton hot path (gfx950).
//
// Phase list of one trust-region iteration (all windows of the batch at once; every kernel
// early-exits for windows whose device-side status says "nothing to do"):
//
//   LIN   k_eval_proj<1> k_eval_imu<1> k_eval_scalar<1> k_eval_prior<1>   r, J, cost per factor
//         k_lm_elim  k_clique_elim                                        group-0 elimination
//         k_lm_gemm                                                        P = Y W^T (landmark Schur)
//         k_assemble                                                       S, rhs, g, diag (owner-computes)
//   STEP  k_chol_solve                                                     S = L L^T, y_f
//         k_backsub_lm  k_backsub_clique                                   y_e
//         k_jtimes_proj<0> k_jtimes_gen<0>                                 |J D^-2 g|^2 (Cauchy point)
//         k_dogleg                                                         step, candidate = Plus(x, step)
//         k_jtimes_*<1>  k_eval_*<0>                                       model decrease, candidate cost
//         k_decide                                                         accept / reject, radius, convergence
//
// Factor math follows the reference sources cited at each kernel (R/ =
// /root/reference/rtk_visual_inertial_src/rtk_visual_inertial/src/).
#pragma once
#include "swf_dev.h"

struct DevOpt {
    int max_iter, step_mode, strategy, jacobi;   // strategy: SWF_DOGLEG / SWF_LEVENBERG_MARQUARDT; jacobi: Solver::Options::jacobi_scaling (LM only)
    double r0, max_r, min_r, min_rel_dec, ftol, gtol, ptol, min_mu, max_mu, mu_inc, min_diag, max_diag;
};



// The diagonal the trust-region damping multiplies.  Without Jacobi scaling: clamp(diag(J^T J)) (LevenbergMarquardtStrategy / DoglegStrategy,
// min_diagonal .. max_diagonal).  With it (ceres default, LM): ceres scales column i of J by s_i = 1 / (1 + sqrt(diag_i)) of the FIRST
// linearisation (TrustRegionMinimizer::IterationZero), damps the scaled system with clamp(diag(J'^T J')) and un-scales the step; in the
// original coordinates that is (J^T J + mu D_eff) d = -g with D_eff_i = clamp(diag_i s_i^2) / s_i^2.  jsc keeps 1 / s_i^2 = (1 + sqrt(diag0_i))^2
// per local dimension; `first` = this is the solve's first linearisation (the slot is written, by every lane that computes it, identically).
__device__ __forceinline__ double damp_diag(const DevOpt& O, double d, double* jsc_slot, bool first) {
    if (!O.jacobi) return clampd(d, O.min_diag, O.max_diag);
    double q;
    if (first) { double r = 1.0 + sqrt(d > 0.0 ? d : 0.0); q = r * r; *jsc_slot = q; }
    else q = *jsc_slot;
    return clampd(d / q, O.min_diag, O.max_diag) * q;
}

// Where an evaluation reads its parameter blocks and what gates it.  Cost-only evaluations (JAC = false) run at the candidate xc of the
// windows whose k_dogleg proposed a step (eval_cand).  Jacobian evaluations run at x for the windows that re-linearise (need_lin) — or,
// in the SPECULATIVE flow of the dogleg loop (DevBatch::spec, set per launch by the engine), at the CANDIDATE of the windows with a
// proposed step: the accepted candidate is the next linearisation point, so its residuals, costs AND Jacobians are formed in one pass
// instead of a cost pass at xc followed by a Jacobian pass at the same point (k_decide then turns the accepted candidate into x and
// the elimination kernels find its Jacobians in place; a rejected dogleg step re-uses the previous reduced system and needs no Jacobian).
// (spec == 2: the fused step kernel of the latency path — its workgroups formed the candidate themselves and evaluate it ungated)
template <bool JAC> __device__ __forceinline__ bool eval_gate(const DevBatch& B, const WinState& s) { return B.spec == 2 ? true : (JAC && !B.spec) ? s.need_lin != 0 : s.eval_cand != 0; }
template <bool JAC> __device__ __forceinline__ const double* eval_src(const DevBatch& B) { return (JAC && !B.spec) ? B.x : B.xc; }

#define CLIGHT_D 299792458.0
#define OMGE_D 7.2921151467E-5

// =========================================================================================
// projection_factor::Evaluate (R/factor/projection_factor.cpp:13-65) + CauchyLoss corrector
// (R/factor/marginalization_factor.cpp:23-45).  One lane per observation; SoA outputs so
// that every store instruction of a wave is one contiguous 512-byte run.
// =========================================================================================
// keep != nullptr (Jacobian evaluations only): the whitened residual and the Jacobian rows also stay in registers,
// keep[0..5] = row 0, keep[6..11] = row 1 of Jp, keep[12..13] = r, keep[14..16] = row 0, keep[17..19] = row 1 of Jl (zero where the
// block is constant) — for the per-frame sums of the fused evaluation kernel, and, with STORE = false (nothing written), for the
// back-substitution pass, which re-derives an observation's Jacobian from its 56 B of inputs instead of re-reading 144 B
// the arithmetic of one observation from its inputs in registers (P pose, E extrinsic, X landmark, (u0, u1) image point; jp / jl: the pose /
// the landmark is variable) — d_eval_proj_at below loads them one dependent level after another; the back-substitution pass fetches
// them together with everything else it needs
// returns the observation's cost (0.5 rho(|r|^2)); the caller's block adds its observations' costs into one partial sum
template <bool JAC, bool STORE>
__device__ __forceinline__ double proj_core(const DevBatch& B, int i, const WinRec& W, const double* P, const double* E, const double* X,
                                          double u0, double u1, bool jp, bool jl, double* keep);
template <bool JAC, bool STORE = true>
__device__ __forceinline__ double d_eval_proj_at(const DevBatch& B, int i, double* keep) {
    int w = B.p_win[i];
    const WinState& s = B.ws[w];
    if (!eval_gate<JAC>(B, s)) return 0.0;
    const WinRec& W = B.win[w];
    const double* xs = eval_src<JAC>(B);
    const double* pose = xs + B.p_xpose[i];
    const double* ex = xs + B.p_xex[i];
    const double* lm = xs + B.p_xlm[i];
    double P[7], E[7], X[3];
#pragma unroll
    for (int k = 0; k < 7; k++) { P[k] = pose[k]; E[k] = ex[k]; }
    X[0] = lm[0]; X[1] = lm[1]; X[2] = lm[2];
    return proj_core<JAC, STORE>(B, i, W, P, E, X, B.p_uv[2 * i], B.p_uv[2 * i + 1], JAC && B.p_lpose[i] >= 0, JAC && B.p_llm[i] >= 0, keep);
}
template <bool JAC, bool STORE>
__device__ __forceinline__ double proj_core(const DevBatch& B, int i, const WinRec& W, const double* P, const double* E, const double* X,
                                          double u0, double u1, bool jp, bool jl, double* keep) {
    double Qj_inv[4], qic_inv[4], d[3], pts_imu[3], t[3], pc[3];
    qinv(P + 3, Qj_inv);
    qinv(E + 3, qic_inv);
    d[0] = X[0] - P[0]; d[1] = X[1] - P[1]; d[2] = X[2] - P[2];
    qrot(Qj_inv, d, pts_imu);
    t[0] = pts_imu[0] + W.pbg[0] - E[0]; t[1] = pts_imu[1] + W.pbg[1] - E[1]; t[2] = pts_imu[2] + W.pbg[2] - E[2];
    qrot(qic_inv, t, pc);
    double dep = pc[2], si = W.proj_sqrt_info;
    double r0 = si * (pc[0] / dep - u0);
    double r1 = si * (pc[1] / dep - u1);
    // Cauchy: rho'' < 0 always => scale r and J by sqrt(rho'); block cost = 0.5 rho(s)
    double sr = 1.0, cost;
    double sq = r0 * r0 + r1 * r1;
    if (W.proj_loss_a > 0) {
        double b = W.proj_loss_a * W.proj_loss_a, c = 1.0 / b;
        double sum = 1.0 + sq * c, inv = 1.0 / sum;
        cost = 0.5 * b * log(sum);
        sr = sqrt(inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308);
    } else cost = 0.5 * sq;
    if (!JAC) return cost;
    int n = B.n_proj;
    if (STORE) { B.p_r[i] = r0 * sr; B.p_r[n + i] = r1 * sr; }
    if (keep) { keep[12] = r0 * sr; keep[13] = r1 * sr; }
    if (!jp && !jl) return cost;
    double Rj[9], ric[9], ricT[9], RjT[9], A[9];
    q2R(P + 3, Rj); q2R(E + 3, ric);
    mat3T(ric, ricT); mat3T(Rj, RjT);
    double red[6] = { si * (1. / dep), 0, si * (-pc[0] / (dep * dep)), 0, si * (1. / dep), si * (-pc[1] / (dep * dep)) };
    mat3mul(ricT, RjT, A);
    if (jp) {
        double S[9], Bm[9];
        skew3(pts_imu, S);
        mat3mul(ricT, S, Bm);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double u = 0, v = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { u += red[a * 3 + k] * -A[k * 3 + j]; v += red[a * 3 + k] * Bm[k * 3 + j]; }
                // the translation half of Jp is -Jl, bit for bit (same products, negated): it is stored only where Jl is not (constant landmark)
                if (STORE) { if (!jl) B.p_Jp[(a * 6 + j) * n + i] = u * sr; B.p_Jp[(a * 6 + 3 + j) * n + i] = v * sr; }
                if (keep) { keep[a * 6 + j] = u * sr; keep[a * 6 + 3 + j] = v * sr; }
            }
    }
    if (jl) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double u = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) u += red[a * 3 + k] * A[k * 3 + j];
                if (STORE) B.p_Jl[(a * 3 + j) * n + i] = u * sr;
                if (keep) keep[14 + a * 3 + j] = u * sr;
            }
    }
    return cost;
}

// cost-only evaluation of one frame-sum block's observations at the candidate (k_post_dogleg): the block's cost in p_cpart, formed
// exactly as the Jacobian evaluation of the same block forms it (d_eval_proj_fs: thread t <-> observation t of the block, block_sum)
__device__ __forceinline__ void d_eval_proj_cost(const DevBatch& B, int blk) {
    __shared__ double csum[16];
    const int w = B.fsb_win[blk];
    if (!eval_gate<false>(B, B.ws[w])) return;               // uniform per block (a block holds observations of one window)
    const int o_beg = B.fsb_obs0[blk], cnt = B.fsb_obs0[blk + 1] - o_beg, tid = threadIdx.x;
    double c = 0.0;
    if (tid < cnt) c = d_eval_proj_at<false>(B, o_beg + tid, nullptr);
    c = block_sum(c, csum);
    if (tid == 0) B.p_cpart[blk] = c;
}

// =========================================================================================
// IMUFactor::Evaluate (R/factor/imu_factor.cpp:5-101) on IntegrationBase::evaluate
// (R/factor/integration_base.cpp:144-174).  One wavefront per factor: lane 0 builds the
// un-whitened residual and the sparse 15x30 Jacobian in LDS, all 64 lanes apply the 15x15
// sqrt-information (the 6.7k-MAC part) and write the whitened blocks row-major.
// =========================================================================================
// part 0: residual + d/d pose_i;  1: d/d sb_i;  2: d/d pose_j;  3: d/d sb_j.  The parts run on different waves of the
// block (they share only cheap prefixes), which cuts the lane-serial critical path of the kernel.
__device__ __forceinline__ void imu_unwhitened(const double* pi, const double* sbi, const double* pj, const double* sbj,
                               const double* pre, const double* pbg, const double* gw,
                               double* raw, double* U, bool jac, int part) {
    const double* Pi = pi; const double* Qi = pi + 3;
    const double* Vi = sbi; const double* Bai = sbi + 3; const double* Bgi = sbi + 6;
    const double* Pj = pj; const double* Qj = pj + 3;
    const double* Vj = sbj; const double* Baj = sbj + 3; const double* Bgj = sbj + 6;
    const double* dp = pre + SWF_PRE_DP; const double* dq = pre + SWF_PRE_DQ; const double* dv = pre + SWF_PRE_DV;
    const double* lba = pre + SWF_PRE_LBA; const double* lbg = pre + SWF_PRE_LBG;
    const double* dp_dba = pre + SWF_PRE_DP_DBA; const double* dp_dbg = pre + SWF_PRE_DP_DBG;
    const double* dq_dbg = pre + SWF_PRE_DQ_DBG; const double* dv_dba = pre + SWF_PRE_DV_DBA;
    const double* dv_dbg = pre + SWF_PRE_DV_DBG;
    double T = pre[SWF_PRE_SUMDT];
    const double* gyri = pre + SWF_PRE_GYRI; const double* gyrj = pre + SWF_PRE_GYRJ;
    double Qi_inv[4];
    qinv(Qi, Qi_inv);
#define SETU(R0, C0, MAT, SGN) for (int i_ = 0; i_ < 3; i_++) for (int j_ = 0; j_ < 3; j_++) U[(R0 + i_) * 30 + C0 + j_] = SGN * MAT[i_ * 3 + j_];
    // U: 15 x 30 row-major, columns [pose_i(6) | sb_i(9) | pose_j(6) | sb_j(9)]; zeroed by the factor's lanes beforehand
    if (part == 0 || part == 2) {
        // corrected delta rotation and w_j x p_bg
        double dbg[3], th[3], dqc[4], cq[4], cq_inv[4], wj[3], wjPbg[3];
        for (int k = 0; k < 3; k++) { dbg[k] = Bgi[k] - lbg[k]; wj[k] = gyrj[k] - Bgj[k]; }
        mat3vec(dq_dbg, dbg, th);
        dqc[0] = th[0] / 2; dqc[1] = th[1] / 2; dqc[2] = th[2] / 2; dqc[3] = 1.0;
        qmul(dq, dqc, cq);
        cross3(wj, pbg, wjPbg);
        qinv(cq, cq_inv);
        if (part == 0) {
            double dba[3], cv[3], cp[3], t1[3], t2[3], QjPbg[3], wi[3], wiPbg[3], QjwjPbg[3];
            for (int k = 0; k < 3; k++) { dba[k] = Bai[k] - lba[k]; wi[k] = gyri[k] - Bgi[k]; }
            mat3vec(dv_dba, dba, t1); mat3vec(dv_dbg, dbg, t2);
            for (int k = 0; k < 3; k++) cv[k] = dv[k] + t1[k] + t2[k];
            mat3vec(dp_dba, dba, t1); mat3vec(dp_dbg, dbg, t2);
            for (int k = 0; k < 3; k++) cp[k] = dp[k] + t1[k] + t2[k];
            qrot(Qj, pbg, QjPbg);
            cross3(wi, pbg, wiPbg);
            qrot(Qj, wjPbg, QjwjPbg);
            double ap[3], av[3], rp[3], rv[3];
            for (int k = 0; k < 3; k++) {
                ap[k] = 0.5 * gw[k] * T * T + ((Pj[k] - Pi[k]) - QjPbg[k]) - Vi[k] * T;
                av[k] = gw[k] * T + (Vj[k] - QjwjPbg[k]) - Vi[k];
            }
            qrot(Qi_inv, ap, rp);
            qrot(Qi_inv, av, rv);
            for (int k = 0; k < 3; k++) {
                raw[0 + k] = rp[k] - cp[k] + pbg[k] + wiPbg[k] * T;
                raw[6 + k] = rv[k] - cv[k] + wiPbg[k];
                raw[9 + k] = Baj[k] - Bai[k];
                raw[12 + k] = Bgj[k] - Bgi[k];
            }
            double qij[4], e[4];
            qmul(Qi_inv, Qj, qij);
            qmul(cq_inv, qij, e);
            raw[3] = 2 * e[0]; raw[4] = 2 * e[1]; raw[5] = 2 * e[2];
            if (!jac) return;
            // d/d pose_i
            double Ri_inv[9], M[9], S[9], tmpq[4], Qj_inv[4];
            q2R(Qi_inv, Ri_inv);
            qinv(Qj, Qj_inv);
            SETU(0, 0, Ri_inv, -1.0)
            skew3(rp, S); SETU(0, 3, S, 1.0)
            qmul(Qj_inv, Qi, tmpq);
            qleft_qright_br(tmpq, cq, M); SETU(3, 3, M, -1.0)
            skew3(rv, S); SETU(6, 3, S, 1.0)
        } else {
            if (!jac) return;
            // d/d pose_j (columns 15..20)
            double Ri_inv[9], Rj[9], M[9], S[9], N[9], tmpq[4], tmpq2[4], Spbg[9], RiRj[9];
            q2R(Qi_inv, Ri_inv);
            q2R(Qj, Rj);
            skew3(pbg, Spbg);
            mat3mul(Ri_inv, Rj, RiRj);
            SETU(0, 15, Ri_inv, 1.0)
            mat3mul(RiRj, Spbg, N); SETU(0, 18, N, 1.0)
            qmul(cq_inv, Qi_inv, tmpq);
            qmul(tmpq, Qj, tmpq2);
            qleft_br(tmpq2, M); SETU(3, 18, M, 1.0)
            skew3(wjPbg, S);
            mat3mul(RiRj, S, N); SETU(6, 18, N, 1.0)
        }
    } else if (part == 1) {
        if (!jac) return;
        // d/d sb_i  (columns 6..14)
        double Ri_inv[9], M[9], N[9], tmpq[4], tmpq2[4], Qj_inv[4], Spbg[9];
        q2R(Qi_inv, Ri_inv);
        qinv(Qj, Qj_inv);
        skew3(pbg, Spbg);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            U[(0 + i) * 30 + 6 + j] = -Ri_inv[i * 3 + j] * T;
            U[(0 + i) * 30 + 9 + j] = -dp_dba[i * 3 + j];
            U[(0 + i) * 30 + 12 + j] = -dp_dbg[i * 3 + j] + Spbg[i * 3 + j] * T;
            U[(6 + i) * 30 + 6 + j] = -Ri_inv[i * 3 + j];
            U[(6 + i) * 30 + 9 + j] = -dv_dba[i * 3 + j];
            U[(6 + i) * 30 + 12 + j] = -dv_dbg[i * 3 + j] + Spbg[i * 3 + j];
        }
        qmul(Qj_inv, Qi, tmpq);
        qmul(tmpq, dq, tmpq2);
        qleft_br(tmpq2, M);
        mat3mul(M, dq_dbg, N); SETU(3, 12, N, -1.0)
        for (int k = 0; k < 3; k++) { U[(9 + k) * 30 + 9 + k] = -1.0; U[(12 + k) * 30 + 12 + k] = -1.0; }
    } else {
        if (!jac) return;
        // d/d sb_j (columns 21..29)
        double Ri_inv[9], Rj[9], N[9], Spbg[9], RiRj[9];
        q2R(Qi_inv, Ri_inv);
        q2R(Qj, Rj);
        skew3(pbg, Spbg);
        mat3mul(Ri_inv, Rj, RiRj);
        SETU(6, 21, Ri_inv, 1.0)
        mat3mul(RiRj, Spbg, N); SETU(6, 27, N, -1.0)
        for (int k = 0; k < 3; k++) { U[(9 + k) * 30 + 24 + k] = 1.0; U[(12 + k) * 30 + 27 + k] = 1.0; }
    }
#undef SETU
}

#define IMU_FPB 8          // factors per block
#define IMU_LPF 32         // lanes per factor: 256-thread blocks = 4 waves, one per un-whitened part (residual only: part 0 alone)
// Whitening of one IMU factor by its IMU_LPF lanes:
// r = SI raw, cost, and (JAC) the whitened Jacobian SI * U written into the clique's dense column-major Jacobian.
template <bool JAC>
__device__ __forceinline__ void imu_whiten_store(const DevBatch& B, const GFac& G, int f, bool act, int sub,
                                                 const double* SIf, double* Uf, const double* rawf) {
    constexpr int LPF = IMU_LPF;
    // whitened residual: lane k < 15 of each factor's lanes (all within one 16-lane row)
    double rk = 0;
    if (act && sub < 15) {
        for (int k = 0; k < 15; k++) rk += SIf[sub * 15 + k] * rawf[k];
        if (JAC) B.g_r[G.roff + sub] = rk;
    }
    double c = grp16_sum((act && sub < 15) ? rk * rk : 0.0);
    if (act && sub == 0) B.g_cost[f] = 0.5 * c;
    if (!JAC || !act) return;
    const int cb[4] = { 0, 6, 15, 21 };
    // Jacobian block positions of the four parameter blocks, fetched before the products (off the store loop's chain)
    const int jo0 = B.s_joff[G.slot0], jo1 = B.s_joff[G.slot0 + 1], jo2 = B.s_joff[G.slot0 + 2], jo3 = B.s_joff[G.slot0 + 3];
    // whitened Jacobian SI * U: lane c < 30 owns column c, holds it in registers and forms the 15 rows from broadcast
    // SI reads at constant offsets (120 multiply-adds, fully unrolled; SI is upper triangular, k ascending from the row)
    if (sub < 30) {
        double u[15];
#pragma unroll
        for (int k = 0; k < 15; k++) u[k] = Uf[k * 30 + sub];
#pragma unroll
        for (int row = 0; row < 15; row++) {
            double a = 0;
#pragma unroll
            for (int k = row; k < 15; k++) a += SIf[row * 15 + k] * u[k];
            Uf[row * 30 + sub] = a;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // the clique's dense Jacobian is column-major in HBM: lanes run over the rows of one column (coalesced stores)
    for (int e = sub; e < 450; e += LPF) {
        int col = e / 15, row = e - col * 15;
        int sl = col < 6 ? 0 : col < 15 ? 1 : col < 21 ? 2 : 3;
        int jo = sl == 0 ? jo0 : sl == 1 ? jo1 : sl == 2 ? jo2 : jo3;
        if (jo >= 0) B.g_J[jo + (col - cb[sl]) * G.jld + row] = Uf[row * 30 + col];
    }
}

template <bool JAC>
__device__ __forceinline__ void d_eval_imu(const DevBatch& B, int bid) {
    constexpr int LPF = IMU_LPF;
    __shared__ double SI[IMU_FPB][225];
    __shared__ double U[JAC ? IMU_FPB : 1][JAC ? 450 : 1];    // Jacobian scratch: linearisation only
    __shared__ double raw[IMU_FPB][16];
    __shared__ double st[IMU_FPB][32];
    __shared__ double pr[IMU_FPB][SWF_PRE_SQRTINFO + 6];     // record head (dp .. gyr_j) | pbg | gw, staged by the factor's 16 lanes
    int tid = threadIdx.x, fl = tid / LPF, sub = tid % LPF;
    int q = bid * IMU_FPB + fl;
    bool valid = q < B.n_imu;
    int f = B.imu_gf[valid ? q : B.n_imu - 1];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    bool act = valid && eval_gate<JAC>(B, s);
    const WinRec& W = B.win[G.win];
    const double* xs = eval_src<JAC>(B);
    const double* pre = B.imu_pre + (size_t)G.data * SWF_PRE_DOUBLES;
    if (act) {
        for (int k = sub; k < SWF_PRE_SQRTINFO; k += LPF) pr[fl][k] = pre[k];
        if (sub < 3) { pr[fl][SWF_PRE_SQRTINFO + sub] = W.pbg[sub]; pr[fl][SWF_PRE_SQRTINFO + 3 + sub] = W.gw[sub]; }
        if (JAC) for (int k = sub; k < 450; k += LPF) U[JAC ? fl : 0][k] = 0.0;          // the serial lane only fills the non-zero blocks
        for (int k = sub; k < 225; k += LPF) SI[fl][k] = pre[SWF_PRE_SQRTINFO + k];
        for (int k = sub; k < 32; k += LPF) {
            int sl = k < 7 ? 0 : k < 16 ? 1 : k < 23 ? 2 : 3;
            int o = k < 7 ? k : k < 16 ? k - 7 : k < 23 ? k - 16 : k - 23;
            st[fl][k] = xs[B.s_x[G.slot0 + sl] + o];
        }
    }
    __syncthreads();
    // lanes 0..7 of each wave: one factor each, wave p = part p (residual-only evaluation: part 0 alone)
    if ((tid & 63) < IMU_FPB && (JAC || tid < 64)) {
        int part = tid >> 6, fq = tid & 63;
        int q2 = bid * IMU_FPB + fq;
        if (q2 < B.n_imu) {
            const GFac& G2 = B.gf[B.imu_gf[q2]];
            const WinState& s2 = B.ws[G2.win];
            if (eval_gate<JAC>(B, s2)) {
                imu_unwhitened(st[fq], st[fq] + 7, st[fq] + 16, st[fq] + 23, pr[fq],
                               pr[fq] + SWF_PRE_SQRTINFO, pr[fq] + SWF_PRE_SQRTINFO + 3, raw[fq], U[JAC ? fq : 0], JAC, part);
            }
        }
    }
    __syncthreads();
    (void)W;
    imu_whiten_store<JAC>(B, G, f, act, sub, SI[fl], U[JAC ? fl : 0], raw[fl]);
}

template <bool JAC>
__global__ void __launch_bounds__(IMU_FPB * IMU_LPF) __attribute__((amdgpu_waves_per_eu(4, 4))) k_eval_imu(DevBatch B) { d_eval_imu<JAC>(B, blockIdx.x); }

// =========================================================================================
// scalar factors, one lane each:
//   RTKCarrierPhaseFactor  R/factor/gnss_factor.cpp:105-138     RTKPseudorangeFactor :140-168
//   SppDopplerFactor       :174-212 (+ velecitydistance, R/gnss/src/common_function.cpp:411-421)
//   InitialBlackFactor     R/factor/initial_factor.cpp:81-87
//   SppPseudorangeFactor   gnss_factor.cpp:9-39   SppCarrierPhaseFactor :45-80   FixedIntegerFactor :85-96
//   distance() R/gnss/src/common_function.cpp:126-134 ; varerr2() gnss_factor.cpp:98-103 (sinf!)
// =========================================================================================
__device__ __forceinline__ double gnss_distance(const double* rr, const double* rs, double* e) {
    e[0] = rr[0] - rs[0]; e[1] = rr[1] - rs[1]; e[2] = rr[2] - rs[2];
    double r = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    e[0] /= r; e[1] /= r; e[2] /= r;
    return r + OMGE_D * (rs[0] * rr[1] - rs[1] * rr[0]) / CLIGHT_D;
}
__device__ __forceinline__ double varerr2(double el, double dt, double mea_var) {
    double b = CLIGHT_D * 5e-12 * dt;
    // the reference calls single-precision sinf() (gnss_factor.cpp:100).  glibc's sinf is
    // correctly rounded; the device libm's is not (1-2 ulp), so round the fp64 sine instead.
    double sinel = (double)(float)sin((double)(float)el);
    return (mea_var / sinel / sinel) + b * b;
}
template <bool JAC>
__device__ __forceinline__ void d_eval_scalar(const DevBatch& B, int bid) {
    int q = bid * blockDim.x + threadIdx.x;
    if (q >= B.n_sc) return;
    int f = B.sc_gf[q];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    if (!eval_gate<JAC>(B, s)) return;
    if (G.type == GF_IDP || G.type == GF_PROJX) return;          // two-row projection factors: their own kernel (k_eval_idp), they only share the J v code
    const WinRec& W = B.win[G.win];
    const double* xs = eval_src<JAC>(B);
    int s0 = G.slot0, ld = G.jld;          // column stride of the clique's dense column-major Jacobian
    double r;
    if (G.type == GF_CP) {
        const double* dat = B.cp_dat + (size_t)G.data * SWF_CP_DOUBLES;
        const double* pose = xs + B.s_x[s0];
        double amb = xs[B.s_x[s0 + 1]], clk = xs[B.s_x[s0 + 2]];
        double xg[3] = { pose[0] + W.base[0], pose[1] + W.base[1], pose[2] + W.base[2] }, e[3];
        double r1 = gnss_distance(xg, dat, e);
        double wgt = 1.0;
        if (dat[8] != 0.0) wgt = 1 / sqrt(varerr2(dat[5], dat[6], dat[7]));
        r = wgt * (r1 - amb * dat[4] - dat[3] + clk);
        if (JAC) {
            int jo = B.s_joff[s0];
            if (jo >= 0) { B.g_J[jo] = wgt * e[0]; B.g_J[jo + 1 * ld] = wgt * e[1]; B.g_J[jo + 2 * ld] = wgt * e[2]; B.g_J[jo + 3 * ld] = 0; B.g_J[jo + 4 * ld] = 0; B.g_J[jo + 5 * ld] = 0; }
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = -wgt * dat[4];
            jo = B.s_joff[s0 + 2]; if (jo >= 0) B.g_J[jo] = wgt;
        }
    } else if (G.type == GF_PR) {
        const double* dat = B.pr_dat + (size_t)G.data * SWF_PR_DOUBLES;
        const double* pose = xs + B.s_x[s0];
        double clk = xs[B.s_x[s0 + 1]];
        double xg[3] = { pose[0] + W.base[0], pose[1] + W.base[1], pose[2] + W.base[2] }, e[3];
        double r1 = gnss_distance(xg, dat, e);
        double wgt = 1 / sqrt(varerr2(dat[4], dat[5], dat[6]));
        r = wgt * (r1 - dat[3] + clk);
        if (JAC) {
            int jo = B.s_joff[s0];
            if (jo >= 0) { B.g_J[jo] = wgt * e[0]; B.g_J[jo + 1 * ld] = wgt * e[1]; B.g_J[jo + 2 * ld] = wgt * e[2]; B.g_J[jo + 3 * ld] = 0; B.g_J[jo + 4 * ld] = 0; B.g_J[jo + 5 * ld] = 0; }
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = wgt;
        }
    } else if (G.type == GF_DOP) {
        const double* dat = B.dop_dat + (size_t)G.data * SWF_DOP_DOUBLES;
        const double* sb = xs + B.s_x[s0];
        double drift = xs[B.s_x[s0 + 1]];
        const double* pose = xs + B.s_x[s0 + 2];
        const double* rs = dat; const double* vs = dat + 3;
        double xg[3] = { pose[0] + W.base[0], pose[1] + W.base[1], pose[2] + W.base[2] }, e[3], ev[3];
        e[0] = xg[0] - rs[0]; e[1] = xg[1] - rs[1]; e[2] = xg[2] - rs[2];
        double rr = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        for (int k = 0; k < 3; k++) { e[k] /= rr; ev[k] = sb[k] - vs[k]; }
        double ee = ev[0] * e[0] + ev[1] * e[1] + ev[2] * e[2];
        double rate = ee + OMGE_D / CLIGHT_D * (vs[1] * xg[0] + rs[1] * sb[0] - vs[0] * xg[1] - rs[0] * sb[1]);
        double istd = dat[7];
        r = istd * (rate + drift + dat[6]);
        if (JAC) {
            int jo = B.s_joff[s0];
            if (jo >= 0) { for (int k = 0; k < 9; k++) B.g_J[jo + k * ld] = 0; B.g_J[jo] = istd * e[0]; B.g_J[jo + 1 * ld] = istd * e[1]; B.g_J[jo + 2 * ld] = istd * e[2]; }
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = istd;
            jo = B.s_joff[s0 + 2];
            if (jo >= 0) { for (int k = 0; k < 3; k++) { B.g_J[jo + k * ld] = istd * (ev[k] - ee * e[k]) / rr; B.g_J[jo + (3 + k) * ld] = 0; } }
        }
    } else if (G.type == GF_SPR) {
        const double* dat = B.gx_dat + G.data;          // sat[3] P1 istd
        const double* pose = xs + B.s_x[s0];
        double clk = xs[B.s_x[s0 + 1]];
        double xg[3] = { pose[0] + W.base[0], pose[1] + W.base[1], pose[2] + W.base[2] }, e[3];
        double r1 = gnss_distance(xg, dat, e);
        double wgt = dat[4];
        r = wgt * (r1 + clk - dat[3]);
        if (JAC) {
            int jo = B.s_joff[s0];
            if (jo >= 0) { B.g_J[jo] = wgt * e[0]; B.g_J[jo + 1 * ld] = wgt * e[1]; B.g_J[jo + 2 * ld] = wgt * e[2]; B.g_J[jo + 3 * ld] = 0; B.g_J[jo + 4 * ld] = 0; B.g_J[jo + 5 * ld] = 0; }
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = wgt;
        }
    } else if (G.type == GF_SCP) {
        const double* dat = B.gx_dat + G.data;          // sat[3] L1_lam istd lam ; blocks pose, clock, ambiguity
        const double* pose = xs + B.s_x[s0];
        double clk = xs[B.s_x[s0 + 1]], amb = xs[B.s_x[s0 + 2]];
        double xg[3] = { pose[0] + W.base[0], pose[1] + W.base[1], pose[2] + W.base[2] }, e[3];
        double r1 = gnss_distance(xg, dat, e);
        double wgt = dat[4];
        r = wgt * (r1 + clk - amb * dat[5] - dat[3]);
        if (JAC) {
            int jo = B.s_joff[s0];
            if (jo >= 0) { B.g_J[jo] = wgt * e[0]; B.g_J[jo + 1 * ld] = wgt * e[1]; B.g_J[jo + 2 * ld] = wgt * e[2]; B.g_J[jo + 3 * ld] = 0; B.g_J[jo + 4 * ld] = 0; B.g_J[jo + 5 * ld] = 0; }
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = wgt;
            jo = B.s_joff[s0 + 2]; if (jo >= 0) B.g_J[jo] = -wgt * dat[5];
        }
    } else if (G.type == GF_FIX) {
        const double* dat = B.gx_dat + G.data;          // N21 istd
        double na = xs[B.s_x[s0]], nb2 = xs[B.s_x[s0 + 1]];
        r = dat[1] * ((nb2 - na) - dat[0]);
        if (JAC) {
            int jo = B.s_joff[s0]; if (jo >= 0) B.g_J[jo] = -dat[1];
            jo = B.s_joff[s0 + 1]; if (jo >= 0) B.g_J[jo] = dat[1];
        }
    } else {   // GF_SP
        double wv = B.sp_w[G.data];
        r = wv * xs[B.s_x[s0]];
        if (JAC) { int jo = B.s_joff[s0]; if (jo >= 0) B.g_J[jo] = wv; }
    }
    B.g_cost[f] = 0.5 * r * r;
    if (JAC) B.g_r[G.roff] = r;
}

// =========================================================================================
// Inverse-depth projection factors (R/factor/projection_factor.cpp:77-329), one lane each, in their own
// kernel: the evaluation is register-hungry and must not drag down the occupancy of the fused k_eval_ps grid.
// =========================================================================================
template <bool JAC>
__global__ void __launch_bounds__(128) k_eval_idp(DevBatch B) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B.n_idp) return;
    int f = B.idp_gf[q];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    if (!eval_gate<JAC>(B, s)) return;
    const WinRec& W = B.win[G.win];
    const double* xs = eval_src<JAC>(B);
    int s0 = G.slot0, ld = G.jld;
    if (G.type == GF_PROJX) {
        // projection_factor::Evaluate (R/factor/projection_factor.cpp:13-65) with ALL THREE Jacobian blocks: pose_j, the camera
        // extrinsic (:50-57) and the landmark.  This is the generic-path twin of d_eval_proj: same residual, same corrector; used
        // when the extrinsic is variable (GlobalMarge un-freezes it, R/swf/swf_image.cpp:384-389) or the landmark is not in group 0.
        const double* dat = B.gx_dat + G.data;          // uv
        const double* pose = xs + B.s_x[s0]; const double* ex = xs + B.s_x[s0 + 1]; const double* lm = xs + B.s_x[s0 + 2];
        double Qj_inv[4], qic_inv[4], d[3], pts_imu[3], t[3], pc[3];
        qinv(pose + 3, Qj_inv); qinv(ex + 3, qic_inv);
        d[0] = lm[0] - pose[0]; d[1] = lm[1] - pose[1]; d[2] = lm[2] - pose[2];
        qrot(Qj_inv, d, pts_imu);
        t[0] = pts_imu[0] + W.pbg[0] - ex[0]; t[1] = pts_imu[1] + W.pbg[1] - ex[1]; t[2] = pts_imu[2] + W.pbg[2] - ex[2];
        qrot(qic_inv, t, pc);
        double dep = pc[2], si = W.proj_sqrt_info;
        double r0 = si * (pc[0] / dep - dat[0]), r1 = si * (pc[1] / dep - dat[1]);
        double sr = 1.0, cost, sq = r0 * r0 + r1 * r1;
        if (W.proj_loss_a > 0) {
            double b = W.proj_loss_a * W.proj_loss_a, c = 1.0 / b;
            double sum = 1.0 + sq * c, inv = 1.0 / sum;
            cost = 0.5 * b * log(sum);
            sr = sqrt(inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308);
        } else cost = 0.5 * sq;
        B.g_cost[f] = cost;
        if (!JAC) return;
        B.g_r[G.roff] = r0 * sr; B.g_r[G.roff + 1] = r1 * sr;
        double Rj[9], ric[9], ricT[9], RjT[9], A[9], S[9], Bm[9];
        q2R(pose + 3, Rj); q2R(ex + 3, ric);
        mat3T(ric, ricT); mat3T(Rj, RjT);
        double red[6] = { si * (1. / dep), 0, si * (-pc[0] / (dep * dep)), 0, si * (1. / dep), si * (-pc[1] / (dep * dep)) };
        mat3mul(ricT, RjT, A);
        int jo = B.s_joff[s0];
        if (jo >= 0) {
            skew3(pts_imu, S); mat3mul(ricT, S, Bm);
            for (int a = 0; a < 2; a++) for (int j = 0; j < 3; j++) {
                double u = 0, v = 0;
                for (int k = 0; k < 3; k++) { u += red[a * 3 + k] * -A[k * 3 + j]; v += red[a * 3 + k] * Bm[k * 3 + j]; }
                B.g_J[jo + j * ld + a] = u * sr; B.g_J[jo + (3 + j) * ld + a] = v * sr;
            }
        }
        jo = B.s_joff[s0 + 1];
        if (jo >= 0) {
            skew3(pc, S);
            for (int a = 0; a < 2; a++) for (int j = 0; j < 3; j++) {
                double u = 0, v = 0;
                for (int k = 0; k < 3; k++) { u += red[a * 3 + k] * -ricT[k * 3 + j]; v += red[a * 3 + k] * S[k * 3 + j]; }
                B.g_J[jo + j * ld + a] = u * sr; B.g_J[jo + (3 + j) * ld + a] = v * sr;
            }
        }
        jo = B.s_joff[s0 + 2];
        if (jo >= 0) {
            for (int a = 0; a < 2; a++) for (int j = 0; j < 3; j++) {
                double u = 0;
                for (int k = 0; k < 3; k++) u += red[a * 3 + k] * A[k * 3 + j];
                B.g_J[jo + j * ld + a] = u * sr;
            }
        }
        return;
    }
    {
        // inverse-depth projection factor (2 residual rows): record = kind | pts_i (3) | pts_j (3); slots in the reference's block order
        const double* dat = B.gx_dat + G.data;
        const int kind = (int)dat[0];
        const double zero7[7] = { 0, 0, 0, 0, 0, 0, 1 };
        const int s_ex = kind == 2 ? 0 : 2, s_ex2 = kind == 0 ? -1 : (kind == 1 ? 3 : 1), s_l = G.nslot - 1;
        const double* Pi = kind == 2 ? zero7 : xs + B.s_x[s0];
        const double* Pj = kind == 2 ? zero7 : xs + B.s_x[s0 + 1];
        const double* ex = xs + B.s_x[s0 + s_ex];
        const double* e2 = kind == 0 ? ex : xs + B.s_x[s0 + s_ex2];
        double r2[2], Jq[50];
        d_idepth_eval(kind, Pi, Pj, ex, e2, xs[B.s_x[s0 + s_l]], dat + 1, W.proj_sqrt_info, W.pbg, r2, Jq, JAC);
        // CauchyLoss as on the world-point projection factors: rho'' < 0 => scale r and J by sqrt(rho'), block cost = rho(s) / 2
        double sr = 1.0, cost, sq = r2[0] * r2[0] + r2[1] * r2[1];
        if (W.proj_loss_a > 0) {
            double b = W.proj_loss_a * W.proj_loss_a, c = 1.0 / b;
            double sum = 1.0 + sq * c, inv = 1.0 / sum;
            cost = 0.5 * b * log(sum);
            sr = sqrt(inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308);
        } else cost = 0.5 * sq;
        B.g_cost[f] = cost;
        if (JAC) {
            B.g_r[G.roff] = r2[0] * sr; B.g_r[G.roff + 1] = r2[1] * sr;
            // Jq blocks: pose_i | pose_j | ex | ex2 | lambda  ->  the slot this kind keeps each of them in (-1: absent).
            // The block loop is unrolled so that Jq is indexed by constants and stays in registers.
#pragma unroll
            for (int bq = 0; bq < 5; bq++) {
                int t = kind == 1 ? bq : (kind == 0 ? (bq == 3 ? -1 : (bq == 4 ? 3 : bq)) : bq - 2);
                if (t < 0) continue;
                int jo = B.s_joff[s0 + t];
                if (jo < 0) continue;
                if (bq == 4) { B.g_J[jo] = Jq[48] * sr; B.g_J[jo + 1] = Jq[49] * sr; }
                else {
#pragma unroll
                    for (int j = 0; j < 6; j++) { B.g_J[jo + j * ld] = Jq[bq * 12 + j] * sr; B.g_J[jo + j * ld + 1] = Jq[bq * 12 + 6 + j] * sr; }
                }
            }
        }
    }
}

// =========================================================================================
// MarginalizationFactor::Evaluate (R/factor/marginalization_factor.cpp:410-446).
// r = r0 + J dx; the Jacobian is the constant J, so its J^T J (the prior clique's C) was
// formed once at upload; per evaluation only r and J^T r are produced.  One workgroup/prior.
// =========================================================================================
__device__ __forceinline__ void prior_block_dx(const double* x, const double* x0, int gs, double* dx) {
    if (gs != 7) { for (int k = 0; k < gs; k++) dx[k] = x[k] - x0[k]; return; }
    dx[0] = x[0] - x0[0]; dx[1] = x[1] - x0[1]; dx[2] = x[2] - x0[2];
    double q0i[4], dq[4];
    qinv(x0 + 3, q0i);
    qmul(q0i, x + 3, dq);
    double sg = (dq[3] >= 0) ? 2.0 : -2.0;
    dx[3] = sg * dq[0]; dx[4] = sg * dq[1]; dx[5] = sg * dq[2];
}
#define PRIOR_LDS_DIM 512               // priors up to this dimension ride as a segment of the fused evaluation grids
#define PRIOR_SPLIT_DIM 96              // priors beyond this dimension are evaluated in row chunks, a workgroup each
#define PRIOR_CHUNK 32
template <bool JAC>
__device__ __forceinline__ void d_eval_prior(const DevBatch& B, int blk, double* sm) {      // sm: dx[n] | r[n] | red[16]
    if (blk >= B.n_pch) return;
    const int q = B.pch_q[blk], i_lo = B.pch_r0[blk];
    const bool chunked = B.prior_nch[q] > 1;
    int f = B.prior_gf[q];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    if (!eval_gate<JAC>(B, s)) return;
    const double* xs = eval_src<JAC>(B);
    int k = G.data, n = G.nres;
    double* dx = sm; double* rr = sm + n; double* red = sm + 2 * n;
    const double* Jp = B.prior_J + B.prior_Joff[k];
    const double* r0 = B.prior_r0 + B.prior_roff[k];
    const double* x0 = B.prior_x0 + B.prior_x0off[k];
    // one thread per kept block computes its dx segment (x0 offsets are prefix sums of sizes)
    for (int sl = threadIdx.x; sl < G.nslot; sl += blockDim.x) {
        int col = B.s_pcol[G.slot0 + sl], xo = B.s_pxo[G.slot0 + sl];        // host-built prefix sums
        int l = B.s_ls[G.slot0 + sl];
        double tmp[9];
        prior_block_dx(xs + B.s_x[G.slot0 + sl], x0 + xo, l == 6 ? 7 : l, tmp);
        for (int j = 0; j < l; j++) dx[col + j] = tmp[j];
    }
    __syncthreads();
    // r = r0 + J dx: FOUR lanes per row, each over every fourth column with four loads in flight, the quad's partial sums added in a fixed
    // order ((s0 + s1) + (s2 + s3)).  (A thread per row walked its n columns one dependent L2 round trip after the other: 11.3 us for the twenty
    // 40-row records of a reference-topology window, the whole launch.)
    double part = 0;
    {
        const int q = threadIdx.x & 3, rq = threadIdx.x >> 2, rpb = blockDim.x >> 2;
        const int i_hi = chunked ? (i_lo + PRIOR_CHUNK < n ? i_lo + PRIOR_CHUNK : n) : n;        // this workgroup's rows
        for (int i0 = i_lo; i0 < i_hi; i0 += rpb) {
            const int i = i0 + rq < i_hi ? i0 + rq : n;
            double a = 0;
            if (i < n) {
                const double* row = Jp + (size_t)i * n;
                int j = q;
                for (; j + 12 < n; j += 16) {
                    const double x0 = row[j], x1 = row[j + 4], x2 = row[j + 8], x3 = row[j + 12];
                    a += x0 * dx[j]; a += x1 * dx[j + 4]; a += x2 * dx[j + 8]; a += x3 * dx[j + 12];
                }
                for (; j < n; j += 4) a += row[j] * dx[j];
            }
            const double b = a + __shfl_xor(a, 1, 64);
            const double c = b + __shfl_xor(b, 2, 64);
            if (i < n && q == 0) {
                const double v = r0[i] + c;
                rr[i] = v; part += v * v;
                if (JAC) B.g_r[G.roff + i] = v;
            }
        }
    }
    double tot = block_sum(part, red);
    if (threadIdx.x == 0) B.pr_cpart[blk] = 0.5 * tot;
    if (!JAC) return;
    const Clique& C = B.cl[G.clique];
    if (chunked && (C.is_static || i_lo > 0)) return;     // static: graw = J^T r needs every chunk's rows (k_prior_graw); else: the first chunk copies the record
    if (!C.is_static) {
        // a prior-type record inside the clique of a group-0 block (a composite factor on an eliminated speed-bias block): its columns
        // go into the clique's dense column-major Jacobian, next to the other factors' rows; J^T J, J^T r and the elimination are the
        // clique kernel's
        // (one pass over the transposed record — column j contiguous, the layout of the clique's Jacobian — with four loads in flight, instead
        // of a loop nest per block whose strided loads each waited for the stores before them: 14 dependent round trips for a 40-column record)
        int* cmap = (int*)dx;                     // dx is dead behind block_sum's barriers: the g_J offset of every column of the record, -1 = not in this clique
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) cmap[j] = -1;
        __syncthreads();
        for (int sl = threadIdx.x; sl < G.nslot; sl += blockDim.x) {
            const int jo = B.s_joff[G.slot0 + sl], l = B.s_ls[G.slot0 + sl], col = B.s_pcol[G.slot0 + sl];
            if (jo >= 0) for (int j = 0; j < l; j++) cmap[col + j] = jo + j * G.jld;
        }
        __syncthreads();
        const double* Jt = B.prior_Jt + B.prior_Joff[k];
        const int tot = n * n, nt = blockDim.x;
        for (int e0 = threadIdx.x; e0 < tot; e0 += 4 * nt) {
            double v[4]; int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u * nt;
                const int j = e < tot ? e / n : 0;
                const int cj = e < tot ? cmap[j] : -1;
                dst[u] = cj >= 0 ? cj + (e - j * n) : -1;
                v[u] = cj >= 0 ? Jt[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (dst[u] >= 0) B.g_J[dst[u]] = v[u];
        }
        return;
    }
    // graw = J^T r into the prior clique's vector slot (members are the kept blocks in order)
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double a = 0;
        for (int i = 0; i < n; i++) a += Jp[(size_t)i * n + j] * rr[i];
        // column j of the prior -> member column (constant blocks are not members): host-built map
        int mc = B.prior_colcc[B.prior_roff[k] + j];
        if (mc >= 0) B.cv_graw[C.v_off + mc] = a;
    }
}

// graw = J^T r of a chunked prior with a static clique: a workgroup per chunk of PRIOR_CHUNK COLUMNS, behind the evaluation that left r
// in g_r (a launch of its own, only when a batch holds such priors).  Thread (rg, jc): rows rg, rg + 8, ... of column jc; the eight
// partial sums of a column added in a fixed order.
__device__ __forceinline__ void d_prior_graw(const DevBatch& B, int blk) {
    __shared__ double ps[8][PRIOR_CHUNK + 1];
    if (blk >= B.n_pch) return;
    const int q = B.pch_q[blk], j_lo = B.pch_r0[blk];
    if (B.prior_nch[q] <= 1) return;
    const GFac& G = B.gf[B.prior_gf[q]];
    const WinState& s = B.ws[G.win];
    if (!s.need_lin) return;
    const Clique& C = B.cl[G.clique];
    if (!C.is_static) return;
    const int k = G.data, n = G.nres, jc = threadIdx.x & (PRIOR_CHUNK - 1), rg = threadIdx.x / PRIOR_CHUNK, j = j_lo + jc;
    const double* Jp = B.prior_J + B.prior_Joff[k];
    const double* r = B.g_r + G.roff;
    double a = 0;
    if (j < n) {
        int i = rg;
        for (; i + 24 < n; i += 32) {
            const double x0 = Jp[(size_t)i * n + j], x1 = Jp[(size_t)(i + 8) * n + j], x2 = Jp[(size_t)(i + 16) * n + j], x3 = Jp[(size_t)(i + 24) * n + j];
            a += x0 * r[i]; a += x1 * r[i + 8]; a += x2 * r[i + 16]; a += x3 * r[i + 24];
        }
        for (; i < n; i += 8) a += Jp[(size_t)i * n + j] * r[i];
    }
    ps[rg][jc] = a;
    __syncthreads();
    if (rg == 0 && j < n) {
        double t = ((ps[0][jc] + ps[1][jc]) + (ps[2][jc] + ps[3][jc])) + ((ps[4][jc] + ps[5][jc]) + (ps[6][jc] + ps[7][jc]));
        const int mc = B.prior_colcc[B.prior_roff[k] + j];
        if (mc >= 0) B.cv_graw[C.v_off + mc] = t;
    }
}
__global__ void __launch_bounds__(256) k_prior_graw(DevBatch B) { d_prior_graw(B, (int)blockIdx.x); }

// stand-alone launch for priors larger than PRIOR_LDS_DIM (dynamic LDS)
template <bool JAC>
__global__ void __launch_bounds__(256) k_eval_prior(DevBatch B) {
    extern __shared__ double sm_dyn[];
    d_eval_prior<JAC>(B, blockIdx.x, sm_dyn);
}

// =========================================================================================
// J * v per factor.  MODE 0: v = D^-2 g (Cauchy point, DoglegStrategy::ComputeCauchyPoint),
// aux = |J v|^2.  MODE 1: v = step, aux = (Jv).(r + Jv/2) (model cost change,
// TrustRegionMinimizer::ComputeTrustRegionStep).
// =========================================================================================
template <int MODE>
__device__ __forceinline__ double vec_at(const DevBatch& B, const DevOpt& O, int loc) {
    if (MODE == 1) return B.step[loc];
    return B.vc[loc];                  // g / clamp(diag), stored by the producers of g and diag
}
// row k of (J v) for a generic factor
template <int MODE>
__device__ __forceinline__ double gf_row_dot(const DevBatch& B, const DevOpt& O, const GFac& G, int k) {
    double a = 0;
    if (G.type == GF_PRIOR) {
        // row k of the transposed record: lanes over rows read adjacent elements; the columns are walked flat through the
        // host-built column -> local index map, eight loads in flight, additions in column order (constant columns skipped)
        const int n = G.nres;
        const double* ck = B.prior_Jt + B.prior_Joff[G.data] + k;
        const int* cl = B.prior_colloc + B.prior_roff[G.data];
        int c = 0;
        for (; c + 8 <= n; c += 8) {
            int lo[8]; double cv[8], vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) lo[u] = cl[c + u];
#pragma unroll
            for (int u = 0; u < 8; u++) { cv[u] = ck[(size_t)(c + u) * n]; vv[u] = lo[u] >= 0 ? vec_at<MODE>(B, O, lo[u]) : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) if (lo[u] >= 0) a += cv[u] * vv[u];
        }
        for (; c < n; c++) { int lo = cl[c]; if (lo >= 0) a += ck[(size_t)c * n] * vec_at<MODE>(B, O, lo); }
    } else {
        for (int t = 0; t < G.nslot; t++) {
            int jo = B.s_joff[G.slot0 + t];
            if (jo < 0) continue;
            int l = B.s_ls[G.slot0 + t], lo = B.s_loc[G.slot0 + t];
            const double* col0 = B.g_J + jo + k;                 // element (k, j) of the block: column stride G.jld
            for (int j0 = 0; j0 < l; j0 += 9) {                  // nine guarded loads in flight, additions in column order
                double c[9], v[9];
#pragma unroll
                for (int j = 0; j < 9; j++) { bool ok = j0 + j < l; c[j] = ok ? col0[(j0 + j) * G.jld] : 0.0; v[j] = ok ? vec_at<MODE>(B, O, lo + j0 + j) : 0.0; }
#pragma unroll
                for (int j = 0; j < 9; j++) if (j0 + j < l) a += c[j] * v[j];
            }
        }
    }
    return a;
}
template <int MODE>
__device__ __forceinline__ double gf_row_term(const DevBatch& B, const GFac& G, int k, double a) {
    return MODE == 0 ? a * a : a * (B.g_r[G.roff + k] + a / 2.0);
}
// scalar (one-row) factors: one lane each
template <int MODE>
__device__ __forceinline__ void d_jtimes_scalar(const DevBatch& B, const DevOpt& O, int bid) {
    int q = bid * blockDim.x + threadIdx.x;
    if (q >= B.n_sc) return;
    int f = B.sc_gf[q];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    if (MODE == 0 ? !s.need_lin : !s.eval_cand) return;
    double a = 0;
    for (int k = 0; k < G.nres; k++) a += gf_row_term<MODE>(B, G, k, gf_row_dot<MODE>(B, O, G, k));     // nres = 1, or 2 for the inverse-depth projections
    B.g_aux[f] = a;
}
// IMU factors: 16 lanes per factor, one residual row per lane
template <int MODE>
__device__ __forceinline__ void d_jtimes_imu(const DevBatch& B, const DevOpt& O, int bid) {
    int q = (bid * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    bool valid = q < B.n_imu;
    int f = B.imu_gf[valid ? q : B.n_imu - 1];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    bool act = valid && (MODE == 0 ? s.need_lin : s.eval_cand);
    double part = 0;
    if (act && sub < 15) part = gf_row_term<MODE>(B, G, sub, gf_row_dot<MODE>(B, O, G, sub));
    part = grp16_sum(part);
    if (act && sub == 0) B.g_aux[f] = part;
}
// priors: one 256-thread workgroup per prior, threads over residual rows.  The vector entries at the prior's columns (the same for every
// row) are staged in LDS once together with the column -> local index map, so a row's dot product is n coalesced loads of the transposed
// record and LDS broadcasts: the stress window's 263-dimension prior took 100 us as one wavefront whose lanes each gathered v per column.
// Sums: columns in order inside a row (constant columns skipped), rows by wave butterfly, waves in order — for priors of up to 64 rows
// (every configuration but the stress window) bit-identical to the one-wavefront form.
#define PRB_MAX 512
template <int MODE>
__device__ __forceinline__ void d_jtimes_prior(const DevBatch& B, const DevOpt& O, int blk, double* sv, int* sl, double* sw) {
    const int tid = threadIdx.x;
    if (blk >= B.n_pch) return;
    const int q = B.pch_q[blk], k_lo = B.pch_r0[blk];
    const bool chunked = B.prior_nch[q] > 1;
    int f = B.prior_gf[q];
    const GFac& G = B.gf[f];
    const WinState& s = B.ws[G.win];
    if (MODE == 0 ? !s.need_lin : !s.eval_cand) return;        // uniform per block
    const int n = G.nres;
    const bool staged = n <= PRB_MAX;
    const int* cl = B.prior_colloc + B.prior_roff[G.data];
    if (staged) for (int c = tid; c < n; c += 256) { int lo = cl[c]; sl[c] = lo; sv[c] = lo >= 0 ? vec_at<MODE>(B, O, lo) : 0.0; }
    __syncthreads();
    double part = 0;
    if (chunked && staged) {
        // a chunk of PRIOR_CHUNK rows: eight lanes per row over every eighth column (row-major record: a row's lanes read one run),
        // the eight partial sums of a row added in a fixed order
        const int p = tid & 7, k = k_lo + (tid >> 3);
        double a = 0;
        if (k < n) {
            const double* row = B.prior_J + B.prior_Joff[G.data] + (size_t)k * n;
            int c = p;
            for (; c + 24 < n; c += 32) {
                const double x0 = row[c], x1 = row[c + 8], x2 = row[c + 16], x3 = row[c + 24];
                if (sl[c] >= 0) a += x0 * sv[c];
                if (sl[c + 8] >= 0) a += x1 * sv[c + 8];
                if (sl[c + 16] >= 0) a += x2 * sv[c + 16];
                if (sl[c + 24] >= 0) a += x3 * sv[c + 24];
            }
            for (; c < n; c += 8) if (sl[c] >= 0) a += row[c] * sv[c];
        }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        if (k < n && p == 0) part = gf_row_term<MODE>(B, G, k, a);
    } else {
        const int k_hi = chunked ? (k_lo + PRIOR_CHUNK < n ? k_lo + PRIOR_CHUNK : n) : n;
        for (int k = k_lo + tid; k < k_hi; k += 256) {
            double a;
            if (staged) {
                const double* ck = B.prior_Jt + B.prior_Joff[G.data] + k;
                a = 0;
                int c = 0;
                for (; c + 8 <= n; c += 8) {
                    double cv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cv[u] = ck[(size_t)(c + u) * n];
#pragma unroll
                    for (int u = 0; u < 8; u++) if (sl[c + u] >= 0) a += cv[u] * sv[c + u];
                }
                for (; c < n; c++) if (sl[c] >= 0) a += ck[(size_t)c * n] * sv[c];
            } else a = gf_row_dot<MODE>(B, O, G, k);
            part += gf_row_term<MODE>(B, G, k, a);
        }
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) sw[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) B.pr_apart[blk] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}

#include "swf_lmschur.h"

// =========================================================================================
// Clique elimination: every group-0 block that is not a landmark (alternate speed-biases,
// receiver clocks, the dummy) together with the factors touching it, and every group of
// factors that touches no group-0 block.  One workgroup per clique builds the small dense
// J^T J over [e | members] in LDS, eliminates e, and leaves
//   C = M_ff - M_fe Einv M_ef,  graw = J_f^T r,  dgraw = diag(M_ff),  cs = -M_fe Einv g_e
// for k_assemble, plus Einv / M_ef / g_e for the back-substitution.
// =========================================================================================
#define CLQ_MAXD 64
#define CLQ_MAXR 64
#ifdef SWF_PROFILE_CLQ
__device__ unsigned long long g_clq_stamps[16];
#ifndef SWF_PROFILE_CLQ_IDX
#define SWF_PROFILE_CLQ_IDX 0
#endif
#define QST(i) do { if (cidx_ == SWF_PROFILE_CLQ_IDX && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_clq_stamps[i] += t_ - tq_; tq_ = t_; } } while (0)
#else
#define QST(i)
#endif
// size classes (host-assigned): 0 = d_e<=1, <=48 rows, <=32 cols (receiver clocks, dummy, small free groups);
// 1 = <=32 rows, <=48 cols (speed-bias cliques); 2 = up to 64 x 64.
// ONE WAVEFRONT PER CLIQUE (the kernel is latency-bound: what matters is how many cliques a CU keeps in flight).
// The wave is synchronous with itself, so the phases need no workgroup barriers, and M_ff is never stored:
// every lane forms 2x2 blocks of M_ff = J_f^T J_f in registers and applies the Schur correction in place.
//   lane c:      column c of M_e* = J_e^T J  and  g_c = J_c^T r
//   lanes < d_e: one row of [M_ee + mu D | I] each, Gauss-Jordan through v_readlane broadcasts of the pivot row
//   lane j:      column j of T = Einv M_ef
//   2x2 blocks:  C = M_ff - M_fe T  written straight to HBM
// PF: values per lane and round of the Jacobian gather (24 in the stand-alone kernel; the fused grid k_lm_clique lives under the 128
// registers of a 1024-thread workgroup and takes fewer).
// NW: wavefronts per clique.  1 = the throughput form (a CU keeps many cliques in flight).  4 = the latency form (few windows): the
// phases whose work is a set of independent output elements — the gather, the rows of M_e*, the 2x2 blocks of C — are dealt over
// 4 waves; the Gauss-Jordan inverse and the columns of T stay on wave 0.  Every output element is still formed by ONE lane with the
// same operands in the same order, so the two forms give the same bits (tested: a window alone vs inside a large batch).
template <int MAXR, int MAXD, int MAXE, int CLS, int PF = 24, int NW = 1, int UK = 4>      // UK: unrolling of the k loops (register budget)
__device__ __forceinline__ void d_clique_elim(const DevBatch& B, const DevOpt& O, const int cidx_) {
    constexpr int LD = MAXD + 1, NT = 64 * NW;
    __shared__ double Jc[MAXR][LD];                 // dense clique Jacobian: rows = residual rows, cols = [e | members]
    // (Me / T are walked along their rows by the lanes, Ei is read as broadcasts: no padding column — the 216 bytes put the speed-bias
    // class at 20 448 B, eight cliques per CU instead of seven)
    __shared__ double Me[MAXE][MAXD];               // M_e* = J_e^T J (rows of M that belong to e)
    __shared__ double T[MAXE][MAXD];                // Einv M_ef
    __shared__ double Ei[MAXE][MAXE];               // Einv
    __shared__ double rv[MAXR];
    __shared__ double Eg[MAXE];
    __shared__ double gcs[NW > 1 ? MAXD : 1];       // (NW > 1) g_c of every column, for the waves that do not hold it in registers
    __shared__ int bad_sh;
#ifdef SWF_PROFILE_CLQ
    unsigned long long tq_ = __builtin_amdgcn_s_memtime();
    if (cidx_ == SWF_PROFILE_CLQ_IDX && threadIdx.x == 0) for (int i = 0; i < 16; i++) g_clq_stamps[i] = 0;
#endif
    if (cidx_ >= B.n_clc[CLS]) return;
    const Clique& C = B.clc_rec[CLS][cidx_];
    WinState& s = B.ws[C.win];
    if (!s.need_lin) return;
    const int tid = threadIdx.x, lane = tid & 63, wq = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    int de = C.d_e, df = C.d_f, d = de + df, nrow = C.n_rows;
    QST(0);
    // the clique's Jacobian is dense and column-major in HBM (the factor kernels write their blocks at their (row, column)
    // position; the structural zeros are static): no gather lists, no indirection
    {
        const double* gJ = B.g_J + C.j_off;
        int tot = nrow * d;
        float rd = 1.0f / (float)nrow;
        if (tid < nrow) rv[tid] = B.g_r[C.r_off + tid];
        for (int k = tid; k < nrow; k += NT) Jc[k][d] = 0.0;                      // zero pad column (odd d, 2x2 blocks)
        if (NW > 1 && tid == 0) bad_sh = 0;
        // flat, fully coalesced loads, all issued before the first use (one exposed round trip per PF values)
        for (int base = 0; base < tot; base += PF * NT) {
            double v[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) { int e = base + u * NT + tid; v[u] = e < tot ? gJ[e] : 0.0; }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                int e = base + u * NT + tid;
                int col = (int)((e + 0.5f) * rd);                                    // exact floor(e / nrow) for e < 4096, nrow <= 64
                col += (e - col * nrow >= nrow) ? 1 : 0; col -= (e - col * nrow < 0) ? 1 : 0;   // (belt and braces)
                if (e < tot) Jc[e - col * nrow][col] = v[u];
            }
        }
    }
    __syncthreads();
    QST(1);
    // lane c < d: column c of M_e* (k-ascending dot products), gradient entry g_c, and the diagonal M_cc.  NW > 1: wave q forms the rows
    // a2 = q, q + NW, ... of M_e*; g_c and M_cc are wave 0's.
    double me[MAXE], gc = 0, mcc = 0;
#pragma unroll
    for (int a2 = 0; a2 < MAXE; a2++) me[a2] = 0;
    {
        int c = lane < d ? lane : 0;
        if (NW == 1) {
#pragma unroll UK
            for (int k = 0; k < nrow; k++) {
                double x = Jc[k][c];
                gc += x * rv[k]; mcc += x * x;
#pragma unroll
                for (int a2 = 0; a2 < MAXE; a2++) me[a2] += Jc[k][a2] * x;      // (rows a2 >= d_e are computed and dropped)
            }
        } else {
            constexpr int RPW = (MAXE + NW - 1) / NW;                          // rows of M_e* per wave
#pragma unroll UK
            for (int k = 0; k < nrow; k++) {
                double x = Jc[k][c];
                if (wq == 0) { gc += x * rv[k]; mcc += x * x; }
#pragma unroll
                for (int u = 0; u < RPW; u++) { int a2 = wq + u * NW; if (a2 < MAXE) me[u] += Jc[k][a2] * x; }
            }
        }
    }
    if (lane < d) {
        // rows >= d_e of Me / Ei / T are kept at zero so the inner loops below need no d_e guards
        if (NW == 1) {
#pragma unroll
            for (int a2 = 0; a2 < MAXE; a2++) Me[a2][lane] = a2 < de ? me[a2] : 0.0;
        } else {
            constexpr int RPW = (MAXE + NW - 1) / NW;
#pragma unroll
            for (int u = 0; u < RPW; u++) { int a2 = wq + u * NW; if (a2 < MAXE) Me[a2][lane] = a2 < de ? me[u] : 0.0; }
        }
        if (wq == 0) {
            if (NW > 1) gcs[lane] = gc;
            if (lane < de) { B.g[C.e_loc + lane] = gc; B.diag[C.e_loc + lane] = mcc; B.vc[C.e_loc + lane] = gc / clampd(mcc, O.min_diag, O.max_diag); }
            else { B.cv_graw[C.v_off + lane - de] = gc; B.cv_dgraw[C.v_off + lane - de] = mcc; }
        }
    }
    __syncthreads();
    QST(2);
    if (de > 0) {
        if (wq == 0) {
            // [M_ee + mu D | I] -> [I | Einv] by Gauss-Jordan (SPD: no pivoting).  Lane r < d_e keeps row r in registers;
            // step k broadcasts the pivot row through SGPRs (v_readlane), so there is no LDS traffic and no barrier.
            double row[2 * MAXE];
#pragma unroll
            for (int j = 0; j < MAXE; j++) {
                double v = (lane < de && j < de) ? Me[lane][j] : 0.0;
                if (j == lane) v += s.mu * (lane < de ? damp_diag(O, v, B.jsc + C.e_loc + lane, s.iter == 0) : clampd(v, O.min_diag, O.max_diag));
                row[j] = v; row[MAXE + j] = (j == lane) ? 1.0 : 0.0;
            }
            bool bad = false;
#pragma unroll
            for (int k = 0; k < MAXE; k++) {
                if (k < de) {
                    double piv = readlane_d(row[k], k);
                    if (!(piv > 0.0)) bad = true;
                    // reciprocal by v_rcp_f64 + 2 Newton steps instead of the IEEE division expansion
                    double ip = __builtin_amdgcn_rcp(piv);
                    ip = ip * (2.0 - piv * ip); ip = ip * (2.0 - piv * ip);
                    double aik = row[k];
#pragma unroll
                    for (int j = 0; j < 2 * MAXE; j++) {
                        double akj = readlane_d(row[j], k) * ip;
                        row[j] = (lane == k) ? akj : row[j] - aik * akj;
                    }
                }
            }
            if (NW == 1) { if (bad) { if (lane == 0) s.lin_fail = 1; return; } }
            else if (bad && lane == 0) { s.lin_fail = 1; bad_sh = 1; }
            if (lane < MAXE) {
#pragma unroll
                for (int j = 0; j < MAXE; j++) Ei[lane][j] = (lane < de && j < de) ? row[MAXE + j] : 0.0;
            }
        }
        __syncthreads();
        if (NW > 1 && bad_sh) return;
        QST(3);
        // lane j < d_f: column j of T = Einv M_ef; lanes < d_e: Eg = Einv g_e
        if (wq == 0) {
            double gE = 0;
            // g_e entries sit in lanes < d_e (gc): broadcast them
            int j = lane < df ? lane : 0;
            if (NW == 1) {
                double tcol[MAXE];
#pragma unroll
                for (int a2 = 0; a2 < MAXE; a2++) {
                    double sv = 0, sg = 0;
#pragma unroll
                    for (int b2 = 0; b2 < MAXE; b2++) { double ev = Ei[a2][b2]; sv += ev * Me[b2][de + j]; sg += ev * readlane_d(gc, b2); }
                    tcol[a2] = sv;
                    if (lane == a2) gE = sg;
                }
                if (lane < df) {
#pragma unroll
                    for (int a2 = 0; a2 < MAXE; a2++) T[a2][lane] = tcol[a2];
                }
            } else {
                // (latency form, 128-register cap: the column of M_ef and g_e once in registers, a row of Einv at a time — the same sums
                // in the same order; the fully unrolled form above keeps all 81 entries of Einv in flight)
                double mcol[MAXE], gb[MAXE];
#pragma unroll
                for (int b2 = 0; b2 < MAXE; b2++) { mcol[b2] = Me[b2][de + j]; gb[b2] = readlane_d(gc, b2); }
#pragma unroll 1
                for (int a2 = 0; a2 < MAXE; a2++) {
                    double sv = 0, sg = 0;
#pragma unroll
                    for (int b2 = 0; b2 < MAXE; b2++) { double ev = Ei[a2][b2]; sv += ev * mcol[b2]; sg += ev * gb[b2]; }
                    if (lane < df) T[a2][lane] = sv;
                    if (lane == a2) gE = sg;
                }
            }
            if (lane < de) Eg[lane] = gE;
        }
        __syncthreads();
        QST(4);
        double* E = B.cE + C.e_off;
        if (NW == 1 || wq == 1) {
            // (NW > 1: the back-substitution record is wave 1's, while wave 0 starts on its blocks of C)
            for (int e = lane; e < de * de; e += 64) E[e] = Ei[e / de][e % de];
            if (lane < df) {
#pragma unroll
                for (int a2 = 0; a2 < MAXE; a2++) if (a2 < de) E[de * de + a2 * df + lane] = Me[a2][de + lane];
            }
            if (lane < de) E[de * de + de * df + lane] = NW == 1 ? gc : gcs[lane];
            // cs_j = -(M_fe Eg)_j
            if (lane < df) {
                double v = 0;
#pragma unroll
                for (int a2 = 0; a2 < MAXE; a2++) v -= Me[a2][de + lane] * (a2 < de ? Eg[a2] : 0.0);
                B.cv_cs[C.v_off + lane] = v;
            }
        }
    } else {
        if (wq == 0) {
            if (lane < df) B.cv_cs[C.v_off + lane] = 0.0;
#pragma unroll
            for (int a2 = 0; a2 < MAXE; a2++) T[a2][lane < MAXD ? lane : 0] = 0.0;
        }
        __syncthreads();
    }
    // C = M_ff - M_fe T in 2x2 blocks of the lower triangle: M_ff block from Jc (k-ascending sums), correction from Me / T
    double* Cm = B.C + C.C_off;
    {
        // two blocks per lane and pass (eight independent accumulators keep the LDS pipe busy); NW > 1: the blocks dealt over all the waves
        int nb = (df + 1) >> 1, nblk = nb * (nb + 1) / 2;
        for (int t0 = tid; t0 < nblk; t0 += 2 * NT) {
            int i0[2], j0[2];
            bool diag[2], on[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                int t = t0 + NT * u;
                on[u] = t < nblk;
                if (!on[u]) t = 0;
                int ba = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
                while ((ba + 1) * (ba + 2) / 2 <= t) ba++;
                while (ba * (ba + 1) / 2 > t) ba--;
                int bb = t - ba * (ba + 1) / 2;
                i0[u] = 2 * ba; j0[u] = 2 * bb; diag[u] = ba == bb;
            }
            double m[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll UK
            for (int k = 0; k < nrow; k++) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    double xa0 = Jc[k][de + i0[u]], xa1 = Jc[k][de + i0[u] + 1], xb0 = Jc[k][de + j0[u]], xb1 = Jc[k][de + j0[u] + 1];
                    m[u][0] += xa0 * xb0; m[u][1] += xa0 * xb1; m[u][2] += xa1 * xb0; m[u][3] += xa1 * xb1;
                }
            }
            // (the latency form lives under a 128-register cap inside k_lm_clique: its correction loop runs over the d_e live rows without
            // unrolling — the rows beyond are zero, the terms they would add are exact zeros — instead of 72 operand loads in flight)
#pragma unroll(NW > 1 ? 1 : MAXE)
            for (int a2 = 0; a2 < (NW > 1 ? de : MAXE); a2++) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    double f0 = Me[a2][de + i0[u]], f1 = Me[a2][de + i0[u] + 1], t0_ = T[a2][j0[u]], t1_ = T[a2][j0[u] + 1];
                    m[u][0] -= f0 * t0_; m[u][1] -= f0 * t1_; m[u][2] -= f1 * t0_; m[u][3] -= f1 * t1_;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!on[u]) continue;
                int i = i0[u], j = j0[u];
                bool i1ok = i + 1 < df, j1ok = j + 1 < df;
                Cm[i * df + j] = m[u][0]; Cm[j * df + i] = m[u][0];
                if (i1ok) { Cm[(i + 1) * df + j] = m[u][2]; Cm[j * df + i + 1] = m[u][2]; }
                if (!diag[u] && j1ok) { Cm[i * df + j + 1] = m[u][1]; Cm[(j + 1) * df + i] = m[u][1]; }
                if (i1ok && j1ok) { Cm[(i + 1) * df + j + 1] = m[u][3]; Cm[(j + 1) * df + i + 1] = m[u][3]; }
            }
        }
    }
    QST(5);
}
template <int MAXR, int MAXD, int MAXE, int CLS>
__global__ void __launch_bounds__(64) k_clique_elim(DevBatch B, DevOpt O) { d_clique_elim<MAXR, MAXD, MAXE, CLS>(B, O, (int)blockIdx.x); }
// the latency form as a kernel of its own: four waves per clique (batches too large for the fused grid below, but still on the latency path)
__global__ void __launch_bounds__(256) k_clique_elim4(DevBatch B, DevOpt O) { d_clique_elim<64, 64, 9, 2, 8, 4>(B, O, (int)blockIdx.x); }
// class 4: up to 96 rows x 64 columns — the clique of a speed-bias block that two composite IMU-GNSS factors touch (the reference's own
// ordering puts every other speed-bias block into group 0, R/swf/swf_gnss.cpp:683-691: 2 x (30 + N) rows, 9 + 6 + 9 + 6 + 6 + 9 + N
// columns: N <= 18 ambiguities fit 96 rows x 64 columns, more take k_clique_big).  The four-wave form of the same function.
#define CLQ_TALLR 96
__global__ void __launch_bounds__(256) k_clique_tall(DevBatch B, DevOpt O) { d_clique_elim<CLQ_TALLR, 64, 9, 4, 8, 4>(B, O, (int)blockIdx.x); }

// Latency path: the landmark Schur complement and the clique eliminations are independent of each other (both follow the factor
// evaluation, both feed the assembly) — ONE grid of 1024-thread workgroups runs both: rows [0, n_parts) of the grid are k_lm_schur's
// workgroups, the rows behind them take one clique each on their first four wavefronts (the other waves leave at once).  One window: 13.5 +
// 15.4 us as two dependent launches, the longer of the two as one.  Same device functions: bit-identical results.
template <int NCW, int TPW, int TW, int LDR>
__global__ void __launch_bounds__(LS_NT(NCW, TW)) k_lm_clique(DevBatch B, DevOpt O, int qpb, int lp, int kms, int s_direct, int n_parts) {
    if ((int)blockIdx.y < n_parts) d_lm_schur<NCW, TPW, TW, LDR, true>(B, O, qpb, lp, kms, s_direct, (int)blockIdx.x, (int)blockIdx.y);
    else {
        // Waves 4 .. 15 END here, ahead of the __syncthreads() inside d_clique_elim<..., NW = 4>.  That leans on a property of the gfx9
        // hardware barrier, not of the HIP programming model: s_barrier counts the waves of the workgroup that have NOT terminated (CDNA ISA,
        // "S_BARRIER": terminated waves are not waited for), so the four surviving waves synchronise among themselves.  Written for
        // gfx950 only (the file has no other target); the surviving-wave count is tied to the clique function's NW below.
        constexpr int CLQ_NW = 4;
        static_assert(CLQ_NW * 64 == 256 && LS_NT(NCW, TW) >= CLQ_NW * 64, "k_lm_clique: the clique rows keep exactly the waves d_clique_elim<..., NW> synchronises");
        if (threadIdx.x >= CLQ_NW * 64) return;
        d_clique_elim<64, 64, 9, 2, 8, 4, CLQ_NW>(B, O, ((int)blockIdx.y - n_parts) * (int)gridDim.x + (int)blockIdx.x);
    }
}

// =========================================================================================
// Cliques beyond one wavefront's reach (more than 64 residual rows or 64 columns): the group-0 block of a landmark with a
// long track on the generic path — an inverse-depth feature seen from ten or more frames (2 rows and 6 columns per
// frame), or a world-point landmark whose factors also touch a VARIABLE camera extrinsic, as in the reference's
// marginalisation solves (GlobalMarge un-freezes para_ex_Pose, R/swf/swf_image.cpp:384-389).  Same outputs as
// k_clique_elim, one 256-thread workgroup per clique, the dense column-major Jacobian staged in LDS when it fits and read
// from L2 otherwise.  This is the general path, not the fast one: the fast paths are k_lm_schur (world points, constant
// extrinsic) and k_clique_elim (everything up to 64 x 64).
//   thread c:       column c of M_e* = J_e^T J, g_c = J_c^T r, M_cc
//   thread 0:       (M_ee + mu D)^-1 by Gauss-Jordan (d_e <= 9)
//   thread j:       column j of T = Einv M_ef
//   thread (i, j):  C_ij = J_i . J_j - M_ei . T_j over the lower triangle
// =========================================================================================
#define CB_MAXD 768                           // columns of a big clique (d_e + d_f)
#define CB_MAXED 1536                         // d_e x (d_e + d_f) of a big clique: the rows of M that belong to e, and T = Einv M_ef, live in LDS
#define CB_LDS_J 12288                        // doubles of Jacobian staged in LDS (96 KB); larger ones are read through L2
#define CB_NT 1024
// Round 5: rewritten for the cliques the reference's RTK topology produces with more than 18 ambiguities (two composite factors on an
// eliminated speed-bias block: 2 (30 + N) rows x 45 + N columns; N = 24: 307 us per launch, half of that window's iteration).  The
// old form ran phase 1 on d threads, the 9 x 9 Gauss-Jordan on ONE thread out of a private array (scratch memory), and re-formed T_j for
// every entry of C.  Now: 1024 threads; phase 1 deals the d (d_e + 2) dot products one per thread; the inverse is wave 0's, a row per
// lane, pivot rows through v_readlane (the code of d_clique_elim); T = Einv M_ef once, into LDS; an entry of C is one thread's
// J_i . J_j - M_ei . T_j.  Sums in the same order as before.
//   thread (c, p):  M_e*[p][c] = J_p . J_c (p < d_e), g_c = J_c . r, M_cc
//   wave 0:         (M_ee + mu D)^-1 by Gauss-Jordan (d_e <= 9), Eg = Einv g_e
//   thread (a, j):  T[a][j] = Einv[a][:] . M_e*[:][d_e + j]
//   thread (i, j):  C_ij = J_i . J_j - M_ei . T_j over the lower triangle
// (STAGED is a template parameter so that every pointer into the Jacobian has ONE address space: with a run-time choice between the
// LDS copy and L2 the loads were flat_load — 100 us per launch for ten 108 x 69 cliques, most of it the latency of un-pipelined flat loads)
template <bool STAGED>
__device__ __forceinline__ void d_clique_big(const DevBatch& B, const DevOpt& O, double* Me, double* Tt, double (*Ei)[9], double* Eg, double* ge, int* bad_sp, double* Jl) {
#define bad_s (*bad_sp)
    if ((int)blockIdx.x >= B.n_clc[3]) return;
    const Clique& C = B.clc_rec[3][blockIdx.x];
    WinState& s = B.ws[C.win];
    if (!s.need_lin) return;
    const int de = C.d_e, df = C.d_f, d = de + df, nrow = C.n_rows, tid = threadIdx.x, lane = tid & 63;
    const double* gJ = B.g_J + C.j_off;
    const double* rv = B.g_r + C.r_off;
    // (staged columns are padded to an odd length: threads that walk different columns in step read distinct banks — with the
    // Jacobian's own column length, 108 rows for two 54-row factors, eight of them shared one)
    constexpr bool staged = STAGED;
    const int ldj = staged ? (nrow | 1) : nrow;
    if (staged) for (int e = tid; e < nrow * d; e += CB_NT) { const int c = e / nrow, k2 = e - c * nrow; Jl[c * ldj + k2] = gJ[e]; }
    if (tid == 0) bad_s = 0;
    __syncthreads();
    const double* __restrict__ Jc = STAGED ? (const double*)Jl : gJ;
    // 1. M_e*, gradient, diagonal: one dot product per thread and round (k ascending)
    for (int e = tid; e < d * (de + 2); e += CB_NT) {
        const int c = e / (de + 2), p = e - c * (de + 2);
        const double* col = Jc + (size_t)c * ldj;
        double acc = 0;
        if (p < de) { const double* ce = Jc + (size_t)p * ldj;
#pragma unroll 4
            for (int k = 0; k < nrow; k++) acc += ce[k] * col[k];
            Me[p * d + c] = acc; }
        else if (p == de) {
#pragma unroll 4
            for (int k = 0; k < nrow; k++) acc += col[k] * rv[k];
            if (c < de) { B.g[C.e_loc + c] = acc; ge[c] = acc; } else B.cv_graw[C.v_off + c - de] = acc;
        } else {
#pragma unroll 4
            for (int k = 0; k < nrow; k++) acc += col[k] * col[k];
            if (c < de) B.diag[C.e_loc + c] = acc; else B.cv_dgraw[C.v_off + c - de] = acc;
        }
    }
    __syncthreads();
    if (tid < de) B.vc[C.e_loc + tid] = ge[tid] / clampd(Me[tid * d + tid], O.min_diag, O.max_diag);       // (M_cc of an e column is M_e*[c][c]: the same sum)
    double* Cm = B.C + C.C_off;
    if (de > 0) {
        // 2. Einv = (M_ee + mu D)^-1: [M_ee + mu D | I] -> [I | Einv] by Gauss-Jordan (SPD: no pivoting), wave 0, lane r < d_e keeps row r in
        //    registers, step k broadcasts the pivot row through SGPRs (v_readlane): no LDS traffic, no barrier
        if (tid < 64) {
            constexpr int MAXE = 9;
            double row[2 * MAXE];
#pragma unroll
            for (int j = 0; j < MAXE; j++) {
                double v = (lane < de && j < de) ? Me[lane * d + j] : 0.0;
                if (j == lane) v += s.mu * (lane < de ? damp_diag(O, v, B.jsc + C.e_loc + lane, s.iter == 0) : clampd(v, O.min_diag, O.max_diag));
                row[j] = v; row[MAXE + j] = (j == lane) ? 1.0 : 0.0;
            }
            bool bad = false;
#pragma unroll
            for (int k = 0; k < MAXE; k++) {
                if (k < de) {
                    double piv = readlane_d(row[k], k);
                    if (!(piv > 0.0)) bad = true;
                    double ip = __builtin_amdgcn_rcp(piv);
                    ip = ip * (2.0 - piv * ip); ip = ip * (2.0 - piv * ip);
                    double aik = row[k];
#pragma unroll
                    for (int j = 0; j < 2 * MAXE; j++) {
                        double akj = readlane_d(row[j], k) * ip;
                        row[j] = (lane == k) ? akj : row[j] - aik * akj;
                    }
                }
            }
            if (bad && lane == 0) { s.lin_fail = 1; bad_s = 1; }
            if (lane < MAXE) {
#pragma unroll
                for (int j = 0; j < MAXE; j++) Ei[lane][j] = (!bad && lane < de && j < de) ? row[MAXE + j] : 0.0;
            }
        }
        __syncthreads();
        if (bad_s) return;
        if (tid < 9) { double v = 0; for (int j = 0; j < de; j++) v += Ei[tid][j] * ge[j]; Eg[tid] = v; }
        // 3. T = Einv M_ef
        for (int e = tid; e < de * df; e += CB_NT) {
            const int a = e / df, j = e - a * df;
            double tj = 0;
            for (int b2 = 0; b2 < de; b2++) tj += Ei[a][b2] * Me[b2 * d + de + j];
            Tt[a * df + j] = tj;
        }
        __syncthreads();
        // what the back-substitution needs (Einv | M_ef | g_e) and cs = -M_fe Einv g_e
        double* E = B.cE + C.e_off;
        for (int e = tid; e < de * de; e += CB_NT) E[e] = Ei[e / de][e % de];
        for (int j = tid; j < df; j += CB_NT) {
            double v = 0;
            for (int a = 0; a < de; a++) { E[de * de + a * df + j] = Me[a * d + de + j]; v -= Me[a * d + de + j] * Eg[a]; }
            B.cv_cs[C.v_off + j] = v;
        }
        if (tid < de) E[de * de + de * df + tid] = ge[tid];
    } else {
        for (int j = tid; j < df; j += CB_NT) B.cv_cs[C.v_off + j] = 0.0;
    }
    // 4. C = M_ff - M_fe Einv M_ef over the lower triangle (mirrored on write)
    const long long ntri = (long long)df * (df + 1) / 2;
    for (long long t = tid; t < ntri; t += CB_NT) {
        int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(i + 1) * (i + 2) / 2 <= t) i++;
        while ((long long)i * (i + 1) / 2 > t) i--;
        int j = (int)(t - (long long)i * (i + 1) / 2);
        const double* ci = Jc + (size_t)(de + i) * ldj; const double* cj = Jc + (size_t)(de + j) * ldj;
        double m = 0;
#pragma unroll 4
        for (int k = 0; k < nrow; k++) m += ci[k] * cj[k];
        for (int a = 0; a < de; a++) m -= Me[a * d + de + i] * Tt[a * df + j];
        Cm[(size_t)i * df + j] = m; Cm[(size_t)j * df + i] = m;
    }
}
#undef bad_s
__global__ void __launch_bounds__(CB_NT) k_clique_big(DevBatch B, DevOpt O) {
    __shared__ double Me[CB_MAXED];           // [d_e][d]
    __shared__ double Tt[CB_MAXED];           // [d_e][d_f]
    __shared__ double Ei[9][9];
    __shared__ double Eg[9], ge[9];
    __shared__ int bad_sv;
    __shared__ double Jl[CB_LDS_J];           // the staged Jacobian [d][nrow | 1], when it fits
    if ((int)blockIdx.x >= B.n_clc[3]) return;
    const Clique& C = B.clc_rec[3][blockIdx.x];
    if ((long long)(C.n_rows | 1) * (C.d_e + C.d_f) <= CB_LDS_J) d_clique_big<true>(B, O, Me, Tt, Ei, Eg, ge, &bad_sv, Jl);
    else d_clique_big<false>(B, O, Me, Tt, Ei, Eg, ge, &bad_sv, Jl);
}
// Latency path of a reference-topology window with 19 or more ambiguities: its speed-bias cliques are class 3 (above), the others class 2 —
// two dependent launches of independent work.  ONE grid: workgroups [0, n_clc[3]) as above, the ones behind them take a class-2 clique on their
// first four waves (the form and the early exit of k_lm_clique).  The LDS of both functions must fit one workgroup: the staging buffer is
// 10 KB shorter here (a Jacobian of 11 008 .. 12 288 doubles is read through L2 instead: the same sums in the same order).
#define CB_LDS_J2 11008
__global__ void __launch_bounds__(CB_NT) k_clique_big2(DevBatch B, DevOpt O) {
    if ((int)blockIdx.x >= B.n_clc[3]) {
        static_assert(CB_NT >= 256, "k_clique_big2: the clique rows keep the four waves d_clique_elim<..., NW = 4> synchronises");
        if (threadIdx.x >= 256) return;          // (terminated waves are not waited for by s_barrier: see k_lm_clique)
        d_clique_elim<64, 64, 9, 2, 8, 4, 4>(B, O, (int)blockIdx.x - B.n_clc[3]);
        return;
    }
    __shared__ double Me[CB_MAXED];
    __shared__ double Tt[CB_MAXED];
    __shared__ double Ei[9][9];
    __shared__ double Eg[9], ge[9];
    __shared__ int bad_sv;
    __shared__ double Jl[CB_LDS_J2];
    const Clique& C = B.clc_rec[3][blockIdx.x];
    if ((long long)(C.n_rows | 1) * (C.d_e + C.d_f) <= CB_LDS_J2) d_clique_big<true>(B, O, Me, Tt, Ei, Eg, ge, &bad_sv, Jl);
    else d_clique_big<false>(B, O, Me, Tt, Ei, Eg, ge, &bad_sv, Jl);
}

// =========================================================================================
// Per-frame raw sums over the projection observations, level 1 of a two-level FIXED-ORDER
// reduction.  Observations are landmark-major in memory, so a pose's observations are strided;
// instead of gathering them (one 64-byte line per 8-byte value, PMC-measured 8x amplification),
// every block of 256 consecutive observations is read coalesced, staged in LDS, and reduced per
// frame through a host-built frame-sorted permutation of the block.  Output per (block, frame):
//   27 doubles = lower(Jp^T Jp)(21) | Jp^T r (6);   k_assemble<true> adds the blocks in order.  (The landmark part of the
//   reduced right-hand side, sum Y g_l per frame, comes out of k_lm_schur's matrix-core pass: DevBatch::lmq.)
// =========================================================================================
#define FS_BLK 256
#define FS_VAL 27
#define FS_HALF 14                            // values staged per pass
// The same, fused into the Jacobian evaluation (k_eval_ps<true>, one workgroup per frame-sum block): thread t evaluates observation t of
// the block and the 27 products never leave the chip — Jp and r are not read back (112 B per observation and one launch less).
__device__ __forceinline__ void d_eval_proj_fs(const DevBatch& B, int blk, double (*V)[FS_HALF], int* foff) {
    int w = B.fsb_win[blk];
    if (!eval_gate<true>(B, B.ws[w])) return;                // uniform per block (a block holds observations of one window)
    const WinRec& W = B.win[w];
    int o_beg = B.fsb_obs0[blk], cnt = B.fsb_obs0[blk + 1] - o_beg, tid = threadIdx.x;
    int nF = W.nF;
    for (int e = tid; e <= nF; e += FS_BLK) foff[e] = B.fsb_foff[B.fsb_foff0[blk] + e];
    double keep[20];
#pragma unroll
    for (int k = 0; k < 20; k++) keep[k] = 0.0;
    int rk = 0;
    double cost = 0.0;
    if (tid < cnt) { rk = B.fsb_perm[o_beg + tid]; cost = d_eval_proj_at<true>(B, o_beg + tid, keep); }
    {   // the block's cost (what the per-window control kernels add up instead of per-observation costs)
        __shared__ double csum[16];
        cost = block_sum(cost, csum);
        if (tid == 0) B.p_cpart[blk] = cost;
    }
    double val[FS_VAL];
    {
        const double* a = keep; const double* b = keep + 6;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) val[k++] = a[i] * a[j] + b[i] * b[j];
#pragma unroll
        for (int i = 0; i < 6; i++) val[21 + i] = a[i] * keep[12] + b[i] * keep[13];
    }
    double* out = B.fs_part + (size_t)B.fsb_out0[blk] * FS_VAL;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int v0 = half * FS_HALF, nv = half == 0 ? FS_HALF : FS_VAL - FS_HALF;
        if (half) __syncthreads();                          // the first half's sums are done
        if (tid < cnt) {
#pragma unroll
            for (int k = 0; k < FS_HALF; k++) if (k < nv) V[rk][k] = val[v0 + k];
        }
        __syncthreads();
        for (int e = tid; e < nF * nv; e += FS_BLK) {
            int f = e / nv, v = e - f * nv;
            double acc = 0;
            int q = foff[f], q1 = foff[f + 1];
            for (; q + 4 <= q1; q += 4) {
                double x0 = V[q][v], x1 = V[q + 1][v], x2 = V[q + 2][v], x3 = V[q + 3][v];
                acc += x0; acc += x1; acc += x2; acc += x3;
            }
            for (; q < q1; q++) acc += V[q][v];
            out[f * FS_VAL + v0 + v] = acc;
        }
    }
}

// =========================================================================================
// Assembly of the reduced system as a static PROGRAM (round 3).  What every entry of S, and every reduced dimension of g / diag /
// rhs, receives is known at build time: the host flattens it into per-entry lists (window-relative offsets into the clique blocks C,
// the landmark product P or the -P k_lm_schur left in S, the frame-sum partials, the clique vectors, the landmark rhs partials) and
// one thread per entry adds its list in order.  Same sums, same order as the pair-walking kernel it replaces (k_assemble_all: one
// wavefront per block pair, descriptor -> value load chains, 142 us per launch over 512 cfg3 windows):
//   S_ab[i][j] = sum_cliques C_k[i][j]  ( + (-P) already in place  |  - sum_parts P_q )  + [a == b observing pose] sum_blocks H
//                + [i == j] mu damp(diag_i)
//   g_i = sum_blocks (Jp^T r)_i + sum_cliques graw,  diag_i = sum_blocks H_ii + sum_cliques dgraw,  rhs_i = g_i + sum_cliques cs - sum_parts q
// grid = (blocks of 256 entries, windows); blocks [0, nbS) walk S entries (write_S only), the rest vector entries.
// =========================================================================================
// sum of base[src[0 .. n)] in list order; indices and values are fetched four at a time (the additions keep their order)
__device__ __forceinline__ double as_sum(const double* base, const int* src, int n, int add = 0) {
    double v = 0;
    int c = 0;
    for (; c + 4 <= n; c += 4) {
        const int i0 = src[c], i1 = src[c + 1], i2 = src[c + 2], i3 = src[c + 3];
        const double x0 = base[i0 + add], x1 = base[i1 + add], x2 = base[i2 + add], x3 = base[i3 + add];
        v += x0; v += x1; v += x2; v += x3;
    }
    for (; c < n; c++) v += base[src[c] + add];
    return v;
}
__global__ void __launch_bounds__(256) k_assemble_flat(DevBatch B, DevOpt O, int write_S, int nbS) {
    const AsmWin& A = B.asw[blockIdx.y];
    const WinState& s = B.ws[A.win];
    if (!s.need_lin) return;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nbS) {
        if (!write_S) return;
        const int e = blockIdx.x * 256 + tid;
        if (e >= A.ne) return;
        const int k = A.se0 + e;
        const unsigned cnt = B.as_cnt[k];
        const int nC = AS_NC(cnt), nP = AS_NP(cnt), nH = AS_NH(cnt);
        const int* src = B.as_src + B.as_src0[k];
        const int dst = B.as_dst[k];
        double* S = B.S + A.S_base;
        const double* C = B.C + A.C_base;
        const double sold = AS_SOLD(cnt) ? S[dst] : 0.0;    // (requested up front: it does not depend on the lists)
        double v = as_sum(C, src, nC);
        if (AS_SOLD(cnt)) v = sold + v;                    // k_lm_schur left -P in place: -P + c == c - P
        else if (nP) v -= as_sum(B.P + A.P_base, src + nC, nP);
        double hs = 0;
        if (nH) { hs = as_sum(B.fs_part + (size_t)A.fs_base * FS_VAL, src + nC + nP, nH); v += hs; }
        if (AS_DIAG(cnt)) {
            // dg = hs + the cliques' raw diagonals, in contribution order
            const double* dgr = B.cv_dgraw + A.v_base;
            const int* sd = src + nC + nP + nH;
            double dg = hs;
            int c = 0;
            for (; c + 4 <= nC; c += 4) { const double x0 = dgr[sd[c]], x1 = dgr[sd[c + 1]], x2 = dgr[sd[c + 2]], x3 = dgr[sd[c + 3]]; dg += x0; dg += x1; dg += x2; dg += x3; }
            for (; c < nC; c++) dg += dgr[sd[c]];
            v += s.mu * damp_diag(O, dg, B.jsc + A.loc_base + B.as_aux[k], s.iter == 0);
        }
        S[dst] = v;
        return;
    }
    const int e = (blockIdx.x - nbS) * 256 + tid;
    if (e >= A.nv) return;
    const int k = A.ve0 + e;
    const unsigned cnt = B.av_cnt[k];
    const int nC = AS_NC(cnt), nQ = AS_NP(cnt), nH = AS_NH(cnt);
    const int* src = B.av_src + B.av_src0[k];
    double gi = 0, dg = 0, cs = 0;
    if (nH) {
        const double* fs = B.fs_part + (size_t)A.fs_base * FS_VAL;
        const int i = B.av_i[k];
        const double g0 = as_sum(fs, src, nH, 21 + i), h0 = as_sum(fs, src, nH, i * (i + 1) / 2 + i);
        const double q0 = as_sum(B.lmq + A.q_base, src + nH + nC, nQ);
        gi = g0; dg = h0; cs = -q0;
    }
    {
        const double* gr = B.cv_graw + A.v_base; const double* dgr = B.cv_dgraw + A.v_base; const double* csr = B.cv_cs + A.v_base;
        int c = 0;
        for (; c + 2 <= nC; c += 2) {
            const int v0 = src[nH + c], v1 = src[nH + c + 1];
            const double a0 = gr[v0], b0 = dgr[v0], c0_ = csr[v0], a1 = gr[v1], b1 = dgr[v1], c1_ = csr[v1];
            gi += a0; dg += b0; cs += c0_; gi += a1; dg += b1; cs += c1_;
        }
        for (; c < nC; c++) { const int vo = src[nH + c]; gi += gr[vo]; dg += dgr[vo]; cs += csr[vo]; }
    }
    const int loc = A.loc_base + B.av_loc[k];
    B.g[loc] = gi; B.diag[loc] = dg; B.vc[loc] = gi / clampd(dg, O.min_diag, O.max_diag);
    // the reduced rhs is kept twice: in the local-space vector, and as row n of the window's S storage
    // (the Cholesky carries it as one more tile row with the same addressing as every other tile)
    if (write_S) { B.rhs[loc] = gi + cs; B.S[A.S_base + (size_t)A.n_red * A.n_red + B.av_red[k]] = gi + cs; }
}
