// swf_kernels4.h — the composite IMU-GNSS factor (8f rank 2) as a batched device operator.
//
// IMUGNSSBase (R/factor/gnss_imu_factor.cpp) hides the GNSS-epoch states between two visual frames: at every Jacobian
// evaluation it (i) back-substitutes the hidden epochs from the outer increment (UpdateHiddenState :601-632), (ii) rebuilds
// the block-tridiagonal normal equations epoch by epoch — IMUFactor::Evaluate2 (R/factor/imu_factor.cpp:103-195), the epoch's
// linearised GNSS prior (:775-778) — eliminating each epoch as it goes (MargPose1 :403-433, MoveHessianData :435-452), and
// (iii) turns the remaining (30+N)^2 system over [pose_i sb_i | pose_j sb_j | N ambiguities] into a residual / Jacobian pair
// (UpdateSchurComponent :454-488); cost-only evaluations use the linear model r = r_lin - J INC (:490-497).
//
// Here: three kernels (k_comp_prep, k_comp_imu, k_comp_elim); in the elimination one 256-thread workgroup per factor, the ten Hessian blocks in LDS, the epochs
// sequential (they are a chain), everything inside an epoch spread over the threads; composite factors of all windows in
// one launch.  The square root of the (30+N)^2 remainder is its Cholesky factor (J = L^T, r = L^-1 rhs): J^T J and J^T r —
// all a Gauss-Newton solver consumes — are those of the reference's eigen square root whenever the remainder is positive
// definite (certified by the factorisation; a failure is reported per factor), at a fraction of a symmetric eigensolve.
#pragma once
#include "swf_kernels.h"

#define CO_MAXN 64                         // ambiguities per composite factor (the reference's data model: 3 constellations x NFREQ 2 on up to MAXOBS 64
                                           // satellites, R/gnss/include/common_function.h:24-37; 30..48 per gap is a normal open-sky epoch)
#define CO_TINYN 12                        // batches: a third instantiation for <= 12 ambiguities (25 KB of LDS and 84 registers: six workgroups per CU; the chain of a factor is latency-bound, so what a batch gains is factors in flight)
#define CO_SMALLN 24                       // k_comp_elim / k_comp_eigroot are instantiated for <= 24 (36 KB of LDS, four workgroups per CU) and <= 64 ambiguities
#define CO_MAXG (30 + CO_MAXN)

struct CompArgs {
    int n;
    const int* M; const int* N;                        // [n]
    const int* e_off; const int* n_off;                // [n+1] prefix sums of M, N
    const long long* pn_off; const long long* nn_off;  // [n+1] prefix sums of 15 M N, N N
    const long long* g_off; const long long* g2_off;   // [n+1] prefix sums of G, G G
    double* pose; double* sb;                          // hidden epochs [sum M][7], [sum M][9]
    const double* pose_lin; const double* sb_lin;
    const double* Hpp; const double* HpN; const double* rhs_p; const double* HNN; const double* rhsN;
    const double* pre;                                 // [sum M + n][SWF_PRE_DOUBLES]; factor f owns records e_off[f] + f ...
    const double* pbgw;                                // [n][6] lever arm and gravity of the factor's window
    const int* active;                                 // [n] or null: 0 = skip the factor in this launch (its window does not re-linearise)
    double* hmn_inv; double* hmn_2; double* hmn_0; double* hmn_N; double* rhsmn;      // saved at each elimination
    double* Hd; double* rd; double* Ld; double* r0;                                      // dense remainder, its factor, L^-1 rhs
    double* old; double* N_old;                        // [n][32] outer states of the last linearisation, [sum N]
    int* history; int* status;
    const double* outer; const double* Nv;             // this call: [n][32] = pose_i sb_i pose_j sb_j, [sum N]
    int want_jac;
    double* res_out; double* jac_out;                  // [sum G], [sum G G] (row-major: row r = v_r of the pivoted square root, zero beyond the rank)
    double* Jw; double* rw;                            // scratch: whitened IMU Jacobians [sum M + n][450] and residuals [sum M + n][16]
    const int* iq_f; const int* iq_k; int n_iq;        // the chains' IMU factors, flattened: owner factor and position k (0 .. M)
    int* todo;                                         // [n] set by k_comp_prep: 1 = this launch re-eliminates the factor
    const int* mid; const double* H12;                 // [n], [n][225]: middle-marginalisation link (AddMidMargInfo :121-240): link mid[f] in 1..M-1
                                                       // carries the cross block H12[f] instead of an IMU factor; 0 = none
};

// x (-) x0 = sgn [p - p0, +-2 vec(q0^-1 q), sb - sb0]   (GetInc :654-670, UpdateDeltaValues :560-598 with sgn = -1)
__device__ __forceinline__ void co_inc15(const double* P, const double* Bv, const double* P0, const double* B0, double sgn, double* dx) {
    double q0i[4], dq[4];
    for (int k = 0; k < 3; k++) dx[k] = sgn * (P[k] - P0[k]);
    qinv(P0 + 3, q0i); qmul(q0i, P + 3, dq);
    double s2 = (dq[3] >= 0) ? 2.0 : -2.0;
    for (int k = 0; k < 3; k++) dx[3 + k] = sgn * s2 * dq[k];
    for (int k = 0; k < 9; k++) dx[6 + k] = sgn * (Bv[k] - B0[k]);
}

// phase 1 of an evaluation: increments, the cost-only answer, or the back-substitution of the hidden epochs
__device__ __forceinline__ bool d_comp_prep(const CompArgs& A, const int f) {
    const int t = threadIdx.x;
    if (f >= A.n) return false;
    if (A.active && !A.active[f]) { if (t == 0) A.todo[f] = 0; return false; }
    const int M = A.M[f], N = A.N[f], G = 30 + N, e0 = A.e_off[f], n0 = A.n_off[f];
    const long long pn0 = A.pn_off[f], nn0 = A.nn_off[f], g0 = A.g_off[f], g20 = A.g2_off[f];
    __shared__ double dl2[15], dlN[CO_MAXN], dl0[15];                      // delta5[Pose2], [N], [Pose0]
    __shared__ double sOut[32], sNv[CO_MAXN], sOld[32], sNold[CO_MAXN], sDx[16], sRm[16];
    constexpr int CP_MAXM = 16;                         // epochs whose increments are kept for the parallel state update (256 threads / 16 lanes)
    __shared__ double sDxAll[CP_MAXM * 16];
    const int hist = A.history[f];
    if (t < 32) { sOut[t] = A.outer[(size_t)f * 32 + t]; sOld[t] = hist ? A.old[(size_t)f * 32 + t] : A.outer[(size_t)f * 32 + t]; }
    if (t < N) { sNv[t] = A.Nv[n0 + t]; sNold[t] = hist ? A.N_old[n0 + t] : A.Nv[n0 + t]; }
    __syncthreads();
    const double* Pi = sOut; const double* Bi = sOut + 7; const double* Pj = sOut + 16; const double* Bj = sOut + 23;
    // UpdateDeltaValues: increments old (-) new
    if (t == 0) co_inc15(Pj, Bj, sOld + 16, sOld + 23, -1.0, dl2);
    if (t == 64) co_inc15(Pi, Bi, sOld, sOld + 7, -1.0, dl0);
    if (t >= 128 && t - 128 < N) dlN[t - 128] = sNold[t - 128] - sNv[t - 128];
    __syncthreads();
    const int update = A.want_jac != 0;
    if (hist && !update) {
        // UpdateJacobResidual, cost-only: r = r_lin - J INC with J = L^T, INC = [dl0 | dl2 | dlN]
        const double* Ld = A.Ld + g20;
        for (int k = t; k < G; k += 256) {
            double s = 0;
            for (int a = 0; a < G; a++) { double inc = a < 15 ? dl0[a] : a < 30 ? dl2[a - 15] : dlN[a - 30]; s += Ld[(size_t)a * G + k] * inc; }      // (column k of Ld is row k of J; not triangular: pivoted factor)
            A.res_out[g0 + k] = A.r0[g0 + k] - s;
        }
        if (t == 0) A.todo[f] = 0;
        return false;
    }
    if (t == 0) A.todo[f] = 1;
    if (hist && update) {
        // UpdateHiddenState: newest epoch first; delta5[Pose2] becomes the epoch's own increment for its older neighbour
        for (int i = M - 1; i >= 0; i--) {
            const double* h2 = A.hmn_2 + (size_t)(e0 + i) * 225; const double* h0 = A.hmn_0 + (size_t)(e0 + i) * 225;
            const double* hN = A.hmn_N + pn0 + (size_t)i * 15 * N; const double* hi = A.hmn_inv + (size_t)(e0 + i) * 225;
            if (t < 15) {
                double s = A.rhsmn[(size_t)(e0 + i) * 15 + t];
                for (int k = 0; k < 15; k++) s -= h2[t * 15 + k] * dl2[k];
                for (int k = 0; k < N; k++) s -= hN[t * N + k] * dlN[k];
                for (int k = 0; k < 15; k++) s -= h0[t * 15 + k] * dl0[k];
                sRm[t] = s;
            }
            __syncthreads();
            if (t < 15) { double s = 0; for (int k = 0; k < 15; k++) s += hi[t * 15 + k] * sRm[k]; sDx[t] = s; }
            __syncthreads();
            if (t < 15) { dl2[t] = sDx[t]; if (i < CP_MAXM) sDxAll[i * 16 + t] = sDx[t]; }
            if (i >= CP_MAXM) {
                // (chains beyond CP_MAXM epochs: the epoch's states move inside the loop, a memory round trip per epoch)
                double* P = A.pose + (size_t)(e0 + i) * 7; double* Bv = A.sb + (size_t)(e0 + i) * 9;
                if (t == 32) {
                    for (int k = 0; k < 3; k++) P[k] -= sDx[k];
                    double q[4], dq[4] = { -sDx[3] / 2, -sDx[4] / 2, -sDx[5] / 2, 1.0 };
                    qmul(P + 3, dq, q);
                    double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                    for (int k = 0; k < 4; k++) P[3 + k] = q[k] / nq;
                }
                if (t >= 64 && t < 73) Bv[t - 64] -= sDx[6 + t - 64];
            }
            __syncthreads();
        }
        // the epochs' states move AFTER the chain, all at once (round 5: the chain needs only the increments; with the read-modify-write of
        // an epoch's pose inside the loop every epoch waited a memory round trip at its last barrier): sixteen lanes per epoch, lane 0 the
        // pose, lanes 1 .. 9 the speed-bias entries — the same operations on the same operands as inside the loop
        {
            const int ep = t >> 4, ln = t & 15;
            if (ep < M && ep < CP_MAXM) {
                const double* dx = sDxAll + ep * 16;
                double* P = A.pose + (size_t)(e0 + ep) * 7; double* Bv = A.sb + (size_t)(e0 + ep) * 9;
                if (ln == 0) {
                    for (int k = 0; k < 3; k++) P[k] -= dx[k];
                    double q[4], dq[4] = { -dx[3] / 2, -dx[4] / 2, -dx[5] / 2, 1.0 };
                    qmul(P + 3, dq, q);
                    double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                    for (int k = 0; k < 4; k++) P[3 + k] = q[k] / nq;
                } else if (ln <= 9) Bv[ln - 1] -= dx[6 + ln - 1];
            }
        }
        __threadfence_block();
    }
    return true;
}
__global__ void __launch_bounds__(256) k_comp_prep(CompArgs A) { (void)d_comp_prep(A, (int)blockIdx.x); }

// phase 2: every IMU factor of every chain that re-eliminates, IMUFactor::Evaluate2 — 8 factors per workgroup, the four un-whitened
// parts on one lane each (part p on wave p, as in k_eval_imu), then 32 lanes per factor whiten; whitened J (15 x 30) and r to scratch
__device__ __forceinline__ void d_comp_imu(const CompArgs& A, const int qb, const int qend) {
    __shared__ double U[8][450], SIs[8][225], raw[8][16], st[8][32], pr[8][SWF_PRE_SQRTINFO + 6];
    const bool on = threadIdx.x < 256;                  // (the fused kernel runs 1024 threads: the others only pass the barriers)
    const int t = on ? threadIdx.x : 0, fl = t >> 5, sub = t & 31;
    const int q = qb + fl;
    const bool valid = on && q < qend;
    const int f = A.iq_f[valid ? q : qend - 1], k = A.iq_k[valid ? q : qend - 1];
    const int M = A.M[f], e0 = A.e_off[f];
    // the link of a middle marginalisation has no IMU factor (Evaluate :738): its scratch rows stay zero, as allocated
    const bool act = valid && A.todo[f] && !(k > 0 && k == A.mid[f]);
    const double* pre = A.pre + (size_t)(e0 + f + k) * SWF_PRE_DOUBLES;
    if (act) {
        const double* outer = A.outer + (size_t)f * 32;
        {
            int sl = sub < 7 ? 0 : sub < 16 ? 1 : sub < 23 ? 2 : 3, o = sub < 7 ? sub : sub < 16 ? sub - 7 : sub < 23 ? sub - 16 : sub - 23;
            int hidx = sl < 2 ? k - 1 : k;                  // older state (frame_i for k = 0) / newer state (frame_j for k = M)
            double v;
            if (hidx < 0) v = outer[(sl == 0 ? 0 : 7) + o];
            else if (hidx >= M) v = outer[(sl == 2 ? 16 : 23) + o];
            else v = (sl & 1) ? A.sb[(size_t)(e0 + hidx) * 9 + o] : A.pose[(size_t)(e0 + hidx) * 7 + o];
            st[fl][sub] = v;
        }
        for (int e = sub; e < SWF_PRE_SQRTINFO; e += 32) pr[fl][e] = pre[e];
        if (sub < 6) pr[fl][SWF_PRE_SQRTINFO + sub] = A.pbgw[(size_t)f * 6 + sub];
        for (int e = sub; e < 450; e += 32) U[fl][e] = 0.0;
        for (int e = sub; e < 225; e += 32) SIs[fl][e] = pre[SWF_PRE_SQRTINFO + e];
    }
    __syncthreads();
    if (on && (t & 63) < 8) {
        int fq = t & 63, q2 = qb + fq;
        if (q2 < qend && A.todo[A.iq_f[q2]] && !(A.iq_k[q2] > 0 && A.iq_k[q2] == A.mid[A.iq_f[q2]]))
            imu_unwhitened(st[fq], st[fq] + 7, st[fq] + 16, st[fq] + 23, pr[fq], pr[fq] + SWF_PRE_SQRTINFO, pr[fq] + SWF_PRE_SQRTINFO + 3, raw[fq], U[fq], true, t >> 6);
    }
    __syncthreads();
    if (!act) return;
    // whitening as in imu_whiten_store: lane c < 30 owns column c of the 15x30 block (registers), SI (upper triangular) is read
    // as LDS broadcasts at constant offsets, k ascending from the row; lanes < 15 also whiten one residual row each
    const double* SI = SIs[fl];
    if (sub < 30) {
        double u[15];
#pragma unroll
        for (int j = 0; j < 15; j++) u[j] = U[fl][j * 30 + sub];
#pragma unroll
        for (int row = 0; row < 15; row++) {
            double a = 0;
#pragma unroll
            for (int j = row; j < 15; j++) a += SI[row * 15 + j] * u[j];
            A.Jw[(size_t)(e0 + f + k) * 450 + row * 30 + sub] = a;
        }
    }
    if (sub < 15) {
        double a = 0;
#pragma unroll
        for (int j = 0; j < 15; j++) a += SI[sub * 15 + j] * raw[fl][j];
        A.rw[(size_t)(e0 + f + k) * 16 + sub] = a;
    }
}
__global__ void __launch_bounds__(256) k_comp_imu(CompArgs A) { d_comp_imu(A, (int)blockIdx.x * 8, A.n_iq); }

// phase 3: re-elimination of the hidden epochs from the whitened IMU Jacobians of phase 2, remainder, square root
// NT = threads per factor: 256 for batches (four workgroups per CU), 1024 on the latency path (a window's ~20 factors have the chip to
// themselves, and with one wave per SIMD every phase below is bound by the instructions that wave issues: 13.6 k cycles for an epoch's
// T products and Schur update at 256 threads).  Every output element is one thread's, from the same operands in the same order:
// the thread count does not change a bit.
template <int NMAX, int NT>
__device__ __forceinline__ void d_comp_elim(const CompArgs& A, const int f, const int n_lo = 0) {
    const int t = threadIdx.x;
    if (f >= A.n || !A.todo[f]) return;
    // the instantiation is the FACTOR's (a launch of each covers a batch with factors of several classes: this one takes n_lo <= N <= NMAX):
    // a factor's arithmetic does not depend on what else is in its batch.  Up to CO_SMALLN ambiguities the instantiations differ in array
    // strides only (same sums, same bits: the latency path runs <CO_SMALLN, 1024> on what a batch gives to <CO_TINYN, 256>); beyond, the
    // square root is another algorithm (round 5), and the class boundary CO_SMALLN is the same everywhere.
    if (A.N[f] < n_lo || A.N[f] > NMAX) return;
    const int M = A.M[f], N = A.N[f], G = 30 + N, e0 = A.e_off[f], n0 = A.n_off[f];
    const long long pn0 = A.pn_off[f], nn0 = A.nn_off[f], g0 = A.g_off[f], g20 = A.g2_off[f];
    CHSTAMP(32);
    constexpr int POOL_A = 6 * 225 + 3 * 15 * NMAX + NMAX * NMAX, POOL_B = (31 + NMAX) * (31 + NMAX);      // (G + 1 rows of padded length G | 1)
    __shared__ double pool[POOL_A > POOL_B ? POOL_A : POOL_B];                         // ten elimination blocks, later the dense remainder + rhs row
    double* const H00 = pool; double* const H01 = pool + 225; double* const H03 = pool + 450; double* const H11 = pool + 675;
    double* const H13 = pool + 900; double* const H33 = pool + 1125; double* const H0N = pool + 1350; double* const H1N = H0N + 15 * NMAX;
    double* const HN3 = H1N + 15 * NMAX; double* const HNN = HN3 + 15 * NMAX;
    double* const sD = pool;
    __shared__ double r0b[15], r1b[15], rNb[NMAX], r3b[15];
    __shared__ double scratch[900 + 15 * NMAX];                            // sJ ; then Ainv | L | T2 | T0 | TN (15 x N)
    double* const sJ = scratch + 450;
    double* const sAinv = scratch; double* const sL = scratch + 225; double* const T2 = scratch + 450; double* const T0 = scratch + 675; double* const TN = scratch + 900;
    __shared__ double sRes[16], sOut[32], sNv[NMAX], sDx[16], sDx2[16];
    __shared__ int sBad;
    const int midk = A.mid[f]; const double* H12 = A.H12 + (size_t)f * 225;
    if (t < 32) sOut[t] = A.outer[(size_t)f * 32 + t];
    if (t < N) sNv[t] = A.Nv[n0 + t];
    if (t == 0) sBad = 0;
    __syncthreads();
    // ---- re-elimination at the current outer / hidden states
    for (int e = t; e < 225; e += NT) { H00[e] = 0; H01[e] = 0; H03[e] = 0; H11[e] = 0; H13[e] = 0; H33[e] = 0; }
    for (int e = t; e < 15 * N; e += NT) { H0N[e] = 0; H1N[e] = 0; HN3[e] = 0; }
    for (int e = t; e < N * N; e += NT) HNN[e] = A.HNN[nn0 + e];
    if (t < 15) { r0b[t] = 0; r1b[t] = 0; r3b[t] = 0; }
    __syncthreads();
    if (t < N) { double s = A.rhsN[n0 + t]; for (int k = 0; k < N; k++) s += HNN[t * N + k] * sNv[k]; rNb[t] = s; }       // UpdateRhsN
    // IMU factor k of the chain links state k-1 -> k (k = 0: frame_i -> e_0, k = M: e_M-1 -> frame_j)
    for (int k = 0; k <= M; k++) {
        CHSTAMP(33 + k);
        for (int e = t; e < 450; e += NT) sJ[e] = A.Jw[(size_t)(e0 + f + k) * 450 + e];
        if (t < 15) sRes[t] = A.rw[(size_t)(e0 + f + k) * 16 + t];
        if (k > 0 && t == 255)       // GetInc of the epoch that this factor completes (its GNSS prior is added in the same pass below)
            co_inc15(A.pose + (size_t)(e0 + k - 1) * 7, A.sb + (size_t)(e0 + k - 1) * 9, A.pose_lin + (size_t)(e0 + k - 1) * 7, A.sb_lin + (size_t)(e0 + k - 1) * 9, 1.0, sDx);
        const bool midl = midk > 0 && k == midk;        // Evaluate :738-759: this link is the cross term of a middle marginalisation (its IMU scratch is zero)
        if (midl && t == 254) co_inc15(A.pose + (size_t)(e0 + k) * 7, A.sb + (size_t)(e0 + k) * 9, A.pose_lin + (size_t)(e0 + k) * 7, A.sb_lin + (size_t)(e0 + k) * 9, 1.0, sDx2);
        __syncthreads();
        // JacobianResidualUpdateHessianRhs: Ja = columns 0..14 (older state), Jb = columns 15..29 (newer state)
        //   k = 0 : blocks (Pose0, Pose1):  H33 += Ja^T Ja, H03 += Jb^T Ja, H00 += Jb^T Jb
        //   k > 0 : blocks (Pose1, Pose2):  H00 += Ja^T Ja, H01 += Ja^T Jb, H11 += Jb^T Jb
        {
            double* Haa = k == 0 ? H33 : H00; double* Hx = k == 0 ? H03 : H01; double* Hbb = k == 0 ? H00 : H11;
            double* ra = k == 0 ? r3b : r0b; double* rb = k == 0 ? r0b : r1b;
            for (int e = t; e < 675; e += NT) {
                int blk = e / 225, ee = e - blk * 225, i = ee / 15, j = ee - i * 15;
                double s = 0;
                if (blk == 0) { for (int q = 0; q < 15; q++) s += sJ[q * 30 + i] * sJ[q * 30 + j]; Haa[ee] += s; }
                else if (blk == 2) { for (int q = 0; q < 15; q++) s += sJ[q * 30 + 15 + i] * sJ[q * 30 + 15 + j]; Hbb[ee] += s; }
                else if (k == 0) { for (int q = 0; q < 15; q++) s += sJ[q * 30 + 15 + i] * sJ[q * 30 + j]; Hx[ee] += s; }        // Jb^T Ja
                else { for (int q = 0; q < 15; q++) s += sJ[q * 30 + i] * sJ[q * 30 + 15 + j]; if (midl) s += H12[ee]; Hx[ee] += s; }    // Ja^T Jb (+ pose1_pose2_hessians)
            }
            if (t < 30) {
                int j = t < 15 ? t : t - 15; double s = 0;
                for (int q = 0; q < 15; q++) s += sJ[q * 30 + t] * sRes[q];
                if (midl) {                                 // rhs(Pose1) += H12 inc(e_k), rhs(Pose2) += H12^T inc(e_k-1)
                    if (t < 15) for (int q = 0; q < 15; q++) s += H12[j * 15 + q] * sDx2[q];
                    else for (int q = 0; q < 15; q++) s += H12[q * 15 + j] * sDx[q];
                }
                if (t < 15) ra[j] += s; else rb[j] += s;
            }
        }
        if (k == 0) { __syncthreads(); continue; }
        if (k == 1) CHSTAMP(40);
        // ---- epoch i = k - 1 is complete: its GNSS prior in the same pass (every element below is touched by the thread that
        // accumulated it above), then eliminate it
        const int i = k - 1;
        const double* Hpp = A.Hpp + (size_t)(e0 + i) * 225; const double* HpN = A.HpN + pn0 + (size_t)i * 15 * N;
        if (t < 15) {                                   // UpdateRhsPose + RhsUpdateRhs
            double s = A.rhs_p[(size_t)(e0 + i) * 15 + t];
            for (int q = 0; q < 15; q++) s += Hpp[t * 15 + q] * sDx[q];
            for (int q = 0; q < N; q++) s += HpN[t * N + q] * sNv[q];
            r0b[t] += s;
        }
        if (t >= 64 && t - 64 < N) { int a = t - 64; double s = 0; for (int q = 0; q < 15; q++) s += HpN[q * N + a] * sDx[q]; rNb[a] += s; }
        for (int e = t; e < 225; e += NT) H00[e] += Hpp[e];
        for (int e = t; e < 15 * N; e += NT) H0N[e] += HpN[e];
        __syncthreads();
        if (k == 1) CHSTAMP(41);
        // MargPose1: Ainv = (H00)^-1 (InvertPSDMatrix<15>, assume_full_rank).  Round 5: in-place Gauss-Jordan without pivoting (read the column
        // and the row of the pivot, write the step): its pivots are the Cholesky pivots d_j of the same matrix (a non-positive one
        // fails the factor as before), no square root, one reciprocal per column.
        // (Rounds 1-4: a Cholesky with a barrier per column, then fifteen threads each running a forward and a backward substitution — thirty
        // dependent IEEE divisions and 210 dependent multiply-adds on one lane: most of an epoch's 40 k cycles.)
        for (int e = t; e < 225; e += NT) { int a = e / 15, b = e - a * 15; sAinv[e] = (b >= a) ? H00[e] : H00[b * 15 + a]; }     // full symmetric from the upper triangle
        __syncthreads();
        if (t < 64) {
            // ONE wavefront, four entries per lane, no barrier inside: a step's reads (the entry, its row's entry in the pivot column, its
            // column's entry in the pivot row, the pivot) are all issued before its writes, and the LDS executes a wave's operations in
            // order (two workgroup barriers per column cost 11.9 k cycles per epoch at 1024 threads, 9 k at 256; this form 8 k)
            // (Measured and not kept, end of round 5: the symmetric sweep form of the same elimination on the 120 entries of the upper triangle,
            // two per lane — half the instructions per step: 10.9 k cycles per epoch against 9.7 k.  A step is a latency chain — the pivot's LDS
            // round trip, its Newton reciprocal, the update, the write the next pivot is read behind — not an instruction count.)
            int ea[4], eb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) { const int e = t + 64 * m; ea[m] = e < 225 ? e / 15 : 0; eb[m] = e < 225 ? e - (e / 15) * 15 : 0; }
            for (int j = 0; j < 15; j++) {
                double d = sAinv[j * 15 + j];
                double aij[4], aik[4], akj[4];
#pragma unroll
                for (int m = 0; m < 4; m++) { aij[m] = sAinv[ea[m] * 15 + eb[m]]; aik[m] = sAinv[ea[m] * 15 + j]; akj[m] = sAinv[j * 15 + eb[m]]; }
                asm volatile("" ::: "memory");
                if (!(d > 0.0)) { if (t == 0) sBad = 1; d = 1.0; }
                const double ip = rcp_nr(d);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const double nv = (ea[m] == j) ? (eb[m] == j ? ip : akj[m] * ip) : (eb[m] == j ? -(aik[m] * ip) : aij[m] - aik[m] * (akj[m] * ip));
                    if (t + 64 * m < 225) sAinv[ea[m] * 15 + eb[m]] = nv;
                }
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        if (k == 1) CHSTAMP(42);
        // T_blk = H0blk^T Ainv for blk = Pose2 (15), N, Pose0 (15)
        for (int e = t; e < 450 + 15 * N; e += NT) {
            const double* Hs; double* Td; int sn, ee;
            if (e < 225) { Hs = H01; Td = T2; sn = 15; ee = e; } else if (e < 450) { Hs = H03; Td = T0; sn = 15; ee = e - 225; } else { Hs = H0N; Td = TN; sn = N; ee = e - 450; }
            int a = ee / 15, b = ee - a * 15;
            double s = 0;
            for (int q = 0; q < 15; q++) s += Hs[q * sn + a] * sAinv[q * 15 + b];
            Td[a * 15 + b] = s;
        }
        __syncthreads();
        // rhs_blk -= T_blk rhs0 ;  H[blk][j >= blk] -= T_blk H0j   (blocks Pose2, N, Pose0 in that order)
        if (t < 15) { double s = 0; for (int q = 0; q < 15; q++) s += T2[t * 15 + q] * r0b[q]; r1b[t] -= s; }
        else if (t >= 32 && t < 47) { int a = t - 32; double s = 0; for (int q = 0; q < 15; q++) s += T0[a * 15 + q] * r0b[q]; r3b[a] -= s; }
        else if (t >= 64 && t - 64 < N) { int a = t - 64; double s = 0; for (int q = 0; q < 15; q++) s += TN[a * 15 + q] * r0b[q]; rNb[a] -= s; }
        {
            const int nA = 225, nB = 15 * N, nC = 225, nD = N * N, nE = 15 * N, nF = 225;     // (2,2) (2,N) (2,0) (N,N) (N,0) (0,0)
            for (int e = t; e < nA + nB + nC + nD + nE + nF; e += NT) {
                const double* Tm; const double* Hs; double* Hd_; int sv, ee;
                if (e < nA) { Tm = T2; Hs = H01; Hd_ = H11; sv = 15; ee = e; }
                else if (e < nA + nB) { Tm = T2; Hs = H0N; Hd_ = H1N; sv = N; ee = e - nA; }
                else if (e < nA + nB + nC) { Tm = T2; Hs = H03; Hd_ = H13; sv = 15; ee = e - nA - nB; }
                else if (e < nA + nB + nC + nD) { Tm = TN; Hs = H0N; Hd_ = HNN; sv = N; ee = e - nA - nB - nC; }
                else if (e < nA + nB + nC + nD + nE) { Tm = TN; Hs = H03; Hd_ = HN3; sv = 15; ee = e - nA - nB - nC - nD; }
                else { Tm = T0; Hs = H03; Hd_ = H33; sv = 15; ee = e - nA - nB - nC - nD - nE; }
                int a = ee / sv, b = ee - a * sv;
                double s = 0;
                for (int q = 0; q < 15; q++) s += Tm[a * 15 + q] * Hs[q * sv + b];
                Hd_[a * sv + b] -= s;
            }
        }
        __syncthreads();
        if (k == 1) CHSTAMP(43);
        // MoveHessianData: save what UpdateHiddenState needs, shift Pose2 -> Pose1
        {
            double* s_inv = A.hmn_inv + (size_t)(e0 + i) * 225; double* s_2 = A.hmn_2 + (size_t)(e0 + i) * 225; double* s_0 = A.hmn_0 + (size_t)(e0 + i) * 225;
            double* s_N = A.hmn_N + pn0 + (size_t)i * 15 * N;
            for (int e = t; e < 225; e += NT) { s_inv[e] = sAinv[e]; s_2[e] = H01[e]; s_0[e] = H03[e]; }
            for (int e = t; e < 15 * N; e += NT) s_N[e] = H0N[e];
            if (t < 15) A.rhsmn[(size_t)(e0 + i) * 15 + t] = r0b[t];
        }
        // (same element -> same thread in the save above and the shift below: no barrier between them)
        for (int e = t; e < 225; e += NT) { H00[e] = H11[e]; H11[e] = 0; H03[e] = H13[e]; H13[e] = 0; H01[e] = 0; }
        for (int e = t; e < 15 * N; e += NT) { H0N[e] = H1N[e]; H1N[e] = 0; }
        if (t < 15) { r0b[t] = r1b[t]; r1b[t] = 0; }
        __syncthreads();
    }
    CHSTAMP(44);
    // UpdateSchurComponent: dense remainder in the order [Pose0 | Pose1 (= frame j) | N], written to HBM first (it is an output, and
    // the LDS pool it is gathered from is about to be reused for the factorisation)
    for (int e = t; e < G * G; e += NT) {
        int a = e / G, b = e - a * G;
        int lo = a < b ? a : b, hi = a < b ? b : a;     // symmetric: take the upper entry (selfadjointView<Upper>)
        int bl = lo < 15 ? 0 : lo < 30 ? 1 : 2, bh = hi < 15 ? 0 : hi < 30 ? 1 : 2;
        int il = lo - (bl == 0 ? 0 : bl == 1 ? 15 : 30), ih = hi - (bh == 0 ? 0 : bh == 1 ? 15 : 30);
        double v;
        if (bl == 0 && bh == 0) v = H33[il * 15 + ih];
        else if (bl == 0 && bh == 1) v = H03[ih * 15 + il];          // (Pose0, Pose1) = H[Pose1, Pose0]^T
        else if (bl == 0 && bh == 2) v = HN3[ih * 15 + il];          // (Pose0, N) = H[N, Pose0]^T
        else if (bl == 1 && bh == 1) v = H00[il * 15 + ih];
        else if (bl == 1 && bh == 2) v = H0N[il * N + ih];
        else v = HNN[il * N + ih];
        A.Hd[g20 + e] = v;
    }
    if (t < G) { double v = t < 15 ? r3b[t] : t < 30 ? r0b[t - 15] : rNb[t - 30]; A.rd[g0 + t] = v; }
    __syncthreads();
    // Square root of the remainder.  The remainder of ONE factor is only positive SEMI-definite in general — between two visual
    // frames nothing in the chain pins, say, the heading — which is why the reference takes an eigen square root here
    // (UpdateSchurComponent, R/factor/gnss_imu_factor.cpp:454-488, eigenvalues <= 1e-8 dropped).  A Gauss-Newton solver consumes
    // J^T J, J^T r and |r|^2 only, so any J with J^T J = H does: here a diagonally pivoted outer-product Cholesky, which is
    // rank-revealing and needs no eigen-solve.  Step r takes the largest remaining diagonal entry H_pp as pivot:
    //   v_r = H[:, p] / sqrt(H_pp)   (row r of J),   rho_r = rhs_p / sqrt(H_pp)   (entry r of the residual),
    //   H -= v_r v_r^T,  rhs -= v_r rho_r,
    // and stops when the remaining diagonal is below 1e-14 of the first pivot or 1e-8 absolute (the reference's eigenvalue
    // threshold): the rows beyond the rank are zero.  sum_r v_r v_r^T = H and sum_r v_r rho_r = rhs on the retained range.
    {
        // Round 5: LEFT-LOOKING in ONE wavefront, no barrier inside (lane i owns row i — and row i + 64 in the instantiation for up to 64
        // ambiguities, G + 1 <= 95 rows — for good; row G is the right-hand side, which rides along: its factor row is L^-1 rhs).  Nothing
        // is updated but the running diagonal (a register per row); step r forms only the pivot's column,
        // c_i = H[i][p] - sum_{s < r} L[i][s] L[p][s], and stores it as column r.  To keep column s of L AT position s (every address of the
        // inner product is `row base + s`: contiguous reads along the lane's own rows — rows are padded to an odd length: conflict-free —
        // and LDS broadcasts of the pivot's row) each lane first exchanges, in its own rows, position r with the position that holds H's
        // column p: a column of H is read once, when its index becomes the pivot; the owner of row q tracks where column q lives (cpos).
        // Rows never move, so the row index is the original index, the arg-max is a ballot, and nothing crosses lanes but the broadcasts.
        // The factor goes to HBM after the loop, coalesced, by all threads.  G^3 / 6 multiply-adds on one wave instead of G full-matrix
        // updates behind three barriers each, an arg-max through six ds_bpermute round trips and an IEEE division + square root per step
        // (rounds 1-4: ~180 k of the small instantiation's 380 k cycles; the large one kept that form until the end of round 5: 234 us per
        // launch at 40 ambiguities).  What is left is the instructions one wave issues per step (~8 cycles each): 1.4 k cycles a step.
        constexpr int RPL = (30 + NMAX + 1 + 63) / 64;      // rows per lane: 1 (G + 1 <= 55) or 2 (G + 1 <= 95)
        __shared__ int sRank;
        const int LDP = G | 1;
        for (int e = t; e < G * G; e += NT) { int a = e / G, b = e - a * G; sD[a * LDP + b] = A.Hd[g20 + e]; }
        if (t < G) sD[G * LDP + t] = A.rd[g0 + t];
        __syncthreads();
        CHSTAMP(45);
        if (t < 64) {
            const double GONE = -1e300;
            int idx[RPL], cpos[RPL];
            double* rowi[RPL]; double di[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++) {
                idx[k] = t + 64 * k;                        // (rows beyond the right-hand side shadow it; they store nothing)
                rowi[k] = sD + (idx[k] <= G ? idx[k] : G) * LDP;
                di[k] = idx[k] < G ? rowi[k][idx[k]] : GONE;
                cpos[k] = idx[k];                           // where column idx of H lives in every row
            }
            double d0 = 0.0;
            int r = 0;
            for (; r < G; r++) {
                double dm = di[0];
#pragma unroll
                for (int k = 1; k < RPL; k++) dm = fmax(dm, di[k]);
                const double m16 = grp16_max(dm);
                const double bv = fmax(fmax(rows_lane(m16, 0), rows_lane(m16, 16)), fmax(rows_lane(m16, 32), rows_lane(m16, 48)));
                int p = 0;                                  // (the first index wins ties)
                {
                    bool found = false;
#pragma unroll
                    for (int k = 0; k < RPL; k++) {
                        const unsigned long long hit = __builtin_amdgcn_ballot_w64(di[k] == bv);
                        if (!found && hit) { p = 64 * k + (int)__builtin_ctzll(hit); found = true; }
                    }
                }
                if (r == 0) d0 = bv;
                if (!(bv > 1e-14 * d0) || !(bv > 1e-8)) break;                 // uniform
                int P = __builtin_amdgcn_readlane(cpos[0], p & 63);            // column p of H sits at position P >= r
#pragma unroll
                for (int k = 1; k < RPL; k++) { const int Pk = __builtin_amdgcn_readlane(cpos[k], p & 63); if ((p >> 6) == k) P = Pk; }
                double c[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    c[k] = rowi[k][P];
                    const double hr = rowi[k][r];
                    asm volatile("" ::: "memory");
                    if (idx[k] <= G && P != r) rowi[k][P] = hr;                // H's column from position r moves to P (its owner notes it)
                    if (cpos[k] == r) cpos[k] = P;
                }
                const double* rp = sD + p * LDP;
#pragma unroll 4
                for (int s2 = 0; s2 < r; s2++) {
                    const double b = rp[s2];
#pragma unroll
                    for (int k = 0; k < RPL; k++) c[k] -= rowi[k][s2] * b;
                }
                const double isq = rsqrt_nr(bv);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const bool live = idx[k] == G || (idx[k] < G && di[k] > 0.5 * GONE);
                    const double v = live ? c[k] * isq : 0.0;                  // (a row that was a pivot has nothing right of its own column)
                    if (idx[k] <= G) rowi[k][r] = v;
                    di[k] = (idx[k] == p) ? GONE : (live && idx[k] < G ? di[k] - v * v : di[k]);
                }
                asm volatile("" ::: "memory");
            }
            if (t == 0) sRank = r;
        }
        CHSTAMP(46);
        __syncthreads();
        // the factor, the rows of the square root and the whitened right-hand side; zero beyond the rank
        const int rank = sRank;
        for (int e = t; e < G * G; e += NT) {
            const int a = e / G, r2 = e - a * G;
            const double v = r2 < rank ? sD[a * LDP + r2] : 0.0;
            A.Ld[g20 + e] = v;
            if (A.jac_out) A.jac_out[g20 + (size_t)r2 * G + a] = v;
        }
        if (t < G) { const double v = t < rank ? sD[G * LDP + t] : 0.0; A.r0[g0 + t] = v; A.res_out[g0 + t] = v; }
    }
    if (t < 32) A.old[(size_t)f * 32 + t] = sOut[t];
    if (t < N) A.N_old[n0 + t] = sNv[t];
    if (t == 0) { A.history[f] = 1; A.status[f] = sBad ? -1 : 0; }
    CHSTAMP(47);
}
template <int NMAX, int NT>
__global__ void __launch_bounds__(NT, (NT == 256 && NMAX <= CO_TINYN) ? 6 : (NT == 256 && NMAX <= CO_SMALLN) ? 4 : 1) k_comp_elim(CompArgs A, int n_lo) { d_comp_elim<NMAX, NT>(A, (int)blockIdx.x, n_lo); }

// Optional phase 4: the reference's square root itself (UpdateSchurComponent, R/factor/gnss_imu_factor.cpp:454-488):
//   H = V diag(lam) V^T,  J = sqrt(lam+) V^T,  r = lam+^-1/2 V^T rhs,  eigenvalues <= 1e-8 dropped, rows in ascending eigenvalue order
// — for callers that want the residual VECTOR of the reference (up to the sign of each eigenvector), not only J^T J, J^T r and |r|^2.
// The pivoted factor of phase 3 is a square root (rows v_r, sum_r v_r v_r^T = H on the retained range): a one-sided (Hestenes) Jacobi
// orthogonalises the v_r — 8 lanes per pair, round-robin pairing — never forms H again and keeps small eigenvalues to high relative
// accuracy (see the kernel for what is rotated, and why).  One 256-thread workgroup per factor; off by default (swf_composite_set_root,
// swf_options::composite_root): it costs some ten sweeps of G - 1 barrier steps.
template <int NMAX>
__global__ void __launch_bounds__(NMAX <= CO_SMALLN ? 256 : 512) k_comp_eigroot(CompArgs A) {
    constexpr int NT = NMAX <= CO_SMALLN ? 256 : 512;          // 8 lanes per pair: 32 / 64 pairs per step (G <= 54 / 94 vectors)
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= A.n || !A.todo[f]) return;
    const int N = A.N[f], G = 30 + N;
    const long long g0 = A.g_off[f], g20 = A.g2_off[f];
    constexpr int GM_ = 30 + NMAX;
    // Round 6: the Jacobi rotates the ROWS v_r of the pivoted factor (H = sum_r v_r v_r^T = L L^T, L = [v_0 v_1 ...]) from the right,
    // L W = U Sigma, instead of its columns with an accumulated V: the implicit Gram matrix L^T L of a diagonally pivoted factor is
    // close to diagonal (Veselic / Hari, the preconditioning of the marginalisation consumer's k_marg_bj), where the columns' Gram
    // matrix was H itself — 40 sweeps did not always converge on remainders with eigenvalues from 1e-4 to 3e11, ~10 do now —, one
    // matrix is rotated instead of two, and nothing is divided by a small singular value: the final vectors ARE the rows sqrt(lam_k) u_k^T
    // of the square root, lam_k their squared norms, r_k = (vector_k . rhs) / lam_k.
    __shared__ double Lm[GM_ * GM_];                      // vector r at [r * G, r * G + G)
    __shared__ double lam[GM_], rho[GM_], srd[GM_];
    __shared__ int srank[GM_];
    for (int e = t; e < G * G; e += NT) { const int r = e / G, a = e - r * G; Lm[e] = A.Ld[g20 + (size_t)a * G + r]; }      // Ld[a * G + r] = component a of row r
    if (t < G) srd[t] = A.rd[g0 + t];
    __syncthreads();
    const int Ge = (G + 1) & ~1, np = Ge / 2;            // round-robin over an even number of players (a bye when G is odd)
    const int pi = t >> 3, ln = t & 7;
#ifndef SWF_EIG_MAXSWEEP
#define SWF_EIG_MAXSWEEP 40
#endif
    for (int sweep = 0; sweep < SWF_EIG_MAXSWEEP; sweep++) {
        int rotated = 0;
        for (int step = 0; step < Ge - 1; step++) {
            // circle method: player Ge-1 stays, the others rotate; pair pi plays (a, b)
            int p = -1, q = -1;
            if (pi < np) {
                int a = pi == 0 ? Ge - 1 : (step + pi) % (Ge - 1), b = (step + Ge - 1 - pi) % (Ge - 1);
                p = a < b ? a : b; q = a < b ? b : a;
                if (q >= G) p = -1;                      // the bye
            }
            double al = 0, be = 0, ga = 0;
            if (p >= 0) for (int r = ln; r < G; r += 8) { double x = Lm[p * G + r], y = Lm[q * G + r]; al += x * x; be += y * y; ga += x * y; }
            for (int o = 4; o > 0; o >>= 1) { al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64); }
            // (rotation threshold G eps: below it the cosine of a pair is rounding noise of the G-term inner products)
            if (p >= 0 && fabs(ga) > (double)G * 1.2e-16 * sqrt(al * be) && fabs(ga) > 1e-300) {
                double zeta = (be - al) / (2.0 * ga);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + tt * tt), sn = c * tt;
                for (int r = ln; r < G; r += 8) { double x = Lm[p * G + r], y = Lm[q * G + r]; Lm[p * G + r] = c * x - sn * y; Lm[q * G + r] = sn * x + c * y; }
                rotated = 1;
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }
    // eigenvalues = squared norms of the final vectors; rows of the output in ascending eigenvalue order (Eigen's SelfAdjointEigenSolver order)
    if (t < G) {
        double s2 = 0, d = 0;
        for (int r = 0; r < G; r++) { double x = Lm[t * G + r]; s2 += x * x; d += x * srd[r]; }
        lam[t] = s2; rho[t] = d;
    }
    __syncthreads();
    if (t < G) { int rk = 0; for (int j = 0; j < G; j++) if (lam[j] < lam[t] || (lam[j] == lam[t] && j < t)) rk++; srank[t] = rk; }
    __syncthreads();
    for (int e = t; e < G * G; e += NT) {
        int k = e / G, a = e - k * G, row = srank[k];
        double v = lam[k] > 1e-8 ? Lm[k * G + a] : 0.0;
        A.Ld[g20 + (size_t)a * G + row] = v;
        if (A.jac_out) A.jac_out[g20 + (size_t)row * G + a] = v;
    }
    if (t < G) { double v = lam[t] > 1e-8 ? rho[t] / lam[t] : 0.0; A.r0[g0 + srank[t]] = v; A.res_out[g0 + srank[t]] = v; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Inside the solver a composite factor IS a linearised prior that is rewritten at every linearisation: between two Jacobian
// evaluations the reference answers from r_lin - J INC, with INC = old (-) new in exactly the coordinates of
// MarginalizationFactor::Evaluate (p - p0, +-2 vec(q0^-1 q), x - x0).  So the engine carries it as a prior-type factor with
// a static clique; at every linearisation of its window  k_comp_gather  collects the outer blocks,  k_comp_prep / k_comp_imu / k_comp_elim  move the
// hidden epochs and re-eliminate, and  k_comp_scatter  rewrites the prior's (J, r0, x0) and its clique's J^T J / diagonal.
// Evaluation, J v products, candidate cost and assembly are the prior's own code, unchanged.
// ---------------------------------------------------------------------------------------------------------------------
struct CompMeta {
    const int* win;                 // [n] window of the factor
    const int* xo_off;              // [n+1] prefix sums of 4 + N
    const int* xo;                  // ambient x offsets of the outer blocks: pose_i, sb_i, pose_j, sb_j, N scalars
    const long long* Joff; const int* roff; const int* x0off;      // the factor's prior record
    const long long* Coff; const int* voff;                        // its static clique: C (G x G), dgraw
    double* prior_J; double* prior_Jt; double* prior_r0; double* prior_x0;
    int* active;
    double* outer; double* Nv;      // = CompArgs.outer / .Nv
};

__device__ __forceinline__ void d_comp_gather(const DevBatch& B, const CompArgs& A, const CompMeta& Mt, const int f) {
    const int t = threadIdx.x;
    if (f >= A.n) return;
    const WinState& s = B.ws[Mt.win[f]];
    int act = (s.status == SWF_RUNNING && s.need_lin) ? 1 : 0;
    if (t == 0) Mt.active[f] = act;
    if (!act) return;
    const int* xo = Mt.xo + Mt.xo_off[f];
    int N = A.N[f];
    if (t < 32) { int sl = t < 7 ? 0 : t < 16 ? 1 : t < 23 ? 2 : 3, o = t < 7 ? t : t < 16 ? t - 7 : t < 23 ? t - 16 : t - 23; Mt.outer[(size_t)f * 32 + t] = B.x[xo[sl] + o]; }
    if (t >= 32 && t - 32 < N) Mt.Nv[A.n_off[f] + t - 32] = B.x[xo[4 + t - 32]];
}
__global__ void __launch_bounds__(128) k_comp_gather(DevBatch B, CompArgs A, CompMeta Mt) { d_comp_gather(B, A, Mt, (int)blockIdx.x); }

template <int NT>
__device__ __forceinline__ void d_comp_scatter(const DevBatch& B, const CompArgs& A, const CompMeta& Mt, const int f) {
    const int t = threadIdx.x;
    if (f >= A.n || !Mt.active[f]) return;
    const int N = A.N[f], G = 30 + N;
    const double* J = A.jac_out + A.g2_off[f]; const double* H = A.Hd + A.g2_off[f]; const double* r = A.res_out + A.g_off[f];
    double* pJ = Mt.prior_J + Mt.Joff[f];
    double* pJt = Mt.prior_Jt + Mt.Joff[f];
    // (a factor inside the clique of a group-0 block — Coff < 0 — has no static clique to fill: the prior evaluation copies its rows
    // into that clique's Jacobian and the clique elimination forms J^T J with the other factors' rows)
    const bool own = Mt.Coff[f] >= 0;
    double* C = B.C + (own ? Mt.Coff[f] : 0);
    for (int e = t; e < G * G; e += NT) { double v = J[e]; pJ[e] = v; pJt[(size_t)(e % G) * G + e / G] = v; if (own) C[e] = H[e]; }
    for (int e = t; e < G; e += NT) { Mt.prior_r0[Mt.roff[f] + e] = r[e]; if (own) B.cv_dgraw[Mt.voff[f] + e] = H[(size_t)e * G + e]; }
    double* x0 = Mt.prior_x0 + Mt.x0off[f];
    if (t < 32) x0[t] = A.outer[(size_t)f * 32 + t];
    if (t >= 32 && t - 32 < N) x0[t] = A.Nv[A.n_off[f] + t - 32];
    if (t == 0 && A.status[f] != 0) B.ws[Mt.win[f]].lin_fail = 1;       // a hidden epoch or the remainder was not positive definite
}
__global__ void __launch_bounds__(256) k_comp_scatter(DevBatch B, CompArgs A, CompMeta Mt) { d_comp_scatter<256>(B, A, Mt, (int)blockIdx.x); }

// (Round 5 measured ONE 1024-thread workgroup per factor through all five phases — gather, hidden-epoch move, the chain's IMU factors,
// re-elimination, prior record — against the five launches: 298.5 against 296.0 us per iteration of a cfg3-size reference-topology
// window, a batch of 16 windows 3.83 against 3.48 ms.  Each phase's exposed memory round trips are the same inside one kernel, the small
// phases lose the chip-wide parallelism they have as grids of their own, and dependent launches on one stream follow each other without
// a gap.  Not kept.  Also measured and not kept: the J^T J / J^T r products of all links formed up front, off the
// elimination chain (k_comp_elim 76.6 -> 80.1 us).)

// =========================================================================================
// Latency path of a window in the reference's topology (few windows, every workgroup resident at once): launches that do not depend on
// each other ride in ONE grid, as k_lm_clique does for the landmark product and the cliques.  The composite chain (gather -> hidden-state
// move -> IMU factors -> re-elimination -> prior records) and the visual branch (projection Jacobians -> landmark Schur product) meet
// only at the cliques, so
//   k_eval_ps_comp_imu = { projection + scalar-factor segments of k_eval_ps | k_comp_imu }        (both: 256 threads)
//   k_lm_comp          = { k_lm_schur's workgroups | k_comp_elim<CO_SMALLN, 1024> + k_comp_scatter }  (both: 1024 threads)
//   k_clique_tall2     = { k_clique_tall | the four-wave form of the 64 x 64 class }              (both: 256 threads)
// with the prior segment of k_eval_ps (the composite factors' records among them) as a launch of its own behind k_lm_comp.
// Same device functions, same operands, same order: bit-identical to the separate launches (tested: a window alone against the
// same window with SWF_NO_COMP_FUSE=1).  One cfg3-size window: 293 -> 271 us per iteration (the chain gather .. cliques 167 -> 141 us).
// =========================================================================================
__global__ void __launch_bounds__(256) k_eval_ps_comp_imu(DevBatch B, Segs S, CompArgs A) {
    __shared__ double sm[FS_BLK * FS_HALF + 168 / 2 + 1];          // the frame sums' staging tile + its frame offsets (as in k_eval_ps)
    const int bid = blockIdx.x;
    if (bid < S.e[0]) d_eval_proj_fs(B, bid, (double (*)[FS_HALF])sm, (int*)(sm + FS_BLK * FS_HALF));
    else if (bid < S.e[1]) d_eval_scalar<true>(B, bid - S.e[0]);
    else d_comp_imu(A, (bid - S.e[1]) * 8, A.n_iq);
}
template <int NCW, int TPW, int TW, int LDR>
__global__ void __launch_bounds__(LS_NT(NCW, TW)) k_lm_comp(DevBatch B, DevOpt O, CompArgs A, CompMeta Mt, int qpb, int lp, int kms, int s_direct, int n_parts) {
    static_assert(LS_NT(NCW, TW) == 1024, "k_lm_comp: d_comp_elim<., 1024> synchronises 1024 threads");
    if ((int)blockIdx.y < n_parts) d_lm_schur<NCW, TPW, TW, LDR, true>(B, O, qpb, lp, kms, s_direct, (int)blockIdx.x, (int)blockIdx.y);
    else {
        // the factor's workgroup rewrites its prior record itself (k_comp_scatter's work: what it reads, this workgroup has just written)
        const int f = ((int)blockIdx.y - n_parts) * (int)gridDim.x + (int)blockIdx.x;
        d_comp_elim<CO_SMALLN, 1024>(A, f);
        __syncthreads();
        d_comp_scatter<1024>(B, A, Mt, f);
        // (measured and not kept: the factor's prior evaluation — d_eval_prior — behind the scatter in this workgroup instead of the prior
        // segment of k_eval_ps as a launch of its own: the workgroup runs 11.5 us longer, the launch it saves took 11.3)
    }
}
// the outer blocks collected by the workgroup that moves the factor's hidden epochs (k_comp_gather + k_comp_prep of the solver path: one launch)
__global__ void __launch_bounds__(256) k_comp_gather_prep(DevBatch B, CompArgs A, CompMeta Mt) {
    d_comp_gather(B, A, Mt, (int)blockIdx.x);
    __syncthreads();
    (void)d_comp_prep(A, (int)blockIdx.x);
}
__global__ void __launch_bounds__(256) k_clique_tall2(DevBatch B, DevOpt O) {
    if ((int)blockIdx.x < B.n_clc[4]) d_clique_elim<CLQ_TALLR, 64, 9, 4, 8, 4>(B, O, (int)blockIdx.x);
    else d_clique_elim<64, 64, 9, 2, 8, 4>(B, O, (int)blockIdx.x - B.n_clc[4]);
}
