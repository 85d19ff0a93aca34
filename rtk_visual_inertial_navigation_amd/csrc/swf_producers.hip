// swf_producers.hip — the input producers of the hot path as batched kernels:
//   (1) IMU pre-integration (§8a row a6), the producer of the IMU factor's record;
//   (2) two-view landmark triangulation, the producer of the landmark blocks' initial values (end of this file).
//
// ---- (1) batched IMU pre-integration
//
// Restates IntegrationBase::{ctor, push_back, propagate, midPointIntegration, get_sqrtinfo}
// (R/factor/integration_base.cpp:5-142) for many keyframe intervals at once: one wavefront per interval, the 15x15
// bias-Jacobian and covariance recursions in LDS, lanes over matrix elements.  gfx950 only, no CPU path.
//
// What the reference computes per IMU sample (mid-point rule):
//   delta_p, delta_q, delta_v                       :30-47
//   F (15x15), V (15x18)                            :48-94
//   jacobian = F jacobian ; covariance = F cov F^T + V Q V^T   :96-97, Q = diag(ACC_N^2, GYR_N^2, ACC_N^2, GYR_N^2, ACC_W^2, GYR_W^2) :14-19
// and at the end sqrt_info = LLT(covariance^-1).matrixL()^T    :105-113.
//
// Structure used here (exact, not an approximation): rows 9..14 of F are unit rows and rows 9..14 of V touch only the
// bias-walk noise, so only the first 9 rows of F X are products, the trailing rows / columns of the new covariance are
// copies of (F cov), and V Q V^T is a 9x9 block plus a diagonal.  sqrt_info is formed as the inverse of the
// upper-triangular R with cov = R R^T (the "reverse" Cholesky factor): cov^-1 = R^-T R^-1, so R^-1 is the transposed
// lower Cholesky factor of cov^-1 — the same matrix the reference forms through an explicit inverse, at eps*sqrt(cond)
// instead of eps*cond.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/swf_solver.h"
#include "swf_dev.h"

void swf_internal_set_error(const std::string& m);
static int pi_fail(int code, const std::string& m) { swf_internal_set_error(m); return code; }
#define PI_HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return pi_fail(SWF_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {

struct PreintArgs {
    const double* samples;   // [sum n][7] dt, acc(3), gyr(3); the first sample of an interval seeds acc_0 / gyr_0
    const int* first;        // [n_int + 1] sample offsets
    const double* bias;      // [n_int][6] linearisation biases ba, bg
    double* pre;             // [n_int][SWF_PRE_DOUBLES]
    double acc_n2, gyr_n2, acc_w2, gyr_w2;
    int n_int;
};

// broadcast of lane N of each 16-lane row to the whole row (DPP row_newbcast)
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + N, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// the matrices F is made of for one push_back (integration_base.cpp:48-75), identical in the 16 lanes of an interval
struct StepMats { double Ms[9], Rs[9], M1[9], ImRw[9], dt; };

// x <- F x for one 15-vector held in registers.  F = [I F01 I*dt F03 F04; 0 ImRw 0 0 -I*dt; 0 F21 I F23 F24; 0 0 0 I 0; 0 0 0 0 I] with
// F01 = -dt^2/4 (M0 + M2), F03 = -dt^2/4 (R0 + R1), F04 = dt^3/4 M1, F21 = -dt/2 (M0 + M2), F23 = -dt/2 (R0 + R1), F24 = dt^2/2 M1
__device__ __forceinline__ void apply_F(const StepMats& S, double* x) {
    double pa[3], pb[3], pc[3], pw[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        pa[a] = S.Ms[a * 3] * x[3] + S.Ms[a * 3 + 1] * x[4] + S.Ms[a * 3 + 2] * x[5];
        pb[a] = S.Rs[a * 3] * x[9] + S.Rs[a * 3 + 1] * x[10] + S.Rs[a * 3 + 2] * x[11];
        pc[a] = S.M1[a * 3] * x[12] + S.M1[a * 3 + 1] * x[13] + S.M1[a * 3 + 2] * x[14];
        pw[a] = S.ImRw[a * 3] * x[3] + S.ImRw[a * 3 + 1] * x[4] + S.ImRw[a * 3 + 2] * x[5];
    }
    const double dt = S.dt, k1 = -0.25 * dt * dt, k3 = 0.25 * dt * dt * dt, h1 = -0.5 * dt, h3 = 0.5 * dt * dt;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double x0 = x[a] + dt * x[6 + a] + k1 * (pa[a] + pb[a]) + k3 * pc[a];
        double x6 = x[6 + a] + h1 * (pa[a] + pb[a]) + h3 * pc[a];
        double x3 = pw[a] - dt * x[12 + a];
        x[a] = x0; x[3 + a] = x3; x[6 + a] = x6;
    }
}

#define PI_GROUPS 4            // intervals per wavefront (16 lanes each: lane c < 15 owns column c of the Jacobian and of the covariance)
#define PI_TLD 17              // padded row of the transpose buffer

__global__ void __launch_bounds__(64) k_preintegrate(PreintArgs A) {
    __shared__ double sT[PI_GROUPS][15 * PI_TLD];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int it = blockIdx.x * PI_GROUPS + g;
    const bool live = it < A.n_int;
    const int itc = live ? it : A.n_int - 1;
    const int s0 = A.first[itc], n = live ? A.first[itc + 1] - s0 : 0;
    const double* __restrict__ smp = A.samples + (size_t)s0 * 7;
    const double* bb = A.bias + (size_t)itc * 6;
    const double ba[3] = { bb[0], bb[1], bb[2] }, bg[3] = { bb[3], bb[4], bb[5] };
    int nmax = n;
    nmax = max(nmax, __shfl_xor(nmax, 16)); nmax = max(nmax, __shfl_xor(nmax, 32));
    double jc[15], cc[15];                                  // column c of jacobian / covariance
#pragma unroll
    for (int i = 0; i < 15; i++) { jc[i] = (i == c) ? 1.0 : 0.0; cc[i] = 0.0; }
    double dp[3] = { 0, 0, 0 }, dq[4] = { 0, 0, 0, 1 }, dv[3] = { 0, 0, 0 }, sum_dt = 0;
    double acc0[3] = { 0, 0, 0 }, gyr0[3] = { 0, 0, 0 };
    if (n > 0) { acc0[0] = smp[1]; acc0[1] = smp[2]; acc0[2] = smp[3]; gyr0[0] = smp[4]; gyr0[1] = smp[5]; gyr0[2] = smp[6]; }
    const double gyri[3] = { gyr0[0], gyr0[1], gyr0[2] };
    double* T = sT[g];
    const int bj = c / 3, b = c - bj * 3;                   // block column / column inside the block of this lane's covariance column
    double nx[7];
#pragma unroll
    for (int k = 0; k < 7; k++) nx[k] = (n > 1) ? smp[7 + k] : 0.0;
    for (int s = 1; s < nmax; s++) {
        const bool act = s < n;
        double cur[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cur[k] = nx[k];
        if (s + 1 < n) {
#pragma unroll
            for (int k = 0; k < 7; k++) nx[k] = smp[(s + 1) * 7 + k];       // prefetch the next sample under this step's arithmetic
        }
        if (act) {
            const double dt = cur[0];
            const double* acc1 = cur + 1; const double* gyr1 = cur + 4;
            double a0[3], a1[3], w[3], un0[3], un1[3], rq[4], hq[4], rp[3], rv[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { a0[k] = acc0[k] - ba[k]; a1[k] = acc1[k] - ba[k]; w[k] = 0.5 * (gyr0[k] + gyr1[k]) - bg[k]; }
            qrot(dq, a0, un0);
            hq[0] = w[0] * dt / 2; hq[1] = w[1] * dt / 2; hq[2] = w[2] * dt / 2; hq[3] = 1;
            qmul(dq, hq, rq);
            qrot(rq, a1, un1);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double un = 0.5 * (un0[k] + un1[k]);
                rp[k] = dp[k] + dv[k] * dt + 0.5 * un * dt * dt;
                rv[k] = dv[k] + un * dt;
            }
            StepMats S;
            double R0[9], R1[9], Rw[9], Ra0[9], Ra1[9], M0[9], M2[9];
            q2R(dq, R0); q2R(rq, R1);
            skew3(w, Rw); skew3(a0, Ra0); skew3(a1, Ra1);
#pragma unroll
            for (int i = 0; i < 9; i++) S.ImRw[i] = -Rw[i] * dt;
            S.ImRw[0] += 1; S.ImRw[4] += 1; S.ImRw[8] += 1;
            mat3mul(R0, Ra0, M0);             // R0 [a0]x
            mat3mul(R1, Ra1, S.M1);           // R1 [a1]x
            mat3mul(S.M1, S.ImRw, M2);        // R1 [a1]x (I - [w]x dt)
#pragma unroll
            for (int i = 0; i < 9; i++) { S.Ms[i] = M0[i] + M2[i]; S.Rs[i] = R0[i] + R1[i]; }
            S.dt = dt;
            // jacobian = F jacobian ; T = F cov, both column by column in registers
            apply_F(S, jc);
            apply_F(S, cc);
            // cov symmetric => row c of (F cov) is column c of cov F^T: transpose through LDS, then F (cov F^T)
#pragma unroll
            for (int i = 0; i < 15; i++) if (c < 15) T[i * PI_TLD + c] = cc[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 15; k++) cc[k] = T[(c < 15 ? c : 0) * PI_TLD + k];
            __builtin_amdgcn_wave_barrier();
            apply_F(S, cc);
            // + V Q V^T (integration_base.cpp:76-97).  With W = [R0 | -dt/2 M1 | R1 | -dt/2 M1]: V rows 0-2 = dt^2/4 W, rows 6-8 = dt/2 W,
            // rows 3-5 = dt/2 [0 I 0 I];  G = W Q W^T = an (R0 R0^T + R1 R1^T) + gn dt^2/2 M1 M1^T,  H = W Q [0 I 0 I]^T = -gn dt M1
            double gcol[3], hcol[3], hrow[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double gv[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double rr = R0[a * 3] * R0[q * 3] + R0[a * 3 + 1] * R0[q * 3 + 1] + R0[a * 3 + 2] * R0[q * 3 + 2]
                              + R1[a * 3] * R1[q * 3] + R1[a * 3 + 1] * R1[q * 3 + 1] + R1[a * 3 + 2] * R1[q * 3 + 2];
                    double mm = S.M1[a * 3] * S.M1[q * 3] + S.M1[a * 3 + 1] * S.M1[q * 3 + 1] + S.M1[a * 3 + 2] * S.M1[q * 3 + 2];
                    gv[q] = A.acc_n2 * rr + 0.5 * A.gyr_n2 * dt * dt * mm;
                }
                gcol[a] = b == 0 ? gv[0] : b == 1 ? gv[1] : gv[2];
                hcol[a] = -A.gyr_n2 * dt * (b == 0 ? S.M1[a * 3] : b == 1 ? S.M1[a * 3 + 1] : S.M1[a * 3 + 2]);
                hrow[a] = -A.gyr_n2 * dt * (b == 0 ? S.M1[a] : b == 1 ? S.M1[3 + a] : S.M1[6 + a]);
            }
            const double d2 = dt * dt, d3 = d2 * dt;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double eb = (a == b) ? 0.5 * A.gyr_n2 * d2 : 0.0;
                double n0 = bj == 0 ? 0.0625 * d2 * d2 * gcol[a] : bj == 1 ? 0.125 * d3 * hcol[a] : 0.125 * d3 * gcol[a];
                double n3 = bj == 0 ? 0.125 * d3 * hrow[a] : bj == 1 ? eb : 0.25 * d2 * hrow[a];
                double n6 = bj == 0 ? 0.125 * d3 * gcol[a] : bj == 1 ? 0.25 * d2 * hcol[a] : 0.25 * d2 * gcol[a];
                if (c < 9) { cc[a] += n0; cc[3 + a] += n3; cc[6 + a] += n6; }
            }
#pragma unroll
            for (int m = 0; m < 6; m++) if (c == 9 + m) cc[9 + m] += (m < 3 ? A.acc_w2 : A.gyr_w2) * d2;
            // propagate(): integration_base.cpp:131-141
            double nq = sqrt(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]);
#pragma unroll
            for (int k = 0; k < 3; k++) { dp[k] = rp[k]; dv[k] = rv[k]; acc0[k] = acc1[k]; gyr0[k] = gyr1[k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) dq[k] = rq[k] / nq;
            sum_dt += dt;
        }
    }
    double* out = A.pre + (size_t)itc * SWF_PRE_DOUBLES;
    if (live && c == 0) {
        for (int k = 0; k < 3; k++) {
            out[SWF_PRE_DP + k] = dp[k]; out[SWF_PRE_DV + k] = dv[k]; out[SWF_PRE_LBA + k] = ba[k]; out[SWF_PRE_LBG + k] = bg[k];
            out[SWF_PRE_GYRI + k] = gyri[k]; out[SWF_PRE_GYRJ + k] = gyr0[k];
        }
        for (int k = 0; k < 4; k++) out[SWF_PRE_DQ + k] = dq[k];
        out[SWF_PRE_SUMDT] = sum_dt;
    }
    if (live && c >= 9 && c < 15) {                         // the five 3x3 bias-Jacobian blocks: columns 9-11 (ba) and 12-14 (bg)
        const int cb = c >= 12 ? c - 12 : c - 9;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (c < 12) { out[SWF_PRE_DP_DBA + a * 3 + cb] = jc[a]; out[SWF_PRE_DV_DBA + a * 3 + cb] = jc[6 + a]; }
            else { out[SWF_PRE_DP_DBG + a * 3 + cb] = jc[a]; out[SWF_PRE_DQ_DBG + a * 3 + cb] = jc[3 + a]; out[SWF_PRE_DV_DBG + a * 3 + cb] = jc[6 + a]; }
        }
    }
    // get_sqrtinfo: cov = R R^T with R upper triangular, built from the last column backwards; lane k holds column k (= row k) of
    // the trailing matrix, so the pivot column comes from lane j by row broadcast and R[k][j] from the lane's own row j
    bool ok = true;
#define PI_CHOL_STEP(J) { \
        double d = row_bcast<J>(cc[J]); \
        ok = ok && (d > 0) && (d < 1e300); \
        double sd = sqrt(ok ? d : 1.0), inv = 1.0 / sd; \
        double rk = cc[J] * inv; \
        _Pragma("unroll") for (int i = 0; i < J; i++) { \
            double rij = row_bcast<J>(cc[i]) * inv; \
            if (c < J) cc[i] -= rij * rk; \
            else if (c == J) cc[i] = rij; \
        } \
        if (c == J) cc[J] = sd; }
    PI_CHOL_STEP(14) PI_CHOL_STEP(13) PI_CHOL_STEP(12) PI_CHOL_STEP(11) PI_CHOL_STEP(10) PI_CHOL_STEP(9) PI_CHOL_STEP(8) PI_CHOL_STEP(7)
    PI_CHOL_STEP(6) PI_CHOL_STEP(5) PI_CHOL_STEP(4) PI_CHOL_STEP(3) PI_CHOL_STEP(2) PI_CHOL_STEP(1) PI_CHOL_STEP(0)
#undef PI_CHOL_STEP
    // U = R^-1: lane c solves R u = e_c backwards; R[i][k] (k > i) is element i of lane k
    double u[15];
#pragma unroll
    for (int i = 0; i < 15; i++) u[i] = 0.0;
#define PI_BACK_ROW(I) { \
        double acc = 0; \
        PI_BACK_TERMS_##I \
        double rii = row_bcast<I>(cc[I]); \
        u[I] = (c == I) ? 1.0 / rii : (c > I ? -acc / rii : 0.0); }
#define PI_T(I, K) acc += row_bcast<K>(cc[I]) * u[K];
#define PI_BACK_TERMS_14
#define PI_BACK_TERMS_13 PI_T(13, 14)
#define PI_BACK_TERMS_12 PI_T(12, 13) PI_T(12, 14)
#define PI_BACK_TERMS_11 PI_T(11, 12) PI_T(11, 13) PI_T(11, 14)
#define PI_BACK_TERMS_10 PI_T(10, 11) PI_T(10, 12) PI_T(10, 13) PI_T(10, 14)
#define PI_BACK_TERMS_9 PI_T(9, 10) PI_T(9, 11) PI_T(9, 12) PI_T(9, 13) PI_T(9, 14)
#define PI_BACK_TERMS_8 PI_T(8, 9) PI_T(8, 10) PI_T(8, 11) PI_T(8, 12) PI_T(8, 13) PI_T(8, 14)
#define PI_BACK_TERMS_7 PI_T(7, 8) PI_T(7, 9) PI_T(7, 10) PI_T(7, 11) PI_T(7, 12) PI_T(7, 13) PI_T(7, 14)
#define PI_BACK_TERMS_6 PI_T(6, 7) PI_T(6, 8) PI_T(6, 9) PI_T(6, 10) PI_T(6, 11) PI_T(6, 12) PI_T(6, 13) PI_T(6, 14)
#define PI_BACK_TERMS_5 PI_T(5, 6) PI_T(5, 7) PI_T(5, 8) PI_T(5, 9) PI_T(5, 10) PI_T(5, 11) PI_T(5, 12) PI_T(5, 13) PI_T(5, 14)
#define PI_BACK_TERMS_4 PI_T(4, 5) PI_T(4, 6) PI_T(4, 7) PI_T(4, 8) PI_T(4, 9) PI_T(4, 10) PI_T(4, 11) PI_T(4, 12) PI_T(4, 13) PI_T(4, 14)
#define PI_BACK_TERMS_3 PI_T(3, 4) PI_T(3, 5) PI_T(3, 6) PI_T(3, 7) PI_T(3, 8) PI_T(3, 9) PI_T(3, 10) PI_T(3, 11) PI_T(3, 12) PI_T(3, 13) PI_T(3, 14)
#define PI_BACK_TERMS_2 PI_T(2, 3) PI_T(2, 4) PI_T(2, 5) PI_T(2, 6) PI_T(2, 7) PI_T(2, 8) PI_T(2, 9) PI_T(2, 10) PI_T(2, 11) PI_T(2, 12) PI_T(2, 13) PI_T(2, 14)
#define PI_BACK_TERMS_1 PI_T(1, 2) PI_T(1, 3) PI_T(1, 4) PI_T(1, 5) PI_T(1, 6) PI_T(1, 7) PI_T(1, 8) PI_T(1, 9) PI_T(1, 10) PI_T(1, 11) PI_T(1, 12) PI_T(1, 13) PI_T(1, 14)
#define PI_BACK_TERMS_0 PI_T(0, 1) PI_T(0, 2) PI_T(0, 3) PI_T(0, 4) PI_T(0, 5) PI_T(0, 6) PI_T(0, 7) PI_T(0, 8) PI_T(0, 9) PI_T(0, 10) PI_T(0, 11) PI_T(0, 12) PI_T(0, 13) PI_T(0, 14)
    PI_BACK_ROW(14) PI_BACK_ROW(13) PI_BACK_ROW(12) PI_BACK_ROW(11) PI_BACK_ROW(10) PI_BACK_ROW(9) PI_BACK_ROW(8) PI_BACK_ROW(7)
    PI_BACK_ROW(6) PI_BACK_ROW(5) PI_BACK_ROW(4) PI_BACK_ROW(3) PI_BACK_ROW(2) PI_BACK_ROW(1) PI_BACK_ROW(0)
#undef PI_T
#undef PI_BACK_ROW
    if (live && c < 15) {
#pragma unroll
        for (int i = 0; i < 15; i++) out[SWF_PRE_SQRTINFO + i * 15 + c] = ok ? u[i] : 0.0;
    }
}

}  // namespace

// C-ABI, include/swf_solver.h
extern "C" int swf_preintegrate_batch(const double* samples, const int32_t* first, int32_t n_intervals, const double* bias,
                                      const double noise[4], double* pre, int32_t on_device, void* stream) {
    if (!samples || !first || !bias || !noise || !pre || n_intervals < 0) return pi_fail(SWF_E_INVALID, "swf_preintegrate_batch: null argument");
    if (n_intervals == 0) return SWF_OK;
    hipStream_t st = (hipStream_t)stream;
    PreintArgs A{};
    A.acc_n2 = noise[0] * noise[0]; A.gyr_n2 = noise[1] * noise[1]; A.acc_w2 = noise[2] * noise[2]; A.gyr_w2 = noise[3] * noise[3];
    A.n_int = n_intervals;
    if (on_device) {
        A.samples = samples; A.first = first; A.bias = bias; A.pre = pre;
        hipLaunchKernelGGL(k_preintegrate, dim3((n_intervals + PI_GROUPS - 1) / PI_GROUPS), dim3(64), 0, st, A);
        PI_HIPCHK(hipGetLastError());
        return SWF_OK;
    }
    for (int i = 0; i < n_intervals; i++)
        if (first[i + 1] < first[i]) return pi_fail(SWF_E_INVALID, "swf_preintegrate_batch: sample offsets must be non-decreasing");
    size_t ns = (size_t)(first[n_intervals] - first[0]);
    double *d_s = nullptr, *d_b = nullptr, *d_o = nullptr; int* d_f = nullptr;
    std::vector<int32_t> rel(first, first + n_intervals + 1);
    for (auto& v : rel) v -= first[0];
    auto cleanup = [&]() { (void)hipFree(d_s); (void)hipFree(d_b); (void)hipFree(d_o); (void)hipFree(d_f); };
#define PI_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return pi_fail(SWF_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    PI_TRY(hipMalloc(&d_s, std::max<size_t>(1, ns) * 7 * sizeof(double)));
    PI_TRY(hipMalloc(&d_b, (size_t)n_intervals * 6 * sizeof(double)));
    PI_TRY(hipMalloc(&d_o, (size_t)n_intervals * SWF_PRE_DOUBLES * sizeof(double)));
    PI_TRY(hipMalloc(&d_f, (size_t)(n_intervals + 1) * sizeof(int)));
    PI_TRY(hipMemcpyAsync(d_s, samples + (size_t)first[0] * 7, ns * 7 * sizeof(double), hipMemcpyHostToDevice, st));
    PI_TRY(hipMemcpyAsync(d_b, bias, (size_t)n_intervals * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    PI_TRY(hipMemcpyAsync(d_f, rel.data(), rel.size() * sizeof(int), hipMemcpyHostToDevice, st));
    A.samples = d_s; A.first = d_f; A.bias = d_b; A.pre = d_o;
    hipLaunchKernelGGL(k_preintegrate, dim3((n_intervals + PI_GROUPS - 1) / PI_GROUPS), dim3(64), 0, st, A);
    PI_TRY(hipGetLastError());
    PI_TRY(hipMemcpyAsync(pre, d_o, (size_t)n_intervals * SWF_PRE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, st));
    PI_TRY(hipStreamSynchronize(st));
#undef PI_TRY
    cleanup();
    return SWF_OK;
}

// =====================================================================================================================
// (2) two-view landmark triangulation: FeatureManager::triangulate, the branch every feature with >= 2 observations takes
// (R/feature/feature_manager.cpp:285-316), with triangulatePoint (:148-161): the DLT design matrix of the first two
// observing frames, its right singular vector of the smallest singular value, depth in the first camera (INIT_DEPTH if not
// positive), and the world point  Rs[i] (ric (pt / idepth) + tic - Pbg) + Ps[i].
// One lane per feature.  The singular vector comes from a one-sided (Hestenes) Jacobi on the four columns of the 4x4
// matrix in registers (Eigen's JacobiSVD is the two-sided variant; the vector is the same up to sign, and the sign
// cancels in the division by its fourth component).
// =====================================================================================================================
namespace {
struct TriArgs {
    const double* Ps; const double* Rs;        // [n_frames][3], [n_frames][9] row-major
    const int* start; const double* pt0; const double* pt1;   // [n], [n][2], [n][2]
    double* depth; double* world;              // [n], [n][3]
    double tic[3], ric[9], pbg[3], init_depth;
    int n, n_frames;
};

__device__ __forceinline__ void tri_cam_pose(const TriArgs& A, int f, double* Rt /*R^T, 9*/, double* mt /*-R^T t, 3*/) {
    const double* P = A.Ps + (size_t)f * 3; const double* R = A.Rs + (size_t)f * 9;
    double t[3], Rc[9];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = P[i] + R[i * 3] * A.tic[0] + R[i * 3 + 1] * A.tic[1] + R[i * 3 + 2] * A.tic[2];
    mat3mul(R, A.ric, Rc);
    mat3T(Rc, Rt);
#pragma unroll
    for (int i = 0; i < 3; i++) mt[i] = -(Rt[i * 3] * t[0] + Rt[i * 3 + 1] * t[1] + Rt[i * 3 + 2] * t[2]);
}

#define TRI_ROT(p, q) { \
        double al = 0, be = 0, ga = 0; \
        _Pragma("unroll") for (int r = 0; r < 4; r++) { al += D[r][p] * D[r][p]; be += D[r][q] * D[r][q]; ga += D[r][p] * D[r][q]; } \
        if (ga != 0.0 && ga * ga > 1e-30 * (al * be)) { \
            double zeta = (be - al) / (2.0 * ga); \
            double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta)); \
            double c = 1.0 / sqrt(1.0 + t * t), sn = c * t; \
            _Pragma("unroll") for (int r = 0; r < 4; r++) { \
                double a = D[r][p], b = D[r][q]; D[r][p] = c * a - sn * b; D[r][q] = sn * a + c * b; \
                double va = V[r][p], vb = V[r][q]; V[r][p] = c * va - sn * vb; V[r][q] = sn * va + c * vb; } \
            rot = true; } }

__global__ void __launch_bounds__(256) k_triangulate(TriArgs A) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    int f0 = A.start[i];
    if (f0 < 0 || f0 + 1 >= A.n_frames) { A.depth[i] = -1.0; A.world[i * 3] = A.world[i * 3 + 1] = A.world[i * 3 + 2] = 0.0; return; }
    double R0t[9], m0[3], R1t[9], m1[3];
    tri_cam_pose(A, f0, R0t, m0);
    tri_cam_pose(A, f0 + 1, R1t, m1);
    const double u0 = A.pt0[i * 2], v0 = A.pt0[i * 2 + 1], u1 = A.pt1[i * 2], v1 = A.pt1[i * 2 + 1];
    double D[4][4], V[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        double p0r0 = c < 3 ? R0t[c] : m0[0], p0r1 = c < 3 ? R0t[3 + c] : m0[1], p0r2 = c < 3 ? R0t[6 + c] : m0[2];
        double p1r0 = c < 3 ? R1t[c] : m1[0], p1r1 = c < 3 ? R1t[3 + c] : m1[1], p1r2 = c < 3 ? R1t[6 + c] : m1[2];
        D[0][c] = u0 * p0r2 - p0r0; D[1][c] = v0 * p0r2 - p0r1; D[2][c] = u1 * p1r2 - p1r0; D[3][c] = v1 * p1r2 - p1r1;
#pragma unroll
        for (int r = 0; r < 4; r++) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rot = false;
        TRI_ROT(0, 1) TRI_ROT(0, 2) TRI_ROT(0, 3) TRI_ROT(1, 2) TRI_ROT(1, 3) TRI_ROT(2, 3)
        if (!rot) break;
    }
    double nr[4];
#pragma unroll
    for (int c = 0; c < 4; c++) nr[c] = D[0][c] * D[0][c] + D[1][c] * D[1][c] + D[2][c] * D[2][c] + D[3][c] * D[3][c];
    int k = 0;
#pragma unroll
    for (int c = 1; c < 4; c++) if (nr[c] < nr[k]) k = c;
    double x[4];
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = k == 0 ? V[r][0] : k == 1 ? V[r][1] : k == 2 ? V[r][2] : V[r][3];
    double X[3] = { x[0] / x[3], x[1] / x[3], x[2] / x[3] };
    double depth = R0t[6] * X[0] + R0t[7] * X[1] + R0t[8] * X[2] + m0[2];
    if (!(depth > 0)) depth = A.init_depth;           // the reference tests depth <= 0; a NaN (x[3] = 0) also takes the default
    const double* P = A.Ps + (size_t)f0 * 3; const double* R = A.Rs + (size_t)f0 * 9;
    double pc[3] = { u0 * depth, v0 * depth, depth }, pb[3];
#pragma unroll
    for (int r = 0; r < 3; r++) pb[r] = A.ric[r * 3] * pc[0] + A.ric[r * 3 + 1] * pc[1] + A.ric[r * 3 + 2] * pc[2] + A.tic[r] - A.pbg[r];
    A.depth[i] = depth;
#pragma unroll
    for (int r = 0; r < 3; r++) A.world[i * 3 + r] = R[r * 3] * pb[0] + R[r * 3 + 1] * pb[1] + R[r * 3 + 2] * pb[2] + P[r];
}
#undef TRI_ROT
}  // namespace

extern "C" int swf_triangulate_batch(const double* Ps, const double* Rs, int32_t n_frames, const double tic[3], const double ric[9],
                                     const double pbg[3], const int32_t* start_frame, const double* pt0, const double* pt1, int32_t n,
                                     double init_depth, double* depth, double* pts_world, int32_t on_device, void* stream) {
    if (!Ps || !Rs || !tic || !ric || !pbg || !start_frame || !pt0 || !pt1 || !depth || !pts_world || n < 0 || n_frames < 0)
        return pi_fail(SWF_E_INVALID, "swf_triangulate_batch: null argument");
    if (n == 0) return SWF_OK;
    hipStream_t st = (hipStream_t)stream;
    TriArgs A{};
    for (int k = 0; k < 3; k++) { A.tic[k] = tic[k]; A.pbg[k] = pbg[k]; }
    for (int k = 0; k < 9; k++) A.ric[k] = ric[k];
    A.init_depth = init_depth; A.n = n; A.n_frames = n_frames;
    if (on_device) {
        A.Ps = Ps; A.Rs = Rs; A.start = start_frame; A.pt0 = pt0; A.pt1 = pt1; A.depth = depth; A.world = pts_world;
        hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, st, A);
        PI_HIPCHK(hipGetLastError());
        return SWF_OK;
    }
    double* d = nullptr; int* d_s = nullptr;
    const size_t nf = (size_t)std::max(1, n_frames);
    const size_t doubles = nf * 12 + (size_t)n * 8;        // Ps | Rs | pt0 | pt1 | depth | world
    auto cleanup = [&]() { (void)hipFree(d); (void)hipFree(d_s); };
#define TR_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return pi_fail(SWF_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    TR_TRY(hipMalloc(&d, doubles * sizeof(double)));
    TR_TRY(hipMalloc(&d_s, (size_t)n * sizeof(int)));
    double* dPs = d; double* dRs = dPs + nf * 3; double* dp0 = dRs + nf * 9; double* dp1 = dp0 + (size_t)n * 2;
    double* ddep = dp1 + (size_t)n * 2; double* dw = ddep + n;
    TR_TRY(hipMemcpyAsync(dPs, Ps, (size_t)n_frames * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    TR_TRY(hipMemcpyAsync(dRs, Rs, (size_t)n_frames * 9 * sizeof(double), hipMemcpyHostToDevice, st));
    TR_TRY(hipMemcpyAsync(dp0, pt0, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    TR_TRY(hipMemcpyAsync(dp1, pt1, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    TR_TRY(hipMemcpyAsync(d_s, start_frame, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    A.Ps = dPs; A.Rs = dRs; A.start = d_s; A.pt0 = dp0; A.pt1 = dp1; A.depth = ddep; A.world = dw;
    hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, st, A);
    TR_TRY(hipGetLastError());
    TR_TRY(hipMemcpyAsync(depth, ddep, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    TR_TRY(hipMemcpyAsync(pts_world, dw, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    TR_TRY(hipStreamSynchronize(st));
#undef TR_TRY
    cleanup();
    return SWF_OK;
}

// =====================================================================================================================
// (3) inverse-depth projection factors, evaluated for a batch, one lane per factor:
//     ProjectionTwoFrameOneCamFactor (R/factor/projection_factor.cpp:179-256)   kind 0: pose_i, pose_j, ex, lambda
//     ProjectionTwoFrameTwoCamFactor (:77-166)                                  kind 1: pose_i, pose_j, ex, ex2, lambda
//     ProjectionOneFrameTwoCamFactor (:269-329)                                 kind 2: ex, ex2, lambda (no lever arm)
// The reference compiles them out by default (USE_INVERSE_DEPTH 0, R/parameter/parameters.h:25).  This is the stand-alone batched
// evaluator; inside the solve loop the same d_idepth_eval (swf_dev.h) runs as a segment of k_eval_ps / k_eval_idp and the feature is
// a scalar group-0 block eliminated by k_clique_elim (swf_add_projection_inverse_depth).
// =====================================================================================================================
namespace {
struct IdepthArgs {
    const int* kind; const int* idx;           // [n], [n][5] = pose_i, pose_j, ex, ex2, lambda (-1 where the kind has none)
    const double* poses; const double* lambda; const double* pts;     // [n_pose][7], [n_lambda], [n][6] = pts_i, pts_j
    double* r; double* J;                      // [n][2], [n][50] = J_pose_i (2x6) | J_pose_j | J_ex | J_ex2 | J_lambda (2)
    double sqrt_info, pbg[3];
    int n;
};

__global__ void __launch_bounds__(128) k_eval_idepth(IdepthArgs A) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.n) return;
    const int kind = A.kind[q];
    const int* ix = A.idx + (size_t)q * 5;
    const double zero7[7] = { 0, 0, 0, 0, 0, 0, 1 };
    const double* Pi = kind == 2 ? zero7 : A.poses + (size_t)ix[0] * 7;
    const double* Pj = kind == 2 ? zero7 : A.poses + (size_t)ix[1] * 7;
    const double* ex = A.poses + (size_t)ix[2] * 7;
    const double* e2 = kind == 0 ? ex : A.poses + (size_t)ix[3] * 7;
    double r2[2], J[50];
    d_idepth_eval(kind, Pi, Pj, ex, e2, A.lambda[ix[4]], A.pts + (size_t)q * 6, A.sqrt_info, A.pbg, r2, J, true);
    A.r[(size_t)q * 2] = r2[0]; A.r[(size_t)q * 2 + 1] = r2[1];
    for (int k = 0; k < 50; k++) A.J[(size_t)q * 50 + k] = J[k];
}
}  // namespace

extern "C" int swf_eval_inverse_depth_batch(const int32_t* kind, const int32_t* idx, int32_t n, const double* poses, int32_t n_pose,
                                            const double* lambda, int32_t n_lambda, const double* pts, double sqrt_info, const double pbg[3],
                                            double* r, double* J, int32_t on_device, void* stream) {
    if (!kind || !idx || !poses || !lambda || !pts || !pbg || !r || !J || n < 0 || n_pose <= 0 || n_lambda <= 0)
        return pi_fail(SWF_E_INVALID, "swf_eval_inverse_depth_batch: bad arguments");
    if (n == 0) return SWF_OK;
    hipStream_t st = (hipStream_t)stream;
    IdepthArgs A{};
    A.sqrt_info = sqrt_info; A.n = n;
    for (int k = 0; k < 3; k++) A.pbg[k] = pbg[k];
    if (on_device) {
        A.kind = kind; A.idx = idx; A.poses = poses; A.lambda = lambda; A.pts = pts; A.r = r; A.J = J;
        hipLaunchKernelGGL(k_eval_idepth, dim3((n + 127) / 128), dim3(128), 0, st, A);
        PI_HIPCHK(hipGetLastError());
        return SWF_OK;
    }
    for (int q = 0; q < n; q++) {
        const int32_t* ix = idx + (size_t)q * 5;
        const int k = kind[q];
        if (k < 0 || k > 2) return pi_fail(SWF_E_INVALID, "swf_eval_inverse_depth_batch: kind must be 0, 1 or 2");
        auto okp = [&](int v) { return v >= 0 && v < n_pose; };
        if ((k != 2 && (!okp(ix[0]) || !okp(ix[1]))) || !okp(ix[2]) || (k != 0 && !okp(ix[3])) || ix[4] < 0 || ix[4] >= n_lambda)
            return pi_fail(SWF_E_INVALID, "swf_eval_inverse_depth_batch: index out of range");
    }
    void* bufs[7] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    auto cleanup = [&]() { for (void* b_ : bufs) (void)hipFree(b_); };
    const size_t sz[7] = { (size_t)n * sizeof(int), (size_t)n * 5 * sizeof(int), (size_t)n_pose * 7 * sizeof(double), (size_t)n_lambda * sizeof(double),
                           (size_t)n * 6 * sizeof(double), (size_t)n * 2 * sizeof(double), (size_t)n * 50 * sizeof(double) };
    const void* src[5] = { kind, idx, poses, lambda, pts };
#define ID_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return pi_fail(SWF_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    for (int b_ = 0; b_ < 7; b_++) ID_TRY(hipMalloc(&bufs[b_], sz[b_]));
    for (int b_ = 0; b_ < 5; b_++) ID_TRY(hipMemcpyAsync(bufs[b_], src[b_], sz[b_], hipMemcpyHostToDevice, st));
    A.kind = (const int*)bufs[0]; A.idx = (const int*)bufs[1]; A.poses = (const double*)bufs[2]; A.lambda = (const double*)bufs[3];
    A.pts = (const double*)bufs[4]; A.r = (double*)bufs[5]; A.J = (double*)bufs[6];
    hipLaunchKernelGGL(k_eval_idepth, dim3((n + 127) / 128), dim3(128), 0, st, A);
    ID_TRY(hipGetLastError());
    ID_TRY(hipMemcpyAsync(r, bufs[5], sz[5], hipMemcpyDeviceToHost, st));
    ID_TRY(hipMemcpyAsync(J, bufs[6], sz[6], hipMemcpyDeviceToHost, st));
    ID_TRY(hipStreamSynchronize(st));
#undef ID_TRY
    cleanup();
    return SWF_OK;
}
