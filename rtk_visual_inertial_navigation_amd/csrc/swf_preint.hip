// swf_preint.hip — batched IMU pre-integration: the input producer of the IMU factor.
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

#define F9(i, j) sF[(i) * 15 + (j)]
#define V9(i, j) sV[(i) * 12 + (j)]

__global__ void __launch_bounds__(64) k_preintegrate(PreintArgs A) {
    __shared__ double jac[225], cov[225], sF[135], sV[108], Tc[135], sU[225];
    const int it = blockIdx.x, lane = threadIdx.x;
    if (it >= A.n_int) return;
    const int s0 = A.first[it], s1 = A.first[it + 1];
    const double* __restrict__ smp = A.samples + (size_t)s0 * 7;
    const int n = s1 - s0;
    const double* bb = A.bias + (size_t)it * 6;
    const double ba[3] = { bb[0], bb[1], bb[2] }, bg[3] = { bb[3], bb[4], bb[5] };
    for (int e = lane; e < 225; e += 64) { int i = e / 15, j = e - i * 15; jac[e] = (i == j) ? 1.0 : 0.0; cov[e] = 0.0; }
    for (int e = lane; e < 135; e += 64) { int i = e / 15, j = e - i * 15; sF[e] = (i == j && (i < 3 || i >= 6)) ? 1.0 : 0.0; }
    for (int e = lane; e < 108; e += 64) sV[e] = 0.0;
    __syncthreads();
    double dp[3] = { 0, 0, 0 }, dq[4] = { 0, 0, 0, 1 }, dv[3] = { 0, 0, 0 }, sum_dt = 0;
    double acc0[3] = { 0, 0, 0 }, gyr0[3] = { 0, 0, 0 };
    if (n > 0) { acc0[0] = smp[1]; acc0[1] = smp[2]; acc0[2] = smp[3]; gyr0[0] = smp[4]; gyr0[1] = smp[5]; gyr0[2] = smp[6]; }
    const double gyri[3] = { gyr0[0], gyr0[1], gyr0[2] };
    const int bi = lane / 3, bj = lane - bi * 3;          // lanes 0..8 own entry (bi, bj) of every dynamic 3x3 block
    const double nz[12] = { A.acc_n2, A.acc_n2, A.acc_n2, A.gyr_n2, A.gyr_n2, A.gyr_n2, A.acc_n2, A.acc_n2, A.acc_n2, A.gyr_n2, A.gyr_n2, A.gyr_n2 };
    for (int s = 1; s < n; s++) {
        const double dt = smp[s * 7];
        const double acc1[3] = { smp[s * 7 + 1], smp[s * 7 + 2], smp[s * 7 + 3] }, gyr1[3] = { smp[s * 7 + 4], smp[s * 7 + 5], smp[s * 7 + 6] };
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
        // F, V (integration_base.cpp:48-94); only the entries that change are rewritten
        double R0[9], R1[9], Rw[9], Ra0[9], Ra1[9], ImRw[9], M0[9], M1[9], M2[9];
        q2R(dq, R0); q2R(rq, R1);
        skew3(w, Rw); skew3(a0, Ra0); skew3(a1, Ra1);
#pragma unroll
        for (int i = 0; i < 9; i++) ImRw[i] = -Rw[i] * dt;
        ImRw[0] += 1; ImRw[4] += 1; ImRw[8] += 1;
        mat3mul(R0, Ra0, M0);             // R0 [a0]x
        mat3mul(R1, Ra1, M1);             // R1 [a1]x
        mat3mul(M1, ImRw, M2);            // R1 [a1]x (I - [w]x dt)
        if (lane < 9) {
            const int q = bi * 3 + bj;
            const double I = (bi == bj) ? 1.0 : 0.0;
            F9(0 + bi, 3 + bj) = -0.25 * M0[q] * dt * dt + -0.25 * M2[q] * dt * dt;
            F9(0 + bi, 6 + bj) = I * dt;
            F9(0 + bi, 9 + bj) = -0.25 * (R0[q] + R1[q]) * dt * dt;
            F9(0 + bi, 12 + bj) = -0.25 * M1[q] * dt * dt * -dt;
            F9(3 + bi, 3 + bj) = ImRw[q];
            F9(3 + bi, 12 + bj) = -1.0 * I * dt;
            F9(6 + bi, 3 + bj) = -0.5 * M0[q] * dt + -0.5 * M2[q] * dt;
            F9(6 + bi, 9 + bj) = -0.5 * (R0[q] + R1[q]) * dt;
            F9(6 + bi, 12 + bj) = -0.5 * M1[q] * dt * -dt;
            const double v03 = 0.25 * -M1[q] * dt * dt * 0.5 * dt, v23 = 0.5 * -M1[q] * dt * 0.5 * dt;
            V9(0 + bi, 0 + bj) = 0.25 * R0[q] * dt * dt;
            V9(0 + bi, 3 + bj) = v03;
            V9(0 + bi, 6 + bj) = 0.25 * R1[q] * dt * dt;
            V9(0 + bi, 9 + bj) = v03;
            V9(3 + bi, 3 + bj) = 0.5 * I * dt;
            V9(3 + bi, 9 + bj) = 0.5 * I * dt;
            V9(6 + bi, 0 + bj) = 0.5 * R0[q] * dt;
            V9(6 + bi, 3 + bj) = v23;
            V9(6 + bi, 6 + bj) = 0.5 * R1[q] * dt;
            V9(6 + bi, 9 + bj) = v23;
        }
        __syncthreads();
        // phase A: first nine rows of F jac and F cov (270 elements, <= 5 per lane)
        double t[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
            int e = lane + 64 * u;
            t[u] = 0;
            if (e < 270) {
                const double* X = e < 135 ? jac : cov;
                int ee = e < 135 ? e : e - 135;
                int i = ee / 15, j = ee - i * 15;
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) acc += F9(i, k) * X[k * 15 + j];
                t[u] = acc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 5; u++) {
            int e = lane + 64 * u;
            if (e < 135) jac[e] = t[u];
            else if (e < 270) {
                int ee = e - 135, i = ee / 15, j = ee - i * 15;
                Tc[ee] = t[u];
                if (j >= 9) { cov[i * 15 + j] = t[u]; cov[j * 15 + i] = t[u]; }     // unit rows of F: (F cov F^T)[i][j>=9] = (F cov)[i][j]
            }
        }
        if (lane < 6) cov[(9 + lane) * 15 + 9 + lane] += (lane < 3 ? A.acc_w2 : A.gyr_w2) * dt * dt;    // V Q V^T of the bias-walk rows
        __syncthreads();
        // phase B: leading 9x9 block  (F cov) F^T + V Q V^T
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int e = lane + 64 * u;
            if (e < 81) {
                int i = e / 9, j = e - i * 9;
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) acc += Tc[i * 15 + k] * F9(j, k);
#pragma unroll
                for (int k = 0; k < 12; k++) acc += V9(i, k) * nz[k] * V9(j, k);
                cov[i * 15 + j] = acc;
            }
        }
        // propagate(): integration_base.cpp:131-141
        double nq = sqrt(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]);
#pragma unroll
        for (int k = 0; k < 3; k++) { dp[k] = rp[k]; dv[k] = rv[k]; acc0[k] = acc1[k]; gyr0[k] = gyr1[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) dq[k] = rq[k] / nq;
        sum_dt += dt;
        __syncthreads();
    }
    double* out = A.pre + (size_t)it * SWF_PRE_DOUBLES;
    if (lane == 0) {
        for (int k = 0; k < 3; k++) {
            out[SWF_PRE_DP + k] = dp[k]; out[SWF_PRE_DV + k] = dv[k]; out[SWF_PRE_LBA + k] = ba[k]; out[SWF_PRE_LBG + k] = bg[k];
            out[SWF_PRE_GYRI + k] = gyri[k]; out[SWF_PRE_GYRJ + k] = gyr0[k];
        }
        for (int k = 0; k < 4; k++) out[SWF_PRE_DQ + k] = dq[k];
        out[SWF_PRE_SUMDT] = sum_dt;
    }
    if (lane < 45) {
        int blk = lane / 9, q = lane - blk * 9, i = q / 3, j = q - i * 3;
        const int rb[5] = { 0, 0, 3, 6, 6 }, cb[5] = { 9, 12, 12, 9, 12 };
        out[SWF_PRE_DP_DBA + lane] = jac[(rb[blk] + i) * 15 + cb[blk] + j];      // the five 3x3 blocks are contiguous in the record
    }
    // get_sqrtinfo: cov = R R^T (R upper, built from the last column backwards), sqrt_info = R^-1
    bool ok = true;
    for (int j = 14; j >= 0; j--) {
        double d = cov[j * 15 + j];
        if (!(d > 0) || !(d < 1e300)) { ok = false; break; }       // uniform: every lane reads the same value
        double sd = sqrt(d);
        __syncthreads();
        if (lane <= j) cov[lane * 15 + j] = (lane == j) ? sd : cov[lane * 15 + j] / sd;
        __syncthreads();
        for (int e = lane; e < 225; e += 64) {
            int i = e / 15, k = e - i * 15;
            if (i <= k && k < j) cov[i * 15 + k] -= cov[i * 15 + j] * cov[k * 15 + j];
        }
        __syncthreads();
    }
    for (int e = lane; e < 225; e += 64) sU[e] = 0.0;
    __syncthreads();
    if (ok && lane < 15) {
        const int c = lane;                                        // column c of U = R^-1
        sU[c * 15 + c] = 1.0 / cov[c * 15 + c];
        for (int i = c - 1; i >= 0; i--) {
            double acc = 0;
            for (int k = i + 1; k <= c; k++) acc += cov[i * 15 + k] * sU[k * 15 + c];
            sU[i * 15 + c] = -acc / cov[i * 15 + i];
        }
    }
    __syncthreads();
    for (int e = lane; e < 225; e += 64) out[SWF_PRE_SQRTINFO + e] = sU[e];
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
        hipLaunchKernelGGL(k_preintegrate, dim3(n_intervals), dim3(64), 0, st, A);
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
    hipLaunchKernelGGL(k_preintegrate, dim3(n_intervals), dim3(64), 0, st, A);
    PI_TRY(hipGetLastError());
    PI_TRY(hipMemcpyAsync(pre, d_o, (size_t)n_intervals * SWF_PRE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, st));
    PI_TRY(hipStreamSynchronize(st));
#undef PI_TRY
    cleanup();
    return SWF_OK;
}
