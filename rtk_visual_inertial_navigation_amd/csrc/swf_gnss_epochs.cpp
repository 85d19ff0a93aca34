// swf_gnss_epochs.cpp — the construction side of the composite IMU-GNSS factor (second half).
//
// What the reference does per GNSS epoch before the window ever sees it (GnssPreprocess, R/swf/swf_gnss.cpp:504-532): the
// epoch's raw factors go into a private MarginalizationInfo, marginalize() (R/factor/marginalization_factor.cpp:260-377)
// eliminates the receiver clocks, and the linear prior over {pose, speed-bias, ambiguities, dummy} that remains is what
// IMUGNSSBase::AddMargInfo (R/factor/gnss_imu_factor.cpp:245-352) files into the composite factor's per-epoch arrays.
//
// Here:
//   swf_batch_marginal_priors  the elimination, for ALL epochs at once: every epoch is a (tiny) flat window whose kept blocks
//                              are its parameter_head tail; one batch through the engine — factor kernels, clique elimination
//                              of the clocks, dense factorisation — and the marginalisation consumer (k_marginalize) on top.
//   swf_composite_assemble     AddMargInfo's bookkeeping for a chain of epochs, on the host: the union of the ambiguity blocks in
//                              first-seen order, and the priors' blocks scattered into Hpp / HpN / rhs_p / HNN / rhsN — exactly
//                              the arrays swf_add_imu_gnss / swf_flat_window::comp_* take.
// Host orchestration only; all arithmetic of the elimination runs in the HIP kernels (no CPU path).
#include <cstring>
#include <string>
#include <vector>
#include "../../include/swf_solver.h"

void swf_internal_set_error(const std::string& m);
static int efail(int code, const std::string& m) { swf_internal_set_error(m); return code; }

extern "C" {

int swf_batch_marginal_priors(const swf_flat_window* const* windows, int32_t n, double eps, int32_t form,
                              int32_t* dims, int32_t* ranks, double* A, double* b, double* J, double* r0, void* stream) {
    if (!windows || n <= 0 || !dims) return efail(SWF_E_INVALID, "swf_batch_marginal_priors: bad arguments");
    if (!A && !b && !J && !r0 && !ranks) {
        // sizing call: the dimension of a prior is the sum of the local sizes of its window's variable tail blocks — no device work
        for (int i = 0; i < n; i++) {
            const swf_flat_window* w = windows[i];
            if (!w || w->n_tail < 0 || w->n_tail > w->n_order) return efail(SWF_E_INVALID, "swf_batch_marginal_priors: bad window");
            int d = 0;
            for (int k = w->n_order - w->n_tail; k < w->n_order; k++) {
                const int g = w->order_block[k];
                if (g < 0 || g >= w->n_pose + w->n_sb + w->n_lm + w->n_sc) return efail(SWF_E_INVALID, "swf_batch_marginal_priors: block id out of range");
                if (w->is_const && w->is_const[g]) continue;
                d += g < w->n_pose ? 6 : g < w->n_pose + w->n_sb ? 9 : g < w->n_pose + w->n_sb + w->n_lm ? 3 : 1;
            }
            dims[i] = d;
        }
        return SWF_OK;
    }
    swf_batch* bt = nullptr;
    int rc = swf_batch_create(windows, n, stream, &bt);
    if (rc != SWF_OK) return rc;
    swf_options opt; swf_options_default(&opt);
    opt.step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY;
    std::vector<swf_summary> sm((size_t)n);
    if ((rc = swf_batch_solve(bt, &opt)) == SWF_OK && (rc = swf_batch_sync(bt)) == SWF_OK && (rc = swf_batch_summaries(bt, sm.data())) == SWF_OK) {
        for (int i = 0; i < n; i++) dims[i] = sm[(size_t)i].tail_dim;
        if (A || b || J || r0 || ranks) {
            if ((rc = swf_batch_marginalize(bt, eps, form)) == SWF_OK) {
                size_t o2 = 0, o1 = 0;
                for (int i = 0; i < n && rc == SWF_OK; i++) {
                    int32_t d = 0, rk = 0;
                    rc = swf_batch_get_prior(bt, i, A ? A + o2 : nullptr, b ? b + o1 : nullptr, J ? J + o2 : nullptr, r0 ? r0 + o1 : nullptr, nullptr, &d, &rk);
                    if (ranks) ranks[i] = rk;
                    o2 += (size_t)d * d; o1 += (size_t)d;
                }
            }
        }
    }
    swf_batch_destroy(bt);
    return rc;
}

int swf_composite_assemble(int32_t M, const int32_t* n_kept, const int32_t* kept_size, double* const* kept_key,
                           const double* A, const double* b, int32_t N_cap, double** N_keys, int32_t* N_out,
                           double* Hpp, double* HpN, double* rhs_p, double* HNN, double* rhsN) {
    if (M < 1 || !n_kept || !kept_size || !kept_key || !A || !b || !N_out) return efail(SWF_E_INVALID, "swf_composite_assemble: bad arguments");
    // pass 1: the ambiguity (scalar) blocks in first-seen order over the epochs — gnss_phase_biases (:268-283)
    std::vector<double*> keys;
    {
        size_t q = 0;
        for (int e = 0; e < M; e++)
            for (int k = 0; k < n_kept[e]; k++, q++) {
                const int sz = kept_size[q];
                if (sz != 7 && sz != 9 && sz != 1) return efail(SWF_E_INVALID, "swf_composite_assemble: kept blocks must be poses (7), speed-biases (9) or scalars (1)");
                if (sz != 1) continue;
                bool seen = false;
                for (double* p : keys) if (p == kept_key[q]) { seen = true; break; }
                if (!seen) keys.push_back(kept_key[q]);
            }
    }
    const int N = (int)keys.size();
    *N_out = N;
    if (!Hpp || !HpN || !rhs_p || !HNN || !rhsN) return SWF_OK;          // sizing call
    if (N > N_cap) return efail(SWF_E_INVALID, "swf_composite_assemble: more ambiguity blocks than the caller's buffers hold");
    if (N_keys) for (int i = 0; i < N; i++) N_keys[i] = keys[(size_t)i];
    // one stride convention for both bookkeeping calls: the ambiguity dimension of HpN / HNN / rhsN is laid out for N_cap columns (what
    // the caller sized them for), here and in swf_composite_add_mid_prior — a caller that leaves room for ambiguities a later middle
    // marginalisation introduces chains the two calls on the same buffers
    const size_t ldN = (size_t)N_cap;
    memset(Hpp, 0, sizeof(double) * (size_t)M * 225); memset(HpN, 0, sizeof(double) * (size_t)M * 15 * ldN); memset(rhs_p, 0, sizeof(double) * (size_t)M * 15);
    memset(HNN, 0, sizeof(double) * ldN * ldN); memset(rhsN, 0, sizeof(double) * ldN);
    // pass 2: scatter every epoch's prior (A_e over its kept blocks' local coordinates, b_e) — :299-347
    size_t q0 = 0, oA = 0, ob = 0;
    for (int e = 0; e < M; e++) {
        const int nk = n_kept[e];
        // local offset of every kept block inside the prior, and where it goes: pose -> rows 0..5 of the 15-block, speed-bias -> rows
        // 6..14, scalar -> its index among the ambiguities
        std::vector<int> off((size_t)nk), dst((size_t)nk), len((size_t)nk);
        int dim = 0, n_pose = 0, n_sb = 0;
        for (int k = 0; k < nk; k++) {
            const int sz = kept_size[q0 + k];
            off[(size_t)k] = dim; len[(size_t)k] = sz == 7 ? 6 : sz; dim += len[(size_t)k];
            if (sz == 7) { dst[(size_t)k] = -1; n_pose++; }
            else if (sz == 9) { dst[(size_t)k] = -2; n_sb++; }
            else { int ix = 0; while (keys[(size_t)ix] != kept_key[q0 + k]) ix++; dst[(size_t)k] = ix; }
        }
        if (n_pose > 1 || n_sb > 1) return efail(SWF_E_INVALID, "swf_composite_assemble: an epoch's prior keeps more than one pose or speed-bias");
        const double* Ae = A + oA; const double* be = b + ob;
        double* Hp = Hpp + (size_t)e * 225; double* HN = HpN + (size_t)e * 15 * ldN; double* rp = rhs_p + (size_t)e * 15;
        for (int k1 = 0; k1 < nk; k1++) {
            const int d1 = dst[(size_t)k1], s1 = d1 == -1 ? 0 : 6;            // row shift inside the 15-block
            for (int i = 0; i < len[(size_t)k1]; i++) {
                const int r = off[(size_t)k1] + i;
                if (d1 >= 0) rhsN[d1] += be[r]; else rp[s1 + i] += be[r];
                for (int k2 = 0; k2 < nk; k2++) {
                    const int d2 = dst[(size_t)k2], s2 = d2 == -1 ? 0 : 6;
                    for (int j = 0; j < len[(size_t)k2]; j++) {
                        const double v = Ae[(size_t)r * dim + off[(size_t)k2] + j];
                        if (d1 >= 0 && d2 >= 0) HNN[(size_t)d1 * ldN + d2] += v;
                        else if (d1 < 0 && d2 < 0) Hp[(s1 + i) * 15 + s2 + j] += v;
                        else if (d1 < 0 && d2 >= 0) HN[(size_t)(s1 + i) * ldN + d2] += v;
                        // (d1 >= 0, d2 < 0) is the transpose of the case above: the factor stores H_pN only
                    }
                }
            }
        }
        q0 += (size_t)nk; oA += (size_t)dim * dim; ob += (size_t)dim;
    }
    return SWF_OK;
}

int swf_composite_add_mid_prior(int32_t M, int32_t k, int32_t n_kept, const int32_t* kept_size, const int32_t* kept_epoch,
                                double* const* kept_key, const double* A, const double* b, int32_t N_cap, double** N_keys,
                                int32_t* N_io, double* Hpp, double* HpN, double* rhs_p, double* HNN, double* rhsN, double* H12) {
    if (M < 2 || k < 1 || k > M - 1 || n_kept < 1 || !kept_size || !kept_key || !A || !b || !N_io || !Hpp || !HpN || !rhs_p || !HNN || !rhsN || !H12 || (N_cap > 0 && !N_keys))
        return efail(SWF_E_INVALID, "swf_composite_add_mid_prior: bad arguments");
    int N = *N_io;
    // where every kept block goes: epoch k-1 / k (pose rows 0..5, speed-bias rows 6..14 of the epoch's 15-block) or an ambiguity
    std::vector<int> off((size_t)n_kept), len((size_t)n_kept), dst((size_t)n_kept), shift((size_t)n_kept);   // dst: -1 epoch k-1, -2 epoch k, >= 0 ambiguity
    int dim = 0;
    for (int q = 0; q < n_kept; q++) {
        const int sz = kept_size[q];
        off[(size_t)q] = dim; len[(size_t)q] = sz == 7 ? 6 : sz; dim += len[(size_t)q]; shift[(size_t)q] = sz == 9 ? 6 : 0;
        if (sz == 7 || sz == 9) {
            if (!kept_epoch || (kept_epoch[q] != k - 1 && kept_epoch[q] != k)) return efail(SWF_E_INVALID, "swf_composite_add_mid_prior: a pose / speed-bias block must belong to epoch k-1 or k");
            dst[(size_t)q] = kept_epoch[q] == k - 1 ? -1 : -2;
        } else if (sz == 1) {
            int ix = 0;
            while (ix < N && N_keys[ix] != kept_key[q]) ix++;
            if (ix == N) {                                        // first seen here: appended (N_size_external, :141-176)
                if (N >= N_cap) return efail(SWF_E_INVALID, "swf_composite_add_mid_prior: more ambiguity blocks than the caller's buffers hold");
                N_keys[N++] = kept_key[q];
            }
            dst[(size_t)q] = ix;
        } else return efail(SWF_E_INVALID, "swf_composite_add_mid_prior: kept blocks must be poses (7), speed-biases (9) or scalars (1)");
    }
    *N_io = N;
    memset(H12, 0, sizeof(double) * 225);
    double* Hp[2] = { Hpp + (size_t)(k - 1) * 225, Hpp + (size_t)k * 225 };
    double* HN[2] = { HpN + (size_t)(k - 1) * 15 * N_cap, HpN + (size_t)k * 15 * N_cap };
    double* rp[2] = { rhs_p + (size_t)(k - 1) * 15, rhs_p + (size_t)k * 15 };
    for (int q1 = 0; q1 < n_kept; q1++)
        for (int i = 0; i < len[(size_t)q1]; i++) {
            const int r = off[(size_t)q1] + i, d1 = dst[(size_t)q1], s1 = shift[(size_t)q1];
            if (d1 >= 0) rhsN[d1] += b[r]; else rp[-1 - d1][s1 + i] += b[r];
            for (int q2 = 0; q2 < n_kept; q2++)
                for (int j = 0; j < len[(size_t)q2]; j++) {
                    const int d2 = dst[(size_t)q2], s2 = shift[(size_t)q2];
                    const double v = A[(size_t)r * dim + off[(size_t)q2] + j];
                    if (d1 >= 0 && d2 >= 0) HNN[(size_t)d1 * N_cap + d2] += v;
                    else if (d1 < 0 && d2 >= 0) HN[-1 - d1][(size_t)(s1 + i) * N_cap + d2] += v;
                    else if (d1 < 0 && d2 < 0 && d1 == d2) Hp[-1 - d1][(s1 + i) * 15 + s2 + j] += v;
                    else if (d1 == -1 && d2 == -2) H12[(s1 + i) * 15 + s2 + j] = v;           // pose1_pose2_hessians (:224)
                    // (ambiguity, epoch) and (epoch k, epoch k-1) are the transposes of cases above
                }
        }
    return SWF_OK;
}

// MarginalizationInfo::ResetLinearizationPoint (R/factor/marginalization_factor.cpp:232-258; call site: the middle-marginalisation
// path, R/swf/swf_core.cpp:636-637).  Host bookkeeping on a prior's own small arrays, as in the reference: dx between the new values
// and the stored linearisation point per kept block (vector blocks x - x0; pose blocks [p - p0 ; +-2 vec(q0^-1 q)], the sign that
// makes the scalar part non-negative — the same dx MarginalizationFactor::Evaluate forms, :416-431), then
// linearized_residuals += linearized_jacobians dx,  b += A dx,  keep_block_data <- parameters.
int swf_prior_reset_linearization_point(int32_t n_kept, const int32_t* kept_size, const double* const* x_new, int32_t dim,
                                        const double* J, const double* A, double* r0, double* b, double* x0) {
    if (n_kept < 1 || !kept_size || !x_new || dim < 1 || !x0 || (J && !r0) || (A && !b))
        return efail(SWF_E_INVALID, "swf_prior_reset_linearization_point: bad arguments");
    std::vector<double> dx((size_t)dim);
    int d = 0; size_t g = 0;
    for (int k = 0; k < n_kept; k++) {
        const int sz = kept_size[k], loc = sz == 7 ? 6 : sz;
        if (sz < 1 || !x_new[k]) return efail(SWF_E_INVALID, "swf_prior_reset_linearization_point: bad kept block");
        if (d + loc > dim) return efail(SWF_E_INVALID, "swf_prior_reset_linearization_point: the kept blocks' local sizes exceed dim");
        const double* x = x_new[k]; const double* y = x0 + g;
        if (sz != 7) for (int i = 0; i < sz; i++) dx[(size_t)(d + i)] = x[i] - y[i];
        else {
            for (int i = 0; i < 3; i++) dx[(size_t)(d + i)] = x[i] - y[i];
            // q0^-1 (conjugate over the squared norm, Eigen's inverse()) times q, storage (x, y, z, w)
            const double n2 = y[3] * y[3] + y[4] * y[4] + y[5] * y[5] + y[6] * y[6];
            const double ax = -y[3] / n2, ay = -y[4] / n2, az = -y[5] / n2, aw = y[6] / n2;
            const double bx = x[3], by = x[4], bz = x[5], bw = x[6];
            const double qx = aw * bx + ax * bw + ay * bz - az * by;
            const double qy = aw * by - ax * bz + ay * bw + az * bx;
            const double qz = aw * bz + ax * by - ay * bx + az * bw;
            const double qw = aw * bw - ax * bx - ay * by - az * bz;
            const double sg = (qw >= 0) ? 2.0 : -2.0;
            dx[(size_t)(d + 3)] = sg * qx; dx[(size_t)(d + 4)] = sg * qy; dx[(size_t)(d + 5)] = sg * qz;
        }
        d += loc; g += (size_t)sz;
    }
    if (d != dim) return efail(SWF_E_INVALID, "swf_prior_reset_linearization_point: the kept blocks' local sizes do not add up to dim");
    if (J) for (int r = 0; r < dim; r++) { double a = 0; for (int c = 0; c < dim; c++) a += J[(size_t)r * dim + c] * dx[(size_t)c]; r0[r] += a; }
    if (A) for (int r = 0; r < dim; r++) { double a = 0; for (int c = 0; c < dim; c++) a += A[(size_t)r * dim + c] * dx[(size_t)c]; b[r] += a; }
    g = 0;
    for (int k = 0; k < n_kept; k++) { memcpy(x0 + g, x_new[k], sizeof(double) * (size_t)kept_size[k]); g += (size_t)kept_size[k]; }
    return SWF_OK;
}

}  // extern "C"
