/*
 * swf_solver.h — C-ABI of the MI355X-native sliding-window Gauss-Newton solver.
 *
 * This is the drop-in boundary for the reference's L4<->L2 line: what
 * SWFOptimization does through ceres::Problem / ceres::Solver / the modified-Ceres
 * private globals, it does here through opaque handles and plain pointers.  No C++ or
 * torch types cross the boundary.  All functions return 0 on success, a negative
 * SWF_E_* code otherwise; the library never falls back to a CPU path: if no HIP device
 * is usable, every compute entry point returns SWF_E_NODEVICE.
 *
 * Two layers:
 *   (1) swf_batch_*   — the engine: a batch of flat windows (include/swf_types.h) resident
 *                       in HBM, solved together.  The batch-throughput path (BASELINE cfg4)
 *                       and what bench.py times.
 *   (2) swf_problem_* — the pointer-keyed, ceres::Problem-shaped surface for one window
 *                       (the single-window latency path).  It flattens to (1) with B = 1.
 *
 * R/ = /root/reference/rtk_visual_inertial_src/rtk_visual_inertial/src/
 */
#ifndef SWF_SOLVER_H
#define SWF_SOLVER_H

#include <stdint.h>
#include "swf_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SWF_OK = 0,
    SWF_E_NODEVICE = -1,      /* no usable HIP device / HIP runtime error */
    SWF_E_INVALID = -2,       /* bad argument / malformed window */
    SWF_E_UNSUPPORTED = -3,   /* structure outside what the kernels cover */
    SWF_E_NOTFOUND = -4,      /* unknown parameter block / factor id */
    SWF_E_STATE = -5          /* call order violated (e.g. export before solve) */
};

/* library / device info */
/* 100 + the number of ABI revisions.  104 (round 4): swf_timing grew by lm_schur_flops_sym / lm_schur_mfma (swf_batch_timing writes
 * sizeof(swf_timing) bytes: a caller compiled against an older header must be rebuilt); swf_composite_assemble / _add_mid_prior stride
 * HpN / HNN by N_cap; after an optimising solve swf_get_reduced / swf_batch_export_reduced return L zero outside the parameter_head tail
 * block (the whole factor only after step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY); swf_prior_reset_linearization_point added.
 * 105 (round 5): no layout change; a composite IMU-GNSS factor may touch one block of elimination group 0 (MyOrdering's own order), which
 * 104 refused with SWF_E_UNSUPPORTED; SWF_PRIOR_EIGEN keeps every direction of information above eps on healthy windows.
 * 106 (round 6): swf_options::reserved became swf_options::composite_root (same offset and size; 0 = the previous behaviour); the
 * environment variable SWF_COMP_EIGEN_ROOT is gone, as are the A/B kernels behind SWF_CHOL_RR2 / RR3 / V1, SWF_ASM_OLD, SWF_POST_*.
 * Structural limits of a group-0 clique (a non-landmark block of elimination group 0 with the factors touching it): at most 9 eliminated
 * dimensions d_e, at most 768 columns d = d_e + d_f, and — for cliques beyond one wavefront's 64 x 64 / 96 x 64, which take the
 * workgroup kernel — d_e * d <= 1536 (d <= 170 at d_e = 9, 256 at d_e = 6, 512 at d_e = 3): SWF_E_UNSUPPORTED beyond.
 * swf_abi_sizes reports sizeof(swf_options), sizeof(swf_summary), sizeof(swf_timing), sizeof(swf_flat_window), sizeof(swf_iteration) so a binding can check its own. */
int swf_version(void);
int swf_abi_sizes(int32_t out[5]);
int swf_device_count(int32_t* n);                 /* hipGetDeviceCount */
int swf_set_device(int32_t device);               /* hipSetDevice: the device of the batches / problems created next on this thread */
void swf_default_options(swf_options* opt);       /* Solver::Options as the reference sets them for the window solves (R/swf/swf.cpp:25-30: DENSE_SCHUR, DOGLEG,
                                                      8 iterations, jacobi_scaling = false) + the public Ceres 2.x defaults of SURVEY.md App. C */
const char* swf_last_error(void);                 /* thread-local message for the last failure */

/* =====================================================================================
 * (1) batch engine
 * ===================================================================================== */
typedef struct swf_batch swf_batch;

/* Build the static (symbolic) structure for n windows and upload structure + state to the
 * current device.  The windows' arrays are read during this call only; their pose/sb/lm/sc
 * pointers are remembered for swf_batch_download_state().  `stream` is a hipStream_t (or
 * NULL for the default stream) on which every later operation of this batch is enqueued. */
int swf_batch_create(const swf_flat_window* const* windows, int32_t n, void* stream, swf_batch** out);

/* ---- several GPUs of one node from ONE process (SURVEY.md 8b "batch API swf_solve_batch(handles[], n, device_mask)", 8e "one host
 * thread + one HIP stream per GPU").  Windows are independent units: they are dealt to the devices in contiguous blocks, there is no
 * data-path collective, and because swf_batch_solve only enqueues, one host thread drives every device.  A batch remembers its device;
 * every swf_batch_* call switches to it for its duration (and back), so batches of different devices may be used from the same thread.
 *   swf_batch_create_on       swf_batch_create on the given device (the caller's current device is left as it was).
 *   swf_batch_create_sharded  windows [0, n) in near-equal contiguous blocks over the devices of device_mask (bit d = device d, 0 = all
 *                             visible devices); out_batches / out_first / out_count need room for one entry per selected device.
 *   swf_solve_batches         swf_batch_solve on every batch, then swf_batch_sync on every batch: the node-level Solve.
 *   swf_batch_device          the device a batch lives on.
 *   swf_shard_partition       the block partition swf_batch_create_sharded applies (and the multi-process harness, shard.py).
 * This replaces the reference's single CPU solve loop (R/swf/swf_image.cpp:198-251) only for the batch-throughput path; the single-window
 * latency path stays on one GPU. */
int swf_batch_create_on(int32_t device, const swf_flat_window* const* windows, int32_t n, void* stream, swf_batch** out);
int swf_batch_create_sharded(const swf_flat_window* const* windows, int32_t n, uint32_t device_mask,
                             swf_batch** out_batches, int32_t* out_first, int32_t* out_count, int32_t* n_batches);
int swf_solve_batches(swf_batch* const* batches, int32_t n, const swf_options* opt);
int swf_batch_device(swf_batch* b, int32_t* device);
int swf_shard_partition(int32_t n, int32_t G, int32_t k, int32_t* first, int32_t* count);   /* shard k of G: the first n % G shards take one more (host only) */
int swf_batch_destroy(swf_batch* b);

/* Re-upload the parameter blocks from the windows' host arrays (new epoch, same structure):
 * the analogue of Vector2Double() before ceres::Solve (R/swf/swf.cpp:140, swf_image.cpp:202). */
int swf_batch_upload_state(swf_batch* b);
/* Device-side restore of the state uploaded last (so a benchmark can re-solve the same
 * windows with inputs resident in HBM). */
int swf_batch_reset_state(swf_batch* b);

/* Enqueue one full solve of every window (= ceres::Solve, R/swf/swf_image.cpp:219):
 * options.step_mode selects OPTIMIZE or ASSEMBLE_ELIMINATE_ONLY (= is_optimize=false,
 * R/swf/swf_image.cpp:404-416).  Asynchronous on the batch stream; no host round-trip
 * happens inside (accept/reject, trust-region radius and convergence live on the device). */
int swf_batch_solve(swf_batch* b, const swf_options* opt);
/* Block until the stream has drained. */
int swf_batch_sync(swf_batch* b);

/* Copy parameter blocks back into the windows' host arrays (Double2Vector, R/swf/swf.cpp:188). */
int swf_batch_download_state(swf_batch* b);
/* Per-window summaries incl. iteration trace; out[n]. */
int swf_batch_summaries(swf_batch* b, swf_summary* out);

/* Export of the reduced system of window w from the LAST linearisation, the analogue of
 * ceres::internal::{lhs_out, rhs_out, lhs_out2, hs_row} (R/swf/swf_gnss.cpp:25-94):
 *   S   hs_row x hs_row row-major, full symmetric (damping included as solved)
 *   rhs hs_row
 *   L   hs_row x hs_row row-major lower Cholesky factor, S = L L^T, in elimination order;
 *       its trailing tail_dim x tail_dim block L_nn gives L_nn L_nn^T = marginal information
 *       of the parameter_head states (R/swf/swf_gnss.cpp:85-87).
 *       After a solve with step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY (UpdateSchur's and the marginalisation's mode) L is the
 *       whole factor.  After an optimising solve of a system of up to 256 dimensions (240 up to library version 104) only the block the reference reads then
 *       (UpdateSchurHessianOnly, R/swf/swf_gnss.cpp:65-94) is kept: rows and columns from the 16-aligned index at or before the
 *       parameter_head tail; the rest is returned as zero (the factor lives in registers and is not written to HBM on solve
 *       paths).  SWF_EXPORT_L=1 in the environment keeps the whole factor everywhere.
 * Any pointer may be NULL.  hs_row is summary.reduced_dim. */
int swf_batch_export_reduced(swf_batch* b, int32_t w, double* S, double* rhs, double* L);
/* Debug/parity export of local-space vectors of window w (n_loc doubles each, ordering
 * order): gradient J^T r, squared column norms, and y = (J^T J + D)^-1 J^T r. */
int swf_batch_export_vectors(swf_batch* b, int32_t w, double* grad, double* diag, double* y);
int swf_batch_dims(swf_batch* b, int32_t w, int32_t* n_loc, int32_t* n_e, int32_t* n_red);
/* Debug/parity export of the LAST linearisation of window w, factor by factor: the residual vector r [n_res] and the dense
 * Jacobian J [n_res][n_loc] (row-major; columns = local coordinates in elimination order, as swf_batch_export_vectors) exactly
 * as the device holds them — loss-corrected (CauchyLoss corrector, R/factor/marginalization_factor.cpp:23-45), whitened.
 * Rows: the window's projection factors in the caller's order (2 rows each), then imu (15 each), cp, pr, dop, sp, spr, scp,
 * fix (1 each), idp (2 each), the linear priors (dim each), the composite factors (30 + N each, as rewritten by their last
 * re-elimination).  Columns of constant blocks do not exist.  r / J may be NULL (then only the sizes are returned).  A solve
 * with step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY leaves the linearisation at the uploaded state.  Not on any solve path: this is
 * what the parity tests compare with numpy factor code and finite differences, and what their dense normal equations are
 * assembled from. */
int swf_batch_export_jacobian(swf_batch* b, int32_t w, double* r, double* J, int32_t* n_res, int32_t* n_loc);

/* Marginalisation consumer: what the reference does with the export of an is_optimize = false solve —
 * SWFOptimization::UpdateSchur (R/swf/swf_gnss.cpp:25-61) followed by MarginalizationInfo::setmarginalizeinfo(..., Sqrt =
 * true) (R/factor/marginalization_factor.cpp:449-488).  Valid after a solve with step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY.
 * For every window, with n = the local dimension of its parameter_head tail:
 *   A  n x n   marginal information of the tail  (= S_nn - S_nm S_mm^-1 S_mn = L_nn L_nn^T)
 *   bv n       its right-hand side               (= b_n  - S_nm S_mm^-1 b_m), sign convention b = J^T r
 *   J  n x n, r0 n : the new linear prior r = r0 + J dx with J^T J = A, J^T r0 = bv:
 *     SWF_PRIOR_EIGEN     J = sqrt(Lambda+) V^T (rows by ascending eigenvalue), r0 = Lambda+^-1/2 V^T bv, eigenvalues <= eps
 *                         dropped — the reference's form; rank = number kept.  Eigenvector signs are not defined.
 *     SWF_PRIOR_CHOLESKY  J = L_nn^T, r0 = L_nn^T y_n: the same quadratic without an eigen-decomposition (rank = n).
 * The reference pseudo-inverts S_mm through an eigen-decomposition with threshold 1e-8; this library uses the Cholesky
 * factor of the solve, which is the same thing whenever S_mm is positive definite.  A marginal that is SINGULAR on the kept
 * states (an unobservable extrinsic, a direction nobody measured) is normal in the reference — its eigen square root drops
 * the null directions — while the factorisation of the whole S breaks down in the tail of such a window (the solve reports
 * SWF_LINEAR_SOLVER_FAILURE): SWF_PRIOR_EIGEN then re-factors the first m columns only and takes a rank-revealing factor of
 * A (rank < n is reported, the prior is valid); SWF_PRIOR_CHOLESKY has no such variant and reports rank -1.  A breakdown
 * inside S_mm itself is a failure in both forms (rank -1, no silent fallback).  Both forms: n <= SWF_MAX_TAIL_DIM = 640, the limit
 * of the tiled factorisation of the reduced system itself (SWF_PRIOR_EIGEN: the Jacobi iteration keeps its matrix in LDS up to n = 140; above
 * that it is a block Jacobi over many workgroups on an HBM scratch, 8-column blocks up to 576 dimensions, 4-column blocks to 640).  A
 * healthy window's eigen form keeps every direction whose information exceeds eps (the pivoted Cholesky that preconditions the sweeps
 * runs down to pivots of eps / (16 n), whatever the largest diagonal entry).  Asynchronous on the batch stream up to tails of 140 dimensions; SWF_PRIOR_EIGEN above that (the block-Jacobi
 * schedule over many workgroups) is SYNCHRONOUS: the host reads the rotation counters back every few sweeps to stop enqueueing, so the
 * call blocks the calling thread until the priors exist. */
enum { SWF_PRIOR_EIGEN = 0, SWF_PRIOR_CHOLESKY = 1 };
#define SWF_MAX_TAIL_DIM 640
#define SWF_MAX_COMPOSITE_AMBIGUITIES 64
int swf_batch_marginalize(swf_batch* b, double eps, int32_t form);
/* Results of the last swf_batch_marginalize for window w (synchronises).  Any pointer may be NULL; eig receives the n
 * eigenvalues (ascending; for SWF_PRIOR_CHOLESKY the squared diagonal of L_nn). */
int swf_batch_get_prior(swf_batch* b, int32_t w, double* A, double* bv, double* J, double* r0, double* eig, int32_t* n, int32_t* rank);

/* Ambiguity covariance hand-off, for every window of the batch, after ANY solve:
 *   A  = L_nn L_nn^T   information of the parameter_head states — what SWFOptimization::UpdateSchurHessianOnly forms from
 *                      lhs_out2 (R/swf/swf_gnss.cpp:65-94);
 *   Qy = A^-1          their covariance — what SWFOptimization::LambdaSearch computes next (R/swf/swf_lambda.cpp:94-99) and feeds,
 *                      with the float ambiguities, to the LAMBDA search (which stays on the host: integer least squares is
 *                      sequential and branchy).
 * L is the Cholesky factor of the last linear solve of each window (it includes the dogleg's mu * diag regularisation, as the
 * exported lhs_out2 does).  n_red <= 512.  swf_batch_tail_covariance is asynchronous; the getter synchronises; both matrices
 * are n x n row-major, n = tail dimension; *n = -1 and SWF_E_STATE for a window without a valid factor. */
int swf_batch_tail_covariance(swf_batch* b);
int swf_batch_get_tail_covariance(swf_batch* b, int32_t w, double* A, double* Qy, int32_t* n);

/* Timing of the last swf_batch_solve, measured with HIP events recorded on the batch stream
 * around individual kernel launches (valid after swf_batch_sync).  `mask` selects which
 * kernels get an event pair per launch (bit k = SWF_K_*); bit 0 brackets the whole solve.
 * ms[k] = summed duration, calls[k] = number of launches bracketed.
 * One id per distinct kernel.  Mutually independent small kernels run as segments of one fused grid:
 *   LM_SCHUR    = k_lm_schur        landmark elimination + reduced-camera product (cells stay in LDS)
 *   EVAL_PS     = k_eval_ps<true>   projection + scalar-factor residuals and Jacobians (round 6, dogleg: at the candidate; one window: k_step_eval,
 *                                   which carries k_dogleg at its head — no DOGLEG bracket then; SWF_K_FRAME_SUMS is never recorded any more)
 *   POST_CHOL   = k_post_chol       back-substitution + |J D^-2 g|^2 of the Cauchy point
 *   POST_DOGLEG = k_post_dogleg     cost-only candidate residuals (Levenberg-Marquardt, windows with composite factors; one window: k_step_eval<., false>)
 *   DECIDE      = k_decide          (one window, dogleg: inside k_decide_lm_clique, recorded under LM_SCHUR)
 *   CAND_EVAL   = k_eval_imu<false> + k_eval_prior<false> (candidate residuals)
 *   CLIQUE_ELIM = the (up to three) size classes of k_clique_elim
 *   ASSEMBLE    = k_assemble_flat   the reduced system and its vectors from the static assembly program */
enum { SWF_K_TOTAL = 0, SWF_K_EVAL_PS = 1, SWF_K_EVAL_IMU = 2, SWF_K_FRAME_SUMS = 3, SWF_K_EVAL_PRIOR = 4,
       SWF_K_LM_SCHUR = 5, SWF_K_CLIQUE_ELIM = 6, SWF_K_LM_ELIM = 7, SWF_K_ASSEMBLE = 8, SWF_K_CHOL = 9,
       SWF_K_POST_CHOL = 10, SWF_K_POST_DOGLEG = 11, SWF_K_DOGLEG = 12, SWF_K_CAND_EVAL = 13, SWF_K_DECIDE = 14,
       SWF_K_COUNT = 16 };
typedef struct swf_timing {
    double ms[SWF_K_COUNT];
    int32_t calls[SWF_K_COUNT];
    int64_t jacobian_bytes;  /* algorithmic bytes of ONE Jacobian evaluation of the batch */
    int64_t proj_bytes;      /* the projection-factor share of it (312 B per observation) */
    int64_t chol_flops;      /* sum over windows of n_red^3 / 3 flops = n_red^3 / 6 multiply-adds (one factorisation of the batch) */
    int64_t lm_schur_flops;  /* landmark Schur product: sum over landmarks of 216 k^2 + 108 k flops */
    int64_t n_obs;           /* projection observations in the batch */
    int32_t n_linearizations;/* Jacobian evaluations enqueued per window in the last solve */
    int32_t reserved;
    int64_t lm_schur_flops_sym;  /* the same product counted over the lower triangle only: sum over landmarks of 108 k (k - 1) + 162 k flops */
    int64_t lm_schur_mfma;       /* v_mfma_f64_16x16x4_f64 instructions one product launch executes over the batch (2048 flops each) */
} swf_timing;
int swf_batch_enable_timing(swf_batch* b, int32_t mask);
int swf_batch_timing(swf_batch* b, swf_timing* out);

/* =====================================================================================
 * Input producer: batched IMU pre-integration
 *
 * IntegrationBase ctor / push_back / propagate / midPointIntegration / get_sqrtinfo
 * (R/factor/integration_base.cpp:5-142) for n_intervals keyframe intervals, one wavefront each.
 *   samples  [first[n_intervals]][7]  dt, acc(3), gyr(3).  The first sample of an interval seeds acc_0 / gyr_0 (its dt
 *            is ignored, as the reference's constructor takes no dt); every further sample is one push_back(dt, acc, gyr).
 *   first    [n_intervals + 1]        sample offsets (interval i owns samples first[i] .. first[i+1]-1)
 *   bias     [n_intervals][6]         linearisation biases ba, bg
 *   noise    ACC_N, GYR_N, ACC_W, GYR_W (yaml; R/parameter/parameters.cpp)
 *   pre      [n_intervals][SWF_PRE_DOUBLES] records, the layout swf_add_imu / swf_flat_window::imu_pre take.
 *            A covariance that is not positive definite (an interval with no push_back) leaves sqrt_info zero.
 * on_device = 0: all pointers are host memory; the call copies in, runs, copies out and synchronises.
 * on_device = 1: all pointers are device memory; the kernel is enqueued on `stream` and the call returns.
 * ===================================================================================== */
int swf_preintegrate_batch(const double* samples, const int32_t* first, int32_t n_intervals, const double* bias,
                           const double noise[4], double* pre, int32_t on_device, void* stream);

/* Input producer: two-view landmark triangulation for a batch of features — FeatureManager::triangulate, the branch every
 * feature with >= 2 observations takes (R/feature/feature_manager.cpp:285-316), with triangulatePoint (:148-161).
 *   Ps [n_frames][3], Rs [n_frames][9] (row-major)  frame positions / rotations as the reference passes them
 *   tic, ric (row-major), pbg                        camera extrinsic and the IMU->antenna lever arm
 *   start_frame [n]                                  first observing frame i of each feature (the second view is i + 1);
 *                                                    concatenate the frames of many windows and use absolute indices
 *   pt0, pt1 [n][2]                                  normalised image coordinates in frames i and i + 1
 *   depth [n]                                        depth in camera i; init_depth (INIT_DEPTH, R/parameter/parameters.h:29) when the
 *                                                    triangulated depth is not positive; -1 for an out-of-range start_frame
 *   pts_world [n][3]                                 Rs[i] (ric (pt0 depth) + tic - pbg) + Ps[i], the landmark block's initial value
 * on_device as for swf_preintegrate_batch. */
int swf_triangulate_batch(const double* Ps, const double* Rs, int32_t n_frames, const double tic[3], const double ric[9],
                          const double pbg[3], const int32_t* start_frame, const double* pt0, const double* pt1, int32_t n,
                          double init_depth, double* depth, double* pts_world, int32_t on_device, void* stream);

/* Inverse-depth projection factors for a batch, one lane per factor — the stand-alone evaluator (parity
 * against the oracle).  Inside the solver the same factors are a factor type of the loop: swf_flat_window::idp_* /
 * swf_add_projection_inverse_depth, the inverse depth being a scalar block of elimination group 0.
 *   kind [n]      0 ProjectionTwoFrameOneCamFactor (R/factor/projection_factor.cpp:179-256): pose_i, pose_j, ex, lambda
 *                 1 ProjectionTwoFrameTwoCamFactor (:77-166): pose_i, pose_j, ex, ex2, lambda
 *                 2 ProjectionOneFrameTwoCamFactor (:269-329): ex, ex2, lambda
 *   idx  [n][5]   pose_i, pose_j, ex, ex2 (rows of poses [n_pose][7]; ignored where the kind has none), lambda (row of lambda [n_lambda])
 *   pts  [n][6]   pts_i (3), pts_j (3): the normalised observations in the anchor and in the second view
 *   r    [n][2];  J [n][50] = d r / d pose_i (2x6, local) | pose_j | ex | ex2 | lambda (2); blocks a kind does not have are zero */
int swf_eval_inverse_depth_batch(const int32_t* kind, const int32_t* idx, int32_t n, const double* poses, int32_t n_pose,
                                 const double* lambda, int32_t n_lambda, const double* pts, double sqrt_info, const double pbg[3],
                                 double* r, double* J, int32_t on_device, void* stream);

/* =====================================================================================
 * Composite IMU-GNSS factors as a batched, stateful device operator
 *
 * IMUGNSSBase / IMUGNSSFactor (R/factor/gnss_imu_factor.cpp): n factors, each hiding M[f] GNSS-epoch states between two
 * visual frames behind N[f] <= 24 ambiguities.  Arrays are concatenated over the factors in order:
 *   pose, sb            [sum M][7], [sum M][9]   hidden epochs (gnss_poses / gnss_speed_bias), copied: the handle owns them
 *   pose_lin, sb_lin    linearisation points of the per-epoch GNSS priors
 *   Hpp [sum M][225], HpN [sum 15 M N], rhs_p [sum M][15], HNN [sum N N], rhsN [sum N]   the blocks AddMargInfo :245-352 keeps
 *   pre                 [sum M + n][SWF_PRE_DOUBLES]: factor f owns M[f] + 1 records: frame_i -> e_0, ..., e_M-1 -> frame_j
 * swf_composite_evaluate = IMUGNSSBase::Evaluate (:678-799) for all factors at once:
 *   outer [n][32] = pose_i(7) sb_i(9) pose_j(7) sb_j(9), Nv [sum N] = the ambiguity values
 *   want_jac != 0: back-substitute the hidden epochs from the outer increment, re-eliminate, return the (30+N)-vector
 *                  residual and the (30+N)^2 Jacobian (row-major, columns = local [pose_i sb_i | pose_j sb_j | N]) of each factor;
 *   want_jac == 0: the linear model r_lin - J INC of the last linearisation (the first call linearises).
 * The Jacobian is the Cholesky square root L^T of the remaining system (the reference takes the eigen square root: same
 * J^T J and J^T r whenever the remainder is positive definite; status[f] = -1 reports a factor where it was not).
 * Hd / rd (optional) receive the remaining system itself ((30+N)^2, 30+N per factor).  Host pointers; synchronous.
 * ===================================================================================== */
typedef struct swf_composite swf_composite;
int swf_composite_create(int32_t n, const int32_t* M, const int32_t* N, const double* pose, const double* sb,
                         const double* pose_lin, const double* sb_lin, const double* Hpp, const double* HpN,
                         const double* rhs_p, const double* HNN, const double* rhsN, const double* pre,
                         const double pbg[3], const double gw[3], void* stream, swf_composite** out);
int swf_composite_evaluate(swf_composite* c, const double* outer, const double* Nv, int32_t want_jac,
                           double* residual, double* jac, double* Hd, double* rd, int32_t* status);
int swf_composite_hidden(swf_composite* c, double* pose, double* sb);
/* The middle-marginalisation branch (AddMidMargInfo :121-240, Evaluate :738-759): mid[f] = k in 1..M[f]-1 replaces the IMU factor of
 * the link e_k-1 -> e_k by the cross block H12[f] (15 x 15 row-major, rows = e_k-1, columns = e_k) of a marginalised stretch of
 * epochs; 0 = none.  Call before the first evaluation. */
int swf_composite_set_mid_links(swf_composite* c, const int32_t* mid, const double* H12);
/* Which square root of the remaining system the factor exposes.  SWF_ROOT_PIVOTED_CHOLESKY (default): rows v_r of a diagonally pivoted
 * outer-product Cholesky.  SWF_ROOT_EIGEN: the reference's own (UpdateSchurComponent :454-488): J = sqrt(lam+) V^T, r = lam+^-1/2 V^T rhs,
 * eigenvalues <= 1e-8 dropped, rows in ascending eigenvalue order — the reference's residual vector up to the sign of each eigenvector.
 * Both give the same J^T J and J^T r (to rounding; the pivoted root stops at pivots the eigen root's absolute 1e-8 cut may keep, and
 * the directions kept by one and dropped by the other are rounding noise of a matrix with entries ~1e8: their r_k = v_k^T rhs / sqrt(lam_k)
 * are O(1), so |r|^2 — the factor's contribution to the COST — differs by a constant of that size, the gradient and the step do not).
 * Inside a solve (swf_flat_window::comp_*) swf_options::composite_root selects the form for every composite factor of the solve
 * (enum in swf_types.h). */
int swf_composite_set_root(swf_composite* c, int32_t form);
int swf_composite_destroy(swf_composite* c);

/* =====================================================================================
 * The construction side of the composite factor: per-epoch GNSS pre-elimination and AddMargInfo
 *
 * swf_batch_marginal_priors: MarginalizationInfo::marginalize (R/factor/marginalization_factor.cpp:260-377) for n small problems
 * at once — what GnssPreprocess runs for every GNSS epoch (R/swf/swf_gnss.cpp:504-532: the epoch's raw factors, receiver clocks
 * dropped, {pose, speed-bias, ambiguities, dummy} kept).  Each problem is a flat window whose KEPT blocks are its parameter_head
 * tail (n_tail) and whose dropped blocks are ordered before it (independent ones, the clocks, in group 0); one batch through the
 * engine with step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY, then swf_batch_marginalize(eps, form).  Outputs, concatenated over the
 * problems in order: dims[i] = dimension of prior i (sum of the kept blocks' local sizes, in tail order), ranks[i], A (dims^2,
 * row-major), b, J, r0 (see swf_batch_marginalize).  A / b / J / r0 / ranks may be NULL (dims only: a sizing call).  Host
 * pointers; synchronous.  The priors are linearised at the windows' current values (the reference zeroes the ambiguities
 * first, PhaseBiasSaveAndReset, R/swf/swf_gnss.cpp:516: do the same in the windows you pass).
 *
 * swf_composite_assemble: IMUGNSSBase::AddMargInfo (R/factor/gnss_imu_factor.cpp:245-352) for a chain of M epochs — pure host
 * bookkeeping.  Epoch e keeps n_kept[e] blocks; kept_size (7 / 9 / 1) and kept_key (the block's address; only the scalars' are
 * looked at) are concatenated over the epochs in prior order, as are A (dim_e^2) and b (dim_e).  The scalar blocks become the
 * factor's ambiguities in first-seen order: *N_out of them, their keys in N_keys.  Hpp [M][225], HpN [M][15][N_cap], rhs_p [M][15],
 * HNN [N_cap][N_cap], rhsN [N_cap] receive what swf_add_imu_gnss / swf_flat_window::comp_* expect (pose rows 0..5, speed-bias rows
 * 6..14 of an epoch's 15-block; the N x N block and its right-hand side accumulate over the epochs).  Passing Hpp = NULL only counts
 * (*N_out).  N_cap = the N the caller sized HpN / HNN / rhsN / N_keys for — and the STRIDE of their ambiguity dimension, here and in
 * swf_composite_add_mid_prior alike (so the two calls chain on the same buffers; compact to *N_out / *N_io before handing the arrays
 * to swf_add_imu_gnss if N_cap was larger).  swf_batch_marginal_priors reports a window whose marginal could not be formed with
 * ranks[i] = -1 (and zeros in its slots): check the ranks before filing the priors into a composite factor.
 * ===================================================================================== */
int swf_batch_marginal_priors(const swf_flat_window* const* windows, int32_t n, double eps, int32_t form,
                              int32_t* dims, int32_t* ranks, double* A, double* b, double* J, double* r0, void* stream);
int swf_composite_assemble(int32_t M, const int32_t* n_kept, const int32_t* kept_size, double* const* kept_key,
                           const double* A, const double* b, int32_t N_cap, double** N_keys, int32_t* N_out,
                           double* Hpp, double* HpN, double* rhs_p, double* HNN, double* rhsN);
/* swf_prior_reset_linearization_point: MarginalizationInfo::ResetLinearizationPoint (R/factor/marginalization_factor.cpp:232-258;
 * called on the two priors bounding a middle marginalisation, R/swf/swf_core.cpp:636-637) — host bookkeeping on the prior's own arrays.
 * kept_size[n_kept] = global sizes of the kept blocks in kept order (7 = pose), x_new[k] = the block's current values, dim = sum of the
 * local sizes.  dx_k = x_new - x0 (pose: [p - p0 ; +-2 vec(q0^-1 q)], sign of the scalar part, as MarginalizationFactor::Evaluate);
 * r0 += J dx (linearized_residuals, J dim x dim row-major), b += A dx (either pair may be NULL), x0 (concatenated global sizes) <- x_new.
 * (The reference asserts that scalar blocks hold 0 at the call — PhaseBiasSaveAndReset zeroes them first; not enforced here.) */
int swf_prior_reset_linearization_point(int32_t n_kept, const int32_t* kept_size, const double* const* x_new, int32_t dim,
                                        const double* J, const double* A, double* r0, double* b, double* x0);
/* swf_composite_add_mid_prior: IMUGNSSBase::AddMidMargInfo (R/factor/gnss_imu_factor.cpp:121-240) — host bookkeeping.  The prior
 * (A, b) of a marginalised stretch of GNSS epochs (MargGNSSFrames, R/swf/swf_core.cpp:570-641) keeps n_kept blocks: the pose (7) and
 * speed-bias (9) of the two epochs either side of the stretch — kept_epoch[q] = k-1 or k for those blocks (ignored for scalars) — and
 * ambiguities (1, identified by kept_key).  Its diagonal / ambiguity blocks are ADDED into the factor's arrays (laid out for N_cap
 * ambiguities per row: HpN [M][15][N_cap], HNN [N_cap][N_cap], rhsN [N_cap]; N_keys holds *N_io keys on entry), ambiguities it
 * sees first are appended (*N_io grows; SWF_E_INVALID beyond N_cap), and the cross block between the two epochs is returned in
 * H12 (15 x 15 row-major, rows = e_k-1): pass k and H12 on as comp_mid / comp_H12 (swf_set_imu_gnss_mid_link,
 * swf_composite_set_mid_links).  Compact HpN / HNN to stride *N_io afterwards if N_cap was larger. */
int swf_composite_add_mid_prior(int32_t M, int32_t k, int32_t n_kept, const int32_t* kept_size, const int32_t* kept_epoch,
                                double* const* kept_key, const double* A, const double* b, int32_t N_cap, double** N_keys,
                                int32_t* N_io, double* Hpp, double* HpN, double* rhs_p, double* HNN, double* rhsN, double* H12);

/* =====================================================================================
 * (2) ceres::Problem-shaped single-window surface
 *
 * Parameter blocks are identified BY ADDRESS like in Ceres; values are read through the
 * pointer at swf_problem_solve() and written back before it returns.
 * ===================================================================================== */
typedef struct swf_problem swf_problem;
/* Factor ids are SLOT numbers: stable while the factor lives; the slot of a removed factor is handed out again (LIFO) to a later
 * swf_add_*, so an id held across its own swf_remove_factor / a cascading swf_remove_parameter_block may name a DIFFERENT live factor
 * afterwards (it does not become SWF_E_NOTFOUND) — drop ids when you remove, as with ceres::ResidualBlockId.  The flat window is built
 * in slot order, so a long-lived problem and a freshly built one with the same factors may sum in a different factor order (results
 * agree to rounding, not bit for bit). */
typedef int32_t swf_factor_id;

enum { SWF_MANIFOLD_NONE = 0, SWF_MANIFOLD_POSE = 1 };   /* PoseLocalParameterization, 7 -> 6 */

int swf_problem_create(swf_problem** out);                      /* ceres::Problem()            */
int swf_problem_destroy(swf_problem* p);

/* ceres::Problem::AddParameterBlock(ptr, size, LocalParameterization*).  size in {7,9,3,1}.  The manifold is fixed by the
 * size — a 7-block is a pose on PoseLocalParameterization whether or not `manifold` says so, everything else is Euclidean —
 * so calling it on an existing block to (re)attach the parameterization (R/swf/swf_core.cpp:53-56) is accepted and changes
 * nothing. */
int swf_add_parameter_block(swf_problem* p, double* key, int32_t size, int32_t manifold);
int swf_has_parameter_block(swf_problem* p, const double* key);           /* 1 / 0             */
int swf_remove_parameter_block(swf_problem* p, double* key);    /* cascades to its factors     */
int swf_set_parameter_block_constant(swf_problem* p, double* key);
int swf_set_parameter_block_variable(swf_problem* p, double* key);
int swf_is_parameter_block_constant(swf_problem* p, const double* key);   /* 1 / 0             */
int swf_parameter_block_size(swf_problem* p, const double* key);
int swf_num_parameter_blocks(swf_problem* p);
int swf_num_residual_blocks(swf_problem* p);

/* Typed AddResidualBlock()s.  Each returns the factor id (>= 0) or a negative error.
 * Unknown parameter blocks are added implicitly, as ceres does. */
/* projection_factor(pts) + CauchyLoss(loss_a) (R/swf/swf_image.cpp:98-100); loss_a<=0: no loss */
swf_factor_id swf_add_projection(swf_problem* p, double* pose, double* ex_pose, double* point,
                                 const double uv[2], double sqrt_info, double loss_a);
/* IMUFactor(pre_integration) (R/swf/swf_imu.cpp:185-193); pre = SWF_PRE_DOUBLES record */
swf_factor_id swf_add_imu(swf_problem* p, double* pose_i, double* sb_i, double* pose_j, double* sb_j,
                          const double* pre);
/* RTKCarrierPhaseFactor (R/swf/swf_core.cpp:113-117); dat = SWF_CP_DOUBLES record */
swf_factor_id swf_add_rtk_carrier_phase(swf_problem* p, double* pose, double* ambiguity, double* clock,
                                        const double* dat);
/* RTKPseudorangeFactor (R/swf/swf_core.cpp:130-133); dat = SWF_PR_DOUBLES record */
swf_factor_id swf_add_rtk_pseudorange(swf_problem* p, double* pose, double* clock, const double* dat);
/* SppDopplerFactor (R/swf/swf_core.cpp:197-201); dat = SWF_DOP_DOUBLES record */
swf_factor_id swf_add_doppler(swf_problem* p, double* speed_bias, double* clock_drift, double* pose,
                              const double* dat);
/* SppPseudorangeFactor (R/swf/swf_core.cpp:157-163); dat = SWF_SPR_DOUBLES record (sat[3] P1 istd) */
swf_factor_id swf_add_spp_pseudorange(swf_problem* p, double* pose, double* clock, const double* dat);
/* SppCarrierPhaseFactor (R/swf/swf_core.cpp:170-190); block order pose, clock, ambiguity as in the reference;
 * dat = SWF_SCP_DOUBLES record (sat[3] L1_lam istd lam) */
swf_factor_id swf_add_spp_carrier_phase(swf_problem* p, double* pose, double* clock, double* ambiguity,
                                        const double* dat);
/* FixedIntegerFactor(N21, istd) (R/swf/swf_lambda.cpp:318-330): r = istd ((*n_b - *n_a) - N21) */
swf_factor_id swf_add_fixed_integer(swf_problem* p, double* n_a, double* n_b, double N21, double istd);
/* ProjectionTwoFrameOneCamFactor (kind 0) / ProjectionTwoFrameTwoCamFactor (1) / ProjectionOneFrameTwoCamFactor (2)
 * (R/factor/projection_factor.h:33-66, USE_INVERSE_DEPTH builds) + CauchyLoss(loss_a): inv_depth is a one-dimensional block (the
 * feature's inverse depth along pts_i in the anchor frame), normally ordered into group 0 like the world-point landmarks.
 * Blocks a kind does not have are ignored.  sqrt_info / loss_a must equal those of the other projection factors. */
swf_factor_id swf_add_projection_inverse_depth(swf_problem* p, int32_t kind, double* pose_i, double* pose_j, double* ex, double* ex2,
                                               double* inv_depth, const double pts_i[3], const double pts_j[3], double sqrt_info, double loss_a);
/* IMUGNSSFactor(IMUGNSS_info) (R/swf/swf.cpp:713-730, R/factor/gnss_imu_factor.cpp:99-119): the composite factor over
 * (pose_i, sb_i, pose_j, sb_j, N ambiguities) hiding M GNSS epochs.  hidden_pose [M][7] / hidden_sb [M][9] are the epochs'
 * parameter memory (gnss_poses / gnss_speed_bias): read at every solve, updated in place by it.  The other arrays (copied)
 * are laid out as in swf_composite_create for one factor; pre holds M + 1 records.  Its blocks must be variable; at most ONE of them
 * may be in elimination group 0 (MyOrdering puts every other speed-bias block there, R/swf/swf_gnss.cpp:683-691: pass the ordering as it
 * is — the factor's rows then join that block's clique); N <= 64 (SWF_MAX_COMPOSITE_AMBIGUITIES). */
swf_factor_id swf_add_imu_gnss(swf_problem* p, double* pose_i, double* sb_i, double* pose_j, double* sb_j, double* const* ambiguities,
                               int32_t N, int32_t M, double* hidden_pose, double* hidden_sb, const double* pose_lin, const double* sb_lin,
                               const double* Hpp, const double* HpN, const double* rhs_p, const double* HNN, const double* rhsN, const double* pre);
/* the factor's middle-marginalisation link (IMUGNSSBase::AddMidMargInfo): see swf_flat_window::comp_mid / comp_H12.  k = 0 clears it. */
int swf_set_imu_gnss_mid_link(swf_problem* p, swf_factor_id id, int32_t k, const double* H12);
/* InitialBlackFactor(w) (R/swf/swf_core.cpp:553-556) */
swf_factor_id swf_add_scalar_prior(swf_problem* p, double* scalar, double w);
/* MarginalizationFactor(info) (R/swf/swf_core.cpp:551-552): kept blocks `keys` (sizes from the
 * problem), J dim x dim row-major, r0[dim], x0 = concatenated linearisation points. */
swf_factor_id swf_add_linear_prior(swf_problem* p, double* const* keys, int32_t n_keys,
                                   const double* J, const double* r0, const double* x0);
int swf_remove_factor(swf_problem* p, swf_factor_id id);                  /* RemoveResidualBlock */
int swf_factor_set_enabled(swf_problem* p, swf_factor_id id, int32_t on); /* ResidualBlock::is_use */
int swf_factor_is_enabled(swf_problem* p, swf_factor_id id);              /* 1 / 0; SWF_E_NOTFOUND */

/* The query surface of ceres::Problem the estimator walks when it marginalises through the solver (GlobalMarge,
 * R/swf/swf_image.cpp:350-367), re-attaches landmark blocks (R/swf/swf.cpp:413-422, R/feature/feature_manager.cpp:456-461)
 * and finds the dummy anchor's block (R/swf/swf_gnss.cpp:651).  Each getter writes the total count to *n and fills at most
 * `cap` entries (call with cap = 0 to size the buffer).  Residual blocks are reported in creation order, parameter blocks in
 * insertion order, the blocks of a residual block in the order of its AddResidualBlock call.
 *   Problem::GetResidualBlocks / GetResidualBlocksForParameterBlock / GetParameterBlocks / GetParameterBlocksForResidualBlock */
int swf_get_residual_blocks(swf_problem* p, swf_factor_id* ids, int32_t cap, int32_t* n);
int swf_get_residual_blocks_for_parameter_block(swf_problem* p, const double* key, swf_factor_id* ids, int32_t cap, int32_t* n);
int swf_get_parameter_blocks(swf_problem* p, double** keys, int32_t cap, int32_t* n);
int swf_get_parameter_blocks_for_residual_block(swf_problem* p, swf_factor_id id, double** keys, int32_t cap, int32_t* n);

/* window constants the factors read from globals in the reference (Pbg, Rwgw*G, base_xyz) */
int swf_set_constants(swf_problem* p, const double pbg[3], const double gw[3], const double base[3]);

/* options.linear_solver_ordering: ParameterBlockOrdering::AddElementToGroup() per entry
 * (R/swf/swf_gnss.cpp:629-783).  Replaces any previous ordering; setting the ordering it already has is free (no structure
 * change).  n = 0 (= a null linear_solver_ordering, the default Solver::Options of R/swf/swf_gnss.cpp:200-216, 562-572) asks
 * for an automatic ordering: every landmark and a greedy independent set of scalars in group 0, one group per remaining
 * variable block in insertion order, the export tail last. */
int swf_set_ordering(swf_problem* p, double* const* keys, const int32_t* groups, int32_t n);
/* ceres::internal::parameter_head: blocks ordered last and exported (R/swf/swf_gnss.cpp:116) */
int swf_set_export_tail(swf_problem* p, double* const* keys, int32_t n);

/* ceres::Solve(options, &problem, &summary) */
int swf_problem_solve(swf_problem* p, const swf_options* opt, swf_summary* summary);
/* UpdateSchur + setmarginalizeinfo(Sqrt = true) on the problem's export tail (R/swf/swf_image.cpp:404-418: is_optimize =
 * false, Solve, UpdateSchur, setmarginalizeinfo): valid after swf_problem_solve with step_mode =
 * SWF_ASSEMBLE_ELIMINATE_ONLY.  See swf_batch_marginalize for the outputs; buffers are solver-owned (n x n row-major / n),
 * valid until the next solve, marginalize or destroy.  Any out pointer may be NULL. */
int swf_problem_marginalize(swf_problem* p, double eps, int32_t form, const double** J, const double** r0,
                            const double** A, const double** bv, int32_t* n, int32_t* rank);
/* lhs_out / rhs_out / lhs_out2 / hs_row of the last solve; buffers solver-owned, valid until the
 * next solve or destroy. */
/* UpdateSchurHessianOnly + the covariance LambdaSearch takes from it (see swf_batch_tail_covariance); pointers are owned by
 * the problem and valid until the next call / solve. */
int swf_problem_tail_covariance(swf_problem* p, const double** A, const double** Qy, int32_t* n);
int swf_get_reduced(swf_problem* p, const double** S, const double** rhs, const double** L, int32_t* hs_row);

#ifdef __cplusplus
}
#endif
#endif /* SWF_SOLVER_H */
