/*
 * swf_types.h — plain-C data formats shared by every side of the solver boundary.
 *
 * A "window" is one sliding-window non-linear least-squares problem of the
 * RTK-visual-inertial filter: parameter blocks (poses, speed-biases, landmarks,
 * scalars) plus typed factor records that refer to them by pool index.  It is the
 * flat, index-based restatement of what the reference keeps as a pointer graph
 * inside ceres::Problem (R/swf/swf_core.cpp:209-361, R/swf/swf_image.cpp:65-114).
 *
 * Parameter layouts follow the reference exactly:
 *   pose         [px py pz qx qy qz qw]              R/swf/swf.cpp:142-149
 *   speed-bias   [v(3) ba(3) bg(3)]                   R/swf/swf.cpp:152-162
 *   landmark     world xyz                            R/feature/feature_manager.h:51
 *   scalar       ambiguity / receiver clock / dummy   R/swf/swf.cpp:61
 *
 * Global block id (used by ordering, priors, is_const):
 *   pose i -> i ; sb i -> n_pose+i ; lm i -> n_pose+n_sb+i ; scalar i -> n_pose+n_sb+n_lm+i
 *
 * (R/ = /root/reference/rtk_visual_inertial_src/rtk_visual_inertial/src/)
 */
#ifndef SWF_TYPES_H
#define SWF_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- IMU pre-integration record (R/factor/integration_base.h:28-46) ------------------
 * doubles, in this order:                                          offset
 *   delta_p[3] delta_q[4](x y z w) delta_v[3]                       0, 3, 7
 *   linearized_ba[3] linearized_bg[3]                               10, 13
 *   dp_dba[9] dp_dbg[9] dq_dbg[9] dv_dba[9] dv_dbg[9] (row-major)   16, 25, 34, 43, 52
 *   sum_dt                                                          61
 *   gyri[3] gyrj[3]   (first / last gyro sample of the interval)    62, 65
 *   sqrt_info[225]    (15x15 row-major, upper triangular)           68
 */
enum {
    SWF_PRE_DP = 0, SWF_PRE_DQ = 3, SWF_PRE_DV = 7, SWF_PRE_LBA = 10, SWF_PRE_LBG = 13,
    SWF_PRE_DP_DBA = 16, SWF_PRE_DP_DBG = 25, SWF_PRE_DQ_DBG = 34, SWF_PRE_DV_DBA = 43,
    SWF_PRE_DV_DBG = 52, SWF_PRE_SUMDT = 61, SWF_PRE_GYRI = 62, SWF_PRE_GYRJ = 65,
    SWF_PRE_SQRTINFO = 68, SWF_PRE_DOUBLES = 293
};

/* carrier-phase record doubles: sat[3] L1_lam lam el dt_br mea_var use_istd
 * (ctor args of RTKCarrierPhaseFactor, R/factor/gnss_factor.h:8-40) */
enum { SWF_CP_DOUBLES = 9 };
/* pseudorange record doubles: sat[3] P1 el dt_br mea_var   (R/factor/gnss_factor.h:43-66) */
enum { SWF_PR_DOUBLES = 7 };
/* Doppler record doubles: sat[3] satvel[3] D1_lam istd       (R/factor/gnss_factor.h:108-131) */
enum { SWF_DOP_DOUBLES = 8 };
/* rover-only pseudorange record doubles: sat[3] P1 istd       (ctor of SppPseudorangeFactor, R/factor/gnss_factor.h:70-83) */
enum { SWF_SPR_DOUBLES = 5 };
/* rover-only carrier-phase record doubles: sat[3] L1_lam istd lam   (SppCarrierPhaseFactor, R/factor/gnss_factor.h:88-104) */
enum { SWF_SCP_DOUBLES = 6 };
/* fixed-integer record doubles: N21 istd                      (FixedIntegerFactor, R/factor/gnss_factor.h:135-139) */
enum { SWF_FIX_DOUBLES = 2 };

typedef struct swf_flat_window {
    /* ---- parameter pools: caller-owned; read at solve start, written back at solve end */
    int32_t n_pose;  double* pose;   /* [n_pose][7]  keyframe poses AND camera extrinsics */
    int32_t n_sb;    double* sb;     /* [n_sb][9] */
    int32_t n_lm;    double* lm;     /* [n_lm][3] */
    int32_t n_sc;    double* sc;     /* [n_sc] */
    const uint8_t* is_const;         /* [n_blocks] by global block id; 1 = SetParameterBlockConstant */

    /* ---- elimination order (ceres::ParameterBlockOrdering as filled by MyOrdering,
     *      R/swf/swf_gnss.cpp:629-783).  Every non-constant block exactly once, groups
     *      ascending.  Group 0 = independent set eliminated in parallel; every later group
     *      holds one block and fixes that block's position in the dense reduced system. */
    int32_t n_order;
    const int32_t* order_block;      /* [n_order] global block ids */
    const int32_t* order_group;      /* [n_order] */
    int32_t n_tail;                  /* trailing entries that are ceres::internal::parameter_head */

    /* ---- visual reprojection factors, projection_factor<2,7,7,3>
     *      (R/factor/projection_factor.cpp:13-65); CauchyLoss(proj_loss_a) if > 0 */
    int32_t n_proj;
    const int32_t* proj_idx;         /* [n_proj][3] pose, extrinsic (pose pool), landmark */
    const double*  proj_uv;          /* [n_proj][2] normalised image coordinates */
    double proj_sqrt_info;           /* FOCAL_LENGTH / FEATUREWEIGHTINVERSE, R/swf/swf.cpp:47 */
    double proj_loss_a;

    /* ---- IMU factors, IMUFactor<15,7,9,7,9> (R/factor/imu_factor.cpp:5-101) */
    int32_t n_imu;
    const int32_t* imu_idx;          /* [n_imu][4] pose_i, sb_i, pose_j, sb_j */
    const double*  imu_pre;          /* [n_imu][SWF_PRE_DOUBLES] */

    /* ---- RTK carrier-phase factors <1,7,1,1> (R/factor/gnss_factor.cpp:105-138) */
    int32_t n_cp;
    const int32_t* cp_idx;           /* [n_cp][3] pose, ambiguity (scalar pool), clock (scalar pool) */
    const double*  cp_dat;           /* [n_cp][SWF_CP_DOUBLES] */

    /* ---- RTK pseudorange factors <1,7,1> (R/factor/gnss_factor.cpp:140-168) */
    int32_t n_pr;
    const int32_t* pr_idx;           /* [n_pr][2] pose, clock */
    const double*  pr_dat;           /* [n_pr][SWF_PR_DOUBLES] */

    /* ---- Doppler factors <1,9,1,7> (R/factor/gnss_factor.cpp:174-212) */
    int32_t n_dop;
    const int32_t* dop_idx;          /* [n_dop][3] speed-bias, clock drift (scalar), pose */
    const double*  dop_dat;          /* [n_dop][SWF_DOP_DOUBLES] */

    /* ---- scalar anchors, InitialBlackFactor<1,1>: r = w*x (R/factor/initial_factor.cpp:81-87) */
    int32_t n_sp;
    const int32_t* sp_idx;           /* [n_sp] scalar pool index */
    const double*  sp_w;             /* [n_sp] */

    /* ---- rover-only pseudorange factors, SppPseudorangeFactor<1,7,1> (R/factor/gnss_factor.cpp:9-39):
     *      r = istd (range + clock - P1) */
    int32_t n_spr;
    const int32_t* spr_idx;          /* [n_spr][2] pose, receiver clock (scalar pool) */
    const double*  spr_dat;          /* [n_spr][SWF_SPR_DOUBLES] */

    /* ---- rover-only carrier-phase factors, SppCarrierPhaseFactor<1,7,1,1> (R/factor/gnss_factor.cpp:45-80):
     *      r = istd (range + clock - N lam - L1_lam); NB block order pose, clock, ambiguity (RTK: pose, ambiguity, clock) */
    int32_t n_scp;
    const int32_t* scp_idx;          /* [n_scp][3] pose, receiver clock, ambiguity */
    const double*  scp_dat;          /* [n_scp][SWF_SCP_DOUBLES] */

    /* ---- fixed-integer factors, FixedIntegerFactor<1,1,1> (R/factor/gnss_factor.cpp:85-96):
     *      r = istd ((N_b - N_a) - N21), injected by the ambiguity resolution (R/swf/swf_lambda.cpp) */
    int32_t n_fix;
    const int32_t* fix_idx;          /* [n_fix][2] scalar a, scalar b */
    const double*  fix_dat;          /* [n_fix][SWF_FIX_DOUBLES] */

    /* ---- inverse-depth projection factors (R/factor/projection_factor.cpp:77-329; USE_INVERSE_DEPTH builds of the reference):
     *      the landmark is a scalar-pool block lambda = inverse depth along pts_i in the anchor frame, normally in elimination
     *      group 0.  idp_kind: 0 ProjectionTwoFrameOneCamFactor <2,7,7,7,1>   blocks pose_i, pose_j, ex, lambda
     *                1 ProjectionTwoFrameTwoCamFactor <2,7,7,7,7,1>           blocks pose_i, pose_j, ex, ex2, lambda
     *                2 ProjectionOneFrameTwoCamFactor <2,7,7,1>               blocks ex, ex2, lambda
     *      Same sqrt_info / CauchyLoss as the world-point projection factors (proj_sqrt_info, proj_loss_a). */
    int32_t n_idp;
    const int32_t* idp_kind;         /* [n_idp] */
    const int32_t* idp_idx;          /* [n_idp][5] pose_i, pose_j, ex, ex2 (pose pool; ignored where the kind has none), lambda (scalar pool) */
    const double*  idp_pts;          /* [n_idp][6] pts_i (3), pts_j (3) normalised observations */

    /* ---- composite IMU-GNSS factors, IMUGNSSFactor over IMUGNSSBase (R/factor/gnss_imu_factor.cpp:802-820, :678-799):
     *      factor f hides comp_M[f] GNSS epochs between two frames behind comp_N[f] ambiguities; residual dim 30 + N.
     *      Arrays are concatenated over the factors (layouts as in swf_composite_create, swf_solver.h).  The hidden epochs are
     *      caller-owned parameter memory like the pools: read at solve start, written back at solve end (the reference
     *      updates gnss_poses / gnss_speed_bias in place, UpdateHiddenState :601-632). */
    int32_t n_comp;
    const int32_t* comp_M;           /* [n_comp] */
    const int32_t* comp_N;           /* [n_comp] */
    const int32_t* comp_idx;         /* per factor: pose_i, sb_i, pose_j, sb_j (pool indices), then its N scalar pool indices */
    double* comp_pose;               /* [sum M][7] */
    double* comp_sb;                 /* [sum M][9] */
    const double* comp_pose_lin; const double* comp_sb_lin;
    const double* comp_Hpp;          /* [sum M][225] */
    const double* comp_HpN;          /* [sum 15 M N] */
    const double* comp_rhs_p;        /* [sum M][15] */
    const double* comp_HNN;          /* [sum N N] */
    const double* comp_rhsN;         /* [sum N] */
    const double* comp_pre;          /* [sum M + n_comp][SWF_PRE_DOUBLES] */
    /*      middle-marginalisation branch (AddMidMargInfo :121-240, Evaluate :738-759), optional (both NULL = no factor has one):
     *      comp_mid[f] = k in 1..M-1: the link e_k-1 -> e_k carries the cross block pose1_pose2_hessians of a marginalised
     *      stretch of epochs instead of an IMU factor (its pre-integration record is ignored); 0 = none.  comp_H12[f] = that block,
     *      15 x 15 row-major, rows = e_k-1, columns = e_k.  The prior's other blocks are expected inside comp_Hpp / HpN / HNN /
     *      rhs_p / rhsN already (swf_composite_add_mid_prior files them). */
    const int32_t* comp_mid;         /* [n_comp] or NULL */
    const double* comp_H12;          /* [n_comp][225] or NULL */

    /* ---- linearised priors, MarginalizationFactor (R/factor/marginalization_factor.cpp:410-446):
     *      r = r0 + J*dx, dx per kept block = x-x0, or [p-p0 ; +-2 vec(q0^-1 q)] for poses.
     *      Records are concatenated; prior k owns prior_nblk[k] entries of prior_blk,
     *      dim = prior_dim[k] (sum of local sizes), J is dim x dim row-major. */
    int32_t n_prior;
    const int32_t* prior_nblk;       /* [n_prior] */
    const int32_t* prior_dim;        /* [n_prior] */
    const int32_t* prior_blk;        /* [sum nblk] global block ids, in kept order */
    const double*  prior_J;          /* [sum dim*dim] */
    const double*  prior_r0;         /* [sum dim] */
    const double*  prior_x0;         /* [sum global sizes] linearisation points */

    /* ---- window constants */
    double pbg[3];                   /* IMU->antenna lever arm, yaml Pbg */
    double gw[3];                    /* Rwgw * G, gravity in the world frame */
    double base[3];                  /* base-station ECEF position added to pose positions */
} swf_flat_window;

/* Solver::Options subset the reference sets (R/swf/swf.cpp:25-30) + Ceres 2.x defaults the
 * restatement adopts. */
typedef struct swf_options {
    int32_t max_num_iterations;          /* yaml MAX_NUM_ITERATIONS = 8 */
    int32_t step_mode;                   /* SWF_OPTIMIZE / SWF_ASSEMBLE_ELIMINATE_ONLY (= is_optimize) */
    int32_t num_threads;                 /* CPU oracle only */
    int32_t trust_region_strategy;       /* SWF_DOGLEG (what R/swf/swf.cpp:26 sets) / SWF_LEVENBERG_MARQUARDT (the ceres default the
                                            Solver::Options of R/swf/swf_gnss.cpp:200-216, 562-572 run with) */
    int32_t jacobi_scaling;              /* Solver::Options::jacobi_scaling.  0: what every Options block of the reference that sets it sets
                                            (R/swf/swf.cpp:27, swf_core.cpp:402,422,449, swf_gnss.cpp:138,152, swf_image.cpp:402,412).  1: ceres'
                                            default, in force for the default-options solves (R/swf/swf_gnss.cpp:205-214, 562-572): columns
                                            scaled by 1 / (1 + sqrt(diag(J^T J))) of the FIRST linearisation; supported with
                                            SWF_LEVENBERG_MARQUARDT (those solves' strategy), refused with SWF_DOGLEG */
    int32_t composite_root;              /* which square root of its (30 + N)^2 remainder a composite IMU-GNSS factor exposes INSIDE a solve:
                                            SWF_ROOT_PIVOTED_CHOLESKY (0, default: rows of a diagonally pivoted outer-product Cholesky) or
                                            SWF_ROOT_EIGEN (1: the reference's own, UpdateSchurComponent R/factor/gnss_imu_factor.cpp:454-488:
                                            SelfAdjointEigenSolver, eigenvalues <= 1e-8 dropped).  Same J^T J and J^T r on the retained
                                            range; |r|^2 differs by the constant the dropped / kept noise directions carry */
    double initial_trust_region_radius;  /* 1e4 */
    double max_trust_region_radius;      /* 1e16 */
    double min_trust_region_radius;      /* 1e-32 */
    double min_relative_decrease;        /* 1e-3 */
    double function_tolerance;           /* 1e-6 */
    double gradient_tolerance;           /* 1e-10 */
    double parameter_tolerance;          /* 1e-8 */
    double min_mu, max_mu, mu_increase_factor;   /* dogleg LM damping 1e-8, 1.0, 10 */
    double min_diagonal, max_diagonal;   /* 1e-6, 1e32 (R/factor/marginalization_factor.h:98-99) */
} swf_options;

enum { SWF_OPTIMIZE = 0, SWF_ASSEMBLE_ELIMINATE_ONLY = 1 };
enum { SWF_ROOT_PIVOTED_CHOLESKY = 0, SWF_ROOT_EIGEN = 1 };
enum { SWF_DOGLEG = 0, SWF_LEVENBERG_MARQUARDT = 1 };

/* termination codes */
enum {
    SWF_RUNNING = 0,
    SWF_CONVERGED_GRADIENT = 1,
    SWF_CONVERGED_PARAMETER = 2,
    SWF_CONVERGED_FUNCTION = 3,
    SWF_NO_CONVERGENCE = 4,          /* max_num_iterations reached */
    SWF_RADIUS_TOO_SMALL = 5,
    SWF_LINEAR_SOLVER_FAILURE = 6,
    SWF_ASSEMBLED_ONLY = 7
};

/* one row per minimizer iteration (row 0 = initial evaluation), mirrors
 * ceres::IterationSummary fields the parity tests compare */
typedef struct swf_iteration {
    double cost;               /* cost at the accepted point after this iteration */
    double cost_change;        /* x_cost - candidate_cost */
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;/* after the update */
    double model_cost_change;
    int32_t step_is_successful;
    int32_t step_is_valid;
} swf_iteration;

enum { SWF_MAX_TRACE = 64 };

/* Solver::Summary subset (R/swf/swf_image.cpp:219-230) + trace */
typedef struct swf_summary {
    double initial_cost;
    double final_cost;
    double minimizer_time_in_seconds;
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t num_iterations;          /* rows in trace minus 1 */
    int32_t termination;
    int32_t reduced_dim;             /* hs_row */
    int32_t tail_dim;                /* sum of local sizes of parameter_head blocks */
    swf_iteration trace[SWF_MAX_TRACE];
} swf_summary;

static inline void swf_options_default(swf_options* o) {
    o->max_num_iterations = 8;
    o->step_mode = SWF_OPTIMIZE;
    o->num_threads = 1;
    o->trust_region_strategy = SWF_DOGLEG;
    o->jacobi_scaling = 0;
    o->composite_root = SWF_ROOT_PIVOTED_CHOLESKY;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->min_mu = 1e-8;
    o->max_mu = 1.0;
    o->mu_increase_factor = 10.0;
    o->min_diagonal = 1e-6;
    o->max_diagonal = 1e32;
}

#ifdef __cplusplus
}
#endif
#endif /* SWF_TYPES_H */
