// swf_ceres.hpp — header-only C++ adapter: the ceres::Problem / ceres::Solver names the
// reference's estimator uses, implemented over the C-ABI of swf_solver.h.
//
// Purpose: estimator code written in the reference's style —
//     ceres::LossFunction* loss = new ceres::CauchyLoss(1.0);
//     projection_factor* f = new projection_factor(pts);
//     my_problem.AddResidualBlock(f, loss, para_pose[j], para_ex_Pose[0], point.data());
//     ceres::Solve(my_options, &my_problem, &summary);
// — compiles against this header instead of <ceres/ceres.h> + the reference's factor headers.
// The factor classes below carry only the CONSTRUCTOR ARGUMENTS (data); their arithmetic runs in
// the HIP kernels.  Nothing here evaluates a residual on the CPU.
//
// Mapping (reference file:line -> here):
//   projection_factor(pts)                      R/factor/projection_factor.h:9-18     -> swf_add_projection
//   IMUFactor(pre_integration)                  R/factor/imu_factor.h:7-19            -> swf_add_imu (SWF_PRE_DOUBLES record; from an
//                                                                                        IntegrationBase* or from the record itself)
//   RTKCarrierPhaseFactor(...)                  R/factor/gnss_factor.h:8-40           -> swf_add_rtk_carrier_phase
//   RTKPseudorangeFactor(...)                   R/factor/gnss_factor.h:43-66          -> swf_add_rtk_pseudorange
//   SppDopplerFactor(...)                       R/factor/gnss_factor.h:108-131        -> swf_add_doppler
//   SppPseudorangeFactor(...)                   R/factor/gnss_factor.h:70-83          -> swf_add_spp_pseudorange
//   SppCarrierPhaseFactor(...)                  R/factor/gnss_factor.h:88-104         -> swf_add_spp_carrier_phase
//   FixedIntegerFactor(N21, istd)               R/factor/gnss_factor.h:135-143        -> swf_add_fixed_integer
//   ProjectionTwoFrameOneCamFactor(pts_i, pts_j) / TwoFrameTwoCam / OneFrameTwoCam
//                                               R/factor/projection_factor.h:33-66    -> swf_add_projection_inverse_depth
//   IMUGNSSFactor(IMUGNSSBase*)                 R/factor/gnss_imu_factor.h:19-151     -> swf_add_imu_gnss (IMUGNSSInfo below)
//   InitialBlackFactor(istd)                    R/factor/initial_factor.h:42-48       -> swf_add_scalar_prior
//   MarginalizationFactor(info)                 R/factor/marginalization_factor.h:104-110 -> swf_add_linear_prior (from a
//                                                                                        MarginalizationInfo* or from J, r0, x0)
//   ceres::internal::{parameter_head,is_optimize,lhs_out,rhs_out,lhs_out2,hs_row}
//                                               R/swf/swf_gnss.cpp:25-94              -> swf_ceres::internal::* below
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "swf_solver.h"

namespace swf_ceres {

enum LinearSolverType { DENSE_SCHUR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };      // ceres' default is LEVENBERG_MARQUARDT; R/swf/swf.cpp:26 sets DOGLEG

class LossFunction { public: virtual ~LossFunction() {} virtual double a() const = 0; };
class CauchyLoss : public LossFunction { public: explicit CauchyLoss(double a) : a_(a) {} double a() const override { return a_; } private: double a_; };
class LocalParameterization { public: virtual ~LocalParameterization() {} };

// ---- typed cost functions (data carriers) ------------------------------------------------
struct CostFunction { virtual ~CostFunction() {} };
struct projection_factor : CostFunction {
    double uv[2];
    static double& sqrt_info() { static double v = 1000.0 / 1.5; return v; }   // R/swf/swf.cpp:47
    template <class V3> explicit projection_factor(const V3& pts) { uv[0] = pts[0]; uv[1] = pts[1]; }
};
struct IMUFactor : CostFunction {
    std::vector<double> pre;   // SWF_PRE_DOUBLES record, see include/swf_types.h
    explicit IMUFactor(const double* record) : pre(record, record + SWF_PRE_DOUBLES) {}
    // The reference's own constructor, IMUFactor(IntegrationBase* _pre_integration) (R/factor/imu_factor.h:11): any type with the
    // member names of R/factor/integration_base.h:28-47 (Eigen or not: only operator()(i), operator()(i, j) and x() y() z() w() are
    // used) is copied into the record — delta_p / delta_q / delta_v, the linearisation biases, the five bias blocks of `jacobian`
    // (O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12, R/factor/integration_base.h and utility.h), sum_dt, gyri / gyrj, and
    // get_sqrtinfo() (what IMUFactor::Evaluate whitens with, R/factor/imu_factor.cpp:26).  A snapshot: re-propagate -> a new factor,
    // as the reference does (it deletes and re-adds its IMU factors with every window, R/swf/swf_image.cpp:198-251).
    template <class IB, class = decltype(std::declval<IB&>().delta_p(0)), class = decltype(std::declval<IB&>().get_sqrtinfo())>
    explicit IMUFactor(IB* ib) : pre(SWF_PRE_DOUBLES, 0.0) {
        for (int k = 0; k < 3; k++) {
            pre[SWF_PRE_DP + k] = ib->delta_p(k); pre[SWF_PRE_DV + k] = ib->delta_v(k);
            pre[SWF_PRE_LBA + k] = ib->linearized_ba(k); pre[SWF_PRE_LBG + k] = ib->linearized_bg(k);
            pre[SWF_PRE_GYRI + k] = ib->gyri(k); pre[SWF_PRE_GYRJ + k] = ib->gyrj(k);
        }
        pre[SWF_PRE_DQ + 0] = ib->delta_q.x(); pre[SWF_PRE_DQ + 1] = ib->delta_q.y(); pre[SWF_PRE_DQ + 2] = ib->delta_q.z(); pre[SWF_PRE_DQ + 3] = ib->delta_q.w();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            pre[SWF_PRE_DP_DBA + 3 * i + j] = ib->jacobian(0 + i, 9 + j); pre[SWF_PRE_DP_DBG + 3 * i + j] = ib->jacobian(0 + i, 12 + j);
            pre[SWF_PRE_DQ_DBG + 3 * i + j] = ib->jacobian(3 + i, 12 + j);
            pre[SWF_PRE_DV_DBA + 3 * i + j] = ib->jacobian(6 + i, 9 + j); pre[SWF_PRE_DV_DBG + 3 * i + j] = ib->jacobian(6 + i, 12 + j);
        }
        pre[SWF_PRE_SUMDT] = ib->sum_dt;
        const auto si = ib->get_sqrtinfo();
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) pre[SWF_PRE_SQRTINFO + 15 * i + j] = si(i, j);
    }
};
struct RTKCarrierPhaseFactor : CostFunction {
    double dat[SWF_CP_DOUBLES];
    RTKCarrierPhaseFactor(const double* sat, double L1_lam, double lam, double el, double br_time_diff, double mea_var,
                          const double* /*base_pos: window constant*/, bool use_istd, int /*sys*/, int /*f*/) {
        dat[0] = sat[0]; dat[1] = sat[1]; dat[2] = sat[2]; dat[3] = L1_lam; dat[4] = lam; dat[5] = el;
        dat[6] = br_time_diff; dat[7] = mea_var; dat[8] = use_istd ? 1.0 : 0.0;
    }
};
struct RTKPseudorangeFactor : CostFunction {
    double dat[SWF_PR_DOUBLES];
    RTKPseudorangeFactor(const double* sat, double P1, double el, double br_time_diff, double mea_var, const double* /*base_pos*/) {
        dat[0] = sat[0]; dat[1] = sat[1]; dat[2] = sat[2]; dat[3] = P1; dat[4] = el; dat[5] = br_time_diff; dat[6] = mea_var;
    }
};
struct SppDopplerFactor : CostFunction {
    double dat[SWF_DOP_DOUBLES];
    SppDopplerFactor(const double* satv, const double* sat, const double* /*xyzt*/, double D1_lam, double istd, const double* /*base_pos*/) {
        for (int k = 0; k < 3; k++) { dat[k] = sat[k]; dat[3 + k] = satv[k]; }
        dat[6] = D1_lam; dat[7] = istd;
    }
};
struct SppPseudorangeFactor : CostFunction {
    double dat[SWF_SPR_DOUBLES];
    SppPseudorangeFactor(const double* sat, double P1, double istd, const double* /*base_pos*/) {
        dat[0] = sat[0]; dat[1] = sat[1]; dat[2] = sat[2]; dat[3] = P1; dat[4] = istd;
    }
};
struct SppCarrierPhaseFactor : CostFunction {
    double dat[SWF_SCP_DOUBLES];
    SppCarrierPhaseFactor(const double* sat, double L1_lam, double istd, const double* /*base_pos*/, double lam) {
        dat[0] = sat[0]; dat[1] = sat[1]; dat[2] = sat[2]; dat[3] = L1_lam; dat[4] = istd; dat[5] = lam;
    }
};
struct FixedIntegerFactor : CostFunction { double N21, istd; FixedIntegerFactor(double N21_, double istd_) : N21(N21_), istd(istd_) {} };
// What IMUGNSSBase holds when SetLastImuFactor adds its factor (R/factor/gnss_imu_factor.cpp:99-119), as plain arrays: a maintainer
// fills it from gnss_poses / gnss_speed_bias (the hidden epochs' parameter memory, updated in place by every Solve), their
// *_lin points, pose_hessians, pose_phase_biases_hessians, pose_rhses, phase_biases_hessians, phase_biases_rhs and the M + 1
// pre-integrations (imu_factors[k]->pre_integration, last_imu_factor) in SWF_PRE_DOUBLES records.
// (the static sqrt_info of each class lives in a class template's static member: one definition across translation units without
// C++17 inline variables — the reference builds with -std=c++14, rtk_visual_inertial/CMakeLists.txt:5)
template <class Tag> struct SqrtInfoOf { static double sqrt_info; };
template <class Tag> double SqrtInfoOf<Tag>::sqrt_info = 0;
struct ProjectionTwoFrameOneCamFactor : CostFunction, SqrtInfoOf<ProjectionTwoFrameOneCamFactor> {
    double pi[3], pj[3];
    ProjectionTwoFrameOneCamFactor(const double* pts_i, const double* pts_j) { for (int k = 0; k < 3; k++) { pi[k] = pts_i[k]; pj[k] = pts_j[k]; } }
};
struct ProjectionTwoFrameTwoCamFactor : CostFunction, SqrtInfoOf<ProjectionTwoFrameTwoCamFactor> {
    double pi[3], pj[3];
    ProjectionTwoFrameTwoCamFactor(const double* pts_i, const double* pts_j) { for (int k = 0; k < 3; k++) { pi[k] = pts_i[k]; pj[k] = pts_j[k]; } }
};
struct ProjectionOneFrameTwoCamFactor : CostFunction, SqrtInfoOf<ProjectionOneFrameTwoCamFactor> {
    double pi[3], pj[3];
    ProjectionOneFrameTwoCamFactor(const double* pts_i, const double* pts_j) { for (int k = 0; k < 3; k++) { pi[k] = pts_i[k]; pj[k] = pts_j[k]; } }
};
struct IMUGNSSInfo {
    int M = 0;                                     // hidden GNSS epochs
    double* hidden_pose = nullptr; double* hidden_sb = nullptr;        // [M][7], [M][9]
    std::vector<double> pose_lin, sb_lin, Hpp, HpN, rhs_p, HNN, rhsN, pre;
    int mid = 0; std::vector<double> H12;          // AddMidMargInfo's product (gnss_Index, pose1_pose2_hessians): link mid in 1..M-1, 15 x 15; 0 = none
    // filled by IMUGNSSFactor(IMUGNSSBase*) only: the reference keeps every hidden epoch in its own allocation (gnss_poses[k],
    // gnss_speed_bias[k]); the C-ABI wants [M][7] / [M][9], so the adapter owns contiguous copies (hidden_pose / hidden_sb point at
    // them), Solve() gathers them from the epochs' memory before the solve and scatters the updated values back after it
    std::vector<double*> pose_ptr, sb_ptr;
    std::vector<double> pose_buf, sb_buf;
    void gather() { for (size_t k = 0; k < pose_ptr.size(); k++) { std::memcpy(&pose_buf[7 * k], pose_ptr[k], 7 * sizeof(double)); std::memcpy(&sb_buf[9 * k], sb_ptr[k], 9 * sizeof(double)); } }
    void scatter() const { for (size_t k = 0; k < pose_ptr.size(); k++) { std::memcpy(pose_ptr[k], &pose_buf[7 * k], 7 * sizeof(double)); std::memcpy(sb_ptr[k], &sb_buf[9 * k], 9 * sizeof(double)); } }
};
struct IMUGNSSFactor : CostFunction {
    IMUGNSSInfo* info; std::unique_ptr<IMUGNSSInfo> owned;
    explicit IMUGNSSFactor(IMUGNSSInfo* i) : info(i) {}
    // The reference's own constructor, IMUGNSSFactor(IMUGNSSBase* IMUGNSS_info_) (R/factor/gnss_imu_factor.h:145-151): any type with the
    // member names of IMUGNSSBase (R/factor/gnss_imu_factor.h:19-140) as SetLastImuFactor leaves them (R/factor/gnss_imu_factor.cpp:99-119):
    // gnss_poses / gnss_speed_bias (the M hidden epochs' parameter memory) and their *_lin points, pose_hessians[k] (15 x 15),
    // pose_phase_biases_hessians[k] (15 x N), pose_rhses[k], phase_biases_hessians (N x N), phase_biases_rhs, the M pre-integrations
    // imu_factors[k]->pre_integration (epoch k-1 -> k; k = 0 starts at para_pose0) and last_imu_factor->pre_integration (epoch M-1 -> the
    // second visual frame), and — after AddMidMargInfo (:121-240) — gnss_middle_marginfo != 0, gnss_Index, pose1_pose2_hessians.
    // Only operator()(i), operator()(i, j) and size() of the matrix members are used (Eigen or not).  A snapshot, like the other
    // factors' constructors: the reference builds a new IMUGNSSFactor whenever it re-adds the residual block (SetLastImuFactor).
    template <class GB, class = decltype(std::declval<GB&>().gnss_poses), class = decltype(std::declval<GB&>().pose_phase_biases_hessians),
              class = decltype(std::declval<GB&>().last_imu_factor)>
    explicit IMUGNSSFactor(GB* b) : info(nullptr), owned(new IMUGNSSInfo()) {
        IMUGNSSInfo& I = *owned; info = owned.get();
        const int M = (int)b->gnss_poses.size(), N = (int)b->gnss_phase_biases.size();
        if (M < 1 || (int)b->gnss_speed_bias.size() != M || (int)b->imu_factors.size() != M || !b->last_imu_factor || (int)b->pose_hessians.size() != M)
            throw std::invalid_argument("IMUGNSSFactor: IMUGNSSBase holds no complete chain of hidden epochs");
        I.M = M;
        I.pose_ptr.assign(b->gnss_poses.begin(), b->gnss_poses.end()); I.sb_ptr.assign(b->gnss_speed_bias.begin(), b->gnss_speed_bias.end());
        I.pose_buf.resize((size_t)7 * M); I.sb_buf.resize((size_t)9 * M); I.gather();
        I.hidden_pose = I.pose_buf.data(); I.hidden_sb = I.sb_buf.data();
        I.pose_lin.resize((size_t)7 * M); I.sb_lin.resize((size_t)9 * M);
        I.Hpp.resize((size_t)225 * M); I.HpN.assign((size_t)15 * N * M, 0.0); I.rhs_p.resize((size_t)15 * M);
        I.HNN.resize((size_t)N * N); I.rhsN.resize((size_t)N); I.pre.resize((size_t)SWF_PRE_DOUBLES * (M + 1));
        for (int k = 0; k < M; k++) {
            std::memcpy(&I.pose_lin[7 * k], b->gnss_poses_lin[k], 7 * sizeof(double));
            std::memcpy(&I.sb_lin[9 * k], b->gnss_speed_bias_lin[k], 9 * sizeof(double));
            for (int i = 0; i < 15; i++) {
                for (int j = 0; j < 15; j++) I.Hpp[(size_t)225 * k + 15 * i + j] = b->pose_hessians[k](i, j);
                for (int j = 0; j < N; j++) I.HpN[((size_t)15 * k + i) * N + j] = b->pose_phase_biases_hessians[k](i, j);
                I.rhs_p[(size_t)15 * k + i] = b->pose_rhses[k](i);
            }
            const IMUFactor rec(b->imu_factors[k]->pre_integration);
            std::memcpy(&I.pre[(size_t)SWF_PRE_DOUBLES * k], rec.pre.data(), SWF_PRE_DOUBLES * sizeof(double));
        }
        const IMUFactor last(b->last_imu_factor->pre_integration);
        std::memcpy(&I.pre[(size_t)SWF_PRE_DOUBLES * M], last.pre.data(), SWF_PRE_DOUBLES * sizeof(double));
        for (int i = 0; i < N; i++) { I.rhsN[i] = b->phase_biases_rhs(i); for (int j = 0; j < N; j++) I.HNN[(size_t)i * N + j] = b->phase_biases_hessians(i, j); }
        if (b->gnss_middle_marginfo) {
            I.mid = b->gnss_Index; I.H12.resize(225);
            for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) I.H12[15 * i + j] = b->pose1_pose2_hessians(i, j);
        }
    }
};
struct InitialBlackFactor : CostFunction { double istd; explicit InitialBlackFactor(double w) : istd(w) {} };
// MarginalizationInfo's product: the linearised prior (J, r0, x0) over its kept blocks
struct MarginalizationFactor : CostFunction {
    std::vector<double> J, r0, x0;
    MarginalizationFactor(const double* J_, const double* r0_, const double* x0_, int dim, int global_sum)
        : J(J_, J_ + (size_t)dim * dim), r0(r0_, r0_ + dim), x0(x0_, x0_ + global_sum) {}
    // The reference's own constructor, MarginalizationFactor(MarginalizationInfo*) (R/factor/marginalization_factor.h:106,
    // .cpp:401-408): any type with its members n, m, keep_block_size (global sizes), keep_block_idx (position of the block's local
    // dimensions among the prior's columns, offset by m), keep_block_data (the linearisation point per block), linearized_jacobians
    // (n x n) and linearized_residuals (n), as MarginalizationFactor::Evaluate reads them (R/factor/marginalization_factor.cpp:410-446).
    // The columns are re-filed in the order of the blocks — the order of the parameter blocks handed to AddResidualBlock, which is the
    // order swf_add_linear_prior expects — so a keep_block_idx that does not ascend with the blocks is handled here.
    template <class MI, class = decltype(std::declval<MI&>().keep_block_idx), class = decltype(std::declval<MI&>().linearized_jacobians(0, 0))>
    explicit MarginalizationFactor(MI* mi) {
        const int n = mi->n, m = mi->m, nb = (int)mi->keep_block_size.size();
        J.assign((size_t)n * n, 0.0); r0.resize(n);
        for (int k = 0; k < n; k++) r0[k] = mi->linearized_residuals(k);
        int col = 0;
        for (int b = 0; b < nb; b++) {
            const int gs = mi->keep_block_size[b], ls = gs == 7 ? 6 : gs, idx = mi->keep_block_idx[b] - m;
            for (int r = 0; r < n; r++) for (int c = 0; c < ls; c++) J[(size_t)r * n + col + c] = mi->linearized_jacobians(r, idx + c);
            x0.insert(x0.end(), mi->keep_block_data[b], mi->keep_block_data[b] + gs);
            col += ls;
        }
        if (col != n) throw std::invalid_argument("MarginalizationFactor: the kept blocks' local sizes do not add up to n");
    }
};

class PoseLocalParameterization : public LocalParameterization {};   // R/factor/pose_local_parameterization.h

class ParameterBlockOrdering {
  public:
    void Clear() { keys_.clear(); groups_.clear(); }
    void AddElementToGroup(double* p, int group) { keys_.push_back(p); groups_.push_back(group); }
    int NumElements() const { return (int)keys_.size(); }
    std::vector<double*> keys_; std::vector<int32_t> groups_;
};

class Problem;
namespace internal {
// the modified Ceres' private globals under their own names, as the reference reads and writes them (R/swf/swf_gnss.cpp:20-94:
// the bodies of UpdateSchur / UpdateSchurHessianOnly compile unchanged against these): one object per program without C++17
// inline variables (the reference builds with -std=c++14) — the objects are static members of a class template, the names are
// namespace-scope references bound to them at static-initialisation time; no macros leak into the including translation unit.
// lhs_out (hs_row x hs_row, full symmetric, row-major) / rhs_out / lhs_out2 (the lower factor, row-major) /
// hs_row belong to the LAST successful Solve; a failed Solve and the destruction of the Problem they point into reset them to null.
// (Non-const pointers because the reference maps them with Eigen::Map<VectorXd> / Map<ceres::Matrix>; the memory is the solver's:
// read only.)
template <class = void> struct Globals {
    static std::vector<double*> parameter_head; static bool is_optimize;
    static double* lhs_out; static double* rhs_out; static double* lhs_out2; static int hs_row; static const Problem* owner;
};
template <class T> std::vector<double*> Globals<T>::parameter_head;
template <class T> bool Globals<T>::is_optimize = true;
template <class T> double* Globals<T>::lhs_out = nullptr;
template <class T> double* Globals<T>::rhs_out = nullptr;
template <class T> double* Globals<T>::lhs_out2 = nullptr;
template <class T> int Globals<T>::hs_row = 0;
template <class T> const Problem* Globals<T>::owner = nullptr;
#if defined(__GNUC__)
#define SWF_CERES_UNUSED __attribute__((unused))
#else
#define SWF_CERES_UNUSED
#endif
static std::vector<double*>& parameter_head SWF_CERES_UNUSED = Globals<>::parameter_head;
static bool& is_optimize SWF_CERES_UNUSED = Globals<>::is_optimize;
static double*& lhs_out SWF_CERES_UNUSED = Globals<>::lhs_out;
static double*& rhs_out SWF_CERES_UNUSED = Globals<>::rhs_out;
static double*& lhs_out2 SWF_CERES_UNUSED = Globals<>::lhs_out2;
static int& hs_row SWF_CERES_UNUSED = Globals<>::hs_row;
#undef SWF_CERES_UNUSED
inline void reset_exports() { Globals<>::lhs_out = Globals<>::rhs_out = Globals<>::lhs_out2 = nullptr; Globals<>::hs_row = 0; Globals<>::owner = nullptr; }
// the same four values as one struct (the accessor of rounds 1-4, kept for callers written against it)
struct Exports { const double* lhs_out = nullptr; const double* rhs_out = nullptr; const double* lhs_out2 = nullptr; int hs_row = 0; const Problem* owner = nullptr; };
inline Exports exports() { Exports e; e.lhs_out = Globals<>::lhs_out; e.rhs_out = Globals<>::rhs_out; e.lhs_out2 = Globals<>::lhs_out2; e.hs_row = Globals<>::hs_row; e.owner = Globals<>::owner; return e; }
// ceres::internal::ResidualBlock as the estimator sees it: a handle whose public is_use flag it flips directly
// (R/swf/swf_image.cpp:353-365,422; R/swf/swf_gnss.cpp:653).  Solve() carries the flags over to swf_factor_set_enabled.
struct ResidualBlock { bool is_use = true; swf_factor_id id = -1; };
}  // namespace internal

typedef internal::ResidualBlock* ResidualBlockId;

class Problem {
  public:
    Problem() { if (swf_problem_create(&h_) != SWF_OK) throw std::runtime_error("swf_problem_create"); }
    ~Problem() {
        if (internal::Globals<>::owner == this) internal::reset_exports();
        swf_problem_destroy(h_);
    }
    Problem(const Problem&) = delete;
    swf_problem* handle() { return h_; }

    void AddParameterBlock(double* p, int size, LocalParameterization* lp = nullptr) {
        chk(swf_add_parameter_block(h_, p, size, lp ? SWF_MANIFOLD_POSE : SWF_MANIFOLD_NONE), "AddParameterBlock");
        delete lp;                       // Problem takes ownership, as in ceres
    }
    bool HasParameterBlock(const double* p) const { return swf_has_parameter_block(h_, p) != 0; }
    void RemoveParameterBlock(double* p) { chk(swf_remove_parameter_block(h_, p), "RemoveParameterBlock"); }
    void SetParameterBlockConstant(double* p) { chk(swf_set_parameter_block_constant(h_, p), "SetParameterBlockConstant"); }
    void SetParameterBlockVariable(double* p) { chk(swf_set_parameter_block_variable(h_, p), "SetParameterBlockVariable"); }
    bool IsParameterBlockConstant(const double* p) const { return swf_is_parameter_block_constant(h_, p) != 0; }
    int ParameterBlockSize(const double* p) const { return swf_parameter_block_size(h_, p); }
    int NumParameterBlocks() const { return swf_num_parameter_blocks(h_); }
    int NumResidualBlocks() const { return swf_num_residual_blocks(h_); }
    void RemoveResidualBlock(ResidualBlockId rb) { chk(swf_remove_factor(h_, rb->id), "RemoveResidualBlock"); hidden_.erase(rb->id); }
    // the query surface GlobalMarge / FeatureManager walk (R/swf/swf_image.cpp:350-367, R/swf/swf.cpp:413-422)
    void GetResidualBlocks(std::vector<ResidualBlockId>* out) const {
        int32_t n = 0; chk(swf_get_residual_blocks(h_, nullptr, 0, &n), "GetResidualBlocks");
        std::vector<swf_factor_id> ids((size_t)n); chk(swf_get_residual_blocks(h_, ids.data(), n, &n), "GetResidualBlocks");
        out->clear(); for (swf_factor_id i : ids) out->push_back(handle_of(i));
    }
    void GetResidualBlocksForParameterBlock(const double* p, std::vector<ResidualBlockId>* out) const {
        int32_t n = 0; chk(swf_get_residual_blocks_for_parameter_block(h_, p, nullptr, 0, &n), "GetResidualBlocksForParameterBlock");
        std::vector<swf_factor_id> ids((size_t)n); chk(swf_get_residual_blocks_for_parameter_block(h_, p, ids.data(), n, &n), "GetResidualBlocksForParameterBlock");
        out->clear(); for (swf_factor_id i : ids) out->push_back(handle_of(i));
    }
    void GetParameterBlocks(std::vector<double*>* out) const {
        int32_t n = 0; chk(swf_get_parameter_blocks(h_, nullptr, 0, &n), "GetParameterBlocks");
        out->assign((size_t)n, nullptr); chk(swf_get_parameter_blocks(h_, out->data(), n, &n), "GetParameterBlocks");
    }
    void GetParameterBlocksForResidualBlock(const ResidualBlockId rb, std::vector<double*>* out) const {
        int32_t n = 0; chk(swf_get_parameter_blocks_for_residual_block(h_, rb->id, nullptr, 0, &n), "GetParameterBlocksForResidualBlock");
        out->assign((size_t)n, nullptr); chk(swf_get_parameter_blocks_for_residual_block(h_, rb->id, out->data(), n, &n), "GetParameterBlocksForResidualBlock");
    }
    // ResidualBlock::is_use -> swf_factor_set_enabled for every live residual block (called by Solve)
    int SyncIsUse() {
        int32_t n = 0; int rc = swf_get_residual_blocks(h_, nullptr, 0, &n);
        if (rc != SWF_OK) return rc;
        std::vector<swf_factor_id> ids((size_t)n);
        if ((rc = swf_get_residual_blocks(h_, ids.data(), n, &n)) != SWF_OK) return rc;
        for (swf_factor_id i : ids) if ((rc = swf_factor_set_enabled(h_, i, handle_of(i)->is_use ? 1 : 0)) != SWF_OK) return rc;
        return SWF_OK;
    }
    // hidden epochs of composite factors built from an IMUGNSSBase: epochs' own memory <-> the contiguous copies the solver works on
    void GatherHiddenEpochs() { for (auto& kv : hidden_) kv.second->gather(); }
    void ScatterHiddenEpochs() const { for (const auto& kv : hidden_) kv.second->scatter(); }
    void SetConstants(const double* pbg, const double* gw, const double* base) { chk(swf_set_constants(h_, pbg, gw, base), "SetConstants"); }

    // AddResidualBlock overloads by cost-function type; Problem takes ownership of cost and loss
    ResidualBlockId AddResidualBlock(projection_factor* f, LossFunction* loss, double* pose, double* ex, double* pt) {
        swf_factor_id id = swf_add_projection(h_, pose, ex, pt, f->uv, projection_factor::sqrt_info(), loss ? loss->a() : 0.0);
        delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(IMUFactor* f, LossFunction* loss, double* pi, double* si, double* pj, double* sj) {
        swf_factor_id id = swf_add_imu(h_, pi, si, pj, sj, f->pre.data()); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(RTKCarrierPhaseFactor* f, LossFunction* loss, double* pose, double* amb, double* clk) {
        swf_factor_id id = swf_add_rtk_carrier_phase(h_, pose, amb, clk, f->dat); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(RTKPseudorangeFactor* f, LossFunction* loss, double* pose, double* clk) {
        swf_factor_id id = swf_add_rtk_pseudorange(h_, pose, clk, f->dat); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(SppDopplerFactor* f, LossFunction* loss, double* sb, double* drift, double* pose) {
        swf_factor_id id = swf_add_doppler(h_, sb, drift, pose, f->dat); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(SppPseudorangeFactor* f, LossFunction* loss, double* pose, double* clk) {
        swf_factor_id id = swf_add_spp_pseudorange(h_, pose, clk, f->dat); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(SppCarrierPhaseFactor* f, LossFunction* loss, double* pose, double* clk, double* amb) {
        swf_factor_id id = swf_add_spp_carrier_phase(h_, pose, clk, amb, f->dat); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(FixedIntegerFactor* f, LossFunction* loss, double* n_a, double* n_b) {
        swf_factor_id id = swf_add_fixed_integer(h_, n_a, n_b, f->N21, f->istd); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(ProjectionTwoFrameOneCamFactor* f, LossFunction* loss, double* pose_i, double* pose_j, double* ex, double* inv_depth) {
        swf_factor_id id = swf_add_projection_inverse_depth(h_, 0, pose_i, pose_j, ex, nullptr, inv_depth, f->pi, f->pj, ProjectionTwoFrameOneCamFactor::sqrt_info, loss ? loss->a() : 0.0);
        delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(ProjectionTwoFrameTwoCamFactor* f, LossFunction* loss, double* pose_i, double* pose_j, double* ex, double* ex2, double* inv_depth) {
        swf_factor_id id = swf_add_projection_inverse_depth(h_, 1, pose_i, pose_j, ex, ex2, inv_depth, f->pi, f->pj, ProjectionTwoFrameTwoCamFactor::sqrt_info, loss ? loss->a() : 0.0);
        delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(ProjectionOneFrameTwoCamFactor* f, LossFunction* loss, double* ex, double* ex2, double* inv_depth) {
        swf_factor_id id = swf_add_projection_inverse_depth(h_, 2, nullptr, nullptr, ex, ex2, inv_depth, f->pi, f->pj, ProjectionOneFrameTwoCamFactor::sqrt_info, loss ? loss->a() : 0.0);
        delete f; delete loss; return ck(id);
    }
    // param = { pose_i, speed_bias_i, pose_j, speed_bias_j, ambiguity_0 .. ambiguity_N-1 } as SetLastImuFactor builds it
    ResidualBlockId AddResidualBlock(IMUGNSSFactor* f, LossFunction* loss, const std::vector<double*>& param) {
        const IMUGNSSInfo& I = *f->info;
        const int N = (int)param.size() - 4;
        swf_factor_id id = swf_add_imu_gnss(h_, param[0], param[1], param[2], param[3], param.data() + 4, N, I.M, I.hidden_pose, I.hidden_sb,
                                              I.pose_lin.data(), I.sb_lin.data(), I.Hpp.data(), I.HpN.data(), I.rhs_p.data(), I.HNN.data(), I.rhsN.data(), I.pre.data());
        if (id >= 0 && I.mid > 0 && swf_set_imu_gnss_mid_link(h_, id, I.mid, I.H12.data()) != SWF_OK) throw std::runtime_error(swf_last_error());
        if (id >= 0 && f->owned) hidden_[id] = std::move(f->owned);      // built from an IMUGNSSBase: the Problem keeps the contiguous hidden epochs alive
        delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(InitialBlackFactor* f, LossFunction* loss, double* scalar) {
        swf_factor_id id = swf_add_scalar_prior(h_, scalar, f->istd); delete f; delete loss; return ck(id);
    }
    ResidualBlockId AddResidualBlock(MarginalizationFactor* f, LossFunction* loss, const std::vector<double*>& blocks) {
        swf_factor_id id = swf_add_linear_prior(h_, blocks.data(), (int32_t)blocks.size(), f->J.data(), f->r0.data(), f->x0.data());
        delete f; delete loss; return ck(id);
    }

  private:
    static void chk(int rc, const char* what) { if (rc != SWF_OK) throw std::runtime_error(std::string(what) + ": " + swf_last_error()); }
    // factor id -> the handle the caller holds; handles live as long as the Problem (a removed block's handle dangles
    // logically, as in ceres, but never physically)
    ResidualBlockId ck(swf_factor_id id) {
        if (id < 0) throw std::runtime_error(std::string("AddResidualBlock: ") + swf_last_error());
        if (rb_.size() <= (size_t)id) rb_.resize((size_t)id + 1);
        rb_[(size_t)id].reset(new internal::ResidualBlock());
        rb_[(size_t)id]->id = id;
        return rb_[(size_t)id].get();
    }
    // the handle of a live factor id; created on demand for factors that were added through handle() and the C API
    // (is_use = true, like a freshly added block)
    ResidualBlockId handle_of(swf_factor_id id) const {
        if (id < 0) throw std::runtime_error("residual block id out of range");
        if (rb_.size() <= (size_t)id) rb_.resize((size_t)id + 1);
        if (!rb_[(size_t)id]) { rb_[(size_t)id].reset(new internal::ResidualBlock()); rb_[(size_t)id]->id = id; }
        return rb_[(size_t)id].get();
    }
    swf_problem* h_ = nullptr;
    mutable std::vector<std::unique_ptr<internal::ResidualBlock>> rb_;
    std::map<swf_factor_id, std::unique_ptr<IMUGNSSInfo>> hidden_;
};

struct Solver {
    struct Options {
        LinearSolverType linear_solver_type = DENSE_SCHUR;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;      // ceres' default; the window solves set DOGLEG
        int max_num_iterations = 50;         // ceres' default; R/swf/swf.cpp:25 sets MAX_NUM_ITERATIONS
        int num_threads = 1;                 // accepted, unused: the device decides its own parallelism
        bool jacobi_scaling = true;          // ceres' default, in force for the reference's default-options solves (R/swf/swf_gnss.cpp:205-214, 562-572:
                                             // Levenberg-Marquardt, where it is supported); every Options block that selects DOGLEG sets it to 0 (R/swf/swf.cpp:26-27)
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16;
        std::shared_ptr<ParameterBlockOrdering> linear_solver_ordering;      // null: automatic ordering (swf_set_ordering, n = 0)
        int swf_composite_root = SWF_ROOT_PIVOTED_CHOLESKY;                  // extension (not in ceres): swf_options::composite_root; SWF_ROOT_EIGEN = UpdateSchurComponent's own root
    };
    struct Summary {
        double initial_cost = 0, final_cost = 0, minimizer_time_in_seconds = 0;
        int num_successful_steps = 0, num_unsuccessful_steps = 0, termination = 0;
        std::string message;                 // swf_last_error() of a failed Solve, empty otherwise
        swf_summary raw;
        std::string BriefReport() const {
            char b[256];
            snprintf(b, sizeof b, "swf (MI355X) Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %d",
                     raw.num_iterations, initial_cost, final_cost, termination);
            return message.empty() ? std::string(b) : std::string(b) + " [" + message + "]";
        }
    };
};

// ceres::Solve(options, &problem, &summary) — R/swf/swf_image.cpp:219.  Ceres reports failures through the summary, never by
// throwing, and the estimator only tests summary.final_cost > 1e10: a failed solve therefore sets final_cost = 1e300, keeps the
// reason in summary.message, prints it to stderr and clears the exports (UpdateSchur must not read a previous solve's buffers).
inline void Solve(const Solver::Options& o, Problem* p, Solver::Summary* s) {
    swf_options opt; swf_options_default(&opt);
    opt.max_num_iterations = o.max_num_iterations;
    opt.initial_trust_region_radius = o.initial_trust_region_radius;
    opt.max_trust_region_radius = o.max_trust_region_radius;
    opt.step_mode = internal::is_optimize ? SWF_OPTIMIZE : SWF_ASSEMBLE_ELIMINATE_ONLY;
    opt.trust_region_strategy = o.trust_region_strategy_type == DOGLEG ? SWF_DOGLEG : SWF_LEVENBERG_MARQUARDT;
    std::memset(&s->raw, 0, sizeof(s->raw));
    s->message.clear();
    int rc = SWF_OK;
    opt.composite_root = o.swf_composite_root;
    opt.jacobi_scaling = o.jacobi_scaling ? 1 : 0;       // (with DOGLEG the engine refuses it: SWF_E_UNSUPPORTED, reported below like any failure)
    if (rc == SWF_OK) rc = p->SyncIsUse();
    if (rc == SWF_OK) {
        const ParameterBlockOrdering* ord = o.linear_solver_ordering.get();
        rc = swf_set_ordering(p->handle(), ord ? ord->keys_.data() : nullptr, ord ? ord->groups_.data() : nullptr, ord ? ord->NumElements() : 0);
    }
    if (rc == SWF_OK) rc = swf_set_export_tail(p->handle(), internal::parameter_head.data(), (int32_t)internal::parameter_head.size());
    if (rc == SWF_OK) {
        p->GatherHiddenEpochs();
        rc = swf_problem_solve(p->handle(), &opt, &s->raw);
        if (rc == SWF_OK) p->ScatterHiddenEpochs();
    }
    internal::reset_exports();
    if (rc == SWF_OK) {
        const double *lo = nullptr, *ro = nullptr, *lo2 = nullptr; int32_t hr = 0;
        if (swf_get_reduced(p->handle(), &lo, &ro, &lo2, &hr) == SWF_OK) {
            internal::Globals<>::lhs_out = const_cast<double*>(lo); internal::Globals<>::rhs_out = const_cast<double*>(ro);
            internal::Globals<>::lhs_out2 = const_cast<double*>(lo2); internal::Globals<>::hs_row = hr; internal::Globals<>::owner = p;
        }
    } else {
        if (s->message.empty()) s->message = swf_last_error();
        std::fprintf(stderr, "swf_ceres::Solve failed (%d): %s\n", rc, s->message.c_str());
    }
    s->initial_cost = s->raw.initial_cost; s->final_cost = rc == SWF_OK ? s->raw.final_cost : 1e300;   // callers test final_cost > 1e10
    s->minimizer_time_in_seconds = s->raw.minimizer_time_in_seconds;
    s->num_successful_steps = s->raw.num_successful_steps; s->num_unsuccessful_steps = s->raw.num_unsuccessful_steps;
    s->termination = s->raw.termination;
}

// The marginalisation consumer in one call: what SWFOptimization::UpdateSchur (R/swf/swf_gnss.cpp:25-61) followed by
// MarginalizationInfo::setmarginalizeinfo(addr, sizes, A, b, true) (R/factor/marginalization_factor.cpp:449-488)
// compute from the export of an is_optimize = false Solve.  Call it right after that Solve (GlobalMarge,
// R/swf/swf_image.cpp:404-418).  linearized_jacobians / linearized_residuals are n x n row-major / n, solver-owned,
// valid until the next Solve; A / b are the marginal system UpdateSchur leaves in SWFOptimization::A, b.
struct MarginalPrior {
    const double* linearized_jacobians = nullptr; const double* linearized_residuals = nullptr;
    const double* A = nullptr; const double* b = nullptr;
    int n = 0, rank = 0;
};
// eigen = true: the reference's eigen square root (tails up to 640 dimensions); false: the Cholesky square root (same quadratic, cheaper)
inline bool UpdateSchurAndSetMarginalizeInfo(Problem* p, MarginalPrior* out, bool eigen = true, double eps = 1e-8) {
    int32_t n = 0, rank = 0;
    int rc = swf_problem_marginalize(p->handle(), eps, eigen ? SWF_PRIOR_EIGEN : SWF_PRIOR_CHOLESKY, &out->linearized_jacobians,
                                     &out->linearized_residuals, &out->A, &out->b, &n, &rank);
    out->n = n; out->rank = rank;
    internal::parameter_head.clear();                       // as the reference's reader does (swf_gnss.cpp:58)
    return rc == SWF_OK && rank >= 0;
}

// MarginalizationInfo::ResetLinearizationPoint (R/factor/marginalization_factor.cpp:232-258; R/swf/swf_core.cpp:636-637): shift the prior
// (linearized_jacobians J, linearized_residuals r0, optionally the marginal system A, b) to the kept blocks' current values.  sizes = the
// kept blocks' global sizes (7 = pose) in kept order, parameters[i] = block i's current values, x0 = keep_block_data concatenated.
inline bool ResetLinearizationPoint(const std::vector<int32_t>& sizes, const std::vector<const double*>& parameters, int n,
                                    const double* J, const double* A, double* r0, double* b, double* x0) {
    return swf_prior_reset_linearization_point((int32_t)sizes.size(), sizes.data(), parameters.data(), n, J, A, r0, b, x0) == SWF_OK;
}

// SWFOptimization::UpdateSchurHessianOnly (R/swf/swf_gnss.cpp:65-94) after an optimising Solve: A = A3 A3^T over the
// parameter_head states, plus the covariance Qy = A^-1 LambdaSearch computes from it (R/swf/swf_lambda.cpp:94-99).
struct TailCovariance { const double* A = nullptr; const double* Qy = nullptr; int n = 0; };
inline bool UpdateSchurHessianOnly(Problem* p, TailCovariance* out) {
    int32_t n = 0;
    int rc = swf_problem_tail_covariance(p->handle(), &out->A, &out->Qy, &n);
    out->n = n;
    internal::parameter_head.clear();                       // swf_gnss.cpp:91
    return rc == SWF_OK && n > 0;
}

}  // namespace swf_ceres
