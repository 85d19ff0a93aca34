// A window built twice and solved on the GPU: once from the C-ABI's own records (IMUFactor(const double*), MarginalizationFactor(J, r0,
// x0, ...), IMUGNSSFactor(IMUGNSSInfo*)), once through the REFERENCE'S constructor signatures
//   IMUFactor(IntegrationBase*)                      R/factor/imu_factor.h:11
//   MarginalizationFactor(MarginalizationInfo*)      R/factor/marginalization_factor.h:106
//   IMUGNSSFactor(IMUGNSSBase*)                      R/factor/gnss_imu_factor.h:145-151
// from stand-ins with the reference's member names (every hidden GNSS epoch in its own allocation, as IMUGNSSBase::gnss_poses /
// gnss_speed_bias hold them).  Both must give the same solve bit for bit — costs, every parameter block, the hidden epochs written
// back into their own memory.  Then the raw globals of the modified Ceres, ceres::internal::lhs_out / rhs_out / lhs_out2 / hs_row,
// are read the way SWFOptimization::UpdateSchur / UpdateSchurHessianOnly read them (R/swf/swf_gnss.cpp:25-94), without Eigen.
// tests/test_host.py compiles it with -std=c++14 (the reference's standard); it RUNS on the GPU box only.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
#include "swf_ceres.hpp"
namespace ceres = swf_ceres;
using namespace swf_ceres;

namespace {
// ---- stand-ins: the reference's member names, Eigen's accessors ------------------------------------------------------------------
struct V3 { double v[3]; double operator()(int i) const { return v[i]; } };
struct Q4 { double x_, y_, z_, w_; double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; } };
struct M15 { double a[15][15]; double operator()(int i, int j) const { return a[i][j]; } };
struct MX {
    int rows = 0, cols = 0; std::vector<double> a;
    MX() {}
    MX(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
    double& at(int i, int j) { return a[(size_t)i * cols + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * cols + j]; }
    double operator()(int i) const { return a[i]; }
};
struct IntegrationBase {                 // R/factor/integration_base.h:28-47
    double dt, sum_dt;
    V3 acc_0, gyr_0, acc_1, gyr_1, linearized_ba, linearized_bg, delta_p, delta_v, gyri, gyrj;
    Q4 delta_q;
    M15 jacobian, covariance, sqrt_info;
    M15 get_sqrtinfo() { return sqrt_info; }
};
struct RefIMUFactor { IntegrationBase* pre_integration; };      // R/factor/imu_factor.h:17 (what IMUGNSSBase::imu_factors point at)
struct MarginalizationInfo {             // R/factor/marginalization_factor.h:42-100
    int m, n;
    std::vector<int> keep_block_size, keep_block_idx;
    std::vector<double*> keep_block_data;
    MX linearized_jacobians, linearized_residuals;
};
struct IMUGNSSBase {                     // R/factor/gnss_imu_factor.h:19-140 (the members SetLastImuFactor's factor reads)
    std::vector<double*> gnss_phase_biases, gnss_speed_bias, gnss_speed_bias_lin, gnss_poses, gnss_poses_lin;
    MX phase_biases_hessians, phase_biases_rhs;
    std::vector<M15> pose_hessians;
    M15 pose1_pose2_hessians;
    std::vector<MX> pose_phase_biases_hessians, pose_rhses;
    std::vector<RefIMUFactor*> imu_factors;
    RefIMUFactor* last_imu_factor = 0;
    MarginalizationInfo* gnss_middle_marginfo = 0;
    int gnss_Index = 0;
    std::vector<double*> param;
};

// ---- the toy window -------------------------------------------------------------------------------------------------------------
const int NF = 3, NL = 10, M = 2, N = 2;
const double T_IMU = 0.2, G = 9.81;
struct State {
    double pose[NF][7], sb[NF][9], ex[7], pt[NL][3], amb[N][1], black;
    double* hid_pose[M]; double* hid_sb[M];             // the hidden epochs: each in its own allocation
    double hid_pose_flat[M][7], hid_sb_flat[M][9];      // ... or contiguous (the C-ABI's own layout)
};
void init_state(State& s, bool separate) {
    for (int i = 0; i < NF; i++) {
        double p[7] = {0.9 * i + 0.02 * (i % 2), 0.03 * i, -0.02 * i, 0.01 * i, -0.008 * i, 0.005 * i, 0};
        p[6] = std::sqrt(1 - p[3] * p[3] - p[4] * p[4] - p[5] * p[5]);
        std::memcpy(s.pose[i], p, sizeof p);
        double b[9] = {4.4, 0.1, -0.05, 0.02, -0.01, 0.015, 0.001, -0.002, 0.0015};
        std::memcpy(s.sb[i], b, sizeof b);
    }
    double ex[7] = {0.02, -0.01, 0.03, 0, 0, 0, 1}; std::memcpy(s.ex, ex, sizeof ex);
    for (int l = 0; l < NL; l++) { s.pt[l][0] = 0.9 + 0.7 * std::sin(1.3 * l); s.pt[l][1] = 0.8 * std::cos(0.9 * l); s.pt[l][2] = 6 + 0.5 * l; }
    s.amb[0][0] = 0.3; s.amb[1][0] = -0.2; s.black = 0;
    for (int k = 0; k < M; k++) {
        double hp[7] = {0.9 + 0.3 * (k + 1), 0.03, -0.02, 0.01, -0.008, 0.005, 0};
        hp[6] = std::sqrt(1 - hp[3] * hp[3] - hp[4] * hp[4] - hp[5] * hp[5]);
        double hb[9] = {4.4, 0.1, -0.05, 0.02, -0.01, 0.015, 0.001, -0.002, 0.0015};
        std::memcpy(s.hid_pose_flat[k], hp, sizeof hp); std::memcpy(s.hid_sb_flat[k], hb, sizeof hb);
        if (separate) { s.hid_pose[k] = new double[7]; s.hid_sb[k] = new double[9]; std::memcpy(s.hid_pose[k], hp, sizeof hp); std::memcpy(s.hid_sb[k], hb, sizeof hb); }
        else { s.hid_pose[k] = s.hid_pose_flat[k]; s.hid_sb[k] = s.hid_sb_flat[k]; }
    }
}
// a pre-integration over `t` seconds of a body moving with ~constant velocity, gravity along -z of the body (identity attitude)
void fill_preintegration(IntegrationBase& ib, double t, int salt) {
    std::memset(&ib, 0, sizeof ib);
    ib.dt = 0.005; ib.sum_dt = t;
    for (int k = 0; k < 3; k++) {
        ib.linearized_ba.v[k] = 0.02 - 0.01 * k; ib.linearized_bg.v[k] = 0.001 * (k + 1);
        ib.gyri.v[k] = 0.01 * (k + 1 + salt); ib.gyrj.v[k] = 0.012 * (k + 1) - 0.001 * salt;
        ib.delta_p.v[k] = 0.001 * (k + salt); ib.delta_v.v[k] = 0.002 * (k - salt);
    }
    ib.delta_p.v[2] += 0.5 * G * t * t; ib.delta_v.v[2] += G * t;
    const double qx = 0.004 + 0.001 * salt, qy = -0.003, qz = 0.002;
    ib.delta_q = Q4{qx, qy, qz, std::sqrt(1 - qx * qx - qy * qy - qz * qz)};
    for (int i = 0; i < 15; i++) ib.jacobian.a[i][i] = 1.0;
    for (int i = 0; i < 3; i++) {
        ib.jacobian.a[0 + i][9 + i] = -0.5 * t * t; ib.jacobian.a[6 + i][9 + i] = -t; ib.jacobian.a[3 + i][12 + i] = -t;
        for (int j = 0; j < 3; j++) { ib.jacobian.a[0 + i][12 + j] += 0.01 * t * t * (i - j); ib.jacobian.a[6 + i][12 + j] += 0.05 * t * (j - i); }
    }
    const double w[5] = {200.0, 800.0, 150.0, 2000.0, 20000.0};         // upper-triangular square-root information, as get_sqrtinfo() leaves it
    for (int i = 0; i < 15; i++) for (int j = i; j < 15; j++) ib.sqrt_info.a[i][j] = i == j ? w[i / 3] : 0.01 * w[i / 3] / (1 + j - i + salt);
}
void project(const double* pose, const double* ex, const double* X, double* uv) {     // identity-ish attitudes: a first-order model is enough for synthetic observations
    uv[0] = (X[0] - pose[0] - ex[0]) / (X[2] - pose[2] - ex[2]); uv[1] = (X[1] - pose[1] - ex[1]) / (X[2] - pose[2] - ex[2]); uv[2] = 1;
}

struct Built { ceres::Problem* problem; std::vector<ceres::ResidualBlockId> ids; };

void add_common(ceres::Problem& P, State& s) {
    for (int i = 0; i < NF; i++) { P.AddParameterBlock(s.pose[i], 7, new PoseLocalParameterization()); P.AddParameterBlock(s.sb[i], 9); }
    P.AddParameterBlock(s.ex, 7, new PoseLocalParameterization()); P.SetParameterBlockConstant(s.ex);
    double pbg[3] = {0, 0, 0}, gw[3] = {0, 0, G}, base[3] = {-2.0e6, 5.4e6, 2.7e6};
    P.SetConstants(pbg, gw, base);
    State truth; init_state(truth, false);
    for (int l = 0; l < NL; l++) for (int i = 0; i < NF; i++) {
        double uv[3]; project(truth.pose[i], truth.ex, truth.pt[l], uv);
        uv[0] += 2e-3 * std::sin(3.1 * l + i); uv[1] += 2e-3 * std::cos(2.3 * l - i);
        P.AddResidualBlock(new projection_factor(uv), new ceres::CauchyLoss(1.0), s.pose[i], s.ex, s.pt[l]);
    }
    P.AddResidualBlock(new InitialBlackFactor(1), 0, &s.black);
}
void order(ceres::Solver::Options& o, State& s, const std::vector<double*>& head) {
    o.linear_solver_ordering.reset(new ceres::ParameterBlockOrdering());
    ceres::ParameterBlockOrdering* ord = o.linear_solver_ordering.get();
    std::set<double*> late(head.begin(), head.end());
    ord->AddElementToGroup(&s.black, 0);
    for (int l = 0; l < NL; l++) ord->AddElementToGroup(s.pt[l], 0);
    int g = 1;
    for (int i = 0; i < NF; i++) if (!late.count(s.sb[i])) ord->AddElementToGroup(s.sb[i], g++);
    for (int i = 0; i < NF; i++) if (!late.count(s.pose[i])) ord->AddElementToGroup(s.pose[i], g++);
    for (int k = 0; k < N; k++) if (!late.count(s.amb[k])) ord->AddElementToGroup(s.amb[k], g++);
    for (double* p : head) ord->AddElementToGroup(p, g++);
}

// the prior over (pose 0, speed-bias 0) and the per-epoch information of the composite factor: the same numbers for both builds
void prior_numbers(std::vector<double>& J, std::vector<double>& r0) {
    J.assign(15 * 15, 0.0); r0.assign(15, 0.0);
    for (int i = 0; i < 15; i++) { J[15 * i + i] = i < 3 ? 300.0 : i < 6 ? 500.0 : i < 9 ? 50.0 : 400.0; for (int j = i + 1; j < 15; j++) J[15 * i + j] = 0.3 * std::sin(i + 2.0 * j); r0[i] = 0.01 * std::cos(1.0 * i); }
}
void epoch_numbers(int k, double* Hpp /*225*/, double* HpN /*15 x N*/, double* rhs_p /*15*/) {
    std::memset(Hpp, 0, 225 * sizeof(double));
    for (int i = 0; i < 15; i++) { Hpp[15 * i + i] = i < 3 ? 2.0e3 : i < 6 ? 1.0e2 : i < 9 ? 4.0e2 : 1.0e3; rhs_p[i] = 0.05 * std::sin(0.7 * i + k); }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (i != j) Hpp[15 * i + j] = 30.0 / (1 + k);
    for (int i = 0; i < 15; i++) for (int j = 0; j < N; j++) HpN[i * N + j] = i < 3 ? 15.0 * (1 + j) / (1 + i + k) : 0.0;
}
}  // namespace

int main() {
    // the shared numbers
    std::vector<double> PJ, Pr0; prior_numbers(PJ, Pr0);
    IntegrationBase ib01; fill_preintegration(ib01, T_IMU, 0);
    IntegrationBase ibc[M + 1]; for (int k = 0; k <= M; k++) fill_preintegration(ibc[k], T_IMU / (M + 1) * 1.5, k + 1);
    double HNN[N * N] = {40.0, 3.0, 3.0, 55.0}, rhsN[N] = {0.2, -0.1};

    ceres::Solver::Options my_options;
    my_options.linear_solver_type = ceres::DENSE_SCHUR; my_options.max_num_iterations = 8; my_options.jacobi_scaling = 0;
    my_options.trust_region_strategy_type = ceres::DOGLEG; my_options.num_threads = 4;

    // ---------------- build A: the C-ABI's own records ----------------
    State A; init_state(A, false);
    ceres::Problem PA; add_common(PA, A);
    for (int k = 0; k < N; k++) PA.AddParameterBlock(A.amb[k], 1);
    {
        ceres::IMUFactor tmp(&ib01);                    // (the record itself comes from the converter; handed on as a plain record)
        PA.AddResidualBlock(new ceres::IMUFactor(tmp.pre.data()), 0, A.pose[0], A.sb[0], A.pose[1], A.sb[1]);
        std::vector<double> x0(A.pose[0], A.pose[0] + 7); x0.insert(x0.end(), A.sb[0], A.sb[0] + 9);
        PA.AddResidualBlock(new ceres::MarginalizationFactor(PJ.data(), Pr0.data(), x0.data(), 15, 16), 0, std::vector<double*>({A.pose[0], A.sb[0]}));
    }
    ceres::IMUGNSSInfo info;
    {
        info.M = M; info.hidden_pose = &A.hid_pose_flat[0][0]; info.hidden_sb = &A.hid_sb_flat[0][0];
        info.pose_lin.assign(&A.hid_pose_flat[0][0], &A.hid_pose_flat[0][0] + 7 * M); info.sb_lin.assign(&A.hid_sb_flat[0][0], &A.hid_sb_flat[0][0] + 9 * M);
        info.Hpp.resize(225 * M); info.HpN.resize(15 * N * M); info.rhs_p.resize(15 * M);
        for (int k = 0; k < M; k++) epoch_numbers(k, &info.Hpp[225 * k], &info.HpN[15 * N * k], &info.rhs_p[15 * k]);
        info.HNN.assign(HNN, HNN + N * N); info.rhsN.assign(rhsN, rhsN + N);
        for (int k = 0; k <= M; k++) { ceres::IMUFactor t(&ibc[k]); info.pre.insert(info.pre.end(), t.pre.begin(), t.pre.end()); }
        PA.AddResidualBlock(new ceres::IMUGNSSFactor(&info), 0, std::vector<double*>({A.pose[1], A.sb[1], A.pose[2], A.sb[2], A.amb[0], A.amb[1]}));
    }

    // ---------------- build B: the reference's constructors over stand-ins ----------------
    State B; init_state(B, true);
    ceres::Problem PB; add_common(PB, B);
    for (int k = 0; k < N; k++) PB.AddParameterBlock(B.amb[k], 1);
    PB.AddResidualBlock(new ceres::IMUFactor(&ib01), 0, B.pose[0], B.sb[0], B.pose[1], B.sb[1]);            // R/swf/swf_imu.cpp:185-193
    MarginalizationInfo mi;
    double keep_pose[7], keep_sb[9]; std::memcpy(keep_pose, B.pose[0], sizeof keep_pose); std::memcpy(keep_sb, B.sb[0], sizeof keep_sb);
    {
        // the prior's columns in ANOTHER order than the blocks handed to AddResidualBlock: speed-bias columns first (m = 4 marginalised dimensions in front)
        mi.m = 4; mi.n = 15; mi.keep_block_size = {7, 9}; mi.keep_block_idx = {4 + 9, 4 + 0}; mi.keep_block_data = {keep_pose, keep_sb};
        mi.linearized_jacobians = MX(15, 15); mi.linearized_residuals = MX(15, 1);
        for (int r = 0; r < 15; r++) {
            mi.linearized_residuals.a[r] = Pr0[r];
            for (int c = 0; c < 6; c++) mi.linearized_jacobians.at(r, 9 + c) = PJ[15 * r + c];
            for (int c = 0; c < 9; c++) mi.linearized_jacobians.at(r, c) = PJ[15 * r + 6 + c];
        }
        PB.AddResidualBlock(new ceres::MarginalizationFactor(&mi), 0, std::vector<double*>({B.pose[0], B.sb[0]}));   // R/swf/swf_image.cpp:205-209
    }
    IMUGNSSBase base;
    RefIMUFactor rf[M + 1];
    double lin_pose[M][7], lin_sb[M][9];
    {
        for (int k = 0; k < N; k++) base.gnss_phase_biases.push_back(B.amb[k]);
        base.phase_biases_hessians = MX(N, N); base.phase_biases_rhs = MX(N, 1);
        for (int i = 0; i < N; i++) { base.phase_biases_rhs.a[i] = rhsN[i]; for (int j = 0; j < N; j++) base.phase_biases_hessians.at(i, j) = HNN[i * N + j]; }
        for (int k = 0; k < M; k++) {
            std::memcpy(lin_pose[k], B.hid_pose[k], sizeof lin_pose[k]); std::memcpy(lin_sb[k], B.hid_sb[k], sizeof lin_sb[k]);
            base.gnss_poses.push_back(B.hid_pose[k]); base.gnss_speed_bias.push_back(B.hid_sb[k]);
            base.gnss_poses_lin.push_back(lin_pose[k]); base.gnss_speed_bias_lin.push_back(lin_sb[k]);
            double Hpp[225], HpN[15 * N], rp[15]; epoch_numbers(k, Hpp, HpN, rp);
            M15 h; MX hn(15, N), r(15, 1);
            for (int i = 0; i < 15; i++) { r.a[i] = rp[i]; for (int j = 0; j < 15; j++) h.a[i][j] = Hpp[15 * i + j]; for (int j = 0; j < N; j++) hn.at(i, j) = HpN[i * N + j]; }
            base.pose_hessians.push_back(h); base.pose_phase_biases_hessians.push_back(hn); base.pose_rhses.push_back(r);
            rf[k].pre_integration = &ibc[k]; base.imu_factors.push_back(&rf[k]);
        }
        rf[M].pre_integration = &ibc[M]; base.last_imu_factor = &rf[M];
        // SetLastImuFactor (R/factor/gnss_imu_factor.cpp:99-119)
        base.param = std::vector<double*>({B.pose[1], B.sb[1], B.pose[2], B.sb[2]});
        for (int i = 0; i < (int)base.gnss_phase_biases.size(); i++) base.param.push_back(base.gnss_phase_biases[i]);
        IMUGNSSFactor* factor = new IMUGNSSFactor(&base);
        PB.AddResidualBlock(factor, 0, base.param);
    }

    // ---------------- solve both ----------------
    ceres::internal::is_optimize = true;
    ceres::Solver::Summary sa, sb_;
    order(my_options, A, {}); ceres::Solve(my_options, &PA, &sa);
    order(my_options, B, {}); ceres::Solve(my_options, &PB, &sb_);
    std::printf("A: %s\nB: %s\n", sa.BriefReport().c_str(), sb_.BriefReport().c_str());
    if (sa.final_cost > 1e10 || sb_.final_cost > 1e10) return 1;
    int bad = 0;
    bad += !(sa.final_cost < 0.9 * sa.initial_cost);                               // the solve did something
    bad += std::memcmp(&sa.final_cost, &sb_.final_cost, sizeof(double)) != 0 || std::memcmp(&sa.initial_cost, &sb_.initial_cost, sizeof(double)) != 0;
    bad += sa.raw.num_iterations != sb_.raw.num_iterations;
    bad += std::memcmp(A.pose, B.pose, sizeof A.pose) != 0 || std::memcmp(A.sb, B.sb, sizeof A.sb) != 0 || std::memcmp(A.pt, B.pt, sizeof A.pt) != 0 || std::memcmp(A.amb, B.amb, sizeof A.amb) != 0;
    State fresh; init_state(fresh, false);
    int moved = 0;
    for (int k = 0; k < M; k++) {
        bad += std::memcmp(A.hid_pose_flat[k], B.hid_pose[k], 7 * sizeof(double)) != 0 || std::memcmp(A.hid_sb_flat[k], B.hid_sb[k], 9 * sizeof(double)) != 0;
        moved += std::memcmp(fresh.hid_pose_flat[k], B.hid_pose[k], 7 * sizeof(double)) != 0;
    }
    bad += moved != M;                                                             // UpdateHiddenState reached the epochs' own memory
    std::printf("reference-constructor build == record build: %s (hidden epochs updated in place: %d of %d)\n", bad ? "NO" : "bit for bit", moved, M);
    if (bad) return 2;

    // ---------------- the raw globals, read as UpdateSchur reads them (R/swf/swf_gnss.cpp:25-61) ----------------
    ceres::internal::parameter_head.clear();
    ceres::internal::parameter_head.push_back(B.pose[2]); ceres::internal::parameter_head.push_back(B.sb[2]);
    ceres::internal::parameter_head.push_back(B.amb[0]); ceres::internal::parameter_head.push_back(B.amb[1]);
    order(my_options, B, ceres::internal::parameter_head);
    ceres::internal::is_optimize = false;
    my_options.max_num_iterations = 1;
    ceres::Solve(my_options, &PB, &sb_);
    assert(!ceres::internal::is_optimize);
    {
        const double* lhs = ceres::internal::lhs_out; const double* rhs = ceres::internal::rhs_out; const int hs_row = ceres::internal::hs_row;
        int parameter_head_all_size = 0;
        for (size_t i = 0; i < ceres::internal::parameter_head.size(); i++) {
            int size = PB.ParameterBlockSize(ceres::internal::parameter_head[i]);
            if (size == 7) parameter_head_all_size += 6; else parameter_head_all_size += size;
        }
        int m = hs_row - parameter_head_all_size, n = parameter_head_all_size;
        bad += !(lhs && rhs && hs_row == 3 * 15 + N - 0 && n == 17 && m == hs_row - 17);
        // selfadjointView<Upper> of a full symmetric matrix: both triangles are there, positive diagonal, finite right-hand side
        double asym = 0, dmin = 1e300;
        for (int i = 0; lhs && i < hs_row; i++) { dmin = std::fmin(dmin, lhs[(size_t)i * hs_row + i]); bad += !std::isfinite(rhs[i]); for (int j = 0; j < i; j++) asym = std::fmax(asym, std::fabs(lhs[(size_t)i * hs_row + j] - lhs[(size_t)j * hs_row + i])); }
        bad += !(asym == 0 && dmin > 0);
        std::printf("UpdateSchur's view: hs_row = %d, m = %d, n = %d, min diagonal %.3e\n", hs_row, m, n, dmin);
        // lhs_out2 (UpdateSchurHessianOnly, :65-94): A3 = tail block of the lower factor; A3 A3^T = marginal information of the head:
        // compared with the adapter's own one-call form of the same reader
        const double* l2 = ceres::internal::lhs_out2;
        std::vector<double> AA((size_t)n * n, 0.0);
        for (int i = 0; l2 && i < n; i++) for (int j = 0; j < n; j++) { double a = 0; for (int k = 0; k < n; k++) a += l2[(size_t)(m + i) * hs_row + m + k] * l2[(size_t)(m + j) * hs_row + m + k]; AA[(size_t)i * n + j] = a; }
        ceres::MarginalPrior mp;
        std::vector<double*> keep = ceres::internal::parameter_head;
        bool ok = ceres::UpdateSchurAndSetMarginalizeInfo(&PB, &mp, false);
        double err = 0, sc = 0;
        for (int i = 0; ok && i < n * n; i++) { err = std::fmax(err, std::fabs(AA[i] - mp.A[i])); sc = std::fmax(sc, std::fabs(mp.A[i])); }
        bad += !(ok && l2 && mp.n == n && err <= 1e-10 * sc) || !ceres::internal::parameter_head.empty();
        std::printf("lhs_out2 tail block: |A3 A3^T - A| / |A| = %.2e\n", sc > 0 ? err / sc : -1.0);
    }
    ceres::internal::is_optimize = true;
    std::printf(bad ? "reference solve: %d checks FAILED\n" : "reference solve: ok\n", bad);
    return bad ? 3 : 0;
}
