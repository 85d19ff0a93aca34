// The reference's own constructor signatures of the two factors that carry a pointer to an estimator-side object,
//   IMUFactor(IntegrationBase*)                      R/factor/imu_factor.h:11
//   MarginalizationFactor(MarginalizationInfo*)      R/factor/marginalization_factor.h:106
// against include/swf_ceres.hpp, with stand-ins that have the reference's MEMBER NAMES and Eigen's accessors (operator()(i),
// operator()(i, j), x() y() z() w()) — the image has no Eigen; the adapter's templates use nothing else of the types.
// Host only: no Problem is created, nothing touches a GPU.
#include <cmath>
#include <cstdio>
#include <vector>
#include "swf_ceres.hpp"
namespace ceres = swf_ceres;

struct V3 { double v[3]; double operator()(int i) const { return v[i]; } };
struct Q4 { double x_, y_, z_, w_; double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; } };
struct M15 { double a[15][15]; double operator()(int i, int j) const { return a[i][j]; } };
struct MX { int rows, cols; std::vector<double> a; double operator()(int i, int j) const { return a[(size_t)i * cols + j]; } double operator()(int i) const { return a[i]; } };

struct IntegrationBase {                 // R/factor/integration_base.h:28-47
    double dt, sum_dt;
    V3 acc_0, gyr_0, acc_1, gyr_1, linearized_ba, linearized_bg, delta_p, delta_v, gyri, gyrj;
    Q4 delta_q;
    M15 jacobian, covariance, sqrt_info;
    M15 get_sqrtinfo() { return sqrt_info; }
};
struct MarginalizationInfo {             // R/factor/marginalization_factor.h:42-100
    int m, n;
    std::vector<int> keep_block_size, keep_block_idx;
    std::vector<double*> keep_block_data;
    MX linearized_jacobians, linearized_residuals;
};

int main() {
    int bad = 0;
    IntegrationBase ib{};
    ib.sum_dt = 0.05;
    for (int k = 0; k < 3; k++) { ib.delta_p.v[k] = 1 + k; ib.delta_v.v[k] = 4 + k; ib.linearized_ba.v[k] = 0.1 * (k + 1); ib.linearized_bg.v[k] = 0.01 * (k + 1); ib.gyri.v[k] = 7 + k; ib.gyrj.v[k] = 10 + k; }
    ib.delta_q = Q4{0.1, 0.2, 0.3, 0.9};
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) { ib.jacobian.a[i][j] = 100 * i + j; ib.sqrt_info.a[i][j] = j >= i ? 1000 + 15 * i + j : 0.0; }
    ceres::IMUFactor* f = new ceres::IMUFactor(&ib);
    const double* p = f->pre.data();
    bad += !(p[SWF_PRE_DP + 1] == 2 && p[SWF_PRE_DV + 2] == 6 && p[SWF_PRE_LBA] == 0.1 && p[SWF_PRE_LBG + 2] == 0.03 && p[SWF_PRE_GYRI] == 7 && p[SWF_PRE_GYRJ + 2] == 12);
    bad += !(p[SWF_PRE_DQ] == 0.1 && p[SWF_PRE_DQ + 1] == 0.2 && p[SWF_PRE_DQ + 2] == 0.3 && p[SWF_PRE_DQ + 3] == 0.9);          // x y z w
    bad += !(p[SWF_PRE_DP_DBA + 3 * 1 + 2] == 100 * 1 + 11 && p[SWF_PRE_DP_DBG + 3 * 2 + 0] == 100 * 2 + 12 && p[SWF_PRE_DQ_DBG + 4] == 100 * 4 + 13 &&
             p[SWF_PRE_DV_DBA + 0] == 100 * 6 + 9 && p[SWF_PRE_DV_DBG + 8] == 100 * 8 + 14);
    bad += !(p[SWF_PRE_SUMDT] == 0.05 && p[SWF_PRE_SQRTINFO + 15 * 3 + 7] == 1000 + 15 * 3 + 7 && p[SWF_PRE_SQRTINFO + 15 * 7 + 3] == 0.0 && (int)f->pre.size() == SWF_PRE_DOUBLES);
    delete f;

    // a prior over a pose (7 / 6), a speed-bias (9) and a scalar (1) whose columns sit in ANOTHER order than the blocks: m = 5, the
    // scalar's column first, then the pose's six, then the speed-bias's nine
    double pose[7] = {1, 2, 3, 0, 0, 0, 1}, sb[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9}, sc[1] = {42};
    MarginalizationInfo mi;
    mi.m = 5; mi.n = 16;
    mi.keep_block_size = {7, 9, 1}; mi.keep_block_idx = {5 + 1, 5 + 7, 5 + 0}; mi.keep_block_data = {pose, sb, sc};
    mi.linearized_jacobians = MX{16, 16, std::vector<double>(256)}; mi.linearized_residuals = MX{16, 1, std::vector<double>(16)};
    for (int r = 0; r < 16; r++) { mi.linearized_residuals.a[r] = -r; for (int c = 0; c < 16; c++) mi.linearized_jacobians.a[r * 16 + c] = 100 * r + c; }
    ceres::MarginalizationFactor* g = new ceres::MarginalizationFactor(&mi);
    bad += !((int)g->J.size() == 256 && (int)g->r0.size() == 16 && (int)g->x0.size() == 17);
    // adapter columns: [pose 0..5 | speed-bias 6..14 | scalar 15]  <-  prior columns [1..6 | 7..15 | 0]
    bad += !(g->J[16 * 3 + 0] == 100 * 3 + 1 && g->J[16 * 3 + 5] == 100 * 3 + 6 && g->J[16 * 9 + 6] == 100 * 9 + 7 && g->J[16 * 9 + 14] == 100 * 9 + 15 && g->J[16 * 2 + 15] == 100 * 2 + 0);
    bad += !(g->r0[7] == -7 && g->x0[6] == 1 && g->x0[7] == 1 && g->x0[15] == 9 && g->x0[16] == 42);
    delete g;
    std::printf(bad ? "reference constructors: %d checks FAILED\n" : "reference constructors: ok\n", bad);
    return bad ? 1 : 0;
}
