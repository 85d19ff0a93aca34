// The reference's marginalisation-through-the-solver, SWFOptimization::GlobalMarge (R/swf/swf_image.cpp:343-433), written against
// include/swf_ceres.hpp statement by statement: freeze every block, switch every residual block off, walk the residual blocks of
// the blocks to marginalise (GetResidualBlocksForParameterBlock / ->is_use / GetParameterBlocksForResidualBlock), un-freeze
// their neighbours — the camera extrinsic among them — as ceres::internal::parameter_head, Solve with is_optimize = false,
// UpdateSchur + setmarginalizeinfo, restore, swap the prior.  tests/test_host.py compiles it; it RUNS on the GPU box only.
#include <cmath>
#include <cstdio>
#include <set>
#include <vector>
#include "swf_ceres.hpp"
namespace ceres = swf_ceres;
using namespace swf_ceres;

namespace {
const int NP = 4, NL = 12;
double para_pose[NP][7], para_ex_Pose[7] = {0.02, -0.01, 0.03, 0, 0, 0, 1}, blackvalue2 = 0;
double ptsInWorld[NL][3];
ceres::Problem my_problem;
ceres::Solver::Options my_options;
ceres::ResidualBlockId marg_residual_block_id = nullptr;
std::vector<double*> last_keep_block_addr;            // last_marg_info->keep_block_addr

void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1]; o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3]; o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void rotate_inv(const double* q, const double* v, double* o) {      // q^-1 v q for a unit quaternion (x, y, z, w)
    double qi[4] = {-q[0], -q[1], -q[2], q[3]}, t[4], vv[4] = {v[0], v[1], v[2], 0}, r[4];
    quat_mul(qi, vv, t); quat_mul(t, q, r); o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
// normalised image coordinates of a world point in frame `pose` through the extrinsic (projection_factor's model, Pbg = 0)
void project(const double* pose, const double* ex, const double* X, double* uv) {
    double d[3] = {X[0] - pose[0], X[1] - pose[1], X[2] - pose[2]}, pi[3], t[3], pc[3];
    rotate_inv(pose + 3, d, pi);
    for (int k = 0; k < 3; k++) t[k] = pi[k] - ex[k];
    rotate_inv(ex + 3, t, pc);
    uv[0] = pc[0] / pc[2]; uv[1] = pc[1] / pc[2];
}
// SWFOptimization::MyOrdering (R/swf/swf_gnss.cpp:629-783) for this toy window: dummy + landmarks in group 0, then the poses and
// the extrinsic, the prior's kept blocks and ceres::internal::parameter_head last; constant blocks are dropped by the solver
void MyOrdering() {
    ceres::ParameterBlockOrdering* ordering = my_options.linear_solver_ordering.get();
    ordering->Clear();
    std::set<double*> late(last_keep_block_addr.begin(), last_keep_block_addr.end());
    late.insert(ceres::internal::parameter_head.begin(), ceres::internal::parameter_head.end());
    int g = 0;
    if (my_problem.HasParameterBlock(&blackvalue2)) ordering->AddElementToGroup(&blackvalue2, 0);
    for (int l = 0; l < NL; l++) if (my_problem.HasParameterBlock(ptsInWorld[l]) && !late.count(ptsInWorld[l])) ordering->AddElementToGroup(ptsInWorld[l], 0);
    g = 1;
    for (int i = 0; i < NP; i++) if (my_problem.HasParameterBlock(para_pose[i]) && !late.count(para_pose[i])) ordering->AddElementToGroup(para_pose[i], g++);
    if (my_problem.HasParameterBlock(para_ex_Pose) && !late.count(para_ex_Pose)) ordering->AddElementToGroup(para_ex_Pose, g++);
    std::set<double*> head(ceres::internal::parameter_head.begin(), ceres::internal::parameter_head.end());
    for (double* p : last_keep_block_addr) if (my_problem.HasParameterBlock(p) && !head.count(p)) ordering->AddElementToGroup(p, g++);
    for (double* p : ceres::internal::parameter_head) ordering->AddElementToGroup(p, g++);
}

int GlobalMarge(const std::set<double*>& MargePoints) {
    std::vector<ceres::ResidualBlockId> residual_blocks_all;
    std::set<double*> parameter_head;
    std::vector<double*> parameter_blocks_all;
    my_problem.GetParameterBlocks(&parameter_blocks_all);
    for (int i = 0; i < (int)parameter_blocks_all.size(); i++) my_problem.SetParameterBlockConstant(parameter_blocks_all[i]);
    my_problem.GetResidualBlocks(&residual_blocks_all);
    for (int i = 0; i < (int)residual_blocks_all.size(); i++) residual_blocks_all[i]->is_use = residual_blocks_all[i] == marg_residual_block_id;
    for (auto it = MargePoints.begin(); it != MargePoints.end(); it++) {
        std::vector<ceres::ResidualBlockId> residual_blocks;
        if (my_problem.HasParameterBlock(*it)) my_problem.GetResidualBlocksForParameterBlock(*it, &residual_blocks);
        for (int i = 0; i < (int)residual_blocks.size(); i++) {
            if (residual_blocks[i]->is_use) continue;
            residual_blocks[i]->is_use = true;
            std::vector<double*> parameter_blocks;
            my_problem.GetParameterBlocksForResidualBlock(residual_blocks[i], &parameter_blocks);
            for (int j = 0; j < (int)parameter_blocks.size(); j++) {
                if (MargePoints.find(parameter_blocks[j]) != MargePoints.end()) continue;
                parameter_head.insert(parameter_blocks[j]);
            }
        }
    }
    for (double* p : last_keep_block_addr) if (MargePoints.find(p) == MargePoints.end()) parameter_head.insert(p);
    for (auto it = parameter_head.begin(); it != parameter_head.end(); it++) {
        my_problem.SetParameterBlockVariable(*it);
        ceres::internal::parameter_head.push_back(*it);
    }
    for (auto it = MargePoints.begin(); it != MargePoints.end(); it++) if (my_problem.HasParameterBlock(*it)) my_problem.SetParameterBlockVariable(*it);
    if (!parameter_head.count(para_ex_Pose)) { std::printf("the extrinsic is not in parameter_head\n"); return 10; }

    my_options.max_num_iterations = 1;
    my_options.jacobi_scaling = false;
    ceres::internal::is_optimize = false;
    ceres::Solver::Summary summary;
    MyOrdering();
    std::vector<double*> keep = ceres::internal::parameter_head;             // UpdateSchur clears the global
    ceres::Solve(my_options, &my_problem, &summary);
    if (summary.final_cost > 1e10) { std::printf("marginalisation solve failed: %s\n", summary.message.c_str()); return 11; }
    my_options.max_num_iterations = 8;
    ceres::MarginalPrior mp;
    bool ok = ceres::UpdateSchurAndSetMarginalizeInfo(&my_problem, &mp);     // UpdateSchur + setmarginalizeinfo(..., Sqrt = true)
    ceres::internal::is_optimize = true;
    if (!ok) { std::printf("UpdateSchur failed: n = %d rank = %d (%s)\n", mp.n, mp.rank, swf_last_error()); return 12; }
    int dim = 0, gsum = 0;
    std::vector<double> x0;
    for (double* p : keep) { int s = my_problem.ParameterBlockSize(p); dim += s == 7 ? 6 : s; gsum += s; x0.insert(x0.end(), p, p + s); }
    if (dim != mp.n) { std::printf("prior dimension %d != sum of the kept blocks' local sizes %d\n", mp.n, dim); return 13; }
    // J^T J reproduces the marginal information A (eigenvalues below the reference's 1e-8 threshold are dropped)
    double err = 0, sc = 0;
    for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) {
        double a = 0;
        for (int k = 0; k < dim; k++) a += mp.linearized_jacobians[k * dim + i] * mp.linearized_jacobians[k * dim + j];
        err = std::fmax(err, std::fabs(a - mp.A[i * dim + j])); sc = std::fmax(sc, std::fabs(mp.A[i * dim + j]));
    }
    std::printf("GlobalMarge: %d kept blocks, prior n = %d rank = %d, |J^T J - A| / |A| = %.2e\n", (int)keep.size(), mp.n, mp.rank, err / sc);
    if (!(err <= 1e-6 * sc) || mp.rank < dim - 7) return 14;
    ceres::MarginalizationFactor* factor = new ceres::MarginalizationFactor(mp.linearized_jacobians, mp.linearized_residuals, x0.data(), dim, gsum);

    for (int i = 0; i < (int)residual_blocks_all.size(); i++) residual_blocks_all[i]->is_use = true;
    for (int i = 0; i < (int)parameter_blocks_all.size(); i++) my_problem.SetParameterBlockVariable(parameter_blocks_all[i]);
    if (marg_residual_block_id) my_problem.RemoveResidualBlock(marg_residual_block_id);
    last_keep_block_addr = keep;
    marg_residual_block_id = my_problem.AddResidualBlock(factor, 0, last_keep_block_addr);
    return 0;
}
}  // namespace

int main() {
    my_options.linear_solver_type = ceres::DENSE_SCHUR;
    my_options.trust_region_strategy_type = ceres::DOGLEG;
    my_options.max_num_iterations = 8;
    my_options.jacobi_scaling = false;
    my_options.num_threads = 4;
    my_options.linear_solver_ordering.reset(new ceres::ParameterBlockOrdering());
    // a camera moving sideways past a cloud of points; landmarks 0..5 are first seen in frame 0 (marginalised with it)
    double truth_pose[NP][7];
    for (int i = 0; i < NP; i++) {
        double yaw = 0.03 * i, q[4] = {0, 0, std::sin(yaw / 2), std::cos(yaw / 2)};
        double p[7] = {0.5 * i, 0.05 * i * i, 0.02 * i, q[0], q[1], q[2], q[3]};
        for (int k = 0; k < 7; k++) { truth_pose[i][k] = p[k]; para_pose[i][k] = p[k]; }
        para_pose[i][0] += 0.02 * ((i * 7) % 5 - 2); para_pose[i][1] -= 0.015 * ((i * 3) % 4 - 1);
    }
    double truth_lm[NL][3];
    for (int l = 0; l < NL; l++) {
        truth_lm[l][0] = -1.5 + 0.4 * l + 0.1 * ((l * 5) % 3); truth_lm[l][1] = -1.0 + 0.25 * ((l * 7) % 9); truth_lm[l][2] = 6.0 + 0.7 * ((l * 3) % 5);
        for (int k = 0; k < 3; k++) ptsInWorld[l][k] = truth_lm[l][k] + 0.05 * (((l + k) * 11) % 7 - 3);
    }
    for (int i = 0; i < NP; i++) my_problem.AddParameterBlock(para_pose[i], 7, new PoseLocalParameterization());
    my_problem.AddParameterBlock(para_ex_Pose, 7, new PoseLocalParameterization());
    for (int l = 0; l < NL; l++)
        for (int i = (l < 6 ? 0 : 1); i < NP; i++) {
            double uv[3] = {0, 0, 1};
            project(truth_pose[i], para_ex_Pose, truth_lm[l], uv);
            my_problem.AddResidualBlock(new projection_factor(uv), new ceres::CauchyLoss(1.0), para_pose[i], para_ex_Pose, ptsInWorld[l]);   // R/swf/swf_image.cpp:98-100
        }
    my_problem.AddResidualBlock(new InitialBlackFactor(1), 0, &blackvalue2);
    // the gauge: a first marginalisation prior over the two oldest poses (what InitializeSqrtInfo + the first slide leave behind)
    {
        std::vector<double> J(144, 0.0), r0(12, 0.0), x0;
        for (int k = 0; k < 12; k++) J[k * 12 + k] = 300.0;
        x0.insert(x0.end(), truth_pose[0], truth_pose[0] + 7); x0.insert(x0.end(), truth_pose[1], truth_pose[1] + 7);
        last_keep_block_addr = {para_pose[0], para_pose[1]};
        marg_residual_block_id = my_problem.AddResidualBlock(new ceres::MarginalizationFactor(J.data(), r0.data(), x0.data(), 12, 14), 0, last_keep_block_addr);
    }
    // MyOptimization: the extrinsic is constant while optimising (ESTIMATE_EXTRINSIC 0, R/swf/swf_image.cpp:174-176)
    my_problem.SetParameterBlockConstant(para_ex_Pose);
    ceres::Solver::Summary summary;
    MyOrdering();
    ceres::Solve(my_options, &my_problem, &summary);
    std::printf("%s\n", summary.BriefReport().c_str());
    if (summary.final_cost > 1e10 || !(summary.final_cost < 1e-2 * summary.initial_cost)) return 1;

    // marginalise the oldest frame and the landmarks first seen in it
    std::set<double*> MargePoints = {para_pose[0]};
    for (int l = 0; l < 6; l++) MargePoints.insert(ptsInWorld[l]);
    int rc = GlobalMarge(MargePoints);
    if (rc) return rc;
    if (my_problem.IsParameterBlockConstant(para_ex_Pose)) return 2;         // GlobalMarge leaves every block variable, as the reference does
    // slide the window: the marginalised blocks leave the problem (their residual blocks go with them), the extrinsic is frozen
    // again, and the next optimisation runs on the new prior
    for (double* p : MargePoints) my_problem.RemoveParameterBlock(p);
    my_problem.SetParameterBlockConstant(para_ex_Pose);
    std::vector<ceres::ResidualBlockId> left;
    my_problem.GetResidualBlocks(&left);
    if ((int)left.size() != 6 * 3 + 1 + 1) { std::printf("%d residual blocks left\n", (int)left.size()); return 3; }
    para_pose[3][0] += 0.05; para_pose[2][1] -= 0.04;
    MyOrdering();
    ceres::Solve(my_options, &my_problem, &summary);
    std::printf("%s\n", summary.BriefReport().c_str());
    if (summary.final_cost > 1e10 || !(summary.final_cost < 0.5 * summary.initial_cost)) return 4;
    // the poses come back to the truth the first prior pinned the gauge to (the new prior carries that information on)
    double perr = 0;
    for (int i = 1; i < NP; i++) for (int k = 0; k < 3; k++) perr = std::fmax(perr, std::fabs(para_pose[i][k] - truth_pose[i][k]));
    std::printf("position error after the slide: %.2e\n", perr);
    return perr < 5e-3 ? 0 : 5;
}
