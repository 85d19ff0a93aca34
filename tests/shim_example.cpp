// Compile-only example: estimator code in the reference's style against include/swf_ceres.hpp.
// (tests/test_host.py compiles it with g++ and links libswf_hip.so; it is only RUN on a GPU box.)
#include <cstdio>
#include <vector>
#include "swf_ceres.hpp"
namespace ceres = swf_ceres;
using namespace swf_ceres;

int main() {
    ceres::Problem my_problem;
    ceres::Solver::Options my_options;
    my_options.linear_solver_type = ceres::DENSE_SCHUR;            // R/swf/swf.cpp:25-30
    my_options.max_num_iterations = 8;
    my_options.jacobi_scaling = 0;
    my_options.trust_region_strategy_type = ceres::DOGLEG;
    my_options.num_threads = 4;
    my_options.linear_solver_ordering.reset(new ceres::ParameterBlockOrdering());

    double pose0[7] = {0, 0, 0, 0, 0, 0, 1}, pose1[7] = {0.4, 0, 0, 0, 0, 0, 1}, ex[7] = {0, 0, 0, 0, 0, 0, 1};
    double pt[4][3] = {{0.5, 0.3, 6}, {-0.4, 0.2, 7}, {0.2, -0.5, 8}, {-0.3, -0.3, 9}};
    double blackvalue2 = 0;
    my_problem.AddParameterBlock(pose0, 7, new PoseLocalParameterization());
    my_problem.AddParameterBlock(pose1, 7, new PoseLocalParameterization());
    my_problem.AddParameterBlock(ex, 7, new PoseLocalParameterization());
    my_problem.SetParameterBlockConstant(ex);
    my_problem.SetParameterBlockConstant(pose0);
    for (int i = 0; i < 4; i++) {
        double u0[3] = {pt[i][0] / pt[i][2], pt[i][1] / pt[i][2], 1}, u1[3] = {(pt[i][0] - 0.4) / pt[i][2] + 1e-3, pt[i][1] / pt[i][2], 1};
        ceres::LossFunction* loss_function = new ceres::CauchyLoss(1.0);
        my_problem.AddResidualBlock(new projection_factor(u0), loss_function, pose0, ex, pt[i]);   // R/swf/swf_image.cpp:98-100
        my_problem.AddResidualBlock(new projection_factor(u1), new ceres::CauchyLoss(1.0), pose1, ex, pt[i]);
    }
    my_problem.AddResidualBlock(new InitialBlackFactor(1), 0, &blackvalue2);                      // R/swf/swf_core.cpp:553-556
    ceres::ParameterBlockOrdering* ordering = my_options.linear_solver_ordering.get();            // MyOrdering
    ordering->Clear();
    ordering->AddElementToGroup(&blackvalue2, 0);
    for (int i = 0; i < 4; i++) ordering->AddElementToGroup(pt[i], 0);
    ordering->AddElementToGroup(pose1, 1);
    ceres::internal::is_optimize = true;
    ceres::Solver::Summary summary;
    ceres::Solve(my_options, &my_problem, &summary);
    std::printf("%s\n", summary.BriefReport().c_str());
    return summary.final_cost > 1e10;
}
