// Compile-only example: estimator code in the reference's style against include/swf_ceres.hpp.
// (tests/test_host.py compiles it with g++ and links libswf_hip.so; it is only RUN on a GPU box.)
#include <cmath>
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
    if (summary.final_cost > 1e10) return 1;
    {   // a default-constructed Solver::Options, as GnssPreprocess / GnssProcess use it (R/swf/swf_gnss.cpp:200-216, 562-572):
        // LEVENBERG_MARQUARDT, no linear_solver_ordering (the solver orders the blocks itself), results never inspected
        ceres::Solver::Options options;
        options.linear_solver_type = ceres::DENSE_SCHUR;
        options.initial_trust_region_radius = options.max_trust_region_radius = 1e15;
        options.max_num_iterations = 2;
        options.num_threads = 1;
        ceres::Solver::Summary s2;
        ceres::Solve(options, &my_problem, &s2);
        std::printf("%s\n", s2.BriefReport().c_str());
        if (s2.final_cost > 1e10 || s2.final_cost > summary.final_cost * (1 + 1e-9)) return 3;
    }
    // GlobalMarge's sequence (R/swf/swf_image.cpp:404-418): keep pose1 as parameter_head, assemble + eliminate only,
    // then UpdateSchur + setmarginalizeinfo in one call
    // (two landmarks are held constant: 4 points seen from a fixed and a free camera give 8 constraints on the free
    //  pose and the depths; with fewer than two fixed points the marginal information of pose1 is rank-deficient)
    my_problem.SetParameterBlockConstant(pt[0]);
    my_problem.SetParameterBlockConstant(pt[1]);
    ordering->Clear();
    ordering->AddElementToGroup(&blackvalue2, 0);
    for (int i = 2; i < 4; i++) ordering->AddElementToGroup(pt[i], 0);
    ordering->AddElementToGroup(pose1, 1);
    ceres::internal::parameter_head.push_back(pose1);
    ceres::internal::is_optimize = false;
    ceres::Solve(my_options, &my_problem, &summary);
    ceres::internal::is_optimize = true;
    ceres::MarginalPrior mp;
    if (!ceres::UpdateSchurAndSetMarginalizeInfo(&my_problem, &mp) || mp.n != 6 || mp.rank != 6) {
        std::printf("marginalisation failed: n = %d rank = %d (%s)\n", mp.n, mp.rank, swf_last_error());
        return 2;
    }
    // J^T J must reproduce the marginal information of pose1
    double err = 0, sc = 0;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
        double a = 0;
        for (int k = 0; k < 6; k++) a += mp.linearized_jacobians[k * 6 + i] * mp.linearized_jacobians[k * 6 + j];
        err = std::fmax(err, std::fabs(a - mp.A[i * 6 + j])); sc = std::fmax(sc, std::fabs(mp.A[i * 6 + j]));
    }
    std::printf("marginal prior: n = %d rank = %d  |J^T J - A| / |A| = %.2e\n", mp.n, mp.rank, err / sc);
    return err > 1e-10 * sc;
}
