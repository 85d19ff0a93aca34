// swf_problem.cpp — placeholder; the ceres::Problem-shaped layer is added in the next step.
