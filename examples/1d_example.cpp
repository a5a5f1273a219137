// The reference's plumbing case (BASELINE.json configs[0], src/1d_example.cpp) on the MI355X path:
// one nonlinear 1-D range-sensor factor, GH degree 10, 10 NGD iterations from mu = 20, Lambda = 1/9
// with base step 0.75; writes mean / precision / cov / cost / factor_costs / costmap CSVs that the
// tests compare with the reference's committed data/1d/*.csv.
//
//   1d_example <output-dir>/
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "gvi/gvi_host.hpp"

using namespace gvi;

// psi of the example (src/1d_example.cpp:25-35): Gaussian prior on x plus a stereo range measurement.
static const double kMuP = 20.0, kF = 400.0, kB = 0.1, kSigR2 = 0.09, kSigP2 = 9.0;
static const double kY = kF * kB / kMuP - 0.8;

double cost_function(const VectorXd& vec_x, const NoneType&) {
  const double x = vec_x(0);
  const double e = x - kMuP, r = kY - kF * kB / x;
  return e * e / kSigP2 / 2 + r * r / kSigR2 / 2;
}

int main(int argc, char** argv) {
  const std::string prefix = argc > 1 ? argv[1] : "./";
  const int dim_state = 1, num_states = 1, dim_factor = 1, start_index = 0, gh_degree = 10, n_iters = 10;
  const double temperature = 1.0, high_temperature = 10.0;

  std::vector<std::shared_ptr<NGDFactorizedSimpleGH>> vec_opt_fact;
  vec_opt_fact.emplace_back(new NGDFactorizedSimpleGH(
      dim_factor, dim_state, gh_degree, cost_function, NoneType{}, num_states, start_index, temperature,
      high_temperature, DevicePsi::Range1D(kY, kMuP, kF * kB, kSigR2, kSigP2)));

  VectorXd init_mu = VectorXd::Constant(1, 20.0);
  SpMat init_prec(1, 1);
  init_prec.coeffRef(0, 0) = 1.0 / 9.0;

  NGDGH<NGDFactorizedSimpleGH> opt{vec_opt_fact, dim_state, num_states, n_iters};
  opt.set_niter_low_temperature(n_iters);
  opt.update_file_names(prefix);
  opt.save_costmap(prefix + "costmap.csv");
  opt.set_initial_values(init_mu, init_prec);
  opt.set_step_size_base(0.75);
  std::printf("opt.mu\n%.15g\n", opt.mean()(0));
  opt.optimize();

  // the generic surface: the same psi as an opaque host function through SparseGaussHermite
  SparseGaussHermite<> gh(gh_degree, 1, opt.mean(), MatrixXd::Constant(1, 1, opt.covariance().coeff(0, 0)));
  MatrixXd e = gh.Integrate([](const VectorXd& x) { return MatrixXd::Constant(1, 1, cost_function(x, NoneType{})); });
  std::printf("E[psi] at the final proposal via the host-callback route: %.15g\n", e(0, 0));
  // the reference's device-variant timing hook (gvibase/GVI-GH-Cuda-impl.h:463-527)
  opt.time_test();
  return 0;
}
