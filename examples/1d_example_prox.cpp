// src/1d_example_proxGVI.cpp of the reference, restated on the header-only shim (include/gvi/gvi_host.hpp):
// the proximal (JKO) update on the 1-D range-sensor factor.  Writes the CSVs the reference committed under
// data/1d_proxgvi/ (tests compare: SURVEY.md 8(f)4).   Usage: 1d_example_prox <output prefix>
#include <cstdio>
#include <string>

#include "gvi/gvi_host.hpp"

using namespace gvi;

static double cost_function(const VectorXd& vec_x, const NoneType&) {          // src/1d_example_proxGVI.cpp:25-36
  const double x = vec_x(0);
  const double mu_p = 20, f = 400, b = 0.1, sig_r_sq = 0.09, sig_p_sq = 9;
  const double y = f * b / mu_p - 0.8;
  return (x - mu_p) * (x - mu_p) / sig_p_sq / 2 + (y - f * b / x) * (y - f * b / x) / sig_r_sq / 2;
}

int main(int argc, char** argv) {
  const std::string prefix = argc > 1 ? argv[1] : "./";
  const int dim_state = 1, num_states = 1, dim_factor = 1, start_index = 0, gh_degree = 10, n_iters = 10;
  const double temperature = 1.0, high_temperature = 10.0;
  std::vector<std::shared_ptr<ProxGVIFactorizedSimpleGH>> vec_opt_fact;
  vec_opt_fact.emplace_back(new ProxGVIFactorizedSimpleGH(
      dim_factor, dim_state, gh_degree, cost_function, NoneType{}, num_states, start_index, temperature, high_temperature,
      DevicePsi::Range1D(400 * 0.1 / 20 - 0.8, 20.0, 400 * 0.1, 0.09, 9.0)));
  VectorXd init_mu = VectorXd::Constant(1, 20.0);
  SpMat init_prec(1, 1);
  init_prec.coeffRef(0, 0) = 1.0 / 9.0;
  ProxGVIGH<ProxGVIFactorizedSimpleGH> opt{vec_opt_fact, dim_state, num_states, n_iters};
  opt.set_niter_low_temperature(n_iters);
  opt.update_file_names(prefix);
  opt.set_initial_values(init_mu, init_prec);
  opt.set_step_size_base(0.75);
  std::printf("opt.mu\n%.15g\n", opt.mean()(0));
  opt.optimize();
  return 0;
}
