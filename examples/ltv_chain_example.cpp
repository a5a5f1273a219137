// BASELINE configs[2] (the headline) as a reference-side caller would write it: a chain of LTV_GP priors
// (gp/LTV_prior.h) built from the system matrices A(t), B(t) -- four piece-wise-constant pairs per factor -- wrapped in
// LinearGpPriorGH (gp/factorized_opts_LTV.h), one FixedPriorGP unary factor per state, optimised with gvi::NGDGH on the
// device-resident path.
//
//   Usage: ltv_chain_example <problem file> <iterations> <output file>
//   iterations < 0: only construct the LTV_GP models and print their (Phi, Q) -- no device is touched (host-side check)
//   problem file (text): T n m p_prior p_unary dt | hA [(4 (T-1) + 1) n n] | hB [(4 (T-1) + 1) n m] | mu0 [T n] | meas [T n]
//                        | kappa [T] | D0 [T n n] | U0 [(T-1) n n]
//   output: "phiq k Phi... Q..." per factor (iterations < 0), else "iter i cost mu..." lines as factorwise_example
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "gvi/factorized_opts_LTV.hpp"

using namespace gvi;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <problem file> <iterations> <output file>\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  const int iters = std::atoi(argv[2]);
  int T = 0, n = 0, m = 0, p_prior = 0, p_unary = 0;
  double dt = 0;
  in >> T >> n >> m >> p_prior >> p_unary >> dt;
  if (!in || T < 2 || n < 2 || n % 2 || m < 1) { std::fprintf(stderr, "bad problem file\n"); return 2; }
  const int K = T - 1, nsys = 4 * K + 1;
  auto read = [&](size_t cnt) { std::vector<double> v(cnt); for (auto& x : v) in >> x; return v; };
  auto read_mats = [&](int count, int r, int c) {
    std::vector<MatrixXd> out;
    for (int i = 0; i < count; ++i) {
      MatrixXd M(r, c);
      for (int a = 0; a < r; ++a) for (int b = 0; b < c; ++b) in >> M(a, b);
      out.push_back(M);
    }
    return out;
  };
  const std::vector<MatrixXd> hA = read_mats(nsys, n, n), hB = read_mats(nsys, n, m);
  const std::vector<double> mu0 = read((size_t)T * n), meas = read((size_t)T * n), kappa = read(T);
  const std::vector<double> D0 = read((size_t)T * n * n), U0 = read((size_t)K * n * n);
  if (!in) { std::fprintf(stderr, "short problem file\n"); return 2; }
  std::vector<VectorXd> target_mean(T, VectorXd::Zero(n));
  const MatrixXd Qc = MatrixXd::Identity(n / 2, n / 2);

  std::FILE* out = std::fopen(argv[3], "w");
  if (!out) { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
  if (iters < 0) {
    for (int k = 0; k < K; ++k) {
      LTV_GP gp{Qc, k, dt, VectorXd::Zero(n), T, hA, hB, target_mean};
      const MatrixXd Phi = gp.Phi(), Q = gp.Q();
      std::fprintf(out, "phiq %d", k);
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) std::fprintf(out, " %.17g", Phi(a, b));
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) std::fprintf(out, " %.17g", Q(a, b));
      std::fprintf(out, "\n");
    }
    std::fclose(out);
    return 0;
  }
  std::vector<std::shared_ptr<GVIFactorizedBase>> factors;
  for (int k = 0; k < K; ++k) {
    LTV_GP gp{Qc, k, dt, VectorXd::Zero(n), T, hA, hB, target_mean};
    factors.emplace_back(new LinearGpPriorGH{2 * n, n, p_prior, cost_linear_gp, gp, T, k, 1.0, 10.0});
  }
  for (int t = 0; t < T; ++t) {
    FixedPriorGP fixed_gp{MatrixXd::Identity(n, n) * (1.0 / kappa[t]), VectorXd(&meas[(size_t)t * n], n)};
    factors.emplace_back(new FixedGpPriorGH{n, n, p_unary, cost_fixed_gp, fixed_gp, T, t, 1.0, 10.0});
  }
  VectorXd init_mu(mu0.data(), T * n);
  SpMat init_prec(T * n, T * n);
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        init_prec.coeffRef(t * n + r, t * n + c) = D0[((size_t)t * n + r) * n + c];
        if (t + 1 < T) {
          init_prec.coeffRef(t * n + r, (t + 1) * n + c) = U0[((size_t)t * n + r) * n + c];
          init_prec.coeffRef((t + 1) * n + c, t * n + r) = U0[((size_t)t * n + r) * n + c];
        }
      }
  NGDGH<GVIFactorizedBase> opt{factors, n, T, iters, 1.0, 10.0};
  opt.set_execution(Execution::DeviceResident);
  opt.set_niter_low_temperature(iters + 1);
  opt.set_step_size_base(0.55);
  opt.set_max_iter_backtrack(10);
  opt.set_initial_values(init_mu, init_prec);
  opt.optimize(false);
  const VIMPResults& r = opt.results();
  for (int it = 0; it < r.recorded(); ++it) {
    std::fprintf(out, "iter %d %.17g", it, r.cost[it]);
    for (double v : r.mean[it]) std::fprintf(out, " %.17g", v);
    std::fprintf(out, "\n");
  }
  const VectorXd mu = opt.mean();
  std::fprintf(out, "iter %d %.17g", r.recorded(), opt.cost_value());
  for (int i = 0; i < mu.size(); ++i) std::fprintf(out, " %.17g", mu(i));
  std::fprintf(out, "\n");
  std::fclose(out);
  return 0;
}
