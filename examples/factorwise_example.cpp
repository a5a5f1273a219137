// The reference-shaped joint loop on the per-factor operator surface.
//
// A GP-prior chain (BASELINE configs[1] shape: minimum-acceleration priors between consecutive states + one Gaussian
// unary factor per state) is built from the reference's own model classes -- gvi::MinimumAccGP / gvi::FixedPriorGP with
// cost_linear_gp / cost_fixed_gp (gp/minimum_acc_prior.h, gp/fixed_prior.h, gp/cost_functions.h) wrapped in
// LinearGpPriorGH / FixedGpPriorGH (gp/factorized_opts_linear.h) -- and optimised twice with gvi::NGDGH:
//
//   factorwise   Execution::FactorWise: NGDGH::compute_gradients runs the loop of ngd/NGD-GH-impl.h:39-60
//                (calculate_partial_V, local2joint_dmu_insertion, local2joint_dprecision_insertion per factor),
//                cost_value(mean, Precision) sums fact_cost_value(mean, Cov) per factor (gvibase/GVI-GH-impl.h:176-197),
//                update_proposal -> set_mu / set_precision -> update_mu_from_joint / update_precision_from_joint;
//   resident     Execution::DeviceResident: the same iteration through gvi_ngd_* (state never leaves HBM).
//
// Both write "<mode> <iteration> <cost> mu..." lines; tests/test_cpp_shim.py compares them with each other and with
// gvi_ngd_step through the Python binding.  A third block drives single factors directly (stand-alone objects, no
// optimiser): E_Phi / E_xMuPhi / E_xMuxMuTPhi, and an opaque std::function psi against its DevicePsi twin.
//
//   Usage: factorwise_example <problem file> <iterations> <output file> [csv prefix | -] [step base] [max backtrack]
//                             [temperature] [high temperature]
//   (the optional tail drives the backtrack-exhaustion -> switch_to_high_temperature branch of optimize(),
//   gvibase/GVI-GH-impl.h:102-117)
//   With a csv prefix the resident run also writes the reference's nine result files (VIMPResults::save_data,
//   helpers/DataRecorder.h:177-224): <prefix>mean.csv, cov.csv, precision.csv, joint_cov.csv, joint_precision.csv, cost.csv,
//   factor_costs.csv, zk_sdf.csv, Sk_sdf.csv.
//   problem file (text): T n p_prior p_unary dt qc | mu0 [T n] | meas [T n] | kappa [T] | D0 [T n n] | U0 [(T-1) n n]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "gvi/gvi_host.hpp"

using namespace gvi;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <problem file> <iterations> <output file>\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  const int iters = std::atoi(argv[2]);
  int T = 0, n = 0, p_prior = 0, p_unary = 0;
  double dt = 0, qc = 0;
  in >> T >> n >> p_prior >> p_unary >> dt >> qc;
  if (!in || T < 2 || n < 2 || n % 2) { std::fprintf(stderr, "bad problem file\n"); return 2; }
  const int nd = n / 2, K = T - 1;
  auto read = [&](size_t cnt) { std::vector<double> v(cnt); for (auto& x : v) in >> x; return v; };
  const std::vector<double> mu0 = read((size_t)T * n), meas = read((size_t)T * n), kappa = read(T);
  const std::vector<double> D0 = read((size_t)T * n * n), U0 = read((size_t)K * n * n);
  if (!in) { std::fprintf(stderr, "short problem file\n"); return 2; }

  const std::string csv_prefix = argc > 4 && std::string(argv[4]) != "-" ? argv[4] : "";
  const double step_base = argc > 5 ? std::atof(argv[5]) : 0.55;
  const int max_backtrack = argc > 6 ? std::atoi(argv[6]) : 10;
  const double temperature = argc > 7 ? std::atof(argv[7]) : 1.0, high_temperature = argc > 8 ? std::atof(argv[8]) : 10.0;
  auto make_factors = [&]() {
    std::vector<std::shared_ptr<GVIFactorizedBase>> f;
    const MatrixXd Qc = MatrixXd::Identity(nd, nd) * qc;
    for (int k = 0; k < K; ++k) {
      MinimumAccGP lin_gp{Qc, (double)k, dt, VectorXd::Zero(n)};
      f.emplace_back(new LinearGpPriorGH{2 * n, n, p_prior, cost_linear_gp, lin_gp, T, k, temperature, high_temperature});
    }
    for (int t = 0; t < T; ++t) {
      FixedPriorGP fixed_gp{MatrixXd::Identity(n, n) * (1.0 / kappa[t]), VectorXd(&meas[(size_t)t * n], n)};
      f.emplace_back(new FixedGpPriorGH{n, n, p_unary, cost_fixed_gp, fixed_gp, T, t, temperature, high_temperature});
    }
    return f;
  };
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

  std::FILE* out = std::fopen(argv[3], "w");
  if (!out) { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
  for (int mode = 0; mode < 2; ++mode) {
    auto factors = make_factors();
    NGDGH<GVIFactorizedBase> opt{factors, n, T, iters, temperature, high_temperature};
    opt.set_execution(mode == 0 ? Execution::FactorWise : Execution::DeviceResident);
    opt.set_niter_low_temperature(iters + 1);
    opt.set_step_size_base(step_base);
    opt.set_max_iter_backtrack(max_backtrack);
    if (mode == 1 && !csv_prefix.empty()) opt.update_file_names(csv_prefix);
    opt.set_initial_values(init_mu, init_prec);
    opt.optimize(false);
    const char* name = mode == 0 ? "factorwise" : "resident";
    const VIMPResults& r = opt.results();
    for (int it = 0; it < r.recorded(); ++it) {
      std::fprintf(out, "%s %d %.17g", name, it, r.cost[it]);
      for (double v : r.mean[it]) std::fprintf(out, " %.17g", v);
      std::fprintf(out, "\n");
    }
    const VectorXd mu = opt.mean();
    std::fprintf(out, "%s %d %.17g", name, r.recorded(), opt.cost_value());
    for (int i = 0; i < mu.size(); ++i) std::fprintf(out, " %.17g", mu(i));
    std::fprintf(out, "\n");
    std::fprintf(out, "%s_final_temperature %.17g\n", name, factors[0]->temperature());
    if (mode == 0) {
      // one pass over K + T factors costs two device calls (one per homogeneous set), not K + T
      std::fprintf(out, "device_calls %ld factors %d\n", opt.factorwise_device_calls(), K + T);
      // the joint (Vdmu, Vddmu) of the last compute_gradients against the per-factor pieces
      auto g = opt.compute_gradients();
      (void)g;
      const VectorXd Vd = opt.Vdmu();
      double worst = 0.0;
      VectorXd sum = VectorXd::Zero(T * n);
      for (auto& f : factors) sum += f->local2joint_dmu_insertion();
      for (int i = 0; i < T * n; ++i) worst = std::fmax(worst, std::fabs(sum(i) - Vd(i)));
      std::fprintf(out, "vdmu_insertion_gap %.3e\n", worst);
    }
  }

  // ---- single factors driven directly (no optimiser): private device set on first use ----
  {
    MinimumAccGP lin_gp{MatrixXd::Identity(nd, nd) * qc, 0.0, dt, VectorXd::Zero(n)};
    LinearGpPriorGH fac{2 * n, n, p_prior, cost_linear_gp, lin_gp, T, 0, temperature, high_temperature};
    // the same psi as an opaque host function: device expand -> host psi -> device reduction
    NGDFactorizedBaseGH<MinimumAccGP> opaque{2 * n, n, p_prior, cost_linear_gp, lin_gp, T, 0, temperature, high_temperature};
    VectorXd m(2 * n);
    MatrixXd P = MatrixXd::Identity(2 * n, 2 * n) * 0.3;
    for (int i = 0; i < 2 * n; ++i) { m(i) = 0.1 * (i + 1); for (int j = 0; j < 2 * n; ++j) if (i != j) P(i, j) = 0.02 / (1 + std::abs(i - j)); }
    fac.updateGH(m, P);
    opaque.updateGH(m, P);
    fac.calculate_partial_V();
    opaque.calculate_partial_V();
    double gap = 0.0;
    const VectorXd a = fac.Vdmu(), b = opaque.Vdmu();
    const MatrixXd A = fac.Vddmu(), B = opaque.Vddmu();
    for (int i = 0; i < 2 * n; ++i) {
      gap = std::fmax(gap, std::fabs(a(i) - b(i)));
      for (int j = 0; j < 2 * n; ++j) gap = std::fmax(gap, std::fabs(A(i, j) - B(i, j)));
    }
    std::fprintf(out, "opaque_vs_device_psi_gap %.3e\n", gap);
    // raw integrals: Vdmu = Lam E[(x-mu)psi] / T  (ngd/NGDFactorizedBaseGH.h:62-64)
    const MatrixXd Ex = fac.E_xMuPhi();
    const MatrixXd Lam = fac.precision();
    double gap2 = 0.0;
    for (int i = 0; i < 2 * n; ++i) {
      double s = 0.0;
      for (int j = 0; j < 2 * n; ++j) s += Lam(i, j) * Ex(j, 0);
      gap2 = std::fmax(gap2, std::fabs(s / fac.temperature() - a(i)));
    }
    std::fprintf(out, "raw_integral_gap %.3e E_Phi %.17g\n", gap2, fac.E_Phi());
  }
  std::fclose(out);
  return 0;
}
