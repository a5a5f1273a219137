// Constructor calls spelled the way reference-side callers spell them, compiled unchanged against the
// shim.  The statements marked [ref] are the reference's own argument lists:
//   src/1d_example.cpp:56-60                 NGDFactorizedSimpleGH, nine arguments
//   ngd/NGDFactorizedBaseGH.h:37-44          ten arguments, std::optional<std::shared_ptr<QuadratureWeightsMap>>
//   ngd/NGDFactorizedLinearGH.h:27-37        ten arguments
//   gvibase/GVIFactorizedBaseGH.h:35-40      seven arguments
//   proxgd/ProxGVIFactorizedBaseGH.h:24-28   ten arguments
//   quadrature/SparseGaussHermite.h:38-132   the three constructors
//   quadrature/SparseGHQuadratureWeights.h:14-16   DimDegTuple / PointsWeightsTuple / QuadratureWeightsMap
//
//   ref_callsites host <table file>   host only (runs in the CPU suite): the map type, the loader, every factor constructor
//   ref_callsites gpu  <table file>   on the device: a shared map == the built-in table; a perturbed weight moves E_Phi
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "gvi/gvi_host.hpp"

using namespace gvi;

static double cost_function(const VectorXd& x, const NoneType&) {   // src/1d_example.cpp:25-35 in spirit: not a polynomial
  double s = 0.0;
  for (int i = 0; i < x.size(); ++i) s += std::log(1.0 + x(i) * x(i)) + 0.3 * x(i);
  return s;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s host|gpu <table file>\n", argv[0]); return 2; }
  const bool gpu = std::strcmp(argv[1], "gpu") == 0;
  const std::string map_file = argv[2];

  // ---- the table file, written in the reference's format, read back into the reference's map type ----
  const int32_t dims[3] = {4, 2, 1}, degs[3] = {3, 3, 6};
  if (gvi_table_file_write(map_file.c_str(), 3, dims, degs) != GVI_OK) { std::fprintf(stderr, "cannot write %s\n", map_file.c_str()); return 1; }
  QuadratureWeightsMap nodes_weights_map = read_quadrature_weights_map(map_file);
  std::shared_ptr<QuadratureWeightsMap> nodes_weights_map_pointer = std::make_shared<QuadratureWeightsMap>(nodes_weights_map);   // [ref] idiom
  if (nodes_weights_map.size() != 3) return 1;
  DimDegTuple dim_deg = std::make_tuple(4, 3);                       // [ref] quadrature/SparseGaussHermite.h:140-141 (ints -> doubles)
  if (nodes_weights_map.count(dim_deg) == 0) return 1;
  PointsWeightsTuple pts_weights = nodes_weights_map.at(dim_deg);
  MatrixXd zeromeanpts = std::get<0>(pts_weights);
  VectorXd Weights = std::get<1>(pts_weights);
  {
    int64_t N = 0;
    if (gvi_spgh_count(4, 3, &N) != GVI_OK || N != 41 || zeromeanpts.rows() != 41 || zeromeanpts.cols() != 4 || Weights.size() != 41) return 1;
    std::vector<double> Z((size_t)N * 4), w((size_t)N);
    gvi_spgh_nodes(4, 3, N, Z.data(), w.data(), nullptr);
    for (int i = 0; i < 41; ++i) {
      if (w[(size_t)i] != Weights(i)) return 1;
      for (int a = 0; a < 4; ++a) if (Z[(size_t)i * 4 + a] != zeromeanpts(i, a)) return 1;
    }
  }
  bool threw = false;
  try { read_quadrature_weights_map(map_file + ".absent"); } catch (const std::runtime_error&) { threw = true; }   // :58-61
  if (!threw) return 1;

  // ---- factor constructors ----
  int dim_state = 1, num_states = 1, dim_factor = 1, start_index = 0, gh_degree = 6;
  double temperature = 1.0, high_temperature = 10.0;
  NoneType none_type;
  std::shared_ptr<NGDFactorizedSimpleGH> p_opt_fac{new NGDFactorizedSimpleGH(dim_factor, dim_state, gh_degree,                 // [ref] src/1d_example.cpp:56-60
                                                                             cost_function, none_type,
                                                                             num_states, start_index,
                                                                             temperature, high_temperature)};
  std::shared_ptr<NGDFactorizedSimpleGH> p_opt_fac_map{new NGDFactorizedSimpleGH(dim_factor, dim_state, gh_degree,             // [ref] + the tenth argument
                                                                                 cost_function, none_type,
                                                                                 num_states, start_index,
                                                                                 temperature, high_temperature, nodes_weights_map_pointer)};
  std::optional<std::shared_ptr<QuadratureWeightsMap>> weight_sigpts_map_option = nodes_weights_map_pointer;                   // the declared parameter type
  NGDFactorizedBaseGH<NoneType> fac_opt{4, 2, 3, cost_function, none_type, 3, 0, temperature, high_temperature, weight_sigpts_map_option};
  NGDFactorizedBaseGH<NoneType> fac_nullopt{4, 2, 3, cost_function, none_type, 3, 0, temperature, high_temperature, std::nullopt};
  GVIFactorizedBaseGH base_gh{4, 2, 3, 0, 10.0, 100.0, nodes_weights_map_pointer};                                             // gvibase/GVIFactorizedBaseGH.h:35-40
  GVIFactorizedBaseGH base_gh_default{4, 2, 3, 0};
  ProxGVIFactorizedBaseGH<NoneType> prox_fac{4, 2, 3, cost_function, none_type, 3, 0, temperature, high_temperature, nodes_weights_map_pointer};
  const int n_states = 3, dim_conf = 1;
  const double delt_t = 0.1;
  MatrixXd Qc = MatrixXd::Identity(dim_conf, dim_conf) * 0.8, K0_fixed = MatrixXd::Identity(2 * dim_conf, 2 * dim_conf) * 1e-2;
  VectorXd mu_start = VectorXd::Zero(2 * dim_conf);
  std::vector<std::shared_ptr<GVIFactorizedBase>> vec_factors, vec_factors_builtin;
  for (int i = 0; i < n_states; ++i) {
    FixedPriorGP fixed_gp{K0_fixed * (1.0 + i), VectorXd::Constant(2 * dim_conf, 0.2 * i)};
    vec_factors.emplace_back(new FixedGpPriorGH{2 * dim_conf, 2 * dim_conf, 3, cost_fixed_gp, fixed_gp, n_states, i,          // [ref] ngd/NGDFactorizedLinearGH.h:27-37
                                                temperature, high_temperature, nodes_weights_map_pointer});
    vec_factors_builtin.emplace_back(new FixedGpPriorGH{2 * dim_conf, 2 * dim_conf, 3, cost_fixed_gp, fixed_gp, n_states, i,
                                                        temperature, high_temperature});
    if (i > 0) {
      MinimumAccGP lin_gp{Qc, (double)(i - 1), delt_t, mu_start};
      vec_factors.emplace_back(new LinearGpPriorGH{4 * dim_conf, 2 * dim_conf, 3, cost_linear_gp, lin_gp, n_states, i - 1,
                                                   temperature, high_temperature, nodes_weights_map_pointer});
      vec_factors_builtin.emplace_back(new LinearGpPriorGH{4 * dim_conf, 2 * dim_conf, 3, cost_linear_gp, lin_gp, n_states, i - 1,
                                                           temperature, high_temperature});
    }
  }
  if (p_opt_fac->weights_map() || p_opt_fac_map->weights_map() != nodes_weights_map_pointer || fac_nullopt.weights_map() ||
      fac_opt.weights_map() != nodes_weights_map_pointer || base_gh.weights_map() != nodes_weights_map_pointer ||
      base_gh_default.weights_map() || prox_fac.weights_map() != nodes_weights_map_pointer ||
      vec_factors[1]->weights_map() != nodes_weights_map_pointer || vec_factors_builtin[1]->weights_map())
    return 1;
  if (!gpu) { std::printf("ok\n"); return 0; }

  // ---- device: SparseGaussHermite's three constructors ----
  VectorXd mean(4);
  MatrixXd P = MatrixXd::Identity(4, 4) * 0.3;
  for (int i = 0; i < 4; ++i) { mean(i) = 0.1 * (i + 1); for (int j = 0; j < 4; ++j) if (i != j) P(i, j) = 0.02 / (1 + std::abs(i - j)); }
  using GHFunction = std::function<MatrixXd(const VectorXd&)>;
  using GH = SparseGaussHermite<GHFunction>;
  GHFunction func_phi = [&](const VectorXd& x) { return MatrixXd::Constant(1, 1, cost_function(x, none_type)); };
  GH gh_builtin{3, 4, mean, P};
  GH gh_shared{3, 4, mean, P, weight_sigpts_map_option};                                                                        // [ref] ngd/NGDFactorizedBaseGH.h:49
  GH gh_value{3, 4, mean, P, std::optional<QuadratureWeightsMap>{nodes_weights_map}};                                           // :38-77
  GH gh_cref{3, 4, mean, P, nodes_weights_map};                                                                                 // :120-132
  const double e_builtin = gh_builtin.Integrate(func_phi)(0, 0);
  std::printf("gh_builtin %.17g gh_shared %.17g gh_value %.17g gh_cref %.17g\n", e_builtin, gh_shared.Integrate(func_phi)(0, 0),
              gh_value.Integrate(func_phi)(0, 0), gh_cref.Integrate(func_phi)(0, 0));
  GH gh_missing{5, 4, mean, P, weight_sigpts_map_option};           // key (4, 5) absent: prints, integrates over zero rows
  std::printf("gh_missing_rows %d gh_missing_integral %.17g\n", gh_missing.sigmapts().rows(), gh_missing.Integrate(func_phi)(0, 0));

  // ---- device: a factor with the shared map, then with ONE weight of the (4, 3) entry perturbed ----
  const int i_pert = 7;
  const double delta = 1e-3;
  auto perturbed = std::make_shared<QuadratureWeightsMap>(nodes_weights_map);
  std::get<1>(perturbed->at(dim_deg))(i_pert) += delta;
  const DevicePsi hinge_free = DevicePsi::QuadPrior(MatrixXd::Identity(2, 2) * 0.9, MatrixXd::Identity(2, 2) * 3.0);
  auto psi_host = [](const VectorXd& x, const NoneType&) {          // the same quadratic as the device kind: 1/2 |0.9 x1 - x2|^2_{3 I}
    double s = 0.0;
    for (int a = 0; a < 2; ++a) { const double r = 0.9 * x(a) - x(2 + a); s += 1.5 * r * r; }
    return s;
  };
  double e_phi[4];
  const std::shared_ptr<QuadratureWeightsMap> maps[4] = {nullptr, nodes_weights_map_pointer, perturbed, perturbed};
  for (int v = 0; v < 4; ++v) {
    // v = 3: the opaque host psi over the same perturbed map (device expand -> host psi -> device reduction)
    std::optional<std::shared_ptr<QuadratureWeightsMap>> opt_map = maps[v] ? std::optional<std::shared_ptr<QuadratureWeightsMap>>{maps[v]} : std::nullopt;
    NGDFactorizedBaseGH<NoneType> fac{4, 2, 3, psi_host, none_type, 3, 0, temperature, high_temperature, opt_map,
                                      v == 3 ? std::nullopt : std::optional<DevicePsi>{hinge_free}};
    fac.updateGH(mean, P);
    if (v == 3) { fac.calculate_partial_V(); e_phi[v] = 0.0; std::printf("opaque_Vdmu0 %.17g\n", fac.Vdmu()(0)); }
    else { e_phi[v] = fac.E_Phi(); fac.calculate_partial_V(); std::printf("device_Vdmu0_%d %.17g\n", v, fac.Vdmu()(0)); }
  }
  GH gh_pert{3, 4, mean, P, std::optional<std::shared_ptr<QuadratureWeightsMap>>{perturbed}};
  VectorXd x_i(4);
  for (int a = 0; a < 4; ++a) x_i(a) = gh_pert.sigmapts()(i_pert, a);
  std::printf("E_Phi_builtin %.17g E_Phi_shared %.17g E_Phi_perturbed %.17g expected_shift %.17g\n", e_phi[0], e_phi[1], e_phi[2],
              delta * psi_host(x_i, none_type));

  // ---- device: the optimiser over factors that share the map == over factors on the built-in table ----
  double final_cost[2];
  std::vector<double> final_mu[2];
  for (int v = 0; v < 2; ++v) {
    auto& f = v == 0 ? vec_factors_builtin : vec_factors;
    NGDGH<GVIFactorizedBase> opt{f, 2 * dim_conf, n_states, 6, temperature, high_temperature};
    opt.set_niter_low_temperature(10);
    opt.set_step_size_base(0.55);
    VectorXd init_mu = VectorXd::Constant(2 * dim_conf * n_states, 0.1);
    SpMat init_prec(2 * dim_conf * n_states, 2 * dim_conf * n_states);
    for (int i = 0; i < 2 * dim_conf * n_states; ++i) init_prec.coeffRef(i, i) = 50.0;
    opt.set_initial_values(init_mu, init_prec);
    opt.optimize(false);
    final_cost[v] = opt.cost_value();
    const VectorXd mu = opt.mean();
    final_mu[v].assign(mu.data(), mu.data() + mu.size());
  }
  double gap = 0.0;
  for (size_t i = 0; i < final_mu[0].size(); ++i) gap = std::fmax(gap, std::fabs(final_mu[0][i] - final_mu[1][i]));
  std::printf("opt_cost_builtin %.17g opt_cost_shared %.17g opt_mu_gap %.3e\n", final_cost[0], final_cost[1], gap);
  return 0;
}
