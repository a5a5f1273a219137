// gvi_host.hpp -- C++17 host shim over the C ABI (include/gvi_hip.h).
//
// Mirrors the reference's operator surface for the NGD Gauss-Hermite path so that a driver written
// against hzyu17/GaussianVI compiles against this header with the include lines changed:
//
//   gvi::SparseGaussHermite<Function>      quadrature/SparseGaussHermite.h:28-277
//   gvi::GVIFactorizedBase                 gvibase/GVIFactorizedBase.h:36-248
//   gvi::NGDFactorizedBaseGH<CostClass>    ngd/NGDFactorizedBaseGH.h:25-133   (NGDFactorizedSimpleGH alias)
//   gvi::GVIGH<Factor> / gvi::NGDGH<Factor> gvibase/GVI-GH.h:27-414, ngd/NGD-GH.h:25-95
//
// The reference is header-only on Eigen; Eigen is not part of this image, so the shim carries two
// tiny dense types (VectorXd, MatrixXd: row-major, the ABI's layout) and a coefficient-map SpMat with
// the handful of members the drivers use.  All arithmetic of the hot path happens on the device
// through the C ABI; there is no CPU fallback -- a missing device makes the constructor throw.
//
// psi: the reference takes an opaque std::function (ngd/NGDFactorizedBaseGH.h:30).  Two routes:
//   * opaque host function  -> device expand, host psi, device reduction (gvi_expand /
//     gvi_moments_from_psi), one call per factor: the generic surface, not the fast path;
//   * DevicePsi descriptor  -> the factor joins a homogeneous device set and the optimiser runs the
//     device-resident iteration (gvi_ngd_*), one launch sequence per pass for ALL factors.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../gvi_hip.h"

namespace gvi {

// ------------------------------------------------------------------------------------------------
// minimal dense types (row-major)
// ------------------------------------------------------------------------------------------------
class VectorXd {
 public:
  VectorXd() = default;
  explicit VectorXd(int n) : v_(n, 0.0) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  static VectorXd Constant(int n, double c) { VectorXd r(n); for (auto& x : r.v_) x = c; return r; }
  int size() const { return (int)v_.size(); }
  int rows() const { return size(); }
  double& operator()(int i) { return v_[i]; }
  double operator()(int i) const { return v_[i]; }
  double* data() { return v_.data(); }
  const double* data() const { return v_.data(); }
  void setZero() { for (auto& x : v_) x = 0.0; }
 private:
  std::vector<double> v_;
};

class MatrixXd {
 public:
  MatrixXd() = default;
  MatrixXd(int r, int c) : r_(r), c_(c), v_((size_t)r * c, 0.0) {}
  static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
  static MatrixXd Constant(int r, int c, double x) { MatrixXd m(r, c); for (auto& y : m.v_) y = x; return m; }
  static MatrixXd Identity(int r, int c) { MatrixXd m(r, c); for (int i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0; return m; }
  int rows() const { return r_; }
  int cols() const { return c_; }
  double& operator()(int i, int j) { return v_[(size_t)i * c_ + j]; }
  double operator()(int i, int j) const { return v_[(size_t)i * c_ + j]; }
  double* data() { return v_.data(); }
  const double* data() const { return v_.data(); }
  void setZero() { for (auto& x : v_) x = 0.0; }
 private:
  int r_ = 0, c_ = 0;
  std::vector<double> v_;
};

// Sparse joint matrix as a coefficient map: only what the drivers touch (coeffRef / coeff / size).
class SpMat {
 public:
  SpMat() = default;
  SpMat(int r, int c) : r_(r), c_(c) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  void setZero() { m_.clear(); }
  double& coeffRef(int i, int j) { return m_[{i, j}]; }
  double coeff(int i, int j) const { auto it = m_.find({i, j}); return it == m_.end() ? 0.0 : it->second; }
 private:
  int r_ = 0, c_ = 0;
  std::map<std::pair<int, int>, double> m_;
};

struct NoneType {};

class GviError : public std::runtime_error {
 public:
  GviError(int status, const std::string& m) : std::runtime_error("gvi status " + std::to_string(status) + ": " + m), status(status) {}
  int status;
};

// One device context shared by the objects of a problem.
class Device {
 public:
  explicit Device(int device = 0) {
    gvi_status s = gvi_ctx_create(device, GVI_F64, &ctx_);
    if (s != GVI_OK) throw GviError(s, gvi_last_error(nullptr));
  }
  ~Device() { gvi_ctx_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  gvi_ctx* get() const { return ctx_; }
  void check(gvi_status s) const { if (s != GVI_OK) throw GviError(s, gvi_last_error(ctx_)); }
 private:
  gvi_ctx* ctx_ = nullptr;
};

// Signed-distance fields and the arm model of the obstacle costs (helpers/CudaOperation.h: PlanarSDF :21-131,
// SignedDistanceField :133-322, ForwardKinematics :325-399).  Shared by the factors that reference them.
struct PlanarSDF {
  double origin_x = 0, origin_y = 0, cell_size = 1;
  MatrixXd field;                        // field(r, c) at (origin_x + c cell, origin_y + r cell)
};
struct SignedDistanceField {
  double origin[3] = {0, 0, 0}, cell_size = 1;
  int rows = 0, cols = 0, nz = 0;
  std::vector<double> data;              // data[r + c rows + z rows cols]
};
struct ArmModel {
  std::vector<double> a, alpha, d, theta_bias;      // DH parameters [ndof]
  std::vector<int32_t> frames;                      // sphere -> frame (non-decreasing)
  std::vector<double> centers, radii;               // [ns][3], [ns]
};

// Device-evaluable psi (include/gvi_hip.h, GVI_PSI_*): kind + the factor's parameter block (+ shared field / arm).
struct DevicePsi {
  int kind = GVI_PSI_HOST_CALLBACK;
  std::vector<double> params;
  std::shared_ptr<const PlanarSDF> sdf2d;
  std::shared_ptr<const SignedDistanceField> sdf3d;
  std::shared_ptr<const ArmModel> arm;
  // CudaOperation_PlanarPR / _Quad / _3dpR / _3dArm cost_obstacle* (helpers/CudaOperation.h:491-523, 565-606, 650-683, 752-771)
  static DevicePsi HingeSdf2D(double sigma, double epsilon, double radius, std::shared_ptr<const PlanarSDF> sdf) {
    DevicePsi p{GVI_PSI_HINGE_SDF_2D, {sigma, epsilon, radius}};
    p.sdf2d = std::move(sdf);
    return p;
  }
  static DevicePsi HingeSdf2DBody(double sigma, double epsilon, double radius, double slope, int n_balls, double L,
                                  std::shared_ptr<const PlanarSDF> sdf) {
    DevicePsi p{GVI_PSI_HINGE_SDF_2D_BODY, {sigma, epsilon, radius, slope, (double)n_balls, L}};
    p.sdf2d = std::move(sdf);
    return p;
  }
  static DevicePsi HingeSdf3D(double sigma, double epsilon, double radius, std::shared_ptr<const SignedDistanceField> sdf) {
    DevicePsi p{GVI_PSI_HINGE_SDF_3D, {sigma, epsilon, radius}};
    p.sdf3d = std::move(sdf);
    return p;
  }
  static DevicePsi HingeSdf3DArm(double sigma, double epsilon, std::shared_ptr<const SignedDistanceField> sdf,
                                 std::shared_ptr<const ArmModel> arm_model) {
    DevicePsi p{GVI_PSI_HINGE_SDF_3D_ARM, {sigma, epsilon}};
    p.sdf3d = std::move(sdf);
    p.arm = std::move(arm_model);
    return p;
  }
  static DevicePsi Range1D(double y, double mu_p, double fb, double sig_r_sq, double sig_p_sq) {
    return {GVI_PSI_RANGE_1D, {y, mu_p, fb, sig_r_sq, sig_p_sq}};
  }
  static DevicePsi QuadPrior(const MatrixXd& Phi, const MatrixXd& Qinv) {
    DevicePsi p{GVI_PSI_QUAD_PRIOR, {}};
    p.params.insert(p.params.end(), Phi.data(), Phi.data() + Phi.rows() * Phi.cols());
    p.params.insert(p.params.end(), Qinv.data(), Qinv.data() + Qinv.rows() * Qinv.cols());
    return p;
  }
  static DevicePsi FixedPrior(const VectorXd& mu0, const MatrixXd& Kinv) {
    DevicePsi p{GVI_PSI_FIXED_PRIOR, {}};
    p.params.insert(p.params.end(), mu0.data(), mu0.data() + mu0.size());
    p.params.insert(p.params.end(), Kinv.data(), Kinv.data() + Kinv.rows() * Kinv.cols());
    return p;
  }
};

// ------------------------------------------------------------------------------------------------
// SparseGaussHermite (quadrature/SparseGaussHermite.h): table lookup, symmetric-sqrt expand on the
// device, weighted reduction of an arbitrary host function.
// ------------------------------------------------------------------------------------------------
template <typename Function = std::function<MatrixXd(const VectorXd&)>>
class SparseGaussHermite {
 public:
  SparseGaussHermite(int deg, int dim, const VectorXd& mean, const MatrixXd& P, std::shared_ptr<Device> dev = nullptr)
      : _deg(deg), _dim(dim), _mean(mean), _P(P), _dev(dev ? dev : std::make_shared<Device>()) {
    computeSigmaPtsWeights();
  }
  void computeSigmaPtsWeights() {   // :138-166
    int64_t N = 0;
    gvi_status s = gvi_spgh_count(_dim, _deg, &N);
    if (s != GVI_OK) {                                    // reference: prints "key does not exist"
      std::printf("(dimension, degree) (%d, %d) key does not exist in the GH weight map.\n", _dim, _deg);
      return;
    }
    _zeromeanpts = MatrixXd((int)N, _dim);
    _Weights = VectorXd((int)N);
    _dev->check(gvi_spgh_nodes(_dim, _deg, N, _zeromeanpts.data(), _Weights.data(), nullptr));
    _dev->check(gvi_chain_set(_dev->get(), 1, _dim));
    const int32_t start = 0;
    _dev->check(gvi_factors_add(_dev->get(), 1, _dim, _deg, &start, GVI_PSI_HOST_CALLBACK, nullptr, 0, nullptr, &_set));
    update_sigmapoints();
  }
  MatrixXd Integrate(const Function& function) {      // :197-221
    MatrixXd res = function(_mean);
    res.setZero();
    VectorXd pt(_dim);
    for (int i = 0; i < _sigmapts.rows(); ++i) {
      for (int a = 0; a < _dim; ++a) pt(a) = _sigmapts(i, a);
      MatrixXd f = function(pt);
      for (int r = 0; r < res.rows(); ++r)
        for (int c = 0; c < res.cols(); ++c) res(r, c) += f(r, c) * _Weights(i);
    }
    return res;
  }
  void update_mean(const VectorXd& mean) { _mean = mean; }
  void update_P(const MatrixXd& P) { _P = P; }
  void update_sigmapoints() {                          // :231-243, on the device
    const int N = _Weights.size();
    std::vector<double> X((size_t)_dim * N);
    _dev->check(gvi_expand(_dev->get(), _set, _mean.data(), _P.data(), X.data()));
    _sigmapts = MatrixXd(N, _dim);
    for (int a = 0; a < _dim; ++a)
      for (int i = 0; i < N; ++i) _sigmapts(i, a) = X[(size_t)a * N + i];
  }
  VectorXd weights() const { return _Weights; }
  MatrixXd sigmapts() const { return _sigmapts; }
  VectorXd mean() const { return _mean; }
 protected:
  int _deg, _dim, _set = -1;
  VectorXd _mean;
  MatrixXd _P;
  VectorXd _Weights;
  MatrixXd _sigmapts, _zeromeanpts;
  std::shared_ptr<Device> _dev;
};

// ------------------------------------------------------------------------------------------------
// Factor operator surface (gvibase/GVIFactorizedBase.h:36-248; ngd/NGDFactorizedBaseGH.h:25-133).
// A factor is a handle: its numbers live in the optimiser's device sets; the accessors read back.
// ------------------------------------------------------------------------------------------------
class GVIFactorizedBase {
 public:
  virtual ~GVIFactorizedBase() {}
  GVIFactorizedBase(int dimension, int state_dim, int num_states, int start_index, double temperature = 10.0,
                    double high_temperature = 100.0)
      : _dim(dimension), _state_dim(state_dim), _num_states(num_states), _start_index(start_index),
        _joint_size(state_dim * num_states), _mu(dimension), _precision(MatrixXd::Identity(dimension, dimension)),
        _covariance(MatrixXd::Identity(dimension, dimension)), _temperature(temperature),
        _high_temperature(high_temperature), _Vdmu(dimension), _Vddmu(dimension, dimension) {}
  inline void set_step_size(double step_size) { _step_size = step_size; }
  inline VectorXd mean() const { return _mu; }
  inline MatrixXd precision() const { return _precision; }
  inline MatrixXd covariance() const { return _covariance; }
  void factor_switch_to_high_temperature() { _temperature = _high_temperature; }
  double temperature() { return _temperature; }
  int _dim, _state_dim, _num_states, _start_index, _joint_size;
  VectorXd _mu;
 protected:
  MatrixXd _precision, _covariance;
  double _step_size = 0.9, _temperature, _high_temperature;
  VectorXd _Vdmu;
  MatrixXd _Vddmu;
};

template <typename CostClass = NoneType>
class NGDFactorizedBaseGH : public GVIFactorizedBase {
 public:
  using Function = std::function<double(const VectorXd&, const CostClass&)>;
  // Reference signature (ngd/NGDFactorizedBaseGH.h:37-44) + an optional device psi descriptor.
  NGDFactorizedBaseGH(int dimension, int state_dim, int gh_degree, const Function& function, const CostClass& cost_class,
                      int num_states, int start_index, double temperature = 1.0, double high_temperature = 10.0,
                      std::optional<DevicePsi> device_psi = std::nullopt)
      : GVIFactorizedBase(dimension, state_dim, num_states, start_index, temperature, high_temperature),
        _gh_degree(gh_degree), _function(function), _cost_class(cost_class),
        _psi(device_psi ? *device_psi : DevicePsi{}) {}
  double psi(const VectorXd& x) const { return _function(x, _cost_class); }
  int gh_degree() const { return _gh_degree; }
  const DevicePsi& device_psi() const { return _psi; }
  VectorXd Vdmu() const { return _Vdmu; }
  MatrixXd Vddmu() const { return _Vddmu; }
 private:
  int _gh_degree;
  Function _function;
  CostClass _cost_class;
  DevicePsi _psi;
};
using NGDFactorizedSimpleGH = NGDFactorizedBaseGH<NoneType>;   // ngd/NGDFactorizedSimpleGH.h

// ------------------------------------------------------------------------------------------------
// Joint optimiser (gvibase/GVI-GH.h, gvibase/GVI-GH-impl.h, ngd/NGD-GH.h, ngd/NGD-GH-impl.h)
// ------------------------------------------------------------------------------------------------
struct VIMPResults {   // helpers/DataRecorder.h:25-225, the columns the 1-D example writes
  std::vector<std::vector<double>> mean, precision, cov, factor_costs;
  std::vector<double> cost;
};

template <typename Factor>
class GVIGH {
 public:
  GVIGH(const std::vector<std::shared_ptr<Factor>>& vec_fact_optimizers, int dim_state, int num_states,
        int niterations = 5, double temperature = 1.0, double high_temperature = 100.0, int device = 0)
      : _dim_state(dim_state), _num_states(num_states), _dim(dim_state * num_states), _niters(niterations),
        _temperature(temperature), _high_temperature(high_temperature), _vec_factors(vec_fact_optimizers),
        _dev(std::make_shared<Device>(device)) {
    build_sets();
  }
  virtual ~GVIGH() {}

  // setters (gvibase/GVI-GH.h:168-248)
  inline void set_step_size_base(double v) { _step_size_base = v; }
  inline void set_max_iter_backtrack(double v) { _niters_backtrack = (int)v; }
  inline void set_niter_low_temperature(int v) { _niters_lowtemp = v; }
  inline void set_stop_err(double v) { _stop_err = v; }
  inline void set_temperature(double t) { _temperature = t; }
  inline void set_high_temperature(double t) { _high_temperature = t; }
  inline void update_file_names(const std::string& prefix) { _prefix = prefix; }

  inline void set_initial_values(const VectorXd& init_mean, const SpMat& init_precision) {
    const int T = _num_states, n = _dim_state;
    std::vector<double> D((size_t)T * n * n, 0.0), U((size_t)(T > 1 ? T - 1 : 0) * n * n, 0.0);
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
          D[((size_t)t * n + r) * n + c] = init_precision.coeff(t * n + r, t * n + c);
          if (t + 1 < T) U[((size_t)t * n + r) * n + c] = init_precision.coeff(t * n + r, (t + 1) * n + c);
        }
    _dev->check(gvi_ngd_init(_dev->get(), init_mean.data(), D.data(), U.data()));
    pull_state();
  }

  inline VectorXd mean() const { return _mu; }
  // joint precision / covariance as block-tridiagonal coefficient maps
  inline SpMat precision() const { return to_spmat(_D, _U); }
  inline SpMat covariance() const { return to_spmat(_SigD, _SigU); }

  void switch_to_high_temperature() {   // GVI-GH-impl.h:19-26
    for (auto& f : _vec_factors) f->factor_switch_to_high_temperature();
    _temperature = _high_temperature;
    push_temperatures();
  }

  virtual double cost_value() {          // cost_value() at the current proposal
    double c = 0.0;
    _dev->check(gvi_ngd_cost(_dev->get(), &c));
    return c;
  }
  virtual VectorXd factor_cost_vector() {   // GVI-GH-impl.h:147-170
    VectorXd out((int)_vec_factors.size());
    for (size_t s = 0; s < _sets.size(); ++s) {
      std::vector<double> c(_sets[s].members.size());
      _dev->check(gvi_ngd_factor_costs(_dev->get(), (int)s, c.data()));
      for (size_t k = 0; k < c.size(); ++k) out(_sets[s].members[k]) = c[k];
    }
    return out;
  }

  // GVIGH::time_test of the reference's device variant (gvibase/GVI-GH-Cuda-impl.h:463-527): _niters + 1 timed
  // evaluations of the factor-cost vector at the current proposal (marginals + one cost pass over every factor),
  // the first discarded; same "% ..." lines.  Returns {average, min, max} in ms.
  std::tuple<double, double, double> time_test(bool print = true) {
    std::vector<double> times;
    for (int i = 0; i < _niters + 1; ++i) {
      // a fresh evaluation each round: re-setting the state drops the device-side cache of the cost
      _dev->check(gvi_ngd_init(_dev->get(), _mu.data(), _D.data(), _U.data()));
      const auto t0 = std::chrono::steady_clock::now();
      (void)factor_cost_vector();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (i != 0) times.push_back(ms);
    }
    double avg = 0.0;
    for (double v : times) avg += v;
    avg /= (double)times.size();
    const double mn = *std::min_element(times.begin(), times.end()), mx = *std::max_element(times.begin(), times.end());
    if (print) {
      std::printf("%% %zu\n", _vec_factors.size());
      std::printf("%% GPU average: %g ms\n%% GPU min: %g ms\n%% GPU max: %g ms\n", avg, mn, mx);
    }
    return {avg, mn, mx};
  }

  // GVIGH::optimize with backtracking (gvibase/GVI-GH-impl.h:33-124)
  virtual void optimize(std::optional<bool> verbose = std::nullopt) {
    const bool is_verbose = verbose.value_or(true);
    bool is_lowtemp = true, converged = false;
    for (int i_iter = 0; i_iter < _niters; i_iter++) {
      if (converged) break;
      if (i_iter == _niters_lowtemp && is_lowtemp) {
        if (is_verbose) std::printf("Switching to high temperature..\n");
        switch_to_high_temperature();
        is_lowtemp = false;
      }
      const double cost_iter = cost_value();
      if (is_verbose) std::printf("========= iteration %d ========= \n--- cost_iter ---\n%.15g\n", i_iter, cost_iter);
      VectorXd fact_costs = factor_cost_vector();
      record(cost_iter, fact_costs);
      _dev->check(gvi_ngd_gradients(_dev->get()));
      int cnt = 0;
      double step_size = _step_size_base;
      while (true) {
        step_size = step_size * 0.75;
        double new_cost = 0.0;
        _dev->check(gvi_ngd_trial(_dev->get(), step_size, &new_cost));
        if (new_cost < cost_iter) {
          _dev->check(gvi_ngd_accept(_dev->get()));
          pull_state();
          break;
        } else {
          cnt += 1;
        }
        if (cnt > _niters_backtrack) {
          if (is_verbose) std::printf("Reached the maximum backtracking steps.\n");
          if (is_lowtemp) { switch_to_high_temperature(); is_lowtemp = false; }
          else converged = true;
          break;
        }
      }
    }
    if (!_prefix.empty()) save_data(is_verbose);
  }

  // 1-D cost map (gvibase/GVI-GH.h:385-412)
  MatrixXd cost_map(double x_start, double x_end, double y_start, double y_end, int nmesh) {
    const double res_x = (x_end - x_start) / nmesh, res_y = (y_end - y_start) / nmesh;
    MatrixXd Z = MatrixXd::Zero(nmesh, nmesh);
    for (int i = 0; i < nmesh; i++)
      for (int j = 0; j < nmesh; j++) {
        const double m = x_start + i * res_x, p = y_start + j * res_y;
        _dev->check(gvi_ngd_init(_dev->get(), &m, &p, nullptr));
        Z(j, i) = cost_value();
      }
    return Z;
  }
  void save_costmap(const std::string& filename = "costmap.csv") {
    MatrixXd m = cost_map(18, 25, 0.05, 1, 40);
    std::ofstream f(filename);
    f.precision(15);
    for (int r = 0; r < m.rows(); ++r)
      for (int c = 0; c < m.cols(); ++c) f << m(r, c) << (c + 1 < m.cols() ? ", " : "\n");
  }

  void save_data(bool verbose = true) {     // VIMPResults::save_data layout (helpers/DataRecorder.h:177-224)
    auto dump = [&](const std::string& name, const std::vector<std::vector<double>>& rows) {
      std::ofstream f(_prefix + name);
      f.precision(15);
      if (rows.empty()) return;
      for (size_t r = 0; r < rows[0].size(); ++r)            // one column per iteration
        for (size_t it = 0; it < rows.size(); ++it) f << rows[it][r] << (it + 1 < rows.size() ? ", " : "\n");
    };
    dump("mean.csv", _rec.mean);
    dump("precision.csv", _rec.precision);
    dump("cov.csv", _rec.cov);
    dump("factor_costs.csv", _rec.factor_costs);
    std::ofstream f(_prefix + "cost.csv");
    f.precision(15);
    for (double c : _rec.cost) f << c << "\n";
    if (verbose) std::printf("=========== Saving Data ===========\n");
  }
  const VIMPResults& results() const { return _rec; }

 protected:
  struct Set { int d, p, kind; std::vector<int> members; };

  void build_sets() {
    // homogeneous device sets in first-appearance order (d, GH degree, psi kind)
    for (size_t i = 0; i < _vec_factors.size(); ++i) {
      auto& f = _vec_factors[i];
      const int kind = f->device_psi().kind;
      if (kind == GVI_PSI_HOST_CALLBACK)
        throw GviError(GVI_ERR_UNSUPPORTED,
                       "GVIGH: the device-resident optimiser needs a DevicePsi per factor; opaque host psi is "
                       "served by SparseGaussHermite / gvi_expand + gvi_moments_from_psi");
      const DevicePsi& dp = f->device_psi();
      size_t s = 0;
      for (; s < _sets.size(); ++s) {
        const DevicePsi& q = _vec_factors[_sets[s].members[0]]->device_psi();
        if (_sets[s].d == f->_dim && _sets[s].p == f->gh_degree() && _sets[s].kind == kind && q.sdf2d == dp.sdf2d &&
            q.sdf3d == dp.sdf3d && q.arm == dp.arm)
          break;
      }
      if (s == _sets.size()) _sets.push_back({f->_dim, f->gh_degree(), kind, {}});
      _sets[s].members.push_back((int)i);
    }
    _dev->check(gvi_chain_set(_dev->get(), _num_states, _dim_state));
    for (auto& st : _sets) {
      const size_t K = st.members.size();
      std::vector<int32_t> start(K);
      std::vector<double> temp(K), params;
      size_t per = _vec_factors[st.members[0]]->device_psi().params.size();
      for (size_t k = 0; k < K; ++k) {
        auto& f = _vec_factors[st.members[k]];
        start[k] = f->_start_index;
        temp[k] = f->temperature();
        params.insert(params.end(), f->device_psi().params.begin(), f->device_psi().params.end());
      }
      int id = -1;
      _dev->check(gvi_factors_add(_dev->get(), (int)K, st.d, st.p, start.data(), st.kind, params.data(), (int64_t)per,
                                  temp.data(), &id));
      const DevicePsi& dp = _vec_factors[st.members[0]]->device_psi();
      if (dp.sdf2d) {                                   // column-major rows x cols, like Eigen's MatrixXd
        const MatrixXd& fld = dp.sdf2d->field;
        std::vector<double> cm((size_t)fld.rows() * fld.cols());
        for (int c = 0; c < fld.cols(); ++c)
          for (int r = 0; r < fld.rows(); ++r) cm[(size_t)c * fld.rows() + r] = fld(r, c);
        _dev->check(gvi_factors_set_sdf2d(_dev->get(), id, dp.sdf2d->origin_x, dp.sdf2d->origin_y, dp.sdf2d->cell_size,
                                          fld.rows(), fld.cols(), cm.data()));
      }
      if (dp.sdf3d)
        _dev->check(gvi_factors_set_sdf3d(_dev->get(), id, dp.sdf3d->origin, dp.sdf3d->cell_size, dp.sdf3d->rows,
                                          dp.sdf3d->cols, dp.sdf3d->nz, dp.sdf3d->data.data()));
      if (dp.arm)
        _dev->check(gvi_factors_set_arm(_dev->get(), id, (int)dp.arm->a.size(), dp.arm->a.data(), dp.arm->alpha.data(),
                                        dp.arm->d.data(), dp.arm->theta_bias.data(), (int)dp.arm->frames.size(),
                                        dp.arm->frames.data(), dp.arm->centers.data(), dp.arm->radii.data()));
    }
  }
  void push_temperatures() {
    for (size_t s = 0; s < _sets.size(); ++s) {
      std::vector<double> temp;
      for (int i : _sets[s].members) temp.push_back(_vec_factors[i]->temperature());
      _dev->check(gvi_factors_set_temperature(_dev->get(), (int)s, temp.data()));
    }
  }
  void pull_state() {
    const int T = _num_states, n = _dim_state;
    _mu = VectorXd(T * n);
    _D.assign((size_t)T * n * n, 0.0); _SigD = _D;
    _U.assign((size_t)(T > 1 ? T - 1 : 0) * n * n, 0.0); _SigU = _U;
    _dev->check(gvi_ngd_get_state(_dev->get(), _mu.data(), _D.data(), _U.data(), _SigD.data(), _SigU.data()));
  }
  SpMat to_spmat(const std::vector<double>& D, const std::vector<double>& U) const {
    const int T = _num_states, n = _dim_state;
    SpMat m(_dim, _dim);
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
          m.coeffRef(t * n + r, t * n + c) = D[((size_t)t * n + r) * n + c];
          if (t + 1 < T) {
            m.coeffRef(t * n + r, (t + 1) * n + c) = U[((size_t)t * n + r) * n + c];
            m.coeffRef((t + 1) * n + c, t * n + r) = U[((size_t)t * n + r) * n + c];
          }
        }
    return m;
  }
  void record(double cost, const VectorXd& fact_costs) {   // helpers/DataRecorder.h:65-118
    _rec.mean.emplace_back(_mu.data(), _mu.data() + _mu.size());
    _rec.precision.push_back(_D);
    _rec.cov.push_back(_SigD);
    _rec.cost.push_back(cost);
    _rec.factor_costs.emplace_back(fact_costs.data(), fact_costs.data() + fact_costs.size());
  }

  int _dim_state, _num_states, _dim, _niters;
  int _niters_lowtemp = 10, _niters_backtrack = 10;      // gvibase/GVI-GH.h:51-53
  double _stop_err = 1e-5, _temperature, _high_temperature;
  double _step_size = 0.9, _step_size_base = 0.55;       // gvibase/GVI-GH.h:92-93
  std::vector<std::shared_ptr<Factor>> _vec_factors;
  std::shared_ptr<Device> _dev;
  std::vector<Set> _sets;
  VectorXd _mu;
  std::vector<double> _D, _U, _SigD, _SigU;
  VIMPResults _rec;
  std::string _prefix;
};

// NGDGH (ngd/NGD-GH.h:25-95): the natural-gradient update law is what the device iteration runs;
// the virtuals of the reference are exposed one to one.
template <typename Factor>
class NGDGH : public GVIGH<Factor> {
  using Base = GVIGH<Factor>;
 public:
  using Base::Base;
  // compute_gradients (ngd/NGD-GH-impl.h:21-63): returns (dmu, dprecision blocks)
  std::tuple<VectorXd, SpMat> compute_gradients(std::optional<double> = std::nullopt) {
    this->_dev->check(gvi_ngd_gradients(this->_dev->get()));
    const int T = this->_num_states, n = this->_dim_state;
    VectorXd dmu(T * n);
    std::vector<double> dD((size_t)T * n * n), dU((size_t)(T > 1 ? T - 1 : 0) * n * n);
    this->_dev->check(gvi_ngd_get_gradients(this->_dev->get(), dmu.data(), dD.data(), dU.data(), nullptr, nullptr, nullptr));
    return std::make_tuple(dmu, this->to_spmat(dD, dU));
  }
  // onestep_linesearch (ngd/NGD-GH-impl.h:130-148): cost of the trial at step_size (kept on device)
  double onestep_linesearch(double step_size) {
    double c = 0.0;
    this->_dev->check(gvi_ngd_trial(this->_dev->get(), step_size, &c));
    return c;
  }
  // update_proposal (ngd/NGD-GH-impl.h:151-156)
  void update_proposal() {
    this->_dev->check(gvi_ngd_accept(this->_dev->get()));
    this->pull_state();
  }
};

// ------------------------------------------------------------------------------------------------
// Proximal (JKO) variant: proxgd/ProxGVI-GH.h, proxgd/ProxGVI-GH-impl.h, proxgd/ProxGVIFactorizedBaseGH.h.
// The factor class carries the same constructor surface (ProxGVIFactorizedBaseGH.h:30-48); the update law lives
// in the device rule GVI_RULE_PROX_JKO.
// ------------------------------------------------------------------------------------------------
template <typename CostClass>
using ProxGVIFactorizedBaseGH = NGDFactorizedBaseGH<CostClass>;
using ProxGVIFactorizedSimpleGH = ProxGVIFactorizedBaseGH<NoneType>;   // proxgd/ProxGVIFactorizedSimpleGH.h

template <typename Factor>
class ProxGVIGH : public GVIGH<Factor> {
  using Base = GVIGH<Factor>;
 public:
  ProxGVIGH(const std::vector<std::shared_ptr<Factor>>& f, int dim_state, int num_states, int niterations = 5,
            double temperature = 1.0, double high_temperature = 100.0, int device = 0)
      : Base(f, dim_state, num_states, niterations, temperature, high_temperature, device) {
    this->_dev->check(gvi_ngd_set_update_rule(this->_dev->get(), GVI_RULE_PROX_JKO));
  }
  // compute_gradients(step) (proxgd/ProxGVI-GH-impl.h:43-88): plain sums of the factor-level JKO increments
  std::tuple<VectorXd, SpMat> compute_gradients(std::optional<double> step_size = std::nullopt) {
    this->_dev->check(gvi_prox_gradients(this->_dev->get(), step_size.value_or(this->_step_size_base)));
    const int T = this->_num_states, n = this->_dim_state;
    VectorXd dmu(T * n);
    std::vector<double> VD((size_t)T * n * n), VU((size_t)(T > 1 ? T - 1 : 0) * n * n);
    this->_dev->check(gvi_ngd_get_gradients(this->_dev->get(), nullptr, nullptr, nullptr, dmu.data(), VD.data(), VU.data()));
    return std::make_tuple(dmu, this->to_spmat(VD, VU));
  }
  double onestep_linesearch(double step_size) {                 // :24-41, trial kept on the device
    double c = 0.0;
    this->_dev->check(gvi_prox_trial(this->_dev->get(), step_size, &c));
    return c;
  }
  void update_proposal() {
    this->_dev->check(gvi_ngd_accept(this->_dev->get()));
    this->pull_state();
  }
  // ProxGVIGH::optimize (proxgd/ProxGVI-GH-impl.h:121-202)
  void optimize(std::optional<bool> verbose = std::nullopt) override {
    const bool is_verbose = verbose.value_or(true);
    for (int i_iter = 0; i_iter < this->_niters; i_iter++) {
      if (i_iter == this->_niters_lowtemp) this->switch_to_high_temperature();
      const double cost_iter = this->cost_value();
      if (is_verbose) std::printf("========= iteration %d ========= \n--- cost_iter ---\n%.15g\n", i_iter, cost_iter);
      VectorXd fact_costs = this->factor_cost_vector();
      this->record(cost_iter, fact_costs);
      int cnt = 0, B = 1;
      this->_dev->check(gvi_prox_gradients(this->_dev->get(), std::pow(this->_step_size_base, B)));
      while (true) {
        const double new_cost = onestep_linesearch(std::pow(this->_step_size_base, B));
        if (new_cost < cost_iter) { update_proposal(); break; }
        B += 1; cnt += 1;
        if (cnt > this->_niters_backtrack) {
          if (is_verbose) std::printf("Reached the maximum backtracking steps.\n");
          update_proposal();
          break;
        }
      }
    }
    if (!this->_prefix.empty()) this->save_data(is_verbose);
  }
};

}  // namespace gvi
