// gvi_host.hpp -- C++17 host shim over the C ABI (include/gvi_hip.h).
//
// Mirrors the reference's operator surface for the NGD Gauss-Hermite path so that a driver written
// against hzyu17/GaussianVI compiles against this header with the include lines changed:
//
//   gvi::SparseGaussHermite<Function>        quadrature/SparseGaussHermite.h:28-277
//   gvi::GVIFactorizedBase / ...BaseGH       gvibase/GVIFactorizedBase.h:36-248, gvibase/GVIFactorizedBaseGH.h:20-78
//   gvi::NGDFactorizedBaseGH<CostClass>      ngd/NGDFactorizedBaseGH.h:25-133      (NGDFactorizedSimpleGH alias)
//   gvi::NGDFactorizedLinearGH<Factor>       ngd/NGDFactorizedLinearGH.h:21-116
//   gvi::NGDFactorizedLinear<Factor>         ngd/NGDFactorizedLinear.h:22-135       (closed form, no sigma points)
//   gvi::MinimumAccGP / FixedPriorGP         gp/minimum_acc_prior.h:39-127, gp/fixed_prior.h:18-50, gp/cost_functions.h:25-39
//   gvi::FixedGpPrior[GH] / LinearGpPrior[GH]  gp/factorized_opts_linear.h:6-13
//   gvi::VIMPResults                         helpers/DataRecorder.h:25-225
//   gvi::GVIGH<Factor> / gvi::NGDGH<Factor>  gvibase/GVI-GH.h:27-414, gvibase/GVI-GH-impl.h, ngd/NGD-GH.h:25-95, ngd/NGD-GH-impl.h
//   gvi::ProxGVIGH<Factor>                   proxgd/ProxGVI-GH.h, proxgd/ProxGVI-GH-impl.h
//
// The reference is header-only on Eigen.  Eigen is not part of this image, so the shim carries two tiny dense types
// (VectorXd, MatrixXd: row-major, the ABI's layout) and a coefficient-map SpMat with the members the drivers use; when
// <Eigen/Dense> IS on the include path the three types convert implicitly from and to Eigen::VectorXd / MatrixXd /
// SparseMatrix<double> (section "Eigen interop"), so a reference driver keeps its Eigen objects.
//
// All arithmetic of the hot path happens on the device through the C ABI; there is no CPU fallback -- a missing device
// makes the first device call throw.  Host code here only moves blocks between the joint and the factor layouts (what
// TrajectoryBlock::extract / block insertion do in the reference) and inverts the small constant matrices of the GP
// model classes (Qc, K) once at construction.
//
// Two execution modes of the joint optimiser, same results:
//   * Execution::DeviceResident (default when every factor has a DevicePsi): the state never leaves HBM, one launch
//     sequence per pass for ALL factors (gvi_ngd_*) -- the fast path;
//   * Execution::FactorWise: the reference-shaped joint loop -- per-factor calculate_partial_V / fact_cost_value /
//     local2joint_*_insertion / update_*_from_joint exactly as ngd/NGD-GH-impl.h:21-63 and gvibase/GVI-GH-impl.h:127-197
//     call them.  The per-factor calls are served by a lazily batched device set (FactorBatch): the first factor that
//     asks triggers ONE device call for all factors of its set, the others read their slice.  Opaque host psi
//     (std::function without a DevicePsi) runs here through device expand -> host psi -> device reduction.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../gvi_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<Eigen/Sparse>) && !defined(GVI_HOST_NO_EIGEN)
#include <Eigen/Dense>
#include <Eigen/Sparse>
#define GVI_HOST_HAVE_EIGEN 1
#endif
#endif

// The key hash of QuadratureWeightsMap, defined where and how the reference defines it (a specialisation in namespace
// std, helpers/SerializeEigenMaps.h:29-41), so that `std::unordered_map<std::tuple<double, double>, ...>` spelled by a
// caller is the same type here.  A translation unit that already carries that specialisation defines
// GVI_HOST_NO_TUPLE_HASH.
#ifndef GVI_HOST_NO_TUPLE_HASH
namespace std {
template <>
struct hash<std::tuple<double, double>> {
  size_t operator()(const std::tuple<double, double>& key) const {
    const size_t hash1 = std::hash<double>{}(std::get<0>(key));
    const size_t hash2 = std::hash<double>{}(std::get<1>(key));
    return hash1 ^ (hash2 << 1);
  }
};
}  // namespace std
#endif

namespace gvi {

// ------------------------------------------------------------------------------------------------
// minimal dense types (row-major)
// ------------------------------------------------------------------------------------------------
class VectorXd {
 public:
  VectorXd() = default;
  explicit VectorXd(int n) : v_(n, 0.0) {}
  VectorXd(const double* p, int n) : v_(p, p + n) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  static VectorXd Constant(int n, double c) { VectorXd r(n); for (auto& x : r.v_) x = c; return r; }
  int size() const { return (int)v_.size(); }
  int rows() const { return size(); }
  double& operator()(int i) { return v_[i]; }
  double operator()(int i) const { return v_[i]; }
  double* data() { return v_.data(); }
  const double* data() const { return v_.data(); }
  void setZero() { for (auto& x : v_) x = 0.0; }
  VectorXd segment(int start, int len) const { return VectorXd(v_.data() + start, len); }
  VectorXd& operator+=(const VectorXd& o) { for (size_t i = 0; i < v_.size(); ++i) v_[i] += o.v_[i]; return *this; }
  VectorXd& operator-=(const VectorXd& o) { for (size_t i = 0; i < v_.size(); ++i) v_[i] -= o.v_[i]; return *this; }
  friend VectorXd operator+(VectorXd a, const VectorXd& b) { a += b; return a; }
  friend VectorXd operator-(VectorXd a, const VectorXd& b) { a -= b; return a; }
  friend VectorXd operator*(double s, VectorXd a) { for (auto& x : a.v_) x *= s; return a; }
  friend VectorXd operator*(VectorXd a, double s) { for (auto& x : a.v_) x *= s; return a; }
  VectorXd operator-() const { VectorXd r(*this); for (auto& x : r.v_) x = -x; return r; }
#ifdef GVI_HOST_HAVE_EIGEN
  VectorXd(const Eigen::VectorXd& e) : v_(e.data(), e.data() + e.size()) {}
  operator Eigen::VectorXd() const { Eigen::VectorXd e(size()); for (int i = 0; i < size(); ++i) e(i) = v_[i]; return e; }
#endif
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
  MatrixXd transpose() const { MatrixXd t(c_, r_); for (int i = 0; i < r_; ++i) for (int j = 0; j < c_; ++j) t(j, i) = (*this)(i, j); return t; }
  MatrixXd block(int i0, int j0, int nr, int nc) const {
    MatrixXd b(nr, nc);
    for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }
  void set_block(int i0, int j0, const MatrixXd& b) {
    for (int i = 0; i < b.rows(); ++i) for (int j = 0; j < b.cols(); ++j) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  friend MatrixXd operator*(const MatrixXd& a, const MatrixXd& b) {
    MatrixXd c(a.rows(), b.cols());
    for (int i = 0; i < a.rows(); ++i)
      for (int k = 0; k < a.cols(); ++k) { const double x = a(i, k); for (int j = 0; j < b.cols(); ++j) c(i, j) += x * b(k, j); }
    return c;
  }
  friend VectorXd operator*(const MatrixXd& a, const VectorXd& x) {
    VectorXd y(a.rows());
    for (int i = 0; i < a.rows(); ++i) { double s = 0.0; for (int j = 0; j < a.cols(); ++j) s += a(i, j) * x(j); y(i) = s; }
    return y;
  }
  friend MatrixXd operator*(double s, MatrixXd a) { for (auto& x : a.v_) x *= s; return a; }
  friend MatrixXd operator*(MatrixXd a, double s) { for (auto& x : a.v_) x *= s; return a; }
  friend MatrixXd operator/(MatrixXd a, double s) { for (auto& x : a.v_) x /= s; return a; }
  friend MatrixXd operator+(MatrixXd a, const MatrixXd& b) { for (size_t i = 0; i < a.v_.size(); ++i) a.v_[i] += b.v_[i]; return a; }
  friend MatrixXd operator-(MatrixXd a, const MatrixXd& b) { for (size_t i = 0; i < a.v_.size(); ++i) a.v_[i] -= b.v_[i]; return a; }
  MatrixXd operator-() const { MatrixXd r(*this); for (auto& x : r.v_) x = -x; return r; }
  // Dense inverse by Gauss-Jordan with partial pivoting.  Host-side, for the CONSTANT matrices of the model classes
  // (Qc, K: gp/minimum_acc_prior.h:47, gp/fixed_prior.h:26) and the precision() accessor of a factor; the hot path's
  // Sigma_k^-1 is computed on the device inside gvi_moments.
  MatrixXd inverse() const {
    const int n = r_;
    if (r_ != c_) throw std::invalid_argument("MatrixXd::inverse: not square");
    MatrixXd a(*this), b = Identity(n, n);
    for (int c = 0; c < n; ++c) {
      int p = c;
      for (int r = c + 1; r < n; ++r) if (std::fabs(a(r, c)) > std::fabs(a(p, c))) p = r;
      if (p != c) for (int j = 0; j < n; ++j) { std::swap(a(p, j), a(c, j)); std::swap(b(p, j), b(c, j)); }
      const double inv = 1.0 / a(c, c);
      for (int j = 0; j < n; ++j) { a(c, j) *= inv; b(c, j) *= inv; }
      for (int r = 0; r < n; ++r) {
        if (r == c) continue;
        const double f = a(r, c);
        if (f == 0.0) continue;
        for (int j = 0; j < n; ++j) { a(r, j) -= f * a(c, j); b(r, j) -= f * b(c, j); }
      }
    }
    return b;
  }
#ifdef GVI_HOST_HAVE_EIGEN
  MatrixXd(const Eigen::MatrixXd& e) : MatrixXd((int)e.rows(), (int)e.cols()) {
    for (int i = 0; i < r_; ++i) for (int j = 0; j < c_; ++j) (*this)(i, j) = e(i, j);        // column- to row-major
  }
  operator Eigen::MatrixXd() const {
    Eigen::MatrixXd e(r_, c_);
    for (int i = 0; i < r_; ++i) for (int j = 0; j < c_; ++j) e(i, j) = (*this)(i, j);
    return e;
  }
#endif
 private:
  int r_ = 0, c_ = 0;
  std::vector<double> v_;
};

// Sparse joint matrix as a coefficient map: what the drivers touch (coeffRef / coeff / insert / sums).
class SpMat {
 public:
  using Map = std::map<std::pair<int, int>, double>;
  SpMat() = default;
  SpMat(int r, int c) : r_(r), c_(c) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  void setZero() { m_.clear(); }
  double& coeffRef(int i, int j) { return m_[{i, j}]; }
  double& insert(int i, int j) { return m_[{i, j}]; }
  double coeff(int i, int j) const { auto it = m_.find({i, j}); return it == m_.end() ? 0.0 : it->second; }
  const Map& entries() const { return m_; }
  SpMat& operator+=(const SpMat& o) {
    if (r_ == 0) { r_ = o.r_; c_ = o.c_; }
    for (const auto& e : o.m_) m_[e.first] += e.second;
    return *this;
  }
  SpMat& operator-=(const SpMat& o) {
    if (r_ == 0) { r_ = o.r_; c_ = o.c_; }
    for (const auto& e : o.m_) m_[e.first] -= e.second;
    return *this;
  }
  friend SpMat operator+(SpMat a, const SpMat& b) { a += b; return a; }
  friend SpMat operator-(SpMat a, const SpMat& b) { a -= b; return a; }
  friend SpMat operator*(double s, SpMat a) { for (auto& e : a.m_) e.second *= s; return a; }
  friend SpMat operator*(SpMat a, double s) { for (auto& e : a.m_) e.second *= s; return a; }
  // dense block (rows i0.., cols j0..) -- TrajectoryBlock::extract (helpers/MatrixHelper.h:132-134)
  MatrixXd block(int i0, int j0, int nr, int nc) const {
    MatrixXd b(nr, nc);
    for (auto it = m_.lower_bound({i0, 0}); it != m_.end() && it->first.first < i0 + nr; ++it) {
      const int j = it->first.second;
      if (j >= j0 && j < j0 + nc) b(it->first.first - i0, j - j0) = it->second;
    }
    return b;
  }
#ifdef GVI_HOST_HAVE_EIGEN
  SpMat(const Eigen::SparseMatrix<double>& e) : r_((int)e.rows()), c_((int)e.cols()) {
    for (int k = 0; k < e.outerSize(); ++k)
      for (Eigen::SparseMatrix<double>::InnerIterator it(e, k); it; ++it) m_[{(int)it.row(), (int)it.col()}] = it.value();
  }
  operator Eigen::SparseMatrix<double>() const {
    Eigen::SparseMatrix<double> e(r_, c_);
    for (const auto& kv : m_) e.coeffRef(kv.first.first, kv.first.second) = kv.second;
    return e;
  }
#endif
 private:
  int r_ = 0, c_ = 0;
  Map m_;
};

struct NoneType {};

// quadrature/SparseGHQuadratureWeights.h:14-16: (dimension, degree) -> (zero-mean points N x d, weights N)
using DimDegTuple = std::tuple<double, double>;
using PointsWeightsTuple = std::tuple<MatrixXd, VectorXd>;
using QuadratureWeightsMap = std::unordered_map<DimDegTuple, PointsWeightsTuple>;

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
  bool same_group(const DevicePsi& o) const { return kind == o.kind && sdf2d == o.sdf2d && sdf3d == o.sdf3d && arm == o.arm; }
};

// Upload of a set's shared field / arm (gvi_factors_set_sdf2d / _sdf3d / _arm).
inline void upload_psi_shared(const Device& dev, int set_id, const DevicePsi& dp) {
  if (dp.sdf2d) {                                   // column-major rows x cols, like Eigen's MatrixXd
    const MatrixXd& fld = dp.sdf2d->field;
    std::vector<double> cm((size_t)fld.rows() * fld.cols());
    for (int c = 0; c < fld.cols(); ++c)
      for (int r = 0; r < fld.rows(); ++r) cm[(size_t)c * fld.rows() + r] = fld(r, c);
    dev.check(gvi_factors_set_sdf2d(dev.get(), set_id, dp.sdf2d->origin_x, dp.sdf2d->origin_y, dp.sdf2d->cell_size,
                                    fld.rows(), fld.cols(), cm.data()));
  }
  if (dp.sdf3d)
    dev.check(gvi_factors_set_sdf3d(dev.get(), set_id, dp.sdf3d->origin, dp.sdf3d->cell_size, dp.sdf3d->rows, dp.sdf3d->cols,
                                    dp.sdf3d->nz, dp.sdf3d->data.data()));
  if (dp.arm)
    dev.check(gvi_factors_set_arm(dev.get(), set_id, (int)dp.arm->a.size(), dp.arm->a.data(), dp.arm->alpha.data(),
                                  dp.arm->d.data(), dp.arm->theta_bias.data(), (int)dp.arm->frames.size(),
                                  dp.arm->frames.data(), dp.arm->centers.data(), dp.arm->radii.data()));
}

// ------------------------------------------------------------------------------------------------
// SparseGaussHermite (quadrature/SparseGaussHermite.h): table lookup, symmetric-sqrt expand on the
// device, weighted reduction of an arbitrary host function.
// ------------------------------------------------------------------------------------------------
// Fills a QuadratureWeightsMap from the reference's table file -- what `cereal::BinaryInputArchive archive(ifs);
// archive(nodes_weights_map);` does at quadrature/SparseGaussHermite.h:62-71 / :95-108 -- through gvi_table_file_list /
// gvi_table_file_read (layout helpers/SerializeEigenMaps.h:195-224).  Same failure as :58-61: std::runtime_error.
inline QuadratureWeightsMap read_quadrature_weights_map(const std::string& map_file) {
  const std::string error_msg = "Failed to open file for GH weights reading in file: " + map_file;
  int64_t count = 0;
  if (gvi_table_file_list(map_file.c_str(), 0, &count, nullptr, nullptr, nullptr) != GVI_OK) throw std::runtime_error(error_msg);
  std::vector<double> dims((size_t)count), degs((size_t)count);
  std::vector<int64_t> rows((size_t)count);
  if (count && gvi_table_file_list(map_file.c_str(), count, &count, dims.data(), degs.data(), rows.data()) != GVI_OK)
    throw std::runtime_error(error_msg);
  QuadratureWeightsMap map;
  for (int64_t e = 0; e < count; ++e) {
    const int d = (int)dims[(size_t)e], p = (int)degs[(size_t)e];
    int64_t N = rows[(size_t)e];
    MatrixXd Z((int)N, d);
    VectorXd w((int)N);
    if (gvi_table_file_read(map_file.c_str(), d, p, &N, Z.data(), w.data()) != GVI_OK) throw std::runtime_error(error_msg);
    map[std::make_tuple(dims[(size_t)e], degs[(size_t)e])] = std::make_tuple(std::move(Z), std::move(w));
  }
  return map;
}
inline std::shared_ptr<QuadratureWeightsMap> read_quadrature_weights_map_shared(const std::string& map_file) {
  return std::make_shared<QuadratureWeightsMap>(read_quadrature_weights_map(map_file));
}
// (dim, deg) -> the map's entry, or nullptr (the lookup of SparseGaussHermite::computeSigmaPtsWeights, :138-166)
inline const PointsWeightsTuple* find_points_weights(const QuadratureWeightsMap& map, int dim, int deg) {
  const auto it = map.find(std::make_tuple((double)dim, (double)deg));
  return it == map.end() ? nullptr : &it->second;
}

template <typename Function = std::function<MatrixXd(const VectorXd&)>>
class SparseGaussHermite {
 public:
  virtual ~SparseGaussHermite() {}
  // The reference's three constructors (quadrature/SparseGaussHermite.h:38-77, :79-117, :120-132) + the device context
  // as an optional trailing argument.  Without a map the reference reads the table FILE at every construction
  // (:53-71); here the built-in nwspgr generator supplies the same table (bit-exact nodes, weights to 1e-13).  Like
  // the reference, the first two are told apart by the type of the fifth argument; only the second keeps its default
  // so that a four-argument call stays unambiguous.
  SparseGaussHermite(const int& deg, const int& dim, const VectorXd& mean, const MatrixXd& P,
                     std::optional<QuadratureWeightsMap> weight_sigpts_map_option, std::shared_ptr<Device> dev = nullptr)
      : _deg(deg), _dim(dim), _mean(mean), _P(P), _dev(dev ? dev : std::make_shared<Device>()) {
    if (weight_sigpts_map_option.has_value()) _nodes_weights_map = std::make_shared<QuadratureWeightsMap>(weight_sigpts_map_option.value());
    computeSigmaPtsWeights();
  }
  SparseGaussHermite(const int& deg, const int& dim, const VectorXd& mean, const MatrixXd& P,
                     std::optional<std::shared_ptr<QuadratureWeightsMap>> weight_sigpts_map_option = std::nullopt,
                     std::shared_ptr<Device> dev = nullptr)
      : _deg(deg), _dim(dim), _mean(mean), _P(P), _dev(dev ? dev : std::make_shared<Device>()) {
    if (weight_sigpts_map_option.has_value()) _nodes_weights_map = weight_sigpts_map_option.value();
    computeSigmaPtsWeights();
  }
  SparseGaussHermite(const int& deg, const int& dim, const VectorXd& mean, const MatrixXd& P, const QuadratureWeightsMap& weights_map,
                     std::shared_ptr<Device> dev = nullptr)
      : _deg(deg), _dim(dim), _mean(mean), _P(P), _dev(dev ? dev : std::make_shared<Device>()) {
    computeSigmaPtsWeights(weights_map);
  }

  void computeSigmaPtsWeights() {   // :138-166
    if (_nodes_weights_map) { lookup(*_nodes_weights_map, false); return; }
    int64_t N = 0;
    gvi_status s = gvi_spgh_count(_dim, _deg, &N);
    if (s != GVI_OK) { missing_key(); return; }
    _zeromeanpts = MatrixXd((int)N, _dim);
    _Weights = VectorXd((int)N);
    _dev->check(gvi_spgh_nodes(_dim, _deg, N, _zeromeanpts.data(), _Weights.data(), nullptr));
    device_set(false);
    update_sigmapoints();
  }
  void computeSigmaPtsWeights(const QuadratureWeightsMap& weights_map) { lookup(weights_map, true); }   // :171-193

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
  inline void update_mean(const VectorXd& mean) { _mean = mean; }
  inline void update_P(const MatrixXd& P) { _P = P; }
  void update_sigmapoints() {                          // :231-243, on the device
    const int N = _Weights.size();
    if (_set < 0 || N == 0) { _sigmapts = MatrixXd(0, _dim); return; }
    std::vector<double> X((size_t)_dim * N);
    _dev->check(gvi_expand(_dev->get(), _set, _mean.data(), _P.data(), X.data()));
    _sigmapts = MatrixXd(N, _dim);
    for (int a = 0; a < _dim; ++a)
      for (int i = 0; i < N; ++i) _sigmapts(i, a) = X[(size_t)a * N + i];
  }
  inline void set_polynomial_deg(const int& deg) { _deg = deg; computeSigmaPtsWeights(); }                 // :249-252
  inline void update_dimension(const int& dim) { _dim = dim; computeSigmaPtsWeights(); }                   // :254-257
  inline void update_parameters(const int& deg, const int& dim, const VectorXd& mean, const MatrixXd& P) {   // :259-271
    _deg = deg; _dim = dim; _mean = mean; _P = P;
    computeSigmaPtsWeights();
  }
  inline VectorXd weights() const { return _Weights; }
  inline MatrixXd sigmapts() const { return _sigmapts; }
  inline VectorXd mean() const { return _mean; }
  inline MatrixXd zeromeanpts() const { return _zeromeanpts; }

 protected:
  void missing_key() {                                  // the reference only prints (:159-162) and keeps what it had
    std::printf("(dimension, degree) (%d, %d) key does not exist in the GH weight map.\n", _dim, _deg);
  }
  void lookup(const QuadratureWeightsMap& map, bool announce) {
    const PointsWeightsTuple* e = find_points_weights(map, _dim, _deg);
    if (!e) { missing_key(); return; }
    if (announce) std::printf("(dimension, degree) tuple: (%d, %d) exists in the GH weight map.\n", _dim, _deg);
    _zeromeanpts = std::get<0>(*e);
    _Weights = std::get<1>(*e);
    if (_zeromeanpts.rows() != _Weights.size() || _zeromeanpts.cols() != _dim || _Weights.size() < 1)
      throw std::invalid_argument("SparseGaussHermite: map entry is not (N x dim points, N weights)");
    device_set(true);
    update_sigmapoints();
  }
  // one single-factor device set carrying the table: the generated (dim, deg) one or the caller's
  void device_set(bool from_map) {
    _dev->check(gvi_chain_set(_dev->get(), 1, _dim));
    const int32_t start = 0;
    if (from_map)
      _dev->check(gvi_factors_add_table(_dev->get(), 1, _dim, _deg, &start, GVI_PSI_HOST_CALLBACK, nullptr, 0, nullptr,
                                        _Weights.size(), _zeromeanpts.data(), _Weights.data(), &_set));
    else
      _dev->check(gvi_factors_add(_dev->get(), 1, _dim, _deg, &start, GVI_PSI_HOST_CALLBACK, nullptr, 0, nullptr, &_set));
  }
  int _deg, _dim, _set = -1;
  VectorXd _mean;
  MatrixXd _P;
  VectorXd _Weights;
  MatrixXd _sigmapts, _zeromeanpts;
  std::shared_ptr<QuadratureWeightsMap> _nodes_weights_map;
  std::shared_ptr<Device> _dev;
};

// ------------------------------------------------------------------------------------------------
// GP model classes (gp/linear_factor.h, gp/minimum_acc_prior.h, gp/fixed_prior.h, gp/cost_functions.h)
// ------------------------------------------------------------------------------------------------
class LinearFactor {   // gp/linear_factor.h:17-32: -log p = C ||A x - B mu_t||^2_{Sigma_t^-1}
 public:
  virtual ~LinearFactor() {}
  virtual VectorXd get_mu() const = 0;
  virtual MatrixXd get_covariance() const = 0;
  virtual MatrixXd get_precision() const = 0;
  virtual MatrixXd get_Lambda() const = 0;
  virtual MatrixXd get_Psi() const = 0;
  virtual double get_Constant() const = 0;
  virtual DevicePsi device_psi() const = 0;          // the device kind this model evaluates to
};

// Minimum-acceleration (constant-velocity) GP prior between two consecutive states [x; v]
// (gp/minimum_acc_prior.h:39-127): Phi = [I dt I; 0 I], Q^-1 blocks {12/dt^3, -6/dt^2, 4/dt} Qc^-1, Lambda = [-Phi, I],
// Psi = 0, target mean 0, constant 1/2.
class MinimumAccGP : public LinearFactor {
 public:
  MinimumAccGP() {}
  MinimumAccGP(const MatrixXd& Qc, double start_index, const double& delta_t, const VectorXd& mu_0)
      : _dim(Qc.cols()), _start_index((int)start_index), _dim_state(2 * Qc.cols()), _delta_t(delta_t), _Qc(Qc),
        _invQc(Qc.inverse()), _m0(mu_0), _target_mu(VectorXd::Zero(4 * Qc.cols())) {
    const int nd = _dim, n = _dim_state;
    _Phi = MatrixXd::Identity(n, n);
    for (int i = 0; i < nd; ++i) _Phi(i, nd + i) = delta_t;
    _Q = MatrixXd::Zero(n, n);
    _Q.set_block(0, 0, _Qc * (std::pow(_delta_t, 3) / 3));
    _Q.set_block(0, nd, _Qc * (std::pow(_delta_t, 2) / 2));
    _Q.set_block(nd, 0, _Qc * (std::pow(_delta_t, 2) / 2));
    _Q.set_block(nd, nd, _Qc * _delta_t);
    compute_invQ();
    _Lambda = MatrixXd::Zero(n, 2 * n);                 // [-Phi, I]
    _Lambda.set_block(0, 0, -_Phi);
    _Lambda.set_block(0, n, MatrixXd::Identity(n, n));
    _Psi = MatrixXd::Zero(n, 2 * n);                    // a(t) = 0: eliminated (:79-82)
  }
  inline MatrixXd Q() const { return _Q; }
  inline MatrixXd Qc() const { return _Qc; }
  inline MatrixXd Phi() const { return _Phi; }
  inline double cost(const VectorXd& theta1, const VectorXd& theta2) const {   // :103-106
    const VectorXd r = _Phi * theta1 - theta2;
    const VectorXd q = _invQ * r;
    double c = 0.0;
    for (int i = 0; i < r.size(); ++i) c += r(i) * q(i);
    return c / 2;
  }
  inline int dim_posvel() const { return 2 * _dim; }
  inline void compute_invQ() {                                                  // :110-116
    const int nd = _dim;
    _invQ = MatrixXd::Zero(2 * nd, 2 * nd);
    _invQ.set_block(0, 0, _invQc * (12.0 / std::pow(_delta_t, 3)));
    _invQ.set_block(0, nd, _invQc * (-6.0 / std::pow(_delta_t, 2)));
    _invQ.set_block(nd, 0, _invQc * (-6.0 / std::pow(_delta_t, 2)));
    _invQ.set_block(nd, nd, _invQc * (4.0 / _delta_t));
  }
  VectorXd get_mu() const override { return _target_mu; }
  MatrixXd get_precision() const override { return _invQ; }
  MatrixXd get_covariance() const override { return _invQ.inverse(); }
  MatrixXd get_Lambda() const override { return _Lambda; }
  MatrixXd get_Psi() const override { return _Psi; }
  double get_Constant() const override { return 0.5; }
  DevicePsi device_psi() const override { return DevicePsi::QuadPrior(_Phi, _invQ); }
 private:
  int _dim = 0, _start_index = 0, _dim_state = 0;
  double _delta_t = 0;
  MatrixXd _Qc, _invQc, _Q, _invQ, _Phi, _Lambda, _Psi;
  VectorXd _m0, _target_mu;
};

// Linear time-varying GP prior between two consecutive states (gp/LTV_prior.h:42-95, 123-197, 223-249): over [0, delta_t] the
// system x' = A(t) x + B(t) w with A, B piece-wise constant on FOUR sub-intervals (hA / hB hold 4 (n_states - 1) + 1 matrices,
// factor `start_index` reads hA[4 start_index .. 4 start_index + 4]); Phi' = A Phi, Phi(0) = I and the Gramian
// Q' = A Q + Q A^T + B B^T, Q(0) = 0 are integrated to tolerance 1e-12 -- the reference drives GSL's rkf45 (absent here); this
// is an embedded Runge-Kutta-Fehlberg 4(5) with step-size control, restarted at the three interior breakpoints so that no
// step straddles a jump of A / B.  Lambda = [-Phi, I], Psi = [Phi, -I], constant 1/2; cost() ignores the target mean exactly
// as the reference's does (:223-226).
class LTV_GP : public LinearFactor {
 public:
  LTV_GP() {}
  LTV_GP(const MatrixXd& Qc, int start_index, const double& delta_t, const VectorXd& mu_0, int n_states,
         const std::vector<MatrixXd>& hA, const std::vector<MatrixXd>& hB, const std::vector<VectorXd>& target_mean)
      : _dim(Qc.cols()), _start_index(start_index), _dim_state(2 * Qc.cols()), _delta_t(delta_t), _Qc(Qc), _invQc(Qc.inverse()),
        _m0(mu_0), _target_mu(VectorXd::Zero(4 * Qc.cols())) {
    (void)n_states;
    const int n = _dim_state;
    if ((int)hA.size() < 4 * start_index + 5 || (int)hB.size() < 4 * start_index + 5)
      throw GviError(GVI_ERR_ARG, "LTV_GP: hA / hB need 4 (n_states - 1) + 1 entries");
    _A_vec.assign(hA.begin() + 4 * start_index, hA.begin() + 4 * start_index + 5);
    _B_vec.assign(hB.begin() + 4 * start_index, hB.begin() + 4 * start_index + 5);
    _Phi = integrate(false);
    for (int i = 0; i < n; ++i) { _target_mu(i) = target_mean[start_index](i); _target_mu(n + i) = target_mean[start_index + 1](i); }
    _Q = integrate(true);
    compute_invQ();
    _Lambda = MatrixXd::Zero(n, 2 * n);                 // [-Phi, I]
    _Lambda.set_block(0, 0, -_Phi);
    _Lambda.set_block(0, n, MatrixXd::Identity(n, n));
    _Psi = MatrixXd::Zero(n, 2 * n);                    // [Phi, -I]
    _Psi.set_block(0, 0, _Phi);
    _Psi.set_block(0, n, -MatrixXd::Identity(n, n));
  }
  MatrixXd A_function(double t) const { return _A_vec[interval(t)]; }                             // :182-186
  std::pair<MatrixXd, MatrixXd> system_param(double t) const { const int i = interval(t); return {_A_vec[i], _B_vec[i]}; }   // :188-192
  inline MatrixXd Q() const { return _Q; }
  inline MatrixXd Qc() const { return _Qc; }
  inline MatrixXd Phi() const { return _Phi; }
  inline double cost(const VectorXd& theta1, const VectorXd& theta2) const {   // :223-226
    const VectorXd r = _Phi * theta1 - theta2;
    const VectorXd q = _invQ * r;
    double c = 0.0;
    for (int i = 0; i < r.size(); ++i) c += r(i) * q(i);
    return c / 2;
  }
  inline int dim_posvel() const { return 2 * _dim; }
  inline void compute_invQ() { _invQ = _Q.inverse(); }                          // :230-233
  VectorXd get_mu() const override { return _target_mu; }
  MatrixXd get_precision() const override { return _invQ; }
  MatrixXd get_covariance() const override { return _Q; }
  MatrixXd get_Lambda() const override { return _Lambda; }
  MatrixXd get_Psi() const override { return _Psi; }
  double get_Constant() const override { return 0.5; }
  DevicePsi device_psi() const override { return DevicePsi::QuadPrior(_Phi, _invQ); }

 private:
  int interval(double t) const { const int i = (int)std::floor(4 * t / _delta_t); return i < 0 ? 0 : (i > 4 ? 4 : i); }
  // right-hand side on sub-interval i: gramian ? A Y + Y A^T + B B^T : A Y
  MatrixXd rhs(int i, const MatrixXd& Y, bool gramian, const MatrixXd& BBt) const {
    MatrixXd d = _A_vec[i] * Y;
    if (gramian) d = d + Y * _A_vec[i].transpose() + BBt;
    return d;
  }
  static double norm_inf(const MatrixXd& M) {
    double m = 0.0;
    for (int i = 0; i < M.rows(); ++i) for (int j = 0; j < M.cols(); ++j) m = std::max(m, std::fabs(M(i, j)));
    return m;
  }
  // Runge-Kutta-Fehlberg 4(5) (the tableau of gsl_odeiv2_step_rkf45), error per step <= 1e-12 (1 + |Y|), 5th-order solution kept
  MatrixXd integrate(bool gramian) const {
    const int n = _dim_state;
    MatrixXd Y = gramian ? MatrixXd::Zero(n, n) : MatrixXd::Identity(n, n);
    const double tol = 1e-12, h4 = _delta_t / 4;
    for (int i = 0; i < 4; ++i) {
      const MatrixXd BBt = gramian ? _B_vec[i] * _B_vec[i].transpose() : MatrixXd::Zero(n, n);
      double t = 0.0, h = h4 / 4;
      while (t < h4) {
        if (t + h > h4) h = h4 - t;
        const MatrixXd k1 = rhs(i, Y, gramian, BBt);
        const MatrixXd k2 = rhs(i, Y + k1 * (h / 4), gramian, BBt);
        const MatrixXd k3 = rhs(i, Y + k1 * (3 * h / 32) + k2 * (9 * h / 32), gramian, BBt);
        const MatrixXd k4 = rhs(i, Y + k1 * (1932 * h / 2197) - k2 * (7200 * h / 2197) + k3 * (7296 * h / 2197), gramian, BBt);
        const MatrixXd k5 = rhs(i, Y + k1 * (439 * h / 216) - k2 * (8 * h) + k3 * (3680 * h / 513) - k4 * (845 * h / 4104), gramian, BBt);
        const MatrixXd k6 = rhs(i, Y - k1 * (8 * h / 27) + k2 * (2 * h) - k3 * (3544 * h / 2565) + k4 * (1859 * h / 4104) - k5 * (11 * h / 40), gramian, BBt);
        const MatrixXd y5 = Y + (k1 * (16.0 / 135) + k3 * (6656.0 / 12825) + k4 * (28561.0 / 56430) - k5 * (9.0 / 50) + k6 * (2.0 / 55)) * h;
        const MatrixXd err = (k1 * (1.0 / 360) - k3 * (128.0 / 4275) - k4 * (2197.0 / 75240) + k5 * (1.0 / 50) + k6 * (2.0 / 55)) * h;
        const double e = norm_inf(err), bound = tol * (1.0 + norm_inf(Y));
        if (e <= bound || h <= 1e-14 * h4) { Y = y5; t += h; }
        const double fac = e > 0.0 ? 0.9 * std::pow(bound / e, 0.2) : 4.0;
        h *= std::min(4.0, std::max(0.2, fac));
      }
    }
    return Y;
  }
  int _dim = 0, _start_index = 0, _dim_state = 0;
  double _delta_t = 0;
  MatrixXd _Qc, _invQc, _Q, _invQ, _Phi, _Lambda, _Psi;
  std::vector<MatrixXd> _A_vec, _B_vec;
  VectorXd _m0, _target_mu;
};

// Fixed Gaussian prior (gp/fixed_prior.h:18-50): psi(x) = (x - mu)^T K^-1 (x - mu), constant 1.
class FixedPriorGP : public LinearFactor {
 public:
  FixedPriorGP() {}
  FixedPriorGP(const MatrixXd& Covariance, const VectorXd& mu) : _K(Covariance), _invK(Covariance.inverse()), _dim(mu.size()), _mu(mu) {}
  double fixed_factor_cost(const VectorXd& x) const {                           // :28-30
    const VectorXd r = x - _mu;
    const VectorXd q = _invK * r;
    double c = 0.0;
    for (int i = 0; i < r.size(); ++i) c += r(i) * q(i);
    return c;
  }
  VectorXd get_mu() const override { return _mu; }
  MatrixXd get_precision() const override { return _invK; }
  MatrixXd get_covariance() const override { return _K; }
  MatrixXd get_Lambda() const override { return MatrixXd::Identity(_dim, _dim); }
  MatrixXd get_Psi() const override { return MatrixXd::Identity(_dim, _dim); }
  double get_Constant() const override { return 1.0; }
  DevicePsi device_psi() const override { return DevicePsi::FixedPrior(_mu, _invK); }
 private:
  MatrixXd _K, _invK;
  int _dim = 0;
  VectorXd _mu;
};

inline double cost_fixed_gp(const VectorXd& x, const FixedPriorGP& fixed_gp) { return fixed_gp.fixed_factor_cost(x); }   // gp/cost_functions.h:25-27
// cost_linear_gp: ONE overload per program, as in the reference (gp/cost_functions.h:36-39 for MinimumAccGP,
// gp/cost_functions_LTV.h:34-37 for LTV_GP; a program includes one of the two headers) -- the name is passed around as a
// plain function, which two visible overloads would make ambiguous
#ifdef GVI_FACTORIZED_OPTS_LTV
inline double cost_linear_gp(const VectorXd& pose_cmb, const LTV_GP& gp_ltv) {
  const int dim = gp_ltv.dim_posvel();
  return gp_ltv.cost(pose_cmb.segment(0, dim), pose_cmb.segment(dim, dim));
}
#else
inline double cost_linear_gp(const VectorXd& pose_cmb, const MinimumAccGP& gp_minacc) {
  const int dim = gp_minacc.dim_posvel();
  return gp_minacc.cost(pose_cmb.segment(0, dim), pose_cmb.segment(dim, dim));
}
#endif

// ------------------------------------------------------------------------------------------------
// Factor operator surface (gvibase/GVIFactorizedBase.h:36-248; gvibase/GVIFactorizedBaseGH.h:20-78)
// ------------------------------------------------------------------------------------------------
class FactorBatch;

class GVIFactorizedBase {
 public:
  virtual ~GVIFactorizedBase();
  GVIFactorizedBase() {}
  GVIFactorizedBase(int dimension, int state_dim, int num_states, int start_index, double temperature = 10.0,
                    double high_temperature = 100.0)
      : _dim(dimension), _state_dim(state_dim), _num_states(num_states), _start_index(start_index),
        _joint_size(state_dim * num_states), _mu(dimension), _precision(MatrixXd::Identity(dimension, dimension)),
        _covariance(MatrixXd::Identity(dimension, dimension)), _temperature(temperature),
        _high_temperature(high_temperature), _Vdmu(dimension), _Vddmu(dimension, dimension) {}

  inline void set_step_size(double step_size) { _step_size = step_size; }
  inline void update_mu(const VectorXd& new_mu) { _mu = new_mu; }                                   // :90-92
  inline void update_covariance(const MatrixXd& new_cov) { _covariance = new_cov; _precision_stale = true; }   // :97-100

  // update_mu_from_joint / update_precision_from_joint / extract_* (:104-122; TrajectoryBlock, helpers/MatrixHelper.h:119-161)
  inline void update_mu_from_joint(const VectorXd& fill_joint_mean) { _mu = extract_mu_from_joint(fill_joint_mean); }
  inline void update_precision_from_joint(const SpMat& fill_joint_cov) {
    _covariance = extract_cov_from_joint(fill_joint_cov);
    _precision_stale = true;            // Sigma_k^-1 of the hot path is formed on the device; the accessor inverts lazily
  }
  inline VectorXd extract_mu_from_joint(const VectorXd& fill_joint_mean) const {
    return fill_joint_mean.segment(_state_dim * _start_index, _dim);
  }
  inline MatrixXd extract_cov_from_joint(const SpMat& fill_joint_cov) const {
    const int o = _state_dim * _start_index;
    return fill_joint_cov.block(o, o, _dim, _dim);
  }

  // the operator surface proper (virtual declarations :128-168)
  virtual void calculate_partial_V(std::optional<double> step_size = std::nullopt) { (void)step_size; }
  virtual double fact_cost_value(const VectorXd& fill_joint_mean, const SpMat& joint_cov) { (void)fill_joint_mean; (void)joint_cov; return 0.0; }

  virtual inline VectorXd local2joint_dmu() { return local2joint_dmu_insertion(); }
  virtual inline VectorXd local2joint_dmu_insertion() {          // ngd/NGDFactorizedBaseGH.h:91-96
    VectorXd res(_joint_size);
    for (int i = 0; i < _dim; ++i) res(_state_dim * _start_index + i) = _Vdmu(i);
    return res;
  }
  virtual inline SpMat local2joint_dprecision() { return local2joint_dprecision_insertion(); }
  virtual inline SpMat local2joint_dprecision_insertion() {      // ngd/NGDFactorizedBaseGH.h:98-106
    SpMat res(_joint_size, _joint_size);
    const int o = _state_dim * _start_index;
    for (int i = 0; i < _dim; ++i)
      for (int j = 0; j < _dim; ++j) res.insert(i + o, j + o) = _Vddmu(i, j);
    return res;
  }
  inline SpMat fill_joint_cov() const {                          // :170-175
    SpMat joint_cov(_joint_size, _joint_size);
    const int o = _state_dim * _start_index;
    for (int i = 0; i < _dim; ++i) for (int j = 0; j < _dim; ++j) joint_cov.insert(i + o, j + o) = _covariance(i, j);
    return joint_cov;
  }
  inline VectorXd fill_joint_mean() const {                      // :177-182
    VectorXd joint_mean(_joint_size);
    for (int i = 0; i < _dim; ++i) joint_mean(_state_dim * _start_index + i) = _mu(i);
    return joint_mean;
  }

  inline VectorXd mean() const { return _mu; }
  inline MatrixXd precision() const {
    if (_precision_stale) { _precision = _covariance.inverse(); _precision_stale = false; }
    return _precision;
  }
  inline MatrixXd covariance() const { return _covariance; }
  inline VectorXd Vdmu() const { return _Vdmu; }
  inline MatrixXd Vddmu() const { return _Vddmu; }

  void factor_switch_to_high_temperature() { _temperature = _high_temperature; }     // :212-214
  double temperature() const { return _temperature; }

  // what the joint optimiser needs to place the factor in a homogeneous device set
  virtual int gh_degree() const { return 0; }
  virtual const DevicePsi& device_psi() const { static const DevicePsi none{}; return none; }
  virtual bool closed_form() const { return false; }            // NGDFactorizedLinear: no sigma points
  // the caller's shared quadrature table (weight_sigpts_map_option of the GH factor constructors); null = built-in
  virtual std::shared_ptr<QuadratureWeightsMap> weights_map() const { return nullptr; }
  virtual double psi(const VectorXd& x) const { (void)x; return 0.0; }   // the opaque host cost function

  // lazily batched device set serving this factor's operator calls (owned by the optimiser once the factor joins one;
  // a stand-alone factor creates a private single-factor set on first use)
  void attach(std::shared_ptr<FactorBatch> batch, int index);
  void detach() { _batch.reset(); _batch_index = -1; }

  int _dim = 0, _state_dim = 0, _num_states = 0, _start_index = 0, _joint_size = 0;
  VectorXd _mu;

 protected:
  FactorBatch& batch();
  mutable MatrixXd _precision;
  MatrixXd _covariance;
  mutable bool _precision_stale = false;
  double _step_size = 0.9, _E_Phi = 0.0;
  double _temperature = 1.0, _high_temperature = 10.0;
  VectorXd _Vdmu;
  MatrixXd _Vddmu;
  std::shared_ptr<FactorBatch> _batch;
  int _batch_index = -1;
  friend class FactorBatch;
};

// One homogeneous device set (same d, GH degree, psi kind, shared field) + the results of its last batched operator
// calls.  A member's call is a cache hit when the inputs it would send equal the ones the batch ran with.
class FactorBatch {
 public:
  FactorBatch(std::shared_ptr<Device> dev, int set_id, int d, int kind, std::vector<GVIFactorizedBase*> members)
      : _dev(std::move(dev)), _set(set_id), _d(d), _kind(kind), _members(std::move(members)) {
    const size_t K = _members.size(), dd = (size_t)d * d;
    _in_mu.assign(K * d, 0.0); _in_cov.assign(K * dd, 0.0); _temp.assign(K, 0.0);
    _Ephi.assign(K, 0.0); _Vdmu.assign(K * d, 0.0); _Vddmu.assign(K * dd, 0.0);
    _c_mu.assign(K * d, 0.0); _c_cov.assign(K * dd, 0.0); _cost.assign(K, 0.0);
    for (size_t k = 0; k < K; ++k) _temp[k] = _members[k]->temperature();
  }
  const std::shared_ptr<Device>& device() const { return _dev; }
  int set_id() const { return _set; }
  void forget(GVIFactorizedBase* f) { std::lock_guard<std::mutex> g(_mx); for (auto& m : _members) if (m == f) m = nullptr; }

  // calculate_partial_V of member k at its current (mu, covariance)
  void moments(int k, double& Ephi, VectorXd& Vdmu, MatrixXd& Vddmu) {
    std::lock_guard<std::mutex> g(_mx);
    const size_t d = _d, dd = d * d;
    const GVIFactorizedBase& f = *_members[k];
    if (!_mom_valid || !same(f._mu.data(), &_in_mu[k * d], d) || !same(f._covariance.data(), &_in_cov[k * dd], dd) ||
        temperatures_changed()) {
      for (size_t j = 0; j < _members.size(); ++j) {
        if (!_members[j]) continue;
        std::memcpy(&_in_mu[j * d], _members[j]->_mu.data(), d * 8);
        std::memcpy(&_in_cov[j * dd], _members[j]->_covariance.data(), dd * 8);
      }
      push_temperatures();
      if (_kind == GVI_PSI_HOST_CALLBACK) host_psi_pass(_in_mu, _in_cov, _Ephi.data(), _Vdmu.data(), _Vddmu.data());
      else _dev->check(gvi_moments(_dev->get(), _set, _in_mu.data(), _in_cov.data(), _Ephi.data(), _Vdmu.data(), _Vddmu.data()));
      _mom_valid = true;
      ++_device_calls;
    }
    Ephi = _Ephi[k];
    Vdmu = VectorXd(&_Vdmu[k * d], (int)d);
    Vddmu = MatrixXd((int)d, (int)d);
    std::memcpy(Vddmu.data(), &_Vddmu[k * dd], dd * 8);
  }

  // fact_cost_value of member k at (mean_k, cov_k) gathered from the joint every member is called with
  double cost(int k, const VectorXd& joint_mean, const SpMat& joint_cov) {
    std::lock_guard<std::mutex> g(_mx);
    const size_t d = _d, dd = d * d;
    const VectorXd mk = _members[k]->extract_mu_from_joint(joint_mean);
    const MatrixXd ck = _members[k]->extract_cov_from_joint(joint_cov);
    if (!_cost_valid || !same(mk.data(), &_c_mu[k * d], d) || !same(ck.data(), &_c_cov[k * dd], dd) || temperatures_changed()) {
      for (size_t j = 0; j < _members.size(); ++j) {
        if (!_members[j]) continue;
        const VectorXd mj = _members[j]->extract_mu_from_joint(joint_mean);
        const MatrixXd cj = _members[j]->extract_cov_from_joint(joint_cov);
        std::memcpy(&_c_mu[j * d], mj.data(), d * 8);
        std::memcpy(&_c_cov[j * dd], cj.data(), dd * 8);
      }
      push_temperatures();
      if (_kind == GVI_PSI_HOST_CALLBACK) {
        std::vector<double> e(_members.size()), vd(_members.size() * d), vdd(_members.size() * dd);
        host_psi_pass(_c_mu, _c_cov, e.data(), vd.data(), vdd.data());
        for (size_t j = 0; j < _members.size(); ++j) _cost[j] = e[j] / _temp[j];
      } else {
        _dev->check(gvi_costs(_dev->get(), _set, _c_mu.data(), _c_cov.data(), _cost.data()));
      }
      _cost_valid = true;
      ++_device_calls;
    }
    return _cost[k];
  }

  // E_Phi / E_xMuPhi / E_xMuxMuTPhi at member k's current (mu, covariance) (gvibase/GVIFactorizedBaseGH.h:54-64)
  void raw(int k, double& E0, VectorXd& E1, MatrixXd& E2) {
    std::lock_guard<std::mutex> g(_mx);
    const size_t K = _members.size(), d = _d, dd = d * d;
    std::vector<double> mu(K * d), cov(K * dd), e0(K), e1(K * d), e2(K * dd);
    for (size_t j = 0; j < K; ++j) {
      const GVIFactorizedBase* m = _members[j] ? _members[j] : _members[k];
      std::memcpy(&mu[j * d], m->_mu.data(), d * 8);
      std::memcpy(&cov[j * dd], m->_covariance.data(), dd * 8);
    }
    if (_kind == GVI_PSI_HOST_CALLBACK)
      throw GviError(GVI_ERR_UNSUPPORTED, "raw integrals of an opaque host psi: use SparseGaussHermite::Integrate");
    _dev->check(gvi_raw_moments(_dev->get(), _set, mu.data(), cov.data(), e0.data(), e1.data(), e2.data()));
    E0 = e0[k];
    E1 = VectorXd(&e1[k * d], (int)d);
    E2 = MatrixXd((int)d, (int)d);
    std::memcpy(E2.data(), &e2[k * dd], dd * 8);
  }
  long device_calls() const { return _device_calls; }

 private:
  static bool same(const double* a, const double* b, size_t n) { return std::memcmp(a, b, n * 8) == 0; }
  bool temperatures_changed() const {
    for (size_t j = 0; j < _members.size(); ++j) if (_members[j] && _members[j]->temperature() != _temp[j]) return true;
    return false;
  }
  void push_temperatures() {
    if (!temperatures_changed()) return;
    for (size_t j = 0; j < _members.size(); ++j) if (_members[j]) _temp[j] = _members[j]->temperature();
    _dev->check(gvi_factors_set_temperature(_dev->get(), _set, _temp.data()));
    _mom_valid = _cost_valid = false;
  }
  // opaque host psi: device expand (symmetric sqrt + sigma points) -> host psi -> device weighted reduction
  void host_psi_pass(const std::vector<double>& mu, const std::vector<double>& cov, double* Ephi, double* Vdmu, double* Vddmu) {
    int K = 0, d = 0, p = 0;
    int64_t N = 0;
    _dev->check(gvi_factors_info(_dev->get(), _set, &K, &d, &p, &N));
    std::vector<double> X((size_t)K * d * N), psi((size_t)K * N);
    _dev->check(gvi_expand(_dev->get(), _set, mu.data(), cov.data(), X.data()));
    VectorXd x(d);
    for (int k = 0; k < K; ++k) {
      const GVIFactorizedBase* f = _members[k];
      for (int64_t i = 0; i < N; ++i) {
        for (int a = 0; a < d; ++a) x(a) = X[((size_t)k * d + a) * N + i];
        psi[(size_t)k * N + i] = f ? f->psi(x) : 0.0;
      }
    }
    _dev->check(gvi_moments_from_psi(_dev->get(), _set, mu.data(), cov.data(), psi.data(), Ephi, Vdmu, Vddmu));
  }

  std::shared_ptr<Device> _dev;
  int _set, _d, _kind;
  std::vector<GVIFactorizedBase*> _members;
  std::vector<double> _in_mu, _in_cov, _temp, _Ephi, _Vdmu, _Vddmu, _c_mu, _c_cov, _cost;
  bool _mom_valid = false, _cost_valid = false;
  long _device_calls = 0;
  std::mutex _mx;
};

inline GVIFactorizedBase::~GVIFactorizedBase() { if (_batch) _batch->forget(this); }
inline void GVIFactorizedBase::attach(std::shared_ptr<FactorBatch> batch, int index) {
  if (_batch && _batch != batch) _batch->forget(this);
  _batch = std::move(batch);
  _batch_index = index;
}
// Adds one homogeneous set (the members' d, GH degree, psi kind) to `dev`'s chain and wraps it in a FactorBatch.
inline std::shared_ptr<FactorBatch> make_factor_batch(const std::shared_ptr<Device>& dev, const std::vector<GVIFactorizedBase*>& members) {
  const GVIFactorizedBase& f0 = *members[0];
  const DevicePsi& dp = f0.device_psi();
  const size_t K = members.size();
  std::vector<int32_t> start(K);
  std::vector<double> temp(K), params;
  for (size_t k = 0; k < K; ++k) {
    start[k] = members[k]->_start_index;
    temp[k] = members[k]->temperature();
    const DevicePsi& q = members[k]->device_psi();
    params.insert(params.end(), q.params.begin(), q.params.end());
  }
  int id = -1;
  const std::shared_ptr<QuadratureWeightsMap> map = f0.weights_map();
  if (map && !f0.closed_form()) {
    // a caller-supplied QuadratureWeightsMap: ONE upload per set (gvi_factors_add_table); SparseGaussHermite's lookup
    // (quadrature/SparseGaussHermite.h:138-166).  A missing key only prints there and leaves the factor without sigma
    // points -- every integral is then a sum over zero rows; a one-point table of weight 0 gives exactly that.
    const PointsWeightsTuple* e = find_points_weights(*map, f0._dim, f0.gh_degree());
    MatrixXd Z0(1, f0._dim);
    VectorXd w0(1);
    const MatrixXd* Z = &Z0;
    const VectorXd* w = &w0;
    if (e) {
      Z = &std::get<0>(*e);
      w = &std::get<1>(*e);
      if (Z->rows() != w->size() || Z->cols() != f0._dim || w->size() < 1)
        throw std::invalid_argument("QuadratureWeightsMap entry is not (N x dim points, N weights)");
    } else {
      std::printf("(dimension, degree) (%d, %d) key does not exist in the GH weight map.\n", f0._dim, f0.gh_degree());
    }
    dev->check(gvi_factors_add_table(dev->get(), (int)K, f0._dim, f0.gh_degree(), start.data(), dp.kind,
                                     params.empty() ? nullptr : params.data(), (int64_t)dp.params.size(), temp.data(),
                                     w->size(), Z->data(), w->data(), &id));
  } else {
    dev->check(gvi_factors_add(dev->get(), (int)K, f0._dim, f0.gh_degree(), start.data(), dp.kind, params.empty() ? nullptr : params.data(),
                               (int64_t)dp.params.size(), temp.data(), &id));
  }
  upload_psi_shared(*dev, id, dp);
  if (f0.closed_form()) dev->check(gvi_factors_set_closed_form(dev->get(), id, 1));
  return std::make_shared<FactorBatch>(dev, id, f0._dim, dp.kind, members);
}
inline FactorBatch& GVIFactorizedBase::batch() {
  if (!_batch) {                                              // stand-alone factor: private context + single-factor set
    auto dev = std::make_shared<Device>();
    dev->check(gvi_chain_set(dev->get(), _num_states, _state_dim));
    attach(make_factor_batch(dev, {this}), 0);
  }
  return *_batch;
}

// gvibase/GVIFactorizedBaseGH.h: the three Gauss-Hermite integrals of the factor at its current marginal
class GVIFactorizedBaseGH : public GVIFactorizedBase {
 public:
  GVIFactorizedBaseGH() {}
  // gvibase/GVIFactorizedBaseGH.h:35-40.  The map is what the factor's SparseGaussHermite looks (dimension, degree) up
  // in (ngd/NGDFactorizedBaseGH.h:49); here it becomes the table of the factor's device set.
  GVIFactorizedBaseGH(int dimension, int state_dim, int num_states, int start_index, double temperature = 10.0,
                      double high_temperature = 100.0,
                      std::optional<std::shared_ptr<QuadratureWeightsMap>> weight_sigpts_map_option = std::nullopt)
      : GVIFactorizedBase(dimension, state_dim, num_states, start_index, temperature, high_temperature),
        _weights_map(weight_sigpts_map_option.has_value() ? weight_sigpts_map_option.value() : nullptr) {}
  std::shared_ptr<QuadratureWeightsMap> weights_map() const override { return _weights_map; }
  // updateGH(x, P) (:44-49) has no separate state here: the integrals below run at (_mu, _covariance)
  void updateGH(const VectorXd& x, const MatrixXd& P) { _mu = x; _covariance = P; _precision_stale = true; }
  inline double E_Phi() { double e; VectorXd a; MatrixXd b; batch().raw(_batch_index, e, a, b); return e; }          // :54-56
  inline MatrixXd E_xMuPhi() {                                                                                      // :58-60
    double e; VectorXd a; MatrixXd b;
    batch().raw(_batch_index, e, a, b);
    MatrixXd r(_dim, 1);
    for (int i = 0; i < _dim; ++i) r(i, 0) = a(i);
    return r;
  }
  inline MatrixXd E_xMuxMuTPhi() { double e; VectorXd a; MatrixXd b; batch().raw(_batch_index, e, a, b); return b; }   // :62-64
 protected:
  std::shared_ptr<QuadratureWeightsMap> _weights_map;
};

// Shared implementation of the GH factor classes: calculate_partial_V (ngd/NGDFactorizedBaseGH.h:53-74 =
// ngd/NGDFactorizedLinearGH.h:86-107) and fact_cost_value (:122-129 = :109-116) on the device.
class NGDFactorDeviceOps : public GVIFactorizedBaseGH {
 public:
  using GVIFactorizedBaseGH::GVIFactorizedBaseGH;
  void calculate_partial_V(std::optional<double> step_size = std::nullopt) override {
    (void)step_size;
    // symmetric sqrt of Sigma_k, sigma points, psi, the three integrals, Lam_k = Sigma_k^-1,
    // Vdmu = Lam E[(x-mu)psi]/T, Vddmu = sym_upper(Lam E[(x-mu)(x-mu)^T psi] Lam - Lam E[psi])/T: one device pass
    batch().moments(_batch_index, _E_Phi, _Vdmu, _Vddmu);
  }
  double fact_cost_value(const VectorXd& fill_joint_mean, const SpMat& joint_cov) override {
    return batch().cost(_batch_index, fill_joint_mean, joint_cov);
  }
};

template <typename CostClass = NoneType>
class NGDFactorizedBaseGH : public NGDFactorDeviceOps {
 public:
  using Function = std::function<double(const VectorXd&, const CostClass&)>;
  // The reference's signature (ngd/NGDFactorizedBaseGH.h:37-44), weight_sigpts_map_option included, + an optional device
  // psi descriptor as an eleventh argument.  Without one the factor's psi stays the opaque host function (device expand
  // -> host psi -> device reduction).
  NGDFactorizedBaseGH(int dimension, int state_dim, int gh_degree, const Function& function, const CostClass& cost_class,
                      int num_states, int start_index, double temperature = 1.0, double high_temperature = 10.0,
                      std::optional<std::shared_ptr<QuadratureWeightsMap>> weight_sigpts_map_option = std::nullopt,
                      std::optional<DevicePsi> device_psi = std::nullopt)
      : NGDFactorDeviceOps(dimension, state_dim, num_states, start_index, temperature, high_temperature, weight_sigpts_map_option),
        _gh_degree(gh_degree), _function(function), _cost_class(cost_class),
        _psi(device_psi ? *device_psi : DevicePsi{}) {}
  // shorthand: a device psi and the built-in table
  NGDFactorizedBaseGH(int dimension, int state_dim, int gh_degree, const Function& function, const CostClass& cost_class,
                      int num_states, int start_index, double temperature, double high_temperature, const DevicePsi& device_psi)
      : NGDFactorizedBaseGH(dimension, state_dim, gh_degree, function, cost_class, num_states, start_index, temperature,
                            high_temperature, std::nullopt, device_psi) {}
  double psi(const VectorXd& x) const override { return _function(x, _cost_class); }
  int gh_degree() const override { return _gh_degree; }
  const DevicePsi& device_psi() const override { return _psi; }
  // (x - mu) psi(x), (x - mu)(x - mu)^T psi(x): the closures of :46-48, for callers that integrate on the host
  inline MatrixXd negative_log_probability(const VectorXd& x) const { return MatrixXd::Constant(1, 1, psi(x)); }
  inline MatrixXd xMu_negative_log_probability(const VectorXd& x) const {
    MatrixXd r(_dim, 1);
    const double p = psi(x);
    for (int i = 0; i < _dim; ++i) r(i, 0) = (x(i) - _mu(i)) * p;
    return r;
  }
  inline MatrixXd xMuxMuT_negative_log_probability(const VectorXd& x) const {
    MatrixXd r(_dim, _dim);
    const double p = psi(x);
    for (int i = 0; i < _dim; ++i) for (int j = 0; j < _dim; ++j) r(i, j) = (x(i) - _mu(i)) * (x(j) - _mu(j)) * p;
    return r;
  }
 private:
  int _gh_degree;
  Function _function;
  CostClass _cost_class;
  DevicePsi _psi;
};
using NGDFactorizedSimpleGH = NGDFactorizedBaseGH<NoneType>;   // ngd/NGDFactorizedSimpleGH.h

// Linear-Gaussian factor evaluated by quadrature (ngd/NGDFactorizedLinearGH.h:21-116): the device kind comes from the
// model object (MinimumAccGP -> QUAD_PRIOR(Phi, Q^-1), FixedPriorGP -> FIXED_PRIOR(mu, K^-1)).
template <typename Factor = NoneType>
class NGDFactorizedLinearGH : public NGDFactorDeviceOps {
 public:
  using CostFunction = std::function<double(const VectorXd&, const Factor&)>;
  NGDFactorizedLinearGH(const int& dimension, int dim_state, int gh_degree, const CostFunction& function, const Factor& linear_factor,
                        int num_states, int start_indx, double temperature, double high_temperature,
                        std::optional<std::shared_ptr<QuadratureWeightsMap>> weight_sigpts_map_option = std::nullopt)   // :27-37
      : NGDFactorDeviceOps(dimension, dim_state, num_states, start_indx, temperature, high_temperature, weight_sigpts_map_option),
        _gh_degree(gh_degree),
        _function(function), _linear_factor(linear_factor), _psi(linear_factor.device_psi()),
        _target_mean(linear_factor.get_mu()), _target_precision(linear_factor.get_precision()), _Lambda(linear_factor.get_Lambda()),
        _Psi(linear_factor.get_Psi()), _constant(linear_factor.get_Constant()) {}
  double constant() const { return _constant; }
  double psi(const VectorXd& x) const override { return _function(x, _linear_factor); }
  int gh_degree() const override { return _gh_degree; }
  const DevicePsi& device_psi() const override { return _psi; }
 protected:
  int _gh_degree;
  CostFunction _function;
  Factor _linear_factor;
  DevicePsi _psi;
  VectorXd _target_mean;
  MatrixXd _target_precision, _Lambda, _Psi;
  double _constant;
};

// Linear-Gaussian factor in closed form (ngd/NGDFactorizedLinear.h:22-135): no sigma points; on the device through
// gvi_factors_set_closed_form (Isserlis in the whitened space instead of the reference's O(d^4) loop, :108-118).
template <typename Factor = NoneType>
class NGDFactorizedLinear : public NGDFactorDeviceOps {
 public:
  using CostFunction = std::function<double(const VectorXd&, const Factor&)>;
  NGDFactorizedLinear(const int& dimension, int dim_state, const CostFunction& function, const Factor& linear_factor, int num_states,
                      int start_indx, double temperature, double high_temperature)
      : NGDFactorDeviceOps(dimension, dim_state, num_states, start_indx, temperature, high_temperature), _function(function),
        _linear_factor(linear_factor), _psi(linear_factor.device_psi()), _constant(linear_factor.get_Constant()) {}
  double constant() const { return _constant; }
  double psi(const VectorXd& x) const override { return _function(x, _linear_factor); }
  int gh_degree() const override { return 3; }            // a table must exist for the set; it is never read in closed form
  const DevicePsi& device_psi() const override { return _psi; }
  bool closed_form() const override { return true; }
 protected:
  CostFunction _function;
  Factor _linear_factor;
  DevicePsi _psi;
  double _constant;
};

// gp/factorized_opts_linear.h:6-13; with GVI_FACTORIZED_OPTS_LTV (include/gvi/factorized_opts_LTV.hpp) the linear prior is
// LTV_GP instead of MinimumAccGP, as in gp/factorized_opts_LTV.h:7-13 -- the reference chooses between the two by which of
// the two headers a program includes, the names are the same
using FixedGpPrior = NGDFactorizedLinear<FixedPriorGP>;
using FixedGpPriorGH = NGDFactorizedLinearGH<FixedPriorGP>;
#ifdef GVI_FACTORIZED_OPTS_LTV
using LinearGpPrior = NGDFactorizedLinear<LTV_GP>;
using LinearGpPriorGH = NGDFactorizedLinearGH<LTV_GP>;
#else
using LinearGpPrior = NGDFactorizedLinear<MinimumAccGP>;
using LinearGpPriorGH = NGDFactorizedLinearGH<MinimumAccGP>;
#endif

// ------------------------------------------------------------------------------------------------
// Result recorder (helpers/DataRecorder.h:25-225): nine CSV files, one COLUMN per iteration, Eigen's column-major
// flattening (compress3d = mat.reshaped(), helpers/EigenWrapper.h:228-232), format `FullPrecision, ", "`
// (helpers/CommonDefinitions.h:31-32).  Rows: mean T n | cov, precision n^2 T (state-major, each n x n block
// column-major) | joint_cov, joint_precision (T n)^2 (column-major) | factor_costs nfactors | cost niters x 1 |
// zk_sdf n x T and Sk_sdf n^2 x T of the LAST recorded iteration.
// ------------------------------------------------------------------------------------------------
class VIMPResults {
 public:
  VIMPResults() {}
  VIMPResults(int niters, int dim_state, int nstates, int n_factors)
      : _niters(niters), _dim_state(dim_state), _nstates(nstates), _nfactors(n_factors) {
    const size_t Tn = (size_t)dim_state * nstates;
    // the reference allocates (T n)^2 x niters doubles up front (302 MB per column at T = 1025, n = 6); the joint
    // files are kept only while they stay below 64 M entries in total
    _keep_joint = Tn * Tn * (size_t)std::max(niters, 1) <= ((size_t)1 << 26);
  }
  // update_data(new_mean, new_joint_cov, new_joint_precision, new_cost, new_factor_costs) (:88-112)
  void update_data(const VectorXd& new_mean, const SpMat& new_joint_cov, const SpMat& new_joint_precision, const double& new_cost,
                   const VectorXd& new_factor_costs) {
    if (_cur_iter >= _niters) { std::printf("reached the last iteration\n"); return; }
    mean.emplace_back(new_mean.data(), new_mean.data() + new_mean.size());
    cov.push_back(joint2marginals(new_joint_cov));
    precision.push_back(joint2marginals(new_joint_precision));
    factor_costs.emplace_back(new_factor_costs.data(), new_factor_costs.data() + new_factor_costs.size());
    if (_keep_joint) {
      joint_cov.push_back(flatten_joint(new_joint_cov));
      joint_precision.push_back(flatten_joint(new_joint_precision));
    }
    cost.push_back(new_cost);
    ++_cur_iter;
  }
  // the same record from the C ABI's block layout (D / U row-major blocks): no coefficient maps on the resident path
  void update_data_blocks(const VectorXd& new_mean, const std::vector<double>& SigD, const std::vector<double>& SigU,
                          const std::vector<double>& D, const std::vector<double>& U, const double& new_cost,
                          const VectorXd& new_factor_costs) {
    if (_cur_iter >= _niters) { std::printf("reached the last iteration\n"); return; }
    mean.emplace_back(new_mean.data(), new_mean.data() + new_mean.size());
    cov.push_back(blocks2marginals(SigD));
    precision.push_back(blocks2marginals(D));
    factor_costs.emplace_back(new_factor_costs.data(), new_factor_costs.data() + new_factor_costs.size());
    if (_keep_joint) {
      joint_cov.push_back(flatten_blocks(SigD, SigU));
      joint_precision.push_back(flatten_blocks(D, U));
    }
    cost.push_back(new_cost);
    ++_cur_iter;
  }
  std::vector<double> blocks2marginals(const std::vector<double>& Dg) const {
    const int n = _dim_state;
    std::vector<double> out((size_t)n * n * _nstates, 0.0);
    for (int t = 0; t < _nstates; ++t)
      for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) out[(size_t)t * n * n + (size_t)c * n + r] = Dg[((size_t)t * n + r) * n + c];
    return out;
  }
  std::vector<double> flatten_blocks(const std::vector<double>& Dg, const std::vector<double>& Ug) const {
    const int n = _dim_state, T = _nstates;
    const size_t Tn = (size_t)n * T;
    std::vector<double> out(Tn * Tn, 0.0);
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
          const size_t i = (size_t)t * n + r, j = (size_t)t * n + c;
          out[j * Tn + i] = Dg[((size_t)t * n + r) * n + c];
          if (t + 1 < T) {
            const double u = Ug[((size_t)t * n + r) * n + c];          // block (t, t+1) and its mirror
            out[(j + n) * Tn + i] = u;
            out[i * Tn + (j + n)] = u;
          }
        }
    return out;
  }
  // joint2marginals (:120-127): the T diagonal n x n blocks, each flattened column-major
  std::vector<double> joint2marginals(const SpMat& joint) const {
    const int n = _dim_state;
    std::vector<double> out((size_t)n * n * _nstates, 0.0);
    for (int t = 0; t < _nstates; ++t) {
      const MatrixXd b = joint.block(t * n, t * n, n, n);
      for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) out[(size_t)t * n * n + (size_t)c * n + r] = b(r, c);
    }
    return out;
  }
  std::vector<double> flatten_joint(const SpMat& joint) const {
    const size_t Tn = (size_t)_dim_state * _nstates;
    std::vector<double> out(Tn * Tn, 0.0);
    for (const auto& e : joint.entries()) out[(size_t)e.first.second * Tn + e.first.first] = e.second;   // column-major
    return out;
  }
  inline void update_file_names(const std::string& file_mean, const std::string& file_cov, const std::string& file_joint_cov,
                                const std::string& file_precision, const std::string& file_joint_precision, const std::string& file_cost,
                                const std::string& file_factor_costs, const std::string& file_zk_sdf, const std::string& file_Sk_sdf) {
    _file_mean = file_mean; _file_cov = file_cov; _file_joint_cov = file_joint_cov; _file_precision = file_precision;
    _file_joint_precision = file_joint_precision; _file_cost = file_cost; _file_factor_costs = file_factor_costs;
    _file_zk_sdf = file_zk_sdf; _file_Sk_sdf = file_Sk_sdf;
  }
  void save_data(bool verbose = true) {     // :177-224
    // like the reference's preallocated (rows x niters) arrays: iterations that never ran stay zero columns
    save_columns(_file_mean, mean, (size_t)_dim_state * _nstates, verbose);
    save_columns(_file_cov, cov, (size_t)_dim_state * _dim_state * _nstates, verbose);
    save_columns(_file_precision, precision, (size_t)_dim_state * _dim_state * _nstates, verbose);
    if (_keep_joint) {
      const size_t Tn = (size_t)_dim_state * _nstates;
      save_columns(_file_joint_cov, joint_cov, Tn * Tn, verbose);
      save_columns(_file_joint_precision, joint_precision, Tn * Tn, verbose);
    } else if (verbose) {
      std::printf("joint_cov / joint_precision not recorded: (T n)^2 x niters exceeds 64 M entries\n");
    }
    {
      note(_file_cost, verbose);
      std::ofstream f(_file_cost);
      f.precision(15);
      for (int it = 0; it < _niters; ++it) f << (it < (int)cost.size() ? cost[it] : 0.0) << "\n";
    }
    save_columns(_file_factor_costs, factor_costs, (size_t)_nfactors, verbose);
    if (!mean.empty()) {                                    // last iteration: zk_sdf n x T, Sk_sdf n^2 x T
      const int n = _dim_state, T = _nstates;
      note(_file_zk_sdf, verbose);
      std::ofstream fz(_file_zk_sdf);
      fz.precision(15);
      for (int r = 0; r < n; ++r) for (int t = 0; t < T; ++t) fz << mean.back()[(size_t)t * n + r] << (t + 1 < T ? ", " : "\n");
      note(_file_Sk_sdf, verbose);
      std::ofstream fs(_file_Sk_sdf);
      fs.precision(15);
      for (int j = 0; j < n * n; ++j) for (int t = 0; t < T; ++t) fs << cov.back()[(size_t)t * n * n + j] << (t + 1 < T ? ", " : "\n");
    }
    if (verbose) std::printf("All data saved\n");
  }
  int recorded() const { return _cur_iter; }
  // collection of all iterations (one vector per recorded iteration, in the files' row order)
  std::vector<std::vector<double>> mean, precision, cov, joint_cov, joint_precision, factor_costs;
  std::vector<double> cost;

 private:
  static void note(const std::string& name, bool verbose) { if (verbose) std::printf("Saving data to: %s\n", name.c_str()); }
  void save_columns(const std::string& name, const std::vector<std::vector<double>>& cols, size_t rows, bool verbose) const {
    note(name, verbose);
    std::ofstream f(name);
    f.precision(15);
    for (size_t r = 0; r < rows; ++r)
      for (int it = 0; it < _niters; ++it)
        f << (it < (int)cols.size() && r < cols[it].size() ? cols[it][r] : 0.0) << (it + 1 < _niters ? ", " : "\n");
  }
  int _niters = 0, _dim_state = 0, _nstates = 0, _nfactors = 0, _cur_iter = 0;
  bool _keep_joint = true;
  std::string _file_mean{"mean.csv"}, _file_cov{"cov.csv"}, _file_joint_cov{"joint_cov.csv"}, _file_precision{"precision.csv"},
      _file_joint_precision{"joint_precision.csv"}, _file_zk_sdf{"zk_sdf.csv"}, _file_Sk_sdf{"Sk_sdf.csv"}, _file_cost{"cost.csv"},
      _file_factor_costs{"factor_costs.csv"};
};

// ------------------------------------------------------------------------------------------------
// Joint optimiser (gvibase/GVI-GH.h, gvibase/GVI-GH-impl.h, ngd/NGD-GH.h, ngd/NGD-GH-impl.h)
// ------------------------------------------------------------------------------------------------
enum class Execution { DeviceResident, FactorWise };

template <typename Factor>
class GVIGH {
 public:
  GVIGH(const std::vector<std::shared_ptr<Factor>>& vec_fact_optimizers, int dim_state, int num_states,
        int niterations = 5, double temperature = 1.0, double high_temperature = 100.0, int device = 0)
      : _dim_state(dim_state), _num_states(num_states), _dim(dim_state * num_states), _niters(niterations),
        _temperature(temperature), _high_temperature(high_temperature), _nfactors((int)vec_fact_optimizers.size()),
        _vec_factors(vec_fact_optimizers), _dev(std::make_shared<Device>(device)), _mu(VectorXd::Zero(dim_state * num_states)),
        _res_recorder(niterations, dim_state, num_states, (int)vec_fact_optimizers.size()) {
    _D.assign((size_t)num_states * dim_state * dim_state, 0.0); _SigD = _D;
    _U.assign((size_t)std::max(num_states - 1, 0) * dim_state * dim_state, 0.0); _SigU = _U;
    build_sets();
  }
  virtual ~GVIGH() {
    for (auto& f : _vec_factors) f->detach();            // the device sets die with the optimiser's context
  }

  // Which path serves optimize() / compute_gradients() / cost_value(): see the header comment.  FactorWise is forced
  // when a factor has no DevicePsi (opaque host psi).
  void set_execution(Execution e) {
    if (e == Execution::DeviceResident && !_all_device)
      throw GviError(GVI_ERR_UNSUPPORTED, "DeviceResident needs a DevicePsi on every factor (opaque host psi runs FactorWise)");
    _exec = e;
  }
  Execution execution() const { return _exec; }

  // setters (gvibase/GVI-GH.h:168-248)
  inline void set_step_size(double step_size) { _step_size = step_size; for (auto& f : _vec_factors) f->set_step_size(step_size); }
  inline void set_step_size_base(double v) { _step_size_base = v; }
  inline void set_max_iter_backtrack(double v) { _niters_backtrack = (int)v; }
  inline void set_niter_low_temperature(int v) { _niters_lowtemp = v; }
  inline void set_stop_err(double v) { _stop_err = v; }
  inline void set_temperature(double t) { _temperature = t; }
  inline void set_high_temperature(double t) { _high_temperature = t; }
  inline void update_file_names(const std::string& prefix = "", const std::string& afterfix = "") {   // :284-312
    _prefix = prefix;
    auto nm = [&](const char* base) { return prefix + base + (afterfix.empty() ? "" : "_" + afterfix) + ".csv"; };
    _res_recorder.update_file_names(nm("mean"), nm("cov"), nm("joint_cov"), nm("precision"), nm("joint_precision"), nm("cost"),
                                    nm("factor_costs"), nm("zk_sdf"), nm("Sk_sdf"));
    _save = true;
  }

  inline void set_mu(const VectorXd& mean) {                     // :182-187
    _mu = mean;
    for (auto& f : _vec_factors) f->update_mu_from_joint(_mu);
    _resident_stale = true;
  }
  inline void set_precision(const SpMat& new_precision) {        // gvibase/GVI-GH-impl.h:127-141
    to_blocks(new_precision, _D, _U);
    _dev->check(gvi_bt_marginals(_dev->get(), _D.data(), _U.data(), _SigD.data(), _SigU.data()));   // inverse_inplace()
    _prec_cache = new_precision; _prec_dirty = false; _cov_dirty = true;
    const SpMat& Cov = covariance_ref();
    for (auto& f : _vec_factors) f->update_precision_from_joint(Cov);
    _resident_stale = true;
  }
  inline void set_initial_values(const VectorXd& init_mean, const SpMat& init_precision) {   // :209-212
    if (_exec == Execution::DeviceResident) {
      std::vector<double> D, U;
      to_blocks(init_precision, D, U);
      _dev->check(gvi_ngd_init(_dev->get(), init_mean.data(), D.data(), U.data()));
      pull_state();
      const SpMat& Cov = covariance_ref();
      for (auto& f : _vec_factors) { f->update_mu_from_joint(_mu); f->update_precision_from_joint(Cov); }
      _resident_stale = false;
    } else {
      set_mu(init_mean);
      set_precision(init_precision);
    }
  }

  inline VectorXd mean() const { return _mu; }
  // joint precision / covariance as block-tridiagonal coefficient maps (built on demand from the block arrays)
  inline SpMat precision() const { return precision_ref(); }
  inline SpMat covariance() const { return covariance_ref(); }

  // inverse(mat) (gvibase/GVI-GH.h:161-165): the block-tridiagonal part of mat^-1 (EigenWrapper::inv_sparse /
  // inverse_GBP), computed by the device's selected inverse
  inline SpMat inverse(const SpMat& mat) {
    std::vector<double> D, U, SD((size_t)_num_states * _dim_state * _dim_state), SU((size_t)std::max(_num_states - 1, 0) * _dim_state * _dim_state);
    to_blocks(mat, D, U);
    _dev->check(gvi_bt_marginals(_dev->get(), D.data(), U.data(), SD.data(), SU.data()));
    return to_spmat(SD, SU);
  }

  void switch_to_high_temperature() {   // GVI-GH-impl.h:19-26
    for (auto& f : _vec_factors) f->factor_switch_to_high_temperature();
    _temperature = _high_temperature;
    push_temperatures();
  }

  // cost_value(mean, Precision) (gvibase/GVI-GH-impl.h:176-197): sum_k fact_cost_value + 1/2 log det, operator level
  double cost_value(const VectorXd& mean, const SpMat& Precision) {
    const SpMat Cov = inverse(Precision);
    double value = 0.0;
    for (auto& f : _vec_factors) value += f->fact_cost_value(mean, Cov);
    std::vector<double> D, U;
    to_blocks(Precision, D, U);
    double hld = 0.0;
    _dev->check(gvi_bt_logdet(_dev->get(), D.data(), U.data(), &hld));     // NaN when not PD -> the trial is rejected
    return value + hld;
  }
  VectorXd factor_cost_vector(const VectorXd& x, const SpMat& Precision) {   // :147-170
    VectorXd fac_costs(_nfactors);
    const SpMat joint_cov = inverse(Precision);
    for (int i = 0; i < _nfactors; ++i) fac_costs(i) = _vec_factors[i]->fact_cost_value(x, joint_cov);
    return fac_costs;
  }

  // cost_value() / factor_cost_vector() at the current proposal
  virtual double cost_value() {
    if (_exec == Execution::FactorWise) return cost_value(_mu, precision_ref());
    sync_resident();
    double c = 0.0;
    _dev->check(gvi_ngd_cost(_dev->get(), &c));
    return c;
  }
  virtual VectorXd factor_cost_vector() {   // GVI-GH-impl.h:147-170
    if (_exec == Execution::FactorWise) return factor_cost_vector(_mu, precision_ref());
    sync_resident();
    VectorXd out(_nfactors);
    for (size_t s = 0; s < _sets.size(); ++s) {
      std::vector<double> c(_sets[s].members.size());
      _dev->check(gvi_ngd_factor_costs(_dev->get(), (int)s, c.data()));
      for (size_t k = 0; k < c.size(); ++k) out(_sets[s].members[k]) = c[k];
    }
    return out;
  }
  virtual double cost_value_no_entropy() {   // ngd/NGD-GH-impl.h:179-190
    const SpMat& Cov = covariance_ref();
    double value = 0.0;
    for (auto& f : _vec_factors) value += f->fact_cost_value(_mu, Cov);
    return value;
  }

  // the reference's virtuals (gvibase/GVI-GH.h:121-128); NGDGH / ProxGVIGH override them
  virtual std::tuple<VectorXd, SpMat> compute_gradients(std::optional<double> step_size = std::nullopt) { (void)step_size; return {}; }
  virtual std::tuple<double, VectorXd, SpMat> onestep_linesearch(const double& step_size, const VectorXd& dmu, const SpMat& dprecision) {
    (void)step_size; (void)dmu; (void)dprecision;
    return {};
  }
  virtual inline void update_proposal(const VectorXd& new_mu, const SpMat& new_precision) { (void)new_mu; (void)new_precision; }

  // GVIGH::time_test of the reference's device variant (gvibase/GVI-GH-Cuda-impl.h:463-527): _niters + 1 timed
  // evaluations of the factor-cost vector at the current proposal (marginals + one cost pass over every factor),
  // the first discarded; same "% ..." lines.  Returns {average, min, max} in ms.
  std::tuple<double, double, double> time_test(bool print = true) {
    std::vector<double> times;
    for (int i = 0; i < _niters + 1; ++i) {
      // a fresh evaluation each round: re-setting the state drops the device-side cache of the cost
      _dev->check(gvi_ngd_init(_dev->get(), _mu.data(), _D.data(), _U.data()));
      const auto t0 = std::chrono::steady_clock::now();
      const Execution keep = _exec;
      _exec = Execution::DeviceResident; _resident_stale = false;
      (void)factor_cost_vector();
      _exec = keep;
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

  // GVIGH::optimize with backtracking (gvibase/GVI-GH-impl.h:33-124).  Statement for statement the reference's loop;
  // in DeviceResident mode the gradient / trial / accept steps stay on the device (the increments are not fetched).
  virtual void optimize(std::optional<bool> verbose = std::nullopt) {
    const bool is_verbose = verbose.value_or(true);
    const bool resident = _exec == Execution::DeviceResident;
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
      VectorXd fact_costs_iter = factor_cost_vector();
      _res_recorder.update_data_blocks(_mu, _SigD, _SigU, _D, _U, cost_iter, fact_costs_iter);
      VectorXd dmu;
      SpMat dprecision;
      if (resident) _dev->check(gvi_ngd_gradients(_dev->get()));
      else std::tie(dmu, dprecision) = compute_gradients();
      int cnt = 0;
      double step_size = _step_size_base;
      while (true) {
        step_size = step_size * 0.75;
        double new_cost = 0.0;
        VectorXd new_mu;
        SpMat new_precision;
        if (resident) _dev->check(gvi_ngd_trial(_dev->get(), step_size, &new_cost));
        else std::tie(new_cost, new_mu, new_precision) = onestep_linesearch(step_size, dmu, dprecision);
        if (new_cost < cost_iter) {
          if (resident) { _dev->check(gvi_ngd_accept(_dev->get())); pull_state(); }
          else update_proposal(new_mu, new_precision);
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
    if (_save) save_data(is_verbose);
  }

  // 1-D cost map (gvibase/GVI-GH.h:385-412)
  MatrixXd cost_map(double x_start, double x_end, double y_start, double y_end, int nmesh) {
    const double res_x = (x_end - x_start) / nmesh, res_y = (y_end - y_start) / nmesh;
    MatrixXd Z = MatrixXd::Zero(nmesh, nmesh);
    for (int i = 0; i < nmesh; i++)
      for (int j = 0; j < nmesh; j++) {
        const double m = x_start + i * res_x, p = y_start + j * res_y;
        _dev->check(gvi_ngd_init(_dev->get(), &m, &p, nullptr));
        double c = 0.0;
        _dev->check(gvi_ngd_cost(_dev->get(), &c));
        Z(j, i) = c;
      }
    _resident_stale = true;
    return Z;
  }
  void save_costmap(const std::string& filename = "costmap.csv") {
    MatrixXd m = cost_map(18, 25, 0.05, 1, 40);
    std::ofstream f(filename);
    f.precision(15);
    for (int r = 0; r < m.rows(); ++r)
      for (int c = 0; c < m.cols(); ++c) f << m(r, c) << (c + 1 < m.cols() ? ", " : "\n");
  }

  void save_data(bool verbose = true) {
    if (verbose) std::printf("=========== Saving Data ===========\n");
    _res_recorder.save_data(verbose);
  }
  const VIMPResults& results() const { return _res_recorder; }
  // device calls issued by the factor-wise path so far (one per set per pass when the batches hit)
  long factorwise_device_calls() const { long n = 0; for (auto& b : _batches) n += b->device_calls(); return n; }

 protected:
  struct Set { int d, p; std::vector<int> members; };

  void build_sets() {
    // homogeneous device sets in first-appearance order (d, GH degree, psi kind, shared field, closed form)
    _all_device = true;
    for (size_t i = 0; i < _vec_factors.size(); ++i) {
      auto& f = _vec_factors[i];
      const DevicePsi& dp = f->device_psi();
      if (dp.kind == GVI_PSI_HOST_CALLBACK) _all_device = false;
      size_t s = 0;
      for (; s < _sets.size(); ++s) {
        const auto& f0 = _vec_factors[_sets[s].members[0]];
        if (_sets[s].d == f->_dim && _sets[s].p == f->gh_degree() && f0->device_psi().same_group(dp) &&
            f0->closed_form() == f->closed_form() && f0->weights_map() == f->weights_map())
          break;
      }
      if (s == _sets.size()) _sets.push_back({f->_dim, f->gh_degree(), {}});
      _sets[s].members.push_back((int)i);
    }
    _dev->check(gvi_chain_set(_dev->get(), _num_states, _dim_state));
    for (auto& st : _sets) {
      std::vector<GVIFactorizedBase*> members;
      for (int i : st.members) members.push_back(_vec_factors[i].get());
      auto batch = make_factor_batch(_dev, members);
      for (size_t k = 0; k < members.size(); ++k) members[k]->attach(batch, (int)k);
      _batches.push_back(batch);
    }
    _exec = _all_device ? Execution::DeviceResident : Execution::FactorWise;
  }
  void push_temperatures() {
    for (size_t s = 0; s < _sets.size(); ++s) {
      std::vector<double> temp;
      for (int i : _sets[s].members) temp.push_back(_vec_factors[i]->temperature());
      _dev->check(gvi_factors_set_temperature(_dev->get(), (int)s, temp.data()));
    }
  }
  // host (mu, precision) changed through set_mu / set_precision / update_proposal: reload the resident state
  void sync_resident() {
    if (!_resident_stale) return;
    _dev->check(gvi_ngd_init(_dev->get(), _mu.data(), _D.data(), _U.data()));
    _resident_stale = false;
  }
  void pull_state() {
    const int T = _num_states, n = _dim_state;
    _mu = VectorXd(T * n);
    _dev->check(gvi_ngd_get_state(_dev->get(), _mu.data(), _D.data(), _U.data(), _SigD.data(), _SigU.data()));
    _prec_dirty = _cov_dirty = true;
    _resident_stale = false;
  }
  const SpMat& precision_ref() const {
    if (_prec_dirty) { _prec_cache = to_spmat(_D, _U); _prec_dirty = false; }
    return _prec_cache;
  }
  const SpMat& covariance_ref() const {
    if (_cov_dirty) { _cov_cache = to_spmat(_SigD, _SigU); _cov_dirty = false; }
    return _cov_cache;
  }
  void to_blocks(const SpMat& m, std::vector<double>& D, std::vector<double>& U) const {
    const int T = _num_states, n = _dim_state;
    D.assign((size_t)T * n * n, 0.0);
    U.assign((size_t)std::max(T - 1, 0) * n * n, 0.0);
    for (const auto& e : m.entries()) {
      const int i = e.first.first, j = e.first.second, ti = i / n, tj = j / n;
      if (ti == tj) D[((size_t)ti * n + i % n) * n + j % n] = e.second;
      else if (tj == ti + 1) U[((size_t)ti * n + i % n) * n + j % n] = e.second;
    }
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

  int _dim_state, _num_states, _dim, _niters;
  int _niters_lowtemp = 10, _niters_backtrack = 10;      // gvibase/GVI-GH.h:51-53
  double _stop_err = 1e-5, _temperature, _high_temperature;
  double _step_size = 0.9, _step_size_base = 0.55;       // gvibase/GVI-GH.h:92-93
  int _nfactors;
  std::vector<std::shared_ptr<Factor>> _vec_factors;
  std::shared_ptr<Device> _dev;
  std::vector<Set> _sets;
  std::vector<std::shared_ptr<FactorBatch>> _batches;
  VectorXd _mu;
  std::vector<double> _D, _U, _SigD, _SigU;              // joint precision / tridiagonal covariance blocks (authoritative)
  mutable SpMat _prec_cache, _cov_cache;
  mutable bool _prec_dirty = true, _cov_dirty = true;
  VIMPResults _res_recorder;
  std::string _prefix;
  bool _save = false, _all_device = true, _resident_stale = true;
  Execution _exec = Execution::DeviceResident;
};

// NGDGH (ngd/NGD-GH.h:25-95, ngd/NGD-GH-impl.h): the natural-gradient update law.
template <typename Factor>
class NGDGH : public GVIGH<Factor> {
  using Base = GVIGH<Factor>;
 public:
  using Base::Base;
  using Base::cost_value;
  using Base::factor_cost_vector;
  // compute_gradients (ngd/NGD-GH-impl.h:21-63): returns (dmu, dprecision)
  std::tuple<VectorXd, SpMat> compute_gradients(std::optional<double> step_size = std::nullopt) override {
    (void)step_size;
    const int T = this->_num_states, n = this->_dim_state;
    if (this->_exec == Execution::DeviceResident) {
      this->sync_resident();
      this->_dev->check(gvi_ngd_gradients(this->_dev->get()));
      VectorXd dmu(T * n), g(T * n);
      std::vector<double> dD((size_t)T * n * n), dU((size_t)std::max(T - 1, 0) * n * n), VD(dD.size()), VU(dU.size());
      this->_dev->check(gvi_ngd_get_gradients(this->_dev->get(), dmu.data(), dD.data(), dU.data(), g.data(), VD.data(), VU.data()));
      _Vdmu = g;
      _Vddmu = this->to_spmat(VD, VU);
      return std::make_tuple(dmu, this->to_spmat(dD, dU));
    }
    // the reference's joint loop: per-factor operator calls, summed in factor order (the OpenMP partial sums of
    // :31-52 differ from this order only by rounding)
    VectorXd Vdmu_sum = VectorXd::Zero(this->_dim);
    SpMat Vddmu_sum(this->_dim, this->_dim);
    for (auto& opt_k : this->_vec_factors) {
      opt_k->calculate_partial_V();
      Vdmu_sum += opt_k->local2joint_dmu_insertion();
      Vddmu_sum += opt_k->local2joint_dprecision_insertion();
    }
    _Vdmu = Vdmu_sum;
    _Vddmu = Vddmu_sum;
    SpMat dprecision = _Vddmu - this->precision_ref();
    // dmu = Vddmu^-1 (-Vdmu): Eigen's ConjugateGradient in the reference (:59-60), the device's direct
    // block-tridiagonal solve here
    std::vector<double> D, U;
    this->to_blocks(_Vddmu, D, U);
    const VectorXd rhs = -_Vdmu;
    VectorXd dmu(T * n);
    this->_dev->check(gvi_bt_solve(this->_dev->get(), D.data(), U.data(), rhs.data(), dmu.data()));
    return std::make_tuple(dmu, dprecision);
  }
  // onestep_linesearch (ngd/NGD-GH-impl.h:130-148)
  std::tuple<double, VectorXd, SpMat> onestep_linesearch(const double& step_size, const VectorXd& dmu, const SpMat& dprecision) override {
    VectorXd new_mu = this->_mu + step_size * dmu;
    SpMat new_precision = this->precision_ref() + step_size * dprecision;
    const double new_cost = Base::cost_value(new_mu, new_precision);
    return std::make_tuple(new_cost, new_mu, new_precision);
  }
  // update_proposal (ngd/NGD-GH-impl.h:151-156)
  inline void update_proposal(const VectorXd& new_mu, const SpMat& new_precision) override {
    Base::set_mu(new_mu);
    Base::set_precision(new_precision);
  }
  // device-resident forms: the trial and the increments stay in HBM (gvi_ngd_trial / gvi_ngd_accept)
  double onestep_linesearch(double step_size) {
    double c = 0.0;
    this->_dev->check(gvi_ngd_trial(this->_dev->get(), step_size, &c));
    return c;
  }
  void update_proposal() {
    this->_dev->check(gvi_ngd_accept(this->_dev->get()));
    this->pull_state();
  }
  inline VectorXd Vdmu() const { return _Vdmu; }       // ngd/NGD-GH.h:90-92
  inline SpMat Vddmu() const { return _Vddmu; }
 protected:
  VectorXd _Vdmu;
  SpMat _Vddmu;
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
  std::tuple<VectorXd, SpMat> compute_gradients(std::optional<double> step_size = std::nullopt) override {
    this->sync_resident();
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
      this->_res_recorder.update_data_blocks(this->_mu, this->_SigD, this->_SigU, this->_D, this->_U, cost_iter, fact_costs);
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
    if (this->_save) this->save_data(is_verbose);
  }
};

}  // namespace gvi
