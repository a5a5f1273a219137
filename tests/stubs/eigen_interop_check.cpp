// Compile-and-run check of the Eigen interop section of include/gvi/gvi_host.hpp (implicit conversions between the
// shim's VectorXd / MatrixXd / SpMat and Eigen's) against tests/stubs/eigen_api_subset.  Host only: no device call.
#include <cstdio>

#include "gvi/gvi_host.hpp"

#ifndef GVI_HOST_HAVE_EIGEN
#error "the Eigen interop section was not enabled"
#endif

static double use_shim_vector(const gvi::VectorXd& v) { return v(1); }               // reference-style call sites:
static double eigen_cost(const Eigen::VectorXd& x, const gvi::NoneType&) { return x(0) * x(0); }   // cost on Eigen types

int main() {
  Eigen::VectorXd ev(3);
  ev(0) = 1; ev(1) = 2; ev(2) = 3;
  gvi::VectorXd gv = ev;                                   // Eigen -> shim
  Eigen::VectorXd back = gv;                               // shim -> Eigen
  int bad = 0;
  for (int i = 0; i < 3; ++i) bad += back(i) != ev(i);
  bad += use_shim_vector(ev) != 2.0;                       // implicit at a call site
  Eigen::MatrixXd em(2, 3);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) em(i, j) = 10 * i + j;
  gvi::MatrixXd gm = em;                                   // column-major -> row-major
  bad += gm.rows() != 2 || gm.cols() != 3 || gm(1, 2) != 12 || gm.data()[1 * 3 + 2] != 12;
  Eigen::MatrixXd em2 = gm;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) bad += em2(i, j) != em(i, j);
  Eigen::SparseMatrix<double> es(4, 4);
  es.coeffRef(0, 0) = 1.5; es.coeffRef(2, 1) = -2.0; es.coeffRef(1, 2) = -2.0;
  gvi::SpMat gs = es;
  bad += gs.coeff(2, 1) != -2.0 || gs.coeff(0, 0) != 1.5 || gs.coeff(3, 3) != 0.0 || gs.rows() != 4;
  Eigen::SparseMatrix<double> es2 = gs;
  bad += es2.coeffRef(1, 2) != -2.0;
  // a cost function written on Eigen::VectorXd binds to the factor's std::function<double(const gvi::VectorXd&, ...)>
  gvi::NGDFactorizedSimpleGH::Function f = eigen_cost;
  gvi::VectorXd x = gvi::VectorXd::Constant(1, 3.0);
  bad += f(x, gvi::NoneType{}) != 9.0;
  std::printf(bad ? "FAIL %d\n" : "ok\n", bad);
  return bad != 0;
}
