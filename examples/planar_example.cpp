// A planar point-robot planning problem on the header-only shim: minimum-acceleration GP priors, one
// hinge-on-signed-distance obstacle factor per state and two end anchors -- the workload the reference's GPU path exists
// for (helpers/CudaOperation.h: PlanarSDF + CudaOperation_PlanarPR), here with the factors described as DevicePsi
// objects next to the reference-shaped constructors.  Prints the cost and the planned (x, y) mean after every NGD
// iteration; tests/test_cpp_shim.py builds the same problem through the Python binding and compares.
//   Usage: planar_example [iterations]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gvi/gvi_host.hpp"

using namespace gvi;

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 5;
  const int T = 17, n = 4, nd = 2, K = T - 1, p = 3;
  const double dt = 0.25, qc = 0.8;
  // MinimumAccGP blocks (gp/minimum_acc_prior.h:50-53, 103-116)
  MatrixXd Phi = MatrixXd::Identity(n, n), Qinv(n, n);
  for (int i = 0; i < nd; ++i) {
    Phi(i, nd + i) = dt;
    Qinv(i, i) = 12.0 / (dt * dt * dt) / qc;
    Qinv(i, nd + i) = Qinv(nd + i, i) = -6.0 / (dt * dt) / qc;
    Qinv(nd + i, nd + i) = 4.0 / dt / qc;
  }
  // signed distance to two discs on a 0.1 grid
  auto sdf = std::make_shared<PlanarSDF>();
  sdf->origin_x = -5.0; sdf->origin_y = -4.0; sdf->cell_size = 0.1;
  sdf->field = MatrixXd(81, 101);
  const double cx[2] = {0.0, -1.0}, cy[2] = {1.6, -2.2}, cr[2] = {1.2, 0.9};
  for (int r = 0; r < 81; ++r)
    for (int c = 0; c < 101; ++c) {
      const double x = -5.0 + c * 0.1, y = -4.0 + r * 0.1;
      double best = 1e300;
      for (int o = 0; o < 2; ++o) best = std::fmin(best, std::hypot(x - cx[o], y - cy[o]) - cr[o]);
      sdf->field(r, c) = best;
    }
  // nominal straight line, anchors, initial precision = sum of prior Hessians + anchors + 0.5 I
  const double sx = -3.0, sy = -0.4, gx = 3.0, gy = 0.4, horizon = (T - 1) * dt;
  const double vx = (gx - sx) / horizon, vy = (gy - sy) / horizon;
  VectorXd init_mu(T * n);
  for (int t = 0; t < T; ++t) {
    init_mu(t * n + 0) = sx + vx * t * dt; init_mu(t * n + 1) = sy + vy * t * dt;
    init_mu(t * n + 2) = vx; init_mu(t * n + 3) = vy;
  }
  MatrixXd Kinv = MatrixXd::Identity(n, n);
  for (int i = 0; i < n; ++i) Kinv(i, i) = 100.0;
  SpMat init_prec(T * n, T * n);
  {
    // M = [-Phi, I]^T Qinv [-Phi, I]
    MatrixXd Lam(n, 2 * n), M(2 * n, 2 * n);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) Lam(i, j) = -Phi(i, j); Lam(i, n + i) = 1.0; }
    for (int a = 0; a < 2 * n; ++a)
      for (int b = 0; b < 2 * n; ++b) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) s += Lam(i, a) * Qinv(i, j) * Lam(j, b);
        M(a, b) = s;
      }
    for (int k = 0; k < K; ++k)
      for (int a = 0; a < 2 * n; ++a)
        for (int b = 0; b < 2 * n; ++b) init_prec.coeffRef(k * n + a, k * n + b) += M(a, b);
    for (int i = 0; i < n; ++i) { init_prec.coeffRef(i, i) += 200.0; init_prec.coeffRef((T - 1) * n + i, (T - 1) * n + i) += 200.0; }
    for (int i = 0; i < T * n; ++i) init_prec.coeffRef(i, i) += 0.5;
  }
  using Factor = NGDFactorizedBaseGH<NoneType>;
  auto none = [](const VectorXd&, const NoneType&) { return 0.0; };          // psi lives on the device (DevicePsi)
  std::vector<std::shared_ptr<Factor>> factors;
  for (int k = 0; k < K; ++k)
    factors.emplace_back(new Factor(2 * n, n, p, none, NoneType{}, T, k, 1.0, 10.0, DevicePsi::QuadPrior(Phi, Qinv)));
  for (int t = 0; t < T; ++t)
    factors.emplace_back(new Factor(n, n, p + 1, none, NoneType{}, T, t, 1.0, 10.0, DevicePsi::HingeSdf2D(15.5, 0.5, 0.3, sdf)));
  for (int e = 0; e < 2; ++e) {
    const int t = e ? T - 1 : 0;
    VectorXd m0(n);
    for (int i = 0; i < n; ++i) m0(i) = init_mu(t * n + i);
    factors.emplace_back(new Factor(n, n, p, none, NoneType{}, T, t, 1.0, 10.0, DevicePsi::FixedPrior(m0, Kinv)));
  }
  NGDGH<Factor> opt{factors, n, T, iters};
  opt.set_niter_low_temperature(iters);
  opt.set_initial_values(init_mu, init_prec);
  for (int it = 0; it < iters; ++it) {
    const double c0 = opt.cost_value();
    opt.compute_gradients();
    double step = 0.55, c1 = c0;
    int cnt = 0;
    while (true) {
      step *= 0.75;
      c1 = opt.onestep_linesearch(step);
      if (c1 < c0 || ++cnt > 10) break;
    }
    if (c1 < c0) opt.update_proposal();
    std::printf("iter %d cost %.12f ->", it, c1);
    const VectorXd mu = opt.mean();
    for (int t = 0; t < T; t += 4) std::printf(" (%.9f, %.9f)", mu(t * n), mu(t * n + 1));
    std::printf("\n");
  }
  return 0;
}
