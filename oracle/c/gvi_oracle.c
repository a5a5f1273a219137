/*
 * CPU restatement (plain C, fp64, OpenMP over factors) of the per-factor hot path -- TEST
 * INFRASTRUCTURE and bench.py's cpu_baseline leg only; the product never links or calls this.
 *
 * Follows the reference's structure (kind "port" in bench.py's cpu_baseline):
 *   updateGH:            symmetric sqrt of Sigma_k + expand X = Z S^T + mu   quadrature/SparseGaussHermite.h:231-243
 *   Integrate x 3:       three separate passes over the N sigma points, psi re-evaluated in each
 *                        through a function pointer (the reference's std::function closures
 *                        _func_Vmu, _func_phi, _func_Vmumu)                   quadrature/SparseGaussHermite.h:197-221,
 *                                                                             ngd/NGDFactorizedBaseGH.h:46-48
 *   calculate_partial_V: Vdmu = Lam E1 / T, Vddmu = sym_upper(Lam E2 Lam - Lam E0) / T   ngd/NGDFactorizedBaseGH.h:53-74
 *   OpenMP over factors                                                       ngd/NGD-GH-impl.h:31-44
 * psi kinds: the 1-D range factor, the two quadratic priors and the planar hinge-on-SDF obstacle cost.
 * `fused = 1` is the best-effort CPU variant (one pass, psi once per point, upper triangle only).
 * Pinned against oracle/gvi_oracle.py (itself pinned by K1-K9) in tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { PSI_RANGE_1D = 0, PSI_QUAD_PRIOR = 1, PSI_FIXED_PRIOR = 2, PSI_HINGE_SDF_2D = 4 };

typedef struct { int kind, d, n; const double* p; } psi_closure;

/* signed-distance grid of the hinge kinds (set once by gvi_oracle_set_sdf2d; read-only afterwards): field[r + c rows] */
static struct { const double* field; int rows, cols; double ox, oy, cell; } g_sdf;

/* PlanarSDF::convertPoint2toCell + signed_distance (helpers/CudaOperation.h:61-103): clamp the query to the grid, bilinear
 * interpolation of the column-major field (data_array_[r + c rows], :130); the upper index is clamped where its weight is 0 */
static double sdf2d_lookup(double px, double py) {
  const double xmax = g_sdf.ox + (g_sdf.cols - 1.0) * g_sdf.cell, ymax = g_sdf.oy + (g_sdf.rows - 1.0) * g_sdf.cell;
  const double xin = px < g_sdf.ox ? g_sdf.ox : (px > xmax ? xmax : px);
  const double yin = py < g_sdf.oy ? g_sdf.oy : (py > ymax ? ymax : py);
  const double col = (xin - g_sdf.ox) / g_sdf.cell, row = (yin - g_sdf.oy) / g_sdf.cell;
  const double lr = floor(row), lc = floor(col), hr = lr + 1.0, hc = lc + 1.0;
  const int lri = (int)lr, lci = (int)lc, R = g_sdf.rows;
  const int hri = lri + 1 < g_sdf.rows ? lri + 1 : g_sdf.rows - 1;
  const int hci = lci + 1 < g_sdf.cols ? lci + 1 : g_sdf.cols - 1;
  const double* f = g_sdf.field;
  return (hr - row) * (hc - col) * f[lri + lci * R] + (row - lr) * (hc - col) * f[hri + lci * R] +
         (hr - row) * (col - lc) * f[lri + hci * R] + (row - lr) * (col - lc) * f[hri + hci * R];
}

/* src/1d_example.cpp:25-35; gp/minimum_acc_prior.h:103-106 / gp/LTV_prior.h:223-226; gp/fixed_prior.h:28-30 */
static double psi_eval(const double* x, const psi_closure* c) {
  if (c->kind == PSI_HINGE_SDF_2D) {
    /* cost_obstacle_planar of the planar point robot (helpers/CudaOperation.h:491-508): one ball, slope 1;
       p = (sigma, eps, r): sigma * max(0, eps + r - sdf(x0, x1))^2 */
    const double* p = c->p;
    const double sd = sdf2d_lookup(x[0], x[1]), thr = p[1] + p[2];
    const double err = sd > thr ? 0.0 : thr - sd;
    return err * err * p[0];
  }
  if (c->kind == PSI_RANGE_1D) {
    const double* p = c->p; /* y, mu_p, fb, sig_r_sq, sig_p_sq */
    double e = x[0] - p[1], r = p[0] - p[2] / x[0];
    return e * e / p[4] / 2 + r * r / p[3] / 2;
  }
  if (c->kind == PSI_QUAD_PRIOR) {
    int n = c->n;
    const double *Phi = c->p, *Qinv = c->p + n * n;
    double r[32], acc = 0.0;
    for (int i = 0; i < n; ++i) {
      double s = -x[n + i];
      for (int j = 0; j < n; ++j) s += Phi[i * n + j] * x[j];
      r[i] = s;
    }
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += Qinv[i * n + j] * r[j];
      acc += r[i] * s;
    }
    return acc / 2;
  }
  {
    int d = c->d;
    const double *mu0 = c->p, *Kinv = c->p + d;
    double e[64], acc = 0.0;
    for (int i = 0; i < d; ++i) e[i] = x[i] - mu0[i];
    for (int i = 0; i < d; ++i) {
      double s = 0.0;
      for (int j = 0; j < d; ++j) s += Kinv[i * d + j] * e[j];
      acc += e[i] * s;
    }
    return acc;
  }
}

/* cyclic Jacobi: A (d x d, symmetric, destroyed) -> eigenvalues lam, eigenvectors in columns of V */
static void jacobi(int d, double* A, double* lam, double* V) {
  for (int i = 0; i < d * d; ++i) V[i] = 0.0;
  for (int i = 0; i < d; ++i) V[i * d + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int i = 0; i < d; ++i) {
      dg += A[i * d + i] * A[i * d + i];
      for (int j = i + 1; j < d; ++j) off += A[i * d + j] * A[i * d + j];
    }
    if (off <= 1e-34 * dg) break;
    for (int p = 0; p < d - 1; ++p)
      for (int q = p + 1; q < d; ++q) {
        double apq = A[p * d + q];
        if (apq == 0.0) continue;
        double th = (A[q * d + q] - A[p * d + p]) / (2.0 * apq);
        double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < d; ++k) {
          double a = A[k * d + p], b = A[k * d + q];
          A[k * d + p] = c * a - s * b; A[k * d + q] = s * a + c * b;
        }
        for (int k = 0; k < d; ++k) {
          double a = A[p * d + k], b = A[q * d + k];
          A[p * d + k] = c * a - s * b; A[q * d + k] = s * a + c * b;
        }
        for (int k = 0; k < d; ++k) {
          double a = V[k * d + p], b = V[k * d + q];
          V[k * d + p] = c * a - s * b; V[k * d + q] = s * a + c * b;
        }
      }
  }
  for (int i = 0; i < d; ++i) lam[i] = A[i * d + i];
}

static void matmul(int d, const double* A, const double* B, double* C) {
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += A[i * d + k] * B[k * d + j];
      C[i * d + j] = s;
    }
}

/* Returns 0.  Z [N][d] row-major, mu [K][d], Sigma [K][d][d], params [K][pstride]. */
int gvi_oracle_moments(int K, int d, int n, long N, const double* Z, const double* w, const double* mu,
                       const double* Sigma, int kind, const double* params, int pstride,
                       const double* temperature, int fused, int nthreads,
                       double* Ephi, double* Vdmu, double* Vddmu) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* X = (double*)malloc(sizeof(double) * (size_t)N * d);
    double* buf = (double*)malloc(sizeof(double) * (size_t)(8 * d * d + 4 * d));
    double *A = buf, *V = A + d * d, *S = V + d * d, *Lam = S + d * d, *E2 = Lam + d * d, *T1 = E2 + d * d,
           *T2 = T1 + d * d, *tmp = T2 + d * d, *lam = tmp + d * d, *E1 = lam + d, *y = E1 + d;
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < K; ++k) {
      const double *mk = mu + (size_t)k * d, *Sk = Sigma + (size_t)k * d * d;
      psi_closure cl = {kind, d, n, params + (size_t)k * pstride};
      const double Tk = temperature ? temperature[k] : 1.0;
      /* updateGH: symmetric sqrt (V sqrt(lam) V^T), precision (V lam^-1 V^T), expand */
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[i * d + j] = i >= j ? Sk[i * d + j] : Sk[j * d + i];
      jacobi(d, A, lam, V);
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
          double s = 0.0, l = 0.0;
          for (int c = 0; c < d; ++c) {
            s += V[i * d + c] * sqrt(lam[c]) * V[j * d + c];
            l += V[i * d + c] / lam[c] * V[j * d + c];
          }
          S[i * d + j] = s; Lam[i * d + j] = l;
        }
      for (long i = 0; i < N; ++i)
        for (int a = 0; a < d; ++a) {
          double s = mk[a];
          for (int b = 0; b < d; ++b) s += Z[(size_t)i * d + b] * S[a * d + b];
          X[(size_t)i * d + a] = s;
        }
      double E0 = 0.0;
      for (int a = 0; a < d; ++a) E1[a] = 0.0;
      for (int a = 0; a < d * d; ++a) E2[a] = 0.0;
      if (!fused) {
        /* Integrate(_func_Vmu) */
        for (long i = 0; i < N; ++i) {
          const double* x = X + (size_t)i * d;
          double ps = psi_eval(x, &cl);
          for (int a = 0; a < d; ++a) tmp[a] = (x[a] - mk[a]) * ps;
          for (int a = 0; a < d; ++a) E1[a] += tmp[a] * w[i];
        }
        /* Integrate(_func_phi) */
        for (long i = 0; i < N; ++i) E0 += psi_eval(X + (size_t)i * d, &cl) * w[i];
        /* Integrate(_func_Vmumu): full d x d outer product per point */
        for (long i = 0; i < N; ++i) {
          const double* x = X + (size_t)i * d;
          double ps = psi_eval(x, &cl);
          for (int a = 0; a < d; ++a) y[a] = x[a] - mk[a];
          for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) tmp[a * d + b] = y[a] * y[b] * ps;
          for (int a = 0; a < d * d; ++a) E2[a] += tmp[a] * w[i];
        }
      } else {
        for (long i = 0; i < N; ++i) {
          const double* x = X + (size_t)i * d;
          double c = psi_eval(x, &cl) * w[i];
          E0 += c;
          for (int a = 0; a < d; ++a) {
            y[a] = x[a] - mk[a];
            double t = c * y[a];
            E1[a] += t;
            for (int b = 0; b <= a; ++b) E2[a * d + b] += t * y[b];
          }
        }
        for (int a = 0; a < d; ++a)
          for (int b = a + 1; b < d; ++b) E2[a * d + b] = E2[b * d + a];
      }
      /* calculate_partial_V */
      if (Ephi) Ephi[k] = E0;
      for (int a = 0; a < d; ++a) {
        double s = 0.0;
        for (int b = 0; b < d; ++b) s += Lam[a * d + b] * E1[b];
        if (Vdmu) Vdmu[(size_t)k * d + a] = s / Tk;
      }
      matmul(d, Lam, E2, T1);
      matmul(d, T1, Lam, T2);
      if (Vddmu)
        for (int a = 0; a < d; ++a)
          for (int b = a; b < d; ++b) {
            double v = (T2[a * d + b] - Lam[a * d + b] * E0) / Tk;
            Vddmu[(size_t)k * d * d + a * d + b] = v;
            Vddmu[(size_t)k * d * d + b * d + a] = v;
          }
    }
    free(X); free(buf);
  }
  return 0;
}

/* field: [rows x cols] column-major (r + c rows), kept by reference */
int gvi_oracle_set_sdf2d(const double* field, int rows, int cols, double ox, double oy, double cell) {
  g_sdf.field = field; g_sdf.rows = rows; g_sdf.cols = cols; g_sdf.ox = ox; g_sdf.oy = oy; g_sdf.cell = cell;
  return 0;
}

int gvi_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
