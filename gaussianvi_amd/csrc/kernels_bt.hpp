// Joint-level (block-tridiagonal) device kernels: scatter/assemble, axpy, chain factorisation
// (log-det + tridiagonal blocks of the inverse), chain solve, marginal gather, cost sums.
//
// Replaces, per SURVEY.md section 8(a):
//   a10/a12  local2joint_*_insertion + sparse "+=" (ngd/NGDFactorizedBaseGH.h:91-106, ngd/NGD-GH-impl.h:39-55)
//   a12      ConjugateGradient solve (ngd/NGD-GH-impl.h:59-60)            -> bt_solve_kernel
//   a14      SimplicialLDLT log-det (gvibase/GVI-GH-impl.h:192-196)       -> bt_factor_kernel
//   a16/a17  inv_sparse / inverse_GBP (helpers/EigenWrapper.h:282-381,
//            gvibase/GVI-GH-GBP-impl.h:246-342)                           -> bt_factor_kernel
//   a11      extract_*_from_joint (gvibase/GVIFactorizedBase.h:104-122)   -> gather_kernel
//
// Chain recursions (blocks n x n, S_0 = D_0):
//   forward   Gauss-Jordan on [S_i | I | U_i] -> pivots (= natural-order LDL^T pivots), S_i^-1,
//             W_i = S_i^-1 U_i;   S_{i+1} = D_{i+1} - U_i^T W_i
//   backward  Sig_{T-1,T-1} = S_{T-1}^-1;  G = W_i Sig_{i+1,i+1};  Sig_{i,i+1} = -G;
//             Sig_{ii} = S_i^-1 + G W_i^T          (block form of the Takahashi recursion)
// One wave walks the chain; blocks live in LDS, the next step's operands are prefetched into
// registers while the current step computes.  Latency-bound by construction (T dependent steps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gvi {

constexpr int BT_MAX_N = 16;                       // block size limit of the LDS/regs budget
constexpr int BT_EPL = (3 * BT_MAX_N * BT_MAX_N + 63) / 64;   // elements per lane of an n x 3n tile

// ---- assemble: one thread per output element, ordered gather over the factors of a state ----
struct ScatterArgs {
  int T, n, d, K;
  const int32_t* start;      // [K]
  const int32_t* ptr;        // [T+1] CSR over states: factors with start == t, ascending k
  const int32_t* idx;        // [K]
  const double* Vdmu;        // [K][d]
  const double* Vddmu;       // [K][d][d]
  double* g;                 // [T][n]
  double* D;                 // [T][n][n]
  double* U;                 // [T-1][n][n]
};

__global__ __launch_bounds__(256) void bt_scatter_kernel(ScatterArgs a) {
  const int n = a.n, nn = n * n, d = a.d, per = n + 2 * nn;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)a.T * per) return;
  const int t = (int)(gid / per), e = (int)(gid % per);
  const bool two = d == 2 * n;
  double s = 0.0;
  if (e < n) {                                    // g[t][e]
    for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q) s += a.Vdmu[(size_t)a.idx[q] * d + e];
    if (two && t > 0)
      for (int q = a.ptr[t - 1]; q < a.ptr[t]; ++q) s += a.Vdmu[(size_t)a.idx[q] * d + n + e];
    a.g[(size_t)t * n + e] += s;
  } else if (e < n + nn) {                        // D[t][r][c]
    const int r = (e - n) / n, c = (e - n) % n;
    for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q) s += a.Vddmu[(size_t)a.idx[q] * d * d + r * d + c];
    if (two && t > 0)
      for (int q = a.ptr[t - 1]; q < a.ptr[t]; ++q)
        s += a.Vddmu[(size_t)a.idx[q] * d * d + (n + r) * d + n + c];
    a.D[(size_t)t * nn + r * n + c] += s;
  } else if (two && t < a.T - 1) {                // U[t][r][c]
    const int r = (e - n - nn) / n, c = (e - n - nn) % n;
    for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q)
      s += a.Vddmu[(size_t)a.idx[q] * d * d + r * d + n + c];
    a.U[(size_t)t * nn + r * n + c] += s;
  }
}

// out = x + alpha * y   (trial point: mu + step dmu, Lambda + step dLambda)
__global__ __launch_bounds__(256) void axpy_kernel(int64_t n, double alpha, const double* __restrict__ x,
                                                   const double* __restrict__ y, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] + alpha * y[i];
}
// out = x - y   (dprecision = Vddmu - Lambda), and optional negation (rhs = -Vdmu)
__global__ __launch_bounds__(256) void sub_kernel(int64_t n, const double* __restrict__ x,
                                                  const double* __restrict__ y, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] - (y ? y[i] : 2.0 * x[i]);
}

// ---- Gauss-Jordan on an n x nc tile in LDS by one wave ----
// Reads for a pivot step are completed into registers before anything is written (barrier between).
// PIVOT: partial (row) pivoting folded into the reads as a row permutation of the old tile.
// Returns the product-relevant info through *logsum (sum of log pivots) and *bad (pivot <= 0).
template <bool PIVOT>
__device__ inline void gauss_jordan(double* Ts, int n, int nc, int lane, double& logsum, int& bad) {
  const int total = n * nc;
  for (int p = 0; p < n; ++p) {
    int rs = p;
    if (PIVOT) {
      double best = fabs(Ts[p * nc + p]);
      for (int r = p + 1; r < n; ++r) {
        const double v = fabs(Ts[r * nc + p]);
        if (v > best) { best = v; rs = r; }
      }
    }
    const double piv = Ts[rs * nc + p];
    if (!(piv > 0.0)) bad = 1;
    logsum += log(piv);
    const double ipiv = 1.0 / piv;
    double nv[BT_EPL];
#pragma unroll
    for (int q = 0; q < BT_EPL; ++q) {
      const int e = lane + q * 64;
      if (e < total) {
        const int r = e / nc, c = e % nc;
        const int rr = (r == p) ? rs : ((r == rs) ? p : r);     // row swap p <-> rs
        const double prc = Ts[rs * nc + c] * ipiv;              // scaled pivot row
        nv[q] = (r == p) ? prc : Ts[rr * nc + c] - Ts[rr * nc + p] * prc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BT_EPL; ++q) {
      const int e = lane + q * 64;
      if (e < total) Ts[e] = nv[q];
    }
    __syncthreads();
  }
}

// ---- chain factorisation: log-det/2 and the tridiagonal blocks of the inverse ----
struct FactorArgs {
  int T, n;
  const double* D;       // [T][n][n]
  const double* U;       // [T-1][n][n]
  double* Wbuf;          // [T][n][n] workspace: W_i
  double* Ibuf;          // [T][n][n] workspace: S_i^-1
  double* SigD;          // [T][n][n] or null (log-det only)
  double* SigU;          // [T-1][n][n]
  double* half_logdet;   // [1]; NaN when not positive definite
};

__global__ __launch_bounds__(64) void bt_factor_kernel(FactorArgs a) {
  extern __shared__ double sm[];
  const int n = a.n, nn = n * n, nc = 3 * n, lane = threadIdx.x, T = a.T;
  double* Ts = sm;                 // [n][3n]  [S | I | U]
  double* Sn = Ts + n * nc;        // [n][n]   next Schur complement
  double* Us = Sn + nn;            // [n][n]   U_i
  double* Sg = Us + nn;            // [n][n]   Sig_{i+1,i+1} (backward)
  double* Gs = Sg + nn;            // [n][n]
  constexpr int EPB = (BT_MAX_N * BT_MAX_N + 63) / 64;   // elements per lane of an n x n block
  double logsum = 0.0;
  int bad = 0;
  double pd[EPB], pu[EPB];
  for (int q = 0; q < EPB; ++q) { pd[q] = 0.0; pu[q] = 0.0; }
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int e = lane + q * 64;
    if (e < nn) { Sn[e] = a.D[e]; pu[q] = T > 1 ? a.U[e] : 0.0; }
  }
  __syncthreads();
  for (int i = 0; i < T; ++i) {
    const bool more = i + 1 < T;
    // prefetch D_{i+1}, U_{i+1} while this step computes
    double nd[EPB], nu[EPB];
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      nd[q] = 0.0; nu[q] = 0.0;
      if (e < nn) {
        if (more) nd[q] = a.D[(size_t)(i + 1) * nn + e];
        if (i + 2 < T) nu[q] = a.U[(size_t)(i + 1) * nn + e];
      }
    }
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      if (e < nn) {
        const int r = e / n, c = e % n;
        Ts[r * nc + c] = Sn[e];
        Ts[r * nc + n + c] = r == c ? 1.0 : 0.0;
        Ts[r * nc + 2 * n + c] = pu[q];
        Us[e] = pu[q];
      }
    }
    __syncthreads();
    gauss_jordan<false>(Ts, n, nc, lane, logsum, bad);
    if (a.SigD) {
#pragma unroll
      for (int q = 0; q < EPB; ++q) {
        const int e = lane + q * 64;
        if (e < nn) {
          const int r = e / n, c = e % n;
          a.Ibuf[(size_t)i * nn + e] = Ts[r * nc + n + c];
          a.Wbuf[(size_t)i * nn + e] = Ts[r * nc + 2 * n + c];
        }
      }
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < EPB; ++q) {
        const int e = lane + q * 64;
        if (e < nn) {
          const int r = e / n, c = e % n;
          double s = nd[q];
          for (int k = 0; k < n; ++k) s -= Us[k * n + r] * Ts[k * nc + 2 * n + c];
          Sn[e] = s;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPB; ++q) pu[q] = nu[q];
  }
  if (lane == 0) a.half_logdet[0] = bad ? __builtin_nan("") : 0.5 * logsum;
  if (!a.SigD) return;
  // backward: Ts[:, n:2n] still holds S_{T-1}^-1
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int e = lane + q * 64;
    if (e < nn) {
      const int r = e / n, c = e % n;
      const double v = Ts[r * nc + n + c];
      Sg[e] = v;
      a.SigD[(size_t)(T - 1) * nn + e] = v;
    }
  }
  double wv[EPB], iv[EPB];
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int e = lane + q * 64;
    wv[q] = 0.0; iv[q] = 0.0;
    if (e < nn && T > 1) { wv[q] = a.Wbuf[(size_t)(T - 2) * nn + e]; iv[q] = a.Ibuf[(size_t)(T - 2) * nn + e]; }
  }
  __syncthreads();
  for (int i = T - 2; i >= 0; --i) {
    double nw[EPB], ni[EPB];
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      nw[q] = 0.0; ni[q] = 0.0;
      if (e < nn && i > 0) { nw[q] = a.Wbuf[(size_t)(i - 1) * nn + e]; ni[q] = a.Ibuf[(size_t)(i - 1) * nn + e]; }
    }
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      if (e < nn) Us[e] = wv[q];                 // W_i
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPB; ++q) {              // G = W_i Sig_{i+1,i+1}
      const int e = lane + q * 64;
      if (e < nn) {
        const int r = e / n, c = e % n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += Us[r * n + k] * Sg[k * n + c];
        Gs[e] = s;
        a.SigU[(size_t)i * nn + e] = -s;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPB; ++q) {              // Sig_ii = S_i^-1 + G W_i^T
      const int e = lane + q * 64;
      if (e < nn) {
        const int r = e / n, c = e % n;
        double s = iv[q];
        for (int k = 0; k < n; ++k) s += Gs[r * n + k] * Us[c * n + k];
        a.SigD[(size_t)i * nn + e] = s;
        Sg[e] = s;                               // Sg was last read before the barrier above
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPB; ++q) { wv[q] = nw[q]; iv[q] = ni[q]; }
  }
}

// ---- chain solve A x = rhs, A = (D, U) symmetric block-tridiagonal (possibly indefinite) ----
struct SolveArgs {
  int T, n;
  const double* D;
  const double* U;
  const double* rhs;     // [T][n]
  double rhs_scale;      // x = A^-1 (rhs_scale * rhs)   (-1 for dmu = Vddmu^-1 (-Vdmu))
  double* Wbuf;          // [T][n][n] workspace
  double* vbuf;          // [T][n] workspace
  double* x;             // [T][n]
};

__global__ __launch_bounds__(64) void bt_solve_kernel(SolveArgs a) {
  extern __shared__ double sm[];
  const int n = a.n, nn = n * n, nc = 2 * n + 1, lane = threadIdx.x, T = a.T;
  double* Ts = sm;                 // [n][2n+1]  [S | U | y]
  double* Sn = Ts + n * nc;        // [n][n]
  double* Us = Sn + nn;            // [n][n]
  double* yn = Us + nn;            // [n]
  double* xs = yn + n;             // [n]
  constexpr int EPB = (BT_MAX_N * BT_MAX_N + 63) / 64;
  double logsum = 0.0;
  int bad = 0;
  double pu[EPB];
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int e = lane + q * 64;
    pu[q] = 0.0;
    if (e < nn) { Sn[e] = a.D[e]; pu[q] = T > 1 ? a.U[e] : 0.0; }
  }
  if (lane < n) yn[lane] = a.rhs_scale * a.rhs[lane];
  __syncthreads();
  for (int i = 0; i < T; ++i) {
    const bool more = i + 1 < T;
    double nd[EPB], nu[EPB], nr = 0.0;
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      nd[q] = 0.0; nu[q] = 0.0;
      if (e < nn) {
        if (more) nd[q] = a.D[(size_t)(i + 1) * nn + e];
        if (i + 2 < T) nu[q] = a.U[(size_t)(i + 1) * nn + e];
      }
    }
    if (more && lane < n) nr = a.rhs_scale * a.rhs[(size_t)(i + 1) * n + lane];
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      if (e < nn) {
        const int r = e / n, c = e % n;
        Ts[r * nc + c] = Sn[e];
        Ts[r * nc + n + c] = pu[q];
        Us[e] = pu[q];
      }
    }
    if (lane < n) Ts[lane * nc + 2 * n] = yn[lane];
    __syncthreads();
    gauss_jordan<true>(Ts, n, nc, lane, logsum, bad);
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      if (e < nn) {
        const int r = e / n, c = e % n;
        a.Wbuf[(size_t)i * nn + e] = Ts[r * nc + n + c];
      }
    }
    if (lane < n) a.vbuf[(size_t)i * n + lane] = Ts[lane * nc + 2 * n];
    if (more) {
#pragma unroll
      for (int q = 0; q < EPB; ++q) {
        const int e = lane + q * 64;
        if (e < nn) {
          const int r = e / n, c = e % n;
          double s = nd[q];
          for (int k = 0; k < n; ++k) s -= Us[k * n + r] * Ts[k * nc + n + c];
          Sn[e] = s;
        }
      }
      if (lane < n) {
        double s = nr;
        for (int k = 0; k < n; ++k) s -= Us[k * n + lane] * Ts[k * nc + 2 * n];
        yn[lane] = s;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPB; ++q) pu[q] = nu[q];
  }
  // backward: x_{T-1} = v_{T-1};  x_i = v_i - W_i x_{i+1}
  if (lane < n) {
    const double v = Ts[lane * nc + 2 * n];
    xs[lane] = v;
    a.x[(size_t)(T - 1) * n + lane] = v;
  }
  double wv[EPB], vv = 0.0;
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int e = lane + q * 64;
    wv[q] = 0.0;
    if (e < nn && T > 1) wv[q] = a.Wbuf[(size_t)(T - 2) * nn + e];
  }
  if (lane < n && T > 1) vv = a.vbuf[(size_t)(T - 2) * n + lane];
  __syncthreads();
  for (int i = T - 2; i >= 0; --i) {
    double nw[EPB], nv = 0.0;
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      nw[q] = 0.0;
      if (e < nn && i > 0) nw[q] = a.Wbuf[(size_t)(i - 1) * nn + e];
    }
    if (lane < n && i > 0) nv = a.vbuf[(size_t)(i - 1) * n + lane];
#pragma unroll
    for (int q = 0; q < EPB; ++q) {
      const int e = lane + q * 64;
      if (e < nn) Us[e] = wv[q];
    }
    __syncthreads();
    double xi = 0.0;
    if (lane < n) {
      xi = vv;
      for (int k = 0; k < n; ++k) xi -= Us[lane * n + k] * xs[k];
      a.x[(size_t)i * n + lane] = xi;
    }
    __syncthreads();
    if (lane < n) xs[lane] = xi;
#pragma unroll
    for (int q = 0; q < EPB; ++q) wv[q] = nw[q];
    vv = nv;
    __syncthreads();
  }
}

// ---- gather (mu_k, Sigma_k) of every factor from the joint mean / covariance blocks ----
__global__ __launch_bounds__(256) void gather_kernel(int K, int d, int n, const int32_t* __restrict__ start,
                                                     const double* __restrict__ mu, const double* __restrict__ SigD,
                                                     const double* __restrict__ SigU, double* __restrict__ mu_k,
                                                     double* __restrict__ Sigma_k) {
  const int per = d + d * d, nn = n * n;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)K * per) return;
  const int k = (int)(gid / per), e = (int)(gid % per), s = start[k];
  if (e < d) { mu_k[(size_t)k * d + e] = mu[(size_t)s * n + e]; return; }
  const int r = (e - d) / d, c = (e - d) % d;
  double v;
  if (r < n && c < n) v = SigD[(size_t)s * nn + r * n + c];
  else if (r >= n && c >= n) v = SigD[(size_t)(s + 1) * nn + (r - n) * n + (c - n)];
  else if (r < n) v = SigU[(size_t)s * nn + r * n + (c - n)];
  else v = SigU[(size_t)s * nn + c * n + (r - n)];
  Sigma_k[(size_t)k * d * d + r * d + c] = v;
}

// ---- ordered sum of a set's factor costs into acc[0] (acc += sum); one block, fixed tree ----
__global__ __launch_bounds__(256) void cost_sum_kernel(int K, const double* __restrict__ cost, double* acc,
                                                       int init) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) s += cost[k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[0] = (init ? 0.0 : acc[0]) + sh[0];
}

// total = cost_sum + half_logdet  (cost_value, gvibase/GVI-GH-impl.h:196)
__global__ void cost_total_kernel(const double* cost_sum, const double* half_logdet, double* total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = cost_sum[0] + half_logdet[0];
}

}  // namespace gvi
