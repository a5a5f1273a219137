// Joint-level (block-tridiagonal) device kernels around the chain operations: scatter / assemble, axpy, marginal gather,
// cost sums, the exchange records of the sharded iteration.
//
// Replaces, per SURVEY.md section 8(a):
//   a10/a12  local2joint_*_insertion + sparse "+=" (ngd/NGDFactorizedBaseGH.h:91-106, ngd/NGD-GH-impl.h:39-55)
//   a11      extract_*_from_joint (gvibase/GVIFactorizedBase.h:104-122)   -> gather_kernel
//
// The chain operations themselves (solve, log-det, tridiagonal blocks of the inverse) are in kernels_chain.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace gvi {

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
// out = x - y   (dprecision = Vddmu - Lambda)
__global__ __launch_bounds__(256) void sub_kernel(int64_t n, const double* __restrict__ x,
                                                  const double* __restrict__ y, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] - y[i];
}

// ---- fused glue of the resident NGD iteration (one launch each instead of one per set) ----
constexpr int MAX_SETS = 8;
struct SetDesc {
  int K, d;
  const int32_t* start;
  const int32_t* ptr;        // CSR over states (scatter)
  const int32_t* idx;
  const double* Vdmu;        // [K][d]
  const double* Vddmu;       // [K][d][d]
  double* mu_k;              // [K][d]      (gather)
  double* Sigma_k;           // [K][d][d]
  const double* cost;        // [K]         (cost sum)
};
struct SetList { int nsets; SetDesc s[MAX_SETS]; };

// assemble over ALL sets: every output element is written once (ordered: set after set, factor
// index ascending), so no memset and no second launch
__global__ __launch_bounds__(256) void bt_scatter_all_kernel(SetList L, int T, int n, double* __restrict__ g,
                                                             double* __restrict__ D, double* __restrict__ U,
                                                             const double* pred, double pred_val,
                                                             double* __restrict__ rec, int rlo, int rlen) {
  // rec != null (sharded factors): the states [rlo, rlo + rlen) this rank's factors touch are ALSO written as exchange
  // records [state - rlo][n + 2 n^2] -- what dist_pack_kernel would copy out of (g, D, U) in a launch of its own
  if (pred_skip(pred, pred_val)) return;
  const int nn = n * n, per = n + 2 * nn;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)T * per) return;
  const int t = (int)(gid / per), e = (int)(gid % per);
  double acc = 0.0;
  for (int si = 0; si < L.nsets; ++si) {
    const SetDesc& a = L.s[si];
    const int d = a.d;
    const bool two = d == 2 * n;
    double s = 0.0;
    if (e < n) {
      for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q) s += a.Vdmu[(size_t)a.idx[q] * d + e];
      if (two && t > 0)
        for (int q = a.ptr[t - 1]; q < a.ptr[t]; ++q) s += a.Vdmu[(size_t)a.idx[q] * d + n + e];
    } else if (e < n + nn) {
      const int r = (e - n) / n, c = (e - n) % n;
      for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q) s += a.Vddmu[(size_t)a.idx[q] * d * d + r * d + c];
      if (two && t > 0)
        for (int q = a.ptr[t - 1]; q < a.ptr[t]; ++q)
          s += a.Vddmu[(size_t)a.idx[q] * d * d + (n + r) * d + n + c];
    } else if (two && t < T - 1) {
      const int r = (e - n - nn) / n, c = (e - n - nn) % n;
      for (int q = a.ptr[t]; q < a.ptr[t + 1]; ++q) s += a.Vddmu[(size_t)a.idx[q] * d * d + r * d + n + c];
    }
    acc += s;                 // same association as memset + one "+=" per set
  }
  if (e < n) g[(size_t)t * n + e] = acc;
  else if (e < n + nn) D[(size_t)t * nn + (e - n)] = acc;
  else if (t < T - 1) U[(size_t)t * nn + (e - n - nn)] = acc;
  if (rec && t >= rlo && t < rlo + rlen) rec[(size_t)(t - rlo) * per + e] = acc;
}

// gather (mu_k, Sigma_k) for ALL sets: blockIdx.y = set.  With dmu != null the trial mean mu + step dmu is formed on
// the fly (same arithmetic as trial_kernel) and the extra slice blockIdx.y == nsets writes it to mu_out [nmu], so the
// separate mu part of trial_kernel disappears from the iteration.
__global__ __launch_bounds__(256) void gather_all_kernel(SetList L, int n, const double* __restrict__ mu,
                                                         const double* __restrict__ SigD,
                                                         const double* __restrict__ SigU,
                                                         const double* __restrict__ dmu, double step,
                                                         double* __restrict__ mu_out, int64_t nmu) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.y == L.nsets) {
    if (gid < nmu) mu_out[gid] = mu[gid] + step * dmu[gid];
    return;
  }
  const SetDesc& a = L.s[blockIdx.y];
  const int d = a.d, per = d + d * d, nn = n * n;
  if (gid >= (int64_t)a.K * per) return;
  const int k = (int)(gid / per), e = (int)(gid % per), s = a.start[k];
  if (e < d) {
    const size_t j = (size_t)s * n + e;
    a.mu_k[(size_t)k * d + e] = dmu ? mu[j] + step * dmu[j] : mu[j];
    return;
  }
  const int r = (e - d) / d, c = (e - d) % d;
  double v;
  if (r < n && c < n) v = SigD[(size_t)s * nn + r * n + c];
  else if (r >= n && c >= n) v = SigD[(size_t)(s + 1) * nn + (r - n) * n + (c - n)];
  else if (r < n) v = SigU[(size_t)s * nn + r * n + (c - n)];
  else v = SigU[(size_t)s * nn + c * n + (r - n)];
  a.Sigma_k[(size_t)k * d * d + r * d + c] = v;
}

// ordered sum of the factor costs of ALL sets -> acc[0]; one block, fixed tree per set
__global__ __launch_bounds__(256) void cost_sum_all_kernel(SetList L, double* acc, double* acc2 = nullptr) {
  __shared__ double sh[256];
  double total = 0.0;
  for (int si = 0; si < L.nsets; ++si) {
    double s = 0.0;
    for (int k = threadIdx.x; k < L.s[si].K; k += 256) s += L.s[si].cost[k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
      __syncthreads();
    }
    total += sh[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    acc[0] = total;
    if (acc2) acc2[0] = total;            // sharded: the cost word of this rank's exchange records
  }
}

// trial point in one launch: mu_t = mu + step dmu ; Lam_t = Lam + step (V - Lam)   (NGD: dprecision = Vddmu - Lam)
// direct = 1 (proximal update): V already is dprecision, Lam_t = Lam + step V
__global__ __launch_bounds__(256) void trial_kernel(int64_t nmu, int64_t nlam, double step, const double* __restrict__ mu,
                                                    const double* __restrict__ dmu, const double* __restrict__ lam,
                                                    const double* __restrict__ V, double* __restrict__ mu_t,
                                                    double* __restrict__ lam_t, int direct = 0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nmu) mu_t[i] = mu[i] + step * dmu[i];
  else if (i < nmu + nlam) {
    const int64_t j = i - nmu;
    const double l = lam[j];
    lam_t[j] = direct ? l + step * V[j] : l + step * (V[j] - l);
  }
}

// ---- sharded factors: exchange 0 as an all-gather of state records ----
// pack: the rank's state range [lo, lo + len) of [g | D | U] -> records [maxlen][n + 2 n^2] (zero padded).  cost != null:
// one more record whose first word is the rank's partial cost sum, so that the fused trial needs ONE all-gather per
// iteration instead of two (the gradient records and the cost used to travel separately).
__global__ __launch_bounds__(256) void dist_pack_kernel(int T, int n, int lo, int len, int maxlen, const double* __restrict__ g,
                                                        const double* __restrict__ D, const double* __restrict__ U,
                                                        double* __restrict__ rec, const double* __restrict__ cost) {
  const int nn = n * n, per = n + 2 * nn;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)(maxlen + (cost ? 1 : 0)) * per) return;
  const int j = (int)(gid / per), e = (int)(gid % per), t = lo + j;
  double v = 0.0;
  if (j == maxlen) {
    if (e == 0) v = cost[0];
  } else if (j < len && t < T) {
    if (e < n) v = g[(size_t)t * n + e];
    else if (e < n + nn) v = D[(size_t)t * nn + (e - n)];
    else if (t < T - 1) v = U[(size_t)t * nn + (e - n - nn)];
  }
  rec[gid] = v;
}
// fold: every state sums the records of the ranks whose range holds it, in rank order -> full [g | D | U]; stride = records
// per rank in `rec` (maxlen, or maxlen + 1 with the cost record, whose ordered sum goes to cost_out[0])
__global__ __launch_bounds__(256) void dist_fold_kernel(int T, int n, int world, int maxlen, int stride, const int32_t* __restrict__ range,
                                                        const double* __restrict__ rec, double* __restrict__ g,
                                                        double* __restrict__ D, double* __restrict__ U, double* cost_out,
                                                        const double* half_logdet, double* host_out, double seq) {
  const int nn = n * n, per = n + 2 * nn;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid == (int64_t)T * per && cost_out) {
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += rec[((size_t)r * stride + maxlen) * per];
    cost_out[0] = s;
    if (host_out) publish_to_host(host_out, s + half_logdet[0], seq);   // what publish_kernel would do in its own launch
    return;
  }
  if (gid >= (int64_t)T * per) return;
  const int t = (int)(gid / per), e = (int)(gid % per);
  double acc = 0.0;
  for (int r = 0; r < world; ++r) {
    const int lo = range[2 * r], hi = range[2 * r + 1];
    if (t >= lo && t <= hi) acc += rec[((size_t)r * stride + (t - lo)) * per + e];
  }
  if (e < n) g[(size_t)t * n + e] = acc;
  else if (e < n + nn) D[(size_t)t * nn + (e - n)] = acc;
  else if (t < T - 1) U[(size_t)t * nn + (e - n - nn)] = acc;
}
// exchange 1: ordered sum of the ranks' partial cost sums -> acc[0]
__global__ void dist_cost_fold_kernel(int world, const double* __restrict__ parts, double* acc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += parts[r];
    acc[0] = s;
  }
}

// publish the (all-reduced) cost sum and the log-det into host-mapped memory: out = {cost_sum, hld}
__global__ void publish_kernel(const double* cost_sum, const double* half_logdet, double* host_out, double seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    publish_to_host(host_out, cost_sum[0] + half_logdet[0], seq);   // cost_value = sum of factor costs + 1/2 log det
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
