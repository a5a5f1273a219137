// Joint-level (block-tridiagonal) device kernels: scatter/assemble, axpy, chain factorisation
// (log-det + tridiagonal blocks of the inverse), chain solve, marginal gather, cost sums.
//
// Replaces, per SURVEY.md section 8(a):
//   a10/a12  local2joint_*_insertion + sparse "+=" (ngd/NGDFactorizedBaseGH.h:91-106, ngd/NGD-GH-impl.h:39-55)
//   a12      ConjugateGradient solve (ngd/NGD-GH-impl.h:59-60)            -> bcr_forward + bcr_back_solve
//   a14      SimplicialLDLT log-det (gvibase/GVI-GH-impl.h:192-196)       -> bcr_forward + bcr_logdet
//   a16/a17  inv_sparse / inverse_GBP (helpers/EigenWrapper.h:282-381,
//            gvibase/GVI-GH-GBP-impl.h:246-342)                           -> bcr_forward + bcr_back_marginals
//   a11      extract_*_from_joint (gvibase/GVIFactorizedBase.h:104-122)   -> gather_kernel
//
// Chain algorithm: BLOCK CYCLIC REDUCTION instead of the reference's strictly sequential sweeps
// (T dependent steps of tiny n x n blocks are latency-bound on a GPU: 5.8 ms per sweep at T = 1025,
// profiles/r01_a_*).  Level l keeps the nodes that are multiples of s = 2^l and eliminates the odd
// ones; all eliminations of a level are independent (one wave each), ceil(log2 T) levels.
//
//   eliminate e (neighbours a = e - s, b = e + s, couplings Ua = A[a,e], Ub = A[e,b]):
//     Gauss-Jordan [D_e | I | Ua^T | Ub | y_e] -> [I | E | GA | GB | v]   (E = D_e^-1, GA = E Ua^T, GB = E Ub)
//     CL[e] = Ua GA   (pending  -=  on D_a)        CR[e] = Ub^T GB  (pending -= on D_b)
//     NU[e] = -Ua GB  (new coupling A[a,b])        yL[e] = Ua v,  yR[e] = Ub^T v  (pending -= on y_a, y_b)
//   Pending updates of level l-1 are applied at level l, in fixed order: by the eliminating wave for
//   the nodes eliminated at level l, and by "update units" of the same launch for the surviving
//   nodes (Deff[x] = base(x) - CR[x - s/2] - CL[x + s/2]); no two waves ever write one block, and a
//   node's loads are independent of its depth in the tree.
//   back-substitution (solve):   x_e = v - GA x_a - GB x_b
//   selected inverse (Takahashi recursion on the elimination tree):
//     Sig[e,a] = -(GA Sig_aa + GB Sig_ba),  Sig[e,b] = -(GA Sig_ab + GB Sig_bb),
//     Sig_ee = E - Sig[e,a] GA^T - Sig[e,b] GB^T ;  level 0 yields exactly the tridiagonal blocks.
//   log-det = sum over nodes of the log-pivots of its Gauss-Jordan (no pivoting): every pivot is
//   positive iff the matrix is positive definite (any symmetric elimination order), so the
//   reference's "NaN when not PD" rule is preserved; the value is order-independent.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace gvi {

constexpr int BT_MAX_N = 16;                                   // block size limit (LDS / register budget)
// per-lane element counts of an n x (4n+1) tile / an n x n block for n <= NMAX (kernels are
// instantiated for NMAX in {6, 8, 12, 16} so the unrolled per-lane loops stay short)
__host__ __device__ constexpr int bt_epl(int nmax) { return (nmax * (4 * nmax + 1) + 63) / 64; }
__host__ __device__ constexpr int bt_epb(int nmax) { return (nmax * nmax + 63) / 64; }
__host__ __device__ inline int bcr_unit_lds_doubles(int n) { return n * (4 * n + 1) + 2 * n * n + n; }
constexpr int BCR_TAIL_WAVES_MAX = 16;

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

// ---- Gauss-Jordan on an n x nc tile in LDS by one wave ----
// Each lane owns the elements e = lane + 64 q; their (row, col) are computed once (er < 0: none).
// Reads of a pivot step complete into registers before anything is written.  PIVOT: partial (row)
// pivoting folded into the reads as a row permutation of the old tile (for the possibly indefinite
// Vddmu solve).  The pivots are parked in pv[] (LDS) so their logs are taken once, in parallel.
template <bool PIVOT, int EPL>
__device__ inline void gauss_jordan(double* Ts, int n, int nc, const int (&er)[EPL], const int (&ec)[EPL],
                                    double* pv, int lane, int& bad) {
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
    if (lane == 0) pv[p] = piv;
    const double ipiv = 1.0 / piv;
    double nv[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      if (er[q] >= 0) {
        const int r = er[q], c = ec[q];
        const int rr = (r == p) ? rs : ((r == rs) ? p : r);     // row swap p <-> rs
        const double prc = Ts[rs * nc + c] * ipiv;              // scaled pivot row
        nv[q] = (r == p) ? prc : Ts[rr * nc + c] - Ts[rr * nc + p] * prc;
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < EPL; ++q)
      if (er[q] >= 0) Ts[er[q] * nc + ec[q]] = nv[q];
    wave_lds_sync();
  }
}

// ---- workspace of one cyclic reduction (all [T][n][n] unless noted) ----
struct BcrWs {
  double *E, *GA, *GB, *CL, *CR, *NU, *SL, *SR, *Deff;
  double *v, *yL, *yR, *yeff;   // [T][n]
  double* logp;            // [T]
  int* bad;                // [T]
};

struct BcrArgs {
  int T, n;
  int level;               // per-level kernel: eliminate nodes (2u+1) 2^level
  int tail_from;           // tail kernel: levels tail_from .. nlevels-1, then the root, in one workgroup
  int nlevels;
  const double* D;         // [T][n][n]
  const double* U;         // [T-1][n][n]
  const double* rhs;       // [T][n] or null
  double rhs_scale;
  int need_E;
  BcrWs w;
};

// number of nodes eliminated at level l
__host__ __device__ inline int bcr_count(int T, int l) { return (int)((((int64_t)T + (1 << l) - 1) >> l) / 2); }

// Forward elimination of ONE node by one wave.  L = its level (root: L = nlevels, e = 0).
template <bool PIVOT, int NMAX>
__device__ inline void bcr_eliminate(const BcrArgs& a, int L, int e, bool root, int lane, double* sm) {
  constexpr int EPL = bt_epl(NMAX), EPB = bt_epb(NMAX);
  const int n = a.n, nn = n * n, T = a.T;
  const int s = root ? 0 : (1 << L);
  const bool has_a = !root;
  const int b = e + s;
  const bool has_b = !root && b < T;
  const bool rhs = a.rhs != nullptr;
  const int cE = n, cA = a.need_E ? 2 * n : n, cB = cA + n, cY = cB + n;
  const int nc = cY + (rhs ? 1 : 0);
  double* Ts = sm;                 // [n][nc]
  double* Ua = Ts + n * (4 * n + 1);   // [n][n]  A[a,e]
  double* Ub = Ua + nn;            // [n][n]  A[e,b]
  double* pv = Ub + nn;            // [n] pivots
  int br[EPB], bc[EPB];
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int el = lane + q * 64;
    br[q] = el < nn ? el / n : -1;
    bc[q] = el < nn ? el % n : 0;
  }
  int er[EPL], ec[EPL];
#pragma unroll
  for (int q = 0; q < EPL; ++q) {
    const int el = lane + q * 64;
    er[q] = el < n * nc ? el / nc : -1;
    ec[q] = el < n * nc ? el % nc : 0;
  }
  // ---- gather the effective diagonal block / rhs (base + the previous level's pending updates) ----
  const int h = L > 0 ? (1 << (L - 1)) : 0;
  const bool cl = L > 0 && e - h >= 0, cr = L > 0 && e + h < T;
  const double* baseD = (L <= 1 ? a.D : a.w.Deff) + (size_t)e * nn;
  double de[EPB];
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    const int el = lane + q * 64;
    double v = 0.0;
    if (br[q] >= 0) {
      v = baseD[el];
      if (cl) v -= a.w.CR[(size_t)(e - h) * nn + el];
      if (cr) v -= a.w.CL[(size_t)(e + h) * nn + el];
    }
    de[q] = v;
  }
  double ye = 0.0;
  if (rhs && lane < n) {
    ye = L <= 1 ? a.rhs_scale * a.rhs[(size_t)e * n + lane] : a.w.yeff[(size_t)e * n + lane];
    if (cl) ye -= a.w.yR[(size_t)(e - h) * n + lane];
    if (cr) ye -= a.w.yL[(size_t)(e + h) * n + lane];
  }
  const double* pUa = !has_a ? nullptr : (L == 0 ? a.U + (size_t)(e - s) * nn : a.w.NU + (size_t)(e - s / 2) * nn);
  const double* pUb = !has_b ? nullptr : (L == 0 ? a.U + (size_t)e * nn : a.w.NU + (size_t)(e + s / 2) * nn);
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    if (br[q] >= 0) {
      const int el = lane + q * 64, r = br[q], c = bc[q];
      const double ua = has_a ? pUa[el] : 0.0, ub = has_b ? pUb[el] : 0.0;
      Ua[el] = ua;
      Ub[el] = ub;
      Ts[r * nc + c] = de[q];
      if (a.need_E) Ts[r * nc + cE + c] = r == c ? 1.0 : 0.0;
      Ts[c * nc + cA + r] = ua;            // Ua^T
      Ts[r * nc + cB + c] = ub;
    }
  }
  if (rhs && lane < n) Ts[lane * nc + cY] = ye;
  wave_lds_sync();
  int bad = 0;
  gauss_jordan<PIVOT, EPL>(Ts, n, nc, er, ec, pv, lane, bad);
  double lg = lane < n ? log(pv[lane]) : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
  if (lane == 0) { a.w.logp[e] = lg; a.w.bad[e] = bad; }
  // ---- store E, GA, GB, v and the pending updates ----
#pragma unroll
  for (int q = 0; q < EPB; ++q) {
    if (br[q] >= 0) {
      const int el = lane + q * 64, r = br[q], c = bc[q];
      if (a.need_E) a.w.E[(size_t)e * nn + el] = Ts[r * nc + cE + c];
      if (has_a) {
        a.w.GA[(size_t)e * nn + el] = Ts[r * nc + cA + c];
        double cl = 0.0;
        for (int k = 0; k < n; ++k) cl += Ua[r * n + k] * Ts[k * nc + cA + c];
        a.w.CL[(size_t)e * nn + el] = cl;
      }
      if (has_b) {
        a.w.GB[(size_t)e * nn + el] = Ts[r * nc + cB + c];
        double cr = 0.0, nu = 0.0;
        for (int k = 0; k < n; ++k) {
          cr += Ub[k * n + r] * Ts[k * nc + cB + c];
          nu += Ua[r * n + k] * Ts[k * nc + cB + c];
        }
        a.w.CR[(size_t)e * nn + el] = cr;
        a.w.NU[(size_t)e * nn + el] = -nu;
      }
    }
  }
  if (rhs && lane < n) {
    a.w.v[(size_t)e * n + lane] = Ts[lane * nc + cY];
    if (has_a) {
      double yl = 0.0;
      for (int k = 0; k < n; ++k) yl += Ua[lane * n + k] * Ts[k * nc + cY];
      a.w.yL[(size_t)e * n + lane] = yl;
    }
    if (has_b) {
      double yr = 0.0;
      for (int k = 0; k < n; ++k) yr += Ub[k * n + lane] * Ts[k * nc + cY];
      a.w.yR[(size_t)e * n + lane] = yr;
    }
  }
  wave_lds_sync();
}

// ---- register-resident elimination for compile-time block size N (4N+1 <= 64) ----
// Lane c holds column c of the augmented tile [D_e | I | Ua^T | Ub | y] in N registers.  A pivot step
// broadcasts the pivot column with v_readlane (wave-uniform scalars), so Gauss-Jordan needs no LDS
// round trips and no waits: ~2N readlanes + N FMAs per pivot.  Ua / Ub are parked in LDS only for the
// three small products (CL, CR, NU) at the end.
__device__ __forceinline__ double readlane_f64(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

template <bool PIVOT, int N>
__device__ inline void bcr_eliminate_reg(const BcrArgs& a, int L, int e, bool root, int lane, double* sm) {
  constexpr int nn = N * N;
  const int T = a.T;
  const int s = root ? 0 : (1 << L);
  const bool has_a = !root;
  const int b = e + s;
  const bool has_b = !root && b < T;
  const bool rhs = a.rhs != nullptr;
  const int cE = N, cA = a.need_E ? 2 * N : N, cB = cA + N, cY = cB + N;
  double* Ua = sm;                 // [N][N]  A[a,e]
  double* Ub = Ua + nn;            // [N][N]  A[e,b]
  const int h = L > 0 ? (1 << (L - 1)) : 0;
  const bool cl = L > 0 && e - h >= 0, cr = L > 0 && e + h < T;
  const double* baseD = (L <= 1 ? a.D : a.w.Deff) + (size_t)e * nn;
  const double* pUa = !has_a ? nullptr : (L == 0 ? a.U + (size_t)(e - s) * nn : a.w.NU + (size_t)(e - s / 2) * nn);
  const double* pUb = !has_b ? nullptr : (L == 0 ? a.U + (size_t)e * nn : a.w.NU + (size_t)(e + s / 2) * nn);
  // ---- load this lane's column ----
  double col[N];
#pragma unroll
  for (int r = 0; r < N; ++r) col[r] = 0.0;
  if (lane < N) {                                            // D_e (symmetric: read row `lane` = column `lane`)
#pragma unroll
    for (int r = 0; r < N; ++r) {
      double v = baseD[lane * N + r];
      if (cl) v -= a.w.CR[(size_t)(e - h) * nn + lane * N + r];
      if (cr) v -= a.w.CL[(size_t)(e + h) * nn + lane * N + r];
      col[r] = v;
    }
  } else if (a.need_E && lane < 2 * N) {
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = (r == lane - cE) ? 1.0 : 0.0;
  } else if (lane >= cA && lane < cA + N) {                  // column j of Ua^T = row j of Ua
    if (has_a) {
      const int j = lane - cA;
#pragma unroll
      for (int r = 0; r < N; ++r) { col[r] = pUa[j * N + r]; Ua[j * N + r] = col[r]; }
    }
  } else if (lane >= cB && lane < cB + N) {                  // column j of Ub
    if (has_b) {
      const int j = lane - cB;
#pragma unroll
      for (int r = 0; r < N; ++r) { col[r] = pUb[r * N + j]; Ub[r * N + j] = col[r]; }
    }
  } else if (rhs && lane == cY) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
      double y = L <= 1 ? a.rhs_scale * a.rhs[(size_t)e * N + r] : a.w.yeff[(size_t)e * N + r];
      if (cl) y -= a.w.yR[(size_t)(e - h) * N + r];
      if (cr) y -= a.w.yL[(size_t)(e + h) * N + r];
      col[r] = y;
    }
  }
  // ---- Gauss-Jordan, pivot column broadcast by readlane ----
  double pivs[N];
  int bad = 0;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    double ap[N];
#pragma unroll
    for (int r = 0; r < N; ++r) ap[r] = readlane_f64(col[r], p);
    if (PIVOT) {
      int rs = p;
      double best = fabs(ap[p]);
#pragma unroll
      for (int r = p + 1; r < N; ++r)
        if (fabs(ap[r]) > best) { best = fabs(ap[r]); rs = r; }
#pragma unroll
      for (int r = p + 1; r < N; ++r) {                      // swap rows p <-> rs (rs is wave-uniform)
        if (r == rs) {
          const double t = col[p]; col[p] = col[r]; col[r] = t;
          const double u = ap[p]; ap[p] = ap[r]; ap[r] = u;
        }
      }
    }
    const double piv = ap[p];
    if (!(piv > 0.0)) bad = 1;
    pivs[p] = piv;
    const double f = col[p] * (1.0 / piv);
#pragma unroll
    for (int r = 0; r < N; ++r)
      if (r != p) col[r] = fma(-ap[r], f, col[r]);
    col[p] = f;
  }
  double lg = 0.0;
#pragma unroll
  for (int p = 0; p < N; ++p) lg = (lane == p) ? pivs[p] : lg;
  lg = lane < N ? log(lg) : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
  if (lane == 0) { a.w.logp[e] = lg; a.w.bad[e] = bad; }
  wave_lds_sync();                                           // Ua / Ub visible to every lane
  // ---- store E, GA, GB, v and the pending updates (lane = output column) ----
  if (a.need_E && lane >= cE && lane < cE + N) {
    const int c = lane - cE;
#pragma unroll
    for (int r = 0; r < N; ++r) a.w.E[(size_t)e * nn + r * N + c] = col[r];
  }
  if (has_a && lane >= cA && lane < cA + N) {                // GA[:,c] and CL[:,c] = Ua GA[:,c]
    const int c = lane - cA;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      a.w.GA[(size_t)e * nn + r * N + c] = col[r];
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) v = fma(Ua[r * N + k], col[k], v);
      a.w.CL[(size_t)e * nn + r * N + c] = v;
    }
  }
  if (has_b && lane >= cB && lane < cB + N) {                // GB[:,c], CR[:,c] = Ub^T GB[:,c], NU[:,c] = -Ua GB[:,c]
    const int c = lane - cB;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      a.w.GB[(size_t)e * nn + r * N + c] = col[r];
      double v = 0.0, u = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) { v = fma(Ub[k * N + r], col[k], v); u = fma(Ua[r * N + k], col[k], u); }
      a.w.CR[(size_t)e * nn + r * N + c] = v;
      a.w.NU[(size_t)e * nn + r * N + c] = -u;
    }
  }
  if (rhs && lane == cY) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
      a.w.v[(size_t)e * N + r] = col[r];
      if (has_a) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) v = fma(Ua[r * N + k], col[k], v);
        a.w.yL[(size_t)e * N + r] = v;
      }
      if (has_b) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) v = fma(Ub[k * N + r], col[k], v);
        a.w.yR[(size_t)e * N + r] = v;
      }
    }
  }
  wave_lds_sync();
}

// NMAX >= 100 selects the register kernel with N = NMAX - 100 (exact block size)
template <bool PIVOT, int NMAX>
__device__ inline void bcr_eliminate_any(const BcrArgs& a, int L, int e, bool root, int lane, double* sm) {
  if constexpr (NMAX >= 100) bcr_eliminate_reg<PIVOT, NMAX - 100>(a, L, e, root, lane, sm);
  else bcr_eliminate<PIVOT, NMAX>(a, L, e, root, lane, sm);
}

// Surviving node x = 2 j s of level L >= 1: fold the pending updates of level L-1 into Deff / yeff.
__device__ inline void bcr_update_survivor(const BcrArgs& a, int L, int x, int lane) {
  const int n = a.n, nn = n * n, T = a.T, h = 1 << (L - 1);
  const bool cl = x - h >= 0, cr = x + h < T;
  const double* baseD = (L <= 1 ? a.D : a.w.Deff) + (size_t)x * nn;
  for (int el = lane; el < nn; el += 64) {
    double v = baseD[el];
    if (cl) v -= a.w.CR[(size_t)(x - h) * nn + el];
    if (cr) v -= a.w.CL[(size_t)(x + h) * nn + el];
    a.w.Deff[(size_t)x * nn + el] = v;
  }
  if (a.rhs != nullptr && lane < n) {
    double y = L <= 1 ? a.rhs_scale * a.rhs[(size_t)x * n + lane] : a.w.yeff[(size_t)x * n + lane];
    if (cl) y -= a.w.yR[(size_t)(x - h) * n + lane];
    if (cr) y -= a.w.yL[(size_t)(x + h) * n + lane];
    a.w.yeff[(size_t)x * n + lane] = y;
  }
}

// number of surviving (even-index) nodes of level l
__host__ __device__ inline int bcr_survivors(int T, int l) {
  const int cnt = (int)(((int64_t)T + (1 << l) - 1) >> l);
  return (cnt + 1) / 2;
}

// one level: blocks [0, elim) eliminate the odd nodes, blocks [elim, elim + surv) update the even ones
template <bool PIVOT, int NMAX>
__global__ __launch_bounds__(64) void bcr_forward_kernel(BcrArgs a) {
  extern __shared__ double sm[];
  const int elim = bcr_count(a.T, a.level);
  if ((int)blockIdx.x < elim) {
    bcr_eliminate_any<PIVOT, NMAX>(a, a.level, (2 * (int)blockIdx.x + 1) << a.level, false, threadIdx.x, sm);
  } else {
    bcr_update_survivor(a, a.level, (2 * ((int)blockIdx.x - elim)) << a.level, threadIdx.x);
  }
}

// the top of the tree in ONE workgroup: levels tail_from .. nlevels-1 (each with <= blockDim/64
// eliminated nodes, one wave per node; the surviving nodes are updated by the remaining waves) then
// the root, separated by workgroup barriers (the waves share the CU's L1, so a workgroup-scope barrier
// makes the previous phase's global stores visible).
template <bool PIVOT, int NMAX>
__global__ __launch_bounds__(1024) void bcr_forward_tail_kernel(BcrArgs a) {
  extern __shared__ double sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  double* my = sm + (size_t)wave * bcr_unit_lds_doubles(a.n);
  for (int l = a.tail_from; l < a.nlevels; ++l) {
    const int elim = bcr_count(a.T, l);
    if (wave < elim) bcr_eliminate_any<PIVOT, NMAX>(a, l, (2 * wave + 1) << l, false, lane, my);
    if (l > 0) {
      const int surv = bcr_survivors(a.T, l);
      for (int j = wave; j < surv; j += nwaves) bcr_update_survivor(a, l, (2 * j) << l, lane);
    }
    __syncthreads();
  }
  if (wave == 0) bcr_eliminate_any<PIVOT, NMAX>(a, a.nlevels, 0, true, lane, my);
}

// half_logdet = 1/2 sum_t logp[t]  (fixed-order tree), NaN if any node saw a non-positive pivot
__global__ __launch_bounds__(256) void bcr_logdet_kernel(int T, const double* __restrict__ logp,
                                                         const int* __restrict__ bad, double* out) {
  __shared__ double sh[256];
  __shared__ int sb[256];
  double s = 0.0;
  int bflag = 0;
  for (int t = threadIdx.x; t < T; t += 256) { s += logp[t]; bflag |= bad[t]; }
  sh[threadIdx.x] = s; sb[threadIdx.x] = bflag;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) { sh[threadIdx.x] += sh[threadIdx.x + w]; sb[threadIdx.x] |= sb[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sb[0] ? __builtin_nan("") : 0.5 * sh[0];
}

// ---- back-substitution ----
__device__ inline void bcr_solve_row(int T, int n, int level, const BcrWs& w, double* __restrict__ x, int u, int r) {
  const int s = 1 << level, nn = n * n;
  const int e = (2 * u + 1) * s;
  if (e >= T) return;
  const int a = e - s, b = e + s;
  double xe = w.v[(size_t)e * n + r];
  const double* GA = w.GA + (size_t)e * nn + r * n;
  for (int k = 0; k < n; ++k) xe -= GA[k] * x[(size_t)a * n + k];
  if (b < T) {
    const double* GB = w.GB + (size_t)e * nn + r * n;
    for (int k = 0; k < n; ++k) xe -= GB[k] * x[(size_t)b * n + k];
  }
  x[(size_t)e * n + r] = xe;
}

// one level: thread per (node, row)
__global__ __launch_bounds__(256) void bcr_back_solve_kernel(int T, int n, int level, BcrWs w, double* __restrict__ x) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  bcr_solve_row(T, n, level, w, x, gid / n, gid % n);
}

// root + levels nlevels-1 .. head_to in one workgroup (each with <= blockDim / n nodes)
__global__ __launch_bounds__(1024) void bcr_back_solve_head_kernel(int T, int n, int nlevels, int head_to, BcrWs w,
                                                                   double* __restrict__ x) {
  const int tid = threadIdx.x;
  if (tid < n) x[tid] = w.v[tid];
  __syncthreads();
  for (int l = nlevels - 1; l >= head_to; --l) {
    if (tid < bcr_count(T, l) * n) bcr_solve_row(T, n, l, w, x, tid / n, tid % n);
    __syncthreads();
  }
}

// ---- selected-inverse recursion: one wave per eliminated node ----
__device__ inline void bcr_marginal_node(int T, int n, int level, int e, const BcrWs& w, double* __restrict__ SigD,
                                         double* __restrict__ SigU, int lane, double* sm) {
  const int nn = n * n;
  const int s = 1 << level;
  const int a = e - s, b = e + s;
  const bool has_b = b < T;
  double* GA = sm;            // [n][n]
  double* GB = GA + nn;
  double* Saa = GB + nn;
  double* Sbb = Saa + nn;
  double* Sab = Sbb + nn;     // Sig[a,b]
  double* SLs = Sab + nn;     // Sig[e,a]
  double* SRs = SLs + nn;     // Sig[e,b]
  // Sig[a,b]: a, b are adjacent at level+1; the odd one of the pair was eliminated there
  const bool a_odd = has_b && (((a / (2 * s)) & 1) != 0);
  for (int el = lane; el < nn; el += 64) {
    const int r = el / n, c = el % n;
    GA[el] = w.GA[(size_t)e * nn + el];
    Saa[el] = SigD[(size_t)a * nn + el];
    if (has_b) {
      GB[el] = w.GB[(size_t)e * nn + el];
      Sbb[el] = SigD[(size_t)b * nn + el];
      Sab[el] = a_odd ? w.SR[(size_t)a * nn + el] : w.SL[(size_t)b * nn + c * n + r];
    }
  }
  wave_lds_sync();
  for (int el = lane; el < nn; el += 64) {
    const int r = el / n, c = el % n;
    double sl = 0.0, sr = 0.0;
    for (int k = 0; k < n; ++k) {
      sl += GA[r * n + k] * Saa[k * n + c];
      if (has_b) {
        sl += GB[r * n + k] * Sab[c * n + k];        // Sig_ba[k][c] = Sig_ab[c][k]
        sr += GA[r * n + k] * Sab[k * n + c] + GB[r * n + k] * Sbb[k * n + c];
      }
    }
    SLs[el] = -sl;
    SRs[el] = -sr;
    w.SL[(size_t)e * nn + el] = -sl;
    if (has_b) w.SR[(size_t)e * nn + el] = -sr;
  }
  wave_lds_sync();
  for (int el = lane; el < nn; el += 64) {
    const int r = el / n, c = el % n;
    double see = w.E[(size_t)e * nn + el];
    for (int k = 0; k < n; ++k) {
      see -= SLs[r * n + k] * GA[c * n + k];
      if (has_b) see -= SRs[r * n + k] * GB[c * n + k];
    }
    SigD[(size_t)e * nn + el] = see;
    if (level == 0) {                                 // tridiagonal blocks of the original chain
      SigU[(size_t)a * nn + c * n + r] = SLs[el];     // Sig[a,e] = Sig[e,a]^T
      if (has_b) SigU[(size_t)e * nn + el] = SRs[el];
    }
  }
  wave_lds_sync();
}

__global__ __launch_bounds__(64) void bcr_back_marginals_kernel(int T, int n, int level, BcrWs w,
                                                                double* __restrict__ SigD, double* __restrict__ SigU) {
  extern __shared__ double sm[];
  bcr_marginal_node(T, n, level, (2 * (int)blockIdx.x + 1) << level, w, SigD, SigU, threadIdx.x, sm);
}

// root (Sig_00 = E_0) + levels nlevels-1 .. head_to in one workgroup, one wave per node
__global__ __launch_bounds__(1024) void bcr_back_marginals_head_kernel(int T, int n, int nlevels, int head_to, BcrWs w,
                                                                       double* __restrict__ SigD,
                                                                       double* __restrict__ SigU) {
  extern __shared__ double sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = n * n;
  for (int el = threadIdx.x; el < nn; el += blockDim.x) SigD[el] = w.E[el];
  __syncthreads();
  for (int l = nlevels - 1; l >= head_to; --l) {
    if (wave < bcr_count(T, l)) bcr_marginal_node(T, n, l, (2 * wave + 1) << l, w, SigD, SigU, lane, sm + (size_t)wave * 7 * nn);
    __syncthreads();
  }
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
