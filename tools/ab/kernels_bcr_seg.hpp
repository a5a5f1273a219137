// Segmented block cyclic reduction: the chain solve / log-det / selected inverse in THREE launches.
//
// The per-level BCR of kernels_bt.hpp needs ~2 log2(T) dependent launches and every tiny launch costs
// ~5 us on this part (profiles/r01_c_*): at T = 1025 the two chain operations of an NGD iteration were
// 260 us of mostly launch floor.  Here a workgroup owns a SEGMENT of S = 2^m consecutive alive nodes and
// runs m levels of cyclic reduction on it inside LDS (workgroup barriers only, eliminations in
// registers, see bcr_eliminate_reg); segments never need each other within a pass:
//
//   pass A  (grid = T / S segments)   levels 0 .. m-1 inside every segment; the segment's first node
//                                     survives; pending updates that cross a segment boundary and the
//                                     per-node factors (E, GA, GB, v) go to global memory
//   pass B  (one workgroup)           the <= CAP surviving nodes: fold the pending updates, remaining
//                                     levels, root, log-det; then the BACKWARD recursion for the same
//                                     nodes (solution / selected inverse), still in LDS
//   pass C  (grid = T / S segments)   backward recursion inside every segment
//
// (If more than CAP nodes survive pass A it is repeated at the coarser spacing -- not needed below
// T ~ 2 000 at n = 6.)  The algebra is exactly that of kernels_bt.hpp (same elimination tree, same
// fixed summation orders => deterministic); only where the blocks live and who waits for whom changes.
//
// Instantiated for the block sizes with a register elimination (N in {1,2,3,4,6,8,12}); other n use
// the per-level kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_bt.hpp"

// Round-2 chain kernels, kept as the A/B leg of tools/ubench/chain_bench.hip only (the library uses kernels_chain.hpp).
namespace gvi {

__device__ __forceinline__ double readlane_f64(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// -DGVI_BCR_TIMING: constant-clock stamps (s_memrealtime, 100 MHz) of block 0 of every pass at its phase boundaries, read back with
// gvi_debug_bcr_stamps -- a profiling aid for the latency-bound chain kernels, compiled out of the product build.
#ifdef GVI_BCR_TIMING
__device__ unsigned long long gvi_bcr_stamps[6][64];
#define GVI_STAMP(slot, idx) do { if (bid == 0 && threadIdx.x == 0 && (idx) < 64) gvi_bcr_stamps[slot][idx] = wall_clock64(); } while (0)
// inside seg_eliminate: shader-clock (s_memtime) stamps of wave 0 / block 0 at the second level of the factor's pass A
__device__ unsigned long long gvi_bcr_stamps_elim[16];
__device__ int gvi_bcr_dbg_on;
#define GVI_ESTAMP(idx) do { if (gvi_bcr_dbg_on && threadIdx.x == 0) gvi_bcr_stamps_elim[idx] = clock64(); } while (0)
#else
#define GVI_STAMP(slot, idx) do { } while (0)
#define GVI_ESTAMP(idx) do { } while (0)
#endif

// The chain workspace is ONE allocation (ensure_chain_ws: nine [T][n][n] arrays, four [T][n] arrays, logp[T]); the kernels
// carry its base and derive the arrays.  Fifteen separate pointers per chain operation (two operations in the dual
// kernels) overflowed the scalar register file: the element phase of an elimination reloaded spilled SGPRs 114 times.
struct SegWs {
  double* base;
  int* bad;                // [T]
};

struct SegArgs {
  int T, n;
  int level0;        // first level of this pass; node spacing st = 1 << level0
  int m;             // levels handled in this pass
  int S;             // local slots per workgroup (segment size 2^m, or the number of alive nodes in the top pass)
  int prev0;         // first level of the previous pass (pending updates of levels [prev0, level0) are folded at load)
  int top;           // last pass: node 0 is the root, log-det, then the backward recursion for these nodes
  int need_E;        // selected inverse wanted (else log-det only / solve)
  const double* D;
  const double* U;
  const double* rhs;     // null: factor; else solve
  double rhs_scale;
  SegWs w;
  double* SigD;      // [T][n][n]   (marginals)
  double* SigU;      // [T-1][n][n]
  double* x;         // [T][n]      (solve)
  double* hld;       // [1]
  // optional fused trial precision (first pass only): the chain operated on is D + mix_step (mixV - D), and it is
  // written to mix_out -- replaces the separate trial_kernel launch.  Both are [D[T] | U[T-1]] in ONE buffer (the U part
  // starts at block T), so one pointer each
  const double* mixVD;
  double* mixOutD;
  double mix_step;
  const double* pred;    // predicated launch (device_common.hpp, pred_skip) or null
  double pred_val;
};

__device__ __forceinline__ double* ws_mat(const SegArgs& a, int idx) { return a.w.base + (size_t)idx * a.T * (a.n * a.n); }
__device__ __forceinline__ double* ws_vec(const SegArgs& a, int idx) {
  return a.w.base + (size_t)9 * a.T * (a.n * a.n) + (size_t)idx * a.T * a.n;
}
__device__ __forceinline__ double* ws_E(const SegArgs& a) { return ws_mat(a, 0); }
__device__ __forceinline__ double* ws_GA(const SegArgs& a) { return ws_mat(a, 1); }
__device__ __forceinline__ double* ws_GB(const SegArgs& a) { return ws_mat(a, 2); }
__device__ __forceinline__ double* ws_CL(const SegArgs& a) { return ws_mat(a, 3); }
__device__ __forceinline__ double* ws_CR(const SegArgs& a) { return ws_mat(a, 4); }
__device__ __forceinline__ double* ws_NU(const SegArgs& a) { return ws_mat(a, 5); }
__device__ __forceinline__ double* ws_SL(const SegArgs& a) { return ws_mat(a, 6); }
__device__ __forceinline__ double* ws_SR(const SegArgs& a) { return ws_mat(a, 7); }
__device__ __forceinline__ double* ws_Deff(const SegArgs& a) { return ws_mat(a, 8); }
__device__ __forceinline__ double* ws_v(const SegArgs& a) { return ws_vec(a, 0); }
__device__ __forceinline__ double* ws_yL(const SegArgs& a) { return ws_vec(a, 1); }
__device__ __forceinline__ double* ws_yR(const SegArgs& a) { return ws_vec(a, 2); }
__device__ __forceinline__ double* ws_yeff(const SegArgs& a) { return ws_vec(a, 3); }
__device__ __forceinline__ double* ws_logp(const SegArgs& a) { return ws_vec(a, 4); }
__device__ __forceinline__ int* ws_bad(const SegArgs& a) { return a.w.bad; }

// forward kernel LDS: 5 block arrays (+3 factor arrays in the top pass), rhs vectors, one elimination tile
// per wave, reduction area
__host__ __device__ inline size_t seg_fwd_lds_doubles(int n, int S, bool rhs, bool top, int nwaves) {
  return (size_t)(top ? 8 : 5) * S * n * n + (rhs ? (size_t)5 * (S + 1) * n : 0) + (size_t)nwaves * (4 * n * n + n) + 1600;
}
__host__ __device__ inline size_t seg_bwd_lds_doubles(int n, int S, bool rhs) {
  return rhs ? (size_t)(S + 1) * n + (size_t)S * (2 * n * n + n) + 16
             : (size_t)3 * (S + 1) * n * n + (size_t)3 * S * n * n;
}

// ---- forward elimination of one node from LDS-resident operands (register Gauss-Jordan) ----
// Dl_e: its effective diagonal block; Ua = A[a,e], Ub = A[e,b] (LDS, null when absent); y_e (LDS).
// Results: per-node factors to global (node id x), pending updates to the LDS slot arrays AND global.
template <bool PIVOT, int N>
__device__ inline void seg_eliminate(const SegArgs& a, int x, const double* Dl_e, const double* Ua, const double* Ub,
                                     const double* y_e, double* CLs, double* CRs, double* NUs, double* yLs,
                                     double* yRs, double* Tl, double* Es, double* GAs, double* GBs, double* vs,
                                     int lane) {
  constexpr int nn = N * N, NC = 4 * N + 1;
  const bool has_a = Ua != nullptr, has_b = Ub != nullptr, rhs = a.rhs != nullptr;
  const int cE = N, cA = a.need_E ? 2 * N : N, cB = cA + N, cY = cB + N;
  GVI_ESTAMP(0);
  double col[N];
#pragma unroll
  for (int r = 0; r < N; ++r) col[r] = 0.0;
  if (lane < N) {
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = Dl_e[lane * N + r];        // symmetric: row = column
  } else if (a.need_E && lane < 2 * N) {
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = (r == lane - cE) ? 1.0 : 0.0;
  } else if (lane >= cA && lane < cA + N) {
    if (has_a) {
#pragma unroll
      for (int r = 0; r < N; ++r) col[r] = Ua[(lane - cA) * N + r];   // column j of Ua^T = row j of Ua
    }
  } else if (lane >= cB && lane < cB + N) {
    if (has_b) {
#pragma unroll
      for (int r = 0; r < N; ++r) col[r] = Ub[r * N + (lane - cB)];
    }
  } else if (rhs && lane == cY) {
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = y_e[r];
  }
  GVI_ESTAMP(1);
  double pivs[N];
  int bad = 0;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    if (PIVOT) {
      // threshold partial pivoting, searched inside lane p (the pivot column lives in that lane's registers):
      // rows are swapped only when the natural pivot is more than 8x smaller than the column maximum -- for the
      // near-SPD blocks of an NGD iteration that is rare, so the common path is one wave-uniform branch
      double best = fabs(col[p]);
      int rs = p;
#pragma unroll
      for (int r = p + 1; r < N; ++r) {
        const bool gt = fabs(col[r]) > best;
        best = gt ? fabs(col[r]) : best;
        rs = gt ? r : rs;
      }
      rs = fabs(col[p]) * 8.0 >= best ? p : rs;
      rs = __builtin_amdgcn_readlane(rs, p);
      if (rs != p) {
#pragma unroll
        for (int r = p + 1; r < N; ++r) {
          if (r == rs) { const double t = col[p]; col[p] = col[r]; col[r] = t; }
        }
      }
    }
    double ap[N];
#pragma unroll
    for (int r = 0; r < N; ++r) ap[r] = readlane_f64(col[r], p);
    const double piv = ap[p];
    if (!(piv > 0.0)) bad = 1;
    pivs[p] = piv;
    double ip = __builtin_amdgcn_rcp(piv);               // reciprocal + two Newton steps (full fp64 accuracy)
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    const double f = col[p] * ip;
#pragma unroll
    for (int r = 0; r < N; ++r)
      if (r != p) col[r] = fma(-ap[r], f, col[r]);
    col[p] = f;
  }
  GVI_ESTAMP(2);
  if (a.hld && lane == 0) {
    // log-pivots only feed the log-det (factor calls).  The pivots are wave-uniform: keep their product as
    // (mantissa product, exponent sum) -- two instructions per pivot -- and take ONE log per node in the final
    // reduction instead of an fp64 log + a 6-step shuffle reduction on the critical path of every level.
    double mp = 1.0;
    int es = 0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      mp *= __builtin_amdgcn_frexp_mant(pivs[p]);
      es += __builtin_amdgcn_frexp_exp(pivs[p]);
    }
    ws_logp(a)[x] = mp;                    // in [2^-N, 1) for positive pivots
    ws_bad(a)[x] = es * 2 + bad;           // exponent sum and the non-positive-pivot flag
  }
  GVI_ESTAMP(3);
  // ---- park the reduced tile [I | E | GA | GB | v] in LDS, then finish element-wise on all lanes ----
  if (lane < NC) {
#pragma unroll
    for (int r = 0; r < N; ++r) Tl[r * NC + lane] = col[r];
  }
  wave_lds_sync();
  GVI_ESTAMP(4);
  for (int el = lane; el < nn; el += 64) {
    const int r = el / N, c = el % N;
    if (a.need_E) { const double ev = Tl[r * NC + cE + c]; ws_E(a)[(size_t)x * nn + el] = ev; if (Es) Es[el] = ev; }
    if (has_a) {
      { const double gv = Tl[r * NC + cA + c]; ws_GA(a)[(size_t)x * nn + el] = gv; if (GAs) GAs[el] = gv; }
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) v = fma(Ua[r * N + k], Tl[k * NC + cA + c], v);
      CLs[el] = v;
      ws_CL(a)[(size_t)x * nn + el] = v;
    }
    if (has_b) {
      { const double gv = Tl[r * NC + cB + c]; ws_GB(a)[(size_t)x * nn + el] = gv; if (GBs) GBs[el] = gv; }
      double v = 0.0, u = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const double gb = Tl[k * NC + cB + c];
        v = fma(Ub[k * N + r], gb, v);
        if (has_a) u = fma(Ua[r * N + k], gb, u);
      }
      CRs[el] = v;
      ws_CR(a)[(size_t)x * nn + el] = v;
      if (has_a) { NUs[el] = -u; ws_NU(a)[(size_t)x * nn + el] = -u; }
    }
  }
  GVI_ESTAMP(5);
  if (rhs && lane < N) {
    const int r = lane;
    { const double vv = Tl[r * NC + cY]; ws_v(a)[(size_t)x * N + r] = vv; if (vs) vs[r] = vv; }
    if (has_a) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) v = fma(Ua[r * N + k], Tl[k * NC + cY], v);
      yLs[r] = v;
      ws_yL(a)[(size_t)x * N + r] = v;
    }
    if (has_b) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) v = fma(Ub[k * N + r], Tl[k * NC + cY], v);
      yRs[r] = v;
      ws_yR(a)[(size_t)x * N + r] = v;
    }
  }
  wave_lds_sync();
  GVI_ESTAMP(6);
}

// ---- backward step of one node: selected inverse (marginals) ----
// E / GA / GB: the node's factors (LDS); Saa / Sbb / Sab: LDS blocks of its neighbours' covariance.
// Writes Sig_ee, Sig[e,a], Sig[e,b] to the LDS slots and to global.
template <int N>
__device__ inline void seg_marginal_node(const SegArgs& a, int x, int xa, int level, bool has_b, const double* E,
                                         const double* GA, const double* GB, const double* Saa, const double* Sbb,
                                         const double* Sab, bool sab_transposed, double* See_s, double* SL_s,
                                         double* SR_s, int lane) {
  constexpr int nn = N * N;
  for (int el = lane; el < nn; el += 64) {
    const int r = el / N, c = el % N;
    double sl = 0.0, sr = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      sl = fma(GA[r * N + k], Saa[k * N + c], sl);
      if (has_b) {
        const double sab_kc = sab_transposed ? Sab[c * N + k] : Sab[k * N + c];   // Sig_ab[k][c]
        const double sba_kc = sab_transposed ? Sab[k * N + c] : Sab[c * N + k];   // Sig_ba[k][c] = Sig_ab[c][k]
        sl = fma(GB[r * N + k], sba_kc, sl);
        sr = fma(GA[r * N + k], sab_kc, sr);
        sr = fma(GB[r * N + k], Sbb[k * N + c], sr);
      }
    }
    SL_s[el] = -sl;
    SR_s[el] = -sr;
    ws_SL(a)[(size_t)x * nn + el] = -sl;
    if (has_b) ws_SR(a)[(size_t)x * nn + el] = -sr;
  }
  wave_lds_sync();
  for (int el = lane; el < nn; el += 64) {
    const int r = el / N, c = el % N;
    double see = E[el];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      see = fma(-SL_s[r * N + k], GA[c * N + k], see);
      if (has_b) see = fma(-SR_s[r * N + k], GB[c * N + k], see);
    }
    See_s[el] = see;
    a.SigD[(size_t)x * nn + el] = see;
    if (level == 0) {                                  // tridiagonal blocks of the original chain
      a.SigU[(size_t)xa * nn + c * N + r] = SL_s[el];  // Sig[a,e] = Sig[e,a]^T
      if (has_b) a.SigU[(size_t)x * nn + el] = SR_s[el];
    }
  }
  wave_lds_sync();
}

// Fold the pending updates of levels [prev0, level0) into node x's diagonal block / rhs (global reads only: the
// caller issues the loads of all its elements before the first use, then stores).
__device__ __forceinline__ double seg_fold_D(const SegArgs& a, int x, int el) {
  const int nn = a.n * a.n;
  double v = (a.level0 == 0 ? a.D : ws_Deff(a))[(size_t)x * nn + el];
  double mv = 0.0;
  const bool mix = a.level0 == 0 && a.mixVD;
  if (mix) mv = a.mixVD[(size_t)x * nn + el];
  // at most 5 levels per pass (seg_plan): a fixed-trip, fully unrolled loop keeps the (independent)
  // loads in flight together instead of one global round trip per level
  double cr[5], cl[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int l = a.prev0 + q, h = 1 << l;
    const bool on = l < a.level0;
    cr[q] = (on && x - h >= 0) ? ws_CR(a)[(size_t)(x - h) * nn + el] : 0.0;
    cl[q] = (on && x + h < a.T) ? ws_CL(a)[(size_t)(x + h) * nn + el] : 0.0;
  }
  if (mix) v = v + a.mix_step * (mv - v);                         // same arithmetic as trial_kernel
#pragma unroll
  for (int q = 0; q < 5; ++q) { v -= cr[q]; v -= cl[q]; }
  return v;
}
__device__ inline double seg_fold_y(const SegArgs& a, int x, int r) {
  const int n = a.n;
  double v = a.level0 == 0 ? a.rhs_scale * a.rhs[(size_t)x * n + r] : ws_yeff(a)[(size_t)x * n + r];
  double yr[5], yl[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int l = a.prev0 + q, h = 1 << l;
    const bool on = l < a.level0;
    yr[q] = (on && x - h >= 0) ? ws_yR(a)[(size_t)(x - h) * n + r] : 0.0;
    yl[q] = (on && x + h < a.T) ? ws_yL(a)[(size_t)(x + h) * n + r] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) { v -= yr[q]; v -= yl[q]; }
  return v;
}

// ---- pass A / B forward (+ pass B backward): one workgroup per segment ----
template <bool PIVOT, int N>
__device__ __forceinline__ void bcr_seg_forward_body(const SegArgs& a, const int bid, double* sm) {
  constexpr int nn = N * N;
  const int T = a.T, S = a.S, st = 1 << a.level0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
  const bool rhs = a.rhs != nullptr;
  const int x0 = bid * S * st;
  [[maybe_unused]] const int tslot = (rhs ? 3 : 0) + (a.top ? 1 : 0);
  [[maybe_unused]] int tix = 0;
  GVI_STAMP(tslot, tix++);
  double* Dl = sm;                    // [S][nn] effective diagonal blocks
  double* Cl = Dl + S * nn;           // [S][nn] coupling A[x_j, x_{j+1}] at this pass's spacing
  double* NUl = Cl + S * nn;          // [S][nn] new couplings, indexed by the eliminated node
  double* CLl = NUl + S * nn;         // [S][nn] pending update to the left neighbour
  double* CRl = CLl + S * nn;         // [S][nn] pending update to the right neighbour
  double* El = CRl + S * nn;          // top pass only: [S][nn] x3 factors kept for the backward recursion
  double* GAl = El + (a.top ? S * nn : 0);
  double* GBl = GAl + (a.top ? S * nn : 0);
  double* yl = GBl + (a.top ? S * nn : 0);        // [S+1][N] (rhs only)
  double* yLl = yl + (rhs ? (S + 1) * N : 0);
  double* yRl = yLl + (rhs ? (S + 1) * N : 0);
  double* xl = yRl + (rhs ? (S + 1) * N : 0);    // [S+1][N] solution (top pass backward)
  double* vl = xl + (rhs ? (S + 1) * N : 0);     // [S+1][N] v (top pass)
  double* scratch = vl + (rhs ? (S + 1) * N : 0);   // [waves][4 nn + N]: elimination tiles
  double* red = scratch + nwaves * (4 * nn + N);  // [1024] doubles + [1024] ints: log-det reduction
  // number of local nodes that exist
  int cnt = 0;
  for (int j = 0; j < S; ++j) if (x0 + j * st < T) cnt = j + 1;
  const bool ext_right = x0 + S * st < T;          // the next segment's first node exists
  // ---- load + fold ----
  // Two elements per thread and round (S nn <= 2 blockDim for the shipped plans): every global load of the round is
  // issued before the first result is used, so the phase costs one memory round trip instead of one per element.
  for (int e0 = tid; e0 < cnt * nn; e0 += 2 * blockDim.x) {
    double dv[2], cu[2];
    bool on[2], hasc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      on[u] = e < cnt * nn;
      hasc[u] = false;
      dv[u] = cu[u] = 0.0;
      if (on[u]) {
        const int j = e / nn, el = e % nn, x = x0 + j * st;
        dv[u] = seg_fold_D(a, x, el);
        hasc[u] = x + st < T;
        if (hasc[u]) {
          cu[u] = (a.level0 == 0 ? a.U + (size_t)x * nn : ws_NU(a) + (size_t)(x + st / 2) * nn)[el];
          if (a.level0 == 0 && a.mixVD) cu[u] = cu[u] + a.mix_step * (a.mixVD[(size_t)(T + x) * nn + el] - cu[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      if (on[u]) {
        const int j = e / nn, el = e % nn, x = x0 + j * st;
        Dl[e] = dv[u];
        if (a.level0 == 0 && a.mixVD) a.mixOutD[(size_t)x * nn + el] = dv[u];
        if (j == 0 && !a.top) ws_Deff(a)[(size_t)x * nn + el] = dv[u];      // the survivor's base for the next pass
        if (hasc[u]) {
          Cl[e] = cu[u];
          if (a.level0 == 0 && a.mixVD) a.mixOutD[(size_t)(T + x) * nn + el] = cu[u];
        }
      }
    }
  }
  if (rhs) {
    for (int e = tid; e < cnt * N; e += blockDim.x) {
      const int j = e / N, r = e % N, x = x0 + j * st;
      const double v = seg_fold_y(a, x, r);
      yl[e] = v;
      if (j == 0 && !a.top) ws_yeff(a)[(size_t)x * N + r] = v;
    }
  }
  __syncthreads();
  GVI_STAMP(tslot, tix++);
  // ---- m local levels ----
  for (int lam = 0; lam < a.m; ++lam) {
    const int h2 = 1 << lam;
    // eliminated local nodes: odd multiples of h2 that exist
    int nel = 0;
    for (int j = h2; j < cnt; j += 2 * h2) ++nel;
#ifdef GVI_BCR_TIMING
    if (bid == 0 && tid == 0) gvi_bcr_dbg_on = (tslot == 0 && lam == 1) ? 1 : 0;
#endif
    for (int u = wave; u < nel; u += nwaves) {
      const int j = (2 * u + 1) * h2, x = x0 + j * st;
      const int ja = j - h2, jb = j + h2;
      const bool has_b = (jb < cnt) || (jb == S && ext_right && !a.top);
      const double* Ua = lam == 0 ? Cl + ja * nn : NUl + (j - h2 / 2) * nn;
      const double* Ub = !has_b ? nullptr : (lam == 0 ? Cl + j * nn : NUl + (j + h2 / 2) * nn);
      seg_eliminate<PIVOT, N>(a, x, Dl + j * nn, Ua, Ub, yl + j * N, CLl + j * nn, CRl + j * nn, NUl + j * nn,
                              yLl + j * N, yRl + j * N, scratch + wave * (4 * nn + N),
                              a.top ? El + j * nn : nullptr, a.top ? GAl + j * nn : nullptr,
                              a.top ? GBl + j * nn : nullptr, (a.top && rhs) ? vl + j * N : nullptr, lane);
    }
    lds_barrier();
    GVI_STAMP(tslot, tix++);
    // eager update of the local survivors of this level (node 0 only in the top pass: elsewhere its
    // left-hand updates live in another segment, so it folds both sides from global in the next pass)
    const int first = a.top ? 0 : 2 * h2;
    for (int j = first + wave * 2 * h2; j < cnt; j += nwaves * 2 * h2) {
      const bool lft = j - h2 >= 0, rgt = j + h2 < cnt;
      for (int el = lane; el < nn; el += 64) {
        double v = Dl[j * nn + el];
        if (lft) v -= CRl[(j - h2) * nn + el];
        if (rgt) v -= CLl[(j + h2) * nn + el];
        Dl[j * nn + el] = v;
      }
      if (rhs && lane < N) {
        double v = yl[j * N + lane];
        if (lft) v -= yRl[(j - h2) * N + lane];
        if (rgt) v -= yLl[(j + h2) * N + lane];
        yl[j * N + lane] = v;
      }
    }
    lds_barrier();
    GVI_STAMP(tslot, tix++);
  }
  if (!a.top) return;
  // ---- root (node 0), log-det ----
  if (wave == 0)
    seg_eliminate<PIVOT, N>(a, 0, Dl, nullptr, nullptr, yl, CLl, CRl, NUl, yLl, yRl, scratch, El, GAl, GBl,
                            rhs ? vl : nullptr, lane);
  lds_barrier();                                     // the factors of these nodes live in LDS (El, GAl, GBl, vl)
  GVI_STAMP(tslot, tix++);
  if (a.hld) {                                       // 1/2 sum of the log-pivots of ALL nodes (every pass)
    int* redb = (int*)(red + 64);
    double s = 0.0;
    int bflag = 0;
    for (int t = tid; t < T; t += blockDim.x) {
      const int eb = ws_bad(a)[t];
      s += log(ws_logp(a)[t]) + (double)(eb >> 1) * 0.6931471805599453094;
      bflag |= eb & 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); bflag |= __shfl_xor(bflag, o); }
    if (lane == 0) { red[wave] = s; redb[wave] = bflag; }
    lds_barrier();
    if (tid == 0) {
      double tot = 0.0;
      int bb = 0;
      for (int wv = 0; wv < nwaves; ++wv) { tot += red[wv]; bb |= redb[wv]; }   // fixed order
      a.hld[0] = bb ? __builtin_nan("") : 0.5 * tot;
    }
    lds_barrier();
  }
  GVI_STAMP(tslot, tix++);
  // ---- backward recursion for the nodes of this pass ----
  if (rhs) {                                          // solve: x_e = v - GA x_a - GB x_b
    if (tid < N) { const double v = vl[tid]; xl[tid] = v; a.x[tid] = v; }
    lds_barrier();
    for (int lam = a.m - 1; lam >= 0; --lam) {
      const int h2 = 1 << lam;
      int nel = 0;
      for (int j = h2; j < cnt; j += 2 * h2) ++nel;
      for (int e = tid; e < nel * N; e += blockDim.x) {
        const int u = e / N, r = e % N, j = (2 * u + 1) * h2, x = x0 + j * st;
        const int ja = j - h2, jb = j + h2;
        double xe = vl[j * N + r];
#pragma unroll
        for (int k = 0; k < N; ++k) xe = fma(-GAl[j * nn + r * N + k], xl[ja * N + k], xe);
        if (jb < cnt) {
#pragma unroll
          for (int k = 0; k < N; ++k) xe = fma(-GBl[j * nn + r * N + k], xl[jb * N + k], xe);
        }
        xl[j * N + r] = xe;
        a.x[(size_t)x * N + r] = xe;
      }
      lds_barrier();
      GVI_STAMP(tslot, tix++);
    }
  } else if (a.need_E) {                              // selected inverse
    double* Sgl = Dl;                                 // [S][nn] Sig_jj          (forward arrays are dead)
    double* SLl = Cl;                                 // [S][nn] Sig[j, left neighbour at its level]
    double* SRl = CLl;                                // [S][nn] Sig[j, right neighbour]
    for (int el = tid; el < nn; el += blockDim.x) { const double v = El[el]; Sgl[el] = v; a.SigD[el] = v; }
    lds_barrier();
    for (int lam = a.m - 1; lam >= 0; --lam) {
      const int h2 = 1 << lam;
      int nel = 0;
      for (int j = h2; j < cnt; j += 2 * h2) ++nel;
      for (int u = wave; u < nel; u += nwaves) {
        const int j = (2 * u + 1) * h2, x = x0 + j * st;
        const int ja = j - h2, jb = j + h2;
        const bool has_b = jb < cnt;
        const bool a_odd = has_b && (((ja / (2 * h2)) & 1) != 0);   // which of a, b was eliminated at the next level
        const double* Sab = a_odd ? SRl + ja * nn : SLl + (has_b ? jb : 0) * nn;
        seg_marginal_node<N>(a, x, x0 + ja * st, a.level0 + lam, has_b, El + j * nn, GAl + j * nn, GBl + j * nn,
                             Sgl + ja * nn, Sgl + (has_b ? jb : 0) * nn, Sab, !a_odd, Sgl + j * nn, SLl + j * nn,
                             SRl + j * nn, lane);
      }
      lds_barrier();
      GVI_STAMP(tslot, tix++);
    }
  }
}

template <bool PIVOT, int N>
__global__ __launch_bounds__(1024) void bcr_seg_forward_kernel(SegArgs a) {
  extern __shared__ double sm[];
  if (pred_skip(a.pred, a.pred_val)) return;
  bcr_seg_forward_body<PIVOT, N>(a, (int)blockIdx.x, sm);
}

// Two chain operations in ONE launch: blocks [0, nb0) run the (unpivoted) factorisation a0, the rest the (pivoted)
// solve a1.  In the resident NGD iteration the trial factorisation (log-det + marginals of Lam_trial, needs Vddmu only) and
// the gradient solve (dmu = Vddmu^-1 (-g)) are independent; on two streams they overlapped but the fork / join event packets
// left ~7 + ~12 us of gaps per iteration (profiles/r02_b kernel trace) -- side by side in the same three launches they cost
// neither a second stream nor an event.
template <int N>
__global__ __launch_bounds__(1024) void bcr_seg_forward_dual_kernel(SegArgs a0, SegArgs a1, int nb0) {
  extern __shared__ double sm[];
  if (pred_skip(a0.pred, a0.pred_val)) return;
  if ((int)blockIdx.x < nb0) bcr_seg_forward_body<false, N>(a0, (int)blockIdx.x, sm);
  else bcr_seg_forward_body<true, N>(a1, (int)blockIdx.x - nb0, sm);
}

// ---- pass C: backward recursion inside every segment of an earlier pass ----
// The segment's factors (E, GA, GB / v) are fetched from global ONCE into LDS, then the m levels run
// on LDS only.
template <int N>
__device__ __forceinline__ void bcr_seg_backward_body(const SegArgs& a, const int bid, double* sm) {
  constexpr int nn = N * N;
  const int T = a.T, S = a.S, st = 1 << a.level0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
  const bool rhs = a.rhs != nullptr;
  const int x0 = bid * S * st;
  int cnt = 0;
  for (int j = 0; j < S; ++j) if (x0 + j * st < T) cnt = j + 1;
  const int xn = x0 + S * st;                         // the next segment's first node (slot S)
  const bool ext_right = xn < T;
  [[maybe_unused]] const int tslot = rhs ? 5 : 2;
  [[maybe_unused]] int tix = 0;
  GVI_STAMP(tslot, tix++);
  if (rhs) {
    double* xl = sm;                                  // [S+1][N]
    double* vl = xl + (S + 1) * N;                    // [S][N]
    double* GAl = vl + S * N;                         // [S][nn]
    double* GBl = GAl + S * nn;                       // [S][nn]
    for (int e = tid; e < cnt * nn; e += blockDim.x) {
      const int j = e / nn, el = e % nn, x = x0 + j * st;
      if (j > 0) { GAl[e] = ws_GA(a)[(size_t)x * nn + el]; GBl[e] = ws_GB(a)[(size_t)x * nn + el]; }
    }
    for (int e = tid; e < cnt * N; e += blockDim.x) {
      const int j = e / N, r = e % N, x = x0 + j * st;
      if (j > 0) vl[e] = ws_v(a)[(size_t)x * N + r];
    }
    if (tid < N) xl[tid] = a.x[(size_t)x0 * N + tid];
    if (ext_right && tid >= 64 && tid < 64 + N) xl[S * N + tid - 64] = a.x[(size_t)xn * N + tid - 64];
    __syncthreads();
    GVI_STAMP(tslot, tix++);
    for (int lam = a.m - 1; lam >= 0; --lam) {
      const int h2 = 1 << lam;
      int nel = 0;
      for (int j = h2; j < cnt; j += 2 * h2) ++nel;
      for (int e = tid; e < nel * N; e += blockDim.x) {
        const int u = e / N, r = e % N, j = (2 * u + 1) * h2, x = x0 + j * st;
        const int ja = j - h2, jb = j + h2;
        const bool has_b = (jb < cnt) || (jb == S && ext_right);
        double xe = vl[j * N + r];
#pragma unroll
        for (int k = 0; k < N; ++k) xe = fma(-GAl[j * nn + r * N + k], xl[ja * N + k], xe);
        if (has_b) {
#pragma unroll
          for (int k = 0; k < N; ++k) xe = fma(-GBl[j * nn + r * N + k], xl[jb * N + k], xe);
        }
        xl[j * N + r] = xe;
        a.x[(size_t)x * N + r] = xe;
      }
      lds_barrier();
      GVI_STAMP(tslot, tix++);
    }
    return;
  }
  double* Sgl = sm;                                   // [S+1][nn] Sig_jj (slot S = next segment's first node)
  double* SLl = Sgl + (S + 1) * nn;                   // [S+1][nn]
  double* SRl = SLl + (S + 1) * nn;                   // [S+1][nn]; SRl[0] is preloaded with Sig[x0, xn]
  double* El = SRl + (S + 1) * nn;                    // [S][nn] factors of the segment's nodes
  double* GAl = El + S * nn;
  double* GBl = GAl + S * nn;
  // Sig[x0, xn]: the two are adjacent at level level0 + m; the odd one was eliminated there
  const int lvl_up = a.level0 + a.m;
  const bool x0_odd = ext_right && (((x0 >> lvl_up) & 1) != 0);
  for (int e0 = tid; e0 < cnt * nn; e0 += 2 * blockDim.x) {      // two elements per round: one memory round trip
    double ev[2], gav[2], gbv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      const int j = e / nn, el = e % nn, x = x0 + j * st;
      ev[u] = gav[u] = gbv[u] = 0.0;
      if (e < cnt * nn && j > 0) {
        ev[u] = ws_E(a)[(size_t)x * nn + el];
        gav[u] = ws_GA(a)[(size_t)x * nn + el];
        gbv[u] = ws_GB(a)[(size_t)x * nn + el];      // unused garbage when the node had no right neighbour
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      if (e < cnt * nn && e >= nn) { El[e] = ev[u]; GAl[e] = gav[u]; GBl[e] = gbv[u]; }
    }
  }
  for (int el = tid; el < nn; el += blockDim.x) {
    const int r = el / N, c = el % N;
    Sgl[el] = a.SigD[(size_t)x0 * nn + el];
    if (ext_right) {
      Sgl[S * nn + el] = a.SigD[(size_t)xn * nn + el];
      SRl[el] = x0_odd ? ws_SR(a)[(size_t)x0 * nn + el] : ws_SL(a)[(size_t)xn * nn + c * N + r];
    }
  }
  __syncthreads();
  GVI_STAMP(tslot, tix++);
  for (int lam = a.m - 1; lam >= 0; --lam) {
    const int h2 = 1 << lam;
    int nel = 0;
    for (int j = h2; j < cnt; j += 2 * h2) ++nel;
    for (int u = wave; u < nel; u += nwaves) {
      const int j = (2 * u + 1) * h2, x = x0 + j * st;
      const int ja = j - h2, jb = j + h2;
      const bool has_b = (jb < cnt) || (jb == S && ext_right);
      // Sig[a,b]: the segment boundary pair was preloaded into SRl[0]; otherwise the odd one of (a, b)
      // at the next local level holds it (SRl[ja] = Sig[a, b] or SLl[jb] = Sig[b, a])
      bool transposed = false;
      const double* Sab = SRl;
      if (has_b && !(ja == 0 && jb == S)) {
        const bool a_odd = ((ja / (2 * h2)) & 1) != 0;
        Sab = a_odd ? SRl + ja * nn : SLl + jb * nn;
        transposed = !a_odd;
      }
      seg_marginal_node<N>(a, x, x0 + ja * st, a.level0 + lam, has_b, El + j * nn, GAl + j * nn, GBl + j * nn,
                           Sgl + ja * nn, Sgl + (has_b ? jb : 0) * nn, Sab, transposed, Sgl + j * nn, SLl + j * nn,
                           SRl + j * nn, lane);
    }
    lds_barrier();
    GVI_STAMP(tslot, tix++);
  }
}

template <int N>
__global__ __launch_bounds__(1024) void bcr_seg_backward_kernel(SegArgs a) {
  extern __shared__ double sm[];
  if (pred_skip(a.pred, a.pred_val)) return;
  bcr_seg_backward_body<N>(a, (int)blockIdx.x, sm);
}

template <int N>
__global__ __launch_bounds__(1024) void bcr_seg_backward_dual_kernel(SegArgs a0, SegArgs a1, int nb0) {
  extern __shared__ double sm[];
  if (pred_skip(a0.pred, a0.pred_val)) return;
  if ((int)blockIdx.x < nb0) bcr_seg_backward_body<N>(a0, (int)blockIdx.x, sm);
  else bcr_seg_backward_body<N>(a1, (int)blockIdx.x - nb0, sm);
}

}  // namespace gvi
