// Block-tridiagonal chain operations (solve / log-det / tridiagonal blocks of the inverse) by SEGMENTED block cyclic
// reduction with register-resident eliminations -- round 3 rewrite of kernels_bcr_seg.hpp.
//
// Replaces, per SURVEY.md section 8(a):
//   a12      ConjugateGradient solve (ngd/NGD-GH-impl.h:59-60)
//   a14      SimplicialLDLT log-det (gvibase/GVI-GH-impl.h:192-196)
//   a16/a17  inv_sparse / inverse_GBP (helpers/EigenWrapper.h:282-381, gvibase/GVI-GH-GBP-impl.h:246-342)
//
// Algorithm (unchanged): level l keeps the nodes that are multiples of 2^l and eliminates the odd ones; all eliminations
// of a level are independent (one wave each), ceil(log2 T) levels.  Eliminating node e with neighbours a = e - 2^l,
// b = e + 2^l and couplings Ua = A[a,e], Ub = A[e,b]:
//     Gauss-Jordan [D_e | I | Ua^T | Ub | y_e] -> [I | E | GA | GB | v]          (E = D_e^-1, GA = E Ua^T, GB = E Ub)
//     D_a -= Ua GA,   D_b -= Ub^T GB,   A[a,b] = -Ua GB,   y_a -= Ua v,   y_b -= Ub^T v
//   back-substitution   x_e = v - GA x_a - GB x_b
//   selected inverse    Sig[e,a] = -(GA Sig_aa + GB Sig_ba),  Sig[e,b] = -(GA Sig_ab + GB Sig_bb),
//                       Sig_ee = E - Sig[e,a] GA^T - Sig[e,b] GB^T          (level 0 = the tridiagonal blocks)
//   1/2 log det         = 1/2 sum of the log-pivots of all eliminations (positive pivots <=> positive definite)
//
// What a level costs is latency, not throughput (profiles/r02_bcr_phases.txt: 0.3 us of a 1.9 us elimination was the
// Gauss-Jordan).  Round-3 form of an elimination -- NO LDS round trip and NO intra-elimination synchronisation:
//   * lane c holds column c of the tile in N registers; pivot columns are broadcast with v_readlane (as before);
//   * the Schur products stay in the same lanes: with every coupling block kept TRANSPOSED in LDS (XT[c][r] = X[r][c]),
//     t = -Ua col and s = -Ub^T col need only broadcast reads of contiguous rows (ds_read_b128, same address in every
//     lane): lanes of the GA columns end up with -Ua GA, lanes of the GB columns with A[a,b] (in t) and -Ub^T GB (in s),
//     the rhs lane with the updates of y_a / y_b;
//   * the updates go straight into the neighbours: every node has TWO accumulators in LDS, Dl (own block + updates from
//     its left) and Rl (updates from its right).  Per level each of them is written by exactly one wave (ds_add_f64 to
//     distinct addresses: deterministic), so the separate "eager update" phase and its barrier are gone: ONE workgroup
//     barrier per level;
//   * log-pivots: mantissa product / exponent sum per WAVE over all its eliminations of the pass; one entry per wave
//     goes to global memory at the end of the pass.
// Selected inverse: lane = element (r, c); all operands row-contiguous in LDS (every Sig[e, neighbour] block is kept
// in both orientations), 16-byte reads.
//
// Passes (host: chain_plan): A -- every workgroup owns S = 2^m consecutive alive nodes and runs m levels on them in LDS;
// B (top) -- one workgroup takes the survivors through the remaining levels, the root, the log-det reduction and the
// backward recursion of those nodes; C -- backward recursion inside every segment.  What crosses a segment boundary is
// accumulated in LDS during the pass and written ONCE at its end: Rs[x0] (updates of the segment's first node from its
// right), Ls[x0 + S st] (updates of the next segment's first node from its left; double-buffered by pass parity -- the
// next segment's workgroup reads the previous pass's value at load), the coupling of the two.
//
// One kernel serves a factorisation (blocks [0, nb0): unpivoted, with E) and a pivoted solve (blocks [nb0, ...)) side
// by side -- the NGD iteration's trial factorisation and gradient solve are independent; a single operation is the same
// launch with the other block range empty.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"
#include "kernels_bt.hpp"

namespace gvi {

constexpr int CHAIN_MATS = 10;   // E, GAt, GBt, NUt, Deff, Ls[2], Rs, SL, SR      each [T][N][N]
constexpr int CHAIN_VECS = 5;    // v, yLs[2], yRs, yeff                           each [T][N]

__host__ __device__ inline size_t chain_lp_entries(int T) { return (size_t)2 * T + 1024; }
__host__ __device__ inline size_t chain_ws_doubles(int T, int n) {
  return (size_t)CHAIN_MATS * T * n * n + (size_t)CHAIN_VECS * T * n + chain_lp_entries(T);
}

struct ChainArgs {
  int T;
  int n;             // block size of the caller's arrays (D, U, rhs, SigD, SigU, x, mix*); the kernels run at the compile-time
                     // size N >= n with the blocks padded by the identity (workspace and LDS are in the padded layout)
  int level0;        // first level of this pass; node spacing st = 1 << level0
  int m;             // levels handled in this pass
  int S;             // local slots per workgroup (segment size 2^m, or the number of alive nodes in the top pass)
  int first;         // first pass: reads D / U / rhs (and forms the trial precision); else the previous pass's survivors
  int par;           // pass parity (Ls / yLs double buffer)
  int need_back;     // factorisation: selected inverse wanted (else log-det only)
  int lp_off;        // first log-pivot entry of this pass (top pass: number of entries written by the earlier passes)
  const double* D;
  const double* U;
  const double* rhs;     // solve only
  double rhs_scale;
  double* ws;            // chain_ws_doubles(T, N)
  int* wsi;              // chain_lp_entries(T)
  double* SigD;          // [T][N][N]   (marginals)
  double* SigU;          // [T-1][N][N]
  double* x;             // [T][N]      (solve)
  double* hld;           // [1]  1/2 log det, NaN when a pivot was not positive
  // optional fused trial precision (first pass): the chain operated on is D + mix_step (mixV - D), written to mixOut; both
  // are [D[T] | U[T-1]] in ONE buffer (the U part starts at block T)
  const double* mixV;
  double* mixOut;
  double mix_step;
  // optional assemble-on-load (first pass; chain-structured factor sets, AsmList below): the matrix V = [V_D | V_U] and the
  // vector g are not read but ASSEMBLED from the per-factor results (the sums of bt_scatter_all_kernel, same order):
  //   solve (rhs != null):  (D, U, rhs) <- assembled (V_D, V_U, g), written out to asmD / asmU / asmG on the way;
  //   factorisation:        the mixed-in matrix mixV <- assembled (the chain operated on is D + mix_step (V - D)).
  int asm_on;
  double* asmD;
  double* asmU;
  double* asmG;
  const double* pred;    // predicated launch (device_common.hpp, pred_skip) or null
  double pred_val;
  // top pass + first backward pass in ONE launch (chain_top_back_kernel): the top pass's workgroup stores sync_seq to *sync
  // (release, agent scope) behind its last store; the backward workgroups wait for it in front of their first load of what
  // the top pass wrote (chain_wait).  null: separate launches
  unsigned* sync;
  unsigned sync_seq;
  unsigned sync_fault;   // test hook: added to the word the producer stores (the consumers then time out)
};
// what changes from pass to pass (chain_launch.hpp::ChainPass), for the second pass of a merged launch
struct ChainPassDev { int level0, m, S, first, par, lp_off; };

// Chain-structured factor sets for the assemble-on-load: factor k of a binary set (d = 2n) couples states k, k + 1, factor
// k of a unary set (d = n) sits on state k -- start[k] = k, so no CSR indirection (three INDEPENDENT loads per element
// instead of ptr -> idx -> value).  Other graphs keep the stand-alone assemble launch.
// nsp > 0: a SPARSE unary set (d = n) of nsp <= ASM_SPARSE_MAX factors on arbitrary states sp[] (the anchors of a planning
// graph: start / goal): compared against immediates, no index load
constexpr int ASM_SPARSE_MAX = 4;
struct AsmSet { int K, d; const double* Vdmu; const double* Vddmu; int nsp; int sp[ASM_SPARSE_MAX]; };
struct AsmList { int nsets; AsmSet s[MAX_SETS]; };

// element (r, c) of V_D[t] (which = 0) / V_U[t] (which = 1), or entry r of g[t] (which = 2): bt_scatter_all_kernel's sums
__device__ __forceinline__ double asm_element(const AsmList& L, const int n, const int t, const int which, const int r, const int c) {
  double acc = 0.0;
  for (int si = 0; si < L.nsets; ++si) {
    const AsmSet& a = L.s[si];
    const int d = a.d;
    const bool two = d == 2 * n;
    double s = 0.0;
    if (a.nsp > 0) {                                   // sparse unary set: factor k sits on state sp[k] (ascending k = CSR order)
#pragma unroll
      for (int k = 0; k < ASM_SPARSE_MAX; ++k) {
        if (k < a.nsp && a.sp[k] == t) {
          if (which == 2) s += a.Vdmu[(size_t)k * d + r];
          else if (which == 0) s += a.Vddmu[(size_t)k * d * d + r * d + c];
        }
      }
    } else if (which == 2) {
      if (t < a.K) s += a.Vdmu[(size_t)t * d + r];
      if (two && t > 0 && t - 1 < a.K) s += a.Vdmu[(size_t)(t - 1) * d + n + r];
    } else if (which == 0) {
      if (t < a.K) s += a.Vddmu[(size_t)t * d * d + r * d + c];
      if (two && t > 0 && t - 1 < a.K) s += a.Vddmu[(size_t)(t - 1) * d * d + (n + r) * d + n + c];
    } else if (two && t < a.K) s += a.Vddmu[(size_t)t * d * d + r * d + n + c];
    acc += s;                 // same association as bt_scatter_all_kernel
  }
  return acc;
}

// asm_element for the two elements a thread of the load phase handles, (V_D, V_U) of each, in ONE sweep over the sets: every
// load of a set is issued before the first add.  asm_element's loop waits for its loads inside every iteration, so four calls
// in a row were up to 4 x nsets dependent round trips in front of a pass's first level.  Same sums, same order.
__device__ __forceinline__ void asm_pair(const AsmList& L, const int n, const int (&t)[2], const int (&r)[2], const int (&c)[2],
                                         const bool (&onD)[2], const bool (&onU)[2], double (&vD)[2], double (&vU)[2]) {
  vD[0] = vD[1] = vU[0] = vU[1] = 0.0;
  for (int si = 0; si < L.nsets; ++si) {
    const AsmSet& a = L.s[si];
    const int d = a.d;
    const bool two = d == 2 * n;
    double d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0}, u0[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!onD[u]) continue;
      if (a.nsp > 0) {
#pragma unroll
        for (int k = 0; k < ASM_SPARSE_MAX; ++k)
          if (k < a.nsp && a.sp[k] == t[u]) d0[u] += a.Vddmu[(size_t)k * d * d + r[u] * d + c[u]];
      } else {
        if (t[u] < a.K) d0[u] = a.Vddmu[(size_t)t[u] * d * d + r[u] * d + c[u]];
        if (two && t[u] > 0 && t[u] - 1 < a.K) d1[u] = a.Vddmu[(size_t)(t[u] - 1) * d * d + (n + r[u]) * d + n + c[u]];
        if (onU[u] && two && t[u] < a.K) u0[u] = a.Vddmu[(size_t)t[u] * d * d + r[u] * d + n + c[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { vD[u] += (0.0 + d0[u]) + d1[u]; vU[u] += 0.0 + u0[u]; }
  }
}

#ifndef GVI_CHAIN_UBLATE
#define GVI_CHAIN_UBLATE 1     // 0: the rows of Ub fetched row by row inside the Schur products (A/B build)
#endif
#ifndef GVI_CHAIN_THREADS_SMALL
#define GVI_CHAIN_THREADS_SMALL 1024
#endif
constexpr int chain_threads(int n) { return n <= 8 ? GVI_CHAIN_THREADS_SMALL : 512; }

// -DGVI_CHAIN_TIMING (tools/ubench/chain_bench.hip only): shader-clock stamps of wave 0 of the top pass's factorisation
#ifdef GVI_CHAIN_TIMING
__device__ unsigned long long gvi_chain_stamps[256];
__device__ int gvi_chain_nstamp;
#ifdef GVI_CHAIN_TIMING_A       // block 0 of the segmented passes (grids of more than two blocks) instead of the top pass
#define CHAIN_STAMP(on) do { if (gridDim.x > 2 && blockIdx.x == 0 && threadIdx.x == 0) { const int i__ = gvi_chain_nstamp; if (i__ < 256) { gvi_chain_stamps[i__] = __builtin_amdgcn_s_memtime(); gvi_chain_nstamp = i__ + 1; } } } while (0)
#else
#define CHAIN_STAMP(on) do { if ((on) && threadIdx.x == 0) { const int i__ = gvi_chain_nstamp; if (i__ < 256) { gvi_chain_stamps[i__] = __builtin_amdgcn_s_memtime(); gvi_chain_nstamp = i__ + 1; } } } while (0)
#endif
#else
#define CHAIN_STAMP(on) do { } while (0)
#endif

namespace chain {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void ld_row(const double* p, double (&o)[N]) {
  if constexpr (N % 2 == 0) {
    const d2* q = (const d2*)p;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { const d2 t = q[i]; o[2 * i] = t.x; o[2 * i + 1] = t.y; }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = p[i];
  }
}
template <int N>
__device__ __forceinline__ void st_row(double* p, const double (&v)[N]) {
  if constexpr (N % 2 == 0) {
    d2* q = (d2*)p;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { d2 t; t.x = v[2 * i]; t.y = v[2 * i + 1]; q[i] = t; }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = v[i];
  }
}

__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// a result other workgroups of the SAME launch read (wt): written through at agent scope
__device__ __forceinline__ void st_out(double* p, const double v, const bool wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

__device__ __forceinline__ double readlane64(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// element el = r N + c of a padded block -> offset inside the caller's n x n block, -1 for the padding
template <int N> __device__ __forceinline__ int ext_el(const int n, const int el) {
  if (n == N) return el;
  const int r = el / N, c = el % N;
  return (r < n && c < n) ? r * n + c : -1;
}

// workspace arrays
template <int N> __device__ __forceinline__ double* ws_mat(const ChainArgs& a, int idx) { return a.ws + (size_t)idx * a.T * (N * N); }
template <int N> __device__ __forceinline__ double* ws_vec(const ChainArgs& a, int idx) {
  return a.ws + (size_t)CHAIN_MATS * a.T * (N * N) + (size_t)idx * a.T * N;
}
template <int N> __device__ __forceinline__ double* ws_lp(const ChainArgs& a) { return ws_vec<N>(a, CHAIN_VECS); }
enum { W_E = 0, W_GA = 1, W_GB = 2, W_NU = 3, W_DEFF = 4, W_LS = 5 /* +par */, W_RS = 7, W_SL = 8, W_SR = 9 };
enum { V_V = 0, V_YLS = 1 /* +par */, V_YRS = 3, V_YEFF = 4 };

// accumulated log-pivots of a wave: product of the pivots' mantissas (renormalised), sum of their exponents, "a pivot was
// not positive"
struct LogPiv {
  double m;
  int e, bad;
};

// column layout of the elimination tile [D_e | I | Ua^T | Ub | y]
template <bool HAS_E, bool HAS_Y, int N>
struct Cols {
  static constexpr int cE = N;
  static constexpr int cA = HAS_E ? 2 * N : N;
  static constexpr int cB = cA + N;
  static constexpr int cY = cB + N;
  static constexpr int NC = cY + (HAS_Y ? 1 : 0);
  static_assert(NC <= 64, "tile does not fit one wave");
};

// LDS offsets (doubles, relative to sm) an elimination works with; -1 = absent
struct ElimIO {
  int oD, oR;          // own block, its right-hand accumulator
  int oUa, oUb;        // couplings A[a,e], A[e,b] in TRANSPOSED layout
  int oRa, oDb;        // targets: right-hand accumulator of a, block (left-hand accumulator) of b
  int oNU;             // new coupling A[a,b] (transposed layout), slot of e
  int oy, oyR, oyRa, oyb;   // rhs: own, own right-hand accumulator, targets
  int oZero;           // N zeros
  // top pass: factors stay in LDS, row-major
  int oEl, oGAl, oGBl, ovl;
};

// ---- forward elimination of one node by one wave ----
template <bool PIVOT, bool HAS_E, bool HAS_Y, bool TOP, int N>
__device__ __forceinline__ void eliminate(double* sm, const int lane, const ElimIO io, double* gE, double* gGA, double* gGB,
                                          double* gv, double* gNU, LogPiv& lp) {
  using C = Cols<HAS_E, HAS_Y, N>;
  const bool has_a = io.oUa >= 0, has_b = io.oUb >= 0;              // wave-uniform
  const bool isD = lane < N;
  const bool isE = HAS_E && lane >= C::cE && lane < C::cA;
  const bool isA = lane >= C::cA && lane < C::cB;
  const bool isB = lane >= C::cB && lane < C::cY;
  const bool isY = HAS_Y && lane == C::cY;
  const int cc = isD ? lane : (isE ? lane - C::cE : (isA ? lane - C::cA : (isB ? lane - C::cB : 0)));   // column inside its group
  // ---- this lane's column: col[r] = sm[p0 + r s0] + sm[p1 + r] ----
  int p0 = io.oZero, s0 = 0, p1 = io.oZero;
  if (isD) { p0 = io.oD + cc * N; s0 = 1; p1 = io.oR + cc * N; }                       // symmetric: row = column
  else if (isA) { if (has_a) { p0 = io.oUa + cc; s0 = N; } }                          // column c of Ua^T = UaT[:, c]
  else if (isB) { if (has_b) { p0 = io.oUb + cc * N; s0 = 1; } }                      // column c of Ub  = UbT[c, :]
  else if (isY) { p0 = io.oy; s0 = 1; p1 = io.oyR; }
  CHAIN_STAMP(TOP && HAS_E);
  double col[N];
#pragma unroll
  for (int r = 0; r < N; ++r) col[r] = sm[p0 + r * s0] + sm[p1 + r];
  CHAIN_STAMP(TOP && HAS_E);
  if (HAS_E) {
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = isE ? (r == cc ? 1.0 : 0.0) : col[r];
  }
  // The rows of Ua the Schur products will read are known now: request them BEFORE the Gauss-Jordan (which covers their
  // latency) where the register budget allows (N <= 6 at 128 registers)
  constexpr int RB = chain_threads(N) == 1024 ? 128 : 256;              // register budget per lane
  constexpr bool PRE = N * N * 2 <= (RB == 128 ? 72 : 100);
  double uap[PRE ? N : 1][N];
  if (PRE && has_a) {
#pragma unroll
    for (int k = 0; k < (PRE ? N : 1); ++k) ld_row<N>(sm + io.oUa + k * N, uap[k]);            // UaT[k][:] = Ua[:, k]
  }
  // ---- Gauss-Jordan, pivot columns broadcast with v_readlane ----
  double pivs[N];
  int bad = 0;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    if (PIVOT) {
      // threshold partial pivoting, searched inside lane p (the pivot column lives in that lane's registers): rows are
      // swapped only when the natural pivot is more than 8x smaller than the column maximum -- rare for the near-SPD blocks
      // of an NGD iteration, so the common path is one wave-uniform branch
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
    for (int r = 0; r < N; ++r) ap[r] = readlane64(col[r], p);
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
  CHAIN_STAMP(TOP && HAS_E);
  // ---- log-pivots (wave-uniform values; every lane keeps the same accumulator) ----
  {
    double mp = 1.0;
    int es = 0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      mp *= __builtin_amdgcn_frexp_mant(pivs[p]);
      es += __builtin_amdgcn_frexp_exp(pivs[p]);
    }
    const double t = lp.m * mp;
    lp.m = __builtin_amdgcn_frexp_mant(t);
    lp.e += es + __builtin_amdgcn_frexp_exp(t);
    lp.bad |= bad;
  }
  CHAIN_STAMP(TOP && HAS_E);
  // ---- factors: E / GA / GB / v of this node ----
  if (TOP) {                                               // kept in LDS, row-major, for the backward recursion of this pass
    if (lane >= N && lane < C::NC) {
      const int o = isE ? io.oEl + cc : (isA ? io.oGAl + cc : (isB ? io.oGBl + cc : io.ovl));
      const int so = isY ? 1 : N;
#pragma unroll
      for (int r = 0; r < N; ++r) sm[o + r * so] = col[r];
    }
  } else {                                                 // to the workspace (column-contiguous = transposed), read by pass C
    if (lane >= N && lane < C::NC) {
      double* g = isE ? gE + cc * N : (isA ? gGA + cc * N : (isB ? gGB + cc * N : gv));
      st_row<N>(g, col);
    }
  }
  CHAIN_STAMP(TOP && HAS_E);
  // ---- Schur products in place: t = -Ua col, s = -Ub^T col (broadcast reads of contiguous rows) ----
  // t first (its operands are in registers when prefetched); the rows of Ub are requested while t is formed
  constexpr bool UBPRE = N * N * 2 <= (RB == 128 ? 32 : 100);  // both couplings in registers at once
  double t[N], ub[UBPRE ? N : 1][N];
#pragma unroll
  for (int r = 0; r < N; ++r) t[r] = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (has_a) {
      double ua[N];
      if (PRE) {
#pragma unroll
        for (int r = 0; r < N; ++r) ua[r] = uap[PRE ? k : 0][r];
      } else ld_row<N>(sm + io.oUa + k * N, ua);            // UaT[k][:] = Ua[:, k]
#pragma unroll
      for (int r = 0; r < N; ++r) t[r] = fma(-ua[r], col[k], t[r]);
    }
    if (UBPRE && has_b) ld_row<N>(sm + io.oUb + k * N, ub[UBPRE ? k : 0]);   // UbT[k][:] = Ub[:, k]
  }
  if (has_a) {
    if (isA || isY) {                                      // D_a -= Ua GA (column c) ; y_a -= Ua v
      double* o = sm + (isY ? io.oyRa : io.oRa + cc * N);
#pragma unroll
      for (int r = 0; r < N; ++r) lds_add(o + r, t[r]);
    }
    if (has_b && isB) {                                    // A[a,b] = -Ua GB, column c contiguous = transposed layout
      st_row<N>(sm + io.oNU + cc * N, t);
      if (gNU) st_row<N>(gNU + cc * N, t);
    }
  }
  if (has_b) {
    double s[N];
    // UBLATE (N <= 6 at 128 registers): the rows of Ub are all requested HERE -- the prefetched rows of Ua are dead once t is
    // formed, so their registers hold Ub -- with one wait in front of the products.  Left to the scheduler the eighteen
    // ds_read_b128 went out in six groups with a wait each: six dependent LDS round trips per elimination (ISA of N = 6).
    // chain_bench T = 1025, n = 6: 47.36 -> 46.83 us.  (The same for the operand rows of the backward step: no gain, 47.4.)
    constexpr bool UBLATE = !UBPRE && GVI_CHAIN_UBLATE != 0 && N * N * 2 <= (RB == 128 ? 72 : 0);
    double ubl[UBLATE ? N : 1][N];
    if constexpr (UBLATE) {
#pragma unroll
      for (int r = 0; r < N; ++r) ld_row<N>(sm + io.oUb + r * N, ubl[r]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      double ubr[N];
      if (UBPRE) {
#pragma unroll
        for (int k = 0; k < N; ++k) ubr[k] = ub[UBPRE ? r : 0][k];
      } else if constexpr (UBLATE) {
#pragma unroll
        for (int k = 0; k < N; ++k) ubr[k] = ubl[UBLATE ? r : 0][k];
      } else ld_row<N>(sm + io.oUb + r * N, ubr);          // UbT[r][:] = Ub[:, r]
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) acc = fma(-ubr[k], col[k], acc);
      s[r] = acc;
    }
    if (isB || isY) {                                      // D_b -= Ub^T GB (column c) ; y_b -= Ub^T v
      double* o = sm + (isY ? io.oyb : io.oDb + cc * N);
#pragma unroll
      for (int r = 0; r < N; ++r) lds_add(o + r, s[r]);
    }
  }
  CHAIN_STAMP(TOP && HAS_E);
}

// ---- backward step of one node: selected inverse; lane = element (r, c) ----
// Factors E / GA / GB row-major in LDS; Saa / Sbb symmetric; X = Sig_ab, Y = Sig_ba, both row-major.
template <int N>
__device__ __forceinline__ void marginal_node(double* sm, const int lane, const bool has_b, const int oE, const int oGA, const int oGB,
                                              const int oSaa, const int oSbb, const int oX, const int oY, const int oSee,
                                              const int oSL, const int oSLt, const int oSR, const int oSRt, double* gSee, double* gSL,
                                              double* gSR, double* gUa, double* gUe, const int n,
                                              [[maybe_unused]] const bool stamp = false, const bool wt = false) {
  // wt: Sig[e,e] and the workspace copies of Sig[e,a] / Sig[e,b] are read by OTHER workgroups of this launch (merged top +
  // backward launch): stored write-through at agent scope
  constexpr int nn = N * N, R = (nn + 63) / 64;
  double slv[R], srv[R];
  CHAIN_STAMP(stamp);
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int el = q * 64 + lane;
    const bool on = el < nn;
    const int r = on ? el / N : 0, c = on ? el % N : 0;
    double ga[N], w[N];
    ld_row<N>(sm + oGA + r * N, ga);
    ld_row<N>(sm + oSaa + c * N, w);
    double sl = 0.0, sr = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) sl = fma(ga[k], w[k], sl);
    if (has_b) {
      double gb[N];
      ld_row<N>(sm + oGB + r * N, gb);
      ld_row<N>(sm + oX + c * N, w);                      // Sig_ba[k][c] = Sig_ab[c][k]
#pragma unroll
      for (int k = 0; k < N; ++k) sl = fma(gb[k], w[k], sl);
      ld_row<N>(sm + oY + c * N, w);                      // Sig_ab[k][c] = Sig_ba[c][k]
#pragma unroll
      for (int k = 0; k < N; ++k) sr = fma(ga[k], w[k], sr);
      ld_row<N>(sm + oSbb + c * N, w);
#pragma unroll
      for (int k = 0; k < N; ++k) sr = fma(gb[k], w[k], sr);
    }
    slv[q] = -sl;
    srv[q] = -sr;
    if (on) {
      sm[oSL + el] = -sl;
      sm[oSLt + c * N + r] = -sl;
      if (has_b) { sm[oSR + el] = -sr; sm[oSRt + c * N + r] = -sr; }
      if (gSL) { st_out(gSL + el, -sl, wt); if (has_b) st_out(gSR + el, -sr, wt); }
      if (gUa && r < n && c < n) { gUa[c * n + r] = -sl; if (has_b) gUe[r * n + c] = -sr; }      // Sig[a,e] = Sig[e,a]^T ; Sig[e,b]
    }
  }
  CHAIN_STAMP(stamp);
  wave_lds_sync();
  CHAIN_STAMP(stamp);
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int el = q * 64 + lane;
    const bool on = el < nn;
    const int r = on ? el / N : 0, c = on ? el % N : 0;
    double u[N], w[N];
    double see = sm[oE + (on ? el : 0)];
    ld_row<N>(sm + oSL + r * N, u);
    ld_row<N>(sm + oGA + c * N, w);
#pragma unroll
    for (int k = 0; k < N; ++k) see = fma(-u[k], w[k], see);
    if (has_b) {
      ld_row<N>(sm + oSR + r * N, u);
      ld_row<N>(sm + oGB + c * N, w);
#pragma unroll
      for (int k = 0; k < N; ++k) see = fma(-u[k], w[k], see);
    }
    if (on) {
      sm[oSee + el] = see;
      if (r < n && c < n) st_out(gSee + r * n + c, see, wt);
    }
  }
  CHAIN_STAMP(stamp);
}

// LDS footprint (doubles)
template <bool HAS_E, bool HAS_Y, bool TOP, int N>
__host__ __device__ constexpr size_t fwd_lds_doubles(int S) {
  constexpr int nn = N * N;
  size_t w = (size_t)2 * (S + 1) * nn + (size_t)2 * S * nn;              // Dl, Rl, Ct, NUt
  if (TOP) w += (size_t)3 * S * nn + (HAS_E ? (size_t)S * nn : 0);       // El, GAl, GBl (+ SRt of the backward recursion)
  if (HAS_Y) w += (size_t)2 * (S + 1) * N + (TOP ? (size_t)(2 * S + 1) * N : 0);   // yl, yRl (+ vl, xl)
  return w + N + (N & 1) + 128;                                          // zero block, log-det reduction
}
template <bool HAS_E, int N>
__host__ __device__ constexpr size_t bwd_lds_doubles(int S) {
  constexpr int nn = N * N;
  return (HAS_E ? (size_t)5 * (S + 1) * nn + (size_t)3 * S * nn            // Sg, SL, SLt, SR, SRt ; E, GA, GB
                : (size_t)(S + 1) * N + (size_t)S * N + (size_t)2 * S * nn)   // x, v, GA, GB
         + 2;                                                              // chain_wait's word
}

// ---- hand-over inside a launch (chain_top_back_kernel) ----
// No fences: a release at agent scope writes the XCD's whole L2 back and an acquire invalidates it -- with both the merged
// launch took exactly as long as the two launches it replaces (chain_bench, T = 1025, n = 6: 47.6 vs 47.4 us).  Instead the
// producer stores what the consumers read WRITE-THROUGH at agent scope (st_out), every thread waits for its stores at the
// barrier (vmcnt(0): the workgroup-scope release of __syncthreads), and thread 0 stores the word; the consumers read those
// values -- and only those -- with agent-scope loads (chain_ld) behind the word.  Same protocol as the epilogue's tail
// (kernels_factor.hpp, epi_tail).
__device__ __forceinline__ void chain_signal(unsigned* sync, const unsigned seq) {
  if (!sync) return;
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(sync, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Consumer: thread 0 polls (the producer's workgroups have the launch's lowest indices: they are resident before any waiting
// one, and a waiting workgroup holds nothing the producer needs).  The wait is bounded (1 s on the 100 MHz counter), after
// which the caller poisons what it would have read -- NaN results, a rejected step, never a hung device.  Returns false on
// the time-out (block-uniform).  (ChainArgs::sync_fault, option chain_merge = 2: the producer stores a wrong word -- the
// test of this path.)
constexpr unsigned long long CHAIN_WAIT_TICKS = 100000000ull;   // 1 s: far above a time slice of a GPU shared between processes
__device__ __forceinline__ bool chain_wait(const unsigned* sync, const unsigned seq, int* lds_word) {
  if (!sync) return true;
  if (threadIdx.x == 0) {
    int ok = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    do {
      if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < CHAIN_WAIT_TICKS);
    *lds_word = ok;
  }
  __syncthreads();                                     // (no load of a consumer is issued before thread 0 has seen the word)
  return *lds_word != 0;
}
// a value the top pass of the same launch wrote: read at agent scope (never from a stale line of this XCD's L2)
__device__ __forceinline__ double chain_ld(const double* p, const bool fresh) {
  return fresh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// ---- passes A / B: one workgroup per segment ----
template <bool PIVOT, bool HAS_E, bool HAS_Y, bool TOP, int N>
__device__ __forceinline__ void forward_body(const ChainArgs& a, const AsmList& AL, const int bid, double* sm) {
  constexpr int nn = N * N;
  const LazyPred lpred = pred_issue(a.pred, a.pred_val);        // checked in front of the first store (device_common.hpp)
  const int T = a.T, S = a.S, st = 1 << a.level0;
  // the wave index as a SCALAR: everything derived from it (node, offsets, has_b) is then wave-uniform for the compiler too
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = blockDim.x >> 6, nthr = blockDim.x;
  const int x0 = bid * S * st;
  // LDS map
  const int oDl = 0;                               // [S+1][nn] block + updates from the left (slot S: the next segment's first node)
  const int oRl = oDl + (S + 1) * nn;              // [S+1][nn] updates from the right
  const int oCt = oRl + (S + 1) * nn;              // [S][nn]   coupling A[x_j, x_{j+1}] at this pass's spacing, transposed
  const int oNU = oCt + S * nn;                    // [S][nn]   couplings created in this pass, by eliminated node, transposed
  const int oEl = oNU + S * nn;                    // top: [S][nn] x 3 factors, row-major
  const int oGAl = oEl + (TOP ? S * nn : 0);
  const int oGBl = oGAl + (TOP ? S * nn : 0);
  const int oSRt = oGBl + (TOP ? S * nn : 0);      // top, selected inverse: [S][nn]
  const int oyl = oSRt + ((TOP && HAS_E) ? S * nn : 0);   // [S+1][N]
  const int oyR = oyl + (HAS_Y ? (S + 1) * N : 0);        // [S+1][N]
  const int ovl = oyR + (HAS_Y ? (S + 1) * N : 0);        // top: [S][N]
  const int oxl = ovl + ((HAS_Y && TOP) ? S * N : 0);     // top: [S+1][N]
  const int oZero = oxl + ((HAS_Y && TOP) ? (S + 1) * N : 0);
  const int oRed = oZero + N + (N & 1);
  CHAIN_STAMP(TOP && HAS_E);
  const int cnt = min(S, (T - x0 + st - 1) >> a.level0);    // local nodes that exist (x0 < T for every launched block)
  const int xn = x0 + S * st;
  const bool ext_right = !TOP && xn < T;            // the next segment's first node exists
  const bool first = a.first != 0;
  const bool mix = first && (a.mixV != nullptr || (a.asm_on && !HAS_Y));
  [[maybe_unused]] double lpm = 1.0;
  [[maybe_unused]] int lpe = 0, lpb = 0;
  if constexpr (TOP) {
    if (a.hld) {
      for (int t = tid; t < a.lp_off; t += nthr) {     // thread-sequential product over its entries (usually one)
        const double mv = ws_lp<N>(a)[t];
        const int eb = a.wsi[t];
        const double pr = lpm * mv;
        lpm = __builtin_amdgcn_frexp_mant(pr);
        lpe += (eb >> 1) + __builtin_amdgcn_frexp_exp(pr);
        lpb |= eb & 1;
      }
    }
  }
  // ---- load (+ fold what the previous pass left at the segment boundaries) ----
  for (int e0 = tid; e0 < cnt * nn; e0 += 2 * nthr) {
    double dv[2], cu[2];
    bool on[2], hasc[2];
    // assemble-on-load: the V_D / V_U elements of both elements of this round in one sweep over the factor sets
    double aD[2] = {0.0, 0.0}, aU[2] = {0.0, 0.0};
    if (first && a.asm_on) {
      int at[2], ar[2], ac[2];
      bool aon[2], aonU[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * nthr;
        const bool o = e < cnt * nn;
        const int j = o ? e / nn : 0, el = o ? e % nn : 0, x = x0 + j * st;
        const int xe = ext_el<N>(a.n, el);
        at[u] = x; ar[u] = xe >= 0 ? xe / a.n : 0; ac[u] = xe >= 0 ? xe % a.n : 0;
        aon[u] = o && xe >= 0; aonU[u] = aon[u] && x + st < T;
      }
      asm_pair(AL, a.n, at, ar, ac, aon, aonU, aD, aU);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                   // all loads of the round are issued before the first use
      const int e = e0 + u * nthr;
      on[u] = e < cnt * nn;
      hasc[u] = false;
      dv[u] = cu[u] = 0.0;
      if (on[u]) {
        const int j = e / nn, el = e % nn, x = x0 + j * st;
        const size_t g = (size_t)x * nn + el;
        hasc[u] = x + st < T;
        if (first) {
          const int xe = ext_el<N>(a.n, el);
          const size_t ge = (size_t)x * (a.n * a.n) + xe, gu = (size_t)(T + x) * (a.n * a.n) + xe;
          const bool asmv = a.asm_on != 0 && xe >= 0;
          const double vD = aD[u], vU = aU[u];
          if (HAS_Y && asmv) {                            // the solve operates ON the assembled matrix and leaves it in memory
            dv[u] = vD;                                   // (written out with the other stores of the round, below)
            if (hasc[u]) cu[u] = vU;
          } else {
            dv[u] = xe >= 0 ? a.D[ge] : ((el / N == el % N) ? 1.0 : 0.0);                            // identity padding
            if (mix && xe >= 0) { const double mv = asmv ? vD : a.mixV[ge]; dv[u] = dv[u] + a.mix_step * (mv - dv[u]); }   // trial_kernel's arithmetic
            if (hasc[u] && xe >= 0) {
              cu[u] = a.U[ge];
              if (mix) { const double mv = asmv ? vU : a.mixV[gu]; cu[u] = cu[u] + a.mix_step * (mv - cu[u]); }
            }
          }
        } else {
          const double b0 = ws_mat<N>(a, W_DEFF)[g];
          const double l0 = x > 0 ? ws_mat<N>(a, W_LS + (a.par ^ 1))[g] : 0.0;
          const double r0 = ws_mat<N>(a, W_RS)[g];
          if (hasc[u]) cu[u] = ws_mat<N>(a, W_NU)[(size_t)(x + st / 2) * nn + el];
          dv[u] = (b0 + l0) + r0;
        }
      }
    }
    if (pred_fail(lpred)) return;                   // (block-uniform; nothing has been written yet)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * nthr;
      if (on[u]) {
        const int j = e / nn, el = e % nn, x = x0 + j * st;
        const size_t g = (size_t)x * nn + el;
        const int xe = first ? ext_el<N>(a.n, el) : -1;
        sm[oDl + e] = dv[u];
        if (mix && xe >= 0) a.mixOut[(size_t)x * (a.n * a.n) + xe] = dv[u];
        if (HAS_Y && first && a.asm_on && xe >= 0) {
          a.asmD[(size_t)x * (a.n * a.n) + xe] = dv[u];
          if (hasc[u]) a.asmU[(size_t)x * (a.n * a.n) + xe] = cu[u];
        }
        if (j == 0 && !TOP) ws_mat<N>(a, W_DEFF)[g] = dv[u];             // the survivor's base for the next pass
        if (hasc[u]) {
          if (first) { const int r = el / N, c = el % N; sm[oCt + j * nn + c * N + r] = cu[u]; }   // transposed
          else sm[oCt + e] = cu[u];                                                                 // NUt is transposed already
          if (mix && xe >= 0) a.mixOut[(size_t)(T + x) * (a.n * a.n) + xe] = cu[u];
        }
      }
    }
  }
  if (pred_fail(lpred)) return;
  for (int e = tid; e < (S + 1) * nn; e += nthr) sm[oRl + e] = 0.0;
  for (int e = tid; e < (S + 1 - cnt) * nn; e += nthr) sm[oDl + cnt * nn + e] = 0.0;
  if (tid < N + (N & 1)) sm[oZero + tid] = 0.0;
  if (HAS_Y) {
    for (int e = tid; e < (S + 1) * N; e += nthr) {
      const int j = e / N, r = e % N, x = x0 + j * st;
      double v = 0.0;
      if (j < cnt) {
        if (first) {
          if (a.asm_on && r < a.n) {
            const double gv = asm_element(AL, a.n, x, 2, r, 0);
            a.asmG[(size_t)x * a.n + r] = gv;
            v = a.rhs_scale * gv;
          } else v = r < a.n ? a.rhs_scale * a.rhs[(size_t)x * a.n + r] : 0.0;
        }
        else {
          const size_t g = (size_t)x * N + r;
          v = (ws_vec<N>(a, V_YEFF)[g] + (x > 0 ? ws_vec<N>(a, V_YLS + (a.par ^ 1))[g] : 0.0)) + ws_vec<N>(a, V_YRS)[g];
        }
        if (j == 0 && !TOP) ws_vec<N>(a, V_YEFF)[(size_t)x * N + r] = v;
      }
      sm[oyl + e] = v;
      sm[oyR + e] = 0.0;
    }
  }
  CHAIN_STAMP(TOP && HAS_E);
  __syncthreads();
  CHAIN_STAMP(TOP && HAS_E);
  if constexpr (TOP) {
    // log-pivot entries of the earlier passes (one per wave of every workgroup): product per wave of THIS workgroup now,
    // in a fixed order; the loads were requested before the chain loads' barrier
    if (a.hld) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double t = lpm * __shfl_xor(lpm, o);
        lpe += __shfl_xor(lpe, o) + __builtin_amdgcn_frexp_exp(t);
        lpm = __builtin_amdgcn_frexp_mant(t);
        lpb |= __shfl_xor(lpb, o);
      }
      if (lane == 0) { sm[oRed + 64 + wave] = lpm; ((int*)(sm + oRed + 96))[wave] = lpe * 2 + lpb; }
    }
  }
  // ---- m local levels: one barrier each ----
  LogPiv lp;
  lp.m = 1.0; lp.e = 0; lp.bad = 0;
  for (int lam = 0; lam < a.m; ++lam) {
    const int h2 = 1 << lam;
    const int nel = (cnt + h2 - 1) >> (lam + 1);            // odd multiples of h2 below cnt
    for (int u = wave; u < nel; u += nwaves) {
      const int j = (2 * u + 1) * h2, x = x0 + j * st;
      const int ja = j - h2, jb = j + h2;
      const bool has_b = (jb < cnt) || (jb == S && ext_right);
      ElimIO io;
      io.oD = oDl + j * nn; io.oR = oRl + j * nn;
      io.oUa = lam == 0 ? oCt + ja * nn : oNU + (j - h2 / 2) * nn;
      io.oUb = !has_b ? -1 : (lam == 0 ? oCt + j * nn : oNU + (j + h2 / 2) * nn);
      io.oRa = oRl + ja * nn; io.oDb = oDl + jb * nn;
      io.oNU = oNU + j * nn;
      io.oy = oyl + j * N; io.oyR = oyR + j * N; io.oyRa = oyR + ja * N; io.oyb = oyl + jb * N;
      io.oZero = oZero;
      io.oEl = oEl + j * nn; io.oGAl = oGAl + j * nn; io.oGBl = oGBl + j * nn; io.ovl = ovl + j * N;
      const size_t gx = (size_t)x * nn;
      // the coupling created at the last level joins this segment's first node and the next one's: the next pass loads it
      double* gNU = (!TOP && lam == a.m - 1 && has_b) ? ws_mat<N>(a, W_NU) + gx : nullptr;
      eliminate<PIVOT, HAS_E, HAS_Y, TOP, N>(sm, lane, io, ws_mat<N>(a, W_E) + gx, ws_mat<N>(a, W_GA) + gx, ws_mat<N>(a, W_GB) + gx,
                                             ws_vec<N>(a, V_V) + (size_t)x * N, gNU, lp);
    }
    lds_barrier();
    CHAIN_STAMP(TOP && HAS_E);
  }
  if (!TOP) {
    // ---- what crosses the segment boundary, once per pass ----
    for (int el = tid; el < nn; el += nthr) {
      ws_mat<N>(a, W_RS)[(size_t)x0 * nn + el] = sm[oRl + el];
      if (ext_right) ws_mat<N>(a, W_LS + a.par)[(size_t)xn * nn + el] = sm[oDl + S * nn + el];
    }
    if (HAS_Y && tid < N) {
      ws_vec<N>(a, V_YRS)[(size_t)x0 * N + tid] = sm[oyR + tid];
      if (ext_right) ws_vec<N>(a, V_YLS + a.par)[(size_t)xn * N + tid] = sm[oyl + S * N + tid];
    }
    if (lane == 0) {
      const int ent = a.lp_off + bid * nwaves + wave;
      ws_lp<N>(a)[ent] = lp.m;
      a.wsi[ent] = lp.e * 2 + lp.bad;
    }
    return;
  }
  if constexpr (TOP) {
    // ---- root (node 0) ----
    if (wave == 0) {
      ElimIO io;
      io.oD = oDl; io.oR = oRl; io.oUa = -1; io.oUb = -1; io.oRa = 0; io.oDb = 0; io.oNU = 0;
      io.oy = oyl; io.oyR = oyR; io.oyRa = 0; io.oyb = 0; io.oZero = oZero;
      io.oEl = oEl; io.oGAl = oGAl; io.oGBl = oGBl; io.ovl = ovl;
      eliminate<PIVOT, HAS_E, HAS_Y, true, N>(sm, lane, io, nullptr, nullptr, nullptr, nullptr, nullptr, lp);
    }
    // ---- 1/2 log det: this pass's waves leave their accumulated pivots in LDS; the LAST wave folds them with the earlier
    // passes' partial products (reduced at the start of this pass, below) and takes the one logarithm -- beside the first
    // level of the backward recursion, not in front of it
    int* redi = (int*)(sm + oRed + 32);
    if (a.hld && lane == 0) { sm[oRed + wave] = lp.m; redi[wave] = lp.e * 2 + lp.bad; }
    lds_barrier();                                     // also: the factors of these nodes are in LDS (El, GAl, GBl, vl)
    if (a.hld && wave == nwaves - 1) {
      const int* redi2 = (const int*)(sm + oRed + 96);
      const int k2 = lane & 31;
      const bool on = k2 < nwaves;
      double mv = on ? (lane < 32 ? sm[oRed + k2] : sm[oRed + 64 + k2]) : 1.0;
      int eb = on ? (lane < 32 ? redi[k2] : redi2[k2]) : 0;
      int ev = eb >> 1, bflag = eb & 1;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {               // fixed tree: product of mantissas (renormalised), sum of exponents
        const double t = mv * __shfl_xor(mv, o);
        ev += __shfl_xor(ev, o) + __builtin_amdgcn_frexp_exp(t);
        mv = __builtin_amdgcn_frexp_mant(t);
        bflag |= __shfl_xor(bflag, o);
      }
      if (lane == 0) a.hld[0] = bflag ? __builtin_nan("") : 0.5 * (log(mv) + (double)ev * 0.6931471805599453094);
    }
    // ---- backward recursion for the nodes of this pass ----
    if constexpr (HAS_Y) {                              // solve: x_e = v - GA x_a - GB x_b
      const bool wt = a.sync != nullptr;                // (merged launch: the backward workgroups beside us read x of our nodes)
      if (tid < N) { const double v = sm[ovl + tid]; sm[oxl + tid] = v; if (tid < a.n) st_out(a.x + tid, v, wt); }
      lds_barrier();
      for (int lam = a.m - 1; lam >= 0; --lam) {
        const int h2 = 1 << lam;
        const int nel = (cnt + h2 - 1) >> (lam + 1);
        for (int e = tid; e < nel * N; e += nthr) {
          const int u = e / N, r = e % N, j = (2 * u + 1) * h2, x = x0 + j * st;
          const int ja = j - h2, jb = j + h2;
          double ga[N], xv[N];
          ld_row<N>(sm + oGAl + j * nn + r * N, ga);
          ld_row<N>(sm + oxl + ja * N, xv);
          double xe = sm[ovl + j * N + r];
#pragma unroll
          for (int k = 0; k < N; ++k) xe = fma(-ga[k], xv[k], xe);
          if (jb < cnt) {
            ld_row<N>(sm + oGBl + j * nn + r * N, ga);
            ld_row<N>(sm + oxl + jb * N, xv);
#pragma unroll
            for (int k = 0; k < N; ++k) xe = fma(-ga[k], xv[k], xe);
          }
          sm[oxl + j * N + r] = xe;
          if (r < a.n) st_out(a.x + (size_t)x * a.n + r, xe, wt);
        }
        lds_barrier();
      }
      chain_signal(a.sync, a.sync_seq + a.sync_fault);
    } else if (HAS_E) {
      if (!a.need_back) return;
      const int oSg = oDl, oSL = oRl, oSLt = oCt, oSR = oNU;   // the forward arrays are dead
      CHAIN_STAMP(true);
      for (int el = tid; el < nn; el += nthr) {
        const double v = sm[oEl + el];
        const int xe = ext_el<N>(a.n, el);
        sm[oSg + el] = v;
        if (xe >= 0) st_out(a.SigD + xe, v, a.sync != nullptr);
      }
      lds_barrier();
      CHAIN_STAMP(true);
      for (int lam = a.m - 1; lam >= 0; --lam) {
        const int h2 = 1 << lam;
        const int nel = (cnt + h2 - 1) >> (lam + 1);
        for (int u = wave; u < nel; u += nwaves) {
          const int j = (2 * u + 1) * h2, x = x0 + j * st;
          const int ja = j - h2, jb = j + h2;
          const bool has_b = jb < cnt;
          const bool a_odd = has_b && (((ja / (2 * h2)) & 1) != 0);   // which of a, b was eliminated at the next level
          const int jbb = has_b ? jb : 0;
          const int oX = a_odd ? oSR + ja * nn : oSLt + jbb * nn;     // Sig_ab
          const int oY = a_odd ? oSRt + ja * nn : oSL + jbb * nn;     // Sig_ba
          const bool lvl0 = a.level0 + lam == 0;
          const size_t gx = (size_t)x * nn;
          marginal_node<N>(sm, lane, has_b, oEl + j * nn, oGAl + j * nn, oGBl + j * nn, oSg + ja * nn, oSg + jbb * nn, oX, oY,
                           oSg + j * nn, oSL + j * nn, oSLt + j * nn, oSR + j * nn, oSRt + j * nn, a.SigD + (size_t)x * (a.n * a.n),
                           ws_mat<N>(a, W_SL) + gx, ws_mat<N>(a, W_SR) + gx,
                           lvl0 ? a.SigU + (size_t)(x0 + ja * st) * (a.n * a.n) : nullptr,
                           lvl0 ? a.SigU + (size_t)x * (a.n * a.n) : nullptr, a.n, true, a.sync != nullptr);
        }
        lds_barrier();
        CHAIN_STAMP(true);
      }
      chain_signal(a.sync, a.sync_seq + a.sync_fault);
    }
  }
}

// ---- pass C: backward recursion inside every segment of an earlier pass ----
template <bool HAS_E, bool HAS_Y, int N>
__device__ __forceinline__ void backward_body(const ChainArgs& a, const int bid, double* sm) {
  constexpr int nn = N * N;
  const LazyPred lpred = pred_issue(a.pred, a.pred_val);        // checked behind the load phase, in front of the first store
  const int T = a.T, S = a.S, st = 1 << a.level0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = blockDim.x >> 6, nthr = blockDim.x;
  const int x0 = bid * S * st;
  const int cnt = min(S, (T - x0 + st - 1) >> a.level0);
  const int xn = x0 + S * st;                         // the next segment's first node (slot S)
  const bool ext_right = xn < T;
  if constexpr (HAS_Y) {
    const int oxl = 0;                                // [S+1][N]
    const int ovl = oxl + (S + 1) * N;                // [S][N]
    const int oGA = ovl + S * N;                      // [S][nn] row-major
    const int oGB = oGA + S * nn;
    for (int e = tid; e < cnt * nn; e += nthr) {
      const int j = e / nn, el = e % nn, x = x0 + j * st;
      if (j > 0) {                                    // workspace layout is transposed (column-contiguous)
        const int c = el / N, r = el % N;
        sm[oGA + j * nn + r * N + c] = ws_mat<N>(a, W_GA)[(size_t)x * nn + el];
        sm[oGB + j * nn + r * N + c] = ws_mat<N>(a, W_GB)[(size_t)x * nn + el];
      }
    }
    for (int e = tid; e < cnt * N; e += nthr) {
      const int j = e / N, r = e % N, x = x0 + j * st;
      if (j > 0) sm[ovl + e] = ws_vec<N>(a, V_V)[(size_t)x * N + r];
    }
    const bool fresh = a.sync != nullptr;               // merged launch: x of the boundary nodes comes from the top pass beside us
    bool poison = false;
    if (fresh) {
      if (pred_fail(lpred)) return;                    // (the top pass's workgroup returns on the same predicate: nothing to wait for)
      poison = !chain_wait(a.sync, a.sync_seq, (int*)(sm + oGB + S * nn));
    }
    const double nanv = __builtin_nan("");
    if (tid < N) sm[oxl + tid] = poison ? nanv : (tid < a.n ? chain_ld(a.x + (size_t)x0 * a.n + tid, fresh) : 0.0);
    if (ext_right && tid >= 64 && tid < 64 + N)
      sm[oxl + S * N + tid - 64] = poison ? nanv : (tid - 64 < a.n ? chain_ld(a.x + (size_t)xn * a.n + tid - 64, fresh) : 0.0);
    if (pred_fail(lpred)) return;                      // (block-uniform; only LDS has been written)
    __syncthreads();
    for (int lam = a.m - 1; lam >= 0; --lam) {
      const int h2 = 1 << lam;
      const int nel = (cnt + h2 - 1) >> (lam + 1);          // odd multiples of h2 below cnt
      for (int e = tid; e < nel * N; e += nthr) {
        const int u = e / N, r = e % N, j = (2 * u + 1) * h2, x = x0 + j * st;
        const int ja = j - h2, jb = j + h2;
        const bool has_b = (jb < cnt) || (jb == S && ext_right);
        double ga[N], xv[N];
        ld_row<N>(sm + oGA + j * nn + r * N, ga);
        ld_row<N>(sm + oxl + ja * N, xv);
        double xe = sm[ovl + j * N + r];
#pragma unroll
        for (int k = 0; k < N; ++k) xe = fma(-ga[k], xv[k], xe);
        if (has_b) {
          ld_row<N>(sm + oGB + j * nn + r * N, ga);
          ld_row<N>(sm + oxl + jb * N, xv);
#pragma unroll
          for (int k = 0; k < N; ++k) xe = fma(-ga[k], xv[k], xe);
        }
        sm[oxl + j * N + r] = xe;
        if (r < a.n) a.x[(size_t)x * a.n + r] = xe;
      }
      lds_barrier();
    }
  } else if constexpr (HAS_E) {
    const int oSg = 0;                                  // [S+1][nn] Sig_jj (slot S = next segment's first node)
    const int oSL = oSg + (S + 1) * nn;                 // [S+1][nn] Sig[j, left neighbour at its level]
    const int oSLt = oSL + (S + 1) * nn;                //           and its transpose
    const int oSR = oSLt + (S + 1) * nn;                // [S+1][nn] Sig[j, right neighbour]; slot 0 is preloaded with Sig[x0, xn]
    const int oSRt = oSR + (S + 1) * nn;
    const int oEl = oSRt + (S + 1) * nn;                // [S][nn] factors of the segment's nodes, row-major
    const int oGA = oEl + S * nn;
    const int oGB = oGA + S * nn;
    // Sig[x0, xn]: the two are adjacent at level level0 + m; the odd one was eliminated there
    const int lvl_up = a.level0 + a.m;
    const bool x0_odd = ext_right && (((x0 >> lvl_up) & 1) != 0);
    for (int e0 = tid; e0 < cnt * nn; e0 += 2 * nthr) {      // two elements per round: one memory round trip
      double ev[2], gav[2], gbv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * nthr;
        const int j = e / nn, el = e % nn, x = x0 + j * st;
        ev[u] = gav[u] = gbv[u] = 0.0;
        if (e < cnt * nn && j > 0) {
          const size_t g = (size_t)x * nn + el;
          ev[u] = ws_mat<N>(a, W_E)[g];
          gav[u] = ws_mat<N>(a, W_GA)[g];
          gbv[u] = ws_mat<N>(a, W_GB)[g];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * nthr;
        if (e < cnt * nn && e >= nn) {
          const int j = e / nn, el = e % nn, c = el / N, r = el % N;
          sm[oEl + e] = ev[u];                               // symmetric: either orientation
          sm[oGA + j * nn + r * N + c] = gav[u];             // workspace layout is transposed
          sm[oGB + j * nn + r * N + c] = gbv[u];
        }
      }
    }
    const bool fresh = a.sync != nullptr;               // merged launch: Sig of the boundary nodes comes from the top pass beside us
    bool poison = false;
    if (fresh) {
      if (pred_fail(lpred)) return;                    // (the top pass's workgroup returns on the same predicate: nothing to wait for)
      poison = !chain_wait(a.sync, a.sync_seq, (int*)(sm + oGB + S * nn));
    }
    for (int el = tid; el < nn; el += nthr) {
      const int r = el / N, c = el % N;
      const int xe = ext_el<N>(a.n, el);
      const double pad = r == c ? 1.0 : 0.0;
      const double nanv = __builtin_nan("");
      sm[oSg + el] = poison ? nanv : (xe >= 0 ? chain_ld(a.SigD + (size_t)x0 * (a.n * a.n) + xe, fresh) : pad);
      if (ext_right) {
        sm[oSg + S * nn + el] = poison ? nanv : (xe >= 0 ? chain_ld(a.SigD + (size_t)xn * (a.n * a.n) + xe, fresh) : pad);
        const double v = x0_odd ? chain_ld(ws_mat<N>(a, W_SR) + (size_t)x0 * nn + el, fresh)
                                : chain_ld(ws_mat<N>(a, W_SL) + (size_t)xn * nn + c * N + r, fresh);
        sm[oSR + el] = v;                                    // Sig[x0, xn] [r][c]
        sm[oSRt + c * N + r] = v;
      }
    }
    if (pred_fail(lpred)) return;                      // (block-uniform; only LDS has been written)
    __syncthreads();
    for (int lam = a.m - 1; lam >= 0; --lam) {
      const int h2 = 1 << lam;
      const int nel = (cnt + h2 - 1) >> (lam + 1);          // odd multiples of h2 below cnt
      for (int u = wave; u < nel; u += nwaves) {
        const int j = (2 * u + 1) * h2, x = x0 + j * st;
        const int ja = j - h2, jb = j + h2;
        const bool has_b = (jb < cnt) || (jb == S && ext_right);
        // Sig[a,b]: the segment boundary pair was preloaded into slot 0 of SR / SRt; otherwise the odd one of (a, b) at the
        // next local level holds it (SR[ja] = Sig[a,b] or SL[jb] = Sig[b,a])
        bool a_odd = true;
        int ia = 0;
        if (has_b && !(ja == 0 && jb == S)) { a_odd = ((ja / (2 * h2)) & 1) != 0; ia = ja; }
        const int jbb = has_b ? jb : 0;
        const int oX = a_odd ? oSR + ia * nn : oSLt + jbb * nn;
        const int oY = a_odd ? oSRt + ia * nn : oSL + jbb * nn;
        const bool lvl0 = a.level0 + lam == 0;
        const bool keep = a.level0 > 0;                      // a lower pass will preload Sig[x0, xn] from the workspace
        const size_t gx = (size_t)x * nn;
        marginal_node<N>(sm, lane, has_b, oEl + j * nn, oGA + j * nn, oGB + j * nn, oSg + ja * nn, oSg + jbb * nn, oX, oY, oSg + j * nn,
                         oSL + j * nn, oSLt + j * nn, oSR + j * nn, oSRt + j * nn, a.SigD + (size_t)x * (a.n * a.n),
                         keep ? ws_mat<N>(a, W_SL) + gx : nullptr, keep ? ws_mat<N>(a, W_SR) + gx : nullptr,
                         lvl0 ? a.SigU + (size_t)(x0 + ja * st) * (a.n * a.n) : nullptr,
                         lvl0 ? a.SigU + (size_t)x * (a.n * a.n) : nullptr, a.n);
      }
      lds_barrier();
    }
  }
}

}  // namespace chain

// Blocks [0, nb0): factorisation a0 (unpivoted; log-det, selected inverse); the rest: pivoted solve a1.

template <int N, bool TOP>
__global__ __launch_bounds__(chain_threads(N)) void chain_forward_kernel(ChainArgs a0, ChainArgs a1, int nb0, AsmList AL) {
  extern __shared__ double sm[];
  constexpr int KA_LINES = (2 * sizeof(ChainArgs) + 8 + sizeof(AsmList) + 63) / 64;
  static_assert(KA_LINES == 13, "kernarg_warm: one specialisation per argument block size");
  kernarg_warm<KA_LINES>();
  if ((int)blockIdx.x < nb0) chain::forward_body<false, true, false, TOP, N>(a0, AL, (int)blockIdx.x, sm);    // (predicate: inside)
  else chain::forward_body<true, false, true, TOP, N>(a1, AL, (int)blockIdx.x - nb0, sm);
}

template <int N>
__global__ __launch_bounds__(chain_threads(N)) void chain_backward_kernel(ChainArgs a0, ChainArgs a1, int nb0) {
  extern __shared__ double sm[];
  constexpr int KA_LINES = (2 * sizeof(ChainArgs) + 4 + 63) / 64;
  static_assert(KA_LINES == 7, "kernarg_warm: one specialisation per argument block size");
  kernarg_warm<KA_LINES>();
  if ((int)blockIdx.x < nb0) chain::backward_body<true, false, N>(a0, (int)blockIdx.x, sm);      // (predicate: inside)
  else chain::backward_body<false, true, N>(a1, (int)blockIdx.x - nb0, sm);
}

// The top pass and the backward recursion of the last segmented pass in ONE launch: blocks [0, nbt) are the top pass (as in
// chain_forward_kernel<N, true>), the blocks behind them pass C's (as in chain_backward_kernel) with the pass parameters cp.
// The backward workgroups load their segment's factors (written by the launch before) while the top pass runs, and wait for
// its word (a0.sync / a1.sync) only in front of the boundary values: a kernel boundary and pass C's load phase leave the
// critical path for one hand-over.
template <int N>
__global__ __launch_bounds__(chain_threads(N)) void chain_top_back_kernel(ChainArgs a0, ChainArgs a1, int nb0, AsmList AL, ChainPassDev cp,
                                                                           int nbt, int nb0c) {
  extern __shared__ double sm[];
  constexpr int KA_LINES = (2 * sizeof(ChainArgs) + 8 + sizeof(AsmList) + sizeof(ChainPassDev) + 8 + 63) / 64;
  static_assert(KA_LINES == 14, "kernarg_warm: one specialisation per argument block size");
  kernarg_warm<KA_LINES>();
  const int b = (int)blockIdx.x;
  if (b < nbt) {
    if (b < nb0) chain::forward_body<false, true, false, true, N>(a0, AL, b, sm);
    else chain::forward_body<true, false, true, true, N>(a1, AL, b - nb0, sm);
    return;
  }
  const int bc = b - nbt;
  ChainArgs c = bc < nb0c ? a0 : a1;
  c.level0 = cp.level0; c.m = cp.m; c.S = cp.S; c.first = cp.first; c.par = cp.par; c.lp_off = cp.lp_off;
  if (bc < nb0c) chain::backward_body<true, false, N>(c, bc, sm);
  else chain::backward_body<false, true, N>(c, bc - nb0c, sm);
}

}  // namespace gvi
