// Block-tridiagonal chain operations for SHORT chains of 2 x 2 blocks: T <= 65, n <= 2 (BASELINE configs[1]: 65 states of
// size 2).  Same cyclic-reduction tree, same arithmetic per node (Gauss-Jordan with the same operation order, the same Schur
// products, the same two accumulators per node, the same Takahashi recursion) as kernels_chain.hpp -- but laid out for a
// chain whose whole problem is 130 unknowns:
//
//   lane = NODE, ONE wave per operation.  Every node's block, coupling, right-hand side and (later) factors live in the
//   registers of its lane; an eliminated node does its 2 x 2 Gauss-Jordan on all nine columns of [D | I | Ua^T | Ub | y] in
//   its own lane and forms the updates of both neighbours there; the neighbours fetch them with lane shuffles
//   (ds_bpermute: no LDS memory, no barrier, no cross-wave hand-over).  A level of the generic kernel at this size is
//   ~0.85 us of barrier + LDS round trips + address arithmetic around 30 flops; here it is the shuffles and the flops.
//   Node 64 (T = 65) rides in a second register set of lane 0: node 0 has no left neighbour and node 64 no right one, and a
//   shuffle "from lane - h" wraps to lane 64 - h, which is exactly node 64's left neighbour at spacing h.
//
// Launch: blocks [0, nb0) of 64 threads = factorisation a0 (unpivoted, log-det, selected inverse when need_back), the rest the
// pivoted solve a1; assemble-on-load, the fused trial precision (mix) and predicates as in chain_forward_kernel.
#pragma once
#include "kernels_chain.hpp"

namespace gvi {
namespace chain_wave {

constexpr int WN = 2;            // compiled block size (n = 1 is padded by the identity)
constexpr int WT_MAX = 65;

__device__ __forceinline__ double shf(const double v, const int src) { return __shfl(v, src); }   // source lane mod 64

// the nine columns of the elimination tile, col[j][r]: j = 0, 1: D ; 2, 3: I ; 4, 5: Ua^T ; 6, 7: Ub ; 8: y
struct Tile { double c[9][2]; };

// Gauss-Jordan of eliminate<> (kernels_chain.hpp) on a 2 x 2 pivot block with every column in this lane
template <bool PIVOT>
__device__ __forceinline__ void gauss_jordan(Tile& t, double (&pivs)[2], int& bad) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (PIVOT && p == 0) {                     // threshold partial pivoting (rows swapped only when |natural pivot| x 8 < column max)
      const bool sw = fabs(t.c[0][1]) > fabs(t.c[0][0]) && !(fabs(t.c[0][0]) * 8.0 >= fabs(t.c[0][1]));
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const double a0 = t.c[j][0], a1 = t.c[j][1];
        t.c[j][0] = sw ? a1 : a0;
        t.c[j][1] = sw ? a0 : a1;
      }
    }
    const double ap0 = t.c[p][0], ap1 = t.c[p][1];
    const double piv = p == 0 ? ap0 : ap1;
    if (!(piv > 0.0)) bad = 1;
    pivs[p] = piv;
    double ip = __builtin_amdgcn_rcp(piv);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const double f = t.c[j][p] * ip;
      if (p == 0) t.c[j][1] = fma(-ap1, f, t.c[j][1]);
      else t.c[j][0] = fma(-ap0, f, t.c[j][0]);
      t.c[j][p] = f;
    }
  }
}

struct Node {                    // forward state of a node: block as columns (Dc[c * 2 + r] = D[c][r]: symmetric), two accumulators
  double DL[4], DR[4], yl[2], yR[2];
};

// asm_element (kernels_chain.hpp) for ALL elements of node t at once: per factor set (outer, runtime) every load of the node's
// V_D / V_U / g elements is issued before the first add -- asm_element's loop over the sets waits for its loads inside the
// loop, and a lane that assembles ten elements one call after the other pays twenty dependent round trips (measured: the
// launch took 17 us inside the iteration against 10 us stand-alone).  Same sums in the same order: per element, per set
// s = [own factor] + [left neighbour's factor], acc += s.
struct AsmAcc { double D[4], U[4], g[2]; };

// the loads of one set for one node (nothing is added to the running sums here)
__device__ __forceinline__ void asm_loads(const AsmSet& a, const int n, const int t, const bool on, const bool hasc,
                                          double (&d0)[4], double (&d1)[4], double (&u0)[4], double (&g0)[2], double (&g1)[2]) {
  const int d = a.d;
  const bool two = d == 2 * n;
#pragma unroll
  for (int i = 0; i < 4; ++i) d0[i] = d1[i] = u0[i] = 0.0;
  g0[0] = g0[1] = g1[0] = g1[1] = 0.0;
  if (!on) return;
  if (a.nsp > 0) {                                   // sparse unary set: ascending k (the order of the CSR lists)
#pragma unroll
    for (int k = 0; k < ASM_SPARSE_MAX; ++k) {
      if (k < a.nsp && a.sp[k] == t) {
#pragma unroll
        for (int el = 0; el < 4; ++el) {
          const int r = el >> 1, c = el & 1;
          if (r < n && c < n) d0[el] += a.Vddmu[(size_t)k * d * d + r * d + c];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (r < n) g0[r] += a.Vdmu[(size_t)k * d + r];
      }
    }
  } else {
    const bool own = t < a.K, left = two && t > 0 && t - 1 < a.K, up = two && t < a.K && hasc;
#pragma unroll
    for (int el = 0; el < 4; ++el) {
      const int r = el >> 1, c = el & 1;
      if (r < n && c < n) {
        if (own) d0[el] = a.Vddmu[(size_t)t * d * d + r * d + c];
        if (left) d1[el] = a.Vddmu[(size_t)(t - 1) * d * d + (n + r) * d + n + c];
        if (up) u0[el] = a.Vddmu[(size_t)t * d * d + r * d + n + c];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r < n) {
        if (own) g0[r] = a.Vdmu[(size_t)t * d + r];
        if (left) g1[r] = a.Vdmu[(size_t)(t - 1) * d + n + r];
      }
    }
  }
}

// both nodes of a lane (its own and, in lane 0 at T = 65, node 64) in one sweep over the sets
__device__ __forceinline__ void asm_node(const AsmList& L, const int n, const int t, const bool on, const bool hasc, const int t2,
                                         const bool on2, AsmAcc& A, AsmAcc& B) {
#pragma unroll
  for (int i = 0; i < 4; ++i) A.D[i] = A.U[i] = B.D[i] = B.U[i] = 0.0;
  A.g[0] = A.g[1] = B.g[0] = B.g[1] = 0.0;
  for (int si = 0; si < L.nsets; si += 2) {           // two sets per round: their loads are in flight together
    const bool second = si + 1 < L.nsets;
    const AsmSet& a = L.s[si];
    const AsmSet& b = L.s[second ? si + 1 : si];
    double d0[4], d1[4], u0[4], g0[2], g1[2], e0[4], e1[4], w0[4], h0[2], h1[2];
    double p0[4], p1[4], q0[4], r0[2], r1[2], s0[4], s1[4], x0[4], y0[2], y1[2];
    asm_loads(a, n, t, on, hasc, d0, d1, u0, g0, g1);
    asm_loads(a, n, t2, on2, false, e0, e1, w0, h0, h1);
    asm_loads(b, n, t, on && second, hasc, p0, p1, q0, r0, r1);
    asm_loads(b, n, t2, on2 && second, false, s0, s1, x0, y0, y1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A.D[i] += (0.0 + d0[i]) + d1[i]; A.U[i] += 0.0 + u0[i];
      B.D[i] += (0.0 + e0[i]) + e1[i]; B.U[i] += 0.0 + w0[i];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) { A.g[r] += (0.0 + g0[r]) + g1[r]; B.g[r] += (0.0 + h0[r]) + h1[r]; }
    if (second) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        A.D[i] += (0.0 + p0[i]) + p1[i]; A.U[i] += 0.0 + q0[i];
        B.D[i] += (0.0 + s0[i]) + s1[i]; B.U[i] += 0.0 + x0[i];
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) { A.g[r] += (0.0 + r0[r]) + r1[r]; B.g[r] += (0.0 + y0[r]) + y1[r]; }
    }
  }
}

// element (r, c) of node t's blocks from the caller's arrays: the first-pass load of forward_body (mix / assemble-on-load).
// Loads only; what the pass leaves in memory on the way (trial precision, assembled matrix and gradient) is written by
// store_side AFTER every load of the lane has been issued -- a store between two elements' loads orders them (the store needs
// its value), and ten elements per lane then cost ten memory round trips (measured: 16.7 instead of 10.3 us per launch).
// the caller's own arrays first (requested before the assemble's sweep over the sets): D, U and the mixed-in matrix
struct RawBlock { double D[4], U[4], mD[4], mU[4], y[2]; };
template <bool HAS_Y>
__device__ __forceinline__ void raw_loads(const ChainArgs& a, const int t, const bool on, const bool hasc, const bool mix, RawBlock& R) {
  const int n = a.n, T = a.T;
  const bool asmv = a.asm_on != 0;
#pragma unroll
  for (int el = 0; el < 4; ++el) {
    const int r = el >> 1, c = el & 1;
    R.D[el] = R.U[el] = R.mD[el] = R.mU[el] = 0.0;
    if (on && r < n && c < n) {
      const int xe = r * n + c;
      const size_t ge = (size_t)t * (n * n) + xe, gu = (size_t)(T + t) * (n * n) + xe;
      if (!(HAS_Y && asmv)) {
        R.D[el] = a.D[ge];
        if (mix && !asmv) R.mD[el] = a.mixV[ge];
        if (hasc) {
          R.U[el] = a.U[ge];
          if (mix && !asmv) R.mU[el] = a.mixV[gu];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) R.y[r] = (HAS_Y && on && r < n && !asmv) ? a.rhs[(size_t)t * n + r] : 0.0;
}

template <bool HAS_Y>
__device__ __forceinline__ void load_block(const ChainArgs& a, const RawBlock& R, const double (&aD)[4], const double (&aU)[4], const bool on,
                                           const bool hasc, const bool mix, double (&D)[4], double (&C)[4]) {
  const int n = a.n;
#pragma unroll
  for (int el = 0; el < 4; ++el) {
    const int r = el >> 1, c = el & 1;
    double dv = r == c ? 1.0 : 0.0, cu = 0.0;            // identity padding / absent node
    const bool in = on && r < n && c < n;
    if (in) {
      const bool asmv = a.asm_on != 0;
      const double vD = aD[el], vU = aU[el];            // (zero when nothing is assembled)
      if (HAS_Y && asmv) {
        dv = vD;
        if (hasc) cu = vU;
      } else {
        dv = R.D[el];
        if (mix) { const double mv = asmv ? vD : R.mD[el]; dv = dv + a.mix_step * (mv - dv); }
        if (hasc) {
          cu = R.U[el];
          if (mix) { const double mv = asmv ? vU : R.mU[el]; cu = cu + a.mix_step * (mv - cu); }
        }
      }
    }
    D[el] = dv;                  // D[r * 2 + c] = D[r][c] = column r by symmetry (the generic kernel reads row cc as column cc)
    C[el] = hasc ? cu : 0.0;     // C[r * 2 + c] = A[t, t + 1][r][c]
  }
}

template <bool HAS_Y>
__device__ __forceinline__ void load_rhs(const ChainArgs& a, const RawBlock& R, const double (&ag)[2], const bool on, double (&y)[2], double (&g)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double v = 0.0, gv = 0.0;
    if (HAS_Y && on && r < a.n) {
      if (a.asm_on) {
        gv = ag[r];
        v = a.rhs_scale * gv;
      } else v = a.rhs_scale * R.y[r];
    }
    y[r] = v;
    g[r] = gv;
  }
}

template <bool HAS_Y>
__device__ __forceinline__ void store_side(const ChainArgs& a, const int t, const bool on, const bool hasc, const bool mix,
                                           const double (&D)[4], const double (&C)[4], const double (&g)[2]) {
  const int n = a.n, T = a.T;
  if (!on) return;
#pragma unroll
  for (int el = 0; el < 4; ++el) {
    const int r = el >> 1, c = el & 1;
    if (r < n && c < n) {
      const int xe = r * n + c;
      const size_t ge = (size_t)t * (n * n) + xe, gu = (size_t)(T + t) * (n * n) + xe;
      if (HAS_Y && a.asm_on) {
        a.asmD[ge] = D[el];
        if (hasc) a.asmU[ge] = C[el];
      }
      if (mix) {
        a.mixOut[ge] = D[el];
        if (hasc) a.mixOut[gu] = C[el];
      }
    }
  }
  if (HAS_Y && a.asm_on) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (r < n) a.asmG[(size_t)t * n + r] = g[r];
  }
}

// elimination of one node inside its lane.  in: block (DL + DR), couplings Ua = A[a, e] (row-major), Ub = A[e, b], rhs;
// out: factors E / GA / GB (row-major) / v and the neighbours' updates
struct Elim {
  double E[4], GA[4], GB[4], v[2];
  double tA[4];      // column c of the update of D_a at [c * 2 + r]
  double nu[4];      // new coupling A[a, b], row-major
  double tB[4];      // column c of the update of D_b at [c * 2 + r]
  double ya[2], yb[2];
  double pivs[2];
  int bad;
};

template <bool PIVOT, bool HAS_E, bool HAS_Y>
__device__ __forceinline__ void eliminate_node(const double (&Dc)[4], const double (&Ua)[4], const double (&Ub)[4], const double (&y)[2], Elim& o) {
  Tile t;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      t.c[c][r] = Dc[c * 2 + r];
      t.c[2 + c][r] = r == c ? 1.0 : 0.0;
      t.c[4 + c][r] = Ua[c * 2 + r];         // (Ua^T)[r][c] = Ua[c][r]
      t.c[6 + c][r] = Ub[r * 2 + c];         // Ub[r][c]
    }
  t.c[8][0] = y[0]; t.c[8][1] = y[1];
  o.bad = 0;
  gauss_jordan<PIVOT>(t, o.pivs, o.bad);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      o.E[r * 2 + c] = t.c[2 + c][r];
      o.GA[r * 2 + c] = t.c[4 + c][r];
      o.GB[r * 2 + c] = t.c[6 + c][r];
    }
  o.v[0] = t.c[8][0]; o.v[1] = t.c[8][1];
  // Schur products of eliminate<>: t = -Ua col (k ascending from 0.0), s = -Ub^T col
#pragma unroll
  for (int j = 4; j < 9; ++j) {
    double tt[2], ss[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) acc = fma(-Ua[r * 2 + k], t.c[j][k], acc);
      tt[r] = acc;
      double acs = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) acs = fma(-Ub[k * 2 + r], t.c[j][k], acs);
      ss[r] = acs;
    }
    if (j < 6) { o.tA[(j - 4) * 2 + 0] = tt[0]; o.tA[(j - 4) * 2 + 1] = tt[1]; }
    else if (j < 8) {
      o.nu[0 * 2 + (j - 6)] = tt[0]; o.nu[1 * 2 + (j - 6)] = tt[1];
      o.tB[(j - 6) * 2 + 0] = ss[0]; o.tB[(j - 6) * 2 + 1] = ss[1];
    } else { o.ya[0] = tt[0]; o.ya[1] = tt[1]; o.yb[0] = ss[0]; o.yb[1] = ss[1]; }
  }
}

// accumulated log-pivots of a lane (LogPiv of kernels_chain.hpp)
__device__ __forceinline__ void lp_add(double& m, int& e, const double (&pivs)[2]) {
  const double mp = __builtin_amdgcn_frexp_mant(pivs[0]) * __builtin_amdgcn_frexp_mant(pivs[1]);
  const int es = __builtin_amdgcn_frexp_exp(pivs[0]) + __builtin_amdgcn_frexp_exp(pivs[1]);
  const double t = m * mp;
  m = __builtin_amdgcn_frexp_mant(t);
  e += es + __builtin_amdgcn_frexp_exp(t);
}

// marginal_node (kernels_chain.hpp) for one node in its lane: SL = Sig[e, a], SR = Sig[e, b], See
__device__ __forceinline__ void marginals(const bool has_b, const double (&E)[4], const double (&GA)[4], const double (&GB)[4],
                                          const double (&Saa)[4], const double (&Sbb)[4], const double (&X)[4], const double (&Y)[4],
                                          double (&SL)[4], double (&SR)[4], double (&See)[4]) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double sl = 0.0, sr = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) sl = fma(GA[r * 2 + k], Saa[c * 2 + k], sl);
      if (has_b) {
#pragma unroll
        for (int k = 0; k < 2; ++k) sl = fma(GB[r * 2 + k], X[c * 2 + k], sl);
#pragma unroll
        for (int k = 0; k < 2; ++k) sr = fma(GA[r * 2 + k], Y[c * 2 + k], sr);
#pragma unroll
        for (int k = 0; k < 2; ++k) sr = fma(GB[r * 2 + k], Sbb[c * 2 + k], sr);
      }
      SL[r * 2 + c] = -sl;
      SR[r * 2 + c] = -sr;
    }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double see = E[r * 2 + c];
#pragma unroll
      for (int k = 0; k < 2; ++k) see = fma(-SL[r * 2 + k], GA[c * 2 + k], see);
      if (has_b) {
#pragma unroll
        for (int k = 0; k < 2; ++k) see = fma(-SR[r * 2 + k], GB[c * 2 + k], see);
      }
      See[r * 2 + c] = see;
    }
}

__device__ __forceinline__ void store_block(double* g, const int n, const double (&M)[4], const bool transposed) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (r < n && c < n) g[transposed ? c * n + r : r * n + c] = M[r * 2 + c];
}

template <bool PIVOT, bool HAS_E, bool HAS_Y>
__device__ __forceinline__ void body(const ChainArgs& a, const AsmList& AL) {
  const LazyPred lpred = pred_issue(a.pred, a.pred_val);     // checked behind the loads, in front of the first store
  const int lane = threadIdx.x & 63, T = a.T, n = a.n;
  const bool two = T == WT_MAX;                      // node 64 lives in lane 0's second register set
  const bool mix = a.mixV != nullptr || (a.asm_on && !HAS_Y);
  const bool on = lane < T;
  // ---- load ----
  Node nd;
  double C[4];                                       // coupling to the next alive node on the right, row-major
  double g1[2], g2[2];
  AsmAcc A1, A2;                                     // assemble-on-load: V_D / V_U / g of the lane's node and of node 64
  const bool hc = on && lane + 1 < T, on2 = two && lane == 0;
  RawBlock R1, R2;
  raw_loads<HAS_Y>(a, lane, on, hc, mix, R1);
  raw_loads<HAS_Y>(a, 64, on2, false, mix, R2);
  if (a.asm_on) asm_node(AL, n, lane, on, hc, 64, on2, A1, A2);
  else {
#pragma unroll
    for (int i = 0; i < 4; ++i) A1.D[i] = A1.U[i] = A2.D[i] = A2.U[i] = 0.0;
    A1.g[0] = A1.g[1] = A2.g[0] = A2.g[1] = 0.0;
  }
  load_block<HAS_Y>(a, R1, A1.D, A1.U, on, hc, mix, nd.DL, C);
  load_rhs<HAS_Y>(a, R1, A1.g, on, nd.yl, g1);
  double DL2[4], C2[4], yl2[2];                      // node 64
  load_block<HAS_Y>(a, R2, A2.D, A2.U, on2, false, mix, DL2, C2);
  load_rhs<HAS_Y>(a, R2, A2.g, on2, yl2, g2);
  if (pred_fail(lpred)) return;
  store_side<HAS_Y>(a, lane, on, on && lane + 1 < T, mix, nd.DL, C, g1);
  store_side<HAS_Y>(a, 64, two && lane == 0, false, mix, DL2, C2, g2);
#pragma unroll
  for (int i = 0; i < 4; ++i) nd.DR[i] = 0.0;
  nd.yR[0] = nd.yR[1] = 0.0;
  // factors of this lane's node (each node is eliminated exactly once) and of node 64
  double fE[4] = {1.0, 0.0, 0.0, 1.0}, fGA[4] = {0, 0, 0, 0}, fGB[4] = {0, 0, 0, 0}, fv[2] = {0, 0};
  double lpm = 1.0;
  int lpe = 0, bad = 0;
  const double ident[4] = {1.0, 0.0, 0.0, 1.0}, zero4[4] = {0.0, 0.0, 0.0, 0.0};
  const int tl = T < 64 ? T : 64;                    // nodes that sit in lanes
  // ---- forward levels inside the lanes ----
  for (int h = 1; h < tl; h <<= 1) {
    const bool elim = on && (lane & (2 * h - 1)) == h;
    const bool has_b = elim && lane + h < T;         // (b = 64 is node 64 when T = 65)
    double Ua[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Ua[i] = shf(C[i], lane - h);          // A[a, e] lives with a = e - h
    double Dc[4], yy[2], Ub[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Dc[i] = elim ? nd.DL[i] + nd.DR[i] : ident[i];
      Ub[i] = has_b ? C[i] : 0.0;
      Ua[i] = elim ? Ua[i] : 0.0;
    }
    yy[0] = elim ? nd.yl[0] + nd.yR[0] : 0.0;
    yy[1] = elim ? nd.yl[1] + nd.yR[1] : 0.0;
    Elim o;
    eliminate_node<PIVOT, HAS_E, HAS_Y>(Dc, Ua, Ub, yy, o);
    if (elim) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fE[i] = o.E[i]; fGA[i] = o.GA[i]; fGB[i] = o.GB[i]; }
      fv[0] = o.v[0]; fv[1] = o.v[1];
      lp_add(lpm, lpe, o.pivs);
      bad |= o.bad;
    }
    // survivors: updates from the eliminated neighbour on the right (lane + h) and on the left (lane - h, wrapping for node 64)
    const bool surv = on && (lane & (2 * h - 1)) == 0;
    const bool from_r = surv && lane + h < T;
    const bool r_has_b = lane + 2 * h < T;           // the eliminated neighbour had a right neighbour: a new coupling exists
    const bool from_l = surv && lane >= 2 * h;       // lanes >= 2h: their left neighbour at spacing h was eliminated
    const bool from_l64 = two && lane == 0;          // node 64: e = 64 - h
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double ta = shf(o.tA[i], lane + h), nu = shf(o.nu[i], lane + h), tb = shf(o.tB[i], lane - h);
      if (from_r) { nd.DR[i] += ta; C[i] = r_has_b ? nu : 0.0; }
      if (from_l) nd.DL[i] += tb;
      if (from_l64) DL2[i] += tb;
    }
    if (HAS_Y) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double ya = shf(o.ya[r], lane + h), yb = shf(o.yb[r], lane - h);
        if (from_r) nd.yR[r] += ya;
        if (from_l) nd.yl[r] += yb;
        if (from_l64) yl2[r] += yb;
      }
    }
  }
  // ---- node 64 (T = 65): a = node 0 in the same lane, no b ----
  double fE2[4] = {1.0, 0.0, 0.0, 1.0}, fGA2[4] = {0, 0, 0, 0}, fv2[2] = {0, 0};
  if (two) {
    const bool me = lane == 0;
    double Dc[4], Ua[4], yy[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { Dc[i] = me ? DL2[i] + 0.0 : ident[i]; Ua[i] = me ? C[i] : 0.0; }
    yy[0] = me ? yl2[0] + 0.0 : 0.0; yy[1] = me ? yl2[1] + 0.0 : 0.0;
    Elim o;
    eliminate_node<PIVOT, HAS_E, HAS_Y>(Dc, Ua, zero4, yy, o);
    if (me) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fE2[i] = o.E[i]; fGA2[i] = o.GA[i]; nd.DR[i] += o.tA[i]; }
      fv2[0] = o.v[0]; fv2[1] = o.v[1];
      if (HAS_Y) { nd.yR[0] += o.ya[0]; nd.yR[1] += o.ya[1]; }
      lp_add(lpm, lpe, o.pivs);
      bad |= o.bad;
    }
  }
  // ---- root: node 0 ----
  {
    const bool me = lane == 0;
    double Dc[4], yy[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) Dc[i] = me ? nd.DL[i] + nd.DR[i] : ident[i];
    yy[0] = me ? nd.yl[0] + nd.yR[0] : 0.0; yy[1] = me ? nd.yl[1] + nd.yR[1] : 0.0;
    Elim o;
    eliminate_node<PIVOT, HAS_E, HAS_Y>(Dc, zero4, zero4, yy, o);
    if (me) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fE[i] = o.E[i];
      fv[0] = o.v[0]; fv[1] = o.v[1];
      lp_add(lpm, lpe, o.pivs);
      bad |= o.bad;
    }
  }
  // ---- 1/2 log det: fixed butterfly over the lanes ----
  if (HAS_E && a.hld) {
    double mv = lpm;
    int ev = lpe, bf = bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double t = mv * __shfl_xor(mv, o);
      ev += __shfl_xor(ev, o) + __builtin_amdgcn_frexp_exp(t);
      mv = __builtin_amdgcn_frexp_mant(t);
      bf |= __shfl_xor(bf, o);
    }
    if (lane == 0) a.hld[0] = bf ? __builtin_nan("") : 0.5 * (log(mv) + (double)ev * 0.6931471805599453094);
  }
  if (HAS_E && !a.need_back) return;
  // ---- backward: x_e = v - GA x_a - GB x_b ; Takahashi recursion for Sig[e, e], Sig[e, a], Sig[e, b] ----
  double x[2] = {fv[0], fv[1]}, S[4], SL[4] = {0, 0, 0, 0}, SR[4] = {0, 0, 0, 0};
  double x2[2] = {0, 0}, S2[4] = {0, 0, 0, 0}, SL2[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) S[i] = fE[i];                         // root: Sig[0, 0] = E_0 (other lanes: overwritten at their level)
  if (lane == 0) {
    if (HAS_Y) { if (n > 0) a.x[0] = x[0]; if (n > 1) a.x[1] = x[1]; }
    if (HAS_E) store_block(a.SigD, n, S, false);
  }
  if (two) {                                                        // node 64 from node 0
    if (HAS_Y) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double xe = fv2[r];
#pragma unroll
        for (int k = 0; k < 2; ++k) xe = fma(-fGA2[r * 2 + k], x[k], xe);
        x2[r] = xe;
      }
      if (lane == 0) { a.x[(size_t)64 * n] = x2[0]; if (n > 1) a.x[(size_t)64 * n + 1] = x2[1]; }
    }
    if (HAS_E) {
      double dumR[4];
      marginals(false, fE2, fGA2, zero4, S, zero4, zero4, zero4, SL2, dumR, S2);
      if (lane == 0) store_block(a.SigD + (size_t)64 * n * n, n, S2, false);
    }
  }
  int hmax = 1;
  while (hmax * 2 < tl) hmax *= 2;
  for (int h = tl > 1 ? hmax : 0; h >= 1; h >>= 1) {
    const bool elim = on && (lane & (2 * h - 1)) == h;
    const bool has_b = elim && lane + h < T;
    // what a lane offers as somebody's LEFT neighbour (its own node) and as somebody's RIGHT neighbour (lane 0: node 64)
    const bool l0 = lane == 0;
    if (HAS_Y) {
      double xa[2], xb[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        xa[r] = shf(x[r], lane - h);
        xb[r] = shf(l0 ? x2[r] : x[r], lane + h);
      }
      if (elim) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double xe = fv[r];
#pragma unroll
          for (int k = 0; k < 2; ++k) xe = fma(-fGA[r * 2 + k], xa[k], xe);
          if (has_b) {
#pragma unroll
            for (int k = 0; k < 2; ++k) xe = fma(-fGB[r * 2 + k], xb[k], xe);
          }
          x[r] = xe;
        }
        a.x[(size_t)lane * n] = x[0];
        if (n > 1) a.x[(size_t)lane * n + 1] = x[1];
      }
    }
    if (HAS_E) {
      double Saa[4], SRa[4], Sbb[4], SLb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Saa[i] = shf(S[i], lane - h);
        SRa[i] = shf(SR[i], lane - h);
        Sbb[i] = shf(l0 ? S2[i] : S[i], lane + h);
        SLb[i] = shf(l0 ? SL2[i] : SL[i], lane + h);
      }
      // Sig[a, b]: a was eliminated at the next level with b as its right neighbour (then it is a's SR), or b with a as
      // its left neighbour (b's SL, transposed)
      const bool a_odd = has_b && ((((lane - h) / (2 * h)) & 1) != 0);
      double X[4], Y[4];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          X[r * 2 + c] = a_odd ? SRa[r * 2 + c] : SLb[c * 2 + r];     // Sig_ab
          Y[r * 2 + c] = a_odd ? SRa[c * 2 + r] : SLb[r * 2 + c];     // Sig_ba
        }
      double nSL[4], nSR[4], nS[4];
      marginals(has_b, fE, fGA, fGB, Saa, Sbb, X, Y, nSL, nSR, nS);
      if (elim) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { SL[i] = nSL[i]; SR[i] = has_b ? nSR[i] : 0.0; S[i] = nS[i]; }
        store_block(a.SigD + (size_t)lane * n * n, n, S, false);
        if (h == 1) {
          store_block(a.SigU + (size_t)(lane - 1) * n * n, n, SL, true);      // Sig[a, e] = Sig[e, a]^T
          if (has_b) store_block(a.SigU + (size_t)lane * n * n, n, SR, false);
        }
      }
    }
  }
}

}  // namespace chain_wave

__global__ __launch_bounds__(64) void chain_wave_kernel(ChainArgs a0, ChainArgs a1, int nb0, AsmList AL) {
  if ((int)blockIdx.x < nb0) chain_wave::body<false, true, false>(a0, AL);      // (predicate: inside)
  else chain_wave::body<true, false, true>(a1, AL);
}

}  // namespace gvi
