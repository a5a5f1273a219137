// factor_fused_kernel: gather -> per-pass products -> sign-orbit psi walk -> chunk sum -> cost (+ publish tail) ->
// back-transform, the full moments pass of the resident NGD iteration in a single launch.
//
// Replaces, for the sum-of-squares factor sets of a chain, the three launches prep_all_kernel -> moments_orbit_pair_kernel ->
// epilogue_all_kernel (kernels_factor.hpp / kernels_orbit.hpp; reference: GVIFactorizedBase::update_*_from_joint,
// gvibase/GVIFactorizedBase.h:104-122; updateGH + calculate_partial_V, ngd/NGDFactorizedBaseGH.h:50-88).  The device
// functions are the SAME ones those kernels call (prep_body_d, orbit_wave, epilogue_body_p, epi_tail_arrive): results are
// bit-identical to the three-launch route with four chunks per factor.
//
// Block b owns a few ITEMS (factors): item b of every set, plus the items b + nblk, b + 2 nblk, ... where a set is longer
// than the grid (the chain pattern: prior b and unary factor b; block 0 also takes the 1025th unary factor) -- at most 4.
//   phase 1   ONE wave per item forms its per-pass products (gather of (mu_k, Sigma_k) from the chain, Cholesky route: ~4 us
//             of dependent work per item, the items side by side) and leaves S^-T, H, u0 in LDS.  Which wave takes which item
//             is read from the hardware SIMD ids, so that the d = 12 items of a CU's four resident blocks sit on four SIMDs;
//   phase 2   all four waves walk the orbit table for one item after the other, wave w = chunk w; the chunk partials stay
//             in LDS (no partial[K][nchunk][91] round trip through HBM: 3.1 of the 4.2 MB the psi launch moved at C3,
//             profiles/r02_traffic.json);
//   phase 3   the item's wave: ordered chunk sum, back-transform from LDS operands; an IDLE wave of the block (blocks of at
//             most two items) forms the same cost from the m0 entries and runs the tail protocol (arrival count, the last one
//             sums and publishes) beside it.
// The light set rides in the shadow of the heavy one (as in moments_orbit_pair_kernel's stacked form), the latency-bound
// phases 1 and 3 are paid once per block instead of once per launch with its ramp, and two kernel boundaries are gone.
// Nothing of the per-pass products goes to memory: S^-T, H and u0 stay in LDS between the phases (the host marks the sets'
// products as not resident; a later pass at the same state runs its own prep).
#pragma once
#include "kernels_orbit.hpp"

namespace gvi {

constexpr int FUSED_MAX_ITEMS = 4;
#ifndef GVI_FUSED_HELPERS
#define GVI_FUSED_HELPERS 1     // 0: every item's wave forms its own H and u0 (A/B build)
#endif

struct FusedSet {
  FactorDev f;
  OrbitArgs oa;                  // nchunk = 4: one chunk per wave
  const double* mu;              // [K][d], [K][d][d] inputs when the launch does not gather
  const double* Sigma;
  const int32_t* start;          // gather: first state of every factor
  double* mu_k;                  // gather outputs (= the inputs of every later pass at this state)
  double* Sigma_k;
  double* Ephi;
  double* cost;
  double* Vdmu;
  double* Vddmu;
};

struct FusedArgs {
  int nsets;
  int nblk;                      // blocks that own items; the ones behind write the chain-level trial mean
  int koff[3];                   // arrival ids of the tail: item k of set s is koff[s] + k
  FusedSet s[2];
  // fused gather (see PrepList): factor marginals pulled out of the chain arrays, the trial mean gmu + gstep gdmu formed on
  // the fly
  int gather, n;
  const double* gmu;
  const double* gdmu;
  double gstep;
  const double* SigD;
  const double* SigU;
  double* mu_out;
  int64_t nmu;
  EpiTail tail;
  CostList cl;
};

// LDS of one item's phase-1 / phase-3 work (doubles): prep area (+ gather staging) or epilogue area.  Item i works in the
// walk region of wave i (dead outside phase 2).
__host__ __device__ inline size_t fused_item_doubles(int d) {
  const size_t dd = (size_t)d * d, dp = d + (d & 1);
  const size_t prep = 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1 + dd + d + 2;
  const size_t epi = epilogue_lds_doubles(d) + 2;
  return ((prep > epi ? prep : epi) + 1) & ~(size_t)1;
}
// per-wave region: the walk's H + accumulator copies, or the item area if that is larger
__host__ __device__ inline size_t fused_region_doubles(int dmax, int M, int copies) {
  const size_t a = (size_t)orbit_lds_doubles(dmax, M, copies), b = fused_item_doubles(dmax);
  return ((a > b ? a : b) + 1) & ~(size_t)1;
}
// per item, outside the regions: S^-T (phase 3), the psi operands H (walk layout) and u0 (phases 1 -> 2)
__host__ __device__ inline size_t fused_keep_doubles(int dmax, int M) {
  return (size_t)dmax * dmax + (size_t)dmax * orbit_hstride(M) + (size_t)(M + (M & 1));
}
// layout: [4 regions] [Z: items x (S^-T | H | u0)] [P: items x 4 x npairs(dmax) (chunk partials)]
__host__ __device__ inline size_t fused_lds_doubles(int dmax, int M, int copies, int items) {
  return 4 * fused_region_doubles(dmax, M, copies) + (size_t)items * (fused_keep_doubles(dmax, M) + 4 * (size_t)npairs(dmax));
}


// (mu_k, Sigma_k) of factor k out of the chain arrays (the trial mean gmu + gstep gdmu formed on the fly), into LDS for the
// prep and into the set's arrays.  Compile-time d: ALL loads of the wave are issued before the first store -- with the
// runtime-d loop (division, three rounds of load -> store) the gather alone took 3.3 us of the launch's first phase.
template <int DT>
__device__ __forceinline__ bool fused_gather(const FusedArgs& A, const FusedSet& S, const int k, const int lane, double* Sl, double* ml,
                                             const LazyPred& lpred) {
  constexpr int dd = DT * DT, NI = (dd + 63) / 64;
  PREP_STAMP(8);
  const int n = A.n, nn = n * n;
  const int s = S.start ? S.start[k] : k;
  double v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = lane + 64 * i, r = e / DT, c = e % DT;
    const double* p;
    if (r < n && c < n) p = A.SigD + (size_t)s * nn + r * n + c;
    else if (r >= n && c >= n) p = A.SigD + (size_t)(s + 1) * nn + (r - n) * n + (c - n);
    else if (r < n) p = A.SigU + (size_t)s * nn + r * n + (c - n);
    else p = A.SigU + (size_t)s * nn + c * n + (r - n);
    v[i] = e < dd ? *p : 0.0;
  }
  double m = 0.0;
  if (lane < DT) {
    const size_t j = (size_t)s * n + lane;
    m = A.gdmu ? A.gmu[j] + A.gstep * A.gdmu[j] : A.gmu[j];
  }
#ifdef GVI_FUSED_TIMING
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  PREP_STAMP(9);
  if (pred_fail(lpred)) return true;             // the launch's predicate (block-uniform), in front of the first store
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = lane + 64 * i;
    if (e < dd) { Sl[e] = v[i]; S.Sigma_k[(size_t)k * dd + e] = v[i]; }
  }
  if (lane < DT) { ml[lane] = m; S.mu_k[(size_t)k * DT + lane] = m; }
  PREP_STAMP(10);
  return false;
}

// D0 / D1: the factor dimensions of set 0 / set 1 at compile time (both sets on the Cholesky route): only the bodies of
// these two shapes are compiled in -- with the runtime-d dispatch of prep_body_d / epilogue_body_p every shape's body was
// inlined into one 230 KB kernel, whose phase 1 then ran at the speed of its instruction fetches (30 us instead of 4)
template <int M, int SMAX, int WAVES, int D0, int D1>
__global__ __launch_bounds__(256, WAVES) void factor_fused_kernel(FusedArgs A, int dmax, int copies_max, int maxitems, [[maybe_unused]] unsigned long long* stamps) {
  extern __shared__ double sm[];
  // -DGVI_FUSED_TIMING + GVI_FUSED_DBG=8: 100 MHz stamps of wave 0 of every 146th block at the phase boundaries
#ifdef GVI_FUSED_TIMING
#define FUSED_STAMP(i) do { const unsigned long long t_ = (i) == 0 ? t_start : __builtin_amdgcn_s_memrealtime(); if (stamps && (threadIdx.x & 63) == 0 && (blockIdx.x % 146) == 0) stamps[(blockIdx.x / 146) * 32 + (threadIdx.x >> 6) * 8 + (i)] = t_; } while (0)
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();     // (before the first argument is fetched)
#else
#define FUSED_STAMP(i) do { } while (0)
#endif
#ifdef GVI_FUSED_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0) { gvi_prep_stamps = stamps ? stamps + 8 * 32 : nullptr; gvi_walk_stamps = stamps ? stamps + 352 : nullptr; }
#endif
  FUSED_STAMP(0);
#ifdef GVI_FUSED_TIMING
  if (stamps && (threadIdx.x & 63) == 0 && (blockIdx.x % 146) == 0)       // HW_ID (simd_id bits 5:4, cu_id 11:8, se_id 15:13) and XCC_ID
    stamps[320 + (blockIdx.x / 146) * 4 + (threadIdx.x >> 6)] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
  // The scalars of the prologue, requested in ONE batch: left to the compiler every branch below fetched its own kernel
  // arguments behind the branch before it -- six dependent round trips of the scalar cache, 1.7 us from the wave's start to
  // its first load of data (stamps of the timing build)
  const double* const pred_p = A.tail.pred;
  const int nblk_ = A.nblk, K0 = A.s[0].f.K, K1_ = A.s[1].f.K, nsets_ = A.nsets;
  // ... behind one pass over every line of the (1.6 KB) argument block (kernarg_warm)
  constexpr int KA_LINES = (sizeof(FusedArgs) + 3 * 8 + 63) / 64;
  static_assert(KA_LINES == 26, "kernarg_warm: one specialisation per argument block size");
  kernarg_warm<KA_LINES>();
  PREP_STAMP(11);
  asm volatile("" :: "s"(pred_p), "s"(nblk_), "s"(K0), "s"(K1_), "s"(nsets_));
  PREP_STAMP(12);
  // the launch's predicate (pipelined iterations): requested here, waited for in front of the first store to memory
  const LazyPred lpred = pred_issue(pred_p, A.tail.pred_val);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (int)blockIdx.x;
  if (b >= nblk_) {                                           // chain-level trial mean (gather mode only)
    const int64_t j = (int64_t)(b - A.nblk) * 256 + threadIdx.x;
    const double v = j < A.nmu ? A.gmu[j] + A.gstep * A.gdmu[j] : 0.0;
    if (pred_fail(lpred)) return;
    if (j < A.nmu) A.mu_out[j] = v;
    return;
  }
  // items of this block: set 0's b, b + nblk, ...; then set 1's
  const int K1 = nsets_ > 1 ? K1_ : 0;
  int c0 = 0, c1 = 0;                                          // (counted, not divided: at most FUSED_MAX_ITEMS of them)
#pragma unroll
  for (int j = 0; j < FUSED_MAX_ITEMS; ++j) { c0 += b + j * nblk_ < K0; c1 += b + j * nblk_ < K1; }
  const int nitems = c0 + c1;                                  // <= FUSED_MAX_ITEMS (host)
  // Which wave takes which item in phases 1 and 3.  Those phases are dependent chains of one wave per item (d = 12: 1500
  // instructions, three times the d = 6 item) and a block has two idle waves in them, so what matters is which SIMD an item's
  // wave sits on: the four resident blocks of a CU start their waves on the SIMDs in a rotating order, and with item i on
  // wave i three of the four d = 12 items of a CU could share one SIMD (stamps + HW_ID of the timing build: phase 1 ended at
  // 8 us in two residency slots and at 13 us in the other two).  The map is taken from the hardware ids instead: the wave on
  // SIMD (slot + i) % 4 takes item i, slot = the block's wave slot on its SIMDs (distinct among the blocks of a CU).  It is
  // a bijection whenever the block's four waves sit on four SIMDs (checked; else item i stays on wave i), and the results
  // do not depend on it.
  __shared__ int hw_simd[4];
  __shared__ int l_ready[FUSED_MAX_ITEMS];     // products phase: item i's Cholesky factor is in its wave's LDS area (prep_chol_hu0)
  const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // HW_ID.simd_id
  const int slot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_ID.wave_id
  if ((threadIdx.x & 63) == 0) { hw_simd[wave] = simd | (slot << 4); l_ready[wave] = 0; }
  __syncthreads();
  PREP_STAMP(13);
  const int h0 = hw_simd[0], h1 = hw_simd[1], h2 = hw_simd[2], h3 = hw_simd[3];
  const bool spread = ((1 << (h0 & 3)) | (1 << (h1 & 3)) | (1 << (h2 & 3)) | (1 << (h3 & 3))) == 15;
  const int iw = __builtin_amdgcn_readfirstlane(spread ? (simd - (h0 >> 4)) & 3 : wave);
  const size_t region = fused_region_doubles(dmax, M, copies_max);
  double* Zbase = sm + 4 * region;                             // [items][S^-T | H | u0]
  const size_t zs = fused_keep_doubles(dmax, M), ps = (size_t)4 * npairs(dmax);
  const size_t oH = (size_t)dmax * dmax, oU = oH + (size_t)dmax * orbit_hstride(M);
  double* Pbase = Zbase + (size_t)maxitems * zs;               // [items][4][npairs(dmax)]
  // ---- phase 1: wave i forms the products of item i; the block's idle waves (4 - nitems of them) form H and u0 of items
  // 0 .. 3 - nitems beside it (prep_chol_hu0): the item's own chain is then gather -> Cholesky -> L^-1 ----
  const int nhelp = GVI_FUSED_HELPERS ? min(4 - nitems, nitems) : 0;      // items [0, nhelp) have a helper: the wave with iw = nitems + item
  const auto helped = [&](const int it) {                      // (the shapes whose rows of A fit two registers a lane: all the fused route takes)
    const FactorDev& f = A.s[it < c0 ? 0 : 1].f;
    return it < nhelp && f.m > 0 && f.m * f.d <= 128;
  };
  if (iw < nitems) {
    const int si = iw < c0 ? 0 : 1;
    const int k = b + (iw < c0 ? iw : iw - c0) * A.nblk;
    const FusedSet& S = A.s[si];
    const FactorDev& f = S.f;
    const int d = f.d, dd = d * d, lane = threadIdx.x & 63;
    double* area = sm + (size_t)wave * region;
    double* Zs = Zbase + (size_t)iw * zs;
    int* flag = helped(iw) ? &l_ready[iw] : nullptr;
    if (!A.gather) {
      if (si == 0) prep_chol_body<D0>(f, S.mu, S.Sigma, k, area, k, Zs, Zs + oH, Zs + oU, orbit_hstride(M), flag);
      else prep_chol_body<D1>(f, S.mu, S.Sigma, k, area, k, Zs, Zs + oH, Zs + oU, orbit_hstride(M), flag);
    } else {
      const int dp = d + (d & 1);
      double* Sl = area + 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1;   // behind prep_body's own LDS
      double* ml = Sl + dd;
      if (si == 0 ? fused_gather<D0>(A, S, k, lane, Sl, ml, lpred) : fused_gather<D1>(A, S, k, lane, Sl, ml, lpred)) return;
      wave_lds_sync();
      if (si == 0) prep_chol_body<D0>(f, ml, Sl, k, area, 0, Zs, Zs + oH, Zs + oU, orbit_hstride(M), flag);
      else prep_chol_body<D1>(f, ml, Sl, k, area, 0, Zs, Zs + oH, Zs + oU, orbit_hstride(M), flag);
    }
  } else if (helped(iw - nitems)) {
    const int it = iw - nitems;                               // the item this wave helps
    const int si = it < c0 ? 0 : 1;
    const int k = b + (it < c0 ? it : it - c0) * A.nblk;
    const FusedSet& S = A.s[si];
    const int d = S.f.d, lane = threadIdx.x & 63;
    // the wave that runs item `it` (the inverse of the SIMD map above), whose LDS area holds L
    int wi = it;
    if (spread) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int hw = w == 0 ? h0 : (w == 1 ? h1 : (w == 2 ? h2 : h3));
        if ((((hw & 3) - (h0 >> 4)) & 3) == it) wi = w;
      }
    }
    double* own = sm + (size_t)wave * region;
    double* Zs = Zbase + (size_t)it * zs;
    // the item's mean, as its own wave gathers it (same expression: bit-identical u0)
    double* mh = own + (size_t)S.f.m * d;
    if (lane < d) {
      double mval;
      if (A.gather) {
        const size_t j = (size_t)(S.start ? S.start[k] : k) * A.n + lane;
        mval = A.gdmu ? A.gmu[j] + A.gstep * A.gdmu[j] : A.gmu[j];
      } else mval = S.mu[(size_t)k * d + lane];
      mh[lane] = mval;
    }
    if (!pred_fail(lpred)) {
      if (si == 0) prep_chol_hu0<D0>(S.f, mh, k, own, sm + (size_t)wi * region, &l_ready[it], Zs + oH, Zs + oU, orbit_hstride(M));
      else prep_chol_hu0<D1>(S.f, mh, k, own, sm + (size_t)wi * region, &l_ready[it], Zs + oH, Zs + oU, orbit_hstride(M));
    }
  }
  FUSED_STAMP(1);
  if (pred_fail(lpred)) return;                                // (waves without an item, and the route without the gather)
  __syncthreads();                                             // S^-T / H / u0 of the items are in LDS
  FUSED_STAMP(2);
  // ---- phase 2: every wave walks its chunk of the orbit table, item after item ----
  for (int it = 0; it < nitems; ++it) {
    const int si = it < c0 ? 0 : 1;
    const int k = b + (it < c0 ? it : it - c0) * A.nblk;
    const FusedSet& S = A.s[si];
    const double* Zi = Zbase + (size_t)it * zs;
    orbit_wave<M, SMAX, true, false, true>(S.oa, k, wave, sm + (size_t)wave * region, Pbase + (size_t)it * ps + (size_t)wave * npairs(S.f.d),
                                           Zi + oH, Zi + oU);
  }
  FUSED_STAMP(3);
  __syncthreads();
  FUSED_STAMP(4);
  // ---- phase 3: the item's wave: ordered chunk sum, cost (+ tail), back-transform ----
  // A block with at most two items has two idle waves here: wave (item + 2) runs the item's tail protocol (the cost needs only
  // the m0 entries of the four chunks: same ordered sum, same division as the item's wave) while the item's wave goes straight
  // through chunk sums and back-transform -- the ~1 us round trip of the arrival atomics leaves every item's chain.
  const bool helpers = A.tail.on && nitems <= 2;
  if (iw >= nitems) {
    const int it = iw - 2;
    if (!helpers || it < 0 || it >= nitems) return;
    const int si = it < c0 ? 0 : 1;
    const int k = b + (it < c0 ? it : it - c0) * A.nblk;
    const FusedSet& S = A.s[si];
    const double Tk = S.f.temperature[k];
    const double* Pl = Pbase + (size_t)it * ps;
    const int npo = npairs(S.f.d);
    double m0 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) m0 += Pl[(size_t)q * npo];      // the order of epilogue_body_t's chunk sum
    const double costk = m0 / Tk;
    epi_tail_arrive(A.cl, A.tail, S.cost + k, costk, (int)(threadIdx.x & 63), (unsigned)(A.koff[si] + k), (unsigned)A.koff[A.nsets],
                    (int*)(sm + (size_t)wave * region));
    return;
  }
  {
    const int si = iw < c0 ? 0 : 1;
    const int k = b + (iw < c0 ? iw : iw - c0) * A.nblk;
    const FusedSet& S = A.s[si];
    const int d = S.f.d;
    double* area = sm + (size_t)wave * region;
    double* Zs = Zbase + (size_t)iw * zs;
    const double* Pl = Pbase + (size_t)iw * ps;
    EpiArgs e;
    e.f = S.f; e.partial = nullptr; e.nchunk = 4; e.full = 1;
    e.Ephi = S.Ephi; e.cost = helpers ? nullptr : S.cost; e.Vdmu = S.Vdmu; e.Vddmu = S.Vddmu; e.E_xmuphi = nullptr; e.E_xxphi = nullptr;
    auto epi = [&](int phase) { return si == 0 ? epilogue_body_t<D0>(e, k, area, phase, Pl, Zs) : epilogue_body_t<D1>(e, k, area, phase, Pl, Zs); };
    if (!A.tail.on || helpers) { epi(0); return; }
    const double costk = epi(1);
    FUSED_STAMP(5);
    int* last = (int*)(area + epilogue_lds_doubles(d));
    epi_tail_arrive(A.cl, A.tail, S.cost + k, costk, (int)(threadIdx.x & 63), (unsigned)(A.koff[si] + k), (unsigned)A.koff[A.nsets], last);
    FUSED_STAMP(6);
    epi(2);
    FUSED_STAMP(7);
  }
}

}  // namespace gvi
