// moments_orbit_kernel: the sum-of-squares psi kinds on the SIGN-ORBIT form of the sparse Gauss-Hermite table.
//
// The lane-per-point kernels (kernels_factor.hpp) treat a sigma point as a dense d-vector: at (12,5) they spend 72 FMAs on
// H z and 90 on c z z^T per evaluation although a point has at most FOUR non-zero coordinates, and every point is one of
// 2^s sign images of the same magnitudes.  Here a lane owns a whole orbit (csrc/orbits.hpp: support c_0 < ... < c_{s-1},
// magnitudes m_j, one weight w) of ONE factor (wave = factor x chunk of orbit tiles; the factor's H and its 91 moment
// accumulators live in LDS):
//   * only the s columns H[:, c_j] of the support are touched: they are read from LDS once per orbit;
//   * the 2^s sign patterns are walked in Gray-code order, one coordinate flips per step:  v += +-2 m_j H[:, c_j]  (M FMAs);
//   * only HALF the orbit is walked: for a +-pair the cross terms cancel (kernels_factor.hpp, sreg_pipe_body):
//         psi(z) + psi(-z) = 2 (q + k0),  psi(z) - psi(-z) = 4 l,   q = sum_r s_r v_r^2,  l = sum_r (s_r u0_r) v_r,  v = H z;
//   * within the orbit the moments are sign-weighted sums of those two scalars (Walsh sums):
//         m0 += 2w sum(q + k0);   m1[c_i] += 4w m_i sum sigma_i l;   M2[c_i][c_j] += 2w m_i m_j sum sigma_i sigma_j (q + k0);
//     they are accumulated in registers with compile-time signs and added to the factor's accumulators ONCE per orbit;
//   * those adds go to data-dependent entries (the orbit's coordinates): LDS atomics (ds_add_f64).  The lanes of one
//     instruction are served in a fixed order and every wave owns its accumulators, so results are run-to-run
//     bit-identical (tests/test_gpu_parity.py checks it); the chunk partials are summed in fixed order by the epilogue as
//     for every other kernel.
// Per evaluation at (12,5): ~14 fp64 instructions in the walk + ~9 of per-orbit work, against 96 for the +-paired
// lane-per-point kernel and 319 FMAs for the reference's x-space algorithm.  Registers: ~110 (M = 6, s = 4), so four and
// more waves per SIMD instead of two.  The same kernel serves d = 24 (M = 12, s <= 6): accumulators and H sit in LDS, not
// in registers, so the factor dimension no longer decides the kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_factor.hpp"

namespace gvi {

#ifndef ORBIT_WHT
#define ORBIT_WHT 1      // 0: per-point accumulation of the sign-weighted sums for every support size (A/B build)
#endif

#ifdef GVI_FUSED_TIMING
__device__ unsigned long long* gvi_walk_stamps;
#endif

struct OrbitDev {
  const uint64_t* cpk;       // [norb_p] support coordinates, one byte each
  const uint64_t* rpk;       // [norb_p] packed-triangle row bases of the coordinates, ten bits each (OrbitRec)
  const double* mag;         // [smax][norb_p]
  const double* w;           // [norb_p]
  const int32_t* bounds;     // [nchunk + 1] tile ranges of the chunks
  int64_t norb_p;
  double w0;                 // weight of the origin
  // The tile list without memory accesses: the classes are stored by descending support size, every class padded to whole
  // tiles, so tile t starts at orbit 64 t and has support s iff cend[s + 1] <= t < cend[s] (cend[s] = one past the last
  // tile of class s; 0 above the table's largest support).  bnd: the chunk bounds themselves when nchunk <= 4 (nb =
  // nchunk; else 0 and `bounds` is read).  A wave used to start every tile with two dependent round trips to memory
  // (tile_s / tile_first, then the orbit records): ~1200 cycles per tile whatever its support (walk stamps of the timing
  // build: 1450 / 2290 / 2950 / 3450 cycles per s = 1 / 2 / 3 / 4 tile against 280 / 440 / 720 / 1340 of VALU issue).
  int32_t cend[8];
  int32_t bnd[5];
  int32_t nb;
  // support-major classes (orbits.hpp): lane `l` of the class's tile `tl` owns the cgrp[s] orbits at entries
  // cbase[s] + 64 tl + l + g cstride[s], g = 0 .. cgrp[s] - 1 (all of one support; cgrp = 1: one orbit per lane)
  int32_t cbase[8], cgrp[8], cstride[8];
};

// the record of one orbit (one lane of a tile).  rpk: ten bits per support coordinate, R_i = packed-triangle index of
// (c_i, 0) in the accumulator layout [m0 | m1[d] | upper triangle by rows], i.e. entry (c_i, c_j) sits at R_i + c_j
// (orbits.hpp) -- computing it per orbit cost ~6 integer instructions per coordinate in a walk that is bound by VALU issue
template <int S>
struct OrbitRec { uint64_t cpk, rpk; double w; double mg[S]; };
// element `boff` BYTES behind a wave-uniform base: scalar base + 32-bit lane offset, no 64-bit address arithmetic
template <typename T>
__device__ __forceinline__ T orbit_ld(const T* base, const uint32_t boff) { return *(const T*)((const char*)base + boff); }
// boff = 8 * orbit index (< 2^31 orbits, host-checked: 8 smax norb_p < 2^32 too)
template <int S>
__device__ __forceinline__ void orbit_load(const OrbitDev& ob, const uint32_t boff, OrbitRec<S>& r) {
  r.cpk = orbit_ld(ob.cpk, boff);
  if constexpr (S >= 1) r.rpk = orbit_ld(ob.rpk, boff);
  r.w = orbit_ld(ob.w, boff);
#pragma unroll
  for (int j = 0; j < S; ++j) r.mg[j] = orbit_ld(ob.mag + (size_t)j * ob.norb_p, boff);
}
struct OrbitNoPre { __device__ __forceinline__ void operator()() const {} };

// 64-bit DPP move (two 32-bit halves): cross-lane sums without the LDS pipe -- __shfl_xor is ds_bpermute, which queues
// behind the wave's own and the other waves' accumulator atomics
template <int CTRL>
__device__ __forceinline__ double dpp_f64(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, valid in every lane's SGPR copy (fixed association: quads, rows of 16, then the four rows)
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);        // quad_perm [1, 0, 3, 2]
  v += dpp_f64<0x4E>(v);        // quad_perm [2, 3, 0, 1]
  v += dpp_f64<0x124>(v);       // row_ror 4
  v += dpp_f64<0x128>(v);       // row_ror 8
  auto row = [&](int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  return (row(0) + row(16)) + (row(32) + row(48));
}

// one factor set on the orbit kernel: the psi operands of prep_kernel + the orbit table
struct OrbitArgs {
  const double* H;           // [K][d][M]
  const double* u0;          // [K][M]
  const double* sgn;         // [K][M]
  double* partial;           // [K][nchunk][npairs(d)] (full) or [K][nchunk] (cost)
  int K, d, nchunk;
  int copies;                // private copies of every accumulator entry (power of two <= 16), selected by lane % copies
  const double* pred;        // predicated launch (device_common.hpp, pred_skip) or null
  double pred_val;
  OrbitDev ob;
};

// Column stride of H in LDS.  The lanes of a tile read DIFFERENT columns with 16-byte loads: at m = 12 a stride of 12
// doubles (24 banks) maps the columns onto 8 bank offsets (8-way conflicts), 14 doubles (28 banks, still 16-byte aligned)
// onto 16.  m <= 6: 12 banks, conflict-free up to 16 columns.
__host__ __device__ constexpr int orbit_hstride(int M) { return M == 12 ? 14 : M; }

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef GVI_EXP_FLATATOM
#define GVI_EXP_FLATATOM 0     // timing experiment (WRONG results): every accumulator add of a lane goes to the lane's own address
#endif                         // (64 consecutive doubles per instruction: what the adds cost without bank / address conflicts)
#if GVI_EXP_FLATATOM
__device__ __forceinline__ char* accb_flat(char* base, int sh) { return base - (((int)(threadIdx.x & 63) & ((1 << (sh - 3)) - 1)) << 3); }
// (= 2: the real address is still computed -- kept alive by an empty asm -- so that only the conflicts are taken out;
//  = 3: flat targets, and the record's coordinates / row bases kept live up to the adds without being used)
__device__ __forceinline__ double* orbit_flat_target(char* real, char* base, int n, int sh) {
#if GVI_EXP_FLATATOM == 2
  asm volatile("" :: "v"(real));
#endif
  return (double*)(accb_flat(base, sh) + ((((n) & 1) * 64 + (int)(threadIdx.x & 63)) << 3));
}
#define ORBIT_ADD_TARGET(expr, n, sh) orbit_flat_target((char*)(expr), base, n, sh)
#else
#define ORBIT_ADD_TARGET(expr, n, sh) ((double*)(expr))
#endif
// The accumulator adds of one orbit.  Values and LDS addresses are formed FIRST (the record's registers die there), then
// pre() requests the next tile's records into those registers, then the adds go out.  accb: byte address of this lane's
// copy of entry 0; entry e lives at accb + (e << sh), sh = log2(copies) + 3.
template <int S, typename Pre>
__device__ __forceinline__ void orbit_accumulate(const int sh, const int (&c)[S], const uint64_t rpk, const double (&mg)[S], const double wp,
                                                 const double E0, const double (&Eij)[S * (S - 1) / 2 + 1], const double (&Oi)[S],
                                                 double* accb, Pre&& pre) {
  constexpr int NV = 2 * S + S * (S - 1) / 2;
  const double w4 = wp + wp;
  double val[NV];
  {
    // The scale factors w m_i, w m_i m_j depend on the record alone: left to itself the scheduler forms all of them at the
    // top of the tile and keeps fourteen doubles alive through the walk (28 registers: the kernel then spills inside its
    // hot loop, 26 -> 39 us).  The empty asm ties the magnitudes to E0, which is final only here.
    double mgl[S];
#pragma unroll
    for (int j = 0; j < S; ++j) { mgl[j] = mg[j]; asm volatile("" : "+v"(mgl[j]) : "v"(E0)); }
    int n = 0, e = 0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const double wm = wp * mgl[i];
      val[n++] = w4 * mgl[i] * Oi[i];
      val[n++] = wm * mgl[i] * E0;
#pragma unroll
      for (int j = i + 1; j < S; ++j) val[n++] = wm * mgl[j] * Eij[e++];
    }
  }
#if GVI_EXP_FLATATOM == 3      // (the record's coordinates and row bases stay live up to here, but no address is formed from them)
#pragma unroll
  for (int i = 0; i < S; ++i) asm volatile("" :: "v"(c[i]));
  asm volatile("" :: "v"(rpk));
#endif
  pre();
  char* const base = (char*)accb;
  char* const base1 = base + (1u << sh);                         // m1[0]
  unsigned A[S], B[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const unsigned lo = (unsigned)rpk, hi = (unsigned)(rpk >> 32);
    const unsigned R = i < 3 ? (lo >> (10 * i)) & 1023u : (hi >> (10 * (i - 3))) & 1023u;
    A[i] = R << sh;
    B[i] = (unsigned)c[i] << sh;
  }
  int n = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    lds_add_f64(ORBIT_ADD_TARGET(base1 + B[i], n, sh), val[n]); ++n;
    lds_add_f64(ORBIT_ADD_TARGET(base + (A[i] + B[i]), n, sh), val[n]); ++n;
#pragma unroll
    for (int j = i + 1; j < S; ++j) { lds_add_f64(ORBIT_ADD_TARGET(base + (A[i] + B[j]), n, sh), val[n]); ++n; }
  }
}

#ifndef ORBIT_LB4
#define ORBIT_LB4 2
#endif
#ifndef GVI_EXP_EXTRA_FMA
#define GVI_EXP_EXTRA_FMA 0      // timing experiment: this many extra independent fp64 FMAs per half-point
#endif
#ifndef GVI_EXP_NOCOL
#define GVI_EXP_NOCOL 0        // timing experiment (WRONG results): the columns of H are made up instead of read from LDS
#endif
#ifndef GVI_EXP_NOATOM
#define GVI_EXP_NOATOM 0       // timing experiment (WRONG results): the accumulator adds of supports <= this size are skipped
#endif


// one orbit per lane, support size S (all lanes of the wave: tiles are uniform in S)
// pre(): called between the Gray walk and the accumulator adds -- where the walk's registers are dead -- to request the
// NEXT tile's records (orbit_class)
template <int M, int S, bool FULL, bool SIGNED, typename Pre>
__device__ __forceinline__ void orbit_walk(const int lc, const uint64_t cpk, const uint64_t rpk, const double (&mg)[S], const double w,
                                           const double* Hl, double* accl, const double (&su0)[M], const double (&sg)[M],
                                           const double k0, double& m0, Pre&& pre) {
  int c[S];
#pragma unroll
  for (int j = 0; j < S; ++j) c[j] = (int)((cpk >> (8 * j)) & 255u);
  // the support's columns of H: in registers while they fit (M S <= 36 doubles); otherwise the column of the NEXT flip is
  // fetched from LDS while the current point is evaluated (the compiler barrier keeps the loads from being hoisted back
  // into one register-resident block)
  // m = 6, s = 4 (24 doubles) would fit too, but the kernel then needs more than the 128 registers of four waves per SIMD
  // and spills inside this loop (measured: 26 -> 39 us); its columns stay in LDS, column 0 in registers (H0REG)
  constexpr bool HREG = M * S <= 36 && !(M == 6 && S >= 4);
  constexpr int NH = 1 << (S - 1);
  double hcol[HREG ? S : 1][M];
  const double* hp[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    hp[j] = Hl + c[j] * orbit_hstride(M);
    if constexpr (HREG) {
#pragma unroll
      for (int r = 0; r < M; ++r) hcol[j][r] = GVI_EXP_NOCOL ? (double)(c[j] + r) * w : hp[j][r];
    }
  }
  // start at the corner (-, ..., -, +): the last coordinate keeps its sign, the other S-1 are walked in Gray order
  int sig[S];
#pragma unroll
  for (int j = 0; j < S; ++j) sig[j] = j == S - 1 ? 1 : -1;
  double v[M];
#pragma unroll
  for (int r = 0; r < M; ++r) v[r] = mg[S - 1] * (HREG ? hcol[S - 1][r] : hp[S - 1][r]);
#pragma unroll
  for (int j = 0; j < S - 1; ++j) {
    if constexpr (!HREG) {                                      // one column in flight at a time
#pragma unroll
      for (int r = 0; r < M; ++r) asm volatile("" : "+v"(v[r]) :: "memory");
    }
#pragma unroll
    for (int r = 0; r < M; ++r) v[r] = fma(-mg[j], HREG ? hcol[j][r] : hp[j][r], v[r]);
  }
  // !HREG: the column of coordinate 0 -- flipped at every other step of the Gray walk -- stays in registers when H0REG
  constexpr bool H0REG = !HREG && M * (S + 9) <= 168;     // m = 12: s <= 5 (s = 6 would spill)
  double h0[H0REG ? M : 1];
  if constexpr (H0REG) {
#pragma unroll
    for (int r = 0; r < M; ++r) h0[r] = hp[0][r];
  }
  // Sign-weighted (Walsh) sums.  The two scalars of 8 consecutive half-points (a reflected Gray code walks the low three
  // coordinates through all their sign patterns before it touches a higher one) are kept and ONE Walsh-Hadamard butterfly
  // per scalar yields their sums over every mask of those coordinates: 2 x 24 additions instead of 8 x (1 + 6 + 4) at
  // s = 4 (unused outputs are dead code); the signs of the higher coordinates are constant inside a block and multiply the
  // block's sums.  WHT_BIG: also for s = 5, 6 (needs the registers: not at m = 12).
  constexpr bool WHT_BIG = M <= 6;
  constexpr bool WHT = FULL && S >= 2 && ORBIT_WHT && (S <= 4 || WHT_BIG);
  // coordinates whose signs vary inside a block of 2^LB consecutive Gray steps.  s = 4 takes two blocks of four instead of
  // one of eight: the same number of additions (two 4-point butterflies per scalar + 11 signed adds for the second block
  // against one 8-point butterfly) with 16 fewer live registers -- which is what lets the next tile's records be requested
  // under the current tile's accumulator adds without spilling (orbit_class)
  constexpr int LBMAX = S == 4 ? ORBIT_LB4 : 3;
  constexpr int LB = S - 1 < LBMAX ? S - 1 : LBMAX;
  constexpr int BLK = 1 << LB;
  double cpv[WHT ? BLK : 1], lv[WHT ? BLK : 1];
  double E0 = 0.0, Eij[S * (S - 1) / 2 + 1], Oi[S];
#pragma unroll
  for (int e = 0; e < S * (S - 1) / 2 + 1; ++e) Eij[e] = 0.0;
#pragma unroll
  for (int j = 0; j < S; ++j) Oi[j] = 0.0;
#if GVI_EXP_EXTRA_FMA
  double xdum[6] = {0, 0, 0, 0, 0, 0};
#endif
#pragma unroll
  for (int g = 0; g < NH; ++g) {
    const int jn = g + 1 < NH ? __builtin_ctz(g + 1) : 0;        // coordinate of the next flip (compile-time after unrolling)
    double hn[M];
    if constexpr (!HREG) {
      if (g + 1 < NH && !(H0REG && jn == 0)) {
        asm volatile("" : "+v"(E0) :: "memory");                // the previous point is finished before the next fetch
#pragma unroll
        for (int r = 0; r < M; ++r) asm volatile("" : "+v"(v[r]) :: "memory");
#pragma unroll
        for (int r = 0; r < M; ++r) hn[r] = hp[jn][r];
      }
    }
    double q = 0.0, l = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) {
      q = SIGNED ? fma(sg[r] * v[r], v[r], q) : fma(v[r], v[r], q);
      if (FULL) l = fma(su0[r], v[r], l);
    }
#if GVI_EXP_EXTRA_FMA
#pragma unroll
    for (int r = 0; r < GVI_EXP_EXTRA_FMA; ++r) xdum[r % 6] = fma(v[r % M], v[(r + 1) % M], xdum[r % 6]);
#endif
    const double cp = q + k0;
    if constexpr (WHT) {
      // keep (c+, l) of this sign pattern of the low coordinates; at the end of a block: butterfly, then add the block's
      // sums into the accumulators with the (compile-time) signs of the higher coordinates
      int b = 0;
#pragma unroll
      for (int j = 0; j < LB; ++j) b |= (sig[j] > 0 ? 1 : 0) << j;             // compile-time after unrolling
      cpv[b] = cp;
      lv[b] = l;
      if ((g & (BLK - 1)) == BLK - 1) {
#pragma unroll
        for (int j = 0; j < LB; ++j) {
#pragma unroll
          for (int bb = 0; bb < BLK; ++bb) {
            if (!(bb & (1 << j))) {            // (x[bit = 0], x[bit = 1]) -> (x0 + x1, x1 - x0): index = mask of multiplying signs
              const double c0v = cpv[bb], c1v = cpv[bb | (1 << j)], l0v = lv[bb], l1v = lv[bb | (1 << j)];
              cpv[bb] = c0v + c1v; cpv[bb | (1 << j)] = c1v - c0v;
              lv[bb] = l0v + l1v; lv[bb | (1 << j)] = l1v - l0v;
            }
          }
        }
        const bool first = g == BLK - 1;
        E0 = first ? cpv[0] : E0 + cpv[0];
        int e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          {
            const double t = i < LB ? lv[1 << i] : lv[0];
            const bool pos = i < LB || sig[i] > 0;
            Oi[i] = first ? (pos ? t : -t) : (pos ? Oi[i] + t : Oi[i] - t);
          }
#pragma unroll
          for (int j = i + 1; j < S; ++j) {
            const double t = j < LB ? cpv[(1 << i) | (1 << j)] : (i < LB ? cpv[1 << i] : cpv[0]);
            const bool pos = (j < LB ? 1 : sig[j]) * (i < LB ? 1 : sig[i]) > 0;
            Eij[e] = first ? (pos ? t : -t) : (pos ? Eij[e] + t : Eij[e] - t);
            ++e;
          }
        }
      }
    } else {
      E0 += cp;
      if (FULL) {
        int e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          Oi[i] = sig[i] > 0 ? Oi[i] + l : Oi[i] - l;
#pragma unroll
          for (int j = i + 1; j < S; ++j) { Eij[e] = sig[i] * sig[j] > 0 ? Eij[e] + cp : Eij[e] - cp; ++e; }
        }
      }
    }
    if (g + 1 < NH) {
      sig[jn] = -sig[jn];
      const double t2 = (sig[jn] > 0 ? 2.0 : -2.0) * mg[jn];
#pragma unroll
      for (int r = 0; r < M; ++r) v[r] = fma(t2, HREG ? hcol[jn][r] : ((H0REG && jn == 0) ? h0[r] : hn[r]), v[r]);
    }
  }
  const double wp = w + w;
  m0 = fma(wp, E0, m0);
#if GVI_EXP_EXTRA_FMA
  m0 = fma(1e-300, ((xdum[0] + xdum[1]) + (xdum[2] + xdum[3])) + (xdum[4] + xdum[5]), m0);
#endif
  if constexpr (FULL && S <= GVI_EXP_NOATOM) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < S; ++i) t += Oi[i];
#pragma unroll
    for (int e = 0; e < S * (S - 1) / 2; ++e) t += Eij[e];
    m0 = fma(1e-300, t, m0);
    pre();
  } else if constexpr (FULL) {
    orbit_accumulate<S>(lc + 3, c, rpk, mg, wp, E0, Eij, Oi, accl, pre);
  } else {
    pre();
  }
}

// m = 12 with s = 4 / 5 / 6 (BASELINE configs[4]: d = 24, degree 7): the twelve rows as TWO walks of six.  Every accumulated
// quantity is LINEAR in the two scalars of a point, q = sum_r s_r v_r^2 (+ k0) and l = sum_r s_r u0_r v_r, and both are sums
// over the rows: the first walk accumulates with (q over rows 0..5 + k0, l over rows 0..5), the second with the rest, into the
// same registers.  What that buys: the support's columns of six rows fit the registers (36 doubles at s = 6), so no column is
// fetched from LDS inside the walk -- orbit_walk at m = 12 fetches twelve doubles at (almost) every Gray step, and its wave
// waits for LDS in 15 % of its cycles -- and the Walsh butterfly of the sign-weighted sums (registers again) applies at
// s = 5 / 6 as it does at m = 6.  Same number of fp64 instructions per orbit; sums re-associated (rows 0..5 first).
#ifndef GVI_ORBIT_SPLIT12
#define GVI_ORBIT_SPLIT12 1
#endif
template <int S, bool SIGNED, typename Pre>
__device__ __forceinline__ void orbit_walk_split(const int lc, const uint64_t cpk, const uint64_t rpk, const double (&mg)[S], const double w,
                                                 const double* Hl, double* accl, const double (&su0)[12], const double (&sg)[12],
                                                 const double k0, double& m0, Pre&& pre) {
  constexpr int MH = 6, HS = orbit_hstride(12), NH = 1 << (S - 1), LB = 3, BLK = 8;
  static_assert(S >= 4 && S <= 6, "split walk: s = 4, 5, 6 (blocks of 8 Gray steps)");
  int c[S];
#pragma unroll
  for (int j = 0; j < S; ++j) c[j] = (int)((cpk >> (8 * j)) & 255u);
  double E0 = 0.0, Eij[S * (S - 1) / 2 + 1], Oi[S];
#pragma unroll
  for (int e = 0; e < S * (S - 1) / 2 + 1; ++e) Eij[e] = 0.0;
#pragma unroll
  for (int j = 0; j < S; ++j) Oi[j] = 0.0;
#pragma clang loop unroll(disable)
  for (int half = 0; half < 2; ++half) {
    double suh[MH], sgh[MH];
#pragma unroll
    for (int r = 0; r < MH; ++r) {
      suh[r] = half ? su0[MH + r] : su0[r];
      sgh[r] = SIGNED ? (half ? sg[MH + r] : sg[r]) : 1.0;
    }
    const double kk = half ? 0.0 : k0;
    double hcol[S][MH];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const double* hp = Hl + c[j] * HS + half * MH;
#pragma unroll
      for (int r = 0; r < MH; ++r) hcol[j][r] = hp[r];
    }
    int sig[S];                                      // corner (-, ..., -, +), as orbit_walk
#pragma unroll
    for (int j = 0; j < S; ++j) sig[j] = j == S - 1 ? 1 : -1;
    double v[MH];
#pragma unroll
    for (int r = 0; r < MH; ++r) v[r] = mg[S - 1] * hcol[S - 1][r];
#pragma unroll
    for (int j = 0; j < S - 1; ++j) {
#pragma unroll
      for (int r = 0; r < MH; ++r) v[r] = fma(-mg[j], hcol[j][r], v[r]);
    }
    double cpv[BLK], lv[BLK];
#pragma unroll
    for (int g = 0; g < NH; ++g) {
      const int jn = g + 1 < NH ? __builtin_ctz(g + 1) : 0;
      double q = 0.0, l = 0.0;
#pragma unroll
      for (int r = 0; r < MH; ++r) {
        q = SIGNED ? fma(sgh[r] * v[r], v[r], q) : fma(v[r], v[r], q);
        l = fma(suh[r], v[r], l);
      }
      const double cp = q + kk;
      int b = 0;
#pragma unroll
      for (int j = 0; j < LB; ++j) b |= (sig[j] > 0 ? 1 : 0) << j;
      cpv[b] = cp;
      lv[b] = l;
      if ((g & (BLK - 1)) == BLK - 1) {
#pragma unroll
        for (int j = 0; j < LB; ++j) {
#pragma unroll
          for (int bb = 0; bb < BLK; ++bb) {
            if (!(bb & (1 << j))) {
              const double c0v = cpv[bb], c1v = cpv[bb | (1 << j)], l0v = lv[bb], l1v = lv[bb | (1 << j)];
              cpv[bb] = c0v + c1v; cpv[bb | (1 << j)] = c1v - c0v;
              lv[bb] = l0v + l1v; lv[bb | (1 << j)] = l1v - l0v;
            }
          }
        }
        E0 += cpv[0];
        int e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          {
            const double t = i < LB ? lv[1 << i] : lv[0];
            const bool pos = i < LB || sig[i] > 0;
            Oi[i] = pos ? Oi[i] + t : Oi[i] - t;
          }
#pragma unroll
          for (int j = i + 1; j < S; ++j) {
            const double t = j < LB ? cpv[(1 << i) | (1 << j)] : (i < LB ? cpv[1 << i] : cpv[0]);
            const bool pos = (j < LB ? 1 : sig[j]) * (i < LB ? 1 : sig[i]) > 0;
            Eij[e] = pos ? Eij[e] + t : Eij[e] - t;
            ++e;
          }
        }
      }
      if (g + 1 < NH) {
        sig[jn] = -sig[jn];
        const double t2 = (sig[jn] > 0 ? 2.0 : -2.0) * mg[jn];
#pragma unroll
        for (int r = 0; r < MH; ++r) v[r] = fma(t2, hcol[jn][r], v[r]);
      }
    }
  }
  const double wp = w + w;
  m0 = fma(wp, E0, m0);
  orbit_accumulate<S>(lc + 3, c, rpk, mg, wp, E0, Eij, Oi, accl, pre);
}

// The Gray walk of ONE orbit, the support's columns in registers (orbit_walk's HREG case as a function of its own): the
// sign-weighted sums E0 = sum (q + k0), Oi = sum sigma_i l, Eij = sum sigma_i sigma_j (q + k0) over the half orbit.
template <int M, int S, bool FULL, bool SIGNED>
__device__ __forceinline__ void orbit_gray(const double (&hcol)[S][M], const double (&mg)[S], const double (&su0)[M], const double (&sg)[M],
                                           const double k0, double& E0, double (&Eij)[S * (S - 1) / 2 + 1], double (&Oi)[S]) {
  static_assert(S <= 3, "orbit_gray: one butterfly block");
  constexpr int NH = 1 << (S - 1), LB = S - 1, BLK = NH;
  constexpr bool WHT = FULL && S >= 2 && ORBIT_WHT;
  int sig[S];
#pragma unroll
  for (int j = 0; j < S; ++j) sig[j] = j == S - 1 ? 1 : -1;
  double v[M];
#pragma unroll
  for (int r = 0; r < M; ++r) v[r] = mg[S - 1] * hcol[S - 1][r];
#pragma unroll
  for (int j = 0; j < S - 1; ++j) {
#pragma unroll
    for (int r = 0; r < M; ++r) v[r] = fma(-mg[j], hcol[j][r], v[r]);
  }
  double cpv[WHT ? BLK : 1], lv[WHT ? BLK : 1];
  E0 = 0.0;
#pragma unroll
  for (int e = 0; e < S * (S - 1) / 2 + 1; ++e) Eij[e] = 0.0;
#pragma unroll
  for (int j = 0; j < S; ++j) Oi[j] = 0.0;
#pragma unroll
  for (int g = 0; g < NH; ++g) {
    const int jn = g + 1 < NH ? __builtin_ctz(g + 1) : 0;
    double q = 0.0, l = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) {
      q = SIGNED ? fma(sg[r] * v[r], v[r], q) : fma(v[r], v[r], q);
      if (FULL) l = fma(su0[r], v[r], l);
    }
    const double cp = q + k0;
    if constexpr (WHT) {
      int b = 0;
#pragma unroll
      for (int j = 0; j < LB; ++j) b |= (sig[j] > 0 ? 1 : 0) << j;
      cpv[b] = cp;
      lv[b] = l;
      if (g == NH - 1) {
#pragma unroll
        for (int j = 0; j < LB; ++j) {
#pragma unroll
          for (int bb = 0; bb < BLK; ++bb) {
            if (!(bb & (1 << j))) {
              const double c0v = cpv[bb], c1v = cpv[bb | (1 << j)], l0v = lv[bb], l1v = lv[bb | (1 << j)];
              cpv[bb] = c0v + c1v; cpv[bb | (1 << j)] = c1v - c0v;
              lv[bb] = l0v + l1v; lv[bb | (1 << j)] = l1v - l0v;
            }
          }
        }
        E0 = cpv[0];
        int e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          Oi[i] = i < LB ? lv[1 << i] : lv[0];              // (the last coordinate keeps its + sign)
#pragma unroll
          for (int j = i + 1; j < S; ++j) { Eij[e] = j < LB ? cpv[(1 << i) | (1 << j)] : cpv[1 << i]; ++e; }
        }
      }
    } else {
      E0 += cp;
      if (FULL) {
        int e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          Oi[i] = sig[i] > 0 ? Oi[i] + l : Oi[i] - l;
#pragma unroll
          for (int j = i + 1; j < S; ++j) { Eij[e] = sig[i] * sig[j] > 0 ? Eij[e] + cp : Eij[e] - cp; ++e; }
        }
      }
    }
    if (g + 1 < NH) {
      sig[jn] = -sig[jn];
      const double t2 = (sig[jn] > 0 ? 2.0 : -2.0) * mg[jn];
#pragma unroll
      for (int r = 0; r < M; ++r) v[r] = fma(t2, hcol[jn][r], v[r]);
    }
  }
}

// weight and magnitudes of one orbit of a support-major lane
template <int S>
struct OrbitWm { double w; double mg[S]; };
template <int S>
__device__ __forceinline__ void orbit_load_wm(const OrbitDev& ob, const uint32_t boff, OrbitWm<S>& r) {
  r.w = orbit_ld(ob.w, boff);
#pragma unroll
  for (int j = 0; j < S; ++j) r.mg[j] = orbit_ld(ob.mag + (size_t)j * ob.norb_p, boff);
}

// SUPPORT-MAJOR class (s <= 3; orbits.hpp): a lane walks the G orbits of its support with the support's columns of H in
// registers, sums their scaled sign-weighted sums in registers and adds the result to the accumulators once -- G times
// fewer LDS atomics and column reads than one orbit per lane.  The next orbit's weight and magnitudes are requested while
// the current one is scaled into the sums.  Local tile tl of the class; t0, t1, tfirst wave-uniform.
template <int M, int S, bool FULL, bool SIGNED>
__device__ __forceinline__ void orbit_class_grouped(const OrbitDev& ob, const int lc, const int t0, const int t1, const int tfirst,
                                                    const uint32_t lane8, const double* Hl, double* accl, const double (&su0)[M],
                                                    const double (&sg)[M], const double k0, double& m0) {
  static_assert(M * S <= 36, "support-major classes keep the support's columns in registers");
  constexpr int NV = 2 * S + S * (S - 1) / 2;
  const int G = ob.cgrp[S];
  const uint32_t gstride8 = (uint32_t)ob.cstride[S] * 8u;
  for (int t = t0; t < t1; ++t) {
    const uint32_t boff0 = ((uint32_t)ob.cbase[S] + (uint32_t)(t - tfirst) * 64u) * 8u + lane8;
    const uint64_t cpk = orbit_ld(ob.cpk, boff0), rpk = orbit_ld(ob.rpk, boff0);
    OrbitWm<S> cur;
    orbit_load_wm<S>(ob, boff0, cur);
    int c[S];
    double hcol[S][M];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      c[j] = (int)((cpk >> (8 * j)) & 255u);
      const double* hp = Hl + c[j] * orbit_hstride(M);
#pragma unroll
      for (int r = 0; r < M; ++r) hcol[j][r] = hp[r];
    }
    double acc[FULL ? NV : 1];
#pragma unroll
    for (int q = 0; q < (FULL ? NV : 1); ++q) acc[q] = 0.0;
    for (int g = 0; g < G; ++g) {
      double mg[S];
#pragma unroll
      for (int j = 0; j < S; ++j) mg[j] = cur.mg[j];
      const double w = cur.w;
      double E0, Eij[S * (S - 1) / 2 + 1], Oi[S];
      orbit_gray<M, S, FULL, SIGNED>(hcol, mg, su0, sg, k0, E0, Eij, Oi);
      // the next orbit of the lane (the last one requests itself again: no branch)
      const uint32_t bn = boff0 + (uint32_t)(g + 1 < G ? g + 1 : g) * gstride8;
      __builtin_amdgcn_sched_barrier(0);
      orbit_load_wm<S>(ob, bn, cur);
      __builtin_amdgcn_sched_barrier(0);
      const double wp = w + w;
      m0 = fma(wp, E0, m0);
      if constexpr (FULL) {
        const double w4 = wp + wp;
        int n = 0, e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          const double wm = wp * mg[i];
          acc[n] = fma(w4 * mg[i], Oi[i], acc[n]); ++n;
          acc[n] = fma(wm * mg[i], E0, acc[n]); ++n;
#pragma unroll
          for (int j = i + 1; j < S; ++j) { acc[n] = fma(wm * mg[j], Eij[e], acc[n]); ++n; ++e; }
        }
      }
    }
    if constexpr (FULL && S > GVI_EXP_NOATOM) {
      const int sh = lc + 3;
      char* const base = (char*)accl;
      char* const base1 = base + (1u << sh);
      unsigned A[S], B[S];
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const unsigned R = ((unsigned)rpk >> (10 * i)) & 1023u;          // (S <= 3: the low word)
        A[i] = R << sh;
        B[i] = (unsigned)c[i] << sh;
      }
      int n = 0;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        lds_add_f64(ORBIT_ADD_TARGET(base1 + B[i], n, sh), acc[n]); ++n;
        lds_add_f64(ORBIT_ADD_TARGET(base + (A[i] + B[i]), n, sh), acc[n]); ++n;
#pragma unroll
        for (int j = i + 1; j < S; ++j) { lds_add_f64(ORBIT_ADD_TARGET(base + (A[i] + B[j]), n, sh), acc[n]); ++n; }
      }
    } else if constexpr (FULL) {
      double tsum = 0.0;
#pragma unroll
      for (int q = 0; q < NV; ++q) tsum += acc[q];
      m0 = fma(1e-300, tsum, m0);
    }
  }
}

// the tiles [t0, t1) of ONE support-size class, software-pipelined: the records of tile t + 1 are requested while tile t
// still has its accumulator adds to issue, so that their round trip overlaps the adds and the next tile's column reads
// largest support size stored support-major (host: build_orbits' group_smax must not exceed it)
#ifndef ORBIT_GROUP_SMAX
#define ORBIT_GROUP_SMAX 3
#endif
#ifndef ORBIT_PIPE_SMAX
#define ORBIT_PIPE_SMAX 3
#endif
// t0, t1 are wave-uniform (SGPRs); lane8 = 8 * lane.  (Requesting the chunk's first tile ahead of orbit_wave's prologue was
// tried: the record stays live through every class loop -- 14 registers, spills -- for no gain.)
template <int M, int S, bool FULL, bool SIGNED>
__device__ __forceinline__ void orbit_class(const OrbitDev& ob, const int lc, const int t0, const int t1, const int tfirst, const uint32_t lane8,
                                            const double* Hl, double* accl, const double (&su0)[M], const double (&sg)[M], const double k0,
                                            double& m0) {
  if (t0 >= t1) return;
  // s >= 4: a tile is long enough (290 VALU instructions and more) for its own load latency not to matter, and its walk
  // leaves no 14 registers for the next tile's records
  constexpr bool PIPE = S <= ORBIT_PIPE_SMAX;
  OrbitRec<S> cur;
  const uint32_t cb8 = ((uint32_t)ob.cbase[S] - (uint32_t)tfirst * 64u) * 8u + lane8;      // entry of (tile t, this lane) = cb8 + 512 t
  if constexpr (PIPE) orbit_load<S>(ob, cb8 + (uint32_t)t0 * 512u, cur);
  for (int t = t0; t < t1; ++t) {
    if constexpr (!PIPE) orbit_load<S>(ob, cb8 + (uint32_t)t * 512u, cur);
    const int tn = t + 1 < t1 ? t + 1 : t;                      // (the last tile requests itself again: no branch)
    // the next tile's records go straight into cur: by the time pre() runs the walk has consumed the current ones
    auto pre = [&]() {
      if constexpr (PIPE) {
        __builtin_amdgcn_sched_barrier(0);
        orbit_load<S>(ob, cb8 + (uint32_t)tn * 512u, cur);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const uint64_t cpk = cur.cpk, rpk = cur.rpk;
    const double w = cur.w;
    double mg[S];
#pragma unroll
    for (int j = 0; j < S; ++j) mg[j] = cur.mg[j];
    if constexpr (M == 12 && S >= 4 && FULL && GVI_ORBIT_SPLIT12 != 0) orbit_walk_split<S, SIGNED>(lc, cpk, rpk, mg, w, Hl, accl, su0, sg, k0, m0, pre);
    else orbit_walk<M, S, FULL, SIGNED>(lc, cpk, rpk, mg, w, Hl, accl, su0, sg, k0, m0, pre);
  }
}

// out: where the chunk's partial sums go -- the set's partial array (stand-alone launches) or LDS (factor_fused_kernel).
// FRESH: the psi operands were written earlier in THIS launch by another wave of the block (factor_fused_kernel): u0 must
// then come through the vector path (a wave-uniform address would otherwise be served by the scalar cache, which stores of
// this launch do not update)
// Hpre / u0pre (LDS, optional): the psi operands of this factor as another wave of the block left them (column stride
// orbit_hstride(M)); nothing is read from a.H / a.u0 then
template <int M, int SMAX, bool FULL, bool SIGNED, bool FRESH = false>
__device__ __forceinline__ void orbit_wave(const OrbitArgs& a, const int k, const int chunk, double* lds, double* out,
                                           const double* Hpre = nullptr, const double* u0pre = nullptr) {
  const OrbitDev& ob = a.ob;
  const int lane = threadIdx.x & 63, d = a.d;
#ifdef GVI_FUSED_TIMING
  const long long wt_start = clock64();
#endif
  const int NP = FULL ? (d + 1) * (d + 2) / 2 : 1;
  int tb, te;
  if (ob.nb) { tb = ob.bnd[chunk]; te = ob.bnd[chunk + 1]; }
  else { tb = ob.bounds[chunk]; te = ob.bounds[chunk + 1]; }
  tb = __builtin_amdgcn_readfirstlane(tb);       // (wave-uniform: the tile loops run on the scalar unit)
  te = __builtin_amdgcn_readfirstlane(te);
  const uint32_t lane8 = (uint32_t)lane * 8u;
  const double* Hl = Hpre ? Hpre : lds;          // [d][M]: column c of H = the M operands of coordinate c
  // [NP][C] moment accumulators of this (factor, chunk): C private copies per entry, a lane adds to copy lane % C.  A
  // ds_add_f64 whose lanes hit one address costs ~3 cycles per lane (64-way: 192 cycles, tools/ubench/lds_atomic.hip);
  // with the copies and the strided orbit order (orbits.hpp) a wave instruction stays near the 8-cycle floor.
  const int C = a.copies, lc = __builtin_ctz((unsigned)C);      // (a power of two: launch_orbit / gvi_set_option)
  double* accl = lds + d * orbit_hstride(M);
  if (!Hpre) {
    const double* Hg = a.H + (size_t)k * M * d;  // stored [d][M] by the prep kernel
    for (int e = lane; e < d * M; e += 64) lds[(e / M) * orbit_hstride(M) + e % M] = Hg[e];
  }
  if (FULL)
    for (int e = lane; e < NP * C; e += 64) accl[e] = 0.0;
  double* accme = accl + (lane & (C - 1));
  double su0[M], sg[M], k0 = 0.0;
#pragma unroll
  for (int r = 0; r < M; ++r) {
    double u;
    if (u0pre) {                                   // wave-uniform value through LDS: back into SGPRs
      const double t = u0pre[r];
      u = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)), __builtin_amdgcn_readfirstlane(__double2loint(t)));
    } else u = FRESH ? __builtin_nontemporal_load(a.u0 + (size_t)k * M + r) : a.u0[(size_t)k * M + r];
    sg[r] = SIGNED ? a.sgn[(size_t)k * M + r] : 1.0;
    su0[r] = SIGNED ? sg[r] * u : u;               // unsigned: u0 itself -- a wave-uniform scalar load, it stays in SGPRs
    k0 = fma(su0[r], u, k0);
  }
  wave_lds_sync();
  double m0 = 0.0;
#ifdef GVI_FUSED_TIMING
  // shader-clock cycles of this wave per tile class (timing build): [prologue | s = 1..4 | reduction] + tile counts
  long long wt_last = clock64(), wt_acc[6] = {0, 0, 0, 0, 0, 0}, wt_n[5] = {0, 0, 0, 0, 0};
  wt_acc[0] = wt_last - wt_start;
#define ORBIT_CLASS_STAMP(S_, n_) do { if ((S_) <= 4) { const long long now = clock64(); wt_acc[S_] += now - wt_last; wt_n[S_] += (n_); wt_last = now; } } while (0)
#else
#define ORBIT_CLASS_STAMP(S_, n_) do { } while (0)
#endif
  // class after class (descending support size); the classes' tile ranges are kernel arguments, tile t starts at orbit 64 t
  int tcur = tb;
#define ORBIT_RUN_CLASS(S_)                                                                           \
  {                                                                                                   \
    const int e_ = te < ob.cend[S_] ? te : ob.cend[S_];                                                \
    if (tcur < e_) {                                                                                  \
      if constexpr (S_ <= ORBIT_GROUP_SMAX && M * S_ <= 36) orbit_class_grouped<M, S_, FULL, SIGNED>(ob, lc, tcur, e_, ob.cend[S_ + 1], lane8, Hl, accme, su0, sg, k0, m0); \
      else orbit_class<M, S_, FULL, SIGNED>(ob, lc, tcur, e_, ob.cend[S_ + 1], lane8, Hl, accme, su0, sg, k0, m0);            \
      ORBIT_CLASS_STAMP(S_, e_ - tcur); tcur = e_;                                                     \
    }                                                                                                 \
  }
  if constexpr (SMAX > 4) {
    ORBIT_RUN_CLASS(6)
    ORBIT_RUN_CLASS(5)
  }
  ORBIT_RUN_CLASS(4)
  ORBIT_RUN_CLASS(3)
  ORBIT_RUN_CLASS(2)
  ORBIT_RUN_CLASS(1)
#undef ORBIT_RUN_CLASS
#undef ORBIT_CLASS_STAMP
  m0 = wave_sum_f64(m0);
  if (chunk == 0) m0 = fma(ob.w0, k0, m0);       // the origin: psi(0) = sum_r s_r u0_r^2
  wave_lds_sync();
  if (lane == 0) out[0] = m0;
  if (FULL) {
    // copies in a fixed, lane-rotated order: with every lane starting at copy 0 each fourth lane sits on the same banks.
    // The usual eight copies: all reads of an entry in flight before the first add (with the run-time copy count the loop
    // below is eight DEPENDENT LDS round trips per entry: 4400 cycles of a wave's 30 000 at (12,5), walk stamps)
    if (C == 8) {
      for (int e = 1 + lane; e < NP; e += 64) {
        int rot = (lane >> 2) & 7;
        // (opaque: the eight rotated addresses are otherwise hoisted out of the fused pass's item loop and kept -- spilled --
        // across the whole walk: 32 of its 56 bytes of scratch per lane)
        asm volatile("" : "+v"(rot));
        double c8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c8[q] = accl[e * 8 + ((q + rot) & 7)];
        double t = c8[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += c8[q];
        out[e] = t;
      }
    } else {
      for (int e = 1 + lane; e < NP; e += 64) {
        const int rot = (lane >> 2) & (C - 1);
        double t = accl[e * C + rot];
        for (int q = 1; q < C; ++q) t += accl[e * C + ((q + rot) & (C - 1))];
        out[e] = t;
      }
    }
  }
#ifdef GVI_FUSED_TIMING
  if (gvi_walk_stamps && lane == 0 && (blockIdx.x % 146) == 0) {
    wt_acc[5] = clock64() - wt_last;
    unsigned long long* ws = gvi_walk_stamps + (((blockIdx.x / 146) * 4 + (threadIdx.x >> 6)) * 2 + (d == 12 ? 0 : 1)) * 12;
    for (int i = 0; i < 6; ++i) ws[i] = (unsigned long long)wt_acc[i];
    for (int i = 1; i < 5; ++i) ws[6 + i] = (unsigned long long)wt_n[i];
  }
#endif
}

template <bool FULL>
__device__ __forceinline__ double* orbit_out(const OrbitArgs& a, const int k, const int chunk) {
  return a.partial + ((size_t)k * a.nchunk + chunk) * (FULL ? (a.d + 1) * (a.d + 2) / 2 : 1);
}

// LDS doubles per wave
__host__ __device__ inline int orbit_lds_doubles(int d, int M, int copies) {
  return (d * orbit_hstride(M) + copies * (d + 1) * (d + 2) / 2 + 1) & ~1;      // even: every wave's H stays 16-byte aligned
}

// grid (ceil(K / 4), nchunk) x 256: four waves = four factors on the same chunk of orbit tiles.
// SMAX: largest support instantiated (4: degree <= 5; 6: degree <= 7); SIGNED: some sgn entry is not +1;
// WAVES: occupancy the register budget is cut for
template <int M, int SMAX, bool FULL, bool SIGNED, int WAVES>
__global__ __launch_bounds__(256, WAVES) void moments_orbit_kernel(OrbitArgs a) {
  extern __shared__ double sm[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = (int)blockIdx.x * 4 + wave;
  if (pred_skip(a.pred, a.pred_val)) return;
  if (k >= a.K) return;                          // no block-level barrier below
  orbit_wave<M, SMAX, FULL, SIGNED>(a, k, (int)blockIdx.y, sm + (size_t)wave * orbit_lds_doubles(a.d, M, a.copies),
                                    orbit_out<FULL>(a, k, (int)blockIdx.y));
}

// two sets in one launch (the chain pattern: binary priors + unary factors): blocks [0, nb0) serve set 0 as
// (bx, chunk) = (id % nbx0, id / nbx0), the rest set 1 likewise with nbx1
template <int M, int SMAX, bool FULL, bool SIGNED, int WAVES>
__global__ __launch_bounds__(256, WAVES) void moments_orbit_pair_kernel(OrbitArgs a0, OrbitArgs a1, int nbx0, int nb0, int nbx1) {
  extern __shared__ double sm[];
  if (pred_skip(a0.pred, a0.pred_val)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int id = (int)blockIdx.x;
  if (nbx1 < 0) {
    // stacked form (grid = max(nb0, nb1)): block b takes item b of set 0 and then item b of set 1, so the light set does not
    // run as a tail behind the single resident round of the heavy one
    const int nx1 = -nbx1;
    // ONE per-wave LDS region for both items (the waves of a block are not synchronised: a wave on its second item must
    // not reach into the region of a wave still on its first)
    const int l0 = orbit_lds_doubles(a0.d, M, a0.copies), l1 = orbit_lds_doubles(a1.d, M, a1.copies);
    double* mine = sm + (size_t)wave * (l0 > l1 ? l0 : l1);
    if (id < nb0) {
      const int k = (id % nbx0) * 4 + wave, chunk = id / nbx0;
      if (k < a0.K) orbit_wave<M, SMAX, FULL, SIGNED>(a0, k, chunk, mine, orbit_out<FULL>(a0, k, chunk));
    }
    const int k1 = (id % nx1) * 4 + wave, chunk1 = id / nx1;
    if (chunk1 < a1.nchunk && k1 < a1.K) orbit_wave<M, SMAX, FULL, SIGNED>(a1, k1, chunk1, mine, orbit_out<FULL>(a1, k1, chunk1));
    return;
  }
  const bool second = id >= nb0;
  if (second) id -= nb0;
  const int nbx = second ? nbx1 : nbx0;
  const int k = (id % nbx) * 4 + wave, chunk = id / nbx;
  if (!second) {
    if (k < a0.K) orbit_wave<M, SMAX, FULL, SIGNED>(a0, k, chunk, sm + (size_t)wave * orbit_lds_doubles(a0.d, M, a0.copies), orbit_out<FULL>(a0, k, chunk));
  } else {
    if (k < a1.K) orbit_wave<M, SMAX, FULL, SIGNED>(a1, k, chunk, sm + (size_t)wave * orbit_lds_doubles(a1.d, M, a1.copies), orbit_out<FULL>(a1, k, chunk));
  }
}

}  // namespace gvi
