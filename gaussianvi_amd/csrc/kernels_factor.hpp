// Per-factor device kernels of the NGD Gauss-Hermite hot path (gfx950, wave64, fp64).
//
//   prep_kernel      Sigma_k -> S_k = Sigma_k^{1/2}, S_k^{-1}, Lam_k = Sigma_k^{-1} (parallel Jacobi,
//                    one wave per factor) and the per-pass psi operands H_k = A_k S_k, u0_k
//                    [SparseGaussHermite::update_sigmapoints, quadrature/SparseGaussHermite.h:231-243;
//                     update_precision_from_joint, gvibase/GVIFactorizedBase.h:111-114]
//   moments_*        c_i = w_i psi(mu + S z_i) and the z-space moments  sum_i c_i [1, z_i, z_i z_i^T]
//                    [SparseGaussHermite::Integrate x3, quadrature/SparseGaussHermite.h:197-221;
//                     closures ngd/NGDFactorizedBaseGH.h:46-48].  Families:
//                      sreg / sreg_pair  sum-of-squares psi, operands from SGPRs, accumulators in registers (dominant)
//                      scost / _pair     cost pass (m0 only), two factors per wave
//                      split             d = 16 / 20 / 24: four waves per factor, 8-bit coded table
//                      reg / wide / tile policy-templated register kernels (other psi kinds; A/B variants)
//                      generic           any d <= 32, any psi kind, host psi
//                      closed            quadrature-free moments of a quadratic psi (NGDFactorizedLinear)
//   cost_tail_kernel per-factor cost + ordered sum + publish for a cost pass, one launch
//   jko_*            factor-level proximal (JKO) map around the prep kernel's Jacobi
//   epilogue_kernel  ordered sum of the chunk partials, back-transform to x-space, Vdmu_k / Vddmu_k
//                    [calculate_partial_V, ngd/NGDFactorizedBaseGH.h:53-74]
//   expand_kernel    X = mu + S z for the host-callback psi route.
//
// Formulation.  With x = mu + S z:  E[(x-mu) psi] = S m1,  E[(x-mu)(x-mu)^T psi] = S M2 S  where
// m1 = sum c_i z_i, M2 = sum c_i z_i z_i^T are accumulated in z-space, so the N x d x d expand GEMM of
// the reference disappears from the per-point work.  Because Lam S = S^{-1}:
//   Vdmu = S^{-1} m1 / T,   Vddmu = (S^{-1} M2 S^{-1} - Lam m0) / T.
// Sum-of-squares psi kinds (quadratic priors) are evaluated as psi = sum_r s_r (u0 + H z)_r^2 with
// H = A S folded per pass, A = diag(sqrt|e|) W^T [Phi, -I] from the eigen-decomposition of Qinv done
// once on the host at gvi_factors_add.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "device_common.hpp"

namespace gvi {

enum { KIND_RANGE_1D = 0, KIND_QUAD_PRIOR = 1, KIND_FIXED_PRIOR = 2, KIND_HOST_CALLBACK = 3, KIND_HINGE_SDF_2D = 4,
       KIND_HINGE_SDF_2D_BODY = 5, KIND_HINGE_SDF_3D = 6, KIND_HINGE_SDF_3D_ARM = 7 };

__host__ __device__ inline int npairs(int d) { return (d + 1) * (d + 2) / 2; }

// packed z-space moment layout: [0] m0 | [1..d] m1 | then upper triangle of M2 row by row
__host__ __device__ inline int pair_index(int d, int a, int b) {  // a <= b < d
  return 1 + d + a * d - a * (a - 1) / 2 + (b - a);
}

struct FactorDev {
  int K, d, m, kind;
  int64_t N, Np;            // points, padded to a multiple of 64 (pad: z = 0, w = 0)
  const double* Zt;         // [d][Np] dimension-major: lane-over-points loads are coalesced
  const double* w;          // [Np]
  const double* Zq;         // [Np / 64][d + 1][64] tile-major copy (row d = w): one contiguous block per 64-point step
  // Mirror-half table: a sparse Gauss-Hermite grid is symmetric -- with every point z it holds -z with the same weight --
  // so only one representative per +-pair is stored (first non-zero coordinate positive; the origin with HALF its weight)
  // and the kernels evaluate psi at z and -z from one load:  Zm [Nmp / 64][d + 1][64], Nm representatives, null if the
  // table is not symmetric.
  const double* Zm;
  int64_t Nm, Nmp;
  int all_pos;              // 1: every sgn entry is +1 (sum-of-squares kinds with a positive-definite weight)
  const uint32_t* codes;    // [d/4][Np] four 8-bit node codes per word (tables with <= 256 distinct values) or null
  const double* lut;        // [256] code -> node value
  const double* A;          // [K][m][d]   sum-of-squares kinds
  const double* b;          // [K][m]
  const double* sgn;        // [K][m]
  const double* raw;        // [K][raw_stride] raw parameter block (RANGE_1D)
  int raw_stride;
  const double* temperature;  // [K]
  // per-pass products of prep_kernel
  double* S;                // [K][d][d]
  double* Sinv;             // [K][d][d]
  double* Lam;              // [K][d][d]
  double* H;                // [K][d][m]  H = A S, stored column by column
  double* Hq;               // [K][4][d][R], R = ceil(m/4): rows v R .. v R + R of H per wave v (split kernel) or null
  double* u0;               // [K][m]
  const double* sdf;        // HINGE_SDF_2D: column-major rows x cols signed-distance grid
  int sdf_rows, sdf_cols, sdf_nz;
  double sdf_ox, sdf_oy, sdf_oz, sdf_cell;
  double sdf_inv_cell;                // 1 / sdf_cell: the look-ups multiply (an fp64 division is ~12 VALU instructions; two per 2-D look-up were 25 of the hinge kernel's 79)
  const double* arm;        // HINGE_SDF_3D_ARM: [ndof, ns, a[ndof], alpha[ndof], d[ndof], bias[ndof], frame[ns], centre[ns][3], radius[ns]]
  double jko_h;             // > 0: the third spectral output is the JKO map 1 / (l/2 + h + sqrt(l (l + 4h))/2) instead of 1/l
  double jtol;              // Jacobi stops when off^2 <= jtol * diag^2 (sums of squares)
  // 1: S is the Cholesky factor L of Sigma instead of its symmetric square root (prep_chol_body), the Sinv slot holds
  // L^-T.  Only for sum-of-squares psi on a degree >= 3 table, where the quadrature of psi {1, z, z z^T} is exact and
  // the moments do not depend on WHICH factor S S^T = Sigma maps the nodes (host: FactorSet::dev).
  int chol;
  double* Vws;              // [K][d][d] eigenvectors of the previous prep (warm start) or null
  int warm;                 // 1: start the Jacobi sweeps from Vws (resident NGD iteration only)
};

// ---------------------------------------------------------------------------------------------
// prep_kernel: one wave (block of 64) per factor.  Parallel-ordered (round-robin) cyclic Jacobi on
// the symmetric d x d block held in LDS; every round applies d/2 disjoint plane rotations at once:
//   A'_ij = al_i al_j A_ij + al_i be_j A_i,pj + be_i al_j A_pi,j + be_i be_j A_pi,pj ,
//   V'_ij = al_j V_ij + be_j V_i,pj           (p. = rotation partner, (al, be) = (c, -/+s)).
// ---------------------------------------------------------------------------------------------
// reciprocal / reciprocal square root: hardware seed + one Newton-Raphson step (~1e-16 relative)
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double rsq_nr(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  return r * fma(-0.5 * x * r, r, 1.5);
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// DT: d at compile time (the d-long inner products unroll and their loads -- LDS, and the GLOBAL rows of A_k in the H = A S
// product -- are issued together instead of one dependent round trip per term) or 0 for any d
template <int EPLP, int DT>   // EPLP: elements of the d x d block per lane: 1 (d <= 8), 4 (d <= 16), 16 (d <= 32)
// kin: index of the factor inside the (mu, Sigma) INPUT arrays (== k, or 0 when the caller staged this factor's
// marginal in LDS); outputs always go to slot k
__device__ __forceinline__ void prep_body(const FactorDev& f, const double* mu, const double* Sigma, int k, double* sm, int kin) {
  const int d = DT ? DT : f.d, dd = d * d, lane = threadIdx.x & 63;      // one wave per factor (the fused kernel runs several per block)
  const int dp = d + (d & 1);
  double* A0 = sm;
  double* A1 = A0 + dd;
  double* V0 = A1 + dd;
  double* V1 = V0 + dd;
  double* al = V1 + dd;          // [dp]
  double* be = al + dp;          // [dp]
  double* lam = be + dp;         // [d] (3 functions of lambda: sqrt, 1/sqrt, 1/x)
  int* pa = (int*)(lam + 3 * d); // [dp]
  int ei[EPLP], ej[EPLP];        // (row, col) of this lane's elements, computed once (-1: none)
#pragma unroll
  for (int q = 0; q < EPLP; ++q) {
    const int e = lane + q * 64;
    ei[q] = e < dd ? e / d : -1;
    ej[q] = e < dd ? e % d : 0;
  }
  const double* Sg = Sigma + (size_t)kin * dd;
  // Warm start (device-resident NGD iteration): consecutive proposals differ little, so the previous
  // eigenvectors W almost diagonalise the new block; sweeping A0 = W^T Sigma W from V0 = W needs 2-4
  // sweeps instead of 7-8.  Any orthogonal start gives the same decomposition up to rounding; a NaN
  // in W (left by a rejected non-PSD trial) falls back to the cold start.
  bool warm = f.warm != 0 && f.Vws != nullptr;
  if (warm) {
    int bad = 0;
#pragma unroll
    for (int q = 0; q < EPLP; ++q) {
      if (ei[q] >= 0) {
        const int i = ei[q], j = ej[q], e = lane + q * 64;
        const double v = f.Vws[(size_t)k * dd + e];
        bad |= !(fabs(v) <= 2.0);
        V0[e] = v;
        A1[e] = i >= j ? Sg[i * d + j] : Sg[j * d + i];
      }
    }
    warm = __ballot(bad) == 0;
    wave_lds_sync();
    if (warm) {
#pragma unroll
      for (int q = 0; q < EPLP; ++q) {              // T = Sigma W  -> V1
        if (ei[q] >= 0) {
          const int i = ei[q], j = ej[q];
          double t = 0.0;
#pragma unroll
          for (int c = 0; c < d; ++c) t += A1[i * d + c] * V0[c * d + j];
          V1[lane + q * 64] = t;
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < EPLP; ++q) {              // A0 = W^T T, upper triangle mirrored (exactly symmetric)
        if (ei[q] >= 0) {
          const int i = ei[q] <= ej[q] ? ei[q] : ej[q], j = ei[q] <= ej[q] ? ej[q] : ei[q];
          double t = 0.0;
#pragma unroll
          for (int c = 0; c < d; ++c) t += V0[c * d + i] * V1[c * d + j];
          A0[lane + q * 64] = t;
        }
      }
    }
  }
  if (!warm) {
#pragma unroll
    for (int q = 0; q < EPLP; ++q) {
      if (ei[q] >= 0) {
        const int i = ei[q], j = ej[q], e = lane + q * 64;
        A0[e] = i >= j ? Sg[i * d + j] : Sg[j * d + i];   // lower triangle, like SelfAdjointEigenSolver
        V0[e] = i == j ? 1.0 : 0.0;
      }
    }
  }
  wave_lds_sync();
  double* A = A0; double* An = A1; double* V = V0; double* Vn = V1;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, dg = 0.0;
#pragma unroll
    for (int q = 0; q < EPLP; ++q) {
      if (ei[q] >= 0) {
        const double v = A[lane + q * 64];
        if (ei[q] == ej[q]) dg += v * v; else if (ei[q] < ej[q]) off += v * v;
      }
    }
    off = wave_sum(off);
    dg = wave_sum(dg);
    if (off <= f.jtol * dg) break;     // false for NaN input: runs the (bounded) 40 sweeps
    for (int r = 0; r < dp - 1; ++r) {
      if (lane < dp / 2) {
        int p, q;
        if (lane == 0) { p = dp - 1; q = r; }
        else {
          p = r + lane; if (p >= dp - 1) p -= dp - 1;
          q = r - lane; if (q < 0) q += dp - 1;
        }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0;
        const bool valid = q < d;
        if (valid) {
          const double apq = A[p * d + q];
          if (apq != 0.0) {
            // fp64 div / sqrt sequences cost ~1 400 cycles per rotation on the critical path; hardware
            // rcp / rsq + Newton steps give the same rotation.  Only (c, s) must be orthonormal to
            // working precision (c gets two Newton steps); errors in theta / t merely perturb the
            // convergence rate.
            const double theta = (A[q * d + q] - A[p * d + p]) * rcp_nr(2.0 * apq);
            if (fabs(theta) < 1e150) {               // else the rotation is the identity to fp64
              const double w1 = fma(theta, theta, 1.0);
              const double sq = w1 * rsq_nr(w1);
              const double t = copysign(rcp_nr(fabs(theta) + sq), theta);
              const double w2 = fma(t, t, 1.0);
              double rc = rsq_nr(w2);
              rc = rc * fma(fma(-0.5 * w2, rc, 0.0), rc, 1.5);   // second Newton step
              c = rc;
              s = t * c;
            }
          }
        }
        al[p] = c; be[p] = -s; pa[p] = valid ? q : p;
        al[q] = c; be[q] = s;  pa[q] = valid ? p : q;
      }
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < EPLP; ++q) {
        if (ei[q] >= 0) {
          const int i = ei[q], j = ej[q], e = lane + q * 64;
          const int pi = pa[i], pj = pa[j];
          const double ai = al[i], bi = be[i], aj = al[j], bj = be[j];
          An[e] = ai * (aj * A[i * d + j] + bj * A[i * d + pj]) + bi * (aj * A[pi * d + j] + bj * A[pi * d + pj]);
          Vn[e] = aj * V[i * d + j] + bj * V[i * d + pj];
        }
      }
      wave_lds_sync();
      double* t = A; A = An; An = t;
      t = V; V = Vn; Vn = t;
    }
  }
  if (lane < d) {
    const double l = A[lane * d + lane];
    lam[lane] = sqrt(l);               // negative eigenvalue -> NaN, as the reference's operatorSqrt
    lam[d + lane] = 1.0 / sqrt(l);
    lam[2 * d + lane] = f.jko_h > 0.0 ? 1.0 / (0.5 * l + f.jko_h + 0.5 * sqrt(l * (l + 4.0 * f.jko_h))) : 1.0 / l;
  }
  wave_lds_sync();
  // S, S^-1, Lam = V f(lambda) V^T; S is also kept in LDS (An) for H = A_k S
#pragma unroll
  for (int q = 0; q < EPLP; ++q) {
    if (ei[q] >= 0) {
      const int i = ei[q], j = ej[q], e = lane + q * 64;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) {
        const double vv = V[i * d + c] * V[j * d + c];
        s0 += vv * lam[c]; s1 += vv * lam[d + c]; s2 += vv * lam[2 * d + c];
      }
      An[e] = s0;
      if (f.Vws) f.Vws[(size_t)k * dd + e] = V[e];
      f.S[(size_t)k * dd + e] = s0;
      f.Sinv[(size_t)k * dd + e] = s1;
      f.Lam[(size_t)k * dd + e] = s2;
    }
  }
  wave_lds_sync();
  if (f.m > 0) {
    const int m = f.m;
    const double* Ak = f.A + (size_t)k * m * d;
    for (int e = lane; e < m * d; e += 64) {
      const int r = e / d, a = e % d;
      double h = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) h += Ak[r * d + c] * An[c * d + a];
      f.H[(size_t)k * m * d + a * m + r] = h;            // column-major [d][m]: a column's m entries contiguous
      if (f.Hq) {
        const int R = (m + 3) / 4;
        f.Hq[(((size_t)k * 4 + r / R) * d + a) * R + r % R] = h;
      }
    }
    if (lane < m) {
      double u = f.b[(size_t)k * m + lane];
#pragma unroll
      for (int c = 0; c < d; ++c) u += Ak[lane * d + c] * mu[(size_t)kin * d + c];
      f.u0[(size_t)k * m + lane] = u;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// prep_chol_body: the per-pass products from a CHOLESKY factor, one wave per factor, rows in registers.
//
// The reference maps the sigma points with the symmetric square root of Sigma (gvibase/GVIFactorizedBaseGH.h:
// updateGH -> SparseGaussHermite::update_sigmapoints, an Eigen self-adjoint eigen-decomposition) -- here a cyclic
// Jacobi solve of ~15 us per pass at d = 12.  For psi = sum_r s_r (A x + b)_r^2 the integrands psi, z psi, z z^T psi are
// polynomials of degree <= 4 in z, which a sparse Gauss-Hermite rule of degree >= 3 integrates EXACTLY: the moments
// E[psi], E[(x - mu) psi], E[(x - mu)(x - mu)^T psi] are then the same for every S with S S^T = Sigma (they are
// functions of (mu, Sigma) alone), and the back-transform only needs S^-T and Sigma^-1:
//     Vdmu = S^-T m1 / T,   Vddmu = (S^-T M2 S^-1 - Sigma^-1 m0) / T.
// So for these sets S = L (Sigma = L L^T): 66 multiply-adds per triangle at d = 12 instead of Jacobi sweeps, and the
// results agree with the symmetric-root route to rounding (tests/test_gpu_parity.py A/Bs the two; option "chol_sqrt").
// Every psi kind that is NOT a polynomial of degree <= 2 (hinge / range factors, host-evaluated psi, gvi_expand) keeps
// the symmetric root: there the node positions matter.  A non-positive pivot gives NaN exactly where a negative
// eigenvalue did (the line search rejects such a trial either way).
// lane i < d owns row i of the lower triangle; pivots / multipliers travel as wave-uniform scalars (v_readlane).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double chol_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// -DGVI_FUSED_TIMING: 100 MHz stamps of wave 0 of block 0 inside the Cholesky prep (slots 8.. of the fused kernel's stamp
// buffer; the kernel parks the pointer here)
#ifdef GVI_FUSED_TIMING
__device__ unsigned long long* gvi_prep_stamps;
#define PREP_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && gvi_prep_stamps) gvi_prep_stamps[(i) + 16 * (threadIdx.x >> 6)] = t_; } while (0)
#else
#define PREP_STAMP(i) do { } while (0)
#endif

// Zs (LDS, optional): a copy of S^-T for an epilogue in the same launch (factor_fused_kernel)
// Hs / u0s (LDS, optional; factor_fused_kernel's lean form): the psi operands H (column c at Hs + c hstride) and u0 stay in LDS
// for the walk of the same launch and NOTHING of the per-pass products goes to memory (no S / S^-T / Sigma^-1 / H / u0
// stores, Sigma^-1 is not formed: the caller's epilogue re-forms what it needs from Zs)
// l_ready (LDS, optional; lean form only): ANOTHER wave of the block forms H = A L and u0 = b + A mu for this item
// (prep_chol_hu0) -- this wave then skips the rows of A altogether and raises *l_ready once L is in its LDS area
template <int DT>
__device__ inline void prep_chol_body(const FactorDev& f, const double* mu, const double* Sigma, int k, double* sm, int kin, double* Zs = nullptr,
                                      double* Hs = nullptr, double* u0s = nullptr, const int hstride = 0, int* l_ready = nullptr) {
  const bool lean = Hs != nullptr;
  constexpr int d = DT, dd = DT * DT;
  const int lane = threadIdx.x & 63;      // one wave per factor (the fused kernel runs several per block)
  double* Ll = sm;            // [d][d] L, zeros above the diagonal
  double* Xl = Ll + dd;       // [d][d] X = L^-1
  const double* Sg = Sigma + (size_t)kin * dd;
  const int li = lane < d ? lane : d - 1;                     // lanes >= d shadow the last row (results unused)
  // The rows of A (and b) the psi operands H = A L, u0 = b + A mu need do not depend on the factorisation: request them
  // NOW, so that their global-memory latency runs under the Cholesky instead of behind it (the prep is a chain of dependent
  // latencies: ~10 us per factor before this, profiles/r03 fused-pass stamps).  Element e = lane (+ 64) of H reads row e / d.
  constexpr int HPL = 2;                                       // H elements per lane: m d <= 128 for the Cholesky shapes (m <= d / 2 ... d)
  const int mm = f.m;
  const bool pre = mm > 0 && mm * d <= 64 * HPL && !l_ready;
  double arow[HPL][DT], brow = 0.0;
  if (pre) {
    const double* Ak = f.A + (size_t)k * mm * d;
#pragma unroll
    for (int q = 0; q < HPL; ++q) {
      const int e = lane + 64 * q;
      const int r = e < mm * d ? e / d : 0;
#pragma unroll
      for (int c = 0; c < d; ++c) arow[q][c] = Ak[r * d + c];
    }
    if (lane < mm) brow = f.b[(size_t)k * mm + lane];
  }
  PREP_STAMP(0);
  double row[DT];
#pragma unroll
  for (int c = 0; c < d; ++c) row[c] = c <= li ? Sg[li * d + c] : 0.0;      // lower triangle, like SelfAdjointEigenSolver
  PREP_STAMP(1);
  double inv[DT];
#pragma unroll
  for (int j = 0; j < d; ++j) {
    const double pj = chol_readlane(row[j], j);
    double r = __builtin_amdgcn_rsq(pj);                      // 1 / sqrt(pj): NaN for a negative pivot
    r = r * fma(-0.5 * pj * r, r, 1.5);
    r = r * fma(-0.5 * pj * r, r, 1.5);
    inv[j] = r;
    row[j] = li == j ? pj * r : (li > j ? row[j] * r : 0.0);
#pragma unroll
    for (int c = j + 1; c < d; ++c) {
      const double lcj = chol_readlane(row[j], c);
      row[c] = li >= c ? fma(-row[j], lcj, row[c]) : 0.0;
    }
  }
  PREP_STAMP(2);
  // X = L^-1 by forward substitution: lane c owns column c.  Column-oriented: x_i updates the right-hand sides of every
  // later row at once, so the dependent chain is d (multiply + one FMA) instead of the d (d - 1) / 2 FMAs of the row-wise
  // dot products -- the same FMAs in the same order per entry (bit-identical), 66 -> 24 instructions deep at d = 12
  // (in place: entry i holds the right-hand side until x_i is formed -- no second array: the 24 registers of one pushed the
  // fused pass from 20 to 56 bytes of scratch per lane, 14 MB of spill traffic per launch)
  double xcol[DT];
#pragma unroll
  for (int i = 0; i < d; ++i) xcol[i] = li == i ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < d; ++i) {
    xcol[i] = li <= i ? xcol[i] * inv[i] : 0.0;
#pragma unroll
    for (int j = i + 1; j < d; ++j) xcol[j] = fma(-chol_readlane(row[i], j), xcol[i], xcol[j]);
  }
  PREP_STAMP(3);
  if (lane < d) {
#pragma unroll
    for (int c = 0; c < d; ++c) { Ll[lane * d + c] = row[c]; Xl[c * d + lane] = xcol[c]; }
  }
  // the prefetched rows of A also go to LDS (one copy per row: lanes e = r d hold row r), for u0 = b + A mu below
  double* Al = Xl + dd;        // [m][d]
  if (pre) {
#pragma unroll
    for (int q = 0; q < HPL; ++q) {
      const int e = lane + 64 * q;
      if (e < mm * d && e % d == 0) {
#pragma unroll
        for (int c = 0; c < d; ++c) Al[e + c] = arow[q][c];
      }
    }
  }
  wave_lds_sync();
  if (l_ready && lane == 0) __hip_atomic_store(l_ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of a wave execute in order: L is there)
  if (lean) {
    for (int e = lane; e < dd; e += 64) Zs[e] = Xl[(e % d) * d + e / d];            // S^-T = X^T
  } else {
    for (int e = lane; e < dd; e += 64) {
      const int i = e / d, j = e % d;
      double lam = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) lam = fma(Xl[c * d + i], Xl[c * d + j], lam);      // Sigma^-1 = X^T X
      f.S[(size_t)k * dd + e] = Ll[e];
      f.Sinv[(size_t)k * dd + e] = Xl[j * d + i];                                      // S^-T = X^T
      f.Lam[(size_t)k * dd + e] = lam;
      if (Zs) Zs[e] = Xl[j * d + i];
    }
  }
  PREP_STAMP(4);
  if (f.m > 0 && !l_ready) {
    const int m = f.m;
    const double* Ak = f.A + (size_t)k * m * d;
    if (pre) {                                           // operands already in registers (same products, same order)
#pragma unroll
      for (int q = 0; q < HPL; ++q) {
        const int e = lane + 64 * q;
        if (e < m * d) {
          const int r = e / d, a = e % d;
          double h = 0.0;
#pragma unroll
          for (int c = 0; c < d; ++c) h += arow[q][c] * Ll[c * d + a];
          if (lean) Hs[a * hstride + r] = h;             // the walk's LDS layout: column a at stride hstride
          else {
            f.H[(size_t)k * m * d + a * m + r] = h;      // column-major [d][m]: a column's m entries contiguous
            if (f.Hq) {
              const int R = (m + 3) / 4;
              f.Hq[(((size_t)k * 4 + r / R) * d + a) * R + r % R] = h;
            }
          }
        }
      }
    } else {
      for (int e = lane; e < m * d; e += 64) {
        const int r = e / d, a = e % d;
        double h = 0.0;
#pragma unroll
        for (int c = 0; c < d; ++c) h += Ak[r * d + c] * Ll[c * d + a];
        f.H[(size_t)k * m * d + a * m + r] = h;
        if (f.Hq) {
          const int R = (m + 3) / 4;
          f.Hq[(((size_t)k * 4 + r / R) * d + a) * R + r % R] = h;
        }
      }
    }
    if (lane < m) {
      double u = pre ? brow : f.b[(size_t)k * m + lane];
#pragma unroll
      for (int c = 0; c < d; ++c) u += (pre ? Al[lane * d + c] : Ak[lane * d + c]) * mu[(size_t)kin * d + c];
      if (lean) u0s[lane] = u;
      else f.u0[(size_t)k * m + lane] = u;
    }
  }
  PREP_STAMP(5);
}

// The psi operands H = A L and u0 = b + A mu of an item whose Cholesky factor ANOTHER wave of the block is forming
// (factor_fused_kernel: two of a block's four waves are idle in the products phase).  u0 does not depend on the factorisation
// and is formed at once; the rows of A are requested at once too; H waits for *l_ready (raised by prep_chol_body once L is in
// that wave's LDS area Ll).  Same products in the same order as prep_chol_body's own H / u0: bit-identical.
// mul: the item's mean (LDS or global, [d] at index 0); own: this wave's LDS area (>= m d + d doubles); 0 < m d <= 128 (caller).
template <int DT>
__device__ inline void prep_chol_hu0(const FactorDev& f, const double* mul, const int k, double* own, const double* Ll, int* l_ready,
                                     double* Hs, double* u0s, const int hstride) {
  constexpr int d = DT, HPL = 2;
  const int lane = threadIdx.x & 63, m = f.m;
  const double* Ak = f.A + (size_t)k * m * d;
  double arow[HPL][DT], brow = 0.0;
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int e = lane + 64 * q;
    const int r = e < m * d ? e / d : 0;
#pragma unroll
    for (int c = 0; c < d; ++c) arow[q][c] = Ak[r * d + c];
  }
  if (lane < m) brow = f.b[(size_t)k * m + lane];
  double* Al = own;            // [m][d]
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int e = lane + 64 * q;
    if (e < m * d && e % d == 0) {
#pragma unroll
      for (int c = 0; c < d; ++c) Al[e + c] = arow[q][c];
    }
  }
  wave_lds_sync();
  if (lane < m) {
    double u = brow;
#pragma unroll
    for (int c = 0; c < d; ++c) u += Al[lane * d + c] * mul[c];
    u0s[lane] = u;
  }
  while (__hip_atomic_load(l_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int e = lane + 64 * q;
    if (e < m * d) {
      const int r = e / d, a = e % d;
      double h = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) h += arow[q][c] * Ll[c * d + a];
      Hs[a * hstride + r] = h;
    }
  }
}

// The Cholesky route for every other dimension (d <= 32): L, X = L^-1 in LDS, any d at run time.  The register form above
// exists for the chain shapes; a d = 28 prior set (the arm graph, gp/minimum_acc_prior.h with 14-dimensional states) fell to
// the symmetric-root Jacobi solve -- 0.6 ms per prep launch, 74 % of that graph's iteration.  One wave per factor:
// right-looking factorisation with the trailing update spread over the lanes (EPLP elements each), forward substitution with
// lane = column, then the same outputs as prep_chol_body's non-lean form (S = L, S^-T, Sigma^-1 = X^T X, H = A L, u0).
template <int EPLP>
__device__ inline void prep_chol_lds_body(const FactorDev& f, const double* mu, const double* Sigma, int k, double* sm, int kin, double* Zs) {
  const int d = f.d, dd = d * d, lane = threadIdx.x & 63;
  double* Ll = sm;             // [d][d] row-major, zeros above the diagonal
  double* Xl = Ll + dd;        // [d][d] X = L^-1 (row-major, lower triangular)
  const double* Sg = Sigma + (size_t)kin * dd;
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    Ll[e] = j <= i ? Sg[i * d + j] : 0.0;        // lower triangle, like SelfAdjointEigenSolver
    Xl[e] = 0.0;
  }
  wave_lds_sync();
  for (int j = 0; j < d; ++j) {
    const double pj = Ll[j * d + j];
    double r = __builtin_amdgcn_rsq(pj);          // NaN for a negative pivot, as the register form
    r = r * fma(-0.5 * pj * r, r, 1.5);
    r = r * fma(-0.5 * pj * r, r, 1.5);
    wave_lds_sync();                              // (every lane has read the pivot before column j is rescaled)
    if (lane >= j && lane < d) Ll[lane * d + j] = lane == j ? pj * r : Ll[lane * d + j] * r;
    if (lane == j) Xl[j * d + j] = r;             // 1 / L_jj parked on X's diagonal
    wave_lds_sync();
    for (int e = lane; e < dd; e += 64) {         // trailing update of the lower triangle: L_ic -= L_ij L_cj, j < c <= i
      const int i = e / d, c = e % d;
      if (c > j && c <= i) Ll[e] = fma(-Ll[i * d + j], Ll[c * d + j], Ll[e]);
    }
    wave_lds_sync();
  }
  // X = L^-1 by forward substitution, lane c = column c: x_i = (delta_ic - sum_{c <= q < i} L_iq x_q) / L_ii
  if (lane < d) {
    const int c = lane;
    for (int i = c; i < d; ++i) {
      double acc = i == c ? 1.0 : 0.0;
      for (int q = c; q < i; ++q) acc = fma(-Ll[i * d + q], Xl[q * d + c], acc);
      const double inv = i == c ? Xl[c * d + c] : Xl[i * d + i];       // (the diagonal still holds 1 / L_ii until row i is written)
      Xl[i * d + c] = acc * inv;
    }
  }
  wave_lds_sync();
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    double lam = 0.0;
    const int lo = i > j ? i : j;
    for (int c = lo; c < d; ++c) lam = fma(Xl[c * d + i], Xl[c * d + j], lam);      // Sigma^-1 = X^T X (X lower triangular)
    f.S[(size_t)k * dd + e] = Ll[e];
    f.Sinv[(size_t)k * dd + e] = Xl[j * d + i];                                      // S^-T = X^T
    f.Lam[(size_t)k * dd + e] = lam;
    if (Zs) Zs[e] = Xl[j * d + i];
  }
  if (f.m > 0) {
    const int m = f.m;
    const double* Ak = f.A + (size_t)k * m * d;
    for (int e = lane; e < m * d; e += 64) {
      const int r = e / d, a = e % d;
      double h = 0.0;
      for (int c = a; c < d; ++c) h = fma(Ak[r * d + c], Ll[c * d + a], h);           // (L lower triangular: rows c >= a)
      f.H[(size_t)k * m * d + a * m + r] = h;
      if (f.Hq) {
        const int R = (m + 3) / 4;
        f.Hq[(((size_t)k * 4 + r / R) * d + a) * R + r % R] = h;
      }
    }
    if (lane < m) {
      double u = f.b[(size_t)k * m + lane];
      for (int c = 0; c < d; ++c) u = fma(Ak[lane * d + c], mu[(size_t)kin * d + c], u);
      f.u0[(size_t)k * m + lane] = u;
    }
  }
}

// the chain shapes of BASELINE.json get unrolled instances
// true when prep_body_d takes the Cholesky route for this set (and can leave [S^-T | Sigma^-1] in LDS)
__device__ __forceinline__ bool prep_is_chol(const FactorDev& f) {
  return f.chol && (f.d == 2 || f.d == 4 || f.d == 6 || f.d == 8 || f.d == 12);
}

template <int EPLP>
__device__ inline void prep_body_d(const FactorDev& f, const double* mu, const double* Sigma, int k, double* sm, int kin, double* Zs = nullptr) {
  if (f.chol) {
    switch (f.d) {
      case 2: prep_chol_body<2>(f, mu, Sigma, k, sm, kin, Zs); return;
      case 4: prep_chol_body<4>(f, mu, Sigma, k, sm, kin, Zs); return;
      case 6: prep_chol_body<6>(f, mu, Sigma, k, sm, kin, Zs); return;
      case 8: prep_chol_body<8>(f, mu, Sigma, k, sm, kin, Zs); return;
      case 12: prep_chol_body<12>(f, mu, Sigma, k, sm, kin, Zs); return;
      default: prep_chol_lds_body<EPLP>(f, mu, Sigma, k, sm, kin, Zs); return;
    }
  }
  if (EPLP == 1 && f.d == 2) prep_body<EPLP, 2>(f, mu, Sigma, k, sm, kin);
  else if (EPLP == 1 && f.d == 4) prep_body<EPLP, 4>(f, mu, Sigma, k, sm, kin);
  else if (EPLP == 1 && f.d == 6) prep_body<EPLP, 6>(f, mu, Sigma, k, sm, kin);
  else if (EPLP == 4 && f.d == 6) prep_body<EPLP, 6>(f, mu, Sigma, k, sm, kin);
  else if (EPLP == 4 && f.d == 12) prep_body<EPLP, 12>(f, mu, Sigma, k, sm, kin);
  else prep_body<EPLP, 0>(f, mu, Sigma, k, sm, kin);
}

template <int EPLP>
__global__ __launch_bounds__(64) void prep_kernel(FactorDev f, const double* __restrict__ mu,
                                                  const double* __restrict__ Sigma) {
  extern __shared__ double sm[];
  prep_body_d<EPLP>(f, mu, Sigma, blockIdx.x, sm, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Proximal (JKO) update, factor level: ProxGVIFactorizedBaseGH::compute_BW_grads + BW_JKO
// (proxgd/ProxGVIFactorizedBaseGH.h:64-113, 152-160).  The quadrature moments are the NGD ones
// (b = Vdmu, S = Vddmu at unit temperature), so the map is three small launches around the prep kernel:
//   jko_half:   Sig_half = (I - h S) Sigma (I - h S)^T,  Vdmu <- -b
//   prep (jko_h = h) on Sig_half:  Lam_new = W diag(1 / (l/2 + h + sqrt(l (l + 4h))/2)) W^T
//   jko_finish: Vddmu <- (Lam_new - Lam) / h
// ---------------------------------------------------------------------------------------------
struct JkoArgs {
  int K, d;
  double h;
  double* Vdmu;          // [K][d]     in: b, out: -b
  double* Vddmu;         // [K][d][d]  in: S, out: (Lam_new - Lam) / h
  const double* Sigma;   // [K][d][d]
  const double* Lam;     // [K][d][d]
  double* Shalf;         // [K][d][d]
  const double* LamNew;  // [K][d][d]
};

__global__ __launch_bounds__(64) void jko_half_kernel(JkoArgs a) {
  extern __shared__ double sm[];
  const int d = a.d, dd = d * d, k = blockIdx.x, lane = threadIdx.x;
  double* M = sm;            // I - h S
  double* Sg = M + dd;
  double* Tm = Sg + dd;      // M Sigma
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    M[e] = (i == j ? 1.0 : 0.0) - a.h * a.Vddmu[(size_t)k * dd + e];
    Sg[e] = a.Sigma[(size_t)k * dd + e];
  }
  for (int e = lane; e < d; e += 64) a.Vdmu[(size_t)k * d + e] = -a.Vdmu[(size_t)k * d + e];
  wave_lds_sync();
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += M[i * d + c] * Sg[c * d + j];
    Tm[e] = s;
  }
  wave_lds_sync();
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    if (i <= j) {                                           // upper triangle, mirrored: exactly symmetric
      double s = 0.0;
      for (int c = 0; c < d; ++c) s += Tm[i * d + c] * M[j * d + c];
      a.Shalf[(size_t)k * dd + i * d + j] = s;
      a.Shalf[(size_t)k * dd + j * d + i] = s;
    }
  }
}

__global__ __launch_bounds__(256) void jko_finish_kernel(JkoArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)a.K * a.d * a.d) a.Vddmu[i] = (a.LamNew[i] - a.Lam[i]) / a.h;
}

// every factor set of the problem in ONE launch: block -> (set, factor) through the offsets
constexpr int MAX_FSETS = 8;
struct PrepList {
  int nsets;
  int koff[MAX_FSETS + 1];
  FactorDev f[MAX_FSETS];
  const double* mu[MAX_FSETS];
  const double* Sigma[MAX_FSETS];
  // optional fused gather (update_mu_from_joint / update_precision_from_joint, gvibase/GVIFactorizedBase.h:104-114):
  // every block first pulls its factor's (mu_k, Sigma_k) out of the chain arrays -- forming the trial mean
  // gmu + gstep * gdmu on the fly when gdmu != null -- parks them in LDS for the decomposition and writes them to
  // mu_k / Sigma_k for the later kernels; blocks past koff[nsets] write the chain-level trial mean mu_out.
  int gather, n;
  const double* gmu;
  const double* gdmu;
  double gstep;
  const double* SigD;
  const double* SigU;
  double* mu_out;
  int64_t nmu;
  const int32_t* start[MAX_FSETS];
  double* mu_k[MAX_FSETS];
  double* Sigma_k[MAX_FSETS];
  const double* pred;    // predicated launch (device_common.hpp, pred_skip) or null
  double pred_val;
};

template <int EPLP>
__global__ __launch_bounds__(64) void prep_all_kernel(PrepList L) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x;
  if (pred_skip(L.pred, L.pred_val)) return;
  if ((int)blockIdx.x >= L.koff[L.nsets]) {                      // chain-level trial mean (gather mode only)
    const int64_t j = (int64_t)((int)blockIdx.x - L.koff[L.nsets]) * 64 + lane;
    if (j < L.nmu) L.mu_out[j] = L.gmu[j] + L.gstep * L.gdmu[j];
    return;
  }
  int si = 0;
  while (si + 1 < L.nsets && (int)blockIdx.x >= L.koff[si + 1]) ++si;
  const int k = (int)blockIdx.x - L.koff[si];
  if (!L.gather) { prep_body_d<EPLP>(L.f[si], L.mu[si], L.Sigma[si], k, sm, k); return; }
  const FactorDev& f = L.f[si];
  const int d = f.d, dd = d * d, dp = d + (d & 1), n = L.n, nn = n * n;
  double* Sl = sm + 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1;  // behind prep_body's own LDS
  double* ml = Sl + dd;
  const int s = L.start[si][k];
  for (int e = lane; e < dd; e += 64) {
    const int r = e / d, c = e % d;
    double v;
    if (r < n && c < n) v = L.SigD[(size_t)s * nn + r * n + c];
    else if (r >= n && c >= n) v = L.SigD[(size_t)(s + 1) * nn + (r - n) * n + (c - n)];
    else if (r < n) v = L.SigU[(size_t)s * nn + r * n + (c - n)];
    else v = L.SigU[(size_t)s * nn + c * n + (r - n)];
    Sl[e] = v;
    L.Sigma_k[si][(size_t)k * dd + e] = v;
  }
  for (int e = lane; e < d; e += 64) {
    const size_t j = (size_t)s * n + e;
    const double v = L.gdmu ? L.gmu[j] + L.gstep * L.gdmu[j] : L.gmu[j];
    ml[e] = v;
    L.mu_k[si][(size_t)k * d + e] = v;
  }
  wave_lds_sync();
  prep_body_d<EPLP>(f, ml, Sl, k, sm, 0);
}

// ---------------------------------------------------------------------------------------------
// psi evaluation at an x-space point (generic kernel, expand-based).
// ---------------------------------------------------------------------------------------------
__device__ inline double psi_range_1d(const double* p, double x) {
  // p = [y, mu_p, f*b, sig_r_sq, sig_p_sq]            (src/1d_example.cpp:25-35)
  const double e = x - p[1], r = p[0] - p[2] / x;
  return e * e / p[4] / 2 + r * r / p[3] / 2;
}

// PlanarSDF::getSignedDistance (helpers/CudaOperation.h:61-103): query clamped to the grid, bilinear
// interpolation of the column-major field (data[r + c rows], :130).
__device__ inline double sdf2d_lookup(const FactorDev& f, double px, double py) {
  const double xmax = f.sdf_ox + (f.sdf_cols - 1.0) * f.sdf_cell, ymax = f.sdf_oy + (f.sdf_rows - 1.0) * f.sdf_cell;
  const double xin = px < f.sdf_ox ? f.sdf_ox : (px > xmax ? xmax : px);
  const double yin = py < f.sdf_oy ? f.sdf_oy : (py > ymax ? ymax : py);
  const double col = (xin - f.sdf_ox) * f.sdf_inv_cell, row = (yin - f.sdf_oy) * f.sdf_inv_cell;
  const double lr = floor(row), lc = floor(col), hr = lr + 1.0, hc = lc + 1.0;
  const int lri = (int)lr, lci = (int)lc;
  const int hri = lri + 1 < f.sdf_rows ? lri + 1 : f.sdf_rows - 1;    // weight is 0 there; keeps the read in bounds
  const int hci = lci + 1 < f.sdf_cols ? lci + 1 : f.sdf_cols - 1;
  const int R = f.sdf_rows;
  return (hr - row) * (hc - col) * f.sdf[lri + lci * R] + (row - lr) * (hc - col) * f.sdf[hri + lci * R] +
         (hr - row) * (col - lc) * f.sdf[lri + hci * R] + (row - lr) * (col - lc) * f.sdf[hri + hci * R];
}

// SignedDistanceField::getSignedDistance (helpers/CudaOperation.h:165-226): clamped trilinear interpolation of
// data[r + c rows + z rows cols] (:304-306); x -> column, y -> row, z -> slice.
__device__ inline double sdf3d_lookup(const FactorDev& f, double px, double py, double pz) {
  const double xmax = f.sdf_ox + (f.sdf_cols - 1.0) * f.sdf_cell, ymax = f.sdf_oy + (f.sdf_rows - 1.0) * f.sdf_cell,
               zmax = f.sdf_oz + (f.sdf_nz - 1.0) * f.sdf_cell;
  const double xin = px < f.sdf_ox ? f.sdf_ox : (px > xmax ? xmax : px);
  const double yin = py < f.sdf_oy ? f.sdf_oy : (py > ymax ? ymax : py);
  const double zin = pz < f.sdf_oz ? f.sdf_oz : (pz > zmax ? zmax : pz);
  const double col = (xin - f.sdf_ox) * f.sdf_inv_cell, row = (yin - f.sdf_oy) * f.sdf_inv_cell, zz = (zin - f.sdf_oz) * f.sdf_inv_cell;
  const double lr = floor(row), lc = floor(col), lz = floor(zz), hr = lr + 1.0, hc = lc + 1.0, hz = lz + 1.0;
  const int lri = (int)lr, lci = (int)lc, lzi = (int)lz;
  const int hri = lri + 1 < f.sdf_rows ? lri + 1 : f.sdf_rows - 1;
  const int hci = lci + 1 < f.sdf_cols ? lci + 1 : f.sdf_cols - 1;
  const int hzi = lzi + 1 < f.sdf_nz ? lzi + 1 : f.sdf_nz - 1;
  const size_t R = f.sdf_rows, RC = R * f.sdf_cols;
  const double* g = f.sdf;
  return (hr - row) * (hc - col) * (hz - zz) * g[lri + lci * R + lzi * RC] + (row - lr) * (hc - col) * (hz - zz) * g[hri + lci * R + lzi * RC] +
         (hr - row) * (col - lc) * (hz - zz) * g[lri + hci * R + lzi * RC] + (row - lr) * (col - lc) * (hz - zz) * g[hri + hci * R + lzi * RC] +
         (hr - row) * (hc - col) * (zz - lz) * g[lri + lci * R + hzi * RC] + (row - lr) * (hc - col) * (zz - lz) * g[hri + lci * R + hzi * RC] +
         (hr - row) * (col - lc) * (zz - lz) * g[lri + hci * R + hzi * RC] + (row - lr) * (col - lc) * (zz - lz) * g[hri + hci * R + hzi * RC];
}

__device__ __forceinline__ double hinge_sq(double sd, double thr, double slope, double sigma) {
  const double err = sd > thr ? 0.0 : (thr - sd) * slope;
  return err * err * sigma;
}

// planar point robot (CudaOperation_PlanarPR::cost_obstacle_planar, helpers/CudaOperation.h:491-523): one ball at
// (x0, x1), slope 1.  p = [sigma, eps, r].
__device__ inline double psi_hinge_sdf2d(const FactorDev& f, const double* p, double px, double py) {
  return hinge_sq(sdf2d_lookup(f, px, py), p[1] + p[2], 1.0, p[0]);
}

// planar quadrotor body (CudaOperation_Quad::cost_obstacle_planar / vec_balls, helpers/CudaOperation.h:565-606):
// pose (x, z, phi) = x[0:3]; n_balls check points along the body axis starting at
// pos - (L - 1.5 r)/2 (cos phi, sin phi), spaced L/n_balls.  p = [sigma, eps, r, slope, n_balls, L].
__device__ inline double psi_hinge_sdf2d_body(const FactorDev& f, const double* p, double px, double pz, double phi) {
  double sn, cs;
  sincos(phi, &sn, &cs);
  const double r = p[2], L = p[5], thr = p[1] + r;
  const int nb = (int)p[4];
  const double lx = px - (L - r * 1.5) * cs / 2.0, lz = pz - (L - r * 1.5) * sn / 2.0;
  double cost = 0.0;
  for (int i = 0; i < nb; ++i)
    cost += hinge_sq(sdf2d_lookup(f, lx + L * cs / nb * i, lz + L * sn / nb * i), thr, p[3], p[0]);
  return cost;
}

// 3-D point robot (CudaOperation_3dpR::cost_obstacle_planar, helpers/CudaOperation.h:650-683).  p = [sigma, eps, r].
__device__ inline double psi_hinge_sdf3d(const FactorDev& f, const double* p, double px, double py, double pz) {
  return hinge_sq(sdf3d_lookup(f, px, py, pz), p[1] + p[2], 1.0, p[0]);
}

// 7-DOF-style arm (CudaOperation_3dArm::cost_obstacle + ForwardKinematics, helpers/CudaOperation.h:325-399, 752-771):
// sphere s sits on frame[s] of the DH chain T = prod_{i <= frame} DH(i, x_i + bias_i); n_balls = the factor dimension
// (theta.size(), :753), cost = sigma sum_s hinge(eps + radius_s - sdf(p_s))^2.  The reference builds every DH matrix
// from cosf / sinf (single precision, :388-395; products of two trig terms are float products) -- restated as such.
// Frames are non-decreasing (checked on the host), so the chain is advanced once.  p = [sigma, eps].
__device__ inline double psi_hinge_sdf3d_arm(const FactorDev& f, const double* p, const double* x, int d) {
  const double* A = f.arm;
  const int nd = (int)A[0], ns = (int)A[1];
  const double *a = A + 2, *al = a + nd, *dl = al + nd, *tb = dl + nd, *fr = tb + nd, *ce = fr + ns, *ra = ce + 3 * ns;
  double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};           // rows 0..2 of the homogeneous transform
  const int nb = d < ns ? d : ns;
  int done = -1;                                                  // last joint folded into T
  double cost = 0.0;
  for (int s = 0; s < nb; ++s) {
    const int frame = (int)fr[s];
    while (done < frame) {
      const int i = ++done;
      const float th = (float)(x[i] + tb[i]), alf = (float)al[i];
      // cosf / sinf of the reference: float in, float out.  Libm's differ in the last float bit; the correctly rounded
      // value (double trig of the float argument, rounded to float) is the representative, as in the oracle
      const float c = (float)cos((double)th), sn = (float)sin((double)th), cA = (float)cos((double)alf), sA = (float)sin((double)alf);
      const double m00 = c, m01 = -sn * cA, m02 = sn * sA, m03 = a[i] * c;
      const double m10 = sn, m11 = c * cA, m12 = -c * sA, m13 = a[i] * sn;
      const double m21 = sA, m22 = cA, m23 = dl[i];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double t0 = T[r * 4], t1 = T[r * 4 + 1], t2 = T[r * 4 + 2], t3 = T[r * 4 + 3];
        T[r * 4] = t0 * m00 + t1 * m10;
        T[r * 4 + 1] = t0 * m01 + t1 * m11 + t2 * m21;
        T[r * 4 + 2] = t0 * m02 + t1 * m12 + t2 * m22;
        T[r * 4 + 3] = t0 * m03 + t1 * m13 + t2 * m23 + t3;
      }
    }
    const double cx = ce[3 * s], cy = ce[3 * s + 1], cz = ce[3 * s + 2];
    const double px = T[3] + (T[0] * cx + T[1] * cy + T[2] * cz);
    const double py = T[7] + (T[4] * cx + T[5] * cy + T[6] * cz);
    const double pz = T[11] + (T[8] * cx + T[9] * cy + T[10] * cz);
    cost += hinge_sq(sdf3d_lookup(f, px, py, pz), p[1] + ra[s], 1.0, p[0]);
  }
  return cost;
}

// ---------------------------------------------------------------------------------------------
// moments_generic_kernel: any d (<= 32), any psi kind.  Block = 256 threads handles one factor and a
// range of points in sub-chunks of 256: stage 1 (thread = point) expands x = mu + S z, evaluates
// psi and parks [z, 1] and c = w psi in LDS; stage 2 (thread = output pair (a,b)) accumulates
// sum_i c_i zh_i[a] zh_i[b].  LDS-bound; it is the reference-shaped fallback and the parity
// cross-check for the register kernel.
// ---------------------------------------------------------------------------------------------
struct MomArgs {
  FactorDev f;
  const double* mu;        // [K][d]
  const double* psi_ext;   // [K][N] host-evaluated psi (HOST_CALLBACK) or nullptr
  double* partial;         // [K][nchunk][npo]
  int64_t chunk;           // points per block (multiple of 256 for generic, 64 for register kernel)
  int nchunk;
  int full;                // 1: all moments, 0: m0 only (cost pass)
  int flush;               // split kernel: steps between second-level flushes (SPLIT_FLUSH; 0 = plain recursive sums, A/B only)
  int64_t mchunk;          // mirror-half table: representatives per chunk (same nchunk)
  const double* pred;      // predicated launch (device_common.hpp, pred_skip) or null
  double pred_val;
};

// ---------------------------------------------------------------------------------------------
// moments_closed_kernel: quadrature-free moments of a sum-of-squares psi = sum_r s_r (u0 + H z)_r^2,
// z ~ N(0, I)  [NGDFactorizedLinear::calculate_partial_V, ngd/NGDFactorizedLinear.h:93-129].  Isserlis in
// the whitened space:  E[psi] = sum_r s_r (u0_r^2 + |h_r|^2),  E[z psi] = 2 sum_r s_r u0_r h_r,
// E[z z^T psi] = E[psi] I + 2 sum_r s_r h_r h_r^T.  Writes the same packed layout as one chunk of the
// sigma-point kernels, so the epilogue (back-transform, Lambda, temperature) is shared.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void moments_closed_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  const FactorDev& f = a.f;
  const int d = f.d, m = f.m, k = blockIdx.x;
  const double* H = f.H + (size_t)k * d * m;      // [d][m]
  const double* u0 = f.u0 + (size_t)k * m;
  const double* sg = f.sgn + (size_t)k * m;
  double m0 = 0.0;
  for (int r = 0; r < m; ++r) {
    double q = u0[r] * u0[r];
    for (int c = 0; c < d; ++c) q = fma(H[c * m + r], H[c * m + r], q);
    m0 = fma(sg[r], q, m0);
  }
  const int npo = a.full ? npairs(d) : 1;
  double* out = a.partial + (size_t)k * a.nchunk * npo;
  if (threadIdx.x == 0) out[0] = m0;
  if (!a.full) return;
  for (int j = 1 + threadIdx.x; j < npo; j += 64) {
    double v = 0.0;
    if (j <= d) {
      const int c = j - 1;
      for (int r = 0; r < m; ++r) v = fma(sg[r] * u0[r], H[c * m + r], v);
      v *= 2.0;
    } else {
      int rem = j - 1 - d, row = 0;
      while (rem >= d - row) { rem -= d - row; ++row; }
      const int col = row + rem;
      for (int r = 0; r < m; ++r) v = fma(sg[r] * H[row * m + r], H[col * m + r], v);
      v = 2.0 * v + (row == col ? m0 : 0.0);
    }
    out[j] = v;
  }
}

constexpr int GEN_BS = 256;
constexpr int GEN_MAX_OUT = 3;   // outputs per thread: npairs(d) <= 768 -> d <= 37

__global__ __launch_bounds__(GEN_BS) void moments_generic_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  extern __shared__ double sm[];
  const FactorDev& f = a.f;
  const int d = f.d, m = f.m, k = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
  const int dz = d + 1;
  double* zs = sm;                         // [256][dz]   z and the constant 1
  double* xs = zs + GEN_BS * dz;           // [256][dz]   x = mu + S z
  double* cs = xs + GEN_BS * dz;           // [256]
  double* Ssh = cs + GEN_BS;               // [d][d]
  double* mush = Ssh + d * d;              // [d]
  double* Ash = mush + d;                  // [m][d]
  double* bsh = Ash + m * d;               // [m]
  double* gsh = bsh + m;                   // [m]
  for (int e = tid; e < d * d; e += GEN_BS) Ssh[e] = f.S[(size_t)k * d * d + e];
  for (int e = tid; e < d; e += GEN_BS) mush[e] = a.mu[(size_t)k * d + e];
  for (int e = tid; e < m * d; e += GEN_BS) Ash[e] = f.A[(size_t)k * m * d + e];
  for (int e = tid; e < m; e += GEN_BS) { bsh[e] = f.b[(size_t)k * m + e]; gsh[e] = f.sgn[(size_t)k * m + e]; }
  const int npo = a.full ? npairs(d) : 1;
  int oa[GEN_MAX_OUT], ob[GEN_MAX_OUT];
  double acc[GEN_MAX_OUT];
#pragma unroll
  for (int o = 0; o < GEN_MAX_OUT; ++o) {
    acc[o] = 0.0;
    const int j = tid + o * GEN_BS;
    int pa_ = d, pb_ = d;                   // j == 0 -> (1,1)
    if (j >= 1 && j <= d) { pa_ = j - 1; pb_ = d; }
    else if (j > d) {
      int rem = j - 1 - d, row = 0;
      while (row < d && rem >= d - row) { rem -= d - row; ++row; }
      pa_ = row; pb_ = row + rem;
    }
    oa[o] = pa_; ob[o] = pb_;
  }
  double csum = 0.0;
  const int64_t i0 = (int64_t)chunk * a.chunk;
  const int64_t i1 = (i0 + a.chunk < f.Np) ? i0 + a.chunk : f.Np;
  __syncthreads();
  for (int64_t base = i0; base < i1; base += GEN_BS) {
    const int64_t i = base + tid;
    double* zr = zs + tid * dz;
    double* xr = xs + tid * dz;
    double wi = 0.0;
    if (i < i1) {
      wi = f.w[i];
      for (int c = 0; c < d; ++c) zr[c] = f.Zt[(size_t)c * f.Np + i];
    } else {
      for (int c = 0; c < d; ++c) zr[c] = 0.0;
    }
    zr[d] = 1.0;
    double psi = 0.0;
    if (i < f.N) {
      for (int r = 0; r < d; ++r) {
        double x = mush[r];
        for (int c = 0; c < d; ++c) x += Ssh[r * d + c] * zr[c];
        xr[r] = x;
      }
      if (a.psi_ext) psi = a.psi_ext[(size_t)k * f.N + i];
      else if (f.kind == KIND_RANGE_1D) psi = psi_range_1d(f.raw + (size_t)k * f.raw_stride, xr[0]);
      else if (f.kind == KIND_HINGE_SDF_2D) psi = psi_hinge_sdf2d(f, f.raw + (size_t)k * f.raw_stride, xr[0], xr[1]);
      else if (f.kind == KIND_HINGE_SDF_2D_BODY) psi = psi_hinge_sdf2d_body(f, f.raw + (size_t)k * f.raw_stride, xr[0], xr[1], xr[2]);
      else if (f.kind == KIND_HINGE_SDF_3D) psi = psi_hinge_sdf3d(f, f.raw + (size_t)k * f.raw_stride, xr[0], xr[1], xr[2]);
      else if (f.kind == KIND_HINGE_SDF_3D_ARM) psi = psi_hinge_sdf3d_arm(f, f.raw + (size_t)k * f.raw_stride, xr, d);
      else {
        for (int r = 0; r < m; ++r) {
          double u = bsh[r];
          for (int c = 0; c < d; ++c) u += Ash[r * d + c] * xr[c];
          psi += gsh[r] * u * u;
        }
      }
    }
    const double c = wi * psi;
    cs[tid] = c;
    csum += c;
    __syncthreads();
    if (a.full) {
#pragma unroll
      for (int o = 0; o < GEN_MAX_OUT; ++o) {
        if (tid + o * GEN_BS < npo) {
          double s = acc[o];
          const int pa_ = oa[o], pb_ = ob[o];
          for (int t = 0; t < GEN_BS; ++t) s += cs[t] * zs[t * dz + pa_] * zs[t * dz + pb_];
          acc[o] = s;
        }
      }
    }
    __syncthreads();
  }
  double* out = a.partial + ((size_t)k * a.nchunk + chunk) * npo;
  if (a.full) {
#pragma unroll
    for (int o = 0; o < GEN_MAX_OUT; ++o)
      if (tid + o * GEN_BS < npo) out[tid + o * GEN_BS] = acc[o];
  } else {
    cs[tid] = csum;
    __syncthreads();
    for (int s = GEN_BS / 2; s > 0; s >>= 1) {
      if (tid < s) cs[tid] += cs[tid + s];
      __syncthreads();
    }
    if (tid == 0) out[0] = cs[0];
  }
}

// ---------------------------------------------------------------------------------------------
// moments_reg_kernel<D, Psi, FULL>: the hot kernel.  Block = 4 waves = 4 consecutive factors over
// the same range of points (so a Z chunk is fetched once per block into L1/L2 and reused by the
// four waves).  Lane owns points i = base + lane: d coalesced 8-byte loads from the dimension-major
// table, psi through the factor's LDS-resident H (wave-uniform broadcast reads), and all
// (d+1)(d+2)/2 z-space accumulators in registers.  Cross-lane reduction once per wave: 16 accumulators
// at a time through a padded LDS tile, then two xor-shuffles.  No atomics: partials are written per
// (factor, chunk) and summed in fixed order by epilogue_kernel.
// ---------------------------------------------------------------------------------------------
template <int D, int M>
struct PsiQuad {
  static constexpr int LDS = M * D + 2 * M;
  static constexpr bool GUARD = false;
  __device__ static void load(const MomArgs& a, int k, double* hs, int lane) {
    for (int e = lane; e < M * D; e += 64) hs[e] = a.f.H[(size_t)k * M * D + e];
    if (lane < M) {
      hs[M * D + lane] = a.f.u0[(size_t)k * M + lane];
      hs[M * D + M + lane] = a.f.sgn[(size_t)k * M + lane];
    }
  }
  __device__ static double eval(const double (&z)[D], const double* hs, const MomArgs&) {
    double u[M];                      // M independent FMA chains (column-outer order)
#pragma unroll
    for (int r = 0; r < M; ++r) u[r] = hs[M * D + r];
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
      for (int r = 0; r < M; ++r) u[r] = fma(hs[c * M + r], z[c], u[r]);
    }
    double psi = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) psi = fma(hs[M * D + M + r] * u[r], u[r], psi);
    return psi;
  }
};

struct PsiRange1D {
  static constexpr int LDS = 8;
  static constexpr bool GUARD = true;
  __device__ static void load(const MomArgs& a, int k, double* hs, int lane) {
    if (lane == 0) { hs[0] = a.mu[k]; hs[1] = a.f.S[k]; }
    if (lane < 5) hs[2 + lane] = a.f.raw[(size_t)k * a.f.raw_stride + lane];
  }
  __device__ static double eval(const double (&z)[1], const double* hs, const MomArgs&) {
    return psi_range_1d(hs + 2, fma(hs[1], z[0], hs[0]));
  }
};

// nonlinear obstacle costs: pose = (mu + S z)[0:NPOSE] (NPOSE rows of S in LDS), SDF look-ups straight from L2
template <int D, int KIND>
struct PsiHingeSdf {
  static constexpr int NPOSE = KIND == KIND_HINGE_SDF_2D ? 2 : 3;
  static constexpr int NPAR = KIND == KIND_HINGE_SDF_2D_BODY ? 6 : 3;
  static constexpr int LDS = NPOSE * D + NPOSE + NPAR;
  static constexpr bool GUARD = false;
  __device__ static void load(const MomArgs& a, int k, double* hs, int lane) {
    if (lane < NPOSE * D) hs[lane] = a.f.S[(size_t)k * D * D + lane];       // rows 0 .. NPOSE-1 of S
    if (lane < NPOSE) hs[NPOSE * D + lane] = a.mu[(size_t)k * D + lane];
    if (lane < NPAR) hs[NPOSE * D + NPOSE + lane] = a.f.raw[(size_t)k * a.f.raw_stride + lane];
  }
  __device__ static double eval(const double (&z)[D], const double* hs, const MomArgs& a) {
    double pose[NPOSE];
#pragma unroll
    for (int r = 0; r < NPOSE; ++r) {
      double v = hs[NPOSE * D + r];
#pragma unroll
      for (int c = 0; c < D; ++c) v = fma(hs[r * D + c], z[c], v);
      pose[r] = v;
    }
    const double* p = hs + NPOSE * D + NPOSE;
    if (KIND == KIND_HINGE_SDF_2D) return psi_hinge_sdf2d(a.f, p, pose[0], pose[1]);
    if (KIND == KIND_HINGE_SDF_2D_BODY) return psi_hinge_sdf2d_body(a.f, p, pose[0], pose[1], pose[NPOSE - 1]);
    return psi_hinge_sdf3d(a.f, p, pose[0], pose[1], pose[NPOSE - 1]);
  }
};
template <int D> using PsiHingeSdf2D = PsiHingeSdf<D, KIND_HINGE_SDF_2D>;

// (bx, by): the block's position in the (ceil(K / 4), nchunk) grid of the set -- the launch's own blockIdx, or a virtual one
// when several sets share a launch (moments_planar3_kernel).  hs_: [4][Psi::LDS], red_: [4][16][65] doubles of LDS.
// kfix >= 0 (factor_block3_kernel): the calling wave takes chunk byfix of factor kfix (idle when kfix >= K or byfix >= nchunk);
// (bx, by) are then unused.  Same points per (factor, chunk), same sums.
template <int D, typename Psi, bool FULL>
__device__ __forceinline__ void reg_body(const MomArgs& a, const int bx, const int by_, double* hs_, double* red_, const int kfix = -1,
                                         const int byfix = 0) {
  constexpr int NP = FULL ? (D + 1) * (D + 2) / 2 : 1;
  constexpr int NB = (NP + 15) / 16;
  double (*hs)[Psi::LDS] = (double (*)[Psi::LDS])hs_;
  double (*red)[16][65] = (double (*)[16][65])red_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = kfix >= 0 ? kfix : bx * 4 + wave;
  const int by = kfix >= 0 ? byfix : by_;
  const bool active = k < a.f.K && by < a.nchunk;
  if (active) Psi::load(a, k, hs[wave], lane);
  __syncthreads();
  double acc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) acc[j] = 0.0;
  const int64_t Np = a.f.Np;
  const int64_t i0 = (int64_t)by * a.chunk;
  const int64_t i1 = (i0 + a.chunk < Np) ? i0 + a.chunk : Np;
  const double* __restrict__ Zt = a.f.Zt;
  const double* __restrict__ w = a.f.w;
  if (active) {
    for (int64_t i = i0 + lane; i < i1; i += 64) {
      double z[D];
#pragma unroll
      for (int c = 0; c < D; ++c) z[c] = Zt[(size_t)c * Np + i];
      // padded tail (i >= N) carries w = 0, z = 0; the select keeps a non-finite psi(mu) out
      const double cw = i < a.f.N ? w[i] * Psi::eval(z, hs[wave], a) : 0.0;
      acc[0] += cw;
      if (FULL) {
        int q = 1 + D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          const double t = cw * z[c];
          acc[1 + c] += t;
#pragma unroll
          for (int e = c; e < D; ++e) { acc[q] = fma(t, z[e], acc[q]); ++q; }
        }
      }
    }
  }
  double* out = a.partial + ((size_t)(active ? k : 0) * a.nchunk + by) * NP;
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (bb * 16 + j < NP) red[wave][j][lane] = acc[bb * 16 + j];
    __syncthreads();
    const int j = lane & 15, part = lane >> 4;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += red[wave][j][part * 16 + t];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (active && lane < 16 && bb * 16 + lane < NP) out[bb * 16 + lane] = s;
    __syncthreads();
  }
}

template <int D, typename Psi, bool FULL>
__global__ __launch_bounds__(256) void moments_reg_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  __shared__ double hs[4 * Psi::LDS];
  __shared__ double red[4 * 16 * 65];
  reg_body<D, Psi, FULL>(a, (int)blockIdx.x, (int)blockIdx.y, hs, red);
}

// ---------------------------------------------------------------------------------------------
// moments_split_kernel<D, FULL>: factor dimensions whose (D+1)(D+2)/2 accumulators do not fit one
// lane (D = 16, 20, 24; BASELINE configs[4]).  Block = 4 waves on ONE factor and the SAME 64 points per
// step:  (1) every wave decodes z from the table's 8-bit node codes through a 256-entry LDS look-up
// (D/4 coalesced dword loads per point instead of D doubles: 32 B/point at D = 24, nodes stay exact
// fp64);  (2) wave v evaluates residual rows r = v, v+4, ... of the sum-of-squares psi and the four
// partial sums meet in a double-buffered LDS tile (one LDS-only barrier per step);  (3) wave v owns
// rows [split_row(v), split_row(v+1)) of the packed upper triangle (+ m1 of those rows, wave 0 also m0)
// and accumulates them in registers.  Row boundaries balance the (D - a + 1) entries per row.
// ---------------------------------------------------------------------------------------------
// Row ownership of wave v: rows [lo, hi) plus the tail rows [lo2, hi2) (short rows from the bottom of the
// triangle top up the waves that own the long rows).  Row a carries D - a + 1 accumulators (m1[a], M2[a][a..D)).
struct SplitRows { int lo, hi, lo2, hi2; };
template <int D>
__host__ __device__ constexpr int split_count(int r0, int r1) {
  int n = 0;
  for (int a = r0; a < r1; ++a) n += D - a + 1;
  return n;
}
template <int D>
__host__ __device__ constexpr SplitRows split_rows(int v) {
  if (D == 24) {                                   // 82 / 82 / 80 / 81 accumulators
    const SplitRows t[4] = {{0, 3, 21, 24}, {3, 7, 0, 0}, {7, 12, 0, 0}, {12, 21, 0, 0}};
    return t[v];
  }
  // contiguous greedy split
  int total = 0;
  for (int a = 0; a < D; ++a) total += D - a + 1;
  int bnd[5] = {0, 0, 0, 0, D};
  int acc = 0, a = 0;
  for (int q = 0; q < 3; ++q) {
    const int target = total * (q + 1) / 4;
    while (a < D && acc + (D - a + 1) / 2 <= target) { acc += D - a + 1; ++a; }
    bnd[q + 1] = a;
  }
  return SplitRows{bnd[v], bnd[v + 1], 0, 0};
}

template <int D>
__device__ __forceinline__ void split_decode(const uint32_t (&cd)[D / 4], const double* lut, double (&z)[D]) {
#pragma unroll
  for (int c = 0; c < D; ++c) z[c] = lut[(cd[c / 4] >> (8 * (c % 4))) & 255u];
}

// cross-lane sums of NT accumulators, 16 at a time through this wave's LDS tile (wave-private: no block
// barrier); local accumulator j lands at packed index dst(j)
template <int NT, typename Dst>
__device__ __forceinline__ void split_reduce(const double* acc, double* rw, double* out, int lane, Dst dst) {
  constexpr int NB = (NT + 15) / 16;
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (bb * 16 + j < NT) rw[j * 65 + lane] = acc[bb * 16 + j];
    wave_lds_sync();
    const int j = lane & 15, part16 = lane >> 4;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += rw[j * 65 + part16 * 16 + t];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const int loc = bb * 16 + lane;
    if (lane < 16 && loc < NT) out[dst(loc)] = s;
    wave_lds_sync();
  }
}

// Second level of the split kernel's accumulation.  At (24,7) sum |w_i| / sum w_i = 1.5e7 and a lane adds ~16 000 terms per
// chunk, so plain recursive summation carries eps * |w|_1 * sqrt(terms) ~ 5e-6 relative (profiles/r01_g_c5_full.json) -- over
// the 1e-6 bar.  Every SPLIT_FLUSH steps the wave's register accumulators are therefore reduced across the lanes (same LDS
// tile transpose as split_reduce) and added, with Neumaier compensation, into a wave-private (sum, comp) pair per moment in
// LDS; the registers restart from zero.  First level: 64 terms per lane; cross-lane: a tree; second level: compensated.
constexpr int SPLIT_FLUSH = 64;
constexpr int SPLIT_LVL2_SLOTS = 96;     // >= accumulators per wave (83 at D = 24)
template <int NT>
__device__ __forceinline__ void split_flush(double* acc, double* rw, double* s2, double* c2, int lane) {
  constexpr int NB = (NT + 15) / 16;
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (bb * 16 + j < NT) { rw[j * 65 + lane] = acc[bb * 16 + j]; acc[bb * 16 + j] = 0.0; }
    wave_lds_sync();
    const int j = lane & 15, part16 = lane >> 4;
    double x = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) x += rw[j * 65 + part16 * 16 + t];
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    const int loc = bb * 16 + lane;
    if (lane < 16 && loc < NT) {
      const double sv = s2[loc], tv = sv + x;
      c2[loc] += fabs(sv) >= fabs(x) ? (sv - tv) + x : (x - tv) + sv;
      s2[loc] = tv;
    }
    wave_lds_sync();
  }
}

// psi operands: H is wave-uniform and read-only for the whole launch, so it is addressed through the constant
// address space -- the compiler then fetches it with s_load_dwordx* into SGPRs (scalar cache) and feeds it to
// v_fma_f64 as the scalar source: no LDS traffic and no VGPRs for the psi operands.
typedef const double __attribute__((address_space(4))) cdouble_t;

// u[rr] = u0 + sum_c Hq[c][rr] z[c] for the R rows of one wave block; returns sum_rr sgn (u)^2.
// The operands are re-fetched every call (scalar cache) in groups of GC columns, one group ahead of its use;
// the empty asm makes each group's offset opaque so the loads are neither hoisted out of the point loop (that
// would need 2 m D / 4 SGPRs and spill) nor merged into one burst.
template <int D, int R>
__device__ __forceinline__ double split_psi_rows(cdouble_t* hq, const double* u0s, const double* sgs, const double (&z)[D]) {
  // columns per group: 12-18 doubles (24-36 SGPRs) in flight, two groups live at a time
  constexpr int GC = (D % 4 == 0 && R <= 3) ? 4 : (D % 3 == 0 && R == 6 ? 3 : (D % 2 == 0 && R <= 6 ? 2 : 1));
  constexpr int NG = D / GC, GS = GC * R;
  double u[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) u[rr] = u0s[rr];
  double h[GS], hn[GS];
  int off = 0;
  asm volatile("" : "+s"(off));
#pragma unroll
  for (int j = 0; j < GS; ++j) h[j] = hq[off + j];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
      int o2 = (g + 1) * GS;
      asm volatile("" : "+s"(o2));
#pragma unroll
      for (int j = 0; j < GS; ++j) hn[j] = hq[o2 + j];
    }
#pragma unroll
    for (int cc = 0; cc < GC; ++cc) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) u[rr] = fma(h[cc * R + rr], z[g * GC + cc], u[rr]);
    }
#pragma unroll
    for (int j = 0; j < GS; ++j) h[j] = hn[j];
  }
  double part = 0.0;
#pragma unroll
  for (int rr = 0; rr < R; ++rr) part = fma(sgs[rr] * u[rr], u[rr], part);
  return part;
}

template <int D, int R, int V>
__device__ __forceinline__ void split_body(const MomArgs& a, const int k, const int lane, const double* lut,
                                           const double* hs, double* px, double* red, double* lvl2) {
  constexpr SplitRows RW = split_rows<D>(V);
  constexpr int NA1 = split_count<D>(RW.lo, RW.hi), NA = NA1 + split_count<D>(RW.lo2, RW.hi2);
  constexpr int NT = NA + (V == 0 ? 1 : 0);                      // wave 0 also carries m0
  static_assert(NT <= SPLIT_LVL2_SLOTS, "second-level slots");
  double acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = 0.0;
  const int64_t Np = a.f.Np;
  const int64_t i0 = (int64_t)blockIdx.y * a.chunk;
  const int64_t i1 = (i0 + a.chunk < Np) ? i0 + a.chunk : Np;
  const uint32_t* __restrict__ codes = a.f.codes;
  const double* __restrict__ w = a.f.w;
  cdouble_t* hq = (cdouble_t*)(a.f.Hq + ((size_t)k * 4 + V) * D * R);
  double* s2 = lvl2 + V * 2 * SPLIT_LVL2_SLOTS;          // wave-private second-level (sum | compensation)
  double* c2 = s2 + SPLIT_LVL2_SLOTS;
  for (int j = lane; j < 2 * SPLIT_LVL2_SLOTS; j += 64) s2[j] = 0.0;
  wave_lds_sync();
  int since_flush = 0;
  int buf = 0;
  uint32_t cn[D / 4];                                            // next step's codes / weight (prefetch)
#pragma unroll
  for (int g = 0; g < D / 4; ++g) cn[g] = codes[(size_t)g * Np + i0 + lane];
  double wn = w[i0 + lane];
  for (int64_t base = i0; base < i1; base += 64) {               // uniform trip count: barrier inside
    const int64_t i = base + lane;
    uint32_t cd[D / 4];
#pragma unroll
    for (int g = 0; g < D / 4; ++g) cd[g] = cn[g];
    const double wi = wn;
    if (base + 64 < i1) {
#pragma unroll
      for (int g = 0; g < D / 4; ++g) cn[g] = codes[(size_t)g * Np + i + 64];
      wn = w[i + 64];
    }
    double z[D];
    split_decode<D>(cd, lut, z);
    const double part = split_psi_rows<D, R>(hq, hs + V * R, hs + 4 * R + V * R, z);
    px[(buf * 4 + V) * 64 + lane] = part;
    lds_barrier();
    const double* pb = px + buf * 4 * 64 + lane;
    const double psi = (pb[0] + pb[64]) + (pb[128] + pb[192]);
    buf ^= 1;
    const double cw = i < a.f.N ? wi * psi : 0.0;               // pad rows: w = 0 and a non-finite psi kept out
    if (V == 0) acc[NT - 1] += cw;
    int q = 0;
#pragma unroll
    for (int c = RW.lo; c < RW.hi; ++c) {
      const double t = cw * z[c];
      acc[q++] += t;
#pragma unroll
      for (int e = c; e < D; ++e) { acc[q] = fma(t, z[e], acc[q]); ++q; }
    }
#pragma unroll
    for (int c = RW.lo2; c < RW.hi2; ++c) {
      const double t = cw * z[c];
      acc[q++] += t;
#pragma unroll
      for (int e = c; e < D; ++e) { acc[q] = fma(t, z[e], acc[q]); ++q; }
    }
    if (a.flush > 0 && ++since_flush == a.flush) {                  // wave-uniform; no block barrier inside
      split_flush<NT>(acc, red + V * 16 * 65, s2, c2, lane);
      since_flush = 0;
    }
  }
  split_flush<NT>(acc, red + V * 16 * 65, s2, c2, lane);
  constexpr int NPK = (D + 1) * (D + 2) / 2;
  double* out = a.partial + ((size_t)k * a.nchunk + blockIdx.y) * NPK;
  auto dst = [](int loc) {
    if (loc >= NA) return 0;                                       // m0
    int row = loc < NA1 ? RW.lo : RW.lo2, rem = loc < NA1 ? loc : loc - NA1;
    while (rem >= D - row + 1) { rem -= D - row + 1; ++row; }
    return rem == 0 ? 1 + row : pair_index(D, row, row + rem - 1);
  };
  for (int loc = lane; loc < NT; loc += 64) out[dst(loc)] = s2[loc] + c2[loc];
}

// cost pass: psi only, so nothing to share -- every wave takes its own 64 points of a 256-point step
template <int D, int R>
__device__ __forceinline__ void split_cost_body(const MomArgs& a, const int k, const int wave, const int lane,
                                                const double* lut, const double* hs, double* red) {
  const int64_t Np = a.f.Np;
  const int64_t i0 = (int64_t)blockIdx.y * a.chunk;
  const int64_t i1 = (i0 + a.chunk < Np) ? i0 + a.chunk : Np;
  const uint32_t* __restrict__ codes = a.f.codes;
  const double* __restrict__ w = a.f.w;
  cdouble_t* hq = (cdouble_t*)(a.f.Hq + (size_t)k * 4 * D * R);
  double acc = 0.0, comp = 0.0;                                   // compensated: same |w|_1 argument as split_flush
  for (int64_t i = i0 + wave * 64 + lane; i < i1; i += 256) {
    uint32_t cd[D / 4];
#pragma unroll
    for (int g = 0; g < D / 4; ++g) cd[g] = codes[(size_t)g * Np + i];
    const double wi = w[i];
    double z[D];
    split_decode<D>(cd, lut, z);
    double psi = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) psi += split_psi_rows<D, R>(hq + v * D * R, hs + v * R, hs + 4 * R + v * R, z);
    const double x = i < a.f.N ? wi * psi : 0.0;
    const double tv = acc + x;
    comp += fabs(acc) >= fabs(x) ? (acc - tv) + x : (x - tv) + acc;
    acc = tv;
  }
  acc += comp;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.partial[(size_t)k * a.nchunk + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int SPLIT_LDS_DOUBLES(int D) { return 256 + 2 * D + 8 + 2 * 4 * 64 + 4 * 16 * 65 + 4 * 2 * SPLIT_LVL2_SLOTS; }

template <int D, int R, bool FULL>
__global__ __launch_bounds__(256, 2) void moments_split_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  extern __shared__ double sm[];
  double* lut = sm;                        // [256]
  double* hs = lut + 256;                  // u0 [4 R] | sgn [4 R]  (rows >= m: 0)
  double* px = hs + 2 * D + 8;             // [2][4][64] partial psi
  double* red = px + 2 * 4 * 64;           // [4][16][65]
  double* lvl2 = red + 4 * 16 * 65;        // [4][2][SPLIT_LVL2_SLOTS] second-level sums
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int k = blockIdx.x, m = a.f.m;
  lut[threadIdx.x] = a.f.lut[threadIdx.x];
  if (threadIdx.x < 4 * R) {
    const bool live = (int)threadIdx.x < m;
    hs[threadIdx.x] = live ? a.f.u0[(size_t)k * m + threadIdx.x] : 0.0;
    hs[4 * R + threadIdx.x] = live ? a.f.sgn[(size_t)k * m + threadIdx.x] : 0.0;
  }
  __syncthreads();
  if (!FULL) { split_cost_body<D, R>(a, k, wave, lane, lut, hs, red); return; }
  switch (wave) {
    case 0: split_body<D, R, 0>(a, k, lane, lut, hs, px, red, lvl2); break;
    case 1: split_body<D, R, 1>(a, k, lane, lut, hs, px, red, lvl2); break;
    case 2: split_body<D, R, 2>(a, k, lane, lut, hs, px, red, lvl2); break;
    default: split_body<D, R, 3>(a, k, lane, lut, hs, px, red, lvl2); break;
  }
}

// ---------------------------------------------------------------------------------------------
// moments_sreg_kernel<D, M, FULL>: moments_reg_kernel for the sum-of-squares kinds with the psi operands
// taken from SGPRs (scalar cache, see split_psi_rows) instead of LDS broadcast reads.  f.H is [D][M]: a
// column's M entries contiguous, which is the group layout split_psi_rows walks.  Block = 4 waves = 4
// consecutive factors over the same range of points.
// ---------------------------------------------------------------------------------------------
template <int D, int M, bool FULL>
__device__ __forceinline__ void sreg_body(const MomArgs& a, const int bx, const int by_, double* usb, double* redb, const int kfix = -1,
                                          const int byfix = 0) {
  constexpr int NP = FULL ? (D + 1) * (D + 2) / 2 : 1;
  constexpr int NB = (NP + 15) / 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* us_w = usb + wave * 2 * M;               // [2 M] u0 | sgn of this wave's factor
  double (*red_w)[65] = (double (*)[65])(redb + wave * 16 * 65);
  const int kq = kfix >= 0 ? kfix : bx * 4 + wave;            // (kfix: see reg_body)
  const int by = kfix >= 0 ? byfix : by_;
  const bool active = kq < a.f.K && by < a.nchunk;
  const int k = __builtin_amdgcn_readfirstlane(active ? kq : a.f.K - 1);   // inactive waves redo the last factor
  if (lane < M) {
    us_w[lane] = a.f.u0[(size_t)k * M + lane];
    us_w[M + lane] = a.f.sgn[(size_t)k * M + lane];
  }
  __syncthreads();
  const uint64_t hbase = (uint64_t)(a.f.H + (size_t)k * M * D);
  cdouble_t* hq = (cdouble_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(hbase >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)hbase));
  double acc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) acc[j] = 0.0;
  const int64_t Np = a.f.Np;
  const int64_t i0 = (int64_t)by * a.chunk;
  const int64_t i1 = (i0 + a.chunk < Np) ? i0 + a.chunk : Np;
  const double* __restrict__ Zt = a.f.Zt;
  const double* __restrict__ w = a.f.w;
  for (int64_t base = i0; base < i1; base += 64) {             // wave-uniform loop (chunks are whole 64-point tiles)
    const int64_t i = base + lane;
    double z[D];
#pragma unroll
    for (int c = 0; c < D; ++c) z[c] = Zt[(size_t)c * Np + i];
    const double wi = w[i];
    const double psi = split_psi_rows<D, M>(hq, us_w, us_w + M, z);
    const double cw = i < a.f.N ? wi * psi : 0.0;
    acc[0] += cw;
    if (FULL) {
      int q = 1 + D;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double t = cw * z[c];
        acc[1 + c] += t;
#pragma unroll
        for (int e = c; e < D; ++e) { acc[q] = fma(t, z[e], acc[q]); ++q; }
      }
    }
  }
  double* out = a.partial + ((size_t)k * a.nchunk + by) * NP;
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (bb * 16 + j < NP) red_w[j][lane] = acc[bb * 16 + j];
    wave_lds_sync();
    const int j = lane & 15, part = lane >> 4;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += red_w[j][part * 16 + t];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (active && lane < 16 && bb * 16 + lane < NP) out[bb * 16 + lane] = s;
    wave_lds_sync();
  }
}

// ---------------------------------------------------------------------------------------------
// sreg_pipe_body<D, M, SIGNED>: the full-moments body of sreg_body, scheduled by hand.
//
// What limits sreg_body (profiles/r01_i_pmc_*: VALU pipe 68 % busy, 23 % of wave-cycles waiting): (i) the 13
// global_load_dwordx2 of a 64-point step are issued at the loop head and waited on -- the 91 accumulators leave no room for
// a second copy of z and the compiler will not reuse z's registers in place; (ii) a wave issues at most one instruction
// per 4-cycle slot, so every non-VALU instruction (64-bit VGPR address arithmetic, SALU address arithmetic of the operand
// loads, LDS reads of u0 / sgn) takes a slot in which the SIMD's only other wave must have a VALU instruction ready;
// (iii) the scalar operand loads of the next group are issued right before the wait of the current one (SMEM returns out
// of order, the wait is lgkmcnt(0)), so two scalar-cache latencies per step are exposed.  Here:
//   * the table is read from its TILE-MAJOR copy Zq[tile][row][64] (row D = weights): ONE wave-uniform base, ONE 32-bit
//     per-lane offset that advances by a tile per step, rows addressed by the instruction's immediate offset -- no address
//     arithmetic at all, every load still one coalesced 512-byte line;
//   * the loads are inline asm (the compiler's waitcnt pass does not track them), so the schedule is explicit: row c of
//     the packed M2 update is the LAST reader of z[c]; right behind it the NEXT step's z[c] is loaded into the same
//     register pair (w goes out as soon as c = w psi is formed).  Loads return in order, so "z[c] has landed" is
//     s_waitcnt vmcnt(D - 1 - c), placed in front of column c of the psi phase: every column has the rest of the
//     accumulation plus part of the psi phase to cover the L2 latency;
//   * psi operands: H is addressed through one opaque base per step (immediate offsets, no SALU arithmetic); the first two
//     operand groups of the NEXT step are requested during the accumulation phase, and inside the psi phase the request of
//     group g+1 sits behind the first column of group g, i.e. behind the wait, so it has a full group of FMAs to land;
//   * u0 stays in registers and rides in as src2 of the first column's v_fma; the sign multiply disappears when every
//     residual row has positive weight (SIGNED = false: the usual positive-definite Q^-1 / K^-1).
// The asm waits carry the registers they guard as in/out operands, so the compiler can neither hoist a use above its wait
// nor reuse a register while a load is in flight; sched_barriers pin the issue points.  Same operations in the same order
// on the same values as sreg_body: results are bit-identical (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------
#define GVI_ZLOAD(dst, voff, base, imm) \
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(imm) : "memory")

// compile-time recursion instead of unrolled loops: the asm statements need their immediates as constant expressions
template <int D, int BIAS, int C = 0>
__device__ __forceinline__ void pipe_zload_all(double (&z)[D], const unsigned voff, const char* const Zq) {
  if constexpr (C < D) {
    GVI_ZLOAD(z[C], voff, Zq, C * 512 - BIAS);
    pipe_zload_all<D, BIAS, C + 1>(z, voff, Zq);
  }
}

// psi phase, column C: u += H[:, C] z[C].  Operand group g = C / GC lives in hA (g even) or hB (g odd); the request of
// group g + 1 goes out behind the first column of group g (g >= 1), i.e. behind that group's wait, into the buffer
// group g - 1 has left.
// FROM_ZERO: u starts at 0 (v = H z alone, the mirror-pair form) instead of u0
template <int D, int M, int GC, bool FROM_ZERO, int C = 0>
__device__ __forceinline__ void pipe_psi_cols(double (&u)[M], double (&z)[D], double (&hA)[GC * M], double (&hB)[GC * M],
                                              const double (&u0v)[M], cdouble_t* const hp) {
  if constexpr (C < D) {
    constexpr int g = C / GC, cc = C % GC, NG = D / GC, GS = GC * M;
    double (&h)[GS] = (g % 2 == 0) ? hA : hB;
    // z[C] has landed once at most the D - 1 - C younger loads are outstanding (loads return in order).  From column 1
    // on u[0] rides along as an in/out operand: the wait then sits between column C-1's and column C's update of u[0]
    // and cannot be hoisted above earlier FMAs (a hoisted wait would wait for the youngest load far too early).
    if constexpr (C == 0) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(z[0]) : "n"(D - 1));
#pragma unroll
      for (int rr = 0; rr < M; ++rr) u[rr] = FROM_ZERO ? h[rr] * z[0] : fma(h[rr], z[0], u0v[rr]);
    } else {
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(z[C]), "+v"(u[0]) : "n"(D - 1 - C));
#pragma unroll
      for (int rr = 0; rr < M; ++rr) u[rr] = fma(h[cc * M + rr], z[C], u[rr]);
    }
    if constexpr (cc == 0 && g >= 1 && g + 1 < NG) {
      double (&hnext)[GS] = (g % 2 == 0) ? hB : hA;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < GS; ++j) hnext[j] = hp[(g + 1) * GS + j];
      __builtin_amdgcn_sched_barrier(0);
    }
    pipe_psi_cols<D, M, GC, FROM_ZERO, C + 1>(u, z, hA, hB, u0v, hp);
  }
}

// accumulation phase, row C of the packed second moment; behind it the NEXT step's z[C] is loaded into the same register
// MIRROR: cw = c(z) + c(-z) feeds the (even) second moment, cm = c(z) - c(-z) the (odd) first moment
template <int D, int BIAS, bool MIRROR, int C = 0>
__device__ __forceinline__ void pipe_acc_rows(double (&acc)[(D + 1) * (D + 2) / 2], double (&z)[D], const double cw, const double cm,
                                              const unsigned voff, const char* const Zq) {
  if constexpr (C < D) {
    constexpr int q0 = 1 + D + C * D - C * (C - 1) / 2;         // packed index of (C, C)
    const double tc = cw * z[C];
    if constexpr (MIRROR) acc[1 + C] = fma(cm, z[C], acc[1 + C]);
    else acc[1 + C] += tc;
#pragma unroll
    for (int e = C; e < D; ++e) acc[q0 + e - C] = fma(tc, z[e], acc[q0 + e - C]);
    __builtin_amdgcn_sched_barrier(0);
    GVI_ZLOAD(z[C], voff, Zq, C * 512 - BIAS);                  // row C was the last reader of z[C]
    __builtin_amdgcn_sched_barrier(0);
    pipe_acc_rows<D, BIAS, MIRROR, C + 1>(acc, z, cw, cm, voff, Zq);
  }
}

template <int D, int M, bool SIGNED, bool MIRROR>
__device__ __forceinline__ void sreg_pipe_body(const MomArgs& a, const int bx, const int by_, double* usb, double* redb, const int kfix = -1,
                                               const int byfix = 0) {
  constexpr int NP = (D + 1) * (D + 2) / 2;
  constexpr int NB = (NP + 15) / 16;
  // columns per SGPR operand group: the grouping of split_psi_rows (same operand traffic, same SGPR budget)
  constexpr int GC = (D % 4 == 0 && M <= 3) ? 4 : (D % 3 == 0 && M == 6 ? 3 : (D % 2 == 0 && M <= 6 ? 2 : 1));
  constexpr int NG = D / GC, GS = GC * M;
  constexpr int TB = (D + 1) * 512;                  // bytes of one 64-point tile of Zq
  constexpr int BIAS = (TB / 2) & ~7;                // centres the row offsets in the signed 13-bit immediate
  static_assert(D * 512 - BIAS <= 4095 && -BIAS >= -4096, "row offsets must fit the immediate field");
  static_assert(D <= 16, "vmcnt immediates");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* us_w = usb + wave * 2 * M;               // [2 M] u0 | sgn of this wave's factor
  double (*red_w)[65] = (double (*)[65])(redb + wave * 16 * 65);
  const int kq = kfix >= 0 ? kfix : bx * 4 + wave;            // (kfix: see reg_body)
  const int by = kfix >= 0 ? byfix : by_;
  const bool active = kq < a.f.K && by < a.nchunk;
  const int k = __builtin_amdgcn_readfirstlane(active ? kq : a.f.K - 1);   // inactive waves redo the last factor
  if (lane < M) {
    us_w[lane] = a.f.u0[(size_t)k * M + lane];
    us_w[M + lane] = a.f.sgn[(size_t)k * M + lane];
  }
  __syncthreads();
  const uint64_t hbase = (uint64_t)(a.f.H + (size_t)k * M * D);
  cdouble_t* const hq = (cdouble_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(hbase >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)hbase));
  double acc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) acc[j] = 0.0;
  // unpaired: u0.  Mirror pairs: s_r u0_r (the odd moment's weights) and k0 = sum_r s_r u0_r^2 (see below)
  double u0v[M], k0 = 0.0;
#pragma unroll
  for (int r = 0; r < M; ++r) {
    u0v[r] = MIRROR ? us_w[M + r] * us_w[r] : us_w[r];
    if (MIRROR) k0 = fma(u0v[r], us_w[r], k0);
  }
  const int64_t Np = MIRROR ? a.f.Nmp : a.f.Np;
  const int64_t ck = MIRROR ? a.mchunk : a.chunk;
  const int64_t i0 = (int64_t)by * ck;
  const int64_t i1 = (i0 + ck < Np) ? i0 + ck : Np;
  const int ntiles = i1 > i0 ? (int)((i1 - i0) >> 6) : 0;   // chunks are whole 64-point tiles
  // (through readfirstlane, as hq: the loads below take the base as an SGPR operand of their asm, and inside a larger kernel
  // -- factor_block3_kernel with the symmetric-root products inlined beside it -- the compiler no longer proved it uniform)
  const uint64_t zbase = (uint64_t)(MIRROR ? a.f.Zm : a.f.Zq);
  const char* const Zq = (const char*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(zbase >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)zbase));
  unsigned voff = (unsigned)(i0 >> 6) * (unsigned)TB + (unsigned)lane * 8u + (unsigned)BIAS;
  unsigned idx = (unsigned)(i0 + lane);
  const unsigned nvalid = (unsigned)(MIRROR ? a.f.Nm : a.f.N);
  double z[D], wi;
  double hA[GS], hB[GS];                             // operand groups of even / odd index (SGPRs)
  if (ntiles > 0) {
    // prologue: the first step's loads in the order the psi phase waits for them (w first), and its first two groups
    GVI_ZLOAD(wi, voff, Zq, D * 512 - BIAS);
    pipe_zload_all<D, BIAS>(z, voff, Zq);
    cdouble_t* hp = hq;
    asm volatile("" : "+s"(hp));
#pragma unroll
    for (int j = 0; j < GS; ++j) hA[j] = hp[j];
    if (NG > 1) {
#pragma unroll
      for (int j = 0; j < GS; ++j) hB[j] = hp[GS + j];
    }
  }
  for (int t = 0; t < ntiles; ++t) {
    cdouble_t* hp = hq;
    asm volatile("" : "+s"(hp));                      // opaque per step: the operand loads stay inside the loop
    double u[M];
    pipe_psi_cols<D, M, GC, MIRROR>(u, z, hA, hB, u0v, hp);
    double psi = 0.0;
    if constexpr (SIGNED) {
#pragma unroll
      for (int rr = 0; rr < M; ++rr) psi = fma(us_w[M + rr] * u[rr], u[rr], psi);
    } else {
#pragma unroll
      for (int rr = 0; rr < M; ++rr) psi = fma(u[rr], u[rr], psi);
    }
    asm volatile("" : "+v"(wi));                       // w was issued before z[0]: it has landed with z[0]'s wait
    double cw, cm = 0.0;
    if constexpr (MIRROR) {
      // +-pair from ONE v = H z (u holds v, psi holds q = sum_r s_r v_r^2):  psi(+-z) = sum_r s_r (u0_r +- v_r)^2, so
      //   c+ = w (psi(z) + psi(-z)) = 2 w (q + k0),          k0 = sum_r s_r u0_r^2  (per factor)
      //   c- = w (psi(z) - psi(-z)) = 4 w sum_r (s_r u0_r) v_r                       (no cancellation)
      double l = 0.0;
#pragma unroll
      for (int rr = 0; rr < M; ++rr) l = fma(u0v[rr], u[rr], l);
      const bool ok = idx < nvalid;                    // pad rows carry w = 0, z = 0 (finite values: the select is for safety)
      const double w2 = wi + wi;
      cw = ok ? w2 * (psi + k0) : 0.0;
      cm = ok ? (w2 + w2) * l : 0.0;
    } else {
      cw = idx < nvalid ? wi * psi : 0.0;              // pad rows carry w = 0, z = 0; the select keeps a NaN psi out
    }
    // next step (the last step re-reads its own tile: no out-of-range address, the values are never used)
    const bool more = t + 1 < ntiles;
    voff += more ? (unsigned)TB : 0u;
    idx += more ? 64u : 0u;
    __builtin_amdgcn_sched_barrier(0);
    GVI_ZLOAD(wi, voff, Zq, D * 512 - BIAS);
    // the next step's first two operand groups: the whole accumulation phase to land
#pragma unroll
    for (int j = 0; j < GS; ++j) hA[j] = hp[j];
    if (NG > 1) {
#pragma unroll
      for (int j = 0; j < GS; ++j) hB[j] = hp[GS + j];
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[0] += cw;
    pipe_acc_rows<D, BIAS, MIRROR>(acc, z, cw, cm, voff, Zq);
  }
  // drain: nothing below may reuse z / w registers while the (unused) loads of the last step are in flight
  if (ntiles > 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wi));
#pragma unroll
    for (int c = 0; c < D; ++c) asm volatile("" : "+v"(z[c]));
  }
  double* out = a.partial + ((size_t)k * a.nchunk + by) * NP;
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (bb * 16 + j < NP) red_w[j][lane] = acc[bb * 16 + j];
    wave_lds_sync();
    const int j = lane & 15, part = lane >> 4;
    double s = 0.0;
#pragma unroll
    for (int t2 = 0; t2 < 16; ++t2) s += red_w[j][part * 16 + t2];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (active && lane < 16 && bb * 16 + lane < NP) out[bb * 16 + lane] = s;
    wave_lds_sync();
  }
}

// every residual row of the set has positive weight (sgn = +1): the sign multiply of psi is dropped
template <int D, int M>
__device__ __forceinline__ void sreg_pipe_dispatch(const MomArgs& a, const int bx, const int by, double* usb, double* redb, const int kfix = -1,
                                                   const int byfix = 0) {
  if (a.f.Zm) {
    if (a.f.all_pos) sreg_pipe_body<D, M, false, true>(a, bx, by, usb, redb, kfix, byfix);
    else sreg_pipe_body<D, M, true, true>(a, bx, by, usb, redb, kfix, byfix);
  } else {
    if (a.f.all_pos) sreg_pipe_body<D, M, false, false>(a, bx, by, usb, redb, kfix, byfix);
    else sreg_pipe_body<D, M, true, false>(a, bx, by, usb, redb, kfix, byfix);
  }
}

// PIPE selects the hand-pipelined body for the full pass (the cost pass has its own kernels)
// (256, 2): two blocks per CU = two waves per SIMD = at most 256 registers per lane INCLUDING AGPRs; without the second
// argument the compiler may take 256 + spill AGPRs and silently halve the occupancy
template <int D, int M, bool FULL, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void moments_sreg_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  __shared__ double us[4 * 2 * M];
  __shared__ double red[4 * 16 * 65];
  if constexpr (FULL && PIPE) sreg_pipe_dispatch<D, M>(a, blockIdx.x, blockIdx.y, us, red);
  else sreg_body<D, M, FULL>(a, blockIdx.x, blockIdx.y, us, red);
}

// Two sets in one launch (the chain pattern: binary priors d = 2n and unary factors d = n).  The launches are
// independent, so fusing them removes one dependent-launch boundary (~6 us on this part) and lets the small set's
// blocks fill the tail of the large one.  Blocks [0, nb0) belong to set 0 (x fastest), the rest to set 1.
template <int D0, int M0, int D1, int M1, bool FULL, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void moments_sreg_pair_kernel(MomArgs a0, MomArgs a1, int nbx0, int nb0, int nbx1) {
  if (pred_skip(a0.pred, a0.pred_val)) return;
  constexpr int MM = M0 > M1 ? M0 : M1;
  __shared__ double us[4 * 2 * MM];
  __shared__ double red[4 * 16 * 65];
  const int b = blockIdx.x;
  if constexpr (FULL && PIPE) {
    // every block takes a prior item and then (strided) unary items, so the short unary blocks do not form a serial tail
    // behind the single resident round of prior blocks; the grid is max(nb0, nb1)
    if (b < nb0) sreg_pipe_dispatch<D0, M0>(a0, b % nbx0, b / nbx0, us, red);
    const int nb1 = nbx1 * a1.nchunk;
    for (int it = b; it < nb1; it += gridDim.x) {
      __syncthreads();
      sreg_pipe_dispatch<D1, M1>(a1, it % nbx1, it / nbx1, us, red);
    }
  } else {
    if (b < nb0) sreg_body<D0, M0, FULL>(a0, b % nbx0, b / nbx0, us, red);
    else sreg_body<D1, M1, FULL>(a1, (b - nb0) % nbx1, (b - nb0) / nbx1, us, red);
  }
}

// The planning graph of the reference's own GPU workload (helpers/CudaOperation.cu:74-119: minimum-acceleration priors d = 8,
// hinge-on-SDF obstacle factors d = 4, two fixed-prior anchors d = 4) as ONE moments launch: the three sets are independent,
// three launches were three dependent kernel boundaries (5.3 + 18.8 + 4.5 us, profiles/r03_f_kernel_stats_planar1k.csv)
// around one 18.8 us kernel.  Blocks [0, nb1) = the obstacle set (the heavy one first), [nb1, nb1 + nb0) = the priors, the rest
// the anchors; every block runs its set's body unchanged (bit-identical to the three launches).
template <bool FULL>
__global__ __launch_bounds__(256, 2) void moments_planar3_kernel(MomArgs a0, MomArgs a1, MomArgs a2, int nbx0, int nb0, int nbx1, int nb1, int nbx2,
                                                                  int pipe) {
  if (pred_skip(a0.pred, a0.pred_val)) return;
  using PsiH = PsiHingeSdf<4, KIND_HINGE_SDF_2D>;
  using PsiA = PsiQuad<4, 4>;
  constexpr int HS = PsiH::LDS > PsiA::LDS ? (PsiH::LDS > 2 * 4 ? PsiH::LDS : 2 * 4) : (PsiA::LDS > 2 * 4 ? PsiA::LDS : 2 * 4);
  __shared__ double hs[4 * HS];
  __shared__ double red[4 * 16 * 65];
  const int b = (int)blockIdx.x;
  if (b < nb1) reg_body<4, PsiH, FULL>(a1, b % nbx1, b / nbx1, hs, red);
  else if (b < nb1 + nb0) {
    // (the body the set's own launch would run: the hand-pipelined one for the full pass when the tile-major table is there)
    if constexpr (FULL) {
      if (pipe) sreg_pipe_dispatch<8, 4>(a0, (b - nb1) % nbx0, (b - nb1) / nbx0, hs, red);
      else sreg_body<8, 4, true>(a0, (b - nb1) % nbx0, (b - nb1) / nbx0, hs, red);
    } else sreg_body<8, 4, false>(a0, (b - nb1) % nbx0, (b - nb1) / nbx0, hs, red);
  }
  else reg_body<4, PsiA, FULL>(a2, (b - nb1 - nb0) % nbx2, (b - nb1 - nb0) / nbx2, hs, red);
}

// ---------------------------------------------------------------------------------------------
// moments_scost_kernel<D, M, F>: cost pass (m0 only) of the sum-of-squares kinds with SGPR operands and F
// factors per wave -- one load of the 64 sigma points feeds F psi evaluations, so the L1/L2 traffic per
// evaluation drops by F (the cost pass executes only M D + M FMAs per point and is otherwise bound by the
// 8-byte-per-lane table loads).  Block = 4 waves = 4 F consecutive factors over the same range of points.
// ---------------------------------------------------------------------------------------------
// Cost pass on the mirror-half table.  For a +-pair the cross terms of the two evaluations cancel:
//   psi(z) + psi(-z) = sum_r s_r [(u0_r + v_r)^2 + (u0_r - v_r)^2] = 2 (sum_r s_r v_r^2 + sum_r s_r u0_r^2),   v = H z,
// so a pair costs ONE H z (started from zero) and one sum of squares; k0 = sum_r s_r u0_r^2 is a per-factor constant.
// (The origin is stored with half its weight: v = 0 there and the pair formula returns w psi(0).)
template <int D, int M, int F>
__device__ __forceinline__ void scost_mirror_body(const MomArgs& a, const int bx, const int by, double* usb) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double (*us_w)[3 * M] = (double (*)[3 * M])(usb + wave * F * 3 * M);     // [F][zeros | sgn | unused]
  const int k0i = (bx * 4 + wave) * F;
  cdouble_t* hq[F];
  double kc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const int k = __builtin_amdgcn_readfirstlane(k0i + f < a.f.K ? k0i + f : a.f.K - 1);
    double kk = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) { const double u = a.f.u0[(size_t)k * M + r]; kk = fma(a.f.sgn[(size_t)k * M + r] * u, u, kk); }
    kc[f] = kk;
    if (lane < M) {
      us_w[f][lane] = 0.0;
      us_w[f][M + lane] = a.f.sgn[(size_t)k * M + lane];
    }
    const uint64_t hbase = (uint64_t)(a.f.H + (size_t)k * M * D);
    hq[f] = (cdouble_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(hbase >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)hbase));
  }
  __syncthreads();
  double acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0;
  const int64_t Np = a.f.Nmp;
  const int64_t i0 = (int64_t)by * a.mchunk;
  const int64_t i1 = (i0 + a.mchunk < Np) ? i0 + a.mchunk : Np;
  const double* __restrict__ Zm = a.f.Zm;
  for (int64_t base = i0; base < i1; base += 64) {             // wave-uniform loop over whole 64-pair tiles
    const double* tile = Zm + (size_t)(base >> 6) * (D + 1) * 64 + lane;
    double z[D];
#pragma unroll
    for (int c = 0; c < D; ++c) z[c] = tile[c * 64];
    const double w2 = 2.0 * tile[D * 64];                      // pad lanes: w = 0, z = 0
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const double q = split_psi_rows<D, M>(hq[f], us_w[f], us_w[f] + M, z);   // sum_r s_r (H z)_r^2
      acc[f] = fma(w2, q + kc[f], acc[f]);
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) {
    double s = acc[f];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && k0i + f < a.f.K) a.partial[(size_t)(k0i + f) * a.nchunk + by] = s;
  }
}

template <int D, int M, int F>
__device__ __forceinline__ void scost_body(const MomArgs& a, const int bx, const int by, double* usb) {
  if (a.f.Zm) { scost_mirror_body<D, M, F>(a, bx, by, usb); return; }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double (*us_w)[2 * M] = (double (*)[2 * M])(usb + wave * F * 2 * M);     // [F][2 M]
  const int k0 = (bx * 4 + wave) * F;
  cdouble_t* hq[F];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const int k = __builtin_amdgcn_readfirstlane(k0 + f < a.f.K ? k0 + f : a.f.K - 1);
    if (lane < M) {
      us_w[f][lane] = a.f.u0[(size_t)k * M + lane];
      us_w[f][M + lane] = a.f.sgn[(size_t)k * M + lane];
    }
    const uint64_t hbase = (uint64_t)(a.f.H + (size_t)k * M * D);
    hq[f] = (cdouble_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(hbase >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)hbase));
  }
  __syncthreads();
  double acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0;
  const int64_t Np = a.f.Np;
  const int64_t i0 = (int64_t)by * a.chunk;
  const int64_t i1 = (i0 + a.chunk < Np) ? i0 + a.chunk : Np;
  const double* __restrict__ Zt = a.f.Zt;
  const double* __restrict__ w = a.f.w;
  for (int64_t base = i0; base < i1; base += 64) {             // wave-uniform loop (chunks are whole 64-point tiles)
    const int64_t i = base + lane;
    double z[D];
#pragma unroll
    for (int c = 0; c < D; ++c) z[c] = Zt[(size_t)c * Np + i];
    const double wi = i < a.f.N ? w[i] : 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const double psi = split_psi_rows<D, M>(hq[f], us_w[f], us_w[f] + M, z);
      acc[f] += i < a.f.N ? wi * psi : 0.0;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) {
    double s = acc[f];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && k0 + f < a.f.K) a.partial[(size_t)(k0 + f) * a.nchunk + by] = s;
  }
}

template <int D, int M, int F>
__global__ __launch_bounds__(256) void moments_scost_kernel(MomArgs a) {
  if (pred_skip(a.pred, a.pred_val)) return;
  __shared__ double us[4 * F * 3 * M];
  scost_body<D, M, F>(a, blockIdx.x, blockIdx.y, us);
}

template <int D0, int M0, int D1, int M1, int F>
__global__ __launch_bounds__(256) void moments_scost_pair_kernel(MomArgs a0, MomArgs a1, int nbx0, int nb0, int nbx1) {
  if (pred_skip(a0.pred, a0.pred_val)) return;
  constexpr int MM = M0 > M1 ? M0 : M1;
  __shared__ double us[4 * F * 3 * MM];
  const int b = blockIdx.x;
  if (b < nb0) scost_body<D0, M0, F>(a0, b % nbx0, b / nbx0, us);
  else scost_body<D1, M1, F>(a1, (b - nb0) % nbx1, (b - nb0) / nbx1, us);
}

// ---------------------------------------------------------------------------------------------
// epilogue_kernel: one wave per factor.
// ---------------------------------------------------------------------------------------------
struct EpiArgs {
  FactorDev f;
  const double* partial;   // [K][nchunk][npo]
  int nchunk, full;
  double* Ephi;            // [K] or null
  double* cost;            // [K] or null   (E[psi] / T_k)
  double* Vdmu;            // [K][d] or null
  double* Vddmu;           // [K][d][d] or null
  double* E_xmuphi;        // [K][d] or null       raw integrals
  double* E_xxphi;         // [K][d][d] or null
};

// LDS doubles of the epilogue: packed moments, M2, one product, and the factor's Sinv / Lam (or S) staged once
__host__ __device__ inline size_t epilogue_lds_doubles(int d) { return (size_t)npairs(d) + 4 * (size_t)d * d; }

// returns the factor's cost E[psi] / T_k (wave-uniform).
// DT = d at compile time (inner products fully unrolled: their LDS loads pipeline) or 0 (any d).  The operands of the
// back-transform (Sinv, Lam; S for the raw integrals) are fetched into LDS in one round trip at the top -- with the
// loads inside the d-long inner loops every product paid d dependent global round trips (profiles/r02_e: 15 us for
// what is ~2 us of arithmetic).  Summation orders are unchanged (c ascending, chunk index ascending).
// phase 0: everything; 1: chunk sums, E[psi] and cost only (Ms stays in LDS); 2: the back-transform after a phase-1 call
// P: the factor's chunk partials ([nchunk][npo]): a.partial + k nchunk npo, or LDS (factor_fused_kernel)
// Zs: S^-T of this factor already in LDS (factor_fused_kernel, Cholesky route; Sigma^-1 = S^-T S^-1 is then re-formed
// here by the prep's own sequence of operations: bit-identical) or null (both fetched here)
template <int DT>
__device__ __forceinline__ double epilogue_body_t(const EpiArgs& a, int k, double* sm, int phase, const double* P, double* Zs = nullptr) {
  const FactorDev& f = a.f;
  const int d = DT ? DT : f.d, dd = d * d, lane = threadIdx.x & 63;      // one wave per factor (the fused kernel runs several per block)
  const int npo = a.full ? npairs(d) : 1;
  double* Ms = sm;              // [npo]
  double* M2 = Ms + npairs(d);  // [d][d]
  double* Tm = M2 + dd;         // [d][d]
  double* Sv = Zs ? Zs : Tm + dd;           // [d][d] Sinv
  double* Lv = Tm + 2 * dd;                 // [d][d] Lam (unused when Zs)
  const double Tk = f.temperature[k];                     // issued with the other loads, used after the chunk sums
  const bool want_v = a.full && (a.Vdmu || a.Vddmu);
  if (want_v && phase != 2 && !Zs) {
    const double* Sinv = f.Sinv + (size_t)k * dd;
    const double* Lam = f.Lam + (size_t)k * dd;
    for (int e = lane; e < dd; e += 64) { Sv[e] = Sinv[e]; Lv[e] = Lam[e]; }
  }
  for (int j = lane; j < npo && phase != 2; j += 64) {
    double s = 0.0;
    for (int c0 = 0; c0 < a.nchunk; c0 += 8) {                   // eight loads in flight, then the ordered sum
      double pv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) pv[q] = c0 + q < a.nchunk ? P[(size_t)(c0 + q) * npo + j] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += pv[q];                    // fixed order (x + 0.0 == x): deterministic
    }
    Ms[j] = s;
  }
  wave_lds_sync();
  const double m0 = Ms[0];
  const double costk = m0 / Tk;
  if (lane == 0 && phase != 2) {
    if (a.Ephi) a.Ephi[k] = m0;
    if (a.cost) a.cost[k] = costk;
  }
  if (!a.full || phase == 1) return costk;
  for (int e = lane; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    M2[e] = i <= j ? Ms[pair_index(d, i, j)] : Ms[pair_index(d, j, i)];
  }
  wave_lds_sync();
  if (a.Vdmu) {
    for (int i = lane; i < d; i += 64) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) s += Sv[i * d + c] * Ms[1 + c];
      a.Vdmu[(size_t)k * d + i] = s / Tk;
    }
  }
  if (a.Vddmu) {
    for (int e = lane; e < dd; e += 64) {
      const int i = e / d, j = e % d;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) s += Sv[i * d + c] * M2[c * d + j];
      Tm[e] = s;
    }
    wave_lds_sync();
    for (int e = lane; e < dd; e += 64) {
      const int i = e / d, j = e % d;
      if (i <= j) {           // upper triangle, mirrored (ngd/NGDFactorizedBaseGH.h:71-72)
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < d; ++c) s += Tm[i * d + c] * Sv[j * d + c];     // (W M2) W^T, W = S^-T (symmetric root: W = Sinv)
        double lam;
        if (Zs) {
          lam = 0.0;
#pragma unroll
          for (int c = 0; c < d; ++c) lam = fma(Sv[i * d + c], Sv[j * d + c], lam);   // prep_chol_body: Sigma^-1 = X^T X
        } else lam = Lv[i * d + j];
        const double v = (s - lam * m0) / Tk;
        a.Vddmu[(size_t)k * dd + i * d + j] = v;
        a.Vddmu[(size_t)k * dd + j * d + i] = v;
      }
    }
    wave_lds_sync();
  }
  if (a.E_xmuphi || a.E_xxphi) {
    const double* S = f.S + (size_t)k * dd;
    for (int e = lane; e < dd; e += 64) Sv[e] = S[e];            // Sinv is dead: reuse its slot
    wave_lds_sync();
  }
  if (a.E_xmuphi) {
    for (int i = lane; i < d; i += 64) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) s += Sv[i * d + c] * Ms[1 + c];
      a.E_xmuphi[(size_t)k * d + i] = s;
    }
  }
  if (a.E_xxphi) {
    for (int e = lane; e < dd; e += 64) {
      const int i = e / d, j = e % d;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) s += Sv[i * d + c] * M2[c * d + j];
      Tm[e] = s;
    }
    wave_lds_sync();
    for (int e = lane; e < dd; e += 64) {
      const int i = e / d, j = e % d;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < d; ++c) s += Tm[i * d + c] * Sv[j * d + c];
      a.E_xxphi[(size_t)k * dd + e] = s;
    }
  }
  return costk;
}

// the chain shapes of BASELINE.json get unrolled instances; everything else runs the runtime-d body
__device__ __forceinline__ double epilogue_body_p(const EpiArgs& a, int k, double* sm, int phase, const double* P, double* Zs = nullptr) {
  switch (a.f.d) {
    case 2: return epilogue_body_t<2>(a, k, sm, phase, P, Zs);
    case 4: return epilogue_body_t<4>(a, k, sm, phase, P, Zs);
    case 6: return epilogue_body_t<6>(a, k, sm, phase, P, Zs);
    case 8: return epilogue_body_t<8>(a, k, sm, phase, P, Zs);
    case 12: return epilogue_body_t<12>(a, k, sm, phase, P, Zs);
    default: return epilogue_body_t<0>(a, k, sm, phase, P, Zs);
  }
}
__device__ inline double epilogue_body(const EpiArgs& a, int k, double* sm, int phase = 0) {
  return epilogue_body_p(a, k, sm, phase, a.partial + (size_t)k * a.nchunk * (a.full ? npairs(a.f.d) : 1));
}

__global__ __launch_bounds__(64) void epilogue_kernel(EpiArgs a) {
  extern __shared__ double sm[];
  epilogue_body(a, blockIdx.x, sm);
}

struct EpiList {
  int nsets;
  int koff[MAX_FSETS + 1];
  EpiArgs e[MAX_FSETS];
};

// Tail of a cost pass in ONE launch: per-factor cost = (ordered chunk sum of m0) / T_k for every set, the ordered
// sum over all factors (same order as cost_sum_all_kernel: thread-strided partial sums, then a fixed 256-leaf tree
// per set) into acc[0], and -- single-process iteration only -- the publish into host-mapped memory
// ({cost_sum, half_logdet, sequence}).  Replaces epilogue_all_kernel(full = 0) -> cost_sum_all_kernel -> publish_kernel.
__global__ __launch_bounds__(256) void cost_tail_kernel(EpiList L, double* acc, const double* half_logdet, double* host_out,
                                                        double seq, unsigned* counter) {
  __shared__ double sh[256];
  __shared__ int last;
  // phase 1 (all blocks): per-factor costs to global, one factor per thread
  const int gtid = (int)blockIdx.x * 256 + threadIdx.x, gthreads = (int)gridDim.x * 256;
  for (int si = 0; si < L.nsets; ++si) {
    const EpiArgs& e = L.e[si];
    for (int k = gtid; k < e.f.K; k += gthreads) {
      const double* P = e.partial + (size_t)k * e.nchunk;
      double m0 = 0.0;
      for (int c0 = 0; c0 < e.nchunk; c0 += 8) {                   // eight loads in flight, then the ordered sum
        double p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = c0 + j < e.nchunk ? P[c0 + j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) m0 += p[j];                    // fixed order (x + 0.0 == x): deterministic
      }
      e.cost[k] = m0 / e.f.temperature[k];
    }
  }
  // the block that arrives last does the (deterministic, fixed-order) reduction over all factors
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // phase 2: the summation order of cost_sum_all_kernel (thread-strided partial sums, 256-leaf tree per set)
  double total = 0.0;
  for (int si = 0; si < L.nsets; ++si) {
    const EpiArgs& e = L.e[si];
    const volatile double* cost = e.cost;
    double s = 0.0;
    for (int k = threadIdx.x; k < e.f.K; k += 256) s += cost[k];
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
    *counter = 0u;                       // ready for the next launch (stream-ordered)
    acc[0] = total;
    if (host_out) {
      publish_to_host(host_out, total + half_logdet[0], seq);
    }
  }
}

// Optional tail of the fused-trial iteration: the block that finishes LAST sums the factor costs of every set (the
// association of cost_sum_all_kernel: 256 strided partial sums, then a fixed 256-leaf tree per set -- bit-identical) and
// publishes {cost sum, half log-det, sequence} to host-mapped memory.  Replaces two dependent launches
// (cost_sum_all_kernel, publish_kernel) behind the epilogue.
constexpr int EPI_GROUP = 64;
struct EpiTail {
  int on;
  double* acc;                 // [1] cost sum
  const double* half_logdet;   // [1]
  double* host_out;            // host-mapped {cost_sum, half_logdet, sequence} or null
  double seq;
  unsigned* counter;           // [32 (1 + ceil(blocks / EPI_GROUP))] arrival counters, 128 B apart, zero before the launch
  // pipelined iterations (gvi_ngd_run): the accept decision is ALSO taken on the device, so that the next iteration's
  // launches -- queued before the host has read this cost -- can be predicated on it.  accept[0] <- seq if
  // cost < current cost else 0; cost_dev[slot_trial] <- cost; the current cost is c0_imm or cost_dev[slot_cur].
  const double* pred;          // predicate of THIS launch (device_common.hpp, pred_skip) or null
  double pred_val;
  double* accept;              // null: no device-side decision
  double* cost_dev;            // [2] cost of the state in slot 0 / 1
  int slot_cur, slot_trial;
  int c0_use_imm;
  double c0_imm;
  // option "safe_publish": the arrival counters carry release / acquire order (agent scope) instead of the fence-free
  // protocol (write-through store + vmcnt(0) + relaxed counters) -- an L2 write-back per arriving block, the price list of
  // DESIGN 4.2 -- and the publish takes the checked four-word form (device_common.hpp)
  int safe;
};

// The factor costs of every set, as the tail sums them
struct CostList {
  int nsets;
  double* cost[MAX_FSETS];
  int K[MAX_FSETS];
};

// Tail protocol, executed by ONE wave of every block (lane = its lane index) after the block's factor cost `costk` is known:
// store it, count the block in; the one that arrives last sums the costs of all sets and publishes.  `last`: one LDS int
// private to the calling wave.
// Cross-block hand-over WITHOUT device-scope fences: on this multi-XCD part a release fence writes the XCD's whole L2
// back (the epilogue has just dirtied megabytes: 2049 blocks x __threadfence() cost ~45 us, measured).  Only the
// factor's cost has to be seen by the last block, so it is stored write-through at agent scope, the wave waits for
// that one store (vmcnt), and the arrival counters are relaxed agent-scope atomics.  Everything else the epilogue
// wrote is consumed by later launches (ordinary end-of-kernel release).
__device__ __forceinline__ void epi_tail_arrive(const CostList& cl, const EpiTail& tail, double* cost_slot, const double costk, const int lane,
                                                const unsigned bid, const unsigned nblocks, int* last) {
  if (lane == 0) {
    __hip_atomic_store(cost_slot, costk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // two-level arrival count: thousands of atomics on ONE address serialise in L2; groups of EPI_GROUP blocks count on
    // their own 128-byte-spaced word, the last of each group counts on the top word
    const unsigned grp = bid / EPI_GROUP, ngrp = (nblocks + EPI_GROUP - 1) / EPI_GROUP;
    const unsigned in_grp = grp + 1 < ngrp ? (unsigned)EPI_GROUP : nblocks - grp * EPI_GROUP;
    unsigned* gc = tail.counter + 32u * (1u + grp);
    int l = 0;
    const bool lastg = tail.safe ? __hip_atomic_fetch_add(gc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1
                                 : __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1;
    if (lastg) {
      __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      // (a single group -- <= EPI_GROUP factors, BASELINE configs[1] -- needs no second level: one atomic round trip less
      // on the chain that ends in the publish)
      l = ngrp == 1 ? 1
                    : (tail.safe ? __hip_atomic_fetch_add(tail.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1
                                 : __hip_atomic_fetch_add(tail.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1);
    }
    *last = l;
  }
  wave_lds_sync();
  if (!*last) return;
  double total = 0.0;
  // The summation order is that of cost_sum_all_kernel (256 strided partial sums, then a fixed 256-leaf tree per set).
  // All loads of a set's first 8 x 256 factors (per lane: 4 virtual threads x 8) are issued before the first add: the
  // tail is one wave on the critical path of the iteration, and one round trip per virtual thread cost ~6 us.
  for (int s2 = 0; s2 < cl.nsets; ++s2) {
    const double* cost = cl.cost[s2];
    const int K = cl.K[s2];
    double p[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = lane + 64 * j + q * 256;
        p[j][q] = k < K ? __hip_atomic_load(cost + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
    double vs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // virtual thread v = lane + 64 j of the 256-thread kernel
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += p[j][q];         // ordered (x + 0.0 == x)
      for (int k0 = lane + 64 * j + 8 * 256; k0 < K; k0 += 8 * 256) {   // K > 2048: further batches of eight
        double pp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = k0 + q * 256;
          pp[q] = k < K ? __hip_atomic_load(cost + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += pp[q];
      }
      vs[j] = s;
    }
    // the 256-leaf tree sh[v] += sh[v + w], w = 128 ... 1, with the leaves in registers: the first two levels pair the
    // virtual threads of one lane, the rest are lane shuffles -- the same association as the LDS tree it replaces
    double t = (vs[0] + vs[2]) + (vs[1] + vs[3]);       // w = 128: v += v + 128 ; w = 64: v += v + 64
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) t += __shfl_down(t, w);
    total += __shfl(t, 0);
  }
  if (lane == 0) {
    __hip_atomic_store(tail.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    tail.acc[0] = total;
    const double c1 = total + tail.half_logdet[0];
    if (tail.host_out) publish_to_host(tail.host_out, c1, tail.seq);
    if (tail.accept) {                                    // same comparison as the host's (NaN compares false: rejected)
      const double c0 = tail.c0_use_imm ? tail.c0_imm : tail.cost_dev[tail.slot_cur];
      tail.cost_dev[tail.slot_trial] = c1;
      tail.accept[0] = c1 < c0 ? tail.seq : 0.0;
    }
  }
  wave_lds_sync();
}

__global__ __launch_bounds__(64) void epilogue_all_kernel(EpiList L, EpiTail tail, CostList cl) {
  extern __shared__ double sm[];
  if (pred_skip(tail.pred, tail.pred_val)) return;
  int si = 0;
  while (si + 1 < L.nsets && (int)blockIdx.x >= L.koff[si + 1]) ++si;
  const int kf = (int)blockIdx.x - L.koff[si];
  if (!tail.on) { epilogue_body(L.e[si], kf, sm); return; }
  // With the tail on, the factor's cost goes out FIRST (phase 1: chunk sums only), the last block to arrive sums and
  // publishes, and only then does every block do its back-transform (phase 2): the host has the trial cost -- and
  // launches the next iteration's chain kernels -- while the epilogue's products and the assemble are still running,
  // instead of ~10 us after them.
  const double costk = epilogue_body(L.e[si], kf, sm, 1);
  __shared__ int last;
  epi_tail_arrive(cl, tail, L.e[si].cost + kf, costk, (int)threadIdx.x, blockIdx.x, gridDim.x, &last);
  epilogue_body(L.e[si], kf, sm, 2);
}

// X[k][a][i] = mu_a + sum_b S_ab z_b[i]  (reference CUDA-path layout [factor][dim][point])
__global__ __launch_bounds__(256) void expand_kernel(FactorDev f, const double* __restrict__ mu,
                                                     double* __restrict__ X) {
  const int k = blockIdx.y, d = f.d;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= f.N) return;
  const double* S = f.S + (size_t)k * d * d;
  for (int r = 0; r < d; ++r) {
    double x = mu[(size_t)k * d + r];
    for (int c = 0; c < d; ++c) x += S[r * d + c] * f.Zt[(size_t)c * f.Np + i];
    X[((size_t)k * d + r) * f.N + i] = x;
  }
}

}  // namespace gvi
