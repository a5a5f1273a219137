// moments_orbit_psi_kernel: the NON-polynomial psi kinds (hinge on a signed-distance field -- planar point robot, planar body,
// 3-D point robot, arm -- and the 1-D range factor; helpers/CudaOperation.h:21-399, 443-771, src/1d_example.cpp:25-35) on the
// sign-orbit form of the sparse Gauss-Hermite table (orbits.hpp; quadrature/SparseGaussHermite.h:197-243 is what it computes).
//
// The lane-per-point kernels treat a sigma point as a dense d-vector; the arm kind (d = 7 / 14) has no register instance at
// all -- 120 moment accumulators do not fit a lane -- and ran on the LDS-bound generic kernel (0.27 ms per launch on the
// 129-factor arm graph).  Here, as in kernels_orbit.hpp, a lane owns the orbits of one support (c_0 < ... < c_{s-1},
// magnitudes m_j, one weight, all 2^s sign images) of ONE factor (wave = factor x chunk of tiles):
//   * psi needs only the first NR coordinates of x = mu + S z (the pose: 2 or 3; the joint angles: 7), and z has s <= 4
//     non-zero coordinates: x moves by ONE column of S per Gray step, NR FMAs instead of NR d;
//   * psi is evaluated at EVERY sign image (no +- identity for a general psi), c = psi(x_sigma);
//   * the z-space moments of the orbit are sign-weighted sums of that scalar,
//         m0 += w sum c,   m1[c_i] += w m_i sum sigma_i c,   M2[c_i][c_j] += w m_i m_j sum sigma_i sigma_j c   (sigma_i^2 = 1),
//     1 + s + s (s - 1) / 2 adds per point instead of the (d + 1)(d + 2) / 2 FMAs of a dense accumulation, added to the
//     factor's accumulators (LDS, private copies, ds_add_f64 in a fixed order: run-to-run bit-identical) once per support.
// Output: the same packed chunk partials [m0 | m1[d] | upper triangle of M2] as every moments kernel, so prep (symmetric
// root) and epilogue (back-transform) are the existing ones.
#pragma once
#include "kernels_orbit.hpp"

namespace gvi {

struct OrbitPsiArgs {
  FactorDev f;               // S [K][d][d] (row-major, symmetric root), raw psi parameters, SDF / arm model
  const double* mu;          // [K][d]
  double* partial;           // [K][nchunk][npairs(d)] (full) or [K][nchunk] (cost)
  int nchunk;
  int copies;                // private copies of every accumulator entry (power of two)
  const double* pred;
  double pred_val;
  OrbitDev ob;
};

// leading coordinates of x that psi reads
__host__ __device__ constexpr int orbit_psi_rows(int kind) {
  return kind == KIND_RANGE_1D ? 1 : kind == KIND_HINGE_SDF_2D ? 2 : kind == KIND_HINGE_SDF_3D_ARM ? 7 : 3;
}

// psi_hinge_sdf3d_arm (kernels_factor.hpp; CudaOperation_3dArm::cost_obstacle + ForwardKinematics, helpers/CudaOperation.h:
// 325-399, 752-771) with the joint angles in REGISTERS: the joints are folded in a compile-time loop (x[i] with a constant
// index), the spheres of frame i follow joint i -- frames are non-decreasing (host-checked), so this is the same sequence of
// operations, sphere by sphere, as the generic form's "advance the chain to the sphere's frame".
template <int NR>
__device__ __forceinline__ double psi_arm_regs(const FactorDev& f, const double* p, const double (&x)[NR], const int d) {
  const double* A = f.arm;
  const int nd = (int)A[0], ns = (int)A[1];
  const double *a = A + 2, *al = a + nd, *dl = al + nd, *tb = dl + nd, *fr = tb + nd, *ce = fr + ns, *ra = ce + 3 * ns;
  double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const int nb = d < ns ? d : ns;
  double cost = 0.0;
  int s = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (i < nd && s < nb) {
      const float th = (float)(x[i] + tb[i]), alf = (float)al[i];
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
      while (s < nb && (int)fr[s] == i) {
        const double cx = ce[3 * s], cy = ce[3 * s + 1], cz = ce[3 * s + 2];
        const double px = T[3] + (T[0] * cx + T[1] * cy + T[2] * cz);
        const double py = T[7] + (T[4] * cx + T[5] * cy + T[6] * cz);
        const double pz = T[11] + (T[8] * cx + T[9] * cy + T[10] * cz);
        cost += hinge_sq(sdf3d_lookup(f, px, py, pz), p[1] + ra[s], 1.0, p[0]);
        ++s;
      }
    }
  }
  return cost;
}

template <int KIND, int NR>
__device__ __forceinline__ double orbit_psi_eval(const FactorDev& f, const double* p, const double (&x)[NR], const int d) {
  if constexpr (KIND == KIND_RANGE_1D) return psi_range_1d(p, x[0]);
  else if constexpr (KIND == KIND_HINGE_SDF_2D) return psi_hinge_sdf2d(f, p, x[0], x[1]);
  else if constexpr (KIND == KIND_HINGE_SDF_2D_BODY) return psi_hinge_sdf2d_body(f, p, x[0], x[1], x[2]);
  else if constexpr (KIND == KIND_HINGE_SDF_3D) return psi_hinge_sdf3d(f, p, x[0], x[1], x[2]);
  else return psi_arm_regs<NR>(f, p, x, d);
}

// LDS doubles per wave: the NR leading rows of S, column-contiguous ([d][NRP]), + the accumulator copies
__host__ __device__ inline int orbit_psi_nrp(int NR) { return (NR + 1) & ~1; }
__host__ __device__ inline int orbit_psi_lds_doubles(int d, int NR, int copies, bool full) {
  return (d * orbit_psi_nrp(NR) + (full ? copies * (d + 1) * (d + 2) / 2 : 0) + 2) & ~1;
}

// the tiles [t0, t1) of the class with support size S
template <int KIND, int NR, int S, bool FULL>
__device__ __forceinline__ void orbit_psi_class(const OrbitPsiArgs& a, const double* par, const int d, const int lc, const int t0, const int t1,
                                                const int tfirst, const uint32_t lane8, const double* Sc, double* accl,
                                                const double (&x0)[NR], double& m0) {
  constexpr int NRP = (NR + 1) & ~1, NV = 2 * S + S * (S - 1) / 2, NPT = 1 << S;
  const OrbitDev& ob = a.ob;
  const int G = ob.cgrp[S];
  const uint32_t gstride8 = (uint32_t)ob.cstride[S] * 8u;
  for (int t = t0; t < t1; ++t) {
    const uint32_t boff0 = ((uint32_t)ob.cbase[S] + (uint32_t)(t - tfirst) * 64u) * 8u + lane8;
    const uint64_t cpk = orbit_ld(ob.cpk, boff0), rpk = orbit_ld(ob.rpk, boff0);
    int c[S];
    double hc[S][NR];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      c[j] = (int)((cpk >> (8 * j)) & 255u);
#pragma unroll
      for (int r = 0; r < NR; ++r) hc[j][r] = Sc[c[j] * NRP + r];
    }
    double acc[FULL ? NV : 1];
#pragma unroll
    for (int q = 0; q < (FULL ? NV : 1); ++q) acc[q] = 0.0;
    for (int g = 0; g < G; ++g) {
      OrbitWm<S> wm;
      orbit_load_wm<S>(ob, boff0 + (uint32_t)g * gstride8, wm);
      // corner (-, ..., -), then all 2^S sign patterns in Gray order
      double x[NR], sg[S];
#pragma unroll
      for (int r = 0; r < NR; ++r) x[r] = x0[r];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        sg[j] = -1.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = fma(-wm.mg[j], hc[j][r], x[r]);
      }
      double E0 = 0.0, Oi[S], Eij[S * (S - 1) / 2 + 1];
#pragma unroll
      for (int j = 0; j < S; ++j) Oi[j] = 0.0;
#pragma unroll
      for (int e = 0; e < S * (S - 1) / 2 + 1; ++e) Eij[e] = 0.0;
      // (not unrolled: the arm's forward kinematics is ~1500 instructions per evaluation)
#pragma clang loop unroll(disable)
      for (int pt = 0; pt < NPT; ++pt) {
        const double psi = orbit_psi_eval<KIND, NR>(a.f, par, x, d);
        E0 += psi;
        if constexpr (FULL) {
          int e = 0;
#pragma unroll
          for (int i = 0; i < S; ++i) {
            const double si = sg[i] * psi;
            Oi[i] += si;
#pragma unroll
            for (int j = i + 1; j < S; ++j) { Eij[e] = fma(sg[j], si, Eij[e]); ++e; }
          }
        }
        if (pt + 1 < NPT) {
          const int jn = __builtin_ctz((unsigned)(pt + 1));
#pragma unroll
          for (int j = 0; j < S; ++j) {
            if (j == jn) {
              sg[j] = -sg[j];
              const double t2 = (sg[j] + sg[j]) * wm.mg[j];
#pragma unroll
              for (int r = 0; r < NR; ++r) x[r] = fma(t2, hc[j][r], x[r]);
            }
          }
        }
      }
      // a padding orbit (w = 0, magnitudes 0) evaluates psi(mu): the select keeps a non-finite value out
      const double w = wm.w;
      const bool live = w != 0.0;
      m0 += live ? w * E0 : 0.0;
      if constexpr (FULL) {
        int n = 0, e = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          const double wmi = w * wm.mg[i];
          acc[n] += live ? wmi * Oi[i] : 0.0; ++n;
          acc[n] += live ? wmi * wm.mg[i] * E0 : 0.0; ++n;
#pragma unroll
          for (int j = i + 1; j < S; ++j) { acc[n] += live ? wmi * wm.mg[j] * Eij[e] : 0.0; ++n; ++e; }
        }
      }
    }
    if constexpr (FULL) {
      const int sh = lc + 3;
      char* const base = (char*)accl;
      char* const base1 = base + (1u << sh);
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
        lds_add_f64((double*)(base1 + B[i]), acc[n++]);
        lds_add_f64((double*)(base + (A[i] + B[i])), acc[n++]);
#pragma unroll
        for (int j = i + 1; j < S; ++j) lds_add_f64((double*)(base + (A[i] + B[j])), acc[n++]);
      }
    }
  }
}

// grid (ceil(K / 4), nchunk) x 256: four waves = four factors on the same chunk of tiles; supports up to four coordinates
// (every table of degree <= 5, every table in d <= 4: the host checks)
template <int KIND, bool FULL>
__global__ __launch_bounds__(256) void moments_orbit_psi_kernel(OrbitPsiArgs a) {
  constexpr int NR = orbit_psi_rows(KIND), NRP = (NR + 1) & ~1;
  extern __shared__ double sm[];
  if (pred_skip(a.pred, a.pred_val)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int k = (int)blockIdx.x * 4 + wave, chunk = (int)blockIdx.y;
  if (k >= a.f.K) return;                        // (no block-level barrier below)
  const OrbitDev& ob = a.ob;
  const int d = a.f.d;
  const int NP = FULL ? (d + 1) * (d + 2) / 2 : 1;
  const int C = a.copies, lc = __builtin_ctz((unsigned)C);
  double* Sc = sm + (size_t)wave * orbit_psi_lds_doubles(d, NR, C, FULL);
  double* accl = Sc + d * NRP;
  // column c of the first NR rows of S (row-major [d][d]): Sc[c][r] = S[r][c]
  const double* Sg = a.f.S + (size_t)k * d * d;
  for (int e = lane; e < NR * d; e += 64) { const int r = e / d, c = e % d; Sc[c * NRP + r] = r < d ? Sg[r * d + c] : 0.0; }
  if (FULL)
    for (int e = lane; e < NP * C; e += 64) accl[e] = 0.0;
  double* accme = accl + (lane & (C - 1));
  double x0[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) x0[r] = r < d ? a.mu[(size_t)k * d + r] : 0.0;
  const double* par = a.f.raw + (size_t)k * a.f.raw_stride;
  wave_lds_sync();
  int tb, te;
  if (ob.nb) { tb = ob.bnd[chunk]; te = ob.bnd[chunk + 1]; }
  else { tb = ob.bounds[chunk]; te = ob.bounds[chunk + 1]; }
  tb = __builtin_amdgcn_readfirstlane(tb);
  te = __builtin_amdgcn_readfirstlane(te);
  const uint32_t lane8 = (uint32_t)lane * 8u;
  double m0 = 0.0;
  int tcur = tb;
#define ORBIT_PSI_CLASS(S_)                                                                                                    \
  {                                                                                                                            \
    const int e_ = te < ob.cend[S_] ? te : ob.cend[S_];                                                                         \
    if (tcur < e_) { orbit_psi_class<KIND, NR, S_, FULL>(a, par, d, lc, tcur, e_, ob.cend[S_ + 1], lane8, Sc, accme, x0, m0); tcur = e_; } \
  }
  ORBIT_PSI_CLASS(4)
  ORBIT_PSI_CLASS(3)
  ORBIT_PSI_CLASS(2)
  ORBIT_PSI_CLASS(1)
#undef ORBIT_PSI_CLASS
  m0 = wave_sum_f64(m0);
  if (chunk == 0 && ob.w0 != 0.0) m0 = fma(ob.w0, orbit_psi_eval<KIND, NR>(a.f, par, x0, d), m0);       // the origin
  wave_lds_sync();
  double* out = a.partial + ((size_t)k * a.nchunk + chunk) * NP;
  if (lane == 0) out[0] = m0;
  if (FULL)
    for (int e = 1 + lane; e < NP; e += 64) {
      const int rot = (lane >> 2) & (C - 1);
      double t = accl[e * C + rot];
      for (int q = 1; q < C; ++q) t += accl[e * C + ((q + rot) & (C - 1))];
      out[e] = t;
    }
}

}  // namespace gvi
