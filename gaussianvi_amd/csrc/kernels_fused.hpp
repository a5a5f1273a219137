// factor_fused_kernel: gather -> per-pass products -> sign-orbit psi walk -> chunk sum -> cost (+ publish tail) ->
// back-transform of ONE factor in ONE workgroup -- the full moments pass of the resident NGD iteration in a single launch.
//
// Replaces, for the sum-of-squares factor sets of a chain, the three launches prep_all_kernel -> moments_orbit_pair_kernel ->
// epilogue_all_kernel (kernels_factor.hpp / kernels_orbit.hpp; reference: GVIFactorizedBase::update_*_from_joint,
// gvibase/GVIFactorizedBase.h:104-122; updateGH + calculate_partial_V, ngd/NGDFactorizedBaseGH.h:50-88).  The device
// functions are the SAME ones those kernels call (prep_body_d, orbit_wave, epilogue_body_p, epi_tail_arrive): results are
// bit-identical to the three-launch route with four chunks per factor.  What changes:
//   * block = the 4 chunks of one factor (wave w walks chunk w) instead of 4 factors x 1 chunk: the chunk partials meet in
//     LDS and are summed there in chunk order -- no partial[K][nchunk][91] round trip through HBM (3.1 of the 4.2 MB the psi
//     launch moved at C3, profiles/r02_traffic.json);
//   * wave 0 forms the factor's products (Cholesky route: ~2 us of dependent work) while the other waves of the CU's
//     resident blocks walk; no kernel boundary between prep, walk and epilogue (two launches and their ramps fewer);
//   * the per-pass products still go to f.S / f.Sinv / f.Lam / f.H / f.u0 as before (a cost-only pass at the same state
//     reuses them), the back-transform reads them back through the CU's L1 / L2.
#pragma once
#include "kernels_orbit.hpp"

namespace gvi {

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
  int koff[3];
  FusedSet s[2];
  // fused gather (see PrepList): factor marginals pulled out of the chain arrays, the trial mean gmu + gstep gdmu formed on
  // the fly; blocks past koff[nsets] write the chain-level trial mean
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

// dynamic LDS (doubles) of a block: the four waves' walk regions (aliased by wave 0's prep area before and its epilogue
// area after), then the four chunk partials
__host__ __device__ inline size_t fused_lds_doubles(int d, int M, int copies) {
  const size_t dd = (size_t)d * d, dp = d + (d & 1);
  const size_t prep = 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1 + dd + d + 2;
  const size_t epi = epilogue_lds_doubles(d) + 256;
  size_t w = (size_t)4 * orbit_lds_doubles(d, M, copies);
  if (prep > w) w = prep;
  if (epi > w) w = epi;
  w = (w + 1) & ~(size_t)1;
  return w + (size_t)4 * npairs(d);
}

template <int M, int SMAX, int WAVES, int EPLP>
__global__ __launch_bounds__(256, WAVES) void factor_fused_kernel(FusedArgs A) {
  extern __shared__ double sm[];
  if (pred_skip(A.tail.pred, A.tail.pred_val)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nfac = A.koff[A.nsets];
  if ((int)blockIdx.x >= nfac) {                              // chain-level trial mean (gather mode only)
    const int64_t j = (int64_t)((int)blockIdx.x - nfac) * 256 + threadIdx.x;
    if (j < A.nmu) A.mu_out[j] = A.gmu[j] + A.gstep * A.gdmu[j];
    return;
  }
  const int si = (A.nsets > 1 && (int)blockIdx.x >= A.koff[1]) ? 1 : 0;
  const FusedSet& S = A.s[si];
  const FactorDev& f = S.f;
  const int k = (int)blockIdx.x - A.koff[si];
  const int d = f.d, dd = d * d;
  // ---- phase 1 (wave 0): the factor's marginal and its per-pass products ----
  if (wave == 0) {
    const int lane = threadIdx.x;
    if (!A.gather) prep_body_d<EPLP>(f, S.mu, S.Sigma, k, sm, k);
    else {
      const int dp = d + (d & 1), n = A.n, nn = n * n;
      double* Sl = sm + 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1;   // behind prep_body's own LDS
      double* ml = Sl + dd;
      const int s = S.start[k];
      for (int e = lane; e < dd; e += 64) {
        const int r = e / d, c = e % d;
        double v;
        if (r < n && c < n) v = A.SigD[(size_t)s * nn + r * n + c];
        else if (r >= n && c >= n) v = A.SigD[(size_t)(s + 1) * nn + (r - n) * n + (c - n)];
        else if (r < n) v = A.SigU[(size_t)s * nn + r * n + (c - n)];
        else v = A.SigU[(size_t)s * nn + c * n + (r - n)];
        Sl[e] = v;
        S.Sigma_k[(size_t)k * dd + e] = v;
      }
      for (int e = lane; e < d; e += 64) {
        const size_t j = (size_t)s * n + e;
        const double v = A.gdmu ? A.gmu[j] + A.gstep * A.gdmu[j] : A.gmu[j];
        ml[e] = v;
        S.mu_k[(size_t)k * d + e] = v;
      }
      wave_lds_sync();
      prep_body_d<EPLP>(f, ml, Sl, k, sm, 0);
    }
  }
  __syncthreads();                                             // H / u0 of this factor are visible to the block
  // ---- phase 2 (every wave): its chunk of the orbit table ----
  const int NP = npairs(d);
  const int ldsw = orbit_lds_doubles(d, M, S.oa.copies);
  size_t walk = (size_t)4 * ldsw;
  {
    const size_t dp = d + (d & 1);
    const size_t prep = 4 * (size_t)dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1 + dd + d + 2, epi = epilogue_lds_doubles(d) + 256;
    if (prep > walk) walk = prep;
    if (epi > walk) walk = epi;
    walk = (walk + 1) & ~(size_t)1;
  }
  double* Pl = sm + walk;                                      // [4][NP] chunk partials
  orbit_wave<M, SMAX, true, false, true>(S.oa, k, wave, sm + (size_t)wave * ldsw, Pl + (size_t)wave * NP);
  __syncthreads();
  if (wave != 0) return;
  // ---- phase 3 (wave 0): ordered chunk sum, cost (+ tail), back-transform ----
  EpiArgs e;
  e.f = f; e.partial = nullptr; e.nchunk = 4; e.full = 1;
  e.Ephi = S.Ephi; e.cost = S.cost; e.Vdmu = S.Vdmu; e.Vddmu = S.Vddmu; e.E_xmuphi = nullptr; e.E_xxphi = nullptr;
  if (!A.tail.on) { epilogue_body_p(e, k, sm, 0, Pl); return; }
  const double costk = epilogue_body_p(e, k, sm, 1, Pl);
  __shared__ int last;
  epi_tail_arrive(A.cl, A.tail, S.cost + k, costk, (int)threadIdx.x, blockIdx.x, (unsigned)nfac, sm + epilogue_lds_doubles(d), &last);
  epilogue_body_p(e, k, sm, 2, Pl);
}

}  // namespace gvi
