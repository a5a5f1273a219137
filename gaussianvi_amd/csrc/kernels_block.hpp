// factor_block3_kernel: the factor stage of a full pass of the PLANNING GRAPH -- products, psi moments, chunk sum +
// back-transform + cost tail -- as ONE launch, for psi that is not a polynomial (the hinge on a signed-distance field,
// helpers/CudaOperation.cu:74-119 in the reference) and therefore has no sign-orbit form (kernels_fused.hpp is that case).
//
// The three launches it replaces (prep_all_kernel -> moments_planar3_kernel -> epilogue_all_kernel: 12.5 + 16.7 + 10.6 us
// at planar1k, profiles/r04_kernel_stats_planar1k.csv) are latency-bound around one 16.7 us kernel: a launch's floor on
// this part is ~3.5 us, the prep and the epilogue are one wave per factor with a dozen dependent loads.  Here a workgroup
// owns G = 4 / nchunk consecutive factors of a set (nchunk = chunks per factor of the set's moments launch: 1, 2 or 4; 3 is
// run as 4 with an idle wave); wave w has chunk w % nchunk of factor w / nchunk:
//   phase 1  the factor's FIRST wave: gather (mu_k, Sigma_k) out of the chain arrays, the products (Cholesky factor or
//            symmetric root, as prep_all_kernel forms them and to memory as it leaves them: later cost passes at the same
//            state reuse them);
//   phase 2  every wave: the set's own moments body on its (factor, chunk) (reg_body / sreg_body / sreg_pipe_body with kfix:
//            the same points per (factor, chunk) and the same sums as the set's own launch);
//   phase 3  the factor's first wave: ordered chunk sum, cost (+ the arrival protocol of the tail, epi_tail_arrive),
//            back-transform.
// Producer and consumer of every intermediate are waves of the SAME workgroup (one CU, one L1): the hand-overs are
// workgroup barriers.  Results are bit-identical to the three launches (tests: test_planning_graph_one_launch_*).
// (First version: one workgroup per factor, phases 1 and 3 on wave 0 -- a quarter of the resident waves active in the two
// latency-bound phases: 54.6 us against the three launches' 46.1.)
// (The tail protocol on a factor's idle second wave, as factor_fused_kernel runs it: every form tried made the compiler merge
// call sites over a pointer into the argument block and copy the block to scratch -- see block_products_of; not kept.)
#pragma once
#include "kernels_factor.hpp"

namespace gvi {

struct BlockSet {
  MomArgs a;                 // the set's moments launch as run_moments plans it (chunk, nchunk <= 4, partial, mu = mu_k)
  const int32_t* start;      // gather: first state of every factor
  double* mu_k;              // gather outputs (= the inputs of the products and of every later pass at this state)
  double* Sigma_k;
  double* Ephi;              // epilogue outputs
  double* cost;
  double* Vdmu;
  double* Vddmu;
};

struct Block3Args {
  BlockSet s[3];             // 0: minimum-acceleration priors d = 8, m = 4; 1: hinge on the SDF d = 4; 2: anchors d = 4
  int nitems;                // K0 + K1 + K2: arrivals of the tail (the workgroups behind the sets' write the chain-level trial mean)
  int pipe;                  // priors on the hand-pipelined body (as their own launch would be)
  // fused gather (see PrepList)
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

// the products of prep_body_d<1>: Cholesky factor for the sum-of-squares sets (CHOL), symmetric root for the others.
// CHOL is a template parameter, not a test of f.chol: with both bodies behind a run-time branch at two call sites the
// compiler merged the sites' code over a pointer into the kernel's argument block, copied the whole block (1.6 KB) to
// scratch memory for that, and the kernel went from 152 registers / no scratch to 191 / 1616 bytes per lane.
template <int DT, bool CHOL>
__device__ __forceinline__ void block_products_of(const FactorDev& f, const double* mu, const double* Sigma, const int k, double* sm, const int kin) {
  if constexpr (CHOL) prep_chol_body<DT>(f, mu, Sigma, k, sm, kin);
  else if constexpr (DT == 4) prep_body<1, 4>(f, mu, Sigma, k, sm, kin);
  else prep_body<1, 0>(f, mu, Sigma, k, sm, kin);
}

// phase 1 of one factor by the calling wave (DT = the set's d): prep_all_kernel's block, verbatim
template <int DT, bool CHOL>
__device__ __forceinline__ void block_products(const Block3Args& A, const BlockSet& S, const int k, double* sm) {
  const FactorDev& f = S.a.f;
  const int lane = threadIdx.x & 63;
  if (!A.gather) { block_products_of<DT, CHOL>(f, S.a.mu, S.Sigma_k, k, sm, k); return; }
  constexpr int d = DT, dd = d * d, dp = d + (d & 1);
  const int n = A.n, nn = n * n;
  double* Sl = sm + 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1;  // behind the products' own LDS
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
  block_products_of<DT, CHOL>(f, ml, Sl, k, sm, 0);
}

// phase 3 of one factor by the calling wave
// arrival: the factor's index among all factors of the launch (A.cl order)
template <int DT>
__device__ __forceinline__ void block_epilogue(const Block3Args& A, const BlockSet& S, const int k, const int arrival, double* sm, int* last) {
  EpiArgs e;
  e.f = S.a.f; e.partial = S.a.partial; e.nchunk = S.a.nchunk; e.full = 1;
  e.Ephi = S.Ephi; e.cost = S.cost; e.Vdmu = S.Vdmu; e.Vddmu = S.Vddmu; e.E_xmuphi = nullptr; e.E_xxphi = nullptr;
  const double* P = e.partial + (size_t)k * e.nchunk * npairs(DT);
  if (!A.tail.on) { epilogue_body_t<DT>(e, k, sm, 0, P); return; }
  // the factor's cost goes out first, the workgroup that arrives last sums and publishes, then the back-transform (as
  // epilogue_all_kernel)
  const double costk = epilogue_body_t<DT>(e, k, sm, 1, P);
  epi_tail_arrive(A.cl, A.tail, S.cost + k, costk, (int)(threadIdx.x & 63), (unsigned)arrival, (unsigned)A.nitems, last);
  epilogue_body_t<DT>(e, k, sm, 2, P);
}

// LDS of one wave's phases 1 and 3 (doubles), d <= 8; a workgroup has four of them
__host__ __device__ inline size_t block3_lds_doubles() {
  constexpr size_t d = 8, dd = 64, dp = 8;
  const size_t prep = 4 * dd + 2 * dp + 3 * d + (dp + 1) / 2 + 1 + dd + d + 2;
  const size_t epi = epilogue_lds_doubles((int)d) + 2;
  return ((prep > epi ? prep : epi) + 1) & ~(size_t)1;
}
// workgroups of a set: G = 4 / nch factors each (nch = 1, 2 or 4 waves per factor)
__host__ __device__ inline int block3_nch(int nchunk) { return nchunk == 3 ? 4 : nchunk; }
__host__ __device__ inline int block3_blocks(int K, int nchunk) { const int G = 4 / block3_nch(nchunk); return (K + G - 1) / G; }

__global__ __launch_bounds__(256, 2) void factor_block3_kernel(Block3Args A) {
  extern __shared__ double sm[];
  using PsiH = PsiHingeSdf<4, KIND_HINGE_SDF_2D>;
  using PsiA = PsiQuad<4, 4>;
  constexpr int HS = PsiH::LDS > PsiA::LDS ? (PsiH::LDS > 2 * 4 ? PsiH::LDS : 2 * 4) : (PsiA::LDS > 2 * 4 ? PsiA::LDS : 2 * 4);
  __shared__ double hs[4 * HS];
  __shared__ double red[4 * 16 * 65];
  __shared__ int last[4];
  if (pred_skip(A.tail.pred, A.tail.pred_val)) return;
  const int b = (int)blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K0 = A.s[0].a.f.K, K1 = A.s[1].a.f.K, K2 = A.s[2].a.f.K;
  const int nb0 = block3_blocks(K0, A.s[0].a.nchunk), nb1 = block3_blocks(K1, A.s[1].a.nchunk), nb2 = block3_blocks(K2, A.s[2].a.nchunk);
  if (b >= nb0 + nb1 + nb2) {                                  // chain-level trial mean (gather mode with a step)
    const int64_t j = (int64_t)(b - nb0 - nb1 - nb2) * 256 + threadIdx.x;
    if (j < A.nmu) A.mu_out[j] = A.gmu[j] + A.gstep * A.gdmu[j];
    return;
  }
  double* smw = sm + (size_t)wave * block3_lds_doubles();
  // the obstacle factors (the long ones) first, then the priors, then the anchors
  if (b < nb1) {
    const BlockSet& S = A.s[1];
    const int nch = block3_nch(S.a.nchunk), k = b * (4 / nch) + wave / nch, c = wave % nch;
    const bool lead = c == 0 && k < K1;
    if (lead) block_products<4, false>(A, S, k, smw);          // (symmetric root: psi is not a sum of squares)
    __syncthreads();
    reg_body<4, PsiH, true>(S.a, 0, 0, hs, red, k, c);
    __syncthreads();
    if (lead) block_epilogue<4>(A, S, k, K0 + k, smw, &last[wave]);
  } else if (b < nb1 + nb0) {
    const BlockSet& S = A.s[0];
    const int nch = block3_nch(S.a.nchunk), k = (b - nb1) * (4 / nch) + wave / nch, c = wave % nch;
    const bool lead = c == 0 && k < K0;
    if (lead) block_products<8, true>(A, S, k, smw);
    __syncthreads();
    if (A.pipe) sreg_pipe_dispatch<8, 4>(S.a, 0, 0, hs, red, k, c);
    else sreg_body<8, 4, true>(S.a, 0, 0, hs, red, k, c);
    __syncthreads();
    if (lead) block_epilogue<8>(A, S, k, k, smw, &last[wave]);
  } else {
    const BlockSet& S = A.s[2];
    const int nch = block3_nch(S.a.nchunk), k = (b - nb1 - nb0) * (4 / nch) + wave / nch, c = wave % nch;
    const bool lead = c == 0 && k < K2;
    if (lead) block_products<4, true>(A, S, k, smw);
    __syncthreads();
    reg_body<4, PsiA, true>(S.a, 0, 0, hs, red, k, c);
    __syncthreads();
    if (lead) block_epilogue<4>(A, S, k, K0 + K1 + k, smw, &last[wave]);
  }
}

}  // namespace gvi
