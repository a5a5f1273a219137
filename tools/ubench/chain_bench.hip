// Stand-alone check + timing of the chain kernels (kernels_chain.hpp) against a sequential host solver and against the
// round-2 segmented kernels (kernels_bcr_seg.hpp), both chain operations of an NGD iteration side by side:
//   factorisation of Lam (1/2 log det + tridiagonal blocks of the inverse)  ||  pivoted solve V x = -g
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -I gaussianvi_amd/csrc -I tools/ab tools/ubench/chain_bench.hip -o tools/ubench/chain_bench
// Run:    tools/ubench/chain_bench [T=1025] [n=6] [reps=200]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "chain_launch.hpp"
#include "kernels_bcr_seg.hpp"

using namespace gvi;

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

// ---- host reference: sequential block LDL^T, solve, selected inverse ----
struct Mat { int n; std::vector<double> a; Mat(int n_ = 0) : n(n_), a((size_t)n_ * n_, 0.0) {} double& operator()(int r, int c) { return a[(size_t)r * n + c]; } double operator()(int r, int c) const { return a[(size_t)r * n + c]; } };
static Mat mul(const Mat& A, const Mat& B, bool tA = false, bool tB = false) {
  const int n = A.n; Mat C(n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { long double s = 0; for (int k = 0; k < n; ++k) s += (long double)(tA ? A(k, i) : A(i, k)) * (tB ? B(j, k) : B(k, j)); C(i, j) = (double)s; }
  return C;
}
static Mat inv(const Mat& A, double* logdet) {
  const int n = A.n; std::vector<long double> M((size_t)n * 2 * n, 0);
  for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) M[(size_t)i * 2 * n + j] = A(i, j); M[(size_t)i * 2 * n + n + i] = 1; }
  long double ld = 0;
  for (int p = 0; p < n; ++p) {
    int best = p; for (int r = p + 1; r < n; ++r) if (fabsl(M[(size_t)r * 2 * n + p]) > fabsl(M[(size_t)best * 2 * n + p])) best = r;
    if (best != p) for (int j = 0; j < 2 * n; ++j) std::swap(M[(size_t)p * 2 * n + j], M[(size_t)best * 2 * n + j]);
    const long double piv = M[(size_t)p * 2 * n + p]; ld += logl(fabsl(piv));
    for (int j = 0; j < 2 * n; ++j) M[(size_t)p * 2 * n + j] /= piv;
    for (int r = 0; r < n; ++r) if (r != p) { const long double f = M[(size_t)r * 2 * n + p]; for (int j = 0; j < 2 * n; ++j) M[(size_t)r * 2 * n + j] -= f * M[(size_t)p * 2 * n + j]; }
  }
  Mat R(n); for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) R(i, j) = (double)M[(size_t)i * 2 * n + n + j];
  if (logdet) *logdet = (double)ld;
  return R;
}
static Mat blk(const std::vector<double>& v, int t, int n) { Mat M(n); for (int i = 0; i < n * n; ++i) M.a[i] = v[(size_t)t * n * n + i]; return M; }

static void host_chain(int T, int n, const std::vector<double>& D, const std::vector<double>& U, const std::vector<double>& rhs, double scale,
                       std::vector<double>& SigD, std::vector<double>& SigU, std::vector<double>& x, double& hld) {
  const int nn = n * n;
  std::vector<Mat> Sinv(T, Mat(n));
  std::vector<std::vector<double>> y(T, std::vector<double>(n));
  double ld = 0, l;
  Mat S = blk(D, 0, n);
  for (int i = 0; i < n; ++i) y[0][i] = scale * rhs[i];
  for (int t = 0; t < T; ++t) {
    Sinv[t] = inv(S, &l); ld += l;
    if (t + 1 < T) {
      const Mat Ut = blk(U, t, n);
      const Mat W = mul(Sinv[t], Ut);            // S_t^-1 U_t
      const Mat C = mul(Ut, W, true, false);     // U_t^T S_t^-1 U_t
      S = blk(D, t + 1, n);
      for (int i = 0; i < nn; ++i) S.a[i] -= C.a[i];
      for (int i = 0; i < n; ++i) { long double s = scale * rhs[(size_t)(t + 1) * n + i]; for (int k = 0; k < n; ++k) { long double w = 0; for (int q = 0; q < n; ++q) w += (long double)Sinv[t](k, q) * y[t][q]; s -= (long double)Ut(k, i) * w; } y[t + 1][i] = (double)s; }
    }
  }
  hld = 0.5 * ld;
  x.assign((size_t)T * n, 0.0); SigD.assign((size_t)T * nn, 0.0); SigU.assign((size_t)std::max(0, T - 1) * nn, 0.0);
  for (int t = T - 1; t >= 0; --t) {
    std::vector<long double> r(n);
    for (int i = 0; i < n; ++i) r[i] = y[t][i];
    if (t + 1 < T) { const Mat Ut = blk(U, t, n); for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) r[i] -= (long double)Ut(i, k) * x[(size_t)(t + 1) * n + k]; }
    for (int i = 0; i < n; ++i) { long double s = 0; for (int k = 0; k < n; ++k) s += (long double)Sinv[t](i, k) * r[k]; x[(size_t)t * n + i] = (double)s; }
    if (t == T - 1) { for (int i = 0; i < nn; ++i) SigD[(size_t)t * nn + i] = Sinv[t].a[i]; }
    else {
      const Mat Ut = blk(U, t, n), Sn = blk(SigD, t + 1, n);
      const Mat W = mul(Sinv[t], Ut);                       // S^-1 U
      const Mat WS = mul(W, Sn);                            // S^-1 U Sig_{t+1}
      for (int i = 0; i < nn; ++i) SigU[(size_t)t * nn + i] = -WS.a[i];
      const Mat Q = mul(WS, W, false, true);                // S^-1 U Sig U^T S^-1
      for (int i = 0; i < nn; ++i) SigD[(size_t)t * nn + i] = Sinv[t].a[i] + Q.a[i];
    }
  }
}

static double relerr(const std::vector<double>& a, const std::vector<double>& b) {
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { num = std::max(num, fabs(a[i] - b[i])); den = std::max(den, fabs(b[i])); }
  return num / (den > 0 ? den : 1);
}

// ---- round-2 kernels, launched as gvi_hip.hip did ----
struct OldPass { int level0, m, S, prev0, top; };
static std::vector<OldPass> old_plan(int T, int n, int* threads) {
  const int m_seg = n <= 6 ? 5 : (n <= 8 ? 4 : 3);
  const int cap = n <= 2 ? 128 : (n <= 4 ? 64 : (n <= 6 ? 48 : (n <= 8 ? 24 : 8)));
  std::vector<OldPass> p;
  int level0 = 0, prev0 = 0;
  auto alive = [&](int l) { return (int)(((int64_t)T + (1 << l) - 1) >> l); };
  while (alive(level0) > cap) { p.push_back({level0, m_seg, 1 << m_seg, prev0, 0}); prev0 = level0; level0 += m_seg; }
  p.push_back({level0, chain_levels(T) - level0, alive(level0), prev0, 1});
  *threads = n <= 8 ? 1024 : 512;
  return p;
}
// CHAIN_MERGE=1: the top pass and the first backward pass in one launch (chain_launch.hpp::ChainSync)
static unsigned* g_sync_words = nullptr;
static unsigned g_sync_seq = 0;
static gvi::ChainSync next_sync() {
  gvi::ChainSync sy;
  const char* e = getenv("CHAIN_MERGE");
  if (!e || atoi(e) == 0) return sy;
  if (!g_sync_words) { (void)hipMalloc(&g_sync_words, 64 * 2 * sizeof(unsigned)); (void)hipMemset(g_sync_words, 0, 64 * 2 * sizeof(unsigned)); }
  if (++g_sync_seq == 0) ++g_sync_seq;
  sy.seq = g_sync_seq;
  sy.words = g_sync_words + 2 * (sy.seq % 64);
  return sy;
}

template <int N>
static void old_launch(int T, SegArgs a0, SegArgs a1, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)bcr_seg_forward_dual_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)bcr_seg_backward_dual_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int threads;
  const auto pl = old_plan(T, N, &threads);
  for (const auto& ps : pl) {
    for (SegArgs* a : {&a0, &a1}) { a->level0 = ps.level0; a->m = ps.m; a->S = ps.S; a->prev0 = ps.prev0; a->top = ps.top; }
    const size_t lds = std::max(seg_fwd_lds_doubles(N, ps.S, false, ps.top != 0, threads / 64), seg_fwd_lds_doubles(N, ps.S, true, ps.top != 0, threads / 64)) * 8;
    const int stride = ps.S << ps.level0;
    const int blocks = ps.top ? 1 : (T + stride - 1) / stride;
    hipLaunchKernelGGL((bcr_seg_forward_dual_kernel<N>), dim3(2 * blocks), dim3(threads), lds, st, a0, a1, blocks);
  }
  for (int i = (int)pl.size() - 2; i >= 0; --i) {
    const auto& ps = pl[i];
    for (SegArgs* a : {&a0, &a1}) { a->level0 = ps.level0; a->m = ps.m; a->S = ps.S; a->prev0 = ps.prev0; a->top = 0; }
    const size_t lds = std::max(seg_bwd_lds_doubles(N, ps.S, false), seg_bwd_lds_doubles(N, ps.S, true)) * 8;
    const int stride = ps.S << ps.level0;
    const int blocks = (T + stride - 1) / stride;
    hipLaunchKernelGGL((bcr_seg_backward_dual_kernel<N>), dim3(2 * blocks), dim3(threads), lds, st, a0, a1, blocks);
  }
  CK(hipGetLastError());
}

template <int N>                     // N = n where the round-2 kernels are instantiated (A/B leg), else 0
static int run(int T, int n, int reps, bool with_old) {
  const int nn = n * n, NP = chain_padded(n);
  if (N == 0) with_old = false;
  std::mt19937_64 rng(1234 + T + n);
  std::normal_distribution<double> nd(0.0, 1.0);
  // two chains: Lam (SPD, factorised) and V (SPD here too, solved with pivoting), diagonally dominant blocks
  auto make = [&](std::vector<double>& D, std::vector<double>& U) {
    D.assign((size_t)T * nn, 0.0); U.assign((size_t)std::max(0, T - 1) * nn, 0.0);
    for (int t = 0; t < T; ++t) {
      std::vector<double> B(nn);
      for (auto& v : B) v = nd(rng);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k]; D[(size_t)t * nn + i * n + j] = s / n + (i == j ? 3.0 : 0.0); }
      if (t + 1 < T) for (int i = 0; i < nn; ++i) U[(size_t)t * nn + i] = 0.6 * nd(rng) / std::sqrt((double)n);
    }
  };
  std::vector<double> D0, U0, D1, U1, rhs((size_t)T * n);
  make(D0, U0); make(D1, U1);
  for (auto& v : rhs) v = nd(rng);
  std::vector<double> rSigD, rSigU, rx, dummyD, dummyU, dx;
  double rhld, dh;
  host_chain(T, n, D0, U0, rhs, 1.0, rSigD, rSigU, dx, rhld);
  host_chain(T, n, D1, U1, rhs, -1.0, dummyD, dummyU, rx, dh);

  const size_t btD = (size_t)T * nn, btU = (size_t)std::max(0, T - 1) * nn;
  double *dD0, *dD1, *drhs, *dSig, *dx_, *dhld, *ws0, *ws1, *ows0, *ows1;
  int *wi0, *wi1, *obad0, *obad1;
  CK(hipMalloc(&dD0, (btD + btU + 1) * 8)); CK(hipMalloc(&dD1, (btD + btU + 1) * 8)); CK(hipMalloc(&drhs, (size_t)T * n * 8));
  CK(hipMalloc(&dSig, (btD + btU + 1) * 8)); CK(hipMalloc(&dx_, (size_t)T * n * 8)); CK(hipMalloc(&dhld, 8));
  CK(hipMalloc(&ws0, chain_ws_doubles(T, NP) * 8)); CK(hipMalloc(&ws1, chain_ws_doubles(T, NP) * 8));
  CK(hipMalloc(&wi0, chain_lp_entries(T) * 4)); CK(hipMalloc(&wi1, chain_lp_entries(T) * 4));
  const size_t oldw = (size_t)9 * T * nn + (size_t)4 * T * n + T;
  CK(hipMalloc(&ows0, oldw * 8)); CK(hipMalloc(&ows1, oldw * 8)); CK(hipMalloc(&obad0, (size_t)T * 4)); CK(hipMalloc(&obad1, (size_t)T * 4));
  CK(hipMemcpy(dD0, D0.data(), btD * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dD0 + btD, U0.data(), btU * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dD1, D1.data(), btD * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dD1 + btD, U1.data(), btU * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(drhs, rhs.data(), (size_t)T * n * 8, hipMemcpyHostToDevice));
  // poison the workspaces: nothing may depend on their initial contents
  CK(hipMemset(ws0, 0xff, chain_ws_doubles(T, NP) * 8)); CK(hipMemset(ws1, 0xff, chain_ws_doubles(T, NP) * 8));
  hipStream_t st;
  CK(hipStreamCreate(&st));

  ChainArgs a0{}, a1{};
  a0.T = T; a0.need_back = 1; a0.D = dD0; a0.U = dD0 + btD; a0.rhs = nullptr; a0.rhs_scale = 1.0; a0.ws = ws0; a0.wsi = wi0;
  a0.SigD = dSig; a0.SigU = dSig + btD; a0.x = nullptr; a0.hld = dhld; a0.mixV = nullptr; a0.mixOut = nullptr; a0.mix_step = 0; a0.pred = nullptr; a0.pred_val = 0;
  a1 = a0;
  a1.need_back = 0; a1.D = dD1; a1.U = dD1 + btD; a1.rhs = drhs; a1.rhs_scale = -1.0; a1.ws = ws1; a1.wsi = wi1; a1.SigD = nullptr; a1.SigU = nullptr; a1.x = dx_; a1.hld = nullptr;
  const ChainPlan pl = chain_plan(T, n);
  printf("T = %d, n = %d (kernels at N = %d): %zu forward pass(es), %d threads\n", T, n, NP, pl.passes.size(), pl.threads);

  auto check = [&](const char* name) {
    std::vector<double> SigD(btD), SigU(btU), x((size_t)T * n);
    double hld;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(SigD.data(), dSig, btD * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(SigU.data(), dSig + btD, btU * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x.data(), dx_, (size_t)T * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hld, dhld, 8, hipMemcpyDeviceToHost));
    const double e1 = relerr(SigD, rSigD), e2 = btU ? relerr(SigU, rSigU) : 0.0, e3 = relerr(x, rx), e4 = fabs(hld - rhld) / fabs(rhld);
    printf("%-6s errors vs host: SigD %.2e  SigU %.2e  x %.2e  half-logdet %.2e (%.12g)\n", name, e1, e2, e3, e4, hld);
    return (e1 < 1e-10 && e2 < 1e-10 && e3 < 1e-10 && e4 < 1e-12) ? 0 : 1;
  };
  auto clear_out = [&]() { CK(hipMemsetAsync(dSig, 0xff, (btD + btU) * 8, st)); CK(hipMemsetAsync(dx_, 0xff, (size_t)T * n * 8, st)); CK(hipMemsetAsync(dhld, 0xff, 8, st)); };

  int fail = 0;
  clear_out();
  CK(chain_launch(n, pl, a0, a1, true, true, st, nullptr, next_sync()));
  fail |= check("new");
  // single operations through the same kernels
  clear_out();
  CK(chain_launch(n, pl, a0, a1, true, false, st, nullptr, next_sync()));
  CK(chain_launch(n, pl, a0, a1, false, true, st, nullptr, next_sync()));
  fail |= check("new/1");

  SegArgs o0{}, o1{};
  if (with_old) {
    o0.T = T; o0.n = n; o0.need_E = 1; o0.D = dD0; o0.U = dD0 + btD; o0.rhs = nullptr; o0.rhs_scale = 1.0; o0.w.base = ows0; o0.w.bad = obad0;
    o0.SigD = dSig; o0.SigU = dSig + btD; o0.x = nullptr; o0.hld = dhld; o0.mixVD = nullptr; o0.mixOutD = nullptr; o0.mix_step = 0; o0.pred = nullptr; o0.pred_val = 0;
    o1 = o0;
    o1.need_E = 0; o1.D = dD1; o1.U = dD1 + btD; o1.rhs = drhs; o1.rhs_scale = -1.0; o1.w.base = ows1; o1.w.bad = obad1; o1.SigD = nullptr; o1.SigU = nullptr; o1.x = dx_; o1.hld = nullptr;
    clear_out();
    if constexpr (N > 0) old_launch<N>(T, o0, o1, st);
    fail |= check("old");
  }
  // run-to-run bit identity of the new kernels
  {
    std::vector<double> A(btD + btU), B(btD + btU);
    CK(chain_launch(n, pl, a0, a1, true, true, st, nullptr, next_sync())); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(A.data(), dSig, (btD + btU) * 8, hipMemcpyDeviceToHost));
    int diff = 0;
    for (int it = 0; it < 5; ++it) {
      CK(chain_launch(n, pl, a0, a1, true, true, st, nullptr, next_sync())); CK(hipStreamSynchronize(st));
      CK(hipMemcpy(B.data(), dSig, (btD + btU) * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < A.size(); ++i) diff += A[i] != B[i];
    }
    printf("run-to-run differing words: %d\n", diff);
    fail |= diff != 0;
  }
#ifdef GVI_CHAIN_TIMING
  {
    // shader-clock stamps of thread 0 of the top pass (factorisation): label every delta by hand from the stamp order in
    // kernels_chain.hpp (pass start, load, barrier, then per elimination of wave 0: [col | GJ | logp | factors | products+adds],
    // barrier after each level, root, ..., backward: per node [phase 1 | sync | phase 2], barrier)
    int zero = 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(gvi_chain_nstamp), &zero, sizeof(int)));
    CK(chain_launch(n, pl, a0, a1, true, false, st, nullptr, next_sync()));
    CK(hipStreamSynchronize(st));
    int ns = 0;
    std::vector<unsigned long long> stp(256);
    CK(hipMemcpyFromSymbol(&ns, HIP_SYMBOL(gvi_chain_nstamp), sizeof(int)));
    CK(hipMemcpyFromSymbol(stp.data(), HIP_SYMBOL(gvi_chain_stamps), 256 * sizeof(unsigned long long)));
    printf("stamps (%d), cycles between:", ns);
    for (int i = 1; i < ns; ++i) printf(" %llu", stp[i] - stp[i - 1]);
    printf("\n");
  }
#endif
  // timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto&& fn) {
    for (int i = 0; i < 20; ++i) fn();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us per call\n", name, 1e3 * ms / reps);
  };
  time_it("new: factor || solve", [&]() { CK(chain_launch(n, pl, a0, a1, true, true, st, nullptr, next_sync())); });
  time_it("new: factor only", [&]() { CK(chain_launch(n, pl, a0, a1, true, false, st, nullptr, next_sync())); });
  time_it("new: solve only", [&]() { CK(chain_launch(n, pl, a0, a1, false, true, st, nullptr, next_sync())); });
  if constexpr (N > 0) { if (with_old) time_it("old: factor || solve", [&]() { old_launch<N>(T, o0, o1, st); }); }
  {
    // the same launches with a predicate that does not hold: every block returns at its first instruction -- what the launch
    // configuration itself costs (dispatch of 2 x 33 workgroups of 16 waves with ~100 KB of LDS each, kernel arguments, drain)
    double* dpred = nullptr;
    CK(hipMalloc(&dpred, 8));
    CK(hipMemset(dpred, 0, 8));
    ChainArgs s0 = a0, s1 = a1;
    s0.pred = s1.pred = dpred; s0.pred_val = s1.pred_val = 1.0;
    time_it("skipped (predicate): both", [&]() { CK(chain_launch(n, pl, s0, s1, true, true, st, nullptr, next_sync())); });
    CK(hipFree(dpred));
  }
  return fail;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1025, n = argc > 2 ? atoi(argv[2]) : 6, reps = argc > 3 ? atoi(argv[3]) : 200;
  const bool with_old = argc > 4 ? atoi(argv[4]) != 0 : true;
  int rc = 1;
  switch (n) {
    case 1: rc = run<1>(T, n, reps, with_old); break;
    case 2: rc = run<2>(T, n, reps, with_old); break;
    case 3: rc = run<3>(T, n, reps, with_old); break;
    case 4: rc = run<4>(T, n, reps, with_old); break;
    case 6: rc = run<6>(T, n, reps, with_old); break;
    case 8: rc = run<8>(T, n, reps, with_old); break;
    case 12: rc = run<12>(T, n, reps, with_old); break;
    default:
      if (!chain_supported(n)) { fprintf(stderr, "n must be in 1..16\n"); return 2; }
      rc = run<0>(T, n, reps, false);
  }
  printf(rc ? "FAILED\n" : "OK\n");
  return rc;
}
