// Host side of kernels_chain.hpp: the pass plan of a chain and the launches of a factorisation and / or a solve.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "kernels_chain.hpp"
#include "kernels_chain_wave.hpp"

namespace gvi {

struct ChainPass { int level0, m, S, top, first, par, lp_off, blocks; };
struct ChainPlan { std::vector<ChainPass> passes; int threads = 0; };

// compile-time block size a chain of n x n blocks runs at (blocks padded by the identity), 0 = unsupported
inline int chain_padded(int n) {
  for (int N : {1, 2, 3, 4, 6, 8, 12, 16}) if (n <= N) return n >= 1 ? N : 0;
  return 0;
}
inline bool chain_supported(int n) { return chain_padded(n) != 0; }

inline int chain_levels(int T) {
  int L = 0;
  while ((1 << L) < T) ++L;
  return L;
}

inline ChainPlan chain_plan(int T, int n_actual) {
  const int n = chain_padded(n_actual);
  const int m_seg = n <= 6 ? 5 : (n <= 8 ? 4 : 3);
  // top-pass nodes: the forward arrays and the factors of all of them must fit LDS.  Small blocks leave room for a long top
  // pass, and a chain that fits it is done in ONE launch (the 65-state planar / configs[1] chains: launch floor, not arithmetic)
  const int cap = n <= 2 ? 128 : (n <= 4 ? 64 : (n <= 6 ? 48 : (n <= 8 ? 24 : 8)));
  ChainPlan p;
  p.threads = chain_threads(n);
  const int nwaves = p.threads / 64, nlevels = chain_levels(T);
  auto alive = [&](int l) { return (int)(((int64_t)T + (1 << l) - 1) >> l); };
  int level0 = 0, lp = 0, par = 0;
  while (alive(level0) > cap) {
    const int S = 1 << m_seg, stride = S << level0;
    const int blocks = (T + stride - 1) / stride;
    p.passes.push_back({level0, m_seg, S, 0, level0 == 0, par, lp, blocks});
    lp += blocks * nwaves;
    level0 += m_seg;
    par ^= 1;
  }
  p.passes.push_back({level0, nlevels - level0, alive(level0), 1, level0 == 0, par, lp, 1});
  return p;
}

// Raise the dynamic-LDS limit of the kernels of block size N once per device.
template <int N>
inline hipError_t chain_allow_lds() {
  static bool done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  const int lim = 160 * 1024;
  if ((e = hipFuncSetAttribute((const void*)chain_forward_kernel<N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)chain_forward_kernel<N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)chain_backward_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)chain_top_back_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if (dev >= 0 && dev < 64) done[dev] = true;
  return hipSuccess;
}

// The hand-over words of a merged top + backward launch (kernels_chain.hpp::chain_top_back_kernel): two device words (one per
// operation) and a sequence number the caller advances with every chain_launch.  words == nullptr: separate launches.
struct ChainSync { unsigned* words = nullptr; unsigned seq = 0; unsigned fault = 0; };   // fault: test hook (ChainArgs::sync_fault)

// on0: factorisation a0 (log-det; + selected inverse when a0.need_back); on1: pivoted solve a1.  Both: side by side in the
// same launches.  Returns hipErrorInvalidValue when a pass does not fit LDS.
template <int N>
inline hipError_t chain_launch_t(const ChainPlan& pl, ChainArgs a0, ChainArgs a1, bool on0, bool on1, hipStream_t st, const AsmList& AL,
                                 const ChainSync& sync) {
  hipError_t e = chain_allow_lds<N>();
  if (e != hipSuccess) return e;
  auto set = [](ChainArgs& a, const ChainPass& ps) {
    a.level0 = ps.level0; a.m = ps.m; a.S = ps.S; a.first = ps.first; a.par = ps.par; a.lp_off = ps.lp_off;
  };
  a0.sync = a1.sync = nullptr; a0.sync_seq = a1.sync_seq = 0; a0.sync_fault = a1.sync_fault = sync.fault;
  const bool back0 = on0 && a0.need_back;
  const int npass = (int)pl.passes.size();
  // the top pass carries the backward workgroups of the last segmented pass when there is one and its LDS fits
  bool merged = false;
  if (sync.words && npass >= 2 && (back0 || on1)) {
    const ChainPass& pt = pl.passes[npass - 1];
    const ChainPass& pc = pl.passes[npass - 2];
    size_t lds = 0;
    if (on0) lds = std::max(lds, chain::fwd_lds_doubles<true, false, true, N>(pt.S));
    if (on1) lds = std::max(lds, chain::fwd_lds_doubles<false, true, true, N>(pt.S));
    if (back0) lds = std::max(lds, chain::bwd_lds_doubles<true, N>(pc.S));
    if (on1) lds = std::max(lds, chain::bwd_lds_doubles<false, N>(pc.S));
    merged = lds * 8 <= 160 * 1024;
  }
  for (const ChainPass& ps : pl.passes) {
    set(a0, ps); set(a1, ps);
    if (ps.top && merged) {
      const ChainPass& pc = pl.passes[npass - 2];
      size_t lds = 0;
      if (on0) lds = std::max(lds, chain::fwd_lds_doubles<true, false, true, N>(ps.S));
      if (on1) lds = std::max(lds, chain::fwd_lds_doubles<false, true, true, N>(ps.S));
      if (back0) lds = std::max(lds, chain::bwd_lds_doubles<true, N>(pc.S));
      if (on1) lds = std::max(lds, chain::bwd_lds_doubles<false, N>(pc.S));
      lds *= 8;
      a0.sync = sync.words; a1.sync = sync.words + 1; a0.sync_seq = a1.sync_seq = sync.seq;
      const int nb0 = on0 ? 1 : 0, nbt = nb0 + (on1 ? 1 : 0);
      const int nb0c = back0 ? pc.blocks : 0, nbc = nb0c + (on1 ? pc.blocks : 0);
      const ChainPassDev cp{pc.level0, pc.m, pc.S, pc.first, pc.par, pc.lp_off};
      hipLaunchKernelGGL((chain_top_back_kernel<N>), dim3(nbt + nbc), dim3(pl.threads), lds, st, a0, a1, nb0, AL, cp, nbt, nb0c);
      a0.sync = a1.sync = nullptr;
      continue;
    }
    size_t lds = 0;
    if (ps.top) {
      if (on0) lds = std::max(lds, chain::fwd_lds_doubles<true, false, true, N>(ps.S));
      if (on1) lds = std::max(lds, chain::fwd_lds_doubles<false, true, true, N>(ps.S));
    } else {
      if (on0) lds = std::max(lds, chain::fwd_lds_doubles<true, false, false, N>(ps.S));
      if (on1) lds = std::max(lds, chain::fwd_lds_doubles<false, true, false, N>(ps.S));
    }
    lds *= 8;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int nb0 = on0 ? ps.blocks : 0, nb = nb0 + (on1 ? ps.blocks : 0);
    if (ps.top) hipLaunchKernelGGL((chain_forward_kernel<N, true>), dim3(nb), dim3(pl.threads), lds, st, a0, a1, nb0, AL);
    else hipLaunchKernelGGL((chain_forward_kernel<N, false>), dim3(nb), dim3(pl.threads), lds, st, a0, a1, nb0, AL);
  }
  if (back0 || on1) {
    for (int i = npass - 2 - (merged ? 1 : 0); i >= 0; --i) {
      const ChainPass& ps = pl.passes[i];
      set(a0, ps); set(a1, ps);
      size_t lds = 0;
      if (back0) lds = std::max(lds, chain::bwd_lds_doubles<true, N>(ps.S));
      if (on1) lds = std::max(lds, chain::bwd_lds_doubles<false, N>(ps.S));
      lds *= 8;
      if (lds > 160 * 1024) return hipErrorInvalidValue;
      const int nb0 = back0 ? ps.blocks : 0, nb = nb0 + (on1 ? ps.blocks : 0);
      hipLaunchKernelGGL((chain_backward_kernel<N>), dim3(nb), dim3(pl.threads), lds, st, a0, a1, nb0);
    }
  }
  return hipGetLastError();
}

// Short chains of 2 x 2 blocks (T <= 65, n <= 2) run lane-per-node in ONE wave per operation (kernels_chain_wave.hpp);
// chain_wave_enabled() = false keeps them on the generic kernels (A/B leg, GVI_CHAIN_WAVE=0)
inline bool& chain_wave_enabled() { static bool on = true; return on; }
inline bool chain_wave_applies(int T, int n) { return chain_wave_enabled() && n >= 1 && n <= chain_wave::WN && T >= 1 && T <= chain_wave::WT_MAX; }

// n: the caller's block size (a0.n / a1.n are set here).  AL: the factor sets of an assemble-on-load (a0.asm_on / a1.asm_on), else null
inline hipError_t chain_launch(int n, const ChainPlan& pl, ChainArgs a0, ChainArgs a1, bool on0, bool on1, hipStream_t st,
                               const AsmList* AL = nullptr, const ChainSync& sync = ChainSync{}) {
  a0.n = a1.n = n;
  AsmList none{};
  const AsmList& L = AL ? *AL : none;
  if (!AL) a0.asm_on = a1.asm_on = 0;
  if (chain_wave_applies(on0 ? a0.T : a1.T, n)) {
    const int nb0 = on0 ? 1 : 0, nb = nb0 + (on1 ? 1 : 0);
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(chain_wave_kernel, dim3(nb), dim3(64), 0, st, a0, a1, nb0, L);
    return hipGetLastError();
  }
  switch (chain_padded(n)) {
    case 1: return chain_launch_t<1>(pl, a0, a1, on0, on1, st, L, sync);
    case 2: return chain_launch_t<2>(pl, a0, a1, on0, on1, st, L, sync);
    case 3: return chain_launch_t<3>(pl, a0, a1, on0, on1, st, L, sync);
    case 4: return chain_launch_t<4>(pl, a0, a1, on0, on1, st, L, sync);
    case 6: return chain_launch_t<6>(pl, a0, a1, on0, on1, st, L, sync);
    case 8: return chain_launch_t<8>(pl, a0, a1, on0, on1, st, L, sync);
    case 12: return chain_launch_t<12>(pl, a0, a1, on0, on1, st, L, sync);
    case 16: return chain_launch_t<16>(pl, a0, a1, on0, on1, st, L, sync);
  }
  return hipErrorInvalidValue;
}

}  // namespace gvi
