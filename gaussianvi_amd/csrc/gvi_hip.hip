// C-ABI implementation (include/gvi_hip.h): context, factor sets, kernel dispatch, resident NGD state.
#include "../../include/gvi_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "host_linalg.hpp"
#include "kernels_bt.hpp"
#include "chain_launch.hpp"
#include "kernels_factor.hpp"
#include "kernels_orbit.hpp"
#include "kernels_fused.hpp"
#include "kernels_block.hpp"
#include "kernels_orbit_psi.hpp"
#include "orbits.hpp"
#include "spgh.hpp"

using namespace gvi;

namespace {

struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  ~DevMem() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  double* d() const { return (double*)p; }
  int32_t* i() const { return (int32_t*)p; }
};

struct Table {
  int d = 0, p = 0;
  int64_t N = 0, Np = 0;
  DevMem Zt, w;
  DevMem Zq;               // tile-major copy [Np / 64][d + 1][64] (row d = w) read by the hand-pipelined kernels
  DevMem Zm;               // mirror-half tile-major table (one representative per +-pair), empty if the table is not symmetric
  int64_t Nm = 0, Nmp = 0;
  DevMem codes, lut;       // 8-bit node codes [d/4][Np] + value look-up (moments_split_kernel); empty when not coded
  bool coded = false;
  // sign-orbit form (orbits.hpp; moments_orbit_kernel): host copy keeps only the tile lists, the rest lives on the device
  OrbitHost orb;
  DevMem orb_cpk, orb_rpk, orb_mag, orb_w;
  std::map<int, std::unique_ptr<DevMem>> orb_bounds;   // nchunk -> [nchunk + 1] tile bounds
};

struct FactorSet {
  int K = 0, d = 0, p = 0, m = 0, kind = 0, raw_stride = 0;
  std::vector<int32_t> start;
  std::shared_ptr<Table> table;
  DevMem dstart, dptr, didx, A, b, sgn, raw, temperature;
  DevMem S, Sinv, Lam, H, Hq, u0;     // per-pass products
  DevMem ones;                        // [K] unit temperatures (proximal rule: the reference's prox classes never divide by T)
  bool unit_temperature = false;
  DevMem jko_half, jko_S, jko_Sinv, jko_Lam;   // scratch of the JKO map
  DevMem sdf;                         // HINGE_SDF_* grid
  DevMem arm;                         // HINGE_SDF_3D_ARM: DH chain + collision spheres
  int sdf_rows = 0, sdf_cols = 0, sdf_nz = 1;
  double sdf_ox = 0, sdf_oy = 0, sdf_oz = 0, sdf_cell = 1;
  DevMem Vws;                         // eigenvectors of the last resident-NGD prep (Jacobi warm start)
  int warm_count = 0;                 // preps since the last cold start
  DevMem partial;
  int nchunk = 1;
  int64_t chunk = 0;
  bool use_reg = false;
  bool use_split = false;
  bool use_orbit = false;
  bool use_opsi = false;              // sign-orbit kernel for the non-polynomial psi kinds (kernels_orbit_psi.hpp)
  int arm_ndof = 0;
  bool fused_pair = false;            // last resident launch went out fused with the other set
  bool closed_form = false;           // NGDFactorizedLinear route (no sigma points)
  bool chain_structured = false;      // start[k] == k (factor k on state k / states k, k + 1): assemble-on-load needs no CSR
  bool all_pos = false;               // every residual row has sgn = +1 (positive-definite weight)
  double jtol = 1e-34;                // ctx->jacobi_tol
  bool use_chol = true;               // ctx->chol_sqrt
  bool force_sym = false;             // set around passes whose caller sees the sigma points
  bool use_mirror = true;             // evaluate +-pairs from the mirror-half table where the kernel supports it (ctx->mirror)
  int prep_slot = -1;                 // NGD slot whose (mu_k, Sigma_k) the per-pass products belong to
  hipStream_t st = nullptr;           // the set's own stream: prep -> moments -> epilogue overlap across sets
  hipEvent_t done = nullptr;
  // operator outputs / NGD per-set state
  DevMem mu_k[2], Sigma_k[2], Ephi, cost, Vdmu, Vddmu, raw1, raw2, X, psi_ext;
  DevMem in_mu, in_Sigma;             // staging of the host-pointer API (never aliases NGD state)
  hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [moments|cost][start|stop]
  bool ev_set[2] = {false, false};
  ~FactorSet() {
    for (auto& a : ev) for (auto& e : a) if (e) (void)hipEventDestroy(e);
    if (done) (void)hipEventDestroy(done);
    if (st) (void)hipStreamDestroy(st);
  }
  FactorDev dev() const {
    FactorDev f;
    f.K = K; f.d = d; f.m = m; f.kind = kind;
    f.N = table->N; f.Np = table->Np;
    f.Zt = table->Zt.d(); f.w = table->w.d();
    f.Zq = table->Zq.p ? table->Zq.d() : nullptr; f.all_pos = all_pos ? 1 : 0;
    f.Zm = (table->Zm.p && use_mirror) ? table->Zm.d() : nullptr; f.Nm = table->Nm; f.Nmp = table->Nmp;
    f.codes = table->coded ? (const uint32_t*)table->codes.p : nullptr; f.lut = table->coded ? table->lut.d() : nullptr;
    f.A = A.d(); f.b = b.d(); f.sgn = sgn.d(); f.raw = raw.d(); f.raw_stride = raw_stride;
    f.temperature = unit_temperature ? ones.d() : temperature.d();
    f.S = S.d(); f.Sinv = Sinv.d(); f.Lam = Lam.d(); f.H = H.d(); f.Hq = Hq.p ? Hq.d() : nullptr; f.u0 = u0.d();
    f.Vws = nullptr; f.warm = 0; f.jko_h = 0.0; f.jtol = jtol;
    // Cholesky factor instead of the symmetric root: sum-of-squares psi on a generated table of degree >= 3 (exact
    // quadrature -- kernels_factor.hpp, prep_chol_body), the instantiated dimensions, and not when the caller is going
    // to look at the sigma points themselves (force_sym: gvi_expand / gvi_moments_from_psi) or runs the JKO map
    f.chol = (use_chol && !force_sym && (kind == KIND_QUAD_PRIOR || kind == KIND_FIXED_PRIOR) && table->p >= 3 && d <= 32) ? 1 : 0;
    f.sdf = sdf.d(); f.sdf_rows = sdf_rows; f.sdf_cols = sdf_cols; f.sdf_ox = sdf_ox; f.sdf_oy = sdf_oy; f.sdf_cell = sdf_cell; f.sdf_inv_cell = 1.0 / sdf_cell; f.sdf_nz = sdf_nz; f.sdf_oz = sdf_oz; f.arm = arm.p ? arm.d() : nullptr;
    return f;
  }
};

struct NgdState {
  bool ready = false;
  int cur = 0;
  DevMem mu[2], Lam[2], Sig[2], hld[2];   // Lam = [D | U], Sig = [SigD | SigU]
  DevMem exch0[2], exch1;                 // gradient buffers [g | VD | VU] (double-buffered for speculation), [cost partial sum]
  DevMem dmu2[2];
  int gcur = 0;                           // gradient buffer holding the gradients of `grad_slot`
  bool grad_valid = false;
  int grad_slot = -1;
  DevMem dmu, dLam, total;
  bool cost_valid[2] = {false, false};
  double cost[2] = {0, 0};
  bool have_trial = false;
  bool spec_ready = false;            // gradients at the TRIAL state sit in exch0[1 - gcur] / dmu2[1 - gcur]
  // gather of slot i deferred into the next prep launch of that slot (ngd_prep_all) -- see PrepList
  struct GatherPending { bool on = false; const double* mu_from = nullptr; const double* dmu = nullptr; double step = 0.0; };
  GatherPending gpend[2];
};

}  // namespace

struct gvi_ctx {
  int device = 0, dtype = GVI_F64;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  int T = 0, n = 0;
  std::vector<std::unique_ptr<FactorSet>> sets;
  std::vector<std::shared_ptr<Table>> tables;
  NgdState ngd;
  DevMem Wbuf, Ibuf, vbuf, scratch, hldtmp;
  std::string err;
  int variant = 0;
  bool warm_start = true;             // resident NGD: Jacobi warm start from the previous eigenvectors
  // gvi_ngd_step: trial cost from the full moments pass at the trial point (one psi pass per accepted iteration).
  // 0 never, 1 always (first trial), 2 adaptive (default): fused while first trials keep being accepted, the m0-only
  // cost pass after a rejection (a rejected fused trial wastes the moment accumulation)
  int fuse_trial = 2;
  bool last_first_accepted = true;
  int64_t n_full_pass = 0, n_cost_pass = 0;   // psi passes launched over all sets (gvi_ngd_counters)
  bool speculate = true;              // gvi_ngd_step: queue the next gradients behind the first trial
  bool profile = false;
  bool profile_all = false;           // events around every moments / cost launch (else: set 0, full pass only)
  int profile_every = 1;              // on = 3: bracket only every 8th dominant launch (an event pair costs ~14 us of queue gaps)
  long profile_count = 0;
  int target_waves = 2048;
  bool mirror = true;                 // GVI_MIRROR=0: never pair z with -z (A/B; results agree to rounding)
  int split_flush = SPLIT_FLUSH;      // GVI_SPLIT_FLUSH=0: plain recursive sums in the split kernel (A/B of the (24,7) rounding)
  bool sreg_pipe = true;              // GVI_SREG_PIPE=0: full pass on the compiler-scheduled body (A/B; bit-identical results)
  bool no_scost = false;              // GVI_NO_SCOST=1: cost pass on the one-factor-per-wave kernel (A/B)
  // sum-of-squares sets on the sign-orbit kernel (kernels_orbit.hpp) when the table decomposes; GVI_ORBIT=0: the
  // lane-per-point kernels (A/B; results agree to rounding)
  bool orbit = true;
  // full moments pass of the resident iteration as ONE launch (kernels_fused.hpp) where every set runs the sign-orbit kernel;
  // GVI_FUSED=0 / option "fused": the three-launch route prep -> psi -> epilogue (A/B; same numbers at four chunks per factor)
  bool fused = true;
  int orbit_waves = 4096;             // waves the orbit launch of one set aims for (chunks = orbit_waves / K, <= tiles)
  // prep: cyclic Jacobi stops when (sum of squared off-diagonals) <= jacobi_tol * (sum of squared diagonals); option
  // "jacobi_tol_exp" / GVI_JACOBI_TOL_EXP sets 10^value
  double jacobi_tol = 1e-34;
  // sum-of-squares sets: per-pass products from the Cholesky factor of the marginal instead of its symmetric square root
  // (same moments -- the quadrature is exact there -- without the Jacobi sweeps); GVI_CHOL_SQRT=0 / option "chol_sqrt"
  bool prefer_opsi = false;           // gvi_set_variant(7)
  bool chol_sqrt = true;
  // option "trust_table_degree": a table handed to gvi_factors_add_table IS the Smolyak rule of the degree it is added
  // under (e.g. the generator's own table, produced once and broadcast to the other ranks); it may then take every route
  // the generated table takes (Cholesky factor for sum-of-squares psi).  Default 0: a caller's table keeps the symmetric
  // root, as the reference maps the nodes.
  bool trust_table_degree = false;
  int orbit_min_tiles = 6;            // GVI_ORBIT_MIN_TILES
  bool orbit_stack = true;            // two-set launch: block b takes item b of both sets (GVI_ORBIT_STACK=0: set 1 behind set 0)
  int orbit_copies = 8;               // private LDS copies of the accumulators (1, 2, 4, 8, 16; fewer when LDS is short)
  static constexpr int cost_chunk_mult = 8;   // cost pass of the F-factor kernel: chunks per factor relative to the full pass
  static constexpr int scost_f = 2;           // factors per wave of the cost kernel (the four-factor form measured no better: removed)
  // run_moments in planning mode: the launch that WOULD be issued is recorded instead (pair fusion of two sets)
  // capture_any: kind 3 = "a lane-per-point launch of this set would go out with these arguments and this grid" (whichever
  // register kernel): the three-set launch of the planning graph (moments_planar3_kernel)
  struct Deferred { int kind = -1; MomArgs a; dim3 grid; int d = 0, m = 0; OrbitArgs oa; int smax = 0; bool all_pos = false; bool capture_any = false; };
  Deferred* defer = nullptr;
  int update_rule = 0;                // 0 natural gradient (NGD-GH), 1 proximal / JKO (ProxGVI-GH)
  bool pair_fuse = true;              // GVI_NO_PAIR=1: one launch per set
  bool defer_gather = false;          // set around the trial-state refresh (a prep of that slot always follows)
  bool fuse_gather = true;            // GVI_NO_FUSE_GATHER=1: stand-alone gather launch before the trial's prep
  hipEvent_t fork = nullptr;
  // side-stream solve: the chain solve of the gradients runs beside the trial factorisation (independent given Vddmu)
  bool side_solve = true;             // GVI_SIDE_SOLVE=0 keeps everything on one stream
  hipStream_t side = nullptr;
  hipEvent_t ev_grad = nullptr, ev_solve[2] = {nullptr, nullptr};
  bool solve_pending[2] = {false, false};
  // dual_chain (GVI_DUAL_CHAIN=0 restores the side stream): the gradient solve is not launched at once but parked, and goes
  // out fused with the next trial factorisation (same three launches, no fork / join events)
  bool dual_chain = true;
  bool solve_deferred[2] = {false, false};
  // assemble-on-load (GVI_ASM_ON_LOAD=0 / option "assemble_on_load": the stand-alone assemble launch): on chain-structured
  // graphs the ordered assemble of (g, V_D, V_U) is not launched behind the factor pass but done by the first pass of the
  // chain operations that consume it (kernels_chain.hpp, AsmList); asm_pending[gb]: gradient buffer gb is NOT assembled yet
  bool asm_on_load = true;
  bool asm_pending[2] = {false, false};
  DevMem Wbuf2, Ibuf2;                // second BCR workspace (the two chains are in flight together)
  // merged top + backward launch of the chain (chain_launch.hpp::ChainSync): a ring of word pairs, one pair per launch
  DevMem chain_sync;
  unsigned chain_seq = 0;
  bool chain_merge = true;
  bool chain_merge_fault = false;     // option chain_merge = 2: the hand-over word is stored wrong (test of the bounded wait)
  DevMem tail_counter;                // arrival counter of cost_tail_kernel (last block reduces)
  DevMem epi_counter;                 // two-level arrival counters of epilogue_all_kernel's tail
  // trial precision formed inside the first pass of the next factorisation (see ChainArgs::mix*)
  struct Mix { const double* VD = nullptr; const double* VU = nullptr; double* outD = nullptr; double* outU = nullptr; double step = 0.0; } mix;
  hipStream_t chain_stream = nullptr; // stream of the chain launches being queued (null: ctx->stream)
  int chain_ws = 0;
  // ---- sharded factors (one process per GPU): the exchange steps run inside the library, on the context stream ----
  struct Dist {
    int rank = 0, world = 1;
    // all-gather of `count` doubles per rank: send [count] -> recv [world][count], ordered after / before the work on `stream`
    gvi_allgather_fn fn = nullptr;       // host-supplied collective (tests: gloo; hosts with their own transport: MPI ...)
    void* user = nullptr;
    void* rccl_lib = nullptr;            // dlopen handle; the default transport
    void* comm = nullptr;                // ncclComm_t
    int (*ncclAllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*ncclCommDestroy)(void*) = nullptr;
    DevMem send, recv, ranges;           // packed state records; [world][2] state ranges (device); cost scratch
    std::vector<int32_t> lo, hi;         // owned state range [lo, hi] of every rank (records sent)
    int maxlen = 0;
    bool rec_fresh = false;           // the last assemble wrote this rank's exchange records itself (no pack launch)
    bool ranges_valid = false;
  } dist;
  // pipelined iterations (gvi_ngd_run): predicate of the launches being queued, device-side accept decision of the tails,
  // ring of two host slots (a speculatively queued iteration publishes into the other one)
  const double* cur_pred = nullptr;
  double cur_pred_val = 0.0;
  DevMem pipe_dev;                    // doubles: [0..1] accept words (ring), [2..3] cost of the state in NGD slot 0 / 1
  // Option "pipeline" (GVI_PIPELINE=0 switches it off): gvi_ngd_run queues iteration i + 1 (predicated) before it has read
  // the cost of iteration i.  Round 2 measured no gain (129.7 vs 130.5 us per C3 iteration: the early publish of the trial
  // cost already gave the host its head start); with the round-3 kernels the iteration is short enough for the host's
  // hand-over to show again: 109.3 -> 107.4 us at C3, 45.8 -> 44.3 us at C2.  On by default since then.
  bool pipeline = true;
  bool pipe_tail = false;             // tails take the accept decision on the device
  bool pipe_c0_imm = true;            // current cost as an immediate (known to the host) or from pipe_dev
  double pipe_c0 = 0.0;
  int pub_ring = 0;                   // host slot the next publish goes to
  double* host_slot = nullptr;        // host-mapped {cost value, sequence}: one 16-byte device store (publish_to_host)
  // kernels whose dynamic-LDS limit was raised on THIS context's device (the attribute is per device, and a
  // process may hold contexts on several devices)
  std::set<const void*> lds_attr_done;
  double seq = 0.0;
  // stage timing (gvi_profile_stages): event pairs around the chain launches / the factor pass / the assemble of the
  // resident iteration.  A pair costs ~14 us of queue gaps, so it is switched on for a few iterations OUTSIDE any timed region.
  bool stage_prof = false;
  struct StageRec { int stage; hipEvent_t e0, e1; };
  std::vector<StageRec> stage_recs;
  std::vector<hipEvent_t> stage_pool;
  int spin_ms = 2;                    // wall-time bound of the host spin on the publish word (GVI_SPIN_MS)
  double* host_slot_dev = nullptr;
  bool safe_publish = false;          // option "safe_publish": checked publish + release / acquire arrival counters (device_common.hpp)
  DevMem dbg_log;                     // gvi_debug_cost_log: ring of the costs the epilogue tails published, indexed by sequence
  int dbg_mask = 0;
};

namespace {

thread_local std::string g_noctx_err;

gvi_status fail(gvi_ctx* c, gvi_status s, const std::string& msg) {
  if (c) c->err = msg; else g_noctx_err = msg;
  return s;
}

#define HIPCK(ctx, expr)                                                                       \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail(ctx, GVI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
  } while (0)
#define GVICK(expr)                          \
  do {                                       \
    gvi_status s__ = (expr);                 \
    if (s__ != GVI_OK) return s__;           \
  } while (0)

// raise a kernel's dynamic-LDS limit once per context (= per device)
gvi_status allow_lds(gvi_ctx* c, const void* func, int bytes) {
  if (c->lds_attr_done.count(func)) return GVI_OK;
  HIPCK(c, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  c->lds_attr_done.insert(func);
  return GVI_OK;
}

// stage timing: StageScope brackets the launches queued during its lifetime with an event pair (no-op unless switched on)
enum { STAGE_CHAIN = 0, STAGE_FACTORS = 1, STAGE_ASSEMBLE = 2, STAGE_COUNT = 3 };
struct StageScope {
  gvi_ctx* c;
  int idx = -1;
  StageScope(gvi_ctx* ctx, int stage) : c(ctx) {
    if (!c->stage_prof || c->stage_recs.size() >= 512) return;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!c->stage_pool.empty()) { e = c->stage_pool.back(); c->stage_pool.pop_back(); }
      else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
      return e;
    };
    gvi_ctx::StageRec r{stage, get(), get()};
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, c->stream);
    c->stage_recs.push_back(r);
    idx = (int)c->stage_recs.size() - 1;
  }
  ~StageScope() { if (idx >= 0) (void)hipEventRecord(c->stage_recs[idx].e1, c->stream); }
};

size_t bt_count(const gvi_ctx* c) { return (size_t)(2 * c->T - 1) * c->n * c->n; }   // [D | U]
size_t nn_(const gvi_ctx* c) { return (size_t)c->n * c->n; }

gvi_status h2d(gvi_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return GVI_OK;
}
gvi_status d2h(gvi_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return GVI_OK;
}
gvi_status sync(gvi_ctx* c) {
  // polled for a while before the blocking wait: a blocked hipStreamSynchronize wakes up 50-70 us after the stream has drained
  // (bench.py, closing synchronisation of a 2 ms timed region), a poll within a few
  const auto t0 = std::chrono::steady_clock::now();
  for (hipStream_t st : {c->stream, c->side}) {
    if (!st) continue;
    hipError_t e = hipStreamQuery(st);
    while (e == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2)) e = hipStreamQuery(st);
    if (e != hipSuccess && e != hipErrorNotReady) HIPCK(c, e);
    HIPCK(c, hipStreamSynchronize(st));
  }
  return GVI_OK;
}

gvi_status upload_table(gvi_ctx* c, Table& t, int d, int p, int64_t N, const double* Z, const double* w) {
  t.d = d; t.p = p; t.N = N; t.Np = (N + 255) / 256 * 256;   // tile kernels walk 256-point tiles
  std::vector<double> zt((size_t)d * t.Np, 0.0), wp(t.Np, 0.0);
  for (int64_t i = 0; i < N; ++i) {
    wp[i] = w[i];
    for (int a = 0; a < d; ++a) zt[(size_t)a * t.Np + i] = Z[(size_t)i * d + a];
  }
  HIPCK(c, t.Zt.ensure(zt.size() * 8));
  HIPCK(c, t.w.ensure(wp.size() * 8));
  HIPCK(c, hipMemcpy(t.Zt.p, zt.data(), zt.size() * 8, hipMemcpyHostToDevice));
  HIPCK(c, hipMemcpy(t.w.p, wp.data(), wp.size() * 8, hipMemcpyHostToDevice));
  if (d <= 12) {                          // shapes of the sreg kernels
    const size_t ntile = (size_t)t.Np / 64, rows = (size_t)d + 1;
    std::vector<double> zq(ntile * rows * 64, 0.0);
    for (int64_t i = 0; i < N; ++i) {
      const size_t base = ((size_t)i / 64) * rows * 64 + (size_t)i % 64;
      for (int a = 0; a < d; ++a) zq[base + (size_t)a * 64] = Z[(size_t)i * d + a];
      zq[base + (size_t)d * 64] = w[i];
    }
    HIPCK(c, t.Zq.ensure(zq.size() * 8));
    HIPCK(c, hipMemcpy(t.Zq.p, zq.data(), zq.size() * 8, hipMemcpyHostToDevice));
    // Mirror-half table.  Rows are in ascending lexicographic order, so the mirror image of row i is row N - 1 - i; the
    // grid of nwspgr (sym = 1: every non-zero coordinate reflected with the weight copied, nwspgr.m:108-126) is exactly
    // symmetric.  A caller-supplied table that is not keeps the unpaired kernels.
    bool sym = true;
    for (int64_t i = 0; i < N / 2 && sym; ++i) {
      sym = w[i] == w[N - 1 - i];
      for (int a = 0; a < d && sym; ++a) sym = Z[(size_t)i * d + a] == -Z[(size_t)(N - 1 - i) * d + a];
    }
    if (sym && (N & 1))
      for (int a = 0; a < d; ++a) sym = sym && Z[(size_t)(N / 2) * d + a] == 0.0;
    t.Nm = t.Nmp = 0;
    t.Zm.release();
    if (sym) {
      t.Nm = N - N / 2;                                   // upper half (first non-zero coordinate positive) + the origin
      t.Nmp = (t.Nm + 63) / 64 * 64;
      std::vector<double> zm((size_t)t.Nmp / 64 * rows * 64, 0.0);
      for (int64_t j = 0; j < t.Nm; ++j) {
        const int64_t i = N / 2 + j;                      // j = 0 is the origin when N is odd
        const size_t base = ((size_t)j / 64) * rows * 64 + (size_t)j % 64;
        for (int a = 0; a < d; ++a) zm[base + (size_t)a * 64] = Z[(size_t)i * d + a];
        zm[base + (size_t)d * 64] = ((N & 1) && j == 0) ? 0.5 * w[i] : w[i];   // the origin is its own mirror image
      }
      HIPCK(c, t.Zm.ensure(zm.size() * 8));
      HIPCK(c, hipMemcpy(t.Zm.p, zm.data(), zm.size() * 8, hipMemcpyHostToDevice));
    }
  }
  // 8-bit node codes for the split kernel: a Smolyak table has a few dozen distinct node values
  t.coded = false;
  if (d >= 16 && d % 4 == 0) {
    std::vector<double> vals{0.0};                                  // sorted distinct values
    bool ok = true;
    std::vector<uint32_t> packed((size_t)(d / 4) * t.Np, 0u);
    // pass 1: distinct values
    double last = 0.0;
    for (size_t e = 0; e < (size_t)N * d && ok; ++e) {
      const double v = Z[e];
      if (v == last) continue;
      last = v;
      auto it = std::lower_bound(vals.begin(), vals.end(), v);
      if (it == vals.end() || *it != v) {
        if (vals.size() >= 256) ok = false;
        else vals.insert(it, v);
      }
    }
    if (ok) {
      const uint32_t zero_code = (uint32_t)(std::lower_bound(vals.begin(), vals.end(), 0.0) - vals.begin());
      uint32_t zero_word = zero_code | zero_code << 8 | zero_code << 16 | zero_code << 24;
      std::fill(packed.begin(), packed.end(), zero_word);          // pad points decode to z = 0
      for (int64_t i = 0; i < N; ++i)
        for (int a = 0; a < d; ++a) {
          const uint32_t code = (uint32_t)(std::lower_bound(vals.begin(), vals.end(), Z[(size_t)i * d + a]) - vals.begin());
          uint32_t& word = packed[(size_t)(a / 4) * t.Np + i];
          word = (word & ~(255u << (8 * (a % 4)))) | code << (8 * (a % 4));
        }
      vals.resize(256, 0.0);
      HIPCK(c, t.codes.ensure(packed.size() * 4));
      HIPCK(c, t.lut.ensure(256 * 8));
      HIPCK(c, hipMemcpy(t.codes.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      HIPCK(c, hipMemcpy(t.lut.p, vals.data(), 256 * 8, hipMemcpyHostToDevice));
      t.coded = true;
    }
  }
  // sign-orbit form: every table whose points come in complete sign orbits with one weight (nwspgr's do by construction)
  t.orb = OrbitHost();
  t.orb_bounds.clear();
  if (N > 1) {
    OrbitHost o = build_orbits(d, N, Z, w, true);
    if (o.ok && o.smax >= 1) {
      HIPCK(c, t.orb_cpk.ensure(o.cpk.size() * 8));
      HIPCK(c, t.orb_rpk.ensure(o.rpk.size() * 8));
      HIPCK(c, hipMemcpy(t.orb_rpk.p, o.rpk.data(), o.rpk.size() * 8, hipMemcpyHostToDevice));
      HIPCK(c, t.orb_mag.ensure(o.mag.size() * 8));
      HIPCK(c, t.orb_w.ensure(o.w.size() * 8));
      HIPCK(c, hipMemcpy(t.orb_cpk.p, o.cpk.data(), o.cpk.size() * 8, hipMemcpyHostToDevice));
      HIPCK(c, hipMemcpy(t.orb_mag.p, o.mag.data(), o.mag.size() * 8, hipMemcpyHostToDevice));
      HIPCK(c, hipMemcpy(t.orb_w.p, o.w.data(), o.w.size() * 8, hipMemcpyHostToDevice));
      std::vector<uint64_t>().swap(o.cpk);
      std::vector<uint64_t>().swap(o.rpk);
      std::vector<double>().swap(o.mag);
      std::vector<double>().swap(o.w);
      t.orb = std::move(o);
    }
  }
  return GVI_OK;
}

// device address of the current publish slot (four doubles per ring entry); bit 0 tags the checked form (publish_to_host)
double* pub_slot(const gvi_ctx* c) {
  return (double*)((unsigned long long)(c->host_slot_dev + 4 * c->pub_ring) | (c->safe_publish ? 1ull : 0ull));
}

FactorSet* get_set(gvi_ctx* c, int id) {
  if (!c || id < 0 || id >= (int)c->sets.size()) return nullptr;
  return c->sets[id].get();
}

// which (d, m) pairs have a register-kernel instantiation
bool reg_supported(int kind, int d, int m) {
  if (kind == KIND_RANGE_1D) return d == 1;
  if (kind == KIND_HINGE_SDF_2D) return d == 2 || d == 4 || d == 6;
  if (kind == KIND_HINGE_SDF_2D_BODY) return d == 3 || d == 6;
  if (kind == KIND_HINGE_SDF_3D) return d == 3 || d == 6;
  if (kind == KIND_QUAD_PRIOR) return (d == 2 && m == 1) || (d == 4 && m == 2) || (d == 6 && m == 3) ||
                                      (d == 8 && m == 4) || (d == 12 && m == 6);
  if (kind == KIND_FIXED_PRIOR) return d == m && (d == 1 || d == 2 || d == 3 || d == 4 || d == 6 || d == 8 || d == 12);
  return false;
}

// split kernel (one block per factor and chunk): sum-of-squares kinds on a coded table
bool split_supported(const FactorSet& s) {
  return (s.kind == KIND_QUAD_PRIOR || s.kind == KIND_FIXED_PRIOR) && s.table->coded &&
         (s.d == 16 || s.d == 20 || s.d == 24) && (s.m == s.d || s.m == s.d / 2);
}

template <int D, int R>
gvi_status launch_split(gvi_ctx* c, const MomArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)SPLIT_LDS_DOUBLES(D) * 8;
  GVICK(allow_lds(c, (const void*)moments_split_kernel<D, R, true>, (int)lds));
  GVICK(allow_lds(c, (const void*)moments_split_kernel<D, R, false>, (int)lds));
  if (a.full) hipLaunchKernelGGL((moments_split_kernel<D, R, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((moments_split_kernel<D, R, false>), grid, dim3(256), lds, st, a);
  return GVI_OK;
}

void plan_chunks(gvi_ctx* c, FactorSet& s, bool reg) {
  const int64_t Np = s.table->Np;
  if (reg) {
    const int64_t iters = Np / 256;                       // 256-point tiles (the tile kernel needs whole tiles)
    int64_t nch = std::max<int64_t>(1, (c->target_waves + s.K - 1) / s.K);
    nch = std::min<int64_t>(nch, std::max<int64_t>(1, iters));
    const int64_t per = (iters + nch - 1) / nch;
    s.chunk = per * 256;
    s.nchunk = (int)((Np + s.chunk - 1) / s.chunk);
  } else {
    const int64_t blocks = (Np + 255) / 256;
    int64_t nch = std::max<int64_t>(1, (1024 + s.K - 1) / s.K);
    nch = std::min<int64_t>(nch, blocks);
    const int64_t per = (blocks + nch - 1) / nch;
    s.chunk = per * 256;
    s.nchunk = (int)((Np + s.chunk - 1) / s.chunk);
  }
}

template <int D, typename Psi>
void launch_reg(const MomArgs& a, dim3 grid, hipStream_t st) {
  if (a.full) hipLaunchKernelGGL((moments_reg_kernel<D, Psi, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((moments_reg_kernel<D, Psi, false>), grid, dim3(256), 0, st, a);
}

template <int D, int M>
void launch_sreg(const MomArgs& a, dim3 grid, hipStream_t st, bool pipe) {
  if (a.full && pipe) hipLaunchKernelGGL((moments_sreg_kernel<D, M, true, true>), grid, dim3(256), 0, st, a);
  else if (a.full) hipLaunchKernelGGL((moments_sreg_kernel<D, M, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((moments_sreg_kernel<D, M, false>), grid, dim3(256), 0, st, a);
}

// cost pass with F factors per wave (grid.x shrinks by F); only the headline shapes
bool scost_supported(const FactorSet& s) {
  return (s.kind == KIND_QUAD_PRIOR && s.d == 12) || (s.kind == KIND_FIXED_PRIOR && s.d == 6);
}
template <int F>
void launch_scost(const FactorSet& s, const MomArgs& a, int nchunk, hipStream_t st) {
  const dim3 grid((s.K + 4 * F - 1) / (4 * F), nchunk);
  if (s.kind == KIND_QUAD_PRIOR) hipLaunchKernelGGL((moments_scost_kernel<12, 6, F>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((moments_scost_kernel<6, 6, F>), grid, dim3(256), 0, st, a);
}
void dispatch_scost(gvi_ctx* c, const FactorSet& s, const MomArgs& a, int nchunk, hipStream_t st);

bool sreg_supported(const FactorSet& s) {
  if (s.kind == KIND_QUAD_PRIOR) return s.d == 4 || s.d == 8 || s.d == 12;
  if (s.kind == KIND_FIXED_PRIOR) return s.d == 6 || s.d == 12;
  return false;
}

// scalar-operand register kernel (variant 5)
bool dispatch_sreg(const FactorSet& s, const MomArgs& a, dim3 grid, hipStream_t st, bool pipe) {
  if (s.kind == KIND_QUAD_PRIOR) {
    switch (s.d) {
      case 4: launch_sreg<4, 2>(a, grid, st, pipe); return true;
      case 8: launch_sreg<8, 4>(a, grid, st, pipe); return true;
      case 12: launch_sreg<12, 6>(a, grid, st, pipe); return true;
    }
  }
  if (s.kind == KIND_FIXED_PRIOR) {
    switch (s.d) {
      case 6: launch_sreg<6, 6>(a, grid, st, pipe); return true;
      case 12: launch_sreg<12, 12>(a, grid, st, false); return true;   // M = 12: the pipelined body does not fit 256 registers
    }
  }
  return false;
}

bool dispatch_reg(const FactorSet& s, const MomArgs& a, dim3 grid, hipStream_t st) {
  const int d = s.d;
  if (s.kind == KIND_RANGE_1D && d == 1) { launch_reg<1, PsiRange1D>(a, grid, st); return true; }
  if (s.kind == KIND_HINGE_SDF_2D) {
    switch (d) {
      case 2: launch_reg<2, PsiHingeSdf2D<2>>(a, grid, st); return true;
      case 4: launch_reg<4, PsiHingeSdf2D<4>>(a, grid, st); return true;
      case 6: launch_reg<6, PsiHingeSdf2D<6>>(a, grid, st); return true;
    }
  }
  if (s.kind == KIND_HINGE_SDF_2D_BODY) {
    if (d == 3) { launch_reg<3, PsiHingeSdf<3, KIND_HINGE_SDF_2D_BODY>>(a, grid, st); return true; }
    if (d == 6) { launch_reg<6, PsiHingeSdf<6, KIND_HINGE_SDF_2D_BODY>>(a, grid, st); return true; }
  }
  if (s.kind == KIND_HINGE_SDF_3D) {
    if (d == 3) { launch_reg<3, PsiHingeSdf<3, KIND_HINGE_SDF_3D>>(a, grid, st); return true; }
    if (d == 6) { launch_reg<6, PsiHingeSdf<6, KIND_HINGE_SDF_3D>>(a, grid, st); return true; }
  }
  if (s.kind == KIND_QUAD_PRIOR) {
    switch (d) {
      case 2: launch_reg<2, PsiQuad<2, 1>>(a, grid, st); return true;
      case 4: launch_reg<4, PsiQuad<4, 2>>(a, grid, st); return true;
      case 6: launch_reg<6, PsiQuad<6, 3>>(a, grid, st); return true;
      case 8: launch_reg<8, PsiQuad<8, 4>>(a, grid, st); return true;
      case 12: launch_reg<12, PsiQuad<12, 6>>(a, grid, st); return true;
    }
  }
  if (s.kind == KIND_FIXED_PRIOR) {
    switch (d) {
      case 1: launch_reg<1, PsiQuad<1, 1>>(a, grid, st); return true;
      case 2: launch_reg<2, PsiQuad<2, 2>>(a, grid, st); return true;
      case 3: launch_reg<3, PsiQuad<3, 3>>(a, grid, st); return true;
      case 4: launch_reg<4, PsiQuad<4, 4>>(a, grid, st); return true;
      case 6: launch_reg<6, PsiQuad<6, 6>>(a, grid, st); return true;
      case 8: launch_reg<8, PsiQuad<8, 8>>(a, grid, st); return true;
      case 12: launch_reg<12, PsiQuad<12, 12>>(a, grid, st); return true;
    }
  }
  return false;
}

void dispatch_scost(gvi_ctx* c, const FactorSet& s, const MomArgs& a, int nchunk, hipStream_t st) {
  launch_scost<2>(s, a, nchunk, st);
}

// ---- sign-orbit kernel (kernels_orbit.hpp) ----
bool orbit_supported(const gvi_ctx* c, const FactorSet& s) {
  if (!c->orbit || !(c->variant == 0 || c->variant == 6)) return false;
  if (s.kind != KIND_QUAD_PRIOR && s.kind != KIND_FIXED_PRIOR) return false;
  if (!(s.m == 2 || s.m == 6 || s.m == 12 || (s.m == 14 && s.table->orb.smax <= 4)) || s.d > 32) return false;   // m = 14: the d = 28 priors of the arm graph
  if (s.m == 2 && s.table->orb.smax > 4) return false;          // m = 2 is instantiated for degree <= 5 only
  const OrbitHost& o = s.table->orb;
  return o.ok && o.smax >= 1 && o.smax <= ORBIT_SMAX && !o.tile_s.empty();
}

gvi_status orbit_args(gvi_ctx* c, FactorSet& s, int full, OrbitArgs* out) {
  Table& t = *s.table;
  auto it = t.orb_bounds.find(s.nchunk);
  if (it == t.orb_bounds.end()) {
    const std::vector<int32_t> b = orbit_chunk_bounds(t.orb, s.nchunk);
    auto mem = std::make_unique<DevMem>();
    HIPCK(c, mem->ensure(b.size() * 4));
    HIPCK(c, hipMemcpy(mem->p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    it = t.orb_bounds.emplace(s.nchunk, std::move(mem)).first;
  }
  OrbitArgs a;
  a.H = s.H.d(); a.u0 = s.u0.d(); a.sgn = s.sgn.d(); a.partial = s.partial.d();
  a.K = s.K; a.d = s.d; a.nchunk = s.nchunk;
  a.pred = c->cur_pred; a.pred_val = c->cur_pred_val;
  // private accumulator copies: as many as the kernel's occupancy leaves LDS for (160 KB per CU, 64 KB per block), capped by orbit_copies
  const int waves = ((s.m == 6 && t.orb.smax <= 4) || s.m == 2) ? 4 : ((s.m == 12 && t.orb.smax <= 4) ? 3 : 2);   // launch_orbit's occupancy
  const size_t lds_cap = std::min<size_t>(64 * 1024, 160 * 1024 / waves);
  a.copies = 1;
  while (a.copies * 2 <= c->orbit_copies && (size_t)4 * orbit_lds_doubles(s.d, s.m, a.copies * 2) * 8 <= lds_cap) a.copies *= 2;
  a.ob.cpk = (const uint64_t*)t.orb_cpk.p; a.ob.rpk = (const uint64_t*)t.orb_rpk.p; a.ob.mag = t.orb_mag.d(); a.ob.w = t.orb_w.d();
  a.ob.bounds = it->second->i();
  a.ob.norb_p = t.orb.norb_p; a.ob.w0 = t.orb.w0;
  {   // the tile list as kernel arguments (OrbitDev): class ends and, for up to four chunks, the chunk bounds
    int32_t cum = 0;
    for (int sz = 7; sz >= 0; --sz) {
      if (sz >= 1 && sz <= t.orb.smax) cum += t.orb.cstride[sz] / 64;           // tiles of the class
      a.ob.cend[sz] = sz > t.orb.smax ? 0 : cum;
    }
    for (int sz = 0; sz < 8; ++sz) {
      const bool in = sz >= 1 && sz <= t.orb.smax;
      a.ob.cbase[sz] = in ? t.orb.cbase[sz] : 0;
      a.ob.cgrp[sz] = in ? t.orb.cgrp[sz] : 1;
      a.ob.cstride[sz] = in ? t.orb.cstride[sz] : 0;
    }
    a.ob.nb = 0;
    for (int i = 0; i < 5; ++i) a.ob.bnd[i] = 0;
    if (s.nchunk <= 4) {
      const std::vector<int32_t> b = orbit_chunk_bounds(t.orb, s.nchunk);
      a.ob.nb = s.nchunk;
      for (int i = 0; i <= s.nchunk; ++i) a.ob.bnd[i] = b[(size_t)i];
    }
  }
  (void)full;
  *out = a;
  return GVI_OK;
}

template <int M, int SMAX, int WAVES>
void launch_orbit_t(const OrbitArgs& a, bool full, bool all_pos, dim3 grid, size_t lds, hipStream_t st) {
  if (full && all_pos) hipLaunchKernelGGL((moments_orbit_kernel<M, SMAX, true, false, WAVES>), grid, dim3(256), lds, st, a);
  else if (full) hipLaunchKernelGGL((moments_orbit_kernel<M, SMAX, true, true, WAVES>), grid, dim3(256), lds, st, a);
  else if (all_pos) hipLaunchKernelGGL((moments_orbit_kernel<M, SMAX, false, false, WAVES>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((moments_orbit_kernel<M, SMAX, false, true, WAVES>), grid, dim3(256), lds, st, a);
}

void launch_orbit(const OrbitArgs& a, int m, int smax, bool full, bool all_pos, hipStream_t st) {
  const dim3 grid((a.K + 3) / 4, a.nchunk);
  const size_t lds = (size_t)4 * orbit_lds_doubles(a.d, m, a.copies) * 8;
  if (m == 2) launch_orbit_t<2, 4, 4>(a, full, all_pos, grid, lds, st);
  else if (m == 14) launch_orbit_t<14, 4, 2>(a, full, all_pos, grid, lds, st);
  else if (m == 6 && smax <= 4) launch_orbit_t<6, 4, 4>(a, full, all_pos, grid, lds, st);
  else if (m == 6) launch_orbit_t<6, 6, 2>(a, full, all_pos, grid, lds, st);
  else if (smax <= 4) launch_orbit_t<12, 4, 3>(a, full, all_pos, grid, lds, st);
  else launch_orbit_t<12, 6, 2>(a, full, all_pos, grid, lds, st);
}

// two sets with the same m and sgn = +1 in one launch.  e0 / e1 non-null: the launch itself carries the start / stop events
// (hipExtLaunchKernel: the kernel's own begin / end timestamps, what rocprofv3 reports -- a hipEventRecord pair around a
// 28 us launch reads ~3 us more than the kernel runs)
template <int M, int SMAX, int WAVES>
void launch_orbit_pair_t(const OrbitArgs& a0, const OrbitArgs& a1, bool full, size_t lds, hipStream_t st, hipEvent_t e0, hipEvent_t e1,
                         bool stack) {
  const int nbx0 = (a0.K + 3) / 4, nbx1 = (a1.K + 3) / 4;
  const int nb0 = nbx0 * a0.nchunk, nb1 = nbx1 * a1.nchunk;
  // stacked: block b takes item b of both sets (negative nbx1 tells the kernel); else set 1's blocks behind set 0's
  const int grid = stack ? std::max(nb0, nb1) : nb0 + nb1;
  const int nx1 = stack ? -nbx1 : nbx1;
  if (full)
    hipExtLaunchKernelGGL((moments_orbit_pair_kernel<M, SMAX, true, false, WAVES>), dim3(grid), dim3(256), (uint32_t)lds, st, e0, e1, 0,
                          a0, a1, nbx0, nb0, nx1);
  else
    hipExtLaunchKernelGGL((moments_orbit_pair_kernel<M, SMAX, false, false, WAVES>), dim3(grid), dim3(256), (uint32_t)lds, st, e0, e1, 0,
                          a0, a1, nbx0, nb0, nx1);
}

void launch_orbit_pair(const OrbitArgs& a0, const OrbitArgs& a1, int m, int smax, bool full, hipStream_t st, bool stack,
                       hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  const size_t lds = (size_t)4 * std::max(orbit_lds_doubles(a0.d, m, a0.copies), orbit_lds_doubles(a1.d, m, a1.copies)) * 8;
  if (m == 2) launch_orbit_pair_t<2, 4, 4>(a0, a1, full, lds, st, e0, e1, stack);
  else if (m == 6 && smax <= 4) launch_orbit_pair_t<6, 4, 4>(a0, a1, full, lds, st, e0, e1, stack);
  else if (m == 6) launch_orbit_pair_t<6, 6, 2>(a0, a1, full, lds, st, e0, e1, stack);
  else if (smax <= 4) launch_orbit_pair_t<12, 4, 3>(a0, a1, full, lds, st, e0, e1, stack);
  else launch_orbit_pair_t<12, 6, 2>(a0, a1, full, lds, st, e0, e1, stack);
}

// the full pass of the resident iteration as one launch (kernels_fused.hpp)
template <int M, int SMAX, int WAVES, int D0, int D1>
gvi_status launch_fused_t(gvi_ctx* c, const FusedArgs& A, unsigned grid, size_t lds, int dmax, int copies, int items, hipEvent_t e0, hipEvent_t e1) {
  if (lds > 64 * 1024) GVICK(allow_lds(c, (const void*)factor_fused_kernel<M, SMAX, WAVES, D0, D1>, (int)lds));
#ifdef GVI_FUSED_TIMING
  const int dbg = getenv("GVI_FUSED_DBG") ? atoi(getenv("GVI_FUSED_DBG")) : 0;
  static unsigned long long* stamps = nullptr;
  static int nprint = 0;
  if ((dbg & 8) && !stamps) { if (hipMalloc(&stamps, 1120 * 8) != hipSuccess) stamps = nullptr; else (void)hipMemset(stamps, 0, 1120 * 8); }
#else
  const int dbg = 0;
  unsigned long long* stamps = nullptr;
#endif
  hipExtLaunchKernelGGL((factor_fused_kernel<M, SMAX, WAVES, D0, D1>), dim3(grid), dim3(256), (uint32_t)lds, c->stream, e0, e1, 0, A, dmax, copies, items,
                        (dbg & 8) ? stamps : (unsigned long long*)nullptr);
#ifdef GVI_FUSED_TIMING
  if ((dbg & 8) && stamps && ++nprint > 200 && nprint <= 202) {          // a few warm launches, 100 MHz ticks
    unsigned long long h[1120];
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 8; ++blk)
      for (int w = 0; w < 4; ++w)
        for (int it = 0; it < 2; ++it) {
          const unsigned long long* ws = h + 352 + ((blk * 4 + w) * 2 + it) * 12;
          fprintf(stderr, "[walk stamps] block %4d wave %d item %d: shader cycles prologue %llu | s=1 %llu (%llu tiles) | s=2 %llu (%llu) | s=3 %llu (%llu) | s=4 %llu (%llu) | reduction %llu\n",
                  blk * 146, w, it, ws[0], ws[1], ws[7], ws[2], ws[8], ws[3], ws[9], ws[4], ws[10], ws[5]);
        }
    for (int w = 0; w < 4; ++w) {
      fprintf(stderr, "[prep stamps] block 0 wave %d, us after the wave's start: prologue [arguments warm | first scalars | barrier] %.2f %.2f %.2f; gather [entry | loads back | stores issued] %.2f %.2f %.2f; products [Sigma row loaded | Cholesky | L^-1 | LDS | Lam + stores | H, u0]:",
              w, (double)(long long)(h[256 + 16 * w + 11] - h[8 * w]) * 0.01, (double)(long long)(h[256 + 16 * w + 12] - h[8 * w]) * 0.01,
              (double)(long long)(h[256 + 16 * w + 13] - h[8 * w]) * 0.01, (double)(long long)(h[256 + 16 * w + 8] - h[8 * w]) * 0.01, (double)(long long)(h[256 + 16 * w + 9] - h[8 * w]) * 0.01, (double)(long long)(h[256 + 16 * w + 10] - h[8 * w]) * 0.01);
      for (int i = 0; i < 6; ++i) fprintf(stderr, " %.2f", (double)(long long)(h[256 + 16 * w + i] - h[8 * w]) * 0.01);
      fprintf(stderr, "\n");
    }
    unsigned long long t0 = ~0ull;
    for (int blk = 0; blk < 8; ++blk) if (h[blk * 32] && h[blk * 32] < t0) t0 = h[blk * 32];
    for (int blk = 0; blk < 8; ++blk)
      for (int w = 0; w < 4; ++w) {
        const unsigned hw = (unsigned)h[320 + blk * 4 + w], xcc = (unsigned)(h[320 + blk * 4 + w] >> 32);
        fprintf(stderr, "[fused stamps] block %4d wave %d (xcc %u se %u cu %2u simd %u slot %2u; absolute, us after the first start):", blk * 146, w,
                xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %.2f", h[blk * 32 + w * 8 + i] ? (double)(long long)(h[blk * 32 + w * 8 + i] - t0) * 0.01 : -1.0);
        fprintf(stderr, "\n");
      }
  }
#endif
  return GVI_OK;
}

// prep (sqrt / inverse / psi operands) for one set at (mu, Sigma) device pointers
gvi_status run_prep(gvi_ctx* c, FactorSet& s, const double* mu, const double* Sigma, int slot = -1,
                    hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  if (slot >= 0 && s.prep_slot == slot) return GVI_OK;     // products of this slot are still resident
  s.prep_slot = slot;
  if (s.K == 0) return GVI_OK;
  const int d = s.d, dp = d + (d & 1);
  const size_t lds = (size_t)(4 * d * d + 2 * dp + 3 * d) * 8 + (size_t)dp * 4 + 16;
  if (d <= 8) hipLaunchKernelGGL(prep_kernel<1>, dim3(s.K), dim3(64), lds, st, s.dev(), mu, Sigma);
  else if (d <= 16) hipLaunchKernelGGL(prep_kernel<4>, dim3(s.K), dim3(64), lds, st, s.dev(), mu, Sigma);
  else if (d <= 32) hipLaunchKernelGGL(prep_kernel<16>, dim3(s.K), dim3(64), lds, st, s.dev(), mu, Sigma);
  else return fail(c, GVI_ERR_UNSUPPORTED, "factor dimension > 32");
  HIPCK(c, hipGetLastError());
  return GVI_OK;
}

// ---- sign-orbit kernel for the non-polynomial psi kinds (kernels_orbit_psi.hpp) ----
// Selected by gvi_set_variant(7) only.  Measured (profiles/r04_*): with a non-polynomial psi the evaluations themselves
// dominate and a lane that walks an orbit evaluates its 2^s points ONE AFTER THE OTHER -- planar hinge factors (d = 4, p = 7):
// 30.8 us against 19.0 us of the lane-per-point register kernel; the arm graph's 129 factors x 421 points: 120 us slower per
// pass than the generic kernel, which keeps 54 k points in flight.  It stays as the A/B leg and as the parity cross-check
// (three independent kernels for these kinds); the arm graph's 0.27 ms launch was its d = 28 PRIOR set on the generic kernel,
// which now takes the m = 14 instance of the sign-orbit kernel.
bool orbit_psi_supported(const gvi_ctx* c, const FactorSet& s, bool reg) {
  if (!c->orbit) return false;
  (void)reg;
  if (!(c->variant == 0 && c->prefer_opsi)) return false;
  if (!(s.kind == KIND_RANGE_1D || s.kind == KIND_HINGE_SDF_2D || s.kind == KIND_HINGE_SDF_2D_BODY || s.kind == KIND_HINGE_SDF_3D ||
        s.kind == KIND_HINGE_SDF_3D_ARM)) return false;
  const OrbitHost& o = s.table->orb;
  if (!(o.ok && o.smax >= 1 && o.smax <= 4 && !o.tile_s.empty()) || s.d > 32) return false;
  if (s.kind == KIND_HINGE_SDF_3D_ARM && (s.arm_ndof < 1 || s.arm_ndof > orbit_psi_rows(KIND_HINGE_SDF_3D_ARM))) return false;
  return orbit_psi_rows(s.kind) <= s.d;
}

template <int KIND>
gvi_status launch_orbit_psi_t(gvi_ctx* c, const OrbitPsiArgs& a, bool full, dim3 grid, size_t lds, hipStream_t st) {
  if (full) {
    if (lds > 64 * 1024) GVICK(allow_lds(c, (const void*)moments_orbit_psi_kernel<KIND, true>, (int)lds));
    hipLaunchKernelGGL((moments_orbit_psi_kernel<KIND, true>), grid, dim3(256), lds, st, a);
  } else {
    hipLaunchKernelGGL((moments_orbit_psi_kernel<KIND, false>), grid, dim3(256), lds, st, a);
  }
  return GVI_OK;
}

gvi_status launch_orbit_psi(gvi_ctx* c, FactorSet& s, const double* mu, int full, hipStream_t st) {
  OrbitArgs oa;
  GVICK(orbit_args(c, s, full, &oa));                       // table pointers, class layout, chunk bounds
  OrbitPsiArgs a;
  a.f = s.dev(); a.mu = mu; a.partial = s.partial.d(); a.nchunk = s.nchunk;
  a.pred = c->cur_pred; a.pred_val = c->cur_pred_val; a.ob = oa.ob;
  const int NR = orbit_psi_rows(s.kind);
  a.copies = 1;
  while (a.copies * 2 <= c->orbit_copies && (size_t)4 * orbit_psi_lds_doubles(s.d, NR, a.copies * 2, true) * 8 <= 64 * 1024) a.copies *= 2;
  const size_t lds = (size_t)4 * orbit_psi_lds_doubles(s.d, NR, a.copies, full != 0) * 8;
  const dim3 grid((s.K + 3) / 4, s.nchunk);
  switch (s.kind) {
    case KIND_RANGE_1D: return launch_orbit_psi_t<KIND_RANGE_1D>(c, a, full != 0, grid, lds, st);
    case KIND_HINGE_SDF_2D: return launch_orbit_psi_t<KIND_HINGE_SDF_2D>(c, a, full != 0, grid, lds, st);
    case KIND_HINGE_SDF_2D_BODY: return launch_orbit_psi_t<KIND_HINGE_SDF_2D_BODY>(c, a, full != 0, grid, lds, st);
    case KIND_HINGE_SDF_3D: return launch_orbit_psi_t<KIND_HINGE_SDF_3D>(c, a, full != 0, grid, lds, st);
    default: return launch_orbit_psi_t<KIND_HINGE_SDF_3D_ARM>(c, a, full != 0, grid, lds, st);
  }
}

// moments (full=1) or cost (full=0) pass for one set; prep must have run for (mu, Sigma).
gvi_status run_moments(gvi_ctx* c, FactorSet& s, const double* mu, const double* psi_ext, int full,
                       hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  if (s.K == 0) { s.nchunk = 1; s.chunk = s.table->Np; s.use_reg = s.use_split = s.use_orbit = s.use_opsi = false; return GVI_OK; }   // empty shard
  if (s.kind == KIND_HINGE_SDF_3D_ARM && !psi_ext && !s.arm.p)
    return fail(c, GVI_ERR_STATE, "HINGE_SDF_3D_ARM set without an arm model: call gvi_factors_set_arm");
  if (s.kind >= KIND_HINGE_SDF_2D && !psi_ext && s.sdf_rows == 0)
    return fail(c, GVI_ERR_STATE, "HINGE_SDF set without a grid: call gvi_factors_set_sdf2d / gvi_factors_set_sdf3d");
  bool reg = reg_supported(s.kind, s.d, s.m) && !psi_ext && c->variant != 1;
  if (c->variant == 2 && !reg && !psi_ext)
    return fail(c, GVI_ERR_UNSUPPORTED, "register kernel not instantiated for this (kind, d)");
  const bool closed = s.closed_form && !psi_ext;
  const bool orbit = !closed && !psi_ext && orbit_supported(c, s);
  const bool opsi = !closed && !orbit && !psi_ext && orbit_psi_supported(c, s, reg);
  const bool split = !closed && !orbit && !opsi && !reg && !psi_ext && c->variant != 1 && split_supported(s);
  if (closed) { reg = false; s.chunk = s.table->Np; s.nchunk = 1; }
  else if (orbit || opsi) {
    reg = false;
    const int64_t tiles = (int64_t)s.table->orb.tile_s.size();
    // chunks: enough waves to fill the chip, but at least orbit_min_tiles tiles per wave -- a wave's prologue (H into LDS,
    // zeroing the accumulator copies) and its reduction of the copies cost about as much as two tiles
    const int64_t by_waves = std::max<int64_t>(1, (c->orbit_waves + s.K - 1) / s.K);
    const int64_t by_tiles = std::max<int64_t>(1, tiles / std::max(1, c->orbit_min_tiles));
    s.nchunk = (int)std::min<int64_t>(tiles, std::min(by_waves, by_tiles));
    s.chunk = s.table->Np;
  } else if (split) {
    const int64_t tiles = s.table->Np / 64;                          // ~2 blocks per CU: K * nchunk >= 1024
    int64_t nch = std::max<int64_t>(1, (1024 + s.K - 1) / s.K);
    nch = std::max<int64_t>(nch, (tiles + 16383) / 16384);         // <= 16k points per lane: bounds the rounding of the
    nch = std::min<int64_t>(nch, tiles);                           // sequential sums (|w|_1 ~ 2e7 at (24,7))
    s.chunk = (tiles + nch - 1) / nch * 64;
    s.nchunk = (int)((s.table->Np + s.chunk - 1) / s.chunk);
  } else plan_chunks(c, s, reg);
  if (reg && !full && !closed && !orbit && (c->variant == 5 || c->variant == 0) && scost_supported(s) && !c->no_scost) {
    // F factors per wave: keep the wave count up with more, shorter chunks
    const int64_t iters = s.table->Np / 256;
    int64_t nch = std::min<int64_t>(std::max<int64_t>(1, (int64_t)s.nchunk * c->cost_chunk_mult), std::max<int64_t>(1, iters));
    const int64_t per = (iters + nch - 1) / nch;
    s.chunk = per * 256;
    s.nchunk = (int)((s.table->Np + s.chunk - 1) / s.chunk);
  }
  s.use_reg = reg;
  s.use_split = split;
  s.use_orbit = orbit;
  s.use_opsi = opsi;
  const size_t need = (size_t)s.K * s.nchunk * npairs(s.d) * 8;
  HIPCK(c, s.partial.ensure(need));
  MomArgs a;
  s.use_mirror = c->mirror;
  a.f = s.dev(); a.mu = mu; a.psi_ext = psi_ext; a.partial = s.partial.d();
  a.chunk = s.chunk; a.nchunk = s.nchunk; a.full = full; a.flush = c->split_flush;
  a.mchunk = 0;
  a.pred = c->cur_pred; a.pred_val = c->cur_pred_val;
  if (s.table->Nmp > 0) {                                   // same number of chunks over the mirror-half table
    const int64_t tiles = s.table->Nmp / 64;
    a.mchunk = (tiles + s.nchunk - 1) / s.nchunk * 64;
  }
  const int which = full ? 0 : 1;
  bool prof = !c->defer && c->profile && (c->profile_all || (full && &s == c->sets[0].get()));
  if (prof && !c->profile_all && (c->profile_count++ % c->profile_every) != 0) prof = false;
  if (prof) {
    for (int e = 0; e < 2; ++e)
      if (!s.ev[which][e]) HIPCK(c, hipEventCreate(&s.ev[which][e]));
    HIPCK(c, hipEventRecord(s.ev[which][0], st));
  }
  if (closed) {
    hipLaunchKernelGGL(moments_closed_kernel, dim3(s.K), dim3(64), 0, st, a);
  } else if (orbit) {
    OrbitArgs oa;
    GVICK(orbit_args(c, s, full, &oa));
    if (c->defer) {
      c->defer->kind = 2; c->defer->oa = oa; c->defer->d = s.d; c->defer->m = s.m;
      c->defer->smax = s.table->orb.smax; c->defer->all_pos = s.all_pos;
      return GVI_OK;
    }
    launch_orbit(oa, s.m, s.table->orb.smax, full != 0, s.all_pos, st);
  } else if (opsi) {
    GVICK(launch_orbit_psi(c, s, mu, full, st));
  } else if (split) {
    const dim3 grid(s.K, s.nchunk);
    const int R = (s.m + 3) / 4;
    if (s.d == 16 && R == 2) GVICK((launch_split<16, 2>(c, a, grid, st)));
    else if (s.d == 16) GVICK((launch_split<16, 4>(c, a, grid, st)));
    else if (s.d == 20 && R == 3) GVICK((launch_split<20, 3>(c, a, grid, st)));
    else if (s.d == 20) GVICK((launch_split<20, 5>(c, a, grid, st)));
    else if (R == 3) GVICK((launch_split<24, 3>(c, a, grid, st)));
    else GVICK((launch_split<24, 6>(c, a, grid, st)));
  } else if (reg) {
    dim3 grid((s.K + 3) / 4, s.nchunk);
    bool done = false;
    if (c->defer && c->defer->capture_any) {
      c->defer->kind = 3; c->defer->a = a; c->defer->grid = grid; c->defer->d = s.d; c->defer->m = s.m;
      return GVI_OK;
    }
    // auto: psi operands from SGPRs where instantiated (fastest for both passes); otherwise the operand-
    // resident kernel for the cost pass and the LDS-operand kernel for the full pass
    if ((c->variant == 5 || c->variant == 0) && !full && scost_supported(s) && !c->no_scost) {
      if (c->defer && c->scost_f == 2) {
        c->defer->kind = 1; c->defer->a = a; c->defer->grid = dim3((s.K + 7) / 8, s.nchunk); c->defer->d = s.d; c->defer->m = s.m;
        return GVI_OK;
      }
      dispatch_scost(c, s, a, s.nchunk, st);
      done = true;
    }
    if (!done && (c->variant == 5 || c->variant == 0) && c->defer && full && sreg_supported(s)) {
      c->defer->kind = 0; c->defer->a = a; c->defer->grid = grid; c->defer->d = s.d; c->defer->m = s.m;
      return GVI_OK;
    }
    if (!done && (c->variant == 5 || c->variant == 0)) done = dispatch_sreg(s, a, grid, st, c->sreg_pipe && s.table->Zq.p);
    if (!done && !dispatch_reg(s, a, grid, st)) return fail(c, GVI_ERR_UNSUPPORTED, "dispatch_reg");
  } else {
    if (s.d > 32) return fail(c, GVI_ERR_UNSUPPORTED, "generic kernel supports d <= 32");
    const int d = s.d, m = s.m;
    const size_t lds = (size_t)(2 * GEN_BS * (d + 1) + GEN_BS + d * d + d + m * d + 2 * m) * 8;
    if (lds > 160 * 1024) return fail(c, GVI_ERR_UNSUPPORTED, "generic kernel LDS budget");
    GVICK(allow_lds(c, (const void*)moments_generic_kernel, 160 * 1024));
    hipLaunchKernelGGL(moments_generic_kernel, dim3(s.K, s.nchunk), dim3(GEN_BS), lds, st, a);
  }
  HIPCK(c, hipGetLastError());
  if (prof) {
    HIPCK(c, hipEventRecord(s.ev[which][1], st));
    s.ev_set[which] = true;
  }
  return GVI_OK;
}

gvi_status run_epilogue(gvi_ctx* c, FactorSet& s, int full, double* Ephi, double* cost, double* Vdmu,
                        double* Vddmu, double* Ex, double* Exx, hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  if (s.K == 0) return GVI_OK;
  EpiArgs e;
  e.f = s.dev(); e.partial = s.partial.d(); e.nchunk = s.nchunk; e.full = full;
  e.Ephi = Ephi; e.cost = cost; e.Vdmu = Vdmu; e.Vddmu = Vddmu; e.E_xmuphi = Ex; e.E_xxphi = Exx;
  const size_t lds = epilogue_lds_doubles(s.d) * 8;
  hipLaunchKernelGGL(epilogue_kernel, dim3(s.K), dim3(64), lds, st, e);
  HIPCK(c, hipGetLastError());
  return GVI_OK;
}

gvi_status ensure_set_buffers(gvi_ctx* c, FactorSet& s) {
  const size_t K = s.K, d = s.d;
  for (int i = 0; i < 2; ++i) {
    HIPCK(c, s.mu_k[i].ensure(K * d * 8));
    HIPCK(c, s.Sigma_k[i].ensure(K * d * d * 8));
  }
  HIPCK(c, s.Ephi.ensure(K * 8));
  HIPCK(c, s.cost.ensure(K * 8));
  HIPCK(c, s.Vdmu.ensure(K * d * 8));
  HIPCK(c, s.Vddmu.ensure(K * d * d * 8));
  return GVI_OK;
}

// workspace of one chain operation (kernels_chain.hpp); two operations are in flight together in the dual launches
struct ChainWs { double* ws; int* wsi; };
gvi_status ensure_chain_ws(gvi_ctx* c, ChainWs& w) {
  const int NP = chain_padded(c->n);
  if (!NP) return fail(c, GVI_ERR_UNSUPPORTED, "state_dim > 16");
  DevMem& Wb = c->chain_ws ? c->Wbuf2 : c->Wbuf;
  DevMem& Ib = c->chain_ws ? c->Ibuf2 : c->Ibuf;
  HIPCK(c, Wb.ensure(chain_ws_doubles(c->T, NP) * 8));
  HIPCK(c, Ib.ensure(chain_lp_entries(c->T) * sizeof(int)));
  w.ws = Wb.d();
  w.wsi = (int*)Ib.p;
  return GVI_OK;
}

ChainArgs make_chain_args(gvi_ctx* c, const ChainWs& w, const double* D, const double* U, const double* rhs, double scale, bool need_back,
                          double* SigD, double* SigU, double* x, double* hld, bool with_mix) {
  ChainArgs a{};
  a.T = c->T; a.n = c->n; a.need_back = need_back ? 1 : 0;
  a.D = D; a.U = U; a.rhs = rhs; a.rhs_scale = scale;
  a.ws = w.ws; a.wsi = w.wsi;
  a.SigD = SigD; a.SigU = SigU; a.x = x; a.hld = hld;
  a.pred = c->cur_pred; a.pred_val = c->cur_pred_val;
  // the mixed-in matrix and its output are [D | U] in one buffer each (ngd_trial_state): the kernels derive the U parts
  a.mixV = with_mix ? c->mix.VD : nullptr;
  a.mixOut = with_mix ? c->mix.outD : nullptr; a.mix_step = with_mix ? c->mix.step : 0.0;
  assert(!(with_mix && c->mix.VD) || (c->mix.VU == c->mix.VD + (size_t)c->T * nn_(c) && c->mix.outU == c->mix.outD + (size_t)c->T * nn_(c)));
  return a;
}

gvi_status run_chain(gvi_ctx* c, const ChainArgs& a0, const ChainArgs& a1, bool on0, bool on1, const AsmList* AL = nullptr) {
  hipStream_t st = c->chain_stream ? c->chain_stream : c->stream;
  StageScope scope(c, STAGE_CHAIN);
  ChainSync sy;
  if (c->chain_merge) {
    constexpr unsigned RING = 64;
    if (!c->chain_sync.p) {
      HIPCK(c, c->chain_sync.ensure(RING * 2 * sizeof(unsigned)));
      HIPCK(c, hipMemset(c->chain_sync.p, 0, RING * 2 * sizeof(unsigned)));       // (once per context)
    }
    if (++c->chain_seq == 0) ++c->chain_seq;       // 0 is the cleared state of a word
    sy.seq = c->chain_seq;
    sy.words = (unsigned*)c->chain_sync.p + 2 * (sy.seq % RING);
    sy.fault = c->chain_merge_fault ? 0x40000000u : 0u;
  }
  const hipError_t e = chain_launch(c->n, chain_plan(c->T, c->n), a0, a1, on0, on1, st, AL, sy);
  if (e == hipErrorInvalidValue) return fail(c, GVI_ERR_UNSUPPORTED, "chain kernels: block size / LDS budget");
  HIPCK(c, e);
  return GVI_OK;
}

// log-det (+ optionally marginals) of the chain (D, U) device arrays; carries the fused trial precision (ctx->mix) if set
gvi_status run_bt_factor(gvi_ctx* c, const double* D, const double* U, double* SigD, double* SigU, double* hld) {
  ChainWs w;
  GVICK(ensure_chain_ws(c, w));
  const ChainArgs a0 = make_chain_args(c, w, D, U, nullptr, 1.0, SigD != nullptr, SigD, SigU, nullptr, hld, true);
  return run_chain(c, a0, a0, true, false);
}

gvi_status run_bt_solve(gvi_ctx* c, const double* D, const double* U, const double* rhs, double scale, double* x) {
  ChainWs w;
  GVICK(ensure_chain_ws(c, w));
  const ChainArgs a1 = make_chain_args(c, w, D, U, rhs, scale, false, nullptr, nullptr, x, nullptr, false);
  return run_chain(c, a1, a1, false, true);
}

gvi_status run_scatter(gvi_ctx* c, FactorSet& s, const double* Vdmu, const double* Vddmu, double* g, double* D, double* U) {
  ScatterArgs a;
  a.T = c->T; a.n = c->n; a.d = s.d; a.K = s.K;
  a.start = s.dstart.i(); a.ptr = s.dptr.i(); a.idx = s.didx.i();
  a.Vdmu = Vdmu; a.Vddmu = Vddmu; a.g = g; a.D = D; a.U = U;
  const int64_t total = (int64_t)c->T * (c->n + 2 * c->n * c->n);
  hipLaunchKernelGGL(bt_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, a);
  HIPCK(c, hipGetLastError());
  return GVI_OK;
}

gvi_status run_gather(gvi_ctx* c, FactorSet& s, const double* mu, const double* SigD, const double* SigU,
                      double* mu_k, double* Sigma_k) {
  const int64_t total = (int64_t)s.K * (s.d + s.d * s.d);
  if (total == 0) return GVI_OK;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, s.K, s.d, c->n,
                     s.dstart.i(), mu, SigD, SigU, mu_k, Sigma_k);
  HIPCK(c, hipGetLastError());
  return GVI_OK;
}

gvi_status check_pass_args(gvi_ctx* c, FactorSet* s, const void* mu, const void* Sigma) {
  if (!c) return GVI_ERR_ARG;
  if (!s) return fail(c, GVI_ERR_ARG, "bad set id");
  if (!mu || !Sigma) return fail(c, GVI_ERR_ARG, "mu / Sigma is NULL");
  return GVI_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* gvi_version(void) { return "gvi_hip 0.1 (gfx950, fp64)"; }

const char* gvi_last_error(const gvi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_noctx_err.c_str(); }

gvi_status gvi_ctx_create(int device, int dtype, gvi_ctx** out) {
  if (!out) return fail(nullptr, GVI_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (dtype == GVI_F32) return fail(nullptr, GVI_ERR_UNSUPPORTED, "GVI_F32 (config 5) is not implemented yet");
  if (dtype != GVI_F64) return fail(nullptr, GVI_ERR_ARG, "unknown dtype");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, GVI_ERR_HIP, "no HIP device: this library has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, GVI_ERR_ARG, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, GVI_ERR_HIP, "hipSetDevice failed");
  std::unique_ptr<gvi_ctx> c(new gvi_ctx);
  c->device = device; c->dtype = dtype;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
    return fail(nullptr, GVI_ERR_HIP, "hipStreamCreate failed");
  if (const char* w = getenv("GVI_NO_SCOST")) c->no_scost = atoi(w) != 0;
  if (const char* w = getenv("GVI_ORBIT")) c->orbit = atoi(w) != 0;
  if (const char* w = getenv("GVI_FUSED")) c->fused = atoi(w) != 0;
  if (const char* w = getenv("GVI_ASM_ON_LOAD")) c->asm_on_load = atoi(w) != 0;
  if (const char* w = getenv("GVI_PIPELINE")) c->pipeline = atoi(w) != 0;
  if (const char* w = getenv("GVI_CHAIN_WAVE")) chain_wave_enabled() = atoi(w) != 0;     // (process-wide: A/B leg of kernels_chain_wave.hpp)
  if (const char* w = getenv("GVI_CHAIN_MERGE")) c->chain_merge = atoi(w) != 0;
  if (const char* w = getenv("GVI_SREG_PIPE")) c->sreg_pipe = atoi(w) != 0;
  if (const char* w = getenv("GVI_MIRROR")) c->mirror = atoi(w) != 0;
  if (const char* w = getenv("GVI_NO_PAIR")) c->pair_fuse = atoi(w) == 0;
  if (const char* w = getenv("GVI_NO_FUSE_GATHER")) c->fuse_gather = atoi(w) == 0;
  if (const char* w = getenv("GVI_SIDE_SOLVE")) c->side_solve = atoi(w) != 0;
  if (const char* w = getenv("GVI_FUSE_TRIAL")) c->fuse_trial = std::min(2, std::max(0, atoi(w)));
  if (const char* w = getenv("GVI_SPIN_MS")) c->spin_ms = std::max(0, atoi(w));
  if (const char* w = getenv("GVI_SAFE_PUBLISH")) c->safe_publish = atoi(w) != 0;
  if (hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void**)&c->host_slot, 128, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->host_slot_dev, c->host_slot, 0) != hipSuccess)
    return fail(nullptr, GVI_ERR_HIP, "event / host-mapped slot allocation failed");
  for (int q = 0; q < 16; ++q) c->host_slot[q] = 0.0;
  *out = c.release();
  return GVI_OK;
}

gvi_status gvi_ctx_destroy(gvi_ctx* ctx) {
  if (!ctx) return GVI_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->sets.clear();
  if (ctx->dist.comm && ctx->dist.ncclCommDestroy) (void)ctx->dist.ncclCommDestroy(ctx->dist.comm);
  if (ctx->fork) (void)hipEventDestroy(ctx->fork);
  for (auto& r : ctx->stage_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto& e : ctx->stage_pool) (void)hipEventDestroy(e);
  if (ctx->side) (void)hipStreamSynchronize(ctx->side);
  if (ctx->ev_grad) (void)hipEventDestroy(ctx->ev_grad);
  for (auto& e : ctx->ev_solve) if (e) (void)hipEventDestroy(e);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->dbg_mask) {                                          // the device-wide log pointer must not outlive its buffer
    double* lg = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gvi_dbg_log), &lg, sizeof(double*));
  }
  if (ctx->host_slot) (void)hipHostFree(ctx->host_slot);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GVI_OK;
}

gvi_status gvi_ctx_set_stream(gvi_ctx* ctx, void* hip_stream) {
  if (!ctx) return GVI_ERR_ARG;
  HIPCK(ctx, hipStreamSynchronize(ctx->stream));
  if (hip_stream) {
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else if (!ctx->own_stream) {
    HIPCK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  return GVI_OK;
}

gvi_status gvi_ctx_sync(gvi_ctx* ctx) { return ctx ? sync(ctx) : GVI_ERR_ARG; }

gvi_status gvi_spgh_count(int d, int p, int64_t* N) {
  if (!N) return fail(nullptr, GVI_ERR_ARG, "N is NULL");
  const int64_t n = spgh_count(d, p);
  if (n < 0) return fail(nullptr, GVI_ERR_NOTABLE, "(d, p) outside the tabulated rules (1 <= p <= 25, d <= 64)");
  *N = n;
  return GVI_OK;
}

gvi_status gvi_spgh_nodes(int d, int p, int64_t N, double* Z, double* w, int8_t* idx) {
  SparseGrid g;
  if (spgh_generate(d, p, g)) return fail(nullptr, GVI_ERR_NOTABLE, "(d, p) outside the tabulated rules");
  if (N != g.N) return fail(nullptr, GVI_ERR_ARG, "N does not match gvi_spgh_count(d, p)");
  if (Z) memcpy(Z, g.Z.data(), g.Z.size() * 8);
  if (w) memcpy(w, g.w.data(), g.w.size() * 8);
  if (idx) memcpy(idx, g.idx.data(), g.idx.size());
  return GVI_OK;
}

gvi_status gvi_table_file_list(const char* path, int64_t cap, int64_t* count, double* dims, double* degs, int64_t* rows) {
  if (!path || !count) return GVI_ERR_ARG;
  return table_file_list(path, cap, count, dims, degs, rows) ? fail(nullptr, GVI_ERR_ARG, "cannot read table file") : GVI_OK;
}

gvi_status gvi_table_file_read(const char* path, int d, int p, int64_t* N, double* Z, double* w) {
  if (!path || !N) return GVI_ERR_ARG;
  int64_t found = 0;
  const int rc = table_file_read(path, d, p, *N, Z, w, &found);
  if (rc == 2) return fail(nullptr, GVI_ERR_NOTABLE, "key (d, p) not in the table file");
  if (rc == 3) { *N = found; return fail(nullptr, GVI_ERR_ARG, "N does not match the entry"); }
  if (rc) return fail(nullptr, GVI_ERR_ARG, "cannot read table file");
  *N = found;
  return GVI_OK;
}

gvi_status gvi_table_file_write(const char* path, int n_entries, const int32_t* dims, const int32_t* degs) {
  if (!path || n_entries < 0 || (n_entries && (!dims || !degs))) return GVI_ERR_ARG;
  const int rc = table_file_write(path, n_entries, dims, degs);
  if (rc == 2) return fail(nullptr, GVI_ERR_NOTABLE, "(d, p) outside the tabulated rules");
  return rc ? fail(nullptr, GVI_ERR_ARG, "cannot write table file") : GVI_OK;
}

gvi_status gvi_chain_set(gvi_ctx* ctx, int T, int n) {
  if (!ctx) return GVI_ERR_ARG;
  if (T < 1 || n < 1) return fail(ctx, GVI_ERR_ARG, "T and n must be >= 1");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  ctx->T = T; ctx->n = n;
  ctx->sets.clear();
  ctx->ngd.ready = false;
  ctx->ngd.have_trial = false;
  ctx->dist.ranges_valid = false;
  return GVI_OK;
}

// gvi_factors_add / gvi_factors_add_table: tabZ != NULL takes the caller's table (private to the set) instead of the
// generated (d, p) one
static gvi_status factors_add_impl(gvi_ctx* ctx, int K, int d, int p, const int32_t* start, int psi_kind,
                                   const double* psi_params, int64_t params_per_factor, const double* temperature,
                                   int64_t tabN, const double* tabZ, const double* tabw, int* set_id) {
  if (!ctx) return GVI_ERR_ARG;
  if (ctx->T < 1) return fail(ctx, GVI_ERR_STATE, "call gvi_chain_set first");
  // K == 0 is a valid EMPTY set: the shard of a rank that received none of a small set's factors (set ids stay
  // aligned across ranks); every launch skips it
  if (K < 0 || (K > 0 && !start)) return fail(ctx, GVI_ERR_ARG, "K < 0 or start is NULL");
  const int n = ctx->n;
  if (d != n && d != 2 * n) return fail(ctx, GVI_ERR_ARG, "factor dimension must be n or 2n");
  for (int k = 0; k < K; ++k)
    if (start[k] < 0 || (int64_t)start[k] * n + d > (int64_t)ctx->T * n)
      return fail(ctx, GVI_ERR_ARG, "start index out of range");
  int m = 0;
  int64_t need = 0;
  switch (psi_kind) {
    case GVI_PSI_RANGE_1D: if (d != 1) return fail(ctx, GVI_ERR_ARG, "RANGE_1D needs d == 1"); need = 5; break;
    case GVI_PSI_QUAD_PRIOR: if (d % 2) return fail(ctx, GVI_ERR_ARG, "QUAD_PRIOR needs even d"); m = d / 2; need = 2 * (int64_t)m * m; break;
    case GVI_PSI_FIXED_PRIOR: m = d; need = d + (int64_t)d * d; break;
    case GVI_PSI_HOST_CALLBACK: need = 0; break;
    case GVI_PSI_HINGE_SDF_2D: if (d < 2) return fail(ctx, GVI_ERR_ARG, "HINGE_SDF_2D needs d >= 2"); need = 3; break;
    case GVI_PSI_HINGE_SDF_2D_BODY: if (d < 3) return fail(ctx, GVI_ERR_ARG, "HINGE_SDF_2D_BODY needs d >= 3"); need = 6; break;
    case GVI_PSI_HINGE_SDF_3D: if (d < 3) return fail(ctx, GVI_ERR_ARG, "HINGE_SDF_3D needs d >= 3"); need = 3; break;
    case GVI_PSI_HINGE_SDF_3D_ARM: need = 2; break;
    default: return fail(ctx, GVI_ERR_ARG, "unknown psi kind");
  }
  if (need > 0 && (!psi_params || params_per_factor < need))
    return fail(ctx, GVI_ERR_ARG, "psi_params missing or params_per_factor too small for this kind");
  HIPCK(ctx, hipSetDevice(ctx->device));

  std::unique_ptr<FactorSet> s(new FactorSet);
  s->jtol = ctx->jacobi_tol;
  s->use_chol = ctx->chol_sqrt;
  s->K = K; s->d = d; s->p = p; s->m = m; s->kind = psi_kind;
  if (K > 0) s->start.assign(start, start + K);
  // quadrature table: shared between sets with the same (d, p)
  if (tabZ) {
    auto t = std::make_shared<Table>();
    GVICK(upload_table(ctx, *t, d, ctx->trust_table_degree ? p : -1, tabN, tabZ, tabw));
    s->table = t;
  }
  for (auto& t : ctx->tables) if (!s->table && t->d == d && t->p == p) s->table = t;
  if (!s->table) {
    SparseGrid g;
    if (spgh_generate(d, p, g)) return fail(ctx, GVI_ERR_NOTABLE, "(d, p) outside the tabulated rules");
    auto t = std::make_shared<Table>();
    GVICK(upload_table(ctx, *t, d, p, g.N, g.Z.data(), g.w.data()));
    ctx->tables.push_back(t);
    s->table = t;
  }
  // psi operands
  std::vector<double> A((size_t)K * m * d, 0.0), b((size_t)K * m, 0.0), sg((size_t)K * m, 1.0);
  if (psi_kind == GVI_PSI_QUAD_PRIOR || psi_kind == GVI_PSI_FIXED_PRIOR) {
    std::vector<double> Q, lam, W;
    for (int k = 0; k < K; ++k) {
      const double* P = psi_params + (size_t)k * params_per_factor;
      const double* Qin = psi_kind == GVI_PSI_QUAD_PRIOR ? P + (size_t)m * m : P + d;
      Q.assign((size_t)m * m, 0.0);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Q[(size_t)i * m + j] = 0.5 * (Qin[i * m + j] + Qin[j * m + i]);
      jacobi_eigh(m, Q, lam, W);
      // psi = c * r^T Qinv r = sum_r sign(e_r) (sqrt(c |e_r|) w_r^T r)^2 ,  c = 1/2 (QUAD) or 1 (FIXED)
      const double cfac = psi_kind == GVI_PSI_QUAD_PRIOR ? 0.5 : 1.0;
      for (int r = 0; r < m; ++r) {
        const double sc = std::sqrt(cfac * std::fabs(lam[r]));
        sg[(size_t)k * m + r] = lam[r] < 0 ? -1.0 : 1.0;
        double* Arow = &A[((size_t)k * m + r) * d];
        if (psi_kind == GVI_PSI_QUAD_PRIOR) {      // r = Phi x1 - x2 = [Phi, -I] x
          for (int a = 0; a < m; ++a) {
            double acc = 0.0;
            for (int j = 0; j < m; ++j) acc += W[(size_t)j * m + r] * P[j * m + a];
            Arow[a] = sc * acc;
            Arow[m + a] = -sc * W[(size_t)a * m + r];
          }
        } else {                                   // r = x - mu0
          double off = 0.0;
          for (int a = 0; a < d; ++a) { Arow[a] = sc * W[(size_t)a * m + r]; off += Arow[a] * P[a]; }
          b[(size_t)k * m + r] = -off;
        }
      }
    }
  }
  auto up = [&](DevMem& dm, const void* src, size_t bytes) -> gvi_status {
    HIPCK(ctx, dm.ensure(bytes ? bytes : 8));
    if (bytes) HIPCK(ctx, hipMemcpy(dm.p, src, bytes, hipMemcpyHostToDevice));
    return GVI_OK;
  };
  GVICK(up(s->A, A.data(), A.size() * 8));
  GVICK(up(s->b, b.data(), b.size() * 8));
  GVICK(up(s->sgn, sg.data(), sg.size() * 8));
  s->all_pos = std::all_of(sg.begin(), sg.end(), [](double v) { return v == 1.0; });
  if (psi_kind == GVI_PSI_RANGE_1D || psi_kind >= GVI_PSI_HINGE_SDF_2D) {
    const int np = (int)need;
    std::vector<double> raw((size_t)K * np);
    for (int k = 0; k < K; ++k) memcpy(&raw[(size_t)k * np], psi_params + (size_t)k * params_per_factor, (size_t)np * 8);
    s->raw_stride = np;
    GVICK(up(s->raw, raw.data(), raw.size() * 8));
  } else {
    GVICK(up(s->raw, nullptr, 0));
  }
  std::vector<double> temp(K, 1.0);
  if (temperature) temp.assign(temperature, temperature + K);
  GVICK(up(s->temperature, temp.data(), temp.size() * 8));
  {
    std::vector<double> one((size_t)K, 1.0);
    GVICK(up(s->ones, one.data(), one.size() * 8));
  }
  // CSR over states for the ordered assemble
  std::vector<int32_t> ptr(ctx->T + 1, 0), idx(K);
  for (int k = 0; k < K; ++k) ptr[start[k] + 1]++;
  for (int t = 0; t < ctx->T; ++t) ptr[t + 1] += ptr[t];
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int k = 0; k < K; ++k) idx[fill[start[k]]++] = k;
  }
  s->chain_structured = (d == ctx->n && K <= ctx->T) || (d == 2 * ctx->n && K <= ctx->T - 1);
  for (int k = 0; k < K && s->chain_structured; ++k) s->chain_structured = s->start[k] == k;
  GVICK(up(s->dstart, s->start.data(), (size_t)K * 4));
  GVICK(up(s->dptr, ptr.data(), ptr.size() * 4));
  GVICK(up(s->didx, idx.data(), idx.size() * 4));
  HIPCK(ctx, s->S.ensure((size_t)K * d * d * 8));
  HIPCK(ctx, s->Sinv.ensure((size_t)K * d * d * 8));
  HIPCK(ctx, s->Lam.ensure((size_t)K * d * d * 8));
  HIPCK(ctx, s->H.ensure((size_t)K * std::max(m, 1) * d * 8));
  if (split_supported(*s)) {            // per-wave row blocks of H for the split kernel (rows >= m stay 0)
    const size_t bytes = (size_t)K * 4 * d * ((m + 3) / 4) * 8;
    HIPCK(ctx, s->Hq.ensure(bytes));
    HIPCK(ctx, hipMemset(s->Hq.p, 0, bytes));
  }
  HIPCK(ctx, s->u0.ensure((size_t)K * std::max(m, 1) * 8));
  GVICK(ensure_set_buffers(ctx, *s));
  HIPCK(ctx, hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
  HIPCK(ctx, hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
  ctx->sets.push_back(std::move(s));
  ctx->dist.ranges_valid = false;
  ctx->ngd.ready = false;
  if (set_id) *set_id = (int)ctx->sets.size() - 1;
  return GVI_OK;
}

gvi_status gvi_factors_add(gvi_ctx* ctx, int K, int d, int p, const int32_t* start, int psi_kind,
                           const double* psi_params, int64_t params_per_factor, const double* temperature,
                           int* set_id) {
  return factors_add_impl(ctx, K, d, p, start, psi_kind, psi_params, params_per_factor, temperature, 0, nullptr, nullptr, set_id);
}

gvi_status gvi_factors_add_table(gvi_ctx* ctx, int K, int d, int p, const int32_t* start, int psi_kind,
                                 const double* psi_params, int64_t params_per_factor, const double* temperature,
                                 int64_t N, const double* Z, const double* w, int* set_id) {
  if (!ctx) return GVI_ERR_ARG;
  if (N < 1 || !Z || !w) return fail(ctx, GVI_ERR_ARG, "bad table");
  return factors_add_impl(ctx, K, d, p, start, psi_kind, psi_params, params_per_factor, temperature, N, Z, w, set_id);
}

gvi_status gvi_factors_set_table(gvi_ctx* ctx, int set_id, int64_t N, const double* Z, const double* w) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (N < 1 || !Z || !w) return fail(ctx, GVI_ERR_ARG, "bad table");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  auto t = std::make_shared<Table>();
  GVICK(upload_table(ctx, *t, s->d, -1, N, Z, w));
  s->table = t;
  s->prep_slot = -1;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;   // cost / gradients of the old table are stale
  ctx->ngd.grad_valid = false;
  ctx->ngd.spec_ready = false;
  return GVI_OK;
}

gvi_status gvi_factors_set_sdf2d(gvi_ctx* ctx, int set_id, double origin_x, double origin_y, double cell_size, int rows,
                                 int cols, const double* data) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (s->kind != KIND_HINGE_SDF_2D && s->kind != KIND_HINGE_SDF_2D_BODY)
    return fail(ctx, GVI_ERR_ARG, "set is not GVI_PSI_HINGE_SDF_2D / _2D_BODY");
  if (rows < 2 || cols < 2 || !(cell_size > 0) || !data) return fail(ctx, GVI_ERR_ARG, "bad grid");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  HIPCK(ctx, s->sdf.ensure((size_t)rows * cols * 8));
  HIPCK(ctx, hipMemcpy(s->sdf.p, data, (size_t)rows * cols * 8, hipMemcpyHostToDevice));
  s->sdf_rows = rows; s->sdf_cols = cols; s->sdf_ox = origin_x; s->sdf_oy = origin_y; s->sdf_cell = cell_size;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_factors_set_sdf3d(gvi_ctx* ctx, int set_id, const double* origin, double cell_size, int rows, int cols,
                                 int nz, const double* data) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (s->kind != KIND_HINGE_SDF_3D && s->kind != KIND_HINGE_SDF_3D_ARM)
    return fail(ctx, GVI_ERR_ARG, "set is not GVI_PSI_HINGE_SDF_3D / _3D_ARM");
  if (rows < 2 || cols < 2 || nz < 2 || !(cell_size > 0) || !data || !origin) return fail(ctx, GVI_ERR_ARG, "bad grid");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  const size_t bytes = (size_t)rows * cols * nz * 8;
  HIPCK(ctx, s->sdf.ensure(bytes));
  HIPCK(ctx, hipMemcpy(s->sdf.p, data, bytes, hipMemcpyHostToDevice));
  s->sdf_rows = rows; s->sdf_cols = cols; s->sdf_nz = nz;
  s->sdf_ox = origin[0]; s->sdf_oy = origin[1]; s->sdf_oz = origin[2]; s->sdf_cell = cell_size;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_factors_set_arm(gvi_ctx* ctx, int set_id, int ndof, const double* a, const double* alpha, const double* d,
                               const double* theta_bias, int nspheres, const int32_t* frames, const double* centers,
                               const double* radii) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (s->kind != KIND_HINGE_SDF_3D_ARM) return fail(ctx, GVI_ERR_ARG, "set is not GVI_PSI_HINGE_SDF_3D_ARM");
  if (ndof < 1 || ndof > s->d || nspheres < s->d || !a || !alpha || !d || !theta_bias || !frames || !centers || !radii)
    return fail(ctx, GVI_ERR_ARG, "bad arm model (need 1 <= ndof <= factor dimension <= nspheres: n_balls = factor dimension)");
  for (int q = 0; q < nspheres; ++q)
    if (frames[q] < 0 || frames[q] >= ndof || (q && frames[q] < frames[q - 1]))
      return fail(ctx, GVI_ERR_ARG, "sphere frames must be non-decreasing and < ndof");
  std::vector<double> pk;
  s->arm_ndof = ndof;
  pk.push_back(ndof); pk.push_back(nspheres);
  pk.insert(pk.end(), a, a + ndof); pk.insert(pk.end(), alpha, alpha + ndof);
  pk.insert(pk.end(), d, d + ndof); pk.insert(pk.end(), theta_bias, theta_bias + ndof);
  for (int q = 0; q < nspheres; ++q) pk.push_back(frames[q]);
  pk.insert(pk.end(), centers, centers + 3 * (size_t)nspheres);
  pk.insert(pk.end(), radii, radii + nspheres);
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  HIPCK(ctx, s->arm.ensure(pk.size() * 8));
  HIPCK(ctx, hipMemcpy(s->arm.p, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_factors_set_closed_form(gvi_ctx* ctx, int set_id, int on) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (on && s->kind != KIND_QUAD_PRIOR && s->kind != KIND_FIXED_PRIOR)
    return fail(ctx, GVI_ERR_ARG, "closed form exists for QUAD_PRIOR / FIXED_PRIOR sets only");
  GVICK(sync(ctx));
  s->closed_form = on != 0;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_factors_set_temperature(gvi_ctx* ctx, int set_id, const double* temperature) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s || !temperature) return fail(ctx, GVI_ERR_ARG, "bad set id / NULL temperature");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  HIPCK(ctx, hipMemcpy(s->temperature.p, temperature, (size_t)s->K * 8, hipMemcpyHostToDevice));
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_factors_info(const gvi_ctx* ctx, int set_id, int* K, int* d, int* p, int64_t* N) {
  FactorSet* s = get_set(const_cast<gvi_ctx*>(ctx), set_id);
  if (!s) return GVI_ERR_ARG;
  if (K) *K = s->K;
  if (d) *d = s->d;
  if (p) *p = s->p;
  if (N) *N = s->table->N;
  return GVI_OK;
}

// ---- per-pass factor operators ----
gvi_status gvi_moments_dev(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* Ephi,
                           double* Vdmu, double* Vddmu) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (s->kind == KIND_HOST_CALLBACK) return fail(ctx, GVI_ERR_ARG, "HOST_CALLBACK set: use gvi_expand + gvi_moments_from_psi");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(run_prep(ctx, *s, mu, Sigma));
  GVICK(run_moments(ctx, *s, mu, nullptr, 1));
  return run_epilogue(ctx, *s, 1, Ephi, nullptr, Vdmu, Vddmu, nullptr, nullptr);
}

gvi_status gvi_costs_dev(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* cost) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (s->kind == KIND_HOST_CALLBACK) return fail(ctx, GVI_ERR_ARG, "HOST_CALLBACK set has no device psi");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(run_prep(ctx, *s, mu, Sigma));
  GVICK(run_moments(ctx, *s, mu, nullptr, 0));
  return run_epilogue(ctx, *s, 0, nullptr, cost, nullptr, nullptr, nullptr, nullptr);
}

static gvi_status upload_pass_inputs(gvi_ctx* ctx, FactorSet* s, const double* mu, const double* Sigma) {
  HIPCK(ctx, s->in_mu.ensure((size_t)s->K * s->d * 8));
  HIPCK(ctx, s->in_Sigma.ensure((size_t)s->K * s->d * s->d * 8));
  GVICK(h2d(ctx, s->in_mu.p, mu, (size_t)s->K * s->d * 8));
  return h2d(ctx, s->in_Sigma.p, Sigma, (size_t)s->K * s->d * s->d * 8);
}

gvi_status gvi_moments(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* Ephi,
                       double* Vdmu, double* Vddmu) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(upload_pass_inputs(ctx, s, mu, Sigma));
  GVICK(gvi_moments_dev(ctx, set_id, s->in_mu.d(), s->in_Sigma.d(), s->Ephi.d(), s->Vdmu.d(), s->Vddmu.d()));
  if (Ephi) GVICK(d2h(ctx, Ephi, s->Ephi.p, (size_t)s->K * 8));
  if (Vdmu) GVICK(d2h(ctx, Vdmu, s->Vdmu.p, (size_t)s->K * s->d * 8));
  if (Vddmu) GVICK(d2h(ctx, Vddmu, s->Vddmu.p, (size_t)s->K * s->d * s->d * 8));
  return sync(ctx);
}

gvi_status gvi_raw_moments(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* E_phi,
                           double* E_xmuphi, double* E_xxphi) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (s->kind == KIND_HOST_CALLBACK) return fail(ctx, GVI_ERR_ARG, "HOST_CALLBACK set has no device psi");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(upload_pass_inputs(ctx, s, mu, Sigma));
  HIPCK(ctx, s->raw1.ensure((size_t)s->K * s->d * 8));
  HIPCK(ctx, s->raw2.ensure((size_t)s->K * s->d * s->d * 8));
  GVICK(run_prep(ctx, *s, s->in_mu.d(), s->in_Sigma.d()));
  GVICK(run_moments(ctx, *s, s->in_mu.d(), nullptr, 1));
  GVICK(run_epilogue(ctx, *s, 1, s->Ephi.d(), nullptr, nullptr, nullptr, s->raw1.d(), s->raw2.d()));
  if (E_phi) GVICK(d2h(ctx, E_phi, s->Ephi.p, (size_t)s->K * 8));
  if (E_xmuphi) GVICK(d2h(ctx, E_xmuphi, s->raw1.p, (size_t)s->K * s->d * 8));
  if (E_xxphi) GVICK(d2h(ctx, E_xxphi, s->raw2.p, (size_t)s->K * s->d * s->d * 8));
  return sync(ctx);
}

gvi_status gvi_costs(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* cost) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (!cost) return fail(ctx, GVI_ERR_ARG, "cost is NULL");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(upload_pass_inputs(ctx, s, mu, Sigma));
  GVICK(gvi_costs_dev(ctx, set_id, s->in_mu.d(), s->in_Sigma.d(), s->cost.d()));
  GVICK(d2h(ctx, cost, s->cost.p, (size_t)s->K * 8));
  return sync(ctx);
}

gvi_status gvi_expand(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* X) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (!X) return fail(ctx, GVI_ERR_ARG, "X is NULL");
  if (s->K == 0) return GVI_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(upload_pass_inputs(ctx, s, mu, Sigma));
  const size_t bytes = (size_t)s->K * s->d * s->table->N * 8;
  HIPCK(ctx, s->X.ensure(bytes));
  s->force_sym = true;                                       // the caller sees the nodes: the reference's symmetric root
  const gvi_status pst = run_prep(ctx, *s, s->in_mu.d(), s->in_Sigma.d());
  s->force_sym = false;
  GVICK(pst);
  hipLaunchKernelGGL(expand_kernel, dim3((unsigned)((s->table->N + 255) / 256), s->K), dim3(256), 0, ctx->stream,
                     s->dev(), s->in_mu.d(), s->X.d());
  HIPCK(ctx, hipGetLastError());
  GVICK(d2h(ctx, X, s->X.p, bytes));
  return sync(ctx);
}

gvi_status gvi_moments_from_psi(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, const double* psi,
                                double* Ephi, double* Vdmu, double* Vddmu) {
  FactorSet* s = get_set(ctx, set_id);
  GVICK(check_pass_args(ctx, s, mu, Sigma));
  if (!psi) return fail(ctx, GVI_ERR_ARG, "psi is NULL");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(upload_pass_inputs(ctx, s, mu, Sigma));
  const size_t bytes = (size_t)s->K * s->table->N * 8;
  HIPCK(ctx, s->psi_ext.ensure(bytes));
  GVICK(h2d(ctx, s->psi_ext.p, psi, bytes));
  s->force_sym = true;                                       // psi was evaluated at the reference's nodes (gvi_expand)
  gvi_status pst = run_prep(ctx, *s, s->in_mu.d(), s->in_Sigma.d());
  if (pst == GVI_OK) pst = run_moments(ctx, *s, s->in_mu.d(), s->psi_ext.d(), 1);
  if (pst == GVI_OK) pst = run_epilogue(ctx, *s, 1, s->Ephi.d(), nullptr, s->Vdmu.d(), s->Vddmu.d(), nullptr, nullptr);
  s->force_sym = false;
  GVICK(pst);
  if (Ephi) GVICK(d2h(ctx, Ephi, s->Ephi.p, (size_t)s->K * 8));
  if (Vdmu) GVICK(d2h(ctx, Vdmu, s->Vdmu.p, (size_t)s->K * s->d * 8));
  if (Vddmu) GVICK(d2h(ctx, Vddmu, s->Vddmu.p, (size_t)s->K * s->d * s->d * 8));
  return sync(ctx);
}

// ---- joint operators (host-pointer forms stage through the scratch buffer) ----
gvi_status gvi_bt_assemble(gvi_ctx* ctx, int nsets, const int* set_ids, const double* const* Vdmu,
                           const double* const* Vddmu, double* g, double* D, double* U) {
  if (!ctx) return GVI_ERR_ARG;
  if (nsets < 1 || !set_ids || !Vdmu || !Vddmu || !g || !D || !U) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t T = ctx->T, n = ctx->n, nn = n * n, tot = T * n + bt_count(ctx);
  HIPCK(ctx, ctx->scratch.ensure(tot * 8));
  double* dg = ctx->scratch.d();
  double* dD = dg + T * n;
  double* dU = dD + T * nn;
  HIPCK(ctx, hipMemsetAsync(dg, 0, tot * 8, ctx->stream));
  for (int i = 0; i < nsets; ++i) {
    FactorSet* s = get_set(ctx, set_ids[i]);
    if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
    GVICK(h2d(ctx, s->Vdmu.p, Vdmu[i], (size_t)s->K * s->d * 8));
    GVICK(h2d(ctx, s->Vddmu.p, Vddmu[i], (size_t)s->K * s->d * s->d * 8));
    GVICK(run_scatter(ctx, *s, s->Vdmu.d(), s->Vddmu.d(), dg, dD, dU));
  }
  GVICK(d2h(ctx, g, dg, T * n * 8));
  GVICK(d2h(ctx, D, dD, T * nn * 8));
  if (T > 1) GVICK(d2h(ctx, U, dU, (T - 1) * nn * 8));
  return sync(ctx);
}

static gvi_status stage_chain(gvi_ctx* ctx, const double* D, const double* U, size_t extra, double** dD, double** dU,
                              double** dextra) {
  const size_t T = ctx->T, nn = nn_(ctx);
  HIPCK(ctx, ctx->scratch.ensure((bt_count(ctx) + extra) * 8));
  *dD = ctx->scratch.d();
  *dU = *dD + T * nn;
  *dextra = *dU + (T - 1) * nn;
  GVICK(h2d(ctx, *dD, D, T * nn * 8));
  if (T > 1) GVICK(h2d(ctx, *dU, U, (T - 1) * nn * 8));
  return GVI_OK;
}

gvi_status gvi_bt_solve(gvi_ctx* ctx, const double* D, const double* U, const double* rhs, double* x) {
  if (!ctx) return GVI_ERR_ARG;
  if (!D || (!U && ctx->T > 1) || !rhs || !x) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t Tn = (size_t)ctx->T * ctx->n;
  double *dD, *dU, *ex;
  GVICK(stage_chain(ctx, D, U, 2 * Tn, &dD, &dU, &ex));
  GVICK(h2d(ctx, ex, rhs, Tn * 8));
  GVICK(run_bt_solve(ctx, dD, dU, ex, 1.0, ex + Tn));
  GVICK(d2h(ctx, x, ex + Tn, Tn * 8));
  return sync(ctx);
}

gvi_status gvi_bt_logdet(gvi_ctx* ctx, const double* D, const double* U, double* half_logdet) {
  if (!ctx) return GVI_ERR_ARG;
  if (!D || (!U && ctx->T > 1) || !half_logdet) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  double *dD, *dU, *ex;
  GVICK(stage_chain(ctx, D, U, 1, &dD, &dU, &ex));
  GVICK(run_bt_factor(ctx, dD, dU, nullptr, nullptr, ex));
  GVICK(d2h(ctx, half_logdet, ex, 8));
  return sync(ctx);
}

gvi_status gvi_bt_marginals(gvi_ctx* ctx, const double* D, const double* U, double* SigD, double* SigU) {
  if (!ctx) return GVI_ERR_ARG;
  if (!D || (!U && ctx->T > 1) || !SigD || (!SigU && ctx->T > 1)) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t T = ctx->T, nn = nn_(ctx);
  double *dD, *dU, *ex;
  GVICK(stage_chain(ctx, D, U, bt_count(ctx) + 1, &dD, &dU, &ex));
  double* sD = ex;
  double* sU = sD + T * nn;
  GVICK(run_bt_factor(ctx, dD, dU, sD, sU, sU + (T - 1) * nn));
  GVICK(d2h(ctx, SigD, sD, T * nn * 8));
  if (T > 1) GVICK(d2h(ctx, SigU, sU, (T - 1) * nn * 8));
  return sync(ctx);
}

gvi_status gvi_gather_marginals(gvi_ctx* ctx, int set_id, const double* mu, const double* SigD, const double* SigU,
                                double* mu_k, double* Sigma_k) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return fail(ctx, GVI_ERR_ARG, "bad set id");
  if (!mu || !SigD || (!SigU && ctx->T > 1) || !mu_k || !Sigma_k) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t T = ctx->T, n = ctx->n, nn = n * n;
  HIPCK(ctx, ctx->scratch.ensure((T * n + bt_count(ctx)) * 8));
  double* dmu = ctx->scratch.d();
  double* sD = dmu + T * n;
  double* sU = sD + T * nn;
  GVICK(h2d(ctx, dmu, mu, T * n * 8));
  GVICK(h2d(ctx, sD, SigD, T * nn * 8));
  if (T > 1) GVICK(h2d(ctx, sU, SigU, (T - 1) * nn * 8));
  HIPCK(ctx, s->in_mu.ensure((size_t)s->K * s->d * 8));
  HIPCK(ctx, s->in_Sigma.ensure((size_t)s->K * s->d * s->d * 8));
  GVICK(run_gather(ctx, *s, dmu, sD, sU, s->in_mu.d(), s->in_Sigma.d()));
  GVICK(d2h(ctx, mu_k, s->in_mu.p, (size_t)s->K * s->d * 8));
  GVICK(d2h(ctx, Sigma_k, s->in_Sigma.p, (size_t)s->K * s->d * s->d * 8));
  return sync(ctx);
}

// ---- device-resident NGD iteration ----
static gvi_status ngd_check(gvi_ctx* ctx) {
  if (!ctx) return GVI_ERR_ARG;
  if (!ctx->ngd.ready) return fail(ctx, GVI_ERR_STATE, "call gvi_ngd_init first");
  return GVI_OK;
}

static SetList make_set_list(gvi_ctx* ctx, int slot) {
  SetList L;
  L.nsets = (int)ctx->sets.size();
  for (int i = 0; i < L.nsets; ++i) {
    FactorSet& s = *ctx->sets[i];
    SetDesc& d = L.s[i];
    d.K = s.K; d.d = s.d;
    d.start = s.dstart.i(); d.ptr = s.dptr.i(); d.idx = s.didx.i();
    d.Vdmu = s.Vdmu.d(); d.Vddmu = s.Vddmu.d();
    d.mu_k = s.mu_k[slot].d(); d.Sigma_k = s.Sigma_k[slot].d();
    d.cost = s.cost.d();
  }
  return L;
}

// prep of every set whose per-pass products are not those of NGD slot i -- one launch
static gvi_status ngd_refresh_gather(gvi_ctx* ctx, int i, const double* mu_from, const double* dmu, double step);

// stand-alone gather of a slot whose gather was left to the next prep (consumers other than prep call this)
static gvi_status ngd_flush_gather(gvi_ctx* ctx, int i) {
  NgdState::GatherPending& gp = ctx->ngd.gpend[i];
  if (!gp.on) return GVI_OK;
  const NgdState::GatherPending p = gp;
  gp.on = false;
  const bool keep = ctx->fuse_gather;
  ctx->fuse_gather = false;
  const gvi_status st = ngd_refresh_gather(ctx, i, p.mu_from, p.dmu, p.step);
  ctx->fuse_gather = keep;
  return st;
}

static gvi_status ngd_prep_all(gvi_ctx* ctx, int i) {
  NgdState& g = ctx->ngd;
  PrepList L;
  L.nsets = 0;
  L.koff[0] = 0;
  L.gather = 0;
  L.pred = ctx->cur_pred; L.pred_val = ctx->cur_pred_val;
  int dmax = 0;
  bool all = true;
  for (auto& s : ctx->sets) {
    if (s->kind == KIND_HOST_CALLBACK) return fail(ctx, GVI_ERR_UNSUPPORTED, "resident NGD needs device psi kinds");
    if (s->prep_slot == i) { all = false; break; }
  }
  if (g.gpend[i].on && !all) GVICK(ngd_flush_gather(ctx, i));      // cannot happen (a pending gather resets every prep_slot)
  for (auto& s : ctx->sets) {
    if (s->prep_slot == i) continue;
    s->prep_slot = i;
    L.f[L.nsets] = s->dev();
    if (ctx->warm_start) {                 // warm-started Jacobi; a cold start every 32nd prep bounds the drift
      if (s->Vws.bytes == 0) { HIPCK(ctx, s->Vws.ensure((size_t)s->K * s->d * s->d * 8)); s->warm_count = 0; }
      L.f[L.nsets].Vws = s->Vws.d();
      L.f[L.nsets].warm = (s->warm_count % 32) != 0;
      s->warm_count++;
    }
    L.mu[L.nsets] = s->mu_k[i].d();
    L.Sigma[L.nsets] = s->Sigma_k[i].d();
    L.start[L.nsets] = (const int32_t*)s->dstart.p;
    L.mu_k[L.nsets] = s->mu_k[i].d();
    L.Sigma_k[L.nsets] = s->Sigma_k[i].d();
    L.koff[L.nsets + 1] = L.koff[L.nsets] + s->K;
    dmax = std::max(dmax, s->d);
    ++L.nsets;
  }
  if (L.nsets == 0) return GVI_OK;
  int extra = 0;
  if (g.gpend[i].on) {
    const size_t T = ctx->T, nn = nn_(ctx);
    L.gather = 1; L.n = ctx->n;
    L.gmu = g.gpend[i].dmu ? g.gpend[i].mu_from : g.mu[i].d();
    L.gdmu = g.gpend[i].dmu; L.gstep = g.gpend[i].step;
    L.SigD = g.Sig[i].d(); L.SigU = g.Sig[i].d() + T * nn;
    L.mu_out = g.mu[i].d(); L.nmu = (int64_t)T * ctx->n;
    if (L.gdmu) extra = (int)((L.nmu + 63) / 64);
    g.gpend[i].on = false;
  }
  const int dp = dmax + (dmax & 1);
  const size_t lds = (size_t)(4 * dmax * dmax + 2 * dp + 3 * dmax + (dp + 1) / 2 + 1 + dmax * dmax + dmax) * 8 + 16;
  const dim3 grid(L.koff[L.nsets] + extra);
  if (grid.x == 0) return GVI_OK;                                     // only empty shards
  if (dmax <= 8) hipLaunchKernelGGL(prep_all_kernel<1>, grid, dim3(64), lds, ctx->stream, L);
  else if (dmax <= 16) hipLaunchKernelGGL(prep_all_kernel<4>, grid, dim3(64), lds, ctx->stream, L);
  else if (dmax <= 32) hipLaunchKernelGGL(prep_all_kernel<16>, grid, dim3(64), lds, ctx->stream, L);
  else return fail(ctx, GVI_ERR_UNSUPPORTED, "factor dimension > 32");
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

// epilogue of every set -- one launch (full: Vdmu / Vddmu / Ephi / cost; else cost only)
static EpiList make_epi_list(gvi_ctx* ctx, int full, int* dmax_out) {
  EpiList L;
  L.nsets = (int)ctx->sets.size();
  L.koff[0] = 0;
  int dmax = 0;
  for (int si = 0; si < L.nsets; ++si) {
    FactorSet& s = *ctx->sets[si];
    EpiArgs& e = L.e[si];
    e.f = s.dev(); e.partial = s.partial.d(); e.nchunk = s.nchunk; e.full = full;
    e.Ephi = full ? s.Ephi.d() : nullptr; e.cost = s.cost.d();
    e.Vdmu = full ? s.Vdmu.d() : nullptr; e.Vddmu = full ? s.Vddmu.d() : nullptr;
    e.E_xmuphi = nullptr; e.E_xxphi = nullptr;
    L.koff[si + 1] = L.koff[si] + s.K;
    dmax = std::max(dmax, s.d);
  }
  if (dmax_out) *dmax_out = dmax;
  return L;
}

// publish_slot >= 0: the launch also sums the factor costs and publishes the cost of NGD slot `publish_slot` (EpiTail)
static gvi_status ngd_epilogue_all(gvi_ctx* ctx, int full, int publish_slot = -1) {
  int dmax = 0;
  const EpiList L = make_epi_list(ctx, full, &dmax);
  const size_t lds = (epilogue_lds_doubles(dmax) + 256) * 8;       // + the tail's 256-leaf tree
  if (L.koff[L.nsets] == 0) return GVI_OK;
  EpiTail tail;
  tail.on = 0; tail.acc = nullptr; tail.half_logdet = nullptr; tail.host_out = nullptr; tail.seq = 0.0; tail.counter = nullptr;
  tail.pred = ctx->cur_pred; tail.pred_val = ctx->cur_pred_val;
  tail.accept = nullptr; tail.cost_dev = nullptr; tail.slot_cur = tail.slot_trial = 0; tail.c0_use_imm = 1; tail.c0_imm = 0.0;
  tail.safe = ctx->safe_publish ? 1 : 0;
  if (publish_slot >= 0) {
    const size_t need = (size_t)128 * (2 + (size_t)L.koff[L.nsets] / EPI_GROUP);
    if (ctx->epi_counter.bytes < need) {
      HIPCK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCK(ctx, ctx->epi_counter.ensure(need));
      HIPCK(ctx, hipMemsetAsync(ctx->epi_counter.p, 0, need, ctx->stream));
    }
    ctx->seq += 1.0;
    tail.on = 1; tail.acc = ctx->ngd.exch1.d(); tail.half_logdet = ctx->ngd.hld[publish_slot].d();
    tail.host_out = pub_slot(ctx); tail.seq = ctx->seq; tail.counter = (unsigned*)ctx->epi_counter.p;
    if (ctx->pipe_tail) {
      tail.accept = ctx->pipe_dev.d() + ctx->pub_ring; tail.cost_dev = ctx->pipe_dev.d() + 2;   // accept word of THIS ring slot
      tail.slot_cur = 1 - publish_slot; tail.slot_trial = publish_slot;
      tail.c0_use_imm = ctx->pipe_c0_imm ? 1 : 0; tail.c0_imm = ctx->pipe_c0;
    }
  }
  CostList cl{};
  cl.nsets = L.nsets;
  for (int si = 0; si < L.nsets; ++si) { cl.cost[si] = L.e[si].cost; cl.K[si] = L.e[si].f.K; }
  hipLaunchKernelGGL(epilogue_all_kernel, dim3(L.koff[L.nsets]), dim3(64), lds, ctx->stream, L, tail, cl);
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

// marginals + log-det of Lam[i], then gather every set's (mu_k, Sigma_k) into slot i (one launch)
static gvi_status ngd_refresh_factor(gvi_ctx* ctx, int i) {
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, nn = nn_(ctx);
  double* D = g.Lam[i].d();
  double* U = D + T * nn;
  double* sD = g.Sig[i].d();
  double* sU = sD + T * nn;
  return run_bt_factor(ctx, D, U, sD, sU, g.hld[i].d());
}

// mu_from / dmu / step: form mu[i] = mu_from + step dmu inside the gather launch (trial state); null: mu[i] is current
static gvi_status ngd_refresh_gather(gvi_ctx* ctx, int i, const double* mu_from, const double* dmu, double step) {
  NgdState& g = ctx->ngd;
  if (ctx->fuse_gather && ctx->defer_gather && !ctx->sets.empty()) {   // trial state: the prep launch that follows gathers
    for (auto& s : ctx->sets) if (s->prep_slot == i) s->prep_slot = -1;
    g.gpend[i].on = true; g.gpend[i].mu_from = mu_from; g.gpend[i].dmu = dmu; g.gpend[i].step = step;
    return GVI_OK;
  }
  g.gpend[i].on = false;
  const size_t T = ctx->T, nn = nn_(ctx);
  const int64_t nmu = (int64_t)T * ctx->n;
  double* sD = g.Sig[i].d();
  double* sU = sD + T * nn;
  if (ctx->sets.empty()) {
    if (dmu) hipLaunchKernelGGL(trial_kernel, dim3((unsigned)((nmu + 255) / 256)), dim3(256), 0, ctx->stream, nmu, (int64_t)0, step,
                                mu_from, dmu, (const double*)nullptr, (const double*)nullptr, g.mu[i].d(), (double*)nullptr);
    return GVI_OK;
  }
  if ((int)ctx->sets.size() > MAX_SETS) return fail(ctx, GVI_ERR_UNSUPPORTED, "more than 8 factor sets");
  int64_t maxwork = dmu ? nmu : 0;
  for (auto& s : ctx->sets) {
    if (s->prep_slot == i) s->prep_slot = -1;
    maxwork = std::max<int64_t>(maxwork, (int64_t)s->K * (s->d + s->d * s->d));
  }
  hipLaunchKernelGGL(gather_all_kernel, dim3((unsigned)std::max<int64_t>(1, (maxwork + 255) / 256), (unsigned)ctx->sets.size() + (dmu ? 1u : 0u)),
                     dim3(256), 0, ctx->stream, make_set_list(ctx, i), ctx->n, dmu ? mu_from : g.mu[i].d(), sD, sU, dmu, step,
                     g.mu[i].d(), nmu);
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

static gvi_status ngd_refresh(gvi_ctx* ctx, int i) {
  GVICK(ngd_refresh_factor(ctx, i));
  return ngd_refresh_gather(ctx, i, nullptr, nullptr, 0.0);
}

// sum over sets of sum_k E[psi]/T_k at slot i -> exch1[0].  Everything stays on ONE stream: side streams
// share the hardware queue on this part and every cross-stream dependency costs a 6-30 us barrier
// packet (profiles/r01_d_*); the small sets ride along inside the fused prep / epilogue launches.
static gvi_status ngd_moments_launch(gvi_ctx* ctx, int slot, int full);

// publish = true (single-process iteration): the same launch also writes {cost, half log-det, sequence} to the
// host-mapped slot, so no separate publish_kernel follows
static gvi_status ngd_cost_local(gvi_ctx* ctx, int i, bool publish = false) {
  NgdState& g = ctx->ngd;
  if (ctx->sets.empty()) {
    HIPCK(ctx, hipMemsetAsync(g.exch1.p, 0, 8, ctx->stream));
    if (publish) {
      ctx->seq += 1.0;
      hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, ctx->stream, g.exch1.d(), g.hld[i].d(), pub_slot(ctx), ctx->seq);
      HIPCK(ctx, hipGetLastError());
    }
    return GVI_OK;
  }
  if ((int)ctx->sets.size() > MAX_SETS) return fail(ctx, GVI_ERR_UNSUPPORTED, "more than 8 factor sets");
  GVICK(ngd_prep_all(ctx, i));
  GVICK(ngd_moments_launch(ctx, i, 0));
  if (publish) ctx->seq += 1.0;
  if (!ctx->tail_counter.p) {
    HIPCK(ctx, ctx->tail_counter.ensure(8));
    HIPCK(ctx, hipMemsetAsync(ctx->tail_counter.p, 0, 8, ctx->stream));
  }
  int64_t nfac = 0;
  for (auto& s : ctx->sets) nfac = std::max<int64_t>(nfac, s->K);
  const unsigned nblk = (unsigned)std::min<int64_t>(32, std::max<int64_t>(1, (nfac + 255) / 256));
  hipLaunchKernelGGL(cost_tail_kernel, dim3(nblk), dim3(256), 0, ctx->stream, make_epi_list(ctx, 0, nullptr), g.exch1.d(),
                     g.hld[i].d(), publish ? pub_slot(ctx) : nullptr, ctx->seq, (unsigned*)ctx->tail_counter.p);
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

static gvi_status ngd_cost_publish(gvi_ctx* ctx, int i) {
  NgdState& g = ctx->ngd;
  ctx->seq += 1.0;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, ctx->stream, g.exch1.d(), g.hld[i].d(),
                     pub_slot(ctx), ctx->seq);
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
}

static gvi_status ngd_cost_wait(gvi_ctx* ctx, int i, double* out, double seq_expected = -1.0, int ring = -1) {
  NgdState& g = ctx->ngd;
  const double want = seq_expected >= 0.0 ? seq_expected : ctx->seq;
  const int rg = ring >= 0 ? ring : ctx->pub_ring;
  // Spin on the host-mapped sequence word: a blocking stream sync costs tens of us of wake-up latency and would
  // also wait for the speculative work queued behind the publish.  The spin is bounded by WALL TIME (2 ms -- an
  // iteration is < 1 ms); past that the wait sleeps between polls (below).
  // {value, sequence} is ONE 16-byte device store (publish_to_host) and is read here with ONE 16-byte load (movdqa: a single
  // access on every AVX-capable x86), so value and sequence always belong to the same publish -- no "new sequence, old value"
  // window between two 8-byte loads
  const double* slot = ctx->host_slot + 4 * rg;
  typedef double v2d __attribute__((vector_size(16), aligned(16)));
  const bool safe = ctx->safe_publish;
  auto poll = [&](double* val) {
    if (safe) {
      // checked form [sequence | value | value | sequence]: taken only when both sequence words and both copies agree
      const volatile double* vs = slot;
      const double s0 = vs[0];
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      const double v1 = vs[1], v2 = vs[2];
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      const double s3 = vs[3];
      *val = v1;
      return s0 == want && s3 == want && std::memcmp(&v1, &v2, 8) == 0;
    }
    const v2d both = *(const volatile v2d*)(const void*)slot;        // (volatile: re-read at every poll)
    *val = both[0];
    return both[1] == want;
  };
  bool seen = false;
  double v = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {
    if (poll(&v)) { seen = true; break; }
    cpu_relax();
    if ((spins & 1023) == 1023 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(ctx->spin_ms)) break;
  }
  if (!seen) {
    // long passes (config 5: seconds per pass): sleep-poll the word and, every millisecond, the stream itself, so a
    // device fault or a drained stream without a publish ends the wait with an error instead of spinning
    for (long polls = 0;; ++polls) {
      if (poll(&v)) { seen = true; break; }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
      if ((polls & 63) == 63) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) break;                       // everything queued has run
        if (q != hipErrorNotReady) return fail(ctx, GVI_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
      }
    }
    if (!seen && !poll(&v))
      return fail(ctx, GVI_ERR_STATE, "cost publish did not arrive (sequence word stale after the stream drained)");
  }
  __sync_synchronize();
  g.cost[i] = v;                                     // cost_value = sum of factor costs + 1/2 log det (added on the device)
  g.cost_valid[i] = true;
  if (out) *out = v;
  return GVI_OK;
}

static gvi_status ngd_cost_finish(gvi_ctx* ctx, int i, double* out) {
  GVICK(ngd_cost_publish(ctx, i));
  return ngd_cost_wait(ctx, i, out);
}

// ---- sharded factors: the two exchange steps (no-ops in a single process) ----
static bool dist_on(const gvi_ctx* ctx) { return ctx->dist.world > 1 || ctx->dist.fn || ctx->dist.comm; }

static gvi_status dist_allgather(gvi_ctx* ctx, const void* send, void* recv, int64_t count) {
  gvi_ctx::Dist& d = ctx->dist;
  if (d.fn) {
    if (d.fn(d.user, send, recv, count, (void*)ctx->stream) != 0) return fail(ctx, GVI_ERR_HIP, "all-gather callback failed");
    return GVI_OK;
  }
  if (!d.comm) return fail(ctx, GVI_ERR_STATE, "sharded context without a transport: call gvi_dist_init_rccl / _callback");
  const int rc = d.ncclAllGather(send, recv, (size_t)count, 8 /* ncclFloat64 */, d.comm, ctx->stream);
  if (rc != 0) return fail(ctx, GVI_ERR_HIP, "ncclAllGather failed with code " + std::to_string(rc));
  return GVI_OK;
}

// state range [lo, hi] this rank's factors touch; gathered once per problem definition over all ranks
static gvi_status dist_ranges(gvi_ctx* ctx) {
  gvi_ctx::Dist& d = ctx->dist;
  if (d.ranges_valid) return GVI_OK;
  int lo = ctx->T, hi = -1;
  for (auto& s : ctx->sets)
    for (int k = 0; k < s->K; ++k) {
      lo = std::min(lo, (int)s->start[k]);
      hi = std::max(hi, (int)s->start[k] + (s->d == 2 * ctx->n ? 1 : 0));
    }
  if (hi < lo) { lo = 0; hi = -1; }                         // a rank without factors sends nothing
  const int W = d.world;
  HIPCK(ctx, d.ranges.ensure((size_t)(3 * W + 4) * 8));     // [int range table | own range | cost parts]
  double mine[2] = {(double)lo, (double)hi};
  double* dsend = d.ranges.d() + 2 * W;
  HIPCK(ctx, hipMemcpyAsync(dsend, mine, 16, hipMemcpyHostToDevice, ctx->stream));
  HIPCK(ctx, hipStreamSynchronize(ctx->stream));            // `mine` is a stack buffer
  HIPCK(ctx, d.recv.ensure((size_t)2 * W * 8));
  GVICK(dist_allgather(ctx, dsend, d.recv.p, 2));
  std::vector<double> all(2 * W);
  HIPCK(ctx, hipMemcpyAsync(all.data(), d.recv.p, (size_t)2 * W * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCK(ctx, hipStreamSynchronize(ctx->stream));
  d.lo.resize(W); d.hi.resize(W);
  std::vector<int32_t> packed(2 * W);
  d.maxlen = 1;
  for (int r = 0; r < W; ++r) {
    d.lo[r] = (int32_t)all[2 * r]; d.hi[r] = (int32_t)all[2 * r + 1];
    packed[2 * r] = d.lo[r]; packed[2 * r + 1] = d.hi[r];
    d.maxlen = std::max(d.maxlen, d.hi[r] - d.lo[r] + 1);
  }
  HIPCK(ctx, hipMemcpyAsync(d.ranges.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCK(ctx, hipStreamSynchronize(ctx->stream));
  const size_t per = (size_t)ctx->n + 2 * nn_(ctx);
  HIPCK(ctx, d.send.ensure((size_t)(d.maxlen + 1) * per * 8));                     // + the cost record of the fused trial
  HIPCK(ctx, d.recv.ensure(std::max<size_t>((size_t)W * (d.maxlen + 1) * per * 8, (size_t)2 * W * 8)));
  HIPCK(ctx, hipMemsetAsync(d.send.p, 0, (size_t)(d.maxlen + 1) * per * 8, ctx->stream));   // padding records stay zero
  d.rec_fresh = false;
  d.ranges_valid = true;
  return GVI_OK;
}

// exchange 0 on gradient buffer gb: pack own records -> all-gather -> fold in rank order back into exch0[gb]
static gvi_status dist_exchange0(gvi_ctx* ctx, int gb, bool with_cost = false, int publish_slot = -1) {
  if (!dist_on(ctx)) return GVI_OK;
  GVICK(dist_ranges(ctx));
  gvi_ctx::Dist& d = ctx->dist;
  const size_t T = ctx->T, n = ctx->n, nn = n * n, per = n + 2 * nn;
  double* eg = ctx->ngd.exch0[gb].d();
  double* eD = eg + T * n;
  double* eU = eD + T * nn;
  const int lo = d.lo[d.rank], len = d.hi[d.rank] - d.lo[d.rank] + 1;
  // with_cost: the rank's partial cost sum (exch1[0]) rides as one more record and comes back as the ordered total in
  // exch1[0] -- one all-gather for the fused trial instead of exchange 0 + exchange 1; publish_slot >= 0: the fold launch
  // also publishes {total + 1/2 log det of that NGD slot, sequence} to the host
  const int stride = d.maxlen + (with_cost ? 1 : 0);
  const int64_t np = (int64_t)stride * per;
  if (!d.rec_fresh) {                                        // else: written by the assemble (and cost_sum_all_kernel)
    hipLaunchKernelGGL(dist_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, ctx->T, ctx->n, lo, len, d.maxlen,
                       eg, eD, eU, d.send.d(), with_cost ? (const double*)ctx->ngd.exch1.d() : (const double*)nullptr);
    HIPCK(ctx, hipGetLastError());
  }
  d.rec_fresh = false;
  GVICK(dist_allgather(ctx, d.send.p, d.recv.p, np));
  const bool pub = with_cost && publish_slot >= 0;
  if (pub) ctx->seq += 1.0;
  const int64_t nf = (int64_t)T * per + 1;
  hipLaunchKernelGGL(dist_fold_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, ctx->stream, ctx->T, ctx->n, d.world, d.maxlen, stride,
                     (const int32_t*)d.ranges.p, d.recv.d(), eg, eD, eU, with_cost ? ctx->ngd.exch1.d() : (double*)nullptr,
                     pub ? (const double*)ctx->ngd.hld[publish_slot].d() : (const double*)nullptr,
                     pub ? pub_slot(ctx) : (double*)nullptr, ctx->seq);
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

// exchange 1: the ranks' partial cost sums (exch1[0]) -> their ordered total (exch1[0]); recv is reused after exchange 0
static gvi_status dist_exchange1(gvi_ctx* ctx) {
  if (!dist_on(ctx)) return GVI_OK;
  GVICK(dist_ranges(ctx));
  gvi_ctx::Dist& d = ctx->dist;
  double* parts = d.ranges.d() + 2 * d.world + 2;           // [world] behind the range table
  GVICK(dist_allgather(ctx, ctx->ngd.exch1.p, parts, 1));
  hipLaunchKernelGGL(dist_cost_fold_kernel, dim3(1), dim3(64), 0, ctx->stream, d.world, parts, ctx->ngd.exch1.d());
  HIPCK(ctx, hipGetLastError());
  return GVI_OK;
}

gvi_status gvi_ngd_init(gvi_ctx* ctx, const double* mu, const double* D, const double* U) {
  if (!ctx) return GVI_ERR_ARG;
  if (ctx->T < 1) return fail(ctx, GVI_ERR_STATE, "call gvi_chain_set first");
  if (!mu || !D || (!U && ctx->T > 1)) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, n = ctx->n, nn = n * n, bt = bt_count(ctx);
  for (int i = 0; i < 2; ++i) {
    HIPCK(ctx, g.mu[i].ensure(T * n * 8));
    HIPCK(ctx, g.Lam[i].ensure(bt * 8));
    HIPCK(ctx, g.Sig[i].ensure(bt * 8));
    HIPCK(ctx, g.hld[i].ensure(8));
  }
  for (int i = 0; i < 2; ++i) {
    HIPCK(ctx, g.exch0[i].ensure((T * n + bt) * 8));
    HIPCK(ctx, g.dmu2[i].ensure(T * n * 8));
  }
  g.gcur = 0; g.grad_valid = false; g.grad_slot = -1;
  ctx->solve_deferred[0] = ctx->solve_deferred[1] = false;
  ctx->asm_pending[0] = ctx->asm_pending[1] = false;
  ctx->last_first_accepted = true;                 // adaptive fusing starts afresh: the pass order of a run must not depend on the previous problem
  HIPCK(ctx, g.exch1.ensure(8));
  HIPCK(ctx, g.dmu.ensure(T * n * 8));
  HIPCK(ctx, g.dLam.ensure(bt * 8));
  HIPCK(ctx, g.total.ensure(8));
  for (auto& s : ctx->sets) GVICK(ensure_set_buffers(ctx, *s));
  g.cur = 0; g.have_trial = false;
  for (auto& s : ctx->sets) { s->prep_slot = -1; s->warm_count = 0; }
  g.cost_valid[0] = g.cost_valid[1] = false;
  GVICK(h2d(ctx, g.mu[0].p, mu, T * n * 8));
  GVICK(h2d(ctx, g.Lam[0].p, D, T * nn * 8));
  if (T > 1) GVICK(h2d(ctx, g.Lam[0].d() + T * nn, U, (T - 1) * nn * 8));
  GVICK(sync(ctx));                                // the caller's buffers are free on return
  GVICK(ngd_refresh(ctx, 0));                      // stream-ordered before everything that follows: not waited for
  HIPCK(ctx, hipGetLastError());
  g.ready = true;
  return GVI_OK;
}

gvi_status gvi_ngd_cost(gvi_ctx* ctx, double* cost) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  if (!g.cost_valid[g.cur]) {
    GVICK(ngd_cost_local(ctx, g.cur));
    GVICK(dist_exchange1(ctx));
    GVICK(ngd_cost_finish(ctx, g.cur, nullptr));
  }
  if (cost) *cost = g.cost[g.cur];
  return GVI_OK;
}

gvi_status gvi_ngd_cost_local(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  return ngd_cost_local(ctx, ctx->ngd.cur);
}

gvi_status gvi_ngd_cost_finish(gvi_ctx* ctx, double* cost) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  return ngd_cost_finish(ctx, ctx->ngd.cur, cost);
}

gvi_status gvi_ngd_factor_costs(gvi_ctx* ctx, int set_id, double* costs) {
  GVICK(ngd_check(ctx));
  FactorSet* s = get_set(ctx, set_id);
  if (!s || !costs) return fail(ctx, GVI_ERR_ARG, "bad set id / NULL");
  HIPCK(ctx, hipSetDevice(ctx->device));
  const int i = ctx->ngd.cur;
  GVICK(ngd_flush_gather(ctx, i));
  GVICK(gvi_costs_dev(ctx, set_id, s->mu_k[i].d(), s->Sigma_k[i].d(), s->cost.d()));
  GVICK(d2h(ctx, costs, s->cost.p, (size_t)s->K * 8));
  return sync(ctx);
}

// full moments pass of every set at NGD slot `slot` (per-factor Vdmu / Vddmu / E[psi] / cost)
// All sets' moments (full = 1) or cost (full = 0) launches at slot.  The chain pattern -- set 0 binary priors (d = 2n,
// m = n), set 1 unary factors (d = m = n), both on the SGPR-operand kernels -- goes out as ONE launch.
static gvi_status ngd_moments_launch(gvi_ctx* ctx, int slot, int full) {
  if (full) ++ctx->n_full_pass; else ++ctx->n_cost_pass;
  if (ctx->pair_fuse && ctx->sets.size() == 2 && !ctx->profile_all) {
    FactorSet& s0 = *ctx->sets[0];
    FactorSet& s1 = *ctx->sets[1];
    const bool shape = s0.kind == KIND_QUAD_PRIOR && s1.kind == KIND_FIXED_PRIOR && s0.d == 12 && s1.d == 6 &&
                       !s0.closed_form && !s1.closed_form && s0.K > 0 && s1.K > 0;
    const bool shape_orbit = orbit_supported(ctx, s0) && orbit_supported(ctx, s1) && s0.m == s1.m && s0.all_pos && s1.all_pos &&
                             !s0.closed_form && !s1.closed_form && s0.K > 0 && s1.K > 0;
    if (shape || shape_orbit) {
      gvi_ctx::Deferred d0, d1;
      ctx->defer = &d0;
      gvi_status st = run_moments(ctx, s0, s0.mu_k[slot].d(), nullptr, full);
      if (st == GVI_OK) { ctx->defer = &d1; st = run_moments(ctx, s1, s1.mu_k[slot].d(), nullptr, full); }
      ctx->defer = nullptr;
      GVICK(st);
      const int want = full ? 0 : 1;
      if (d0.kind == 2 && d1.kind == 2) {              // both sets on the sign-orbit kernel
        const bool prof = ctx->profile && full && (ctx->profile_count++ % ctx->profile_every) == 0;
        if (prof) {
          for (int e = 0; e < 2; ++e)
            if (!s0.ev[0][e]) HIPCK(ctx, hipEventCreate(&s0.ev[0][e]));
        }
        launch_orbit_pair(d0.oa, d1.oa, d0.m, std::max(d0.smax, d1.smax), full != 0, ctx->stream, ctx->orbit_stack,
                          prof ? s0.ev[0][0] : nullptr, prof ? s0.ev[0][1] : nullptr);
        HIPCK(ctx, hipGetLastError());
        if (prof) s0.ev_set[0] = true;
        s0.fused_pair = s1.fused_pair = true;
        return GVI_OK;
      }
      if (d0.kind == 2 || d1.kind == 2) {              // only one of them: issue both on their own
        s0.fused_pair = s1.fused_pair = false;
        if (d0.kind >= 0) GVICK(run_moments(ctx, s0, s0.mu_k[slot].d(), nullptr, full));
        if (d1.kind >= 0) GVICK(run_moments(ctx, s1, s1.mu_k[slot].d(), nullptr, full));
        return GVI_OK;
      }
      if (d0.kind == want && d1.kind == want) {
        const int nb0 = (int)(d0.grid.x * d0.grid.y), nb1 = (int)(d1.grid.x * d1.grid.y);
        const bool prof = ctx->profile && full && (ctx->profile_count++ % ctx->profile_every) == 0;
        if (prof) {
          for (int e = 0; e < 2; ++e)
            if (!s0.ev[0][e]) HIPCK(ctx, hipEventCreate(&s0.ev[0][e]));
          HIPCK(ctx, hipEventRecord(s0.ev[0][0], ctx->stream));
        }
        if (full && ctx->sreg_pipe && s0.table->Zq.p && s1.table->Zq.p)    // every block: one prior item + strided unary items
          hipLaunchKernelGGL((moments_sreg_pair_kernel<12, 6, 6, 6, true, true>), dim3(std::max(nb0, nb1)), dim3(256), 0, ctx->stream,
                             d0.a, d1.a, (int)d0.grid.x, nb0, (int)d1.grid.x);
        else if (full)
          hipLaunchKernelGGL((moments_sreg_pair_kernel<12, 6, 6, 6, true>), dim3(nb0 + nb1), dim3(256), 0, ctx->stream, d0.a,
                             d1.a, (int)d0.grid.x, nb0, (int)d1.grid.x);
        else
          hipLaunchKernelGGL((moments_scost_pair_kernel<12, 6, 6, 6, 2>), dim3(nb0 + nb1), dim3(256), 0, ctx->stream, d0.a, d1.a,
                             (int)d0.grid.x, nb0, (int)d1.grid.x);
        HIPCK(ctx, hipGetLastError());
        if (prof) { HIPCK(ctx, hipEventRecord(s0.ev[0][1], ctx->stream)); s0.ev_set[0] = true; }
        s0.fused_pair = s1.fused_pair = true;
        return GVI_OK;
      }
      // a deferred launch that did not pair up: issue it on its own below (run_moments again, undeferred)
      if (d0.kind < 0 && d1.kind < 0) { s0.fused_pair = s1.fused_pair = false; return GVI_OK; }   // both already launched
      if (d0.kind >= 0 && d1.kind < 0) { s0.fused_pair = false; return run_moments(ctx, s0, s0.mu_k[slot].d(), nullptr, full); }
      if (d1.kind >= 0 && d0.kind < 0) { s1.fused_pair = false; return run_moments(ctx, s1, s1.mu_k[slot].d(), nullptr, full); }
    }
  }
  // the planning graph: d = 8 priors + d = 4 hinge-on-SDF obstacle factors + d = 4 anchors, all three on lane-per-point
  // register kernels -> ONE launch (moments_planar3_kernel); any other shape: one launch per set
  if (ctx->pair_fuse && ctx->sets.size() == 3 && !ctx->profile_all && ctx->variant == 0 && !ctx->prefer_opsi) {
    FactorSet& s0 = *ctx->sets[0];
    FactorSet& s1 = *ctx->sets[1];
    FactorSet& s2 = *ctx->sets[2];
    const bool shape = s0.kind == KIND_QUAD_PRIOR && s0.d == 8 && s0.m == 4 && s1.kind == KIND_HINGE_SDF_2D && s1.d == 4 &&
                       s2.kind == KIND_FIXED_PRIOR && s2.d == 4 && !s0.closed_form && !s2.closed_form && s0.K > 0 && s1.K > 0 && s2.K > 0;
    if (shape) {
      gvi_ctx::Deferred dq[3];
      FactorSet* ss[3] = {&s0, &s1, &s2};
      gvi_status st = GVI_OK;
      for (int q = 0; q < 3 && st == GVI_OK; ++q) {
        dq[q].capture_any = true;
        ctx->defer = &dq[q];
        st = run_moments(ctx, *ss[q], ss[q]->mu_k[slot].d(), nullptr, full);
      }
      ctx->defer = nullptr;
      GVICK(st);
      if (dq[0].kind == 3 && dq[1].kind == 3 && dq[2].kind == 3) {
        const int nb0 = (int)(dq[0].grid.x * dq[0].grid.y), nb1 = (int)(dq[1].grid.x * dq[1].grid.y), nb2 = (int)(dq[2].grid.x * dq[2].grid.y);
        const bool prof = ctx->profile && full && (ctx->profile_count++ % ctx->profile_every) == 0;
        if (prof) {                                          // the bracket is booked on the obstacle set (the dominant one)
          for (int e = 0; e < 2; ++e)
            if (!s1.ev[0][e]) HIPCK(ctx, hipEventCreate(&s1.ev[0][e]));
          HIPCK(ctx, hipEventRecord(s1.ev[0][0], ctx->stream));
        }
        if (full)
          hipLaunchKernelGGL((moments_planar3_kernel<true>), dim3(nb0 + nb1 + nb2), dim3(256), 0, ctx->stream, dq[0].a, dq[1].a, dq[2].a,
                             (int)dq[0].grid.x, nb0, (int)dq[1].grid.x, nb1, (int)dq[2].grid.x, (ctx->sreg_pipe && s0.table->Zq.p) ? 1 : 0);
        else
          hipLaunchKernelGGL((moments_planar3_kernel<false>), dim3(nb0 + nb1 + nb2), dim3(256), 0, ctx->stream, dq[0].a, dq[1].a, dq[2].a,
                             (int)dq[0].grid.x, nb0, (int)dq[1].grid.x, nb1, (int)dq[2].grid.x, 0);
        HIPCK(ctx, hipGetLastError());
        if (prof) { HIPCK(ctx, hipEventRecord(s1.ev[0][1], ctx->stream)); s1.ev_set[0] = true; }
        for (auto* q : ss) q->fused_pair = false;
        return GVI_OK;
      }
      for (int q = 0; q < 3; ++q)                            // some set took another route: whatever was captured goes out on its own
        if (dq[q].kind >= 0) { ss[q]->fused_pair = false; GVICK(run_moments(ctx, *ss[q], ss[q]->mu_k[slot].d(), nullptr, full)); }
      return GVI_OK;
    }
  }
  for (auto& s : ctx->sets) { s->fused_pair = false; GVICK(run_moments(ctx, *s, s->mu_k[slot].d(), nullptr, full)); }
  return GVI_OK;
}

// ---- the full pass as ONE launch (kernels_fused.hpp) ----
// Blocks that own items: one per factor of the first (heavy) set; a longer second set wraps around (block b also takes its
// items b + nblk, ...).  0: more than FUSED_MAX_ITEMS items per block would be needed.
static int fused_nblk(const gvi_ctx* ctx) {
  const int K0 = ctx->sets[0]->K, K1 = ctx->sets.size() > 1 ? ctx->sets[1]->K : 0;
  const int nblk = K0;
  const int per = 1 + (K1 + nblk - 1) / nblk;
  return per <= FUSED_MAX_ITEMS ? nblk : 0;
}

static bool fused_ok(const gvi_ctx* ctx, int slot) {
  if (!ctx->fused || !ctx->pair_fuse || ctx->profile_all || ctx->sets.empty() || ctx->sets.size() > 2) return false;
  if (ctx->update_rule != GVI_RULE_NGD) return false;     // the JKO map reads the sets' Sigma^-1 after the pass
  const int m = ctx->sets[0]->m;
  for (auto& sp : ctx->sets) {
    const FactorSet& s = *sp;
    if (!orbit_supported(ctx, s) || !s.all_pos || s.closed_form || s.K <= 0 || s.m != m) return false;
    if (s.prep_slot == slot) return false;          // products already resident: the plain route skips the prep
    if (s.table->orb.tile_s.empty()) return false;
  }
  // instantiated shapes: the chain patterns of BASELINE configs[1..3] (binary prior d = 2n + unary factor d = n), Cholesky route
  const int d0 = ctx->sets[0]->d, d1 = ctx->sets.size() > 1 ? ctx->sets[1]->d : d0 / 2;
  if (!((m == 6 && d0 == 12 && d1 == 6) || (m == 2 && d0 == 4 && d1 == 2))) return false;
  for (auto& sp : ctx->sets)
    if (!sp->dev().chol) return false;
  return fused_nblk(ctx) > 0;
}

static gvi_status ngd_fused_full(gvi_ctx* ctx, int slot, int publish_slot) {
  NgdState& g = ctx->ngd;
  FusedArgs A{};
  A.nsets = (int)ctx->sets.size();
  A.koff[0] = 0;
  int smax = 0, dmax = 0, copies = 1;
  const int m = ctx->sets[0]->m;
  A.nblk = fused_nblk(ctx);
  for (int si = 0; si < A.nsets; ++si) {
    FactorSet& s = *ctx->sets[si];
    FusedSet& F = A.s[si];
    s.prep_slot = -1;                      // the fused pass keeps its products in LDS: nothing resident for a later pass
    s.nchunk = 4; s.chunk = s.table->Np;
    s.use_reg = s.use_split = false; s.use_orbit = true; s.fused_pair = true;
    F.f = s.dev();
    if (ctx->warm_start) {                 // (symmetric-root sets only; the Cholesky route ignores it)
      if (s.Vws.bytes == 0) { HIPCK(ctx, s.Vws.ensure((size_t)s.K * s.d * s.d * 8)); s.warm_count = 0; }
      F.f.Vws = s.Vws.d();
      F.f.warm = (s.warm_count % 32) != 0;
      s.warm_count++;
    }
    GVICK(orbit_args(ctx, s, 1, &F.oa));
    F.oa.partial = nullptr;
    F.mu = s.mu_k[slot].d(); F.Sigma = s.Sigma_k[slot].d();
    F.start = s.chain_structured ? nullptr : (const int32_t*)s.dstart.p;      // null: start[k] == k, one dependent load less
    F.mu_k = s.mu_k[slot].d(); F.Sigma_k = s.Sigma_k[slot].d();
    F.Ephi = s.Ephi.d(); F.cost = s.cost.d(); F.Vdmu = s.Vdmu.d(); F.Vddmu = s.Vddmu.d();
    A.koff[si + 1] = A.koff[si] + s.K;
    A.cl.cost[si] = s.cost.d(); A.cl.K[si] = s.K;
    smax = std::max(smax, s.table->orb.smax);
    dmax = std::max(dmax, s.d);
    copies = std::max(copies, F.oa.copies);
  }
  // (the walk regions are sized for the widest set with the most accumulator copies: an upper bound for every set)
  const int items = 1 + (A.nsets > 1 ? (ctx->sets[1]->K + A.nblk - 1) / A.nblk : 0);      // most items of a block
  const size_t lds = fused_lds_doubles(dmax, m, copies, items) * 8;
  if (lds > 160 * 1024) return fail(ctx, GVI_ERR_UNSUPPORTED, "fused pass: LDS budget");
  A.cl.nsets = A.nsets;
  if (A.nsets == 1) A.koff[2] = A.koff[1];
  unsigned extra = 0;
  if (g.gpend[slot].on) {
    const size_t T = ctx->T, nn = nn_(ctx);
    A.gather = 1; A.n = ctx->n;
    A.gmu = g.gpend[slot].dmu ? g.gpend[slot].mu_from : g.mu[slot].d();
    A.gdmu = g.gpend[slot].dmu; A.gstep = g.gpend[slot].step;
    A.SigD = g.Sig[slot].d(); A.SigU = g.Sig[slot].d() + T * nn;
    A.mu_out = g.mu[slot].d(); A.nmu = (int64_t)T * ctx->n;
    if (A.gdmu) extra = (unsigned)((A.nmu + 255) / 256);
    g.gpend[slot].on = false;
  }
  EpiTail& tail = A.tail;
  tail.on = 0; tail.pred = ctx->cur_pred; tail.pred_val = ctx->cur_pred_val; tail.c0_use_imm = 1;
  tail.safe = ctx->safe_publish ? 1 : 0;
  if (publish_slot >= 0) {
    const size_t need = (size_t)128 * (2 + (size_t)A.koff[A.nsets] / EPI_GROUP);
    if (ctx->epi_counter.bytes < need) {
      HIPCK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCK(ctx, ctx->epi_counter.ensure(need));
      HIPCK(ctx, hipMemsetAsync(ctx->epi_counter.p, 0, need, ctx->stream));
    }
    ctx->seq += 1.0;
    tail.on = 1; tail.acc = g.exch1.d(); tail.half_logdet = g.hld[publish_slot].d();
    tail.host_out = pub_slot(ctx); tail.seq = ctx->seq; tail.counter = (unsigned*)ctx->epi_counter.p;
    if (ctx->pipe_tail) {
      tail.accept = ctx->pipe_dev.d() + ctx->pub_ring; tail.cost_dev = ctx->pipe_dev.d() + 2;
      tail.slot_cur = 1 - publish_slot; tail.slot_trial = publish_slot;
      tail.c0_use_imm = ctx->pipe_c0_imm ? 1 : 0; tail.c0_imm = ctx->pipe_c0;
    }
  }
  ++ctx->n_full_pass;
  FactorSet& s0 = *ctx->sets[0];
  const bool prof = ctx->profile && (ctx->profile_count++ % ctx->profile_every) == 0;
  if (prof)
    for (int e = 0; e < 2; ++e)
      if (!s0.ev[0][e]) HIPCK(ctx, hipEventCreate(&s0.ev[0][e]));
  hipEvent_t e0 = prof ? s0.ev[0][0] : nullptr, e1 = prof ? s0.ev[0][1] : nullptr;
  const unsigned grid = (unsigned)A.nblk + extra;
  const int d0 = ctx->sets[0]->d, d1 = A.nsets > 1 ? ctx->sets[1]->d : d0 / 2;
  if (m == 6 && smax <= 4 && d0 == 12 && d1 == 6) GVICK((launch_fused_t<6, 4, 4, 12, 6>(ctx, A, grid, lds, dmax, copies, items, e0, e1)));
  else if (m == 6 && d0 == 12 && d1 == 6) GVICK((launch_fused_t<6, 6, 2, 12, 6>(ctx, A, grid, lds, dmax, copies, items, e0, e1)));
  else if (m == 2 && d0 == 4 && d1 == 2) GVICK((launch_fused_t<2, 4, 4, 4, 2>(ctx, A, grid, lds, dmax, copies, items, e0, e1)));
  else return fail(ctx, GVI_ERR_UNSUPPORTED, "fused pass: shape not instantiated");
  HIPCK(ctx, hipGetLastError());
  if (prof) s0.ev_set[0] = true;
  return GVI_OK;
}

// ---- the planning graph's full pass as ONE launch (kernels_block.hpp) ----
// The shape of moments_planar3_kernel (priors d = 8 / hinge on the SDF d = 4 / anchors d = 4, all on their lane-per-point
// kernels), at most four chunks per factor (a workgroup's four waves take them), products
// of this state not resident yet.  *done = false: not this shape -- the caller takes the three launches.
static gvi_status ngd_block3_full(gvi_ctx* ctx, int slot, int publish_slot, bool* done) {
  *done = false;
  if (!ctx->fused || !ctx->pair_fuse || ctx->profile_all || ctx->sets.size() != 3 || ctx->variant != 0 || ctx->prefer_opsi) return GVI_OK;
  if (ctx->update_rule != GVI_RULE_NGD) return GVI_OK;     // the JKO map reads the sets' Sigma^-1 after the pass
  FactorSet* ss[3] = {ctx->sets[0].get(), ctx->sets[1].get(), ctx->sets[2].get()};
  FactorSet &s0 = *ss[0], &s1 = *ss[1], &s2 = *ss[2];
  const bool shape = s0.kind == KIND_QUAD_PRIOR && s0.d == 8 && s0.m == 4 && s1.kind == KIND_HINGE_SDF_2D && s1.d == 4 &&
                     s2.kind == KIND_FIXED_PRIOR && s2.d == 4 && !s0.closed_form && !s2.closed_form && s0.K > 0 && s1.K > 0 && s2.K > 0;
  if (!shape) return GVI_OK;
  for (auto* q : ss)
    if (q->prep_slot == slot) return GVI_OK;                 // products already resident: the plain route skips the prep
  if (!s0.dev().chol || s1.dev().chol || !s2.dev().chol) return GVI_OK;     // the products the kernel is compiled for (kernels_block.hpp)
  NgdState& g = ctx->ngd;
  // the sets' moments launches exactly as they would go out on their own (chunking, table pointers, partial buffers)
  gvi_ctx::Deferred dq[3];
  gvi_status st = GVI_OK;
  for (int q = 0; q < 3 && st == GVI_OK; ++q) {
    dq[q].capture_any = true;
    ctx->defer = &dq[q];
    st = run_moments(ctx, *ss[q], ss[q]->mu_k[slot].d(), nullptr, 1);
  }
  ctx->defer = nullptr;
  GVICK(st);
  bool ok = true;
  for (int q = 0; q < 3; ++q) ok = ok && dq[q].kind == 3 && dq[q].grid.y <= 4 && (int)dq[q].grid.y == dq[q].a.nchunk;
  if (!ok) {                                   // some set took another route: whatever was captured is re-planned by the caller
    for (int q = 0; q < 3; ++q)
      if (dq[q].kind < 0) return fail(ctx, GVI_ERR_STATE, "planning-graph pass: a set launched while its siblings were captured");
    return GVI_OK;
  }
  ++ctx->n_full_pass;
  Block3Args A{};
  A.nitems = s0.K + s1.K + s2.K;
  A.pipe = (ctx->sreg_pipe && s0.table->Zq.p) ? 1 : 0;
  for (int q = 0; q < 3; ++q) {
    FactorSet& s = *ss[q];
    BlockSet& B = A.s[q];
    B.a = dq[q].a;
    if (ctx->warm_start) {                 // warm-started Jacobi of the symmetric-root sets, as ngd_prep_all
      if (s.Vws.bytes == 0) { HIPCK(ctx, s.Vws.ensure((size_t)s.K * s.d * s.d * 8)); s.warm_count = 0; }
      B.a.f.Vws = s.Vws.d();
      B.a.f.warm = (s.warm_count % 32) != 0;
      s.warm_count++;
    }
    B.start = (const int32_t*)s.dstart.p;
    B.mu_k = s.mu_k[slot].d(); B.Sigma_k = s.Sigma_k[slot].d();
    B.Ephi = s.Ephi.d(); B.cost = s.cost.d(); B.Vdmu = s.Vdmu.d(); B.Vddmu = s.Vddmu.d();
    s.prep_slot = slot;                        // the products go to memory as prep_all_kernel leaves them
    s.fused_pair = false;
  }
  unsigned extra = 0;
  if (g.gpend[slot].on) {
    const size_t T = ctx->T, nn = nn_(ctx);
    A.gather = 1; A.n = ctx->n;
    A.gmu = g.gpend[slot].dmu ? g.gpend[slot].mu_from : g.mu[slot].d();
    A.gdmu = g.gpend[slot].dmu; A.gstep = g.gpend[slot].step;
    A.SigD = g.Sig[slot].d(); A.SigU = g.Sig[slot].d() + T * nn;
    A.mu_out = g.mu[slot].d(); A.nmu = (int64_t)T * ctx->n;
    if (A.gdmu) extra = (unsigned)((A.nmu + 255) / 256);
    g.gpend[slot].on = false;
  }
  EpiTail& tail = A.tail;
  tail.on = 0; tail.pred = ctx->cur_pred; tail.pred_val = ctx->cur_pred_val; tail.c0_use_imm = 1;
  tail.safe = ctx->safe_publish ? 1 : 0;
  if (publish_slot >= 0) {
    const size_t need = (size_t)128 * (2 + (size_t)A.nitems / EPI_GROUP);
    if (ctx->epi_counter.bytes < need) {
      HIPCK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCK(ctx, ctx->epi_counter.ensure(need));
      HIPCK(ctx, hipMemsetAsync(ctx->epi_counter.p, 0, need, ctx->stream));
    }
    ctx->seq += 1.0;
    tail.on = 1; tail.acc = g.exch1.d(); tail.half_logdet = g.hld[publish_slot].d();
    tail.host_out = pub_slot(ctx); tail.seq = ctx->seq; tail.counter = (unsigned*)ctx->epi_counter.p;
    if (ctx->pipe_tail) {
      tail.accept = ctx->pipe_dev.d() + ctx->pub_ring; tail.cost_dev = ctx->pipe_dev.d() + 2;
      tail.slot_cur = 1 - publish_slot; tail.slot_trial = publish_slot;
      tail.c0_use_imm = ctx->pipe_c0_imm ? 1 : 0; tail.c0_imm = ctx->pipe_c0;
    }
  }
  A.cl.nsets = 3;
  for (int q = 0; q < 3; ++q) { A.cl.cost[q] = ss[q]->cost.d(); A.cl.K[q] = ss[q]->K; }
  const bool prof = ctx->profile && (ctx->profile_count++ % ctx->profile_every) == 0;
  if (prof) {                                  // the bracket is booked on the obstacle set (the dominant one)
    for (int e = 0; e < 2; ++e)
      if (!s1.ev[0][e]) HIPCK(ctx, hipEventCreate(&s1.ev[0][e]));
    HIPCK(ctx, hipEventRecord(s1.ev[0][0], ctx->stream));
  }
  const unsigned nblk = (unsigned)(block3_blocks(s0.K, A.s[0].a.nchunk) + block3_blocks(s1.K, A.s[1].a.nchunk) + block3_blocks(s2.K, A.s[2].a.nchunk));
  hipLaunchKernelGGL(factor_block3_kernel, dim3(nblk + extra), dim3(256), 4 * block3_lds_doubles() * 8, ctx->stream, A);
  HIPCK(ctx, hipGetLastError());
  if (prof) { HIPCK(ctx, hipEventRecord(s1.ev[0][1], ctx->stream)); s1.ev_set[0] = true; }
  *done = true;
  return GVI_OK;
}

static gvi_status ngd_moments_full(gvi_ctx* ctx, int slot, int publish_slot = -1) {
  if ((int)ctx->sets.size() > MAX_SETS) return fail(ctx, GVI_ERR_UNSUPPORTED, "more than 8 factor sets");
  for (auto& s : ctx->sets)
    if (s->kind == KIND_HOST_CALLBACK) return fail(ctx, GVI_ERR_UNSUPPORTED, "resident NGD needs device psi kinds");
  StageScope scope(ctx, STAGE_FACTORS);
  if (fused_ok(ctx, slot)) return ngd_fused_full(ctx, slot, publish_slot);
  {
    bool done = false;
    GVICK(ngd_block3_full(ctx, slot, publish_slot, &done));
    if (done) return GVI_OK;
  }
  GVICK(ngd_prep_all(ctx, slot));
  GVICK(ngd_moments_launch(ctx, slot, 1));
  return ngd_epilogue_all(ctx, 1, publish_slot);
}

// assemble-on-load applies: every set chain-structured, the solve of this buffer will go out through the dual launch (or
// the flushing solve), nobody needs [g | V_D | V_U] in memory before that
static bool asm_on_load_ok(const gvi_ctx* ctx) {
  if (!ctx->asm_on_load || dist_on(ctx) || ctx->update_rule != GVI_RULE_NGD || ctx->sets.empty()) return false;
  if (!(ctx->dual_chain && ctx->side_solve && chain_supported(ctx->n) && ctx->T > 1)) return false;
  for (auto& s : ctx->sets)
    if (!s->chain_structured && !(s->d == ctx->n && s->K <= ASM_SPARSE_MAX)) return false;    // (sparse unary sets: AsmSet::sp)
  return true;
}

static AsmList make_asm_list(gvi_ctx* ctx) {
  AsmList L{};
  L.nsets = (int)ctx->sets.size();
  for (int i = 0; i < L.nsets; ++i) {
    FactorSet& s = *ctx->sets[i];
    L.s[i].K = s.K; L.s[i].d = s.d; L.s[i].Vdmu = s.Vdmu.d(); L.s[i].Vddmu = s.Vddmu.d();
    L.s[i].nsp = 0;
    if (!s.chain_structured) {
      L.s[i].nsp = s.K;
      for (int k = 0; k < s.K && k < ASM_SPARSE_MAX; ++k) L.s[i].sp[k] = s.start[k];
    }
  }
  return L;
}

static gvi_status ngd_scatter_now(gvi_ctx* ctx, int slot, int gb);

// ordered assemble of the per-factor results into gradient buffer `gb` -- launched now, or left to the first pass of the
// chain operations that consume the buffer (assemble-on-load)
static gvi_status ngd_scatter(gvi_ctx* ctx, int slot, int gb) {
  if ((int)ctx->sets.size() <= MAX_SETS && asm_on_load_ok(ctx)) {
    ctx->asm_pending[gb] = true;
    ctx->asm_pending[1 - gb] = false;               // its source (the sets' Vdmu / Vddmu) has just been overwritten
    return GVI_OK;
  }
  ctx->asm_pending[gb] = false;
  return ngd_scatter_now(ctx, slot, gb);
}

// stand-alone assemble of a buffer whose assemble was left pending (consumers other than the chain launches)
static gvi_status ngd_flush_assemble(gvi_ctx* ctx, int gb) {
  if (!ctx->asm_pending[gb]) return GVI_OK;
  ctx->asm_pending[gb] = false;
  return ngd_scatter_now(ctx, ctx->ngd.cur, gb);
}

static gvi_status ngd_scatter_now(gvi_ctx* ctx, int slot, int gb) {
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, n = ctx->n, nn = n * n;
  double* eg = g.exch0[gb].d();
  double* eD = eg + T * n;
  double* eU = eD + T * nn;
  const int64_t total = (int64_t)T * (n + 2 * nn);
  StageScope scope(ctx, STAGE_ASSEMBLE);
  double* rec = nullptr;                                     // sharded: the assemble writes this rank's exchange records too
  int rlo = 0, rlen = 0;
  if (dist_on(ctx)) {
    GVICK(dist_ranges(ctx));
    rec = ctx->dist.send.d(); rlo = ctx->dist.lo[ctx->dist.rank]; rlen = ctx->dist.hi[ctx->dist.rank] - rlo + 1;
  }
  hipLaunchKernelGGL(bt_scatter_all_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     make_set_list(ctx, slot), ctx->T, ctx->n, eg, eD, eU, ctx->cur_pred, ctx->cur_pred_val, rec, rlo, rlen);
  HIPCK(ctx, hipGetLastError());
  if (rec) ctx->dist.rec_fresh = true;
  return GVI_OK;
}

static gvi_status ngd_grad_local(gvi_ctx* ctx, int slot, int gb) {
  ctx->solve_deferred[gb] = false;                      // a parked solve of this buffer's previous contents is moot
  GVICK(ngd_moments_full(ctx, slot));
  return ngd_scatter(ctx, slot, gb);
}

// dmu = Vddmu^-1 (-Vdmu) from gradient buffer `gb` (dprecision = V - Lambda is formed inside the trial)
static gvi_status ngd_grad_finish(gvi_ctx* ctx, int gb) {
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, n = ctx->n, nn = n * n;
  double* eg = g.exch0[gb].d();
  double* eD = eg + T * n;
  double* eU = eD + T * nn;
  if (ctx->dual_chain && ctx->side_solve && chain_supported(ctx->n) && ctx->T > 1) {
    ctx->solve_deferred[gb] = true;                     // goes out with the next trial factorisation (ngd_trial_state)
    return GVI_OK;
  }
  GVICK(ngd_flush_assemble(ctx, gb));
  if (ctx->side_solve && chain_supported(ctx->n)) {
    // the solve only feeds mu_trial; the trial precision and its factorisation need Vddmu alone, so the solve goes
    // to the side stream and ngd_join_solve() waits for it right before the first reader of dmu
    if (!ctx->side) {
      HIPCK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
      HIPCK(ctx, hipEventCreateWithFlags(&ctx->ev_grad, hipEventDisableTiming));
      for (auto& e : ctx->ev_solve) HIPCK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIPCK(ctx, hipEventRecord(ctx->ev_grad, ctx->stream));
    HIPCK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_grad, 0));
    ctx->chain_stream = ctx->side; ctx->chain_ws = 1;
    const gvi_status st = run_bt_solve(ctx, eD, eU, eg, -1.0, g.dmu2[gb].d());
    ctx->chain_stream = nullptr; ctx->chain_ws = 0;
    GVICK(st);
    HIPCK(ctx, hipEventRecord(ctx->ev_solve[gb], ctx->side));
    ctx->solve_pending[gb] = true;
    return GVI_OK;
  }
  return run_bt_solve(ctx, eD, eU, eg, -1.0, g.dmu2[gb].d());
}

// main stream waits for a side-stream solve of gradient buffer gb (no-op when none is pending)
static gvi_status ngd_join_solve(gvi_ctx* ctx, int gb) {
  if (ctx->solve_deferred[gb]) {                        // nobody fused it with a factorisation: run it now, in stream order
    ctx->solve_deferred[gb] = false;
    NgdState& g = ctx->ngd;
    const size_t T = ctx->T, n = ctx->n, nn = n * n;
    double* eg = g.exch0[gb].d();
    ctx->chain_ws = 1;
    gvi_status st;
    if (ctx->asm_pending[gb]) {
      ctx->asm_pending[gb] = false;
      ChainWs w;
      st = ensure_chain_ws(ctx, w);
      if (st == GVI_OK) {
        ChainArgs a1 = make_chain_args(ctx, w, eg + T * n, eg + T * n + T * nn, eg, -1.0, false, nullptr, nullptr, g.dmu2[gb].d(), nullptr, false);
        a1.asm_on = 1; a1.asmG = eg; a1.asmD = eg + T * n; a1.asmU = eg + T * n + T * nn;
        const AsmList AL = make_asm_list(ctx);
        st = run_chain(ctx, a1, a1, false, true, &AL);
      }
    } else st = run_bt_solve(ctx, eg + T * n, eg + T * n + T * nn, eg, -1.0, g.dmu2[gb].d());
    ctx->chain_ws = 0;
    return st;
  }
  GVICK(ngd_flush_assemble(ctx, gb));
  if (ctx->solve_pending[gb]) {
    HIPCK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_solve[gb], 0));
    ctx->solve_pending[gb] = false;
  }
  return GVI_OK;
}

gvi_status gvi_ngd_gradients_local(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  ctx->ngd.grad_valid = false;
  return ngd_grad_local(ctx, ctx->ngd.cur, ctx->ngd.gcur);
}

gvi_status gvi_ngd_gradients_finish(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(ngd_grad_finish(ctx, ctx->ngd.gcur));
  ctx->ngd.grad_valid = true;
  ctx->ngd.grad_slot = ctx->ngd.cur;
  return GVI_OK;
}

gvi_status gvi_ngd_gradients(gvi_ctx* ctx) {
  GVICK(gvi_ngd_gradients_local(ctx));
  GVICK(dist_exchange0(ctx, ctx->ngd.gcur));
  return gvi_ngd_gradients_finish(ctx);
}

// trial proposal into the other slot: state, chain factorisation (log-det + marginals), gather
static gvi_status ngd_trial_state(gvi_ctx* ctx, double step) {
  NgdState& g = ctx->ngd;
  if (!(g.grad_valid && g.grad_slot == g.cur)) return fail(ctx, GVI_ERR_STATE, "call gvi_ngd_gradients first");
  const size_t Tn = (size_t)ctx->T * ctx->n, bt = bt_count(ctx);
  const int c = g.cur, t = 1 - c;
  g.cost_valid[t] = false;
  g.spec_ready = false;
  if (ctx->solve_deferred[g.gcur]) {
    // Lam_trial formed and factorised (log-det + marginals) and the parked gradient solve, side by side in three launches
    const size_t T = ctx->T, Tnn = T * nn_(ctx);
    double* eg = g.exch0[g.gcur].d();
    const double* V = eg + Tn;
    ctx->mix.VD = V; ctx->mix.VU = V + Tnn; ctx->mix.outD = g.Lam[t].d(); ctx->mix.outU = g.Lam[t].d() + Tnn; ctx->mix.step = step;
    ChainWs w0, w1;
    ctx->chain_ws = 0;
    gvi_status fs = ensure_chain_ws(ctx, w0);
    ctx->chain_ws = 1;
    if (fs == GVI_OK) fs = ensure_chain_ws(ctx, w1);
    ctx->chain_ws = 0;
    if (fs == GVI_OK) {
      double* sD = g.Sig[t].d();
      ChainArgs a0 = make_chain_args(ctx, w0, g.Lam[c].d(), g.Lam[c].d() + Tnn, nullptr, 1.0, true, sD, sD + Tnn, nullptr, g.hld[t].d(), true);
      ChainArgs a1 = make_chain_args(ctx, w1, V, V + Tnn, eg, -1.0, false, nullptr, nullptr, g.dmu2[g.gcur].d(), nullptr, false);
      if (ctx->asm_pending[g.gcur]) {
        // the assemble of this gradient buffer was left to these launches: both operations form V from the per-factor
        // results while they load, the solve leaves [g | V_D | V_U] in the buffer (later trials of this iteration read it)
        ctx->asm_pending[g.gcur] = false;
        a0.asm_on = a1.asm_on = 1;
        a1.asmG = eg; a1.asmD = eg + Tn; a1.asmU = eg + Tn + Tnn;
        const AsmList AL = make_asm_list(ctx);
        fs = run_chain(ctx, a0, a1, true, true, &AL);
      } else fs = run_chain(ctx, a0, a1, true, true);
    }
    ctx->mix = gvi_ctx::Mix();
    GVICK(fs);
    ctx->solve_deferred[g.gcur] = false;
    ctx->defer_gather = true;               // mu_trial and the gather ride in the prep launch of the cost pass
    const gvi_status gs = ngd_refresh_gather(ctx, t, g.mu[c].d(), g.dmu2[g.gcur].d(), step);
    ctx->defer_gather = false;
    GVICK(gs);
  } else if (ctx->solve_pending[g.gcur]) {
    // precision part first (needs no dmu), factorise, then join the side-stream solve and form mu_trial
    if (chain_supported(ctx->n) && ctx->T > 1) {
      // Lam_trial = Lam + step (V - Lam) is formed by the first BCR pass while it loads the chain (and written to
      // Lam[t] there): factorise "Lam[c] mixed with V" instead of launching trial_kernel first
      const size_t Tnn = (size_t)ctx->T * nn_(ctx);
      const double* V = g.exch0[g.gcur].d() + Tn;
      ctx->mix.VD = V; ctx->mix.VU = V + Tnn; ctx->mix.outD = g.Lam[t].d(); ctx->mix.outU = g.Lam[t].d() + Tnn; ctx->mix.step = step;
      double* sD = g.Sig[t].d();
      const gvi_status fs = run_bt_factor(ctx, g.Lam[c].d(), g.Lam[c].d() + Tnn, sD, sD + Tnn, g.hld[t].d());
      ctx->mix = gvi_ctx::Mix();
      GVICK(fs);
    } else {
      hipLaunchKernelGGL(trial_kernel, dim3((unsigned)((bt + 255) / 256)), dim3(256), 0, ctx->stream, (int64_t)0, (int64_t)bt,
                         step, g.mu[c].d(), g.dmu2[g.gcur].d(), g.Lam[c].d(), g.exch0[g.gcur].d() + Tn, g.mu[t].d(), g.Lam[t].d());
      HIPCK(ctx, hipGetLastError());
      GVICK(ngd_refresh_factor(ctx, t));
    }
    GVICK(ngd_join_solve(ctx, g.gcur));
    ctx->defer_gather = true;               // mu_trial and the gather ride in the prep launch of the cost pass
    const gvi_status gs = ngd_refresh_gather(ctx, t, g.mu[c].d(), g.dmu2[g.gcur].d(), step);
    ctx->defer_gather = false;
    GVICK(gs);
  } else {
    hipLaunchKernelGGL(trial_kernel, dim3((unsigned)((Tn + bt + 255) / 256)), dim3(256), 0, ctx->stream, (int64_t)Tn,
                       (int64_t)bt, step, g.mu[c].d(), g.dmu2[g.gcur].d(), g.Lam[c].d(), g.exch0[g.gcur].d() + Tn,
                       g.mu[t].d(), g.Lam[t].d());
    HIPCK(ctx, hipGetLastError());
    GVICK(ngd_refresh(ctx, t));
  }
  g.have_trial = true;
  return GVI_OK;
}

gvi_status gvi_ngd_trial_local(gvi_ctx* ctx, double step) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(ngd_trial_state(ctx, step));
  return ngd_cost_local(ctx, 1 - ctx->ngd.cur);
}

gvi_status gvi_ngd_trial_finish(gvi_ctx* ctx, double* new_cost) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  HIPCK(ctx, hipSetDevice(ctx->device));
  return ngd_cost_finish(ctx, 1 - ctx->ngd.cur, new_cost);
}

gvi_status gvi_ngd_trial(gvi_ctx* ctx, double step, double* new_cost) {
  GVICK(gvi_ngd_trial_local(ctx, step));
  GVICK(dist_exchange1(ctx));
  return gvi_ngd_trial_finish(ctx, new_cost);
}

// ---- split / speculative forms for the sharded driver: publish the (all-reduced) trial cost without waiting, queue the
// NEXT iteration's gradients at the trial state into the other gradient buffer behind it, then wait and decide ----
gvi_status gvi_ngd_trial_publish(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  HIPCK(ctx, hipSetDevice(ctx->device));
  return ngd_cost_publish(ctx, 1 - ctx->ngd.cur);
}

gvi_status gvi_ngd_trial_wait(gvi_ctx* ctx, double* new_cost) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  return ngd_cost_wait(ctx, 1 - ctx->ngd.cur, new_cost);
}

gvi_status gvi_ngd_spec_gradients_local(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  HIPCK(ctx, hipSetDevice(ctx->device));
  ctx->ngd.spec_ready = false;
  return ngd_grad_local(ctx, 1 - ctx->ngd.cur, 1 - ctx->ngd.gcur);
}

gvi_status gvi_ngd_spec_gradients_finish(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(ngd_grad_finish(ctx, 1 - ctx->ngd.gcur));
  ctx->ngd.spec_ready = true;
  return GVI_OK;
}

gvi_status gvi_ngd_accept_spec(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.spec_ready) return fail(ctx, GVI_ERR_STATE, "no speculative gradients at the trial state");
  GVICK(gvi_ngd_accept(ctx));
  NgdState& g = ctx->ngd;
  g.gcur = 1 - g.gcur; g.grad_valid = true; g.grad_slot = g.cur; g.spec_ready = false;
  return GVI_OK;
}

gvi_status gvi_ngd_accept(gvi_ctx* ctx) {
  GVICK(ngd_check(ctx));
  if (!ctx->ngd.have_trial) return fail(ctx, GVI_ERR_STATE, "no trial pending");
  GVICK(ngd_flush_gather(ctx, 1 - ctx->ngd.cur));      // no-op once the trial's cost pass has run
  ctx->ngd.cur = 1 - ctx->ngd.cur;
  ctx->ngd.have_trial = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

// Cost and gradients at the current state for the start of an iteration.  When NEITHER is there yet (first iteration after
// gvi_ngd_init) one full moments pass serves both -- its m0 column is the cost, exactly as in a fused trial -- instead of a
// cost-only pass followed by the gradient pass.
static gvi_status ngd_iteration_entry(gvi_ctx* ctx, double* c0) {
  NgdState& g = ctx->ngd;
  const bool have_grad = g.grad_valid && g.grad_slot == g.cur;
  if (!g.cost_valid[g.cur] && !have_grad && ctx->fuse_trial != 0 && !dist_on(ctx) && !ctx->sets.empty()) {
    ctx->solve_deferred[g.gcur] = false;
    GVICK(ngd_moments_full(ctx, g.cur, g.cur));          // epilogue + ordered cost sum + publish in one launch
    GVICK(ngd_scatter(ctx, g.cur, g.gcur));
    GVICK(ngd_grad_finish(ctx, g.gcur));
    g.grad_valid = true; g.grad_slot = g.cur;
    return ngd_cost_wait(ctx, g.cur, c0);
  }
  GVICK(gvi_ngd_cost(ctx, c0));
  if (!have_grad) GVICK(gvi_ngd_gradients(ctx));         // else: computed speculatively
  return GVI_OK;
}

// The backtracking loop of one iteration from trial number `cnt` on (step = the step of the LAST trial taken, or the base
// before the first): step *= 0.75 per trial, first decrease accepted, give up after max_backtrack + 1 trials.
static gvi_status ngd_linesearch(gvi_ctx* ctx, double c0, double step, int cnt, int max_backtrack, double* c1_out, int* ok_out,
                                 int* cnt_out) {
  NgdState& g = ctx->ngd;
  double c1 = c0;
  int ok = 0;
  while (true) {
    step *= 0.75;                                  // gvibase/GVI-GH-impl.h:83
    const int t = 1 - g.cur;
    const bool spec = cnt == 0 && ctx->speculate;
    GVICK(ngd_trial_state(ctx, step));
    const bool fuse = spec && (ctx->fuse_trial == 1 || (ctx->fuse_trial == 2 && ctx->last_first_accepted));
    if (fuse) {
      // Fused form: ONE full moments pass at the trial point serves both the trial cost (its m0 column)
      // and -- if the trial is accepted -- the next iteration's gradients.  Same numbers, one psi pass
      // less per accepted iteration; a rejected first trial wasted the moment accumulation.
      if (dist_on(ctx)) {
        // sharded: the local cost sum and the local [g | D | U] go through the two exchanges before publish / solve
        // (ONE all-gather: the partial cost sum rides with the gradient records; the publish follows it)
        GVICK(ngd_moments_full(ctx, t));
        GVICK(dist_ranges(ctx));
        hipLaunchKernelGGL(cost_sum_all_kernel, dim3(1), dim3(256), 0, ctx->stream, make_set_list(ctx, t), g.exch1.d(),
                           ctx->dist.send.d() + (size_t)ctx->dist.maxlen * (ctx->n + 2 * nn_(ctx)));
        HIPCK(ctx, hipGetLastError());
        GVICK(ngd_scatter(ctx, t, 1 - g.gcur));               // also writes this rank's records
        GVICK(dist_exchange0(ctx, 1 - g.gcur, true, t));      // all-gather, fold, ordered cost total, publish
        GVICK(ngd_grad_finish(ctx, 1 - g.gcur));
      } else if (ctx->sets.empty()) {
        HIPCK(ctx, hipMemsetAsync(g.exch1.p, 0, 8, ctx->stream));
        GVICK(ngd_cost_publish(ctx, t));
      } else {
        GVICK(ngd_moments_full(ctx, t, t));           // epilogue + ordered cost sum + publish in one launch
      }
      if (!dist_on(ctx)) {
        GVICK(ngd_scatter(ctx, t, 1 - g.gcur));
        GVICK(ngd_grad_finish(ctx, 1 - g.gcur));
      }
    } else {
      if (dist_on(ctx)) {
        GVICK(ngd_cost_local(ctx, t, false));
        GVICK(dist_exchange1(ctx));
        GVICK(ngd_cost_publish(ctx, t));
      } else {
        GVICK(ngd_cost_local(ctx, t, true));        // cost tail publishes in the same launch
      }
      // Speculation: the first trial is accepted in the common case, and then the next iteration starts
      // with the gradients at exactly this trial state.  Queue them BEHIND the publish, into the other
      // gradient buffer, so the device never idles while the host reads the cost and decides.  A rejected
      // trial just leaves that buffer unused (same numbers either way).
      if (spec) {
        GVICK(ngd_grad_local(ctx, t, 1 - g.gcur));
        GVICK(dist_exchange0(ctx, 1 - g.gcur));
        GVICK(ngd_grad_finish(ctx, 1 - g.gcur));
      }
    }
    GVICK(ngd_cost_wait(ctx, t, &c1));
    ++cnt;
    if (cnt == 1) ctx->last_first_accepted = c1 < c0;
    if (c1 < c0) {                                 // NaN compares false -> rejected
      GVICK(gvi_ngd_accept(ctx));
      if (spec) { g.gcur = 1 - g.gcur; g.grad_valid = true; g.grad_slot = g.cur; }
      ok = 1;
      break;
    }
    if (cnt > max_backtrack) break;
  }
  *c1_out = c1; *ok_out = ok; *cnt_out = cnt;
  return GVI_OK;
}

gvi_status gvi_ngd_step(gvi_ctx* ctx, double step_size_base, int max_backtrack, double* cost_iter, int* accepted,
                        double* new_cost, int* ntrials) {
  GVICK(ngd_check(ctx));
  if (ctx->update_rule != GVI_RULE_NGD) return fail(ctx, GVI_ERR_STATE, "proximal rule selected: use gvi_prox_step");
  HIPCK(ctx, hipSetDevice(ctx->device));
  double c0 = 0.0;
  GVICK(ngd_iteration_entry(ctx, &c0));
  if (cost_iter) *cost_iter = c0;
  double c1 = c0;
  int cnt = 0, ok = 0;
  GVICK(ngd_linesearch(ctx, c0, step_size_base, 0, max_backtrack, &c1, &ok, &cnt));
  if (accepted) *accepted = ok;
  if (new_cost) *new_cost = ok ? c1 : c0;
  if (ntrials) *ntrials = cnt;
  return GVI_OK;
}

// ---- gvi_ngd_run: the iteration loop, pipelined one iteration deep ----
// gvi_ngd_step queues an iteration, waits for its trial cost on the host, decides, and only then queues the next one: the
// device idles for the host's wake-up + launch latency (4-7 us of a 0.14 ms iteration in the kernel trace).  Here the launches
// of iteration i + 1 are queued BEFORE the host has read the cost of iteration i, assuming its first trial is accepted (it
// is, in the steady state); they are PREDICATED on a device word that the tail of iteration i sets only if its cost
// decreased (same comparison, same doubles as the host's), so a rejected trial turns them into no-ops, the host rolls its
// bookkeeping back and continues that iteration's backtracking exactly as gvi_ngd_step would.  Same numbers either way
// (tests/test_gpu_parity.py::test_ngd_run_equals_the_same_sequence_of_steps, with rejected first trials).
namespace {
struct PipeSnapshot {
  int cur, gcur, grad_slot;
  bool grad_valid, have_trial, spec_ready, cost_valid[2], solve_deferred[2], solve_pending[2], asm_pending[2], last_first_accepted;
  double cost[2];
  NgdState::GatherPending gpend[2];
  int64_t n_full, n_cost;
  long profile_count;
  std::vector<int> prep_slot, warm_count;
};

void pipe_save(const gvi_ctx* c, PipeSnapshot& p) {
  const NgdState& g = c->ngd;
  p.cur = g.cur; p.gcur = g.gcur; p.grad_slot = g.grad_slot; p.grad_valid = g.grad_valid; p.have_trial = g.have_trial;
  p.spec_ready = g.spec_ready; p.last_first_accepted = c->last_first_accepted;
  for (int i = 0; i < 2; ++i) {
    p.cost_valid[i] = g.cost_valid[i]; p.cost[i] = g.cost[i]; p.gpend[i] = g.gpend[i];
    p.solve_deferred[i] = c->solve_deferred[i]; p.solve_pending[i] = c->solve_pending[i]; p.asm_pending[i] = c->asm_pending[i];
  }
  p.n_full = c->n_full_pass; p.n_cost = c->n_cost_pass; p.profile_count = c->profile_count;
  p.prep_slot.clear(); p.warm_count.clear();
  for (const auto& s : c->sets) { p.prep_slot.push_back(s->prep_slot); p.warm_count.push_back(s->warm_count); }
}

void pipe_restore(gvi_ctx* c, const PipeSnapshot& p) {
  NgdState& g = c->ngd;
  g.cur = p.cur; g.gcur = p.gcur; g.grad_slot = p.grad_slot; g.grad_valid = p.grad_valid; g.have_trial = p.have_trial;
  g.spec_ready = p.spec_ready; c->last_first_accepted = p.last_first_accepted;
  for (int i = 0; i < 2; ++i) {
    g.cost_valid[i] = p.cost_valid[i]; g.cost[i] = p.cost[i]; g.gpend[i] = p.gpend[i];
    c->solve_deferred[i] = p.solve_deferred[i]; c->solve_pending[i] = p.solve_pending[i]; c->asm_pending[i] = p.asm_pending[i];
  }
  c->n_full_pass = p.n_full; c->n_cost_pass = p.n_cost; c->profile_count = p.profile_count;
  for (size_t i = 0; i < c->sets.size(); ++i) { c->sets[i]->prep_slot = p.prep_slot[i]; c->sets[i]->warm_count = p.warm_count[i]; }
}

// every launch of a queued iteration must be one of the predicated kernels: the dual chain launches, the fused factor pass or
// the prep launch with the fused gather + the sets' moments launches (every MomArgs / OrbitArgs kernel) + the epilogue with its
// tail, the assemble
bool pipe_ok(const gvi_ctx* c) {
  if (!c->pipeline || dist_on(c) || c->update_rule != GVI_RULE_NGD || !c->speculate || c->fuse_trial == 0) return false;
  if (!(c->dual_chain && c->side_solve && chain_supported(c->n) && c->T > 1)) return false;
  if (!c->fuse_gather || !c->pair_fuse || c->profile_all || c->sets.empty() || (int)c->sets.size() > MAX_SETS) return false;
  for (const auto& s : c->sets)
    if (s->K <= 0 || s->kind == KIND_HOST_CALLBACK) return false;       // (host-evaluated psi: the host is in the loop anyway)
  return true;
}
}  // namespace

// one iteration's launches with a fused first trial (the non-sharded branch of ngd_linesearch, first trial)
static gvi_status pipe_enqueue(gvi_ctx* ctx, double step) {
  NgdState& g = ctx->ngd;
  const int t = 1 - g.cur;
  GVICK(ngd_trial_state(ctx, step));
  GVICK(ngd_moments_full(ctx, t, t));
  GVICK(ngd_scatter(ctx, t, 1 - g.gcur));
  return ngd_grad_finish(ctx, 1 - g.gcur);
}

gvi_status gvi_ngd_run(gvi_ctx* ctx, int max_iters, double step_size_base, int max_backtrack, double* cost_iter,
                       int* accepted, double* new_cost, int* ntrials, int* iters_done) {
  GVICK(ngd_check(ctx));
  if (max_iters < 0) return fail(ctx, GVI_ERR_ARG, "max_iters < 0");
  if (ctx->update_rule != GVI_RULE_NGD) return fail(ctx, GVI_ERR_STATE, "proximal rule selected: use gvi_prox_step");
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  int done = 0;
  auto record = [&](double c0, int ok, double c1, int nt) {
    if (cost_iter) cost_iter[done] = c0;
    if (accepted) accepted[done] = ok;
    if (new_cost) new_cost[done] = c1;
    if (ntrials) ntrials[done] = nt;
    ++done;
  };
  auto finish = [&](gvi_status st) {
    ctx->cur_pred = nullptr; ctx->pipe_tail = false; ctx->pipe_c0_imm = true; ctx->pub_ring = 0;
    if (iters_done) *iters_done = done;
    return st;
  };
  bool pending = false;          // the launches of iteration `done` are already queued (speculatively; their predicate held)
  double pend_seq = 0.0;
  int pend_ring = 0;
  const double step1 = step_size_base * 0.75;
  while (done < max_iters) {
    const bool fused_first = ctx->fuse_trial == 1 || (ctx->fuse_trial == 2 && ctx->last_first_accepted);
    if (!pending && !(pipe_ok(ctx) && fused_first)) {           // the plain iteration
      double c0 = 0.0, c1 = 0.0;
      int ok = 0, nt = 0;
      const gvi_status st = gvi_ngd_step(ctx, step_size_base, max_backtrack, &c0, &ok, &c1, &nt);
      if (st != GVI_OK) return finish(st);
      record(c0, ok, c1, nt);
      if (!ok) break;
      continue;
    }
    double c0 = 0.0, seq_i = 0.0;
    int ring_i = 0;
    if (!pending) {
      gvi_status st = ngd_iteration_entry(ctx, &c0);
      if (st != GVI_OK) return finish(st);
      if (!ctx->solve_deferred[g.gcur]) {                       // not in the dual-launch state: one plain iteration
        double c1 = 0.0;
        int ok = 0, nt = 0;
        st = gvi_ngd_step(ctx, step_size_base, max_backtrack, &c0, &ok, &c1, &nt);
        if (st != GVI_OK) return finish(st);
        record(c0, ok, c1, nt);
        if (!ok) break;
        continue;
      }
      if (!ctx->pipe_dev.p) {
        HIPCK(ctx, ctx->pipe_dev.ensure(64));
        HIPCK(ctx, hipMemsetAsync(ctx->pipe_dev.p, 0, 64, ctx->stream));
      }
      ctx->cur_pred = nullptr; ctx->pipe_tail = true; ctx->pipe_c0_imm = true; ctx->pipe_c0 = c0; ctx->pub_ring = 0;
      st = pipe_enqueue(ctx, step1);
      ctx->pipe_tail = false;
      if (st != GVI_OK) return finish(st);
      seq_i = ctx->seq; ring_i = 0;
    } else {
      c0 = g.cost[g.cur]; seq_i = pend_seq; ring_i = pend_ring;
    }
    // queue iteration i + 1 behind it, predicated on the acceptance of trial i
    const bool spec_next = done + 1 < max_iters;
    PipeSnapshot snap;
    double seq_n = 0.0;
    if (spec_next) {
      pipe_save(ctx, snap);
      gvi_status st = gvi_ngd_accept(ctx);                      // provisional: cur <- trial slot
      if (st != GVI_OK) { pipe_restore(ctx, snap); return finish(st); }
      g.gcur = 1 - g.gcur; g.grad_valid = true; g.grad_slot = g.cur;
      ctx->last_first_accepted = true;
      // (the accept words alternate like the host slots: this iteration's own tail writes the OTHER word, so its assemble,
      // which runs after that tail, still sees the predicate it was queued under)
      ctx->cur_pred = ctx->pipe_dev.d() + ring_i; ctx->cur_pred_val = seq_i;
      ctx->pipe_tail = true; ctx->pipe_c0_imm = false; ctx->pub_ring = 1 - ring_i;
      st = pipe_enqueue(ctx, step1);
      ctx->cur_pred = nullptr; ctx->pipe_tail = false; ctx->pipe_c0_imm = true; ctx->pub_ring = 0;
      if (st != GVI_OK) { pipe_restore(ctx, snap); return finish(st); }   // undo the provisional accept (cur / gcur flipped)
      seq_n = ctx->seq;
    }
    const int ti = spec_next ? g.cur : 1 - g.cur;                // NGD slot of trial i
    double c1 = 0.0;
    {
      const gvi_status st = ngd_cost_wait(ctx, ti, &c1, seq_i, ring_i);
      if (st != GVI_OK) { if (spec_next) pipe_restore(ctx, snap); return finish(st); }
    }
    if (c1 < c0) {                                               // accepted (NaN compares false)
      ctx->last_first_accepted = true;
      if (!spec_next) {
        const gvi_status st = gvi_ngd_accept(ctx);
        if (st != GVI_OK) return finish(st);
        g.gcur = 1 - g.gcur; g.grad_valid = true; g.grad_slot = g.cur;
      }
      record(c0, 1, c1, 1);
      pending = spec_next; pend_seq = seq_n; pend_ring = 1 - ring_i;
      continue;
    }
    // first trial rejected: the device skipped everything queued for iteration i + 1
    if (spec_next) pipe_restore(ctx, snap);
    pending = false;
    ctx->last_first_accepted = false;
    double c1b = c0;
    int ok = 0, cnt = 1;
    if (cnt <= max_backtrack) {
      const gvi_status st = ngd_linesearch(ctx, c0, step1, 1, max_backtrack, &c1b, &ok, &cnt);
      if (st != GVI_OK) return finish(st);
    }
    record(c0, ok, ok ? c1b : c0, cnt);
    if (!ok) break;
  }
  return finish(GVI_OK);
}

gvi_status gvi_ngd_set_update_rule(gvi_ctx* ctx, int rule) {
  if (!ctx || (rule != GVI_RULE_NGD && rule != GVI_RULE_PROX_JKO)) return GVI_ERR_ARG;
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  ctx->update_rule = rule;
  for (auto& s : ctx->sets) { s->unit_temperature = rule == GVI_RULE_PROX_JKO; s->prep_slot = -1; }
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

// factor-level JKO increments at step h for every set (moments at the current proposal), assembled into exch0[gcur]:
// g = joint dmu, [D | U] = joint dprecision (proxgd/ProxGVI-GH-impl.h:43-60: plain sums, no solve)
gvi_status gvi_prox_gradients(gvi_ctx* ctx, double h) {
  GVICK(ngd_check(ctx));
  if (ctx->update_rule != GVI_RULE_PROX_JKO) return fail(ctx, GVI_ERR_STATE, "call gvi_ngd_set_update_rule(GVI_RULE_PROX_JKO) first");
  if (!(h > 0.0)) return fail(ctx, GVI_ERR_ARG, "step must be positive");
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  g.grad_valid = false;
  GVICK(ngd_flush_gather(ctx, g.cur));
  GVICK(ngd_moments_full(ctx, g.cur));                    // Vdmu = b, Vddmu = S at unit temperature; f.Lam = Lam_k
  for (auto& sp : ctx->sets) {
    FactorSet& s = *sp;
    if (s.K == 0) continue;
    const size_t K = s.K, d = s.d, dd = d * d;
    HIPCK(ctx, s.jko_half.ensure(K * dd * 8));
    HIPCK(ctx, s.jko_S.ensure(K * dd * 8));
    HIPCK(ctx, s.jko_Sinv.ensure(K * dd * 8));
    HIPCK(ctx, s.jko_Lam.ensure(K * dd * 8));
    JkoArgs a;
    a.K = s.K; a.d = s.d; a.h = h; a.Vdmu = s.Vdmu.d(); a.Vddmu = s.Vddmu.d(); a.Sigma = s.Sigma_k[g.cur].d();
    a.Lam = s.Lam.d(); a.Shalf = s.jko_half.d(); a.LamNew = s.jko_Lam.d();
    hipLaunchKernelGGL(jko_half_kernel, dim3(s.K), dim3(64), 3 * dd * 8, ctx->stream, a);
    FactorDev f = s.dev();                                 // spectral map of Sig_half into scratch (no psi operands)
    f.m = 0; f.S = s.jko_S.d(); f.Sinv = s.jko_Sinv.d(); f.Lam = s.jko_Lam.d(); f.H = nullptr; f.Hq = nullptr; f.u0 = nullptr;
    f.jko_h = h;
    f.chol = 0;                                            // the JKO map is a function of the eigenvalues
    const int dp = s.d + (s.d & 1);
    const size_t lds = (size_t)(4 * dd + 2 * dp + 3 * d) * 8 + (size_t)dp * 4 + 16;
    if (s.d <= 8) hipLaunchKernelGGL(prep_kernel<1>, dim3(s.K), dim3(64), lds, ctx->stream, f, (const double*)nullptr, (const double*)s.jko_half.d());
    else if (s.d <= 16) hipLaunchKernelGGL(prep_kernel<4>, dim3(s.K), dim3(64), lds, ctx->stream, f, (const double*)nullptr, (const double*)s.jko_half.d());
    else if (s.d <= 32) hipLaunchKernelGGL(prep_kernel<16>, dim3(s.K), dim3(64), lds, ctx->stream, f, (const double*)nullptr, (const double*)s.jko_half.d());
    else return fail(ctx, GVI_ERR_UNSUPPORTED, "factor dimension > 32");
    hipLaunchKernelGGL(jko_finish_kernel, dim3((unsigned)((K * dd + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCK(ctx, hipGetLastError());
  }
  GVICK(ngd_scatter(ctx, g.cur, g.gcur));
  g.grad_valid = true;
  g.grad_slot = g.cur;
  return GVI_OK;
}

// trial of the proximal rule: mu + step dmu, Lam + step dprecision (proxgd/ProxGVI-GH-impl.h:24-41)
gvi_status gvi_prox_trial(gvi_ctx* ctx, double step, double* new_cost) {
  GVICK(ngd_check(ctx));
  if (ctx->update_rule != GVI_RULE_PROX_JKO) return fail(ctx, GVI_ERR_STATE, "call gvi_ngd_set_update_rule(GVI_RULE_PROX_JKO) first");
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  if (!(g.grad_valid && g.grad_slot == g.cur)) return fail(ctx, GVI_ERR_STATE, "call gvi_prox_gradients first");
  const size_t Tn = (size_t)ctx->T * ctx->n, bt = bt_count(ctx);
  const int c = g.cur, t = 1 - c;
  const double* ex = g.exch0[g.gcur].d();
  hipLaunchKernelGGL(trial_kernel, dim3((unsigned)((Tn + bt + 255) / 256)), dim3(256), 0, ctx->stream, (int64_t)Tn, (int64_t)bt,
                     step, g.mu[c].d(), ex, g.Lam[c].d(), ex + Tn, g.mu[t].d(), g.Lam[t].d(), 1);
  HIPCK(ctx, hipGetLastError());
  g.cost_valid[t] = false;
  GVICK(ngd_refresh(ctx, t));
  g.have_trial = true;
  GVICK(ngd_cost_local(ctx, t));
  return ngd_cost_finish(ctx, t, new_cost);
}

// ProxGVIGH::optimize body (proxgd/ProxGVI-GH-impl.h:121-202): gradients once at step = base, trial B uses base^B,
// the first decreasing trial is accepted; after max_backtrack failures the last trial is accepted anyway (:177-184)
gvi_status gvi_prox_step(gvi_ctx* ctx, double step_size_base, int max_backtrack, double* cost_iter, int* decreased,
                         double* new_cost, int* ntrials) {
  GVICK(ngd_check(ctx));
  if (ctx->update_rule != GVI_RULE_PROX_JKO) return fail(ctx, GVI_ERR_STATE, "call gvi_ngd_set_update_rule(GVI_RULE_PROX_JKO) first");
  double c0 = 0.0;
  GVICK(gvi_ngd_cost(ctx, &c0));
  if (cost_iter) *cost_iter = c0;
  GVICK(gvi_prox_gradients(ctx, step_size_base));
  int B = 1, cnt = 0, ok = 0;
  double c1 = c0;
  while (true) {
    GVICK(gvi_prox_trial(ctx, std::pow(step_size_base, B), &c1));
    ok = c1 < c0;
    if (!ok) { ++B; ++cnt; }
    if (ok || cnt > max_backtrack) { GVICK(gvi_ngd_accept(ctx)); break; }
  }
  if (decreased) *decreased = ok;
  if (new_cost) *new_cost = c1;
  if (ntrials) *ntrials = cnt + (ok ? 1 : 0);
  return GVI_OK;
}

gvi_status gvi_ngd_set_mode(gvi_ctx* ctx, int speculate, int fuse_trial) {
  if (!ctx) return GVI_ERR_ARG;
  if (fuse_trial < 0 || fuse_trial > 2) return fail(ctx, GVI_ERR_ARG, "fuse_trial must be 0, 1 or 2");
  ctx->speculate = speculate != 0;
  ctx->fuse_trial = fuse_trial;
  ctx->last_first_accepted = true;
  return GVI_OK;
}

// ---- transports of the exchange ----
namespace {
struct RcclApi {
  struct Id128 { char b[128]; };       // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
};
bool load_rccl(RcclApi& r, std::string& err) {
  // GVI_RCCL_PATH, when set, names the ONLY candidate (an explicit choice is not second-guessed); else the usual names
  const char* forced = getenv("GVI_RCCL_PATH");
  const bool only = forced && *forced;
  const char* names[] = {forced, only ? nullptr : "librccl.so.1", only ? nullptr : "librccl.so", only ? nullptr : "/opt/rocm/lib/librccl.so.1"};
  std::string why;                                   // dlerror() clears the message it returns: read it ONCE per failed dlopen
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
    const char* e = dlerror();
    why = e ? e : "?";
  }
  if (!r.lib) { err = "cannot load librccl: " + (why.empty() ? std::string("no candidate name") : why); return false; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { err = "librccl lacks an expected symbol"; return false; }
  return true;
}
}  // namespace

gvi_status gvi_dist_unique_id(void* id128) {
  if (!id128) return fail(nullptr, GVI_ERR_ARG, "id128 is NULL");
  RcclApi r;
  std::string err;
  if (!load_rccl(r, err)) return fail(nullptr, GVI_ERR_HIP, err);
  const int rc = r.GetUniqueId(id128);
  return rc == 0 ? GVI_OK : fail(nullptr, GVI_ERR_HIP, "ncclGetUniqueId failed with code " + std::to_string(rc));
}

gvi_status gvi_dist_init_rccl(gvi_ctx* ctx, int rank, int world, const void* id128) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, GVI_ERR_ARG, "bad rank / world / id");
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  RcclApi r;
  std::string err;
  if (!load_rccl(r, err)) return fail(ctx, GVI_ERR_HIP, err);
  RcclApi::Id128 id;
  memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, world, id, rank);
  if (rc != 0 || !comm) return fail(ctx, GVI_ERR_HIP, "ncclCommInitRank failed with code " + std::to_string(rc));
  gvi_ctx::Dist& d = ctx->dist;
  if (d.comm && d.ncclCommDestroy) d.ncclCommDestroy(d.comm);
  d.rank = rank; d.world = world; d.fn = nullptr; d.user = nullptr;
  d.rccl_lib = r.lib; d.comm = comm; d.ncclAllGather = r.AllGather; d.ncclCommDestroy = r.CommDestroy;
  d.ranges_valid = false;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_dist_init_callback(gvi_ctx* ctx, int rank, int world, gvi_allgather_fn fn, void* user) {
  if (!ctx || !fn || world < 1 || rank < 0 || rank >= world) return fail(ctx, GVI_ERR_ARG, "bad rank / world / callback");
  GVICK(sync(ctx));
  gvi_ctx::Dist& d = ctx->dist;
  if (d.comm && d.ncclCommDestroy) { d.ncclCommDestroy(d.comm); d.comm = nullptr; }
  d.rank = rank; d.world = world; d.fn = fn; d.user = user;
  d.ranges_valid = false;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  return GVI_OK;
}

gvi_status gvi_dist_info(const gvi_ctx* ctx, int* rank, int* world, int* records_per_rank) {
  if (!ctx) return GVI_ERR_ARG;
  if (rank) *rank = ctx->dist.rank;
  if (world) *world = ctx->dist.world;
  if (records_per_rank) *records_per_rank = ctx->dist.ranges_valid ? ctx->dist.maxlen : 0;
  return GVI_OK;
}

gvi_status gvi_ngd_counters(gvi_ctx* ctx, int64_t* full_passes, int64_t* cost_passes, int reset) {
  if (!ctx) return GVI_ERR_ARG;
  if (full_passes) *full_passes = ctx->n_full_pass;
  if (cost_passes) *cost_passes = ctx->n_cost_pass;
  if (reset) ctx->n_full_pass = ctx->n_cost_pass = 0;
  return GVI_OK;
}

gvi_status gvi_ngd_exchange(gvi_ctx* ctx, int which, void** dev_ptr, int64_t* count) {
  GVICK(ngd_check(ctx));
  if (!dev_ptr || !count) return fail(ctx, GVI_ERR_ARG, "NULL argument");
  GVICK(ngd_flush_assemble(ctx, 0));                    // the caller is going to read / reduce the buffers themselves
  GVICK(ngd_flush_assemble(ctx, 1));
  if (which == 0) { *dev_ptr = ctx->ngd.exch0[ctx->ngd.gcur].p; *count = (int64_t)((size_t)ctx->T * ctx->n + bt_count(ctx)); }
  else if (which == 1) { *dev_ptr = ctx->ngd.exch1.p; *count = 1; }
  else if (which == 2) { *dev_ptr = ctx->ngd.exch0[1 - ctx->ngd.gcur].p; *count = (int64_t)((size_t)ctx->T * ctx->n + bt_count(ctx)); }
  else return fail(ctx, GVI_ERR_ARG, "which must be 0, 1 or 2");
  return GVI_OK;
}

gvi_status gvi_ngd_get_state(gvi_ctx* ctx, double* mu, double* D, double* U, double* SigD, double* SigU) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, n = ctx->n, nn = n * n;
  const int i = g.cur;
  if (mu) GVICK(d2h(ctx, mu, g.mu[i].p, T * n * 8));
  if (D) GVICK(d2h(ctx, D, g.Lam[i].p, T * nn * 8));
  if (U && T > 1) GVICK(d2h(ctx, U, g.Lam[i].d() + T * nn, (T - 1) * nn * 8));
  if (SigD) GVICK(d2h(ctx, SigD, g.Sig[i].p, T * nn * 8));
  if (SigU && T > 1) GVICK(d2h(ctx, SigU, g.Sig[i].d() + T * nn, (T - 1) * nn * 8));
  return sync(ctx);
}

gvi_status gvi_ngd_get_gradients(gvi_ctx* ctx, double* dmu, double* dD, double* dU, double* gq, double* VD, double* VU) {
  GVICK(ngd_check(ctx));
  HIPCK(ctx, hipSetDevice(ctx->device));
  NgdState& g = ctx->ngd;
  const size_t T = ctx->T, n = ctx->n, nn = n * n, bt = bt_count(ctx);
  if (!(g.grad_valid)) return fail(ctx, GVI_ERR_STATE, "no gradients computed for the current proposal");
  GVICK(ngd_join_solve(ctx, g.gcur));
  if (dmu) GVICK(d2h(ctx, dmu, g.dmu2[g.gcur].p, T * n * 8));
  if (gq) GVICK(d2h(ctx, gq, g.exch0[g.gcur].p, T * n * 8));
  std::vector<double> V, L;
  if (dD || dU || VD || VU) {
    V.resize(bt);
    GVICK(d2h(ctx, V.data(), g.exch0[g.gcur].d() + T * n, bt * 8));
  }
  if (dD || dU) {
    L.resize(bt);
    GVICK(d2h(ctx, L.data(), g.Lam[g.cur].p, bt * 8));
  }
  GVICK(sync(ctx));
  if (VD) memcpy(VD, V.data(), T * nn * 8);
  if (VU && T > 1) memcpy(VU, V.data() + T * nn, (T - 1) * nn * 8);
  if (dD) for (size_t j = 0; j < T * nn; ++j) dD[j] = V[j] - L[j];          // dprecision = Vddmu - Lambda
  if (dU) for (size_t j = 0; j < (T - 1) * nn; ++j) dU[j] = V[T * nn + j] - L[T * nn + j];
  return GVI_OK;
}

gvi_status gvi_profile_enable(gvi_ctx* ctx, int on) {
  if (!ctx) return GVI_ERR_ARG;
  ctx->profile = on != 0;
  ctx->profile_all = on == 2;
  ctx->profile_every = on == 3 ? 8 : 1;
  ctx->profile_count = 0;
  for (auto& s : ctx->sets) s->ev_set[0] = s->ev_set[1] = false;
  return GVI_OK;
}

gvi_status gvi_profile_last(gvi_ctx* ctx, int set_id, int what, float* ms) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s || !ms || what < 0 || what > 1) return GVI_ERR_ARG;
  if (!s->ev_set[what]) return fail(ctx, GVI_ERR_STATE, "no profiled launch recorded");
  HIPCK(ctx, hipEventSynchronize(s->ev[what][1]));
  HIPCK(ctx, hipEventElapsedTime(ms, s->ev[what][0], s->ev[what][1]));
  return GVI_OK;
}

gvi_status gvi_profile_stages(gvi_ctx* ctx, int on, float* mean_us, int* counts) {
  if (!ctx) return GVI_ERR_ARG;
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (mean_us || counts) {
    GVICK(sync(ctx));
    double sum[STAGE_COUNT] = {0, 0, 0};
    int cnt[STAGE_COUNT] = {0, 0, 0};
    for (auto& r : ctx->stage_recs) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { sum[r.stage] += ms; ++cnt[r.stage]; }
    }
    for (int i = 0; i < STAGE_COUNT; ++i) {
      if (mean_us) mean_us[i] = cnt[i] ? (float)(1e3 * sum[i] / cnt[i]) : 0.f;
      if (counts) counts[i] = cnt[i];
    }
  }
  for (auto& r : ctx->stage_recs) { ctx->stage_pool.push_back(r.e0); ctx->stage_pool.push_back(r.e1); }
  ctx->stage_recs.clear();
  ctx->stage_prof = on != 0;
  return GVI_OK;
}

gvi_status gvi_profile_geometry(gvi_ctx* ctx, int set_id, int* variant, int* nchunk, int64_t* chunk) {
  FactorSet* s = get_set(ctx, set_id);
  if (!s) return GVI_ERR_ARG;
  if (variant) *variant = s->closed_form ? 0 : (s->use_opsi ? 7 : s->use_orbit ? 6 : (s->fused_pair ? 5 : (s->use_reg ? 2 : (s->use_split ? 3 : 1))));
  if (nchunk) *nchunk = s->nchunk;
  if (chunk) *chunk = s->chunk;
  return GVI_OK;
}

gvi_status gvi_set_option(gvi_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return GVI_ERR_ARG;
  const std::string n(name);
  if (ctx->ngd.ready) { GVICK(ngd_flush_assemble(ctx, 0)); GVICK(ngd_flush_assemble(ctx, 1)); }
  if (n == "split_flush") ctx->split_flush = std::max(0, value);
  else if (n == "sreg_pipe") ctx->sreg_pipe = value != 0;
  else if (n == "mirror") ctx->mirror = value != 0;
  else if (n == "pair_fuse") ctx->pair_fuse = value != 0;
  else if (n == "fuse_gather") ctx->fuse_gather = value != 0;
  else if (n == "side_solve") ctx->side_solve = value != 0;
  else if (n == "dual_chain") ctx->dual_chain = value != 0;
  else if (n == "warm_start") ctx->warm_start = value != 0;
  else if (n == "no_scost") ctx->no_scost = value != 0;
  else if (n == "target_waves") ctx->target_waves = std::max(1, value);
  else if (n == "orbit") ctx->orbit = value != 0;
  else if (n == "fused") ctx->fused = value != 0;
  else if (n == "assemble_on_load") ctx->asm_on_load = value != 0;
  else if (n == "pipeline") ctx->pipeline = value != 0;
  else if (n == "chain_wave") chain_wave_enabled() = value != 0;
  else if (n == "chain_merge") { ctx->chain_merge = value != 0; ctx->chain_merge_fault = value == 2; }
  else if (n == "trust_table_degree") ctx->trust_table_degree = value != 0;
  else if (n == "safe_publish") {
    GVICK(sync(ctx));
    ctx->safe_publish = value != 0;
    for (int q = 0; q < 16; ++q) ctx->host_slot[q] = 0.0;       // the two forms lay the slot out differently
  }
  else if (n == "chol_sqrt") { ctx->chol_sqrt = value != 0; for (auto& s : ctx->sets) s->use_chol = ctx->chol_sqrt; }
  else if (n == "jacobi_tol_exp") {
    // the threshold compares SQUARED off-diagonal mass with squared diagonal mass: anything looser than 1e-20 (1e-10 relative)
    // would break the 1e-9 operator parity, so it is refused rather than silently clamped
    if (value > -20) return fail(ctx, GVI_ERR_ARG, "jacobi_tol_exp must be <= -20 (threshold 10^value on squared magnitudes)");
    ctx->jacobi_tol = std::pow(10.0, (double)value);
    for (auto& s : ctx->sets) s->jtol = ctx->jacobi_tol;
  }
  else if (n == "orbit_waves") ctx->orbit_waves = std::max(1, value);
  else if (n == "orbit_min_tiles") ctx->orbit_min_tiles = std::max(1, value);
  else if (n == "orbit_stack") ctx->orbit_stack = value != 0;
  else if (n == "orbit_copies") ctx->orbit_copies = std::min(16, std::max(1, value));
  else return fail(ctx, GVI_ERR_ARG, "unknown option: " + n);
  for (auto& s : ctx->sets) s->prep_slot = -1;
  ctx->ngd.cost_valid[0] = ctx->ngd.cost_valid[1] = false;
  ctx->ngd.grad_valid = false;
  ctx->ngd.spec_ready = false;
  return GVI_OK;
}


gvi_status gvi_debug_cost_log(gvi_ctx* ctx, int entries, double* out, double* seq_now) {
  if (!ctx) return GVI_ERR_ARG;
  HIPCK(ctx, hipSetDevice(ctx->device));
  GVICK(sync(ctx));
  if (entries > 0) {
    if (entries & (entries - 1)) return fail(ctx, GVI_ERR_ARG, "entries must be a power of two");
    HIPCK(ctx, ctx->dbg_log.ensure((size_t)entries * 8));
    HIPCK(ctx, hipMemset(ctx->dbg_log.p, 0, (size_t)entries * 8));
    ctx->dbg_mask = entries - 1;
    double* lg = ctx->dbg_log.d();
    HIPCK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(gvi_dbg_mask), &ctx->dbg_mask, sizeof(int)));
    HIPCK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(gvi_dbg_log), &lg, sizeof(double*)));
  } else if (entries == 0 && out) {
    if (!ctx->dbg_mask) return fail(ctx, GVI_ERR_STATE, "cost log not enabled");
    HIPCK(ctx, hipMemcpy(out, ctx->dbg_log.p, (size_t)(ctx->dbg_mask + 1) * 8, hipMemcpyDeviceToHost));
  } else {
    ctx->dbg_mask = 0;
    double* lg = nullptr;
    HIPCK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(gvi_dbg_log), &lg, sizeof(double*)));
  }
  if (seq_now) *seq_now = ctx->seq;
  return GVI_OK;
}

gvi_status gvi_set_variant(gvi_ctx* ctx, int variant) {
  if (!ctx || variant < 0 || variant > 7 || variant == 3 || variant == 4) return GVI_ERR_ARG;
  // 7 = auto, but the non-polynomial psi kinds take the sign-orbit kernel also where a register kernel exists
  ctx->prefer_opsi = variant == 7;
  ctx->variant = variant == 7 ? 0 : variant;
  return GVI_OK;
}

}  // extern "C"
