#!/usr/bin/env python3
"""Benchmark of the NGD Gauss-Hermite hot path on MI355X (contract: see the round prompt).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5|c3lit|planar1k|arm7x]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the synthetic 1024-factor LTV-prior
chain, d = 12, sparse-GH p = 5 (N = 17 217 sigma points per factor), T = 1025 states of size 6, fp64, plus one unary
d = 6 measurement factor per state (gaussianvi_amd/synthetic.py says why).  One STEP = one NGD iteration of
GVIGH::optimize (gvibase/GVI-GH-impl.h:39-118) entirely on the device: moments pass of every factor, ordered assemble,
block-tridiagonal solve, then line-search trials (axpy, chain factorisation = log-det + marginals, gather, cost) until the
first accepted one.  Inputs are resident in HBM before the timed region.

What the JSON line reports
  value                      every psi evaluation executed in the timed region / wall time, whole job.  The passes are
                             COUNTED BY THE LIBRARY (gvi_ngd_counters): `full` passes accumulate all moments, `cost`
                             passes only m0.  With the default adaptive scheduling an accepted iteration runs ONE full
                             pass (its m0 is the trial cost, its moments are the next gradients), so value ~ the
                             moments-pass rate; both rates are given separately as well.
  roofline.frac              EXECUTED fp64 VALU issue slots of the dominant launch (each v_fma/v_mul/v_add_f64 counted
                             as one FMA = 2 flop) / the 78.6 TF fp64 peak -- the fraction of the pipe that is used.
                             The SURVEY 8(d) algorithmic figure (638 flop per evaluation for the reference's x-space
                             formulation) is kept as roofline.algorithmic.
  N > 1                      default --scaling strong = BASELINE configs[3]: the SAME 1024-factor chain, factor list
                             sharded contiguously over the ranks; --scaling weak runs a (1024 N)-factor chain ("c3xN")
                             and says so in `metric`.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_PEAK = 78.6e12        # FLOP/s, AMD public spec sheet (vector = matrix fp64 on MI355X); not in the guide
TRAFFIC_FILE = os.path.join("profiles", "r04_traffic.json")
KSTATS_FILE = os.path.join("profiles", "r04_kernel_stats.csv")      # rocprofv3 --kernel-trace --stats of this command (c3)


def git_blob_sha1(path):
    """What `git hash-object` prints for the file: lets a reader check which committed file a number was read from."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def ensure_library(rank: int) -> None:
    """Build the in-tree C-ABI library if the checkout has none -- BEFORE anything touches the GPU or a process group
    (hipcc must not be spawned from a process that has initialised HIP, and every rank has to take the same path):
    rank 0 builds into a temporary file and renames it into place (atomic), the other ranks wait for the file."""
    from gaussianvi_amd import _lib, build
    if os.path.exists(_lib.LIB_PATH):
        return
    if rank == 0:
        build.build_lib()
        return
    t0 = time.time()
    while not os.path.exists(_lib.LIB_PATH):
        if time.time() - t0 > 900:
            raise SystemExit(f"rank {rank}: {_lib.LIB_PATH} did not appear (rank 0 builds it)")
        time.sleep(0.5)


KIND_NAMES = {0: "range-1D", 1: "quadratic-prior", 2: "fixed-prior", 4: "hinge-SDF-2D", 5: "hinge-SDF-2D-body", 6: "hinge-SDF-3D", 7: "hinge-SDF-3D-arm"}
# fp64 VALU instructions of ONE 64-point step of the lane-per-point register kernel, counted in the ISA of the instance
# (hipcc --save-temps of moments_reg_kernel<D, Psi, full>): (kind, d) -> count.  79 at the planar hinge: 8 (pose rows) +
# ~45 (clamp, two fp64 divisions, floor, bilinear weights and blend) + 5 (hinge) + 21 (c = w psi, m0, m1, packed M2)
REG_KERNEL_FP64_PER_EVAL = {(4, 4): 79}


def api_count(d, p):
    from gaussianvi_amd import api
    return api.spgh_count(d, p)


def cpu_baseline(chain, seconds, set_index=0):
    """Reference-shaped CPU port (oracle/c/gvi_oracle.c: per-factor symmetric sqrt + expand, three
    Integrate passes with psi re-evaluated through a function pointer, OpenMP over factors) timed on
    this box's host cores on a bounded sample of the SAME workload: the first factors of the prior set
    at their start-state marginals (as many as ~`seconds` of CPU work allow, at most 256), repeated."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import gvi_oracle as o
    spec = chain["specs"][set_index]
    d, n = spec["d"], chain["n"]
    if spec["kind"] == 4:                               # HINGE_SDF_2D: the grid goes to the port once
        c_oracle.set_sdf2d(spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    elif spec["kind"] not in (0, 1, 2):
        return {"value": None, "unit": "psi-evals/s", "cores": 0, "kind": "port",
                "sample": f"skipped: the C port has no psi kind {spec['kind']}"}
    n_arg = n if spec["kind"] in (1, 4) else d
    Kall = len(spec["start"])
    N_est = api_count(d, spec["p"])
    if N_est * d * 8 > 2 ** 28:
        # the reference-shaped port materialises an N x d sigma-point batch PER THREAD (as the reference does per
        # factor): at (24,7) that is 3.9 GB x 128 threads.  Not run: it would exhaust the host (it took a box down once).
        return {"value": None, "unit": "psi-evals/s", "cores": 0, "kind": "port",
                "sample": f"skipped: the (d={d}, p={spec['p']}) table has {N_est} points; the reference-shaped CPU path needs "
                          f"{N_est * d * 8 / 2**30:.1f} GiB per thread"}
    SD, SU = o.inverse_gbp(chain["D0"], chain["U0"])
    mk, Sk = o.gather_marginals(chain["mu0"], SD, SU, spec["start"][:min(256, Kall)], d)
    if spec["kind"] == 4:                               # marginals that reach the obstacles (the hinge is active)
        mk = mk.copy(); mk[:, 1] += np.linspace(0.0, 1.2, len(mk))
    Z, w = o.nwspgr(d, spec["p"])
    threads = c_oracle.max_threads()
    # size the sample from a probe of one factor per thread (all threads)
    Kp = min(Kall, threads, len(mk))
    t0 = time.perf_counter()
    c_oracle.moments(Z, w, mk[:Kp], Sk[:Kp], spec["kind"], spec["params"][:Kp], n_arg, fused=False)
    t_probe = max(time.perf_counter() - t0, 1e-6)
    K = int(max(1, min(256, Kall, len(mk), Kp * max(1, int(seconds * 0.1 / t_probe)))))
    out = {}
    for name, fused in (("reference_style", False), ("fused", True)):
        c_oracle.moments(Z, w, mk[:min(8, K)], Sk[:min(8, K)], spec["kind"], spec["params"][:min(8, K)], n_arg, fused=fused)   # warm
        reps, t0 = 0, time.perf_counter()
        budget = seconds * (0.7 if not fused else 0.3)
        while True:
            c_oracle.moments(Z, w, mk[:K], Sk[:K], spec["kind"], spec["params"][:K], n_arg, fused=fused)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget:
                break
        out[name] = K * len(w) * reps / el
    return {"value": out["reference_style"], "unit": "psi-evals/s", "cores": threads, "kind": "port",
            "sample": f"{K} of the {Kall} d={d} p={spec['p']} {KIND_NAMES.get(spec['kind'], 'kind %d' % spec['kind'])} factors x {len(w)} sigma points, full moments pass "
                      f"(3 Integrate passes, psi x3 per point) repeated for ~{seconds:.0f} s, OpenMP {threads} threads",
            "fused_single_pass_value": out["fused"]}


def exec_ops(d: int, m: int, full: bool, signed: bool = False, mirror: bool = False) -> float:
    """fp64 VALU instructions per EVALUATION of a sum-of-squares psi with m residual rows in the z-space formulation
    (DESIGN section 2): u = u0 + H z (m d FMA), psi = sum s_r u_r^2 (m FMA; m more MUL for the signs when some residual
    weight is negative -- `signed`), c = w psi (1), m0 (1); the full pass adds t = c z (d), m1 (d) and the packed upper
    triangle of M2 (d (d + 1) / 2 FMA).  `mirror`: the kernel evaluates a +-pair from ONE v = H z: q = sum s v^2 (m, + m
    signed), l = sum (s u0) v (m), c+ = 2 w (q + k0) and c- = 4 w l (5) -- the count of one pair is halved."""
    sq = 2 * m if signed else m
    if mirror:
        pair = m * d + sq + m + 5 + 1
        if full:
            pair += 2 * d + d * (d + 1) // 2
        return pair / 2
    ops = m * d + sq + 2
    if full:
        ops += 2 * d + d * (d + 1) // 2
    return ops


def orbit_exec_ops(d: int, p: int, m: int, full: bool, signed: bool = False) -> float:
    """fp64 VALU instructions per EVALUATION of the sign-orbit kernel (csrc/kernels_orbit.hpp), averaged over the (d, p)
    table: an orbit of support size s costs  s m (start corner)  +  2^(s-1) half-points x [q: m (+ m signed), l: m, c+ and
    its sum: 2, sign-weighted sums: s + s (s - 1) / 2]  +  (2^(s-1) - 1) flips x (m + 1)  +  the per-orbit scaling
    (3 + 5 s + 2 s (s - 1) / 2 multiplies); the cost pass keeps q, c+ and the flips only.  For s <= 4 the full pass forms
    the sign-weighted sums by a butterfly after the walk instead (checked against the ISA: 251 fp64 instructions in the s = 4
    tile body, 130 at s = 3).  m = 12, s >= 4 runs as two walks of six rows with the butterfly at every s (orbit_walk_split):
    about the same instruction count (PMC: 4.43e10 vs 4.31e10 VALU instructions per (24,7) launch), so the count above is kept.
    The LDS atomics that fold an orbit into the factor's accumulators are not VALU instructions."""
    import numpy as np
    from gaussianvi_amd import api
    Z, w, _ = api.spgh_nodes(d, p)
    rep = np.all(Z >= 0.0, axis=1)
    sizes = np.count_nonzero(Z[rep], axis=1)
    total = 0.0
    for sz in range(1, int(sizes.max()) + 1):
        n = int(np.sum(sizes == sz))
        nh = 2 ** (sz - 1)
        sq = 2 * m if signed else m
        if full and 2 <= sz <= 4:
            # the two scalars of every half-point are kept and one Walsh-Hadamard butterfly per scalar gives the sign-weighted
            # sums (dead outputs pruned): c+ needs every mask of <= 2 coordinates, l the empty mask and the singletons
            log2nh = sz - 1
            wht = nh * log2nh - (1 if sz == 4 else 0) + {2: 2, 4: 7, 8: 18}[nh]
            per_orbit = sz * m + nh * (sq + m + 1) + (nh - 1) * (m + 1) + wht + (3 + 5 * sz + sz * (sz - 1))
        else:
            per_point = sq + 2 + ((m + sz + sz * (sz - 1) // 2) if full else 0)
            per_orbit = sz * m + nh * per_point + (nh - 1) * (m + 1) + ((3 + 5 * sz + sz * (sz - 1)) if full else 2)
        total += n * per_orbit
    return total / len(w)


def alg_flops(d: int, m: int) -> int:
    """expand d^2 + residual + quadratic form (m^2 + m each) + accumulate 1 + d + d (d + 1) / 2 FMA."""
    return 2 * (d * d + 2 * (m * m + m) + 1 + d + d * (d + 1) // 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-calls", action="store_true", help="one gvi_ngd_step call per iteration from Python instead of gvi_ngd_run blocks")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 generic, 2 register (LDS operands), 3 operand-resident, 5 register (SGPR operands), 6 sign-orbit")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default, BASELINE configs[3]): the same chain over N GPUs (128 factors per GPU at N = 8: bounded "
                         "by the replicated chain recursions and the exchange); weak: 1024 factors PER GPU, i.e. a (1024 N)-factor "
                         "chain (config c3xN) -- the metric string then names c3xN")
    ap.add_argument("--no-c5-strong", action="store_true",
                    help="N > 1: skip the additional sharded run of BASELINE configs[4] (block `c5_strong` of the JSON line; ~80 s of set-up)")
    ap.add_argument("--fuse-trial", type=int, default=2, choices=[0, 1, 2],
                    help="gvi_ngd_set_mode: 0 the reference's pass order (gradient pass + one cost pass per trial), 1 fused, "
                         "2 adaptive (library default)")
    ap.add_argument("--restart-every", type=int, default=None,
                    help="re-initialise (mu0, precision0) inside the timed region every R steps so that every step is a "
                         "descending iteration with one accepted trial (the chain converges after ~35 steps; default 30, 6 for the "
                         "obstacle graphs planar1k / arm7x)")
    args = ap.parse_args()
    big = args.config.startswith("c5") and args.config != "c5mini"
    if args.restart_every is None:                     # the obstacle graphs converge within ~8 iterations, the quadratic chains in ~35
        args.restart_every = 6 if args.config in ("planar1k", "planar", "arm7x") else 30
    if args.steps is None:
        args.steps = 3 if big else 200
    if args.warmup is None:
        args.warmup = 1 if big else 20

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ensure_library(rank)                               # before torch.cuda / the process group (see docstring)

    import numpy as np
    import torch
    import torch.distributed as dist

    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device):
    # GVI_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges through gloo.  Timings of such a run mean nothing.
    rehearsal = os.environ.get("GVI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_pg = world > 1 or "RANK" in os.environ
    rccl_ranks = None
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            probe = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(probe)                     # an actual RCCL collective: how many ranks took part
            rccl_ranks = int(round(float(probe.item())))

    from gaussianvi_amd import api, synthetic
    from gaussianvi_amd.dist import shard_chain, torch_allgather

    if args.scaling == "weak" and world > 1 and args.config == "c3":
        args.config = f"c3x{world}"
    chain = synthetic.make_chain(args.config)
    local = shard_chain(chain, rank, world)
    ctx, ids = api.context_for_chain(local, device=local_rank)
    ctx.set_variant(args.variant)
    ctx.ngd_set_mode(True, args.fuse_trial)
    # N > 1 (or a forced size-1 group): both exchange steps of a pass run INSIDE the library on its own stream
    # (include/gvi_hip.h, gvi_dist_init_*): RCCL all-gathers; the rehearsal on one GPU goes through a gloo callback
    sharded = use_pg and (world > 1 or os.environ.get("GVI_FORCE_ALLREDUCE") == "1")

    def init_transport(cx):
        """Exchange transport of a sharded context.  The library's own RCCL communicator (dlopen of librccl); should loading
        it or creating the unique id fail on ANY rank, every rank falls back to the callback transport over
        torch.distributed's RCCL group -- the same all-gathers on the library's stream, one Python call each.  (The fallback
        covers load / unique-id failures; ncclCommInitRank itself is collective: a rank stuck inside it is a hung job, which
        the launcher's timeout ends -- no in-process retry.)"""
        if rehearsal:
            cx.dist_init_callback(rank, world, torch_allgather(local_rank))
            return "gloo callback (rehearsal)"
        ok = torch.ones(1, dtype=torch.float64, device="cuda")
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        try:
            my_id = api.dist_unique_id()           # every rank: also the probe that librccl loads here
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
        except Exception as e:
            ok.zero_()
            print(f"[bench rank {rank}] gvi_dist_unique_id failed: {e}", file=sys.stderr, flush=True)
        dist.broadcast(uid, 0)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0.5:
            # ncclCommInitRank is collective: a rank that never returns from it would leave the job to the launcher's
            # timeout.  Wall-clock guard: the call runs in a worker thread; if it has not returned after
            # GVI_BENCH_INIT_TIMEOUT_S (default 120 s) the process ends with a non-zero code -- a fresh process is the only
            # safe retry (never a re-exec, never an in-process retry of a half-initialised communicator).
            import threading
            box = {}

            def _init():
                try:
                    cx.dist_init_rccl(rank, world, bytes(uid.cpu().numpy().tobytes()))
                    box["ok"] = True
                except Exception as e:                          # noqa: BLE001 (reported below)
                    box["err"] = e
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("GVI_BENCH_INIT_TIMEOUT_S", "120")))
            if th.is_alive():
                print(f"[bench rank {rank}] gvi_dist_init_rccl (ncclCommInitRank) has not returned: giving up", file=sys.stderr, flush=True)
                os._exit(3)
            if "err" in box:
                ok.zero_()
                print(f"[bench rank {rank}] gvi_dist_init_rccl failed: {box['err']}", file=sys.stderr, flush=True)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 0.5:
            cx.dist_init_callback(rank, world, torch_allgather(local_rank))
            return "RCCL (torch.distributed group through the callback transport)"
        return "RCCL (library communicator)"

    transport = init_transport(ctx) if sharded else "none"
    ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
    # HIP events around every 8th dominant launch (a pair costs ~14 us of queue gaps); every launch for the seconds-long c5 passes
    ctx.profile_enable(1 if big else 3)

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    evals_pass = sum(K * N for (K, d, p, N) in ctx.sets)          # local, per pass over every set
    # the whole iteration (backtracking loop and, when sharded, both exchanges included) is one C-ABI call
    single = not sharded
    step_fn = lambda: ctx.ngd_step(0.55, 10)

    def restart():
        ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])

    # one-time costs (buffer growth on the first cost pass, side stream / event creation, kernel attribute calls, the
    # cold Jacobi start) are primed outside the contract's W warm-up steps as well, so a small W does not time them
    # (and a fixed ~25 ms of untimed iterations first: a fresh process on a fresh box starts at low clocks)
    # No cyclic-GC pass of the interpreter inside the timed region (what timeit does): with torch imported a full collection
    # walks ~10^6 objects and takes ~40 ms -- observed as ONE restart block of 6.35 ms per step among 84 blocks of 0.11 ms in a
    # planar1k run (0.205 instead of 0.13 ms per step), and as the occasional 0.17 ms c3 line.  Collected HERE, in front of the
    # priming and warm-up steps, not between the warm-up and the timed steps: 40 ms of an idle GPU right in front of the timed
    # region dropped its clocks, and the first block of a 20-step region ran 8 % slower than the blocks behind it.
    gc.collect()
    gc.disable()
    if not big:
        for j in range(200):
            if j % args.restart_every == 0:
                restart()
            step_fn()
    # the W warm-up steps start from the initial state; the timed steps continue from there, so the timed region
    # begins with the iteration pipeline warm (a restart costs one cold step: synchronous upload + chain refresh)
    restart()
    pos = 0                                             # position inside the current restart block
    for i in range(args.warmup):
        if pos == args.restart_every:
            restart(); pos = 0
        step_fn(); pos += 1
    barrier()
    ctx.ngd_counters(reset=True)
    t0 = time.perf_counter()
    kern_ms, log, block_ms = [], [], []
    if single and not args.per_step_calls:
        # the iteration loop of GVIGH::optimize as ONE C call per block of steps (gvi_ngd_run): no interpreter between the
        # decision of an iteration and the launches of the next
        done = 0
        while done < args.steps:
            if pos == args.restart_every:
                restart(); pos = 0                      # timed: host upload + one refresh of the chain products
            tb = time.perf_counter()
            blk = ctx.ngd_run(min(args.steps - done, args.restart_every - pos, 64), 0.55, 10)
            block_ms.append(1e3 * (time.perf_counter() - tb) / max(1, len(blk)))
            log.extend(blk); done += len(blk); pos += len(blk)
    else:
        for i in range(args.steps):
            if pos == args.restart_every:
                restart(); pos = 0                          # timed: host upload + one refresh of the chain products
            r = step_fn(); pos += 1
            log.append(r)
            if i % 8 == 7 or i == args.steps - 1:
                try:
                    kern_ms.append(ctx.profile_last(ids[0], 0))      # last bracketed launch (sampled: every 8th)
                except api.GviError:
                    pass
    t_blocks = time.perf_counter() - t0
    ctx.sync()                                          # the library's stream, polled: the contract's barrier + synchronize below then finds the device idle
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    n_full, n_cost = ctx.ngd_counters()
    if single and not args.per_step_calls:
        # event time of the bracketed dominant launch: sampled OUTSIDE the timed region (reading an event pair is a host-side
        # synchronisation of ~30 us; inside a 20-step region it was 4 % of the line)
        for _ in range(1 if big else 8):
            if pos == args.restart_every:
                restart(); pos = 0
            blk = ctx.ngd_run(min(1 if big else 8, args.restart_every - pos), 0.55, 10)
            pos += len(blk)
            try:
                kern_ms.append(ctx.profile_last(ids[0], 0))
            except api.GviError:
                pass

    # A/B leg (not part of `value`): the reference's pass order -- a cost-only pass per trial and a separate
    # gradient pass per iteration (gvi_ngd_set_mode fuse_trial = 0); same iterates, one more psi pass per iteration
    ab = None
    if single and not big and args.fuse_trial != 0:
        ctx.ngd_set_mode(True, 0)
        ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
        fpos = 0
        for _ in range(max(1, args.warmup)):
            if fpos == args.restart_every:
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"]); fpos = 0
            ctx.ngd_step(0.55, 10); fpos += 1
        torch.cuda.synchronize()
        ctx.ngd_counters(reset=True)
        tf0 = time.perf_counter()
        flog = []
        while len(flog) < args.steps:
            if fpos == args.restart_every:
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"]); fpos = 0
            if args.per_step_calls:
                blk = [ctx.ngd_step(0.55, 10)]
            else:
                blk = ctx.ngd_run(min(args.steps - len(flog), args.restart_every - fpos, 64), 0.55, 10)
            flog.extend(blk); fpos += len(blk)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf0
        f_full, f_cost = ctx.ngd_counters()
        ab = {"ms_per_step": 1e3 * tf / args.steps, "ngd_iters_per_s": args.steps / tf,
              "final_cost": flog[-1]["new_cost"], "trials_per_step": float(np.mean([r["ntrials"] for r in flog])),
              "full_passes": f_full, "cost_passes": f_cost,
              "all_pass_evals_per_s": (f_full + f_cost) * evals_pass / tf,
              "note": "gvi_ngd_set_mode(fuse_trial=0): the reference's pass order (one gradient pass per iteration + one "
                      "m0-only cost pass per trial); identical iterates"}
        ctx.ngd_set_mode(True, args.fuse_trial)

    stats = torch.tensor([elapsed, float(n_full * evals_pass), float(n_cost * evals_pass)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed, evals_full, evals_cost = float(tmax[0]), float(stats[1]), float(stats[2])
    else:
        evals_full, evals_cost = float(stats[1]), float(stats[2])

    # ---- outside the timed region: per-stage event timing (chain / factor pass / assemble) and, for workloads whose
    # dominant launch is not the bracketed one of set 0, the event time of the dominant set's moments kernel ----
    dom = int(np.argmax([K * N for (K, d, p, N) in ctx.sets])) if ctx.sets else 0
    chain_pattern = len(ctx.sets) <= 2 and all(spec["kind"] in (1, 2) for spec in local["specs"])
    stages = None
    if single and not big:
        def descending_steps(nsteps):                   # same restart rule as the timed region: every step an accepted one
            done = 0
            while done < nsteps:
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
                for _ in range(min(nsteps - done, args.restart_every)):
                    ctx.ngd_step(0.55, 10); done += 1
        descending_steps(4)
        ctx.profile_stages(True, read=False)
        descending_steps(16)
        st = ctx.profile_stages(False)
        stages = {k: {"mean_us": round(float(v[0]), 2), "brackets": int(v[1])} for k, v in st.items()}
        if not chain_pattern:
            ctx.profile_enable(2)
            kern_ms = []
            for _ in range(3):
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
                for _ in range(min(4, args.restart_every)):
                    ctx.ngd_step(0.55, 10)
                    try:
                        kern_ms.append(ctx.profile_last(ids[dom], 0))
                    except api.GviError:
                        pass
            ctx.profile_enable(0)

    # ---- N > 1: the sharded result against the SAME steps unsharded on rank 0's GPU (the first run over RCCL / xGMI has
    # to say by itself whether it computed the right thing): every rank repeats a fresh block of steps on the sharded
    # context, rank 0 also on a one-GPU context of the whole chain; relative gaps of the final cost and mean, expected <= 1e-9
    def parity_vs_single_gpu(cx_sharded, ch, nsteps, tables=None):
        cx_sharded.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
        rs = [cx_sharded.ngd_step(0.55, 10) for _ in range(nsteps)]
        mu_s = cx_sharded.ngd_get_state()["mu"].copy()
        out = None
        if rank == 0:
            t_ref = time.perf_counter()
            ref, ref_ids = api.context_for_chain(ch, device=local_rank, tables=tables)
            ref.ngd_set_mode(True, args.fuse_trial)
            ref.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
            rr = [ref.ngd_step(0.55, 10) for _ in range(nsteps)]
            mu_r = ref.ngd_get_state()["mu"]
            st_ref = None
            if not tables:                                   # stage times of the one-GPU iteration (strong_scaling_model)
                ref.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
                ref.profile_stages(True, read=False)
                for _ in range(min(16, args.restart_every)):
                    ref.ngd_step(0.55, 10)
                st_ref = {k: float(v[0]) for k, v in ref.profile_stages(False).items()}
            ref.close()
            out = {"steps": nsteps, "final_cost_sharded": rs[-1]["new_cost"], "final_cost_single_gpu": rr[-1]["new_cost"],
                   "rel_gap_cost": abs(rs[-1]["new_cost"] - rr[-1]["new_cost"]) / abs(rr[-1]["new_cost"]),
                   "rel_gap_mu": float(np.abs(mu_s - mu_r).max() / np.abs(mu_r).max()),
                   "same_accept_decisions": [a["accepted"] for a in rs] == [b["accepted"] for b in rr] and
                                            [a["ntrials"] for a in rs] == [b["ntrials"] for b in rr],
                   "single_gpu_stage_us": st_ref, "expected": "<= 1e-9", "reference_s": round(time.perf_counter() - t_ref, 2)}
        barrier()
        return out

    parity = parity_vs_single_gpu(ctx, chain, min(12, args.restart_every)) if world > 1 else None

    # ---- N > 1: BASELINE configs[4] (4096 factors, d = 24, p = 7; fp64) sharded over the same ranks, two iterations:
    # the workload whose factor pass (0.095 s on one GPU) dwarfs the replicated chain operations, i.e. where sharding has
    # work to split (strong scaling).  GVI_BENCH_C5_CONFIG selects a smaller d = 24 chain for rehearsals.
    c5_block = None
    if world > 1 and not args.no_c5_strong:
        cfg5 = os.environ.get("GVI_BENCH_C5_CONFIG", "c5")
        t_build = time.perf_counter()
        chain5 = synthetic.make_chain(cfg5)
        local5 = shard_chain(chain5, rank, world)
        # the (d, p) table of the prior set is generated ONCE, on rank 0 (70 s at (24,7)), and handed to the other ranks
        # (broadcast + gvi_factors_add_table) instead of being generated by every rank
        d5, p5 = int(chain5["specs"][0]["d"]), int(chain5["specs"][0]["p"])
        N5 = api.spgh_count(d5, p5)
        if rank == 0:
            Z5, w5, _ = api.spgh_nodes(d5, p5)
        else:
            Z5, w5 = np.empty((N5, d5)), np.empty(N5)
        t_table = time.perf_counter() - t_build
        dev = "cpu" if rehearsal else "cuda"
        for arr in (Z5, w5):
            flat = arr.reshape(-1)
            for lo in range(0, flat.size, 1 << 27):          # 1 GiB pieces
                piece = torch.from_numpy(flat[lo:lo + (1 << 27)]).to(dev)
                dist.broadcast(piece, 0)
                if rank != 0:
                    flat[lo:lo + (1 << 27)] = piece.cpu().numpy()
                del piece
        tables5 = {(d5, p5): (Z5, w5)}
        ctx5, ids5 = api.context_for_chain(local5, device=local_rank, tables=tables5)
        ctx5.ngd_set_mode(True, 2)
        tr5 = init_transport(ctx5)
        ctx5.ngd_init(chain5["mu0"], chain5["D0"], chain5["U0"])
        ctx5.ngd_step(0.55, 10)                              # warm-up (table upload, buffers)
        ctx5.ngd_init(chain5["mu0"], chain5["D0"], chain5["U0"])
        t_build = time.perf_counter() - t_build
        barrier()
        ctx5.ngd_counters(reset=True)
        gc.collect()
        gc.disable()
        t5 = time.perf_counter()
        log5 = [ctx5.ngd_step(0.55, 10) for _ in range(2)]
        barrier()
        t5 = time.perf_counter() - t5
        gc.enable()
        f5, c5c = ctx5.ngd_counters()
        ev5 = sum(K * N for (K, d, p, N) in ctx5.sets)
        st5 = torch.tensor([t5, float((f5 + c5c) * ev5)], dtype=torch.float64, device="cuda")
        mx5 = st5.clone()
        dist.all_reduce(mx5, op=dist.ReduceOp.MAX)
        dist.all_reduce(st5, op=dist.ReduceOp.SUM)
        nfac = torch.zeros(world, dtype=torch.float64, device="cuda")
        nfac[rank] = float(ctx5.sets[0][0])
        dist.all_reduce(nfac, op=dist.ReduceOp.SUM)
        parity5 = parity_vs_single_gpu(ctx5, chain5, 2, tables=tables5)
        c5_block = {"config": cfg5, "parity_vs_single_gpu": parity5, "table_generation_s_rank0": round(t_table, 1),
                    "table": f"({d5}, {p5}): {N5} points, generated on rank 0 and broadcast", "workload": f"T={chain5['T']} n={chain5['n']}: {chain5['T'] - 1} prior factors d={ctx5.sets[0][1]} p={ctx5.sets[0][2]} "
                                                  f"(N={ctx5.sets[0][3]}) + {chain5['T']} unary d={ctx5.sets[1][1]} factors, fp64, factor list sharded over {world} ranks",
                    "steps": 2, "ms_per_step": 1e3 * float(mx5[0]) / 2, "value": float(st5[1]) / float(mx5[0]), "unit": "psi-evals/s",
                    "accepted_steps": int(sum(r["accepted"] for r in log5)), "final_cost": log5[-1]["new_cost"],
                    "factors_per_rank": [int(v) for v in nfac.tolist()], "transport": tr5, "scaling": "strong",
                    "setup_s_rank0": round(t_build, 1),
                    "note": "BASELINE configs[4] in fp64 (fp32 refused: DESIGN section 4.4); one-GPU reference: bench.py --config c5"}
        ctx5.close()

    # ---- BASELINE configs[2] as SURVEY 8(d) writes it (`c3lit`: the 1024 LTV priors and two end anchors, no unary set,
    # cond(V) = 2.3e5): 20 driver-timed steps beside the default chain (which adds a unary factor per state)
    c3_literal = None
    if world == 1 and args.config == "c3":
        lit = synthetic.make_chain("c3lit")
        cl, _ = api.context_for_chain(lit, device=local_rank)
        cl.ngd_set_mode(True, args.fuse_trial)
        cl.ngd_init(lit["mu0"], lit["D0"], lit["U0"])
        for _ in range(5):
            cl.ngd_step(0.55, 10)
        cl.ngd_init(lit["mu0"], lit["D0"], lit["U0"])
        torch.cuda.synchronize()
        cl.ngd_counters(reset=True)
        tl = time.perf_counter()
        llog = cl.ngd_run(20, 0.55, 10)
        torch.cuda.synchronize()
        tl = time.perf_counter() - tl
        lf, lc = cl.ngd_counters()
        evl = sum(K * N for (K, d, p, N) in cl.sets)
        c3_literal = {"config": "c3lit", "steps": len(llog), "ms_per_step": 1e3 * tl / max(1, len(llog)),
                      "trials_per_step": float(np.mean([r["ntrials"] for r in llog])), "accepted_steps": int(sum(r["accepted"] for r in llog)),
                      "final_cost": llog[-1]["new_cost"], "psi_evals_per_s": (lf + lc) * evl / tl, "factor_sets": [list(map(int, x)) for x in cl.sets],
                      "note": "the literal configs[2] chain (no unary factors; ill-conditioned, DESIGN section 6): 20 iterations from the "
                              "initial state in one gvi_ngd_run call, backtracking included"}
        cl.close()

    if rank == 0:
        Kd, dd_, pd_, Nd = ctx.sets[dom]
        K0, d0, p0, N0 = ctx.sets[0]
        m0 = d0 // 2
        geo = ctx.profile_geometry(ids[dom])
        km = float(np.mean(kern_ms)) * 1e-3 if kern_ms else float("nan")
        kind_dom = int(local["specs"][dom]["kind"])

        def weights_signed(spec):                      # any negative eigenvalue of the residual weight (Q^-1 / K^-1)?
            W = spec.get("Qinv", spec.get("Kinv"))
            return bool(np.linalg.eigvalsh(0.5 * (W + np.transpose(W, (0, 2, 1)))).min() <= 0.0)

        fused_launch = os.environ.get("GVI_FUSED", "1") != "0" and chain_pattern and geo["variant"] == 6 and geo["nchunk"] == 4 and \
            ((m0 == 6 and d0 == 12) or (m0 == 2 and d0 == 4))
        block3_launch = (os.environ.get("GVI_FUSED", "1") != "0" and os.environ.get("GVI_NO_PAIR", "0") == "0" and world == 1 and
                         [(int(sp["kind"]), int(sp["d"])) for sp in local["specs"]] == [(1, 8), (4, 4), (2, 4)])
        executed_note = None
        if chain_pattern:
            sets_in_launch = [(K0, d0, m0, N0, weights_signed(local["specs"][0]))]
            both = geo["variant"] in (5, 6) and len(ctx.sets) == 2
            if both:
                K1, d1, p1, N1 = ctx.sets[1]
                sets_in_launch.append((K1, d1, d1, N1, weights_signed(local["specs"][1])))   # unary: m = d
            if geo["variant"] == 6:
                kernel_name = (f"factor_fused_kernel<{m0}, {4 if p0 <= 5 else 6}, ..., {d0}, {ctx.sets[1][1] if both else d0 // 2}> (gather + Cholesky products + "
                               f"sign-orbit walk + chunk sum + cost tail + back-transform in one launch)") if fused_launch else \
                    f"moments_orbit{'_pair' if both else ''}_kernel<{m0}, {4 if p0 <= 5 else 6}, full>"
                p_of = {d: p for (K, d, p, N) in ctx.sets}
                ops_of = lambda d, m, sg: orbit_exec_ops(d, p_of[d], m, True, sg)
            elif geo["variant"] == 5:
                kernel_name = f"moments_sreg_pair_kernel<{d0}, {m0}, {m0}, {m0}, full>"
                mirror = os.environ.get("GVI_MIRROR", "1") != "0"
                ops_of = lambda d, m, sg: exec_ops(d, m, True, sg, mirror)
            elif geo["variant"] == 3:
                kernel_name = f"moments_split_kernel<{d0}, {(m0 + 3) // 4}, full>"
                ops_of = lambda d, m, sg: exec_ops(d, m, True, True) + 3 + 12      # four waves per factor: sign multiply, 4 partial psi sums
            else:
                kernel_name = f"moments_reg_kernel<{d0}, PsiQuad<{d0},{m0}>, full>" if geo["variant"] == 2 else "moments_generic_kernel"
                ops_of = lambda d, m, sg: exec_ops(d, m, True, sg, False)
            evals_launch = sum(K * N for K, d, m, N, sg in sets_in_launch)
            exec_flop = sum(2 * ops_of(d, m, sg) * K * N for K, d, m, N, sg in sets_in_launch)
            alg_flop = sum(alg_flops(d, m) * K * N for K, d, m, N, sg in sets_in_launch)
            alg_bytes = sum(K * N * (d + 1) * 8 for K, d, m, N, sg in sets_in_launch)   # (d+1) s bytes per eval
            ops_table = {f"d={d},m={m}": ops_of(d, m, sg) for K, d, m, N, sg in sets_in_launch}
            launch_desc = ("every factor set, all stages of the pass" if fused_launch else
                           ("prior set + unary set in one launch" if len(sets_in_launch) == 2 else "prior set"))
        else:
            # non-polynomial psi (hinge on a signed-distance field ...): the lane-per-point register kernel of the dominant set
            per = REG_KERNEL_FP64_PER_EVAL.get((kind_dom, dd_))
            kernel_name = (f"moments_reg_kernel<{dd_}, Psi[{KIND_NAMES.get(kind_dom, kind_dom)}], full>" if geo["variant"] == 2
                           else "moments_generic_kernel")
            evals_launch = Kd * Nd
            exec_flop = 2.0 * per * evals_launch if per and geo["variant"] == 2 else float("nan")
            alg_flop = float("nan")
            alg_bytes = evals_launch * (dd_ + 1) * 8
            ops_table = {f"kind={KIND_NAMES.get(kind_dom, kind_dom)},d={dd_}": per}
            launch_desc = f"set {dom} ({KIND_NAMES.get(kind_dom, kind_dom)}, d = {dd_}, p = {pd_}: the set with the most evaluations)"
            executed_note = ("fp64 VALU instructions per 64-point step counted in the ISA of the instance (REG_KERNEL_FP64_PER_EVAL), each as one "
                             "FMA = 2 flop; the SDF look-up adds 4 x 8 B of gathered reads per evaluation from L2")
        traffic, traffic_source, valu_issue = None, None, None
        # PMC passes are separate runs (tools/gpu_pmc.sh); their per-launch figures are committed per config
        tfile = TRAFFIC_FILE if args.config == "c3" else TRAFFIC_FILE.replace(".json", f"_{args.config}.json")
        tpath = os.path.join(ROOT, tfile)
        if world == 1 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if kernel_name.split("<")[0] in tj.get("kernel", ""):     # counters of the kernel that actually ran
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = (f"{tfile} (git blob {git_blob_sha1(tpath)}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command; not re-measured in this run)")
                    valu_issue = {"busy_frac": tj["derived"]["valu_pipe_busy"], "valu_instructions_per_launch": tj["sq"]["SQ_INSTS_VALU"],
                                  "wait_frac_of_wave_cycles": tj["derived"]["wait_fraction_of_wave_cycles"],
                                  "source": tfile + " (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / kernel cycles; counts every VALU "
                                            "instruction, fp64 or integer -- the resource this kernel is bound by)"}
            except Exception:
                traffic = None
        # the dominant kernel's duration as rocprofv3 reports it (kernel-trace of this command, committed): the HIP-event
        # bracket around a launch reads ~9 us more than the kernel runs.  frac is computed on the trace figure when the
        # committed file is about the kernel that ran; both are in the line.
        km_events = km
        trace_src = None
        kpath = os.path.join(ROOT, KSTATS_FILE)
        if world == 1 and args.config == "c3" and os.path.exists(kpath):
            import csv
            for row in csv.reader(open(kpath)):
                if row and kernel_name.split("<")[0] in row[0] and "AverageNs" not in row:
                    km = float(row[3]) * 1e-9
                    trace_src = f"{KSTATS_FILE} (git blob {git_blob_sha1(kpath)}): AverageNs of {row[1]} launches"
                    break
        base_metric = "sigma-point psi-evals/sec + NGD iters/sec, 1024-factor d=12 p=5 chain"
        if args.config == "c3":
            metric = base_metric
        else:
            metric = f"sigma-point psi-evals/sec + NGD iters/sec, {chain['T'] - 1}-factor d={d0} p={p0} chain ({args.config})"
        sets_desc = "; ".join(f"{K} x {KIND_NAMES.get(int(sp['kind']), sp['kind'])} d={d} p={p} (N={N})"
                              for (K, d, p, N), sp in zip(ctx.sets, local["specs"])) if world == 1 else None
        out = {
            "metric": metric,
            "value": (evals_full + evals_cost) / elapsed, "unit": "psi-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
            **({"rehearsal": "all ranks on cuda:0, gloo exchange: not a measurement"} if rehearsal else {}),
            **({"rccl_ranks": rccl_ranks} if rccl_ranks is not None else {}),
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: T={chain['T']} states, n={chain['n']}; 1 step = 1 device-resident NGD iteration; restart every {args.restart_every}",
                       "name": args.config, "factor_sets": [list(map(int, s)) for s in ctx.sets] if world == 1 else None,
                       "factor_sets_columns": "[K factors, d, GH degree p, N sigma points]", "factor_set_kinds": sets_desc,
                       "sharding": (f"factors/{world} contiguous; per pass: all-gather of each rank's state records of [g|D|U] "
                                    f"({ctx.dist_info()['records_per_rank']} states per rank, folded in rank order) with the partial cost sum as one "
                                    f"more record (one all-gather per iteration), issued inside the library ({'gloo callback (rehearsal)' if rehearsal else transport})") if sharded else "none",
                       "kernel_variant": geo["variant"], "chunks_per_factor": geo["nchunk"],
                       "fused_pass": bool(fused_launch),
                       # the planning graph's full pass as one launch (kernels_block.hpp: products -> psi moments -> epilogue in a
                       # workgroup); roofline.kernel below is its psi body, sampled as the set's own launch after the timed region
                       "one_launch_factor_stage": bool(block3_launch),
                       "mirror_pairs": bool(geo["variant"] == 5 and os.environ.get("GVI_MIRROR", "1") != "0"),
                       "fuse_trial": args.fuse_trial},
            "ngd_iters_per_s": args.steps / elapsed,
            # diagnosis only (`value` / `ms_per_step` are the whole timed region): per-step time of the gvi_ngd_run blocks without
            # the restarts between them; a max far above the median is a host / box stall (tens of ms have been seen), not the path
            "timed_region_ms": {"total": 1e3 * elapsed, "step_calls": 1e3 * t_blocks, "closing_barrier_and_synchronize": 1e3 * (elapsed - t_blocks)},
            "block_ms_per_step": ({"median": float(np.median(block_ms)), "max": float(np.max(block_ms)), "blocks": len(block_ms)}
                                  if block_ms else None),
            # `value` split by pass kind (SURVEY 8(d) defines an evaluation "in one moments pass")
            "passes": {"full": n_full, "cost_only": n_cost, "evals_per_pass": evals_pass * world if world > 1 else evals_pass,
                       "moments_pass_evals_per_s": evals_full / elapsed, "cost_pass_evals_per_s": evals_cost / elapsed,
                       "all_pass_evals_per_s": (evals_full + evals_cost) / elapsed},
            "accepted_steps": int(sum(r["accepted"] for r in log)),
            "trials_per_step": float(np.mean([r["ntrials"] for r in log])),
            "final_cost": log[-1]["new_cost"],
            "reference_pass_order": ab,
            "c3_literal": c3_literal,
            # event-timed stages of one iteration (gvi_profile_stages, 16 iterations outside the timed region): "chain" = trial
            # factorisation || gradient solve (kernels_chain.hpp, three launches), "factors" = the factor pass, "assemble"
            "iteration_breakdown_us": stages,
            "chain_us": stages["chain"]["mean_us"] if stages else None,
            "moments_kernel": {"ms": km_events * 1e3, "psi_evals_per_s": evals_launch / km_events, "launch": launch_desc, "timed_by": "HIP events, this run"},
            # The dominant kernel streams only the quadrature table, which is L2-resident (HBM traffic ~0.5 % of the algorithmic
            # bytes), so the binding roof is the fp64 VALU pipe, not HBM; it issues NO MFMA instruction (fp64 MFMA and fp64 VALU
            # share one pipe on MI355X and deliver the same rate: profiles/r02_fp64_pipes.txt), so the label says what runs.
            "roofline": {"bound": "valu_fp64 (the contract's compute label would be 'mfma'; zero MFMA instructions are issued)",
                         "achieved": exec_flop / km / 1e12, "peak": FP64_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": exec_flop / km / FP64_PEAK,
                         "traffic": traffic, "traffic_source": traffic_source, "valu_issue": valu_issue,
                         "kernel": kernel_name, "kernel_ms": km * 1e3, "kernel_ms_hip_events": km_events * 1e3,
                         "kernel_ms_source": trace_src or "HIP events around the launch (this run)", "evals_per_launch": evals_launch,
                         "executed_fp64_ops_per_eval": ops_table,
                         "note": executed_note or ("achieved/frac = EXECUTED fp64 VALU instructions of the bracketed launch (every set in it), each "
                                                   "counted as one FMA (2 flop), / HIP-event time / 78.6 TF"
                                                   + ("; the bracket covers the whole fused pass (products, walk, chunk sums, tail, back-transform), "
                                                      "only the walk's instructions are counted" if fused_launch else "")),
                         "algorithmic": {"flop_per_eval": alg_flops(d0, m0) if chain_pattern else None, "tflops": alg_flop / km / 1e12,
                                         "frac_of_peak": alg_flop / km / FP64_PEAK,
                                         "note": "SURVEY 8(d) count of the reference's x-space algorithm (expand GEMM + psi + three "
                                                 "moment passes); exceeds the executed figure because the z-space reformulation "
                                                 "removes the expand -- a statement about the algorithm, not about the pipe"},
                         "peak_source": "AMD MI355X spec sheet: 78.6 TF fp64 (vector = matrix); not in the local guide. Measured on this pool "
                                        "(tools/ubench/fp64_pipes.hip, profiles/r02_fp64_pipes.txt): 66.5 TF v_fma_f64, 67 TF v_mfma_f64_16x16x4 "
                                        "(one pipe: 62 TF interleaved)",
                         "hbm_algorithmic": {"bytes_per_launch": alg_bytes, "achieved_GBps": alg_bytes / km / 1e9,
                                             "peak_GBps": HBM_PEAK / 1e9, "frac": alg_bytes / km / HBM_PEAK,
                                             "note": "BASELINE's '>= 60 % of HBM roofline' figure: (d+1)*8 B per eval over 8 TB/s; "
                                                     "exceeds 1 where the table is served from L2"}},
        }
        if world > 1:
            out["transport"] = transport
            out.setdefault("rccl_ranks", None)                # (no RCCL group in a rehearsal)
            out["c5_strong"] = c5_block
            out["parity_vs_single_gpu"] = parity
            # Expected ceiling of strong scaling (BASELINE configs[3]): t(N) = W / N + R + X with W = the factor pass (shards),
            # R = chain operations + assemble (replicated) -- both event-timed in THIS run on rank 0's one-GPU reference context --
            # and X = what is left of the measured sharded step (exchange, launches around it, the non-pipelined hand-over)
            st1 = (parity or {}).get("single_gpu_stage_us") or {}
            W = st1.get("factors", float("nan")) * 1e-3
            R = (st1.get("chain", float("nan")) + st1.get("assemble", 0.0)) * 1e-3
            X = out["ms_per_step"] - (W / world + R)
            out["strong_scaling_model"] = {"W_ms_sharded": W, "R_ms_replicated": R, "X_ms_exchange": X,
                                           "expected_speedup_at_n": (W + R) / (W / world + R + max(X, 0.0)),
                                           "constants": "W, R: gvi_profile_stages on rank 0's unsharded context in this run; X = ms_per_step - (W / N + R)",
                                           "note": "t(N) = W / N + R + X; the ceiling as N grows is (W + R) / (R + X)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(chain, args.cpu_seconds, dom if not chain_pattern else 0)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
