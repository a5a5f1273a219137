#!/usr/bin/env python3
"""Benchmark of the NGD Gauss-Hermite hot path on MI355X (contract: see the round prompt).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the synthetic
1024-factor LTV-prior chain, d = 12, sparse-GH p = 5 (N = 17 217 sigma points per factor), T = 1025
states of size 6, fp64, plus one unary d = 6 measurement factor per state (see
gaussianvi_amd/synthetic.py for why).  One STEP = one NGD iteration of GVIGH::optimize
(gvibase/GVI-GH-impl.h:39-118) entirely on the device: moments pass of every factor, ordered
assemble, block-tridiagonal solve, then line-search trials (axpy, chain factorisation = log-det +
marginals, gather, cost pass) until the first accepted one.  Inputs are resident in HBM before the
timed region.  `value` = every psi evaluation executed in the timed region (one per (factor, sigma
point) per pass, counted once) / wall time, whole job.

N > 1: the factor list is sharded contiguously over the ranks; the assembled [g | D | U] partials and the trial
cost are all-reduced over RCCL; the chain recursions are replicated.  Default --scaling weak: 1024 factors per
GPU (a 1024 N-factor chain, "c3xN"), per-GPU work fixed; --scaling strong: BASELINE configs[3], the same
1024-factor chain over N GPUs (latency-bound: 128 factors per GPU at N = 8).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_PEAK = 78.6e12        # FLOP/s, AMD public spec sheet (vector = matrix fp64 on MI355X); not in the guide


def api_count(d, p):
    from gaussianvi_amd import api
    return api.spgh_count(d, p)


def cpu_baseline(chain, seconds):
    """Reference-shaped CPU port (oracle/c/gvi_oracle.c: per-factor symmetric sqrt + expand, three
    Integrate passes with psi re-evaluated through a function pointer, OpenMP over factors) timed on
    this box's host cores on a bounded sample of the SAME workload: the first factors of the prior set
    at their start-state marginals (as many as ~`seconds` of CPU work allow, at most 256), repeated."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import gvi_oracle as o
    spec = chain["specs"][0]
    d, n = spec["d"], chain["n"]
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
    Z, w = o.nwspgr(d, spec["p"])
    threads = c_oracle.max_threads()
    # size the sample from a probe of one factor per thread (all threads)
    Kp = min(Kall, threads, len(mk))
    t0 = time.perf_counter()
    c_oracle.moments(Z, w, mk[:Kp], Sk[:Kp], spec["kind"], spec["params"][:Kp], n, fused=False)
    t_probe = max(time.perf_counter() - t0, 1e-6)
    K = int(max(1, min(256, Kall, len(mk), Kp * max(1, int(seconds * 0.1 / t_probe)))))
    out = {}
    for name, fused in (("reference_style", False), ("fused", True)):
        c_oracle.moments(Z, w, mk[:min(8, K)], Sk[:min(8, K)], spec["kind"], spec["params"][:min(8, K)], n, fused=fused)   # warm
        reps, t0 = 0, time.perf_counter()
        budget = seconds * (0.7 if not fused else 0.3)
        while True:
            c_oracle.moments(Z, w, mk[:K], Sk[:K], spec["kind"], spec["params"][:K], n, fused=fused)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget:
                break
        out[name] = K * len(w) * reps / el
    return {"value": out["reference_style"], "unit": "psi-evals/s", "cores": threads, "kind": "port",
            "sample": f"{K} of the {Kall} d={d} p={spec['p']} prior factors x {len(w)} sigma points, full moments pass "
                      f"(3 Integrate passes, psi x3 per point) repeated for ~{seconds:.0f} s, OpenMP {threads} threads",
            "fused_single_pass_value": out["fused"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 generic, 2 register (LDS operands), 3 operand-resident, 5 register (SGPR operands)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak",
                    help="weak (default): 1024 factors PER GPU, i.e. a (1024 N)-factor chain (config c3xN) sharded over the N "
                         "ranks -- per-GPU work fixed; strong (BASELINE configs[3]): the same 1024-factor chain over N GPUs "
                         "(128 factors per GPU at N = 8: bounded by the replicated chain recursions and two collectives)")
    ap.add_argument("--restart-every", type=int, default=30,
                    help="re-initialise (mu0, precision0) inside the timed region every R steps so that every step is a "
                         "descending iteration with one accepted trial (the chain converges after ~35 steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from gaussianvi_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):          # checkout without build artefacts: build once (rank 0 first)
        if rank == 0:
            build.build_lib()
        if use_pg:
            dist.barrier()
    from gaussianvi_amd import api, synthetic
    from gaussianvi_amd.dist import HipEngine, ShardedNGD, shard_chain

    if args.scaling == "weak" and world > 1 and args.config == "c3":
        args.config = f"c3x{world}"
    chain = synthetic.make_chain(args.config)
    local = shard_chain(chain, rank, world)
    ctx, ids = api.context_for_chain(local, device=local_rank)
    ctx.set_variant(args.variant)
    engine = HipEngine(ctx, local_rank)
    ngd = ShardedNGD(engine, world=world)
    ngd.group_forced = use_pg and world == 1
    ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
    ctx.profile_enable(3)       # HIP events around every 8th dominant launch (a pair costs ~14 us of queue gaps)

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    evals_moments = sum(K * N for (K, d, p, N) in ctx.sets)       # local, per pass
    # one process: the whole iteration (backtracking loop included) is one C-ABI call; sharded: the
    # Python driver interleaves the two all-reduces between the *_local / *_finish halves
    step_fn = (lambda: ctx.ngd_step(0.55, 10)) if (world == 1 and not ngd.group_forced) else (lambda: ngd.step(0.55, 10))
    def restart():
        ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
        ngd.reset()

    # one-time costs (buffer growth on the first cost pass, side stream / event creation, kernel attribute calls, the
    # cold Jacobi start) are primed outside the contract's W warm-up steps as well, so a small W does not time them
    for i in range(max(0, 8 - args.warmup)):
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
    t0 = time.perf_counter()
    passes, kern_ms, log = 0, [], []
    for i in range(args.steps):
        if pos == args.restart_every:
            restart(); pos = 0                          # timed: host upload + one refresh of the chain products
        r = step_fn(); pos += 1
        log.append(r)
        passes += 1 + r["ntrials"]
        if i % 8 == 7 or i == args.steps - 1:
            kern_ms.append(ctx.profile_last(ids[0], 0))      # last bracketed launch (sampled: every 8th)
    barrier()
    elapsed = time.perf_counter() - t0

    # extra (not part of `value`): the fused single-pass scheduling of the same iteration
    fused = None
    if world == 1 and not ngd.group_forced:
        ctx.ngd_set_mode(True, True)
        ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"])
        fpos = 0
        for _ in range(max(1, args.warmup)):
            if fpos == args.restart_every:
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"]); fpos = 0
            ctx.ngd_step(0.55, 10); fpos += 1
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        flog = []
        for i in range(args.steps):
            if fpos == args.restart_every:
                ctx.ngd_init(chain["mu0"], chain["D0"], chain["U0"]); fpos = 0
            flog.append(ctx.ngd_step(0.55, 10)); fpos += 1
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf0
        fused = {"ms_per_step": 1e3 * tf / args.steps, "ngd_iters_per_s": args.steps / tf,
                 "final_cost": flog[-1]["new_cost"], "trials_per_step": float(np.mean([r["ntrials"] for r in flog])),
                 "note": "gvi_ngd_set_mode(fuse_trial=1): one psi pass per accepted iteration (trial cost = m0 of the "
                         "full moments pass that is also the next gradient pass); identical iterates"}
        ctx.ngd_set_mode(True, False)

    stats = torch.tensor([elapsed, float(passes * evals_moments)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed, total_evals = float(tmax[0]), float(stats[1])
    else:
        total_evals = float(stats[1])

    if rank == 0:
        K0, d0, p0, N0 = ctx.sets[0]
        geo = ctx.profile_geometry(ids[0])
        km = float(np.mean(kern_ms)) * 1e-3
        alg_bytes = K0 * N0 * (d0 + 1) * 8                      # (d+1) s bytes per eval
        n_half = d0 // 2
        f_alg = 2 * (d0 * d0 + 2 * n_half * n_half + 2 * n_half + 1 + d0 + d0 * (d0 + 1) // 2)
        sreg_shapes = (4, 8, 12)
        if geo["variant"] == 5:
            kernel_name = f"moments_sreg_pair_kernel<{d0}, {d0 // 2}, {d0 // 2}, {d0 // 2}, full>"
        elif geo["variant"] == 2:
            kernel_name = {0: f"moments_sreg_kernel<{d0}, {d0 // 2}, full>" if d0 in sreg_shapes else f"moments_reg_kernel<{d0}, PsiQuad<{d0},{d0 // 2}>, full>",
                           5: f"moments_sreg_kernel<{d0}, {d0 // 2}, full>", 2: f"moments_reg_kernel<{d0}, PsiQuad<{d0},{d0 // 2}>, full>",
                           3: f"moments_wide_kernel<{d0}, PsiQuad<{d0},{d0 // 2}>, full>",
                           4: f"moments_tile_kernel<{d0}, PsiQuad<{d0},{d0 // 2}>, full>"}.get(args.variant, "moments_reg_kernel")
        elif geo["variant"] == 3:
            kernel_name = f"moments_split_kernel<{d0}, {(d0 // 2 + 3) // 4}, full>"
        else:
            kernel_name = "moments_generic_kernel"
        flop_launch, evals_launch = f_alg * K0 * N0, K0 * N0
        exec_ops = n_half * d0 + 2 * n_half + 2 + 2 * d0 + d0 * (d0 + 1) // 2   # psi rows (+ square, sign), c = w psi, m0, t = c z, m1, packed M2
        fused_pair = geo["variant"] == 5 and len(ctx.sets) == 2
        if fused_pair:
            # the timed launch also carries the unary set (d = n, psi = (x-mu0)^T Kinv (x-mu0): d^2 + d for psi):
            # SURVEY 8(d) formula with the unary residual count
            K1, d1, p1, N1 = ctx.sets[1]
            f_alg1 = 2 * (d1 * d1 + d1 * d1 + d1 + 1 + d1 + d1 * (d1 + 1) // 2)
            flop_launch += f_alg1 * K1 * N1
            evals_launch += K1 * N1
            alg_bytes += K1 * N1 * (d1 + 1) * 8
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "sigma-point psi-evals/sec + NGD iters/sec, 1024-factor d=12 p=5 chain" if args.config.startswith("c3")
                      else f"sigma-point psi-evals/sec + NGD iters/sec, {chain['T'] - 1}-factor d={d0} p={p0} chain",
            "value": total_evals / elapsed, "unit": "psi-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, **({"rehearsal": "all ranks on cuda:0, gloo exchange: not a measurement"} if rehearsal else {}), "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {chain['T'] - 1}-factor prior chain, d={d0}, sparse-GH p={p0} (N={N0}), "
                                   f"T={chain['T']} n={chain['n']}, +{chain['T']} unary d={chain['n']} factors; one step = one device-resident NGD iteration "
                                   f"(state re-initialised inside the timed region every {args.restart_every} steps)",
                       "name": args.config, "factor_sets": [list(map(int, s)) for s in ctx.sets] if world == 1 else None,
                       "sharding": f"factors/{world} contiguous, all-reduce [g|D|U] + trial cost (RCCL)" if world > 1 else "none",
                       "kernel_variant": geo["variant"], "chunks_per_factor": geo["nchunk"]},
            "ngd_iters_per_s": args.steps / elapsed,
            "accepted_steps": int(sum(r["accepted"] for r in log)),
            "trials_per_step": float(np.mean([r["ntrials"] for r in log])),
            "final_cost": log[-1]["new_cost"],
            "fused_trial_mode": fused,
            "moments_kernel": {"ms": km * 1e3, "psi_evals_per_s": evals_launch / km,
                               "launch": "prior set + unary set in one launch" if fused_pair else "prior set"},
            # The dominant kernel streams only the (d,p) table, which is L2-resident (profiles/r01_traffic.json:
            # HBM traffic ~0.5 % of the algorithmic bytes), so the binding roof is the fp64 FMA pipe, not HBM.
            # The schema's compute label is "mfma"; fp64 MFMA and fp64 VALU share one pipe on MI355X and the
            # VALU form is the faster one (profiles/r01_fp64_pipes.txt), so the kernel uses v_fma_f64.
            "roofline": {"bound": "mfma", "achieved": flop_launch / km / 1e12, "peak": FP64_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": flop_launch / km / FP64_PEAK, "traffic": traffic,
                         "kernel": kernel_name,
                         "note": "achieved/frac use SURVEY 8(d)'s ALGORITHMIC count (the reference's x-space expand + psi + three "
                                 "moment passes = 638 flop per evaluation).  The kernel's z-space formulation executes 188 fp64 "
                                 "FMA/MUL (376 flop) per evaluation, so frac can exceed 1; executed_tflops / 78.6 is the "
                                 "fraction of the pipe actually used.",
                         "algorithmic_flop_per_eval": f_alg, "executed_fp64_ops_per_eval": exec_ops,
                         "executed_tflops": 2 * exec_ops * K0 * N0 / km / 1e12,   # prior set only
                         "peak_source": "AMD MI355X spec sheet: 78.6 TF fp64 (vector = matrix); not in the local guide. "
                                        "Measured on this box (tools/ubench/fp64_pipes.hip): 70 TF v_fma_f64, 48 TF v_mfma_f64_16x16x4",
                         "hbm_algorithmic": {"bytes_per_launch": alg_bytes, "achieved_GBps": alg_bytes / km / 1e9,
                                             "peak_GBps": HBM_PEAK / 1e9, "frac": alg_bytes / km / HBM_PEAK,
                                             "note": "BASELINE's '>= 60 % of HBM roofline' figure: (d+1)*8 B per eval over 8 TB/s; "
                                                     "exceeds 1 because the table is served from L2"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(chain, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
