"""Sharded HIP path on ONE GPU: two processes (two contexts on cuda:0), each owning half of the factors, exchanging
through torch.distributed -- gloo here, because RCCL refuses two ranks on one device; the driver's N>1 runs use
RCCL on one GPU per rank with the same code.  The replicated state after the iterations must equal the
single-process device run (and thereby the oracle, tests/test_gpu_parity.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from chains import make_chain

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, iters, speculate, q):
    import torch.distributed as dist
    from gaussianvi_amd import api
    from gaussianvi_amd.dist import HipEngine, ShardedNGD, shard_chain
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ch = make_chain(name)
        ctx, ids = api.context_for_chain(shard_chain(ch, rank, world), device=0)
        ngd = ShardedNGD(HipEngine(ctx, 0), world=world)
        ngd.speculate = speculate
        ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
        log = [ngd.step(40.0 if it == 1 else 0.55, 10) for it in range(iters)]     # one huge base step: rejected trials
        st = ctx.ngd_get_state()
        q.put((rank, log, st["mu"], st["D"], st["SigD"]))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("speculate", [True, False])
@pytest.mark.parametrize("name,iters", [("c2", 4), ("c3small", 3)])
def test_two_ranks_on_one_gpu_match_single_process(name, iters, speculate):
    from gaussianvi_amd import api
    ch = make_chain(name)
    ctx, ids = api.context_for_chain(ch)
    ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
    ref_log = [ctx.ngd_step(40.0 if it == 1 else 0.55, 10) for it in range(iters)]
    ref = ctx.ngd_get_state()
    ctx.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, name, iters, speculate, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, log, mu, D, SigD in res:
        for a, b in zip(log, ref_log):
            assert a["accepted"] == b["accepted"] and a["ntrials"] == b["ntrials"]
            assert np.isclose(a["new_cost"], b["new_cost"], rtol=1e-10)
        s = np.abs(ref["mu"]).max()
        assert np.abs(mu - ref["mu"]).max() < 1e-9 * s
        assert np.abs(D - ref["D"]).max() < 1e-9 * np.abs(ref["D"]).max()
        assert np.abs(SigD - ref["SigD"]).max() < 1e-9 * np.abs(ref["SigD"]).max()
    # the replicated chain state is bit-identical across ranks (same all-reduced inputs, same chain code)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def _worker_inlib(rank, world, port, name, iters, fuse, q):
    """Same problem, exchange INSIDE the library (gvi_dist_init_callback with a gloo all-gather): one gvi_ngd_step per
    iteration, no Python between the halves."""
    import torch.distributed as dist
    from gaussianvi_amd import api
    from gaussianvi_amd.dist import HipEngine, shard_chain, torch_allgather
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ch = make_chain(name)
        ctx, ids = api.context_for_chain(shard_chain(ch, rank, world), device=0)
        eng = HipEngine(ctx, 0)                                   # puts the library on a torch stream
        ctx.dist_init_callback(rank, world, torch_allgather(0))
        ctx.ngd_set_mode(True, fuse)
        ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
        log = [ctx.ngd_step(40.0 if it == 1 else 0.55, 10) for it in range(iters)]
        st = ctx.ngd_get_state()
        q.put((rank, log, st["mu"], st["D"], st["SigD"], ctx.dist_info()))
        ctx.close()
        del eng
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fuse", [0, 2])
@pytest.mark.parametrize("name,iters,world", [("c2", 4, 2), ("c3small", 3, 3), ("planar", 3, 4), ("c5small", 2, 2)])
def test_in_library_exchange_matches_single_process(name, iters, world, fuse):
    """exchange 0 as an all-gather of each rank's state records (+ the one state neighbours share),
    exchange 1 as an all-gather of the partial cost sums, both issued by the library itself.  2, 3 and 4 ranks (4 ranks on
    the planar graph leave two ranks with an EMPTY shard of the two-anchor set) against the single-process run."""
    from gaussianvi_amd import api
    ch = make_chain(name)
    ctx, ids = api.context_for_chain(ch)
    ctx.ngd_set_mode(True, fuse)
    ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
    ref_log = [ctx.ngd_step(40.0 if it == 1 else 0.55, 10) for it in range(iters)]
    ref = ctx.ngd_get_state()
    ctx.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_inlib, args=(r, world, port, name, iters, fuse, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = ch["T"]
    for rank, log, mu, D, SigD, info in res:
        # a rank sends the state range its factors touch: ~T / world + 1 on a chain pattern (the planar graph's two-anchor
        # set puts state T-1 on rank 1, which stretches that rank's range)
        assert info["world"] == world and 0 < info["records_per_rank"] <= T
        if name != "planar":
            assert info["records_per_rank"] <= (T + world - 1) // world + 2
        for a, b in zip(log, ref_log):
            assert a["accepted"] == b["accepted"] and a["ntrials"] == b["ntrials"]
            assert np.isclose(a["new_cost"], b["new_cost"], rtol=1e-10)
        s = np.abs(ref["mu"]).max()
        assert np.abs(mu - ref["mu"]).max() < 1e-9 * s
        assert np.abs(D - ref["D"]).max() < 1e-9 * np.abs(ref["D"]).max()
        assert np.abs(SigD - ref["SigD"]).max() < 1e-9 * np.abs(ref["SigD"]).max()
    for r in res[1:]:                                  # replicated state: bit-identical across ranks
        assert np.array_equal(res[0][2], r[2]) and np.array_equal(res[0][3], r[3])


def test_rccl_transport_with_one_rank():
    """The RCCL transport itself (dlopen of librccl, ncclCommInitRank, ncclAllGather on the context stream) with a
    communicator of size one -- the only size a one-GPU box allows; N > 1 over xGMI is first run by the driver."""
    from gaussianvi_amd import api
    ch = make_chain("c2")
    ctx, ids = api.context_for_chain(ch)
    ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
    ref = [ctx.ngd_step(0.55, 10) for _ in range(3)]
    ref_mu = ctx.ngd_get_state()["mu"]
    ctx.close()
    ctx, ids = api.context_for_chain(ch)
    ctx.dist_init_rccl(0, 1, api.dist_unique_id())
    ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
    got = [ctx.ngd_step(0.55, 10) for _ in range(3)]
    for a, b in zip(got, ref):
        assert a["accepted"] == b["accepted"] and np.isclose(a["new_cost"], b["new_cost"], rtol=1e-12)
    assert np.abs(ctx.ngd_get_state()["mu"] - ref_mu).max() < 1e-12 * np.abs(ref_mu).max()
    assert ctx.dist_info()["records_per_rank"] == ch["T"]
    ctx.close()


def test_bench_two_rank_rehearsal_reproduces_the_single_gpu_cost():
    """Guard for the first real multi-GPU run: `bench.py --gpus 2` (strong scaling = BASELINE configs[3], the same 1024-factor
    chain) rehearsed with both ranks on cuda:0 (GVI_BENCH_REHEARSAL=1: gloo callback transport) must end at the same
    final_cost as the single-process run, and says what it sharded."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "12", "--warmup", "3", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    env = dict(os.environ, GVI_BENCH_REHEARSAL="1", GVI_BENCH_C5_CONFIG="c5small")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                        capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["metric"] == one["metric"]
    assert "rehearsal" in two and "all-gather" in two["config"]["sharding"]
    # what the first real N > 1 run will be read by: who took part, over which transport, the expected ceiling, and the
    # sharded showcase block (rehearsed on the small d = 24 chain)
    assert "rccl_ranks" in two and "transport" in two and "gloo" in two["transport"]
    assert two["strong_scaling_model"]["expected_speedup_at_n"] > 0
    assert two["c5_strong"]["config"] == "c5small" and two["c5_strong"]["accepted_steps"] == two["c5_strong"]["steps"] == 2
    assert two["c5_strong"]["factors_per_rank"] == [16, 16]
    # the N > 1 line validates itself: the sharded steps against the same steps unsharded on rank 0's GPU, for both workloads,
    # and the model's W / R come from this run's own stage times
    for blk in (two["parity_vs_single_gpu"], two["c5_strong"]["parity_vs_single_gpu"]):
        assert blk["rel_gap_cost"] < 1e-9 and blk["rel_gap_mu"] < 1e-9 and blk["same_accept_decisions"], blk
    assert two["parity_vs_single_gpu"]["single_gpu_stage_us"]["factors"] > 0
    assert two["strong_scaling_model"]["W_ms_sharded"] > 0 and two["strong_scaling_model"]["R_ms_replicated"] > 0
    assert "generated on rank 0 and broadcast" in two["c5_strong"]["table"]
    assert one["c3_literal"]["steps"] == 20 and one["c3_literal"]["accepted_steps"] >= 1 and len(one["config"]["workload"]) < 120
    assert abs(two["final_cost"] - one["final_cost"]) < 1e-10 * abs(one["final_cost"])
    assert two["accepted_steps"] == one["accepted_steps"] == 12
