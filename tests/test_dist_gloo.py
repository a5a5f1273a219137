"""N>1 path on CPU: world_size-2 gloo run of gaussianvi_amd.dist.ShardedNGD (factor sharding, the two
exchange steps, accept logic) equals the single-rank run and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gvi_oracle as o
from chains import make_chain
from cpu_engine import OracleEngine
from gaussianvi_amd.dist import ShardedNGD, shard_chain, shard_range


def test_shard_range_partitions():
    for K in [1, 2, 7, 64, 1024, 1025]:
        for world in [1, 2, 3, 8]:
            cuts = [shard_range(K, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_shard_chain_keeps_global_start_indices():
    ch = make_chain("tiny")
    parts = [shard_chain(ch, r, 2) for r in range(2)]
    for s in range(len(ch["specs"])):
        joined = np.concatenate([p["specs"][s]["start"] for p in parts])
        assert np.array_equal(joined, ch["specs"][s]["start"])
        assert sum(len(p["specs"][s]["params"]) for p in parts) == len(ch["specs"][s]["params"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, iters, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ch = make_chain(name)
        eng = OracleEngine(shard_chain(ch, rank, world))
        ngd = ShardedNGD(eng, world=world)
        log = [ngd.step(0.55, 10) for _ in range(iters)]
        q.put((rank, log, eng.mu[eng.cur], eng.D[eng.cur]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny", "c2"])
def test_two_rank_gloo_matches_single_rank_and_oracle(name):
    iters = 2
    ch = make_chain(name)
    single = ShardedNGD(OracleEngine(ch), world=1)
    ref_log = [single.step(0.55, 10) for _ in range(iters)]
    ref_eng = single.e
    chain = o.ChainNGD(ch["T"], ch["n"], ch["oracle_sets"](), ch["mu0"], ch["D0"], ch["U0"])
    for it in range(iters):
        ok, cost, ntr = chain.step()
        assert ref_log[it]["accepted"] == ok and ref_log[it]["ntrials"] == ntr
        assert np.isclose(ref_log[it]["new_cost"], cost, rtol=1e-11)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, iters, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, log, mu, D in res:
        for a, b in zip(log, ref_log):
            assert a["accepted"] == b["accepted"] and a["ntrials"] == b["ntrials"]
            assert np.isclose(a["new_cost"], b["new_cost"], rtol=1e-11)
        assert np.allclose(mu, ref_eng.mu[ref_eng.cur], rtol=1e-9, atol=1e-12)
        assert np.allclose(D, ref_eng.D[ref_eng.cur], rtol=1e-9, atol=1e-9)
    # both ranks hold bit-identical replicated state (same all-reduced inputs, same chain code)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def test_four_rank_gloo_with_uneven_and_empty_shards():
    """World size 4 on the 5-state chain: 4 priors -> one per rank; 5 unary factors -> 2 + 1 + 1 + 1 (uneven); and the
    planar graph's two-anchor set over four ranks leaves two ranks with an EMPTY shard of that set.  Every rank must end with
    the single-rank state."""
    iters = 2
    for name in ("tiny", "planar"):
        ch = make_chain(name)
        sizes = [[len(shard_chain(ch, r, 4)["specs"][s]["start"]) for r in range(4)] for s in range(len(ch["specs"]))]
        if name == "tiny":
            assert sizes[0] == [1, 1, 1, 1] and sizes[1] == [2, 1, 1, 1]
        else:
            assert sizes[2] == [1, 1, 0, 0]                              # empty shards of the anchor set
        single = ShardedNGD(OracleEngine(ch), world=1)
        ref_log = [single.step(0.55, 10) for _ in range(iters)]
        ref_eng = single.e
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 4, port, name, iters, q)) for r in range(4)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, log, mu, D in res:
            for a, b in zip(log, ref_log):
                assert a["accepted"] == b["accepted"] and a["ntrials"] == b["ntrials"]
                assert np.isclose(a["new_cost"], b["new_cost"], rtol=1e-10)
            assert np.allclose(mu, ref_eng.mu[ref_eng.cur], rtol=1e-9, atol=1e-12)
        for r in res[1:]:
            assert np.array_equal(res[0][2], r[2]) and np.array_equal(res[0][3], r[3])


def test_missing_rccl_library_is_an_error_not_a_crash(monkeypatch):
    """gvi_dist_unique_id with an RCCL library that cannot be loaded must come back as a GviError
    (bench.py's all-rank fallback to the callback transport depends on catching it); the old code built the message from a
    second dlerror() call, i.e. from NULL.  GVI_RCCL_PATH names the ONLY candidate when it is set."""
    from gaussianvi_amd import api
    monkeypatch.setenv("GVI_RCCL_PATH", "/nonexistent/librccl-not-here.so")
    with pytest.raises(api.GviError) as e:
        api.dist_unique_id()
    assert "librccl" in str(e.value) and "nonexistent" in str(e.value)


def _fallback_worker(rank, world, port, q):
    """bench.py's transport selection, as a function of what every rank reports: rank 1 cannot load librccl
    (GVI_RCCL_PATH=/nonexistent), so the MIN all-reduce of the `ok` flags must send ALL ranks to the callback transport."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 1:
        os.environ["GVI_RCCL_PATH"] = "/nonexistent/librccl.so"
    else:
        os.environ.pop("GVI_RCCL_PATH", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussianvi_amd import api
        ok = torch.ones(1, dtype=torch.float64)
        if rank == 1:                                    # (rank 0 stands for a rank whose library loaded: no RCCL call on a box without GPUs)
            try:
                api.dist_unique_id()
            except api.GviError:
                ok.zero_()
        mine = float(ok.item())
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        q.put((rank, mine, float(ok.item())))
    finally:
        dist.destroy_process_group()


def test_one_rank_without_rccl_sends_every_rank_to_the_callback_transport():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][1] == 0.0                              # rank 1 failed locally ...
    assert res[0][2] == 0.0 and res[1][2] == 0.0         # ... and both ranks agree on the fallback
