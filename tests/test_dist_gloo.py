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
