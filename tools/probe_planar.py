"""Per-block wall times of the planar1k iteration (restart every 6 steps): where the bimodal bench numbers come from."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from gaussianvi_amd import api, synthetic
ch = synthetic.make_chain("planar1k")
ctx, ids = api.context_for_chain(ch)
ctx.ngd_set_mode(True, 2)
def run(nblocks, use_run, prof):
    ctx.profile_enable(prof)
    t_init, t_blk = [], []
    for b in range(nblocks):
        t0 = time.perf_counter(); ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"]); t1 = time.perf_counter()
        if use_run:
            ctx.ngd_run(6, 0.55, 10)
        else:
            for s in range(6):
                ctx.ngd_step(0.55, 10)
        if prof:
            try: ctx.profile_last(ids[0], 0)
            except api.GviError: pass
        t2 = time.perf_counter()
        t_init.append(t1 - t0); t_blk.append(t2 - t1)
    return 1e3 * np.mean(t_init), 1e3 * np.mean(t_blk), 1e3 * np.max(t_blk)
for use_run in (False, True):
    for prof in (0, 3):
        for nb in (20, 80, 80):
            print("ngd_run" if use_run else "ngd_step", "profile", prof, "blocks", nb, "init %.3f ms, block %.3f ms (max %.3f)" % run(nb, use_run, prof), flush=True)
ctx.close()
