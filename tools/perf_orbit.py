"""Stand-alone time of the sign-orbit moments / cost kernel per factor set of a chain (GPU box): the walk without the
fused pass around it.  python tools/perf_orbit.py [config]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from gaussianvi_amd import api, synthetic as syn

ch = syn.make_chain(sys.argv[1] if len(sys.argv) > 1 else "c3")
ctx = api.Context(0)
ctx.chain_set(ch["T"], ch["n"])
rng = np.random.default_rng(0)
ctx.profile_enable(2)
for spec in ch["specs"]:
    sid = ctx.factors_add(spec["d"], spec["p"], spec["start"], spec["kind"], spec["params"], spec["temperature"])
    K, d, p, N = ctx.sets[sid]
    mu, Sigma = syn.random_marginals(rng, K, d, 0.3)
    best = [1e9, 1e9]
    for it in range(8):
        r = ctx.moments(sid, mu, Sigma); best[0] = min(best[0], ctx.profile_last(sid, 0))
        ctx.costs(sid, mu, Sigma); best[1] = min(best[1], ctx.profile_last(sid, 1))
    print(f"set {sid} (K={K}, d={d}, p={p}, N={N}): moments {best[0]*1e3:.2f} us ({K*N/best[0]/1e6:.1f} Gevals/s)  cost {best[1]*1e3:.2f} us  checksum {float(np.sum(r[2])):.12e}", flush=True)
