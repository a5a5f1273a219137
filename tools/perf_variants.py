"""Moments / cost kernel time of the C3 prior set per kernel variant (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from gaussianvi_amd import api, synthetic as syn

ch = syn.make_chain(sys.argv[1] if len(sys.argv) > 1 else "c3")
spec = ch["specs"][0]
ctx = api.Context(0)
ctx.chain_set(ch["T"], ch["n"])
sid = ctx.factors_add(spec["d"], spec["p"], spec["start"], spec["kind"], spec["params"], spec["temperature"])
K, d, p, N = ctx.sets[sid]
rng = np.random.default_rng(0)
mu, Sigma = syn.random_marginals(rng, K, d, 0.3)
ctx.profile_enable(2)
for v in (2, 3, 5, 0):
    ctx.set_variant(v)
    best = [1e9, 1e9]
    for it in range(5):
        ctx.moments(sid, mu, Sigma); best[0] = min(best[0], ctx.profile_last(sid, 0))
        ctx.costs(sid, mu, Sigma); best[1] = min(best[1], ctx.profile_last(sid, 1))
    print(f"variant {v}: moments {best[0]*1e3:.1f} us ({K*N/best[0]/1e6:.1f} Gevals/s)  cost {best[1]*1e3:.1f} us ({K*N/best[1]/1e6:.1f} Gevals/s)", flush=True)
