"""Throughput of the d = 24 split kernel on the GPU box (BASELINE configs[4] shapes)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gaussianvi_amd import api, synthetic as syn

K, p = int(sys.argv[1]), int(sys.argv[2])
d, n = 24, 12
rng = np.random.default_rng(1)
Phi = np.stack([np.eye(n) + 0.1 * rng.normal(size=(n, n)) for _ in range(K)])
Qh = rng.normal(size=(K, n, n)); Qinv = Qh @ np.transpose(Qh, (0, 2, 1)) + 0.5 * np.eye(n)
ctx = api.Context(0)
ctx.chain_set(2, n)
t0 = time.time()
sid = ctx.factors_add(d, p, np.zeros(K, dtype=np.int32), api.PSI_QUAD_PRIOR,
                      np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1), None)
N = ctx.sets[sid][3]
print(f"table ({d},{p}) N={N} built+uploaded in {time.time()-t0:.1f}s", flush=True)
mu, Sigma = syn.random_marginals(rng, K, d, 0.3)
ctx.profile_enable(2)
for it in range(3):
    t0 = time.time(); ctx.moments(sid, mu, Sigma); t1 = time.time()
    ms = ctx.profile_last(sid, 0)
    print(f"moments: kernel {ms:.2f} ms  {K*N/ms/1e6:.2f} Gevals/s  (call {1e3*(t1-t0):.1f} ms)  geom {ctx.profile_geometry(sid)}", flush=True)
for it in range(2):
    ctx.costs(sid, mu, Sigma)
    ms = ctx.profile_last(sid, 1)
    print(f"cost: kernel {ms:.2f} ms  {K*N/ms/1e6:.2f} Gevals/s", flush=True)
