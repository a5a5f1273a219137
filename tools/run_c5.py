"""BASELINE configs[4] at full size on one MI355X: 4096-factor d=24 chain, sparse-GH p=7
(N = 20 557 057 sigma points per factor, 8.42e10 psi-evals per pass), fp64 on the 8-bit coded table.

Prints one JSON object: table build time, kernel times of the full moments pass and the cost pass,
closed-form parity of sampled factors (GH at p >= 3 is exact for a quadratic psi), and the wall time
of NGD iterations.  Usage: python tools/run_c5.py [K] [p] [iters]   (defaults 4096 7 2)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gaussianvi_amd import api, synthetic as syn

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = int(sys.argv[2]) if len(sys.argv) > 2 else 7
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2

syn.CONFIGS["c5run"] = (5, K + 1, 12, p, "ltv", 5)
t0 = time.time()
ch = syn.make_chain("c5run")
t_chain = time.time() - t0
print(f"chain built in {t_chain:.1f}s", flush=True)
t0 = time.time()
ctx, ids = api.context_for_chain(ch)
t_table = time.time() - t0
Kp, d, pp, N = ctx.sets[ids[0]]
print(f"context + tables ({d},{pp}) N={N} in {t_table:.1f}s", flush=True)
ctx.profile_enable(2)

# operator level: moments of the prior set at seeded marginals, closed-form check on a sample
rng = np.random.default_rng(5)
mu, Sigma = syn.random_marginals(rng, K, d, 0.3)
t0 = time.time()
Ephi, Vdmu, Vddmu = ctx.moments(ids[0], mu, Sigma)
t_call = time.time() - t0
ms_full = ctx.profile_last(ids[0], 0)
cost = ctx.costs(ids[0], mu, Sigma)
ms_cost = ctx.profile_last(ids[0], 1)
print(f"moments kernel {ms_full:.1f} ms ({K*N/ms_full/1e6:.2f} Gevals/s), cost kernel {ms_cost:.1f} ms "
      f"({K*N/ms_cost/1e6:.2f} Gevals/s)", flush=True)
spec = ch["specs"][0]
worst = 0.0
for k in np.linspace(0, K - 1, 8).astype(int):
    # analytic Gaussian moments of psi = 1/2 (Lam x)^T Qinv (Lam x):  E = 1/2 (tr(M Sigma) + r^T Qinv r),
    # Vdmu = M mu, Vddmu = M with M = Lam^T Qinv Lam (what ngd/NGDFactorizedLinear.h:93-129 evaluates)
    Lam = np.hstack([-spec["Phi"][k], np.eye(12)])
    M = Lam.T @ spec["Qinv"][k] @ Lam
    r = Lam @ mu[k]
    c, vd, vdd = 0.5 * (np.trace(M @ Sigma[k]) + r @ spec["Qinv"][k] @ r), M @ mu[k], M
    worst = max(worst, abs(cost[k] - c) / abs(c), np.abs(Vdmu[k] - vd).max() / np.abs(vd).max(),
                np.abs(Vddmu[k] - vdd).max() / np.abs(vdd).max())
print(f"closed-form parity (8 sampled factors): worst relative error {worst:.2e}", flush=True)

# NGD iterations on the resident chain
ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
steps = []
for it in range(iters):
    t0 = time.time()
    r = ctx.ngd_step(0.55, 10)
    steps.append(dict(seconds=time.time() - t0, accepted=bool(r["accepted"]), ntrials=int(r["ntrials"]), cost=float(r["new_cost"])))
    print(steps[-1], flush=True)
out = dict(config=f"c5: {K}-factor d=24 chain, sparse-GH p={p}, N={N}, T={K+1}, n=12, fp64, coded table",
           chain_build_s=t_chain, table_build_upload_s=t_table, evals_per_pass=K * N,
           moments_kernel_ms=ms_full, moments_evals_per_s=K * N / ms_full * 1e3,
           cost_kernel_ms=ms_cost, cost_evals_per_s=K * N / ms_cost * 1e3,
           fp64_tflops_algorithmic=2426 * K * N / ms_full * 1e3 / 1e12,
           closed_form_worst_rel_err=worst, ngd_steps=steps, geometry=ctx.profile_geometry(ids[0]))
print(json.dumps(out))
