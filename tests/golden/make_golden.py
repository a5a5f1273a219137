#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the repo's own oracle (there is nothing of the reference to
import: it is C++ and cannot be built here -- SURVEY.md section 8(c)).  Small fixtures:

  spgh_tables.npz : full (Z, w, idx) for (1,10), (5,2), (4,3), (2,10); first/last 8 rows, size,
                    checksums of (12,5)                       -> pins the PRODUCT generator (host C++)
  k9_moments.npz  : seeded (mu, Sigma, Phi, Qinv) -> (E_phi, Vdmu, Vddmu), d = 4 and d = 12,
                    GH moments and the closed form of ngd/NGDFactorizedLinear.h:93-129
  chain_step.npz  : one NGD iteration of the 'tiny' and 'c3mini' synthetic chains (gradients, step,
                    accepted cost, new mean / precision / marginals)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import gvi_oracle as o  # noqa: E402
from chains import make_chain  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def idx_checksum(idx):
    """Order-sensitive 64-bit FNV-1a over the int8 index stream."""
    h = np.uint64(0xcbf29ce484222325)
    prime = np.uint64(0x100000001b3)
    with np.errstate(over="ignore"):
        for b in idx.astype(np.uint8).reshape(-1):
            h = (h ^ np.uint64(b)) * prime
    return int(h)


def tables():
    out = {}
    for d, p in [(1, 10), (5, 2), (4, 3), (2, 10), (6, 5)]:
        Z, w, idx = o.nwspgr(d, p, True)
        out[f"Z_{d}_{p}"], out[f"w_{d}_{p}"], out[f"idx_{d}_{p}"] = Z, w, idx
    Z, w, idx = o.nwspgr(12, 5, True)
    out["N_12_5"] = np.array(Z.shape[0])
    out["Zhead_12_5"], out["Ztail_12_5"] = Z[:8], Z[-8:]
    out["whead_12_5"], out["wtail_12_5"] = w[:8], w[-8:]
    out["idxsum_12_5"] = np.array(idx_checksum(idx), dtype=np.uint64)
    out["wabs_12_5"] = np.array(np.abs(w).sum())
    out["wneg_12_5"] = np.array((w < 0).sum())
    out["Zcolabs_12_5"] = np.abs(Z).sum(axis=0)
    out["moment2_12_5"] = np.einsum("n,na,nb->ab", w, Z, Z)
    np.savez_compressed(os.path.join(G, "spgh_tables.npz"), **out)


def k9():
    out = {}
    for nd, p in [(1, 3), (3, 5)]:
        n, d = 2 * nd, 4 * nd
        rng = np.random.default_rng(90 + d)
        K = 5
        Phi = np.stack([np.eye(n) + 0.1 * rng.normal(size=(n, n)) for _ in range(K)])
        Qh = rng.normal(size=(K, n, n))
        Qinv = Qh @ np.transpose(Qh, (0, 2, 1)) + 0.5 * np.eye(n)
        mu = rng.normal(size=(K, d))
        B = rng.normal(size=(K, d, d))
        Sigma = 0.3 * (B @ np.transpose(B, (0, 2, 1)) / d + 0.2 * np.eye(d))
        temp = rng.uniform(0.5, 5.0, K)
        Z, w = o.nwspgr(d, p)
        r = o.batched_moments(Z, w, mu, Sigma, o.psi_batch_quad_prior(Phi, Qinv), temp)
        cf = [o.linear_factor_closed_form(mu[k], Sigma[k], np.linalg.inv(Sigma[k]),
                                          np.hstack([-Phi[k], np.eye(n)]), Qinv[k], np.zeros(n), 0.5, temp[k])
              for k in range(K)]
        tag = f"d{d}"
        out.update({f"{tag}_p": np.array(p), f"{tag}_Phi": Phi, f"{tag}_Qinv": Qinv, f"{tag}_mu": mu,
                    f"{tag}_Sigma": Sigma, f"{tag}_temp": temp,
                    f"{tag}_cost": r["cost"], f"{tag}_Vdmu": r["Vdmu"], f"{tag}_Vddmu": r["Vddmu"],
                    f"{tag}_E_phi": r["E_phi"], f"{tag}_E_xmuphi": r["E_xmuphi"], f"{tag}_E_xxphi": r["E_xxphi"],
                    f"{tag}_cf_cost": np.array([c[0] for c in cf]),
                    f"{tag}_cf_Vdmu": np.stack([c[1] for c in cf]),
                    f"{tag}_cf_Vddmu": np.stack([c[2] for c in cf])})
    np.savez_compressed(os.path.join(G, "k9_moments.npz"), **out)


def chain_step():
    out = {}
    for name in ["tiny", "c3mini"]:
        ch = make_chain(name)
        c = o.ChainNGD(ch["T"], ch["n"], ch["oracle_sets"](), ch["mu0"], ch["D0"], ch["U0"])
        cost0 = c.cost_value(c.mu, c.D, c.U, c.SigD, c.SigU)
        dmu, dD, dU, (g, Dv, Uv) = c.gradients()
        ok, cost1, ntr = c.step()
        out.update({f"{name}_cost0": np.array(cost0), f"{name}_dmu": dmu, f"{name}_dD": dD, f"{name}_dU": dU,
                    f"{name}_g": g, f"{name}_VD": Dv, f"{name}_VU": Uv, f"{name}_ok": np.array(ok),
                    f"{name}_cost1": np.array(cost1), f"{name}_ntrials": np.array(ntr),
                    f"{name}_mu1": c.mu, f"{name}_D1": c.D, f"{name}_U1": c.U,
                    f"{name}_SigD1": c.SigD, f"{name}_SigU1": c.SigU})
    np.savez_compressed(os.path.join(G, "chain_step.npz"), **out)


if __name__ == "__main__":
    tables(); k9(); chain_step()
    for f in sorted(os.listdir(G)):
        p = os.path.join(G, f)
        if os.path.isfile(p):
            print(f, os.path.getsize(p))
