"""The committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) still match
the oracle, and the K9 fixtures match the reference's closed form.  CPU only."""
import os

import numpy as np

import gvi_oracle as o
from chains import make_chain


def test_tables_match_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "spgh_tables.npz"))
    for d, p in [(1, 10), (5, 2), (4, 3), (2, 10), (6, 5)]:
        Z, w, idx = o.nwspgr(d, p, True)
        assert np.array_equal(Z, g[f"Z_{d}_{p}"]) and np.array_equal(idx, g[f"idx_{d}_{p}"])
        assert np.allclose(w, g[f"w_{d}_{p}"], rtol=1e-13, atol=1e-15)
    assert int(g["N_12_5"]) == 17217
    assert g["Z_4_3"].shape == (41, 4) and (g["w_4_3"] < 0).sum() == 8


def test_table_moments_identity(golden_dir):
    """Exactness sanity: a (d,p>=2) rule reproduces E[z z^T] = I."""
    g = np.load(os.path.join(golden_dir, "spgh_tables.npz"))
    assert np.allclose(g["moment2_12_5"], np.eye(12), atol=1e-9)


def test_k9_fixture_matches_closed_form(golden_dir):
    g = np.load(os.path.join(golden_dir, "k9_moments.npz"))
    for tag in ["d4", "d12"]:
        sV = np.abs(g[f"{tag}_cf_Vddmu"]).max()
        assert np.allclose(g[f"{tag}_cost"], g[f"{tag}_cf_cost"], rtol=1e-10)
        assert np.allclose(g[f"{tag}_Vdmu"], g[f"{tag}_cf_Vdmu"], rtol=1e-8, atol=1e-10 * np.abs(g[f"{tag}_cf_Vdmu"]).max())
        assert np.allclose(g[f"{tag}_Vddmu"], g[f"{tag}_cf_Vddmu"], rtol=1e-6, atol=1e-9 * sV)


def test_chain_step_fixture_matches_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "chain_step.npz"))
    ch = make_chain("tiny")
    c = o.ChainNGD(ch["T"], ch["n"], ch["oracle_sets"](), ch["mu0"], ch["D0"], ch["U0"])
    dmu, dD, dU, _ = c.gradients()
    assert np.allclose(dmu, g["tiny_dmu"], rtol=1e-9, atol=1e-12)
    ok, cost1, ntr = c.step()
    assert ok and ntr == int(g["tiny_ntrials"]) and np.isclose(cost1, float(g["tiny_cost1"]), rtol=1e-12)
    assert np.allclose(c.mu, g["tiny_mu1"], rtol=1e-10, atol=1e-12)
