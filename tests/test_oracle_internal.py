"""Self-consistency of the oracle's two restatements (per-point reference-shaped classes vs the
batched/block forms used as the checker at C2/C3 sizes) and of the block-tridiagonal routines
against dense numpy linear algebra.  CPU only."""
import numpy as np
import pytest

import gvi_oracle as o
from chains import make_chain


def _spd_bt(T, n, rng):
    A = np.zeros((T * n, T * n))
    for i in range(T - 1 if T > 1 else 1):
        w = 2 * n if T > 1 else n
        B = rng.normal(size=(w, w))
        A[i * n:i * n + w, i * n:i * n + w] += B @ B.T / w + 0.3 * np.eye(w)
    return A


@pytest.mark.parametrize("T,n", [(1, 1), (1, 3), (2, 2), (7, 3), (20, 6)])
def test_bt_routines_match_dense(T, n):
    rng = np.random.default_rng(T * 10 + n)
    A = _spd_bt(T, n, rng)
    D, U = o.dense_to_bt(A, n)
    assert np.allclose(o.bt_to_dense(D, U), A)
    Sig = np.linalg.inv(A)
    SD, SU = o.inverse_gbp(D, U)
    eD, eU = o.dense_to_bt(Sig, n)
    assert np.allclose(SD, eD, rtol=1e-9, atol=1e-12) and np.allclose(SU, eU, rtol=1e-9, atol=1e-12)
    # the two reference variants (Takahashi selected inverse, GBP) agree on the pattern
    Zt = o.inv_sparse_takahashi(A, n)
    tD, tU = o.dense_to_bt(Zt, n)
    assert np.allclose(tD, eD, rtol=1e-9, atol=1e-12) and np.allclose(tU, eU, rtol=1e-9, atol=1e-12)
    # pivots / log-det
    _, dv = o.ldlt_pivots_dense(A)
    assert np.allclose(o.bt_ldlt_pivots(D, U), dv, rtol=1e-10)
    assert np.isclose(o.logdet_half(dv), np.linalg.slogdet(A)[1] / 2, rtol=1e-12)
    # solve: direct == Eigen-style CG == dense
    rhs = rng.normal(size=T * n)
    x = np.linalg.solve(A, rhs)
    assert np.allclose(o.bt_solve(D, U, rhs), x, rtol=1e-9, atol=1e-12)
    assert np.allclose(o.cg_eigen(A, rhs), x, rtol=1e-7, atol=1e-10)


def test_logdet_nan_on_indefinite():
    A = np.array([[1.0, 2.0], [2.0, 1.0]])
    _, dv = o.ldlt_pivots_dense(A)
    assert np.isnan(o.logdet_half(dv))
    D, U = o.dense_to_bt(A, 1)
    assert np.isnan(o.logdet_half(o.bt_ldlt_pivots(D, U)))


def test_batched_moments_match_reference_shaped_factor():
    rng = np.random.default_rng(11)
    Phi, Qinv = o.minimum_acc_phi_qinv(np.eye(1) * 0.8, 0.1)
    K, d, n, p = 3, 4, 2, 3
    Z, w = o.nwspgr(d, p)
    mu = rng.normal(size=(K, d))
    B = rng.normal(size=(K, d, d))
    Sigma = B @ np.transpose(B, (0, 2, 1)) / d + 0.2 * np.eye(d)
    temp = np.array([1.0, 2.0, 10.0])
    r = o.batched_moments(Z, w, mu, Sigma, o.psi_batch_quad_prior(np.stack([Phi] * K), np.stack([Qinv] * K)), temp)
    for k in range(K):
        f = o.NGDFactorizedBaseGH(d, n, p, lambda x: o.psi_quad_prior(x, Phi, Qinv), 2, 0, temp[k], 10.0)
        f.update_mu_from_joint(mu[k])
        f.update_precision_from_joint(Sigma[k])
        f.calculate_partial_V()
        assert np.allclose(r["Vdmu"][k], f._Vdmu, rtol=1e-10, atol=1e-12)
        assert np.allclose(r["Vddmu"][k], f._Vddmu, rtol=1e-9, atol=1e-10)
        assert np.isclose(r["cost"][k], f.fact_cost_value(mu[k], Sigma[k]), rtol=1e-12)


def test_chain_ngd_matches_dense_ngdgh_on_small_chain():
    """Block-level iteration (ChainNGD) == dense-joint reference-shaped optimiser (NGDGH) on a
    T=5 minimum-acceleration chain with two fixed-prior anchors."""
    ch = make_chain("tiny")
    T, n = ch["T"], ch["n"]
    sets = ch["oracle_sets"]()
    chain = o.ChainNGD(T, n, sets, ch["mu0"], ch["D0"], ch["U0"])
    facs = []
    for spec in ch["specs"]:
        for k, s in enumerate(spec["start"]):
            facs.append(o.NGDFactorizedBaseGH(spec["d"], n, spec["p"], spec["psi_point"](k), T, int(s), 1.0, 10.0))
    opt = o.NGDGH(facs, n, T, 3, solver="cg")
    opt.set_initial_values(ch["mu0"].reshape(-1), o.bt_to_dense(ch["D0"], ch["U0"]))
    for it in range(2):
        c0 = opt.cost_value()
        dmu, dprec = opt.compute_gradients()
        cdmu, cdD, cdU, _ = chain.gradients()
        assert np.allclose(cdmu.reshape(-1), dmu, rtol=1e-7, atol=1e-9)
        assert np.allclose(o.bt_to_dense(cdD, cdU), dprec, rtol=1e-8, atol=1e-9 * np.abs(dprec).max())
        assert np.isclose(chain.cost_value(chain.mu, chain.D, chain.U), c0, rtol=1e-11)
        ok, cost, ntr = chain.step()
        step = 0.55
        for _ in range(ntr):
            step *= 0.75
        nc, nm, nprec = opt.onestep_linesearch(step, dmu, dprec)
        assert ok and np.isclose(nc, cost, rtol=1e-10)
        opt.set_mu(nm)
        opt.set_precision(nprec)
        assert np.allclose(chain.mu.reshape(-1), opt._mu, rtol=1e-9, atol=1e-11)
