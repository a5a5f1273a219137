"""Pins the CPU oracle (oracle/gvi_oracle.py) against every known answer the reference holds for the
path: K1-K9 of SURVEY.md section 4.  CPU only."""
import os

import numpy as np
import pytest

import gvi_oracle as o


def _csv(golden_dir, name):
    return np.loadtxt(os.path.join(golden_dir, "ref_1d", name + ".csv"), delimiter=",")


# K1 -- tests/test_spgh_table_IO.cpp:68-90
def test_k1_table_5_2():
    Z, w = o.nwspgr(5, 2)
    assert Z.shape == (11, 5)
    expect = np.zeros((11, 5))
    for r in range(5):
        expect[r, r] = -1.0
        expect[10 - r, r] = 1.0
    assert np.linalg.norm(Z - expect) <= 1e-6
    assert np.linalg.norm(w - np.array([.5] * 5 + [-4.0] + [.5] * 5)) <= 1e-6


# K2 -- tests/test_GH.cpp:79-88 and nwspgr.m:301-303
def test_k2_rule_1_10():
    Z, w = o.nwspgr(1, 10)
    nodes = np.array([0.4849357075, 1.4659890944, 2.4843258416, 3.5818234836, 4.8594628283])
    wts = np.array([0.3446423349, 0.1354837030, 0.0191115805, 7.5807093e-4, 4.3106526e-6])
    assert Z.shape == (10, 1)
    assert np.allclose(Z[5:, 0], nodes, atol=1e-9) and np.allclose(Z[:5, 0], -nodes[::-1], atol=1e-9)
    assert np.allclose(w[5:], wts, atol=1e-9) and np.allclose(w[:5], wts[::-1], atol=1e-9)
    # independent check of the data table: Golub-Welsch (probabilists' Hermite)
    x, ww = np.polynomial.hermite_e.hermegauss(10)
    assert np.allclose(Z[:, 0], x, rtol=0, atol=1e-13)
    assert np.allclose(w, ww / ww.sum(), rtol=1e-12)


@pytest.mark.parametrize("level", range(1, 26))
def test_gqn_table_against_golub_welsch(level):
    n, w = o.gqn(level)
    x, ww = np.polynomial.hermite_e.hermegauss(level)
    ww = ww / ww.sum()
    half = x[x > -1e-14] if level % 2 else x[x > 0]
    assert np.allclose(n, np.abs(half), atol=2e-13)
    assert np.allclose(w, ww[-len(n):], rtol=1e-10, atol=1e-25)


# K3 -- tests/test_GH.cpp:134-161 (psi of test_GH.cpp:20-33: y = f b / mu_p + 0.05)
def test_k3_1d_nonlinear():
    psi = lambda x: o.psi_range_1d(x, y=40.0 / 20.0 + 0.05)
    gh = o.SparseGaussHermite(6, 1, np.array([20.0]), np.array([[9.0]]))
    assert abs(gh.Integrate(lambda x: np.array([[psi(x)]]))[0, 0] - 1.1129) <= 1e-4
    assert abs(gh.Integrate(lambda x: np.array([[(x[0] - 20.0) * psi(x)]]))[0, 0] + 1.2144) <= 1e-4


def _ph22(x):
    return np.array([[3.0 * x[0] * x[0]], [2.0 * x[0] * x[1]]])


# K4 -- tests/test_GH.cpp:164-183
def test_k4_2d():
    cov = np.array([[2.210433244916004, 1.635720601237843], [1.635720601237843, 2.210433244916004]])
    gh = o.SparseGaussHermite(10, 2, np.array([1.0, 1.0]), cov)
    r = gh.Integrate(_ph22)[:, 0]
    assert np.linalg.norm(r - np.array([9.631450087970276, 5.271519032251217])) <= 1e-3


# K5 -- tests/test_gh_spgh.cpp:145-162
def test_k5_2d_deg25():
    cov = np.linalg.inv(np.array([[1.0, -0.74], [-0.74, 1.0]]))
    gh = o.SparseGaussHermite(25, 2, np.array([1.0, 1.0]), cov)
    r = gh.Integrate(_ph22)[:, 0]
    assert np.linalg.norm(r - np.array([9.6313, 5.27144])) <= 1e-4


# K6 -- tests/test_gh_spgh.cpp:194-220
def test_k6_3d():
    gh = o.SparseGaussHermite(8, 3, np.ones(3), np.eye(3))
    r = gh.Integrate(lambda x: np.array([[1e4 * float(x @ x)]]))[0, 0]
    assert abs(r - 6.0e4) <= 1e-7 * 6e4


# K7 -- tests/test_gh_spgh.cpp:76-90 (dense class there; valid for sparse (4,3))
def test_k7_4d():
    gh = o.SparseGaussHermite(3, 4, np.zeros(4), 1e-4 * np.eye(4))
    r = gh.Integrate(lambda x: np.array([[1e4 * float(x @ x)]]))[0, 0]
    assert abs(r - 4.0) <= 1e-10


# K8 -- src/1d_example.cpp + data/1d/*.csv (golden NGD trace)
@pytest.mark.parametrize("variant", ["gbp", "takahashi"])
def test_k8_golden_ngd_trace(golden_dir, variant):
    f = o.NGDFactorizedBaseGH(1, 1, 10, o.psi_range_1d, 1, 0, 1.0, 10.0)
    opt = o.NGDGH([f], 1, 1, 10, variant=variant)
    opt.set_niter_low_temperature(10)
    opt.set_initial_values(np.array([20.0]), np.array([[1.0 / 9.0]]))
    opt.set_step_size_base(0.75)
    opt.optimize()
    rec = opt.record
    assert len(rec["cost"]) == 10
    assert np.abs(np.ravel(rec["mean"]) - _csv(golden_dir, "mean")).max() < 1e-12
    assert np.abs(np.ravel(rec["precision"]) - _csv(golden_dir, "precision")).max() < 1e-13
    assert np.abs(np.ravel(rec["cov"]) - _csv(golden_dir, "cov")).max() < 1e-12
    assert np.abs(np.ravel(rec["cost"]) - _csv(golden_dir, "cost").ravel()).max() < 1e-13
    assert np.abs(np.ravel(rec["factor_costs"]) - _csv(golden_dir, "factor_costs")).max() < 1e-12


def test_k8_costmap(golden_dir):
    f = o.NGDFactorizedBaseGH(1, 1, 10, o.psi_range_1d, 1, 0, 1.0, 10.0)
    opt = o.NGDGH([f], 1, 1, 10)
    cm = opt.cost_map(18, 25, 0.05, 1, 40)
    ref = _csv(golden_dir, "costmap")
    assert ref.shape == (40, 40)
    assert np.abs(cm - ref).max() < 1e-12 * np.abs(ref).max()


# K9 -- quadratic prior: sparse GH deg >= 3 equals the closed form of ngd/NGDFactorizedLinear.h:93-129
@pytest.mark.parametrize("nd,deg", [(1, 3), (2, 3), (2, 4)])
def test_k9_quadratic_prior_closed_form(nd, deg):
    rng = np.random.default_rng(7 + nd)
    n, d = 2 * nd, 4 * nd
    Phi, Qinv = o.minimum_acc_phi_qinv(np.eye(nd) * 0.8, 0.1)
    fac = o.NGDFactorizedBaseGH(d, n, deg, lambda x: o.psi_quad_prior(x, Phi, Qinv), 2, 0, 2.5, 10.0)
    mu = rng.normal(size=d)
    A = rng.normal(size=(d, d))
    Sigma = A @ A.T / d + 0.2 * np.eye(d)
    fac.update_mu_from_joint(mu)
    fac.update_precision_from_joint(Sigma)
    fac.calculate_partial_V()
    Lambda = np.hstack([-Phi, np.eye(n)])
    Ephi, Vdmu, Vddmu = o.linear_factor_closed_form(mu, Sigma, np.linalg.inv(Sigma), Lambda, Qinv,
                                                    np.zeros(n), 0.5, 2.5)
    assert np.allclose(fac._Vdmu, Vdmu, rtol=1e-11, atol=1e-11 * np.abs(Vdmu).max())
    assert np.allclose(fac._Vddmu, Vddmu, rtol=1e-9, atol=1e-10 * np.abs(Vddmu).max())
    assert np.isclose(fac.fact_cost_value(mu, Sigma), Ephi, rtol=1e-12)


def test_k9_degree2_is_not_exact_for_vddmu():
    """deg 2 integrates the 4th-order integrand wrongly -- guards against a silently
    'exact-by-construction' oracle."""
    Phi, Qinv = o.minimum_acc_phi_qinv(np.eye(1) * 0.8, 0.1)
    fac = o.NGDFactorizedBaseGH(4, 2, 2, lambda x: o.psi_quad_prior(x, Phi, Qinv), 2, 0, 1.0, 10.0)
    rng = np.random.default_rng(3)
    mu = rng.normal(size=4)
    A = rng.normal(size=(4, 4))
    Sigma = A @ A.T / 4 + 0.2 * np.eye(4)
    fac.update_mu_from_joint(mu)
    fac.update_precision_from_joint(Sigma)
    fac.calculate_partial_V()
    _, _, Vddmu = o.linear_factor_closed_form(mu, Sigma, np.linalg.inv(Sigma),
                                              np.hstack([-Phi, np.eye(2)]), Qinv, np.zeros(2), 0.5, 1.0)
    assert np.abs(fac._Vddmu - Vddmu).max() > 1e-2 * np.abs(Vddmu).max()


def test_prox_jko_reference_trace(golden_dir):
    """SURVEY 8(f)4 pin: the oracle's proximal (JKO) restatement reproduces the reference's committed 1-D prox run
    data/1d_proxgvi/{mean,precision,cost,factor_costs}.csv (src/1d_example_proxGVI.cpp, 10 iterations)."""
    import os
    g = os.path.join(golden_dir, "ref_1d_proxgvi")
    mean = np.loadtxt(os.path.join(g, "mean.csv"), delimiter=",").ravel()
    prec = np.loadtxt(os.path.join(g, "precision.csv"), delimiter=",").ravel()
    cov = np.loadtxt(os.path.join(g, "cov.csv"), delimiter=",").ravel()
    cost = np.loadtxt(os.path.join(g, "cost.csv")).ravel()
    fc = np.loadtxt(os.path.join(g, "factor_costs.csv"), delimiter=",").ravel()

    class FS:
        pass
    fs = FS()
    fs.d, fs.start = 1, np.array([0])
    fs.Z, fs.w = o.nwspgr(1, 10)
    fs.psi_batch = o.psi_batch_range_1d(400 * 0.1 / 20 - 0.8)
    ch = o.ChainProx(1, 1, [fs], np.array([[20.0]]), np.array([[[1 / 9.0]]]), np.zeros((0, 1, 1)), step_size_base=0.75)
    for it in range(10):
        c = ch.cost_value(ch.mu, ch.D, ch.U, ch.SigD, ch.SigU)
        f = ch.factor_costs(ch.mu, ch.SigD, ch.SigU)[0][0]
        assert abs(ch.mu[0, 0] - mean[it]) < 1e-12 and abs(ch.D[0, 0, 0] - prec[it]) < 1e-14
        assert abs(ch.SigD[0, 0, 0] - cov[it]) < 1e-12 and abs(c - cost[it]) < 1e-13 and abs(f - fc[it]) < 1e-13
        ch.step()

