"""The C restatement (oracle/c/gvi_oracle.c, used as bench.py's cpu_baseline) agrees with the numpy
oracle for every psi kind, in both its reference-shaped and fused variants.  CPU only."""
import numpy as np
import pytest

import c_oracle
import gvi_oracle as o
from gaussianvi_amd import synthetic as syn


@pytest.mark.parametrize("fused", [False, True])
def test_c_oracle_quad_prior(fused):
    rng = np.random.default_rng(1)
    K, n, p = 5, 3, 4
    d = 2 * n
    Phi = np.stack([np.eye(n) + 0.1 * rng.normal(size=(n, n)) for _ in range(K)])
    Qh = rng.normal(size=(K, n, n))
    Qinv = Qh @ np.transpose(Qh, (0, 2, 1)) + 0.5 * np.eye(n)
    params = np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1)
    mu, Sigma = syn.random_marginals(rng, K, d, 0.3)
    temp = rng.uniform(0.5, 3, K)
    Z, w = o.nwspgr(d, p)
    r = o.batched_moments(Z, w, mu, Sigma, o.psi_batch_quad_prior(Phi, Qinv), temp)
    E, Vd, Vdd = c_oracle.moments(Z, w, mu, Sigma, syn.PSI_QUAD_PRIOR, params, n, temp, fused=fused, nthreads=2)
    assert np.allclose(E, r["E_phi"], rtol=1e-11)
    assert np.allclose(Vd, r["Vdmu"], rtol=1e-9, atol=1e-11 * np.abs(r["Vdmu"]).max())
    assert np.allclose(Vdd, r["Vddmu"], rtol=1e-8, atol=1e-10 * np.abs(r["Vddmu"]).max())


def test_c_oracle_fixed_and_range():
    rng = np.random.default_rng(2)
    K, d, p = 3, 3, 3
    mu0 = rng.normal(size=(K, d))
    Kh = rng.normal(size=(K, d, d))
    Kinv = Kh @ np.transpose(Kh, (0, 2, 1)) + 0.3 * np.eye(d)
    mu, Sigma = syn.random_marginals(rng, K, d, 0.5)
    Z, w = o.nwspgr(d, p)
    r = o.batched_moments(Z, w, mu, Sigma, o.psi_batch_fixed_prior(mu0, Kinv), np.ones(K))
    E, Vd, Vdd = c_oracle.moments(Z, w, mu, Sigma, syn.PSI_FIXED_PRIOR,
                                  np.concatenate([mu0, Kinv.reshape(K, -1)], axis=1), d)
    assert np.allclose(E, r["E_phi"], rtol=1e-11) and np.allclose(Vdd, r["Vddmu"], rtol=1e-8, atol=1e-10)
    Z, w = o.nwspgr(1, 10)
    r = o.batched_moments(Z, w, np.array([[20.0]]), np.array([[[9.0]]]), o.psi_batch_range_1d(1.2), np.ones(1))
    E, Vd, Vdd = c_oracle.moments(Z, w, np.array([[20.0]]), np.array([[[9.0]]]), syn.PSI_RANGE_1D,
                                  np.array([[1.2, 20.0, 40.0, 0.09, 9.0]]), 1)
    assert np.isclose(E[0], r["E_phi"][0], rtol=1e-13) and np.isclose(Vdd[0, 0, 0], r["Vddmu"][0, 0, 0], rtol=1e-11)


@pytest.mark.parametrize("fused", [False, True])
def test_c_oracle_hinge_sdf2d(fused):
    """The planar hinge-on-SDF port (bench.py's cpu_baseline for --config planar1k) against the numpy oracle: marginals
    near and inside the obstacles, so the hinge is active on part of the sigma points."""
    ch = syn.make_planar_chain(T=9, p=3)
    spec = ch["specs"][1]
    K, d = len(spec["start"]), spec["d"]
    rng = np.random.default_rng(7)
    mu = ch["mu0"][spec["start"]] + 0.3 * rng.normal(size=(K, d))
    mu[:3, :2] = [[0.0, 0.9], [-1.0, -1.6], [0.4, 1.2]]              # at the discs' rims
    _, Sigma = syn.random_marginals(rng, K, d, 0.2)
    Z, w = o.nwspgr(d, spec["p"])
    r = o.batched_moments(Z, w, mu, Sigma, o.psi_batch_hinge_sdf2d(spec["params"], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"]),
                          np.ones(K))
    c_oracle.set_sdf2d(spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    E, Vd, Vdd = c_oracle.moments(Z, w, mu, Sigma, syn.PSI_HINGE_SDF_2D, spec["params"], ch["n"], fused=fused, nthreads=2)
    assert np.abs(r["E_phi"]).max() > 1e-3                            # the hinge is active
    assert np.allclose(E, r["E_phi"], rtol=1e-11, atol=1e-13)
    assert np.allclose(Vd, r["Vdmu"], rtol=1e-9, atol=1e-11 * np.abs(r["Vdmu"]).max())
    assert np.allclose(Vdd, r["Vddmu"], rtol=1e-8, atol=1e-10 * np.abs(r["Vddmu"]).max())
