"""Test-side glue: synthetic chains (gaussianvi_amd.synthetic) + the oracle's psi closures."""
import functools

import numpy as np

import gvi_oracle as o
from gaussianvi_amd import synthetic as syn


def oracle_psi_batch(spec):
    if spec["kind"] == syn.PSI_QUAD_PRIOR:
        return o.psi_batch_quad_prior(spec["Phi"], spec["Qinv"])
    if spec["kind"] == syn.PSI_FIXED_PRIOR:
        return o.psi_batch_fixed_prior(spec["mu0"], spec["Kinv"])
    if spec["kind"] == syn.PSI_HINGE_SDF_3D_ARM:
        return o.psi_batch_hinge_sdf3d_arm(spec["params"], spec["arm"], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    if spec["kind"] == syn.PSI_HINGE_SDF_2D_BODY:
        return o.psi_batch_hinge_sdf2d_body(spec["params"], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    if spec["kind"] == syn.PSI_HINGE_SDF_3D:
        return o.psi_batch_hinge_sdf3d(spec["params"], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    if spec["kind"] == syn.PSI_HINGE_SDF_2D:
        return o.psi_batch_hinge_sdf2d(spec["params"], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
    if spec["kind"] == syn.PSI_RANGE_1D:
        y, mu_p, fb, srs, sps = spec["params"][0]
        return o.psi_batch_range_1d(y, mu_p, fb, srs, sps)
    raise ValueError(spec["kind"])


def oracle_psi_point(spec):
    def make(k):
        if spec["kind"] == syn.PSI_QUAD_PRIOR:
            return lambda x: o.psi_quad_prior(x, spec["Phi"][k], spec["Qinv"][k])
        if spec["kind"] == syn.PSI_FIXED_PRIOR:
            return lambda x: o.psi_fixed_prior(x, spec["mu0"][k], spec["Kinv"][k])
        raise ValueError(spec["kind"])
    return make


def make_chain(name):
    if name == "planar":
        ch = syn.make_planar_chain()
    elif name in ("quad2d", "pr3d"):
        ch = syn.make_obstacle_chain(name)
    elif name == "arm7":
        ch = syn.make_obstacle_chain(name, T=5)
    else:
        ch = syn.make_chain(name)
    for spec in ch["specs"]:
        spec["psi_batch"] = oracle_psi_batch(spec)
        if spec["kind"] < syn.PSI_HINGE_SDF_2D:
            spec["psi_point"] = oracle_psi_point(spec)

    def oracle_sets(fast=False):
        """fast: quadratic kinds go through the oracle's C restatement (oracle/c, OpenMP) instead of numpy -- the same
        checker, pinned against the numpy one in tests/test_oracle_c.py; for full-size multi-iteration runs."""
        out = []
        for spec in ch["specs"]:
            fs = o.FactorSet(spec["start"], spec["d"], spec["p"], spec["psi_batch"])
            fs.temperature = np.asarray(spec["temperature"], dtype=np.float64)
            if fast and spec["kind"] in (syn.PSI_QUAD_PRIOR, syn.PSI_FIXED_PRIOR):
                import c_oracle
                n_arg = ch["n"] if spec["kind"] == syn.PSI_QUAD_PRIOR else spec["d"]
                fs.fast_moments = (lambda sp, fs_, n_: lambda mk, Sk, temp: c_oracle.moments(
                    fs_.Z, fs_.w, mk, Sk, sp["kind"], sp["params"], n_, temp, fused=False))(spec, fs, n_arg)
            out.append(fs)
        return out
    ch["oracle_sets"] = oracle_sets
    return ch


def oracle_table(d, p):
    """o.nwspgr memoised for the session: (24,5) takes tens of seconds in numpy."""
    return o.nwspgr_cached(d, p)
