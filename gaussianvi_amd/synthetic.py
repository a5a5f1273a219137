"""Deterministic synthetic factor chains for tests and bench.py.

Pure numpy/scipy input generation -- no oracle, no device code.  A chain is T states of size n with
block-tridiagonal precision; factors come in homogeneous *sets* (same kind, d, GH degree), the unit
the C-ABI works on.  psi parameter blocks use the C-ABI layouts of include/gvi_hip.h:

  QUAD_PRIOR  [Phi (n x n, row-major) | Qinv (n x n)]          d = 2n   (gp/minimum_acc_prior.h, gp/LTV_prior.h)
  FIXED_PRIOR [mu0 (d) | Kinv (d x d)]                         d = n    (gp/fixed_prior.h)
  RANGE_1D    [y, mu_p, f*b, sig_r_sq, sig_p_sq]               d = 1    (src/1d_example.cpp:25-35)
"""
from __future__ import annotations

import numpy as np

PSI_RANGE_1D, PSI_QUAD_PRIOR, PSI_FIXED_PRIOR, PSI_HOST_CALLBACK, PSI_HINGE_SDF_2D = 0, 1, 2, 3, 4
PSI_HINGE_SDF_2D_BODY, PSI_HINGE_SDF_3D, PSI_HINGE_SDF_3D_ARM = 5, 6, 7

CONFIGS = {
    # name: (cfg#, T, n, p, prior kind)
    "tiny": (0, 5, 2, 3, "minacc"),
    "c2": (2, 65, 2, 3, "minacc"),          # BASELINE.json configs[1]
    "c3mini": (31, 9, 6, 5, "ltv"),
    "c3small": (32, 33, 6, 5, "ltv"),
    "c3t2": (33, 2, 6, 5, "ltv"),           # shortest chains: one / two prior factors (tail blocks of the fused launches)
    "c3t3": (34, 3, 6, 5, "ltv"),
    "c3": (3, 1025, 6, 5, "ltv"),           # BASELINE.json configs[2] (headline)
    # the LITERAL SURVEY 8(d) C3 chain: dt = 0.05, seeded stable A_k (6 x 6) / B_k (6 x 3), two end anchors only.
    # cond(Hessian) ~ 1e7: kept for parity at operator level and for the checked conditioning bound (tests), not benched
    "c3lit": (36, 1025, 6, 5, "ltvlit"),
    "c3litmini": (37, 17, 6, 5, "ltvlit"),
    "c5mini": (51, 5, 12, 5, "ltv"),        # d = 24 slice (split kernel), N(24,5) = 243 905
    "c5small": (52, 33, 12, 6, "ltv", 5),   # d = 24, p = 6: N = 2 438 801; unary factors at p = 5
    "c5": (5, 4097, 12, 7, "ltv", 5),       # BASELINE.json configs[4]: N(24,7) = 20 557 057 (fp64, coded table)
}

# Physical scales.  They are chosen so that the joint Hessian is well conditioned (cond ~ 5e2 at
# T = 1025): a pure prior chain pinned only at its two ends has cond ~ 2e7 at T = 1025, dt = 0.05,
# and then NO two fp64 implementations of the reference algorithm (including the reference against
# itself under a different OpenMP schedule) agree to 1e-6 on the NGD iterate -- the solve
# amplifies the ~1e-11..1e-8 rounding of the GH sums (|w|_1 ~ 5e3 at (12,5)) by cond.  Hence every
# state also carries a unary Gaussian "measurement" factor (the batch-estimation graph of Barfoot
# et al. that src/1d_example.cpp cites): binary motion priors + unary factors on every state.
DT = {"minacc": 0.25, "ltv": 0.2}
QC = 0.8                  # minimum-acceleration power spectral density
B_SCALE = 3.0             # LTV input gain
KAPPA_UNARY, KAPPA_ANCHOR = 1.0, 50.0


def _minacc(nd, qc, dt):
    I = np.eye(nd)
    Phi = np.block([[I, dt * I], [np.zeros((nd, nd)), I]])
    iQc = np.eye(nd) / qc
    Qinv = np.block([[12 * iQc / dt ** 3, -6 * iQc / dt ** 2], [-6 * iQc / dt ** 2, 4 * iQc / dt]])
    return Phi, Qinv


def _ltv_system(rng, nd):
    """Four (A, B) pairs -- one per piece-wise-constant sub-interval -- of a seeded stable second-order system
    x'' = -Kp x - Kd x' + B2 u (state [x, x']): the hA / hB entries LTV_GP reads (gp/LTV_prior.h:57-61)."""
    out = []
    for _ in range(4):
        Kp = np.diag(rng.uniform(0.5, 2.0, nd)) + 0.1 * rng.normal(size=(nd, nd))
        Kd = np.diag(rng.uniform(0.5, 1.5, nd)) + 0.1 * rng.normal(size=(nd, nd))
        A = np.block([[np.zeros((nd, nd)), np.eye(nd)], [-Kp, -Kd]])
        B = np.vstack([np.zeros((nd, nd)), B_SCALE * (np.eye(nd) + 0.1 * rng.normal(size=(nd, nd)))])
        out.append((A, B))
    return out


def _ltv_from_system(system, dt):
    """Exact discretisation over the 4 sub-intervals (product of matrix exponentials, Van Loan block form for the Gramian):
    the maths of gp/LTV_prior.h:123-197."""
    from scipy.linalg import expm
    n = system[0][0].shape[0]
    Phi, Q = np.eye(n), np.zeros((n, n))
    h = dt / 4
    for A, B in system:
        M = np.zeros((2 * n, 2 * n))
        M[:n, :n], M[:n, n:], M[n:, n:] = -A, B @ B.T, A.T
        E = expm(M * h)
        Ad = E[n:, n:].T
        Qd = Ad @ E[:n, n:]
        Phi = Ad @ Phi
        Q = Ad @ Q @ Ad.T + (Qd + Qd.T) / 2
    return Phi, np.linalg.inv(Q)


def _ltv(rng, nd, dt):
    """(Phi, Q^-1) of one seeded LTV prior factor."""
    return _ltv_from_system(_ltv_system(rng, nd), dt)


def ltv_chain_system(name: str):
    """The (hA, hB) lists behind make_chain(name)'s LTV priors, in LTV_GP's layout: 4 (T - 1) + 1 matrices each, factor k
    reading entries 4k .. 4k + 4 (the 5th is only touched at t = delta_t exactly; here the next factor's first)."""
    cfg, T, n, p, kind = CONFIGS[name][:5]
    assert kind == "ltv"
    rng = np.random.default_rng(0x5EED + cfg)
    hA, hB = [], []
    for k in range(T - 1):
        for A, B in _ltv_system(rng, n // 2):
            hA.append(A); hB.append(B)
    hA.append(hA[-1].copy()); hB.append(hB[-1].copy())
    return np.stack(hA), np.stack(hB)


def _ltv_literal(rng, n, dt):
    """SURVEY 8(d) C3 as written: (Phi_k, Q_k^-1) = exact discretisation over 4 piece-wise-constant sub-intervals (product of
    matrix exponentials / Van Loan) of a seeded STABLE A_k (n x n, spectral radius <= 1) and B_k (n x n/2) -- the maths of
    gp/LTV_prior.h:123-197 without GSL."""
    from scipy.linalg import expm
    Phi, Q = np.eye(n), np.zeros((n, n))
    h = dt / 4
    for _ in range(4):
        G = rng.normal(size=(n, n))
        G = G - (np.linalg.eigvals(G).real.max() + 0.1) * np.eye(n)            # stable: Re(lambda) < 0
        A = G / max(1.0, np.abs(np.linalg.eigvals(G)).max())                   # spectral radius <= 1
        B = rng.normal(size=(n, n // 2))
        M = np.zeros((2 * n, 2 * n))
        M[:n, :n], M[:n, n:], M[n:, n:] = -A, B @ B.T, A.T
        E = expm(M * h)
        Ad = E[n:, n:].T
        Qd = Ad @ E[:n, n:]
        Phi = Ad @ Phi
        Q = Ad @ Q @ Ad.T + (Qd + Qd.T) / 2
    return Phi, np.linalg.inv(Q)


def make_literal_chain(name: str):
    """The literal BASELINE configs[2] chain of SURVEY 8(d): T - 1 LTV-form priors (d = 2n) and TWO FixedPriorGP end anchors
    (K0 = 1e-2 I) -- no unary factor on the interior states; start state = straight line + N(0, 0.05^2) jitter, initial
    precision = sum of the factor Hessians."""
    cfg, T, n, p, kind = CONFIGS[name][:5]
    rng = np.random.default_rng(0x5EED + cfg)
    K, dt = T - 1, 0.05
    Phi = np.zeros((K, n, n))
    Qinv = np.zeros((K, n, n))
    for k in range(K):
        Phi[k], Qinv[k] = _ltv_literal(rng, n, dt)
    goal = rng.uniform(1.0, 2.0, n)
    t = np.arange(T)[:, None] / (T - 1)
    nominal = goal[None, :] * t
    mu0 = nominal + 0.05 * rng.normal(size=nominal.shape)
    anchors = np.stack([nominal[0], nominal[-1]])
    Kinv = np.stack([np.eye(n) / 1e-2] * 2)
    D0 = np.zeros((T, n, n))
    U0 = np.zeros((T - 1, n, n))
    for k in range(K):
        Lam = np.hstack([-Phi[k], np.eye(n)])
        M = Lam.T @ Qinv[k] @ Lam
        D0[k] += M[:n, :n]
        D0[k + 1] += M[n:, n:]
        U0[k] += M[:n, n:]
    D0[0] += 2 * Kinv[0]
    D0[-1] += 2 * Kinv[1]
    specs = [
        dict(kind=PSI_QUAD_PRIOR, d=2 * n, p=p, start=np.arange(K, dtype=np.int32),
             params=np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1),
             temperature=np.ones(K), Phi=Phi, Qinv=Qinv),
        dict(kind=PSI_FIXED_PRIOR, d=n, p=p, start=np.array([0, T - 1], dtype=np.int32),
             params=np.concatenate([anchors, Kinv.reshape(2, -1)], axis=1), temperature=np.ones(2), mu0=anchors, Kinv=Kinv),
    ]
    return dict(name=name, T=T, n=n, specs=specs, mu0=mu0, D0=D0, U0=U0)


def make_chain(name: str):
    """Returns dict(T, n, specs, mu0, D0, U0).  specs[0]: the T-1 binary prior factors (d = 2n,
    QUAD_PRIOR); specs[1]: T unary measurement factors (d = n, FIXED_PRIOR), the first and the last
    being the strong end anchors.  "planar1k" / "planar": the planning graph with hinge-on-SDF obstacle factors
    (make_planar_chain); "arm7x": the 7-DOF arm graph (make_obstacle_chain) at T = 129."""
    if name == "planar1k":          # bench.py --config planar1k: 1025 states, 1025 obstacle factors d = 4 at p = 7 (2145 points)
        ch = make_planar_chain(T=1025, p=3, p_obstacle=7)
        ch["name"] = name
        # obstacle factors at the reference's HIGH temperature (gvibase/GVI-GH-impl.h:104-117 switches to it when the line search
        # is exhausted; its planar experiments start there): at T_k = 1 the hinge rejects every step after two iterations
        ch["specs"][1]["temperature"] = np.full(len(ch["specs"][1]["start"]), 30.0)
        return ch
    if name == "planar":
        return make_planar_chain()
    if name == "arm7x":
        ch = make_obstacle_chain("arm7", T=129)
        ch["name"] = name
        ch["specs"][1]["temperature"] = np.full(len(ch["specs"][1]["start"]), 30.0)
        return ch
    if name not in CONFIGS and name.startswith("c3x") and name[3:].isdigit():
        CONFIGS[name] = (3, 1024 * int(name[3:]) + 1, 6, 5, "ltv")      # weak-scaling family: 1024 factors per GPU
    if CONFIGS[name][4] == "ltvlit":
        return make_literal_chain(name)
    cfg, T, n, p, kind = CONFIGS[name][:5]
    p_unary = CONFIGS[name][5] if len(CONFIGS[name]) > 5 else p
    rng = np.random.default_rng(0x5EED + cfg)
    nd, K = n // 2, T - 1
    dt = DT[kind]
    Phi = np.zeros((K, n, n))
    Qinv = np.zeros((K, n, n))
    for k in range(K):
        Phi[k], Qinv[k] = _minacc(nd, QC, dt) if kind == "minacc" else _ltv(rng, nd, dt)
    # nominal trajectory: straight line with constant velocity; start state = nominal + jitter
    goal = rng.uniform(1.0, 2.0, nd)
    t = np.arange(T)[:, None] * dt
    horizon = (T - 1) * dt
    nominal = np.hstack([goal[None, :] * t / horizon, np.tile(goal / horizon, (T, 1))])
    mu0 = nominal + 0.05 * rng.normal(size=nominal.shape)
    meas = nominal + 0.1 * rng.normal(size=nominal.shape)     # unary factor means
    meas[0], meas[-1] = nominal[0], nominal[-1]
    kappa = np.full(T, KAPPA_UNARY)
    kappa[0] = kappa[-1] = KAPPA_ANCHOR
    Kinv = kappa[:, None, None] * np.eye(n)[None]
    # initial precision: 0.7 x (sum of factor Hessians) -- PD block-tridiagonal
    D0 = np.zeros((T, n, n))
    U0 = np.zeros((T - 1, n, n))
    for k in range(K):
        Lam = np.hstack([-Phi[k], np.eye(n)])
        M = Lam.T @ Qinv[k] @ Lam
        D0[k] += M[:n, :n]
        D0[k + 1] += M[n:, n:]
        U0[k] += M[:n, n:]
    D0 += 2 * Kinv
    D0, U0 = 0.7 * D0, 0.7 * U0
    specs = [
        dict(kind=PSI_QUAD_PRIOR, d=2 * n, p=p, start=np.arange(K, dtype=np.int32),
             params=np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1),
             temperature=np.ones(K), Phi=Phi, Qinv=Qinv),
        dict(kind=PSI_FIXED_PRIOR, d=n, p=p_unary, start=np.arange(T, dtype=np.int32),
             params=np.concatenate([meas, Kinv.reshape(T, -1)], axis=1),
             temperature=np.ones(T), mu0=meas, Kinv=Kinv),
    ]
    return dict(name=name, T=T, n=n, specs=specs, mu0=mu0, D0=D0, U0=U0)


def circle_sdf(origin, cell, rows, cols, centers, radii):
    """Signed distance to a union of discs on a grid: field[r, c] at (origin + (c, r) * cell)."""
    xs = origin[0] + np.arange(cols) * cell
    ys = origin[1] + np.arange(rows) * cell
    X, Y = np.meshgrid(xs, ys)
    f = np.full((rows, cols), np.inf)
    for (cx, cy), r in zip(centers, radii):
        f = np.minimum(f, np.hypot(X - cx, Y - cy) - r)
    return f


def make_planar_chain(T=17, p=3, seed=0x5EED + 40, jitter=0.02, p_obstacle=None, horizon=None):
    """Planar point-robot planning graph of the reference's own GPU workload: states
    [x, y, vx, vy] (n = 4), T-1 minimum-acceleration priors (d = 8), T hinge-on-SDF obstacle factors on
    every state (d = 4, helpers/CudaOperation.h:491-523) and two fixed-prior end anchors.  p_obstacle: GH degree of the
    obstacle factors (default p + 1); horizon: total time (default (T - 1) / 4, i.e. dt = 0.25)."""
    rng = np.random.default_rng(seed)
    n, nd, K = 4, 2, T - 1
    dt = 0.25 if horizon is None else horizon / (T - 1)
    Phi1, Qinv1 = _minacc(nd, QC, dt)
    Phi, Qinv = np.stack([Phi1] * K), np.stack([Qinv1] * K)
    start_xy, goal_xy = np.array([-3.0, -0.4]), np.array([3.0, 0.4])
    horizon = (T - 1) * dt
    vel = (goal_xy - start_xy) / horizon
    t = np.arange(T)[:, None] * dt
    nominal = np.hstack([start_xy[None] + vel[None] * t, np.tile(vel, (T, 1))])
    mu0 = nominal + jitter * rng.normal(size=nominal.shape)
    origin, cell, rows, cols = (-5.0, -4.0), 0.1, 81, 101
    field = circle_sdf(origin, cell, rows, cols, [(0.0, 1.6), (-1.0, -2.2)], [1.2, 0.9])
    anchors = np.stack([nominal[0], nominal[-1]])
    Kinv = np.stack([np.eye(n) / 1e-2] * 2)
    D0 = np.zeros((T, n, n)); U0 = np.zeros((T - 1, n, n))
    for k in range(K):
        Lam = np.hstack([-Phi[k], np.eye(n)])
        M = Lam.T @ Qinv[k] @ Lam
        D0[k] += M[:n, :n]; D0[k + 1] += M[n:, n:]; U0[k] += M[:n, n:]
    D0[0] += 2 * Kinv[0]; D0[-1] += 2 * Kinv[1]
    D0 += 0.5 * np.eye(n)
    specs = [
        dict(kind=PSI_QUAD_PRIOR, d=2 * n, p=p, start=np.arange(K, dtype=np.int32),
             params=np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1), temperature=np.ones(K), Phi=Phi, Qinv=Qinv),
        dict(kind=PSI_HINGE_SDF_2D, d=n, p=p + 1 if p_obstacle is None else p_obstacle, start=np.arange(T, dtype=np.int32),
             params=np.tile(np.array([[15.5, 0.5, 0.3]]), (T, 1)), temperature=np.ones(T),
             sdf_origin=origin, sdf_cell=cell, sdf_field=field),
        dict(kind=PSI_FIXED_PRIOR, d=n, p=p, start=np.array([0, T - 1], dtype=np.int32),
             params=np.concatenate([anchors, Kinv.reshape(2, -1)], axis=1), temperature=np.ones(2), mu0=anchors, Kinv=Kinv),
    ]
    return dict(name="planar", T=T, n=n, specs=specs, mu0=mu0, D0=D0, U0=U0)


def sphere_sdf3d(origin, cell, rows, cols, nz, centers, radii):
    """Signed distance to a union of balls: field[r, c, z] at origin + (c, r, z) * cell."""
    xs = origin[0] + np.arange(cols) * cell
    ys = origin[1] + np.arange(rows) * cell
    zs = origin[2] + np.arange(nz) * cell
    Y, X, Zz = np.meshgrid(ys, xs, zs, indexing="ij")
    f = np.full((rows, cols, nz), np.inf)
    for (cx, cy, cz), r in zip(centers, radii):
        f = np.minimum(f, np.sqrt((X - cx) ** 2 + (Y - cy) ** 2 + (Zz - cz) ** 2) - r)
    return f


def make_obstacle_chain(kind: str, T=9, p=3, seed=0x5EED + 41):
    """n = 6 planning graphs of the reference's other two obstacle workloads: "quad2d" (planar quadrotor,
    state [x, z, phi, vx, vz, w], body of 5 check points, helpers/CudaOperation.h:565-606) and "pr3d" (3-D point
    robot, state [x, y, z, v], trilinear field, :650-683): minimum-acceleration priors (d = 12), one obstacle
    factor per state (d = 6) and two end anchors."""
    rng = np.random.default_rng(seed + {"quad2d": 0, "pr3d": 1, "arm7": 2}[kind])
    nd = 7 if kind == "arm7" else 3
    n, K = 2 * nd, T - 1
    if kind == "arm7":
        p = 3                                    # d = 28 priors: N(28,3) = 1625 sigma points (p = 2 is not exact for Vddmu)
    dt = 0.25
    Phi1, Qinv1 = _minacc(nd, QC, dt)
    Phi, Qinv = np.stack([Phi1] * K), np.stack([Qinv1] * K)
    horizon = (T - 1) * dt
    if kind == "quad2d":
        start, goal = np.array([-3.0, -0.5, 0.2]), np.array([3.0, 0.6, -0.1])
        origin, cell = (-6.0, -5.0), 0.1
        field = circle_sdf(origin, cell, 101, 121, [(0.0, 2.2), (-1.0, -3.0)], [1.2, 0.9])
        obst = dict(kind=PSI_HINGE_SDF_2D_BODY, params=np.tile(np.array([[15.5, 0.5, 0.3, 5.0, 5.0, 1.0]]), (T, 1)))
    elif kind == "arm7":
        start = np.array([-0.8, 0.5, 0.3, 1.2, -0.4, 0.6, 0.0])
        goal = np.array([0.7, 0.9, -0.2, 0.8, 0.3, 0.2, 0.4])
        origin, cell = (-1.5, -1.5, -0.5), 0.05
        field = sphere_sdf3d(origin, cell, 61, 61, 41, [(0.4, 0.2, 0.5), (-0.3, -0.4, 0.3)], [0.25, 0.2])
        obst = dict(kind=PSI_HINGE_SDF_3D_ARM, params=np.tile(np.array([[15.5, 0.1]]), (T, 1)), arm=wam_like_arm())
    else:
        start, goal = np.array([-2.0, -0.5, 0.0]), np.array([2.0, 0.5, 0.6])
        origin, cell = (-4.0, -3.0, -2.0), 0.2
        field = sphere_sdf3d(origin, cell, 31, 41, 21, [(0.0, 1.4, 0.3), (-0.5, -1.8, 0.0)], [1.0, 0.8])
        obst = dict(kind=PSI_HINGE_SDF_3D, params=np.tile(np.array([[15.5, 0.5, 0.3]]), (T, 1)))
    vel = (goal - start) / horizon
    t = np.arange(T)[:, None] * dt
    nominal = np.hstack([start[None] + vel[None] * t, np.tile(vel, (T, 1))])
    mu0 = nominal + 0.02 * rng.normal(size=nominal.shape)
    anchors = np.stack([nominal[0], nominal[-1]])
    Kinv = np.stack([np.eye(n) / 1e-2] * 2)
    D0 = np.zeros((T, n, n)); U0 = np.zeros((T - 1, n, n))
    for k in range(K):
        Lam = np.hstack([-Phi[k], np.eye(n)])
        M = Lam.T @ Qinv[k] @ Lam
        D0[k] += M[:n, :n]; D0[k + 1] += M[n:, n:]; U0[k] += M[:n, n:]
    D0[0] += 2 * Kinv[0]; D0[-1] += 2 * Kinv[1]
    D0 += 0.5 * np.eye(n)
    specs = [
        dict(kind=PSI_QUAD_PRIOR, d=2 * n, p=p, start=np.arange(K, dtype=np.int32),
             params=np.concatenate([Phi.reshape(K, -1), Qinv.reshape(K, -1)], axis=1), temperature=np.ones(K), Phi=Phi, Qinv=Qinv),
        dict(d=n, p=p if kind == "arm7" else p + 1, start=np.arange(T, dtype=np.int32), temperature=np.ones(T),
             sdf_origin=origin, sdf_cell=cell, sdf_field=field, **obst),
        dict(kind=PSI_FIXED_PRIOR, d=n, p=p, start=np.array([0, T - 1], dtype=np.int32),
             params=np.concatenate([anchors, Kinv.reshape(2, -1)], axis=1), temperature=np.ones(2), mu0=anchors, Kinv=Kinv),
    ]
    return dict(name=kind, T=T, n=n, specs=specs, mu0=mu0, D0=D0, U0=U0)


def wam_like_arm(nspheres=16):
    """A 7-DOF arm in DH form (Barrett-WAM-like link lengths) with collision spheres on its frames -- the inputs of
    the reference's CudaOperation_3dArm / ForwardKinematics (helpers/CudaOperation.h:325-399, 686-716)."""
    rng = np.random.default_rng(0x5EED + 60)
    h = np.pi / 2
    frames = np.array([0, 1, 1, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 6, 6, 6][:nspheres], dtype=np.int32)
    return dict(a=np.array([0.0, 0.0, 0.045, -0.045, 0.0, 0.0, 0.0]), alpha=np.array([-h, h, -h, h, -h, h, 0.0]),
                d=np.array([0.0, 0.0, 0.55, 0.0, 0.3, 0.0, 0.06]), theta_bias=np.zeros(7), frames=frames,
                centers=0.05 * rng.normal(size=(len(frames), 3)), radii=rng.uniform(0.05, 0.12, len(frames)))


def random_marginals(rng, K, d, scale=1.0):
    """Seeded (mu_k, Sigma_k) with SPD Sigma_k, for operator-level parity tests."""
    mu = rng.normal(size=(K, d))
    B = rng.normal(size=(K, d, d))
    Sigma = scale * (B @ np.transpose(B, (0, 2, 1)) / d + 0.2 * np.eye(d))
    return mu, Sigma
