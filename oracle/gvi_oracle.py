"""CPU oracle for the NGD Gauss-Hermite hot path of hzyu17/GaussianVI.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product path (gaussianvi_amd/, the C-ABI library)
never calls into oracle/ and fails loudly when the HIP extension is missing.

It is a plain numpy restatement of the reference's algorithm for the path, written from the
reference sources (cited per function as file:line, relative to the reference root).  The reference
itself cannot be built here (needs Eigen 3.4 / GSL / MATLAB MCR, none installed), so the oracle is
pinned by the reference's own known answers instead: K1-K9 of SURVEY.md section 4, in particular the
committed 10-iteration NGD trace data/1d/*.csv (tests/golden/ref_1d/) -- see tests/test_oracle_*.py.

Parity pins that the reference's own tests do NOT provide (stated in DESIGN.md too):
  * symmetric square root for d>1 with a non-polynomial psi  -> "parity unpinned" by reference
    fixtures (pinned here only by mathematical uniqueness of the PSD square root);
  * multi-block LDL^T log-det / CG solve / selected inverse  -> pinned only in the 1x1 case (K8);
    cross-checked here against dense numpy linear algebra.

Third-party arithmetic restated from its published algorithm (Eigen 3.4.0, un-vendored dependency
of the reference, CMakeLists.txt:46): SelfAdjointEigenSolver::operatorSqrt (V sqrt(L) V^T),
SimplicialLDLT<Lower,NaturalOrdering> (scalar LDL^T without pivoting), ConjugateGradient with the
default DiagonalPreconditioner (tol = eps, maxit = 2 n), dense inverse().
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------------
# a1: sparse-grid generator  (quadrature/GH/SparseGH/nwspgr.m)
# --------------------------------------------------------------------------------------------
_GQN = None


def gqn(level: int):
    """Positive-half 1-D Gauss-Hermite rule of `level` nodes (nwspgr.m:272-350, data table)."""
    global _GQN
    if _GQN is None:
        with open(os.path.join(_HERE, "gqn_table.json")) as f:
            raw = json.load(f)
        _GQN = {int(k): (np.array([float(s) for s in v["n"]]), np.array([float(s) for s in v["w"]]))
                for k, v in raw.items()}
    return _GQN[level]


def spgr_get_seq(d: int, norm: int) -> np.ndarray:
    """All rows of N^d (entries >= 1) with row sum == norm, in the reference's order
    (nwspgr.m:147-169, SpGrGetSeq)."""
    seq = np.zeros(d, dtype=np.int64)
    a = norm - d
    seq[0] = a
    fs = [seq.copy()]
    c = 1  # 1-based cursor like the MATLAB code
    while seq[d - 1] < a:
        if c == d:
            for i in range(c - 1, 0, -1):
                c = i
                if seq[i - 1] != 0:
                    break
        seq[c - 1] -= 1
        c += 1
        seq[c - 1] = a - seq[: c - 1].sum()
        if c < d:
            seq[c:] = 0
        fs.append(seq.copy())
    return np.array(fs, dtype=np.int64) + 1


def _kron_prod(levels, wdtype=np.float64):
    """Tensor product of positive-half rules, last dimension fastest (nwspgr.m:183-190).
    Returns nodes, weights and the (level, node-index) pair of every coordinate."""
    n0, w0 = gqn(int(levels[0]))
    nodes = n0[:, None]
    weights = w0.astype(wdtype)
    idx = np.stack([np.full(len(n0), levels[0]), np.arange(len(n0))], axis=1)[:, None, :]
    for lv in levels[1:]:
        nn, ww = gqn(int(lv))
        r_old, r_new = nodes.shape[0], len(nn)
        nodes = np.hstack([np.repeat(nodes, r_new, axis=0), np.tile(nn, r_old)[:, None]])
        new_idx = np.stack([np.full(r_new, lv), np.arange(r_new)], axis=1)
        idx = np.concatenate([np.repeat(idx, r_new, axis=0),
                              np.tile(new_idx, (r_old, 1))[:, None, :]], axis=1)
        weights = np.kron(weights, ww.astype(wdtype))
    return nodes, weights, idx


def _sortrows(nodes):
    """MATLAB sortrows: ascending lexicographic, stable."""
    return np.lexsort(nodes.T[::-1])


_REF_MAX_DEG = (25, 25, 19, 13, 11, 9, 8, 7, 7, 7, 6, 6, 6)


def is_reference_table_key(dim: int, k: int) -> bool:
    """Keys of the reference's quadrature table file (quadrature/saveSparseGHWeightMap.h:16-23)."""
    return (dim <= 13 and k <= _REF_MAX_DEG[dim - 1]) or (14 <= dim <= 20 and k <= 5)


def nwspgr(dim: int, k: int, return_index: bool = False, extended=None):
    """nwspgr('GQN', dim, k, sym=1) (nwspgr.m:32-134).

    `extended`: merge and normalise the weights in long double and round once, instead of the reference's plain double
    sums.  Default: only for keys OUTSIDE the reference's table file -- there is no reference table to be in parity with,
    and the double sums degrade there (at (24,7) a quadratic psi is integrated to 1e-5 only; see csrc/spgh.cpp).

    Returns Z (N x dim, rows sorted lexicographically), w (N, sums to 1, some negative) and, when
    asked, idx (N x dim x 3 int8: level, node index inside the positive half-rule, sign) -- the
    'sigma-point index' that the drop-in contract keeps bit-exact.  The value 0 is shared by every
    odd level; it is canonicalised to (1, 0, 0).
    """
    if extended is None:
        extended = not is_reference_table_key(dim, k)
    wdtype = np.longdouble if extended else np.float64
    minq, maxq = max(0, k - dim), k - 1
    nodes = np.zeros((0, dim))
    weights = np.zeros(0, dtype=wdtype)
    idx = np.zeros((0, dim, 2), dtype=np.int64)
    for q in range(minq, maxq + 1):
        bq = (-1) ** (maxq - q) * math.comb(dim - 1, dim + q - k)
        for midx in spgr_get_seq(dim, dim + q):
            nn, ww, ii = _kron_prod(midx, wdtype)
            nodes = np.vstack([nodes, nn])
            weights = np.concatenate([weights, bq * ww])
            idx = np.concatenate([idx, ii], axis=0)
        order = _sortrows(nodes)
        nodes, weights, idx = nodes[order], weights[order], idx[order]
        # merge exactly-equal rows, summing weights in sorted order into the first (:88-103)
        if len(nodes) > 1:
            same = np.all(nodes[1:] == nodes[:-1], axis=1)
            keep = np.concatenate([[True], ~same])
            group = np.cumsum(keep) - 1
            merged = np.zeros(keep.sum(), dtype=wdtype)
            for j in range(len(weights)):  # sequential sum order of the reference
                merged[group[j]] += weights[j]
            nodes, weights, idx = nodes[keep], merged, idx[keep]
    # reflect to the other orthants (:108-126); m = n1D{1} = 0
    sign = np.where(nodes != 0.0, 1, 0).astype(np.int64)
    for j in range(dim):
        nz = nodes[:, j] != 0.0
        if nz.any():
            refl = nodes[nz].copy()
            refl[:, j] = 2 * 0.0 - refl[:, j]
            s2 = sign[nz].copy()
            s2[:, j] = -s2[:, j]
            nodes = np.vstack([nodes, refl])
            weights = np.concatenate([weights, weights[nz]])
            idx = np.concatenate([idx, idx[nz]], axis=0)
            sign = np.vstack([sign, s2])
    order = _sortrows(nodes)
    nodes, weights, idx, sign = nodes[order], weights[order], idx[order], sign[order]
    weights = (weights / weights.sum()).astype(np.float64)
    if not return_index:
        return nodes, weights
    full = np.concatenate([idx, sign[:, :, None]], axis=2)
    zero = nodes == 0.0
    full[zero] = (1, 0, 0)
    return nodes, weights, full.astype(np.int8)


def spgh_count(dim: int, k: int) -> int:
    return nwspgr(dim, k)[0].shape[0]


# --------------------------------------------------------------------------------------------
# a3-a5: SparseGaussHermite  (quadrature/SparseGaussHermite.h)
# --------------------------------------------------------------------------------------------
def sym_sqrt(P: np.ndarray) -> np.ndarray:
    """Eigen SelfAdjointEigenSolver::operatorSqrt: V diag(sqrt(lambda)) V^T
    (quadrature/SparseGaussHermite.h:232-233).  Negative eigenvalue -> NaN like the reference."""
    lam, V = np.linalg.eigh(np.asarray(P, dtype=np.float64))
    with np.errstate(invalid="ignore"):
        return (V * np.sqrt(lam)) @ V.T


class SparseGaussHermite:
    """quadrature/SparseGaussHermite.h:138-166,197-243 with an in-memory (dim,deg) table."""
    _table = {}

    def __init__(self, deg: int, dim: int, mean, P):
        self._deg, self._dim = deg, dim
        key = (dim, deg)
        if key not in SparseGaussHermite._table:
            SparseGaussHermite._table[key] = nwspgr(dim, deg)
        self._zeromeanpts, self._Weights = SparseGaussHermite._table[key]
        self._mean = np.asarray(mean, dtype=np.float64).reshape(dim)
        self._P = np.asarray(P, dtype=np.float64).reshape(dim, dim)
        self.update_sigmapoints()

    def update_P(self, P):
        self._P = np.asarray(P, dtype=np.float64).reshape(self._dim, self._dim)

    def update_mean(self, mean):
        self._mean = np.asarray(mean, dtype=np.float64).reshape(self._dim)

    def update_sigmapoints(self):  # :231-243
        self._sqrtP = sym_sqrt(self._P)
        self._sigmapts = self._zeromeanpts @ self._sqrtP.T + self._mean[None, :]

    def Integrate(self, function):  # :197-221 (serial order; the reference's order is OMP-dependent)
        res = np.zeros_like(np.atleast_2d(np.asarray(function(self._mean), dtype=np.float64)))
        for i in range(self._sigmapts.shape[0]):
            res = res + np.atleast_2d(function(self._sigmapts[i])) * self._Weights[i]
        return res

    def weights(self):
        return self._Weights

    def sigmapts(self):
        return self._sigmapts


# --------------------------------------------------------------------------------------------
# a7: psi kinds actually shipped
# --------------------------------------------------------------------------------------------
def psi_range_1d(x, y=40.0 / 20.0 - 0.8, mu_p=20.0, fb=40.0, sig_r_sq=0.09, sig_p_sq=9.0):
    """src/1d_example.cpp:25-35 (y = f b / mu_p - 0.8 there; tests/test_GH.cpp:20-33 uses +0.05)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)[0]
    return (x - mu_p) * (x - mu_p) / sig_p_sq / 2 + (y - fb / x) * (y - fb / x) / sig_r_sq / 2


def minimum_acc_phi_qinv(Qc: np.ndarray, delta_t: float):
    """MinimumAccGP (gp/minimum_acc_prior.h:39-80,103-116): Phi = [[I, dt I],[0, I]],
    Q^-1 = [[12/dt^3, -6/dt^2],[-6/dt^2, 4/dt]] (x) Qc^-1."""
    Qc = np.atleast_2d(np.asarray(Qc, dtype=np.float64))
    nd = Qc.shape[0]
    I = np.eye(nd)
    Phi = np.block([[I, delta_t * I], [np.zeros((nd, nd)), I]])
    iQc = np.linalg.inv(Qc)
    Qinv = np.block([[12 * iQc / delta_t ** 3, -6 * iQc / delta_t ** 2],
                     [-6 * iQc / delta_t ** 2, 4 * iQc / delta_t]])
    return Phi, Qinv


def psi_quad_prior(x, Phi, Qinv):
    """cost_linear_gp -> MinimumAccGP::cost / LTV_GP::cost: 1/2 (Phi th1 - th2)^T Q^-1 (Phi th1 - th2)
    (gp/cost_functions.h:36-39; gp/minimum_acc_prior.h:103-106; gp/LTV_prior.h:223-226)."""
    n = Phi.shape[0]
    r = Phi @ x[:n] - x[n:2 * n]
    return float(r @ Qinv @ r) / 2


def psi_fixed_prior(x, mu0, Kinv):
    """FixedPriorGP::fixed_factor_cost (gp/fixed_prior.h:28-30): (x-mu0)^T K^-1 (x-mu0), no 1/2."""
    e = x - mu0
    return float(e @ Kinv @ e)


_NWSPGR_CACHE = {}


def nwspgr_cached(dim, k):
    """nwspgr memoised per process ((24,5) takes tens of seconds in numpy)."""
    if (dim, k) not in _NWSPGR_CACHE:
        _NWSPGR_CACHE[(dim, k)] = nwspgr(dim, k)
    return _NWSPGR_CACHE[(dim, k)]


def cereal_table_bytes(entries):
    """Byte image of a QuadratureWeightsMap written by cereal::BinaryOutputArchive
    (quadrature/saveSparseGHWeightMap.h:44-50; helpers/SerializeEigenMaps.h:195-224): u64 count, then per
    entry the key tuple (f64 dim, f64 deg), MatrixXd (i32 rows, i32 cols, elements row by row) and
    VectorXd (i32 len, elements).  entries: [(dim, deg, Z[N,d], w[N])] in the order to write."""
    import struct
    out = [struct.pack("<Q", len(entries))]
    for dim, deg, Z, w in entries:
        Z = np.ascontiguousarray(Z, dtype="<f8"); w = np.ascontiguousarray(w, dtype="<f8")
        out += [struct.pack("<dd", float(dim), float(deg)), struct.pack("<ii", Z.shape[0], Z.shape[1]), Z.tobytes(),
                struct.pack("<i", w.shape[0]), w.tobytes()]
    return b"".join(out)


def cereal_table_parse(buf):
    """Inverse of cereal_table_bytes -> {(dim, deg): (Z, w)} (the load side, quadrature/SparseGaussHermite.h:80-117)."""
    import struct
    (n,), off, res = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(n):
        dim, deg = struct.unpack_from("<dd", buf, off); off += 16
        r, c = struct.unpack_from("<ii", buf, off); off += 8
        Z = np.frombuffer(buf, dtype="<f8", count=r * c, offset=off).reshape(r, c); off += 8 * r * c
        (ln,) = struct.unpack_from("<i", buf, off); off += 4
        w = np.frombuffer(buf, dtype="<f8", count=ln, offset=off); off += 8 * ln
        res[(dim, deg)] = (Z, w)
    assert off == len(buf)
    return res


def planar_sdf_lookup(px, py, origin, cell, field):
    """PlanarSDF::convertPoint2toCell + signed_distance (helpers/CudaOperation.h:61-103): clamp the query
    to the grid, bilinear interpolation; field[r, c] (Eigen column-major data_array_[r + c rows], :130).
    The reference reads one row/column past the edge with weight 0 there; the index is clamped instead."""
    rows, cols = field.shape
    xin = np.clip(px, origin[0], origin[0] + (cols - 1.0) * cell)
    yin = np.clip(py, origin[1], origin[1] + (rows - 1.0) * cell)
    col, row = (xin - origin[0]) / cell, (yin - origin[1]) / cell
    lr, lc = np.floor(row), np.floor(col)
    hr, hc = lr + 1.0, lc + 1.0
    lri = np.nan_to_num(lr, nan=0.0).astype(int)                      # NaN queries (rejected trials) stay NaN via the weights
    lci = np.nan_to_num(lc, nan=0.0).astype(int)
    hri, hci = np.minimum(lri + 1, rows - 1), np.minimum(lci + 1, cols - 1)
    return ((hr - row) * (hc - col) * field[lri, lci] + (row - lr) * (hc - col) * field[hri, lci] +
            (hr - row) * (col - lc) * field[lri, hci] + (row - lr) * (col - lc) * field[hri, hci])


def psi_batch_hinge_sdf2d(params, origin, cell, field):
    """cost_obstacle_planar of the planar point robot (helpers/CudaOperation.h:491-508, one ball, slope 1):
    sigma * max(0, eps + r - sdf(x0, x1))^2; params [K,3] = (sigma, eps, r)."""
    def f(X, sel=slice(None)):
        P = params[sel]
        sd = planar_sdf_lookup(X[:, :, 0], X[:, :, 1], origin, cell, field)
        thr = (P[:, 1] + P[:, 2])[:, None]
        err = np.where(sd > thr, 0.0, thr - sd)
        return err * err * P[:, 0][:, None]
    return f


def sdf3d_lookup(px, py, pz, origin, cell, field):
    """SignedDistanceField::convertPoint3toCell + signed_distance (helpers/CudaOperation.h:176-226): clamp, trilinear
    interpolation; field[r, c, z] = data_array_[r + c rows + z rows cols] (:304-306), x -> col, y -> row."""
    rows, cols, nz = field.shape
    xin = np.clip(px, origin[0], origin[0] + (cols - 1.0) * cell)
    yin = np.clip(py, origin[1], origin[1] + (rows - 1.0) * cell)
    zin = np.clip(pz, origin[2], origin[2] + (nz - 1.0) * cell)
    col, row, zz = (xin - origin[0]) / cell, (yin - origin[1]) / cell, (zin - origin[2]) / cell
    lr, lc, lz = np.floor(row), np.floor(col), np.floor(zz)
    hr, hc, hz = lr + 1.0, lc + 1.0, lz + 1.0
    lri, lci, lzi = (np.nan_to_num(v, nan=0.0).astype(int) for v in (lr, lc, lz))
    hri, hci, hzi = np.minimum(lri + 1, rows - 1), np.minimum(lci + 1, cols - 1), np.minimum(lzi + 1, nz - 1)
    g = field
    return ((hr - row) * (hc - col) * (hz - zz) * g[lri, lci, lzi] + (row - lr) * (hc - col) * (hz - zz) * g[hri, lci, lzi] +
            (hr - row) * (col - lc) * (hz - zz) * g[lri, hci, lzi] + (row - lr) * (col - lc) * (hz - zz) * g[hri, hci, lzi] +
            (hr - row) * (hc - col) * (zz - lz) * g[lri, lci, hzi] + (row - lr) * (hc - col) * (zz - lz) * g[hri, lci, hzi] +
            (hr - row) * (col - lc) * (zz - lz) * g[lri, hci, hzi] + (row - lr) * (col - lc) * (zz - lz) * g[hri, hci, hzi])


def psi_batch_hinge_sdf2d_body(params, origin, cell, field):
    """CudaOperation_Quad::cost_obstacle_planar + vec_balls (helpers/CudaOperation.h:565-606): pose (x, z, phi);
    params [K,6] = (sigma, eps, r, slope, n_balls, L)."""
    def f(X, sel=slice(None)):
        P = params[sel]
        px, pz, phi = X[:, :, 0], X[:, :, 1], X[:, :, 2]
        sig, eps, r, slope, nb, L = (P[:, j][:, None] for j in range(6))
        lx = px - (L - r * 1.5) * np.cos(phi) / 2.0
        lz = pz - (L - r * 1.5) * np.sin(phi) / 2.0
        cost = np.zeros_like(px)
        for i in range(int(P[0, 4])):
            sd = planar_sdf_lookup(lx + L * np.cos(phi) / nb * i, lz + L * np.sin(phi) / nb * i, origin, cell, field)
            err = np.where(sd > eps + r, 0.0, (eps + r - sd) * slope)
            cost = cost + err * err * sig
        return cost
    return f


def psi_batch_hinge_sdf3d(params, origin, cell, field):
    """CudaOperation_3dpR::cost_obstacle_planar (helpers/CudaOperation.h:650-668): one ball at x[0:3], slope 1."""
    def f(X, sel=slice(None)):
        P = params[sel]
        sd = sdf3d_lookup(X[:, :, 0], X[:, :, 1], X[:, :, 2], origin, cell, field)
        thr = (P[:, 1] + P[:, 2])[:, None]
        err = np.where(sd > thr, 0.0, thr - sd)
        return err * err * P[:, 0][:, None]
    return f


def arm_sphere_centers(theta, a, alpha, dd, theta_bias, frames, centers):
    """ForwardKinematics::compute_transformed_sphere_centers (helpers/CudaOperation.h:362-386): for sphere s the
    chain T = prod_{i <= frames[s]} DH(i, theta_i + bias_i), position = T[:3,3] + T[:3,:3] @ centers[s].
    dh_matrix (:388-395) evaluates every trig term with cosf / sinf -- SINGLE precision -- and stores the result in a
    double matrix; restated as float32 trig on the float32-rounded angle.  theta [..., >= ndof] -> [..., ns, 3]."""
    nd = len(a)
    shp = theta.shape[:-1]
    T = np.broadcast_to(np.eye(4), shp + (4, 4)).copy()
    cum = []
    # cosf / sinf: float argument, float result.  No two libm's agree on the last float bit (the reference's is CUDA's),
    # so the representative used here and on the device is the correctly rounded one: double trig of the float
    # argument, rounded to float.
    f32 = lambda v: v.astype(np.float64)
    cosf = lambda v: np.cos(f32(v)).astype(np.float32)
    sinf = lambda v: np.sin(f32(v)).astype(np.float32)
    ca = cosf(np.asarray(alpha, dtype=np.float32))
    sa = sinf(np.asarray(alpha, dtype=np.float32))
    for i in range(nd):
        th = (theta[..., i] + theta_bias[i]).astype(np.float32)
        c, s = cosf(th), sinf(th)                              # float32
        M = np.zeros(shp + (4, 4))
        # float * float products stay float32 (-sinf(theta)*cosf(alpha), ...); a(i)*cosf(theta) is double * float
        M[..., 0, 0], M[..., 0, 1], M[..., 0, 2], M[..., 0, 3] = c, -s * ca[i], s * sa[i], a[i] * c.astype(np.float64)
        M[..., 1, 0], M[..., 1, 1], M[..., 1, 2], M[..., 1, 3] = s, c * ca[i], -c * sa[i], a[i] * s.astype(np.float64)
        M[..., 2, 1], M[..., 2, 2], M[..., 2, 3] = sa[i], ca[i], dd[i]
        M[..., 3, 3] = 1.0
        T = T @ M
        cum.append(T)
    out = np.zeros(shp + (len(frames), 3))
    for s_, fr in enumerate(frames):
        Tf = cum[int(fr)]
        out[..., s_, :] = Tf[..., :3, 3] + np.einsum("...ij,j->...i", Tf[..., :3, :3], centers[s_])
    return out


def psi_batch_hinge_sdf3d_arm(params, arm, origin, cell, field):
    """CudaOperation_3dArm::cost_obstacle (helpers/CudaOperation.h:752-771): n_balls = theta.size() (the factor
    dimension), err_i = hinge(eps + radius_i - sdf(p_i)), cost = sigma sum err_i^2.  params [K,2] = (sigma, eps);
    arm = dict(a, alpha, d, theta_bias, frames, centers [ns,3], radii [ns])."""
    def f(X, sel=slice(None)):
        P = params[sel]
        nb = X.shape[-1]
        pts = arm_sphere_centers(X, arm["a"], arm["alpha"], arm["d"], arm["theta_bias"], arm["frames"][:nb], arm["centers"][:nb])
        cost = np.zeros(X.shape[:2])
        for i in range(nb):
            sd = sdf3d_lookup(pts[:, :, i, 0], pts[:, :, i, 1], pts[:, :, i, 2], origin, cell, field)
            thr = P[:, 1][:, None] + arm["radii"][i]
            err = np.where(sd > thr, 0.0, thr - sd)
            cost = cost + err * err * P[:, 0][:, None]
        return cost
    return f


def ltv_phi_q(A_list, B_list, delta_t: float):
    """(Phi, Q) of LTV_GP (gp/LTV_prior.h:123-197): Phi' = A(t) Phi, Q' = A Q + Q A^T + B B^T over
    [0, dt] with A, B piece-wise constant on 4 sub-intervals.  The reference integrates with GSL
    rkf45 (tol 1e-12, absent here); this is the exact solution of the same ODE (product of matrix
    exponentials, Van Loan block form)."""
    from scipy.linalg import expm
    n = A_list[0].shape[0]
    Phi, Q = np.eye(n), np.zeros((n, n))
    h = delta_t / 4
    for A, B in zip(A_list[:4], B_list[:4]):
        M = np.zeros((2 * n, 2 * n))
        M[:n, :n] = -A
        M[:n, n:] = B @ B.T
        M[n:, n:] = A.T
        E = expm(M * h)
        Ad = E[n:, n:].T
        Qd = Ad @ E[:n, n:]
        Phi = Ad @ Phi
        Q = Ad @ Q @ Ad.T + (Qd + Qd.T) / 2
    return Phi, Q


# --------------------------------------------------------------------------------------------
# a6, a8, a9, a11: factor operator  (gvibase/GVIFactorizedBase*.h, ngd/NGDFactorizedBaseGH.h)
# --------------------------------------------------------------------------------------------
class NGDFactorizedBaseGH:
    """ngd/NGDFactorizedBaseGH.h:37-129 (== ngd/NGDFactorizedLinearGH.h:27-116 for the GH part)."""

    def __init__(self, dimension, state_dim, gh_degree, function, num_states, start_index,
                 temperature=1.0, high_temperature=10.0):
        self._dim, self._state_dim = dimension, state_dim
        self._num_states, self._start_index = num_states, start_index
        self._temperature, self._high_temperature = temperature, high_temperature
        self._joint_size = state_dim * num_states
        self._mu = np.zeros(dimension)
        self._covariance = np.eye(dimension)
        self._precision = np.eye(dimension)
        self._function = function
        self._Vdmu = np.zeros(dimension)
        self._Vddmu = np.zeros((dimension, dimension))
        self._func_phi = lambda x: np.array([[self._function(x)]])
        self._func_Vmu = lambda x: ((x - self._mu) * self._function(x))[:, None]
        self._func_Vmumu = lambda x: np.outer(x - self._mu, x - self._mu) * self._function(x)
        self._gh = SparseGaussHermite(gh_degree, dimension, self._mu, self._covariance)

    # gvibase/GVIFactorizedBase.h:104-122
    def _sl(self):
        s = self._state_dim * self._start_index
        return slice(s, s + self._dim)

    def extract_mu_from_joint(self, joint_mean):
        return np.asarray(joint_mean)[self._sl()].copy()

    def extract_cov_from_joint(self, joint_cov):
        return np.asarray(joint_cov)[self._sl(), self._sl()].copy()

    def update_mu_from_joint(self, joint_mean):
        self._mu = self.extract_mu_from_joint(joint_mean)

    def update_precision_from_joint(self, joint_cov):
        self._covariance = self.extract_cov_from_joint(joint_cov)
        self._precision = np.linalg.inv(self._covariance)

    def factor_switch_to_high_temperature(self):
        self._temperature = self._high_temperature

    def temperature(self):
        return self._temperature

    def updateGH(self, x, P):  # gvibase/GVIFactorizedBaseGH.h:44-49
        self._gh.update_P(P)
        self._gh.update_mean(x)
        self._gh.update_sigmapoints()

    def calculate_partial_V(self):  # ngd/NGDFactorizedBaseGH.h:53-74
        self.updateGH(self._mu, self._covariance)
        Vdmu = self._gh.Integrate(self._func_Vmu)[:, 0]
        Vdmu = self._precision @ Vdmu
        self._Vdmu = Vdmu / self.temperature()
        E_phi = self._gh.Integrate(self._func_phi)[0, 0]
        E_xxphi = self._gh.Integrate(self._func_Vmumu)
        full = self._precision @ E_xxphi @ self._precision - self._precision * E_phi
        up = np.triu(full)
        self._Vddmu = (up + np.triu(full, 1).T) / self.temperature()
        self._E_phi, self._E_xmuphi, self._E_xxphi = E_phi, Vdmu, E_xxphi

    def local2joint_dmu_insertion(self):  # :91-96
        res = np.zeros(self._joint_size)
        res[self._sl()] = self._Vdmu
        return res

    def local2joint_dprecision_insertion(self):  # :98-106
        res = np.zeros((self._joint_size, self._joint_size))
        res[self._sl(), self._sl()] = self._Vddmu
        return res

    def fact_cost_value(self, joint_mean, joint_cov):  # :122-129
        mean_k = self.extract_mu_from_joint(joint_mean)
        cov_k = self.extract_cov_from_joint(joint_cov)
        self.updateGH(mean_k, cov_k)
        return self._gh.Integrate(self._func_phi)[0, 0] / self.temperature()


def linear_factor_closed_form(mu, Sigma, Lam_k, Lambda, Kinv, Psi_mu_t, constant, temperature):
    """a19 analytic oracle: NGDFactorizedLinear::calculate_partial_V / fact_cost_value
    (ngd/NGDFactorizedLinear.h:93-129).  Lam_k = Sigma^-1 (marginal precision)."""
    M = Lambda.T @ Kinv @ Lambda
    r = Lambda @ mu - Psi_mu_t
    Vdmu = 2 * Lambda.T @ Kinv @ r * constant / temperature
    tmp = (Sigma * np.sum(Sigma * M) + Sigma @ M.T @ Sigma + Sigma @ M @ Sigma)
    Vddmu = (Lam_k @ tmp @ Lam_k - Lam_k * np.trace(M @ Sigma)) * constant / temperature
    E_phi = (np.trace(M @ Sigma) + r @ Kinv @ r) * constant
    return E_phi / temperature, Vdmu, Vddmu


# --------------------------------------------------------------------------------------------
# Vectorised (batched) restatement of a4-a9 for all factors of one homogeneous set.  Same maths,
# one expand + one psi evaluation per point (the reference evaluates psi three times); used as
# the checker at sizes where the per-point python loops above are too slow.
# --------------------------------------------------------------------------------------------
def batched_moments(Z, w, mu, Sigma, psi_batch, temperature, block=None):
    """mu [K,d], Sigma [K,d,d], psi_batch(X[K,N,d]) -> [K,N].  Returns dict with E_phi, E_xmuphi,
    E_xxphi (raw GH integrals) and Vdmu, Vddmu (ngd/NGDFactorizedBaseGH.h:53-74).
    Large batches are processed in blocks of factors to bound memory (psi_batch must then accept a
    `sel` keyword selecting the factors of the block)."""
    K = mu.shape[0]
    if block is None:
        block = max(1, int(2e8 // max(1, Z.shape[0] * Z.shape[1] * 8)))
    if K > block:
        T = np.broadcast_to(np.asarray(temperature, dtype=np.float64).reshape(-1), (K,))
        parts = []
        for a in range(0, K, block):
            sel = slice(a, min(K, a + block))
            parts.append(batched_moments(Z, w, mu[sel], Sigma[sel], lambda X, s=sel: psi_batch(X, sel=s), T[sel], block=K))
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    lam, V = np.linalg.eigh(Sigma)
    with np.errstate(invalid="ignore", divide="ignore"):
        S = np.einsum("kij,kj,klj->kil", V, np.sqrt(lam), V)
    Lam = np.linalg.inv(Sigma)
    Y = np.einsum("na,kba->knb", Z, S)          # (Z S^T)
    X = Y + mu[:, None, :]
    psi = psi_batch(X)
    c = psi * w[None, :]
    E_phi = c.sum(axis=1)
    E_xmuphi = np.einsum("kn,kna->ka", c, Y)
    E_xxphi = np.einsum("kna,knb->kab", c[:, :, None] * Y, Y)
    T = np.asarray(temperature, dtype=np.float64).reshape(-1)
    Vdmu = np.einsum("kab,kb->ka", Lam, E_xmuphi) / T[:, None]
    Vddmu = (Lam @ E_xxphi @ Lam - Lam * E_phi[:, None, None]) / T[:, None, None]
    Vddmu = (Vddmu + np.transpose(Vddmu, (0, 2, 1))) / 2
    return dict(E_phi=E_phi, E_xmuphi=E_xmuphi, E_xxphi=E_xxphi, Vdmu=Vdmu, Vddmu=Vddmu,
                cost=E_phi / T, S=S, Lam=Lam)


def psi_batch_quad(A, b, sgn=None, half=True):
    """psi(x) = (1/2) sum_r sgn_r (A x + b)_r^2 for a batch: A [K,m,d], b [K,m]."""
    def f(X, sel=slice(None)):
        U = np.einsum("kmd,knd->knm", A[sel], X) + b[sel][:, None, :]
        s = np.ones(A.shape[:2])[sel] if sgn is None else sgn[sel]
        val = np.einsum("knm,km->kn", U * U, s)
        return val / 2 if half else val
    return f


def psi_batch_quad_prior(Phi, Qinv):
    """Batched psi_quad_prior: Phi [K,n,n], Qinv [K,n,n]."""
    n = Phi.shape[1]

    def f(X, sel=slice(None)):
        R = np.einsum("kij,knj->kni", Phi[sel], X[:, :, :n]) - X[:, :, n:2 * n]
        return np.einsum("kni,kni->kn", np.einsum("kni,kij->knj", R, Qinv[sel]), R) / 2
    return f


def psi_batch_fixed_prior(mu0, Kinv):
    def f(X, sel=slice(None)):
        E = X - mu0[sel][:, None, :]
        return np.einsum("kni,kni->kn", np.einsum("kni,kij->knj", E, Kinv[sel]), E)
    return f


def psi_batch_range_1d(y=1.2, mu_p=20.0, fb=40.0, sig_r_sq=0.09, sig_p_sq=9.0):
    def f(X, sel=slice(None)):
        x = X[:, :, 0]
        return (x - mu_p) ** 2 / sig_p_sq / 2 + (y - fb / x) ** 2 / sig_r_sq / 2
    return f


# --------------------------------------------------------------------------------------------
# block-tridiagonal helpers (joint level).  D [T,n,n] diagonal blocks, U [T-1,n,n] blocks (i,i+1).
# --------------------------------------------------------------------------------------------
def bt_to_dense(D, U):
    T, n = D.shape[0], D.shape[1]
    A = np.zeros((T * n, T * n))
    for i in range(T):
        A[i * n:(i + 1) * n, i * n:(i + 1) * n] = D[i]
        if i + 1 < T:
            A[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = U[i]
            A[(i + 1) * n:(i + 2) * n, i * n:(i + 1) * n] = U[i].T
    return A


def dense_to_bt(A, n):
    T = A.shape[0] // n
    D = np.stack([A[i * n:(i + 1) * n, i * n:(i + 1) * n] for i in range(T)])
    U = (np.stack([A[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] for i in range(T - 1)])
         if T > 1 else np.zeros((0, n, n)))
    return D, U


def bt_assemble(T, n, sets):
    """a10 + a12 first half (ngd/NGDFactorizedBaseGH.h:91-106, ngd/NGD-GH-impl.h:39-55): scatter-add
    Vdmu_k / Vddmu_k of every factor into the joint vector and the block-tridiagonal matrix.
    sets: list of (start[K], Vdmu[K,d], Vddmu[K,d,d]) with d in {n, 2n}.  Factors are added in
    index order, set after set (the reference's order depends on OpenMP scheduling)."""
    g = np.zeros((T, n))
    D = np.zeros((T, n, n))
    U = np.zeros((max(T - 1, 0), n, n))
    for start, Vdmu, Vddmu in sets:
        d = Vdmu.shape[1]
        for k in range(len(start)):
            s = int(start[k])
            g[s] += Vdmu[k, :n]
            D[s] += Vddmu[k, :n, :n]
            if d == 2 * n:
                g[s + 1] += Vdmu[k, n:]
                D[s + 1] += Vddmu[k, n:, n:]
                U[s] += Vddmu[k, :n, n:]
    return g, D, U


def ldlt_pivots_dense(A):
    """Eigen SimplicialLDLT<Lower, NaturalOrdering>: scalar LDL^T without pivoting; returns the
    pivot vector D (helpers/CommonDefinitions.h:23-27; gvibase/GVI-GH-impl.h:192-196)."""
    A = np.array(A, dtype=np.float64)
    N = A.shape[0]
    L = np.eye(N)
    Dv = np.zeros(N)
    for j in range(N):
        Dv[j] = A[j, j] - (L[j, :j] ** 2) @ Dv[:j]
        if j + 1 < N:
            with np.errstate(divide="ignore", invalid="ignore"):
                L[j + 1:, j] = (A[j + 1:, j] - (L[j + 1:, :j] * L[j, :j]) @ Dv[:j]) / Dv[j]
    return L, Dv


def bt_ldlt_pivots(D, U):
    """Same pivots as ldlt_pivots_dense(bt_to_dense(D,U)) computed block-wise: the scalar pivots of
    the natural-order LDL^T are those of each Schur-complemented diagonal block."""
    T, n = D.shape[0], D.shape[1]
    piv = np.zeros((T, n))
    Sc = D[0].copy()
    for i in range(T):
        L, dv = ldlt_pivots_dense(Sc)
        piv[i] = dv
        if i + 1 < T:
            with np.errstate(divide="ignore", invalid="ignore"):
                X = np.linalg.solve(L, U[i])          # L^-1 U
                Sc = D[i + 1] - X.T @ (X / dv[:, None])
    return piv.reshape(-1)


def logdet_half(pivots):
    """sum(log D)/2 with the reference's NaN semantics (negative pivot -> NaN)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(np.sum(np.log(pivots)) / 2)


def cg_eigen(A, rhs):
    """Eigen 3.4 ConjugateGradient<SpMat, Upper> with DiagonalPreconditioner, x0 = 0, tol = eps,
    maxIterations = 2 n (ngd/NGD-GH-impl.h:59-60)."""
    A = np.asarray(A, dtype=np.float64)
    A = np.triu(A) + np.triu(A, 1).T           # selfadjointView<Upper>
    rhs = np.asarray(rhs, dtype=np.float64)
    n = A.shape[0]
    tol = np.finfo(np.float64).eps
    x = np.zeros(n)
    diag = np.diag(A)
    invdiag = np.where(diag != 0, 1.0 / np.where(diag != 0, diag, 1.0), 1.0)
    residual = rhs - A @ x
    rhs2 = rhs @ rhs
    if rhs2 == 0:
        return x
    threshold = max(tol * tol * rhs2, np.finfo(np.float64).tiny)
    res2 = residual @ residual
    if res2 < threshold:
        return x
    p = invdiag * residual
    abs_new = residual @ p
    it = 0
    while it < 2 * n:
        tmp = A @ p
        alpha = abs_new / (p @ tmp)
        x = x + alpha * p
        residual = residual - alpha * tmp
        res2 = residual @ residual
        if res2 < threshold:
            break
        z = invdiag * residual
        abs_old = abs_new
        abs_new = residual @ z
        p = z + (abs_new / abs_old) * p
        it += 1
    return x


def bt_solve(D, U, rhs):
    """Direct solve of the symmetric block-tridiagonal system (block forward elimination /
    back substitution); what the CG of the reference converges to."""
    T, n = D.shape[0], D.shape[1]
    rhs = np.asarray(rhs, dtype=np.float64).reshape(T, n)
    Sc = [None] * T
    y = np.zeros((T, n))
    W = [None] * max(T - 1, 0)
    Sc[0] = D[0].copy()
    y[0] = rhs[0]
    for i in range(T - 1):
        W[i] = np.linalg.solve(Sc[i], U[i])                 # S_i^-1 U_i
        Sc[i + 1] = D[i + 1] - U[i].T @ W[i]
        y[i + 1] = rhs[i + 1] - U[i].T @ np.linalg.solve(Sc[i], y[i])
    x = np.zeros((T, n))
    x[T - 1] = np.linalg.solve(Sc[T - 1], y[T - 1])
    for i in range(T - 2, -1, -1):
        x[i] = np.linalg.solve(Sc[i], y[i]) - W[i] @ x[i + 1]
    return x.reshape(-1)


def inv_sparse_takahashi(A, n):
    """a16: EigenWrapper::inv_sparse (helpers/EigenWrapper.h:282-331) on the lower-triangular
    pattern of overlapping 2n x 2n blocks (gvibase/GVI-GH.h:214-230), restated on dense storage:
    Z_jk = [j==k]/D_j - sum_{l>k, L_lk != 0 in pattern} Z_{max(l,j),min(l,j)} L_lk, from the last
    pattern entry backwards; mirrored to the upper triangle."""
    N = A.shape[0]
    T = N // n
    L, Dv = ldlt_pivots_dense(A)
    pat = np.zeros((N, N), dtype=bool)
    if T == 1:
        pat[:, :] = True
    else:
        for i in range(T - 1):
            pat[i * n:(i + 2) * n, i * n:(i + 2) * n] = True
    pat = np.tril(pat)
    rows, cols = np.nonzero(pat.T)          # column-major listing: iterate k (col) then j (row)
    rows, cols = cols, rows
    Zi = np.zeros((N, N))
    for index in range(len(rows) - 1, -1, -1):
        j, k = rows[index], cols[index]
        cur = 1.0 / Dv[j] if j == k else 0.0
        for l in range(k + 1, N):
            if not pat[l, k]:
                continue
            cur -= (Zi[l, j] if l > j else Zi[j, l]) * L[l, k]
        Zi[j, k] = cur
    return Zi + np.tril(Zi, -1).T


def inverse_gbp(D, U):
    """a17: GVIGH::inverse_GBP (gvibase/GVI-GH-GBP-impl.h:246-342; GVI-GH-GBP.h:181-183).
    Returns the tridiagonal blocks (SigD [T,n,n], SigU [T-1,n,n]) of the covariance."""
    T, n = D.shape[0], D.shape[1]
    F = np.zeros((T, n, n))
    B = np.zeros((T, n, n))
    for i in range(T - 1):
        # variable message (+ incoming) then factor message = Schur complement onto the target
        F[i + 1] = -U[i].T @ np.linalg.inv(D[i] + F[i]) @ U[i]
        j = T - 1 - i
        B[j - 1] = -U[j - 1] @ np.linalg.inv(D[j] + B[j]) @ U[j - 1].T
    SigD = np.zeros((T, n, n))
    SigU = np.zeros((max(T - 1, 0), n, n))
    if T == 1:
        SigD[0] = np.linalg.inv(F[0] + B[0] + D[0])
    for i in range(T - 1):
        J = np.block([[D[i] + F[i], U[i]], [U[i].T, D[i + 1] + B[i + 1]]])
        Vj = np.linalg.inv(J)
        SigD[i] = Vj[:n, :n]
        SigD[i + 1] = Vj[n:, n:]
        SigU[i] = Vj[:n, n:]
    return SigD, SigU


def gather_marginals(mu, SigD, SigU, start, d):
    """a11: mu_k = mu[n s : n s + d], Sigma_k = Sigma[block] (gvibase/GVIFactorizedBase.h:104-122)."""
    T, n = SigD.shape[0], SigD.shape[1]
    mu = np.asarray(mu).reshape(T, n)
    K = len(start)
    mk = np.zeros((K, d))
    Sk = np.zeros((K, d, d))
    for k in range(K):
        s = int(start[k])
        mk[k, :n] = mu[s]
        Sk[k, :n, :n] = SigD[s]
        if d == 2 * n:
            mk[k, n:] = mu[s + 1]
            Sk[k, n:, n:] = SigD[s + 1]
            Sk[k, :n, n:] = SigU[s]
            Sk[k, n:, :n] = SigU[s].T
    return mk, Sk


# --------------------------------------------------------------------------------------------
# a12-a18: joint optimiser  (gvibase/GVI-GH*.h, ngd/NGD-GH*.h), dense-joint restatement for small
# problems (what the reference does with SpMat), `variant` selects the covariance routine.
# --------------------------------------------------------------------------------------------
class NGDGH:
    def __init__(self, vec_factors, dim_state, num_states, niterations=5, temperature=1.0,
                 high_temperature=100.0, variant="gbp", solver="cg"):
        self._vec_factors = vec_factors
        self._dim_state, self._num_states = dim_state, num_states
        self._dim = dim_state * num_states
        self._niters = niterations
        self._niters_lowtemp, self._niters_backtrack, self._stop_err = 10, 10, 1e-5  # GVI-GH.h:51-53
        self._temperature, self._high_temperature = temperature, high_temperature
        self._mu = np.zeros(self._dim)
        self._precision = np.zeros((self._dim, self._dim))
        self._covariance = np.zeros((self._dim, self._dim))
        self._step_size_base = 0.55  # GVI-GH.h:93
        self._variant, self._solver = variant, solver
        self.record = dict(mean=[], cov=[], precision=[], cost=[], factor_costs=[])

    # setters (gvibase/GVI-GH.h:168-248)
    def set_step_size_base(self, v): self._step_size_base = v
    def set_niter_low_temperature(self, v): self._niters_lowtemp = v
    def set_max_iter_backtrack(self, v): self._niters_backtrack = v

    def set_mu(self, mean):
        self._mu = np.asarray(mean, dtype=np.float64).copy()
        for f in self._vec_factors:
            f.update_mu_from_joint(self._mu)

    def inverse(self, P):
        n = self._dim_state
        if self._variant == "takahashi":
            return inv_sparse_takahashi(P, n)
        D, U = dense_to_bt(P, n)
        SD, SU = inverse_gbp(D, U)
        return bt_to_dense(SD, SU)

    def set_precision(self, P):  # GVI-GH-impl.h:127-141 / GVI-GH-GBP-impl.h:170-183
        self._precision = np.asarray(P, dtype=np.float64).copy()
        self._covariance = self.inverse(self._precision)
        for f in self._vec_factors:
            f.update_precision_from_joint(self._covariance)

    def set_initial_values(self, mean, precision):
        self.set_mu(mean)
        self.set_precision(precision)

    def switch_to_high_temperature(self):
        for f in self._vec_factors:
            f.factor_switch_to_high_temperature()
        self._temperature = self._high_temperature

    def cost_value(self, mean=None, precision=None):  # GVI-GH-impl.h:176-197
        mean = self._mu if mean is None else mean
        precision = self._precision if precision is None else precision
        cov = self.inverse(precision)
        value = 0.0
        for f in self._vec_factors:
            value += f.fact_cost_value(mean, cov)
        _, Dv = ldlt_pivots_dense(precision)
        return value + logdet_half(Dv)

    def factor_cost_vector(self):  # GVI-GH-impl.h:147-170
        cov = self.inverse(self._precision)
        return np.array([f.fact_cost_value(self._mu, cov) for f in self._vec_factors])

    def compute_gradients(self):  # ngd/NGD-GH-impl.h:21-63
        Vdmu = np.zeros(self._dim)
        Vddmu = np.zeros((self._dim, self._dim))
        for f in self._vec_factors:
            f.calculate_partial_V()
            Vdmu += f.local2joint_dmu_insertion()
            Vddmu += f.local2joint_dprecision_insertion()
        self._Vdmu, self._Vddmu = Vdmu, Vddmu
        dprecision = Vddmu - self._precision
        if self._solver == "cg":
            dmu = cg_eigen(Vddmu, -Vdmu)
        else:
            dmu = np.linalg.solve(Vddmu, -Vdmu)
        return dmu, dprecision

    def onestep_linesearch(self, step, dmu, dprecision):  # ngd/NGD-GH-impl.h:130-148
        new_mu = self._mu + step * dmu
        new_prec = self._precision + step * dprecision
        return self.cost_value(new_mu, new_prec), new_mu, new_prec

    def optimize(self):  # gvibase/GVI-GH-impl.h:33-124
        is_lowtemp, converged = True, False
        for i_iter in range(self._niters):
            if converged:
                break
            if i_iter == self._niters_lowtemp and is_lowtemp:
                self.switch_to_high_temperature()
                is_lowtemp = False
            cost_iter = self.cost_value()
            fact_costs = self.factor_cost_vector()
            self.record["mean"].append(self._mu.copy())
            self.record["cov"].append(self._covariance.copy())
            self.record["precision"].append(self._precision.copy())
            self.record["cost"].append(cost_iter)
            self.record["factor_costs"].append(fact_costs)
            dmu, dprecision = self.compute_gradients()
            cnt = 0
            step = self._step_size_base
            while True:
                step = step * 0.75
                new_cost, new_mu, new_prec = self.onestep_linesearch(step, dmu, dprecision)
                if new_cost < cost_iter:
                    self.set_mu(new_mu)
                    self.set_precision(new_prec)
                    break
                cnt += 1
                if cnt > self._niters_backtrack:
                    if is_lowtemp:
                        self.switch_to_high_temperature()
                        is_lowtemp = False
                    else:
                        converged = True
                    break
        return self

    def cost_map(self, x_start, x_end, y_start, y_end, nmesh):  # gvibase/GVI-GH.h:385-404
        Zm = np.zeros((nmesh, nmesh))
        rx, ry = (x_end - x_start) / nmesh, (y_end - y_start) / nmesh
        for i in range(nmesh):
            for j in range(nmesh):
                Zm[j, i] = self.cost_value(np.array([x_start + i * rx]),
                                           np.array([[y_start + j * ry]]))
        return Zm


# --------------------------------------------------------------------------------------------
# Block-level (large chain) restatement of one NGD iteration over homogeneous factor sets; the
# checker for the C-ABI at C2/C3 sizes.  Same update law as NGDGH above.
# --------------------------------------------------------------------------------------------
class FactorSet:
    """K factors of one kind: start [K], d, psi_batch(X)->[K,N], temperature [K], GH degree p."""

    def __init__(self, start, d, p, psi_batch, temperature=1.0):
        self.start = np.asarray(start, dtype=np.int64)
        self.d, self.p, self.psi_batch = d, p, psi_batch
        self.temperature = np.full(len(self.start), float(temperature))
        self.Z, self.w = nwspgr_cached(d, p)
        # optional (mk, Sk, temperature) -> (E_phi, Vdmu, Vddmu): the C restatement (oracle/c, reference-shaped three-pass
        # form, OpenMP over factors) instead of the numpy one -- same numbers (tests/test_oracle_c.py), seconds instead
        # of a minute per pass at BASELINE sizes
        self.fast_moments = None

    def moments(self, mk, Sk):
        if self.fast_moments is None:
            return batched_moments(self.Z, self.w, mk, Sk, self.psi_batch, self.temperature)
        E, Vd, Vdd = self.fast_moments(mk, Sk, self.temperature)
        return dict(E_phi=E, cost=E / self.temperature, Vdmu=Vd, Vddmu=Vdd)


class ChainNGD:
    def __init__(self, T, n, sets, mu0, D0, U0, step_size_base=0.55, max_backtrack=10):
        self.T, self.n, self.sets = T, n, sets
        self.mu = np.asarray(mu0, dtype=np.float64).reshape(T, n).copy()
        self.D, self.U = D0.copy(), U0.copy()
        self.step_size_base, self.max_backtrack = step_size_base, max_backtrack
        self.SigD, self.SigU = inverse_gbp(self.D, self.U)

    def cost_value(self, mu, D, U, SigD=None, SigU=None):
        if SigD is None:
            SigD, SigU = inverse_gbp(D, U)
        value = 0.0
        for fs in self.sets:
            mk, Sk = gather_marginals(mu, SigD, SigU, fs.start, fs.d)
            r = fs.moments(mk, Sk)
            value += r["cost"].sum()
        return value + logdet_half(bt_ldlt_pivots(D, U))

    def gradients(self):
        parts = []
        for fs in self.sets:
            mk, Sk = gather_marginals(self.mu, self.SigD, self.SigU, fs.start, fs.d)
            r = fs.moments(mk, Sk)
            parts.append((fs.start, r["Vdmu"], r["Vddmu"]))
        g, Dv, Uv = bt_assemble(self.T, self.n, parts)
        dmu = bt_solve(Dv, Uv, -g.reshape(-1)).reshape(self.T, self.n)
        return dmu, Dv - self.D, Uv - self.U, (g, Dv, Uv)

    def step(self):
        """One NGD iteration (gvibase/GVI-GH-impl.h:39-118).  Returns (accepted, cost, n_trials)."""
        cost_iter = self.cost_value(self.mu, self.D, self.U, self.SigD, self.SigU)
        dmu, dD, dU, _ = self.gradients()
        step, cnt = self.step_size_base, 0
        while True:
            step *= 0.75
            mu, D, U = self.mu + step * dmu, self.D + step * dD, self.U + step * dU
            SigD, SigU = inverse_gbp(D, U)
            new_cost = self.cost_value(mu, D, U, SigD, SigU)
            if new_cost < cost_iter:
                self.mu, self.D, self.U, self.SigD, self.SigU = mu, D, U, SigD, SigU
                return True, new_cost, cnt + 1
            cnt += 1
            if cnt > self.max_backtrack:
                return False, cost_iter, cnt


# --------------------------------------------------------------------------------------------
# VIMPResults (helpers/DataRecorder.h:25-225): what the nine CSV files hold.  `iterations` is a list of
# (mean [T n], joint_cov [Tn, Tn], joint_precision [Tn, Tn], cost, factor_costs [K]) recorded by update_data (:88-112);
# compress3d flattens a matrix with Eigen's reshaped(), i.e. COLUMN-major (helpers/EigenWrapper.h:228-232), one column
# per iteration; columns of iterations that never ran stay zero (the arrays are preallocated with niters columns).
# --------------------------------------------------------------------------------------------
def vimp_results_files(iterations, niters, dim_state, nstates):
    n, T = dim_state, nstates
    Tn = n * T
    nf = len(iterations[0][4]) if iterations else 0
    out = dict(mean=np.zeros((Tn, niters)), cov=np.zeros((n * n * T, niters)), precision=np.zeros((n * n * T, niters)),
               joint_cov=np.zeros((Tn * Tn, niters)), joint_precision=np.zeros((Tn * Tn, niters)), cost=np.zeros(niters),
               factor_costs=np.zeros((nf, niters)))

    def joint2marginals(J):                               # :120-127
        return np.concatenate([J[t * n:(t + 1) * n, t * n:(t + 1) * n].reshape(-1, order="F") for t in range(T)])

    for it, (mean, jc, jp, cost, fc) in enumerate(iterations[:niters]):
        out["mean"][:, it] = np.asarray(mean).reshape(-1)
        out["cov"][:, it] = joint2marginals(jc)
        out["precision"][:, it] = joint2marginals(jp)
        out["joint_cov"][:, it] = jc.reshape(-1, order="F")
        out["joint_precision"][:, it] = jp.reshape(-1, order="F")
        out["cost"][it] = cost
        out["factor_costs"][:, it] = fc
    last = min(len(iterations), niters) - 1               # save_data :203-218: decomp3d of the last recorded column
    out["zk_sdf"] = out["mean"][:, last].reshape((n, T), order="F")
    out["Sk_sdf"] = out["cov"][:, last].reshape((n * n, T), order="F")
    return out


# --------------------------------------------------------------------------------------------
# Proximal (JKO / Bures-Wasserstein) update -- SURVEY 8(f)4.  Same quadrature moments as the NGD path; the
# factor-level map and the joint loop follow proxgd/ProxGVIFactorizedBaseGH.h and proxgd/ProxGVI-GH-impl.h.
# --------------------------------------------------------------------------------------------
def bw_jko(mu, Sigma, Lam, E_phi, E_xmuphi, E_xxphi, h):
    """compute_BW_grads + BW_JKO for one factor (proxgd/ProxGVIFactorizedBaseGH.h:64-113, 152-160):
    b = Lam E[(x-mu) psi];  S = Lam E[(x-mu)(x-mu)^T psi] Lam - Lam E[psi];  M = I - h S;
    Sig_half = M Sigma M^T;  Sigma_new = Sig_half/2 + h I + sqrtm(Sig_half (Sig_half + 4 h I))/2;
    Vdmu = -b (= (mu_new - mu)/h), Vddmu = (Sigma_new^-1 - Lam)/h.  Sig_half commutes with Sig_half + 4hI, so the
    product is symmetric PSD and the reference's Schur-based sqrtm (:213-243) is the symmetric square root."""
    d = mu.shape[0]
    b = Lam @ E_xmuphi
    S = Lam @ E_xxphi @ Lam - Lam * E_phi
    M = np.eye(d) - h * S
    Sh = M @ Sigma @ M.T
    lam, W = np.linalg.eigh(0.5 * (Sh + Sh.T))
    with np.errstate(invalid="ignore"):
        root = np.sqrt(lam * (lam + 4.0 * h))
    new = 0.5 * lam + h + 0.5 * root
    Lam_new = (W / new) @ W.T
    return -b, (Lam_new - Lam) / h


class ChainProx:
    """ProxGVIGH::optimize (proxgd/ProxGVI-GH-impl.h:121-202) on a chain.  Differences from ChainNGD restated from
    the source: gradients are the scattered sums of the factor-level JKO increments computed ONCE at step = base
    (:149-151), dmu is used directly (no solve, :30-31), the step of trial B is base**B (:160), the raw integrals are
    not divided by the temperature (ProxGVIFactorizedBaseGH.h:152-160, 245-252), and after max_backtrack failed
    trials the last trial is accepted anyway (:177-184)."""

    def __init__(self, T, n, sets, mu0, D0, U0, step_size_base=0.55, max_backtrack=10):
        self.T, self.n, self.sets = T, n, sets
        self.mu = np.asarray(mu0, dtype=np.float64).reshape(T, n).copy()
        self.D, self.U = D0.copy(), U0.copy()
        self.step_size_base, self.max_backtrack = step_size_base, max_backtrack
        self.SigD, self.SigU = inverse_gbp(self.D, self.U)

    def factor_costs(self, mu, SigD, SigU):
        out = []
        for fs in self.sets:
            mk, Sk = gather_marginals(mu, SigD, SigU, fs.start, fs.d)
            out.append(batched_moments(fs.Z, fs.w, mk, Sk, fs.psi_batch, 1.0)["E_phi"])
        return out

    def cost_value(self, mu, D, U, SigD=None, SigU=None):
        if SigD is None:
            SigD, SigU = inverse_gbp(D, U)
        return sum(c.sum() for c in self.factor_costs(mu, SigD, SigU)) + logdet_half(bt_ldlt_pivots(D, U))

    def gradients(self, h):
        parts = []
        for fs in self.sets:
            mk, Sk = gather_marginals(self.mu, self.SigD, self.SigU, fs.start, fs.d)
            r = batched_moments(fs.Z, fs.w, mk, Sk, fs.psi_batch, 1.0)
            K = mk.shape[0]
            Vd = np.zeros((K, fs.d)); Vdd = np.zeros((K, fs.d, fs.d))
            for k in range(K):
                Vd[k], Vdd[k] = bw_jko(mk[k], Sk[k], np.linalg.inv(Sk[k]), r["E_phi"][k], r["E_xmuphi"][k], r["E_xxphi"][k], h)
            parts.append((fs.start, Vd, Vdd))
        g, Dv, Uv = bt_assemble(self.T, self.n, parts)
        return g, Dv, Uv

    def step(self):
        """One iteration.  Returns (decreased, cost after the update, n_trials)."""
        cost_iter = self.cost_value(self.mu, self.D, self.U, self.SigD, self.SigU)
        dmu, dD, dU = self.gradients(self.step_size_base)
        B, cnt = 1, 0
        while True:
            step = self.step_size_base ** B
            mu, D, U = self.mu + step * dmu, self.D + step * dD, self.U + step * dU
            SigD, SigU = inverse_gbp(D, U)
            new_cost = self.cost_value(mu, D, U, SigD, SigU)
            ok = bool(new_cost < cost_iter)
            if not ok:
                B += 1
                cnt += 1
            if ok or cnt > self.max_backtrack:
                self.mu, self.D, self.U, self.SigD, self.SigU = mu, D, U, SigD, SigU     # accepted even when not decreased
                return ok, new_cost, cnt + (1 if ok else 0)

