"""Thin numpy-facing wrapper over the C ABI (include/gvi_hip.h).  One method per entry point, same
names and argument meaning; arrays are float64 C-contiguous.  All compute happens in the HIP
library -- nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

PSI_RANGE_1D, PSI_QUAD_PRIOR, PSI_FIXED_PRIOR, PSI_HOST_CALLBACK, PSI_HINGE_SDF_2D = 0, 1, 2, 3, 4
PSI_HINGE_SDF_2D_BODY, PSI_HINGE_SDF_3D, PSI_HINGE_SDF_3D_ARM = 5, 6, 7
RULE_NGD, RULE_PROX_JKO = 0, 1
GVI_F64, GVI_F32 = 0, 1


class GviError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"gvi status {status}: {msg}")
        self.status = status


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def spgh_count(d: int, p: int) -> int:
    lib = _lib.load()
    n = C.c_int64()
    st = lib.gvi_spgh_count(d, p, C.byref(n))
    if st:
        raise GviError(st, lib.gvi_last_error(None).decode())
    return n.value


def spgh_nodes(d: int, p: int):
    """(Z [N,d], w [N], idx [N,d,3] int8) from the product's host generator."""
    lib = _lib.load()
    N = spgh_count(d, p)
    Z = np.empty((N, d)); w = np.empty(N); idx = np.empty((N, d, 3), dtype=np.int8)
    st = lib.gvi_spgh_nodes(d, p, N, _p(Z), _p(w), _p(idx))
    if st:
        raise GviError(st, lib.gvi_last_error(None).decode())
    return Z, w, idx


def _ck_global(lib, st):
    if st:
        raise GviError(st, lib.gvi_last_error(None).decode())


def table_file_list(path: str):
    """[(dim, deg, rows)] of the reference-format (cereal binary) quadrature-table file, in file order."""
    lib = _lib.load()
    n = C.c_int64()
    _ck_global(lib, lib.gvi_table_file_list(path.encode(), 0, C.byref(n), None, None, None))
    dims = np.empty(n.value); degs = np.empty(n.value); rows = np.empty(n.value, dtype=np.int64)
    _ck_global(lib, lib.gvi_table_file_list(path.encode(), n.value, C.byref(n), _p(dims), _p(degs), _p(rows)))
    return [(float(a), float(b), int(c)) for a, b, c in zip(dims, degs, rows)]


def table_file_read(path: str, d: int, p: int):
    lib = _lib.load()
    N = C.c_int64(0)
    _ck_global(lib, lib.gvi_table_file_read(path.encode(), d, p, C.byref(N), None, None))
    Z = np.empty((N.value, d)); w = np.empty(N.value)
    _ck_global(lib, lib.gvi_table_file_read(path.encode(), d, p, C.byref(N), _p(Z), _p(w)))
    return Z, w


def table_file_write(path: str, keys):
    lib = _lib.load()
    dims = np.array([k[0] for k in keys], dtype=np.int32); degs = np.array([k[1] for k in keys], dtype=np.int32)
    _ck_global(lib, lib.gvi_table_file_write(path.encode(), len(keys), _p(dims), _p(degs)))


def dist_unique_id() -> bytes:
    """ncclUniqueId (128 bytes) for gvi_dist_init_rccl; call on rank 0 and distribute."""
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    _ck_global(lib, lib.gvi_dist_unique_id(buf))
    return buf.raw


class Context:
    def __init__(self, device: int = 0, dtype: int = GVI_F64):
        self.lib = _lib.load()
        h = C.c_void_p()
        st = self.lib.gvi_ctx_create(device, dtype, C.byref(h))
        if st:
            raise GviError(st, self.lib.gvi_last_error(None).decode())
        self.h = h
        self.T = self.n = 0
        self.sets = []          # (K, d, p, N)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st:
            raise GviError(st, self.lib.gvi_last_error(self.h).decode())

    # ---- setup ----
    def set_stream(self, stream_ptr):
        self._ck(self.lib.gvi_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self._ck(self.lib.gvi_ctx_sync(self.h))

    def chain_set(self, T, n):
        self._ck(self.lib.gvi_chain_set(self.h, T, n))
        self.T, self.n, self.sets = T, n, []

    def factors_add(self, d, p, start, kind, params=None, temperature=None, table=None):
        """table = (Z [N][d], w [N]): the set takes the caller's quadrature table, nothing is generated
        (gvi_factors_add_table: the shared QuadratureWeightsMap of the reference's factor constructors)."""
        start = np.ascontiguousarray(start, dtype=np.int32)
        K = len(start)
        params = None if params is None else (_f64(params).reshape(K, -1) if K else np.zeros((0, int(np.prod(np.shape(params)[1:])) or 1)))
        temperature = None if temperature is None else _f64(temperature)
        sid = C.c_int()
        if table is None:
            self._ck(self.lib.gvi_factors_add(self.h, K, d, p, _p(start), kind, _p(params),
                                              0 if params is None else params.shape[1], _p(temperature), C.byref(sid)))
        else:
            Z, w = _f64(table[0]), _f64(table[1])
            self._ck(self.lib.gvi_factors_add_table(self.h, K, d, p, _p(start), kind, _p(params),
                                                    0 if params is None else params.shape[1], _p(temperature),
                                                    len(w), _p(Z), _p(w), C.byref(sid)))
        k_, d_, p_, N = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        self._ck(self.lib.gvi_factors_info(self.h, sid.value, C.byref(k_), C.byref(d_), C.byref(p_), C.byref(N)))
        self.sets.append((K, d, p, N.value))
        return sid.value

    def factors_set_table(self, sid, Z, w):
        Z, w = _f64(Z), _f64(w)
        self._ck(self.lib.gvi_factors_set_table(self.h, sid, len(w), _p(Z), _p(w)))
        K, d, p, _ = self.sets[sid]
        self.sets[sid] = (K, d, p, len(w))

    def factors_set_sdf2d(self, sid, origin, cell_size, field):
        """field[r, c]: signed distance at (x = origin[0] + c cell, y = origin[1] + r cell)."""
        f = np.asfortranarray(field, dtype=np.float64)                 # column-major like Eigen's MatrixXd
        self._ck(self.lib.gvi_factors_set_sdf2d(self.h, sid, float(origin[0]), float(origin[1]), float(cell_size),
                                                f.shape[0], f.shape[1], f.ctypes.data_as(C.c_void_p)))

    def factors_set_sdf3d(self, sid, origin, cell_size, field):
        """field[r, c, z]: signed distance at (x = origin[0] + c cell, y = origin[1] + r cell, z = origin[2] + z cell)."""
        f = np.asfortranarray(field, dtype=np.float64)                 # r fastest, then c, then z
        o3 = _f64(np.asarray(origin, dtype=np.float64))
        self._ck(self.lib.gvi_factors_set_sdf3d(self.h, sid, _p(o3), float(cell_size), f.shape[0], f.shape[1], f.shape[2],
                                                f.ctypes.data_as(C.c_void_p)))

    def factors_set_arm(self, sid, arm):
        """arm: dict(a, alpha, d, theta_bias [ndof], frames [ns] int, centers [ns,3], radii [ns])."""
        f = lambda k: _f64(np.asarray(arm[k], dtype=np.float64))
        a_, al, d_, tb, ce, ra = f("a"), f("alpha"), f("d"), f("theta_bias"), f("centers"), f("radii")
        fr = np.ascontiguousarray(arm["frames"], dtype=np.int32)
        self._ck(self.lib.gvi_factors_set_arm(self.h, sid, len(a_), _p(a_), _p(al), _p(d_), _p(tb), len(fr), _p(fr), _p(ce), _p(ra)))

    def factors_set_closed_form(self, sid, on=True):
        self._ck(self.lib.gvi_factors_set_closed_form(self.h, sid, int(on)))

    def factors_set_temperature(self, sid, temperature):
        self._ck(self.lib.gvi_factors_set_temperature(self.h, sid, _p(_f64(temperature))))

    # ---- per-pass factor operators (host buffers) ----
    def moments(self, sid, mu, Sigma):
        K, d, _, _ = self.sets[sid]
        mu, Sigma = _f64(mu), _f64(Sigma)
        Ephi, Vdmu, Vddmu = np.empty(K), np.empty((K, d)), np.empty((K, d, d))
        self._ck(self.lib.gvi_moments(self.h, sid, _p(mu), _p(Sigma), _p(Ephi), _p(Vdmu), _p(Vddmu)))
        return Ephi, Vdmu, Vddmu

    def raw_moments(self, sid, mu, Sigma):
        K, d, _, _ = self.sets[sid]
        mu, Sigma = _f64(mu), _f64(Sigma)
        E0, E1, E2 = np.empty(K), np.empty((K, d)), np.empty((K, d, d))
        self._ck(self.lib.gvi_raw_moments(self.h, sid, _p(mu), _p(Sigma), _p(E0), _p(E1), _p(E2)))
        return E0, E1, E2

    def costs(self, sid, mu, Sigma):
        K = self.sets[sid][0]
        mu, Sigma = _f64(mu), _f64(Sigma)
        cost = np.empty(K)
        self._ck(self.lib.gvi_costs(self.h, sid, _p(mu), _p(Sigma), _p(cost)))
        return cost

    def expand(self, sid, mu, Sigma):
        K, d, _, N = self.sets[sid]
        mu, Sigma = _f64(mu), _f64(Sigma)
        X = np.empty((K, d, N))
        self._ck(self.lib.gvi_expand(self.h, sid, _p(mu), _p(Sigma), _p(X)))
        return X

    def moments_from_psi(self, sid, mu, Sigma, psi):
        K, d, _, _ = self.sets[sid]
        mu, Sigma, psi = _f64(mu), _f64(Sigma), _f64(psi)
        Ephi, Vdmu, Vddmu = np.empty(K), np.empty((K, d)), np.empty((K, d, d))
        self._ck(self.lib.gvi_moments_from_psi(self.h, sid, _p(mu), _p(Sigma), _p(psi), _p(Ephi), _p(Vdmu), _p(Vddmu)))
        return Ephi, Vdmu, Vddmu

    # ---- joint operators ----
    def bt_assemble(self, sids, Vdmus, Vddmus):
        T, n = self.T, self.n
        ids = (C.c_int * len(sids))(*sids)
        Vd = [_f64(v) for v in Vdmus]
        Vdd = [_f64(v) for v in Vddmus]
        pv = (C.c_void_p * len(sids))(*[v.ctypes.data for v in Vd])
        pvv = (C.c_void_p * len(sids))(*[v.ctypes.data for v in Vdd])
        g, D, U = np.empty((T, n)), np.empty((T, n, n)), np.empty((max(T - 1, 0), n, n))
        self._ck(self.lib.gvi_bt_assemble(self.h, len(sids), ids, pv, pvv, _p(g), _p(D), _p(U)))
        return g, D, U

    def bt_solve(self, D, U, rhs):
        D, U, rhs = _f64(D), _f64(U), _f64(rhs)
        x = np.empty(self.T * self.n)
        self._ck(self.lib.gvi_bt_solve(self.h, _p(D), _p(U), _p(rhs), _p(x)))
        return x.reshape(self.T, self.n)

    def bt_logdet(self, D, U):
        D, U = _f64(D), _f64(U)
        out = np.empty(1)
        self._ck(self.lib.gvi_bt_logdet(self.h, _p(D), _p(U), _p(out)))
        return float(out[0])

    def bt_marginals(self, D, U):
        D, U = _f64(D), _f64(U)
        T, n = self.T, self.n
        SD, SU = np.empty((T, n, n)), np.empty((max(T - 1, 0), n, n))
        self._ck(self.lib.gvi_bt_marginals(self.h, _p(D), _p(U), _p(SD), _p(SU)))
        return SD, SU

    def gather_marginals(self, sid, mu, SigD, SigU):
        K, d, _, _ = self.sets[sid]
        mu, SigD, SigU = _f64(mu), _f64(SigD), _f64(SigU)
        mk, Sk = np.empty((K, d)), np.empty((K, d, d))
        self._ck(self.lib.gvi_gather_marginals(self.h, sid, _p(mu), _p(SigD), _p(SigU), _p(mk), _p(Sk)))
        return mk, Sk

    # ---- resident NGD iteration ----
    def ngd_init(self, mu, D, U):
        mu, D, U = _f64(mu), _f64(D), _f64(U)
        self._ck(self.lib.gvi_ngd_init(self.h, _p(mu), _p(D), _p(U)))

    def ngd_cost(self):
        v = C.c_double()
        self._ck(self.lib.gvi_ngd_cost(self.h, C.byref(v)))
        return v.value

    def ngd_factor_costs(self, sid):
        out = np.empty(self.sets[sid][0])
        self._ck(self.lib.gvi_ngd_factor_costs(self.h, sid, _p(out)))
        return out

    def ngd_gradients(self):
        self._ck(self.lib.gvi_ngd_gradients(self.h))

    def ngd_gradients_local(self):
        self._ck(self.lib.gvi_ngd_gradients_local(self.h))

    def ngd_gradients_finish(self):
        self._ck(self.lib.gvi_ngd_gradients_finish(self.h))

    def ngd_cost_local(self):
        self._ck(self.lib.gvi_ngd_cost_local(self.h))

    def ngd_cost_finish(self):
        v = C.c_double()
        self._ck(self.lib.gvi_ngd_cost_finish(self.h, C.byref(v)))
        return v.value

    def ngd_trial(self, step):
        v = C.c_double()
        self._ck(self.lib.gvi_ngd_trial(self.h, step, C.byref(v)))
        return v.value

    def ngd_trial_local(self, step):
        self._ck(self.lib.gvi_ngd_trial_local(self.h, step))

    def ngd_trial_finish(self):
        v = C.c_double()
        self._ck(self.lib.gvi_ngd_trial_finish(self.h, C.byref(v)))
        return v.value

    def ngd_accept(self):
        self._ck(self.lib.gvi_ngd_accept(self.h))

    def ngd_step(self, step_size_base=0.55, max_backtrack=10):
        c0, c1 = C.c_double(), C.c_double()
        ok, ntr = C.c_int(), C.c_int()
        self._ck(self.lib.gvi_ngd_step(self.h, step_size_base, max_backtrack, C.byref(c0), C.byref(ok),
                                       C.byref(c1), C.byref(ntr)))
        return dict(cost_iter=c0.value, accepted=bool(ok.value), new_cost=c1.value, ntrials=ntr.value)

    def ngd_run(self, max_iters, step_size_base=0.55, max_backtrack=10):
        """Up to max_iters iterations in one C call (the loop of GVIGH::optimize); returns a list of the per-iteration
        records gvi_ngd_step reports.  Stops after an iteration whose backtracking was exhausted."""
        c0 = np.zeros(max_iters); c1 = np.zeros(max_iters)
        ok = np.zeros(max_iters, dtype=np.int32); ntr = np.zeros(max_iters, dtype=np.int32)
        done = C.c_int()
        self._ck(self.lib.gvi_ngd_run(self.h, int(max_iters), step_size_base, max_backtrack, _p(c0), _p(ok), _p(c1), _p(ntr),
                                      C.byref(done)))
        return [dict(cost_iter=float(c0[i]), accepted=bool(ok[i]), new_cost=float(c1[i]), ntrials=int(ntr[i]))
                for i in range(done.value)]

    def ngd_set_mode(self, speculate=True, fuse_trial=2):
        """fuse_trial: 0 separate cost pass, 1 fused (full pass at the trial point), 2 adaptive (default)."""
        self._ck(self.lib.gvi_ngd_set_mode(self.h, int(speculate), int(fuse_trial)))

    def ngd_counters(self, reset=False):
        """(full psi passes, cost-only psi passes) launched by the resident iteration since the last reset."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.gvi_ngd_counters(self.h, C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def ngd_exchange(self, which):
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._ck(self.lib.gvi_ngd_exchange(self.h, which, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def ngd_get_state(self):
        T, n = self.T, self.n
        mu, D, U = np.empty((T, n)), np.empty((T, n, n)), np.empty((max(T - 1, 0), n, n))
        SD, SU = np.empty((T, n, n)), np.empty((max(T - 1, 0), n, n))
        self._ck(self.lib.gvi_ngd_get_state(self.h, _p(mu), _p(D), _p(U), _p(SD), _p(SU)))
        return dict(mu=mu, D=D, U=U, SigD=SD, SigU=SU)

    def ngd_get_gradients(self):
        T, n = self.T, self.n
        m1 = max(T - 1, 0)
        out = dict(dmu=np.empty((T, n)), dD=np.empty((T, n, n)), dU=np.empty((m1, n, n)),
                   g=np.empty((T, n)), VD=np.empty((T, n, n)), VU=np.empty((m1, n, n)))
        self._ck(self.lib.gvi_ngd_get_gradients(self.h, *[_p(out[k]) for k in ("dmu", "dD", "dU", "g", "VD", "VU")]))
        return out

    # ---- measurement ----
    def profile_enable(self, on=True):
        self._ck(self.lib.gvi_profile_enable(self.h, int(on)))

    def profile_last(self, sid, what=0):
        ms = C.c_float()
        self._ck(self.lib.gvi_profile_last(self.h, sid, what, C.byref(ms)))
        return ms.value

    def profile_stages(self, on, read=True):
        """Switch the stage timing on / off; with read: {stage: (mean us, brackets)} of the records since the last call."""
        us, cnt = (C.c_float * 3)(), (C.c_int * 3)()
        self._ck(self.lib.gvi_profile_stages(self.h, int(on), us if read else None, cnt if read else None))
        return {name: (us[i], cnt[i]) for i, name in enumerate(("chain", "factors", "assemble"))} if read else None

    def profile_geometry(self, sid):
        v, nch, ch = C.c_int(), C.c_int(), C.c_int64()
        self._ck(self.lib.gvi_profile_geometry(self.h, sid, C.byref(v), C.byref(nch), C.byref(ch)))
        return dict(variant=v.value, nchunk=nch.value, chunk=ch.value)

    # ---- speculative halves (sharded driver) ----
    def ngd_trial_publish(self):
        self._ck(self.lib.gvi_ngd_trial_publish(self.h))

    def ngd_trial_wait(self):
        c = C.c_double()
        self._ck(self.lib.gvi_ngd_trial_wait(self.h, C.byref(c)))
        return c.value

    def ngd_spec_gradients_local(self):
        self._ck(self.lib.gvi_ngd_spec_gradients_local(self.h))

    def ngd_spec_gradients_finish(self):
        self._ck(self.lib.gvi_ngd_spec_gradients_finish(self.h))

    def ngd_accept_spec(self):
        self._ck(self.lib.gvi_ngd_accept_spec(self.h))

    # ---- proximal (JKO) rule ----
    def ngd_set_update_rule(self, rule):
        self._ck(self.lib.gvi_ngd_set_update_rule(self.h, int(rule)))

    def prox_gradients(self, h):
        self._ck(self.lib.gvi_prox_gradients(self.h, float(h)))

    def prox_trial(self, step):
        c = C.c_double()
        self._ck(self.lib.gvi_prox_trial(self.h, float(step), C.byref(c)))
        return c.value

    def prox_step(self, step_size_base=0.55, max_backtrack=10):
        c0, c1, ok, nt = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        self._ck(self.lib.gvi_prox_step(self.h, float(step_size_base), int(max_backtrack), C.byref(c0), C.byref(ok),
                                        C.byref(c1), C.byref(nt)))
        return dict(cost_iter=c0.value, decreased=bool(ok.value), new_cost=c1.value, ntrials=nt.value)

    # ---- sharded factors: exchange inside the library ----
    def dist_init_rccl(self, rank, world, unique_id: bytes):
        assert len(unique_id) == 128
        self._ck(self.lib.gvi_dist_init_rccl(self.h, int(rank), int(world), C.c_char_p(unique_id)))

    def dist_init_callback(self, rank, world, fn):
        """fn(send_ptr, recv_ptr, count, stream_ptr) -> None (raise on failure): all-gather of `count` doubles per rank."""
        def trampoline(user, send, recv, count, stream):
            try:
                fn(send, recv, int(count), stream)
                return 0
            except Exception:                      # a Python exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._allgather_cb = _lib.ALLGATHER_FN(trampoline)      # keep alive as long as the context
        self._ck(self.lib.gvi_dist_init_callback(self.h, int(rank), int(world), C.cast(self._allgather_cb, C.c_void_p), None))

    def dist_info(self):
        r, w, n = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.gvi_dist_info(self.h, C.byref(r), C.byref(w), C.byref(n)))
        return dict(rank=r.value, world=w.value, records_per_rank=n.value)

    def debug_cost_log(self, entries=0, read=False):
        """gvi_debug_cost_log: entries > 0 starts the device-side ring of published costs; read=True returns (ring, sequence)."""
        seq = C.c_double()
        if read:
            buf = np.zeros(self._dbg_entries)
            self._ck(self.lib.gvi_debug_cost_log(self.h, 0, _p(buf), C.byref(seq)))
            return buf, seq.value
        self._dbg_entries = entries if entries > 0 else 0
        self._ck(self.lib.gvi_debug_cost_log(self.h, entries if entries > 0 else -1, None, C.byref(seq)))
        return seq.value

    def set_variant(self, v):
        self._ck(self.lib.gvi_set_variant(self.h, v))

    def set_option(self, name, value):
        self._ck(self.lib.gvi_set_option(self.h, name.encode(), int(value)))


def context_for_chain(chain, device=0, specs=None, tables=None):
    """Context with the chain of gaussianvi_amd.synthetic.make_chain loaded; returns (ctx, set ids).
    tables: {(d, p): (Z, w)} -- the generator's own tables produced elsewhere (e.g. once on rank 0 and broadcast): a set
    whose key is present takes that table through gvi_factors_add_table with "trust_table_degree" set (same routes as the
    generated table, nothing is generated here)."""
    ctx = Context(device)
    ctx.chain_set(chain["T"], chain["n"])
    ids = []
    if tables:
        ctx.set_option("trust_table_degree", 1)
    for spec in (chain["specs"] if specs is None else specs):
        ids.append(ctx.factors_add(spec["d"], spec["p"], spec["start"], spec["kind"], spec["params"],
                                   spec["temperature"], table=(tables or {}).get((spec["d"], spec["p"]))))
        if spec["kind"] in (PSI_HINGE_SDF_2D, PSI_HINGE_SDF_2D_BODY):
            ctx.factors_set_sdf2d(ids[-1], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
        if spec["kind"] in (PSI_HINGE_SDF_3D, PSI_HINGE_SDF_3D_ARM):
            ctx.factors_set_sdf3d(ids[-1], spec["sdf_origin"], spec["sdf_cell"], spec["sdf_field"])
        if spec["kind"] == PSI_HINGE_SDF_3D_ARM:
            ctx.factors_set_arm(ids[-1], spec["arm"])
    return ctx, ids
