"""CPU stand-in for the C-ABI NGD engine (test infrastructure): the *_local / *_finish split of
include/gvi_hip.h implemented with the oracle's block routines, so gaussianvi_amd.dist.ShardedNGD
(shard ranges, exchange sequence, accept logic) can be exercised over gloo without a GPU."""
import numpy as np
import torch

import gvi_oracle as o
from chains import oracle_psi_batch


class OracleEngine:
    def __init__(self, chain):
        self.T, self.n = chain["T"], chain["n"]
        self.sets = []
        for spec in chain["specs"]:
            fs = o.FactorSet(spec["start"], spec["d"], spec["p"], oracle_psi_batch(spec))
            fs.temperature = np.asarray(spec["temperature"], dtype=np.float64)
            self.sets.append(fs)
        T, n = self.T, self.n
        self.mu = [np.asarray(chain["mu0"], dtype=np.float64).reshape(T, n).copy(), None]
        self.D = [chain["D0"].copy(), None]
        self.U = [chain["U0"].copy(), None]
        self.Sig = [o.inverse_gbp(self.D[0], self.U[0]), None]
        self.cur = 0
        self.ex = {0: torch.zeros(T * n + (2 * T - 1) * n * n, dtype=torch.float64),
                   1: torch.zeros(1, dtype=torch.float64)}

    def exchange_tensor(self, which):
        return self.ex[which]

    def _local_cost(self, i):
        total = 0.0
        for fs in self.sets:
            if len(fs.start) == 0:
                continue
            mk, Sk = o.gather_marginals(self.mu[i], self.Sig[i][0], self.Sig[i][1], fs.start, fs.d)
            total += o.batched_moments(fs.Z, fs.w, mk, Sk, fs.psi_batch, fs.temperature)["cost"].sum()
        self.ex[1][0] = total

    def cost_local(self):
        self._local_cost(self.cur)

    def cost_finish(self):
        i = self.cur
        return float(self.ex[1][0]) + o.logdet_half(o.bt_ldlt_pivots(self.D[i], self.U[i]))

    def gradients_local(self):
        i, T, n = self.cur, self.T, self.n
        parts = []
        for fs in self.sets:
            if len(fs.start) == 0:
                continue
            mk, Sk = o.gather_marginals(self.mu[i], self.Sig[i][0], self.Sig[i][1], fs.start, fs.d)
            r = o.batched_moments(fs.Z, fs.w, mk, Sk, fs.psi_batch, fs.temperature)
            parts.append((fs.start, r["Vdmu"], r["Vddmu"]))
        g, Dv, Uv = o.bt_assemble(T, n, parts)
        self.ex[0][:] = torch.from_numpy(np.concatenate([g.ravel(), Dv.ravel(), Uv.ravel()]))

    def gradients_finish(self):
        i, T, n = self.cur, self.T, self.n
        buf = self.ex[0].numpy()
        g = buf[:T * n].reshape(T, n)
        Dv = buf[T * n:T * n + T * n * n].reshape(T, n, n)
        Uv = buf[T * n + T * n * n:].reshape(T - 1, n, n)
        self.dD, self.dU = Dv - self.D[i], Uv - self.U[i]
        self.dmu = o.bt_solve(Dv, Uv, -g.reshape(-1)).reshape(T, n)

    def trial_local(self, step):
        c, t = self.cur, 1 - self.cur
        self.mu[t] = self.mu[c] + step * self.dmu
        self.D[t] = self.D[c] + step * self.dD
        self.U[t] = self.U[c] + step * self.dU
        with np.errstate(all="ignore"):
            self.Sig[t] = o.inverse_gbp(self.D[t], self.U[t])
            self._local_cost(t)

    def trial_finish(self):
        t = 1 - self.cur
        return float(self.ex[1][0]) + o.logdet_half(o.bt_ldlt_pivots(self.D[t], self.U[t]))

    def accept(self):
        self.cur = 1 - self.cur
