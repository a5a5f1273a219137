"""Factor sharding across the GPUs of one node.

One process per GPU.  Factors are independent given the joint (mu, Sigma blocks), so every factor set
is cut into `world` contiguous ranges; a rank runs prep / moments / epilogue / scatter for its range
only.  The path has exactly two exchange steps per pass, both tiny and latency-bound:

  exchange 0  all-reduce of the packed partial [g | D | U] (T n + (2T-1) n^2 doubles; 0.64 MB at C3)
              -- reduce-scatter + all-gather fused, because every rank needs the whole chain for the
              replicated block-tridiagonal solve / log-det / marginals;
  exchange 1  all-reduce of one double: the partial sum of factor costs of a line-search trial.

`ShardedNGD` is backend-agnostic: `engine` is anything with the *_local / *_finish split of the
C ABI (gaussianvi_amd.api.Context wrapped by HipEngine on GPUs; tests drive it with a CPU engine
over gloo).  torch.distributed is plumbing here (RCCL when backend == "nccl").
"""
from __future__ import annotations

import numpy as np


def shard_range(K: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of K items for `rank` (first K % world ranks get one more)."""
    base, rem = divmod(K, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_chain(chain: dict, rank: int, world: int) -> dict:
    """Same chain (T, n, start state), every factor set restricted to this rank's range.
    Start indices stay global."""
    out = dict(chain)
    specs = []
    for spec in chain["specs"]:
        lo, hi = shard_range(len(spec["start"]), rank, world)
        cut = {}
        for k, v in spec.items():
            if isinstance(v, np.ndarray) and v.shape[:1] == (len(spec["start"]),):
                cut[k] = v[lo:hi]
            else:
                cut[k] = v
        cut["range"] = (lo, hi)
        specs.append(cut)
    out["specs"] = [s for s in specs]
    return out


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


class HipEngine:
    """The C-ABI context as a ShardedNGD engine.  Exchange buffers are aliased as torch tensors so
    torch.distributed (RCCL) can all-reduce them in place; the library runs on torch's stream."""

    def __init__(self, ctx, device_index: int):
        import torch
        self.ctx, self.torch = ctx, torch
        self.device = torch.device("cuda", device_index)
        # a dedicated non-default stream shared by the library's kernels and the collectives
        self.stream = torch.cuda.Stream(self.device)
        ctx.set_stream(self.stream.cuda_stream)
        self._ex = {}

    def allreduce(self, which, group=None):
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):      # RCCL orders itself after / before this stream
            dist.all_reduce(self.exchange_tensor(which), op=dist.ReduceOp.SUM, group=group)

    def exchange_tensor(self, which: int):
        ptr, cnt = self.ctx.ngd_exchange(which)          # the gradient buffer can flip (gvi_ngd_step speculation)
        if self._ex.get(which, (None,))[0] != ptr:
            self._ex[which] = (ptr, self.torch.as_tensor(_DevArray(ptr, cnt), device=self.device))
        return self._ex[which][1]

    def gradients_local(self): self.ctx.ngd_gradients_local()
    def gradients_finish(self): self.ctx.ngd_gradients_finish()
    def cost_local(self): self.ctx.ngd_cost_local()
    def cost_finish(self): return self.ctx.ngd_cost_finish()
    def trial_local(self, step): self.ctx.ngd_trial_local(step)
    def trial_finish(self): return self.ctx.ngd_trial_finish()
    def accept(self): self.ctx.ngd_accept()
    # speculative pipeline (see include/gvi_hip.h): present only on engines that can queue work behind the publish
    def trial_publish(self): self.ctx.ngd_trial_publish()
    def trial_wait(self): return self.ctx.ngd_trial_wait()
    def spec_gradients_local(self): self.ctx.ngd_spec_gradients_local()
    def spec_gradients_finish(self): self.ctx.ngd_spec_gradients_finish()
    def accept_spec(self): self.ctx.ngd_accept_spec()


def torch_allgather(device_index: int, group=None):
    """All-gather callback for gvi_dist_init_callback built on torch.distributed: RCCL when the group's backend is nccl
    (stream-ordered on the library's stream), a host round trip through gloo otherwise (tests / rehearsal on one GPU)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device_index)

    def fn(send, recv, count, stream):
        world = dist.get_world_size(group)
        t_send = torch.as_tensor(_DevArray(send, count), device=dev)
        t_recv = torch.as_tensor(_DevArray(recv, count * world), device=dev)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
            if dist.get_backend(group) == "nccl":
                dist.all_gather_into_tensor(t_recv, t_send, group=group)
            else:
                cpu = t_send.cpu()                                   # synchronises with the stream
                parts = [torch.empty_like(cpu) for _ in range(world)]
                dist.all_gather(parts, cpu, group=group)
                t_recv.copy_(torch.cat(parts), non_blocking=False)
    return fn


class InLibraryNGD:
    """The sharded iteration with BOTH exchange steps inside the C library (gvi_dist_init_*): the driver issues one
    gvi_ngd_step per iteration, exactly as in a single process."""

    def __init__(self, ctx):
        self.ctx = ctx

    def reset(self):
        pass

    def step(self, step_size_base=0.55, max_backtrack=10):
        return self.ctx.ngd_step(step_size_base, max_backtrack)


class ShardedNGD:
    """GVIGH::optimize body (gvibase/GVI-GH-impl.h:39-118) over sharded factors."""

    def __init__(self, engine, group=None, world: int = 1):
        self.e, self.group, self.world = engine, group, world
        self.group_forced = False      # bench.py sets it when a size-1 process group exists (plumbing test)
        self._cost = None
        self._grads_ready = False      # gradients of the current proposal already computed (speculatively)
        self.speculate = True

    def reset(self):
        """Forget cached cost / speculative gradients (call after the engine's state was re-initialised)."""
        self._cost = None
        self._grads_ready = False

    def _allreduce(self, which):
        import os
        if self.world > 1 or (self.group_forced and os.environ.get("GVI_FORCE_ALLREDUCE") == "1"):
            if hasattr(self.e, "allreduce"):
                self.e.allreduce(which, self.group)
                return
            import torch.distributed as dist
            dist.all_reduce(self.e.exchange_tensor(which), op=dist.ReduceOp.SUM, group=self.group)

    def cost(self):
        if self._cost is None:
            self.e.cost_local()
            self._allreduce(1)
            self._cost = self.e.cost_finish()
        return self._cost

    def gradients(self):
        self.e.gradients_local()
        self._allreduce(0)
        self.e.gradients_finish()

    def trial(self, step):
        self.e.trial_local(step)
        self._allreduce(1)
        return self.e.trial_finish()

    def step(self, step_size_base=0.55, max_backtrack=10):
        c0 = self.cost()
        if not self._grads_ready:
            self.gradients()
        self._grads_ready = False
        speculate = self.speculate and hasattr(self.e, "spec_gradients_local")
        step, cnt, ok, c1 = step_size_base, 0, False, c0
        while True:
            step *= 0.75                       # gvibase/GVI-GH-impl.h:83
            spec = speculate and cnt == 0
            if spec:
                # the first trial is accepted in the common case: queue the next iteration's gradients at the trial
                # state (other gradient buffer) behind the publish of the trial cost, THEN wait for the cost -- the
                # device never idles while the host reads the scalar and decides (same numbers either way)
                self.e.trial_local(step)
                self._allreduce(1)
                self.e.trial_publish()
                self.e.spec_gradients_local()
                self._allreduce(2)
                self.e.spec_gradients_finish()
                c1 = self.e.trial_wait()
            else:
                c1 = self.trial(step)
            cnt += 1
            if c1 < c0:                        # NaN compares false -> rejected
                if spec:
                    self.e.accept_spec()
                    self._grads_ready = True
                else:
                    self.e.accept()
                self._cost = c1
                ok = True
                break
            if cnt > max_backtrack:
                break
        return dict(cost_iter=c0, accepted=ok, new_cost=c1 if ok else c0, ntrials=cnt)
