import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
# probe of the two-HIP-runtimes order problem (gaussianvi_amd/_lib.py::_torch_runtime_first):
#   torch_after_probe.py N       -> library first, torch imported afterwards: torch.cuda.init() fails on this pool
#   torch_after_probe.py N pre   -> torch imported (not initialised) first: _lib.load() initialises it before the library
if len(sys.argv) > 2 and sys.argv[2] == "pre":
    import torch  # noqa: F401
from gaussianvi_amd import api, synthetic as syn
ch = syn.make_chain("c2")
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for i in range(n_ctx):
    ctx, ids = api.context_for_chain(ch)
    ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
    ctx.ngd_gradients()
    ctx.close()
print("lib ok", n_ctx, flush=True)
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l][::4], flush=True)
import torch
try:
    torch.cuda.init()
    print("torch ok", torch.cuda.device_count(), flush=True)
except Exception as e:
    print("torch FAILED", e, flush=True)
print(sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l)), flush=True)
