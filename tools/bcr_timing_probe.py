"""Phase timing of the segmented-BCR chain kernels (build with GVI_BUILD_DEFINES=GVI_BCR_TIMING):
prints the 100 MHz stamps of block 0 of each pass as microsecond deltas."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussianvi_amd import _lib, api, synthetic as syn

ch = syn.make_chain(sys.argv[1] if len(sys.argv) > 1 else "c3")
ctx, ids = api.context_for_chain(ch)
ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
for _ in range(6):
    ctx.ngd_step(0.9, 10)
lib = _lib.load()
fn = lib.gvi_debug_bcr_stamps
fn.argtypes = [C.c_void_p, C.c_void_p]
fn.restype = C.c_int
out = np.zeros((6 * 64 + 16,), dtype=np.uint64)
assert fn(ctx.h, out.ctypes.data_as(C.c_void_p)) == 0
elim = out[6 * 64:].astype(np.int64)
out = out[:6 * 64].reshape(6, 64)
names = ["factor pass A", "factor pass B (top)", "factor pass C (backward)", "solve pass A", "solve pass B (top)", "solve pass C (backward)"]
for k in range(6):
    st = out[k][out[k] > 0].astype(np.int64)
    if len(st) < 2:
        continue
    d = np.diff(st) * 0.01
    print(f"{names[k]:28s} total {0.01 * (st[-1] - st[0]):6.2f} us  phases:", " ".join(f"{v:.2f}" for v in d))
print("seg_eliminate (wave 0, level 1 of the factor's pass A), shader-clock cycles between stamps [load col | Gauss-Jordan | log-piv | park | sync | element phase | rhs+sync]:",
      np.diff(elim[:7]))
