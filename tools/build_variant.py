"""A/B builds of the library: python tools/build_variant.py <name> [DEFINE[=v] ...] -> build/variants/libgvi_hip_<name>.so
(select with GVI_LIB_PATH; build/ is git-ignored and travels to the GPU box with the gpurun snapshot)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianvi_amd import build  # noqa: E402

name, defines = sys.argv[1], tuple(sys.argv[2:])
out_dir = os.path.join(ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
print(build.build_lib(force=True, out=os.path.join(out_dir, f"libgvi_hip_{name}.so"), defines=defines))
