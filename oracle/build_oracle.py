"""Build the oracle's C restatement: oracle/libgvi_oracle.so (gcc -O3 -fopenmp, baseline x86-64 so
the binary built here also runs on the GPU box's host CPU).  Checker / cpu_baseline only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "gvi_oracle.c")
LIB = os.path.join(HERE, "libgvi_oracle.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", LIB, SRC, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed: " + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
