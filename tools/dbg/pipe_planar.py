import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import numpy as np
from chains import make_chain
from gaussianvi_amd import api

name = sys.argv[1] if len(sys.argv) > 1 else "planar"
ch = make_chain(name)
for asm in (1, 0):
    ctx, ids = api.context_for_chain(ch)
    ctx.set_option("assemble_on_load", asm)
    for base in (1.9,):
        ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
        ref = [ctx.ngd_step(base, 10) for _ in range(12)]
        for pipeline in (1, 0):
            ctx.set_option("pipeline", pipeline)
            ctx.ngd_init(ch["mu0"], ch["D0"], ch["U0"])
            got = ctx.ngd_run(5, base, 10) + ctx.ngd_run(1, base, 10) + ctx.ngd_run(6, base, 10)
            print("asm", asm, "pipeline", pipeline, "equal", got == ref)
            if got != ref:
                for i, (a, b) in enumerate(zip(got, ref)):
                    print(i, "got", a, "| ref", b)
    ctx.close()
