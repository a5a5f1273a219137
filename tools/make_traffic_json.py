#!/usr/bin/env python3
"""profiles/rNN_traffic.json (what bench.py reports as roofline.traffic) from a PMC summary (tools/summarize_pmc.py):
   make_traffic_json.py <summary.json> <kernel-name substring> <algorithmic bytes per launch> <kernel us (rocprof)> <out.json>"""
import json
import sys

summary, needle, alg_bytes, kernel_us, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
d = json.load(open(summary))
name = [k for k in d if needle in k]
assert len(name) == 1, name
c = {k: v["mean"] for k, v in d[name[0]].items()}
cycles = c["SQ_BUSY_CYCLES"] / 32.0
res = {
    "kernel": name[0],
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "hbm_bytes_per_launch": 1024.0 * (c["FETCH_SIZE"] + c["WRITE_SIZE"]),
    "hbm_bytes_per_launch_if_reads_doubled": 1024.0 * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]),
    "TCC_HIT_sum": c["TCC_HIT_sum"], "TCC_MISS_sum": c["TCC_MISS_sum"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "algorithmic_bytes_per_launch": alg_bytes,
    "sq": {k: v for k, v in c.items() if k.startswith("SQ_")},
    "derived": {
        "kernel_cycles": cycles,
        "clock_GHz_implied": cycles / (kernel_us * 1e3),
        "valu_pipe_busy": c["SQ_INSTS_VALU"] / 1024.0 * 4.0 / cycles,
        "wait_fraction_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "lds_wait_fraction_of_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"],
        "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
        "lds_instructions_per_wave": c["SQ_INSTS_LDS"] / c["SQ_WAVES"],
    },
    "note": "rocprofv3 --pmc passes of `bench.py --steps 6 --warmup 2 --no-cpu-baseline` (tools/gpu_pmc.sh: FETCH_SIZE, WRITE_SIZE, "
            "SQ_*, TCC_* in separate runs with --kernel-trace only), means over the launches of the run.  FETCH / WRITE in KB as "
            "rocprofv3 reports them; loads are 8 B per lane, so the guide's x2 FETCH_SIZE correction (calibrated on 16 B/lane streams) "
            "is given as a second figure.  SQ_BUSY_CYCLES / 32 shader engines = kernel duration in cycles; VALU pipe busy = "
            "SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / that (every wave64 VALU instruction, fp64 or integer, holds its SIMD for 4 cycles).",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["derived"]), res["hbm_bytes_per_launch"])
