#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of `bench.py --eager --minimal` (tools/pmc_kernels.sh) per kernel into one JSON:
    usage: pmc_summary.py <dir with one sub-dir per pass> <out.json>

Per kernel (template arguments kept, namespaces / argument lists dropped): the mean of every collected counter per
launch and a few derived figures.  HBM traffic per launch = FETCH_SIZE * 2 + WRITE_SIZE (KiB): on gfx950 FETCH_SIZE
reports exactly half of the bytes of 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE was checked
against corr_iter's known 6.55 MB of output per launch (6400 KiB measured).  TCP_TCC_READ_REQ counts 128-byte lines
(TCC_MISS x 128 B reproduces the corrected FETCH_SIZE of corr_iter)."""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1:3]
KEEP = ("lat_conv_kernel", "res_chain16_kernel", "corr_iter", "corr_init", "corrnet_kernel", "lateral_up2_kernel", "conv_tile_kernel", "conv_tile3_kernel", "deconv_tile_kernel", "conv_mfma", "conv_direct_kernel",
        "head_fused", "head_coop", "gru_conv_kernel", "stack2_coop", "stem_kernel", "softmax_max", "view_aggregate", "ref_quarter", "convex_upsample")


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("itermvs::", "")
    return re.sub(r"\(.*$", "", name)


vals = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(s in k for s in KEEP):
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
kernels = {}
for k, cs in sorted(vals.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches_sampled": max(len(v) for v in cs.values()), "grid_threads": grid[k][0], "workgroup": grid[k][1],
         "vgpr_arch": grid[k][2], "lds_bytes": grid[k][3], "counters_mean_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["traffic_bytes_per_launch"] = (m["FETCH_SIZE"] * 2 + m["WRITE_SIZE"]) * 1024
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m:
        e["l1_hit_rate"] = 1.0 - m["TCP_TCC_READ_REQ_sum"] / max(m["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
        e["l1_to_l2_bytes"] = m["TCP_TCC_READ_REQ_sum"] * 128
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0)
    if "SQ_WAVES" in m and "SQ_INSTS_VALU" in m:
        e["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / max(m["SQ_WAVES"], 1.0)
    if "SQ_ACTIVE_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:      # both in quad-cycles summed over waves
        e["wave_time_issuing"] = m["SQ_ACTIVE_INST_ANY"] / max(m["SQ_WAVE_CYCLES"], 1.0)
        e["wave_time_waiting"] = m.get("SQ_WAIT_ANY", 0.0) / max(m["SQ_WAVE_CYCLES"], 1.0)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"] > 0:
        # MFMA-busy cycles summed over the 1024 SIMDs / (busy cycles of the SQs x SIMDs per SQ): see profiles/README.md
        e["mfma_busy_cycles_per_simd"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
    kernels[k] = e
# the workload the counters belong to = the flags of the profiled command (tools/pmc_kernels.sh exports it as PMC_CMD_USED);
# bench.py takes roofline.traffic from this file only when `workload` equals what it runs
import os
cmd = os.environ.get("PMC_CMD_USED", "python bench.py --steps 3 --warmup 2 --repeats 1 --eager --minimal")


def flag(name: str, default: int) -> int:
    m = re.search(rf"--{name}[ =](\d+)", cmd)
    return int(m.group(1)) if m else default


fd = re.search(r"--feature-dtype[ =](\w+)", cmd)
res = {"workload": [flag("views", 5), flag("height", 512), flag("width", 640), flag("batch", 1)],
       "iterations": flag("iters", 4), "feature_dtype": fd.group(1) if fd else "fp32",
       "source": f"rocprofv3 --pmc (one pass per counter set, tools/pmc_kernels.sh) on `{re.sub(r'/[^ ]*/bench.py', 'bench.py', cmd)}`; FETCH_SIZE x2 per MI355X_MICROARCH.md",
       "kernels": kernels}
json.dump(res, open(out, "w"), indent=1)
for k, e in kernels.items():
    print(k, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in e.items() if x != "counters_mean_per_launch"})
