#!/usr/bin/env python3
"""Collect (box probe, depth-maps/s) pairs from the bench lines of this round's GPU sessions (gpurun_out/s*_*.json) into a
markdown table + per-probe correlation with the step time:   python tools/box_table.py > profiles/r06_box_probe.md"""
import glob
import json
import os
import statistics

rows = []
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "s*_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        continue
    if not isinstance(d, dict) or "box" not in d or not d.get("box"):
        continue
    box = dict(d["box"])
    if d.get("roofline"):
        box["corr_iter_us"] = d["roofline"]["avg_launch_ms"] * 1e3        # the unchanged kernel, event-bracketed in the timed region
    rows.append((os.path.basename(f), d["value"], d["ms_per_step"], box, d["config"].get("cpu_affinity", {}).get("gpu")))
keys = ["corr_iter_us", "mfma_f32_tflops", "copy_GBps", "sclk_MHz", "graph_node_us", "l2_latency_ns", "hbm_latency_ns", "sclk_idle_MHz", "sclk_workload_MHz"]
print("# Box probes of round 6 (`bench.py` `box`) beside the depth-maps/s of the same run\n")
print("Every row is one `bench.py` run on a fresh MI355X box of the pool (session, file under `gpurun_out/` at the time).  `frozen` rows ran a")
print("frozen copy of the source (tools/freeze_reference.sh), the others the source of their moment -- compare values only within a group.\n")
print("| run | depth-maps/s | ms/step | " + " | ".join(keys) + " | GPU |")
print("|---|---|---|" + "---|" * (len(keys) + 1))
for name, v, ms, box, gpu in rows:
    print(f"| {name} | {v:.1f} | {ms:.4f} | " + " | ".join(f"{box[k]:.2f}" if k in box and box[k] else "" for k in keys) + f" | {gpu} |")
def _sess(name):          # session number of "s27_bench_1.json"
    digits = "".join(ch for ch in name.split("_")[0][1:] if ch.isdigit())
    return int(digits) if digits else -1


for grp, sel in (("frozen source of the round's first half (sessions s1-s9)", [r for r in rows if "frozen" in r[0] and _sess(r[0]) < 12]),
                 ("frozen source eb59622 (sessions s12-s33: the A/B partner of every later change)", [r for r in rows if "frozen" in r[0] and _sess(r[0]) >= 12]),
                 ("final source (sessions s31, s32: XCD-banded conv tiles, lat_conv, res_chain16, bf16x3 head)",
                  [r for r in rows if "frozen" not in r[0] and _sess(r[0]) in (31, 32)]),
                 ("round-6 source after the tail fusions (sessions s3, s6-s9; kernels differ by < 1 %)",
                  [r for r in rows if "frozen" not in r[0] and _sess(r[0]) in (3, 6, 7, 8, 9)])):
    if len(sel) < 3:
        continue
    print(f"\n## {grp}: {len(sel)} runs, depth-maps/s {min(r[1] for r in sel):.1f} .. {max(r[1] for r in sel):.1f} "
          f"(spread {100 * (max(r[1] for r in sel) / min(r[1] for r in sel) - 1):.1f} %)\n")
    print("| probe | min .. max | spread % | Pearson r with depth-maps/s | spread of value x (median / probe) % |")
    print("|---|---|---|---|---|")
    vals = [r[1] for r in sel]
    for k in keys:
        xs = [r[3].get(k) for r in sel]
        if any(x is None or not x for x in xs) or (k == "sclk_workload_MHz" and max(xs) > 3000):
            continue
        mx, my = statistics.mean(xs), statistics.mean(vals)
        sx = sum((x - mx) ** 2 for x in xs) ** 0.5
        sy = sum((y - my) ** 2 for y in vals) ** 0.5
        r = sum((x - mx) * (y - my) for x, y in zip(xs, vals)) / (sx * sy) if sx > 0 and sy > 0 else 0.0
        med = statistics.median(xs)
        lower = k in ("graph_node_us", "l2_latency_ns", "hbm_latency_ns", "corr_iter_us")
        norm = [v * (x / med if lower else med / x) for v, x in zip(vals, xs)]
        print(f"| {k} | {min(xs):.2f} .. {max(xs):.2f} | {100 * (max(xs) / min(xs) - 1):.1f} | {r:+.2f} | {100 * (max(norm) / min(norm) - 1):.1f} |")
