#!/usr/bin/env python3
"""one-screen digest of a bench.py JSON line (tools/gpu_session.sh)"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(f"value {d['value']:.1f} {d['unit']}  ms/step {d['ms_per_step']:.4f} (min {d.get('ms_per_step_min', 0):.4f} max {d.get('ms_per_step_max', 0):.4f}, "
      f"{d.get('repeats', 1)} x {d['steps']} steps)  with_transfers {d.get('value_with_transfers')}")
if r:
    ci = r.get("corr_init") or {}
    print(f"corr_iter {r['avg_launch_ms'] * 1e3:.1f} us frac {r['frac']:.3f} traffic {r.get('traffic')}  corr_init {ci.get('avg_launch_ms', 0) * 1e3:.1f} us frac {ci.get('frac', 0):.3f}")
for k in ("staged_inputs", "pipelined", "with_transfers", "roofline_conv", "cpu_baseline"):
    v = d.get(k)
    if v:
        extra = ""
        if k == "with_transfers" and "uint8_images" in v:
            extra = f"  uint8 {v['uint8_images']['value']:.1f} ({v['uint8_images']['ms_per_step']:.3f} ms)"
        if k == "pipelined" and "with_transfers" in v:
            extra = f"  with transfers {v['with_transfers']['value']:.1f} ({v['with_transfers']['depth_maps_in_flight']} in flight)"
        print(f"{k}: {v.get('value', v.get('achieved')):.2f} {v.get('unit', '')} {('ms/step %.3f' % v['ms_per_step']) if 'ms_per_step' in v else ''}{extra}")
print("workload:", d["config"]["workload"])
if d.get("box"):
    print("box:", {k: (round(v, 2) if isinstance(v, (int, float)) else v) for k, v in d["box"].items()}, " normalised:", d.get("value_normalised"), " affinity:", d["config"].get("cpu_affinity"))
