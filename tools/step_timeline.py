#!/usr/bin/env python3
"""Per-kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV (steps are delimited by the first kernel of
FeatureNet, stem_kernel; traces of older builds by compose_proj_kernel).  usage: step_timeline.py <kernel_trace.csv> [step_index]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
    if len(idx) < 2:
        idx = [i for i, r in enumerate(rows) if "compose_proj" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
    s, e = idx[k], idx[k + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    busy = 0
    agg = collections.Counter()
    for r in rows[s:e]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("itermvs::", "")[:60]
        grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d / 1e3:7.1f} {name:60s} {grid}")
        busy += d
        agg[name] += d
    print(f"busy {busy / 1e3:.1f} us, span {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for name, d in agg.most_common(12):
        print(f"  {d / 1e3:8.1f} us  {name}")


if __name__ == "__main__":
    main()
