#!/usr/bin/env python3
"""Print the kernel sequence of the last full step from a rocprofv3 --kernel-trace CSV."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'convex_upsample' in r['Kernel_Name']]
a, b = ends[-2] + 2, ends[-1] + 2
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print(len(step), 'kernels; wall', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'us')
tot = 0
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = re.sub(r'void |itermvs::|\(.*', '', r['Kernel_Name'])[:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:7.0f} {d:6.1f} g=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {name}")
print('sum of kernel time', tot)
