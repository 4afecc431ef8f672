#!/usr/bin/env python3
"""Where a short timed region of bench.py spends its time on the GPU: from a rocprofv3 --kernel-trace CSV of
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config0 --no-clustered --no-batched --no-single-lane --latency-queries 5
prints, for every burst of streamed 8-bit scans (bursts are separated by more than 12 us of idle GPU: the fences between the regions), the span from the first
kernel's start to the last kernel's end, the sum of the kernels' durations per hardware queue, and the timeline of the launches."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "mi355::" not in n:
        continue
    short = n.split("mi355::")[1].split("(")[0][:48]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short))
rows.sort()
bursts, cur = [], []
for r in rows:
    if cur and r[0] - max(x[1] for x in cur) > 12_000:
        bursts.append(cur)
        cur = []
    cur.append(r)
if cur:
    bursts.append(cur)
shown = 0
for b in bursts:
    scans = [x for x in b if x[3].startswith("scan_q8_kernel") and "true, true, false" in x[3]]
    if len(scans) < 18 or len(scans) > 26 or len(b) > 40:
        continue
    t0 = b[0][0]
    span = (max(x[1] for x in b) - t0) / 1e3
    busy = {}
    for x in b:
        busy[x[2]] = busy.get(x[2], 0) + (x[1] - x[0]) / 1e3
    print(f"burst of {len(b)} launches ({len(scans)} streamed scans): span {span:.1f} us, busy per queue {busy}")
    if shown < 2:
        for x in b:
            print(f"   q{x[2]} +{(x[0] - t0) / 1e3:7.1f} .. +{(x[1] - t0) / 1e3:7.1f}  ({(x[1] - x[0]) / 1e3:5.1f} us)  {x[3]}")
    shown += 1
