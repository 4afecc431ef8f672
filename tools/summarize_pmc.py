#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv files -> one small JSON under profiles/:
  python tools/summarize_pmc.py profiles/r03_pmc_x.json "<command that was profiled>" <kernel substring> <dir> [<dir> ...]
Means per launch and launch counts per (kernel, counter); every directory is one --pmc pass of the same command."""
import collections
import csv
import glob
import json
import os
import sys

out_path, command, needle = sys.argv[1:4]
dirs = sys.argv[4:]
kernels = collections.defaultdict(dict)
for d in dirs:
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if needle in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            kernels[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump({"command": command,
           "units": "SQ_INSTS_* wave-instructions, SQ_*_CYCLES / SQ_WAIT_* quad-cycles summed over waves, GRBM_GUI_ACTIVE cycles "
                    "summed over the 8 XCDs, FETCH_SIZE / WRITE_SIZE KiB (FETCH_SIZE counts 64 B per 128-B request of a wide "
                    "coalesced read on gfx950: read bytes = 2 x FETCH_SIZE x 1024); means per launch",
           "kernels": kernels}, open(out_path, "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"], 1) for c, v in cs.items()} for k, cs in kernels.items()}, indent=1))
