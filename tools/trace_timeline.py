#!/usr/bin/env python3
"""Print a compact timeline (start offset, duration, stream/queue) of the mfar kernels of the last few steps from a
rocprofv3 kernel-trace CSV."""
import csv, sys, glob, os
fn = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
ALL = "--all" in sys.argv      # also torch / runtime kernels (copies, fills)
sys.argv = [a for a in sys.argv if a != "--all"]
rows = [r for r in csv.DictReader(open(fn)) if (ALL or "mfar" in r["Kernel_Name"]) and "tile_rows" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -30:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:10.1f} us  +{(e-s)/1e3:9.1f} us  q={r.get('Queue_Id','?'):>3s} {r['Kernel_Name'].split('(')[0][:40]}")
