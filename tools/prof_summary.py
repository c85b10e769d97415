#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel trace / stats / counter collection) to a small text summary.
usage: prof_summary.py <dir> [<dir> ...]   (prints to stdout)"""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    return name.split("(")[0][:70]

for d in [a for a in sys.argv[1:] if not a.startswith('--') and os.path.isdir(a)]:
    print(f"== {d}")
    for fn in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"-- {os.path.basename(fn)} (rocprofv3 --kernel-trace --stats)")
        rows = list(csv.DictReader(open(fn)))
        for r in rows[:25]:
            print(f"{short(r['Name']):72s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}")
    for fn in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print(f"-- {os.path.basename(fn)} (rocprofv3 --pmc)")
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if not k.startswith("mfar") and "mfar" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        for k in acc:
            for c in acc[k]:
                print(f"{k:60s} {c:28s} dispatches={cnt[k][c]:5d} avg_per_dispatch={acc[k][c]/cnt[k][c]:.6g}")
    for fn in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        dur = defaultdict(list); meta = {}
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if "mfar" not in k: continue
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
        if dur: print(f"-- {os.path.basename(fn)} (mfar kernels only)")
        for k, v in dur.items():
            print(f"{k:60s} n={len(v):4d} avg_us={sum(v)/len(v)/1e3:10.1f} min_us={min(v)/1e3:10.1f} max_us={max(v)/1e3:10.1f} vgpr/agpr/sgpr/lds/grid/wg={meta[k]}")


# optional: --traffic-json <out.json> <fetch_dir> <write_dir> [kernel name prefix]  -> per-launch HBM bytes of that kernel
if "--traffic-json" in sys.argv:
    import json
    i = sys.argv.index("--traffic-json")
    out, fdir, wdir = sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3]
    kern = sys.argv[i + 4] if len(sys.argv) > i + 4 else "mfar_stage1_kernel"
    def avg(d, counter):
        fn = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fn))
                if r["Kernel_Name"].startswith(kern) and r["Counter_Name"] == counter]
        return sum(vals) / len(vals), len(vals)
    f, nf = avg(fdir, "FETCH_SIZE")
    w, nw = avg(wdir, "WRITE_SIZE")
    json.dump({"kernel": kern, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": [nf, nw],
               "hbm_read_bytes_per_launch": 2.0 * f * 1024, "hbm_write_bytes_per_launch": w * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE reports half "
                       "the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE x 1024"},
              open(out, "w"), indent=1)
