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


# optional: --counters-json <out.json> <fetch_dir> <write_dir> <sq_dir> <source_hash> D F E Q N
#   -> per mfar kernel: HBM bytes per launch (2 x FETCH_SIZE KB + WRITE_SIZE KB) and MFMA utilisation
#      (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x shader cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs).
#   bench.py quotes these counters only when source_hash and shape match the run (profile_counters()).
if "--counters-json" in sys.argv:
    import json
    i = sys.argv.index("--counters-json")
    out, fdir, wdir, sdir, shash = sys.argv[i + 1:i + 6]
    shape = [int(x) for x in sys.argv[i + 6:i + 11]]
    def table(d, off=False):
        # <d>/off/ holds the --screen off pass of the same counter set: only its exact fp32 scan kernels are taken from there
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for fn in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            in_off = (os.sep + "off" + os.sep) in fn
            for r in csv.DictReader(open(fn)):
                exact = short(r["Kernel_Name"]).startswith(("mfar_stage1_kernel", "mfar_stage1_f32r"))      # the exact fp32 scan kernels
                if in_off != exact and os.path.isdir(os.path.join(d, "off")): continue
                k = short(r["Kernel_Name"])
                if "mfar" not in k: continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}, {k: max(cnt[k].values()) for k in cnt}
    tf, nf = table(fdir); tw, _ = table(wdir); ts, _ = table(sdir)
    kernels = {}
    for k in sorted(set(tf) | set(ts)):
        e = {"launches_in_fetch_pass": nf.get(k, 0)}
        if k in tf and "FETCH_SIZE" in tf[k]:
            rd = 2.0 * tf[k]["FETCH_SIZE"] * 1024
            wr = tw.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
            e.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
        s = ts.get(k, {})
        if s.get("GRBM_GUI_ACTIVE"):
            cyc = s["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_util"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * cyc)
            e["lds_bank_conflict_cycles"] = s.get("SQ_LDS_BANK_CONFLICT")
            e["sq_wave_cycles"] = s.get("SQ_WAVE_CYCLES")
        kernels[k] = e
    json.dump({"source_hash": shash, "shape": shape, "shape_is": "docs, fields, dim, query batch, gpus",
               "method": "rocprofv3 --kernel-trace --pmc, one pass each for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set; HBM read bytes = "
                         "2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md, HBM section); averages per launch",
               "kernels": kernels}, open(out, "w"), indent=1)
