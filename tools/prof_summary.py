#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the launch sequence of the last frame."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0]
tot = collections.OrderedDict()
for r in rows:
    k = short(r["Kernel_Name"]); d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t = tot.setdefault(k, [0, 0, r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]])
    t[0] += 1; t[1] += d
print(f"{'kernel':28s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} vgpr sgpr lds scratch")
for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {t[0]:6d} {t[1]/1e3:10.1f} {t[1]/t[0]/1e3:9.2f} {t[2]:>4s} {t[3]:>4s} {t[4]:>3s} {t[5]:>3s}")
# a frame = the launches from a level-0 kernel on; the last one by default (bench.py times its levels one after the
# other there), --frame K for the K-th from the end (1 = the one before the last: untimed, kernels may overlap)
back = int(sys.argv[sys.argv.index("--frame") + 1]) if "--frame" in sys.argv else 0
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(("k_trace<true", "k_level<true", "k_bvh_trace<true"))]
idx = starts[-1 - back]
t0 = int(rows[idx]["Start_Timestamp"])
print("\nframe launch sequence (start offset us, duration us, grid):")
for n, r in enumerate(rows[idx:]):
    k = short(r["Kernel_Name"])
    if n > 0 and (not k.startswith("k_") or k.startswith(("k_trace<true", "k_level<true", "k_bvh_trace<true"))):
        break      # the frame's kernels are over (a copy, or the next frame's first kernel)
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f}  {k:24s} {r['Grid_Size_X']}")
