#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel (per dispatch)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in [a for a in sys.argv[1:] if a.endswith(".csv")]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not k.startswith("k_"):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")

# --traffic-json PATH: HBM bytes per k_level launch from the FETCH_SIZE / WRITE_SIZE passes.
# Both counters are in KiB; on gfx950 FETCH_SIZE tallies wide (16 B per lane) reads at half their size,
# which is the access pattern of the ray / node records, so the read side is doubled.
if "--traffic-json" in sys.argv:
    import json
    out = sys.argv[sys.argv.index("--traffic-json") + 1]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cornell_1080p_b4"
    # one "launch" = the tracing of one tree level: a k_level launch, or k_bvh_trace + k_bvh_process + k_bvh_shadow
    def level_sum(counter):
        vals = [x for k, cs in acc.items() if k.startswith(("k_level", "k_bvh_")) for x in cs.get(counter, [])]
        levels = sum(len(cs.get(counter, [])) for k, cs in acc.items() if k.startswith(("k_level", "k_bvh_trace")))
        return (sum(vals) / levels, levels) if levels else (None, 0)
    (fk, nf), (wk, nw) = level_sum("FETCH_SIZE"), level_sum("WRITE_SIZE")
    # executed vector-ALU work of one tree level: wave instructions (SQ_INSTS_VALU) and how many of a wave's 64 lanes an
    # instruction had active on average (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; both count 4 cycles per instruction)
    (iv, ni), (tc, _), (ac, _) = level_sum("SQ_INSTS_VALU"), level_sum("SQ_THREAD_CYCLES_VALU"), level_sum("SQ_ACTIVE_INST_VALU")
    extra = {}
    if iv is not None:
        extra["valu_wave_insts_per_launch"] = int(iv)
        extra["valu_launches_sampled"] = ni
    if tc is not None and ac:
        extra["valu_active_lanes"] = round(tc / ac, 2)
    import datetime, subprocess
    try:
        extra["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        extra["commit"] = None
    extra["date"] = datetime.date.today().isoformat()
    # which kernel sources the counters belong to: bench.py uses the numbers only while the tree still holds these
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_digest
    extra["sources"] = source_digest()
    if fk is not None and wk is not None:
        json.dump({"kernel": "k_level", "workload": workload, "fetch_size_kib_per_launch": round(fk, 1),
                   "write_size_kib_per_launch": round(wk, 1), "fetch_correction": 2.0,
                   "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024), "launches_sampled": [nf, nw], **extra},
                  open(out, "w"), indent=1)
