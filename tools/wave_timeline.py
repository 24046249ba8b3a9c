"""Timeline of the waves of every k_level launch of one benchmark frame, from a library built with -DT3D_WAVE_TRACE
(T3D_HIP_LIB=...): when the waves start and end (100 MHz wall clock), how many rounds they run, how many are resident.
usage: T3D_HIP_LIB=tools/exp/libt3d_wtrace.so python tools/wave_timeline.py [workload]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import _native
from raytracing_amd.workloads import get_flat_workload

workload = sys.argv[1] if len(sys.argv) > 1 else "cornell_1080p_b4"
flat, opt = get_flat_workload(workload)
ctx = _native.Context(0)
ctx.upload(flat)
H, W = opt["height"], opt["width"]
out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    st = ctx.render_rows_device(opt, 0, 0, H, out.data_ptr(), want_stats=True)
lib = _native.load()
lib.t3d_debug_wave_trace.argtypes = [C.c_int, C.c_void_p, C.c_int]
N = 1 << 16
print("level ms", [round(x, 4) for x in st["level_ms"][:opt["bounces"] + 1]])
for level in range(opt["bounces"] + 1):
    buf = np.zeros((N, 4), np.uint64)
    assert lib.t3d_debug_wave_trace(level, buf.ctypes.data, N) == 0
    used = buf[:, 1] > 0
    b = buf[used]
    if len(b) == 0:      # one launch per band (k_tree): its waves are recorded as level 0
        continue
    t0 = b[:, 0].min()
    start = (b[:, 0] - t0).astype(np.float64) / 100.0     # us
    end = (b[:, 1] - t0).astype(np.float64) / 100.0
    dur = end - start
    tr = (b[:, 2] & np.uint64(0xffffffff)).astype(int)
    sh = (b[:, 2] >> np.uint64(32)).astype(int)
    span = end.max()
    pct = lambda a: " ".join(f"{np.percentile(a, p):7.1f}" for p in (0, 10, 50, 90, 99, 100))
    print(f"level {level}: {len(b)} waves, span {span:.1f} us; start us (min p10 p50 p90 p99 max) {pct(start)}; end {pct(end)}; duration {pct(dur)}")
    print(f"    trace rounds per wave: mean {tr.mean():.2f} max {tr.max()}; shadow rounds: mean {sh.mean():.2f} max {sh.max()}; "
          f"us per round (duration / rounds): {pct(dur / np.maximum(1, tr + sh))}")
    ts = np.linspace(0, span, 21)[1:-1]
    res = [(int(((start <= t) & (end > t)).sum())) for t in ts]
    print("    resident waves at 5 % steps of the span:", res)
    hw = (b[:, 3] & np.uint64(0xffffffff)).astype(np.uint32)
    xcc = (b[:, 3] >> np.uint64(32)).astype(np.uint32) & 0xf
    cu = (hw >> 8) & 0xf
    sh_id = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    key = xcc.astype(int) * 4096 + se.astype(int) * 512 + sh_id.astype(int) * 256 + cu.astype(int) * 4 + simd.astype(int)
    uniq, counts = np.unique(key, return_counts=True)
    print(f"    distinct (xcc, se, sh, cu, simd): {len(uniq)}; waves per SIMD min {counts.min()} max {counts.max()}; xcc histogram {np.bincount(xcc, minlength=8).tolist()}")
    # how long is each SIMD busy, and how much work did it get
    by = {}
    for k, s_, e_, r_ in zip(key, start, end, tr + sh):
        a = by.setdefault(k, [1e9, 0, 0])
        a[0] = min(a[0], s_); a[1] = max(a[1], e_); a[2] += r_
    ends = np.array([v[1] for v in by.values()]); rounds = np.array([v[2] for v in by.values()])
    print(f"    per SIMD: last end us {pct(ends)}; rounds {pct(rounds)}")
    inp = (b[:, 3] >> np.uint64(40)).astype(int)
    if inp.any():
        print(f"    input rounds per wave (min p10 p50 p90 p99 max): {pct(inp)}")
        gws = np.nonzero(used)[0]
        sel = (gws % 64) == 5
        order = np.argsort(end[sel])
        print("    waves of shard 5 by end time: (end us, input rounds, trace rounds incl. input, shadow rounds, simd key)")
        print("     ", [(round(float(end[sel][i]), 1), int(inp[sel][i]), int(tr[sel][i]), int(sh[sel][i]), int(key[sel][i])) for i in order])
    late = start > 1.0
    if late.any():
        print(f"    waves that start later than 1 us: {int(late.sum())}, their start {pct(start[late])}")
