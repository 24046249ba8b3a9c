import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from raytracing_amd import _native
from raytracing_amd.workloads import get_flat_workload
for wl in ("cornell_1080p_b4", "cornell2_8k_b8"):
    flat, opt = get_flat_workload(wl)
    if wl == "cornell2_8k_b8":
        opt = dict(opt); opt["width"] //= 4; opt["height"] //= 4
    ctx = _native.Context(0); ctx.set_accel("lbvh"); ctx.upload(flat)
    out = torch.empty((opt["height"], opt["width"], 3), dtype=torch.uint8, device="cuda")
    st = ctx.render_rows_device(opt, 0, 0, opt["height"], out.data_ptr(), want_stats=True)
    print(wl, "shadow", st["shadow_queries"], "culled", st["shadow_culled"], "frac", st["shadow_culled"] / max(1, st["shadow_queries"]), "queries", st["queries"])
