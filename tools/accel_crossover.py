"""Brute force against the LBVH on synthetic scenes of growing size (1080p, 2 bounces): where the automatic choice
(T3D_BRUTE_FORCE_MAX_PRIMS) should switch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _native
from raytracing_amd.workloads import synthetic_flat
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev); torch.cuda.set_stream(s)
opt = dict(width=1920, height=1080, bounces=2, samples=1)
out = torch.empty((1080, 1920, 3), dtype=torch.uint8, device=dev)
for n_random, grid, n_sph in ((40, 4, 4), (100, 6, 8), (200, 8, 12), (400, 10, 20), (800, 14, 40)):
    flat = synthetic_flat(n_random_tris=n_random, grid=grid, n_spheres=n_sph, seed=3, extent=12.0, camera=(0.37, 0.21, -28.8))
    res = {}
    for accel in ("brute", "lbvh"):
        ctx = _native.Context(0); ctx.set_accel(accel); ctx.upload(flat); ctx.set_timing(False)
        for _ in range(3):
            ctx.render_rows_device(opt, 0, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.render_rows_device(opt, 0, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        res[accel] = (time.perf_counter() - t0) / 20 * 1e3
        ctx.close()
    print(f"{flat.n_tris + flat.n_spheres:5d} primitives: brute {res['brute']:.3f} ms, lbvh {res['lbvh']:.3f} ms")
