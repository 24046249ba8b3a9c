import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from raytracing_amd import _native
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.workloads import synthetic_scene
from oracle import oracle
oracle.build()
flat = flatten_scene(synthetic_scene(n_random_tris=3000, grid=20, n_spheres=150, seed=1234, extent=12.0, camera=(0.3713, 0.2117, -28.8)))
opt = dict(width=96, height=54, bounces=int(sys.argv[1]) if len(sys.argv) > 1 else 2, samples=2)
ref = oracle.render(flat, opt, seed=7)
frag = oracle.primary_fragility(flat, opt)
for mode in ("brute", "lbvh"):
    ctx = _native.Context(0); ctx.set_accel(mode); ctx.upload(flat)
    out = ctx.render_rows(opt, seed=7, want_radiance=True, want_primary=True)
    diff = np.abs(out["bgr"].astype(int) - ref["bgr"].astype(int)).max(axis=2)
    mism = (out["prim"] != ref["prim"]).any(axis=2)
    print(mode, "pix>1:", (diff > 1).sum(), "prim mism:", mism.sum(), "away from edges:", (mism & (frag > 1e-5)).sum(),
          "queries", out["stats"]["queries"], ref["stats"]["queries"], "rays", out["stats"]["rays_api"], ref["stats"]["rays_api"])
    ys, xs = np.nonzero(diff > 1)
    for y, x in list(zip(ys, xs))[:8]:
        print("   ", x, y, "gpu", out["bgr"][y, x], "ref", ref["bgr"][y, x], "prim", out["prim"][y, x], ref["prim"][y, x], "mism", mism[y, x])
    ctx.close()
