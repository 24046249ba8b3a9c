import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from raytracing_amd import _native
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.workloads import synthetic_scene
from oracle import oracle
oracle.build()
flat = flatten_scene(synthetic_scene(n_random_tris=3000, grid=20, n_spheres=150, seed=1234, extent=12.0, camera=(0.3713, 0.2117, -28.8)))
opt = dict(width=96, height=54, bounces=2, samples=2)
ref = oracle.render(flat, opt, seed=7, want_nodes=True)
ctx = _native.Context(0); ctx.set_accel(sys.argv[1] if len(sys.argv) > 1 else "lbvh"); ctx.upload(flat)
out = ctx.render_rows(opt, seed=7, want_radiance=True, want_primary=True)
err = np.abs(out["radiance"].astype(np.float64) - ref["radiance"]).max(axis=2)
ys, xs = np.nonzero(err > 0.05)
by_key = {}
for n in ref["nodes"]:
    by_key.setdefault((int(n["px"]), int(n["py"]), int(n["sample"])), {})[int(n["heap"])] = n
for y, x in list(zip(ys, xs))[:6]:
    print("pixel", x, y, "err", err[y, x], out["radiance"][y, x], ref["radiance"][y, x])
    for s in range(2):
        nodes = ctx.trace_pixel(opt, int(x), int(y), s, seed=7)
        want = by_key.get((int(x), int(y), s), {})
        print("  sample", s, "gpu heaps", sorted(n.heap for n in nodes), "ref heaps", sorted(want))
        for n in nodes:
            w = want.get(n.heap)
            print("    heap", n.heap, "gpu", (n.obj, n.face, n.direction), "t %.6f" % n.t, "col", [round(c, 3) for c in n.color],
                  "| ref", None if w is None else ((int(w["obj"]), int(w["face"]), int(w["direction"])), "t %.6f" % w["t"], [round(float(c), 3) for c in w["color"]]))
