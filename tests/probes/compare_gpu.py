#!/usr/bin/env python3
"""Developer probe: render every golden case on the GPU and print how it differs from the CPU oracle."""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracing_amd.scene import Scene
from raytracing_amd.flatten import flatten_scene
from raytracing_amd import _native
from oracle import oracle

def main():
    print("devices", _native.device_count())
    ctx = _native.Context(0)
    for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/g*.npz"))):
        g = np.load(f); meta = json.loads(str(g["meta"]))
        path = os.path.join(ROOT, "raytracing_amd/data", meta["scene"])
        flat = flatten_scene(Scene.fromJSON(path if os.path.exists(path) else os.path.join(ROOT, "tests/golden/scenes", meta["scene"])))
        opt = dict(meta["options"])
        seed = meta["seed"]
        ref = oracle.render(flat, opt, seed=seed, rng_mode=0)
        ctx.upload(flat)
        t0 = time.time()
        out = ctx.render_rows(opt, seed=seed, want_radiance=True, want_primary=True)
        dt = time.time() - t0
        diff = np.abs(out["bgr"].astype(int) - ref["bgr"].astype(int)).max(axis=2)
        n = diff.size
        idm = (out["prim"] != ref["prim"]).any(axis=2)
        raderr = np.abs(out["radiance"].astype(np.float64) - ref["radiance"])
        st, so = out["stats"], ref["stats"]
        print(f"{os.path.basename(f):48s} px {n:6d} >1LSB {int((diff>1).sum()):5d} ({100.0*(diff>1).sum()/n:.3f}%) "
              f"=1LSB {int((diff==1).sum()):5d} idmis {int(idm.sum()):3d} rad max {raderr.max():.3g} p99.9 {np.quantile(raderr,0.999):.3g} "
              f"rays {st['rays_api']} vs {so['rays_api']} q {st['queries']} vs {so['queries']} nodes {st['nodes']} "
              f"ms_dev {st['ms_device']:.3f} wall {dt*1e3:.1f}ms")
        if idm.sum():
            ys, xs = np.nonzero(idm)
            for y, x in list(zip(ys, xs))[:6]:
                print("     id mismatch at", (x, y), "gpu", out["prim"][y, x], out["prim_t"][y, x], "oracle", ref["prim"][y, x], ref["prim_t"][y, x])
        if (diff > 1).sum():
            ys, xs = np.nonzero(diff > 1)
            for y, x in list(zip(ys, xs))[:6]:
                print("     px", (x, y), "gpu", out["bgr"][y, x], "oracle", ref["bgr"][y, x], "rad", out["radiance"][y, x], ref["radiance"][y, x])

if __name__ == "__main__":
    main()
