"""Why does a pixel of a random case differ from the oracle?  For a seed of tools/fuzz_parity.py: the pixel with the
largest radiance error, its trace trees on the device and in the oracle side by side, and the first node at which they
part.  usage: python tools/explain_drift.py seed [big] [deep]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from fuzz_parity import random_scene
from oracle import oracle
from raytracing_amd import _native
from raytracing_amd.flatten import flatten_scene

seed = int(sys.argv[1])
scene, opt = random_scene(seed, "big" in sys.argv[2:], "deep" in sys.argv[2:])
flat = flatten_scene(scene)
oracle.build()
ref = oracle.render(flat, opt, seed=seed, want_fragility=True, want_nodes=True, node_cap=4000000)
ctx = _native.Context(0)
ctx.set_accel("brute")
ctx.upload(flat)
out = ctx.render_rows(opt, seed=seed, want_radiance=True, want_primary=True)
err = np.abs(out["radiance"].astype(np.float64) - ref["radiance"]).max(axis=2)
y, x = np.unravel_index(np.argmax(err), err.shape)
print(f"seed {seed}: {opt}; pixel ({x}, {y}) radiance device {out['radiance'][y, x]} oracle {ref['radiance'][y, x]} error {err[y, x]:.4f} "
      f"fragility {ref['fragility'][y, x]}")
S = opt["samples"] if opt["bounces"] > 1 else 1
L = len(flat.light_strength)
for smp in range(S):
    dev = {n.heap: n for n in ctx.trace_pixel(opt, int(x), int(y), smp, seed=seed)}
    want = {int(n["heap"]): n for n in ref["nodes"] if (int(n["px"]), int(n["py"]), int(n["sample"])) == (x, y, smp)}
    print(f" sample {smp}: device nodes {sorted(dev)}  oracle nodes {sorted(want)}")
    for heap in sorted(set(dev) | set(want)):
        d, w = dev.get(heap), want.get(heap)
        if d is None or w is None:
            print(f"   heap {heap}: only in the {'oracle' if d is None else 'device'} tree")
            continue
        dc, wc = np.array(list(d.color)), np.array(w["color"])
        flag = "" if np.abs(dc - wc).max() < 0.05 else "   <-- colours differ"
        print(f"   heap {heap}: obj {d.obj}/{int(w['obj'])} face {d.face}/{int(w['face'])} t {d.t:.6f}/{float(w['t']):.6f} colour {np.round(dc, 3)} / {np.round(wc, 3)}{flag}")
        if flag and not any(h in dev and np.abs(np.array(list(dev[h].color)) - np.array(want[h]["color"])).max() >= 0.05
                            for h in (2 * heap, 2 * heap + 1) if h in want):
            # the children agree: the node's own light term differs.  Which light, and why?
            P, N = np.array(list(d.point), np.float64), np.array(list(d.normal), np.float64)
            N = N / np.linalg.norm(N)
            V = np.array(flat.cam_pos) - P
            for l in range(L):
                to = np.asarray(flat.light_pos[l]) - P
                Ld = to / np.linalg.norm(to)
                R = 2 * (Ld @ N) * N - Ld
                spec = (R / np.linalg.norm(R)) @ (V / np.linalg.norm(V))
                start = P + 0.2 * Ld
                w64 = oracle.first_collision(flat, start, Ld, True, False)
                g32 = ctx.closest_hits([start], [Ld], True, False)
                print(f"      light {l}: L.N {Ld @ N:+.7f}  R^.V^ {spec:.7f} (the highlight threshold is 0.96: margin {spec - 0.96:+.2e}); its shadow ray hits "
                      f"obj {int(g32['obj'][0])} face {int(g32['face'][0])} at {float(g32['t'][0]):.6g} on the device, obj {w64['obj']} face {w64['face']} at {w64['t']:.6g} in the oracle")
