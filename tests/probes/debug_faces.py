import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.scene import Scene
from raytracing_amd.flatten import flatten_scene
from raytracing_amd import _native
from oracle import oracle
flat = flatten_scene(Scene.fromJSON(os.path.join(ROOT, "raytracing_amd/data/cornellScene.json")))
ctx = _native.Context(0); ctx.upload(flat)
rng = np.random.default_rng(0)
n = 3000
orig = rng.uniform(-3, 3, (n, 3)) + [0, 0, 1]
dirs = rng.normal(size=(n, 3))
g = ctx.closest_hits(orig, dirs, True, False)
bad = {}
tot = {}
for i in range(n):
    w = oracle.first_collision(flat, orig[i], dirs[i], True, False)
    key = (w["obj"], w["face"], w["direction"])
    tot[key] = tot.get(key, 0) + 1
    if (g["obj"][i], g["face"][i]) != (w["obj"], w["face"]):
        bad[key] = bad.get(key, 0) + 1
for k in sorted(tot):
    print(k, "oracle hits", tot[k], "gpu differs", bad.get(k, 0))
