"""End-to-end times of the reference-style API on the benchmark scene: Photon(scene, options).render() (first call:
flatten + upload + render; later calls of an unchanged scene: the device starts on the frame while the host
re-flattens the scene to see whether it changed) and single-ray picking (Photon.getFirstCollision, the raw ABI call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import T3DR
from raytracing_amd.workloads import get_workload
scene, opt = get_workload(sys.argv[1] if len(sys.argv) > 1 else "cornell_1080p_b4")
photon = T3DR.Photon(scene, dict(opt))
for i in range(6):
    t = time.perf_counter()
    r = photon.render()
    dt = time.perf_counter() - t
    print(f"render() call {i}: {dt * 1e3:.3f} ms wall, Rendered.time {r.time * 1e3:.3f} ms, rays {r.rays}, uploads so far {photon.uploads}")
times = []
for i in range(30):
    t = time.perf_counter(); photon.render(); times.append(time.perf_counter() - t)
print(f"render() of an unchanged scene: median {np.median(times) * 1e3:.3f} ms, best {min(times) * 1e3:.3f} ms over 30 calls")
ray = photon.getRay(photon.pixelToVector(opt["width"] // 2, opt["height"] // 2))
photon.getFirstCollision(ray, True, False)
times = []
for i in range(200):
    t = time.perf_counter(); photon.getFirstCollision(ray, True, False); times.append(time.perf_counter() - t)
print(f"Photon.getFirstCollision (one ray): median {np.median(times) * 1e6:.1f} us")
ctx = photon._context()
o = np.array([[*ray.start]], np.float32); d = np.array([[*ray.direction]], np.float32)
times = []
for i in range(200):
    t = time.perf_counter(); ctx.closest_hits(o, d, True, False); times.append(time.perf_counter() - t)
print(f"t3d_closest_hits through ctypes (one ray): median {np.median(times) * 1e6:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); photon.render(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)

# Picking on a LIVE scene (transformBefore False: every query looks at the scene as it is now, explorer.py:324-355) with
# one Light moved between two queries, on the glass-monkey scene (1018 triangles: traced through the hierarchy)
import T3DE
live_scene, live_opt = get_workload("suzanne_4k_b4")
tracer = T3DR.Photon(live_scene, dict(live_opt, transformBefore=False))
light = next(o for o in live_scene.objects if o.type == "light")
ray = tracer.getRay(tracer.pixelToVector(live_opt["width"] // 2, live_opt["height"] // 2))
tracer.getFirstCollision(ray, True, False)
for label, move in (("nothing changes between the queries", False), ("one Light moves before every query", True)):
    times = []
    for i in range(60):
        if move:
            light.position = T3DE.Vec3(0.3 * ((i % 5) - 2), 3.0, 1.0)
        t = time.perf_counter(); hit = tracer.getFirstCollision(ray, True, False); times.append(time.perf_counter() - t)
    print(f"picking on the live glass-monkey scene, {label}: median {np.median(times) * 1e6:.1f} us per query (uploads so far {tracer.uploads})")
