"""End-to-end time of the reference-style API: Photon(scene, options).render() on the benchmark scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import T3DR
from raytracing_amd.workloads import get_workload
scene, opt = get_workload(sys.argv[1] if len(sys.argv) > 1 else "cornell_1080p_b4")
photon = T3DR.Photon(scene, dict(opt))
for i in range(4):
    t = time.perf_counter()
    r = photon.render()
    dt = time.perf_counter() - t
    print(f"render() call {i}: {dt * 1e3:.2f} ms wall, Rendered.time {r.time * 1e3:.3f} ms, rays {r.rays}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); photon.render(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
