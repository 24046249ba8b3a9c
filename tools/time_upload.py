"""LBVH build time: upload of the 1M-triangle synthetic scene (host grouping + device build)."""
import sys, time
sys.path.insert(0, '.')
from raytracing_amd import _native
from raytracing_amd.workloads import get_flat_workload
flat, opt = get_flat_workload(sys.argv[1] if len(sys.argv) > 1 else "synthetic_1m_4k_b2")
ctx = _native.Context(0)
ctx.upload(flat)
for _ in range(2):
    t = time.time(); ctx.upload(flat); print("upload (auto) %.3f s" % (time.time() - t), "tris", flat.n_tris)
