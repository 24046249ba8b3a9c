"""Per tree level of one frame: rays traced, shaded nodes, shadow queries, k_level duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import _native
from raytracing_amd.workloads import get_flat_workload
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_1080p_b4"
flat, opt = get_flat_workload(name)
ctx = _native.Context(0); ctx.upload(flat)
ctx.render_rows(opt, seed=0)
st = ctx.render_rows(opt, seed=0)["stats"]
L = len(flat.light_strength)
print(name, "lights", L, "queries", st["queries"], "ms_k_level %.4f" % st["ms_k_level"], "ms_device %.4f" % st["ms_device"])
for d in range(opt["bounces"] + 1):
    r, n, ms = st["level_rays"][d], st["level_nodes"][d], st["level_ms"][d]
    q = r + n * L
    print(f"  level {d}: rays {r:9d}  nodes {n:9d}  shadow {n * L:9d}  queries {q:9d}  {ms * 1e3:8.1f} us  {q / ms / 1e6 if ms else 0:8.2f} Gq/s  rounds {(r + 127) // 128 + (n * L + 127) // 128}")
