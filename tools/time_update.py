"""A large scene with a moving Light: t3d_update_scene (spheres, Lights and camera rewritten in place, the hierarchy refitted)
before every frame.  usage (GPU box): python tools/time_update.py [workload]   (default synthetic_1m_4k_b2)
Prints the time of the update call itself, of update + frame, and of the frame alone."""
import os
import sys
import time
from dataclasses import replace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    from raytracing_amd import _native
    from raytracing_amd.flatten import OBJ_LIGHT
    from raytracing_amd.workloads import get_flat_workload
    name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_1m_4k_b2"
    flat, opt = get_flat_workload(name)
    out = torch.empty((opt["height"], opt["width"], 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream("cuda:0")
    ctx = _native.Context(0)
    ctx.upload(flat)
    light_objs = np.nonzero(np.asarray(flat.obj_type) == OBJ_LIGHT)[0]

    def moved(k):
        centre = np.array(flat.obj_center, copy=True)
        pos = np.array(flat.light_pos, copy=True)
        shift = np.array([0.05 * k, 0.0, 0.02 * k])
        centre[light_objs[0]] += shift
        pos[0] += shift
        return replace(flat, obj_center=centre, light_pos=pos, signature_digest=None)

    def frame(seed):
        ctx.render_rows_device(opt, seed, 0, opt["height"], out.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()

    for k in range(3):
        frame(k)
    t_frame, t_update, t_both = [], [], []
    for k in range(12):
        scene = moved(k + 1)
        t0 = time.perf_counter()
        ctx.update(scene)
        t1 = time.perf_counter()
        frame(k)
        t2 = time.perf_counter()
        frame(k)
        t3 = time.perf_counter()
        t_update.append((t1 - t0) * 1e3); t_both.append((t2 - t0) * 1e3); t_frame.append((t3 - t2) * 1e3)
    ctx.last_call_stats()
    ctx.close()
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{name}: t3d_update_scene returns after {med(t_update):.2f} ms (not waited for); update + frame {med(t_both):.2f} ms, "
          f"the frame alone {med(t_frame):.2f} ms -> a moved Light costs {med(t_both) - med(t_frame):.2f} ms per frame")


if __name__ == "__main__":
    main()
