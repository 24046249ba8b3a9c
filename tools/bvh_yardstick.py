"""Hierarchy yardstick for the large synthetic scene, on the host: writes the scene of a workload to a file, builds
tools/bvh_yardstick.cpp and runs it (no GPU needed).  usage: python tools/bvh_yardstick.py [workload] [pixel step]"""
import os
import subprocess
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.flatten import OBJ_LIGHT, OBJ_SPHERE
from raytracing_amd.workloads import get_flat_workload

workload = sys.argv[1] if len(sys.argv) > 1 else "synthetic_1m_4k_b2"
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flat, opt = get_flat_workload(workload)
tri = np.asarray(flat.tri_verts, np.float32).reshape(-1, 9)
types = np.asarray(flat.obj_type)
sph_objs = np.nonzero(types == OBJ_SPHERE)[0]        # Light objects are spheres to rays too, but four of them do not matter here
sph = np.concatenate([np.asarray(flat.obj_center, np.float32).reshape(-1, 3)[sph_objs], np.asarray(flat.obj_radius, np.float32)[sph_objs, None]], 1)
rough_of, refl_of = np.asarray(flat.shader_rough, np.float32), np.asarray(flat.shader_refl, np.float32)
tri_sh = np.asarray(flat.tri_shader)
sph_sh = np.asarray(flat.obj_shader0)[sph_objs]
rough = np.concatenate([rough_of[tri_sh], rough_of[sph_sh]]).astype(np.float32)
refl = np.concatenate([refl_of[tri_sh], refl_of[sph_sh]]).astype(np.float32)
lights = np.asarray(flat.light_pos, np.float32).reshape(-1, 3)
path = "/tmp/bvh_yardstick_scene.bin"
with open(path, "wb") as f:
    np.array([len(tri), len(sph), len(lights), opt["width"], opt["height"], step, opt["bounces"]], np.int32).tofile(f)
    np.array([*flat.cam_pos, flat.cam_zlen], np.float32).tofile(f)
    tri.tofile(f); sph.tofile(f); lights.tofile(f); rough.tofile(f); refl.tofile(f)
exe = "/tmp/bvh_yardstick"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "bvh_yardstick.cpp")])
subprocess.check_call([exe, path])
