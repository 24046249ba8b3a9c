"""A large mesh in a room: the suzanne scene with every triangle of the monkey cut into 4^k (midpoint subdivision), traced
with the hierarchy built on the device (Morton order) and on the host (surface area).  usage: mesh_in_room.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import _native
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.scene import RGB, Light, Material, Mesh, Object, Scale, Shader, Vec3
from raytracing_amd import workloads as W


def subdivided(mesh, k):
    v = np.array([[p.x, p.y, p.z] for p in mesh.vertices])
    f = np.array(mesh.faces, dtype=np.int64)
    for _ in range(k):
        tri = v[f]                                     # [F, 3, 3]
        m01, m12, m20 = (tri[:, 0] + tri[:, 1]) / 2, (tri[:, 1] + tri[:, 2]) / 2, (tri[:, 2] + tri[:, 0]) / 2
        parts = np.concatenate([np.stack([tri[:, 0], m01, m20], 1), np.stack([m01, tri[:, 1], m12], 1),
                                np.stack([m20, m12, tri[:, 2]], 1), np.stack([m01, m12, m20], 1)])
        v = parts.reshape(-1, 3)
        f = np.arange(len(v)).reshape(-1, 3)
    return Mesh([Vec3(*map(float, p)) for p in v], [tuple(int(i) for i in t) for t in f])


def scene_with(k):
    scene = W.Scene()
    W._cornell_box(scene)
    monkey = Object(scene=scene, name="Monkey", mesh=subdivided(Mesh.fromSTL(os.path.join(W.DATA_DIR, "suzanne.stl")), k))
    monkey.setShader(Shader(reflectivity=0.2, roughness=0.04, color=RGB(180, 180, 180), debugColor=RGB(180, 180, 180)))
    monkey.material = Material(ior=1.4)
    monkey.setTransforms()
    monkey.position = Vec3(1.44, -1.8, 1.4)
    monkey.scale = Scale(1.8, 1.8, 1.8)
    W._mirror_ball(scene)
    light = Light(scene=scene, name="Light0", position=Vec3(0, 3, 1), strength=1.0)
    light.setShader(Shader(color=RGB(200, 200, 200)))
    W._camera(scene)
    return scene


dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev); torch.cuda.set_stream(s)
opt = dict(width=1920, height=1080, bounces=4, samples=1)
out = torch.empty((1080, 1920, 3), dtype=torch.uint8, device=dev)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    flat = flatten_scene(scene_with(k))
    for where, limit in (("device", "0"), ("host", "100000000")):
        os.environ["T3D_HOST_TREE_MAX"] = limit
        ctx = _native.Context(0); ctx.set_timing(False)
        t0 = time.perf_counter(); ctx.upload(flat); up = time.perf_counter() - t0
        for i in range(3):
            ctx.render_rows_device(opt, i, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            ctx.render_rows_device(opt, i, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        st = ctx.render_rows(dict(opt, width=480, height=270), seed=0)["stats"]
        print(f"{flat.n_tris:7d} triangles, tree built on the {where:6s}: upload {up * 1e3:7.1f} ms, 1080p x 4 bounces {ms:6.3f} ms/frame, "
              f"node records per query {st['aabb_tests'] / max(1, st['queries'] + st['shadow_queries'] - st.get('shadow_culled', 0)):.2f}", flush=True)
        ctx.close()
