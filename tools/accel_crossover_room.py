"""Brute force against the LBVH in a CLOSED room (the Cornell box with a tessellated glass ball of growing triangle count in
it, 1080p, 4 bounces): the other end of the compromise that tools/accel_crossover.py measures on sparse scenes."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _native
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.scene import RGB, Material, Mesh, Object, Scale, Shader, Vec3
from raytracing_amd import workloads as W


def ball_mesh(k):
    """UV sphere of k stacks and 2k sectors: 4 k (k - 1) triangles, wound outward."""
    verts, faces = [Vec3(0, 0.5, 0)], []
    for i in range(1, k):
        th = math.pi * i / k
        for j in range(2 * k):
            ph = math.pi * j / k
            verts.append(Vec3(0.5 * math.sin(th) * math.cos(ph), 0.5 * math.cos(th), 0.5 * math.sin(th) * math.sin(ph)))
    verts.append(Vec3(0, -0.5, 0))
    ring = lambda i, j: 1 + (i - 1) * 2 * k + j % (2 * k)
    for j in range(2 * k):
        faces.append((0, ring(1, j + 1), ring(1, j)))
        faces.append((len(verts) - 1, ring(k - 1, j), ring(k - 1, j + 1)))
    for i in range(1, k - 1):
        for j in range(2 * k):
            faces.append((ring(i, j), ring(i, j + 1), ring(i + 1, j)))
            faces.append((ring(i, j + 1), ring(i + 1, j + 1), ring(i + 1, j)))
    return Mesh(verts, faces)


dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev); torch.cuda.set_stream(s)
opt = dict(width=1920, height=1080, bounces=4, samples=1)
out = torch.empty((1080, 1920, 3), dtype=torch.uint8, device=dev)
for k in (3, 4, 5, 6, 7, 8):
    scene = W.cornell_scene(with_light=True)
    ball = Object(scene=scene, name="Ball", mesh=ball_mesh(k))
    ball.setShader(Shader(reflectivity=0.2, roughness=0.04, color=RGB(180, 180, 180), debugColor=RGB(180, 180, 180)))
    ball.material = Material(ior=1.4)
    ball.position = Vec3(-1.2, -1.6, 0.2)
    ball.scale = Scale(1.6, 1.6, 1.6)
    flat = flatten_scene(scene)
    res = {}
    for accel in ("brute", "lbvh"):
        ctx = _native.Context(0); ctx.set_accel(accel); ctx.upload(flat); ctx.set_timing(False)
        for _ in range(3):
            ctx.render_rows_device(opt, 0, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            ctx.render_rows_device(opt, i, 0, 1080, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        res[accel] = (time.perf_counter() - t0) / 20 * 1e3
        ctx.close()
    print(f"{flat.n_tris + flat.n_spheres:5d} primitives: brute {res['brute']:.3f} ms, lbvh {res['lbvh']:.3f} ms", flush=True)
