#!/usr/bin/env python3
"""Function-level known answers captured from the reference -> tests/golden/kat.json.

    python tests/golden/make_kat.py --ref /root/reference

Each entry stores the inputs and what the reference function returned (floats as JSON numbers,
which round-trip exactly).  See make_golden.py for how the reference is imported.
"""
import argparse
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import KeyedRandom, import_reference  # noqa: E402


def v(x):
    return None if x is None else [float(c) for c in x]


def main(ref):
    T3DE, T3DR = import_reference(ref)
    T3DE.Vec3.__rmul__ = lambda self, k: self * k
    V, R = T3DE.Vec3, T3DE.Ray
    kat = {}

    # --- transforms -------------------------------------------------------------------------
    kat["apply_transforms"] = []
    for vec, pos, scale, rot in [((1.0, 2.0, 3.0), (0.5, -1.5, 2.0), (1.4, 2.6, 1.8), (0.3, 0.17453292519943295, -1.1)),
                                 ((-0.25, 0.75, 1.0), (0, 0, 0), (1, 1, 1), (0, 0, 0)),
                                 ((2.5, -1.0, 0.125), (1.44, -0.9, 1.4), (1.8, 1.8, 1.8), (math.radians(90), math.radians(-45), math.radians(10)))]:
        out = V(*vec).applyTransforms(V(*pos), T3DE.Scale(*scale), T3DE.Euler(*rot))
        kat["apply_transforms"].append(dict(v=list(vec), pos=list(pos), scale=list(scale), rot=list(rot), out=v(out),
                                            matrix=[float(e) for e in T3DE.Euler(*rot).toMatrix().elements]))

    # --- camera -----------------------------------------------------------------------------
    scene = T3DE.Scene.fromJSON(os.path.join(ref, "cornellScene.json"))
    kat["camera"] = []
    for (w, h, rot, length) in [(160, 120, (0, 0, 0), 20.0), (1920, 1080, (0.07, -0.2, 0.03), 35.0), (64, 36, (0, 0, 0), 1.0)]:
        scene.camera.rotation = T3DE.Euler(*rot)
        scene.camera.length = length
        ph = T3DR.Photon(scene, {"width": w, "height": h})
        for (px, py) in [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (w // 2, h // 2), (w // 3, h // 5)]:
            pv = ph.pixelToVector(px, py)
            ray = ph.getRay(pv)
            kat["camera"].append(dict(width=w, height=h, rot=list(rot), length=length, pos=v(scene.camera.position),
                                      px=px, py=py, pixvec=v(pv), start=v(ray.start), dir=v(ray.direction)))
    scene.camera.rotation = T3DE.Euler(0, 0, 0)
    scene.camera.length = 20.0

    # --- ray / triangle -----------------------------------------------------------------------
    ph = T3DR.Photon(scene, {"width": 160, "height": 120})
    tri = [(0.0, 0.0, 5.0), (2.0, 0.0, 5.0), (0.0, 2.0, 5.0)]
    cases = {
        "back": ((0.5, 0.5, 0.0), (0.0, 0.0, 1.0)),   # normal = (0,0,4): dir.normal > 0 -> direction 1
        "front": ((0.5, 0.5, 10.0), (0.0, 0.0, -1.0)),
        "edge_exact": ((1.0, 0.0, 0.0), (0.0, 0.0, 1.0)),
        "vertex_exact": ((0.0, 0.0, 0.0), (0.0, 0.0, 1.0)),
        "hypotenuse_exact": ((1.0, 1.0, 0.0), (0.0, 0.0, 1.0)),
        "parallel": ((0.5, 0.5, 0.0), (1.0, 0.0, 0.0)),
        "near_parallel_rejected": ((0.5, 0.5, 0.0), (1.0, 0.0, 1e-7)),
        "t_zero": ((0.5, 0.5, 5.0), (0.0, 0.0, 1.0)),
        "behind": ((0.5, 0.5, 6.0), (0.0, 0.0, 1.0)),
        "oblique": ((-3.0, 0.2, 0.0), (0.55, 0.1, 0.8)),
        "outside": ((3.0, 3.0, 0.0), (0.0, 0.0, 1.0)),
    }
    kat["face_collision"] = []
    for name, (start, d) in cases.items():
        ray = R(V(*start), V(*d), 1, None)
        res = ph.getFaceCollision(ray, [V(*p) for p in tri])
        kat["face_collision"].append(dict(name=name, start=list(start), dir=list(d), verts=[list(p) for p in tri],
                                          hit=res is not None,
                                          t=None if res is None else float(res["t"]),
                                          point=None if res is None else v(res["point"]),
                                          normal=None if res is None else v(res["normal"]),
                                          direction=None if res is None else res["direction"]))

    # --- ray / sphere -------------------------------------------------------------------------
    kat["sphere_collision"] = []
    sph = T3DE.Sphere(T3DE.Scene(), position=V(-1.54, -2.1, 2))
    sph.scale = T3DE.Scale(3, 3, 3)
    lig = T3DE.Light(T3DE.Scene(), position=V(-1.54, -2.1, 2))
    lig.scale = T3DE.Scale(3, 3, 3)
    for name, obj, start, d, cull in [
        ("outside_front", sph, (0, 0, -11), (-0.1, -0.15, 1.0), True),
        ("inside_nocull", sph, (-1.5, -2.0, 2.1), (0.3, 0.2, 0.5), False),
        ("inside_cull", sph, (-1.5, -2.0, 2.1), (0.3, 0.2, 0.5), True),
        ("light_outside", lig, (0, 0, -11), (-0.1, -0.15, 1.0), False),
        ("light_inside", lig, (-1.5, -2.0, 2.1), (0.3, 0.2, 0.5), False),
        ("miss", sph, (0, 0, -11), (0.5, 0.5, 1.0), True),
        ("behind", sph, (0, 0, 11), (-0.1, -0.15, 1.0), False),
        ("tangent_grazing", sph, (-0.04, -5.0, 2.0), (0.0, 1.0, 0.0), True),
    ]:
        ray = R(V(*start), V(*d), 1, None)
        res = ph.getSphereCollision(ray, obj, cull)
        kat["sphere_collision"].append(dict(name=name, start=list(start), dir=list(d), center=v(obj.position),
                                            radius=obj.scale.x / 2, is_light=obj.type == "light", back_cull=cull,
                                            hit=res is not None, t=None if res is None else float(res["t"]),
                                            point=None if res is None else v(res["point"]),
                                            normal=None if res is None else v(res["normal"]),
                                            direction=None if res is None else res["direction"]))

    # --- ray / box ----------------------------------------------------------------------------
    kat["enters_aabb"] = []
    box = (V(-1.0, -2.0, 3.0), V(1.0, 2.0, 5.0))
    for name, start, d in [("through", (0, 0, 0), (0.1, 0.2, 1.0)), ("axis_parallel_inside_slab", (0.5, 1.0, 0), (0, 0, 1)),
                           ("axis_parallel_outside_slab", (1.5, 1.0, 0), (0, 0, 1)), ("origin_inside", (0, 0, 4), (1, 0, 0)),
                           ("behind", (0, 0, 6), (0, 0, 1)), ("miss", (0, 0, 0), (1, 0, 0.2)),
                           ("on_face_plane_parallel", (1.0, 0.0, 0.0), (0, 0, 1)), ("touch_corner", (-2.0, -3.0, 2.0), (1, 1, 1))]:
        ray = R(V(*start), V(*d), 1, None)
        kat["enters_aabb"].append(dict(name=name, start=list(start), dir=list(d), bmin=v(box[0]), bmax=v(box[1]),
                                       result=bool(ray.entersAABB(box))))

    # --- BVH of the Cornell box mesh (50 faces) and the cube (12) -------------------------------
    import copy
    ts = copy.deepcopy(scene)
    kat["bvh"] = []
    for oi, obj in enumerate(ts.objects):
        if obj.type != "mesh":
            continue
        obj.setTransforms()
        root = obj.buildBVH()
        leaves = []

        def walk(node):
            if node.faces:
                leaves.append([obj.mesh.faces.index(f) for f in node.faces])
                return
            if node.left:
                walk(node.left)
            if node.right:
                walk(node.right)
        walk(root)
        kat["bvh"].append(dict(object=oi, name=obj.name, box=[v(root.box[0]), v(root.box[1])], leaves=leaves,
                               baked_vertices=[v(p) for p in obj.mesh.vertices]))
    for oi, obj in enumerate(ts.objects):
        if obj.type in ("sphere", "light"):
            b = obj.buildBVH().box
            kat["bvh"].append(dict(object=oi, name=obj.name, box=[v(b[0]), v(b[1])], leaves=[]))

    # --- lights list -----------------------------------------------------------------------------
    kat["lights"] = []
    for fname in ["cornellScene.json", "cornellScene2.json"]:
        sc = T3DE.Scene.fromJSON(os.path.join(ref, fname))
        p = T3DR.Photon(sc, {"width": 16, "height": 12})
        p.transformedScene = copy.deepcopy(sc)
        for o in p.transformedScene.objects:
            if o.type == "mesh":
                o.setTransforms()
        kat["lights"].append(dict(scene=fname, lights=[dict(pos=v(l[0]), strength=float(l[1]), color=v(l[2]))
                                                        for l in p.getAllLightsPos()]))

    # --- bounce / transmit ------------------------------------------------------------------------
    rng = KeyedRandom(11)
    T3DR.random = rng
    sc = copy.deepcopy(scene)
    for o in sc.objects:
        if o.type == "mesh":
            o.setTransforms()
    for o in sc.objects:
        o.bvh = o.buildBVH()
    p = T3DR.Photon(sc, {"width": 16, "height": 12})
    p.transformedScene = sc
    cube = [o for o in sc.objects if o.name == "cube"][0]
    ball = [o for o in sc.objects if o.type == "sphere"][0]
    kat["bounce"] = []
    kat["transmit"] = []

    def ci_of(ray, lights=True, cull=True):
        return p.getFirstCollision(ray, lights, cull)

    rays = {
        "to_ball": R(V(0, 0, -11), V(-1.54, -2.1, 13).normalise(), 1, None),
        "to_cube": R(V(0, 0, -11), V(1.3, -0.9, 12.0).normalise(), 1, None),
        "to_wall": R(V(0, 0, -11), V(-0.25, 0.05, 1.0).normalise(), 1, None),
        "graze_cube": R(V(0, 0, -11), V(0.82, -0.3, 11.6).normalise(), 1, None),
    }
    for name, ray in rays.items():
        ci = ci_of(ray)
        if ci["object"] is None:
            continue
        heap = 2
        rng.set_key(3, 4, 0, heap)
        out = p.bounce(ray, ci)
        oi = sc.objects.index(ci["object"])
        fi = ci["object"].mesh.faces.index(ci["face"]) if ci["object"].type == "mesh" else -1
        kat["bounce"].append(dict(name=name, seed=11, px=3, py=4, sample=0, heap=heap, start=v(ray.start),
                                  dir=v(ray.direction), obj=oi, face=fi, point=v(ci["point"]), normal=v(ci["normal"]),
                                  out_start=v(out.start), out_dir=v(out.direction)))
        heap = 3
        rng.set_key(3, 4, 0, heap)
        tr = p.transmit(ray, ci)
        entry = dict(name=name + "_enter", seed=11, px=3, py=4, sample=0, heap=heap, start=v(ray.start),
                     dir=v(ray.direction), ray_ior=1, ray_obj=-1, obj=oi, face=fi, point=v(ci["point"]),
                     normal=v(ci["normal"]), tir=tr is None)
        if tr is not None:
            entry.update(out_start=v(tr.start), out_dir=v(tr.direction), out_ior=float(tr.ior),
                         out_obj=-1 if tr.object is None else sc.objects.index(tr.object))
        kat["transmit"].append(entry)
        # continue inside the object (exit / total internal reflection)
        if tr is not None and tr.object is not None:
            ci2 = p.getFirstCollision(tr, True, tr.ior == 1)
            if ci2["object"] is not None:
                rng.set_key(3, 4, 0, 7)
                tr2 = p.transmit(tr, ci2)
                oi2 = sc.objects.index(ci2["object"])
                fi2 = ci2["object"].mesh.faces.index(ci2["face"]) if ci2["object"].type == "mesh" else -1
                e2 = dict(name=name + "_exit", seed=11, px=3, py=4, sample=0, heap=7, start=v(tr.start),
                          dir=v(tr.direction), ray_ior=float(tr.ior), ray_obj=sc.objects.index(tr.object), obj=oi2,
                          face=fi2, point=v(ci2["point"]), normal=v(ci2["normal"]), tir=tr2 is None)
                if tr2 is not None:
                    e2.update(out_start=v(tr2.start), out_dir=v(tr2.direction), out_ior=float(tr2.ior),
                              out_obj=-1 if tr2.object is None else sc.objects.index(tr2.object))
                kat["transmit"].append(e2)
    # a steep ray inside the glass cube -> total internal reflection
    inside = R(V(1.44, -0.9, 1.4), V(0.05, 1.0, 0.72).normalise(), 1.4, cube)
    ci3 = p.getFirstCollision(inside, True, False)
    rng.set_key(3, 4, 0, 9)
    tr3 = p.transmit(inside, ci3)
    kat["transmit"].append(dict(name="steep_inside", seed=11, px=3, py=4, sample=0, heap=9, start=v(inside.start),
                                dir=v(inside.direction), ray_ior=1.4, ray_obj=sc.objects.index(cube),
                                obj=sc.objects.index(ci3["object"]),
                                face=ci3["object"].mesh.faces.index(ci3["face"]), point=v(ci3["point"]),
                                normal=v(ci3["normal"]), tir=tr3 is None))

    # --- small scalar functions ---------------------------------------------------------------------
    kat["attenuation"] = [dict(d=d, out=float(p.getAttenuation(d))) for d in [0, 0.5, 3.0, 11.984035929035942, 100.0, 1e9]]
    kat["rgb_mean"] = []
    for cols in [[(10.5, 20.25, 255.0)], [(10.5, 20.25, 255.0), (11.0, 21.0, 255.0)], [(1, 2, 3), (2, 3, 5), (4, 4, 4)],
                 [(254.9, 0.3, 127.5), (255.0, 0.9, 128.4), (255.0, 0.1, 129.2)], [(20, 20, 20), (20, 20, 20)]]:
        m = T3DE.RGB.mean([T3DE.RGB(*c) for c in cols])
        kat["rgb_mean"].append(dict(colors=[list(c) for c in cols], mean=[float(m.r), float(m.g), float(m.b)]))
    kat["mt_uniform"] = []
    for seed in [0, 5, 123456789, 4294967295]:
        random.seed(seed)
        kat["mt_uniform"].append(dict(seed=seed, values=[random.uniform(-1, 1) for _ in range(8)]))

    # Post filters (T3DR.py:1550-1647) on a small fixed image
    import numpy as np
    rng = np.random.default_rng(2024)
    img = rng.integers(0, 256, (7, 9, 3))
    img[2, 2:7] = img[2, 2]                      # a run of equal pixels
    img[4, 1:5] = img[4, 1] + np.array([[3, -2, 1]])  # a run within the default tolerance
    img = np.clip(img, 0, 255)
    as_list = img.tolist()
    comp = T3DR.Post.compressRenderedDict({"imgData": as_list, "width": 9, "height": 7}, tolerance=10)
    comp0 = T3DR.Post.compressRenderedDict({"imgData": as_list, "width": 9, "height": 7}, tolerance=0)
    kat["post"] = dict(image=as_list, compressed_tol10=comp["imgData"], compressed_tol0=comp0["imgData"],
                       decompressed_tol10=T3DR.Post.decompressRenderedDict(comp)["imgData"],
                       upscaled_x2=T3DR.Post.upscale(img.astype(np.uint8), 2).tolist(),
                       denoised_k3=[[list(map(int, px)) for px in row] for row in T3DR.Post.denoise(as_list, 3)])

    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f)
    print({k: len(val) for k, val in kat.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    main(ap.parse_args().ref)
