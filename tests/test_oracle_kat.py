"""Function-level known answers captured from the reference (tests/golden/kat.json) against the
oracle's per-function entry points and against the host-side scene code."""
import math
import os

import numpy as np
import pytest

from conftest import flat_for, scene_path
from raytracing_amd import keyed_rng
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.scene import RGB, Euler, Scale, Scene, Vec3


def test_apply_transforms(kat, oracle_mod):
    import ctypes as C
    lib = oracle_mod.lib()
    for e in kat["apply_transforms"]:
        out = (C.c_double * 3)()
        lib.t3do_apply_transforms((C.c_double * 3)(*e["v"]), (C.c_double * 3)(*e["pos"]), (C.c_double * 3)(*e["scale"]),
                                  (C.c_double * 3)(*e["rot"]), out)
        assert list(out) == e["out"]
        m = (C.c_double * 9)()
        lib.t3do_euler_matrix((C.c_double * 3)(*e["rot"]), m)
        assert list(m) == e["matrix"]
        # host-side scene classes use the same arithmetic
        py = Vec3(*e["v"]).applyTransforms(Vec3(*e["pos"]), Scale(*e["scale"]), Euler(*e["rot"]))
        assert [*py] == e["out"]
        assert [float(x) for x in Euler(*e["rot"]).toMatrix().elements] == e["matrix"]


def test_camera_rays(kat, oracle_mod):
    scene = Scene.fromJSON(scene_path("cornellScene.json"))
    for e in kat["camera"]:
        scene.camera.rotation = Euler(*e["rot"])
        scene.camera.length = e["length"]
        flat = flatten_scene(scene)
        pv, start, d = oracle_mod.camera_ray(flat, dict(width=e["width"], height=e["height"]), e["px"], e["py"])
        assert pv == e["pixvec"] and start == e["start"] and d == e["dir"]


def test_face_collision(kat, oracle_mod):
    for e in kat["face_collision"]:
        r = oracle_mod.face_collision(e["start"], e["dir"], e["verts"])
        assert (r is not None) == e["hit"], e["name"]
        if r:
            assert r["t"] == e["t"] and r["point"] == e["point"] and r["normal"] == e["normal"]
            assert r["direction"] == e["direction"]
    names = {e["name"]: e for e in kat["face_collision"]}
    # the semantics that define the edge-tie pixels
    assert not names["edge_exact"]["hit"] and not names["vertex_exact"]["hit"] and not names["hypotenuse_exact"]["hit"]
    assert names["t_zero"]["hit"] and names["t_zero"]["t"] == 0.0
    assert not names["near_parallel_rejected"]["hit"]


def test_sphere_collision(kat, oracle_mod):
    for e in kat["sphere_collision"]:
        r = oracle_mod.sphere_collision(e["start"], e["dir"], e["center"], e["radius"], e["is_light"], e["back_cull"])
        assert (r is not None) == e["hit"], e["name"]
        if r:
            assert r["t"] == e["t"] and r["point"] == e["point"] and r["normal"] == e["normal"]
            assert r["direction"] == e["direction"]


def test_enters_aabb(kat, oracle_mod):
    for e in kat["enters_aabb"]:
        assert oracle_mod.enters_aabb(e["start"], e["dir"], e["bmin"], e["bmax"]) == e["result"], e["name"]


def test_bvh_and_baking(kat, oracle_mod):
    flat = flat_for("cornellScene.json")
    for e in kat["bvh"]:
        lo, hi = oracle_mod.object_box(flat, e["object"])
        assert [lo, hi] == e["box"], e["name"]
        if e["leaves"]:
            order, sizes = oracle_mod.bvh_leaf_order(flat, e["object"])
            assert order == [f for leaf in e["leaves"] for f in leaf]
            assert sizes == [len(leaf) for leaf in e["leaves"]]
            # baked vertices of the flattening step == Object.setTransforms of the reference
            scene = Scene.fromJSON(scene_path("cornellScene.json"))
            obj = scene.objects[e["object"]]
            baked = np.array(e["baked_vertices"])
            faces = np.array(obj.mesh.faces)
            b0 = flat.obj_tri_begin[e["object"]]
            assert np.array_equal(flat.tri_verts[b0:b0 + len(faces)], baked[faces])
            # the Python-side BVH of the scene classes has the same leaves
            obj.setTransforms()
            leaves = []

            def walk(node):
                if node.faces:
                    leaves.append([obj.mesh.faces.index(f) for f in node.faces])
                    return
                walk(node.left)
                walk(node.right)
            walk(obj.buildBVH())
            assert leaves == e["leaves"]


def test_light_list(kat):
    for e in kat["lights"]:
        flat = flat_for(e["scene"])
        assert len(flat.light_strength) == len(e["lights"])
        for k, l in enumerate(e["lights"]):
            assert flat.light_pos[k].tolist() == l["pos"]
            assert flat.light_strength[k] == l["strength"]
            assert flat.light_color[k].tolist() == l["color"]


def test_bounce_and_transmit_via_trace(kat, oracle_mod):
    """bounce()/transmit() are checked through a render: the child rays recorded by the oracle for the
    same (seed, pixel, sample, heap) key must equal the reference's.  Here: direction and start of the
    child given the parent hit, using the keyed draw."""
    for e in kat["bounce"] + [t for t in kat["transmit"] if not t["tir"]]:
        # the keyed triple is a pure function of the key
        u = [keyed_rng.keyed_uniform(e["seed"], e["px"], e["py"], e["sample"], e["heap"], c) for c in range(3)]
        bits = [oracle_mod.lib().t3do_keyed_bits24(e["seed"], e["px"], e["py"], e["sample"], e["heap"], c) for c in range(3)]
        assert u == [-1.0 + 2.0 * (b / 16777216.0) for b in bits]
        assert all(-1.0 <= x < 1.0 for x in u)
    tir = [t for t in kat["transmit"] if t["tir"]]
    assert tir, "fixture must contain a total-internal-reflection case"


def test_scalar_helpers(kat, oracle_mod):
    lib = oracle_mod.lib()
    for e in kat["attenuation"]:
        assert lib.t3do_attenuation(float(e["d"])) == e["out"]
    for e in kat["rgb_mean"]:
        n = len(e["colors"])
        for ch in range(3):
            total = 0
            for c in e["colors"]:
                total = total + c[ch]
            assert min(lib.t3do_py_floordiv(float(total), float(n)), 255) == e["mean"][ch]
        m = RGB.mean([RGB(*c) for c in e["colors"]])
        assert [float(m.r), float(m.g), float(m.b)] == e["mean"]
    for e in kat["mt_uniform"]:
        assert oracle_mod.mt_uniforms(e["seed"], len(e["values"])).tolist() == e["values"]


def test_keyed_rng_python_equals_c(oracle_mod):
    lib = oracle_mod.lib()
    rng = np.random.default_rng(1)
    for _ in range(200):
        seed, px, py = int(rng.integers(0, 2**32)), int(rng.integers(0, 8192)), int(rng.integers(0, 8192))
        sample, heap, comp = int(rng.integers(0, 64)), int(rng.integers(1, 1024)), int(rng.integers(0, 3))
        assert keyed_rng.keyed_bits24(seed, px, py, sample, heap, comp) == lib.t3do_keyed_bits24(seed, px, py, sample, heap, comp)


def test_fragility_of_crossing_faces(oracle_mod):
    """Test aid, not reference behaviour: a primary ray that meets two faces at all but the same distance (it runs past
    the line in which they cross) is reported as fragile, as a ray on a triangle's edge is; faces that coincide tie exactly
    in every precision and are not."""
    from raytracing_amd.scene import Camera, Mesh, Object, Shader

    def scene_of(tris):
        s = Scene()
        Camera(s, position=Vec3(0.001, 0, -5), length=50)
        verts = [Vec3(*p) for t in tris for p in t]
        o = Object(scene=s, name="faces", mesh=Mesh(verts, [(3 * k, 3 * k + 2, 3 * k + 1) for k in range(len(tris))]))   # facing the camera
        o.setShader(Shader(color=RGB(200, 200, 200), roughness=1.0, reflectivity=1.0))
        return flatten_scene(s)

    # two triangles far larger than the view, in the planes z = 3 + 0.1 x and z = 3 - 0.1 x: they cross in the line x = 0
    a = [(-50, -50, -2), (50, -50, 8), (0, 80, 3)]
    b = [(-50, -50, 8), (50, -50, -2), (0, 80, 3)]
    opt = dict(width=64, height=8, bounces=0)
    flat = scene_of([a, b])
    ids = oracle_mod.render(flat, opt)["prim"][:, :, 1]
    assert (ids[:, :30] == 0).all() and (ids[:, 34:] == 1).all()       # the nearer face on either side of the line
    frag = oracle_mod.primary_fragility(flat, opt)
    assert frag[:, 31:33].min() < 2e-4 and frag[:, :24].min() > 1e-3 and frag[:, 40:].min() > 1e-3
    # the same face twice: equal distances everywhere, nothing fragile about it
    frag2 = oracle_mod.primary_fragility(scene_of([a, a]), opt)
    assert frag2.min() > 1e-3
