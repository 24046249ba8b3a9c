"""Host-side scene model and the Photon/Rendered surface (no GPU needed)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import scene_path
from raytracing_amd.flatten import flatten_scene, make_opts
from raytracing_amd.scene import (RGB, Camera, Cube, Euler, Light, Material, Mesh, Object, Ray, Scale, Scene, Shader,
                                  Sphere, Vec3)
from raytracing_amd.workloads import cornell_scene


def test_drop_in_module_names():
    import T3DE
    import T3DR
    for name in ("Scene", "Object", "Mesh", "Sphere", "Light", "Camera", "Shader", "Material", "Vec3", "RGB", "Cube",
                 "Scale", "Euler", "Ray", "TraceTreeNode", "BVHNode", "Matrix9", "MathematicalPlane", "MathematicalSphere"):
        assert hasattr(T3DE, name)
    for name in ("Photon", "Rendered", "NoBoundCameraError", "ThreadsProgress", "Debug", "Texel", "Post"):
        assert hasattr(T3DR, name)


def test_cornell_json_loads_unchanged():
    scene = Scene.fromJSON(scene_path("cornellScene.json"))
    assert [o.type for o in scene.objects] == ["camera", "mesh", "mesh", "sphere"]   # bound camera first
    assert scene.camera is scene.objects[0] and scene.camera.length == 20.0
    box = scene.objects[1]
    assert len(box.mesh.faces) == 50 and len(box.shaders) == 5
    # shader assignment of the demo script: faces 48,49 emissive, 28,29 / 26,27 coloured walls
    idx = box.shaderIndices
    assert idx[48] == idx[49] == 2 and idx[28] == idx[29] == 3 and idx[26] == idx[27] == 4
    assert all(i == 1 for k, i in enumerate(idx) if k not in (26, 27, 28, 29, 48, 49))
    old = Scene.fromJSON(scene_path("cornellScene2.json"))                # older schema
    assert old.camera.length == 1.0 and len(old.objects) == 4


def test_stl_import_reproduces_reference_mesh():
    """cornellbox.stl -> swap y/z, reverse winding, rotate 90 degrees about y == the Box of cornellScene.json,
    face by face, exactly."""
    built = flatten_scene(cornell_scene(with_light=False))
    loaded = flatten_scene(Scene.fromJSON(scene_path("cornellScene.json")))
    assert np.array_equal(built.tri_verts, loaded.tri_verts)
    assert np.array_equal(built.shader_rough[built.tri_shader], loaded.shader_rough[loaded.tri_shader])
    assert np.array_equal(built.shader_emis[built.tri_shader], loaded.shader_emis[loaded.tri_shader])
    for name in ("config2_cornell_light.json", "config2_suzanne.json"):
        ref = flatten_scene(Scene.fromJSON(scene_path(name)))
        mine = flatten_scene(cornell_scene(with_light=True, suzanne="suzanne" in name))
        assert np.array_equal(mine.tri_verts, ref.tri_verts)
        assert np.array_equal(mine.light_pos, ref.light_pos)


def test_scene_json_round_trip(tmp_path):
    scene = cornell_scene(with_light=True)
    path = tmp_path / "s.json"
    scene.saveJSON(str(path))
    back = Scene.fromJSON(str(path))
    a, b = flatten_scene(scene), flatten_scene(back)
    assert np.array_equal(a.tri_verts, b.tri_verts)
    assert np.array_equal(a.light_pos, b.light_pos)
    assert sorted(a.obj_type.tolist()) == sorted(b.obj_type.tolist())


def test_get_shader_rules():
    scene = Scene()
    cube = Cube(scene)
    red = Shader(color=RGB(255, 0, 0))
    cube.setShader(red, [3])
    assert cube.getShader(cube.mesh.faces[3]) is red
    assert cube.getShader(cube.mesh.faces[0]) is cube.shaders[0]
    # duplicate index triple: the first occurrence decides (faces.index)
    cube.mesh.faces.append(cube.mesh.faces[3])
    cube.shaderIndices.append(0)
    assert cube.getShader(cube.mesh.faces[-1]) is red
    flat = flatten_scene(_with_camera(scene))
    assert flat.tri_shader[12] == flat.tri_shader[3]
    assert Sphere(scene).getShader() is not None
    assert Light(scene).getShader() is None and Camera(scene).getShader() is None


def _with_camera(scene):
    if scene.camera is None:
        Camera(scene)
    return scene


def test_vec_rgb_algebra():
    v = Vec3(1, 2, 3)
    assert [*(v * 2)] == [2, 4, 6] and [*(2 * v)] == [2, 4, 6] and [*(v * Scale(1, 2, 3))] == [1, 4, 9]
    assert v.dot(Vec3(1, 1, 1)) == 6 and [*v.cross(Vec3(0, 0, 1))] == [2, -1, 0]
    assert math.isclose(v.normalise().magnitude(), 1.0)
    f = Vec3(1, -1, 0).flip(Vec3(0, 1, 0))
    assert [round(c, 12) for c in f] == [1, 1, 0]
    with pytest.raises(TypeError):
        v * "x"
    assert [*(RGB(200, 200, 200) + RGB(100, 10, 0))] == [255, 210, 200]       # saturating add
    assert [*(RGB(10, 20, 30) * 0.5)] == [5, 10, 15]
    assert RGB(1, 2, 3).toTuple("bgr") == (3, 2, 1)
    assert [*RGB.mean([RGB(1, 2, 3), RGB(2, 2, 4)])] == [1, 2, 3]              # floor mean
    r = Ray(Vec3(0, 0, 0), Vec3(0, 0, 2), 1, None)
    assert [*r.direction] == [0, 0, 1] and r.entersAABB((Vec3(-1, -1, 3), Vec3(1, 1, 5)))


def test_photon_options_and_errors():
    from raytracing_amd.photon import NoBoundCameraError, Photon, Rendered
    opts = {"bounces": 3, "emissionSampleDensity": 1}
    p = Photon(cornell_scene(), opts)
    assert p.options is opts                                            # merged in place
    for key, val in {"bounces": 3, "samples": 1, "threads": 4, "width": 1440, "height": 900, "lights": True,
                     "bvh": True, "aabb": True, "transformBefore": True, "emissionSampleDensity": 1}.items():
        assert opts[key] == val
    assert [*opts["ambient"]] == [20, 20, 20]
    empty = Scene()
    Object(empty)
    with pytest.raises(NoBoundCameraError):
        Photon(empty, {"width": 8, "height": 8}).render()
    with pytest.raises(ValueError, match="Number of threads must be a factor of image height"):
        Photon(cornell_scene(), {"width": 8, "height": 10, "threads": 4}).render()
    pv = p.pixelToVector(0, 0)
    assert isinstance(pv, Vec3) and [*p.getRay(pv).start] == [0, 0, -11]
    assert p.getAttenuation(0) == 1 and 0 < p.getAttenuation(10) < 1
    lights = p.getAllLightsPos()
    assert len(lights) == 3 and [*lights[2][0]] == [0, 3, 1]
    img = np.zeros((2, 3, 3), np.uint8)
    img[..., 0] = 7                                                     # B channel
    r = Rendered(img, {}, 3, 2, time=0.1, rays=5, options={"ambient": RGB(1, 2, 3)}, engine="photon")
    assert r.imageAs("np")[0, 0].tolist() == [0, 0, 7] and r.imageAs("cv2")[0, 0].tolist() == [7, 0, 0]
    assert r.imageAs("listList")[0][0] == [0, 0, 7] and r.imageAs("tupleList")[0][0] == (0, 0, 7)
    back = Rendered.fromDict(json.loads(json.dumps(r.toDict())))
    assert np.array_equal(back.imageAs("np"), r.imageAs("np")) and back.rays == 5


def test_missing_library_fails_loudly(monkeypatch):
    from raytracing_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libt3d_hip.so")
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.load()


def test_make_opts_layout():
    o = make_opts(dict(width=5, height=7, bounces=2, samples=3, lights=False, ambient=(1, 2, 3)), seed=9, row_stride=4)
    assert (o.width, o.height, o.bounces, o.samples, o.lights, o.seed, o.row_stride) == (5, 7, 2, 3, 0, 9, 4)
    assert list(o.ambient) == [1.0, 2.0, 3.0]


def test_synthetic_flat_equals_flattened_object_graph():
    import dataclasses
    from raytracing_amd.workloads import synthetic_flat, synthetic_scene
    kw = dict(n_random_tris=60, grid=3, n_spheres=11, extent=12.0, camera=(0.3, 0.2, -28.8))
    a, b = flatten_scene(synthetic_scene(**kw)), synthetic_flat(**kw)
    for f in dataclasses.fields(a):
        x, y = np.asarray(getattr(a, f.name)), np.asarray(getattr(b, f.name))
        assert x.dtype == y.dtype and np.array_equal(x, y), f.name


def test_post_filters_match_reference_known_answers(kat):
    """Post.compress / decompress / upscale / denoise against outputs of the reference (tests/golden/make_kat.py)."""
    import T3DR
    k = kat["post"]
    img = k["image"]
    base = {"imgData": img, "width": 9, "height": 7, "rays": 5}
    c10 = T3DR.Post.compressRenderedDict(base, tolerance=10)
    assert c10["imgData"] == k["compressed_tol10"] and c10["rays"] == 5 and base["imgData"] is img
    assert T3DR.Post.compressRenderedDict(base, tolerance=0)["imgData"] == k["compressed_tol0"]
    assert T3DR.Post.decompressRenderedDict(c10)["imgData"] == k["decompressed_tol10"]
    up = T3DR.Post.upscale(np.array(img, np.uint8), 2)
    assert up.dtype == np.uint8 and up.tolist() == k["upscaled_x2"]
    den = T3DR.Post.denoise(img, 3)
    assert [[list(px) for px in row] for row in den] == k["denoised_k3"] and isinstance(den[0][0], tuple)
    # lossless round trip at tolerance 0
    assert T3DR.Post.decompressRenderedDict(T3DR.Post.compressRenderedDict(base, tolerance=0))["imgData"] == img


def test_rendered_json_round_trip(tmp_path):
    """Rendered.saveJSON / fromJSON (T3DR.py:59-118) and every imageAs form, without a device."""
    import T3DR
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (4, 6, 3)).astype(np.uint8)
    r = T3DR.Rendered(bgr, {}, 6, 4, time=0.25, rays=99, options={"bounces": 2, "ambient": RGB(1, 2, 3)}, engine="photon")
    assert np.array_equal(r.imageAs("np"), bgr[:, :, ::-1]) and np.array_equal(r.imageAs("cv2"), bgr)
    assert r.imageAs("tupleList")[1][2] == tuple(int(c) for c in bgr[1, 2, ::-1]) and r.imageAs("listList")[3][5] == bgr[3, 5, ::-1].tolist()
    path = tmp_path / "render.json"
    r.saveJSON(str(path))
    back = T3DR.Rendered.fromJSON(str(path))
    assert np.array_equal(back.imageAs("np"), r.imageAs("np")) and (back.width, back.height, back.rays, back.time) == (6, 4, 99, 0.25)
    assert back.engine == "photon" and [*back.options["ambient"]] == [1, 2, 3] and back.options["bounces"] == 2
    # the compressed form survives the same path
    packed = T3DR.Post.compressRenderedDict(r.toDict(), tolerance=0)
    assert np.array_equal(T3DR.Rendered.fromDict(T3DR.Post.decompressRenderedDict(packed)).imageAs("np"), r.imageAs("np"))


def test_flat_scene_digest_tracks_every_input():
    """Equal flattenings have equal digests; moving an object, a shader value or the camera changes it."""
    import T3DE
    from raytracing_amd.flatten import flatten_scene
    path = scene_path("cornellScene.json")
    a, b = T3DE.Scene.fromJSON(path), T3DE.Scene.fromJSON(path)
    base = flatten_scene(a).digest()
    assert flatten_scene(b).digest() == base and len(base) == 16
    mesh = next(o for o in b.objects if o.type == "mesh")
    mesh.position = T3DE.Vec3(mesh.position.x + 1e-6, mesh.position.y, mesh.position.z)
    assert flatten_scene(b).digest() != base
    c = T3DE.Scene.fromJSON(path)
    c.camera.length = c.camera.length * 2
    assert flatten_scene(c).digest() != base
    d = T3DE.Scene.fromJSON(path)
    next(o for o in d.objects if o.type == "mesh").shaders[0].roughness += 0.25
    assert flatten_scene(d).digest() != base


def test_stl_reader_ascii_binary_and_cut_off_files(tmp_path):
    """Binary files (also with a header that starts with the word "solid") and ASCII files give the same mesh; a file
    that is neither -- a binary file cut short -- is refused instead of read as an empty mesh."""
    import struct
    from raytracing_amd.scene import Mesh
    tris = [((0, 0, 0), (1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, 0, 0), (0, 0, 1), (1, 0, 0)),
            ((1, 0, 0), (0, 0, 1), (0, 1, 0))]
    ascii_file, binary_file, cut_file, odd_file = (tmp_path / n for n in ("a.stl", "b.stl", "c.stl", "d.stl"))
    lines = ["solid t"]
    for t in tris:
        lines += [" facet normal 0 0 0", "  outer loop"] + ["   vertex %g %g %g" % v for v in t] + ["  endloop", " endfacet"]
    ascii_file.write_text("\n".join(lines + ["endsolid t"]) + "\n")
    blob = b"solid is how this binary header starts".ljust(80, b" ") + struct.pack("<I", len(tris))
    for t in tris:
        blob += struct.pack("<3f", 0, 0, 0) + b"".join(struct.pack("<3f", *v) for v in t) + struct.pack("<H", 0)
    binary_file.write_bytes(blob)
    a, b = Mesh.fromSTL(str(ascii_file)), Mesh.fromSTL(str(binary_file))
    assert a.faces == b.faces and len(a.faces) == 4 and len(a.vertices) == 4
    assert [(v.x, v.y, v.z) for v in a.vertices] == [(v.x, v.y, v.z) for v in b.vertices]
    assert (a.vertices[1].x, a.vertices[1].y, a.vertices[1].z) == (1.0, 0.0, 0.0)
    assert (a.vertices[2].x, a.vertices[2].y, a.vertices[2].z) == (0.0, 0.0, 1.0)      # y and z exchanged
    cut_file.write_bytes(blob[:150])
    with pytest.raises(ValueError, match="not a complete binary STL"):
        Mesh.fromSTL(str(cut_file))
    odd_file.write_text("\n".join(lines[:-3]) + "\n")                                  # the last facet lost a vertex line
    with pytest.raises(ValueError, match="multiple of three"):
        Mesh.fromSTL(str(odd_file))


def test_flatten_cache_bakes_only_what_changed():
    """flatten_scene with a cache: an unchanged scene gives back the previous result; moved spheres / Lights / camera
    reuse the triangle arrays ("moved": the device copy can be updated in place); a moved mesh is baked again; a
    coordinate edited in place is seen by a thorough call; every result equals an uncached flattening."""
    from raytracing_amd.flatten import FlattenCache, flatten_scene
    from raytracing_amd.scene import Vec3
    from raytracing_amd.workloads import cornell_scene
    scene = cornell_scene(with_light=True, suzanne=True)
    names = ("tri_verts", "tri_shader", "light_pos", "light_strength", "light_color", "obj_center", "obj_radius", "obj_tri_begin",
             "obj_tri_count", "shader_color", "shader_rough", "obj_type")

    def same_as_uncached(flat):
        ref = flatten_scene(scene)
        return all(np.array_equal(getattr(flat, n), getattr(ref, n)) for n in names) and flat.cam_pos == ref.cam_pos

    cache = FlattenCache()
    a = flatten_scene(scene, cache, thorough=False)
    assert cache.change == "full" and same_as_uncached(a)
    assert flatten_scene(scene, cache, thorough=False) is a and cache.change == "same"
    assert flatten_scene(scene, cache, thorough=True) is a and cache.change == "same"       # the two kinds of call share the cache
    light = next(o for o in scene.objects if o.type == "light")
    light.position = Vec3(0.2, 2.9, 1.1)
    scene.camera.position = Vec3(0.0, 0.1, -10.0)
    b = flatten_scene(scene, cache, thorough=False)
    assert cache.change == "moved" and np.shares_memory(b.tri_verts, a.tri_verts) and same_as_uncached(b) and b.digest() != a.digest()
    monkey = next(o for o in scene.objects if o.type == "mesh" and len(o.mesh.faces) > 500)
    monkey.rotation.y += 0.3
    c = flatten_scene(scene, cache, thorough=False)
    assert cache.change == "full" and same_as_uncached(c) and not np.shares_memory(c.tri_verts, b.tri_verts)
    monkey.mesh.vertices[7].x += 0.125                       # not one of the sampled vertices
    assert flatten_scene(scene, cache, thorough=False) is c  # the quick look does not see it ...
    d = flatten_scene(scene, cache, thorough=True)           # ... the thorough one does
    assert cache.change == "full" and d is not c and same_as_uncached(d)
    assert flatten_scene(scene, cache, thorough=False) is d
    light.shaders[0].color.r = 13                            # a shader value: not a move
    e = flatten_scene(scene, cache, thorough=False)
    assert cache.change == "full" and same_as_uncached(e)
