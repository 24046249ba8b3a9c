"""Scenes and options of the benchmark configurations, built through the scene API.

The Cornell scene follows the reference's demo script (cornellTest.py:7-94): box mesh from
``cornellbox.stl`` rotated 90 degrees about y and baked, glass cube, mirror sphere, camera of
length 20 at (0, 0, -11).  The data files sit in ``raytracing_amd/data``.
"""
from __future__ import annotations

import math
import os

import numpy as np

from .scene import RGB, Camera, Cube, Light, Material, Mesh, Object, Scale, Scene, Shader, Sphere, Vec3

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _cornell_box(scene):
    box = Object(scene=scene, name="Box", mesh=Mesh.fromSTL(os.path.join(DATA_DIR, "cornellbox.stl")))
    box.rotation.y = math.radians(90)
    box.setTransforms()
    box.setShader(Shader(reflectivity=1, roughness=1, color=RGB(200, 200, 200), debugColor=RGB(100, 100, 100)))
    box.setShader(Shader(color=RGB(255, 255, 120), debugColor=RGB(120, 120, 20), emissionStrength=1), [48, 49])
    box.setShader(Shader(reflectivity=1, roughness=0.68, color=RGB(220, 40, 160), debugColor=RGB(220, 40, 160)), [28, 29])
    box.setShader(Shader(reflectivity=1, roughness=0.68, color=RGB(20, 220, 220), debugColor=RGB(20, 220, 220)), [27, 26])
    box.position = Vec3(0, 0, 1)
    return box


def _mirror_ball(scene):
    ball = Sphere(scene=scene, name="Mirror ball")
    ball.setShader(Shader(reflectivity=1, roughness=0, color=RGB(210, 210, 210), debugColor=RGB(180, 180, 180)))
    ball.position = Vec3(-1.54, -2.1, 2)
    ball.scale = Scale(3, 3, 3)
    return ball


def _camera(scene, position=(0, 0, -11), length=20):
    cam = Camera(scene=scene, name="Camera")
    cam.length = length
    cam.position = Vec3(*position)
    return cam


def cornell_scene(with_light=True, suzanne=False) -> Scene:
    """Cornell box + (glass cube | glass monkey) + mirror ball (+ one Light at (0, 3, 1))."""
    scene = Scene()
    _cornell_box(scene)
    if suzanne:
        monkey = Object(scene=scene, name="Monkey", mesh=Mesh.fromSTL(os.path.join(DATA_DIR, "suzanne.stl")))
        monkey.setShader(Shader(reflectivity=0.2, roughness=0.04, color=RGB(180, 180, 180), debugColor=RGB(180, 180, 180)))
        monkey.material = Material(ior=1.4)
        monkey.setTransforms()
        monkey.position = Vec3(1.44, -1.8, 1.4)
        monkey.scale = Scale(1.8, 1.8, 1.8)
    else:
        cube = Cube(scene=scene, name="cube")
        cube.setShader(Shader(reflectivity=0.15, roughness=0.5, color=RGB(255, 255, 255), debugColor=RGB(0, 0, 0)))
        cube.material = Material(ior=1.4)
        cube.position = Vec3(1.44, -0.9, 1.4)
        cube.scale = Scale(1.4, 2.6, 1.8)
        cube.rotation.y = math.radians(10)
    _mirror_ball(scene)
    if with_light:
        light = Light(scene=scene, name="Light0", position=Vec3(0, 3, 1), strength=1.0)
        light.setShader(Shader(color=RGB(200, 200, 200)))
    _camera(scene)
    return scene


_SYNTHETIC_SHADERS = [(0, 1, (220, 220, 220)), (0.5, 1, (200, 80, 80)), (1, 1, (80, 200, 80)), (1, 1, (80, 80, 200)),
                      (0, 0.5, (240, 240, 240)), (0.5, 0.5, (200, 200, 80)), (1, 0.5, (80, 200, 200)), (1, 1, (180, 180, 180))]
_SYNTHETIC_LIGHTS = [(0, 40, -60), (-40, 30, 0), (40, 30, 0), (0, 45, 40)]


def _synthetic_arrays(n_random_tris, grid, n_spheres, seed, E):
    """Geometry of the synthetic scene: triangles [T, 3, 3], shader pick per triangle, sphere centres and radii."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-E, E, (n_random_tris, 3))
    e1 = rng.uniform(-0.5, 0.5, (n_random_tris, 3))
    e2 = rng.uniform(-0.5, 0.5, (n_random_tris, 3))
    tri = np.stack([centres, centres + e1, centres + e2], axis=1)
    xs = np.linspace(-E, E, grid + 1)
    gx, gz = np.meshgrid(xs[:-1], xs[:-1], indexing="ij")
    step = xs[1] - xs[0]
    y = np.full(gx.size, -E)
    p00 = np.stack([gx.ravel(), y, gz.ravel()], 1)
    p10 = p00 + [step, 0, 0]
    p01 = p00 + [0, 0, step]
    p11 = p00 + [step, 0, step]
    ground = np.concatenate([np.stack([p00, p01, p10], 1), np.stack([p10, p01, p11], 1)])
    allv = np.concatenate([tri, ground])
    pick = rng.integers(0, 8, len(allv))
    sc = rng.uniform(-E, E, (n_spheres, 3))
    sr = rng.uniform(0.1, 0.6, n_spheres)
    return allv, pick, sc, sr


def synthetic_flat(n_random_tris=500_000, grid=500, n_spheres=10_000, seed=1234, extent=50.0, camera=None):
    """The flattened form of ``synthetic_scene(...)`` built straight from arrays: a million-triangle scene as
    a Python object graph takes minutes and gigabytes, the arrays take a second.  Same result as
    ``flatten_scene(synthetic_scene(...))`` (tests/test_scene_api.py checks it on a small instance)."""
    from .flatten import OBJ_LIGHT, OBJ_MESH, OBJ_OTHER, OBJ_SPHERE, FlatScene
    E = float(extent)
    allv, pick, sc, sr = _synthetic_arrays(n_random_tris, grid, n_spheres, seed, E)
    T, S, L = len(allv), n_spheres, len(_SYNTHETIC_LIGHTS)
    n = 1 + S + L + 1
    default = (1.0, 1.0, (200.0, 200.0, 200.0))  # Shader(): roughness, reflectivity, colour
    eight = [(float(r), float(f), tuple(float(c) for c in col)) for r, f, col in _SYNTHETIC_SHADERS]
    table = [default] + eight + [eight[i % 8] for i in range(S)] + [default] * (L + 1)
    obj_type = np.array([OBJ_MESH] + [OBJ_SPHERE] * S + [OBJ_LIGHT] * L + [OBJ_OTHER], np.int32)
    cam = tuple(float(c) for c in (camera if camera is not None else (0.0, 0.0, -2.4 * E)))
    lights = np.array(_SYNTHETIC_LIGHTS, np.float64) * E / 50.0
    center = np.concatenate([np.zeros((1, 3)), sc, lights, np.array([cam])])
    radius = np.concatenate([[0.5], (2 * sr) / 2, np.full(L + 1, 0.5)])
    tri_count = np.zeros(n, np.int32)
    tri_count[0] = T
    ior = np.ones(n)
    ior[0] = 1.3
    strength = np.zeros(n)
    strength[1 + S:1 + S + L] = 1.0
    shader0 = np.concatenate([[0], 9 + np.arange(S + L + 1)]).astype(np.int32)
    c = np.ascontiguousarray
    return FlatScene(
        obj_type=obj_type, obj_tri_begin=np.zeros(n, np.int32), obj_tri_count=tri_count, obj_center=c(center, np.float64),
        obj_radius=c(radius, np.float64), obj_ior=ior, obj_strength=strength, obj_shader0=shader0,
        tri_verts=c(allv, np.float64), tri_shader=(pick + 1).astype(np.int32),
        shader_color=np.array([t[2] for t in table], np.float64), shader_rough=np.array([t[0] for t in table], np.float64),
        shader_refl=np.array([t[1] for t in table], np.float64), shader_emis=np.zeros(len(table)),
        light_pos=c(lights, np.float64), light_strength=np.ones(L), light_color=np.full((L, 3), 200.0),
        cam_pos=cam, cam_euler=(0.0, 0.0, 0.0), cam_zlen=1.0, obj_ids=list(range(n)))


def synthetic_scene(n_random_tris=500_000, grid=500, n_spheres=10_000, seed=1234, extent=50.0, camera=None) -> Scene:
    """Large synthetic scene: random small triangles in [-extent, extent]^3, a ground grid of 2*grid^2
    triangles at y = -extent, random spheres, 8 shaders, 4 lights, camera at (0, 0, -2.4 extent) unless given
    (with the default camera many pixel rays meet the regular ground grid exactly on its edges)."""
    scene = Scene()
    E = float(extent)
    allv, pick, sc, sr = _synthetic_arrays(n_random_tris, grid, n_spheres, seed, E)
    allv = allv.reshape(-1, 3)
    mesh = Mesh([Vec3(*map(float, p)) for p in allv], [(3 * i, 3 * i + 1, 3 * i + 2) for i in range(len(allv) // 3)])
    obj = Object(scene=scene, name="Soup", mesh=mesh)
    shaders = [Shader(roughness=r, reflectivity=f, color=RGB(*c)) for (r, f, c) in _SYNTHETIC_SHADERS]
    for k, sh in enumerate(shaders):
        obj.setShader(sh, np.nonzero(pick == k)[0].tolist())
    obj.material = Material(ior=1.3)
    for i in range(n_spheres):
        s = Sphere(scene=scene, name=f"S{i}", position=Vec3(*map(float, sc[i])))
        s.scale = Scale(float(2 * sr[i]), float(2 * sr[i]), float(2 * sr[i]))
        s.setShader(shaders[i % 8])
    for i, pos in enumerate(_SYNTHETIC_LIGHTS):
        Light(scene=scene, name=f"L{i}", position=Vec3(*(float(q) * E / 50.0 for q in pos)), strength=1.0)
    _camera(scene, position=camera if camera is not None else (0, 0, -2.4 * E), length=20)
    return scene


def cornell2_multilight() -> Scene:
    """cornellScene2.json with three extra lights and a usable focal length."""
    scene = Scene.fromJSON(os.path.join(DATA_DIR, "cornellScene2.json"))
    scene.camera.length = 20
    for i, (pos, strength, col) in enumerate([((2.5, 2.0, -1.0), 0.8, (255, 200, 160)), ((-2.5, 1.0, 0.0), 0.6, (160, 200, 255)),
                                              ((0.0, -2.5, 3.0), 0.5, (200, 255, 200))]):
        light = Light(scene=scene, name=f"L{i}", position=Vec3(*pos), strength=strength)
        light.setShader(Shader(color=RGB(*col)))
    return scene


WORKLOADS = {
    # name: (scene factory, options)
    "cornell_160x120_b1": (lambda: Scene.fromJSON(os.path.join(DATA_DIR, "cornellScene.json")),
                           dict(width=160, height=120, bounces=1, samples=1, ambient=(20, 20, 20))),
    "cornell_1080p_b4": (lambda: cornell_scene(with_light=True),
                         dict(width=1920, height=1080, bounces=4, samples=1, ambient=(32, 32, 32))),
    "suzanne_4k_b4": (lambda: cornell_scene(with_light=True, suzanne=True),
                      dict(width=3840, height=2160, bounces=4, samples=1, ambient=(32, 32, 32))),
    "cornell2_8k_b8": (cornell2_multilight, dict(width=7680, height=4320, bounces=8, samples=1, ambient=(20, 20, 20))),
}


# Workloads too large for an object graph: name -> (flat-scene factory, options)
FLAT_WORKLOADS = {
    "synthetic_1m_4k_b2": (lambda: synthetic_flat(), dict(width=3840, height=2160, bounces=2, samples=1, ambient=(20, 20, 20))),
    "synthetic_200k_1080p_b2": (lambda: synthetic_flat(100_000, 224, 2000, extent=12.0, camera=(0.3713, 0.2117, -28.8)),
                                dict(width=1920, height=1080, bounces=2, samples=1, ambient=(20, 20, 20))),
}


def get_workload(name: str):
    factory, options = WORKLOADS[name]
    return factory(), dict(options)


def get_flat_workload(name: str):
    """-> (FlatScene, options) for any workload name."""
    if name in FLAT_WORKLOADS:
        factory, options = FLAT_WORKLOADS[name]
        return factory(), dict(options)
    from .flatten import flatten_scene
    scene, options = get_workload(name)
    return flatten_scene(scene), options
