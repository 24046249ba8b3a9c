"""Random scenes and options through both closest-hit paths of the HIP library against the CPU oracle.
usage: fuzz_parity.py [first_seed] [count] [big] [untimed] [deep]; prints one line per case and exits non-zero on the first
disagreement beyond the tolerances of tests/test_gpu_parity.py.  deep: 6 to 12 bounces instead of 0 to 5.  With
T3D_MEM_BUDGET_MB=1 in the environment the library cuts every frame into bands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def random_scene(seed, big=False, deep=False):
    from raytracing_amd.scene import RGB, Camera, Euler, Light, Material, Mesh, Object, Scale, Scene, Shader, Sphere, Vec3
    rng = np.random.default_rng(seed)
    s = Scene()
    cam = Camera(s, position=Vec3(*(float(v) for v in rng.uniform(-0.5, 0.5, 3) + [0, 0, -7])), length=float(rng.uniform(30, 70)))
    cam.rotation = Euler(*(float(v) for v in rng.uniform(-0.08, 0.08, 3)))

    def shader():
        glass = rng.random() < 0.35
        return Shader(color=RGB(*(int(v) for v in rng.integers(20, 256, 3))), roughness=float(rng.choice([0.0, 0.2, 0.7, 1.0])),
                      reflectivity=float(rng.choice([0.1, 0.5]) if glass else rng.choice([1.0, 1.0, 0.8, 0.0])),
                      emissionStrength=float(0.4 if rng.random() < 0.06 else 0.0))

    for m in range(int(rng.integers(1, 4))):
        n = int(rng.integers(200, 1500)) if big else int(rng.integers(1, 40))
        verts, faces = [], []
        centre = rng.uniform(-2.5, 2.5, 3) + [0, 0, 2]
        for k in range(n):
            c = centre + rng.uniform(-1.5, 1.5, 3)
            e1, e2 = rng.uniform(-1.2, 1.2, 3) * (0.25 if big else 1.0), rng.uniform(-1.2, 1.2, 3) * (0.25 if big else 1.0)
            for p in (c, c + e1, c + e2):
                verts.append(Vec3(*(float(v) for v in p)))
            faces.append((3 * k, 3 * k + 1, 3 * k + 2))
        o = Object(scene=s, name=f"mesh{m}", mesh=Mesh(verts, faces))
        shaders = [shader() for _ in range(int(rng.integers(1, 4)))]
        for k in range(n):
            o.setShader(shaders[k % len(shaders)], faces=[k])
        o.material = Material(ior=float(rng.choice([1.0, 1.3, 1.5])))
    # a floor and a back wall, so that most rays end somewhere
    room = Object(scene=s, name="room", mesh=Mesh([Vec3(-6, -3, -2), Vec3(6, -3, -2), Vec3(6, -3, 9), Vec3(-6, -3, 9),
                                                  Vec3(-6, 6, 9), Vec3(6, 6, 9)], [(0, 2, 1), (0, 3, 2), (3, 4, 5), (3, 5, 2), (0, 1, 2), (0, 2, 3), (3, 5, 4), (3, 2, 5)]))   # both windings
    room.setShader(Shader(color=RGB(200, 200, 190), roughness=1.0, reflectivity=1.0))
    for b in range(int(rng.integers(0, 5))):
        ball = Sphere(s, name=f"ball{b}", position=Vec3(*(float(v) for v in rng.uniform(-2.5, 2.5, 3) + [0, 0, 2.5])))
        d = float(rng.uniform(0.4, 2.2))
        ball.scale = Scale(d, d, d)
        ball.setShader(shader())
        ball.material = Material(ior=float(rng.choice([1.0, 1.33, 1.5])))
    for l in range(int(rng.integers(0, 4))):
        Light(s, name=f"light{l}", position=Vec3(*(float(v) for v in rng.uniform(-3, 3, 3) + [0, 3, 0])), strength=float(rng.uniform(0.3, 1.5)))
    opt = dict(width=int(rng.integers(8, 90)), height=int(rng.integers(6, 60)), bounces=int(rng.integers(6, 13)) if deep else int(rng.integers(0, 6)),
               samples=int(rng.integers(1, 4)), ambient=tuple(int(v) for v in rng.integers(0, 60, 3)), lights=bool(rng.random() < 0.85))
    return s, opt


def check_case(out, ref, frag, opt, n_lights=0):
    """The bars of tests/test_gpu_parity.py, with the allowances a random scene needs.  A secondary ray that grazes an
    edge may hit in float32 and miss in float64 (or the other way round): one node more or less, and the pixel it
    belongs to may move.  So one pixel (or 0.1 %) may differ, and the counts are compared to 0.1 %; primary hits away
    from edges must be identical."""
    from test_gpu_parity import EDGE_TIE_MARGIN, RADIANCE_TOL
    allowed = max(1, out["bgr"].shape[0] * out["bgr"].shape[1] // 1000)
    # pixels in whose trees some ray of some sample sits on a decision boundary as far as float32 can tell (a start point
    # in the plane of a face, a hit at distance zero) are reported by the oracle with a margin of zero and set aside
    sturdy = ref["fragility"] > 0
    diff = np.abs(out["bgr"].astype(int) - ref["bgr"].astype(int)).max(axis=2)
    assert ((diff > 1) & sturdy).sum() <= allowed, f"{((diff > 1) & sturdy).sum()} pixels differ by more than 1 LSB"
    err = np.abs(out["radiance"].astype(np.float64) - ref["radiance"]).max(axis=2)
    assert ((err > RADIANCE_TOL) & sturdy).sum() <= allowed, f"{((err > RADIANCE_TOL) & sturdy).sum()} pixels with a radiance error, max {err.max()}"
    mism = (out["prim"] != ref["prim"]).any(axis=2) & (frag > EDGE_TIE_MARGIN)
    assert mism.sum() == 0, f"{mism.sum()} primary-hit id mismatches away from triangle edges"
    for k in ("rays_api", "queries") if opt["lights"] else ("rays_api",):   # lights off: the reference traces on behind emitters
        # one node more or less is one shadow ray per light more or less, and its children
        assert abs(out["stats"][k] - ref["stats"][k]) <= max(2, ref["stats"][k] // 1000) * (n_lights + 2), (k, out["stats"][k], ref["stats"][k])
    # how far from the oracle the case was, inside the tolerances: reported so that drift shows from one run to the next
    return {"pixels": int(diff.size), "differ": int((diff > 0).sum()), "beyond_1_lsb": int((diff > 1).sum()), "max_radiance_error": float(err.max()),
            "ids_differ": int((out["prim"] != ref["prim"]).any(axis=2).sum()), "fragile": int((~sturdy).sum())}


def run_cases(first, count, report=print, big=False, untimed=False, drift=None, deep=False):
    """drift: a dict that collects, over all cases and both paths, how many pixels differed from the oracle at all, by more
    than 1 LSB, in their primary ids, and the largest radiance error."""
    from oracle import oracle
    from raytracing_amd import _native
    from raytracing_amd.flatten import flatten_scene
    oracle.build()
    ctxs = {a: _native.Context(0) for a in ("brute", "lbvh")}
    try:
        for a, c in ctxs.items():
            c.set_accel(a)
            if untimed:      # no per-kernel events: LBVH scenes run their shadow kernels on a stream of their own, overlapping the next level
                c.set_timing(False)
        for seed in range(first, first + count):
            scene, opt = random_scene(seed, big, deep)
            flat = flatten_scene(scene)
            ref = oracle.render(flat, opt, seed=seed, want_fragility=True)
            frag = oracle.primary_fragility(flat, opt)
            counts = {}
            for a, c in ctxs.items():
                c.upload(flat)
                out = c.render_rows(opt, seed=seed, want_radiance=True, want_primary=True)
                try:
                    d = check_case(out, ref, frag, opt, len(flat.light_strength))
                    counts["drift_" + a] = d
                    if drift is not None:
                        for k in ("pixels", "differ", "beyond_1_lsb", "ids_differ", "fragile"):
                            drift[k] = drift.get(k, 0) + d[k]
                        drift["max_radiance_error"] = max(drift.get("max_radiance_error", 0.0), d["max_radiance_error"])
                    # the two paths evaluate the interior test differently (two in-plane coordinates against three):
                    # they too may part ways at an edge, by as little
                    # any part of the frame -- a block of rows, every n-th row -- is the same pixels, exactly
                    prng = np.random.default_rng(seed + 77)
                    stride = int(prng.integers(1, 4))
                    row0 = int(prng.integers(0, opt["height"]))
                    nrows = int(prng.integers(1, (opt["height"] - row0 + stride - 1) // stride + 1))
                    part = c.render_rows(opt, seed=seed, row0=row0, nrows=nrows, row_stride=stride)
                    assert np.array_equal(part["bgr"], out["bgr"][row0::stride][:nrows]), ("rows", row0, nrows, stride)
                    # the same rows as the first of two frames of one call; the second frame is the call with the next seed
                    step = int(prng.integers(1, 1000))
                    pair = c.render_rows(opt, seed=seed, row0=row0, nrows=nrows, row_stride=stride, frames=2, frame_seed_step=step)
                    assert np.array_equal(pair["bgr"][0], part["bgr"]), ("frame 0 of 2", row0, nrows, stride)
                    other = c.render_rows(opt, seed=seed + step, row0=row0, nrows=nrows, row_stride=stride)
                    assert np.array_equal(pair["bgr"][1], other["bgr"]), ("frame 1 of 2", row0, nrows, stride, step)
                    # the trees of one pixel's samples, traced one by one, carry the colours the frame averaged
                    x, y = int(prng.integers(0, opt["width"])), int(prng.integers(0, opt["height"]))
                    S = opt["samples"] if opt["bounces"] > 1 else 1
                    roots = []
                    for smp in range(S):
                        nodes = c.trace_pixel(opt, x, y, smp, seed=seed)
                        assert not nodes or nodes[0].heap == 1
                        roots.append(list(nodes[0].color) if nodes else [float(v) for v in opt["ambient"]])
                    assert np.allclose(np.mean(roots, axis=0), out["radiance"][y, x], atol=2e-3), ("pixel", x, y, roots, out["radiance"][y, x])
                    # picking: closest hits of arbitrary rays (getFirstCollision) with the flag combinations its callers use
                    n_rays = 24
                    origins = prng.uniform(-2.5, 2.5, (n_rays, 3)) + [0, 0, 2]
                    dirs = prng.normal(size=(n_rays, 3))
                    lights_flag, cull_flag = bool(prng.integers(0, 2)), bool(prng.integers(0, 2))
                    got = c.closest_hits(origins, dirs, lights_flag, cull_flag)
                    bad = 0
                    for i in range(n_rays):
                        w = oracle.first_collision(flat, origins[i], dirs[i], lights_flag, cull_flag)
                        same = got["obj"][i] == w["obj"] and got["face"][i] == w["face"]
                        if same and w["obj"] >= 0:
                            assert abs(got["t"][i] - w["t"]) <= 2e-4 * max(1.0, w["t"]) and got["direction"][i] == w["direction"], ("ray", i, got["t"][i], w["t"])
                        bad += not same
                    assert bad <= 1, f"{bad} of {n_rays} closest hits differ"
                    counts[a] = out["stats"]["nodes"]
                    assert abs(counts[a] - counts["brute"]) <= max(2, counts[a] // 1000), ("the two paths disagree", counts)
                except AssertionError as e:
                    raise AssertionError(f"seed {seed} {a}: {opt} tris {flat.n_tris} spheres {flat.n_spheres} "
                                         f"lights {len(flat.light_strength)}: {str(e)[:300]}") from None
            report(f"seed {seed}: {opt['width']}x{opt['height']} b{opt['bounces']} s{opt['samples']} tris {flat.n_tris} spheres {flat.n_spheres} "
                   f"lights {len(flat.light_strength)} queries {ref['stats']['queries']} ok; pixels that differ (brute / lbvh) "
                   f"{counts['drift_brute']['differ']} / {counts['drift_lbvh']['differ']}, by more than 1 LSB "
                   f"{counts['drift_brute']['beyond_1_lsb']} / {counts['drift_lbvh']['beyond_1_lsb']}, max radiance error "
                   f"{max(counts['drift_brute']['max_radiance_error'], counts['drift_lbvh']['max_radiance_error']):.4f}")
    finally:
        for c in ctxs.values():
            c.close()


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    big = "big" in sys.argv[3:]              # meshes of hundreds of small triangles: deep hierarchies
    untimed = "untimed" in sys.argv[3:]      # per-kernel timing off
    deep = "deep" in sys.argv[3:]            # 6 to 12 bounces
    drift = {}
    try:
        run_cases(first, count, report=lambda line: print(line, flush=True), big=big, untimed=untimed, drift=drift, deep=deep)
    except AssertionError as e:
        print("FAILED:", e)
        sys.exit(1)
    print("drift over all cases and both paths:", drift)


if __name__ == "__main__":
    main()
