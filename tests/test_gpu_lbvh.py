"""The LBVH path (large scenes; tree built on the device, or on the host for small ones) against the CPU oracle and against the brute-force path.

The hierarchy is a conservative cull, as the reference's own BVH is (T3DR.py:553-569, 659-707): the closest
hit it finds must be the brute-force closest hit, so every tolerance of test_gpu_parity.py applies unchanged.
"""
import numpy as np
import pytest

from conftest import flat_for, golden_names, load_golden
from test_gpu_parity import EDGE_TIE_MARGIN, LSB_OK_FRACTION, check_against_oracle, check_growing_trees

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lbvh_ctx():
    from raytracing_amd import _native
    ctx = _native.Context(0)
    ctx.set_accel("lbvh")
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", golden_names())
def test_golden_cases_through_lbvh(name, lbvh_ctx, oracle_mod):
    g, meta = load_golden(name)
    flat = flat_for(meta["scene"])
    opt, seed = meta["options"], meta["seed"]
    lbvh_ctx.upload(flat)
    out = lbvh_ctx.render_rows(opt, seed=seed, want_radiance=True, want_primary=True)
    ref = oracle_mod.render(flat, opt, seed=seed, rng_mode=0)
    frag = oracle_mod.primary_fragility(flat, opt)
    check_against_oracle(out, ref, frag, lights=opt.get("lights", True))
    st = out["stats"]
    # the cull must have culled: fewer triangle tests than brute force, and boxes were tested
    assert st["aabb_tests"] > 0 and st["tri_tests"] < st["queries"] * flat.n_tris
    if meta["rng"] == "keyed":
        diff = np.abs(out["bgr"][:, :, ::-1].astype(int) - g["image_rgb"].astype(int)).max(axis=2)
        assert (diff <= 1).mean() >= LSB_OK_FRACTION
        assert st["rays_api"] == meta["rays"]
        ids_equal = (out["prim"][..., 0] == g["prim_obj"]) & (out["prim"][..., 1] == g["prim_face"])
        assert (~ids_equal & (frag > EDGE_TIE_MARGIN)).sum() == 0


def test_maximum_bounces_and_growing_trees_through_lbvh(lbvh_ctx, oracle_mod):
    check_growing_trees(lbvh_ctx, oracle_mod)


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_coincident_primitives_keep_the_reference_order(accel, oracle_mod):
    """Two hundred copies of one triangle, each with corners and a shader of its own, and three spheres in one place:
    every copy is hit at the same distance and the first one in face order wins (T3DR.py:715), whatever order
    the hierarchy (whose sort keys all collide) or the plane groups visit them in."""
    from raytracing_amd import _native
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.scene import RGB, Camera, Light, Mesh, Object, Scale, Scene, Shader, Sphere, Vec3
    s = Scene()
    Camera(s, position=Vec3(0.11, 0.07, -6), length=20)
    verts, faces = [], []
    for k in range(200):
        verts += [Vec3(-2, -2, 2), Vec3(2, 2, 2), Vec3(2, -2, 2)]
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    verts += [Vec3(-2, -2, 2.5), Vec3(-2, 2, 2.5), Vec3(2, 2, 2.5)]      # another triangle behind, seen beside the stack
    faces.append((600, 601, 602))
    stack = Object(scene=s, name="stack", mesh=Mesh(verts, faces))
    for k in range(201):
        stack.setShader(Shader(name=f"s{k}", color=RGB(10 + k, 250 - k, 30 + (k * 7) % 200), roughness=0.8, reflectivity=0.9), faces=[k])
    for k in range(3):
        ball = Sphere(s, name=f"ball{k}", position=Vec3(-1.2, 1.2, 1.0))
        ball.scale = Scale(1, 1, 1)
        ball.setShader(Shader(color=RGB(250 - 100 * k, 40 + 100 * k, 40), roughness=0.5, reflectivity=0.8))
    Light(s, position=Vec3(0, 1, -3), strength=1.0)
    flat = flatten_scene(s)
    ctx = _native.Context(0)
    try:
        ctx.set_accel(accel)
        ctx.upload(flat)
        opt = dict(width=48, height=36, bounces=2, samples=1)
        out = ctx.render_rows(opt, seed=6, want_radiance=True, want_primary=True)
    finally:
        ctx.close()
    ref = oracle_mod.render(flat, opt, seed=6)
    check_against_oracle(out, ref, oracle_mod.primary_fragility(flat, opt))
    seen = out["prim"][..., 1][out["prim"][..., 0] == 1]          # faces of the stack object seen by primary rays
    assert set(np.unique(seen)) == {0, 200}                         # the first copy and the triangle behind, no other copy
    balls = np.unique(out["prim"][..., 0][out["prim"][..., 1] == -1])
    assert [int(b) for b in balls if 2 <= b <= 4] == [2]           # the first of the three spheres (object 5 is the Light)


def test_tiny_scenes_through_lbvh(lbvh_ctx, oracle_mod):
    """No primitive at all, one, two, three: hierarchies of zero, zero, one and two inner nodes."""
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.scene import RGB, Camera, Light, Mesh, Object, Scale, Scene, Shader, Sphere, Vec3
    s = Scene()
    Camera(s, position=Vec3(0.13, 0.05, -5), length=20)
    opt = dict(width=21, height=13, bounces=2, samples=2, ambient=(9, 8, 7))

    def check():
        flat = flatten_scene(s)
        lbvh_ctx.upload(flat)
        out = lbvh_ctx.render_rows(opt, seed=4, want_radiance=True, want_primary=True)
        check_against_oracle(out, oracle_mod.render(flat, opt, seed=4), oracle_mod.primary_fragility(flat, opt))
        return out

    out = check()
    assert (out["bgr"] == np.array([7, 8, 9], np.uint8)).all()
    ball = Sphere(s, position=Vec3(0, 0, 3))
    ball.scale = Scale(2, 2, 2)
    ball.setShader(Shader(color=RGB(250, 10, 10), roughness=0.3, reflectivity=0.6))
    assert (check()["prim"][..., 0] == 1).any()
    Light(s, position=Vec3(2, 3, -1), strength=1.5)                  # a Light is a primitive too
    check()
    tri = Object(scene=s, name="tri", mesh=Mesh([Vec3(-3, -1.5, 0), Vec3(3, -1.5, 0), Vec3(0, -1.5, 8)], [(0, 2, 1)]))
    tri.setShader(Shader(color=RGB(90, 200, 90), roughness=0.9, reflectivity=0.5))
    out = check()
    assert (out["prim"][..., 0] == 3).any() and out["stats"]["aabb_tests"] > 0


def _soup(n_random, grid, n_spheres):
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.workloads import synthetic_scene
    # camera off the grid's symmetry planes: with the default one, whole columns of pixel rays meet ground-grid
    # edges exactly, where float32 and float64 may legitimately decide the strict interior test differently
    return flatten_scene(synthetic_scene(n_random_tris=n_random, grid=grid, n_spheres=n_spheres, seed=1234, extent=12.0,
                                         camera=(0.3713, 0.2117, -28.8)))


def test_synthetic_scene_matches_oracle(gpu_ctx, oracle_mod):
    """3 800 triangles + 150 spheres, 4 lights: the automatic choice is the LBVH."""
    flat = _soup(3000, 20, 150)
    assert flat.n_tris == 3800
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    opt = dict(width=96, height=54, bounces=2, samples=2)
    out = gpu_ctx.render_rows(opt, seed=7, want_radiance=True, want_primary=True)
    assert out["stats"]["aabb_tests"] > 0
    ref = oracle_mod.render(flat, opt, seed=7)
    frag = oracle_mod.primary_fragility(flat, opt)
    check_against_oracle(out, ref, frag)


def test_lbvh_agrees_with_brute_force_on_the_gpu(gpu_ctx, lbvh_ctx):
    """20 000 triangles + 500 spheres at 320x180: the two device paths give the same picture and counts."""
    flat = _soup(15000, 50, 500)
    opt = dict(width=320, height=180, bounces=2, samples=1)
    gpu_ctx.set_accel("brute")
    gpu_ctx.upload(flat)
    brute = gpu_ctx.render_rows(opt, seed=3, want_primary=True)
    gpu_ctx.set_accel("auto")
    lbvh_ctx.upload(flat)
    fast = lbvh_ctx.render_rows(opt, seed=3, want_primary=True)
    diff = np.abs(brute["bgr"].astype(int) - fast["bgr"].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= LSB_OK_FRACTION
    same_ids = (brute["prim"] == fast["prim"]).all(axis=2)
    assert same_ids.mean() >= 0.9995            # the two paths round the interior test differently on edges
    assert abs(brute["stats"]["queries"] - fast["stats"]["queries"]) <= 1e-3 * brute["stats"]["queries"]
    assert fast["stats"]["tri_tests"] < 0.05 * brute["stats"]["tri_tests"]


def test_host_built_and_device_built_trees_give_the_same_picture(lbvh_ctx, monkeypatch):
    """Scenes of up to some thousand primitives get a hierarchy built on the host by surface area, larger ones the
    Morton-order one built on the device (t3d_set_host_tree_limit moves the line).  Either is only a cull: pixels, ids and
    distances must be the same bit for bit; the number of boxes tested is what differs."""
    flat = _soup(3000, 20, 150)
    opt = dict(width=160, height=90, bounces=2, samples=1)
    from raytracing_amd import _native
    ctx = _native.Context(0)
    ctx.set_accel("lbvh")
    ctx.set_host_tree_limit(0)
    ctx.upload(flat)
    dev = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
    ctx.set_host_tree_limit(1_000_000)
    ctx.upload(flat)
    host = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
    monkeypatch.setenv("T3D_HOST_TREE_MAX", "0")          # the environment wins over the call
    ctx.upload(flat)
    env = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
    # (box counts of one tree vary by a fraction of a percent from run to run: which rays share a wave decides how stale the
    # best distance is that a postponed leaf is tested against)
    assert abs(env["stats"]["aabb_tests"] - dev["stats"]["aabb_tests"]) < 0.005 * dev["stats"]["aabb_tests"]
    monkeypatch.delenv("T3D_HOST_TREE_MAX")
    with pytest.raises(_native.NativeError, match="negative"):
        ctx.set_host_tree_limit(-1)
    ctx.close()
    for key in ("bgr", "radiance", "prim", "prim_t"):
        assert np.array_equal(dev[key], host[key], equal_nan=dev[key].dtype.kind == "f"), key   # prim_t: NaN where nothing is hit
    assert dev["stats"]["queries"] == host["stats"]["queries"] and dev["stats"]["shadow_queries"] == host["stats"]["shadow_queries"]
    assert host["stats"]["aabb_tests"] < 0.97 * dev["stats"]["aabb_tests"]    # two different trees were traversed, the host's the better one
    # spheres and the camera moved: the refit keeps whichever tree was built
    from dataclasses import replace
    from raytracing_amd.flatten import OBJ_SPHERE
    centre = flat.obj_center.copy()
    centre[flat.obj_type == OBJ_SPHERE] += np.array([0.05, -0.02, 0.03])
    moved = replace(flat, obj_center=centre, signature_digest=None)
    lbvh_ctx.upload(flat)
    lbvh_ctx.update(moved)
    refit = lbvh_ctx.render_rows(opt, seed=9, want_primary=True)
    lbvh_ctx.upload(moved)
    rebuilt = lbvh_ctx.render_rows(opt, seed=9, want_primary=True)
    assert np.array_equal(refit["bgr"], rebuilt["bgr"]) and np.array_equal(refit["prim"], rebuilt["prim"])


def test_background_tree_is_adopted_between_two_calls():
    """t3d_set_background_tree: a scene whose hierarchy the device builds is traced through that tree at once while a host
    thread cuts the surface-area tree; the first call that finds it finished puts it in place.  Same buffers, same
    pictures bit for bit, fewer boxes tested -- through plain calls, a launch graph replayed across the swap, a moved
    sphere (refit of the adopted tree) and picking; uploads and the context's end wait for a build that still runs."""
    import torch
    from dataclasses import replace
    from raytracing_amd import _native
    from raytracing_amd.flatten import OBJ_SPHERE
    flat = _soup(15000, 50, 500)
    opt = dict(width=160, height=90, bounces=2, samples=1)
    ctx = _native.Context(0)
    try:
        ctx.set_host_tree_limit(0)               # the device builds
        ctx.upload(flat)
        assert ctx.tree_status() == "device"
        dev = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
        ctx.set_host_tree_limit(1_000_000)       # the host builds, at upload
        ctx.upload(flat)
        assert ctx.tree_status() == "host"
        host = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
        ctx.set_host_tree_limit(0)
        ctx.set_background_tree(True)
        ctx.upload(flat)
        assert ctx.tree_status() in ("building", "background")     # (a call that finds the build finished adopts it)
        # a band asked for again and again becomes a launch graph, which must survive the swap: same addresses
        out = torch.empty((opt["height"], opt["width"], 3), dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.Stream("cuda:0")
        for _ in range(4):
            ctx.render_rows_device(opt, 9, 0, opt["height"], out.data_ptr(), stream=stream.cuda_stream)
        assert ctx.tree_status(wait=True) == "background"
        for _ in range(2):
            ctx.render_rows_device(opt, 9, 0, opt["height"], out.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        ctx.last_call_stats()
        assert np.array_equal(out.cpu().numpy(), dev["bgr"])
        bg = ctx.render_rows(opt, seed=9, want_radiance=True, want_primary=True)
        for key in ("bgr", "radiance", "prim", "prim_t"):
            assert np.array_equal(dev[key], bg[key], equal_nan=dev[key].dtype.kind == "f"), key
            assert np.array_equal(host[key], bg[key], equal_nan=dev[key].dtype.kind == "f"), key
        assert bg["stats"]["queries"] == dev["stats"]["queries"]
        assert bg["stats"]["aabb_tests"] < 0.97 * dev["stats"]["aabb_tests"]
        assert abs(bg["stats"]["aabb_tests"] - host["stats"]["aabb_tests"]) < 0.005 * host["stats"]["aabb_tests"]   # the same tree
        # spheres moved: the adopted tree is refitted like any other
        centre = flat.obj_center.copy()
        centre[flat.obj_type == OBJ_SPHERE] += np.array([0.05, -0.02, 0.03])
        moved = replace(flat, obj_center=centre, signature_digest=None)
        ctx.update(moved)
        assert ctx.tree_status() == "background"
        refit = ctx.render_rows(opt, seed=9, want_primary=True)
        # an upload while a build runs waits for it and starts the next; so does the context's end; a moved scene before the
        # adoption is adopted with its spheres where they are now; picking adopts as well
        ctx.upload(flat)
        ctx.upload(moved)
        assert ctx.tree_status() in ("building", "background")
        ctx.update(flat)
        ctx.update(moved)
        rng = np.random.default_rng(5)
        origins = np.tile(np.array([0.3713, 0.2117, -28.8]), (64, 1))
        dirs = rng.normal(size=(64, 3)) * 0.2 + np.array([0.0, 0.0, 1.0])
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        early = ctx.closest_hits(origins, dirs, lights=True, back_cull=True)
        assert ctx.tree_status(wait=True) == "background"
        late = ctx.closest_hits(origins, dirs, lights=True, back_cull=True)
        for key in early:
            assert np.array_equal(early[key], late[key], equal_nan=True), key
        rebuilt = ctx.render_rows(opt, seed=9, want_primary=True)
        assert np.array_equal(refit["bgr"], rebuilt["bgr"]) and np.array_equal(refit["prim"], rebuilt["prim"])
        # a scene the host builds at upload, or one without a hierarchy, starts no thread
        ctx.set_host_tree_limit(1_000_000)
        ctx.upload(flat)
        assert ctx.tree_status() == "host"
        ctx.set_host_tree_limit(0)
        ctx.set_accel("brute")
        ctx.upload(_soup(20, 2, 3))
        assert ctx.tree_status(wait=True) == "none"
        ctx.set_accel("auto")
        ctx.upload(flat)                          # ... and the context ends while this build runs
    finally:
        ctx.close()


def test_too_deep_a_host_tree_is_not_used(lbvh_ctx, monkeypatch):
    """Triangles whose sizes fall off geometrically over 27 decades, 8 coincident copies of each: the surface-area build
    peels the large ones off a few per level and ends 52 levels deep, more than the traversal stack is sized for; the
    library notices and takes the Morton-order tree (same box count as with the host build switched off)."""
    from dataclasses import replace
    from raytracing_amd.workloads import synthetic_flat
    n = 8000
    base = synthetic_flat(n_random_tris=4, grid=1, n_spheres=2, seed=1, extent=12.0, camera=(0.3713, 0.2117, -28.8))
    s = 10.0 * (15.0 / 16.0) ** (np.arange(n) % 1000)
    c = np.stack([s, s, s], 1)
    tri = np.empty((n, 3, 3))
    tri[:, 0] = c + np.stack([-0.4 * s, -0.4 * s, 0 * s], 1)
    tri[:, 1] = c + np.stack([0 * s, 0.4 * s, 0.1 * s], 1)
    tri[:, 2] = c + np.stack([0.4 * s, -0.4 * s, -0.1 * s], 1)
    count = base.obj_tri_count.copy()
    count[0] = n
    flat = replace(base, tri_verts=np.ascontiguousarray(tri), tri_shader=np.full(n, 3, np.int32), obj_tri_count=count)
    opt = dict(width=96, height=54, bounces=1, samples=1)
    monkeypatch.setenv("T3D_HOST_TREE_MAX", "0")
    lbvh_ctx.upload(flat)
    dev = lbvh_ctx.render_rows(opt, seed=2, want_primary=True)
    monkeypatch.delenv("T3D_HOST_TREE_MAX")
    lbvh_ctx.upload(flat)
    auto = lbvh_ctx.render_rows(opt, seed=2, want_primary=True)
    assert np.array_equal(dev["bgr"], auto["bgr"]) and np.array_equal(dev["prim"], auto["prim"])
    assert abs(dev["stats"]["aabb_tests"] - auto["stats"]["aabb_tests"]) < 0.005 * dev["stats"]["aabb_tests"]


def test_closest_hits_through_lbvh(lbvh_ctx, oracle_mod):
    flat = flat_for("config2_suzanne.json")
    lbvh_ctx.upload(flat)
    rng = np.random.default_rng(11)
    n = 300
    origins = np.concatenate([np.tile([0.0, 0.0, -11.0], (n // 2, 1)), rng.uniform(-2.5, 2.5, (n // 2, 3)) + [0, 0, 1]])
    dirs = np.concatenate([np.stack([rng.uniform(-0.5, 0.5, n // 2), rng.uniform(-0.4, 0.4, n // 2), np.ones(n // 2)], 1),
                           rng.normal(size=(n // 2, 3))])
    for lights, cull in ((True, True), (True, False)):
        got = lbvh_ctx.closest_hits(origins, dirs, lights, cull)
        bad = 0
        for i in range(n):
            w = oracle_mod.first_collision(flat, origins[i], dirs[i], lights, cull)
            same = got["obj"][i] == w["obj"] and got["face"][i] == w["face"]
            if same and w["obj"] >= 0:
                assert abs(got["t"][i] - w["t"]) <= 2e-5 * max(1.0, w["t"])
            bad += not same
        assert bad <= 2, f"{bad} of {n} closest hits differ (lights={lights}, cull={cull})"


def test_large_scene_properties(gpu_ctx, oracle_mod):
    """200 000 triangles + 2 000 spheres, 960x540: determinism, partition independence, oracle on every 20th row."""
    flat = _soup(100_000, 224, 2000)
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    opt = dict(width=960, height=540, bounces=2, samples=1)
    H = opt["height"]
    full = gpu_ctx.render_rows(opt, seed=0, want_primary=True)
    again = gpu_ctx.render_rows(opt, seed=0)
    assert np.array_equal(full["bgr"], again["bgr"])
    frame = np.empty_like(full["bgr"])
    rays = 0
    for r in range(4):
        o = gpu_ctx.render_rows(opt, seed=0, row0=r, nrows=H // 4, row_stride=4)
        frame[r::4] = o["bgr"]
        rays += o["stats"]["rays_api"]
    assert np.array_equal(frame, full["bgr"]) and rays == full["stats"]["rays_api"]
    step = 20   # 27 rows, spread over the host's threads
    from conftest import oracle_rows_threaded
    ref, frag = oracle_rows_threaded(oracle_mod, flat, opt, 0, range(0, H, step))
    sub = dict(bgr=full["bgr"][::step], prim=full["prim"][::step], prim_t=full["prim_t"][::step], radiance=None, stats=ref["stats"])
    check_against_oracle(sub, ref, frag)


def _frame_properties(ctx, oracle_mod, flat, opt, row_step, parts=8):
    """Determinism, independence of the row partition, and the oracle on every row_step-th row."""
    H = opt["height"]
    full = ctx.render_rows(opt, seed=0, want_primary=True)
    again = ctx.render_rows(opt, seed=0)
    assert np.array_equal(full["bgr"], again["bgr"]) and full["stats"]["queries"] == again["stats"]["queries"]
    frame = np.empty_like(full["bgr"])
    rays = 0
    for r in range(parts):
        o = ctx.render_rows(opt, seed=0, row0=r, nrows=H // parts, row_stride=parts)
        frame[r::parts] = o["bgr"]
        rays += o["stats"]["rays_api"]
    assert np.array_equal(frame, full["bgr"]) and rays == full["stats"]["rays_api"]
    from conftest import oracle_rows_threaded
    ref, frag = oracle_rows_threaded(oracle_mod, flat, opt, 0, range(0, H, row_step))   # the rows spread over the host's threads
    sub = dict(bgr=full["bgr"][::row_step], prim=full["prim"][::row_step], prim_t=full["prim_t"][::row_step], radiance=None,
               stats=ref["stats"])
    check_against_oracle(sub, ref, frag)
    return full


def test_suzanne_4k_full_size(gpu_ctx, oracle_mod):
    """suzanne.stl inside the box, 3840x2160, 4 bounces (1 018 triangles: LBVH)."""
    from raytracing_amd.workloads import get_flat_workload
    flat, opt = get_flat_workload("suzanne_4k_b4")
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    full = _frame_properties(gpu_ctx, oracle_mod, flat, opt, row_step=8)    # 270 rows against the oracle
    assert full["stats"]["aabb_tests"] > 0


def test_synthetic_1m_4k_full_size(gpu_ctx, oracle_mod):
    """1 000 000 triangles + 10 000 spheres, 4 lights, 3840x2160, 2 bounces (the large-scene benchmark workload): a
    20-level hierarchy over quantised records with colliding sort keys.  Determinism, independence of the row
    partition, and the oracle on sixteen rows spread over the frame (each a full closest-hit search through the
    reference's own per-mesh hierarchy in float64)."""
    from conftest import oracle_rows_threaded
    from raytracing_amd.workloads import get_flat_workload
    flat, opt = get_flat_workload("synthetic_1m_4k_b2")
    assert flat.n_tris == 1_000_000 and flat.n_spheres == 10_004
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    H = opt["height"]
    full = gpu_ctx.render_rows(opt, seed=0, want_primary=True)
    st = full["stats"]
    assert st["aabb_tests"] > 0 and st["bvh_deep_pushes"] == 0 and 0 < st["tri_tests"] < 1e-3 * st["queries"] * flat.n_tris
    again = gpu_ctx.render_rows(opt, seed=0)
    assert np.array_equal(full["bgr"], again["bgr"]) and st["queries"] == again["stats"]["queries"]
    frame = np.empty_like(full["bgr"])
    rays = 0
    for r in range(4):
        o = gpu_ctx.render_rows(opt, seed=0, row0=r, nrows=H // 4, row_stride=4)
        frame[r::4] = o["bgr"]
        rays += o["stats"]["rays_api"]
    assert np.array_equal(frame, full["bgr"]) and rays == st["rays_api"]
    rows = [7, 151, 301, 455, 611, 760, 905, 1051, 1203, 1360, 1511, 1655, 1799, 1970, 2152, 2159]
    ref, frag = oracle_rows_threaded(oracle_mod, flat, opt, 0, rows, threads=16)
    sub = dict(bgr=full["bgr"][rows], prim=full["prim"][rows], prim_t=full["prim_t"][rows], radiance=None, stats=ref["stats"])
    # the counters of the sub-frame are the oracle's own: what is compared is pixels, ids and distances
    check_against_oracle(sub, ref, frag)


def test_cornell2_8k_full_size(gpu_ctx, oracle_mod):
    """cornellScene2.json with three added lights, 7680x4320, 8 bounces: the frame is split into bands."""
    from raytracing_amd.workloads import get_flat_workload
    flat, opt = get_flat_workload("cornell2_8k_b8")
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    _frame_properties(gpu_ctx, oracle_mod, flat, opt, row_step=24)     # 180 rows against the oracle


def test_deep_traversal_stack(gpu_ctx, oracle_mod):
    """200 000 parallel sheets, every one of them in the way of every camera ray: on its first descent a ray
    postpones three siblings per level of the hierarchy, more than the part of the stack that lives in LDS, so the
    entries kept in private memory are used (and counted)."""
    from dataclasses import replace
    from raytracing_amd.workloads import synthetic_flat
    n = 200_000
    base = synthetic_flat(n_random_tris=4, grid=1, n_spheres=2, seed=1, extent=12.0, camera=(0.3713, 0.2117, -28.8))
    z = np.linspace(-10.0, 10.0, n)
    tri = np.empty((n, 3, 3))
    tri[:, 0] = np.stack([np.full(n, -30.0), np.full(n, -30.0), z], 1)
    tri[:, 1] = np.stack([np.zeros(n), np.full(n, 40.0), z], 1)          # wound to face the camera (normal along -z)
    tri[:, 2] = np.stack([np.full(n, 30.0), np.full(n, -30.0), z], 1)
    count = base.obj_tri_count.copy()
    count[0] = n
    flat = replace(base, tri_verts=np.ascontiguousarray(tri), tri_shader=np.full(n, 3, np.int32), obj_tri_count=count)
    gpu_ctx.set_accel("auto")
    gpu_ctx.upload(flat)
    opt = dict(width=48, height=27, bounces=2, samples=1)
    out = gpu_ctx.render_rows(opt, seed=5, want_radiance=True, want_primary=True)
    assert out["stats"]["bvh_deep_pushes"] > 0
    face = out["prim"][..., 1]
    assert ((face == 0) | (face == -1)).all() and (face == 0).mean() > 0.5     # the first sheet or nothing
    ref = oracle_mod.render(flat, opt, seed=5)
    frag = oracle_mod.primary_fragility(flat, opt)
    check_against_oracle(out, ref, frag)
    # single-ray entry point through the same stack
    hit = gpu_ctx.closest_hits([[0.1, 0.2, -28.8]], [[0.0, 0.0, 1.0]], True, True)
    assert hit["face"][0] == 0 and abs(hit["t"][0] - 18.8) < 1e-4


def test_scene_without_lights_through_lbvh(lbvh_ctx, oracle_mod):
    """No Light object and no emissive face: no shadow rays at all (the shadow kernel is not launched)."""
    import os
    from conftest import scene_path
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.scene import Scene
    scene = Scene.fromJSON(scene_path("config2_suzanne.json"))
    scene.objects = [o for o in scene.objects if o.type != "light"]
    for o in scene.objects:
        for sh in getattr(o, "shaders", None) or []:
            sh.emissionStrength = 0
    flat = flatten_scene(scene)
    assert len(flat.light_strength) == 0
    lbvh_ctx.upload(flat)
    opt = dict(width=96, height=54, bounces=3, samples=2)
    out = lbvh_ctx.render_rows(opt, seed=4, want_radiance=True, want_primary=True)
    assert out["stats"]["shadow_queries"] == 0 and out["stats"]["aabb_tests"] > 0
    ref = oracle_mod.render(flat, opt, seed=4)
    frag = oracle_mod.primary_fragility(flat, opt)
    check_against_oracle(out, ref, frag)
