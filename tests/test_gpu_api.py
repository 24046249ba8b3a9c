"""Behaviour of the C ABI around the tracing itself: asynchronous calls that overflow, cancellation, the frame on
several contexts at once, the asynchronous host render, and what ``Photon.render`` builds from them."""
import numpy as np
import pytest

from conftest import flat_for, scene_path
from test_gpu_parity import nested_glass_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", ["brute", "lbvh"])
def test_async_overflow_is_reported_once_and_buffers_grow(monkeypatch, path):
    """An asynchronous call (no stats) whose trace trees outgrow the work buffers cannot say so itself.  The overflow
    stays on record -- also when the frame was cut into several bands, whose counters replace each other -- until the
    next look: t3d_last_call_stats, or the next render call, returns T3D_E_CAPACITY once, and asynchronous calls start
    with larger buffers from then on.  On both closest-hit paths."""
    from raytracing_amd import _native
    flat = nested_glass_scene()
    opt = dict(width=24, height=16, bounces=9, samples=2)
    monkeypatch.setenv("T3D_MEM_BUDGET_MB", "1")       # several bands per frame

    def context():
        c = _native.Context(0)
        c.set_accel("lbvh" if path == "lbvh" else "brute")
        c.upload(flat)
        return c

    ctx = context()
    try:
        want = ctx.render_rows(opt, seed=5)                      # checked call: grows per band as needed
        assert want["stats"]["launches"] > 30
        fresh = context()                                        # a context whose buffers have not grown yet
        try:
            img = _native.pinned_image((16, 24, 3))
            seen = 0
            for attempt in range(12):
                fresh.render_rows_async(opt, 5, 0, 16, img)
                try:
                    fresh.last_call_stats()
                    break
                except _native.NativeError as e:
                    assert e.code == _native.E_CAPACITY
                    seen += 1
            assert seen >= 1, "the first asynchronous frame cannot have fitted buffers sized for two rays per sample"
            assert attempt < 11 and np.array_equal(img, want["bgr"])
            # the overflow of a call nobody asked about is reported by the next call instead
            small = context()
            try:
                small.render_rows_async(opt, 5, 0, 16, img)
                import time
                deadline = time.time() + 5.0
                with pytest.raises(_native.NativeError) as info:
                    while time.time() < deadline:                # the flag appears when the kernels have run
                        small.render_rows_async(opt, 5, 0, 16, img)
                assert info.value.code == _native.E_CAPACITY
            finally:
                small.close()
        finally:
            fresh.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("path", ["brute", "lbvh"])
def test_cancel_stops_inside_a_render_and_the_context_recovers(path):
    """The cancel callback is asked before every band and after the launches of every tree level."""
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    ctx = _native.Context(0)
    try:
        ctx.set_accel(path)
        ctx.upload(flat)
        opt = dict(width=160, height=96, bounces=4, samples=2)
        want = ctx.render_rows(opt, seed=1)
        polls = []

        def stop_at(k):
            def poll():
                polls.append(1)
                return len(polls) > k
            return poll

        for k in (0, 1, 3):          # before the band, after its first level, after its third
            polls.clear()
            ctx.set_cancel(stop_at(k))
            with pytest.raises(_native.Cancelled):
                ctx.render_rows(opt, seed=1)
            assert len(polls) == k + 1
            with pytest.raises(_native.NativeError):                 # a cancelled call has no statistics
                ctx.last_call_stats()
        polls.clear()
        ctx.set_cancel(lambda: polls.append(1) and False)            # asked, never stops: 1 band + 5 levels
        got = ctx.render_rows(opt, seed=1)
        assert len(polls) == 6 and np.array_equal(got["bgr"], want["bgr"]) and got["stats"]["queries"] == want["stats"]["queries"]
        ctx.set_cancel(None)
        polls.clear()
        again = ctx.render_rows(opt, seed=1)
        assert not polls and np.array_equal(again["bgr"], want["bgr"]) and again["stats"]["queries"] == want["stats"]["queries"]
    finally:
        ctx.close()


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_frame_on_several_contexts(accel):
    """t3d_render_frame_multi with three contexts (of the one device a test box has): same frame and counters as one
    context alone; a height the context count does not divide is refused."""
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    opt = dict(width=96, height=54, bounces=3, samples=2)
    ctxs = [_native.Context(0) for _ in range(3)]
    try:
        for c in ctxs:
            c.set_accel(accel)
            c.upload(flat)
        alone = ctxs[0].render_rows(opt, seed=2)
        frame = _native.pinned_image((54, 96, 3))
        frame[:] = 7
        st = _native.render_frame_multi(ctxs, opt, 2, out=frame)
        assert np.array_equal(frame, alone["bgr"])
        for k in ("queries", "rays_api", "nodes", "shadow_queries", "trace_queries"):
            assert st[k] == alone["stats"][k], k
        pageable = np.zeros((54, 96, 3), np.uint8)               # any host memory will do, pinned is only faster
        _native.render_frame_multi(ctxs[:2], opt, 2, out=pageable)
        assert np.array_equal(pageable, alone["bgr"])
        with pytest.raises(_native.NativeError, match="factor of image height"):
            _native.render_frame_multi(ctxs, dict(opt, height=55), 2, out=np.zeros((55, 96, 3), np.uint8))
        # trees that outgrow the unchecked call's buffers: the tile is rendered again, checked
        glass = nested_glass_scene()
        for c in ctxs:
            c.upload(glass)
        gopt = dict(width=24, height=18, bounces=9, samples=2)
        galone = ctxs[0].render_rows(gopt, seed=5)
        fresh = [_native.Context(0) for _ in range(3)]
        try:
            for c in fresh:
                c.set_accel(accel)
                c.upload(glass)
            gframe = _native.pinned_image((18, 24, 3))
            gst = _native.render_frame_multi(fresh, gopt, 5, out=gframe)
            assert np.array_equal(gframe, galone["bgr"]) and gst["queries"] == galone["stats"]["queries"]
        finally:
            for c in fresh:
                c.close()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_several_frames_per_call(accel, monkeypatch):
    """t3d_opts.frames: one call renders its rows several times, frame f with the seed seed + f * frame_seed_step, into
    outputs that hold the frames one after the other -- pixel for pixel (image, radiance, primary hits) what separate calls
    give, and the counters are their sums; also when the call is cut into bands that straddle frames, with a row stride,
    through device pointers, and with the entry point that allocates nothing."""
    import torch
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    opt = dict(width=48, height=45, bounces=3, samples=2)

    def context():
        c = _native.Context(0)
        c.set_accel(accel)
        c.upload(flat)
        return c

    ctx = context()
    try:
        B, step, seed = 3, 5, 11
        single = [ctx.render_rows(opt, seed=seed + f * step, row0=2, nrows=9, row_stride=5, want_radiance=True, want_primary=True) for f in range(B)]
        assert not np.array_equal(single[0]["bgr"], single[1]["bgr"])      # the seeds matter in this scene
        got = ctx.render_rows(opt, seed=seed, row0=2, nrows=9, row_stride=5, want_radiance=True, want_primary=True, frames=B, frame_seed_step=step)
        assert got["bgr"].shape == (B, 9, 48, 3)
        for f in range(B):
            for k in ("bgr", "radiance", "prim"):
                assert np.array_equal(got[k][f], single[f][k]), (k, f)
            assert np.array_equal(got["prim_t"][f], single[f]["prim_t"], equal_nan=True)
        for k in ("queries", "rays_api", "nodes", "shadow_queries", "trace_queries", "shadow_culled"):
            assert got["stats"][k] == sum(s["stats"][k] for s in single), k
        # 2 x 12 rows: a band whose rays are walked in 16 x 8 pixel tiles, one of which straddles the two frames
        two = ctx.render_rows(opt, seed=seed, row0=1, nrows=12, row_stride=3, want_radiance=True, frames=2, frame_seed_step=step)
        for f in range(2):
            one = ctx.render_rows(opt, seed=seed + f * step, row0=1, nrows=12, row_stride=3, want_radiance=True)
            assert np.array_equal(two["bgr"][f], one["bgr"]) and np.array_equal(two["radiance"][f], one["radiance"]), f
        # a step of zero: B copies of the same frame
        same = ctx.render_rows(opt, seed=seed, row0=2, nrows=9, row_stride=5, frames=2)
        assert np.array_equal(same["bgr"][0], single[0]["bgr"]) and np.array_equal(same["bgr"][1], single[0]["bgr"])
        # device pointers, asynchronous, and the entry point that allocates nothing
        for static in (False, True):
            out = torch.zeros((B, 9, 48, 3), dtype=torch.uint8, device="cuda:0")
            stream = torch.cuda.Stream()
            for attempt in range(3):       # an asynchronous call may first have to learn the capacity the trees need
                ctx.render_rows_device(opt, seed, 2, 9, out.data_ptr(), stream=stream.cuda_stream, want_stats=False, row_stride=5,
                                       static=static, frames=B, frame_seed_step=step)
                try:
                    ctx.last_call_stats()
                    break
                except _native.NativeError as e:
                    assert e.code == _native.E_CAPACITY
            assert np.array_equal(out.cpu().numpy(), got["bgr"]), static
        with pytest.raises(_native.NativeError):
            ctx.render_rows(opt, seed=seed, row0=2, nrows=9, row_stride=5, frames=-1)
    finally:
        ctx.close()
    # bands of a few rows each, which straddle the frames
    monkeypatch.setenv("T3D_MEM_BUDGET_MB", "1")
    ctx = context()
    try:
        banded = ctx.render_rows(opt, seed=seed, row0=2, nrows=9, row_stride=5, want_radiance=True, frames=B, frame_seed_step=step)
        assert banded["stats"]["n_k_level"] >= 4 * 2, "the call was meant to be cut into several bands"
        assert np.array_equal(banded["bgr"], got["bgr"]) and np.array_equal(banded["radiance"], got["radiance"])
    finally:
        ctx.close()


def test_frame_on_several_contexts_into_device_memory():
    import torch
    from raytracing_amd import _native
    flat = flat_for("cornellScene.json")
    opt = dict(width=64, height=48, bounces=2, samples=1)
    ctxs = [_native.Context(0) for _ in range(2)]
    try:
        for c in ctxs:
            c.upload(flat)
        alone = ctxs[0].render_rows(opt, seed=0)
        frame = torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        _native.render_frame_multi(ctxs, opt, 0, frame_ptr=frame.data_ptr())
        assert np.array_equal(frame.cpu().numpy(), alone["bgr"])
    finally:
        for c in ctxs:
            c.close()


def test_photon_render_reuses_the_device_scene_and_sees_edits():
    """A second render() of an unchanged scene uploads nothing (the device starts on the frame while the host checks the
    scene's fingerprint); an edit between two renders shows, as it does in the reference, which copies the scene on
    every call (T3DR.py:1090)."""
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("cornellScene.json"))
    opts = {"width": 96, "height": 72, "bounces": 3, "samples": 2, "seed": 4}
    photon = T3DR.Photon(scene, opts)
    first = photon.render()
    assert photon.uploads == 1
    second = photon.render()
    assert photon.uploads == 1 and np.array_equal(first.imageAs("np"), second.imageAs("np")) and second.rays == first.rays
    keep = first.imageAs("np").copy()
    third = photon.render(returnType="np")                       # buffers of earlier results are not recycled under them
    assert np.array_equal(first.imageAs("np"), keep) and np.array_equal(third, first.imageAs("cv2"))
    ball = next(o for o in scene.objects if o.type == "sphere")
    ball.position = T3DE.Vec3(ball.position.x + 0.7, ball.position.y, ball.position.z)
    moved = photon.render()
    assert photon.uploads == 2 and not np.array_equal(moved.imageAs("np"), keep)
    fresh = T3DR.Photon(scene, dict(opts)).render()
    assert np.array_equal(moved.imageAs("np"), fresh.imageAs("np")) and moved.rays == fresh.rays
    photon.options["bounces"] = 1                                # options are read on every call (explorer.py:1171)
    assert not np.array_equal(photon.render().imageAs("np"), moved.imageAs("np"))
    # trees that do not fit an unchecked call's buffers: the frame is rendered again, checked
    from raytracing_amd.scene import Camera, Light, Material, Scale, Scene, Shader, Sphere, Vec3, RGB
    s = Scene()
    Camera(s, position=Vec3(0, 0, -4), length=30)
    for r in (3.0, 2.6, 2.2, 1.8, 1.4, 1.0):
        b = Sphere(s, position=Vec3(0, 0, 6))
        b.scale = Scale(2 * r, 2 * r, 2 * r)
        b.setShader(Shader(color=RGB(200, 220, 250), roughness=0.0, reflectivity=0.5))
        b.material = Material(ior=1.3)
    Light(s, position=Vec3(4, 5, -2), strength=1.0)
    glass = T3DR.Photon(s, {"width": 24, "height": 16, "bounces": 9, "samples": 2, "seed": 5})
    shallow = dict(glass.options, bounces=1)
    glass.options["bounces"] = 1
    glass.render()                                               # small buffers, scene on the device
    glass.options["bounces"] = 9
    deep = glass.render()                                        # starts unchecked, overflows, is rendered again
    assert np.array_equal(deep.imageAs("np"), T3DR.Photon(s, dict(shallow, bounces=9)).render().imageAs("np"))


def test_photon_render_on_several_gpus():
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("cornellScene.json"))
    opts = {"width": 64, "height": 48, "bounces": 2, "samples": 2, "seed": 3}
    one = T3DR.Photon(scene, dict(opts)).render()
    calls = []
    multi = T3DR.Photon(scene, dict(opts, gpus=3, devices=[0, 0, 0]))
    got = multi.render(progressCallback=lambda table, key, total: calls.append((key, table[key], total)))
    assert np.array_equal(got.imageAs("np"), one.imageAs("np")) and got.rays == one.rays and not got.failed
    assert calls[0] == (0, 0, 48 * 64) and calls[-1] == (0, 48 * 64 - 1, 48 * 64)
    assert np.array_equal(multi.render(returnType="np"), one.imageAs("cv2")) and multi.uploads == 3
    with pytest.raises(ValueError, match="factor of image height"):
        T3DR.Photon(scene, dict(opts, gpus=5, devices=[0] * 5)).render()
    assert isinstance(got.traces[(32, 30)][0], T3DE.TraceTreeNode)     # single-ray queries use a context of their own
    assert multi.uploads == 4


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_rows_as_bands_traced_at_once(accel):
    """t3d_render_rows_device_split / t3d_render_rows_async_split: the rows of a call as contiguous bands on several
    contexts of the device at once -- the same pixels and, summed over the contexts, the same counters as one call."""
    import torch
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    opt = dict(width=96, height=54, bounces=3, samples=2)
    ctxs = [_native.Context(0) for _ in range(3)]
    try:
        for c in ctxs:
            c.set_accel(accel)
            c.upload(flat)
        want = ctxs[0].render_rows(opt, seed=2)
        keys = ("queries", "rays_api", "nodes", "shadow_queries", "trace_queries", "shadow_culled")
        stream = torch.cuda.Stream()
        for n in (1, 2, 3):
            out = torch.zeros((54, 96, 3), dtype=torch.uint8, device="cuda:0")
            for attempt in range(4):                       # an unchecked call may first have to learn the capacity it needs
                _native.render_rows_split(ctxs[:n], opt, 2, 0, 54, bgr_ptr=out.data_ptr(), stream=stream.cuda_stream)
                try:
                    st = _native.split_call_stats(ctxs[:n])
                    break
                except _native.NativeError as e:
                    assert e.code == _native.E_CAPACITY
            stream.synchronize()
            assert np.array_equal(out.cpu().numpy(), want["bgr"]), n
            for k in keys:
                assert st[k] == want["stats"][k], (k, n)
        # a block of rows with a stride, into pinned host memory; fewer rows than eight per band
        part = ctxs[0].render_rows(opt, seed=2, row0=3, nrows=10, row_stride=4)
        img = _native.pinned_image((10, 96, 3))
        _native.render_rows_split(ctxs, opt, 2, 3, 10, out=img, row_stride=4)
        st = _native.split_call_stats(ctxs)
        assert np.array_equal(img, part["bgr"]) and st["queries"] == part["stats"]["queries"]
        with pytest.raises(_native.NativeError):             # one frame per call
            lib_opts = dict(opt)
            from raytracing_amd.flatten import make_opts
            import ctypes as C
            o = make_opts(lib_opts, 2, frames=2)
            handles = (C.c_void_p * 2)(ctxs[0]._h, ctxs[1]._h)
            ctxs[0]._check(_native.load().t3d_render_rows_device_split(handles, 2, C.byref(o), 0, 54, C.c_void_p(out.data_ptr()), None), "split")
        with pytest.raises(_native.NativeError):             # the same context twice
            _native.render_rows_split([ctxs[0], ctxs[0]], opt, 2, 0, 54, bgr_ptr=out.data_ptr())
    finally:
        for c in ctxs:
            c.close()


def test_photon_render_in_bands():
    """options['bands']: a plain render() of an uploaded scene is traced as bands at once; same image, rays and stats."""
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("cornellScene.json"))
    opts = {"width": 96, "height": 72, "bounces": 3, "samples": 2, "seed": 4}
    one = T3DR.Photon(scene, dict(opts, bands=1))
    a0, a1 = one.render(), one.render()
    for bands in (2, 3):
        many = T3DR.Photon(scene, dict(opts, bands=bands))
        b0 = many.render()                                     # first call: checked, one context
        b1 = many.render()                                     # unchecked, in bands
        b2 = many.render()
        for b in (b0, b1, b2):
            assert np.array_equal(b.imageAs("np"), a0.imageAs("np")) and b.rays == a0.rays
        assert b2.stats["queries"] == a1.stats["queries"] and many.uploads == 1 and len(many._helpers) == bands - 1
        ball = next(o for o in scene.objects if o.type == "sphere")
        ball.position = T3DE.Vec3(ball.position.x + 0.3, ball.position.y, ball.position.z)
        moved = many.render()
        again = many.render()
        fresh = T3DR.Photon(scene, dict(opts, bands=1)).render()
        assert np.array_equal(moved.imageAs("np"), fresh.imageAs("np")) and np.array_equal(again.imageAs("np"), fresh.imageAs("np"))
        ball.position = T3DE.Vec3(ball.position.x - 0.3, ball.position.y, ball.position.z)
    # the default: two bands for a large frame of a small scene, one otherwise
    big = T3DR.Photon(scene, {"width": 1024, "height": 512, "bounces": 1})
    big.render()
    assert big._bands() == 2 and one._bands() == 1
    # deep trees in bands: the unchecked bands overflow, the frame is rendered again, checked
    s = _glass_onion(T3DE)
    glass = T3DR.Photon(s, {"width": 24, "height": 32, "bounces": 1, "samples": 2, "seed": 5, "bands": 2})
    glass.render()
    glass.options["bounces"] = 9
    deep = glass.render()
    assert np.array_equal(deep.imageAs("np"), T3DR.Photon(s, {"width": 24, "height": 32, "bounces": 9, "samples": 2, "seed": 5}).render().imageAs("np"))
    deep2 = glass.render()
    assert np.array_equal(deep2.imageAs("np"), deep.imageAs("np")) and deep2.rays == deep.rays


def test_photon_host_tree_limit_option():
    """options['hostTreeLimit'] (t3d_set_host_tree_limit): who builds the hierarchy of the suzanne scene changes how many
    boxes a frame tests and nothing else."""
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("config2_suzanne.json"))
    opts = {"width": 96, "height": 56, "bounces": 3, "samples": 1, "seed": 8}
    host = T3DR.Photon(scene, dict(opts)).render()                     # 970 primitives: cut on the host by default
    dev = T3DR.Photon(scene, dict(opts, hostTreeLimit=0)).render()
    assert np.array_equal(host.imageAs("np"), dev.imageAs("np")) and host.rays == dev.rays
    assert 0 < host.stats["aabb_tests"] < 0.97 * dev.stats["aabb_tests"]


def test_photon_background_tree_option():
    """options['backgroundTree'] (t3d_set_background_tree): with the device building the hierarchy of the suzanne scene, a
    host thread cuts the surface-area tree meanwhile; once it is in place the frames test as few boxes as with the tree cut
    at upload, and every frame on the way is the same picture."""
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("config2_suzanne.json"))
    opts = {"width": 96, "height": 56, "bounces": 3, "samples": 1, "seed": 8}
    host = T3DR.Photon(scene, dict(opts)).render()
    ph = T3DR.Photon(scene, dict(opts, hostTreeLimit=0, backgroundTree=True))
    frames = [ph.render() for _ in range(3)]
    assert ph._ctx.tree_status(wait=True) == "background" and ph.uploads == 1
    frames.append(ph.render())
    for f in frames:
        assert np.array_equal(host.imageAs("np"), f.imageAs("np")) and host.rays == f.rays
    assert abs(frames[-1].stats["aabb_tests"] - host.stats["aabb_tests"]) < 0.005 * host.stats["aabb_tests"]


@pytest.mark.parametrize("workload", ["cornell", "monkey"])
def test_live_scene_updates_in_place_when_only_spheres_lights_or_the_camera_move(workload):
    """Picking on a live scene (transformBefore False, explorer.py:324-355): when only spheres, Lights or the camera moved
    since the last query the device copy is updated in place (t3d_update_scene: no plane grouping; the hierarchy of the
    monkey scene is refitted); a moved mesh is uploaded.  Picks, trees and renders equal those of a fresh Photon."""
    import T3DE
    import T3DR
    from raytracing_amd.workloads import cornell_scene
    scene = cornell_scene(with_light=True, suzanne=workload == "monkey")
    opts = {"width": 64, "height": 48, "bounces": 2, "samples": 1, "seed": 2, "transformBefore": False}
    live = T3DR.Photon(scene, dict(opts))

    def picks(photon):
        out = []
        for x, y in ((32, 24), (20, 30), (44, 30), (30, 12), (5, 5)):
            hit = photon.getFirstCollision(photon.getRay(photon.pixelToVector(x, y)), True, False)
            out.append((hit["object"].id if hit["object"] else None, None if hit["t"] is None else round(hit["t"], 4)))
        return out

    first = picks(live)
    assert live.uploads == 1 and live.updates == 0
    assert picks(live) == first and live.uploads == 1                       # nothing moved: nothing sent
    light = next(o for o in scene.objects if o.type == "light")
    ball = next(o for o in scene.objects if o.type == "sphere")
    light.position = T3DE.Vec3(0.4, 2.6, 0.8)
    ball.position = T3DE.Vec3(ball.position.x + 0.9, ball.position.y + 0.5, ball.position.z - 0.3)
    ball.scale = T3DE.Scale(2.4, 2.4, 2.4)
    scene.camera.position = T3DE.Vec3(0.3, 0.1, -10.5)
    moved = picks(live)
    assert live.uploads == 2 and live.updates == 1
    fresh = T3DR.Photon(scene, dict(opts))
    assert moved == picks(fresh) and moved != first
    a, b = live.render(), fresh.render()
    assert np.array_equal(a.imageAs("np"), b.imageAs("np")) and a.rays == b.rays
    assert live.uploads == 2
    # a mesh moves: an upload
    mesh = next(o for o in scene.objects if o.type == "mesh" and o.name != "Box")
    mesh.position = T3DE.Vec3(mesh.position.x - 0.4, mesh.position.y, mesh.position.z)
    after = picks(live)
    assert live.uploads == 3 and live.updates == 1 and after == picks(T3DR.Photon(scene, dict(opts)))
    # a single coordinate edited in place is seen by the next render (which compares the meshes entry by entry)
    mesh.mesh.vertices[1].y += 0.25
    c, d = live.render(), T3DR.Photon(scene, dict(opts)).render()
    assert live.uploads == 4 and np.array_equal(c.imageAs("np"), d.imageAs("np"))
    # the library refuses an update that is not the uploaded scene
    from raytracing_amd import _native
    from raytracing_amd.flatten import flatten_scene
    ctx = live._context()
    T3DE.Light(scene, position=T3DE.Vec3(1, 1, 1))
    with pytest.raises(_native.NativeError, match="not the uploaded scene"):
        ctx.update(flatten_scene(scene))


def _glass_onion(T3DE):
    """Six nested glass spheres: trace trees that double with every bounce."""
    from raytracing_amd.scene import Camera, Light, Material, Scale, Scene, Shader, Sphere, Vec3, RGB
    s = Scene()
    Camera(s, position=Vec3(0, 0, -4), length=30)
    for r in (3.0, 2.6, 2.2, 1.8, 1.4, 1.0):
        b = Sphere(s, position=Vec3(0, 0, 6))
        b.scale = Scale(2 * r, 2 * r, 2 * r)
        b.setShader(Shader(color=RGB(200, 220, 250), roughness=0.0, reflectivity=0.5))
        b.material = Material(ior=1.3)
    Light(s, position=Vec3(4, 5, -2), strength=1.0)
    return s


def test_photon_render_discards_the_overflow_of_a_frame_it_threw_away():
    """render() starts the device on the scene it uploaded last time before it knows whether the scene changed.  When it
    did change, that frame is thrown away -- and an overflow of ITS node buffers (new options: deeper trees) must not be
    reported by the checked call that follows for the new scene."""
    import T3DE
    import T3DR
    s = _glass_onion(T3DE)
    photon = T3DR.Photon(s, {"width": 24, "height": 16, "bounces": 1, "samples": 2, "seed": 5})
    photon.render()                                              # small buffers, scene on the device
    photon.options["bounces"] = 9                                # the speculative frame will overflow them ...
    light = next(o for o in s.objects if o.type == "light")
    light.position = T3DE.Vec3(3, 5, -2)                         # ... and is not wanted: the scene changed
    deep = photon.render()
    assert photon.uploads == 2
    fresh = T3DR.Photon(s, {"width": 24, "height": 16, "bounces": 9, "samples": 2, "seed": 5}).render()
    assert np.array_equal(deep.imageAs("np"), fresh.imageAs("np")) and deep.rays == fresh.rays


def test_traces_follow_scene_edits_between_renders_on_several_gpus():
    """The context that serves traces / picking is not one of the several-GPU path's: after an edit and another render on
    several contexts it must be given the edited scene too."""
    import T3DE
    import T3DR
    scene = T3DE.Scene.fromJSON(scene_path("cornellScene.json"))
    opts = {"width": 64, "height": 48, "bounces": 2, "samples": 2, "seed": 3}
    multi = T3DR.Photon(scene, dict(opts, gpus=2, devices=[0, 0]))
    before = multi.render()
    hit_before = before.traces[(20, 30)][0].collisionInfo["point"]
    ball = next(o for o in scene.objects if o.type == "sphere")
    ball.position = T3DE.Vec3(ball.position.x + 0.9, ball.position.y + 0.4, ball.position.z)
    after = multi.render()
    single = T3DR.Photon(scene, dict(opts))
    want = single.render()
    assert np.array_equal(after.imageAs("np"), want.imageAs("np"))
    moved = 0
    for key in [(20, 30), (14, 33), (25, 36), (40, 20)]:
        a, b = after.traces[key][0], want.traces[key][0]
        assert (a is None) == (b is None)
        if a is not None:
            assert a.collisionInfo["object"].id == b.collisionInfo["object"].id
            assert np.allclose([*a.collisionInfo["point"]], [*b.collisionInfo["point"]], atol=1e-5)
    pick = multi.getFirstCollision(multi.getRay(multi.pixelToVector(20, 30)), True, False)
    ref = single.getFirstCollision(single.getRay(single.pixelToVector(20, 30)), True, False)
    assert (pick["object"].id, pick["t"]) == (ref["object"].id, ref["t"])
    assert hit_before is not None


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_counters_of_an_unwaited_call_cover_all_its_bands(accel, monkeypatch):
    """A call that is not waited for and is cut into several bands reports, through t3d_last_call_stats, the counters of
    all its bands (summed on the device), not those of the last one: the asynchronous host render, the device-pointer
    call without stats (also through the static work array), and the frame on several contexts."""
    import torch
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    opt = dict(width=96, height=54, bounces=3, samples=2)
    monkeypatch.setenv("T3D_MEM_BUDGET_MB", "1")       # a few rows per band
    keys = ("queries", "rays_api", "nodes", "shadow_queries", "trace_queries", "shadow_culled")
    if accel == "brute":      # through the hierarchy the number of tests depends on which rays share a wave, i.e. on the buffer sizes
        keys += ("tri_tests", "sphere_tests")
    ctxs = [_native.Context(0) for _ in range(3)]
    try:
        for c in ctxs:
            c.set_accel(accel)
            c.upload(flat)
        want = ctxs[0].render_rows(opt, seed=2)
        assert want["stats"]["n_k_level"] >= 4 * 3, "the frame was meant to be cut into several bands"
        img = _native.pinned_image((54, 96, 3))
        for attempt in range(4):                       # an unchecked call may first have to learn the capacity it needs
            ctxs[0].render_rows_async(opt, 2, 0, 54, img)
            try:
                st = ctxs[0].last_call_stats()
                break
            except _native.NativeError as e:
                assert e.code == _native.E_CAPACITY
        assert np.array_equal(img, want["bgr"])
        for k in keys + ("n_k_level",):
            assert st[k] == want["stats"][k], k
        assert list(st["level_rays"]) == list(want["stats"]["level_rays"]) and list(st["level_nodes"]) == list(want["stats"]["level_nodes"])
        assert st["ms_device"] > 0
        for static in (False, True):
            out = torch.zeros((54, 96, 3), dtype=torch.uint8, device="cuda:0")
            for attempt in range(4):
                ctxs[0].render_rows_device(opt, 2, 0, 54, out.data_ptr(), want_stats=False, static=static)
                try:
                    st = ctxs[0].last_call_stats()
                    break
                except _native.NativeError as e:
                    assert e.code == _native.E_CAPACITY
            assert np.array_equal(out.cpu().numpy(), want["bgr"])
            for k in keys:
                assert st[k] == want["stats"][k], (k, static)
        # a one-band call afterwards reports its own counters again
        small = ctxs[0].render_rows(dict(opt, height=2), seed=2)
        ctxs[0].render_rows_async(dict(opt, height=2), 2, 0, 2, _native.pinned_image((2, 96, 3)))
        assert ctxs[0].last_call_stats()["queries"] == small["stats"]["queries"]
        frame = _native.pinned_image((54, 96, 3))
        mst = _native.render_frame_multi(ctxs, opt, 2, out=frame)
        assert np.array_equal(frame, want["bgr"])
        for k in keys:
            assert mst[k] == want["stats"][k], k
    finally:
        for c in ctxs:
            c.close()


def test_single_ray_trees_with_many_lights_need_little_memory(gpu_ctx, oracle_mod):
    """The tree of one pixel at 12 bounces in a scene whose 968 emissive faces are 968 lights: the work set of a
    single ray is sized by its tree (at most 2^12 nodes a level), not by a band of rows."""
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.scene import Scene
    scene = Scene.fromJSON(scene_path("config2_suzanne.json"))
    monkey = next(o for o in scene.objects if o.type == "mesh" and len(o.mesh.faces) > 500)
    for sh in monkey.shaders:
        sh.emissionStrength = 0.05
    flat = flatten_scene(scene)
    assert len(flat.light_strength) >= 968
    gpu_ctx.upload(flat)
    opt = dict(width=96, height=54, bounces=12, samples=1)
    nodes = gpu_ctx.trace_pixel(opt, 30, 40, 0, seed=1)
    ref = oracle_mod.render(flat, opt, seed=1, row0=40, nrows=1, want_nodes=True, node_cap=400000)
    want = {int(n["heap"]): n for n in ref["nodes"] if (int(n["px"]), int(n["py"]), int(n["sample"])) == (30, 40, 0)}
    assert sorted(n.heap for n in nodes) == sorted(want) and len(nodes) >= 1
    for n in nodes:
        assert np.allclose(list(n.color), want[n.heap]["color"], atol=0.05)


def test_a_replayed_launch_graph_takes_the_seed_of_every_call():
    """The frames of an animation: the same band asked for again and again, every time with another seed.  From the third
    request on the band is one launch graph, and the graph's first kernel node -- the only launch that reads the seed -- is
    given each call's parameters before the replay.  Every frame must equal the frame a checked (ungraphed) call renders."""
    import torch
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    opt = dict(width=96, height=64, bounces=3, samples=1)
    ctx = _native.Context(0)
    try:
        ctx.upload(flat)
        want = [ctx.render_rows(opt, seed=s)["bgr"] for s in range(9)]
        assert not np.array_equal(want[0], want[1])
        ctx.set_timing(False)
        out = torch.zeros((64, 96, 3), dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.Stream()
        before = ctx.graph_counters()
        order = [0, 0, 0, 0, 0, 0, 1, 2, 2, 3, 8, 0, 5, 5, 4]
        for s in order:
            ctx.render_rows_device(opt, s, 0, 64, out.data_ptr(), stream=stream.cuda_stream, want_stats=False)
            stream.synchronize()
            assert np.array_equal(out.cpu().numpy(), want[s]), s
        after = ctx.graph_counters()
        assert after["captured"] - before["captured"] >= 1
        assert after["replayed"] - before["replayed"] >= 8, after
        assert after["reseeded"] - before["reseeded"] >= 6, after     # 0->1, 1->2, 2->3, 3->8, 8->0, 0->5, 5->4 on one graph or two
        # several frames per call: the seed of frame f is seed + f * step, and the first seed is what a replay changes
        out2 = torch.zeros((2, 64, 96, 3), dtype=torch.uint8, device="cuda:0")
        for s in (0, 0, 0, 0, 0, 0, 3, 6, 1):
            ctx.render_rows_device(opt, s, 0, 64, out2.data_ptr(), stream=stream.cuda_stream, want_stats=False, frames=2, frame_seed_step=2)
            stream.synchronize()
            got = out2.cpu().numpy()
            assert np.array_equal(got[0], want[s]) and np.array_equal(got[1], want[s + 2]), s
        ctx.last_call_stats()
    finally:
        ctx.close()


def _run_bench_ranks(n, extra, steps=3, warmup=1, timeout=280):
    import json
    import os
    import subprocess
    import sys
    from test_tiles import _free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    launcher = [sys.executable]
    if n > 1:
        launcher += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                     "--master-port", str(_free_port())]
    cmd = launcher + [os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + extra
    env = dict(os.environ, MASTER_PORT=str(_free_port())) if n == 1 else None
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


FRAME_QUERIES = 17874992     # closest-hit queries of the benchmark frame with seed 0, in the reference's count


def _check_frame_counts(d):
    """Every frame of the run has its own seed, so the per-frame counts are averages: near those of the frame with seed 0,
    and the traced and the untraced queries add up to the reference's count."""
    cfg = d["config"]
    assert abs(cfg["reference_queries_per_frame"] - FRAME_QUERIES) < 0.002 * FRAME_QUERIES
    assert abs(cfg["queries_per_frame"] + cfg["shadow_rays_answered_untraced_per_frame"] - cfg["reference_queries_per_frame"]) < 0.5


def test_bench_with_two_ranks_rehearsed_on_one_gpu():
    """bench.py as the driver starts it for N > 1 -- torch.distributed.run, one process per rank, interleaved rows,
    several frames in flight per rank, a gather per frame, counters summed, time the maximum over ranks -- with the
    gloo backend, so that two ranks can share the one GPU a test box has (RCCL refuses that).  bench.py itself compares
    every frame of every chain's last call with the tile rendered alone and raises on a difference."""
    d = _run_bench_ranks(2, ["--dist-backend", "gloo"])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["steps"] == 3
    assert d["scaling"] == "strong (batched, 2 frames per call)"
    assert d["config"]["partition"] == "interleaved rows x2" and "gloo" in d["config"]["collective"]
    _check_frame_counts(d)
    assert d["value"] > 0 and "cpu_baseline" not in d          # the CPU baseline is a one-rank field
    # a call carries a rank's rows of two frames; the three timed frames are one such call and one call of a single frame
    assert d["config"]["frames_per_call"] == 2 and d["config"]["frames_in_flight"] == 2 * d["config"]["calls_in_flight"]
    # beside the batched figure: one frame per call, the latency of one frame, and the whole frame on one GPU alone
    sf = d["single_frame"]
    assert sf["frames_per_call"] == 1 and sf["ms_per_frame"] > 0 and sf["frame_latency_ms"] > 0 and sf["one_gpu_whole_frame_alone_ms"] > 0
    assert d["ms_per_step_varying_seed"] == d["ms_per_step"] and d["ms_per_step_fixed_seed"] > 0


def test_bench_with_four_ranks_rehearsed_on_one_gpu():
    """Four ranks: a call carries a rank's 270 rows of four frames, whose pixel tiles (eight rows high) straddle the frames."""
    d = _run_bench_ranks(4, ["--dist-backend", "gloo", "--no-secondary"])
    assert d["n_gpus"] == 4 and d["config"]["partition"] == "interleaved rows x4" and d["steps"] == 3
    assert d["config"]["frames_per_call"] == 4 and d["config"]["calls_in_flight"] == 4 and d["config"]["frames_in_flight"] == 16
    _check_frame_counts(d)
    assert "single_frame" not in d


def test_bench_with_eight_ranks_and_the_drivers_arguments():
    """The command the driver runs on an 8-GPU node, argument for argument (20 steps after 5 of warm-up: two calls of eight
    frames and one of four), with eight ranks sharing this box's GPU over gloo."""
    d = _run_bench_ranks(8, ["--dist-backend", "gloo"], steps=20, warmup=5, timeout=420)
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["partition"] == "interleaved rows x8"
    assert d["scaling"] == "strong (batched, 8 frames per call)"
    assert d["config"]["frames_per_call"] == 8 and d["config"]["calls_in_flight"] == 4
    _check_frame_counts(d)
    assert d["single_frame"]["ms_per_frame"] > 0 and d["single_frame"]["speedup_vs_one_gpu_frame_alone"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e3 - d["config"]["queries_per_frame"]) < 1e-3 * d["config"]["queries_per_frame"]


def test_bench_gather_in_a_one_rank_rccl_group():
    """One rank, but in a real RCCL process group: every frame of every chain goes through dist.gather and the strided copy
    on its chain's stream while the other chains' launch graphs replay -- the several-rank code path, collective library
    and its watchdog thread included, on a one-GPU box.  bench.py checks the gathered frames against the tiles."""
    d = _run_bench_ranks(1, ["--one-rank-group", "--no-cpu-baseline"], steps=24, warmup=2)
    assert d["n_gpus"] == 1 and d["config"]["collective"] == "RCCL gather in a group of one rank"
    assert d["config"]["calls_in_flight"] == 4 and d["scaling"] == "strong"
    _check_frame_counts(d)


def test_bench_with_two_ranks_over_rccl():
    """The same with the RCCL backend, one GPU per rank: runs where two GPUs are visible."""
    from raytracing_amd import _native
    if _native.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _run_bench_ranks(2, [])
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "RCCL gather"
    _check_frame_counts(d)


def test_demo_script_writes_the_scene_and_the_picture(tmp_path, monkeypatch, oracle_mod):
    """cornell_test.py, the counterpart of the reference's demo script (cornellTest.py:7-155): the scene it builds through
    the API loads again from the JSON it writes (same geometry), the PNG holds the frame it rendered, and the frame is the oracle's."""
    import json
    import struct
    import sys
    import zlib
    import cornell_test
    import T3DE
    from raytracing_amd.flatten import flatten_scene
    out = tmp_path / "renders"
    monkeypatch.setattr(sys, "argv", ["cornell_test.py", "--width", "64", "--height", "40", "--bounces", "2", "--samples", "2", "--seed", "5",
                                      "--light", "--out", str(out)])
    render = cornell_test.main()
    files = sorted(p.name for p in out.iterdir())
    assert files[0] == "cornellScene.json" and files[1].startswith("cornell|") and files[1].endswith(f"|{render.rays}.png")
    from raytracing_amd.workloads import cornell_scene
    flat = flatten_scene(cornell_scene(with_light=True))
    again = flatten_scene(T3DE.Scene.fromDict(json.load(open(out / "cornellScene.json"))))   # (the loader puts the camera first)
    assert again.n_tris == flat.n_tris and again.n_spheres == flat.n_spheres and len(again.light_strength) == len(flat.light_strength)
    assert np.array_equal(np.sort(again.tri_verts.reshape(-1, 9), axis=0), np.sort(flat.tri_verts.reshape(-1, 9), axis=0))
    opt = dict(width=64, height=40, bounces=2, samples=2, ambient=(32, 32, 32))
    ref = oracle_mod.render(flat, opt, seed=5)
    img = render.imageAs("np")                                   # RGB
    diff = np.abs(img.astype(int) - ref["bgr"][:, :, ::-1].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999 and render.rays == ref["stats"]["rays_api"]
    # the PNG: 8-bit RGB, one filter byte per row, the same pixels
    data = open(out / files[1], "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, size = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            size = struct.unpack(">II", body[:8])
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(size[1], 1 + 3 * size[0])
    assert size == (64, 40) and (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(40, 64, 3), img)
