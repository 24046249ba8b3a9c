"""t3d_render_rows_device_static -- the render entry point that allocates nothing (its work buffers are pieces of the
library's per-GPU static device array) -- against t3d_render_rows_device, through the C ABI: identical pixels, primary
ids, distances and counters, on two calls in a row, the second with a smaller input; both closest-hit paths."""
import numpy as np
import pytest

from conftest import flat_for

pytestmark = pytest.mark.gpu


def _both(ctx, opt, seed, row0, nrows, stride, want_stats):
    import torch
    W = opt["width"]
    outs = []
    for static in (False, True):
        bgr = torch.full((nrows, W, 3), 9, dtype=torch.uint8, device="cuda:0")
        rad = torch.zeros((nrows, W, 3), dtype=torch.float32, device="cuda:0")
        prim = torch.zeros((nrows, W, 2), dtype=torch.int32, device="cuda:0")
        prim_t = torch.zeros((nrows, W), dtype=torch.float32, device="cuda:0")
        stream = torch.cuda.Stream("cuda:0")
        torch.cuda.synchronize()
        st = ctx.render_rows_device(opt, seed, row0, nrows, bgr.data_ptr(), rad.data_ptr(), prim.data_ptr(), prim_t.data_ptr(),
                                    stream=stream.cuda_stream, want_stats=want_stats, row_stride=stride, static=static)
        if not want_stats:
            st = ctx.last_call_stats()
        stream.synchronize()
        outs.append((bgr.cpu().numpy(), rad.cpu().numpy(), prim.cpu().numpy(), prim_t.cpu().numpy(), st))
    return outs


def _same(a, b):
    for k in range(3):
        assert np.array_equal(a[k], b[k]), ("bgr", "radiance", "prim")[k]
    assert np.array_equal(a[3], b[3], equal_nan=True)
    for k in ("queries", "rays_api", "nodes", "shadow_queries", "trace_queries"):
        assert a[4][k] == b[4][k], k


@pytest.mark.parametrize("accel", ["brute", "lbvh"])
def test_static_entry_point_matches_the_allocating_one(accel):
    from raytracing_amd import _native
    flat = flat_for("config2_cornell_light.json")
    ctx = _native.Context(0)
    try:
        ctx.set_accel(accel)
        ctx.upload(flat)
        # a frame that needs several bands of the 128 MiB array (640 x 480 x 3 samples at 5 bounces), with statistics
        big = dict(width=640, height=480, bounces=5, samples=3)
        ref, got = _both(ctx, big, 2, 0, 480, 1, True)
        _same(ref, got)
        assert got[4]["launches"] > ref[4]["launches"]               # more bands than the allocating call needed
        # then a smaller one, asynchronous, interleaved rows: nothing of the first call may show through
        small = dict(width=96, height=54, bounces=3, samples=1)
        ref, got = _both(ctx, small, 7, 1, 27, 2, False)
        _same(ref, got)
        # and the first again
        ref2, got2 = _both(ctx, big, 2, 0, 480, 1, True)
        _same(ref2, got2)
    finally:
        ctx.close()


def test_static_entry_point_limits_and_growing_trees():
    import torch
    from raytracing_amd import _native
    from test_gpu_parity import nested_glass_scene
    ctx = _native.Context(0)
    try:
        # trees that outgrow the optimistic sizing: the checked call halves its bands, same result as the allocating call
        ctx.upload(nested_glass_scene())
        opt = dict(width=24, height=16, bounces=11, samples=2)
        ref, got = _both(ctx, opt, 5, 0, 16, 1, True)
        _same(ref, got)
        # a row that cannot fit: 8192 pixels x 16 samples at 14 bounces is 190 MB of queues and node arrays for one row
        out = torch.zeros((1, 8192, 3), dtype=torch.uint8, device="cuda:0")
        with pytest.raises(_native.NativeError, match="static work array") as info:
            ctx.render_rows_device(dict(width=8192, height=8, bounces=14, samples=16), 0, 0, 1, out.data_ptr(), static=True)
        assert info.value.code == _native.E_CAPACITY
    finally:
        ctx.close()
