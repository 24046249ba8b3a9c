"""Post.upscale / Post.denoise on the GPU (through the C-ABI) against the host mirror of the reference's Post class
(T3DR.py:1600-1647), which the CPU suite pins to the reference's own outputs, and against those outputs directly."""
import numpy as np
import pytest

from raytracing_amd import _native
from raytracing_amd.post import Post

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _host_denoise(img, k):
    return np.asarray(Post.denoise(img, k), dtype=np.uint8).reshape(img.shape)


@pytest.mark.parametrize("shape,factor", [((5, 7, 3), 1), ((5, 7, 3), 2), ((33, 65, 3), 3), ((9, 4, 1), 4), ((17, 300, 4), 5),
                                          ((1, 1, 3), 7), ((270, 480, 3), 4), ((6, 6, 3), 2), ((3, 5, 2), 3),
                                          ((2, 1000, 3), 16), ((3, 4, 20), 2)])
def test_upscale_matches_host(ctx, shape, factor):
    rng = np.random.default_rng(shape[0] * 131 + factor)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = ctx.post_upscale(img, factor)
    assert got.dtype == np.uint8 and got.shape == (shape[0] * factor, shape[1] * factor, shape[2])
    assert np.array_equal(got, Post.upscale(img, factor))
    assert np.array_equal(Post.upscale(img, factor, context=ctx), got)


@pytest.mark.parametrize("shape,k", [((8, 8), 0), ((8, 8), 1), ((9, 11), 2), ((9, 11), 3), ((40, 70), 5), ((33, 17), 4),
                                     ((64, 48), 7), ((50, 37), 9), ((31, 45), 15), ((16, 16), 3), ((17, 33), 3)])
def test_denoise_matches_host(ctx, shape, k):
    rng = np.random.default_rng(shape[0] * 977 + k)
    img = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    if k == 5:
        img[::3, ::2] = 255          # many equal values: ties in the selection
        img[5:9, 5:30] = 0
    got = ctx.post_denoise(img, k)
    assert np.array_equal(got, _host_denoise(img, k))
    assert np.array_equal(np.asarray(Post.denoise(img, k, context=ctx), dtype=np.uint8).reshape(img.shape), got)


def test_denoise_image_smaller_than_window_is_all_zero(ctx):
    img = np.full((4, 9, 3), 200, np.uint8)
    assert not ctx.post_denoise(img, 9).any()
    assert not ctx.post_denoise(np.full((9, 4, 3), 200, np.uint8), 9).any()
    # exactly one pixel has a full window
    img = np.arange(5 * 5 * 3, dtype=np.uint8).reshape(5, 5, 3)
    got = ctx.post_denoise(img, 5)
    assert np.array_equal(got, _host_denoise(img, 5)) and got[2, 2].any() and np.count_nonzero(got.any(axis=2)) == 1


def test_reference_known_answers(ctx, kat):
    """The reference's own outputs for its Post filters (tests/golden/kat.json, written by tests/golden/make_kat.py)."""
    k = kat["post"]
    img = np.asarray(k["image"], dtype=np.uint8)
    assert np.array_equal(ctx.post_upscale(img, 2), np.asarray(k["upscaled_x2"], dtype=np.uint8))
    assert np.array_equal(ctx.post_denoise(img, 3), np.asarray(k["denoised_k3"], dtype=np.uint8).reshape(img.shape))


def test_frame_sized_image_and_device_pointers(ctx):
    """A 1080p frame, filtered in place on the device (torch only holds the memory): denoise then upscale by 2."""
    import torch
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    d_in = torch.from_numpy(img).cuda()
    d_mid = torch.empty_like(d_in)
    d_out = torch.empty((2160, 3840, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.post_denoise_device(d_in.data_ptr(), 1080, 1920, 3, d_mid.data_ptr(), stream)
    ctx.post_upscale_device(d_mid.data_ptr(), 1080, 1920, 3, 2, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
    mid = d_mid.cpu().numpy()
    rows = slice(500, 540)           # the host mirror sorts every window: a band of rows is enough
    want = _host_denoise(img[rows.start - 1:rows.stop + 1], 3)[1:-1]
    assert np.array_equal(mid[rows], want)
    assert not mid[0].any() and not mid[-1].any() and not mid[:, 0].any() and not mid[:, -1].any()
    assert np.array_equal(d_out.cpu().numpy(), np.repeat(np.repeat(mid, 2, axis=0), 2, axis=1))


def test_upscale_store_width_follows_alignment(ctx):
    """The kernel writes 16, 4 or 1 bytes per lane depending on the output's row length and base address."""
    import torch
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(6, 16, 4), dtype=np.uint8)
    want = Post.upscale(img, 2)
    d_in = torch.from_numpy(img).cuda()
    for offset in (0, 4, 1):
        raw = torch.zeros(want.size + 16, dtype=torch.uint8, device="cuda")
        ctx.post_upscale_device(d_in.data_ptr(), 6, 16, 4, 2, raw.data_ptr() + offset, 0)
        torch.cuda.synchronize()
        host = raw.cpu().numpy()
        assert np.array_equal(host[offset:offset + want.size].reshape(want.shape), want)
        assert not host[:offset].any() and not host[offset + want.size:].any()


def test_bad_arguments(ctx):
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(_native.NativeError) as e:
        ctx.post_denoise(img, 17)
    assert e.value.code == _native.E_CAPACITY
    with pytest.raises(_native.NativeError) as e:
        ctx.post_upscale(img, 0)
    assert e.value.code == -1
    with pytest.raises(_native.NativeError) as e:
        ctx.post_denoise_device(0, 8, 8, 3, 0)
    assert e.value.code == -1
    with pytest.raises(_native.NativeError) as e:
        ctx.post_denoise(img, -1)
    assert e.value.code == -1
