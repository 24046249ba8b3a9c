"""Time the GPU Post filters on a 1080p frame held in device memory (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import _native

ctx = _native.Context(0)
H, W = 1080, 1920
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).cuda()
out = torch.empty_like(img)
side = torch.cuda.Stream()           # the default stream's handle is 0, which the library reads as "the context's own stream"
torch.cuda.set_stream(side)
stream = side.cuda_stream


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for k in (3, 5, 7, 9, 15):
    us = timed(lambda: ctx.post_denoise_device(img.data_ptr(), H, W, k, out.data_ptr(), stream))
    print(f"denoise k={k}: {us:8.1f} us  ({H * W / us:8.1f} Mpixel/s)")
for f in (2, 4):
    big = torch.empty((H * f, W * f, 3), dtype=torch.uint8, device="cuda")
    us = timed(lambda: ctx.post_upscale_device(img.data_ptr(), H, W, 3, f, big.data_ptr(), stream))
    gb = (H * W * 3 * (1 + f * f)) / 1e9
    print(f"upscale x{f}: {us:8.1f} us  ({gb / (us * 1e-6):8.1f} GB/s algorithmic)")
