"""Frames in flight against bands per frame on one GPU: F chains (a stream each), every frame of a chain traced as `bands` bands of rows by
`bands` contexts at once (t3d_render_rows_device_split).  Prints ms per frame for a short (20 frames) and a long (200) timed region, both
started and ended with an idle device.  usage (GPU box): python tools/chains_and_bands.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from raytracing_amd import _native
    from raytracing_amd.workloads import get_flat_workload
    flat, opt = get_flat_workload("cornell_1080p_b4")
    H, W = opt["height"], opt["width"]
    for F, bands in ((4, 1), (3, 1), (2, 2), (3, 2), (2, 3), (1, 2), (1, 4)):
        chains = []
        for _ in range(F):
            ctxs = [_native.Context(0) for _ in range(bands)]
            for c in ctxs:
                c.set_full_rounds(F * bands >= 3)
                c.upload(flat)
                c.set_timing(False)
            chains.append((ctxs, torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0"), torch.cuda.Stream("cuda:0")))

        def step(k, seed, grow=False):
            ctxs, out, stream = chains[k % F]
            if bands == 1:
                ctxs[0].render_rows_device(opt, seed, 0, H, out.data_ptr(), stream=stream.cuda_stream, want_stats=grow)
            else:
                _native.render_rows_split(ctxs, opt, seed, 0, H, bgr_ptr=out.data_ptr(), stream=stream.cuda_stream)

        for k in range(F):
            step(k, k, grow=True)     # a checked call per chain: work buffers grown to what the trees need
        for k in range(12 * F):
            step(k, k)
        torch.cuda.synchronize()
        res = []
        for steps in (20, 20, 200):
            t0 = time.perf_counter()
            for k in range(steps):
                step(k, k)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
        for ctxs, _, _ in chains:
            if bands == 1:
                ctxs[0].last_call_stats()
            else:
                _native.split_call_stats(ctxs)
            for c in ctxs:
                c.close()
        print(f"{F} chains x {bands} bands: 20 frames {res[0]:.4f} / {res[1]:.4f} ms per frame, 200 frames {res[2]:.4f}", flush=True)


if __name__ == "__main__":
    main()
