"""A render loop over a large unchanged scene with t3d_set_background_tree: when the better tree arrives and what it is worth.
usage (GPU box): python tools/background_tree.py [workload ...]   (default: synthetic_200k_1080p_b2 synthetic_1m_4k_b2)
Prints the upload time with the option off and on, the frame at which the surface-area tree was adopted (and what that
call cost), and the frame time before and after."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from raytracing_amd import _native
    from raytracing_amd.workloads import get_flat_workload
    names = sys.argv[1:] or ["synthetic_200k_1080p_b2", "synthetic_1m_4k_b2"]
    for name in names:
        flat, opt = get_flat_workload(name)
        out = torch.empty((opt["height"], opt["width"], 3), dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.Stream("cuda:0")
        ctx = _native.Context(0)
        ctx.upload(flat)          # the first upload of a process pays for the library's start
        uploads = {}
        for on in (False, True):
            ctx.set_background_tree(on)
            t0 = time.perf_counter()
            ctx.upload(flat)
            uploads[on] = (time.perf_counter() - t0) * 1e3
        t_up = time.perf_counter()
        frames = []
        adopted_at = None
        for k in range(400):
            t0 = time.perf_counter()
            ctx.render_rows_device(opt, k, 0, opt["height"], out.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            frames.append((time.perf_counter() - t0) * 1e3)
            if adopted_at is None:
                t0 = time.perf_counter()
                status = ctx.tree_status()      # (adopts a finished tree, as the next render call would)
                if status == "background":
                    adopted_at = (k + 1, (time.perf_counter() - t_up) * 1e3, (time.perf_counter() - t0) * 1e3)
            if adopted_at is not None and k >= adopted_at[0] + 12:
                break
        ctx.last_call_stats()
        ctx.close()
        if adopted_at is None:
            print(f"{name}: no tree adopted within {len(frames)} frames")
            continue
        k = adopted_at[0]
        before = sorted(frames[max(2, k - 8):k]) or [float("nan")]
        after = sorted(frames[k + 1:])
        print(f"{name}: upload {uploads[False]:.0f} ms, with the background build started {uploads[True]:.0f} ms; "
              f"tree adopted before frame {k}, {adopted_at[1]:.0f} ms after the upload returned (the swap: {adopted_at[2]:.1f} ms); "
              f"ms/frame before {before[len(before) // 2]:.2f}, after {after[len(after) // 2]:.2f}")


if __name__ == "__main__":
    main()
