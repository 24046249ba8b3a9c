"""What one rank of an N-GPU run does per frame, timed on one GPU: rows r, r+N, ... of the benchmark frame
; usage: rank_emulation.py [workload] [fF] [bB] [a] [g] [N ...]; fF: F calls in flight, each with a context and a
stream of its own (default f1); bB: B frames per render call (t3d_opts.frames, default b1), bw: as many as there are
ranks, which is what bench.py does; g: every frame is followed by a gather in a one-rank RCCL group and the strided
copy into the frame, on the frame's stream (the host-side cost of the real gather; no bytes cross a link).
Shows the fixed per-frame cost that limits strong scaling."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _native
from raytracing_amd.workloads import get_flat_workload
args = sys.argv[1:]
workload = "cornell_1080p_b4"
if args and not args[0].isdigit() and not (args[0][0] == "f" and args[0][1:].isdigit()):
    workload = args.pop(0)
F = 1
if args and args[0].startswith("f"):
    F = int(args.pop(0)[1:])
B_arg = 1
if args and args[0].startswith("b") and (args[0][1:].isdigit() or args[0] == "bw"):
    a = args.pop(0)
    B_arg = 0 if a == "bw" else int(a[1:])
adaptive = bool(args) and args[0] == "a"      # a: keep the adaptive cut of small launches whatever F is
if adaptive:
    args.pop(0)
with_gather = bool(args) and args[0] == "g"
if with_gather:
    args.pop(0)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flat, opt = get_flat_workload(workload)
ctx = _native.Context(0); ctx.upload(flat)
more = [_native.Context(0) for _ in range(F - 1)]
for c in [ctx] + more:
    c.set_full_rounds(F >= 3 and not adaptive)
for c in more:
    c.upload(flat); c.set_timing(False)
more_streams = [torch.cuda.Stream(torch.device("cuda", 0)) for _ in range(F - 1)]
H, W = opt["height"], opt["width"]
dev = torch.device("cuda", 0)
torch_stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(torch_stream)
stream = torch_stream.cuda_stream
for n in ([int(a) for a in args] or [1, 2, 4, 8]):
    rows = H // n
    B = B_arg or n
    out = torch.empty((B, rows, W, 3), dtype=torch.uint8, device=dev)
    st = ctx.render_rows_device(opt, 0, 0, rows, out.data_ptr(), stream=stream, want_stats=True, row_stride=n, frames=B)
    ctx.set_timing(False)
    for _ in range(3):
        ctx.render_rows_device(opt, 0, 0, rows, out.data_ptr(), stream=stream, want_stats=False, row_stride=n, frames=B)
    torch.cuda.synchronize()
    chains = [(ctx, stream, out)] + [(c, s.cuda_stream, torch.empty_like(out)) for c, s in zip(more, more_streams)]
    if os.environ.get("RANK_EMU_OWN_STREAMS"):   # every chain on its context's own stream instead of a stream of the caller's
        chains = [(c, 0, o) for c, _, o in chains]
    K = 200 if workload == "cornell_1080p_b4" else 6
    tstreams = [torch_stream] + more_streams
    recv = [torch.empty_like(out) for _ in range(F)]
    frames = [torch.empty((B, rows, n, W, 3), dtype=torch.uint8, device=dev) for _ in range(F)]

    def one(i):
        c, s_, o_ = chains[i % F]
        c.render_rows_device(opt, 0, 0, rows, o_.data_ptr(), stream=s_, want_stats=False, row_stride=n, frames=B)
        if with_gather:
            torch.cuda.set_stream(tstreams[i % F])
            dist.gather(o_, [recv[i % F]], dst=0)
            frames[i % F][:, :, 0].copy_(recv[i % F])
    for i in range(5 * F):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        one(i)
    host_ms = (time.perf_counter() - t0) / (K * B) * 1e3
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch_stream)
    ms = (time.perf_counter() - t0) / (K * B) * 1e3
    ctx.set_timing(True)
    print(f"ranks {n} ({workload}): rows {rows}  {ms:.4f} ms/frame per rank ({F} calls in flight, {B} frames per call)  queries {st['queries']}  -> speedup bound {0:.2f}" if n == 1 else
          f"ranks {n} ({workload}): rows {rows}  {ms:.4f} ms/frame per rank ({F} calls in flight, {B} frames per call{', gather' if with_gather else ''}; host {host_ms:.4f} ms/frame)  queries {st['queries']}")
    print("   k_level ms per level", [round(x, 4) for x in st["level_ms"][:opt["bounces"] + 1]], "resolve+finish ms", round(st["ms_resolve"], 4),
          "rays per level", st["level_rays"][:opt["bounces"] + 1], "nodes", st["level_nodes"][:opt["bounces"] + 1])
