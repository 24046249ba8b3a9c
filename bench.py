#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (closest-hit queries per second) and frame time, Cornell box 1920x1080,
4 bounces, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one complete frame: every rank traces its interleaved image rows through libt3d_hip.so
(scene resident in HBM, output tile in HBM), then the tiles are gathered on rank 0 (RCCL).  A "ray" is
one closest-hit query that is traced (primary + bounce + transmit + shadow); shadow rays whose light term is zero
whatever they hit are answered without a ray and not counted (value_in_reference_queries counts them, as the
reference does).  Consecutive steps are independent frames, and
several are in flight at a time (--frames-in-flight, config.frames_in_flight), each a chain of its own: a
context, a stream, a tile buffer; trace, then gather, on that stream.  With several ranks one render call carries a
rank's rows of several consecutive frames (--frames-per-call, config.frames_per_call; t3d_opts.frames), so that a
rank's launches stay large enough to fill its GPU, and their tiles travel in one gather: every frame is still split
over all ranks, traced and gathered.  Prints ONE JSON line on rank 0.

The line also carries
  roofline      FP32 vector-ALU roofline of the dominant kernel (k_level, which runs every closest-hit
                query of a tree level): algorithmic flops (51 per ray-triangle test, 19 per ray-sphere
                test) divided by the kernel's launch durations, measured with HIP events on one more
                frame after the timed steps, alone on the device
  pipeline      the same flops over the time a step takes with the frames in flight, and the latency of
                one frame alone
  cpu_baseline  the float64 CPU oracle (oracle/, a restatement of the reference's algorithm) timed on this
                host's cores on a bounded sample of the same frame (N = 1 only)
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TRI_TEST = 51.0     # Moeller-Trumbore with on-the-fly edges
FLOP_PER_SPHERE_TEST = 19.0
FP32_VALU_PEAK_TFLOPS = 157.3  # 256 CU x 128 FMA lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
WARM_CALLS_PER_CHAIN = int(os.environ.get("T3D_BENCH_WARM_CALLS", "20"))    # untimed render calls per chain in front of every timed region (at least; see timed())


def source_digest():
    """Fingerprint of the tracing kernels' sources: says which build a counter run belongs to (there is no git on the GPU box)."""
    import hashlib
    h = hashlib.sha1()
    for name in ("t3d_hip.hip", "t3d_lbvh.h", "t3d_host_tree.h"):
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def pmc_traffic(workload):
    """Hardware counters of one tree level's launches from the newest committed counter run (profiles/*_pmc_traffic.json,
    made by tools/gpu_pmc.sh + tools/pmc_summary.py): HBM bytes, executed vector-ALU wave instructions, active lanes per
    instruction; None when there is none for this workload.  Counters cannot be read from inside this process, so these
    are the roofline fields that are not measured live.  A run made from other kernel sources than the ones in the tree
    (source_digest) is stale: it is named, and its numbers are not used."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload and d.get("kernel") == "k_level":
            best = (d, os.path.basename(path))
    if best is None:
        return None
    stale = best[0].get("sources") != source_digest()
    return best[0], best[1], stale


def cpu_baseline(flat, options, seed, row_step=1, nthreads=None):
    """Oracle timed on host threads: every `row_step`-th image row of the same frame, rows dealt
    round-robin to the threads (ctypes releases the GIL during the C call)."""
    from oracle import oracle
    oracle.lib()
    nthreads = nthreads or max(1, min(16, os.cpu_count() or 1))
    H = options["height"]
    if row_step <= 0:
        row_step = max(1, H // nthreads)   # one row per thread
    rows = list(range(0, H, row_step))
    stride = row_step * nthreads
    results = [None] * nthreads

    def work(t):
        first = t * row_step
        if first < H:
            results[t] = oracle.render(flat, options, seed=seed, row0=first, row_stride=stride, want_radiance=False)["stats"]

    t0 = time.time()
    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.time() - t0
    queries = sum(r["queries"] for r in results if r)
    return {"value": queries / dt / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "port",
            "sample": f"{len(rows)} of {H} rows of the same frame (row step {row_step}), {queries} queries in {dt:.2f} s on {nthreads} threads",
            "seconds": round(dt, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cornell_1080p_b4")
    ap.add_argument("--partition", default="interleaved", choices=["interleaved", "bands"])
    ap.add_argument("--accel", default="auto", choices=["auto", "brute", "lbvh"], help="closest-hit structure (auto: by scene size)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent render calls traced at the same time, each with a context and a stream of its own "
                         "(0: one for LBVH scenes, four for brute-force scenes: as many as the process has hardware queues)")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="consecutive frames whose rows one render call traces together, and one gather collects "
                         "(0: as many as there are ranks, so that a call traces a whole frame's worth of rows whatever their number)")
    ap.add_argument("--fixed-seed", action="store_true",
                    help="the timed steps render the same frame(s) over and over (every call starts at seed 0) instead of a new "
                         "frame per step (frame g of the run has seed g)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary timings (the other seed mode, one frame per call, the latency of a frame)")
    ap.add_argument("--background-tree", action="store_true",
                    help="LBVH scenes above the host-tree limit: a host thread cuts the surface-area tree while the first (untimed) "
                         "frames are traced through the device's, and the clock starts after it has been adopted (t3d_set_background_tree)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (RCCL over xGMI, one GPU per rank) or gloo: tiles gathered through host memory, and ranks share the "
                         "visible GPUs when there are fewer of them than ranks -- a rehearsal of the several-rank path on a small box")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="with one rank: create a process group of that one rank all the same (backend as --dist-backend) and send every "
                         "frame through its gather and the strided copy -- the several-rank code path on a one-GPU box")
    ap.add_argument("--cpu-row-step", type=int, default=0, help="CPU baseline renders every n-th row (0: every row for the headline workload, one row per thread otherwise)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    grouped = world > 1 or args.one_rank_group
    gloo = grouped and args.dist_backend == "gloo"
    if gloo:
        local_rank %= max(1, torch.cuda.device_count())
    if world > 1:
        torch.cuda.set_device(local_rank)
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus != 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    elif args.one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    comm_device = torch.device("cpu") if gloo else device

    from raytracing_amd import _native
    from raytracing_amd.tiles import gather_frame, partition_rows, sum_counters
    from raytracing_amd.workloads import get_flat_workload

    flat, options = get_flat_workload(args.workload)
    # Consecutive frames are independent, and the launches of one 1080p frame are too small to keep the device full
    # through their starts and ends: F render calls are in flight, each a chain of its own (context, stream, tile
    # buffer; trace, then gather, on that stream) with nothing that orders one chain against another.  Every step is
    # still one complete frame, traced and gathered, and every frame of the run is a different one: frame g has seed g
    # (--fixed-seed: every call starts again at seed 0).
    lbvh_scene = args.accel == "lbvh" or (args.accel == "auto" and flat.n_tris + flat.n_spheres > _native.BRUTE_FORCE_MAX_PRIMS)
    # one rank's share of a frame is a chain of small launches: more of them in flight, and full rounds (no slot of a
    # round wasted; the other chains hide the latency the adaptive cut is there to shorten)
    # (four: one chain per hardware queue of the process; a fifth shares a queue and costs 9 %, three leave 2.5 % unused)
    F = args.frames_in_flight if args.frames_in_flight > 0 else (1 if lbvh_scene else 4)
    # One rank's share of a frame shrinks with the number of ranks, and launches that small leave most of a GPU idle:
    # a call traces the rank's rows of B consecutive frames together (t3d_opts.frames) and one gather moves their tiles.
    B = args.frames_per_call if args.frames_per_call > 0 else world
    H, W = options["height"], options["width"]
    tile = partition_rows(H, world, rank, args.partition)
    ctxs, outs, streams = [], [], []
    for _ in range(F):
        c = _native.Context(local_rank)
        c.set_accel(args.accel)
        c.set_full_rounds(F >= 3)
        c.set_background_tree(args.background_tree)
        c.upload(flat)
        ctxs.append(c)
        outs.append(torch.empty((B, tile.nrows, W, 3), dtype=torch.uint8, device=device))
        # explicit streams: a null handle would mean "the context's own stream" to the library, and the gather that
        # follows every render has to be ordered behind it
        streams.append(torch.cuda.Stream(device))
    ctx = ctxs[0]

    class Run:
        """A sequence of render calls dealt to the chains in turn; remembers what every chain rendered last."""

        def __init__(self, vary):
            self.vary, self.calls, self.next_frame, self.last, self.frames = vary, 0, 0, [None] * F, [None] * F

        def step(self, frames=B, want_stats=False):
            """One render call on the next chain: this rank's rows of `frames` consecutive frames, then their gather."""
            k = self.calls % F
            self.calls += 1
            seed0 = self.next_frame if self.vary else 0
            self.next_frame += frames
            torch.cuda.set_stream(streams[k])
            out = outs[k][:frames] if B > 1 else outs[k][0]
            st = ctxs[k].render_rows_device(options, seed0, tile.row0, tile.nrows, out.data_ptr(), stream=streams[k].cuda_stream,
                                            want_stats=want_stats, row_stride=tile.stride, frames=frames, frame_seed_step=1)
            if gloo:
                streams[k].synchronize()
                self.frames[k] = gather_frame(out.cpu(), H, args.partition, slot=k, collective_always=grouped)
            else:
                self.frames[k] = gather_frame(out, H, args.partition, slot=k, collective_always=grouped)
            self.last[k] = (seed0, frames)
            return st, (seed0, frames)

    def sync():
        torch.cuda.synchronize(device)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(vary, steps, per_call, warm_calls):
        """`steps` frames in calls of `per_call` frames (and one smaller call for the remainder), bracketed by a barrier and
        a device synchronisation on both sides -> (seconds on this rank, the calls made, the run)."""
        run = Run(vary)
        full_calls, rest = divmod(steps, per_call)
        # untimed calls: at least as many as every chain needs to reach its steady state (a band is replayed as a launch
        # graph from its third request with the same counter set, i.e. a chain's fifth call, and the device needs some
        # 25 ms of work before a short timed region runs at the speed of a long one: 20 timed frames of the headline
        # workload measured 0.349 ms per frame after 5 calls per chain, 0.339 after 10, 0.325 after 20, 0.329 after 40)
        for _ in range(warm_calls):
            run.step(per_call)
        if rest:
            for _ in range(WARM_CALLS_PER_CHAIN * F):
                run.step(rest)
        sync()
        made = []
        t0 = time.perf_counter()
        for _ in range(full_calls):
            made.append(run.step(per_call)[1])
        if rest:
            made.append(run.step(rest)[1])
        sync()
        return time.perf_counter() - t0, made, run

    def check_chains(run):
        """The timed calls were asynchronous and checked nothing: every chain's last call must have fitted its buffers
        (raises on a capacity overflow), and its tiles -- left by replayed, concurrent calls -- must equal the same frames
        rendered one by one, alone on the device, by a checked call; so must the number of queries."""
        alone = torch.empty((tile.nrows, W, 3), dtype=torch.uint8, device=device)
        torch.cuda.set_stream(streams[0])
        chain_stats = [c.last_call_stats() for c in ctxs]
        for k in range(F):
            if run.last[k] is None:
                continue
            seed0, frames = run.last[k]
            queries = 0
            for b in range(frames):
                st = ctx.render_rows_device(options, seed0 + b, tile.row0, tile.nrows, alone.data_ptr(), stream=streams[0].cuda_stream,
                                            want_stats=True, row_stride=tile.stride)
                queries += st["queries"]
                if not torch.equal(outs[k][b], alone):
                    raise RuntimeError(f"the tile of chain {k}, frame {b} of its last call (seed {seed0 + b}), differs from the tile rendered alone")
            # the frame(s) the gather assembled on rank 0 hold this rank's rows where they belong
            got = run.frames[k]
            if got is not None and rank == 0:
                got = got.reshape((frames, H, W, 3))
                mine_rows = got[:, tile.row0:tile.row0 + tile.nrows * tile.stride:tile.stride]
                want_rows = outs[k][:frames]
                if not torch.equal(mine_rows.to(want_rows.device), want_rows):
                    raise RuntimeError(f"chain {k}: the gathered frame does not hold rank 0's rows of its last call")
            if chain_stats[k]["queries"] != queries:
                raise RuntimeError(f"chain {k}: the last timed call traced {chain_stats[k]['queries']} queries, the same frames rendered alone {queries}")

    keys = ["queries", "rays_api", "tri_tests", "sphere_tests", "aabb_tests", "shadow_queries", "trace_queries", "nodes", "shadow_culled"]

    def count(made):
        """Exact counters of the timed calls on this rank: every one of them rendered again, waited for, with statistics."""
        tot = dict.fromkeys(keys, 0)
        torch.cuda.set_stream(streams[0])
        seen = {}
        for call in made:
            if call not in seen:
                seed0, frames = call
                out = outs[0][:frames] if B > 1 else outs[0][0]
                seen[call] = ctx.render_rows_device(options, seed0, tile.row0, tile.nrows, out.data_ptr(), stream=streams[0].cuda_stream,
                                                    want_stats=True, row_stride=tile.stride, frames=frames, frame_seed_step=1)
            for k in keys:
                tot[k] += seen[call][k]
        return tot

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=comm_device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # checked calls, one per chain: work buffers grown to what this scene's trace trees need
    grow = Run(False)
    for _ in range(F):
        grow.step(B, want_stats=True)
    # where the hierarchy of an LBVH scene comes from; a tree still being cut on a host thread is waited for and adopted here
    hierarchy = sorted({c.tree_status(wait=True) for c in ctxs})
    for c in ctxs:
        c.set_timing(False)
    vary = not args.fixed_seed
    warm = max((args.warmup + B - 1) // B - 1, WARM_CALLS_PER_CHAIN * F)
    # the timed region: exactly args.steps frames
    elapsed, made, run = timed(vary, args.steps, B, warm)
    check_chains(run)
    mine = count(made)
    elapsed = max_over_ranks(elapsed)
    secondary = {}
    if not args.no_secondary:
        # the other seed mode (--fixed-seed: the same frame(s) over and over, every call a launch graph replayed unchanged)
        e2, _, run2 = timed(not vary, args.steps, B, warm)
        check_chains(run2)
        secondary["other_seed_mode_s"] = max_over_ranks(e2)
        if B > 1:
            # one frame per call: what a frame costs when it is not traced together with its successors
            e1, _, run1 = timed(vary, args.steps, 1, WARM_CALLS_PER_CHAIN * F)
            check_chains(run1)
            secondary["one_frame_per_call_s"] = max_over_ranks(e1)
        if world > 1:
            # one frame at a time, nothing else in flight, every frame waited for by every rank: the latency of a frame that
            # is split over the ranks, gather included
            n_lat = min(args.steps, 20)
            lat_run = Run(vary)
            for _ in range(3):
                lat_run.step(1)
            sync()
            t0 = time.perf_counter()
            for _ in range(n_lat):
                lat_run.step(1)
                sync()
            secondary["frame_latency_s"] = max_over_ranks((time.perf_counter() - t0) / n_lat)
    # the same frame as two bands of rows traced at once by two contexts (t3d_render_rows_device_split): what a caller
    # with ONE frame to render gets; timed by events on the stream the bands are forked from and joined to
    two_bands_ms = None
    if not lbvh_scene and not args.no_secondary and tile.nrows >= 16:
        pair = ctxs[:2] if F >= 2 else [ctx, _native.Context(local_rank)]
        if F < 2:
            pair[1].upload(flat)
        for c in pair:
            c.set_timing(False)
            c.set_full_rounds(False)
        torch.cuda.set_stream(streams[0])
        alone = torch.empty((tile.nrows, W, 3), dtype=torch.uint8, device=device)
        ctx.render_rows_device(options, 0, tile.row0, tile.nrows, alone.data_ptr(), stream=streams[0].cuda_stream, want_stats=True,
                               row_stride=tile.stride)
        banded = torch.empty_like(alone)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for i in range(8 + 21):
            e0.record(streams[0])
            _native.render_rows_split(pair, options, 0, tile.row0, tile.nrows, bgr_ptr=banded.data_ptr(), stream=streams[0].cuda_stream,
                                      row_stride=tile.stride)
            e1.record(streams[0])
            streams[0].synchronize()
            if i >= 8:
                times.append(e0.elapsed_time(e1))
        _native.split_call_stats(pair)          # raises if a band overflowed its buffers
        if not torch.equal(banded, alone):
            raise RuntimeError("the frame traced as two bands differs from the frame traced alone")
        two_bands_ms = sorted(times)[len(times) // 2]
    # kernel durations (HIP events on the launch stream) come from one more frame, outside the timed region and alone
    # on the device: events between the launches are stops in the stream, and kernels that share the device with
    # another frame's have no duration of their own
    sync()
    ctx.set_timing(True)
    torch.cuda.set_stream(streams[0])
    alone = torch.empty((tile.nrows, W, 3), dtype=torch.uint8, device=device)
    last = ctx.render_rows_device(options, 0, tile.row0, tile.nrows, alone.data_ptr(), stream=streams[0].cuda_stream,
                                  want_stats=True, row_stride=tile.stride)
    sync()
    whole_alone_ms = None
    if world > 1 and not args.no_secondary:
        # the whole frame on ONE GPU (rank 0's), alone, while the other ranks wait: what the several-rank figures compare with
        if rank == 0:
            whole = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
            ctx.set_timing(False)
            for _ in range(3):
                st = ctx.render_rows_device(options, 0, 0, H, whole.data_ptr(), stream=streams[0].cuda_stream, want_stats=True)
            whole_alone_ms = st["ms_device"]
            ctx.set_timing(True)
        sync()

    totals = dict(zip(keys, sum_counters([mine[k] for k in keys], device=comm_device)))

    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed / steps * 1e3
        # only queries that were traced count: shadow rays whose light term is zero whatever they hit are answered without a
        # traversal (t3d_stats.shadow_culled); the reference traces and counts them, they are reported beside the value
        traced = totals["queries"] - totals["shadow_culled"]          # of all the timed frames, all ranks
        last_traced = last["queries"] - last["shadow_culled"]         # of this rank's share of the frame with seed 0
        value = traced / elapsed / 1e6
        # roofline of the dominant kernel on this rank (k_level: every closest-hit query of the frame):
        # algorithmic flops / measured kernel time
        T, S = flat.n_tris, flat.n_spheres
        flop_per_query = FLOP_PER_TRI_TEST * T + FLOP_PER_SPHERE_TEST * S
        flops = last_traced * flop_per_query
        level_s = last["ms_k_level"] * 1e-3
        achieved = flops / level_s / 1e12 if level_s > 0 else 0.0
        frame_flops = traced / steps * flop_per_query                 # an average timed frame, all ranks
        line = {
            "metric": "Mrays/s (primary+secondary closest-hit queries), " +
                      ("Cornell 1080p 4 bounces" if args.workload == "cornell_1080p_b4" else args.workload),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            # the frame is fixed, the ranks split it; with B > 1 a render call traces a rank's rows of B consecutive frames
            # together, so the figure is the throughput of batches -- the latency of one frame is in `single_frame`
            "scaling": f"strong (batched, {B} frames per call)" if B > 1 else "strong", "vs_baseline": None,
            # the same frames per second in the reference's count of rays (it traces, and counts, the shadow rays that this
            # library answers without tracing): the figure to set against cpu_baseline, whose oracle traces them all
            "value_in_reference_queries": round(totals["queries"] / elapsed / 1e6, 2),
            "dtype": "f32", "data": "synthetic",
            "seeds": ("every timed frame is a different one: frame g of the run has seed g" if vary else
                      "every call starts at seed 0: the same frame(s) over and over"),
            "config": {"workload": args.workload, "ranks": world, "collective": ("gloo gather through host memory (rehearsal)" if gloo else
                                                                               ("RCCL gather" if world > 1 else
                                                                                ("RCCL gather in a group of one rank" if grouped else "none"))),
                       "width": W, "height": H, "bounces": options["bounces"],
                       "samples": options["samples"], "triangles": T, "spheres_and_lights": S,
                       "lights": int(len(flat.light_strength)), "partition": f"{args.partition} rows x{world}",
                       "frames_in_flight": F * B, "calls_in_flight": F, "frames_per_call": B,
                       # none: brute force; device: Morton order; host: surface area, cut during the upload; background: the
                       # same, cut on a host thread during the first untimed frames (--background-tree)
                       "hierarchy": "/".join(hierarchy),
                       "untimed_calls_per_chain": max(warm // F, WARM_CALLS_PER_CHAIN),
                       # averages over the timed frames (exact: every timed call was rendered again with statistics)
                       "queries_per_frame": round(traced / steps, 1),
                       "shadow_rays_answered_untraced_per_frame": round(totals["shadow_culled"] / steps, 1),
                       "reference_queries_per_frame": round(totals["queries"] / steps, 1),
                       "rendered_rays_per_frame": round(totals["rays_api"] / steps, 1)},
            "roofline": {"bound": "valu", "kernel": "k_level", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4), "traffic": None,
                         # frac counts the work a brute-force Moeller-Trumbore loop would do for these queries, not the
                         # instructions this kernel executes (plane grouping and whole-wave skips do the same tests in fewer)
                         "flops": "algorithmic: 51 per ray-triangle test, 19 per ray-sphere test; not executed instructions",
                         "launches_per_frame": last["n_k_level"],
                         "avg_launch_ms": round(last["ms_k_level"] / max(1, last["n_k_level"]), 5),
                         "flops_per_launch": flops / max(1, last["n_k_level"]),
                         "queries_per_launch": last_traced / max(1, last["n_k_level"]),
                         "frame_device_ms": round(last["ms_device"], 4), "resolve_ms": round(last["ms_resolve"], 4),
                         "measured_on": "the frame with seed 0, alone on the device, after the timed steps",
                         # the same launches priced by the queries the reference counts (the shadow rays answered without a ray
                         # included): comparable with figures that predate the culling of those rays
                         "frac_in_reference_queries": round(last["queries"] * flop_per_query
                                                            / level_s / 1e12 / FP32_VALU_PEAK_TFLOPS, 4) if level_s > 0 else None},
            # the whole pipeline as timed: every flop of a frame over the time a frame takes with F calls in flight
            "pipeline": {"frames_in_flight": F * B, "frame_latency_ms": round(last["ms_device"], 4),
                         # one frame alone on the device again, its rows as two bands traced at once by two contexts
                         "frame_latency_two_bands_ms": round(two_bands_ms, 4) if two_bands_ms else None,
                         # this rank's share of one frame with nothing else on the device, as a rate
                         "frame_alone_mrays": round(last_traced / (last["ms_device"] * 1e-3) / 1e6, 2) if last["ms_device"] > 0 else None,
                         "achieved": round(frame_flops / (ms_per_step * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                         "frac": round(frame_flops / (ms_per_step * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS / world, 4)},
        }
        if "other_seed_mode_s" in secondary:
            other = round(secondary["other_seed_mode_s"] / steps * 1e3, 4)
            line["ms_per_step_varying_seed"] = round(ms_per_step, 4) if vary else other
            line["ms_per_step_fixed_seed"] = other if vary else round(ms_per_step, 4)
        if world > 1 and secondary:
            # what the batched headline hides: the cost of a frame that is not traced together with its successors
            sf = {"frames_per_call": 1}
            if "one_frame_per_call_s" in secondary:
                sf["ms_per_frame"] = round(secondary["one_frame_per_call_s"] / steps * 1e3, 4)
                sf["calls_in_flight"] = F
            if "frame_latency_s" in secondary:
                sf["frame_latency_ms"] = round(secondary["frame_latency_s"] * 1e3, 4)
                sf["frame_latency_is"] = "one frame at a time: every rank's rows, the gather, a barrier; nothing else in flight"
            if whole_alone_ms:
                sf["one_gpu_whole_frame_alone_ms"] = round(whole_alone_ms, 4)
                sf["measured_on"] = "rank 0's GPU in this run, the other ranks waiting"
                if sf.get("ms_per_frame"):
                    sf["speedup_vs_one_gpu_frame_alone"] = round(whole_alone_ms / sf["ms_per_frame"], 2)
                if sf.get("frame_latency_ms"):
                    sf["latency_speedup_vs_one_gpu_frame_alone"] = round(whole_alone_ms / sf["frame_latency_ms"], 2)
                line["batched_speedup_vs_one_gpu_frame_alone"] = round(whole_alone_ms / ms_per_step, 2)
            line["single_frame"] = sf
        tr = pmc_traffic(args.workload) if world == 1 else None

        def counters_into(roof):
            """Counter-derived fields (from the committed run, see pmc_traffic) next to the live ones."""
            if not tr:
                return
            d, name, stale = tr
            roof["traffic_source"] = f"profiles/{name} (commit {d.get('commit')}, {d.get('date')}, kernel sources {d.get('sources')})"
            if stale:
                # counters of another build say nothing about this one
                roof["stale"] = True
                roof["traffic_source"] += f"; STALE: the kernel sources in the tree are {source_digest()}, the counters are not used"
                return
            roof["traffic"] = d["hbm_bytes_per_launch"]
            if d.get("valu_wave_insts_per_launch") and roof.get("avg_launch_ms"):
                # executed vector-ALU instructions against the issue rate of the chip: one wave instruction per SIMD every 4 cycles
                issue_peak = 1024 * 2.4e9 / 4
                rate = d["valu_wave_insts_per_launch"] / (roof["avg_launch_ms"] * 1e-3)
                roof["valu_issue_frac"] = round(rate / issue_peak, 4)
                roof["valu_wave_insts_per_launch"] = d["valu_wave_insts_per_launch"]
            if d.get("valu_active_lanes"):
                roof["valu_active_lanes_of_64"] = d["valu_active_lanes"]

        counters_into(line["roofline"])
        if last["aabb_tests"] > 0:
            # LBVH traversal.  The records a query visits (64 B per node record: four quantised boxes and their links; 64 B
            # per triangle or sphere test: one leaf record) are mostly served by the caches -- their rate can exceed the
            # HBM peak, so HBM is not the roof.  What bounds the kernels is vector-ALU issue at part-filled waves: the
            # roofline is executed wave instructions per second against the chip's issue rate (counters of a committed
            # run, duration live), with the memory side reported beside it.
            nbytes = 64.0 * (last["aabb_tests"] / 4) + 64.0 * last["tri_tests"] + 64.0 * last["sphere_tests"]
            # the byte count of the formula "32 B per node visited, 36 B per triangle test, 16 B per sphere test" (a binary
            # hierarchy of 32-byte nodes over 36-byte triangles): a record visited here holds four boxes
            survey_bytes = 32.0 * (last["aabb_tests"] / 4) + 36.0 * last["tri_tests"] + 16.0 * last["sphere_tests"]
            gbs = nbytes / level_s / 1e9 if level_s > 0 else 0.0
            n_launch = max(1, last["n_k_level"])
            roof = {"bound": "valu-issue", "kernel": "k_bvh_trace + k_bvh_process + k_bvh_shadow (one tree level)",
                    "achieved": None, "peak": round(1024 * 2.4e9 / 4 / 1e9, 1), "unit": "G wave-instructions/s", "frac": None, "traffic": None,
                    "launches_per_frame": last["n_k_level"], "avg_launch_ms": round(last["ms_k_level"] / n_launch, 5),
                    "nodes_visited_per_query": round(last["aabb_tests"] / 4 / max(1, last_traced), 2),
                    "tri_tests_per_query": round(last["tri_tests"] / max(1, last_traced), 2),
                    "deep_stack_pushes": last["bvh_deep_pushes"],
                    "frame_device_ms": round(last["ms_device"], 4), "resolve_ms": round(last["ms_resolve"], 4),
                    "memory": {"algorithmic_bytes_per_launch": nbytes / n_launch, "algorithmic_gbs": round(gbs, 1),
                               "survey_bytes_per_frame": survey_bytes,
                               "survey_gbs": round(survey_bytes / level_s / 1e9, 1) if level_s > 0 else None,
                               "survey_frac_of_hbm_peak": round(survey_bytes / level_s / 1e9 / HBM_PEAK_GBS, 4) if level_s > 0 else None,
                               "hbm_peak_gbs": HBM_PEAK_GBS, "note": "record bytes requested by the lanes; most are cache hits"}}
            counters_into(roof)
            if roof.get("valu_issue_frac") is not None:
                roof["frac"] = roof["valu_issue_frac"]
                roof["achieved"] = round(roof["valu_wave_insts_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
                if roof.get("traffic"):
                    roof["memory"]["fabric_bytes_per_launch"] = roof["traffic"]
                    roof["memory"]["fabric_gbs"] = round(roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
                    roof["memory"]["cache_hit_ratio"] = round(1.0 - roof["traffic"] / (nbytes / n_launch), 3)
            line["roofline"] = roof
            # brute-force-equivalent flops mean nothing for a frame traced through the tree
            line["pipeline"].pop("achieved"); line["pipeline"].pop("unit"); line["pipeline"].pop("frac")
            line["metric"] = f"Mrays/s (primary+secondary closest-hit queries), {args.workload}"
        if world == 1 and not args.no_cpu_baseline:
            headline = args.workload == "cornell_1080p_b4"
            base = cpu_baseline(flat, options, 0, row_step=args.cpu_row_step or (1 if headline else 0))
            if headline:
                # the same oracle on one thread (every 16th row of the frame), and the reference's own small case
                # (cornellScene.json 160x120, default bounces) on one thread and on all of them
                one = cpu_baseline(flat, options, 0, row_step=max(16, args.cpu_row_step), nthreads=1)
                base["one_thread"] = {k: one[k] for k in ("value", "unit", "cores", "sample", "seconds")}
                small_flat, small_opt = get_flat_workload("cornell_160x120_b1")
                m1 = cpu_baseline(small_flat, small_opt, 0, row_step=1, nthreads=1)
                m1_all = cpu_baseline(small_flat, small_opt, 0, row_step=1)
                base["cornell_160x120_b1"] = {"one_thread": {k: m1[k] for k in ("value", "unit", "cores", "sample", "seconds")},
                                              "all_threads": {k: m1_all[k] for k in ("value", "unit", "cores", "sample", "seconds")}}
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
