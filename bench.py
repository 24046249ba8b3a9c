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


def pmc_traffic(workload):
    """Hardware counters of one tree level's launches from the newest committed counter run (profiles/*_pmc_traffic.json,
    made by tools/gpu_pmc.sh + tools/pmc_summary.py): HBM bytes, executed vector-ALU wave instructions, active lanes per
    instruction; None when there is none for this workload.  Counters cannot be read from inside this process, so these
    are the roofline fields that are not measured live; the file's commit and date go into the line with them."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload and d.get("kernel") == "k_level":
            best = (d, os.path.basename(path))
    return best


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
                    help="independent frames traced at the same time, each with a context and a stream of its own "
                         "(0: one for LBVH scenes; brute-force scenes two with one rank, three with two, four with more)")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="consecutive frames whose rows one render call traces together, and one gather collects "
                         "(0: as many as there are ranks, so that a call traces a whole frame's worth of rows whatever their number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (RCCL over xGMI, one GPU per rank) or gloo: tiles gathered through host memory, and ranks share the "
                         "visible GPUs when there are fewer of them than ranks -- a rehearsal of the several-rank path on a small box")
    ap.add_argument("--cpu-row-step", type=int, default=0, help="CPU baseline renders every n-th row (0: every row for the headline workload, one row per thread otherwise)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gloo = world > 1 and args.dist_backend == "gloo"
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
    seed = 0
    # Consecutive frames are independent, and the launches of one 1080p frame are too small to keep the device full
    # through their starts and ends: F frames are in flight, each a chain of its own (context, stream, tile buffer;
    # trace, then gather, on that stream) with nothing that orders one chain against another.  Every step is still one
    # complete frame, traced and gathered.
    lbvh_scene = args.accel == "lbvh" or (args.accel == "auto" and flat.n_tris + flat.n_spheres > 256)
    # one rank's share of a frame is a chain of small launches: more of them in flight, and full rounds (no slot of a
    # round wasted; the other chains hide the latency the adaptive cut is there to shorten)
    F = args.frames_in_flight if args.frames_in_flight > 0 else (1 if lbvh_scene else (3 if world <= 2 else 4))
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
        c.upload(flat)
        ctxs.append(c)
        outs.append(torch.empty((B, tile.nrows, W, 3) if B > 1 else (tile.nrows, W, 3), dtype=torch.uint8, device=device))
        # explicit streams: a null handle would mean "the context's own stream" to the library, and the gather that
        # follows every render has to be ordered behind it
        streams.append(torch.cuda.Stream(device))
    ctx = ctxs[0]
    counter = [0]

    def step(want_stats, frames=B):
        """One render call on the next chain: this rank's rows of `frames` consecutive frames, then their gather."""
        k = counter[0] % F
        counter[0] += 1
        torch.cuda.set_stream(streams[k])
        out = outs[k] if frames == B else outs[k][:frames]
        st = ctxs[k].render_rows_device(options, seed, tile.row0, tile.nrows, out.data_ptr(), stream=streams[k].cuda_stream,
                                        want_stats=want_stats, row_stride=tile.stride, frames=frames)
        if gloo:
            streams[k].synchronize()
            frame = gather_frame(out.cpu(), H, args.partition, slot=k)
        else:
            frame = gather_frame(out, H, args.partition, slot=k)
        return st, frame

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # checked calls, one per chain: counters, and work buffers grown to what this scene's trace trees need
    first_stats, frame = step(True)
    for _ in range(F - 1):
        step(True)
    for c in ctxs:
        c.set_timing(False)
    # the timed region is exactly args.steps frames: calls of B frames, and one smaller call for what is left over
    full_calls, rest = divmod(args.steps, B) if B > 1 else (args.steps, 0)
    # untimed calls: the warm-up asked for, and at least as many as every chain needs to reach its steady state (a
    # band is replayed as a launch graph from its third request with the same counter set, i.e. a chain's fifth call)
    for _ in range(max((args.warmup + B - 1) // B - 1, 5 * F)):
        step(False)
    if rest:
        for _ in range(5 * F):
            step(False, rest)
    sync()
    t0 = time.perf_counter()
    for i in range(full_calls):
        step(False)
    if rest:
        step(False, rest)
    sync()
    elapsed = time.perf_counter() - t0
    if rest:   # every chain's last call is to be one of B frames for the checks below
        for _ in range(F):
            step(False)
        sync()
    # the timed calls were asynchronous and checked nothing: every chain's last frame must have fitted its buffers
    # and counted the same queries as the checked first step (raises on a capacity overflow)
    for c in ctxs:
        chain_stats = c.last_call_stats()
        if chain_stats["queries"] != first_stats["queries"]:
            raise RuntimeError(f"a timed frame traced {chain_stats['queries']} queries, the checked frame {first_stats['queries']}")
    # kernel durations (HIP events on the launch stream) come from one more frame, outside the timed region and alone
    # on the device: events between the launches are stops in the stream, and kernels that share the device with
    # another frame's have no duration of their own
    ctx.set_timing(True)
    torch.cuda.set_stream(streams[0])
    alone = torch.empty((tile.nrows, W, 3), dtype=torch.uint8, device=device)
    last = ctx.render_rows_device(options, seed, tile.row0, tile.nrows, alone.data_ptr(), stream=streams[0].cuda_stream,
                                  want_stats=True, row_stride=tile.stride)
    sync()
    # every chain rendered the same frame, B times per call: the tiles left by the timed, replayed, concurrent calls
    # must equal the tile of the call just made alone
    for k in range(F):
        for b, got in enumerate(outs[k] if B > 1 else [outs[k]]):
            if not torch.equal(got, alone):
                raise RuntimeError(f"the tile of chain {k}, frame {b} of its call, differs from the tile rendered alone")
    # counters of the checked call are those of its B frames, which are the same frame
    for key in list(first_stats):
        if isinstance(first_stats[key], int) and key not in ("launches", "n_k_level"):
            if first_stats[key] % B:
                raise RuntimeError(f"counter {key} of a call of {B} equal frames is not a multiple of {B}")
            first_stats[key] //= B
    if first_stats["queries"] != last["queries"]:
        raise RuntimeError(f"a frame of the checked call traced {first_stats['queries']} queries, the frame rendered alone {last['queries']}")

    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    keys = ["queries", "rays_api", "tri_tests", "sphere_tests", "aabb_tests", "shadow_queries", "trace_queries", "nodes", "shadow_culled"]
    totals = dict(zip(keys, sum_counters([first_stats[k] for k in keys], device=comm_device)))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # only queries that were traced count: shadow rays whose light term is zero whatever they hit are answered without a
        # traversal (t3d_stats.shadow_culled); the reference traces and counts them, they are reported beside the value
        traced = totals["queries"] - totals["shadow_culled"]
        last_traced = last["queries"] - last["shadow_culled"]
        value = traced * args.steps / elapsed / 1e6
        # roofline of the dominant kernel on this rank (k_level: every closest-hit query of the frame):
        # algorithmic flops / measured kernel time
        T, S = flat.n_tris, flat.n_spheres
        flops = last_traced * (FLOP_PER_TRI_TEST * T + FLOP_PER_SPHERE_TEST * S)
        level_s = last["ms_k_level"] * 1e-3
        achieved = flops / level_s / 1e12 if level_s > 0 else 0.0
        line = {
            "metric": "Mrays/s (primary+secondary closest-hit queries), " +
                      ("Cornell 1080p 4 bounces" if args.workload == "cornell_1080p_b4" else args.workload),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # the same frames per second in the reference's count of rays (it traces, and counts, the shadow rays that this
            # library answers without tracing): the figure to set against cpu_baseline, whose oracle traces them all
            "value_in_reference_queries": round(totals["queries"] * args.steps / elapsed / 1e6, 2),
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "ranks": world, "collective": ("gloo gather through host memory (rehearsal)" if gloo else
                                                                               ("RCCL gather" if world > 1 else "none")),
                       "width": W, "height": H, "bounces": options["bounces"],
                       "samples": options["samples"], "triangles": T, "spheres_and_lights": S,
                       "lights": int(len(flat.light_strength)), "partition": f"{args.partition} rows x{world}",
                       "frames_in_flight": F * B, "calls_in_flight": F, "frames_per_call": B,
                       "queries_per_frame": traced, "shadow_rays_answered_untraced_per_frame": totals["shadow_culled"],
                       "reference_queries_per_frame": totals["queries"], "rendered_rays_per_frame": totals["rays_api"]},
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
                         "measured_on": "one frame alone on the device, after the timed steps",
                         # the same launches priced by the queries the reference counts (the shadow rays answered without a ray
                         # included): comparable with figures that predate the culling of those rays
                         "frac_in_reference_queries": round(last["queries"] * (FLOP_PER_TRI_TEST * T + FLOP_PER_SPHERE_TEST * S)
                                                            / level_s / 1e12 / FP32_VALU_PEAK_TFLOPS, 4) if level_s > 0 else None},
            # the whole pipeline as timed: every flop of the frame over the time a frame takes with F in flight
            "pipeline": {"frames_in_flight": F * B, "frame_latency_ms": round(last["ms_device"], 4),
                         # this rank's share of one frame with nothing else on the device, as a rate
                         "frame_alone_mrays": round(last_traced / (last["ms_device"] * 1e-3) / 1e6, 2) if last["ms_device"] > 0 else None,
                         "achieved": round(flops / (ms_per_step * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                         "frac": round(flops / (ms_per_step * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4)},
        }
        tr = pmc_traffic(args.workload) if world == 1 else None

        def counters_into(roof):
            """Counter-derived fields (from the committed run, see pmc_traffic) next to the live ones."""
            if not tr:
                return
            d = tr[0]
            roof["traffic"] = d["hbm_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{tr[1]} (commit {d.get('commit')}, {d.get('date')})"
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
                               "hbm_peak_gbs": HBM_PEAK_GBS, "note": "record bytes requested by the lanes; most are cache hits"}}
            counters_into(roof)
            if roof.get("valu_issue_frac") is not None:
                roof["frac"] = roof["valu_issue_frac"]
                roof["achieved"] = round(roof["valu_wave_insts_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
                if roof.get("traffic"):
                    roof["memory"]["fabric_bytes_per_launch"] = roof["traffic"]
                    roof["memory"]["cache_hit_ratio"] = round(1.0 - roof["traffic"] / (nbytes / n_launch), 3)
            line["roofline"] = roof
            # brute-force-equivalent flops mean nothing for a frame traced through the tree
            line["pipeline"].pop("achieved"); line["pipeline"].pop("unit"); line["pipeline"].pop("frac")
            line["metric"] = f"Mrays/s (primary+secondary closest-hit queries), {args.workload}"
        if world == 1 and not args.no_cpu_baseline:
            headline = args.workload == "cornell_1080p_b4"
            base = cpu_baseline(flat, options, seed, row_step=args.cpu_row_step or (1 if headline else 0))
            if headline:
                # the same oracle on one thread (every 16th row of the frame), and the reference's own small case
                # (cornellScene.json 160x120, default bounces) on one thread and on all of them
                one = cpu_baseline(flat, options, seed, row_step=max(16, args.cpu_row_step), nthreads=1)
                base["one_thread"] = {k: one[k] for k in ("value", "unit", "cores", "sample", "seconds")}
                small_flat, small_opt = get_flat_workload("cornell_160x120_b1")
                m1 = cpu_baseline(small_flat, small_opt, seed, row_step=1, nthreads=1)
                m1_all = cpu_baseline(small_flat, small_opt, seed, row_step=1)
                base["cornell_160x120_b1"] = {"one_thread": {k: m1[k] for k in ("value", "unit", "cores", "sample", "seconds")},
                                              "all_threads": {k: m1_all[k] for k in ("value", "unit", "cores", "sample", "seconds")}}
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
