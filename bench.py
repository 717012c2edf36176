#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflection) on scene7 @ 3840x2160, max depth 5.

One "step" is one full frame: every rank renders its interleaved row tiles of the same 4K frame
(scene and grid replicated per GPU), then the rows are gathered to rank 0 over RCCL.  The scene, the
grid and all queues are resident in HBM before the timed region; the timed region is K frames between
barrier + synchronize pairs, maximum over ranks.  Total work per step is fixed, so scaling is "strong".

`value` is the throughput of a stream of frames: at every N, including N = 1, a few frames are in flight (each on its
own HIP stream and device context), because the end of a frame is a handful of long chains of dependent rays that
leave most of the GPU idle; `one_frame_at_a_time` holds the latency figure (one frame in flight, its gather completed
before the next frame starts) at every N as well, so that a curve over N compares equals either way.
Which pipeline of the persistent launch the frames run is the library's choice by frame size (`pipeline` in the JSON
line: generations of one kind of ray for whole 4K frames that share the GPU, per-lane paths for a lone frame or a small
share).  Every timed frame is rendered as a first frame: the renderer's scheduling hint from the previous frame of the same
camera (an ordering of the screen tiles, never a result) is switched off, because the reference renders one frame
per process.  The rates with the hint are reported beside it (`with_previous_frame_hint`).  `value` counts the
reference's rays (one camera ray per pixel, one shadow ray per light and one mirror ray per hit); camera rays that
cannot meet any primitive are answered by a fill instead of a grid walk, so the rays that walked the grid and their
rate are reported at top level too (`rays_walked`, `Mrays_walked_per_s`).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob as glob_mod
import json
import os
import sys
import time



def _frames_in_flight(world, sync_gather, hw_queues):
    """(frames in flight, workgroups per CU of each frame's launch) of the throughput figure: enough frames to keep a GPU
    busy while the last paths of a frame drain.  Measured on one GPU, ms per frame of a rank's share (tools/ab_inflight.sh,
    tools/ab_ranks2.sh): whole frames 3 x 2 1.64, 4 x 2 1.65, 2 x 4 1.75; half frames 3 x 2 0.96, 4 x 2 0.92, 6 x 1 1.05;
    quarters 4 x 2 0.52, 6 x 1 0.51; eighths 4 x 2 0.32, 6 x 1 0.31 (8 x 1 0.46 before the library ran such launches
    with its four-wave build).  Six streams need
    more than the four hardware queues a process gets by default: streams that share a queue run their kernels one after
    the other (eighths: 0.37 ms per frame with 8 queues, 0.57 with 4), so with fewer than seven queues four frames it is."""
    if sync_gather:
        return 1, 0
    if world >= 4 and hw_queues >= 7:
        return 6, 1
    return (4, 2) if world >= 2 else (3, 2)


# GPU_MAX_HW_QUEUES must be set before the HIP runtime starts.  A value the operator has set is left alone (and the
# number of frames in flight follows it); only a run that wants six frames in flight and finds the variable unset asks
# for eight.  The effective value goes into the JSON line.
_ONE_AT_A_TIME = any(a in sys.argv for a in ("--sync-gather", "--one-frame-at-a-time"))
if int(os.environ.get("WORLD_SIZE", "1")) >= 4 and not _ONE_AT_A_TIME and "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
HW_QUEUES = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracer_amd as rt  # noqa: E402
from raytracer_amd.multi_gpu import FrameGather  # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3      # 256 CU x 128 lanes x 2 (FMA) x 2.4 GHz
PEAK_UNFUSED_TFLOPS = PEAK_FP32_VALU_TFLOPS / 4   # IEEE mul/add kept separate and unpacked: 1 flop/lane/clk
PEAK_PACKED_UNFUSED_TFLOPS = PEAK_FP32_VALU_TFLOPS / 2   # separate mul/add, two lanes' worth per packed instruction
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(w):
    """fp32 operations of the reference algorithm for the counted work of one kernel family
    (tri, sphere, xform, cell steps, fine steps, light evals, nearest hits, rays):
    75 per triangle test, 31 per sphere test, 33 per ray->object transform, 50 per ray for slab test + walk
    setup, 5 per cell step (coarse or fine), 110 per unshadowed light evaluation, 60 per nearest hit."""
    tri, sph, xf, cells, fine, light, hits, rays = (float(v) for v in w)
    return 75 * tri + 31 * sph + 33 * xf + 50 * rays + 5 * (cells + fine) + 110 * light + 60 * hits


def cpu_baseline(scene_path, width, height, max_depth, row_start, row_step):
    """Times the CPU oracle (reference-shaped port) on a bounded row subset of the same frame."""
    from oracle import ref_oracle
    counted = ref_oracle.RefScene(scene_path, counting=True)
    _, _, _, cnt = counted.render(width, height, max_depth, row_start, row_step)
    counted.close()
    plain = ref_oracle.RefScene(scene_path, counting=False)
    ref_oracle.set_schedule(True)                 # rows to threads as the reference's `#pragma omp parallel for` does
    _, _, sec, _ = plain.render(width, height, max_depth, row_start, row_step)
    ref_oracle.set_schedule(False)
    _, _, sec_dynamic, _ = plain.render(width, height, max_depth, row_start, row_step)
    plain.close()
    rays = cnt["primary"] + cnt["shadow"] + cnt["reflect"]
    n_rows = len(range(row_start, height, row_step))
    return {"value": rays / sec / 1e6, "unit": "Mrays/s", "cores": ref_oracle.threads(), "kind": "port",
            "sample": f"rows {row_start}::{row_step} ({n_rows} of {height} rows, {n_rows * width} pixels, {rays} rays) "
                      f"of the same scene7 {width}x{height} depth-{max_depth} frame, OpenMP over rows with the reference's "
                      f"static schedule, {sec:.2f} s",
            "seconds": sec, "rays": rays, "value_dynamic_schedule": rays / sec_dynamic / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "scene7.test"))
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--sub", type=int, default=6, help="fine cells per coarse cell and axis of the nested grid")
    ap.add_argument("--search", choices=["nested", "reference-lists"], default="nested")
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-lists", action="store_true", help="skip the secondary reference-list measurement")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline measurement and its roofline: no frames without the scheduling hint, no secondary "
                         "figures, no CPU baseline (what the committed rocprofv3 kernel trace is taken over, so that every "
                         "launch of the frame kernel in it is one of the measured kind)")
    ap.add_argument("--cpu-row-step", type=int, default=16)
    # rehearsal of the multi-rank control flow on a machine with fewer GPUs than ranks: collectives over gloo on
    # host copies, every rank rendering on GPU 0; never used for reported numbers
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--check-image", action="store_true", help="rank 0 also renders the whole frame alone and compares")
    ap.add_argument("--sync-gather", "--one-frame-at-a-time", dest="sync_gather", action="store_true",
                    help="one frame in flight: every frame is rendered, gathered and assembled before the next one starts "
                         "(`value` then is the latency figure)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="override the number of frames kept in flight (tuning runs)")
    ap.add_argument("--workgroups-per-cu", type=int, default=0, help="... and the workgroups per CU of each frame's launch")
    ap.add_argument("--gather", choices=["f32", "u8"], default="f32",
                    help="what travels to rank 0: the float framebuffer (Image3f, the default) or the 8-bit image the "
                         "reference saves, converted on each GPU before the gather (a quarter of the bytes)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_reference_lists = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the renderer has no CPU path", file=sys.stderr)
        sys.exit(2)
    if args.share_gpu:
        local_rank = 0
    visible = torch.cuda.device_count()
    if not args.share_gpu and (args.gpus > visible or local_rank >= visible):
        # before any timing: a run that cannot give every rank a GPU of its own must not produce a line
        print(f"bench.py: --gpus {args.gpus} (local rank {local_rank}) but only {visible} GPU(s) are visible to this process; "
              f"a rehearsal with ranks sharing a GPU needs --share-gpu --backend gloo", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_gather = args.backend == "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_gather:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, max_depth = args.width, args.height, args.max_depth
    search = rt.SEARCH_NESTED if args.search == "nested" else rt.SEARCH_REFERENCE_LISTS
    sr = rt.SceneReader().readScene(args.scene)
    r = rt.Renderer(local_rank)
    r.set_scene(sr)
    # the nested search needs the nested grid only; the reference's cell lists are built where they are searched
    grid_flags = rt.GRID_NESTED if search == rt.SEARCH_NESTED else rt.GRID_REFERENCE
    gi = r.build_grid(15, args.sub, grid_flags)

    as_u8 = args.gather == "u8"
    # Three to six frames are in flight (see _frames_in_flight), each on its own stream and device context.  The end of a frame -- and, with many
    # ranks, all of a rank's small share of it -- is set by the longest chains of dependent rays rather than by the
    # amount of work; several frames side by side (each kernel launched with one or two workgroups per CU) fill the GPU,
    # and the gather of a frame travels while the next ones render.  Measured on one GPU: whole frames, scheduled as
    # first frames, 2.73 ms one at a time, 2.18 two in flight, 2.14 three; an eighth of a frame takes 1.4 ms however many
    # CUs work on it, so at 8 ranks six frames with one workgroup per CU each run side by side (no gather: 0.33 ms per
    # frame against 0.51 with three frames, 0.75 with two).
    depth, wg_per_cu = _frames_in_flight(world, args.sync_gather, HW_QUEUES)
    if args.frames_in_flight > 0 and not args.sync_gather:
        depth, wg_per_cu = args.frames_in_flight, max(args.workgroups_per_cu, 1)
    renderers = [r]
    for _ in range(1, depth):
        extra = rt.Renderer(local_rank)
        extra.set_scene(sr)
        extra.build_grid(15, args.sub, grid_flags)
        renderers.append(extra)
    streams = [torch.cuda.current_stream()] if depth == 1 else [torch.cuda.Stream() for _ in range(depth)]
    fg = FrameGather(H, W, 3, args.tile_rows, rank, world, torch.device("cpu") if host_gather else dev,
                     dtype=torch.uint8 if as_u8 else torch.float32, depth=depth)
    # per slot: `local` receives the float rows of this rank; `staged` is what the gather sends (the same tensor
    # unless the rows are converted to 8 bits first or the collective runs on host copies)
    staged = [torch.empty_like(fg.local_view(b), device=dev) if host_gather else fg.local_view(b) for b in range(depth)]
    local = [torch.empty(tuple(t.shape), dtype=torch.float32, device=dev) if as_u8 else t for t in staged]

    def render(slot=0, want_stats=False, count=False, mode=search, wg=0, hint=False, generations=None):
        """Enqueues this rank's rows of one frame on the slot's stream (the caller has made it current).  hint=False:
        the frame is scheduled as a first frame (no tile order from the previous frame).  generations=None: the library
        picks the pipeline of the persistent launch (by frame size, for launches that share the GPU)."""
        stream = streams[slot].cuda_stream
        rr = renderers[slot]
        st = rr.render_device(local[slot].data_ptr(), 0, W, H, max_depth, mode, args.tile_rows, rank, world, stream=stream,
                              want_stats=want_stats, count_work=count, workgroups_per_cu=wg, tile_hint=hint, generations=generations)
        if as_u8:
            rr.quantize_device(local[slot].data_ptr(), local[slot].shape[0] * W, staged[slot].data_ptr(), stream=stream)
        if host_gather:
            fg.local_view(slot).copy_(staged[slot])
        return st

    def frame(want_stats=False, count=False, mode=search, hint=False, generations=None):
        """One complete frame on slot 0: render, exchange, assemble (nothing left in flight)."""
        with torch.cuda.stream(streams[0]):
            st = render(0, want_stats, count, mode, hint=hint, generations=generations)
            img = fg.gather(0)
        streams[0].synchronize()
        return st, img

    def timed(fn, n):
        """Wall time of fn(n) between barrier + synchronize pairs, maximum over the ranks, in ms per frame."""
        sync()
        t0 = time.perf_counter()
        fn(n)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_gather else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / n * 1e3

    launch_ms = []        # device time of every launch of the frame kernel in the last frames() call (rank 0's launches)

    def frames(n, hint=False, collect=False):
        """n frames back to back; a slot's exchange is completed when the slot is needed again.  collect: the device
        time of every launch of the frame kernel is read (the library's HIP events on the slot's stream) when the slot's
        frame has completed -- which makes the host wait for that frame before it enqueues the slot's next one, so the
        headline loop runs without it and the roofline's launch times come from a second, identical loop."""
        launch_ms.clear()
        pending = [False] * depth
        for k in range(n):
            b = k % depth
            with torch.cuda.stream(streams[b]):
                if fg.in_flight(b):
                    fg.finish(b)
                if collect and pending[b]:
                    launch_ms.append(renderers[b].frame_times()[1])
                render(b, wg=wg_per_cu, hint=hint)
                pending[b] = True
                fg.gather_async(b)
        for b in range(depth):
            with torch.cuda.stream(streams[b]):
                if fg.in_flight(b):
                    fg.finish(b)
                if collect and pending[b]:
                    launch_ms.append(renderers[b].frame_times()[1])

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every slot's device context renders its first two frames here (its allocations), so that the W warm-up frames and
    # the timed ones meet the same state whichever slot they fall on
    for b in range(depth if depth > 1 else 0):
        with torch.cuda.stream(streams[b]):
            for _ in range(2):
                render(b, wg=wg_per_cu)
    sync()
    frames(args.warmup)
    ms_per_step = timed(frames, args.steps)
    # the same loop once more with the device time of every launch read out (see frames): the roofline's launch figures
    collect_ms_per_step = timed(lambda n: frames(n, collect=True), args.steps)
    timed_launch_ms = list(launch_ms)
    # latency: one frame at a time, each frame's gather completed before the next one starts
    one_at_a_time = lambda n, hint=False: [frame(hint=hint) for _ in range(n)]
    hinted_ms = latency_ms = latency_hinted_ms = None
    if not args.headline_only:
        one_at_a_time(max(args.warmup, 1))
        latency_ms = timed(one_at_a_time, args.steps)
        # the same two loops with the previous frame's scheduling hint (a camera that stands still or moves a little)
        frames(max(args.warmup, 2), hint=True)
        hinted_ms = timed(lambda n: frames(n, hint=True), args.steps)
        one_at_a_time(max(args.warmup, 2), hint=True)
        latency_hinted_ms = timed(lambda n: one_at_a_time(n, hint=True), args.steps)

    # which pipeline the timed launches ran (the library chooses by frame size for launches that share the GPU)
    with torch.cuda.stream(streams[0]):
        pipeline = render(0, want_stats=True, wg=wg_per_cu).pipeline
    streams[0].synchronize()
    gen = pipeline == 1
    # ray counts, work counters and per-kernel times of this rank's share (separate, untimed frames, the timed launches'
    # pipeline)
    st_count, _ = frame(want_stats=True, count=True, generations=gen)
    # the same kernel with the GPU to itself: HIP events around it (on the stream it is launched on), mean over as many
    # frames as were timed above
    kern = []
    for _ in range(max(args.steps, 5)):
        st_t, _ = frame(want_stats=True, generations=gen)
        kern.append((st_t.frame_ms, st_t.trace_ms, st_t.shade_ms, st_t.resolve_ms))
    kern = np.mean(np.array(kern), axis=0)
    rays_local = torch.tensor([st_count.primary_rays, st_count.shadow_rays, st_count.reflection_rays, st_count.traced_primary_rays],
                              dtype=torch.float64, device="cpu" if host_gather else dev)
    if world > 1:
        dist.all_reduce(rays_local)
    rays = [int(v) for v in rays_local.tolist()]
    # who took part: the device every rank rendered on, as the collective backend saw them
    me = {"rank": rank, "device_index": local_rank, "device": torch.cuda.get_device_name(local_rank)}
    participants = [me]
    if world > 1:
        participants = [None] * world
        dist.all_gather_object(participants, me)
    total_rays = sum(rays[:3])
    walked_rays = rays[3] + rays[1] + rays[2]

    out = None
    if rank == 0:
        value = total_rays / (ms_per_step * 1e-3) / 1e6
        # the persistent frame kernel does all of a frame's work in one launch (trace_ms = its HIP-event time);
        # its work counters come from the instrumented build of the same kernel on the same frame
        work = [a + b for a, b in zip(st_count.trace_work, st_count.shade_work)]
        flops = algorithmic_flops(work)
        # Launches of the timed region: their mean device time (HIP events on each launch's own stream) and how many of
        # them ran side by side on average (sum of their durations / length of the timed region).  One launch's rate is
        # flops / avg_launch_ms; the GPU runs `launches_running_at_once` of them, so its rate over the timed region is
        # the product, which equals flops * launches / time.  With one frame in flight the two are the same number.
        avg_launch_ms = float(np.mean(timed_launch_ms)) if timed_launch_ms else kern[1]
        at_once = float(np.sum(timed_launch_ms)) / (collect_ms_per_step * args.steps) if timed_launch_ms else 1.0
        achieved = flops / (avg_launch_ms * 1e-3) / 1e12 * at_once
        achieved_alone = flops / (kern[1] * 1e-3) / 1e12
        frame_bytes = 12.0 * W * H / world
        roofline = {
            "bound": "valu",
            "kernel": ("k_frame_gen" if gen else "k_frame") + " (persistent: camera rays, coarse + nested 3D-DDA, triangle/sphere "
                      "tests, shadow and mirror rays, Blinn-Phong, fold), 1 launch per frame; figures are per launch on rank 0 "
                      "over a repeat of the timed loop (first-frame schedule; the library's HIP events on each launch's stream, "
                      "read when a slot's frame has completed, which holds the host back a little: hence the repeat)",
            "avg_launch_ms": avg_launch_ms, "launches_timed": len(timed_launch_ms), "launches_running_at_once": at_once,
            "ms_per_step_of_the_loop_the_launches_were_timed_in": collect_ms_per_step,
            "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
            "frac_of_packed_unfused_peak": achieved / PEAK_PACKED_UNFUSED_TFLOPS, "peak_packed_unfused": PEAK_PACKED_UNFUSED_TFLOPS,
            "frac_of_unfused_peak": achieved / PEAK_UNFUSED_TFLOPS, "peak_unfused": PEAK_UNFUSED_TFLOPS,
            "alone": {"what": "the same kernel with the GPU to itself (one frame at a time, untimed frames after the timed loop)",
                      "avg_launch_ms": kern[1], "achieved": achieved_alone, "frac": achieved_alone / PEAK_FP32_VALU_TFLOPS},
            "algorithmic_flops_per_launch": flops,
            "work_per_launch": dict(zip(("tri_tests", "sphere_tests", "xform_evals", "cell_steps", "fine_steps",
                                         "light_evals", "nearest_hits", "rays"), (int(v) for v in work))),
            "hbm_algorithmic_GBs": frame_bytes / (kern[0] * 1e-3) / 1e9, "hbm_peak_GBs": PEAK_HBM_GBS,
            "traffic": None,
        }
        # Counters of this kernel against the hardware's capacity, from the last committed rocprofv3 PMC passes of the same
        # kernel on the same frame (bench.py does not run the profiler; the passes profile one launch at a time, the timed
        # loop keeps several in flight: `valu_issue_busy_timed_loop` rescales the instruction count to the timed loop)
        counters_file = sorted(glob_mod.glob(os.path.join(ROOT, "profiles", "*_counters.json")))
        default_workload = (W, H, max_depth, args.sub) == (3840, 2160, 5, 6) and os.path.basename(args.scene) == "scene7.test"
        if counters_file and world == 1 and args.search == "nested" and default_workload and gen:
            with open(counters_file[-1]) as fh:
                pmc = json.load(fh)
            simd_cycles = 1024 * 2.4e9          # 256 CUs x 4 SIMDs, 2.4 GHz: a wave64 VALU instruction holds a SIMD for 4 cycles
            roofline["counters"] = dict(pmc, source=os.path.relpath(counters_file[-1], ROOT),
                                        measured_on="committed profile: one launch at a time (lone launch), not the timed loop",
                                        valu_issue_busy_timed_loop=pmc["valu_insts_per_launch"] * 4 / (simd_cycles * ms_per_step * 1e-3))
        # memory-side bytes per launch from the last committed rocprofv3 PMC passes of this same kernel and frame
        # (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as gfx950 requires); bench.py itself does
        # not run the profiler
        traffic_files = sorted(glob_mod.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        if traffic_files and world == 1 and args.search == "nested" and default_workload:
            with open(traffic_files[-1]) as fh:
                roofline["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
            roofline["traffic_source"] = (os.path.relpath(traffic_files[-1], ROOT) + " (committed profile of one launch at a time: a lone "
                                          "launch of this kernel on this frame, not the timed loop's overlapping launches)")
        out = {
            "metric": "Mrays/sec (primary+shadow+reflection) on scene7 @4K", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "rays_walked": walked_rays, "Mrays_walked_per_s": walked_rays / (ms_per_step * 1e-3) / 1e6,
            "frames_in_flight": depth,
            "pipeline": "generations of one kind of ray (k_frame_gen)" if gen else "per-lane paths (k_frame)",
            "frame_schedule": "every timed frame is a first frame: no tile order from a previous frame",
            "one_frame_at_a_time": None if latency_ms is None else {
                "ms_per_frame": latency_ms, "value": total_rays / (latency_ms * 1e-3) / 1e6,
                "Mrays_walked_per_s": walked_rays / (latency_ms * 1e-3) / 1e6,
                "what": "latency figure: one frame in flight, rendered, gathered and assembled before the next one starts "
                        "(first frames, as the headline)"},
            "with_previous_frame_hint": None if hinted_ms is None else {
                "ms_per_step": hinted_ms, "value": total_rays / (hinted_ms * 1e-3) / 1e6,
                "Mrays_walked_per_s": walked_rays / (hinted_ms * 1e-3) / 1e6,
                "one_frame_at_a_time_ms": latency_hinted_ms,
                "what": "the same loops with the screen tiles ordered by the path depths the previous frame of the same "
                        "camera recorded (order only; every ray of every frame is traced)"},
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default: 4)"),
            "backend": ("gloo on host copies (rehearsal)" if host_gather else "nccl (RCCL)") if world > 1 else "none (one rank: no collective runs)",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "visible_devices": visible, "participants": participants,
            "data": "scenes/scene7.test (the reference's own scene file: 100000-triangle dragon), no randomness",
            "config": {"workload": f"scene7.test @ {W}x{H}, maxdepth {max_depth}, 15^3 grid, search={args.search}, "
                                   f"sub={gi.sub}, interleaved {args.tile_rows}-row tiles over {world} GPU(s), "
                                   f"{args.gather} gather to rank 0, {depth} frame(s) in flight, every frame scheduled as a "
                                   f"first frame; {walked_rays} of the {total_rays} rays walk the grid (camera rays "
                                   f"through screen tiles that see no geometry are filled)",
                       "rays_per_frame": {"primary": rays[0], "shadow": rays[1], "reflection": rays[2], "total": total_rays,
                                          "primary_walked": rays[3], "primary_filled": rays[0] - rays[3]},
                       "kernel_ms_rank0": {"frame": kern[0], "k_frame": kern[1]},
                       "grid": {"fine_cells": gi.n_fine_cells, "occupied_fine_cells": gi.n_occupied_fine_cells,
                                "refs_fine": gi.n_fine_refs, "build_ms": gi.build_ms}},
            "roofline": roofline,
        }
    # secondary figure: the reference's own workload (every primitive of every reference cell list tested)
    if not args.no_reference_lists and world == 1 and search == rt.SEARCH_NESTED:
        gi_ref = r.build_grid(15, args.sub, rt.GRID_REFERENCE | rt.GRID_NESTED)
        out["config"]["grid"]["refs_reference_lists"] = gi_ref.n_refs
        out["config"]["grid"]["build_ms_with_reference_lists"] = gi_ref.build_ms
        stc, _ = frame(want_stats=True, count=True, mode=rt.SEARCH_REFERENCE_LISTS)
        stt, _ = frame(want_stats=True, mode=rt.SEARCH_REFERENCE_LISTS)
        tf = algorithmic_flops([a + b for a, b in zip(stc.trace_work, stc.shade_work)])
        out["reference_lists"] = {
            "what": "same frame with every primitive of each visited reference cell list tested (the reference's work)",
            "value": stc.total_rays / (stt.frame_ms * 1e-3) / 1e6, "unit": "Mrays/s", "frame_ms": stt.frame_ms,
            "kernel_ms": stt.trace_ms, "tri_tests": stc.tri_tests, "algorithmic_flops": tf,
            "achieved_TFLOPs": tf / (stt.trace_ms * 1e-3) / 1e12,
            "frac_of_peak": tf / (stt.trace_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "frac_of_unfused_peak": tf / (stt.trace_ms * 1e-3) / 1e12 / PEAK_UNFUSED_TFLOPS,
        }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.scene, W, H, max_depth, args.cpu_row_step // 2, args.cpu_row_step)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if args.check_image:
        _, img = frame()
        if rank == 0:
            whole = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            r.render_device(whole.data_ptr(), 0, W, H, max_depth, search, args.tile_rows, 0, 1, stream=streams[0].cuda_stream)
            torch.cuda.synchronize()
            if as_u8:
                whole_q = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
                r.quantize_device(whole.data_ptr(), H * W, whole_q.data_ptr(), stream=streams[0].cuda_stream)
                torch.cuda.synchronize()
                whole = whole_q
            out["image_check"] = bool(torch.equal(img.to(dev), whole))
            if not out["image_check"]:
                bad = (img.to(dev) != whole).any(dim=2).nonzero()
                out["image_check_detail"] = {"pixels_differing": int(bad.shape[0]), "first": bad[:4].tolist(),
                                             "rows": sorted(set(bad[:, 0].tolist()))[:24]}
    if rank == 0:
        print(json.dumps(out))
    for rr in renderers:
        rr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
