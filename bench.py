#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the path-tracing loop on the dog scene.

    python bench.py --gpus N --steps K --warmup W

A step = one whole frame of the workload (config 4 of the benchmark list: the room with
one textured dog, 1024x1024, 64 samples per pixel, full 64-segment paths) rendered by the
HIP kernels through the C ABI (include/rt_gpu.h) into device memory.  The scene is resident
in HBM before the timed region.  With N > 1 (one process per GPU, torch.distributed/RCCL)
every frame is cut into 8-row bands dealt round-robin to the ranks; each rank renders its
bands into a packed buffer and one all-gather over xGMI assembles the frame on every rank.
Frame i of a run is made of samples i*spp ... i*spp + spp - 1 of every pixel under one seed: the
frames are independent and the same pictures whatever N is (the generator is keyed by pixel and
sample), so this is strong scaling of fixed frames.

The end of a frame is latency-bound (a few long paths, a few thousand rays per launch) and a
1/N share of a frame makes small launches, so (a) with N > 1 one render call produces this
rank's share of --frames-per-call consecutive frames (default min(N, 4); options.frames of the
C ABI): launches as large as a whole frame's, one gather for all of them; (b) --frames-in-flight
calls (default 3 or 4) run concurrently, each on its own stream with its own device work lists:
the tail of one overlaps the machine-filling first launch of the next and, with N > 1, the
all-gather overlaps rendering.  Every timed frame is complete (rendered, gathered, assembled)
inside the timed region; ms_per_step is elapsed / K, not the latency of one frame
(--frames-in-flight 1 --frames-per-call 1 gives that).

Rank 0 prints one JSON line.  A frame is rendered by two kernels that alternate: rt_shade_kernel
(path bookkeeping, sampling, the scene's lights/spheres/planes, instance root boxes) and
rt_tree_kernel (mesh tree walks for the rays the shade kernel parked), a few dozen launches per
frame.  `roofline` prices them against HBM bandwidth with the ALGORITHMIC bytes of the path (bytes
the reference's data structures would move for the primitives, boxes, triangles, attributes and
texels the kernels actually tested, counted on the device), divided by the kernels' own duration
measured with HIP events on their stream: the top-level figure is the whole frame (all launches of
both kernels), `kernels` splits bytes and time per kernel and gives the average launch duration
that a rocprofv3 kernel trace of the same command shows.
`cpu_baseline` times the reference itself (compiled from its own sources with its build.sh
flags, oracle/_ref) on this machine's host cores over a bounded number of samples of the same
scene; when that library is absent it times the CPU oracle port instead.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X vector fp32 spec peak (256 CUs x 128 lanes x 2 x 2.4 GHz)


def work_model():
    """The per-event byte and flop constants of include/rt_work_model.h."""
    import re
    with open(os.path.join(ROOT, "include", "rt_work_model.h")) as f:
        return {k: int(v) for k, v in re.findall(r"#define\s+(RT_(?:BYTES|FLOPS)_\w+)\s+(\d+)", f.read())}


def algorithmic_bytes(st, pixels):
    """Bytes the reference's records would move for the work the kernels counted, per kernel:
    lights, spheres, planes, instance root tests, triangle-hit attributes, texels and the pixel store
    (shade kernel); the two child nodes per interior pop and the triangle with its three points per
    triangle test (tree kernel)."""
    m = work_model()
    shade = (m["RT_BYTES_LIGHT_TEST"] * st["light_tests"] + m["RT_BYTES_SPHERE_TEST"] * st["sphere_tests"] +
             m["RT_BYTES_PLANE_TEST"] * st["plane_tests"] + m["RT_BYTES_ROOT_TEST"] * st["root_tests"] +
             m["RT_BYTES_TRI_HIT"] * st["tri_hits"] + m["RT_BYTES_TEX_LOOKUP"] * st["tex_lookups"] +
             m["RT_BYTES_PIXEL"] * pixels)
    tree = m["RT_BYTES_INTERIOR_POP"] * st["interior_pops"] + m["RT_BYTES_TRI_TEST"] * st["tri_tests"]
    return shade, tree


def algorithmic_flops(st):
    """Flops of the reference's expressions for the same counted work (transcendentals excluded)."""
    m = work_model()
    return (m["RT_FLOPS_LIGHT_TEST"] * st["light_tests"] + m["RT_FLOPS_SPHERE_TEST"] * st["sphere_tests"] +
            m["RT_FLOPS_PLANE_TEST"] * st["plane_tests"] + m["RT_FLOPS_ROOT_TEST"] * st["root_tests"] +
            m["RT_FLOPS_INTERIOR_POP"] * st["interior_pops"] + m["RT_FLOPS_TRI_TEST"] * st["tri_tests"] +
            m["RT_FLOPS_TRI_HIT"] * st["tri_hits"] +
            m["RT_FLOPS_PATH_SEGMENT"] * (st["rays"] - st["shadow_rays"]))


def cpu_baseline(sh, width, height, target_seconds=12.0):
    """Time the CPU path on this host over a bounded sample of the workload."""
    import _libs as L
    from raytracing_amd import host
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    use_ref = L.have_ref("build")
    tile = 64
    tiles = ((width + tile - 1) // tile) * ((height + tile - 1) // tile)
    seeds = (L.T.RandomSerie * tiles)(*[L.serie_from_seed(1000 + i) for i in range(tiles)])

    def run(spp):
        img, _ = host.new_image(width, height)
        if use_ref:
            return L.ref("build").ref_render_tiles(sh.ptr, C.byref(img), spp, tile, threads, seeds)
        return L.oracle(contract=True).orc_render_tiles(sh.ptr, C.byref(img), spp, tile, threads,
                                                        L.RNG_AES, L.MATH_LIBM, 64, seeds, 0, None)

    t1 = run(1)
    spp = int(max(1, min(64, round(target_seconds / max(t1, 1e-3)))))
    t = run(spp) if spp > 1 else t1
    # rays per sample of this scene, counted by the oracle on a small frame (same scene, same path rules)
    st = L.Stats()
    small, _ = host.new_image(128, 128)
    L.oracle().orc_render_tiles(sh.ptr, C.byref(small), 4, 64, threads, L.RNG_AES, L.MATH_LIBM, 64,
                                seeds, 0, C.byref(st))
    rays_per_sample = st.rays / max(1, st.samples)
    samples = width * height * spp
    return {
        "value": round(samples * rays_per_sample / t / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "reference" if use_ref else "port",
        "sample": f"{width}x{height} at {spp} spp ({samples / 1e6:.2f} Msamples in {t:.2f} s, "
                  f"{samples / t / 1e6:.3f} Msamples/s, {rays_per_sample:.2f} rays/sample counted by the oracle), "
                  f"64x64 tiles on an atomic queue over {threads} threads, render only",
        "ms_per_frame_64spp_extrapolated": round(1e3 * t * 64 / spp, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--scene", default="dog_room")
    ap.add_argument("--max-segments", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend: nccl (= RCCL over xGMI, one GPU per rank); gloo gathers through host "
                         "memory and lets several ranks share a GPU (a functional check of the multi-process path on a "
                         "machine with fewer GPUs than ranks, not a measurement)")
    ap.add_argument("--split", choices=("bands", "samples"), default="bands",
                    help="how N devices share a frame: 8-row bands (bit-identical to one device) or sample ranges of "
                         "every pixel, summed in rank order (equal up to float summation order; for small frames)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="frames whose shares one render call produces (0 = 1 on one GPU, min(N, 4) when N GPUs share a frame)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="render calls running concurrently, each with its own stream and device work lists "
                         "(0 = 3 or 4 on one GPU by the number of steps, 2 to 4 when several GPUs share the frames)")
    args = ap.parse_args()

    # every frame in flight has its own stream; HIP maps streams onto 4 hardware queues unless told otherwise,
    # and frames that share a queue run one after the other
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    from raytracing_amd import gpu, multi_gpu, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())      # ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    W, H, spp = args.width, args.height, args.spp
    band = multi_gpu.BAND_ROWS
    sh = scenes.CONFIGS[args.scene]()
    by_samples = world > 1 and args.split == "samples"
    my_spp = spp
    # Frame i of a run = samples i*spp ... i*spp + spp - 1 of every pixel (one seed): independent frames
    # that are the same pictures whatever N is.  When several GPUs share a frame, a call renders this
    # rank's share of `per_call` consecutive frames at once (options.frames): launches as large as a whole
    # frame's instead of 1/N of it, one gather for all of them.
    if args.frames_per_call > 0:
        per_call = args.frames_per_call
    else:
        per_call = 1 if (world == 1 or by_samples) else min(world, 4)
    if by_samples:
        per_call = 1
    # more calls in flight overlap more, but the timed region starts and ends with a partly filled pipeline
    calls = (args.steps + per_call - 1) // per_call
    if args.frames_in_flight > 0:
        in_flight = args.frames_in_flight
    elif world == 1:
        in_flight = 4 if args.steps >= 16 else 3
    else:
        in_flight = max(2, min(4, calls))

    # One "lane" per call in flight: its own resident scene (with its own work lists on the device),
    # stream and output buffers.  Calls are made by worker threads (the C call releases the GIL),
    # so the latency-bound last rounds of one overlap the first, machine-filling launch of the next;
    # collectives stay on this thread, in frame order.
    class Lane:
        pass
    lanes = []
    for k in range(in_flight):
        ln = Lane()
        ln.g = gpu.GpuScene(sh)                        # flatten + upload: outside the timed region
        ln.stream = torch.cuda.Stream()
        if world > 1:
            rows = H if by_samples else multi_gpu.packed_rows(world, H, band)
            ln.mine = torch.zeros((per_call, rows, W, 3), dtype=torch.float32, device="cuda")
            ln.gathered = torch.zeros((world, per_call, rows, W, 3), dtype=torch.float32, device="cuda")
        else:
            ln.mine = torch.zeros((per_call, H, W, 3), dtype=torch.float32, device="cuda")
            ln.gathered = None
        lanes.append(ln)
    first_sample = 0
    if by_samples:
        first_sample, my_spp = multi_gpu.sample_range(rank, world, spp)
        base = dict(max_segments=args.max_segments, flags=gpu.FLAG_ACCUMULATE)
    elif world > 1:
        base = dict(max_segments=args.max_segments, band_rows=band, band_rank=rank, band_count=world,
                    flags=gpu.FLAG_COMPACT_BANDS)
    else:
        base = dict(max_segments=args.max_segments)
    g = lanes[0].g
    frame = None
    torch.cuda.synchronize()

    def render(ln, first_frame, count, extra_flags):
        ln.count = count
        if by_samples:
            with torch.cuda.stream(ln.stream):
                ln.mine.zero_()                        # this rank's samples are added to zeros
        o = gpu.Options(**base)
        o.flags |= extra_flags
        o.sample_offset = first_frame * spp + first_sample
        o.frames = count
        ln.g.render_region_device(ln.mine.data_ptr(), W, H, (0, 0, W, H), my_spp, args.seed, o, ln.stream.cuda_stream)

    def finish_call(ln):
        """Gather the shares of the call's frames and put every frame together (the last one stays in `frame`)."""
        nonlocal frame
        if world > 1:
            with torch.cuda.stream(ln.stream):
                flat_all, flat_mine = ln.gathered.view(-1, W, 3), ln.mine.view(-1, W, 3)
                if args.backend == "gloo":               # through host memory
                    ln.stream.synchronize()
                    host_all = torch.empty(flat_all.shape, dtype=torch.float32)
                    dist.all_gather_into_tensor(host_all, flat_mine.cpu())
                    flat_all.copy_(host_all)
                else:
                    dist.all_gather_into_tensor(flat_all, flat_mine)
                for j in range(ln.count):
                    shares = ln.gathered[:, j].reshape(-1, W, 3)          # (world * rows, W, 3)
                    if by_samples:
                        frame = multi_gpu.combine_partial_sums(shares, world, spp)
                    else:
                        frame = multi_gpu.assemble(shares, world, H, W, band)
        else:
            frame = ln.mine[ln.count - 1]

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=in_flight)

    def run_frames(first_frame, count, extra_flags=0, depth=in_flight):
        """Render frames first_frame ... first_frame + count - 1, `per_call` per call, `depth` calls in flight."""
        todo = [(f, min(per_call, first_frame + count - f)) for f in range(first_frame, first_frame + count, per_call)]
        pending = {}
        for j in range(min(depth, len(todo))):
            pending[j] = pool.submit(render, lanes[j % depth], todo[j][0], todo[j][1], extra_flags)
        for i in range(len(todo)):
            pending.pop(i).result()
            finish_call(lanes[i % depth])
            if i + depth < len(todo):
                k = i + depth
                pending[k] = pool.submit(render, lanes[k % depth], todo[k][0], todo[k][1], extra_flags)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run_frames(20_000, in_flight * per_call)                    # set-up: every lane allocates its device work lists
    fence()
    run_frames(10_000, args.warmup)
    fence()
    for ln in lanes:
        ln.g.reset_stats()
    t0 = time.perf_counter()
    run_frames(0, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    rays = float(sum(ln.g.stats()["rays"] for ln in lanes))

    # kernel durations: frames rendered one at a time (HIP events around every launch; with frames
    # overlapping, an event pair would also span the other frame's kernels)
    g.reset_stats()
    run_frames(0, 3, depth=1)
    fence()
    st = g.stats()
    frames = 3                                       # (a frame is several launches when its samples come in batches)
    fresh_ms, tree_ms, finish_ms = st["fresh_ms"] / frames, st["tree_ms"] / frames, st["finish_ms"] / frames
    resume_ms = (st["shade_ms"] - st["fresh_ms"]) / frames
    resume_launches = max(0.0, st["rounds"] / frames - 1)           # each resume launch follows one tree launch
    if world > 1:
        where = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, fresh_ms, resume_ms, tree_ms, finish_ms], dtype=torch.float64, device=where)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        r = torch.tensor([rays], dtype=torch.float64, device=where)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        elapsed, fresh_ms, resume_ms, tree_ms, finish_ms = (float(x) for x in t)
        rays = float(r[0])
    shade_ms = fresh_ms + resume_ms
    kernel_ms = shade_ms + tree_ms + finish_ms

    # algorithmic bytes of ONE frame: frame 0 again with the detailed counters on, untimed
    g.reset_stats()
    run_frames(0, 1, extra_flags=gpu.FLAG_DETAILED_STATS, depth=1)
    torch.cuda.synchronize()
    det = g.stats()
    own_pixels = H * W if by_samples else (H // world) * W
    shade_bytes, tree_bytes = algorithmic_bytes(det, own_pixels)
    abytes = shade_bytes + tree_bytes
    aflops = algorithmic_flops(det)
    checksum = float(frame.double().sum().item())

    # HBM bytes per frame as counted by the L2's memory-side request counters in a separate
    # rocprofv3 --pmc run of this command (a bench run cannot collect them itself)
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world == 1 and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("workload") == f"{args.scene} {W}x{H} {spp} spp":
            traffic, traffic_source = tj["bytes_per_frame"], tj["source"]

    if rank == 0:
        achieved = abytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s (all bounces), dog.obj 1024x1024 64 spp",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "dog model + texture from the parsed-asset fixtures in tests/golden, room scene from the "
                    "reference's main(); random numbers from the counter generator, one seed, frame i = samples i*spp ... of it",
            "config": {
                "workload": f"{args.scene} {W}x{H} {spp} spp, {args.max_segments}-segment paths" +
                            (" (config 4: dog.obj textured, BVH, indirect lighting)" if args.scene == "dog_room" else ""),
                "rays_per_step": round(rays / args.steps),
                "rays_per_sample": round(rays / args.steps / (W * H * spp), 3),
                "ms_per_frame": round(1e3 * elapsed / args.steps, 3),
                "split": ("sample ranges + all-gather + sum in rank order" if by_samples else
                          "8-row bands round-robin + all-gather") if world > 1 else "single device",
                "frames_per_call": per_call,
                "calls_in_flight": in_flight,
                "frames_in_flight": in_flight * per_call,
                "frame_checksum_seed0": checksum,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "rt_shade_kernel + rt_tree_kernel + rt_finish_kernel, all launches of one frame",
                "kernel_ms": round(kernel_ms, 3),
                "kernels": {
                    "rt_shade_kernel<true, false, *> (new paths)": {
                        "ms_per_frame": round(fresh_ms, 3), "launches_per_frame": 1,
                        "avg_launch_ms": round(fresh_ms, 3)},
                    "rt_shade_kernel<false, false, *> (resumed paths)": {
                        "ms_per_frame": round(resume_ms, 3), "launches_per_frame": round(resume_launches, 2),
                        "avg_launch_ms": round(resume_ms / max(1.0, resume_launches), 4)},
                    "rt_tree_kernel<false>": {
                        "ms_per_frame": round(tree_ms, 3), "launches_per_frame": round(resume_launches, 2),
                        "avg_launch_ms": round(tree_ms / max(1.0, resume_launches), 4)},
                    "rt_finish_kernel<false, *> (last paths, shading and trees in one launch)": {
                        "ms_per_frame": round(finish_ms, 3), "launches_per_frame": 1,
                        "avg_launch_ms": round(finish_ms, 3)},
                },
                "algorithmic_bytes_by_work": {
                    "scene primitives, root boxes, hit attributes, texels, pixels": int(shade_bytes),
                    "tree nodes and triangles": int(tree_bytes)},
                "algorithmic_bytes_per_frame": int(abytes),
                "algorithmic_bytes_per_ray": round(abytes / max(1, det["rays"]), 1),
                # the roof that binds in practice: every float operation is a separately rounded
                # vector instruction (no contraction), so a frame is priced against vector fp32 issue too
                "valu_fp32": {
                    "achieved": round(aflops / (kernel_ms * 1e-3) / 1e12, 3) if kernel_ms > 0 else 0.0,
                    "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(aflops / (kernel_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS, 4) if kernel_ms > 0 else 0.0,
                    "algorithmic_flops_per_frame": int(aflops),
                    "algorithmic_flops_per_ray": round(aflops / max(1, det["rays"]), 1),
                    "note": "flops of the reference's expressions (include/rt_work_model.h), one per add, multiply, "
                            "divide, compare or square root; a correctly rounded divide or square root costs about "
                            "ten instructions, and the peak counts a fused multiply-add as two"},
                "counted": {k: int(det[k]) for k in ("rays", "shadow_rays", "root_tests", "interior_pops",
                                                     "tri_tests", "tri_hits", "tex_lookups", "light_tests",
                                                     "sphere_tests", "plane_tests")},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sh, W, H)
            except Exception as e:  # the baseline is a side measurement; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pool.shutdown()
    for ln in lanes:
        ln.g.close()


if __name__ == "__main__":
    main()
