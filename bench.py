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

One process, one host thread, ONE resident scene handle.  A render call is asynchronous (the
rounds of a frame are decided on the device, nothing is read back), so the host keeps
--frames-in-flight calls under way (default 3), call j ordered on its own stream and using the
handle's work-memory lane j mod F (options.lane): the latency-bound end of one frame runs beside
the start of the next and, with N > 1, the all-gather overlaps rendering.  With N > 1 a call
renders this rank's share of --frames-per-call consecutive frames (default N, at most 8; launches as
large as a whole frame's, one gather for all of them).  Every timed frame is complete (rendered,
gathered, assembled) inside the timed region; ms_per_step is elapsed / K.  `ms_per_frame_single`
is one frame alone (one call, synchronised), `scene_create_ms` the flatten + upload of the scene.

Rank 0 prints one JSON line.  A frame is rendered by kernels that alternate: rt_shade_kernel
(path bookkeeping, sampling, the scene's lights/spheres/planes, instance root boxes),
rt_tree_kernel (mesh tree walks for the rays the shade kernel parked) and rt_finish_kernel (the
last paths).  `roofline` prices them against the roof that binds this path, vector instruction
issue: wave-level vector instructions per second (instruction counts per launch kind from a
rocprofv3 SQ_INSTS_VALU pass of this same command, committed under profiles/, divided by the
kernels' durations measured live with HIP events on their streams) against the issue peak of the
chip (one 64-lane vector instruction per SIMD every 2 cycles).  `roofline.hbm` is the secondary
entry: HBM bytes from the memory-side counters (FETCH_SIZE corrected x2, WRITE_SIZE) over the same
time against 8 TB/s, next to the ALGORITHMIC bytes of the path (bytes the reference's records
would move for the primitives, boxes, triangles, attributes and texels the kernels tested,
counted on the device per launch kind).
`cpu_baseline` times the reference itself (compiled from its own sources with its build.sh
flags, oracle/_ref) on this machine's host cores over a bounded number of samples of the same
scene, at the reference's THREAD_COUNT (15) and at all cores available to the job; when that
library is absent it times the CPU oracle port instead.
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X vector fp32 spec peak (256 CUs x 128 lanes x 2 x 2.4 GHz)
# vector issue: a 64-lane instruction occupies a SIMD-32 for 2 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 2
SIMD_COUNT = 256 * 4
SHADER_CLOCK_HZ = 2.4e9
KINDS = ("new_paths", "resumed_paths", "tree", "finish")
KERNEL_OF = {"new_paths": "rt_shade_kernel<true, false, *>", "resumed_paths": "rt_shade_kernel<false, false, *>",
             "tree": "rt_tree_kernel<false>", "finish": "rt_finish_kernel<false, *>"}


def work_model():
    """The per-event byte and flop constants of include/rt_work_model.h."""
    import re
    with open(os.path.join(ROOT, "include", "rt_work_model.h")) as f:
        return {k: int(v) for k, v in re.findall(r"#define\s+(RT_(?:BYTES|FLOPS)_\w+)\s+(\d+)", f.read())}


def algorithmic_bytes(st, pixels=0):
    """Bytes the reference's records would move for the work one kind of launch counted: lights,
    spheres, planes, instance root tests, triangle-hit attributes, texels, the two child nodes per
    interior pop, the triangle with its three points per triangle test, and the pixel store."""
    m = work_model()
    return (m["RT_BYTES_LIGHT_TEST"] * st["light_tests"] + m["RT_BYTES_SPHERE_TEST"] * st["sphere_tests"] +
            m["RT_BYTES_PLANE_TEST"] * st["plane_tests"] + m["RT_BYTES_ROOT_TEST"] * st["root_tests"] +
            m["RT_BYTES_TRI_HIT"] * st["tri_hits"] + m["RT_BYTES_TEX_LOOKUP"] * st["tex_lookups"] +
            m["RT_BYTES_INTERIOR_POP"] * st["interior_pops"] + m["RT_BYTES_TRI_TEST"] * st["tri_tests"] +
            m["RT_BYTES_PIXEL"] * pixels)


def algorithmic_flops(st):
    """Flops of the reference's expressions for the same counted work (transcendentals excluded)."""
    m = work_model()
    return (m["RT_FLOPS_LIGHT_TEST"] * st["light_tests"] + m["RT_FLOPS_SPHERE_TEST"] * st["sphere_tests"] +
            m["RT_FLOPS_PLANE_TEST"] * st["plane_tests"] + m["RT_FLOPS_ROOT_TEST"] * st["root_tests"] +
            m["RT_FLOPS_INTERIOR_POP"] * st["interior_pops"] + m["RT_FLOPS_TRI_TEST"] * st["tri_tests"] +
            m["RT_FLOPS_TRI_HIT"] * st["tri_hits"] +
            m["RT_FLOPS_PATH_SEGMENT"] * (st["rays"] - st["shadow_rays"]))


def host_cores():
    """Cores this job may use: the scheduler affinity, capped by the share the environment declares
    for the job (OMP_NUM_THREADS on a shared machine whose affinity mask shows every core)."""
    n = len(os.sched_getaffinity(0))
    declared = os.environ.get("OMP_NUM_THREADS", "")
    if declared.isdigit() and 0 < int(declared) < n:
        return int(declared), f"{n} in the affinity mask, {declared} declared as this job's share (OMP_NUM_THREADS)"
    return n, f"{n} in the affinity mask"


def cpu_baseline(sh, width, height, target_seconds=7.0):
    """Time the CPU path on this host over a bounded sample of the workload, at the reference's own
    thread count (THREAD_COUNT 15, src/main.h:76) and at every core the job has."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L
    from raytracing_amd import host
    use_ref = L.have_ref("build")
    tile = 64
    tiles = ((width + tile - 1) // tile) * ((height + tile - 1) // tile)
    seeds = (L.T.RandomSerie * tiles)(*[L.serie_from_seed(1000 + i) for i in range(tiles)])

    def run(spp, threads):
        img, _ = host.new_image(width, height)
        if use_ref:
            return L.ref("build").ref_render_tiles(sh.ptr, C.byref(img), spp, tile, threads, seeds)
        return L.oracle(contract=True).orc_render_tiles(sh.ptr, C.byref(img), spp, tile, threads,
                                                        L.RNG_AES, L.MATH_LIBM, 64, seeds, 0, None)

    # rays per sample of this scene, counted by the oracle on a small frame (same scene, same path rules)
    st = L.Stats()
    small, _ = host.new_image(128, 128)
    L.oracle().orc_render_tiles(sh.ptr, C.byref(small), 4, 64, 8, L.RNG_AES, L.MATH_LIBM, 64, seeds, 0, C.byref(st))
    rays_per_sample = st.rays / max(1, st.samples)

    def measure(threads):
        t1 = run(1, threads)
        spp = int(max(1, min(64, round(target_seconds / max(t1, 1e-3)))))
        t = run(spp, threads) if spp > 1 else t1
        samples = width * height * spp
        return {"value": round(samples * rays_per_sample / t / 1e6, 3), "unit": "Mrays/s", "cores": threads,
                "sample": f"{width}x{height} at {spp} spp ({samples / 1e6:.2f} Msamples in {t:.2f} s, "
                          f"{samples / t / 1e6:.3f} Msamples/s), 64x64 tiles on an atomic queue over {threads} threads, render only",
                "ms_per_frame_64spp_extrapolated": round(1e3 * t * 64 / spp, 1)}

    cores, how = host_cores()
    full = measure(cores)
    out = dict(full)
    out["kind"] = "reference" if use_ref else "port"
    out["sample"] += f"; {rays_per_sample:.2f} rays/sample counted by the oracle; host cores: {how}"
    out["at_reference_thread_count"] = measure(15)      # THREAD_COUNT of the reference (src/main.h:76)
    return out


def load_counters(args, W, H, spp):
    """Per-frame hardware counters of this workload from rocprofv3 --pmc passes of this command
    (tools/profile_round.sh -> tools/make_counters_json.py); a bench run cannot collect them itself."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        cj = json.load(f)
    return cj if cj.get("workload") == f"{args.scene} {W}x{H} {spp} spp" else None


def cycle_estimate(c, seconds):
    """Share of the vector units' cycles that the counted instructions of one kind of launch fill, with the kernel's static
    mix priced by measured issue costs (profiles/isa_cycles.json: 2 / 4 / 8 cycles per 64-lane instruction by opcode and
    operand kind): instructions x cycles per instruction / (launch time x 1024 SIMDs x 2.4 GHz).  An estimate: the static
    mix of the kernel stands in for the executed one."""
    path = os.path.join(ROOT, "profiles", "isa_cycles.json")
    if not os.path.exists(path) or not c.get("kernel_names") or seconds <= 0:
        return {}
    with open(path) as f:
        table = json.load(f)["kernels"]
    cpis = [table[n]["cycles_per_instruction"] for n in c["kernel_names"] if n in table]
    if not cpis:
        return {}
    cpi = sum(cpis) / len(cpis)
    return {"valu_cycles_per_instruction_static": round(cpi, 3),
            "valu_cycle_frac_estimate": round(c["valu_wave_instructions"] * cpi / (seconds * SIMD_COUNT * SHADER_CLOCK_HZ), 4)}


def instruction_mix(c, rays):
    """Vector instructions of one kind of launch per ray of the frame, from the SQ_INSTS_VALU* counters: 64-lane issue
    slots (wave instructions x 64), the slots on active lanes, and the slots by class (f32 add / multiply / fma come almost
    entirely from the expansions of correctly rounded divisions and square roots and from the dot products; "other" is
    what no class counter sees: moves, selects, compares, lane exchanges)."""
    total = c["valu_wave_instructions"]
    per_ray = lambda n: round(64.0 * n / max(1.0, rays), 1)
    classes = {"f32_add": c.get("SQ_INSTS_VALU_ADD_F32", 0.0), "f32_mul": c.get("SQ_INSTS_VALU_MUL_F32", 0.0),
               "f32_fma": c.get("SQ_INSTS_VALU_FMA_F32", 0.0),
               "f64": c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_FMA_F64", 0.0),
               "int32": c.get("SQ_INSTS_VALU_INT32", 0.0), "transcendental": c.get("SQ_INSTS_VALU_TRANS_F32", 0.0),
               "convert": c.get("SQ_INSTS_VALU_CVT", 0.0)}
    out = {k: per_ray(v) for k, v in classes.items()}
    out["other_move_select_compare_exchange"] = per_ray(max(0.0, total - sum(classes.values())))
    return {"valu_lane_slots_per_ray": per_ray(total),
            "valu_lane_instructions_per_ray": round(total * (c.get("lanes_active") or 64.0) / max(1.0, rays), 1),
            "valu_lane_slots_per_ray_by_class": out}


def secondary_scene(name, W, H, spp, seed, max_segments):
    """One more scene beside the headline, frames one at a time on a handle of its own: the scene of the largest
    configuration (dog + cat meshes, four instances, mirror sphere), where the tree walk is most of the frame.  Returns the
    entry for the JSON line: frame time, rays per second, device time per kind of launch and, when the counter passes of
    this scene are committed (profiles/counters_<scene>.json), the same issue and memory fractions as the headline."""
    import torch
    from raytracing_amd import gpu, scenes
    sh = scenes.CONFIGS[name]()
    g = gpu.GpuScene(sh)
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    o = gpu.Options(max_segments=max_segments)
    times = []
    for k in range(4):                                   # the first call allocates the lane's work memory
        torch.cuda.synchronize()
        if k == 1:
            g.reset_stats()
        t1 = time.perf_counter()
        g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, seed, o)
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t1))
    st = g.stats()
    rays = st["rays"] / 3.0
    ms = sorted(times[1:])[1]
    g.reset_stats()
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, seed, gpu.Options(max_segments=max_segments, flags=gpu.FLAG_KERNEL_TIMES))
    torch.cuda.synchronize()
    by_kind = g.stats_by_kernel()
    rounds = g.stats()["rounds"]
    out = {"workload": f"{name} {W}x{H} {spp} spp, {max_segments}-segment paths (the scene of config 5: dog + cat meshes + mirrors)",
           "ms_per_frame_single": round(ms, 3), "value": round(rays / (ms * 1e-3) / 1e6, 2), "unit": "Mrays/s",
           "rays_per_frame": round(rays), "rays_per_sample": round(rays / (W * H * spp), 3),
           "shade_launches_with_work": int(rounds),
           "frame_checksum": float(buf.double().sum().item()),
           "work_memory_bytes": int(g.info()["work_bytes"]),
           "kernels": {k: {"ms_per_frame": round(by_kind[k]["kernel_ms"], 3)} for k in KINDS}}
    path = os.path.join(ROOT, "profiles", f"counters_{name}.json")
    if os.path.exists(path):
        with open(path) as f:
            cj = json.load(f)
        if cj.get("workload") == f"{name} {W}x{H} {spp} spp":
            out["counters_source"] = cj["source"]
            for k in KINDS:
                c, e = cj["kinds"].get(k), out["kernels"][k]
                if not c or e["ms_per_frame"] <= 0:
                    continue
                sec = e["ms_per_frame"] * 1e-3
                hbm = 2 * c["fetch_bytes_raw"] + c["write_bytes"]
                e.update({"valu_issue_frac": round(c["valu_wave_instructions"] / sec / 1e9 / VALU_ISSUE_PEAK_GINST, 4),
                          "lanes_active_per_valu_instruction": c.get("lanes_active"),
                          "hbm_GBs": round(hbm / sec / 1e9, 1), "hbm_frac": round(hbm / sec / 1e9 / HBM_PEAK_GBS, 4)})
                e.update(instruction_mix(c, rays))
                e.update(cycle_estimate(c, sec))
    g.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--scene", default="dog_room")
    ap.add_argument("--max-segments", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-measurements", action="store_true",
                    help="skip the frames rendered beside the benchmark (the surface-area tree): profiling runs count only the benchmark's own launches")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend: nccl (= RCCL over xGMI, one GPU per rank); gloo gathers through host "
                         "memory and lets several ranks share a GPU (a functional check of the multi-process path on a "
                         "machine with fewer GPUs than ranks, not a measurement)")
    ap.add_argument("--split", choices=("bands", "samples"), default="bands",
                    help="how N devices share a frame: 8-row bands (bit-identical to one device) or sample ranges of "
                         "every pixel, summed in rank order (equal up to float summation order; for small frames)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-process path (process group, packed bands, all-gather, assembly, reductions) even "
                         "with one rank: a functional check of the RCCL calls on a one-GPU machine")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="frames whose shares one render call produces (0 = 1 on one GPU, min(N, 8) when N GPUs share a frame)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="render calls under way at once, each ordered on its own stream and using its own work-memory "
                         "lane of the one scene handle (0 = 4, the hardware queues of a process)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    in_flight = args.frames_in_flight if args.frames_in_flight > 0 else 4
    os.environ.setdefault("RT_GPU_STREAMS", str(max(3, in_flight)))      # lanes of the scene handle (read when it is made)
    import torch
    from raytracing_amd import gpu, multi_gpu, scenes

    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())      # ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    shared = world > 1 or args.force_dist              # the frame goes through the band split and the collective
    if shared:
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    W, H, spp = args.width, args.height, args.spp
    band = multi_gpu.BAND_ROWS
    t0 = time.perf_counter()
    sh = scenes.CONFIGS[args.scene]()                  # parsed model arrays -> triangulated mesh + box tree, texture bytes -> linear floats (host C)
    host_assets_ms = 1e3 * (time.perf_counter() - t0)
    by_samples = shared and args.split == "samples"
    my_spp = spp
    # Frame i of a run = samples i*spp ... i*spp + spp - 1 of every pixel (one seed): independent frames
    # that are the same pictures whatever N is.  When several GPUs share a frame, a call renders this
    # rank's share of `per_call` consecutive frames at once (options.frames): launches as large as a whole
    # frame's instead of 1/N of it, one gather for all of them.
    if args.frames_per_call > 0:
        per_call = args.frames_per_call
    else:
        per_call = 1 if (not shared or by_samples) else min(world, 8)
    if by_samples:
        per_call = 1

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = gpu.GpuScene(sh)                               # flatten + upload: outside the timed region, reported
    torch.cuda.synchronize()
    scene_create_first_ms = 1e3 * (time.perf_counter() - t0)   # the process's first use of the library: includes loading its code object
    t0 = time.perf_counter()
    again = gpu.GpuScene(sh)
    torch.cuda.synchronize()
    scene_create_ms = 1e3 * (time.perf_counter() - t0)
    create_info = again.info()
    again.close()

    # one slot per call in flight: a stream to order the call on, output buffers, work-memory lane `slot` of the handle
    class Slot:
        pass
    slots = []
    for k in range(in_flight):
        sl = Slot()
        sl.index = k
        sl.stream = torch.cuda.Stream()
        if shared:
            rows = H if by_samples else multi_gpu.packed_rows(world, H, band)
            sl.mine = torch.zeros((per_call, rows, W, 3), dtype=torch.float32, device="cuda")
            sl.gathered = torch.zeros((world, per_call, rows, W, 3), dtype=torch.float32, device="cuda")
        else:
            sl.mine = torch.zeros((per_call, H, W, 3), dtype=torch.float32, device="cuda")
            sl.gathered = None
        slots.append(sl)
    first_sample = 0
    if by_samples:
        first_sample, my_spp = multi_gpu.sample_range(rank, world, spp)
        base = dict(max_segments=args.max_segments, flags=gpu.FLAG_ACCUMULATE)
    elif shared:
        base = dict(max_segments=args.max_segments, band_rows=band, band_rank=rank, band_count=world,
                    flags=gpu.FLAG_COMPACT_BANDS)
    else:
        base = dict(max_segments=args.max_segments)
    frame = None
    torch.cuda.synchronize()

    def render(sl, first_frame, count, extra_flags):
        """Enqueue the call (asynchronous) and, when several GPUs share the frames, the gather behind it."""
        nonlocal frame
        sl.count = count
        sl.first_frame = first_frame
        sl.frames_out = []
        if by_samples:
            with torch.cuda.stream(sl.stream):
                sl.mine.zero_()                        # this rank's samples are added to zeros
        o = gpu.Options(**base)
        o.flags |= extra_flags
        o.sample_offset = first_frame * spp + first_sample
        o.frames = count
        if shared and not by_samples:
            o.frame_stride = sl.mine.shape[1] * W * 3      # frames lie one packed buffer apart (the rank with the most rows), whatever this rank owns
        o.lane = sl.index
        g.render_region_device(sl.mine.data_ptr(), W, H, (0, 0, W, H), my_spp, args.seed, o, sl.stream.cuda_stream)
        if shared:
            with torch.cuda.stream(sl.stream):
                flat_all, flat_mine = sl.gathered.view(-1, W, 3), sl.mine.view(-1, W, 3)
                if args.backend == "gloo":               # through host memory
                    sl.stream.synchronize()
                    host_all = torch.empty(flat_all.shape, dtype=torch.float32)
                    dist.all_gather_into_tensor(host_all, flat_mine.cpu())
                    flat_all.copy_(host_all)
                else:
                    dist.all_gather_into_tensor(flat_all, flat_mine)
                for j in range(sl.count):
                    shares = sl.gathered[:, j].reshape(-1, W, 3)          # (world * rows, W, 3)
                    if by_samples:
                        frame = multi_gpu.combine_partial_sums(shares, world, spp)
                    else:
                        frame = multi_gpu.assemble(shares, world, H, W, band)
                    sl.frames_out.append(frame)
        else:
            sl.frames_out = [sl.mine[j] for j in range(sl.count)]
            frame = sl.mine[sl.count - 1]

    def run_frames(first_frame, count, extra_flags=0, depth=in_flight):
        """Render frames first_frame ... first_frame + count - 1, `per_call` per call, `depth` calls in flight
        (call j waits, through its stream, for call j - depth, whose buffers and lane it takes over)."""
        j = 0
        for f in range(first_frame, first_frame + count, per_call):
            render(slots[j % depth], f, min(per_call, first_frame + count - f), extra_flags)
            j += 1

    def fence():
        torch.cuda.synchronize()
        if shared:
            dist.barrier()
            torch.cuda.synchronize()

    run_frames(20_000, in_flight * per_call)                    # set-up: every lane allocates its device work memory
    fence()
    run_frames(10_000, args.warmup)
    fence()
    g.reset_stats()
    t0 = time.perf_counter()
    run_frames(0, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    timed_stats = g.stats()
    rays = float(timed_stats["rays"])
    # the frames of the timed run that are still in the slots' buffers (the last calls in flight), by frame index:
    # frame i is samples i*spp ... of seed --seed whatever N and the frames per call are, so the sums must agree
    timed_checksums = {}
    for sl in slots:
        if getattr(sl, "first_frame", 10**9) < args.steps:
            for j, f in enumerate(sl.frames_out):
                timed_checksums[str(sl.first_frame + j)] = float(f.double().sum().item())

    # one frame alone: a call, then wait for it
    single = []
    for k in range(5):
        fence()
        t1 = time.perf_counter()
        run_frames(k, 1, depth=1)
        fence()
        single.append(1e3 * (time.perf_counter() - t1))
    ms_single = sorted(single)[len(single) // 2]

    # kernel durations: frames rendered one at a time with HIP events around every launch
    g.reset_stats()
    frames = 3
    for k in range(frames):
        run_frames(k, 1, extra_flags=gpu.FLAG_KERNEL_TIMES, depth=1)
        fence()
    timed = g.stats_by_kernel()
    kind_ms = [timed[k]["kernel_ms"] / frames for k in KINDS]
    launches_with_work = g.stats()["rounds"] / frames            # shade launches that had work (first one included)
    if shared:
        where = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, ms_single] + kind_ms, dtype=torch.float64, device=where)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        r = torch.tensor([rays], dtype=torch.float64, device=where)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        elapsed, ms_single = float(t[0]), float(t[1])
        kind_ms = [float(x) for x in t[2:]]
        rays = float(r[0])
    kernel_ms = sum(kind_ms)

    # algorithmic bytes of ONE frame per kind of launch: frame 0 again with the detailed counters on, untimed
    g.reset_stats()
    run_frames(0, 1, extra_flags=gpu.FLAG_DETAILED_STATS, depth=1)
    torch.cuda.synchronize()
    det = g.stats()
    det_kind = g.stats_by_kernel()
    own_pixels = H * W if (by_samples or not shared) else len(multi_gpu.rows_of_rank(rank, world, H, band)) * W
    abytes_kind = [algorithmic_bytes(det_kind[k], own_pixels if k == "new_paths" else 0) for k in KINDS]
    abytes = sum(abytes_kind)
    aflops = algorithmic_flops(det)
    checksum = float(frame.double().sum().item())

    # side measurement, never the headline: the same frame with the opt-in surface-area tree (other boxes, same triangles)
    sah = None
    if not shared and not args.no_side_measurements and args.scene in ("dog_room", "three_dogs", "dogs_and_cat"):
        t1 = time.perf_counter()
        sh2 = scenes.CONFIGS[args.scene](tree="sah", leaf_triangles=8)
        sah_assets_ms = 1e3 * (time.perf_counter() - t1)
        g2 = gpu.GpuScene(sh2)
        buf2 = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        times = []
        for k in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            g2.render_region_device(buf2.data_ptr(), W, H, (0, 0, W, H), spp, args.seed, gpu.Options(max_segments=args.max_segments))
            torch.cuda.synchronize()
            times.append(1e3 * (time.perf_counter() - t1))
        sah = {"ms_per_frame_single": round(sorted(times[1:])[1], 3), "host_assets_ms": round(sah_assets_ms, 1),
               "frame_checksum_seed0": float(buf2.double().sum().item()),
               "note": "rt_host_mesh_build_tree_sah, 8 triangles per leaf: not the reference's tree (pictures agree except where "
                       "the tree shape is observable); compare with ms_per_frame_single"}
        g2.close()

    secondary = None
    if not shared and not args.no_side_measurements and args.scene == "dog_room":
        try:
            secondary = secondary_scene("dogs_and_cat", W, H, spp, args.seed, args.max_segments)
        except Exception as e:      # a side measurement: never lose the headline line over it
            secondary = {"workload": "dogs_and_cat", "failed": str(e)}
    model = {"dog_room": "dog.obj", "three_dogs": "dog.obj x3", "dogs_and_cat": "dog.obj + cat.obj"}.get(args.scene, args.scene)
    metric_name = f"Mrays/s (all bounces), {model} {W}x{H} {spp} spp"
    if rank == 0:
        counters = load_counters(args, W, H, spp) if not shared else None
        resumed_launches = max(0.0, launches_with_work - 1)
        launches = {"new_paths": 1.0, "resumed_paths": resumed_launches, "tree": resumed_launches, "finish": 1.0}
        kernels = {}
        for i, k in enumerate(KINDS):
            e = {"kernel": KERNEL_OF[k], "ms_per_frame": round(kind_ms[i], 3),
                 "launches_with_work_per_frame": round(launches[k], 2),
                 "avg_launch_ms": round(kind_ms[i] / max(1.0, launches[k]), 4),
                 "algorithmic_bytes_per_frame": int(abytes_kind[i]),
                 "algorithmic_GBs": round(abytes_kind[i] / max(1e-9, kind_ms[i] * 1e-3) / 1e9, 1)}
            if counters and k in counters["kinds"]:
                c = counters["kinds"][k]
                e.update(instruction_mix(c, float(det["rays"])))
                ginst = c["valu_wave_instructions"] / max(1e-9, kind_ms[i] * 1e-3) / 1e9
                hbm = (2 * c["fetch_bytes_raw"] + c["write_bytes"])
                e.update({"valu_wave_instructions_per_frame": int(c["valu_wave_instructions"]),
                          "valu_issue_Ginst_s": round(ginst, 1), "valu_issue_frac": round(ginst / VALU_ISSUE_PEAK_GINST, 4),
                          "lanes_active_per_valu_instruction": c.get("lanes_active"),
                          "hbm_bytes_per_frame": int(hbm), "hbm_GBs": round(hbm / max(1e-9, kind_ms[i] * 1e-3) / 1e9, 1),
                          "hbm_frac": round(hbm / max(1e-9, kind_ms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
                e.update(cycle_estimate(c, kind_ms[i] * 1e-3))
            kernels[k] = e
        # the dominant kernel: the kind whose launches are the longest (its average launch with work); by share of the
        # frame the launch that starts the paths and the six tree launches together are within a few percent of each other
        dominant = max(KINDS, key=lambda k: kernels[k]["avg_launch_ms"])
        dk = kernels[dominant]
        total_inst = sum(c["valu_wave_instructions"] for c in counters["kinds"].values()) if counters else None
        total_hbm = sum(2 * c["fetch_bytes_raw"] + c["write_bytes"] for c in counters["kinds"].values()) if counters else None
        roofline = {
            "bound": "valu_issue",
            "kernel": f"{dk['kernel']} ({dominant}): the kind of launch with the longest launches "
                      f"({dk['avg_launch_ms']:.2f} ms each, {dk['ms_per_frame']:.2f} ms of the frame's {kernel_ms:.2f} ms)",
            "achieved": dk.get("valu_issue_Ginst_s"),
            "peak": round(VALU_ISSUE_PEAK_GINST, 1),
            "unit": "G wave-instructions/s (64-lane vector instructions issued)",
            "frac": dk.get("valu_issue_frac"),
            "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per 64-lane instruction.  Measured per opcode "
                         "(tools/micro/issue_cost.hip, profiles/r3c_issue_cost.txt): only fp32 / integer add, subtract, multiply, fma, "
                         "move, and, or, xor on vector-register operands issue in 2 cycles (1060-1190 G/s); compares, selects, min / max, "
                         "conversions, fp64, cross-lane operations and every instruction with a scalar-register operand take 4 "
                         "(600 G/s), transcendentals 8 (300 G/s).  valu_cycle_frac_estimate prices the kernel's static mix with "
                         "those costs",
            "valu_cycle_frac_estimate": dk.get("valu_cycle_frac_estimate"),
            "traffic": dk.get("hbm_bytes_per_frame"),
            "counters_source": counters["source"] if counters else None,
            "kernel_ms": round(kernel_ms, 3),
            "frame": {"valu_wave_instructions": total_inst,
                      "valu_issue_Ginst_s": round(total_inst / (kernel_ms * 1e-3) / 1e9, 1) if total_inst and kernel_ms > 0 else None,
                      "valu_issue_frac": round(total_inst / (kernel_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINST, 4) if total_inst and kernel_ms > 0 else None},
            "kernels": kernels,
            "hbm": {"bound": "hbm", "achieved": round(total_hbm / (kernel_ms * 1e-3) / 1e9, 1) if total_hbm and kernel_ms > 0 else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(total_hbm / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if total_hbm and kernel_ms > 0 else None,
                    "traffic": total_hbm,
                    "note": "counter traffic (FETCH_SIZE x 2 + WRITE_SIZE) over the kernels' time: the scene is cache-resident, "
                            "HBM carries parked records and per-sample values",
                    "algorithmic_bytes_per_frame": int(abytes),
                    "algorithmic_bytes_per_ray": round(abytes / max(1, det["rays"]), 1),
                    "algorithmic_GBs": round(abytes / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None},
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
        }
        out = {
            "metric": metric_name,
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
            "data": "dog model + texture from the parsed-asset files in raytracing_amd/assets, room scene from the "
                    "reference's main(); random numbers from the counter generator, one seed, frame i = samples i*spp ... of it",
            "ms_per_frame_single": round(ms_single, 3),
            "scene_create_ms": round(scene_create_ms, 2),
            "scene_create": {"flatten_ms": round(create_info["flatten_ms"], 2), "upload_ms": round(create_info["upload_ms"], 2),
                             "resident_bytes": int(create_info["resident_bytes"]),
                             "first_call_in_process_ms": round(scene_create_first_ms, 1),
                             "host_assets_ms": round(host_assets_ms, 1),
                             "note": "scene_create_ms: rt_gpu_scene_create of a second handle (flatten on host threads + device "
                                     "allocation and copies); the first call of a process also loads the library's code object; "
                                     "host_assets_ms: model arrays -> mesh with its box tree, texture bytes -> linear floats, plus reading the asset files"},
            "config": {
                "workload": f"{args.scene} {W}x{H} {spp} spp, {args.max_segments}-segment paths" +
                            (" (config 4: dog.obj textured, BVH, indirect lighting)" if args.scene == "dog_room" else ""),
                "rays_per_step": round(rays / args.steps),
                "rays_per_sample": round(rays / args.steps / (W * H * spp), 3),
                "ms_per_frame": round(1e3 * elapsed / args.steps, 3),
                "split": ("sample ranges + all-gather + sum in rank order" if by_samples else
                          "8-row bands round-robin + all-gather") if shared else "single device",
                "scene_handles": 1,
                "host_threads": 1,
                "frames_per_call": per_call,
                "calls_in_flight": in_flight,
                "frames_in_flight": in_flight * per_call,
                "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                "frame_checksum_seed0": checksum,
                "frame_checksums_timed": timed_checksums,
                "passes_per_batch": round(timed_stats["passes_with_work"] / max(1, timed_stats["launches"]), 2),
                "passes_enqueued_per_batch": round(timed_stats["passes"] / max(1, timed_stats["launches"]), 2),
                "parked_lists": "sized for every path of a batch" if timed_stats["lists_full"] else
                                "the handle's starting size (RT_GPU_PARKED_PERCENT, default 40 % of the worst case); no batch needed a second pass",
                "work_memory_bytes": int(g.info()["work_bytes"]),
            },
            "roofline": roofline,
        }
        if sah:
            out["surface_area_tree"] = sah
        if secondary:
            out["secondary"] = secondary
        if not shared and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sh, W, H)
            except Exception as e:  # the baseline is a side measurement; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if shared:
        dist.barrier()
        dist.destroy_process_group()
    g.close()


if __name__ == "__main__":
    main()
