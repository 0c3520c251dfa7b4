#!/usr/bin/env python3
"""bench.py — headline benchmark of the render path (BASELINE.json metric: Msamples/s, path samples, max depth 8,
1920x1080).

One "step" = one pass of the hot path over one batch: a full render of config C2 (scene S1
"default+emitter" with the synthetic 1024x512 EXR-style sky, 1920x1080, 1024 spp, bounces 8, Russian roulette from
bounce 2, seed 0x0123456789ABCDEF) starting from a reset accumulator.  Inputs (scene, sky, camera) are resident on
the device before the timed region; the timed region contains the integrator launches (and, for N > 1, the RCCL
reduce of the partial radiance buffers to rank 0 — the path's one exchange step).

N > 1 (one process per GPU, launched by torch.distributed.run — or by this script itself when it is started plainly as
`python bench.py --gpus N`: the parent touches no GPU, spawns the ranks, relays their output and exit code): sample-range
partition, every rank renders the whole frame for its own 1024 sample indices (weak scaling: per-GPU work fixed); the
partial films are combined on rank 0 by the library's own RCCL step (ir_combine_async: snapshot, grouped reduce on a
stream of its own, overlapped with the next step's render); `exchange` in the JSON line reports its device time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, SPP, BOUNCES, RR = 1920, 1080, 1024, 8, 2
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
NOFMA_CEILING_TFLOPS = 78.6    # the same issue rate with one flop per lane-op (mul/add, no FMA contraction)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def flops_per_sample(st, ast=None):
    """F = 36 + sum over executed bounces [25*T + 5*D + (hit ? 112 : 25)].  Linear scan: T = N.
    LBVH (`ast` given): the *useful* count, T = leaf tests executed + 24 flop per node slab test.
    The bounce that ends in a miss pays 25 (sky lookup + accumulate); hits pay 112."""
    samples = st["samples"]
    misses = st["traces"] - st["hits"]
    total = 36 * samples + 5 * st["disc_pos"] + 112 * st["hits"] + 25 * misses
    if ast is None:
        total += 25 * st["tests"]
    else:
        total += 25 * ast["bvh_leaf_tests"] + 24 * ast["bvh_node_visits"]
    return total / samples


def executed_flops_per_step(st, pst, n_spheres, spp, launches, ast=None):
    """Flops the launches of one step really executed, in SURVEY 8d's units.  flops_per_sample charges every sample its
    own primary ray, primary closest hit and (for a pixel that sees the sky) sky lookup, as the reference runs them; the
    kernels evaluate those once per pixel and launch (no random input, pinhole_camera.rs:81-84) and finish a sky pixel
    with four adds per sample.  pst = ir_get_primary_stats (once per pixel and launch)."""
    samples = st["samples"]
    alg = flops_per_sample(st, ast) * samples
    pixels_l, path_l, d0_l = pst["pixels"], pst["path_pixels"], pst["primary_disc_pos"]
    per_launch = spp / launches              # samples a launch adds to each pixel
    sky_l = pixels_l - path_l
    saved = 36.0 * (samples - pixels_l)                                  # ray generation
    if ast is None:
        saved += 25.0 * n_spheres * (samples - pixels_l)                 # the primary scan (the tree's work is counted as executed already)
    saved += 5.0 * d0_l * (per_launch - 1.0)                             # its disc >= 0 tests
    saved += 22.0 * path_l * (per_launch - 1.0)                          # closest_hit of the primary hit
    saved += sky_l * (25.0 * per_launch - (21.0 + 4.0 * per_launch))     # sky pixels: one lookup, four adds per sample
    return alg - saved


def host_cpu():
    """(threads this process may use, CPUs the machine shows, model string).  On a shared box the usable share is the
    affinity mask capped by OMP_NUM_THREADS (the GPU boxes export their per-GPU CPU share that way)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        usable = min(usable, int(share))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return max(1, usable), os.cpu_count() or 1, model


def cpu_baseline(scene, cam_params, budget_s=12.0, max_spp=64):
    """The oracle (C++ restatement of the reference CPU path; the Rust reference cannot be built here) timed on this
    box's host cores on a bounded sample of the same workload: full 1920x1080 frame, same scene/seed/depth, fewer spp.
    Every core this process may use, plus the reference-shaped figure (4 worker threads, renderer.rs:45)."""
    from tests import oracle_lib as O
    cores, visible, model = host_cpu()
    oc = O.OracleCamera(*cam_params, W, H)
    # calibrate with 1 spp, then size the sample to about budget_s seconds
    _, _, t1, _ = O.render(oc, scene, 1, bounces=BOUNCES, rr_start=RR, threads=cores)
    spp = int(max(1, min(max_spp, budget_s / max(t1, 1e-3))))
    _, _, t, _ = O.render(oc, scene, spp, bounces=BOUNCES, rr_start=RR, threads=cores)
    _, _, t4, _ = O.render(oc, scene, max(1, spp // 4), bounces=BOUNCES, rr_start=RR, threads=4)
    return {
        "value": W * H * spp / t / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"1920x1080, {spp} spp of the workload's (same scene/seed/depth {BOUNCES}), 64x64 tiles on {cores} threads "
                  f"(every core this process may use: affinity mask capped by the box's OMP_NUM_THREADS share); "
                  f"C++ restatement of the reference integrator (linear scan), libm math",
        "cpu_model": model,
        "host_cpus_visible": visible,
        "reference_shaped_4_threads_Msamples_s": W * H * max(1, spp // 4) / t4 / 1e6,
    }


def self_launch(argv, n):
    """`python bench.py --gpus N` started plainly: spawn the N ranks with torch.distributed.run from this process, which
    has made no GPU call (and makes none), relay their output and return their exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    # (three warm-up steps = 38 ms on C2: the shader clock is still ramping during the first two launches of a process)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (the metric is quoted at 1024)")
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2 (default, the metric's config): S1 1920x1080 at 1024 spp; c3: 10k spheres at 256 spp; "
                         "c4: S1 3840x2160 at 4096 spp; c5: env-lit S1 (no emitter) 1920x1080 at 65536 spp")
    ap.add_argument("--partition", choices=["samples", "tiles"], default=None,
                    help="N > 1 only. samples (default for c2/c3/c5): every rank renders the whole frame for its own "
                         "sample indices, per-GPU spp fixed (weak scaling). tiles (default for c4): 64x64 tiles dealt "
                         "round-robin, total work fixed (strong scaling). Both end in one RCCL reduce(sum) to rank 0.")
    ap.add_argument("--accel", choices=["auto", "linear", "lbvh"], default="auto")
    ap.add_argument("--accel-build", choices=["auto", "host", "device", "sah"], default="auto",
                    help="LBVH builder (ir_set_accel_build); sah = host build with surface-area splits, for static scenes")
    ap.add_argument("--schedule", choices=["auto", "regeneration", "wavefront"], default="auto",
                    help="integrator schedule (ir_set_schedule); results are identical, only speed differs")
    ap.add_argument("--bounces", type=int, default=None,
                    help="path depth (the metric is quoted at 8; the reference's own default is 5, renderer.rs:39)")
    ap.add_argument("--sky-sampling", choices=["bsdf", "mis"], default="bsdf",
                    help="bsdf = the reference's estimator (the metric's); mis = sky importance sampling (extension: one more "
                         "trace per bounce, far lower variance under a sun-dominated sky; see tools/sky_sampling_convergence.py)")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1 only. native (default): the library's RCCL step (ir_comm_init / ir_combine_async), overlapped "
                         "with the next render; falls back to torch if RCCL cannot be initialised. torch: torch.distributed "
                         "reduce of the accumulator tensors on the render stream.")
    ap.add_argument("--sky-size", default=None, metavar="WxH",
                    help="size of the synthetic equirect sky (default 1024x512 = 8 MB on the device, cache-resident); 8192x4096 "
                         "is 537 MB as float4, twice the Infinity Cache: every lookup of an escaping path is an HBM access")
    ap.add_argument("--sky-filter", choices=["nearest", "bilinear"], default="nearest",
                    help="nearest = the reference (exr.rs:20-23) and the metric's; bilinear = north_star's variant (ir_set_sky_filter)")
    ap.add_argument("--sample-block", type=int, default=None,
                    help="ir_set_sample_block; default: IR_SAMPLE_BLOCK_AUTO (0): 124 on C2 (8 x 124 + 32), 127 on C4 and C5, 60 on C3 "
                         "(config.sample_block says what the run used)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a box with ONE GPU: every rank uses device 0 and the exchange step goes "
                         "through gloo on host copies (RCCL refuses two ranks on one device). Not a measurement.")
    args = ap.parse_args()
    global BOUNCES
    if args.bounces is not None:
        BOUNCES = args.bounces
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves (this parent never touches the GPU)
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import path_tracer_amd as pta
    from path_tracer_amd import _capi, workloads as Wk

    sky_w, sky_h = 1024, 512
    if args.sky_size:
        sky_w, sky_h = (int(v) for v in args.sky_size.lower().split("x"))
    sky_note = f"synthetic {sky_w}x{sky_h} equirect sky ({sky_w * sky_h * 16 / 1e6:.0f} MB on the device), {args.sky_filter} lookup"

    global W, H
    if args.workload == "c4":
        W, H = 3840, 2160
        spp = args.spp or 4096
        scene = Wk.scene_s1(skybox=Wk.synthetic_sky(sky_w, sky_h))
        cam = Wk.default_camera(W, H)
        wl_name = f"C4: scene S1 default+emitter + {sky_note}, {W}x{H}, {spp} spp"
    elif args.workload == "c5":
        spp = args.spp or 65536
        scene = Wk.scene_s1(emitter=False, skybox=Wk.synthetic_sky(sky_w, sky_h, sun_scale=4000.0))
        cam = Wk.default_camera(W, H)
        wl_name = f"C5: scene S1 without emitter, sun-dominated {sky_note}, {W}x{H}, {spp} spp per GPU"
    elif args.workload == "c3":
        spp = args.spp or 256
        scene = Wk.scene_s2()
        cam = Wk.camera_s2(W, H)
        wl_name = f"C3: scene S2 (10 000 random Lambertian/emissive spheres, constant sky), {W}x{H}, {spp} spp per GPU"
    else:
        spp = args.spp or SPP
        scene = Wk.scene_s1(skybox=Wk.synthetic_sky(sky_w, sky_h))
        cam = Wk.default_camera(W, H)
        wl_name = f"C2: scene S1 default+emitter (3 spheres) + {sky_note}, {W}x{H}, {spp} spp per GPU"
    rt = pta.RayTracer(0, 0, device=local_rank)
    rt.set_active_camera(cam)
    rt.set_integrator(bounces=BOUNCES, max_compulsory_bounces=RR, seed=Wk.RNG_SEED)
    rt.set_schedule({"auto": _capi.IR_SCHEDULE_AUTO, "regeneration": _capi.IR_SCHEDULE_REGENERATION,
                     "wavefront": _capi.IR_SCHEDULE_WAVEFRONT}[args.schedule])
    rt.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS if args.sky_sampling == "mis" else _capi.IR_SKY_SAMPLING_BSDF)
    rt.set_sky_filter(_capi.IR_SKY_BILINEAR if args.sky_filter == "bilinear" else _capi.IR_SKY_NEAREST)
    rt.set_accel({"auto": _capi.IR_ACCEL_AUTO, "linear": _capi.IR_ACCEL_LINEAR, "lbvh": _capi.IR_ACCEL_LBVH}[args.accel])
    rt.set_accel_build({"auto": _capi.IR_BUILD_AUTO, "host": _capi.IR_BUILD_HOST, "device": _capi.IR_BUILD_DEVICE,
                        "sah": _capi.IR_BUILD_HOST_SAH}[args.accel_build])
    rt.set_sample_block(args.sample_block if args.sample_block is not None else _capi.IR_SAMPLE_BLOCK_AUTO)
    partition = args.partition or ("tiles" if args.workload == "c4" else "samples")
    if world > 1:
        rt.set_partition(_capi.IR_PARTITION_TILES if partition == "tiles" else _capi.IR_PARTITION_SAMPLES, rank, world)
    # samples: every rank adds `spp` samples to every pixel (work grows with N); tiles: the frame is split (work fixed)
    work_ranks = world if partition == "samples" else 1

    # ---- the exchange step of N > 1
    exchange, native, native_error = "none", None, None
    if world > 1 and args.rehearse_one_gpu:
        exchange = "gloo on host copies (one-GPU rehearsal)"
    elif world > 1 and args.exchange == "native":
        from path_tracer_amd.distributed import NativeExchange
        try:
            native = NativeExchange(rt, rank, world)
        except Exception as e:  # RCCL missing or refusing: say so and use torch's
            native_error = repr(e)
        ok = torch.tensor([1 if native is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if native is not None:
                native.close()
            native = None
        exchange = "ir_combine_async (library RCCL step, overlapped with the next render)" if native else \
                   "torch.distributed reduce (native RCCL init failed: %s)" % native_error
    elif world > 1:
        exchange = "torch.distributed reduce on the render stream"
    sums = counts = None
    if native is None:
        # accumulator in torch-owned device memory so that torch's RCCL can reduce it in place, launches on torch's
        # current stream so that the reduce and the next step's accumulator reset are ordered by the stream
        sums = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        counts = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        _capi.check(rt._ctx, _capi.lib.ir_bind_accumulator(rt._ctx, sums.data_ptr(), counts.data_ptr(), W, H))
        _capi.check(rt._ctx, _capi.lib.ir_set_stream(rt._ctx, torch.cuda.current_stream().cuda_stream))

    kernel_ms = []  # (device ms of all launches of a step, launches)
    paths_ms = []   # device ms of the path kernel of a step's last launch

    def step():
        if native is not None:
            # nothing here blocks the host: render k+1 is enqueued behind the snapshot of render k and overlaps exchange k
            rt.render(scene, W, H, False, spp=spp, blocking=False)
            native.combine_async(0)
            return
        if world == 1:
            # enqueue only: the K steps queue up on the stream and the fence below waits for all of them, so the GPU does
            # not idle while the host comes back from a synchronisation and issues the next step's four launches
            rt.render(scene, W, H, False, spp=spp, blocking=False)
            return
        rt.render(scene, W, H, False, spp=spp)  # blocks until the launches are done
        kernel_ms.append(rt.last_kernel_ms())
        paths_ms.append(rt.last_paths_kernel_ms())
        if world > 1 and args.rehearse_one_gpu:
            hs, hc = sums.cpu(), counts.cpu()
            dist.reduce(hs, dst=0, op=dist.ReduceOp.SUM)
            dist.reduce(hc, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                sums.copy_(hs)
                counts.copy_(hc)
        elif world > 1:
            dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
            dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM)

    def fence():
        if native is not None or world == 1:
            rt.synchronize()
        if native is not None:
            native.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: a collective that never completes must not hang the job silently; this process ends itself with a message
    # (the launcher then ends the others) when warm-up + timed steps take longer than IR_BENCH_WATCHDOG_S (default 900 s)
    watchdog = None
    if world > 1:
        import threading
        limit = float(os.environ.get("IR_BENCH_WATCHDOG_S", "900"))

        def _stalled():
            sys.stderr.write("bench.py rank %d: no progress for %.0f s (exchange: %s); giving up\n" % (rank, limit, exchange))
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(limit, _stalled)
        watchdog.daemon = True
        watchdog.start()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms.clear()
    paths_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if world == 1:
        kernel_ms.append(rt.last_kernel_ms())  # the last step's launch sequence (the steps are identical)
        # the path kernel of every launch of the timed region (the library keeps the last 64 event pairs)
        n_timed = args.steps * max(1, kernel_ms[-1][1])
        paths_ms.extend(rt.paths_kernel_history(64)[-n_timed:])
    exchange_info = None
    if world > 1:
        exchange_info = {"how": exchange}
        if native is not None:
            kernel_ms.append(rt.last_kernel_ms())  # the last step's launch sequence (the steps are identical)
            n_timed = args.steps * max(1, kernel_ms[-1][1])
            paths_ms.extend(rt.paths_kernel_history(64)[-n_timed:])
            exchange_info["last_exchange_ms"] = native.last_ms()
            exchange_info["bytes_per_rank"] = (int(_capi.lib.ir_tile_payload_bytes(W, H, rank, world)) if partition == "tiles"
                                               else W * H * 20)
            if rank == 0:
                # the combined film must hold every rank's samples
                _, cc = rt.get_radiance()
                want = spp * (world if partition == "samples" else 1)
                exchange_info["verified_counts"] = bool((cc == want).all())

    # ---- roofline of the dominant kernel (the persistent path kernel), rank 0 only
    out = None
    if rank == 0:
        ms = [m for m, _ in kernel_ms]
        launches = sum(n for _, n in kernel_ms)
        launches_per_step = max(1, kernel_ms[0][1])
        avg_sequence_ms = sum(ms) / max(launches, 1)  # primary pass + path kernel + merge of one launch, HIP events
        # The dominant kernel is the persistent path kernel; HIP events right around it, on the stream it runs on
        avg_launch_ms = sum(paths_ms) / max(len(paths_ms), 1)
        # algorithmic flops per sample from the kernel's own counters (untimed extra render, same workload)
        if world > 1:
            rt.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
        rt.enable_stats()
        rt.render(scene, W, H, False, spp=spp)
        st = rt.get_stats()
        prof = rt.get_lane_profile()
        info = rt.get_accel_info()
        lbvh = info["accel"] == _capi.IR_ACCEL_LBVH
        ast = rt.get_accel_stats() if lbvh else None
        F = flops_per_sample(st, ast)
        samples_per_launch = W * H * spp / launches_per_step / (world if partition == "tiles" else 1)
        achieved = samples_per_launch * F / (avg_launch_ms * 1e-3) / 1e12
        total_samples = work_ranks * W * H * spp * args.steps
        pst = rt.get_primary_stats()
        n_spheres = len(scene.flat_spheres()[1])
        executed_flops = executed_flops_per_step(st, pst, n_spheres, spp, launches_per_step, ast)
        # algorithmic bytes of one launch: accumulator read-modify-write 40 B per pixel, block sums written by the
        # path kernel and read by the merge (16 B each way per task), one texel per executed sky lookup (16 B as float4;
        # bilinear: four of them)
        per_launch = spp / launches_per_step
        n_blocks = -(-int(per_launch) // rt.get_sample_block(W, H, spp))
        sky_pixels = (pst["pixels"] - pst["path_pixels"]) / launches_per_step
        lookups = ((st["traces"] - st["hits"]) / launches_per_step - sky_pixels * per_launch + sky_pixels) if scene.skybox is not None else 0.0
        texel_bytes = 64 if args.sky_filter == "bilinear" else 16
        algorithmic_bytes = (W * H * 40.0 + (pst["path_pixels"] / launches_per_step) * n_blocks * 32.0 * (n_blocks > 1)
                             + lookups * texel_bytes)
        # Counter-derived figures of this workload's launch from the round's committed rocprofv3 PMC passes
        # (profiles/r3_pmc_roofline_<workload>.json, written by tools/pmc_summary.py --json from tools/pmc_passes.sh over
        # the tree that is benchmarked): bytes at the L2's far side, lane utilisation, executed FP instruction mix, and
        # which limit the launch runs into.  Rates are formed with the launch duration measured live above.
        traffic = lane_util = executed_tf = pmc_source = l2_hit = limiter = None
        bound = "valu"
        key = args.workload + ("" if (sky_w, sky_h, args.sky_filter) == (1024, 512, "nearest") else f"_sky{sky_w}x{sky_h}_{args.sky_filter}")
        tpath = os.path.join(ROOT, "profiles", f"r3_pmc_roofline_{key}.json")
        pmc_launch_spp = {"c2": 1024, "c3": 256, "c4": 4096, "c5": 4096}  # samples per pixel of the launch the passes profiled
        # (a call of many launches cuts them at whole sample blocks: C5's are 4061 samples where the passes profiled 4096;
        # per-launch quantities are scaled by that ratio, and a launch further than a quarter off gets no counters)
        scale = per_launch / pmc_launch_spp[args.workload]
        if 0.8 <= scale <= 1.25 and BOUNCES == 8 and args.sky_sampling == "bsdf" and os.path.exists(tpath):
            with open(tpath) as f:
                pmc = json.load(f)
            traffic = pmc.get("traffic_bytes_per_launch")
            if traffic is not None:
                traffic *= scale
            lane_util = pmc.get("lane_utilisation")
            l2_hit = pmc.get("l2_hit_rate")
            bound = pmc.get("bound", "valu")
            if pmc.get("executed_flop_per_launch"):
                executed_tf = pmc["executed_flop_per_launch"] * scale / (avg_launch_ms * 1e-3) / 1e12
            pmc_source = pmc.get("source")
            limiter = pmc.get("limiter")
        hbm_gbs = algorithmic_bytes / (avg_launch_ms * 1e-3) / 1e9
        if bound == "hbm":
            roof = {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                    "valu_achieved_tflops": achieved, "valu_frac": achieved / PEAK_FP32_VALU_TFLOPS}
        else:
            roof = {"bound": bound, "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_VALU_TFLOPS, "algorithmic_gbs": hbm_gbs}
        out = {
            "metric": "Msamples/sec (path samples, max depth 8) at 1920x1080",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if partition == "samples" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_name}, bounces {BOUNCES}, rr_start {RR}, "
                            + (f"LBVH ({info['nodes']} nodes, depth {info['depth']}, build {args.accel_build})" if lbvh else "linear LDS scan"),
                "partition": "none" if world == 1 else (f"sample-range x{world}" if partition == "samples" else
                                                       f"64x64 tiles round-robin x{world}") + (" + RCCL reduce(sum) to rank 0" if partition == "samples" or native is None else " + RCCL gather of owned tiles to rank 0"),
                "seed": hex(Wk.RNG_SEED),
                "sky_sampling": args.sky_sampling,
                "sample_block": rt.get_sample_block(W, H, spp),
            },
            "roofline": {
                **roof,
                # flops all three kernels of a launch really executed (primary ray and primary closest hit once per pixel and
                # launch, sky pixels four adds per sample) / the launch sequence's device time / the FP32 vector peak
                "frac_executed": executed_flops / launches_per_step / (avg_sequence_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                "executed_flops_per_sample": executed_flops / st["samples"],
                "algorithmic_bytes_per_launch": algorithmic_bytes,
                "traffic": traffic,
                "traffic_unit": "bytes per launch at the L2's far side (FETCH_SIZE + WRITE_SIZE, separate --pmc passes): HBM "
                                "plus what the 256 MB Infinity Cache served",
                "l2_hit_rate": l2_hit,
                "limiter": limiter,
                "frac_of_nofma_ceiling": achieved / NOFMA_CEILING_TFLOPS,
                "executed_fp_tflops": executed_tf,
                "lane_utilisation": lane_util,
                "pmc_source": pmc_source,
                "kernel": "paths_lbvh_kernel<false, false>" if lbvh else "paths_kernel<false, false>",
                "avg_launch_ms": avg_launch_ms,
                "launches_averaged": len(paths_ms),
                "avg_launch_sequence_ms": avg_sequence_ms,
                "launches_per_step": launches_per_step,
                "loop_lane_utilisation": prof[list(prof)[0]][1] / max(1, 64 * prof[list(prof)[0]][0]),
                "algorithmic_flops_per_sample": F,
                "mean_traces_per_sample": st["traces"] / st["samples"],
                "brute_force_equivalent_flops_per_sample": flops_per_sample(st, None),
                "note": "FP32 vector peak 157.3 TF assumes FMA; this path is bit-exact (no FMA contraction), so its "
                        "issue ceiling is 78.6 TF (one flop per lane-op); algorithmic flops count the primary ray "
                        "per sample although the primary pass evaluates it once per pixel; avg_launch_ms = the persistent "
                        "path kernel alone (avg_launch_sequence_ms adds the primary pass and the merge); "
                        "loop_lane_utilisation = lanes holding a task at the top of the path loop / 64, counted by the "
                        "kernel's statistics build; executed_fp_tflops = "
                        "FP32 ops the kernel actually issued (ADD + MUL + TRANS + 2 FMA wave-instructions x 64 lanes x "
                        "lane utilisation, from the PMC passes) / the live launch time",
            },
        }
        if exchange_info is not None:
            out["exchange"] = exchange_info
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, (cam.position, cam.rotation, cam.focal_length, cam.sensor_size),
                                               max_spp=64 if args.workload == "c2" else 1)
        elif world == 1:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    rt.close()


if __name__ == "__main__":
    main()
