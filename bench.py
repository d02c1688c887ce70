#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (and samples/pixel/s) of the wavefront path tracer.

Workload (BASELINE.json configs[3], the 10M-triangle scene the metric is quoted on): the
instanced teapot_lowpoly grid baked into one 10 036 800-triangle mesh / one BVH, 3840x2160,
Disney rough material + one rectangular area light, primary + secondary rays, 30 bounces max.
One "step" = one batch of --spp-per-step (default 64: the configuration's 64 spp as ONE batch, 0.65 s, 151 GB of path pool) samples per pixel of the whole frame through the full hot path
(init -> {extend, shade, connect} x bounces -> resolve) with the scene resident in HBM.

N GPUs: one process per GPU (torchrun; torch.distributed with the gloo backend is the control plane only: rendezvous, barriers,
the max-over-ranks clock).  Rank 0 builds the scene and its BVH ONCE and saves the flattened arrays (mi355pt_scene_save); every
rank maps that file and uploads it (mi355pt_scene_load).  The work is split with no data-path collective -- weak scaling
(default: per-GPU work fixed) or strong (--scaling strong: the step's samples are divided among the GPUs) -- either by sample
index (default: GPU g renders samples g, g+N, ... of every pixel) or by image tile (--split tile: GPU g renders the 64x64 tiles
k with k % N == g).  The per-GPU running means are combined by ONE RCCL reduce inside the library (mi355pt_comm_* +
mi355pt_render_reduce, librccl of the ROCm the library is linked against) at the end of the timed region; rank 0 checks the
combined sample counts.  torch never touches the GPU here: the process holds one HIP runtime.

Prints ONE JSON line (rank 0).
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, MI355X_MICROARCH.md
NODE_BYTES, TRI_BYTES, QUEUE_BYTES = 64, 48, 96      # bytes/ray = 64 n_node + 48 n_tri + 96


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_scene(args):
    from pathtracer_amd import scenes
    if args.workload == "c4":
        return scenes.grid_c4(args.grid[0], args.grid[1], args.width, args.height), \
            f"C4 instanced teapot_lowpoly grid {args.grid[0]}x{args.grid[1]} baked into one mesh"
    if args.workload == "c1":
        return scenes.cornell_c1(args.width, args.height), "C1 assets/cornell.xml Cornell (2 planes + rect light)"
    if args.workload == "c2":
        return scenes.textured_c2(args.width, args.height), "C2 teapot.obj + cat.obj textured Disney"
    if args.workload == "c3":
        return scenes.cornell_c3(args.width, args.height), "C3 Cook-Torrance Cornell + DOF"
    if args.workload == "c5":
        return scenes.soup_c5(args.tris, args.width, args.height), f"C5 randomized triangle soup, {args.tris} triangles"
    raise SystemExit("unknown workload")


def cpu_baseline(scene, pt, args):
    """The oracle in reference mode (exhaustive unordered BVH walk, pixel-strided std::threads) timed
    on a bounded crop of the same workload.  The oracle imports the BVH the product built (verified
    node-for-node against the oracle's literal builder on small meshes; the literal 153-sweep builder
    is O(153 N depth) and not runnable at 10 M triangles)."""
    from oracle.binding import OracleScene
    from pathtracer_amd import capi
    mi = 0
    for n in scene.rootNode.children:
        if n.obj is not None and getattr(n.obj, "meshes", None):
            for m in n.obj.meshes:
                nodes, faces, nn = pt.dscene.get_bvh(mi)
                m.bvh_nodes = (capi.BvhNode * nn).from_buffer_copy(bytes(nodes)[:nn * C.sizeof(capi.BvhNode)])
                m.bvh_faces = np.ascontiguousarray(faces)
                mi += 1
    orc = OracleScene(scene)
    threads = max(1, min(16, (os.cpu_count() or 2) - 1))
    cw, ch = min(args.cpu_crop[0], args.width), min(args.cpu_crop[1], args.height)
    x0, y0 = (args.width - cw) // 2, (args.height - ch) // 2
    prm = capi.make_params(args.width, args.height, seed=args.seed, max_bounces=args.max_bounces)
    t0 = time.perf_counter()
    _, _, _, st = orc.render(prm, args.cpu_spp, rect=(x0, y0, x0 + cw, y0 + ch), threads=threads)
    dt = time.perf_counter() - t0
    rays = st["closest"] + st["shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"centre crop {cw}x{ch} x {args.cpu_spp} spp of the same frame ({rays} rays, {dt:.1f} s), reference-mode traversal",
            "samples_per_s": st["samples"] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--grid", type=int, nargs=2, default=[205, 204])
    ap.add_argument("--tris", type=int, default=100_000_000, help="triangle count of the c5 soup")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--seed", type=int, default=0x5EED)
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[3840, 2160])
    ap.add_argument("--cpu-spp", type=int, default=3)
    ap.add_argument("--split", choices=["sample", "tile"], default="sample")
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="samples per pixel in one step = one batch through the hot path (the late bounces' fixed cost is paid once per batch)")
    ap.add_argument("--max-bounces", type=int, default=30,
                    help="IndirectLightBounceCount of the reference (config.h: 30); 1 = primary rays and their light connections only")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU renders --spp-per-step samples per step (of every pixel / of its tiles x N); strong: the step's --spp-per-step samples are divided among the GPUs")
    ap.add_argument("--total-spp", type=int, default=0, help="strong scaling: total samples per pixel of the job (sets --steps = total / spp-per-step), e.g. C3's 1024")
    ap.add_argument("--readout", choices=["library", "gloo"], default="library",
                    help="N > 1: library = RCCL reduce inside libmi355pt.so (falls back to gloo if the communicator cannot be created); gloo = host-staged reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # stdout carries ONE JSON line.  Libraries underneath write banners to file descriptor 1 (gloo's "[Gloo] Rank ... connected", RCCL's
    # version block): until the result is printed, descriptor 1 points at stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
    import resource
    import torch
    dist = None
    force = bool(os.environ.get("BENCH_FORCE_DIST"))         # rehearse the N > 1 paths (scene file, communicator, reduce) on one GPU
    if world > 1 or force:
        import torch.distributed as dist
        if force and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            log("rehearsing the distributed paths with world size 1")
        dist.init_process_group("gloo")                     # control plane only; the data plane is RCCL inside the library
    device = local_rank if world > 1 else 0
    if os.environ.get("BENCH_SHARE_DEVICE"):                # rehearsal on a one-GPU box: every rank on device 0 (RCCL refuses a communicator
        device = 0                                          # with one GPU twice, so this run also exercises the gloo fallback of the read-out)

    from pathtracer_amd import DeviceScene, PathTracer, capi, distributed
    lib = capi.load_library()
    ndev = lib.mi355pt_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a gfx950 device; there is no CPU fallback")
    if device >= ndev:                                      # a launcher that shows each rank its own GPU only (HIP_VISIBLE_DEVICES per rank)
        log(f"rank {rank}: LOCAL_RANK {local_rank} but {ndev} visible device(s); using device {device % ndev}")
        device %= ndev
    if args.scaling == "strong":
        if args.total_spp:
            args.steps = max(1, args.total_spp // args.spp_per_step)
        if args.split == "sample" and args.spp_per_step % world:
            raise SystemExit("--scaling strong with the sample split needs --spp-per-step divisible by the number of GPUs")

    t0 = time.perf_counter()
    # building a 100 M-triangle scene (C5), its counting-build step and its CPU leg take minutes in silence: a line on stderr
    # every 30 s tells a watchdog (gpurun's 180 s rule) that the run is alive.  A host thread that sleeps does not touch the GPU.
    import threading
    building = threading.Event()
    phase = ["building the scene"]

    def heartbeat():
        while not building.wait(30.0):
            log(f"[bench] rank {rank}: {phase[0]}, {time.perf_counter() - t0:.0f} s")
    threading.Thread(target=heartbeat, daemon=True).start()

    def peak_rss_gb():
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0

    # ---- the scene: ONE host build.  One rank: built and uploaded in place.  N ranks: rank 0 flattens it into a file that every
    # rank maps (the file's pages are shared; a loading rank never holds a private copy of the scene).
    scene, workload, host = None, None, {}
    if dist is None:
        scene, workload = build_scene(args)
        dscene = DeviceScene(scene, device)
        tris = int(dscene.desc.meshes[0].num_faces) if dscene.desc.num_meshes else 0
        host = {"host_build_s": time.perf_counter() - t0, "host_peak_rss_gb": peak_rss_gb(), "builds": 1}
    else:
        sdir = os.environ.get("MI355PT_SCENE_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
        spath = os.path.join(sdir, f"mi355pt_bench_{os.environ.get('MASTER_PORT', '0')}.scene")
        tris = 0
        if rank == 0:
            scene, workload = build_scene(args)
            for n in scene.rootNode.children:
                if n.obj is not None and getattr(n.obj, "meshes", None):
                    tris = max(tris, int(n.obj.meshes[0].faces.shape[0]))
            DeviceScene.save(scene, spath)
            host = {"host_build_s": time.perf_counter() - t0, "host_peak_rss_gb": peak_rss_gb(), "builds": 1,
                    "scene_file_gb": os.path.getsize(spath) / 1e9}
            scene = None                                       # the Python arrays are not needed any more (N > 1 runs no CPU leg)
        dist.barrier()
        phase[0] = "loading the scene file"
        tl = time.perf_counter()
        dscene = DeviceScene.load(spath, device)
        load_s = time.perf_counter() - tl
        v = torch.tensor([load_s, peak_rss_gb() if rank else 0.0], dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        host.update({"scene_load_s_max": float(v[0]), "loader_peak_rss_gb_max": float(v[1])})
        dist.barrier()
        if rank == 0:
            os.remove(spath)
    strong = args.scaling == "strong"
    if args.split == "tile":
        ts, toff, tstride = distributed.rank_tile_split(rank, world, 64)
        pt = PathTracer(dscene, device=device, seed=args.seed, max_bounces=args.max_bounces, tile_size=ts, tile_offset=toff, tile_stride=tstride)
        # a rank owns 1/N of the pixels: weak = N samples of each keep the per-GPU work fixed; strong = the step's samples of its tiles
        spp_step = args.spp_per_step if strong else world * args.spp_per_step
    else:
        off, stride = distributed.rank_sample_split(rank, world)
        pt = PathTracer(dscene, device=device, seed=args.seed, max_bounces=args.max_bounces, sample_offset=off, sample_stride=stride)
        spp_step = args.spp_per_step // world if strong else args.spp_per_step
    pt.Init(args.width, args.height)
    phase[0] = "running"
    if rank == 0:
        log(f"scene + BVH + upload: {time.perf_counter() - t0:.1f} s  {host}")

    # ---- the read-out of N > 1: a communicator inside the library (RCCL); every rank must have one, or all take the gloo path
    comm, readout, comm_hung = None, None, False
    if dist is not None:
        readout = "gloo"
        if args.readout == "library":
            uid = (C.c_uint8 * capi.COMM_ID_BYTES)()
            ok = 1
            if rank == 0 and lib.mi355pt_comm_unique_id(uid) != 0:
                log("mi355pt_comm_unique_id failed:", lib.mi355pt_last_error().decode()); ok = 0
            t = torch.tensor(list(bytes(uid)) + [ok], dtype=torch.int32)
            dist.broadcast(t, 0)
            ok = int(t[-1])
            if ok:
                uid = (C.c_uint8 * capi.COMM_ID_BYTES)(*[int(x) for x in t[:-1]])
                # the rendezvous of N processes blocks until all have arrived: on its own thread with a limit, so that a rank that
                # never arrives ends in the fallback (and this process in os._exit at the end) instead of in the driver's timeout
                h, res = C.c_void_p(), {}

                def create():
                    rc = lib.mi355pt_comm_create(uid, rank, world, device, C.byref(h))
                    res["rc"], res["msg"] = rc, (lib.mi355pt_last_error().decode() if rc else "")
                th = threading.Thread(target=create, daemon=True)
                th.start(); th.join(float(os.environ.get("BENCH_COMM_TIMEOUT", "120")))
                if th.is_alive():
                    log(f"rank {rank}: mi355pt_comm_create has not returned; giving it up"); ok = 0; comm_hung = True
                elif res["rc"] != 0:
                    log(f"rank {rank}: mi355pt_comm_create failed:", res["msg"]); ok = 0
                else:
                    comm = h
            t = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t[0]):
                readout = "rccl-in-library"
            elif rank == 0:
                log("falling back to the host-staged gloo reduce")

    def read_out():
        """(mean, count, device ms) of the combined image on rank 0; (None, None, ms) elsewhere."""
        split = capi.SPLIT_TILE if args.split == "tile" else capi.SPLIT_SAMPLE
        if readout == "rccl-in-library":
            mean = np.zeros((pt.size, 3), dtype=np.float32) if rank == 0 else None
            cnt = np.zeros(pt.size, dtype=np.uint32) if rank == 0 else None
            ms = C.c_double(0)
            capi.check(lib, lib.mi355pt_render_reduce(pt.h, comm, 0, split, mean.ctypes.data if rank == 0 else None,
                                                      cnt.ctypes.data if rank == 0 else None, None, C.byref(ms)), "mi355pt_render_reduce")
            return mean, cnt, ms.value
        m, c = pt.read_accum()
        res = distributed.combine_running_means(torch.from_numpy(m), torch.from_numpy(c.astype(np.int32)), dist, 0)
        if res is None:
            return None, None, 0.0
        return res[0].numpy(), res[1].numpy().astype(np.uint32), 0.0

    def barrier():
        if dist is not None:
            dist.barrier()
        pt.sync()
        if torch.cuda.is_initialized():      # (it is not: this process runs ONE HIP runtime, the library's; pt.sync() is the device sync)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        pt.step(spp_step)
    pt.sync()
    if readout == "rccl-in-library":
        # The read-out is part of the job: warm it up too (RCCL channels, buffers) -- on its own thread with a limit, because a
        # collective that one rank never enters blocks the others for good.  Every rank must have come back, or all of them
        # take the host-staged path (which reads the accumulators on the renderer's own stream, not the communicator's).
        res = {}

        def probe():
            try:
                read_out(); res["ok"] = 1
            except Exception as e:      # noqa: BLE001 -- any failure of the probe means "fall back"
                res["ok"] = 0; log(f"rank {rank}: read-out through the library failed: {e}")
        th = threading.Thread(target=probe, daemon=True)
        th.start(); th.join(float(os.environ.get("BENCH_COMM_TIMEOUT", "120")))
        if th.is_alive():
            log(f"rank {rank}: the RCCL reduce has not returned; giving it up"); comm_hung = True
        t = torch.tensor([0 if th.is_alive() else res.get("ok", 0)], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not int(t[0]):
            readout = "gloo"
            if rank == 0:
                log("falling back to the host-staged gloo reduce")
    elif dist is not None and args.warmup > 0:
        read_out()
    pt.reset_stats()
    pt.set_profiling(events=1, count_work=False)        # HIP events around every launch of the dominant kernel, on its stream
    npix = args.width * args.height
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pt.step(spp_step)
    pt.sync()
    readout_ms, combined = None, None
    if dist is not None:            # read-out: ONE reduce of 4 floats per pixel, straight from the accumulators in HBM
        tr = time.perf_counter()
        cmean, ccnt, dev_ms = read_out()
        readout_ms = {"wall": (time.perf_counter() - tr) * 1e3, "device": dev_ms}
        combined = (cmean, ccnt)
    barrier()
    dt = time.perf_counter() - t0
    st = pt.stats()
    rays = st["closest_rays"] + st["shadow_rays"]
    if dist is not None:
        if rank == 0:
            # the image the job produced: every pixel must hold every sample of every rank, and finite tone-mapped values
            want = (args.warmup + args.steps) * (spp_step * world if args.split == "sample" else spp_step)
            cmean, ccnt = combined
            if not (ccnt == want).all():
                raise SystemExit(f"combined sample counts are wrong: expected {want} everywhere, got min {int(ccnt.min())} max {int(ccnt.max())}")
            if not (np.isfinite(cmean).all() and cmean.min() >= 0.0 and cmean.max() <= 1.0 and cmean.max() > 0.0):
                raise SystemExit("the combined image is not a finite tone-mapped image in [0, 1]")
        v = torch.tensor([dt, float(rays), float(st["samples"])], dtype=torch.float64)
        mx = v.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = v.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, rays_total, samples_total = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        rays_total, samples_total = float(rays), float(st["samples"])

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/s", "value": rays_total / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "triangles": tris,
                       "resolution": [args.width, args.height], "spp_per_step": args.spp_per_step, "max_bounces": args.max_bounces,
                       "spp_per_step_per_gpu": spp_step,
                       "split": ("sample index g, g+N, ... per GPU" if args.split == "sample" else "64x64 image tiles k % N == g")
                                + "; one RCCL reduce at the end",
                       # the distance cull of DESIGN.md section 3 item 8 (MI355PT_NO_CULL=1: the reference's exhaustive walk)
                       "cull": "none" if os.environ.get("MI355PT_NO_CULL") else ("plain" if os.environ.get("MI355PT_CULL_PLAIN") else "margin")},
            "samples_per_pixel_per_s": samples_total / npix / dt,
            "ms_per_sample": dt / max(samples_total / npix, 1e-9) * 1e3,
            "rays_per_sample": rays_total / max(samples_total, 1.0),
            # LightVisTest calls that entered a visibility kernel; the rest of shadow_rays were decided in k_shade (still one call each)
            "shadow_rays_traversed_share": st["shadow_rays_traversed"] / max(st["shadow_rays"], 1),
            "host": host,
        }
        if dist is not None:
            out["readout"] = readout
            out["readout_ms"] = readout_ms
            out["combined_counts_checked"] = True
    # ---- roofline of the workload's dominant kernel: algorithmic bytes / measured launch time (HIP events on its stream)
    if rank == 0 and not args.no_roofline:
        ext_ms, launches = st["extend_ms"], st["extend_launches"]
        closest = st["closest_rays"]
        # per-stage times of ONE extra, untimed step with events around every stage (the visibility stage overlaps the
        # next bounce's closest-hit trace on a second stream, so the stages do not add up to the step)
        pt.set_profiling(events=2, count_work=False)
        pt.reset_stats()
        pt.step(spp_step)
        pt.sync()
        ks = pt.stats()
        out["kernel_ms"] = {"extend": ks["extend_ms"], "shade": ks["shade_ms"], "visibility+connect": ks["connect_ms"], "other": ks["other_ms"], "steps": 1}
        # the same kernel with the chip to itself: one more untimed step with every kernel back to back on one stream.
        # In the timed region k_trace<closest>(b + 1) shares the CUs with k_trace<any>(b), so its launches last longer
        # there although the step is shorter; `frac` is the timed region's, `frac_exclusive` this step's.
        pt.set_overlap(1 << 30)
        pt.set_profiling(events=2, count_work=False)
        pt.reset_stats()
        pt.step(spp_step)
        pt.sync()
        xs = pt.stats()
        pt.set_overlap(0)
        out["kernel_ms_exclusive"] = {"extend": xs["extend_ms"], "shade": xs["shade_ms"], "visibility+connect": xs["connect_ms"], "other": xs["other_ms"], "steps": 1}
        pt.set_profiling(events=False, count_work=True)      # counting build, untimed
        pt.reset_stats()
        pt.step(spp_step)
        pt.sync()
        cs = pt.stats()
        n_node = cs["bvh_node_visits"] / max(cs["closest_rays"], 1)
        n_tri = cs["bvh_tri_tests"] / max(cs["closest_rays"], 1)
        bytes_per_ray = NODE_BYTES * n_node + TRI_BYTES * n_tri + QUEUE_BYTES
        achieved = closest * bytes_per_ray / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        # HBM traffic per launch: NOT measured by this run.  It comes from the builder's PMC passes of the same kernels
        # (tools/gpu_measure.sh -> tools/pmc_traffic.py -> profiles/pmc_traffic.json; FETCH_SIZE and WRITE_SIZE in separate runs,
        # FETCH_SIZE scaled by the factor calibrated for scattered 64-B records, tools/calib_records.hip), and is reported only
        # while the kernel sources are the ones those passes ran.
        traffic, traffic_source, tj = None, None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and args.workload == "c4" and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pmc_traffic
            with open(tp) as f:
                tj = json.load(f)
            if tj.get("kernel_source_hash") == pmc_traffic.source_hash() and tj.get("spp_per_step") == args.spp_per_step:
                traffic_source = f"profiles/pmc_traffic.json (builder PMC pass {tj.get('tag')}, not this run; FETCH_SIZE x {tj.get('fetch_size_factor_scattered_records'):.2f} per tools/calib_records.hip)"
            else:
                traffic_source = "none: profiles/pmc_traffic.json was taken from other kernel sources or another batch size"
                tj = None
        trace = {"bound": "hbm", "kernel": "k_trace<closest> (extend)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBS, "traffic": tj.get("k_trace_closest_bytes_per_launch") if tj else None, "traffic_source": traffic_source,
                 "bytes_per_ray": bytes_per_ray, "n_node_per_ray": n_node, "n_tri_per_ray": n_tri,
                 "launches": int(launches), "avg_launch_ms": ext_ms / max(launches, 1),
                 "frac_exclusive": (xs["closest_rays"] * bytes_per_ray / (xs["extend_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if xs["extend_ms"] > 0 else None,
                 "avg_launch_ms_exclusive": xs["extend_ms"] / max(xs["extend_launches"], 1),
                 # what the memory system sustains for this access shape: random 64-B records of a 1 GiB table, measured by
                 # tools/calib_records.hip on the same GPU model (3.5 TB/s; the 8 TB/s peak is a streaming figure)
                 "scattered_record_ceiling_gbs": 3540.0,
                 "avg_bytes_per_launch": closest * bytes_per_ray / max(launches, 1),
                 "top_records_in_lds_per_ray": cs["debug"][7] / max(cs["closest_rays"], 1),
                 "shadow_n_node_per_ray": cs["shadow_node_visits"] / max(cs["shadow_rays"], 1),
                 "shadow_n_tri_per_ray": cs["shadow_tri_tests"] / max(cs["shadow_rays"], 1)}
        # k_shade: SURVEY 8(d) per vertex: 96 B attributes + 128 B path state + 4 texels x 3 B per bilinear lookup
        verts, taps = xs["shaded_vertices"], xs["texture_taps"]
        shade_bytes = verts * (96.0 + 128.0) + taps * 12.0
        shade_ach = shade_bytes / (xs["shade_ms"] * 1e-3) / 1e9 if xs["shade_ms"] > 0 else 0.0
        shade = {"bound": "hbm", "kernel": "k_shade", "achieved": shade_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": shade_ach / HBM_PEAK_GBS,
                 "traffic": tj.get("k_shade_bytes_per_launch") if tj else None, "traffic_source": traffic_source,
                 "bytes_per_vertex": shade_bytes / max(verts, 1), "vertices": int(verts), "texture_lookups_per_vertex": taps / max(verts, 1),
                 "ms_exclusive": xs["shade_ms"],
                 "note": "instruction-bound: about 4 000 VALU wave-instructions per vertex (DESIGN.md section 3); the HBM fraction is reported because the contract asks for it"}
        # visibility stage (k_trace<any> + the MIS combination): the same record formula with the visibility traversal's own node and
        # triangle counts, over the LightVisTest calls that reached the kernel, in the exclusive step's time for the stage
        vis_rays = cs["shadow_rays_traversed"]
        vn, vt = cs["shadow_node_visits"] / max(vis_rays, 1), cs["shadow_tri_tests"] / max(vis_rays, 1)
        vis_bytes_per_ray = NODE_BYTES * vn + TRI_BYTES * vt + QUEUE_BYTES
        vis_ach = xs["shadow_rays_traversed"] * vis_bytes_per_ray / (xs["connect_ms"] * 1e-3) / 1e9 if xs["connect_ms"] > 0 else 0.0
        out["roofline_visibility"] = {"bound": "hbm", "kernel": "k_trace<any> + connect (visibility stage)", "achieved": vis_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": vis_ach / HBM_PEAK_GBS, "traffic": tj.get("k_trace_any_bytes_per_launch") if tj else None, "traffic_source": traffic_source,
                                      "bytes_per_ray": vis_bytes_per_ray, "n_node_per_ray": vn, "n_tri_per_ray": vt,
                                      "rays_traversed": int(xs["shadow_rays_traversed"]), "light_vis_tests": int(xs["shadow_rays"]), "ms_exclusive": xs["connect_ms"]}
        # the workload's dominant kernel goes to `roofline`, the other is kept beside it
        if xs["shade_ms"] > xs["extend_ms"]:
            out["roofline"], out["roofline_extend"] = shade, trace
        else:
            out["roofline"], out["roofline_shade"] = trace, shade
    if rank == 0 and dist is None and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, pt, args)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    building.set()
    pt.close_render()
    if comm is not None and not comm_hung:
        lib.mi355pt_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if comm_hung:                   # a thread of this process is still inside the rendezvous: leave without waiting for it
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
