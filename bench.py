#!/usr/bin/env python3
"""Benchmark of the ray-tracing hot path on MI355X.

Headline (every N): one path-traced frame of the ~1.31 M-triangle blob-in-a-room scene -- 3840x2160, 8x8 = 64
samples per pixel, path length 9, diffuse + glossy, one area and one point light -- i.e. incoherent extension and
shadow rays on a ~1 M-triangle scene.  A step is one frame.  The frame is cut into 32x32 squares shared out over the
ranks (one process per GPU, scene replicated, strong scaling); every rank renders its squares into a compact buffer
and the buffers are gathered on rank 0 by the library's RCCL send / receive pairs inside the timed region.
value = rays traced (primary + extension + shadow, every query counted once) by all ranks / wall time.

At N = 1 the line also carries: the roofline of the dominant kernel (trace_stage_kernel: the resource that binds it --
VALU instruction issue -- from committed counters of this kernel over this run's live kernel time, beside the algorithmic
bytes of a counting render against the L2 and HBM peaks), the CPU baseline (the oracle on a tile of the same frame) and,
under "extras", the other configurations of BASELINE.json with a roofline each: configs[0] the scenegen scene's primary +
shadow rays beside the oracle on the same rays, configs[1] coherent primary rays and bounce rays on the blob, configs[2]
the interior frame, configs[3] the same interior at 3840x2160 and 256 spp on one GPU (the single-GPU reference of the
2 / 4 / 8-GPU runs), configs[4] the 49.8 M-triangle cluster.

Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the 8 XCDs (MI355X_MICROARCH.md, "L2 (per XCD)")
MALL_BYTES = 256 << 20  # Infinity Cache: scenes below this size are served on-die, HBM is not what binds them
# VALU issue: a SIMD issues one wave64 vector instruction per 4 cycles; 256 CUs x 4 SIMDs x 2.4 GHz / 4
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0
TILE = 32


def ray_bytes(rays, boxes, tris, out_bytes):
    """Algorithmic bytes of a class of rays: 32 B ray in, `out_bytes` out, 32 B per child box and 36 B per triangle tested
    by the ray traversed on its own."""
    return (32.0 + out_bytes) * rays + 32.0 * boxes + 36.0 * tris


def roofline(alg_bytes, kernel_ms, scene_bytes, kernel, extra=None):
    """Roofline record of a traversal kernel from what this run measures: the algorithmic bytes (every child box and
    triangle a ray tests when traversed on its own, 32 / 36 B each) over the kernel's time, against the HBM peak.  For scenes
    that fit the caches that rate exceeds the HBM peak -- the tree is served by the compute units' L1s (79-84 % of the
    look-ups), L2 and the Infinity Cache -- so it is a statement about the work done, not about a resource in use, and the
    record says so (`algorithmic_GBps`, `frac_of_hbm_peak`, `frac_of_l2_peak`).  What actually binds these kernels is
    vector-instruction issue; `attach_counters` replaces bound / achieved / peak / frac by that resource when committed
    counters of this kernel on this workload exist, and adds the traffic the counters saw at each level."""
    alg = alg_bytes / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
           "traffic": None, "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
           "algorithmic_GBps": alg, "frac_of_hbm_peak": alg / HBM_PEAK_GBS, "frac_of_l2_peak": alg / L2_PEAK_GBS,
           "scene_bytes": scene_bytes, "scene_fits_infinity_cache": bool(scene_bytes <= MALL_BYTES),
           "note": "no committed counters for this kernel and workload: algorithmic bytes against the HBM peak (a fraction above 1 "
                   "means the scene is served from the caches, not that HBM delivered it)"}
    if extra:
        out.update(extra)
    return out


def attach_counters(rl, c, src, per):
    """Names the binding resource from committed PMC figures of the same kernel on the same workload.  The instruction count
    of a frame (or launch) belongs to the kernel build and the workload, not to the run (the profile's commit is in the
    summary file); the time it is divided by is this run's."""
    label = "committed profile " + src + " (counters not measured in this run; kernel time measured in this run)"
    rl["traffic"] = c.get("hbm_bytes_per_" + per)
    rl["counters"] = dict(c, source=label)
    seconds = rl["kernel_ms"] * 1e-3
    if c.get("SQ_INSTS_VALU"):
        ginst = c["SQ_INSTS_VALU"] / seconds / 1e9
        rl.update({"bound": "valu_issue", "achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instructions/s",
                   "frac": ginst / VALU_PEAK_GINST,
                   "note": "SQ_INSTS_VALU of one %s (%s) / this run's kernel time, against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per "
                           "instruction; frac in the profiled run itself: %.3f" % (per, label, c.get("valu_issue_frac", float("nan")))})
        rl["binding"] = {"resource": "valu_issue", "frac": rl["frac"], "frac_in_profiled_run": c.get("valu_issue_frac"),
                         "lanes_per_valu_instruction": c.get("lanes_per_valu_instruction"),
                         "waiting_share_of_wave_cycles": c.get("wait_any_share_of_wave_cycles"), "source": label}
    levels = {}
    if c.get("l1_access_bytes"):
        levels["l1"] = {"bytes": c["l1_access_bytes"], "GBps": c["l1_access_bytes"] / seconds / 1e9, "hit_rate": c.get("l1_hit_rate"),
                        "note": "TCP_TOTAL_CACHE_ACCESSES x 64 B"}
    if c.get("l2_to_cu_bytes"):
        g = c["l2_to_cu_bytes"] / seconds / 1e9
        levels["l2_to_cu"] = {"bytes": c["l2_to_cu_bytes"], "GBps": g, "frac_of_l2_peak": g / L2_PEAK_GBS, "l2_hit_rate": c.get("l2_hit_rate"),
                              "note": "TCP_TCC_READ_REQ x 64 B: what the L1s asked L2 for"}
    if c.get("hbm_bytes_per_" + per):
        g = c["hbm_bytes_per_" + per] / seconds / 1e9
        levels["hbm"] = {"bytes": c["hbm_bytes_per_" + per], "GBps": g, "frac_of_peak": g / HBM_PEAK_GBS,
                         "note": "2 x FETCH_SIZE + WRITE_SIZE; an upper bound for 16-byte gathers"}
    if levels:
        rl["traffic_by_level"] = dict(levels, source=label)


def committed_counters(tag):
    """PMC figures of profiles/<tag>_pmc_summary.json (collected by scripts/gpu_profile.sh in separate rocprofv3 --pmc
    passes of this program; NOT measured in this run)."""
    path = os.path.join(HERE, "profiles", tag + "_pmc_summary.json")
    try:
        summary = json.load(open(path))
        return summary, os.path.relpath(path, HERE) + " taken at commit " + str(summary.get("commit", "?"))
    except Exception:
        return None, None


def frame_rooflines(torch, api, abi, scene, cam, st, dev, info, kernel_name):
    """Counting render + profiling render of one frame -> roofline of the trace stage and the stage split."""
    fb = torch.zeros((st.height, st.width, 3), dtype=torch.float32, device=dev)
    r = api.Renderer()
    work = r.render_work_device(scene, cam, st, fb)
    prof = r.render_profile_device(scene, cam, st, fb)
    del fb
    ext = ray_bytes(work.extension_rays, work.extension_boxes, work.extension_triangles, 16.0)
    shd = ray_bytes(work.shadow_rays, work.shadow_boxes, work.shadow_triangles, 4.0)
    n_ext, n_shd = max(1, work.extension_rays), max(1, work.shadow_rays)
    rl = roofline(ext + shd, prof.trace_ms, int(info.device_bytes), kernel_name, {
        "kernel_ms_avg_per_launch": prof.trace_ms / max(1, prof.trace_launches), "launches_per_frame": prof.trace_launches,
        "extension": {"rays": work.extension_rays, "boxes_per_ray": work.extension_boxes / n_ext, "tris_per_ray": work.extension_triangles / n_ext,
                      "bytes_per_ray": ext / n_ext},
        "shadow": {"rays": work.shadow_rays, "boxes_per_ray": work.shadow_boxes / n_shd, "tris_per_ray": work.shadow_triangles / n_shd,
                   "bytes_per_ray": shd / n_shd},
        "stage_mrays_per_s": (work.extension_rays + work.shadow_rays) / prof.trace_ms / 1e3,
        "measured": "counting render + profiling render (one batch in flight, HIP events around every launch) in this run"})
    stages = {"primary_ms": prof.primary_ms, "shade_ms": prof.shade_ms, "trace_ms": prof.trace_ms, "gather_ms": prof.gather_ms,
              "resolve_ms": prof.resolve_ms, "total_ms": prof.total_ms,
              # shade stage: 64 B in / 48 B out per path vertex shaded (one vertex per path and pass)
              "shade_GBps": (work.extension_rays + work.primary_rays * st.material_aa) * 112.0 / max(prof.shade_ms, 1e-9) / 1e6}
    return rl, stages


def native_comm(torch, dist, api, rank, world, dev, timeout_s=90.0):
    """The library's RCCL communicator for this rank, checked with a small gather before anything is timed.  Made in a
    helper thread so that a rendezvous that never completes cannot hold the benchmark: returns (comm or None, note)."""
    from raytracer_amd import abi, distributed
    ident = torch.zeros(api.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        ident.copy_(torch.tensor(list(api.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(ident, 0)
    ident_bytes = bytes(ident.cpu().numpy().tolist())
    box = {}

    def make():
        try:
            api.set_device(dev.index)
            torch.cuda.set_device(dev)
            comm = api.Comm(ident_bytes, rank, world)
            W, H, T = 67, 45, 8
            st = abi.make_settings(W, H)
            ys, xs, cs = np.arange(H)[:, None, None], np.arange(W)[None, :, None], np.arange(3)[None, None, :]
            pattern = (ys * 1000 + xs * 3 + cs + 1).astype(np.float32)
            local = torch.from_numpy(distributed.compact_from_frame(pattern, st, T, rank, world)).to(dev)
            frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
            s = torch.cuda.Stream(device=dev)
            comm.gather_tiles_device(st, T, local, frame, s)
            s.synchronize()
            if rank == 0 and not np.array_equal(frame.cpu().numpy(), pattern):
                raise RuntimeError("the gathered frame differs from the pattern")
            box["comm"] = comm
        except Exception as e:  # noqa: BLE001
            box["error"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=make, daemon=True)
    th.start()
    th.join(timeout_s)
    ok = 1.0 if ("comm" in box and not th.is_alive()) else 0.0
    flag = torch.tensor([ok], dtype=torch.float64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if float(flag.item()) == 1.0:
        box["comm"].set_timeout_ms(int(timeout_s * 1e3))   # a peer lost later ends the run with an error instead of a hang
        return box["comm"], "rtgpu_gather_tiles_device: RCCL ncclSend / ncclRecv pairs to rank 0 + scatter kernel"
    why = box.get("error", "timed out after %.0f s" % timeout_s if th.is_alive() else "failed on another rank")
    sys.stderr.write("rank %d: native communicator unusable (%s); gathering with torch.distributed instead\n" % (rank, why))
    return None, "torch.distributed gather (the library's communicator was unusable: %s)" % why


def bench_headline(args, torch, dist, api, abi, scenes, rank, world, dev):
    from raytracer_amd import distributed
    W, H, G, M = args.width, args.height, args.gaa, args.maa
    sc, cam_m, yfov = scenes.blob_room_scene(args.subdiv)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    info = scene.info()
    st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)

    # pre-flight on every rank before any collective: a rank that cannot render must not leave the others waiting
    ok = 1.0
    try:
        probe = torch.zeros(api.tile_buffer_floats(abi.make_settings(W, H, tile=(0, 0, 64, 32)), TILE, 0, 1), dtype=torch.float32, device=dev)
        api.Renderer().render_tiles_device(scene, cam, abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1, tile=(0, 0, 64, 32)),
                                           TILE, 0, 1, probe)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        ok = 0.0
        sys.stderr.write("pre-flight render failed on rank %d: %s\n" % (rank, e))
    comm, gather_note = None, "one rank: scatter kernel only"
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the benchmark's own small collectives live
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) == 0.0:
            raise SystemExit("pre-flight render failed on some rank (see stderr)")
        if args.backend == "nccl":
            comm, gather_note = native_comm(torch, dist, api, rank, world, dev)
        else:
            gather_note = "torch.distributed gather over %s (rehearsal: ranks share a GPU)" % args.backend
    elif ok == 0.0:
        raise SystemExit("pre-flight render failed")
    if world == 1:
        comm = api.Comm(api.comm_unique_id(), 0, 1)
    sharder = distributed.FrameSharder(scene, rank, world, dev, comm=comm, tile_size=TILE)

    def frame():
        return sharder.render(cam, st)

    for _ in range(args.warmup):
        frame()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rays = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, stats = frame()
        rays += stats.primary_rays + stats.extension_rays + stats.shadow_rays
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        rr = torch.tensor([float(rays)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        elapsed, rays = float(tt.item()), float(rr.item())
    checks = {}
    if rank == 0:
        img = out if isinstance(out, np.ndarray) else out.cpu().numpy()
        finite = np.isfinite(img)
        checks = {"frame_mean": float(img[finite].mean()), "non_finite_pixels": int((~finite).any(axis=2).sum())}
    res = {"scene": sc, "cam": cam, "st": st, "handle": scene, "info": info, "elapsed": elapsed, "rays": float(rays),
           "gather": gather_note, "checks": checks, "frame": img if rank == 0 else None,
           # which gather was timed: the library's own ("native"; with one rank its scatter kernel alone) or torch.distributed
           # standing in for it ("fallback": such a line does not measure the library's RCCL exchange)
           "gather_kind": "native" if comm is not None else "fallback"}
    return res


def bench_c1(args, torch, api, abi, scenes, dev):
    """configs[0]: the scenegen.py scene (1 + 100 instances of a 960-triangle sphere mesh plus 16 analytic unit spheres, one
    point light), 640x480, one primary ray per pixel and one shadow ray per hit towards the light (o = p + l * 1e-4,
    tmax = |L - p|, material/NextEventEstimation.cpp:10-16).  The GPU beside the oracle on the very same rays; ids,
    distances and occlusion flags are compared in the run."""
    import oracle
    W, H = 640, 480
    sc, cam_m, yfov = scenes.c1_scene(seed=0)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    info = scene.info()
    stream = torch.cuda.current_stream()
    n = W * H
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
    d_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
    hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
    cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    scene.trace_counts_device(d_o, d_d, n, cnt, stream)
    torch.cuda.synchronize()
    boxes, tris = cnt.to(torch.float64).sum(dim=0).tolist()

    def timed(f, reps=50):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record(stream)
        for k in range(reps):
            f()
            ev[k + 1].record(stream)
        torch.cuda.synchronize()
        return float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))

    ms_closest = timed(lambda: scene.trace_rays_device(d_o, d_d, n, hits, stream))
    got = hits.cpu().numpy().view(abi.HIT_DTYPE).reshape(-1)
    o, d = d_o.cpu().numpy(), d_d.cpu().numpy()
    hit = got["inst"] != abi.INVALID_ID
    light = np.asarray(sc.point_lights[0][0], dtype=np.float32)
    hp = (o[hit] + d[hit] * got["t"][hit][:, None]).astype(np.float32)
    to_l = light[None, :] - hp
    lamp_t = np.linalg.norm(to_l, axis=1).astype(np.float32)
    ldir = (to_l / np.maximum(lamp_t, 1e-20)[:, None]).astype(np.float32)
    so = (hp + ldir * np.float32(0.0001)).astype(np.float32)
    m = len(so)
    s_o, s_d, s_t = torch.from_numpy(so).to(dev), torch.from_numpy(ldir).to(dev), torch.from_numpy(lamp_t).to(dev)
    occ = torch.empty(m, dtype=torch.uint8, device=dev)
    ms_any = timed(lambda: scene.test_visibility_device(s_o, s_d, s_t, m, occ, stream))
    gpu_rate = (n + m) / (ms_closest + ms_any) / 1e3
    # the oracle on the same rays, every host thread the box gives this process
    threads = oracle.default_threads()
    S = oracle.OracleScene(sc, threads=threads)
    ref = S.trace_closest(o, d)
    ref_occ = S.trace_any(so, ldir, lamp_t)
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        S.trace_closest(o, d)
        S.trace_any(so, ldir, lamp_t)
        reps += 1
    cpu_s = (time.perf_counter() - t0) / reps
    same = bool(np.array_equal(ref["inst"], got["inst"]) and np.array_equal(ref["prim"], got["prim"]) and np.array_equal(ref["t"], got["t"])
                and np.array_equal(ref_occ, occ.cpu().numpy()))
    extra = {"kernel_ms_avg_per_launch": ms_closest, "boxes_per_ray": boxes / n, "tris_per_ray": tris / n,
             "measured": "HIP events around each launch in this run",
             "note": "a launch of 307 200 rays is 4 800 waves, fewer than the chip's 7 168 wave slots: its time is the walk of its slowest "
                     "waves, no throughput roof is approached; algorithmic bytes against the HBM peak for the record"}
    out = {"workload": "configs[0]: scenegen scene (%d mesh instances x 960 triangles + 16 analytic spheres), %dx%d, 1 primary ray per pixel + "
                       "1 shadow ray per hit" % (len(sc.instances) - 16, W, H),
           "mrays_per_s": gpu_rate, "primary_ms": ms_closest, "shadow_ms": ms_any, "primary_rays": n, "shadow_rays": m,
           "identical_to_oracle": same,
           "roofline": roofline(ray_bytes(n, boxes, tris, 16.0), ms_closest, int(info.device_bytes), "trace_closest_kernel<true, false> (reference-order top level, one lane per ray)", extra),
           "cpu_baseline": {"value": (n + m) / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                            "sample": "the same %d primary + %d shadow rays, %d repetitions in %.1f s" % (n, m, reps, cpu_s * reps)}}
    scene.close()
    return out


def bench_c2(args, torch, api, abi, scenes, dev):
    """configs[1]: closest hit for 1920x1080 primary rays of the blob alone (coherent), and for diffuse-bounce rays leaving
    its surface (incoherent)."""
    W, H = 1920, 1080
    sc, cam_m, yfov = scenes.blob_scene(args.subdiv)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    info = scene.info()
    stream = torch.cuda.current_stream()
    n = W * H
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
    d_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
    torch.cuda.synchronize()

    def timed(d_o, d_d, reps=20):
        m = d_o.shape[0]
        hits = torch.empty((m, 5), dtype=torch.int32, device=dev)
        cnt = torch.zeros((m, 2), dtype=torch.int32, device=dev)
        scene.trace_counts_device(d_o, d_d, m, cnt, stream)
        torch.cuda.synchronize()
        boxes, tris = cnt.to(torch.float64).sum(dim=0).tolist()
        for _ in range(3):
            scene.trace_rays_device(d_o, d_d, m, hits, stream)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record(stream)
        for k in range(reps):
            scene.trace_rays_device(d_o, d_d, m, hits, stream)
            ev[k + 1].record(stream)
        torch.cuda.synchronize()
        ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))
        return ms, boxes, tris, hits

    out = {}
    ms, boxes, tris, hits = timed(d_o, d_d)
    counters, src = committed_counters(args.profile_tag)
    extra = {"kernel_ms_avg_per_launch": ms, "boxes_per_ray": boxes / n, "tris_per_ray": tris / n, "measured": "HIP events around each launch in this run"}
    out["primary"] = {"workload": "configs[1]: %d triangles, %dx%d primary rays, closest hit" % (sc.n_triangles(), W, H),
                      "mrays_per_s": n / ms / 1e3, "ms_per_launch": ms,
                      "roofline": roofline(ray_bytes(n, boxes, tris, 16.0), ms, int(info.device_bytes), "trace_closest_steal_kernel<false>", extra)}
    if counters and "c2_primary" in counters and args.subdiv == 8:   # (the blob the profile was taken on)
        attach_counters(out["primary"]["roofline"], counters["c2_primary"], src, "launch")
    # bounce rays: from every primary hit point, a direction in the hemisphere of the (radial) outward normal
    h = hits.cpu().numpy().view(np.uint32)
    hit = h[:, 2] != 0xFFFFFFFF
    o, d = d_o.cpu().numpy()[hit], d_d.cpu().numpy()[hit]
    t = h[hit, 0].view(np.float32)
    hp = (o + d * t[:, None]).astype(np.float32)
    rng = np.random.RandomState(1)
    reps = int(np.ceil(n / len(hp)))
    so = np.tile(hp, (reps, 1))[:n]
    nn = so / np.linalg.norm(so, axis=1, keepdims=True)
    so = (so + nn * 1e-3).astype(np.float32)
    sd = rng.normal(size=so.shape)
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    sd[(sd * nn).sum(1) < 0] *= -1
    b_o, b_d = torch.from_numpy(so).to(dev), torch.from_numpy(sd.astype(np.float32)).to(dev)
    ms, boxes, tris, _ = timed(b_o, b_d)
    extra = {"kernel_ms_avg_per_launch": ms, "boxes_per_ray": boxes / n, "tris_per_ray": tris / n, "measured": "HIP events around each launch in this run"}
    out["bounce"] = {"workload": "configs[1] scene, %d diffuse-bounce rays from the primary hit points in pixel order (incoherent)" % n,
                     "mrays_per_s": n / ms / 1e3, "ms_per_launch": ms,
                     "roofline": roofline(ray_bytes(n, boxes, tris, 16.0), ms, int(info.device_bytes), "trace_closest_steal_kernel<false>", extra)}
    if counters and "c2_bounce" in counters and args.subdiv == 8:
        attach_counters(out["bounce"]["roofline"], counters["c2_bounce"], src, "launch")
    # the host-pointer entry point (rays and hits cross PCIe inside the call); the median of 9 calls: copies of a call that
    # follows the release of large device buffers wait for the driver's own DMA work for some tens of milliseconds
    def host_rate(*arrays):
        scene.trace_rays(*arrays)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            scene.trace_rays(*arrays)
            ts.append(time.perf_counter() - t0)
        return n / float(np.median(ts)) / 1e6

    o_h, d_h = d_o.cpu().numpy(), d_d.cpu().numpy()
    out["primary_host_pointers_mrays_per_s"] = host_rate(o_h, d_h)
    # the same call on arrays in page-locked memory (rtgpu_host_alloc): no staging copies, the DMA engines use the caller's arrays
    p_o, p_d, p_hits = api.host_empty(o_h.shape), api.host_empty(d_h.shape), api.host_empty(n, abi.HIT_DTYPE)
    p_o[:] = o_h
    p_d[:] = d_h
    out["primary_page_locked_host_arrays_mrays_per_s"] = host_rate(p_o, p_d, p_hits)
    scene.close()
    return out


def bench_frame(torch, api, abi, scene_data, cam_m, yfov, W, H, G, M, dev, label, kernel_name, frames=3):
    from raytracer_amd import scenes
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(scene_data)
    info = scene.info()
    st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    r = api.Renderer()
    r.render_device(scene, cam, st, fb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(frames):
        s = r.render_device(scene, cam, st, fb)
        rays = s.primary_rays + s.extension_rays + s.shadow_rays
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    del fb
    rl, stages = frame_rooflines(torch, api, abi, scene, cam, st, dev, info, kernel_name)
    out = {"workload": label % scene_data.n_triangles(), "ms_per_frame": ms, "mrays_per_s": rays / ms / 1e3, "rays_per_frame": rays,
           "scene_hbm_bytes": int(info.device_bytes), "scene_build_ms": float(info.build_ms), "roofline": rl, "stages_one_batch_in_flight": stages}
    scene.close()
    return out


def cpu_baseline(res, abi, api, torch, dev):
    """The oracle on a tile of the headline frame (same scene, camera, samples, seed), its host threads on the GPU box's cores;
    the GPU renders the same tile and the two images are compared."""
    import oracle
    threads = oracle.default_threads()
    st0 = res["st"]
    t0 = time.perf_counter()
    osc = oracle.OracleScene(res["scene"], threads=threads)
    build_s = time.perf_counter() - t0
    W, H = st0.width, st0.height

    def tile_settings(tw, th):
        x0, y0 = (W - tw) // 2, (H - th) // 2
        return abi.make_settings(W, H, geometry_aa=st0.geometry_aa, material_aa=st0.material_aa, max_path_length=st0.max_path_length,
                                 seed=st0.seed, tile=(x0, y0, x0 + tw, y0 + th))

    st = tile_settings(48, 27)
    t0 = time.perf_counter()
    _, stats = osc.render(res["cam"], st, threads=threads)
    dt = time.perf_counter() - t0
    rate = (stats.primary_rays + stats.extension_rays + stats.shadow_rays) / dt
    # second tile sized for about 15 s at that rate
    scale = max(1.0, min(28.0, (15.0 / max(dt, 1e-3)) ** 0.5))
    tw, th = min(W, int(48 * scale) // 2 * 2), min(H, int(27 * scale) // 2 * 2)
    st = tile_settings(tw, th)
    t0 = time.perf_counter()
    ref, stats = osc.render(res["cam"], st, threads=threads)
    dt = time.perf_counter() - t0
    rays = stats.primary_rays + stats.extension_rays + stats.shadow_rays
    got = res["frame"][st.y_start:st.y_end, st.x_start:st.x_end]
    want = ref[st.y_start:st.y_end, st.x_start:st.x_end]
    fin = np.isfinite(want).all(axis=2) & np.isfinite(got).all(axis=2)
    rel_rmse = float(np.sqrt(np.mean((got[fin] - want[fin]) ** 2)) / max(want[fin].mean(), 1e-12))
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "a %dx%d tile in the middle of the same frame (same samples per pixel, path length and random numbers): %d rays in %.1f s; "
                      "tree build %.1f s excluded" % (tw, th, rays, dt, build_s),
            "first_estimate_mrays_per_s": rate / 1e6, "gpu_tile_vs_oracle_rel_rmse": rel_rmse}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gaa", type=int, default=8, help="geometry samples per pixel (geometryAAModifier)")
    ap.add_argument("--maa", type=int, default=8, help="material samples per geometry sample (materialAAModifier)")
    ap.add_argument("--subdiv", type=int, default=8, help="icosphere subdivisions of the blob (8 -> 1 310 720 triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configurations reported beside the headline value")
    ap.add_argument("--no-c5", action="store_true", help="skip the 49.8 M-triangle configuration (about 40 s of scene set-up)")
    ap.add_argument("--no-c3", action="store_true", help="skip the 3840x2160 256-spp interior frame (about 5 s of rendering)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-tag", default="r03", help="profiles/<tag>_pmc_summary.json holds the committed counters quoted beside the live numbers")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracer_amd import abi, api, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the ray-tracing path)")
    # RTGPU_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks share devices,
    # collectives run on host copies, the frame is gathered with torch.distributed); the measured configuration is
    # always nccl = RCCL with the library's own gather
    backend = os.environ.get("RTGPU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    api.set_device(local_rank)
    args.backend = backend

    res = bench_headline(args, torch, dist, api, abi, scenes, rank, world, dev)
    st, info = res["st"], res["info"]
    mrays = res["rays"] / res["elapsed"] / 1e6
    out = {
        "metric": "Mrays/s (primary+secondary) and ms/frame at fixed spp, 1/2/4/8 MI355X",
        "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "gather": res["gather_kind"],
        "config": {"workload": "north-star %dx%d, %dx%d = %d spp, path length 9: path-traced blob-in-a-room (%d triangles, diffuse + glossy, "
                               "area + point light); %dx%d squares over %d GPU(s), gathered on rank 0"
                               % (st.width, st.height, st.geometry_aa, st.material_aa, st.geometry_aa * st.material_aa, res["scene"].n_triangles(),
                                  TILE, TILE, world),
                   "resolution": [st.width, st.height], "spp": st.geometry_aa * st.material_aa, "max_path_length": st.max_path_length,
                   "rays_per_frame": res["rays"] / args.steps, "ms_per_frame": res["elapsed"] / args.steps * 1e3,
                   "tree_nodes": int(info.n_nodes), "tree_max_depth": int(info.max_depth), "scene_build_ms": float(info.build_ms),
                   "scene_hbm_bytes": int(info.device_bytes), "gather": res["gather"], **res["checks"]},
    }
    if rank == 0:
        if not args.no_roofline:
            # (with several ranks: the whole frame once more on rank 0's GPU alone -- the kernels' roofline does not depend on
            # how the frame is shared out; the other ranks wait at the final barrier meanwhile)
            rl, stages = frame_rooflines(torch, api, abi, res["handle"], res["cam"], st, dev, info, "trace_stage_kernel<kTraceSingle>")
            if world > 1:
                rl["measured"] += "; whole frame on rank 0's GPU alone"
            counters, src = committed_counters(args.profile_tag)
            default_frame = (args.width, args.height, args.gaa, args.maa, args.subdiv) == (3840, 2160, 8, 8, 8)   # the frame the profile was taken on
            if counters and "headline_trace_stage" in counters and default_frame:
                attach_counters(rl, counters["headline_trace_stage"], src, "frame")
            out["roofline"] = rl
            out["stages_one_batch_in_flight"] = stages
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(res, abi, api, torch, dev)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    res["handle"].close()
    res["frame"] = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        steps = [("configs[0]", lambda: bench_c1(args, torch, api, abi, scenes, dev)),
                 ("configs[1]", lambda: bench_c2(args, torch, api, abi, scenes, dev)),
                 ("configs[2]", lambda: bench_frame(torch, api, abi, *scenes.interior_scene(scenes.C3_DETAIL), 1920, 1080, 8, 8, dev,
                                                    "configs[2]: interior scene, %d triangles, 1920x1080, 8x8 spp, path length 9",
                                                    "trace_stage_kernel<kTraceSingle>")),
                 ("configs[3]", None if args.no_c3 else lambda: bench_frame(torch, api, abi, *scenes.interior_scene(scenes.C3_DETAIL), 3840, 2160, 16, 16, dev,
                                                    "configs[3] on one GPU: interior scene, %d triangles, 3840x2160, 16x16 = 256 spp, path length 9 (the "
                                                    "single-GPU reference of the 2 / 4 / 8-GPU runs)", "trace_stage_kernel<kTraceSingle>", frames=1))]
        if not args.no_c5:
            steps.append(("configs[4]", lambda: bench_frame(torch, api, abi, *scenes.cluster_scene(38, args.subdiv, flatten=True), 1920, 1080, 4, 4, dev,
                                                            "configs[4]: 38 copies of the blob flattened into one mesh, %d triangles, camera inside the cluster, "
                                                            "diffuse, 1920x1080, 4x4 = 16 spp, path length 9", "trace_stage_kernel<kTraceSingle>", frames=2)))
        for name, fn in steps:
            if fn is None:
                continue
            try:
                extras[name] = fn()
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        counters, src = committed_counters(args.profile_tag)
        # committed counters of these very frames (scripts/prof_c5.py, scripts/prof_interior.py render them with the same settings)
        for name, key, ok in (("configs[4]", "c5_trace_stage", args.subdiv == 8), ("configs[2]", "interior_1080p_trace_stage", True),
                              ("configs[3]", "interior_4k_trace_stage", True)):
            e = extras.get(name, {})
            if counters and key in counters and "roofline" in e and ok:
                attach_counters(e["roofline"], counters[key], src, "frame")
        out["extras"] = extras
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        os._exit(0)   # a helper thread stuck in a communicator rendezvous must not keep the process alive


if __name__ == "__main__":
    main()
