#!/usr/bin/env python3
"""Benchmark of the ray-tracing hot path on MI355X.

Default workload (BASELINE.json configs[1]): a ~1.31 M-triangle fractal-displaced icosphere,
1920x1080 primary rays (one per pixel), closest-hit traversal on one GPU.  A step is one pass
of the closest-hit kernel over the whole frame's rays, which are resident in HBM before the
timed region starts.  With N > 1 ranks (torch.distributed, one process per GPU) every rank
traces its own full frame of jittered primary rays (weak scaling, no data-path collective) and
the reported value is the total over ranks divided by the slowest rank's time.

Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def bench_frame(torch, dist, dev, rank, world, all_reduce, frames=3):
    """ms/frame at fixed spp on configs[2]: the interior scene (~250 k triangles, diffuse + glossy, one
    area and one point light), 1920x1080, 8x8 samples per pixel, path length 9.  The frame is cut into
    interleaved row bands over the ranks (strong scaling) and gathered to rank 0 inside the timed region."""
    from raytracer_amd import abi, api, distributed, scenes

    W, H, G, M = 1920, 1080, 8, 8
    sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)
    bpr = 1 if world == 1 else 4
    # pre-flight on every rank before any collective of this section: a rank that cannot render must
    # not leave the others waiting in the gather
    ok = 1.0
    try:
        probe = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        api.Renderer().render_device(scene, cam, abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1,
                                                                  tile=(0, 0, 64, 8)), probe, torch.cuda.current_stream())
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        ok = 0.0
        sys.stderr.write("frame pre-flight failed on rank %d: %s\n" % (rank, e))
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.float64, device=dev)
        all_reduce(flag, dist.ReduceOp.MIN)
        ok = float(flag.item())
    if ok == 0.0:
        scene.close()
        return {"error": "frame pre-flight failed on some rank (see stderr)"}
    times, rays = [], 0
    for k in range(frames + 1):  # the first frame is the warm-up
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        frame, tot = distributed.render_frame_distributed(scene, cam, st, rank, world, dev, bands_per_rank=bpr)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        r = float(tot["primary_rays"] + tot["extension_rays"] + tot["shadow_rays"])
        if world > 1:
            t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
            r_sum = torch.tensor([r], dtype=torch.float64, device=dev)
            all_reduce(t_max, dist.ReduceOp.MAX)
            all_reduce(r_sum, dist.ReduceOp.SUM)
            dt, r = float(t_max.item()), float(r_sum.item())
        if k > 0:
            times.append(dt)
            rays = r
    scene.close()
    ms = float(np.mean(times)) * 1e3
    return {"workload": "configs[2]: interior scene, %d triangles, %dx%d, %dx%d spp, path length 9, row bands over %d GPU(s) + gather"
                        % (sc.n_triangles(), W, H, G, M, world),
            "ms_per_frame": ms, "mrays_per_s": rays / ms / 1e3, "rays_per_frame": rays, "frames": frames, "scaling": "strong"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--subdiv", type=int, default=8, help="icosphere subdivisions (8 -> 1 310 720 triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame", action="store_true", help="skip the path-traced frame (configs[2]) reported beside the headline value")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracer_amd import abi, api, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the ray-tracing path)")
    # RTGPU_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks share
    # devices, collectives run on host copies); the measured configuration is always nccl = RCCL
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

    def all_reduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    W, H = args.width, args.height
    sc, cam_m, yfov = scenes.blob_scene(args.subdiv)
    ntri = sc.n_triangles()
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    info = scene.info()

    n = W * H
    # rank 0 traces pixel-centre rays, other ranks jittered ones (different sample of the same frame)
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, seed=rank,
                           jitter_mode=abi.JITTER_CENTER if rank == 0 else abi.JITTER_RANDOM)
    d_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_hits = torch.empty((n, 5), dtype=torch.int32, device=dev)  # rtgpu_hit = 5 dwords
    stream = torch.cuda.current_stream()
    api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
    torch.cuda.synchronize()

    # per-ray work counters (boxes / triangles a ray tests when it is traversed on its own: the
    # figure the algorithmic byte count is defined on; the timed kernel's cooperative schedule
    # tests some more)
    d_cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    scene.trace_counts_device(d_o, d_d, n, d_cnt, stream)
    torch.cuda.synchronize()
    cnt = d_cnt.to(torch.float64).mean(dim=0).tolist()
    n_nodes, n_tris = cnt[0], cnt[1]

    for _ in range(args.warmup):
        scene.trace_rays_device(d_o, d_d, n, d_hits, stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(args.steps):
        scene.trace_rays_device(d_o, d_d, n, d_hits, stream)
        ev[k + 1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(tt, dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kernel_ms_avg = float(np.mean(kernel_ms))
    hits = d_hits.cpu().numpy().view(np.uint32)
    hit_frac = float((hits[:, 2] != 0xFFFFFFFF).mean())

    total_rays = float(n) * args.steps * world
    mrays = total_rays / elapsed / 1e6
    # algorithmic bytes per ray: ray in 32 B, hit out 16 B, 32 B per tree-node box record and
    # 36 B per triangle tested
    bytes_per_ray = 32.0 + 16.0 + 32.0 * n_nodes + 36.0 * n_tris
    achieved = bytes_per_ray * n / (kernel_ms_avg * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(HERE, "profiles", "traffic_trace_closest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # VALU wave-instructions per launch from the committed PMC pass (SQ_INSTS_VALU): each occupies a SIMD's issue port
    # for 4 cycles (wave64 on a 16-lane SIMD); 256 CUs x 4 SIMDs at 2.4 GHz
    valu_issue = None
    ppath = os.path.join(HERE, "profiles", "r01f_pmc_summary.json")
    if os.path.exists(ppath) and (args.width, args.height, args.subdiv) == (1920, 1080, 8):
        try:
            valu = json.load(open(ppath))["per_launch"]["SQ_INSTS_VALU"]["mean"]
            valu_issue = valu * 4.0 / 1024.0 / (kernel_ms_avg * 1e-3 * 2.4e9)
        except Exception:
            valu_issue = None

    out = {
        "metric": "Mrays/s (primary+secondary) and ms/frame at fixed spp, 1/2/4/8 MI355X",
        "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: fractal-displaced icosphere, %d triangles, %dx%d primary rays (1 per pixel), "
                               "closest hit on one MI355X per rank" % (ntri, W, H),
                   "rays_per_step_per_gpu": n, "hit_fraction": hit_frac, "tree_nodes": int(info.n_nodes),
                   "tree_max_depth": int(info.max_depth), "scene_build_ms": float(info.build_ms),
                   "scene_hbm_bytes": int(info.device_bytes)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "trace_closest_steal_kernel<false>", "kernel_ms_avg": kernel_ms_avg,
                     "bytes_per_ray": bytes_per_ray, "nodes_per_ray": n_nodes, "tris_per_ray": n_tris,
                     # the tree and triangles (scene_hbm_bytes) fit the 256 MiB Infinity Cache, so most of the
                     # algorithmic bytes never reach HBM: the measured HBM rate is reported beside the algorithmic one
                     "hbm_measured_GBps": (traffic / (kernel_ms_avg * 1e-3) / 1e9) if traffic else None,
                     # what actually binds this kernel: the share of the VALU issue slots (at 2.4 GHz) its instructions take
                     "valu_issue_frac": valu_issue},
    }

    if not args.no_frame:
        try:
            out["frame"] = bench_frame(torch, dist, dev, rank, world, all_reduce)
        except Exception as e:  # noqa: BLE001  (the headline line above must still be printed)
            out["frame"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        threads = oracle.default_threads()
        t0 = time.perf_counter()
        osc = oracle.OracleScene(sc, threads=threads)
        build_s = time.perf_counter() - t0
        o_h = d_o.cpu().numpy()
        d_h = d_d.cpu().numpy()
        reps, spent = 0, 0.0
        while spent < 12.0 and reps < 2000:
            t0 = time.perf_counter()
            ref = osc.trace_closest(o_h, d_h, threads=threads)
            spent += time.perf_counter() - t0
            reps += 1
        same = (ref["inst"] == hits[:, 2]) & (ref["prim"] == hits[:, 1])
        # rays whose ids differ must be exact ties in t (two triangles at the bit-identical distance)
        bad = np.nonzero(~same)[0][:64]
        ties = 0
        if bad.size:
            br, tied = osc.brute_closest(o_h[bad], d_h[bad])
            ties = int(((tied >= 2) & (br["prim"] == hits[bad, 1])).sum())
        out["cpu_baseline"] = {"value": n * reps / spent / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": "the same %d primary rays, %d passes of the CPU oracle (%.1f s), tree build %.1f s excluded"
                                         % (n, reps, spent, build_s),
                               "id_mismatches_vs_gpu": int((~same).sum()), "of_which_exact_t_ties": ties}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
