#!/usr/bin/env python
"""bench.py — Msamples/s of the sparse-convolution GPIS hot path on scene S.

A "step" renders one full frame of the configuration BASELINE.json quotes the metric on
(C1: 1920x1080, 64 spp, 3D isotropic sampling, impulse_density=32, renewal, single realization):
primary sampleDistance, shading, one shadow transmittance per hit, per-pixel accumulation.
Inputs (the medium's constants, the workspace) are resident in HBM before the timed region.

Multi-GPU (one process per GPU, torch.distributed / RCCL): rank 0 broadcasts the parameter
block; rank r renders sample indices [r*spp, (r+1)*spp) of every pixel (per-GPU work is fixed:
weak scaling; every sample's randomness is a function of (pixel, spp index) only, so the sharded
image equals the single-GPU image at spp*N); the per-rank radiance sums are reduced to rank 0
inside the timed region (the one exchange step of the path).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md "Chip-level parameters")
B_SEG = 224                    # bytes per segment: 128 in + 96 out


def algorithmic_bytes_per_eval(params):
    """27*rho*16*L bytes (3D) or 3*rho*8*L bytes (1D); L=2 for multi-resolution."""
    rho = int(params["impulse_density"])
    L = 2 if (params["nonstationary"] and params["multi_resolution_grid"]) else 1
    return (3 * rho * 8 * L) if params["sampling_1d"] else (27 * rho * 16 * L)


def algorithmic_ops_per_eval(params):
    """SURVEY.md §8d, secondary figure for the VALU roof: 27*[H + (2+4*rho)*P + rho*D] + 0.155*27*rho*K
    with H=24 (hash), P=14 (PCG draw), D=10 (distance test), K=40 (kernel); 3 cells and 2 draws per impulse in 1D."""
    rho = int(params["impulse_density"])
    L = 2 if (params["nonstationary"] and params["multi_resolution_grid"]) else 1
    if params["sampling_1d"]:
        return L * (3 * (24 + (2 + 2 * rho) * 14 + rho * 6) + 3 * rho * 40)
    return L * (27 * (24 + (2 + 4 * rho) * 14 + rho * 10) + 0.155 * 27 * rho * 40)


VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12      # one simple VALU op per lane and clock: 39.3 Tops/s (MI355X_MICROARCH.md: 157.3 TF = x2 packed x2 FMA)


def load_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(kernel_key)
    except Exception:
        return None


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but grants a 16-CPU share per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, params, cores):
    """The oracle (CPU restatement, kind "port") on a bounded sample of the same workload: the
    C1 scene with the same camera at 1/32 of the linear resolution (60x34 pixels, same spp)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_bindings as ob
    orc = ob.Oracle(params, threads=cores)
    w, h, spp = 60, 34, 64
    scene = ob.default_scene_s(w, h, spp)
    t0 = time.perf_counter()
    orc.render_scene_s(scene)
    dt = time.perf_counter() - t0
    n_eval, n_seg = orc.counters()
    return {
        "value": w * h * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "scene S, C1 medium, same camera at %dx%d, %d spp (%d samples, %.1f s, %d noise evals)" % (w, h, spp, w * h * spp, dt, n_eval),
        "us_per_eval": dt * cores / max(n_eval, 1) * 1e6,
    }


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` with no rank environment: start the N ranks ourselves, as CHILD processes of a
    parent that never touches the GPU (no torch.cuda call, no libgpis call: a process that has initialised the
    GPU must not exec or fork GPU workers), with the launcher the driver would use, and relay rank 0's one JSON
    line and the job's exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in proc.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line is not None:
        print(line)
    elif proc.returncode == 0:
        print("bench.py launcher: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def dry_run(args, world, rank):
    """Launcher / sharding self-test without a GPU (tests/test_bench_launcher.py): the same process-group set-up,
    parameter broadcast, tile-row sharding and reduce as the real run, over gloo, with a renderer that writes a
    known function of the pixel index — so rank 0 can check that every pixel was rendered exactly once."""
    import torch
    import torch.distributed as dist
    import _gpis_pkg
    pkg = _gpis_pkg.load_package()        # numpy mirrors only: the library is not loaded
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    params = pkg.params_for_config(args.config) if rank == 0 else np.zeros((), dtype=pkg.PARAMS)
    params = pkg.dist.broadcast_params(params, pkg.PARAMS, dist)
    W, H, spp = args.width, args.height, args.spp
    scene = pkg.default_scene_s(W, H, spp)
    rad = torch.zeros(H * W, dtype=torch.float32)

    def render_into(part, acc):
        k = int(part["spp_count"])
        for y in pkg.dist.shard_rows(part, int(part["shard_index"]), max(int(part["shard_count"]), 1), int(part["tile_size"])):
            idx = torch.arange(y * W, (y + 1) * W)
            acc[idx] += (1 + idx % 7).to(torch.float32) * k

    pkg.dist.render_sharded(scene, render_into, rad, dist=dist if world > 1 else None, mode=args.shard)
    if rank == 0:
        idx = torch.arange(H * W)
        want = (1 + idx % 7).to(torch.float32) * pkg.dist.total_spp(scene, world, args.shard)
        print(json.dumps({"metric": "dry-run (no GPU work)", "dry_run": True, "value": None, "n_gpus": world,
                          "ranks_seen": dist.get_world_size() if world > 1 else 1, "shard": args.shard,
                          "coverage_ok": bool(torch.equal(rad, want)), "impulse_density": float(params["impulse_density"])}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C1")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--shard", choices=["rows", "spp"], default="rows",
                    help="rows: 16-pixel tile rows dealt round-robin to the ranks — the tile split of the reference's "
                         "integrator, the SAME image at every N (strong scaling); spp: every rank renders its own spp "
                         "slice of every pixel (weak scaling, N times the samples)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / sharding self-test over gloo on CPU: no GPU work, no metric")
    ap.add_argument("--guide", default="16:64", help="certified guide field 'half_extent_cells:points_per_cell' for "
                    "single-realization media, or 'off' (built once before the timed region: 34 GB / 3.3 s at 16:64, "
                    "4.3 GB / 0.3 s at 16:32; falls back to 16:32 if the allocation fails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the driver starts it (`python bench.py --gpus N`): become the launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    import _gpis_pkg
    pkg = _gpis_pkg.load_package()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # rank 0 owns the scene/kernel parameters; broadcast the POD block over RCCL
    params = pkg.params_for_config(args.config) if rank == 0 else np.zeros((), dtype=pkg.PARAMS)
    params = pkg.dist.broadcast_params(params, pkg.PARAMS, dist, device="cuda")

    med = pkg.Medium(params, device=local_rank)
    lib = med.L.lib
    guide_info = None
    if args.guide != "off" and int(med.derived()["fast_path"]):
        half, ppc = (int(x) for x in args.guide.split(":"))
        t_g = time.perf_counter()
        try:
            med.build_guide(half, ppc)
        except RuntimeError:
            half, ppc = 16, 32
            med.build_guide(half, ppc)
        guide_info = {"half_extent_cells": half, "points_per_cell": ppc, "bytes": (2 * half * ppc) ** 3 * 4,
                      "build_s": time.perf_counter() - t_g}
    W, H, spp = args.width, args.height, args.spp
    scene = np.zeros((), dtype=pkg.SCENE_S)
    lib.gpis_default_scene_s(scene.ctypes.data, W, H, spp)
    rad = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def render_into(part, acc):
        med.call("gpis_render_scene_s", part.ctypes.data_as(ctypes.c_void_p), acc.data_ptr(), None, stream)

    def step():
        rad.zero_()
        # weak scaling ("spp"): each rank renders its own spp slice of every pixel, then one
        # reduce(sum) to rank 0; "rows" deals 16-pixel tile rows round-robin (strong scaling)
        pkg.dist.render_sharded(scene, render_into, rad, dist=dist if world > 1 else None, mode=args.shard)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()      # set-up, like the guide build: the first render allocates the driver's workspace (50 GB for a C1 frame)
    for _ in range(args.warmup):
        step()
    fence()
    med.reset_counters()
    med.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    med.set_profiling(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    prof = [med.kernel_profile(k) for k in (0, 1)]     # (ms, launches, n_eval, n_seg)
    n_guide = med.guide_steps() if guide_info else 0
    if rank == 0:
        total_samples = W * H * pkg.dist.total_spp(scene, world, args.shard) * args.steps
        b_eval = algorithmic_bytes_per_eval(params)
        names = ("sample_distance", "transmittance")
        dom = 0 if prof[0][0] >= prof[1][0] else 1
        ms, launches, n_eval, n_seg = prof[dom]
        bytes_alg = n_eval * b_eval + n_seg * B_SEG
        achieved = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # the same figure priced at the evaluations the REFERENCE algorithm performs on these segments:
        # every march step the guide certified stands for one evaluateValue of GPM.cpp:189-193
        n_eval_all = prof[0][2] + prof[1][2] + n_guide
        n_seg_all = prof[0][3] + prof[1][3]
        ms_all = prof[0][0] + prof[1][0]
        ref_equiv = (n_eval_all * b_eval + n_seg_all * B_SEG) / (ms_all * 1e-3) / 1e9 if ms_all > 0 else 0.0
        fast = int(med.derived()["fast_path"])
        kernel = ("k_guided_" if guide_info else ("k_fast_" if fast else "k_")) + names[dom]
        res = {
            "metric": "Msamples/s (primary rays x spp / s)", "value": total_samples / dt_max / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.shard == "spp" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: scene S %dx%d, %d spp/GPU, SparseConvolutionNoiseMedium (3D isotropic, "
                                   "impulse_density=%d, ctx=renewal, single_realization)" % (args.config, W, H, spp, int(params["impulse_density"]))
                       if args.config == "C1" else "%s: scene S %dx%d, %d spp/GPU" % (args.config, W, H, spp),
                       "sharding": ("%s shards per rank + reduce(sum) to rank 0" % args.shard) if world > 1 else "single GPU",
                       "kernel_path": ("guided (certified guide field + wave-cooperative exact evaluations)" if guide_info else
                                       "fast (wave-cooperative)") if fast else "generic (on-the-fly impulses)",
                       "guide": guide_info},
            "roofline": {
                "bound": "hbm", "kernel": kernel,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(kernel),
                "algorithmic_bytes_per_launch": bytes_alg / max(launches, 1),
                "avg_launch_ms": ms / max(launches, 1), "launches": launches,
                "bytes_per_eval": b_eval, "bytes_per_segment": B_SEG, "n_eval": n_eval, "n_seg": n_seg,
                "evals_per_s": (prof[0][2] + prof[1][2]) / dt_max,
                "certified_steps": n_guide, "certified_steps_per_s": n_guide / dt_max,
                "valu": {"doc": "secondary roof (SURVEY.md 8d): algorithmic VALU ops of the exact evaluations the dominant kernel performed / its time; "
                                "exceeds the peak because one generated or gathered impulse serves the 64 queries of a wave",
                         "ops_per_eval": algorithmic_ops_per_eval(params), "achieved_Tops": n_eval * algorithmic_ops_per_eval(params) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                         "peak_Tops": VALU_PEAK_TOPS},
                "reference_equivalent": {"evals": n_eval_all, "segments": n_seg_all, "GBps": ref_equiv,
                                         "doc": "both medium kernels, counting a certified march step as the evaluation it replaces"},
                "kernel_ms": {names[0]: prof[0][0], names[1]: prof[1][0]},
                "note": "impulses are generated or gathered on chip; the binding roof is VALU integer/fp32 issue, "
                        "the HBM figure uses SURVEY.md 8d's algorithmic-bytes definition",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            cores = usable_cores()
            res["cpu_baseline"] = cpu_baseline(pkg, params, cores)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
