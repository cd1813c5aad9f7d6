#!/usr/bin/env python
"""bench.py — Msamples/s of the sparse-convolution GPIS hot path on scene S.

A "step" renders one full frame of the configuration BASELINE.json quotes the metric on
(C1: 1920x1080, 64 spp, 3D isotropic sampling, impulse_density=32, renewal, single realization):
primary sampleDistance, shading, one shadow transmittance per hit, per-pixel accumulation.
Inputs (the medium's constants, the guide field, the workspace) are resident in HBM before the timed
region; `value_cold` is the same frame with every one-off cost inside the timer, `value_unguided` the
frame without the guide field (every march step evaluated exactly).

`python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed / RCCL): rank 0
broadcasts the parameter block, the image's 16-pixel tile rows are dealt round-robin to the ranks (the tile
split of the reference's integrator: the SAME image at every N, strong scaling), and the disjoint tile rows
are assembled with ONE gather to rank 0 inside the timed region.

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
N_SIMD = 256 * 4               # CUs x SIMDs
PEAK_CLOCK_HZ = 2.4e9          # MI355X_MICROARCH.md "Max clock"


def algorithmic_bytes_per_eval(params):
    """27*rho*16*L bytes (3D) or 3*rho*8*L bytes (1D); L=2 for multi-resolution."""
    rho = int(params["impulse_density"])
    L = 2 if (params["nonstationary"] and params["multi_resolution_grid"]) else 1
    return (3 * rho * 8 * L) if params["sampling_1d"] else (27 * rho * 16 * L)


def sources_sha256():
    """Identity of the kernels the committed counters belong to: sha256 over the HIP sources and the ABI header (the built .so
    embeds build paths, so its own hash differs between two builds of the same code)."""
    import hashlib
    csrc = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc"))) + [os.path.join(ROOT, "include", "gpis.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def issue_roofline(kernel, workload_key, avg_launch_ms, lib_path):
    """The dominant kernel against the roof that binds it: vector-instruction ISSUE.

    achieved = the issue cycles one launch of the kernel needs — per PMC instruction class, the counted wave
               instructions (rocprofv3 --pmc SQ_INSTS_VALU_*, profiles/) x the measured issue cost of that class
               (tools/valu_issue_bench: 2 SIMD cycles for a full-rate wave64 op, 4 for f64 / packed f32 / 32-bit
               multiply / 3-operand integer ops, 8 for transcendentals and v_readlane; profiles/) — divided by the
               launch duration measured live with HIP events in this run;
    peak     = 1024 SIMDs x 2.4 GHz: every SIMD issuing on every cycle of the launch (the launch cannot use more).
    The counters belong to one launch of the SAME workload with the SAME library (sha256 compared); they are
    deterministic for it.  `traffic` = FETCH_SIZE + WRITE_SIZE of the same launch."""
    import glob
    model, k, src = None, None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_issue_model*.json")), reverse=True):     # one file per profiled workload, the latest round first
        try:
            cand = json.load(open(path))
        except Exception:
            continue
        model = model or cand
        if cand.get("workload") == workload_key and kernel in cand.get("kernels", {}):
            model, k, src = cand, cand["kernels"][kernel], os.path.relpath(path, ROOT)
            break
    if model is None:
        return None
    if k is None:
        return {"bound": "valu_issue", "kernel": kernel, "achieved": None, "peak": N_SIMD * PEAK_CLOCK_HZ / 1e12, "unit": "T SIMD issue cycles/s",
                "frac": None, "traffic": None, "note": "no committed instruction counters for this kernel / workload (%s)" % workload_key}
    stale = model.get("source_sha256") != sources_sha256()
    sec = avg_launch_ms * 1e-3
    peak = N_SIMD * PEAK_CLOCK_HZ
    cyc = k["issue_cycles"]
    traffic = (k["traffic_bytes"]["fetch"] + k["traffic_bytes"]["write"]) if k.get("traffic_bytes") else None
    return {
        "bound": "valu_issue", "kernel": kernel,
        # the fitted estimate cannot exceed the roof: where the fit is poor (C3: an 18 k-instruction outer loop the 12 class counters
        # cannot separate) it can come out above 1 — then the roof is the statement, and `fit.relative_residual` says why
        "achieved": min(cyc["model"] / sec, peak) / 1e12, "peak": peak / 1e12, "unit": "T SIMD issue cycles/s",
        "frac": min(cyc["model"] / sec / peak, 1.0), # lo: every instruction at the full rate except the classes whose cost is known exactly; hi: SQ_ACTIVE_INST_VALU x 4, a
        # counter with 4-cycle granularity that also charges 4 cycles to a 2-cycle instruction — an upper bound, capped at the roof
        "frac_range": [cyc["lo"] / sec / peak, min(1.0, cyc["hi"] / sec / peak) if cyc.get("hi") else None],
        "traffic": traffic,
        "hbm": None if traffic is None else {"bytes_per_launch": traffic, "GBps": traffic / sec / 1e9, "frac_of_peak": traffic / sec / 1e9 / HBM_PEAK_GBS,
                                              "compulsory_bytes_per_launch": k.get("compulsory_bytes")},
        "issue_cycles_per_launch": cyc, "valu_instructions_per_launch": k["valu_insts"],
        "mean_cycles_per_valu_instruction": cyc["model"] / max(k["valu_insts"], 1),
        "fit": {"relative_residual": k["fit"]["relative_residual"], "loops_used": k["fit"]["loops_used"]} if k.get("fit") else None,
        "avg_launch_ms": avg_launch_ms, "profiled_launch_ms": k.get("launch_ms_profiled"),
        "counters_source": src + " (" + model.get("collected", "") + ")", "counters_stale": stale,
    }


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

    st = pkg.dist.render_sharded(scene, render_into, rad, dist=dist if world > 1 else None, mode=args.shard)
    # what this rank's driver call would do on the GPU: the scene-S driver marches its share in chunks of at most 2^27 samples
    # (375 B of workspace per sample, csrc/gpis_hip.hip: gpis_render_scene_s), one launch per stage and chunk
    my_rows = len(pkg.dist.shard_rows(scene, rank, world)) if args.shard == "rows" else H
    my_samples = my_rows * W * spp
    chunk_samples = min(((1 << 27) // spp) * spp, my_samples) if my_samples else 0
    plan = torch.tensor([my_rows, my_samples, -(-my_samples // chunk_samples) if chunk_samples else 0, chunk_samples * 375, st["wire_bytes"]], dtype=torch.float64)
    plans = [torch.zeros_like(plan) for _ in range(world)]
    if world > 1:
        dist.all_gather(plans, plan)
    else:
        plans = [plan]
    if rank == 0:
        idx = torch.arange(H * W)
        want = (1 + idx % 7).to(torch.float32) * pkg.dist.total_spp(scene, world, args.shard)
        print(json.dumps({"metric": "dry-run (no GPU work)", "dry_run": True, "value": None, "n_gpus": world,
                          "ranks_seen": dist.get_world_size() if world > 1 else 1, "shard": args.shard,
                          "coverage_ok": bool(torch.equal(rad, want)), "impulse_density": float(params["impulse_density"]),
                          "plan": {"rows_per_rank": [int(p[0]) for p in plans], "samples_per_rank": [int(p[1]) for p in plans],
                                   "chunks_per_rank": [int(p[2]) for p in plans], "workspace_bytes_per_rank": [int(p[3]) for p in plans],
                                   "wire_bytes_per_rank": [int(p[4]) for p in plans],
                                   "samples_total": int(sum(p[1] for p in plans))}}))
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
                    "single-realization media, or 'off' (built once before the timed region; only the bricks near the surface are "
                    "tabulated: a few GB at 16:64 for scene S; falls back to 16:32 if the allocation fails)")
    ap.add_argument("--estimator", choices=["auto", "lambert", "nee"], default="auto",
                    help="auto: C2 (1D sampling, MIS, conductor) renders through the conductor NEE estimator (gpis_render_scene_s_nee, "
                         "TraceBase.cpp:346-420 / ConductorBsdf.cpp:68-137) as BASELINE.json states it, everything else through scene S's "
                         "Lambert + one shadow ray estimator")
    ap.add_argument("--guide-cold", default="16:32", help="guide field of the cold frame ('same' = --guide): a one-frame render is fastest "
                    "with the coarser field (0.07 s to build); the warm frames run on --guide")
    ap.add_argument("--reserve-thread", action="store_true", help="cold frame: gpis_reserve_scene_workspace from a second thread during the guide build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unguided", action="store_true", help="skip the extra unguided frame (value_unguided)")
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

    W, H, spp = args.width, args.height, args.spp
    scene = pkg.default_scene_s(W, H, spp)
    rad = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- cold frame: everything a first frame pays — medium constants + cell table, guide field, the driver's
    #      workspace (4 GB: a handle's first frame runs in 16 Mi-sample chunks) and the frame itself — inside ONE timer
    fence()
    t_cold0 = time.perf_counter()
    med = pkg.Medium(params, device=local_rank)
    torch.cuda.synchronize()
    t_created = time.perf_counter()
    lib = med.L.lib
    use_nee = args.estimator == "nee" or (args.estimator == "auto" and args.config == "C2")
    # --reserve-thread: the Lambert driver's workspace is reserved from a second host thread while the guide field is built.
    # Measured ("cold frame"): on never-touched VRAM the whole-frame allocation then stalls the guide build's own
    # allocations and launches (guide 0.23 -> 1.19 s), so it is off by default; the driver allocates on demand instead.
    reserve = None
    if not use_nee and args.reserve_thread:
        import threading
        parts = pkg.dist.shard_scene(scene, rank, world, args.shard)
        reserve_rc = []
        reserve = threading.Thread(target=lambda: reserve_rc.extend(
            lib.gpis_reserve_scene_workspace(med.h, p.ctypes.data_as(ctypes.c_void_p)) for p in parts))
        reserve.start()
    guide_info = None
    guide_cold = None

    def build_guide_as(spec):
        half, ppc = (int(x) for x in spec.split(":"))
        t_g = time.perf_counter()
        try:
            med.build_guide(half, ppc)
        except RuntimeError:
            half, ppc = 16, 32
            med.build_guide(half, ppc)
        gi = med.guide_info()
        return {"half_extent_cells": half, "points_per_cell": ppc, "bytes": gi["bytes_samples"] + gi["bytes_bounds"],
                "bytes_dense_equivalent": gi["bytes_dense"], "bricks_tabulated": gi["bricks_allocated"], "bricks_total": gi["bricks_total"],
                "build_s": time.perf_counter() - t_g}

    if args.guide != "off" and int(med.derived()["fast_path"]):
        # The cold frame is what a ONE-frame render sees, so it gets the guide resolution that is best for one frame: 16:32
        # builds in 0.07 s instead of 0.23 s and costs 0.04 s of the frame (measured, gpurun session s65); the warm frames
        # below run on --guide, rebuilt outside their timer as a multi-frame render would do once.
        guide_cold = build_guide_as(args.guide if args.guide_cold == "same" else args.guide_cold)
        guide_info = guide_cold

    surf = np.array(pkg.default_surface_s(), dtype=pkg.SURFACE_S)

    def render_into(part, acc):
        if use_nee:
            med.call("gpis_render_scene_s_nee", part.ctypes.data_as(ctypes.c_void_p), surf.ctypes.data_as(ctypes.c_void_p), acc.data_ptr(), stream)
        else:
            med.call("gpis_render_scene_s", part.ctypes.data_as(ctypes.c_void_p), acc.data_ptr(), None, stream)

    rank_stats = []

    def step():
        rad.zero_()
        # "rows": 16-pixel tile rows dealt round-robin, one batch per rank, ONE gather of the disjoint tile rows to rank 0
        rank_stats.append(pkg.dist.render_sharded(scene, render_into, rad, dist=dist if world > 1 else None, mode=args.shard,
                                                  sync=torch.cuda.synchronize))

    torch.cuda.synchronize()
    t_first0 = time.perf_counter()
    step()          # does not wait for the reserve thread: the driver takes each array when it first needs it
    fence()
    dt_cold = time.perf_counter() - t_cold0
    if reserve is not None:
        reserve.join()
        assert all(rc == 0 for rc in reserve_rc), reserve_rc
    cold_parts = {"create_s": t_created - t_cold0, "guide_build_s": guide_info["build_s"] if guide_info else 0.0,
                  "guide": ("%d:%d" % (guide_cold["half_extent_cells"], guide_cold["points_per_cell"])) if guide_cold else None,
                  "first_frame_s": time.perf_counter() - t_first0}      # the first frame runs in 16 Mi-sample chunks (4 GB workspace)
    if guide_cold is not None and args.guide_cold != "same" and args.guide_cold != args.guide:
        guide_info = build_guide_as(args.guide)                           # the resolution of the warm frames; not timed
        cold_parts["guide_upgrade_s"] = guide_info["build_s"]

    # ---- warm frames: the metric
    for _ in range(args.warmup):
        step()
    fence()
    med.reset_counters()
    med.set_profiling(True)
    del rank_stats[:]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    med.set_profiling(False)

    tt = torch.tensor([dt, dt_cold], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_max, dt_cold_max = float(tt[0].item()), float(tt[1].item())
    # per-rank diagnostics (N > 1): every rank's own render and collective time per frame and its guide-field build, so that an
    # imbalance of the interleaved tile rows or one slow guide build is visible behind the max-over-ranks metric
    mine = torch.tensor([sum(r["render_s"] for r in rank_stats) / max(len(rank_stats), 1) * 1e3,
                         sum(r["collective_s"] for r in rank_stats) / max(len(rank_stats), 1) * 1e3,
                         guide_info["build_s"] if guide_info else 0.0, float(rank_stats[-1]["wire_bytes"]) if rank_stats else 0.0], dtype=torch.float64, device="cuda")
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = [[float(v) for v in t.cpu()] for t in per_rank]

    prof = [med.kernel_profile(k) for k in (0, 1)]     # (ms, launches, n_eval, n_seg)
    prof_nee = med.kernel_profile(2) if use_nee else None
    n_guide = med.guide_steps() if guide_info else 0

    # ---- the same frame without the guide field: every march step evaluated exactly (N = 1 only; one frame) — and the
    #      run's own parity check: the guided frame that was timed must equal this one bit for bit (the certificate only ever
    #      replaces evaluations whose sign it has proven, SCNM.cpp:132-174), otherwise the bench fails
    dt_unguided, frame_bit_identical = None, None
    if guide_info and world == 1 and not args.no_unguided:
        rad_guided = rad.clone()
        med.drop_guide()
        fence()
        t1 = time.perf_counter()
        step()
        fence()
        dt_unguided = time.perf_counter() - t1
        frame_bit_identical = bool(torch.equal(rad, rad_guided))
        if not frame_bit_identical:
            n_diff = int((rad != rad_guided).sum().item())
            print("bench.py: the guided frame differs from the unguided frame in %d of %d pixels" % (n_diff, rad.numel()), file=sys.stderr)

    if rank == 0:
        samples_per_step = W * H * pkg.dist.total_spp(scene, world, args.shard)
        total_samples = samples_per_step * args.steps
        b_eval = algorithmic_bytes_per_eval(params)
        names = ("sample_distance", "transmittance")
        dom = 0 if prof[0][0] >= prof[1][0] else 1
        ms, launches, n_eval, n_seg = prof[dom]
        n_eval_all = prof[0][2] + prof[1][2]
        n_seg_all = prof[0][3] + prof[1][3]
        fast = int(med.derived()["fast_path"])
        persistent = (not fast) and med.get_option("persistent") == 1
        kernel = ("k_guided_" if guide_info else ("k_fast_" if fast else ("k_persist_march_" if persistent else "k_"))) + names[dom]
        workload_key = "%s %dx%dx%d guide %s n_gpus %d" % (args.config, W, H, spp, ("%d:%d" % (guide_info["half_extent_cells"], guide_info["points_per_cell"])) if guide_info else "off", world)
        roof = issue_roofline(kernel, workload_key, ms / max(launches, 1), med.L.path)
        if roof is None:
            roof = {"bound": "valu_issue", "kernel": kernel, "achieved": None, "peak": N_SIMD * PEAK_CLOCK_HZ / 1e12, "unit": "T SIMD issue cycles/s",
                    "frac": None, "traffic": None, "note": "profiles/r02_issue_model.json not found"}
        if True:
            roof["kernel_ms"] = {names[0]: prof[0][0] / max(prof[0][1], 1), names[1]: prof[1][0] / max(prof[1][1], 1)}     # per launch
            if prof_nee:
                # per frame: the march kernels (one sampleDistance, two masked transmittance launches) and the two k_nee launches
                per_frame = {"sample_distance": prof[0][0] / args.steps, "transmittance": prof[1][0] / args.steps, "k_nee": prof_nee[0] / args.steps}
                roof["kernel_ms_per_frame"] = per_frame
                roof["k_nee_share_of_kernel_time"] = per_frame["k_nee"] / max(sum(per_frame.values()), 1e-9)
            roof["launches"] = launches
            roof["evals_per_s"] = n_eval_all / dt_max
            roof["certified_steps_per_s"] = n_guide / dt_max
            # SURVEY.md 8d's bookkeeping figures — labelled, and never divided by a hardware peak: impulses are generated
            # or served from cache, not loaded from HBM, and one generated impulse serves the 64 queries of a wave
            roof["algorithmic"] = {"bytes_per_eval": b_eval, "bytes_per_segment": B_SEG, "n_eval": n_eval, "n_seg": n_seg,
                                   "certified_steps": n_guide,
                                   "reference_equivalent_evals_per_s": (n_eval_all + n_guide) / dt_max,
                                   "doc": "exact evaluations the kernels performed; a certified march step stands for one evaluateValue of the reference"}
        res = {
            "metric": "Msamples/s (primary rays x spp / s)", "value": total_samples / dt_max / 1e6, "unit": "Msamples/s",
            "value_cold": samples_per_step / dt_cold_max / 1e6,
            "value_unguided": (samples_per_step / dt_unguided / 1e6) if dt_unguided else None,
            "frame_bit_identical": frame_bit_identical,
            "n_gpus": world, "ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.shard == "spp" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: scene S %dx%d, %d spp, SparseConvolutionNoiseMedium (3D isotropic, "
                                   "impulse_density=%d, ctx=renewal, single_realization)" % (args.config, W, H, spp, int(params["impulse_density"]))
                       if args.config == "C1" else
                       ("C2: scene S %dx%d, %d spp, 1D sampling, 1D_sampling_scheme=mis (NEE on), Renewal+ memory, conductor BSDF "
                        "(gpis_render_scene_s_nee: volumeLightSample + volumePhaseSample with neePDF / neeGrad, one cap light)" % (W, H, spp)
                        if use_nee else "%s: scene S %dx%d, %d spp" % (args.config, W, H, spp)),
                       "estimator": "conductor NEE / MIS (TraceBase.cpp:346-420, ConductorBsdf.cpp:68-137)" if use_nee else "Lambert + one shadow ray",
                       "sharding": ("%s: %d-pixel tile rows dealt round-robin, one batch per rank, one gather of the disjoint tile rows to rank 0" % (args.shard, pkg.dist.rows_tile(scene, world))
                                    if args.shard == "rows" else "spp slices per rank + reduce(sum) to rank 0") if world > 1 else "single GPU",
                       "kernel_path": ("guided (certified guide field + wave-cooperative exact evaluations)" if guide_info else
                                       "fast (wave-cooperative)") if fast else ("per-path: persistent refilling march" if persistent else "per-path: one ray per lane"),
                       "guide": guide_info,
                       "value_cold_doc": "one frame with gpis_create (cell table), the guide-field build (--guide-cold: the resolution that is fastest for a single frame) and the workspace allocation inside the timer (%.2f s)" % dt_cold_max,
                       "value_cold_parts": cold_parts,
                       "value_unguided_doc": "one frame after gpis_drop_guide: every march step is an exact wave-cooperative evaluation" if dt_unguided else None},
            "roofline": roof,
            "per_rank": None if world == 1 else {
                "render_ms": {"min": min(p[0] for p in per_rank), "max": max(p[0] for p in per_rank), "mean": sum(p[0] for p in per_rank) / world, "all": [p[0] for p in per_rank]},
                "collective_ms": {"max": max(p[1] for p in per_rank), "all": [p[1] for p in per_rank]},
                "guide_build_s": [p[2] for p in per_rank], "wire_bytes_per_frame": [int(p[3]) for p in per_rank],
                "doc": "each rank's own clock around its share of a frame (device synchronised) and around the gather / reduce"},
        }
        if not args.no_cpu_baseline and world == 1:
            cores = usable_cores()
            res["cpu_baseline"] = cpu_baseline(pkg, params, cores)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()
    if frame_bit_identical is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
