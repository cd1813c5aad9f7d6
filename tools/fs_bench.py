"""Config C4 (BASELINE.json): the function-space comparison path, 64 sample points, correlation context "global".
Scene-S camera segments (first segment of every path, unconditioned) followed by the shadow-like second segment conditioned on the
whole context of the first (66 x 66 pseudo-inverse + 64 x 64 square root).  Prints one JSON object: segments/s on the GPU
(device-resident records, the C-ABI call bracketed by synchronisations) and of the CPU restatement on a bounded sample."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _gpis_pkg  # noqa: E402

pkg = _gpis_pkg.load_package()


def main():
    import torch
    import oracle_bindings as ob
    W, H = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (256, 144)))
    params = pkg.params_for_config("C4")
    med = pkg.Medium(params)
    import bench
    cores = bench.usable_cores()           # the cgroup's CPU share (16 on the GPU box), not the 256 logical CPUs of the host
    orc = ob.Oracle(params, threads=cores)
    scene = ob.default_scene_s(W, H, 1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import scene_rays
    rays, _ = scene_rays(ob, orc, scene, step=1)
    n = len(rays)
    rng = np.random.default_rng(1)
    st = np.zeros(n, dtype=pkg.FS_STATE)
    st["sampler_state"] = rng.integers(1, 2**63, size=n, dtype=np.uint64)
    dev = torch.device("cuda", 0)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    def run(r, s):
        d_r, d_s = up(r), up(s)
        d_o = torch.zeros(len(r) * pkg.SEG_OUT.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = med.L.lib.gpis_fs_sample_distance_batch(med.h, len(r), ctypes.c_void_p(d_r.data_ptr()), ctypes.c_void_p(d_s.data_ptr()),
                                                     ctypes.c_void_p(d_o.data_ptr()), None)
        med.L.check(rc, "gpis_fs_sample_distance_batch")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, d_o.cpu().numpy().view(pkg.SEG_OUT).copy(), d_s.cpu().numpy().view(pkg.FS_STATE).copy()

    run(rays[:4096], st[:4096])                                   # warm-up
    dt1, out1, st1 = run(rays, st)
    # second segment: towards the light from where the first ended, conditioned on its context
    r2 = rays.copy()
    r2["pos"] = rays["pos"] + rays["dir"] * out1["sample_t"][:, None]
    light = np.asarray(scene["light_dir"], dtype=np.float64).reshape(3)
    r2["dir"] = (light / np.linalg.norm(light)).astype(np.float32)
    r2["near_t"], r2["far_t"], r2["first_scatter"], r2["bounce"] = 0.0, 1.0, 0, 1
    r2["last_aniso"] = out1["aniso"]
    dt2, out2, _ = run(r2, st1)
    m = min(n, 1500)
    t0 = time.perf_counter()
    orc.fs_sample_distance(rays[:m], st[:m])
    c1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.fs_sample_distance(r2[:m], st1[:m])
    c2 = time.perf_counter() - t0
    print(json.dumps({
        "workload": "C4: function-space path, 64 sample points, ctx global, scene S %dx%d camera segments + conditioned second segments" % (W, H),
        "segments": n, "unit": "segments/s",
        "gpu_first_segments_per_s": n / dt1, "gpu_conditioned_segments_per_s": n / dt2,
        "hit_fraction_first": float((out1["exited"] == 0).mean()), "ok_fraction_second": float((out2["ok"] == 1).mean()),
        "cpu_port": {"cores": cores, "sample": m, "first_segments_per_s": m / c1, "conditioned_segments_per_s": m / c2},
        "kernel": "k_fs_march (one wave per segment, four segments per CU: 37 KB LDS + 104 KB L2-resident workspace each, fp64)"}))


if __name__ == "__main__":
    main()
