"""function-space kernel: rate against the number of sample points (where does the time go: n^2 or n^3 terms)"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gpis_pkg
pkg = _gpis_pkg.load_package()
import torch
dev = torch.device("cuda", 0)
rng = np.random.default_rng(2)
N = 8192
for n, off in ((8, 1.0), (16, 1.0), (32, 1.0), (64, 1.0), (64, 0.0)):
    p = pkg.params_for_config("C4")
    p["mean"]["type"] = pkg.MEAN_TYPE.HOMOGENEOUS
    p["mean"]["offset"] = off
    p["fs_sample_points"] = n
    med = pkg.Medium(p)
    r = np.zeros(N, dtype=pkg.RAY_IN)
    r["pos"] = rng.uniform(-0.5, 0.5, (N, 3)); r["dir"] = (0, 0, 1); r["far_t"] = 0.64; r["first_scatter"] = 1
    st = np.zeros(N, dtype=pkg.FS_STATE); st["sampler_state"] = rng.integers(1, 2**63, size=N, dtype=np.uint64)
    d_r = torch.from_numpy(r.view(np.uint8).reshape(-1)).to(dev); d_s = torch.from_numpy(st.view(np.uint8).reshape(-1)).to(dev)
    d_o = torch.zeros(N * pkg.SEG_OUT.itemsize, dtype=torch.uint8, device=dev)
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        med.L.check(med.L.lib.gpis_fs_sample_distance_batch(med.h, N, ctypes.c_void_p(d_r.data_ptr()), ctypes.c_void_p(d_s.data_ptr()), ctypes.c_void_p(d_o.data_ptr()), None), "fs")
        torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
        r2 = r.copy(); r2["first_scatter"] = 0; r2["bounce"] = 1
        d_r = torch.from_numpy(r2.view(np.uint8).reshape(-1)).to(dev)
    print("n=%d offset=%.1f  first %.0f seg/s (%.1f us/seg/CU), conditioned %.0f, %.0f seg/s" % (n, off, N / res[0], res[0] / N * 256 * 1e6, N / res[1], N / res[2]), flush=True)
