"""phase cycles of k_fs_march (library built with -DGPIS_FS_PROF, GPIS_LIBRARY points at it)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gpis_pkg
pkg = _gpis_pkg.load_package()
import torch
names = ["tridiag", "Q form", "QR sweeps", "sort", "pinv product", "cov builds", "sol = P s12", "mean + S update", "cholesky", "variates + T z", "whole segment"]
rng = np.random.default_rng(2)
N = 4096
p = pkg.params_for_config("C4"); p["mean"]["type"] = pkg.MEAN_TYPE.HOMOGENEOUS; p["mean"]["offset"] = 1.0
med = pkg.Medium(p)
lib = med.L.lib
r = np.zeros(N, dtype=pkg.RAY_IN); r["pos"] = rng.uniform(-0.5, 0.5, (N, 3)); r["dir"] = (0, 0, 1); r["far_t"] = 0.64; r["first_scatter"] = 1
st = np.zeros(N, dtype=pkg.FS_STATE); st["sampler_state"] = rng.integers(1, 2**63, size=N, dtype=np.uint64)
out = (ctypes.c_ulonglong * 16)()
for label in ("first (unconditioned)", "second (global context)"):
    lib.gpis_fs_prof_read(out, 1)
    o, st = med.fs_sample_distance(r, st)
    lib.gpis_fs_prof_read(out, 1)
    tot = out[10]
    print(label, "cycles per segment %.0f" % (tot / N))
    for i, nm in enumerate(names[:10]):
        print("   %-18s %5.1f %%  (%.0f cycles/segment)" % (nm, 100.0 * out[i] / tot, out[i] / N))
    r = r.copy(); r["first_scatter"] = 0; r["bounce"] = 1
