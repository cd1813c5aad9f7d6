"""Diagnostic: where the time of gpis_build_guide goes (allocation vs kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gpis_pkg
pkg = _gpis_pkg.load_package()
import torch
med = pkg.Medium(pkg.params_for_config("C1"))
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter(); med.build_guide(16, 64); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("build_guide(16, 64) call %d: %.3f s" % (i, t1 - t0))
med.drop_guide()
for i in range(2):
    t0 = time.perf_counter(); x = torch.empty(34 * 2**30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("torch.empty(34 GiB) %d: %.3f s" % (i, t1 - t0)); del x; torch.cuda.empty_cache()
