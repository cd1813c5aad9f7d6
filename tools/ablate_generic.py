"""Diagnostic (GPU box): times scene-S renders of the lane-per-ray configurations (C2, C3) with each
build/variants/libgpis_*.so (one process each)."""
import ctypes, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    sys.path.insert(0, ROOT)
    import _gpis_pkg, torch
    pkg = _gpis_pkg.load_package()
    lib = pkg.GpisLib(sys.argv[2])
    res = {"so": os.path.basename(sys.argv[2])}
    for cfg, (w, h, spp) in (("C2", (480, 270, 8)), ("C3", (240, 136, 4))):
        med = pkg.Medium(pkg.params_for_config(cfg), lib=lib)
        scene = np.zeros((), dtype=pkg.SCENE_S)
        lib.lib.gpis_default_scene_s(scene.ctypes.data, w, h, spp)
        rad = torch.zeros(h * w, dtype=torch.float32, device="cuda")
        best = None
        for rep in range(2):
            rad.zero_(); med.reset_counters(); med.set_profiling(True)
            med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
            a, b = med.kernel_profile(0), med.kernel_profile(1)
            t = a[0] + b[0]
            best = t if best is None or t < best else best
        res[cfg] = {"ms": best, "Msamples_per_s": w * h * spp / best / 1e3, "sum": float(rad.sum().item())}
    print(json.dumps(res))
else:
    for so in sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so"))):
        out = subprocess.run([sys.executable, __file__, "--one", so], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
