"""Diagnostic (GPU box): times scene-S renders with each build/variants/libgpis_*.so (one process each)."""
import ctypes, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    sys.path.insert(0, ROOT)
    import _gpis_pkg, torch
    pkg = _gpis_pkg.load_package()
    lib = pkg.GpisLib(sys.argv[2])
    w, h, spp = (int(x) for x in os.environ.get("ABLATE_RES", "960x540x64").split("x"))
    med = pkg.Medium(pkg.params_for_config("C1"), lib=lib)
    if os.environ.get("ABLATE_GUIDE", "16:32") != "off":
        gh, gp = (int(x) for x in os.environ.get("ABLATE_GUIDE", "16:32").split(":"))
        med.build_guide(gh, gp)
    scene = np.zeros((), dtype=pkg.SCENE_S)
    lib.lib.gpis_default_scene_s(scene.ctypes.data, w, h, spp)
    rad = torch.zeros(h * w, dtype=torch.float32, device="cuda")
    import time
    best, wall = None, None
    for rep in range(3):
        med.reset_counters(); med.set_profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        a, b = med.kernel_profile(0), med.kernel_profile(1)
        t = (a[0], b[0])
        best = t if best is None or sum(t) < sum(best) else best
        wall = dt if wall is None or dt < wall else wall
    print(json.dumps({"so": os.path.basename(sys.argv[2]), "sd_ms": best[0], "tr_ms": best[1], "wall_ms": wall, "sum": float(rad.sum().item())}))
else:
    for so in sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so"))):
        out = subprocess.run([sys.executable, __file__, "--one", so], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
