"""GPU-box measurement of the wavefront drivers beyond scene S.
usage: python tools/paths_bench.py [--config C1] [--width 1920 --height 1080 --spp 16] [--bounces 4] [--guide 16:64]
       python tools/paths_bench.py --mode nee --config C2 --guide off [--width 960 --height 540 --spp 8]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _gpis_pkg

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["paths", "nee"], default="paths")
ap.add_argument("--config", default="C1")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--bounces", type=int, default=4)
ap.add_argument("--albedo", type=float, default=0.8)
ap.add_argument("--guide", default="16:64")
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
import torch
pkg = _gpis_pkg.load_package()
med = pkg.Medium(pkg.params_for_config(a.config))
if a.guide != "off":
    half, ppc = (int(v) for v in a.guide.split(":"))
    med.build_guide(half, ppc)
scene = np.zeros((), dtype=pkg.SCENE_S)
med.L.lib.gpis_default_scene_s(scene.ctypes.data, a.width, a.height, a.spp)
rad = torch.zeros(a.width * a.height, dtype=torch.float32, device="cuda")
med.set_profiling(True) if hasattr(med, "set_profiling") else None
out = []
for r in range(a.reps + 1):
    rad.zero_(); med.reset_counters()
    torch.cuda.synchronize(); t = time.time()
    if a.mode == "paths":
        med.call("gpis_render_scene_s_paths", scene.ctypes.data_as(ctypes.c_void_p), a.bounces, a.albedo, rad.data_ptr(), None)
    else:
        surf = np.array(pkg.default_surface_s(), dtype=pkg.SURFACE_S)
        med.call("gpis_render_scene_s_nee", scene.ctypes.data_as(ctypes.c_void_p), surf.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None)
    torch.cuda.synchronize(); dt = time.time() - t
    if r:
        out.append(dt)
ev, seg = med.counters()
n = a.width * a.height * a.spp
wl = ("scene-S paths %s %dx%dx%d bounces=%d albedo=%g guide=%s" % (a.config, a.width, a.height, a.spp, a.bounces, a.albedo, a.guide)
      if a.mode == "paths" else "scene-S conductor NEE %s %dx%dx%d" % (a.config, a.width, a.height, a.spp))
print(json.dumps({"workload": wl,
                  "Mpaths_per_s": n / min(out) / 1e6, "Msegments_per_s": seg / min(out) / 1e6, "segments_per_path": seg / n,
                  "exact_evals_per_segment": ev / max(seg, 1), "ms": min(out) * 1e3, "image_sum": float(rad.sum())}))
