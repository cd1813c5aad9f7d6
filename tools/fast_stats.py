"""Diagnostic (GPU box only): builds the library with -DGPIS_FAST_STATS into gpurun_out/ and prints the
wave-level work counters of the cooperative loop for a small scene-S render."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gpis_pkg
pkg = _gpis_pkg.load_package()
so = os.path.join(ROOT, "gpurun_out", "libgpis_hip_stats.so")
csrc = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "csrc")
prebuilt = os.path.join(ROOT, "build", "stats", "libgpis_hip_stats.so")
if os.path.exists(prebuilt):
    so = prebuilt
else:
    import __graft_entry__ as g
    g.build_hip(extra_flags=("-DGPIS_FAST_STATS",), out=so)        # every translation unit with the counters compiled in
import torch
lib = pkg.GpisLib(so)
w, h, spp = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (480, 270, 64)))
med = pkg.Medium(pkg.params_for_config(sys.argv[4] if len(sys.argv) > 4 else "C1"), lib=lib)
scene = np.zeros((), dtype=pkg.SCENE_S)
lib.lib.gpis_default_scene_s(scene.ctypes.data, w, h, spp)
if len(sys.argv) > 5 and sys.argv[5] != "off":
    half, ppc = (int(x) for x in sys.argv[5].split(":"))
    med.build_guide(half, ppc)
rad = torch.zeros(h * w, dtype=torch.float32, device="cuda")
paths_bounces = int(os.environ.get("STATS_PATHS_BOUNCES", "0"))     # > 0: the multi-bounce driver instead of scene S
if paths_bounces:
    med.call("gpis_render_scene_s_paths", scene.ctypes.data_as(ctypes.c_void_p), paths_bounces, 0.8, rad.data_ptr(), None)
else:
    med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 32)()
lib.lib.gpis_debug_fast_stats(out)
names = ["wave_evals", "active_lanes", "cells_visited", "cells_mine", "cells_cand", "candidates", "union_pass", "lane_pass", "incoherent",
         "lanes_1_2", "lanes_3_4", "lanes_5_8", "lanes_9_16", "lanes_17_32", "lanes_33_64"]
st = dict(zip(names, list(out)))
e = max(st["wave_evals"], 1)
print(st)
print("per wave-eval: active lanes %.1f, cells visited %.1f, processed %.1f, with candidates %.1f, candidates %.1f, bodies %.1f, "
      "lane-passes/body %.1f, incoherent %.4f" % (st["active_lanes"] / e, st["cells_visited"] / e, st["cells_mine"] / e, st["cells_cand"] / e,
                                                 st["candidates"] / e, st["union_pass"] / e, st["lane_pass"] / max(st["union_pass"], 1), st["incoherent"] / e))
g = list(out)[16:26]
if g[2]:
    tot = g[2] + g[3]
    print("guided kernels (wave cycles, both kernels): guide loop %.1f%%, exact values %.1f%%, gradient tail %.1f%%; "
          "guide-loop iterations %d (%.1f lanes stepping), exact rounds %d (%.1f lanes parked), of which sideways %d (%.1f lanes)" % (
          100.0 * g[0] / tot, 100.0 * g[1] / tot, 100.0 * g[3] / tot, g[4], g[5] / max(g[4], 1), g[6], g[7] / max(g[6], 1), g[8], g[9] / max(g[8], 1)))
ph = list(out)[26:31]
if sum(ph):
    print("exact value requests served, by phase: " + ", ".join("%s %d (%.2f/segment)" % (n, v, v / max(med.counters()[1], 1))
                                                               for n, v in zip(("X_F0", "X_CUR", "X_PREV", "X_REFINE", "X_FINAL"), ph)))
print("lane evals", med.counters(), "guide steps", med.guide_steps())
print("active lanes per exact wave-eval: %.1f" % (med.counters()[0] / e))
