"""The spill experiment of VERDICT round 2 / DESIGN.md 8: the resident guided sampleDistance WITHOUT its gradient tail
(GPIS_OPT_DEFER_GRAD: the march writes pending records, k_guided_range_grad completes them) against the default kernel, whole C1 frames.
Prints one JSON object: per form the frame ms, the two kernels' ms (HIP events), and that the frames are bit-identical."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _gpis_pkg
import torch
pkg = _gpis_pkg.load_package()
W, H, SPP = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 64)))
med = pkg.Medium(pkg.params_for_config("C1"))
med.build_guide(16, 64)
scene = np.zeros((), dtype=pkg.SCENE_S)
med.L.lib.gpis_default_scene_s(scene.ctypes.data, W, H, SPP)
rad = torch.zeros(H * W, dtype=torch.float32, device="cuda")
out, frames = {}, {}
for form in ("default", "defer_grad", "default", "defer_grad"):
    med.set_option("defer_grad", 1 if form == "defer_grad" else 0)
    best = None
    for rep in range(3):
        rad.zero_(); med.reset_counters(); med.set_profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        sd, tr = med.kernel_profile(0), med.kernel_profile(1)
        med.set_profiling(False)
        cur = {"frame_ms": dt, "sample_distance_ms": sd[0], "transmittance_ms": tr[0]}
        best = cur if best is None or cur["frame_ms"] < best["frame_ms"] else best
    out.setdefault(form, []).append(best)
    frames[form] = rad.clone()
print(json.dumps({"workload": "C1 %dx%dx%d guide 16:64" % (W, H, SPP), "frames_bit_identical": bool(torch.equal(frames["default"], frames["defer_grad"])),
                  "sample_distance_ms_includes": "default: k_guided_sample_distance; defer_grad: k_guided_sample_distance_nograd + k_guided_range_grad", **out}))
