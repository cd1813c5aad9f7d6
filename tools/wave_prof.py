"""Diagnostic (GPU box): one scene-S render (C1 960x540x64, guide 16:64) for a rocprofv3 kernel trace."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _gpis_pkg, torch
pkg = _gpis_pkg.load_package()
med = pkg.Medium(pkg.params_for_config("C1"))
med.build_guide(16, 64)
scene = np.zeros((), dtype=pkg.SCENE_S)
med.L.lib.gpis_default_scene_s(scene.ctypes.data, 960, 540, 16)
rad = torch.zeros(960 * 540, dtype=torch.float32, device="cuda")
med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
torch.cuda.synchronize()
print(float(rad.sum()), med.counters(), med.guide_steps())
