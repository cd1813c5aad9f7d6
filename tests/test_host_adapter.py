"""The C++ `Medium`-shaped adapter (sparse-conv-gpis-tungsten_amd/host): builds with g++, parses the
reference's JSON keys with the reference's error behaviour (CPU), and — on the GPU box — behaves
like GaussianProcessMedium::sampleDistance / transmittance as seen from PathTracer.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "host")


def _build():
    import __graft_entry__ as g
    g.build_hip()
    g.build_host_adapter()
    return os.path.join(HOST, "host_adapter_test")


def test_adapter_parses_reference_json_and_fails_like_it():
    exe = _build()
    out = subprocess.run([exe, "parse"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "parse: ok" in out.stdout


@pytest.mark.gpu
def test_adapter_on_gpu():
    exe = _build()
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu: ok" in out.stdout


def test_multi_gpu_driver_builds_and_rejects_bad_arguments():
    """host/gpis_multi_gpu.cpp (the C++ form of the multi-GPU tile driver: forked ranks, RCCL broadcast + gather of tile rows)
    links against librccl / libamdhip64 / libgpis_hip here; argument errors are reported before anything touches a GPU."""
    _build()
    exe = os.path.join(HOST, "gpis_multi_gpu")
    assert os.path.exists(exe)
    out = subprocess.run([exe, "--config", "C9"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "bad arguments" in out.stderr


@pytest.mark.gpu
def test_multi_gpu_driver_single_rank_frame_equals_the_python_driver(pkg):
    """One rank through the C++ driver (RCCL communicator of size 1: init, broadcast, all-reduce barrier, all-gather run for real)
    renders the same frame as gpis_render_scene_s called from Python: equal radiance sums.  N > 1 needs an N-GPU node."""
    import ctypes
    import json
    import numpy as np
    import torch
    _build()
    exe = os.path.join(HOST, "gpis_multi_gpu")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "--gpus", "1", "--width", "480", "--height", "270", "--spp", "8", "--steps", "1", "--warmup", "1", "--guide", "16:32"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["per_rank"]) == 1
    med = pkg.Medium(pkg.params_for_config("C1"))
    med.build_guide(16, 32)
    scene = np.array(pkg.default_scene_s(480, 270, 8), dtype=pkg.SCENE_S)
    rad = torch.zeros(480 * 270, dtype=torch.float32, device="cuda")
    med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert float(rad.cpu().numpy().astype(np.float64).sum()) == pytest.approx(line["radiance_sum"], rel=1e-10)      # the same pixels, summed in another order
