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
