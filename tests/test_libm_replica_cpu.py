"""csrc/gpis_libm.hpp restates, bit for bit, the libm functions the reference's double-precision code calls on this path as the
host's glibc evaluates them (exp, logf, log, sin, cos, pow in their FMA variants, sincos in its SSE2 one).  This test compiles the header
for the host together with tests/native/libm_replica_check.cpp and compares every function with the libm of this process on 10^7
arguments each; the device side of the same header is tests/test_gpu_libm.py."""
import os
import platform
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in (line + " ") and " avx2 " in (line + " ")
    except OSError:
        pass
    return False


@pytest.mark.skipif(platform.machine() != "x86_64" or shutil.which("g++") is None, reason="x86-64 host with g++ only")
@pytest.mark.skipif(not _cpu_has_fma(), reason="glibc selects the FMA variants only on a CPU with FMA and AVX2")
def test_replicas_equal_the_host_libm_bit_for_bit(tmp_path):
    exe = str(tmp_path / "libm_replica_check")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-I", os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "libm_replica_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "10000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "exp 0, logf 0, log 0, sin 0, cos 0, sincos 0, pow 0, sincosf 0 mismatches" in r.stdout


def test_sincos_table_is_what_the_generator_writes():
    """the committed table = tools/make_sincos_table.py's output (mpmath present) — guards against a hand edit of either"""
    mp = pytest.importorskip("mpmath")  # noqa: F841
    out = subprocess.run(["python", os.path.join(ROOT, "tools", "make_sincos_table.py")], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "csrc", "gpis_sincos_table.inc")) as f:
        assert f.read() == out
