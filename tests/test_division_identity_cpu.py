"""The fma-corrected division by an invariant divisor used in the 1D lattice sum equals IEEE division (CPU check of the identity the
HIP code relies on; the GPU parity tests check the kernels themselves bit for bit)."""
import os
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None or not _cpu_has_fma(), reason="needs gcc and a CPU with FMA")
def test_fma_corrected_division_is_exact():
    tmp = tempfile.mkdtemp()
    try:
        exe = os.path.join(tmp, "divcheck")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(HERE, "division_identity_check.c"), "-lm"])
        for seed in ("1", "2"):
            out = subprocess.check_output([exe, seed, "20000000"], text=True)
            assert "bad32=0 bad64=0" in out, out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
