"""Register / scratch / LDS budget of the hot kernels, read from the code object inside libgpis_hip.so (no GPU needed).
Round 2 lost half of the C1 rate for a while because the guided kernel's cold fallback resolved to the all-features instance of
the path section and its scratch grew from 320 to 668 B per lane: this test is the tripwire for that class of regression."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(lib):
    """kernel name -> resources, over every code object of the library (one offload bundle per translation unit)"""
    tmp = tempfile.mkdtemp()
    notes = ""
    try:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, a in enumerate(starts):
            part, co = os.path.join(tmp, "b%d.bin" % i), os.path.join(tmp, "b%d.co" % i)
            open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co])
            notes += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True, stderr=subprocess.DEVNULL)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for blk in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            out[name.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                                  for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size")}
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="LLVM tools of the ROCm image")
def test_hot_kernels_keep_their_budget(pkg):
    k = _kernels(pkg.library_path())

    def one(sub):
        hits = [v for n, v in k.items() if sub in n]
        assert hits, sub
        return hits

    for v in one("k_guided_sample_distanceILb"):          # allocated for 5 waves/SIMD: 96 VGPRs, a few hundred bytes of scratch
        assert v["vgpr_count"] <= 96 and v["private_segment_fixed_size"] <= 400, v
    for v in one("k_guided_transmittanceILb"):
        assert v["vgpr_count"] <= 96 and v["private_segment_fixed_size"] <= 360, v
    for v in one("k_fast_sample_distance"):
        assert v["private_segment_fixed_size"] == 0, v
    for sub, scratch in (("7spec_3d7PersistE", 0), ("16spec_3d_multires7PersistE", 64), ("7spec_1d7PersistE", 160)):
        for v in one("k_persist_marchINS_" + sub):
            assert v["vgpr_count"] <= 168 and v["private_segment_fixed_size"] <= scratch, (sub, v)   # 3 waves/SIMD
            # LDS comes in 1280-byte granules: 12 one-wave workgroups per CU (3 per SIMD) leave 10 granules each.  13 072 B once
            # cost every persistent launch 5-8 % while the occupancy query still said 12 ("refill").
            assert v["group_segment_fixed_size"] <= 12800, (sub, v)
    for v in one("k_fs_marchILb"):                        # four workgroups per CU (one per SIMD), no scratch
        assert v["group_segment_fixed_size"] <= 40 * 1024 and v["private_segment_fixed_size"] == 0, v
