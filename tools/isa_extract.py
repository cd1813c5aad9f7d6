"""Keeps only the named kernels of an ISA listing (hipcc -S --cuda-device-only): the full listing of gpis_hip.hip is 70 MB, the four
kernels the issue models need are 3 MB, which travels to the GPU box with the repository snapshot.
usage: python tools/isa_extract.py gpis_hip.s out.s <mangled-name substring> [...]"""
import re
import sys

src, dst, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
label = re.compile(r"^(_Z\w+):")
keep, n = False, 0
with open(src) as f, open(dst, "w") as g:
    for line in f:
        m = label.match(line)
        if m:
            keep = any(s in m.group(1) for s in subs)
            n += keep
        if keep:
            g.write(line)
            if line.startswith(".Lfunc_end"):
                keep = False
print("kept %d functions" % n)
