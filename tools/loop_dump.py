"""Prints the instructions of one loop (innermost-loop attribution, as tools/issue_model.py groups them) of a kernel in an ISA listing.
usage: python tools/loop_dump.py gpis_hip.s <mangled-name substring> <.LBBn_m | top> [--all]"""
import re
import sys

path, ksub, want = sys.argv[1:4]
label_re = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
cur, pending, inside = "top", None, False
for line in open(path):
    if not inside:
        m = label_re.match(line)
        if m and m.group(1).startswith("_Z") and ksub in m.group(1):
            inside = True
        continue
    if line.lstrip().startswith(".end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        break
    m = label_re.match(line)
    if m:
        pending = m.group(1)
        h = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", line)
        if "Loop Header" in line:
            cur = pending
        elif h:
            cur = ".L" + h.group(1)
        if cur == want:
            print(line.rstrip())
        continue
    if re.search(r";\s+(?:=>)?\s*This (?:Inner )?Loop Header: Depth=\d+", line):
        cur = pending
        continue
    h = re.search(r";\s+in Loop: Header=(\S+) Depth=\d+", line)
    if h:
        cur = ".L" + h.group(1)
        continue
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur == want:
        print("    " + t)
