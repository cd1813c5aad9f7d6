#!/bin/bash
# bench of library variants (no rebuild of the in-tree library)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s25; mkdir -p $O
export TMPDIR=/tmp
for R in 1 2; do for L in build/variants/libgpis_*.so; do
  GPIS_LIBRARY=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err || { echo "bench $L failed"; tail -3 $O/bench.err; continue; }
  python - <<PY
import json
r = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("$L", "%.1f" % r["value"], r["roofline"]["kernel_ms"])
PY
done; done
