#!/bin/bash
# A/B of a library variant against the in-tree library: guide tests on the variant, then alternating bench runs
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
V=${1:-pf}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
GPIS_LIBRARY=build/variants/libgpis_$V.so timeout -k 10 600 python -m pytest tests/test_gpu_guide.py tests/test_gpu_golden.py -m gpu -x -q > $O/gpu_tests_a.log 2>&1; rc=$?; echo "variant tests rc=$rc"; tail -3 $O/gpu_tests_a.log
[ $rc = 0 ] || exit $rc
for R in 1 2; do for L in "" build/variants/libgpis_$V.so; do
  GPIS_LIBRARY=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  python - <<PY
import json
r = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("${L:-in-tree}", r["value"], r["roofline"]["kernel_ms"])
PY
done; done
