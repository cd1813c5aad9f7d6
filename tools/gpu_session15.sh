#!/bin/bash
# packed sideways evaluator vs one cell per pass (same source otherwise), work counters of the guided kernels, whole parity suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s15; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python tools/fast_stats.py 1920 1080 16 C1 16:64 > $O/fast_stats.log 2>&1; echo "stats rc=$?"; tail -9 $O/fast_stats.log
for V in p1 p0; do
  L=""; [ $V = p0 ] && L="build/variants/libgpis_p0.so"
  GPIS_LIBRARY=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_$V.json 2> $O/bench_$V.err; echo "bench $V rc=$?"
  python - <<PY
import json
r = json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1])
print("$V", r["value"], r["roofline"]["kernel_ms"])
PY
done
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guide.py -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $O/gpu_tests.log
