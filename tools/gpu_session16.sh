#!/bin/bash
# candidate-list sideways evaluator: guide / golden / fuzz / full-size tests, bench, work counters
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s16; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_guide.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_kernel_resources.py -m gpu -x -q > $O/gpu_tests_a.log 2>&1; rc=$?; echo "gpu tests a rc=$rc"; tail -6 $O/gpu_tests_a.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench rc=$?"; python - <<PY
import json
r = json.loads(open("$O/bench_C1.json").read().strip().splitlines()[-1])
print(r["value"], r["value_cold"], r["roofline"]["kernel_ms"])
PY
for V in a1 a5 sm6; do
  GPIS_LIBRARY=build/variants/libgpis_$V.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_$V.json 2> $O/bench_$V.err; echo "bench $V rc=$?"
  python - <<PY
import json
r = json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1])
print("$V", r["value"], r["roofline"]["kernel_ms"])
PY
done
timeout -k 10 300 python tools/fast_stats.py 1920 1080 64 C1 16:64 > $O/fast_stats.log 2>&1; echo "stats rc=$?"; tail -9 $O/fast_stats.log
