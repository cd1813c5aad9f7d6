#!/bin/bash
# k-way candidate split: bit-exact parity of the march paths, guide tests, bench, work counters
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s24; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_guide.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_kernel_resources.py -m gpu -x -q > $O/gpu_tests_a.log 2>&1; rc=$?; echo "gpu tests a rc=$rc"; tail -4 $O/gpu_tests_a.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "march or incoherent or render_scene or dense" > $O/gpu_tests_b.log 2>&1; rc=$?; echo "gpu tests b rc=$rc"; tail -4 $O/gpu_tests_b.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench rc=$?"; python - <<PY
import json
r = json.loads(open("$O/bench_C1.json").read().strip().splitlines()[-1])
print(r["value"], r["value_cold"], r["roofline"]["kernel_ms"], r["roofline"]["algorithmic"]["n_eval"])
PY
