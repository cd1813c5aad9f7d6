#!/bin/bash
# guide-field build with the block-level inside class: bound self-check, certificate ray-check, guided parity, build time, evaluation counts
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s20; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_guide.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -s > $O/gpu_tests_a.log 2>&1; rc=$?; echo "gpu tests a rc=$rc"; grep -a "guide \|certified\|exact evaluations\|passed\|failed" $O/gpu_tests_a.log | tail -20
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench rc=$?"; python - <<PY
import json
r = json.loads(open("$O/bench_C1.json").read().strip().splitlines()[-1])
print(r["value"], r["value_cold"], r["value_unguided"], r["config"]["guide"], r["roofline"]["kernel_ms"], r["roofline"]["algorithmic"])
PY
timeout -k 10 300 python bench.py --config C0 --width 256 --height 256 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C0.json 2> $O/bench_C0.err; echo "bench C0 rc=$?"; tail -c 600 $O/bench_C0.json | head -c 300
