#!/bin/bash
# division by the invariant 2 ls^2 in the 1D sum: 1D / per-path parity, golden fixtures, fuzz, C2 bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s26; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q -k "1d or persistent or per_path or nee or golden or fuzz or multi_resolution or drivers_on_other or absorption" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 $O/gpu_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --config C2 --guide off --width 1920 --height 1080 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C2.json 2> $O/bench_C2.err; echo "bench C2 rc=$?"; python - <<PY
import json
r = json.loads(open("$O/bench_C2.json").read().strip().splitlines()[-1])
print(r["value"], r["roofline"]["kernel_ms"])
PY
