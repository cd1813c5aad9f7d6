#!/bin/bash
# re-entry check: whole GPU suite, smoke, function-space rate, default bench line on the restored tree
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s13; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $O/gpu_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python tools/fs_bench.py > $O/fs_bench.json 2> $O/fs_bench.err; echo "fs bench rc=$?"; cat $O/fs_bench.json
timeout -k 10 400 python bench.py > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench rc=$?"; cat $O/bench_C1.json
