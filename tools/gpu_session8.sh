#!/bin/bash
# round-2 GPU session 8: full GPU suite on the final library + C4 function-space rate
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s8; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python tools/fs_bench.py > $O/fs_bench.json 2> $O/fs_bench.err; echo "fs bench rc=$?"; tail -3 $O/fs_bench.err; cat $O/fs_bench.json
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
