#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s19; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_cells" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -15 $O/gpu_tests.log
