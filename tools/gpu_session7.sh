#!/bin/bash
# round-2 GPU session 7: function-space path on the device
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_fs.py -m gpu -x -q > $O/gpu_fs.log 2>&1; echo "fs tests rc=$?"; tail -25 $O/gpu_fs.log
