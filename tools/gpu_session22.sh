#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s22; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/s22/build.log 2>&1 || exit 1
timeout -k 10 200 python tools/guide_build_time.py 2>&1 | tee gpurun_out/s22/time.log
