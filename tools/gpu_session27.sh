#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s27; export TMPDIR=/tmp
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 2 --warmup 1 2> gpurun_out/s27/bench.err | tee gpurun_out/s27/bench.json | cut -c1-300
