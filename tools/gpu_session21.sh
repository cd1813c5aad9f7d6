#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s21; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o c1 -- python3 bench.py --no-cpu-baseline --no-unguided --steps 1 --warmup 0 > $O/bench.json 2> $O/prof.log; echo "rocprof rc=$?"
db=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py kernels $db | head -6
