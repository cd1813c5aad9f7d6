#!/bin/bash
# round-2 GPU session 9: occupancy variants of the guided kernels on the final sources
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in base gocc3 gocc5 gtr4 gtr6; do
  if [ $v = base ]; then unset GPIS_LIBRARY; else export GPIS_LIBRARY=$PWD/sparse-conv-gpis-tungsten_amd/csrc/libgpis_hip_$v.so; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unguided > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r = json.load(open(f)); print(f, "%.1f Msamples/s" % r["value"], r["roofline"].get("kernel_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
