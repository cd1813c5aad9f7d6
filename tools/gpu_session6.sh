#!/bin/bash
# round-2 GPU session 6: GPU suite with the Matern / Gabor kernels, C1 and C3 bench lines as regression check
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matern or aniso_field or variance_field or absorption or persistent or c3_at or multi_resolution" > $O/gpu_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/gpu_new.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 rc=$?"
timeout -k 10 300 python bench.py --config C3 --guide off --width 480 --height 270 --spp 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench C3 rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C*.json")):
    try:
        r = json.load(open(f)); print(f, "%.3f Msamples/s" % r["value"], "cold", r.get("value_cold"), "unguided", r.get("value_unguided"), r["roofline"].get("kernel_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
ls $O
