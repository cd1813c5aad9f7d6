#!/bin/bash
# round-2 GPU session 5: full GPU suite after the a23 / ABI-2 changes, host path with 256 Ki chunks, C2 / C0 bench lines
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 300 python tools/host_path_bench.py > $O/host_path.json 2> $O/host_path.err; echo "host path rc=$?"; tail -3 $O/host_path.err
timeout -k 10 300 python bench.py --config C2 --guide off --width 1920 --height 1080 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C2.json 2> $O/bench_C2.err; echo "bench C2 rc=$?"
timeout -k 10 300 python bench.py --config C0 --width 256 --height 256 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C0.json 2> $O/bench_C0.err; echo "bench C0 rc=$?"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C*.json")):
    try:
        r = json.load(open(f)); print(f, "%.3f Msamples/s" % r["value"], "cold", r.get("value_cold"), "unguided", r.get("value_unguided"), r["roofline"].get("kernel_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
ls $O
