#!/bin/bash
# round-2 final measurements, part B: the bench lines (reading the issue models fitted from part A's counters), the kernel-trace
# summaries of the same commands, the host path, the multi-bounce driver.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s18; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
sha256sum sparse-conv-gpis-tungsten_amd/csrc/libgpis_hip.so > $O/lib.sha256
SZ3="--config C3 --guide off --width 480 --height 270 --spp 8"
SZ2="--config C2 --guide off --width 1920 --height 1080 --spp 16"
timeout -k 10 500 python bench.py > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 (default command) rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python3 bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_C1_under_rocprof.json 2> $O/prof_c1.log; echo "rocprof C1 rc=$?"
timeout -k 10 300 python bench.py $SZ3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench C3 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 bench.py $SZ3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C3_under_rocprof.json 2> $O/prof_c3.log; echo "rocprof C3 rc=$?"
timeout -k 10 300 python bench.py $SZ2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C2.json 2> $O/bench_C2.err; echo "bench C2 rc=$?"
timeout -k 10 300 python bench.py --config C0 --width 256 --height 256 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C0.json 2> $O/bench_C0.err; echo "bench C0 rc=$?"
timeout -k 10 300 python tools/host_path_bench.py > $O/host_path.json 2> $O/host_path.err; echo "host path rc=$?"
timeout -k 10 300 python tools/paths_bench.py > $O/paths_bench.json 2> $O/paths_bench.err; echo "paths rc=$?"; tail -1 $O/paths_bench.json
for V in occ6 occ4 sm2 sm5; do
  [ -f build/variants/libgpis_$V.so ] || continue
  GPIS_LIBRARY=build/variants/libgpis_$V.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_v_$V.json 2> $O/bench_v_$V.err; echo "variant $V rc=$?"
  python - <<PY
import json
r = json.loads(open("$O/bench_v_$V.json").read().strip().splitlines()[-1])
print("$V", r["value"], r["roofline"]["kernel_ms"])
PY
done
du -sh $O
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1]); ro = r["roofline"]
        print(f, "%.3f Msamples/s" % r["value"], "cold", r.get("value_cold"), "unguided", r.get("value_unguided"), "frac", ro.get("frac"), ro.get("frac_range"), "stale", ro.get("counters_stale"), ro.get("kernel_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
