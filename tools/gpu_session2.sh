#!/bin/bash
# round-2 GPU session 2: persistent kernels parity + speed, VALU microbench per entry, instruction-mix counters
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/gpu_parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/gpu_parity.log
# microbench: one entry per process, so a bad one cannot take the table down
: > $O/valu_entries.txt
for i in $(seq 0 38); do
  timeout -k 5 25 tools/valu_issue_bench $i 1 >> $O/valu_entries.txt 2>> $O/valu_entries.err; rc=$?
  if [ $rc -ne 0 ]; then echo "entry $i rc=$rc" | tee -a $O/valu_entries.err; fi
done
echo "valu entries done: $(wc -l < $O/valu_entries.txt) lines"
# speed: old one-ray-per-lane kernels vs the persistent march
for cfg in C3 C2; do
  if [ $cfg = C3 ]; then SZ="--width 240 --height 136 --spp 4"; else SZ="--width 960 --height 540 --spp 8"; fi
  GPIS_PERSIST=0 timeout -k 10 300 python bench.py --config $cfg --guide off $SZ --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_${cfg}_lane.json 2> $O/bench_${cfg}_lane.err; echo "bench $cfg lane rc=$?"
  timeout -k 10 300 python bench.py --config $cfg --guide off $SZ --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_${cfg}_persist.json 2> $O/bench_${cfg}_persist.err; echo "bench $cfg persist rc=$?"
  python - <<PY
import json
for k in ("lane", "persist"):
    try:
        r = json.load(open("$O/bench_${cfg}_%s.json" % k))
        print("$cfg", k, "%.3f Msamples/s" % r["value"], "evals/s %.3e" % r["roofline"]["evals_per_s"], r["roofline"]["kernel_ms"])
    except Exception as e:
        print("$cfg", k, "failed", e)
PY
done
# instruction mix + lane utilisation of the persistent C3 kernel
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $O/pmc_C3_p1 -o pmc -- python3 bench.py --config C3 --guide off --width 240 --height 136 --spp 4 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_C3_p1.log 2>&1; echo "pmc C3 p1 rc=$?"
ls $O
