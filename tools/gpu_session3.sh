#!/bin/bash
# round-2 GPU session 3: full GPU suite, microbench (+ counter-class calibration), C1 headline with/without in-wave refill, C2/C3 at size
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 5 200 tools/valu_issue_bench > $O/valu_issue.json 2> $O/valu_issue.err; echo "valu bench rc=$?"
CL1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
CL2="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CL1 -d $O/cal1 -o pmc -- tools/valu_issue_bench > $O/cal1.log 2>&1; echo "cal1 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CL2 -d $O/cal2 -o pmc -- tools/valu_issue_bench > $O/cal2.log 2>&1; echo "cal2 rc=$?"
# headline: C1 full frame, in-wave refill vs the round-1 resident kernels
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1_range.json 2> $O/bench_C1_range.err; echo "bench C1 range rc=$?"
GPIS_RANGE_LEN=0 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1_lane.json 2> $O/bench_C1_lane.err; echo "bench C1 lane rc=$?"
for L in 256 4096; do
  GPIS_RANGE_LEN=$L timeout -k 10 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_C1_range$L.json 2> $O/bench_C1_range$L.err; echo "bench C1 range$L rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C1_*.json")):
    try:
        r = json.load(open(f)); print(f, "%.1f Msamples/s" % r["value"], r["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
# per-path media at a size that fills the chip
for cfg in C3 C2; do
  if [ $cfg = C3 ]; then SZ="--width 480 --height 270 --spp 8"; else SZ="--width 1920 --height 1080 --spp 8"; fi
  timeout -k 10 400 python bench.py --config $cfg --guide off $SZ --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_${cfg}_persist.json 2> $O/bench_${cfg}_persist.err; echo "bench $cfg persist rc=$?"
  GPIS_PERSIST=0 timeout -k 10 600 python bench.py --config $cfg --guide off $SZ --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_${cfg}_lane.json 2> $O/bench_${cfg}_lane.err; echo "bench $cfg lane rc=$?"
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
ls $O
