#!/bin/bash
# round-2 GPU session 4: tests, microbench v2 + class calibration, C1 (split / no split / range64), C3 variants, host path, C1 counters
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 5 300 tools/valu_issue_bench > $O/valu_issue.json 2> $O/valu_issue.err; echo "valu bench rc=$?"
CL1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
CL2="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU"
CL3="SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CL1 -d $O/cal1 -o pmc -- tools/valu_issue_bench > $O/cal1.log 2>&1; echo "cal1 rc=$?"
L=$PWD/sparse-conv-gpis-tungsten_amd/csrc
# C1 headline: two-way split (default build), no split, and the 64-ray-range variant (gradient split only)
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 rc=$?"
GPIS_LIBRARY=$L/libgpis_hip_nosplit.so timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1_nosplit.json 2> $O/bench_C1_nosplit.err; echo "bench C1 nosplit rc=$?"
GPIS_RANGE_LEN=64 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_C1_range64.json 2> $O/bench_C1_range64.err; echo "bench C1 range64 rc=$?"
# C3 at size: default build, occupancy variants
SZ="--width 480 --height 270 --spp 8"
for v in default occ4 occ2; do
  if [ $v = default ]; then unset GPIS_LIBRARY; else export GPIS_LIBRARY=$L/libgpis_hip_$v.so; fi
  timeout -k 10 400 python bench.py --config C3 --guide off $SZ --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_C3_$v.json 2> $O/bench_C3_$v.err; echo "bench C3 $v rc=$?"
done
unset GPIS_LIBRARY
GPIS_SOLO_MAX=0 timeout -k 10 400 python bench.py --config C3 --guide off $SZ --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_C3_nosolo.json 2> $O/bench_C3_nosolo.err; echo "bench C3 nosolo rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C*.json")):
    try:
        r = json.load(open(f)); print(f, "%.3f Msamples/s" % r["value"], "evals/s %.3e" % r["roofline"]["evals_per_s"], r["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
timeout -k 10 400 python tools/host_path_bench.py > $O/host_path.json 2> $O/host_path.err; echo "host path rc=$?"; tail -3 $O/host_path.err
# class counters of the C1 frame (two launches of each march kernel per run: the set-up render and the timed one)
i=0
for C in "$CL1" "$CL2" "$CL3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/c1_pmc$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/c1_pmc$i.log 2>&1; echo "c1 pmc$i rc=$?"
done
ls $O
