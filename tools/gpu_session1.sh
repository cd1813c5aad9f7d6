#!/bin/bash
# round-2 GPU session 1: parity suite, VALU issue microbenchmark, counter list, first PMC passes on the lane-per-ray kernels
set -o pipefail
mkdir -p gpurun_out/s1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/s1/build.log 2>&1 || { tail -20 gpurun_out/s1/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1/gpu_tests.log
tail -5 gpurun_out/s1/gpu_tests.log
timeout -k 10 300 tools/valu_issue_bench > gpurun_out/s1/valu_issue.json 2> gpurun_out/s1/valu_issue.err; echo "valu bench rc=$?"
rocprofv3 -L > gpurun_out/s1/counters_list.txt 2>&1; echo "list rc=$?"
for cfg in C3 C2; do
  if [ $cfg = C3 ]; then SZ="--width 240 --height 136 --spp 4"; else SZ="--width 960 --height 540 --spp 8"; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM -d gpurun_out/s1/pmc_${cfg}_a -o pmc -- python3 bench.py --config $cfg --guide off $SZ --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/s1/pmc_${cfg}_a.log 2>&1; echo "pmc $cfg a rc=$?"
done
ls -R gpurun_out/s1 | head -40
