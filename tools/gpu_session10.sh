#!/bin/bash
# round-2 final measurement session: bench lines, kernel-trace stats, class counters (C1 guided, C3 / C2 persistent), host path.
# The issue models are built ON THE BOX from these counters and the ISA of the library that ran (tools/make_issue_model.py),
# so that the sha256 recorded in them is the profiled library's.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
sha256sum sparse-conv-gpis-tungsten_amd/csrc/libgpis_hip.so > $O/lib.sha256
ISA=/tmp/gpis_hip_final.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I sparse-conv-gpis-tungsten_amd/csrc -S --cuda-device-only -o $ISA sparse-conv-gpis-tungsten_amd/csrc/gpis_hip.hip 2> /dev/null; echo "isa rc=$?"
CL1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
CL2="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU"
CL3="SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
PART=${1:-a}
if [ $PART = a ]; then
# ---- C1 (headline): counters first, model, then the bench lines that read it
i=0
for C in "$CL1" "$CL2" "$CL3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/c1_pmc$i -o pmc -- python3 bench.py --no-cpu-baseline --no-unguided --steps 1 --warmup 0 > $O/c1_pmc$i.log 2>&1; echo "c1 pmc$i rc=$?"
done
python tools/make_issue_model.py --workload "C1 1920x1080x64 guide 16:64 n_gpus 1" --collected "rocprofv3 --kernel-trace --pmc <class counters | FETCH_SIZE | WRITE_SIZE> -- python3 bench.py --no-cpu-baseline --no-unguided --steps 1 --warmup 0 (tools/gpu_session10.sh)" \
   --isa $ISA --out profiles/r02_issue_model_C1.json $O/c1_pmc*/pmc_results.db > $O/model_C1.log 2>&1; echo "model C1 rc=$?"; tail -3 $O/model_C1.log
timeout -k 10 500 python bench.py > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 (default command) rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python3 bench.py --no-cpu-baseline --no-unguided --steps 3 --warmup 1 > $O/bench_C1_under_rocprof.json 2> $O/prof_c1.log; echo "rocprof C1 rc=$?"
cp profiles/r02_issue_model_C1.json $O/
else
# ---- C3 at its own density (persistent refilling march), 480x270x8
SZ3="--config C3 --guide off --width 480 --height 270 --spp 8"
i=0
for C in "$CL1" "$CL2" "$CL3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/c3_pmc$i -o pmc -- python3 bench.py $SZ3 --no-cpu-baseline --steps 1 --warmup 0 > $O/c3_pmc$i.log 2>&1; echo "c3 pmc$i rc=$?"
done
python tools/make_issue_model.py --workload "C3 480x270x8 guide off n_gpus 1" --collected "as C1, bench.py $SZ3 (tools/gpu_session10.sh)" \
   --isa $ISA --out profiles/r02_issue_model_C3.json $O/c3_pmc*/pmc_results.db > $O/model_C3.log 2>&1; echo "model C3 rc=$?"; tail -3 $O/model_C3.log
# ---- C2 (1D sampling, persistent march), 1920x1080x16
SZ2="--config C2 --guide off --width 1920 --height 1080 --spp 16"
i=0
for C in "$CL1" "$CL2" "$CL3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/c2_pmc$i -o pmc -- python3 bench.py $SZ2 --no-cpu-baseline --steps 1 --warmup 0 > $O/c2_pmc$i.log 2>&1; echo "c2 pmc$i rc=$?"
done
python tools/make_issue_model.py --workload "C2 1920x1080x16 guide off n_gpus 1" --collected "as C1, bench.py $SZ2 (tools/gpu_session10.sh)" \
   --isa $ISA --out profiles/r02_issue_model_C2.json $O/c2_pmc*/pmc_results.db > $O/model_C2.log 2>&1; echo "model C2 rc=$?"; tail -3 $O/model_C2.log
cp profiles/r02_issue_model_C*.json $O/
# ---- bench lines (now with the roofline block filled) and the kernel-trace summaries of the same commands
timeout -k 10 300 python bench.py $SZ3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench C3 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 bench.py $SZ3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C3_under_rocprof.json 2> $O/prof_c3.log; echo "rocprof C3 rc=$?"
timeout -k 10 300 python bench.py $SZ2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C2.json 2> $O/bench_C2.err; echo "bench C2 rc=$?"
timeout -k 10 300 python bench.py --config C0 --width 256 --height 256 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C0.json 2> $O/bench_C0.err; echo "bench C0 rc=$?"
timeout -k 10 300 python tools/host_path_bench.py > $O/host_path.json 2> $O/host_path.err; echo "host path rc=$?"
timeout -k 10 300 python tools/fs_bench.py > $O/fs_bench.json 2> $O/fs_bench.err; echo "fs bench rc=$?"
fi
find $O -name "*stats*.csv" -o -name "*kernel_trace.csv" | head; du -sh $O
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_C*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1]); ro = r["roofline"]
        print(f, "%.3f Msamples/s" % r["value"], "cold", r.get("value_cold"), "unguided", r.get("value_unguided"), "frac", ro.get("frac"), ro.get("frac_range"), "stale", ro.get("counters_stale"), ro.get("kernel_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
