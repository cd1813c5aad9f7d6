#!/bin/bash
# round-2 final measurements, part A: whole GPU suite, then the counter passes of the final library (C1 guided, C3 / C2 persistent).
# The issue models are fitted from these result files with tools/make_issue_model.py and the ISA of the same sources; part B
# (tools/gpu_session18.sh) then prints the bench lines that read them.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s17; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
sha256sum sparse-conv-gpis-tungsten_amd/csrc/libgpis_hip.so > $O/lib.sha256
if [ "${1:-all}" != pmc ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $O/gpu_tests.log
[ $rc = 0 ] || exit $rc
fi
CL1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
CL2="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU"
CL3="SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
SZ3="--config C3 --guide off --width 480 --height 270 --spp 8"
SZ2="--config C2 --guide off --width 1920 --height 1080 --spp 16"
run_pmc() {   # tag, bench args...
  local tag=$1; shift
  local i=0
  for C in "$CL1" "$CL2" "$CL3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/${tag}_pmc$i -o pmc -- python3 bench.py "$@" --no-cpu-baseline --no-unguided --steps 1 --warmup 0 > $O/${tag}_pmc$i.log 2>&1; echo "$tag pmc$i rc=$?"
    find /tmp/${tag}_pmc$i -name "*results.db" -exec cp {} $O/${tag}_pmc$i.db \;
  done
}
run_pmc c1
run_pmc c3 $SZ3
run_pmc c2 $SZ2
du -sh $O; ls -la $O/*.db
