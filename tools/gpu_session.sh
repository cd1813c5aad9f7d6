#!/bin/bash
# One parametrised GPU-box session (replaces the per-session scripts of rounds 1-2):
#   tools/gpu_session.sh <out-subdir> <step> [<step> ...]
# steps: build | tests[:<pytest -k expr>] | alltests[:<-k expr>] (no -x: every failure is listed) | file:<tests/file.py>[:<-k expr>] | smoke | bench[:<extra bench.py args>] | py:<script + args> | prof:<tag>:<script + args> | pmc:<tag>:<script + args>
# Every step logs into gpurun_out/<out-subdir>/ and a failing step ends the session (no GPU step runs after a failure).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/$1; shift
mkdir -p "$O"; export TMPDIR=/tmp
i=0
for step in "$@"; do
    i=$((i + 1)); kind=${step%%:*}; arg=${step#*:}; [ "$kind" = "$step" ] && arg=""
    case $kind in
    build) python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; } ;;
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q ${arg:+-k "$arg"} > $O/tests_$i.log 2>&1; rc=$?; tail -5 $O/tests_$i.log; [ $rc = 0 ] || exit $rc ;;
    alltests) timeout -k 10 1100 python -m pytest tests -m gpu -q ${arg:+-k "$arg"} > $O/tests_$i.log 2>&1; rc=$?; tail -25 $O/tests_$i.log; [ $rc = 0 ] || exit $rc ;;
    file) f=${arg%%:*}; k=${arg#*:}; [ "$f" = "$arg" ] && k=""
          timeout -k 10 1100 python -m pytest $f -m gpu -x -q -s ${k:+-k "$k"} > $O/file_$i.log 2>&1; rc=$?; tail -8 $O/file_$i.log; [ $rc = 0 ] || exit $rc ;;
    smoke) timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1 ;;
    bench) timeout -k 10 900 python bench.py $arg > $O/bench_$i.json 2> $O/bench_$i.err; rc=$?; cut -c1-600 $O/bench_$i.json; [ $rc = 0 ] || { tail -5 $O/bench_$i.err; exit $rc; } ;;
    prof) tag=${arg%%:*}; cmd=${arg#*:}       # prof:<tag>:<script + args> — rocprofv3 --kernel-trace --stats, per-kernel table -> <tag>_kernel_stats.csv
          timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/${tag}_kt -o kt -- python3 $cmd > $O/${tag}_kt.log 2>&1; rc=$?
          db=$(find /tmp/${tag}_kt -name "*results.db" | head -1)
          [ -n "$db" ] && python tools/rocpd_summary.py kernels "$db" > $O/${tag}_kernel_stats.csv && head -8 $O/${tag}_kernel_stats.csv
          [ $rc = 0 ] || { tail -5 $O/${tag}_kt.log; exit $rc; } ;;
    pmc) tag=${arg%%:*}; cmd=${arg#*:}        # pmc:<tag>:<script + args> — five counter passes (own runs, --kernel-trace only beside --pmc) -> <tag>_pmc<k>.db
          k=0
          for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
                   "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU" \
                   "SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
                   "FETCH_SIZE" "WRITE_SIZE"; do
              k=$((k + 1))
              case " ${PMC_ONLY:-1 2 3 4 5} " in *" $k "*) ;; *) continue ;; esac      # PMC_ONLY="1 2 3": a long workload split over two calls
              timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/${tag}_pmc$k -o pmc -- python3 $cmd > $O/${tag}_pmc$k.log 2>&1; rc=$?; echo "$tag pmc$k rc=$rc"
              [ $rc = 0 ] || { tail -5 $O/${tag}_pmc$k.log; exit $rc; }
              find /tmp/${tag}_pmc$k -name "*results.db" -exec cp {} $O/${tag}_pmc$k.db \;
          done ;;
    py) timeout -k 10 1100 python $arg > $O/py_$i.log 2>&1; rc=$?; tail -15 $O/py_$i.log; [ $rc = 0 ] || exit $rc ;;
    *) echo "unknown step $step"; exit 2 ;;
    esac
done
