#!/bin/bash
# One parametrised GPU-box session (replaces the per-session scripts of rounds 1-2):
#   tools/gpu_session.sh <out-subdir> <step> [<step> ...]
# steps: build | tests[:<pytest -k expr>] | alltests[:<-k expr>] (no -x: every failure is listed) | file:<tests/file.py>[:<-k expr>] | smoke | bench[:<extra bench.py args>] | py:<script + args>
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
    py) timeout -k 10 1100 python $arg > $O/py_$i.log 2>&1; rc=$?; tail -15 $O/py_$i.log; [ $rc = 0 ] || exit $rc ;;
    *) echo "unknown step $step"; exit 2 ;;
    esac
done
