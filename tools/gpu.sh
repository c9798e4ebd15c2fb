#!/bin/bash
# tools/gpu.sh -- the GPU-box recipes, meant for `gpurun -- 'bash tools/gpu.sh <cmd> ...'`.
# Everything is written under gpurun_out/; summaries worth keeping are copied to profiles/.
#
#   tests                       pytest -m gpu, then __graft_entry__.smoke()
#   bench [args]                python bench.py [args]            -> gpurun_out/bench.json
#   stats TAG SCRIPT [args]     rocprofv3 --kernel-trace --stats of `python3 SCRIPT args`
#                               -> gpurun_out/TAG_kernel_stats.csv + TAG_kernels.txt
#   traffic TAG SCRIPT [args]   FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#                               (MI355X_MICROARCH.md, HBM) -> gpurun_out/TAG_traffic.json
#   pmc TAG "C1 C2 .." SCRIPT [args]   one --pmc pass of the listed counters -> TAG_pmc.txt
# rocprofv3 runs the program itself after `--` (python3 <script>), never a shell or env hop.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cmd=$1; shift
# rocprofv3 runs from /tmp: make the script path of `stats / traffic / pmc` absolute
abs() { case $1 in /*) echo $1 ;; *) echo $R/$1 ;; esac; }
case $cmd in
tests)
    cd $R && python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
    tail -5 $OUT/tests.log
    [ $rc -eq 0 ] && python __graft_entry__.py smoke 2>&1 | tail -1
    exit $rc ;;
bench)
    cd $R && python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; rc=$?
    tail -c 400 $OUT/bench.err; head -c 600 $OUT/bench.json; echo
    exit $rc ;;
stats)
    tag=$1; script=$(abs $2); shift 2
    cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -- python3 $script "$@" > $OUT/${tag}_run.log 2>&1
    rc=$?
    cd $R && python tools/kernel_stats.py $tag | head -40
    exit $rc ;;
traffic)
    tag=$1; script=$(abs $2); shift 2
    cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/pmc_fetch $OUT/pmc_write
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $script "$@" > $OUT/pmc_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $script "$@" > $OUT/pmc_write.log 2>&1
    rc=$?
    cd $R && python tools/pmc_traffic_report.py $tag
    exit $rc ;;
pmc)
    tag=$1; counters=$2; script=$(abs $3); shift 3
    cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/pmc_$tag
    rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/pmc_$tag -- python3 $script "$@" > $OUT/pmc_$tag.log 2>&1
    rc=$?
    cd $R && python tools/kernel_stats.py --pmc $tag | tee $OUT/${tag}_pmc.txt
    exit $rc ;;
*)
    echo "unknown command: $cmd"; exit 2 ;;
esac
