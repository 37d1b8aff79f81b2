#!/bin/bash
set -e
TAG=${1:-ro}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "n=" | grep -v "^void at\|rocclr" | cut -c1-200
