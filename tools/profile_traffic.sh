#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the step kernel for one device count (separate --pmc passes, per the guide)
set -e
D=${1:-4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/traffic_d$D
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 128 --warmup 32 --repeats 4 --no-cpu-baseline --no-secondaries --devices $D"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/f.json 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/w.json 2> $OUT/w.err
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "ct_step.*SIZE" | cut -c1-160
