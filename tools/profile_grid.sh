#!/bin/bash
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NO_CPU=1 python3 $REPO/tools/bench_grid.py > $OUT/bench_grid.json 2>$OUT/bench.err
NO_CPU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_grid.py > $OUT/bench_trace.json 2> $OUT/trace.err
NO_CPU=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/bench_grid.py > $OUT/bench_pmc.json 2> $OUT/pmc.err || echo "pmc failed"
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "grid_run|^#" | cut -c1-200
