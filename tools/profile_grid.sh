#!/bin/bash
# Run ON THE GPU BOX: bench + kernel trace + SQ counters of the PHY grid kernel.  Usage: N=65536 SIM=0.25 bash tools/profile_grid.sh <tag>
set -e
TAG=${1:-r3_grid}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/bench_grid.py > $OUT/bench_grid.json 2>$OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_grid.py > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/bench_grid.py > $OUT/bench_pmc.json 2> $OUT/pmc.err || echo "pmc failed"
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "grid_run|^#" | cut -c1-200 > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
