#!/bin/bash
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_plant
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/bench_plant.py > $OUT/bench_plant.json 2>$OUT/bench.err; tail -1 $OUT/bench_plant.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_plant.py > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/bench_plant.py > $OUT/bench_pmc.json 2> $OUT/pmc.err || echo "pmc pass failed (counter names?)"
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "n=|mean=" | grep -v "^void at\|rocclr" | cut -c1-220
grep -h "plant" $OUT/pmc_sq/*/*counter_collection.csv 2>/dev/null | awk -F, '{print $(NF-1), $NF}' | sort | uniq -c | sort -rn | head -3
