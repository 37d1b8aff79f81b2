#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + SQ counters of the fused rollout (bench.py's secondary measurement), for
# the step-synchronous kernel and, with GW_ROLLOUT_EVENT_LOOP=1, the event loop it replaced.  Usage: bash tools/profile_rollout.sh <tag>
set -e
TAG=${1:-r3_rollout}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for form in sync event; do
    OUT=$REPO/gpurun_out/prof_${TAG}_$form
    mkdir -p $OUT
    if [ $form = event ]; then export GW_ROLLOUT_EVENT_LOOP=1; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline --repeats 8 > $OUT/bench_trace.json 2> $OUT/trace.err
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline --repeats 8 > $OUT/bench_sq.json 2> $OUT/sq.err
    python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt
    echo "== $form"; grep -E "rollout|pack_actions|expand_feedback" $OUT/SUMMARY.txt | cut -c1-220
done
