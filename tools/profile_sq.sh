#!/bin/bash
# SQ issue/stall counters of the step kernel (own --pmc passes, no tracing domains besides the kernel trace)
set -e
D=${1:-4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_d$D
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 128 --warmup 32 --no-cpu-baseline --no-rollout --no-graph --no-steady --no-graph --devices $D"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > $OUT/b.json 2> $OUT/b.err
python3 $REPO/tools/summarize_prof.py $OUT | grep -E "ct_step_sfx" | cut -c1-170
