#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + PMC passes of the generic (explicit-queue) step kernel's bench.
# Usage: bash tools/profile_generic.sh <tag> [script]   (script defaults to tools/bench_generic.py)
set -e
TAG=${1:-r2_generic}
SCRIPT=${2:-tools/bench_generic.py}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GW_BENCH_MODES=${GW_BENCH_MODES:-plain}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/$SCRIPT > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/$SCRIPT > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/$SCRIPT > $OUT/bench_write.json 2> $OUT/write.err
echo "traffic done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $REPO/$SCRIPT > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/$SCRIPT > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_mem -- python3 $REPO/$SCRIPT > $OUT/bench_mem.json 2> $OUT/mem.err || echo "mem failed"
echo "sq done"
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
