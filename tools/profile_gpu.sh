#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + PMC passes of the bench command.
# Usage: bash tools/profile_gpu.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/...; copy the summaries you want judged into profiles/.
set -e
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 128 --warmup 64 --repeats 6 --no-cpu-baseline --no-graph --no-steady $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
echo "sq done"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
echo "sq2 done"
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
