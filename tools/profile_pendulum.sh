#!/bin/bash
# Run ON THE GPU BOX: kernel trace + PMC passes of `bench.py --config 4` (pendulum env, one launch per env.step()).
set -e
TAG=${1:-r2_config4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config 4 --steps 128 --warmup 64 --repeats 6 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err || echo "sq failed"
echo "pmc done"
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
