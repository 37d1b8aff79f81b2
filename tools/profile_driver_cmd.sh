#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC traffic passes of the DRIVER's exact bench command
#   python3 bench.py --gpus 1 --steps 20 --warmup 5
# Writes gpurun_out/prof_<tag>/; tools/summarize_driver_prof.py condenses it (copy the summary into profiles/).
set -e
TAG=${1:-r2_driver_cmd}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
# counters in their own passes, fewer windows (every launch is serialised by the counter collection)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS --repeats 40 --no-secondaries --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS --repeats 40 --no-secondaries --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
echo "write done"
python3 $REPO/tools/summarize_driver_prof.py $OUT $REPO/gpurun_out/plain_$TAG.json > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
