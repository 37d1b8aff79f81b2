#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + PMC passes of the live-PHY step kernel (per-env geometry, suffix queues)
# at D devices x 65 536 envs.  Usage: bash tools/profile_live_phy.sh <tag> <D>
set -e
TAG=${1:-r3_live_phy_d4}; export D=${2:-4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GW_BENCH_MODES=suffix
S=$REPO/tools/bench_live_phy.py
python3 $S > $OUT/bench_plain.json 2> $OUT/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $S > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $S > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $S > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $S > $OUT/bench_sq.json 2> $OUT/sq.err
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt
grep -E "ct_step_live|ct_set_position" $OUT/SUMMARY.txt | cut -c1-200
cat $OUT/bench_plain.json
