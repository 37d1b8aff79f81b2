#!/bin/bash
# A/B of library builds on ONE box: the driver's bench window, alternating builds, three rounds.
# usage: bash tools/ab.sh name1 name2 ...   (names of gymwipe_amd/lib/libgymwipe_amd_<name>.so; "base" = the product library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-3}); do
  for n in "$@"; do
    if [ "$n" = base ]; then unset GW_LIB; else export GW_LIB=$REPO/gymwipe_amd/lib/libgymwipe_amd_$n.so; fi
    python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondaries --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'round $round', '%.3f G' % (d['value']/1e9), '%.2f us wall' % (d['ms_per_step']*1e3), '%.2f us kern' % d['roofline']['kernel_avg_us'])"
  done
done
