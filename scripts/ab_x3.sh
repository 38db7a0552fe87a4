#!/bin/bash
# ab_x3.sh "" "A=1" "A=1 B=2" ... [ROUNDS=2]: interleaved runs of several environment settings of the bf16x3 Hd step in ONE box
# (bench.py --dtype bf16x3, 10 steps): ms/step per setting and round.  ROUNDS env var = number of interleaved rounds (default 2).
ROUNDS=${ROUNDS:-2}
STEPS=${STEPS:-10}
for r in $(seq 1 $ROUNDS); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --dtype bf16x3 --steps $STEPS --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %.1f slices/s' % (d['ms_per_step'], d['value']))")
    echo "round $r [${cfg:-default}] $ms"
  done
done
