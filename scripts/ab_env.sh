#!/bin/bash
# ab_env.sh DTYPE "" "A=1" "A=1 B=2" ...: interleaved runs of several environment settings of the Hd step in ONE box
# (bench.py --dtype DTYPE, $STEPS steps): ms/step per setting and round.  ROUNDS env var = number of interleaved rounds (default 2).
DT=$1; shift
ROUNDS=${ROUNDS:-2}
STEPS=${STEPS:-10}
for r in $(seq 1 $ROUNDS); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --dtype $DT --steps $STEPS --warmup 3 --no-cpu-baseline --no-kernel-events --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %.1f slices/s' % (d['ms_per_step'], d['value']))")
    echo "$DT round $r [${cfg:-default}] $ms"
  done
done
