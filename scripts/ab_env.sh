# usage: bash scripts/ab_env.sh VAR VALUE  -- interleaved A/B of VAR=VALUE against the default inside one box
V=$1; X=$2
for i in 1 2 3; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  env $V=$X python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$X', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
