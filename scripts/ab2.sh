# usage: bash scripts/ab2.sh "VAR1=x VAR2=y"  -- interleaved A/B of several environment settings against the default
for i in 1 2 3; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  env $1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
