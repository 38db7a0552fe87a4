# usage: bash scripts/ab.sh VAR  -- interleaved A/B of an environment switch inside ONE box (devices differ by several %)
V=$1
for i in 1 2 3; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default   ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  env $V=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
