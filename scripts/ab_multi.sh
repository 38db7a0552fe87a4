# usage: bash scripts/ab_multi.sh "" "A=1" "A=1 B=2" ...  -- interleaved runs of several environment settings inside ONE box
# (3 rounds; each line: setting, slices/s, ms/step, dominant-conv TF)
for i in 1 2 3; do
  for S in "$@"; do
    env $S python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h={r['key']: r['avg_ms'] for r in (d['roofline'].get('hbm') or [])}; print('%-28s' % ('[$S]'), d['value'], d['ms_per_step'], d['roofline']['achieved'], h)"
  done
done
