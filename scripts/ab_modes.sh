# usage: bash scripts/ab_modes.sh "VAR=1" [rounds] -- interleaved A/B of an environment switch for BOTH bench legs (bf16 and the
# bf16x3 parity mode) inside ONE box (boxes of the pool differ by several %)
V=$1
N=${2:-3}
for i in $(seq $N); do
  for e in "" "$V"; do
    env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['parity_mode']; print('%-24s bf16 %.1f slices/s %.2f ms (dominant %.4f) | bf16x3 %.1f slices/s %.2f ms (dominant %.4f)' % ('$e' or 'default', d['value'], d['ms_per_step'], d['roofline']['frac'], p['value'], p['ms_per_step'], p['roofline']['frac']))"
  done
done
