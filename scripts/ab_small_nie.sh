for b in 1 2 4 8; do
  for e in "" "CTG_NO_NIE=1"; do
    env $e python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b %-14s %.1f slices/s %.2f ms' % ('$e' or 'default', d['value'], d['ms_per_step']))"
  done
done
