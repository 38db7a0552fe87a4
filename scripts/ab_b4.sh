for r in 1 2; do
for cfg in "" "CTG_TH8_WGS=600" "CTG_TH8_WGS=1100" "CTG_NIE_SHARE=1" "CTG_NIE_SHARE=4" "CTG_NO_SIDE_STREAM=1"; do
  ms=$(env $cfg python bench.py --batch 4 --steps 30 --warmup 4 --no-cpu-baseline --no-kernel-events --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %.1f slices/s' % (d['ms_per_step'], d['value']))")
  echo "B=4 round $r [${cfg:-default}] $ms"
done; done
