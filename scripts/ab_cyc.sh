# interleaved A/B of an environment switch on the CycleGan step (B=8, 512^2, bf16) inside one box
V=$1
for i in 1 2 3; do
  python bench.py --workload cyc --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default   ', d['value'], d['ms_per_step'])"
  env $V=1 python bench.py --workload cyc --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=1', d['value'], d['ms_per_step'])"
done
