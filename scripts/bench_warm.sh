# does the CPU baseline (GPU idle for ~15 s) or a short warm-up depress the GPU measurement?  one box, interleaved
j='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"])'
for i in 1 2; do
  python bench.py 2>/dev/null | tail -1 | python -c "$j" "default(cpu,w2,s8) "
  python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "$j" "nocpu,w2,s8        "
  python bench.py --no-cpu-baseline --no-parity-mode --warmup 6 --steps 20 2>/dev/null | tail -1 | python -c "$j" "nocpu,w6,s20       "
  python bench.py --warmup 6 --steps 20 2>/dev/null | tail -1 | python -c "$j" "cpu,w6,s20         "
done
