# every bench.py workload / dtype / small-batch point in ONE box (numbers for DESIGN.md; boxes differ by several %)
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"] or {}; print(sys.argv[1].ljust(14), "%8.1f slices/s %8.2f ms/step" % (d["value"], d["ms_per_step"]), "  res-block convs %s TF (%s of peak)" % (r.get("achieved"), r.get("frac")), [ (k["avg_ms"]) for k in r.get("kernels", [])])'
run() { name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "$P" "$name"; }
run "hd bf16"      --steps 30 --warmup 3
run "hd bf16x3"    --steps 8 --warmup 2 --dtype bf16x3
run "hd bf16x3f"   --steps 8 --warmup 2 --dtype bf16x3f
run "hd fp32"      --steps 4 --warmup 1 --dtype fp32
run "gen fp32 B8"  --workload gen --steps 20 --warmup 3
run "gen bf16x3 B8" --workload gen --dtype bf16x3 --steps 20 --warmup 3
run "gen bf16 B8"  --workload gen --dtype bf16 --steps 30 --warmup 3
run "cyc bf16 B8"  --workload cyc --steps 10 --warmup 2
run "cyc bf16x3 B8" --workload cyc --dtype bf16x3 --steps 6 --warmup 2
run "cyc bf16x3f B8" --workload cyc --dtype bf16x3f --steps 6 --warmup 2
run "p2p bf16"     --workload p2p --steps 20 --warmup 3
run "reg bf16"     --workload reg --steps 20 --warmup 3
for b in 1 2 4 8; do run "hd bf16 B$b" --batch $b --steps 30 --warmup 4; done
