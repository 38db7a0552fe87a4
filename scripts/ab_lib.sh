# interleaved A/B of the current library against cta_gan_amd/_build/libctagan_hip_prev.so (scripts/build_prev.sh) in ONE box;
# prints slices/s, ms/step and the avg launch time (ms) of the three residual-block conv kernels (fwd, bwd-data, wgrad)
ARGS=${@:---steps 20 --warmup 3}
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], [k["avg_ms"] for k in d["roofline"]["kernels"]])'
for i in 1 2 3; do
  python bench.py $ARGS --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "$P" current
  env CTG_LIB=$PWD/cta_gan_amd/_build/libctagan_hip_prev.so python bench.py $ARGS --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "$P" previous
done
