# interleaved A/B of the current library against cta_gan_amd/_build/libctagan_hip_prev.so (scripts/build_prev.sh) in ONE box
for i in 1 2 3; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  env CTG_LIB=$PWD/cta_gan_amd/_build/libctagan_hip_prev.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('previous', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
