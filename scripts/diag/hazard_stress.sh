# On the GPU box: tests/test_neighbour_stress_gpu.py (the PatchGAN victim: the one the failing build fails) on every variant
# library of scripts/diag/hazard_variants.py.   usage: bash scripts/diag/hazard_stress.sh [REPEATS] [variants...]
N=${1:-2}; shift
D=cta_gan_amd/_build/diag
VARS=${@:-$(ls $D/libctagan_hip_*.so | sed 's/.*libctagan_hip_\(.*\)\.so/\1/')}
mkdir -p gpurun_out
for v in $VARS shipped; do
  for i in $(seq $N); do
    if [ $v = shipped ]; then L=""; else L=$PWD/$D/libctagan_hip_$v.so; fi
    R=$(CTG_LIB=$L timeout -k 10 300 python -m pytest tests/test_neighbour_stress_gpu.py -m gpu -q -k patchgan 2>&1 | tail -1)
    echo "$v run $i: $R" | tee -a gpurun_out/hazard_stress.log
  done
done
