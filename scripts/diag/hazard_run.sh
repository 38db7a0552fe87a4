# On the GPU box: scripts/lds_neighbour_stress.py on every variant library of scripts/diag/hazard_variants.py.
# usage: bash scripts/diag/hazard_run.sh [WORKLOAD] [variants...]   (default workload: the 64 -> 32 channel 3x3 conv, forward only)
W=${1:-probe_64_32_3_128_fwd}; shift
D=cta_gan_amd/_build/diag
VARS=${@:-$(ls $D/libctagan_hip_*.so | sed 's/.*libctagan_hip_\(.*\)\.so/\1/')}
mkdir -p gpurun_out
for v in $VARS; do
  echo "== $v" | tee -a gpurun_out/hazard.log
  CTG_LIB=$PWD/$D/libctagan_hip_$v.so CTG_NO_COUT1=1 timeout -k 10 240 python scripts/lds_neighbour_stress.py $W 2>&1 | grep -E "TOTAL|worst|by lane" | tee -a gpurun_out/hazard.log
done
echo "== shipped library" | tee -a gpurun_out/hazard.log
CTG_NO_COUT1=1 timeout -k 10 240 python scripts/lds_neighbour_stress.py $W 2>&1 | grep -E "TOTAL" | tee -a gpurun_out/hazard.log
