# usage: bash scripts/build_prev.sh FILE.hip [REV] [HEADER.h]
# Build cta_gan_amd/_build/libctagan_hip_prev.so with FILE (and, if given, the HEADER it includes) taken from git REV
# (default HEAD) and every other object from the current build: the "previous" side of an in-box A/B of one kernel
# file (CTG_LIB, scripts/ab_lib.sh).
set -e
F=$1; REV=${2:-HEAD}; H=$3; C=cta_gan_amd/csrc; B=cta_gan_amd/_build
FLAGS=$(python -c "from cta_gan_amd import build; print(' '.join(build.flags_for('$F')))")
git show $REV:$C/$F > $C/_prev_$F
if [ -n "$H" ]; then
  git show $REV:$C/$H > $C/_prev_$H
  sed -i "s/#include \"$H\"/#include \"_prev_$H\"/" $C/_prev_$F
fi
/opt/rocm/bin/hipcc $FLAGS -c $C/_prev_$F -o $B/_prev.o
rm -f $C/_prev_$F $C/_prev_$H
OBJS=$(ls $B/*.o | grep -v "_prev.o" | grep -v "/${F%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $B/libctagan_hip_prev.so $OBJS $B/_prev.o
rm $B/_prev.o
ls -la $B/libctagan_hip_prev.so
