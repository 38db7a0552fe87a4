# HBM-byte and SQ-counter views of the benchmarked step (separate --pmc passes, --kernel-trace only), on the GPU box:
#   bash scripts/pmc_step.sh OUTDIR [bf16|bf16x3]
# -> OUTDIR/{trace,fetch,write,sq1,sq2,sq3}/p_*.csv, then scripts/pmc_tables.py writes profiles/pmc_dominant[_bf16x3].json,
#    profiles/pmc_hbm[_bf16x3].json (HBM bytes per launch, stamped with the kernel-build digest) and prints the SQ table
OUT=${1:-gpurun_out/pmc_step}
DT=${2:-bf16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --dtype $DT --no-cpu-baseline --no-parity-mode --no-kernel-events"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 bench.py --steps 4 --warmup 2 --dtype $DT --no-cpu-baseline --no-parity-mode --no-kernel-events > $OUT/trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ARGS > $OUT/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ARGS > $OUT/write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS > $OUT/sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq3 -o p -- python3 $ARGS > $OUT/sq3.log 2>&1 &&
python3 scripts/trace_summary.py $OUT/trace/p_kernel_trace.csv 6 80 > $OUT/step_trace_summary.txt &&
python3 scripts/pmc_tables.py $OUT $DT > $OUT/tables.md && cat $OUT/tables.md
