# HBM-byte and SQ-counter views of the benchmarked step (separate --pmc passes, --kernel-trace only), on the GPU box:
#   bash scripts/pmc_step.sh OUTDIR
# -> OUTDIR/{fetch,write,sq1,sq2}/p_counter_collection.csv; then profiles/pmc_dominant.json + profiles/r03_pmc_hbm.json
#    (scripts/pmc_dominant.py) and OUTDIR/sq_table.txt (scripts/sq_table.py)
OUT=${1:-gpurun_out/pmc_step}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-kernel-events"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ARGS > $OUT.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ARGS > $OUT.write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS > $OUT.sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT.sq2.log 2>&1 &&
python3 scripts/pmc_dominant.py $OUT/fetch/p_counter_collection.csv $OUT/write/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; bench.py --steps 2 --warmup 1 (scripts/pmc_step.sh)" > $OUT.pmc.json.log &&
python3 scripts/sq_table.py $OUT/sq1/p_counter_collection.csv $OUT/sq2/p_counter_collection.csv > $OUT/sq_table.txt && cat $OUT/sq_table.txt
