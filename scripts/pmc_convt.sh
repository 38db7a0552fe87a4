# SQ / L2 counter view of the transposed-conv micro-benchmark (scripts/convt_bench.py): sliding-window kernel vs merged-class halo kernel.
# usage (on the GPU box): bash scripts/pmc_convt.sh OUTDIR
OUT=${1:-gpurun_out/pmct}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH"
P3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="TCC_HIT_sum TCC_MISS_sum"
MODES=${MODES:-"strip mc"}
for mode in $MODES; do
  if [ $mode = mc ]; then export CTG_NO_STRIPT=1; else unset CTG_NO_STRIPT; fi
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
    i=$((i+1))
    case " ${PASSES:-1 2 3 4 5 6} " in *" $i "*) ;; *) continue;; esac
    echo "pass $mode $i"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/${mode}_$i -o p -- python3 scripts/convt_bench.py > $OUT.${mode}_$i.log 2>&1
  done
done
python3 - "$OUT" $MODES <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for mode in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s_*/p_counter_collection.csv" % (out, mode)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_halo" not in k and "conv_stript" not in k:
                continue
            agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in sorted(agg.items()):
        print("==", mode, key)
        m = {c: sum(v) / len(v) for c, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1.0)
        for c in sorted(m):
            print("  %-28s %14.0f   %6.3f of wave cycles" % (c, m[c], m[c] / wc))
PY
