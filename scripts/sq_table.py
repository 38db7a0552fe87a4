"""SQ-counter table of the step's main kernels (shares of wave cycles; instructions per wave) from two rocprofv3 --pmc passes
(scripts/pmc_step.sh): python scripts/sq_table.py SQ1.csv SQ2.csv"""
import collections, csv, sys
KERNELS = [   # label, name fragment, Grid_Size (work-items of the launch) or None
    ("conv_halo fwd (res blocks)", "conv_halo_kernelIDF16bDF16bLi128ELi4ELi2ELi8ELi1ELi16ELb0ELi3E", 1048576),
    ("conv_halo FUSE bwd-data (res blocks)", "conv_halo_kernelIDF16bDF16bLi128ELi4ELi2ELi8ELi1ELi16ELb1ELi3E", 1048576),
    ("conv_wgrad_halo<64,64,9> (res blocks)", "conv_wgrad_halo_kernel<64, 64, 9, 1, 3>", 131072),
    ("conv_igemm<256,128> (what is left on the gather kernel)", "conv_igemm_kernelIDF16bDF16bLi256ELi128E", None),
    ("conv_stript_128_64 (u2 forward / d1 bwd-data, sliding window)", "conv_stript_128_64_kernel", None),
    ("conv_strips2_64_128 (d1 forward / u2 bwd-data, sliding window)", "conv_strips2_64_128_kernel", None),
    ("conv_strips2_128_256 (d2 forward / u1 bwd-data, sliding window, 8 waves)", "conv_strips2_128_256_kernel", None),
    ("conv_strip32 (Reg, 32 -> 32 ch @ 512^2, non-fused launches)", "conv_strip32_kernel", None),
    ("conv_halo BN=32 FUSE (Reg, 32 ch @ 512^2 backward-data)", "bLi32ELi4ELi1E", 4194304),
    ("in_apply (res-block maps)", "in_apply_kernel", 524288),
    ("in_bwd_apply (res-block maps)", "in_bwd_apply_kernel", 524288),
]


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        for label, frag, grid in KERNELS:
            name = r["Kernel_Name"]
            if (frag in name or frag.replace(", ", "ELi") in name) and (grid is None or int(r["Grid_Size"]) == grid):
                acc[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


a, b = load(sys.argv[1]), load(sys.argv[2])
print("| kernel | parked (s_waitcnt / barrier) | issue-stalled | issuing | of which VALU | LDS conflict / LDS-active | instructions per wave: VALU / SALU / LDS / MFMA / branch / VMEM |")
print("|---|---|---|---|---|---|---|")
for label, _, _ in KERNELS:
    m, n = a.get(label), b.get(label)
    if not m:
        continue
    wc = m["SQ_WAVE_CYCLES"]
    row = "| %s | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.2f |" % (
        label, 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc,
        100 * m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0))
    if n and n.get("SQ_WAVES"):
        w = n["SQ_WAVES"]
        row += " %.0f / %.0f / %.0f / %.0f / %.0f / %.0f |" % (n["SQ_INSTS_VALU"] / w, n["SQ_INSTS_SALU"] / w, n["SQ_INSTS_LDS"] / w,
                                                            n["SQ_INSTS_MFMA"] / w, n["SQ_INSTS_BRANCH"] / w,
                                                            (n["SQ_INSTS_VMEM_RD"] + n["SQ_INSTS_VMEM_WR"]) / w)
    else:
        row += " |"
    print(row)
