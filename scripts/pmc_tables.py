"""The counter tables of profiles/ from the passes of scripts/pmc_step.sh:   python scripts/pmc_tables.py OUTDIR bf16|bf16x3

* profiles/pmc_dominant[_bf16x3].json -- HBM bytes per launch of the three residual-block conv kernels (what bench.py reports as
  `roofline.traffic`), profiles/pmc_hbm[_bf16x3].json -- the same for the HBM-bound kernels of `roofline.hbm`; both stamped with
  the kernel-build digest.  FETCH_SIZE x 2 (gfx950: it counts half the bytes of wide coalesced streaming reads,
  /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE, separate passes, --kernel-trace only.
* stdout: the SQ table (shares of wave cycles, instructions per wave, MFMA-pipe busy share).
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE): the counter sums busy cycles over the chip's 1024 SIMDs,
  GRBM_GUI_ACTIVE sums the dispatch's cycles over the 8 XCDs."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cta_gan_amd import build

out_dir, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
x3 = mode == "bf16x3"
ES = 4 if x3 else 2                           # bytes per activation element
T256 = 16 * 128 * 128 * 256 * ES              # one [16,128,128,256] map
WB = 9 * 256 * 256 * (4 if x3 else 2)         # its 3x3 weights ([hi | lo] halves in bf16x3)
# the whole template argument list, NIE (last) included: the fused conv + InstanceNorm instantiations are their own row
HALO = ("conv_halo_kernelIDF16b8bfpair_tLi128ELi4ELi2ELi8ELi1ELi16ELb%dELi3ELb0ELb1ELb0ELb%dE" if x3
        else "conv_halo_kernelIDF16bDF16bLi128ELi4ELi2ELi8ELi1ELi16ELb%dELi3ELb0ELb0ELb0ELb%dE")
# label -> (name fragment, Grid_Size (work-items) or None, algorithmic bytes per launch or None)
DOMINANT = collections.OrderedDict([
    ("fwd", (HALO % (0, 0), 1048576, 2 * T256 + WB)),
    ("bwd_data", (HALO % (1, 0), 1048576, 4 * T256 + WB)),      # gradient, skip gradient, InstanceNorm input z in; result out
    ("wgrad", ("conv_wgrad_halo_kernel<64, 64, 9, 1, 3>", 131072, 2 * T256)),      # (+ 75.5 MB of fp32 split-K partials out)
])
PAIR = "bfpair_t" if x3 else "DF16b"
HBM = collections.OrderedDict([
    ("in_apply", (("in_apply_kernel", "bfpair_t" if x3 else "16"), 128 * 256 * 16, 2 * T256)),
    ("in_bwd_apply", (("in_bwd_apply_kernel", "bfpair_t" if x3 else "16"), 128 * 256 * 16, 3 * T256)),
    ("conv32", ("conv_strip32_kernel", None, 2 * 16 * 512 * 512 * 32 * 2 + 9 * 32 * 32 * 2)),
    ("convt64", ("conv_stript_128_64_kernel", None, 16 * 256 * 256 * 128 * 2 + 16 * 512 * 512 * 64 * 2 + 9 * 128 * 64 * 2)),
    ("convs2", ("conv_strips2_64_128_kernel", None, 16 * 256 * 256 * 128 * 2 + 16 * 512 * 512 * 64 * 2 + 9 * 128 * 64 * 2)),
    # their split-pair forms (round 5): 4 bytes per value
    ("convt64p", ("conv_striptp_128_64_kernel", None, 16 * 256 * 256 * 128 * 4 + 16 * 512 * 512 * 64 * 4 + 9 * 128 * 64 * 4)),
    ("convs2p", ("conv_strips2p_64_128_kernel", None, 16 * 256 * 256 * 128 * 4 + 16 * 512 * 512 * 64 * 4 + 9 * 128 * 64 * 4)),
])
SQ_ROWS = list(DOMINANT.items()) + [
    ("fwd + InstanceNorm + ReLU in one launch (NIE; forwards that keep nothing)", (HALO % (0, 1), 1048576, None)),
    ("fwd + InstanceNorm + skip in one launch (NIE)", (HALO % (1, 1), 1048576, None)),
    ("conv_igemm (gather kernel, all launches)", ("conv_igemm_kernel", None, None)),
    ("conv_stript_128_64", ("conv_stript_128_64_kernel", None, None)),
    ("conv_strips2_64_128", ("conv_strips2_64_128_kernel", None, None)),
    ("conv_strips2_128_256", ("conv_strips2_128_256_kernel", None, None)),
    ("conv_striptp_128_64 (split pair, round 5)", ("conv_striptp_128_64_kernel", None, None)),
    ("conv_strips2p_64_128 (split pair, round 5)", ("conv_strips2p_64_128_kernel", None, None)),
    ("conv_wgrad_s2m (merged polyphase stride-2 weight gradient, round 5)", ("conv_wgrad_s2m_kernel", None, None)),
    ("conv_strip32 (Reg 32->32 @ 512^2)", ("conv_strip32_kernel", None, None)),
    ("conv_halo BN=32 (Reg 32-channel layers)", ("Li32ELi4ELi1E", 4194304, None)),
    ("conv_small (first layers)", ("conv_small_kernel", None, None)),
    ("in_apply (res-block maps)", HBM["in_apply"]),
    ("in_bwd_apply (res-block maps)", HBM["in_bwd_apply"]),
]


def name_ok(frag, name):
    """frag: a substring (also tried in its mangled spelling), or a tuple of substrings that must all occur"""
    if isinstance(frag, tuple):
        return all(f in name for f in frag)
    return frag in name or frag.replace(", ", "ELi") in name


def load(path, rows):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        for label, (frag, grid, _) in rows:
            if name_ok(frag, r["Kernel_Name"]) and (grid is None or int(r["Grid_Size"]) == grid):
                acc[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"_n": len(next(iter(d.values())))} for k, d in acc.items()}


def traffic_table(rows, fname, note):
    f = load(os.path.join(out_dir, "fetch", "p_counter_collection.csv"), rows.items())
    w = load(os.path.join(out_dir, "write", "p_counter_collection.csv"), rows.items())
    out = {"per_gpu_batch": 16, "size": 512, "dtype": mode, "build": build._digest()[:16], "fetch_correction": 2.0, "source": note,
           "kernels": {}}
    n_all = b_all = 0
    for key, (frag, grid, alg) in rows.items():
        if key not in f or key not in w:
            continue
        fk, wk = f[key]["FETCH_SIZE"], w[key]["WRITE_SIZE"]
        traffic = int(fk * 1024 * 2.0 + wk * 1024)
        out["kernels"][key] = {"kernel": " ".join(frag) if isinstance(frag, tuple) else frag, "dispatches": f[key]["_n"], "fetch_size_kb_avg": round(fk, 2),
                               "write_size_kb_avg": round(wk, 2), "traffic_bytes_per_launch": traffic,
                               "algorithmic_bytes_per_launch": alg}
        n_all += f[key]["_n"]
        b_all += traffic * f[key]["_n"]
    if n_all:
        out["traffic_bytes_per_launch"] = int(b_all / n_all)
    return out, fname


def in_apply_split(table):
    """in_apply with and without a residual operand are ONE kernel on one grid: told apart by what they fetch (the two passes
    dispatch the same sequence, so the lists are index-aligned)."""
    frag, grid, _ = HBM["in_apply"]

    def seq(path, counter):
        rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
                if r["Counter_Name"] == counter and name_ok(frag, r["Kernel_Name"]) and int(r["Grid_Size"]) == grid]
        return [v for _, v in sorted(rows)]
    fp, wp = (os.path.join(out_dir, d, "p_counter_collection.csv") for d in ("fetch", "write"))
    if not (os.path.exists(fp) and os.path.exists(wp)):
        return
    fv, wv = seq(fp, "FETCH_SIZE"), seq(wp, "WRITE_SIZE")
    if len(fv) != len(wv) or not fv:
        return
    for key, lo, hi, alg in (("in_apply", 0.5, 1.5, 2 * T256), ("in_apply_res", 1.5, 2.6, 3 * T256)):
        keep = [i for i, v in enumerate(fv) if lo * T256 <= v * 1024 * 2.0 < hi * T256]
        if keep:
            fk, wk = sum(fv[i] for i in keep) / len(keep), sum(wv[i] for i in keep) / len(keep)
            table["kernels"][key] = {"kernel": " ".join(frag), "dispatches": len(keep), "fetch_size_kb_avg": round(fk, 2),
                                     "write_size_kb_avg": round(wk, 2), "traffic_bytes_per_launch": int(fk * 2048 + wk * 1024),
                                     "algorithmic_bytes_per_launch": alg}


note = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; bench.py --steps 2 --warmup 1 --dtype %s (scripts/pmc_step.sh)" % mode
sfx = "_bf16x3" if x3 else ""
dom, dom_name = traffic_table(DOMINANT, "pmc_dominant%s.json" % sfx, note)
hbm, hbm_name = traffic_table(HBM, "pmc_hbm%s.json" % sfx, note)
in_apply_split(hbm)

a = load(os.path.join(out_dir, "sq1", "p_counter_collection.csv"), SQ_ROWS)
b = load(os.path.join(out_dir, "sq2", "p_counter_collection.csv"), SQ_ROWS)
c = load(os.path.join(out_dir, "sq3", "p_counter_collection.csv"), SQ_ROWS)
busy = {}
print("## SQ counters, Hd step B=16 @ 512^2, %s (build %s)\n" % (mode, build._digest()[:16]))
print("| kernel | parked (s_waitcnt / barrier) | issue-stalled | issuing | of which VALU | MFMA pipe busy | LDS conflict / LDS-active | "
      "instructions per wave: VALU / SALU / LDS / MFMA / branch / VMEM |\n|---|---|---|---|---|---|---|---|")
for label, _ in SQ_ROWS:
    m, n, k = a.get(label), b.get(label), c.get(label)
    if not m:
        continue
    wc = m["SQ_WAVE_CYCLES"]
    mb = None
    if k and k.get("GRBM_GUI_ACTIVE"):
        mb = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * k["GRBM_GUI_ACTIVE"])
        busy[label] = round(mb, 4)
    row = "| %s | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %s | %.2f |" % (
        label, 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc,
        100 * m["SQ_ACTIVE_INST_VALU"] / wc, "-" if mb is None else "%.1f %%" % (100 * mb),
        m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0))
    if n and n.get("SQ_WAVES"):
        w_ = n["SQ_WAVES"]
        row += " %.0f / %.0f / %.0f / %.0f / %.0f / %.0f |" % (n["SQ_INSTS_VALU"] / w_, n["SQ_INSTS_SALU"] / w_, n["SQ_INSTS_LDS"] / w_,
                                                            n["SQ_INSTS_MFMA"] / w_, n["SQ_INSTS_BRANCH"] / w_,
                                                            (n["SQ_INSTS_VMEM_RD"] + n["SQ_INSTS_VMEM_WR"]) / w_)
    else:
        row += " |"
    print(row)
if all(k in busy for k in DOMINANT):      # launch-weighted like bench.py's avg_launch_ms (36 : 18 : 18)
    dom["mfma_busy"] = {"fwd": busy["fwd"], "bwd_data": busy["bwd_data"], "wgrad": busy["wgrad"],
                        "launch_weighted": round((2 * busy["fwd"] + busy["bwd_data"] + busy["wgrad"]) / 4, 4),
                        "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE): share of SIMD cycles the matrix pipe is busy"}
for table, name in ((dom, dom_name), (hbm, hbm_name)):
    json.dump(table, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print("\n## HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE)\n")
print("| kernel | dispatches | traffic MB | algorithmic MB | ratio |\n|---|---|---|---|---|")
for table in (dom, hbm):
    for key, v in table["kernels"].items():
        print("| %s | %d | %.1f | %.1f | %.2f |" % (key, v["dispatches"], v["traffic_bytes_per_launch"] / 1e6,
                                                   v["algorithmic_bytes_per_launch"] / 1e6,
                                                   v["traffic_bytes_per_launch"] / v["algorithmic_bytes_per_launch"]))
