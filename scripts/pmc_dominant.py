"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) of `bench.py` -> profiles/pmc_dominant.json:
HBM bytes per launch of the three residual-block conv kernels (forward, backward-data with fused epilogue, weight gradient),
with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced
streaming reads: x2; WRITE_SIZE is exact for 16-byte-per-lane stores), stamped with the kernel build it was measured on.

    python scripts/pmc_dominant.py FETCH_counter_collection.csv WRITE_counter_collection.csv SOURCE_NOTE
"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cta_gan_amd import build

KERNELS = {   # bench.py's roofline.kernels key -> (kernel name fragment, x-grid of the B=16 512^2 res-block launch, algorithmic bytes)
    "fwd": ("conv_halo_kernelIDF16bDF16bLi128ELi4ELi2ELi8ELi1ELi16ELb0ELi3E", 1048576, 134217728 * 2 + 1179648),
    # gradient in + weights + skip gradient (res) + InstanceNorm input z (IN-backward sums) + result out
    "bwd_data": ("conv_halo_kernelIDF16bDF16bLi128ELi4ELi2ELi8ELi1ELi16ELb1ELi3E", 1048576, 134217728 * 4 + 1179648),
    # gradient + layer input in; fp32 partials out (slabs x 9 x 256 x 256 x 4 B)
    "wgrad": ("conv_wgrad_halo_kernel<64, 64, 9, 1, 3>", 131072, 134217728 * 2),
}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key, (frag, grid, _) in KERNELS.items():
            if frag in r["Kernel_Name"] and int(r["Grid_Size"]) == grid:
                acc[key].append(float(r["Counter_Value"]))
    return acc


# ---- the HBM-bound kernels of bench.py's `roofline.hbm` (same two passes) -> profiles/r03_pmc_hbm.json
# key -> (kernel name fragment, (work-items in x, workgroups in y), algorithmic bytes per launch, fetch window in bytes or None).  in_apply with
# and without a residual operand are ONE kernel on one grid: told apart by what they fetch (two tensors vs one).
T256 = 16 * 128 * 128 * 256 * 2          # one [16,128,128,256] bf16 map
HBM = {
    "in_apply": ("in_apply_kernel", (128 * 256, 16), 2 * T256, (0.5 * T256, 1.5 * T256)),
    "in_apply_res": ("in_apply_kernel", (128 * 256, 16), 3 * T256, (1.5 * T256, 2.6 * T256)),
    "in_bwd_apply": ("in_bwd_apply_kernel", (128 * 256, 16), 3 * T256, None),
    # (the non-fused 32 -> 32 launches of Reg's full-resolution level run on csrc/conv_strip.h: any grid)
    "conv32": ("conv_strip32_kernel", None, 2 * 16 * 512 * 512 * 32 * 2 + 9 * 32 * 32 * 2, None),
    # the 128 <-> 64 channel stride-2 layers (csrc/conv_stript.h, conv_strips2.h): [16,256,256,128] and [16,512,512,64] once each
    "convt64": ("conv_stript_128_64_kernel", None, 16 * 256 * 256 * 128 * 2 + 16 * 512 * 512 * 64 * 2 + 9 * 128 * 64 * 2, None),
    "convs2": ("conv_strips2_64_128_kernel", None, 16 * 256 * 256 * 128 * 2 + 16 * 512 * 512 * 64 * 2 + 9 * 128 * 64 * 2, None),
}


def hbm_rows(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key, (frag, grid, _, _) in HBM.items():
            if frag in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid[0] * grid[1]):      # Grid_Size = work-items of the launch
                rows[key].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


def hbm_table(fetch_csv, write_csv, note):
    fr, wr = hbm_rows(fetch_csv, "FETCH_SIZE"), hbm_rows(write_csv, "WRITE_SIZE")
    out = {"per_gpu_batch": 16, "size": 512, "dtype": "bf16", "build": build._digest()[:16], "fetch_correction": 2.0,
           "source": note, "kernels": {}}
    for key, (frag, grid, alg, window) in HBM.items():
        fv, wv = fr.get(key, []), wr.get(key, [])
        if window is not None and len(fv) == len(wv):      # the two passes dispatch the same sequence: classify by fetch, index-aligned
            keep = [i for i, v in enumerate(fv) if window[0] <= v * 1024 * 2.0 < window[1]]
            fv, wv = [fv[i] for i in keep], [wv[i] for i in keep]
        if not fv or not wv:
            continue
        fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
        out["kernels"][key] = {"kernel": frag, "dispatches": len(fv), "fetch_size_kb_avg": round(fk, 2),
                               "write_size_kb_avg": round(wk, 2), "traffic_bytes_per_launch": int(fk * 1024 * 2.0 + wk * 1024),
                               "algorithmic_bytes_per_launch": alg}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_pmc_hbm.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


hbm_table(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"per_gpu_batch": 16, "size": 512, "dtype": "bf16", "build": build._digest()[:16], "fetch_correction": 2.0,
       "source": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
n_all = b_all = 0
for key, (frag, grid, alg) in KERNELS.items():
    fk, wk = sum(f[key]) / len(f[key]), sum(w[key]) / len(w[key])
    traffic = int(fk * 1024 * 2.0 + wk * 1024)
    out["kernels"][key] = {"kernel": frag, "dispatches": len(f[key]), "fetch_size_kb_avg": round(fk, 2),
                           "write_size_kb_avg": round(wk, 2), "traffic_bytes_per_launch": traffic,
                           "algorithmic_bytes_per_launch": alg}
    n_all += len(f[key])
    b_all += traffic * len(f[key])
out["traffic_bytes_per_launch"] = int(b_all / n_all)     # launch-weighted over the three kernels, like bench.py's avg_launch_ms
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_dominant.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
