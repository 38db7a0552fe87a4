"""Generate tests/golden/metrics_*.npz from the REFERENCE's own evaluation functions (run in the build container only).

trainer/HdTrainer.py cannot be imported (torchvision, visdom, lpips, pydicom, SimpleITK, skimage, cv2 are absent), so the
script parses that file, compiles ONLY `to_windowdata` (:41-64) and the `PSNR` / `MAE` / `UQI` methods (:1089-1125) --
pure numpy functions -- and runs them.  The masking sequence of the test loop (:1008-1050) is driven through
oracle.ref_metrics.slice_metrics with the reference functions plugged in, and independently checked here against an
inline transcription.  Nothing of the reference's text is written to the repo: only input seeds and output numbers.
"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/trainer/HdTrainer.py"


def load_reference_functions(path=REF, cls="Hd_Trainer_x2"):
    tree = ast.parse(open(path).read())
    wanted_top, wanted_methods = {"to_windowdata"}, {"PSNR", "MAE", "UQI"}
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted_top:
            body.append(node)
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in wanted_methods:
                    body.append(sub)
    names = {n.name for n in body}
    assert names == wanted_top | wanted_methods, names
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return (ns["to_windowdata"], lambda f, r: ns["MAE"](None, f, r), lambda f, r: ns["PSNR"](None, f, r),
            lambda f, r: ns["UQI"](None, f, r))


def cases():
    from cta_gan_amd import synth
    out = {}
    # smooth synthetic "CT slices" in [-1, 1] with an exact -1 background rim, like the reference's normalised DICOMs
    for name, size, wc, ww, seed in (("a", 96, 40.0, 400.0, 1), ("b", 128, 60.0, 300.0, 2), ("c", 64, 300.0, 1500.0, 3)):
        real = synth.synth_smooth_images("met_real_%d" % seed, 1, size)[0, 0].numpy().astype(np.float32)
        fake = (real + 0.15 * synth.synth_smooth_images("met_noise_%d" % seed, 1, size)[0, 0].numpy()).astype(np.float32)
        fake = np.clip(fake, -1, 1)
        real[:6] = -1
        real[:, -5:] = -1
        fake[:6] = -1
        out[name] = (fake, real, wc, ww)
    # degenerate: everything background after masking (exercises the x.size == 0 branches)
    z = np.full((32, 32), -1.0, np.float32)
    out["allbg"] = (z.copy(), z.copy(), 40.0, 400.0)
    return out


def main():
    from oracle import ref_metrics
    fns = load_reference_functions()
    fns_cyc = load_reference_functions("/root/reference/trainer/CycTrainer.py", "Cyc_Trainer")
    gold = os.path.join(ROOT, "tests", "golden")
    for name, (fake, real, wc, ww) in cases().items():
        got = ref_metrics.slice_metrics(fake.copy(), real.copy(), wc, ww, fns=fns)
        got_cyc = ref_metrics.slice_metrics_cyc(fake.copy(), real.copy(), wc, ww, fns=fns_cyc)
        win_real = fns[0](real.copy(), wc, ww)
        np.savez_compressed(os.path.join(gold, "metrics_%s.npz" % name), fake=fake, real=real, wc=wc, ww=ww,
                            metrics=got, metrics_cyc=got_cyc, win_real=win_real.astype(np.float32))
        print(name, got.tolist(), got_cyc.tolist())


if __name__ == "__main__":
    main()
