"""ctypes binding of libctagan_hip.so (include/ctagan_hip.h).

The product path has NO fallback: if the library is missing or a symbol does not
resolve, importing an op raises.  Every call returns an int status that `_check`
turns into a RuntimeError ('Errors').
"""
from __future__ import annotations

import ctypes
import os

# torch FIRST: the ROCm wheel bundles its own libamdhip64; if libctagan_hip.so were loaded before it, the system
# runtime under /opt/rocm would be bound instead and the process would hold two HIP runtimes (our launches then
# fail with hipErrorNoDevice while torch works).  Importing torch here pins the load order for every entry path.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTG_LIB: another build of the same library (developer A/B of two kernel versions inside one box, scripts/ab_lib.sh)
LIB_PATH = os.environ.get("CTG_LIB") or os.path.join(_HERE, "_build", "libctagan_hip.so")

_I, _L, _P, _F = ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_float

# name -> argument type string: i int, l long, p pointer, f float, d double  (order as in include/ctagan_hip.h)
SIGNATURES = {
    "ctg_conv_igemm": "iipppp" + "i" * 20 + "ppppp",
    "ctg_conv_igemm_classes": "i" + "pppp" + "i" * 14 + "ppppppp",
    "ctg_conv_wgrad": "ipppiiiiiiiiiiiiipp",
    "ctg_wgrad_reduce": "piiiipiilllip",
    "ctg_wgrad_reduce_multi": "ippppppppppppp",
    "ctg_in_stats": "ipiiiiiipppp",
    "ctg_in_finalize": "piiiiippp",
    "ctg_in_apply": "ipippipipiiiiip",
    "ctg_in_apply_part": "ipipippipipiiiiip",
    "ctg_in_bwd_partial": "ipipiippiiiiiipp",
    "ctg_in_bwd_apply": "ipipiippppipiiiiip",
    "ctg_in_bwd_stats": "ipipiippipiiiiiipp",
    "ctg_in_bwd": "ipipiippipiiiiiipppp",
    "ctg_grad_combine": "ipipiipiipiiiiip",
    "ctg_fold_f32": "ppiiiiip",
    "ctg_act_bwd_f32": "ppiplp",
    "ctg_bias_grad": "ipiiiiiiiippip",
    "ctg_bias_grad_act": "ipiipiipiiiiiiippip",
    "ctg_maxpool2_fwd": "ipipiiiiip",
    "ctg_maxpool2_bwd": "ipipipiiiiiip",
    "ctg_bilinear_fwd": "ipipiiiiiiip",
    "ctg_bilinear_bwd": "ipipiiiiiiip",
    "ctg_copy_channels": "ipipiilp",
    "ctg_split_weights": "plpilp",
    "ctg_split_weights_multi": "ippppp",
    "ctg_pair_convert": "iplplilp",
    "ctg_abi_version": "",
    "ctg_chan_pad": "ipipilp",
    "ctg_im2col_pack": "ippiiiiiiiiipiiip",
    "ctg_conv_smallcin": "ippiiiiiiiiipiiipipiiiippp",
    "ctg_conv_tail7": "ipipppiiiip",
    "ctg_corr_smallcin": "piiiiiippiiiiiiiiiipip",
    "ctg_weight_pack": "ipllliipiiip",
    "ctg_weight_pack_multi": "iippppppppppp",
    "ctg_warp_fwd": "ppllllpiiip",
    "ctg_warp_bwd": "ppllllpppiiipp",
    "ctg_smooth_fwd": "plllliiiifppp",
    "ctg_smooth_bwd": "plllliiiifppip",
    "ctg_l1_fwd": "ppplfppp",
    "ctg_l1_bwd": "ppplfppip",
    "ctg_lsgan_fwd": "piiiffffippp",
    "ctg_lsgan_bwd": "piiiffffippp",
    "ctg_sum_scalars": "ippp",
    "ctg_act_bwd_sum_f32": "ppiplppip",
    "ctg_avgpool_fwd": "piipp",
    "ctg_avgpool_bwd": "piipp",
    "ctg_to_windowdata": "ppppilp",
    "ctg_window_metrics": "ppppiliippp",
    "ctg_ssim": "ppppiiiiidppp",
    "ctg_conv_cout1_fwd": "ipipppiiiiiiiiip",
    "ctg_conv_cout1_bwd": "ipppiiiiiiiiip",
    "ctg_conv_cout1_wgrad": "ippipiiiiiiiiip",
    "ctg_hu_to_inputs": "pffpplp",
    "ctg_resize_nearest": "piiipiip",
    "ctg_adam_step": "ipppppffffipp",
    "ctg_adam_tick": "pffp",
}
# include/ctagan_hip_diag.h: diagnostics that ship in the same library but are not part of the product ABI (bound when present)
DIAG_SIGNATURES = {
    "ctg_lds_canary": "iipip",
}
_CT = {"i": _I, "l": _L, "p": _P, "f": _F, "d": ctypes.c_double}
ABI_VERSION = 11      # CTG_ABI_VERSION of include/ctagan_hip.h this table was written against

_lib = None


def ensure_built():
    """Build the library in-tree when the kernel sources changed since it was built and hipcc is here (a fresh clone, an
    edited kernel); a box without hipcc runs the library that travelled with the tree.  Launchers call this in the PARENT
    before they start ranks (bench.py spawn_ranks, train.py), so a multi-minute compile never runs inside a rank's first kernel
    call with the process group already up.  A failed rebuild raises."""
    if os.environ.get("CTG_LIB") or os.environ.get("CTG_NO_AUTOBUILD") is not None:
        return
    from . import build as _build
    if _build.is_current():
        return
    try:
        _build._hipcc()
    except RuntimeError:
        return
    import sys
    print("cta_gan_amd: kernel sources changed -- building %s (hipcc, a few minutes)" % LIB_PATH, file=sys.stderr, flush=True)
    # a failed rebuild is fatal: the signatures above belong to the NEW sources, and an older library would be called with
    # mis-typed arguments (round-3 advisor); load() also refuses a library whose ctg_abi_version() differs
    _build.build()


def load():
    """Load (once) and return the ctypes library; raises if it is absent -- no CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    ensure_built()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libctagan_hip.so not found at %s and hipcc is not available to build it (`python -m cta_gan_amd.build`): "
            "the HIP path is the only implementation; there is no fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    try:
        ver = lib.ctg_abi_version
    except AttributeError:
        raise RuntimeError("%s predates ctg_abi_version(): rebuild it (`python -m cta_gan_amd.build --force`)" % LIB_PATH)
    ver.argtypes, ver.restype = [], _I
    if ver() != ABI_VERSION:
        raise RuntimeError("%s has C-ABI version %d, this binding was written against %d: rebuild it "
                           "(`python -m cta_gan_amd.build --force`)" % (LIB_PATH, ver(), ABI_VERSION))
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = [_CT[c] for c in sig]
        fn.restype = _I
    for name, sig in DIAG_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = [_CT[c] for c in sig]
            fn.restype = _I
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        if status == 1:
            raise RuntimeError("%s: invalid argument (CTG_EINVAL)" % what)
        raise RuntimeError("%s: HIP error %d" % (what, status - 1000))
