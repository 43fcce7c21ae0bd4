"""ctypes binding of libvideoitg_hip.so (the C ABI declared in include/videoitg_hip.h).

The product path has NO fallback: if the library is missing or an entry point fails, this raises.
PyTorch is used only as plumbing around it (device memory, streams).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITG_PROFILING_LIB=1 (set by tools/, never by the product) selects the profiling build: the same kernels plus the research
# GEMM variants and ablation switches of include/videoitg_hip_profiling.h
PROFILING = os.environ.get("VITG_PROFILING_LIB", "0") not in ("", "0")
LIB_PATH = os.path.join(_HERE, "libvideoitg_hip_prof.so" if PROFILING else "libvideoitg_hip.so")
# A/B measurements of experiment builds (csrc/build.sh --exp NAME; tools/archive/r04_calls.sh.txt): an explicit path to another build of the SAME C ABI.  Still no
# fallback - if the named file is missing, load() raises.
if os.environ.get("VITG_LIB_PATH"):
    _alt = os.path.abspath(os.environ["VITG_LIB_PATH"])
    # only another build of THIS library, sitting next to it (csrc/build.sh writes them there): not a way to load arbitrary code
    if os.path.dirname(_alt) != _HERE or not (os.path.basename(_alt).startswith("libvideoitg_hip_exp_") and _alt.endswith(".so")):
        raise ImportError(f"VITG_LIB_PATH must name a libvideoitg_hip_exp_<NAME>.so inside {_HERE} (csrc/build.sh --exp NAME), not {_alt}")
    LIB_PATH = _alt
    import warnings
    warnings.warn(f"videoitg_amd: VITG_LIB_PATH is set - loading the experiment build {os.path.basename(_alt)} instead of the product library "
                  f"(bench.py reports it as library.path)", RuntimeWarning, stacklevel=2)

BF16, F16, F32 = 0, 1, 2
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF = 0, 1, 2
GEMM_STORE16, GEMM_F32, GEMM_SWIGLU = 0, 1, 2

_ERR = {-1: "VITG_ERR_SHAPE", -2: "VITG_ERR_ALIGN", -3: "VITG_ERR_DTYPE", -4: "VITG_ERR_ARG", -5: "VITG_ERR_LAUNCH"}

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes; must list every symbol include/videoitg_hip.h declares (tests/test_abi.py checks both ways)
SIGNATURES = {
    "vitg_version": [],
    "vitg_device_info": [C.c_char_p, _i],
    "vitg_set_option": [C.c_char_p, _i],
    "vitg_patchify": [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    "vitg_gemm": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "vitg_gemm_res16": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _i, _p],
    "vitg_patch_embed16": [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p, _i, _p],
    "vitg_gemm_splitk_plan": [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_long)],
    "vitg_gemm_splitk": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p],
    "vitg_gemm_splitk16": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p],
    "vitg_quant_rows_fp8": [_p, _l, _i, _p, _l, _p, _l, _i, _p],
    "vitg_norm_quant_fp8": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _f, _p],
    "vitg_gemm_fp8": [_p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "vitg_gemm_fp8_res16": [_p, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p],
    "vitg_norm_quant_fp8_16": [_p, _l, _i, _p, _p, _p, _l, _p, _l, _i, _f, _p],
    "vitg_gemm_fp8_mx": [_p, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p],
    "vitg_mfma_probe": [_p, _i, _i, _i, _p, C.POINTER(C.c_double), _p],
    "vitg_layernorm": [_p, _i, _p, _p, _p, _i, _i, _l, _i, _f, _p],
    "vitg_layernorm16": [_p, _i, _p, _p, _p, _i, _i, _l, _i, _f, _p],
    "vitg_rmsnorm": [_p, _i, _p, _p, _i, _i, _l, _i, _f, _p],
    "vitg_rope": [_p, _i, _p, _p, _l, _i, _i, _i, _p],
    "vitg_attention": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "vitg_attention_dyn": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "vitg_bilinear_pool": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "vitg_bilinear_pool16": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "vitg_gather_rows": [_p, _l, _p, _i, _i, _i, _p, _l, _i, _p],
    "vitg_embed_gather": [_p, _i, _p, _i, _p, _i, _i, _p],
    "vitg_pool_score": [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    "vitg_rank": [_p, _i, _p, _p, _p],
    "vitg_bce_loss": [_p, _p, _i, _p, _p],
    "vitg_preprocess_tmp_bytes": [_i, _i, _i, _i, C.POINTER(C.c_long)],
    "vitg_preprocess_u8": [_p, _i, _i, _i, _l, _i, _p, _i, _i, _p, _l, _p],
    # operand-split precision tier (csrc/split.hip; round 6)
    "vitg_norm_split": [_p, _i, _p, _p, _p, _i, _i, _l, _i, _f, _p],
    "vitg_rope_split": [_p, _i, _l, _p, _p, _l, _i, _i, _p],
    "vitg_bilinear_pool_split": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "vitg_gemm_split16": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _l, _p],
    "vitg_attention_split": [_p, _p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p],
}

# include/videoitg_hip_profiling.h: only in libvideoitg_hip_prof.so
PROFILING_SIGNATURES = {
    "vitg_gemm_streamk_plan": [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_long)],
    "vitg_gemm_streamk": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p],
    "vitg_gemm_wf": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "vitg_weight_fragments_bytes": [_i, _i, C.POINTER(C.c_long)],
    "vitg_pack_weight_fragments": [_p, _i, _i, _i, _p, _p],
    "vitg_gemm_norm_plan": [_i, _i, _i, C.POINTER(C.c_int)],
    "vitg_gemm_norm_producer": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _i, _p, _p, _i, _p, _p],
    "vitg_norm_stats_finalize": [_p, _l, _i, _i, _f, _i, _p, _p],
    "vitg_gemm_norm_consumer": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p],
    "vitg_gemm_timeline_buffer": [_p, _i],
}

_lib = None


class VitgError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library (no GPU needed for this) and set prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitgError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            + ("(profiling flavour: `bash videoitg_amd/csrc/build.sh --profiling`) " if PROFILING else "") +
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the selector forward.")
    lib = C.CDLL(LIB_PATH)
    for name, args in list(SIGNATURES.items()) + (list(PROFILING_SIGNATURES.items()) if PROFILING else []):
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise VitgError(f"{what} failed: {_ERR.get(rc, rc)}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
