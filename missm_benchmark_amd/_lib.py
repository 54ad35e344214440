"""ctypes binding of ``libmissm_hip.so`` (the C ABI declared in ``include/missm_hip.h``).

The product path has no fallback: if the shared library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISSM_LIB_PATH") or os.path.join(_HERE, "libmissm_hip.so")     # (MISSM_LIB_PATH: A/B runs against an experiment build)

P, I, F, L, U64 = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_ulonglong

# name -> argtypes, exactly mirroring include/missm_hip.h (tests check the export list against the header)
SIGNATURES = {
    "missm_gemm_nt": [P, P, P, I, I, I, I, I, I, F, P, P, P, P, I, I, I, I, I, P],
    "missm_gemm": [P, P, P, I, I, I, I, I, I, I, I, F, P, P, P, P, I, I, I, I, I, P, I, P],
    "missm_gemm_grouped": [I, P, P, P, I, I, I, I, I, I, I, I, F, P, P, P, P, I, I, I, I, I, P, I, P],
    "missm_transpose_pad": [P, P, I, I, I, I, P, I, P],
    "missm_colsum": [P, P, I, I, I, I, I, I, P],
    "missm_cast_weight": [P, P, P, I, I, I, P],
    "missm_cast_weights_batched": [P, I, I, P],
    "missm_layernorm_fwd": [P, P, P, I, I, I, P, P, P, P, P, P, I, I, F, I, P],
    "missm_layernorm_bwd": [P, I, F, P, I, P, P, P, P, P, I, P, P, P, I, I, I, P],
    "missm_layernorm_fwd_grouped": [I, P, P, P, P, P, P, I, I, F, I, P],
    "missm_layernorm_bwd_groupsum": [P, F, P, P, P, P, P, I, P, P, P, P, I, I, I, I, I, P],
    "missm_layernorm_bwd_grouped": [I, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "missm_cast_rows": [P, P, L, I, I, I, I, P],
    "missm_mean_rows": [P, P, I, I, I, P],
    "missm_attention_fwd": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, F, I, P],
    "missm_attention_bwd": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, F, I, P],
    "missm_unfold_patches": [P, P, I, I, I, I, I, I, L, L, L, I, P],
    "missm_embed_assemble": [P, P, P, P, I, I, I, I, P],
    "missm_token_embed_fwd": [P, P, P, P, I, I, I, I, P],
    "missm_token_embed_bwd": [P, P, P, P, I, I, I, I, P],
    "missm_argmax_rows": [P, P, I, I, P],
    "missm_small_linear_fwd": [P, P, P, P, I, I, I, I, I, P, L, P, I, F, I, P],
    "missm_add_block": [P, I, P, I, I, I, P],
    "missm_masked_copy_block": [P, I, P, I, I, I, P, L, I, P],
    "missm_small_linear_bwd": [P, I, P, P, P, P, P, P, I, I, I, P, L, P, I, F, I, I, P],
    "missm_gate_fwd": [P, I, P, P, I, I, P, L, I, P],
    "missm_gate_bwd": [P, P, I, P, P, I, P, I, I, P, L, I, P],
    "missm_l2norm_scale_fwd": [P, P, I, I, F, P],
    "missm_l2norm_scale_bwd": [P, P, P, I, I, F, P],
    "missm_cross_entropy": [P, P, P, P, I, I, P],
    "missm_kl_loss": [P, P, P, P, P, I, I, F, P],
    "missm_mse_loss": [P, P, P, P, L, P],
    "missm_ema_update": [P, P, L, F, P],
    "missm_preprocess_image": [P, I, I, I, I, I, P, I, F, F, F, F, P, P, P],
    "missm_sgat_fwd": [P, P, P, P, P, P, P, P, I, I, I, I, P],
    "missm_gelu_bwd": [P, P, P, L, P],
    "missm_sgat_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "missm_dropout_fwd": [P, P, P, L, F, U64, P],
    "missm_dropout_bwd": [P, P, P, L, F, P],
    "missm_adam_step": [P, P, P, P, L, I, F, F, F, F, F, F, P],
    "missm_adam_cast_batched": [P, I, L, L, L, I, F, F, F, F, F, F, I, P],
    "missm_lora_merge": [P, I, P, P, I, I, I, F, P],
    "missm_lora_grad": [P, I, P, P, P, P, I, I, I, F, P],
    "missm_buffer_mean": [P, L, P, P],
    "missm_kaldi_fbank": [P, L, P, P, I, F, F, F, F, F, F, P],
    "missm_mel_assemble": [P, I, I, P, I, I, I, I, F, F, P],
    "missm_sinc_resample": [P, L, P, I, I, I, I, P, L, P],
}
PLAIN = {"missm_last_error": ([], C.c_char_p), "missm_abi_version": ([], I), "missm_device_count": ([], I),
         "missm_gemm_set_debug_buffer": ([P], None), "missm_gemm_release_workspaces": ([], None),
         "missm_fbank_frames": ([L, F, F, F], I)}

ABI_VERSION = 11     # bumped with every signature change in include/missm_hip.h (capi.cpp: missm_abi_version)
_lib = None


class MissmError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise loudly if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so); this library links the same SONAME.  torch must be
    # imported first so that both resolve to ONE runtime instance: loaded the other way round, /opt/rocm's copy serves this
    # library, torch's copy serves torch, and the second runtime to touch the device reports "no ROCm-capable device".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MissmError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(missm_benchmark_amd/csrc/build.sh). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = I
    for name, (args, res) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.missm_abi_version() != ABI_VERSION:
        raise MissmError("libmissm_hip.so ABI version mismatch; rebuild")
    if "HIP_FORCE_DEV_KERNARG" not in os.environ:
        # (the library never sets process-wide environment itself - see missm_benchmark_amd/__init__.py)
        import warnings
        warnings.warn("HIP_FORCE_DEV_KERNARG is unset: kernel arguments stay in host-coherent memory (about -1.6 % on the training "
                      "step).  Export HIP_FORCE_DEV_KERNARG=1 before torch is imported, as bench.py and the train_ddp CLI do.",
                      RuntimeWarning, stacklevel=2)
    _lib = lib
    return lib


def check(rc: int, name: str):
    if rc != 0:
        msg = load().missm_last_error().decode(errors="replace")
        raise MissmError(f"{name} failed ({rc}): {msg}")


def call(name: str, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
