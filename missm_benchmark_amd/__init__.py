"""MI355X-native hot path of MissM-Benchmark: LanguageBind CLIP towers + missing-modality fusion head on HIP kernels.

``install()`` registers this package's drop-in modules under the reference's import names so that reference-style code
(``from languagebind import LanguageBind``; ``from src.model.baseline import finetune_model``) runs on this build.
"""
import os as _os
import sys as _sys

__all__ = ["install", "kernarg_mode"]

# Kernel-argument placement: HIP puts kernel arguments in host-coherent memory by default and every workgroup of every
# launch fetches them from there (the GEMM kernels carry up to eight operand sets by value, ~0.7 KB); HIP_FORCE_DEV_KERNARG=1 stages
# them in device memory, +1.6 % on the training step.  The runtime reads the variable once, when it initialises (at `import torch`),
# and the setting is process-wide (torch's own kernels and RCCL see it too) - so it is the ENTRY POINTS that default it (bench.py,
# `python -m missm_benchmark_amd.train_ddp`, tests/conftest.py), never this library: importing the package changes no environment.
# `kernarg_mode()` reports what the process runs with; `_lib.load()` says so once when the variable is unset.


def kernarg_mode() -> str:
    """'device' (HIP_FORCE_DEV_KERNARG=1), 'host' (=0) or 'default (host-coherent)' when the variable is unset"""
    v = _os.environ.get("HIP_FORCE_DEV_KERNARG")
    return "default (host-coherent)" if v is None else ("device" if v == "1" else "host")


def install():
    from . import languagebind as _lb
    from .src import model as _m
    from .src.model import baseline as _b
    from . import src as _src
    _sys.modules["languagebind"] = _lb
    _sys.modules["src"] = _src
    _sys.modules["src.model"] = _m
    _sys.modules["src.model.baseline"] = _b
    return _lb, _b
