"""MI355X-native hot path of MissM-Benchmark: LanguageBind CLIP towers + missing-modality fusion head on HIP kernels.

``install()`` registers this package's drop-in modules under the reference's import names so that reference-style code
(``from languagebind import LanguageBind``; ``from src.model.baseline import finetune_model``) runs on this build.
"""
import sys as _sys

__all__ = ["install"]


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
