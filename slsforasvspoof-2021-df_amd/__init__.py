"""MI355X-native scoring path for XLS-R-300M + SLS / TopK-SAE audio-deepfake
detection: host-side mirror of the reference's `model` module over a C-ABI HIP
library (csrc/ -> libslsmi.so)."""
from .config import BackboneConfig, XLSR_300M, UTT_SAMPLES, flops_per_utt  # noqa: F401
from . import weights  # noqa: F401
from . import _lib  # noqa: F401
from ._lib import SlsmiError  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # torch-dependent modules are imported on first use
    if name in ("model", "engine", "ops", "scoring"):
        import importlib
        return importlib.import_module(f"slsmi.{name}")
    if name in ("Model", "SSLModel", "AutoEncoderTopK", "getAttenF"):
        from . import model as _m
        return getattr(_m, name)
    raise AttributeError(name)
