"""Drop-in replacement for the reference's `model` module (reference model.py):
`from model import Model, SSLModel, AutoEncoderTopK` keeps working for
main.py's eval path and the eval_* scripts, with forward() running on the
MI355X HIP path.  See slsforasvspoof-2021-df_amd/model.py."""
import slsmi  # noqa: F401  (registers the package under the name `slsmi`)
from slsmi.model import AutoEncoderTopK, Model, SSLModel, getAttenF  # noqa: F401

__all__ = ["Model", "SSLModel", "AutoEncoderTopK", "getAttenF"]
