"""Drop-in replacement for the reference's `model` module (reference model.py):
`from model import Model, SSLModel, AutoEncoderTopK` keeps working for
main.py's eval path and the eval_* scripts, with forward() running on the
MI355X HIP path.  See slsforasvspoof-2021-df_amd/model.py.

`Model` here has the reference's own signature and defaults (model.py:150-160: use_sae=True, the in-tree
TopK-SAE model), so scripts that rely on the defaults (analyze_sae_neurons.py:57-65) get the sae.* /
classifier.* keys their checkpoints hold.  The upstream SLS model, `Model(args, device)` of the project the
reference forked, is `SLSModel(args, device)` (or `Model(args, device, head="sls")`)."""
import slsmi  # noqa: F401  (registers the package under the name `slsmi`)
from slsmi.model import AutoEncoderTopK, SSLModel, getAttenF  # noqa: F401
from slsmi.model import Model as _Model

__all__ = ["Model", "SLSModel", "SSLModel", "AutoEncoderTopK", "getAttenF"]


class Model(_Model):
    def __init__(self, args, device, cp_path='xlsr2_300m.pt', use_sae=True, use_sparse_features=True,
                 sae_dict_size=4096, sae_k=128, sae_weight=0.1, **kw):
        super().__init__(args, device, cp_path=cp_path, use_sae=use_sae, use_sparse_features=use_sparse_features,
                         sae_dict_size=sae_dict_size, sae_k=sae_k, sae_weight=sae_weight, **kw)


class SLSModel(_Model):
    """The upstream SLS model: XLS-R + layer-weighted sum (getAttenF, model_backup.py:186-202) + fc0 / first_bn /
    fc1 / fc3 head."""

    def __init__(self, args, device, cp_path='xlsr2_300m.pt', **kw):
        kw.pop("head", None)
        super().__init__(args, device, cp_path=cp_path, head="sls", **kw)
