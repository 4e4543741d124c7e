#!/usr/bin/env python3
"""State-dict keys and shapes of the reference's Model built with its DEFAULT arguments
(`Model(None, 'cpu')`, model.py:150-160) and of its other constructor forms, produced by the reference's
own model.py / wav2vec2.py in the build container (stand-ins as in make_golden.py).  Two encoder layers:
the keys of the other layers follow the same pattern.  Writes state_dict_keys.json."""
import json
import os
import sys

import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import BackboneConfig  # noqa: E402


def main():
    w2v, ref_model, _, _ = mg.load_reference()
    cfg = BackboneConfig(layers=2)
    backbone = mg.ref_backbone(w2v, cfg)

    class _SSL(nn.Module):
        def __init__(self, device=None, cp_path=None):
            super().__init__()
            self.model = backbone
            self.device = device
            self.out_dim = cfg.embed_dim

    orig = ref_model.SSLModel
    ref_model.SSLModel = _SSL
    out = {}
    try:
        forms = {"default": {}, "use_sparse_features_false": {"use_sparse_features": False}, "use_sae_false": {"use_sae": False}}
        for name, kw in forms.items():
            m = ref_model.Model(None, "cpu", **kw)
            out[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    finally:
        ref_model.SSLModel = orig
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
