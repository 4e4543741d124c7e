#!/usr/bin/env python3
"""Golden vectors for the in-tree Model's other constructor forms and its interpretability
output, produced by the reference's own model.py (run in the build container only):

  * Model(..., use_sae=True, use_sparse_features=False)   -> "sae_recon"
  * Model(..., use_sae=False)                              -> "ssl"
  * Model(..., use_sae=True)(x, return_interpretability=True) -> get_interpretability_info dict

Same stand-ins and seeded weights as make_golden.py (imported from it).  Writes
head_variants_tiny.npz, head_variants_xlsr2.npz and head_variants_report.json.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs nothing until its functions are called)
from make_golden import BackboneConfig, synth_state_dict, orc, BACKBONE_PREFIX  # noqa: E402


def build(w2v, ref_model, cfg, sd, use_sae, use_sparse, sae_dict, sae_k):
    backbone = mg.ref_backbone(w2v, cfg)

    class _SSL(nn.Module):
        def __init__(self, device=None, cp_path=None):
            super().__init__()
            self.model = backbone
            self.device = device
            self.out_dim = cfg.embed_dim

    _SSL.extract_feat = ref_model.SSLModel.extract_feat
    orig = ref_model.SSLModel
    ref_model.SSLModel = _SSL
    try:
        m = ref_model.Model(None, "cpu", use_sae=use_sae, use_sparse_features=use_sparse,
                            sae_dict_size=sae_dict, sae_k=sae_k)
    finally:
        ref_model.SSLModel = orig
    if cfg.embed_dim != 1024:  # the reference hard-codes 1024 (model.py:173, :177-179)
        if use_sae:
            m.sae = ref_model.AutoEncoderTopK(cfg.embed_dim, sae_dict, sae_k)
        in_dim = sae_dict if (use_sae and use_sparse) else cfg.embed_dim
        m.classifier = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, 256), nn.ReLU(), nn.Dropout(0.3),
                                     nn.Linear(256, 2))
    m.load_state_dict(sd, strict=True)
    return m.eval()


@torch.no_grad()
def run(w2v, ref_model, cfg, L, B, seed, sae_dict, sae_k):
    T = cfg.num_frames(L)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    ocfg = orc.OracleConfig(conv_layers=cfg.conv_layers, embed_dim=cfg.embed_dim, ffn_dim=cfg.ffn_dim,
                            heads=cfg.heads, layers=cfg.layers, conv_pos=cfg.conv_pos,
                            conv_pos_groups=cfg.conv_pos_groups, sae_dict=sae_dict, sae_k=sae_k)
    out, rep = {"seed": np.int64(seed), "B": np.int64(B), "L": np.int64(L)}, {}
    for head, use_sae, sparse in (("sae_recon", True, False), ("ssl", False, True)):
        sd = synth_state_dict(cfg, head, T, seed, sae_dict, sae_k)
        m = build(w2v, ref_model, cfg, sd, use_sae, sparse, sae_dict, sae_k)
        logp, loss = m(wav)
        o = orc.model_forward(sd, wav, ocfg, head)
        rep[f"logp_{head}_maxabs"] = float((o - logp).abs().max())
        out[f"logp_{head}"] = logp.numpy()
        if loss is not None:
            out[f"sae_loss_{head}"] = np.float32(loss)
    sd = synth_state_dict(cfg, "sae", T, seed, sae_dict, sae_k)
    m = build(w2v, ref_model, cfg, sd, True, True, sae_dict, sae_k)
    logp, loss, info = m(wav, return_interpretability=True)
    logp2, info2 = m(wav, return_sae_loss=False, return_interpretability=True)
    assert torch.equal(logp, logp2) and set(info) == set(info2)
    bb = orc.backbone_forward(sd, wav, ocfg, pfx="ssl_model.model.")
    _, _, code = orc.sae_head(sd, bb["x"], ocfg, return_code=True)
    oi = orc.interpretability_info(code)
    for k in ("avg_activation", "top20_values", "sparsity", "activation_freq"):
        rep[f"interp_{k}_maxabs"] = float((oi[k] - info[k]).abs().max())
    rep["interp_top20_same"] = bool(torch.equal(oi["top20_features"], info["top20_features"]))
    out.update({"interp_logp": logp.numpy(), "interp_avg_activation": info["avg_activation"].numpy(),
                "interp_top20_features": info["top20_features"].numpy(),
                "interp_top20_values": info["top20_values"].numpy(), "interp_sparsity": info["sparsity"].numpy(),
                "interp_activation_freq": info["activation_freq"].numpy(),
                "interp_code_shape": np.array(info["sparse_features"].shape),
                "interp_code_nnz_per_row": np.array([(info["sparse_features"] > 0).sum(-1).max().item(),
                                                     (info["sparse_features"] > 0).sum(-1).min().item()])})
    return out, rep


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, ref_model, ref_backup, ref_eer = mg.load_reference()
    report = {}
    tiny = BackboneConfig(conv_layers=((32, 10, 5),) + ((32, 3, 2),) * 4 + ((32, 2, 2),) * 2,
                          embed_dim=64, ffn_dim=128, heads=4, layers=2, conv_pos=16, conv_pos_groups=4,
                          final_dim=32, latent_vars=8, latent_groups=2)
    out, rep = run(w2v, ref_model, tiny, 3200, 2, 1234, 256, 16)
    np.savez_compressed(os.path.join(HERE, "head_variants_tiny.npz"), **out)
    report["tiny"] = rep
    out, rep = run(w2v, ref_model, BackboneConfig(layers=2), 64600, 2, 1234, 4096, 128)
    np.savez_compressed(os.path.join(HERE, "head_variants_xlsr2.npz"), **out)
    report["xlsr_2layer"] = rep
    with open(os.path.join(HERE, "head_variants_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
