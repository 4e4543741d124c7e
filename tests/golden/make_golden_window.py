#!/usr/bin/env python3
"""Golden vectors for the window-TopK variant, produced by the reference's own
model_window_topk.py (run in the build container only):

  * AutoEncoderTopK._window_topk (model_window_topk.py:118-203) on seeded activations
  * Model(..., sae_window_size=8) (model_window_topk.py:271-393): log-probs and SAE loss

Same stand-ins and seeded weights as make_golden.py (imported from it).  Writes
window_topk_cases.npz, window_model.npz and window_report.json.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import BackboneConfig, synth_state_dict, orc  # noqa: E402

# (name, B, T, D, k, window, seed, sparsity shift): activations = relu(randn - shift)
CASES = [("t37_w8", 2, 37, 256, 16, 8, 11, 0.0), ("t201_w8", 1, 201, 512, 32, 8, 12, 0.5),
         ("t40_w6", 2, 40, 128, 8, 6, 13, 0.0), ("t9_w8", 3, 9, 128, 8, 8, 14, 0.0),
         ("t50_w2", 1, 50, 128, 8, 2, 15, 0.3)]


def case_acts(B, T, D, seed, shift):
    g = torch.Generator().manual_seed(seed)
    return torch.relu(torch.randn(B, T, D, generator=g) - shift)


def build(w2v, ref_win, cfg, sd, sae_dict, sae_k, window):
    backbone = mg.ref_backbone(w2v, cfg)

    class _SSL(nn.Module):
        def __init__(self, device=None, cp_path=None):
            super().__init__()
            self.model = backbone
            self.device = device
            self.out_dim = cfg.embed_dim

    _SSL.extract_feat = ref_win.SSLModel.extract_feat
    orig = ref_win.SSLModel
    ref_win.SSLModel = _SSL
    try:
        m = ref_win.Model(None, "cpu", use_sae=True, use_sparse_features=True, sae_dict_size=sae_dict, sae_k=sae_k,
                          sae_window_size=window)
    finally:
        ref_win.SSLModel = orig
    if cfg.embed_dim != 1024:  # the reference hard-codes 1024 (model_window_topk.py:297)
        m.sae = ref_win.AutoEncoderTopK(cfg.embed_dim, sae_dict, sae_k, window)
    m.load_state_dict(sd, strict=True)
    return m.eval()


@torch.no_grad()
def run_model(w2v, ref_win, cfg, L, B, seed, sae_dict, sae_k, window, tag, out, rep):
    T = cfg.num_frames(L)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    sd = synth_state_dict(cfg, "sae", T, seed, sae_dict, sae_k)
    m = build(w2v, ref_win, cfg, sd, sae_dict, sae_k, window)
    logp, loss = m(wav)
    ocfg = orc.OracleConfig(conv_layers=cfg.conv_layers, embed_dim=cfg.embed_dim, ffn_dim=cfg.ffn_dim,
                            heads=cfg.heads, layers=cfg.layers, conv_pos=cfg.conv_pos,
                            conv_pos_groups=cfg.conv_pos_groups, sae_dict=sae_dict, sae_k=sae_k, sae_window=window)
    bb = orc.backbone_forward(sd, wav, ocfg, pfx="ssl_model.model.")
    ologp, oloss = orc.sae_head(sd, bb["x"], ocfg, return_sae_loss=True)
    rep[f"{tag}_logp_maxabs"] = float((ologp - logp).abs().max())
    rep[f"{tag}_loss_abs"] = float((oloss - loss).abs())
    out.update({f"{tag}_seed": np.int64(seed), f"{tag}_B": np.int64(B), f"{tag}_L": np.int64(L),
                f"{tag}_window": np.int64(window), f"{tag}_logp": logp.numpy(), f"{tag}_sae_loss": np.float32(loss)})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, ref_model, ref_backup, ref_eer = mg.load_reference()
    ref_win = mg.load_ref_module("ref_model_window_topk", os.path.join(mg.REF, "model_window_topk.py"))
    report, cases = {}, {}
    for name, B, T, D, k, w, seed, shift in CASES:
        acts = case_acts(B, T, D, seed, shift)
        sae = ref_win.AutoEncoderTopK(8, D, k, w)
        ref = sae._window_topk(acts.clone(), k, w)
        mine = orc.window_topk(acts, k, w)
        report[f"{name}_oracle_equal"] = bool(torch.equal(ref, mine))
        cnt = orc.window_vote_counts(acts, k, w)
        report[f"{name}_frames_with_fewer_than_k_votes"] = int((cnt < k).sum())
        cases[name] = ref.numpy()
        cases[name + "_meta"] = np.array([B, T, D, k, w, seed], dtype=np.int64)
        cases[name + "_shift"] = np.float32(shift)
    np.savez_compressed(os.path.join(HERE, "window_topk_cases.npz"), **cases)
    out = {}
    tiny = BackboneConfig(conv_layers=((32, 10, 5),) + ((32, 3, 2),) * 4 + ((32, 2, 2),) * 2,
                          embed_dim=64, ffn_dim=128, heads=4, layers=2, conv_pos=16, conv_pos_groups=4,
                          final_dim=32, latent_vars=8, latent_groups=2)
    run_model(w2v, ref_win, tiny, 3200, 2, 1234, 256, 16, 4, "tiny", out, report)
    run_model(w2v, ref_win, BackboneConfig(layers=2), 64600, 2, 1234, 4096, 128, 8, "xlsr2", out, report)
    np.savez_compressed(os.path.join(HERE, "window_model.npz"), **out)
    with open(os.path.join(HERE, "window_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
