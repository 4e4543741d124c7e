#!/usr/bin/env python3
"""Golden vectors for batches of utterances of different lengths, produced by the reference's own
wav2vec/wav2vec2.py run with a padding_mask (wav2vec2.py:540-547, :567-586, :912-913, :1050), in the
build container only.  Also runs the reference on every utterance alone at its own length and records
how far the two agree on the utterance's own frames (the claim the batched HIP path rests on).

Same stand-ins and seeded weights as make_golden.py (imported from it).  Writes varlen_cases.npz and
varlen_report.json.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import BackboneConfig, synth_state_dict, orc, BACKBONE_PREFIX  # noqa: E402

TINY = BackboneConfig(conv_layers=((32, 10, 5),) + ((32, 3, 2),) * 4 + ((32, 2, 2),) * 2,
                      embed_dim=64, ffn_dim=128, heads=4, layers=2, conv_pos=16, conv_pos_groups=4,
                      final_dim=32, latent_vars=8, latent_groups=2)


@torch.no_grad()
def run(w2v, cfg, tag, L, lengths, seed, full, out, rep):
    B = len(lengths)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    pad = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lengths):
        pad[b, n:] = True
        wav[b, n:] = 0.0
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), 1234, 256 if cfg.embed_dim != 1024 else 4096,
                          16 if cfg.embed_dim != 1024 else 128)
    bb = {k[len(BACKBONE_PREFIX):]: v for k, v in sd.items() if k.startswith(BACKBONE_PREFIX)}
    model = mg.ref_backbone(w2v, cfg)
    model.load_state_dict(bb, strict=True)
    res = model(wav, padding_mask=pad, mask=False, features_only=True)
    x = res["x"]
    fmask = res["padding_mask"]
    frames = [int((~fmask[b]).sum()) if fmask is not None else x.shape[1] for b in range(B)]
    assert frames == [cfg.num_frames(n) for n in lengths], frames
    layers = [lr[0].transpose(0, 1) for lr in res["layer_results"]]  # [B, T, C]
    # each utterance alone, at its own length, no mask
    worst_x, worst_l = 0.0, 0.0
    for b, n in enumerate(lengths):
        one = model(wav[b:b + 1, :n], mask=False, features_only=True)
        worst_x = max(worst_x, float((one["x"][0] - x[b, :frames[b]]).abs().max()))
        worst_l = max(worst_l, float((one["layer_results"][-1][0][:, 0] - layers[-1][b, :frames[b]]).abs().max()))
    rep[f"{tag}_alone_vs_masked_x_maxabs"] = worst_x
    rep[f"{tag}_alone_vs_masked_last_layer_maxabs"] = worst_l
    # oracle with lengths against the reference
    ocfg = orc.OracleConfig(conv_layers=cfg.conv_layers, embed_dim=cfg.embed_dim, ffn_dim=cfg.ffn_dim,
                            heads=cfg.heads, layers=cfg.layers, conv_pos=cfg.conv_pos,
                            conv_pos_groups=cfg.conv_pos_groups)
    o = orc.backbone_forward(sd, wav, ocfg, pfx=BACKBONE_PREFIX, lengths=lengths)
    rep[f"{tag}_oracle_x_maxabs_own_frames"] = max(
        float((o["x"][b, :frames[b]] - x[b, :frames[b]]).abs().max()) for b in range(B))
    out[f"{tag}_seed"], out[f"{tag}_L"] = np.int64(seed), np.int64(L)
    out[f"{tag}_lengths"] = np.array(lengths, dtype=np.int64)
    out[f"{tag}_frames"] = np.array(frames, dtype=np.int64)
    if full:
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_layers"] = torch.stack(layers).numpy()
    else:
        out[f"{tag}_x_slice"] = x[:, ::10, :16].numpy()
        out[f"{tag}_last_layer_slice"] = layers[-1][:, ::10, :16].numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, _, _, _ = mg.load_reference()
    out, rep = {}, {}
    run(w2v, TINY, "tiny", 3200, [3200, 2000, 1000], 1234, True, out, rep)
    run(w2v, BackboneConfig(layers=2), "xlsr2", 64600, [64600, 40000, 19000], 1234, False, out, rep)
    np.savez_compressed(os.path.join(HERE, "varlen_cases.npz"), **out)
    with open(os.path.join(HERE, "varlen_report.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
