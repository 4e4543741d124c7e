"""Frozen architecture description of the scoring path.

Replaces the fairseq dataclass config the reference reads out of the XLS-R
checkpoint (wav2vec/wav2vec2.py:39-252).  Field comments name the fairseq
field each value stands for.  The HIP engine is specialised for the XLS-R-300M
geometry (conv width 512, model width 1024, head width 64); only the layer
count, the FFN width and the input length are free.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

HEAD_SLS = 0
HEAD_SAE = 1


@dataclass(frozen=True)
class BackboneConfig:
    conv_layers: Tuple[Tuple[int, int, int], ...] = (
        ((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2
    )  # conv_feature_layers
    embed_dim: int = 1024  # encoder_embed_dim
    ffn_dim: int = 4096  # encoder_ffn_embed_dim
    heads: int = 16  # encoder_attention_heads
    layers: int = 24  # encoder_layers
    conv_pos: int = 128  # conv_pos
    conv_pos_groups: int = 16  # conv_pos_groups
    ln_eps: float = 1e-5
    # pre-training leftovers kept only so reference checkpoints load strictly
    final_dim: int = 768
    latent_vars: int = 320
    latent_groups: int = 2

    @property
    def conv_dim(self) -> int:
        return self.conv_layers[-1][0]

    def conv_out_lengths(self, length: int) -> List[int]:
        """Frame count after every conv layer (wav2vec/wav2vec2.py:523-538)."""
        out, cur = [], int(length)
        for _, k, s in self.conv_layers:
            cur = int(math.floor((cur - k) / s + 1))
            out.append(cur)
        return out

    def num_frames(self, length: int) -> int:
        return self.conv_out_lengths(length)[-1]


XLSR_300M = BackboneConfig()

# 4.04 s at 16 kHz: the fixed cut the reference's loader produces
# (data_utils_SSL.py:58-65) -> 201 frames.
UTT_SAMPLES = 64600

# multiply-accumulate based FLOP count per utterance (2*MACs of convolutions
# and matrix products only), used for the roofline report.
def flops_per_utt(cfg: BackboneConfig, length: int, head: str = "sls",
                  sae_dict: int = 4096, sae_decode: bool = False) -> float:
    lens = cfg.conv_out_lengths(length)
    fl, cin = 0.0, 1
    for (cout, k, _), t in zip(cfg.conv_layers, lens):
        fl += 2.0 * t * cout * cin * k
        cin = cout
    T, C, Fd = lens[-1], cfg.embed_dim, cfg.ffn_dim
    fl += 2.0 * T * cin * C  # post_extract_proj
    fl += 2.0 * T * C * (C // cfg.conv_pos_groups) * cfg.conv_pos  # pos_conv
    per_layer = 2.0 * T * C * 3 * C + 2.0 * T * C * C + 4.0 * T * T * C + 4.0 * T * C * Fd
    fl += cfg.layers * per_layer
    if head == "sls":
        pooled = (T // 3) * (C // 3)
        fl += 2.0 * cfg.layers * C + 2.0 * cfg.layers * T * C + 2.0 * pooled * 1024 + 2.0 * 1024 * 2
    else:
        fl += 2.0 * T * C * sae_dict * (2 if sae_decode else 1) + 2.0 * sae_dict * 256 + 2.0 * 256 * 2
    return fl
