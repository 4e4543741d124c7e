"""State-dict key inventory and a deterministic synthetic-weight generator.

Key names follow what the reference's checkpoints carry (main.py:531-560 strips
or adds a 'module.' prefix; backbone keys sit under 'ssl_model.model.' and use
fairseq's parameter names).  No checkpoint ships with this build, so tests and
benchmarks fill the same keys from a seeded generator: every tensor is drawn
from its own torch.Generator seeded by crc32(key) ^ seed, which makes a tensor
reproducible on any machine without moving a 1.2 GB file.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import BackboneConfig, XLSR_300M

BACKBONE_PREFIX = "ssl_model.model."
SLS_POOLED = lambda T, C: (T // 3) * (C // 3)  # noqa: E731  max_pool2d(3,3) output size


def backbone_shapes(cfg: BackboneConfig = XLSR_300M, leftovers: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter name -> shape for the wav2vec2 backbone (names relative to the
    fairseq model root; wav2vec/wav2vec2.py:257-362, :773-899, :989-1028)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    C, Fd = cfg.embed_dim, cfg.ffn_dim
    if leftovers:
        s["mask_emb"] = (C,)
    cin = 1
    for i, (cout, k, _) in enumerate(cfg.conv_layers):
        p = f"feature_extractor.conv_layers.{i}."
        s[p + "0.weight"] = (cout, cin, k)
        s[p + "0.bias"] = (cout,)
        s[p + "2.1.weight"] = (cout,)
        s[p + "2.1.bias"] = (cout,)
        cin = cout
    s["post_extract_proj.weight"] = (C, cin)
    s["post_extract_proj.bias"] = (C,)
    if leftovers:
        vq = cfg.final_dim
        s["quantizer.vars"] = (1, cfg.latent_groups * cfg.latent_vars, vq // cfg.latent_groups)
        s["quantizer.weight_proj.weight"] = (cfg.latent_groups * cfg.latent_vars, cin)
        s["quantizer.weight_proj.bias"] = (cfg.latent_groups * cfg.latent_vars,)
        s["project_q.weight"] = (cfg.final_dim, vq)
        s["project_q.bias"] = (cfg.final_dim,)
    s["encoder.pos_conv.0.bias"] = (C,)
    s["encoder.pos_conv.0.weight_g"] = (1, 1, cfg.conv_pos)
    s["encoder.pos_conv.0.weight_v"] = (C, C // cfg.conv_pos_groups, cfg.conv_pos)
    for l in range(cfg.layers):
        p = f"encoder.layers.{l}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (C, C)
            s[p + f"self_attn.{n}.bias"] = (C,)
        s[p + "self_attn_layer_norm.weight"] = (C,)
        s[p + "self_attn_layer_norm.bias"] = (C,)
        s[p + "fc1.weight"] = (Fd, C)
        s[p + "fc1.bias"] = (Fd,)
        s[p + "fc2.weight"] = (C, Fd)
        s[p + "fc2.bias"] = (C,)
        s[p + "final_layer_norm.weight"] = (C,)
        s[p + "final_layer_norm.bias"] = (C,)
    s["encoder.layer_norm.weight"] = (C,)
    s["encoder.layer_norm.bias"] = (C,)
    s["layer_norm.weight"] = (cin,)
    s["layer_norm.bias"] = (cin,)
    if leftovers:
        s["final_proj.weight"] = (cfg.final_dim, C)
        s["final_proj.bias"] = (cfg.final_dim,)
    return s


def sls_head_shapes(cfg: BackboneConfig, frames: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """Upstream SLS head parameters (first_bn, fc0, fc1, fc3)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["first_bn.weight"] = (1,)
    s["first_bn.bias"] = (1,)
    s["first_bn.running_mean"] = (1,)
    s["first_bn.running_var"] = (1,)
    s["fc0.weight"] = (1, cfg.embed_dim)
    s["fc0.bias"] = (1,)
    s["fc1.weight"] = (1024, SLS_POOLED(frames, cfg.embed_dim))
    s["fc1.bias"] = (1024,)
    s["fc3.weight"] = (2, 1024)
    s["fc3.bias"] = (2,)
    return s


def sae_head_shapes(cfg: BackboneConfig, dict_size: int = 4096, head: str = "sae") -> "OrderedDict[str, Tuple[int, ...]]":
    """In-tree head parameters (model.py:53-66, :171-189): head "sae" pools the sparse code,
    "sae_recon" its reconstruction, "ssl" the encoder output (no auto-encoder)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    C = cfg.embed_dim
    if head != "ssl":
        s["sae.decoder.weight"] = (C, dict_size)
        s["sae.encoder.weight"] = (dict_size, C)
        s["sae.encoder.bias"] = (dict_size,)
        s["sae.b_dec"] = (C,)
    in_dim = dict_size if head == "sae" else C
    s["classifier.0.weight"] = (in_dim,)
    s["classifier.0.bias"] = (in_dim,)
    s["classifier.1.weight"] = (256, in_dim)
    s["classifier.1.bias"] = (256,)
    s["classifier.4.weight"] = (2, 256)
    s["classifier.4.bias"] = (2,)
    return s


def _draw(key: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """Deterministic fp32 tensor for one state-dict key.  Scales are chosen so
    every stage carries signal of order one (unit-gain fan-in init for weight
    matrices, LayerNorm gains near one, small biases)."""
    leaf = key.rsplit(".", 1)[-1]
    z = _draw(key, shape, seed)
    if "running_var" in key:
        return 0.5 + z.abs()
    if "running_mean" in key:
        return 0.1 * z
    if key.endswith("first_bn.weight"):
        return 1.0 + 0.1 * z
    if leaf == "weight_g":
        return 1.0 + 0.25 * z.abs()
    if leaf in ("weight", "weight_v", "vars") and len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return z * (1.0 / math.sqrt(fan_in))
    if leaf == "weight":  # LayerNorm / BatchNorm gain
        return 1.0 + 0.1 * z
    if key.endswith("mask_emb"):
        return z.abs().clamp(max=1.0)
    return 0.05 * z  # biases, b_dec


def synth_state_dict(cfg: BackboneConfig = XLSR_300M, head: str = "sls", frames: int = 201,
                     seed: int = 1234, sae_dict: int = 4096, sae_k: int = 128,
                     leftovers: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Full Model state dict (no 'module.' prefix) with seeded values."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in backbone_shapes(cfg, leftovers).items():
        sd[BACKBONE_PREFIX + k] = synth_tensor(BACKBONE_PREFIX + k, shp, seed)
    if head == "sls":
        for k, shp in sls_head_shapes(cfg, frames).items():
            sd[k] = synth_tensor(k, shp, seed)
        # A trained BatchNorm holds the statistics of its input.  With these
        # seeded weights the variance of the weighted layer sum grows with depth
        # (measured: 1.24 at 2 layers, 5.8e3 at 24); mirror that so the head works
        # in its un-saturated range and score parity is a real test.
        var = 1.24 * (5800.0 / 1.24) ** ((cfg.layers - 2) / 22.0)
        sd["first_bn.running_var"] = torch.tensor([var], dtype=torch.float32)
        sd["first_bn.running_mean"] = torch.tensor([-0.5 * min(1.0, cfg.layers / 24.0)], dtype=torch.float32)
        sd["first_bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    elif head in ("sae", "sae_recon", "ssl"):
        if head != "ssl":
            sd["sae.k"] = torch.tensor(sae_k, dtype=torch.int)
        for k, shp in sae_head_shapes(cfg, sae_dict, head).items():
            sd[k] = synth_tensor(k, shp, seed)
    else:
        raise ValueError(f"unknown head {head!r}")
    return sd


def strip_module_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept DataParallel-saved dicts (main.py:542-560)."""
    if sd and all(k.startswith("module.") for k in sd):
        return {k[len("module."):]: v for k, v in sd.items()}
    return dict(sd)


def unwrap_checkpoint(obj) -> Dict[str, torch.Tensor]:
    """main.py:531-540: a checkpoint is either a bare state dict or a dict
    holding it under 'model_state_dict' / 'state_dict' / 'model'."""
    if isinstance(obj, dict):
        for k in ("model_state_dict", "state_dict", "model"):
            if k in obj and isinstance(obj[k], dict):
                return obj[k]
    return obj
