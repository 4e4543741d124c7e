"""Reading the checkpoint files the reference reads, without executing anything from them.

Two kinds of file reach the scoring path:

* the pre-trained XLS-R backbone in fairseq's layout, `{'args': Namespace | None, 'cfg': dict | None,
  'model': state_dict, ...}`, which the reference hands to fairseq's loader (model.py:113-115) and whose
  embedded configuration decides the architecture that loader builds;
* a trained detector as main.py:753-764 writes it: `{'epoch', 'model_state_dict', 'optimizer_state_dict',
  'train_eer', 'val_eer', 'best_val_eer' (numpy.float64 scalars: compute_eer returns eer * 100 as one),
  'args': vars(args), ...}`, or a bare state dict, with or without DataParallel's 'module.' prefix
  (main.py:531-560).

Both are read with `torch.load(weights_only=True)` plus an allow-list of exactly the inert value types those
files hold beside tensors and containers: `argparse.Namespace` (a bag of attributes) and numpy's scalar
reconstructor with the dtype classes of float64 / int64 scalars.  Anything else in a file is refused.
"""
from __future__ import annotations

import argparse
import ast
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import BackboneConfig


def _allow_list():
    items = [argparse.Namespace, np.dtype, np.float64, np.int64, np.float32, np.bool_]
    try:
        from numpy._core import multiarray as _ma   # numpy >= 2
    except ImportError:  # pragma: no cover
        from numpy.core import multiarray as _ma
    # files written under numpy 1.x name the reconstructor by its old module path
    items += [(_ma.scalar, "numpy.core.multiarray.scalar"), (_ma.scalar, "numpy._core.multiarray.scalar")]
    for name in ("Float64DType", "Int64DType", "Float32DType", "BoolDType"):
        cls = getattr(np.dtypes, name, None)
        if cls is not None:
            items.append(cls)
    return items


def safe_load(path: str):
    """torch.load(path, weights_only=True) with the allow-list above; raises SlsmiError naming the refused
    type otherwise (the remedy is then to re-save the state dict alone)."""
    try:
        with torch.serialization.safe_globals(_allow_list()):
            return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        raise _lib.SlsmiError(
            f"cannot read {path} without executing code from it ({type(e).__name__}: {e}); re-save the "
            "weights as a plain state dict, e.g. torch.save(ckpt['model'], path)") from e


def embedded_model_cfg(obj) -> Optional[dict]:
    """The model section of the configuration a fairseq checkpoint embeds: `cfg['model']` (dict; newer files)
    or `args` (Namespace; files of the xlsr_53_56k.pt era), as a plain dict; None when the file has neither
    (a bare state dict)."""
    if not isinstance(obj, dict):
        return None
    cfg = obj.get("cfg")
    if isinstance(cfg, dict) and isinstance(cfg.get("model"), (dict, argparse.Namespace)):
        m = cfg["model"]
        return dict(vars(m)) if isinstance(m, argparse.Namespace) else dict(m)
    args = obj.get("args")
    if isinstance(args, argparse.Namespace):
        return dict(vars(args))
    if isinstance(args, dict) and "encoder_layers" in args:
        return dict(args)
    return None


def _conv_layers(v) -> Tuple[Tuple[int, int, int], ...]:
    if isinstance(v, str):   # fairseq stores the Python expression, e.g. "[(512, 10, 5)] + [(512, 3, 2)] * 4 + ..."
        v = _eval_layer_expr(ast.parse(v.strip(), mode="eval").body)
    return tuple(tuple(int(x) for x in t) for t in v)


def _eval_layer_expr(node):
    """Lists / tuples of integers joined by + and * only: the grammar of conv_feature_layers."""
    if isinstance(node, ast.Constant) and isinstance(node.value, int):
        return node.value
    if isinstance(node, (ast.List, ast.Tuple)):
        vals = [_eval_layer_expr(e) for e in node.elts]
        return vals if isinstance(node, ast.List) else tuple(vals)
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Mult)):
        a, b = _eval_layer_expr(node.left), _eval_layer_expr(node.right)
        return a + b if isinstance(node.op, ast.Add) else a * b
    raise ValueError("conv_feature_layers holds something other than lists of integer tuples")


# fairseq field -> (BackboneConfig attribute or fixed value, what this library implements)
_FIXED = {
    "extractor_mode": "layer_norm",   # Fp32LayerNorm after every conv block (wav2vec2.py:803-813)
    "layer_norm_first": True,         # pre-LN encoder layers (wav2vec2.py:1042-1062)
    "conv_bias": True,
    "activation_fn": "gelu",
}
_DIMS = {"encoder_layers": "layers", "encoder_embed_dim": "embed_dim", "encoder_ffn_embed_dim": "ffn_dim",
         "encoder_attention_heads": "heads", "conv_pos": "conv_pos", "conv_pos_groups": "conv_pos_groups"}
# train-mode stochastic paths (wav2vec2.py:598-599, :933, :941-942, :1053, :1059-1061, MultiheadAttention dropout)
STOCHASTIC_FIELDS = ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "dropout_input",
                     "dropout_features")


def validate_backbone_cfg(mcfg: dict, cfg: BackboneConfig) -> Dict[str, float]:
    """Compares the configuration a checkpoint embeds with the geometry this library is built for; raises
    SlsmiError listing every field that differs.  A field the file does not hold is not checked (fairseq's
    defaults then apply, and the state-dict shapes still have to match).  Returns the file's train-mode
    dropout / layerdrop probabilities (absent = 0)."""
    def norm(v):
        v = getattr(v, "name", v)   # a ChoiceEnum member carries its string as .name
        return v.lower() if isinstance(v, str) else v

    bad = []
    for k, want in _FIXED.items():
        if k in mcfg and norm(mcfg[k]) != want:
            bad.append(f"{k}={mcfg[k]!r} (built for {want!r})")
    for k, attr in _DIMS.items():
        if k in mcfg and int(mcfg[k]) != getattr(cfg, attr):
            bad.append(f"{k}={mcfg[k]!r} (built for {getattr(cfg, attr)})")
    if "conv_feature_layers" in mcfg:
        try:
            layers = _conv_layers(mcfg["conv_feature_layers"])
        except (ValueError, SyntaxError, TypeError) as e:
            bad.append(f"conv_feature_layers unreadable: {e}")
        else:
            if layers != tuple(cfg.conv_layers):
                bad.append(f"conv_feature_layers={layers} (built for {tuple(cfg.conv_layers)})")
    if bad:
        raise _lib.SlsmiError("the checkpoint's embedded configuration does not describe the architecture this "
                              "library implements: " + "; ".join(bad))
    return {k: float(mcfg.get(k) or 0.0) for k in STOCHASTIC_FIELDS}
