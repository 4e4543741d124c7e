"""CPU restatement of the reference scoring path -- TEST INFRASTRUCTURE ONLY.

This file is the parity oracle: a plain fp32 PyTorch/numpy restatement of the
algorithm the reference runs for one batch of utterances.  Nothing in the
product package may import it; only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py do, and only as the checker / reported baseline.

Pinning status
--------------
* Backbone + in-tree SAE head: pinned against the reference's own Python
  (wav2vec/wav2vec2.py, model.py, model_backup.py imported in the build
  container behind a fairseq stand-in, see tests/golden/make_golden.py) and
  cross-checked against an independent implementation (transformers'
  Wav2Vec2Model).  Fixtures: tests/golden/*.npz.
* fairseq.modules.MultiheadAttention / LayerNorm / SamePad are NOT in the
  reference tree (fairseq @ a54021305d6b3c4c5959ac9395135f63202db8f1 is an
  un-vendored dependency): their published algorithm is restated here and the
  reference holds no test vector for them -> "parity unpinned" at that
  boundary by the reference's own tests.
* SLS head beyond getAttenF (model_backup.py:186-202) is not in the reference
  tree; it is restated from the upstream SLS model definition -> "parity
  unpinned" by any file under the reference.

Every function cites the reference lines it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class OracleConfig:
    """Architecture constants (XLS-R-300M defaults; wav2vec2.py:39-252 field
    names in comments)."""

    conv_layers: Tuple[Tuple[int, int, int], ...] = (
        ((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2
    )  # conv_feature_layers, wav2vec2.py:97-98
    embed_dim: int = 1024  # encoder_embed_dim
    ffn_dim: int = 4096  # encoder_ffn_embed_dim
    heads: int = 16  # encoder_attention_heads
    layers: int = 24  # encoder_layers
    conv_pos: int = 128  # conv_pos
    conv_pos_groups: int = 16  # conv_pos_groups
    ln_eps: float = 1e-5
    sae_dict: int = 4096  # model.py:157
    sae_k: int = 128  # model.py:158
    sae_window: int = 1  # model_window_topk.py:282 (sae_window_size; 1 = the per-frame TopK of model.py)


# --------------------------------------------------------------------------
# data_utils_SSL.py:58-65
# --------------------------------------------------------------------------
def pad_audio(x: np.ndarray, max_len: int = 64600) -> np.ndarray:
    """Crop to max_len, or tile-repeat then crop (data_utils_SSL.py:58-65)."""
    n = x.shape[0]
    if n >= max_len:
        return x[:max_len]
    reps = int(max_len / n) + 1
    return np.tile(x, reps)[:max_len]


# --------------------------------------------------------------------------
# wav2vec2.py:523-538
# --------------------------------------------------------------------------
def conv_out_lengths(length: int, conv_layers: Sequence[Tuple[int, int, int]]) -> List[int]:
    out = []
    cur = length
    for _, k, s in conv_layers:
        cur = int(math.floor((cur - k) / s + 1))
        out.append(cur)
    return out


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# --------------------------------------------------------------------------
# wav2vec2.py:773-851 (layer_norm extractor mode, blocks :803-813)
# --------------------------------------------------------------------------
def feature_extractor(sd: Dict[str, Tensor], pfx: str, wav: Tensor, cfg: OracleConfig) -> Tensor:
    """[B, L] -> [B, C, T].  Conv1d(bias) -> LN over channels (fp32) -> erf-GELU."""
    x = wav.unsqueeze(1)
    for i, (_, _, stride) in enumerate(cfg.conv_layers):
        p = f"{pfx}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[p + "0.weight"], sd.get(p + "0.bias"), stride=stride)
        x = _ln(x.transpose(-2, -1).float(), sd[p + "2.1.weight"], sd[p + "2.1.bias"], cfg.ln_eps)
        x = F.gelu(x.transpose(-2, -1))
    return x


# --------------------------------------------------------------------------
# fairseq.modules.MultiheadAttention (external; call site wav2vec2.py:1009-1014,
# :1046-1052).  Published algorithm: separate q/k/v/out projections with bias,
# q scaled by head_dim**-0.5 AFTER the projection, softmax in fp32, no dropout
# in eval.
# --------------------------------------------------------------------------
def self_attention(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int,
                   key_padding_mask: Optional[Tensor] = None, attn_drop: Optional[Tensor] = None) -> Tensor:
    """x: [T, B, C] -> [T, B, C].  attn_drop: train mode only -- the multipliers (0 or 1 / keep probability)
    [B, heads, T, T] that MultiheadAttention's dropout (attention_dropout, wav2vec2.py:1012) would draw for the
    attention probabilities."""
    T, B, C = x.shape
    hd = C // heads
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    q = q.contiguous().view(T, B * heads, hd).transpose(0, 1)
    k = k.contiguous().view(T, B * heads, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * heads, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))  # [B*H, T, T]
    if key_padding_mask is not None:
        w = w.view(B, heads, T, T).masked_fill(
            key_padding_mask[:, None, None, :].to(torch.bool), float("-inf")).view(B * heads, T, T)
    w = F.softmax(w.float(), dim=-1).type_as(w)
    if attn_drop is not None:
        w = (w.view(B, heads, T, T) * attn_drop).view(B * heads, T, T)
    a = torch.bmm(w, v)  # [B*H, T, hd]
    a = a.transpose(0, 1).contiguous().view(T, B, C)
    return F.linear(a, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


# --------------------------------------------------------------------------
# wav2vec2.py:1042-1062 (pre-LN branch)
# --------------------------------------------------------------------------
def encoder_layer(sd: Dict[str, Tensor], p: str, x: Tensor, cfg: OracleConfig,
                  key_padding_mask: Optional[Tensor] = None,
                  drop: Optional[Tuple[Optional[Tensor], ...]] = None) -> Tensor:
    """drop: train mode only -- the multipliers (0 or 1 / keep probability) nn.Dropout would draw for dropout1,
    dropout2 and dropout3 (wav2vec2.py:1053, :1059, :1061), each [T, B, C'] or None, and optionally a fourth entry,
    those of the attention probabilities [B, heads, T, T] (attention_dropout)."""
    d1, d2, d3 = drop[:3] if drop is not None else (None, None, None)
    da = drop[3] if drop is not None and len(drop) > 3 else None
    r = x
    h = _ln(x, sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], cfg.ln_eps)
    h = self_attention(sd, p + "self_attn.", h, cfg.heads, key_padding_mask, da)
    if d1 is not None:
        h = h * d1
    x = r + h
    r = x
    h = _ln(x, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], cfg.ln_eps)
    h = F.gelu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"]).float()).type_as(h)
    if d2 is not None:
        h = h * d2
    h = F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    if d3 is not None:
        h = h * d3
    return r + h


def pos_conv_weight(sd: Dict[str, Tensor], p: str) -> Tensor:
    """nn.utils.weight_norm(conv, name='weight', dim=2) (wav2vec2.py:874):
    w = g * v / ||v|| with the norm taken over every dim except dim 2."""
    g = sd[p + "weight_g"]
    v = sd[p + "weight_v"]
    return v * (g / v.norm(2, dim=(0, 1), keepdim=True))


# --------------------------------------------------------------------------
# wav2vec2.py:901-972 (TransformerEncoder.forward / extract_features)
# --------------------------------------------------------------------------
def encoder(sd: Dict[str, Tensor], pfx: str, x: Tensor, cfg: OracleConfig,
            padding_mask: Optional[Tensor] = None, train_masks: Optional[Dict[str, Tensor]] = None
            ) -> Tuple[Tensor, List[Tensor]]:
    """x: [B, T, C] -> (x after final LN [B, T, C], list of per-layer outputs
    [T, B, C] taken BEFORE the final LN, wav2vec2.py:958).
    train_masks: the random draws of the backbone's train mode, made by the caller (all optional, [B, T, C'] frame
    order): "encoder" F.dropout after the positional conv (:933); "attn_out", "ffn_act", "ffn_out" [layers, B, T, C']
    dropout1 / 2 / 3 of every layer; "attn" [layers, B, heads, T, T] the attention probabilities' dropout;
    "layer_keep" [layers] booleans, False = the layer is skipped by layerdrop and
    appends nothing to the layer results (:941-958)."""
    tm = train_masks or {}
    p = pfx + "encoder."
    if padding_mask is not None:
        x = x.masked_fill(padding_mask.unsqueeze(-1), 0.0)  # :912-913
    w = pos_conv_weight(sd, p + "pos_conv.0.")
    k = cfg.conv_pos
    xc = F.conv1d(x.transpose(1, 2), w, sd[p + "pos_conv.0.bias"], padding=k // 2,
                  groups=cfg.conv_pos_groups)
    if k % 2 == 0:  # SamePad: drop the last frame for an even kernel
        xc = xc[:, :, :-1]
    x = x + F.gelu(xc).transpose(1, 2)  # :915-917
    if tm.get("encoder") is not None:
        x = x * tm["encoder"]  # :933
    x = x.transpose(0, 1)  # :936
    outs: List[Tensor] = []
    for l in range(cfg.layers):
        if tm.get("layer_keep") is not None and not bool(tm["layer_keep"][l]):
            continue
        drop = tuple(tm[k][l].transpose(0, 1) if tm.get(k) is not None else None
                     for k in ("attn_out", "ffn_act", "ffn_out"))
        drop = drop + (tm["attn"][l] if tm.get("attn") is not None else None,)
        x = encoder_layer(sd, f"{p}layers.{l}.", x, cfg, padding_mask, drop)
        outs.append(x)
    x = x.transpose(0, 1)
    x = _ln(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.ln_eps)  # :905-906
    return x, outs


# --------------------------------------------------------------------------
# wav2vec2.py:540-647 with mask=False, features_only=True, padding_mask=None
# --------------------------------------------------------------------------
def frame_padding_mask(lengths: Sequence[int], frames: int, cfg: OracleConfig) -> Optional[Tensor]:
    """wav2vec2.py:567-586: utterance b keeps its first conv_out_lengths(lengths[b]) frames; True marks
    a padded frame.  None when nothing is padded."""
    mask = torch.zeros(len(lengths), frames, dtype=torch.bool)
    for b, n in enumerate(lengths):
        mask[b, conv_out_lengths(int(n), cfg.conv_layers)[-1]:] = True
    return mask if bool(mask.any()) else None


def backbone_forward(sd: Dict[str, Tensor], wav: Tensor, cfg: OracleConfig,
                     pfx: str = "", lengths: Optional[Sequence[int]] = None,
                     train_masks: Optional[Dict[str, Tensor]] = None) -> Dict[str, object]:
    """lengths: sample counts of the utterances of a padded batch (the reference's padding_mask
    argument holds True past them, wav2vec2.py:540-547, :567-586).
    train_masks: see encoder(); also "input" [B, T, C]: dropout_input on the projected features (:598).
    (dropout_features, :599, acts on a copy that only the pre-training loss reads.)"""
    feats = feature_extractor(sd, pfx, wav, cfg)  # [B, C, T]
    feats = feats.transpose(1, 2)
    feats = _ln(feats, sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"], cfg.ln_eps)  # :563-564
    mask = frame_padding_mask(lengths, feats.shape[1], cfg) if lengths is not None else None
    x = F.linear(feats, sd[pfx + "post_extract_proj.weight"], sd[pfx + "post_extract_proj.bias"])  # :595-596
    if train_masks is not None and train_masks.get("input") is not None:
        x = x * train_masks["input"]  # :598
    x, layer_results = encoder(sd, pfx, x, cfg, mask, train_masks)
    return {"x": x, "features": feats, "layer_results": layer_results, "padding_mask": mask}


# --------------------------------------------------------------------------
# model_backup.py:186-202
# --------------------------------------------------------------------------
def get_atten_f(layer_results: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
    """per-layer [T,B,C] -> (layery [B, nl, C] = time means,
    fullfeature [B, nl, T, C])."""
    pooled, full = [], []
    for x in layer_results:
        xb = x.transpose(0, 1)  # [B, T, C]
        pooled.append(xb.mean(dim=1, keepdim=True))
        full.append(xb.unsqueeze(1))
    return torch.cat(pooled, dim=1), torch.cat(full, dim=1)


def sls_head(sd: Dict[str, Tensor], layer_results: Sequence[Tensor], pfx: str = "",
             bn_eps: float = 1e-5, training: bool = False) -> Tensor:
    """Upstream SLS head (not in the reference tree; restated): sigmoid(fc0)
    layer weights -> weighted layer sum -> BatchNorm2d(1) in eval -> SELU ->
    max_pool2d(3,3) -> flatten -> fc1 -> SELU -> fc3 -> SELU -> log_softmax.
    training: BatchNorm2d on the batch's own statistics, as in model.train() (running statistics untouched)."""
    y0, full = get_atten_f(layer_results)
    w = torch.sigmoid(F.linear(y0, sd[pfx + "fc0.weight"], sd[pfx + "fc0.bias"]))  # [B, nl, 1]
    full = (full * w.unsqueeze(-1)).sum(dim=1).unsqueeze(1)  # [B,1,T,C]
    if training:
        x = F.batch_norm(full, None, None, sd[pfx + "first_bn.weight"], sd[pfx + "first_bn.bias"], True, 0.0, bn_eps)
    else:
        x = F.batch_norm(full, sd[pfx + "first_bn.running_mean"], sd[pfx + "first_bn.running_var"],
                         sd[pfx + "first_bn.weight"], sd[pfx + "first_bn.bias"], False, 0.0, bn_eps)
    x = F.selu(x)
    x = F.max_pool2d(x, (3, 3))
    x = torch.flatten(x, 1)
    x = F.selu(F.linear(x, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
    x = F.selu(F.linear(x, sd[pfx + "fc3.weight"], sd[pfx + "fc3.bias"]))
    return F.log_softmax(x, dim=1)


# --------------------------------------------------------------------------
# model.py:68-89 (AutoEncoderTopK) and model.py:215-247 (pool + classifier)
# --------------------------------------------------------------------------
def sae_encode(sd: Dict[str, Tensor], x_flat: Tensor, k: int, pfx: str = "sae.") -> Tensor:
    acts = F.relu(F.linear(x_flat - sd[pfx + "b_dec"], sd[pfx + "encoder.weight"], sd[pfx + "encoder.bias"]))
    vals, idx = acts.topk(k, sorted=False, dim=-1)
    return torch.zeros_like(acts).scatter_(dim=-1, index=idx, src=vals)


def window_topk(acts: Tensor, k: int, window_size: int) -> Tensor:
    """model_window_topk.py:118-203 (_window_topk) on non-negative activations [B, T, D]: windows of
    `window_size` frames every window_size // 2 frames; each window keeps the k features with the
    largest activation sums; a frame's vote for a feature is its activation times the number of
    covering windows that kept the feature; each frame then keeps its k largest votes.
    Where fewer than k votes of a frame are positive the reference's topk fills up with zero-vote
    features in an order its topk implementation happens to produce; this restatement calls the same
    torch.topk so that it agrees with the reference run on the same torch build."""
    B, T, D = acts.shape
    stride = max(1, window_size // 2)
    if stride >= T:
        num_windows, pad = 1, 0
    else:
        num_windows = (T - window_size) // stride + 1
        pad = max(0, (num_windows - 1) * stride + window_size - T)
    x = F.pad(acts, (0, 0, 0, pad)) if pad > 0 else acts
    Tp = T + pad
    windows = x.transpose(1, 2).unfold(2, window_size, stride).permute(0, 2, 3, 1)  # [B, nW, w, D]
    sums = windows.sum(dim=2)
    _, idx = sums.topk(k, dim=-1, sorted=False)
    keep = torch.zeros_like(sums).scatter_(dim=-1, index=idx, value=1.0)
    votes = torch.zeros(B, Tp, D)
    for i in range(num_windows):
        votes[:, i * stride:i * stride + window_size, :] += windows[:, i] * keep[:, i:i + 1, :]
    mask = torch.zeros_like(votes)
    for t in range(Tp):
        _, ti = votes[:, t, :].topk(k, dim=-1, sorted=False)
        mask[:, t, :].scatter_(dim=-1, index=ti, value=1.0)
    return (x * mask)[:, :T, :]


def window_votes(acts: Tensor, k: int, window_size: int) -> Tensor:
    """The vote tensor of window_topk [B, T, D] (model_window_topk.py:171-185): a frame's activation
    times the number of covering windows that kept the feature."""
    B, T, D = acts.shape
    stride = max(1, window_size // 2)
    num_windows = 1 if stride >= T else (T - window_size) // stride + 1
    pad = 0 if stride >= T else max(0, (num_windows - 1) * stride + window_size - T)
    x = F.pad(acts, (0, 0, 0, pad)) if pad > 0 else acts
    windows = x.transpose(1, 2).unfold(2, window_size, stride).permute(0, 2, 3, 1)
    sums = windows.sum(dim=2)
    _, idx = sums.topk(k, dim=-1, sorted=False)
    keep = torch.zeros_like(sums).scatter_(dim=-1, index=idx, value=1.0)
    votes = torch.zeros(B, T + pad, D)
    for i in range(num_windows):
        votes[:, i * stride:i * stride + window_size, :] += windows[:, i] * keep[:, i:i + 1, :]
    return votes[:, :T]


def window_vote_counts(acts: Tensor, k: int, window_size: int) -> Tensor:
    """Number of positive votes per frame [B, T] in window_topk: where it is >= k the kept set is
    determined by the votes alone (up to exact ties between positive votes)."""
    return (window_votes(acts, k, window_size) > 0).sum(-1)


def sae_head(sd: Dict[str, Tensor], x: Tensor, cfg: OracleConfig, pfx: str = "",
             return_sae_loss: bool = False, use_sae: bool = True, use_sparse_features: bool = True,
             return_code: bool = False, dropout_mask: Optional[Tensor] = None):
    """model.py:215-247: with use_sae the classifier pools the TopK code or, with
    use_sparse_features=False, its reconstruction; without it the encoder output.
    dropout_mask: None (eval), or the [B, 256] multipliers (0 or 1 / 0.7) nn.Dropout(0.3) of the
    classifier (model.py:187) drew in train mode."""
    B, T, C = x.shape
    x_flat = x.reshape(B * T, C)
    loss, enc = None, None
    feats = x
    if use_sae:
        if cfg.sae_window > 1:  # model_window_topk.py:345-349, :96-105
            acts = F.relu(F.linear(x_flat - sd[pfx + "sae.b_dec"], sd[pfx + "sae.encoder.weight"],
                                   sd[pfx + "sae.encoder.bias"]))
            enc = window_topk(acts.reshape(B, T, -1), cfg.sae_k, cfg.sae_window).reshape(B * T, -1)
        else:
            enc = sae_encode(sd, x_flat, cfg.sae_k, pfx + "sae.")
        recon = None
        if return_sae_loss or not use_sparse_features:
            recon = F.linear(enc, sd[pfx + "sae.decoder.weight"]) + sd[pfx + "sae.b_dec"]
        if return_sae_loss:
            loss = F.mse_loss(recon, x_flat)
        feats = enc.reshape(B, T, -1) if use_sparse_features else recon.reshape(B, T, C)
    pooled = feats.mean(dim=1)  # AdaptiveAvgPool1d(1), model.py:245
    h = _ln(pooled, sd[pfx + "classifier.0.weight"], sd[pfx + "classifier.0.bias"], 1e-5)
    h = F.relu(F.linear(h, sd[pfx + "classifier.1.weight"], sd[pfx + "classifier.1.bias"]))
    if dropout_mask is not None:
        h = h * dropout_mask
    logits = F.linear(h, sd[pfx + "classifier.4.weight"], sd[pfx + "classifier.4.bias"])
    out = F.log_softmax(logits, dim=-1)
    if return_code:
        return out, loss, (enc.reshape(B, T, -1) if enc is not None else None)
    return (out, loss) if return_sae_loss else out


def interpretability_info(code: Tensor) -> Dict[str, Tensor]:
    """Model.get_interpretability_info (model.py:262-293) on the sparse code [B, T, D]."""
    avg = code.mean(dim=1)
    vals, idx = avg.topk(k=min(20, code.shape[-1]), dim=-1)
    active = (code > 0).float()
    return {"avg_activation": avg, "top20_features": idx, "top20_values": vals,
            "sparsity": active.mean(dim=[1, 2]), "activation_freq": active.mean(dim=1), "sparse_features": code}


@torch.no_grad()
def model_forward(sd: Dict[str, Tensor], wav: Tensor, cfg: OracleConfig, head: str = "sls",
                  return_intermediates: bool = False):
    """Model.forward in eval mode (model.py:195-260): [B, L] or [B, L, 1] fp32
    -> log-probs [B, 2]."""
    if wav.ndim == 3:
        wav = wav[:, :, 0]  # model.py:134-135
    bb = backbone_forward(sd, wav.float(), cfg, pfx="ssl_model.model.")
    if head == "sls":
        logp = sls_head(sd, bb["layer_results"])
    elif head == "sae":
        logp = sae_head(sd, bb["x"], cfg)
    elif head == "sae_recon":
        logp = sae_head(sd, bb["x"], cfg, use_sparse_features=False)
    elif head == "ssl":
        logp = sae_head(sd, bb["x"], cfg, use_sae=False)
    else:
        raise ValueError(head)
    return (logp, bb) if return_intermediates else logp


# --------------------------------------------------------------------------
# main.py:158-199
# --------------------------------------------------------------------------
def scores_from_logp(logp: Tensor) -> np.ndarray:
    """exp(log-prob)[:, 1] as float32 (main.py:183-184)."""
    return torch.exp(logp)[:, 1].detach().cpu().numpy().ravel()


def format_score_lines(utt_ids: Sequence[str], scores: np.ndarray) -> List[str]:
    """'{utt} {score}\\n' with the Python float repr of the float32 value
    (main.py:188-192: .tolist() then str.format)."""
    return ["{} {}\n".format(u, s) for u, s in zip(utt_ids, scores.tolist())]


# --------------------------------------------------------------------------
# eval_metrics_DF.py:21-48
# --------------------------------------------------------------------------
def compute_det_curve(target_scores: np.ndarray, nontarget_scores: np.ndarray):
    n = target_scores.size + nontarget_scores.size
    all_scores = np.concatenate((target_scores, nontarget_scores))
    labels = np.concatenate((np.ones(target_scores.size), np.zeros(nontarget_scores.size)))
    order = np.argsort(all_scores, kind="mergesort")
    labels = labels[order]
    tar_sums = np.cumsum(labels)
    non_sums = nontarget_scores.size - (np.arange(1, n + 1) - tar_sums)
    frr = np.concatenate((np.atleast_1d(0), tar_sums / target_scores.size))
    far = np.concatenate((np.atleast_1d(1), non_sums / nontarget_scores.size))
    thr = np.concatenate((np.atleast_1d(all_scores[order[0]] - 0.001), all_scores[order]))
    return frr, far, thr


def compute_eer(target_scores: np.ndarray, nontarget_scores: np.ndarray):
    frr, far, thr = compute_det_curve(target_scores, nontarget_scores)
    i = int(np.argmin(np.abs(frr - far)))
    return float(np.mean((frr[i], far[i]))), float(thr[i])
