#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference's own code.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's wav2vec/wav2vec2.py, model.py, model_backup.py, eval_metrics_DF.py
and data_utils_SSL.py, runs them on seeded inputs and stores inputs + outputs
as small .npz / .json fixtures.  Nothing from the reference's source text is
stored.

fairseq is an un-vendored dependency of the reference (fairseq @
a54021305d6b3c4c5959ac9395135f63202db8f1) and is absent here, so the names
wav2vec2.py imports from it are provided by a small in-memory stand-in that
restates their published behaviour with stock PyTorch modules
(MultiheadAttention -> F.multi_head_attention_forward with separate projection
weights, LayerNorm -> nn.LayerNorm, SamePad -> drop the last frame for even
kernels, ...).  Results at that boundary are therefore pinned by PyTorch's MHA
and, independently, by transformers' Wav2Vec2Model (cross-check below), not by
fairseq itself.

Usage:  python tests/golden/make_golden.py [--skip-xlsr]
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import slsmi  # noqa: E402
from slsmi.config import BackboneConfig, XLSR_300M  # noqa: E402
from slsmi.weights import synth_state_dict, BACKBONE_PREFIX  # noqa: E402
from oracle import reference_cpu as orc  # noqa: E402


# ---------------------------------------------------------------- fairseq stand-in
def install_fairseq_standin():
    from dataclasses import dataclass

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    fs = mod("fairseq")
    utils = mod("fairseq.utils")
    data = mod("fairseq.data")
    data_utils = mod("fairseq.data.data_utils")
    dc = mod("fairseq.dataclass")
    models = mod("fairseq.models")
    modules = mod("fairseq.modules")
    ca = mod("fairseq.modules.checkpoint_activations")
    tse = mod("fairseq.modules.transformer_sentence_encoder")
    dist = mod("fairseq.distributed")
    cu = mod("fairseq.checkpoint_utils")
    fs.utils, fs.data, fs.dataclass, fs.models, fs.modules = utils, data, dc, models, modules
    fs.distributed, fs.checkpoint_utils = dist, cu
    data.data_utils = data_utils
    modules.checkpoint_activations, modules.transformer_sentence_encoder = ca, tse

    def get_activation_fn(name):
        if name == "gelu":
            return lambda x: F.gelu(x.float()).type_as(x)
        if name == "relu":
            return F.relu
        raise NotImplementedError(name)

    utils.get_activation_fn = get_activation_fn
    utils.get_available_activation_fns = lambda: ["relu", "gelu", "gelu_fast", "gelu_accurate", "tanh", "linear"]
    utils.buffered_arange = lambda n: torch.arange(n)
    utils.is_xla_tensor = lambda t: False

    def index_put(t, mask, value):
        t[mask] = value
        return t

    utils.index_put = index_put
    data_utils.compute_mask_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())

    def ChoiceEnum(choices):
        return str

    @dataclass
    class FairseqDataclass:
        pass

    dc.ChoiceEnum, dc.FairseqDataclass = ChoiceEnum, FairseqDataclass
    models.BaseFairseqModel = nn.Module
    models.register_model = lambda name, dataclass=None: (lambda cls: cls)

    class TransposeLast(nn.Module):
        def forward(self, x):
            return x.transpose(-2, -1)

    class Fp32LayerNorm(nn.LayerNorm):
        def forward(self, x):
            out = F.layer_norm(x.float(), self.normalized_shape,
                               self.weight.float() if self.weight is not None else None,
                               self.bias.float() if self.bias is not None else None, self.eps)
            return out.type_as(x)

    class Fp32GroupNorm(nn.GroupNorm):
        pass

    class SamePad(nn.Module):
        def __init__(self, kernel_size, causal=False):
            super().__init__()
            self.remove = 1 if kernel_size % 2 == 0 else 0

        def forward(self, x):
            return x[:, :, :-self.remove] if self.remove > 0 else x

    class GradMultiply(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, scale):
            ctx.scale = scale
            return x.new(x)

        @staticmethod
        def backward(ctx, g):
            return g * ctx.scale, None

    class MultiheadAttention(nn.Module):
        """Self-attention with separate q/k/v/out projections (fairseq layout:
        time-major input, returns (out, head-averaged weights))."""

        def __init__(self, embed_dim, num_heads, dropout=0.0, self_attention=False, **kw):
            super().__init__()
            self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
            self.k_proj = nn.Linear(embed_dim, embed_dim)
            self.v_proj = nn.Linear(embed_dim, embed_dim)
            self.q_proj = nn.Linear(embed_dim, embed_dim)
            self.out_proj = nn.Linear(embed_dim, embed_dim)

        def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None, **kw):
            return F.multi_head_attention_forward(
                query, key, value, self.embed_dim, self.num_heads,
                torch.empty([0]), torch.cat((self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)),
                None, None, False, self.dropout, self.out_proj.weight, self.out_proj.bias,
                self.training, key_padding_mask, need_weights, attn_mask,
                use_separate_proj_weight=True, q_proj_weight=self.q_proj.weight,
                k_proj_weight=self.k_proj.weight, v_proj_weight=self.v_proj.weight)

    class GumbelVectorQuantizer(nn.Module):
        """Parameter container only (never called with mask=False, features_only=True)."""

        def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first,
                     weight_proj_depth=1, weight_proj_factor=1, **kw):
            super().__init__()
            self.vars = nn.Parameter(torch.zeros(1, groups * num_vars, vq_dim // groups))
            self.weight_proj = nn.Linear(dim, groups * num_vars)

    modules.Fp32GroupNorm, modules.Fp32LayerNorm, modules.GradMultiply = Fp32GroupNorm, Fp32LayerNorm, GradMultiply
    modules.GumbelVectorQuantizer, modules.LayerNorm = GumbelVectorQuantizer, nn.LayerNorm
    modules.MultiheadAttention, modules.SamePad, modules.TransposeLast = MultiheadAttention, SamePad, TransposeLast
    ca.checkpoint_wrapper = lambda m, *a, **k: m
    tse.init_bert_params = lambda m: None
    dist.fsdp_wrap = lambda m, *a, **k: m
    cu.load_model_ensemble_and_task = lambda *a, **k: (_ for _ in ()).throw(
        RuntimeError("no fairseq checkpoint loader in the fixture generator"))


def load_ref_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    install_fairseq_standin()
    pkg = types.ModuleType("wav2vec")
    pkg.__path__ = [os.path.join(REF, "wav2vec")]
    sys.modules["wav2vec"] = pkg
    load_ref_module("wav2vec.utils", os.path.join(REF, "wav2vec", "utils.py"))
    w2v = load_ref_module("wav2vec.wav2vec2", os.path.join(REF, "wav2vec", "wav2vec2.py"))
    ref_model = load_ref_module("ref_model", os.path.join(REF, "model.py"))
    ref_backup = load_ref_module("ref_model_backup", os.path.join(REF, "model_backup.py"))
    ref_eer = load_ref_module("ref_eval_metrics_DF", os.path.join(REF, "eval_metrics_DF.py"))
    return w2v, ref_model, ref_backup, ref_eer


def ref_backbone(w2v, cfg: BackboneConfig):
    rc = w2v.Wav2Vec2Config(
        extractor_mode="layer_norm", encoder_layers=cfg.layers, encoder_embed_dim=cfg.embed_dim,
        encoder_ffn_embed_dim=cfg.ffn_dim, encoder_attention_heads=cfg.heads, layer_norm_first=True,
        conv_bias=True, final_dim=cfg.final_dim, dropout=0.0, attention_dropout=0.0,
        activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0,
        conv_feature_layers=str([tuple(c) for c in cfg.conv_layers]), conv_pos=cfg.conv_pos,
        conv_pos_groups=cfg.conv_pos_groups, latent_vars=cfg.latent_vars, latent_groups=cfg.latent_groups,
        quantize_targets=True)
    return w2v.Wav2Vec2Model(rc).eval()


def build_ref_model(w2v, ref_model, cfg, sd, head, sae_dict, sae_k):
    """Reference Model with its SSLModel replaced by one that wraps the
    shim-built backbone (SSLModel.__init__ needs the fairseq checkpoint loader)."""
    backbone = ref_backbone(w2v, cfg)

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
        m = ref_model.Model(None, "cpu", use_sae=True, use_sparse_features=True,
                            sae_dict_size=sae_dict, sae_k=sae_k)
    finally:
        ref_model.SSLModel = orig
    if cfg.embed_dim != 1024:  # the reference hard-codes activation_dim=1024 (model.py:173)
        m.sae = ref_model.AutoEncoderTopK(cfg.embed_dim, sae_dict, sae_k)
    if head == "sae":
        missing = m.load_state_dict(sd, strict=True)
    else:
        bb = {k[len(BACKBONE_PREFIX):]: v for k, v in sd.items() if k.startswith(BACKBONE_PREFIX)}
        missing = backbone.load_state_dict(bb, strict=True)
    return m.eval(), backbone, missing


def stats(t: torch.Tensor):
    return [float(t.mean()), float(t.abs().max()), float(t.std())]


@torch.no_grad()
def run_case(w2v, ref_model, ref_backup, cfg, L, B, seed, sae_dict, sae_k, full: bool):
    T = cfg.num_frames(L)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    out = {"wav": wav.numpy()} if full else {}
    ocfg = orc.OracleConfig(conv_layers=cfg.conv_layers, embed_dim=cfg.embed_dim, ffn_dim=cfg.ffn_dim,
                            heads=cfg.heads, layers=cfg.layers, conv_pos=cfg.conv_pos,
                            conv_pos_groups=cfg.conv_pos_groups, sae_dict=sae_dict, sae_k=sae_k)
    report = {}
    # ---- SAE head: the in-tree Model end to end
    sd_sae = synth_state_dict(cfg, "sae", T, seed, sae_dict, sae_k)
    m, backbone, _ = build_ref_model(w2v, ref_model, cfg, sd_sae, "sae", sae_dict, sae_k)
    logp_sae = m(wav, return_sae_loss=False)
    logp_sae2, sae_loss = m(wav)  # default return_sae_loss=True
    res = backbone(wav, mask=False, features_only=True)
    x_ref = res["x"]
    layers_ref = [lr[0] for lr in res["layer_results"]]
    # ---- SLS head: reference getAttenF + restated upstream tail
    sd_sls = synth_state_dict(cfg, "sls", T, seed, sae_dict, sae_k)
    y0, full_f = ref_backup.getAttenF(res["layer_results"])
    y0o, fullo = orc.get_atten_f(layers_ref)
    report["getAttenF_layery_maxabs"] = float((y0 - y0o).abs().max())
    report["getAttenF_full_maxabs"] = float((full_f - fullo).abs().max())
    logp_sls_ref_layers = orc.sls_head(sd_sls, layers_ref)
    # ---- oracle restatement on the same inputs
    o_sae, bb = orc.model_forward(sd_sae, wav, ocfg, "sae", return_intermediates=True)
    o_sae_l, o_loss = orc.sae_head(sd_sae, bb["x"], ocfg, return_sae_loss=True)
    o_sls = orc.model_forward(sd_sls, wav, ocfg, "sls")
    report["x_maxabs"] = float((bb["x"] - x_ref).abs().max())
    report["x_absmax_ref"] = float(x_ref.abs().max())
    report["layers_maxabs"] = max(float((a - b).abs().max()) for a, b in zip(bb["layer_results"], layers_ref))
    report["logp_sae_maxabs"] = float((o_sae - logp_sae).abs().max())
    report["logp_sls_maxabs"] = float((o_sls - logp_sls_ref_layers).abs().max())
    report["sae_loss_ref"] = float(sae_loss)
    report["sae_loss_oracle"] = float(o_loss)
    out.update({
        "logp_sae": logp_sae.numpy(), "logp_sls": logp_sls_ref_layers.numpy(),
        "sae_loss": np.float32(sae_loss),
        "layer_stats": np.array([stats(t) for t in layers_ref], dtype=np.float64),
        "layer_first8": np.stack([t[0, :, :8].numpy() for t in layers_ref]),
        "x_slice": x_ref[:, ::50, :8].numpy(),
        "conv_lengths": np.array(cfg.conv_out_lengths(L)),
        "seed": np.int64(seed), "B": np.int64(B), "L": np.int64(L),
    })
    if full:
        out["x"] = x_ref.numpy()
        out["layer_results"] = np.stack([t.numpy() for t in layers_ref])
        out["features"] = res["features"].numpy()
    return out, report, (backbone, sd_sae, wav, x_ref, layers_ref)


@torch.no_grad()
def hf_crosscheck(cfg, backbone_sd, wav, x_ref, layers_ref):
    """Independent implementation check: transformers' Wav2Vec2Model loaded with
    the same weights through a pure key-rename map."""
    from transformers import Wav2Vec2Config as HC, Wav2Vec2Model as HM
    hc = HC(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
            intermediate_size=cfg.ffn_dim, feat_extract_norm="layer", conv_bias=True,
            conv_dim=[c[0] for c in cfg.conv_layers], conv_kernel=[c[1] for c in cfg.conv_layers],
            conv_stride=[c[2] for c in cfg.conv_layers], do_stable_layer_norm=True,
            num_conv_pos_embeddings=cfg.conv_pos, num_conv_pos_embedding_groups=cfg.conv_pos_groups,
            hidden_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, activation_dropout=0.0,
            layerdrop=0.0, hidden_act="gelu", apply_spec_augment=False)
    hm = HM(hc).eval()
    ren = {}
    for k, v in backbone_sd.items():
        n = k
        n = n.replace("feature_extractor.conv_layers.", "feature_extractor.conv_layers.")
        n = n.replace(".0.weight", ".conv.weight").replace(".0.bias", ".conv.bias") if "conv_layers" in n else n
        n = n.replace(".2.1.", ".layer_norm.")
        if n.startswith("layer_norm."):
            n = "feature_projection." + n
        n = n.replace("post_extract_proj.", "feature_projection.projection.")
        n = n.replace("encoder.pos_conv.0.weight_g", "encoder.pos_conv_embed.conv.parametrizations.weight.original0")
        n = n.replace("encoder.pos_conv.0.weight_v", "encoder.pos_conv_embed.conv.parametrizations.weight.original1")
        n = n.replace("encoder.pos_conv.0.bias", "encoder.pos_conv_embed.conv.bias")
        n = n.replace("self_attn_layer_norm", "layer_norm").replace("self_attn.", "attention.")
        n = n.replace(".fc1.", ".feed_forward.intermediate_dense.").replace(".fc2.", ".feed_forward.output_dense.")
        ren[n] = v
    skip = ("mask_emb", "quantizer", "project_q", "final_proj")
    ren = {k: v for k, v in ren.items() if not k.startswith(skip)}
    r = hm.load_state_dict(ren, strict=False)
    unexpected = [k for k in r.unexpected_keys]
    missing = [k for k in r.missing_keys if "masked_spec_embed" not in k]
    o = hm(wav, output_hidden_states=True)
    rep = {"hf_unexpected": unexpected, "hf_missing": missing,
           "hf_x_maxabs": float((o.last_hidden_state - x_ref).abs().max())}
    for i in (0, cfg.layers // 2, cfg.layers - 2):
        if i + 1 < len(o.hidden_states) - 1 or cfg.layers == 1:
            rep[f"hf_layer{i}_maxabs"] = float(
                (o.hidden_states[i + 1] - layers_ref[i].transpose(0, 1)).abs().max())
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-xlsr", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, ref_model, ref_backup, ref_eer = load_reference()
    report = {}

    # 1. tiny configuration: full tensors
    tiny = BackboneConfig(conv_layers=((32, 10, 5),) + ((32, 3, 2),) * 4 + ((32, 2, 2),) * 2,
                          embed_dim=64, ffn_dim=128, heads=4, layers=2, conv_pos=16, conv_pos_groups=4,
                          final_dim=32, latent_vars=8, latent_groups=2)
    out, rep, ctx = run_case(w2v, ref_model, ref_backup, tiny, L=3200, B=2, seed=1234,
                             sae_dict=256, sae_k=16, full=True)
    rep.update(hf_crosscheck(tiny, {k[len(BACKBONE_PREFIX):]: v for k, v in ctx[1].items()
                                    if k.startswith(BACKBONE_PREFIX)}, ctx[2], ctx[3], ctx[4]))
    np.savez_compressed(os.path.join(HERE, "tiny_case.npz"), **out)
    report["tiny"] = rep
    print("tiny:", json.dumps(rep, indent=1))

    # 2. XLS-R-300M geometry, seeded weights (weights are NOT stored: the
    #    generator in slsmi.weights reproduces them from the seed)
    if not a.skip_xlsr:
        for nl, name in ((2, "xlsr_2layer_case"), (24, "xlsr_case")):
            cfg = BackboneConfig(layers=nl)
            out, rep, ctx = run_case(w2v, ref_model, ref_backup, cfg, L=64600, B=2, seed=1234,
                                     sae_dict=4096, sae_k=128, full=False)
            if nl == 24:
                rep.update(hf_crosscheck(cfg, {k[len(BACKBONE_PREFIX):]: v for k, v in ctx[1].items()
                                               if k.startswith(BACKBONE_PREFIX)}, ctx[2], ctx[3], ctx[4]))
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            report[name] = rep
            print(name, json.dumps(rep, indent=1))

    # 3. EER known answer from the reference scorer (eval_metrics_DF.py:42-48)
    rng = np.random.default_rng(0)
    bona = rng.normal(1, 1, 1000)
    spoof = rng.normal(-1, 1, 9000)
    eer, thr = ref_eer.compute_eer(bona, spoof)
    o_eer, o_thr = orc.compute_eer(bona, spoof)
    report["eer"] = {"ref": [float(eer), float(thr)], "oracle": [o_eer, o_thr]}
    with open(os.path.join(HERE, "eer_known_answer.json"), "w") as f:
        json.dump({"generator": "np.random.default_rng(0); bona=normal(1,1,1000); spoof=normal(-1,1,9000)",
                   "eer": float(eer), "threshold": float(thr)}, f, indent=1)

    # 4. pad() cases (data_utils_SSL.py:58-65); librosa is only used inside the
    #    Dataset classes, so an empty placeholder module lets the file import.
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    sys.path.insert(0, REF)
    du = load_ref_module("ref_data_utils_SSL", os.path.join(REF, "data_utils_SSL.py"))
    rng = np.random.default_rng(1)
    cases = {}
    for n in (70000, 64600, 30000, 7, 64599):
        x = rng.standard_normal(n).astype(np.float32)
        y = du.pad(x, 64600)
        assert np.array_equal(y, orc.pad_audio(x, 64600))
        cases[f"in_{n}"] = x[:64] if n > 64 else x
        cases[f"out_{n}_head"] = y[:64]
        cases[f"out_{n}_tail"] = y[-64:]
        cases[f"out_{n}_sum"] = np.float64(y.astype(np.float64).sum())
        cases[f"seed_{n}"] = np.int64(1)
    np.savez_compressed(os.path.join(HERE, "pad_cases.npz"), **cases)

    # 5. score-line formatting (main.py:188-192)
    sc = np.array([0.98765433, 1e-8, 0.5, 1.0, 3.4e-5], dtype=np.float32)
    lines = ["{} {}\n".format(u, s) for u, s in zip([f"DF_E_{i:07d}" for i in range(5)], sc.tolist())]
    with open(os.path.join(HERE, "score_lines.json"), "w") as f:
        json.dump({"scores_f32_le_hex": sc.astype("<f4").tobytes().hex(), "lines": lines}, f, indent=1)

    with open(os.path.join(HERE, "golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report["eer"], indent=1))


if __name__ == "__main__":
    main()
