"""CPU emulation of the HIP pipeline's rounding points -- TEST INFRASTRUCTURE ONLY.

Re-runs the oracle's arithmetic with the operand roundings the device path
applies (16-bit GEMM/conv operands and stored activations, fp32 accumulation,
fp32 residual stream and statistics) so the error budget of a storage format
can be measured without a GPU:

    python -m oracle.precision_sim --layers 2 --fmt bf16
"""
from __future__ import annotations

import argparse
import sys

import torch
import torch.nn.functional as F

from . import reference_cpu as orc


def make_round(fmt: str):
    if fmt == "fp32":
        return lambda t: t
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]
    return lambda t: t.to(dt).float()


@torch.no_grad()
def forward_emulated(sd, wav, cfg: orc.OracleConfig, head: str, fmt: str, conv_pre_ln_fp32: bool = False):
    R = make_round(fmt)
    p = "ssl_model.model."
    x = wav.unsqueeze(1)
    for i, (_, _, stride) in enumerate(cfg.conv_layers):
        q = f"{p}feature_extractor.conv_layers.{i}."
        w = sd[q + "0.weight"] if i == 0 else R(sd[q + "0.weight"])
        x = F.conv1d(x, w, sd[q + "0.bias"], stride=stride)
        if i > 0 and not conv_pre_ln_fp32:
            x = R(x)  # pre-LN conv output stored in 16 bit
        x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[q + "2.1.weight"], sd[q + "2.1.bias"], cfg.ln_eps)
        x = R(F.gelu(x)).transpose(1, 2)
    f = x.transpose(1, 2)
    f = R(F.layer_norm(f, (f.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.ln_eps))
    x = F.linear(f, R(sd[p + "post_extract_proj.weight"]), sd[p + "post_extract_proj.bias"])
    e = p + "encoder."
    wpos = R(orc.pos_conv_weight(sd, e + "pos_conv.0."))
    xc = F.conv1d(R(x).transpose(1, 2), wpos, sd[e + "pos_conv.0.bias"], padding=cfg.conv_pos // 2,
                  groups=cfg.conv_pos_groups)[:, :, :-1]
    x = x + F.gelu(xc).transpose(1, 2)
    B, T, C = x.shape
    H, hd = cfg.heads, C // cfg.heads
    layers = []
    for l in range(cfg.layers):
        q = f"{e}layers.{l}."
        h = R(F.layer_norm(x, (C,), sd[q + "self_attn_layer_norm.weight"], sd[q + "self_attn_layer_norm.bias"], cfg.ln_eps))
        qq = R(F.linear(h, R(sd[q + "self_attn.q_proj.weight"] * 0.125), sd[q + "self_attn.q_proj.bias"] * 0.125))
        kk = R(F.linear(h, R(sd[q + "self_attn.k_proj.weight"]), sd[q + "self_attn.k_proj.bias"]))
        vv = R(F.linear(h, R(sd[q + "self_attn.v_proj.weight"]), sd[q + "self_attn.v_proj.bias"]))
        sh = lambda t: t.view(B, T, H, hd).transpose(1, 2)  # noqa: E731
        pr = torch.softmax(sh(qq) @ sh(kk).transpose(-1, -2), dim=-1)
        # device path: un-normalised exp() rounded to 16 bit, normalised after P.V in fp32
        mx = (sh(qq) @ sh(kk).transpose(-1, -2)).amax(-1, keepdim=True)
        pe = torch.exp(sh(qq) @ sh(kk).transpose(-1, -2) - mx)
        a = (R(pe) @ sh(vv)) / pe.sum(-1, keepdim=True)
        del pr
        a = R(a.transpose(1, 2).reshape(B, T, C))
        x = x + F.linear(a, R(sd[q + "self_attn.out_proj.weight"]), sd[q + "self_attn.out_proj.bias"])
        h = R(F.layer_norm(x, (C,), sd[q + "final_layer_norm.weight"], sd[q + "final_layer_norm.bias"], cfg.ln_eps))
        h = R(F.gelu(F.linear(h, R(sd[q + "fc1.weight"]), sd[q + "fc1.bias"])))
        x = x + F.linear(h, R(sd[q + "fc2.weight"]), sd[q + "fc2.bias"])
        layers.append(x.transpose(0, 1))
    xf = F.layer_norm(x, (C,), sd[e + "layer_norm.weight"], sd[e + "layer_norm.bias"], cfg.ln_eps)
    if head == "sls":
        y0, full = orc.get_atten_f(layers)
        w = torch.sigmoid(F.linear(y0, sd["fc0.weight"], sd["fc0.bias"]))
        full = (full * w.unsqueeze(-1)).sum(dim=1).unsqueeze(1)
        z = F.batch_norm(full, sd["first_bn.running_mean"], sd["first_bn.running_var"], sd["first_bn.weight"],
                         sd["first_bn.bias"], False, 0.0, 1e-5)
        z = R(torch.flatten(F.max_pool2d(F.selu(z), (3, 3)), 1))
        z = F.selu(F.linear(z, R(sd["fc1.weight"]), sd["fc1.bias"]))
        z = F.selu(F.linear(z, sd["fc3.weight"], sd["fc3.bias"]))
        return F.log_softmax(z, dim=1), layers, xf
    xs = R(xf.reshape(B * T, C) - sd["sae.b_dec"])
    acts = F.relu(F.linear(xs, R(sd["sae.encoder.weight"]), sd["sae.encoder.bias"]))
    vals, idx = acts.topk(cfg.sae_k, sorted=False, dim=-1)
    enc = torch.zeros_like(acts).scatter_(-1, idx, vals)
    pooled = enc.reshape(B, T, -1).mean(1)
    h = F.layer_norm(pooled, (pooled.shape[-1],), sd["classifier.0.weight"], sd["classifier.0.bias"], 1e-5)
    h = F.relu(F.linear(h, sd["classifier.1.weight"], sd["classifier.1.bias"]))
    return F.log_softmax(F.linear(h, sd["classifier.4.weight"], sd["classifier.4.bias"]), dim=-1), layers, xf


def main():
    sys.path.insert(0, ".")
    import slsmi  # noqa: F401
    from slsmi.config import BackboneConfig
    from slsmi.weights import synth_state_dict
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--fmt", nargs="+", default=["bf16", "f16"])
    a = ap.parse_args()
    cfg = BackboneConfig(layers=a.layers)
    oc = orc.OracleConfig(layers=a.layers)
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn(a.batch, 64600, generator=g)).clamp(-1, 1)
    for head in ("sls", "sae"):
        sd = synth_state_dict(cfg, head, 201, 1234)
        ref, bb = orc.model_forward(sd, wav, oc, head, return_intermediates=True)
        if head == "sls":
            y0, full = orc.get_atten_f(bb["layer_results"])
            w = torch.sigmoid(F.linear(y0, sd["fc0.weight"], sd["fc0.bias"]))
            ff = (full * w.unsqueeze(-1)).sum(dim=1)
            print(f"[sls] weighted layer sum: mean {ff.mean():.4f} var {ff.var():.4f}")
        print(f"[{head}] oracle logp {ref.tolist()}")
        for fmt in a.fmt:
            for pre in (False, True):
                lp, layers, xf = forward_emulated(sd, wav, oc, head, fmt, conv_pre_ln_fp32=pre)
                de = (torch.exp(lp)[:, 1] - torch.exp(ref)[:, 1]).abs().max().item()
                le = max(((a_ - b_).abs().max() / b_.abs().max()).item() for a_, b_ in zip(layers, bb["layer_results"]))
                rms = max(((a_ - b_).pow(2).mean().sqrt() / b_.pow(2).mean().sqrt()).item()
                          for a_, b_ in zip(layers, bb["layer_results"]))
                print(f"  fmt={fmt} conv_pre_ln_fp32={pre}: score err {de:.2e}  layer max-rel {le:.2e} rms-rel {rms:.2e}")


if __name__ == "__main__":
    main()
