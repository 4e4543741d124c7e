"""Which 16-bit rounding points of the scoring path cost how much score error -- TEST INFRASTRUCTURE ONLY.

The oracle's arithmetic with the device path's roundings switched on and off one group at a time, over the
first clips of the committed 2048-clip fixture (tests/golden/oracle_scores_2048.npz holds the fp32 oracle's
scores of the same clips), SLS head, 24 layers:

    python -m oracle.precision_ablate --clips 64 --case base --case base,-cw4,-cw5,-cw6,-pre --case only,w1
    python -m oracle.precision_ablate --head sae --case base --case base,-sx,-sw     (in-tree head: sx, sw below)

A case names the rounding points that are ACTIVE, as a comma list; "base" = all of them, "base,-cw1,-pre" = all
but those, "only,cw1" = that one alone.  A name without a digit means every layer of its kind.  Points:
    cw1..cw6    conv layer's weights            pre1..pre6  conv output before its LayerNorm
    post0..post6  conv block output (the next layer's operand)
    fln, pw     feature LayerNorm output, post_extract_proj weights      posw, posx  positional conv operands
    wq wo w1 w2   encoder weights by kind        h1 qkv pr ao h2 f1   encoder activations (LN1 out, q/k/v, softmax
    numerators, attention out, LN2 out, fc1+GELU out)                   pool, hw    pooled features, head fc1 weights
    sx, sw      (--head sae) the TopK auto-encoder's encoder operands: x - b_dec, encoder weights
    fmt=bf16|f16  16-bit format (default f16)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import reference_cpu as orc


ALL_POINTS = ([f"cw{i}" for i in range(1, 7)] + [f"pre{i}" for i in range(1, 7)] + [f"post{i}" for i in range(7)] +
              ["fln", "pw", "posw", "posx", "wq", "wo", "w1", "w2", "h1", "qkv", "pr", "ao", "h2", "f1", "pool", "hw",
               "sx", "sw"])


def expand(name: str):
    hit = [p for p in ALL_POINTS if p == name or (p.rstrip("0123456789") == name and p != name)]
    if not hit:
        raise SystemExit(f"unknown rounding point {name}")
    return hit


def parse_case(text: str) -> dict:
    toks = [t for t in text.split(",") if t]
    fmt = "f16"
    active = set()
    for tok in toks:
        if tok.startswith("fmt="):
            fmt = tok[4:]
        elif tok == "base":
            active |= set(ALL_POINTS)
        elif tok == "only":
            active = set()
        elif tok.startswith("-"):
            active -= set(expand(tok[1:]))
        else:
            active |= set(expand(tok))
    return {"on": active, "fmt": fmt, "name": text}


@torch.no_grad()
def forward_case(sd, wav, cfg: orc.OracleConfig, case: dict, head: str = "sls"):
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[case["fmt"]]
    on = case["on"]

    def R(t, point):
        return t.to(dt).float() if point in on else t

    p = "ssl_model.model."
    x = wav.unsqueeze(1)
    for i, (_, _, stride) in enumerate(cfg.conv_layers):
        q = f"{p}feature_extractor.conv_layers.{i}."
        w = sd[q + "0.weight"]
        if i > 0:
            w = R(w, f"cw{i}")
        x = F.conv1d(x, w, sd[q + "0.bias"], stride=stride)
        if i > 0:
            x = R(x, f"pre{i}")
        x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[q + "2.1.weight"], sd[q + "2.1.bias"], cfg.ln_eps)
        x = R(F.gelu(x), f"post{i}").transpose(1, 2)
    f = x.transpose(1, 2)
    f = R(F.layer_norm(f, (f.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.ln_eps), "fln")
    x = F.linear(f, R(sd[p + "post_extract_proj.weight"], "pw"), sd[p + "post_extract_proj.bias"])
    e = p + "encoder."
    wpos = R(orc.pos_conv_weight(sd, e + "pos_conv.0."), "posw")
    xc = F.conv1d(R(x, "posx").transpose(1, 2), wpos, sd[e + "pos_conv.0.bias"], padding=cfg.conv_pos // 2,
                  groups=cfg.conv_pos_groups)[:, :, :-1]
    x = x + F.gelu(xc).transpose(1, 2)
    B, T, C = x.shape
    H, hd = cfg.heads, C // cfg.heads
    layers = []
    for l in range(cfg.layers):
        q = f"{e}layers.{l}."
        h = R(F.layer_norm(x, (C,), sd[q + "self_attn_layer_norm.weight"], sd[q + "self_attn_layer_norm.bias"], cfg.ln_eps), "h1")
        qq = R(F.linear(h, R(sd[q + "self_attn.q_proj.weight"] * 0.125, "wq"), sd[q + "self_attn.q_proj.bias"] * 0.125), "qkv")
        kk = R(F.linear(h, R(sd[q + "self_attn.k_proj.weight"], "wq"), sd[q + "self_attn.k_proj.bias"]), "qkv")
        vv = R(F.linear(h, R(sd[q + "self_attn.v_proj.weight"], "wq"), sd[q + "self_attn.v_proj.bias"]), "qkv")
        sh = lambda t: t.view(B, T, H, hd).transpose(1, 2)  # noqa: E731
        s = sh(qq) @ sh(kk).transpose(-1, -2)
        pe = torch.exp(s - s.amax(-1, keepdim=True))
        a = (R(pe, "pr") @ sh(vv)) / pe.sum(-1, keepdim=True)
        a = R(a.transpose(1, 2).reshape(B, T, C), "ao")
        x = x + F.linear(a, R(sd[q + "self_attn.out_proj.weight"], "wo"), sd[q + "self_attn.out_proj.bias"])
        h = R(F.layer_norm(x, (C,), sd[q + "final_layer_norm.weight"], sd[q + "final_layer_norm.bias"], cfg.ln_eps), "h2")
        h = R(F.gelu(F.linear(h, R(sd[q + "fc1.weight"], "w1"), sd[q + "fc1.bias"])), "f1")
        x = x + F.linear(h, R(sd[q + "fc2.weight"], "w2"), sd[q + "fc2.bias"])
        layers.append(x.transpose(0, 1))
    if head == "sae":   # model.py:215-247
        xf = F.layer_norm(x, (C,), sd[e + "layer_norm.weight"], sd[e + "layer_norm.bias"], cfg.ln_eps)
        xs = R(xf.reshape(B * T, C) - sd["sae.b_dec"], "sx")
        acts = F.relu(F.linear(xs, R(sd["sae.encoder.weight"], "sw"), sd["sae.encoder.bias"]))
        vals, idx = acts.topk(cfg.sae_k, sorted=False, dim=-1)
        pooled = torch.zeros_like(acts).scatter_(-1, idx, vals).reshape(B, T, -1).mean(1)
        h = F.layer_norm(pooled, (pooled.shape[-1],), sd["classifier.0.weight"], sd["classifier.0.bias"], 1e-5)
        h = F.relu(F.linear(h, sd["classifier.1.weight"], sd["classifier.1.bias"]))
        return F.log_softmax(F.linear(h, sd["classifier.4.weight"], sd["classifier.4.bias"]), dim=-1)
    y0, full = orc.get_atten_f(layers)
    w = torch.sigmoid(F.linear(y0, sd["fc0.weight"], sd["fc0.bias"]))
    full = (full * w.unsqueeze(-1)).sum(dim=1).unsqueeze(1)
    z = F.batch_norm(full, sd["first_bn.running_mean"], sd["first_bn.running_var"], sd["first_bn.weight"],
                     sd["first_bn.bias"], False, 0.0, 1e-5)
    z = R(torch.flatten(F.max_pool2d(F.selu(z), (3, 3)), 1), "pool")
    z = F.selu(F.linear(z, R(sd["fc1.weight"], "hw"), sd["fc1.bias"]))
    z = F.selu(F.linear(z, sd["fc3.weight"], sd["fc3.bias"]))
    return F.log_softmax(z, dim=1)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import slsmi  # noqa: F401
    from slsmi.config import XLSR_300M
    from slsmi.weights import synth_state_dict
    import clips as clipgen
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--case", action="append", default=[])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--head", default="sls", choices=["sls", "sae"])
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    fx = np.load(os.path.join(root, "tests", "golden", "oracle_scores_2048.npz"))
    ref = fx[f"{a.head}_scores"][a.first:a.first + a.clips]
    sd = synth_state_dict(XLSR_300M, a.head, 201, 1234)
    oc = orc.OracleConfig()
    wavs = [clipgen.clips(a.first + i, min(16, a.clips - i)) for i in range(0, a.clips, 16)]
    for text in a.case or ["base"]:
        case = parse_case(text)
        t0 = time.time()
        dev = np.concatenate([torch.exp(forward_case(sd, w, oc, case, a.head))[:, 1].numpy() for w in wavs])
        d = (dev - ref).astype(np.float64)
        err = np.abs(d)
        # a rounded weight tensor shifts every clip's score the same way: mean = the shift, sd = the per-clip part
        print(f"{text:40s} rms {np.sqrt((err ** 2).mean()):.3e}  median {np.median(err):.3e}  max {err.max():.3e}  "
              f"mean {d.mean():+.3e}  sd {d.std():.3e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
