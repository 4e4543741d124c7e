#!/usr/bin/env python3
"""Golden gradients of the training step, produced by autograd through the reference's own model.py
and wav2vec/wav2vec2.py (build container only): the loss of main.py's train_epoch (main.py:224-238:
NLLLoss with class weights 0.1 / 0.9, main.py:524-525, plus sae_weight x reconstruction loss,
model.py:295-299) on a seeded batch, backward, and for every parameter the gradient's norm and its
first eight values.  The model is in eval() mode so that the classifier's Dropout draws nothing
(the mask is an input of the HIP step and of the oracle); every other module of the forward
computes the same in both modes (the backbone's dropouts and layerdrop are 0 in this configuration).

Same stand-ins and seeded weights as make_golden.py (imported from it).  Writes grads_xlsr2.npz and
grads_report.json.

`make_golden_grads.py deep` adds the shape main.py really trains (main.py:201-258, model.py:209-210: all 24
encoder layers fine-tuned, batch 5, 64600 samples): grads_xlsr24.npz with, for every parameter, the gradient's
norm and 64 values at evenly spaced positions of the flattened tensor (the whole gradients are 1.2 GB), plus the
same comparison of the oracle's autograd with the reference's.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import BackboneConfig, synth_state_dict, orc  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, ref_model, _, _ = mg.load_reference()
    cfg = BackboneConfig(layers=2)
    L, B, seed = 64600, 2, 1234
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    labels = torch.tensor([1, 0])
    weight = torch.tensor([0.1, 0.9])
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), seed)
    m, _, _ = mg.build_ref_model(w2v, ref_model, cfg, sd, "sae", 4096, 128)
    m.eval()
    logp, sae_loss = m(wav)
    loss = m.compute_total_loss(F.nll_loss(logp, labels, weight=weight), sae_loss)
    loss.backward()
    out = {"seed": np.int64(seed), "B": np.int64(B), "L": np.int64(L), "labels": labels.numpy(),
           "loss": np.float32(loss.detach()), "logp": logp.detach().numpy()}
    names = []
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        out["norm:" + n] = np.float64(p.grad.double().norm())
        out["head:" + n] = p.grad.flatten()[:8].numpy().copy()
    out["names"] = np.array(names)
    # the oracle's autograd on the same batch
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    oc = orc.OracleConfig(layers=2)
    bb = orc.backbone_forward(params, wav, oc, pfx="ssl_model.model.")
    ologp, oloss = orc.sae_head(params, bb["x"], oc, return_sae_loss=True)
    (F.nll_loss(ologp, labels, weight=weight) + 0.1 * oloss).backward()
    worst = 0.0
    for n in names:
        a, b = params[n].grad.flatten(), dict(m.named_parameters())[n].grad.flatten()
        if float(b.norm()) > 1e-7:  # the key biases have a zero gradient (softmax ignores a common shift of the scores)
            worst = max(worst, float((a - b).norm() / b.norm()))
    rep = {"parameters_with_gradient": len(names), "oracle_vs_reference_worst_relative_l2": worst,
           "loss": float(loss.detach())}
    np.savez_compressed(os.path.join(HERE, "grads_xlsr2.npz"), **out)
    with open(os.path.join(HERE, "grads_report.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


def slice_index(numel: int, count: int = 64) -> np.ndarray:
    """`count` evenly spaced positions of a flattened tensor (all of them when it is shorter)."""
    if numel <= count:
        return np.arange(numel, dtype=np.int64)
    return np.floor(np.arange(count, dtype=np.float64) * (numel - 1) / (count - 1)).astype(np.int64)


def deep():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    w2v, ref_model, _, _ = mg.load_reference()
    layers, L, B, seed = 24, 64600, 5, 1234
    cfg = BackboneConfig(layers=layers)
    g = torch.Generator().manual_seed(77)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)
    labels = torch.tensor([1, 0, 1, 1, 0])
    weight = torch.tensor([0.1, 0.9])
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), seed)
    m, _, _ = mg.build_ref_model(w2v, ref_model, cfg, sd, "sae", 4096, 128)
    m.eval()
    logp, sae_loss = m(wav)
    loss = m.compute_total_loss(F.nll_loss(logp, labels, weight=weight), sae_loss)
    loss.backward()
    out = {"seed": np.int64(seed), "wav_seed": np.int64(77), "B": np.int64(B), "L": np.int64(L),
           "layers": np.int64(layers), "labels": labels.numpy(), "loss": np.float32(loss.detach()),
           "sae_loss": np.float32(sae_loss.detach()), "logp": logp.detach().numpy()}
    names = []
    ref_grads = {}
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        flat = p.grad.flatten()
        out["norm:" + n] = np.float64(flat.double().norm())
        out["slice:" + n] = flat[torch.from_numpy(slice_index(flat.numel()))].numpy().copy()
        ref_grads[n] = flat.clone()
    out["names"] = np.array(names)
    del m, logp, sae_loss, loss
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    oc = orc.OracleConfig(layers=layers)
    bb = orc.backbone_forward(params, wav, oc, pfx="ssl_model.model.")
    ologp, oloss = orc.sae_head(params, bb["x"], oc, return_sae_loss=True)
    (F.nll_loss(ologp, labels, weight=weight) + 0.1 * oloss).backward()
    worst, worst_name = 0.0, ""
    for n in names:
        a, b = params[n].grad.flatten(), ref_grads[n]
        if float(b.norm()) > 1e-7 and not n.endswith("k_proj.bias"):
            r = float((a - b).norm() / b.norm())
            if r > worst:
                worst, worst_name = r, n
    rep = {"parameters_with_gradient": len(names), "oracle_vs_reference_worst_relative_l2": worst,
           "worst_tensor": worst_name, "loss": float(out["loss"]), "layers": layers, "batch": B}
    np.savez_compressed(os.path.join(HERE, "grads_xlsr24.npz"), **out)
    with open(os.path.join(HERE, "grads_deep_report.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deep":
        deep()
    else:
        main()
