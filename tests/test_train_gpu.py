"""The HIP training step (slsmi_train_forward / slsmi_train_backward behind Model.forward in train
mode) against autograd through the CPU oracle: same weights, same input, same loss; every parameter
gradient the library produces is compared tensor by tensor.

Tolerances: the backward contractions run on bfloat16 operands (8-bit mantissa) over activations the
forward stored in 16 bits, so a gradient tensor is compared by direction and size -- cosine >= 0.99 and
norm within 5 % -- not element by element.  The TopK selection of the device and of the oracle can
differ in members at the k-th value, which moves single entries of the SAE gradients.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import slsmi
from slsmi.config import BackboneConfig
from slsmi.model import Model
from slsmi.weights import synth_state_dict
from oracle import reference_cpu as orc

DEV = torch.device("cuda:0")


def seeded_wav(B, L=64600, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)


def oracle_grads(sd, wav, oc, labels, weight, sae_weight, mask):
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    bb = orc.backbone_forward(params, wav, oc, pfx="ssl_model.model.")
    logp, sae_loss = orc.sae_head(params, bb["x"], oc, return_sae_loss=True, dropout_mask=mask)
    loss = F.nll_loss(logp, labels, weight=weight) + sae_weight * sae_loss   # main.py:524-525, model.py:295-299
    loss.backward()
    return logp.detach(), sae_loss.detach(), loss.detach(), {k: v.grad for k, v in params.items()}


@pytest.mark.parametrize("layers,B,L", [(2, 3, 64600), (1, 2, 32000), (1, 11, 32000)])
def test_training_step_gradients(layers, B, L):
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    sd = synth_state_dict(cfg, "sae", T, 1234)
    oc = orc.OracleConfig(layers=layers)
    wav = seeded_wav(B, L, seed=5)
    labels = torch.tensor(([1, 0, 1] * B)[:B])
    weight = torch.tensor([0.1, 0.9])          # main.py:524
    g = torch.Generator().manual_seed(9)
    mask = torch.bernoulli(torch.full((B, 256), 0.7), generator=g) / 0.7
    ref_logp, ref_sae, ref_loss, ref = oracle_grads(sd, wav, oc, labels, weight, 0.1, mask)

    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True, use_sparse_features=True,
              num_samples=L).train()
    m.load_state_dict(sd)
    m.to(DEV)
    logp, sae_loss = m(wav.to(DEV), dropout_mask=mask.to(DEV))          # return_sae_loss defaults to True
    cls = F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV))
    loss = m.compute_total_loss(cls, sae_loss)
    loss.backward()
    assert (logp.detach().cpu() - ref_logp).abs().max() < 2e-2
    assert abs(float(sae_loss.detach()) - float(ref_sae)) / float(ref_sae) < 1e-2
    eng = m._engine
    names = eng.grad_names([n for n, _ in m.named_parameters()])
    assert len(names) >= 16 * layers + 14 + 4 * 7 + 2   # + conv blocks and the feature LayerNorm
    worst = []
    for n, p in m.named_parameters():
        if n not in names:
            continue
        assert p.grad is not None, n
        a, b = p.grad.detach().float().cpu().flatten(), ref[n].flatten()
        assert torch.isfinite(a).all(), n
        nb = float(b.norm())
        if n.endswith("k_proj.bias"):
            # softmax is invariant to a shift of all keys' scores: the true gradient is zero, both sides
            # hold rounding noise only -- compare with the size of the value bias gradient instead
            assert float(a.norm()) < 0.02 * float(ref[n.replace("k_proj", "v_proj")].norm()), n
            continue
        if nb < 1e-12:
            assert float(a.norm()) < 1e-9, n
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        ratio = float(a.norm()) / nb
        worst.append((cos, ratio, n))
    worst.sort()
    for cos, ratio, n in worst[:8]:
        print(f"[grad] cos {cos:.5f} norm ratio {ratio:.4f} {n}")
    bad = [(c, r, n) for c, r, n in worst if c < 0.99 or abs(r - 1) > 0.05]
    assert not bad, bad[:10]


@pytest.mark.parametrize("form", ["window", "recon", "ssl"])
def test_other_model_forms_train(form):
    """The other forms main.py can build and train: window TopK (--use_window_topk, main.py:493-505,
    model_window_topk.py:271-393), the reconstruction as classifier input (use_sparse_features=False,
    model.py:228-233) and no auto-encoder (use_sae=False, model.py:171-179).  Output, reconstruction loss and every
    gradient against autograd through the CPU oracle's restatement of the same form; then a few optimizer steps
    lower the loss."""
    from slsmi.model import WindowTopKModel
    layers, B, L = 2, 3, 32000
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    head = {"window": "sae", "recon": "sae_recon", "ssl": "ssl"}[form]
    sd = synth_state_dict(cfg, head, T, 1234)
    oc = orc.OracleConfig(layers=layers, sae_window=8 if form == "window" else 1)
    wav = seeded_wav(B, L, seed=31)
    labels = torch.tensor([1, 0, 1])
    weight = torch.tensor([0.1, 0.9])
    mask = torch.bernoulli(torch.full((B, 256), 0.7), generator=torch.Generator().manual_seed(4)) / 0.7
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    bb = orc.backbone_forward(params, wav, oc, pfx="ssl_model.model.")
    ref_logp, ref_sae = orc.sae_head(params, bb["x"], oc, return_sae_loss=True, use_sae=form != "ssl",
                                     use_sparse_features=form == "window", dropout_mask=mask)
    ref_loss = F.nll_loss(ref_logp, labels, weight=weight)
    if ref_sae is not None:
        ref_loss = ref_loss + 0.1 * ref_sae
    ref_loss.backward()

    if form == "window":
        m = WindowTopKModel(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L)
    else:
        m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, use_sae=form == "recon",
                  use_sparse_features=False)
    m.train()
    m.load_state_dict(sd)
    m.to(DEV)
    logp, sae_loss = m(wav.to(DEV), dropout_mask=mask.to(DEV))
    assert (sae_loss is None) == (form == "ssl")
    loss = m.compute_total_loss(F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)), sae_loss)
    loss.backward()
    assert (logp.detach().cpu() - ref_logp.detach()).abs().max() < 2e-2
    if ref_sae is not None:
        assert abs(float(sae_loss.detach()) - float(ref_sae.detach())) < 1e-2 * float(ref_sae.detach())
    worst = []
    for n, p in m.named_parameters():
        ref = params[n].grad if n in params else None
        if ref is None:   # pre-training leftovers; the auto-encoder's parameters when the form has none
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        a, b = p.grad.detach().float().cpu().flatten(), ref.flatten()
        assert torch.isfinite(a).all(), n
        if n.endswith("k_proj.bias"):   # true gradient zero (see test_training_step_gradients)
            assert float(a.norm()) < 0.02 * float(params[n.replace("k_proj", "v_proj")].grad.norm()), n
            continue
        if float(b.norm()) < 1e-12:
            assert float(a.norm()) < 1e-9, n
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        worst.append((cos, float(a.norm()) / float(b.norm()), n))
    worst.sort()
    for cos, ratio, n in worst[:5]:
        print(f"[{form} grad] cos {cos:.5f} norm ratio {ratio:.4f} {n}")
    assert len(worst) >= 15 * layers + 4 * 7 + 2 + 2 + 3 + 2 + 6
    bad = [(c, r, n) for c, r, n in worst if c < 0.99 or abs(r - 1) > 0.05]
    assert not bad, bad[:10]
    # a few steps of the recipe's optimizer lower the loss on this batch
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    first = float(loss.detach())
    for _ in range(4):
        opt.step()
        opt.zero_grad()
        logp, sae_loss = m(wav.to(DEV), dropout_mask=mask.to(DEV))
        loss = m.compute_total_loss(F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)), sae_loss)
        loss.backward()
    assert float(loss.detach()) < first, (first, float(loss.detach()))


def test_layer_weight_gradients_do_not_depend_on_where_the_buffers_lie(monkeypatch):
    """The encoder layers' weight gradients are one GEMM launch per kind of matrix when the caller's buffers lie at
    equal distances (what Engine.grad_buffers hands out), one launch per layer otherwise: same bits."""
    cfg = BackboneConfig(layers=3)
    B, L = 3, 32000
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), 77)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True, use_sparse_features=True,
              num_samples=L).train()
    m.load_state_dict(sd)
    m.to(DEV)
    wav = seeded_wav(B, L, seed=8).to(DEV)
    mask = torch.ones(B, 256, device=DEV)
    labels = torch.tensor([1, 0, 1], device=DEV)

    def step():
        m.zero_grad(set_to_none=True)
        logp, sae_loss = m(wav, dropout_mask=mask)
        m.compute_total_loss(F.nll_loss(logp, labels), sae_loss).backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    even = step()
    lw = {n: p for n, p in m.named_parameters() if n.endswith("encoder.layers.0.fc1.weight") or
          n.endswith("encoder.layers.1.fc1.weight") or n.endswith("encoder.layers.2.fc1.weight")}
    ptrs = [p.grad.data_ptr() for _, p in sorted(lw.items())]
    assert ptrs[1] - ptrs[0] == ptrs[2] - ptrs[1] == 4096 * 1024 * 4      # slices of one tensor
    eng = m._engine
    spacers = []

    def scattered(params):
        out = {}
        for i, (n, p) in enumerate(params.items()):
            spacers.append(torch.empty(4096 * (1 + i % 3), device=DEV))   # uneven distances between the buffers
            out[n] = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return out

    monkeypatch.setattr(eng, "grad_buffers", scattered)
    apart = step()
    assert set(apart) == set(even)
    for n in even:
        assert torch.equal(even[n], apart[n]), n


@pytest.mark.parametrize("layers,B,L", [(2, 3, 64600), (3, 5, 32000), (1, 2, 48000), (1, 18, 32000)])
def test_sls_model_training_step_gradients(layers, B, L):
    """Train mode of the upstream SLS model, Model(args, device): backbone and head, forward and backward, in libslsmi
    (slsmi_train_sls_forward / _backward).  Every gradient -- backbone and head -- against autograd through the CPU
    oracle with BatchNorm2d on batch statistics; the final encoder LayerNorm, which this head does not read, gets
    none, as in the reference.  (48000 samples: 149 frames, two of which the 3 x 3 pool drops; 18 clips: the head's
    fc1 kernels take the batch in more than one pass.)"""
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    sd = synth_state_dict(cfg, "sls", T, 1234)
    oc = orc.OracleConfig(layers=layers)
    wav = seeded_wav(B, L, seed=7)
    labels = torch.tensor(([1, 0, 1, 1, 0] * B)[:B])
    weight = torch.tensor([0.1, 0.9])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    full = dict(sd, **params)
    bb = orc.backbone_forward(full, wav, oc, pfx="ssl_model.model.")
    ref_logp = orc.sls_head(full, bb["layer_results"], training=True)
    F.nll_loss(ref_logp, labels, weight=weight).backward()

    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L).train()
    m.load_state_dict(sd)
    m.to(DEV)
    rm0 = m.first_bn.running_mean.clone()
    logp = m(wav.to(DEV))                                   # upstream arity: the output alone
    assert m(wav.to(DEV), return_sae_loss=True)[1] is None
    F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)).backward()
    assert (logp.detach().cpu() - ref_logp.detach()).abs().max() < 2e-2
    assert not torch.equal(m.first_bn.running_mean, rm0)   # nn.BatchNorm2d in train mode tracks its statistics
    worst = []
    for n, p in m.named_parameters():
        ref = params[n].grad
        if ref is None:   # the final encoder LayerNorm and the checkpoint's pre-training leftovers
            assert p.grad is None, n
            continue
        assert p.grad is not None and ref is not None, n
        a, b = p.grad.detach().float().cpu().flatten(), ref.flatten()
        assert torch.isfinite(a).all(), n
        if n.endswith("k_proj.bias"):   # true gradient zero (see test_training_step_gradients)
            assert float(a.norm()) < 0.02 * float(params[n.replace("k_proj", "v_proj")].grad.norm()), n
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        worst.append((cos, float(a.norm()) / float(b.norm()), n))
    worst.sort()
    for cos, ratio, n in worst[:6]:
        print(f"[sls grad] cos {cos:.5f} norm ratio {ratio:.4f} {n}")
    # fc0 (the layer weights) sits in front of a BatchNorm on batch statistics, which removes the scale of the
    # weighted sum: its gradient is a small difference of large terms and follows the 16-bit forward's rounding
    # more than any other (direction kept, size within 20 %); checked exactly below on the device's own features
    bad = [(c, r, n) for c, r, n in worst if c < 0.99 or abs(r - 1) > (0.2 if n.startswith("fc0.") else 0.05)]
    assert not bad, bad[:10]
    assert len(worst) == 15 * layers + 4 * 7 + 2 + 2 + 3 + 8   # k_proj.bias aside: layers, conv blocks, feature LN, projection, pos_conv, head
    # the head alone: the oracle's head on the layer outputs the device produced gives the device's head gradients
    feats = m._engine.train_backbone_forward(wav.to(DEV)).cpu()
    head = {k: v.clone().requires_grad_(True) for k, v in sd.items()
            if v.is_floating_point() and not k.startswith("ssl_model.") and "running" not in k}
    logp2 = orc.sls_head(dict(sd, **head), [feats[l].transpose(0, 1) for l in range(layers)], training=True)
    F.nll_loss(logp2, labels, weight=weight).backward()
    for n, v in head.items():
        got = dict(m.named_parameters())[n].grad.cpu()
        # (fc0 with ONE layer: the BatchNorm removes the weighted sum's scale, so the layer weight's true gradient is
        # ~0 and what both sides hold is the rounding of a cancelling sum)
        floor = 2e-6 if layers == 1 and n.startswith("fc0.") else 1e-9
        assert (got - v.grad).norm() < 2e-3 * v.grad.norm() + floor, n


def test_sls_head_kernels_against_autograd_on_the_same_features(monkeypatch):
    """The SLS head's HIP forward and backward against torch autograd on the modules the model holds, both behind
    the same HIP backbone (SLSMI_HEAD_AUTOGRAD=1 selects the second): log-probs, every head gradient, the
    BatchNorm's running statistics and -- through the gradient the head hands down -- every backbone gradient.
    With a layer skipped by layerdrop as well."""
    cfg = BackboneConfig(layers=3)
    L, B = 32000, 3
    sd = synth_state_dict(cfg, "sls", cfg.num_frames(L), 1234)
    wav = seeded_wav(B, L, seed=21).to(DEV)
    labels = torch.tensor([1, 0, 1], device=DEV)
    weight = torch.tensor([0.1, 0.9], device=DEV)
    for keep in (None, [True, False, True]):
        out = {}
        for mode in ("hip", "autograd"):
            if mode == "autograd":
                monkeypatch.setenv("SLSMI_HEAD_AUTOGRAD", "1")
            else:
                monkeypatch.delenv("SLSMI_HEAD_AUTOGRAD", raising=False)
            m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L).train()
            m.load_state_dict(sd)
            m.to(DEV)
            masks = None if keep is None else {"layer_keep": torch.tensor(keep)}
            logp = m(wav, backbone_masks=masks)
            F.nll_loss(logp, labels, weight=weight).backward()
            out[mode] = (logp.detach(), {n: p.grad for n, p in m.named_parameters()},
                         m.first_bn.running_mean.clone(), m.first_bn.running_var.clone(), int(m.first_bn.num_batches_tracked))
        (la, ga, rma, rva, na), (lb, gb, rmb, rvb, nb) = out["hip"], out["autograd"]
        assert (la - lb).abs().max() < 2e-5
        assert na == nb == 1
        assert (rma - rmb).abs().max() < 1e-6 and (rva - rvb).abs().max() < 1e-5 * float(rvb.abs().max()) + 1e-7
        diffs = []
        for n in gb:
            if gb[n] is None:
                assert ga[n] is None, n
                continue
            assert ga[n] is not None, n
            if n.endswith("k_proj.bias"):
                continue   # true gradient zero: rounding noise on both sides
            diffs.append((float((ga[n] - gb[n]).norm()) / (float(gb[n].norm()) + 1e-20), n))
        head = [(d, n) for d, n in diffs if not n.startswith("ssl_model.")]
        top = [(d, n) for d, n in diffs if ".encoder.layers.2." in n]
        rest = [(d, n) for d, n in diffs if n.startswith("ssl_model.") and ".encoder.layers.2." not in n]
        print(f"[sls head hip vs autograd, keep={keep}] head worst {max(head)}, top layer worst {max(top)}, below worst {max(rest)}")
        # the head's own parameters: fp32 against fp32.  The backbone's: the same bfloat16 backward fed two gradients
        # that agree to fp32 rounding -- the top layer's parameters see them after one rounding to bfloat16; further
        # down, every rounding step lets a last-bit difference flip more roundings (the gradient's own error against
        # the fp32 oracle there is a few per cent, test_sls_model_training_step_gradients)
        assert len(head) == 8 and max(head)[0] < 2e-4, max(head)
        assert max(top)[0] < 5e-3, max(top)
        assert max(rest)[0] < 3e-2, max(rest)


def test_sls_training_entry_points_refuse_what_they_cannot_run():
    """slsmi_train_sls_forward: only on a context with the SLS head; an input whose frames do not pool to fc1's
    in-features is an error (the reference's fc1 would raise a shape mismatch); a backward needs its forward."""
    cfg = BackboneConfig(layers=1)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16").train()          # fc1 sized for 201 frames
    m.load_state_dict(synth_state_dict(cfg, "sls", 201, 1234))
    m.to(DEV)
    with pytest.raises(slsmi.SlsmiError, match="pool to"):
        m(seeded_wav(2, 32000).to(DEV))
    wav = seeded_wav(2).to(DEV)
    m(wav).sum().backward()
    assert m.fc1.weight.grad is not None and torch.isfinite(m.fc1.weight.grad).all()
    with pytest.raises(slsmi.SlsmiError, match="must follow"):       # a second backward of the same forward
        m._engine.train_sls_backward(wav, torch.zeros(2, 2, device=DEV), {}, m.fc1.weight)
    sae = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
    sae.load_state_dict(synth_state_dict(cfg, "sae", 201, 1234))
    sae.to(DEV)
    sae(seeded_wav(2).to(DEV))
    with pytest.raises(slsmi.SlsmiError, match="SLS head"):
        sae._engine.train_sls_forward(seeded_wav(2).to(DEV), m.fc1.weight)


def _backbone_masks(layers, B, T, ffn=4096, seed=3):
    g = torch.Generator().manual_seed(seed)

    def bern(p, *shape):
        return torch.bernoulli(torch.full(shape, 1.0 - p), generator=g) / (1.0 - p)
    keep = torch.tensor([True] * layers)
    if layers >= 3:
        keep[1] = False
    return {"input": bern(0.1, B, T, 1024), "encoder": bern(0.1, B, T, 1024), "attn_out": bern(0.1, layers, B, T, 1024),
            "ffn_act": bern(0.2, layers, B, T, ffn), "ffn_out": bern(0.1, layers, B, T, 1024), "layer_keep": keep,
            "attn": bern(0.1, layers, B, 16, T, T)}


@pytest.mark.parametrize("head", ["sae", "sls"])
def test_backbone_dropout_and_layerdrop_masks(head):
    """The backbone's train-mode stochastic paths with the draws made by the caller: dropout_input on the projected
    features (wav2vec2.py:598), F.dropout after the positional conv (:933), dropout1 / 2 / 3 of every layer
    (:1053-1061), MultiheadAttention's dropout on the attention probabilities (:1012) and layerdrop (:941-942, the skipped layer appends no layer result).  Output and every gradient
    against autograd through the CPU oracle under the same masks; the skipped layer's parameters get NO gradient
    (grad stays None, as in the reference, where a skipped layer is not in the graph: torch.optim.Adam then leaves
    them alone -- no weight decay, no momentum, no step count)."""
    layers, B, L = 3, 2, 32000
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    sd = synth_state_dict(cfg, head, T, 1234)
    oc = orc.OracleConfig(layers=layers)
    wav = seeded_wav(B, L, seed=15)
    labels = torch.tensor([1, 0])
    weight = torch.tensor([0.1, 0.9])
    masks = _backbone_masks(layers, B, T)
    cmask = torch.ones(B, 256)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    full = dict(sd, **params)
    bb = orc.backbone_forward(full, wav, oc, pfx="ssl_model.model.", train_masks=masks)
    assert len(bb["layer_results"]) == layers - 1
    if head == "sae":
        ref_logp, ref_sae = orc.sae_head(full, bb["x"], oc, return_sae_loss=True, dropout_mask=cmask)
        (F.nll_loss(ref_logp, labels, weight=weight) + 0.1 * ref_sae).backward()
    else:
        ref_logp = orc.sls_head(full, bb["layer_results"], training=True)
        F.nll_loss(ref_logp, labels, weight=weight).backward()

    kw = dict(use_sae=True) if head == "sae" else {}
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, **kw).train()
    m.load_state_dict(sd)
    m.to(DEV)
    dmasks = {k: v.to(DEV) for k, v in masks.items()}
    if head == "sae":
        logp, sae_loss = m(wav.to(DEV), dropout_mask=cmask.to(DEV), backbone_masks=dmasks)
        m.compute_total_loss(F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)), sae_loss).backward()
    else:
        logp = m(wav.to(DEV), backbone_masks=dmasks)
        F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)).backward()
    assert (logp.detach().cpu() - ref_logp.detach()).abs().max() < 2e-2
    worst, zeros = [], 0
    for n, p in m.named_parameters():
        ref = params[n].grad
        if ".encoder.layers.1." in n:       # skipped by layerdrop: never used by the oracle, no gradient from the device
            assert ref is None and p.grad is None, n
            zeros += 1
            continue
        if ref is None or n.endswith("k_proj.bias"):
            continue
        assert p.grad is not None, n
        a, b = p.grad.detach().float().cpu().flatten(), ref.flatten()
        assert torch.isfinite(a).all(), n
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        worst.append((cos, float(a.norm()) / float(b.norm()), n))
    assert zeros == 16
    worst.sort()
    for cos, ratio, n in worst[:5]:
        print(f"[masks {head}] cos {cos:.5f} norm ratio {ratio:.4f} {n}")
    bad = [(c, r, n) for c, r, n in worst if c < 0.99 or abs(r - 1) > (0.2 if n.startswith("fc0.") else 0.05)]
    assert not bad, bad[:10]
    # the attention probabilities' dropout is applied: without it the output moves away from the oracle's
    no_attn = {k: v for k, v in dmasks.items() if k != "attn"}
    out_na = m(wav.to(DEV), dropout_mask=cmask.to(DEV), backbone_masks=no_attn) if head == "sae" else m(wav.to(DEV), backbone_masks=no_attn)
    out_na = out_na[0] if isinstance(out_na, tuple) else out_na
    err_with = float((logp.detach().cpu() - ref_logp.detach()).abs().max())
    err_without = float((out_na.detach().cpu() - ref_logp.detach()).abs().max())
    print(f"[masks {head}] |logp - oracle| with the attention masks {err_with:.2e}, without {err_without:.2e}")
    assert err_without > 4 * err_with + 1e-4
    # the masks belong to one forward / backward pair: the next step runs without them
    out2 = m(wav.to(DEV), dropout_mask=cmask.to(DEV)) if head == "sae" else m(wav.to(DEV))
    out2 = out2[0] if isinstance(out2, tuple) else out2
    assert (out2.detach() - logp.detach()).abs().max() > 1e-4


def test_backbone_masks_drawn_from_the_checkpoint_configuration(tmp_path):
    """A fairseq-layout checkpoint whose configuration sets dropout / layerdrop: Model.forward in train mode draws
    the masks itself (two forwards differ, eval does not) -- with the dropouts of the layers alone and with
    MultiheadAttention's attention_dropout alone."""
    import argparse
    cfg = BackboneConfig(layers=2)
    sd = synth_state_dict(cfg, "sls", 201, 1234)
    bsd = {k[len("ssl_model.model."):]: v for k, v in sd.items() if k.startswith("ssl_model.model.")}
    ns = argparse.Namespace(encoder_layers=2, dropout=0.1, activation_dropout=0.1, dropout_input=0.1, encoder_layerdrop=0.3,
                            attention_dropout=0.0)
    path = tmp_path / "bb.pt"
    torch.save({"args": ns, "model": bsd}, str(path))
    m = Model(None, DEV, cp_path=str(path), cfg=cfg, precision="f16").train()
    m.load_state_dict(sd)
    m.to(DEV)
    wav = seeded_wav(2, seed=3).to(DEV)
    # (layerdrop draws come from numpy's global generator, as in the reference: with this seed the two forwards keep
    # layers [0] and [0, 1] -- with every layer dropped the reference's getAttenF has nothing to concatenate and
    # raises, and so does this model)
    import numpy as np
    np.random.seed(2)
    a, b = m(wav).detach(), m(wav).detach()
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    with torch.no_grad():
        m.eval()
        assert torch.equal(m(wav), m(wav))
    ns = argparse.Namespace(encoder_layers=2, dropout=0.0, activation_dropout=0.0, dropout_input=0.0, encoder_layerdrop=0.0,
                            attention_dropout=0.1)
    torch.save({"args": ns, "model": bsd}, str(path))
    m2 = Model(None, DEV, cp_path=str(path), cfg=cfg, precision="f16").train()
    m2.load_state_dict(sd)
    m2.to(DEV)
    a, b = m2(wav).detach(), m2(wav).detach()
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    with torch.no_grad():
        m2.eval()
        assert torch.equal(m2(wav), m2(wav))


def test_sls_model_optimizer_steps_and_scores_with_the_trained_weights():
    """A few Adam steps of the SLS model through training.train_step (main.py:224-238): the loss falls, and in eval
    mode the library scores with the updated weights and BatchNorm statistics exactly as a fresh model given the
    trained state dict does."""
    from slsmi import training
    cfg = BackboneConfig(layers=2)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16").train()
    m.load_state_dict(synth_state_dict(cfg, "sls", 201, 1234))
    m.to(DEV)
    assert training.freeze_unused(m) >= 2
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-4)
    wav = seeded_wav(4, seed=6).to(DEV)
    labels = torch.tensor([1, 0, 1, 0], device=DEV)
    losses = [float(training.train_step(m, opt, wav, labels)[0]) for _ in range(4)]
    print("[sls train] losses", [f"{v:.4f}" for v in losses])
    assert losses[-1] < losses[0] and len(set(losses)) == len(losses)
    with torch.no_grad():
        got = m.eval()(wav)
        fresh = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16").eval()
        fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        want = fresh.to(DEV)(wav)
    assert torch.equal(got, want)


def test_gradient_norms_against_reference_fixture(golden_dir):
    """The same batch and loss as tests/golden/make_golden_grads.py ran through the reference's own
    modules with autograd: every gradient norm within 5 %."""
    import os
    import numpy as np
    z = np.load(os.path.join(golden_dir, "grads_xlsr2.npz"))
    cfg = BackboneConfig(layers=2)
    sd = synth_state_dict(cfg, "sae", 201, 1234)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
    m.load_state_dict(sd)
    m.to(DEV)
    wav = seeded_wav(int(z["B"]), int(z["L"]), int(z["seed"])).to(DEV)
    logp, sae_loss = m(wav, dropout_mask=torch.ones(int(z["B"]), 256, device=DEV))
    loss = m.compute_total_loss(F.nll_loss(logp, torch.from_numpy(z["labels"]).to(DEV),
                                           weight=torch.tensor([0.1, 0.9], device=DEV)), sae_loss)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 5e-3
    grads = dict(m.named_parameters())
    worst = 0.0
    for n in (str(v) for v in z["names"]):
        if n.endswith("k_proj.bias"):
            continue
        ratio = float(grads[n].grad.float().norm()) / float(z["norm:" + n])
        worst = max(worst, abs(ratio - 1))
        assert abs(ratio - 1) < 0.05, (n, ratio)
    print(f"[grad norms vs reference] worst deviation {worst:.4f}")


def test_gradients_at_the_recipe_depth_against_reference_fixture(golden_dir):
    """main.py fine-tunes all 24 encoder layers at batch 5 (main.py:201-258, model.py:209-210; README.md:46).  The
    same batch, weights and loss as `tests/golden/make_golden_grads.py deep` ran through the reference's own modules
    with autograd (grads_xlsr24.npz: every gradient's norm and 64 evenly spaced values): every tensor's norm within
    5 % and the slice's cosine >= 0.99 -- conv layer 0 and encoder layer 0, 24 bfloat16 backward layers away from the
    loss, included.  Also against the oracle's whole gradients (autograd on the host, which the fixture pins to
    4.7e-6): cosine >= 0.99 over every whole tensor.  Prints where the error peaks."""
    import os
    import numpy as np
    z = np.load(os.path.join(golden_dir, "grads_xlsr24.npz"))
    layers, B, L = int(z["layers"]), int(z["B"]), int(z["L"])
    assert (layers, B, L) == (24, 5, 64600)
    cfg = BackboneConfig(layers=layers)
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), int(z["seed"]))
    wav = seeded_wav(B, L, int(z["wav_seed"]))
    labels = torch.from_numpy(z["labels"])
    weight = torch.tensor([0.1, 0.9])
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
    m.load_state_dict(sd)
    m.to(DEV)
    logp, sae_loss = m(wav.to(DEV), dropout_mask=torch.ones(B, 256, device=DEV))
    loss = m.compute_total_loss(F.nll_loss(logp, labels.to(DEV), weight=weight.to(DEV)), sae_loss)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(z["loss"])) < 5e-3, (float(loss.detach()), float(z["loss"]))
    assert abs(float(sae_loss.detach()) - float(z["sae_loss"])) < 1e-2 * float(z["sae_loss"])
    dev = {n: p.grad.detach().float().cpu().flatten() for n, p in m.named_parameters() if p.grad is not None}
    del m
    # the oracle's whole gradients, on the host
    _, _, _, ref = oracle_grads(sd, wav, orc.OracleConfig(layers=layers), labels, weight, 0.1, torch.ones(B, 256))
    names = [str(n) for n in z["names"]]
    assert len(names) == 16 * layers + 4 * 7 + 2 + 2 + 3 + 2 + 10   # layers, conv blocks, feature LN, projection, pos_conv, final LN, head
    rows = []
    for n in names:
        assert n in dev, n
        a = dev[n]
        assert torch.isfinite(a).all(), n
        if n.endswith("k_proj.bias"):   # true gradient zero: rounding noise on both sides (see test_training_step_gradients)
            assert float(a.norm()) < 0.02 * float(z["norm:" + n.replace("k_proj", "v_proj")]), n
            continue
        ref_norm = float(z["norm:" + n])
        idx = torch.from_numpy(np.floor(np.arange(64, dtype=np.float64) * (a.numel() - 1) / 63).astype(np.int64)) \
            if a.numel() > 64 else torch.arange(a.numel())
        sl_dev, sl_ref = a[idx].double(), torch.from_numpy(z["slice:" + n]).double()
        cos_slice = float(torch.dot(sl_dev, sl_ref) / (sl_dev.norm() * sl_ref.norm() + 1e-300))
        b = ref[n].flatten()
        cos_full = float(torch.dot(a.double(), b.double()) / (a.double().norm() * b.double().norm() + 1e-300))
        rows.append((cos_full, cos_slice, float(a.norm()) / ref_norm, n))
    rows.sort()
    for cos_full, cos_slice, ratio, n in rows[:8]:
        print(f"[24-layer grad] cos {cos_full:.5f} (slice {cos_slice:.5f}) norm ratio {ratio:.4f} {n}")
    by_ratio = max(rows, key=lambda r: abs(r[2] - 1))
    print(f"[24-layer grad] largest norm deviation {by_ratio[2] - 1:+.4f} at {by_ratio[3]}; {len(rows)} tensors")
    bad = [r for r in rows if r[0] < 0.99 or r[1] < 0.99 or abs(r[2] - 1) > 0.05]
    assert not bad, bad[:10]


def test_layerdrop_steps_like_torch_adam_with_weight_decay():
    """Two optimizer steps of the recipe (main.py:526: Adam with weight decay) under layerdrop, a different layer
    skipped in each: the skipped layer's parameters must not move at all in that step (torch.optim.Adam skips a
    parameter whose grad is None: no weight decay, no momentum, no step count), and slsmi.training.Adam -- one step
    count per tensor -- must do exactly what torch.optim.Adam does from the same gradients."""
    from slsmi.training import Adam
    layers, B, L = 3, 2, 32000
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    sd = synth_state_dict(cfg, "sae", T, 1234)
    wav = seeded_wav(B, L, seed=15).to(DEV)
    labels = torch.tensor([1, 0], device=DEV)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, use_sae=True).train()
    m.load_state_dict(sd)
    m.to(DEV)
    twin = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)   # the class's default betas; a decay that shows
    mine = Adam(m.parameters(), **kw)
    ref = torch.optim.Adam(list(twin.values()), foreach=False, fused=False, **kw)
    for step, skipped in enumerate((1, 2)):
        keep = torch.tensor([l != skipped for l in range(layers)])
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        mine.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        logp, sae_loss = m(wav, dropout_mask=torch.ones(B, 256, device=DEV), backbone_masks={"layer_keep": keep})
        m.compute_total_loss(F.nll_loss(logp, labels), sae_loss).backward()
        for n, p in m.named_parameters():
            if p.grad is not None:
                twin[n].grad = p.grad.detach().clone()
        mine.step()
        ref.step()
        for n, p in m.named_parameters():
            if f".encoder.layers.{skipped}." in n:
                assert p.grad is None and torch.equal(p.detach(), before[n]), n          # untouched: no decay either
            torch.testing.assert_close(p.detach(), twin[n].detach(), rtol=2e-6, atol=2e-8)
    # per-tensor step counts: the layers skipped once are one step behind
    for n, p in m.named_parameters():
        if p in mine.state and mine.state[p]:
            want = 1 if (".encoder.layers.1." in n or ".encoder.layers.2." in n) else 2
            assert int(mine.state[p]["step"]) == want == int(ref.state[twin[n]]["step"]), n


@pytest.mark.parametrize("head,precision", [("sae", "f16"), ("sls", "f16"), ("sae", "bf16")])
def test_adam_step_inside_the_backward_equals_backward_then_step(head, precision):
    """slsmi.training.Adam(fuse_backward=model): the encoder layers' weight matrices take their Adam step in the
    epilogue of their weight-gradient kernel (slsmi_train_set_fused_adam) and never get a gradient.  Two models from
    one state dict, three steps of main.py's loop (main.py:224-238) each, the second step with a layer skipped by
    layerdrop: every parameter, every moment and every step count of the fused run EQUALS the plain run's, bit for
    bit; a state dict of one continues in the other."""
    from slsmi.training import Adam
    layers, B, L = 3, 3, 64600
    cfg = BackboneConfig(layers=layers)
    sd = synth_state_dict(cfg, head, 201, 1234)
    wav = seeded_wav(B, L, seed=21).to(DEV)
    labels = torch.tensor([1, 0, 1], device=DEV)
    kw = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)

    def build(fuse):
        m = (Model(None, DEV, cp_path=None, cfg=cfg, precision=precision) if head == "sls" else
             Model(None, DEV, cp_path=None, cfg=cfg, precision=precision, use_sae=True)).train()
        m.load_state_dict(sd)
        m.to(DEV)
        return m, Adam(m.parameters(), fuse_backward=m if fuse else None, **kw)

    def step(m, opt, skipped):
        masks = None if skipped is None else {"layer_keep": torch.tensor([l != skipped for l in range(layers)])}
        kwf = {} if head == "sls" else {"dropout_mask": torch.ones(B, 256, device=DEV)}
        out = m(wav, backbone_masks=masks, **kwf)
        logp, sae_loss = (out, None) if head == "sls" else out
        loss = F.nll_loss(logp, labels)
        if sae_loss is not None:
            loss = m.compute_total_loss(loss, sae_loss)
        opt.zero_grad()
        loss.backward()
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
        opt.step()
        return grads

    (plain, o_plain), (fused, o_fused) = build(False), build(True)
    is_fused = lambda n: ".encoder.layers." in n and n.endswith((  # noqa: E731
        "_proj.weight", "fc1.weight", "fc2.weight")) and ".self_attn_layer_norm" not in n
    for it, skipped in enumerate((None, 1, None)):
        gp, gf = step(plain, o_plain, skipped), step(fused, o_fused, skipped)
        for n in gp:
            if is_fused(n):
                assert gf[n] is None, n                       # never stored
                assert (gp[n] is None) == (skipped is not None and f".layers.{skipped}." in n), n
            else:
                assert (gp[n] is None) == (gf[n] is None), n
                if gp[n] is not None:
                    assert torch.equal(gp[n], gf[n]), (it, n)
        for (n, a), (_, b) in zip(plain.named_parameters(), fused.named_parameters()):
            assert torch.equal(a.detach(), b.detach()), (it, n)
    n_fused = 0
    for (n, a), (_, b) in zip(plain.named_parameters(), fused.named_parameters()):
        sa, sb = o_plain.state.get(a, {}), o_fused.state.get(b, {})
        assert bool(sa) == bool(sb), n
        if sa:
            assert int(sa["step"]) == int(sb["step"]) == (2 if ".encoder.layers.1." in n else 3), n
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), n
            n_fused += is_fused(n)
    assert n_fused == 6 * layers
    # the state moves between the two forms (a copy: torch's load_state_dict keeps the tensors it is given)
    import copy
    o_plain.load_state_dict(copy.deepcopy(o_fused.state_dict()))
    step(plain, o_plain, None)
    step(fused, o_fused, None)
    for (n, a), (_, b) in zip(plain.named_parameters(), fused.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), n
    # the backward also wrote the library's own 16-bit copies of the stepped tensors (the next reload skips them):
    # the trained model scores like a fresh model given its state dict, bit for bit ...
    def scores(m):
        m.eval()
        with torch.no_grad():
            out = m(wav)
        m.train()
        return (out if head == "sls" else out[0]).clone()

    def fresh_like(m):
        f = (Model(None, DEV, cp_path=None, cfg=cfg, precision=precision) if head == "sls" else
             Model(None, DEV, cp_path=None, cfg=cfg, precision=precision, use_sae=True))
        f.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        return f.to(DEV)

    assert torch.equal(scores(fused), scores(fresh_like(fused)))
    # ... and a change by other means between a backward and the next forward is not lost
    step(fused, o_fused, None)
    with torch.no_grad():
        dict(fused.named_parameters())["ssl_model.model.encoder.layers.0.fc1.weight"].mul_(0.5)
    assert torch.equal(scores(fused), scores(fresh_like(fused)))
    # the library refuses names the fused step does not exist for
    eng = fused._engine
    w = fused.fc0.weight if head == "sls" else fused.classifier[1].weight
    with pytest.raises(slsmi.SlsmiError):
        eng.set_fused_adam([("ssl_model.model.encoder.layers.0.fc1.bias", w.data, w.data.clone(), w.data.clone(), 1)], 1e-3,
                           (0.9, 0.999), 1e-8, 0.0)
    with pytest.raises(slsmi.SlsmiError):
        eng.set_fused_adam([("ssl_model.model.encoder.layers.7.fc1.weight", w.data, w.data.clone(), w.data.clone(), 1)], 1e-3,
                           (0.9, 0.999), 1e-8, 0.0)


def test_training_step_beyond_the_batched_kernels_table_sizes():
    """More clips than the per-launch tables of the weight-gradient kernel hold entries (32): the conv stack's weight
    gradients are one batch entry per clip, so a step at 40 clips walks past the tables of the layer-batched launches
    (bias-gradient pointers, fused-step tensors).  One step, plain and with the Adam step inside the backward: finite,
    and equal to each other bit for bit."""
    from slsmi.training import Adam
    layers, B, L = 1, 40, 16000
    cfg = BackboneConfig(layers=layers)
    T = cfg.num_frames(L)
    sd = synth_state_dict(cfg, "sae", T, 1234)
    wav = seeded_wav(B, L, seed=33).to(DEV)
    labels = (torch.arange(B, device=DEV) % 2)
    out = []
    for fuse in (False, True):
        m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, use_sae=True).train()
        m.load_state_dict(sd)
        m.to(DEV)
        opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-3, fuse_backward=m if fuse else None)
        logp, sae_loss = m(wav, dropout_mask=torch.ones(B, 256, device=DEV))
        opt.zero_grad()
        m.compute_total_loss(F.nll_loss(logp, labels), sae_loss).backward()
        opt.step()
        torch.cuda.synchronize()
        out.append({n: p.detach().clone() for n, p in m.named_parameters()})
    for n, a in out[0].items():
        assert torch.isfinite(a).all(), n
        assert torch.equal(a, out[1][n]), n
    moved = sum(float((out[0][n] - sd[n].to(DEV)).abs().max()) > 0 for n in out[0] if n in sd and "encoder.layers" in n)
    assert moved >= 12   # the layer's matrices and vectors all took a step


def test_sharded_adam_on_the_device_steps_like_torch_adam():
    """slsmi.training.ShardedAdam with the HIP kernel on one rank (the exchange itself is covered by the two-rank
    gloo test on the host): the model's parameters become views of two flat buckets, the first bucket's pack and
    step run on a side stream behind the event the backward records once the encoder layers' gradients are
    complete, and four steps of the training recipe move every parameter exactly as torch.optim.Adam moves a twin
    from the same gradients; the library scores with the updated weights."""
    from slsmi.training import ShardedAdam
    layers, B, L = 2, 4, 64600
    cfg = BackboneConfig(layers=layers)
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), 1234)
    wav = seeded_wav(B, L, seed=6).to(DEV)
    labels = torch.tensor([1, 0, 1, 0], device=DEV)
    weight = torch.tensor([0.1, 0.9], device=DEV)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, use_sae=True).train()
    m.load_state_dict(sd)
    m.to(DEV)
    from slsmi.training import freeze_unused
    freeze_unused(m)
    kw = dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)   # (a larger decay would outweigh these gradients)
    opt = ShardedAdam(m.parameters(), model=m, kernel="hip", **kw)
    assert len(opt.buckets) == 2 and opt._side is not None
    twin = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters() if p.requires_grad}
    ref = torch.optim.Adam(list(twin.values()), foreach=False, fused=False, **kw)
    losses = []
    for step in range(4):
        opt.zero_grad()
        ref.zero_grad(set_to_none=True)
        logp, sae_loss = m(wav, dropout_mask=torch.ones(B, 256, device=DEV))
        loss = m.compute_total_loss(F.nll_loss(logp, labels, weight=weight), sae_loss)
        loss.backward()
        losses.append(float(loss.detach()))
        for n, p in m.named_parameters():
            if p.grad is not None:
                twin[n].grad = p.grad.detach().clone()
        opt.step()
        ref.step()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if n in twin:
                torch.testing.assert_close(p.detach(), twin[n].detach(), rtol=2e-6, atol=2e-9, msg=n)
    assert losses[-1] < losses[0], losses
    # the scoring forward sees the stepped weights: same scores as a fresh model given the state dict
    m.eval()
    fresh = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", num_samples=L, use_sae=True).eval()
    fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    with torch.no_grad():
        assert torch.equal(m(wav, return_sae_loss=False), fresh(wav, return_sae_loss=False))


@pytest.mark.parametrize("betas", [(0.9, 0.98), (0.9, 0.999)])   # (the second: torch's defaults, what main.py:526 runs with)
def test_adam_step_in_one_launch_against_torch(betas):
    """slsmi.training.Adam (slsmi_adam_step: every tensor's step in one launch) against torch.optim.Adam's
    single-tensor form on the same parameters and gradients, three steps with weight decay: tensors of sizes
    that are no multiple of the chunk or of four, and slices of a flat buffer that are not 16-byte aligned (the
    scalar path).  Stated tolerance: 2e-6 relative on parameters and moments (the two differ in one rounding of
    the bias-corrected denominator and in where products are fused; moments that are sums of cancelling terms
    within 2e-7 of the tensor's largest value).  A state dict of one continues in the other."""
    from slsmi.training import Adam
    g = torch.Generator().manual_seed(5)
    flat = torch.randn(70000, generator=g)
    shapes = [(1024, 1024), (3, 5), (1,), (16385,), (4099, 7)]
    base = [torch.randn(*sh, generator=g) for sh in shapes] + [flat[1:30001], flat[30003:70000]]
    tview = torch.randn(33, 130, generator=g)

    def make():
        buf = flat.clone().to(DEV)
        ps = [t.clone().to(DEV).requires_grad_(True) for t in base[:len(shapes)]]
        ps += [buf[1:30001].detach().requires_grad_(True), buf[30003:70000].detach().requires_grad_(True)]
        ps.insert(0, tview.clone().to(DEV).t().detach().requires_grad_(True))   # a transposed view, as sae.encoder.weight
        return ps

    mine, ref = make(), make()
    assert not mine[0].is_contiguous()
    assert mine[-1].data_ptr() % 16 != 0
    kw = dict(lr=3e-3, betas=betas, eps=1e-8, weight_decay=1e-2)
    o_mine, o_ref = Adam(mine, **kw), torch.optim.Adam(ref, foreach=False, fused=False, **kw)
    for step in range(3):
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** (step - 1))
            a.grad, b.grad = torch.empty_like(a).copy_(gr), torch.empty_like(b).copy_(gr)   # the parameter's layout
        o_mine.step()
        o_ref.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-8)   # (steps of ~3e-3: 2e-8 is ~1e-5 of a step)
        sa, sb = o_mine.state[a], o_ref.state[b]
        assert int(sa["step"]) == int(sb["step"]) == 3
        for k in ("exp_avg", "exp_avg_sq"):   # (sums of terms that cancel: absolute bound from the tensor's scale)
            torch.testing.assert_close(sa[k], sb[k], rtol=2e-6, atol=2e-7 * float(sb[k].abs().max()))
    # torch's state continues in this optimizer, and back
    o2 = Adam(mine, **kw)
    o2.load_state_dict(o_ref.state_dict())
    o3 = torch.optim.Adam(ref, foreach=False, fused=False, **kw)
    o3.load_state_dict(o_mine.state_dict())
    for a, b in zip(mine, ref):
        gr = torch.randn(a.shape, generator=g).to(DEV)
        a.grad, b.grad = torch.empty_like(a).copy_(gr), torch.empty_like(b).copy_(gr)
    o2.step()
    o3.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=4e-6, atol=2e-8)
        assert int(o2.state[a]["step"]) == 4
    with pytest.raises(slsmi.SlsmiError):
        bad = torch.zeros(8, dtype=torch.float64, device=DEV, requires_grad=True)
        bad.grad = torch.zeros_like(bad)
        Adam([bad]).step()


@pytest.mark.parametrize("fused", [False, True, "slsmi"])
def test_optimizer_step_lowers_the_loss(fused):
    """A few Adam steps on one batch through the HIP training step (main.py:526, :236-238): the loss falls,
    and the library scores with the updated weights.  torch's fused Adam changes the parameters without
    raising their version counters: the step must be seen all the same."""
    cfg = BackboneConfig(layers=2)
    sd = synth_state_dict(cfg, "sae", 201, 1234)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
    m.load_state_dict(sd)
    m.to(DEV)
    if fused == "slsmi":   # the library's own step: parameters change without raising version counters, as with fused=True
        from slsmi.training import Adam
        opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=fused)
    wav = seeded_wav(4, seed=6).to(DEV)
    labels = torch.tensor([1, 0, 1, 0], device=DEV)
    weight = torch.tensor([0.1, 0.9], device=DEV)
    mask = torch.ones(4, 256, device=DEV)
    losses = []
    for _ in range(4):
        logp, sae_loss = m(wav, dropout_mask=mask)
        loss = m.compute_total_loss(F.nll_loss(logp, labels, weight=weight), sae_loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("[train] losses", [f"{v:.4f}" for v in losses])
    assert losses[-1] < losses[0]
    assert len(set(losses)) == len(losses)  # every step saw new weights
    # the trained model scores as a fresh model given its state dict does
    with torch.no_grad():
        got = m.eval()(wav, return_sae_loss=False)
        fresh = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).eval()
        fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        want = fresh.to(DEV)(wav, return_sae_loss=False)
    assert torch.equal(got, want)


def test_backward_must_follow_its_forward():
    """slsmi_train_backward reads what slsmi_train_forward left in the workspace: a second backward, or one
    for another batch, is refused with an error instead of producing gradients of stale intermediates.
    A [B, L, 1] input (model.py:134-135) trains like the [B, L] one."""
    cfg = BackboneConfig(layers=1)
    m = Model(None, DEV, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
    m.load_state_dict(synth_state_dict(cfg, "sae", 201, 1234))
    m.to(DEV)
    wav = seeded_wav(2, seed=8).to(DEV)
    mask = torch.ones(2, 256, device=DEV)
    logp, sae_loss = m(wav.unsqueeze(-1), dropout_mask=mask)
    (logp.sum() + sae_loss).backward()
    g3 = m.classifier[1].weight.grad.clone()
    m.zero_grad()
    logp, sae_loss = m(wav, dropout_mask=mask)
    (logp.sum() + sae_loss).backward()
    assert torch.equal(g3, m.classifier[1].weight.grad)
    eng = m._engine
    grads = {"classifier.1.weight": torch.empty_like(m.classifier[1].weight)}
    dlogp = torch.ones(2, 2, device=DEV)
    with pytest.raises(slsmi.SlsmiError, match="must follow"):       # second backward of the same forward
        eng.train_backward(wav, dlogp, None, grads, True)
    eng.train_forward(wav, mask, True)
    with pytest.raises(slsmi.SlsmiError, match="must follow"):       # another batch size
        eng.train_backward(wav[:1].contiguous(), dlogp[:1].contiguous(), None, grads, True)


DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
log = open(os.path.join(sys.argv[1], "gpurun_out", f"dp_rank{os.environ.get('RANK', '0')}.log"), "w")
def note(msg):
    log.write(msg + "\\n"); log.flush()
import torch, torch.distributed as dist
import slsmi
from slsmi.config import BackboneConfig
from slsmi.model import Model
from slsmi.weights import synth_state_dict
from slsmi import training
dist.init_process_group("gloo", init_method="env://")     # two ranks on the one GPU of the test box
rank, world = dist.get_rank(), dist.get_world_size()
note("process group up")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = BackboneConfig(layers=1)
m = Model(None, dev, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
m.load_state_dict(synth_state_dict(cfg, "sae", 201, 1234))
m.to(dev)
assert training.freeze_unused(m) > 0
params = [p for p in m.parameters() if p.requires_grad]
sync = training.GradSync(params, comm_device=torch.device("cpu"))   # gloo: the exchange goes through host memory
opt = torch.optim.Adam(params, lr=1e-5, weight_decay=1e-4)
g = torch.Generator().manual_seed(100 + rank)              # every rank its own utterances
wav = (0.1 * torch.randn(3, 64600, generator=g)).clamp(-1, 1).to(dev)
y = torch.tensor([1, 0, 1], device=dev) if rank == 0 else torch.tensor([0, 0, 1], device=dev)
mask = torch.ones(3, 256, device=dev)
losses = []
for step in range(3):
    total, cls, sae = training.train_step(m, opt, wav, y, dropout_mask=mask, grad_sync=sync)
    losses.append(float(total))
    note(f"step {step} loss {losses[-1]:.4f}")
    if step == 0:   # the averaged gradients are the same on both ranks
        gn = torch.stack([p.grad.float().norm() for p in params if p.grad is not None]).cpu()
        both = [torch.empty_like(gn) for _ in range(world)]
        dist.all_gather(both, gn)
        assert torch.equal(both[0], both[1]), "gradients differ between ranks"
        assert len(gn) >= 16 + 14 + 30
# identical updates from identical gradients: the replicas stay in step
chk = torch.stack([p.detach().float().sum() for p in params]).cpu()
both = [torch.empty_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1]), "replicas diverged"
print(f"rank {rank} losses {losses}", flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_data_parallel_two_ranks_train_in_step(tmp_path):
    """Two ranks (gloo, both on the test box's one GPU), every rank its own batch, GradSync averaging the
    gradients once per step: both ranks hold the same gradients and apply identical updates."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(script), root]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "rank 0 losses" in r.stdout and "rank 1 losses" in r.stdout
