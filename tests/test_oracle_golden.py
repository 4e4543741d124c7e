"""The CPU oracle against the fixtures generated from the reference's own code
(tests/golden/make_golden.py).  Weights are re-created from the seed."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

import slsmi
from slsmi.config import BackboneConfig
from slsmi.weights import synth_state_dict
from oracle import reference_cpu as orc

TINY = BackboneConfig(conv_layers=((32, 10, 5),) + ((32, 3, 2),) * 4 + ((32, 2, 2),) * 2,
                      embed_dim=64, ffn_dim=128, heads=4, layers=2, conv_pos=16, conv_pos_groups=4,
                      final_dim=32, latent_vars=8, latent_groups=2)


def ocfg(cfg, sae_dict=4096, sae_k=128):
    return orc.OracleConfig(conv_layers=cfg.conv_layers, embed_dim=cfg.embed_dim, ffn_dim=cfg.ffn_dim,
                            heads=cfg.heads, layers=cfg.layers, conv_pos=cfg.conv_pos,
                            conv_pos_groups=cfg.conv_pos_groups, sae_dict=sae_dict, sae_k=sae_k)


def seeded_wav(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)


def test_tiny_full_tensors(golden_dir):
    z = np.load(os.path.join(golden_dir, "tiny_case.npz"))
    wav = torch.from_numpy(z["wav"])
    assert torch.equal(wav, seeded_wav(2, 3200, 1234))
    T = TINY.num_frames(3200)
    assert list(z["conv_lengths"]) == TINY.conv_out_lengths(3200)
    oc = ocfg(TINY, 256, 16)
    sd = synth_state_dict(TINY, "sae", T, 1234, 256, 16)
    logp, bb = orc.model_forward(sd, wav, oc, "sae", return_intermediates=True)
    np.testing.assert_allclose(bb["x"].numpy(), z["x"], atol=5e-6)
    np.testing.assert_allclose(bb["features"].numpy(), z["features"], atol=5e-6)
    np.testing.assert_allclose(torch.stack(bb["layer_results"]).numpy(), z["layer_results"], atol=5e-6)
    np.testing.assert_allclose(logp.numpy(), z["logp_sae"], atol=2e-6)
    _, loss = orc.sae_head(sd, bb["x"], oc, return_sae_loss=True)
    assert abs(float(loss) - float(z["sae_loss"])) < 1e-6
    sd = synth_state_dict(TINY, "sls", T, 1234, 256, 16)
    np.testing.assert_allclose(orc.model_forward(sd, wav, oc, "sls").numpy(), z["logp_sls"], atol=5e-6)


@pytest.mark.parametrize("name,layers", [("xlsr_2layer_case", 2), ("xlsr_case", 24)])
def test_xlsr_geometry(golden_dir, name, layers):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = BackboneConfig(layers=layers)
    wav = seeded_wav(int(z["B"]), int(z["L"]), int(z["seed"]))
    assert list(z["conv_lengths"]) == [12919, 6459, 3229, 1614, 806, 403, 201]
    oc = ocfg(cfg)
    sd = synth_state_dict(cfg, "sae", 201, 1234)
    logp, bb = orc.model_forward(sd, wav, oc, "sae", return_intermediates=True)
    np.testing.assert_allclose(logp.numpy(), z["logp_sae"], atol=2e-5)
    np.testing.assert_allclose(bb["x"][:, ::50, :8].numpy(), z["x_slice"], atol=5e-5)
    first8 = np.stack([t[0, :, :8].numpy() for t in bb["layer_results"]])
    np.testing.assert_allclose(first8, z["layer_first8"], atol=2e-4)
    st = np.array([[float(t.mean()), float(t.abs().max()), float(t.std())] for t in bb["layer_results"]])
    np.testing.assert_allclose(st, z["layer_stats"], rtol=1e-4, atol=1e-4)
    sd = synth_state_dict(cfg, "sls", 201, 1234)
    # fp32 summation-order noise through 24 layers reaches 2e-4 on the SLS log-probs
    np.testing.assert_allclose(orc.sls_head(sd, bb["layer_results"]).numpy(), z["logp_sls"], atol=1e-3)


@pytest.mark.parametrize("name,cfg,dict_,k", [("head_variants_tiny", TINY, 256, 16),
                                              ("head_variants_xlsr2", BackboneConfig(layers=2), 4096, 128)])
def test_head_variants_and_interpretability(golden_dir, name, cfg, dict_, k):
    """Model(use_sae=True, use_sparse_features=False), Model(use_sae=False) and
    get_interpretability_info, against vectors from the reference's model.py
    (tests/golden/make_golden_variants.py)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    wav = seeded_wav(int(z["B"]), int(z["L"]), int(z["seed"]))
    T = cfg.num_frames(int(z["L"]))
    oc = ocfg(cfg, dict_, k)
    for head in ("sae_recon", "ssl"):
        sd = synth_state_dict(cfg, head, T, 1234, dict_, k)
        np.testing.assert_allclose(orc.model_forward(sd, wav, oc, head).numpy(), z["logp_" + head], atol=5e-6)
    sd = synth_state_dict(cfg, "sae", T, 1234, dict_, k)
    bb = orc.backbone_forward(sd, wav, oc, pfx="ssl_model.model.")
    logp, _, code = orc.sae_head(sd, bb["x"], oc, return_code=True)
    info = orc.interpretability_info(code)
    np.testing.assert_allclose(logp.numpy(), z["interp_logp"], atol=5e-6)
    assert list(code.shape) == list(z["interp_code_shape"])
    np.testing.assert_allclose(info["avg_activation"].numpy(), z["interp_avg_activation"], atol=1e-5)
    np.testing.assert_array_equal(info["top20_features"].numpy(), z["interp_top20_features"])
    np.testing.assert_allclose(info["top20_values"].numpy(), z["interp_top20_values"], atol=1e-5)
    np.testing.assert_allclose(info["sparsity"].numpy(), z["interp_sparsity"], atol=1e-7)
    np.testing.assert_allclose(info["activation_freq"].numpy(), z["interp_activation_freq"], atol=1e-7)


def test_eer_known_answer(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "eer_known_answer.json")))
    rng = np.random.default_rng(0)
    bona = rng.normal(1, 1, 1000)
    spoof = rng.normal(-1, 1, 9000)
    eer, thr = orc.compute_eer(bona, spoof)
    assert eer == pytest.approx(ka["eer"], abs=1e-15)
    assert thr == pytest.approx(ka["threshold"], abs=1e-15)


def test_pad_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "pad_cases.npz"))
    rng = np.random.default_rng(1)
    from slsmi.scoring import pad
    for n in (70000, 64600, 30000, 7, 64599):
        x = rng.standard_normal(n).astype(np.float32)
        for fn in (orc.pad_audio, pad):
            y = fn(x, 64600)
            assert y.shape == (64600,)
            np.testing.assert_array_equal(y[:64], z[f"out_{n}_head"])
            np.testing.assert_array_equal(y[-64:], z[f"out_{n}_tail"])
            assert float(y.astype(np.float64).sum()) == float(z[f"out_{n}_sum"])


def test_score_lines(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "score_lines.json")))
    sc = np.frombuffer(bytes.fromhex(ka["scores_f32_le_hex"]), dtype="<f4")
    ids = [f"DF_E_{i:07d}" for i in range(len(sc))]
    from slsmi.scoring import format_score_lines
    assert orc.format_score_lines(ids, sc) == ka["lines"]
    assert format_score_lines(ids, sc) == ka["lines"]


WINDOW_CASES = ["t37_w8", "t201_w8", "t40_w6", "t9_w8", "t50_w2"]


@pytest.mark.parametrize("name", WINDOW_CASES)
def test_window_topk_cases(golden_dir, name):
    """window_topk against AutoEncoderTopK._window_topk of the reference's model_window_topk.py
    (tests/golden/make_golden_window.py).  A frame with at least k positive votes is determined by
    the votes; elsewhere the reference keeps whichever zero-vote features its torch.topk returns,
    so there only the positive-vote features and the count of kept features are compared."""
    z = np.load(os.path.join(golden_dir, "window_topk_cases.npz"))
    B, T, D, k, w, seed = (int(v) for v in z[name + "_meta"])
    g = torch.Generator().manual_seed(seed)
    acts = torch.relu(torch.randn(B, T, D, generator=g) - float(z[name + "_shift"]))
    ref = torch.from_numpy(z[name])
    mine = orc.window_topk(acts, k, w)
    cnt = orc.window_vote_counts(acts, k, w)
    full = cnt >= k
    assert torch.equal(mine[full], ref[full])
    votes = orc.window_votes(acts, k, w)
    part = ~full
    assert torch.equal((mine * (votes > 0))[part], (ref * (votes > 0))[part])
    # every frame keeps at most k features, all taken from the frame's own activations
    assert int((ref != 0).sum(-1).max()) <= k
    assert torch.equal(ref[ref != 0], acts[ref != 0])


@pytest.mark.parametrize("tag,cfg,L,dict_,k", [("tiny", TINY, 3200, 256, 16),
                                               ("xlsr2", BackboneConfig(layers=2), 64600, 4096, 128)])
def test_window_topk_model(golden_dir, tag, cfg, L, dict_, k):
    """Model(..., sae_window_size=w) of model_window_topk.py:271-393: log-probs and SAE loss."""
    z = np.load(os.path.join(golden_dir, "window_model.npz"))
    wav = seeded_wav(int(z[tag + "_B"]), L, int(z[tag + "_seed"]))
    oc = dataclasses.replace(ocfg(cfg, dict_, k), sae_window=int(z[tag + "_window"]))
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), 1234, dict_, k)
    bb = orc.backbone_forward(sd, wav, oc, pfx="ssl_model.model.")
    logp, loss = orc.sae_head(sd, bb["x"], oc, return_sae_loss=True)
    np.testing.assert_allclose(logp.numpy(), z[tag + "_logp"], atol=5e-6)
    assert abs(float(loss) - float(z[tag + "_sae_loss"])) < 1e-6


@pytest.mark.parametrize("tag,cfg,full", [("tiny", TINY, True), ("xlsr2", BackboneConfig(layers=2), False)])
def test_variable_length_batch(golden_dir, tag, cfg, full):
    """backbone_forward(lengths=) against the reference's Wav2Vec2Model run with a padding_mask
    (tests/golden/make_golden_varlen.py), on every utterance's own frames; and against the oracle
    run on each utterance alone at its own length, which is what the batched HIP path is tested with."""
    z = np.load(os.path.join(golden_dir, "varlen_cases.npz"))
    L, lengths, frames = int(z[tag + "_L"]), [int(v) for v in z[tag + "_lengths"]], [int(v) for v in z[tag + "_frames"]]
    wav = seeded_wav(len(lengths), L, int(z[tag + "_seed"]))
    for b, n in enumerate(lengths):
        wav[b, n:] = 0.0
    oc = ocfg(cfg)
    sd = synth_state_dict(cfg, "sae", cfg.num_frames(L), 1234, 256 if full else 4096, 16 if full else 128)
    bb = orc.backbone_forward(sd, wav, oc, pfx="ssl_model.model.", lengths=lengths)
    assert [int((~bb["padding_mask"][b]).sum()) for b in range(len(lengths))] == frames
    last = bb["layer_results"][-1].transpose(0, 1)
    for b, f in enumerate(frames):
        if full:
            np.testing.assert_allclose(bb["x"][b, :f].numpy(), z[tag + "_x"][b, :f], atol=1e-5)
            np.testing.assert_allclose(last[b, :f].numpy(), z[tag + "_layers"][-1][b, :f], atol=2e-5)
        else:
            own = (f + 9) // 10  # rows of the [::10] slice inside the utterance's own frames
            np.testing.assert_allclose(bb["x"][b, ::10, :16][:own].numpy(), z[tag + "_x_slice"][b, :own], atol=5e-5)
            np.testing.assert_allclose(last[b, ::10, :16][:own].numpy(), z[tag + "_last_layer_slice"][b, :own],
                                       atol=1e-4)
        alone = orc.backbone_forward(sd, wav[b:b + 1, :lengths[b]], oc, pfx="ssl_model.model.")
        np.testing.assert_allclose(alone["x"][0].numpy(), bb["x"][b, :f].numpy(), atol=5e-5)


def test_training_gradients_against_reference_autograd(golden_dir):
    """Autograd through the oracle against autograd through the reference's own model.py / wav2vec2.py
    (tests/golden/make_golden_grads.py): main.py's training loss on a seeded batch, every parameter's
    gradient norm and first values.  This oracle gradient is what the HIP training step is tested with."""
    import torch.nn.functional as F
    z = np.load(os.path.join(golden_dir, "grads_xlsr2.npz"))
    cfg = BackboneConfig(layers=2)
    wav = seeded_wav(int(z["B"]), int(z["L"]), int(z["seed"]))
    sd = synth_state_dict(cfg, "sae", 201, 1234)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    oc = ocfg(cfg)
    bb = orc.backbone_forward(params, wav, oc, pfx="ssl_model.model.")
    logp, sae_loss = orc.sae_head(params, bb["x"], oc, return_sae_loss=True)
    loss = F.nll_loss(logp, torch.from_numpy(z["labels"]), weight=torch.tensor([0.1, 0.9])) + 0.1 * sae_loss
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-5
    names = [str(n) for n in z["names"]]
    assert len(names) == 79
    for n in names:
        g = params[n].grad
        ref_norm = float(z["norm:" + n])
        if n.endswith("k_proj.bias"):      # exactly zero in exact arithmetic: rounding noise on both sides
            assert float(g.norm()) < 1e-7 and ref_norm < 1e-7
            continue
        assert abs(float(g.double().norm()) - ref_norm) <= 1e-4 * ref_norm, n
        np.testing.assert_allclose(g.flatten()[:8].numpy(), z["head:" + n], rtol=2e-3, atol=1e-5 * ref_norm)
