"""End-to-end parity of Model.forward on the HIP path against the CPU oracle
and the golden fixtures, plus size-independent properties at the benchmark
batch size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import slsmi
from slsmi.config import BackboneConfig
from slsmi.model import Model
from slsmi.weights import synth_state_dict
from slsmi import scoring
from oracle import reference_cpu as orc

DEV = torch.device("cuda:0")
SCORE_TOL = 1e-3      # stated tolerance on the per-utterance score exp(logp)[:, 1]


def seeded_wav(B, L=64600, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp(-1, 1)


def build(head, layers, seed=1234):
    cfg = BackboneConfig(layers=layers)
    sd = synth_state_dict(cfg, head, 201, seed)
    m = Model(None, DEV, cp_path=None, head=head, cfg=cfg).eval()
    m.load_state_dict(sd)
    return m, sd, orc.OracleConfig(layers=layers)


@pytest.fixture(scope="module")
def full_sls():
    return build("sls", 24)


@pytest.mark.parametrize("head", ["sls", "sae"])
def test_two_layer_against_oracle_and_golden(head, golden_dir):
    m, sd, oc = build(head, 2)
    wav = seeded_wav(2)
    with torch.no_grad():
        logp = m(wav.to(DEV), return_sae_loss=False).cpu()
    ref = orc.model_forward(sd, wav, oc, head)
    z = np.load(os.path.join(golden_dir, "xlsr_2layer_case.npz"))
    gold = torch.from_numpy(z["logp_" + head])
    print(head, "hip", logp.tolist(), "oracle", ref.tolist(), "golden", gold.tolist())
    assert (torch.exp(logp)[:, 1] - torch.exp(ref)[:, 1]).abs().max() < 5e-3
    assert (torch.exp(logp)[:, 1] - torch.exp(gold)[:, 1]).abs().max() < 5e-3


def test_backbone_intermediates_two_layers():
    m, sd, oc = build("sls", 2)
    wav = seeded_wav(2)
    emb, layers = m.ssl_model.extract_feat(wav.to(DEV), return_layers=True)
    bb = orc.backbone_forward(sd, wav, oc, pfx="ssl_model.model.")
    assert emb.shape == (2, 201, 1024) and len(layers) == 2 and layers[0][0].shape == (201, 2, 1024)
    for l in range(2):
        d = (layers[l][0].cpu() - bb["layer_results"][l]).abs()
        scale = bb["layer_results"][l].abs().max()
        print("layer", l, "max abs err", d.max().item(), "ref absmax", scale.item())
        assert d.max() < 0.03 * scale
    assert (emb.cpu() - bb["x"]).abs().max() < 0.05 * bb["x"].abs().max()


def test_full_model_scores_sls(full_sls, golden_dir):
    m, sd, oc = full_sls
    wav = seeded_wav(2)
    with torch.no_grad():
        logp = m(wav.to(DEV), return_sae_loss=False).cpu()
    z = np.load(os.path.join(golden_dir, "xlsr_case.npz"))
    gold = torch.from_numpy(z["logp_sls"])
    err = (torch.exp(logp)[:, 1] - torch.exp(gold)[:, 1]).abs().max().item()
    print("sls 24 layers: hip", logp.tolist(), "golden", gold.tolist(), "score err", err)
    assert err < SCORE_TOL


def test_full_model_scores_sae(golden_dir):
    m, sd, oc = build("sae", 24)
    wav = seeded_wav(2)
    with torch.no_grad():
        logp, loss = m(wav.to(DEV))  # default arity: (output, sae_loss)
    z = np.load(os.path.join(golden_dir, "xlsr_case.npz"))
    gold = torch.from_numpy(z["logp_sae"])
    err = (torch.exp(logp.cpu())[:, 1] - torch.exp(gold)[:, 1]).abs().max().item()
    print("sae 24 layers: hip", logp.tolist(), "golden", gold.tolist(), "score err", err,
          "sae_loss", float(loss), "golden", float(z["sae_loss"]))
    assert err < SCORE_TOL
    assert abs(float(loss) - float(z["sae_loss"])) < 2e-2 * float(z["sae_loss"])


def test_batch64_properties(full_sls):
    """At the benchmark batch (64): every score finite and in (0,1); an
    utterance's score does not depend on its batch mates or position."""
    m, _, _ = full_sls
    wav = seeded_wav(64, seed=99).to(DEV)
    with torch.no_grad():
        a = torch.exp(m(wav, return_sae_loss=False))[:, 1]
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(0)).to(DEV)
        b = torch.exp(m(wav[perm], return_sae_loss=False))[:, 1]
        c = torch.exp(m(wav[:3], return_sae_loss=False))[:, 1]
    assert torch.isfinite(a).all() and (a > 0).all() and (a < 1).all()
    assert (a[perm] - b).abs().max() < 1e-5
    assert (a[:3] - c).abs().max() < 1e-5


def test_three_dim_input_and_score_file(full_sls, tmp_path):
    m, _, _ = full_sls
    wav = seeded_wav(5, seed=3)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return wav[i], f"DF_E_{i:07d}"

    path = str(tmp_path / "scores.txt")
    scoring.produce_evaluation_file(DS(), m, DEV, path, batch_size=2)
    lines = open(path).read().splitlines()
    assert [l.split(" ")[0] for l in lines] == [f"DF_E_{i:07d}" for i in range(5)]
    with torch.no_grad():
        direct = torch.exp(m(wav.to(DEV)[:, :, None], return_sae_loss=False))[:, 1].cpu().numpy()
    got = np.array([float(l.split(" ")[1]) for l in lines], dtype=np.float32)
    np.testing.assert_allclose(got, direct, atol=1e-5)
    assert all(len(l.split(" ")) == 2 for l in lines)
