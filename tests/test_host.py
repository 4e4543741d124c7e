"""Host logic that needs no GPU: drop-in surface, key inventory, C-ABI symbols,
sharding and the gloo (CPU) rehearsal of the multi-rank score gather."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import slsmi
from slsmi import _lib
from slsmi.config import BackboneConfig, XLSR_300M, flops_per_utt
from slsmi.weights import synth_state_dict, backbone_shapes, strip_module_prefix, unwrap_checkpoint
from slsmi import scoring

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frames_and_flops():
    assert XLSR_300M.conv_out_lengths(64600) == [12919, 6459, 3229, 1614, 806, 403, 201]
    assert XLSR_300M.num_frames(160000) == 499
    assert flops_per_utt(XLSR_300M, 64600, "sls") / 1e9 == pytest.approx(148.82, abs=0.01)
    n_params = sum(int(np.prod(s)) for k, s in backbone_shapes(XLSR_300M, leftovers=False).items())
    assert n_params / 1e6 == pytest.approx(315.44, abs=0.6)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "slsmi.h")).read()
    declared = set(re.findall(r"\b(slsmi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"slsmi_status"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.slsmi_version() >= 100
    cfg = _lib.Config()
    lib.slsmi_default_config(ctypes.byref(cfg), _lib.HEAD_SLS)
    assert (cfg.n_conv, cfg.embed_dim, cfg.layers, cfg.ffn_dim) == (7, 1024, 24, 4096)
    assert list(cfg.conv_kernel)[:7] == [10, 3, 3, 3, 3, 2, 2]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libslsmi.so")
    with pytest.raises(_lib.SlsmiError):
        _lib.load()


def test_model_surface_and_keys():
    import model as dropin  # the root-level drop-in module
    assert {"Model", "SSLModel", "AutoEncoderTopK", "getAttenF"} <= set(dir(dropin))
    cfg = BackboneConfig(layers=1)
    m = dropin.Model(None, "cpu", cp_path=None, cfg=cfg)  # upstream form -> SLS head
    sd = synth_state_dict(cfg, "sls", 201, 7)
    assert set(m.state_dict().keys()) == set(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    # DataParallel-style checkpoint (main.py:531-560)
    wrapped = {"model_state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 3}
    m.load_state_dict(wrapped, strict=True)
    assert torch.equal(m.fc0.weight, sd["fc0.weight"])
    ms = dropin.Model(None, "cpu", cp_path=None, use_sae=True, use_sparse_features=True, sae_dict_size=4096,
                      sae_k=128, sae_weight=0.1, cfg=cfg)  # main.py:508-517 form -> SAE head
    sd2 = synth_state_dict(cfg, "sae", 201, 7)
    assert set(ms.state_dict().keys()) == set(sd2.keys())
    assert ms.compute_total_loss(torch.tensor(1.0), torch.tensor(2.0)).item() == pytest.approx(1.2)
    assert m.compute_total_loss(torch.tensor(1.0), torch.tensor(2.0)).item() == 1.0
    with pytest.raises(RuntimeError):
        dropin.Model(None, "cpu", cp_path="/nonexistent/xlsr2_300m.pt", cfg=cfg)


def test_multi_gpu_dataparallel_replica_is_refused():
    """A replica made by nn.DataParallel over several devices fails with directions, not silently slowly."""
    from slsmi.config import BackboneConfig
    from slsmi.model import Model
    m = Model(None, "cpu", cp_path=None, cfg=BackboneConfig(layers=1))
    m._is_replica = True            # what torch.nn.parallel.replicate sets on its copies
    with pytest.raises(slsmi.SlsmiError, match="one process per GPU"):
        m(torch.zeros(1, 64600))


def test_engine_follows_parameter_updates(monkeypatch):
    """When Model hands its weights to the engine again (host logic, with a stand-in engine): a new state dict or
    moved parameters -> full load; in-place updates that raise version counters -> one reload call; a backward
    (after which a fused optimizer may have stepped without raising them) -> one reload call; nothing -> nothing."""
    from slsmi import model as model_mod

    class FakeEngine:
        def __init__(self, device, *a, **kw):
            self.device, self.calls, self.reloadable, self.training_enabled = torch.device(device), [], True, False

        def load_state_dict(self, sd):
            self.calls.append("load")

        def reload_weights(self):
            self.calls.append("reload")

        def enable_training(self):
            self.training_enabled = True

    monkeypatch.setattr(model_mod, "Engine", FakeEngine)
    m = model_mod.Model(None, "cpu", cp_path=None, cfg=BackboneConfig(layers=1), use_sae=True)
    eng = m._get_engine("cpu")
    assert eng.calls == ["load"]
    m._get_engine("cpu")
    assert eng.calls == ["load"]                          # nothing changed
    with torch.no_grad():
        m.classifier[0].weight.add_(1.0)                  # in place: version counter moves, address stays
    m._get_engine("cpu")
    assert eng.calls == ["load", "reload"]
    m._engine_dirty = True                                # what the backward of the training step leaves behind
    m._get_engine("cpu")
    assert eng.calls == ["load", "reload", "reload"] and not m._engine_dirty
    m.classifier[0].weight.data = m.classifier[0].weight.data.clone()   # another buffer
    m._get_engine("cpu")
    assert eng.calls[-1] == "load"
    eng.reloadable = False                                # weights that came from temporary copies
    m._engine_dirty = True
    m._get_engine("cpu")
    assert eng.calls[-1] == "load" and eng.calls.count("reload") == 2
    m._get_engine("cpu", training=True)                   # switching to training loads everything again
    assert eng.training_enabled and eng.calls[-1] == "load" and eng.calls.count("load") == 4


def test_no_cpu_fallback():
    from slsmi.model import Model
    m = Model(None, "cpu", cp_path=None, cfg=BackboneConfig(layers=1)).eval()
    with torch.no_grad(), pytest.raises(_lib.SlsmiError):
        m(torch.zeros(1, 64600))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "slsforasvspoof-2021-df_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
    assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(ROOT, "model.py")).read(), re.M)


def test_getAttenF_matches_definition():
    from slsmi.model import getAttenF
    layers = [(torch.randn(5, 2, 8), None) for _ in range(3)]
    y, full = getAttenF(layers)
    assert y.shape == (2, 3, 8) and full.shape == (2, 3, 5, 8)
    torch.testing.assert_close(y[:, 1], layers[1][0].mean(0))
    torch.testing.assert_close(full[:, 2], layers[2][0].transpose(0, 1))


def test_shard_merge_roundtrip():
    for n, w in [(10, 1), (10, 3), (611829, 8), (5, 8), (0, 2)]:
        truth = np.arange(n, dtype=np.float32) * 0.5
        shards = [truth[scoring.shard_indices(n, r, w)] for r in range(w)]
        np.testing.assert_array_equal(scoring.merge_shards(n, w, shards), truth)
    with pytest.raises(ValueError):
        scoring.merge_shards(4, 2, [np.zeros(1), np.zeros(2)])


def test_checkpoint_unwrap():
    sd = {"module.a": torch.zeros(1)}
    assert list(strip_module_prefix(sd)) == ["a"]
    assert unwrap_checkpoint({"state_dict": sd}) is sd
    assert unwrap_checkpoint(sd) is sd


GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import slsmi
from slsmi import scoring
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
n = 23
ids = [f"utt{i:03d}" for i in range(n)]

class Fake(torch.nn.Module):           # stands in for Model: log-probs from the input itself
    def forward(self, x, return_sae_loss=False):
        p = torch.sigmoid(x[:, 0:1])
        return torch.log(torch.cat([1 - p, p], dim=1))

def load_batch(idx):
    return torch.tensor(np.asarray(idx, dtype=np.float32))[:, None].repeat(1, 4) * 0.1

out = scoring.score_sharded(Fake(), ids, load_batch, torch.device("cpu"), batch_size=4,
                            save_path=sys.argv[2] if rank == 0 else None)
if rank == 0:
    want = torch.sigmoid(torch.arange(n, dtype=torch.float32) * 0.1).numpy()
    assert np.allclose(out, want, atol=1e-6), (out, want)
    lines = open(sys.argv[2]).read().splitlines()
    assert len(lines) == n and lines[5].split(" ")[0] == "utt005"
    assert all(len(l.split(" ")) == 2 for l in lines)
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


GLOO_WORKER_VARLEN = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import slsmi
from slsmi import scoring, data
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)
n = 37
lengths = rng.integers(16000, 160001, size=n)
ids = [f"wild{i:03d}.wav" for i in range(n)]
batches = data.bucket_batches(lengths, batch_size=5)[rank::world]

class Fake(torch.nn.Module):           # log-probs from each clip's own samples and length
    def forward(self, x, return_sae_loss=False, lengths=None):
        m = torch.stack([x[b, :int(lengths[b])].sum() / 1e5 for b in range(x.shape[0])])
        p = torch.sigmoid(m)[:, None]
        return torch.log(torch.cat([1 - p, p], dim=1))

def loader():
    for idx, padded in batches:
        x = torch.full((len(idx), padded), 7.0)          # the padding holds junk
        for r, j in enumerate(idx):
            x[r, :lengths[j]] = float(j % 5) - 2.0
        yield x, torch.tensor(lengths[idx].astype(np.int32)), [ids[j] for j in idx], idx

out = scoring.score_variable_length(Fake(), loader(), ids, save_path=sys.argv[2] if rank == 0 else None)
if rank == 0:
    want = torch.sigmoid(torch.tensor([(j % 5 - 2.0) * lengths[j] / 1e5 for j in range(n)], dtype=torch.float32)).numpy()
    assert np.allclose(out, want, atol=1e-5), (out, want)
    lines = open(sys.argv[2]).read().splitlines()
    assert [l.split(" ")[0] for l in lines] == ids
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


def test_gloo_world2_variable_length_scores(tmp_path):
    """Length-bucketed batches dealt out to two ranks; scores come back in protocol order."""
    script = tmp_path / "worker_varlen.py"
    script.write_text(GLOO_WORKER_VARLEN)
    out = tmp_path / "scores_wild.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT, str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(out.read_text().splitlines()) == 37


GLOO_WORKER_GRADSYNC = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import slsmi
from slsmi import training
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
g = torch.Generator().manual_seed(3)
shapes = [(7, 5), (33,), (4, 3, 2)]
base = [torch.randn(s, generator=g) for s in shapes]            # the same on both ranks
params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
for dtype, tol in ((torch.float32, 0.0), (torch.bfloat16, 2 ** -8)):
    for p, b in zip(params, base):
        p.grad = b * (1.0 + rank)                                   # rank 0: b, rank 1: 2 b -> average 1.5 b
    training.GradSync(params + [frozen], comm_dtype=dtype).sync()
    for p, b in zip(params, base):
        err = (p.grad - 1.5 * b).abs().max().item()
        assert err <= tol * 3.0 * b.abs().max().item() + 1e-7, (dtype, err)
    assert frozen.grad is None
dist.barrier()
dist.destroy_process_group()
'''


def test_gloo_world2_gradient_exchange(tmp_path):
    """GradSync: one flat all-reduce averages every gradient over two ranks, exactly in fp32 and to
    bfloat16 rounding with comm_dtype=torch.bfloat16; parameters without a gradient stay out."""
    script = tmp_path / "worker_gradsync.py"
    script.write_text(GLOO_WORKER_GRADSYNC)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29535", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gloo_world2_score_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    out = tmp_path / "scores.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script), ROOT, str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(out.read_text().splitlines()) == 23
