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
    # the reference's defaults (model.py:150-160, use_sae=True): the in-tree model's keys
    assert set(dropin.Model(None, "cpu", cp_path=None, cfg=cfg).state_dict().keys()) == \
        set(synth_state_dict(cfg, "sae", 201, 7).keys())
    m = dropin.SLSModel(None, "cpu", cp_path=None, cfg=cfg)  # upstream form -> SLS head
    assert set(dropin.Model(None, "cpu", cp_path=None, cfg=cfg, head="sls").state_dict().keys()) == set(m.state_dict().keys())
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


def test_dataparallel_copy_without_its_module_or_off_device_is_refused():
    """A per-device copy of the module (what nn.DataParallel's replicate makes) scores through the module it was
    copied from; one that has lost that module, or is fed a CPU batch, fails with a message instead of guessing."""
    from slsmi.config import BackboneConfig
    from slsmi.model import Model
    m = Model(None, "cpu", cp_path=None, cfg=BackboneConfig(layers=1)).eval()
    copy = m._replicate_for_data_parallel()
    assert copy._is_replica and copy.__dict__["_dp_home"] is m and "_dp_home" not in copy._modules
    with pytest.raises(slsmi.SlsmiError, match="no CPU path"):
        copy(torch.zeros(1, 64600))
    del copy.__dict__["_dp_home"]
    with pytest.raises(slsmi.SlsmiError, match="cannot find its engine"):
        copy(torch.zeros(1, 64600))


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


GLOO_WORKER_SHARDED = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import slsmi
from slsmi import training
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
pre = "ssl_model.model."
names = [pre + "encoder.layers.0.fc1.weight", pre + "encoder.layers.0.fc1.bias", pre + "encoder.layers.1.fc2.weight",
         pre + "feature_extractor.conv_layers.1.0.weight", "sae.encoder.weight", "classifier.4.bias"]
shapes = [(37, 29), (37,), (1500, 3), (5, 4, 3), (40, 17), (2,)]

def make():
    g = torch.Generator().manual_seed(11)                            # the same values on both ranks
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    ps[4] = torch.nn.Parameter(torch.randn(17, 40, generator=g).t())   # a transposed view, as the in-tree sae.encoder.weight
    return ps

kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
for dtype, tol in ((torch.float32, 0.0), (torch.bfloat16, 2e-2)):
    a, b = make(), make()
    ref_opt = torch.optim.Adam(a, foreach=False, fused=False, **kw)
    sync = training.GradSync(a, comm_dtype=dtype)
    mine = training.ShardedAdam(b, kernel="torch", comm_dtype=dtype, names=names, **kw)
    assert len(mine.buckets) == 2 and not b[4].is_contiguous()
    assert sum(n for bk in mine.buckets for _, _, n in bk["segs"]) <= sum(bk["shard"] for bk in mine.buckets)
    g = torch.Generator().manual_seed(100 + rank)                    # every rank its own gradients
    for step in range(4):
        skip = 2 if step == 1 else -1                                # one step in which a parameter has no gradient
        for i, (x, y) in enumerate(zip(a, b)):
            gr = torch.randn(x.shape, generator=g)
            x.grad = None if i == skip else torch.empty_like(x).copy_(gr)
            y.grad = None if i == skip else torch.empty_like(y).copy_(gr)
        sync.sync()
        ref_opt.step()
        mine.step()
        for i, (x, y) in enumerate(zip(a, b)):
            err = (x.detach() - y.detach()).abs().max().item()
            assert err <= tol * 3e-3 * (step + 1) + (0.0 if tol else 0.0), (dtype, step, i, err)
    if dtype == torch.float32:                                       # exactly GradSync + torch.optim.Adam, bit for bit
        for x, y in zip(a, b):
            assert torch.equal(x.detach(), y.detach())
    assert mine.state_step[id(b[2])] == 3 and mine.state_step[id(b[0])] == 4   # the skipped step did not count
    if dtype == torch.float32:
        # the optimizer state as main.py:753-764 stores it: torch.optim.Adam's own format, assembled from the shards;
        # torch's Adam continues from it exactly as it continues from its own, and so does a new ShardedAdam
        cons = mine.consolidated_state_dict()
        own = ref_opt.state_dict()
        assert set(cons["state"]) == set(own["state"])
        for i in own["state"]:
            assert float(cons["state"][i]["step"]) == float(own["state"][i]["step"])
            assert torch.equal(cons["state"][i]["exp_avg"], own["state"][i]["exp_avg"])
            assert torch.equal(cons["state"][i]["exp_avg_sq"], own["state"][i]["exp_avg_sq"])
        c = make()
        with torch.no_grad():
            for z, y in zip(c, b):
                z.copy_(y)
        third = torch.optim.Adam(c, foreach=False, fused=False, **kw)
        third.load_state_dict(cons)
        d = make()
        with torch.no_grad():
            for z, y in zip(d, b):
                z.copy_(y)
        fourth = training.ShardedAdam(d, kernel="torch", names=names, **kw)
        fourth.load_consolidated_state_dict(cons)
        for x, y, z, u in zip(a, b, c, d):
            gr = torch.randn(x.shape, generator=g)
            for t in (x, y, z, u):
                t.grad = torch.empty_like(t).copy_(gr)
        sync.sync()
        training.GradSync(c).sync()
        ref_opt.step(); mine.step(); third.step(); fourth.step()
        for x, y, z, u in zip(a, b, c, d):
            assert torch.equal(x.detach(), y.detach()) and torch.equal(x.detach(), z.detach()) and torch.equal(x.detach(), u.detach())
    # every rank ends with the same parameters (the all-gather delivered the other rank's shards)
    flat = torch.cat([y.detach().flatten() for y in b])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert torch.equal(other[0], other[1])
dist.barrier()
dist.destroy_process_group()
'''


def test_gloo_world2_sharded_optimizer_equals_allreduce_plus_adam(tmp_path):
    """ShardedAdam (reduce-scatter of the gradients, Adam on the rank's shard of two buckets, all-gather of the
    parameters) against GradSync's all-reduce followed by torch.optim.Adam on every rank: four steps on two ranks
    with weight decay, torch's default betas, a transposed-view parameter and one step in which a parameter has no
    gradient -- bit-equal parameters with fp32 on the wire, within bfloat16 rounding of the gradients otherwise."""
    script = tmp_path / "worker_sharded.py"
    script.write_text(GLOO_WORKER_SHARDED)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29537", str(script), ROOT]
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


# ---------------------------------------------------------------- checkpoint files (no code is run from them)
def _tiny_cfg():
    return BackboneConfig(layers=1)


def test_default_constructor_keys_match_the_reference_fixture():
    """`from model import Model; Model(None, dev)` must hold the keys the reference's default constructor
    gives (tests/golden/state_dict_keys.json, written by the reference's own model.py): its checkpoints load
    strictly."""
    import json
    import model as dropin
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))
    cfg = BackboneConfig(layers=2)
    for form, kw in (("default", {}), ("use_sparse_features_false", {"use_sparse_features": False}),
                     ("use_sae_false", {"use_sae": False})):
        m = dropin.Model(None, "cpu", cp_path=None, cfg=cfg, **kw)
        mine = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert mine == ref[form], (form, set(mine) ^ set(ref[form]))


def test_fairseq_layout_checkpoint_is_read_and_its_cfg_checked(tmp_path):
    """model.py:113-115 hands xlsr2_300m.pt to fairseq's loader; here the same file layout ({'args': Namespace |
    None, 'cfg': {'model': {...}}, 'model': state dict}) is read by the weights-only loader with
    argparse.Namespace allow-listed, and the embedded configuration must describe XLS-R-300M's structure."""
    import argparse
    from slsmi.model import SSLModel
    from slsmi.weights import backbone_shapes, synth_tensor, BACKBONE_PREFIX
    cfg = _tiny_cfg()
    sd = {k: synth_tensor(BACKBONE_PREFIX + k, shp, 5) for k, shp in backbone_shapes(cfg).items()}
    mcfg = {"_name": "wav2vec2", "extractor_mode": "layer_norm", "layer_norm_first": True, "conv_bias": True,
            "encoder_layers": 1, "encoder_embed_dim": 1024, "encoder_ffn_embed_dim": 4096,
            "encoder_attention_heads": 16, "conv_pos": 128, "conv_pos_groups": 16, "activation_fn": "gelu",
            "conv_feature_layers": "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]",
            "dropout": 0.0, "attention_dropout": 0.0, "encoder_layerdrop": 0.05, "dropout_input": 0.1}
    new_style = tmp_path / "new.pt"
    torch.save({"args": None, "cfg": {"model": mcfg, "task": {"_name": "audio_pretraining"}}, "model": sd,
                "optimizer_history": [{"num_updates": 10}], "extra_state": {"epoch": 3}}, new_style)
    m = SSLModel("cpu", cp_path=str(new_style), cfg=cfg)
    assert torch.equal(m.model.state_dict()["post_extract_proj.weight"], sd["post_extract_proj.weight"])
    assert m.train_stochastic["encoder_layerdrop"] == 0.05 and m.train_stochastic["dropout_input"] == 0.1
    assert m.train_stochastic["activation_dropout"] == 0.0
    # the older layout: everything in an argparse.Namespace under 'args'
    old_style = tmp_path / "old.pt"
    torch.save({"args": argparse.Namespace(**mcfg), "model": sd}, old_style)
    SSLModel("cpu", cp_path=str(old_style), cfg=cfg)
    # a file describing another architecture is refused, every differing field named
    for bad_field, val in (("layer_norm_first", False), ("extractor_mode", "default"), ("encoder_embed_dim", 768),
                           ("conv_feature_layers", "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 3")):
        p = tmp_path / f"bad_{bad_field}.pt"
        torch.save({"args": None, "cfg": {"model": dict(mcfg, **{bad_field: val})}, "model": sd}, p)
        with pytest.raises(slsmi.SlsmiError, match=bad_field):
            SSLModel("cpu", cp_path=str(p), cfg=cfg)
    # a plain state dict has no configuration to check
    plain = tmp_path / "plain.pt"
    torch.save(sd, plain)
    assert SSLModel("cpu", cp_path=str(plain), cfg=cfg).train_stochastic is None


def test_checkpoint_with_code_objects_is_refused(tmp_path):
    """Anything beyond tensors, containers, Namespace and numpy scalars stays refused."""
    import fractions
    from slsmi import checkpoint
    p = tmp_path / "evil.pt"
    torch.save({"model": {}, "hook": fractions.Fraction(1, 3)}, p)
    with pytest.raises(slsmi.SlsmiError, match="without executing code"):
        checkpoint.safe_load(str(p))


def test_trained_checkpoint_as_main_py_writes_it(tmp_path, capsys):
    """main.py:753-764: epoch, model_state_dict (DataParallel's 'module.' prefix), optimizer_state_dict, EERs as
    numpy.float64 (compute_eer returns eer * 100 as a numpy scalar), args = vars(args).  eval_sharded's loader
    reads it, strictly when the keys match; a file that lacks weights the model needs is an error."""
    import numpy as np
    import eval_sharded
    from slsmi.model import Model
    cfg = _tiny_cfg()
    sd = synth_state_dict(cfg, "sae", 201, 11)
    opt_state = {"state": {0: {"step": torch.tensor(5.0), "exp_avg": torch.zeros(3)}},
                 "param_groups": [{"lr": 1e-6, "betas": (0.9, 0.999), "weight_decay": 1e-4, "params": [0]}]}
    ckpt = {"epoch": 7, "model_state_dict": {"module." + k: v for k, v in sd.items()},
            "optimizer_state_dict": opt_state, "train_loss": 0.12, "train_eer": np.float64(1.25),
            "val_loss": 0.2, "val_acc": 98.5, "val_eer": np.float64(2.5), "best_val_eer": np.float64(2.25),
            "args": {"track": "DF", "lr": 1e-6, "comment": None, "algo": 3}}
    path = tmp_path / "best_checkpoint_eer.pth"
    torch.save(ckpt, path)
    m = Model(None, "cpu", cp_path=None, cfg=cfg, use_sae=True)
    assert eval_sharded.load_trained(m, str(path), backbone_given=False, allow_partial=False) == 0
    assert "strict=True" in capsys.readouterr().out
    assert torch.equal(m.classifier[1].weight, sd["classifier.1.weight"])
    # head weights missing: refused (the score file would come from random weights), unless asked for
    partial = dict(ckpt, model_state_dict={k: v for k, v in ckpt["model_state_dict"].items() if "classifier" not in k})
    torch.save(partial, path)
    m2 = Model(None, "cpu", cp_path=None, cfg=cfg, use_sae=True)
    assert eval_sharded.load_trained(m2, str(path), backbone_given=False, allow_partial=False) == 2
    assert eval_sharded.load_trained(m2, str(path), backbone_given=False, allow_partial=True) == 0
    # backbone keys missing are fine only when --cp_path supplied the backbone
    heads_only = dict(ckpt, model_state_dict={k: v for k, v in ckpt["model_state_dict"].items() if "ssl_model" not in k})
    torch.save(heads_only, path)
    assert eval_sharded.load_trained(m2, str(path), backbone_given=False, allow_partial=False) == 2
    assert eval_sharded.load_trained(m2, str(path), backbone_given=True, allow_partial=False) == 0


def test_extract_feat_warns_when_the_reference_would_have_recorded_gradients(monkeypatch):
    from slsmi import model as model_mod
    m = model_mod.SSLModel("cpu", cp_path=None, cfg=_tiny_cfg()).train()
    monkeypatch.setattr(model_mod.SSLModel, "_extract_feat", lambda self, x, rl, ln: torch.zeros(1))
    model_mod._warned.clear()
    with pytest.warns(RuntimeWarning, match="WITHOUT an autograd graph"):
        m.extract_feat(torch.zeros(1, 64600))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            m.extract_feat(torch.zeros(1, 64600))     # the caller asked for no graph: nothing to say
        m.eval().extract_feat(torch.zeros(1, 64600))


def test_autoencoder_from_pretrained(tmp_path):
    """AutoEncoderTopK.from_pretrained (reference model.py:91-104): shapes and k come from the file."""
    from slsmi.model import AutoEncoderTopK
    ae = AutoEncoderTopK(64, 256, 16)
    path = tmp_path / "ae.pt"
    torch.save(ae.state_dict(), path)
    back = AutoEncoderTopK.from_pretrained(str(path))
    assert (back.activation_dim, back.dict_size, int(back.k)) == (64, 256, 16)
    assert torch.equal(back.encoder.weight, ae.encoder.weight)
    assert int(AutoEncoderTopK.from_pretrained(str(path), k=8).k) == 16   # load_state_dict restores the saved k, as in the reference
