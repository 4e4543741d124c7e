"""Host-side mirror of the reference's `model` module for the scoring path.

Same names, constructor arguments, forward arity and state-dict keys as the
reference (model.py:42-104 AutoEncoderTopK, :106-141 SSLModel, :144-260 Model,
model_backup.py:186-202 getAttenF), so main.py's produce_evaluation_file and the
eval scripts can import it unchanged.  The nn.Modules here only HOLD parameters
(under the reference's key names, including fairseq's backbone names); all
arithmetic of forward() runs in libslsmi's HIP kernels through `Engine`.

Heads behind one Model class:
  head="sls": upstream SLS head, Model(args, device) (fc0 / first_bn / fc1 / fc3)
  head="sae": the in-tree TopK-SAE head, Model(args, device, use_sae=True, ...)
  head="sae_recon": the same with use_sparse_features=False (classify the reconstruction)
  head="ssl": the in-tree model with use_sae=False (classify the mean encoder output)
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, checkpoint
from .config import BackboneConfig, XLSR_300M
from .engine import Engine
from .weights import backbone_shapes, synth_tensor, strip_module_prefix, unwrap_checkpoint, BACKBONE_PREFIX


class _ParamTree(nn.Module):
    """Parameter container whose state-dict keys are given dotted names."""

    def __init__(self, shapes: Dict[str, tuple], key_prefix: str, seed: int):
        super().__init__()
        for name, shape in shapes.items():
            mod = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _ParamTree({}, "", seed))
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(synth_tensor(key_prefix + name, shape, seed)))

    def forward(self, *a, **k):  # pragma: no cover
        raise _lib.SlsmiError("backbone parameters are consumed by the HIP engine; call Model/SSLModel instead")


class AutoEncoderTopK(nn.Module):
    """TopK sparse auto-encoder parameters (reference model.py:42-104)."""

    def __init__(self, activation_dim: int, dict_size: int, k: int):
        super().__init__()
        assert isinstance(k, int) and k > 0, f"k={k} must be a positive integer"
        self.activation_dim, self.dict_size = activation_dim, dict_size
        self.register_buffer("k", torch.tensor(k, dtype=torch.int))
        self.decoder = nn.Linear(dict_size, activation_dim, bias=False)
        self.decoder.weight.data = self.decoder.weight.data / torch.norm(self.decoder.weight.data, dim=0, keepdim=True)
        self.encoder = nn.Linear(activation_dim, dict_size)
        self.encoder.weight.data = self.decoder.weight.T.clone()
        self.encoder.bias.data.zero_()
        self.b_dec = nn.Parameter(torch.zeros(activation_dim))

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """relu(encoder(x - b_dec)) with all but the k largest entries of every
        row zeroed (model.py:68-79), on the HIP GEMM + radix-select kernels."""
        from . import ops
        return ops.sae_encode(x, self.encoder.weight, self.encoder.bias, self.b_dec, int(self.k))

    def decode(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.linear_bf16(x, self.decoder.weight, None) + self.b_dec

    def forward(self, x: torch.Tensor):
        enc = self.encode(x)
        return self.decode(enc), enc

    @staticmethod
    def from_pretrained(path, k: Optional[int] = None, device=None):
        """An auto-encoder from a saved state dict (reference model.py:91-104); the file is read without
        executing anything from it."""
        state_dict = checkpoint.safe_load(path)
        dict_size, activation_dim = state_dict["encoder.weight"].shape
        if k is None:
            k = int(state_dict["k"].item())
        autoencoder = AutoEncoderTopK(activation_dim, dict_size, k)
        autoencoder.load_state_dict(state_dict)
        if device is not None:
            autoencoder.to(device)
        return autoencoder


def _load_backbone_checkpoint(cp_path: str, cfg: BackboneConfig):
    """The backbone's state dict out of a fairseq-layout checkpoint ({'args' | 'cfg', 'model'}, what the
    reference hands to fairseq's loader, model.py:113-115) or out of a plain / main.py-style one, without
    executing anything from the file.  The configuration a fairseq file embeds is compared with `cfg`: the
    reference builds whatever architecture the file describes, this library implements one.  Returns
    (state dict, the file's train-mode dropout / layerdrop probabilities or None)."""
    obj = checkpoint.safe_load(cp_path)
    mcfg = checkpoint.embedded_model_cfg(obj)
    stochastic = checkpoint.validate_backbone_cfg(mcfg, cfg) if mcfg is not None else None
    sd = unwrap_checkpoint(obj)
    sd = strip_module_prefix(sd)
    if any(k.startswith(BACKBONE_PREFIX) for k in sd):
        sd = {k[len(BACKBONE_PREFIX):]: v for k, v in sd.items() if k.startswith(BACKBONE_PREFIX)}
    return sd, stochastic


class SSLModel(nn.Module):
    """XLS-R feature extractor (reference model.py:106-141).

    `cp_path` names a checkpoint holding the backbone weights; when the file
    does not exist the reference fails (no fairseq checkpoint to build from),
    and so does this class -- unless cp_path is None, which builds the
    XLS-R-300M geometry with seeded random weights (tests, benchmarks)."""

    def __init__(self, device, cp_path: Optional[str] = "xlsr2_300m.pt", cfg: BackboneConfig = XLSR_300M,
                 seed: int = 1234, precision: Optional[str] = None):
        super().__init__()
        self.cfg = cfg
        self.precision = precision or os.environ.get("SLSMI_PRECISION", "f16")
        self.model = _ParamTree(backbone_shapes(cfg), BACKBONE_PREFIX, seed)
        self.device = device
        self.out_dim = cfg.embed_dim
        self._owner = None  # weakref to the Model that owns the engine, if any
        self._engine: Optional[Engine] = None
        self._stamp = None
        # dropout / layerdrop probabilities of the backbone in train mode, as the checkpoint's embedded
        # configuration gives them (None: the file had none, or seeded weights)
        self.train_stochastic: Optional[Dict[str, float]] = None
        if cp_path is not None:
            if not os.path.exists(cp_path):
                fallback = "xlsr_53_56k.pt"  # reference model.py:116-122
                if os.path.exists(fallback):
                    cp_path = fallback
                else:
                    raise RuntimeError(f"Could not load any model: {cp_path} not found "
                                       "(pass cp_path=None for seeded random weights)")
            sd, self.train_stochastic = _load_backbone_checkpoint(cp_path, cfg)
            missing, unexpected = self.model.load_state_dict(sd, strict=False)
            if missing:
                raise RuntimeError(f"{cp_path} lacks backbone weights: {missing[:5]} ...")

    def _backbone_engine(self, device: torch.device) -> Engine:
        owner = self._owner() if self._owner is not None else None
        if owner is not None:
            return owner._get_engine(device)
        stamp = _stamp(self)
        if self._engine is None or self._engine.device != device or self._stamp != stamp:
            if self._engine is None or self._engine.device != device:
                self._engine = Engine(device, self.cfg, "none", precision=self.precision)
            self._engine.load_tensors((BACKBONE_PREFIX + k, v) for k, v in self.model.state_dict().items())
            self._stamp = stamp
        return self._engine

    def extract_feat(self, input_data: torch.Tensor, return_layers: bool = False,
                     lengths: Optional[torch.Tensor] = None):
        """[B, L] or [B, L, 1] -> emb [B, T, 1024] (model.py:128-141).  With
        return_layers=True also returns layer_results as the 2-value variant
        does (model_backup.py:167-184): a list of (x [T, B, 1024], None).
        lengths: sample counts of a padded batch (see Model.forward); rows past an utterance's own
        frames are unspecified.  The result carries no autograd graph (the reference's does,
        model.py:209-210): gradients exist only through Model.forward in train mode; a call that would have
        recorded one in the reference warns once."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            _warn_once("SSLModel.extract_feat returns features WITHOUT an autograd graph: the reference's is "
                       "differentiable (model.py:209-210), here gradients exist only through Model.forward in train "
                       "mode (the HIP training step).  Call it under torch.no_grad() / in eval mode to silence this.")
        with torch.no_grad():
            return self._extract_feat(input_data, return_layers, lengths)

    def _extract_feat(self, input_data, return_layers, lengths):
        _move_to_input_device(self, input_data)
        wav = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        eng = self._backbone_engine(wav.device)
        out = eng.forward(wav, want_x=True, want_layers=return_layers, lengths=lengths)
        if not return_layers:
            return out["x"]
        layers = [(out["layers"][l].transpose(0, 1), None) for l in range(self.cfg.layers)]
        return out["x"], layers


_warned = set()


def _warn_once(msg: str) -> None:
    if msg not in _warned:
        _warned.add(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def getAttenF(layerResult):
    """Reference model_backup.py:186-202: per-layer time means and the stacked
    full features, from layer_results as extract_feat(return_layers=True) gives."""
    pool, full = [], []
    for layer in layerResult:
        xb = layer[0].transpose(0, 1)  # [B, T, C]
        pool.append(F.adaptive_avg_pool1d(xb.transpose(1, 2), 1).transpose(1, 2))
        full.append(xb.unsqueeze(1))
    return torch.cat(pool, dim=1), torch.cat(full, dim=1)


class _TrainStep(torch.autograd.Function):
    """Model.forward in train mode as one autograd node: forward and backward both run in libslsmi
    (slsmi_train_forward / slsmi_train_backward); autograd only carries the gradients of the two
    outputs in and the parameter gradients out, so optimizers, DistributedDataParallel hooks and
    the caller's loss (main.py:224-238) work unchanged."""

    @staticmethod
    def forward(ctx, model, wav, mask, want_loss, names, bb_masks, *params):
        eng = model._get_engine(wav.device, training=True)
        eng.set_train_masks(bb_masks, wav.shape[0], eng.num_frames(wav.shape[1]))
        logp, loss = eng.train_forward(wav, mask, want_loss)
        ctx.model, ctx.names, ctx.wav, ctx.used_mask, ctx.want_loss = model, names, wav, mask is not None, want_loss
        ctx.params = params
        ctx.dropped = _dropped_layer_params(names, bb_masks)
        if loss is None:
            loss = torch.zeros(1, device=wav.device)
        return logp, loss.reshape(())

    @staticmethod
    def backward(ctx, dlogp, dloss):
        model = ctx.model
        eng = model._engine
        have = set(eng.grad_names(ctx.names))
        want = {n: p for n, p in zip(ctx.names, ctx.params) if n in have and p.requires_grad and n not in ctx.dropped}
        fused = model._fused_adam_begin(eng, want)
        grads = eng.grad_buffers({n: p for n, p in want.items() if n not in fused})
        eng.train_backward(ctx.wav, dlogp, dloss if ctx.want_loss else None, grads, ctx.used_mask)
        # An optimizer step follows.  Fused optimizers (torch.optim.Adam(fused=True)) update the parameters
        # without raising their version counters, so the stamp would not see it: reload on the next forward.
        model._engine_dirty = True
        return (None, None, None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


class _BackboneTrainStep(torch.autograd.Function):
    """The backbone in train mode as one autograd node: wav -> the encoder layers' outputs [layers, B, T, 1024]
    (what extract_feat returns as layer_results, model_backup.py:167-184), forward and backward in libslsmi
    (slsmi_train_backbone_forward / _backward).  The head that consumes them is the caller's."""

    @staticmethod
    def forward(ctx, model, wav, names, bb_masks, *params):
        eng = model._get_engine(wav.device, training=True)
        eng.set_train_masks(bb_masks, wav.shape[0], eng.num_frames(wav.shape[1]))
        layers = eng.train_backbone_forward(wav)
        ctx.model, ctx.names, ctx.wav, ctx.params = model, names, wav, params
        ctx.dropped = _dropped_layer_params(names, bb_masks)
        return layers

    @staticmethod
    def backward(ctx, dlayers):
        model = ctx.model
        eng = model._engine
        have = set(eng.grad_names(ctx.names))   # pre-training leftovers (mask_emb, quantizer, ...) are never read
        want = {n: p for n, p in zip(ctx.names, ctx.params) if n in have and p.requires_grad and n not in ctx.dropped}
        fused = model._fused_adam_begin(eng, want)
        grads = eng.grad_buffers({n: p for n, p in want.items() if n not in fused})
        eng.train_backbone_backward(ctx.wav, dlayers, grads)
        model._engine_dirty = True   # see _TrainStep.backward
        return (None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


class _SLSTrainStep(torch.autograd.Function):
    """The upstream SLS model, Model(args, device), in train mode as one autograd node: backbone AND head run in
    libslsmi, forward and backward (slsmi_train_sls_forward / _backward); autograd carries dLoss/dlogp in and the
    parameter gradients out.  The BatchNorm's running statistics are updated from the batch statistics the forward
    returns, as nn.BatchNorm2d does in train mode."""

    @staticmethod
    def forward(ctx, model, wav, names, bb_masks, *params):
        eng = model._get_engine(wav.device, training=True)
        eng.set_train_masks(bb_masks, wav.shape[0], eng.num_frames(wav.shape[1]))
        logp, stats = eng.train_sls_forward(wav, model.fc1.weight)
        bn = model.first_bn
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1.0 - mom).add_(stats[0:1], alpha=mom)
                bn.running_var.mul_(1.0 - mom).add_(stats[1:2], alpha=mom)
        ctx.model, ctx.names, ctx.wav, ctx.params = model, names, wav, params
        ctx.dropped = _dropped_layer_params(names, bb_masks)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        model = ctx.model
        eng = model._engine
        have = set(eng.grad_names(ctx.names))
        want = {n: p for n, p in zip(ctx.names, ctx.params) if n in have and p.requires_grad and n not in ctx.dropped}
        fused = model._fused_adam_begin(eng, want)
        grads = eng.grad_buffers({n: p for n, p in want.items() if n not in fused})
        eng.train_sls_backward(ctx.wav, dlogp, grads, model.fc1.weight)
        model._engine_dirty = True   # see _TrainStep.backward
        return (None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


_LAYER_OF = None


def _dropped_layer_params(names, bb_masks):
    """Names among `names` that belong to encoder layers skipped by layerdrop in this step.  The reference never
    calls a skipped layer (wav2vec2.py:941-958), so its parameters are not in the graph: they keep grad = None and the
    optimizer leaves them alone -- no weight decay, no momentum update, no step count."""
    global _LAYER_OF
    if not bb_masks or bb_masks.get("layer_keep") is None:
        return set()
    if _LAYER_OF is None:
        import re
        _LAYER_OF = re.compile(r"^ssl_model\.model\.encoder\.layers\.(\d+)\.")
    keep = [bool(k) for k in bb_masks["layer_keep"]]
    out = set()
    for n in names:
        m = _LAYER_OF.match(n)
        if m and not keep[int(m.group(1))]:
            out.add(n)
    return out


def _stamp(module: nn.Module):
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def _replica_state_dict(module: nn.Module) -> Dict[str, torch.Tensor]:
    """state_dict() of a per-device copy made by torch.nn.parallel.replicate: there the broadcast parameters are
    plain attributes listed in _former_parameters (not in _parameters, which state_dict() walks)."""
    out: Dict[str, torch.Tensor] = {}
    for name, mod in module.named_modules():
        pre = name + "." if name else ""
        for k, v in list(getattr(mod, "_former_parameters", {}).items()) + list(mod._parameters.items()):
            if v is not None:
                out[pre + k] = v.detach()
        for k, v in mod._buffers.items():
            if v is not None and k not in mod._non_persistent_buffers_set:
                out[pre + k] = v
    return out


def _move_to_input_device(module: nn.Module, x: torch.Tensor) -> None:
    """The reference moves the backbone to the input's device on first use
    (model.py:130-132); do the same for every parameter held here."""
    if x.device.type != "cuda":
        raise _lib.SlsmiError(f"input is on {x.device}: slsmi runs on ROCm devices only, there is no CPU path")
    p = next(module.parameters())
    if p.device != x.device:
        module.to(x.device)


class Model(nn.Module):
    """Audio deepfake detector: XLS-R backbone + SLS or TopK-SAE head.

    Constructor and forward follow reference model.py:150-260; `args` is
    accepted and ignored like in the reference (it may be None)."""

    def __init__(self, args, device, cp_path: Optional[str] = "xlsr2_300m.pt", use_sae: Optional[bool] = None,
                 use_sparse_features: bool = True, sae_dict_size: int = 4096, sae_k: int = 128,
                 sae_weight: float = 0.1, head: Optional[str] = None, cfg: BackboneConfig = XLSR_300M,
                 num_samples: int = 64600, seed: int = 1234, precision: Optional[str] = None,
                 sae_window_size: int = 1, low_latency: Optional[bool] = None):
        super().__init__()
        # low_latency (or SLSMI_LOW_LATENCY=1): small batches score faster (GEMMs cut along K to fill the chip),
        # at the price that a score depends on the batch size in its last bits (slsmi_set_low_latency)
        self.low_latency = bool(int(os.environ.get("SLSMI_LOW_LATENCY", "0"))) if low_latency is None else bool(low_latency)
        # sae_window_size > 1: the window TopK of model_window_topk.py:118-203 instead of the per-frame
        # TopK of model.py:73-77 (WindowTopKModel below has the reference's default of 8)
        self.sae_window_size = int(sae_window_size)
        # 16-bit operand format of the HIP GEMMs: "f16" (default, 11-bit mantissa)
        # or "bf16"; fp32 accumulation and residual stream in both.
        self.precision = precision or os.environ.get("SLSMI_PRECISION", "f16")
        if self.precision not in _lib.FORMATS:
            raise ValueError(f"precision must be one of {sorted(_lib.FORMATS)}, got {self.precision!r}")
        # use_sae=None (argument not given) builds the upstream SLS model, Model(args, device);
        # use_sae=True / False build the in-tree model with or without the auto-encoder
        # (model.py:171-179): the classifier pools the sparse code, its reconstruction
        # (use_sparse_features=False) or the encoder output itself (use_sae=False).
        if head is None:
            if use_sae is None:
                head = "sls"
            elif not use_sae:
                head = "ssl"
            else:
                head = "sae" if use_sparse_features else "sae_recon"
        if head not in ("sls", "sae", "sae_recon", "ssl"):
            raise ValueError(f"unknown head {head!r}")
        self.device = device
        self.head = head
        self.cfg = cfg
        self.use_sae = head in ("sae", "sae_recon")
        self.use_sparse_features = use_sparse_features
        self.sae_weight = sae_weight
        self.sae_dict_size, self.sae_k = sae_dict_size, sae_k
        self.ssl_model = SSLModel(device=device, cp_path=cp_path, cfg=cfg, seed=seed, precision=self.precision)
        self.ssl_model._owner = weakref.ref(self)
        frames = cfg.num_frames(num_samples)
        if head == "sls":
            self.first_bn = nn.BatchNorm2d(num_features=1)
            self.selu = nn.SELU(inplace=True)
            self.fc0 = nn.Linear(cfg.embed_dim, 1)
            self.sig = nn.Sigmoid()
            self.fc1 = nn.Linear((frames // 3) * (cfg.embed_dim // 3), 1024)
            self.fc3 = nn.Linear(1024, 2)
            self.logsoftmax = nn.LogSoftmax(dim=1)
        else:
            if self.use_sae:
                self.sae = AutoEncoderTopK(activation_dim=cfg.embed_dim, dict_size=sae_dict_size, k=sae_k)
            input_dim = sae_dict_size if head == "sae" else cfg.embed_dim
            self.pool = nn.AdaptiveAvgPool1d(1)
            self.classifier = nn.Sequential(nn.LayerNorm(input_dim), nn.Linear(input_dim, 256), nn.ReLU(),
                                            nn.Dropout(0.3), nn.Linear(256, 2))
        self.last_sparse_features = None
        self.last_feature_indices = None
        self._engine: Optional[Engine] = None
        self._engine_stamp = None
        self._engine_dirty = False  # a backward ran: an optimizer step may have changed the parameters since

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self, device: torch.device, training: bool = False) -> Engine:
        device = torch.device(device)
        stamp = _stamp(self)
        if self._engine is None or self._engine.device != device:
            self._engine = Engine(device, self.cfg, self.head, self.sae_dict_size, self.sae_k, self.precision,
                                  self.sae_window_size if self.use_sae else 1, low_latency=self.low_latency)
            self._engine_stamp = None
        if training and not self._engine.training_enabled:
            self._engine.enable_training()
            self._engine_stamp = None
        if training and getattr(self, "_layer_event", None) is not None and \
                getattr(self._engine, "_layer_event", None) is not self._layer_event:
            self._engine.set_layer_grads_event(self._layer_event)
        if self._engine_stamp != stamp or self._engine_dirty:
            same_buffers = (self._engine_stamp is not None and self._engine.reloadable and
                            [p for p, _ in self._engine_stamp] == [p for p, _ in stamp])
            if same_buffers:  # updated in place (an optimizer step): one library call reads them again
                seen = self.__dict__.pop("_fused_seen", None)
                if seen and any(p._version != v for p, v in seen):
                    self._engine.drop_fused_copies()   # someone else wrote to a tensor the backward had stepped
                self._engine.reload_weights()
            else:
                self._engine.load_state_dict(self.state_dict())
            self._engine_stamp = stamp
            self._engine_dirty = False
        return self._engine

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        return super().load_state_dict(strip_module_prefix(unwrap_checkpoint(state_dict)), strict=strict, **kw)

    def _fused_adam_begin(self, eng: Engine, want: Dict[str, torch.Tensor]):
        """Called by the training backward with the parameters it is about to compute gradients for: an optimizer
        attached with slsmi.training.Adam(fuse_backward=model) names the ones whose step the backward takes itself
        (they get no gradient).  Returns their names."""
        ref = self.__dict__.get("_fused_opt")
        opt = ref() if ref is not None else None
        if opt is None:
            return frozenset()
        names = opt._fuse_into_backward(eng, want)
        # (the backward also leaves the library's 16-bit copies of these tensors current, and the next reload skips
        # them: remember their version counters, so that a change by other means before that reload is noticed)
        self.__dict__["_fused_seen"] = [(want[n], want[n]._version) for n in names]
        return names

    def layer_grads_event(self) -> "torch.cuda.Event":
        """An event that every training backward records once the gradients of all encoder-layer parameters are
        enqueued, ~1.5 ms of kernels before the backward ends (slsmi_train_set_layer_grads_event): what a gradient
        exchange waits for to start early (slsmi.training.ShardedAdam)."""
        if getattr(self, "_layer_event", None) is None:
            self.__dict__["_layer_event"] = torch.cuda.Event()
        return self._layer_event

    # ------------------------------------------------------------------ forward
    def forward(self, input_data: torch.Tensor, return_sae_loss: Optional[bool] = None,
                return_interpretability: bool = False, lengths: Optional[torch.Tensor] = None,
                dropout_mask: Optional[torch.Tensor] = None, backbone_masks: Optional[Dict[str, torch.Tensor]] = None):
        """[B, L] (or [B, L, 1]) fp32 on a ROCm device -> log-probs [B, 2].

        In train mode with gradients enabled (main.py:204, :224) the outputs carry the autograd node of
        the HIP training step (every form of the model; up to 256 frames, batch <= 64);
        dropout_mask ([B, 256] multipliers) overrides the mask drawn for the classifier's Dropout(0.3);
        backbone_masks overrides the draws of the backbone's own train-mode dropout / layerdrop, which apply when the
        backbone checkpoint's configuration sets them (Engine.set_train_masks names the entries).

        lengths (not in the reference, whose datasets cut or tile every clip to one length): the sample
        count of every utterance of a padded batch; utterance b is then scored exactly as if it were run
        alone at its own length (the backbone's padding_mask path, wav2vec/wav2vec2.py:567-586, :912-913,
        :1050; time mean over its own frames).  In-tree heads only; no SAE loss.

        Return arity as the reference (model.py:254-260): SAE head returns
        (output, sae_loss) unless return_sae_loss=False; SLS head returns output
        (and (output, None) when return_sae_loss=True is forced)."""
        if getattr(self, "_is_replica", False):
            return self._forward_replica(input_data, return_sae_loss, return_interpretability, lengths)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(input_data, return_sae_loss, return_interpretability, lengths, dropout_mask,
                                       backbone_masks)
        _move_to_input_device(self, input_data)
        wav = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        return self._forward_eval(self._get_engine(wav.device), wav, return_sae_loss, return_interpretability, lengths)

    def _forward_eval(self, eng: Engine, wav, return_sae_loss, return_interpretability, lengths):
        if return_sae_loss is None:
            return_sae_loss = self.head != "sls" and lengths is None
        want_code = bool(return_interpretability and self.use_sae)
        out = eng.forward(wav, want_sae_loss=bool(return_sae_loss and self.use_sae), want_sae_code=want_code,
                          lengths=lengths)
        output = out["logp"]
        loss = out["sae_loss"].reshape(()) if out["sae_loss"] is not None else None
        if return_interpretability:
            interp = None
            if want_code:  # model.py:236-240, :250-251
                self.last_sparse_features = out["sae_code"]
                self.last_feature_indices = out["sae_code"] > 0
                interp = self.get_interpretability_info(None)
            return (output, loss, interp) if return_sae_loss else (output, interp)
        if return_sae_loss:
            return output, loss
        return output

    # ------------------------------------------------------------------ nn.DataParallel over several GPUs
    def _replicate_for_data_parallel(self):
        """nn.DataParallel over several devices (main.py:518 on a multi-GPU node) calls this once per device and
        forward: the copy keeps a reference to the module it was made from, which owns one engine per device."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_home"] = self   # (not a submodule: kept out of _modules)
        return replica

    def _forward_replica(self, input_data, return_sae_loss, return_interpretability, lengths):
        """Scoring forward of a per-device copy made by nn.DataParallel.  The copies (and their broadcast
        parameters) are new objects on every call, so the engines live on the module they were copied from, one
        per device, and take their weights again only when that module's parameters have changed: a scoring run
        packs the 315 M weights once per GPU, not once per batch.  (torch still broadcasts the parameters to
        every device on every call; one process per GPU -- eval_sharded.py -- avoids that too.)"""
        home = self.__dict__.get("_dp_home")
        if home is None:
            raise _lib.SlsmiError("a module copy made outside nn.DataParallel's replicate() cannot find its engine")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in home.parameters()):
            raise _lib.SlsmiError("training under nn.DataParallel over several GPUs is not supported: run one process "
                                  "per GPU with slsmi.training.GradSync (gradients averaged by one RCCL all-reduce), "
                                  "or make one device visible (HIP_VISIBLE_DEVICES)")
        if input_data.device.type != "cuda":
            raise _lib.SlsmiError(f"input is on {input_data.device}: slsmi runs on ROCm devices only, there is no CPU path")
        wav = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        dev = wav.device
        if next(home.parameters()).device == dev:
            eng = home._get_engine(dev)          # the copy on the module's own device shares its parameters
        else:
            engines = home.__dict__.setdefault("_dp_engines", {})
            stamp = _stamp(home)
            entry = engines.get(dev.index)
            if entry is None:
                entry = [Engine(dev, self.cfg, self.head, self.sae_dict_size, self.sae_k, self.precision,
                                self.sae_window_size if self.use_sae else 1, low_latency=self.low_latency), None]
                engines[dev.index] = entry
            if entry[1] != stamp:
                entry[0].load_state_dict(_replica_state_dict(self))   # this copy's tensors are on `dev`
                entry[1] = stamp
            eng = entry[0]
        return self._forward_eval(eng, wav, return_sae_loss, return_interpretability, lengths)

    def _draw_backbone_masks(self, B: int, T: int, device) -> Optional[Dict[str, torch.Tensor]]:
        """The backbone's train-mode draws from the probabilities its checkpoint's configuration holds
        (SSLModel.train_stochastic): nn.Dropout multipliers from torch's generator; the layerdrop decisions from
        numpy's global generator compared as the reference does (wav2vec2.py:941-942)."""
        st = self.ssl_model.train_stochastic or {}
        L, Fd = self.cfg.layers, self.cfg.ffn_dim

        def bern(p, *shape):
            return torch.bernoulli(torch.full(shape, 1.0 - p, device=device)) / (1.0 - p)

        masks: Dict[str, torch.Tensor] = {}
        if st.get("dropout_input", 0) > 0:
            masks["input"] = bern(st["dropout_input"], B, T, 1024)
        if st.get("dropout", 0) > 0:
            masks["encoder"] = bern(st["dropout"], B, T, 1024)
            masks["attn_out"] = bern(st["dropout"], L, B, T, 1024)
            masks["ffn_out"] = bern(st["dropout"], L, B, T, 1024)
        if st.get("activation_dropout", 0) > 0:
            masks["ffn_act"] = bern(st["activation_dropout"], L, B, T, Fd)
        if st.get("attention_dropout", 0) > 0:   # MultiheadAttention's dropout on the probabilities (:1012)
            p = float(st["attention_dropout"])
            masks["attn_keep"] = torch.stack([torch.rand(B, 16, T, T, device=device) >= p for _ in range(L)])
            masks["attn_p"] = p
        if st.get("encoder_layerdrop", 0) > 0:
            import numpy as np
            masks["layer_keep"] = torch.tensor([np.random.random() > st["encoder_layerdrop"] for _ in range(L)])
        return masks or None

    def _forward_train(self, input_data, return_sae_loss, return_interpretability, lengths, dropout_mask,
                       backbone_masks=None):
        if lengths is not None or return_interpretability:
            raise NotImplementedError("lengths= / return_interpretability are scoring options")
        # the reference's extract_feat leaves the backbone in train mode (model.py:130-132): what the checkpoint's
        # configuration sets applies while fine-tuning.  dropout_features acts on a copy this path never reads.
        st = self.ssl_model.train_stochastic or {}
        ignore = os.environ.get("SLSMI_IGNORE_BACKBONE_DROPOUT") == "1"
        if return_sae_loss is None:
            return_sae_loss = self.head != "sls"
        _move_to_input_device(self, input_data)
        wav = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        # one fp32 contiguous copy held by the autograd node: the backward must see the buffer the forward read
        wav = wav.detach().to(torch.float32).contiguous()
        if backbone_masks is None and not ignore:
            backbone_masks = self._draw_backbone_masks(wav.shape[0], self.cfg.num_frames(wav.shape[1]), wav.device)
        if self.head == "sls":
            return self._forward_train_sls(wav, return_sae_loss, backbone_masks)
        if os.environ.get("SLSMI_HEAD_AUTOGRAD") == "1":   # cross-check: the head by autograd on this class's modules
            return self._forward_train_variant(wav, return_sae_loss, backbone_masks, dropout_mask)
        # every form main.py can build runs as one node in the library: the classifier on the sparse code, on its
        # reconstruction (model.py:228-233) or on the encoder output (model.py:171-179); TopK per frame or by window
        # votes (main.py:493-505)
        if dropout_mask is None:  # nn.Dropout(0.3) of the classifier (model.py:187), drawn by torch's generator
            p = self.classifier[3].p
            dropout_mask = torch.bernoulli(torch.full((wav.shape[0], 256), 1.0 - p, device=wav.device)) / (1.0 - p)
        named = [(n, p) for n, p in self.named_parameters()]
        names = tuple(n for n, _ in named)
        want_loss = bool(return_sae_loss and self.use_sae)
        logp, loss = _TrainStep.apply(self, wav, dropout_mask, want_loss, names, backbone_masks, *[p for _, p in named])
        if not return_sae_loss:
            return logp
        return (logp, loss) if self.use_sae else (logp, None)

    def _forward_train_sls(self, wav, return_sae_loss, backbone_masks=None):
        """Train mode of the upstream SLS model: backbone and head -- getAttenF (model_backup.py:186-202),
        sigmoid(fc0) layer weights, weighted layer sum, BatchNorm2d(1) on batch statistics (its running statistics
        are updated as nn.BatchNorm2d does), SELU, max_pool2d(3, 3), fc1, fc3, log-softmax -- run in libslsmi,
        forward and backward (_SLSTrainStep).  SLSMI_HEAD_AUTOGRAD=1 differentiates the head by autograd on the
        modules this class holds instead (the backbone stays in the library): a cross-check of the head's kernels."""
        pre = "ssl_model.model."
        skip = pre + "encoder.layer_norm."   # applied to the encoder's last output only, which this head does not read
        if os.environ.get("SLSMI_HEAD_AUTOGRAD") != "1":
            named = [(n, p) for n, p in self.named_parameters() if not n.startswith(skip)]
            names = tuple(n for n, _ in named)
            output = _SLSTrainStep.apply(self, wav, names, backbone_masks, *[p for _, p in named])
            return (output, None) if return_sae_loss else output
        named = [(n, p) for n, p in self.named_parameters() if n.startswith(pre) and not n.startswith(skip)]
        names = tuple(n for n, _ in named)
        layers = _BackboneTrainStep.apply(self, wav, names, backbone_masks, *[p for _, p in named])   # [layers, B, T, C]
        if backbone_masks and backbone_masks.get("layer_keep") is not None:
            # a layer skipped by layerdrop appends no layer result (wav2vec2.py:941-958)
            kept = [l for l, k in enumerate(backbone_masks["layer_keep"]) if bool(k)]
            layers = layers[torch.tensor(kept, device=layers.device, dtype=torch.long)]
        full = layers.permute(1, 0, 2, 3)                                   # fullfeature [B, layers, T, C]
        y0 = self.sig(self.fc0(full.mean(dim=2)))                           # [B, layers, 1]
        x = (full * y0.unsqueeze(-1)).sum(dim=1).unsqueeze(1)               # [B, 1, T, C]
        x = self.selu(self.first_bn(x))
        x = torch.flatten(F.max_pool2d(x, (3, 3)), 1)
        x = self.selu(self.fc1(x))
        x = self.selu(self.fc3(x))
        output = self.logsoftmax(x)
        return (output, None) if return_sae_loss else output

    def _forward_train_variant(self, wav, return_sae_loss, backbone_masks=None, dropout_mask=None):
        """SLSMI_HEAD_AUTOGRAD=1 only: the in-tree model's head differentiated by autograd on the modules this class
        holds, behind the library's backbone step (_BackboneTrainStep) -- a cross-check of the library's own head
        kernels in every form (per-frame or window TopK, model_window_topk.py:271-393; the reconstruction as classifier
        input, model.py:228-233; no auto-encoder, model.py:171-179).  A TopK selection -- per frame or by window
        votes -- is a constant 0 / 1 mask on the activations for the backward, exactly as in the reference (the code is
        `x * mask`, model_window_topk.py:196; scatter of the top-k values, model.py:73-77); the window selection
        itself comes from the library's window TopK kernels (slsmi_window_topk)."""
        from . import ops
        pre = "ssl_model.model."
        ln_pre = pre + "encoder.layer_norm."   # applied here, by autograd
        named = [(n, p) for n, p in self.named_parameters() if n.startswith(pre) and not n.startswith(ln_pre)]
        names = tuple(n for n, _ in named)
        layers = _BackboneTrainStep.apply(self, wav, names, backbone_masks, *[p for _, p in named])   # [layers, B, T, C]
        ln = self.ssl_model.model.encoder.layer_norm
        x = F.layer_norm(layers[-1], (self.cfg.embed_dim,), ln.weight, ln.bias, self.cfg.ln_eps)   # wav2vec2.py:905-906
        B, T, C = x.shape
        sae_loss = None
        if self.use_sae:
            x_flat = x.reshape(B * T, C)
            acts = F.relu(F.linear(x_flat - self.sae.b_dec, self.sae.encoder.weight, self.sae.encoder.bias))
            k = int(self.sae.k)
            if self.sae_window_size > 1:
                with torch.no_grad():
                    kept = ops.window_topk(acts.detach().reshape(B, T, -1), k, self.sae_window_size) != 0
                encoded = acts * kept.reshape(B * T, -1)
            else:
                vals, idx = acts.topk(k, sorted=False, dim=-1)
                encoded = torch.zeros_like(acts).scatter(-1, idx, vals)
            x_recon = F.linear(encoded, self.sae.decoder.weight) + self.sae.b_dec
            if return_sae_loss:
                sae_loss = F.mse_loss(x_recon, x_flat)
            feats = encoded.reshape(B, T, -1) if self.head == "sae" else x_recon.reshape(B, T, C)
        else:
            feats = x
        pooled = feats.mean(dim=1)   # AdaptiveAvgPool1d(1) over the frames (model.py:245)
        if dropout_mask is None:
            logits = self.classifier(pooled)
        else:   # the caller's draw for nn.Dropout(0.3) (model.py:187)
            h = F.relu(self.classifier[1](self.classifier[0](pooled))) * dropout_mask
            logits = self.classifier[4](h)
        output = F.log_softmax(logits, dim=-1)
        return (output, sae_loss) if return_sae_loss else output

    def get_interpretability_info(self, pooled_features=None):
        """Activation statistics of the last sparse code (model.py:262-293); the code comes
        from the HIP forward, these summaries are device-side tensor reductions of it."""
        if self.last_sparse_features is None:
            return None
        code = self.last_sparse_features
        B, T, D = code.shape
        avg_activation = code.mean(dim=1)
        topk_values, topk_indices = avg_activation.topk(k=min(20, D), dim=-1)
        active = (code > 0).float()
        return {"avg_activation": avg_activation, "top20_features": topk_indices, "top20_values": topk_values,
                "sparsity": active.mean(dim=[1, 2]), "activation_freq": active.mean(dim=1),
                "sparse_features": code}

    def compute_total_loss(self, classification_loss: torch.Tensor, sae_loss: torch.Tensor = None):
        """classification + sae_weight * reconstruction (model.py:295-299)."""
        if sae_loss is None or not self.use_sae:
            return classification_loss
        return classification_loss + (self.sae_weight * sae_loss)


class WindowTopKModel(Model):
    """`Model` of the reference's model_window_topk.py (:271-393), which main.py builds under
    --use_window_topk (main.py:493-505): the in-tree model with window TopK, sae_window_size = 8 by
    default.  Same state-dict keys as Model."""

    def __init__(self, args, device, cp_path: Optional[str] = "xlsr2_300m.pt", use_sae: bool = True,
                 use_sparse_features: bool = True, sae_dict_size: int = 4096, sae_k: int = 128,
                 sae_window_size: int = 8, sae_weight: float = 0.1, **kw):
        super().__init__(args, device, cp_path=cp_path, use_sae=use_sae, use_sparse_features=use_sparse_features,
                         sae_dict_size=sae_dict_size, sae_k=sae_k, sae_weight=sae_weight,
                         sae_window_size=sae_window_size, **kw)
