"""Python handle on one libslsmi context (one per model per device).

PyTorch is used for device memory and streams only: tensors are handed to the
C ABI as raw device pointers together with torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from .config import BackboneConfig, XLSR_300M

# head name -> (library head, what the SAE-head classifier pools)
_HEADS = {"sls": (_lib.HEAD_SLS, _lib.SAE_SPARSE), "sae": (_lib.HEAD_SAE, _lib.SAE_SPARSE),
          "sae_recon": (_lib.HEAD_SAE, _lib.SAE_RECON), "ssl": (_lib.HEAD_SAE, _lib.SAE_OFF),
          "none": (_lib.HEAD_NONE, _lib.SAE_SPARSE)}


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def make_config(cfg: BackboneConfig, head: str, sae_dict: int = 4096, sae_k: int = 128,
                precision: str = "f16", sae_window: int = 1) -> _lib.Config:
    c = _lib.Config()
    _lib.load().slsmi_default_config(C.byref(c), _HEADS[head][0])
    c.sae_features = _HEADS[head][1]
    c.n_conv = len(cfg.conv_layers)
    for i, (d, k, s) in enumerate(cfg.conv_layers):
        c.conv_dim[i], c.conv_kernel[i], c.conv_stride[i] = d, k, s
    c.embed_dim, c.ffn_dim, c.heads, c.layers = cfg.embed_dim, cfg.ffn_dim, cfg.heads, cfg.layers
    c.conv_pos, c.conv_pos_groups, c.ln_eps = cfg.conv_pos, cfg.conv_pos_groups, cfg.ln_eps
    c.sae_dict, c.sae_k = sae_dict, sae_k
    c.sae_window = sae_window
    c.operand_format = _lib.FORMATS[precision]
    return c


_LAYER_WEIGHT = re.compile(r"(.*encoder\.layers\.)(\d+)\.(.+\.weight)$")

class Engine:
    """Owns a slsmi_ctx; loads weights from device tensors and runs forwards."""

    def __init__(self, device: torch.device, cfg: BackboneConfig = XLSR_300M, head: str = "sls",
                 sae_dict: int = 4096, sae_k: int = 128, precision: str = "f16", sae_window: int = 1,
                 low_latency: bool = False, ln_fold: Optional[bool] = None):
        device = torch.device(device)
        if precision not in _lib.FORMATS:
            raise ValueError(f"precision must be one of {sorted(_lib.FORMATS)}, got {precision!r}")
        self.precision = precision
        if device.type != "cuda":
            raise _lib.SlsmiError(f"slsmi runs on ROCm devices only (got device {device}); there is no CPU path")
        if not torch.cuda.is_available():
            raise _lib.SlsmiError("no ROCm device visible to PyTorch; slsmi has no CPU path")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.cfg, self.head = cfg, head
        self._ccfg = make_config(cfg, head, sae_dict, sae_k, precision, sae_window)
        self._ctx = C.c_void_p()
        _lib.check(self.lib.slsmi_create(self.device.index, C.byref(self._ccfg), C.byref(self._ctx)), None,
                   "slsmi_create")
        if low_latency:  # small batches: GEMMs cut along K to fill the chip (scores then depend on the batch size
            # in their last bits; see slsmi_set_low_latency)
            _lib.check(self.lib.slsmi_set_low_latency(self._ctx, 1), self._ctx, "slsmi_set_low_latency")
        # ln_fold (or SLSMI_LN_FOLD=0/1; on by default): the scoring forward folds the encoder's LayerNorms into the
        # GEMMs around them (slsmi_set_ln_fold)
        if ln_fold is None:
            ln_fold = int(os.environ.get("SLSMI_LN_FOLD", "1"))
        self.ln_fold = int(ln_fold)   # 0 off, 1 on, 2 (tests) the 208 x 256 form for every batch it can run
        _lib.check(self.lib.slsmi_set_ln_fold(self._ctx, self.ln_fold), self._ctx, "slsmi_set_ln_fold")
        # SLSMI_POSCONV_SLAB=0: the positional conv as a batched GEMM everywhere (same bits; for A/B timing)
        if os.environ.get("SLSMI_POSCONV_SLAB", "1") == "0":
            _lib.check(self.lib.slsmi_set_posconv_slab(self._ctx, 0), self._ctx, "slsmi_set_posconv_slab")
        # SLSMI_CONV_PRECISION="<split mask>,<fp32 mask>" (e.g. "0,0": no extra precision in the feature extractor; for
        # A/B measurements -- the library's defaults are what the parity tests hold): slsmi_set_conv_precision
        if os.environ.get("SLSMI_CONV_PRECISION"):
            sm, fm = (int(v, 0) for v in os.environ["SLSMI_CONV_PRECISION"].split(","))
            self.set_conv_precision(sm, fm)
        self._ws: Optional[torch.Tensor] = None
        self._ws_key: Optional[Tuple[int, int]] = None
        self.training_enabled = False
        self.reloadable = False  # the buffers the weights were loaded from are still there (see reload_weights)
        self._staged: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._unbacked: set = set()  # weights whose source buffer is gone
        self._tws: Optional[torch.Tensor] = None
        # training step: the input copy, the mask and the two outputs live in buffers of the engine's own
        self._tbuf: Dict[str, torch.Tensor] = {}
        self._registered: Dict[str, int] = {}

    def set_conv_precision(self, split_mask: int, fp32_mask: int) -> None:
        """Bit i of a mask = conv layer i of the feature extractor: multiply by both 16-bit halves of its weights /
        hand its output to the LayerNorm as fp32 (slsmi_set_conv_precision)."""
        _lib.check(self.lib.slsmi_set_conv_precision(self._ctx, split_mask, fp32_mask), self._ctx,
                   "slsmi_set_conv_precision")
        self._ws_key = None   # the workspace size follows the fp32 mask

    RANGE_KINDS = ("LayerNorm 1 operand", "q/k/v", "attention output", "LayerNorm 2 operand", "feed-forward hidden")

    def range_report(self, enable: bool = True) -> None:
        """Switch the range report of the scoring forward's 16-bit stores on or off (slsmi_set_range_report)."""
        if not enable:
            _lib.check(self.lib.slsmi_set_range_report(self._ctx, None, 0), self._ctx, "slsmi_set_range_report")
            self._range = None
            return
        n = 10 + 5 * self.cfg.layers
        self._range = torch.zeros(n, 2, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.slsmi_set_range_report(self._ctx, C.c_void_p(self._range.data_ptr()), n), self._ctx,
                   "slsmi_set_range_report")

    def last_ranges(self):
        """What the last forward's 16-bit stores held, as a list of (name, largest |value|, saturated count)."""
        if getattr(self, "_range", None) is None:
            raise _lib.SlsmiError("range_report() has not been switched on")
        r = self._range.cpu()
        mx = r[:, 0].contiguous().view(torch.float32)
        rows = [(f"conv block {i}", float(mx[i]), int(r[i, 1])) for i in range(len(self.cfg.conv_layers))]
        rows.append(("features before post_extract_proj", float(mx[8]), int(r[8, 1])))
        for l in range(self.cfg.layers):
            for k, kind in enumerate(self.RANGE_KINDS):
                i = 10 + 5 * l + k
                rows.append((f"layer {l} {kind}", float(mx[i]), int(r[i, 1])))
        if self.head == "sls":
            rows.append(("SLS pooled features", float(mx[9]), int(r[9, 1])))
        return rows

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._layer_event = None
            self.lib.slsmi_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_tensors(self, items: Iterable[Tuple[str, torch.Tensor]]) -> None:
        """items: (reference state-dict key without 'module.', tensor).  Tensors
        are moved to the engine's device as fp32 if they are not there yet."""
        with torch.cuda.device(self.device):
            sp = _stream_ptr(self.device)
            keep = []
            for name, src in items:
                if not src.is_floating_point():
                    continue  # sae.k, num_batches_tracked
                t = src.detach().to(self.device, torch.float32).contiguous()
                self._unbacked.discard(name)
                if t.data_ptr() != src.data_ptr():  # loaded from a copy (other device, dtype or strides)
                    self._staged[name] = (src.detach(), t)
                else:
                    self._staged.pop(name, None)
                keep.append(t)
                shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                rc = self.lib.slsmi_load_weight(self._ctx, name.encode(), C.c_void_p(t.data_ptr()), 0, shape,
                                                t.dim(), C.c_void_p(sp))
                _lib.check(rc, self._ctx, f"slsmi_load_weight({name})")
            _lib.check(self.lib.slsmi_finalize(self._ctx, C.c_void_p(sp)), self._ctx, "slsmi_finalize")
            del keep
            # reload_weights reads the same buffers again: the few copies are kept and refreshed from their
            # sources then (the in-tree model's sae.encoder.weight is a transposed view); a state dict that
            # came from elsewhere altogether is not held on to
            if sum(t.numel() * 4 for _, t in self._staged.values()) > (64 << 20):
                self._unbacked.update(self._staged)  # their copies go: nothing to read again for these names
                self._staged = {}
            self.reloadable = not self._unbacked

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.load_tensors(sd.items())

    def reload_weights(self) -> None:
        """Reads every weight again from the tensor it was loaded from (one library call): for parameters an
        optimizer updated in place.  Only valid while `reloadable` holds and those tensors are alive at the
        same addresses."""
        if not self.reloadable:
            raise _lib.SlsmiError("weights were loaded from temporary copies: load them again instead")
        with torch.cuda.device(self.device):
            for src, buf in self._staged.values():
                buf.copy_(src)
            _lib.check(self.lib.slsmi_reload_weights(self._ctx, C.c_void_p(_stream_ptr(self.device))), self._ctx,
                       "slsmi_reload_weights")

    # ------------------------------------------------------------------ forward
    def num_frames(self, length: int) -> int:
        t = self.lib.slsmi_num_frames(self._ctx, int(length))
        if t < 0:
            raise _lib.SlsmiError(f"input of {length} samples is too short for the conv stack")
        return t

    def workspace(self, batch: int, length: int) -> torch.Tensor:
        key = (batch, length)
        if self._ws_key != key:
            n = self.lib.slsmi_workspace_bytes(self._ctx, batch, length)
            if n == 0:
                raise _lib.SlsmiError(f"unsupported batch/length {key}")
            if self._ws is None or self._ws.numel() < n:
                self._ws = None
                self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def forward(self, wav: torch.Tensor, want_x: bool = False, want_layers: bool = False,
                want_sae_loss: bool = False, want_sae_code: bool = False,
                lengths: Optional[torch.Tensor] = None):
        """wav [B, L] fp32 on the engine's device -> dict(logp [B,2], x, layers, sae_loss, sae_code).
        sae_code: a copy of the TopK sparse code [B, T, dict] the head computed.
        lengths: None, or the sample count of every utterance of a padded batch ([B] integers, any
        device): utterance b is scored from its first lengths[b] samples alone."""
        if wav.device != self.device:
            raise _lib.SlsmiError(f"input on {wav.device}, engine on {self.device}")
        wav = wav.to(torch.float32).contiguous()
        B, L = wav.shape
        if lengths is not None:
            lengths = torch.as_tensor(lengths)
            if lengths.shape != (B,) or lengths.is_floating_point():
                raise _lib.SlsmiError(f"lengths must be {B} integers, got {tuple(lengths.shape)} {lengths.dtype}")
            if lengths.device.type == "cpu" and (int(lengths.min()) < 1 or int(lengths.max()) > L):
                raise _lib.SlsmiError(f"lengths must lie in [1, {L}]")
            lengths = lengths.to(self.device, torch.int32).contiguous()
        T = self.num_frames(L)
        with torch.cuda.device(self.device):
            ws = self.workspace(B, L)
            out: Dict[str, Optional[torch.Tensor]] = {"logp": None, "x": None, "layers": None, "sae_loss": None}
            if self.head != "none":
                out["logp"] = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            if want_x:
                out["x"] = torch.empty(B, T, 1024, dtype=torch.float32, device=self.device)
            if want_layers:
                out["layers"] = torch.empty(self.cfg.layers, B, T, 1024, dtype=torch.float32, device=self.device)
            if want_sae_loss:
                out["sae_loss"] = torch.empty(1, dtype=torch.float32, device=self.device)
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
            if lengths is None:
                rc = self.lib.slsmi_forward(self._ctx, ptr(wav), B, L, ptr(ws), ws.numel(), ptr(out["logp"]),
                                            ptr(out["x"]), ptr(out["layers"]), ptr(out["sae_loss"]),
                                            C.c_void_p(_stream_ptr(self.device)))
            else:
                if want_sae_loss:
                    raise _lib.SlsmiError("the SAE loss is not defined for a variable-length batch")
                rc = self.lib.slsmi_forward_varlen(self._ctx, ptr(wav), B, L, ptr(lengths), ptr(ws), ws.numel(),
                                                   ptr(out["logp"]), ptr(out["x"]), ptr(out["layers"]),
                                                   C.c_void_p(_stream_ptr(self.device)))
            _lib.check(rc, self._ctx, "slsmi_forward")
            out["sae_code"] = None
            if want_sae_code:
                off, nbytes = C.c_size_t(0), C.c_size_t(0)
                _lib.check(self.lib.slsmi_workspace_view(self._ctx, B, L, _lib.VIEW_SAE_CODE, C.byref(off),
                                                         C.byref(nbytes)), self._ctx, "slsmi_workspace_view")
                code = ws[off.value:off.value + nbytes.value].view(torch.float32)
                out["sae_code"] = code.view(B, T, -1).clone()
        return out

    # ------------------------------------------------------------------ training step
    def enable_training(self) -> None:
        """Training mode: the next weight load also builds what the backward needs."""
        _lib.check(self.lib.slsmi_enable_training(self._ctx), self._ctx, "slsmi_enable_training")
        self.training_enabled = True

    def set_layer_grads_event(self, event: Optional["torch.cuda.Event"]) -> None:
        """A CUDA/HIP event every later backward records once all encoder-layer parameter gradients are enqueued
        (slsmi_train_set_layer_grads_event); None withdraws it.  The engine keeps the event alive."""
        handle = None
        if event is not None:
            with torch.cuda.device(self.device):
                event.record(torch.cuda.current_stream(self.device))   # events are created on first use
            handle = C.c_void_p(event.cuda_event)
        _lib.check(self.lib.slsmi_train_set_layer_grads_event(self._ctx, handle), self._ctx,
                   "slsmi_train_set_layer_grads_event")
        self._layer_event = event

    def set_fused_adam(self, entries, lr: float, betas, eps: float, weight_decay: float) -> None:
        """The encoder-layer weight matrices that take their Adam step inside the NEXT training backward instead of
        getting a gradient (slsmi_train_set_fused_adam): entries = (state-dict name, parameter, exp_avg, exp_avg_sq,
        the tensor's step count of the coming step).  Used up by that backward."""
        _lib.check(self.lib.slsmi_train_set_fused_adam_hyper(self._ctx, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                             float(weight_decay)), self._ctx, "slsmi_train_set_fused_adam_hyper")
        for name, p, m, v, step in entries:
            for t in (p, m, v):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device or t.shape != p.shape:
                    raise _lib.SlsmiError(f"fused Adam: {name} and its moments must be contiguous fp32 tensors of one shape on {self.device}")
            _lib.check(self.lib.slsmi_train_set_fused_adam(self._ctx, name.encode(), C.c_void_p(p.data_ptr()),
                                                           C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), int(step)),
                       self._ctx, f"slsmi_train_set_fused_adam({name})")

    def drop_fused_copies(self) -> None:
        """The next reload_weights reads EVERY tensor again, also those whose 16-bit copies the last backward wrote
        with their fused Adam step (slsmi_train_drop_fused_copies): for a caller who changed them since."""
        _lib.check(self.lib.slsmi_train_drop_fused_copies(self._ctx), self._ctx, "slsmi_train_drop_fused_copies")

    def _fixed(self, name: str, shape, src: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A float32 device buffer that keeps its address from step to step, optionally filled from `src`."""
        t = self._tbuf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
            self._tbuf[name] = t
        if src is not None:
            t.copy_(src.detach().reshape(t.shape), non_blocking=True)
        return t

    def grad_buffers(self, params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One fp32 gradient tensor per named parameter, fresh from torch's caching allocator: autograd takes
        them over as .grad without a copy (buffers the engine kept would be cloned, 1.27 GB per step:
        23.7 against 22.5 ms per step).  The encoder layers' weight matrices of one kind are slices of one tensor."""
        out: Dict[str, torch.Tensor] = {}
        stacks: Dict[tuple, list] = {}
        for n, p in params.items():
            m = _LAYER_WEIGHT.match(n)
            if m and p.dim() == 2:
                stacks.setdefault((m.group(1), m.group(3), tuple(p.shape)), []).append((int(m.group(2)), n))
            else:
                out[n] = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        # the encoder layers' matrices of one kind as slices of one tensor, in layer order: at equal distances the
        # library writes all of them with one GEMM launch (slsmi_train_backward)
        for (_, _, shape), items in stacks.items():
            items.sort()
            whole = torch.empty((len(items),) + shape, dtype=torch.float32, device=self.device)
            for i, (_, n) in enumerate(items):
                out[n] = whole[i]
        return {n: out[n] for n in params}

    def grad_names(self, names: Iterable[str]):
        """The parameters among `names` (reference state-dict keys) whose gradient the library computes."""
        pre = "ssl_model.model."
        keep = (pre + "encoder.", pre + "post_extract_proj.", pre + "layer_norm.", pre + "feature_extractor.", "sae.",
                "classifier.")
        if self.head == "sls":   # slsmi_train_sls_backward also writes the upstream head's
            keep += ("fc0.", "first_bn.weight", "first_bn.bias", "fc1.", "fc3.")
        return [n for n in names if n.startswith(keep) and not n.endswith("sae.k")]

    def _train_ws(self, B: int, L: int) -> torch.Tensor:
        n = self.lib.slsmi_train_workspace_bytes(self._ctx, B, L)
        if n == 0:
            raise _lib.SlsmiError(f"unsupported batch/length for training: {(B, L)} (batch <= 64, at most 256 frames)")
        if self._tws is None or self._tws.numel() < n:
            self._tws = None
            self._tws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._tws

    def train_forward(self, wav: torch.Tensor, dropout_mask: Optional[torch.Tensor] = None, want_sae_loss: bool = True):
        """wav [B, L] -> (logp [B, 2], sae_loss [1] or None), keeping the intermediates for train_backward."""
        if wav.device != self.device:
            raise _lib.SlsmiError(f"input on {wav.device}, engine on {self.device}")
        B, L = wav.shape
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L), wav)       # the backward reads this copy again
            logp = self._fixed("logp", (B, 2))
            loss = self._fixed("loss", (1,)) if want_sae_loss else None
            if dropout_mask is not None:
                if tuple(dropout_mask.shape) != (B, 256):
                    raise _lib.SlsmiError(f"dropout mask must be [{B}, 256], got {tuple(dropout_mask.shape)}")
                dropout_mask = self._fixed("mask", (B, 256), dropout_mask.to(self.device))
            rc = self.lib.slsmi_train_forward(self._ctx, ptr(wav), B, L, ptr(ws), ws.numel(), ptr(dropout_mask), ptr(logp),
                                              ptr(loss), C.c_void_p(_stream_ptr(self.device)))
            _lib.check(rc, self._ctx, "slsmi_train_forward")
            return logp.clone(), (loss.clone() if loss is not None else None)

    def train_backward(self, wav: torch.Tensor, dlogp: torch.Tensor, dloss: Optional[torch.Tensor],
                       grads: Dict[str, torch.Tensor], used_mask: bool) -> None:
        """Writes the gradient of every parameter named in `grads` into its tensor (fp32, on the device).
        The input is the copy train_forward kept (`wav` gives the batch shape)."""
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        B, L = wav.shape
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L))
            want = {}
            for name, g in grads.items():
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.device:
                    raise _lib.SlsmiError(f"gradient buffer of {name} must be a contiguous fp32 tensor on {self.device}")
                want[name] = g.data_ptr()
            for name in [n for n in self._registered if n not in want]:
                self.lib.slsmi_set_grad(self._ctx, name.encode(), None)
                del self._registered[name]
            for name, p in want.items():
                if self._registered.get(name) != p:
                    _lib.check(self.lib.slsmi_set_grad(self._ctx, name.encode(), C.c_void_p(p)), self._ctx,
                               f"slsmi_set_grad({name})")
                    self._registered[name] = p
            dlogp = self._fixed("dlogp", (B, 2), dlogp.to(self.device))
            if dloss is not None:
                dloss = self._fixed("dloss", (1,), dloss.to(self.device))
            try:
                rc = self.lib.slsmi_train_backward(self._ctx, ptr(wav), B, L, ptr(ws), ws.numel(), ptr(dlogp), ptr(dloss),
                                                   1 if used_mask else 0, C.c_void_p(_stream_ptr(self.device)))
                _lib.check(rc, self._ctx, "slsmi_train_backward")
            finally:
                for name in grads:  # the buffers are the caller's again, also after an error
                    self.lib.slsmi_set_grad(self._ctx, name.encode(), None)
                self._registered.clear()

    MASK_KEYS = ("input", "encoder", "attn_out", "ffn_act", "ffn_out")

    def set_train_masks(self, masks: Optional[Dict[str, torch.Tensor]], B: int = 0, T: int = 0) -> None:
        """The backbone's train-mode dropout / layerdrop draws for the next training forward and its backward
        (slsmi_train_set_masks): multipliers "input", "encoder" [B, T, 1024], "attn_out", "ffn_out"
        [layers, B, T, 1024], "ffn_act" [layers, B, T, ffn], and "layer_keep" [layers] booleans; any subset.
        The tensors are held until the next call."""
        self._mask_hold = None
        self._set_attention_masks(masks, B, T)
        if not masks:
            _lib.check(self.lib.slsmi_train_set_masks(self._ctx, None), self._ctx, "slsmi_train_set_masks")
            return
        unknown = set(masks) - set(self.MASK_KEYS) - {"layer_keep", "attn", "attn_keep", "attn_p"}
        if unknown:
            raise _lib.SlsmiError(f"unknown backbone masks {sorted(unknown)}")
        L, Fd = self.cfg.layers, self.cfg.ffn_dim
        shapes = {"input": (B, T, 1024), "encoder": (B, T, 1024), "attn_out": (L, B, T, 1024), "ffn_act": (L, B, T, Fd),
                  "ffn_out": (L, B, T, 1024)}
        st = _lib.TrainMasks()
        hold = []
        for k in self.MASK_KEYS:
            m = masks.get(k)
            if m is None:
                continue
            if tuple(m.shape) != shapes[k]:
                raise _lib.SlsmiError(f"backbone mask {k!r} must have shape {shapes[k]}, got {tuple(m.shape)}")
            m = m.to(self.device, torch.float32).contiguous()
            hold.append(m)
            setattr(st, k, m.data_ptr())
        keep = masks.get("layer_keep")
        if keep is not None:
            keep = [bool(v) for v in (keep.tolist() if hasattr(keep, "tolist") else keep)]
            if len(keep) != L:
                raise _lib.SlsmiError(f"layer_keep must have {L} entries")
            arr = (C.c_uint8 * L)(*[1 if v else 0 for v in keep])
            hold.append(arr)
            st.layer_keep = C.cast(arr, C.c_void_p).value
        _lib.check(self.lib.slsmi_train_set_masks(self._ctx, C.byref(st)), self._ctx, "slsmi_train_set_masks")
        self._mask_hold = hold

    def _set_attention_masks(self, masks: Optional[Dict[str, torch.Tensor]], B: int, T: int) -> None:
        """MultiheadAttention's dropout on the attention probabilities (slsmi_train_set_attention_masks), from
        "attn" -- multipliers 0 or 1 / (1 - p), [layers, B, 16, T, T] (query, key) -- or from "attn_keep" (booleans of
        that shape) with "attn_p" (the drop probability).  The library takes keep bytes in both orientations with
        rows padded to whole 32-key steps."""
        self._attn_hold = None
        m = None if not masks else masks.get("attn")
        keep = None if not masks else masks.get("attn_keep")
        if m is None and keep is None:
            _lib.check(self.lib.slsmi_train_set_attention_masks(self._ctx, None, None, 0, 0, 1.0), self._ctx,
                       "slsmi_train_set_attention_masks")
            return
        L = self.cfg.layers
        if m is not None:
            m = m.to(self.device)
            keep = m != 0
            nz = m[keep]
            scale = float(nz.max()) if nz.numel() else 1.0
            if nz.numel() and float(nz.min()) != scale:
                raise _lib.SlsmiError('backbone mask "attn" must hold 0 and one other value, 1 / (1 - p)')
        else:
            keep = keep.to(self.device) != 0
            scale = 1.0 / (1.0 - float(masks["attn_p"]))
        if tuple(keep.shape) != (L, B, 16, T, T):
            raise _lib.SlsmiError(f"attention masks must have shape {(L, B, 16, T, T)}, got {tuple(keep.shape)}")
        pitch = (T + 31) // 32 * 32
        qk = torch.zeros(L, B * 16, T, pitch, dtype=torch.uint8, device=self.device)
        kq = torch.zeros(L, B * 16, T, pitch, dtype=torch.uint8, device=self.device)
        k8 = keep.reshape(L, B * 16, T, T).to(torch.uint8)
        qk[..., :T] = k8
        kq[..., :T] = k8.transpose(-1, -2)
        self._attn_hold = (qk, kq)
        _lib.check(self.lib.slsmi_train_set_attention_masks(self._ctx, C.c_void_p(qk.data_ptr()), C.c_void_p(kq.data_ptr()),
                                                            B * 16 * T * pitch, pitch, scale), self._ctx,
                   "slsmi_train_set_attention_masks")

    def train_backbone_forward(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B, L] -> the encoder layers' outputs [layers, B, T, 1024] (fp32, a tensor of its own), keeping the
        intermediates for train_backbone_backward (slsmi_train_backbone_forward)."""
        if wav.device != self.device:
            raise _lib.SlsmiError(f"input on {wav.device}, engine on {self.device}")
        B, L = wav.shape
        T = self.num_frames(L)
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L), wav)       # the backward reads this copy again
            layers = torch.empty(self.cfg.layers, B, T, 1024, dtype=torch.float32, device=self.device)
            rc = self.lib.slsmi_train_backbone_forward(self._ctx, C.c_void_p(wav.data_ptr()), B, L,
                                                       C.c_void_p(ws.data_ptr()), ws.numel(),
                                                       C.c_void_p(layers.data_ptr()), C.c_void_p(_stream_ptr(self.device)))
            _lib.check(rc, self._ctx, "slsmi_train_backbone_forward")
        return layers

    def train_backbone_backward(self, wav: torch.Tensor, dlayers: torch.Tensor, grads: Dict[str, torch.Tensor]) -> None:
        """The gradient of every backbone parameter named in `grads`, from the loss gradient at the layers'
        outputs (dlayers [layers, B, T, 1024])."""
        B, L = wav.shape
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L))
            dlayers = dlayers.to(self.device, torch.float32).contiguous()
            if tuple(dlayers.shape) != (self.cfg.layers, B, self.num_frames(L), 1024):
                raise _lib.SlsmiError(f"dlayers has shape {tuple(dlayers.shape)}")
            for name, g in grads.items():
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.device:
                    raise _lib.SlsmiError(f"gradient buffer of {name} must be a contiguous fp32 tensor on {self.device}")
            for name in list(self._registered):
                self.lib.slsmi_set_grad(self._ctx, name.encode(), None)
            self._registered.clear()
            try:
                for name, g in grads.items():
                    _lib.check(self.lib.slsmi_set_grad(self._ctx, name.encode(), C.c_void_p(g.data_ptr())), self._ctx,
                               f"slsmi_set_grad({name})")
                rc = self.lib.slsmi_train_backbone_backward(self._ctx, C.c_void_p(wav.data_ptr()), B, L,
                                                            C.c_void_p(ws.data_ptr()), ws.numel(),
                                                            C.c_void_p(dlayers.data_ptr()),
                                                            C.c_void_p(_stream_ptr(self.device)))
                _lib.check(rc, self._ctx, "slsmi_train_backbone_backward")
            finally:
                for name in grads:
                    self.lib.slsmi_set_grad(self._ctx, name.encode(), None)

    def _register_grads(self, grads: Dict[str, torch.Tensor]) -> None:
        for name, g in grads.items():
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.device:
                raise _lib.SlsmiError(f"gradient buffer of {name} must be a contiguous fp32 tensor on {self.device}")
        for name in list(self._registered):
            self.lib.slsmi_set_grad(self._ctx, name.encode(), None)
        self._registered.clear()
        for name, g in grads.items():
            _lib.check(self.lib.slsmi_set_grad(self._ctx, name.encode(), C.c_void_p(g.data_ptr())), self._ctx,
                       f"slsmi_set_grad({name})")

    def _fc1_weight(self, fc1_weight: torch.Tensor) -> torch.Tensor:
        if fc1_weight.dtype != torch.float32 or not fc1_weight.is_contiguous() or fc1_weight.device != self.device:
            raise _lib.SlsmiError(f"fc1.weight must be a contiguous fp32 tensor on {self.device}")
        return fc1_weight.detach()

    def train_sls_forward(self, wav: torch.Tensor, fc1_weight: torch.Tensor):
        """The upstream SLS model's training forward (slsmi_train_sls_forward): wav [B, L] -> (logp [B, 2],
        batch statistics [2] = mean and unbiased variance of the BatchNorm's input), keeping the intermediates for
        train_sls_backward.  fc1_weight: the model's fp32 fc1.weight parameter."""
        if wav.device != self.device:
            raise _lib.SlsmiError(f"input on {wav.device}, engine on {self.device}")
        B, L = wav.shape
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L), wav)       # the backward reads this copy again
            w1 = self._fc1_weight(fc1_weight)
            logp = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            stats = torch.empty(2, dtype=torch.float32, device=self.device)
            rc = self.lib.slsmi_train_sls_forward(self._ctx, C.c_void_p(wav.data_ptr()), B, L, C.c_void_p(ws.data_ptr()),
                                                  ws.numel(), C.c_void_p(w1.data_ptr()), C.c_void_p(logp.data_ptr()),
                                                  C.c_void_p(stats.data_ptr()), C.c_void_p(_stream_ptr(self.device)))
            _lib.check(rc, self._ctx, "slsmi_train_sls_forward")
        return logp, stats

    def train_sls_backward(self, wav: torch.Tensor, dlogp: torch.Tensor, grads: Dict[str, torch.Tensor],
                           fc1_weight: torch.Tensor) -> None:
        """The gradient of every parameter named in `grads` -- backbone and SLS head -- from dLoss/dlogp [B, 2]
        (slsmi_train_sls_backward)."""
        B, L = wav.shape
        with torch.cuda.device(self.device):
            ws = self._train_ws(B, L)
            wav = self._fixed("wav", (B, L))
            w1 = self._fc1_weight(fc1_weight)
            dlogp = self._fixed("dlogp", (B, 2), dlogp.to(self.device))
            try:
                self._register_grads(grads)
                rc = self.lib.slsmi_train_sls_backward(self._ctx, C.c_void_p(wav.data_ptr()), B, L, C.c_void_p(ws.data_ptr()),
                                                       ws.numel(), C.c_void_p(w1.data_ptr()), C.c_void_p(dlogp.data_ptr()),
                                                       C.c_void_p(_stream_ptr(self.device)))
                _lib.check(rc, self._ctx, "slsmi_train_sls_backward")
            finally:
                for name in grads:
                    self.lib.slsmi_set_grad(self._ctx, name.encode(), None)
