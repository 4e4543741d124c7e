"""Training around the HIP step: what main.py's train_epoch does per batch (main.py:201-258) and the
multi-GPU form of it.  The reference trains under single-process nn.DataParallel (main.py:518); here
it is one process per GPU, every rank on its own utterances, and one all-reduce of the gradients over
RCCL per step (GradSync).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .weights import BACKBONE_PREFIX

# pre-training leftovers of the fairseq checkpoint that the forward never reads (wav2vec2.py:313-362)
_UNUSED = ("mask_emb", "quantizer.", "project_q.", "final_proj.")


def unused_parameter_names(model: nn.Module) -> List[str]:
    """Parameters no forward of this model reads: they never get a gradient."""
    core = model.module if hasattr(model, "module") else model
    unused = _UNUSED
    if getattr(core, "head", None) == "sls":  # the SLS head reads the layers' outputs, not the final LayerNorm's
        unused = unused + ("encoder.layer_norm.",)
    strip = len("module.") if core is not model else 0
    return [n for n, _ in model.named_parameters()
            if n[strip:].startswith(BACKBONE_PREFIX) and n[strip + len(BACKBONE_PREFIX):].startswith(unused)]


def freeze_unused(model: nn.Module) -> int:
    """requires_grad = False on the parameters the forward never reads (they never get a gradient in
    the reference either, so its Adam skips them too); keeps them out of the gradient exchange."""
    names = set(unused_parameter_names(model))
    for n, p in model.named_parameters():
        if n in names:
            p.requires_grad_(False)
    return len(names)


class GradSync:
    """Data-parallel gradient exchange for one process per GPU: all parameter gradients are packed into
    ONE flat fp32 buffer (1.27 GB for the full model), all-reduced in a single collective and averaged.
    The HIP backward is one autograd node, so every gradient appears at the same moment and there is
    nothing to overlap bucket by bucket; one large collective pays RCCL's per-call latency once and keeps
    all of a GPU's point-to-point xGMI links busy for its whole length.
    comm_device: where the collective runs -- None = the gradients' own device (RCCL); a CPU device lets
    the same code run over gloo (tests, rehearsals).
    comm_dtype: torch.float32 (default: the exchange adds exactly what single-process training would) or
    torch.bfloat16, which halves the bytes on the xGMI links (0.64 GB per step); every rank rounds its
    gradients to bfloat16 before the sum, the average is applied in fp32."""

    def __init__(self, params, group=None, comm_device: Optional[torch.device] = None,
                 comm_dtype: torch.dtype = torch.float32):
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be torch.float32 or torch.bfloat16")
        self.params = [p for p in params if p.requires_grad]
        self.group, self.comm_device, self.comm_dtype = group, comm_device, comm_dtype
        self._flat: Optional[torch.Tensor] = None

    def sync(self) -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        ps = [p for p in self.params if p.grad is not None]
        n = sum(p.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n:
            self._flat = torch.empty(n, dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(self._flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        torch._foreach_copy_(views, [p.grad for p in ps])
        buf = self._flat if self.comm_dtype == torch.float32 else self._flat.to(self.comm_dtype)
        if self.comm_device is not None:
            buf = buf.to(self.comm_device)
        dist.all_reduce(buf, group=self.group)
        if buf is not self._flat:
            self._flat.copy_(buf)
        self._flat.mul_(1.0 / dist.get_world_size(self.group))
        torch._foreach_copy_([p.grad for p in ps], views)


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) of the reference's recipe (main.py:526) with the step
    of ALL parameter tensors in one HIP launch (slsmi_adam_step): the same arithmetic (single-tensor form, L2
    weight decay, no amsgrad) and the same state entries ('step', 'exp_avg', 'exp_avg_sq'), so a state dict
    moves between the two.  fp32 parameters on one ROCm device; the step is asynchronous on the current stream.

    What does not change between steps (work items, parameter and moment addresses) is prepared once; the
    gradients' addresses are read every step (autograd may attach new gradient tensors) and reach the device in
    one small asynchronous copy.

    fuse_backward (optional, a slsmi Model in one process on one GPU): the encoder layers' projection and feed-forward
    weight matrices among `params` (95 % of the model's bytes) take their step INSIDE loss.backward(), in the epilogue
    of the kernel that computes their gradient (slsmi_train_set_fused_adam): the gradient is never stored (their
    .grad stays None), p / exp_avg / exp_avg_sq are updated in place with this optimizer's hyper-parameters and state
    -- the same bits as backward() followed by step() -- and step() then handles the other parameters as always.  For
    loops in which every backward is followed by exactly one step and nothing in between looks at those gradients
    (main.py:236-238 is such a loop; gradient accumulation, clipping or an exchange between ranks are not: leave it
    out there).  A layer skipped by layerdrop takes no step and keeps its step count, as with torch."""

    CHUNK = 4096    # elements per workgroup (16 per thread: 5.6 TB/s over the model's 318 M values; 16384: 5.2, 65536: 5.0)

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fuse_backward: Optional[nn.Module] = None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        self._tables = {}
        self._fused_ids = frozenset()
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        core = fuse_backward.module if (fuse_backward is not None and hasattr(fuse_backward, "module")) else fuse_backward
        if core is not None:
            if not hasattr(core, "_fused_adam_begin"):
                raise ValueError("fuse_backward must be a slsmi Model")
            import re
            import weakref
            kind = re.compile(r"^" + re.escape(BACKBONE_PREFIX) + r"encoder\.layers\.\d+\.(self_attn\.(q|k|v|out)_proj|fc1|fc2)\.weight$")
            mine = {id(p) for g in self.param_groups for p in g["params"]}
            self._fused_ids = frozenset(id(p) for n, p in core.named_parameters()
                                        if kind.match(n) and p.requires_grad and id(p) in mine)
            core.__dict__["_fused_opt"] = weakref.ref(self)

    def _fuse_into_backward(self, eng, want):
        """The backward's question (Model._fused_adam_begin): which of the parameters it is about to differentiate step
        inside it.  Advances their step counts, creates their moments on first use and hands the engine their addresses."""
        names = [n for n, p in want.items() if id(p) in self._fused_ids]
        if not names:
            return frozenset()
        group = None
        for g in self.param_groups:
            ids = {id(p) for p in g["params"]}
            if any(id(want[n]) in ids for n in names):
                if group is not None or not all(id(want[n]) in ids for n in names):
                    raise ValueError("slsmi.training.Adam(fuse_backward=): the encoder layers' weights must be in one parameter group")
                group = g
        entries = []
        with torch.no_grad():
            for n in names:
                p = want[n]
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                entries.append((n, p.data, st["exp_avg"], st["exp_avg_sq"], st["step"]))
        eng.set_fused_adam(entries, group["lr"], group["betas"], group["eps"], group["weight_decay"])
        return frozenset(names)

    def _plan(self, gi: int, ps, device):
        """What does not change from step to step for the group's tensors with a gradient: the table's rows without
        the gradient addresses (host), the work items (device), the parameters' strides.  Layout checks happen here."""
        import numpy as np
        from . import _lib
        key = tuple(id(p) for p in ps)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1]
        rows = np.zeros((len(ps), 6), dtype=np.int64)
        for t, p in enumerate(ps):
            st = self.state[p]
            # the step is elementwise over memory: parameter, moments (and, checked every step, the gradient) must
            # cover their memory densely in ONE layout (contiguous, or e.g. the same transposed view: sae.encoder.weight)
            three = (p, st["exp_avg"], st["exp_avg_sq"])
            if any(x.device != device or x.dtype != torch.float32 or x.stride() != p.stride() for x in three) or \
                    device.type != "cuda" or not torch._prims_common.is_non_overlapping_and_dense(p):
                raise _lib.SlsmiError("slsmi.training.Adam steps dense fp32 parameters on one ROCm device whose moments "
                                      "have the parameter's own memory layout")
            rows[t] = (p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), 0)
        counts = (rows[:, 4] + self.CHUNK - 1) // self.CHUNK
        tensor_of = np.repeat(np.arange(len(ps)), counts)
        chunk_of = np.arange(int(counts.sum())) - np.repeat(np.cumsum(counts) - counts, counts)
        items = torch.from_numpy(np.stack([tensor_of, chunk_of], axis=1).astype(np.int32)).to(device)
        plan = {"rows": torch.from_numpy(rows), "items": items, "n_items": int(counts.sum()),
                "strides": [p.stride() for p in ps]}
        self._tables[gi] = (key, plan)
        return plan

    def _table(self, plan, ps, device):
        """The device table of this step: the plan's rows with the gradients' addresses of now (autograd may hand a
        parameter a new gradient tensor every step) and the tensors' own step counts.  Staged in pinned memory and
        copied asynchronously: a pageable copy would wait for everything queued on the device."""
        from . import _lib
        ptrs = []
        for p, stride in zip(ps, plan["strides"]):
            g = p.grad
            if g.stride() != stride or g.dtype is not torch.float32 or g.device != device:
                raise _lib.SlsmiError("slsmi.training.Adam: a gradient is not fp32 in its parameter's memory layout on its device")
            ptrs.append(g.data_ptr())
        rows = plan["rows"].clone()
        rows[:, 1] = torch.tensor(ptrs, dtype=torch.int64)
        rows[:, 5] = torch.tensor([int(self.state[p]["step"]) for p in ps], dtype=torch.int64)
        return rows.pin_memory().to(device, non_blocking=True)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}   # the moment tensors were replaced

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            device = ps[0].device
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
            # (torch keeps one step count per parameter: one without a gradient in some step -- a layer skipped by
            # layerdrop -- is not touched and falls behind; the kernel takes every tensor's own count)
            plan = self._plan(gi, ps, device)
            if plan["n_items"] == 0:   # nothing but empty tensors
                continue
            tab = self._table(plan, ps, device)
            b1, b2 = group["betas"]
            with torch.cuda.device(device):
                rc = _lib.load().slsmi_adam_step(tab.data_ptr(), plan["items"].data_ptr(), plan["n_items"], self.CHUNK,
                                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                 float(group["weight_decay"]),
                                                 torch.cuda.current_stream(device).cuda_stream)
            _lib.check(rc, None, "slsmi_adam_step")
        return loss


class ShardedAdam:
    """main.py:526's Adam for one process per GPU with the optimizer state and its work divided over the ranks:

        reduce-scatter of the gradients  ->  Adam on this rank's 1/W of every bucket  ->  all-gather of the parameters

    instead of GradSync's all-reduce followed by W identical full-size Adam steps.  Per step and rank the links carry
    the same bytes as a ring all-reduce would (each phase moves (W - 1)/W of a bucket in and out), but Adam reads
    and writes 1/W of the 8.8 GB it touches on one GPU (1.82 ms -> ~0.25 ms on 8), the moments take 1/W of the memory,
    and the averaged gradients are never written back.  The parameters live in ONE flat fp32 buffer per bucket (each
    nn.Parameter becomes a view of its slice, values kept), so the all-gather lands in place and the library's
    weight reload reads the new values where it always did; the gradients are packed into a flat buffer of
    `comm_dtype` (one pass; bfloat16 halves the bytes on the links, the sum then adds rounded terms) whose rank-r
    slice the reduce-scatter fills with the sum.

    Two buckets: the encoder layers' parameters (95 % of the bytes) and the rest.  With `model` given, the first
    bucket's pack and reduce-scatter are enqueued on a side stream that waits only for the event the backward
    records when those gradients are complete (Model.layer_grads_event), so they overlap the backward's tail (the
    positional conv's and the feature extractor's gradients); the second bucket follows on the main stream.

    Semantics are torch.optim.Adam's, per parameter: a parameter whose grad is None on this rank (a layer skipped
    by layerdrop: give every rank the same layer draws -- Model takes them from numpy's global generator) is left
    out of the step on the ranks that own parts of it, and its step count does not advance.
    kernel: "hip" = slsmi_adam_step over the rank's segments in one launch; "torch" = the same arithmetic with
    torch operations (any device: what the CPU tests run)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 group=None, comm_dtype: torch.dtype = torch.float32, kernel: str = "hip", model: Optional[nn.Module] = None,
                 names: Optional[List[str]] = None):
        import torch.distributed as dist
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be torch.float32 or torch.bfloat16")
        if kernel not in ("hip", "torch"):
            raise ValueError("kernel must be 'hip' or 'torch'")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.group, self.comm_dtype, self.kernel = group, comm_dtype, kernel
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        core = model.module if (model is not None and hasattr(model, "module")) else model
        if core is not None and names is None:
            by_id = {id(p): n for n, p in core.named_parameters()}
            plist = [p for p in params if p.requires_grad]
            names = [by_id.get(id(p), "") for p in plist]
        else:
            plist = [p for p in params if p.requires_grad]
            names = list(names) if names is not None else [""] * len(plist)
        if not plist:
            raise ValueError("no parameters to optimise")
        self.device = plist[0].device
        layer_tag = BACKBONE_PREFIX + "encoder.layers."
        first = [i for i, n in enumerate(names) if n.startswith(layer_tag)]
        rest = [i for i, n in enumerate(names) if not n.startswith(layer_tag)]
        self.buckets = []
        for idx in (first, rest):
            if idx:
                self.buckets.append(self._make_bucket([plist[i] for i in idx]))
        self.params = plist
        self.state_step = {id(p): 0 for p in plist}
        self._side = None
        self._event = None
        if core is not None and self.device.type == "cuda" and first and hasattr(core, "layer_grads_event"):
            self._event = core.layer_grads_event()
            self._side = torch.cuda.Stream(self.device)

    # ---- layout
    def _make_bucket(self, ps):
        W = self.world
        unit = 1024 * W   # every shard starts on a 4 KiB boundary
        offs, n = [], 0
        for p in ps:
            if p.dtype != torch.float32 or not torch._prims_common.is_non_overlapping_and_dense(p):
                raise ValueError("ShardedAdam takes dense fp32 parameters")
            offs.append(n)
            n += p.numel()
        total = (n + unit - 1) // unit * unit
        shard = total // W
        flat_p = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
        with torch.no_grad():
            for p, o in zip(ps, offs):   # the parameter becomes a view of its slice, in its own memory layout
                view = flat_p[o:o + p.numel()].as_strided(p.shape, p.stride())
                view.copy_(p.detach())
                p.data = view
        lo, hi = self.rank * shard, (self.rank + 1) * shard
        segs = []   # this rank's part of every parameter: (param index, offset in the bucket, length)
        for i, (p, o) in enumerate(zip(ps, offs)):
            a, b = max(o, lo), min(o + p.numel(), hi)
            if a < b:
                segs.append((i, a, b - a))
        return {"params": ps, "offs": offs, "total": total, "shard": shard, "lo": lo, "flat_p": flat_p,
                "flat_g": torch.zeros(total, dtype=self.comm_dtype, device=ps[0].device),
                "g": torch.zeros(shard, dtype=self.comm_dtype, device=ps[0].device),
                "m": torch.zeros(shard, dtype=torch.float32, device=ps[0].device),
                "v": torch.zeros(shard, dtype=torch.float32, device=ps[0].device), "segs": segs, "plan": None}

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # ---- one step
    def _pack(self, bk) -> list:
        """Gradients -> the bucket's flat buffer (a parameter without one contributes zeros).  Returns, per parameter
        of the bucket, whether it had a gradient."""
        ps, offs, flat = bk["params"], bk["offs"], bk["flat_g"]
        have = [p.grad is not None for p in ps]
        views = [flat[o:o + p.numel()].as_strided(p.shape, p.stride()) for p, o, h in zip(ps, offs, have) if h]
        if len(views) != len(ps):
            flat.zero_()
        if views:
            torch._foreach_copy_(views, [p.grad for p, h in zip(ps, have) if h])
        return have

    def _reduce(self, bk):
        import torch.distributed as dist
        if self.world == 1:
            bk["g"].copy_(bk["flat_g"][:bk["shard"]])
            return None
        return dist.reduce_scatter_tensor(bk["g"], bk["flat_g"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _adam(self, bk, have) -> None:
        segs = [(i, a, n) for i, a, n in bk["segs"] if have[i]]
        for i in {i for i, h in enumerate(have) if h}:
            self.state_step[id(bk["params"][i])] += 1
        if not segs:
            return
        lo, W = bk["lo"], self.world
        g32 = bk["g"] if bk["g"].dtype == torch.float32 else bk["g"].float()
        if W > 1:
            g32 = g32.mul_(1.0 / W) if g32 is not bk["g"] else g32.mul(1.0 / W)   # the average main.py's single process sees
        b1, b2 = self.betas
        if self.kernel == "hip":
            import numpy as np
            from . import _lib
            rows = np.zeros((len(segs), 6), dtype=np.int64)
            pbase, gbase, mbase, vbase = bk["flat_p"].data_ptr(), g32.data_ptr(), bk["m"].data_ptr(), bk["v"].data_ptr()
            for r, (i, a, n) in enumerate(segs):
                rows[r] = (pbase + 4 * a, gbase + 4 * (a - lo), mbase + 4 * (a - lo), vbase + 4 * (a - lo), n,
                           self.state_step[id(bk["params"][i])])
            chunk = Adam.CHUNK
            counts = (rows[:, 4] + chunk - 1) // chunk
            key = tuple((i, a, n) for i, a, n in segs)
            if bk["plan"] is None or bk["plan"][0] != key:
                tensor_of = np.repeat(np.arange(len(segs)), counts)
                chunk_of = np.arange(int(counts.sum())) - np.repeat(np.cumsum(counts) - counts, counts)
                items = torch.from_numpy(np.stack([tensor_of, chunk_of], axis=1).astype(np.int32)).to(self.device)
                bk["plan"] = (key, items)
            items = bk["plan"][1]
            tab = torch.from_numpy(rows).pin_memory().to(self.device, non_blocking=True)
            bk["_hold"] = (tab, g32)   # alive until the next step: the launch is asynchronous
            with torch.cuda.device(self.device):
                rc = _lib.load().slsmi_adam_step(tab.data_ptr(), items.data_ptr(), items.shape[0], chunk, self.lr, b1, b2,
                                                 self.eps, self.weight_decay,
                                                 torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, None, "slsmi_adam_step")
            return
        with torch.no_grad():   # torch.optim.Adam's single-tensor arithmetic on every segment
            for i, a, n in segs:
                t = self.state_step[id(bk["params"][i])]
                p = bk["flat_p"][a:a + n]
                g = g32[a - lo:a - lo + n]
                m, v = bk["m"][a - lo:a - lo + n], bk["v"][a - lo:a - lo + n]
                if self.weight_decay != 0:
                    g = g.add(p, alpha=self.weight_decay)
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
                p.addcdiv_(m, denom, value=-(self.lr / bc1))

    def _gather(self, bk):
        import torch.distributed as dist
        if self.world == 1:
            return None
        mine = bk["flat_p"][bk["lo"]:bk["lo"] + bk["shard"]]
        return dist.all_gather_into_tensor(bk["flat_p"], mine, group=self.group, async_op=True)

    @torch.no_grad()
    def step(self) -> None:
        cuda = self.device.type == "cuda"
        main = torch.cuda.current_stream(self.device) if cuda else None
        n = len(self.buckets)
        have, red = [None] * n, [None] * n
        for b, bk in enumerate(self.buckets):
            early = b == 0 and self._side is not None and n > 1
            if early:   # behind the backward's event, beside its tail
                self._side.wait_event(self._event)
                with torch.cuda.stream(self._side):
                    have[b] = self._pack(bk)
                    red[b] = self._reduce(bk)
            else:
                have[b] = self._pack(bk)
                red[b] = self._reduce(bk)
        gat = [None] * n
        for b, bk in enumerate(self.buckets):
            early = b == 0 and self._side is not None and n > 1
            ctx = torch.cuda.stream(self._side) if early else _nullcontext()
            with ctx:
                if red[b] is not None:
                    red[b].wait()
                self._adam(bk, have[b])
                gat[b] = self._gather(bk)
        for b, bk in enumerate(self.buckets):
            early = b == 0 and self._side is not None and n > 1
            with (torch.cuda.stream(self._side) if early else _nullcontext()):
                if gat[b] is not None:
                    gat[b].wait()
        if self._side is not None and n > 1:
            main.wait_stream(self._side)   # the next forward's weight reload reads what the side stream wrote

    # ---- state (every rank holds its own shard: save per rank, or gather before saving)
    def state_dict(self):
        return {"rank": self.rank, "world": self.world,
                "steps": [self.state_step[id(p)] for p in self.params],
                "buckets": [{"m": bk["m"].clone(), "v": bk["v"].clone()} for bk in self.buckets],
                "hyper": {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}}

    def consolidated_state_dict(self):
        """torch.optim.Adam's state dict for the same parameter list ('state': {i: step, exp_avg, exp_avg_sq},
        'param_groups'), assembled from every rank's shard of the moments (a collective: call it on all ranks) -- what
        main.py:753-764 stores as optimizer_state_dict, and what torch.optim.Adam(...).load_state_dict or
        load_consolidated_state_dict on any world size takes back."""
        import torch.distributed as dist
        state = {}
        index = {id(p): i for i, p in enumerate(self.params)}
        for bk in self.buckets:
            full_m = torch.empty(bk["total"], dtype=torch.float32, device=bk["m"].device)
            full_v = torch.empty_like(full_m)
            if self.world > 1:
                dist.all_gather_into_tensor(full_m, bk["m"], group=self.group)
                dist.all_gather_into_tensor(full_v, bk["v"], group=self.group)
            else:
                full_m.copy_(bk["m"])
                full_v.copy_(bk["v"])
            for p, o in zip(bk["params"], bk["offs"]):
                t = self.state_step[id(p)]
                if t == 0:
                    continue   # torch creates a parameter's state at its first step
                n = p.numel()
                state[index[id(p)]] = {"step": torch.tensor(float(t)),
                                       "exp_avg": full_m[o:o + n].as_strided(p.shape, p.stride()).clone(),
                                       "exp_avg_sq": full_v[o:o + n].as_strided(p.shape, p.stride()).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_consolidated_state_dict(self, sd) -> None:
        """Takes a torch.optim.Adam state dict over the same parameter list (any world size): every rank keeps its own
        shard of the moments."""
        with torch.no_grad():
            index = {id(p): i for i, p in enumerate(self.params)}
            for bk in self.buckets:
                full_m = torch.zeros(bk["total"], dtype=torch.float32, device=bk["m"].device)
                full_v = torch.zeros_like(full_m)
                for p, o in zip(bk["params"], bk["offs"]):
                    st = sd["state"].get(index[id(p)])
                    self.state_step[id(p)] = int(st["step"]) if st else 0
                    if st:
                        n = p.numel()
                        full_m[o:o + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg"])
                        full_v[o:o + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg_sq"])
                bk["m"].copy_(full_m[bk["lo"]:bk["lo"] + bk["shard"]])
                bk["v"].copy_(full_v[bk["lo"]:bk["lo"] + bk["shard"]])
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])

    def load_state_dict(self, sd) -> None:
        if sd["world"] != self.world or sd["rank"] != self.rank or len(sd["buckets"]) != len(self.buckets):
            raise ValueError("a ShardedAdam state belongs to the rank and world size it was saved on")
        for p, t in zip(self.params, sd["steps"]):
            self.state_step[id(p)] = int(t)
        for bk, s in zip(self.buckets, sd["buckets"]):
            bk["m"].copy_(s["m"])
            bk["v"].copy_(s["v"])


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def train_loader(dataset, batch_size: int = 5, num_workers: int = 8, rank: int = 0, world: int = 1, seed: int = 1234,
                 epoch: int = 0):
    """The reference's training DataLoader (main.py: batch_size from --batch_size, shuffle=True, drop_last=True,
    num_workers=8) for one process per GPU: rank r of `world` draws its own share of every epoch's
    permutation (DistributedSampler), and every worker process seeds numpy's global generator -- the one
    RawBoost draws from -- with (seed, epoch, rank, worker), so that workers do not repeat each other's
    augmentation (forked workers of the reference all inherit the same generator state)."""
    import numpy as np
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    sampler = None
    if world > 1:
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=True)
        sampler.set_epoch(epoch)

    def seed_worker(worker_id: int) -> None:
        np.random.seed(np.random.SeedSequence([seed, epoch, rank, worker_id]).generate_state(1)[0])

    g = torch.Generator()
    g.manual_seed(seed + epoch)
    return DataLoader(dataset, batch_size=batch_size, shuffle=sampler is None, sampler=sampler, drop_last=True,
                      num_workers=num_workers, worker_init_fn=seed_worker, generator=g, pin_memory=True,
                      persistent_workers=False)


_CLASS_WEIGHT = {}


def _default_class_weight(device: torch.device) -> torch.Tensor:
    """main.py:524's [0.1, 0.9], made once per device: building it from a Python list copies from pageable host
    memory, which waits for everything queued on the device -- once per step that would drain the queue the host
    otherwise keeps a step ahead of."""
    w = _CLASS_WEIGHT.get(device)
    if w is None:
        w = _CLASS_WEIGHT[device] = torch.tensor([0.1, 0.9], device=device)
    return w


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, batch_x: torch.Tensor, batch_y: torch.Tensor,
               class_weight: Optional[torch.Tensor] = None, dropout_mask: Optional[torch.Tensor] = None,
               grad_sync: Optional[GradSync] = None):
    """One iteration of main.py's train_epoch (main.py:224-238): forward with the reconstruction loss,
    NLLLoss with class weights (main.py:524-525), compute_total_loss (model.py:295-299), backward, step.
    Returns (total, classification, sae) losses as 0-d device tensors (no host synchronisation)."""
    core = model.module if hasattr(model, "module") else model
    if class_weight is None:
        class_weight = _default_class_weight(batch_x.device)
    kw = {} if dropout_mask is None else {"dropout_mask": dropout_mask}
    output, sae_loss = model(batch_x, return_sae_loss=True, **kw)
    cls_loss = F.nll_loss(output, batch_y.view(-1).long(), weight=class_weight)
    loss = core.compute_total_loss(cls_loss, sae_loss)
    optimizer.zero_grad()
    loss.backward()
    if grad_sync is not None:  # data parallel: average the gradients over the ranks
        grad_sync.sync()
    optimizer.step()
    return loss.detach(), cls_loss.detach(), (sae_loss.detach() if sae_loss is not None else None)
