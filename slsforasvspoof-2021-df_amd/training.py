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
    one small asynchronous copy."""

    CHUNK = 16384   # elements per workgroup

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        self._tables = {}
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

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
