"""Score emission and utterance sharding around Model.forward.

Mirrors the reference's eval loop (main.py:158-199): score = exp(log-prob)[:, 1]
(P(bonafide); class 1 = bonafide, data_utils_SSL.py:38), one "utt score" line per
trial in protocol order.  Multi-GPU: one process per GPU, rank r scores trials
r, r+W, r+2W, ... and rank 0 gathers the fp32 scores (the only exchange step).
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

UTT_SAMPLES = 64600


def pad(x: np.ndarray, max_len: int = UTT_SAMPLES) -> np.ndarray:
    """Crop to max_len or tile-repeat then crop (reference data_utils_SSL.py:58-65)."""
    n = x.shape[0]
    if n >= max_len:
        return x[:max_len]
    return np.tile(x, int(max_len / n) + 1)[:max_len]


def scores_from_logp(logp: torch.Tensor, log_scores: bool = False) -> np.ndarray:
    """fp32 scores for a batch of log-probs: exp(logp)[:, 1] as the in-tree
    main.py:183-184 writes; log_scores=True gives the upstream form logp[:, 1]."""
    col = logp[:, 1] if log_scores else torch.exp(logp)[:, 1]
    return col.detach().float().cpu().numpy().ravel()


def format_score_lines(utt_ids: Sequence[str], scores: np.ndarray) -> List[str]:
    """'{utt} {score}\\n' with Python's repr of the float32 value widened to
    float (main.py:188-192)."""
    return ["{} {}\n".format(u, s) for u, s in zip(utt_ids, np.asarray(scores, dtype=np.float32).tolist())]


def write_score_file(path: str, utt_ids: Sequence[str], scores: np.ndarray, append: bool = False) -> None:
    with open(path, "a+" if append else "w") as fh:
        fh.writelines(format_score_lines(utt_ids, scores))


def genSpoof_list_eval(protocol_path: str) -> List[str]:
    """Eval protocol: one utterance id per line (data_utils_SSL.py:41-46)."""
    with open(protocol_path) as f:
        return [ln.strip() for ln in f.readlines()]


def produce_evaluation_file(dataset, model, device, save_path: str, quick_test: bool = False,
                            batch_size: int = 20, num_workers: int = 0, log_scores: bool = False) -> None:
    """Drop-in for main.py:158-199: iterates `dataset` ((wav, utt_id) items) in
    order, scores each batch with model(batch_x, return_sae_loss=False) and
    appends the lines to save_path."""
    from torch.utils.data import DataLoader
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, drop_last=False,
                        pin_memory=True, num_workers=num_workers)
    model.eval()
    for i, (batch_x, utt_id) in enumerate(loader):
        if quick_test and i >= 5:
            break
        batch_x = batch_x.to(device, non_blocking=True)
        with torch.no_grad():
            batch_out = model(batch_x, return_sae_loss=False)
        write_score_file(save_path, list(utt_id), scores_from_logp(batch_out, log_scores), append=True)
    print("Scores saved to {}".format(save_path))


# ----------------------------------------------------------------------------- sharding
def shard_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Trials owned by `rank`: rank, rank + world, ... (protocol order kept inside a shard)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def merge_shards(n_items: int, world: int, shard_scores: Sequence[np.ndarray]) -> np.ndarray:
    """Inverse of shard_indices: interleave per-rank score vectors back into protocol order."""
    out = np.empty(n_items, dtype=np.float32)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        if len(shard_scores[r]) != len(idx):
            raise ValueError(f"rank {r}: {len(shard_scores[r])} scores for {len(idx)} trials")
        out[idx] = shard_scores[r]
    return out


def gather_scores(local_scores: torch.Tensor, n_items: int, group=None) -> Optional[np.ndarray]:
    """All ranks pass their shard's scores (1-D fp32 tensor on the device the
    process group communicates on: a ROCm tensor for RCCL, a CPU tensor for
    gloo); rank 0 gets the scores in protocol order, other ranks None.  One
    all_gather of ceil(n/W) floats per rank -- the path's only collective."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_scores.detach().float().cpu().numpy()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_items + world - 1) // world
    buf = torch.zeros(per, dtype=torch.float32, device=local_scores.device)
    buf[: local_scores.numel()] = local_scores.float()
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    if rank != 0:
        return None
    shards = [bufs[r][: len(shard_indices(n_items, r, world))].cpu().numpy() for r in range(world)]
    return merge_shards(n_items, world, shards)


def score_sharded(model, utt_ids: Sequence[str], load_batch: Callable[[Sequence[int]], torch.Tensor],
                  device, batch_size: int = 64, save_path: Optional[str] = None, log_scores: bool = False,
                  comm_device: Optional[torch.device] = None) -> Optional[np.ndarray]:
    """Score a whole protocol list across the ranks of the default process
    group.  load_batch(indices) returns the [len(indices), L] fp32 batch on
    `device`.  Rank 0 returns (and optionally writes) the scores in protocol
    order."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = shard_indices(len(utt_ids), rank, world)
    chunks = []
    model.eval()
    with torch.no_grad():
        for s in range(0, len(mine), batch_size):
            idx = mine[s:s + batch_size]
            logp = model(load_batch(idx), return_sae_loss=False)
            chunks.append(logp[:, 1] if log_scores else torch.exp(logp)[:, 1])
    local = torch.cat(chunks) if chunks else torch.zeros(0, device=device)
    if comm_device is not None:
        local = local.to(comm_device)
    allscores = gather_scores(local, len(utt_ids))
    if rank == 0 and save_path is not None:
        if os.path.exists(save_path):
            os.remove(save_path)  # main.py:646-649 removes an existing file
        write_score_file(save_path, utt_ids, allscores)
    return allscores if rank == 0 else None


def gather_indexed_scores(local_idx: torch.Tensor, local_scores: torch.Tensor, n_items: int,
                          group=None) -> Optional[np.ndarray]:
    """Like gather_scores for shards that are not the r, r + W, ... pattern (length-bucketed batches
    dealt out to the ranks): every rank passes the protocol positions it scored (int64) and the
    scores; rank 0 gets all n_items scores in protocol order.  Two all_gathers of padded vectors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = np.full(n_items, np.nan, dtype=np.float32)
        out[local_idx.cpu().numpy()] = local_scores.detach().float().cpu().numpy()
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local_scores.device
    cnt = torch.tensor([local_scores.numel()], dtype=torch.int64, device=dev)
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    per = max(int(c.item()) for c in cnts)
    sbuf = torch.zeros(per, dtype=torch.float32, device=dev)
    ibuf = torch.full((per,), -1, dtype=torch.int64, device=dev)
    sbuf[: local_scores.numel()] = local_scores.float()
    ibuf[: local_idx.numel()] = local_idx.to(dev)
    sall = [torch.empty_like(sbuf) for _ in range(world)]
    iall = [torch.empty_like(ibuf) for _ in range(world)]
    dist.all_gather(sall, sbuf, group=group)
    dist.all_gather(iall, ibuf, group=group)
    if rank != 0:
        return None
    out = np.full(n_items, np.nan, dtype=np.float32)
    for r in range(world):
        k = int(cnts[r].item())
        out[iall[r][:k].cpu().numpy()] = sall[r][:k].cpu().numpy()
    if np.isnan(out).any():
        raise RuntimeError(f"{int(np.isnan(out).sum())} of {n_items} trials were scored by no rank")
    return out


def score_variable_length(model, loader, utt_ids: Sequence[str], save_path: Optional[str] = None,
                          log_scores: bool = False, comm_device: Optional[torch.device] = None) -> Optional[np.ndarray]:
    """Score clips of different lengths at their own lengths: `loader` (data.VariableLengthLoader, built
    with this process's rank and world size) yields length-bucketed padded batches, every batch goes
    through model(batch, return_sae_loss=False, lengths=lengths), and rank 0 returns (and optionally
    writes, in protocol order) the scores of all ranks."""
    idx_chunks, chunks = [], []
    model.eval()
    dev = None
    with torch.no_grad():
        for batch, lengths, _, idx in loader:
            logp = model(batch, return_sae_loss=False, lengths=lengths)
            chunks.append(logp[:, 1] if log_scores else torch.exp(logp)[:, 1])
            idx_chunks.append(torch.as_tensor(np.asarray(idx, dtype=np.int64)))
            dev = logp.device
    local = torch.cat(chunks) if chunks else torch.zeros(0, device=dev or "cpu")
    lidx = torch.cat(idx_chunks) if idx_chunks else torch.zeros(0, dtype=torch.int64)
    if comm_device is not None:
        local = local.to(comm_device)
    allscores = gather_indexed_scores(lidx, local, len(utt_ids))
    if allscores is not None and save_path is not None:
        if os.path.exists(save_path):
            os.remove(save_path)
        write_score_file(save_path, utt_ids, allscores)
    return allscores


class RangeTracker:
    """Over the batches of a scoring run: the largest |value| and the number of saturated values of every 16-bit store
    of the forward (Engine.range_report / slsmi_set_range_report).  A first run with a real checkpoint tells from it
    whether half precision's range (65504) holds or precision="bf16" is needed: a store whose count is not zero has
    clipped values."""

    def __init__(self, engine):
        self.engine = engine
        engine.range_report(True)
        self.rows = None

    def update(self) -> None:
        now = self.engine.last_ranges()
        if self.rows is None:
            self.rows = [list(r) for r in now]
        else:
            for acc, r in zip(self.rows, now):
                acc[1] = max(acc[1], r[1])
                acc[2] += r[2]

    def close(self) -> None:
        self.engine.range_report(False)

    def saturated(self):
        return [(n, c) for n, _, c in (self.rows or []) if c > 0]

    def lines(self, limit: float = 65504.0):
        out = [f"{'16-bit store':44s} {'max |value|':>12s} {'of range':>9s} {'saturated':>10s}"]
        for n, mx, c in self.rows or []:
            out.append(f"{n:44s} {mx:12.4g} {100.0 * mx / limit:8.2f}% {c:10d}")
        return out
