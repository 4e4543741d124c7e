"""Audio ingest: the reference's Dataset classes and protocol readers
(data_utils_SSL.py:13-135) on libslsmi's native WAV / FLAC decoder, plus a batch
loader that decodes whole batches on host threads into pinned memory and copies them to the
device ahead of the forward that consumes them.

The reference decodes one utterance per DataLoader worker call with librosa (libsndfile) and
pads it in numpy; here `slsmi_audio_load_batch` decodes, averages channels, crops / tiles to
64600 samples and writes a whole batch straight into the pinned buffer.  Files at another rate
than 16 kHz (ASVspoof and In-the-Wild have none) are converted first, as librosa.load(sr=16000) does.
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000
CUT = 64600  # ~4 s, data_utils_SSL.py:103


def genSpoof_list(dir_meta, is_train=False, is_eval=False):
    """Protocol reader of data_utils_SSL.py:26-53: eval lists hold one key per line; train /
    dev lists hold `speaker key - - label` and also give {key: 1 if bonafide else 0}."""
    with open(dir_meta, "r") as f:
        lines = f.readlines()
    if is_eval and not is_train:
        return [line.strip() for line in lines]
    d_meta, file_list = {}, []
    for line in lines:
        _, key, _, _, label = line.strip().split()
        file_list.append(key)
        d_meta[key] = 1 if label == "bonafide" else 0
    return d_meta, file_list


def genSpoof_list2019(dir_meta, is_train=False, is_eval=False):
    """data_utils_SSL.py:13-24: only the eval form returns a list."""
    if is_eval:
        with open(dir_meta, "r") as f:
            return [line.strip() for line in f.readlines()]
    return None


def audio_info(path: str) -> Tuple[int, int, int]:
    """(sample_rate, channels, frames) of a WAV or FLAC file."""
    sr, ch, n = C.c_int(0), C.c_int(0), C.c_int64(0)
    _lib.check(_lib.load().slsmi_audio_info(os.fsencode(path), C.byref(sr), C.byref(ch), C.byref(n)), None,
               f"slsmi_audio_info({path})")
    return sr.value, ch.value, n.value


def resample(x: np.ndarray, src_rate: int, dst_rate: int) -> np.ndarray:
    """Band-limited sample-rate conversion of a mono float32 clip (slsmi_audio_resample)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if src_rate == dst_rate:
        return x
    cap = int(np.ceil(len(x) * dst_rate / src_rate)) + 1
    out = np.empty(cap, dtype=np.float32)
    got = C.c_int64(0)
    _lib.check(_lib.load().slsmi_audio_resample(x.ctypes.data_as(C.c_void_p), len(x), int(src_rate), int(dst_rate),
                                                out.ctypes.data_as(C.c_void_p), cap, C.byref(got)), None,
               "slsmi_audio_resample")
    return out[:got.value]


def load_audio(path: str, sr: Optional[int] = SAMPLE_RATE) -> Tuple[np.ndarray, int]:
    """Mono float32 samples and their rate, as librosa.load(path, sr=sr) (data_utils_SSL.py:112): a file at
    another rate is converted to `sr` (band-limited interpolation; librosa's own resampler is a third-party
    library outside the reference tree, so the converted samples agree with it closely, not bit for bit);
    sr=None keeps the file's rate."""
    rate, _, frames = audio_info(path)
    out = np.empty(frames, dtype=np.float32)
    got, r = C.c_int64(0), C.c_int(0)
    _lib.check(_lib.load().slsmi_audio_read(os.fsencode(path), out.ctypes.data_as(C.c_void_p), frames,
                                            C.byref(got), C.byref(r)), None, f"slsmi_audio_read({path})")
    x = out[:got.value]
    if sr is not None and r.value != sr:
        return resample(x, r.value, sr), sr
    return x, r.value


def load_batch_into(paths: Sequence[str], out: torch.Tensor, cut: int = CUT, threads: int = 8,
                    expect_rate: int = -SAMPLE_RATE) -> None:
    """Decode `paths` into the first len(paths) rows of the host tensor `out` [>= n, cut].
    expect_rate < 0 (default -16000): files at another rate are converted to -expect_rate, as librosa.load(sr=16000)
    does in the reference's datasets; > 0: such a file is an error; 0: samples are taken as they are."""
    n = len(paths)
    if out.device.type != "cpu" or out.dtype != torch.float32 or not out.is_contiguous() or out.shape[1] != cut \
            or out.shape[0] < n:
        raise ValueError("out must be a contiguous float32 host tensor [>= len(paths), cut]")
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    status = (C.c_int * n)()
    rc = _lib.load().slsmi_audio_load_batch(arr, n, cut, expect_rate, C.c_void_p(out.data_ptr()), threads, status)
    if rc != 0:
        bad = [(paths[i], _lib.STATUS.get(status[i], status[i])) for i in range(n) if status[i] != 0]
        raise _lib.SlsmiError(f"{len(bad)} of {n} files failed to decode, first: {bad[:3]}")


class _EvalDataset(torch.utils.data.Dataset):
    cut = CUT

    def __init__(self, list_IDs, base_dir):
        self.list_IDs = list_IDs
        self.base_dir = base_dir

    def __len__(self):
        return len(self.list_IDs)

    def path_of(self, utt_id: str) -> str:
        raise NotImplementedError

    def __getitem__(self, index):
        from .scoring import pad
        utt_id = self.list_IDs[index]
        x, _ = load_audio(self.path_of(utt_id), SAMPLE_RATE)
        return torch.from_numpy(np.ascontiguousarray(pad(x, self.cut))), utt_id


class Dataset_ASVspoof2021_eval(_EvalDataset):
    """data_utils_SSL.py:96-115: base_dir/flac/<utt_id>.flac -> (Tensor[64600], utt_id)."""

    def path_of(self, utt_id):
        return self.base_dir + "/flac/" + utt_id + ".flac"


class Dataset_in_the_wild_eval(_EvalDataset):
    """data_utils_SSL.py:118-135: base_dir + <utt_id> (the id carries the file extension)."""

    def path_of(self, utt_id):
        return self.base_dir + utt_id


class Dataset_ASVspoof2019_train(torch.utils.data.Dataset):
    """data_utils_SSL.py:68-93: base_dir + 'flac/<utt_id>.flac' -> RawBoost algorithm `algo` with the
    parameters in `args` (main.py:443-459) -> crop / tile to 64600 samples -> (Tensor[64600], label).
    `labels` maps utt_id to 1 (bonafide) / 0 (spoof), as genSpoof_list(is_train=True) returns it.
    The augmentation draws from numpy's global generator, as the reference's does: seed it per
    DataLoader worker (`worker_init_fn`) when runs must be repeatable."""

    def __init__(self, args, list_IDs, labels, base_dir, algo):
        self.list_IDs = list_IDs
        self.labels = labels
        self.base_dir = base_dir
        self.algo = algo
        self.args = args
        self.cut = CUT

    def __len__(self):
        return len(self.list_IDs)

    def __getitem__(self, index):
        from .rawboost import process_Rawboost_feature
        from .scoring import pad
        utt_id = self.list_IDs[index]
        x, fs = load_audio(self.base_dir + "flac/" + utt_id + ".flac", SAMPLE_RATE)
        y = process_Rawboost_feature(x, fs, self.args, self.algo)
        x_inp = torch.from_numpy(np.ascontiguousarray(pad(np.asarray(y), self.cut), dtype=np.float32))
        return x_inp, self.labels[utt_id]


def _frames_of(samples: int) -> int:
    """Frames the conv feature extractor makes of `samples` samples (receptive field 400, hop 320)."""
    return (samples - 400) // 320 + 1


def bucket_batches(lengths: Sequence[int], batch_size: int = 64, max_len: int = 160000, step: int = 3200,
                   max_waste: float = 0.25, min_len: int = 400,
                   max_frames: Optional[int] = None) -> List[Tuple[List[int], int]]:
    """Length-bucketed dynamic batching for clips of different lengths (nothing like it exists in the
    reference, whose datasets cut or tile every clip to 64600 samples, data_utils_SSL.py:103, :125).

    lengths[i]: samples of clip i (clips longer than max_len are cut to it, shorter than `min_len`, the
    conv stack's receptive field, are rejected).  Returns [(indices, padded_len)]: clips sorted by
    length, longest first, and cut into batches of at most `batch_size` clips; a batch is closed early
    when its shortest clip would make the padded samples exceed `max_waste` of the batch.  padded_len
    is the batch's longest clip rounded up to a multiple of `step` samples (10 frames), so that the
    set of batch shapes a long run visits stays small.  Every index appears in exactly one batch.
    max_frames: also close a batch before its clips x frames of the padded length (the rows of every encoder GEMM)
    would exceed this budget -- with batch_size as the cap on clips, batches of short clips grow and batches of long
    clips shrink to about the same number of rows.  12896 = 62 x 208 rows is one full round of the encoder GEMM
    kernel's 208-row tiles on 256 CUs, the shape the fixed-length benchmark batch (64 x 201) has."""
    n = np.minimum(np.asarray(lengths, dtype=np.int64), max_len)
    if n.size and int(n.min()) < min_len:
        raise ValueError(f"clip {int(n.argmin())} has {int(n.min())} samples: shorter than {min_len}")
    order = np.argsort(-n, kind="stable")
    batches: List[Tuple[List[int], int]] = []
    cur: List[int] = []
    padded = real = 0
    for i in order:
        li = int(n[i])
        if cur:
            waste = 1.0 - (real + li) / float(padded * (len(cur) + 1))
            over = max_frames is not None and (len(cur) + 1) * _frames_of(padded) > max_frames
            if len(cur) >= batch_size or waste > max_waste or over:
                batches.append((cur, padded))
                cur = []
        if not cur:
            padded = min((li + step - 1) // step * step, max((max_len + step - 1) // step * step, li))
            real = 0
        cur.append(int(i))
        real += li
    if cur:
        batches.append((cur, padded))
    return batches


class VariableLengthLoader:
    """Iterates (batch [b, padded_len] fp32 on `device`, lengths [b] int32 on `device`, utt_ids) over
    the length-bucketed batches of a dataset (or of this rank's share of them): for
    Model.forward(batch, lengths=lengths).  Row i holds clip i's samples followed by padding that the
    variable-length forward never reads into a result (the decoder fills it by repeating the clip)."""

    def __init__(self, dataset: "_EvalDataset", batch_size: int = 64, device="cuda", threads: int = 16,
                 max_len: int = 160000, rank: int = 0, world: int = 1, lengths: Optional[Sequence[int]] = None,
                 max_frames: Optional[int] = None):
        self.ds, self.threads, self.device = dataset, threads, torch.device(device)
        self.max_len = max_len
        if lengths is None:  # header reads only; a file at another rate is converted on load (load_batch_into):
            # its length is what the conversion writes, ceil(frames * 16000 / rate) (slsmi_audio_resample)
            lengths = []
            for u in dataset.list_IDs:
                rate, _, frames = audio_info(dataset.path_of(u))
                lengths.append(frames if rate == SAMPLE_RATE else -((-frames * SAMPLE_RATE) // rate))
        self.lengths = np.minimum(np.asarray(lengths, dtype=np.int64), max_len)
        self.batches = bucket_batches(self.lengths, batch_size, max_len, max_frames=max_frames)[rank::world]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        pinned = self.device.type == "cuda"
        for idx, padded in self.batches:
            ids = [self.ds.list_IDs[j] for j in idx]
            host = torch.empty(len(idx), padded, dtype=torch.float32, pin_memory=pinned)
            load_batch_into([self.ds.path_of(u) for u in ids], host, padded, self.threads)
            lens = torch.from_numpy(self.lengths[idx].astype(np.int32))
            yield host.to(self.device, non_blocking=pinned), lens.to(self.device), ids, idx


class BatchLoader:
    """Iterates (batch [b, cut] fp32 on `device`, utt_ids) over a dataset in protocol order (or
    over this rank's shard of it).  A host thread decodes batch k+1 into one of `depth` pinned
    buffers while the device works on batch k; the copy to the device runs on its own stream and
    the consumer's stream waits for it."""

    def __init__(self, dataset: _EvalDataset, batch_size: int = 64, device="cuda", threads: int = 16,
                 indices: Optional[Sequence[int]] = None, depth: int = 3):
        self.ds, self.batch_size, self.threads, self.depth = dataset, batch_size, threads, max(2, depth)
        self.device = torch.device(device)
        self.indices = list(range(len(dataset))) if indices is None else list(indices)

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, List[str]]]:
        cut = self.ds.cut
        pinned = self.device.type == "cuda"
        bufs = [torch.empty(self.batch_size, cut, dtype=torch.float32, pin_memory=pinned) for _ in range(self.depth)]
        free: "queue.Queue[int]" = queue.Queue()
        for i in range(self.depth):
            free.put(i)
        ready: "queue.Queue" = queue.Queue()

        def producer():
            try:
                for s in range(0, len(self.indices), self.batch_size):
                    ids = [self.ds.list_IDs[j] for j in self.indices[s:s + self.batch_size]]
                    slot = free.get()
                    load_batch_into([self.ds.path_of(u) for u in ids], bufs[slot], cut, self.threads)
                    ready.put((slot, ids))
                ready.put(None)
            except BaseException as e:  # surfaces in the consumer
                ready.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        copy_stream = torch.cuda.Stream(self.device) if pinned else None
        while True:
            item = ready.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            slot, ids = item
            host = bufs[slot][:len(ids)]
            if pinned:
                with torch.cuda.stream(copy_stream):
                    dev = host.to(self.device, non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy_stream)
                torch.cuda.current_stream(self.device).wait_event(done)
                dev.record_stream(torch.cuda.current_stream(self.device))
                done.synchronize()  # the pinned buffer may be refilled once the copy has left it
            else:
                dev = host.clone()
            free.put(slot)
            yield dev, ids
        th.join()
