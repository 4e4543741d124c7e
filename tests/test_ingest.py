"""Native audio ingest (libslsmi's WAV / FLAC decoder, pad, batch loader) against independent
CPU checks: Python's stdlib `wave` reader for PCM WAV, and for FLAC -- a lossless format -- the
exact integer samples that tests/flac_writer.py (an encoder written separately from the format
description) was given.  No FLAC file or decoder ships with the reference (it calls librosa),
so FLAC parity with libsndfile is unpinned; the round trip pins the decoder to the format.
pad() is checked against the oracle's restatement of data_utils_SSL.py:58-65."""
import os
import struct
import wave

import numpy as np
import pytest
import torch

import slsmi
from slsmi import _lib, data
from oracle import reference_cpu as orc
from flac_writer import encode_flac


def speechlike(n, seed=0, amp=6000):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = amp * (np.sin(2 * np.pi * 140 * t / 16000) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t / 16000))
               + 0.3 * np.sin(2 * np.pi * 1900 * t / 16000)) + rng.normal(0, 200, n)
    return np.clip(np.round(x), -32768, 32767).astype(np.int64)


def write_pcm_wav(path, x, rate=16000, width=2, extra_chunk=False, extensible=False):
    """x int array [frames] or [frames, ch]; width bytes per sample (1 = unsigned 8-bit)."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    ch = x.shape[1]
    if width == 1:
        body = (x + 128).astype(np.uint8).tobytes()
    elif width == 2:
        body = x.astype("<i2").tobytes()
    elif width == 3:
        b = x.astype("<i4").tobytes()
        body = b"".join(b[i:i + 3] for i in range(0, len(b), 4))
    else:
        body = x.astype("<i4").tobytes()
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, ch, rate, rate * ch * width, ch * width, width * 8, 22,
                          width * 8, 0, 1, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    else:
        fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * ch * width, ch * width, width * 8)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"  # odd size: one pad byte
    chunks += b"data" + struct.pack("<I", len(body)) + body
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def test_wav_pcm16_against_stdlib_reader(tmp_path):
    x = speechlike(30000, 1)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(x.astype("<i2").tobytes())
    with wave.open(p, "rb") as w:
        ref = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    y, sr = data.load_audio(p)
    assert sr == 16000 and data.audio_info(p) == (16000, 1, 30000)
    np.testing.assert_array_equal(y, ref)


@pytest.mark.parametrize("width,scale", [(1, 128.0), (2, 32768.0), (3, 8388608.0), (4, 2147483648.0)])
def test_wav_widths_stereo_and_chunks(tmp_path, width, scale):
    rng = np.random.default_rng(width)
    hi = 2 ** (8 * width - 1)
    x = rng.integers(-hi, hi, size=(5000, 2))
    p = str(tmp_path / "b.wav")
    write_pcm_wav(p, x, width=width, extra_chunk=True, extensible=(width >= 3))
    y, _ = data.load_audio(p)
    ref = ((x[:, 0] / scale).astype(np.float32) + (x[:, 1] / scale).astype(np.float32)) * np.float32(0.5)
    np.testing.assert_allclose(y, ref, atol=1e-7)
    assert data.audio_info(p) == (16000, 2, 5000)


def test_wav_float32(tmp_path):
    x = np.random.default_rng(3).uniform(-1, 1, 4000).astype("<f4")
    fmt = struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32)
    body = x.tobytes()
    blob = b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(body)) + body
    p = str(tmp_path / "f.wav")
    open(p, "wb").write(b"RIFF" + struct.pack("<I", 4 + len(blob)) + b"WAVE" + blob)
    np.testing.assert_array_equal(data.load_audio(p)[0], x)


FLAC_CASES = [
    dict(kinds=("fixed2",), blocksize=4096, porder=3),
    dict(kinds=("verbatim",), blocksize=1152),
    dict(kinds=("fixed0", "fixed1", "fixed3", "fixed4"), blocksize=192, porder=1),      # 2-byte frame numbers
    dict(kinds=("lpc",), blocksize=4096, lpc_order=8, lpc_precision=13, lpc_shift=10, porder=4),
    dict(kinds=("lpc", "fixed2"), blocksize=1000, lpc_order=12, rice_method=1),             # 16-bit size field
    dict(kinds=("fixed1",), blocksize=200, escape=True, porder=2),                          # 8-bit size field
    dict(kinds=("fixed2",), blocksize=2048, padding=300),
]


@pytest.mark.parametrize("case", range(len(FLAC_CASES)))
def test_flac_round_trip_mono(tmp_path, case):
    kw = dict(FLAC_CASES[case])
    x = speechlike(23456, seed=case)
    p = str(tmp_path / "m.flac")
    open(p, "wb").write(encode_flac(x, **kw))
    y, sr = data.load_audio(p)
    assert sr == 16000
    np.testing.assert_array_equal(y, (x / 32768.0).astype(np.float32))


@pytest.mark.parametrize("stereo", ["independent", "left_side", "side_right", "mid_side"])
def test_flac_stereo_modes(tmp_path, stereo):
    a, b = speechlike(9000, 5), speechlike(9000, 6, amp=3000)
    x = np.stack([a, (0.7 * a).astype(np.int64) + b // 3], axis=1)
    p = str(tmp_path / "s.flac")
    open(p, "wb").write(encode_flac(x, kinds=("fixed2", "lpc"), stereo=stereo, blocksize=1152, porder=2))
    y, _ = data.load_audio(p)
    ref = ((x[:, 0] / 32768.0).astype(np.float32) + (x[:, 1] / 32768.0).astype(np.float32)) * np.float32(0.5)
    np.testing.assert_allclose(y, ref, atol=1e-7)
    assert data.audio_info(p)[1] == 2


@pytest.mark.parametrize("bps", [8, 24])
def test_flac_other_sample_sizes_constant_and_wasted_bits(tmp_path, bps):
    rng = np.random.default_rng(bps)
    hi = 2 ** (bps - 1)
    x = rng.integers(-hi // 4, hi // 4, size=6000) & ~3          # two wasted bits
    x[2048:4096] = 12                                           # a constant block
    p = str(tmp_path / "w.flac")
    open(p, "wb").write(encode_flac(x, bps=bps, kinds=("fixed1", "constant", "fixed2"), blocksize=2048, wasted=True))
    y, _ = data.load_audio(p)
    np.testing.assert_array_equal(y, (x / float(hi)).astype(np.float32))


def test_flac_corruption_is_reported(tmp_path):
    x = speechlike(8192, 9)
    blob = bytearray(encode_flac(x, kinds=("fixed2",), blocksize=4096))
    good = str(tmp_path / "g.flac")
    open(good, "wb").write(bytes(blob))
    assert data.load_audio(good)[0].shape == (8192,)
    blob[len(blob) // 2] ^= 0x10                                # one flipped bit in a frame body
    bad = str(tmp_path / "bad.flac")
    open(bad, "wb").write(bytes(blob))
    with pytest.raises(_lib.SlsmiError, match="AUDIO_DATA"):
        data.load_audio(bad)
    trunc = str(tmp_path / "t.flac")
    open(trunc, "wb").write(bytes(encode_flac(x, kinds=("fixed2",), blocksize=4096))[:3000])
    with pytest.raises(_lib.SlsmiError, match="AUDIO_DATA"):
        data.load_audio(trunc)
    with pytest.raises(_lib.SlsmiError, match="AUDIO_OPEN"):
        data.load_audio(str(tmp_path / "missing.flac"))
    junk = str(tmp_path / "junk.flac")
    open(junk, "wb").write(b"not audio at all")
    with pytest.raises(_lib.SlsmiError, match="AUDIO_FORMAT"):
        data.load_audio(junk)


def test_mutated_files_are_decoded_or_refused(tmp_path):
    """Seeded damage to valid files (byte flips, header damage, truncation, bit flips inside frames): every file
    either decodes or is refused with one of the audio status codes -- no crash, no other exception.  (The same
    mutations run under AddressSanitizer and UBSan through tools/fuzz_ingest/run.sh.)"""
    rng = np.random.default_rng(7)
    seeds = [bytes(encode_flac(speechlike(6000, 1), kinds=("fixed2", "lpc"), blocksize=1152, porder=2)),
             bytes(encode_flac(np.stack([speechlike(3000, 2), speechlike(3000, 3)], 1), kinds=("fixed2", "lpc"),
                               stereo="mid_side", blocksize=1152)),
             bytes(encode_flac(speechlike(4000, 4), kinds=("fixed1", "constant", "verbatim"), blocksize=2048, wasted=True))]
    wav = str(tmp_path / "seed.wav")
    write_pcm_wav(wav, speechlike(4000, 5))
    seeds.append(open(wav, "rb").read())
    decoded = refused = 0
    for si, seed in enumerate(seeds):
        for k in range(120):
            blob = bytearray(seed)
            if k % 4 == 0:
                for _ in range(rng.integers(1, 6)):
                    blob[rng.integers(0, len(blob))] = rng.integers(0, 256)
            elif k % 4 == 1:
                for _ in range(rng.integers(1, 4)):
                    blob[rng.integers(0, 64)] = rng.integers(0, 256)
            elif k % 4 == 2:
                blob = blob[: rng.integers(0, len(blob))]
            else:
                for _ in range(rng.integers(1, 10)):
                    blob[rng.integers(42, len(blob))] ^= 1 << int(rng.integers(0, 8))
            path = str(tmp_path / ("m.wav" if si == 3 else "m.flac"))
            open(path, "wb").write(bytes(blob))
            try:
                x, _ = data.load_audio(path)
                assert np.isfinite(x).all()
                decoded += 1
            except _lib.SlsmiError as e:
                assert "AUDIO_" in str(e), e
                refused += 1
    assert decoded + refused == 480 and refused > 100 and decoded > 0


def test_batch_loading_pad_and_status(tmp_path):
    lens = [70000, 64600, 30000, 7, 64599]
    paths = []
    for i, n in enumerate(lens):
        p = str(tmp_path / f"u{i}.flac")
        open(p, "wb").write(encode_flac(speechlike(n, 20 + i), kinds=("fixed2", "lpc"), blocksize=4096))
        paths.append(p)
    out = torch.zeros(len(lens) + 1, 64600)
    data.load_batch_into(paths, out, threads=3)
    for i, n in enumerate(lens):
        x = (speechlike(n, 20 + i) / 32768.0).astype(np.float32)
        np.testing.assert_array_equal(out[i].numpy(), orc.pad_audio(x, 64600))
    assert torch.count_nonzero(out[len(lens)]) == 0
    # one unreadable file and one at another rate: reported, the others still decoded
    p8k = str(tmp_path / "r8k.flac")
    open(p8k, "wb").write(encode_flac(speechlike(5000, 1), sample_rate=8000))
    with pytest.raises(_lib.SlsmiError, match="2 of 4 files"):   # expect_rate > 0: another rate is an error
        data.load_batch_into([paths[0], str(tmp_path / "nope.flac"), p8k, paths[2]], out, threads=2, expect_rate=16000)
    with pytest.raises(_lib.SlsmiError, match="1 of 4 files"):   # default: converted to 16 kHz like librosa.load
        data.load_batch_into([paths[0], str(tmp_path / "nope.flac"), p8k, paths[2]], out, threads=2)
    x8, r8 = data.load_audio(p8k, sr=None)
    assert r8 == 8000 and len(x8) == 5000
    x16, r16 = data.load_audio(p8k)
    assert r16 == 16000 and len(x16) == 10000
    np.testing.assert_array_equal(out[2].numpy(), orc.pad_audio(x16, 64600))   # the batch path converts the same way


@pytest.mark.parametrize("src,dst", [(48000, 16000), (44100, 16000), (8000, 16000), (22050, 16000), (16000, 8000),
                                     (16001, 16000)])
def test_sample_rate_conversion_of_tones(src, dst):
    """librosa.load(sr=16000) (data_utils_SSL.py:112) resamples; so does load_audio.  A sum of tones below both
    Nyquist frequencies must come out as the same tones sampled at the new rate (band-limited interpolation is
    exact for them up to the filter's ripple), a tone between the new and the old Nyquist frequency must vanish,
    and the length is ceil(n * dst / src)."""
    n = src // 2
    t = np.arange(n) / src
    lo = 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 1234.5 * t + 0.7) + 0.2 * np.cos(2 * np.pi * 3100.0 * t)
    y = data.resample(lo.astype(np.float32), src, dst)
    m = int(np.ceil(n * dst / src))
    assert y.dtype == np.float32 and len(y) == m
    u = np.arange(m) / dst
    want = 0.4 * np.sin(2 * np.pi * 440.0 * u) + 0.3 * np.sin(2 * np.pi * 1234.5 * u + 0.7) + 0.2 * np.cos(2 * np.pi * 3100.0 * u)
    edge = int(0.02 * dst)    # the clip's ends see half a filter
    assert np.abs(y[edge:-edge] - want[edge:-edge]).max() < 2e-3
    if dst < src:
        f_alias = 0.5 * dst + 0.2 * (0.5 * src - 0.5 * dst)       # above the new Nyquist frequency, below the old one
        hi = np.sin(2 * np.pi * f_alias * t).astype(np.float32)
        z = data.resample(hi, src, dst)
        assert np.abs(z[edge:-edge]).max() < 2e-2
    np.testing.assert_array_equal(data.resample(lo.astype(np.float32), src, src), lo.astype(np.float32))


def test_datasets_and_batch_loader_order(tmp_path):
    os.makedirs(tmp_path / "flac")
    ids = [f"DF_E_{i:07d}" for i in range(11)]
    for i, u in enumerate(ids):
        open(tmp_path / "flac" / (u + ".flac"), "wb").write(
            encode_flac(speechlike(3000 + 997 * i, 40 + i), kinds=("fixed2",), blocksize=1152))
    proto = tmp_path / "trial_metadata.txt"
    proto.write_text("".join(u + "\n" for u in ids))
    assert data.genSpoof_list(str(proto), is_train=False, is_eval=True) == ids
    ds = data.Dataset_ASVspoof2021_eval(ids, str(tmp_path))
    x0, u0 = ds[3]
    assert u0 == ids[3] and x0.shape == (64600,) and x0.dtype == torch.float32
    got_ids, rows = [], []
    for batch, bid in data.BatchLoader(ds, batch_size=4, device="cpu", threads=2):
        assert batch.shape == (len(bid), 64600)
        got_ids += bid
        rows.append(batch)
    assert got_ids == ids
    allrows = torch.cat(rows)
    for i in (0, 3, 10):
        assert torch.equal(allrows[i], ds[i][0])
    shard = list(data.BatchLoader(ds, batch_size=4, device="cpu", indices=range(1, 11, 2)))
    assert [u for _, b in shard for u in b] == ids[1::2]
    wild = data.Dataset_in_the_wild_eval(["flac/" + ids[2] + ".flac"], str(tmp_path) + "/")
    assert torch.equal(wild[0][0], ds[2][0])
    tr = tmp_path / "train.txt"
    tr.write_text("LA_0079 LA_T_1 - - bonafide\nLA_0080 LA_T_2 - A01 spoof\n")
    assert data.genSpoof_list(str(tr), is_train=True) == ({"LA_T_1": 1, "LA_T_2": 0}, ["LA_T_1", "LA_T_2"])


def test_bucket_batches_cover_and_bound_padding():
    rng = np.random.default_rng(1234)
    lengths = rng.integers(16000, 160001, size=1000)
    lengths[:3] = [400, 200000, 160000]                     # shortest allowed, one to be cut, the maximum
    batches = data.bucket_batches(lengths, batch_size=64, max_len=160000)
    seen = sorted(i for idx, _ in batches for i in idx)
    assert seen == list(range(1000))
    clipped = np.minimum(lengths, 160000)
    padded_total = real_total = 0
    for idx, padded in batches:
        assert 1 <= len(idx) <= 64 and padded % 3200 == 0 and padded <= 160000
        assert padded >= clipped[idx].max() and padded - clipped[idx].max() < 3200
        padded_total += padded * len(idx)
        real_total += int(clipped[idx].sum())
    assert 1.0 - real_total / padded_total < 0.1            # uniform lengths: a few percent of padding
    assert len({p for _, p in batches}) <= 50
    with pytest.raises(ValueError):
        data.bucket_batches([399, 16000])
    assert data.bucket_batches([], 8) == []
    # two ranks see disjoint halves of the batches
    a, b = batches[0::2], batches[1::2]
    assert len(a) + len(b) == len(batches)
    # a frame budget: every batch's clips x frames of the padded length stays inside it, short clips make larger batches
    budget = 12896
    framed = data.bucket_batches(lengths, batch_size=256, max_len=160000, max_frames=budget)
    assert sorted(i for idx, _ in framed for i in idx) == list(range(1000))
    rows = [len(idx) * ((padded - 400) // 320 + 1) for idx, padded in framed]
    assert max(rows) <= budget and sorted(rows)[len(rows) // 2] > 0.9 * budget
    assert max(len(idx) for idx, _ in framed) > 64 > min(len(idx) for idx, _ in framed)


def test_variable_length_loader_rows_and_lengths(tmp_path):
    os.makedirs(tmp_path / "flac")
    ids = [f"W_{i:03d}" for i in range(9)]
    lens = [16000, 52000, 9000, 33333, 160000, 170000, 21000, 20000, 64600]
    for i, u in enumerate(ids):
        open(tmp_path / "flac" / (u + ".flac"), "wb").write(
            encode_flac(speechlike(lens[i], 60 + i), kinds=("fixed2",), blocksize=4096))
    ds = data.Dataset_ASVspoof2021_eval(ids, str(tmp_path))
    ld = data.VariableLengthLoader(ds, batch_size=4, device="cpu", threads=2)
    assert list(ld.lengths) == [min(n, 160000) for n in lens]
    got = {}
    for batch, lengths, bid, idx in ld:
        assert batch.shape[0] == len(bid) == len(idx) and lengths.dtype == torch.int32
        for r, (u, j) in enumerate(zip(bid, idx)):
            assert u == ids[j] and int(lengths[r]) == min(lens[j], 160000) <= batch.shape[1]
            got[j] = batch[r, :int(lengths[r])].clone()
    assert sorted(got) == list(range(9))
    for j in (0, 4, 5):
        x = (speechlike(lens[j], 60 + j) / 32768.0).astype(np.float32)[:160000]
        np.testing.assert_array_equal(got[j].numpy(), x)
    halves = [data.VariableLengthLoader(ds, batch_size=4, device="cpu", threads=2, rank=r, world=2,
                                        lengths=ld.lengths) for r in range(2)]
    assert sorted(j for h in halves for _, _, _, idx in h for j in idx) == list(range(9))


def test_variable_length_loader_converts_lengths_of_files_at_another_rate(tmp_path):
    """A file that is not at 16 kHz is converted on load (as librosa.load(sr=16000) does, data_utils_SSL.py:112): the
    length the loader hands to Model.forward(lengths=) must be that of the converted clip, ceil(frames * 16000 / rate),
    not the header's frame count -- otherwise tiled padding would be scored as signal (8 kHz) or signal cut (48 kHz)."""
    os.makedirs(tmp_path / "flac")
    ids = ["R_8k", "R_16k", "R_48k"]
    rates, frames = [8000, 16000, 48000], [9001, 20000, 50000]
    for u, r, n in zip(ids, rates, frames):
        open(tmp_path / "flac" / (u + ".flac"), "wb").write(encode_flac(speechlike(n, 5), sample_rate=r))
    ds = data.Dataset_ASVspoof2021_eval(ids, str(tmp_path))
    ld = data.VariableLengthLoader(ds, batch_size=4, device="cpu", threads=2)
    want = [-(-n * 16000 // r) for r, n in zip(rates, frames)]
    assert list(ld.lengths) == want == [18002, 20000, 16667]
    for batch, lengths, bid, idx in ld:
        for r, j in enumerate(idx):
            x16, rate = data.load_audio(ds.path_of(ids[j]))
            assert rate == 16000 and len(x16) == want[j] == int(lengths[r])
            np.testing.assert_array_equal(batch[r, :want[j]].numpy(), x16)


def _patch_total_samples(blob: bytes, total: int) -> bytes:
    """STREAMINFO's 36-bit total-samples field sits in bytes 13 (low nibble) .. 17 of the block body, which starts
    at file offset 8 (after 'fLaC' and the 4-byte block header)."""
    b = bytearray(blob)
    o = 8
    b[o + 13] = (b[o + 13] & 0xF0) | ((total >> 32) & 0x0F)
    for i in range(4):
        b[o + 14 + i] = (total >> (24 - 8 * i)) & 0xFF
    return bytes(b)


def test_audio_info_reads_headers_and_untrusted_lengths_are_refused(tmp_path):
    """slsmi_audio_info answers from the header (STREAMINFO / the data chunk's length) without decoding; a FLAC
    whose header says "length unknown" (0) is decoded to find out; a header that claims 2^36 - 1 samples (256 GB as
    floats) is refused instead of being allocated; a header that over-states the length a little decodes what is there."""
    x = speechlike(10000, 3)
    blob = encode_flac(x, kinds=("fixed2",), blocksize=4096)
    p = str(tmp_path / "a.flac")
    open(p, "wb").write(blob)
    assert data.audio_info(p) == (16000, 1, 10000)
    # a frame body corrupted: the header still answers, the decode reports it
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x10
    pb = str(tmp_path / "b.flac")
    open(pb, "wb").write(bytes(bad))
    assert data.audio_info(pb)[2] == 10000
    unknown = str(tmp_path / "u.flac")
    open(unknown, "wb").write(_patch_total_samples(blob, 0))
    assert data.audio_info(unknown)[2] == 10000
    assert data.load_audio(unknown)[0].shape == (10000,)
    huge = str(tmp_path / "h.flac")
    open(huge, "wb").write(_patch_total_samples(blob, (1 << 36) - 1))
    with pytest.raises(_lib.SlsmiError, match="AUDIO_DATA"):
        data.audio_info(huge)
    with pytest.raises(_lib.SlsmiError, match="AUDIO_DATA"):
        data.load_audio(huge)
    more = str(tmp_path / "m.flac")
    open(more, "wb").write(_patch_total_samples(blob, 200_000_000))   # 800 MB if it were reserved as claimed
    n = ctypes_read_frames(more)
    assert n == 10000


def ctypes_read_frames(path):
    import ctypes as C
    lib = _lib.load()
    buf = np.zeros(16, np.float32)
    frames, sr = C.c_int64(0), C.c_int(0)
    rc = lib.slsmi_audio_read(os.fsencode(path), buf.ctypes.data_as(C.c_void_p), 16, C.byref(frames), C.byref(sr))
    assert rc == 0
    return frames.value
