"""A small FLAC encoder for the ingest tests (test infrastructure, not shipped).

Writes valid FLAC streams (RFC 9639) that exercise every decoder path: CONSTANT, VERBATIM,
FIXED orders 0-4 and LPC subframes, Rice partitions with 4- and 5-bit parameters and escape
partitions, wasted bits, independent / left-side / side-right / mid-side stereo, 8-bit and
16-bit block-size fields, CRC-8 and CRC-16.  It is written from the format description and
shares no code with the decoder, so decode(encode(x)) == x is an independent check.
"""
import struct

import numpy as np


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def write(self, value, bits):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.write(0, 32)
            q -= 32
        self.write(1, q + 1)

    def align(self):
        if self.n:
            self.write(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.buf)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out = []
    n = 1
    while v >= (1 << (6 * n + (6 - n))):
        n += 1
    for i in range(n):
        out.append(0x80 | (v & 0x3F))
        v >>= 6
    lead = ((0xFF << (7 - n)) & 0xFF) | v
    return bytes([lead] + out[::-1])


def rice_bits(res, k):
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.int64)
    return int(((u >> k) + 1 + k).sum())


def write_residual(bw, res, blocksize, order, porder, method=0, force_escape=False):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    bw.write(method, 2)
    bw.write(porder, 4)
    parts = 1 << porder
    idx = 0
    for pt in range(parts):
        count = (blocksize >> porder) - (order if pt == 0 else 0)
        r = res[idx:idx + count].astype(np.int64)
        idx += count
        if force_escape and pt == parts - 1:
            raw = int(max(1, max((int(abs(int(v))).bit_length() + 1) for v in r))) if count else 0
            bw.write(esc, pbits)
            bw.write(raw, 5)
            for v in r:
                bw.write(int(v), raw)
            continue
        best_k = min(range(0, esc), key=lambda k: rice_bits(r, k)) if count else 0
        bw.write(best_k, pbits)
        for v in r:
            v = int(v)
            u = 2 * v if v >= 0 else -2 * v - 1
            bw.unary(u >> best_k)
            bw.write(u & ((1 << best_k) - 1), best_k)
    assert idx == len(res)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def predict_residual(x, coefs, shift):
    order = len(coefs)
    x = x.astype(np.int64)
    n = len(x)
    s = np.zeros(n - order, dtype=np.int64)
    for j, c in enumerate(coefs):
        s += c * x[order - 1 - j:n - 1 - j]
    return x[order:] - (s >> shift)


def write_subframe(bw, x, bps, kind, rng, opts):
    """x: int64 samples of one channel; kind in constant|verbatim|fixedN|lpc."""
    x = np.asarray(x, dtype=np.int64)
    n = len(x)
    wasted = 0
    if opts.get("wasted") and np.any(x != 0):
        while np.all((x >> wasted) & 1 == 0) and wasted < bps - 1:
            wasted += 1
    xs = x >> wasted
    eb = bps - wasted
    bw.write(0, 1)
    if kind == "constant":
        assert np.all(xs == xs[0])
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        code = 8 + order
    else:
        order = opts.get("lpc_order", 6)
        code = 31 + order
    bw.write(code, 6)
    if wasted:
        bw.write(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.write(0, 1)
    if kind == "constant":
        bw.write(int(xs[0]), eb)
        return
    if kind == "verbatim":
        for v in xs:
            bw.write(int(v), eb)
        return
    porder = opts.get("porder", 0)
    while porder > 0 and ((n >> porder) << porder != n or (n >> porder) <= order):
        porder -= 1
    if kind.startswith("fixed"):
        for v in xs[:order]:
            bw.write(int(v), eb)
        res = predict_residual(xs, FIXED[order], 0)
    else:
        prec, shift = opts.get("lpc_precision", 12), opts.get("lpc_shift", 9)
        # a quantised predictor fitted by least squares on this block
        rows = np.stack([xs[order - 1 - j:n - 1 - j] for j in range(order)], axis=1).astype(np.float64)
        sol, *_ = np.linalg.lstsq(rows, xs[order:].astype(np.float64), rcond=None)
        lim = (1 << (prec - 1)) - 1
        coefs = [int(np.clip(round(c * (1 << shift)), -lim - 1, lim)) for c in sol]
        for v in xs[:order]:
            bw.write(int(v), eb)
        bw.write(prec - 1, 4)
        bw.write(shift, 5)
        for c in coefs:
            bw.write(c, prec)
        res = predict_residual(xs, coefs, shift)
    write_residual(bw, res, n, order, porder, method=opts.get("rice_method", 0),
                   force_escape=opts.get("escape", False))


def encode_flac(samples, sample_rate=16000, bps=16, blocksize=4096, kinds=("fixed2",), stereo="independent",
                seed=0, **opts):
    """samples: int array [frames] or [frames, channels].  `kinds` cycles over the frames' subframes."""
    x = np.asarray(samples)
    if x.ndim == 1:
        x = x[:, None]
    frames, ch = x.shape
    rng = np.random.default_rng(seed)
    out = bytearray(b"fLaC")
    info = BitWriter()
    info.write(blocksize, 16)
    info.write(blocksize, 16)
    info.write(0, 24)
    info.write(0, 24)
    info.write(sample_rate, 20)
    info.write(ch - 1, 3)
    info.write(bps - 1, 5)
    info.write(frames, 36)
    for _ in range(16):
        info.write(0, 8)  # MD5 not set
    body = info.bytes()
    pad_len = opts.get("padding", 0)
    out += bytes([0x00 if pad_len else 0x80]) + struct.pack(">I", len(body))[1:] + body
    if pad_len:
        out += bytes([0x81]) + struct.pack(">I", pad_len)[1:] + bytes(pad_len)
    ki = 0
    fno = 0
    for start in range(0, frames, blocksize):
        blk = x[start:start + blocksize].astype(np.int64)
        n = len(blk)
        hdr = BitWriter()
        hdr.write(0x3FFE, 14)
        hdr.write(0, 1)
        hdr.write(0, 1)  # fixed block size: frame number coded
        if n == blocksize and blocksize in (192, 576, 1152, 2304, 4608, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
            code = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5}.get(blocksize) or (8 + int(np.log2(blocksize // 256)))
            tail = None
        elif n <= 256:
            code, tail = 6, (n - 1, 8)
        else:
            code, tail = 7, (n - 1, 16)
        hdr.write(code, 4)
        hdr.write(0, 4)  # sample rate from STREAMINFO
        if ch == 2 and stereo != "independent":
            ch_code = {"left_side": 8, "side_right": 9, "mid_side": 10}[stereo]
        else:
            ch_code = ch - 1
        hdr.write(ch_code, 4)
        hdr.write({8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}.get(bps, 0), 3)
        hdr.write(0, 1)
        for b in utf8_number(fno):
            hdr.write(b, 8)
        if tail:
            hdr.write(*tail)
        hb = hdr.bytes()
        hb += bytes([crc8(hb)])
        bw = BitWriter()
        chans = [blk[:, c] for c in range(ch)]
        widths = [bps] * ch
        if ch == 2 and stereo == "left_side":
            chans, widths = [chans[0], chans[0] - chans[1]], [bps, bps + 1]
        elif ch == 2 and stereo == "side_right":
            chans, widths = [chans[0] - chans[1], chans[1]], [bps + 1, bps]
        elif ch == 2 and stereo == "mid_side":
            chans, widths = [(chans[0] + chans[1]) >> 1, chans[0] - chans[1]], [bps, bps + 1]
        for c, w in zip(chans, widths):
            kind = kinds[ki % len(kinds)]
            ki += 1
            if kind == "constant" and not np.all(c == c[0]):
                kind = "verbatim"
            if kind.startswith("fixed") and int(kind[5:]) >= n:
                kind = "verbatim"
            if kind == "lpc" and opts.get("lpc_order", 6) >= n:
                kind = "verbatim"
            write_subframe(bw, c, w, kind, rng, opts)
        bw.align()
        frame = hb + bw.bytes()
        frame += struct.pack(">H", crc16(frame))
        out += frame
        fno += 1
    return bytes(out)
