"""Test infrastructure: an independent FLAC bitstream WRITER (RFC 9639) that can produce every construct the decoder in
csrc/flac.hip has to understand — CONSTANT / VERBATIM / FIXED / LPC subframes, both Rice methods, partition orders, escaped
partitions, wasted bits, the three stereo decorrelation modes, every block-size and sample-rate code, an ID3v2 prefix and
extra metadata blocks.  The product's own encoder only emits a small subset (fixed predictors, one partition, independent
channels); without this writer most decoder paths would go untested.  Written bit by bit in Python, sharing no code with
the library."""
import hashlib

import numpy as np


USED = set()          # constructs written since the last clear (the tests assert that the parametrisation covers the format)


class Bits:
    def __init__(self):
        self.buf = bytearray(); self.acc = 0; self.n = 0

    def put(self, v, k):
        if k == 0:
            return
        v &= (1 << k) - 1
        self.acc = (self.acc << k) | v; self.n += k
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1 if self.n else 0

    def unary(self, q):
        while q >= 32:
            self.put(0, 32); q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

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


def utf8(v):
    if v < 0x80:
        return bytes([v])
    out = []
    nb = 2
    while v >= (1 << (5 * nb + 1)) and nb < 7:
        nb += 1
    for i in range(nb - 1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    lead = ((0xFF << (8 - nb)) & 0xFF) | (v >> (6 * (nb - 1)))
    return bytes([lead] + out[::-1])


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SIZE_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


def _residual(bw, res, order, blocksize, rng):
    method = int(rng.integers(0, 2))
    pbits, esc = (5, 31) if method else (4, 15)
    porder = int(rng.integers(0, 4))
    while porder > 0 and (blocksize % (1 << porder) or (blocksize >> porder) < order):
        porder -= 1
    bw.put(method, 2); bw.put(porder, 4)
    USED.add(f"rice{method}"); USED.add(f"porder{porder}")
    i = 0
    for pt in range(1 << porder):
        cnt = (blocksize >> porder) - (order if pt == 0 else 0)
        part = res[i:i + cnt]; i += cnt
        u = np.where(part >= 0, 2 * part, -2 * part - 1)
        mean = float(u.mean()) if cnt else 0.0
        if cnt and (rng.random() < 0.15 or np.log2(mean + 1) > esc - 1):     # escaped partition: raw two's complement
            nb = int(max(1, max(int(abs(int(v))).bit_length() for v in part) + 1)) if cnt else 1
            bw.put(esc, pbits); bw.put(nb, 5); USED.add("escape")
            for v in part:
                bw.put(int(v), nb)
            continue
        k = max(0, min(esc - 1, int(np.log2(mean + 1))))
        bw.put(k, pbits)
        for v in u:
            v = int(v)
            bw.unary(v >> k); bw.put(v & ((1 << k) - 1), k)
    assert i == len(res)


def _subframe(bw, s, bps, rng, force=None):
    """s: int array of one channel's block at `bps` bits."""
    n = len(s)
    wasted = 0
    orv = int(np.bitwise_or.reduce(s.astype(np.int64)))
    if orv:
        wasted = min((orv & -orv).bit_length() - 1, bps - 1)
    if wasted and rng.random() < 0.7:
        s = s >> wasted
    else:
        wasted = 0
    eb = bps - wasted
    kind = force or rng.choice(["fixed", "lpc", "verbatim"], p=[0.45, 0.45, 0.1])
    if (s == s[0]).all():
        kind = "constant"
    USED.add(kind)
    if wasted:
        USED.add("wasted")
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0, 6)
    elif kind == "verbatim":
        bw.put(1, 6)
    elif kind == "fixed":
        order = int(min(rng.integers(0, 5), n))
        bw.put(8 + order, 6)
    else:
        order = int(min(rng.integers(1, 13), n))
        bw.put(31 + order, 6)
    if wasted:
        bw.put(1, 1); bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    s64 = s.astype(np.int64)
    if kind == "constant":
        bw.put(int(s64[0]), eb)
    elif kind == "verbatim":
        for v in s64:
            bw.put(int(v), eb)
    elif kind == "fixed":
        for v in s64[:order]:
            bw.put(int(v), eb)
        pred = np.zeros(n, np.int64)
        if order == 1: pred[1:] = s64[:-1]
        elif order == 2: pred[2:] = 2 * s64[1:-1] - s64[:-2]
        elif order == 3: pred[3:] = 3 * s64[2:-1] - 3 * s64[1:-2] + s64[:-3]
        elif order == 4: pred[4:] = 4 * s64[3:-1] - 6 * s64[2:-2] + 4 * s64[1:-3] - s64[:-4]
        _residual(bw, (s64 - pred)[order:], order, n, rng)
    else:
        for v in s64[:order]:
            bw.put(int(v), eb)
        prec = int(rng.integers(5, 16)); shift = int(rng.integers(2, prec - 1))
        c = rng.normal(0, 0.04, order); c[0] += 1.0           # a perturbed first-order predictor: residuals stay small
        coef = np.clip(np.round(c * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
        bw.put(prec - 1, 4); bw.put(shift, 5)
        for c in coef:
            bw.put(int(c), prec)
        res = np.zeros(n - order, np.int64)
        for i in range(order, n):
            acc = int(np.dot(coef, s64[i - order:i][::-1]))
            res[i - order] = int(s64[i]) - (acc >> shift)
        _residual(bw, res, order, n, rng)


def encode(pcm, bits, sr, seed=0, id3=False, extra_blocks=True, blocksizes=None):
    """pcm: int array [channels, n] of `bits`-bit samples -> FLAC file bytes exercising randomly chosen constructs."""
    rng = np.random.default_rng(seed)
    pcm = np.atleast_2d(np.asarray(pcm, np.int64))
    ch, n = pcm.shape
    raw = np.ascontiguousarray(pcm.T).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :(bits + 7) // 8].tobytes()
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 37]) + bytes(37)
    out += b"fLaC"
    hd = Bits()
    hd.put(0 if extra_blocks else 1, 1); hd.put(0, 7); hd.put(34, 24)
    hd.put(16, 16); hd.put(32768, 16); hd.put(0, 24); hd.put(0, 24)
    hd.put(sr, 20); hd.put(ch - 1, 3); hd.put(bits - 1, 5); hd.put(n, 36)
    out += hd.bytes() + hashlib.md5(raw).digest()
    if extra_blocks:
        out += bytes([4, 0, 0, 8]) + bytes([0, 0, 0, 0, 0, 0, 0, 0])                 # VORBIS_COMMENT, empty
        out += bytes([0x81, 0, 0, 5]) + bytes(5)                                     # PADDING, last
    pos = 0
    variable = blocksizes is None
    choices = [192, 576, 256, 1024, 4096, 100, 1000, 5000] if variable else blocksizes
    while pos < n:
        bs = int(rng.choice(choices))
        bs = min(bs, n - pos)
        bw = Bits()
        bw.put(0x3FFE, 14); bw.put(0, 1); bw.put(1, 1)                              # variable block size stream: sample number coded
        code = BLOCK_CODES.get(bs)
        if code is None:
            code = 6 if bs <= 256 else 7
        bw.put(code, 4); USED.add(f"bs{code}")
        rcode = RATE_CODES.get(sr, None)
        if rcode is None or rng.random() < 0.3:
            rcode = 12 if sr % 1000 == 0 and sr // 1000 < 256 else (13 if sr < 65536 else 14)
            if rng.random() < 0.3:
                rcode = 0
        bw.put(rcode, 4); USED.add(f"rate{rcode}")
        mode = 0
        if ch == 2:
            mode = int(rng.integers(0, 4))
        bw.put(ch - 1 if mode == 0 else 7 + mode, 4); USED.add(f"stereo{mode}")
        bw.put(SIZE_CODES[bits] if rng.random() < 0.7 else 0, 3); bw.put(0, 1)
        for b in utf8(pos):
            bw.put(b, 8)
        if code == 6: bw.put(bs - 1, 8)
        if code == 7: bw.put(bs - 1, 16)
        if rcode == 12: bw.put(sr // 1000, 8)
        if rcode == 13: bw.put(sr, 16)
        if rcode == 14: bw.put(sr // 10, 16)
        bw.put(crc8(bw.bytes()), 8)
        blk = pcm[:, pos:pos + bs]
        if mode == 0:
            subs = [(blk[c], bits) for c in range(ch)]
        elif mode == 1:
            subs = [(blk[0], bits), (blk[0] - blk[1], bits + 1)]
        elif mode == 2:
            subs = [(blk[0] - blk[1], bits + 1), (blk[1], bits)]
        else:
            subs = [((blk[0] + blk[1]) >> 1, bits), (blk[0] - blk[1], bits + 1)]
        for s, b in subs:
            _subframe(bw, s, b, rng)
        bw.align()
        body = bw.bytes()
        out += body + crc16(body).to_bytes(2, "big")
        pos += bs
    return bytes(out)
