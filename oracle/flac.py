"""Minimal pure-Python FLAC decoder (mono/stereo, 8-24 bit, FIXED/LPC/VERBATIM/CONSTANT subframes).

TEST INFRASTRUCTURE ONLY: used once by oracle/gen_golden.py to turn the reference's test recording
(training/tests/test_data/*.flac) into a sample fixture, because this image has no audio decoder
(no soundfile / torchaudio / ffmpeg).  Follows the FLAC format specification (RFC 9639).
"""
import numpy as np


class _Bits:
    def __init__(self, data, pos=0):
        self.d, self.p = data, pos * 8

    def read(self, n):
        v = 0
        while n > 0:
            byte = self.d[self.p >> 3]
            avail = 8 - (self.p & 7)
            take = min(avail, n)
            v = (v << take) | ((byte >> (avail - take)) & ((1 << take) - 1))
            self.p += take
            n -= take
        return v

    def read_signed(self, n):
        v = self.read(n)
        return v - (1 << n) if v >> (n - 1) else v

    def unary(self):
        n = 0
        while self.read(1) == 0:
            n += 1
        return n

    def align(self):
        self.p = (self.p + 7) & ~7


def _residual(br, order, blocksize):
    method = br.read(2)
    pbits = 4 if method == 0 else 5
    porder = br.read(4)
    nparts = 1 << porder
    out = []
    for part in range(nparts):
        n = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == (1 << pbits) - 1:  # escape: raw
            nb = br.read(5)
            out += [br.read_signed(nb) if nb else 0 for _ in range(n)]
        else:
            for _ in range(n):
                q = br.unary()
                u = (q << k) | (br.read(k) if k else 0)
                out.append((u >> 1) ^ -(u & 1))
    return out


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(br, bps, blocksize):
    assert br.read(1) == 0
    t = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    if t == 0:
        s = [br.read_signed(bps)] * blocksize
    elif t == 1:
        s = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= t <= 12:
        order = t - 8
        s = [br.read_signed(bps) for _ in range(order)]
        res = _residual(br, order, blocksize)
        c = _FIXED[order]
        for r in res:
            s.append(r + sum(ci * s[-1 - i] for i, ci in enumerate(c)))
    elif t >= 32:
        order = t - 31
        s = [br.read_signed(bps) for _ in range(order)]
        prec = br.read(4) + 1
        shift = br.read_signed(5)
        coef = [br.read_signed(prec) for _ in range(order)]
        res = _residual(br, order, blocksize)
        for r in res:
            s.append(r + (sum(ci * s[-1 - i] for i, ci in enumerate(coef)) >> shift))
    else:
        raise ValueError(f"reserved subframe type {t}")
    return [v << wasted for v in s] if wasted else s


def decode(path):
    """-> (samples int32 [n, channels], sample_rate, bits_per_sample)."""
    data = open(path, "rb").read()
    assert data[:4] == b"fLaC"
    pos, sr, ch, bps, total = 4, None, None, None, None
    while True:
        hdr = data[pos]
        size = int.from_bytes(data[pos + 1:pos + 4], "big")
        if hdr & 0x7F == 0:
            b = _Bits(data, pos + 4)
            b.read(16); b.read(16); b.read(24); b.read(24)
            sr, ch, bps, total = b.read(20), b.read(3) + 1, b.read(5) + 1, b.read(36)
        pos += 4 + size
        if hdr & 0x80:
            break
    out = [[] for _ in range(ch)]
    br = _Bits(data, pos)
    while sum(1 for _ in [0]) and len(out[0]) < total:
        assert br.read(14) == 0x3FFE, "lost frame sync"
        br.read(1)
        br.read(1)  # blocking strategy
        bs_code, sr_code, ch_code, bps_code = br.read(4), br.read(4), br.read(4), br.read(3)
        br.read(1)
        first = br.read(8)  # UTF-8 coded frame / sample number
        n = 0
        while first & (0x80 >> n):
            n += 1
        for _ in range(max(n - 1, 0)):
            br.read(8)
        if bs_code == 1:
            blocksize = 192
        elif 2 <= bs_code <= 5:
            blocksize = 576 << (bs_code - 2)
        elif bs_code == 6:
            blocksize = br.read(8) + 1
        elif bs_code == 7:
            blocksize = br.read(16) + 1
        else:
            blocksize = 256 << (bs_code - 8)
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        br.read(8)  # CRC-8
        fbps = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24}[bps_code]
        if ch_code < 8:
            subs = [_subframe(br, fbps, blocksize) for _ in range(ch_code + 1)]
        else:  # stereo decorrelation
            side_ch = {8: 1, 9: 0, 10: 1}[ch_code]
            a = _subframe(br, fbps + (1 if side_ch == 0 else 0), blocksize)
            b = _subframe(br, fbps + (1 if side_ch == 1 else 0), blocksize)
            if ch_code == 8:
                subs = [a, [x - y for x, y in zip(a, b)]]
            elif ch_code == 9:
                subs = [[x + y for x, y in zip(a, b)], b]
            else:
                subs = [[(((m << 1) | (s & 1)) + s) >> 1 for m, s in zip(a, b)],
                        [(((m << 1) | (s & 1)) - s) >> 1 for m, s in zip(a, b)]]
        br.align()
        br.read(16)  # CRC-16
        for c in range(ch):
            out[c] += subs[c]
    return np.array(out, dtype=np.int32).T[:total], sr, bps
