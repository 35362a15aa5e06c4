"""A small baseline JPEG writer for the tests: any luminance sampling factors (chroma at 1 x 1), an optional restart
interval; encode_scans: the same image with every component in a scan of its own (sequential, non-interleaved), as SOF0 or SOF1.  Quantisation and Huffman tables are lifted from a file libjpeg (PIL) wrote at the same quality -- the Annex K
tables -- so nothing is typed in here; the samples are this file's own DCT.  Test infrastructure only."""
import io

import numpy as np
from scipy.fft import dctn

ZIG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
       57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _segments(data):
    pos, out = 2, []
    while pos < len(data):
        assert data[pos] == 0xFF
        m, n = data[pos + 1], (data[pos + 2] << 8) | data[pos + 3]
        out.append((m, data[pos + 4:pos + 2 + n]))
        if m == 0xDA:
            break
        pos += 2 + n
    return out


def _tables(quality):
    from PIL import Image
    buf = io.BytesIO()
    Image.new("RGB", (16, 16)).save(buf, "JPEG", quality=quality, subsampling=2)
    q, huff, raw = {}, {}, []
    for m, seg in _segments(buf.getvalue()):
        if m in (0xDB, 0xC4):
            raw.append(bytes([0xFF, m, (len(seg) + 2) >> 8, (len(seg) + 2) & 255]) + seg)
        o = 0
        while m == 0xDB and o < len(seg):
            t = np.zeros(64, np.int32)
            t[ZIG] = np.frombuffer(seg[o + 1:o + 65], np.uint8)
            q[seg[o] & 15] = t.reshape(8, 8)
            o += 65
        while m == 0xC4 and o < len(seg):
            bits = list(seg[o + 1:o + 17])
            vals = list(seg[o + 17:o + 17 + sum(bits)])
            codes, code, k = {}, 0, 0
            for ln in range(1, 17):
                for _ in range(bits[ln - 1]):
                    codes[vals[k]] = (code, ln)
                    code += 1
                    k += 1
                code <<= 1
            huff[(seg[o] >> 4, seg[o] & 15)] = codes
            o += 17 + sum(bits)
    return q, huff, b"".join(raw)


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | code
        self.n += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _amp(v):
    n = int(abs(v)).bit_length()
    return n, (v if v >= 0 else v + (1 << n) - 1)


def encode(img, hy, vy, quality=85, restart=0):
    """RGB(A) uint8 -> a baseline JFIF file, luminance factors hy x vy (1, 2, 4 x 1, 2), chroma 1 x 1"""
    q, huff, tab_bytes = _tables(quality)
    rgb = img[..., :3].astype(np.float64)
    h, w = rgb.shape[:2]
    mw, mh = 8 * hy, 8 * vy
    mx, my = (w + mw - 1) // mw, (h + mh - 1) // mh
    rgb = np.pad(rgb, ((0, my * mh - h), (0, mx * mw - w), (0, 0)), mode="edge")
    yy = 0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]
    cb = -0.168736 * rgb[..., 0] - 0.331264 * rgb[..., 1] + 0.5 * rgb[..., 2] + 128
    cr = 0.5 * rgb[..., 0] - 0.418688 * rgb[..., 1] - 0.081312 * rgb[..., 2] + 128
    down = lambda p: p.reshape(my * 8, vy, mx * 8, hy).mean(axis=(1, 3))
    planes = [yy, down(cb), down(cr)]

    def block(p, by, bx, qt):
        c = dctn(p[8 * by:8 * by + 8, 8 * bx:8 * bx + 8] - 128.0, norm="ortho")
        return np.rint(c / qt).astype(np.int32).reshape(64)[ZIG]

    bw = _Bits()
    pred = [0, 0, 0]
    body = bytearray()
    for m in range(mx * my):
        if restart and m and m % restart == 0:
            bw.flush()
            body += bw.out + bytes([0xFF, 0xD0 + ((m // restart - 1) & 7)])
            bw, pred = _Bits(), [0, 0, 0]
        my0, mx0 = divmod(m, mx)
        for c in range(3):
            for i in range(hy * vy if c == 0 else 1):
                by, bx = (my0 * vy + i // hy, mx0 * hy + i % hy) if c == 0 else (my0, mx0)
                zz = block(planes[c], by, bx, q[0 if c == 0 else 1])
                dc, ac = huff[(0, 0 if c == 0 else 1)], huff[(1, 0 if c == 0 else 1)]
                n, bits = _amp(int(zz[0]) - pred[c])
                pred[c] = int(zz[0])
                bw.put(*dc[n])
                if n:
                    bw.put(bits, n)
                run = 0
                last = max([k for k in range(1, 64) if zz[k]], default=0)
                for k in range(1, last + 1):
                    if zz[k] == 0:
                        run += 1
                        continue
                    while run > 15:
                        bw.put(*ac[0xF0])
                        run -= 16
                    n, bits = _amp(int(zz[k]))
                    bw.put(*ac[(run << 4) | n])
                    bw.put(bits, n)
                    run = 0
                if last < 63:
                    bw.put(*ac[0])
    bw.flush()
    body += bw.out
    sof = bytes([0xFF, 0xC0, 0, 17, 8, h >> 8, h & 255, w >> 8, w & 255, 3, 1, (hy << 4) | vy, 0, 2, 0x11, 1, 3, 0x11, 1])
    dri = bytes([0xFF, 0xDD, 0, 4, restart >> 8, restart & 255]) if restart else b""
    sos = bytes([0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0])
    app0 = bytes([0xFF, 0xE0, 0, 16]) + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    return b"\xff\xd8" + app0 + tab_bytes + sof + dri + sos + bytes(body) + b"\xff\xd9"


def _put_block(bw, zz, pred, dc, ac):
    n, bits = _amp(int(zz[0]) - pred)
    bw.put(*dc[n])
    if n:
        bw.put(bits, n)
    run = 0
    last = max([k for k in range(1, 64) if zz[k]], default=0)
    for k in range(1, last + 1):
        if zz[k] == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*ac[0xF0])
            run -= 16
        n, bits = _amp(int(zz[k]))
        bw.put(*ac[(run << 4) | n])
        bw.put(bits, n)
        run = 0
    if last < 63:
        bw.put(*ac[0])
    return int(zz[0])


def encode_scans(img, hy, vy, quality=85, sof=0xC0, order=(0, 1, 2), restart=0, requant_between=False):
    """As encode(), but sequential with one scan per component (T.81 A.2.3: the component's own blocks in raster order, none for
    blocks wholly outside the image), the scans in `order`; sof 0xC0 (baseline) or 0xC1 (extended sequential).  restart: blocks
    per interval (a one-component scan's MCU is one block).  requant_between: a DQT that redefines table 1 between the Cb and
    the Cr scan -- a decoder that dequantises at the end of the file instead of scan by scan gets Cb wrong."""
    q, huff, tab_bytes = _tables(quality)
    rgb = img[..., :3].astype(np.float64)
    h, w = rgb.shape[:2]
    mw, mh = 8 * hy, 8 * vy
    mx, my = (w + mw - 1) // mw, (h + mh - 1) // mh
    rgb = np.pad(rgb, ((0, my * mh - h), (0, mx * mw - w), (0, 0)), mode="edge")
    yy = 0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]
    cb = -0.168736 * rgb[..., 0] - 0.331264 * rgb[..., 1] + 0.5 * rgb[..., 2] + 128
    cr = 0.5 * rgb[..., 0] - 0.418688 * rgb[..., 1] - 0.081312 * rgb[..., 2] + 128
    down = lambda p: p.reshape(my * 8, vy, mx * 8, hy).mean(axis=(1, 3))
    planes = [yy, down(cb), down(cr)]
    q2 = np.maximum(1, (q[1] * 2) // 3)                      # the chroma table Cr is coded with when requant_between
    out = bytearray()
    for c in order:
        cw, ch = (w, h) if c == 0 else ((w + hy - 1) // hy, (h + vy - 1) // vy)
        nbx, nby = (cw + 7) // 8, (ch + 7) // 8
        qt = q[0] if c == 0 else (q2 if (requant_between and c == 2) else q[1])
        if requant_between and c == 2:
            zz = np.zeros(64, np.uint8)
            zz[:] = q2.reshape(64)[ZIG]
            out += bytes([0xFF, 0xDB, 0, 67, 1]) + zz.tobytes()
        out += bytes([0xFF, 0xDA, 0, 8, 1, c + 1, 0x00 if c == 0 else 0x11, 0, 63, 0])
        bw, pred, k = _Bits(), 0, 0
        for by in range(nby):
            for bx in range(nbx):
                if restart and k and k % restart == 0:
                    bw.flush()
                    out += bw.out + bytes([0xFF, 0xD0 + ((k // restart - 1) & 7)])
                    bw, pred = _Bits(), 0
                blk = dctn(planes[c][8 * by:8 * by + 8, 8 * bx:8 * bx + 8] - 128.0, norm="ortho")
                zz = np.rint(blk / qt).astype(np.int32).reshape(64)[ZIG]
                pred = _put_block(bw, zz, pred, huff[(0, 0 if c == 0 else 1)], huff[(1, 0 if c == 0 else 1)])
                k += 1
        bw.flush()
        out += bw.out
    frame = bytes([0xFF, sof, 0, 17, 8, h >> 8, h & 255, w >> 8, w & 255, 3, 1, (hy << 4) | vy, 0, 2, 0x11, 1, 3, 0x11, 1])
    dri = bytes([0xFF, 0xDD, 0, 4, restart >> 8, restart & 255]) if restart else b""
    app0 = bytes([0xFF, 0xE0, 0, 16]) + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    return b"\xff\xd8" + app0 + tab_bytes + frame + dri + bytes(out) + b"\xff\xd9"
