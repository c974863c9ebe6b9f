"""CPU restatement of the device JPEG encoder (TEST INFRASTRUCTURE ONLY): numpy integer arithmetic for the pixel stages,
plain Python loops for the Huffman coder (small frames only).

Not a reference format (the reference's writer is cv2's mp4v encoder, inference_cli.py:666-706): this file DEFINES the
arithmetic the kernels in csrc/jpeg.hip must reproduce bit for bit -- baseline JPEG, ITU-T T.81: JFIF YCbCr in 16-bit fixed
point, 4:2:0 by rounded 2x2 means, the LL&M integer forward DCT with 13-bit constants (output scaled by 8), quantisation
by rounding division, zig-zag, Huffman coding with the Annex K tables and one restart interval per MCU row -- and the
tests also decode the result with an independent decoder (PIL / libjpeg).
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F_0_298, F_0_390, F_0_541, F_0_765, F_0_899, F_1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501, F_1_847, F_1_961, F_2_053, F_2_562, F_3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _pass(d, shift_even_up: bool):
    """One 1-D pass of the LL&M forward DCT along the last axis (int64 in, int64 out)."""
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    t0, t7, t1, t6, t2, t5, t3, t4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    n_odd = CONST_BITS - PASS1_BITS if shift_even_up else CONST_BITS + PASS1_BITS
    if shift_even_up:
        o0, o4 = (t10 + t11) << PASS1_BITS, (t10 - t11) << PASS1_BITS
    else:
        o0, o4 = _descale(t10 + t11, PASS1_BITS), _descale(t10 - t11, PASS1_BITS)
    z1 = (t12 + t13) * F_0_541
    o2 = _descale(z1 + t13 * F_0_765, n_odd)
    o6 = _descale(z1 - t12 * F_1_847, n_odd)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * F_1_175
    t4, t5, t6, t7 = t4 * F_0_298, t5 * F_2_053, t6 * F_3_072, t7 * F_1_501
    z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
    o7, o5, o3, o1 = _descale(t4 + z1 + z3, n_odd), _descale(t5 + z2 + z4, n_odd), _descale(t6 + z2 + z3, n_odd), _descale(t7 + z1 + z4, n_odd)
    return np.stack([o0, o1, o2, o3, o4, o5, o6, o7], axis=-1)


def fdct(blocks: np.ndarray) -> np.ndarray:
    """[...,8,8] level-shifted samples (row, column) -> [...,8,8] DCT coefficients scaled by 8 (rows first, then columns)."""
    x = _pass(blocks.astype(np.int64), True)
    return np.swapaxes(_pass(np.swapaxes(x, -1, -2), False), -1, -2)


def ycbcr420(rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """uint8 [H,W,3] -> (Y [Hp,Wp], Cb, Cr [Hp/2,Wp/2]) int32; Hp, Wp = sizes rounded up to 16 by edge replication."""
    H, W, _ = rgb.shape
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    p = np.pad(rgb.astype(np.int64), ((0, Hp - H), (0, Wp - W), (0, 0)), mode="edge")
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    q = (p.reshape(Hp // 2, 2, Wp // 2, 2, 3).sum(axis=(1, 3)) + 2) >> 2           # rounded 2x2 mean of RGB
    r, g, b = q[..., 0], q[..., 1], q[..., 2]
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    return y.astype(np.int32), cb.astype(np.int32), cr.astype(np.int32)


def coefficients(rgb: np.ndarray, div_zz: np.ndarray, zigzag: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3] -> int32 [mcu_rows, mcu_cols, 6, 64] quantised coefficients in zig-zag order (Y00 Y01 Y10 Y11 Cb Cr);
    div_zz int32 [2][64] = 8*Q in zig-zag order."""
    y, cb, cr = ycbcr420(rgb)
    R, Cn = y.shape[0] // 16, y.shape[1] // 16
    yb = y.reshape(R, 2, 8, Cn, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(R, Cn, 4, 8, 8)
    cbb = cb.reshape(R, 8, Cn, 8).transpose(0, 2, 1, 3)[:, :, None]
    crb = cr.reshape(R, 8, Cn, 8).transpose(0, 2, 1, 3)[:, :, None]
    blocks = np.concatenate([yb, cbb, crb], axis=2) - 128
    c = fdct(blocks).reshape(R, Cn, 6, 64)[..., zigzag]
    d = np.stack([div_zz[0]] * 4 + [div_zz[1]] * 2).astype(np.int64)
    mag = (np.abs(c) + (d >> 1)) // d
    return (np.sign(c) * mag).astype(np.int32)


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code: int, length: int):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def encode_row(coefs: np.ndarray, huff: List[Tuple[Dict[int, Tuple[int, int]], Dict[int, Tuple[int, int]]]]) -> bytes:
    """One restart interval: coefs int32 [mcu_cols, 6, 64]; huff[t] = (dc codes, ac codes) for t = 0 luma, 1 chroma."""
    bw, last = _Bits(), [0, 0, 0]
    for mcu in coefs:
        for k in range(6):
            comp = 0 if k < 4 else k - 3
            dc, ac = huff[0 if k < 4 else 1]
            blk = [int(v) for v in mcu[k]]
            diff = blk[0] - last[comp]
            last[comp] = blk[0]
            s = abs(diff).bit_length()
            bw.put(*dc[s])
            if s:
                bw.put(diff if diff > 0 else diff - 1, s)
            run = 0
            for v in blk[1:]:
                if v == 0:
                    run += 1
                    continue
                while run >= 16:
                    bw.put(*ac[0xF0])
                    run -= 16
                s = abs(v).bit_length()
                bw.put(*ac[(run << 4) | s])
                bw.put(v if v > 0 else v - 1, s)
                run = 0
            if run:
                bw.put(*ac[0x00])
    bw.flush()
    return bytes(bw.out)


def encode_scan(rgb: np.ndarray, div_zz: np.ndarray, zigzag: np.ndarray, huff) -> bytes:
    """Entropy-coded segment of one frame: MCU rows separated by RST0..RST7 markers."""
    c = coefficients(rgb, div_zz, zigzag)
    out = bytearray()
    for r in range(c.shape[0]):
        out += encode_row(c[r], huff)
        if r + 1 < c.shape[0]:
            out += bytes([0xFF, 0xD0 + (r & 7)])
    return bytes(out)
