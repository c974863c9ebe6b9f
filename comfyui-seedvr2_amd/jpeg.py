"""Baseline JPEG on the device for Motion-JPEG output ("device -> encoder writer").

The reference hands fp32 frames to cv2's `mp4v` encoder on the host [ref: inference_cli.py:666-706]; no codec library
exists in this image, so the compressed container offered here is Motion-JPEG in AVI: every stage of a baseline JPEG
(ITU-T T.81) runs as HIP kernels on the finished bf16 frames (`csrc/jpeg.hip`) -- 8-bit quantisation, JFIF YCbCr, 4:2:0
subsampling, the 8x8 integer forward DCT (LL&M, 13-bit constants), quantisation, zig-zag, and Huffman coding with one
restart interval per MCU row so that rows are coded in parallel -- and the host only concatenates header + scan + EOI.
Not a format of the reference (its mp4v bit stream is produced inside cv2/ffmpeg): parity is defined by the numpy
restatement in oracle/seedvr2_ref/jpeg.py (bit-exact) and by an independent decoder (PIL/libjpeg) reading the files.

This module holds the tables (T.81 Annex K) and the header writer; they are the single source for the kernels' inputs.
"""

from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63], dtype=np.int32)

# T.81 Table K.1 / K.2 (natural order)
Q_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                   87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120,
                   101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int32)
Q_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99,
                     99, 99, 99, 99] + [99] * 32, dtype=np.int32)

# T.81 Tables K.3 - K.6: number of codes per length 1..16, then the symbols in code order
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
           [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32,
            0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16,
            0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
            0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
            0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94,
            0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
            0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
            0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
            0xf9, 0xfa])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81,
              0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34,
              0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
              0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
              0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92,
              0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
              0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
              0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
              0xf9, 0xfa])


def quant_tables(quality: int) -> Tuple[np.ndarray, np.ndarray]:
    """Annex K tables scaled like libjpeg's jpeg_set_quality (natural order, values 1..255)."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    f = lambda t: np.clip((t * scale + 50) // 100, 1, 255).astype(np.int32)
    return f(Q_LUMA), f(Q_CHROMA)


def huffman_codes(bits, vals) -> Dict[int, Tuple[int, int]]:
    """symbol -> (code, length), canonical assignment of T.81 Annex C."""
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


def device_tables(quality: int) -> Tuple[np.ndarray, np.ndarray]:
    """(divisors int32 [2][64] in ZIG-ZAG order = 8 * Q, huff uint32 [2][12 + 256] = length << 16 | code; 0 = no code)."""
    ql, qc = quant_tables(quality)
    div = np.stack([ql[ZIGZAG] * 8, qc[ZIGZAG] * 8]).astype(np.int32)
    huff = np.zeros((2, 12 + 256), dtype=np.uint32)
    for t, (dc, ac) in enumerate(((DC_LUMA, AC_LUMA), (DC_CHROMA, AC_CHROMA))):
        for sym, (code, ln) in huffman_codes(*dc).items():
            huff[t, sym] = (ln << 16) | code
        for sym, (code, ln) in huffman_codes(*ac).items():
            huff[t, 12 + sym] = (ln << 16) | code
    return div, huff


def frame_header(width: int, height: int, quality: int) -> bytes:
    """SOI, JFIF APP0, DQT x2, SOF0 (4:2:0), DHT x4, DRI (one restart interval per MCU row), SOS."""
    ql, qc = quant_tables(quality)
    seg = lambda marker, payload: struct.pack(">BBH", 0xFF, marker, len(payload) + 2) + payload
    out = b"\xff\xd8" + seg(0xE0, b"JFIF\0" + struct.pack(">BBBHHBB", 1, 1, 0, 1, 1, 0, 0))
    out += seg(0xDB, bytes([0]) + bytes(ql[ZIGZAG].astype(np.uint8).tolist()))
    out += seg(0xDB, bytes([1]) + bytes(qc[ZIGZAG].astype(np.uint8).tolist()))
    out += seg(0xC0, struct.pack(">BHHB", 8, height, width, 3) + bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1]))
    for tc_th, (bits, vals) in ((0x00, DC_LUMA), (0x10, AC_LUMA), (0x01, DC_CHROMA), (0x11, AC_CHROMA)):
        out += seg(0xC4, bytes([tc_th]) + bytes(bits) + bytes(vals))
    out += seg(0xDD, struct.pack(">H", (width + 15) // 16))
    out += seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    return out
