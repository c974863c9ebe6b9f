"""Host-side window geometry and index tables for the NaDiT swin attention.

The device kernels consume flat int32 tables; this module builds them once per
(token grid, window method, text length) and caches them.  Geometry follows the reference:
  * window extents at 720p-equivalent scale, regular vs half-shifted layouts
    [ref: src/models/dit_3b/window.py:28-49, :51-83]
  * windows listed w-outer / h / t-inner, tokens inside a window in (t,h,w) raster order
    [ref: src/models/dit_3b/na.py:593-612, 616-641]
  * every window's sequence = [its video tokens ; all text tokens]
    [ref: src/models/dit_3b/na.py:320-424]
  * rope coordinates are window-local, video time offset by the text length
    [ref: src/models/dit_3b/rope.py:158-175]
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from functools import lru_cache
from typing import List, Tuple

import numpy as np

REGULAR = "720pwin_by_size_bysize"
SHIFTED = "720pswin_by_size_bysize"
QTILE = 128  # queries per attention workgroup (csrc/attn.hip)


def _extent(t: int, h: int, w: int, num) -> Tuple[int, int, int]:
    s = math.sqrt(3600.0 / (h * w))  # 45*80 tokens = 720p
    return (math.ceil(min(t, 30) / num[0]), math.ceil(round(h * s) / num[1]), math.ceil(round(w * s) / num[2]))


def _cuts_regular(n: int, win: int) -> List[Tuple[int, int]]:
    return [(a, min(a + win, n)) for a in range(0, n, win)]


def _cuts_shifted(n: int, win: int) -> List[Tuple[int, int]]:
    if not win < n:  # axis not split: one window, no shift
        return [(0, min(int(1.0 * win), n))] if min(win, n) > 0 else []
    cnt = math.ceil((n - 0.5) / win) + 1
    out = []
    for i in range(cnt):
        a, b = max(int((i - 0.5) * win), 0), min(int((i + 0.5) * win), n)
        if b > a:
            out.append((a, b))
    return out


def window_boxes(method: str, grid: Tuple[int, int, int], num=(4, 3, 3)) -> List[Tuple[int, int, int, int, int, int]]:
    t, h, w = grid
    wt, wh, ww = _extent(t, h, w, num)
    if method == REGULAR:
        ct, ch, cw = _cuts_regular(t, wt), _cuts_regular(h, wh), _cuts_regular(w, ww)
    elif method == SHIFTED:
        ct, ch, cw = _cuts_shifted(t, wt), _cuts_shifted(h, wh), _cuts_shifted(w, ww)
    else:
        raise ValueError(f"Unknown windowing method: {method}")
    return [(a[0], a[1], b[0], b[1], c[0], c[1]) for c in cw for b in ch for a in ct]


@dataclass(frozen=True)
class WindowPlan:
    grid: Tuple[int, int, int]
    method: str
    txt_len: int
    nwin: int
    total: int              # S = sum(window tokens) + nwin * txt_len
    src_row: np.ndarray     # int32 [S]   >=0 video token (natural order), <0 text row -1-j
    pos: np.ndarray         # int32 [S,3] rope coordinates
    cu_seqlens: np.ndarray  # int32 [nwin+1]
    tile_seq: np.ndarray    # int32 [ntiles]
    tile_q0: np.ndarray     # int32 [ntiles]
    inv_rows: np.ndarray    # int32 [L]   window-order position of natural token m
    txt_row0: np.ndarray    # int32 [nwin] position of the first text row of every window
    max_pos: int


@lru_cache(maxsize=64)
def plan(grid: Tuple[int, int, int], method: str, txt_len: int, num=(4, 3, 3)) -> WindowPlan:
    t, h, w = grid
    boxes = window_boxes(method, grid, num)
    ids = np.arange(t * h * w, dtype=np.int64).reshape(t, h, w)
    src, pos, cu, tseq, tq0, trow0 = [], [], [0], [], [], []
    txt_src = -1 - np.arange(txt_len, dtype=np.int64)
    txt_pos = np.repeat(np.arange(txt_len, dtype=np.int64)[:, None], 3, axis=1)
    off = 0
    for wi, (t0, t1, h0, h1, w0, w1) in enumerate(boxes):
        blk = ids[t0:t1, h0:h1, w0:w1]
        n = blk.size
        tt, yy, xx = np.meshgrid(np.arange(t1 - t0) + txt_len, np.arange(h1 - h0), np.arange(w1 - w0), indexing="ij")
        src.append(blk.reshape(-1)); src.append(txt_src)
        pos.append(np.stack((tt.reshape(-1), yy.reshape(-1), xx.reshape(-1)), axis=1)); pos.append(txt_pos)
        trow0.append(off + n)
        seqlen = n + txt_len
        for q0 in range(0, seqlen, QTILE):
            tseq.append(wi); tq0.append(q0)
        off += seqlen
        cu.append(off)
    src = np.concatenate(src)
    pos = np.concatenate(pos)
    inv = np.full(t * h * w, -1, dtype=np.int64)
    vid_mask = src >= 0
    inv[src[vid_mask]] = np.nonzero(vid_mask)[0]
    if (inv < 0).any():
        raise RuntimeError("window layout does not cover the token grid")  # windows must partition the grid
    return WindowPlan(grid, method, txt_len, len(boxes), int(off), src.astype(np.int32), pos.astype(np.int32),
                      np.asarray(cu, dtype=np.int32), np.asarray(tseq, dtype=np.int32),
                      np.asarray(tq0, dtype=np.int32), inv.astype(np.int32), np.asarray(trow0, dtype=np.int32),
                      int(pos.max()) + 1)
