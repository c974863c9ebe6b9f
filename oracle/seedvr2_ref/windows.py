"""Window geometry of the NaDiT swin attention (oracle restatement).

Follows /root/reference/src/models/dit_3b/window.py:28-83 (regular / shifted
"720p windows by size") and the index bookkeeping of
/root/reference/src/models/dit_3b/na.py:616-641 (window_idx) and
na.py:320-424 (repeat_concat_idx) for a single sample (the runner always feeds
one sample: generation_phases.py:678).

A window is a box [t0,t1) x [h0,h1) x [w0,w1) of the DiT token grid.  Windows are
listed with w outermost, then h, then t (window.py:43-48, :77-82); tokens inside a
window are in (t, h, w) raster order (x[st, sh, sw] then flatten, na.py:607-612).
"""

from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

Box = Tuple[int, int, int, int, int, int]  # t0, t1, h0, h1, w0, w1

REGULAR = "720pwin_by_size_bysize"
SHIFTED = "720pswin_by_size_bysize"


def _window_size(t: int, h: int, w: int, num: Tuple[int, int, int]) -> Tuple[int, int, int]:
    """window.py:31-35 / :54-58 - window extent derived at 720p-equivalent scale."""
    nt, nh, nw = num
    scale = math.sqrt((45 * 80) / (h * w))
    rh, rw = round(h * scale), round(w * scale)  # python round(): banker's rounding
    wh, ww = math.ceil(rh / nh), math.ceil(rw / nw)
    wt = math.ceil(min(t, 30) / nt)
    return wt, wh, ww


def regular_windows(size: Tuple[int, int, int], num: Tuple[int, int, int] = (4, 3, 3)) -> List[Box]:
    """window.py:28-49."""
    t, h, w = size
    wt, wh, ww = _window_size(t, h, w, num)
    nt, nh, nw = math.ceil(t / wt), math.ceil(h / wh), math.ceil(w / ww)
    out: List[Box] = []
    for iw in range(nw):
        w0, w1 = iw * ww, min((iw + 1) * ww, w)
        if w1 <= w0:
            continue
        for ih in range(nh):
            h0, h1 = ih * wh, min((ih + 1) * wh, h)
            if h1 <= h0:
                continue
            for it in range(nt):
                t0, t1 = it * wt, min((it + 1) * wt, t)
                if t1 <= t0:
                    continue
                out.append((t0, t1, h0, h1, w0, w1))
    return out


def shifted_windows(size: Tuple[int, int, int], num: Tuple[int, int, int] = (4, 3, 3)) -> List[Box]:
    """window.py:51-83 - half-window shift, int() truncation, empty windows dropped."""
    t, h, w = size
    wt, wh, ww = _window_size(t, h, w, num)
    st = 0.5 if wt < t else 0
    sh = 0.5 if wh < h else 0
    sw = 0.5 if ww < w else 0
    nt, nh, nw = math.ceil((t - st) / wt), math.ceil((h - sh) / wh), math.ceil((w - sw) / ww)
    nt = nt + 1 if st > 0 else 1
    nh = nh + 1 if sh > 0 else 1
    nw = nw + 1 if sw > 0 else 1

    def span(i: int, s: float, win: int, lim: int) -> Tuple[int, int]:
        return max(int((i - s) * win), 0), min(int((i - s + 1) * win), lim)

    out: List[Box] = []
    for iw in range(nw):
        w0, w1 = span(iw, sw, ww, w)
        if w1 <= w0:
            continue
        for ih in range(nh):
            h0, h1 = span(ih, sh, wh, h)
            if h1 <= h0:
                continue
            for it in range(nt):
                t0, t1 = span(it, st, wt, t)
                if t1 <= t0:
                    continue
                out.append((t0, t1, h0, h1, w0, w1))
    return out


def windows_for(method: str, size, num=(4, 3, 3)) -> List[Box]:
    if method == REGULAR:
        return regular_windows(size, num)
    if method == SHIFTED:
        return shifted_windows(size, num)
    raise ValueError(f"Unknown windowing method: {method}")


def window_token_index(size: Tuple[int, int, int], boxes: List[Box]):
    """Natural-order token ids listed window by window (= ``tgt_idx`` of na.window_idx,
    na.py:633-636) plus per-window lengths and (t,h,w) extents."""
    t, h, w = size
    ids = np.arange(t * h * w, dtype=np.int64).reshape(t, h, w)
    idx, lens, shapes = [], [], []
    for (t0, t1, h0, h1, w0, w1) in boxes:
        blk = ids[t0:t1, h0:h1, w0:w1]
        idx.append(blk.reshape(-1))
        lens.append(blk.size)
        shapes.append(blk.shape)
    return np.concatenate(idx), np.asarray(lens, dtype=np.int64), np.asarray(shapes, dtype=np.int64)
