"""CPU restatement of the reference CLI's frame I/O arithmetic and chunk streaming (TEST INFRASTRUCTURE ONLY:
imported by tests/, smoke() and bench.py's cpu_baseline, never by the product path).

PARITY UNPINNED for the loop and the colour matrix: `/root/reference/inference_cli.py` imports `cv2` and (through
`src.utils.model_registry`) `diffusers` at module level, neither of which exists in this image, so the module cannot be
run here and holds no fixtures for this path; the functions below restate its text line by line.  The uint8 <-> float
chains are plain numpy/torch casts (same library calls as the reference, so they are the reference arithmetic).
The YUV matrix is the 8-bit integer BT.601 studio-range form; the reference delegates it to cv2's ffmpeg backend
(BGR24 -> yuv420p in front of the `mp4v` encoder), whose arithmetic is not in the reference tree.
"""

from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch


def u8_to_unit(u8_rgb: np.ndarray) -> torch.Tensor:
    """uint8 RGB [T,H,W,3] -> bf16 in [0,1]:  astype(float32) / 255 [ref: inference_cli.py:543] -> .to(float16)
    [ref: :627] -> compute dtype bf16 [ref: src/core/generation_phases.py:376-384]."""
    f = u8_rgb.astype(np.float32) / 255.0
    return torch.from_numpy(f).to(torch.float16).to(torch.bfloat16)


def unit_to_u8(frames: torch.Tensor) -> np.ndarray:
    """bf16/fp32 [T,H,W,3] in [0,1] -> uint8:  result.to(float32) [ref: inference_cli.py:931] then
    (frames.cpu().numpy() * 255.0).astype(np.uint8) [ref: :520, :691, :734] (truncation)."""
    return (frames.to(torch.float32).cpu().numpy() * 255.0).astype(np.uint8)


def rgb_to_bgr_rows(u8_rgb: np.ndarray, bottom_up: bool = False, row_stride: Optional[int] = None) -> np.ndarray:
    """RGB frames -> the byte rows of a BGR24 container frame (cv2.COLOR_RGB2BGR, inference_cli.py:661-664, 702), DIB row
    order and zero padding to `row_stride` bytes.  Returns uint8 [T, H*row_stride]."""
    T, H, W, _ = u8_rgb.shape
    stride = row_stride or W * 3
    out = np.zeros((T, H, stride), dtype=np.uint8)
    bgr = u8_rgb[..., ::-1].reshape(T, H, W * 3)
    out[:, :, : W * 3] = bgr[:, ::-1] if bottom_up else bgr
    return out.reshape(T, H * stride)


def bgr_rows_to_rgb(raw: np.ndarray, H: int, W: int, bottom_up: bool = False, row_stride: Optional[int] = None) -> np.ndarray:
    stride = row_stride or W * 3
    rows = raw.reshape(raw.shape[0], H, stride)[:, :, : W * 3].reshape(raw.shape[0], H, W, 3)
    if bottom_up:
        rows = rows[:, ::-1]
    return np.ascontiguousarray(rows[..., ::-1])


def rgb_to_yuv(u8_rgb: np.ndarray, sub: int = 2) -> np.ndarray:
    """uint8 RGB [T,H,W,3] -> planar frames [T, H*W + 2*(H/sub)*(W/sub)] (Y, U, V), 8-bit integer BT.601 studio range;
    4:2:0 chroma from the rounded 2x2 mean of RGB."""
    T, H, W, _ = u8_rgb.shape
    p = u8_rgb.astype(np.int32)
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    if sub == 2:
        q = (p.reshape(T, H // 2, 2, W // 2, 2, 3).sum(axis=(2, 4)) + 2) >> 2
        r, g, b = q[..., 0], q[..., 1], q[..., 2]
    u = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128
    v = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128
    planes = [np.clip(a, 0, 255).astype(np.uint8).reshape(T, -1) for a in (y, u, v)]
    return np.concatenate(planes, axis=1)


def yuv_to_rgb(raw: np.ndarray, H: int, W: int, sub: int = 2) -> np.ndarray:
    """Inverse of the matrix above with replicated chroma -> uint8 RGB [T,H,W,3]."""
    T = raw.shape[0]
    ch, cw = (H + sub - 1) // sub, (W + sub - 1) // sub
    y = raw[:, : H * W].reshape(T, H, W).astype(np.int32)
    u = raw[:, H * W: H * W + ch * cw].reshape(T, ch, cw).astype(np.int32)
    v = raw[:, H * W + ch * cw: H * W + 2 * ch * cw].reshape(T, ch, cw).astype(np.int32)
    if sub == 2:
        u = np.repeat(np.repeat(u, 2, axis=1), 2, axis=2)[:, :H, :W]
        v = np.repeat(np.repeat(v, 2, axis=1), 2, axis=2)[:, :H, :W]
    c, d, e = 298 * (y - 16), u - 128, v - 128
    rgb = np.stack(((c + 409 * e + 128) >> 8, (c - 100 * d - 208 * e + 128) >> 8, (c + 516 * d + 128) >> 8), axis=-1)
    return np.clip(rgb, 0, 255).astype(np.uint8)


def stream_chunks(frames: torch.Tensor, chunk_size: int, overlap: int, prepend_frames: int,
                  process: Callable[[torch.Tensor, int], torch.Tensor]) -> List[torch.Tensor]:
    """The reference's `_stream_video_chunks` on an in-memory clip [ref: inference_cli.py:566-654]:
    read `chunk_size` new frames (:601-606); prepend_frames only for the first chunk (:609-611); context = the last
    `overlap` RAW frames of the previous chunk (:613-619); process; drop the context frames from the result (:640-642);
    keep the tail of the NEW frames for the next chunk (:644-645).  `process(frames, prepend_frames)` stands for
    `_process_frames_core`.  Returns the yielded results."""
    total = frames.shape[0]
    chunk = chunk_size if chunk_size > 0 else total                        # :440
    out, read, idx, prev_tail = [], 0, 0, None
    while read < total:
        n = min(chunk, total - read)
        new = frames[read:read + n]
        read += n
        idx += 1
        prepend = prepend_frames if idx == 1 else 0
        if prev_tail is not None and overlap > 0:
            ctx = min(overlap, prev_tail.shape[0])
            cur = torch.cat([prev_tail[-ctx:], new], dim=0)
        else:
            cur, ctx = new, 0
        res = process(cur, prepend)
        if ctx > 0:
            res = res[ctx:]
        prev_tail = new[-overlap:].clone() if overlap > 0 else None
        out.append(res)
    return out
