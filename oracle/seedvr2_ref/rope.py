"""Rotary position embedding used by NaDiT (oracle restatement).

The arithmetic lives in the third-party package ``rotary_embedding_torch`` (>=0.5.3,
requirements.txt:11; not vendored, absent from this image) - "parity unpinned".
Restated from its published algorithm, anchored on the reference call sites:
  * 3B: /root/reference/src/models/dit_3b/rope.py:74-85 (lang freqs, dim=128//3=42,
    theta=1e4), :118-126 (fp32 rotation), :158-175 (window-local video coordinates
    offset by the text length on the temporal axis; text token p at (p,p,p)).
  * 7B: /root/reference/src/models/dit_7b/rope.py:25-47 (pixel freqs, dim=64//3=21,
    max_freq=256), positions linspace(-1,1,n) per axis; video only.
"""

from __future__ import annotations

import math

import torch


def lang_freqs(dim: int, theta: float = 10000.0) -> torch.Tensor:
    """RotaryEmbedding(freqs_for='lang'): 1/theta^(arange(0,dim,2)[:dim//2]/dim)."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def pixel_freqs(dim: int, max_freq: float = 256.0) -> torch.Tensor:
    """RotaryEmbedding(freqs_for='pixel'): linspace(1, max_freq/2, dim//2) * pi."""
    return torch.linspace(1.0, max_freq / 2, dim // 2) * math.pi


def axis_angles(pos: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """outer(pos, freqs) with every frequency repeated twice ('... n -> ... (n r)', r=2)."""
    ang = pos.float()[:, None] * freqs[None, :]
    return ang.repeat_interleave(2, dim=-1)


def rotate_interleaved(x: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb: x[..., :rot] = x*cos + rotate_half(x)*sin, pairs (2i, 2i+1),
    rotate_half: (x1, x2) -> (-x2, x1); dims >= rot pass through.  fp32 in, fp32 out."""
    rot = angles.shape[-1]
    xm, xr = x[..., :rot], x[..., rot:]
    x1, x2 = xm[..., 0::2], xm[..., 1::2]
    rh = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return torch.cat((xm * angles.cos() + rh * angles.sin(), xr), dim=-1)


def rotate_interleaved_cs(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Same rotation with precomputed (possibly low-precision) cos/sin tables."""
    rot = cos.shape[-1]
    xm, xr = x[..., :rot], x[..., rot:]
    x1, x2 = xm[..., 0::2], xm[..., 1::2]
    rh = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return torch.cat((xm * cos + rh * sin, xr), dim=-1)


def mm_window_angles_3b(shape_thw, txt_len: int, head_dim: int = 128):
    """Angles for one window of extent (f,h,w) and for the text tokens (3B mmrope3d).

    dit_3b/rope.py:171-173: vid = axial[l:l+f, :h, :w], txt = axial1d[:l].repeat(1,3).
    Returns (vid [f*h*w, 126], txt [l, 126]) in fp32.
    """
    f, h, w = (int(v) for v in shape_thw)
    fr = lang_freqs(head_dim // 3)
    at = axis_angles(torch.arange(txt_len, txt_len + f), fr)  # [f, 42]
    ah = axis_angles(torch.arange(h), fr)
    aw = axis_angles(torch.arange(w), fr)
    n = fr.numel() * 2
    vid = torch.cat(
        (
            at[:, None, None, :].expand(f, h, w, n),
            ah[None, :, None, :].expand(f, h, w, n),
            aw[None, None, :, :].expand(f, h, w, n),
        ),
        dim=-1,
    ).reshape(f * h * w, 3 * n)
    a1 = axis_angles(torch.arange(txt_len), fr)
    txt = a1.repeat(1, 3)
    return vid, txt


def window_angles_7b(shape_thw, head_dim: int = 128):
    """7B: RotaryEmbedding(dim=(head_dim//2)//3=21, 'pixel', max_freq=256) -> 10 freqs,
    positions linspace(-1,1,n) per window axis, 60 rotated dims (dit_7b/rope.py:25-47,
    :100-111; rope_dim defaults to head_dim//2, dit_7b/nadit.py)."""
    f, h, w = (int(v) for v in shape_thw)
    fr = pixel_freqs((head_dim // 2) // 3)
    at = axis_angles(torch.linspace(-1, 1, f), fr)
    ah = axis_angles(torch.linspace(-1, 1, h), fr)
    aw = axis_angles(torch.linspace(-1, 1, w), fr)
    n = fr.numel() * 2
    return torch.cat(
        (
            at[:, None, None, :].expand(f, h, w, n),
            ah[None, :, None, :].expand(f, h, w, n),
            aw[None, None, :, :].expand(f, h, w, n),
        ),
        dim=-1,
    ).reshape(f * h * w, 3 * n)
