"""CPU restatement of the reference's edge-guided alpha upscaling (TEST INFRASTRUCTURE ONLY)
[ref: src/core/alpha_upscaling.py:124-430; call site src/core/generation_phases.py:1132-1190].

Pinned by oracle/pin_against_reference.py (G14) against `edge_guided_alpha_upscale` itself, imported from the reference
with a `cv2` stand-in whose two functions are `rgb_to_gray_u8` and `sobel_f64` below.  Those two restate OpenCV's
documented 8-bit arithmetic (cv2 is absent from this image and from the reference tree): PARITY UNPINNED for them,
everything around them (alpha statistics, antialiased bicubic resize, guided filter, the binary-mask refinement) is
pinned.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def rgb_to_gray_u8(frame: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(frame, COLOR_RGB2GRAY) on uint8 [H,W,3]: 14-bit fixed point, (4899 R + 9617 G + 1868 B + 8192) >> 14."""
    f = frame.astype(np.int32)
    return ((4899 * f[..., 0] + 9617 * f[..., 1] + 1868 * f[..., 2] + (1 << 13)) >> 14).astype(np.uint8)


def sobel_f64(gray: np.ndarray, dx: int, dy: int) -> np.ndarray:
    """cv2.Sobel(gray, CV_64F, dx, dy, ksize=3) with the default BORDER_REFLECT_101: [1 2 1] smoothing across,
    [-1 0 1] along the derivative axis (exact integers)."""
    g = np.pad(gray.astype(np.float64), 1, mode="reflect")
    if dx == 1 and dy == 0:
        d = g[:, 2:] - g[:, :-2]
        return d[:-2] + 2.0 * d[1:-1] + d[2:]
    if dx == 0 and dy == 1:
        d = g[2:, :] - g[:-2, :]
        return d[:, :-2] + 2.0 * d[:, 1:-1] + d[:, 2:]
    raise ValueError("only first derivatives")


def detect_edges(images: torch.Tensor) -> torch.Tensor:
    """detect_edges_batch(method='sobel') (:124-189): [T,3,H,W] fp32 -> [T,1,H,W] fp32 in [0,1]; per-frame maximum
    normalisation and uint8 truncation.  A frame without any gradient divides 0/0 in the reference; numpy's cast of NaN
    to uint8 gives 0 on x86, which is what this returns."""
    x = images.float().cpu().numpy()
    if x.min() < 0:
        x = (x + 1) / 2
    u8 = (x * 255).clip(0, 255).astype(np.uint8)
    out = []
    for t in range(u8.shape[0]):
        gray = rgb_to_gray_u8(u8[t].transpose(1, 2, 0))
        sx, sy = sobel_f64(gray, 1, 0), sobel_f64(gray, 0, 1)
        e = np.sqrt(sx ** 2 + sy ** 2)
        m = e.max()
        out.append((e / m * 255).astype(np.uint8) if m > 0 else np.zeros_like(gray))
    return (torch.from_numpy(np.stack(out)).float() / 255.0).unsqueeze(1)


def box_filter(x: torch.Tensor, r: int) -> torch.Tensor:
    return F.avg_pool2d(x, kernel_size=2 * r + 1, stride=1, padding=r)


def guided_filter(guide_rgb: torch.Tensor, src: torch.Tensor, radius: int, eps: float) -> torch.Tensor:
    """guided_filter_pytorch + _apply_guided_filter (:192-290), fp32."""
    g = guide_rgb.float().mean(dim=1, keepdim=True)
    src = src.float()
    mean_g, mean_s = box_filter(g, radius), box_filter(src, radius)
    corr_g, corr_gs = box_filter(g * g, radius), box_filter(g * src, radius)
    var_g = corr_g - mean_g * mean_g
    cov = corr_gs - mean_g * mean_s
    a = cov / (var_g + eps)
    b = mean_s - a * mean_g
    return box_filter(a, radius) * g + box_filter(b, radius)


def edge_guided_alpha_upscale(input_alpha: torch.Tensor, upscaled_rgb: torch.Tensor) -> torch.Tensor:
    """(:293-430) input_alpha [T,1,h,w] in [0,1], upscaled_rgb [T,3,H,W] in [-1,1] or [0,1] -> alpha [T,1,H,W] fp32."""
    _, _, H, W = upscaled_rgb.shape
    alpha = input_alpha.float()
    rgb = upscaled_rgb.float()
    flat = alpha.flatten()
    ratio = ((flat < 0.1).sum().float() + (flat > 0.9).sum().float()) / flat.numel()
    is_binary = bool(ratio > 0.95)
    rgbn = rgb.clone()
    if rgbn.min() < 0:
        rgbn = (rgbn + 1) / 2
    edges = detect_edges(rgbn)
    up = F.interpolate(alpha, size=(H, W), mode="bicubic", align_corners=False, antialias=True).clamp(0, 1)
    if not is_binary:
        return guided_filter(rgbn, up.clone(), 3, 0.002).clamp(0, 1)
    refined = guided_filter(rgbn, up, 2, 0.002)
    zone = F.max_pool2d(edges, kernel_size=3, stride=1, padding=1)
    is_solid = zone < 0.05
    binary = (refined > 0.5).float()
    enhanced = torch.sigmoid((refined - 0.5) * 12.0)
    strength = torch.clamp(edges / 0.25, 0, 1)
    in_edges = refined * (1 - strength) + enhanced * strength
    combined = torch.where(is_solid, binary, in_edges)
    final = torch.where(zone < 0.03, (combined > 0.5).float(), combined)
    mid = (final > 0.3) & (final < 0.7)
    final = torch.where(mid & ~(edges > 0.15), (final > 0.5).float(), final)
    return final.clamp(0, 1)
