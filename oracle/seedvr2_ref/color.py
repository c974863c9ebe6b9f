"""CPU restatement of the phase-4 colour correction.  TEST INFRASTRUCTURE ONLY (see __init__).

Follows /root/reference/src/utils/color_fix.py:
  calc_mean_std :72-91, adaptive_instance_normalization :94-119, wavelet_blur :122-157,
  wavelet_decomposition :160-184, wavelet_reconstruction :187-246, lab_color_transfer :249-365,
  _rgb_to_lab_batch :368-411, _lab_to_rgb_batch :414-470, _histogram_matching_channel :473-513,
and the call site src/core/generation_phases.py:1290-1304 (content = decoded sample, style = transformed input,
both [T,3,H,W] in [-1,1] in the compute dtype - bf16 on the reference's GPU/CPU path).

Every function works on [B,C,H,W] tensors in the dtype it is given (the reference's conv2d / in-place adds round to
that dtype after every op; LAB promotes to fp32 like ensure_float32_precision).  Pinned by
oracle/pin_against_reference.py (G12) against the reference module itself.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def calc_mean_std(feat: torch.Tensor, eps: float = 1e-5):
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps          # unbiased, like Tensor.var()
    return feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), var.sqrt().reshape(b, c, 1, 1)


def adaptive_instance_normalization(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    sm, ss = calc_mean_std(style)
    cm, cs = calc_mean_std(content)
    return (content - cm) / cs * ss + sm


def wavelet_blur(image: torch.Tensor, radius: int) -> torch.Tensor:
    radius = min(radius, max(1, min(image.shape[-2:]) // 8))
    c = image.shape[1]
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=image.dtype)
    k = k[None, None].repeat(c, 1, 1, 1)
    x = F.pad(image.float(), (radius,) * 4, mode="replicate").to(image.dtype)
    return F.conv2d(x, k, groups=c, dilation=radius)


def wavelet_decomposition(image: torch.Tensor, levels: int = 5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high.add_(image).sub_(low)                          # two roundings in the working dtype, like the reference
        image = low
    return high, low


def wavelet_reconstruction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    if content.shape != style.shape:
        style = F.interpolate(style.float(), size=content.shape[-2:], mode="bilinear", align_corners=False).to(style.dtype)
    ch, _ = wavelet_decomposition(content)
    _, sl = wavelet_decomposition(style)
    return ch.add_(sl).clamp_(-1.0, 1.0)


_RGB2XYZ = [[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]]
_XYZ2RGB = [[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]]
EPS = 6.0 / 29.0
KAPPA = (29.0 / 3.0) ** 3


def rgb_to_lab(rgb: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] fp32 in [0,1] -> CIELAB (D65)."""
    lin = torch.where(rgb > 0.04045, torch.pow((rgb + 0.055) / 1.055, 2.4), rgb / 12.92)
    m = torch.tensor(_RGB2XYZ, dtype=torch.float32)
    b, _, h, w = lin.shape
    xyz = torch.matmul(lin.permute(0, 2, 3, 1).reshape(-1, 3), m.T).reshape(b, h, w, 3).permute(0, 3, 1, 2).clone()
    xyz[:, 0].div_(0.95047)
    xyz[:, 2].div_(1.08883)
    f = torch.where(xyz > EPS ** 3, torch.pow(xyz, 1.0 / 3.0), xyz.mul(KAPPA).add(16.0).div(116.0))
    return torch.stack([f[:, 1] * 116.0 - 16.0, (f[:, 0] - f[:, 1]) * 500.0, (f[:, 1] - f[:, 2]) * 200.0], dim=1)


def lab_to_rgb(lab: torch.Tensor) -> torch.Tensor:
    L, a, b_ = lab[:, 0], lab[:, 1], lab[:, 2]
    fy = (L + 16.0) / 116.0
    fx = a / 500.0 + fy
    fz = fy - b_ / 200.0
    inv = lambda t: torch.where(t > EPS, torch.pow(t, 3.0), (t * 116.0 - 16.0) / KAPPA)
    xyz = torch.stack([inv(fx) * 0.95047, inv(fy), inv(fz) * 1.08883], dim=1)
    m = torch.tensor(_XYZ2RGB, dtype=torch.float32)
    b, _, h, w = xyz.shape
    lin = torch.matmul(xyz.permute(0, 2, 3, 1).reshape(-1, 3), m.T).reshape(b, h, w, 3).permute(0, 3, 1, 2)
    rgb = torch.where(lin > 0.0031308, torch.pow(torch.clamp(lin, min=0.0), 1.0 / 2.4) * 1.055 - 0.055, lin * 12.92)
    return torch.clamp(rgb, 0.0, 1.0)


def histogram_match(source: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """The i-th smallest source element receives the i-th smallest reference element (equal sizes: the
    reference's quantile mapping degenerates to this); unequal sizes: reference quantile floor((n_ref-1)*i/(n-1)).
    Tied source values (frequent: the [0,1] clamp produces exact 0s and 1s) receive neighbouring reference quantiles in
    the order torch.sort happens to give them - the reference does not ask for a stable sort, so which tied pixel gets
    which of those quantiles is unspecified; parity tests compare tied groups as sets."""
    s = source.flatten()
    r, _ = torch.sort(reference.flatten())
    _, idx = torch.sort(s)
    if r.numel() != s.numel():
        q = torch.linspace(0, 1, s.numel())
        r = r[(q * (r.numel() - 1)).long().clamp_(0, r.numel() - 1)]
    out = torch.empty_like(s)
    out.scatter_(0, idx, r)
    return out.reshape(source.shape)


def lab_color_transfer(content: torch.Tensor, style: torch.Tensor, luminance_weight: float = 0.8) -> torch.Tensor:
    dtype = content.dtype
    content = wavelet_reconstruction(content, style)
    c = content.float().add(1.0).mul(0.5).clamp(0.0, 1.0)
    s = style.float().add(1.0).mul(0.5).clamp(0.0, 1.0)
    cl, sl = rgb_to_lab(c), rgb_to_lab(s)
    a = histogram_match(cl[:, 1], sl[:, 1])
    b = histogram_match(cl[:, 2], sl[:, 2])
    L = cl[:, 0]
    if luminance_weight < 1.0:
        L = cl[:, 0] * luminance_weight + histogram_match(cl[:, 0], sl[:, 0]) * (1.0 - luminance_weight)
    return lab_to_rgb(torch.stack([L, a, b], dim=1)).mul(2.0).sub(1.0).to(dtype)


METHODS = {"lab": lab_color_transfer, "wavelet": wavelet_reconstruction, "adain": adaptive_instance_normalization}


def color_correct(sample_thwc: torch.Tensor, input_thwc: torch.Tensor, method: str) -> torch.Tensor:
    """[T,H,W,3] in [-1,1] (decoded batch, transformed input of the same frames) -> corrected [T,H,W,3]."""
    fn = METHODS[method]
    out = fn(sample_thwc.permute(0, 3, 1, 2).contiguous(), input_thwc.permute(0, 3, 1, 2).contiguous())
    return out.permute(0, 2, 3, 1).contiguous()
