"""CPU restatement of the phase-4 colour correction.  TEST INFRASTRUCTURE ONLY (see __init__).

Follows /root/reference/src/utils/color_fix.py:
  calc_mean_std :72-91, adaptive_instance_normalization :94-119, wavelet_blur :122-157,
  wavelet_decomposition :160-184, wavelet_reconstruction :187-246, lab_color_transfer :249-365,
  _rgb_to_lab_batch :368-411, _lab_to_rgb_batch :414-470, _histogram_matching_channel :473-513,
  hsv_saturation_histogram_match :516-603, _rgb_to_hsv_batch :606-641, _hsv_to_rgb_batch :644-687,
  _hue_conditional_saturation_match :690-733, _histogram_match_1d :736-757,
  wavelet_adaptive_color_correction :760-843, _get_saturation_map :846-860,
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


def rgb_to_hsv(rgb: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] fp32 in [0,1] -> (h, s, v), all in [0,1]  [ref: color_fix.py:606-641]."""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    maxc, minc = torch.max(rgb, dim=1)[0], torch.min(rgb, dim=1)[0]
    rng = maxc - minc
    nz = torch.where(rng > 1e-10, rng, torch.ones_like(rng))
    h = torch.zeros_like(maxc)
    m = (maxc == r) & (rng > 1e-10)
    h[m] = ((g[m] - b[m]) / nz[m]) % 6.0
    m = (maxc == g) & (rng > 1e-10)
    h[m] = ((b[m] - r[m]) / nz[m]) + 2.0
    m = (maxc == b) & (rng > 1e-10)
    h[m] = ((r[m] - g[m]) / nz[m]) + 4.0
    h.div_(6.0)
    s = torch.where(maxc > 1e-10, rng / torch.clamp(maxc, min=1e-10), torch.zeros_like(maxc))
    return torch.stack([h, s, maxc], dim=1)


def hsv_to_rgb(hsv: torch.Tensor) -> torch.Tensor:
    """[ref: color_fix.py:644-687]"""
    h, s, v = hsv[:, 0].mul(6.0), hsv[:, 1], hsv[:, 2]
    i = torch.floor(h).long() % 6
    f = h - torch.floor(h)
    p, q, t = v * (1.0 - s), v * (1.0 - s * f), v * (1.0 - s * (1.0 - f))
    r, g, b = torch.zeros_like(v), torch.zeros_like(v), torch.zeros_like(v)
    for k, (rr, gg, bb) in enumerate(((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))):
        m = i == k
        r[m], g[m], b[m] = rr[m], gg[m], bb[m]
    return torch.stack([r, g, b], dim=1)


def histogram_match_1d(source: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """[ref: color_fix.py:736-757]; unequal sizes: reference quantile long(linspace(0,1,n)[i] * (n_ref - 1))."""
    ss, si = torch.sort(source)
    rs, _ = torch.sort(reference)
    if ss.numel() != rs.numel():
        q = torch.linspace(0, 1, ss.numel())
        rs = rs[(q * (rs.numel() - 1)).long().clamp_(0, rs.numel() - 1)]
    out = torch.empty_like(source)
    out.scatter_(0, si, rs)
    return out


def hue_conditional_saturation_match(ch, cs, sh, ss_, num_bins: int = 12, min_pixels: int = 100) -> torch.Tensor:
    """Saturation histogram match per 30-degree hue bin; bin 0 also takes the wrap-around hues >= 11/12 (which bin 11
    then overwrites) [ref: color_fix.py:690-733]."""
    bw = 1.0 / num_bins
    out = cs.clone()
    for k in range(num_bins):
        lo, hi = k * bw, (k + 1) * bw
        if k == 0:
            cm = ((ch >= 0) & (ch < hi)) | (ch >= (1.0 - bw))
            sm = ((sh >= 0) & (sh < hi)) | (sh >= (1.0 - bw))
        else:
            cm, sm = (ch >= lo) & (ch < hi), (sh >= lo) & (sh < hi)
        a, b = cs[cm], ss_[sm]
        if len(a) > min_pixels and len(b) > min_pixels:
            out[cm] = histogram_match_1d(a, b)
    return out


def hsv_saturation_histogram_match(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    """[ref: color_fix.py:516-603]"""
    dtype = content.dtype
    c = rgb_to_hsv(content.float().add(1.0).mul(0.5).clamp(0.0, 1.0))
    s = rgb_to_hsv(style.float().add(1.0).mul(0.5).clamp(0.0, 1.0))
    ms = hue_conditional_saturation_match(c[:, 0], c[:, 1], s[:, 0], s[:, 1])
    rgb = hsv_to_rgb(torch.stack([c[:, 0], ms, c[:, 2]], dim=1)).clamp_(0.0, 1.0)
    return rgb.mul(2.0).sub(1.0).to(dtype)


def saturation_map(x: torch.Tensor) -> torch.Tensor:
    """[ref: color_fix.py:846-860]"""
    rgb = x.add(1.0).mul(0.5).clamp(0.0, 1.0)
    maxc, minc = torch.max(rgb, dim=1, keepdim=True)[0], torch.min(rgb, dim=1, keepdim=True)[0]
    return torch.where(maxc > 1e-10, (maxc - minc) / torch.clamp(maxc, min=1e-10), torch.zeros_like(maxc))


def wavelet_adaptive_color_correction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    """Wavelet base, HSV saturation correction blended into the regions that are (and stay) oversaturated; everything
    in fp32 (the reference promotes BEFORE the wavelet here, unlike 'lab')  [ref: color_fix.py:760-843]."""
    dtype = content.dtype
    c, s = content.float(), style.float()
    wav = wavelet_reconstruction(c.clone(), s.clone())
    hsv = hsv_saturation_histogram_match(c, s)
    cs, ss_, ws = saturation_map(c), saturation_map(s), saturation_map(wav)
    w = torch.sigmoid(5.0 * ((cs - ss_) - 0.15))
    w = (w * ((ws - ss_) > 0.075).float()).clamp_(0.0, 1.0)
    return (wav * (1.0 - w) + hsv * w).to(dtype)


METHODS = {"lab": lab_color_transfer, "wavelet": wavelet_reconstruction, "adain": adaptive_instance_normalization,
           "hsv": hsv_saturation_histogram_match, "wavelet_adaptive": wavelet_adaptive_color_correction}


def color_correct(sample_thwc: torch.Tensor, input_thwc: torch.Tensor, method: str) -> torch.Tensor:
    """[T,H,W,3] in [-1,1] (decoded batch, transformed input of the same frames) -> corrected [T,H,W,3]."""
    fn = METHODS[method]
    out = fn(sample_thwc.permute(0, 3, 1, 2).contiguous(), input_thwc.permute(0, 3, 1, 2).contiguous())
    return out.permute(0, 2, 3, 1).contiguous()
