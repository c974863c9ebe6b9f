"""The oracle's VAE decoder (oracle/seedvr2_ref/vae.py, unchanged code) with its library calls served on PIECES - for the
one at-size case that the plain calls do not survive: 5 frames of 4K through the decoder in one piece end in a crash of the
CPU convolution library (core dump after ~7 minutes; the 4K single-frame case, whose largest conv input has 6.4 G elements, runs).
Test infrastructure only.

What changes is the size of the tensors handed to `F.conv3d` / `F.group_norm`, not the ops:
  conv3d: one output frame x one strip of output rows at a time (input = the k_t frames and the k_h - 1 halo rows it needs);
  group_norm: one frame at a time (the oracle's norm is per frame anyway: causal_norm_wrapper folds t into the batch);
and the decoder runs one latent frame per temporal chunk with the oracle's own carried state (`State`), which the
reference's tests pin as result-neutral (sliced == unsliced).  tests/test_cpu_pipeline.py checks, with tiny piece limits, that
this executor reproduces the plain oracle to fp32 rounding.
"""

from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as RealF


class _PiecewiseF:
    """Drop-in for `torch.nn.functional` inside the oracle's vae module."""

    def __init__(self, max_elems: int):
        self.max_elems = max_elems

    def __getattr__(self, name):
        return getattr(RealF, name)

    def conv3d(self, x, w, b=None, stride=1):
        stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        N, C, T, H, W = x.shape
        kt, kh, kw = w.shape[2:]
        if x.numel() <= self.max_elems and w.shape[0] * T * H * W <= self.max_elems:
            return RealF.conv3d(x, w, b, stride=stride)
        To, Ho, Wo = (T - kt) // stride[0] + 1, (H - kh) // stride[1] + 1, (W - kw) // stride[2] + 1
        out = torch.empty(N, w.shape[0], To, Ho, Wo, dtype=x.dtype, device=x.device)
        per_row = max(C * kt * W, w.shape[0] * Wo)
        rows = max(1, min(Ho, self.max_elems // per_row - kh))
        for t in range(To):
            xt = x[:, :, t * stride[0]: t * stride[0] + kt]
            for y0 in range(0, Ho, rows):
                y1 = min(Ho, y0 + rows)
                piece = xt[:, :, :, y0 * stride[1]: (y1 - 1) * stride[1] + kh]
                out[:, :, t:t + 1, y0:y1] = RealF.conv3d(piece.contiguous(), w, b, stride=stride)
        return out

    def group_norm(self, x, groups, weight=None, bias=None, eps=1e-5):
        if x.numel() <= self.max_elems or x.shape[0] == 1:
            return RealF.group_norm(x, groups, weight, bias, eps)
        return torch.cat([RealF.group_norm(x[i:i + 1], groups, weight, bias, eps) for i in range(x.shape[0])], dim=0)


@contextlib.contextmanager
def piecewise(ovae, max_elems: int = 1 << 30):
    keep = ovae.F
    ovae.F = _PiecewiseF(max_elems)
    try:
        yield
    finally:
        ovae.F = keep


def decode_chunked(p, latent, ovae, dtype=torch.float32, max_elems: int = 1 << 30):
    """ovae.decode with one latent frame per temporal chunk and piecewise library calls.  latent [T_l,h,w,16] ->
    [T,H,W,3] (same device / dtype as the computation)."""
    cfg = ovae.CFG
    pp = {k: v.to(dtype) for k, v in p.items()}
    z = (latent.to(dtype) / cfg.scaling_factor).permute(3, 0, 1, 2)
    st = ovae.State()
    outs = []
    with piecewise(ovae, max_elems):
        for t in range(z.shape[1]):
            outs.append(ovae.decoder(pp, z[:, t:t + 1], cfg, st))
            st.active = True
    return torch.cat(outs, dim=1).permute(1, 2, 3, 0).contiguous()


def encode_piecewise(p, x, ovae, dtype=torch.float32, max_elems: int = 1 << 30):
    """ovae.encode (unsliced) with piecewise library calls.  x [3,T,H,W] in [-1,1] -> latent [T_l,h,w,16]."""
    cfg = ovae.CFG
    pp = {k: v.to(dtype) for k, v in p.items()}
    with piecewise(ovae, max_elems):
        m = ovae.encoder(pp, x.to(dtype), cfg)[: cfg.latent_channels]
    return (m * cfg.scaling_factor).permute(1, 2, 3, 0).contiguous()
