"""Causal 3-D video VAE (s8 / t4 / 16 latent channels) - oracle restatement in plain PyTorch.

Follows /root/reference/src/models/video_vae_v3/modules/attn_video_vae.py:
  Encoder3D :671-856, Decoder3D :859-1035, ResnetBlock3D :253-362, Upsample3D :58-174,
  Downsample3D :177-250, UNetMidBlock3D :563-668, VideoAutoencoderKL(.Wrapper) :1063-1292,:1652-1690
and causal_inflation_lib.py: InflatedCausalConv3d :62-305 (temporal padding replaced by a causal head:
2*pad_t replicas of frame 0, or the previous chunk's last k_t - s_t frames), causal_norm_wrapper :354-409
(GroupNorm per frame), remove_head :412-419.

The ResnetBlock2D / Attention / Upsample2D / Downsample2D base classes come from ``diffusers`` (>=0.33.1,
not vendored, absent here) - their semantics are restated from the library ("parity unpinned");
InflatedCausalConv3d / causal_norm_wrapper / remove_head ARE pinned against the importable reference
module by oracle/pin_against_reference.py.

Temporal slicing (attn_video_vae.py:1246-1292) only bounds memory: with the carried conv state the
sliced result equals the unsliced causal computation, so ``encode``/``decode`` here run unsliced and
``*_sliced`` variants exist to demonstrate (and test) the equality.

Tensors are [C, T, H, W] (batch of one), parameters a flat dict keyed like the reference state dict.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    temporal_scale_num: int = 2
    eps: float = 1e-6
    scaling_factor: float = 0.9152


CFG = VAEConfig()


# ----------------------------------------------------------------------------- parameter layout
def param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels

    def conv(name, cin, cout, k):
        s[name + ".weight"] = (cout, cin) + k
        s[name + ".bias"] = (cout,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cin, cout, (3, 3, 3))
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, (3, 3, 3))
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, (1, 1, 1))

    def mid(name, c):
        resnet(name + ".resnets.0", c, c)
        norm(name + ".attentions.0.group_norm", c)
        for l in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{name}.attentions.0.{l}.weight"] = (c, c)
            s[f"{name}.attentions.0.{l}.bias"] = (c,)
        resnet(name + ".resnets.1", c, c)

    # encoder (attn_video_vae.py:700-806)
    conv("encoder.conv_in", cfg.in_channels, ch[0], (3, 3, 3))
    cout = ch[0]
    n = len(ch)
    for i in range(n):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != n - 1:
            tdown = i >= n - cfg.temporal_scale_num - 1
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, (3 if tdown else 1, 3, 3))
    mid("encoder.mid_block", ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", ch[-1], 2 * cfg.latent_channels, (3, 3, 3))
    # decoder (attn_video_vae.py:885-980)
    rev = list(reversed(ch))
    conv("decoder.conv_in", cfg.latent_channels, rev[0], (3, 3, 3))
    mid("decoder.mid_block", rev[0])
    cout = rev[0]
    for i in range(n):
        cin, cout = cout, rev[i]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != n - 1:
            tz = 2 if i < cfg.temporal_scale_num else 1
            conv(f"decoder.up_blocks.{i}.upsamplers.0.upscale_conv", cout, cout * 4 * tz, (1, 1, 1))
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, (3, 3, 3))
    norm("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out", ch[0], cfg.out_channels, (3, 3, 3))
    return s


def synthetic_params(cfg: VAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if ("norm" in k) and k.endswith(".weight"):
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            v = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        out[k] = v
    return out


# ----------------------------------------------------------------------------- primitives
class State:
    """Carried conv inputs for chunked (temporally sliced) evaluation; None = unsliced."""

    def __init__(self):
        self.mem: Dict[str, torch.Tensor] = {}
        self.active = False  # False: first chunk (INITIALIZING), True: later chunks (ACTIVE)


def causal_conv(p, name, x, stride=(1, 1, 1), pad_hw=(1, 1, 1, 1), st: Optional[State] = None):
    """InflatedCausalConv3d.forward (causal_inflation_lib.py:213-305).  x [C,T,H,W]."""
    w, b = p[name + ".weight"].to(x.dtype), p[name + ".bias"].to(x.dtype)
    kt = w.shape[2]
    tpad = (kt - 1) // 2  # Conv3d padding[0] given to the constructor (1 for k_t = 3, 0 for k_t = 1)
    if st is not None and st.active and (kt - stride[0]) > 0:
        head = st.mem[name]                                  # previous chunk's last k_t - s_t frames
    else:
        head = x[:, :1].expand(-1, 2 * tpad, -1, -1)         # extend_head(times = 2*pad_t), :422-437
    xin = torch.cat((head, x), dim=1) if head.shape[1] > 0 else x
    if st is not None and (kt - stride[0]) > 0:
        st.mem[name] = xin[:, xin.shape[1] - (kt - stride[0]):].clone()   # :266-278
    xin = F.pad(xin, (pad_hw[2], pad_hw[3], pad_hw[0], pad_hw[1]))
    return F.conv3d(xin[None], w, b, stride=stride)[0]


def group_norm_per_frame(p, name, x, groups, eps):
    """causal_norm_wrapper (causal_inflation_lib.py:354-409): 'b c t h w -> (b t) c h w' then GroupNorm."""
    y = F.group_norm(x.permute(1, 0, 2, 3), groups, p[name + ".weight"].to(x.dtype), p[name + ".bias"].to(x.dtype), eps)
    return y.permute(1, 0, 2, 3)


def resnet(p, name, x, cfg, st):
    """ResnetBlock3D.forward (attn_video_vae.py:311-362), temb None, output_scale_factor 1."""
    h = F.silu(group_norm_per_frame(p, name + ".norm1", x, cfg.norm_groups, cfg.eps))
    h = causal_conv(p, name + ".conv1", h, st=st)
    h = F.silu(group_norm_per_frame(p, name + ".norm2", h, cfg.norm_groups, cfg.eps))
    h = causal_conv(p, name + ".conv2", h, st=st)
    if (name + ".conv_shortcut.weight") in p:
        x = causal_conv(p, name + ".conv_shortcut", x, pad_hw=(0, 0, 0, 0), st=st)
    return x + h


def mid_attention(p, name, x, cfg):
    """diffusers Attention(heads=1, dim_head=C, GroupNorm, residual, bias) applied per frame
    (attn_video_vae.py:615-631, :656-668).  x [C,T,H,W]."""
    C, T, H, W = x.shape
    res = x
    h = group_norm_per_frame(p, name + ".group_norm", x, cfg.norm_groups, cfg.eps)
    tok = h.permute(1, 2, 3, 0).reshape(T, H * W, C)
    lin = lambda t, l: F.linear(t, p[f"{name}.{l}.weight"].to(t.dtype), p[f"{name}.{l}.bias"].to(t.dtype))
    q, k, v = lin(tok, "to_q"), lin(tok, "to_k"), lin(tok, "to_v")
    sdpa = lambda q_, k_, v_: torch.softmax((q_ @ k_.transpose(1, 2)).float() / math.sqrt(C), dim=-1).to(q_.dtype) @ v_
    if T > 1 and H * W > 16384:
        # frames attend independently: at 4K (129 600 tokens, 67 GB of fp32 scores per frame) the frames of a chunk are
        # walked one at a time - the same per-frame arithmetic, bounded memory
        a = torch.cat([sdpa(q[t:t + 1], k[t:t + 1], v[t:t + 1]) for t in range(T)], dim=0)
    else:
        a = sdpa(q, k, v)
    o = lin(a, "to_out.0").reshape(T, H, W, C).permute(3, 0, 1, 2)
    return o + res


def mid_block(p, name, x, cfg, st):
    x = resnet(p, name + ".resnets.0", x, cfg, st)
    x = mid_attention(p, name + ".attentions.0", x, cfg)
    return resnet(p, name + ".resnets.1", x, cfg, st)


def upsample(p, name, x, tz, cfg, st):
    """Upsample3D.forward (attn_video_vae.py:110-174)."""
    C, T, H, W = x.shape
    w, b = p[name + ".upscale_conv.weight"].to(x.dtype), p[name + ".upscale_conv.bias"].to(x.dtype)
    y = F.conv3d(x[None], w, b)[0]                             # [(x y z c), T, H, W]
    y = y.reshape(2, 2, tz, C, T, H, W).permute(3, 4, 2, 5, 0, 6, 1).reshape(C, T * tz, H * 2, W * 2)
    if tz == 2 and not (st is not None and st.active):
        y = torch.cat((y[:, :1], y[:, 2:]), dim=1)             # remove_head (causal_inflation_lib.py:412-419)
    return causal_conv(p, name + ".conv", y, st=st)


# ----------------------------------------------------------------------------- encoder / decoder
def encoder(p, x, cfg: VAEConfig = CFG, st: Optional[State] = None):
    """Encoder3D.forward (attn_video_vae.py:808-856).  x [3,T,H,W] -> moments [32,T_l,h,w]."""
    ch = cfg.block_out_channels
    n = len(ch)
    h = causal_conv(p, "encoder.conv_in", x, st=st)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, cfg, st)
        if i != n - 1:
            tdown = i >= n - cfg.temporal_scale_num - 1
            # Downsample3D (:229-250): downsample_padding 0 -> explicit zero pad right/bottom, stride 2
            h = causal_conv(p, f"encoder.down_blocks.{i}.downsamplers.0.conv", h,
                            stride=(2 if tdown else 1, 2, 2), pad_hw=(0, 1, 0, 1), st=st)
    h = mid_block(p, "encoder.mid_block", h, cfg, st)
    h = F.silu(group_norm_per_frame(p, "encoder.conv_norm_out", h, cfg.norm_groups, cfg.eps))
    return causal_conv(p, "encoder.conv_out", h, st=st)


def decoder(p, z, cfg: VAEConfig = CFG, st: Optional[State] = None):
    """Decoder3D.forward (attn_video_vae.py:983-1035).  z [16,T_l,h,w] -> [3,T,H,W]."""
    n = len(cfg.block_out_channels)
    h = causal_conv(p, "decoder.conv_in", z, st=st)
    h = mid_block(p, "decoder.mid_block", h, cfg, st)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, cfg, st)
        if i != n - 1:
            h = upsample(p, f"decoder.up_blocks.{i}.upsamplers.0", h, 2 if i < cfg.temporal_scale_num else 1, cfg, st)
    h = F.silu(group_norm_per_frame(p, "decoder.conv_norm_out", h, cfg.norm_groups, cfg.eps))
    return causal_conv(p, "decoder.conv_out", h, st=st)


def encode(p, x, cfg: VAEConfig = CFG, dtype=torch.float32):
    """core/infer.py:117-188: latent = posterior.mode() * scaling_factor, channels-last [T_l,h,w,16].
    x [3,T,H,W] in [-1,1], T = 4n+1."""
    pp = {k: v.to(dtype) for k, v in p.items()}
    m = encoder(pp, x.to(dtype), cfg)[: cfg.latent_channels]      # DiagonalGaussianDistribution.mode() = mean
    return (m * cfg.scaling_factor).permute(1, 2, 3, 0).contiguous()


def decode(p, latent, cfg: VAEConfig = CFG, dtype=torch.float32):
    """core/infer.py:192-257: sample = dec(latent / scaling_factor).  latent [T_l,h,w,16] -> [3,T,H,W]."""
    pp = {k: v.to(dtype) for k, v in p.items()}
    z = (latent.to(dtype) / cfg.scaling_factor).permute(3, 0, 1, 2)
    return decoder(pp, z, cfg)


def encode_sliced(p, x, cfg: VAEConfig = CFG, split: int = 4):
    """slicing_encode (attn_video_vae.py:1246-1268): first slice = frame 0 + `split`, then `split` per slice."""
    st = State()
    outs = [encoder(p, x[:, : 1 + split], cfg, st)]
    st.active = True
    for a in range(1 + split, x.shape[1], split):
        outs.append(encoder(p, x[:, a:a + split], cfg, st))
    return torch.cat(outs, dim=1)


def decode_sliced(p, z, cfg: VAEConfig = CFG, split: int = 1):
    """slicing_decode (attn_video_vae.py:1270-1292)."""
    st = State()
    outs = [decoder(p, z[:, : 1 + split], cfg, st)]
    st.active = True
    for a in range(1 + split, z.shape[1], split):
        outs.append(decoder(p, z[:, a:a + split], cfg, st))
    return torch.cat(outs, dim=1)
