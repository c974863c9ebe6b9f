"""NaDiT (3B flavour, parametrised) forward pass - oracle restatement in plain PyTorch.

Follows, for a single sample and one positive text branch (cfg scale 1):
  * /root/reference/src/models/dit_3b/nadit.py:190-248          NaDiT.forward
  * .../dit_3b/nablocks/mmsr_block.py:84-128                     block: norm -> ada -> attn/mlp -> gate -> residual
  * .../dit_3b/nablocks/attention/mmattn.py:161-271              window joint attention
  * .../dit_3b/attention.py:24-61                                per-window SDPA
  * .../dit_3b/modulation.py:65-118 + common/cache.py:18-47      AdaSingle incl. the cache-key collision of vid_out_ada
  * .../dit_3b/normalization.py:88-109                           RMSNorm (eps inside sqrt)
  * .../dit_3b/mlp.py:46-62, mm.py:35-74, patch/patch_v1.py:76-127, embedding.py:39-62

Parameters are a flat dict keyed like the reference's state_dict (so checkpoint
ingestion can reuse it).  ``dtype`` selects the arithmetic: float32 (ground truth)
or bfloat16 (every op rounds to bf16 like the reference's CPU path,
optimization/compatibility.py:207-221); RoPE always rotates in fp32 (rope.py:118-126).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import rope as _rope
from . import windows as _win


@dataclass
class DiTConfig:
    vid_in_channels: int = 33
    vid_out_channels: int = 16
    vid_dim: int = 2560
    txt_in_dim: int = 5120
    heads: int = 20
    head_dim: int = 128
    expand_ratio: int = 4
    num_layers: int = 32
    mm_layers: int = 10
    norm_eps: float = 1e-5
    patch: Tuple[int, int, int] = (1, 2, 2)
    window: Tuple[int, int, int] = (4, 3, 3)
    mlp_multiple_of: int = 256
    # 7B deltas (dit_7b/nadit.py:39-190, configs_7b/main.yaml): GELU MLP with biases, no vid_out_norm/ada,
    # no vid-only last layer, all layers with separate vid/txt weights (mm_layers = num_layers)
    mlp_type: str = "swiglu"       # "swiglu" (3B, mlp.py:46-62) | "normal" (7B, dit_7b/mlp.py:28-43)
    out_norm: bool = True          # vid_out_norm + vid_out_ada (3B only, nadit.py:167-179)
    last_vid_only: bool = True     # 3B: last layer's mlp/ada are vid_only (mmsr_block.py:73-82)

    @property
    def emb_dim(self) -> int:
        return 6 * self.vid_dim

    @property
    def mlp_hidden(self) -> int:
        if self.mlp_type == "normal":
            return self.vid_dim * self.expand_ratio  # dit_7b/mlp.py:34
        # mlp.py:54-55
        hid = int(2 * self.vid_dim * self.expand_ratio / 3)
        m = self.mlp_multiple_of
        return m * ((hid + m - 1) // m)

    def window_method(self, layer: int) -> str:
        # configs_3b/main.yaml:34 - even layers regular, odd layers shifted
        return _win.REGULAR if layer % 2 == 0 else _win.SHIFTED

    def shared(self, layer: int) -> bool:
        # nadit.py:149-151
        return not (layer < self.mm_layers)


CFG_3B = DiTConfig()
CFG_7B = DiTConfig(vid_dim=3072, heads=24, num_layers=36, mm_layers=36, mlp_type="normal", out_norm=False,
                   last_vid_only=False)


def param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of the reference NaDiT-3B (names as produced by nn.Module nesting)."""
    d, hd, H = cfg.vid_dim, cfg.head_dim, cfg.heads
    pt, ph, pw = cfg.patch
    s: Dict[str, Tuple[int, ...]] = {
        "vid_in.proj.weight": (d, cfg.vid_in_channels * pt * ph * pw),
        "vid_in.proj.bias": (d,),
        "txt_in.weight": (d, cfg.txt_in_dim),
        "txt_in.bias": (d,),
        "emb_in.proj_in.weight": (d, 256),
        "emb_in.proj_in.bias": (d,),
        "emb_in.proj_hid.weight": (d, d),
        "emb_in.proj_hid.bias": (d,),
        "emb_in.proj_out.weight": (cfg.emb_dim, d),
        "emb_in.proj_out.bias": (cfg.emb_dim,),
        "vid_out.proj.weight": (cfg.vid_out_channels * pt * ph * pw, d),
        "vid_out.proj.bias": (cfg.vid_out_channels * pt * ph * pw,),
    }
    if cfg.out_norm:
        s.update({"vid_out_norm.weight": (d,), "vid_out_ada.out_shift": (d,), "vid_out_ada.out_scale": (d,)})
    for i in range(cfg.num_layers):
        last = cfg.last_vid_only and i == cfg.num_layers - 1
        branches = ("all",) if cfg.shared(i) else ("vid", "txt")
        p = f"blocks.{i}."
        for b in branches:
            s[p + f"attn.proj_qkv.{b}.weight"] = (3 * H * hd, d)
            s[p + f"attn.proj_out.{b}.weight"] = (d, H * hd)
            s[p + f"attn.proj_out.{b}.bias"] = (d,)
            s[p + f"attn.norm_q.{b}.weight"] = (hd,)
            s[p + f"attn.norm_k.{b}.weight"] = (hd,)
            if last and b == "txt":
                continue  # vid_only modules have no txt copy (mm.py:52-57)
            if cfg.mlp_type == "swiglu":
                s[p + f"mlp.{b}.proj_in_gate.weight"] = (cfg.mlp_hidden, d)
                s[p + f"mlp.{b}.proj_in.weight"] = (cfg.mlp_hidden, d)
                s[p + f"mlp.{b}.proj_out.weight"] = (d, cfg.mlp_hidden)
            else:
                s[p + f"mlp.{b}.proj_in.weight"] = (cfg.mlp_hidden, d)
                s[p + f"mlp.{b}.proj_in.bias"] = (cfg.mlp_hidden,)
                s[p + f"mlp.{b}.proj_out.weight"] = (d, cfg.mlp_hidden)
                s[p + f"mlp.{b}.proj_out.bias"] = (d,)
            for l in ("attn", "mlp"):
                for g in ("shift", "scale", "gate"):
                    s[p + f"ada.{b}.{l}_{g}"] = (d,)
    return s


def synthetic_params(cfg: DiTConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no checkpoint exists offline): fan-in scaled matrices,
    norm weights ~1, ada scale ~1 (modulation.py:56-61 init), small biases."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if "norm_q." in k or "norm_k." in k or k == "vid_out_norm.weight":
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("_scale"):
            v = 1.0 + torch.randn(shp, generator=g) / math.sqrt(shp[0])
        elif k.endswith("_shift"):
            v = torch.randn(shp, generator=g) / math.sqrt(shp[0])
        elif k.endswith("_gate"):
            v = 0.5 + torch.randn(shp, generator=g) / math.sqrt(shp[0])  # keep residual branches alive
        elif k.endswith(".bias"):
            v = 0.02 * torch.randn(shp, generator=g)
        else:
            v = torch.randn(shp, generator=g) / math.sqrt(shp[1])  # fan-in scaling keeps activations O(1)
        out[k] = v
    return out


# ----------------------------------------------------------------------------- pieces

def rms_norm(x: torch.Tensor, eps: float, weight: torch.Tensor | None = None) -> torch.Tensor:
    """normalization.py:88-109 - x / sqrt(mean(x^2) + eps) [* w], in x's dtype op by op."""
    var = x.pow(2).mean(dim=-1, keepdim=True)
    y = x / torch.sqrt(var + eps)
    return y if weight is None else y * weight.to(y.dtype)


def timestep_embedding(t: float, dim: int = 256) -> torch.Tensor:
    """diffusers get_timestep_embedding(t, 256, flip_sin_to_cos=False, downscale_freq_shift=0)
    (absent dependency, restated): [sin(t f_i), cos(t f_i)], f_i = exp(-ln(1e4) i/half)."""
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = torch.tensor([float(t)], dtype=torch.float32)[:, None] * f[None]
    return torch.cat((a.sin(), a.cos()), dim=-1)


def time_embedding(p, t: float, dtype) -> torch.Tensor:
    """embedding.py:39-62."""
    e = timestep_embedding(t).to(dtype)
    e = F.silu(F.linear(e, p["emb_in.proj_in.weight"].to(dtype), p["emb_in.proj_in.bias"].to(dtype)))
    e = F.silu(F.linear(e, p["emb_in.proj_hid.weight"].to(dtype), p["emb_in.proj_hid.bias"].to(dtype)))
    return F.linear(e, p["emb_in.proj_out.weight"].to(dtype), p["emb_in.proj_out.bias"].to(dtype))  # [1, 6d]


def ada_slices(emb: torch.Tensor, layer_idx: int, n_layers: int = 2):
    """modulation.py:76 - 'b (d l g) -> b d l g' then [..., idx, :] -> shift, scale, gate ([d] each)."""
    e = emb.reshape(-1, n_layers, 3)  # [d, l, g]
    return e[:, layer_idx, 0], e[:, layer_idx, 1], e[:, layer_idx, 2]


def patch_in(vid: torch.Tensor, thw, patch) -> torch.Tensor:
    """patch_v1.py:86-93 - '(T t)(H h)(W w) c -> T H W (t h w c)' for t=1 (c fastest)."""
    T, H, W = thw
    pt, ph, pw = patch
    assert pt == 1
    c = vid.shape[-1]
    x = vid.reshape(T, H // ph, ph, W // pw, pw, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(T * (H // ph) * (W // pw), ph * pw * c)


def patch_out(x: torch.Tensor, grid, patch, c: int) -> torch.Tensor:
    """patch_v1.py:119-124 - 'T H W (t h w c) -> (T t)(H h)(W w) c'."""
    T, H, W = grid
    pt, ph, pw = patch
    y = x.reshape(T, H, W, ph, pw, c).permute(0, 1, 3, 2, 4, 5)
    return y.reshape(T * H * ph * W * pw, c)


def window_attention(q, k, v, tq, tk, tv, grid, boxes, txt_len, cfg: DiTConfig, nq_v, nk_v, nq_t, nk_t, dtype,
                     rope_mode: str = "3b"):
    """mmattn.py:195-264.  q,k,v: [L, H, D] video tokens (natural order); tq,tk,tv: [Lt, H, D].
    Returns (vid_out [L, H*D], txt_out [Lt, H*D])."""
    L, H, D = q.shape
    idx, lens, shapes = _win.window_token_index(grid, boxes)
    idx_t = torch.from_numpy(idx)
    eps = cfg.norm_eps
    # per-head RMSNorm with weight (mmattn.py:207-208)
    q = rms_norm(q, eps, nq_v)
    k = rms_norm(k, eps, nk_v)
    tq = rms_norm(tq, eps, nq_t)
    tk = rms_norm(tk, eps, nk_t)
    vid_out = torch.zeros(L, H * D, dtype=dtype)
    txt_acc = []
    off = 0
    for n, shp in zip(lens.tolist(), shapes.tolist()):
        sel = idx_t[off:off + n]
        off += n
        qw, kw, vw = q[sel], k[sel], v[sel]
        tqw, tkw = tq, tk
        if rope_mode == "3b":
            av, at = _rope.mm_window_angles_3b(shp, txt_len, D)
            qw = _rope.rotate_interleaved(qw.float(), av[:, None, :]).to(dtype)
            kw = _rope.rotate_interleaved(kw.float(), av[:, None, :]).to(dtype)
            tqw = _rope.rotate_interleaved(tq.float(), at[:, None, :]).to(dtype)
            tkw = _rope.rotate_interleaved(tk.float(), at[:, None, :]).to(dtype)
        elif rope_mode == "7b":
            # dit_7b/rope.py:88-89: freqs are cast to q's dtype before apply_rotary_emb, so under bf16 the
            # ANGLES (and their cos/sin) are bf16-quantised; the rotation itself runs on q.float()
            av = _rope.window_angles_7b(shp, D).to(dtype)
            cs, sn = av.cos(), av.sin()
            qw = _rope.rotate_interleaved_cs(qw.float(), cs[:, None, :], sn[:, None, :]).to(dtype)
            kw = _rope.rotate_interleaved_cs(kw.float(), cs[:, None, :], sn[:, None, :]).to(dtype)
        # sequence = [window video tokens ; all text tokens] (na.py:386-393)
        qs = torch.cat((qw, tqw)).permute(1, 0, 2)  # [H, S, D]
        ks = torch.cat((kw, tkw)).permute(1, 0, 2)
        vs = torch.cat((vw, tv)).permute(1, 0, 2)
        # attention.py:42-47 - SDPA, non-causal, scale 1/sqrt(D)
        s = (qs @ ks.transpose(1, 2)) / math.sqrt(D)
        pr = torch.softmax(s.float(), dim=-1).to(dtype)
        o = (pr @ vs).permute(1, 0, 2).reshape(n + txt_len, H * D)
        vid_out[sel] = o[:n]
        txt_acc.append(o[n:])
    # na.py:414 - text outputs averaged over the windows
    txt_out = torch.stack(txt_acc, dim=1).mean(1)
    return vid_out, txt_out


def block(p, cfg: DiTConfig, i: int, x, y, emb, grid, dtype=torch.float32, rope_mode: str = "3b",
          method: str | None = None):
    """One NaMMSRTransformerBlock (mmsr_block.py:84-128).  x [L,d] video tokens (natural order),
    y [Lt,d] text tokens, emb [6d].  Returns (x_out, y_out, extras)."""
    H, D = cfg.heads, cfg.head_dim
    W_ = lambda k: p[k].to(dtype)
    Lt = y.shape[0]
    last = cfg.last_vid_only and i == cfg.num_layers - 1
    bv, bt = ("all", "all") if cfg.shared(i) else ("vid", "txt")
    pre = f"blocks.{i}."
    boxes = _win.windows_for(method or cfg.window_method(i), grid, cfg.window)

    def ada(h, branch, layer, mode):
        sh, sc, ga = ada_slices(emb, 0 if layer == "attn" else 1)
        if mode == "in":
            return h * (sc + W_(pre + f"ada.{branch}.{layer}_scale")) + (sh + W_(pre + f"ada.{branch}.{layer}_shift"))
        return h * (ga + W_(pre + f"ada.{branch}.{layer}_gate"))

    # --- attention half (mmsr_block.py:108-113)
    xa = ada(rms_norm(x, cfg.norm_eps), bv, "attn", "in")
    ya = rms_norm(y, cfg.norm_eps)
    if not last:  # ada is vid_only in the last layer (mmsr_block.py:82)
        ya = ada(ya, bt, "attn", "in")
    qkv_v = F.linear(xa, W_(pre + f"attn.proj_qkv.{bv}.weight")).reshape(-1, 3, H, D)
    qkv_t = F.linear(ya, W_(pre + f"attn.proj_qkv.{bt}.weight")).reshape(-1, 3, H, D)
    ov, ot = window_attention(
        qkv_v[:, 0], qkv_v[:, 1], qkv_v[:, 2], qkv_t[:, 0], qkv_t[:, 1], qkv_t[:, 2],
        grid, boxes, Lt, cfg,
        W_(pre + f"attn.norm_q.{bv}.weight"), W_(pre + f"attn.norm_k.{bv}.weight"),
        W_(pre + f"attn.norm_q.{bt}.weight"), W_(pre + f"attn.norm_k.{bt}.weight"),
        dtype, rope_mode)
    extras = {"attn_v": ov, "attn_t": ot}
    ov = F.linear(ov, W_(pre + f"attn.proj_out.{bv}.weight"), W_(pre + f"attn.proj_out.{bv}.bias"))
    ot = F.linear(ot, W_(pre + f"attn.proj_out.{bt}.weight"), W_(pre + f"attn.proj_out.{bt}.bias"))
    extras.update(ov=ov, ot=ot)
    x1 = ada(ov, bv, "attn", "out") + x
    y1 = (ot if last else ada(ot, bt, "attn", "out")) + y
    # --- mlp half (mmsr_block.py:115-126)
    def mlp(h, b):
        if cfg.mlp_type == "swiglu":  # mlp.py:60-61
            return F.linear(F.silu(F.linear(h, W_(pre + f"mlp.{b}.proj_in_gate.weight"))) *
                            F.linear(h, W_(pre + f"mlp.{b}.proj_in.weight")), W_(pre + f"mlp.{b}.proj_out.weight"))
        h = F.linear(h, W_(pre + f"mlp.{b}.proj_in.weight"), W_(pre + f"mlp.{b}.proj_in.bias"))  # dit_7b/mlp.py:39-43
        return F.linear(F.gelu(h, approximate="tanh"), W_(pre + f"mlp.{b}.proj_out.weight"),
                        W_(pre + f"mlp.{b}.proj_out.bias"))

    xm = mlp(ada(rms_norm(x1, cfg.norm_eps), bv, "mlp", "in"), bv)
    xo = ada(xm, bv, "mlp", "out") + x1
    if not last:
        ym = mlp(ada(rms_norm(y1, cfg.norm_eps), bt, "mlp", "in"), bt)
        yo = ada(ym, bt, "mlp", "out") + y1
    else:
        yo = y1 + y1  # mmsr_block.py:126 with vid_only pass-through; value unused downstream
    extras.update(x1=x1, y1=y1)
    return xo, yo, extras


def forward(p: Dict[str, torch.Tensor], cfg: DiTConfig, vid: torch.Tensor, txt: torch.Tensor, thw,
            timestep: float = 1000.0, dtype=torch.float32, rope_mode: str = "3b", taps: dict | None = None):
    """NaDiT.forward for one sample.  vid [T*H*W, 33] (latent grid T,H,W, channels-last),
    txt [Lt, txt_in_dim].  Returns [T*H*W, 16]."""
    T, Hh, Ww = (int(v) for v in thw)
    pt, ph, pw = cfg.patch
    grid = (T // pt, Hh // ph, Ww // pw)
    d, H, D = cfg.vid_dim, cfg.heads, cfg.head_dim
    W_ = lambda k: p[k].to(dtype)
    x = F.linear(patch_in(vid.to(dtype), (T, Hh, Ww), cfg.patch), W_("vid_in.proj.weight"), W_("vid_in.proj.bias"))
    y = F.linear(txt.to(dtype), W_("txt_in.weight"), W_("txt_in.bias"))
    Lt = y.shape[0]
    emb = time_embedding(p, timestep, dtype)[0]  # [6d]
    if taps is not None:
        taps["emb"] = emb.clone(); taps["x0"] = x.clone(); taps["y0"] = y.clone()

    for i in range(cfg.num_layers):
        x, y, ex = block(p, cfg, i, x, y, emb, grid, dtype, rope_mode)
        if taps is not None:
            taps[f"x{i + 1}"] = x.clone()
            if i == 0:
                taps["attn_v0"] = ex["ov"].clone(); taps["attn_t0"] = ex["ot"].clone(); taps["y1"] = y.clone()

    # nadit.py:234-244 - vid_out_norm (affine) then vid_out_ada: reuses the cached attn-layer
    # emb slice (cache key 'emb_repeat_0_vid', modulation.py:79-86) + its own out_shift/out_scale.
    if cfg.out_norm:  # 7B has neither (dit_7b/nadit.py:137-141, :188)
        x = rms_norm(x, cfg.norm_eps, W_("vid_out_norm.weight"))
        sh, sc, _ = ada_slices(emb, 0)
        x = x * (sc + W_("vid_out_ada.out_scale")) + (sh + W_("vid_out_ada.out_shift"))
    x = F.linear(x, W_("vid_out.proj.weight"), W_("vid_out.proj.bias"))
    return patch_out(x, grid, cfg.patch, cfg.vid_out_channels)


def one_step_sr(p, cfg, latent: torch.Tensor, noise: torch.Tensor, txt: torch.Tensor, dtype=torch.float32,
                rope_mode: str = "3b"):
    """core/infer.py:54-78,294-374 + samplers/euler.py:36-66 + schedules/base.py:89-114 with steps=1,
    cfg scale 1: cond = [latent, 1]; x0 = noise - DiT(cat[noise, cond]) at t = T = 1000."""
    T, Hh, Ww, C = latent.shape
    cond = torch.cat((latent, torch.ones(T, Hh, Ww, 1, dtype=latent.dtype)), dim=-1)
    vid = torch.cat((noise, cond), dim=-1).reshape(T * Hh * Ww, 2 * C + 1)
    pred = forward(p, cfg, vid, txt, (T, Hh, Ww), 1000.0, dtype, rope_mode)
    return (noise.reshape(-1, C).to(dtype) - pred).reshape(T, Hh, Ww, C)
