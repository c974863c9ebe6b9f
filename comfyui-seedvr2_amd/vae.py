"""Causal 3-D video VAE (s8 / t4 / 16 latent channels) on the HIP kernels, channels-last (NDHWC).

Mirrors ``VideoAutoencoderKLWrapper.encode/decode`` and the temporal slicing schedule of the reference
[ref: src/models/video_vae_v3/modules/attn_video_vae.py:1246-1292 (slicing_encode/decode),
 :808-856 (Encoder3D.forward), :983-1035 (Decoder3D.forward), :311-362 (ResnetBlock3D),
 :110-174 (Upsample3D), :229-250 (Downsample3D), :656-668 (mid block);
 causal_inflation_lib.py:213-305 (causal conv with carried frames)].

Every stage is one launch from libseedvr2_hip:
    conv3d (implicit GEMM, causal head from carry buffers, fused bias + residual)
    groupnorm_silu (per-frame statistics + fused SiLU)
    gemm (1x1x1 convs, attention projections, Upsample3D's channel expansion with fused pixel-shuffle store)
    attn_d512 (mid-block spatial attention)
Temporal chunks carry each conv's last (k_t - s_t) input frames exactly like the reference's
``MemoryState.INITIALIZING / ACTIVE`` protocol, so chunking only bounds memory and never changes results.
Weights are taken in the reference's state-dict layout ([Cout, Cin, kt, kh, kw] convs).
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import hip


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
    # temporal slicing [ref: configs_3b/main.yaml:53-55; core/model_configuration.py:1237-1248]
    # The reference's first slice is 1 + split frames; with the carried conv state any slicing gives the same
    # result (oracle/pin_against_reference.py G10, tests), so the first slice here is frame 0 alone: the
    # full-resolution working set is then 4 frames instead of 5.
    encode_split: int = 8   # input frames per slice after the first (measured: 8 beats 4 by 1 % at 4K, same peak)
    decode_split: int = 1   # latent frames per slice after the first
    encode_first: int = 1   # input frames in the first slice (reference: 1 + encode_split)
    decode_first: int = 1   # latent frames in the first slice (reference: 1 + decode_split)
    norm_carry_copy: bool = True   # a GroupNorm that feeds a causal conv also writes that conv's next-chunk state (a copy of
    #                                its last frames); False: the state is a VIEW of the norm's output, which then stays
    #                                allocated until the next chunk has used it (no extra writes, more memory)


class _State:
    def __init__(self):
        self.mem: Dict[str, torch.Tensor] = {}
        self.active = False
        self.last = False   # final chunk: nothing to carry
        self.stats = None   # (tensor, hip.GnPartial): GroupNorm partial sums the last halo conv wrote for its output
        self.next_carry: Dict[str, torch.Tensor] = {}   # conv name -> carry its producer (a norm) already wrote


class VideoVAEHip:
    def __init__(self, cfg: VAEConfig, params: Dict[str, torch.Tensor], device):
        self.device = torch.device(device)
        self._init_on_device(cfg, params)

    @hip.on_device
    def _init_on_device(self, cfg: VAEConfig, params: Dict[str, torch.Tensor]):
        device = self.device
        self.cfg = cfg
        self.device = torch.device(device)
        dev = self.device
        self.fused_conv_in = os.environ.get("SVR2_CONV_IN_FUSED", "1") != "0"      # A/B switches, resolved once
        self.first_carry_view = os.environ.get("SVR2_FIRST_CARRY_VIEW", "1") != "0"
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.convs: Dict[str, dict] = {}
        self.norms: Dict[str, tuple] = {}
        self.attn: Dict[str, dict] = {}
        for k, v in params.items():
            if k.endswith(".weight") and v.dim() == 5:
                name = k[: -len(".weight")]
                cout, cin, kt, kh, kw = v.shape
                ent = {"cout": cout, "cin": cin, "k": (kt, kh, kw), "bias": f32(params[name + ".bias"])}
                if cin % 64 == 0:
                    # the K layout depends on the kernel that will run the layer (stride / padding are call-site
                    # facts): built on first use by _conv_weight(), then the 5-D copy is dropped
                    ent["w5"] = v.detach().to(device=dev, dtype=torch.bfloat16)
                else:  # small-Cin layers (3, 16): explicit im2col + GEMM ...
                    ent["kpad"] = (kt * kh * kw * cin + 63) // 64 * 64
                    ent["w"] = hip.pad_weight(hip.conv_weight(v.detach().to(dev)), ent["kpad"])
                    if (cout, cin, kt, kh, kw) == (128, 3, 3, 3, 3) and self.fused_conv_in:
                        # ... except the encoder's conv_in, which has a fused kernel (csrc/conv_in.hip): no column matrix
                        ent["w_cin3"] = hip.conv_weight_cin3(v.detach().to(dev))
                self.convs[name] = ent
            elif k.endswith(".weight") and v.dim() == 1:
                name = k[: -len(".weight")]
                self.norms[name] = (f32(v), f32(params[name + ".bias"]))
        for side in ("encoder", "decoder"):
            a = f"{side}.mid_block.attentions.0"
            if (a + ".to_q.weight") in params:
                wq = torch.cat([params[f"{a}.{l}.weight"].detach().to(dev) for l in ("to_q", "to_k", "to_v")], dim=0)
                bq = torch.cat([f32(params[f"{a}.{l}.bias"]) for l in ("to_q", "to_k", "to_v")]).contiguous()
                self.attn[a] = {"w_qkv": hip.pad_weight(wq), "b_qkv": bq,
                                "w_out": hip.pad_weight(params[f"{a}.to_out.0.weight"].detach().to(dev)),
                                "b_out": f32(params[f"{a}.to_out.0.bias"]), "c": wq.shape[1]}

    # ------------------------------------------------------------------ primitives
    @staticmethod
    def _conv_weight(c: dict, stride, pad_hw) -> None:
        """Pick the weight layout of a Cin % 64 == 0 conv on first use: the 3x3, stride-1, pad-1 layers run on the
        halo-resident kernel (SVR2_WL_TILED_128x32); the rest (strided downsamplers, 1x1x1 shortcuts) use the row
        layout of the tap-staged kernels."""
        kt, kh, kw = c["k"]
        tiled = (kh, kw) == (3, 3) and tuple(stride) == (1, 1, 1) and tuple(pad_hw) == (1, 1, 1, 1)
        w5 = c.pop("w5")
        c["layout"] = hip.WL_TILED_128x32 if tiled else hip.WL_ROWS
        c["w"] = hip.conv_weight_tiled(w5) if tiled else hip.pad_weight(hip.conv_weight(w5))

    def _conv(self, name: str, x: torch.Tensor, st: _State, stride=(1, 1, 1), pad_hw=None,
              resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        c = self.convs[name]
        kt, kh, kw = c["k"]
        if pad_hw is None:
            pad_hw = (kh // 2, kh // 2, kw // 2, kw // 2)
        keep = kt - stride[0]  # frames of input carried to the next chunk
        if st.active and keep > 0:
            carry, npad = st.mem[name], keep
        else:
            carry, npad = None, 2 * ((kt - 1) // 2)  # first chunk: replicate frame 0  [ref: causal_inflation_lib.py:422-437]
        if "w_cin3" in c and tuple(stride) == (1, 1, 1) and tuple(pad_hw) == (1, 1, 1, 1) and npad == 2 and resid is None:
            y, part = hip.conv3d(x, c["w_cin3"], c["cout"], c["k"], stride, npad=npad, pad_hw=pad_hw, carry=carry,
                                 bias=c["bias"], w_layout=hip.WL_CIN3, gn_stats=True)
            st.stats = (y, part)
        elif "kpad" in c:
            col = hip.im2col3d(x, c["k"], stride, npad=npad, pad_hw=pad_hw, carry=carry, kpad=c["kpad"])
            to, ho, wo = hip.conv_out_size(x.shape[0], x.shape[1], x.shape[2], c["k"], stride, npad, pad_hw)
            y = hip.gemm(col, c["w"], c["cout"], bias=c["bias"]).view(to, ho, wo, c["cout"])
            if resid is not None:
                raise hip.HipError("small-Cin conv has no residual path")
        else:
            if "w" not in c:
                self._conv_weight(c, stride, pad_hw)
            if c["layout"] == hip.WL_TILED_128x32 and c["cout"] % 8 == 0:
                # halo-resident kernel: its epilogue also reduces the GroupNorm statistics of the output per tile
                y, part = hip.conv3d(x, c["w"], c["cout"], c["k"], stride, npad=npad, pad_hw=pad_hw, carry=carry,
                                     bias=c["bias"], resid=resid, w_layout=c["layout"], gn_stats=True)
                st.stats = (y, part)
            else:
                y = hip.conv3d(x, c["w"], c["cout"], c["k"], stride, npad=npad, pad_hw=pad_hw, carry=carry,
                               bias=c["bias"], resid=resid, w_layout=c["layout"])
        if keep > 0 and not st.last:
            # carry the last `keep` input frames to the next chunk  [ref: causal_inflation_lib.py:266-278].  The old
            # carry is released first (stream-ordered reuse: the conv that reads it is already enqueued).
            carry = None
            if name in st.next_carry:          # the norm that produced x already stored them (_norm carry_for=)
                st.mem[name] = st.next_carry.pop(name)
            elif x.shape[0] >= keep:
                # a VIEW of the chunk's last frames, not a copy: x is the output of a kernel that nothing writes again (a
                # resnet's conv2, the pixel-shuffle GEMM, the input chunk), the slice of the leading dimension is
                # contiguous, and the copies were 1 % of a 4K step (27 GB of traffic per decoder chunk).  Price: the whole
                # chunk tensor stays allocated until the next chunk has consumed the carry.
                st.mem[name] = None
                st.mem[name] = x[x.shape[0] - keep:]
            elif not st.active and x.shape[0] == 1 and not self.first_carry_view:
                st.mem[name] = x.expand(keep, -1, -1, -1).contiguous()   # first chunk of one frame: replicas of it (A/B)
            elif not st.active and x.shape[0] == 1:
                # first chunk of one frame: the state is [f0, f0].  The kernels take the single frame as the carry and let
                # it stand for the missing earlier one (svr2_conv3d_desc.carry_frames), so this is a view too - the two
                # replicas were two 2.1 GB copies per full-resolution conv of a 4K batch (0.4 % of the step)
                st.mem[name] = None
                st.mem[name] = x
            else:
                head = st.mem.get(name) if st.active else None
                head = head if head is not None else x[:1].expand(npad, -1, -1, -1)
                st.mem[name] = torch.cat((head, x), dim=0)[-keep:].clone()
        elif keep > 0:
            st.mem.pop(name, None)
        st.next_carry.pop(name, None)
        return y

    def _norm(self, name: str, x: torch.Tensor, st: _State, silu: bool = True,
              carry_for: Optional[str] = None) -> torch.Tensor:
        """GroupNorm (+SiLU).  Uses the statistics the producing conv reduced in its epilogue when x is that conv's
        output; carry_for = name of the causal conv that consumes the result: its next-chunk state (the last kt-1
        frames of the result) is written by the same pass instead of being copied out afterwards."""
        g, b = self.norms[name]
        part = st.stats[1] if (st.stats is not None and st.stats[0] is x) else None
        st.stats = None
        carry_out = None
        if carry_for is not None and not st.last and self.cfg.norm_carry_copy:
            keep = self.convs[carry_for]["k"][0] - 1
            if 0 < keep <= x.shape[0]:
                carry_out = torch.empty((keep,) + tuple(x.shape[1:]), dtype=torch.bfloat16, device=x.device)
                st.next_carry[carry_for] = carry_out
        return hip.groupnorm_silu(x, g, b, self.cfg.norm_groups, self.cfg.eps, silu=silu, partial=part,
                                  carry_out=carry_out)

    def _resnet(self, name: str, x: torch.Tensor, st: _State) -> torch.Tensor:
        h = self._norm(name + ".norm1", x, st, carry_for=name + ".conv1")
        h = self._conv(name + ".conv1", h, st)
        h = self._norm(name + ".norm2", h, st, carry_for=name + ".conv2")
        skip = x
        if (name + ".conv_shortcut") in self.convs:
            skip = self._conv(name + ".conv_shortcut", x, st, pad_hw=(0, 0, 0, 0))
        return self._conv(name + ".conv2", h, st, resid=skip)

    def _mid_attention(self, name: str, x: torch.Tensor, st: _State) -> torch.Tensor:
        a = self.attn[name]
        T, H, W, Cc = x.shape
        if Cc != 512:
            raise hip.HipError("mid-block attention kernel is built for 512 channels")
        h = self._norm(name + ".group_norm", x, st, silu=False).view(T * H * W, Cc)
        qkv = hip.gemm(h, a["w_qkv"], 3 * Cc, bias=a["b_qkv"])
        o = hip.attn_d512(qkv, T, H * W)
        out = hip.gemm(o, a["w_out"], Cc, bias=a["b_out"], resid=x.view(T * H * W, Cc))
        return out.view(T, H, W, Cc)

    def _mid(self, side: str, x: torch.Tensor, st: _State) -> torch.Tensor:
        x = self._resnet(f"{side}.mid_block.resnets.0", x, st)
        x = self._mid_attention(f"{side}.mid_block.attentions.0", x, st)
        return self._resnet(f"{side}.mid_block.resnets.1", x, st)

    def _upsample(self, name: str, x: torch.Tensor, tz: int, st: _State) -> torch.Tensor:
        c = self.convs[name + ".upscale_conv"]
        T, H, W, Cc = x.shape
        drop = tz == 2 and not st.active  # remove_head on the first chunk  [ref: attn_video_vae.py:152-153]
        if "w" not in c:
            self._conv_weight(c, (1, 1, 1), (0, 0, 0, 0))   # 1x1x1: plain [N, K] rows
        y = hip.gemm(x.view(T * H * W, Cc), c["w"], c["cout"], epilogue=hip.EPI_PIXSHUF3D, bias=c["bias"],
                     pixshuf=(T, H, W, Cc, tz, drop))
        return self._conv(name + ".conv", y, st)

    # ------------------------------------------------------------------ encoder / decoder on one chunk
    def _encoder(self, x: torch.Tensor, st: _State) -> torch.Tensor:
        cfg = self.cfg
        n = len(cfg.block_out_channels)
        h = self._conv("encoder.conv_in", x, st)
        for i in range(n):
            for j in range(cfg.layers_per_block):
                h = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", h, st)
            if i != n - 1:
                tdown = i >= n - cfg.temporal_scale_num - 1
                h = self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", h, st,
                               stride=(2 if tdown else 1, 2, 2), pad_hw=(0, 1, 0, 1))
        h = self._mid("encoder", h, st)
        h = self._norm("encoder.conv_norm_out", h, st, carry_for="encoder.conv_out")
        return self._conv("encoder.conv_out", h, st)

    def _decoder(self, z: torch.Tensor, st: _State) -> torch.Tensor:
        cfg = self.cfg
        n = len(cfg.block_out_channels)
        h = self._conv("decoder.conv_in", z, st)
        h = self._mid("decoder", h, st)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                h = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", h, st)
            if i != n - 1:
                h = self._upsample(f"decoder.up_blocks.{i}.upsamplers.0", h, 2 if i < cfg.temporal_scale_num else 1, st)
        h = self._norm("decoder.conv_norm_out", h, st, carry_for="decoder.conv_out")
        return self._conv("decoder.conv_out", h, st)

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    @hip.on_device
    def encode(self, x: torch.Tensor, tiled: bool = False, tile_size: Tuple[int, int] = (512, 512),
               tile_overlap: Tuple[int, int] = (64, 64)) -> torch.Tensor:
        """x bf16 [T,H,W,3] in [-1,1], T = 4n+1, H,W multiples of 8 -> latent bf16 [T_l,h,w,16]
        (posterior mean * scaling_factor)  [ref: src/core/infer.py:117-188; attn_video_vae.py:1672-1681].
        tiled: spatial tiles of `tile_size` input pixels blended over `tile_overlap` [ref: attn_video_vae.py:1294-1458]."""
        cfg = self.cfg
        T = x.shape[0]
        if x.dim() != 4 or x.shape[3] != cfg.in_channels or T % 4 != 1 or x.shape[1] % 8 or x.shape[2] % 8:
            raise hip.HipError(f"encode: expected [4n+1, 8a, 8b, {cfg.in_channels}], got {tuple(x.shape)}")
        if cfg.encode_split % 4 or (cfg.encode_first - 1) % 4:
            raise hip.HipError("encode slices must keep the x4 temporal stride aligned (first = 4a+1, split = 4b)")
        m = self._tiled(x, self._encode_mean, tile_size, tile_overlap, 8, 1) if tiled else self._encode_mean(x)
        return (m.float() * cfg.scaling_factor).to(torch.bfloat16).contiguous()

    def _encode_mean(self, x: torch.Tensor) -> torch.Tensor:
        """Temporal slices with carried conv state [ref: attn_video_vae.py:1246-1268] -> posterior mean [T_l,h,w,16]."""
        cfg, T = self.cfg, x.shape[0]
        st = _State()
        outs = []
        a = 0
        while a < T:
            b = min(T, a + (cfg.encode_first if a == 0 else cfg.encode_split))
            st.last = b >= T
            outs.append(self._encoder(x[a:b].contiguous(), st)[..., : cfg.latent_channels])
            st.active = True
            a = b
        return (torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]).contiguous()

    @torch.no_grad()
    @hip.on_device
    def decode(self, latent: torch.Tensor, tiled: bool = False, tile_size: Tuple[int, int] = (512, 512),
               tile_overlap: Tuple[int, int] = (64, 64)) -> torch.Tensor:
        """latent bf16 [T_l,h,w,16] -> bf16 [T,H,W,3], T = 4(T_l-1)+1  [ref: src/core/infer.py:192-257].
        tiled: spatial tiles of `tile_size` OUTPUT pixels blended over `tile_overlap` [ref: attn_video_vae.py:1460-1622]."""
        cfg = self.cfg
        if latent.dim() != 4 or latent.shape[3] != cfg.latent_channels:
            raise hip.HipError(f"decode: expected [T_l,h,w,{cfg.latent_channels}], got {tuple(latent.shape)}")
        z = (latent.float() / cfg.scaling_factor).to(torch.bfloat16).contiguous()
        return self._tiled(z, self._decode_frames, tile_size, tile_overlap, 1, 8) if tiled else self._decode_frames(z)

    def _decode_frames(self, z: torch.Tensor) -> torch.Tensor:
        cfg, Tl = self.cfg, z.shape[0]
        st = _State()
        outs = []
        a = 0
        while a < Tl:
            b = min(Tl, a + (cfg.decode_first if a == 0 else cfg.decode_split))
            st.last = b >= Tl
            outs.append(self._decoder(z[a:b].contiguous(), st))
            st.active = True
            a = b
        return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]

    # spatial tiling -----------------------------------------------------------------------------
    @staticmethod
    def tile_plan(total_h: int, total_w: int, tile: Tuple[int, int], overlap: Tuple[int, int]):
        """Latent-space tiles (y, y_end, x, x_end) in visiting order; a tile that lies inside the overlap of its
        predecessor is skipped [ref: attn_video_vae.py:1354-1364, 1517-1527]."""
        (th, tw), (oh, ow) = tile, overlap
        sh, sw = max(1, th - oh), max(1, tw - ow)
        return [(y, min(y + th, total_h), x, min(x + tw, total_w))
                for y in range(0, total_h, sh) for x in range(0, total_w, sw)
                if not ((y > 0 and min(y + th, total_h) - y <= oh) or (x > 0 and min(x + tw, total_w) - x <= ow))]

    @staticmethod
    def _fade(n: int, ov: int, ramp: Optional[torch.Tensor], head: bool, tail: bool) -> torch.Tensor:
        """Ones with the raised-cosine ramp on interior edges only [ref: attn_video_vae.py:1417-1431, 1579-1593]."""
        w = torch.ones(n, dtype=torch.bfloat16)
        if ov > 0:
            if head:
                w[:ov] = ramp[:ov]
            if tail:
                w[-ov:] = 1 - ramp[:ov]
        return w

    def _tiled(self, x: torch.Tensor, codec, tile_size, tile_overlap, in_scale: int, out_scale: int) -> torch.Tensor:
        """Shared body of tiled_encode (in_scale 8, out_scale 1) and tiled_decode (in_scale 1, out_scale 8): tiles are
        planned in latent units, cut from `x` at in_scale, and their results are blended at out_scale by the HIP
        kernels, which keep the reference's bf16 rounding after every in-place op.  The fade weights are a few hundred
        bf16 numbers computed on the host with the reference's own torch expressions."""
        H, W = x.shape[1], x.shape[2]
        sf = 8
        (tile_h, tile_w), (ov_h, ov_w) = tile_size, tile_overlap
        lt_h, lt_w = max(1, tile_h // sf), max(1, tile_w // sf)
        Hl, Wl = (H + in_scale - 1) // in_scale, (W + in_scale - 1) // in_scale
        if (in_scale == sf and H <= tile_h and W <= tile_w) or (in_scale == 1 and H <= lt_h and W <= lt_w):
            return codec(x)
        lo_h, lo_w = max(0, min(ov_h // sf, lt_h - 1)), max(0, min(ov_w // sf, lt_w - 1))
        # encode fades over the latent overlap, decode over `tile_overlap` output pixels (:1346-1351 vs :1510-1515)
        rl_h, rl_w = (lo_h, lo_w) if out_scale == 1 else (ov_h, ov_w)

        def ramp(steps):
            if steps <= 0:
                return None
            t = torch.linspace(0, 1, steps=steps, dtype=torch.bfloat16)
            return 0.5 - 0.5 * torch.cos(t * torch.pi)

        rh, rw = ramp(rl_h), ramp(rl_w)
        result = count = None
        for (y, y_end, xx, x_end) in self.tile_plan(Hl, Wl, (lt_h, lt_w), (lo_h, lo_w)):
            tile = x[:, y * in_scale:min(y_end * in_scale, H), xx * in_scale:min(x_end * in_scale, W)].contiguous()
            out = codec(tile).contiguous()
            if result is None:
                result = torch.zeros(out.shape[0], Hl * out_scale, Wl * out_scale, out.shape[3], dtype=torch.bfloat16,
                                     device=x.device)
                count = torch.zeros(Hl * out_scale, Wl * out_scale, dtype=torch.bfloat16, device=x.device)
            y0, x0 = y * out_scale, xx * out_scale
            eh = min((y_end - y) * out_scale, out.shape[1], result.shape[1] - y0)
            ew = min((x_end - xx) * out_scale, out.shape[2], result.shape[2] - x0)
            wh = self._fade(eh, max(0, min(rl_h, eh - 1)), rh, y > 0, y_end < Hl).to(x.device)
            ww = self._fade(ew, max(0, min(rl_w, ew - 1)), rw, xx > 0, x_end < Wl).to(x.device)
            hip.tile_accumulate(out, result, count, y0, x0, eh, ew, wh, ww)
            del out, tile
        return hip.tile_normalize(result, count)


def synthetic_state_dict(cfg: VAEConfig, device, seed: int = 0, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init VAE weights in the reference's key layout, generated on `device` (bench only)."""
    import math
    g = torch.Generator(device=device).manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    ch = cfg.block_out_channels
    n = len(ch)

    def conv(name, cin, cout, k):
        fan = cin * k[0] * k[1] * k[2]
        out[name + ".weight"] = (torch.randn(cout, cin, *k, generator=g, device=device) / math.sqrt(fan)).to(dtype)
        out[name + ".bias"] = (0.05 * torch.randn(cout, generator=g, device=device)).to(dtype)

    def norm(name, c):
        out[name + ".weight"] = (1 + 0.1 * torch.randn(c, generator=g, device=device)).to(dtype)
        out[name + ".bias"] = (0.05 * torch.randn(c, generator=g, device=device)).to(dtype)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cin, cout, (3, 3, 3))
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, (3, 3, 3))
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, (1, 1, 1))

    def mid(name, c):
        resnet(name + ".resnets.0", c, c)
        norm(name + ".attentions.0.group_norm", c)
        for l in ("to_q", "to_k", "to_v", "to_out.0"):
            out[f"{name}.attentions.0.{l}.weight"] = (torch.randn(c, c, generator=g, device=device) / math.sqrt(c)).to(dtype)
            out[f"{name}.attentions.0.{l}.bias"] = (0.05 * torch.randn(c, generator=g, device=device)).to(dtype)
        resnet(name + ".resnets.1", c, c)

    conv("encoder.conv_in", cfg.in_channels, ch[0], (3, 3, 3))
    cout = ch[0]
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
    return out
