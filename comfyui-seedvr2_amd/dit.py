"""NaDiT (SeedVR2 3B / 7B-style) one-step forward on the HIP kernels.

Mirrors the call structure of the reference's ``NaDiT.forward`` [ref: src/models/dit_3b/nadit.py:190-248]
and ``NaMMSRTransformerBlock.forward`` [ref: src/models/dit_3b/nablocks/mmsr_block.py:84-128], but every
stage is one launch of a fused kernel from libseedvr2_hip (see include/seedvr2_hip.h):

    rmsnorm_mod -> gemm(qkv) -> qkv_prep (window gather + qk-norm + rope) -> window_attn
    -> gemm(proj_out, fused window-reverse gather + bias + gate + residual)
    -> rmsnorm_mod -> gemm(SwiGLU) -> gemm(proj_out, gate + residual)

At inference the AdaSingle modulation is a set of per-layer constant vectors (the timestep embedding is
fixed at t = T = 1000 for the one-step sampler [ref: src/core/generation_phases.py:595-597,
src/models/dit_3b/modulation.py:65-118]); they are folded at load time into fp32 scale/shift/gate vectors,
including the ``vid_out_ada`` cache-key collision [ref: nadit.py:234-244, common/cache.py:18-47].
Weights are taken in the reference's state-dict layout.
"""

from __future__ import annotations

import contextlib
import math
import os
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

from . import hip
from . import windows as W


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
    window: Tuple[int, int, int] = (4, 3, 3)
    mlp_multiple_of: int = 256
    rope_mode: int = 1  # 1 = 3B mm-rope ('lang'), 2 = 7B video-only 'pixel' rope, 0 = none
    # 7B deltas [ref: src/models/dit_7b/nadit.py:39-190, configs_7b/main.yaml:11-34]
    mlp_type: str = "swiglu"     # "swiglu" | "normal" (Linear+b -> GELU(tanh) -> Linear+b)
    out_norm: bool = True        # vid_out_norm + vid_out_ada (3B only)
    last_vid_only: bool = True   # 3B: last layer's mlp / ada are vid-only

    @property
    def mlp_hidden(self) -> int:
        if self.mlp_type == "normal":
            return self.vid_dim * self.expand_ratio  # [ref: dit_7b/mlp.py:34]
        hid = int(2 * self.vid_dim * self.expand_ratio / 3)  # [ref: dit_3b/mlp.py:54-55]
        m = self.mlp_multiple_of
        return m * ((hid + m - 1) // m)

    def shared(self, i: int) -> bool:
        return not (i < self.mm_layers)  # [ref: dit_3b/nadit.py:149-151]

    def method(self, i: int) -> str:
        return W.REGULAR if i % 2 == 0 else W.SHIFTED  # [ref: configs_3b/main.yaml:34]


CFG_3B = DiTConfig()
CFG_7B = DiTConfig(vid_dim=3072, heads=24, num_layers=36, mm_layers=36, rope_mode=2, mlp_type="normal",
                   out_norm=False, last_vid_only=False)


def _timestep_embedding(t: float, dim: int = 256) -> torch.Tensor:
    """[sin(t f), cos(t f)], f_i = exp(-ln(1e4) i / half)  (diffusers get_timestep_embedding as called at
    [ref: src/models/dit_3b/embedding.py:50-55]: flip_sin_to_cos=False, downscale_freq_shift=0)."""
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = torch.tensor([float(t)], dtype=torch.float32)[:, None] * f[None]
    return torch.cat((a.sin(), a.cos()), dim=-1)


class NaDiTHip:
    def __init__(self, cfg: DiTConfig, params: Dict[str, torch.Tensor], device, timestep: float = 1000.0):
        self.device = torch.device(device)
        self._init_on_device(cfg, params, timestep)

    @hip.on_device
    def _init_on_device(self, cfg: DiTConfig, params: Dict[str, torch.Tensor], timestep: float):
        self.cfg = cfg
        self._plans = {}
        # A/B switches, resolved ONCE here (tests set the attributes; nothing on the per-block / per-call path reads the
        # environment): graph_mode / txt_lane_mode "auto" | "0" | "1"
        self.graph_mode = os.environ.get("SVR2_DIT_GRAPH", "auto")
        self.txt_lane_mode = os.environ.get("SVR2_DIT_TXT_LANE", "auto")
        d = cfg.vid_dim
        dev = self.device
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()

        # ---- constant time embedding -> modulation vectors (fp32)  [ref: dit_3b/embedding.py:39-62]
        e = _timestep_embedding(timestep).to(dev)
        lin = lambda x, k: torch.nn.functional.linear(x, f32(params[k + ".weight"]), f32(params[k + ".bias"]))
        e = torch.nn.functional.silu(lin(e, "emb_in.proj_in"))
        e = torch.nn.functional.silu(lin(e, "emb_in.proj_hid"))
        emb = lin(e, "emb_in.proj_out")[0].reshape(d, 2, 3)  # [d, layer, (shift, scale, gate)]
        self.emb = emb

        self.w_vid_in = hip.pad_weight(params["vid_in.proj.weight"].to(dev))
        self.b_vid_in = f32(params["vid_in.proj.bias"])
        self.kpad_in = self.w_vid_in.shape[1]
        self.w_txt_in = hip.pad_weight(params["txt_in.weight"].to(dev))
        self.b_txt_in = f32(params["txt_in.bias"])
        self.w_vid_out = hip.pad_weight(params["vid_out.proj.weight"].to(dev))
        self.b_vid_out = f32(params["vid_out.proj.bias"])
        if cfg.out_norm:
            # vid_out_norm (affine) + vid_out_ada reuse of the attn-layer slice  [ref: nadit.py:234-244]
            self.out_scale = (f32(params["vid_out_norm.weight"]) * (emb[:, 0, 1] + f32(params["vid_out_ada.out_scale"]))).contiguous()
            self.out_shift = (emb[:, 0, 0] + f32(params["vid_out_ada.out_shift"])).contiguous()

        ones = torch.ones(d, dtype=torch.float32, device=dev)
        self.layers = []
        for i in range(cfg.num_layers):
            last = cfg.last_vid_only and i == cfg.num_layers - 1
            bv, bt = ("all", "all") if cfg.shared(i) else ("vid", "txt")
            pre = f"blocks.{i}."
            L = {}
            for tag, b in (("v", bv), ("t", bt)):
                if tag == "t" and b == bv and "v" in L:
                    L["t"] = L["v"] if not last else self._txt_last_layer(L["v"], ones)
                    continue
                g = lambda name: params[pre + name]
                ent = {
                    "w_qkv": hip.pad_weight(g(f"attn.proj_qkv.{b}.weight").to(dev)),
                    "w_out": hip.pad_weight(g(f"attn.proj_out.{b}.weight").to(dev)),
                    "b_out": f32(g(f"attn.proj_out.{b}.bias")),
                    "qn": f32(g(f"attn.norm_q.{b}.weight")), "kn": f32(g(f"attn.norm_k.{b}.weight")),
                }
                vid_only_skip = last and tag == "t"
                if not vid_only_skip:
                    if cfg.mlp_type == "swiglu":
                        ent["w_gi"] = hip.pad_weight(hip.interleave_swiglu(g(f"mlp.{b}.proj_in_gate.weight").to(dev),
                                                                           g(f"mlp.{b}.proj_in.weight").to(dev)))
                    else:
                        ent["w_gi"] = hip.pad_weight(g(f"mlp.{b}.proj_in.weight").to(dev))
                        ent["b_gi"] = f32(g(f"mlp.{b}.proj_in.bias"))
                        ent["b_mo"] = f32(g(f"mlp.{b}.proj_out.bias"))
                    ent["w_mo"] = hip.pad_weight(g(f"mlp.{b}.proj_out.weight").to(dev))
                    for li, lname in enumerate(("attn", "mlp")):
                        ent[f"{lname}_scale"] = (emb[:, li, 1] + f32(g(f"ada.{b}.{lname}_scale"))).contiguous()
                        ent[f"{lname}_shift"] = (emb[:, li, 0] + f32(g(f"ada.{b}.{lname}_shift"))).contiguous()
                        ent[f"{lname}_gate"] = (emb[:, li, 2] + f32(g(f"ada.{b}.{lname}_gate"))).contiguous()
                else:  # last layer: ada/mlp are vid_only -> text is only normalised  [ref: mmsr_block.py:73-82]
                    ent.update({"attn_scale": ones, "attn_shift": None, "attn_gate": ones})
                L[tag] = ent
            self.layers.append(L)

        self._rope7 = {}
        self._txt_stream = None
        # rope tables (3B): angle = pos * theta^(-2i/42), i < 21  [ref: dit_3b/rope.py:74-85]
        nfreq = (cfg.head_dim // 3) // 2
        fr = 1.0 / (10000.0 ** (torch.arange(0, cfg.head_dim // 3, 2)[:nfreq].float() / (cfg.head_dim // 3)))
        ang = torch.arange(1024, dtype=torch.float32)[:, None] * fr[None]
        self.rope_cos = ang.cos().contiguous().to(dev)
        self.rope_sin = ang.sin().contiguous().to(dev)

    @staticmethod
    def _txt_last_layer(v_ent, ones):
        ent = dict(v_ent)
        ent.update({"attn_scale": ones, "attn_shift": None, "attn_gate": ones})
        for k in ("w_gi", "w_mo", "mlp_scale", "mlp_shift", "mlp_gate"):
            ent.pop(k, None)
        return ent

    # ------------------------------------------------------------------ window tables on the device
    def _plan(self, grid, method: str, txt_len: int):
        key = (grid, method, txt_len)
        if key not in self._plans:
            p = W.plan(grid, method, txt_len, self.cfg.window)
            if p.max_pos > self.rope_cos.shape[0]:
                raise hip.HipError(f"rope table too small for position {p.max_pos}")
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            # the attention kernel scatters its rows: video rows back to natural token order (window_reverse), the
            # replicated text rows behind them, so the output projection reads a dense [L, H*D] matrix
            L = int(p.inv_rows.shape[0])
            out_rows = np.full(p.total, -1, dtype=np.int32)
            out_rows[p.inv_rows] = np.arange(L, dtype=np.int32)
            txt_out0 = (L + np.arange(p.nwin, dtype=np.int64) * txt_len).astype(np.int32)
            for w in range(p.nwin):
                out_rows[p.txt_row0[w]:p.txt_row0[w] + txt_len] = txt_out0[w] + np.arange(txt_len, dtype=np.int32)
            assert (np.sort(out_rows) == np.arange(p.total)).all()
            self._plans[key] = {
                "S": p.total, "L": L, "nwin": p.nwin, "src_row": up(p.src_row), "pos": up(p.pos), "cu": up(p.cu_seqlens),
                "tile_seq": up(p.tile_seq), "tile_q0": up(p.tile_q0),
                "out_rows": up(out_rows), "txt_out0": up(txt_out0),
            }
        return self._plans[key]

    def _rope_tables_7b(self, grid, method: str, txt_len: int):
        """Per-position cos/sin [S, 30] of the 7B 'pixel' rope: 10 freqs linspace(1,128,10)*pi per axis,
        window-local positions linspace(-1,1,n); the reference casts the ANGLES to the activations' dtype
        (bf16) before cos/sin [ref: src/models/dit_7b/rope.py:25-47,73-124]; text rows are not rotated."""
        key = (grid, method, txt_len)
        if key not in self._rope7:
            p = W.plan(grid, method, txt_len, self.cfg.window)
            nf = ((self.cfg.head_dim // 2) // 3) // 2
            fr = torch.linspace(1.0, 256.0 / 2, nf) * math.pi
            ang = torch.zeros(p.total, 3 * nf, dtype=torch.float32)
            boxes = W.window_boxes(method, grid, self.cfg.window)
            off = 0
            for (t0, t1, h0, h1, w0, w1) in boxes:
                f, hh, ww = t1 - t0, h1 - h0, w1 - w0
                at = (torch.linspace(-1, 1, f)[:, None] * fr[None])[:, None, None, :].expand(f, hh, ww, nf)
                ah = (torch.linspace(-1, 1, hh)[:, None] * fr[None])[None, :, None, :].expand(f, hh, ww, nf)
                aw = (torch.linspace(-1, 1, ww)[:, None] * fr[None])[None, None, :, :].expand(f, hh, ww, nf)
                n = f * hh * ww
                ang[off:off + n] = torch.cat((at, ah, aw), dim=-1).reshape(n, 3 * nf)
                off += n + txt_len
            angq = ang.to(torch.bfloat16)
            self._rope7[key] = (angq.cos().float().contiguous().to(self.device),
                                angq.sin().float().contiguous().to(self.device))
        return self._rope7[key]

    # ------------------------------------------------------------------ forward
    def block(self, i: int, x: torch.Tensor, y: torch.Tensor, grid, taps=None, txt_lane=None):
        """One MMDiT block.  txt_lane = (side stream, keep list): the text rows' norm / projections / MLP (58 rows: a
        handful of workgroups per launch, latency-bound) are issued on the side stream and overlap the video rows' launches;
        the two lanes meet at qkv_prep (needs both projections) and part again after txt_mean.  Every tensor that crosses
        the lanes is parked in `keep` until the final join, so the caching allocator cannot hand its block out early."""
        cfg = self.cfg
        last = cfg.last_vid_only and i == cfg.num_layers - 1
        Lv, Lt_ = self.layers[i]["v"], self.layers[i]["t"]
        H, D, d = cfg.heads, cfg.head_dim, cfg.vid_dim
        ntxt = y.shape[0]
        pl = self._plan(grid, cfg.method(i), ntxt)
        main = torch.cuda.current_stream(self.device)
        side, keep = txt_lane if txt_lane is not None else (None, None)
        lane = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        # attention half
        with lane():
            ya = hip.rmsnorm_mod(y, Lt_["attn_scale"], Lt_["attn_shift"], cfg.norm_eps)
            qkv_t = hip.gemm(ya, Lt_["w_qkv"], 3 * H * D)
        xa = hip.rmsnorm_mod(x, Lv["attn_scale"], Lv["attn_shift"], cfg.norm_eps)
        qkv_v = hip.gemm(xa, Lv["w_qkv"], 3 * H * D)
        if side is not None:
            keep += [y, ya, qkv_t]
            main.wait_stream(side)
        rc, rs = (self._rope_tables_7b(grid, cfg.method(i), ntxt) if cfg.rope_mode == 2
                  else (self.rope_cos, self.rope_sin))
        qkvw = hip.qkv_prep(qkv_v, qkv_t, pl["src_row"], pl["pos"], Lv["qn"], Lv["kn"], Lt_["qn"], Lt_["kn"],
                            rc, rs, H, cfg.rope_mode, cfg.norm_eps)
        del qkv_v, qkv_t, xa, ya
        # the attention kernel scatters its rows back to natural token order (video rows first, the windows' text rows
        # behind them), so the projection is a dense GEMM; the alternative - window-ordered output, gather fused into the
        # projection's A loads - cost 20 ms more per 4K step and is gone
        o = hip.window_attn(qkvw[0], qkvw[1], qkvw[2], pl["cu"], pl["tile_seq"], pl["tile_q0"], 1.0 / math.sqrt(D),
                            out_rows=pl["out_rows"])
        del qkvw
        o2 = o.view(pl["S"], H * D)
        ot = hip.txt_mean(o2, pl["txt_out0"], ntxt)
        x1 = hip.gemm(o2[: pl["L"]], Lv["w_out"], d, epilogue=hip.EPI_GATE_RES, bias=Lv["b_out"], gate=Lv["attn_gate"],
                      resid=x)
        if side is not None:
            keep.append(ot)
            side.wait_stream(main)
        with lane():
            y1 = hip.gemm(ot, Lt_["w_out"], d, epilogue=hip.EPI_GATE_RES, bias=Lt_["b_out"], gate=Lt_["attn_gate"], resid=y)
        if taps is not None:
            taps["x1"] = x1; taps["y1"] = y1
        del o, o2
        # mlp half
        def mlp(h_in, Lp, resid, park=None):
            hn = hip.rmsnorm_mod(h_in, Lp["mlp_scale"], Lp["mlp_shift"], cfg.norm_eps)
            if cfg.mlp_type == "swiglu":
                hmid = hip.gemm(hn, Lp["w_gi"], 2 * cfg.mlp_hidden, epilogue=hip.EPI_SWIGLU)
            else:
                hmid = hip.gemm(hn, Lp["w_gi"], cfg.mlp_hidden, epilogue=hip.EPI_GELU, bias=Lp["b_gi"])
            if park is not None:
                park += [hn, hmid]
            return hip.gemm(hmid, Lp["w_mo"], d, epilogue=hip.EPI_GATE_RES, bias=Lp.get("b_mo"), gate=Lp["mlp_gate"],
                            resid=resid)

        if not last:
            with lane():
                y2 = mlp(y1, Lt_, y1, keep)
            if side is not None:
                keep += [y1, y2]
        else:
            y2 = y1  # unused downstream
        x2 = mlp(x1, Lv, x1)
        return x2, y2

    GRAPH_MAX_TOKENS = 32768   # below this the ~450 launches of a forward pass cost more than their kernels

    @hip.on_device
    @torch.no_grad()
    def one_step(self, noise: torch.Tensor, latent: torch.Tensor, txt: torch.Tensor, taps=None) -> torch.Tensor:
        """x0 = noise - NaDiT(cat[noise, latent, 1])  [ref: src/core/infer.py:54-78,294-374;
        common/diffusion/schedules/base.py:108-110].  noise, latent: bf16 [T,H,W,16]; txt bf16 [Lt, txt_in_dim].
        Small grids replay a captured hipGraph of the whole forward pass (SVR2_DIT_GRAPH=0 disables, =1 forces):
        the pass is a fixed sequence of launches for a given grid, with no host synchronisation inside."""
        mode = self.graph_mode
        tokens = noise.shape[0] * noise.shape[1] * noise.shape[2] // 4
        if taps is None and mode != "0" and (mode == "1" or tokens <= self.GRAPH_MAX_TOKENS):
            return self._one_step_graph(noise, latent, txt)
        return self._one_step_eager(noise, latent, txt, taps)

    @torch.no_grad()
    def _one_step_graph(self, noise: torch.Tensor, latent: torch.Tensor, txt: torch.Tensor) -> torch.Tensor:
        txt = txt.to(device=self.device, dtype=torch.bfloat16).contiguous()
        key = (tuple(noise.shape), tuple(txt.shape))
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if key not in self._graphs:
            s_noise, s_latent, s_txt = noise.clone(), latent.clone(), txt.clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):        # warm-up: plans, rope tables, kernel attributes, allocator
                for _ in range(2):
                    self._one_step_eager(s_noise, s_latent, s_txt, None)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # thread_local: the frame reader / writer threads may allocate pinned memory while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._one_step_eager(s_noise, s_latent, s_txt, None)
            if len(self._graphs) >= 4:           # every graph keeps its intermediates in a private pool: bound the cache
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = (g, s_noise, s_latent, s_txt, out)
        g, s_noise, s_latent, s_txt, out = self._graphs[key]
        s_noise.copy_(noise)
        s_latent.copy_(latent)
        s_txt.copy_(txt)
        g.replay()
        return out.clone()

    @torch.no_grad()
    def _one_step_eager(self, noise: torch.Tensor, latent: torch.Tensor, txt: torch.Tensor, taps=None) -> torch.Tensor:
        cfg = self.cfg
        T, Hh, Ww, Cc = noise.shape
        if Hh % 2 or Ww % 2:
            raise hip.HipError("latent H and W must be even (patch size 2)")
        if 2 * Cc + 1 != cfg.vid_in_channels:
            raise hip.HipError("latent channel count does not match vid_in_channels")
        grid = (T, Hh // 2, Ww // 2)
        rows = hip.patch_in(noise, latent, self.kpad_in)
        x = hip.gemm(rows, self.w_vid_in, cfg.vid_dim, bias=self.b_vid_in)
        del rows
        txt = txt.to(device=self.device, dtype=torch.bfloat16).contiguous()
        y = hip.gemm(txt, self.w_txt_in, cfg.vid_dim, bias=self.b_txt_in)
        if taps is not None:
            taps["x0"] = x; taps["y0"] = y
        # small grids: the text rows run on a lane of their own (SVR2_DIT_TXT_LANE=0 disables, =1 forces)
        lane_mode = self.txt_lane_mode
        txt_lane = None
        if lane_mode != "0" and (lane_mode == "1" or grid[0] * grid[1] * grid[2] <= self.GRAPH_MAX_TOKENS):
            main = torch.cuda.current_stream(self.device)
            if self._txt_stream is None:
                self._txt_stream = torch.cuda.Stream(device=self.device)
            txt_lane = (self._txt_stream, [])
            self._txt_stream.wait_stream(main)
        for i in range(cfg.num_layers):
            x, y = self.block(i, x, y, grid, taps if (taps is not None and i == 0) else None, txt_lane)
            if taps is not None:
                taps[f"x{i + 1}"] = x
        if txt_lane is not None:
            main.wait_stream(self._txt_stream)      # the lanes join before anything parked in keep is released
            txt_lane[1].clear()
        xo = hip.rmsnorm_mod(x, self.out_scale, self.out_shift, cfg.norm_eps) if cfg.out_norm else x
        pred = hip.gemm(xo, self.w_vid_out, cfg.vid_out_channels * 4, bias=self.b_vid_out)
        return hip.patch_out_x0(pred, noise)


def synthetic_state_dict(cfg: DiTConfig, device, seed: int = 0, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init weights of the right architecture generated on `device` (bench only: there is no
    checkpoint offline).  Same key layout as the reference state dict; fan-in scaled so activations stay O(1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    d, hd, H = cfg.vid_dim, cfg.head_dim, cfg.heads
    out: Dict[str, torch.Tensor] = {}

    def mat(n, k):
        return (torch.randn(n, k, generator=g, device=device, dtype=torch.float32) / math.sqrt(k)).to(dtype)

    def vec(n, mean=0.0, std=0.02):
        return (mean + std * torch.randn(n, generator=g, device=device, dtype=torch.float32)).to(dtype)

    out["vid_in.proj.weight"] = mat(d, cfg.vid_in_channels * 4); out["vid_in.proj.bias"] = vec(d)
    out["txt_in.weight"] = mat(d, cfg.txt_in_dim); out["txt_in.bias"] = vec(d)
    out["emb_in.proj_in.weight"] = mat(d, 256); out["emb_in.proj_in.bias"] = vec(d)
    out["emb_in.proj_hid.weight"] = mat(d, d); out["emb_in.proj_hid.bias"] = vec(d)
    out["emb_in.proj_out.weight"] = mat(6 * d, d); out["emb_in.proj_out.bias"] = vec(6 * d)
    if cfg.out_norm:
        out["vid_out_norm.weight"] = vec(d, 1.0, 0.1)
        out["vid_out_ada.out_shift"] = vec(d, 0.0, 1 / math.sqrt(d)); out["vid_out_ada.out_scale"] = vec(d, 1.0, 1 / math.sqrt(d))
    out["vid_out.proj.weight"] = mat(cfg.vid_out_channels * 4, d); out["vid_out.proj.bias"] = vec(cfg.vid_out_channels * 4)
    for i in range(cfg.num_layers):
        last = cfg.last_vid_only and i == cfg.num_layers - 1
        p = f"blocks.{i}."
        for b in (("all",) if cfg.shared(i) else ("vid", "txt")):
            out[p + f"attn.proj_qkv.{b}.weight"] = mat(3 * H * hd, d)
            out[p + f"attn.proj_out.{b}.weight"] = mat(d, H * hd)
            out[p + f"attn.proj_out.{b}.bias"] = vec(d)
            out[p + f"attn.norm_q.{b}.weight"] = vec(hd, 1.0, 0.1)
            out[p + f"attn.norm_k.{b}.weight"] = vec(hd, 1.0, 0.1)
            if last and b == "txt":
                continue
            if cfg.mlp_type == "swiglu":
                out[p + f"mlp.{b}.proj_in_gate.weight"] = mat(cfg.mlp_hidden, d)
            else:
                out[p + f"mlp.{b}.proj_in.bias"] = vec(cfg.mlp_hidden)
                out[p + f"mlp.{b}.proj_out.bias"] = vec(d)
            out[p + f"mlp.{b}.proj_in.weight"] = mat(cfg.mlp_hidden, d)
            out[p + f"mlp.{b}.proj_out.weight"] = mat(d, cfg.mlp_hidden)
            for l in ("attn", "mlp"):
                out[p + f"ada.{b}.{l}_shift"] = vec(d, 0.0, 1 / math.sqrt(d))
                out[p + f"ada.{b}.{l}_scale"] = vec(d, 1.0, 1 / math.sqrt(d))
                out[p + f"ada.{b}.{l}_gate"] = vec(d, 0.5, 1 / math.sqrt(d))
    return out
