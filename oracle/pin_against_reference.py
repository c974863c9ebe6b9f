#!/usr/bin/env python3
"""Pin the oracle against the importable parts of the reference and write golden vectors.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's Python modules that import cleanly here,
feeds them seeded synthetic inputs, checks that ``oracle/seedvr2_ref`` reproduces them,
and commits the input/output vectors as small fixtures under ``tests/golden/`` so that
the CPU test-suite can re-check the oracle (and the product's host logic) without the
reference.  Fixtures are data only (inputs + expected outputs).

Name-only stubs (no arithmetic) are installed for ``omegaconf``, ``diffusers`` and
``rotary_embedding_torch`` so that modules which merely mention those names can be
imported; anything that would need their arithmetic stays "parity unpinned".

usage:  python oracle/pin_against_reference.py
"""

from __future__ import annotations

import json
import os
import sys
import types

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import numpy as np
import torch
from safetensors.torch import save_file

from seedvr2_ref import dit as odit
from seedvr2_ref import windows as owin


def _install_name_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Name:  # placeholder type, never instantiated on the paths we run
        pass

    mod("omegaconf", DictConfig=type("DictConfig", (dict,), {}), ListConfig=type("ListConfig", (list,), {}),
        OmegaConf=_Name)
    mod("diffusers")
    mod("diffusers.models")
    mod("diffusers.models.normalization", RMSNorm=type("RMSNorm", (torch.nn.Module,), {}))
    mod("rotary_embedding_torch", RotaryEmbedding=_Name, apply_rotary_emb=None)
    # color_fix.py imports two converter NAMES from torchvision (used only by its PIL entry points, never here)
    mod("torchvision")
    mod("torchvision.transforms", ToTensor=_Name, ToPILImage=_Name)


def check(name, a, b, tol=0.0):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ok = a.shape == b.shape and err <= tol
    print(f"  [{'ok' if ok else 'FAIL'}] {name}: shape {tuple(a.shape)} max|d|={err:.3e} (tol {tol:g})")
    if not ok:
        raise SystemExit(f"oracle disagrees with the reference on {name}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    _install_name_stubs()
    torch.manual_seed(0)
    tensors = {}
    meta = {}

    # ------------------------------------------------------------------ G1 windows
    from src.models.dit_3b import window as rwin
    grids = [(1, 32, 32), (3, 68, 120), (5, 135, 240), (6, 135, 240), (17, 135, 240), (2, 24, 40), (3, 20, 36),
             (1, 16, 16), (5, 34, 60), (9, 45, 80), (33, 45, 80)]
    g1 = {}
    for g in grids:
        for meth, ofn in ((owin.REGULAR, owin.regular_windows), (owin.SHIFTED, owin.shifted_windows)):
            ref = [(s[0].start, s[0].stop, s[1].start, s[1].stop, s[2].start, s[2].stop)
                   for s in rwin.get_window_op(meth)(g, (4, 3, 3))]
            mine = ofn(g, (4, 3, 3))
            assert [tuple(x) for x in mine] == ref, (g, meth)
            g1[f"{g[0]}x{g[1]}x{g[2]}|{meth}"] = ref
    print(f"  [ok] G1 window boxes: {len(g1)} (grid, method) cases identical")
    meta["G1_windows"] = g1

    # ------------------------------------------------------------------ G2 na index helpers
    from src.models.dit_3b import na as rna
    g2 = {}
    for g in [(2, 24, 40), (3, 20, 36), (1, 16, 16)]:
        for meth in (owin.REGULAR, owin.SHIFTED):
            op = rwin.get_window_op(meth)
            shape = torch.tensor([list(g)])

            def make_window(x, op=op):
                t, h, w, _ = x.shape
                return [x[st, sh, sw] for (st, sh, sw) in op((t, h, w), (4, 3, 3))]

            part, rev, wshape, wcount = rna.window_idx(shape, make_window)
            L = g[0] * g[1] * g[2]
            ids = torch.arange(L)[:, None].float()
            tgt = part(ids)[:, 0].long()
            idx, lens, shapes = owin.window_token_index(g, owin.windows_for(meth, g))
            check(f"G2 window_idx {g} {meth}", tgt, idx)
            check(f"G2 window shapes {g} {meth}", wshape, shapes)
            assert int(wcount[0]) == len(lens)
            check("G2 window reverse", rev(part(ids)), ids)
            g2[f"{g[0]}x{g[1]}x{g[2]}|{meth}"] = {"idx_sum": int(tgt.sum()), "idx_head": tgt[:64].tolist(),
                                                  "lens": lens.tolist()}
    # repeat_concat_idx + text mean (na.py:320-424)
    vid_len = torch.tensor([3, 2, 4])
    txt_len = torch.tensor([2])
    cat, uncat = rna.repeat_concat_idx(vid_len, txt_len, torch.tensor([3]))
    vid = torch.arange(9).float()[:, None]
    txt = torch.tensor([100.0, 200.0])[:, None]
    allv = cat(vid, txt)
    check("G2 repeat_concat order", allv[:, 0], [0, 1, 2, 100, 200, 3, 4, 100, 200, 5, 6, 7, 8, 100, 200])
    noisy = allv + torch.arange(allv.shape[0]).float()[:, None] * 0.5
    v_out, t_out = uncat(noisy)
    check("G2 unconcat vid", v_out[:, 0], [0, 1.5, 3, 5.5, 7, 9.5, 11, 12.5, 14])
    check("G2 unconcat txt mean", t_out[:, 0], [100 + (1.5 + 3.5 + 6.5) / 3, 200 + (2 + 4 + 7) / 3])
    meta["G2_index"] = g2

    # ------------------------------------------------------------------ G3 varlen attention
    from src.models.dit_3b.attention import pytorch_varlen_attention
    lens = [37, 58 + 13, 128, 5]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    q = torch.randn(sum(lens), 2, 128)
    k = torch.randn(sum(lens), 2, 128)
    v = torch.randn(sum(lens), 2, 128)
    ref = pytorch_varlen_attention(q, k, v, cu, cu)
    mine = []
    off = 0
    for n in lens:
        qs, ks, vs = (t[off:off + n].permute(1, 0, 2) for t in (q, k, v))
        pr = torch.softmax((qs @ ks.transpose(1, 2)) / 128 ** 0.5, dim=-1)
        mine.append((pr @ vs).permute(1, 0, 2))
        off += n
    check("G3 varlen attention (fp32)", torch.cat(mine), ref, 2e-6)
    tensors.update({"G3.q": q, "G3.k": k, "G3.v": v, "G3.cu": cu, "G3.out": ref})

    # ------------------------------------------------------------------ G4 AdaSingle + cache collision
    from src.common.cache import Cache
    from src.models.dit_3b.modulation import AdaSingle
    dim = 32
    emb = torch.randn(1, 6 * dim)
    blk = AdaSingle(dim, 6 * dim, layers=["attn", "mlp"])
    outada = AdaSingle(dim, 6 * dim, layers=["out"], modes=["in"])
    hid = torch.randn(7, dim)
    cache = Cache()
    hl = torch.tensor([7])
    r_in = blk(hid.clone(), emb, layer="attn", mode="in", cache=cache, branch_tag="vid", hid_len=hl)
    r_mlp = blk(hid.clone(), emb, layer="mlp", mode="in", cache=cache, branch_tag="vid", hid_len=hl)
    r_gate = blk(hid.clone(), emb, layer="mlp", mode="out", cache=cache, branch_tag="vid", hid_len=hl)
    r_out = outada(hid.clone(), emb, layer="out", mode="in", cache=cache, branch_tag="vid", hid_len=hl)
    sh0, sc0, ga0 = odit.ada_slices(emb[0], 0)
    sh1, sc1, ga1 = odit.ada_slices(emb[0], 1)
    check("G4 ada attn in", hid * (sc0 + blk.attn_scale) + (sh0 + blk.attn_shift), r_in.detach(), 1e-6)
    check("G4 ada mlp in", hid * (sc1 + blk.mlp_scale) + (sh1 + blk.mlp_shift), r_mlp.detach(), 1e-6)
    check("G4 ada mlp gate", hid * (ga1 + blk.mlp_gate), r_gate.detach(), 1e-6)
    check("G4 vid_out_ada (cache collision -> attn slice)", hid * (sc0 + outada.out_scale) + (sh0 + outada.out_shift),
          r_out.detach(), 1e-6)
    tensors.update({"G4.emb": emb, "G4.hid": hid, "G4.attn_shift": blk.attn_shift.detach(),
                    "G4.attn_scale": blk.attn_scale.detach(), "G4.mlp_gate": blk.mlp_gate.detach(),
                    "G4.out_shift": outada.out_shift.detach(), "G4.out_scale": outada.out_scale.detach(),
                    "G4.r_in": r_in.detach(), "G4.r_gate": r_gate.detach(), "G4.r_out": r_out.detach()})

    # ------------------------------------------------------------------ G6 patch in/out, G7 rmsnorm
    from src.models.dit_3b.patch.patch_v1 import NaPatchIn, NaPatchOut
    pin = NaPatchIn(in_channels=5, patch_size=(1, 2, 2), dim=16)
    pout = NaPatchOut(out_channels=3, patch_size=(1, 2, 2), dim=16)
    shp = torch.tensor([[3, 4, 6]])
    xin = torch.randn(3 * 4 * 6, 5)
    c = Cache()
    yref, gshape = pin(xin, shp, c)
    ymine = torch.nn.functional.linear(odit.patch_in(xin, (3, 4, 6), (1, 2, 2)), pin.proj.weight, pin.proj.bias)
    check("G6 NaPatchIn", ymine.detach(), yref.detach(), 1e-6)
    zref, _ = pout(yref, gshape, c)
    zmine = odit.patch_out(torch.nn.functional.linear(yref, pout.proj.weight, pout.proj.bias), (3, 2, 3), (1, 2, 2), 3)
    check("G6 NaPatchOut", zmine.detach(), zref.detach(), 1e-6)
    tensors.update({"G6.x": xin, "G6.win": pin.proj.weight.detach(), "G6.bin": pin.proj.bias.detach(),
                    "G6.y": yref.detach(), "G6.wout": pout.proj.weight.detach(), "G6.bout": pout.proj.bias.detach(),
                    "G6.z": zref.detach()})

    from src.models.dit_3b.normalization import CustomRMSNorm
    rn = CustomRMSNorm(128, eps=1e-5, elementwise_affine=True)
    with torch.no_grad():
        rn.weight.copy_(1 + 0.1 * torch.randn(128))
    xr = torch.randn(9, 4, 128)
    check("G7 CustomRMSNorm fp32", odit.rms_norm(xr, 1e-5, rn.weight.detach()), rn(xr).detach(), 1e-6)
    check("G7 CustomRMSNorm bf16", odit.rms_norm(xr.bfloat16(), 1e-5, rn.weight.detach().bfloat16()).float(),
          rn.bfloat16()(xr.bfloat16()).detach().float(), 0.0)
    rn.float()
    tensors.update({"G7.x": xr, "G7.w": rn.weight.detach().float(), "G7.y": rn(xr).detach()})

    # ------------------------------------------------------------------ G5/G8 a full transformer block (no RoPE)
    # mmsr_block + mmattn + mlp + mm + modulation + normalization run as-is; rope_type=None because the
    # RoPE arithmetic lives in the absent rotary_embedding_torch.
    from src.models.dit_3b.nablocks.mmsr_block import NaMMSRTransformerBlock
    from src.models.dit_3b.modulation import get_ada_layer
    from src.models.dit_3b.normalization import get_norm_layer
    for tag, shared, last, meth in (("sep", False, False, owin.REGULAR), ("shared_last", True, True, owin.SHIFTED)):
        torch.manual_seed(7)
        cfg = odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=1 if last else 2,
                             mm_layers=0 if shared else 1, mlp_multiple_of=32)
        blk = NaMMSRTransformerBlock(
            vid_dim=64, txt_dim=64, emb_dim=384, heads=2, head_dim=32, expand_ratio=4,
            norm=get_norm_layer("fusedrms"), norm_eps=1e-5, ada=get_ada_layer("single"), qk_bias=False,
            qk_norm=get_norm_layer("fusedrms"), mlp_type="swiglu", shared_weights=shared, rope_type=None,
            rope_dim=32, is_last_layer=last, window=(4, 3, 3), window_method=meth)
        with torch.no_grad():
            for n_, p_ in blk.named_parameters():
                if "norm_" in n_:
                    p_.copy_(1 + 0.1 * torch.randn_like(p_))
        grid = (2, 12, 20)
        L = grid[0] * grid[1] * grid[2]
        vid = torch.randn(L, 64)
        txt = torch.randn(58, 64)
        emb = torch.randn(1, 384)
        with torch.no_grad():
            v_ref, t_ref, _, _ = blk(vid.clone(), txt.clone(), torch.tensor([list(grid)]), torch.tensor([[58]]),
                                     emb, Cache())
        # same block through the oracle's forward() by wrapping it as a 1-layer model
        p = {f"blocks.0.{k}": v.detach() for k, v in blk.state_dict().items()}
        v_mine, t_mine, _ = odit.block(p, cfg, 0, vid, txt, emb[0], grid, torch.float32, "none", method=meth)
        check(f"G8 transformer block [{tag}] vid", v_mine, v_ref, 2e-5)
        if not last:
            check(f"G8 transformer block [{tag}] txt", t_mine, t_ref, 2e-5)
        for k, v in p.items():
            tensors[f"G8.{tag}.p.{k}"] = v.contiguous()
        tensors.update({f"G8.{tag}.vid": vid, f"G8.{tag}.txt": txt, f"G8.{tag}.emb": emb,
                        f"G8.{tag}.vid_out": v_ref, f"G8.{tag}.txt_out": t_ref})
        meta[f"G8.{tag}"] = {"grid": list(grid), "method": meth, "shared": shared, "last": last,
                             "cfg": {"vid_dim": 64, "heads": 2, "head_dim": 32, "num_layers": cfg.num_layers,
                                     "mm_layers": cfg.mm_layers, "mlp_multiple_of": 32}}

    # ------------------------------------------------------------------ G11 the 7B transformer block (RoPE off)
    from src.models.dit_7b.nablocks.mmsr_block import NaMMSRTransformerBlock as Block7B
    from src.models.dit_7b.modulation import get_ada_layer as get_ada7
    from src.models.dit_7b.normalization import get_norm_layer as get_norm7
    torch.manual_seed(13)
    cfg7 = odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=2, mm_layers=2, mlp_type="normal",
                          out_norm=False, last_vid_only=False)
    blk7 = Block7B(vid_dim=64, txt_dim=64, emb_dim=384, heads=2, head_dim=32, expand_ratio=4,
                   norm=get_norm7("fusedrms"), norm_eps=1e-5, ada=get_ada7("single"), qk_bias=False, qk_rope=False,
                   qk_norm=get_norm7("fusedrms"), shared_qkv=False, shared_mlp=False, mlp_type="normal",
                   window=(4, 3, 3), window_method=owin.SHIFTED)
    with torch.no_grad():
        for n_, p_ in blk7.named_parameters():
            if "norm_" in n_:
                p_.copy_(1 + 0.1 * torch.randn_like(p_))
    grid7 = (3, 10, 14)
    vid7, txt7, emb7 = torch.randn(3 * 10 * 14, 64), torch.randn(58, 64), torch.randn(1, 384)
    with torch.no_grad():
        v7, t7, _, _ = blk7(vid7.clone(), txt7.clone(), torch.tensor([list(grid7)]), torch.tensor([[58]]), emb7,
                            Cache(disable=True))
    p7 = {f"blocks.0.{k}": v.detach() for k, v in blk7.state_dict().items()}
    v7m, t7m, _ = odit.block(p7, cfg7, 0, vid7, txt7, emb7[0], grid7, torch.float32, "none", method=owin.SHIFTED)
    check("G11 7B transformer block vid", v7m, v7, 2e-5)
    check("G11 7B transformer block txt", t7m, t7, 2e-5)
    for k, v in p7.items():
        tensors[f"G11.p.{k}"] = v.contiguous()
    tensors.update({"G11.vid": vid7, "G11.txt": txt7, "G11.emb": emb7, "G11.vid_out": v7, "G11.txt_out": t7})
    meta["G11"] = {"grid": list(grid7), "method": owin.SHIFTED}

    # ------------------------------------------------------------------ G9 one-step Euler endpoint
    from src.common.diffusion.schedules.lerp import LinearInterpolationSchedule
    from src.common.diffusion.samplers.euler import EulerSampler
    from src.common.diffusion.timesteps.sampling.trailing import UniformTrailingSamplingTimesteps
    from src.common.diffusion.types import PredictionType
    sched = LinearInterpolationSchedule(T=1000.0)
    ts = UniformTrailingSamplingTimesteps(T=sched.T, steps=1)
    smp = EulerSampler(schedule=sched, timesteps=ts, prediction_type="v_lerp")
    x = torch.randn(11, 16)
    seen = {}

    def f(args):
        seen["t"] = args.t.clone()
        return 0.3 * args.x_t + 1.0

    res = smp.sample(x=x, f=f)
    check("G9 timesteps == [1000]", ts.timesteps, [1000.0])
    check("G9 one-step endpoint x0 = x - f(x)", x - (0.3 * x + 1.0), res, 1e-6)
    meta["G9"] = {"t_seen": float(seen["t"])}

    # ------------------------------------------------------------------ G10 causal conv / norm of the VAE
    from seedvr2_ref import vae as ovae
    from src.models.video_vae_v3.modules import causal_inflation_lib as cil
    from src.models.video_vae_v3.modules.types import MemoryState
    vt = {}
    for tag, k, s, pad in (("k3s1", 3, 1, 1), ("k3s2", 3, 2, 1), ("k133", (1, 3, 3), (1, 2, 2), (0, 0, 0)),
                           ("k1", 1, 1, 0)):
        torch.manual_seed(11)
        conv = cil.init_causal_conv3d(8, 12, k, stride=s, padding=pad, inflation_mode="pad")
        conv.eval()
        x = torch.randn(1, 8, 9, 10, 12)
        name = "c"
        pp = {"c.weight": conv.weight.detach(), "c.bias": conv.bias.detach()}
        sp = (0, 0, 0, 0) if tag in ("k133", "k1") else (1, 1, 1, 1)
        st3 = (s, s, s) if isinstance(s, int) else s
        with torch.no_grad():
            full = conv(x, memory_state=MemoryState.DISABLED)
            mine = ovae.causal_conv(pp, name, x[0], stride=st3, pad_hw=sp)
            check(f"G10 causal conv {tag} unsliced", mine, full[0], 1e-5)
            for limit in (float("inf"), 1e-5):   # 1e-5 GiB forces the recursive H/W splitting path
                conv.set_memory_limit(limit)
                outs = [conv(x[:, :, :5].clone(), memory_state=MemoryState.INITIALIZING)]
                for a in (5, 7):
                    outs.append(conv(x[:, :, a:a + 2].clone(), memory_state=MemoryState.ACTIVE))
                sliced = torch.cat(outs, dim=2)
                conv.memory = None
                check(f"G10 reference sliced == unsliced {tag} limit={limit}", sliced, full, 1e-5)
            sto = ovae.State()
            mo = [ovae.causal_conv(pp, name, x[0, :, :5], stride=st3, pad_hw=sp, st=sto)]
            sto.active = True
            for a in (5, 7):
                mo.append(ovae.causal_conv(pp, name, x[0, :, a:a + 2], stride=st3, pad_hw=sp, st=sto))
            check(f"G10 oracle sliced == reference {tag}", torch.cat(mo, dim=1), full[0], 1e-5)
        vt.update({f"G10.{tag}.x": x[0], f"G10.{tag}.w": pp["c.weight"], f"G10.{tag}.b": pp["c.bias"],
                   f"G10.{tag}.y": full[0]})
    gn = torch.nn.GroupNorm(4, 8, eps=1e-6)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.1 * torch.randn(8)); gn.bias.copy_(0.1 * torch.randn(8))
        xg = torch.randn(1, 8, 3, 6, 5)
        ref = cil.causal_norm_wrapper(gn, xg)
        mine = ovae.group_norm_per_frame({"n.weight": gn.weight, "n.bias": gn.bias}, "n", xg[0], 4, 1e-6)
        check("G10 causal_norm_wrapper (per-frame GroupNorm)", mine, ref[0], 1e-6)
        xh = torch.randn(1, 2, 6, 3, 3)
        check("G10 remove_head", torch.cat((xh[0][:, :1], xh[0][:, 2:]), dim=1), cil.remove_head(xh)[0])
    vt.update({"G10.gn.x": xg[0], "G10.gn.w": gn.weight.detach(), "G10.gn.b": gn.bias.detach(), "G10.gn.y": ref[0]})
    save_file({k: v.contiguous() for k, v in vt.items()}, os.path.join(GOLD, "vae_pins.safetensors"))

    # ------------------------------------------------------------------ G12 phase-4 colour correction (next-2)
    import importlib
    cf = importlib.import_module("src.utils.color_fix")
    from seedvr2_ref import color as ocol

    class _Dbg:
        def log(self, *a, **k):
            pass

    ct = {}
    g = torch.Generator().manual_seed(77)
    base = torch.rand(2, 3, 6, 8, generator=g)
    style32 = torch.nn.functional.interpolate(base, size=(40, 56), mode="bicubic").clamp(0, 1) * 2 - 1
    content32 = (style32 * 0.8 + 0.1 + 0.05 * torch.randn(2, 3, 40, 56, generator=g)).clamp(-1, 1)
    content32[:, 0] = (content32[:, 0] * 1.3).clamp(-1, 1)      # oversaturate one channel: exercises the adaptive blend
    for tag, dt, tol in (("f32", torch.float32, 1e-6), ("bf16", torch.bfloat16, 0.0)):
        c, s_ = content32.to(dt), style32.to(dt)
        r_ad = cf.adaptive_instance_normalization(c.clone(), s_.clone())
        check(f"G12 adain {tag}", ocol.adaptive_instance_normalization(c.clone(), s_.clone()).float(), r_ad.float(), tol)
        r_wv = cf.wavelet_reconstruction(c.clone(), s_.clone(), _Dbg())
        check(f"G12 wavelet_reconstruction {tag}", ocol.wavelet_reconstruction(c.clone(), s_.clone()).float(), r_wv.float(), tol)
        r_lab = cf.lab_color_transfer(c.clone(), s_.clone(), _Dbg(), luminance_weight=0.8)
        check(f"G12 lab_color_transfer {tag}", ocol.lab_color_transfer(c.clone(), s_.clone()).float(), r_lab.float(),
              2e-6 if dt == torch.float32 else 0.0)
        r_hsv = cf.hsv_saturation_histogram_match(c.clone(), s_.clone(), _Dbg())
        check(f"G12 hsv_saturation_histogram_match {tag}", ocol.hsv_saturation_histogram_match(c.clone(), s_.clone()).float(),
              r_hsv.float(), 2e-6 if dt == torch.float32 else 0.0)
        r_wa = cf.wavelet_adaptive_color_correction(c.clone(), s_.clone(), _Dbg())
        check(f"G12 wavelet_adaptive {tag}", ocol.wavelet_adaptive_color_correction(c.clone(), s_.clone()).float(),
              r_wa.float(), 2e-6 if dt == torch.float32 else 0.0)
        ct.update({f"G12.{tag}.content": c, f"G12.{tag}.style": s_, f"G12.{tag}.adain": r_ad, f"G12.{tag}.wavelet": r_wv,
                   f"G12.{tag}.lab": r_lab, f"G12.{tag}.hsv": r_hsv, f"G12.{tag}.wavelet_adaptive": r_wa})
    lab = cf._rgb_to_lab_batch((style32 + 1) / 2, torch.device("cpu"), torch.tensor(ocol._RGB2XYZ), ocol.EPS, ocol.KAPPA)
    check("G12 rgb->lab", ocol.rgb_to_lab((style32 + 1) / 2), lab, 1e-5)
    ct["G12.lab_of_style"] = lab
    save_file({k: v.contiguous() for k, v in ct.items()}, os.path.join(GOLD, "color_pins.safetensors"))

    # ------------------------------------------------------------------ G13 spatial VAE tiling (next-4)
    # attn_video_vae.py imports diffusers at module level, so the module cannot be imported here; its two tiling
    # methods depend on nothing but torch, so they are taken out of the file's syntax tree and run as they are, bound to
    # a stand-in `self` that supplies a synthetic tile codec for slicing_encode / slicing_decode.
    import ast
    from typing import Tuple
    from seedvr2_ref import tiling as otile
    vae_src = os.path.join(REF, "src/models/video_vae_v3/modules/attn_video_vae.py")
    with open(vae_src) as fh:
        tree = ast.parse(fh.read())
    fns = [n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "VideoAutoencoderKL"
           for n in c.body if isinstance(n, ast.FunctionDef) and n.name in ("tiled_encode", "tiled_decode")]
    assert len(fns) == 2
    ns = {"torch": torch, "Tuple": Tuple}
    exec(compile(ast.Module(body=fns, type_ignores=[]), vae_src, "exec"), ns)

    enc_codec, dec_codec = otile.synthetic_enc_codec, otile.synthetic_dec_codec

    class _FakeVAE:
        spatial_downsample_factor = 8
        debug = None
        slicing_encode = staticmethod(enc_codec)
        slicing_decode = staticmethod(dec_codec)

    tt, tmeta = {}, {}
    g = torch.Generator().manual_seed(91)
    enc_cases = [((3, 5, 48, 64), (32, 32), (8, 8)), ((3, 5, 48, 64), (32, 40), (16, 12)), ((3, 1, 40, 72), (24, 24), (23, 9)),
                 ((3, 5, 32, 32), (64, 64), (16, 16)), ((3, 5, 56, 48), (16, 48), (0, 24)), ((3, 5, 48, 80), (32, 32), (40, 40))]
    dec_cases = [((4, 2, 6, 8), (32, 32), (8, 8)), ((4, 2, 6, 8), (32, 40), (13, 20)), ((4, 1, 5, 9), (24, 24), (23, 9)),
                 ((4, 2, 4, 4), (64, 64), (16, 16)), ((4, 2, 7, 6), (16, 48), (0, 24)), ((4, 2, 6, 10), (32, 32), (40, 40))]
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for i, (shape, ts, ov) in enumerate(enc_cases):
            x = torch.randn(*shape, generator=g).to(dt)
            ref = ns["tiled_encode"](_FakeVAE(), x[None].clone(), tile_size=ts, tile_overlap=ov)[0]
            mine = otile.tiled_encode(x.clone(), enc_codec, ts, ov)
            ref = ref.reshape(mine.shape)      # single-frame inputs come back without the frame axis (:1455-1456)
            check(f"G13 tiled_encode {tag} #{i} {shape} tile {ts} overlap {ov}", mine.float(), ref.float(), 0.0)
            tt[f"G13.enc.{tag}.{i}.x"], tt[f"G13.enc.{tag}.{i}.out"] = x, ref
            tmeta[f"enc.{i}"] = {"tile": ts, "overlap": ov}
        for i, (shape, ts, ov) in enumerate(dec_cases):
            z = torch.randn(*shape, generator=g).to(dt)
            ref = ns["tiled_decode"](_FakeVAE(), z[None].clone(), tile_size=ts, tile_overlap=ov)[0]
            mine = otile.tiled_decode(z.clone(), dec_codec, ts, ov)
            ref = ref.reshape(mine.shape)
            check(f"G13 tiled_decode {tag} #{i} {shape} tile {ts} overlap {ov}", mine.float(), ref.float(), 0.0)
            tt[f"G13.dec.{tag}.{i}.z"], tt[f"G13.dec.{tag}.{i}.out"] = z, ref
            tmeta[f"dec.{i}"] = {"tile": ts, "overlap": ov}
    save_file({k: v.contiguous() for k, v in tt.items()}, os.path.join(GOLD, "tiling_pins.safetensors"),
              metadata={"cases": json.dumps(tmeta)})

    # ------------------------------------------------------------------ G14 edge-guided alpha upscaling (next-4)
    # alpha_upscaling.py needs cv2 for two calls (RGB2GRAY on uint8, 3x3 Sobel in float64); the stand-in below serves
    # them from the oracle's restatement of OpenCV's documented arithmetic (unpinned), everything else is the reference.
    from seedvr2_ref import alpha as oalpha
    mod_cv2 = types.ModuleType("cv2")
    mod_cv2.COLOR_RGB2GRAY, mod_cv2.CV_64F = 7, 6
    mod_cv2.cvtColor = lambda frame, code: oalpha.rgb_to_gray_u8(frame)
    mod_cv2.Sobel = lambda gray, ddepth, dx, dy, ksize=3: oalpha.sobel_f64(gray, dx, dy)
    sys.modules["cv2"] = mod_cv2
    au = importlib.import_module("src.core.alpha_upscaling")
    at = {}
    g = torch.Generator().manual_seed(93)
    yy, xx = torch.meshgrid(torch.arange(12.0), torch.arange(16.0), indexing="ij")
    blob = ((yy - 5.5) ** 2 / 14 + (xx - 8) ** 2 / 30 < 1).float()
    a_bin = torch.stack([blob, blob.roll(2, 1)])[:, None]                                   # binary mask
    a_grad = torch.stack([(xx / 15), ((yy / 11) * (xx / 15)).sqrt()])[:, None].contiguous()  # gradient alpha
    small = torch.rand(2, 3, 12, 16, generator=g)
    for tag, a_in in (("binary", a_bin), ("gradient", a_grad), ("matte", a_bin)):
        # the upscaled RGB shows the matte's edge (like a real cut-out) plus texture, in [-1, 1]; "matte" is a clean
        # cut-out (flat foreground on a faint gradient): edges only along the matte, so the snapping rules dominate
        if tag == "matte":
            up = torch.nn.functional.interpolate(a_in, size=(48, 64), mode="bilinear")
            bg = torch.linspace(0.1, 0.2, 64).view(1, 1, 1, 64).expand(2, 3, 48, 64)
            rgb = (up * torch.tensor([0.9, 0.6, 0.3]).view(1, 3, 1, 1) + (1 - up) * bg) * 2 - 1
        else:
            rgb = torch.nn.functional.interpolate(small * 0.6 + a_in * 0.4, size=(48, 64), mode="bicubic").clamp(0, 1) * 2 - 1
        rgb = rgb.to(torch.bfloat16)
        ref = au.edge_guided_alpha_upscale(a_in.to(torch.bfloat16), torch.zeros(2, 3, 12, 16), rgb, method="guided", debug=None)
        mine = oalpha.edge_guided_alpha_upscale(a_in.to(torch.bfloat16), rgb)
        check(f"G14 edge_guided_alpha_upscale {tag}", mine, ref, 0.0)
        at[f"G14.{tag}.alpha"], at[f"G14.{tag}.rgb"], at[f"G14.{tag}.out"] = a_in.to(torch.bfloat16), rgb, ref
    edges_ref = au.detect_edges_batch(images=(at["G14.binary.rgb"].float() + 1) / 2, method="sobel", debug=None)
    check("G14 detect_edges_batch", oalpha.detect_edges((at["G14.binary.rgb"].float() + 1) / 2), edges_ref, 0.0)
    at["G14.binary.edges"] = edges_ref
    save_file({k: v.contiguous() for k, v in at.items()}, os.path.join(GOLD, "alpha_pins.safetensors"))

    # ------------------------------------------------------------------ G15 latent blur = schedule.forward (noise knobs)
    from src.common.diffusion.schedules.lerp import LinearInterpolationSchedule
    from seedvr2_ref import pipeline as opipe
    sch = LinearInterpolationSchedule(T=1000.0)
    g = torch.Generator().manual_seed(95)
    lat = torch.randn(2, 4, 6, 16, generator=g).to(torch.bfloat16)
    n1, n2 = (torch.randn(2, 4, 6, 16, generator=g).to(torch.bfloat16) for _ in range(2))
    for sc in (0.1, 0.5, 1.0):
        t = opipe.blur_timestep(lat.shape, sc)                     # timestep_transform lives in core/infer.py (not importable)
        ref = sch.forward(lat, n1 * 0.1 + n2 * 0.05, t).to(torch.bfloat16)
        check(f"G15 latent_blur scale {sc}", opipe.latent_blur(lat, n1, n2, sc).float(), ref.float(), 0.0)

    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(GOLD, "dit_pins.safetensors"))
    with open(os.path.join(GOLD, "dit_pins.json"), "w") as fh:
        json.dump(meta, fh)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
