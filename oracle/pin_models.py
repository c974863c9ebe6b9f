#!/usr/bin/env python3
"""Pin the oracle's MODEL-LEVEL forwards against the reference's own model files (G16, G17, G10b).

Runs ONLY in the build container (needs /root/reference).  With the stand-ins of ``ref_standins.py`` for the
absent third-party packages, the reference's model files are imported and run AS THEY ARE:

  G16  src/models/video_vae_v3/modules/attn_video_vae.py -- ``VideoAutoencoderKLWrapper`` built from the
       reference's own YAML (s8_c16_t4_inflation_sd3.yaml) and configured like model_configuration.py:1237-1248
       (set_causal_slicing(split_size=4, memory_device='same'), set_memory_limit(0.5, 0.5)): its encode / decode run
       Encoder3D / Decoder3D / ResnetBlock3D / Upsample3D (einops rearrange) / Downsample3D / UNetMidBlock3D /
       slicing_encode / slicing_decode / _encode / _decode, sliced and unsliced.  Also pins the VAE's state-dict
       key set and shapes (the oracle's / product's ``param_shapes``).
  G17  src/models/dit_3b/nadit.py and src/models/dit_7b/nadit.py -- the WHOLE ``NaDiT.forward`` with RoPE on
       (layer order, mm_layers split, shared / separate weights, vid_out_ada cache-key collision in situ, vid-only
       last layer, 7B per-layer rebuilds), state-dict keys and shapes.
  G10b src/core/generation_utils.py -- ``pad_video_temporal`` and ``blend_overlapping_frames`` taken out of the
       file's syntax tree (the module imports torchvision) and run as they are; src/core/generation_phases.py
       ``_apply_4n1_padding`` likewise.

Fixtures (data only: inputs, weights of the small seeded models, expected outputs):
  tests/golden/vae_model_pins.safetensors, dit_model_pins.safetensors, model_pins.json

usage:  python oracle/pin_models.py
"""

from __future__ import annotations

import ast
import json
import os
import sys

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import torch
import yaml
from safetensors.torch import save_file

import ref_standins
from seedvr2_ref import dit as odit
from seedvr2_ref import pipeline as opipe
from seedvr2_ref import vae as ovae


def check(name, a, b, tol=0.0):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ok = a.shape == b.shape and err <= tol
    print(f"  [{'ok' if ok else 'FAIL'}] {name}: shape {tuple(a.shape)} max|d|={err:.3e} (tol {tol:g})")
    if not ok:
        raise SystemExit(f"oracle disagrees with the reference on {name}")


def bf16_values_(module):
    """Round every parameter to a bf16-representable value in place (fixtures store bf16: half the bytes)."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(p.to(torch.bfloat16).float())


# ================================================================================================ G16: the VAE
def build_reference_vae(block_out_channels, norm_groups, seed):
    from src.models.video_vae_v3.modules import attn_video_vae as rvae
    with open(os.path.join(REF, "src/models/video_vae_v3/s8_c16_t4_inflation_sd3.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    with open(os.path.join(REF, "configs_3b/main.yaml")) as fh:
        main = yaml.safe_load(fh)
    cfg["block_out_channels"] = list(block_out_channels)
    cfg["norm_num_groups"] = norm_groups
    torch.manual_seed(seed)
    vae = rvae.VideoAutoencoderKLWrapper(**cfg, freeze_encoder=True)
    vae.eval()
    with torch.no_grad():
        for n_, p_ in vae.named_parameters():
            if "norm" in n_:                       # GroupNorm affine params are 1 / 0 at init: make them matter
                p_.copy_((1 + 0.1 * torch.randn_like(p_)) if n_.endswith("weight") else 0.1 * torch.randn_like(p_))
            elif n_.endswith(".bias"):
                p_.copy_(0.05 * torch.randn_like(p_))
            elif "upscale_conv.weight" in n_:      # identity-initialised (attn_video_vae.py:98-104): perturb
                p_.add_(0.1 * torch.randn_like(p_))
    bf16_values_(vae)
    # model_configuration.py:1237-1248 with configs_3b/main.yaml:53-58
    vae.set_causal_slicing(**main["vae"]["slicing"])
    vae.set_memory_limit(**main["vae"]["memory_limit"])
    return vae, cfg, main


def pin_vae(tensors, meta):
    print("G16 VideoAutoencoderKLWrapper (reference module run as is)")
    for tag, ch, groups, thw, keep in (("small", (16, 32, 32, 32), 8, (13, 24, 32), True),
                                       ("fullwidth", (128, 256, 512, 512), 32, (9, 16, 24), False)):
        vae, cfg, main = build_reference_vae(ch, groups, seed=21)
        ocfg = ovae.VAEConfig(block_out_channels=tuple(ch), norm_groups=groups,
                              scaling_factor=main["vae"]["scaling_factor"])
        sd = {k: v.detach().clone() for k, v in vae.state_dict().items()}
        want = ovae.param_shapes(ocfg)
        assert set(sd) == set(want), (sorted(set(sd) ^ set(want))[:10])
        assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
        print(f"  [ok] G16[{tag}] state-dict: {len(sd)} keys and shapes identical to the oracle's param_shapes")
        T, H, W = thw
        g = torch.Generator().manual_seed(22)
        x = (torch.rand(1, 3, T, H, W, generator=g) * 2 - 1).to(torch.bfloat16).float()
        with torch.no_grad():
            enc = vae.encode(x)                        # slicing_encode: 1+4, 4, 4 ... (attn_video_vae.py:1246-1268)
            mom_ref, z_ref = enc.posterior.parameters, enc.latent
            dec_ref = vae.decode(z_ref).sample         # slicing_decode: 2, 1, 1 ...   (:1270-1292)
            vae.disable_slicing()
            mom_whole = vae.encode(x).posterior.parameters
            dec_whole = vae.decode(z_ref).sample
            vae.enable_slicing()
        check(f"G16[{tag}] reference sliced == unsliced (encode)", mom_ref, mom_whole, 2e-5)
        check(f"G16[{tag}] reference sliced == unsliced (decode)", dec_ref, dec_whole, 2e-5)
        with torch.no_grad():
            mom = ovae.encoder(sd, x[0], ocfg)
            check(f"G16[{tag}] Encoder3D moments", mom, mom_ref[0], 3e-5)
            check(f"G16[{tag}] posterior mode = first {ocfg.latent_channels} channels", mom[: ocfg.latent_channels], z_ref[0], 3e-5)
            check(f"G16[{tag}] oracle encode_sliced", ovae.encode_sliced(sd, x[0], ocfg, split=4), mom_ref[0], 3e-5)
            dec = ovae.decoder(sd, z_ref[0], ocfg)
            check(f"G16[{tag}] Decoder3D", dec, dec_ref[0], 5e-5)
            check(f"G16[{tag}] oracle decode_sliced", ovae.decode_sliced(sd, z_ref[0], ocfg, split=1), dec_ref[0], 5e-5)
            # the runner's scaling (infer.py:117-188, 192-257) on top: encode() / decode() of the oracle
            lat = ovae.encode(sd, x[0], ocfg)
            check(f"G16[{tag}] oracle.encode = mode * scaling_factor, channels-last", lat,
                  (z_ref[0] * ocfg.scaling_factor).permute(1, 2, 3, 0), 3e-5)
        if keep:
            for k, v in sd.items():
                tensors[f"G16.p.{k}"] = v.to(torch.bfloat16).contiguous()
            tensors.update({"G16.x": x[0].contiguous(), "G16.moments": mom_ref[0].contiguous(),
                            "G16.z": z_ref[0].contiguous(), "G16.dec": dec_ref[0].contiguous()})
            meta["G16"] = {"block_out_channels": list(ch), "norm_groups": groups, "input_tchw": [T, 3, H, W],
                           "scaling_factor": ocfg.scaling_factor, "slicing": main["vae"]["slicing"],
                           "memory_limit": main["vae"]["memory_limit"]}
    meta["G16.keys_fullwidth"] = {k: list(v) for k, v in ovae.param_shapes(ovae.VAEConfig()).items()}


# ================================================================================================ G17: NaDiT
def _expand(n, v):
    return [v] * n


def build_reference_dit(model, overrides, seed):
    """NaDiT from the reference's YAML (the ${...} entries are the yaml's own eval expressions, spelled out)."""
    with open(os.path.join(REF, f"configs_{model}/main.yaml")) as fh:
        cfg = yaml.safe_load(fh)["dit"]["model"]
    cfg.pop("__object__")
    cfg["norm_eps"] = float(cfg["norm_eps"])     # '1e-5' is a float for OmegaConf's resolver, a string for YAML 1.1
    cfg.update(overrides)
    n = cfg["num_layers"]
    cfg["txt_dim"] = cfg["vid_dim"]
    cfg["emb_dim"] = 6 * cfg["vid_dim"]
    cfg["block_type"] = _expand(n, "mmdit_sr")
    cfg["window"] = _expand(n, (4, 3, 3))
    cfg["window_method"] = (n // 2) * ["720pwin_by_size_bysize", "720pswin_by_size_bysize"] + \
                           (["720pwin_by_size_bysize"] if n % 2 else [])
    if cfg.get("msa_type") == "None":
        cfg["msa_type"] = None
    import importlib
    rmod = importlib.import_module(f"src.models.dit_{model}.nadit")
    torch.manual_seed(seed)
    net = rmod.NaDiT(**cfg)
    net.eval()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if "norm" in n_:
                p_.copy_(1 + 0.1 * torch.randn_like(p_))
            elif n_.split(".")[-1].endswith(("_shift", "_scale", "_gate")):
                base = {"shift": 0.0, "scale": 1.0, "gate": 0.5}[n_.rsplit("_", 1)[-1]]
                p_.copy_(base + 0.3 * torch.randn_like(p_))
            elif n_.endswith(".bias"):
                p_.copy_(0.05 * torch.randn_like(p_))
    bf16_values_(net)
    return net, cfg


def pin_dit(tensors, meta):
    print("G17 NaDiT.forward (reference modules run as they are, RoPE on)")
    cases = (
        # 3B flavour: 4 layers, mm_layers 2 -> separate, separate, shared, shared + vid-only last; grid (2,12,20)
        ("3b", dict(vid_dim=64, heads=2, head_dim=32, rope_dim=32, num_layers=4, mm_layers=2, txt_in_dim=48), (2, 24, 40), "3b",
         odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=4, mm_layers=2, txt_in_dim=48)),
        # 7B flavour: 3 layers (odd count exercises the window-method alternation), all separate, GELU MLP, pixel rope
        ("7b", dict(vid_dim=64, heads=2, head_dim=32, num_layers=3, txt_in_dim=48), (3, 20, 28), "7b",
         odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=3, mm_layers=3, txt_in_dim=48, mlp_type="normal",
                        out_norm=False, last_vid_only=False)),
    )
    for model, over, thw, rope, ocfg in cases:
        net, cfg = build_reference_dit(model, over, seed=31)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items() if "rope" not in k}
        want = odit.param_shapes(ocfg)
        assert set(sd) == set(want), sorted(set(sd) ^ set(want))[:10]
        assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
        print(f"  [ok] G17[{model}] state-dict: {len(sd)} keys and shapes identical to the oracle's param_shapes")
        T, H, W = thw
        g = torch.Generator().manual_seed(32)
        vid = torch.randn(T * H * W, 33, generator=g).to(torch.bfloat16).float()
        txt = torch.randn(58, cfg["txt_in_dim"], generator=g).to(torch.bfloat16).float()
        with torch.no_grad():
            ref = net(vid=vid, txt=txt, vid_shape=torch.tensor([[T, H, W]]), txt_shape=torch.tensor([[58]]),
                      timestep=torch.tensor([1000.0])).vid_sample
            mine = odit.forward(sd, ocfg, vid, txt, (T, H, W), 1000.0, torch.float32, rope)
        check(f"G17[{model}] NaDiT.forward", mine, ref, 5e-5)
        # the restated 'none' rope must differ (the pin is sensitive to the rotary embedding)
        with torch.no_grad():
            off = odit.forward(sd, ocfg, vid, txt, (T, H, W), 1000.0, torch.float32, "none")
        assert (off - ref).abs().max() > 1e-3
        if model == "3b":
            # bf16: autocast-free module in bf16 == the oracle's op-by-op bf16 path up to summation order
            netb = net.bfloat16()
            with torch.no_grad():
                refb = netb(vid=vid.bfloat16(), txt=txt.bfloat16(), vid_shape=torch.tensor([[T, H, W]]),
                            txt_shape=torch.tensor([[58]]), timestep=torch.tensor([1000.0])).vid_sample.float()
                mineb = odit.forward(sd, ocfg, vid.bfloat16(), txt.bfloat16(), (T, H, W), 1000.0, torch.bfloat16, rope).float()
            rel = ((mineb - refb).pow(2).mean().sqrt() / refb.pow(2).mean().sqrt()).item()
            rel32 = ((refb - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            print(f"  [..] G17[3b] bf16 module vs bf16 oracle rel-rms {rel:.2e} (bf16 vs fp32 reference: {rel32:.2e})")
            assert rel <= rel32, "bf16 oracle is further from the bf16 reference than bf16 is from fp32"
            net.float()
        for k, v in sd.items():
            tensors[f"G17.{model}.p.{k}"] = v.to(torch.bfloat16).contiguous()
        tensors.update({f"G17.{model}.vid": vid, f"G17.{model}.txt": txt, f"G17.{model}.out": ref.contiguous()})
        meta[f"G17.{model}"] = {"latent_thw": list(thw), "rope": rope,
                                "cfg": {k: over[k] for k in over}, "num_layers": cfg["num_layers"]}
    meta["G17.keys_3b"] = {k: list(v) for k, v in odit.param_shapes(odit.CFG_3B).items()
                           if k.startswith(("blocks.0.", "blocks.10.", "blocks.31.")) or not k.startswith("blocks.")}


def pin_dit_keys_fullsize(meta):
    """State-dict keys / shapes of the FULL-size NaDiT-3B and 7B on the meta device (no memory): the checkpoint
    layout the product's loader consumes."""
    for model, ocfg in (("3b", odit.CFG_3B), ("7b", odit.CFG_7B)):
        with torch.device("meta"):
            net, _ = build_reference_dit_meta(model)
        sd = {k: tuple(v.shape) for k, v in net.state_dict().items() if "rope" not in k}
        want = {k: tuple(v) for k, v in odit.param_shapes(ocfg).items()}
        assert sd == want, [k for k in set(sd) ^ set(want)][:10]
        n = sum(int(torch.tensor(v).prod()) for v in sd.values())
        print(f"  [ok] G17 full-size NaDiT-{model}: {len(sd)} keys, {n / 1e9:.2f} G parameters, shapes identical to the oracle's")
        meta[f"G17.fullsize_{model}"] = {"keys": len(sd), "params": n}


def build_reference_dit_meta(model):
    with open(os.path.join(REF, f"configs_{model}/main.yaml")) as fh:
        cfg = yaml.safe_load(fh)["dit"]["model"]
    cfg.pop("__object__")
    cfg["norm_eps"] = float(cfg["norm_eps"])
    n = cfg["num_layers"]
    cfg["txt_dim"], cfg["emb_dim"] = cfg["vid_dim"], 6 * cfg["vid_dim"]
    cfg["block_type"] = _expand(n, "mmdit_sr")
    cfg["window"] = _expand(n, (4, 3, 3))
    cfg["window_method"] = (n // 2) * ["720pwin_by_size_bysize", "720pswin_by_size_bysize"]
    if cfg.get("msa_type") == "None":
        cfg["msa_type"] = None
    import importlib
    rmod = importlib.import_module(f"src.models.dit_{model}.nadit")
    return rmod.NaDiT(**cfg), cfg


# ================================================================================================ G10b: host helpers
def _extract(path, names, ns):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (path, names)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return ns


def pin_host_helpers(tensors, meta):
    print("G10b pad_video_temporal / blend_overlapping_frames / _apply_4n1_padding (taken from the syntax tree, run as they are)")
    from typing import Optional
    from src.optimization.performance import optimized_single_video_rearrange   # the reference's own helper
    ns = {"torch": torch, "Optional": Optional, "optimized_single_video_rearrange": optimized_single_video_rearrange}
    _extract(os.path.join(REF, "src/core/generation_utils.py"), ("pad_video_temporal", "blend_overlapping_frames"), ns)
    _extract(os.path.join(REF, "src/core/generation_phases.py"), ("_apply_4n1_padding",), ns)
    pad, blend, pad4 = ns["pad_video_temporal"], ns["blend_overlapping_frames"], ns["_apply_4n1_padding"]
    table = {}
    for t in range(1, 24):
        v = torch.arange(t).float().view(t, 1, 1, 1).expand(t, 2, 1, 1).contiguous()
        for count in (0, 1, 2, 3, 5, 8, 21, 30):
            ref = pad(v, count=count, temporal_dim=0)[:, 0, 0, 0].long().tolist()
            mine = opipe.pad_video_temporal(v, count)[:, 0, 0, 0].long().tolist()
            assert ref == mine, (t, count, ref, mine)
            table[f"{t}|{count}"] = ref
        ref4 = pad4(v)[:, 0, 0, 0].long().tolist()
        assert ref4 == table[f"{t}|0"]
        # CTHW flavour (temporal_dim=1), which encode_all_batches never uses for padding but prepend does
        vc = v.permute(1, 0, 2, 3)
        for count in (1, 3, 30):
            pre = pad(vc, count=count, temporal_dim=1, prepend=True)[0, :, 0, 0].long().tolist()
            table[f"prepend|{t}|{count}"] = pre
    meta["G10b.pad_video_temporal"] = table
    print(f"  [ok] G10b pad_video_temporal: {len(table)} (length, count) cases identical")
    g = torch.Generator().manual_seed(41)
    for ov in (1, 2, 3, 4, 5, 8):
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            a = (torch.rand(ov, 6, 5, 3, generator=g) * 2 - 1).to(dt)
            b = (torch.rand(ov, 6, 5, 3, generator=g) * 2 - 1).to(dt)
            ref = blend(a, b, ov)
            check(f"G10b blend_overlapping_frames overlap {ov} {tag}", opipe.blend_overlapping_frames(a, b, ov).float(), ref.float(), 0.0)
            tensors[f"G10b.blend.{ov}.{tag}.a"], tensors[f"G10b.blend.{ov}.{tag}.b"] = a, b
            tensors[f"G10b.blend.{ov}.{tag}.out"] = ref


# ================================================================================================ G16h / G17h
def _fingerprint(params):
    """Order-independent checksum of a seeded parameter set (the fixtures store it instead of the weights)."""
    return float(sum(v.double().abs().sum().item() * (1 + (i % 7)) for i, (k, v) in enumerate(sorted(params.items()))))


def pin_hip_sized(tensors, dit_tensors, meta):
    """Reference models at sizes the HIP kernels accept (head_dim 128; VAE widths 64/128/512/512), weights =
    the oracle's seeded synthetic parameters rounded to bf16 (regenerated by the tests from the seed: only inputs
    and the REFERENCE's outputs are stored).  The GPU tests compare the HIP path with these outputs directly."""
    print("G16h / G17h reference outputs at HIP-compatible sizes (seeded weights, not stored)")
    # ---- VAE
    ch = (64, 128, 512, 512)
    vae, cfg, main = build_reference_vae(ch, 32, seed=51)
    ocfg = ovae.VAEConfig(block_out_channels=ch)
    p = {k: v.to(torch.bfloat16).float() for k, v in ovae.synthetic_params(ocfg, 52).items()}
    missing, unexpected = vae.load_state_dict(p, strict=True), None
    g = torch.Generator().manual_seed(53)
    x = (torch.rand(1, 3, 9, 32, 48, generator=g) * 2 - 1).to(torch.bfloat16).float()
    with torch.no_grad():
        enc = vae.encode(x)
        z = enc.latent
        zq = (z * ocfg.scaling_factor).to(torch.bfloat16).float()          # the latent the runner hands on (infer.py:176)
        dec = vae.decode(zq / ocfg.scaling_factor).sample
        check("G16h oracle encoder == reference", ovae.encoder(p, x[0], ocfg)[:16], z[0], 5e-5)
        check("G16h oracle decoder == reference", ovae.decoder(p, (zq / ocfg.scaling_factor)[0], ocfg), dec[0], 1e-4)
    tensors.update({"G16h.x": x[0].clone(), "G16h.z": z[0].clone(), "G16h.latent_in": zq[0].clone(), "G16h.dec": dec[0].clone()})
    meta["G16h"] = {"block_out_channels": list(ch), "param_seed": 52, "fingerprint": _fingerprint(p)}
    # ---- NaDiT (3B flavour and 7B flavour), head_dim 128
    for model, over, thw, rope, okw in (
        ("3b", dict(vid_dim=256, heads=2, head_dim=128, rope_dim=128, num_layers=4, mm_layers=2, txt_in_dim=64), (2, 16, 24), "3b",
         dict(vid_dim=256, heads=2, head_dim=128, num_layers=4, mm_layers=2, txt_in_dim=64)),
        ("7b", dict(vid_dim=256, heads=2, head_dim=128, num_layers=3, txt_in_dim=64), (3, 20, 36), "7b",
         dict(vid_dim=256, heads=2, head_dim=128, num_layers=3, mm_layers=3, txt_in_dim=64, mlp_type="normal", out_norm=False,
              last_vid_only=False)),
    ):
        net, cfg = build_reference_dit(model, over, seed=54)
        ocfg = odit.DiTConfig(**okw)
        p = {k: v.to(torch.bfloat16).float() for k, v in odit.synthetic_params(ocfg, 55).items()}
        res = net.load_state_dict(p, strict=False)
        assert not res.unexpected_keys and all("rope" in k for k in res.missing_keys), res
        T, H, W = thw
        g = torch.Generator().manual_seed(56)
        noise = torch.randn(T, H, W, 16, generator=g).to(torch.bfloat16).float()
        latent = torch.randn(T, H, W, 16, generator=g).to(torch.bfloat16).float()
        txt = torch.randn(58, 64, generator=g).to(torch.bfloat16).float()
        vid = torch.cat((noise, latent, torch.ones(T, H, W, 1)), dim=-1).reshape(-1, 33)     # infer.py:54-78 condition
        with torch.no_grad():
            ref = net(vid=vid, txt=txt, vid_shape=torch.tensor([[T, H, W]]), txt_shape=torch.tensor([[58]]),
                      timestep=torch.tensor([1000.0])).vid_sample
            check(f"G17h[{model}] oracle forward == reference", odit.forward(p, ocfg, vid, txt, (T, H, W), 1000.0, torch.float32, rope),
                  ref, 1e-4)
        dit_tensors.update({f"G17h.{model}.noise": noise, f"G17h.{model}.latent": latent, f"G17h.{model}.txt": txt,
                        f"G17h.{model}.pred": ref.clone()})
        meta[f"G17h.{model}"] = {"cfg": okw, "param_seed": 55, "fingerprint": _fingerprint(p), "latent_thw": list(thw), "rope": rope}


def pin_side_resize(tensors, meta):
    """G10c: SideResize.__call__ (side_resize.py:36-75) and DivisiblePad (divisible_crop.py:43-80) taken from the syntax
    tree and run as they are; `TVF.resize` is served by ATen's antialiased bicubic with torchvision's documented tensor
    semantics (int size = short edge, long = int(size*long/short); non-float32 tensors are computed in fp32 and cast
    back) -- torchvision is absent, so that one function is the unpinned part; the two-pass max_size logic is the reference's."""
    print("G10c SideResize / DivisiblePad (classes taken from the syntax tree, TVF.resize served by ATen)")
    import types
    import torch.nn.functional as F
    from typing import Union

    def tvf_resize(img, size, interpolation=None, antialias=True):
        h, w = img.shape[-2:]
        if isinstance(size, int):
            short, long_ = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long_ / short)
            size = (new_long, new_short) if w <= h else (new_short, new_long)
        out = F.interpolate(img.float(), size=tuple(size), mode="bicubic", align_corners=False, antialias=antialias)
        return out.to(img.dtype)

    class _Img:   # PIL.Image.Image stand-in (never instantiated)
        pass

    ns = {"torch": torch, "Union": Union, "Image": types.SimpleNamespace(Image=_Img),
          "InterpolationMode": types.SimpleNamespace(BICUBIC="bicubic", BILINEAR="bilinear"),
          "TVF": types.SimpleNamespace(resize=tvf_resize)}
    for path, cls in (("src/data/image/transforms/side_resize.py", "SideResize"),
                      ("src/data/image/transforms/divisible_crop.py", "DivisiblePad")):
        with open(os.path.join(REF, path)) as fh:
            tree = ast.parse(fh.read())
        node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]
        assert len(node) == 1
        exec(compile(ast.Module(body=node, type_ignores=[]), path, "exec"), ns)
    g = torch.Generator().manual_seed(61)
    cases = [((3, 27, 40), 54, 0), ((2, 27, 40), 54, 64), ((2, 40, 27), 54, 70), ((1, 30, 90), 64, 100), ((2, 36, 36), 50, 48)]
    table = []
    for i, ((t, h, w), res, cap) in enumerate(cases):
        x = torch.rand(t, 3, h, w, generator=g).to(torch.bfloat16)
        y = ns["SideResize"](size=res, max_size=cap)(x)
        y = torch.clamp(y, 0.0, 1.0)
        y = ns["DivisiblePad"]((16, 16))(y)
        y = ((y - 0.5) / 0.5).permute(1, 0, 2, 3)          # Normalize(0.5, 0.5); t c h w -> c t h w (generation_utils.py:70-84)
        mine = opipe.transform(x, res, cap)
        check(f"G10c transform {h}x{w} -> {res} (max {cap})", mine.float(), y.float(), 0.0)
        rh, rw = opipe.target_dims(h, w, res, cap)[:2]
        table.append({"in": [t, h, w], "resolution": res, "max_resolution": cap, "resized": [rh, rw], "padded": list(y.shape[2:])})
        tensors[f"G10c.{i}.x"], tensors[f"G10c.{i}.y"] = x, y.contiguous()
    meta["G10c"] = table


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref_standins.install()
    meta = {}
    vt, dt_, ht = {}, {}, {}
    pin_host_helpers(ht, meta)
    pin_side_resize(ht, meta)
    pin_vae(vt, meta)
    pin_dit(dt_, meta)
    pin_dit_keys_fullsize(meta)
    pin_hip_sized(ht, dt_, meta)
    vt.update(ht)
    save_file({k: v.contiguous().clone() for k, v in vt.items()}, os.path.join(GOLD, "vae_model_pins.safetensors"))
    save_file({k: v.contiguous().clone() for k, v in dt_.items()}, os.path.join(GOLD, "dit_model_pins.safetensors"))
    with open(os.path.join(GOLD, "model_pins.json"), "w") as fh:
        json.dump(meta, fh)
    print("model-level golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
