#!/usr/bin/env python3
"""Writes the at-size golden crops of tests/golden/fullsize/ (test infrastructure: it runs the CPU oracle, for minutes per
case, on the GPU box's host cores - too slow for the collected suite; the collected tests then compare the HIP path with
the stored crops in seconds: tests/test_gpu_fullsize_golden.py).

    python tests/fullsize_golden.py cfg2|4k3b|4k7b       the assembled pipeline at true size vs the bf16 oracle
                                                        (stage by stage with per-stage CPU seconds: the measured CPU baseline)
    python tests/fullsize_golden.py dit4k               3-layer full-width NaDiT on grid (6,135,240) vs bf16 + fp32 oracle
    python tests/fullsize_golden.py vae4k_enc           5-frame 4K clip through the VAE encoder (chunks 1 + 4, carries) vs bf16 oracle
    python tests/fullsize_golden.py vae4k_dec           2 latent frames -> 5 frames of 4K through the decoder (carries)

Outputs go to gpurun_out/golden/<case>.safetensors + .json; they are copied into tests/golden/fullsize/ and committed.
"""
import json
import os
import sys
import threading
import time

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import __graft_entry__ as ge
import fullsize_cases as fc

pkg = ge.load_package()
from seedvr2_ref import dit as odit, pipeline as opipe, vae as ovae

CASE = sys.argv[1]
MODE = sys.argv[2] if len(sys.argv) > 2 else ""
OUT_DIR = os.path.join(ROOT, "gpurun_out", "golden")
os.makedirs(OUT_DIR, exist_ok=True)
t_start = time.time()
stop = False


def heartbeat():
    while not stop:
        with open(os.path.join(OUT_DIR, f"{CASE}_heartbeat.txt"), "w") as fh:
            fh.write(f"{time.time() - t_start:.0f}s\n")
        print(f"[heartbeat] {time.time() - t_start:.0f}s", flush=True)
        time.sleep(45)


threading.Thread(target=heartbeat, daemon=True).start()
NTHREADS = min(os.cpu_count() or 1, 64)
torch.set_num_threads(NTHREADS)
dev = torch.device("cuda:0")
tensors, meta = {}, {"case": CASE, "cpu_threads": NTHREADS, "written": time.strftime("%Y-%m-%d %H:%M:%S")}


def put_crops(name, x, size, seed=7):
    """x [T,H,W,C] (CPU): boxes, crops (bf16) and per-frame channel means of the WHOLE tensor (fp32)."""
    boxes = fc.crop_boxes(x.shape[0], x.shape[1], x.shape[2], size, seed)
    tensors[f"{name}.boxes"] = torch.tensor(boxes, dtype=torch.int32)
    tensors[f"{name}.crops"] = fc.take_crops(x, boxes, size).to(torch.bfloat16).contiguous()
    tensors[f"{name}.means"] = x.float().mean(dim=(1, 2)).contiguous()
    meta.setdefault("crop_size", {})[name] = size
    meta.setdefault("shape", {})[name] = list(x.shape)
    return boxes


def compare(name, got, ref_name, size, peak=1.0):
    """Record what the HIP path measured against the stored crops at writing time (the test's thresholds derive from it)."""
    boxes = [tuple(b) for b in tensors[f"{ref_name}.boxes"].tolist()]
    g = fc.take_crops(got, boxes, size).float()
    r = tensors[f"{ref_name}.crops"].float()
    row = {"psnr_crops": fc.psnr(g, r, peak), "peak": peak, "maxabs_crops": float((g - r).abs().max()), "rel_rms_crops": fc.rel_rms(g, r),
           "means_maxabs": float((got.float().mean(dim=(1, 2)) - tensors[f"{ref_name}.means"]).abs().max())}
    meta.setdefault("recorded", {})[name] = row
    print(name, row, flush=True)


def finish():
    global stop
    stop = True
    meta["total_seconds"] = time.time() - t_start
    save_file(tensors, os.path.join(OUT_DIR, f"{CASE}.safetensors"))
    with open(os.path.join(OUT_DIR, f"{CASE}.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("done", json.dumps(meta), flush=True)


def timed(mod, attr, bucket):
    fn = getattr(mod, attr)

    def inner(*a, **k):
        t0 = time.time()
        r = fn(*a, **k)
        meta.setdefault("oracle_stage_seconds", {})[bucket] = meta.get("oracle_stage_seconds", {}).get(bucket, 0.0) + time.time() - t0
        print(f"[oracle] {bucket}: {time.time() - t0:.1f}s", flush=True)
        return r
    setattr(mod, attr, inner)


def build_models(model):
    vcfg_h = pkg.vae.VAEConfig()
    dcfg_h = pkg.dit.CFG_7B if model == "7b" else pkg.dit.CFG_3B
    vsd = pkg.vae.synthetic_state_dict(vcfg_h, dev, seed=1)
    dsd = pkg.dit.synthetic_state_dict(dcfg_h, dev, seed=2)
    meta["weights"] = {"vae": fc.fingerprint(vsd), "dit": fc.fingerprint(dsd)}
    vae = pkg.vae.VideoVAEHip(vcfg_h, vsd, dev)
    dit = pkg.dit.NaDiTHip(dcfg_h, dsd, dev)
    vp = {k: v.detach().cpu() for k, v in vsd.items()}
    dp = {k: v.detach().cpu() for k, v in dsd.items()}
    return vae, dit, vp, dp


if CASE in fc.PIPELINE_CASES:
    nfr, hin, win, res, bsz, model = fc.PIPELINE_CASES[CASE]
    meta["input"] = f"[{nfr},{hin},{win},3] -> {res}p, batch_size {bsz}, seed 42, color_correction none, NaDiT-{model} + VAE 128/256/512/512"
    vae, dit, vp, dp = build_models(model)
    txt = fc.text_embedding()
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, txt, dev)
    lr = fc.lr_clip(nfr, hin, win)
    t0 = time.time()
    got = pipe.generate(lr, resolution=res, batch_size=bsz, temporal_overlap=0, seed=42, color_correction="none").float().cpu()
    torch.cuda.synchronize()
    meta["hip_seconds_first_call"] = time.time() - t0
    # the same stages one by one for the taps (latent after encode, x0 after the DiT step)
    batches, _ = pkg.pipeline.plan_batches(nfr, bsz, 0)          # one batch; its index list carries the 4n+1 padding
    vid = lr.to(dev).to(torch.bfloat16).index_select(0, torch.tensor(batches[0].index, device=dev))
    x = pipe.transform(vid, res)
    lat = vae.encode(x)
    noise = pipe.make_noise(lat.shape, 42)
    x0 = dit.one_step(noise.to(dev), lat, pipe.text_pos)
    torch.cuda.synchronize()
    hip_taps = {"latent": lat.float().cpu(), "x0": x0.float().cpu()}
    del x, lat, x0, vid
    # the bf16 oracle = the reference's CPU path, stage by stage with its seconds (the measured CPU baseline of this size)
    timed(opipe, "transform", "transform")
    timed(opipe._vae, "encode", "vae_encode")
    timed(opipe._dit, "one_step_sr", "dit")
    timed(opipe._vae, "decode", "vae_decode")
    taps = {}
    dcfg_o = odit.CFG_7B if model == "7b" else odit.CFG_3B
    t0 = time.time()
    ref = opipe.generate(lr, vp, ovae.VAEConfig(), dp, dcfg_o, txt, res, bsz, 0, 42, torch.bfloat16, taps=taps, rope_mode=model).float()
    meta["oracle_bf16_seconds"] = time.time() - t0
    meta["frames"] = nfr
    put_crops("rgb", ref, 64)
    put_crops("latent", taps["latent"].float(), 32)
    put_crops("x0", taps["x0"].float(), 32)
    compare("rgb", got, "rgb", 64)
    compare("latent", hip_taps["latent"], "latent", 32)
    compare("x0", hip_taps["x0"], "x0", 32)
    meta["psnr_full_hip_vs_bf16oracle"] = fc.psnr(got, ref)
    meta["maxabs_full_hip_vs_bf16oracle"] = float((got - ref).abs().max())
    finish()

elif CASE == "dit4k":
    kw = dict(num_layers=3, mm_layers=1)     # separate weights, shared weights, vid-only last layer; methods regular / shifted / regular
    hcfg, ocfg = pkg.dit.DiTConfig(**kw), odit.DiTConfig(**kw)
    sd = pkg.dit.synthetic_state_dict(hcfg, dev, seed=4)
    meta["weights"] = {"dit": fc.fingerprint(sd)}
    meta["input"] = f"NaDiT full 3B width (2560, 20 heads, hidden 6912), 3 layers, grid {fc.DIT4K_GRID}: 194 400 tokens"
    model = pkg.dit.NaDiTHip(hcfg, sd, dev)
    p = {k: v.detach().cpu() for k, v in sd.items()}
    del sd
    noise, latent, txt = fc.dit4k_inputs()
    t0 = time.time()
    got = model.one_step(noise.to(dev), latent.to(dev), txt.to(dev)).float().cpu()
    torch.cuda.synchronize()
    meta["hip_seconds_first_call"] = time.time() - t0
    t0 = time.time()
    ref16 = odit.one_step_sr(p, ocfg, latent, noise, txt, torch.bfloat16).float()
    meta["oracle_bf16_seconds"] = time.time() - t0
    put_crops("x0_bf16", ref16, 32)
    compare("x0_vs_bf16", got, "x0_bf16", 32)
    meta["rel_rms_full_hip_vs_bf16oracle"] = fc.rel_rms(got, ref16)
    with open(os.path.join(OUT_DIR, f"{CASE}.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    t0 = time.time()
    ref32 = odit.one_step_sr(p, ocfg, latent.float(), noise.float(), txt.float(), torch.float32)
    meta["oracle_fp32_seconds"] = time.time() - t0
    put_crops("x0_fp32", ref32, 32)
    compare("x0_vs_fp32", got, "x0_fp32", 32)
    meta["rel_rms_full_hip_vs_fp32oracle"] = fc.rel_rms(got, ref32)
    meta["rel_rms_full_bf16oracle_vs_fp32oracle"] = fc.rel_rms(ref16, ref32)
    finish()

elif CASE == "vae4k_enc":
    nfr, hin, win, res = fc.VAE4K
    vcfg_h = pkg.vae.VAEConfig()
    vsd = pkg.vae.synthetic_state_dict(vcfg_h, dev, seed=1)
    meta["weights"] = {"vae": fc.fingerprint(vsd)}
    meta["input"] = f"[{nfr},{hin},{win},3] -> transform -> [5,2160,3840,3] -> VAE encode (chunks 1 + 4, causal carries) -> latent [2,270,480,16]"
    vae = pkg.vae.VideoVAEHip(vcfg_h, vsd, dev)
    vp = {k: v.detach().cpu() for k, v in vsd.items()}
    del vsd
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, None, fc.text_embedding(), dev)
    lr = fc.lr_clip(nfr, hin, win)
    t0 = time.time()
    lat = vae.encode(pipe.transform(lr.to(dev).to(torch.bfloat16), res)).float().cpu()
    torch.cuda.synchronize()
    meta["hip_seconds_first_call"] = time.time() - t0
    t0 = time.time()
    x = opipe.transform(lr.to(torch.bfloat16).permute(0, 3, 1, 2), res)
    import piecewise_oracle
    vpe = {k: v for k, v in vp.items() if k.startswith("encoder.")}
    with torch.no_grad():   # library calls on pieces below 2^31 elements (see tests/piecewise_oracle.py)
        ref = piecewise_oracle.encode_piecewise(vpe, x, ovae, torch.bfloat16, max_elems=1 << 31).to(torch.bfloat16).float()
    meta["oracle_flavour"] = "bf16 CPU oracle, library calls on pieces (tests/piecewise_oracle.py)"
    meta["oracle_bf16_seconds"] = time.time() - t0
    put_crops("latent", ref, 32)
    compare("latent", lat, "latent", 32)
    meta["rel_rms_full_hip_vs_bf16oracle"] = fc.rel_rms(lat, ref)
    finish()

elif CASE == "vae4k_dec":
    vcfg_h = pkg.vae.VAEConfig()
    vsd = pkg.vae.synthetic_state_dict(vcfg_h, dev, seed=1)
    meta["weights"] = {"vae": fc.fingerprint(vsd)}
    vae = pkg.vae.VideoVAEHip(vcfg_h, vsd, dev)
    vp = {k: v.detach().cpu() for k, v in vsd.items()}
    del vsd
    z = fc.vae4k_decode_latent()
    meta["input"] = (f"seeded latent {list(z.shape)} -> VAE decode (HIP: one latent frame per chunk, causal carries) -> "
                     f"[5,2160,{fc.VAE4K_DEC_W},3]")
    t0 = time.time()
    got = vae.decode(z.to(dev)).float().cpu()
    torch.cuda.synchronize()
    meta["hip_seconds_first_call"] = time.time() - t0
    t0 = time.time()
    import piecewise_oracle
    vpd = {k: v for k, v in vp.items() if k.startswith("decoder.")}
    with torch.no_grad():   # the plain 5-frame call crashes the CPU conv library at this size: pieces below 2^31 elements
        ref = piecewise_oracle.decode_chunked(vpd, z, ovae, torch.bfloat16, max_elems=1 << 31).to(torch.bfloat16).float()
    meta["oracle_bf16_seconds"] = time.time() - t0
    meta["oracle_flavour"] = "bf16 CPU oracle, library calls on pieces (tests/piecewise_oracle.py), one latent frame per temporal chunk"
    put_crops("dec", ref, 64)
    compare("dec", got, "dec", 64, peak=2.0)
    meta["psnr_full_hip_vs_oracle_pm1"] = fc.psnr(got, ref, peak=2.0)
    finish()
else:
    raise SystemExit(f"unknown case {CASE}")
