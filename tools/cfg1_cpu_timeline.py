#!/usr/bin/env python3
"""Where the host is while the GPU runs one cfg1 image (512 x 512): host time stamps after each phase's launches have been
issued (no synchronisation), against the GPU time of the whole step.  Tells a launch-bound phase from a GPU-bound one."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
hip = pkg.hip
dev = torch.device("cuda:0")
hip.device_check(0)
vcfg = pkg.vae.VAEConfig()
vae = pkg.vae.VideoVAEHip(vcfg, pkg.vae.synthetic_state_dict(vcfg, dev, seed=1), dev)
dit = pkg.dit.NaDiTHip(pkg.dit.CFG_3B, pkg.dit.synthetic_state_dict(pkg.dit.CFG_3B, dev, seed=2), dev)
text = (0.5 * torch.randn(58, 5120, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, text, dev)
lr = torch.rand(1, 256, 256, 3, device=dev).to(torch.bfloat16)

marks = []
def mark(name):
    marks.append((name, time.perf_counter()))

orig_enc, orig_dec, orig_step = vae.encode, vae.decode, dit.one_step
def enc(*a, **k):
    mark("transform issued"); r = orig_enc(*a, **k); mark("encode issued"); return r
def dec(*a, **k):
    r = orig_dec(*a, **k); mark("decode issued"); return r
def one(*a, **k):
    mark("noise drawn + uploaded"); r = orig_step(*a, **k); mark("DiT issued"); return r
vae.encode, vae.decode, dit.one_step = enc, dec, one

with torch.cuda.device(dev):
    for _ in range(5):
        pipe.upscale_batch(lr, 512, 42)
    torch.cuda.synchronize()
    for rep in range(3):
        marks.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mark("start")
        pipe.upscale_batch(lr, 512, 42)
        mark("returned")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"-- image {rep}: GPU done after {1e3 * (t1 - t0):.2f} ms; host:")
        for n, t in marks:
            print(f"   {1e3 * (t - t0):8.2f} ms  {n}")

# ---- host duration of every hip.* call inside the mid-block attention (free-running loop, no synchronisation)
import collections
durs = collections.defaultdict(list)
def timed(name, fn):
    def inner(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); durs[name].append(1e6 * (time.perf_counter() - t)); return r
    return inner
vae.encode, vae.decode, dit.one_step = orig_enc, orig_dec, orig_step
orig_mid = vae._mid_attention
state = {"on": False}
for nm in ("groupnorm_silu", "gemm", "attn_d512"):
    f = getattr(hip, nm)
    def mk(nm, f):
        def inner(*a, **k):
            if not state["on"]:
                return f(*a, **k)
            t = time.perf_counter(); r = f(*a, **k); durs[nm].append(1e6 * (time.perf_counter() - t)); return r
        return inner
    setattr(hip, nm, mk(nm, f))
pkg.vae.hip = hip
def mid(*a, **k):
    state["on"] = True
    t = time.perf_counter(); r = orig_mid(*a, **k); durs["_mid_attention total"].append(1e6 * (time.perf_counter() - t))
    state["on"] = False
    return r
vae._mid_attention = mid
with torch.cuda.device(dev):
    for _ in range(20):
        pipe.upscale_batch(lr, 512, 42)
    torch.cuda.synchronize()
for k, v in durs.items():
    v = sorted(v)
    print(f"host us per call  {k:24s} n={len(v):4d} median {v[len(v)//2]:7.1f}  max {v[-1]:7.1f}")
