#!/usr/bin/env python3
"""Peak HBM per phase of one cfg3 batch (GPU box)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
dev = torch.device("cuda:0")
T, hin, win, res, ov, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
vcfg = pkg.vae.VAEConfig()
for k, v in (a.split("=") for a in sys.argv[2:]):
    setattr(vcfg, k, int(v))
vae = pkg.vae.VideoVAEHip(vcfg, pkg.vae.synthetic_state_dict(vcfg, dev, seed=1), dev)
dit = pkg.dit.NaDiTHip(pkg.dit.CFG_3B, pkg.dit.synthetic_state_dict(pkg.dit.CFG_3B, dev, seed=2), dev)
pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, torch.randn(58, 5120), dev)
lr = bench.synth_lr_video(T, hin, win, 1, dev)
def phase(name, fn):
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    t0 = time.time(); out = fn(); torch.cuda.synchronize()
    print(f"{name}: {time.time()-t0:.2f}s  base {base/1e9:.1f} GB  peak allocated {torch.cuda.max_memory_allocated()/1e9:.1f} GB  reserved {torch.cuda.max_memory_reserved()/1e9:.1f} GB", flush=True)
    return out
for rep in range(2):
    x = phase("transform", lambda: pipe.transform(lr, res))
    lat = phase("encode", lambda: vae.encode(x)); del x
    noise = pipe.make_noise(lat.shape, 42).to(dev)
    x0 = phase("dit", lambda: dit.one_step(noise, lat, pipe.text_pos))
    dec = phase("decode", lambda: vae.decode(x0)); del dec, x0, lat
