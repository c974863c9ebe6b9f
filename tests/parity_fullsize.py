#!/usr/bin/env python3
"""One-off parity runs at TRUE sizes (default BASELINE configs[1]: [8,540,960,3] -> 1080p, batch_size 9 (4n+1 padding), seed 42;
`1 720 1280 2160 1 4k_frame` = one 4K frame through the kernels' 4K shapes),
NaDiT-3B + full VAE on synthetic weights; HIP path vs the bf16 oracle (and the fp32 oracle if the time limit allows) on the
box's host threads.  Test infrastructure (it runs the oracle), too slow for the collected suite (the oracle needs minutes); writes gpurun_out/parity_cfg2_fullsize.json."""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from tests_support import psnr

pkg = ge.load_package()
from seedvr2_ref import dit as odit, pipeline as opipe, vae as ovae

# argv: frames hin win resolution batch_size tag [3b|7b]   (default: configs[1], 3B)
NFR, HIN, WIN, RES, BSZ = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 540, 960, 1080, 9)))
TAG = sys.argv[6] if len(sys.argv) > 6 else "cfg2_fullsize"
MODEL = sys.argv[7] if len(sys.argv) > 7 else "3b"
OUT = os.path.join(ROOT, "gpurun_out", f"parity_{TAG}.json")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
t_start = time.time()
stop = False


def heartbeat():
    while not stop:
        with open(os.path.join(ROOT, "gpurun_out", f"parity_{TAG}_heartbeat.txt"), "w") as fh:
            fh.write(f"{time.time() - t_start:.0f}s\n")
        print(f"[heartbeat] {time.time() - t_start:.0f}s", flush=True)
        time.sleep(45)


threading.Thread(target=heartbeat, daemon=True).start()
torch.set_num_threads(min(os.cpu_count() or 1, 64))
dev = torch.device("cuda:0")
vcfg_h, vcfg_o = pkg.vae.VAEConfig(), ovae.VAEConfig()
vsd = pkg.vae.synthetic_state_dict(vcfg_h, dev, seed=1)
DCFG_H = pkg.dit.CFG_7B if MODEL == "7b" else pkg.dit.CFG_3B
DCFG_O = odit.CFG_7B if MODEL == "7b" else odit.CFG_3B
dsd = pkg.dit.synthetic_state_dict(DCFG_H, dev, seed=2)
vae = pkg.vae.VideoVAEHip(vcfg_h, vsd, dev)
dit = pkg.dit.NaDiTHip(DCFG_H, dsd, dev)
vp = {k: v.detach().cpu() for k, v in vsd.items()}
dp = {k: v.detach().cpu() for k, v in dsd.items()}
del vsd, dsd
txt = (0.5 * torch.randn(58, 5120, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).float()
pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, txt, dev)
g = torch.Generator().manual_seed(1234)
low = 0.5 + 0.2 * torch.randn(NFR, 3, max(HIN // 16, 2), max(WIN // 16, 2), generator=g)
lr = torch.nn.functional.interpolate(low, size=(HIN, WIN), mode="bicubic", align_corners=False).clamp(0, 1)
lr = ((lr * 255).round() / 255).permute(0, 2, 3, 1).contiguous()

t0 = time.time()
got = pipe.generate(lr, resolution=RES, batch_size=BSZ, temporal_overlap=0, seed=42, color_correction="none").float().cpu()
torch.cuda.synchronize()
row = {"input": f"[{NFR},{HIN},{WIN},3] -> {RES}p, batch_size {BSZ}, seed 42, color_correction none, NaDiT-{MODEL} + VAE 128/256/512/512",
       "hip_seconds": time.time() - t0, "shape": list(got.shape), "cpu_threads": torch.get_num_threads()}
print(row, flush=True)
refs = {}
limit = float(os.environ.get("SVR2_PARITY_BUDGET_S", "1000"))
only = os.environ.get("SVR2_PARITY_ONLY")          # "fp32": just that oracle flavour (it needs most of a 20-minute call alone)
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    if only and name != only:
        continue
    if name == "fp32" and not only and (time.time() - t_start) * 2.2 > limit:      # the second oracle run would not fit
        row["fp32"] = "skipped: time limit"
        break
    t0 = time.time()
    refs[name] = opipe.generate(lr, vp, vcfg_o, dp, DCFG_O, txt, RES, BSZ, 0, 42, dt, rope_mode=MODEL).float()
    row[f"oracle_{name}_seconds"] = time.time() - t0
    row[f"psnr_hip_vs_{name}oracle"] = psnr(got, refs[name])
    row[f"maxabs_hip_vs_{name}oracle"] = (got - refs[name]).abs().max().item()
    if len(refs) == 2:
        row["psnr_bf16oracle_vs_fp32oracle"] = psnr(refs["bf16"], refs["fp32"])
    print(row, flush=True)
    with open(OUT, "w") as fh:
        json.dump(row, fh, indent=1)
stop = True
with open(OUT, "w") as fh:
    json.dump(row, fh, indent=1)
print("done", row, flush=True)
