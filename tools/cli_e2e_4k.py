#!/usr/bin/env python3
"""End-to-end run of the CLI the way a user runs it: 64-frame 720p y4m file -> 4K y4m file, NaDiT-3B + VAE loaded from
checkpoint files (synthetic weights in the reference's layout, written to /tmp first), batch 21, overlap 4, default colour
correction (lab).  Prints wall-clock of the whole call (including model load, like the reference's 'Average FPS') and of the
processing alone, next to bench.py's number for the same clip (which has no file I/O and no colour correction)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from safetensors.torch import save_file

pkg = ge.load_package()
dev = torch.device("cuda:0")
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/svr2_cli_e2e"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cc = sys.argv[3] if len(sys.argv) > 3 else "lab"
mdir = os.path.join(work, "SEEDVR2")
os.makedirs(mdir, exist_ok=True)
t0 = time.time()
dsd = pkg.dit.synthetic_state_dict(pkg.dit.CFG_3B, dev, seed=2)
vsd = pkg.vae.synthetic_state_dict(pkg.vae.VAEConfig(), dev, seed=1)
save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in dsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_DIT))
save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in vsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_VAE))
torch.save((0.5 * torch.randn(58, 5120, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16), os.path.join(mdir, "pos_emb.pt"))
del dsd, vsd
torch.cuda.empty_cache()
print(f"checkpoint files written in {time.time() - t0:.1f}s", flush=True)

W, H = 1280, 720
g = torch.Generator().manual_seed(1234)
low = torch.rand(frames, 3, H // 16, W // 16, generator=g)
vid = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)
rgb = (vid * 255).round().to(torch.uint8).numpy()            # [T,3,H,W]
src = os.path.join(work, "clip_720p.y4m")
with open(src, "wb") as fh:
    fh.write(f"YUV4MPEG2 W{W} H{H} F30:1 Ip A1:1 C420jpeg\n".encode())
    for f in rgb:
        r, gg, b = (f[i].astype(np.float32) for i in range(3))
        y = 0.299 * r + 0.587 * gg + 0.114 * b
        u = (-0.168736 * r - 0.331264 * gg + 0.5 * b + 128)[::2, ::2]
        v = (0.5 * r - 0.418688 * gg - 0.081312 * b + 128)[::2, ::2]
        fh.write(b"FRAME\n")
        for p in (y, u, v):
            fh.write(np.clip(np.rint(p), 0, 255).astype(np.uint8).tobytes())
print(f"{frames}-frame 720p clip: {os.path.getsize(src) / 1e6:.0f} MB", flush=True)

cli = importlib.import_module(pkg.__name__ + ".inference_cli")
dst = os.path.join(work, "clip_4k.y4m")
argv = [src, "--resolution", "2160", "--batch_size", "21", "--temporal_overlap", "4", "--seed", "42", "--color_correction", cc,
        "--model_dir", mdir, "--output", dst]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.time()
    rc = cli.main(argv)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"run {rep}: rc={rc}  {dt:.1f}s wall for {frames} frames = {frames / dt:.2f} frames/s including model load; "
          f"output {os.path.getsize(dst) / 1e6:.0f} MB", flush=True)
