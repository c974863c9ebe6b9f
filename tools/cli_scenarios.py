#!/usr/bin/env python3
"""A handful of CLI invocations the way users type them (small inputs, full-size NaDiT-3B + VAE from checkpoint files in
/tmp): image -> png, directory batch mode, video -> PNG sequence, chunked streaming with prepend frames, default output path,
an unsupported option.  Prints rc and what was written; anything but the expected rc is a drop-in bug."""
import glob
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
work = "/tmp/svr2_cli_scen"
mdir = os.path.join(work, "models", "SEEDVR2")
os.makedirs(mdir, exist_ok=True)
dsd = pkg.dit.synthetic_state_dict(pkg.dit.CFG_3B, dev, seed=2)
vsd = pkg.vae.synthetic_state_dict(pkg.vae.VAEConfig(), dev, seed=1)
save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in dsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_DIT))
save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in vsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_VAE))
torch.save((0.5 * torch.randn(58, 5120, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16), os.path.join(mdir, "pos_emb.pt"))
del dsd, vsd
from PIL import Image

g = torch.Generator().manual_seed(5)
def img(h, w):
    return (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
os.makedirs(os.path.join(work, "dir_in"), exist_ok=True)
Image.fromarray(img(48, 64)).save(os.path.join(work, "one.png"))
Image.fromarray(img(40, 56)).save(os.path.join(work, "dir_in", "a.png"))
Image.fromarray(np.concatenate([img(32, 48), img(32, 48)[..., :1]], axis=-1), "RGBA").save(os.path.join(work, "dir_in", "b_rgba.png"))
np.save(os.path.join(work, "clip.npy"), (torch.rand(11, 32, 48, 3, generator=g) * 255).to(torch.uint8).numpy())

cli = importlib.import_module(pkg.__name__ + ".inference_cli")
os.chdir(work)          # the default model directory is ./models/SEEDVR2 relative to the working directory
cases = [
    ("image -> png (default model dir, explicit output)", ["one.png", "--resolution", "96", "--output", "one_up.png"], 0),
    ("image, default output path", ["one.png", "--resolution", "96"], 0),
    ("directory batch mode (RGB + RGBA png)", ["dir_in", "--resolution", "64", "--output", "dir_out"], 0),
    ("video -> png sequence", ["clip.npy", "--resolution", "64", "--batch_size", "5", "--temporal_overlap", "1", "--output_format", "png", "--output", "clip_png"], 0),
    ("video, chunked streaming + prepend frames + wavelet", ["clip.npy", "--resolution", "64", "--batch_size", "5", "--chunk_size", "4", "--temporal_overlap", "1",
      "--prepend_frames", "2", "--color_correction", "wavelet", "--output", "clip_up.y4m"], 0),
    ("video, skip / cap", ["clip.npy", "--resolution", "64", "--skip_first_frames", "3", "--load_cap", "5", "--batch_size", "5", "--output", "clip_cap.npy"], 0),
    ("BlockSwap requested (rejected loudly)", ["one.png", "--blocks_to_swap", "4"], 2),
    ("missing input", ["nothing.png"], 1),
]
for name, argv, want in cases:
    t0 = time.time()
    try:
        rc = cli.main(argv)
    except SystemExit as e:
        rc = e.code if isinstance(e.code, int) else 1
    print(f"[{'ok' if rc == want else 'UNEXPECTED'}] rc={rc} (want {want}) {time.time() - t0:5.1f}s  {name}", flush=True)
for p in sorted(glob.glob(os.path.join(work, "*"))):
    if os.path.isdir(p):
        print("   ", os.path.basename(p) + "/", sorted(os.listdir(p))[:6])
    else:
        print("   ", os.path.basename(p), os.path.getsize(p))
a = np.load(os.path.join(work, "clip_cap.npy"))
print("clip_cap.npy", a.shape, a.dtype)
