#!/usr/bin/env python3
"""Diagnostic (GPU box): does sharing ONE GPU between two processes change results?  Runs the single-rank CLI alone, then two
copies of it concurrently (no torch.distributed involved), and compares the files byte for byte."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    pkg = ge.load_package()
    from safetensors.torch import save_file
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="svr2_mp_")
    mdir = os.path.join(tmp, "SEEDVR2")
    os.makedirs(mdir)
    dsd = pkg.dit.synthetic_state_dict(pkg.dit.CFG_3B, dev, seed=31)
    vsd = pkg.vae.synthetic_state_dict(pkg.vae.VAEConfig(), dev, seed=32)
    save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in dsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_DIT))
    save_file({k: v.to(torch.float16).cpu().contiguous() for k, v in vsd.items()}, os.path.join(mdir, pkg.weights.DEFAULT_VAE))
    torch.save((torch.randn(58, 5120) * 0.5).to(torch.bfloat16), os.path.join(mdir, "pos_emb.pt"))
    del dsd, vsd
    torch.cuda.empty_cache()
    clip = np.random.default_rng(36).integers(0, 256, size=(10, 24, 32, 3), dtype=np.uint8)
    np.save(os.path.join(tmp, "clip.npy"), clip)
    cli = os.path.join(ROOT, "comfyui-seedvr2_amd", "inference_cli.py")
    env = dict(os.environ, SEEDVR2_MODEL_DIR=mdir)
    env.pop("WORLD_SIZE", None)

    def cmd(tag):
        return [sys.executable, cli, os.path.join(tmp, "clip.npy"), "--resolution", "48", "--batch_size", "5", "--seed", "7",
                "--temporal_overlap", "2", "--color_correction", "none", "--model_dir", mdir, "--output", os.path.join(tmp, f"out_{tag}.npy")]

    subprocess.check_call(cmd("solo"), env=env, stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd("solo2"), env=env, stdout=subprocess.DEVNULL)
    solo = np.load(os.path.join(tmp, "out_solo.npy"))
    print("solo run reproducible:", np.array_equal(solo, np.load(os.path.join(tmp, "out_solo2.npy"))), flush=True)
    extra = os.environ.get("SVR2_MP_EXTRA_ENV", "")
    for k in filter(None, extra.split(",")):
        name, val = k.split("=")
        env[name] = val
    for r in range(int(os.environ.get("SVR2_MP_ROUNDS", "4"))):
        ps = [subprocess.Popen(cmd(f"a{r}"), env=env, stdout=subprocess.DEVNULL), subprocess.Popen(cmd(f"b{r}"), env=env, stdout=subprocess.DEVNULL)]
        rcs = [p.wait() for p in ps]
        outs = [np.load(os.path.join(tmp, f"out_{t}{r}.npy")) for t in "ab"]
        rep = []
        for o in outs:
            bad = [int((o[i] != solo[i]).sum()) for i in range(o.shape[0])]
            rep.append("equal" if not any(bad) else f"DIFFERS per frame {bad}")
        print(f"round {r}: rc {rcs}; process A {rep[0]}; process B {rep[1]}", flush=True)


if __name__ == "__main__":
    main()
