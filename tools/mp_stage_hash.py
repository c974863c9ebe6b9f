#!/usr/bin/env python3
"""Diagnostic (GPU box): per-stage hashes of one batch (transform -> encode -> DiT -> decode), solo and with a second
process sharing the GPU.  `python tools/mp_stage_hash.py` = launcher; `... worker <tag>` = one process."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def h(t):
    return int(t.contiguous().view(torch.int16).to(torch.int64).sum().item()) & 0xffffffff


CUR = []


def worker(tag):
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    small = os.environ.get("SVR2_MP_SMALL", "0") == "1"
    dcfg = pkg.dit.DiTConfig(num_layers=4, mm_layers=2) if small else pkg.dit.CFG_3B
    vae = pkg.vae.VideoVAEHip(pkg.vae.VAEConfig(), pkg.vae.synthetic_state_dict(pkg.vae.VAEConfig(), dev, seed=32), dev)
    dit = pkg.dit.NaDiTHip(dcfg, pkg.dit.synthetic_state_dict(dcfg, dev, seed=31), dev)
    txt = (torch.randn(58, 5120, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16)
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, txt, dev)
    g = torch.Generator().manual_seed(9)
    lr = ((torch.rand(5, 24, 32, 3, generator=g) * 255).round() / 255).to(torch.bfloat16).to(dev)
    out = []
    for it in range(int(os.environ.get("SVR2_MP_ITERS", "3"))):
        x = pipe.transform(lr, 48)
        lat = vae.encode(x)
        noise = pipe.make_noise(lat.shape, 7).to(dev)
        if os.environ.get("SVR2_MP_TAPS") == "1":      # eager forward, every kernel's output hashed: which op diverges first?
            rec = []
            if it == 0:
                for name in ("rmsnorm_mod", "gemm", "qkv_prep", "window_attn", "txt_mean", "patch_in", "patch_out_x0"):
                    def wrap(fn, name=name):
                        def inner(*a, **k):
                            if os.environ.get("SVR2_MP_SYNC") == name or os.environ.get("SVR2_MP_SYNC") == "all":
                                torch.cuda.synchronize()
                            r = fn(*a, **k)
                            outs = r if isinstance(r, (tuple, list)) else (r,)
                            tag_ = name + ("[" + "x".join(str(d) for d in a[0].shape) + (",epi%d" % k.get("epilogue", 0) if name == "gemm" else "") + "]")
                            CUR.append((tag_, [t for t in outs if torch.is_tensor(t)]))
                            return r
                        return inner
                    setattr(pkg.hip, name, wrap(getattr(pkg.hip, name)))
            CUR.clear()
            x0 = dit._one_step_eager(noise, lat, pipe.text_pos, None)
            torch.cuda.synchronize()
            hs = [f"{i}:{n}:" + ",".join(f"{h(t):08x}" for t in ts) for i, (n, ts) in enumerate(CUR)]
            out.append(f"it{it} " + " ".join(hs))
            continue
        x0 = dit.one_step(noise, lat, pipe.text_pos)
        dec = vae.decode(x0)
        torch.cuda.synchronize()
        out.append(f"it{it} x {h(x):08x} lat {h(lat):08x} noise {h(noise):08x} x0 {h(x0):08x} dec {h(dec):08x}")
    if os.environ.get("SVR2_MP_TAPS") == "1":
        ref = os.environ.get("SVR2_MP_REF")
        if ref and os.path.exists(ref):
            want = open(ref).read().split("\n")
            for i, (a, b) in enumerate(zip(out, want)):
                if a != b:
                    da, db = a.split()[1:], b.split()[1:]
                    first, ref_ = next(((x, y) for x, y in zip(da, db) if x != y), ("?", "?"))
                    print(tag, f"iteration {i} DIVERGES first at {first} (solo {ref_.split(':')[-1]})", flush=True)
            if out == want[:len(out)]:
                print(tag, "all iterations equal to solo", flush=True)
        elif ref:
            open(ref, "w").write("\n".join(out))
            print(tag, "reference written", flush=True)
        return
    print(tag, " | ".join(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        return worker(sys.argv[2])
    me = [sys.executable, os.path.abspath(__file__), "worker"]
    if os.environ.get("SVR2_MP_TAPS") == "1":
        os.environ["SVR2_MP_REF"] = "/tmp/svr2_mp_ref.txt"
        if os.path.exists(os.environ["SVR2_MP_REF"]):
            os.remove(os.environ["SVR2_MP_REF"])
    subprocess.check_call(me + ["solo "])
    for r in range(int(os.environ.get("SVR2_MP_ROUNDS", "4"))):
        ps = [subprocess.Popen(me + [f"r{r}-A "]), subprocess.Popen(me + [f"r{r}-B "])]
        [p.wait() for p in ps]


if __name__ == "__main__":
    main()
