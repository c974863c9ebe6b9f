#!/usr/bin/env python3
"""Diagnostic (GPU box): qkv_prep alone, repeated on fixed inputs, solo and with a second process on the same GPU."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(tag):
    import __graft_entry__ as ge
    pkg = ge.load_package()
    hip, Wn = pkg.hip, pkg.windows
    dev = torch.device("cuda:0")
    H, grid, Lt = 20, (2, 3, 4), 58
    g = torch.Generator().manual_seed(1)
    pl = Wn.plan(grid, Wn.REGULAR, Lt)
    L = grid[0] * grid[1] * grid[2]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    qkv_v = torch.randn(L, 3 * H * 128, generator=g).to(torch.bfloat16).to(dev)
    qkv_t = torch.randn(Lt, 3 * H * 128, generator=g).to(torch.bfloat16).to(dev)
    qn, kn = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (1 + 0.1 * torch.randn(128, generator=g)).to(dev)
    fr = 1.0 / (10000.0 ** (torch.arange(0, 42, 2)[:21].float() / 42))
    ang = torch.arange(1024, dtype=torch.float32)[:, None] * fr[None]
    rc, rs = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    src, pos = up(pl.src_row), up(pl.pos)
    mode = int(os.environ.get("SVR2_MP_ROPE", "1"))
    if mode == 2:
        rc, rs = torch.rand(pl.total, 30, generator=g).to(dev), torch.rand(pl.total, 30, generator=g).to(dev)
    vk = os.environ.get("SVR2_MP_VICTIM", "qkv_prep")
    xr = torch.randn(4096, 2560, generator=g).to(torch.bfloat16).to(dev)
    sc = torch.randn(2560, generator=g).to(dev)
    xs = torch.randn(2048, 1000, generator=g).to(dev)
    if vk == "qkv_prep":
        call = lambda: hip.qkv_prep(qkv_v, qkv_t, src, pos, qn, kn, qn, kn, rc, rs, H, mode, 1e-5)
    elif vk == "rmsnorm":
        call = lambda: hip.rmsnorm_mod(xr, sc, sc, 1e-5).view(1, 4096, 2560, 1)[:, :, :, :].expand(3, -1, -1, -1)
    else:
        call = lambda: torch.softmax(xs, dim=-1).view(1, 2048, 1000, 1).expand(3, -1, -1, -1)
    ref = call().clone()
    torch.cuda.synchronize()
    side = None
    if os.environ.get("SVR2_MP_SAME_PROCESS") == "1":     # the 128-row GEMM on a second stream of THIS process instead
        side = torch.cuda.Stream()
        ga = torch.randn(24, 2560, device=dev).to(torch.bfloat16)
        gw = hip.pad_weight((torch.randn(7680, 2560, device=dev) * 0.02).to(torch.bfloat16))
        gout = torch.empty(24, 7680, dtype=torch.bfloat16, device=dev)
    bad = 0
    n = int(os.environ.get("SVR2_MP_N", "3000"))
    for i in range(n):
        if side is not None:
            with torch.cuda.stream(side):
                for _ in range(4):
                    hip.gemm(ga, gw, 7680, gout)
        out = call()
        if not torch.equal(out, ref):
            bad += 1
            if bad <= 3:
                d = (out.float() - ref.float())
                parts = [int((d[j] != 0).sum()) for j in range(3)]
                idx = (d != 0).nonzero()[:4].tolist()
                print(tag, f"call {i}: differing elements q/k/v = {parts}, max |d| {float(d.abs().max()):.4g}, first {idx}", flush=True)
    print(tag, f"{bad} of {n} calls differ from the first call", flush=True)


def aggressor(kind):
    """Runs one kind of kernel for a few seconds (the 'other process')."""
    import math
    import time
    import __graft_entry__ as ge
    pkg = ge.load_package()
    hip = pkg.hip
    dev = torch.device("cuda:0")
    rb = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
    if kind == "gemm128full":
        a, w = rb(128, 2560), hip.pad_weight(rb(7680, 2560, scale=0.02))
        fn = lambda: hip.gemm(a, w, 7680)
    elif kind == "gemm128":
        a, w = rb(24, 2560), hip.pad_weight(rb(7680, 2560, scale=0.02))
        fn = lambda: hip.gemm(a, w, 7680)
    elif kind == "gemm4":
        a, w = rb(4096, 2560), hip.pad_weight(rb(7680, 2560, scale=0.02))
        fn = lambda: hip.gemm(a, w, 7680)
    elif kind == "attn":
        pl = pkg.windows.plan((2, 16, 24), pkg.windows.REGULAR, 58)
        S = pl.total
        q, k, v = rb(S, 20, 128), rb(S, 20, 128), rb(S, 20, 128)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        cu, ts, tq = up(pl.cu_seqlens), up(pl.tile_seq), up(pl.tile_q0)
        fn = lambda: hip.window_attn(q, k, v, cu, ts, tq, 1 / math.sqrt(128))
    elif kind == "rmsnorm":
        x, sc = rb(4096, 2560), torch.randn(2560, device=dev)
        fn = lambda: hip.rmsnorm_mod(x, sc, sc, 1e-5)
    elif kind == "torch":
        x = torch.randn(4096, 4096, device=dev)
        fn = lambda: x.mul_(1.0001)
    else:
        raise SystemExit(kind)
    t0 = time.time()
    n = 0
    while time.time() - t0 < float(os.environ.get("SVR2_MP_SECONDS", "6")):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    print("aggressor", kind, n, "launches", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        return worker(sys.argv[2])
    if len(sys.argv) > 2 and sys.argv[1] == "aggressor":
        return aggressor(sys.argv[2])
    me = [sys.executable, os.path.abspath(__file__)]
    if os.environ.get("SVR2_MP_SAME_PROCESS") == "1":
        return worker("victim with the GEMM on a second stream of the same process")
    os.environ["SVR2_MP_N"] = os.environ.get("SVR2_MP_N", "20000")
    for kind in os.environ.get("SVR2_MP_KINDS", "torch,rmsnorm,gemm128,gemm4,attn").split(","):
        ps = [subprocess.Popen(me + ["aggressor", kind]), subprocess.Popen(me + ["worker", f"victim-vs-{kind}"])]
        [p.wait() for p in ps]


if __name__ == "__main__":
    main()
