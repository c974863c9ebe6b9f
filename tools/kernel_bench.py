#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the shapes of the named configs (GPU box only).
Prints one line per kernel: time, algorithmic TFLOP/s or GB/s."""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
hip = pkg.hip
dev = torch.device("cuda:0")
hip.device_check(0)


def timeit(fn, iters=10, warm=2):
    iters = int(os.environ.get('SVR2_KB_ITERS', iters)); warm = int(os.environ.get('SVR2_KB_WARM', warm))
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rb(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def bench_gemm(M, N, K, epi="bias", gather=False):
    a = rb(M, K)
    w = hip.pad_weight(rb(N, K, scale=1 / math.sqrt(K)))
    out = torch.empty(M, N if epi != "swiglu" else N // 2, dtype=torch.bfloat16, device=dev)
    if epi == "bias":
        fn = lambda: hip.gemm(a, w, N, out)
    elif epi == "swiglu":
        fn = lambda: hip.gemm(a, w, N, out, epilogue=hip.EPI_SWIGLU)
    else:
        res, gate = rb(M, N), torch.randn(N, device=dev)
        rows = torch.randperm(M, device=dev).to(torch.int32) if gather else None
        fn = lambda: hip.gemm(a, w, N, out, epilogue=hip.EPI_GATE_RES, gate=gate, resid=res, a_rows=rows, m=M)
    t = timeit(fn)
    print(f"gemm[{epi}] M={M} N={N} K={K}: {t*1e3:.3f} ms  {2*M*N*K/t/1e12:.1f} TFLOP/s", flush=True)


def bench_conv(T, H, W, Cin, Cout, k=(3, 3, 3), tiled=False, resid=False):
    x = rb(T, H, W, Cin)
    if tiled:
        w = hip.conv_weight_tiled(rb(Cout, Cin, *k, scale=0.02))
    else:
        w = hip.pad_weight(rb(Cout, k[0] * k[1] * k[2] * Cin, scale=0.02))
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(T, H, W, Cout, dtype=torch.bfloat16, device=dev)
    res = rb(T, H, W, Cout) if resid else None
    stats = tiled and os.environ.get("SVR2_KB_GNSTATS") == "1"      # also reduce the GroupNorm partial sums, like the VAE does
    fn = lambda: hip.conv3d(x, w, Cout, k, npad=k[0] - 1, pad_hw=(k[1] // 2,) * 4, bias=b, out=out, resid=res,
                            w_layout=hip.WL_TILED_128x32 if tiled else hip.WL_ROWS, gn_stats=stats)
    t = timeit(fn)
    fl = 2 * T * H * W * Cin * Cout * k[0] * k[1] * k[2]
    tag = ("halo/tiled" if tiled else "tap-staged") + ("+resid" if resid else "") + ("+stats" if stats else "")
    print(f"conv3d[{tag}] T={T} {H}x{W} {Cin}->{Cout} k={k}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s", flush=True)


def bench_attn(grid, method, H=20):
    pl = pkg.windows.plan(grid, method, 58)
    S = pl.total
    q, k, v = rb(S, H, 128), rb(S, H, 128), rb(S, H, 128)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cu, ts, tq = up(pl.cu_seqlens), up(pl.tile_seq), up(pl.tile_q0)
    out = torch.empty_like(q)
    fn = lambda: hip.window_attn(q, k, v, cu, ts, tq, 1 / math.sqrt(128), out)
    t = timeit(fn)
    lens = np.diff(pl.cu_seqlens).astype(np.float64)
    fl = 4 * 128 * (lens ** 2).sum() * H
    print(f"window_attn grid={grid} {method[:7]} S={S} nwin={pl.nwin}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s", flush=True)


def bench_attn_ab(grid, H=20, versions=(1, 2, 3), rounds=3):
    """Same-process interleaved A/B of the window-attention variants (SVR2_ATTN_V) on both window methods."""
    for method in (pkg.windows.REGULAR, pkg.windows.SHIFTED):
        pl = pkg.windows.plan(grid, method, 58)
        S = pl.total
        q, k, v = rb(S, H, 128), rb(S, H, 128), rb(S, H, 128)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        cu, ts, tq = up(pl.cu_seqlens), up(pl.tile_seq), up(pl.tile_q0)
        out = torch.empty_like(q)
        lens = np.diff(pl.cu_seqlens).astype(np.float64)
        fl = 4 * 128 * (lens ** 2).sum() * H
        ref = None
        for r in range(rounds):
            for ver in versions:
                hip.set_option("SVR2_ATTN_V", ver)
                t = timeit(lambda: hip.window_attn(q, k, v, cu, ts, tq, 1 / math.sqrt(128), out), iters=10, warm=2)
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out, ref))
                print(f"window_attn v{ver} grid={grid} {method[:7]} H={H} S={S} nwin={pl.nwin} tiles={len(pl.tile_seq)}: "
                      f"{t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s  bit-identical-to-v{versions[0]}={same}", flush=True)
        hip.set_option("SVR2_ATTN_V", None)


def bench_attn512(frames, P):
    qkv = rb(frames * P, 1536)
    out = torch.empty(frames * P, 512, dtype=torch.bfloat16, device=dev)
    # same-process A/B, interleaved: SVR2_ATTN5_V 0 = plain loops; 1 = software-pipelined fragment groups (default)
    variants = tuple(int(v) for v in os.environ.get("SVR2_KB_ATTN5_VARS", "0,1").split(","))   # "1": counters on the default only
    for rnd in range(3 if len(variants) > 1 else 1):
        for var in variants:
            hip.set_option("SVR2_ATTN5_V", var)
            t = timeit(lambda: hip.attn_d512(qkv, frames, P, out), iters=3, warm=1)
            print(f"attn_d512[SVR2_ATTN5_V={var}] frames={frames} P={P}: {t*1e3:.3f} ms  {4*512*P*P*frames/t/1e12:.1f} TFLOP/s", flush=True)
    hip.set_option("SVR2_ATTN5_V", None)


def bench_norm(rows, dim):
    x = rb(rows, dim)
    sc, sh = torch.randn(dim, device=dev), torch.randn(dim, device=dev)
    out = torch.empty_like(x)
    t = timeit(lambda: hip.rmsnorm_mod(x, sc, sh, 1e-5, out))
    print(f"rmsnorm_mod {rows}x{dim}: {t*1e3:.3f} ms  {rows*dim*4/t/1e9:.0f} GB/s", flush=True)


def bench_gn(F_, H, W, Cc):
    x = rb(F_, H, W, Cc)
    g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    out = torch.empty_like(x)
    t = timeit(lambda: hip.groupnorm_silu(x, g, b, 32, 1e-6, out=out))
    nbytes = x.numel() * 2
    print(f"groupnorm_silu (stats + apply) {tuple(x.shape)}: {t*1e3:.3f} ms  {3*nbytes/t/1e9:.0f} GB/s (2 reads + 1 write)", flush=True)


def bench_color(T, H, W):
    c = (torch.rand(T, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16)
    s_ = (torch.rand(T, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16)
    nbytes = c.numel() * 2
    for m in ("wavelet", "adain", "lab"):
        t = timeit(lambda: hip.color_correct(c, s_, m), iters=3, warm=1)
        print(f"color_correct[{m}] {T}x{H}x{W}: {t*1e3:.1f} ms  ({nbytes/1e9:.2f} GB per tensor, {T/t:.1f} frames/s)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "norm"]
    if "gemm" in which:
        for M in (24480, 162000):
            bench_gemm(M, 7680, 2560)
            bench_gemm(M, 2560, 2560, "gate")
            bench_gemm(M, 13824, 2560, "swiglu")
            bench_gemm(M, 2560, 6912, "gate")
        bench_gemm(8192, 8192, 8192)
    if "pmc" in which:  # one launch of each hot kernel shape for counter collection
        bench_gemm(162000, 7680, 2560)
        bench_conv(5, 272, 480, 512, 512)
        bench_conv(4, 2160, 3840, 128, 128)
        bench_attn((5, 135, 240), pkg.windows.REGULAR)
    if "stalls" in which:  # one shape of each big MFMA kernel for the stall-counter passes (tools/profile_stalls.sh)
        bench_conv(4, 1080, 1920, 512, 512, tiled=True)     # conv_halo_wr<16,256>
        bench_conv(4, 2160, 3840, 128, 128, tiled=True)     # conv_halo_wr<32,128>
        bench_gemm(194400, 7680, 2560)                      # gemm4 (qkv)
        bench_gemm(194400, 2560, 6912, "gate")              # gemm4 (ffn_out)
        os.environ["SVR2_KB_ATTN5_VARS"] = "1"
        bench_attn512(1, 129600)
    if "gemmsmallm" in which:   # the NaDiT-3B projections of a 512 x 512 image (M = 1024 tokens): tile-starved
        for r in range(3):
            bench_gemm(1024, 7680, 2560)
            bench_gemm(1024, 2560, 2560, "gate")
            bench_gemm(1024, 13824, 2560, "swiglu")
            bench_gemm(1024, 2560, 6912, "gate")
            bench_gemm(58, 2560, 2560, "gate")
    if "gemmsmallm_ab" in which:   # the same projections under every tile geometry the dispatcher can pick, weights rotated
        # through > 256 MB of copies (a layer's weights are never cache-hot in the pipeline), hipBLASLt beside them
        shapes = [("qkv", 1024, 7680, 2560, "bias"), ("out", 1024, 2560, 2560, "gate"),
                  ("ffn_in", 1024, 13824, 2560, "swiglu"), ("ffn_out", 1024, 2560, 6912, "gate")]
        if os.environ.get("SVR2_KB_SHAPES"):     # name:M:N:K:epi,...
            shapes = [(f[0], int(f[1]), int(f[2]), int(f[3]), f[4]) for f in
                      (t.split(":") for t in os.environ["SVR2_KB_SHAPES"].split(","))]
        settings = [("default", {}), ("no residual prefetch", {"SVR2_EPI_PREF": 0}), ("bm256", {"SVR2_GEMM_BM": 256}), ("gemm4", {"SVR2_GEMM_V": 3}),
                    ("bm128_2deep", {"SVR2_GEMM_BM": 128, "SVR2_GEMM_V": 1}),
                    ("gemm2_256x256", {"SVR2_GEMM_V": 3, "SVR2_GEMM4": 0}),
                    ("gemm2_256x128", {"SVR2_GEMM_V": 3, "SVR2_GEMM4": 0, "SVR2_GEMM2_NARROW": 1})]
        for name, M, N, K, epi in shapes:
            ncopy = max(2, int(300e6 // (N * K * 2)) + 1)
            a = rb(M, K)
            ws = [hip.pad_weight(rb(N, K, scale=1 / math.sqrt(K))) for _ in range(ncopy)]
            wt = [w[:N].t() for w in ws]
            out = torch.empty(M, N if epi != "swiglu" else N // 2, dtype=torch.bfloat16, device=dev)
            res, gate = rb(M, N), torch.randn(N, device=dev)
            it = [0]
            def call():
                w = ws[it[0] % ncopy]; it[0] += 1
                if epi == "bias":
                    hip.gemm(a, w, N, out)
                elif epi == "swiglu":
                    hip.gemm(a, w, N, out, epilogue=hip.EPI_SWIGLU)
                else:
                    hip.gemm(a, w, N, out, epilogue=hip.EPI_GATE_RES, gate=gate, resid=res, m=M)
            def lib_call():
                w = wt[it[0] % ncopy]; it[0] += 1
                torch.matmul(a, w)
            for r in range(2):
                for sname, opts in settings:
                    for k, v in opts.items():
                        hip.set_option(k, v)
                    t = timeit(call, iters=10 * ncopy, warm=ncopy)
                    for k in opts:
                        hip.set_option(k, None)
                    print(f"{name} M={M} N={N} K={K} [{sname}]: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.0f} TFLOP/s", flush=True)
                t = timeit(lib_call, iters=10 * ncopy, warm=ncopy)
                print(f"{name} M={M} N={N} K={K} [torch.matmul, no epilogue]: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.0f} TFLOP/s", flush=True)
            del ws, wt
    if "convsmall_cold" in which:   # the VAE's 3x3x3 convs of ONE 512 x 512 frame (cfg1), replayed on one weight (cache-hot)
        # and with the weight rotated through > 300 MB of copies (what the pipeline sees: every layer has its own)
        for (H, W, Cin, Cout) in ((64, 64, 512, 512), (128, 128, 512, 512), (256, 256, 256, 256), (512, 512, 128, 128)):
            x = rb(1, H, W, Cin)
            ncopy = max(2, int(300e6 // (Cout * Cin * 27 * 2)) + 1)
            ws = [hip.conv_weight_tiled(rb(Cout, Cin, 3, 3, 3, scale=0.02)) for _ in range(ncopy)]
            b = torch.zeros(Cout, device=dev)
            out = torch.empty(1, H, W, Cout, dtype=torch.bfloat16, device=dev)
            it = [0]
            def call(rot):
                w = ws[it[0] % ncopy if rot else 0]; it[0] += 1
                hip.conv3d(x, w, Cout, (3, 3, 3), npad=2, pad_hw=(1, 1, 1, 1), bias=b, out=out, w_layout=hip.WL_TILED_128x32,
                           gn_stats=True)
            fl = 2 * H * W * Cin * Cout * 27
            for r in range(2):
                th = timeit(lambda: call(False), iters=10 * ncopy, warm=ncopy)
                tc = timeit(lambda: call(True), iters=10 * ncopy, warm=ncopy)
                print(f"conv {H}x{W} {Cin}->{Cout}: hot {th*1e6:.1f} us ({fl/th/1e12:.0f} TFLOP/s)  cold {tc*1e6:.1f} us "
                      f"({fl/tc/1e12:.0f} TFLOP/s)  x{tc/th:.2f}", flush=True)
            del ws
    if "shortcut4k" in which:   # the decoder's 1x1x1 shortcut conv 256 -> 128 on four 4K frames (HBM-bound: 17 GB in, 8.5 out)
        T, H, W, Cin, Cout = 4, 2160, 3840, 256, 128
        x = rb(T, H, W, Cin)
        w5 = rb(Cout, Cin, 1, 1, 1, scale=1 / 16)
        wr = hip.pad_weight(hip.conv_weight(w5))
        b = torch.zeros(Cout, device=dev)
        out = torch.empty(T, H, W, Cout, dtype=torch.bfloat16, device=dev)
        nbytes = x.numel() * 2 + out.numel() * 2
        def conv():
            hip.conv3d(x, wr, Cout, (1, 1, 1), npad=0, pad_hw=(0, 0, 0, 0), bias=b, out=out)
        def dense():
            hip.gemm(x.view(-1, Cin), wr, Cout, out.view(-1, Cout), bias=b)
        for name, fn, opts in (("conv path", conv, {}), ("conv path, BM=256", conv, {"SVR2_GEMM_BM": 256}),
                               ("conv path, gemm2", conv, {"SVR2_GEMM_V": 3}), ("dense gemm", dense, {}),
                               ("dense gemm, BM=256", dense, {"SVR2_GEMM_BM": 256}),
                               ("dense gemm, gemm2 256x128", dense, {"SVR2_GEMM_V": 3, "SVR2_GEMM4": 0}),
                               ("torch.matmul", lambda: torch.matmul(x.view(-1, Cin), w5.view(Cout, Cin).t()), {})):
            for k, v in opts.items():
                hip.set_option(k, v)
            t = timeit(fn, iters=5, warm=2)
            for k in opts:
                hip.set_option(k, None)
            print(f"shortcut 4x4K 256->128 [{name}]: {t*1e3:.2f} ms  {nbytes/t/1e12:.2f} TB/s", flush=True)
    if "pixshuf4k" in which:   # the last Upsample3D of the decoder: 1x1x1 conv 256 -> 1024 + pixel shuffle, 1080p -> 4K, 4 frames
        T, H, W, Cc, N = 4, 1080, 1920, 256, 1024
        x = rb(T, H, W, Cc)
        w = hip.pad_weight(rb(N, Cc, scale=1 / 16))
        b = torch.zeros(N, device=dev)
        nbytes = x.numel() * 2 + T * H * W * N * 2
        fn = lambda: hip.gemm(x.view(T * H * W, Cc), w, N, epilogue=hip.EPI_PIXSHUF3D, bias=b, pixshuf=(T, H, W, Cc, 1, False))
        outp = torch.empty(T * H * W, N, dtype=torch.bfloat16, device=dev)
        for name, opts in (("default", {}), ("gemm2 256x128", {"SVR2_GEMM2_NARROW": 1})):   # same GEMM, row-major stores
            for k, v in opts.items():
                hip.set_option(k, v)
            t = timeit(lambda: hip.gemm(x.view(T * H * W, Cc), w, N, outp, bias=b), iters=5, warm=2)
            for k in opts:
                hip.set_option(k, None)
            print(f"same GEMM with the plain bias epilogue [{name}]: {t*1e3:.2f} ms  {nbytes/t/1e12:.2f} TB/s", flush=True)
        del outp
        for name, opts in (("default (gemm2 256x256)", {}), ("gemm2 256x128", {"SVR2_GEMM2_NARROW": 1}),
                           ("gemm 128x128 2-deep", {"SVR2_GEMM_V": 1}), ("gemm 256x128 3-deep", {"SVR2_GEMM_V": 1, "SVR2_GEMM_BM": 256})):
            for k, v in opts.items():
                hip.set_option(k, v)
            t = timeit(fn, iters=5, warm=2)
            for k in opts:
                hip.set_option(k, None)
            print(f"pixel-shuffle GEMM 4x1080p 256->1024 [{name}]: {t*1e3:.2f} ms  {nbytes/t/1e12:.2f} TB/s  "
                  f"{2*T*H*W*Cc*N/t/1e12:.0f} TFLOP/s", flush=True)
    if "gnapply" in which:   # GroupNorm apply + SiLU alone (statistics given as partials): 4 frames of 4K, 128 channels
        Fr, H, W, Cc = 4, 2160, 3840, 128
        x = rb(Fr, H, W, Cc)
        g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        out = torch.empty_like(x)
        buf = torch.zeros(Fr, 64, 2, Cc, device=dev); buf[:, :, 1] = H * W / 64.0
        part = hip.GnPartial(buf, 64)
        t = timeit(lambda: hip.groupnorm_silu(x, g, b, 32, 1e-6, out=out, partial=part), iters=10, warm=3)
        print(f"gn finalize+apply {tuple(x.shape)}: {t*1e3:.3f} ms  {2*x.numel()*2/t/1e12:.2f} TB/s", flush=True)
        y2 = torch.empty_like(x)
        t = timeit(lambda: y2.copy_(x), iters=10, warm=3)
        print(f"torch copy of the same tensor: {t*1e3:.3f} ms  {2*x.numel()*2/t/1e12:.2f} TB/s", flush=True)
    if "gemmsquare" in which:   # square bf16 GEMMs on random data, for comparison with published MI355X templates
        for n in (4096, 8192):
            for r in range(2):
                bench_gemm(n, n, n)
            a_, b_ = rb(n, n), rb(n, n)
            t = timeit(lambda: torch.matmul(a_, b_.t()), iters=10, warm=3)
            print(f"torch.matmul {n}^3: {t*1e3:.3f} ms  {2*n**3/t/1e12:.0f} TFLOP/s", flush=True)
    if "convsplit" in which:   # latent-resolution 512-channel convs of one 1080p / 1440p frame: tiles between 192 and 1024
        for (H, W) in ((135, 240), (135, 480), (90, 160), (180, 320), (68, 120), (200, 256)):
            x = rb(1, H, W, 512)
            w = hip.conv_weight_tiled(rb(512, 512, 3, 3, 3, scale=0.02))
            b = torch.zeros(512, device=dev)
            out = torch.empty(1, H, W, 512, dtype=torch.bfloat16, device=dev)
            fl = 2 * H * W * 512 * 512 * 27
            tiles = ((H + 15) // 16) * ((W + 15) // 16) * 2
            for mode in (0, 1):
                hip.set_option("SVR2_CONV_SPLIT", mode)
                t = timeit(lambda: hip.conv3d(x, w, 512, (3, 3, 3), npad=2, pad_hw=(1, 1, 1, 1), bias=b, out=out,
                                              w_layout=hip.WL_TILED_128x32, gn_stats=True), iters=20, warm=3)
                print(f"conv {H}x{W} 512->512 ({tiles} tiles) SVR2_CONV_SPLIT={mode}: {t*1e6:.0f} us  {fl/t/1e12:.0f} TFLOP/s", flush=True)
            hip.set_option("SVR2_CONV_SPLIT", None)
    if "gateres" in which:   # the two gated-residual projections of a NaDiT-3B layer at the 4K batch size (gemm4), A/B of the
        # residual prefetch in the epilogue; then the NaDiT-7B pair
        for (M, N, K_) in ((194400, 2560, 2560), (194400, 2560, 6912), (194400, 3072, 3072), (194400, 3072, 12288)):
            for r in range(2):
                for mode in (0, 1):
                    hip.set_option("SVR2_EPI_PREF", mode)
                    print(f"SVR2_EPI_PREF={mode}: ", end="")
                    bench_gemm(M, N, K_, "gate")
            hip.set_option("SVR2_EPI_PREF", None)
    if "attn4k" in which:   # the 4K window set (T=21 -> latent grid (6,135,240)), NaDiT-3B heads; then the 7B head count
        vers = tuple(int(x) for x in os.environ.get("SVR2_KB_ATTN_VERS", "1,5").split(","))
        bench_attn_ab((6, 135, 240), 20, vers)
        if os.environ.get("SVR2_KB_ATTN_7B"):
            bench_attn_ab((6, 135, 240), 24, vers, rounds=1)
    if "color" in which:
        bench_color(17, 2160, 3840)
    if "gn" in which:
        bench_gn(4, 2160, 3840, 128)
        bench_gn(4, 1080, 1920, 256)
        bench_gn(4, 2160, 3840, 256)
    if "gemmdit" in which:   # the four GEMMs of a NaDiT-3B layer at the 4K batch size, proj_out with its row gather
        print("SVR2_GEMM4 =", os.environ.get("SVR2_GEMM4", "1"))
        M = 194400
        bench_gemm(M, 7680, 2560)
        bench_gemm(M, 2560, 2560, "gate", gather=True)
        bench_gemm(M, 13824, 2560, "swiglu")
        bench_gemm(M, 2560, 6912, "gate")
    if "gemmrounds" in which:   # one, four and many rounds of 256 x 256 tiles on 256 CUs: how much of gemm4's DMA cost is
        # lost L2 sharing between workgroups that drift apart in time (first round: every sharer starts together)
        bench_gemm(8192, 2048, 2560)
        bench_gemm(16384, 2048, 2560)
        bench_gemm(16384, 4096, 2560)
        bench_gemm(32768, 4096, 2560)
        bench_gemm(65536, 7680, 2560)
        bench_gemm(194400, 7680, 2560)
    if "gemmsmall" in which:   # the memory-bound / narrow GEMMs of the VAE (run once per SVR2_GEMM_V value)
        print("SVR2_GEMM_V =", os.environ.get("SVR2_GEMM_V", "2"))
        bench_gemm(33177600, 128, 128)        # encoder conv_in after im2col, 4 frames at 4K
        bench_gemm(777600, 1536, 512)         # mid-block attention qkv, 6 latent frames
        bench_gemm(777600, 512, 512)
        bench_gemm(8294400, 512, 256)
    if "haloresid" in which:
        for shp in ((4, 2160, 3840, 128, 128), (4, 1080, 1920, 256, 256), (2, 540, 960, 512, 512)):
            bench_conv(*shp, tiled=True)
            bench_conv(*shp, tiled=True, resid=True)
    if "halotfast" in which:   # same-box A/B of the frame-fastest tile order (SVR2_CONV_TFAST)
        for rnd in range(3):
            for v in (0, 1):
                hip.set_option("SVR2_CONV_TFAST", v)
                print(f"-- round {rnd} SVR2_CONV_TFAST={v}", flush=True)
                bench_conv(4, 2160, 3840, 128, 128, tiled=True)
                bench_conv(4, 1080, 1920, 256, 256, tiled=True)
                bench_conv(4, 1080, 1920, 512, 512, tiled=True)
                bench_conv(8, 1080, 1920, 256, 256, tiled=True)
        hip.set_option("SVR2_CONV_TFAST", None)
    if "halopref" in which:   # same-process A/B of the fragment prefetch (SVR2_HALO_PREF)
        for rnd in range(3):
            for v in (0, 1):
                hip.set_option("SVR2_HALO_PREF", v)
                print(f"-- round {rnd} SVR2_HALO_PREF={v}", flush=True)
                bench_conv(4, 2160, 3840, 128, 128, tiled=True)
                bench_conv(4, 2160, 3840, 256, 128, tiled=True)
                bench_conv(4, 1080, 1920, 256, 256, tiled=True)
                bench_conv(4, 1080, 1920, 512, 512, tiled=True)
                bench_conv(5, 272, 480, 512, 512, tiled=True)
        hip.set_option("SVR2_HALO_PREF", None)
    if "halopmc" in which:   # few launches of the three dominant halo-conv shapes for counter collection
        bench_conv(4, 2160, 3840, 128, 128, tiled=True)
        bench_conv(4, 1080, 1920, 256, 256, tiled=True)
        bench_conv(4, 1080, 1920, 512, 512, tiled=True)
    if "halo" in which:   # conv_halo.hip (tiled weights) against the tap-staged kernels on the 4K decoder shapes
        for shp in ((5, 272, 480, 512, 512), (5, 544, 960, 512, 512), (5, 1088, 1920, 256, 256),
                    (4, 1080, 1920, 512, 256), (4, 2160, 3840, 128, 128), (4, 2160, 3840, 256, 128),
                    (4, 2160, 3840, 256, 256)):
            bench_conv(*shp)
            bench_conv(*shp, tiled=True)
        bench_conv(4, 2160, 3840, 128, 3, tiled=True)
    if "conv" in which:
        bench_conv(5, 136, 240, 512, 512)
        bench_conv(5, 272, 480, 512, 512)
        bench_conv(5, 544, 960, 256, 256)
        bench_conv(5, 1088, 1920, 128, 128)
        bench_conv(4, 2160, 3840, 128, 128)
    if "io" in which:   # HBM-bound byte kernels of the frame I/O at 4K, 17 frames
        T, H, W = 17, 2160, 3840
        x = torch.rand(T, H, W, 3, device=dev).to(torch.bfloat16)
        vio = pkg.video_io
        for name, lay, byts in (("frames_to_u8 (BGR, bottom-up)", vio.Layout("rgb", W, H, flags=3), 6 + 3),
                                ("frames_to_yuv 4:2:0", vio.Layout("yuv", W, H, sub=2), 6 + 1.5)):
            out = torch.empty(T, lay.frame_bytes, dtype=torch.uint8, device=dev)
            t = timeit(lambda: vio.frames_to_bytes(x, lay, out))
            print(f"{name}: {t*1e3:.3f} ms  {T*H*W*byts/t/1e9:.0f} GB/s ({T/t:.0f} frames/s)", flush=True)
            raw = out
            t = timeit(lambda: vio.frames_from_bytes(raw, lay))
            print(f"  inverse: {t*1e3:.3f} ms  {T*H*W*byts/t/1e9:.0f} GB/s", flush=True)
        t = timeit(lambda: hip.jpeg_encode(x[:8], 90), iters=3, warm=1)
        print(f"jpeg_encode q90 (8 frames, noise-like worst case): {t*1e3:.1f} ms  ({8/t:.0f} frames/s)", flush=True)
        g = torch.Generator().manual_seed(1)
        low = 0.5 + 0.2 * torch.randn(8, 3, 135, 240, generator=g)
        sm = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic").clamp(0, 1).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
        t = timeit(lambda: hip.jpeg_encode(sm, 90), iters=3, warm=1)
        _, lens = hip.jpeg_encode(sm, 90)
        print(f"jpeg_encode q90 (8 smooth frames): {t*1e3:.1f} ms  ({8/t:.0f} frames/s, {lens.float().mean().item()/1e6:.2f} MB per frame)", flush=True)
    if "attn5" in which:
        bench_attn512(1, 129600)
        bench_attn512(2, 32640)
    if "attn" in which:
        bench_attn((3, 68, 120), pkg.windows.REGULAR)
        bench_attn((5, 135, 240), pkg.windows.REGULAR)
        bench_attn((5, 135, 240), pkg.windows.SHIFTED)
    if "attn512" in which:
        bench_attn512(2, 32640)
        bench_attn512(1, 129600)
    if "norm" in which:
        bench_norm(162000, 2560)
