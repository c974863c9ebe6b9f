#!/usr/bin/env python3
"""Headline benchmark: upscaled frames/s (+ peak HBM) of the SeedVR2-3B hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg1]

A "step" is one pass of the hot path over one temporal batch of synthetic LR video on every rank:
antialiased-bicubic transform -> causal VAE encode -> one-step NaDiT-3B -> causal VAE decode -> crop,
Hann-blend of the `temporal_overlap` frames with the previous batch's tail; with N > 1 every rank owns
its own contiguous frame segment (weak scaling along the frame-chunk axis) and after its batch sends the
first `overlap` decoded frames to its left neighbour, which blends them into its tail (RCCL send/recv over
xGMI) - the only exchange the path has.  Inputs are resident in HBM when the timed region starts; weights
are random-init tensors of the real architecture (no checkpoints offline) -> "data": "synthetic".

Rank 0 prints ONE JSON line (see the driver contract in the task description).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

WORKLOADS = {
    # name: (frames per batch, LR h, LR w, resolution, temporal_overlap, description)
    "cfg3": (21, 720, 1280, 2160, 4, "SeedVR2-3B bf16, 720p->4K, batch_size 21, temporal_overlap 4 (BASELINE configs[2])"),
    "cfg2": (9, 540, 960, 1080, 0, "SeedVR2-3B bf16, 8(+1 pad)-frame 540p->1080p (BASELINE configs[1])"),
    "cfg1": (1, 256, 256, 512, 0, "SeedVR2-3B bf16, single 256x256 image x2 (BASELINE configs[0])"),
    "cfg2o": (9, 540, 960, 1080, 4, "rehearsal: cfg2 shape with temporal_overlap 4 (exercises the neighbour exchange)"),
    "cfg4": (21, 720, 1280, 2160, 4, "SeedVR2-7B bf16, 720p->4K, batch_size 21, temporal_overlap 4 (BASELINE configs[3])"),
}
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synth_lr_video(t, h, w, seed, device):
    """Seeded smooth content: bicubic-upsampled low-res noise, clamped, uint8-quantised then /255
    (the CLI feeds uint8 frames / 255, inference_cli.py:543)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    low = 0.5 + 0.2 * torch.randn(t, 3, max(h // 16, 2), max(w // 16, 2), generator=g)
    x = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
    x = (x * 255).round().to(torch.uint8).float() / 255.0
    return x.permute(0, 2, 3, 1).contiguous().to(device).to(torch.bfloat16)


class KernelTimer:
    """HIP-event timing of one kernel family inside the timed region (events recorded on the launching
    stream = torch's current stream, which is the stream the C ABI launches on)."""

    def __init__(self):
        self.pairs = []
        self.flops = 0.0
        self.enabled = False
        self.only = None   # optional predicate on the call's arguments: which launches belong to the kernel

    def wrap(self, fn, flop_fn):
        def inner(*a, **k):
            if not self.enabled or (self.only is not None and not self.only(*a, **k)):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = fn(*a, **k)
            e1.record()
            out = res[0] if isinstance(res, tuple) else res   # conv3d(..., gn_stats=True) returns (y, partial sums)
            f = flop_fn(out, *a, **k)
            self.pairs.append((e0, e1, f, (tuple(a[0].shape), tuple(out.shape), a[3], k.get("carry") is not None,
                                           k.get("resid") is not None)))
            self.flops += f
            return res
        return inner

    def summary(self):
        if not self.pairs:
            return 0, 0.0, 0.0
        t = sum(p[0].elapsed_time(p[1]) for p in self.pairs) * 1e-3
        return len(self.pairs), t, self.flops

    def table(self):
        import collections
        agg = collections.OrderedDict()
        for e0, e1, f, key in self.pairs:
            d = agg.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return [f"{k}: calls={v[0]} time={v[1]*1e3:.1f}ms {v[2]/v[1]/1e12:.0f} TFLOP/s" for k, v in rows]


def cpu_baseline(workload, n_threads):
    """Reference CPU path (the oracle, a port) timed on a bounded sample on this host's cores and
    extrapolated per stage by algorithmic FLOPs (336 TFLOP per 4K frame is days of CPU time)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from seedvr2_ref import dit as odit, vae as ovae
    torch.set_num_threads(n_threads)
    T, hin, win, res, ov, _ = WORKLOADS[workload]
    # (a) one full-width NaDiT-3B block at 4,096 video tokens + 58 text tokens, bf16 like the reference CPU path
    cfg = odit.DiTConfig(num_layers=2, mm_layers=1)
    p = {k: v.to(torch.bfloat16) for k, v in odit.synthetic_params(cfg, 0).items() if k.startswith("blocks.0.")}
    x = torch.randn(4096, cfg.vid_dim).to(torch.bfloat16)
    y = torch.randn(58, cfg.vid_dim).to(torch.bfloat16)
    emb = torch.randn(6 * cfg.vid_dim).to(torch.bfloat16)
    odit.block(p, cfg, 0, x[:256], y, emb, (1, 16, 16), torch.bfloat16, "3b")  # warm the thread pool
    t0 = time.time()
    for _ in range(3):
        odit.block(p, cfg, 0, x, y, emb, (1, 64, 64), torch.bfloat16, "3b")
    t_dit = (time.time() - t0) / 3
    f_dit = 2 * 4096 * (2560 * 7680 + 2560 * 2560 + 3 * 2560 * 6912)
    del p
    # (b) the VAE on a 9-frame 128x128 clip (3 latent frames of 16x16), bf16
    vcfg = ovae.VAEConfig()
    vp = {k: v.to(torch.bfloat16) for k, v in ovae.synthetic_params(vcfg, 0).items()}
    xv = (torch.rand(3, 9, 128, 128) * 2 - 1).to(torch.bfloat16)
    t0 = time.time()
    mom = ovae.encoder(vp, xv, vcfg)
    t_enc = time.time() - t0
    t0 = time.time()
    ovae.decoder(vp, mom[:16], vcfg)
    t_dec = time.time() - t0
    f_enc, f_dec = _vae_flops(9, 128, 128)
    rate_dit, rate_enc, rate_dec = f_dit / t_dit, f_enc / t_enc, f_dec / t_dec
    _, _, _, _, ph, pw = ge.load_package().pipeline.target_dims(hin, win, res)
    enc_f, dec_f = _vae_flops(T, ph, pw)
    dit_f = _dit_flops(T, ph, pw)
    t_step = enc_f / rate_enc + dec_f / rate_dec + dit_f / rate_dit
    new_frames = T - ov
    return {
        "value": new_frames / t_step, "unit": "frames/s", "cores": n_threads, "kind": "port",
        "sample": (f"oracle (bf16, torch CPU, {n_threads} threads): NaDiT-3B block @4096 tokens ({t_dit:.1f}s, "
                   f"{rate_dit/1e12:.2f} TFLOP/s) + VAE encode/decode of 9 frames 128x128 ({t_enc:.1f}s/{t_dec:.1f}s, "
                   f"{rate_enc/1e12:.2f}/{rate_dec/1e12:.2f} TFLOP/s), extrapolated per stage by algorithmic FLOPs to "
                   f"the {workload} batch"),
    }


def _vae_flops(T, H, W):
    """Algorithmic FLOPs (2*MACs) of VAE encode / decode for T frames of HxW."""
    Tl = (T - 1) // 4 + 1
    conv = lambda t, h, w, ci, co, k=27: 2.0 * t * h * w * ci * co * k
    res = lambda t, h, w, ci, co: conv(t, h, w, ci, co) + conv(t, h, w, co, co) + (conv(t, h, w, ci, co, 1) if ci != co else 0)
    attn = lambda t, h, w: 4.0 * 512 * (h * w) ** 2 * t + 2.0 * t * h * w * 512 * 512 * 4
    T2 = (T - 1) // 2 + 1
    enc = conv(T, H, W, 3, 128) + 2 * res(T, H, W, 128, 128) + conv(T, H // 2, W // 2, 128, 128, 9)
    enc += res(T, H // 2, W // 2, 128, 256) + res(T, H // 2, W // 2, 256, 256) + conv(T2, H // 4, W // 4, 256, 256)
    enc += res(T2, H // 4, W // 4, 256, 512) + res(T2, H // 4, W // 4, 512, 512) + conv(Tl, H // 8, W // 8, 512, 512)
    enc += 4 * res(Tl, H // 8, W // 8, 512, 512) + attn(Tl, H // 8, W // 8) + conv(Tl, H // 8, W // 8, 512, 32)
    h8, w8 = H // 8, W // 8
    dec = conv(Tl, h8, w8, 16, 512) + 2 * res(Tl, h8, w8, 512, 512) + attn(Tl, h8, w8) + 3 * res(Tl, h8, w8, 512, 512)
    dec += conv(Tl, h8, w8, 512, 4096, 1) + conv(T2, h8 * 2, w8 * 2, 512, 512)
    dec += 3 * res(T2, h8 * 2, w8 * 2, 512, 512) + conv(T2, h8 * 2, w8 * 2, 512, 4096, 1) + conv(T, h8 * 4, w8 * 4, 512, 512)
    dec += res(T, h8 * 4, w8 * 4, 512, 256) + 2 * res(T, h8 * 4, w8 * 4, 256, 256) + conv(T, h8 * 4, w8 * 4, 256, 1024, 1)
    dec += conv(T, H, W, 256, 256) + res(T, H, W, 256, 128) + 2 * res(T, H, W, 128, 128) + conv(T, H, W, 128, 3)
    return enc, dec


def _dit_flops(T, H, W, model="3b"):
    Tl = (T - 1) // 4 + 1
    L = Tl * (H // 16) * (W // 16)
    if model == "7b":
        return 36 * 2.0 * L * (3072 * 9216 + 3072 * 3072 + 2 * 3072 * 12288)
    return 32 * 2.0 * L * (2560 * 7680 + 2560 * 2560 + 3 * 2560 * 6912)  # GEMMs only (95 % of the DiT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SVR2_BENCH_WORKLOAD", "cfg3"), choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    backend = os.environ.get("SVR2_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 control flow on one GPU
    if backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")

    pkg = ge.load_package()
    hip = pkg.hip
    hip.device_check(local)

    T, hin, win, res, overlap, desc = WORKLOADS[args.workload]
    t_build = time.time()
    vcfg = pkg.vae.VAEConfig()
    if os.environ.get("SVR2_DECODE_SPLIT"):   # A/B of the temporal chunk sizes (results do not depend on them)
        vcfg.decode_split = int(os.environ["SVR2_DECODE_SPLIT"])
    if os.environ.get("SVR2_ENCODE_SPLIT"):
        vcfg.encode_split = int(os.environ["SVR2_ENCODE_SPLIT"])
    vae = pkg.vae.VideoVAEHip(vcfg, pkg.vae.synthetic_state_dict(vcfg, dev, seed=1), dev)
    dcfg = pkg.dit.CFG_7B if args.workload == "cfg4" else pkg.dit.CFG_3B
    dit = pkg.dit.NaDiTHip(dcfg, pkg.dit.synthetic_state_dict(dcfg, dev, seed=2), dev)
    gtxt = torch.Generator(device="cpu").manual_seed(3)
    text_pos = (0.5 * torch.randn(58, 5120, generator=gtxt)).to(torch.bfloat16)  # shape of the shipped pos_emb.pt
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, text_pos, dev)
    torch.cuda.synchronize()
    log(f"[rank {rank}] models built in {time.time() - t_build:.1f}s; weights {torch.cuda.memory_allocated() / 2**30:.1f} GiB")

    rh, rw, th, tw, ph, pw = pkg.pipeline.target_dims(hin, win, res)
    lr = synth_lr_video(T, hin, win, 1234 + rank, dev)            # resident in HBM before timing
    latent_shape = ((T - 1) // 4 + 1, ph // 8, pw // 8, 16)
    noise = pipe.make_noise(latent_shape, 42).to(dev)             # CPU-generator noise, uploaded once
    w_prev = pkg.pipeline.blend_weights(overlap).to(dev) if overlap else None
    prev_tail = [None]
    new_frames = T - overlap

    ktimer = KernelTimer()
    # the roofline kernel = conv_halo_kernel (all three geometries): the conv3d calls with the tiled weight layout
    ktimer.only = lambda *a, **kw: kw.get("w_layout", 0) == hip.WL_TILED_128x32
    hip.conv3d = ktimer.wrap(hip.conv3d, lambda out, x, w, cout, k, *a, **kw: 2.0 * out.numel() * x.shape[3] * k[0] * k[1] * k[2])
    pkg.vae.hip = hip

    def step():
        dec = pipe.upscale_batch(lr, res, 42, noise=noise)
        sample = hip.crop_to_unit(dec, th, tw, to_unit=False)
        del dec
        if overlap:
            if prev_tail[0] is not None:  # blend with the previous batch's tail (same rank)
                hip.blend_overlap(prev_tail[0], sample[:overlap].contiguous(), w_prev, prev_tail[0])
            if world > 1:                 # segment boundary: head frames go to the left neighbour
                ops = []
                stage = (lambda t: t.cpu()) if backend == "gloo" else (lambda t: t)  # gloo p2p needs host tensors
                recv = stage(torch.empty_like(sample[:overlap]))
                if rank > 0:
                    ops.append(dist.P2POp(dist.isend, stage(sample[:overlap].contiguous()), rank - 1))
                if rank < world - 1:
                    ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
                recv = recv.to(dev)
                if rank < world - 1:
                    tail = sample[-overlap:].contiguous()
                    hip.blend_overlap(tail, recv, w_prev, tail)
            prev_tail[0] = sample[-overlap:].clone()
        out = hip.crop_to_unit(sample[: new_frames].contiguous() if overlap else sample, th, tw, to_unit=True)
        return out

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        t0 = time.time()
        step()
        torch.cuda.synchronize()
        log(f"[rank {rank}] warmup step {i}: {time.time() - t0:.2f}s")
    torch.cuda.reset_peak_memory_stats()
    ktimer.enabled = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ktimer.enabled = False
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    peak_gb = torch.cuda.max_memory_reserved() / 1e9
    log(f"[rank {rank}] peak reserved {peak_gb:.1f} GB, peak allocated {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")

    if rank == 0:
        nl, kt, kf = ktimer.summary()
        if os.environ.get("SVR2_BENCH_TABLE"):
            for line in ktimer.table():
                log("  conv3d", line)
        enc_f, dec_f = _vae_flops(T, ph, pw)
        dit_f = _dit_flops(T, ph, pw, "7b" if args.workload == "cfg4" else "3b")
        achieved = kf / kt / 1e12 if kt > 0 else 0.0
        traffic = None   # HBM-side bytes per conv launch from the committed rocprofv3 PMC passes of this workload
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic_cfg3.json")
        if args.workload == "cfg3" and os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    traffic = json.load(fh)["kernels"]["conv3d halo-resident (conv_halo*_kernel)"]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        result = {
            "metric": "upscaled frames/sec + peak HBM GB, SeedVR2-3B 720p->4K, 1/2/4/8 MI355X",
            "value": new_frames * world * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "peak_hbm_gb": peak_gb,
            "config": {"workload": f"{args.workload}: {desc}", "frames_per_batch": T, "new_frames_per_step": new_frames,
                       "lr": [hin, win], "out": [th, tw], "padded": [ph, pw], "parallelism": f"frame-chunk x{world}",
                       "algorithmic_tflop_per_step": {"vae_encode": enc_f / 1e12, "dit_gemm": dit_f / 1e12,
                                                      "vae_decode": dec_f / 1e12}},
            "roofline": {"bound": "mfma", "kernel": "conv_halo_wr_kernel<16,256|32,128> + conv_halo_kernel<64,16> (halo-resident causal conv3d: every 3x3 stride-1 conv of the VAE)",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (FETCH_SIZE*2 + WRITE_SIZE, rocprofv3 PMC, profiles/r01_hbm_traffic_cfg3.json)",
                         "algorithmic_flop_per_launch": kf / nl if nl else None,
                         "launches": nl, "kernel_time_s": kt, "share_of_step_time": kt / dt if dt > 0 else None},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args.workload, min(os.cpu_count() or 1, 64))
            except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
