#!/usr/bin/env python3
"""Headline benchmark: upscaled frames/s (+ peak HBM) of the SeedVR2-3B hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg4|cfg5|cfg2|cfg1]

A "step" is one pass of the hot path over one temporal batch of the workload's synthetic LR clip on every rank:
antialiased-bicubic transform -> causal VAE encode -> noise (CPU generator, like the reference's CPU path) + upload ->
one-step NaDiT -> causal VAE decode -> crop, Hann-blend of the `temporal_overlap` frames with the previous batch's
tail, write into the clip's output.  Steps cycle through the batches of the clip exactly as `generate()` walks them
(cfg3: 64 frames, batch_size 21, overlap 4 -> batches of 21/21/21/13 frames, 4 steps = one clip = 64 finished frames),
so `value` = finished frames / time over whole and partial clips.  With N > 1 every rank owns its own contiguous
frame segment (weak scaling along the frame-chunk axis; cfg5: 64 frames + 4 overlap per rank = the reference's
streaming split of a 512-frame clip over 8 GPUs) and, when its segment is complete, sends its first `overlap` finished
frames to its left neighbour, which cross-fades them into its tail (RCCL send/recv over xGMI, parallel.py) - the only
exchange the path has.  The LR clip is resident in HBM when the timed region starts (cfg5: as 8-bit container pixels,
converted on the device per batch like the streaming reader does); weights are random-init tensors of the real
architecture (no checkpoints offline) -> "data": "synthetic".

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
    # name: (model, clip frames per rank, batch_size, LR h, LR w, resolution, temporal_overlap, description)
    "cfg3": ("3b", 64, 21, 720, 1280, 2160, 4, "SeedVR2-3B bf16, 64-frame 720p->4K, batch_size 21, temporal_overlap 4 (BASELINE configs[2])"),
    "cfg4": ("7b", 64, 21, 720, 1280, 2160, 4, "SeedVR2-7B bf16, 64-frame 720p->4K, batch_size 21, temporal_overlap 4 (BASELINE configs[3])"),
    "cfg5": ("7b", 68, 21, 720, 1280, 2160, 4, "SeedVR2-7B bf16, 512-frame 720p->4K streamed, frame-chunk shard: 64 frames (+4 overlap) per rank (BASELINE configs[4])"),
    "cfg2": ("3b", 8, 9, 540, 960, 1080, 0, "SeedVR2-3B bf16, 8-frame (padded to 9) 540p->1080p (BASELINE configs[1])"),
    "cfg1": ("3b", 1, 1, 256, 256, 512, 0, "SeedVR2-3B bf16, single 256x256 image x2 (BASELINE configs[0])"),
    "cfg2o": ("3b", 26, 9, 540, 960, 1080, 4, "rehearsal: 540p->1080p clip of 26 frames, batch 9, overlap 4 (exercises blend + neighbour exchange)"),
    "cfg3t65": ("3b", 65, 65, 720, 1280, 2160, 0, "memory rehearsal: ONE batch of 65 frames 720p->4K (DiT grid (17,135,240), 550 800 tokens: SURVEY 8's largest grid)"),
    "cfg0": ("3b", 26, 9, 32, 48, 64, 4, "rehearsal of the launcher / exchange / JSON contract: 32x48 -> 64x96 clip of 26 frames, batch 9, overlap 4"),
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


def _measured_cpu(workload, model, pkg, clip, T, hin, win, res, ov):
    """The oracle's MEASURED seconds at this workload's true spatial size (tests/fullsize_golden.py ran it stage by stage on
    the GPU box's 64 host threads and committed the seconds with the golden crops, tests/golden/fullsize/*.json):
    cfg2 = the whole workload end to end; the 4K workloads = one 4K frame end to end, per stage, turned into rates
    (algorithmic FLOPs / seconds at 4K shapes) and summed over the clip's batches.  None when no record exists."""
    gdir = os.path.join(ROOT, "tests", "golden", "fullsize")
    name = {"cfg2": "cfg2", "cfg3": "4k3b", "cfg4": "4k7b", "cfg5": "4k7b"}.get(workload)
    path = os.path.join(gdir, f"{name}.json") if name else None
    if not path or not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    st = rec.get("oracle_stage_seconds")
    if not st:
        return None
    if workload == "cfg2":
        secs = rec["oracle_bf16_seconds"]
        return {"value": rec["frames"] / secs, "seconds": secs, "threads": rec["cpu_threads"],
                "what": f"the whole cfg2 workload end to end through the oracle: {secs:.0f} s (tests/golden/fullsize/cfg2.json)"}
    _, _, _, _, ph, pw = pkg.pipeline.target_dims(hin, win, res)
    enc1, dec1 = _vae_flops(1, ph, pw)
    r_enc, r_dec = enc1 / st["vae_encode"], dec1 / st["vae_decode"]
    r_dit = _dit_flops(1, ph, pw, model) / st["dit"]
    batches, _ = pkg.pipeline.plan_batches(clip, T, ov)
    t_clip = 0.0
    for b in batches:
        e, d = _vae_flops(len(b.index), ph, pw)
        t_clip += e / r_enc + d / r_dec + _dit_flops(len(b.index), ph, pw, model) / r_dit + st.get("transform", 0.0) * len(b.index)
    return {"value": clip / t_clip, "seconds_one_4k_frame": rec["oracle_bf16_seconds"], "threads": rec["cpu_threads"],
            "frames_per_s_one_4k_frame": 1.0 / rec["oracle_bf16_seconds"],
            "what": (f"one 4K frame end to end through the oracle MEASURED per stage ({rec['oracle_bf16_seconds']:.0f} s: encode "
                     f"{st['vae_encode']:.0f} s, NaDiT-{model} {st['dit']:.0f} s, decode {st['vae_decode']:.0f} s; "
                     f"tests/golden/fullsize/{name}.json), its per-stage rates at 4K shapes "
                     f"({r_enc/1e12:.2f}/{r_dit/1e12:.2f}/{r_dec/1e12:.2f} TFLOP/s) applied to the clip's batches")}


def cpu_baseline(workload, n_threads, pkg=None, vsd=None, dsd=None):
    """Reference CPU path (the oracle, a port) on the GPU box's host cores.
    cfg1: the workload itself, end to end, live (transform -> encode -> NaDiT -> decode, bf16 like the reference's CPU path,
    on the same synthetic weights as the GPU run).  Larger workloads: `value` comes from the oracle's MEASURED seconds at
    the workload's true spatial size (_measured_cpu: minutes to hours of CPU, run once on this pool's 64 host threads and
    committed); the live bounded sample of every run (one full-width NaDiT block at 4,096 tokens and the VAE on a 9-frame
    128 x 128 clip, extrapolated per stage by algorithmic FLOPs) is reported beside it as `live_sample_value` - it runs
    small, cache-friendly shapes and is 1.5-2 x kinder to the CPU than the measurement at size."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from seedvr2_ref import dit as odit, pipeline as opipe, vae as ovae
    torch.set_num_threads(n_threads)
    model, clip, T, hin, win, res, ov, _ = WORKLOADS[workload]
    if workload == "cfg1" and vsd is not None and dsd is not None:
        vp = {k: v.detach().cpu() for k, v in vsd.items()}
        dp = {k: v.detach().cpu() for k, v in dsd.items()}
        txt = (0.5 * torch.randn(58, 5120, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
        lr = synth_lr_video(1, hin, win, 1234, "cpu").float()
        t0 = time.time()
        out = opipe.generate(lr, vp, ovae.VAEConfig(), dp, odit.CFG_3B, txt, res, 1, 0, 42, torch.bfloat16)
        dt = time.time() - t0
        assert tuple(out.shape[:3]) == (1, res, res)
        return {"value": 1.0 / dt, "unit": "frames/s", "cores": n_threads, "kind": "port",
                "sample": f"the whole cfg1 workload end to end through the oracle (bf16, torch CPU, {n_threads} threads): {dt:.1f} s"}
    # (a) one full-width NaDiT block at 4,096 video tokens + 58 text tokens, bf16 like the reference CPU path
    cfg = odit.DiTConfig(num_layers=2, mm_layers=1) if model == "3b" else \
        odit.DiTConfig(vid_dim=3072, heads=24, num_layers=2, mm_layers=2, mlp_type="normal", out_norm=False, last_vid_only=False)
    p = {k: v.to(torch.bfloat16) for k, v in odit.synthetic_params(cfg, 0).items() if k.startswith("blocks.0.")}
    x = torch.randn(4096, cfg.vid_dim).to(torch.bfloat16)
    y = torch.randn(58, cfg.vid_dim).to(torch.bfloat16)
    emb = torch.randn(6 * cfg.vid_dim).to(torch.bfloat16)
    rope = "3b" if model == "3b" else "7b"
    odit.block(p, cfg, 0, x[:256], y, emb, (1, 16, 16), torch.bfloat16, rope)  # warm the thread pool
    t0 = time.time()
    for _ in range(3):
        odit.block(p, cfg, 0, x, y, emb, (1, 64, 64), torch.bfloat16, rope)
    t_dit = (time.time() - t0) / 3
    f_dit = _dit_flops(1, 1024, 1024, model) / (32 if model == "3b" else 36)
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
    batches, _ = ge.load_package().pipeline.plan_batches(clip, T, ov)
    t_clip = 0.0
    for b in batches:
        enc_f, dec_f = _vae_flops(len(b.index), ph, pw)
        t_clip += enc_f / rate_enc + dec_f / rate_dec + _dit_flops(len(b.index), ph, pw, model) / rate_dit
    live = (f"live bounded sample of this run (oracle, bf16, torch CPU, {n_threads} threads): NaDiT-{model} block @4096 tokens "
            f"({t_dit:.1f}s, {rate_dit/1e12:.2f} TFLOP/s) + VAE encode/decode of 9 frames 128x128 ({t_enc:.1f}s/{t_dec:.1f}s, "
            f"{rate_enc/1e12:.2f}/{rate_dec/1e12:.2f} TFLOP/s), extrapolated per stage by algorithmic FLOPs to the {workload} "
            f"clip ({clip} frames in batches of {[len(b.index) for b in batches]})")
    meas = _measured_cpu(workload, model, ge.load_package(), clip, T, hin, win, res, ov)
    if meas is None:
        return {"value": clip / t_clip, "unit": "frames/s", "cores": n_threads, "kind": "port", "sample": live}
    out = {"value": meas["value"], "unit": "frames/s", "cores": meas["threads"], "kind": "port",
           "sample": "port, measured at the workload's true spatial size: " + meas["what"],
           "live_sample_value": clip / t_clip, "live_sample": live}
    for k in ("seconds", "seconds_one_4k_frame", "frames_per_s_one_4k_frame"):
        if k in meas:
            out[k] = meas[k]
    return out


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


class Runtime:
    """What `main` needs from the machine.  The defaults are the product: one HIP device per rank, RCCL, the HIP pipeline.
    tests/bench_cpu_child.py substitutes a CPU flavour (gloo, a per-frame stand-in for the models) to drive the
    self-launch, the segment arithmetic, the exchange and the JSON contract without a GPU - the same role
    `inference_cli.Runtime` plays for the CLI."""

    def __init__(self):
        self.child_cmd = [sys.executable, os.path.abspath(__file__)]
        self.backend = os.environ.get("SVR2_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 control flow on one GPU

    def device(self, local, rank):
        ndev = torch.cuda.device_count()                          # counting devices does not initialise the GPU
        if self.backend == "gloo":
            local = local % max(ndev, 1)
        elif local >= ndev:
            raise SystemExit(f"bench: rank {rank} wants cuda:{local} but this node shows {ndev} GPU(s) "
                             f"(SVR2_BENCH_BACKEND=gloo rehearses several ranks on one GPU)")
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def package(self, dev):
        pkg = ge.load_package()
        pkg.hip.device_check(dev.index)        # raises unless gfx950 + the HIP library: there is no fallback path
        return pkg

    def build(self, pkg, model, dev):
        """(pipeline, VAE state dict, DiT state dict): random-init weights of the real architecture."""
        vcfg = pkg.vae.VAEConfig()
        if os.environ.get("SVR2_DECODE_SPLIT"):   # A/B of the temporal chunk sizes (results do not depend on them)
            vcfg.decode_split = int(os.environ["SVR2_DECODE_SPLIT"])
        if os.environ.get("SVR2_ENCODE_SPLIT"):
            vcfg.encode_split = int(os.environ["SVR2_ENCODE_SPLIT"])
        if os.environ.get("SVR2_ENCODE_FIRST"):
            vcfg.encode_first = int(os.environ["SVR2_ENCODE_FIRST"])
        if os.environ.get("SVR2_DECODE_FIRST"):
            vcfg.decode_first = int(os.environ["SVR2_DECODE_FIRST"])
        if os.environ.get("SVR2_NORM_CARRY_COPY"):
            vcfg.norm_carry_copy = os.environ["SVR2_NORM_CARRY_COPY"] != "0"
        vsd = pkg.vae.synthetic_state_dict(vcfg, dev, seed=1)
        vae = pkg.vae.VideoVAEHip(vcfg, vsd, dev)
        dcfg = pkg.dit.CFG_7B if model == "7b" else pkg.dit.CFG_3B
        dsd = pkg.dit.synthetic_state_dict(dcfg, dev, seed=2)
        dit = pkg.dit.NaDiTHip(dcfg, dsd, dev)
        gtxt = torch.Generator(device="cpu").manual_seed(3)
        text_pos = (0.5 * torch.randn(58, 5120, generator=gtxt)).to(torch.bfloat16)  # shape of the shipped pos_emb.pt
        return pkg.pipeline.SeedVR2Pipeline(vae, dit, text_pos, dev), vsd, dsd

    def sync(self):
        torch.cuda.synchronize()

    def reset_peak(self):
        torch.cuda.reset_peak_memory_stats()

    def peak_gb(self):
        return torch.cuda.max_memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9

    def allocated_gib(self):
        return torch.cuda.memory_allocated() / 2**30

    def event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e


def self_launch(n: int, argv, rt) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh ranks of this script, one per GPU,
    BEFORE anything here has touched a GPU (this parent never does: no torch.cuda call, no package load), with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, like the reference's entry point spawns its own
    workers [ref: inference_cli.py:1048-1112].  Rank 0 prints the one JSON line on the inherited stdout; the first
    failing rank takes the others down and the exit code is non-zero.  torchrun-launched ranks never come here."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, failed = [], 0

    def stop_children(*_):
        for p in procs:                     # exactly the processes started below
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 15
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_term(signum, frame):
        stop_children()
        sys.exit(128 + signum)

    try:
        signal.signal(signal.SIGTERM, on_term)
    except ValueError:              # not the main thread: the finally clause below still cleans up
        pass
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), SVR2_BENCH_SELF_LAUNCHED="1",
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # every rank draws its noise with the CPU generator: share the host cores instead of oversubscribing them n-fold
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
            procs.append(subprocess.Popen(rt.child_cmd + list(argv), env=env))
        while any(p.poll() is None for p in procs):
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:
                failed = bad[0].returncode
                break
            time.sleep(0.2)
    finally:
        stop_children()
    codes = [p.wait() for p in procs]
    if failed or any(codes):
        log(f"bench: rank exit codes {codes}")
        return 1
    return 0


def main(argv=None, rt: Runtime = None):
    rt = rt or Runtime()
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SVR2_BENCH_WORKLOAD", "cfg3"), choices=list(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own clip, a step = one temporal batch per rank; strong: ONE clip split over "
                         "the ranks like the reference's multi-GPU CLI, a step = the whole clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, argv, rt))      # nothing above this line initialises a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: refusing to report a "
                         f"number under the wrong n_gpus")
    backend = rt.backend
    dev = rt.device(local, rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")
    cdev = dev if backend == "nccl" else "cpu"                   # where the few control all-reduces live

    pkg = rt.package(dev)
    hip = pkg.hip

    model, clip, bsz, hin, win, res, overlap, desc = WORKLOADS[args.workload]
    strong = args.scaling == "strong"
    seg0 = 0
    if strong:
        # ONE clip, split like the reference's multi-GPU mode: base + remainder frames per rank, + `overlap` for all but the
        # last rank [ref: inference_cli.py:1082-1097]; every rank synthesises the same clip and keeps its own segment
        if args.workload == "cfg5":
            clip = 512
        total = clip
        ranges = pkg.parallel.segment_ranges(total, world, overlap, bsz, True)
        seg0, seg1 = ranges[rank]
        clip = seg1 - seg0
        seg_counts = [b - a for a, b in ranges]
    else:
        if args.workload == "cfg5" and rank == world - 1 and world > 1:
            clip -= overlap        # the last rank's segment has no overlap frames [ref: inference_cli.py:1091-1093]
        seg_counts = [WORKLOADS[args.workload][1]] * world
        if args.workload == "cfg5" and world > 1:
            seg_counts[-1] -= overlap
        total = sum(pkg.parallel.owned_counts(seg_counts, overlap if world > 1 else 0))
    t_build = time.time()
    pipe, vsd, dsd = rt.build(pkg, model, dev)
    keep_sd = args.workload == "cfg1" and not args.no_cpu_baseline and world == 1   # the oracle runs the same weights
    if not keep_sd:
        vsd = dsd = None
    rt.sync()
    log(f"[rank {rank}] models built in {time.time() - t_build:.1f}s; weights {rt.allocated_gib():.1f} GiB")

    rh, rw, th, tw, ph, pw = pkg.pipeline.target_dims(hin, win, res)
    batches, overlap = pkg.pipeline.plan_batches(clip, bsz, overlap)
    nb = len(batches)
    if strong:     # the same clip on every rank (one seed), this rank's segment of it resident in HBM before timing
        lr = synth_lr_video(total, hin, win, 1234, "cpu")[seg0:seg0 + clip].to(dev)
    else:
        lr = synth_lr_video(clip, hin, win, 1234 + rank, dev)     # the rank's own clip, resident in HBM before timing
    streamed = args.workload == "cfg5"
    if streamed:   # held as the container's 8-bit pixels; every batch is converted on the device like video_io's reader does
        lr_u8 = (lr.float() * 255.0).round().to(torch.uint8).reshape(clip, -1).contiguous()
        lay = pkg.video_io.Layout("rgb", win, hin)
        del lr
    idx_dev = [torch.tensor(b.index, device=dev) for b in batches]
    w_prev = pkg.pipeline.blend_weights(overlap).to(dev) if overlap else None
    final = torch.empty(clip, th, tw, 3, dtype=torch.bfloat16, device=dev)   # the clip's output, written batch by batch
    state = {"write": 0, "frames": 0, "clips": 0, "exchange_s": 0.0, "exchanges": 0}

    ktimer = KernelTimer()
    # the roofline kernel = conv_halo_kernel (all three geometries): the conv3d calls with the tiled weight layout
    ktimer.only = lambda *a, **kw: kw.get("w_layout", 0) == hip.WL_TILED_128x32
    hip.conv3d = ktimer.wrap(hip.conv3d, lambda out, x, w, cout, k, *a, **kw: 2.0 * out.numel() * x.shape[3] * k[0] * k[1] * k[2])
    pkg.vae.hip = hip

    def exchange(out):
        """The boundary exchange of a finished segment (timed on its own: HIP events around it over RCCL, where the
        current stream waits for the transfer; wall clock for the host-staged gloo rehearsal)."""
        if backend == "gloo":          # gloo p2p needs host tensors (rehearsal only)
            t0 = time.perf_counter()
            res_ = pkg.parallel.exchange_and_blend(out.float().cpu(), overlap, rank, world, dist, seg_counts).to(dev)
            state["exchange_s"] += time.perf_counter() - t0
        else:
            e0 = rt.event()
            res_ = pkg.parallel.exchange_and_blend(out, overlap, rank, world, dist, seg_counts)
            state.setdefault("exchange_events", []).append((e0, rt.event()))
        state["exchanges"] += 1
        return res_

    def run_batch(bi):
        """Batch bi of the rank's clip through P1..P7, as SeedVR2Pipeline.generate walks it
        [ref: src/core/generation_phases.py:342-521, 650-753, 904-1026, 1337]."""
        b = batches[bi]
        if bi == 0:
            state["write"] = 0
        if streamed:
            frames = pkg.video_io.frames_from_bytes(lr_u8.index_select(0, idx_dev[bi]), lay)
        else:
            frames = lr.index_select(0, idx_dev[bi]).contiguous()
        # transform, encode, noise drawn by the CPU generator and uploaded (inside the step), one-step DiT, decode
        dec = pipe.upscale_batch(frames, res, 42)
        sample = hip.crop_to_unit(dec[: b.ori_len].contiguous(), th, tw, to_unit=False)
        del dec
        write = state["write"]
        if bi > 0 and overlap and overlap < sample.shape[0] and write >= overlap:
            tail = final[write - overlap:write]
            hip.blend_overlap(tail.clone(), sample[:overlap].contiguous(), w_prev, tail)
            sample = sample[overlap:]
        n = sample.shape[0]
        final[write:write + n] = sample
        state["write"] = write + n
        state["frames"] += n
        if bi == nb - 1:                       # segment complete: [0,1] frames, then the boundary exchange
            out = hip.crop_to_unit(final[: state["write"]], th, tw, to_unit=True)
            if world > 1 and overlap:
                out = exchange(out)
                state["frames"] -= state["write"] - out.shape[0]   # frames handed to the left neighbour
            state["clips"] += 1
            return out
        return None

    def step(i):
        """weak: one temporal batch of the rank's clip (cycling through its batches); strong: the rank's whole segment of
        the shared clip, i.e. one finished clip per step over all ranks."""
        if strong:
            for bi in range(nb):
                run_batch(bi)
        else:
            run_batch(i % nb)

    def barrier():
        rt.sync()
        if world > 1:
            dist.barrier()
        rt.sync()

    if world > 1 and overlap:   # first use of a send/recv pair sets up its RCCL channel: not part of any step
        dummy = torch.zeros(overlap + 1, 8, 8, 3, dtype=torch.bfloat16, device=dev)
        pkg.parallel.exchange_and_blend(dummy.float().cpu() if backend == "gloo" else dummy, overlap, rank, world, dist)
    for i in range(args.warmup):
        t0 = time.time()
        step(i)
        rt.sync()
        what = f"segment of {clip} frames" if strong else f"batch of {len(batches[i % nb].index)} frames"
        log(f"[rank {rank}] warmup step {i} ({what}): {time.time() - t0:.2f}s")
    rt.reset_peak()
    state.update(frames=0, clips=0, exchange_s=0.0, exchanges=0)
    state.pop("exchange_events", None)
    ktimer.enabled = True
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    rt.sync()
    dt_own = time.perf_counter() - t0          # this rank's own time, before it waits for the slowest one
    barrier()
    dt = time.perf_counter() - t0
    ktimer.enabled = False
    frames_done = state["frames"]
    exch_s = state["exchange_s"] + sum(a.elapsed_time(b) for a, b in state.get("exchange_events", [])) * 1e-3
    ranks_seen, own_ms = 1, [dt_own / args.steps * 1e3]
    exch_ms = [exch_s * 1e3 / max(args.steps, 1)]
    if world > 1:
        ranks_seen = dist.get_world_size()
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ff = torch.tensor([float(frames_done)], device=cdev, dtype=torch.float64)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
        per = torch.zeros(world, 2, device=cdev, dtype=torch.float64)   # every rank's own ms per step and exchange ms
        per[rank, 0], per[rank, 1] = own_ms[0], exch_ms[0]
        dist.all_reduce(per, op=dist.ReduceOp.SUM)
        dt, frames_done = float(tt[0].item()), int(ff.item())
        own_ms, exch_ms = per[:, 0].tolist(), per[:, 1].tolist()
    peak_gb, peak_alloc_gb = rt.peak_gb()
    log(f"[rank {rank}] peak reserved {peak_gb:.1f} GB, peak allocated {peak_alloc_gb:.1f} GB")

    if rank == 0:
        nl, kt, kf = ktimer.summary()
        if os.environ.get("SVR2_BENCH_TABLE"):
            for line in ktimer.table():
                log("  conv3d", line)
        per_batch = {}
        for b in batches:
            T = len(b.index)
            enc_f, dec_f = _vae_flops(T, ph, pw)
            per_batch[str(T)] = {"vae_encode": enc_f / 1e12, "dit_gemm": _dit_flops(T, ph, pw, model) / 1e12, "vae_decode": dec_f / 1e12}
        achieved = kf / kt / 1e12 if kt > 0 else 0.0
        traffic, tsrc = None, None   # HBM-side bytes per conv launch from the committed rocprofv3 PMC passes of this workload
        for name in ("r03_hbm_traffic_cfg3.json", "r02_hbm_traffic_cfg3.json", "r01_hbm_traffic_cfg3.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if args.workload == "cfg3" and os.path.exists(tpath):
                try:
                    with open(tpath) as fh:
                        traffic = json.load(fh)["kernels"]["conv3d halo-resident (conv_halo*_kernel)"]["hbm_bytes_per_launch"]
                    tsrc = name
                    break
                except Exception:
                    traffic = None
        metric = "upscaled frames/sec + peak HBM GB, SeedVR2-3B 720p->4K, 1/2/4/8 MI355X"
        if strong:
            step_desc = ("one whole clip: every rank walks the batches of its segment of the SAME clip (reference split, "
                         "inference_cli.py:1082-1097), then the boundary exchange")
        else:
            step_desc = "one temporal batch of the clip (noise generation + upload inside), cycling through the clip's batches"
        result = {
            "metric": metric,
            "value": frames_done / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "peak_hbm_gb": peak_gb,
            "ranks_seen": ranks_seen, "launcher": "self" if os.environ.get("SVR2_BENCH_SELF_LAUNCHED") else
                                                  ("torchrun/env" if world > 1 else "single"),
            "backend": backend if world > 1 else None,
            "rank_ms_per_step": {"min": min(own_ms), "max": max(own_ms)},
            "exchange_ms": {"per_step_max": max(exch_ms), "per_step_min": min(exch_ms),
                            "exchanges_rank0": state["exchanges"]} if world > 1 else None,
            "config": {"workload": f"{args.workload}: {desc}", "clip_frames_per_rank": clip, "clip_frames_total": total,
                       "segment_frames_by_rank": seg_counts, "batch_size": bsz,
                       "temporal_overlap": overlap, "frames_per_batch": [len(b.index) for b in batches],
                       "finished_frames": frames_done, "clips_completed_rank0": state["clips"],
                       "lr": [hin, win], "out": [th, tw], "padded": [ph, pw], "parallelism": f"frame-chunk x{world}",
                       "step": step_desc,
                       "algorithmic_tflop_per_batch_by_frames": per_batch},
            "roofline": {"bound": "mfma", "kernel": "conv_halo_wr_kernel<16,256|32,128> + conv_halo_kernel<64,16> (halo-resident causal conv3d: every 3x3 stride-1 conv of the VAE)",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "traffic_unit": f"bytes per launch (FETCH_SIZE*2 + WRITE_SIZE, rocprofv3 PMC, profiles/{tsrc})" if tsrc else None,
                         "algorithmic_flop_per_launch": kf / nl if nl else None,
                         "launches": nl, "kernel_time_s": kt, "share_of_step_time": kt / dt if dt > 0 else None},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args.workload, min(os.cpu_count() or 1, 64), pkg, vsd, dsd)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
