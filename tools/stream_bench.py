#!/usr/bin/env python3
"""End-to-end streaming throughput of the CLI path at 4K (GPU box only): y4m file -> reader thread -> device ->
upscale (NaDiT-3B + VAE, synthetic weights) -> device quantisation -> writer thread -> y4m file, against the same
upscale fed from device-resident frames.  Shows what the frame I/O of SURVEY 8f next-3 costs on top of the models.

usage: python tools/stream_bench.py [frames=55] [chunk=38] [batch=21] [overlap=4] [out=/tmp/svr2_out.y4m]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
vio = pkg.video_io
dev = torch.device("cuda:0")
pkg.hip.device_check(0)
N, CHUNK, BATCH, OV = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 55), (2, 38), (3, 21), (4, 4)))
OUT = sys.argv[5] if len(sys.argv) > 5 else "/tmp/svr2_out.y4m"
H, W, RES = 720, 1280, 2160

vcfg, dcfg = pkg.vae.VAEConfig(), pkg.dit.CFG_3B
vae = pkg.vae.VideoVAEHip(vcfg, pkg.vae.synthetic_state_dict(vcfg, dev, seed=1), dev)
dit = pkg.dit.NaDiTHip(dcfg, pkg.dit.synthetic_state_dict(dcfg, dev, seed=2), dev)
txt = (torch.randn(58, 5120, generator=torch.Generator().manual_seed(3)) * 0.5)
pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, txt, dev)

# a 720p 4:2:0 clip on disk (smooth content, like bench.py)
g = torch.Generator().manual_seed(7)
low = 0.5 + 0.2 * torch.randn(N, 3, H // 16, W // 16, generator=g)
rgb = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic").clamp(0, 1)
u8 = (rgb * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
src_path = "/tmp/svr2_in.y4m"
sink = vio.DeviceSink(vio.open_writer(src_path, W, H, 24.0), dev)       # the product's own writer makes the input file
for i in range(0, N, 16):
    sink.write(vio.frames_from_bytes(u8[i:i + 16].reshape(-1, H * W * 3).to(dev), vio.Layout("rgb", W, H)))
sink.close()
del rgb, low


def run(frames, prepend):
    return pkg.nodes.SeedVR2VideoUpscaler.run_device(frames, {}, {}, 42, RES, 0, BATCH, False, OV, prepend, "lab", pipeline=pipe)


def stream():
    reader = vio.open_reader(src_path)
    source = vio.DeviceSource(reader, dev)
    t0 = time.time()
    n = vio.stream_upscale(source, lambda ww, hh: vio.DeviceSink(vio.open_writer(OUT, ww, hh, reader.fps), dev), run,
                           N, CHUNK, OV, 0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    source.close()
    return n, dt


def resident():
    reader = vio.open_reader(src_path)
    source = vio.DeviceSource(reader, dev)
    chunks, read = [], 0
    while read < N:
        c = source.next_chunk(min(CHUNK, N - read))
        chunks.append(c)
        read += c.shape[0]
    source.close()
    torch.cuda.synchronize()
    t0 = time.time()
    tail, n = None, 0
    for c in chunks:
        x = torch.cat((tail, c)) if tail is not None else c
        out = run(x, 0)
        n += out.shape[0] - (tail.shape[0] if tail is not None else 0)
        tail = c[-OV:].clone() if OV else None
        del out
    torch.cuda.synchronize()
    return n, time.time() - t0


stream()                                   # warm-up (allocator, pinned buffers, page cache)
n1, t1 = stream()
n2, t2 = resident()
sz = os.path.getsize(OUT) / 1e9
print(f"streamed  : {n1} frames 720p->4K in {t1:.2f} s = {n1 / t1:.2f} frames/s (y4m in, {sz:.2f} GB y4m out, lab colour correction)")
print(f"resident  : {n2} frames in {t2:.2f} s = {n2 / t2:.2f} frames/s (same chunks, frames already on the device, nothing written)")
print(f"I/O cost  : {(t1 - t2) / t1 * 100:.1f} % of the streamed wall time")
