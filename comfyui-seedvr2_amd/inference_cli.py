#!/usr/bin/env python3
"""Standalone CLI with the reference's flags [ref: inference_cli.py:1216-1401] on the MI355X path.

Offload / BlockSwap / tiling / torch.compile flags are accepted for drop-in compatibility and validated
(they must stay at their "off" values: everything is resident in 288 GB of HBM).  Frame I/O: .png/.jpg via
PIL and .pt/.npy tensors ([N,H,W,3] in [0,1]); video containers need cv2, which this image lacks.
Multi-GPU: ``--cuda_device 0,1,...`` runs one process per GPU (torch.distributed / RCCL) with the
reference's segment split and boundary blend (parallel.py)."""

from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(description="SeedVR2 video upscaler (MI355X-native hot path)")
    p.add_argument("input", type=str)
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--output_format", type=str, default=None, choices=["mp4", "png", None])
    p.add_argument("--model_dir", type=str, default=None)
    p.add_argument("--dit_model", type=str, default="seedvr2_ema_3b_fp16.safetensors")
    p.add_argument("--resolution", type=int, default=1080)
    p.add_argument("--max_resolution", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=5)
    p.add_argument("--uniform_batch_size", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--skip_first_frames", type=int, default=0)
    p.add_argument("--load_cap", type=int, default=0)
    p.add_argument("--chunk_size", type=int, default=0)
    p.add_argument("--prepend_frames", type=int, default=0)
    p.add_argument("--temporal_overlap", type=int, default=0)
    p.add_argument("--color_correction", type=str, default="lab",
                   choices=["lab", "wavelet", "wavelet_adaptive", "hsv", "adain", "none"])
    p.add_argument("--input_noise_scale", type=float, default=0.0)
    p.add_argument("--latent_noise_scale", type=float, default=0.0)
    p.add_argument("--cuda_device", type=str, default=None)
    p.add_argument("--dit_offload_device", type=str, default="none")
    p.add_argument("--vae_offload_device", type=str, default="none")
    p.add_argument("--tensor_offload_device", type=str, default="cpu")
    p.add_argument("--blocks_to_swap", type=int, default=0)
    p.add_argument("--swap_io_components", action="store_true")
    p.add_argument("--vae_encode_tiled", action="store_true")
    p.add_argument("--vae_encode_tile_size", type=int, default=1024)
    p.add_argument("--vae_encode_tile_overlap", type=int, default=128)
    p.add_argument("--vae_decode_tiled", action="store_true")
    p.add_argument("--vae_decode_tile_size", type=int, default=1024)
    p.add_argument("--vae_decode_tile_overlap", type=int, default=128)
    p.add_argument("--tile_debug", type=str, default="false", choices=["false", "encode", "decode"])
    p.add_argument("--attention_mode", type=str, default="sdpa", choices=["sdpa", "flash_attn"])
    p.add_argument("--compile_dit", action="store_true")
    p.add_argument("--compile_vae", action="store_true")
    p.add_argument("--compile_backend", type=str, default="inductor", choices=["inductor", "cudagraphs"])
    p.add_argument("--compile_mode", type=str, default="default",
                   choices=["default", "reduce-overhead", "max-autotune", "max-autotune-no-cudagraphs"])
    p.add_argument("--compile_fullgraph", action="store_true")
    p.add_argument("--compile_dynamic", action="store_true")
    p.add_argument("--compile_dynamo_cache_size_limit", type=int, default=64)
    p.add_argument("--compile_dynamo_recompile_limit", type=int, default=128)
    p.add_argument("--cache_dit", action="store_true")
    p.add_argument("--cache_vae", action="store_true")
    p.add_argument("--debug", action="store_true")
    a = p.parse_args(argv)
    if a.blocks_to_swap or a.swap_io_components:
        p.error("BlockSwap is unnecessary on 288 GB HBM: leave --blocks_to_swap 0")
    if a.vae_encode_tiled or a.vae_decode_tiled:
        p.error("VAE tiling is not implemented on the MI355X path (untiled fits in HBM)")
    if a.batch_size % 4 != 1:
        p.error("--batch_size must follow 4n+1")
    return a


def load_frames(path: str) -> torch.Tensor:
    """-> [N,H,W,3] float32 in [0,1] (uint8 / 255 like the reference, inference_cli.py:231-250,543)."""
    ext = os.path.splitext(path)[1].lower()
    if ext in (".png", ".jpg", ".jpeg", ".bmp", ".webp"):
        from PIL import Image
        return torch.from_numpy(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8).copy()).float().div(255)[None]
    if ext == ".npy":
        return torch.from_numpy(np.load(path, allow_pickle=False)).float()
    if ext == ".pt":
        return torch.load(path, weights_only=True).float()
    raise SystemExit(f"unsupported input {ext!r}: video containers need cv2, which this image does not have")


def save_frames(frames: torch.Tensor, path: str):
    arr = (frames.clamp(0, 1) * 255.0).to(torch.uint8).cpu().numpy()  # truncation like the reference (:520)
    if path.endswith(".npy"):
        np.save(path, arr)
    elif arr.shape[0] == 1:
        from PIL import Image
        Image.fromarray(arr[0]).save(path)
    else:
        os.makedirs(path, exist_ok=True)
        from PIL import Image
        for i, f in enumerate(arr):
            Image.fromarray(f).save(os.path.join(path, f"{i:06d}.png"))


def main(argv=None) -> int:
    a = parse_arguments(argv)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    t0 = time.time()
    try:
        frames = load_frames(a.input)
        if a.skip_first_frames:
            frames = frames[a.skip_first_frames:]
        if a.load_cap:
            frames = frames[:a.load_cap]
        devs = [d for d in (a.cuda_device or "0").split(",") if d != ""]
        world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
        if len(devs) > 1 and world == 1:
            raise SystemExit(f"multi-GPU: launch one process per GPU, `python -m torch.distributed.run --nproc-per-node "
                             f"{len(devs)} --master-addr 127.0.0.1 inference_cli.py ...` (frame segments + one RCCL "
                             "neighbour exchange, parallel.py / INTEGRATION.md)")
        dev = f"cuda:{devs[local % len(devs)] if world > 1 else devs[0]}"
        dist = None
        if world > 1:   # the reference's multi-GPU mode: one segment per rank [ref: inference_cli.py:1048-1209]
            import torch.distributed as dist
            torch.cuda.set_device(torch.device(dev))
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        dit_cfg, = pkg.nodes.SeedVR2LoadDiTModel.execute(model=a.dit_model, device=dev)
        vae_cfg, = pkg.nodes.SeedVR2LoadVAEModel.execute(device=dev)
        txt = pkg.weights.find_model_file("pos_emb.pt", a.model_dir) or os.path.join(here, "pos_emb.pt")
        text_pos = torch.load(txt, weights_only=True)

        def run(seg, prepend):
            return pkg.nodes.SeedVR2VideoUpscaler.execute(
                seg, dit_cfg, vae_cfg, a.seed, a.resolution, a.max_resolution, a.batch_size, a.uniform_batch_size,
                a.temporal_overlap, prepend, a.color_correction, a.input_noise_scale, a.latent_noise_scale,
                text_pos=text_pos)[0]

        if world > 1:   # rank 0 alone applies prepend_frames [ref: inference_cli.py:981-984]
            out = pkg.parallel.upscale_distributed(
                lambda seg: run(seg, a.prepend_frames if rank == 0 else 0).to(dev), frames, a.temporal_overlap, rank,
                world, dist, a.batch_size)
            dist.barrier()
            dist.destroy_process_group()
            if rank != 0:
                return 0
            out = out.cpu()
        else:
            out = run(frames, a.prepend_frames)
        outp = a.output or (os.path.splitext(a.input)[0] + "_upscaled" + (".png" if out.shape[0] == 1 else ""))
        save_frames(out, outp)
        dt = time.time() - t0
        print(f"Average FPS: {out.shape[0] / dt:.2f} frames/sec ({out.shape[0]} frames in {dt:.2f}s)")
        return 0
    except Exception as e:  # exit code 1 on error [ref: inference_cli.py:1625-1629]
        print(f"Error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
