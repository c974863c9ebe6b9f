#!/usr/bin/env python3
"""Standalone CLI with the reference's flags [ref: inference_cli.py:1216-1401] on the MI355X path.

Offload / BlockSwap / torch.compile flags are accepted for drop-in compatibility and validated (they must stay at
their "off" values: everything is resident in 288 GB of HBM); the --vae_*_tiled flags run the reference's spatial tiling
(never needed for memory here, kept for output parity with tiled reference runs).  Frame I/O streams through
video_io.py: .y4m, uncompressed/MJPEG .avi, image files, .npy/.pt tensors natively, other video
containers through cv2 or an ffmpeg binary when present; the output container follows the extension of --output
(--output_format mp4|png|y4m|avi|mjpeg|npy names it when --output has none; mjpeg = Motion-JPEG AVI encoded on the device).  --chunk_size streams the clip in chunks with
--temporal_overlap raw context frames, like the reference.
Multi-GPU: ``--cuda_device 0,1,...`` fans out by itself like the reference [ref: inference_cli.py:1048-1209]: the parent
starts one fresh child process per listed GPU BEFORE touching any GPU (same CLI, RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_* set), the children form a torch.distributed group (RCCL over xGMI) and run the reference's segment split and
boundary blend on the devices (parallel.py); a child that dies takes the others down instead of leaving them waiting.
Launching the ranks with torchrun works too (WORLD_SIZE already set -> no fan-out)."""

from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional

import numpy as np
import torch


class Runtime:
    """What `main` needs from the machine: the defaults are the product (one HIP device per rank, RCCL); tests substitute
    a CPU flavour to drive the fan-out and the multi-rank control flow over gloo."""

    def __init__(self, device_for: Optional[Callable[[str], str]] = None, set_device: Optional[Callable[[str], None]] = None,
                 dist_backend: str = "nccl", child_cmd: Optional[List[str]] = None, poll_seconds: float = 0.2):
        self.device_for = device_for or (lambda dev_id: f"cuda:{dev_id}")
        self.set_device = set_device or (lambda dev: torch.cuda.set_device(torch.device(dev)))
        # SVR2_DIST_BACKEND=gloo: rehearsal of the multi-rank path where RCCL cannot run (several ranks on ONE GPU)
        self.dist_backend = os.environ.get("SVR2_DIST_BACKEND", dist_backend)
        self.child_cmd = child_cmd or [sys.executable, os.path.abspath(__file__)]
        self.poll_seconds = poll_seconds


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(description="SeedVR2 video upscaler (MI355X-native hot path)")
    p.add_argument("input", type=str)
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--output_format", type=str, default=None, choices=["mp4", "png", "y4m", "avi", "mjpeg", "npy", None])
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
    if a.tile_debug != "false":
        p.error("--tile_debug overlays are a debugging aid of the reference, not built here")
    if a.batch_size < 1:
        p.error("--batch_size must be positive")
    # any batch size is accepted: every batch is padded to 4n+1 with reversed tail frames like the reference
    # [ref: src/core/generation_phases.py:109-124, 386-399]
    return a


def get_input_type(path: str, vio) -> str:
    """'video' | 'image' | 'directory' | 'unknown' [ref: inference_cli.py:281-306]; tensor files count as video."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Input path not found: {path}")
    if os.path.isdir(path):
        return "directory"
    ext = os.path.splitext(path)[1].lower()
    if ext in vio.VIDEO_EXTENSIONS or ext in vio.TENSOR_EXTENSIONS:
        return "video"
    return "image" if ext in vio.IMAGE_EXTENSIONS else "unknown"


def generate_output_path(input_path: str, output_format: str, output_dir=None, input_type=None, from_directory=False) -> str:
    """[ref: inference_cli.py:309-353]: `<stem>_upscaled.<fmt>` next to the input, `<dir>_upscaled/<stem>.<fmt>` in batch
    mode, `<output_dir>/<stem>.<fmt>` when a directory is given; png sequences are directories."""
    stem = os.path.splitext(os.path.basename(input_path))[0]
    parent = os.path.dirname(os.path.abspath(input_path))
    if output_dir:
        base, suffix = output_dir, ""
    elif from_directory:
        base, suffix = os.path.join(os.path.dirname(parent), os.path.basename(parent) + "_upscaled"), ""
    else:
        base, suffix = parent, "_upscaled"
    if output_format == "png":
        name = f"{stem}{suffix}.png" if input_type == "image" else f"{stem}{suffix}"
    else:
        name = f"{stem}{suffix}.{output_format}"
    return os.path.abspath(os.path.join(base, name))


class _CollectSink:
    """Stand-in for DeviceSink that keeps the finished frames on the device (multi-rank: exchanged before writing)."""

    def __init__(self):
        self.parts, self.frames_written = [], 0

    def write(self, frames):
        self.parts.append(frames)
        self.frames_written += frames.shape[0]

    def close(self):
        pass


def process_single_file(input_path: str, a, pkg, dev: str, dit_cfg, vae_cfg, text_pos, output_path=None,
                        dist=None, rank: int = 0, world: int = 1) -> int:
    """One video / image / tensor file -> upscaled output; returns frames written [ref: inference_cli.py:356-523].
    Frames stay on the device between the reader and the writer (video_io.py)."""
    vio = pkg.video_io
    itype = get_input_type(input_path, vio)
    if itype == "unknown":
        print(f"Skipping unsupported file: {input_path}", file=sys.stderr)
        return 0
    fmt = a.output_format or ("mp4" if itype == "video" else "png")
    mjpeg = 90 if fmt == "mjpeg" else 0          # Motion-JPEG in AVI, encoded on the device (video_io.MJPEGWriter)
    ext = "avi" if mjpeg else fmt
    if output_path is None:
        output_path = generate_output_path(input_path, ext, input_type=itype)
    elif not os.path.splitext(output_path)[1] or (fmt == "png" and itype != "image"):
        output_path = generate_output_path(input_path, ext, output_dir=output_path, input_type=itype)
    reader = vio.open_reader(input_path)
    total = reader.total_frames
    todo = total - a.skip_first_frames if itype == "video" else total
    if itype == "video" and a.load_cap > 0:
        todo = min(todo, a.load_cap)
    if todo <= 0:
        print(f"No frames to process after skipping {a.skip_first_frames} of {total}", file=sys.stderr)
        reader.close()
        return 0
    first = a.skip_first_frames if itype == "video" else 0
    base_name = os.path.splitext(os.path.basename(input_path))[0]

    def run(frames, prepend):
        return pkg.nodes.SeedVR2VideoUpscaler.run_device(
            frames, dit_cfg, vae_cfg, a.seed, a.resolution, a.max_resolution, a.batch_size, a.uniform_batch_size,
            a.temporal_overlap, prepend, a.color_correction, text_pos=text_pos, input_noise_scale=a.input_noise_scale,
            latent_noise_scale=a.latent_noise_scale,
            # one GPU: the reference never strips the warm-up frames' outputs [ref: inference_cli.py:920]; several GPUs:
            # its parent strips them after concatenation [ref: :1200-1204] = rank 0 strips them here
            keep_prepended=(world == 1))

    def sink_for(w, h):
        return vio.DeviceSink(vio.open_writer(output_path, w, h, reader.fps, base_name, single_image=itype == "image",
                                              alpha=reader.layout.channels == 4, mjpeg_quality=mjpeg), dev)

    log = (lambda m: print(m, flush=True)) if rank == 0 else None
    if world == 1:
        reader.seek(first)
        src = vio.DeviceSource(reader, dev)
        try:
            n = vio.stream_upscale(src, sink_for, run, todo, a.chunk_size, a.temporal_overlap, a.prepend_frames, log)
        finally:
            src.close()
        print(f"Output saved to: {output_path}")
        return n
    # multi-GPU: every rank streams its own segment [ref: inference_cli.py:452-471, 1048-1209], boundaries are cross-faded
    # through the neighbour exchange, rank 0 writes
    ranges = pkg.parallel.segment_ranges(todo, world, a.temporal_overlap, a.batch_size, True)
    s, e = ranges[rank]
    counts = [hi - lo for lo, hi in ranges]   # finished frames per rank: local arithmetic, no count exchange
    reader.seek(first + s)
    src, coll = vio.DeviceSource(reader, dev), _CollectSink()
    try:
        vio.stream_upscale(src, lambda w, h: coll, run, e - s, a.chunk_size, a.temporal_overlap,
                           a.prepend_frames if rank == 0 else 0, None)
    finally:
        src.close()
    mine = torch.cat(coll.parts, dim=0) if coll.parts else torch.empty(0, 1, 1, 3, dtype=torch.bfloat16, device=dev)
    owned = pkg.parallel.exchange_and_blend(mine, a.temporal_overlap, rank, world, dist, counts)
    full = pkg.parallel.gather_frames(owned, rank, world, 0, dist, pkg.parallel.owned_counts(counts, a.temporal_overlap))
    if rank != 0:
        return 0
    sink = sink_for(full.shape[2], full.shape[1])
    sink.write(full)
    sink.close()
    print(f"Output saved to: {output_path}")
    return full.shape[0]


def fan_out(devs: List[str], argv: List[str], rt: Runtime) -> int:
    """The reference's multi-GPU entry [ref: inference_cli.py:1048-1209]: one fresh worker process per listed device,
    started by a parent that has not touched (and never touches) a GPU.  Unlike the reference's workers, which return
    their frames to the parent through host shared memory, the children exchange segment boundaries among themselves on
    the devices and rank 0 writes the output.  Returns 0 when every child did; the first failing child takes the others
    down (the reference's parent would wait forever on its queue)."""
    procs, failed = [], 0

    def stop_children(*_):
        for p in procs:                          # exactly the processes started below
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_term(signum, frame):
        stop_children()
        sys.exit(128 + signum)

    old_term = None
    try:
        try:
            old_term = signal.signal(signal.SIGTERM, on_term)
        except ValueError:                       # not the main thread (tests): the finally clause still cleans up
            old_term = None
        port = free_port()
        for i in range(len(devs)):
            env = dict(os.environ, RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(len(devs)), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # the ranks share the host cores (CPU-generator noise, reader / writer threads) instead of oversubscribing them
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or len(devs)) // len(devs))))
            procs.append(subprocess.Popen(rt.child_cmd + list(argv), env=env))
        while any(p.poll() is None for p in procs):
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:
                failed = bad[0].returncode
                break
            time.sleep(rt.poll_seconds)
    finally:
        # also on KeyboardInterrupt / a Popen that raised for a later rank: no worker is left blocked in the rendezvous
        stop_children()
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    codes = [p.wait() for p in procs]
    if failed or any(codes):
        print(f"Error: multi-GPU worker exit codes {codes}", file=sys.stderr)
        return 1
    return 0


def free_port() -> int:
    """A rendezvous port for one run.  Picked by bind-and-close, so another process could take it before rank 0 binds it:
    the children then fail their rendezvous and the run ends non-zero (never retried inside a process that touched a GPU)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main(argv=None, runtime: Runtime = None) -> int:
    rt = runtime or Runtime()
    a = parse_arguments(argv)
    devs = [d for d in (a.cuda_device or "0").split(",") if d != ""]
    if len(devs) > 1 and "WORLD_SIZE" not in os.environ:
        # before the package (and with it the HIP library) is loaded: the parent never initialises a GPU
        return fan_out(devs, list(sys.argv[1:] if argv is None else argv), rt)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    vio = pkg.video_io
    t0 = time.time()
    dist = None
    try:
        world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
        dev = rt.device_for(devs[local % len(devs)] if world > 1 else devs[0])
        rt.set_device(dev)   # also for one rank: --cuda_device 3 must run on GPU 3's stream, not on GPU 0's
        if world > 1:   # the reference's multi-GPU mode: one segment per rank [ref: inference_cli.py:1048-1209]
            import torch.distributed as dist
            if rt.dist_backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device(dev))
            else:
                dist.init_process_group(backend=rt.dist_backend)
        if a.model_dir:     # --model_dir is where the DiT, VAE and text-embedding files live [ref: inference_cli.py:835-844]
            os.environ["SEEDVR2_MODEL_DIR"] = a.model_dir
        dit_cfg, = pkg.nodes.SeedVR2LoadDiTModel.execute(model=a.dit_model, device=dev)
        vae_cfg, = pkg.nodes.SeedVR2LoadVAEModel.execute(
            device=dev, encode_tiled=a.vae_encode_tiled, encode_tile_size=a.vae_encode_tile_size,
            encode_tile_overlap=a.vae_encode_tile_overlap, decode_tiled=a.vae_decode_tiled,
            decode_tile_size=a.vae_decode_tile_size, decode_tile_overlap=a.vae_decode_tile_overlap)
        text_pos = pkg.weights.load_text_embedding("pos_emb.pt", a.model_dir)
        itype = get_input_type(a.input, vio)
        nframes = 0
        if itype == "directory":   # batch mode: every media file on its own [ref: inference_cli.py:1525-1575]
            files = vio.list_media_files(a.input)
            if not files:
                raise SystemExit(f"No video or image files found in directory: {a.input}")
            for i, fp in enumerate(files, 1):
                if rank == 0:
                    print(f"Processing file {i}/{len(files)}: {os.path.basename(fp)}", flush=True)
                ftype = get_input_type(fp, vio)
                fmt = a.output_format or ("mp4" if ftype == "video" else "png")
                outp = generate_output_path(fp, "avi" if fmt == "mjpeg" else fmt, a.output, input_type=ftype, from_directory=True)
                keep, a.output_format = a.output_format, fmt
                nframes += process_single_file(fp, a, pkg, dev, dit_cfg, vae_cfg, text_pos, outp, dist, rank, world)
                a.output_format = keep
        elif itype in ("video", "image"):
            nframes = process_single_file(a.input, a, pkg, dev, dit_cfg, vae_cfg, text_pos, a.output, dist, rank, world)
        else:
            raise SystemExit(f"Unsupported input type: {a.input}")
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        dt = time.time() - t0
        if rank == 0 and nframes:
            print(f"Average FPS: {nframes / dt:.2f} frames/sec ({nframes} frames in {dt:.2f}s)")
        return 0
    except (Exception, SystemExit) as e:  # exit code 1 on error [ref: inference_cli.py:1625-1629]
        print(f"Error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
