#!/usr/bin/env python3
"""Child process of tests/test_cpu_cli_fanout.py: the product CLI's `main` with a CPU Runtime (gloo, device 'cpu').

The only substitutions are the things that need a GPU: the upscaler (a per-frame function that depends on the rank, so a
mixed-up segment shows) and the two device frame conversions of video_io (served by the oracle's restatement of the
reference's casts).  Everything else -- argument parsing, rank / device selection, process group, segment split,
chunk streaming, neighbour exchange, blend, gather, writer -- is the product's code."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import __graft_entry__ as ge


def fake_upscale(frames: torch.Tensor, rank: int) -> torch.Tensor:
    """bf16 [T,H,W,3] in [0,1] -> bf16, same count: x/2 + rank/10 on a 2x nearest-upsampled frame."""
    x = frames.float().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    return (x * 0.5 + 0.1 * rank).to(torch.bfloat16)


def main():
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("SVR2_TEST_FAIL_RANK") == str(rank):
        print(f"rank {rank}: simulated worker failure", file=sys.stderr)
        return 7
    pkg = ge.load_package()
    from seedvr2_ref import video_io as ovio
    vio = pkg.video_io
    cli = pkg.inference_cli if hasattr(pkg, "inference_cli") else __import__("importlib").import_module("seedvr2_amd.inference_cli")

    def from_bytes(raw, lay):
        assert lay.kind == "rgb" and lay.flags == 0
        a = raw.numpy().reshape(raw.shape[0], lay.height, lay.row_stride)[:, :, : lay.width * 3]
        return ovio.u8_to_unit(np.ascontiguousarray(a).reshape(raw.shape[0], lay.height, lay.width, 3))

    def to_bytes(frames, lay, out=None):
        assert lay.kind == "rgb" and lay.flags == 0 and lay.row_stride == lay.width * 3
        return torch.from_numpy(ovio.unit_to_u8(frames)).reshape(frames.shape[0], -1)

    vio.frames_from_bytes, vio.frames_to_bytes = from_bytes, to_bytes
    seen = {}

    def run_device(image, dit, vae, *a, **k):
        seen["device"] = dit["device"]
        return fake_upscale(image, rank)

    pkg.nodes.SeedVR2VideoUpscaler.run_device = staticmethod(run_device)
    pkg.weights.load_text_embedding = lambda *a, **k: torch.zeros(58, 8)
    rt = cli.Runtime(device_for=lambda dev_id: "cpu", set_device=lambda dev: None, dist_backend="gloo")
    rc = cli.main(sys.argv[1:], runtime=rt)
    return rc


if __name__ == "__main__":
    sys.exit(main())
