#!/usr/bin/env python3
"""Child process of tests/test_cpu_bench.py: the product `bench.main` with a CPU Runtime (gloo, device 'cpu').

The only substitutions are the things that need a GPU: the models (a per-frame stand-in that depends on the rank) and the
three device ops the bench's step calls directly.  Everything else - argument parsing, the self-launch of N ranks, rank /
world handling, the process group, the segment split of --scaling strong, batch walking, the neighbour exchange, the
timing protocol and the JSON line - is the bench's own code."""

import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

import bench
import __graft_entry__ as ge


class CpuRuntime(bench.Runtime):
    def __init__(self):
        super().__init__()
        self.child_cmd = [sys.executable, os.path.abspath(__file__)]
        self.backend = "gloo"

    def device(self, local, rank):
        self.rank = rank
        return torch.device("cpu")

    def package(self, dev):
        pkg = ge.load_package()

        def crop_to_unit(x, th, tw, to_unit=True):
            y = x[:, :th, :tw]
            return (y.float().clamp(-1, 1) * 0.5 + 0.5).to(x.dtype) if to_unit else y.contiguous()

        def blend_overlap(prev, cur, w, out):
            wv = w.float().view(-1, 1, 1, 1)
            out.copy_((prev.float() * wv + cur.float() * (1 - wv)).to(out.dtype))

        pkg.hip.crop_to_unit, pkg.hip.blend_overlap = crop_to_unit, blend_overlap
        return pkg

    def build(self, pkg, model, dev):
        rank = self.rank
        if os.environ.get("SVR2_TEST_FAIL_RANK") == str(rank):
            print(f"rank {rank}: simulated failure", file=sys.stderr)
            sys.exit(7)

        def upscale_batch(frames, res, seed):
            h, w = frames.shape[1:3]
            _, _, th, tw, ph, pw = pkg.pipeline.target_dims(h, w, res)
            x = frames.float().permute(0, 3, 1, 2)
            y = torch.nn.functional.interpolate(x, size=(th, tw), mode="nearest")
            y = torch.nn.functional.pad(y, (0, pw - tw, 0, ph - th)).permute(0, 2, 3, 1)
            return ((y * 2 - 1) * 0.5 + 0.05 * rank).to(torch.bfloat16)

        return types.SimpleNamespace(upscale_batch=upscale_batch), None, None

    def sync(self):
        pass

    def reset_peak(self):
        pass

    def peak_gb(self):
        return 0.0, 0.0

    def allocated_gib(self):
        return 0.0

    def event(self):
        raise RuntimeError("no device events on the CPU runtime")


if __name__ == "__main__":
    bench.main(sys.argv[1:], CpuRuntime())
