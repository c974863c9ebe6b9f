"""Child of tests/test_gpu_parallel.py: one rank over the RCCL backend ("nccl" on ROCm) on cuda:0 - process-group
initialisation with a device id, barrier, the two all-reduces bench.py ends with, and the frame-exchange entry points at
world size 1.  One-GPU boxes cannot run a second RCCL rank; this checks that the library loads and the calls the N > 1 path
makes are accepted on this image."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.25, 0.0], device=dev, dtype=torch.float64)
dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
f = torch.tensor([64.0], device=dev, dtype=torch.float64)
dist.all_reduce(f, op=dist.ReduceOp.SUM)
counts = torch.zeros(1, dtype=torch.int64, device=dev)
counts[0] = 7
dist.all_reduce(counts)
frames = torch.rand(7, 8, 8, 3, device=dev).to(torch.bfloat16)
out = pkg.parallel.exchange_and_blend(frames, 2, 0, 1, dist)
gathered = pkg.parallel.gather_frames(out, 0, 1, 0, dist)
torch.cuda.synchronize()
ok = float(t[0]) == 1.25 and float(f[0]) == 64.0 and int(counts[0]) == 7 and torch.equal(out, frames) and torch.equal(gathered, frames)
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK" if ok else "RCCL_SINGLE_RANK_MISMATCH")
