"""Frame-chunk sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path shards along the frame axis only: every rank runs the whole encode -> DiT -> decode pipeline on
its own contiguous segment; the single exchange is the `temporal_overlap` boundary: rank r+1 sends its
first `overlap` finished frames to rank r (one point-to-point message on the direct xGMI link), rank r
cross-fades them into its tail and rank r+1 drops them - the device-side equivalent of the reference's
multi-GPU mode, which ships every worker's whole output through host shared memory and blends on the CPU
[ref: inference_cli.py:1048-1209 (_gpu_processing): segment split :1082-1097 / :1118-1133,
 boundary blend :1163-1198, blend function src/core/generation_utils.py:284-312].
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def segment_ranges(total: int, world: int, overlap: int, batch_size: int = 1, streaming: bool = True,
                   start: int = 0) -> List[Tuple[int, int]]:
    """[start, end) of every rank's segment, overlap frames included for all but the last rank.
    streaming=True: base = total // world, first total % world ranks get one extra, +overlap
    [ref: inference_cli.py:1082-1097]; streaming=False: the pre-loaded-frames split, chunk rounded up to a
    multiple of batch_size [ref: inference_cli.py:1118-1133]."""
    if world <= 1:
        return [(start, start + total)]
    out = []
    if streaming:
        base, rem = divmod(total, world)
        cur = start
        for r in range(world):
            n = base + (1 if r < rem else 0)
            end = cur + n
            if r < world - 1 and overlap > 0:
                end = min(end + overlap, start + total)
            out.append((cur, end))
            cur += n
        return out
    if overlap > 0:
        with_ov = total // world + overlap
        if batch_size > 1:
            with_ov = ((with_ov + batch_size - 1) // batch_size) * batch_size
        base = with_ov - overlap
        for r in range(world):
            s = r * base
            e = total if r == world - 1 else min(s + with_ov, total)
            out.append((start + s, start + e))
        return out
    sizes = [len(c) for c in torch.arange(total).chunk(world)]  # torch.chunk semantics (may yield fewer chunks)
    cur = start
    for n in sizes:
        out.append((cur, cur + n))
        cur += n
    return out + [(cur, cur)] * (world - len(sizes))


def blend_weights_f32(overlap: int, device=None) -> torch.Tensor:
    """w_prev of blend_overlapping_frames in fp32 (segment boundaries are blended on fp32 [0,1] frames,
    [ref: inference_cli.py:1167-1188])."""
    if overlap >= 3:
        t = torch.linspace(0.0, 1.0, steps=overlap, dtype=torch.float32, device=device)
        u = ((t - 1.0 / 3.0) / (2.0 / 3.0 - 1.0 / 3.0)).clamp(0.0, 1.0)
        return 0.5 + 0.5 * torch.cos(torch.pi * u)
    return torch.linspace(1.0, 0.0, steps=overlap, dtype=torch.float32, device=device)


def _host_staged(frames: torch.Tensor, dist) -> bool:
    """gloo moves host memory only: device frames are staged through the host (rehearsals of the multi-rank control
    flow, e.g. several ranks on one GPU); RCCL ("nccl") sends device memory over xGMI as it is."""
    try:
        return frames.is_cuda and dist.get_backend() == "gloo"
    except Exception:
        return False


def owned_counts(counts: Sequence[int], overlap: int) -> List[int]:
    """Frames every rank owns AFTER the boundary exchange, from the finished-frame counts before it: later ranks
    contribute chunk[overlap:] (nothing when chunk.shape[0] <= overlap) [ref: inference_cli.py:1163-1198]."""
    if overlap <= 0:
        return list(counts)
    return [c if r == 0 else (c - overlap if c > overlap else 0) for r, c in enumerate(counts)]


def _p2p(ops, dist):
    """All of a rank's sends and receives of one exchange as ONE group: over RCCL a lone isend and a lone irecv go to
    different per-peer communicators and can serialise against each other; a batch is one ncclGroup."""
    if not ops:
        return
    for req in dist.batch_isend_irecv([dist.P2POp(op, t, peer) for op, t, peer in ops]):
        req.wait()


def exchange_and_blend(frames: torch.Tensor, overlap: int, rank: int, world: int, dist=None,
                       counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """frames: this rank's finished frames [T,H,W,C] (any float dtype, on the rank's device).
    counts: the finished-frame count of EVERY rank before the exchange - local arithmetic on every rank (the segment
    split is deterministic: `segment_ranges`), so no control message is needed; None = every rank holds `T` frames.
    Returns the frames this rank owns after the boundary exchange: tail blended with the right neighbour's
    head, own head (already blended by the left neighbour) dropped.  The only communication is one grouped
    neighbour send/recv; ranks with too few frames follow the reference's edge cases."""
    if world <= 1 or overlap <= 0:
        return frames
    if dist is None:
        import torch.distributed as dist
    T = frames.shape[0]
    counts = [T] * world if counts is None else [int(c) for c in counts]
    if len(counts) != world or counts[rank] != T:
        raise ValueError(f"rank {rank}: holds {T} frames but the segment plan says {counts}")
    staged = _host_staged(frames, dist)
    cdev = "cpu" if staged else frames.device
    # frame counts of the neighbours decide whether a boundary is blended (reference: chunk.shape[0] > overlap)
    recv = None
    send_ok = rank > 0 and T > overlap and counts[rank - 1] >= overlap
    recv_ok = rank < world - 1 and counts[rank + 1] > overlap and T >= overlap
    ops = []
    if send_ok:
        head = frames[:overlap].contiguous()
        ops.append((dist.isend, head.cpu() if staged else head, rank - 1))
    if recv_ok:
        recv = torch.empty_like(frames[:overlap], device=cdev)
        ops.append((dist.irecv, recv, rank + 1))
    _p2p(ops, dist)
    if recv is not None and staged:
        recv = recv.to(frames.device)
    out = frames
    if recv_ok:
        w = blend_weights_f32(overlap, frames.device).view(overlap, *([1] * (frames.dim() - 1)))
        tail = out[-overlap:].float() * w + recv.float() * (1.0 - w)
        out = torch.cat((out[:-overlap], tail.to(out.dtype)), dim=0)
    if rank > 0:
        # reference: later chunks contribute chunk[overlap:] (nothing at all when chunk.shape[0] <= overlap)
        out = out[overlap:] if T > overlap else out[:0]
    return out


def gather_frames(frames: torch.Tensor, rank: int, world: int, dst: int = 0, dist=None,
                  counts: Optional[Sequence[int]] = None) -> Optional[torch.Tensor]:
    """Optional second step: collect every rank's owned frames on the writer rank (point-to-point, one group).
    counts: frames every rank holds NOW (`owned_counts` of the segment plan); None = `frames.shape[0]` everywhere."""
    if world <= 1:
        return frames
    if dist is None:
        import torch.distributed as dist
    counts = [frames.shape[0]] * world if counts is None else [int(c) for c in counts]
    if len(counts) != world or counts[rank] != frames.shape[0]:
        raise ValueError(f"rank {rank}: holds {frames.shape[0]} frames but the segment plan says {counts}")
    staged = _host_staged(frames, dist)
    cdev = "cpu" if staged else frames.device
    if rank != dst:
        if frames.shape[0] > 0:
            _p2p([(dist.isend, frames.contiguous().cpu() if staged else frames.contiguous(), dst)], dist)
        return None
    bufs = {r: torch.empty((counts[r],) + tuple(frames.shape[1:]), dtype=frames.dtype, device=cdev)
            for r in range(world) if r != dst and counts[r] > 0}
    _p2p([(dist.irecv, b, r) for r, b in bufs.items()], dist)
    parts = [frames if r == dst else (bufs[r].to(frames.device) if staged else bufs[r])
             for r in range(world) if r == dst or r in bufs]
    return torch.cat(parts, dim=0)


def upscale_distributed(generate, frames: torch.Tensor, overlap: int, rank: int, world: int, dist=None,
                        batch_size: int = 1, streaming: bool = True, gather_to: Optional[int] = 0):
    """The reference's multi-GPU mode as one call per rank [ref: inference_cli.py:1048-1209]: rank r upscales its
    segment of `frames` (all ranks hold the same LR clip) with `generate(segment_frames) -> [T', H, W, C]`, the segment
    boundaries are cross-faded through the neighbour exchange, and (gather_to is not None) the finished frames are
    collected on rank `gather_to`.  Returns the gathered clip on that rank (None elsewhere), or this rank's owned frames
    when gather_to is None."""
    ranges = segment_ranges(frames.shape[0], world, overlap, batch_size, streaming)
    s, e = ranges[rank]
    counts = [b - a for a, b in ranges]     # an upscaled segment has as many frames as its input: no count exchange
    mine = generate(frames[s:e])
    owned = exchange_and_blend(mine, overlap, rank, world, dist, counts)
    if gather_to is None:
        return owned
    return gather_frames(owned, rank, world, gather_to, dist, owned_counts(counts, overlap))
