"""world_size-2 (and 3) gloo runs of the frame-chunk sharding on the CPU: partition + neighbour exchange +
blend must equal the reference's multi-GPU semantics (segment-wise processing, boundary blend on the
final frames), restated here from inference_cli.py:1082-1097 and :1163-1198."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _fake_upscale(seg: torch.Tensor, rank: int) -> torch.Tensor:
    """Deterministic stand-in for the per-segment pipeline: depends on the frame id and on the segment."""
    return (seg.float() * 0.01 + 0.1 * rank).view(-1, 1, 1, 1).expand(-1, 2, 3, 3).contiguous()


def _reference_concat(results, overlap):
    """The reference's parent-process concatenation [ref: inference_cli.py:1163-1198]."""
    from seedvr2_ref.pipeline import blend_overlapping_frames
    if overlap <= 0:
        return torch.cat([c.float() for c in results], dim=0)
    out = None
    for idx, chunk in enumerate(results):
        chunk = chunk.float()
        if idx == 0:
            out = chunk
        elif chunk.shape[0] > overlap and out.shape[0] >= overlap:
            blended = blend_overlapping_frames(out[-overlap:], chunk[:overlap], overlap)
            out = torch.cat([out[:-overlap], blended, chunk[overlap:]], dim=0)
        elif chunk.shape[0] > overlap:
            out = torch.cat([out, chunk[overlap:]], dim=0)
    return out


def _worker(rank, world, port, total, overlap, q, done):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank holds the whole LR clip (here: the frame ids); the driver segments, exchanges and gathers
        full = pkg.parallel.upscale_distributed(lambda seg: _fake_upscale(seg, rank), torch.arange(total), overlap, rank,
                                                world, dist)
        if rank == 0:
            # plain bytes through the pipe (no shared-memory handle that could outlive this process), and rank 0 stays
            # alive until the parent has the result
            q.put(full.float().numpy())
            done.wait(timeout=120)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (8, 512, 4) is BASELINE configs[4]'s partition: 8 ranks x 64 frames (+4 overlap), as 8 gloo processes
@pytest.mark.parametrize("world,total,overlap", [(2, 23, 4), (2, 10, 0), (3, 17, 2), (2, 9, 3), (8, 512, 4)])
def test_frame_sharding_matches_reference_semantics(pkg, world, total, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    done = ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, overlap, q, done)) for r in range(world)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=240))
    done.set()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = pkg.parallel.segment_ranges(total, world, overlap)
    ref = _reference_concat([_fake_upscale(torch.arange(s, e), r) for r, (s, e) in enumerate(ranges)], overlap)
    assert got.shape == ref.shape and got.shape[0] == total
    assert torch.allclose(got.float(), ref, atol=1e-6)


def test_segment_ranges_reference_split(pkg):
    P = pkg.parallel
    # streaming split: base + remainder, + overlap for all but the last rank [ref: inference_cli.py:1082-1097]
    assert P.segment_ranges(512, 8, 4) == [(64 * r, 64 * (r + 1) + (4 if r < 7 else 0)) for r in range(8)]
    assert P.segment_ranges(10, 3, 0) == [(0, 4), (4, 7), (7, 10)]
    assert P.segment_ranges(10, 1, 4) == [(0, 10)]
    # pre-loaded split rounds the chunk up to a multiple of batch_size [ref: inference_cli.py:1118-1133]
    assert P.segment_ranges(100, 2, 4, batch_size=5, streaming=False) == [(0, 55), (51, 100)]


def test_counts_are_local_arithmetic_and_checked_before_any_communication(pkg):
    """Round 3: no count exchange.  owned_counts restates the reference's concatenation rule (later chunks contribute
    chunk[overlap:], nothing when chunk.shape[0] <= overlap, inference_cli.py:1163-1198); a rank whose frame count contradicts
    the segment plan raises BEFORE it touches the process group, so it cannot leave a neighbour waiting on a half-made exchange."""
    P = pkg.parallel
    assert P.owned_counts([68, 68, 64], 4) == [68, 64, 60]
    assert P.owned_counts([10, 3, 9], 4) == [10, 0, 5]
    assert P.owned_counts([7, 7], 0) == [7, 7]

    class NoDist:                       # any use of the process group fails the test
        def __getattr__(self, name):
            raise AssertionError(f"dist.{name} used")

    frames = torch.zeros(5, 2, 2, 3)
    with pytest.raises(ValueError):
        P.exchange_and_blend(frames, 2, 0, 2, NoDist(), counts=[6, 9])
    with pytest.raises(ValueError):
        P.gather_frames(frames, 1, 2, 0, NoDist(), counts=[5, 4])
    # world 1 / no overlap: nothing to exchange, the frames come back as they are
    assert P.exchange_and_blend(frames, 2, 0, 1, NoDist()) is frames
    assert P.exchange_and_blend(frames, 0, 1, 2, NoDist()) is frames
