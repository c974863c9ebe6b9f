"""`python bench.py --gpus N` starts its N ranks by itself (the reference's entry point spawns its own workers,
inference_cli.py:1048-1112): here through tests/bench_cpu_child.py, which substitutes only the GPU-bound pieces (models,
three device ops) and runs the bench's own launcher, rank handling, gloo process group, segment split, exchange, timing
protocol and JSON line on the CPU."""

import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = [sys.executable, os.path.join(HERE, "bench_cpu_child.py")]


def _run(args, env_extra=None, expect_rc=0):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run(CHILD + args, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return lines, r


def test_gpus_2_self_launches_two_ranks_weak():
    lines, r = _run(["--gpus", "2", "--workload", "cfg0", "--steps", "6", "--warmup", "3", "--no-cpu-baseline"])
    assert len(lines) == 1, r.stdout                 # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["launcher"] == "self" and j["backend"] == "gloo"
    assert j["scaling"] == "weak" and j["steps"] == 6 and j["warmup"] == 3 and j["unit"] == "frames/s"
    # 26-frame clip per rank, batch 9, overlap 4: five batches of 9 frames (the last one 6 + 3 padding frames);
    # warm-up 3 + 6 steps = batches 3, 4 (segment complete -> exchange), 0, 1, 2, 3
    c = j["config"]
    assert c["frames_per_batch"] == [9, 9, 9, 9, 9]
    assert c["segment_frames_by_rank"] == [26, 26]
    assert j["exchange_ms"]["exchanges_rank0"] >= 1 and j["rank_ms_per_step"]["max"] >= j["rank_ms_per_step"]["min"] > 0
    assert j["value"] > 0 and abs(j["value"] - c["finished_frames"] / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    for key in ("metric", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in j


def test_strong_scaling_splits_one_clip_like_the_reference():
    lines, _ = _run(["--gpus", "3", "--workload", "cfg0", "--steps", "2", "--warmup", "1", "--scaling", "strong",
                     "--no-cpu-baseline"])
    j = json.loads(lines[0])
    c = j["config"]
    # 26 frames over 3 ranks: 9 + 9 + 8, + 4 overlap frames for all but the last rank [ref: inference_cli.py:1082-1097]
    assert j["n_gpus"] == 3 and j["scaling"] == "strong" and c["segment_frames_by_rank"] == [13, 13, 8]
    assert c["clip_frames_total"] == 26 and c["finished_frames"] == 2 * 26        # every step finishes the whole clip once
    assert j["exchange_ms"]["exchanges_rank0"] == 2


def test_single_rank_and_wrong_world_size():
    lines, _ = _run(["--workload", "cfg0", "--steps", "5", "--warmup", "0", "--no-cpu-baseline"])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["launcher"] == "single" and j["exchange_ms"] is None and j["config"]["finished_frames"] == 26
    # a launcher that set WORLD_SIZE to something else than --gpus: refuse instead of reporting under a wrong n_gpus
    _, r = _run(["--gpus", "2", "--workload", "cfg0", "--no-cpu-baseline"],
                {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_rc=1)
    assert "WORLD_SIZE=1" in r.stderr


def test_a_dead_rank_ends_the_bench_non_zero():
    lines, r = _run(["--gpus", "2", "--workload", "cfg0", "--no-cpu-baseline"], {"SVR2_TEST_FAIL_RANK": "1"}, expect_rc=1)
    assert not lines and "exit codes" in r.stderr


def test_eight_ranks_strong_scaling_with_the_references_edge_case():
    """--gpus 8 (the driver's largest run) through the self-launch: 26 frames over 8 ranks -> segments of 3 or 4 (+4 overlap,
    clipped at the clip's end) frames; the last rank holds 3 <= overlap frames, which the reference's concatenation drops
    (inference_cli.py:1163-1198) - they are already inside its left neighbour's clipped overlap: the exchange must neither
    hang nor lose or double-count a frame."""
    lines, _ = _run(["--gpus", "8", "--workload", "cfg0", "--steps", "2", "--warmup", "1", "--scaling", "strong",
                     "--no-cpu-baseline"])
    j = json.loads(lines[0])
    c = j["config"]
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8
    assert c["segment_frames_by_rank"] == [8, 8, 7, 7, 7, 7, 6, 3] and c["clip_frames_total"] == 26
    owned = [8] + [s - 4 if s > 4 else 0 for s in c["segment_frames_by_rank"][1:]]
    assert c["finished_frames"] == 2 * sum(owned) == 2 * 26
