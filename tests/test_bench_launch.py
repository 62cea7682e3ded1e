"""bench.py's launch contract: `--gpus N` means N ranks — started by torch.distributed.run, or spawned by bench.py itself when it
is started alone — never a silent single rank.  The dry run exercises exactly that path (spawn, rendezvous on 127.0.0.1 over gloo, id-range shards,
the counter all-reduce) without a device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_2_spawns_two_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--pairs", "1001"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["world_size"] == 2
    assert line["pairs_global"] == 2002 and line["rank_sum"] == 3            # the all-reduce saw both ranks
    assert line["shards"] == [[0, 1001], [1001, 2002]]                       # weak scaling: --pairs per rank, contiguous id ranges


def test_gpus_2_strong_scaling_splits_one_batch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--strong", "--pairs", "1001"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["shards"] == [[0, 500], [500, 1001]] and line["pairs_global"] == 1001 and line["scaling"] == "strong"


def test_world_size_must_equal_gpus():
    e = _env()
    e.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_more_gpus_than_devices_is_refused_not_shrunk():
    """here: no device at all; on a 1-GPU box the same refusal says '2 requested, 1 visible' (tests/test_gpu_bench_launch.py)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 2
    assert "2 GPUs requested" in r.stderr and "visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no line: nothing to mistake for a measurement
