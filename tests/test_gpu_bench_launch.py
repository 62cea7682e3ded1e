"""On a GPU box: `bench.py --gpus N` with fewer than N devices fails loudly."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_more_ranks_than_devices_fails_loudly():
    import torch
    ndev = torch.cuda.device_count()
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 2
    assert "%d GPUs requested" % (ndev + 1) in r.stderr and "%d visible" % ndev in r.stderr
