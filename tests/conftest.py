import gzip
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def g1_index(tmp_path_factory, golden_dir):
    """Unpack the committed reference-built index of the small genome g1; returns its basename."""
    d = tmp_path_factory.mktemp("g1")
    for k in range(1, 9):
        with gzip.open(os.path.join(golden_dir, f"g1.{k}.ht2.gz"), "rb") as f, open(d / f"g1.{k}.ht2", "wb") as o:
            shutil.copyfileobj(f, o)
    return str(d / "g1")


@pytest.fixture(scope="session")
def g1s_index(tmp_path_factory, golden_dir):
    """The reference-built GRAPH index (g1 + ~500 seeded variants, hisat2-build-s --snp)."""
    d = tmp_path_factory.mktemp("g1s")
    for k in range(1, 9):
        with gzip.open(os.path.join(golden_dir, f"g1s.{k}.ht2.gz"), "rb") as f, open(d / f"g1s.{k}.ht2", "wb") as o:
            shutil.copyfileobj(f, o)
    return str(d / "g1s")


@pytest.fixture(scope="session")
def oracle_lib():
    """Build (if needed) and load the CPU oracle.  Test infrastructure only."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    import h2o_py
    return h2o_py.load()
