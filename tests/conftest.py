import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def patterns_blob():
    with open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle_trie(patterns_blob):
    import oraclelib
    return oraclelib.Trie(blob=patterns_blob)
