import importlib
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def trt():
    """The product package (loads libtinyrt.so; raises if the HIP extension is not built)."""
    lib_path = os.path.join(ROOT, "tiny-raytracer_amd", "libtinyrt.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    return importlib.import_module("tiny-raytracer_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as o
    return o


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


def sym(v):
    """Numbers in the KAT file may be symbolic strings."""
    table = {"inf": math.inf, "sqrt2": math.sqrt(2.0), "sqrt3": math.sqrt(3.0), "sqrt3/2": math.sqrt(3.0) / 2.0,
             "sqrt3/4": math.sqrt(3.0) / 4.0, "16/9": 16.0 / 9.0, "-2*16/9": -2.0 * 16.0 / 9.0}
    if isinstance(v, str):
        return table[v]
    if isinstance(v, list):
        return [sym(x) for x in v]
    return float(v)
