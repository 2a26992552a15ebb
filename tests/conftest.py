import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import parallel-genomeseq_amd/ (hyphenated directory) as module `parallel_genomeseq_amd`."""
    name = "parallel_genomeseq_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "parallel-genomeseq_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pgs():
    return load_package()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "ref_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def data_small():
    with open(os.path.join(ROOT, "tests", "golden", "data_small.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding
