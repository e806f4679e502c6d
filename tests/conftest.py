import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# every cooperative launch of the suite is checked for a device-side wait that gave up (device_index.hip: coop_check)
os.environ.setdefault("SGPU_COOP_CHECK", "1")
# the suite forces code paths through undocumented hooks (SGPU_BLOCK, SGPU_ITEMS_MAX, SGPU_COOP_CHUNK_MIN ...): the library
# honours them only while this is set (device_index.hip: hooks_on); a deployment's behaviour does not depend on them
os.environ.setdefault("SGPU_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
