import os.path as osp
import sys

import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
for p in (ROOT, osp.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pack():
    from spark_sched_sim_amd import workload

    return workload.default_pack()
