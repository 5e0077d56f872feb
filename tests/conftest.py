import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _excused_share_report(request, record_property):
    """Every AdamAwareCompare a test used leaves the largest share of a tensor it held to the looser bound in the
    test's report (junit property `excused_share[...]`; with DPCGAN_EXCUSED_LOG=<file> also one line per comparer)."""
    import golden_util
    golden_util.LIVE.clear()
    yield
    for i, c in enumerate(golden_util.LIVE):
        c.report(record_property, '%s#%d' % (request.node.name, i))
    golden_util.LIVE.clear()
