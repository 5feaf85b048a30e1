import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: longer CPU test')


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('test marked gpu but no HIP device is visible')
    return torch.device('cuda', 0)


def pytest_sessionfinish(session, exitstatus):
    """-m gpu sessions: the measured margin of every golden case that ran (max|d|/max, cosine, per-firing trace error) as
    gpurun_out/parity_report.json, so the distance to each tolerance is visible (a copy is committed under profiles/rNN/)."""
    mod = sys.modules.get('test_gpu_parity')
    rep = getattr(mod, 'PARITY_REPORT', None) if mod else None
    if not rep:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'parity_report.json'), 'w') as f:
        json.dump({'cases': len(rep), 'schedule': 'default (lean where it applies)', 'results': dict(sorted(rep.items()))}, f, indent=1)
