import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never silently pass without the device
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    r"""GPU runs: write the parity ledger (tests/util.py: record) next to the other scratch output."""
    try:
        import json
        import util
        if not util.LEDGER:
            return
        import torch
        out = os.environ.get('MRPHY_PARITY_LEDGER') or os.path.join(ROOT, 'gpurun_out', 'parity_ledger.json')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        meta = {'device': torch.cuda.get_device_name(0) if torch.cuda.is_available() else 'cpu',
                'torch': torch.__version__, 'exitstatus': int(exitstatus), 'entries': len(util.LEDGER)}
        with open(out, 'w') as f:
            json.dump({'meta': meta, 'distances': util.LEDGER}, f, indent=1, sort_keys=True)
    except Exception as e:                      # never turn a green suite red over bookkeeping
        print(f'parity ledger not written: {e}')
