import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _ensure_oracle_built():
    lib = os.path.join(ROOT, 'oracle', 'libklang_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'klang_oracle.c')
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libklang_oracle.so'],
                              stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure; never the product path)."""
    _ensure_oracle_built()
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope='session')
def golden():
    """Loader for fixtures produced by tools/gen_golden.py from the real reference."""
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
        return cache[name]

    return load


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def hip():
    """The product library through its ctypes binding. Fails loudly when the
    HIP library is missing; skips only when there is no GPU."""
    from klang_amd import native
    native.load()  # raises if libklang_amd.so is absent
    if not has_gpu():
        pytest.skip('no GPU in this environment')
    return native


def pytest_sessionfinish(session, exitstatus):
    """Per-block-type parity table of this run (float32 output vs the reference's float64)."""
    import json
    from tests.helpers import REPORT
    if not REPORT:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    table = {k: REPORT[k] for k in sorted(REPORT)}
    with open(os.path.join(out, 'parity_report.json'), 'w') as fh:
        json.dump({'bound': 1e-5, 'unit': 'max abs error, float32 output vs reference float64',
                   'exit_status': int(exitstatus), 'block_types': table}, fh, indent=1)
