"""pytest configuration: registers the `gpu` marker and puts the repo on sys.path.

`-m "not gpu"`: oracle-vs-golden, host logic, C-ABI symbol checks (runs on CPU).
`-m gpu`: parity tests proper - the HIP path (through the C-ABI) against the oracle / golden vectors.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _lib_is_stale():
    lib = os.path.join(ROOT, 'yolo-somi_amd', 'lib', 'libsomi_hip.so')
    if not os.path.exists(lib):
        return True
    src = os.path.join(ROOT, 'yolo-somi_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith(('.hip', '.h')) or f == 'Makefile')
    return max(newest, os.path.getmtime(os.path.join(ROOT, 'include', 'somi_hip.h'))) > os.path.getmtime(lib)


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """The tests exercise the in-tree libsomi_hip.so: (re)build it when it is missing or older than its sources, so a stale
    binary can never be what gets tested.  Without hipcc the library tests fail loudly on load, as the product does."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)
    if hipcc and _lib_is_stale():
        subprocess.run(['make', '-C', os.path.join(ROOT, 'yolo-somi_amd', 'csrc'), '-j', str(min(8, os.cpu_count() or 1))],
                       check=True, env=dict(os.environ, HIPCC=hipcc), stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load
