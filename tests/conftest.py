import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# tests of rejected kernel designs / A/B switches that exist in the LAB library only (VQCPC_LAB=1 python -m vqcpc_bach_amd.build;
# run them with VQCPC_LAB=1 python -m pytest tests -m gpu -k lab)
lab_only = pytest.mark.skipif(os.environ.get('VQCPC_LAB', '0') != '1', reason='needs the lab build (VQCPC_LAB=1)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub_state(g, prefix):
    """{'sd0/encoder/downscaler.x': arr} -> {'encoder.downscaler.x': tensor}"""
    out = {}
    for k, v in g.items():
        if k.startswith(prefix + '/'):
            out[k[len(prefix) + 1:].replace('/', '.')] = torch.from_numpy(np.array(v))
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
