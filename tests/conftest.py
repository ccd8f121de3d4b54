import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z['meta']).decode())
    return z, meta


@pytest.fixture(scope='session')
def lsigf_golden():
    return _load('lsigf_cases.npz')


@pytest.fixture(scope='session')
def policy_golden():
    return _load('policy_model.npz')


@pytest.fixture(scope='session')
def policy_large_golden():
    """Teams of 50 / 64 / 100 agents through the REAL reference (oracle/gen_golden.py large); parameters are
    policy_model.npz's."""
    return _load('policy_large.npz')


def golden_state_dict(z, K=3):
    import torch
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    if K != 3:
        sd['GFL.0.weight'] = torch.from_numpy(np.array(z['gfl_w_K%d' % K]))
    return sd


@pytest.fixture(scope='session')
def rollout_golden():
    return _load('rollout_traces.npz')


@pytest.fixture(scope='session')
def rollout_large_golden():
    """Teams of 50 / 100 agents on 50 x 50 / 100 x 100 maps through the REAL reference simulator
    (oracle/gen_golden_rollout.py large)."""
    return _load('rollout_traces_large.npz')


@pytest.fixture(scope='session')
def training_golden():
    return _load('training_grads.npz')


@pytest.fixture(scope='session')
def multilayer_golden():
    return _load('policy_multilayer.npz')


@pytest.fixture(scope='session')
def multilayer_training_golden():
    return _load('training_multilayer.npz')


def multilayer_state_dict(zp, zm, ci, prefix='m'):
    """Encoder parameters of policy_model.npz + the graph-filter layers / head of multilayer case ci."""
    import torch
    sd = {k: v for k, v in golden_state_dict(zp, 3).items() if not k.startswith(('GFL.', 'actionsMLP.'))}
    pre = '%s%d_' % (prefix, ci)
    for k in zm.files:
        if k.startswith(pre + 'GFL.') or k.startswith(pre + 'actionsMLP.'):
            sd[k[len(pre):]] = torch.from_numpy(np.array(zm[k]))
    return sd
