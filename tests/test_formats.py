"""CPU: checkpoint and .mat case formats.  The .mat fixture was written
with the reference's key set and read back by the REFERENCE's own loader methods
(oracle/gen_golden_rollout.py); our readers must return the same tensors."""
import os

import numpy as np
import torch

from conftest import GOLDEN, golden_state_dict
from gnn_pathplanning_amd import formats

MAT = os.path.join(GOLDEN, 'case_fixture.mat')


def test_mat_case_readers_match_reference_loader():
    want = np.load(os.path.join(GOLDEN, 'case_fixture_expected.npz'))
    inp, tgt, gso, grid = formats.load_training_step(MAT, 3)
    assert inp.dtype == torch.float32 and tgt.dtype == torch.int64 and gso.dtype == torch.float32
    assert np.array_equal(inp.numpy(), want['train_input'])
    assert np.array_equal(tgt.numpy(), want['train_target'])
    assert np.array_equal(gso.numpy(), want['train_gso'])
    assert np.array_equal(grid.numpy(), want['train_map'])
    tinp, ttgt, tmap, makespan = formats.load_test_case(MAT, from_training_set=True)
    assert np.array_equal(tinp.numpy(), want['test_input'])
    assert np.array_equal(ttgt.numpy(), want['test_target'])
    assert np.array_equal(tmap.numpy(), want['test_map'])
    assert makespan == want['test_target'].shape[1]


def test_mat_case_roundtrip(tmp_path):
    want = np.load(os.path.join(GOLDEN, 'case_fixture_expected.npz'))
    import scipy.io as sio
    d = sio.loadmat(MAT)
    p = str(tmp_path / 'case.mat')
    formats.save_case_mat(p, d['map'], d['goal'], d['inputState'], d['target'], int(d['makespan'][0, 0]),
                          input_tensor=d['inputTensor'], gso=d['GSO'])
    inp, tgt, gso, grid = formats.load_training_step(p, 3)
    assert np.array_equal(inp.numpy(), want['train_input']) and np.array_equal(gso.numpy(), want['train_gso'])
    assert set(sio.loadmat(p)) >= {'map', 'goal', 'inputState', 'inputTensor', 'target', 'GSO', 'makespan'}


class Cfg:
    num_agents, nGraphFilterTaps, device = 10, 3, torch.device('cpu')


def test_checkpoint_format_and_roundtrip(tmp_path, policy_golden):
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    z, _ = policy_golden
    net = DecentralPlannerNet(Cfg())
    net.load_state_dict(golden_state_dict(z, 3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=150, eta_min=1e-6)
    path = formats.save_checkpoint(str(tmp_path), net, opt, sch, epoch=7, iteration=1234, is_best=True)
    assert os.path.basename(path) == 'checkpoint.pth.tar'
    assert os.path.exists(os.path.join(str(tmp_path), 'model_best.pth.tar'))
    assert formats.checkpoint_name(12, latest=False) == 'checkpoint_012.pth.tar'
    raw = torch.load(path, map_location='cpu')
    assert set(raw) == {'epoch', 'iteration', 'state_dict', 'optimizer', 'scheduler_state_dict'}
    assert raw['epoch'] == 8 and raw['iteration'] == 1234            # current_epoch + 1 (:125)
    assert list(raw['state_dict']) == list(golden_state_dict(z, 3))  # the reference's key order
    net2 = DecentralPlannerNet(Cfg())
    opt2 = torch.optim.Adam(net2.parameters(), lr=5e-2)
    sch2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=150, eta_min=1e-6)
    ep, it = formats.load_checkpoint(path, net2, opt2, sch2, train_TL=True)
    assert (ep, it) == (8, 1234) and opt2.param_groups[0]['lr'] == 1e-3
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    trainable = [n for n, p in net2.named_parameters() if p.requires_grad]
    assert trainable == ['GFL.0.weight', 'GFL.0.bias', 'actionsMLP.0.weight', 'actionsMLP.0.bias']
