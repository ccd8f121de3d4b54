"""Wire formats of the reference, so its checkpoints and datasets run
through the MI355X path unchanged.

Checkpoints -- agents/decentralplannerlocal.py:114-214: torch.save of
    {'epoch', 'iteration', 'state_dict', 'optimizer', 'scheduler_state_dict'}
as `checkpoint.pth.tar` / `checkpoint_{epoch:03d}.pth.tar` / `model_best.pth.tar`.
Cases -- offlineExpert/DataGen_Transformer.py:351-371 writes one .mat per solved MAPF case with the
keys map [W,H], goal [N,2], inputState [T,N,2], inputTensor [T,N,3,11,11], target [T,N,5],
GSO [T,N,N], makespan; dataloader/Dataloader_dcplocal_notTF_onlineExpert.py:142-205 reads them
(one timestep per training item; initial positions only for validation / test).
"""
import os
import shutil
from fnmatch import fnmatch

import numpy as np
import torch

CHECKPOINT_LATEST = 'checkpoint.pth.tar'
CHECKPOINT_BEST = 'model_best.pth.tar'


def checkpoint_name(epoch=None, latest=True, best=False):
    """File naming of agents/decentralplannerlocal.py:121-124 / :146-151."""
    if latest:
        return CHECKPOINT_LATEST
    if best:
        return CHECKPOINT_BEST
    return 'checkpoint_{:03d}.pth.tar'.format(epoch)


def save_checkpoint(checkpoint_dir, model, optimizer, scheduler, epoch, iteration, is_best=False,
                    latest=True):
    """agents/decentralplannerlocal.py:114-138 (note: the stored epoch is current_epoch + 1)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    name = checkpoint_name(epoch, latest)
    state = {'epoch': epoch + 1, 'iteration': iteration, 'state_dict': model.state_dict(),
             'optimizer': optimizer.state_dict(), 'scheduler_state_dict': scheduler.state_dict()}
    path = os.path.join(checkpoint_dir, name)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(checkpoint_dir, CHECKPOINT_BEST))
    return path


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location=None, train_TL=False):
    """agents/decentralplannerlocal.py:187-214 (and :140-184 with train_TL: everything except the
    graph filter and the action head is frozen).  `map_location` defaults to the model's device
    (the reference hard-codes 'cuda:N', which works unchanged on ROCm).  Returns (epoch, iteration)."""
    if map_location is None:
        map_location = next(model.parameters()).device
    ckpt = torch.load(path, map_location=map_location)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    if scheduler is not None and 'scheduler_state_dict' in ckpt:
        scheduler.load_state_dict(ckpt['scheduler_state_dict'])
    if train_TL:
        freeze_for_transfer_learning(model)
    return ckpt.get('epoch', 0), ckpt.get('iteration', 0)


def freeze_for_transfer_learning(model, keep=('*GFL*', '*actions*')):
    """agents/decentralplannerlocal.py:168-179."""
    for name, p in model.named_parameters():
        p.requires_grad = any(fnmatch(name, pat) for pat in keep)


# ---- .mat cases ---------------------------------------------------------------------------------
def _loadmat(path):
    import scipy.io as sio
    return sio.loadmat(path)


def save_case_mat(path, grid, goal, input_state, target, makespan, input_tensor=None, gso=None):
    """Write a case with the key set of offlineExpert/DataGen_Transformer.py:366-368."""
    import scipy.io as sio
    d = {'map': np.asarray(grid), 'goal': np.asarray(goal), 'inputState': np.asarray(input_state),
         'target': np.asarray(target), 'makespan': makespan}
    if input_tensor is not None:
        d['inputTensor'] = np.asarray(input_tensor)
    if gso is not None:
        d['GSO'] = np.asarray(gso)
    sio.savemat(path, d, do_compression=True)


def load_training_step(path, id_step):
    """One training item, Dataloader...:142-157: (input [N,3,11,11] float, target [N,5] long,
    GSO [N,N] float, map float)."""
    d = _loadmat(path)
    return (torch.from_numpy(d['inputTensor'][id_step][:]).float(),
            torch.from_numpy(d['target'][id_step, :, :]).long(),
            torch.from_numpy(d['GSO'][id_step, :, :]).float(),
            torch.from_numpy(d['map']).float())


def load_test_case(path, from_training_set=False):
    """A validation / test item, Dataloader...:160-205: (input [2,N,2] = stack(goal, start),
    target [N,T,5], map).  Training-set files hold the whole trajectory in inputState; only the
    initial positions are used (:168)."""
    d = _loadmat(path)
    state = d['inputState'][0] if (from_training_set or d['inputState'].ndim == 3) else d['inputState']
    inp = torch.FloatTensor(np.stack((d['goal'], state)))
    target = torch.from_numpy(d['target']).long().permute(1, 0, 2)
    return inp, target, torch.from_numpy(d['map']).float(), int(np.asarray(d['makespan']).reshape(-1)[0])


def rollout_from_cases(paths, device, rate_maxstep=2, commR=6.0, **kw):
    """Build a BatchedRollout from reference test-case files (all with the same map size and number
    of agents).  maxstep = rate_maxstep * makespan (3 when N >= 20), multirobotsim_dcenlocal.py:76-81."""
    from .rollout import BatchedRollout
    grids, starts, goals, maxsteps = [], [], [], []
    for p in paths:
        inp, _, grid, makespan = load_test_case(p)
        goals.append(inp[0].long())
        starts.append(inp[1].long())
        grids.append(grid)
        rate = 3 if inp.shape[1] >= 20 else rate_maxstep
        maxsteps.append(int(makespan * rate))
    return BatchedRollout(torch.stack(grids), torch.stack(starts), torch.stack(goals),
                          torch.tensor(maxsteps), device, commR=commR, **kw)
