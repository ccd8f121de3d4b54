"""Parameter initialisation rule of the reference (graphs/weights_initializer.py:11-23):
Conv*/Linear weights xavier-normal, Linear bias 0 (conv bias keeps torch's default), BatchNorm
weight ~ N(1, 0.02) and bias 0.  Dispatch is by class name exactly like the reference, so
GraphFilter* modules (own reset_parameters) are untouched."""
import torch

from . import _native


def weights_init(m):
    name = type(m).__name__
    if 'Conv' in name:
        torch.nn.init.xavier_normal_(m.weight)
    elif 'BatchNorm' in name:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0.0)
    elif 'Linear' in name:
        torch.nn.init.xavier_normal_(m.weight)
        m.bias.data.fill_(0.0)
    _native.invalidate_packs()                  # `.data` writes are invisible to the version counters
