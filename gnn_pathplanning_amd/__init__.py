"""gnn_pathplanning_amd -- MI355X (gfx950) implementation of the GNN policy forward pass of
proroklab/gnn_pathplanning behind the reference's own module API.

    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet      # graphs/models/decentralplanner.py
    import gnn_pathplanning_amd.graphML as gml                                  # utils/graphUtils/graphML.py

The compute path is the hand-written HIP library libgnnpp.so (C ABI in include/gnnpp.h); there is
no CPU or eager-PyTorch fallback: calling a forward without the library or with non-GPU tensors
raises.
"""
from . import _native  # noqa: F401

__all__ = ['_native']
__version__ = '0.1.0'
