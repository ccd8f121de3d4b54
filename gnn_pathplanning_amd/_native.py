"""Build and bind libgnnpp.so (hipcc, gfx950) -- the only compute path of this package.

The library is built IN-TREE next to this file so that it travels with the source snapshot to the
GPU box.  Binding is plain ctypes over the C ABI of include/gnnpp.h: raw device pointers, sizes and
a hipStream_t; PyTorch only supplies device memory and the stream.
"""
import ctypes
import os
import shutil
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libgnnpp.so')
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'gnnpp.h')
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
               '-Wno-unused-result']

EXPORTS = ('gnnpp_version', 'gnnpp_error_string', 'gnnpp_set_tuning', 'gnnpp_get_tuning', 'gnnpp_filter_packed_floats',
           'gnnpp_filter_pack', 'gnnpp_lsigf_fwd', 'gnnpp_lsigf_fwd_save', 'gnnpp_encoder_packed_floats',
           'gnnpp_encoder_pack', 'gnnpp_encoder_fwd', 'gnnpp_policy_fwd', 'gnnpp_decode_actions', 'gnnpp_rollout_observe', 'gnnpp_rollout_gso',
           'gnnpp_rollout_move', 'gnnpp_rollout_step', 'gnnpp_rollout_policy_step')


class GnnppError(RuntimeError):
    pass


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [HEADER]


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 csrc/gnnpp_api.hip -> libgnnpp.so (cross-compiles without a GPU)."""
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in _sources())):
        return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise GnnppError('hipcc not found: libgnnpp.so cannot be built on this machine')
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, 'gnnpp_api.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


class EncoderParams(ctypes.Structure):
    """struct gnnpp_encoder_params (include/gnnpp.h)."""
    _fields_ = [('conv_w', ctypes.c_void_p * 5), ('conv_b', ctypes.c_void_p * 5),
                ('bn_w', ctypes.c_void_p * 5), ('bn_b', ctypes.c_void_p * 5),
                ('bn_mean', ctypes.c_void_p * 5), ('bn_var', ctypes.c_void_p * 5),
                ('fc_w', ctypes.c_void_p), ('fc_b', ctypes.c_void_p), ('bn_eps', ctypes.c_float)]


class RolloutStruct(ctypes.Structure):
    """struct gnnpp_rollout (include/gnnpp.h)."""
    _fields_ = [('grid', ctypes.c_void_p), ('grid_batched', ctypes.c_int), ('goal', ctypes.c_void_p),
                ('pos', ctypes.c_void_p), ('B', ctypes.c_int), ('N', ctypes.c_int),
                ('H', ctypes.c_int), ('W', ctypes.c_int), ('obs', ctypes.c_void_p),
                ('radius', ctypes.c_void_p), ('S', ctypes.c_void_p), ('connected', ctypes.c_void_p),
                ('grow', ctypes.c_int), ('logits', ctypes.c_void_p), ('actions', ctypes.c_void_p),
                ('reached', ctypes.c_void_p), ('start_step', ctypes.c_void_p),
                ('end_step', ctypes.c_void_p), ('maxstep', ctypes.c_void_p),
                ('flags', ctypes.c_void_p), ('stats', ctypes.c_void_p), ('currentstep', ctypes.c_int),
                ('tie_mode', ctypes.c_int), ('seed', ctypes.c_uint), ('choices', ctypes.c_void_p),
                ('choice_count', ctypes.c_void_p), ('max_choices', ctypes.c_int)]


_lib = None


def lib():
    """The loaded library; raises loudly when it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnnppError(
            'libgnnpp.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.gnnpp_version.restype = ci
    L.gnnpp_error_string.restype = ctypes.c_char_p
    L.gnnpp_error_string.argtypes = [ci]
    L.gnnpp_set_tuning.argtypes = [ci, ci]
    L.gnnpp_set_tuning.restype = ci
    L.gnnpp_get_tuning.argtypes = [ci]
    L.gnnpp_get_tuning.restype = ci
    L.gnnpp_filter_packed_floats.restype = cs
    L.gnnpp_filter_packed_floats.argtypes = [ci] * 4
    L.gnnpp_filter_pack.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.gnnpp_lsigf_fwd.argtypes = [vp] * 5 + [ci] * 12 + [vp]
    L.gnnpp_lsigf_fwd_save.argtypes = [vp] * 6 + [ci] * 13 + [vp]
    L.gnnpp_lsigf_fwd_save.restype = ci
    L.gnnpp_encoder_packed_floats.restype = cs
    L.gnnpp_encoder_packed_floats.argtypes = []
    L.gnnpp_encoder_pack.argtypes = [ctypes.POINTER(EncoderParams), vp, vp]
    L.gnnpp_encoder_fwd.argtypes = [vp, vp, vp, ci, vp]
    L.gnnpp_policy_fwd.argtypes = [vp] * 9 + [ci] * 4 + [vp]
    L.gnnpp_decode_actions.argtypes = [vp, vp, ci, ci, vp]
    for f in ('gnnpp_rollout_observe', 'gnnpp_rollout_gso', 'gnnpp_rollout_move', 'gnnpp_rollout_step'):
        getattr(L, f).argtypes = [ctypes.POINTER(RolloutStruct), vp]
        getattr(L, f).restype = ci
    L.gnnpp_rollout_policy_step.argtypes = [ctypes.POINTER(RolloutStruct)] + [vp] * 5 + [ci, vp]
    L.gnnpp_rollout_policy_step.restype = ci
    for f in ('gnnpp_filter_pack', 'gnnpp_lsigf_fwd', 'gnnpp_encoder_pack', 'gnnpp_encoder_fwd',
              'gnnpp_policy_fwd', 'gnnpp_decode_actions', 'gnnpp_rollout_observe', 'gnnpp_rollout_gso',
           'gnnpp_rollout_move'):
        getattr(L, f).restype = ci
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise GnnppError('%s failed: %s (code %d)' % (what, lib().gnnpp_error_string(rc).decode(), rc))


def require_gpu(*tensors):
    """Every tensor must live on one HIP device; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GnnppError('gnn_pathplanning_amd runs on MI355X only: got a %s tensor; move the '
                             'module and its inputs to a HIP device (there is no CPU fallback)'
                             % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GnnppError('tensors on different devices: %s vs %s' % (dev, t.device))
    return dev


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class device_guard:
    """Make `device` current for raw HIP launches (no-op in the one-process-per-GPU layout)."""

    def __init__(self, device):
        self.dev = device
        self.ctx = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class PackCache:
    """Packed (MFMA-fragment-ordered) copy of some parameters, rebuilt when any of them changes
    (load_state_dict / optimizer step / .to()): keyed on (data_ptr, _version) of each tensor."""

    def __init__(self):
        self.key = None
        self.buf = None

    def get(self, tensors, pack_fn):
        # _version changes on every in-place update; id() changes when .to()/.cuda() replaces the
        # tensor objects' storage holders.  (data_ptr() per tensor is 3x slower than this.)
        key = tuple([t._version for t in tensors] + [id(t) for t in tensors] +
                    [tensors[0].data_ptr(), tensors[-1].data_ptr()])   # .to(device) swaps storage
        if key != self.key:
            self.buf = pack_fn()
            self.key = key
        return self.buf
