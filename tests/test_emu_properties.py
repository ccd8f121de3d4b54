"""CPU: property tests of the graph-filter algebra on the emulated HIP kernel (SURVEY.md section 4
(iii)): random shapes/taps/GSOs from hypothesis, checked against the float64 einsum statement and
against the algebraic identities the reference's definition implies."""
import os
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import policy_oracle as orc      # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'),
                                reason='host clang++ from ROCm not present')
SET = dict(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])


@pytest.fixture(scope='module')
def emu():
    import emu_lib
    return emu_lib, emu_lib.load()


shapes = st.tuples(st.integers(1, 3), st.integers(1, 40), st.integers(1, 40), st.integers(1, 4),
                   st.integers(1, 2), st.integers(1, 12), st.integers(0, 10 ** 6))


def make(B, G, F, K, E, N, seed, density=0.5):
    g = np.random.default_rng(seed)
    h = (g.standard_normal((F, E, K, G)) / np.sqrt(G * K)).astype(np.float32)
    x = g.standard_normal((B, G, N)).astype(np.float32)
    S = ((g.random((B, E, N, N)) < density) * g.random((B, E, N, N))).astype(np.float32)
    b = (0.1 * g.standard_normal((F, 1))).astype(np.float32)
    return h, x, S, b


@settings(**SET)
@given(shapes)
def test_matches_float64_statement(emu, s):
    el, lib = emu
    B, G, F, K, E, N, seed = s
    h, x, S, b = make(B, G, F, K, E, N, seed)
    y = el.lsigf(lib, h, S, x, b, True)
    want = orc.lsigf_f64(h, S, x, b)
    assert np.abs(y - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@settings(**SET)
@given(shapes)
def test_zero_gso_keeps_only_tap_zero_and_k1_is_linear(emu, s):
    el, lib = emu
    B, G, F, K, E, N, seed = s
    h, x, S, b = make(B, G, F, K, E, N, seed)
    y0 = el.lsigf(lib, h, np.zeros_like(S), x, None, True)
    want = np.einsum('fg,bgn->bfn', h[:, :, 0].sum(1).astype(np.float64), x.astype(np.float64))
    assert np.abs(y0 - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    y1 = el.lsigf(lib, np.ascontiguousarray(h[:, :, :1]), S, x, None, True)     # K = 1: S irrelevant
    assert np.abs(y1 - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@settings(**SET)
@given(shapes)
def test_permutation_equivariance_and_padding(emu, s):
    el, lib = emu
    B, G, F, K, E, N, seed = s
    h, x, S, b = make(B, G, F, K, E, N, seed)
    y = el.lsigf(lib, h, S, x, b, True)
    perm = np.random.default_rng(seed + 1).permutation(N)
    Sp = np.ascontiguousarray(S[:, :, perm][:, :, :, perm])
    yp = el.lsigf(lib, h, Sp, np.ascontiguousarray(x[:, :, perm]), b, True)
    assert np.abs(yp - y[:, :, perm]).max() <= 1e-4 * max(1.0, np.abs(y).max())
    if N > 1:                                     # Nin < N == zero padding + slicing
        Nin = N - 1
        xz = x.copy()
        xz[:, :, Nin:] = 0
        ypad = el.lsigf(lib, h, S, np.ascontiguousarray(x[:, :, :Nin]), b, True, Nin=Nin)
        yfull = el.lsigf(lib, h, S, xz, b, True)
        assert np.abs(ypad - yfull[:, :, :Nin]).max() <= 1e-5 * max(1.0, np.abs(yfull).max())


@settings(**SET)
@given(shapes)
def test_shared_gso_equals_repeated_batched_gso(emu, s):
    el, lib = emu
    B, G, F, K, E, N, seed = s
    h, x, S, b = make(B, G, F, K, E, N, seed)
    y_shared = el.lsigf(lib, h, np.ascontiguousarray(S[0]), x, b, False)
    y_batched = el.lsigf(lib, h, np.ascontiguousarray(np.repeat(S[:1], B, axis=0)), x, b, True)
    assert np.abs(y_shared - y_batched).max() <= 1e-6 * max(1.0, np.abs(y_batched).max())
