"""The arithmetic behind node2vec's running sums (n2v_kernels.h: WaveSumsVec) restated
in numpy (tools/n2v_binade_model.py) equals the sequential f32 adds of the reference's
BuildWeights / RandomSelect pair (tf_euler/kernels/random_walk_op.cc:83-168) on every
input class the kernel meets: random and dyadic weights (ties), zeros, binade crossings,
tiny and zero carries, negative entries."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import n2v_binade_model as M  # noqa: E402

f32 = np.float32


def test_scheme_equals_sequential_sums():
    rng = np.random.default_rng(5)
    used_integer_path = 0
    for trial in range(1500):
        mode = trial % 6
        n = int(rng.integers(1, 257))
        if mode == 0:
            d = (rng.random(n) * 7.5 + 0.5).astype(f32) / f32(4)
        elif mode == 1:
            d = (rng.integers(0, 64, n) / 8).astype(f32)                 # ties
        elif mode == 2:
            d = (rng.random(n) * 3).astype(f32)
            d[rng.random(n) < 0.2] = 0
        elif mode == 3:
            d = rng.integers(0, 1 << 12, n).astype(f32) * f32(2.0 ** -int(rng.integers(0, 14)))
        elif mode == 4:
            d = (rng.standard_normal(n) * 2).astype(f32)                 # negative entries
        else:
            d = M.hub_like_list(rng, n, 0.05)
        carry = f32(0) if trial % 7 == 0 else f32(rng.random() * 2.0 ** int(rng.integers(-40, 22)))
        want = M.sequential(carry, d)
        got, adds = M.scheme(carry, d)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (trial, mode, carry)
        used_integer_path += adds < n
        got4, _ = M.scheme(carry, d, max_restarts=4)
        assert np.array_equal(got4.view(np.uint32), want.view(np.uint32))
    assert used_integer_path > 700          # the scheme is not vacuous


def test_hub_list_mostly_integer_sums():
    rng = np.random.default_rng(2)
    wq = M.hub_like_list(rng, 20000, 0.0)
    carry, chained = f32(0), 0
    for j in range(0, len(wq), 256):
        ch = wq[j:j + 256]
        sums, adds = M.scheme(carry, ch, max_restarts=4)
        assert np.array_equal(sums, M.sequential(carry, ch))
        chained += adds > 4
        carry = sums[-1]
    assert chained <= 8                     # the start of the list crosses a binade per chunk
