"""numpy restatement of the node2vec running-sum scheme (n2v_kernels.h: WaveSumsVec)
against the sequential f32 adds it replaces.

For a carry m * ulp in [2^e, 2^(e+1)) and carry + d below 2^(e+1), fl(carry + d) =
(m + n) * ulp with n = d / ulp rounded to nearest - the same n for every m unless d / ulp
ends in exactly .5 (a tie goes to the even m + n).  So inside one binade and without a
tie the sequential adds ARE an integer running sum over the mantissa; n comes out of the
adder itself (fl(2^e + d) has mantissa offset n) and d - (fl(2^e + d) - 2^e) is exact
and equals +-ulp/2 exactly on a tie.  An entry the integer sum cannot pass gets its sum
from one real add and the entries after it start over from there.

  python tools/n2v_binade_model.py        # the fallback rate on a hub-like list
"""
import numpy as np

f32 = np.float32


def sequential(carry, d):
    out = np.empty(len(d), f32)
    acc = f32(carry)
    for i, x in enumerate(d):
        acc = f32(acc + x)
        out[i] = acc
    return out


def _bits(x):
    return int(np.array([x], f32).view(np.uint32)[0])


def _float(b):
    return np.array([b], np.uint32).view(f32)[0]


def scheme(carry, d, max_restarts=None):
    """The kernel's scheme on one chunk: returns (sums, real adds used)."""
    n = len(d)
    out = np.empty(n, f32)
    start, carry, real_adds = 0, f32(carry), 0
    while start < n:
        cb = _bits(carry)
        e = cb >> 23
        problem = n
        if 30 <= e < 254:
            bb = cb & 0xFF800000
            B = _float(bb)
            half = _float(bb - (24 << 23))
            dd = d[start:]
            t = (B + dd).astype(f32)
            err = (dd - (t - B).astype(f32)).astype(f32)
            ok = (dd >= 0) & (np.abs(err) != half) & (t < f32(2) * B)
            nn = np.where(ok, t.view(np.uint32).astype(np.int64) - bb, 0)
            off = (cb - bb) + np.cumsum(nn)
            bad = np.nonzero(~ok | (off >= (1 << 23)))[0]
            stop = int(bad[0]) if len(bad) else len(dd)
            out[start:start + stop] = (np.uint32(bb) + off[:stop].astype(np.uint32)).view(f32)
            problem = start + stop
        else:
            problem = start
        if problem >= n:
            break
        before = carry if problem == start else out[problem - 1]
        carry = f32(f32(before) + d[problem])          # one real add
        out[problem] = carry
        real_adds += 1
        start = problem + 1
        if max_restarts is not None and real_adds > max_restarts:
            out[start:] = sequential(carry, d[start:])
            real_adds += n - start
            break
    return out, real_adds


def hub_like_list(rng, n, keep_frac, q=4.0):
    """weights as the metric graph has them: differences of a row's f32 running sums,
    divided by q except for a `keep_frac` of common neighbours"""
    x = rng.integers(0, 1 << 24, n).astype(f32) * f32(1 / 16777216)
    w = (f32(0.5) + f32(7.5) * x).astype(f32)
    prefix = sequential(f32(0), w)
    d = np.diff(np.concatenate([[f32(0)], prefix])).astype(f32)
    keep = rng.random(n) < keep_frac
    return np.where(keep, d, (d / f32(q)).astype(f32)).astype(f32)


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n, keep in ((100000, 0.0), (100000, 0.02), (10000, 0.02)):
        wq = hub_like_list(rng, n, keep)
        carry, chunks, chained = f32(0), 0, 0
        for j in range(0, n, 256):
            ch = wq[j:j + 256]
            sums, adds = scheme(carry, ch, max_restarts=4)
            assert np.array_equal(sums, sequential(carry, ch))
            chunks += 1
            chained += adds > 4
            carry = sums[-1]
        print("list of %d, %.0f %% kept: %d of %d chunks of 256 fall back to the add chain"
              % (n, keep * 100, chained, chunks))
