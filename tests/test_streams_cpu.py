"""The stream tables of the weight gradient (oracle/streams_ref.py, the CPU restatement of csrc/wgrad_streams.hip): what
the streamed kernel computes from them is the plain sum over the rule lists -- torchsparse's grad_weight loop
(convolution_backward_cuda: gw[k] = a[in_k]^T b[out_k]) -- for ragged, empty and skewed offsets, with and without a
spatial key."""
import numpy as np
import pytest

from oracle import streams_ref as S


def _rules(rng, n, sizes):
    pairs = []
    for s in sizes:
        out = np.sort(rng.permutation(n)[:s])
        pairs.append(np.stack([rng.integers(0, n, s), out], 1))
    return np.concatenate(pairs).astype(np.int32) if sum(sizes) else np.zeros((0, 2), np.int32)


def _plain(a, b, pairs, sizes):
    gw = np.zeros((len(sizes), a.shape[1], b.shape[1]))
    o = 0
    for k, s in enumerate(sizes):
        p = pairs[o:o + s]
        gw[k] = a[p[:, 0]].T @ b[p[:, 1]]
        o += s
    return gw


@pytest.mark.parametrize('case', ['ragged', 'one_offset', 'tiny', 'empty', 'keyed'])
def test_streams_compute_the_plain_sum(case):
    rng = np.random.default_rng(5)
    n, k, n_wg = 3000, 27, 256
    sizes = [int(v) for v in rng.integers(40, 700, k)]
    sizes[13] = n
    sizes[3] = 0
    sizes[0] = 7
    key, key_range = None, n
    if case == 'one_offset':
        sizes = [0] * k
        sizes[13] = n
    elif case == 'tiny':
        n = 37
        sizes = [int(v) for v in rng.integers(0, 5, k)]
        sizes[13] = n
    elif case == 'empty':
        sizes = [0] * k
    elif case == 'keyed':
        key_range = 977
        key = rng.integers(0, key_range, n)
    pairs = _rules(rng, n, sizes)
    sp, sd = S.build_streams(pairs, sizes, key, key_range, n, n_wg)
    assert sp.shape[0] == int(sd[2]) * 64 and sd[0] == n_wg and sd[1] == k
    # every rule exactly once, under its own offset's set
    live = (sp[:, 1] != S.PAD)
    assert int(live.sum()) == sum(sizes)
    a, b = rng.standard_normal((n, 8)), rng.standard_normal((n, 16))
    got = S.run_streams(a, b, sp, sd)
    want = _plain(a, b, pairs, sizes)
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_slots_cover_every_offset_with_at_most_two_per_workgroup():
    rng = np.random.default_rng(1)
    for _ in range(200):
        k = int(rng.integers(1, 28))
        counts = [int(v) for v in rng.integers(0, 5, k) * rng.integers(0, 100000, k)]
        lens = S.slot_lengths(counts, 64)
        assert sum(lens) <= 64 * S.UNIT
        assert all((c == 0) == (ln == 0) for c, ln in zip(counts, lens))
        assert all(ln >= S.UNIT for ln in lens if ln)          # hence no slot meets three offsets
