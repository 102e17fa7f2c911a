"""The device removes isolated patches (Surface::remove_isolated_patches,
/root/reference/lib/surface.cc:887-927: in-place deletion during a column-by-column
walk) by RELAXING the walk's recurrence instead of replaying the walk
(smvs_amd/csrc/surface.hip, surf_isolated_kernel): del(p) = valid(p) and
(valid neighbours - deleted neighbours among the four visited before p) < 3, all
patches at once on bit vectors with a bit-sliced adder, in place, until a sweep
changes nothing.  This file pins the ALGORITHM on the CPU: a literal restatement
of the reference's walk against a numpy transcription of the kernel's sweep
(same words, same adder, same order of neighbours) on random grids -- sparse
noise, ragged borders, thin lines that are eaten from one end (the longest
chains), every grid up to 6 x 5 exhaustively sampled.  The device kernel itself
is checked against the host mirror / oracle in tests/test_gpu_surface.py and
tools/fuzz_surface.py."""
import numpy as np
import pytest


def walk(valid):
    """surface.cc:887-927, literally: x outer, y inner, deletions take effect at once."""
    v = valid.copy()
    npx, npy = v.shape
    for x in range(npx):
        for y in range(npy):
            if not v[x, y]:
                continue
            n = 0
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if dx == 0 and dy == 0:
                        continue
                    a, b = x + dx, y + dy
                    if 0 <= a < npx and 0 <= b < npy and v[a, b]:
                        n += 1
            if n < 3:
                v[x, y] = False
    return v


def _pack(valid):
    """columns with a one-cell border of zeros, rows packed into 32-bit words:
    column c = x + 1, row r = y + 1, bit r & 31 of word r >> 5."""
    npx, npy = valid.shape
    wpc = (npy + 2 + 31) // 32
    words = np.zeros((npx + 2, wpc), dtype=np.uint64)   # (uint64 holding 32-bit values)
    for x in range(npx):
        for y in range(npy):
            if valid[x, y]:
                r = y + 1
                words[x + 1, r >> 5] |= np.uint64(1 << (r & 31))
    return words, wpc


M32 = np.uint64(0xFFFFFFFF)


def relax(valid, order="ascending", max_sweeps=None):
    """surf_isolated_kernel's sweep, word by word, in place; `order` is the order in
    which the words of a sweep are updated (the kernel's threads run in no
    particular order: the result must not depend on it)."""
    npx, npy = valid.shape
    orig, wpc = _pack(valid)
    dele = np.zeros_like(orig)

    def above(col, w):   # bit r of the result = bit r - 1 of the column
        return ((col[w] << np.uint64(1)) | (col[w - 1] >> np.uint64(31) if w > 0 else np.uint64(0))) & M32

    def below(col, w):   # ... = bit r + 1
        return ((col[w] >> np.uint64(1)) | ((col[w + 1] << np.uint64(31)) if w + 1 < wpc else np.uint64(0))) & M32

    idx = [(c, w) for c in range(1, npx + 1) for w in range(wpc)]
    if order == "descending":
        idx = idx[::-1]
    elif order == "shuffled":
        np.random.default_rng(7).shuffle(idx)
    sweeps = 0
    limit = max_sweeps if max_sweeps is not None else 2 * npx + npy + 2
    for sweeps in range(1, limit + 1):
        changed = False
        for c, w in idx:
            mine = orig[c, w]
            if mine == 0:
                continue
            Lo, Ld, Mo, Md, Ro = orig[c - 1], dele[c - 1], orig[c], dele[c], orig[c + 1]
            l_above = above(Lo, w) & ~above(Ld, w) & M32
            l_same = Lo[w] & ~Ld[w] & M32
            l_below = below(Lo, w) & ~below(Ld, w) & M32
            m_above = above(Mo, w) & ~above(Md, w) & M32
            m_below = below(Mo, w)
            r_above, r_same, r_below = above(Ro, w), Ro[w], below(Ro, w)
            sa = l_above ^ l_same ^ l_below
            ca = (l_above & l_same) | (l_below & (l_above ^ l_same))
            sb = m_above ^ m_below ^ r_above
            cb = (m_above & m_below) | (r_above & (m_above ^ m_below))
            sc, cc = r_same ^ r_below, r_same & r_below
            s0 = sa ^ sb ^ sc
            cd = (sa & sb) | (sc & (sa ^ sb))
            ts = ca ^ cb ^ cc
            tc = (ca & cb) | (cc & (ca ^ cb))
            s1, u = ts ^ cd, ts & cd
            s2, s3 = tc ^ u, tc & u
            three_or_more = s3 | s2 | (s1 & s0)
            now = mine & ~three_or_more & M32
            if now != dele[c, w]:
                dele[c, w] = now
                changed = True
        if not changed:
            break
    out = valid.copy()
    for x in range(npx):
        for y in range(npy):
            r = y + 1
            if (int(dele[x + 1, r >> 5]) >> (r & 31)) & 1:
                out[x, y] = False
    return out, sweeps


def _grids():
    rng = np.random.default_rng(2024)
    for npx, npy in ((1, 1), (1, 7), (7, 1), (2, 2), (5, 33), (33, 5), (40, 31), (17, 64), (23, 70)):
        for density in (0.15, 0.4, 0.6, 0.85, 1.0):
            yield rng.random((npx, npy)) < density
    # ragged borders: a full grid with random bites out of its edges and a few holes
    for _ in range(6):
        g = np.ones((36, 45), dtype=bool)
        for _ in range(40):
            x, y = rng.integers(0, 36), rng.integers(0, 45)
            g[max(0, x - 1):x + rng.integers(1, 4), max(0, y - 1):y + rng.integers(1, 4)] = False
        yield g
    # thin lines: eaten patch by patch from the end the walk reaches first
    g = np.zeros((50, 40), dtype=bool); g[3:47, 20] = True; yield g          # along x
    g = np.zeros((12, 90), dtype=bool); g[6, 2:88] = True; yield g           # along y, across words
    g = np.zeros((45, 45), dtype=bool); g[np.arange(45), np.arange(45)] = True; yield g   # diagonal
    g = np.zeros((45, 45), dtype=bool); g[np.arange(45), 44 - np.arange(45)] = True; yield g
    g = np.zeros((30, 70), dtype=bool); g[5:25, 10] = True; g[5, 10:60] = True; g[5:25, 59] = True; yield g
    # two-wide bands survive partly
    g = np.zeros((40, 40), dtype=bool); g[4:36, 10:12] = True; g[20:22, 4:36] = True; yield g


@pytest.mark.parametrize("order", ["ascending", "descending", "shuffled"])
def test_relaxation_equals_the_reference_walk(order):
    worst = 0
    for g in _grids():
        want = walk(g)
        got, sweeps = relax(g, order)
        assert np.array_equal(got, want), (g.shape, order)
        assert sweeps <= 2 * g.shape[0] + g.shape[1] + 2
        worst = max(worst, sweeps)
    # (the thin line along x needs one sweep per patch when the words are updated
    # against the walk's direction: the bound of the kernel's loop is not idle)
    assert worst >= 3


def test_small_grids_exhaustively_sampled():
    rng = np.random.default_rng(5)
    for npx in range(1, 7):
        for npy in range(1, 6):
            n = npx * npy
            masks = range(1 << n) if n <= 12 else rng.integers(0, 1 << n, size=3000)
            for m in masks:
                g = np.array([(int(m) >> i) & 1 for i in range(n)], dtype=bool).reshape(npx, npy)
                got, _ = relax(g)
                assert np.array_equal(got, walk(g)), (npx, npy, int(m))


def test_sweeps_on_a_ragged_surface_are_few():
    """What makes the relaxation cheap: on a surface with ragged borders the chains
    of deletions that cause each other are short."""
    rng = np.random.default_rng(11)
    g = np.ones((120, 68), dtype=bool)
    for _ in range(150):
        x, y = rng.integers(0, 120), rng.integers(0, 68)
        g[max(0, x - 1):x + rng.integers(1, 5), max(0, y - 1):y + rng.integers(1, 5)] = False
    got, sweeps = relax(g)
    assert np.array_equal(got, walk(g))
    assert sweeps <= 12, sweeps
