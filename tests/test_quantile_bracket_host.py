"""The sample-bracket front end of the two-workgroup quantile kernel (csrc/sampler.hip: k_quantile_mask_multi), restated in numpy step by
step - sample positions, 2048-bin histogram levels over the sample, bracket [A, B], exact counts, the decision between the candidate
select and the fallback - and held against the oracle's sort-based quantile on distributions the GPU tests use and a few nastier ones.
What it pins on the CPU: (1) whenever the exact counts accept the bracket, the ranks lo and hi lie inside it, so selecting among the
candidates gives the oracle's q bit for bit; (2) the rule never depends on luck - a bracket that misses is always detected by the
counts; (3) on stationary continuous data of the sizes the kernel serves the bracket is hit and holds well under the 12 288 candidate slots
(a sample whose scale drifts over forty binades along the row - the GPU tests' `range` case - may miss: the fallback's business)."""
import math

import numpy as np
import pytest

QBINS, QM_S4, QM_CAP = 2048, 768, 12288


def f2key(f):
    b = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    return np.where(b & np.uint32(0x80000000), ~b, b | np.uint32(0x80000000)).astype(np.uint32)


def key2f(k):
    k = np.uint32(k)
    b = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else ~k
    return np.array([b], dtype=np.uint32).view(np.float32)[0]


def rank_split(p, n):
    pos = np.float32(p) * np.float32(n - 1)
    lo = int(math.floor(pos))
    hi = min(lo + 1, n - 1) if pos > lo else lo
    return lo, hi, np.float32(pos - np.float32(lo))


def bracket(u, lo):
    """Steps 1 - 2 of the kernel: the bracket every workgroup of a sample computes from the common 3072-key sample."""
    n = u.size
    n4 = n // 4
    st4 = n4 // QM_S4
    i = np.arange(QM_S4)
    f = i * st4 + ((i * 7) & ((1 << (st4.bit_length() - 1)) - 1))
    sk = f2key(u.reshape(n4, 4)[f].reshape(-1)).astype(np.int64)
    S = 4 * QM_S4
    pr = lo / n
    delta = int(math.ceil(5.0 * math.sqrt(pr * (1 - pr) * S))) + 8
    rs = lo * S // n
    open_a, open_b = rs < delta, rs + delta + 1 >= S - 1
    ra = 0 if open_a else rs - delta
    rb = S - 1 if open_b else rs + delta + 1
    cap_s = rb - ra + 1 + 96
    base, span = int(sk.min()), int(sk.max() - sk.min())
    for _level in range(4):
        shift = (span.bit_length() - 11) if span >= QBINS else 0
        d = sk - base
        c = (d >= 0) & (d <= span)
        hist = np.bincount(d[c] >> shift, minlength=QBINS)
        cum = np.cumsum(hist)
        bin_a, bin_b = int(np.searchsorted(cum, ra, side="right")), int(np.searchsorted(cum, rb, side="right"))
        ex_a, in_b = int(cum[bin_a] - hist[bin_a]), int(cum[bin_b])
        lo_off, hi_off = bin_a << shift, min(((bin_b + 1) << shift) - 1, span)
        nspan = hi_off - lo_off
        stuck = nspan == span
        base, span, ra, rb = base + lo_off, nspan, ra - ex_a, rb - ex_a
        if in_b - ex_a <= cap_s or shift == 0 or stuck:
            break
    return (0 if open_a else base), (0xFFFFFFFE if open_b else base + span)


def multi_select(u, p):
    """-> (q, took_the_candidate_path, candidates)."""
    n = u.size
    lo, hi, w = rank_split(p, n)
    A, B = bracket(u, lo)
    k = f2key(u).astype(np.int64)
    below = int((k < A).sum())
    cand = np.sort(k[(k >= A) & (k <= B)])
    # the kernel's per-workgroup halves (interleaved 16-KB stripes) each have to fit their region too; the total is what matters here
    inside = cand.size <= QM_CAP and below <= lo and hi < below + cand.size
    if not inside:
        return None, False, cand.size
    a = key2f(cand[lo - below])
    b = key2f(cand[hi - below])
    d = np.float32(b - a)
    q = np.float32(a + w * d) if w < np.float32(0.5) else np.float32(b - d * (np.float32(1) - w))
    return q, True, cand.size


def cases(n, seed):
    g = np.random.default_rng(seed)
    base = g.standard_normal(n).astype(np.float32)
    return {"chi2": base ** 2, "signed": base, "range": base * np.logspace(-20, 20, n).astype(np.float32), "cluster": (1.0 + base * 1e-6).astype(np.float32),
            "walk": np.cumsum(base).astype(np.float32), "ties": (np.round(base * 3) / 3).astype(np.float32), "const": np.full(n, 0.25, np.float32),
            "mostly_zero": np.where(g.random(n) < 0.7, 0.0, base ** 2).astype(np.float32),
            "image": (np.abs(np.sin(np.arange(n) * 0.013)) + 0.05 * base ** 2).astype(np.float32)}


@pytest.mark.parametrize("n", [3 * 128 * 128, 20000, 32768, 65536, 16388])
def test_bracket_is_exact_whenever_it_is_accepted(n):
    from oracle import guidance
    hit = {}
    for name, u in cases(n, seed=n).items():
        for p in (0.9, 0.5, 0.0, 1.0, 0.99):
            q, took, ncand = multi_select(u, p)
            qref = guidance.quantile_linear(u.reshape(1, -1), p)[0]
            if took:
                # (bit for bit, except that a zero may come out with either sign: the keys order -0.0 before +0.0, a sort does not)
                assert q == qref and (q == 0 or np.float32(q).view(np.uint32) == np.float32(qref).view(np.uint32)), (name, p, q, qref)
            hit[(name, p)] = (took, ncand)
    # continuous values at the percentile the reference uses: the bracket is hit, with room to spare
    for name in ("chi2", "signed", "image"):
        took, ncand = hit[(name, 0.9)]
        assert took and ncand <= QM_CAP * 3 // 4, (name, ncand)
    # constant / heavily tied samples cannot fit 12 288 slots around the target: the counts send them to the fallback
    assert not hit[("const", 0.9)][0]
