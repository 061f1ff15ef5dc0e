"""Lockstep model of the quad step machines of the bucket reduction (gmsm_quad.h: k_combine_q and its work-efficient twin
k_combine_we, the level-1 combine of the reduction, and k_reduce2_q, level 2) over the additive group Z (add = +, dbl = *2, infinity = None): every step
is "all quads read their operands - barrier - compute and store - barrier", records may be one quad's destination and
another quad's source in the same step, and the result must be
    W_blk = sum W_t + L * sum_{t>=1} Suf_t,   S_blk = 2^prescale * sum S_t        (level 1)
    total = sum_j W_j + 2^log2span * sum_{j>=1} Suf_j                              (level 2)
- the identity of multiexp_jacobian.go:44-52 cut into segments. Pure Python; it pins the index arithmetic, the parking of
S_blk and the schedule of the prescaling doublings of the device code, which the GPU suite then runs on curve points."""
import random


def add(x, y):  # group law with infinity
    if y is None:
        return x
    if x is None:
        return y
    return x + y


def dbl(x):
    return None if x is None else 2 * x


def val(x):
    return 0 if x is None else x


def combine_q(N, log2L, prescale, S, W):
    """k_combine_q<U, INL, N>: quad j holds pair j."""
    S, W = list(S), list(W)
    lg = N.bit_length() - 1
    for s in range(lg):  # suffix scan, in place
        d = 1 << s
        loaded = [(j, S[j], S[j + d]) for j in range(N) if j + d < N]      # quad_add_load of every active quad
        for j, x, y in loaded:                                               # after the barrier
            S[j] = add(x, y)
    park = S[0]
    S[0] = None
    dbl_left = prescale
    n_tree, n_fin = lg, log2L + 1
    for s in range(n_tree + n_fin):
        loaded, fin_dbl = [], False
        for j in range(N):
            upper = j >= N // 2
            jj = j - N // 2 if upper else j
            arr = W if upper else S
            if s < n_tree:
                d = N >> (s + 1)
                if d >= 1 and jj < d:
                    loaded.append((arr, jj, arr[jj], arr[jj + d]))
            elif j == 0:
                step = s - n_tree
                if step < log2L:
                    fin_dbl = True
                else:
                    loaded.append((W, 0, W[0], S[0]))
            if j == N - 1 and s == 0:
                assert upper and jj < (N >> 1), "quad N-1 is busy in the first tree step"
        for arr, jj, x, y in loaded:
            arr[jj] = add(x, y)
        if fin_dbl:
            S[0] = dbl(S[0])
        if s >= 1 and dbl_left > 0:  # the doubler (quad N-1)
            jj = N // 2 - 1
            assert s >= n_tree or jj >= (N >> (s + 1)), "the doubler must be free"
            park = dbl(park)
            dbl_left -= 1
    assert dbl_left == 0
    return park, W[0]


def combine_we(log2L, prescale, S, W):
    """k_combine_we (N = 64): work-efficient form - pair sums per index bit, the odd elements' trees in place, then
    U = sum_l 2^l M_l by three two-term pairs. Same result as combine_q with 3.4 instead of 8 additions per pair."""
    N = 64
    S, W = list(S), list(W)
    for s in range(1, 7):
        g = 32 >> (s - 1)
        loaded = []
        for q in range(N):
            G, i = q // g, q % g
            if G > s:
                continue
            if G == 0:
                l = s - 1
                loaded.append((S, (2 * i) << l, S[(2 * i) << l], S[(2 * i + 1) << l]))
            elif G < s:
                l = G - 1
                loaded.append((S, (2 * i + 1) << l, S[(2 * i + 1) << l], S[(2 * (i + g) + 1) << l]))
            else:
                loaded.append((W, i, W[i], W[i + g]))
        assert len(loaded) == (s + 1) * g <= N
        dests = [(id(arr), k) for arr, k, _, _ in loaded]
        assert len(set(dests)) == len(dests), "two tasks write one record"
        for arr, k, x, y in loaded:
            arr[k] = add(x, y)
    park = S[0]
    dbl_left = prescale
    m = lambda l: 1 << l  # slot of M_l
    tail = [  # (doublings, additions) of each tail step on the slots S[1], S[2], S[4], S[8], S[16], S[32]
        ([m(1), m(3), m(5)], []),
        ([], [(m(0), m(1)), (m(2), m(3)), (m(4), m(5))]),
        ([m(2), m(4)], []),
        ([m(2), m(4)], []),
        ([m(4)], [(m(0), m(2))]),
        ([m(4)], []),
        ([], [(m(0), m(4))]),
    ]
    tail += [([m(0)], [])] * log2L
    for dbls, adds in tail + [([], [])]:
        last = (dbls, adds) == ([], [])
        loaded = [(x, S[x], S[y]) for x, y in adds]
        assert not (set(dbls) & {x for x, _ in adds}) and not (set(dbls) & {y for _, y in adds})
        for x, vx, vy in loaded:
            S[x] = add(vx, vy)
        for x in dbls:
            S[x] = dbl(S[x])
        if last:
            W[0] = add(W[0], S[1])
        if dbl_left > 0:
            park = dbl(park)
            dbl_left -= 1
    assert dbl_left == 0, "the prescaling doublings must fit into the tail"
    return park, W[0]


def reduce2_q(active, nblocks1, log2span, S, W):
    """k_reduce2_q: quad j holds level-1 block j (j < nblocks1, the rest infinity)."""
    S = [S[j] if j < nblocks1 else None for j in range(active)]
    W = [W[j] if j < nblocks1 else None for j in range(active)]
    d = 1
    while d < active:
        loaded = [(j, S[j], S[j + d]) for j in range(active) if j + d < active]
        for j, x, y in loaded:
            S[j] = add(x, y)
        d <<= 1
    S[0] = None
    for _ in range(log2span):
        S = [dbl(x) for x in S]
    W = [add(W[j], S[j]) for j in range(active)]
    d = active >> 1
    while d >= 1:
        loaded = [(j, W[j], W[j + d]) for j in range(d)]
        for j, x, y in loaded:
            W[j] = add(x, y)
        d >>= 1
    return W[0]


def test_combine_q_identity():
    rng = random.Random(20260925)
    for N in (16, 32, 64, 128):
        lg = N.bit_length() - 1
        for log2L in (1, 2, 3, 4, 8):
            for prescale in (0, log2L + lg):  # log2span = log2L + log2 N, as the host passes it
                for _ in range(10):
                    S = [rng.randrange(1, 1 << 40) if rng.random() < 0.8 else None for _ in range(N)]
                    W = [rng.randrange(1, 1 << 40) if rng.random() < 0.9 else None for _ in range(N)]
                    s_blk, w_blk = combine_q(N, log2L, prescale, S, W)
                    suf = [sum(val(x) for x in S[t:]) for t in range(N)]
                    assert val(w_blk) == sum(val(x) for x in W) + (1 << log2L) * sum(suf[1:])
                    assert val(s_blk) == sum(val(x) for x in S) << prescale


def test_combine_we_identity():
    rng = random.Random(31)
    N = 64
    for log2L in (1, 2, 3, 4, 8):
        for prescale in (0, log2L + 6):
            for _ in range(20):
                S = [rng.randrange(1, 1 << 40) if rng.random() < 0.8 else None for _ in range(N)]
                W = [rng.randrange(1, 1 << 40) if rng.random() < 0.9 else None for _ in range(N)]
                s_blk, w_blk = combine_we(log2L, prescale, S, W)
                suf = [sum(val(x) for x in S[t:]) for t in range(N)]
                assert val(w_blk) == sum(val(x) for x in W) + (1 << log2L) * sum(suf[1:])
                assert val(s_blk) == sum(val(x) for x in S) << prescale


def test_reduce2_q_identity():
    rng = random.Random(7)
    for active in (2, 4, 16, 64):
        for nblocks1 in {1, 2, active // 2 + 1, active} & set(range(1, active + 1)):
            for log2span in (0, 3, 11):
                S = [rng.randrange(1, 1 << 40) if rng.random() < 0.8 else None for _ in range(active)]
                W = [rng.randrange(1, 1 << 40) for _ in range(active)]
                got = reduce2_q(active, nblocks1, log2span, S, W)
                suf = [sum(val(x) for x in S[j:nblocks1]) for j in range(nblocks1)]
                assert val(got) == sum(val(x) for x in W[:nblocks1]) + (1 << log2span) * sum(suf[1:])
