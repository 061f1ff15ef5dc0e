"""Lockstep model of the level-1 combine programs of the bucket reduction (gmsm_kernels.h: reduce_program's combine
phases and k_reduce_combine_lds) over the additive group Z (add = +, dbl = *2, infinity = 0): the step machine - suffix
scan in place, the two trees, the parked block sum with its prescaling doublings, the finish - must produce
W_blk = sum W_t + L * sum_{t>=1} Suf_t and S_blk = 2^prescale * sum S_t (the identity of multiexp_jacobian.go:44-52
cut into segments). Pure Python; it pins the index arithmetic of the device code, which the GPU suite then runs."""
import random


def combine_lds(TPB, log2L, prescale, S, W):
    SUF, TOT, PARK = list(S), list(W), [0]
    lg = TPB.bit_length() - 1
    n_scan = n_tree = lg
    total = n_scan + n_tree + log2L + 1
    dbl_left = prescale
    for s in range(total):
        ops = []
        for t in range(TPB):
            upper = t >= TPB // 2
            tt = t - TPB // 2 if upper else t
            doubler = prescale != 0 and t == TPB // 4
            X = Y = 0
            dest, do_dbl = -1, False
            if s < n_scan:
                d = 1 << s
                X = SUF[t]
                if t + d < TPB:
                    Y = SUF[t + d]
                dest = 0
            elif s < n_scan + n_tree:
                step = s - n_scan
                d = TPB >> (step + 1)
                arr = TOT if upper else SUF
                if tt < d:
                    if not (step == 0 and not upper and tt == 0):
                        X = arr[tt]
                    Y = arr[tt + d]
                    dest = 1
                elif doubler and step >= 1 and dbl_left > 0:
                    X, do_dbl, dest = PARK[0], True, 4
                    dbl_left -= 1
            else:
                step = s - n_scan - n_tree
                if t == 0:
                    if step < log2L:
                        X, do_dbl, dest = SUF[0], True, 2
                    else:
                        X, Y, dest = TOT[0], SUF[0], 3
                elif doubler and dbl_left > 0:
                    X, do_dbl, dest = PARK[0], True, 4
                    dbl_left -= 1
            ops.append((t, upper, tt, X, Y, dest, do_dbl))
        for t, upper, tt, X, Y, dest, do_dbl in ops:  # after the barrier: compute, write back
            X = 2 * X if do_dbl else (X + Y if dest >= 0 else X)
            if dest == 0:
                SUF[t] = X
                if s + 1 == n_scan and t == 0:
                    PARK[0] = X
            elif dest == 1:
                (TOT if upper else SUF)[tt] = X
            elif dest == 2:
                SUF[0] = X
            elif dest == 3:
                TOT[0] = X
            elif dest == 4:
                PARK[0] = X
    assert dbl_left == 0
    return PARK[0], TOT[0]


def test_combine_program_identity():
    rng = random.Random(20260925)
    for TPB in (128, 256):
        for log2L in (1, 3, 4, 5, 8):
            for prescale in (0, log2L + TPB.bit_length() - 1):  # log2span = log2L + log2 TPB, as the host passes it
                for _ in range(10):
                    S = [rng.randrange(1 << 40) if rng.random() < 0.8 else 0 for _ in range(TPB)]
                    W = [rng.randrange(1 << 40) for _ in range(TPB)]
                    s_blk, w_blk = combine_lds(TPB, log2L, prescale, S, W)
                    suf = [sum(S[t:]) for t in range(TPB)]
                    assert w_blk == sum(W) + (1 << log2L) * sum(suf[1:])
                    assert s_blk == sum(S) << prescale
