"""Philox4x32-10 counter RNG and the draw algorithms built on it (oracle side).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws from numpy's
global MT19937 stream (LiteRateForward.py:405-409), which cannot be reproduced
per chain on a GPU; the device engine therefore defines its own stream:

    words = philox4x32_10(ctr=(it_lo, it_hi, purpose, index), key=(seed, chain))
    u_a   = ((w0 >> 5) * 2**26 + (w1 >> 6)) / 2**53        in [0, 1)
    u_b   = ((w2 >> 5) * 2**26 + (w3 >> 6)) / 2**53

so every draw is addressed by (iteration, purpose, index) and no draw depends
on how many draws came before it.  This file restates that scheme in
numpy/pure Python so the oracle MCMC can be fed the identical randomness the
HIP kernel (literate_amd/csrc/lr_device.h) uses.  Algorithm: Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3" (SC'11), 10 rounds.
"""
import math

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF

# purposes (must match literate_amd/csrc/lr_device.h)
P_MOVE = 0      # idx 0 -> r[0], r[1] (u_a, u_b)
P_MULT = 1      # idx j -> (binomial-uniform, multiplier-uniform) for element j
P_TIMES = 2     # idx 0 -> (choice, random)
P_RJ = 3        # idx 0 -> (r'[0], r'[1]); idx 1 -> (choice, delta)
P_BETA_A = 4    # gamma(a) variate, attempts
P_BETA_B = 5
P_GIBBS_POI = 6
P_GIBBS_L = 7
P_GIBBS_M = 8
P_ACCEPT = 9    # idx 0 -> u_a
P_INIT = 10     # idx 0.. gamma(2,2) initial rates (attempts), L at base 0, M at base 64

GAMMA_MAX_ATTEMPTS = 32


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """One Philox4x32-10 block.  All arguments are python ints < 2**32."""
    for r in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def _to_double(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


class Stream:
    """Addressable uniform source for one chain: u(it, purpose, idx) -> (u_a, u_b)."""

    def __init__(self, seed, chain):
        self.k0 = seed & MASK
        self.k1 = chain & MASK

    def pair(self, it, purpose, idx):
        w = philox4x32_10(it & MASK, (it >> 32) & MASK, purpose & MASK, idx & MASK,
                          self.k0, self.k1)
        return _to_double(w[0], w[1]), _to_double(w[2], w[3])

    def normal(self, it, purpose, idx):
        """Box-Muller on the pair at (it, purpose, idx): sqrt(-2 log(1-u_a)) cos(2 pi u_b)."""
        ua, ub = self.pair(it, purpose, idx)
        return math.sqrt(-2.0 * math.log(1.0 - ua)) * math.cos(2.0 * math.pi * ub)

    def gamma(self, it, purpose, base, shape):
        """Standard Gamma(shape>=1) by Marsaglia-Tsang (2000); attempt a uses
        idx base+2a (normal) and base+2a+1 (acceptance uniform u_a)."""
        d = shape - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        for a in range(GAMMA_MAX_ATTEMPTS):
            x = self.normal(it, purpose, base + 2 * a)
            t = 1.0 + c * x
            v = t * t * t
            if v <= 0.0:
                continue
            u, _ = self.pair(it, purpose, base + 2 * a + 1)
            if u <= 0.0:
                return d * v
            if math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v):
                return d * v
        return d


# ---- vectorised block (numpy uint64 arithmetic) for the lineage simulator -------------------------
P_SIM = 24      # simulator: key = (seed, lineage slot), counter = (step, P_SIM, 0) -> u_a decides birth / death


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """philox4x32_10 on numpy arrays (any broadcastable shapes); returns four uint64 arrays < 2**32."""
    import numpy as np
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    k0, k1 = np.asarray(k0, dtype=np.uint64), np.asarray(k1, dtype=np.uint64)
    m = np.uint64(MASK)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & m
        hi1, lo1 = p1 >> np.uint64(32), p1 & m
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & m, lo1, (hi0 ^ c3 ^ k1) & m, lo0
        k0 = (k0 + np.uint64(W0)) & m
        k1 = (k1 + np.uint64(W1)) & m
    return c0, c1, c2, c3


def uniform_a_np(it, purpose, idx, k0, k1):
    """u_a of Stream(k0, k1).pair(it, purpose, idx), vectorised over any of the arguments."""
    import numpy as np
    it = np.asarray(it, dtype=np.uint64)
    w0, w1, _, _ = philox4x32_10_np(it & np.uint64(MASK), it >> np.uint64(32), purpose, idx, k0, k1)
    return ((w0 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (w1 >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0
