"""Philox4x32-10 counter RNG: the sampler's uniforms.  Identical arithmetic in csrc/llm.hip (k_sample), so a
test can inject the same noise into the CPU oracle: counter = (seq, step, trial, 0), key = (seed_lo, seed_hi)."""

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32(c, k):
    c0, c1, c2, c3 = c
    k0, k1 = k
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniforms(seq, step, trial, seed):
    """Two float64 uniforms in [0,1) for (slot, loop step, EOS re-draw trial)."""
    r = philox4x32((seq, step, trial, 0), (seed & MASK, (seed >> 32) & MASK))
    u = lambda hi, lo: (((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)
    return u(r[0], r[1]), u(r[2], r[3])
