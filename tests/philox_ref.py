"""numpy restatement of the theta kernel's counter-based normal draws (csrc/vihds_elbo.hip: philox4x32_10,
philox_normal), used by the tests as the checker."""
import numpy as np


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy Philox4x32-10 (Salmon et al. SC'11) on uint64 arrays holding 32-bit words."""
    m32 = np.uint64(0xFFFFFFFF)
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    k = [np.asarray(x, dtype=np.uint64) for x in (k0, k1)]
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & m32]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return c


def expected_kernel_normals(B, S, P, seed, step, S_total=None, s_off=0):
    """The draws vihds_theta_fwd makes with opts.rng = {seed lo, seed hi, step, .} (include/vihds_hip.h)."""
    S_total = S if S_total is None else S_total
    b, s, p = np.meshgrid(np.arange(B), np.arange(S), np.arange(P), indexing="ij")
    idx = (b * S_total + s_off + s).astype(np.uint64)
    r = philox4x32_10(idx, (p >> 2).astype(np.uint64), np.full_like(idx, step), np.zeros_like(idx),
                       np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32))
    q = p & 3
    x = np.where(q & 2, r[2], r[0]).astype(np.float32)
    y = np.where(q & 2, r[3], r[1]).astype(np.float32)
    u1 = np.minimum((x + np.float32(0.5)) * np.float32(2.0 ** -32), np.float32(0.99999994))
    u2 = (y + np.float32(0.5)) * np.float32(2.0 ** -32)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64)))
    ang = 2.0 * np.pi * u2.astype(np.float64)
    return np.where(q & 1, rad * np.sin(ang), rad * np.cos(ang)).astype(np.float32)
