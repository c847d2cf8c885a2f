"""Pure-Python restatement of the exact negacyclic transform of MKT_ARITH_EXACT (test infrastructure): the textbook
Cooley-Tukey / Gentleman-Sande pair over Z_p[X]/(X^N + 1) for the two 30-bit primes p1 = 131063 * 2^13 + 1 and
p2 = 131066 * 2^13 + 1 side by side, with the table psi_rev[k] = psi^bitrev(k) -- the integer twin of the reference's network
(src/ring/fft.jl:105-209, Psi[m + i]).  A point is the pair (x mod p1, x mod p2) packed as x1 | x2 << 32; the integer a
pair stands for is the one of least magnitude mod P = p1 p2 (Chinese remainder theorem)."""
PRIMES = (131063 * 2**13 + 1, 131066 * 2**13 + 1)
P = PRIMES[0] * PRIMES[1]


def tables(N, p):
    logN = N.bit_length() - 1
    g = 2
    while pow(g, (p - 1) // 2, p) != p - 1:                      # the smallest quadratic non-residue: psi^N = -1
        g += 1
    psi = pow(g, (p - 1) // (2 * N), p)
    assert pow(psi, N, p) == p - 1
    rev = lambda k: int(format(k, f"0{logN}b")[::-1], 2)
    psiinv = pow(psi, p - 2, p)
    return [pow(psi, rev(k), p) for k in range(N)], [pow(psiinv, rev(k), p) for k in range(N)], pow(N, p - 2, p)


def _fwd1(z, p):
    N = len(z)
    psi_rev, _, _ = tables(N, p)
    t, m = N, 1
    while m < N:
        t //= 2
        for i in range(m):
            S = psi_rev[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                U, V = z[j], z[j + t] * S % p
                z[j], z[j + t] = (U + V) % p, (U - V) % p
        m *= 2
    return z


def _inv1(z, p):
    N = len(z)
    _, psiinv_rev, ninv = tables(N, p)
    t, m = 1, N
    while m > 1:
        h, j1 = m // 2, 0
        for i in range(h):
            S = psiinv_rev[h + i]
            for j in range(j1, j1 + t):
                U, V = z[j], z[j + t]
                z[j], z[j + t] = (U + V) % p, (U - V) * S % p
            j1 += 2 * t
        t *= 2
        m //= 2
    return [v * ninv % p for v in z]


def pack(r1, r2):
    return [a | (b << 32) for a, b in zip(r1, r2)]


def unpack(z):
    return [int(v) & 0xFFFFFFFF for v in z], [int(v) >> 32 for v in z]


def fwd(a, W):
    """a: N ring words, read as signed W-bit integers -> N packed residue pairs in the transform's (bit-reversed) order"""
    s = [int(x) - (1 << W) if int(x) >> (W - 1) else int(x) for x in a]
    return pack(*[_fwd1([v % p for v in s], p) for p in PRIMES])


def pmul(x, y):
    """pointwise product of two transformed polynomials"""
    (x1, x2), (y1, y2) = unpack(x), unpack(y)
    return pack([a * b % PRIMES[0] for a, b in zip(x1, y1)], [a * b % PRIMES[1] for a, b in zip(x2, y2)])


def inv(z, W):
    r1, r2 = unpack(z)
    r1, r2 = _inv1(r1, PRIMES[0]), _inv1(r2, PRIMES[1])
    c = pow(PRIMES[0], PRIMES[1] - 2, PRIMES[1])
    out = []
    for a, b in zip(r1, r2):
        v = a + PRIMES[0] * ((b - a) * c % PRIMES[1])
        if v > P // 2:
            v -= P
        out.append(v % (1 << W))
    return out
