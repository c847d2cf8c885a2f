"""Pure-Python restatement of the exact negacyclic transform of MKT_ARITH_EXACT (test infrastructure): the textbook
Cooley-Tukey / Gentleman-Sande pair over Z_p[X]/(X^N + 1), p = 2^64 - 2^32 + 1, with the table psi_rev[k] =
psi^bitrev(k) -- the integer twin of the reference's network (src/ring/fft.jl:105-209, Psi[m + i])."""
P = 2**64 - 2**32 + 1


def tables(N):
    logN = N.bit_length() - 1
    psi = pow(7, (P - 1) // (2 * N), P)
    assert pow(psi, N, P) == P - 1
    rev = lambda k: int(format(k, f"0{logN}b")[::-1], 2)
    psiinv = pow(psi, P - 2, P)
    return [pow(psi, rev(k), P) for k in range(N)], [pow(psiinv, rev(k), P) for k in range(N)], pow(N, P - 2, P)


def fwd(a, W):
    """a: N ring words, read as signed W-bit integers -> N residues in the transform's (bit-reversed) order"""
    N = len(a)
    psi_rev, _, _ = tables(N)
    z = [(int(x) - (1 << W) if int(x) >> (W - 1) else int(x)) % P for x in a]
    t, m = N, 1
    while m < N:
        t //= 2
        for i in range(m):
            S = psi_rev[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                U, V = z[j], z[j + t] * S % P
                z[j], z[j + t] = (U + V) % P, (U - V) % P
        m *= 2
    return z


def inv(z, W):
    N = len(z)
    _, psiinv_rev, ninv = tables(N)
    z = [int(v) for v in z]
    t, m = 1, N
    while m > 1:
        h, j1 = m // 2, 0
        for i in range(h):
            S = psiinv_rev[h + i]
            for j in range(j1, j1 + t):
                U, V = z[j], z[j + t]
                z[j], z[j + t] = (U + V) % P, (U - V) * S % P
            j1 += 2 * t
        t *= 2
        m //= 2
    out = []
    for v in z:
        v = v * ninv % P
        if v > P // 2:
            v -= P
        out.append(v % (1 << W))
    return out
