#!/usr/bin/env python3
"""Offline search for an LDS staging layout without bank conflicts for the in-transform exchanges.

Simulates ds_read_b128 / ds_write_b128 bank behaviour on gfx950 as documented in
/opt/skills/guides/MI355X_MICROARCH.md (LDS section): b128 reads are serviced in 4 groups of 16 lanes
{0-3,12-15,20-27},{4-11,16-19,28-31},{32-35,44-47,52-59},{36-43,48-51,60-63} over 64 banks (16 columns of
16 B); b128 writes in 8 groups of 8 contiguous lanes over 32 banks (8 columns).  A group costs as many LDS
cycles as the most loaded column has distinct addresses.
"""
import itertools
import sys

RGROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
           list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
           list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
           list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
WGROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]
LOGR = 2


def pt_index(t, e, lo):
    return ((t >> lo) << (lo + LOGR)) | (e << lo) | (t & ((1 << lo) - 1))


def lo_of(LOGM, p):
    return max(LOGM - (p + 1) * LOGR, 0)


def cost(pos, groups, ncols):
    tot = 0
    for g in groups:
        cols = {}
        for l in g:
            if l < len(pos):
                cols.setdefault(pos[l] % ncols, set()).add(pos[l])
        if cols:
            tot += max(len(v) for v in cols.values())
    return tot


def schedule(LOGM):
    """window sequence of the shipped schedule (fft_device.h Plan): for 4 points per thread and odd LOGM >= 7 the
    single-stage window sits at 4 (... 7, 5, 4, 2, 0) and its one-bit exchange is done in the wave (permlane swap)"""
    npass = (LOGM + LOGR - 1) // LOGR
    if LOGR == 2 and (LOGM & 1) and LOGM >= 7:
        pa = (LOGM - 5) // 2
        return [LOGM - 2 * (p + 1) for p in range(pa)] + [4, 2, 0]
    return [lo_of(LOGM, p) for p in range(npass)]


def evaluate(LOGM, swz, los=None):
    """-> (read cycles, ideal read, write cycles, ideal write) summed over all LDS exchanges of fwd+inv
    (`los`: window sequence, default the textbook top-down one; one-bit exchanges of `schedule()` are in-wave)"""
    M = 1 << LOGM
    NT = M >> LOGR
    inwave_odd = los is not None
    if los is None:
        los = [lo_of(LOGM, p) for p in range((LOGM + LOGR - 1) // LOGR)]
    npass = len(los)
    rd = wr = ird = iwr = 0
    pairs = [(los[p], los[p + 1]) for p in range(npass - 1)] + [(los[p], los[p - 1]) for p in range(npass - 1, 0, -1)]
    for lo_w, lo_r in pairs:
        if inwave_odd and abs(lo_w - lo_r) == 1:
            continue
        for wave in range(max(NT // 64, 1)):
            lanes = range(wave * 64, min(wave * 64 + 64, NT))
            for e in range(1 << LOGR):
                pw = [swz(pt_index(t, e, lo_w)) for t in lanes]
                pr = [swz(pt_index(t, e, lo_r)) for t in lanes]
                wr += cost(pw, WGROUPS, 8); iwr += (len(pw) + 7) // 8
                rd += cost(pr, RGROUPS, 16); ird += (len(pr) + 15) // 16
    return rd, ird, wr, iwr


def xor_family():
    # p = idx ^ ((idx >> a) & ma) ^ ((idx >> b) & mb)
    for a, b in itertools.product(range(1, 9), repeat=2):
        for ma, mb in itertools.product((0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 15, 10, 9), repeat=2):
            yield (a, ma, b, mb)


if __name__ == "__main__":
    ident = lambda i: i
    pad = lambda i: i + (i >> 4)
    for LOGM in (9, 10):
        print("LOGM", LOGM, "identity", evaluate(LOGM, ident), "pad16", evaluate(LOGM, pad))
    best = []
    for a, ma, b, mb in xor_family():
        f = lambda i, a=a, ma=ma, b=b, mb=mb: i ^ ((i >> a) & ma) ^ ((i >> b) & mb)
        # bijectivity on 0..2047
        if len({f(i) for i in range(2048)}) != 2048 or max(f(i) for i in range(2048)) >= 2048:
            continue
        tot = 0
        for LOGM in (9, 10):
            rd, ird, wr, iwr = evaluate(LOGM, f)
            tot += (rd - ird) + (wr - iwr)
        best.append((tot, a, ma, b, mb))
    best.sort()
    for x in best[:12]:
        print(x)
