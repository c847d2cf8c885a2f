"""Client-side randomness (mktfhe_amd/csrc/rng_chacha.h, client.cpp): ChaCha20 against the RFC 8439 block-function
vector, the deterministic Box-Muller against libm, and the API contract -- fresh OS entropy by default on every key
generation and every encryption (reference: ChaCha20Stream per call, sampler.jl:1-34), reproducible only when a
deterministic_seed is passed explicitly."""
import os
import subprocess

import numpy as np

from helpers import ROOT, mk

RFC8439_BLOCK = ("e4e7f110 15593bd1 1fdd0f50 c47120a3 c7f4d1c7 0368c033 9aaa2204 4e6cd4c3 "
                 "466482d2 09aa9f07 05d7c214 a2028bd9 d19c12b5 b94e16de e883d0cb 4e3c50a2")


def test_chacha20_rfc8439_vector_and_gaussian(tmp_path):
    exe = str(tmp_path / "rng_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "mktfhe_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "rng_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].strip() == RFC8439_BLOCK                      # RFC 8439 section 2.3.2 test vector
    stats = dict(zip(out[1].split()[0::2], out[1].split()[1::2]))
    assert abs(float(stats["mean"])) < 5e-3 and abs(float(stats["var"]) - 1) < 5e-3 and abs(float(stats["kurt"]) - 3) < 0.02
    assert float(stats["max"]) > 4.8                            # real Gaussian tails (4e6 draws reach ~5.2 sigma)
    assert float(out[2].split()[-1]) < 1e-12                    # agrees with libm's sqrt(-2 ln u1) cos(2 pi u2)


def test_default_randomness_is_fresh():
    p = mk.CGGIparam.scaled(n=16, N=64)
    k1, k2 = mk.PartyKeys(p), mk.PartyKeys(p)
    assert not np.array_equal(k1.brk, k2.brk) and not np.array_equal(k1.ksk, k2.ksk)
    keys = [mk.PartyKeys(p) for _ in range(6)]
    assert len({bytes(k.lwekey) for k in keys}) == 6            # 16-bit keys: a collision among 6 is < 0.03 %
    c0, c1, c0b = mk.lwe_encrypt(0, k1, p), mk.lwe_encrypt(1, k1, p), mk.lwe_encrypt(0, k1, p)
    assert not np.array_equal(c0[:-1], c1[:-1]) and not np.array_equal(c0, c0b)     # fresh mask and noise per call
    assert mk.lwe_decrypt(c0, k1, p) is False and mk.lwe_decrypt(c1, k1, p) is True
    q = mk.KMS2party.scaled(n=8, N=64)
    assert not np.array_equal(mk.CRS(q), mk.CRS(q))


def test_deterministic_seed_is_explicit_and_reproducible():
    p = mk.KMS2party.scaled(n=8, N=64)
    a = mk.CRS(p, deterministic_seed=3)
    assert np.array_equal(a, mk.CRS(p, deterministic_seed=3)) and not np.array_equal(a, mk.CRS(p, deterministic_seed=4))
    k0 = mk.party_keygen(a, p, party=0, deterministic_seed=3)
    k0b = mk.party_keygen(a, p, party=0, deterministic_seed=3)
    k1 = mk.party_keygen(a, p, party=1, deterministic_seed=3)
    assert np.array_equal(k0.brk, k0b.brk) and np.array_equal(k0.lwekey, k0b.lwekey)
    assert not np.array_equal(k0.lwekey, k1.lwekey)            # parties draw from different streams of one seed
    c = mk.lwe_ith_encrypt(1, 1, k1, p, deterministic_seed=9)
    assert np.array_equal(c, mk.lwe_ith_encrypt(1, 1, k1, p, deterministic_seed=9))
    raw = bytes(range(32))
    assert np.array_equal(mk.CRS(p, deterministic_seed=raw), mk.CRS(p, deterministic_seed=raw))


def test_native_conversion_without_compare_matches_the_reference_form(tmp_path):
    """fft_device.h native(): the select-free conversion (low dword of v + 2^52) equals arithmetic.jl:1-9's
    `x == 2^W ? 0 : trunc(x)` on every input class: both signs, every exponent from denormals to 2^119, exact multiples
    of 2^W minus tiny offsets (the rounded-up-to-2^W case), fractions"""
    exe = str(tmp_path / "native_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "csrc", "native_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith(" 0 mismatches"), out.stdout[-500:]
