"""GPU tests of the round-4 boundary additions, through the C ABI against the CPU oracle, bit for bit:

* mkt_gate_batch_ops  -- a different gate (and optional input NOTs) per ciphertext pair, the shape of the reference's own
  tests (test/KMS.jl:29-34 draws a random gate per step; gate.jl:1-58);
* mkt_gate_batch_gather -- one circuit level per call, operands picked from a ciphertext pool;
* mkt_multi_* -- ONE evaluator over several shards, one caller process: keys uploaded once and replicated,
  contiguous balanced slices, every shard writing its slice of the caller's one output array.  The GPU box has one
  device, so the shards are LOGICAL shards of device 0: forked contexts over one key set (the default), or -- with
  private_keys=True -- one replicated key copy per shard, which runs the device-to-device replication path.
"""
import os
import sys

import numpy as np
import pytest

from helpers import GATE_FUNCS, O, ROOT, encrypt_bits, gpu_scheme, keygen, mk, oracle_scheme

pytestmark = pytest.mark.gpu

SMALL = [
    mk.CGGIparam.scaled(n=20, N=256),
    mk.Blockparam.scaled(n=30, N=256, blk_d=10),
    mk.KMS2party.scaled(n=16, N=256),
    mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8),
    mk.CCS2party.scaled(n=12, N=256),
    mk.KMS4party.scaled(n=8, N=256),
]


def _neg(c):
    return (0 - c.astype(np.int64)).astype(np.uint32)


def oracle_gate_ops(so, ops, x, y):
    """the reference semantics of one coded gate: NOT! (gate.jl:55-58) on the flagged inputs, then the gate (gate.jl:1-53)"""
    out = np.empty_like(x)
    for j in range(len(ops)):
        a = _neg(x[j]) if ops[j] & mk.OP_NOT_X else x[j]
        b = _neg(y[j]) if ops[j] & mk.OP_NOT_Y else y[j]
        out[j] = so.gate(int(ops[j] & 7), a, b)
    return out


def plain_gate_ops(ops, bx, by):
    out = np.empty(len(ops), dtype=bool)
    for j, o in enumerate(ops):
        a = ~bx[j] if o & mk.OP_NOT_X else bx[j]
        b = ~by[j] if o & mk.OP_NOT_Y else by[j]
        out[j] = GATE_FUNCS[int(o & 7)](a, b)
    return out


@pytest.mark.parametrize("p", SMALL, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}")
def test_mixed_gate_batch_matches_the_oracle(require_gpu, p):
    """a random gate per ciphertext pair in ONE call == the oracle gate by gate, host and device memory; the counterpart of
    the reference's test loop (test/KMS.jl:29-34: `rand(1:6)` picks the gate of every step)"""
    import torch
    crs, keys = keygen(p, 41)
    so, sg = oracle_scheme(p, crs, keys), gpu_scheme(p, crs, keys)
    B = 13
    rng = np.random.default_rng(42)
    bits = rng.integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=4200)
    x, y = c[:B], c[B:][::-1].copy()            # cross-party pairs
    bx, by = bits[:B], bits[B:][::-1]
    ops = rng.integers(0, 6, B).astype(np.uint8)
    ops[:6] = np.arange(6)                       # every gate at least once
    ops[1::3] |= mk.OP_NOT_X
    ops[2::4] |= mk.OP_NOT_Y
    want = oracle_gate_ops(so, ops, x, y)
    got_h = sg.gate_ops(ops, x, y)
    assert np.array_equal(got_h, want)
    td = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()   # noqa: E731
    got_d = sg.gate_ops(td(ops), td(x), td(y))
    torch.cuda.synchronize()
    assert np.array_equal(got_d.cpu().numpy().view(np.uint32), want)
    assert np.array_equal(mk.lwe_decrypt(got_h, keys if p.multikey else keys[0], p), plain_gate_ops(ops, bx, by))
    # one op for the whole batch through the coded entry point == mkt_gate_batch
    for op in (0, 3):
        assert np.array_equal(sg.gate_ops(np.full(B, op, np.uint8), x, y), sg.gate(op, x, y))
    with pytest.raises(mk.MktError):
        sg.gate_ops(np.full(B, 6, np.uint8), x, y)          # not a gate
    sg.close()


@pytest.mark.parametrize("p", [mk.CGGIparam.scaled(n=20, N=256), mk.KMS2party.scaled(n=16, N=256)], ids=lambda p: p.name)
def test_gather_level_matches_the_oracle(require_gpu, p):
    """mkt_gate_batch_gather: operands by row index from a pool, output into a later region of the SAME pool (device memory)
    and into a separate array (host memory)"""
    import torch
    crs, keys = keygen(p, 43)
    so, sg = oracle_scheme(p, crs, keys), gpu_scheme(p, crs, keys)
    rng = np.random.default_rng(44)
    P, B = 9, 11
    bits = rng.integers(0, 2, P).astype(bool)
    pool = encrypt_bits(p, keys, bits, seed=4400)
    ix = rng.integers(0, P, B).astype(np.uint32)
    iy = rng.integers(0, P, B).astype(np.uint32)
    ops = (rng.integers(0, 6, B) | (rng.integers(0, 2, B) << 3) | (rng.integers(0, 2, B) << 4)).astype(np.uint8)
    want = oracle_gate_ops(so, ops, pool[ix], pool[iy])
    out_h = np.empty((B, p.lwe_len), dtype=np.uint32)
    sg.gate_gather(ops, pool, ix, iy, out_h)
    assert np.array_equal(out_h, want)
    big = torch.zeros((P + B, p.lwe_len), dtype=torch.int32, device="cuda")
    big[:P] = torch.from_numpy(pool.view(np.int32)).cuda()
    sg.gate_gather(torch.from_numpy(ops).cuda(), big, torch.from_numpy(ix.view(np.int32)).cuda(), torch.from_numpy(iy.view(np.int32)).cuda(), big[P:])
    torch.cuda.synchronize()
    assert np.array_equal(big[P:].cpu().numpy().view(np.uint32), want)
    assert np.array_equal(big[:P].cpu().numpy().view(np.uint32), pool)          # the operand rows are untouched
    with pytest.raises(mk.MktError):
        sg.gate_gather(ops, pool, ix + P, iy, out_h)                              # index outside the pool
    # gates over an EMPTY pool are refused in both memory kinds (device-side index arrays are clamped into the pool: there must be a row to clamp to)
    with pytest.raises(mk.MktError, match="empty pool"):
        sg.gate_gather(ops, pool[:0], ix, iy, out_h)
    with pytest.raises(mk.MktError):                                              # (an empty device view has no address: refused as a null argument)
        sg.gate_gather(torch.from_numpy(ops).cuda(), big[:0], torch.from_numpy(ix.view(np.int32)).cuda(), torch.from_numpy(iy.view(np.int32)).cuda(), big[P:])
    sg.close()


def test_circuit_level_is_one_call_whatever_the_gate_mix(require_gpu):
    """circuit.evaluate_on issues ONE engine call per level (adder: XOR + AND + OR mixed in a level), the ciphertext words equal
    the per-(level, op) evaluation through the oracle backend"""
    from mktfhe_amd import circuit as CI
    p = mk.KMS2party.scaled(n=16, N=256)
    crs, keys = keygen(p, 45)
    so, sg = oracle_scheme(p, crs, keys), gpu_scheme(p, crs, keys)
    circ = CI.ripple_adder(3)
    B = 5
    rng = np.random.default_rng(46)
    bits = rng.integers(0, 2, (6, B)).astype(bool)
    inputs = [np.stack([mk.lwe_ith_encrypt(int(bits[i, j]), (i + j) % 2, keys[(i + j) % 2], p, deterministic_seed=4600 + 10 * i + j) for j in range(B)]) for i in range(6)]
    calls = []
    orig = sg.gate_gather
    sg.gate_gather = lambda *a: (calls.append(len(a[0])), orig(*a))[1]
    outs = CI.evaluate_on(circ, inputs, sg)
    depth, sched = circ.levels()
    assert len(calls) == len(sched) == max(depth)                # one call per level ...
    assert sum(calls) == B * sum(len(v) for lv in sched.values() for v in lv.values())   # ... covering every gate
    ref = CI.evaluate(circ, inputs, lambda op, x, y: so.gate_batch(op, x, y, threads=8), _neg)
    for o, r in zip(outs, ref):
        assert np.array_equal(o, r)
    a = sum(bits[i].astype(int) << i for i in range(3)); b = sum(bits[3 + i].astype(int) << i for i in range(3))
    assert np.array_equal(sum(mk.lwe_decrypt(o, keys, p).astype(int) << i for i, o in enumerate(outs)), a + b)
    # a circuit whose OUTPUT is a NOT and whose gates read NOTs of NOTs
    c2 = CI.Circuit(); u, v = c2.input(), c2.input()
    c2.output(c2.NOT(c2.XOR(c2.NOT(c2.NOT(u)), c2.NOT(v))))
    (o2,) = CI.evaluate_on(c2, inputs[:2], sg)
    (r2,) = CI.evaluate(c2, inputs[:2], lambda op, x, y: so.gate_batch(op, x, y, threads=8), _neg)
    assert np.array_equal(o2, r2)
    sg.close()


def oracle_mux(so, p, s, a, b):
    """the native MUX restated WITHOUT composite gates, on the oracle's operators: two blindrotate! (bootstrapping.jl:8-24 on the
    AND-linear parts, gate.jl:10-17 / NOT! :55-58), the accumulators added polynomial by polynomial, + 1/8 at X^0 of b, one keyswitch!"""
    accs = []
    for x, y in ((s, a), (_neg(s), b)):
        lin = O.gate_linear(1, x, y)
        at, bt = so.modswitch(lin)
        accs.append(so.blindrotate(at, so.testvector(bt)).astype(np.uint64))
    mask = np.uint64((1 << p.W) - 1)
    acc = ((accs[0] + accs[1]) & mask).reshape(-1, p.N)              # [(b, a_0 ..)][N]
    acc[0, 0] = np.uint64((int(acc[0, 0]) + (1 << (p.W - 3))) & int(mask))
    return so.keyswitch(acc)


MUX_SETS = [mk.CGGIparam.scaled(n=20, N=256), mk.Blockparam.scaled(n=30, N=256, blk_d=10), mk.KMS2party.scaled(n=16, N=256),
            mk.KMS2partyblock.scaled(n=24, N=256, blk_d=8), mk.CCS2party.scaled(n=12, N=256), mk.KMS4party.scaled(n=8, N=256),
            mk.CGGIparam.scaled(n=16, N=256, k=2)]


@pytest.mark.parametrize("p", MUX_SETS, ids=lambda p: f"{p.name}-n{p.n}-N{p.N}-k{p.k}")
def test_native_mux_matches_its_oracle_restatement(require_gpu, p):
    """mkt_mux_batch (two rotations + one key switch) == the same construction on the oracle's operators, word for word, for all
    eight (s, a, b) combinations with the three operands under different parties; it decrypts to s ? a : b; the composite of the
    reference's gates decrypts to the same bits; host and device memory; ragged batch"""
    import torch
    crs, keys = keygen(p, 61)
    so, sg = oracle_scheme(p, crs, keys), gpu_scheme(p, crs, keys)
    B = 11
    combos = np.array([[(i >> 2) & 1, (i >> 1) & 1, i & 1] for i in range(B)], dtype=bool)      # every combination, then repeats
    k = p.nparty
    enc = lambda bit, party, seed: mk.lwe_ith_encrypt(int(bit), party % k, keys[party % k], p, deterministic_seed=seed)   # noqa: E731
    s = np.stack([enc(combos[j, 0], j, 6100 + j) for j in range(B)])
    a = np.stack([enc(combos[j, 1], j + 1, 6200 + j) for j in range(B)])
    b = np.stack([enc(combos[j, 2], j + 2, 6300 + j) for j in range(B)])
    want = np.stack([oracle_mux(so, p, s[j], a[j], b[j]) for j in range(B)])
    got = mk.MUX(s, a, b, sg)
    assert np.array_equal(got, want)
    td = lambda v: torch.from_numpy(v.view(np.int32)).cuda()   # noqa: E731
    got_d = mk.MUX(td(s), td(a), td(b), sg)
    torch.cuda.synchronize()
    assert np.array_equal(got_d.cpu().numpy().view(np.uint32), want)
    kk = keys if p.multikey else keys[0]
    plain = np.where(combos[:, 0], combos[:, 1], combos[:, 2])
    assert np.array_equal(mk.lwe_decrypt(got, kk, p), plain)
    assert np.array_equal(mk.lwe_decrypt(mk.MUX_composite(s, a, b, sg), kk, p), plain)
    # a second level on the outputs (the output noise of the native MUX must leave room for another gate)
    lvl2 = mk.NAND(got[:5], got[5:10], sg)
    assert np.array_equal(mk.lwe_decrypt(lvl2, kk, p), ~(plain[:5] & plain[5:10]))
    sg.close()


def test_native_mux_full_size_exact_and_sharded(require_gpu):
    """the headline shape: native MUX == its oracle restatement on a sample; the EXACT arithmetic and a three-shard evaluator
    decrypt to s ? a : b as well (EXACT words differ from the Float64 ones by construction)"""
    p = mk.KMS2party_N1024_l2
    crs, keys = keygen(p, 62)
    so = oracle_scheme(p, crs, keys)
    B = 8
    combos = np.array([[(i >> 2) & 1, (i >> 1) & 1, i & 1] for i in range(B)], dtype=bool)
    s = np.stack([mk.lwe_ith_encrypt(int(combos[j, 0]), j % 2, keys[j % 2], p, deterministic_seed=6400 + j) for j in range(B)])
    a = np.stack([mk.lwe_ith_encrypt(int(combos[j, 1]), (j + 1) % 2, keys[(j + 1) % 2], p, deterministic_seed=6500 + j) for j in range(B)])
    b = np.stack([mk.lwe_ith_encrypt(int(combos[j, 2]), j % 2, keys[j % 2], p, deterministic_seed=6600 + j) for j in range(B)])
    plain = np.where(combos[:, 0], combos[:, 1], combos[:, 2])
    sg = gpu_scheme(p, crs, keys)
    got = mk.MUX(s, a, b, sg)
    assert np.array_equal(got[:2], np.stack([oracle_mux(so, p, s[j], a[j], b[j]) for j in range(2)]))
    assert np.array_equal(mk.lwe_decrypt(got, keys, p), plain)
    multi = multi_scheme(p, crs, keys, [0, 0, 0], mk.ARITH_F64REF)
    assert np.array_equal(multi.mux(s, a, b), got)
    multi.close(); sg.close()
    sx = gpu_scheme(p, crs, keys, arith=mk.ARITH_EXACT)
    assert np.array_equal(mk.lwe_decrypt(mk.MUX(s, a, b, sx), keys, p), plain)
    sx.close()


@pytest.mark.parametrize("p", [mk.KMS2party.scaled(n=16, N=256), mk.CGGIparam.scaled(n=20, N=256), mk.CCS2party.scaled(n=12, N=256)], ids=lambda p: p.name)
def test_circuit_with_native_mux_nodes(require_gpu, p):
    """Circuit.MUXN: a level of native MUX gates is ONE mkt_mux_batch_gather (operands from the pool, free NOTs as flags / swapped
    operands), mixed with two-input gates of the same level; the ciphertext words equal the evaluation through the oracle backend, whose MUX
    is the composite-free restatement"""
    import torch
    from mktfhe_amd import circuit as CI
    crs, keys = keygen(p, 65)
    so, sg = oracle_scheme(p, crs, keys), gpu_scheme(p, crs, keys)
    c = CI.Circuit(); s_, a_, b_, d_ = (c.input() for _ in range(4))
    m = c.MUXN(c.NOT(s_), a_, c.NOT(b_)); x = c.XOR(a_, d_)
    c.output(c.MUXN(m, x, d_)); c.output(m); c.output(c.NOT(c.MUXN(s_, c.NOT(a_), b_)))
    B = 6
    rng = np.random.default_rng(66)
    bits = rng.integers(0, 2, (4, B)).astype(bool)
    k = p.nparty
    inputs = [np.stack([mk.lwe_ith_encrypt(int(bits[i, j]), (i + j) % k, keys[(i + j) % k], p, deterministic_seed=6600 + 10 * i + j) for j in range(B)]) for i in range(4)]
    mux_o = lambda S, A, Bv: np.stack([oracle_mux(so, p, S[j], A[j], Bv[j]) for j in range(len(S))])   # noqa: E731
    ref = CI.evaluate(c, inputs, lambda op, xx, yy: so.gate_batch(op, xx, yy, threads=8), _neg, mux_fn=mux_o)
    outs = CI.evaluate_on(c, inputs, sg)
    kk = keys if p.multikey else keys[0]
    for o, r, w in zip(outs, ref, c.plain(bits)):
        assert np.array_equal(o, r)
        assert np.array_equal(mk.lwe_decrypt(o, kk, p), w)
    outs_d = CI.evaluate_on(c, [torch.from_numpy(v.view(np.int32)).cuda() for v in inputs], sg)
    torch.cuda.synchronize()
    for o, r in zip(outs_d, ref):
        assert np.array_equal(o.cpu().numpy().view(np.uint32), r)
    with pytest.raises(mk.MktError):
        sg.mux_gather(inputs[0], np.array([0], np.uint32), np.array([9], np.uint32), np.array([0], np.uint32), np.empty((1, p.lwe_len), np.uint32))
    sg.close()


# ---------------------------------------------------------------- the multi-shard evaluator
MULTI_SETS = [
    (mk.KMS2party.scaled(n=16, N=256), mk.ARITH_F64REF),
    (mk.CGGIparam.scaled(n=20, N=256), mk.ARITH_F64REF),
    (mk.CCS2party.scaled(n=12, N=256), mk.ARITH_F64REF),
    (mk.Blockparam.scaled(n=30, N=256, blk_d=10), mk.ARITH_F64REF),
    (mk.KMS2party.scaled(n=8, N=256), mk.ARITH_EXACT),
    (mk.CGGIparam.scaled(n=12, N=256), mk.ARITH_EXACT),
]


def multi_scheme(p, crs, keys, devices, arith, private_keys=False):
    # replicated keys are tested together with the staged data path (device arrays through the shards' staging buffers)
    return mk.setup_multi(p, devices, keys=keys if p.multikey else keys[0], a=crs, arith=arith, private_keys=private_keys, stage_always=private_keys)


@pytest.mark.parametrize("private", [False, True], ids=["shared-keys", "replicated-keys-staged-io"])
@pytest.mark.parametrize("nshards", [2, 3])
@pytest.mark.parametrize("p,arith", MULTI_SETS, ids=lambda v: v.name if hasattr(v, "name") else ("exact" if v else "f64ref"))
def test_multi_shard_evaluator_equals_the_single_context(require_gpu, p, arith, nshards, private):
    """N logical shards of device 0 behind ONE mkt_multi handle: every batch entry point gives, word for word, what one
    context gives on the whole batch (and, in F64REF, what the oracle gives), for ragged splits (7 = 3+2+2, 2 over 3 shards
    leaves one shard empty, 1), in host and in device memory, with the keys shared per device or replicated per shard."""
    import torch
    crs, keys = keygen(p, 51)
    single = gpu_scheme(p, crs, keys, arith=arith)
    multi = multi_scheme(p, crs, keys, [0] * nshards, arith, private)
    so = oracle_scheme(p, crs, keys) if arith == mk.ARITH_F64REF else None
    assert multi.nshards == nshards and multi.shard_range(7, 0) == (0, 7 // nshards + 1)
    rng = np.random.default_rng(52)
    td = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()   # noqa: E731
    for B in (7, 2, 1):
        bits = rng.integers(0, 2, 2 * B).astype(bool)
        c = encrypt_bits(p, keys, bits, seed=5200 + B)
        x, y = c[:B], c[B:]
        for op in (0, 3):
            want = single.gate(op, x, y)
            assert np.array_equal(multi.gate(op, x, y), want), (B, op, "host")
            got = multi.gate(op, td(x), td(y))
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want), (B, op, "device")
            if so is not None:
                assert np.array_equal(want, np.stack([so.gate(op, x[j], y[j]) for j in range(B)]))
            assert np.array_equal(mk.lwe_decrypt(want, keys if p.multikey else keys[0], p), GATE_FUNCS[op](bits[:B], bits[B:]))
        ops = (rng.integers(0, 6, B) | (rng.integers(0, 2, B) << 3)).astype(np.uint8)
        assert np.array_equal(multi.gate_ops(ops, x, y), single.gate_ops(ops, x, y))
        assert np.array_equal(multi.gate_ops(td(ops), td(x), td(y)).cpu().numpy().view(np.uint32), single.gate_ops(ops, x, y))
        # bootstrapping! in place, NOT!, blindrotate! / keyswitch! (the operators the north star names)
        lin = np.stack([O.gate_linear(0, x[j], y[j]) for j in range(B)])
        assert np.array_equal(multi.bootstrapping_(lin.copy()), single.bootstrapping_(lin.copy()))
        assert np.array_equal(multi.not_(x.copy()), single.not_(x.copy()))
        at, bt = single.modswitch(lin)
        W = p.W
        acc0 = np.zeros((B, 1 + p.k, p.N), dtype=p.ring_dtype)
        for j in range(B):      # bootstrapping.jl:11-23
            e = 1 << (W - 3)
            tb = int(bt[j])
            lo, hi = (e, (1 << W) - e) if tb <= p.N else ((1 << W) - e, e)
            tb = tb if tb <= p.N else tb - p.N
            acc0[j, 0, :tb] = lo; acc0[j, 0, tb:] = hi
        acc_s = single.blindrotate_(at, acc0.copy())
        acc_m = multi.blindrotate_(at, acc0.copy())
        assert np.array_equal(acc_m, acc_s)
        assert np.array_equal(multi.keyswitch(acc_m), single.keyswitch(acc_s))
    multi.close(); single.close()


def test_circuit_instances_sharded_over_a_multi_scheme(require_gpu):
    """circuit.evaluate_sharded: every shard of a MultiScheme evaluates the whole circuit on its slice of the instances (7 instances over
    3 shards: 3 + 2 + 2), host arrays and GPU tensors: the same words as one context on all instances"""
    import torch
    from mktfhe_amd import circuit as CI
    p = mk.KMS2party.scaled(n=16, N=256)
    crs, keys = keygen(p, 57)
    single = gpu_scheme(p, crs, keys)
    multi = multi_scheme(p, crs, keys, [0, 0, 0], mk.ARITH_F64REF)
    circ = CI.ripple_adder(2)
    cm = CI.Circuit(); s_, a_, b_ = cm.input(), cm.input(), cm.input(); cm.output(cm.MUXN(s_, a_, cm.NOT(b_))); cm.output(cm.NAND(a_, b_))
    B = 7
    rng = np.random.default_rng(58)
    for cc in (circ, cm):
        bits = rng.integers(0, 2, (cc.n_inputs, B)).astype(bool)
        inputs = [np.stack([mk.lwe_ith_encrypt(int(bits[i, j]), (i + j) % 2, keys[(i + j) % 2], p, deterministic_seed=5800 + 10 * i + j) for j in range(B)]) for i in range(cc.n_inputs)]
        want = CI.evaluate_on(cc, inputs, single)
        got = CI.evaluate_sharded(cc, inputs, multi)
        got_d = CI.evaluate_sharded(cc, [torch.from_numpy(v.view(np.int32)).cuda() for v in inputs], multi)
        torch.cuda.synchronize()
        for w, g, gd, pl in zip(want, got, got_d, cc.plain(bits)):
            assert np.array_equal(g, w) and np.array_equal(gd.cpu().numpy().view(np.uint32), w)
            assert np.array_equal(mk.lwe_decrypt(g, keys, p), pl)
    assert np.array_equal(multi.gate(0, inputs[0], inputs[1]), single.gate(0, inputs[0], inputs[1]))      # the handle's own calls still work after the shards were borrowed
    multi.close(); single.close()


def test_multi_shard_evaluator_full_size_and_errors(require_gpu):
    """the headline shape (BASELINE configs[1]) on three logical shards == one context == the oracle sample; keys are
    immutable once replicated; calls before mkt_multi_replicate are refused"""
    p = mk.KMS2party_N1024_l2
    crs, keys = keygen(p, 53)
    single = gpu_scheme(p, crs, keys)
    multi = mk.MultiScheme(p, [0, 0, 0])
    B = 10
    bits = np.random.default_rng(54).integers(0, 2, 2 * B).astype(bool)
    c = encrypt_bits(p, keys, bits, seed=5400)
    with pytest.raises(mk.MktError, match="mkt_multi_replicate"):
        multi.gate(0, c[:B], c[B:])
    multi.load_crs(crs)
    for i, kk in enumerate(keys):
        multi.load_party(i, kk)
    multi.replicate()
    with pytest.raises(mk.MktError, match="immutable"):
        multi.load_crs(crs)
    want = single.gate(0, c[:B], c[B:])
    assert np.array_equal(multi.gate(0, c[:B], c[B:]), want)
    so = oracle_scheme(p, crs, keys)
    assert np.array_equal(want[:3], so.gate_batch(0, c[:3], c[B:B + 3], threads=3))
    # per-shard timing and kernel names through the borrowed contexts
    sh = multi.shard(1)
    sh.enable_timing(True)
    multi.gate(0, c[:B], c[B:])
    ms, n = sh.kernel_ms(1)
    assert n == 1 and ms > 0 and sh.last_kernel_name().startswith("blindrotate_")
    with pytest.raises(mk.MktError):
        mk.MultiScheme(p, [0, 99])                               # no such device
    multi.close(); single.close()


def test_last_kernel_name_follows_the_dispatch(require_gpu):
    """mkt_last_kernel_name: bench.py's roofline names the kernel that actually ran"""
    p = mk.Blockparam.scaled(n=24, N=1024, blk_d=8)
    crs, keys = keygen(p, 55)
    sg = gpu_scheme(p, crs, keys)
    c = encrypt_bits(p, keys, np.zeros(10, dtype=bool), seed=5500)
    assert sg.last_kernel_name() == ""
    sg.set_option("rot_blkg", 1); sg.set_option("rot_wide", 1)
    sg.gate(0, c[:5], c[5:])
    assert sg.last_kernel_name() == "blindrotate_k1_kernel"
    sg.set_option("rot_blkg", 4)
    sg.gate(0, c[:5], c[5:])
    assert sg.last_kernel_name() == "blindrotate_blk_kernel"
    with pytest.raises(mk.MktError, match="unknown option"):
        sg.set_option("no_such_switch", 1)
    sg.close()


@pytest.mark.parametrize("p,B", [(mk.KMS4party, 65536), (mk.CCS8party_N2048, 8192)], ids=lambda v: getattr(v, "name", str(v)))
def test_baseline_multi_gpu_configs_at_full_batch_on_eight_logical_shards(require_gpu, p, B):
    """BASELINE.json configs[2] and [3] are FIXED batches over the 8 GPUs of a node (65 536 KMS k = 4 gates, 8 192 CCS k = 8 N = 2048
    gates).  Here the WHOLE batch runs through one mkt_multi handle with eight shards (logical shards of the one GPU: the same code
    path, keys shared): every output decrypts, gates sampled at the shard boundaries and inside equal the oracle bit for bit, a gate's
    words do not depend on which shard evaluated it, and a second evaluation reproduces the first word for word."""
    crs, keys = keygen(p, 21)
    multi = multi_scheme(p, crs, keys, [0] * 8, mk.ARITH_F64REF)
    rng = np.random.default_rng(81)
    nu = 64 * p.nparty
    ubits = rng.integers(0, 2, nu).astype(bool)
    uniq = encrypt_bits(p, keys, ubits, seed=9100)
    ix, iy = rng.integers(0, nu, B), rng.integers(0, nu, B)
    x, y = uniq[ix], uniq[iy]
    out = multi.gate(0, x, y)
    got = mk.lwe_decrypt(out, keys, p)
    assert np.array_equal(got, ~(ubits[ix] & ubits[iy]))
    bounds = [multi.shard_range(B, s) for s in range(8)]
    assert bounds[0][0] == 0 and bounds[-1][1] == B and all(bounds[s][1] == bounds[s + 1][0] for s in range(7))
    so = oracle_scheme(p, crs, keys)
    pick = np.array(sorted({0, B - 1, bounds[3][0] - 1, bounds[3][0], bounds[6][1] - 1, int(rng.integers(0, B))}))
    assert np.array_equal(out[pick], so.gate_batch(0, x[pick], y[pick], threads=8))
    perm = rng.permutation(B)                                            # the same gates, dealt to other shards
    sub = perm[:2048]
    assert np.array_equal(multi.gate(0, x[sub], y[sub]), out[sub])
    assert np.array_equal(multi.gate(0, x, y), out)
    multi.close()


# ---- first contact with a multi-GPU node, rehearsed on one GPU ----
def _bench_line(args, env_extra, timeout):
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-cpu-baseline", "--no-roofline", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_eight_ranks_rendezvous_on_a_shared_gpu(require_gpu):
    """the driver's 8-GPU command shape -- `bench.py --gpus 8` -- with all eight ranks on this box's one GPU: eight processes, eight ports'
    worth of rendezvous, the store time-outs, the census all-reduce (ranks_seen == 8) and the per-rank step times"""
    j = _bench_line(["--gpus", "8", "--steps", "1", "--warmup", "1", "--batch", "16", "--workload", "cggi"], dict(MKT_BENCH_SHARE_GPU="1", MKT_BENCH_BACKEND="gloo"), 1500)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and len(j["per_rank_ms_per_step"]) == 8 and all(v > 0 for v in j["per_rank_ms_per_step"])
    assert j["config"]["batch_total"] == 8 * 16 and j["decrypt_checked"] == 8 * 16 and j["decrypt_errors"] == 0
    assert abs(j["value"] - 8 * 16 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


@pytest.mark.gpu
def test_bench_inproc_launcher_eight_shards_strong_scaling(require_gpu):
    """BASELINE configs[2]'s shape through the ONE-process launcher: `--launcher inproc --gpus 8 --scaling strong --workload kms4party --batch 64`
    (eight logical shards of this box's one GPU, eight gates each)"""
    j = _bench_line(["--gpus", "8", "--launcher", "inproc", "--scaling", "strong", "--workload", "kms4party", "--batch", "64", "--steps", "1", "--warmup", "0"],
                    dict(MKT_BENCH_SHARE_GPU="1"), 1500)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["scaling"] == "strong" and len(j["per_rank_ms_per_step"]) == 8
    assert j["config"]["batch_total"] == 64 and j["config"]["batch_per_gpu"] == 8 and j["decrypt_checked"] == 64 and j["decrypt_ok"]


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [mk.ARITH_F64REF, mk.ARITH_EXACT], ids=["f64ref", "exact"])
def test_eight_shards_with_keys_replicated_through_the_host(require_gpu, arith):
    """mkt_multi_create over eight logical shards with every device-to-device copy forced through a host buffer (MKT_MULTI_NO_PEER: the path
    the engine takes by itself where hipMemcpyPeer is refused), each shard holding its OWN replica of the key set and device arrays
    travelling through the shards' staging buffers: word-identical to one context, in host and in device memory, ragged split included"""
    import torch
    p = mk.KMS2party.scaled(n=12, N=512)
    crs, keys = keygen(p, 61)
    single = gpu_scheme(p, crs, keys, arith=arith)
    multi = mk.setup_multi(p, [0] * 8, keys=keys, a=crs, arith=arith, private_keys=True, stage_always=True, no_peer=True)
    assert multi.nshards == 8
    rng = np.random.default_rng(62)
    for B in (19, 8, 3):
        bits = rng.integers(0, 2, 2 * B).astype(bool)
        c = encrypt_bits(p, keys, bits, seed=6200 + B)
        x, y = c[:B], c[B:]
        want = single.gate(0, x, y)
        assert np.array_equal(multi.gate(0, x, y), want), (B, "host")
        xd, yd = torch.from_numpy(x.view(np.int32)).cuda(), torch.from_numpy(y.view(np.int32)).cuda()
        assert np.array_equal(multi.gate(0, xd, yd).cpu().numpy().view(np.uint32), want), (B, "device")
        assert np.array_equal(mk.lwe_decrypt(want, keys, p), ~(bits[:B] & bits[B:]))
    multi.close(); single.close()
