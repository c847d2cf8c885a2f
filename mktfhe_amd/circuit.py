"""Levelised gate-circuit evaluation on top of the batched gate primitive.

The reference's tests fold gates one at a time, a random gate per step (test/KMS.jl:29-34).  Here a circuit is a DAG of
two-input bootstrapped gates (gate.jl:1-53) and free NOTs (gate.jl:55-58); ALL gates of equal depth -- whatever their
type -- are evaluated in ONE engine call over all `B` independent input sets at once (mkt_gate_batch_gather: a per-gate op
code, operands picked by row index from a ciphertext pool that stays in HBM, NOTs folded into the gate's linear part), so a
level costs one round of launches however many gate kinds it mixes.
"""
from collections import defaultdict

import numpy as np

from .params import NAND_OP, AND_OP, OR_OP, XOR_OP, XNOR_OP, NOR_OP

_NOT = -1
_MUX = 6        # native three-input node (mkt_mux_batch_gather): nodes entry (_MUX, s, (a, b))


class Circuit:
    def __init__(self):
        self.nodes = []      # (op, a, b) ; inputs: ("in", index, None)
        self.n_inputs = 0
        self.outputs = []

    def input(self):
        self.nodes.append(("in", self.n_inputs, None))
        self.n_inputs += 1
        return len(self.nodes) - 1

    def gate(self, op, a, b):
        assert 0 <= op <= 5 and 0 <= a < len(self.nodes) and 0 <= b < len(self.nodes)
        self.nodes.append((op, a, b))
        return len(self.nodes) - 1

    def NAND(self, a, b): return self.gate(NAND_OP, a, b)
    def AND(self, a, b): return self.gate(AND_OP, a, b)
    def OR(self, a, b): return self.gate(OR_OP, a, b)
    def XOR(self, a, b): return self.gate(XOR_OP, a, b)
    def XNOR(self, a, b): return self.gate(XNOR_OP, a, b)
    def NOR(self, a, b): return self.gate(NOR_OP, a, b)

    def NOT(self, a):
        self.nodes.append((_NOT, a, None))
        return len(self.nodes) - 1

    def MUX(self, s, a, b):
        """s ? a : b.  The reference has no MUX gate (gate.jl exports NAND..NOR, NOT!); this is the composite
        OR(AND(s, a), AND(NOT s, b)) of its gates: two levels, three bootstraps, the two ANDs in one batch."""
        return self.OR(self.AND(s, a), self.AND(self.NOT(s), b))

    def MUXN(self, s, a, b):
        """s ? a : b as ONE node: the engine's native MUX (two blind rotations + one key switch; mkt_mux_batch) instead of the three
        bootstraps of the composite.  One level deep."""
        assert all(0 <= v < len(self.nodes) for v in (s, a, b))
        self.nodes.append((_MUX, s, (a, b)))
        return len(self.nodes) - 1

    def output(self, w):
        self.outputs.append(w)
        return w

    def levels(self):
        """-> (depth per node, {level: {op: [node ids]}}) ; NOT and inputs cost no level"""
        depth = []
        for op, a, b in self.nodes:
            if op == "in":
                depth.append(0)
            elif op == _NOT:
                depth.append(depth[a])
            elif op == _MUX:
                depth.append(1 + max(depth[a], depth[b[0]], depth[b[1]]))
            else:
                depth.append(1 + max(depth[a], depth[b]))
        sched = defaultdict(lambda: defaultdict(list))
        for i, (op, a, b) in enumerate(self.nodes):
            if op not in ("in", _NOT):
                sched[depth[i]][op].append(i)
        return depth, sched

    def plain(self, bits):
        """reference evaluation on plaintext bits: bits (n_inputs, ...) bool"""
        f = {0: lambda x, y: ~(x & y), 1: lambda x, y: x & y, 2: lambda x, y: x | y, 3: lambda x, y: x ^ y,
             4: lambda x, y: ~(x ^ y), 5: lambda x, y: ~(x | y)}
        v = []
        for op, a, b in self.nodes:
            if op == _MUX:
                v.append(np.where(v[a], v[b[0]], v[b[1]]))
            else:
                v.append(bits[a] if op == "in" else (~v[a] if op == _NOT else f[op](v[a], v[b])))
        return [v[w] for w in self.outputs]


def _cat(xs):
    if type(xs[0]).__module__.startswith("torch"):
        import torch
        return torch.cat(xs, 0)
    return np.concatenate(xs, 0)


def evaluate(circ: Circuit, inputs, gate_fn, not_fn, mux_fn=None):
    """inputs: list of n_inputs arrays [B, lwe_len] (numpy or GPU tensors).  gate_fn(op, x, y) -> out is the
    batched gate (Scheme.gate), not_fn(x) -> negated COPY, mux_fn(s, a, b) -> out the native MUX (needed only for MUXN nodes).
    Returns the output ciphertext arrays [B, lwe_len].
    Number of gate_fn calls = number of distinct (level, op) pairs, independent of the circuit width."""
    assert len(inputs) == circ.n_inputs
    depth, sched = circ.levels()
    val = [None] * len(circ.nodes)

    def resolve(i):
        if val[i] is None:
            op, a, _ = circ.nodes[i]
            val[i] = inputs[a] if op == "in" else not_fn(resolve(a))
        return val[i]

    B = inputs[0].shape[0] if circ.n_inputs else 0
    for lvl in sorted(sched):
        for op, ids in sched[lvl].items():
            x = _cat([resolve(circ.nodes[i][1]) for i in ids])
            if op == _MUX:
                out = mux_fn(x, _cat([resolve(circ.nodes[i][2][0]) for i in ids]), _cat([resolve(circ.nodes[i][2][1]) for i in ids]))
            else:
                y = _cat([resolve(circ.nodes[i][2]) for i in ids])
                out = gate_fn(op, x, y)
            for j, i in enumerate(ids):
                val[i] = out[j * B:(j + 1) * B]
    return [resolve(w) for w in circ.outputs]


class Plan:
    """The launch schedule of a circuit over B instances: one mkt_gate_batch_gather per level.  Pool rows: node slot s,
    instance b -> row s * B + b; slots 0 .. n_inputs-1 are the inputs, then the gates level by level (so a level's outputs
    are one contiguous region of the pool); NOT nodes own no slot -- they resolve to (slot of their source, negated)."""

    def __init__(self, circ: Circuit, B):
        depth, sched = circ.levels()
        self.B, self.n_inputs = B, circ.n_inputs
        slot, neg = [None] * len(circ.nodes), [False] * len(circ.nodes)
        for i, (op, a, _) in enumerate(circ.nodes):
            if op == "in":
                slot[i] = a
        nslots = circ.n_inputs
        order = []
        for lvl in sorted(sched):
            ids = sorted((i for lst in sched[lvl].values() for i in lst), key=lambda i: (circ.nodes[i][0] == _MUX, i))   # two-input gates first, native MUX nodes after: each kind one contiguous region of the pool
            for i in ids:
                slot[i] = nslots
                nslots += 1
            order.append(ids)

        def src(i):                      # -> (slot, negated) through chains of NOTs
            n = False
            while circ.nodes[i][0] == _NOT:
                n = not n
                i = circ.nodes[i][1]
            return slot[i], n

        inst = np.arange(B, dtype=np.uint32)
        self.levels = []        # two-input gates of a level: (first slot, count, ops, ix, iy)
        self.mux_levels = []    # its native MUX nodes, if any: (first slot, count, is, ia, ib, not_ab) or None
        for ids in order:
            g2 = [i for i in ids if circ.nodes[i][0] != _MUX]
            g3 = [i for i in ids if circ.nodes[i][0] == _MUX]
            ops = np.empty((len(g2), B), dtype=np.uint8)
            ix = np.empty((len(g2), B), dtype=np.uint32)
            iy = np.empty((len(g2), B), dtype=np.uint32)
            for j, i in enumerate(g2):
                op, a, b = circ.nodes[i]
                (sa, na), (sb, nb) = src(a), src(b)
                ops[j] = op | (8 if na else 0) | (16 if nb else 0)
                ix[j] = sa * B + inst
                iy[j] = sb * B + inst
            self.levels.append((slot[g2[0]] if g2 else 0, len(g2), ops.ravel(), ix.ravel(), iy.ravel()))
            if g3:
                js, ja, jb = (np.empty((len(g3), B), dtype=np.uint32) for _ in range(3))
                fl = np.empty((len(g3), B), dtype=np.uint8)
                for j, i in enumerate(g3):
                    _, s_, (a, b) = circ.nodes[i]
                    (ss, ns), (sa, na), (sb, nb) = src(s_), src(a), src(b)
                    if ns:                                   # MUX(NOT s, a, b) = MUX(s, b, a)
                        (sa, na), (sb, nb) = (sb, nb), (sa, na)
                    js[j], ja[j], jb[j] = ss * B + inst, sa * B + inst, sb * B + inst
                    fl[j] = (1 if na else 0) | (2 if nb else 0)
                self.mux_levels.append((slot[g3[0]], len(g3), js.ravel(), ja.ravel(), jb.ravel(), fl.ravel()))
            else:
                self.mux_levels.append(None)
        self.rows = nslots * B
        self.gates = (sum(n for _, n, _, _, _ in self.levels) + sum(m[1] for m in self.mux_levels if m)) * B
        self.outputs = [src(w) for w in circ.outputs]


def evaluate_on(circ: Circuit, inputs, scheme, plan: Plan = None):
    """evaluate with an engine Scheme (GPU tensors or numpy arrays): one engine call per LEVEL (all gate kinds merged)."""
    B = inputs[0].shape[0]
    plan = plan or Plan(circ, B)
    assert plan.B == B and len(inputs) == plan.n_inputs
    lwe_len = inputs[0].shape[1]
    on_gpu = type(inputs[0]).__module__.startswith("torch")
    if on_gpu:
        import torch
        dev = inputs[0].device
        pool = torch.empty((plan.rows, lwe_len), dtype=inputs[0].dtype, device=dev)
        up = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)   # noqa: E731
    else:
        pool = np.empty((plan.rows, lwe_len), dtype=np.uint32)
        up = lambda a: a                                                                              # noqa: E731
    for s, x in enumerate(inputs):
        pool[s * B:(s + 1) * B] = x
    for (slot0, ngates, ops, ix, iy), mx in zip(plan.levels, plan.mux_levels):
        if ngates:
            scheme.gate_gather(up(ops), pool, up(ix), up(iy), pool[slot0 * B:(slot0 + ngates) * B])
        if mx:
            m0, nm, js, ja, jb, fl = mx
            scheme.mux_gather(pool, up(js), up(ja), up(jb), pool[m0 * B:(m0 + nm) * B], not_ab=up(fl))
    res = []
    for sl, negated in plan.outputs:
        o = pool[sl * B:(sl + 1) * B]
        o = o.clone() if on_gpu else o.copy()
        if negated:
            scheme.not_(o)
        res.append(o)
    return res


def evaluate_sharded(circ: Circuit, inputs, multi):
    """The B instances of a circuit sharded over the shards of a MultiScheme (one process, all GPUs): gates of different instances
    never meet, so every shard evaluates the WHOLE circuit on its contiguous slice of the instances (mkt_multi_shard_range), one host
    thread per shard, ciphertexts of a slice staying on that shard's device between levels; no exchange between shards.  inputs:
    host arrays, or GPU tensors (then all shards must sit on the tensors' device: logical shards)."""
    import threading
    B, n = inputs[0].shape[0], multi.nshards
    parts, errs = [None] * n, []

    def work(i):
        try:
            lo, hi = multi.shard_range(B, i)
            if lo < hi:
                sh = multi.shard(i)
                own = sh.get_stream()              # the shard's own stream: put back afterwards
                sh._user_stream = False            # GPU tensors: the shard enqueues on torch's current stream of its device, like Scheme
                try:
                    parts[i] = evaluate_on(circ, [x[lo:hi] for x in inputs], sh)
                    sh.synchronize()
                finally:
                    sh.set_stream(own or None)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(1, n)]
    for t in th:
        t.start()
    work(0)
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    done = [p for p in parts if p is not None]
    return [_cat([p[w] for p in done]) for w in range(len(circ.outputs))]


def ripple_adder(nbits):
    """nbits-bit adder: inputs a0..a{n-1}, b0..b{n-1} (LSB first), outputs s0..s{n-1}, carry"""
    c = Circuit()
    a = [c.input() for _ in range(nbits)]
    b = [c.input() for _ in range(nbits)]
    carry = None
    for i in range(nbits):
        p = c.XOR(a[i], b[i])
        g = c.AND(a[i], b[i])
        if carry is None:
            c.output(p); carry = g
        else:
            c.output(c.XOR(p, carry))
            carry = c.OR(g, c.AND(p, carry))
    c.output(carry)
    return c
