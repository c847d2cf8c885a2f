#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X gate-bootstrapping engine.

Metric (BASELINE.json): NAND gate-bootstraps/s at k parties; transform ("NTT") HBM GB/s vs 8 TB/s.
A "step" = one mkt_gate_batch call: B independent NAND gates (gate.jl:1-8 -> bootstrapping!,
bootstrapping.jl:4-27) on ciphertexts already resident in HBM.  Default workload = BASELINE.json
configs[1]: KMS multi-key k=2, N=1024, batch=1024 (synthetic shape, l_gsw=2 -- SURVEY.md 0.5).

  python bench.py [--gpus N --steps K --warmup W] [--workload NAME --batch B]

--scaling weak (default): --batch gates PER GPU; --scaling strong: --batch gates IN TOTAL, contiguous shards per rank
(mktfhe_amd.distributed.shard_slices) -- BASELINE.json configs[2] / [3] are fixed-total batches:
  python bench.py --gpus 8 --workload kms4party --batch 65536 --scaling strong
  python bench.py --gpus 8 --workload ccs8_n2048 --batch 8192 --scaling strong

--gpus N > 1 without a torch.distributed.run environment: this process -- before it touches the GPU -- starts N
fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one GPU each), waits for
them, relays rank 0's JSON line and exits non-zero if any rank failed.  Under torch.distributed.run it is a rank.
Gates shard across ranks with no data-path collective (keys replicated per GPU); RCCL carries the timing barrier,
the max-over-ranks of the step time and the `ranks_seen` census only.

--launcher inproc: ONE process drives all N GPUs through the library's multi-device evaluator (mkt_multi_*: keys uploaded
once and replicated device to device, the batch cut into N contiguous shards, one host thread per GPU, no collective, no
torch.distributed) -- the shape of a Julia caller of the ccall shim, which is one process.  The default launcher stays one
process per GPU under torch.distributed (what the driver's census expects).

--workload adder8 --instances I: circuit throughput -- I independent 8-bit ripple-carry adders evaluated level by level with
one engine call per level (mkt_gate_batch_gather: mixed gate kinds, operands gathered from a ciphertext pool in HBM), beside
the flat NAND rate of the same parameter set measured in the same run.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline           -- the kernel that costs the time, blindrotate_k1_kernel (93 % of a step): bound = f64 VALU issue
                        WITHOUT FMA (the reference's arithmetic rounds after every multiply and every add), achieved =
                        algorithmic flop per launch / average launch time from HIP events on the engine's stream;
                        issue_roofline inside it = the kernel's MEASURED VALU instruction count (committed PMC pass of this
                        build) priced at the class costs of its own instruction mix (profiles/isa_<build_id>.json).
  roofline_transform -- the batched negacyclic transforms HBM -> HBM (BASELINE.json's second metric), forward and
                        inverse at N = 1024 and N = 2048, working set >= 4 GiB, against 8 TB/s.
  secondary          -- the same measurement on the reference's own two-party set KMS2party (params.jl:47-53).
  exact_mode         -- the headline workload in MKT_ARITH_EXACT under both implementations (Float64 FMA transforms over 16-bit key
                        limbs; the integer NTT the north star names), the same words, N = 1 only.
  cpu_baseline       -- the C oracle (restatement of the reference CPU path, F64REF) timed on this box's host cores
                        on a bounded sample of the same workload; kind "port".
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    "kms2_n1024": ("KMS2party_N1024_l2", "KMS k=2, N=1024, l_gsw=2 (BASELINE.json configs[1], synthetic shape)"),
    "kms2party": ("KMS2party", "KMS2party k=2, N=2048, l_gsw=3 (src/tfhe/params.jl:47-53)"),
    "cggi": ("CGGIparam", "CGGIparam single-key, N=1024, l=3 (src/tfhe/params.jl:1-6)"),
    "cggi_l2": ("CGGI_N1024_l2", "CGGI single-key, N=1024, n=630, l=2 (BASELINE.json configs[0], synthetic gadget)"),
    "lmss": ("Blockparam", "Blockparam LMSS block-binary single-key, N=1024 (src/tfhe/params.jl:8-13)"),
    "lmss_k2": ("Blockparam_k2", "LMSS block-binary, N=1024, RLWE length k=2 (BASELINE.json configs[4], synthetic shape)"),
    "kms2partyblock": ("KMS2partyblock", "KMS2partyblock k=2, N=2048, block-binary keys (src/tfhe/params.jl:87-93)"),
    "kms4party": ("KMS4party", "KMS4party k=4, N=2048 (src/tfhe/params.jl:55-61)"),
    "ccs2party": ("CCS2party", "CCS2party k=2, N=1024 (src/tfhe/params.jl:15-21)"),
    "ccs8party": ("CCS8party", "CCS8party k=8, N=1024 (src/tfhe/params.jl:31-37)"),
    "ccs8_n2048": ("CCS8party_N2048", "CCS k=8, N=2048 (BASELINE.json configs[3], synthetic shape)"),
    "adder8": ("KMS2party_N1024_l2", "8-bit ripple-carry adder circuits on the BASELINE configs[1] parameter set (KMS k=2, N=1024, l_gsw=2), one engine call per circuit level"),
}

# Parameter sets whose own output noise leaves less than 6 sigma of decryption margin.  PREDICTED: the per-gate failure probability the
# noise theory gives for gates on all-party ciphertexts (tools/noise_theory.py: input phase error of a NAND = sqrt(2) x the output sigma
# of the previous level, Gaussian tail; profiles/r03_noise_theory_vs_measured.md).  Every other set must decrypt every gate.
PREDICTED_FAILURE = {"KMS2party_N1024_l2": 1.5e-3, "KMS2party": 1.5e-5, "KMS2partyblock": 5e-6, "KMS8party": 1e-3,
                     "CCS2party": 4.6e-3, "CCS4party": 0.20, "CCS8party": 0.033, "CCS16party": 0.37}
# MEASURED in the reference's own arithmetic on 16 384 gates (round 4, tools/ks_big_check.py: 7 / 5 wrong, each wrong gate's words equal
# to the CPU oracle's): on the 64-bit ring the Float64 transform error is heavier-tailed than the Gaussian the sigma predicts.  These
# rates are reported beside the prediction, never in its place, and they only widen the accepted band when THIS run's wrong gates are
# shown to be the oracle's own words (verify_wrong_gates, inside the cpu_baseline leg) -- a kernel regression cannot hide in the band.
REFERENCE_ARITH_FAILURE = {"KMS2party": 4.3e-4, "KMS2partyblock": 3.1e-4}


def allowed_wrong(pname, checked, verified=False):
    """wrong decryptions a run may show: 3 x the predicted rate (+3); 3 x the rate measured in the reference's arithmetic only
    when the wrong gates were verified word for word against the oracle"""
    rate = PREDICTED_FAILURE.get(pname)
    if rate is None:
        return 0
    if verified:
        rate = max(rate, REFERENCE_ARITH_FAILURE.get(pname, 0.0))
    return int(3 * rate * checked) + 3


def decrypt_fields(pname, errs, checked):
    return {"decrypt_ok": errs <= allowed_wrong(pname, checked), "decrypt_errors": errs, "decrypt_checked": checked,
            "decrypt_failure_rate_measured": errs / max(checked, 1), "decrypt_failure_rate_predicted": PREDICTED_FAILURE.get(pname, 0.0),
            "decrypt_failure_rate_reference_arithmetic": REFERENCE_ARITH_FAILURE.get(pname), "wrong_gates_match_oracle": None}


def verify_wrong_gates(line, mk, p, crs, keys, allc, B, res, want, limit=64):
    """the checker's part of the decryption gate (runs inside the cpu_baseline leg, F64REF only): every gate of this rank's shard that
    decrypts wrongly is recomputed by the CPU oracle; if its words are the oracle's words, the failure is the parameter set's noise in
    the reference's arithmetic and the band of REFERENCE_ARITH_FAILURE applies; if any differs, decrypt_ok is false whatever the count"""
    got = mk.lwe_decrypt(res, keys if p.multikey else keys[0], p)
    wrong = np.flatnonzero(got != want)[:limit]
    if len(wrong) == 0:
        return
    from helpers import oracle_scheme
    so = oracle_scheme(p, crs, keys)
    same = all(np.array_equal(so.gate(0, allc[int(g)], allc[B + int(g)]), res[int(g)]) for g in wrong)
    line["wrong_gates_match_oracle"] = bool(same)
    line["wrong_gates_checked"] = int(len(wrong))
    # the widened band is for VERIFIED wrong gates only: every wrong gate of the whole job must have been compared with the oracle -- on a multi-rank run
    # that is the case only when all of them fell into this rank's shard (and none beyond `limit`); otherwise the predicted band stays
    all_verified = bool(same) and int(np.count_nonzero(got != want)) == len(wrong) and line["decrypt_errors"] == len(wrong)
    line["wrong_gates_all_verified"] = all_verified
    line["decrypt_ok"] = bool(same) and line["decrypt_errors"] <= allowed_wrong(line["config"]["params"], line["decrypt_checked"], verified=all_verified)


EXACT_ARITH = {False: "EXACT (integer NTT, residues mod 131063*2^13+1 and 131066*2^13+1)",
               True: "EXACT (Float64 FMA transforms over centered 16-bit key limbs, products rounded to the exact integer; phase 2 and tables on the integer NTT)"}

# no-FMA f64 vector peak: 256 CUs x 4 SIMDs x 16 lanes/clk (a wave64 v_add_f64 / v_mul_f64 issues over 4 cycles) x
# 2.4 GHz = 39.3 TFLOP/s (half of the 78.6 TFLOP/s FMA datasheet figure; MI355X_MICROARCH.md: FP32 vector 157.3)
PEAK_F64_NOFMA_TFLOPS = 39.3216


def profiled_counters(kernel_prefix, workload, want=("FETCH_SIZE", "WRITE_SIZE"), near_ms=None, variant=None):
    """per-launch PMC means of `kernel_prefix` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*_bench_<workload>[_<batch> | _mux]_pmc.txt; FETCH_SIZE / WRITE_SIZE in KiB).  PMC cannot be collected from inside the
    timed run, so these are the profiled figures, newest round last; None when no profile is committed.  A profile holds one row
    per (kernel, grid): the row whose mean duration is nearest `near_ms` (the launch time measured now) is the one of THIS launch
    shape -- refused if it is more than 30 % off (another batch size) --, without `near_ms` the longest-running grid."""
    import glob
    import re
    best, best_err = None, None
    pat = re.compile(r"^r\d+[a-z]?_bench_" + re.escape(workload) + (r"_mux" if variant == "mux" else r"(_\d+)?") + r"_pmc\.txt$")
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{workload}*_pmc.txt"))) if pat.match(os.path.basename(f))]
    newest = max((int(re.match(r"r(\d+)", os.path.basename(f)).group(1)) for f in files), default=None)
    for f in files:
        if int(re.match(r"r(\d+)", os.path.basename(f)).group(1)) != newest:     # earlier rounds profiled other kernels
            continue
        rows = {}
        bid = None
        for ln in open(f):
            if ln.startswith("# build_id:"):                 # which library these passes measured (tools/pmc_pass.sh)
                bid = ln.split()[2]
            if not (ln.startswith(kernel_prefix + ",") or ln.startswith(kernel_prefix + ">")):   # `kernel<first template argument`
                continue
            parts = ln.rsplit(",", 5)          # kernel, grid, counter, mean, ms, n
            if len(parts) == 6:
                rows.setdefault(parts[1], {})[parts[2]] = float(parts[3])
                rows[parts[1]]["ms:" + parts[2]] = float(parts[4])
                rows[parts[1]]["kernel"] = parts[0]
        for grid, r in sorted(rows.items(), key=lambda kv: kv[1].get("ms:" + want[0], 0.0)):   # the longest-running grid last
            if not all(w in r for w in want):
                continue
            if near_ms is None:
                best = (r, os.path.relpath(f, ROOT), bid)
            else:
                err = abs(r["ms:" + want[0]] / near_ms - 1.0)
                if err <= 0.30 and (best_err is None or err <= best_err + 0.02):    # newer rounds win ties
                    best, best_err = (r, os.path.relpath(f, ROOT), bid), err
    return best


_ISA = {}


def isa_of_build():
    """instruction-class counts of the loaded library's kernels (tools/isa_report.py reads them out of the built .so): profiles/isa_<build_id>.json,
    used only when it is the file of THE LIBRARY THIS PROCESS HAS LOADED"""
    import mktfhe_amd as mk
    bid = mk.build_id()
    if bid not in _ISA:
        f = os.path.join(ROOT, "profiles", f"isa_{bid}.json")
        _ISA[bid] = json.load(open(f)) if os.path.exists(f) else None
    return _ISA[bid]


def issue_roofline(c, avg_ms, wave_steps=None):
    """what the SIMDs could at best do with the instruction stream this kernel really executes: VALU wave-instructions of one launch (PMC SQ_INSTS_VALU, the
    committed pass of this build) x the cycles its instruction MIX costs per wave instruction and SIMD (slow class 4.4: every Float64 instruction, integer
    multiplies, three-operand / carry / 64-bit forms, DPP and lane permutes, compares + selects; fast class 2.4: 32-bit add / sub / logic / shift / move --
    tools/int_probe.hip; mix = the kernel's main loop in the ISA of this build, tools/isa_report.py) / 1024 SIMDs / clock, against the launch time"""
    isa = isa_of_build()
    if not isa or "SQ_INSTS_VALU" not in c or c.get("kernel") not in isa["kernels"]:
        return None
    k = isa["kernels"][c["kernel"]]
    valu = k["slow"] + k["fast"]
    cyc = (k["slow"] * 4.4 + k["fast"] * 2.4) / valu
    bound_ms = c["SQ_INSTS_VALU"] * cyc / (1024 * 2.4e9) * 1e3
    o = {"valu_wave_instr_per_launch": c["SQ_INSTS_VALU"], "slow_class_fraction": k["slow"] / valu, "f64_fraction": k["f64"] / valu, "cycles_per_wave_instr": cyc,
         "issue_bound_ms_at_2.4GHz": bound_ms, "frac": bound_ms / avg_ms, "isa": f"profiles/isa_{isa['build_id']}.json",
         "note": "time the measured VALU instruction count needs at the class costs of tools/int_probe.hip on 1024 SIMDs at 2.4 GHz / the measured launch time: the distance from THIS kernel's own issue bound"}
    if "GRBM_GUI_ACTIVE" in c:
        ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["ms:GRBM_GUI_ACTIVE"] * 1e-3) / 1e9
        o["frac_at_sustained_clock"] = o["frac"] * 2.4 / ghz
    if wave_steps:
        o["valu_instr_per_wave_and_cmux"] = c["SQ_INSTS_VALU"] / wave_steps
        o["f64_instr_per_wave_and_cmux"] = c["SQ_INSTS_VALU"] / wave_steps * k["f64"] / valu
    for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS"):
        if n in c and "SQ_WAVE_CYCLES" in c:
            o[n.lower() + "_per_wave_cycle"] = c[n] / c["SQ_WAVE_CYCLES"]
    return o


def attach_profile(r, prof, achieved=None, peak=None, wave_steps=None):
    """profile-derived fields of a roofline object -- only when the committed passes measured THE LIBRARY THIS PROCESS HAS LOADED (same
    mkt_build_id: same kernel sources and flags); otherwise traffic stays null and the line says the profile is stale"""
    if not prof:
        return r
    c, src, bid = prof
    import mktfhe_amd as mk
    r["traffic_source"] = src
    if bid is None or bid != mk.build_id():
        r["traffic_stale"] = True
        r["traffic_stale_note"] = f"profile build_id {bid}, loaded library {mk.build_id()}: numbers of another build are not quoted"
        return r
    r["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0     # gfx950: FETCH_SIZE counts half (guide, HBM section)
    r["traffic_build_id"] = bid
    if achieved is not None and "GRBM_GUI_ACTIVE" in c:       # sum over the 8 XCDs / 8 / duration = the clock the part held under this kernel
        ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["ms:GRBM_GUI_ACTIVE"] * 1e-3) / 1e9
        r["sustained_clock_ghz"] = ghz
        r["frac_at_sustained_clock"] = achieved / (peak * ghz / 2.4)
    if achieved is not None and "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        r["valu_active_per_wave_cycle"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    if "avg_launch_ms" in r:
        ir = issue_roofline(c, r["avg_launch_ms"], wave_steps)
        if ir:
            r["issue_roofline"] = ir
    return r


def effective_cpus():
    """CPUs this process can really use: affinity mask and cgroup CPU quota (the GPU boxes expose 256 logical CPUs
    under a 16-CPU quota; more threads than the quota only add contention)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def spawn_ranks(n, argv):
    """parent of a `--gpus n` run: start n rank processes and relay rank 0's line.  Runs BEFORE anything touches the
    GPU in this process (no torch.cuda / HIP call has been made; the children are fresh interpreters)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # MKT_BENCH_PIN=1: each rank sees only its own GPU (HIP_VISIBLE_DEVICES = rank; the rank then uses device 0).  Off by
        # default: RCCL wants every peer visible for its topology search.
        if os.environ.get("MKT_BENCH_PIN") == "1" and os.environ.get("MKT_BENCH_SHARE_GPU") != "1":
            env.update(HIP_VISIBLE_DEVICES=str(r), MKT_BENCH_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        sys.exit(1)
    sys.exit(0)


def blindrotate_flop(mk, p, B):
    """algorithmic f64 flop of one blind-rotation launch over B gates: per CMux 2l+2
    transforms of M points (5 M log2 M butterfly flop + 6 M twist flop each) + the transform-domain MACs (8 flop per
    complex multiply-add, 6 per multiply)"""
    M = p.N // 2
    lg = int(np.log2(M))
    if p.scheme == mk.CCS:
        # bootstrapping.jl:234-328, step (party idx, key bit i), np = idx + 1 mask polynomials so far: for each of the
        # np + 1 input polynomials l transforms for u, v (:279-294: 2 MACs each), one inverse for v (:297-300), l
        # transforms for w (:313-320: 2 MACs each); then np + 1 monomial products and inverses (:322-324)
        T = 5 * M * lg + 6 * M
        l = p.l_uni
        per_poly = 2 * l * T + 2 * T + 4 * l * 8 * M + 6 * M
        return sum((idx + 2) * per_poly for idx in range(p.k)) * p.n * B, 1
    rows = (1 + (p.k - 1) * p.l_lev) if p.scheme in (mk.KMS, mk.KMS_BLOCK) else 1
    l = max(p.l_gsw, 1)
    LB = max(p.blk_len, 1)     # block schemes: one decomposition + (kr+1)(l+1) transforms per block of LB key bits
    kr = 1 if p.multikey else p.k          # RLWE length of the rotation (KMS rotates length-1 RLWE rows)
    T = 5 * M * lg + 6 * M
    per_iter = (kr + 1) * (l + 1) * T + LB * ((kr + 1) ** 2 * l * 8 * M + (kr + 1) * (8 if LB > 1 else 6) * M)
    return per_iter * (p.n // LB) * rows * B, rows


def make_scheme(mk, p, local, need_host_keys, arith=0):
    """synthetic keys: pinned seed 1 (benchmark only); the large keys are generated on the GPU (mkt_keygen_device: the
    host generator's words), the host copies exist only where the CPU baseline needs them"""
    if p.multikey:
        crs = mk.CRS(p, 1)
        keys = [mk.party_keygen(crs, p, party=i, secrets_only=not need_host_keys, deterministic_seed=1) for i in range(p.k)]
        return crs, keys, mk.setup(p, keys=keys, a=crs, device=local, arith=arith)
    keys = [mk.PartyKeys(p, secrets_only=not need_host_keys, deterministic_seed=1)]
    return None, keys, mk.setup(p, keys=keys[0], device=local, arith=arith)[1]


def make_inputs(mk, torch, p, keys, sch, B, rank, dev, kind):
    """2*B distinct input ciphertexts.  A fresh multi-key encryption has only its own party's mask block populated, a
    gate between ciphertexts of one party keeps it that way, and the blind rotation skips zero mask words
    (bootstrapping.jl:413, :261): such gates do a fraction of the work.  `mixed` inputs therefore involve EVERY party, as
    inside any multi-party circuit: each is an untimed NAND fold over k distinct fresh encryptions, one per party
    (test/KMS.jl:29-34 folds its inputs the same way).  `fresh` = single-party first-level encryptions."""
    rng = np.random.default_rng(2 + rank)

    def fresh(nct, seed0):
        b = rng.integers(0, 2, nct).astype(bool)
        ct = np.empty((nct, p.lwe_len), dtype=np.uint32)
        for j in range(nct):
            ct[j] = mk.lwe_ith_encrypt(int(b[j]), j % p.nparty, keys[j % p.nparty], p, deterministic_seed=seed0 + j)
        return b, torch.from_numpy(ct.view(np.int32)).to(dev)

    seed0 = 10_000_000 * (rank + 1)
    if kind == "mixed":
        def folded(s0):
            b, ct = fresh(p.nparty * B, s0)            # ct[i::k] are the B ciphertexts under party i
            acc_b, acc = b[0::p.nparty].copy(), ct[0::p.nparty].contiguous()
            for i in range(1, p.nparty):
                acc = mk.NAND(acc, ct[i::p.nparty].contiguous(), sch)
                acc_b = ~(acc_b & b[i::p.nparty])
            return acc_b, acc
        bx, x = folded(seed0)
        by, y = folded(seed0 + 5_000_000)
        if p.nparty == 1:
            x, y = x.clone(), y.clone()
        return np.concatenate([bx, by]), x, y
    bits, fct = fresh(2 * B, seed0)
    return bits, fct[:B].clone(), fct[B:].clone()


def time_gates(mk, torch, dist, D, sch, p, keys, x, y, bits, B, steps, warmup, world, red_dev, op="nand"):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; max over ranks.
    op "mux": the native MUX(x, y, z) with z = x rotated by one position in the batch (two blind rotations + one key switch per
    gate; z must not be x itself: AND-linear(NOT x, x) has an all-zero mask and its rotation would be skipped)"""
    out = torch.empty_like(x)
    z = torch.roll(x, 1, 0).contiguous() if op == "mux" else None
    step = (lambda: mk.NAND(x, y, sch, out=out)) if op == "nand" else (lambda: mk.MUX(x, y, z, sch, out=out))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    sch.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    rot_ms, rot_n = sch.kernel_ms(1)
    ks_ms, _ = sch.kernel_ms(2)
    p2_ms, _ = sch.kernel_ms(4)
    sch.enable_timing(False)
    mine = elapsed
    if world > 1:                                   # world == 1 also marks a leg only rank 0 runs: no collective there
        elapsed = D.max_over_ranks(elapsed, device=red_dev)
    per_rank = [1e3 * mine / max(steps, 1)]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=red_dev)
        t[dist.get_rank()] = per_rank[0]
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.cpu()]
    res = out.cpu().numpy().view(np.uint32)
    want = ~(bits[:B] & bits[B:]) if op == "nand" else np.where(bits[:B], bits[B:], np.roll(bits[:B], 1))
    got = mk.lwe_decrypt(res, keys if p.multikey else keys[0], p)
    errs = int(np.count_nonzero(got != want))
    if world > 1:                                   # wrong decryptions of the whole job, not of rank 0's shard
        e = torch.tensor([errs, B], dtype=torch.int64, device=red_dev)
        dist.all_reduce(e)
        errs_all, checked_all = int(e[0]), int(e[1])
    else:
        errs_all, checked_all = errs, B
    return dict(elapsed=elapsed, rot_ms=rot_ms, rot_n=rot_n, ks_ms=ks_ms, p2_ms=p2_ms, res=res, decrypt_errors=errs,
                decrypt_errors_all=errs_all, decrypt_checked_all=checked_all, per_rank_ms=per_rank)


def rot_roofline(mk, p, B, t, workload, kern=None, variant=None):
    """the roofline object of the dominant kernel of this workload; `kern` = the name the engine reports for the kernel it
    actually launched (mkt_last_kernel_name), so the line and the rocprof trace name the same kernel"""
    flop_step, rows = blindrotate_flop(mk, p, B)              # one step = ceil(B / 8192) launches (the engine's workspace chunk)
    launches_per_step = max(1, -(-B // 8192))
    flop = flop_step / launches_per_step
    avg_ms = t["rot_ms"] / max(t["rot_n"], 1)
    achieved = flop / (avg_ms * 1e-3) / 1e12
    kern = kern or ("ccs_blindrotate_kernel" if p.scheme == mk.CCS else ("blindrotate_kr_kernel" if (not p.multikey and p.k > 1) else "blindrotate_k1_kernel"))
    r = {"bound": "f64-valu-nofma", "kernel": kern, "achieved": achieved, "peak": PEAK_F64_NOFMA_TFLOPS,
         "unit": "TFLOP/s", "frac": achieved / PEAK_F64_NOFMA_TFLOPS, "traffic": None,
         "algorithmic_flop_per_launch": flop, "rotations_per_launch": rows * B / launches_per_step, "cmux_per_rotation": p.n // max(p.blk_len, 1),
         "avg_launch_ms": avg_ms, "launches_timed": t["rot_n"],
         "frac_of_fma_peak": achieved / (2.0 * PEAK_F64_NOFMA_TFLOPS),
         "peak_note": "256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz, mul and add issued separately (no FMA: bit parity with the reference); frac_of_fma_peak prices the same flop against the 78.6 TFLOP/s datasheet FMA rate"}
    # the kernel's first template argument is log2 M: the row of THIS transform size (a profile may also hold the secondary leg's)
    prof = profiled_counters(f"mktd::{kern}<{int(np.log2(p.N // 2))}", "kms2_n1024" if workload == "adder8" else workload, want=("FETCH_SIZE", "WRITE_SIZE"), near_ms=avg_ms, variant=variant)
    # wave x CMux-step count of one launch (4 points per thread: M / 256 waves per rotation) -- CCS runs one workgroup per ciphertext over sum_idx (idx + 2) polynomials per key bit
    waves = max(1, (p.N // 2) // 256)
    wave_steps = None if p.scheme == mk.CCS else rows * B / launches_per_step * waves * (p.n // max(p.blk_len, 1))
    return attach_profile(r, prof, achieved, PEAK_F64_NOFMA_TFLOPS, wave_steps)


def transform_roofline(mk, torch, local, dev):
    """BASELINE.json metric 2: batched forward (fft.jl:57-63) and inverse (fft.jl:74-81) transforms streamed HBM -> HBM at
    N = 1024 and 2048, >= 4 GiB per launch (>> 256 MiB Infinity Cache), on both rings: algorithmic bytes per transform
    N (W/8 + 8) -- 16 N on the 64-bit ring (KMS), 12 N on the 32-bit ring (CGGI / LMSS / CCS: BASELINE.md 4)"""
    out = []
    for W in (64, 32):
        for N in (1024, 2048):
            wb = W // 8
            per = N * (wb + 8)
            p = (mk.KMS2party if W == 64 else mk.CGGIparam).scaled(n=8, N=N)
            nb = (4 << 30) // per
            if W == 64:
                polys = torch.randint(-2**31, 2**31 - 1, (nb, 2 * N), dtype=torch.int32, device=dev).view(torch.int64)
            else:
                polys = torch.randint(-2**31, 2**31 - 1, (nb, N), dtype=torch.int32, device=dev)
            tr = torch.empty((nb, N // 2), dtype=torch.complex128, device=dev)
            back = torch.empty_like(polys)
            # Sustained rate: the part needs tens of milliseconds of back-to-back work to reach the clock it then holds -- 5 launches timed
            # right after 2 warm-ups read 5-25 % below what 30 launches after 30 untimed ones do (the integer legs most: tools/legs.py
            # --warm 2 --reps 5, profiles/r05_experiments.txt item 11).  `frac` is the sustained figure; the cold one is kept beside it.
            for arith, reps, warm in ((mk.ARITH_F64REF, 30, 30), (mk.ARITH_EXACT, 30, 30)):
                if arith == mk.ARITH_EXACT and W == 32 and N == 2048:
                    continue                                   # the integer legs are reported at three shapes; keeps the default run short
                s = mk.Scheme(p, device=local, arith=arith)
                for direction in ("forward", "inverse"):
                    fn = (lambda: s.transform_fwd(polys, out=tr)) if direction == "forward" else (lambda: s.transform_inv(tr, out=back))
                    fn(); fn()                                 # first touches of a fresh working set
                    torch.cuda.synchronize()
                    s.enable_timing(True)
                    for _ in range(5):
                        fn()
                    ms0, cnt0 = s.kernel_ms(3)
                    s.enable_timing(False)
                    for _ in range(max(warm - 7, 0)):
                        fn()
                    s.enable_timing(True)
                    for _ in range(reps):
                        fn()
                    ms, cnt = s.kernel_ms(3)
                    s.enable_timing(False)
                    achieved = nb * per / (ms / cnt * 1e-3) / 1e9
                    cold = nb * per / (ms0 / cnt0 * 1e-3) / 1e9
                    if arith == mk.ARITH_F64REF:
                        kern = "transform_fwd_kernel" if direction == "forward" else "transform_inv_kernel"
                        prefix = f"mktd::{kern}<{int(np.log2(N)) - 1}, unsigned {'long' if W == 64 else 'int'}"
                        e = {"bound": "hbm", "kernel": kern}
                    else:
                        kern = "ntt_fwd_kernel" if direction == "forward" else "ntt_inv_kernel"
                        prefix = f"mktd::{kern}<{int(np.log2(N))}, unsigned {'long' if W == 64 else 'int'}"
                        # these legs are bound by integer ISSUE, not by HBM: VALU instructions of the kernel's loop per thread and polynomial (8 points; slow
                        # class = multiplies, v_min, three-operand and carry forms at 4.4 cycles per wave instruction and SIMD, fast class = add / sub / logic /
                        # moves at 2.4: tools/int_probe.hip), counted in the code object of THIS build (tools/isa_report.py -> profiles/isa_<build_id>.json)
                        e = {"bound": "int32-valu-issue", "kernel": kern, "arith": "EXACT (integer NTT, residues mod 131063*2^13+1 and 131066*2^13+1)",
                             "bound_note": "integer issue, not HBM: `frac` stays the BASELINE metric (achieved / 8 TB/s), `issue_roofline.frac` is the distance from this kernel's own bound"}
                        isa = isa_of_build()
                        kname = f"mktd::{kern}<{int(np.log2(N))}, unsigned {'long' if W == 64 else 'int'}" + (", false>" if direction == "forward" else ">")
                        if isa and kname in isa["kernels"]:
                            ki = isa["kernels"][kname]
                            cyc_poly = (ki["slow"] * 4.4 + ki["fast"] * 2.4) * (N // 512)          # one wave carries 512 points
                            issue_peak = 256 * 4 * 2.4e9 / cyc_poly * per / 1e9                      # GB/s of algorithmic bytes at which the VALU is full (2.4 GHz)
                            e["issue_roofline"] = {"peak": issue_peak, "unit": "GB/s", "frac": achieved / issue_peak, "valu_instr_per_thread_and_polynomial": {"slow_class": ki["slow"], "fast_class": ki["fast"]},
                                                   "isa": f"profiles/isa_{isa['build_id']}.json",
                                                   "peak_note": "1024 SIMDs x 2.4 GHz / (slow x 4.4 + fast x 2.4 cycles per 512 points) x algorithmic bytes per polynomial; the part holds 2.0-2.2 GHz inside these kernels (tools/ntt_clock_probe.hip, profiles/r05_ntt_clock_probe.txt: another device, so not priced in here)"}
                    e.update({"direction": direction, "N": N, "ring_bits": W, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                              "frac_first_launches": cold / 8000.0, "launches_timed": cnt, "launches_before": warm,
                              "traffic": None, "algorithmic_bytes_per_launch": nb * per, "bytes_per_transform": per, "transforms_per_launch": nb, "avg_launch_ms": ms / cnt})
                    prof = profiled_counters(prefix, "kms2_n1024")      # the headline workload's PMC passes include these legs
                    out.append(attach_profile(e, prof))
                s.close()
            del polys, tr, back
            torch.cuda.empty_cache()
    return out


def cpu_baseline(p, crs, keys, allc, B, res, args, check_bits):
    """the C oracle ("port": F64REF restatement of the reference CPU path) timed on this box's host cores on a bounded sample of
    the same workload -- one gate per thread over all the cores this process may use, and one thread alone --
    compiled for these cores (-O3 -march=native, contraction off: same bits, checked against the GPU result)"""
    from oracle import oracle as ORA
    flags = ORA.use_native_build()
    from helpers import oracle_scheme
    so = oracle_scheme(p, crs, keys)
    cores = effective_cpus()                   # host threads this process may actually run on (cgroup quota aware)
    npilot = min(cores, B)                     # pilot round (one gate per thread) sizes the sample to ~10 s of CPU work
    t0 = time.perf_counter()
    so.gate_batch(0, allc[:npilot], allc[B:B + npilot], threads=npilot)
    pilot = time.perf_counter() - t0
    sample = args.cpu_sample or cores * max(1, min(int(10.0 / max(pilot, 1e-3)), 64))
    sample = min(B, sample)
    cores = min(cores, sample)
    xs, ys = allc[:sample], allc[B:B + sample]
    t0 = time.perf_counter()
    ref = so.gate_batch(0, xs, ys, threads=cores)
    dt = time.perf_counter() - t0
    n1 = max(1, min(sample, int(3.0 / max(pilot, 1e-3))))          # ~3 s on one thread
    t0 = time.perf_counter()
    so.gate_batch(0, xs[:n1], ys[:n1], threads=1)
    dt1 = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "gates/s", "cores": cores, "kind": "port",
           "sample": f"{sample} NAND gates of the same workload, C oracle (F64REF restatement of the reference CPU path), {cores} threads (the host's CPU quota), one gate per thread",
           "seconds": dt, "one_thread": {"value": n1 / dt1, "unit": "gates/s", "cores": 1, "sample": f"{n1} gates", "seconds": dt1},
           "build": flags}
    bitexact = bool(np.array_equal(ref, res[:sample])) if check_bits else None
    return out, bitexact


def fx_rot_roofline(mk, p, B, t, kern, workload):
    """Float64-issue roofline of the EXACT blind rotation on the Float64 pipe (fx_exact.hip): f64 VALU instructions the algorithm needs per CMux and transform
    point -- 2l digit transforms and 2 W/16 limb inverses of T = 3 (log2 M - 2) + 4 fused operations per point (6 per butterfly, the two product-free stages 2 per
    point) + 6 for twist / untwist, conversion and rounding, and 2l x 2 W/16 complex multiply-adds of 4 -- against the rate at which one MI355X issues Float64
    instructions (16 lanes / clk / SIMD; a fused multiply-add is ONE issue slot).  One "launch" here is the blind rotation of the whole batch = one kernel launch
    per chip-fill of rotations."""
    M = p.N // 2
    lg = int(np.log2(M))
    l, NL = p.l_gsw, p.W // 16
    T = 3 * (lg - 2) + 4
    per_point = 2 * l * (T + 6) + 2 * NL * (T + 6) + 2 * l * 2 * NL * 4
    rows = (1 + (p.k - 1) * p.l_lev) if p.scheme == mk.KMS else 1
    launches_per_step = max(1, -(-B // 8192))
    instr = per_point * M * p.n * rows * B / launches_per_step
    avg_ms = t["rot_ms"] / max(t["rot_n"], 1)
    peak = 256 * 4 * 16 * 2.4e9 / 1e12
    ach = instr / (avg_ms * 1e-3) / 1e12
    fill = 256 * 4 if (p.W == 64 and lg <= 9) else 1 << 62             # launch_fx_blindrotate: one launch per chip-fill on the 64-bit ring up to N = 1024, one launch elsewhere
    r = {"bound": "f64-valu-issue (FMA)", "kernel": kern, "achieved": ach, "peak": peak, "unit": "T f64 lane-instr/s", "frac": ach / peak, "traffic": None,
         "algorithmic_f64_instr_per_launch": instr, "f64_instr_per_point_and_cmux": per_point, "avg_launch_ms": avg_ms, "launches_timed": t["rot_n"],
         "kernel_launches_per_batch": -(-int(rows * B / launches_per_step) // fill),
         "avg_launch_note": "one blind rotation of the batch: a kernel launch per chip-fill of rotations (L2 locality of the key stream) + the rows' residue transforms + phase 2",
         "peak_note": "256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz = 39.3 T f64 lane-instructions/s (fused multiply-add, add or multiply: one issue slot each; 78.6 TFLOP/s if every slot were an FMA)"}
    prof = profiled_counters(f"mktd::{kern}<{lg}", workload + "_exact", want=("FETCH_SIZE", "WRITE_SIZE"))
    if prof is not None:                      # the profile's rows are per kernel launch (one chip-fill); the line's unit is the whole batch
        c = dict(prof[0]); n = r["kernel_launches_per_batch"]
        for k_ in list(c):
            if k_ != "kernel":
                c[k_] = c[k_] * n
        prof = (c, prof[1], prof[2])
        r["traffic_note"] = f"counters of one kernel launch x {n} launches per batch"
    waves = max(1, M // 256)
    return attach_profile(r, prof, ach, peak, rows * B / launches_per_step * waves * p.n)


def exact_rot_roofline(mk, p, B, t, kern, workload):
    """integer-issue roofline of the EXACT (two-prime NTT) blind-rotation kernels: VALU instructions the arithmetic itself
    needs per launch (ntt_exact.hip; DESIGN.md 2: a two-residue butterfly is 14 instructions forward / 16 inverse, a lazy
    Montgomery multiply-accumulate over both residues 12, conversion / lift per coefficient as counted below) against the rate
    at which one MI355X issues 32-bit integer multiply-class instructions (tools/valu_probe.hip: v_mul_lo_u32, v_mul_hi_u32,
    v_mad_u64_u32 at 4.4 cycles per wave64 instruction per SIMD = 16 lanes/clk, like a v_add_f64; plain 32-bit ALU ops at
    2.5).  Model: 60 % of the instruction stream is multiply-class (quarter rate), 40 % full rate -> a mean of 3.6 cycles per
    wave instruction; peak = 256 CU x 4 SIMD x 64 lanes / 3.6 cycles x 2.4 GHz."""
    if kern == "fx_blindrotate_kernel":
        return fx_rot_roofline(mk, p, B, t, kern, workload)
    N = p.N
    lg = int(np.log2(N))
    bf_fwd, bf_inv, mac = 14, 16, 12
    fwd = (N // 2) * lg * bf_fwd + 6 * N          # butterflies + input conversion of N coefficients (both residues)
    inv = (N // 2) * lg * bf_inv + 14 * N         # butterflies + CRT lift, sign and reduction mod 2^W per coefficient
    rows = (1 + (p.k - 1) * p.l_lev) if p.scheme in (mk.KMS, mk.KMS_BLOCK) else 1
    l = max(p.l_gsw, 1)
    LB = max(p.blk_len, 1)
    if p.scheme == mk.CCS:
        lu = p.l_uni
        per_poly = 2 * lu * fwd + 2 * inv + 4 * lu * mac * N + 8 * N
        instr = sum((idx + 2) * per_poly for idx in range(p.k)) * p.n * B
        rows = 1
    elif p.W == 64:      # split tables: 2l forward and (low, high) x (b, a) = 4 inverses per block, 2 x 2 x 2l MACs per key bit, 4 monomial products per key bit of a block
        # (what the ALGORITHM needs: until round 5 the KMS_block kernel transformed the digits once per key bit and this line counted that too)
        per_blk = 2 * l * fwd + 4 * inv + LB * (8 * l * mac * N) + (LB * 4 * mac * N if LB > 1 else 0) + 16 * N
        instr = per_blk * (p.n // LB) * rows * B
    else:                # 32-bit ring, RLWE length kr: (kr+1) l forward and kr+1 inverses per block, (kr+1)^2 l MACs per key bit, monomial product in the transform domain
        kr = p.k
        per_blk = (kr + 1) * l * fwd + (kr + 1) * inv + LB * ((kr + 1) ** 2 * l * mac * N + (kr + 1) * mac * N) + 4 * (kr + 1) * N
        instr = per_blk * (p.n // LB) * rows * B
    launches_per_step = max(1, -(-B // 8192))
    avg_ms = t["rot_ms"] / max(t["rot_n"], 1)
    peak = 256 * 4 * 64 / 3.6 * 2.4e9 / 1e12           # T lane-instructions / s
    ach = instr / launches_per_step / (avg_ms * 1e-3) / 1e12
    r = {"bound": "int32-valu-issue", "kernel": kern, "achieved": ach, "peak": peak, "unit": "T lane-instr/s", "frac": ach / peak, "traffic": None,
         "algorithmic_instr_per_launch": instr / launches_per_step, "avg_launch_ms": avg_ms, "launches_timed": t["rot_n"],
         "peak_note": "256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 3.6 cycles per wave instruction (60 % slow class -- multiplies, v_min, three-operand and carry forms -- at 4.4, 40 % add / sub / logic at 2.4, whatever the number of resident waves: tools/int_probe.hip, profiles/r05_int_probe.txt)"}
    prof = profiled_counters(f"mktd::{kern}<{lg}", workload + "_exact", want=("FETCH_SIZE", "WRITE_SIZE"), near_ms=avg_ms)
    if prof is None and kern == "exact_kms_phase1_kernel":       # the engine reports the family name; at l_gsw = 2 the launched kernel is the paired-transform member with key rows requested ahead
        prof = profiled_counters(f"mktd::exact_kms_phase1_p2pf_kernel<{lg}", workload + "_exact", want=("FETCH_SIZE", "WRITE_SIZE"), near_ms=avg_ms)
        if prof is not None:
            r["kernel"] = "exact_kms_phase1_p2pf_kernel"
    return attach_profile(r, prof, ach, peak)


def run_circuit(mk, torch, p, keys, sch, args, dev, flat_rate):
    """--workload adder8: I independent 8-bit ripple-carry adders, ciphertexts resident in HBM, one mkt_gate_batch_gather per
    level (mixed XOR / AND / OR, NOT-free here).  A step = one evaluation of all I circuits."""
    from mktfhe_amd import circuit as CI
    circ = CI.ripple_adder(8)
    inst = args.instances
    plan = CI.Plan(circ, inst)
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, (circ.n_inputs, inst)).astype(bool)
    # inputs: one fresh encryption per (wire, bit), replicated over the instances (the work does not depend on the values)
    # operand a under party 0, operand b under party 1 (mod k): already the first level's gates span parties, as every later one does
    party = lambda i: (i // 8) % p.nparty                                     # noqa: E731  (ripple_adder: inputs 0-7 = a, 8-15 = b)
    enc = {(i, v): mk.lwe_ith_encrypt(v, party(i), keys[party(i)], p, deterministic_seed=9000 + 2 * i + v) for i in range(circ.n_inputs) for v in (0, 1)}
    inputs = [torch.from_numpy(np.stack([enc[(i, int(bits[i, j]))] for j in range(inst)]).view(np.int32)).to(dev) for i in range(circ.n_inputs)]
    for _ in range(args.warmup):
        outs = CI.evaluate_on(circ, inputs, sch, plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = CI.evaluate_on(circ, inputs, sch, plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    want = circ.plain(bits)
    ok = 0
    for o, w in zip(outs, want):
        ok += int(np.count_nonzero(mk.lwe_decrypt(o.cpu().numpy().view(np.uint32), keys if p.multikey else keys[0], p) == w))
    checked = len(want) * inst
    gates = plan.gates * args.steps
    return {"circuit": "8-bit ripple-carry adder", "gates_per_instance": plan.gates // inst, "levels": len(plan.levels), "instances": inst,
            "level_widths": [n * inst for _, n, _, _, _ in plan.levels], "engine_calls_per_evaluation": len(plan.levels),
            "value": gates / dt, "unit": "gates/s", "ms_per_evaluation": 1e3 * dt / args.steps,
            "flat_nand_gates_per_s": flat_rate, "vs_flat": gates / dt / flat_rate if flat_rate else None,
            "output_bits_correct": ok, "output_bits_checked": checked}


def main_inproc(args):
    """--launcher inproc: one process, N GPUs through the library's multi-device evaluator (no torch.distributed)"""
    import torch
    import mktfhe_amd as mk
    n = args.gpus
    share = os.environ.get("MKT_BENCH_SHARE_GPU") == "1" or torch.cuda.device_count() < n
    devices = [0] * n if share else list(range(n))
    pname, desc = WORKLOADS[args.workload]
    p = getattr(mk, pname)
    total = args.batch if args.scaling == "strong" else n * args.batch
    arith = mk.ARITH_EXACT if args.arith == "exact" else mk.ARITH_F64REF
    need_host_keys = not args.no_cpu_baseline
    if p.multikey:
        crs = mk.CRS(p, 1)
        keys = [mk.party_keygen(crs, p, party=i, secrets_only=not need_host_keys, deterministic_seed=1) for i in range(p.k)]
    else:
        crs, keys = None, [mk.PartyKeys(p, secrets_only=not need_host_keys, deterministic_seed=1)]
    sch = mk.setup_multi(p, devices, keys=keys if p.multikey else keys[0], a=crs, arith=arith)
    torch.cuda.set_device(devices[0])
    dev = torch.device("cuda", devices[0])
    bits, x, y = make_inputs(mk, torch, p, keys, sch, total, 0, dev, args.inputs)     # the folds run through the multi evaluator too
    torch.cuda.synchronize()
    out = torch.empty_like(x)
    for _ in range(args.warmup):
        sch.gate(0, x, y, out=out)
    shards = [sch.shard(i) for i in range(n)]
    for sh in shards:
        sh.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sch.gate(0, x, y, out=out)                   # synchronous: returns when every shard is done
    elapsed = time.perf_counter() - t0
    per_shard, rot, ks, p2, rot_n = [], 0.0, 0.0, 0.0, 0
    for sh in shards:
        w, _ = sh.kernel_ms(0)
        per_shard.append(w / max(args.steps, 1))
    r0, rot_n = shards[0].kernel_ms(1); k0, _ = shards[0].kernel_ms(2); q0, _ = shards[0].kernel_ms(4)
    kern = shards[0].last_kernel_name()
    for sh in shards:
        sh.enable_timing(False)
    res = out.cpu().numpy().view(np.uint32)
    want = ~(bits[:total] & bits[total:])
    errs = int(np.count_nonzero(mk.lwe_decrypt(res, keys if p.multikey else keys[0], p) != want))
    lo, hi = sch.shard_range(total, 0)
    t = {"rot_ms": r0, "rot_n": rot_n}
    line = {
        "metric": "NAND gate-bootstraps/sec", "value": total * args.steps / elapsed, "unit": "gates/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64" if args.arith == "f64ref" else "u32x2-residue", "data": "synthetic",
        "config": {"workload": desc, "params": pname, "parties": p.k, "N": p.N, "n": p.n, "ring_bits": p.W, "l_gsw": p.l_gsw,
                   "batch_per_gpu": hi - lo, "batch_total": total, "op": "NAND", "inputs": args.inputs,
                   "arith": "F64REF" if args.arith == "f64ref" else "EXACT (integer NTT, residues mod 131063*2^13+1 and 131066*2^13+1)",
                   "sharding": "gates across GPUs, keys replicated device to device (hipMemcpyPeer), one process, one host thread per GPU",
                   "launcher": "inproc (mkt_multi_*)", "devices": devices,
                   "io": f"inputs and outputs are ONE array each on device {devices[0]}; the other shards' slices travel by peer copy inside the timed region"},
        "ranks_seen": n, "per_rank_ms_per_step": per_shard,
        **decrypt_fields(pname, errs, total),
        "kernels_ms_per_step": {"shard": 0, "blindrotate": r0 / max(args.steps, 1), "kms_phase2": q0 / max(args.steps, 1), "keyswitch": k0 / max(args.steps, 1)},
    }
    line["roofline"] = rot_roofline(mk, p, hi - lo, t, args.workload, kern) if args.arith == "f64ref" else exact_rot_roofline(mk, p, hi - lo, t, kern, args.workload)
    if args.arith == "exact":
        line["config"]["arith"] = EXACT_ARITH[kern == "fx_blindrotate_kernel"]
        line["dtype"] = "f64-fma-exact" if kern == "fx_blindrotate_kernel" else "u32x2-residue"
    if not args.no_cpu_baseline:
        allc = np.concatenate([x.cpu().numpy(), y.cpu().numpy()]).view(np.uint32)
        line["cpu_baseline"], line["oracle_bitexact"] = cpu_baseline(p, crs, keys, allc, total, res, args, args.arith == "f64ref")
        if args.arith == "f64ref":
            verify_wrong_gates(line, mk, p, crs, keys, allc, total, res, want)
    print(json.dumps(line), flush=True)
    sch.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--workload", default="kms2_n1024", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --batch gates per GPU; strong: --batch gates in total, sharded over the ranks")
    ap.add_argument("--launcher", default="ranks", choices=["ranks", "inproc"], help="ranks: one process per GPU under torch.distributed (default); inproc: one process, all GPUs through mkt_multi_*")
    ap.add_argument("--instances", type=int, default=1024, help="--workload adder8: independent circuit instances")
    ap.add_argument("--op", default="nand", choices=["nand", "mux"], help="nand (default: the metric); mux: the native MUX gate, two blind rotations + one key switch per gate (ranks launcher)")
    ap.add_argument("--arith", default="f64ref", choices=["f64ref", "exact"], help="f64ref: the reference's Float64 transforms, bit-identical to it (default); exact: MKT_ARITH_EXACT, exact products -- Float64 FMA transforms over 16-bit key limbs where the engine has that kernel (CGGI, KMS phase 1), the integer NTT over two 30-bit primes elsewhere (all five schemes); MKT_EXACT_IMPL=0 forces the NTT")
    ap.add_argument("--inputs", default="mixed", choices=["mixed", "fresh"], help="mixed: every ciphertext involves all k parties (default); fresh: single-party first-level encryptions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the transform legs (roofline_transform)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="gates in the CPU-baseline sample (0 = auto)")
    args = ap.parse_args()

    if args.launcher == "inproc":
        return main_inproc(args)                   # one process; nothing is spawned or exec'ed
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])       # never returns; nothing has touched the GPU in this process

    import torch
    import torch.distributed as dist
    import mktfhe_amd as mk
    from mktfhe_amd import distributed as D
    rank, world, local = D.env()
    # MKT_BENCH_SHARE_GPU=1 (+ gloo): exercise the multi-process path on a 1-GPU box; also the fallback when a node
    # exposes fewer devices than ranks
    backend = os.environ.get("MKT_BENCH_BACKEND", "nccl")
    share = os.environ.get("MKT_BENCH_SHARE_GPU") == "1"
    if share:
        local, backend = 0, os.environ.get("MKT_BENCH_BACKEND", "gloo")
    if "MKT_BENCH_DEVICE" in os.environ:            # a rank pinned to one visible device (MKT_BENCH_PIN)
        local = int(os.environ["MKT_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D.init_process_group(backend, device=dev)     # "nccl" is RCCL on ROCm; rendezvous + timing barrier only (falls back to gloo if RCCL cannot start)
    backend = D.ACTIVE_BACKEND or backend
    red_dev = dev if backend == "nccl" else "cpu"
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    pname, desc = WORKLOADS[args.workload]
    p = getattr(mk, pname)
    B = args.batch
    if args.scaling == "strong":                    # this rank's contiguous shard of the fixed total batch
        lo, hi = D.shard_slices(args.batch, world)[rank]
        B = hi - lo
        if B == 0:
            raise SystemExit(f"rank {rank}: empty shard (batch {args.batch} over {world} ranks)")
    need_host_keys = rank == 0 and not args.no_cpu_baseline
    arith = mk.ARITH_EXACT if args.arith == "exact" else mk.ARITH_F64REF
    crs, keys, sch = make_scheme(mk, p, local, need_host_keys, arith)
    bits, x, y = make_inputs(mk, torch, p, keys, sch, B, rank, dev, args.inputs)
    torch.cuda.synchronize()
    allc = np.concatenate([x.cpu().numpy(), y.cpu().numpy()]).view(np.uint32)
    t = time_gates(mk, torch, dist, D, sch, p, keys, x, y, bits, B, args.steps, args.warmup, world, red_dev, args.op)
    res = t["res"]
    kern = sch.last_kernel_name()

    line = None
    if rank == 0:
        total_batch = args.batch if args.scaling == "strong" else world * B
        gates = total_batch * args.steps
        line = {
            "metric": "NAND gate-bootstraps/sec" if args.op == "nand" else "MUX gates/sec (native: two blind rotations + one key switch per gate)", "value": gates / t["elapsed"], "unit": "gates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64" if args.arith == "f64ref" else "u32x2-residue", "data": "synthetic",
            "config": {"workload": desc, "params": pname, "parties": p.k, "N": p.N, "n": p.n, "ring_bits": p.W,
                       "l_gsw": p.l_gsw, "batch_per_gpu": B, "batch_total": total_batch, "op": args.op.upper(), "inputs": args.inputs, "arith": "F64REF" if args.arith == "f64ref" else "EXACT (integer NTT, residues mod 131063*2^13+1 and 131066*2^13+1)", "sharding": "gates across GPUs, keys replicated",
                       "launcher": "ranks (one process per GPU, torch.distributed)"},
            "ranks_seen": ranks_seen, "per_rank_ms_per_step": t["per_rank_ms"], "rendezvous_backend": backend if world > 1 else None,
            # Wrong decryptions are the parameter set's own output noise (profiles/r03_noise_theory_vs_measured.md).  Sets whose margin is
            # >= 6 sigma must decrypt EVERY gate; the noisy ones are held to 3 x the PREDICTED rate, and to 3 x the rate measured in the
            # reference's arithmetic only once this run's wrong gates are shown to be the oracle's words (verify_wrong_gates below)
            **decrypt_fields(pname, t["decrypt_errors_all"], t["decrypt_checked_all"]),
            "kernels_ms_per_step": {"blindrotate": t["rot_ms"] / max(args.steps, 1), "kms_phase2": t["p2_ms"] / max(args.steps, 1),
                                    "keyswitch": t["ks_ms"] / max(args.steps, 1)},
        }
        nrot = B if args.op == "nand" else 2 * B            # a MUX gate is two blind rotations
        line["roofline"] = rot_roofline(mk, p, nrot, t, args.workload, kern, "mux" if args.op == "mux" else None) if args.arith == "f64ref" else exact_rot_roofline(mk, p, nrot, t, kern, args.workload)
        if args.arith == "exact":            # which implementation of the EXACT arithmetic the engine chose for this shape (option exact_impl; MKT_EXACT_IMPL seeds it)
            fx = kern == "fx_blindrotate_kernel"
            line["config"]["arith"] = EXACT_ARITH[fx]
            line["dtype"] = "f64-fma-exact" if fx else "u32x2-residue"

    # ---- circuit throughput (--workload adder8): rank 0, N = 1 ----
    if rank == 0 and world == 1 and args.workload == "adder8":
        line["circuit"] = run_circuit(mk, torch, p, keys, sch, args, dev, line["value"])

    # ---- secondary leg: the reference's own 2-party set, same measurement, short ----
    if not args.no_secondary and args.workload == "kms2_n1024":
        p2 = mk.KMS2party
        crs2, keys2, sch2 = make_scheme(mk, p2, local, False)
        bits2, x2, y2 = make_inputs(mk, torch, p2, keys2, sch2, B, rank, dev, args.inputs)
        t2 = time_gates(mk, torch, dist, D, sch2, p2, keys2, x2, y2, bits2, B, max(2, args.steps // 4), 1, world, red_dev)
        if rank == 0:
            st2 = max(2, args.steps // 4)
            sec = {"workload": WORKLOADS["kms2party"][1], "params": "KMS2party", "value": world * B * st2 / t2["elapsed"], "unit": "gates/s",
                   "steps": st2, "ms_per_step": 1e3 * t2["elapsed"] / st2, "batch_per_gpu": B,
                   "decrypt_errors": t2["decrypt_errors"], "decrypt_checked": B,
                   "kernels_ms_per_step": {"blindrotate": t2["rot_ms"] / st2, "kms_phase2": t2["p2_ms"] / st2, "keyswitch": t2["ks_ms"] / st2}}
            rr = rot_roofline(mk, p2, B, t2, "kms2party", sch2.last_kernel_name())
            sec["roofline"] = {k: rr[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic")}
            line["secondary"] = sec
        sch2.close()
        del x2, y2
        torch.cuda.empty_cache()

    # ---- the same workload in the integer-NTT arithmetic the north star names (MKT_ARITH_EXACT), short; rank 0 only ----
    if rank == 0 and not args.no_secondary and args.arith == "f64ref" and args.workload == "kms2_n1024":
        crsx, keysx, schx = make_scheme(mk, p, local, False, mk.ARITH_EXACT)
        bitsx, xx, yx = make_inputs(mk, torch, p, keysx, schx, B, rank, dev, args.inputs)
        stx = max(2, args.steps // 2)
        impls = {}
        outs = []
        for name, impl in (("float64_pipe", 1), ("integer_ntt", 0)):          # the same context, keys and inputs; both implementations give the same words
            schx.set_option("exact_impl", impl)
            tx = time_gates(mk, torch, None, D, schx, p, keysx, xx, yx, bitsx, B, stx, 1, 1, red_dev)
            outs.append(tx["res"])
            impls[name] = {"arith": EXACT_ARITH[schx.last_kernel_name() == "fx_blindrotate_kernel"], "dtype": "f64-fma-exact" if schx.last_kernel_name() == "fx_blindrotate_kernel" else "u32x2-residue",
                           "value": B * stx / tx["elapsed"], "unit": "gates/s", "steps": stx,
                           "ms_per_step": 1e3 * tx["elapsed"] / stx, "batch_per_gpu": B, "decrypt_errors": tx["decrypt_errors"], "decrypt_checked": B,
                           "kernels_ms_per_step": {"blindrotate": tx["rot_ms"] / stx, "keyswitch": tx["ks_ms"] / stx},
                           "roofline": exact_rot_roofline(mk, p, B, tx, schx.last_kernel_name(), args.workload)}
        best = max(impls, key=lambda k_: impls[k_]["value"])
        line["exact_mode"] = dict(impls[best], implementation=best, implementations=impls, implementations_word_identical=bool(np.array_equal(outs[0], outs[1])),
                                  fx_error_bound=schx.get_metric("fx_bound"), fx_key_max_transform_magnitude=schx.get_metric("fx_kmax"),
                                  note="MKT_ARITH_EXACT: valid ciphertexts, bitwise unrelated to the Float64 reference's (not the parity mode), word-identical to a big-integer restatement (tests/ref_exact.py) in BOTH implementations -- the two-prime integer NTT the north star names, and Float64 FMA transforms over 16-bit key limbs whose rounding error is proven below 1/2 for the loaded keys (fx_error_bound; DESIGN.md section 2)")
        schx.close()
        del xx, yx
        torch.cuda.empty_cache()

    # ---- transform roofline legs (BASELINE.json metric 2): rank 0, at every world size ----
    if rank == 0 and not args.no_roofline:
        line["roofline_transform"] = transform_roofline(mk, torch, local, dev)

    # ---- CPU baseline leg (oracle, "port"): rank 0, at every world size (the other ranks wait at the final barrier) ----
    if rank == 0 and not args.no_cpu_baseline and args.op == "nand":
        line["cpu_baseline"], line["oracle_bitexact"] = cpu_baseline(p, crs, keys, allc, B, res, args, args.arith == "f64ref")   # EXACT words differ from the Float64 reference by construction (checked against big-integer arithmetic in tests)
        if args.arith == "f64ref":
            verify_wrong_gates(line, mk, p, crs, keys, allc, B, res, ~(bits[:B] & bits[B:]))

    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        # the other ranks sleep on the rendezvous store while rank 0 ran its extra legs (a collective would have them spin on the
        # host cores the CPU baseline is timed on); then one last barrier
        try:
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("mkt_rank0_done", "1")
            else:
                store.wait(["mkt_rank0_done"])
        except Exception:       # noqa: BLE001  (a torch without this private accessor: the barrier alone)
            pass
        dist.barrier()
        dist.destroy_process_group()
    sch.close()


if __name__ == "__main__":
    main()
