#!/usr/bin/env python3
"""Instruction-class counts of the loaded library's rotation / transform kernels, from the gfx950 code objects INSIDE the built .so
(no recompile): profiles/isa_<build_id>.json, which bench.py reads under the same build-id rule as the PMC traffic.
    python tools/isa_report.py [path/to/libmktfhe_hip.so] [--out profiles/]
Per kernel: the instruction count of its largest loop (one CMux step / one block of key bits / one polynomial of a batched transform) by issue
class -- the classes of tools/int_probe.hip / valu_probe.hip (profiles/r05_int_probe.txt): `slow` = every Float64 instruction, integer multiplies,
v_min / v_max, three-operand and carry forms, 64-bit shifts and adds, compares + selects, DPP moves, lane permutes, conversions: 4.4 cycles per
wave instruction and SIMD; `fast` = 32-bit add / sub / logic / shift / move: 2.4 cycles (two or more resident waves)."""
import collections, json, os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FAST = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|mov_b32|mov_b64|bfe_u32|bfe_i32|bfi_b32|and_or_b32|or3_b32|xad_u32|lshl_or_b32|accvgpr_\w+|nop)(_e32|_e64)?$")
WATCH = ("blindrotate_k1_kernel", "blindrotate_blk_kernel", "blindrotate_kr_kernel", "ccs_blindrotate_kernel", "fx_blindrotate_kernel", "exact_kms_phase1_p2pf_kernel",
         "exact_kms_phase1_kernel", "exact_blindrotate_kernel", "ntt_fwd_kernel", "ntt_inv_kernel", "transform_fwd_kernel", "transform_inv_kernel", "exact_ccs_kernel",
         "exact_kms_block_phase1_kernel", "blindrotate_wide_kernel")


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def classify(op, line):
    if op.startswith(("s_", "ds_", "buffer_", "global_", "scratch_", "flat_")):
        return "scalar" if op.startswith("s_") else ("lds" if op.startswith("ds_") else ("scratch" if op.startswith("scratch_") else "vmem"))
    if "dpp" in line or "row_" in line or "quad_perm" in line:
        return "slow"
    return "fast" if FAST.match(op) else "slow"


def report(path):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(path)):
            f = os.path.join(td, f"co{i}.o")
            open(f, "wb").write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--demangle", "--mcpu=gfx950", f], capture_output=True, text=True).stdout
            cur, body = None, []
            funcs = {}
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    cur = m.group(1); funcs[cur] = []
                elif cur is not None and ln.strip():
                    funcs[cur].append(ln.strip())
            for name, lines in funcs.items():
                base = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
                if not any(w in base for w in WATCH) or "[clone" in name:
                    continue
                # instruction lines look like: "v_fma_f64 v[2:3], ... // 000000001234: ..." ; branch targets as "<sym+0xOFF>"
                ins = []
                for ln in lines:
                    m = re.match(r"^(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
                    if m:
                        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
                if not ins:
                    continue
                addr_idx = {a: k for k, (a, _, _) in enumerate(ins)}
                loops = []
                for k, (a, op, rest) in enumerate(ins):
                    if op.startswith(("s_cbranch", "s_branch")) and rest.strip().isdigit():
                        simm = int(rest.strip())
                        tgt = a + 4 + 4 * (simm - 65536 if simm >= 32768 else simm)          # signed 16-bit dword offset from the next instruction
                        if tgt in addr_idx and addr_idx[tgt] < k:
                            loops.append((addr_idx[tgt], k))
                a0, b0 = max(loops, key=lambda x: x[1] - x[0]) if loops else (0, len(ins) - 1)
                cls = collections.Counter()
                ops = collections.Counter()
                for a, op, rest in ins[a0:b0 + 1]:
                    c = classify(op, op + " " + rest)
                    cls[c] += 1
                    if "f64" in op and not op.startswith("v_cvt"):
                        cls["f64"] += 1
                    if c in ("slow", "fast"):
                        ops[op] += 1
                out[base] = {"loop_instructions": sum(v for k, v in cls.items() if k != "f64"), "slow": cls["slow"], "fast": cls["fast"], "f64": cls["f64"], "lds": cls["lds"],
                             "vmem": cls["vmem"], "scratch": cls["scratch"], "scalar": cls["scalar"], "whole_kernel_instructions": len(ins),
                             "cycles_per_wave_and_iteration": round(cls["slow"] * 4.4 + cls["fast"] * 2.4, 1), "top_valu": dict(ops.most_common(8))}
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    outdir = os.path.join(ROOT, "profiles")
    if "--out" in argv:
        i = argv.index("--out"); outdir = argv[i + 1]; del argv[i:i + 2]
    lib = argv[0] if argv else os.path.join(ROOT, "mktfhe_amd", "lib", "libmktfhe_hip.so")
    blob = open(lib, "rb").read()
    m = re.search(rb"MKT_BUILD_ID=([0-9a-f]{16})", blob)
    bid = m.group(1).decode() if m else None
    if bid is None:     # the id string itself is in .rodata; ask the library
        sys.path.insert(0, ROOT)
        import ctypes
        L = ctypes.CDLL(lib); L.mkt_build_id.restype = ctypes.c_char_p
        bid = L.mkt_build_id().decode()
    rep = {"build_id": bid, "classes": {"slow_cycles": 4.4, "fast_cycles": 2.4, "source": "tools/int_probe.hip, tools/valu_probe.hip (profiles/r05_int_probe.txt, r02_valu_probe.txt)"}, "kernels": report(lib)}
    dst = os.path.join(outdir, f"isa_{bid}.json")
    json.dump(rep, open(dst, "w"), indent=1, sort_keys=True)
    print(dst, len(rep["kernels"]), "kernels")
