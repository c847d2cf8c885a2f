"""The measured table of README.md from the committed bench lines: python3 tools/readme_table.py [tag]   (profiles/<tag>_bench_*.json)"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def line(name):
    f = os.path.join(P, f"{tag}_bench_{name}.json")
    return json.loads(open(f).readline()) if os.path.exists(f) else None


def k(v):
    return f"{v / 1e3:.1f} k" if v >= 1e4 else (f"{v / 1e3:.2f} k" if v >= 1e3 else f"{v:.0f}")


rows = [("**KMS k = 2, N = 1024, ℓ = 2** (BASELINE configs[1], the driver's command)", "kms2_n1024", None), ("KMS2party (N = 2048)", "kms2party", None),
        ("KMS2partyblock", "kms2partyblock", "kms2partyblock_16384"), ("KMS4party", "kms4party", "kms4party_8192"), ("CGGIparam", "cggi", None),
        ("CGGI n = 630, N = 1024, ℓ = 2 (configs[0])", "cggi_l2", None), ("Blockparam (LMSS)", "lmss", "lmss_16384"),
        ("LMSS with RLWE length 2 (configs[4]), 16 384 gates", "lmss_k2_16384", None), ("CCS2party", "ccs2party", None), ("CCS8party", "ccs8party", None),
        ("CCS k = 8, N = 2048 (configs[3])", "ccs8_n2048", None), ("native MUX, headline shape", "kms2_n1024_mux", None)]
print("| workload | NAND gates/s (Float64 = the reference's arithmetic) | blind rotation | of the no-FMA f64 peak / of the kernel's own issue bound | EXACT mode (implementation) | C oracle, host threads |")
print("|---|---|---|---|---|---|")
for title, name, big in rows:
    d = line(name)
    if d is None:
        continue
    r = d["roofline"]
    ir = r.get("issue_roofline") or {}
    v = k(d["value"]) + (" MUX/s" if "mux" in name else "")
    if big and line(big):
        b = line(big)
        v += f" ({k(b['value'])} at {b['config']['batch_per_gpu']:,} gates)".replace(",", " ")
    ex = line((name[:-6] if name.endswith("_16384") else name) + "_exact")
    if name == "kms2_n1024" and "exact_mode" in d:
        e = d["exact_mode"]["implementations"]
        exs = f"{k(e['float64_pipe']['value'])} (Float64 pipe) / {k(e['integer_ntt']['value'])} (integer NTT)"
    elif ex:
        exs = f"{k(ex['value'])} ({'Float64 pipe' if 'fx_' in ex['roofline']['kernel'] else 'integer NTT'}" + (", 1 024 gates)" if name.endswith("_16384") else ")")
    else:
        exs = "—"
    cb = d.get("cpu_baseline")
    print(f"| {title} | {('**' + v + '**') if name == 'kms2_n1024' else v} | {d['kernels_ms_per_step']['blindrotate']:.1f} ms | {r['frac']:.3f} / {('%.2f' % ir['frac_at_sustained_clock']) if 'frac_at_sustained_clock' in ir else '—'} | {exs} | {(str(round(cb['value'])) + ' (' + str(cb['cores']) + ')') if cb else '—'} |")
a = line("adder8")
if a:
    print(f"| 1024 eight-bit adder circuits, one call per level | {k(a['circuit']['gates_per_s'] if 'circuit' in a and 'gates_per_s' in a['circuit'] else a['value'])} gates/s | — | — | — | — |")
d = line("kms2_n1024")
if d and "roofline_transform" in d:
    print()
    for x in d["roofline_transform"]:
        print(f"  {x['kernel']:22s} N={x['N']} W={x['ring_bits']} {x['direction']:8s} frac {x['frac']:.3f} (first launches {x['frac_first_launches']:.3f})" + (f"  issue {x['issue_roofline']['frac']:.2f}" if 'issue_roofline' in x else ""))
