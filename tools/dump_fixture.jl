# dump_fixture.jl -- UNEXECUTED in this repository's build environment (Julia is not installed there).
#
# Run next to a checkout of SNUCP/MKTFHE:   julia --project=. dump_fixture.jl <outdir> [KMS|CGGI|LMSS|CCS|KMSblock]
# It generates keys with the REFERENCE, evaluates NAND gates with the REFERENCE and writes everything the MI355X
# engine needs to replay the same computation bit for bit (tools/replay_fixture.py):
#   manifest.json            parameters + file table (name, dtype, shape)
#   *.bin                    raw little-endian arrays in the flat layouts of include/mktfhe.h
# Besides the gate outputs it dumps the intermediates of bootstrapping! (bootstrapping.jl:4-27) for every gate -- the
# mod-switched mask `atilde`, `btilde`, and the accumulator after blindrotate! (:25) -- so a mismatch localises to
# blind rotation or key switching.  The transform-domain keys are the reference's own Trans* values
# (MKT_FMT_F64_FFT) and the twiddle tables its own ffter tables (mkt_set_twiddles): the replay depends on nothing but
# IEEE-754 arithmetic.
include("src/MKTFHE.jl")
using .MKTFHE
import .MKTFHE: divbits, bits, zeronativepoly, RLWE, LWE, blindrotate!, keyswitch!

outdir = ARGS[1]; which = length(ARGS) > 1 ? ARGS[2] : "KMS"
mkpath(outdir)
files = String[]
function put(name, arr, dtype)
    open(joinpath(outdir, name * ".bin"), "w") do io; write(io, arr); end
    push!(files, "{\"name\":\"$name\",\"dtype\":\"$dtype\",\"shape\":[$(join(size(arr), ","))]}")
end
c64(v) = reinterpret(Float64, collect(ComplexF64, v))                      # interleaved (re, im)
polys(v) = reduce(vcat, (c64(p.coeffs) for p in v))                        # Vector{TransNativePoly} -> [len][M][2]
lweflat(c) = vcat(c.a, c.b)                                                # [a..., b]
trlwe(r) = vcat(c64(r.b.coeffs), (c64(a.coeffs) for a in r.a)...)          # TransRLWE -> (b, a[1:k])
# TransRGSW (gsw.jl:219-227) -> rows basketb.stack[1:l], basketa[1].stack[1:l], ...
rgsw(g) = reduce(vcat, (trlwe(r) for r in vcat(g.basketb.stack, (b.stack for b in g.basketa)...)))
# TransUniEnc (unienc.jl:92-99) -> d[1:l], then (f.stack[j].b, f.stack[j].a[1])
unienc(u) = vcat(polys(u.d), reduce(vcat, (trlwe(r) for r in u.f.stack)))
# ksk (Drows, N[, k]) of references -> [k][N][Drows][f][n+1]; #undef entries (block schemes: the embedded LWE key) -> zeros
function kskflat(k, n, f)
    D1, N = size(k, 1), size(k, 2); K = ndims(k) == 3 ? size(k, 3) : 1
    out = zeros(UInt32, (n + 1) * f * D1 * N * K); o = 0
    for c in 1:K, j in 1:N, d in 1:D1
        ok = ndims(k) == 3 ? isassigned(k, d, j, c) : isassigned(k, d, j)
        for t in 1:f
            if ok; lev = ndims(k) == 3 ? k[d, j, c] : k[d, j]; out[o+1:o+n+1] = lweflat(lev.stack[t]); end
            o += n + 1
        end
    end
    out
end
accflat(acc) = vcat(acc.b.coeffs, (a.coeffs for a in acc.a)...)

# bootstrapping! (bootstrapping.jl:4-27) with its intermediates exposed: the same statements, in the same order
function bootstrap_traced!(ctxt::LWE{T}, scheme::MKTFHE.TFHEscheme{R, S}) where {T, R, S}
    N, logN = scheme.N, trailing_zeros(scheme.N)
    tildea = divbits.(ctxt.a, bits(T) - logN - 1)
    tildeb = divbits(ctxt.b, bits(T) - logN - 1)
    tb0 = tildeb
    oneovereight = R(1) << (bits(R) - 3)
    b = zeronativepoly(N, R)
    if tildeb ≤ N
        for i = 1:N; b.coeffs[i] = i ≤ tildeb ? oneovereight : -oneovereight; end
    else
        tildeb -= R(N)
        for i = 1:N; b.coeffs[i] = i ≤ tildeb ? -oneovereight : oneovereight; end
    end
    acc = RLWE(b, [zeronativepoly(N, R) for _ = 1:scheme.k])
    blindrotate!(tildea, acc, scheme)
    after = accflat(acc)
    keyswitch!(ctxt, acc, scheme)
    tildea, UInt32(tb0), after
end
function nand_traced(c1::LWE{T}, c2::LWE{T}, scheme) where T                 # gate.jl:1-8
    res = LWE(T(1) << (bits(T) - 3) - c1.b - c2.b, @. -c1.a - c2.a)
    tr = bootstrap_traced!(res, scheme)
    res, tr
end

B = 8
bits01 = rand(Bool, 2B)
multikey = which in ("KMS", "CCS", "KMSblock")
params = Dict("KMS" => KMS2party, "CGGI" => CGGIparam, "LMSS" => Blockparam, "CCS" => CCS2party, "KMSblock" => KMS2partyblock)[which]
if multikey
    a = CRS(params)
    keys = [party_keygen(a, params) for _ = 1:params.k]
    lwekeys = first.(keys); btk = last.(keys)
    scheme = setup(a, btk, params)
    put("crs", polys(scheme.a), "f64")
    for (i, b) in enumerate(btk)
        put("brk$(i-1)", which == "CCS" ? reduce(vcat, (unienc(u) for u in b.brk)) : reduce(vcat, (rgsw(g) for g in b.brk)), "f64")
        put("ksk$(i-1)", kskflat(b.ksk, scheme.n, scheme.kskpar.l), "u32")
        if which != "CCS"
            put("rlk_d$(i-1)", polys(b.rlk.d), "f64"); put("rlk_f$(i-1)", reduce(vcat, (trlwe(r) for r in b.rlk.f.stack)), "f64")
        end
        put("pubkey$(i-1)", polys(b.b), "f64")
        put("lwekey$(i-1)", lwekeys[i].key, "u32")
    end
    # inputs that involve every party: NAND folds over one fresh encryption per party (test/KMS.jl:29-34)
    fold(off) = begin
        cs = [lwe_ith_encrypt(bits01[off], i, lwekeys[i], params) for i = 1:params.k]
        acc = cs[1]; for i = 2:params.k; acc = NAND(acc, cs[i], scheme); end; acc
    end
    xs = [fold(j) for j = 1:B]; ys = [fold(B + j) for j = 1:B]
else
    lwekey, ringkey, scheme = setup(params)
    put("brk0", reduce(vcat, (rgsw(g) for g in scheme.btk.brk)), "f64")
    put("ksk0", kskflat(scheme.btk.ksk, scheme.n, scheme.kskpar.l), "u32"); put("lwekey0", lwekey.key, "u32")
    xs = [lwe_encrypt(bits01[j], lwekey, params) for j = 1:B]; ys = [lwe_encrypt(bits01[B+j], lwekey, params) for j = 1:B]
end
f = scheme.ffter
put("psi", c64(f.Ψ), "f64"); put("psiinv", c64(f.Ψinv), "f64"); put("roots", c64(f.roots), "f64"); put("rootsinv", c64(f.rootsinv), "f64")
traced = [nand_traced(xs[j], ys[j], scheme) for j = 1:B]
zs = first.(traced)
@assert all(lweflat(zs[j]) == lweflat(NAND(xs[j], ys[j], scheme)) for j = 1:B)   # the traced path IS bootstrapping!
put("x", reduce(vcat, lweflat.(xs)), "u32"); put("y", reduce(vcat, lweflat.(ys)), "u32"); put("nand", reduce(vcat, lweflat.(zs)), "u32")
put("atilde", reduce(vcat, (UInt32.(t[2][1]) for t in traced)), "u32"); put("btilde", [t[2][2] for t in traced], "u32")
put("acc", reduce(vcat, (t[2][3] for t in traced)), eltype(traced[1][2][3]) == UInt64 ? "u64" : "u32")
put("bits", UInt8.(bits01), "u8")
W = eltype(traced[1][2][3]) == UInt64 ? 64 : 32
g(x, f) = hasproperty(x, f) ? getproperty(x, f) : nothing
gsw = multikey ? (which == "CCS" ? nothing : btk[1].gswpar) : scheme.gswpar
lev = which in ("KMS", "KMSblock") ? btk[1].levpar : nothing
uni = which == "CCS" ? scheme.unipar : (which in ("KMS", "KMSblock") ? btk[1].unipar : nothing)
kind = Dict("CGGI" => 0, "LMSS" => 1, "CCS" => 2, "KMS" => 3, "KMSblock" => 4)[which]
l_(p) = p === nothing ? 0 : p.l; lb_(p) = p === nothing ? 0 : p.logB
blk = which in ("LMSS", "KMSblock")
pj = "\"scheme\":$kind,\"n\":$(scheme.n),\"N\":$(scheme.N),\"k\":$(scheme.k),\"W\":$W,\"l_gsw\":$(l_(gsw)),\"logB_gsw\":$(lb_(gsw)),\"l_lev\":$(l_(lev)),\"logB_lev\":$(lb_(lev)),\"l_uni\":$(l_(uni)),\"logB_uni\":$(lb_(uni)),\"f\":$(scheme.kskpar.l),\"logD\":$(scheme.kskpar.logB),\"blk_len\":$(blk ? scheme.ℓ : 0),\"blk_d\":$(blk ? scheme.d : 0)"
open(joinpath(outdir, "manifest.json"), "w") do io
    write(io, "{\"format\":\"mktfhe-fixture-2\",\"producer\":\"julia-reference\",\"params\":{$pj},\"batch\":$B,\"files\":[$(join(files, ","))]}")
end
println("wrote fixture to ", outdir)
