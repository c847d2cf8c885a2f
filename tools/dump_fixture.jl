# dump_fixture.jl -- UNEXECUTED in this repository's build environment (Julia is not installed there).
#
# Run next to a checkout of SNUCP/MKTFHE:   julia --project=. dump_fixture.jl <outdir> [KMS|CGGI]
# It generates keys with the REFERENCE, evaluates NAND gates with the REFERENCE and writes everything the
# MI355X engine needs to replay the same computation bit for bit (tools/replay_fixture.py):
#   manifest.json            parameters + file table (name, dtype, shape)
#   *.bin                    raw little-endian arrays in the flat layouts of include/mktfhe.h
# The transform-domain keys are the reference's own Trans* values (MKT_FMT_F64_FFT) and the twiddle tables are
# the reference's own ffter tables (mkt_set_twiddles), so the replay depends on nothing but IEEE-754 arithmetic.
include("src/MKTFHE.jl")
using .MKTFHE

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
# TransRGSW (gsw.jl:219-227) -> rows basketb.stack[1:l], basketa[1].stack[1:l]; each row (b, a[1])
rgsw(g) = reduce(vcat, (vcat(c64(r.b.coeffs), c64(r.a[1].coeffs)) for r in vcat(g.basketb.stack, g.basketa[1].stack)))
ksk2(k) = (D1, N = size(k); reduce(vcat, (vcat(k[d, j].stack[t].a, k[d, j].stack[t].b) for j in 1:N for d in 1:D1 for t in 1:k[1, 1].l)))

if which == "KMS"
    params = KMS2party
    a = CRS(params)
    keys = [party_keygen(a, params) for _ = 1:params.k]
    lwekeys = first.(keys); btk = last.(keys)
    scheme = setup(a, btk, params)
    f = scheme.ffter
    put("psi", c64(f.Ψ), "f64"); put("psiinv", c64(f.Ψinv), "f64"); put("roots", c64(f.roots), "f64"); put("rootsinv", c64(f.rootsinv), "f64")
    put("crs", polys(scheme.a), "f64")
    for (i, b) in enumerate(btk)
        put("brk$(i-1)", reduce(vcat, (rgsw(g) for g in b.brk)), "f64")
        put("ksk$(i-1)", ksk2(b.ksk), "u32")
        put("rlk_d$(i-1)", polys(b.rlk.d), "f64")
        put("rlk_f$(i-1)", reduce(vcat, (vcat(c64(r.b.coeffs), c64(r.a[1].coeffs)) for r in b.rlk.f.stack)), "f64")
        put("pubkey$(i-1)", polys(b.b), "f64")
        put("lwekey$(i-1)", lwekeys[i].key, "u32")
    end
    B = 8
    bits = rand(Bool, 2B)
    xs = [lwe_ith_encrypt(bits[j], 1, lwekeys[1], params) for j = 1:B]
    ys = [lwe_ith_encrypt(bits[B+j], 2, lwekeys[2], params) for j = 1:B]
    zs = [NAND(xs[j], ys[j], scheme) for j = 1:B]
    put("x", reduce(vcat, lweflat.(xs)), "u32"); put("y", reduce(vcat, lweflat.(ys)), "u32"); put("nand", reduce(vcat, lweflat.(zs)), "u32")
    put("bits", UInt8.(bits), "u8")
    g, lv, u = btk[1].gswpar, btk[1].levpar, btk[1].unipar
    pj = "\"scheme\":3,\"n\":$(params.n),\"N\":$(params.N),\"k\":$(params.k),\"W\":64,\"l_gsw\":$(g.l),\"logB_gsw\":$(g.logB),\"l_lev\":$(lv.l),\"logB_lev\":$(lv.logB),\"l_uni\":$(u.l),\"logB_uni\":$(u.logB),\"f\":$(scheme.kskpar.l),\"logD\":$(scheme.kskpar.logB),\"blk_len\":0,\"blk_d\":0"
else
    params = CGGIparam
    lwekey, ringkey, scheme = setup(params)
    f = scheme.ffter
    put("psi", c64(f.Ψ), "f64"); put("psiinv", c64(f.Ψinv), "f64"); put("roots", c64(f.roots), "f64"); put("rootsinv", c64(f.rootsinv), "f64")
    put("brk0", reduce(vcat, (rgsw(g) for g in scheme.btk.brk)), "f64")
    k3 = scheme.btk.ksk
    put("ksk0", ksk2(k3[:, :, 1]), "u32"); put("lwekey0", lwekey.key, "u32")
    B = 8
    bits = rand(Bool, 2B)
    xs = [lwe_encrypt(bits[j], lwekey, params) for j = 1:B]; ys = [lwe_encrypt(bits[B+j], lwekey, params) for j = 1:B]
    zs = [NAND(xs[j], ys[j], scheme) for j = 1:B]
    put("x", reduce(vcat, lweflat.(xs)), "u32"); put("y", reduce(vcat, lweflat.(ys)), "u32"); put("nand", reduce(vcat, lweflat.(zs)), "u32")
    put("bits", UInt8.(bits), "u8")
    g = scheme.gswpar
    pj = "\"scheme\":0,\"n\":$(params.n),\"N\":$(params.N),\"k\":1,\"W\":32,\"l_gsw\":$(g.l),\"logB_gsw\":$(g.logB),\"l_lev\":0,\"logB_lev\":0,\"l_uni\":0,\"logB_uni\":0,\"f\":$(scheme.kskpar.l),\"logD\":$(scheme.kskpar.logB),\"blk_len\":0,\"blk_d\":0"
end
open(joinpath(outdir, "manifest.json"), "w") do io
    write(io, "{\"format\":\"mktfhe-fixture-1\",\"producer\":\"julia-reference\",\"params\":{$pj},\"batch\":8,\"files\":[$(join(files, ","))]}")
end
println("wrote fixture to ", outdir)
