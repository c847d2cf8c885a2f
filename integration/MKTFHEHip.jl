# MKTFHEHip.jl -- reference-side binding of the MI355X engine (the shim of INTEGRATION.md as a file).
#
# Put next to src/MKTFHE.jl in a checkout of SNUCP/MKTFHE and `include("MKTFHEHip.jl")` from MKTFHE.jl after the
# scheme definitions; set LIB to the built mktfhe_amd/lib/libmktfhe_hip.so.  UNEXECUTED in the build image (no Julia
# there): the same C entry points are exercised through ctypes by this repo's test suite.
module MKTFHEHip
using ..MKTFHE
const LIB = "/path/to/mktfhe_amd/lib/libmktfhe_hip.so"

struct MktParams            # mirrors mkt_params (include/mktfhe.h), all Int32
    scheme::Int32; n::Int32; N::Int32; k::Int32; W::Int32
    l_gsw::Int32; logB_gsw::Int32; l_lev::Int32; logB_lev::Int32; l_uni::Int32; logB_uni::Int32
    f::Int32; logD::Int32; blk_len::Int32; blk_d::Int32
end

mutable struct HipScheme    # stands in for MKTFHE.KMS etc. on the evaluator side
    ctx::Ptr{Cvoid}; k::Int; n::Int
end

check(rc, ctx=C_NULL) = rc < 0 ? error(unsafe_string(ccall((:mkt_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))) : rc

# --- flatten the pointer graphs (SURVEY.md 8b) -------------------------------------------------
# TransRGSW (gsw.jl:219-227): rows basketb.stack[1:l], basketa[1].stack[1:l]; each TransRLWE = (b, a[1]);
# each TransNativePoly.coeffs is a contiguous Vector{ComplexF64} of length N/2 -> [n][2l][2][N/2]
function pack_brk(brk::Vector{<:MKTFHE.TransRGSW})
    rows(g) = vcat(g.basketb.stack, (b.stack for b in g.basketa)...)
    reduce(vcat, (reduce(vcat, (vcat(r.b.coeffs, (a.coeffs for a in r.a)...) for r in rows(g))) for g in brk))
end
# ksk::Array{LEV,2} (D-1, N) of references -> [N][D-1][f][n+1] UInt32 with LWE rows [a..., b]
function pack_ksk(ksk::Array{<:MKTFHE.LEV,2})
    D1, N = size(ksk)
    reduce(vcat, (vcat(ksk[d, j].stack[t].a, ksk[d, j].stack[t].b) for j in 1:N for d in 1:D1 for t in 1:ksk[1,1].l))
end
pack_polys(v) = reduce(vcat, (p.coeffs for p in v))          # Vector{TransNativePoly} -> [len][N/2]

function HipScheme(s::MKTFHE.KMS, params::MKTFHE.KMSparams; device = 0)
    g, lv, u = s.btk[1].gswpar, s.btk[1].levpar, s.btk[1].unipar
    p = MktParams(3, s.n, s.N, s.k, 64, g.l, g.logB, lv.l, lv.logB, u.l, u.logB, s.kskpar.l, s.kskpar.logB, 0, 0)
    ctx = Ref{Ptr{Cvoid}}()
    check(ccall((:mkt_ctx_create, LIB), Cint, (Ref{MktParams}, Cint, Cint, Ref{Ptr{Cvoid}}), p, 0, device, ctx))
    c = ctx[]
    f = s.ffter                                             # install the caller's own tables verbatim
    check(ccall((:mkt_set_twiddles, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                c, f.Ψ, f.Ψinv, f.roots, f.rootsinv), c)
    check(ccall((:mkt_load_crs, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Cint), c, pack_polys(s.a), 1), c)      # MKT_FMT_F64_FFT
    for (i, b) in enumerate(s.btk)
        check(ccall((:mkt_load_brk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Cint), c, i - 1, pack_brk(b.brk), 1), c)
        check(ccall((:mkt_load_ksk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt32}), c, i - 1, pack_ksk(b.ksk)), c)
        fpoly = reduce(vcat, (vcat(r.b.coeffs, r.a[1].coeffs) for r in b.rlk.f.stack))
        check(ccall((:mkt_load_rlk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint), c, i - 1, pack_polys(b.rlk.d), fpoly, 1), c)
        check(ccall((:mkt_load_pubkey, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Cint), c, i - 1, pack_polys(b.b), 1), c)
    end
    HipScheme(c, s.k, s.n)
end

# LWE (lwe.jl:1-9) <-> [a..., b]
flat(c::MKTFHE.LWE{UInt32}) = vcat(c.a, c.b)
function unflat!(c::MKTFHE.LWE{UInt32}, v) ; c.a .= @view v[1:end-1]; c.b = v[end]; c end

# bootstrapping!(ctxt, scheme)  (bootstrapping.jl:4) -- batch of one; vectors of LWE batch the same way
function MKTFHE.bootstrapping!(ctxt::MKTFHE.LWE{UInt32}, s::HipScheme)
    v = flat(ctxt)
    check(ccall((:mkt_bootstrap_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Csize_t, Cint), s.ctx, v, 1, 1), s.ctx)   # MKT_MEM_HOST
    unflat!(ctxt, v)
end

# NAND(c1, c2, scheme) (gate.jl:1-8); AND/OR/XOR/XNOR/NOR are op = 1..5
function gate(op, c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipScheme)
    x = reduce(hcat, flat.(c1)); y = reduce(hcat, flat.(c2)); out = similar(x)     # column-major: one ciphertext per column = [B][kn+1] row-major
    check(ccall((:mkt_gate_batch, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Csize_t, Cint),
                s.ctx, op, x, y, out, length(c1), 1), s.ctx)
    [MKTFHE.LWE(out[end, j], out[1:end-1, j]) for j in 1:length(c1)]
end
MKTFHE.NAND(c1::MKTFHE.LWE, c2::MKTFHE.LWE, s::HipScheme) = gate(0, [c1], [c2], s)[1]
end
