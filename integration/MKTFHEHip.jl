# MKTFHEHip.jl -- reference-side binding of the MI355X engine (the shim of INTEGRATION.md as a file).
#
# Put next to src/MKTFHE.jl in a checkout of SNUCP/MKTFHE and `include("MKTFHEHip.jl")` from MKTFHE.jl after the
# scheme definitions; set LIB to the built mktfhe_amd/lib/libmktfhe_hip.so.  UNEXECUTED in the build image (no Julia
# there): the same C entry points are exercised through ctypes by this repo's test suite.
#
# Covers all five scheme types (scheme.jl:107 CGGI, :168 LMSS, :209 CCS, :256 KMS, :301 KMS_block) and the operator
# surface of the path: bootstrapping! (bootstrapping.jl:4), blindrotate! (:32, :114, :234, :369), keyswitch! (:81, :170,
# :333, :564, :664), NAND/AND/OR/XOR/XNOR/NOR (gate.jl:1-53), NOT! (gate.jl:55).
module MKTFHEHip
using ..MKTFHE
const LIB = "/path/to/mktfhe_amd/lib/libmktfhe_hip.so"

struct MktParams            # mirrors mkt_params (include/mktfhe.h), all Int32
    scheme::Int32; n::Int32; N::Int32; k::Int32; W::Int32
    l_gsw::Int32; logB_gsw::Int32; l_lev::Int32; logB_lev::Int32; l_uni::Int32; logB_uni::Int32
    f::Int32; logD::Int32; blk_len::Int32; blk_d::Int32
end

mutable struct HipScheme{R<:Unsigned}    # stands in for MKTFHE.CGGI / LMSS / CCS / KMS / KMS_block on the evaluator side
    ctx::Ptr{Cvoid}
    k::Int          # RLWE length (single-key) or number of parties (multi-key)
    n::Int
    N::Int
    nparty::Int     # 1 for the single-key schemes
end

check(rc, ctx=C_NULL) = rc < 0 ? error(unsafe_string(ccall((:mkt_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))) : rc
const FFT_FORM = Cint(1)    # MKT_FMT_F64_FFT: the reference's own Trans* values
const HOST = Cint(1)        # MKT_MEM_HOST

# --- flatten the pointer graphs (SURVEY.md 8b) -------------------------------------------------
tp(r::MKTFHE.TransRLWE) = vcat(r.b.coeffs, (a.coeffs for a in r.a)...)          # TransRLWE (lwe.jl:165-179) = (b, a[1:k])
# TransRGSW (gsw.jl:219-227): rows basketb.stack[1:l], basketa[1].stack[1:l], ...; -> [n][(k+1)l][k+1][N/2]
pack_rgsw(brk::Vector{<:MKTFHE.TransRGSW}) =
    reduce(vcat, (reduce(vcat, (tp(r) for r in vcat(g.basketb.stack, (b.stack for b in g.basketa)...))) for g in brk))
# TransUniEnc (unienc.jl:92-99): d[1:l], then (f.stack[j].b, f.stack[j].a[1]) j-major -> [n][3l][N/2]
pack_unienc(u::MKTFHE.TransUniEnc) = vcat(reduce(vcat, (p.coeffs for p in u.d)), reduce(vcat, (tp(r) for r in u.f.stack)))
pack_unienc(brk::Vector{<:MKTFHE.TransUniEnc}) = reduce(vcat, (pack_unienc(u) for u in brk))
pack_polys(v) = reduce(vcat, (p.coeffs for p in v))                              # Vector{TransNativePoly} -> [len][N/2]
lwe_row(c::MKTFHE.LWE) = vcat(c.a, c.b)                                          # LWE (lwe.jl:1-9) -> [a..., b]
# ksk::Array{LEV,2} (Drows, N) / Array{LEV,3} (Drows, N, k) of references -> [k][N][Drows][f][n+1] UInt32.
# Block schemes (keygen.jl:37-51, :141-151) leave the entries of the embedded LWE key #undef: rows of zeros.
function pack_ksk(ksk::Array{<:MKTFHE.LEV}, n::Int, f::Int)
    D1, N = size(ksk, 1), size(ksk, 2); K = ndims(ksk) == 3 ? size(ksk, 3) : 1
    out = zeros(UInt32, (n + 1) * f * D1 * N * K)
    o = 0
    for c in 1:K, j in 1:N, d in 1:D1
        ok = ndims(ksk) == 3 ? isassigned(ksk, d, j, c) : isassigned(ksk, d, j)
        lev = ok ? (ndims(ksk) == 3 ? ksk[d, j, c] : ksk[d, j]) : nothing
        for t in 1:f
            if ok; out[o+1:o+n+1] = lwe_row(lev.stack[t]); end
            o += n + 1
        end
    end
    out
end

function install_tables(c, ffter)                          # the caller's own tables verbatim (fft.jl:18-45)
    check(ccall((:mkt_set_twiddles, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                c, ffter.Ψ, ffter.Ψinv, ffter.roots, ffter.rootsinv), c)
    c
end
function create(p::MktParams, ffter, device)
    ctx = Ref{Ptr{Cvoid}}()
    check(ccall((:mkt_ctx_create, LIB), Cint, (Ref{MktParams}, Cint, Cint, Ref{Ptr{Cvoid}}), p, 0, device, ctx))
    install_tables(ctx[], ffter)
end
load_brk(c, i, v) = check(ccall((:mkt_load_brk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Cint), c, i, v, FFT_FORM), c)
load_ksk(c, i, v) = check(ccall((:mkt_load_ksk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt32}), c, i, v), c)
load_pub(c, i, v) = check(ccall((:mkt_load_pubkey, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Cint), c, i, v, FFT_FORM), c)
load_crs(c, v) = check(ccall((:mkt_load_crs, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Cint), c, v, FFT_FORM), c)
function load_rlk(c, i, rlk::MKTFHE.TransUniEnc)
    fpoly = reduce(vcat, (tp(r) for r in rlk.f.stack))
    check(ccall((:mkt_load_rlk, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint), c, i, pack_polys(rlk.d), fpoly, FFT_FORM), c)
end
wbits(::Type{UInt32}) = 32
wbits(::Type{UInt64}) = 64

# parameter block and key upload per scheme type: params_of(s) -> (MktParams, ring word type, parties); upload!(c, s) fills context c
# scheme.jl:107-116 -- single-key, RLWE length k
params_of(s::MKTFHE.CGGI{T}) where T = (MktParams(0, s.n, s.N, s.k, wbits(T), s.gswpar.l, s.gswpar.logB, 0, 0, 0, 0, s.kskpar.l, s.kskpar.logB, 0, 0), T, 1)
# scheme.jl:168-179 -- block-binary keys: n = d * ℓ
params_of(s::MKTFHE.LMSS{T}) where T = (MktParams(1, s.n, s.N, s.k, wbits(T), s.gswpar.l, s.gswpar.logB, 0, 0, 0, 0, s.kskpar.l, s.kskpar.logB, s.ℓ, s.d), T, 1)
function upload!(c, s::Union{MKTFHE.CGGI, MKTFHE.LMSS})
    load_brk(c, 0, pack_rgsw(s.btk.brk)); load_ksk(c, 0, pack_ksk(s.btk.ksk, s.n, s.kskpar.l))
end
# scheme.jl:209-219
params_of(s::MKTFHE.CCS{T}) where T = (MktParams(2, s.n, s.N, s.k, wbits(T), 0, 0, 0, 0, s.unipar.l, s.unipar.logB, s.kskpar.l, s.kskpar.logB, 0, 0), T, s.k)
function upload!(c, s::MKTFHE.CCS)
    load_crs(c, pack_polys(s.a))
    for (i, b) in enumerate(s.btk)
        load_brk(c, i - 1, pack_unienc(b.brk)); load_ksk(c, i - 1, pack_ksk(b.ksk, s.n, s.kskpar.l)); load_pub(c, i - 1, pack_polys(b.b))
    end
end
# scheme.jl:256-265 and :301-312 (T = LWE word, R = ring word)
function params_of(s::Union{MKTFHE.KMS{T, R}, MKTFHE.KMS_block{T, R}}) where {T, R}
    g, lv, u = s.btk[1].gswpar, s.btk[1].levpar, s.btk[1].unipar
    blk = s isa MKTFHE.KMS_block
    (MktParams(blk ? 4 : 3, s.n, s.N, s.k, wbits(R), g.l, g.logB, lv.l, lv.logB, u.l, u.logB, s.kskpar.l, s.kskpar.logB, blk ? s.ℓ : 0, blk ? s.d : 0), R, s.k)
end
function upload!(c, s::Union{MKTFHE.KMS, MKTFHE.KMS_block})
    load_crs(c, pack_polys(s.a))
    for (i, b) in enumerate(s.btk)
        load_brk(c, i - 1, pack_rgsw(b.brk)); load_ksk(c, i - 1, pack_ksk(b.ksk, s.n, s.kskpar.l))
        load_rlk(c, i - 1, b.rlk); load_pub(c, i - 1, pack_polys(b.b))
    end
end
const RefScheme = Union{MKTFHE.CGGI, MKTFHE.LMSS, MKTFHE.CCS, MKTFHE.KMS, MKTFHE.KMS_block}
function HipScheme(s::RefScheme; device = 0)
    p, R, np = params_of(s)
    c = create(p, s.ffter, device)
    upload!(c, s)
    HipScheme{R}(c, s.k, s.n, s.N, np)
end
close!(s::HipScheme) = (ccall((:mkt_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), s.ctx); s.ctx = C_NULL; nothing)
# a second handle over the same resident keys for another Julia thread (own stream / workspace): mkt_ctx_fork
function Base.copy(s::HipScheme{R}) where R
    out = Ref{Ptr{Cvoid}}()
    check(ccall((:mkt_ctx_fork, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), s.ctx, out), s.ctx)
    HipScheme{R}(out[], s.k, s.n, s.N, s.nparty)
end

# LWE (lwe.jl:1-9) <-> [a..., b];  RLWE accumulator (lwe.jl:61-76) <-> [1 + k][N] ring words (b, a_1 .. a_k)
flat(c::MKTFHE.LWE{UInt32}) = lwe_row(c)
function unflat!(c::MKTFHE.LWE{UInt32}, v) ; c.a .= @view v[1:end-1]; c.b = v[end]; c end
flat(acc::MKTFHE.RLWE) = vcat(acc.b.coeffs, (a.coeffs for a in acc.a)...)
function unflat!(acc::MKTFHE.RLWE, v)
    N = acc.N
    acc.b.coeffs .= @view v[1:N]
    for (i, a) in enumerate(acc.a); a.coeffs .= @view v[i*N+1:(i+1)*N]; end
    acc
end

# bootstrapping!(ctxt, scheme)  (bootstrapping.jl:4) -- batch of one; vectors of LWE batch the same way
function MKTFHE.bootstrapping!(ctxt::MKTFHE.LWE{UInt32}, s::HipScheme)
    v = flat(ctxt)
    check(ccall((:mkt_bootstrap_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Csize_t, Cint), s.ctx, v, 1, HOST), s.ctx)
    unflat!(ctxt, v)
end
function MKTFHE.bootstrapping!(ctxts::Vector{MKTFHE.LWE{UInt32}}, s::HipScheme)
    v = reduce(hcat, flat.(ctxts))                           # column-major: one ciphertext per column = [B][kn+1] row-major
    check(ccall((:mkt_bootstrap_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Csize_t, Cint), s.ctx, v, length(ctxts), HOST), s.ctx)
    for (j, c) in enumerate(ctxts); unflat!(c, @view v[:, j]); end
    ctxts
end

# blindrotate!(ã, acc, scheme)  (bootstrapping.jl:32 / :114 / :234 / :369): ã = mod-switched mask (k*n words in [0, 2N]),
# acc updated in place
function MKTFHE.blindrotate!(atilde::Vector{UInt32}, acc::MKTFHE.RLWE{R}, s::HipScheme{R}) where R
    v = flat(acc)
    check(ccall((:mkt_blindrotate_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Ptr{R}, Csize_t, Cint), s.ctx, atilde, v, 1, HOST), s.ctx)
    unflat!(acc, v)
end

# keyswitch!(res, acc, scheme)  (bootstrapping.jl:81 / :170 / :333 / :564 / :664): res overwritten
function MKTFHE.keyswitch!(res::MKTFHE.LWE{UInt32}, acc::MKTFHE.RLWE{R}, s::HipScheme{R}) where R
    out = Vector{UInt32}(undef, s.nparty * s.n + 1)
    check(ccall((:mkt_keyswitch_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{R}, Ptr{UInt32}, Csize_t, Cint), s.ctx, flat(acc), out, 1, HOST), s.ctx)
    unflat!(res, out)
end

# gates (gate.jl:1-53): op = 0 NAND, 1 AND, 2 OR, 3 XOR, 4 XNOR, 5 NOR; vectors evaluate as ONE batch on the GPU
function gate(op, c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipScheme)
    x = reduce(hcat, flat.(c1)); y = reduce(hcat, flat.(c2)); out = similar(x)
    check(ccall((:mkt_gate_batch, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Csize_t, Cint),
                s.ctx, op, x, y, out, length(c1), HOST), s.ctx)
    [MKTFHE.LWE(out[end, j], out[1:end-1, j]) for j in 1:length(c1)]
end
for (op, name) in enumerate((:NAND, :AND, :OR, :XOR, :XNOR, :NOR))
    @eval MKTFHE.$name(c1::MKTFHE.LWE, c2::MKTFHE.LWE, s::HipScheme) = gate($(op - 1), [c1], [c2], s)[1]
    @eval MKTFHE.$name(c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipScheme) = gate($(op - 1), c1, c2, s)
end
# NOT! (gate.jl:55-58) needs no scheme: the reference's own method applies unchanged.

# a different gate per pair, ONE batch (mkt_gate_batch_ops) -- the shape of test/KMS.jl:29-34, which draws a gate per step:
# ops[j] in 0:5 (NAND .. NOR), + 8 / + 16 to negate the first / second input first (NOT!)
function gates(ops::Vector{<:Integer}, c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipScheme)
    x = reduce(hcat, flat.(c1)); y = reduce(hcat, flat.(c2)); out = similar(x)
    check(ccall((:mkt_gate_batch_ops, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Csize_t, Cint),
                s.ctx, UInt8.(ops), x, y, out, length(c1), HOST), s.ctx)
    [MKTFHE.LWE(out[end, j], out[1:end-1, j]) for j in 1:length(c1)]
end
# MUX(sel, a, b) = sel ? a : b -- the reference has no MUX gate; two blind rotations + one key switch (mkt_mux_batch)
function MUX(sel::Vector{<:MKTFHE.LWE}, a::Vector{<:MKTFHE.LWE}, b::Vector{<:MKTFHE.LWE}, s::HipScheme)
    xs = reduce(hcat, flat.(sel)); xa = reduce(hcat, flat.(a)); xb = reduce(hcat, flat.(b)); out = similar(xs)
    check(ccall((:mkt_mux_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Csize_t, Cint),
                s.ctx, xs, xa, xb, out, length(sel), HOST), s.ctx)
    [MKTFHE.LWE(out[end, j], out[1:end-1, j]) for j in 1:length(sel)]
end
MUX(sel::MKTFHE.LWE, a::MKTFHE.LWE, b::MKTFHE.LWE, s::HipScheme) = MUX([sel], [a], [b], s)[1]

# ---- one scheme over all the GPUs of the node, one Julia process (mkt_multi_*: keys uploaded once on devices[1], replicated
#      device to device, the batch cut into contiguous shards, results written into the one output array; no collective) ----
mutable struct HipMultiScheme{R<:Unsigned}
    m::Ptr{Cvoid}
    k::Int; n::Int; N::Int; nparty::Int
end
mcheck(rc, m=C_NULL) = rc < 0 ? error(unsafe_string(ccall((:mkt_multi_last_error, LIB), Cstring, (Ptr{Cvoid},), m))) : rc
function HipMultiScheme(s::RefScheme; devices::Vector{<:Integer} = [0])
    p, R, np = params_of(s)
    h = Ref{Ptr{Cvoid}}()
    mcheck(ccall((:mkt_multi_create, LIB), Cint, (Ref{MktParams}, Cint, Ptr{Cint}, Cint, Cint, Ref{Ptr{Cvoid}}), p, 0, Cint.(devices), length(devices), 0, h))
    c0 = ccall((:mkt_multi_ctx, LIB), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), h[], 0)    # the first device's context: tables and keys go there ...
    install_tables(c0, s.ffter); upload!(c0, s)
    mcheck(ccall((:mkt_multi_replicate, LIB), Cint, (Ptr{Cvoid},), h[]), h[])      # ... and are copied to the other devices (hipMemcpyPeer)
    HipMultiScheme{R}(h[], s.k, s.n, s.N, np)
end
close!(s::HipMultiScheme) = (ccall((:mkt_multi_destroy, LIB), Cint, (Ptr{Cvoid},), s.m); s.m = C_NULL; nothing)
function gate(op, c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipMultiScheme)
    x = reduce(hcat, flat.(c1)); y = reduce(hcat, flat.(c2)); out = similar(x)
    mcheck(ccall((:mkt_multi_gate_batch, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{UInt32}, Ptr{UInt32}, Csize_t, Cint),
                 s.m, op, x, y, out, length(c1), HOST), s.m)
    [MKTFHE.LWE(out[end, j], out[1:end-1, j]) for j in 1:length(c1)]
end
function MKTFHE.bootstrapping!(ctxts::Vector{MKTFHE.LWE{UInt32}}, s::HipMultiScheme)
    v = reduce(hcat, flat.(ctxts))
    mcheck(ccall((:mkt_multi_bootstrap_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Csize_t, Cint), s.m, v, length(ctxts), HOST), s.m)
    for (j, c) in enumerate(ctxts); unflat!(c, @view v[:, j]); end
    ctxts
end
for (op, name) in enumerate((:NAND, :AND, :OR, :XOR, :XNOR, :NOR))
    @eval MKTFHE.$name(c1::Vector{<:MKTFHE.LWE}, c2::Vector{<:MKTFHE.LWE}, s::HipMultiScheme) = gate($(op - 1), c1, c2, s)
end
end
