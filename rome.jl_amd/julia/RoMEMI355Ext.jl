# RoMEMI355Ext.jl -- thin `ccall` shim binding librome_mi355.so behind RoME.jl's factor plugin surface.
#
# WRITTEN BLIND: this container has no Julia toolchain (and IncrementalInference/Manifolds/Optim are not
# vendored with the reference), so this file has never been executed.  It mirrors include/rome_mi355.h 1:1
# and is the binding a RoME maintainer would drop into `ext/` (weak-dependency extension, like
# ext/RoMEFluxExt.jl).  Nothing below redefines a factor: the RoME structs, `CalcFactor` functors,
# `getSample` and `getManifold` stay as they are (src/factors/Pose2D.jl:30-67, PriorPose2.jl:13-47,
# BearingRange2D.jl:10-64, Pose3Pose3.jl:9-29); only the batch loop IIF runs around them is replaced.
module RoMEMI355Ext

using RoME
using IncrementalInference
import IncrementalInference: approxConvBelief
using StaticArrays, RecursiveArrayTools, Distributions, LinearAlgebra

const LIB = get(ENV, "ROME_MI355_LIB", "librome_mi355.so")

# ---- rome_opts (include/rome_mi355.h) -------------------------------------------------------------
struct RomeOpts
  n_particles::Int32
  solver::Int32          # 0 closed form, 1 Newton (default), 2 Nelder-Mead (Optim defaults)
  max_iters::Int32
  inflate_cycles::Int32
  tol::Float64
  inflation::Float64
  seed::UInt64
  stream_offset::UInt64
  layout::Int32          # 0 SoA [block][dim][N], 1 AoS [block][N][dim], 2 AoS of the reference's point containers
  presampled::Int32      # 0: `noise` rows are standard normals; 1: they are the measurement samples themselves (any SamplableBelief)
  spread_nh::Float64     # IIF spreadNH
  nullhypo::Float64      # IIF nullhypo= of the factor(s) in the call
end

function default_opts(fg::AbstractDFG; solver::Integer=1, seed::Integer=rand(UInt64), stream_offset::Integer=0)
  o = Ref{RomeOpts}()
  ccall((:rome_opts_default, LIB), Cvoid, (Ref{RomeOpts}, Int32), o, solver)
  p = getSolverParams(fg)
  d = o[]
  RomeOpts(p.N, d.solver, d.max_iters, p.inflateCycles, d.tol, p.inflation, seed, stream_offset, 1 #=AoS=#, 0, p.spreadNH, 0.0)
end

# ---- context ---------------------------------------------------------------------------------------
mutable struct RomeCtx
  h::Ptr{Cvoid}
  function RomeCtx(device::Integer=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:rome_ctx_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device)
    rc == 0 || error("rome_ctx_create: " * unsafe_string(ccall((:rome_strerror, LIB), Cstring, (Cint,), rc)))
    c = new(r[])
    finalizer(c -> ccall((:rome_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.h), c)
  end
end
const _ctx = Ref{Union{Nothing,RomeCtx}}(nothing)   # one context per Julia thread/clique task in a real deployment
ctx() = (_ctx[] === nothing && (_ctx[] = RomeCtx()); _ctx[]::RomeCtx)

check(rc) = rc == 0 ? nothing : error("librome_mi355: " * unsafe_string(ccall((:rome_strerror, LIB), Cstring, (Cint,), rc)))

# ---- layout helpers: Vector of manifold points <-> N x dim coordinate matrix (AoS rows) -----------------
coords(::Type{Pose2}, pts) = reduce(hcat, [SA[p.x[1][1], p.x[1][2], atan(p.x[2][2,1], p.x[2][1,1])] for p in pts])  # 3 x N (column = particle = AoS row)
coords(::Type{Point2}, pts) = reduce(hcat, pts)
points(::Type{Pose2}, C) = [getPoint(Pose2, C[:, i]) for i in axes(C, 2)]
points(::Type{Point2}, C) = [SVector{2}(C[:, i]) for i in axes(C, 2)]

# ---- batch convolution: replaces N x (getSample + _solveCCWNumeric!) inside approxConvBelief ------------
function conv_pose2pose2(fg, f::Pose2Pose2, fixedpts, u0pts, dir::Integer; solver=1)
  o = default_opts(fg; solver)
  N = Int(o.n_particles)
  μ = collect(Float64, mean(f.Z)); Σ = collect(Float64, cov(f.Z))'     # row-major
  fixed = coords(Pose2, fixedpts); target = coords(Pose2, u0pts)          # 3 x N column-major == N x 3 row-major
  d = Int32[dir]
  GC.@preserve μ Σ fixed target d begin
    check(ccall((:rome_conv_pose2pose2, LIB), Cint,
      (Ptr{Cvoid}, Ref{RomeOpts}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
      ctx().h, o, 1, d, μ, Σ, fixed, C_NULL, target, C_NULL))
  end
  points(Pose2, target)
end

# Zero-conversion variant: Vector{ArrayPartition{Float64,Tuple{SVector{2},SMatrix{2,2}}}} is isbits, i.e. 6 contiguous
# doubles [tx,ty,R11,R21,R12,R22] per point == ROME_LAYOUT_AOS_POINTS (layout = 2): pass the belief vectors as they are.
function conv_pose2pose2!(fg, f::Pose2Pose2, fixedpts::Vector{P}, u0pts::Vector{P}, dir::Integer; solver=1) where {P}
  d0 = default_opts(fg; solver)
  o = RomeOpts(d0.n_particles, d0.solver, d0.max_iters, d0.inflate_cycles, d0.tol, d0.inflation, d0.seed, d0.stream_offset, 2, 0, d0.spread_nh, d0.nullhypo)
  μ = collect(Float64, mean(f.Z)); Σ = collect(Float64, cov(f.Z))'
  d = Int32[dir]
  GC.@preserve μ Σ fixedpts u0pts d begin
    check(ccall((:rome_conv_pose2pose2, LIB), Cint,
      (Ptr{Cvoid}, Ref{RomeOpts}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
      ctx().h, o, 1, d, μ, Σ, Ptr{Float64}(pointer(fixedpts)), C_NULL, Ptr{Float64}(pointer(u0pts)), C_NULL))
  end
  u0pts   # overwritten in place with the N solution points
end

function conv_bearingrange(fg, f::Pose2Point2BearingRange{<:Normal,<:Normal}, fixedpts, u0pts, dir::Integer; solver=1)
  o = default_opts(fg; solver)
  μ = Float64[mean(f.bearing), mean(f.range)]; σ = Float64[std(f.bearing), std(f.range)]
  Tf, Tt = dir == 0 ? (Pose2, Point2) : (Point2, Pose2)
  fixed = coords(Tf, fixedpts); target = coords(Tt, u0pts)
  GC.@preserve μ σ fixed target begin
    check(ccall((:rome_conv_pose2point2br, LIB), Cint,
      (Ptr{Cvoid}, Ref{RomeOpts}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
      ctx().h, o, 1, dir, μ, σ, fixed, C_NULL, target, C_NULL))
  end
  points(Tt, target)
end

# ---- the drop-in: specialise IIF's entry for the accelerated factor types -------------------------------
# IIF 0.35: approxConvBelief(dfg, fc::DFGFactor, target::Symbol, measurement; solveKey, N, ...) -> ManifoldKernelDensity.
# Everything not matched here falls through to IIF's per-sample path (the RoME functors are untouched).
function approxConvBelief(dfg::AbstractDFG, fc::DFGFactor{<:CommonConvWrapper{<:Pose2Pose2}}, target::Symbol,
                          measurement::AbstractVector=Tuple[]; solveKey::Symbol=:default, kw...)
  vars = getVariableOrder(fc)
  dir = vars[2] == target ? 0 : 1
  other = dir == 0 ? vars[1] : vars[2]
  pts = conv_pose2pose2(dfg, getFactorType(fc), getVal(dfg, other; solveKey), getVal(dfg, target; solveKey), dir)
  return manikde!(getManifold(Pose2), pts)     # KDE wrap stays in AMP (SURVEY 8(a) row a11: not part of the unit)
end

function approxConvBelief(dfg::AbstractDFG, fc::DFGFactor{<:CommonConvWrapper{<:Pose2Point2BearingRange{<:Normal,<:Normal}}},
                          target::Symbol, measurement::AbstractVector=Tuple[]; solveKey::Symbol=:default, kw...)
  vars = getVariableOrder(fc)
  dir = vars[2] == target ? 0 : 1
  other = dir == 0 ? vars[1] : vars[2]
  pts = conv_bearingrange(dfg, getFactorType(fc), getVal(dfg, other; solveKey), getVal(dfg, target; solveKey), dir)
  return manikde!(getManifold(getVariableType(dfg, target)), pts)
end

# ---- Pose3Pose3: coordinates (t, ω) = get_coordinates(M, ϵ, log(M, ϵ, p), DefaultOrthogonalBasis()) ----------
const M3 = getManifold(Pose3)
coords(::Type{Pose3}, pts) = reduce(hcat, [get_coordinates(M3, getPointIdentity(M3), log(M3, getPointIdentity(M3), p), DefaultOrthogonalBasis()) for p in pts])
points(::Type{Pose3}, C) = [getPoint(Pose3, C[:, i]) for i in axes(C, 2)]

function conv_pose3pose3(fg, f::Pose3Pose3, fixedpts, u0pts, dir::Integer; solver=1)
  o = default_opts(fg; solver)
  μ = collect(Float64, mean(f.Z)); Σ = collect(Float64, cov(f.Z))'
  fixed = coords(Pose3, fixedpts); target = coords(Pose3, u0pts)
  d = Int32[dir]
  GC.@preserve μ Σ fixed target d begin
    check(ccall((:rome_conv_pose3pose3, LIB), Cint,
      (Ptr{Cvoid}, Ref{RomeOpts}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
      ctx().h, o, 1, d, μ, Σ, fixed, C_NULL, target, C_NULL))
  end
  points(Pose3, target)
end

function approxConvBelief(dfg::AbstractDFG, fc::DFGFactor{<:CommonConvWrapper{<:Pose3Pose3}}, target::Symbol,
                          measurement::AbstractVector=Tuple[]; solveKey::Symbol=:default, kw...)
  vars = getVariableOrder(fc)
  dir = vars[2] == target ? 0 : 1
  other = dir == 0 ? vars[1] : vars[2]
  pts = conv_pose3pose3(dfg, getFactorType(fc), getVal(dfg, other; solveKey), getVal(dfg, target; solveKey), dir)
  return manikde!(M3, pts)
end

# ---- priors: N samples of the prior as points (IIF samplePoint) ------------------------------------------------
function sample_prior(fg, f::Union{PriorPose2,PriorPose3})
  T, sym, d = f isa PriorPose2 ? (Pose2, :rome_sample_priorpose2, 3) : (Pose3, :rome_sample_priorpose3, 6)
  o = default_opts(fg)
  μ = collect(Float64, mean(f.Z)); Σ = collect(Float64, cov(f.Z))'
  out = Matrix{Float64}(undef, d, Int(o.n_particles))
  GC.@preserve μ Σ out begin
    check(ccall((sym, LIB), Cint, (Ptr{Cvoid}, Ref{RomeOpts}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx().h, o, 1, μ, Σ, C_NULL, out))
  end
  points(T, out)
end

# ---- clique-level batch: every proposal of a destination variable in ONE library call (rome_clique_proposals) -----------
# IIF 0.35: proposalbeliefs!(dfg, destlbl, factors, dens, measurement; solveKey, N, ...) pushes one approxConvBelief per factor onto
# `dens`.  The per-factor specialisations above cost one H2D + launch + D2H (~45 µs) per convolution; this method gathers the
# beliefs of every variable the factors touch ONCE, builds the row tables (factor, dir, fixed_var, target_var) and gets all
# proposals back from a single call (1.9e6 convolutions/s PCIe-inclusive at 1000 rows, measured through the Python mirror
# rome_jl_amd.proposalbeliefs, tests/test_gpu_clique.py).  Factors of other types fall through to IIF one by one.
struct RomeCliqueHost            # include/rome_mi355.h: rome_clique_host
  n_pose2::Int32; n_point2::Int32; n_pose3::Int32; reserved0::Int32
  bel_pose2::Ptr{Float64}; bel_point2::Ptr{Float64}; bel_pose3::Ptr{Float64}
  n_p2p2::Int32; f_p2p2::Int32; p2p2_rows4::Ptr{Int32}; p2p2_mu::Ptr{Float64}; p2p2_cov::Ptr{Float64}; out_p2p2::Ptr{Float64}
  n_br1::Int32; n_br0::Int32; f_br::Int32; reserved1::Int32
  br1_rows4::Ptr{Int32}; br0_rows4::Ptr{Int32}; br_mu::Ptr{Float64}; br_sigma::Ptr{Float64}; out_br1::Ptr{Float64}; out_br0::Ptr{Float64}
  n_p3p3::Int32; f_p3p3::Int32; p3p3_rows4::Ptr{Int32}; p3p3_mu::Ptr{Float64}; p3p3_cov::Ptr{Float64}; out_p3p3::Ptr{Float64}
  n_prpt2::Int32; f_prpt2::Int32; prpt2_rows4::Ptr{Int32}; prpt2_mu::Ptr{Float64}; prpt2_cov::Ptr{Float64}; out_prpt2::Ptr{Float64}   # PriorPoint2 rows
  # per-row hypothesis columns (C_NULL = none): multihypo=[1, w, 1-w] -> <fam>_alt (index of the other candidate, -1 = ordinary row) +
  # <fam>_hypo_w; nullhypo=p -> <fam>_nullhypo
  p2p2_alt::Ptr{Int32}; p2p2_hypo_w::Ptr{Float64}; p2p2_nullhypo::Ptr{Float64}
  br1_alt::Ptr{Int32}; br1_hypo_w::Ptr{Float64}; br1_nullhypo::Ptr{Float64}
  br0_alt::Ptr{Int32}; br0_hypo_w::Ptr{Float64}; br0_nullhypo::Ptr{Float64}
  p3p3_nullhypo::Ptr{Float64}
  # per-row Philox stream ids (C_NULL = the row's index): see the header
  p2p2_stream::Ptr{Int32}; br1_stream::Ptr{Int32}; br0_stream::Ptr{Int32}; p3p3_stream::Ptr{Int32}; prpt2_stream::Ptr{Int32}
  # rome_upsolve_plan only: per-row store block holding the row's N MEASUREMENT samples (-1 = ordinary row): the relative up-message of a
  # child clique as a sampled-measurement factor of its parent (C_NULL = none)
  p2p2_meas::Ptr{Int32}; br1_meas::Ptr{Int32}; br0_meas::Ptr{Int32}
end

const _accelerated = Union{Pose2Pose2, PriorPose2, Pose2Point2BearingRange{<:Normal,<:Normal}, Pose3Pose3, PriorPose3, PriorPoint2}
# IIF keeps `multihypo=[1, w, 1-w]` / `nullhypo=p` in the factor's solver data; the batch carries them as per-row columns for a
# bearing-range or Pose2Pose2 factor over [first, cand1, cand2] (every use in the reference: test/testMultimodalRangeBearing.jl:53)
_mh(fc) = (m = getSolverData(fc).multihypo; (m === nothing || isempty(m)) ? nothing : m)
_nh(fc) = (n = getSolverData(fc).nullhypo; n === nothing ? 0.0 : Float64(n))
_batchable(fc) = (f = getFactorType(fc); f isa _accelerated &&
                  (_mh(fc) === nothing || (f isa Union{Pose2Pose2,Pose2Point2BearingRange} && length(_mh(fc)) == 3 && _mh(fc)[1] == 1.0)))

# row tables of a list of (factor, destination) pairs over the variables they touch: what rome_clique_host carries
function _clique_tables(dfg::AbstractDFG, pairs::AbstractVector; solveKey::Symbol=:default, extra_vars::AbstractVector{Symbol}=Symbol[],
                        index::Union{Nothing,Dict{Symbol,Int32}}=nothing)
  # index === nothing: variables are numbered per type as they are met and their beliefs gathered (the one-shot entries);
  # index = a RomeStore's numbering: the tables address the store's blocks and no belief leaves the graph
  vidx = Dict{Symbol,Int32}(); vars = Dict(Pose2 => Symbol[], Point2 => Symbol[], Pose3 => Symbol[])
  function var!(l)
    index === nothing || return (vidx[l] = index[l])
    get!(vidx, l) do
      T = typeof(getVariableType(dfg, l)); push!(vars[T], l); Int32(length(vars[T]) - 1)
    end
  end
  fams = (:p2p2, :br1, :br0, :p3p3, :prpt2)
  rows = Dict(f => Int32[] for f in fams)
  alt = Dict(f => Int32[] for f in fams); hw = Dict(f => Float64[] for f in fams); nh = Dict(f => Float64[] for f in fams)
  tabμ = Dict(f => Float64[] for f in (:p2p2, :br, :p3p3, :prpt2)); tabΣ = Dict(f => Float64[] for f in (:p2p2, :br, :p3p3, :prpt2))
  nfac = Dict(:p2p2 => 0, :br => 0, :p3p3 => 0, :prpt2 => 0); order = Tuple{Symbol,Int}[]
  for (fc, destlbl) in pairs
    f = getFactorType(fc); vo = getVariableOrder(fc); m = _mh(fc)
    a, w = Int32(-1), 1.0
    if f isa Union{PriorPose2,PriorPose3,PriorPoint2}
      fam = f isa PriorPose2 ? :p2p2 : f isa PriorPose3 ? :p3p3 : :prpt2; tab = fam
      append!(tabμ[tab], mean(f.Z)); append!(tabΣ[tab], vec(collect(cov(f.Z))'))
      append!(rows[fam], Int32[nfac[tab], 2, var!(destlbl), var!(destlbl)])
    else
      if m === nothing
        dir = vo[2] == destlbl ? 0 : 1; other = dir == 0 ? vo[1] : vo[2]
      elseif destlbl == vo[1]                   # solve the first variable: the fixed candidate is drawn per particle
        dir, other, a, w = 1, vo[2], var!(vo[3]), m[2]
      else                                      # solve one of the candidates: particles drawn for the other one only get entropy
        own, oth, w = destlbl == vo[2] ? (vo[2], vo[3], m[2]) : (vo[3], vo[2], m[3])
        dir, other, a = 0, vo[1], var!(oth)
      end
      if f isa Union{Pose2Pose2,Pose3Pose3}
        fam = f isa Pose2Pose2 ? :p2p2 : :p3p3; tab = fam
        append!(tabμ[tab], mean(f.Z)); append!(tabΣ[tab], vec(collect(cov(f.Z))'))
      else
        fam = dir == 0 ? :br0 : :br1; tab = :br
        append!(tabμ[tab], Float64[mean(f.bearing), mean(f.range)]); append!(tabΣ[tab], Float64[std(f.bearing), std(f.range)])
      end
      append!(rows[fam], Int32[nfac[tab], dir, var!(other), var!(destlbl)])
    end
    nfac[tab] += 1
    push!(alt[fam], a); push!(hw[fam], w); push!(nh[fam], f isa Union{PriorPose2,PriorPose3,PriorPoint2} ? 0.0 : _nh(fc))
    push!(order, (fam, length(rows[fam]) ÷ 4))
  end
  foreach(var!, extra_vars)
  # beliefs: the reference's point containers as they are for Pose2 (6 doubles) / Pose3 (12 doubles) = ROME_LAYOUT_AOS_POINTS
  blk(T) = isempty(vars[T]) ? Float64[] : reduce(vcat, [reinterpret(Float64, getVal(dfg, l; solveKey)) for l in vars[T]])
  # hypothesis columns only where a row carries one (C_NULL otherwise: the plain kernels)
  for fam in fams
    any(>=(0), alt[fam]) || (empty!(alt[fam]); empty!(hw[fam]))
    any(>(0.0), nh[fam]) || empty!(nh[fam])
  end
  (; vidx, vars, rows, alt, hw, nh, tabμ, tabΣ, nfac, order, b2 = blk(Pose2), bl = blk(Point2), b3 = blk(Pose3))
end
_p(x) = isempty(x) ? Ptr{eltype(x)}(C_NULL) : pointer(x)
_clique_host(t, o2, o1, o0, o3, opt = Float64[], nv = (length(t.vars[Pose2]), length(t.vars[Point2]), length(t.vars[Pose3]))) =
  RomeCliqueHost(nv[1], nv[2], nv[3], 0, _p(t.b2), _p(t.bl), _p(t.b3),
                 length(t.rows[:p2p2]) ÷ 4, t.nfac[:p2p2], _p(t.rows[:p2p2]), _p(t.tabμ[:p2p2]), _p(t.tabΣ[:p2p2]), _p(o2),
                 length(t.rows[:br1]) ÷ 4, length(t.rows[:br0]) ÷ 4, t.nfac[:br], 0, _p(t.rows[:br1]), _p(t.rows[:br0]), _p(t.tabμ[:br]), _p(t.tabΣ[:br]), _p(o1), _p(o0),
                 length(t.rows[:p3p3]) ÷ 4, t.nfac[:p3p3], _p(t.rows[:p3p3]), _p(t.tabμ[:p3p3]), _p(t.tabΣ[:p3p3]), _p(o3),
                 length(t.rows[:prpt2]) ÷ 4, t.nfac[:prpt2], _p(t.rows[:prpt2]), _p(t.tabμ[:prpt2]), _p(t.tabΣ[:prpt2]), _p(opt),
                 _p(t.alt[:p2p2]), _p(t.hw[:p2p2]), _p(t.nh[:p2p2]), _p(t.alt[:br1]), _p(t.hw[:br1]), _p(t.nh[:br1]),
                 _p(t.alt[:br0]), _p(t.hw[:br0]), _p(t.nh[:br0]), _p(t.nh[:p3p3]),
                 Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL),
                 Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL))
_points_opts(dfg, N) = (d0 = default_opts(dfg);
  RomeOpts(Int32(N), d0.solver, d0.max_iters, d0.inflate_cycles, d0.tol, d0.inflation, d0.seed, d0.stream_offset, 2 #=points=#, 0, d0.spread_nh, 0.0))

# The clique-level batch as a function of its own.  It is NOT installed over IIF's `proposalbeliefs!` by loading this file (that would
# be method piracy on generic argument types): call `enable_clique_batch!()` to opt in.  Returns what IIF's method returns -- the
# inferred dimension per factor (`ipc`): full dimension (1.0) for every accelerated factor, then the fallback's values.
function proposalbeliefs_mi355!(dfg::AbstractDFG, destlbl::Symbol, factors::AbstractVector{<:DFGFactor},
                                dens::AbstractVector, measurement::AbstractVector=Tuple[];
                                solveKey::Symbol=:default, N::Integer=getSolverParams(dfg).N, kw...)
  fast = [fc for fc in factors if _batchable(fc)]     # multihypo / nullhypo factors travel as per-row columns of the batch
  slow = [fc for fc in factors if !(fc in fast)]
  t = _clique_tables(dfg, [(fc, destlbl) for fc in fast]; solveKey)
  out(fam, w) = Vector{Float64}(undef, (length(t.rows[fam]) ÷ 4) * N * w)
  o2, o1, o0, o3, opt = out(:p2p2, 6), out(:br1, 6), out(:br0, 2), out(:p3p3, 12), out(:prpt2, 2)
  o = _points_opts(dfg, N)
  GC.@preserve t o2 o1 o0 o3 opt begin
    q = _clique_host(t, o2, o1, o0, o3, opt)
    check(ccall((:rome_clique_proposals, LIB), Cint, (Ptr{Cvoid}, Ref{RomeOpts}, Ref{RomeCliqueHost}), ctx().h, o, q))
  end
  T = typeof(getVariableType(dfg, destlbl)); M = getManifold(T)
  P = eltype(getVal(dfg, destlbl; solveKey))
  for (fam, r) in t.order
    buf, w = fam == :p2p2 ? (o2, 6) : fam == :br1 ? (o1, 6) : fam == :br0 ? (o0, 2) : fam == :prpt2 ? (opt, 2) : (o3, 12)
    pts = collect(reinterpret(P, view(buf, (r - 1) * N * w + 1 : r * N * w)))
    push!(dens, manikde!(M, pts))             # the KDE wrap (bandwidth selection) stays in AMP; rome_kde_bandwidth can supply it
  end
  ipc = ones(Float64, length(fast))
  if !isempty(slow)
    rest = invoke(IncrementalInference.proposalbeliefs!, Tuple{AbstractDFG,Symbol,AbstractVector,AbstractVector,AbstractVector},
                  dfg, destlbl, slow, dens, measurement; solveKey, N, kw...)
    rest isa AbstractVector && append!(ipc, rest)
  end
  return ipc
end

const _clique_batch_enabled = Ref(false)
function enable_clique_batch!()
  _clique_batch_enabled[] && return nothing
  @eval IncrementalInference.proposalbeliefs!(dfg::AbstractDFG, destlbl::Symbol, factors::AbstractVector{<:DFGFactor},
                                              dens::AbstractVector, measurement::AbstractVector=Tuple[]; kw...) =
    proposalbeliefs_mi355!(dfg, destlbl, factors, dens, measurement; kw...)
  _clique_batch_enabled[] = true
  nothing
end

# ---- clique up-solve: IIF upGibbsCliqueDensity in ONE library call (rome_clique_upsolve) ---------------------------------------
# gibbsIters x for each frontal {proposalbeliefs! -> manikde! -> manifoldProduct -> setValKDE!} stays on the device; the beliefs of the
# clique's variables cross PCIe once, the new frontal beliefs and their manikde! bandwidths come back.  Python twin (what the tests
# drive): rome_jl_amd.upGibbsCliqueDensity, tests/test_gpu_upsolve.py.
struct RomeCliqueUpsolveHost     # include/rome_mi355.h: rome_clique_upsolve_host
  clique::RomeCliqueHost
  gibbs_iters::Int32; product_iters::Int32; schedule::Int32; n_up::Int32
  up_type::Ptr{Int32}; up_var::Ptr{Int32}
  n_msg_pose2::Int32; n_msg_point2::Int32; n_msg_pose3::Int32; reserved0::Int32
  msg_pose2::Ptr{Float64}; msg_pose2_up::Ptr{Int32}; msg_point2::Ptr{Float64}; msg_point2_up::Ptr{Int32}
  msg_pose3::Ptr{Float64}; msg_pose3_up::Ptr{Int32}
  new_pose2::Ptr{Float64}; bw_pose2::Ptr{Float64}; new_point2::Ptr{Float64}; bw_point2::Ptr{Float64}
  new_pose3::Ptr{Float64}; bw_pose3::Ptr{Float64}
  up_group::Ptr{Int32}          # optional update groups (a frontier of independent cliques in one call); C_NULL: follow `schedule`
  up_stream::Ptr{Int32}         # optional product stream ids (partition-independent frontiers); C_NULL: position within the type
  up_mirror::Ptr{Int32}         # rome_upsolve_plan only: blocks of a device send buffer; C_NULL here
  # rome_upsolve_plan only: messages that live in the store (a child clique's separator beliefs, written one tree level earlier)
  n_smsg_pose2::Int32; n_smsg_point2::Int32; n_smsg_pose3::Int32; reserved1::Int32
  smsg_pose2_src::Ptr{Int32}; smsg_pose2_up::Ptr{Int32}; smsg_point2_src::Ptr{Int32}; smsg_point2_up::Ptr{Int32}
  smsg_pose3_src::Ptr{Int32}; smsg_pose3_up::Ptr{Int32}
end


function upsolve_clique!(dfg::AbstractDFG, frontals::AbstractVector{Symbol}, factors::AbstractVector{<:DFGFactor};
                         gibbsIters::Integer=3, Niter::Integer=1, sequential::Bool=true,
                         solveKey::Symbol=:default, N::Integer=getSolverParams(dfg).N)
  # Every factor of a frontal must be batchable: dropping one silently would write back a clique posterior that omits it.  A clique
  # with any other factor (a user factor, a partial, a multihypo over more than two candidates) goes to IIF's own upGibbsCliqueDensity.
  touching = [fc for fc in factors if any(in(getVariableOrder(fc)), frontals)]
  bad = [getLabel(fc) for fc in touching if !_batchable(fc)]
  isempty(bad) || throw(ArgumentError("upsolve_clique!: factors $(bad) are outside the accelerated set; use IIF's upGibbsCliqueDensity for this clique"))
  # pairs grouped by frontal in Gibbs order (the library checks the grouping)
  pairs = [(fc, l) for l in frontals for fc in touching if l in getVariableOrder(fc)]
  t = _clique_tables(dfg, pairs; solveKey, extra_vars = collect(frontals))
  tcode(l) = (T = typeof(getVariableType(dfg, l)); T === Pose2 ? Int32(0) : T === Point2 ? Int32(1) : Int32(2))
  upt = Int32[tcode(l) for l in frontals]; upv = Int32[t.vidx[l] for l in frontals]
  cnt(c) = count(==(Int32(c)), upt)
  n2, nl, n3 = cnt(0), cnt(1), cnt(2)
  new2, bw2 = Vector{Float64}(undef, n2 * N * 6), Vector{Float64}(undef, n2 * 3)
  newl, bwl = Vector{Float64}(undef, nl * N * 2), Vector{Float64}(undef, nl * 2)
  new3, bw3 = Vector{Float64}(undef, n3 * N * 12), Vector{Float64}(undef, n3 * 6)
  o = _points_opts(dfg, N)
  GC.@preserve t upt upv new2 bw2 newl bwl new3 bw3 begin
    u = RomeCliqueUpsolveHost(_clique_host(t, Float64[], Float64[], Float64[], Float64[]),
                              Int32(gibbsIters), Int32(Niter), Int32(sequential ? 0 : 1), Int32(length(frontals)), pointer(upt), pointer(upv),
                              0, 0, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                              _p(new2), _p(bw2), _p(newl), _p(bwl), _p(new3), _p(bw3), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL),
                              Int32(0), Int32(0), Int32(0), Int32(0), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL))
    check(ccall((:rome_clique_upsolve, LIB), Cint, (Ptr{Cvoid}, Ref{RomeOpts}, Ref{RomeCliqueUpsolveHost}), ctx().h, o, u))
  end
  k = Dict(0 => 0, 1 => 0, 2 => 0)
  for (l, tc) in zip(frontals, upt)
    buf, bwb, w, d = tc == 0 ? (new2, bw2, 6, 3) : tc == 1 ? (newl, bwl, 2, 2) : (new3, bw3, 12, 6)
    r = k[Int(tc)]; k[Int(tc)] += 1
    P = eltype(getVal(dfg, l; solveKey)); M = getManifold(typeof(getVariableType(dfg, l)))
    pts = collect(reinterpret(P, view(buf, r * N * w + 1 : (r + 1) * N * w)))
    setValKDE!(dfg, l, manikde!(M, pts; bw = bwb[r * d + 1 : (r + 1) * d]), false, 1.0; solveKey)   # the bandwidth the device selected
  end
  nothing
end

# ---- device-resident beliefs across clique up-solves: rome_store + rome_upsolve_plan (+ rome_scatter_plan) -----------------------------
# A tree solve visits thousands of cliques, and the separator beliefs one clique writes are what its parent reads: with a store the
# beliefs cross PCIe at upload! and download! only; a plan is validated and uploaded ONCE per clique (or per frontier of independent
# cliques) and run! is kernel launches on the context's stream -- no copy, no synchronisation.  Python twins (what the tests drive):
# rome_jl_amd.clique.DeviceStore / UpsolvePlan / ScatterPlan, tests/test_gpu_clique_hypo.py, tests/test_gpu_upsolve.py.
_tcode(T) = T === Pose2 ? Int32(0) : T === Point2 ? Int32(1) : T === Pose3 ? Int32(2) : error("RoMEMI355Ext: variable type $T is not in the store")
_ptlen(c) = (6, 2, 12)[c + 1]      # doubles per native point (ROME_LAYOUT_AOS_POINTS)
const _VT = (Pose2, Point2, Pose3)

mutable struct RomeStore
  h::Ptr{Cvoid}
  N::Int
  labels::NTuple{3,Vector{Symbol}}          # per type code, in block order
  index::Dict{Symbol,Int32}                 # label -> block within its type
end

function RomeStore(dfg::AbstractDFG, labels::AbstractVector{Symbol}=ls(dfg); solveKey::Symbol=:default, N::Integer=getSolverParams(dfg).N)
  lab = (Symbol[], Symbol[], Symbol[]); idx = Dict{Symbol,Int32}()
  for l in labels
    c = _tcode(typeof(getVariableType(dfg, l))); idx[l] = Int32(length(lab[c + 1])); push!(lab[c + 1], l)
  end
  r = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:rome_store_create, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
              ctx().h, N, length(lab[1]), length(lab[2]), length(lab[3]), r))
  st = RomeStore(r[], Int(N), lab, idx)
  finalizer(s -> ccall((:rome_store_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.h), st)   # (destroy the plans of a store before the store)
  upload!(st, dfg; solveKey)
  st
end

# beliefs graph -> store: the reference's point containers as they are, one copy per run of consecutive blocks
function upload!(st::RomeStore, dfg::AbstractDFG, labels=nothing; solveKey::Symbol=:default)
  for c in 0:2
    want = [k for (k, l) in enumerate(st.labels[c + 1]) if (labels === nothing || l in labels) && isInitialized(dfg, l)]
    k = 1
    while k <= length(want)
      j = k
      while j < length(want) && want[j + 1] == want[j] + 1; j += 1; end
      buf = reduce(vcat, [collect(reinterpret(Float64, getVal(dfg, st.labels[c + 1][i]; solveKey))) for i in want[k:j]])
      length(buf) == (j - k + 1) * st.N * _ptlen(c) || error("RomeStore: every uploaded belief must hold N = $(st.N) points")
      GC.@preserve buf check(ccall((:rome_store_upload, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Float64}),
                                   st.h, 2, c, want[k] - 1, j - k + 1, buf))
      k = j + 1
    end
  end
  nothing
end

# beliefs store -> graph (setValKDE! with AMP's own bandwidth selection; synchronises the context's stream)
function download!(st::RomeStore, dfg::AbstractDFG, labels::AbstractVector{Symbol}; solveKey::Symbol=:default)
  for l in labels
    T = typeof(getVariableType(dfg, l)); c = _tcode(T)
    buf = Vector{Float64}(undef, st.N * _ptlen(c))
    GC.@preserve buf check(ccall((:rome_store_download, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Float64}),
                                 st.h, 2, c, st.index[l], 1, buf))
    P = eltype(getVal(dfg, l; solveKey))
    setValKDE!(dfg, l, manikde!(getManifold(T), collect(reinterpret(P, buf))), false, 1.0; solveKey)
  end
  nothing
end

# One clique (frontals in Gibbs order) or a frontier of independent cliques (`group`: frontals with the same group id are updated
# together, groups in order) as a plan over the store.  mirror: label => block of a device exchange buffer the product kernel ALSO
# writes the new belief to (run!(…; mirror_out, mirror_stride)); stream ids (`up_stream`, per-row streams) default to positions.
mutable struct RomeUpsolvePlan
  h::Ptr{Cvoid}
  store::RomeStore
  opts::RomeOpts
  has_mirror::Bool
end

function RomeUpsolvePlan(st::RomeStore, dfg::AbstractDFG, frontals::AbstractVector{Symbol}, factors::AbstractVector{<:DFGFactor};
                         gibbsIters::Integer=3, Niter::Integer=1, sequential::Bool=true, group::Union{Nothing,Vector{Int32}}=nothing,
                         mirror::Union{Nothing,Dict{Symbol,Int}}=nothing, solveKey::Symbol=:default)
  touching = [fc for fc in factors if any(in(getVariableOrder(fc)), frontals)]
  bad = [getLabel(fc) for fc in touching if !_batchable(fc)]
  isempty(bad) || throw(ArgumentError("RomeUpsolvePlan: factors $(bad) are outside the accelerated set; this clique stays with IIF"))
  pairs = [(fc, l) for l in frontals for fc in touching if l in getVariableOrder(fc)]
  t = _clique_tables(dfg, pairs; solveKey, index = st.index)
  upt = Int32[_tcode(typeof(getVariableType(dfg, l))) for l in frontals]; upv = Int32[st.index[l] for l in frontals]
  grp = group === nothing ? Int32[] : group
  mir = mirror === nothing ? Int32[] : Int32[get(mirror, l, -1) for l in frontals]
  o = _points_opts(dfg, st.N)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve t upt upv grp mir begin
    q = _clique_host(t, Float64[], Float64[], Float64[], Float64[], Float64[], (length(st.labels[1]), length(st.labels[2]), length(st.labels[3])))
    u = RomeCliqueUpsolveHost(q, Int32(gibbsIters), Int32(Niter), Int32(sequential ? 0 : 1), Int32(length(frontals)), _p(upt), _p(upv),
                              0, 0, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                              C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,          # no host outputs: run! copies nothing and does not wait
                              _p(grp), Ptr{Int32}(C_NULL), _p(mir), Int32(0), Int32(0), Int32(0), Int32(0), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL), Ptr{Int32}(C_NULL))
    check(ccall((:rome_upsolve_plan_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RomeOpts}, Ref{RomeCliqueUpsolveHost}, Ref{Ptr{Cvoid}}),
                ctx().h, st.h, o, u, r))
  end
  p = RomeUpsolvePlan(r[], st, o, mirror !== nothing)
  finalizer(x -> ccall((:rome_upsolve_plan_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), p)
end

# gibbs_iters x {proposals -> manikde! bandwidths -> multiscale Gibbs product -> in-place write}, asynchronous on the context's stream.
# A fresh seed per run: the draws of two runs of a plan must differ.
function run!(p::RomeUpsolvePlan; seed::Integer=rand(UInt64), stream_offset::Integer=0,
              mirror_out::Ptr{Float64}=Ptr{Float64}(C_NULL), mirror_stride::Integer=0)
  d = p.opts
  o = RomeOpts(d.n_particles, d.solver, d.max_iters, d.inflate_cycles, d.tol, d.inflation, seed, stream_offset, d.layout, d.presampled, d.spread_nh, d.nullhypo)
  check(ccall((:rome_upsolve_plan_run, LIB), Cint, (Ptr{Cvoid}, Ref{RomeOpts}, Ptr{Float64}, Int64), p.h, o, mirror_out, mirror_stride))
end

# The receive side of an exchange: blocks src_block[k] (of `stride` doubles; 0 = 6 N) of a device buffer -> the store's variables.
mutable struct RomeScatterPlan
  h::Ptr{Cvoid}
  store::RomeStore
end
function RomeScatterPlan(st::RomeStore, dfg::AbstractDFG, labels::AbstractVector{Symbol}, src_blocks::AbstractVector{<:Integer}; stride::Integer=0)
  ty = Int32[_tcode(typeof(getVariableType(dfg, l))) for l in labels]; va = Int32[st.index[l] for l in labels]; sb = Int32.(src_blocks)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve ty va sb check(ccall((:rome_scatter_plan_create, LIB), Cint,
      (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64, Ref{Ptr{Cvoid}}),
      ctx().h, st.h, length(labels), _p(ty), _p(va), _p(sb), stride, r))
  sp = RomeScatterPlan(r[], st)
  finalizer(x -> ccall((:rome_scatter_plan_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), sp)
end
run!(sp::RomeScatterPlan, src_dev::Ptr{Float64}) = check(ccall((:rome_scatter_plan_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), sp.h, src_dev))

# ---- block operations inside a store (rome_blockop_plan): the glue between tree levels that stays on the device --------------------------
# op = :copy (a clique's sub-graph starts from the graph's current values), :anchor (N copies of the mean of a belief: the clique conditions
# on its anchor separator being there), :relative (samples of anchor^-1 * s, or (bearing, range) of a landmark: the measurement samples a
# p2p2_meas / br*_meas row of the parent consumes).  Entries are (a, [b,] dst) labels of the store; Python twin: rome_jl_amd.tree.BlockOpPlan.
mutable struct RomeBlockOpPlan
  h::Ptr{Cvoid}
  store::RomeStore
end
# :compose (round 6: dst_i = A'_i (+) B'_i on Pose2 blocks of relative-pose samples; types[k] |= 0x100 / 0x200 takes A^-1 / B^-1 -- the pair
# marginal c^-1 k = (v^-1 c)^-1 (+) (v^-1 k) of an eliminated pose's neighbours; `inflate` (n x 2: translation, heading) scales the composed
# deviations about their mean: star-mesh transform) and :mix (pooling of independent passes: types[k] |= p << 8).  Python twin of the
# elimination driver: rome_jl_amd.elimination.RelativeEliminationSolver.
const _BLOCKOPS = Dict(:copy => Int32(0), :anchor => Int32(1), :relative => Int32(2), :compose => Int32(3), :mix => Int32(4))
function RomeBlockOpPlan(st::RomeStore, op::Symbol, types::AbstractVector{<:Integer}, a::AbstractVector{<:Integer}, b::AbstractVector{<:Integer},
                         dst::AbstractVector{<:Integer}, inflate::AbstractMatrix{Float64})
  ty = Int32.(types); va = Int32.(a); vb = Int32.(b); vd = Int32.(dst)
  pr = collect(transpose(inflate))       # (n, 2) row-major for the C side
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve ty va vb vd pr check(ccall((:rome_blockop_plan_create_ex, LIB), Cint,
      (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
      ctx().h, st.h, _BLOCKOPS[op], length(ty), _p(ty), _p(va), _p(vb), _p(vd), pointer(pr), r))
  bp = RomeBlockOpPlan(r[], st)
  finalizer(x -> ccall((:rome_blockop_plan_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), bp)
end
function RomeBlockOpPlan(st::RomeStore, op::Symbol, types::AbstractVector{<:Integer}, a::AbstractVector{<:Integer}, b::AbstractVector{<:Integer},
                         dst::AbstractVector{<:Integer})
  ty = Int32.(types); va = Int32.(a); vb = Int32.(b); vd = Int32.(dst)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve ty va vb vd check(ccall((:rome_blockop_plan_create, LIB), Cint,
      (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
      ctx().h, st.h, _BLOCKOPS[op], length(ty), _p(ty), _p(va), _p(vb), _p(vd), r))
  bp = RomeBlockOpPlan(r[], st)
  finalizer(x -> ccall((:rome_blockop_plan_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), bp)
end
run!(bp::RomeBlockOpPlan) = check(ccall((:rome_blockop_plan_run, LIB), Cint, (Ptr{Cvoid},), bp.h))
synchronize() = check(ccall((:rome_ctx_synchronize, LIB), Cint, (Ptr{Cvoid},), ctx().h))

# ---- parametric path: batched whitened residuals + Jacobians (rome_linearize) ------------------------------
# kind: 0 PriorPose2, 1 Pose2Pose2, 2 Pose2Point2BearingRange, 3 PriorPoint2, 4 Pose3Pose3, 5 PriorPose3
function linearize(kind::Integer, μ::Matrix{Float64}, W::Array{Float64,3}, xa::Matrix{Float64}, xb::Union{Nothing,Matrix{Float64}})
  dz, dr, da, db = ((3,3,3,0), (3,3,3,3), (2,2,3,2), (2,2,2,0), (6,6,6,6), (6,6,6,0))[kind + 1]
  F = size(μ, 2)                                  # columns = factors (column-major == row-major F x d on the C side)
  r = Matrix{Float64}(undef, dr, F); Ja = Array{Float64}(undef, da, dr, F)
  Jb = db > 0 ? Array{Float64}(undef, db, dr, F) : nothing
  check(ccall((:rome_linearize, LIB), Cint,
    (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
    ctx().h, kind, F, μ, W, xa, xb === nothing ? C_NULL : xb, r, Ja, Jb === nothing ? C_NULL : Jb))
  r, Ja, Jb                                       # Ja[:, :, f] is the TRANSPOSE of factor f's row-major dr x da block
end

# ---- manikde! bandwidths and getKDEMax on the device (rome_kde_bandwidth / rome_kde_max) -------------------------
# X: N x d coordinate matrix of ONE belief (column k = coordinate k: exactly the [dim][N] block the C side takes).
function kde_bandwidths(X::Matrix{Float64}; circular_mask::Integer = size(X, 2) == 3 ? 0b100 : 0)
  N, d = size(X)
  bw = Vector{Float64}(undef, d)
  check(ccall((:rome_kde_bandwidth, LIB), Cint,
    (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}, UInt32, Float64, Float64, Ptr{Float64}),
    ctx().h, d, 1, N, X, UInt32(circular_mask), 0.0, 0.0, bw))
  bw
end

function kde_max(X::Matrix{Float64}, bw::Vector{Float64} = kde_bandwidths(X))
  N, d = size(X)
  m = Vector{Float64}(undef, d)
  check(ccall((:rome_kde_max, LIB), Cint,
    (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}),
    ctx().h, d, 1, N, X, bw, 0, m))
  m
end

# ---- residual KATs through the library (mirrors calcFactorResidualTemporary) ---------------------------
function residual_pose2pose2(z::AbstractMatrix, p::AbstractMatrix, q::AbstractMatrix)   # 3 x n each
  r = similar(z)
  check(ccall((:rome_residual_pose2pose2, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              ctx().h, size(z, 2), z, p, q, r))
  r
end

end # module
