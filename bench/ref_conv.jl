# bench/ref_conv.jl -- the REFERENCE's own timing of the hot path (SURVEY.md §8(d) "CPU reference timing", item 2).
#
#   JULIA_NUM_THREADS=$(nproc) julia --project=<env with RoME 0.24 / IncrementalInference 0.35> bench/ref_conv.jl [manhattan.g2o] [max_edges] [reps]
#
# Times `approxConvBelief` -- IIF's N x inflateCycles x {entropy, Optim NelderMead root-find} loop around RoME's Pose2Pose2 functor
# (src/factors/Pose2D.jl:51-67) -- for every (factor, direction) of the Manhattan pose graph, exactly the unit bench.py counts
# ("factor convolutions/sec (N=100)"), on the host cores of this box with Threads.@threads over convolutions, >= 5 repetitions,
# median.  Prints ONE JSON line that bench.py merges into `cpu_baseline.julia_reference` when `julia` is on PATH.
# WRITTEN BLIND: there is no Julia toolchain in the build image (and none on the GPU box), so this file has never been executed;
# bench.py records "julia reference not runnable on this box" in that case.
using RoME, IncrementalInference, DistributedFactorGraphs
using Statistics, Dates

g2o    = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden", "manhattan.g2o")
nedges = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 500        # a bounded prefix: the full graph takes minutes per repetition on a CPU
reps   = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 5

# the graph of examples/ManhattanDatasetBatch.jl:28-37 (prior on x0, every g2o edge as a Pose2Pose2)
fg = initfg()
getSolverParams(fg).N = 100
addVariable!(fg, :x0, Pose2)
addFactor!(fg, [:x0], PriorPose2(MvNormal(zeros(3), diagm([0.1, 0.1, 0.05] .^ 2))))
instructions = importG2o(g2o)
for ins in Iterators.take(Iterators.filter(i -> i[1] == "EDGE_SE2", instructions), nedges)
  parseG2oInstruction!(fg, ins)
end
initAll!(fg)                                      # every variable gets N particles (dead-reckoning through the factors)

pairs = Tuple{Symbol,Symbol}[]
for f in lsf(fg)
  vo = getVariableOrder(fg, f)
  length(vo) == 2 || continue
  push!(pairs, (f, vo[2])); push!(pairs, (f, vo[1]))
end

function sweep(fg, pairs)
  Threads.@threads for k in eachindex(pairs)
    approxConvBelief(fg, pairs[k][1], pairs[k][2])
  end
end

sweep(fg, pairs[1:min(end, 2 * Threads.nthreads())])    # compile
times = Float64[]
for _ in 1:reps
  t = @elapsed sweep(fg, pairs)
  push!(times, t)
end
med = median(times)
println("{\"kind\": \"reference\", \"what\": \"IIF approxConvBelief over $(length(pairs)) (factor, direction) pairs of the first $(nedges) Manhattan edges, N=100\", " *
        "\"value\": $(length(pairs) / med), \"unit\": \"convolutions/s\", \"cores\": $(Threads.nthreads()), \"reps\": $(reps), " *
        "\"median_s\": $(med), \"min_s\": $(minimum(times)), \"max_s\": $(maximum(times))}")
