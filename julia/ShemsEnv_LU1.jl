# ShemsEnv_LU1.jl -- drop-in replacement for RL-SHEMS/RL_environments/envs/shems_LU1.jl over libshems_hip.so.
#
# `include` this file where input.jl:151 includes shems_LU1.jl (same module name, so `using .ShemsEnv_LU1: Shems` at
# input.jl:152 keeps working).  Every method below is a `ccall` into the C ABI of include/shems_hip.h; the environment's
# arithmetic runs on the MI355X (csrc/shems_env.hip), the CSV is parsed ONCE in the constructor instead of on every
# reset!/step! (LU1:217, 265).
#
# STATUS: Julia is installed neither in the build container nor on the GPU box, so this file has never been executed
# (INTEGRATION.md).  It is kept next to the Python ctypes mirror (<package>/env.py), which binds the identical entry
# points and is what the test-suite drives; the two are meant to be read side by side.
#
# What the callers of the reference rely on, and where it is kept here:
#   input.jl:180-183   STATE_SIZE = length(env.state); ACTION_SIZE = length(env.a);
#                      ACTION_BOUND_HI = maximum(env.a); ACTION_BOUND_LO = minimum(env.a)
#                      -> env.a is a ShemsAction whose minimum/maximum are the BOUNDS (0,0)/(1,1), not the extrema of the
#                         current action (LU1:146-155); scale_action (DDPG.jl:178-184) then maps [-1,1] -> [0,1].
#   DDPG.jl:199, MPS:15  copy(env.state) -> Vector{Float32}(9)
#   DDPG.jl:205-212      step!(env, s, a; track) -> (r::Float64, s′::Vector{Float32}) [+ results::Matrix{Float64} 1x23]
#   DDPG.jl:209, MPS     action(env, track) -> Vector{Float32}[B, EV]
#   DDPG.jl:229-233      finished(env, s′) -> false
module ShemsEnv_LU1

using Reinforce: AbstractEnvironment
import Reinforce: reset!, action, finished, step!, state
using Distributions: Uniform
using Random
using CSV, DataFrames

export Shems, reset!, step!, action, finished, state, track_pass

const LIB = get(ENV, "SHEMS_HIP_LIB", joinpath(@__DIR__, "..", "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd", "libshems_hip.so"))

# ---- error mapping (include/shems_hip.h: SHEMS_ERR_*) ---------------------------------------------------------------
function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:shems_last_error, LIB), Cstring, ()))
    rc == -3 && throw(BoundsError(msg))          # SHEMS_ERR_INDEX: next_state! would read row idx+1 > nrow (LU1:265-279)
    error("libshems_hip [$rc]: $msg")
end

# ---- module globals of shems_LU1.jl:17-59 ---------------------------------------------------------------------------
const Job_ID = ENV["JOB_ID"]                                   # LU1:17
const DISCOMFORT_WEIGHT_EV = 0.01f0                            # LU1:40
const DISC_POT = 2f0                                           # LU1:41
const penalty_weight = 0.1f0                                   # LU1:43
const charger_id = (parse(Int, Job_ID) ÷ 100) % 100            # LU1:45: third and fourth last digit of JOB_ID

const capacities = Dict{Int, Tuple{Float32, Float32, Float64}}(   # LU1:47-59 (cap_ev, soc_max = nominal * 0.9 in Float32, rate_max)
    1 => (48.250f0, 7.5f0 * 0.9f0, 3.3), 2 => (36.271f0, 10f0 * 0.9f0, 3.3), 3 => (45.508f0, 10f0 * 0.9f0, 3.3),
    4 => (78.993f0, 11f0 * 0.9f0, 4.6), 5 => (37.207f0, 10f0 * 0.9f0, 4.6), 6 => (35.816f0, 15f0 * 0.9f0, 4.6),
    7 => (36.521f0, 12f0 * 0.9f0, 3.3), 8 => (45.728f0, 10f0 * 0.9f0, 3.3), 9 => (21.935f0, 7.5f0 * 0.9f0, 3.3),
    98 => (35.816f0, 7.5f0 * 0.9f0, 3.3), 97 => (78.993f0, 11f0 * 0.9f0, 4.6))

struct ShemsConfig                      # == shems_config (48 bytes, include/shems_hip.h)
    cap_ev::Float32; soc_max::Float32; rate_max::Float64
    disc_weight::Float64; disc_pot::Float64; penalty_weight::Float32
    table_row0::Int32; nrow::Int32; reserved::Int32
end

# ---- ShemsState / ShemsAction: the vector types callers index and measure (LU1:101-167) ------------------------------
mutable struct ShemsAction{T<:AbstractFloat} <: AbstractVector{T}
    B::T
    EV::T
end
ShemsAction() = ShemsAction(0.7f0, 1f0)                        # LU1:151
Base.size(::ShemsAction) = (2,)
Base.minimum(::ShemsAction) = (0f0, 0f0)                       # LU1:154: the action BOUNDS, whatever the current action is
Base.maximum(::ShemsAction) = (1f0, 1f0)                       # LU1:155
Base.getindex(a::ShemsAction, i::Int) = i == 1 ? a.B : i == 2 ? a.EV : throw(BoundsError(a, i))
function Base.setindex!(a::ShemsAction, x, i::Int)
    i == 1 ? (a.B = x) : i == 2 ? (a.EV = x) : throw(BoundsError(a, i))
end

# ---- the environment ------------------------------------------------------------------------------------------------
mutable struct Shems <: AbstractEnvironment      # same public fields as LU1:169-177
    state::Vector{Float32}                       # [Soc_b, Soc_ev, c_ev, d_e, g_e, p_buy, h_cos, h_sin, season]
    reward::Float64
    a::ShemsAction{Float32}
    step::Int
    maxsteps::Int
    idx::Int
    path::String
    handle::Ptr{Cvoid}                           # shems_handle
    nrow::Int                                    # rows of the table, counted once (LU1:225 calls nrow(df) per reset)
end
Base.size(::Shems) = (7,)                        # LU1:179

const COLS = (:h_countdown, :soc_ev, :electkwh, :PV_generation, :p_buy, :hour_cos, :hour_sin, :season)

function Shems(maxsteps, path)                   # LU1:203
    df = CSV.read(path, DataFrame)
    n = nrow(df)
    rows = Matrix{Float32}(undef, 8, n)          # column-major 8 x nrow == C [nrow][8]
    for (j, c) in enumerate(COLS)
        rows[j, :] .= Float32.(df[!, c])         # Float32(Float64 cell): the rounding LU1:251-260 applies on store
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    gpu = parse(Int, get(ENV, "GPU_ID", "0"))    # DDPG_reinforce_charger_v1.jl:12-14
    check(ccall((:shems_create, LIB), Cint, (Int64, Int32, Cint, Ptr{Ptr{Cvoid}}), 1, maxsteps, gpu, h))
    check(ccall((:shems_set_tables, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), h[], rows, n))
    cap, socmax, rate = capacities[charger_id]
    # Market(0.2f0, DISCOMFORT_WEIGHT_EV, DISC_POT) stores Float64(Float32 literal) (LU1:90-99)
    cfg = Ref(ShemsConfig(cap, socmax, rate, Float64(DISCOMFORT_WEIGHT_EV), Float64(DISC_POT), penalty_weight, 0, n, 0))
    check(ccall((:shems_set_configs, LIB), Cint, (Ptr{Cvoid}, Ptr{ShemsConfig}, Int32, Ptr{UInt16}), h[], cfg, 1, C_NULL))
    env = Shems(Float32[0, 0, -1, 0, 0, 0, 1, 0, 1], 0.0, ShemsAction(), 0, maxsteps, 1, path, h[], n)   # ShemsState() LU1:115
    st = Int32[1]; zero_step = Int32[0]
    check(ccall((:shems_set_state, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Int32}, Ptr{Int32}), h[], env.state, st, zero_step))
    finalizer(e -> ccall((:shems_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), env)
    return env
end

function pull!(env::Shems)
    idx = Ref{Int32}(0); st = Ref{Int32}(0)
    check(ccall((:shems_get_state, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Int32}, Ptr{Int32}), env.handle, env.state, idx, st))
    env.idx = idx[]; env.step = st[]
    return env
end

function reset!(env::Shems; rng=0)               # LU1:206-262
    if rng == -1                                 # tracking / evaluation: Soc_b = mid, idx = 1 (LU1:220-222)
        check(ccall((:shems_reset, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Float32}), env.handle, 1, C_NULL, C_NULL))
    else
        # The two draws of LU1:224-225 stay on the Julia side (same MersenneTwister streams as the reference); the extension loop
        # LU1:227-246 runs on the device.  Its redraw `rand(MersenneTwister(rng), ...)` re-seeds, i.e. returns idx0 again.
        soc_max = capacities[charger_id][2]
        socb = Float32[rand(MersenneTwister(rng), Uniform(0f0, soc_max))]
        idx0 = Int32[rand(MersenneTwister(rng), 1:(env.nrow - env.maxsteps))]
        check(ccall((:shems_reset, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Float32}), env.handle, 0, idx0, socb))
    end
    env.reward = 0.0
    env.a = ShemsAction()
    return pull!(env)
end

function step!(env::Shems, s, a; track=0)        # LU1:343-485 (`s` is ignored there as well: LU1:344 reads env.state)
    act = Float32[a[1], a[2]]
    r = Ref{Float64}(0.0)
    res = zeros(Float64, 1, 23)
    mode = track == 0 ? 0 : (track > 0 ? 1 : -1)
    check(ccall((:shems_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cint, Ptr{Float64}, Ptr{Float32}, Ptr{Float64}),
                env.handle, act, mode, r, env.state, track == 0 ? C_NULL : res))
    env.reward = r[]
    env.a = track >= 0 ? ShemsAction(act[1], act[2]) : ShemsAction(0f0, 0f0)       # LU1:349, 351-353
    pull!(env)
    return track == 0 ? (env.reward, copy(env.state)) : (env.reward, copy(env.state), res)
end

function action(env::Shems, a::ShemsAction)      # LU1:283-316: SoC targets -> kWh set-points, Float32.([B, EV])
    out = zeros(Float32, 2)
    check(ccall((:shems_action, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), env.handle, Float32[a.B, a.EV], out))
    return out
end

function action(env::Shems, track::Real=-1)      # LU1:318-340: rule-based controller
    out = zeros(Float32, 2)
    check(ccall((:shems_rule_action, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out))
    return out
end

# inference(env; track != 0) (memory_plotting_saving.jl:62-89) without a Julia-side loop: the whole pass over the data set -- reset!(rng = -1),
# then EP_LENGTH hours of { a = actor(normalize(s)) or action(env, track); step!(env, s, a; track) } -- is ONE call.  `actor_params` = the
# actor's parameters in Flux.params order, each weight matrix flattened column-major (vcat(vec.(Flux.params(actor))...) on the CPU copy of
# the Chain); for track < 0 (rule-based) pass nothing.  Returns (reward_eps::Float64, results::Matrix{Float64} nsteps x 23), what
# episode!(...; track) returns to inference, ready for write_to_results_file.
function track_pass(env::Shems, nsteps::Integer; track=1, actor_params::Vector{Float32}=Float32[], s_min::Vector{Float32}=Float32[],
                    s_max::Vector{Float32}=Float32[])
    reset!(env; rng=-1)
    res = Matrix{Float64}(undef, 23, nsteps)                  # column-major 23 x nsteps == C [nsteps][23]
    ret = Float64[0.0]
    mode = track > 0 ? 1 : -1
    check(ccall((:shems_track, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint, Int32, Ptr{Float64}, Ptr{Float64}),
                env.handle, track > 0 ? actor_params : C_NULL, track > 0 ? s_min : C_NULL, track > 0 ? s_max : C_NULL, mode, nsteps, res, ret))
    pull!(env)
    env.reward = ret[1]
    return ret[1], permutedims(res)
end

finished(env::Shems, s′) = false                 # LU1:487-502
state(env::Shems) = env.state

end # module
