# DDPG_hip.jl -- the learner side of the reference (RL-SHEMS/algorithms/DDPG.jl + src/memory_plotting_saving.jl) over libshems_hip.so:
# N households per launch, the fused act -> scale_action -> step! -> remember kernel, the GPU-resident replay ring and replay() as HIP
# kernels.  `include` it where DDPG_reinforce_charger_v1.jl:24 includes algorithms/DDPG.jl; the function names and call order are the
# reference's (populate_memory -> min_max_buffer -> run_episodes -> inference, DDPG_reinforce_charger_v1.jl:27-105).
#
# STATUS: Julia is installed neither in the build container nor on the GPU box, so this file has never been executed (INTEGRATION.md).
# It is the Julia twin of <package>/ddpg.py + env.py + harness.py (which bind the identical entry points and are what the test-suite
# drives); tests/test_abi_host.py checks it statically: every ccall names a declared and exported entry point with ABI-identical argument
# types, and every struct mirror below has the field types of its C struct in include/shems_hip.h, in order.
#
# Device memory comes straight from the HIP runtime (hipMalloc / hipMemcpy through ccall), so the module needs no GPU array package; with
# AMDGPU.jl a caller can pass `pointer(::ROCArray)` wherever a DevBuf's `ptr` goes.  Network parameters are ONE flat Float32 vector per
# network in Flux's own order: `vcat(vec.(Flux.params(chain))...)` of Chain(Dense(in, L1, relu), Dense(L1, L2, relu), Dense(L2, out[, tanh]))
# (DDPG.jl:30-46) IS the layout (each W a column-major out x in matrix), so checkpoints convert without touching a number.
module DDPG_hip

using Random
using Statistics: mean

export EnvBatch, Agent, ReplayRing, act, act_step!, replay, populate_memory, min_max_buffer, episode!, run_episodes, inference, train_steps!,
       flat_params, set_params!, actor_params, STATE_SIZE, ACTION_SIZE, LearnerGroup, flux!, learner_params

const LIB = get(ENV, "SHEMS_HIP_LIB", joinpath(@__DIR__, "..", "master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd", "libshems_hip.so"))
const HIP = "libamdhip64"

const STATE_SIZE, ACTION_SIZE = 9, 2                            # length(env.state), length(env.a) (input.jl:180-181)
const L1, L2 = 250, 500                                         # the tuned architecture (input09_08_on_01-09_eval.jl:66); (300, 600): the shems_wide_* entry points (INTEGRATION.md)
const N_ACTOR, N_CRITIC = 129002, 129001                        # SHEMS_ACTOR_PARAMS / SHEMS_CRITIC_PARAMS
const BATCH_SIZE, MEM_SIZE, EP_LENGTH = 120, 24000, 72          # input09_08_on_01-09_eval.jl:64-91

function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:shems_last_error, LIB), Cstring, ()))
    rc == -3 && throw(BoundsError(msg))                         # SHEMS_ERR_INDEX (shems_LU1.jl:265-279)
    error("libshems_hip [$rc]: $msg")
end
hipcheck(rc::Integer) = rc == 0 ? nothing : error("HIP runtime error $rc")

# ---- device buffers ----------------------------------------------------------------------------------------------------------------
mutable struct DevBuf{T}
    ptr::Ptr{T}
    n::Int
end
function DevBuf{T}(n::Integer) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    hipcheck(ccall((:hipMalloc, HIP), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, n * sizeof(T)))
    hipcheck(ccall((:hipMemset, HIP), Cint, (Ptr{Cvoid}, Cint, Csize_t), p[], 0, n * sizeof(T)))
    b = DevBuf{T}(Ptr{T}(p[]), n)
    finalizer(x -> ccall((:hipFree, HIP), Cint, (Ptr{Cvoid},), x.ptr), b)
    return b
end
function upload!(b::DevBuf{T}, host::Array{T}) where {T}
    length(host) == b.n || throw(DimensionMismatch("upload!: $(length(host)) values into a buffer of $(b.n)"))
    hipcheck(ccall((:hipMemcpy, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), b.ptr, host, b.n * sizeof(T), 1))   # hipMemcpyHostToDevice
    return b
end
function download(b::DevBuf{T}) where {T}
    host = Vector{T}(undef, b.n)
    hipcheck(ccall((:hipMemcpy, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), host, b.ptr, b.n * sizeof(T), 2))   # hipMemcpyDeviceToHost
    return host
end
function copy_dev!(dst::DevBuf{T}, src::DevBuf{T}) where {T}
    hipcheck(ccall((:hipMemcpy, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), dst.ptr, src.ptr, src.n * sizeof(T), 3))   # DeviceToDevice
    return dst
end
DevBuf(host::Array{T}) where {T} = upload!(DevBuf{T}(length(host)), host)

# ---- struct mirrors of include/shems_hip.h (field for field; checked by tests/test_abi_host.py) ----------------------------------------
struct ShemsConfig                     # shems_config
    cap_ev::Float32; soc_max::Float32; rate_max::Float64
    disc_weight::Float64; disc_pot::Float64; penalty_weight::Float32
    table_row0::Int32; nrow::Int32; reserved::Int32
end
struct ShemsView                       # shems_view
    n_envs::Int64
    maxsteps::Int32
    n_cfg::Int32
    obs::Ptr{Float32}
    idx::Ptr{Int32}
    step::Ptr{Int32}
    cfg_of_env::Ptr{UInt16}
    cfgs::Ptr{ShemsConfig}
    tables::Ptr{Float32}
    total_rows::Int64
    err::Ptr{Int32}
end
struct ShemsReplay                     # shems_replay
    capacity::Int64
    s::Ptr{Float32}
    a::Ptr{Float32}
    r::Ptr{Float32}
    s2::Ptr{Float32}
    done::Ptr{UInt8}
end
struct ShemsActParams                  # shems_act_params
    actor::Ptr{Float32}
    s_min::Ptr{Float32}
    s_max::Ptr{Float32}
    noise_mu::Float32
    noise_sigma::Float32
    train::Int32
    tick::UInt32
    seed::UInt64
    noise_kind::Int32
    ou_theta::Float32
    ou_dt::Float32
    eps::Float32
    ou_state::Ptr{Float32}
    noise_acc::Ptr{Float32}
end
struct ShemsRingWindow                 # shems_ring_window
    pos::Int64
    count::Int64
    offset::Int64
end
struct ShemsDdpg                       # shems_ddpg
    actor::Ptr{Float32}
    critic::Ptr{Float32}
    actor_t::Ptr{Float32}
    critic_t::Ptr{Float32}
    m_actor::Ptr{Float32}
    v_actor::Ptr{Float32}
    m_critic::Ptr{Float32}
    v_critic::Ptr{Float32}
    grad_actor::Ptr{Float32}
    grad_critic::Ptr{Float32}
    s_min::Ptr{Float32}
    s_max::Ptr{Float32}
    ws::Ptr{Float32}
    losses::Ptr{Float32}
    gamma::Float32
    tau::Float32
    batch::Int32
    flags::Int32
end
struct ShemsTrainLoop                  # shems_train_loop: the hour loop of episode! enqueued natively (shems_train_steps)
    view::ShemsView
    act::ShemsActParams
    ddpg::ShemsDdpg
    ring::ShemsReplay
    rewards_f32::Ptr{Float32}
    actor_pub::NTuple{2, Ptr{Float32}}
    window::Int64
    ring_pushed::Int64
    t::Int64
    updates::Int64
    env_seed::UInt64
    sample_seed::UInt64
    episode::UInt32
    ep_len::Int32
    updates_per_step::Int32
    mode::Int32
    eta_crit::Float64
    bp_crit::NTuple{2, Float64}
    eta_act::Float64
    bp_act::NTuple{2, Float64}
    sync::Ptr{Cvoid}
    dp::Ptr{Cvoid}
end

struct ShemsGroup                      # shems_group: `count` independent learners, learner l's buffers at learner 0's pointers + l * stride_bytes
    count::Int32
    reserved::Int32
    stride_bytes::Int64
    envs_per_learner::Int64
end
struct ShemsGroupW2T                   # shems_group_w2t: the tiled working layout of the two networks' layer-2 state (learner 0's regions)
    actor::Ptr{Float32}
    critic::Ptr{Float32}
end

# ---- N households on one table (the batched Shems; shems_LU1.jl:169-262) -------------------------------------------------------------
mutable struct EnvBatch
    handle::Ptr{Cvoid}
    n::Int
    maxsteps::Int
    nrow::Int
    view::ShemsView
end
"""EnvBatch(n, maxsteps, rows, cfg): `rows` = the 8 x nrow Float32 table (h_countdown, soc_ev, electkwh, PV_generation, p_buy, hour_cos,
hour_sin, season per column: what ShemsEnv_LU1.Shems parses from the CSV), `cfg` = the profile's ShemsConfig."""
function EnvBatch(n::Integer, maxsteps::Integer, rows::Matrix{Float32}, cfg::ShemsConfig; device::Integer=parse(Int, get(ENV, "GPU_ID", "0")))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:shems_create, LIB), Cint, (Int64, Int32, Cint, Ptr{Ptr{Cvoid}}), n, maxsteps, device, h))
    check(ccall((:shems_set_tables, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), h[], rows, size(rows, 2)))
    check(ccall((:shems_set_configs, LIB), Cint, (Ptr{Cvoid}, Ptr{ShemsConfig}, Int32, Ptr{UInt16}), h[], Ref(cfg), 1, C_NULL))
    v = Ref{ShemsView}()
    check(ccall((:shems_get_view, LIB), Cint, (Ptr{Cvoid}, Ptr{ShemsView}), h[], v))
    env = EnvBatch(h[], n, maxsteps, size(rows, 2), v[])
    finalizer(e -> ccall((:shems_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), env)
    return env
end
"reset!(env; rng): rng == -1 -> Soc_b = mid, idx = 1 (tracking); else the seeded draws of LU1:224-225 for every env (Philox keyed by (rng, episode, env))"
function reset_batch!(env::EnvBatch; rng::Integer=0, episode::Integer=0)
    if rng == -1
        check(ccall((:shems_reset_dev, LIB), Cint, (Ptr{ShemsView}, Cint, Ptr{Int32}, Ptr{Float32}, Ptr{Cvoid}), Ref(env.view), 1, C_NULL, C_NULL, C_NULL))
    else
        check(ccall((:shems_reset_seeded_dev, LIB), Cint, (Ptr{ShemsView}, UInt64, UInt32, Ptr{Cvoid}), Ref(env.view), rng, episode, C_NULL))
    end
    return env
end
check_error(env::EnvBatch) = check(ccall((:shems_check_error, LIB), Cint, (Ptr{Cvoid},), env.handle))

# ---- memory = CircularBuffer{Any}(MEM_SIZE) (input.jl:139-140), in HBM -----------------------------------------------------------------
mutable struct ReplayRing
    capacity::Int
    pushed::Int                        # transitions pushed so far; the push position is pushed % capacity
    s::DevBuf{Float32}; a::DevBuf{Float32}; r::DevBuf{Float32}; s2::DevBuf{Float32}; done::DevBuf{UInt8}
end
ReplayRing(capacity::Integer=MEM_SIZE) = ReplayRing(capacity, 0, DevBuf{Float32}(capacity * STATE_SIZE), DevBuf{Float32}(capacity * ACTION_SIZE),
                                                    DevBuf{Float32}(capacity), DevBuf{Float32}(capacity * STATE_SIZE), DevBuf{UInt8}(capacity))
Base.length(m::ReplayRing) = min(m.pushed, m.capacity)
ring_struct(m::ReplayRing) = ShemsReplay(m.capacity, m.s.ptr, m.a.ptr, m.r.ptr, m.s2.ptr, m.done.ptr)
ring_pos(m::ReplayRing) = m.pushed % m.capacity

# ---- the learner (actor, critic, targets, opt_act, opt_crit: DDPG.jl:30-46, input.jl:126-127) -----------------------------------------
mutable struct Agent
    actor::DevBuf{Float32}; critic::DevBuf{Float32}; actor_t::DevBuf{Float32}; critic_t::DevBuf{Float32}
    m_actor::DevBuf{Float32}; v_actor::DevBuf{Float32}; m_critic::DevBuf{Float32}; v_critic::DevBuf{Float32}
    grad_actor::DevBuf{Float32}; grad_critic::DevBuf{Float32}
    s_min::DevBuf{Float32}; s_max::DevBuf{Float32}; ws::DevBuf{Float32}; losses::DevBuf{Float32}
    gamma::Float32; tau::Float32; eta_act::Float64; eta_crit::Float64; batch::Int
    sigma::Float32; mu::Float32                         # GNoise(mu, sigma_act) (input.jl:190-205)
    seed::UInt64                                        # key of the noise / minibatch streams (the reference's rng_run)
    bp_actor::Vector{Float64}; bp_critic::Vector{Float64}   # Flux ADAM state beta^t, advanced after every step
    updates::Int
    tick::Int
end
"flat_params(chain): a Flux Chain of three Dense layers as the flat vector the kernels read (Flux.params order, column-major matrices)"
flat_params(params) = Float32.(vcat([vec(p) for p in params]...))

"""Agent(actor_params, critic_params; ...): the flat vectors of the freshly initialised Flux chains (DDPG.jl:21-46:
glorot_uniform hidden layers, U(-3f-3, 3f-3) output layer), e.g. flat_params(Flux.params(cpu(actor)))."""
function Agent(actor_params::Vector{Float32}, critic_params::Vector{Float32}; gamma=0.99f0, tau=1f-3, eta_act=1f-4, eta_crit=1f-3,
               batch::Integer=BATCH_SIZE, sigma=0.1f0, mu=0f0, seed::Integer=1231)
    length(actor_params) == N_ACTOR && length(critic_params) == N_CRITIC || throw(DimensionMismatch("expected $N_ACTOR / $N_CRITIC parameters (9/11 -> 250 -> 500 -> 2/1)"))
    nws = Ref{Int64}(0)
    check(ccall((:shems_ddpg_workspace_floats, LIB), Cint, (Ptr{Int64},), nws))
    z(n) = DevBuf{Float32}(n)
    ag = Agent(DevBuf(actor_params), DevBuf(critic_params), DevBuf(actor_params), DevBuf(critic_params),      # deepcopy(actor), DDPG.jl:38
               z(N_ACTOR), z(N_ACTOR), z(N_CRITIC), z(N_CRITIC), z(N_ACTOR), z(N_CRITIC),
               DevBuf(zeros(Float32, STATE_SIZE)), DevBuf(ones(Float32, STATE_SIZE)), z(nws[]), z(2),
               Float32(gamma), Float32(tau), Float64(Float32(eta_act)), Float64(Float32(eta_crit)), batch, Float32(sigma), Float32(mu), UInt64(seed),
               [0.9, 0.999], [0.9, 0.999], 0, 0)
    return ag
end
actor_params(ag::Agent) = download(ag.actor)            # what saveBSON turns back into a Chain (memory_plotting_saving.jl:263-268)
set_params!(ag::Agent, actor::Vector{Float32}) = (upload!(ag.actor, actor); ag)

ddpg_struct(ag::Agent) = ShemsDdpg(ag.actor.ptr, ag.critic.ptr, ag.actor_t.ptr, ag.critic_t.ptr, ag.m_actor.ptr, ag.v_actor.ptr,
                                   ag.m_critic.ptr, ag.v_critic.ptr, ag.grad_actor.ptr, ag.grad_critic.ptr, ag.s_min.ptr, ag.s_max.ptr,
                                   ag.ws.ptr, ag.losses.ptr, ag.gamma, ag.tau, Int32(ag.batch), Int32(0))
act_struct(ag::Agent, train::Bool, tick::Integer; noise_acc::Ptr{Float32}=Ptr{Float32}(C_NULL)) =
    ShemsActParams(ag.actor.ptr, ag.s_min.ptr, ag.s_max.ptr, ag.mu, ag.sigma, Int32(train), UInt32(tick % 0x100000000), ag.seed, Int32(0),
                   0f0, 0f0, 0f0, Ptr{Float32}(C_NULL), noise_acc)

"act(s_norm; train) (DDPG.jl:148-176) for m observations in device memory ([9 x m] column-major): returns the [2 x m] actions in [-1, 1]"
function act(ag::Agent, obs::DevBuf{Float32}, m::Integer; train::Bool=true, tick::Integer=ag.tick)
    out = DevBuf{Float32}(ACTION_SIZE * m)
    check(ccall((:shems_actor_forward_dev, LIB), Cint, (Ptr{ShemsActParams}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Cvoid}),
                Ref(act_struct(ag, train, tick)), obs.ptr, m, out.ptr, C_NULL))
    return out
end

"""One fused vector step of episode! (DDPG.jl:195-234) for every env: s = env.state; a = act(normalize(s)); step!(env, s, scale_action(a));
remember(s, a, r, s', false) for the `window` envs (ring === nothing: an evaluation step)."""
function act_step!(ag::Agent, env::EnvBatch; train::Bool=true, tick::Integer=ag.tick, returns::Union{Nothing, DevBuf{Float64}}=nothing,
                   ring::Union{Nothing, ReplayRing}=nothing, window::Integer=0, noise_acc::Union{Nothing, DevBuf{Float32}}=nothing)
    p = Ref(act_struct(ag, train, tick; noise_acc = noise_acc === nothing ? Ptr{Float32}(C_NULL) : noise_acc.ptr))
    ret = returns === nothing ? Ptr{Float64}(C_NULL) : returns.ptr
    if ring === nothing
        check(ccall((:shems_act_step_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Ptr{Float32}, Ptr{Float64}, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}, Ptr{ShemsReplay}, Ptr{ShemsRingWindow}, Ptr{Cvoid}),
                    Ref(env.view), p, C_NULL, C_NULL, C_NULL, C_NULL, ret, C_NULL, C_NULL, C_NULL))
    else
        w = Ref(ShemsRingWindow(ring_pos(ring), window, (ag.tick * window) % env.n))      # a rotating window of envs inserts (SURVEY.md 8d)
        check(ccall((:shems_act_step_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Ptr{Float32}, Ptr{Float64}, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}, Ptr{ShemsReplay}, Ptr{ShemsRingWindow}, Ptr{Cvoid}),
                    Ref(env.view), p, C_NULL, C_NULL, C_NULL, C_NULL, ret, Ref(ring_struct(ring)), w, C_NULL))
        ring.pushed += window
    end
    return nothing
end

"replay(; rng_rpl) (DDPG.jl:121-145): getData -> targets -> update_model!(critic) -> update_model!(actor) -> soft_update! x2, five launches"
function replay(ag::Agent, memory::ReplayRing; rng_rpl::Integer=ag.updates)
    check(ccall((:shems_ddpg_update, LIB), Cint,
                (Ptr{ShemsDdpg}, Ptr{ShemsReplay}, Int64, UInt64, UInt32, Int64, Int64, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Float32}, Ptr{Cvoid}),
                Ref(ddpg_struct(ag)), Ref(ring_struct(memory)), length(memory), ag.seed, rng_rpl % 0x100000000, 0, 0,
                ag.eta_crit, ag.bp_critic[1], ag.bp_critic[2], ag.eta_act, ag.bp_actor[1], ag.bp_actor[2], C_NULL, C_NULL))
    ag.bp_critic .*= [0.9, 0.999]
    ag.bp_actor .*= [0.9, 0.999]
    ag.updates += 1
    return nothing
end

"""train_steps!(ag, env, memory, k; t, episode, rng_ep): k vector steps of the training hour loop of episode! (DDPG.jl:195-234) -- act, step!,
remember, replay() -- enqueued by ONE foreign call (shems_train_steps, program order): what `for _ in 1:k act_step!(...); replay(...) end` launches,
without a ccall per launch.  `t` = vector steps done so far in this run (noise tick, window rotation, episode boundaries every EP_LENGTH
steps: reset!(env) with (rng_ep, episode)).  Returns (t, episode) advanced."""
function train_steps!(ag::Agent, env::EnvBatch, memory::ReplayRing, k::Integer; t::Integer=ag.tick, episode::Integer=1, rng_ep::Integer=ag.seed,
                      ep_len::Integer=EP_LENGTH, updates_per_step::Integer=1)
    window = min(env.n, max(1, memory.capacity ÷ ep_len))
    loop = Ref(ShemsTrainLoop(env.view, act_struct(ag, true, 0), ddpg_struct(ag), ring_struct(memory), Ptr{Float32}(C_NULL),
                              (Ptr{Float32}(C_NULL), Ptr{Float32}(C_NULL)), window, memory.pushed, t, ag.updates, UInt64(rng_ep), ag.seed,
                              UInt32(episode), Int32(ep_len), Int32(updates_per_step), Int32(0),          # SHEMS_LOOP_ORDERED
                              ag.eta_crit, (ag.bp_critic[1], ag.bp_critic[2]), ag.eta_act, (ag.bp_actor[1], ag.bp_actor[2]), C_NULL, C_NULL))
    check(ccall((:shems_train_steps, LIB), Cint, (Ptr{ShemsTrainLoop}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), loop, k, C_NULL, C_NULL))
    l = loop[]
    memory.pushed = l.ring_pushed
    ag.updates = l.updates
    ag.bp_critic .= l.bp_crit
    ag.bp_actor .= l.bp_act
    ag.tick = l.t
    check(ccall((:shems_train_loop_release, LIB), Cint, (Ptr{ShemsTrainLoop},), loop))
    return l.t, Int(l.episode)
end

"populate_memory(env; rng) (memory_plotting_saving.jl:9-29): uniform random actions until the buffer holds MIN_EXP_SIZE transitions"
function populate_memory(ag::Agent, env::EnvBatch, memory::ReplayRing; rng::Integer=ag.seed)
    n_ep = cld(memory.capacity, env.maxsteps)
    while length(memory) < memory.capacity
        reset_batch!(env; rng=rng, episode=0x7FFF0000 + memory.pushed ÷ env.maxsteps)
        k = min(env.n, n_ep)
        check(ccall((:shems_rollout_dev, LIB), Cint,
                    (Ptr{ShemsView}, Cint, Int32, UInt64, Ptr{Float64}, Ptr{ShemsReplay}, Int64, Int64, Ptr{Cvoid}),
                    Ref(env.view), 1, env.maxsteps, rng + memory.pushed, C_NULL, Ref(ring_struct(memory)), ring_pos(memory), k, C_NULL))
        memory.pushed += k * env.maxsteps
    end
    return memory
end

"s_min, s_max = min_max_buffer(MIN_EXP_SIZE; rng_mm) (memory_plotting_saving.jl:50-53)"
function min_max_buffer(ag::Agent, memory::ReplayRing; count::Integer=length(memory), rng_mm::Integer=ag.seed)
    check(ccall((:shems_minmax_dev, LIB), Cint, (Ptr{ShemsReplay}, Int64, Int64, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                Ref(ring_struct(memory)), length(memory), count, rng_mm, ag.s_min.ptr, ag.s_max.ptr, C_NULL))
    return download(ag.s_min), download(ag.s_max)
end

"""episode!(env; train, rng_ep) (DDPG.jl:186-242) for every env of the batch at once: returns the per-env sums of rewards (reward_eps)."""
function episode!(ag::Agent, env::EnvBatch, memory::Union{Nothing, ReplayRing}=nothing; train::Bool=true, rng_ep::Integer=0, episode::Integer=0,
                  num_steps::Integer=env.maxsteps)
    reset_batch!(env; rng=rng_ep, episode=episode)
    returns = DevBuf{Float64}(env.n)
    window = memory === nothing ? 0 : min(env.n, max(1, memory.capacity ÷ num_steps))
    for step in 0:(num_steps - 1)
        tick = (episode * 4096 + step) % 0x100000000
        act_step!(ag, env; train=train, tick=tick, returns=returns, ring=train ? memory : nothing, window=window)
        if train
            replay(ag, memory)
            ag.tick += 1
        end
    end
    check_error(env)
    return download(returns)
end

"""run_episodes(env_train, env_eval, total_reward, score_mean, best_run, noise_mean, test_every, test_runs, num_ep; train, ...)
(DDPG.jl:244-298): training episodes, an evaluation sweep every `test_every` episodes, the best-scoring actor kept."""
function run_episodes(ag::Agent, env_train::EnvBatch, env_eval::EnvBatch, memory::ReplayRing, num_ep::Integer; test_every::Integer=100,
                      rng_run::Integer=ag.seed, on_best=nothing)
    total_reward = zeros(Float32, num_ep)
    score_mean = zeros(Float64, cld(num_ep, test_every))
    best_score, best_run, best_actor = -100000.0, 0, Float32[]
    for i in 1:num_ep
        total_reward[i] = mean(episode!(ag, env_train, memory; train=true, rng_ep=rng_run, episode=i))
        if i % test_every == 1
            idx = cld(i, test_every)
            # every sweep runs the SAME test seeds "123" * test_ep (DDPG.jl:273-277): fixed reset key, env j = test episode j + 1
            score_mean[idx] = mean(episode!(ag, env_eval, nothing; train=false, rng_ep=123, episode=0, num_steps=EP_LENGTH))
            if score_mean[idx] > best_score
                best_score, best_run, best_actor = score_mean[idx], i, actor_params(ag)
                on_best === nothing || on_best(i, best_actor, total_reward, score_mean)      # saveBSON(...; idx=i, path="temp"), DDPG.jl:282-286
            end
        end
    end
    return total_reward, score_mean, best_run, best_actor
end

"""inference(env; track) (memory_plotting_saving.jl:62-89) as ONE launch: every env of the batch runs `nsteps` hours from reset!(rng = -1);
env e with the actor at `actors` + e * stride (a slab of the job's 80 actors, or stride 0).  Returns (returns, results [23 x nsteps x n])."""
function inference(env::EnvBatch, nsteps::Integer; track::Real=1, actors::Union{Nothing, DevBuf{Float32}}=nothing, stride_bytes::Integer=0,
                   s_min::Union{Nothing, DevBuf{Float32}}=nothing, s_max::Union{Nothing, DevBuf{Float32}}=nothing)
    reset_batch!(env; rng=-1)
    res = DevBuf{Float64}(23 * nsteps * env.n)
    ret = DevBuf{Float64}(env.n)
    if track > 0
        p = Ref(ShemsActParams(actors.ptr, s_min.ptr, s_max.ptr, 0f0, 0f0, Int32(0), UInt32(0), UInt64(0), Int32(0), 0f0, 0f0, 0f0,
                               Ptr{Float32}(C_NULL), Ptr{Float32}(C_NULL)))
        check(ccall((:shems_track_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Int64, Cint, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cvoid}),
                    Ref(env.view), p, stride_bytes, 1, nsteps, res.ptr, -1, ret.ptr, C_NULL))
    else
        check(ccall((:shems_track_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Int64, Cint, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cvoid}),
                    Ref(env.view), C_NULL, 0, -1, nsteps, res.ptr, -1, ret.ptr, C_NULL))
    end
    check_error(env)
    return download(ret), reshape(download(res), 23, nsteps, env.n)
end

# ---- learner groups: the scheduler's 40 seeds x 10 chargers in ONE process (RL-SHEMS_bs_scheduler_1179_08_on_01-98.sh:67-87) ------------------
const W2T_FLOATS = 32 * 4 * 64 * 64                              # SHEMS_W2T_FLOATS
pad4(n) = (n + 3) & ~3

"""LearnerGroup(actor_params, critic_params, envs_per_learner): `count = length(actor_params)` independent learners (one flat parameter vector
each: the scheduler's seeds / chargers), every learner's networks, targets, ADAM moments, tiled layer-2 regions, workspace, normalisation and
replay ring carved identically out of ONE slab; learner l owns the households [l * E, (l + 1) * E) of the env batch (E a multiple of 32).
replay / act_step! advance all of them with the same launches (shems_ddpg_group_update_tiled: eight launches for the whole group).
While the group trains, the layer-2 state lives in the tiled regions; flux!(g) brings the Flux-order blocks (what learner_params reads) up to date."""
mutable struct LearnerGroup
    count::Int; envs_per_learner::Int; capacity::Int
    slab::DevBuf{Float32}; slab_floats::Int; off::Dict{Symbol, Int}
    gamma::Float32; tau::Float32; eta_act::Float64; eta_crit::Float64; batch::Int; sigma::Float32; seed::UInt64
    bp_actor::Vector{Float64}; bp_critic::Vector{Float64}
    updates::Int; tick::Int; pushed::Int
    flux_valid::Bool; tiled_valid::Bool
end
function LearnerGroup(actors::Vector{Vector{Float32}}, critics::Vector{Vector{Float32}}, envs_per_learner::Integer; capacity::Integer=MEM_SIZE,
                      gamma=0.99f0, tau=1f-3, eta_act=1f-4, eta_crit=1f-3, batch::Integer=BATCH_SIZE, sigma=0.1f0, seed::Integer=1231)
    count = length(actors)
    count == length(critics) && count >= 1 && envs_per_learner % 32 == 0 || throw(ArgumentError("one actor and one critic per learner, env blocks of a multiple of 32"))
    nws = Ref{Int64}(0)
    check(ccall((:shems_ddpg_workspace_floats, LIB), Cint, (Ptr{Int64},), nws))
    off, o = Dict{Symbol, Int}(), 0
    for (name, n) in ((:actor, N_ACTOR), (:critic, N_CRITIC), (:actor_t, N_ACTOR), (:critic_t, N_CRITIC), (:m_actor, N_ACTOR), (:v_actor, N_ACTOR),
                      (:m_critic, N_CRITIC), (:v_critic, N_CRITIC), (:w2t_actor, W2T_FLOATS), (:w2t_critic, W2T_FLOATS), (:grad_actor, N_ACTOR),
                      (:grad_critic, N_CRITIC), (:s_min, STATE_SIZE), (:s_max, STATE_SIZE), (:losses, 2), (:ws, nws[]), (:ring_s, capacity * STATE_SIZE),
                      (:ring_a, capacity * ACTION_SIZE), (:ring_r, capacity), (:ring_s2, capacity * STATE_SIZE), (:ring_done, cld(capacity, 4)))
        off[name] = o
        o = pad4(o + n)
    end
    host = zeros(Float32, o, count)                              # column l = learner l's slab
    for l in 1:count
        length(actors[l]) == N_ACTOR && length(critics[l]) == N_CRITIC || throw(DimensionMismatch("expected $N_ACTOR / $N_CRITIC parameters per learner"))
        host[off[:actor] + 1:off[:actor] + N_ACTOR, l] = actors[l];   host[off[:actor_t] + 1:off[:actor_t] + N_ACTOR, l] = actors[l]      # deepcopy(actor), DDPG.jl:38
        host[off[:critic] + 1:off[:critic] + N_CRITIC, l] = critics[l]; host[off[:critic_t] + 1:off[:critic_t] + N_CRITIC, l] = critics[l]
        host[off[:s_max] + 1:off[:s_max] + STATE_SIZE, l] .= 1f0
    end
    return LearnerGroup(count, envs_per_learner, capacity, DevBuf(vec(host)), o, off, Float32(gamma), Float32(tau), Float64(Float32(eta_act)),
                        Float64(Float32(eta_crit)), batch, Float32(sigma), UInt64(seed), [0.9, 0.999], [0.9, 0.999], 0, 0, 0, true, false)
end
at(g::LearnerGroup, name::Symbol) = g.slab.ptr + 4 * g.off[name]                       # learner 0's block (Ptr{Float32} arithmetic is in bytes)
group_struct(g::LearnerGroup) = ShemsGroup(Int32(g.count), Int32(0), 4 * g.slab_floats, g.envs_per_learner)
w2t_struct(g::LearnerGroup) = ShemsGroupW2T(at(g, :w2t_actor), at(g, :w2t_critic))
ddpg_struct(g::LearnerGroup) = ShemsDdpg(at(g, :actor), at(g, :critic), at(g, :actor_t), at(g, :critic_t), at(g, :m_actor), at(g, :v_actor), at(g, :m_critic),
                                         at(g, :v_critic), at(g, :grad_actor), at(g, :grad_critic), at(g, :s_min), at(g, :s_max), at(g, :ws), at(g, :losses),
                                         g.gamma, g.tau, Int32(g.batch), Int32(0))
ring_struct(g::LearnerGroup) = ShemsReplay(g.capacity, at(g, :ring_s), at(g, :ring_a), at(g, :ring_r), at(g, :ring_s2), Ptr{UInt8}(at(g, :ring_done)))
function use_tiled!(g::LearnerGroup)
    g.tiled_valid && return nothing
    check(ccall((:shems_group_w2_to_tiled, LIB), Cint, (Ptr{ShemsDdpg}, Ptr{ShemsGroup}, Ptr{ShemsGroupW2T}, Ptr{Cvoid}),
                Ref(ddpg_struct(g)), Ref(group_struct(g)), Ref(w2t_struct(g)), C_NULL))
    g.tiled_valid = true
    return nothing
end
"flux!(g): bring the W2 ranges of the Flux-order blocks up to date (before learner_params, a checkpoint, or a single-learner call on a learner's buffers)"
function flux!(g::LearnerGroup)
    g.flux_valid && return g
    check(ccall((:shems_group_w2_to_flux, LIB), Cint, (Ptr{ShemsDdpg}, Ptr{ShemsGroup}, Ptr{ShemsGroupW2T}, Ptr{Cvoid}),
                Ref(ddpg_struct(g)), Ref(group_struct(g)), Ref(w2t_struct(g)), C_NULL))
    g.flux_valid = true
    return g
end
"learner_params(g, l): learner l's actor as the flat Flux-order vector (what saveBSON turns back into a Chain, memory_plotting_saving.jl:263-268)"
function learner_params(g::LearnerGroup, l::Integer)
    flux!(g)
    host = Vector{Float32}(undef, N_ACTOR)
    hipcheck(ccall((:hipMemcpy, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), host, at(g, :actor) + 4 * g.slab_floats * (l - 1), N_ACTOR * sizeof(Float32), 2))
    return host
end

"""populate_memory (memory_plotting_saving.jl:9-29) per learner on its own env block: uniform random actions until every ring is full (learner l: key rng + l).
The households of learner l are a sub-view of the batch (the same arrays, offset pointers), its ring the slab's ring blocks + l * stride."""
function populate_memory(g::LearnerGroup, env::EnvBatch; rng::Integer=g.seed)
    E, v, stride = g.envs_per_learner, env.view, 4 * g.slab_floats
    n_ep = cld(g.capacity, env.maxsteps)
    k = min(E, n_ep)
    pushed = 0
    while pushed < g.capacity
        reset_batch!(env; rng=rng, episode=0x7FFF0000 + pushed ÷ env.maxsteps)
        for l in 0:(g.count - 1)
            sub = ShemsView(E, v.maxsteps, v.n_cfg, v.obs + 4 * STATE_SIZE * E * l, v.idx + 4 * E * l, v.step + 4 * E * l,
                            v.cfg_of_env == C_NULL ? v.cfg_of_env : v.cfg_of_env + 2 * E * l, v.cfgs, v.tables, v.total_rows, v.err)
            r0 = ring_struct(g)
            ring = ShemsReplay(r0.capacity, r0.s + stride * l, r0.a + stride * l, r0.r + stride * l, r0.s2 + stride * l, r0.done + stride * l)
            check(ccall((:shems_rollout_dev, LIB), Cint,
                        (Ptr{ShemsView}, Cint, Int32, UInt64, Ptr{Float64}, Ptr{ShemsReplay}, Int64, Int64, Ptr{Cvoid}),
                        Ref(sub), 1, env.maxsteps, rng + l + pushed, C_NULL, Ref(ring), pushed % g.capacity, k, C_NULL))
        end
        pushed += k * env.maxsteps
    end
    g.pushed = pushed
    return g
end

"""min_max_buffer (memory_plotting_saving.jl:50-53) for every learner of the group in one launch (learner l: key seed + l)."""
function min_max_buffer(g::LearnerGroup; rng_mm::Integer=g.seed)
    n = min(g.pushed, g.capacity)
    check(ccall((:shems_minmax_group_dev, LIB), Cint, (Ptr{ShemsReplay}, Ptr{ShemsGroup}, Int64, Int64, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                Ref(ring_struct(g)), Ref(group_struct(g)), n, n, rng_mm, at(g, :s_min), at(g, :s_max), C_NULL))
    return nothing
end

"""One fused vector step for all learners (DDPG.jl:195-234): household i acts with learner i ÷ E's actor; with `window` > 0 every learner remembers
`window` transitions of its own block (window = 1: household 0 -- the reference's ONE transition per replay(), DDPG.jl:229-233)."""
function act_step!(g::LearnerGroup, env::EnvBatch; train::Bool=true, tick::Integer=g.tick, returns::Union{Nothing, DevBuf{Float64}}=nothing, window::Integer=0)
    env.n == g.count * g.envs_per_learner || throw(DimensionMismatch("the env batch must hold count * envs_per_learner households"))
    use_tiled!(g)
    p = Ref(ShemsActParams(at(g, :actor), at(g, :s_min), at(g, :s_max), 0f0, g.sigma, Int32(train), UInt32(tick % 0x100000000), g.seed, Int32(0),
                           0f0, 0f0, 0f0, Ptr{Float32}(C_NULL), Ptr{Float32}(C_NULL)))
    ret = returns === nothing ? Ptr{Float64}(C_NULL) : returns.ptr
    if window > 0
        w = Ref(ShemsRingWindow(g.pushed % g.capacity, window, window == 1 ? 0 : (g.tick * window) % g.envs_per_learner))
        check(ccall((:shems_act_step_group_tiled_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Ptr{ShemsGroup}, Ptr{ShemsGroupW2T}, Ptr{Float32}, Ptr{Float64}, Ptr{ShemsReplay}, Ptr{ShemsRingWindow}, Ptr{Cvoid}),
                    Ref(env.view), p, Ref(group_struct(g)), Ref(w2t_struct(g)), C_NULL, ret, Ref(ring_struct(g)), w, C_NULL))
        g.pushed += window
    else
        check(ccall((:shems_act_step_group_tiled_dev, LIB), Cint,
                    (Ptr{ShemsView}, Ptr{ShemsActParams}, Ptr{ShemsGroup}, Ptr{ShemsGroupW2T}, Ptr{Float32}, Ptr{Float64}, Ptr{ShemsReplay}, Ptr{ShemsRingWindow}, Ptr{Cvoid}),
                    Ref(env.view), p, Ref(group_struct(g)), Ref(w2t_struct(g)), C_NULL, ret, C_NULL, C_NULL, C_NULL))
    end
    return nothing
end

"replay (DDPG.jl:121-145) for every learner of the group: eight launches in all (the throughput form on the tiled layout); minibatch l = key seed + l"
function replay(g::LearnerGroup; rng_rpl::Integer=g.updates)
    use_tiled!(g)
    check(ccall((:shems_ddpg_group_update_tiled, LIB), Cint,
                (Ptr{ShemsDdpg}, Ptr{ShemsReplay}, Ptr{ShemsGroup}, Ptr{ShemsGroupW2T}, Int64, UInt64, UInt32, Float64, Float64, Float64, Float64, Float64, Float64, Int32, Ptr{Cvoid}),
                Ref(ddpg_struct(g)), Ref(ring_struct(g)), Ref(group_struct(g)), Ref(w2t_struct(g)), min(g.pushed, g.capacity), g.seed, UInt32(rng_rpl % 0x100000000),
                g.eta_crit, g.bp_critic[1], g.bp_critic[2], g.eta_act, g.bp_actor[1], g.bp_actor[2], Int32(0), C_NULL))
    g.flux_valid = false
    g.bp_critic .*= [0.9, 0.999]
    g.bp_actor .*= [0.9, 0.999]
    g.updates += 1
    return nothing
end

"""episode! (DDPG.jl:186-242) for all learners at once; window = 1 is the reference's update-to-data ratio.  Returns the per-household returns."""
function episode!(g::LearnerGroup, env::EnvBatch; train::Bool=true, num_steps::Integer=EP_LENGTH, rng_ep::Integer=0, episode::Integer=0, window::Integer=1)
    reset_batch!(env; rng=rng_ep, episode=episode)
    returns = DevBuf{Float64}(env.n)
    for step in 0:(num_steps - 1)
        act_step!(g, env; train=train, tick=(episode * 4096 + step) % 0x100000000, returns=returns, window=train ? window : 0)
        if train
            replay(g)
            g.tick += 1
        end
    end
    check_error(env)
    return download(returns)
end

end # module
