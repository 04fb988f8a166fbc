// hostcheck.cpp -- TEST TOOL, not a product path.  Compiles the per-env device functions of
// csrc/shems_core.h as ordinary host C++ (g++ -ffp-contract=off) so that the hand-typed
// Float32/Float64 logic the GPU threads execute can be compared with the oracle inside the
// GPU-less build container.  The GPU parity tests (-m gpu) remain the authoritative check.
#include <cstring>
#include "../../master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd/csrc/shems_core.h"

using namespace shems;

extern "C" {

// obs[9], table rows for idx (current) and idx+1; returns reward, new obs, flows (12), B, EV.
int hc_step(const shems_config *c, const float *obs_in, const float *row_cur, const float *row_next,
            const float *a, int track_mode, float *obs_out, double *reward, double *flows12, float *b_ev)
{
    const EnvIn s{obs_in[0], obs_in[1], obs_in[2], obs_in[3], obs_in[4], obs_in[5]};
    float B, EV, Bt, EVt;
    if (track_mode >= 0) { Bt = a[0]; EVt = a[1]; action_drl(*c, s, Bt, EVt, B, EV); }
    else { Bt = 0.f; EVt = 0.f; B = a[0]; EV = a[1]; }
    float sb, se; StepFlows f;
    step_flows(*c, s, EVt, B, EV, track_mode < 0, sb, se, *reward, f);
    if (row_next[0] >= 0.0f && row_cur[0] == -1.0f) se = row_next[1];
    obs_out[0] = sb; obs_out[1] = se; obs_out[2] = row_next[0]; obs_out[3] = row_next[2]; obs_out[4] = row_next[3];
    obs_out[5] = row_next[4]; obs_out[6] = row_next[5]; obs_out[7] = row_next[6]; obs_out[8] = row_next[7];
    const double fl[12] = {f.PV_DE, f.B_DE, f.GR_DE, f.PV_B, f.PV_GR, f.PV_EV, f.B_EV, f.GR_EV, f.EX_EV,
                           f.profit, f.discomfort, f.penalty};
    std::memcpy(flows12, fl, sizeof fl);
    b_ev[0] = B; b_ev[1] = EV;
    return 0;
}

void hc_action(const shems_config *c, const float *obs, const float *targets, int rule, float *out)
{
    const EnvIn s{obs[0], obs[1], obs[2], obs[3], obs[4], obs[5]};
    if (rule) action_rule(*c, s, out[0], out[1]); else action_drl(*c, s, targets[0], targets[1], out[0], out[1]);
}

float hc_scale_action(float a) { return scale_action(a); }

int hc_resolve_start(const float *table, int nrow, int maxsteps, int idx0)
{
    return resolve_start(idx0, nrow, maxsteps, [&](int32_t r) { return table[(size_t)(r - 1) * 8]; });
}

}
