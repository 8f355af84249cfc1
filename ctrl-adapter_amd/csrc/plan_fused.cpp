// Fused denoise-step entry point: ControlNet forward and Ctrl-Adapter forward of one step as ONE call, overlapped.
//
// The reference runs them back to back in its denoise loops (sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1323 then
// :1338; svd/pipelines/...:684,709; i2vgen_xl/pipelines/...:957,1042), but adapter block i only reads ControlNet output i
// (model/ctrl_adapter.py:181-191), and the ControlNet is a long chain of small launches (M = N*64*64 rows and below) that
// leaves most of the 256 CUs idle.  Here the ControlNet runs on its plan's own HIP stream and signals each output with an
// event as soon as its zero-conv has been enqueued; every adapter block waits for exactly its input on its lane, so the
// 128x128 adapter blocks (slots 0-2, ready after the first ControlNet down block) run while the rest of the ControlNet is
// still in flight.  Fork / join with events only: hipGraph-capturable.  Results are bit-identical to the two separate calls.
#include "plan_common.h"

struct ctrl_controlnet;
struct ctrl_adapter;
int controlnet_forward_async(ctrl_controlnet* h, const void* sample, int sample_dtype, int N, int Hs, int Ws,
                             const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype, int Lk,
                             const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                             void* const* outs, int out_dtype, hipStream_t main_s, bool async,
                             hipEvent_t** out_events, hipEvent_t* done);
int adapter_forward_events(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                           const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                           int ehs_batch, int Lk, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                           void* stream, const hipEvent_t* in_ev);

extern "C" int ctrl_step_forward(ctrl_controlnet* cn, ctrl_adapter* ad, const void* sample, int sample_dtype, int N, int Hs, int Ws,
                                 const float* cn_timesteps, int cn_t_count, const void* cn_ehs, int cn_ehs_dtype, int cn_Lk,
                                 const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                                 void* const* cn_outs, int cn_out_dtype, int num_frames, const float* ad_timesteps,
                                 int ad_t_count, const void* ad_ehs, int ad_ehs_dtype, int ad_ehs_batch, int ad_Lk,
                                 int use_mid, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                                 void* stream) {
    CTRL_CHECK(cn && ad && cn_outs && outs, "step_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    const bool env_off = policy_int(P_STEP_OVERLAP, 1) == 0;
    const bool async = !g_prof_on && !env_off;      // per-launch profiling needs one kernel at a time
    hipEvent_t* out_ev = nullptr; hipEvent_t done = nullptr;
    TRY(controlnet_forward_async(cn, sample, sample_dtype, N, Hs, Ws, cn_timesteps, cn_t_count, cn_ehs, cn_ehs_dtype, cn_Lk,
                                 controlnet_cond, cond_dtype, conditioning_scale, flags, cn_outs, cn_out_dtype, s, async,
                                 &out_ev, &done));
    const void* ins[13];
    for (int i = 0; i < 13; ++i) ins[i] = cn_outs[i];
    if (!use_mid) ins[12] = nullptr;
    int rc = adapter_forward_events(ad, ins, cn_out_dtype, N, Hs, Ws, num_frames, ad_timesteps, ad_t_count, ad_ehs, ad_ehs_dtype,
                                    ad_ehs_batch, ad_Lk, outs, out_dtype, frame_pos, N_out, stream, out_ev);
    // the caller's stream must not run past this call before the whole ControlNet has finished (its outputs are
    // results of the call too, and its tail -- outputs no adapter block consumes -- may still be in flight)
    if (done) HIP_TRY(hipStreamWaitEvent(s, done, 0));
    return rc;
}
