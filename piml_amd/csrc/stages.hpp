// Launch stages of the fused PINNSF network, shared between the per-component C entries (encoder.hip, decoder.hip)
// and the network-level entries that fork them over HIP streams (network.hip).  Private to libpiml_hip.so.
// Every function enqueues on `s` only, validates its arguments and returns a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/piml_hip.h"

namespace piml {

int enc_stage_pack(const piml_encoder_branch* br, int nbr, hipStream_t s);
// false: no kernel the current settings dispatch reads the f32-instruction fragment images of `packed` (split products everywhere)
bool enc_f32_images_needed();
// packed image must be current; `zero` (optional): zero_n floats cleared by the launch
int enc_stage_fwd(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero = nullptr, long long zero_n = 0, bool msum = false);
// msum = PIML_POOL_MSGS: the agents' sums of the messages written by the forward (sum_a / sum_b), message rows only where a branch
// carries `msgs`; enc_pool_msgs_ok: the configuration that form serves
bool enc_pool_msgs_ok(const piml_encoder_branch* br, int nbr);
// PIML_POOL_H2 (inference): the forward up to layer 2 and the agents' sums of h2 into `msgs` / `h2`, both (agents, 128)
// (enc_fwd_pool_x3_kernel); enc_pool_h2_ok: the configuration it serves
bool enc_pool_h2_ok(const piml_encoder_branch* br, int nbr);
int enc_stage_fwd_pool(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero = nullptr, long long zero_n = 0);
// PIML_POOL_TRAIN: training on the agents' sums of h2 (enc_fwd_sum_x3_kernel / the SUMS form of the one-pass backward);
// enc_pool_train_ok: the configuration they serve
bool enc_pool_train_ok(const piml_encoder_branch* br, int nbr);
int enc_stage_fwd_sum(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero = nullptr, long long zero_n = 0);
int enc_stage_bwd_sum(const piml_encoder_branch* br, int nbr, hipStream_t s);        // slots: one DW2_PART1 slot per workgroup
int enc_stage_bwd_dx(const piml_encoder_branch* br, int nbr, hipStream_t s);
int enc_stage_bwd_dw(const piml_encoder_branch* br, int nbr, hipStream_t s);          // dW partials (after bwd_dx)
int enc_stage_reduce(const piml_encoder_branch* br, int nbr, hipStream_t s, bool accumulate = false, bool defer = false);      // defer: PIML_DEFER_SLOT_SUMS
// true when enc_stage_bwd_dw writes these branches' partials as layer-split slots (encoder_dw2.hip); then
// n0[i] / n1[i] = the layer-0 / layer-1 slots of branch i
bool enc_dw2_used(const piml_encoder_branch* br, int nbr, int* n0, int* n1);

int dec_stage_pack(const piml_decoder_branch* br, int nbr, hipStream_t s);
int dec_stage_pool(const piml_decoder_branch* br, int nbr, hipStream_t s);
// pooling + decoder tails + (head may be NULL) the collision head in one launch; after the packs
int dec_stage_fwd_fused(const piml_decoder_branch* br, int nbr, const piml_collision_head* h, const float* self_features,
                        float tau, float* acc, hipStream_t s);
int dec_stage_fwd_ph2(const piml_decoder_branch* br, int nbr, const float* self_features, float tau, float* acc,
                      hipStream_t s);      // PIML_POOL_H2: decoder tails on the agents' sums (`pooled` + second parts in `msgs`)
int dec_stage_fwd(const piml_decoder_branch* br, int nbr, const float* self_features, float tau, float* acc,
                  hipStream_t s);                                                     // after pack + pool
int dec_stage_bwd_dx(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features, float tau,
                     float* g_self, hipStream_t s);
// dX chain + weight-gradient partials (no slot sum) in one launch
int dec_stage_bwd_fused(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features, float tau,
                        float* g_self, hipStream_t s, bool sums = false);      // sums: PIML_POOL_TRAIN (folded first layer)
// PIML_POOL_TRAIN: decoder tails on the agents' sums of h2 (`pooled` + second parts in `msgs`) with the folded first layer, and
// the collision head (may be NULL) on the h2 rows with the folded W1, in one launch
// fold = false (PIML_POOL_MSGS): the sums are the agents' sums of the MESSAGES (plain first layers, the head on the message rows)
int dec_stage_fwd_sum(const piml_decoder_branch* br, int nbr, const piml_collision_head* h, const float* self_features, float tau,
                      float* acc, hipStream_t s, bool fold = true);
int dec_stage_bwd_dw(const piml_decoder_branch* br, int nbr, const float* g_pred, bool reduce, hipStream_t s);   // partials (+ slot sum)

// keep-masks of up to two row sets in ONE launch (one draw, streams[i] tells them apart); advances the draw counter
int dropout_stage(unsigned long long* state, const long long* rows, unsigned* const* bits, const unsigned* streams, int njobs,
                  int cols, float p, hipStream_t s);

int head_stage_pack(const piml_collision_head* h, hipStream_t s);
int head_stage_fwd(const piml_collision_head* h, hipStream_t s);                      // packed image must be current

}  // namespace piml
