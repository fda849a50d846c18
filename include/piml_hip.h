/*
 * piml_hip.h -- C ABI of libpiml_hip.so, the MI355X (gfx950) implementation of PIML's
 * per-timestep pairwise hot path.
 *
 * The reference (tsinghua-fib-lab/PIML) is pure Python/PyTorch and has no FFI layer; the
 * entry points below are what a binding for this path binds (ctypes stub: INTEGRATION.md).
 * Each entry cites the reference interface it replaces (paths relative to the reference
 * repository root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch allocates); float32,
 *    C-contiguous; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - nothing is allocated, freed or synchronised inside: every call only enqueues work on
 *    `stream`, so calls are re-entrant and capturable into a hipGraph;
 *  - return value is a hipError_t as int (0 = hipSuccess, 1 = hipErrorInvalidValue for a
 *    rejected argument); no exceptions cross the boundary;
 *  - absent agents are NaN positions (the reference's sentinel, src/data/data.py:141-143),
 *    never compacted;
 *  - leading dimensions (channels, time) are flattened by the caller into C "slices".
 *
 * Where to start.  A host that replaces the reference's per-timestep path binds FIFTEEN of the entries below (INTEGRATION.md
 * sections 2 - 4); everything else in this header is the same path cut into components (for hosts that keep parts of the
 * reference's torch.nn network, and for the parity tests), and measurement plumbing / A-B switches live in piml_hip_tuning.h:
 *   features      piml_relfeat_self_fwd / piml_relfeat_self_bwd   (get_relative_features + the self_features cat, both ways)
 *                 piml_heading_fwd                                 (multi-frame windows: the temporal heading fill)
 *   network       piml_pinnsf_pack / piml_pinnsf_fwd / piml_pinnsf_bwd   (PINNSF.forward and its autograd, three calls)
 *   closed form   piml_mlapm_step_fwd / piml_mlapm_step_bwd_ws    (MLAPM.step and its analytic gradient)
 *   collisions    piml_collision_counts / piml_collision_label     (collision_detection(...).sum(-1), calculate_collision_label)
 *   integrator    piml_rollout_step / piml_train_step_fwd / piml_train_step_bwd   (the frame bodies of the two rollout loops)
 *   multi-GPU     piml_allgather_state + piml_reducescatter_grad (RCCL), or piml_p2p_exchange (P2P stores, capturable)
 */
#ifndef PIML_HIP_H
#define PIML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIML_HIP_ABI_VERSION 32
#define PIML_MAX_TOPK 32 /* topk_ped / topk_obs upper bound (reference defaults 6 / 10) */

/* ABI version of the loaded library (== PIML_HIP_ABI_VERSION). */
int piml_abi_version(void);

/* hipGetErrorString for the codes returned below. */
const char* piml_error_string(int err);

/*
 * Heading direction.  Replaces Pedestrians.get_heading_direction
 * (src/data/data.py:350-395): zero-velocity frames are filled from the temporally nearest
 * non-zero frame (backward sweep then forward sweep), then h/|h| (|h| == 0 -> h/0.1).
 * velocity, heading: (C, T, N, 2).
 */
int piml_heading_fwd(const float* velocity, int C, int T, int N, float* heading, void* stream);

/*
 * Relative features, forward.  Replaces Pedestrians.get_relative_features
 * (src/data/data.py:466-512) = get_nearby_obj_in_sight (:416-447) + get_relative_quantity
 * (:397-414) + get_filtered_features (:449-464), for C slices of N agents and M shared
 * obstacle points, without materialising any N x N tensor.
 *
 *   position, velocity, acceleration : (C, N, .) per-agent records whose consecutive agents
 *             are `state_ld` floats apart: state_ld = 2 for three separate (C, N, 2) arrays,
 *             6 for one interleaved (C, N, 6) = (p, v, a) buffer (pass base, base+2, base+4),
 *             which is what the per-step all-gather of agent-block sharding produces
 *   destination : (C, focal_count, 2), destinations of the focal rows only
 *   heading : (C, N, 2) unit heading from piml_heading_fwd, or NULL = derive it from
 *             `velocity` (the T == 1 per-step case, where heading = v/|v|)
 *   obstacles : (M, 2), may be NULL when M == 0
 *   focal_begin, focal_count : the block of focal agents this call computes (multi-GPU
 *             agent-block sharding; single GPU: 0, N).  Sources are always all N agents.
 *   topk_ped/topk_obs : k (<= PIML_MAX_TOPK); effective k is min(k, N) / min(k, M)
 *   cos_thr_* : float32(cos(3.14 * sight_angle / 180))  (the reference's 3.14, quirk Q1)
 *   dist_thr_* : neighbours farther than this (strict >) are zero-padded
 * Outputs (row i = focal_begin + i):
 *   ped_feat (C, focal_count, kp_eff, 6) = (p_j - p_i, v_j - v_i, a_j - a_i), zero-padded
 *   obs_feat (C, focal_count, ko_eff, 6) = (o_j - p_i, -v_i, -a_i), zero-padded
 *   dest_feat (C, focal_count, 2) = destination - position, NaN -> 0; rows `dest_feat_ld` floats
 *             apart (2 = dense; 7 writes columns 0..1 of a (C, focal_count, 7) self_features buffer)
 *   ped_idx / obs_idx (int32, same leading shape, k_eff) : source index per slot, -1 = empty
 * NaN velocity / acceleration entries are read as 0 (the reference zeroes them in place
 * first, data.py:483-484; the host wrapper performs that in-place write).
 * Ordering rule: slots ascend by (distance, index); exact distance ties resolve to the
 * lower index (torch.sort leaves ties unspecified).
 */
int piml_relfeat_fwd(const float* position, const float* heading, const float* velocity,
                     const float* acceleration, int state_ld, const float* destination,
                     const float* obstacles, int C, int N, int M, int focal_begin,
                     int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                     float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                     float* ped_feat, float* obs_feat, float* dest_feat, int dest_feat_ld,
                     int32_t* ped_idx, int32_t* obs_idx, void* stream);

/* piml_relfeat_fwd + `*tick += 1` (a device-side int64 frame counter the kernel never reads): the captured inference
 * rollout frame (src/models/simulators.py:595-652) ends with this launch. */
int piml_relfeat_fwd_tick(const float* position, const float* heading, const float* velocity, const float* acceleration,
                          int state_ld, const float* destination, const float* obstacles, int C, int N, int M,
                          int focal_begin, int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                          float cos_thr_obs, float dist_thr_ped, float dist_thr_obs, float* ped_feat, float* obs_feat,
                          float* dest_feat, int dest_feat_ld, int32_t* ped_idx, int32_t* obs_idx, long long* tick,
                          void* stream);
/* piml_relfeat_fwd / piml_relfeat_bwd with the model's self_features rows [dest - p, v, a, v0] (C, n, 7) in place of the
 * destination features: the per-frame torch.cat of the training rollout (src/models/simulators.py:778-779) inside the launch.
 * desired_speed (C, n); g_state_zero (C, N, 6; may be NULL): cleared by the forward launch for the backward to accumulate into.
 * bwd: g_self (C, n, 7); ACCUMULATES into g_state (C, N, 6) = d/d(p, v, a) (cleared by the caller / by the forward);
 * writes g_destination (C, n, 2) and g_speed (C, n; may be NULL). */
int piml_relfeat_fwd_self(const float* position, const float* heading, const float* velocity, const float* acceleration,
                          int state_ld, const float* destination, const float* obstacles, const float* desired_speed, int C,
                          int N, int M, int focal_begin, int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                          float cos_thr_obs, float dist_thr_ped, float dist_thr_obs, float* ped_feat, float* obs_feat,
                          float* self_features, int32_t* ped_idx, int32_t* obs_idx, float* g_state_zero, void* stream);
int piml_relfeat_bwd_self(const float* g_ped_feat, const float* g_obs_feat, const float* g_self, const int32_t* ped_idx,
                          const int32_t* obs_idx, const float* position, int state_ld, const float* destination, int C, int N,
                          int focal_begin, int focal_count, int kp_eff, int ko_eff, float* g_state, float* g_destination,
                          float* g_speed, void* stream);


/*
 * piml_relfeat_fwd for ONE scene of packed (N, 6) = (p, v, a) records that also writes the model's self_features rows
 * [dest - p, v, a, v0] (n, 7) (src/models/simulators.py:169-173 builds them with a torch.cat per frame) and, when
 * `g_state_zero` (N * 6 floats) is given, clears it for the backward to accumulate into: one launch instead of three.
 * piml_relfeat_self_bwd is its backward in one launch: g_self (n, 7) = d/d(self_features); ACCUMULATES into the cleared
 * g_state (N, 6); writes g_destination (n, 2) and g_speed (n, may be NULL).
 */
int piml_relfeat_self_fwd(const float* state, const float* destination_rows, const float* obstacles,
                          const float* desired_speed, int N, int M, int focal_begin, int focal_count, int topk_ped,
                          int topk_obs, float cos_thr_ped, float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                          float* ped_feat, float* obs_feat, float* self_features, int32_t* ped_idx, int32_t* obs_idx,
                          float* g_state_zero, void* stream);
/* The same in two launches for agent-block sharding: PIML_RELFEAT_LOCAL reads only the focal block's own records (and the
 * obstacles), so it can run while the all-gather of the other blocks is in flight; it writes obs_feat, self_features,
 * obs_idx and leaves its pedestrian list in ped_idx.  PIML_RELFEAT_REMOTE continues from that list over the agents either
 * side of the block and writes ped_feat / ped_idx.  Bit-identical to the single launch (part 0). */
#define PIML_RELFEAT_LOCAL 1
#define PIML_RELFEAT_REMOTE 2
int piml_relfeat_self_fwd_part(int part, const float* state, const float* destination_rows, const float* obstacles,
                               const float* desired_speed, int N, int M, int focal_begin, int focal_count, int topk_ped,
                               int topk_obs, float cos_thr_ped, float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                               float* ped_feat, float* obs_feat, float* self_features, int32_t* ped_idx,
                               int32_t* obs_idx, float* g_state_zero, void* stream);
int piml_relfeat_self_bwd(const float* g_ped_feat, const float* g_obs_feat, const float* g_self, const int* ped_idx,
                          const int* obs_idx, const float* state, const float* destination_rows, int N, int focal_begin,
                          int focal_count, int kp_eff, int ko_eff, float* g_state, float* g_destination, float* g_speed,
                          void* stream);

/*
 * Relative features, backward: what autograd computes through gather / repeat / masked
 * zeroing in src/data/data.py:397-414, 449-464, 491-510.
 *   g_ped_feat (C, focal_count, kp_eff, 6), g_obs_feat (C, focal_count, ko_eff, 6),
 *   g_dest_feat (C, focal_count, 2) : upstream gradients
 *   ped_idx, obs_idx : as written by piml_relfeat_fwd
 *   position (stride state_ld), destination (focal rows) : forward inputs (only their NaN
 *             pattern is used)
 * Outputs:
 *   g_state (C, N, 6) : d/d(position, velocity, acceleration) concatenated per agent, for
 *             ALL N sources (a rank's partial sum under agent-block sharding); the kernel
 *             ACCUMULATES into it with float atomics, the caller zeroes it first;
 *   g_destination (C, focal_count, 2) : overwritten.
 */
int piml_relfeat_bwd(const float* g_ped_feat, const float* g_obs_feat, const float* g_dest_feat,
                     const int32_t* ped_idx, const int32_t* obs_idx, const float* position,
                     int state_ld, const float* destination, int C, int N, int focal_begin, int focal_count,
                     int kp_eff, int ko_eff, float* g_state, float* g_destination, void* stream);

/*
 * Deterministic piml_relfeat_bwd (same gradient, no atomics, bit-reproducible; slower -- opt-in): the caller supplies the
 * neighbour-list entries sorted by source: sorted_keys[e] = slice * N + ped_idx of entry e (C * N for an empty slot),
 * ascending, and order[e] = row * kp_eff + slot of that entry, in a STABLE order (C * focal_count * kp_eff int64 each).
 * accumulate != 0: g_state (C, N, 6) is added to instead of overwritten.  Every g_state element is written.
 */
int piml_relfeat_bwd_det(const float* g_ped_feat, const float* g_obs_feat, const float* g_dest_feat, const int* ped_idx,
                         const int* obs_idx, const long long* sorted_keys, const long long* order, const float* position,
                         int state_ld, const float* destination, int C, int N, int focal_begin, int focal_count,
                         int kp_eff, int ko_eff, int accumulate, float* g_state, float* g_destination, void* stream);

/*
 * Closed-form social-force step.  Replaces MLAPM.step (src/models/mlapm.py:10-58).
 *   position, velocity, destination (N, 2); desired_speed (N); variant 0 = 'raw', 1 = 'GC',
 *   2 = 'UCY' (with the one-line coll.unsqueeze(-1) fix the shipped code needs for N > 2);
 *   tau, A, B, C, D, theta_deg = self.args[...]; radius, dt = step() arguments.
 * Outputs: action (N, 2) = velocity + force * dt; force (N, 2) optional (NULL to skip).
 * skip_absent = 0: like the reference, a NaN position poisons every row (its callers filter absent
 * agents first, src/main_mlapm.py:19-25).  skip_absent = 1: agents with a NaN position are treated
 * as absent -- they exert no force and their own action is NaN -- which replaces that host-side
 * compaction and keeps shapes static (graph-capturable simulation loop).
 */
int piml_mlapm_step_fwd(const float* position, const float* velocity, const float* desired_speed,
                        const float* destination, int N, int variant, float tau, float A, float B,
                        float C, float D, float theta_deg, float radius, float dt, int skip_absent,
                        float* action, float* force, void* stream);

/*
 * One frame of the simulation loop of src/main_mlapm.py:18-36 in ONE launch: the state of frame t - 1 = *frame_counter - 1 is
 * read from the trajectories (frames, N, 2) themselves -- an agent within `radius` of its destination in a frame >= 1 is absent
 * (NaN) from the next frame on, absent agents contribute nothing (skip_absent) -- MLAPM.step's velocity and p + v dt are
 * written to frame t, and the last workgroup to finish sets *frame_counter = t + 1 (done_counter: a zeroed device word).
 * Frame 0 is the caller's initial state.  A launch with *frame_counter outside [1, frames) does nothing.
 */
int piml_mlapm_rollout_step(float* traj_position, float* traj_velocity, const float* desired_speed,
                            const float* destination, long long frames, int N, int variant, float tau, float A,
                            float B, float C, float D, float theta_deg, float radius, float dt,
                            long long* frame_counter, unsigned* done_counter, void* stream);

/*
 * Analytic gradient of piml_mlapm_step_fwd (what autograd gives through mlapm.py:10-58):
 * g_action (N, 2) -> g_position, g_velocity, g_destination (N, 2), g_desired_speed (N), all
 * overwritten.  view / rotation sign / collision flags are piecewise constant (no gradient).
 * No atomics: every pair is evaluated in both roles by the owning wavefront.
 */
int piml_mlapm_step_bwd(const float* g_action, const float* position, const float* velocity,
                        const float* desired_speed, const float* destination, int N, int variant,
                        float tau, float A, float B, float C, float D, float theta_deg, float radius,
                        float dt, float* g_position, float* g_velocity, float* g_desired_speed,
                        float* g_destination, void* stream);

/*
 * The same gradient with every ordered pair evaluated ONCE (N >= 512): a wavefront keeps 128 source agents in its lanes and
 * rotates 64 focal agents -- with their focal-side sums -- through them, the two sides leave as partial rows in `workspace`
 * (piml_mlapm_bwd_workspace_floats(N, variant) floats, 0 = this form does not apply) and a second launch adds an agent's
 * rows in a fixed order (UCY: the rotating pass gives every pair its flag-off term, the second launch -- a wavefront per
 * agent -- adds [flag on] - [flag off] for the pairs the exact predicate flags).  No atomics.  With workspace == NULL or a
 * small scene this IS piml_mlapm_step_bwd; a non-NULL workspace that is too small is hipErrorInvalidValue.
 */
long long piml_mlapm_bwd_workspace_floats(int N, int variant);
int piml_mlapm_step_bwd_ws(const float* g_action, const float* position, const float* velocity,
                           const float* desired_speed, const float* destination, int N, int variant,
                           float tau, float A, float B, float C, float D, float theta_deg, float radius,
                           float dt, float* g_position, float* g_velocity, float* g_desired_speed,
                           float* g_destination, float* workspace, long long workspace_floats, void* stream);

/*
 * Collision matrix, pair part of Pedestrians.collision_detection (src/data/data.py:549-564):
 * coll (S, N, N) = [|p_j - p_i| < threshold] (- I when minus_identity), NaN -> 0; S slices.
 * With minus_identity = 0 it is the `real_position` matrix of data.py:576-581.
 */
int piml_collision_matrix(const float* position, int S, int N, float threshold, int minus_identity,
                          float* coll, void* stream);

/*
 * "Friends" filter of collision_detection, in place on coll (C, T, N, N):
 *   base != NULL : 3-D rule (data.py:573-591), C must be 1: pairs whose `base` (S_base, N, N)
 *                  sum over slices exceeds 25 are zeroed in all T slices (base may be coll);
 *   base == NULL : 4-D rule (data.py:592-598): per channel, pairs colliding in any of the first
 *                  4 frames are zeroed in every frame.
 */
int piml_collision_friends(float* coll, const float* base, int C, int T, int S_base, int N, void* stream);

/*
 * Fused collision_detection(position (S,N,2), thr_h).sum(-1) for n_thresholds thresholds
 * (device array) with the 3-D friends rule, without materialising (S,N,N):
 * counts (n_thresholds, S, N), overwritten.  Callers: src/models/simulators.py:708-724,
 * src/functions/metrics.py:16-26.
 */
/*
 * Rollout losses of the fine-tuning step (src/models/simulators.py:172-249 as assembled at :790-819), forward and the
 * gradient fields in ONE launch:  p (C, T, N, 2) predicted positions; labels (C, T, N, labels_ld) with the label position in
 * columns 0-1; mask_pred (C, T, N) int64 (!= 0: the agent is predicted in that frame); gates (T) bytes (!= 0: the frame has
 * a predicted agent at all); collisions / hard_collisions (C, T, N) counts or NULL; abnormal_mask (N) or NULL.
 *   p_res = where(mask & gate, p, 0), lab = where(mask, labels[..., :2], 0), decay_t = time_decay^(T - 1 - t)
 *   out[0] = sum (p_res - lab)^2 decay_t                                       (multiple_rollout_mse_loss, reduction 'sum')
 *   out[1] = sum [sum_t collisions > 0] abnormal ((p_res - (p_res.n) n) - (lab - (lab.n) n))^2 decay_t,
 *            n = (lab[T-1] - lab[0]) / (|.| + 1e-6)                            (multiple_rollout_collision_loss, 'sum')
 *   out[2] = the same with hard_collisions
 * g_mse / g_coll / g_hard (C, T, N, 2) = d out[i] / d p.  piml_rollout_losses_bwd: g_p = sum_i *g_out_i * g_i, the three
 * upstream gradients as device scalars (NULL = 0).  The sums
 * run in a fixed order.  partial: 6 * piml_rollout_losses_blocks(C, N) floats, ticket: one zeroed unsigned (left zero);
 * both only used when piml_rollout_losses_blocks(C, N) > 1.
 */
/* n device-to-device copies (dst[i] <- src[i], bytes[i] bytes; the ranges of one pair do not overlap) in one launch per 24
 * pairs: the tensors of a training batch into the static inputs of the captured step (src/models/simulators.py:699-779). */
int piml_multi_copy(void* const* dst, const void* const* src, const size_t* bytes, int n, void* stream);
/* The prologue of the differentiable training rollout in one launch (src/models/simulators.py:672-697, :707): frame t_start of
 * position / velocity / acceleration / destination (C, T, N, 2) and dest_idx (C, T, N) int64 copied out, new_flag =
 * (long(mask_p - mask_p_pred) == 1) as bytes, mask_pred = long(mask_p_pred), gates[t] = (sum over (c, n) of mask_pred > 0) as
 * bytes and as floats, speed = self_features[..., t_start, :, 6], *nan_flag = 0. */
int piml_rollout_prologue(const float* position, const float* velocity, const float* acceleration, const float* destination,
                          const long long* dest_idx, const float* mask_p, const float* mask_p_pred, const float* self_features,
                          int C, int T, int N, int t_start, float* p0, float* v0, float* a0, float* d0, long long* di0,
                          unsigned char* new_flag, long long* mask_pred, unsigned char* gates, float* gates_f, float* speed,
                          int* nan_flag, void* stream);
int piml_rollout_losses_blocks(int C, int N);
int piml_rollout_losses(const float* p, const float* labels, long long labels_ld, const long long* mask_pred,
                        const unsigned char* gates, const float* collisions, const float* hard_collisions,
                        const float* abnormal_mask, int C, int T, int N, float time_decay, float* out, float* g_mse,
                        float* g_coll, float* g_hard, float* partial, unsigned* ticket, void* stream);
int piml_rollout_losses_bwd(const float* g_out0, const float* g_out1, const float* g_out2, const float* g_mse,
                            const float* g_coll, const float* g_hard, long long n, float* g_p, void* stream);
/* piml_rollout_losses on the collision count records of the frames AS PRODUCED, with the statistics the step logs and the
 * weighted total: count_frames = HOST array of T device pointers, frame t's (2, C, N) floats [collisions | hard collisions]
 * (piml_collision_counts of its positions; NULL = zeros), gated here by gates[t] (src/models/simulators.py:708-715, :728);
 * focus != 0: they also weight the two collision-focus sums (:800-819).  out (12 floats): [0] mse, [1] focus sum of the
 * collisions, [2] of the hard collisions, [3] sum of the gated collisions, [4] of the hard ones, [5] number of entries with
 * mask_pred == 1, [6] total = mse + w_coll [1] + w_hard [2], [7] w_coll [1], [8] w_hard [2].  T <= 32.
 * bwd: upstream gradients of [0], [7], [8] and [6] (device scalars; the first three may be NULL, the last not). */
int piml_rollout_losses_frames(const float* p, const float* labels, long long labels_ld, const long long* mask_pred,
                               const unsigned char* gates, const float* const* count_frames, int focus,
                               const float* abnormal_mask, int C, int T, int N, float time_decay, float w_coll, float w_hard,
                               float* out, float* g_mse, float* g_coll, float* g_hard, float* partial, unsigned* ticket,
                               void* stream);
int piml_rollout_losses_frames_bwd(const float* g_mse_out, const float* g_collw_out, const float* g_hardw_out,
                                   const float* g_total_out, float w_coll, float w_hard, const float* g_mse, const float* g_coll,
                                   const float* g_hard, long long n, float* g_p, void* stream);

/* The collision-prediction loss of `pinnsf_bm` over the frames of a training rollout (src/models/simulators.py:731-733 per frame,
 * :826-830 behind the loop) in ONE launch: pred_frames / feature_frames = HOST arrays of nframes (<= 32) device pointers, frame f's
 * predictions (n = windows * agents * k values in (0, 1), the model's last output) and pedestrian features (n rows of feature_ld >= 4
 * floats: relative position, relative velocity -- the label is calculate_collision_label of them, src/data/data.py:514-535); frame f
 * is rollout frame t_start + f of T_total, gates (T_total floats, 0 / 1) = the reference's per-frame `if torch.sum(mask) > 0`.
 *     out[0] = weight * F.binary_cross_entropy(pred * gate, label * gate, reduction='sum')   (logarithms clamped at -100)
 *     out[1] = sum(round(pred * gate) == label * gate) / (T_total * n)                         (frames outside the rollout: equal)
 * grad (nframes, n) = d out[0] / d pred (binary_cross_entropy_backward's (p - y) / max((1 - p) p, 1e-12)).  partial (2 floats per
 * workgroup of piml_collision_pred_loss_blocks) and a zeroed ticket are needed when that is > 1.  bwd: g_pred = *g_loss * grad
 * (g_loss NULL = 1), n = nframes * n values. */
int piml_collision_pred_loss_blocks(long long n, int nframes);
int piml_collision_pred_loss(const float* const* pred_frames, const float* const* feature_frames, int nframes, long long n, int k,
                             int feature_ld, const float* gates, int t_start, int T_total, float weight, float* out, float* grad,
                             float* partial, unsigned* ticket, void* stream);
int piml_collision_pred_loss_bwd(const float* g_loss, const float* grad, long long n, float* g_pred, void* stream);

/* The losses of a pointwise pre-training batch (src/models/simulators.py:333-352, pinnsf_interaction 'sim') and their gradients in ONE
 * launch: pred (rows, 2), labels (rows, labels_ld >= 6 [+ k]) -- columns 4, 5 the acceleration label, 6 .. 6 + k - 1 the collision labels
 * --, msgs (nmsg values, the model's second output) or NULL (reg_weight == 0), coll_pred (rows, k) (the model's last output,
 * `pinnsf_bm` under collision_pred_weight > 0) or NULL.
 *     out[1] = F.mse_loss(pred, labels[:, 4:6], 'sum')   out[2] = sum(reg_weight |msgs|)   out[3] = F.binary_cross_entropy(coll_pred,
 *     labels[:, 6:], 'sum')   out[0] = out[1] (+ out[2]) (+ out[3])
 * grad = [2 rows | nmsg | rows k] floats: d out[0] / d (pred | msgs | coll_pred); its backward is piml_collision_pred_loss_bwd (a copy
 * scaled by the upstream gradient).  partial (3 floats per workgroup of piml_pointwise_losses_blocks) and a zeroed ticket when > 1. */
/* One Adam step of n parameter tensors in ONE launch (the optimiser of both training loops: torch.optim.Adam(model.parameters(), lr,
 * weight_decay), src/models/simulators.py:69-71, :104-129, stepped at :319 / :358-359): params / grads / exp_avg / exp_avg_sq / steps =
 * HOST arrays of n device pointers (float32, contiguous; steps: one float counter per tensor, as torch keeps them with
 * capturable=True), sizes their element counts.  Per tensor: step += 1; grad += weight_decay * param; exp_avg = beta1 exp_avg +
 * (1 - beta1) grad; exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) grad^2; param -= (lr / (1 - beta1^step)) exp_avg / (sqrt(exp_avg_sq) /
 * sqrt(1 - beta2^step) + eps) -- torch's fused kernel operation for operation (doubles for the hyper-parameters), bitwise equal to it.
 * A tensor is cut into pieces of 1024 elements, one workgroup each; the tensor's last workgroup out (tickets) writes its counter. */
int piml_adam_tickets(void);   /* unsigned counters `tickets` must hold (zeroed once; every launch leaves them zero) */
int piml_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, float* const* steps,
                   const long long* sizes, int n, double lr, double beta1, double beta2, double weight_decay, double eps,
                   unsigned* tickets, void* stream);

int piml_pointwise_losses_blocks(long long rows, long long nmsg, int k);
int piml_pointwise_losses(const float* pred, const float* labels, long long labels_ld, long long rows, const float* msgs, long long nmsg,
                          float reg_weight, const float* coll_pred, int k, float* out, float* grad, float* partial, unsigned* ticket,
                          void* stream);


int piml_collision_counts(const float* position, int S, int N, const float* thresholds, int n_thresholds,
                          float* counts, void* stream);
/* The same for up to 32 frames of a training rollout in ONE launch (src/models/simulators.py:708-715 counts every frame on its
 * own): frames[f] = a HOST array of device pointers to (S, N, 2) tensors, S <= 25 slices each (independent slices, as in
 * piml_collision_counts below 26); counts (nframes, n_thresholds, S, N): record f is what a call on frame f alone writes. */
int piml_collision_counts_frames(const float* const* frames, int nframes, int S, int N, const float* thresholds,
                                 int n_thresholds, float* counts, void* stream);

/*
 * The same counts for stacks of MORE than 25 slices (the evaluation's (t, N, 2) rollouts) in parallel form: needs
 * `totals_zeroed`, n_thresholds * N * N int32 of caller scratch, zero on entry (it holds the per-pair collision totals
 * of the friends rule afterwards).  n_thresholds <= 4.  piml_collision_counts itself needs no scratch (one wavefront
 * per agent) and is the slower choice there.
 */
int piml_collision_counts_scratch(const float* position, int S, int N, const float* thresholds, int n_thresholds,
                                  int* totals_zeroed, float* counts, void* stream);
/* The same result from a per-frame cell grid (N <= 8192): one workgroup per slice bins its positions into cells of 1.02 x the
 * largest threshold and tests the 3 x 3 block around every agent -- O(N x occupancy) pair tests per slice instead of N^2;
 * two launches (totals, then counts), totals_zeroed as above. */
int piml_collision_counts_grid(const float* position, int S, int N, const float* thresholds, int n_thresholds,
                               int* totals_zeroed, float* counts, void* stream);


/*
 * Pedestrians.calculate_collision_label (src/data/data.py:514-535): rows of >= 4 floats
 * (dp, dv, ...), `row_stride` floats apart -> label[rows] in {0, 1}.
 */
int piml_collision_label(const float* ped_features, size_t rows, int row_stride, float* label, void* stream);

/*
 * utils.calc_acceleration (src/utils/utils.py:31-100): version 0/1/2 = 'v0'/'v1'/'v2' with the
 * caller-supplied constants (A, B, C, D, theta [rad]); rows of >= 2 floats -> acc (rows, 2).
 */
int piml_calc_acceleration(const float* relative_data, size_t rows, int row_stride, int version, float A,
                           float B, float C, float D, float theta, float eps, float* acc, void* stream);

/*
 * Fused per-frame epilogue of the inference rollout: the body of
 * BaseSimulator.get_multiple_rollouts between the model call and the feature recomputation
 * (src/models/simulators.py:596-639, 651) in one launch.  For every (slice, agent):
 *   record (p, v, a) and the presence mask of frame t; v' = v + a dt, p' = p + v dt (lagged
 *   Euler); waypoint switch when |p - dest| < 0.5; past the last waypoint the agent leaves the
 *   scene (p' = NaN, only if remove_arrived); agents entering at frame t+1 are re-initialised
 *   from the ground-truth series; history-velocity shift; columns 2.. of the next self_features
 *   row = (history, a', desired speed)  [columns 0..1 are written by piml_relfeat_fwd].
 * State (C,N,.) is updated IN PLACE; a_next (C,N,2) is the model's prediction.  Series are
 * (C,T,N,.); new_flag (C,T+1,N) uint8 (frame T all zero); F = hist_width + 5; waypoints
 * (D,N,2) shared (waypoints_per_slice = 0) or (C,D,N,2).  The frame index t is read from DEVICE
 * memory (frame_counter), so one captured launch serves every frame of a replayed hipGraph.
 */
int piml_rollout_step(float* position, float* velocity, float* acceleration, float* destination,
                      int64_t* dest_idx, float* hist_velocity, int hist_width, const float* a_next,
                      const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                      const float* position_series, const float* velocity_series,
                      const float* acceleration_series, const float* destination_series,
                      const int64_t* dest_idx_series, const float* self_features_series, int F,
                      const uint8_t* new_flag, float* position_out, float* velocity_out,
                      float* acceleration_out, float* mask_out, float* self_features_next,
                      const float* desired_speed, const int64_t* frame_counter, int C, int T, int N,
                      float dt, int remove_arrived, void* stream);

/* piml_rollout_step with the bottleneck variants' network epilogue in front of it (piml_pinnsf_epilogue_ksum_fwd's arithmetic: the sum
 * of the agent's kp / ko per-neighbour predictions pred_ped (C N kp, 2) [+ pred_obs (C N ko, 2) or NULL] + the desired-force
 * term of its self_features row, per-row norm): an inference frame of `pinnsf_bm` / `pinnsf_bottleneck` needs one launch for
 * both.  The self_features read are the rows of `self_features_next` BEFORE this launch rewrites them (F must be 7); a_next is
 * ignored (may be NULL). */
int piml_rollout_step_ksum(const float* pred_ped, int kp, const float* pred_obs, int ko, float tau, float* position,
                           float* velocity, float* acceleration, float* destination, int64_t* dest_idx, float* hist_velocity,
                           int hist_width, const float* a_next, const float* waypoints, int D, int waypoints_per_slice,
                           const int64_t* dest_num, const float* position_series, const float* velocity_series,
                           const float* acceleration_series, const float* destination_series,
                           const int64_t* dest_idx_series, const float* self_features_series, int F, const uint8_t* new_flag,
                           float* position_out, float* velocity_out, float* acceleration_out, float* mask_out,
                           float* self_features_next, const float* desired_speed, const int64_t* frame_counter, int C, int T,
                           int N, float dt, int remove_arrived, void* stream);

/*
 * The hand-written collision handling that closes PINNSF_polar_bottleneck_collision.forward
 * (`--model pinnsf_pbc`, src/models/model.py:1383-1444), on the k gathered neighbours of every
 * agent: rows of ped_features (row_stride floats: p_j - p_i, v_j - v_i, ...), the agent's velocity v_i
 * (self_features[..., 2:4]) and the predicted acceleration.  Reaction radius = collision_threshold +
 * 1.34 * 2 * time_unit; a neighbour inside it is "head-on" when (v_i . p_ji)(v_j . -p_ji) > 0, else "chasing";
 * for the nearest of each kind the approaching normal component of the prediction is removed and the
 * acceleration that cancels the normal closing speed within one time unit is added (step 2, then step 3 on
 * the result).  bwd is the analytic gradient w.r.t. predictions, velocity and the (p_ji, v_ji) columns of the
 * two selected neighbour rows (flags / selections are piecewise constant); any gradient output may be NULL.
 */
int piml_collision_correction_fwd(const float* predictions, const float* ped_features, const float* velocity,
                                  size_t rows, int k, int row_stride, float collision_threshold, float time_unit,
                                  float* out, void* stream);
int piml_collision_correction_bwd(const float* g_out, const float* predictions, const float* ped_features,
                                  const float* velocity, size_t rows, int k, int row_stride,
                                  float collision_threshold, float time_unit, float* g_predictions,
                                  float* g_ped_features, float* g_velocity, void* stream);

/*
 * Differentiable frame step of the fine-tuning rollout, BaseSimulator.test_multiple_rollouts_for_training
 * between the model call and the feature recomputation (src/models/simulators.py:741-769), one launch:
 *   v' = v + a dt, p' = p + v dt (lagged Euler), a' = a_pred; waypoint switch when |p - dest| < 0.5 (the
 *   index is clamped at the last waypoint: nobody leaves the scene here); agents flagged in
 *   new_flag[c, t_next, i] are re-initialised from the (C,T,N,.) ground-truth series of frame t_next
 *   (new_flag NULL or t_next >= T: no injection).  nan_flag (optional) is OR-ed with 1 when a_pred has a NaN
 *   (the assert at :745, kept on the device).  State (C,N,.) in, new state out; waypoints as in
 *   piml_rollout_step.  bwd: with keep = not re-initialised,
 *   g_p = keep g_p', g_v = keep (g_v' + dt g_p'), g_a = keep dt g_v', g_a_pred = keep g_a'
 *   (any g_*_out may be NULL = zero; any output may be NULL = not wanted).
 * zero_mask (C,N) uint8, optional: when given, NaN components of v' and a' are replaced by 0 -- what the
 * next Pedestrians.get_relative_features call would do to them in place (src/data/data.py:483-484) -- and
 * the mask records which (bits 0-1: v'.xy, bits 2-3: a'.xy); bwd passes no gradient through those.
 */
int piml_train_step_fwd(const float* position, const float* velocity, const float* acceleration,
                        const float* a_pred, const float* destination, const int64_t* dest_idx,
                        const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                        const uint8_t* new_flag, const float* position_series, const float* velocity_series,
                        const float* acceleration_series, const float* destination_series,
                        const int64_t* dest_idx_series, int C, int T, int N, int t_next, float dt,
                        float* position_out, float* velocity_out, float* acceleration_out,
                        float* destination_out, int64_t* dest_idx_out, int* nan_flag, uint8_t* zero_mask,
                        void* stream);
/* piml_train_step_fwd that also leaves the frame's INPUT position in a caller's buffer (position_copy: C slices of (N, 2)
 * floats, position_copy_slice_stride floats apart -- frame t of the (C, T, N, 2) array of predicted positions the rollout loss
 * reads, src/models/simulators.py:728, :790): the frames write that array themselves instead of a concatenation behind the loop.
 * position_copy may be NULL (= piml_train_step_fwd). */
int piml_train_step_fwd_copy(const float* position, const float* velocity, const float* acceleration, const float* a_pred,
                             const float* destination, const int64_t* dest_idx, const float* waypoints, int D,
                             int waypoints_per_slice, const int64_t* dest_num, const uint8_t* new_flag,
                             const float* position_series, const float* velocity_series, const float* acceleration_series,
                             const float* destination_series, const int64_t* dest_idx_series, int C, int T, int N, int t_next,
                             float dt, float* position_out, float* velocity_out, float* acceleration_out, float* destination_out,
                             int64_t* dest_idx_out, int* nan_flag, uint8_t* zero_mask, float* position_copy,
                             long long position_copy_slice_stride, void* stream);
int piml_train_step_bwd(const float* g_position_out, const float* g_velocity_out, const float* g_acceleration_out,
                        const uint8_t* new_flag, const uint8_t* zero_mask, int C, int T, int N, int t_next, float dt,
                        float* g_position, float* g_velocity, float* g_acceleration, float* g_a_pred, void* stream);
/* piml_train_step_bwd with a second source of d/d(outputs): g_state6 (C, N, 6; may be NULL) = d/d(p', v', a') interleaved, as
 * piml_relfeat_bwd_self leaves it, is ADDED to the three separate gradients (each may be NULL) before the frame's backward --
 * the two consumers of a frame's state (the next frame's step and its features) without a gradient accumulation in between. */
int piml_train_step_bwd6(const float* g_position_out, const float* g_velocity_out, const float* g_acceleration_out,
                         const float* g_state6, const unsigned char* new_flag, const unsigned char* zero_mask, int C, int T, int N,
                         int t_next, float dt, float* g_position, float* g_velocity, float* g_acceleration, float* g_a_pred,
                         void* stream);
/* piml_train_step_bwd6 with a third source: g_position_in (may be NULL) = a gradient that arrives on the frame's INPUT position
 * itself -- the rollout loss reads every frame's position (src/models/simulators.py:728, :790-819) and the next frame's step
 * reads it too; handed in here (C slices of (N, 2) contiguous floats, g_position_in_slice_stride floats apart: a time slice of
 * the loss's (C, T, N, 2) gradient as it stands) it is added to g_position inside the launch instead of by a strided addition
 * of the autograd engine per frame.  g_position_out_slice_stride != 0: g_position_out is laid out the same way (0: C * N
 * contiguous rows). */
int piml_train_step_bwd7(const float* g_position_out, long long g_position_out_slice_stride, const float* g_velocity_out,
                         const float* g_acceleration_out,
                         const float* g_state6, const float* g_position_in, long long g_position_in_slice_stride,
                         const unsigned char* new_flag, const unsigned char* zero_mask, int C, int T, int N, int t_next, float dt,
                         float* g_position, float* g_velocity, float* g_acceleration, float* g_a_pred, void* stream);
/* The frame step with the model's TAIL in front of it, one launch each way.  In the training rollout the network sees channelled
 * (C, N, .) input, where the reference's desired-force term takes its norm over the AGENT axis (quirk Q2: `torch.norm(self_features[...,
 * :2], p=2, dim=1, keepdim=True)`, src/models/model.py:1290 / :1125 as evaluated at src/models/simulators.py:701), so the tail
 *     predictions = sum_k acc_ped[c, n, k] (+ sum_k acc_obs[c, n, k]) + (v0 d / t - v) / tau,   t[c, comp] = || d[c, :, comp] ||_2 (0 -> 0.1)
 * is a reduction over a slice's agents followed by piml_train_step_fwd_copy on the result (a_pred = predictions).  acc_ped (C, N, kp, 2):
 * the decoder tails' sum (kp = 1) or the bottleneck variants' per-neighbour predictions; acc_obs likewise or NULL; self_features
 * (C, N, 7).  Backward = piml_train_step_bwd7, then with g = d/d(predictions) (written to g_prediction (C, N, 2), required):
 * g_acc_ped (C, N, kp, 2) / g_acc_obs = g broadcast over k (NULL: not written -- for kp = 1 g_prediction IS that gradient), g_self
 * (C, N, 7) = piml_pinnsf_epilogue_agentnorm_bwd's (may be NULL).  Bitwise what the separate launches produce. */
int piml_train_step_tail_fwd(const float* position, const float* velocity, const float* acceleration, const float* acc_ped, int kp,
                             const float* acc_obs, int ko, const float* self_features, float tau, const float* destination,
                             const int64_t* dest_idx, const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                             const uint8_t* new_flag, const float* position_series, const float* velocity_series,
                             const float* acceleration_series, const float* destination_series, const int64_t* dest_idx_series, int C,
                             int T, int N, int t_next, float dt, float* position_out, float* velocity_out, float* acceleration_out,
                             float* destination_out, int64_t* dest_idx_out, int* nan_flag, uint8_t* zero_mask, float* position_copy,
                             long long position_copy_slice_stride, void* stream);
int piml_train_step_tail_bwd(const float* g_position_out, long long g_position_out_slice_stride, const float* g_velocity_out,
                             const float* g_acceleration_out, const float* g_state6, const float* g_position_in,
                             long long g_position_in_slice_stride, const unsigned char* new_flag, const unsigned char* zero_mask, int C,
                             int T, int N, int t_next, float dt, float* g_position, float* g_velocity, float* g_acceleration,
                             float* g_prediction, const float* self_features, float tau, int kp, int ko, float* g_acc_ped,
                             float* g_acc_obs, float* g_self, void* stream);


/*
 * Glue of the PINNSF network around its (PyTorch-ROCm / rocBLAS) GEMMs -- SURVEY.md row a8:
 * "only the desired-force term and k-sum are candidates to fuse".  The GEMMs stay in torch.
 *
 * piml_pinnsf_epilogue_fwd/bwd: the tail of every PINNSF.forward,
 *   predictions = acc_ped + acc_obs + (v0 * dest/|dest| - v) / tau     (src/models/model.py:1289-1294)
 * on rows of self_features = [dest - p (2), v (2), a (2), v0 (1)] (7 floats, contiguous), with the
 * reference's guard |dest| == 0 -> |dest| + 0.1.  The per-row norm only (2-D input, or
 * fix_dest_norm): the channelled dim=1 quirk (Q2) stays a torch expression.  acc_obs may be NULL.
 * bwd: g_self (rows,7) from g_out (rows,2); d/d(acc_ped) = d/d(acc_obs) = g_out (no kernel).
 */
int piml_pinnsf_epilogue_fwd(const float* acc_ped, const float* acc_obs, const float* self_features,
                             size_t rows, float tau, float* out, void* stream);
int piml_pinnsf_epilogue_bwd(const float* g_out, const float* self_features, size_t rows, float tau,
                             float* g_self, void* stream);
/*
 * The same tail for the bottleneck variants (src/models/model.py:1116-1134): predictions = sum over the neighbour axis of
 * the per-row predictor outputs pred_ped (rows, kp, 2) [+ pred_obs (rows, ko, 2), may be NULL] + the desired-force term;
 * backward: g_self (may be NULL) and the upstream gradient broadcast to every neighbour row (either may be NULL).
 */
int piml_pinnsf_epilogue_ksum_fwd(const float* pred_ped, int kp, const float* pred_obs, int ko, const float* self_features,
                                  size_t rows, float tau, float* out, void* stream);
int piml_pinnsf_epilogue_ksum_bwd(const float* g_out, const float* self_features, size_t rows, float tau, int kp, int ko,
                                  float* g_self, float* g_pred_ped, float* g_pred_obs, void* stream);

/*
 * The same tail for channelled (C, N, 7) input with the reference's literal `dim=1` norm (quirk Q2): the
 * norm of each destination component is taken over the N AGENTS of slice c,
 * t[c, comp] = || self_features[c, :, comp] ||_2 (+0.1 where 0), as src/models/model.py:1290 computes it in
 * the fine-tuning rollouts (src/models/simulators.py:701).  One workgroup per slice.
 */
int piml_pinnsf_epilogue_agentnorm_fwd(const float* acc_ped, const float* acc_obs, const float* self_features,
                                       int C, int N, float tau, float* out, void* stream);
/* piml_pinnsf_epilogue_ksum_fwd with the agent-axis norm of piml_pinnsf_epilogue_agentnorm_fwd (quirk Q2): the bottleneck
 * variants' tail on the channelled (C, N, .) frames of the training rollout -- neighbour-axis sums + desired force in one launch.
 * Backward: piml_pinnsf_epilogue_ksum_bwd with g_self = NULL (the broadcasts) + piml_pinnsf_epilogue_agentnorm_bwd. */
int piml_pinnsf_epilogue_ksum_agentnorm_fwd(const float* pred_ped, int kp, const float* pred_obs, int ko, const float* self_features,
                                            int C, int N, float tau, float* out, void* stream);
int piml_pinnsf_epilogue_agentnorm_bwd(const float* g_out, const float* self_features, int C, int N, float tau,
                                       float* g_self, void* stream);

/*
 * self_features rows for the model from the packed state: out (rows,7) = [dest_feat (ld dest_ld),
 * state[:, 2:6] (v, a; interleaved (p,v,a) records, 6 floats), desired_speed] -- the torch.cat at
 * src/models/simulators.py:648-650 / 777-779.  dest_feat NULL: columns 0-1 of `out` are left as they
 * are (piml_relfeat_fwd wrote them in place with dest_feat_ld = 7).  bwd splits g_self into g_dest (rows,2),
 * g_state (rows,6; position columns zero) and g_speed (rows).
 */
int piml_self_features_fwd(const float* dest_feat, int dest_ld, const float* state, const float* desired_speed,
                           size_t rows, float* out, void* stream);
int piml_self_features_bwd(const float* g_self, size_t rows, float* g_dest, float* g_state, float* g_speed,
                           void* stream);

/*
 * Backward glue of one Linear(+ReLU) layer: g_pre = g * [y > 0] (y = the layer's output; NULL: no
 * activation, g_pre is not written) and bias gradient db[c] = sum_r g_pre[r,c], in ONE pass over
 * g (replaces torch's threshold_backward + column reduce_kernel + its semaphore memset).
 * g, y, g_pre: (rows, cols) row-major.  Deterministic two-level sum in a fixed order: slabs of rows ->
 * `partials` (piml_colsum_blocks(rows, cols) * cols floats, caller-owned scratch; may be NULL when
 * that count is 1), then one small second launch over the partials.  No atomics, no global state.
 * cols <= 1024 when cols % 4 == 0, else cols <= 256.
 */
int piml_colsum_blocks(size_t rows, int cols);
int piml_act_bwd_colsum(const float* g, const float* y, size_t rows, int cols, float* g_pre, float* partials,
                        float* db, void* stream);

/*
 * out[j] = sum_b parts[b, j] for parts (B, n) row-major, n % 4 == 0, in a fixed order: the reduction step of
 * the chunked weight-gradient formulation dW = sum_b G_b^T X_b (the per-chunk products are one strided-batched
 * library GEMM; see ops._MLPChain), replacing the library's split-K GEMM + its reduction kernel.
 */
int piml_sum_leading(const float* parts, int B, size_t n, float* out, void* stream);
/*
 * The two small reductions that close one layer's backward in ONE launch: piml_sum_leading on (parts, B, n,
 * out) and the second stage of the bias-gradient sum on the `col_partials` (nb, cols) that
 * piml_act_bwd_colsum_stage1 left behind (same contract as piml_act_bwd_colsum minus its second launch; with
 * piml_colsum_blocks(rows, cols) == 1 it has already written db and there is nothing left to do).
 * Either half may be absent (parts NULL / col_partials NULL or nb <= 1).
 */
int piml_act_bwd_colsum_stage1(const float* g, const float* y, size_t rows, int cols, float* g_pre, float* partials,
                               float* db, void* stream);
int piml_layer_reduce(const float* parts, int B, size_t n, float* out, const float* col_partials, int nb, int cols,
                      float* db, void* stream);

/*
 * Neighbour-axis sum of the PINNSF processor output (src/models/model.py:1279-1283 with quirk Q3:
 * processor(x) = scale * x, scale = 2 in eval mode): msgs = scale * e (rows,cols) and
 * pooled[r / k] = sum over the k rows of one agent of msgs.  bwd: g_e = scale * (g_pooled[r / k] +
 * g_msgs[r]) with g_msgs optional (NULL).  cols % 4 == 0.
 * col_partials (optional, piml_ksum_blocks(agents, cols) * cols floats, needs 256 % (cols / 4) == 0): per-block
 * column sums of g_e, i.e. the first stage of the bias gradient of the Linear that produced e (finish with
 * piml_layer_reduce) -- saves that layer's own pass over g_e.
 * bias (cols, optional): msgs = scale * (e + bias) -- the bias of the encoder's last Linear, when that layer was
 * run as a plain GEMM (its bias-epilogue variant is the slower kernel); the gradient w.r.t. that bias is still
 * the column sum of g_e, which the layer's own backward computes.
 * keep_bits (optional, piml_dropout_keep_bits layout): the processor's train-mode dropout, msgs = keep * scale * (..)
 * with scale = 2 / (1 - p) chosen by the caller; bwd: g_e = keep * scale * (..).
 */
int piml_scale_ksum_fwd(const float* e, const float* bias, size_t agents, int k, int cols, float scale,
                        const unsigned* keep_bits, float* msgs, float* pooled, void* stream);
int piml_ksum_blocks(size_t agents, int cols);
int piml_scale_ksum_bwd(const float* g_pooled, const float* g_msgs, size_t agents, int k, int cols, float scale,
                        const unsigned* keep_bits, float* g_e, float* col_partials, void* stream);

/*
 * Keep-mask of the PINNSF processor's dropout.  The reference's processor is Dropout_p(2 x) in train mode
 * (src/models/model.py:82-119 with quirk Q3; model.train() at src/models/simulators.py:311; --dropout 0.5 at
 * src/main.py:45): the fused kernels take the mask as bits -- keep_bits (rows, ceil(cols / 32)) dwords, bit (c & 31)
 * of word (c >> 5) of a row = feature c of that row is kept -- and the caller folds 1 / (1 - p) into `scale`.
 *   state: 4 x uint64 on the device: [seed, offset, ticket, unused].  Every call draws with draw number `offset` and
 *          advances `offset` by one ON THE DEVICE (last block out), so a call captured into a hipGraph draws a fresh
 *          mask on every replay.  Philox4x32-10, counter = (offset lo, offset hi, row, (stream_id << 16) | sub),
 *          key = (seed lo, seed hi).  p == 0.5: keep word w of a row = output word w & 3 of the call
 *          sub = 0xFFFF - (w >> 2) (one call per 128 features).  Other p: feature c takes 16 bits -- call sub = c >> 3,
 *          output word (c >> 1) & 3, half c & 1 -- and is kept iff they are >= round(p * 65536).
 *   stream_id (0 .. 65535) tells masks of the same draw apart (the two branches of one encoder launch use 0 and 1).
 *   p in [0, 1]; p = 1 keeps nothing.
 */
int piml_dropout_keep_bits(unsigned long long* state, long long rows, int cols, float p, int stream_id, unsigned* keep_bits,
                           void* stream);

/*
 * Fused PINNSF encoder on the matrix cores (piml_amd/csrc/encoder_x3.hip: f32 products as split bf16 products, the
 * default; piml_amd/csrc/encoder.hip: the f32 matrix instruction; see piml_encoder_products): the reference's
 *   ped_encoder / obs_encoder = MLP(in, [128, 128, 128]) (src/models/model.py:40-65, built at :1232-1236),
 *   the processor in its effective form `scale * x` (ResDNN with >= 2 "layers" and inactive dropout, :82-119,
 *   SURVEY quirk Q3) and the neighbour-axis sum (:1279-1283),
 * i.e. msgs = scale * (W3 relu(W2 relu(W1 x + b1) + b2) + b3) for every neighbour row and pooled = sum over the k
 * rows of an agent.  Weights use the nn.Linear layout (out, in), row-major.  Up to two encoders (pedestrian and
 * obstacle branch) run in ONE launch, workgroups split in proportion to their rows.
 *
 * fwd:  x (rows, in_dim <= 8) -> msgs (rows, 128); h1 / h2 (rows, 128): post-ReLU activations of layers 1 / 2,
 *       saved for the backward (NULL for inference).  pooled comes from piml_encoder_ksum.
 *       h1 may be NULL in forward AND backward (every branch) when the backward does without it: relu_mask given, split
 *       products, more than piml_encoder_split_tiles() tiles (the dX chain then masks with the sign bits), layer-split
 *       weight gradients (piml_encoder_dw2, which recompute h1 from x on the matrix cores) and at least two workgroups per
 *       branch (piml_encoder_workgroups) -- 33 MB less to write and 33 MB less to read at the 4096-agent scene.
 * bwd:  upstream g_pooled (rows / k, 128) and / or g_msgs (rows, 128) (one may be NULL)
 *       -> g2, g1 (rows, 128): caller-provided scratch, the gradients at the pre-activations of layers 2 and 1;
 *          g_x (rows, in_dim) or NULL when the inputs need no gradient;
 *          partials: piml_encoder_workgroups() slots of piml_encoder_partial_floats() floats for THIS branch, slot p
 *          = [dW3 128x128 | dW2 128x128 | dW1 128 x in_dim (row-major, in a 1024-float field) | db3 | db2 | db1] of workgroup p's
 *          row slab; piml_encoder_bwd sums them into `grads` itself (one extra launch for both branches).
 */
typedef struct piml_encoder_branch {
    const float* x;
    long long rows;
    int in_dim;
    int k;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float scale;
    float *h1, *h2, *msgs;
    const float *g_pooled, *g_msgs;
    float *g2, *g1, *g_x;
    float* partials;
    float* grads;  /* bwd out: piml_encoder_partial_floats() floats = the slots of `partials` summed (same layout) */
    float* packed; /* piml_encoder_pack_floats() floats of caller-provided scratch: the weights re-ordered into MFMA
                      operand fragments; written by piml_encoder_fwd (or piml_encoder_pack), read by piml_encoder_bwd */
    unsigned* relu_mask; /* optional scratch, 256 dwords per 32-row tile (ceil(rows / 32) tiles): the signs of h1 and h2 as
                      bits.  A forward on the split-product kernels with more than piml_encoder_split_tiles() tiles writes
                      it, and the backward of the SAME configuration then reads these 2 MB instead of h1 and h2 (2 x 33 MB at
                      the 4096-agent scene) in its dX chain; pass the same pointer to both, or NULL to both */
    unsigned* keep_bits; /* optional (rows, 4) dwords, piml_dropout_keep_bits layout: the processor's train-mode
                      dropout.  msgs = keep * scale * (...) in the forward, g3 = keep * scale * (g_pooled + g_msgs) in the
                      backward (dX chain and dW3 / db3); the caller passes scale = 2 / (1 - p) and the same bits to both */
    unsigned long long* drop_state; /* forward only, optional: a piml_dropout_keep_bits state.  Non-NULL: the forward DRAWS the
                      mask of this launch (stream_id = index of the branch) with probability drop_p and leaves it in keep_bits
                      for the backward: for drop_p == 0.5 on the split-product kernels inside the forward kernel itself (one
                      Philox call per row, no extra launch), otherwise by one generator launch for all branches in front of it.
                      The draw counter advances once per launch.  Same state for every branch.  NULL: keep_bits is given */
    float drop_p;
    /* PIML_POOL_TRAIN (piml_pinnsf_fwd / bwd): the branch on the agents' SUMS of h2 instead of on message rows.
     * fwd out: sum_a / sum_b (agents, 128) = the agents' sums of h2 over their k rows, in two parts: sum_a from the 32-row tile
     * the agent's first row lies in, sum_b from the next tile (only for agents whose rows straddle two tiles:
     * (a k) >> 5 != (a k + k - 1) >> 5; undefined for the others).  `relu_mask` is required and holds the signs of h2 in the
     * layout of the exchanged layer (dwords 128 .. 255 of a tile: lane l = (feature j, half h) owns dwords 128 + 2 l, + 1;
     * bit 16 (blk & 1) + r of dword blk >> 1 = h2[row rho(r) + 4 h][32 blk + j] > 0).  `h2` (rows, 128) is written when
     * non-NULL (the collision head reads the rows), `msgs` is not written.  bwd in: g_pooled = d/d(sums) (agents, 128);
     * dW3 / db3 are not produced here (they follow from the decoder's folded first layer: piml_pinnsf_bwd). */
    float *sum_a, *sum_b;
} piml_encoder_branch;

/* floats of one partial slot / of one `packed` buffer */
int piml_encoder_partial_floats(void);
int piml_encoder_pack_floats(void);
/* (re)fill `packed` from the weights; piml_encoder_fwd does this itself, piml_encoder_bwd expects it done */
int piml_encoder_pack(const piml_encoder_branch* branches, int nbranches, void* stream);
/* total workgroups of a launch over these branches; *wg_branch0 = how many of them serve branch 0 (the rest serve
 * branch 1): the number of partial slots each branch's `partials` must hold. */
int piml_encoder_workgroups(const piml_encoder_branch* branches, int nbranches, int* wg_branch0);
int piml_encoder_fwd(const piml_encoder_branch* branches, int nbranches, void* stream);
/* the forward alone: `packed` already holds the images of these weights (piml_encoder_pack / piml_pinnsf_pack), e.g. one
 * pack per rollout or per back-propagated window instead of one per frame */
int piml_encoder_fwd_packed(const piml_encoder_branch* branches, int nbranches, void* stream);
int piml_encoder_bwd(const piml_encoder_branch* branches, int nbranches, void* stream);
/* accumulate = 1 (or PIML_ACCUMULATE): `grads` += the slot sums instead of = (a further backward pass through the same weights
 * within one optimiser step -- the frames of a training rollout; cf. PIML_ACCUMULATE of piml_pinnsf_bwd).  | PIML_DEFER_SLOT_SUMS:
 * the slot sums are not launched here but left for the next piml_relfeat_self_bwd on the stream (or piml_pinnsf_slot_sums_flush),
 * which runs them as leading workgroups of its launch -- together with those a piml_rowdecoder_bwd_acc of the same backward pass
 * left (the bottleneck variants: three launches become one); `partials` and `grads` must stay alive until then. */
int piml_encoder_bwd_acc(const piml_encoder_branch* branches, int nbranches, int accumulate, void* stream);
/* pooled (agents, 128) = sum over k consecutive rows of msgs (agents * k, 128) */
int piml_encoder_ksum(const float* msgs, long long agents, int k, float* pooled, void* stream);

/*
 * Fused PINNSF decoder tail on the f32 matrix cores (piml_amd/csrc/decoder.hip), reference
 * src/models/model.py:1283-1294 (`pinnsf_m`, same in `pinnsf`):
 *   pooled = sum over the k neighbour rows of msgs;  acc_branch = predictor(decoder(pooled)) with
 *   decoder = Linear(128, 64) ReLU Linear(64, 64), predictor = Linear(64, 2);
 *   acc = acc_ped [+ acc_obs] [+ (v0 * dest / |dest| - v) / tau when self_features (agents, 7) is given].
 * Both branches have the same number of agents.  Weights: nn.Linear layouts w1 (64,128) b1 (64) w2 (64,64) b2 (64)
 * w3 (2,64) b3 (2).  fwd writes pooled (agents,128), h1 (agents,64: post-ReLU), d2 (agents,64: decoder output) when
 * the pointers are given (needed by bwd) and `packed` (piml_decoder_pack_floats() floats, re-used by bwd).
 * bwd: g_pred (agents,2) -> g_pooled (agents,128) per branch; g_pre2 / g_pre1 (agents,64) are caller scratch;
 * g_self (agents,7, optional) = gradient of the desired-force term; partials: piml_decoder_workgroups(agents) slots of
 * piml_decoder_partial_floats() floats per branch = [dW1 64x128 | dW2 64x64 | dW3 2x64 | db1 64 | db2 64 | db3 2 +
 * 6 pad]; piml_decoder_bwd sums the slots into `grads` (same layout) itself.
 */
typedef struct piml_decoder_branch {
    const float* msgs;
    long long agents;
    int k;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *pooled, *h1, *d2;
    float *g_pre2, *g_pre1, *g_pooled;
    float* partials;
    float* grads; /* bwd out: piml_decoder_partial_floats() floats, the partial slots summed */
    float* packed;
    /* per-ROW use of the same network (the bottleneck variants, piml_rowdecoder_*): */
    float* pred;              /* fwd out (rows, 2): predictor output of every row */
    const float* g_pred_rows; /* bwd in (rows, 2) */
    const float* g_d2;        /* bwd in (rows, 64) or NULL: extra gradient on the decoder output (the `decoded` collision head) */
    /* PIML_POOL_TRAIN: the encoder's last layer folded into this decoder's first (msgs = scale (W3 h2 + b3) is linear in h2, so
     * W_d1 sum_r msgs_r + b_d1 = (scale W_d1 W3) sum_r h2_r + (b_d1 + k scale W_d1 b3)).  Non-NULL fold_w3 (128, 128) / fold_b3
     * (128) = that encoder's w3 / b3, fold_scale = its processor scale: the pack ALSO writes the folded images (float64
     * products, rounded once) behind the plain ones in `packed`; forward and backward use them under PIML_POOL_TRAIN.
     * `w1` / `b1` stay the raw decoder weights.  Then pooled = the agents' sums of h2 (first parts in, completed sums out),
     * msgs = their second parts (agents, 128), g_pooled = d/d(sums); in `grads` the dW1 / db1 fields hold the gradient of the
     * FOLDED layer until the slot sums' epilogue has unfolded them (dw1_out below). */
    const float *fold_w3, *fold_b3;
    float fold_scale;
    float* dw1_out; /* PIML_POOL_TRAIN bwd out (64, 128): d/d(w1) of the raw first layer (`grads`' dW1 field keeps d/d(W1')) */
} piml_decoder_branch;

int piml_decoder_pack_floats(void);
int piml_decoder_partial_floats(void);
int piml_decoder_workgroups(long long agents);
int piml_decoder_fwd(const piml_decoder_branch* branches, int nbranches, const float* self_features, float tau,
                     float* acc, void* stream);
int piml_decoder_bwd(const piml_decoder_branch* branches, int nbranches, const float* g_pred,
                     const float* self_features, float tau, float* g_self, void* stream);

/*
 * Collision head of `pinnsf_m` (src/models/model.py:1246, 1296-1300): out[row] = sigmoid(w2 . relu(W1 msgs[row] + b1)
 * + b2), msgs (rows, 128), W1 (64, 128), w2 (1, 64); forward only.  `packed`: piml_collision_head_pack_floats()
 * floats of scratch.  The 128 -> 64 layer runs as f32 arithmetic on split bf16 products like the encoder layers
 * (piml_amd/csrc/decoder.hip: head_fwd_body_x3; environment PIML_HEAD_PRODUCTS=f32 at load time: the f32 matrix
 * instruction); both stay within 1e-6 of float64 (tests/test_encoder_gpu.py).
 */
/*
 * The bottleneck variants (`pinnsf_bottleneck`, `pinnsf_bm`, src/models/model.py:1062-1221) apply decoder + predictor to
 * every NEIGHBOUR row and sum the (rows, 2) outputs over the neighbour axis afterwards (:1116-1122).  Same kernels, the
 * rows in the role of the agents: `msgs` = the (rows, 128) embeddings (read directly, no pooling; `pooled` unused),
 * `agents` = rows of the branch (the two branches may differ), `pred` / `h1` / `d2` per row; backward: `g_pred_rows`
 * (+ `g_d2`), `g_pooled` = d/d(embeddings) (rows, 128), weight-gradient partials in piml_rowdecoder_slots(rows) slots per
 * branch (slabs of a multiple of 32 rows), summed into `grads`.  One launch forward, three backward (dX, dW, slot sum).
 */
int piml_rowdecoder_slots(long long rows);
int piml_rowdecoder_fwd(const piml_decoder_branch* branches, int nbranches, void* stream);
int piml_rowdecoder_fwd_packed(const piml_decoder_branch* branches, int nbranches, void* stream); /* without the pack launch */
int piml_rowdecoder_bwd(const piml_decoder_branch* branches, int nbranches, void* stream);
int piml_rowdecoder_bwd_acc(const piml_decoder_branch* branches, int nbranches, int accumulate, void* stream);   /* grads +=, see piml_encoder_bwd_acc */

typedef struct piml_collision_head {
    const float* msgs; /* (rows, 128) */
    long long rows;
    const float *w1, *b1, *w2, *b2;
    float* packed; /* piml_collision_head_pack_floats() floats */
    float* out;    /* (rows) */
    /* PIML_POOL_TRAIN: the head on h2 rows (msgs = that array) with the encoder's last layer folded into W1:
     * W1' = fold_scale W1 fold_w3, b1' = b1 + fold_scale W1 fold_b3 (packed behind the plain images when fold_w3 is given) */
    const float *fold_w3, *fold_b3;
    float fold_scale;
} piml_collision_head;

int piml_collision_head_pack_floats(void);
int piml_collision_head_fwd(const float* msgs, long long rows, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* packed, float* out, void* stream);

/*
 * Collision head of `pinnsf_bm`, forward and backward (piml_amd/csrc/head64.hip).  Reference: src/models/model.py:1183
 * `ped_collision_predictor = MLP(64, [64, 1])`, :1214-1215 `sigmoid(...)` on the decoder output of every pedestrian
 * neighbour row; trained with BCE against the 1-s collision labels (src/models/simulators.py:348-355).
 *   out[row] = sigmoid(w2 . relu(W1 x[row] + b1) + b2),   x (rows, 64), W1 (64, 64), w2 (1, 64), nn.Linear layouts.
 * fwd: x -> out (rows), hidden (rows, 64: post-ReLU, saved for bwd; NULL for inference).
 * bwd: g_out (rows), out, hidden, x -> g_x (rows, 64; NULL = not wanted); partials = piml_head64_slots(rows) slots of
 *      piml_head64_partial_floats() floats of scratch; grads (same layout as one slot) = [dW1 64x64 | db1 64 | dw2 64 |
 *      db2 1 | 3 pad].  Two launches (tiles + slot sum), no atomics.  One slot per workgroup: four 32-row tiles per workgroup above
 *      2048 tiles, below that ONE tile per workgroup with its four waves on disjoint blocks of the tile's products (the training
 *      loops' few thousand rows: 14.6 -> 6 us per launch).
 */
typedef struct piml_head64 {
    const float* x;
    long long rows;
    const float *w1, *b1, *w2, *b2;
    float* hidden;
    float* out;
    const float* g_out;
    float* g_x;
    float* partials;
    float* grads;
} piml_head64;
int piml_head64_partial_floats(void);
int piml_head64_slots(long long rows);
int piml_head64_fwd(const piml_head64* head, void* stream);
int piml_head64_bwd(const piml_head64* head, void* stream);
int piml_head64_bwd_acc(const piml_head64* head, int accumulate, void* stream);   /* grads +=, | PIML_DEFER_SLOT_SUMS: see piml_encoder_bwd_acc */

/*
 * The corrector of `pinnsf_res` (src/models/model.py:1016-1020, :1050-1052; attn_pooling :950-970; ResDNN :82-119) on
 * hand-written kernels (piml_amd/csrc/corrector.hip), forward and backward:
 *     r = keep * scale * enc,  hid = relu(Wa r + ba),  s = wb . hid + bb,  attn = softmax_k(exp(s)),
 *     pooled = sum_k attn r,   out = Wd relu(Wc pooled + bc) + bd
 * enc (agents * k, 128): the pedestrian encoder's raw output, an agent's k neighbour rows consecutive; scale / keep_bits:
 * what corrector[0] (a ResDNN of >= 2 "layers" = Dropout(2 x)) does to it -- scale 2 in eval mode, 2 / (1 - p) and the
 * keep-mask bits (rows, 4) int32 of piml_dropout_keep_bits in train mode (NULL: keep everything); weights in nn.Linear
 * layouts: wa (128, 128), ba (128), wb (1, 128), bb (1), wc (64, 128), bc (64), wd (2, 64), bd (2).
 * fwd writes hid (rows, 128; NULL for inference), score / attn (rows), pooled (agents, 128), chid (agents, 64), out (agents, 2).
 * bwd reads them (not `out`, which may be NULL there) and g_out (agents, 2); writes g_enc (rows, 128; NULL = not wanted); scratch g_pooled (agents, 128),
 *     g_score (rows), g_chid (agents, 64), partials_a = piml_corrector_slots(0, ..) x piml_corrector_partial_floats(0) floats, partials_b likewise
 *     with 1; grads = [dWa 128x128 | dba 128 | dwb 128 | dbb 1 | 3 pad | dWc 64x128 | dbc 64 | dWd 2x64 | dbd 2 | 2 pad];
 *     accumulate != 0: grads += (see piml_encoder_bwd_acc).  No atomics: bit-reproducible.
 */
typedef struct piml_corrector {
    long long agents;
    int k;
    float scale;
    const float* enc;
    const unsigned* keep_bits;
    const float *wa, *ba, *wb, *bb, *wc, *bc, *wd, *bd;
    float *hid, *score, *attn, *pooled, *chid, *out;
    const float* g_out;
    float *g_pooled, *g_score, *g_chid, *g_enc;
    float *partials_a, *partials_b, *grads;
} piml_corrector;
int piml_corrector_partial_floats(int which);
int piml_corrector_slots(int which, long long agents, int k);
int piml_corrector_fwd(const piml_corrector* c, void* stream);
int piml_corrector_bwd(const piml_corrector* c, int accumulate, void* stream);

/*
 * The whole non-bottleneck PINNSF network (src/models/model.py:1271-1305: both encoders, the decoder tails, the
 * desired-force epilogue and, for `pinnsf_m`, the collision head) as ONE call per direction.  Default: the stages run in
 * program order on `stream` -- what a captured HIP graph wants:
 *
 *   piml_pinnsf_pack   ONE launch for the MFMA operand images of every weight (encoder x 2, decoder x 2, head) -> the
 *                      `packed` fields.  Weights that did not change since the last pack (rollouts, evaluation, the frames
 *                      of one back-propagated window) need no repack: pass PIML_PACKED_VALID to fwd.  flags: reserved (0).
 *   piml_pinnsf_fwd    [pack unless PIML_PACKED_VALID] -> encoders (both branches, one launch; draws the dropout masks
 *                      itself when the branches carry drop_state) -> [neighbour-axis sums + decoder tails + desired force +
 *                      collision head] (one launch).  head may be NULL.
 *   piml_pinnsf_bwd    [decoder dX chain + decoder dW partials] -> encoder dX -> encoder dW -> ONE slot sum for all four
 *                      partial sets.  Same fields as piml_decoder_bwd / piml_encoder_bwd.
 * flags: PIML_PACKED_VALID (fwd: skip the pack); PIML_FORK (opt-in: the independent stages on library-owned side streams
 * with event fork / join -- packs and collision head beside the encoders, decoder dW beside the encoder chain.  Measured
 * SLOWER inside captured graphs on ROCm 7.2, where every cross-stream edge of a replayed graph costs more than the ~5 us
 * stage it hides: 0.247 ms/step serial vs 0.295 forked at cfg3, DESIGN.md.  The side streams are created per device on
 * first use, outside any capture: piml_pinnsf_streams_init, idempotent).
 */
/* piml_pinnsf_unfold_defer(1, NULL): from now on (this device) a PIML_POOL_TRAIN backward does not launch the unfold of its folded
 * layers' gradients behind its slot sums but leaves it with the library; piml_pinnsf_unfold_defer(0, stream) launches what is
 * waiting (on `stream`, or on the stream it was left on when NULL) and ends the deferral.  For the backward passes of ONE
 * optimiser step that accumulate into the same buffers (PIML_ACCUMULATE: the frames of the rollout of
 * src/models/simulators.py:699-779): the unfold reads the folded layers' summed gradients and overwrites the unfolded ones, so
 * only the last pass's matters -- one launch per step instead of one per pass.  Another network's backward in between launches
 * what is waiting first.  Until the closing call dw1_out and the dW3 / db3 fields of the encoders' `grads` are NOT valid. */
int piml_pinnsf_unfold_defer(int on, void* stream);
#define PIML_PACKED_VALID 1
#define PIML_FORK 2
#define PIML_ACCUMULATE 4 /* piml_pinnsf_bwd: every branch's `grads` += the slot sums instead of = (a further backward pass through
                             the same weights inside one optimiser step: the frames of a training rollout); not with PIML_FORK */
#define PIML_DEFER_SLOT_SUMS 8 /* piml_pinnsf_bwd: do NOT launch the slot sums; leave them with the library, which runs them as the
                                  leading workgroups of the next piml_relfeat_self_bwd on the same stream (the two kernels are
                                  independent and small: one launch boundary, ~5 us, less per step), or -- whichever comes first
                                  -- at piml_pinnsf_slot_sums_flush / at the next deferring piml_pinnsf_bwd.  Until then the
                                  branches' `grads` are NOT valid: for callers that know a relfeat backward follows (the step of
                                  src/models/simulators.py:699-779: network backward, then the features' backward); not with PIML_FORK */
#define PIML_DEFER_PACK 16 /* piml_pinnsf_pack: do NOT launch; the next relfeat FORWARD launch on the same stream (piml_relfeat_self_fwd,
                              piml_relfeat_self_fwd_part, piml_relfeat_fwd_self) runs the pack as its trailing workgroups -- the
                              pack depends on the weights only and is a ~4 us launch in front of the step's chain by itself.
                              Whichever comes first: every consumer of packed images (piml_pinnsf_fwd, piml_encoder_fwd_packed,
                              piml_rowdecoder_fwd_packed) and piml_pinnsf_pack_flush launch a pack that is still waiting */
#define PIML_POOL_H2 32 /* piml_pinnsf_fwd, INFERENCE only (no head, no dropout, no backward): msgs = scale (W3 h2 + b3) is linear in
                           h2, so the neighbour-axis sum is taken BEFORE the last encoder layer.  The encoder launch stops after layer 2
                           and writes the agents' sums of h2: enc[i].msgs (agents, 128) = the part from the 32-row tile the agent's
                           first row lies in, enc[i].h2 (agents, 128) = the part from the next tile (agents whose k rows straddle two
                           tiles: (a k) >> 5 != (a k + k - 1) >> 5; undefined for the others).  dec[i].pooled = the same buffer as
                           enc[i].msgs, dec[i].msgs = the same as enc[i].h2; the decoder tails run on their sum.  The CALLER folds
                           the skipped layer into the decoder's first layer -- dec[i].w1 = scale * W_d1 * W3 (64 x 128),
                           dec[i].b1 = b_d1 + scale * k * W_d1 * b3 -- when it packs; enc[i].w3 / b3 / scale are not read.
                           Served when piml_pinnsf_pool_h2_ok(enc, nbranches): k = 6 or 10, split products, more than 32 tiles of
                           32 rows (environment PIML_POOL_H2_MIN_TILES); hipErrorInvalidValue otherwise */
#define PIML_POOL_TRAIN 64 /* piml_pinnsf_fwd AND piml_pinnsf_bwd (the same call pair): TRAINING on the agents' sums of h2 -- for
                           processors without an active dropout mask (eval mode, or p = 0) and callers that read neither branch's
                           per-row messages (the reference's training loops read predictions[0] only unless reg_weight > 0:
                           src/models/simulators.py:331-347, :702-737).  The messages are linear in h2, so the neighbour-axis sum moves
                           in front of the encoders' last layer (layer 2 runs with exchanged operands: a tile's rows sit on registers
                           and an agent's sum is register additions), that layer is folded into the decoders' first layer and into
                           the collision head's (fold_w3 / fold_b3 / fold_scale of piml_decoder_branch / piml_collision_head, packed
                           by piml_pinnsf_pack), and its gradient is recovered from the folded layers' gradients after the slot sums:
                               dW_d1 = s (G W3^T + k g_b b3^T),  dW3 = s W_d1^T G,  db3 = s k W_d1^T g_b     (G = d/d(W_d1'), g_b = d/d(b_d1')).
                           Backward: every row of an agent sees the same upstream gradient g = d/d(sum), so G2 = g[agent] * [h2 > 0]
                           directly -- the W3^T chain layer and the dW3 = G3^T H2 product over all rows (half of the backward's
                           matrix work, and the 33 MB of saved h2) are gone.  enc[i].sum_a / sum_b / relu_mask, dec[i].pooled =
                           enc[i].sum_a, dec[i].msgs = enc[i].sum_b, dec[i].fold_*, head->msgs = enc[0].h2, head->fold_*.
                           Served when piml_pinnsf_pool_train_ok(enc, nbranches): k in {2, 6, 10}, whole agents, split products,
                           more than piml_encoder_split_tiles() tiles, no keep_bits / drop_state; hipErrorInvalidValue otherwise */
#define PIML_POOL_MSGS 128 /* piml_pinnsf_fwd only (the backward is the plain one): the agents' sums of the MESSAGES written by the
                           encoder forward itself -- for training passes whose dropout mask keeps the sum from moving in front of the
                           last layer (PIML_POOL_TRAIN) and whose caller does not read the per-row messages.  The last encoder layer
                           runs with exchanged operands (a tile's rows on registers), the keep bits of a row travel from the lane that
                           drew / loaded them to the lanes that own its features, and the sums leave as enc[i].sum_a / sum_b
                           (dec[i].pooled = enc[i].sum_a, dec[i].msgs = enc[i].sum_b; the decoder completes them in `pooled`, where
                           the backward's dW1 reads them).  enc[i].msgs may be NULL: the rows of a branch are stored only where it is
                           not (the collision head's input: head->msgs = enc[0].msgs).  Everything the plain backward reads is left
                           as by the plain forward (h2, relu_mask, keep_bits, the decoders' h1 / d2).  Reference arithmetic:
                           src/models/model.py:82-119 (processor), :1279-1283 (the sum).  Served when
                           piml_pinnsf_pool_msgs_ok(enc, nbranches): k in {2, 6, 10}, whole agents, split products, relu_mask given,
                           more than piml_encoder_split_tiles_train() tiles; hipErrorInvalidValue otherwise */
int piml_pinnsf_pool_train_ok(const piml_encoder_branch* enc, int nbranches);
int piml_pinnsf_pool_msgs_ok(const piml_encoder_branch* enc, int nbranches);
int piml_pinnsf_pool_h2_ok(const piml_encoder_branch* enc, int nbranches);
int piml_pinnsf_pack_flush(void);
int piml_pinnsf_slot_sums_flush(void);   /* launch the deferred slot sums of the current device, if any are waiting (on their stream) */
int piml_pinnsf_pack(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbranches,
                     const piml_collision_head* head, int flags, void* stream);
int piml_pinnsf_fwd(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbranches,
                    const piml_collision_head* head, const float* self_features, float tau, float* acc, int flags,
                    void* stream);
int piml_pinnsf_bwd(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbranches, const float* g_pred,
                    const float* self_features, float tau, float* g_self, int flags, void* stream);

/*
 * RCCL exchange of agent-block sharding (the reference's only multi-GPU mechanism is
 * nn.DataParallel, src/models/simulators.py:64-67).  `comm` is an ncclComm_t (as void*): either one the host already
 * owns, or one made with piml_comm_unique_id (one rank) + piml_comm_init (every rank, after the 128-byte id has been
 * distributed by the host, e.g. through the torch.distributed store).  RCCL is bound at run time (dlopen of the
 * librccl already in the process); piml_comm_available() == 0 when there is none.  Return: 0, a hipError_t, or
 * 10000 + ncclResult_t.
 *   piml_allgather_state     own (floats_per_rank) -> full (world * floats_per_rank): the (p, v, a) records, 6 floats / agent
 *   piml_reducescatter_grad  full (world * floats_per_rank) partial d/d(state) -> own (floats_per_rank), summed over ranks
 *   piml_allreduce_sum       in place, e.g. the flat bucket [d/d(state) | weight gradients]
 */
typedef struct piml_comm_id { char internal[128]; } piml_comm_id;
int piml_comm_available(void);
int piml_comm_unique_id(piml_comm_id* id);
int piml_comm_init(void** comm, int world, int rank, const piml_comm_id* id);
int piml_comm_destroy(void* comm);
int piml_allgather_state(void* comm, const float* own, size_t floats_per_rank, float* full, void* stream);
int piml_reducescatter_grad(void* comm, const float* full, float* own, size_t floats_per_rank, void* stream);
int piml_allreduce_sum(void* comm, float* buf, size_t count, void* stream);

/*
 * P2P-store all-gather of the state records ("fall back to / compare with a P2P-store epilogue ... into peers'
 * double-buffered state arrays"; piml_amd/csrc/p2p.hip; the reference has no counterpart, src/models/simulators.py:64-67 is
 * nn.DataParallel).  No collective library in the data path: every rank stores its block into every peer's receive buffer
 * (xGMI stores between GPUs of a node, plain stores when ranks share a GPU) and raises a flag word per (receiver, sender).
 *   Per rank: recv = piml_p2p_alloc(2 * world * floats_per_rank * 4) and flags = piml_p2p_alloc(2 * world * 4) (zeroed);
 *   both exported (piml_p2p_export -> 64 opaque bytes, carried to the peers by the host: a pipe, the torch.distributed store)
 *   and opened there (piml_p2p_open); a rank's own buffers enter the tables as they are.
 *   piml_allgather_state_p2p(own, floats_per_rank (multiple of 4), rank, world, peer_recv[world], peer_flags[world], seq, ...):
 *   step seq = 1, 2, 3, ... (the same on every rank); on return of the launch's completion recv[seq & 1][s] holds rank s's
 *   block for every s, i.e. recv + (seq & 1) * world * floats_per_rank is the gathered (N, 6) array.  spin_limit: rounds of
 *   ~4 us the wait for the peers' flags may take (0: ~0.5 s); when it runs out status[0] (a device int the caller zeroed) is
 *   set to 1 and the launch ends -- a lost peer is an error to read back, never a hang.
 */
typedef struct piml_ipc_handle { char internal[64]; } piml_ipc_handle;
int piml_p2p_alloc(size_t bytes, void** devptr);
int piml_p2p_free(void* devptr);
int piml_p2p_export(void* devptr, piml_ipc_handle* out);
int piml_p2p_open(const piml_ipc_handle* in, void** devptr);
int piml_p2p_close(void* devptr);
int piml_p2p_copy(void* dst, const void* src, size_t bytes, void* stream);
/* The general exchange step, step counter ON THE DEVICE (no argument changes from step to step: the launch sits inside a captured
 * HIP graph and is replayed with it).  Every rank sends to every receiver r the message [scatter part | broadcast parts ...]:
 *   scatter part    = scatter_src + r * scatter_floats   (the partial d/d(state) rows of r's agent block: reduce-scatter input)
 *   broadcast parts = bcast_src[j], bcast_floats[j] floats, j < n_bcast <= PIML_P2P_MAX_PARTS
 *                     (the rank's own records forward; its weight-gradient buffers backward)
 * into recv_r[parity][rank] (parity = step & 1), raises r's flag, waits for its own `world` flags, then writes
 *   sum == 0: out_scatter[s * scatter_floats + e] / out_bcast[j][s * bcast_floats[j] + e] = what sender s sent (all-gather layout);
 *   sum == 1: out_scatter[e] / out_bcast[j][e] = the senders' parts added IN RANK ORDER (the same sum on every rank).
 * An out pointer may be NULL (that part is not wanted) and may EQUAL its source (in-place sums: no result is written before every
 * workgroup of the launch has read its sources).  Every count is a multiple of 4 floats; slot_floats >= the sum of the parts is
 * the capacity of one (parity, sender) slot of the receive buffers (piml_p2p_alloc(2 * world * slot_floats * 4)).
 * ctr: 3 + world dwords of device memory the host zeroed ONCE (ctr[0] = completed steps).  status: a device int the host zeroed;
 * STICKY -- once a wait ran out (spin_limit rounds of ~0.5 us, 0: ~0.5 s) it is 1 and every later step returns at once, on this
 * rank; the exchange is dead until the hosts rebuild it.  Reference: none (nn.DataParallel, src/models/simulators.py:64-67). */
#define PIML_P2P_MAX_PARTS 8
typedef struct piml_p2p_msg {
    const float* scatter_src;
    size_t scatter_floats;
    float* out_scatter;
    int n_bcast;
    const float* bcast_src[PIML_P2P_MAX_PARTS];
    size_t bcast_floats[PIML_P2P_MAX_PARTS];
    float* out_bcast[PIML_P2P_MAX_PARTS];
    int sum;
} piml_p2p_msg;
int piml_p2p_exchange(const piml_p2p_msg* msg, int rank, int world, float* const* peer_recv, unsigned* const* peer_flags,
                      size_t slot_floats, unsigned* ctr, unsigned spin_limit, int* status, void* stream);
int piml_allgather_state_p2p(const float* own, size_t floats_per_rank, int rank, int world, float* const* peer_recv,
                             unsigned* const* peer_flags, unsigned seq, unsigned spin_limit, int* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIML_HIP_H */
