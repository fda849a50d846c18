/*
 * libpiml_hip.so -- measurement plumbing, diagnostics and A/B switches.
 *
 * NOT part of the drop-in boundary: a host that replaces the reference's per-timestep path needs include/piml_hip.h only
 * (INTEGRATION.md sections 2 - 3).  What lives here: the HIP-event timer and the stage trace bench.py measures with, the
 * arithmetic probe the parity tests pin the selection predicates with, and the switches that select between kernel forms
 * (kept for A/B timings and for the tests that compare the forms bit for bit; every default is the measured best).  The
 * same library exports them; tests/test_abi.py checks both headers against the exports.
 */
#ifndef PIML_HIP_TUNING_H
#define PIML_HIP_TUNING_H

#include "piml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * HIP-event timer for measuring a kernel live on the stream it is launched on, also while
 * that stream is being captured into a hipGraph (the record then becomes an external
 * event-record node).  create -> record(start) -> [kernel launches] -> record(stop) ->
 * elapsed_ms (synchronises on `stop`).  Not a reference interface: measurement plumbing.
 */
int piml_timer_create(void** event);
int piml_timer_record(void* event, void* stream);
int piml_timer_elapsed_ms(void* start, void* stop, float* ms);
int piml_timer_destroy(void* event);

/*
 * Stage trace (measurement plumbing, bench.py's live per-kernel times): between piml_trace_begin and piml_trace_end every
 * launch stage of the library records a HIP event behind its launch on the launch stream (never inside a stream
 * capture); piml_trace_mark adds a mark of the caller's (`name` must stay valid until piml_trace_end; the first mark is
 * the start).  piml_trace_end closes the trace, waits for the last mark and returns the number n of intervals:
 * us[i] = microseconds between mark i and mark i + 1, names = the '\n'-separated names of marks 1 .. n.  At most 64
 * marks.  Meaningful only when the stream is kept busy in front of the traced launches (else host gaps are included).
 */
int piml_trace_begin(void);
int piml_trace_mark(const char* name, void* stream);
int piml_trace_end(char* names, int names_cap, float* us, int us_cap);

/*
 * Diagnostic (tests only): evaluates, per element, the exact float32 arithmetic of the
 * neighbour-selection predicates -- dist = |r| as torch.norm computes it and
 * cos = torch.cosine_similarity(r, h) (src/data/data.py:434, 439-440) -- so that it can be
 * pinned bit-for-bit against the CPU restatement.  All arrays have n elements.
 */
int piml_probe_arith(const float* rx, const float* ry, const float* hx, const float* hy,
                     float* dist, float* cosv, int n, void* stream);

/* Up to this many 32-row tiles (both branches together; default 640) a LONE forward -- no relu_mask, no backward -- runs with
 * four waves per tile instead of one (few rows: rollouts of real clips); the two forms are bitwise identical.  Returns the
 * previous value; < 0 only queries. */
long long piml_encoder_split_tiles(long long tiles);
/* The same bound for a TRAINING pass (every branch carries relu_mask, i.e. a backward follows; with or without a dropout mask):
 * default 48 tiles -- with the one-pass backward and the wave-major tile order the one-wave kernels win from the real clips'
 * sizes on (122 agents = 62 tiles), while a lone forward still wants four waves per tile up to piml_encoder_split_tiles().  piml_encoder_split_tiles(tiles >= 0) sets BOTH bounds (A/B),
 * piml_encoder_split_tiles(-2) puts both back to their defaults (environment PIML_ENC_SPLIT_TILES[_TRAIN] at load time).
 * Returns the previous value; < 0 only queries. */
long long piml_encoder_split_tiles_train(long long tiles);
/* Arithmetic of the two 128 x 128 layers' products.  1 (default): every f32 product as six bf16 x bf16 partial products of
 * exact three-way splits of both factors, accumulated in f32 (v_mfma_f32_32x32x16_bf16; what is dropped is below one f32
 * rounding of the product); 0: the f32 matrix-core instruction (v_mfma_f32_32x32x2_f32).  Environment at load time:
 * PIML_ENC_PRODUCTS=f32.  Returns the previous value; < 0 only queries. */
int piml_encoder_products(int split_bf16);
/* The same choice for the many-rows kernels of the row decoder (the bottleneck variants' decoder + predictor per neighbour row,
 * src/models/model.py:1116-1122, 1182-1190): 1 (default) = split bf16 products, the weights split while the workgroup stages them;
 * 0 = the f32 matrix-core instruction.  Environment at load time: PIML_ROWDEC_PRODUCTS=f32.  Returns the previous value; < 0 queries. */
int piml_rowdecoder_products(int split_bf16);
/* Weight gradients of the split-product backward above piml_encoder_split_tiles() tiles: 1 (default) = layer-split
 * workgroups (piml_amd/csrc/encoder_dw2.hip: a workgroup takes ONE of the two 128 x 128 products over a longer slab --
 * half the partial bytes -- and recomputes h1 from x when the branches carry none), 0 = one slab and both products per
 * workgroup (enc_bwd_dw_x3_kernel).  Environment at load time: PIML_ENC_DW2=0.  Returns the previous value; < 0 queries. */
int piml_encoder_dw2(int layer_split);
/* One-pass backward above piml_encoder_split_tiles() tiles (piml_amd/csrc/encoder_bwd3.hip; reference: the autograd of
 * src/models/model.py:40-65 under :82-119): 1 (default) = where the layer-split weight gradients run, the forward left
 * `relu_mask` and the branches carry the same kinds of upstream gradients, the dX chain and dW2 / dW1 / db2 / db1 are ONE launch
 * that keeps the pre-activation gradients on the CU -- `g2` / `g1` are neither written nor read and may be NULL -- and dW3 /
 * db3 are the layer-0 workgroups of piml_encoder_dw2's kernel; 0 = the dX kernel writes g2 / g1 and the weight-gradient kernel
 * reads them back; 2 = the one-pass kernel as eight waves of 16-feature blocks (encoder_bwd4.hip, two waves per SIMD; measured
 * level with the four-wave form, kept for A/B).  In the one-pass forms dW3 / db3 are a second phase of the same launch
 * (PIML_ENC_FUSED_DW3=0: the layer-0 workgroups of piml_encoder_dw2's kernel in a launch of their own).  Environment at load
 * time: PIML_ENC_FUSED_BWD=0 / 1 / 2.  Returns the previous value; < 0 only queries. */
int piml_encoder_fused_bwd(int on);
/* Backward of the sums path (PIML_POOL_TRAIN of piml_pinnsf_bwd; reference: the autograd of src/models/model.py:40-65 below
 * the neighbour-axis sum :1279-1283): 2 (default) = two crews of four waves per workgroup -- the dX chain and the weight-gradient
 * products on two waves of every SIMD (piml_amd/csrc/encoder_bwd5.hip); 1 = one wave per SIMD (encoder_bwd3.hip, round 5).  The
 * two forms agree bitwise.  Environment at load time: PIML_ENC_SUMS_BWD=1 / 2.  Returns the previous value; other arguments
 * only query. */
int piml_encoder_sums_bwd(int form);
/* The library-owned side streams of PIML_FORK (piml_pinnsf_fwd / bwd with the independent stages forked: measured slower inside
 * captured graphs, kept for A/B): created per device on first use, outside any capture; idempotent. */
int piml_pinnsf_streams_init(void);

#ifdef __cplusplus
}
#endif
#endif /* PIML_HIP_TUNING_H */
