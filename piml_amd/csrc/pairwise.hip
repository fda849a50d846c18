// Closed-form social force (MLAPM.step) forward + analytic backward, collision matrices and
// the small per-neighbour physics labels, for gfx950.
//
// MLAPM (reference src/models/mlapm.py:10-58) materialises (N,N,2) x 3-5 and an (N,N,2,2)
// rotation tensor; here one 64-lane wavefront owns one focal agent, every agent's
// (p, v[, e, G]) record is staged once per workgroup into an LDS tile, lanes stream the
// sources and the per-agent force is a wave reduction.  No MFMA: there is no contraction.
#include "common.hpp"
#include "../../include/piml_hip.h"

#include <cmath>
#include <cstdlib>

namespace piml {

constexpr int kMlTile = 2048;   // agents per LDS tile (fwd 32 KiB, bwd 64 KiB)

struct MlapmParams {
    int variant;                 // 0 raw, 1 GC, 2 UCY (mlapm.py:28-53)
    float tau, A, B, Cc, D, cth, sth, r2;   // cos/sin of theta, 2*radius
    float B2, C2, D2;            // B, C, D pre-multiplied by log2(e): exp(x) = exp2(x * log2 e)
    int skip_absent;             // 1: sources with a NaN position contribute nothing (absent agents)
    int ucy_two_phase;           // backward, UCY: the two-phase form (PIML_MLAPM_UCY_TWO_PHASE=0 keeps the scalar loop)
};

// MLAPM is a smooth force law checked to 1e-5 relative (not a discrete selection like relfeat), so
// its pair arithmetic uses the hardware reciprocal-sqrt / reciprocal / exp2 units (<= 1 ulp
// each) instead of the multi-instruction IEEE division / sqrt / expf expansions.
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// UCY collision flag (mlapm.py:43-47) with EXACTLY the reference's float32 operations -- it is a discrete decision,
// so the fast reciprocal / squared-domain shortcuts of the smooth terms do not apply here:
//   |vr| < 2R  or  |vr + vv| < 2R  or  (0 < tmin < 1 and dmin < 2R),
//   tmin = -(vr.vv) / (vv.vv),  dmin = sqrt(vr.vr - (vr.vv)^2 / (vv.vv)),  dots = x*x' + y*y' (two products, one add),
//   norms = torch.norm on 2-vectors = sqrt(fma(y, y, x*x)).  A NaN dmin (negative argument) compares false.
__device__ __forceinline__ bool ucy_collision(float rx, float ry, float wx, float wy, float two_r) {
    bool coll = norm2(rx, ry) < two_r;
    coll |= norm2(__fadd_rn(rx, wx), __fadd_rn(ry, wy)) < two_r;
    const float rw = __fadd_rn(__fmul_rn(rx, wx), __fmul_rn(ry, wy));
    const float ww = __fadd_rn(__fmul_rn(wx, wx), __fmul_rn(wy, wy));
    const float rr = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
    const float tmin = __fdiv_rn(-rw, ww);
    const float dmin = sqrtf(__fsub_rn(rr, __fdiv_rn(__fmul_rn(rw, rw), ww)));
    coll |= (tmin > 0.f) && (tmin < 1.f) && (dmin < two_r);
    return coll;
}

// One ordered pair: focal (vix, viy, ex, ey) at the origin, source at (rx, ry) with
// relative velocity (wx, wy).  Returns view * A * g * direction (mlapm.py:25-53).
__device__ __forceinline__ float2 mlapm_pair(const MlapmParams& P, float rx, float ry, float wx, float wy,
                                             float vix, float viy, float ex, float ey) {
    const float d2 = rx * rx + ry * ry;
    const bool pos = d2 > 0.f;                                      // NaN -> false, handled below
    const float rinv = fast_rsq(d2);
    const float r = pos ? d2 * rinv : d2;                           // :26 (0 stays 0, NaN stays NaN)
    const float view = (vix * rx + viy * ry > 0.f) ? 1.f : 0.f;     // :27
    const float ninv = pos ? rinv : 0.f;                            // F.normalize: 0 / eps = 0
    const float nx = rx * ninv, ny = ry * ninv;
    float g, dx, dy;
    if (P.variant == 0) {
        g = fast_exp2(P.B2 * r);                                    // :29
        dx = nx; dy = ny;
    } else {
        const float cr = rx * ey - ry * ex;                         // :34 / :48
        // theta = -sign(cr) * theta, 0 -> +theta; sign(NaN) = NaN propagates like the reference
        const float st = cr > 0.f ? -P.sth : (cr <= 0.f ? P.sth : cr);
        dx = P.cth * nx - st * ny; dy = st * nx + P.cth * ny;       // :36-39
        if (P.variant == 1) {
            const float w2 = wx * wx + wy * wy;
            // cosine_similarity clamps both norms at 1e-8 (:32)
            const float cs = (rx * wx + ry * wy) * fminf(rinv, 1e8f) * fminf(fast_rsq(w2), 1e8f);
            g = fast_exp2(P.B2 * r + P.C2 * cs + P.D2 * r * cs);    // :40
        } else {
            const bool coll = ucy_collision(rx, ry, wx, wy, P.r2);      // :43-47, exact
            g = coll ? fast_exp2(P.B2 * r + P.C2) : 1.f;            // :53 (with coll.unsqueeze(-1), Q8)
            if (r != r) g = r;                                      // NaN poisons like the reference
        }
    }
    const float s = view * P.A * g;
    return make_float2(s * dx, s * dy);
}

// Two ordered pairs per lane with packed fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32): the raw and GC force laws
// (variants 0, 1).  Same expressions as mlapm_pair, element-wise on 2-vectors; products feeding sums are fused
// (fma), which the 1e-5 relative bar of this smooth force law allows (the selections are unaffected).
// (The variant stays a run-time value on purpose: a kernel specialised per variant at compile time measured
// slower -- GC forward 38.7 us against 26.1 us at N = 4096; round 4, again, on the rollout frame: 47.5 against 34.1 us.)
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_sel(bool c0, bool c1, v2f a, v2f b) { return v2f{c0 ? a.x : b.x, c1 ? a.y : b.y}; }

__device__ __forceinline__ void mlapm_pair2(const MlapmParams& P, v2f rx, v2f ry, v2f wx, v2f wy, float vix, float viy,
                                            float ex, float ey, v2f& fx, v2f& fy) {
    const v2f zero = {0.f, 0.f}, one = {1.f, 1.f};
    const v2f d2 = pk_fma(ry, ry, rx * rx);
    const bool p0 = d2.x > 0.f, p1 = d2.y > 0.f;                                  // NaN -> false
    const v2f rinv = {fast_rsq(d2.x), fast_rsq(d2.y)};
    const v2f r = pk_sel(p0, p1, d2 * rinv, d2);                                  // :26
    const v2f dot = pk_fma(v2f{viy, viy}, ry, v2f{vix, vix} * rx);
    const v2f view = pk_sel(dot.x > 0.f, dot.y > 0.f, one, zero);                 // :27
    const v2f ninv = pk_sel(p0, p1, rinv, zero);
    const v2f nx = rx * ninv, ny = ry * ninv;
    v2f g, dx, dy;
    if (P.variant == 0) {
        const v2f a = v2f{P.B2, P.B2} * r;
        g = v2f{fast_exp2(a.x), fast_exp2(a.y)};                                  // :29
        dx = nx; dy = ny;
    } else {
        const v2f cr = pk_fma(rx, v2f{ey, ey}, -(ry * v2f{ex, ex}));              // :34
        const v2f st = {cr.x > 0.f ? -P.sth : (cr.x <= 0.f ? P.sth : cr.x),
                        cr.y > 0.f ? -P.sth : (cr.y <= 0.f ? P.sth : cr.y)};
        const v2f cth = {P.cth, P.cth};
        dx = pk_fma(cth, nx, -(st * ny));                                         // :36-39
        dy = pk_fma(st, nx, cth * ny);
        const v2f w2 = pk_fma(wy, wy, wx * wx);
        const v2f ri8 = {fminf(rinv.x, 1e8f), fminf(rinv.y, 1e8f)};
        const v2f qi8 = {fminf(fast_rsq(w2.x), 1e8f), fminf(fast_rsq(w2.y), 1e8f)};
        const v2f cs = pk_fma(ry, wy, rx * wx) * ri8 * qi8;                       // :32
        const v2f a = pk_fma(v2f{P.D2, P.D2} * r, cs, pk_fma(v2f{P.C2, P.C2}, cs, v2f{P.B2, P.B2} * r));
        g = v2f{fast_exp2(a.x), fast_exp2(a.y)};                                  // :40
    }
    const v2f sc = view * v2f{P.A, P.A} * g;
    fx = sc * dx;
    fy = sc * dy;
}

// UCY (variant 2) in two phases (round 4).  g = exp(B r + C) only for pairs the collision predicate flags, 1 otherwise
// (mlapm.py:43-53), and the predicate -- three square roots and two divisions in the reference's exact float32 operations --
// cost more than the rest of the pair.  Phase 1 gives every pair its g = 1 term on packed arithmetic and a CONSERVATIVE
// distance test: every clause of the predicate implies that the relative position comes within 2R of the origin for some
// t in [0, 1] of r + t w, hence |r| - |w| < 2R; a pair with (|r| - |w|)^2 - (2R)^2 > 1e-6 (|r|^2 + 1) (the slack covers the
// roundings of both sides, 1e-7 relative, with a factor of ten) cannot be flagged.  The others -- ~4 % of the pairs of a
// 4096-agent hall -- are compacted into a per-wave ring and get the exact predicate 64 at a time; a flagged pair adds the
// difference (g - 1) x its g = 1 term.  Same sums up to the order of the additions.
__device__ __forceinline__ void mlapm_pair2_ucy(const MlapmParams& P, v2f rx, v2f ry, v2f wx, v2f wy, float vix, float viy,
                                                float ex, float ey, v2f& fx, v2f& fy, bool& near0, bool& near1) {
    const v2f zero = {0.f, 0.f}, one = {1.f, 1.f};
    const v2f d2 = pk_fma(ry, ry, rx * rx);
    const bool p0 = d2.x > 0.f, p1 = d2.y > 0.f;                                  // NaN -> false
    const v2f rinv = {fast_rsq(d2.x), fast_rsq(d2.y)};
    const v2f r = pk_sel(p0, p1, d2 * rinv, d2);
    const v2f dot = pk_fma(v2f{viy, viy}, ry, v2f{vix, vix} * rx);
    const v2f view = pk_sel(dot.x > 0.f, dot.y > 0.f, one, zero);                 // :27
    const v2f ninv = pk_sel(p0, p1, rinv, zero);
    const v2f nx = rx * ninv, ny = ry * ninv;
    const v2f cr = pk_fma(rx, v2f{ey, ey}, -(ry * v2f{ex, ex}));                  // :48
    const v2f st = {cr.x > 0.f ? -P.sth : (cr.x <= 0.f ? P.sth : cr.x),
                    cr.y > 0.f ? -P.sth : (cr.y <= 0.f ? P.sth : cr.y)};
    const v2f cth = {P.cth, P.cth};
    const v2f sc = view * v2f{P.A, P.A};                                          // g = 1
    fx = sc * pk_fma(cth, nx, -(st * ny));
    fy = sc * pk_fma(st, nx, cth * ny);
    // conservative test: far = certainly not flagged
    const v2f w2 = pk_fma(wy, wy, wx * wx);
    const v2f wn = w2 * v2f{fast_rsq(fmaxf(w2.x, 1e-30f)), fast_rsq(fmaxf(w2.y, 1e-30f))};
    const v2f a = r - wn;
    const v2f slack = pk_fma(v2f{1e-6f, 1e-6f}, d2, v2f{1e-6f + P.r2 * P.r2, 1e-6f + P.r2 * P.r2});
    near0 = !(a.x > 0.f && a.x * a.x > slack.x);                                  // (NaN: near)
    near1 = !(a.y > 0.f && a.y * a.y > slack.y);
}

// exact second phase of one candidate: the difference between its flagged term and the g = 1 term phase 1 added
__device__ __forceinline__ float2 mlapm_ucy_correction(const MlapmParams& P, float rx, float ry, float wx, float wy,
                                                       float vix, float viy, float ex, float ey) {
    if (!ucy_collision(rx, ry, wx, wy, P.r2)) return make_float2(0.f, 0.f);       // :43-47, exact
    const float d2 = rx * rx + ry * ry;
    const bool pos = d2 > 0.f;
    const float rinv = fast_rsq(d2);
    const float r = pos ? d2 * rinv : d2;
    const float view = (vix * rx + viy * ry > 0.f) ? 1.f : 0.f;
    const float ninv = pos ? rinv : 0.f;
    const float nx = rx * ninv, ny = ry * ninv;
    const float cr = rx * ey - ry * ex;
    const float st = cr > 0.f ? -P.sth : (cr <= 0.f ? P.sth : cr);
    const float s = view * P.A * (fast_exp2(P.B2 * r + P.C2) - 1.f);              // :53 minus the g = 1 term
    return make_float2(s * (P.cth * nx - st * ny), s * (st * nx + P.cth * ny));
}

// ROLL: one frame of the simulation loop of src/main_mlapm.py:18-36 in this launch.  The state of frame t - 1 is read from
// the trajectory itself -- an agent that came within `radius` of its destination in frame t - 1 >= 1 is absent (NaN) from
// frame t on (:34; frame 0 is the caller's initial state, not tested) -- the new velocity and p + v dt go to frame t, and the
// last workgroup to finish moves the device-side frame counter on: a frame is ONE launch, K frames one captured graph.
struct MlapmRoll {
    float2* traj_p;              // (frames, N, 2)
    float2* traj_v;
    long long* t;                // frame to produce (device side)
    unsigned* done;              // workgroups that finished this launch
    long long frames;
    float radius;
};

template <bool ROLL>
__device__ __forceinline__ void mlapm_state(const float2* __restrict__ p, const float2* __restrict__ v, const float2* __restrict__ dest,
                                            int j, bool test_arrival, float radius, float2& a, float2& b) {
    a = p[j]; b = v[j];
    if (ROLL && test_arrival) {
        const float2 d = dest[j];
        if (norm2(a.x - d.x, a.y - d.y) < radius) {                 // NaN compares false: absent stays absent
            const float nan = __builtin_nanf("");
            a = make_float2(nan, nan); b = a;
        }
    }
}

template <int WAVES, bool ROLL = false>
__global__ __launch_bounds__(WAVES * 64) void mlapm_fwd_kernel(
        const float2* __restrict__ p, const float2* __restrict__ v, const float* __restrict__ v0,
        const float2* __restrict__ dest, int N, MlapmParams P, float dt, float2* __restrict__ action,
        float2* __restrict__ force, MlapmRoll Rr) {
    __shared__ float4 tile[kMlTile];                        // (px, py, vx, vy)
    __shared__ unsigned short ucy_ring[WAVES][256];        // UCY: tile-local indices of the pairs that need the exact predicate
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * WAVES + wave;
    const bool has = i < N;
    bool test_arrival = false;
    long long tf = 0;
    if (ROLL) {
        tf = *Rr.t;
        if (tf < 1 || tf >= Rr.frames) return;              // past the trajectory: nothing to do (the counter stays)
        p = Rr.traj_p + (tf - 1) * N; v = Rr.traj_v + (tf - 1) * N;
        action = Rr.traj_v + tf * N;
        test_arrival = tf - 1 >= 1;
    }
    float2 pi, vi;
    mlapm_state<ROLL>(p, v, dest, has ? i : 0, test_arrival, Rr.radius, pi, vi);
    const float2 di = dest[has ? i : 0];
    float ex = di.x - pi.x, ey = di.y - pi.y;
    const float en = fmaxf(norm2(ex, ey), 1e-12f);          // :21
    ex /= en; ey /= en;
    float sx = 0.f, sy = 0.f;
    v2f acc2x = {0.f, 0.f}, acc2y = {0.f, 0.f};
    for (int base = 0; base < N; base += kMlTile) {
        const int tn = min(kMlTile, N - base);
        __syncthreads();
        for (int t = threadIdx.x; t < tn; t += WAVES * 64) {
            float2 a, b;
            mlapm_state<ROLL>(p, v, dest, base + t, test_arrival, Rr.radius, a, b);
            tile[t] = make_float4(a.x, a.y, b.x, b.y);
        }
        __syncthreads();
        if (!has) continue;
        int j = lane;
        if (P.variant != 2) {
            // two sources per lane and iteration (j, j + 64), packed arithmetic
            const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, vix2 = {vi.x, vi.x}, viy2 = {vi.y, vi.y};
            for (; j + 64 < tn; j += 128) {
                const float4 a = tile[j], b = tile[j + 64];
                v2f fx, fy;
                mlapm_pair2(P, v2f{a.x, b.x} - pix, v2f{a.y, b.y} - piy, v2f{a.z, b.z} - vix2, v2f{a.w, b.w} - viy2,
                            vi.x, vi.y, ex, ey, fx, fy);
                if (P.skip_absent) {                                        // absent sources contribute nothing
                    if (a.x != a.x || a.y != a.y) { fx.x = 0.f; fy.x = 0.f; }
                    if (b.x != b.x || b.y != b.y) { fx.y = 0.f; fy.y = 0.f; }
                }
                acc2x += fx; acc2y += fy;
            }
        }
        if (P.variant == 2) {
            // phase 1 on two sources per lane (j, j + 64; the second clamped and masked at the tile's end), candidates into
            // the ring; phase 2 whenever 64 are waiting, and for what is left at the end of the tile
            const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, vix2 = {vi.x, vi.x}, viy2 = {vi.y, vi.y};
            unsigned short* ring = ucy_ring[uniform(wave)];
            unsigned head = 0, tail = 0;
            for (int j0 = 0;; j0 += 128) {
                const bool more = j0 < tn;
                if (more) {
                    const int ja = j0 + lane, jb = j0 + 64 + lane;
                    const bool va = ja < tn, vb = jb < tn;
                    const float4 a = tile[va ? ja : 0], b = tile[vb ? jb : 0];
                    v2f fx, fy;
                    bool na, nb;
                    mlapm_pair2_ucy(P, v2f{a.x, b.x} - pix, v2f{a.y, b.y} - piy, v2f{a.z, b.z} - vix2, v2f{a.w, b.w} - viy2,
                                    vi.x, vi.y, ex, ey, fx, fy, na, nb);
                    const bool absent_a = P.skip_absent && (a.x != a.x || a.y != a.y);
                    const bool absent_b = P.skip_absent && (b.x != b.x || b.y != b.y);
                    if (!va || absent_a) { fx.x = 0.f; fy.x = 0.f; na = false; }
                    if (!vb || absent_b) { fx.y = 0.f; fy.y = 0.f; nb = false; }
                    acc2x += fx; acc2y += fy;
                    const u64 ma = __builtin_amdgcn_ballot_w64(na);
                    if (ma) {
                        if (na) ring[(tail + mbcnt(ma)) & 255] = (unsigned short)ja;
                        tail += (unsigned)__builtin_popcountll(ma);
                    }
                    const u64 mb = __builtin_amdgcn_ballot_w64(nb);
                    if (mb) {
                        if (nb) ring[(tail + mbcnt(mb)) & 255] = (unsigned short)jb;
                        tail += (unsigned)__builtin_popcountll(mb);
                    }
                }
                while (tail - head >= (more ? 64u : 1u)) {
                    const unsigned n = min(64u, tail - head);
                    __builtin_amdgcn_wave_barrier();
                    if ((unsigned)lane < n) {
                        const float4 c = tile[ring[(head + lane) & 255]];
                        const float2 t = mlapm_ucy_correction(P, c.x - pi.x, c.y - pi.y, c.z - vi.x, c.w - vi.y, vi.x, vi.y, ex, ey);
                        sx += t.x; sy += t.y;
                    }
                    head += n;
                }
                if (!more) break;
            }
            continue;
        }
        for (; j < tn; j += 64) {
            const float4 s = tile[j];
            if (P.skip_absent && (s.x != s.x || s.y != s.y)) continue;      // absent source
            const float2 t = mlapm_pair(P, s.x - pi.x, s.y - pi.y, s.z - vi.x, s.w - vi.y, vi.x, vi.y, ex, ey);
            sx += t.x; sy += t.y;
        }
    }
    sx += acc2x.x + acc2x.y; sy += acc2y.x + acc2y.y;
    sx = wave_sum(sx); sy = wave_sum(sy);
    if (has && lane == 0) {
        const float v0i = v0[i];
        const float fx = (v0i * ex - vi.x) / P.tau - sx;    // :22, :29/:40/:53
        const float fy = (v0i * ey - vi.y) / P.tau - sy;
        if (force) force[i] = make_float2(fx, fy);
        const float2 vn = make_float2(vi.x + fx * dt, vi.y + fy * dt);   // :57
        action[i] = vn;
        if (ROLL) Rr.traj_p[tf * N + i] = make_float2(pi.x + vn.x * dt, pi.y + vn.y * dt);   // main_mlapm.py:25 (explicit Euler)
    }
    if (ROLL) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(Rr.done, 1u) == gridDim.x - 1) {       // every workgroup has read the counter and written its rows
                *Rr.done = 0u;
                *Rr.t = tf + 1;
            }
        }
    }
}

// d(-G . T)/d(vr), d(-G . T)/d(vv) of one ordered pair, T the pair term of mlapm_pair and
// (Gx, Gy) the upstream gradient on the focal agent's force.  view, the rotation sign and the
// UCY collision flag are piecewise constant and carry no gradient (as in autograd).
__device__ __forceinline__ void mlapm_pair_grad(const MlapmParams& P, float rx, float ry, float wx, float wy,
                                                float vix, float viy, float ex, float ey, float Gx, float Gy,
                                                float& ax, float& ay, float& bx, float& by, int ucy_flag = -1) {
    ax = ay = bx = by = 0.f;
    const float d2 = rx * rx + ry * ry;
    if (!(d2 > 0.f) || !(vix * rx + viy * ry > 0.f)) return;
    const float rinv = fast_rsq(d2), r = d2 * rinv;
    const float nx = rx * rinv, ny = ry * rinv;
    float st = 0.f, ct = 1.f;
    if (P.variant != 0) {
        const float cr = rx * ey - ry * ex;
        st = cr > 0.f ? -P.sth : P.sth; ct = P.cth;
    }
    const float ux = ct * Gx + st * Gy, uy = -st * Gx + ct * Gy;   // R^T G
    const float un = ux * nx + uy * ny;
    float phi2, fx, fy, hx = 0.f, hy = 0.f;                        // phi*log2e, d(phi)/d(vr), d(phi)/d(vv)
    if (P.variant == 0) {
        phi2 = P.B2 * r; fx = P.B * nx; fy = P.B * ny;
    } else if (P.variant == 1) {
        const float w2 = wx * wx + wy * wy;
        const float ri8 = fminf(rinv, 1e8f), qi8 = fminf(fast_rsq(w2), 1e8f);
        const float n8x = rx * ri8, n8y = ry * ri8, mx = wx * qi8, my = wy * qi8;
        const float cs = n8x * mx + n8y * my;
        phi2 = P.B2 * r + P.C2 * cs + P.D2 * r * cs;
        const float k1 = P.Cc + P.D * r;
        const bool r_ok = r > 1e-8f, q_ok = w2 > 1e-16f;
        const float csx = (r_ok ? mx - cs * n8x : mx) * ri8;       // d(cs)/d(vr)
        const float csy = (r_ok ? my - cs * n8y : my) * ri8;
        fx = P.B * nx + k1 * csx + P.D * cs * nx;
        fy = P.B * ny + k1 * csy + P.D * cs * ny;
        hx = k1 * (q_ok ? n8x - cs * mx : n8x) * qi8;              // d(cs)/d(vv)
        hy = k1 * (q_ok ? n8y - cs * my : n8y) * qi8;
    } else {
        const float cf = (ucy_flag >= 0 ? ucy_flag != 0 : ucy_collision(rx, ry, wx, wy, P.r2)) ? 1.f : 0.f;      // the forward's exact flag
        phi2 = (P.B2 * r + P.C2) * cf; fx = P.B * cf * nx; fy = P.B * cf * ny;
    }
    const float AE = -P.A * fast_exp2(phi2);
    ax = AE * (un * fx + (ux - un * nx) * rinv);
    ay = AE * (un * fy + (uy - un * ny) * rinv);
    bx = AE * un * hx;
    by = AE * un * hy;
}

// mlapm_pair_grad for the raw / GC laws on 2-vectors: element 0 and element 1 are two ordered pairs that may
// differ in everything (the backward kernel feeds it the two roles of one unordered pair).  Same expressions,
// packed fp32 arithmetic; a pair outside the view half plane (or at zero distance) yields exact zeros.
__device__ __forceinline__ void mlapm_pair_grad2(const MlapmParams& P, v2f rx, v2f ry, v2f wx, v2f wy, v2f vix, v2f viy,
                                                 v2f ex, v2f ey, v2f Gx, v2f Gy, v2f& ax, v2f& ay, v2f& bx, v2f& by) {
    const v2f zero = {0.f, 0.f};
    const v2f d2 = pk_fma(ry, ry, rx * rx);
    const v2f dot = pk_fma(viy, ry, vix * rx);
    const bool ok0 = (d2.x > 0.f) && (dot.x > 0.f), ok1 = (d2.y > 0.f) && (dot.y > 0.f);
    const v2f rinv = {fast_rsq(d2.x), fast_rsq(d2.y)};
    const v2f r = d2 * rinv;
    const v2f nx = rx * rinv, ny = ry * rinv;
    v2f ux = Gx, uy = Gy;                                          // R^T G with theta = 0
    if (P.variant != 0) {
        const v2f cr = pk_fma(rx, ey, -(ry * ex));
        const v2f st = {cr.x > 0.f ? -P.sth : P.sth, cr.y > 0.f ? -P.sth : P.sth};
        const v2f ct = {P.cth, P.cth};
        ux = pk_fma(ct, Gx, st * Gy);
        uy = pk_fma(ct, Gy, -(st * Gx));
    }
    const v2f un = pk_fma(ux, nx, uy * ny);
    const v2f Bv = {P.B, P.B};
    v2f phi2, fx, fy, hx = zero, hy = zero;
    if (P.variant == 0) {
        phi2 = v2f{P.B2, P.B2} * r;
        fx = Bv * nx; fy = Bv * ny;
    } else if (P.variant == 2) {      // UCY with the collision flag off (the flagged pairs get their difference afterwards)
        phi2 = zero; fx = zero; fy = zero;
    } else {
        const v2f w2 = pk_fma(wy, wy, wx * wx);
        const v2f ri8 = {fminf(rinv.x, 1e8f), fminf(rinv.y, 1e8f)};
        const v2f qi8 = {fminf(fast_rsq(w2.x), 1e8f), fminf(fast_rsq(w2.y), 1e8f)};
        const v2f n8x = rx * ri8, n8y = ry * ri8, mx = wx * qi8, my = wy * qi8;
        const v2f cs = pk_fma(n8x, mx, n8y * my);
        phi2 = pk_fma(v2f{P.D2, P.D2} * r, cs, pk_fma(v2f{P.C2, P.C2}, cs, v2f{P.B2, P.B2} * r));
        const v2f Dv = {P.D, P.D};
        const v2f k1 = pk_fma(Dv, r, v2f{P.Cc, P.Cc});
        const bool r0 = r.x > 1e-8f, r1 = r.y > 1e-8f, q0 = w2.x > 1e-16f, q1 = w2.y > 1e-16f;
        const v2f csx = pk_sel(r0, r1, pk_fma(-cs, n8x, mx), mx) * ri8;          // d(cs)/d(vr)
        const v2f csy = pk_sel(r0, r1, pk_fma(-cs, n8y, my), my) * ri8;
        const v2f dcs = Dv * cs;
        fx = pk_fma(dcs, nx, pk_fma(k1, csx, Bv * nx));
        fy = pk_fma(dcs, ny, pk_fma(k1, csy, Bv * ny));
        hx = k1 * pk_sel(q0, q1, pk_fma(-cs, mx, n8x), n8x) * qi8;                // d(cs)/d(vv)
        hy = k1 * pk_sel(q0, q1, pk_fma(-cs, my, n8y), n8y) * qi8;
    }
    const v2f AE = v2f{-P.A, -P.A} * v2f{fast_exp2(phi2.x), fast_exp2(phi2.y)};
    const v2f aun = AE * un;
    ax = pk_sel(ok0, ok1, AE * pk_fma(un, fx, pk_fma(-un, nx, ux) * rinv), zero);
    ay = pk_sel(ok0, ok1, AE * pk_fma(un, fy, pk_fma(-un, ny, uy) * rinv), zero);
    bx = pk_sel(ok0, ok1, aun * hx, zero);
    by = pk_sel(ok0, ok1, aun * hy, zero);
}

// Backward: one wavefront per agent x accumulates BOTH its focal-side sums (-a_xo, -b_xo) and
// its source-side sums (+a_ox, +b_ox) by evaluating every pair in both roles, so no atomics
// and a fixed summation order (bitwise reproducible).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mlapm_bwd_kernel(
        const float2* __restrict__ g_action, const float2* __restrict__ p, const float2* __restrict__ v,
        const float* __restrict__ v0, const float2* __restrict__ dest, int N, MlapmParams P, float dt,
        float2* __restrict__ g_p, float2* __restrict__ g_v, float* __restrict__ g_v0,
        float2* __restrict__ g_dest) {
    __shared__ float4 tile_pv[kMlTile];                     // (px, py, vx, vy)
    __shared__ float4 tile_eg[kMlTile];                     // (ex, ey, Gx, Gy), G = dt * g_action
    __shared__ unsigned short ucy_ring[WAVES][128];        // UCY: tile-local indices of the pairs that need the exact predicate
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * WAVES + wave;
    const bool has = x < N;
    const float2 px = p[has ? x : 0], vx = v[has ? x : 0], dx = dest[has ? x : 0], ga = g_action[has ? x : 0];
    float ex = dx.x - px.x, ey = dx.y - px.y;
    const float dn = norm2(ex, ey), en = fmaxf(dn, 1e-12f);
    ex /= en; ey /= en;
    const float Gx = ga.x * dt, Gy = ga.y * dt;
    float spx = 0.f, spy = 0.f, svx = 0.f, svy = 0.f;
    for (int base = 0; base < N; base += kMlTile) {
        const int tn = min(kMlTile, N - base);
        __syncthreads();
        for (int t = threadIdx.x; t < tn; t += WAVES * 64) {
            const float2 a = p[base + t], b = v[base + t], d = dest[base + t], g = g_action[base + t];
            float qx = d.x - a.x, qy = d.y - a.y;
            const float qn = fmaxf(norm2(qx, qy), 1e-12f);
            tile_pv[t] = make_float4(a.x, a.y, b.x, b.y);
            tile_eg[t] = make_float4(qx / qn, qy / qn, g.x * dt, g.y * dt);
        }
        __syncthreads();
        if (!has) continue;
        if (P.variant != 2) {
            // both roles of the pair (x focal / o source, o focal / x source) as the two elements of packed vectors
            for (int j = lane; j < tn; j += 64) {
                const float4 s = tile_pv[j], t = tile_eg[j];
                const float rx = s.x - px.x, ry = s.y - px.y, wx = s.z - vx.x, wy = s.w - vx.y;
                v2f ax, ay, bx, by;
                mlapm_pair_grad2(P, v2f{rx, -rx}, v2f{ry, -ry}, v2f{wx, -wx}, v2f{wy, -wy}, v2f{vx.x, s.z},
                                 v2f{vx.y, s.w}, v2f{ex, t.x}, v2f{ey, t.y}, v2f{Gx, t.z}, v2f{Gy, t.w}, ax, ay, bx, by);
                spx += ax.y - ax.x; spy += ay.y - ay.x; svx += bx.y - bx.x; svy += by.y - by.x;
            }
            continue;
        }
        if (P.ucy_two_phase) {
            // UCY in two phases (see mlapm_pair2_ucy): every pair with the flag off on packed arithmetic, both roles at once;
            // the pairs the conservative distance test cannot rule out -- the predicate is the same for both roles -- go
            // through a ring, get the exact flag once, and a flagged pair adds [flag on] - [flag off] for both roles
            unsigned short* ring = ucy_ring[uniform(wave)];
            unsigned head = 0, tail = 0;
            const float lim = 1e-6f + P.r2 * P.r2;
            for (int j0 = 0;; j0 += 64) {
                const bool more = j0 < tn;
                if (more) {
                    const int j = j0 + lane;
                    const bool in = j < tn;
                    const float4 s = tile_pv[in ? j : 0], t = tile_eg[in ? j : 0];
                    const float rx = s.x - px.x, ry = s.y - px.y, wx = s.z - vx.x, wy = s.w - vx.y;
                    v2f ax, ay, bx, by;
                    mlapm_pair_grad2(P, v2f{rx, -rx}, v2f{ry, -ry}, v2f{wx, -wx}, v2f{wy, -wy}, v2f{vx.x, s.z},
                                     v2f{vx.y, s.w}, v2f{ex, t.x}, v2f{ey, t.y}, v2f{Gx, t.z}, v2f{Gy, t.w}, ax, ay, bx, by);
                    if (in) { spx += ax.y - ax.x; spy += ay.y - ay.x; svx += bx.y - bx.x; svy += by.y - by.x; }
                    const float d2 = rx * rx + ry * ry, w2 = wx * wx + wy * wy;
                    const float a = d2 * fast_rsq(d2) - w2 * fast_rsq(fmaxf(w2, 1e-30f));
                    const bool near = in && !(a > 0.f && a * a > 1e-6f * d2 + lim);
                    const u64 m = __builtin_amdgcn_ballot_w64(near);
                    if (m) {
                        if (near) ring[(tail + mbcnt(m)) & 127] = (unsigned short)j;
                        tail += (unsigned)__builtin_popcountll(m);
                    }
                }
                while (tail - head >= (more ? 64u : 1u)) {
                    const unsigned n = min(64u, tail - head);
                    __builtin_amdgcn_wave_barrier();
                    if ((unsigned)lane < n) {
                        const int j = ring[(head + lane) & 127];
                        const float4 s = tile_pv[j], t = tile_eg[j];
                        const float rx = s.x - px.x, ry = s.y - px.y, wx = s.z - vx.x, wy = s.w - vx.y;
                        if (ucy_collision(rx, ry, wx, wy, P.r2)) {
                            float a1x, a1y, b1x, b1y, a0x, a0y, b0x, b0y;
                            mlapm_pair_grad(P, rx, ry, wx, wy, vx.x, vx.y, ex, ey, Gx, Gy, a1x, a1y, b1x, b1y, 1);      // x focal
                            mlapm_pair_grad(P, rx, ry, wx, wy, vx.x, vx.y, ex, ey, Gx, Gy, a0x, a0y, b0x, b0y, 0);
                            spx -= a1x - a0x; spy -= a1y - a0y; svx -= b1x - b0x; svy -= b1y - b0y;
                            mlapm_pair_grad(P, -rx, -ry, -wx, -wy, s.z, s.w, t.x, t.y, t.z, t.w, a1x, a1y, b1x, b1y, 1);  // o focal
                            mlapm_pair_grad(P, -rx, -ry, -wx, -wy, s.z, s.w, t.x, t.y, t.z, t.w, a0x, a0y, b0x, b0y, 0);
                            spx += a1x - a0x; spy += a1y - a0y; svx += b1x - b0x; svy += b1y - b0y;
                        }
                    }
                    head += n;
                }
                if (!more) break;
            }
            continue;
        }
        for (int j = lane; j < tn; j += 64) {
            const float4 s = tile_pv[j], t = tile_eg[j];
            float ax, ay, bx, by;
            // x focal, o source
            mlapm_pair_grad(P, s.x - px.x, s.y - px.y, s.z - vx.x, s.w - vx.y, vx.x, vx.y, ex, ey, Gx, Gy,
                            ax, ay, bx, by);
            spx -= ax; spy -= ay; svx -= bx; svy -= by;
            // o focal, x source
            mlapm_pair_grad(P, px.x - s.x, px.y - s.y, vx.x - s.z, vx.y - s.w, s.z, s.w, t.x, t.y, t.z, t.w,
                            ax, ay, bx, by);
            spx += ax; spy += ay; svx += bx; svy += by;
        }
    }
    spx = wave_sum(spx); spy = wave_sum(spy); svx = wave_sum(svx); svy = wave_sum(svy);
    if (has && lane == 0) {
        // desired force (v0 e - v)/tau, e = d/|d| (mlapm.py:21-22), and action = v + F dt (:57)
        const float v0x = v0[x];
        const float ge = Gx * ex + Gy * ey;
        g_v0[x] = ge / P.tau;
        float tdx = 0.f, tdy = 0.f;
        if (dn > 1e-12f) {
            tdx = v0x / P.tau * (Gx - ge * ex) / dn;
            tdy = v0x / P.tau * (Gy - ge * ey) / dn;
        } else {
            tdx = v0x / P.tau * Gx / 1e-12f; tdy = v0x / P.tau * Gy / 1e-12f;
        }
        g_dest[x] = make_float2(tdx, tdy);
        g_p[x] = make_float2(spx - tdx, spy - tdy);
        g_v[x] = make_float2(svx + ga.x - Gx / P.tau, svy + ga.y - Gy / P.tau);
    }
}

// ---- backward, every ordered pair evaluated ONCE (round 4) ----
// mlapm_bwd_kernel evaluates the ordered pair (f focal, s source) twice: in the wavefront that owns f and in the one that
// owns s (2 N^2 evaluations for N^2 pairs), because each wavefront wants both of its agent's sums in its own registers.
// Here a wavefront takes a 64-agent block X as focal agents and a 128-agent block O as sources: lane l keeps sources
// o_l, o_(l+64) (position, velocity: 8 registers) and the focal records (p, v, e, G: 8 registers) travel through the lanes
// by a one-lane wave rotation per step TOGETHER with their four focal-side sums, so after `steps` rotations every focal
// agent has met every source of the block: the source-side sums stay where the source is (packed adds), the focal-side
// sums ride with the focal agent (sum of the two packed elements), no atomics, no LDS, a fixed order.  The focal side of
// (X, O) and the source side of (X-group, O) leave as partial rows; mlapm_bwd_sys_reduce_kernel adds an agent's rows in
// row order and finishes with the desired-force terms.  `split` wavefronts share one (X, O) and do 64 / split steps each
// (the focal records start rotated by h * 64 / split lanes) -- more wavefronts for small scenes.
struct MlapmSysGeom {
    int nxb, nob, gx, nxg, split, npad, focal_rows, rows;   // 64-blocks, 128-blocks, X blocks per wavefront, X groups
};

// (old == src: every lane is written, and the compiler may keep the value in place -- with old = 0 it spends a zero fill
// and a copy around every rotation)
__device__ __forceinline__ int wave_rot1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x13C, 0xF, 0xF, false); }
__device__ __forceinline__ float wave_rot1(float v) { return __builtin_bit_cast(float, wave_rot1(__builtin_bit_cast(int, v))); }

template <int VARIANT>
__global__ __launch_bounds__(256) void mlapm_bwd_sys_kernel(
        const float2* __restrict__ g_action, const float2* __restrict__ p, const float2* __restrict__ v,
        const float2* __restrict__ dest, int N, MlapmParams P, float dt, MlapmSysGeom G, float4* __restrict__ part) {
    MlapmParams Q = P;
    Q.variant = VARIANT;                                    // the law's branches fold at compile time
    const int lane = threadIdx.x & 63;
    const int unit = uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int h = unit % G.split, u = unit / G.split;
    const int ob = u % G.nob, xg = u / G.nob;
    if (xg >= G.nxg) return;
    const float nan = __builtin_nanf("");
    // sources: an agent past N is a NaN position -- d2 > 0 is false, the pair contributes exact zeros
    const int o0 = ob * 128 + lane, o1 = o0 + 64;
    const float2 p0 = o0 < N ? p[o0] : make_float2(nan, nan), p1 = o1 < N ? p[o1] : make_float2(nan, nan);
    const float2 v0 = o0 < N ? v[o0] : make_float2(0.f, 0.f), v1 = o1 < N ? v[o1] : make_float2(0.f, 0.f);
    const v2f opx = {p0.x, p1.x}, opy = {p0.y, p1.y}, ovx = {v0.x, v1.x}, ovy = {v0.y, v1.y};
    v2f sax = {0.f, 0.f}, say = sax, sbx = sax, sby = sax;
    const int steps = 64 / G.split;
    for (int xb = xg * G.gx; xb < min(G.nxb, (xg + 1) * G.gx); ++xb) {
        int xi = xb * 64 + ((lane + h * steps) & 63);
        const bool in = xi < N;
        const float2 px = in ? p[xi] : make_float2(nan, nan), vx = in ? v[xi] : make_float2(0.f, 0.f);
        const float2 dx = in ? dest[xi] : make_float2(0.f, 0.f), ga = in ? g_action[xi] : make_float2(0.f, 0.f);
        float ex = dx.x - px.x, ey = dx.y - px.y;
        const float en = fmaxf(norm2(ex, ey), 1e-12f);
        ex /= en; ey /= en;
        float xpx = px.x, xpy = px.y, xvx = vx.x, xvy = vx.y, Gx = ga.x * dt, Gy = ga.y * dt;
        float fax = 0.f, fay = 0.f, fbx = 0.f, fby = 0.f;
        // a step: the focal records move on by one lane, meet the two sources of their new lane, and the focal-side sums
        // follow them inside the addition itself (sum' = rotated sum + this pair's terms: one DPP add per sum)
        for (int s = 0; s < steps; ++s) {
            xpx = wave_rot1(xpx); xpy = wave_rot1(xpy); xvx = wave_rot1(xvx); xvy = wave_rot1(xvy);
            ex = wave_rot1(ex); ey = wave_rot1(ey); Gx = wave_rot1(Gx); Gy = wave_rot1(Gy);
            xi = wave_rot1(xi);
            v2f ax, ay, bx, by;
            mlapm_pair_grad2(Q, opx - v2f{xpx, xpx}, opy - v2f{xpy, xpy}, ovx - v2f{xvx, xvx}, ovy - v2f{xvy, xvy},
                             v2f{xvx, xvx}, v2f{xvy, xvy}, v2f{ex, ex}, v2f{ey, ey}, v2f{Gx, Gx}, v2f{Gy, Gy}, ax, ay, bx, by);
            sax += ax; say += ay; sbx += bx; sby += by;
            fax = wave_rot1(fax) + (ax.x + ax.y); fay = wave_rot1(fay) + (ay.x + ay.y);
            fbx = wave_rot1(fbx) + (bx.x + bx.y); fby = wave_rot1(fby) + (by.x + by.y);
        }
        // the focal agent loses what its sources gain
        part[(size_t)(ob * G.split + h) * G.npad + xi] = make_float4(-fax, -fay, -fbx, -fby);
    }
    float4* row = part + (size_t)(G.focal_rows + xg * G.split + h) * G.npad;
    row[o0] = make_float4(sax.x, say.x, sbx.x, sby.x);
    row[o1] = make_float4(sax.y, say.y, sbx.y, sby.y);
}

// g_position / g_velocity = the partial rows of an agent in row order + the desired-force terms (the epilogue of
// mlapm_bwd_kernel); 64 agents x 4 row slices per workgroup, the slices added in slice order.
__global__ __launch_bounds__(256) void mlapm_bwd_sys_reduce_kernel(
        const float4* __restrict__ part, int rows, int npad, const float2* __restrict__ g_action,
        const float2* __restrict__ p, const float2* __restrict__ v, const float* __restrict__ v0,
        const float2* __restrict__ dest, int N, MlapmParams P, float dt, float2* __restrict__ g_p,
        float2* __restrict__ g_v, float* __restrict__ g_v0, float2* __restrict__ g_dest) {
    __shared__ float4 red[4][64];
    const int a = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + a;                      // < npad
    const int per = (rows + 3) / 4, r0 = sl * per, r1 = min(rows, r0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* q = part + (size_t)r0 * npad + x;
#pragma unroll 8
    for (int r = r0; r < r1; ++r, q += npad) {
        const float4 t = *q;
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    red[sl][a] = s;
    __syncthreads();
    if (sl != 0 || x >= N) return;
    for (int k = 1; k < 4; ++k) {
        const float4 t = red[k][a];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float2 px = p[x], dx = dest[x], ga = g_action[x];
    float ex = dx.x - px.x, ey = dx.y - px.y;
    const float dn = norm2(ex, ey), en = fmaxf(dn, 1e-12f);
    ex /= en; ey /= en;
    const float Gx = ga.x * dt, Gy = ga.y * dt;
    // desired force (v0 e - v)/tau, e = d/|d| (mlapm.py:21-22), and action = v + F dt (:57)
    const float v0x = v0[x];
    const float ge = Gx * ex + Gy * ey;
    g_v0[x] = ge / P.tau;
    float tdx, tdy;
    if (dn > 1e-12f) {
        tdx = v0x / P.tau * (Gx - ge * ex) / dn;
        tdy = v0x / P.tau * (Gy - ge * ey) / dn;
    } else {
        tdx = v0x / P.tau * Gx / 1e-12f; tdy = v0x / P.tau * Gy / 1e-12f;
    }
    g_dest[x] = make_float2(tdx, tdy);
    g_p[x] = make_float2(s.x - tdx, s.y - tdy);
    g_v[x] = make_float2(s.z + ga.x - Gx / P.tau, s.w + ga.y - Gy / P.tau);
}

// UCY on the once-per-pair backward: mlapm_bwd_sys_kernel<2> gives every ordered pair its flag-off term (g = 1); this
// kernel -- one wavefront per agent like mlapm_bwd_kernel -- finds the pairs the conservative distance test cannot rule out
// (see mlapm_pair2_ucy), gives them the exact predicate once and adds [flag on] - [flag off] for both roles of a flagged
// pair, then adds the agent's partial rows (lane = row, wave sum) and finishes like mlapm_bwd_sys_reduce_kernel.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mlapm_bwd_ucy_fix_kernel(
        const float4* __restrict__ part, int rows, int npad, const float2* __restrict__ g_action,
        const float2* __restrict__ p, const float2* __restrict__ v, const float* __restrict__ v0,
        const float2* __restrict__ dest, int N, MlapmParams P, float dt, float2* __restrict__ g_p,
        float2* __restrict__ g_v, float* __restrict__ g_v0, float2* __restrict__ g_dest) {
    __shared__ float4 tile_pv[kMlTile];                     // (px, py, vx, vy)
    __shared__ float4 tile_eg[kMlTile];                     // (ex, ey, Gx, Gy), G = dt * g_action
    __shared__ unsigned short ring_all[WAVES][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * WAVES + wave;
    const bool has = x < N;
    const float2 px = p[has ? x : 0], vx = v[has ? x : 0], dx = dest[has ? x : 0], ga = g_action[has ? x : 0];
    float ex = dx.x - px.x, ey = dx.y - px.y;
    const float dn = norm2(ex, ey), en = fmaxf(dn, 1e-12f);
    ex /= en; ey /= en;
    const float Gx = ga.x * dt, Gy = ga.y * dt;
    float spx = 0.f, spy = 0.f, svx = 0.f, svy = 0.f;
    unsigned short* ring = ring_all[uniform(wave)];
    const float lim = 1e-6f + P.r2 * P.r2;
    for (int base = 0; base < N; base += kMlTile) {
        const int tn = min(kMlTile, N - base);
        __syncthreads();
        for (int t = threadIdx.x; t < tn; t += WAVES * 64) {
            const float2 a = p[base + t], b = v[base + t], d = dest[base + t], g = g_action[base + t];
            float qx = d.x - a.x, qy = d.y - a.y;
            const float qn = fmaxf(norm2(qx, qy), 1e-12f);
            tile_pv[t] = make_float4(a.x, a.y, b.x, b.y);
            tile_eg[t] = make_float4(qx / qn, qy / qn, g.x * dt, g.y * dt);
        }
        __syncthreads();
        if (!has) continue;
        unsigned head = 0, tail = 0;
        for (int j0 = 0;; j0 += 64) {
            const bool more = j0 < tn;
            if (more) {
                const int j = j0 + lane;
                const bool in = j < tn;
                const float4 s = tile_pv[in ? j : 0];
                const float rx = s.x - px.x, ry = s.y - px.y, wx = s.z - vx.x, wy = s.w - vx.y;
                const float d2 = rx * rx + ry * ry, w2 = wx * wx + wy * wy;
                const float a = d2 * fast_rsq(d2) - w2 * fast_rsq(fmaxf(w2, 1e-30f));
                const bool near = in && !(a > 0.f && a * a > 1e-6f * d2 + lim);
                const u64 m = __builtin_amdgcn_ballot_w64(near);
                if (m) {
                    if (near) ring[(tail + mbcnt(m)) & 127] = (unsigned short)j;
                    tail += (unsigned)__builtin_popcountll(m);
                }
            }
            while (tail - head >= (more ? 64u : 1u)) {
                const unsigned n = min(64u, tail - head);
                __builtin_amdgcn_wave_barrier();
                if ((unsigned)lane < n) {
                    const int j = ring[(head + lane) & 127];
                    const float4 s = tile_pv[j], t = tile_eg[j];
                    const float rx = s.x - px.x, ry = s.y - px.y, wx = s.z - vx.x, wy = s.w - vx.y;
                    if (ucy_collision(rx, ry, wx, wy, P.r2)) {
                        float a1x, a1y, b1x, b1y, a0x, a0y, b0x, b0y;
                        mlapm_pair_grad(P, rx, ry, wx, wy, vx.x, vx.y, ex, ey, Gx, Gy, a1x, a1y, b1x, b1y, 1);      // x focal
                        mlapm_pair_grad(P, rx, ry, wx, wy, vx.x, vx.y, ex, ey, Gx, Gy, a0x, a0y, b0x, b0y, 0);
                        spx -= a1x - a0x; spy -= a1y - a0y; svx -= b1x - b0x; svy -= b1y - b0y;
                        mlapm_pair_grad(P, -rx, -ry, -wx, -wy, s.z, s.w, t.x, t.y, t.z, t.w, a1x, a1y, b1x, b1y, 1);  // o focal
                        mlapm_pair_grad(P, -rx, -ry, -wx, -wy, s.z, s.w, t.x, t.y, t.z, t.w, a0x, a0y, b0x, b0y, 0);
                        spx += a1x - a0x; spy += a1y - a0y; svx += b1x - b0x; svy += b1y - b0y;
                    }
                }
                head += n;
            }
            if (!more) break;
        }
    }
    if (!has) return;
    for (int r = lane; r < rows; r += 64) {
        const float4 t = part[(size_t)r * npad + x];
        spx += t.x; spy += t.y; svx += t.z; svy += t.w;
    }
    spx = wave_sum(spx); spy = wave_sum(spy); svx = wave_sum(svx); svy = wave_sum(svy);
    if (lane == 0) {
        const float v0x = v0[x];
        const float ge = Gx * ex + Gy * ey;
        g_v0[x] = ge / P.tau;
        float tdx, tdy;
        if (dn > 1e-12f) {
            tdx = v0x / P.tau * (Gx - ge * ex) / dn;
            tdy = v0x / P.tau * (Gy - ge * ey) / dn;
        } else {
            tdx = v0x / P.tau * Gx / 1e-12f; tdy = v0x / P.tau * Gy / 1e-12f;
        }
        g_dest[x] = make_float2(tdx, tdy);
        g_p[x] = make_float2(spx - tdx, spy - tdy);
        g_v[x] = make_float2(svx + ga.x - Gx / P.tau, svy + ga.y - Gy / P.tau);
    }
}

// Geometry of the once-per-pair backward for N agents, or rows == 0 when mlapm_bwd_kernel is the launch (scenes below 512
// agents: two launches cost what the pairs do -- 10.7 against 14.9 us at 512; UCY keeps its two-phase kernel).
static MlapmSysGeom mlapm_sys_geom(int N, int variant) {
    MlapmSysGeom G = {};
    static const int off = getenv("PIML_MLAPM_BWD_SYS") ? atoi(getenv("PIML_MLAPM_BWD_SYS")) == 0 : 0;
    static const int split_env = getenv("PIML_MLAPM_BWD_SPLIT") ? atoi(getenv("PIML_MLAPM_BWD_SPLIT")) : 0;
    static const int min_n = getenv("PIML_MLAPM_BWD_SYS_MIN") ? atoi(getenv("PIML_MLAPM_BWD_SYS_MIN")) : 512;
    static const bool two_phase_off = getenv("PIML_MLAPM_UCY_TWO_PHASE") && atoi(getenv("PIML_MLAPM_UCY_TWO_PHASE")) == 0;
    if (off || (variant == 2 && two_phase_off) || N < min_n) return G;
    G.nxb = (N + 63) / 64; G.nob = (N + 127) / 128; G.npad = G.nob * 128;
    // wavefronts per block pair (measured, GC law): 1 from 2048 pairs on (N = 4096: 41.8 us against 45.5 / 50.7 with 2 / 4),
    // 2 from 256 pairs (N = 2048: 17.0 against 23.8 / 18.9 with 1 / 4), 4 below (N = 1024: 11.2 against 15.0 with 2)
    const int pairs = G.nxb * G.nob;
    G.split = (split_env == 1 || split_env == 2 || split_env == 4) ? split_env : (pairs >= 2048 ? 1 : pairs >= 256 ? 2 : 4);
    // at most 8192 wavefronts: one X block per wavefront up to 8192 agents, more beyond (fewer partial rows)
    G.gx = 1;
    while ((long long)G.nob * ((G.nxb + G.gx - 1) / G.gx) * G.split > 8192 && G.gx < G.nxb) G.gx *= 2;
    G.nxg = (G.nxb + G.gx - 1) / G.gx;
    G.focal_rows = G.nob * G.split;
    G.rows = G.focal_rows + G.nxg * G.split;
    return G;
}

// ---- collision matrices (data.py:537-601) ----
// coll[s,i,j] = [|p_j - p_i| < thr] (- 1 on the diagonal when `minus_identity`), NaN -> 0.
__global__ void collision_pairs_kernel(const float2* __restrict__ p, int S, int N, float thr,
                                       int minus_identity, float* __restrict__ coll) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, s = blockIdx.z;
    if (j >= N) return;
    const float2 pi = p[(size_t)s * N + i], pj = p[(size_t)s * N + j];
    const float d = norm2(pj.x - pi.x, pj.y - pi.y);
    float c = d != d ? d : (d < thr ? 1.f : 0.f);          // NaN stays NaN until the final NaN -> 0
    if (minus_identity && i == j) c -= 1.f;
    coll[((size_t)s * N + i) * N + j] = c != c ? 0.f : c;
}

// The same for N % 4 == 0: four j per thread (one float4 store), kPairRows rows i per block with the four
// source points kept in registers -- 8192 instead of 262144 workgroups at N = 4096, S = 4.
constexpr int kPairRows = 8;
__global__ __launch_bounds__(256) void collision_pairs4_kernel(const float2* __restrict__ p, int S, int N, float thr,
                                                                int minus_identity, float* __restrict__ coll) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int i0 = blockIdx.y * kPairRows, s = blockIdx.z;
    if (j >= N) return;
    const float4 a = *reinterpret_cast<const float4*>(p + (size_t)s * N + j);        // p[j], p[j+1]
    const float4 b = *reinterpret_cast<const float4*>(p + (size_t)s * N + j + 2);    // p[j+2], p[j+3]
    const float px[4] = {a.x, a.z, b.x, b.z}, py[4] = {a.y, a.w, b.y, b.w};
    for (int r = 0; r < kPairRows; ++r) {
        const int i = i0 + r;
        if (i >= N) break;
        const float2 pi = p[(size_t)s * N + i];
        float c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = norm2(px[q] - pi.x, py[q] - pi.y);
            float v = d != d ? d : (d < thr ? 1.f : 0.f);
            if (minus_identity && i == j + q) v -= 1.f;
            c[q] = v != v ? 0.f : v;
        }
        *reinterpret_cast<float4*>(coll + ((size_t)s * N + i) * N + j) = make_float4(c[0], c[1], c[2], c[3]);
    }
}

// 3-D friends rule (data.py:573-591): pairs whose `base` sum over the leading dim exceeds 25
// are zeroed in every slice of `coll`.
// `base` may alias `coll` (the in-place 3-D rule without real_position): no __restrict__ on either.
__global__ void collision_friends3_kernel(float* coll, const float* base,
                                          int S_coll, int S_base, size_t NN) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= NN) return;
    float sum = 0.f;
    for (int s = 0; s < S_base; ++s) sum += base[(size_t)s * NN + q];
    if (!(sum <= 25.f))
        for (int s = 0; s < S_coll; ++s) coll[(size_t)s * NN + q] = 0.f;
}

// 4-D friends rule (data.py:592-598): pairs colliding in any of the first 4 frames of a channel
// are dropped for every frame of that channel.
__global__ void collision_friends4_kernel(float* __restrict__ coll, int C, int T, size_t NN) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (q >= NN) return;
    float sum = 0.f;
    for (int t = 0; t < min(T, 4); ++t) sum += coll[((size_t)c * T + t) * NN + q];
    if (sum > 0.f)
        for (int t = 0; t < T; ++t) coll[((size_t)c * T + t) * NN + q] = 0.f;
}

// Fused per-agent collision counts for a 3-D stack (S, N, 2): counts[h, s, i] =
// sum_j coll_h[s,i,j] * friends_h[i,j] for `nthr` thresholds, i.e.
// collision_detection(position, thr_h).sum(-1) without the (S,N,N) matrices (callers:
// simulators.py:708-724, metrics.py:16-26).  One wavefront per agent i, lane = j; the friends
// total over all slices is taken first, then the per-slice counts (pairs are recomputed, not
// stored).
__device__ __forceinline__ bool collide(const float2* __restrict__ p, int N, int s, int i, int j, float th) {
    if (j >= N || i == j) return false;                    // diagonal: 1 - 1 = 0 (or NaN -> 0)
    const float2 pi = p[(size_t)s * N + i], pj = p[(size_t)s * N + j];
    return norm2(pj.x - pi.x, pj.y - pi.y) < th;           // NaN compares false -> 0
}

// General path (more than 25 slices, e.g. the (t, N, 2) rollouts of the evaluation): one wavefront per agent i.
// Pass 1, per block of 64 partners j: the pair's collisions summed over all slices (the friends total).  Pass 2,
// only for the (few) partners that collide at all and are not friends: lanes take slices and bump a wave-private
// per-slice counter in LDS.  Every count is written exactly once with a plain store (no atomics, no zero fill).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void collision_counts_kernel(const float2* __restrict__ p, int S, int N,
                                                                       const float* __restrict__ thr, int nthr,
                                                                       float* __restrict__ counts) {
    extern __shared__ float cc_lds[];                       // WAVES x S per-slice counters
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * WAVES + wave;
    if (i >= N) return;                                     // wave-uniform; no block barrier below
    float* cnt = cc_lds + (size_t)wave * S;
    for (int h = 0; h < nthr; ++h) {
        const float th = thr[h];
        for (int s = lane; s < S; s += 64) cnt[s] = 0.f;
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int j = j0 + lane;
            int total = 0;
            for (int s = 0; s < S; ++s) total += collide(p, N, s, i, j, th) ? 1 : 0;
            u64 m = __builtin_amdgcn_ballot_w64(total > 0 && total <= 25);     // friends rule, data.py:587-591
            while (m) {
                const int jj = j0 + __builtin_ctzll(m);
                m &= m - 1;
                for (int s = lane; s < S; s += 64)
                    if (collide(p, N, s, i, jj, th)) cnt[s] += 1.f;
            }
        }
        for (int s = lane; s < S; s += 64) counts[((size_t)h * S + s) * N + i] = cnt[s];
    }
}

constexpr int kCollMaxThr = 4;

// General path, parallel form (more than 25 slices with a scratch buffer): two sweeps over the S x N x N pairs.
//   K1  totals[h][i][j] += number of slices in which the pair collides (integer atomics: exact in any order).  A wavefront
//       takes 8 agents i x 64 partners j x a chunk of 32 slices: a slice's 64 partner positions are fetched once for 8 x 64
//       pairs, and "norm2(...) < thr" is tested on the SQUARED distance against the smallest float whose correctly rounded
//       square root reaches thr (lt_cut2: the same predicate, no square root per pair);
//   K2  counts[h][s][i] = sum_j collide(s, i, j) and 0 < totals[h][i][j] <= 25 (friends rule, data.py:587-591), tiled the same way.
// Round 3's form evaluated one agent per wavefront with a square root per pair: 1.60 ms at S = 750, N = 1024.
constexpr int CC_CHUNK = 32;
constexpr int CC_IB = 8;

// smallest non-negative float x with sqrtf(x) >= th, i.e. "sqrtf(d2) < th"  <=>  "d2 < x" for the correctly rounded,
// monotone sqrtf both sides use; ok = false when th is outside the range where the short search below is exact (the caller
// then keeps the square root)
// (the flag travels in the RETURN value -- a negative result = not exact here: by reference it was a stack slot, 16 B of scratch
// per lane in every kernel that calls this deliberately not-inlined function)
__device__ __noinline__ float lt_cut2(float th) {
    if (!(th > 1e-18f && th < 1e18f)) return -1.f;
    float x = th * th;
#pragma unroll 1
    for (int it = 0; it < 8 && sqrtf(x) >= th; ++it) x = __uint_as_float(__float_as_uint(x) - 1u);       // now sqrtf(x) < th
    if (!(sqrtf(x) < th)) return -1.f;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const float nx = __uint_as_float(__float_as_uint(x) + 1u);
        if (sqrtf(nx) >= th) return nx;
        x = nx;
    }
    return -1.f;
}

// limits of the pair test for T thresholds: lim[h] is compared with d2 (use_sqrt false) or with sqrtf(d2) (true)
template <int T>
__device__ __forceinline__ bool cc_limits(const float* __restrict__ thr, float (&lim)[T]) {
    bool all = true;
    float c2[T];
#pragma unroll
    for (int h = 0; h < T; ++h) {
        c2[h] = lt_cut2(thr[h]);
        all = all && c2[h] >= 0.f;
    }
#pragma unroll
    for (int h = 0; h < T; ++h) lim[h] = all ? c2[h] : thr[h];
    return !all;
}

template <int T>
__global__ __launch_bounds__(256) void collision_totals_kernel(const float2* __restrict__ p, int S, int N,
                                                               const float* __restrict__ thr, int* __restrict__ totals) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = blockIdx.x * 64 + lane, i0 = (blockIdx.y * 4 + wave) * CC_IB;
    const int s0 = blockIdx.z * CC_CHUNK, s1 = min(S, s0 + CC_CHUNK);
    if (i0 >= N) return;
    float lim[T];
    const bool use_sqrt = cc_limits<T>(thr, lim);
    int cnt[CC_IB][T];
#pragma unroll
    for (int u = 0; u < CC_IB; ++u)
#pragma unroll
        for (int h = 0; h < T; ++h) cnt[u][h] = 0;
    const int jj = j < N ? j : 0;
#pragma unroll 1
    for (int s = s0; s < s1; ++s) {
        const float2* row = p + (size_t)s * N;
        const float2 pj = row[jj];
#pragma unroll
        for (int u = 0; u < CC_IB; ++u) {
            const float2 pi = row[min(i0 + u, N - 1)];                       // wave-uniform address
            float x = sq2(pj.x - pi.x, pj.y - pi.y);                         // NaN compares false
            if (use_sqrt) x = sqrtf(x);
#pragma unroll
            for (int h = 0; h < T; ++h) cnt[u][h] += x < lim[h] ? 1 : 0;
        }
    }
    if (j >= N) return;
#pragma unroll
    for (int u = 0; u < CC_IB; ++u) {
        const int i = i0 + u;
        if (i >= N || i == j) continue;
#pragma unroll
        for (int h = 0; h < T; ++h)
            if (cnt[u][h]) atomicAdd(totals + ((size_t)h * N + i) * N + j, cnt[u][h]);
    }
}

// K2: one wavefront per (slice s, 8 agents i), lanes stride over the partners j: the pairs of the slice once more (squared-
// distance test, a partner's position fetched once for the 8 agents); only a pair that collides reads its total.
// (A per-agent list of the colliding non-friend partners instead of this second sweep was built and measured: with
// temporally coherent rollouts the lists are a handful of entries, but uncorrelated frames make them the whole row --
// 5.1 ms at S = 750, N = 1024 -- so the sweep stays.)
template <int T>
__global__ __launch_bounds__(256) void collision_counts_from_totals_kernel(const float2* __restrict__ p, int S, int N,
                                                                           const float* __restrict__ thr,
                                                                           const int* __restrict__ totals,
                                                                           float* __restrict__ counts) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int ib = (N + CC_IB - 1) / CC_IB;
    const long long w = (long long)blockIdx.x * 4 + wave;           // (s, i block)
    if (w >= (long long)S * ib) return;
    const int s = (int)(w / ib), i0 = (int)(w - (long long)s * ib) * CC_IB;
    float lim[T];
    const bool use_sqrt = cc_limits<T>(thr, lim);
    const float2* row = p + (size_t)s * N;
    float2 pi[CC_IB];
#pragma unroll
    for (int u = 0; u < CC_IB; ++u) pi[u] = row[min(i0 + u, N - 1)];
    int cnt[CC_IB][T];
#pragma unroll
    for (int u = 0; u < CC_IB; ++u)
#pragma unroll
        for (int h = 0; h < T; ++h) cnt[u][h] = 0;
#pragma unroll 1
    for (int j = lane; j < N; j += 64) {
        const float2 pj = row[j];
#pragma unroll
        for (int u = 0; u < CC_IB; ++u) {
            const int i = i0 + u;
            float x = sq2(pj.x - pi[u].x, pj.y - pi[u].y);
            if (use_sqrt) x = sqrtf(x);
#pragma unroll
            for (int h = 0; h < T; ++h)
                if (x < lim[h] && i < N && i != j) {
                    const int t = totals[((size_t)h * N + i) * N + j];
                    cnt[u][h] += (t > 0 && t <= 25) ? 1 : 0;
                }
        }
    }
#pragma unroll
    for (int u = 0; u < CC_IB; ++u)
#pragma unroll
        for (int h = 0; h < T; ++h) {
            int c = cnt[u][h];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == 0 && i0 + u < N) counts[((size_t)h * S + s) * N + i0 + u] = (float)c;
        }
}

// Grid form of the many-slice path (round 4): a pair collides only within the largest threshold, so every frame is binned
// once -- one workgroup per frame counting-sorts its N positions into a 32 x 32 torus of c x c cells (c = 1.02 x the largest
// threshold; LDS atomics, one scan, one scatter) -- and an agent tests the 3 x 3 block around its cell: O(N x occupancy) pair
// tests per frame instead of N^2.  PASS 0 adds every colliding ordered pair to totals[h][i][j] (integer atomics: exact in any
// order); PASS 1 (a second launch) finds the same pairs again and counts those with 0 < total <= 25 (friends rule,
// data.py:587-591).  Coordinates beyond 1e5 cells (or a non-finite cell size) make the frame's workgroup walk all pairs.
// 750 x 1024: 0.97 ms (tiled two-sweep form above) -> see DESIGN_HISTORY.md 4.7.
constexpr int CG_DIM = 32, CG_CELLS = CG_DIM * CG_DIM;

template <int T, int PASS>
__global__ __launch_bounds__(256) void collision_grid_kernel(const float2* __restrict__ p, int S, int N,
                                                             const float* __restrict__ thr, int* __restrict__ totals,
                                                             float* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned cg_lds[];
    unsigned* const cnt = cg_lds;                                   // [CG_CELLS] counters -> start offsets
    unsigned* const cur = cg_lds + CG_CELLS;                        // [CG_CELLS] scatter cursors
    unsigned* const wsum = cur + CG_CELLS;                          // [4] wave totals + [1] overflow flag
    float2* const sp = reinterpret_cast<float2*>(wsum + 8);         // [N] sorted positions
    unsigned short* const si = reinterpret_cast<unsigned short*>(sp + N);   // [N] their agent indices
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2* row = p + (size_t)s * N;
    float lim[T];
    const bool use_sqrt = cc_limits<T>(thr, lim);
    float tmax = 0.f;
#pragma unroll
    for (int h = 0; h < T; ++h) tmax = fmaxf(tmax, thr[h]);
    const float c = tmax * 1.02f, inv_c = 1.f / c;
    const bool grid_ok = c > 1e-18f && c < 1e18f;
    for (int e = tid; e < CG_CELLS; e += 256) cnt[e] = 0;
    if (tid == 0) wsum[4] = grid_ok ? 0u : 1u;
    __syncthreads();
    auto cell_of = [&](float2 q, bool& far) -> int {
        const float fx = floorf(q.x * inv_c), fy = floorf(q.y * inv_c);
        far = !(fabsf(fx) < 1e5f && fabsf(fy) < 1e5f);
        return (((int)fy & (CG_DIM - 1)) << 5) | ((int)fx & (CG_DIM - 1));
    };
    bool any_far = false;
    for (int n = tid; n < N; n += 256) {
        const float2 q = row[n];
        if (q.x != q.x || q.y != q.y) continue;                     // absent agents collide with nobody
        bool far;
        const int ce = cell_of(q, far);
        any_far |= far;
        atomicAdd(&cnt[ce], 1u);
    }
    if (any_far) wsum[4] = 1u;
    __syncthreads();
    // exclusive scan of the 1024 counters: thread t owns cells 4 t .. 4 t + 3
    unsigned c4[4], tot = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { c4[u] = cnt[4 * tid + u]; tot += c4[u]; }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = (unsigned)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += y;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned run = inc - tot;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) { cnt[4 * tid + u] = run; cur[4 * tid + u] = run; run += c4[u]; }
    const bool overflow = wsum[4] != 0;
    const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    if (!overflow)
        for (int n = tid; n < N; n += 256) {
            const float2 q = row[n];
            if (q.x != q.x || q.y != q.y) continue;
            bool far;
            const unsigned pos = atomicAdd(&cur[cell_of(q, far)], 1u);
            sp[pos] = q; si[pos] = (unsigned short)n;
        }
    __syncthreads();
    // start(cell) = cnt[cell], end(cell) = cnt[cell + 1] (total for the last)
    auto test = [&](int i, int j, float2 pi, float2 pj, int (&acc)[T]) {
        if (i == j) return;
        float x = sq2(pj.x - pi.x, pj.y - pi.y);                    // NaN compares false
        if (use_sqrt) x = sqrtf(x);
#pragma unroll
        for (int h = 0; h < T; ++h)
            if (x < lim[h]) {
                if (PASS == 0) atomicAdd(totals + ((size_t)h * N + i) * N + j, 1);
                else {
                    const int t = totals[((size_t)h * N + i) * N + j];
                    acc[h] += (t > 0 && t <= 25) ? 1 : 0;
                }
            }
    };
    for (int i = tid; i < N; i += 256) {
        const float2 pi = row[i];
        int acc[T];
#pragma unroll
        for (int h = 0; h < T; ++h) acc[h] = 0;
        if (pi.x == pi.x && pi.y == pi.y) {
            if (overflow) {
                for (int j = 0; j < N; ++j) test(i, j, pi, row[j], acc);
            } else {
                const int fx = (int)floorf(pi.x * inv_c), fy = (int)floorf(pi.y * inv_c);
                for (int dy = -1; dy <= 1; ++dy) {
                    const int rowb = ((fy + dy) & (CG_DIM - 1)) << 5;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int ce = rowb | ((fx + dx) & (CG_DIM - 1));
                        const unsigned b0 = cnt[ce], b1 = ce + 1 < CG_CELLS ? cnt[ce + 1] : total;
                        for (unsigned q = b0; q < b1; ++q) test(i, (int)si[q], pi, sp[q], acc);
                    }
                }
            }
        }
        if (PASS == 1) {
#pragma unroll
            for (int h = 0; h < T; ++h) counts[((size_t)h * S + s) * N + i] = (float)acc[h];
        }
    }
}

// Fast path of collision_counts for stacks of at most 25 slices (the training rollouts: S = number
// of windows in the batch): a pair can then collide in at most 25 slices, so the friends rule
// (sum over slices <= 25, data.py:587-591) never removes anything and every slice is independent.
// One wavefront per (slice, agent); the slice's positions are staged in an LDS tile
// (structure-of-arrays, ds_read_b128), all thresholds are counted in one sweep, and "|r| < thr" is
// decided exactly in the squared domain (largest float whose correctly rounded sqrt is < thr).
constexpr int kCollTile = 4096;

__device__ __forceinline__ float sq_below(float thr) {       // max { y : sqrtf(y) < thr }, thr > 0
    if (!(thr > 0.f)) return -1.f;
    float y = thr * thr;
    while (sqrtf(y) >= thr && y > 0.f) y = __uint_as_float(__float_as_uint(y) - 1u);
    while (sqrtf(__uint_as_float(__float_as_uint(y) + 1u)) < thr) y = __uint_as_float(__float_as_uint(y) + 1u);
    return sqrtf(y) < thr ? y : -1.f;
}

// counts of ONE slice: agent i = this wave's, `ps` its slice's N points; out[h * hstride] = count for threshold h
template <int WAVES, int NTHR>
__device__ __forceinline__ void cc_fast_slice(const float2* __restrict__ ps, int N, int i, const float* __restrict__ thr,
                                              float* __restrict__ out, size_t hstride) {
    __shared__ __attribute__((aligned(16))) float tx[kCollTile], ty[kCollTile];
    const int lane = threadIdx.x & 63;
    const bool has = i < N;
    const float2 pi = ps[has ? i : 0];
    float cut[NTHR];
    int cnt[NTHR];
#pragma unroll
    for (int h = 0; h < NTHR; ++h) { cut[h] = sq_below(thr[h]); cnt[h] = 0; }
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int base = 0; base < N; base += kCollTile) {
        const int tn = min(kCollTile, N - base), tn_pad = (tn + 255) & ~255;
        __syncthreads();
        for (int t = threadIdx.x; t < tn_pad; t += WAVES * 64) {
            float2 q = make_float2(qnan, qnan);
            if (t < tn) q = ps[base + t];
            tx[t] = q.x; ty[t] = q.y;
        }
        __syncthreads();
        if (!has) continue;
        for (int j0 = 0; j0 < tn_pad; j0 += 256) {
            const float4 x = *reinterpret_cast<const float4*>(&tx[j0 + 4 * lane]);
            const float4 y = *reinterpret_cast<const float4*>(&ty[j0 + 4 * lane]);
            const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d2 = sq2(xs[u] - pi.x, ys[u] - pi.y);            // NaN compares false
                const bool other = base + j0 + 4 * lane + u != i;             // diagonal: 1 - 1 = 0
#pragma unroll
                for (int h = 0; h < NTHR; ++h) cnt[h] += (other && d2 <= cut[h]) ? 1 : 0;
            }
        }
    }
    if (!has) return;
#pragma unroll
    for (int h = 0; h < NTHR; ++h) {
        int c = cnt[h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) out[(size_t)h * hstride] = (float)c;
    }
}

template <int WAVES, int NTHR>
__global__ __launch_bounds__(WAVES * 64) void collision_counts_fast_kernel(
        const float2* __restrict__ p, int S, int N, const float* __restrict__ thr, int nthr,
        float* __restrict__ counts) {
    const int wave = threadIdx.x >> 6;
    const int bps = (N + WAVES - 1) / WAVES;
    const int s = blockIdx.x / bps;
    const int i = (blockIdx.x - s * bps) * WAVES + wave;
    cc_fast_slice<WAVES, NTHR>(p + (size_t)s * N, N, i, thr, counts + (size_t)s * N + (i < N ? i : 0), (size_t)S * N);
}

// The same for the frames of a training rollout in ONE launch (src/models/simulators.py:708-715, once per frame there): frame f
// is its own (S, N, 2) tensor, record f = counts[f] (nthr, S, N) is exactly what a launch on that frame alone writes.
struct CcFrames {
    const float2* p[32];
    int nframes;
};
template <int WAVES, int NTHR>
__global__ __launch_bounds__(WAVES * 64) void collision_counts_frames_kernel(CcFrames F, int S, int N, const float* __restrict__ thr,
                                                                             float* __restrict__ counts) {
    const int wave = threadIdx.x >> 6;
    const int bps = (N + WAVES - 1) / WAVES;
    const int fs = blockIdx.x / bps, f = fs / S, s = fs - f * S;
    const int i = (blockIdx.x - fs * bps) * WAVES + wave;
    const float2* ps = F.p[0];
#pragma unroll 1
    for (int q = 1; q < 32; ++q) ps = (q == f) ? F.p[q] : ps;            // (a by-value pointer table indexed at run time lands in scratch)
    cc_fast_slice<WAVES, NTHR>(ps + (size_t)s * N, N, i, thr, counts + ((size_t)f * NTHR * S + s) * N + (i < N ? i : 0), (size_t)S * N);
}

// calculate_collision_label (data.py:514-535): any tau in {0,.1,...,.9} with 0 != |dp + dv tau| < 0.5
__global__ void collision_label_kernel(const float* __restrict__ feat, size_t R, int ld, float* __restrict__ label) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* f = feat + r * ld;
    const float px = f[0], py = f[1], vx = f[2], vy = f[3];
    float hit = 0.f;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const float tau = __fmul_rn((float)t, 0.1f);        // torch.arange(10) * 0.1 in float32
        const float d = norm2(__fadd_rn(px, __fmul_rn(vx, tau)), __fadd_rn(py, __fmul_rn(vy, tau)));
        if (d < 0.5f && d != 0.f) hit = 1.f;
    }
    label[r] = hit;
}

// calc_acceleration (utils.py:31-100); version 0/1/2 = 'v0'/'v1'/'v2' incl. quirk Q10
__global__ void calc_acceleration_kernel(const float* __restrict__ rel, size_t R, int ld, int version, float A,
                                         float B, float Cc, float D, float ct, float st, float eps,
                                         float2* __restrict__ acc) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= R) return;
    const float dx = rel[q * ld], dy = rel[q * ld + 1];
    const float r = norm2(dx, dy) + eps;                    // utils.py:54-55
    const float ux = dx / r, uy = dy / r;
    float g, bx = ux, by = uy;
    if (version == 0) {
        g = A * expf(B * r);
    } else {
        const float vn = norm2(dx, dy) + eps;               // dv is read from the position slice (Q10)
        const float cs = (dx * dx + dy * dy) / r / vn;
        g = A * expf(B * r + Cc * cs + D * r * cs);
        if (version == 2) { bx = ct * ux - st * uy; by = st * ux + ct * uy; }
    }
    acc[q] = make_float2(-g * bx, -g * by);
}

// agents from which the 16-wave workgroups take over from the 4-wave ones (forward, rollout frame, two-role backward).  A
// rollout frame -- a launch that waits for the one before it -- GC law, 4 / 16 waves: 1024 agents 13.7 / 12.5 us, 2048
// 23.8 / 19.9, 4005 46.7 / 34.1 (512: 9.0 / 9.1); back-to-back forwards are level (tools/time_mlapm_wg.py).  Was 4096.
static int mlapm_wg16_min() {
    static const int v = getenv("PIML_MLAPM_WG16_MIN") ? atoi(getenv("PIML_MLAPM_WG16_MIN")) : 1024;
    return v;
}

static MlapmParams make_params(int variant, float tau, float A, float B, float Cc, float D, float theta_deg,
                               float radius, int skip_absent = 0) {
    MlapmParams P;
    P.skip_absent = skip_absent;
    static const bool two_phase_off = getenv("PIML_MLAPM_UCY_TWO_PHASE") && atoi(getenv("PIML_MLAPM_UCY_TWO_PHASE")) == 0;
    P.ucy_two_phase = two_phase_off ? 0 : 1;
    P.variant = variant; P.tau = tau; P.A = A; P.B = B; P.Cc = Cc; P.D = D;
    // the reference forms theta = sign * theta / 180 * pi in float32 (mlapm.py:34)
    const float th = theta_deg / 180.f * 3.14159265358979323846f;
    P.cth = cosf(th); P.sth = sinf(th);
    P.r2 = radius * 2.f;
    const float log2e = 1.4426950408889634f;
    P.B2 = B * log2e; P.C2 = Cc * log2e; P.D2 = D * log2e;
    return P;
}

}  // namespace piml

using namespace piml;

PIML_API int piml_mlapm_step_fwd(const float* position, const float* velocity, const float* desired_speed,
                                 const float* destination, int N, int variant, float tau, float A, float B,
                                 float C, float D, float theta_deg, float radius, float dt, int skip_absent,
                                 float* action, float* force, void* stream) {
    if (N < 0 || variant < 0 || variant > 2) return hipErrorInvalidValue;
    if (N == 0) return hipSuccess;
    if (!position || !velocity || !desired_speed || !destination || !action) return hipErrorInvalidValue;
    const MlapmParams P = make_params(variant, tau, A, B, C, D, theta_deg, radius, skip_absent);
    // big scenes: 16-wave workgroups (the whole (p,v) array is staged once per workgroup)
    if (N >= mlapm_wg16_min())
        hipLaunchKernelGGL(mlapm_fwd_kernel<16>, dim3((N + 15) / 16), dim3(1024), 0, as_stream(stream),
                           (const float2*)position, (const float2*)velocity, desired_speed,
                           (const float2*)destination, N, P, dt, (float2*)action, (float2*)force, MlapmRoll{});
    else
        hipLaunchKernelGGL(mlapm_fwd_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream),
                           (const float2*)position, (const float2*)velocity, desired_speed,
                           (const float2*)destination, N, P, dt, (float2*)action, (float2*)force, MlapmRoll{});
    return hipGetLastError();
}

PIML_API int piml_mlapm_rollout_step(float* traj_position, float* traj_velocity, const float* desired_speed,
                                     const float* destination, long long frames, int N, int variant, float tau, float A,
                                     float B, float C, float D, float theta_deg, float radius, float dt,
                                     long long* frame_counter, unsigned* done_counter, void* stream) {
    if (N < 0 || frames < 0 || variant < 0 || variant > 2) return hipErrorInvalidValue;
    if (N == 0 || frames < 2) return hipSuccess;
    if (!traj_position || !traj_velocity || !desired_speed || !destination || !frame_counter || !done_counter)
        return hipErrorInvalidValue;
    const MlapmParams P = make_params(variant, tau, A, B, C, D, theta_deg, radius, 1);
    const MlapmRoll Rr = {(float2*)traj_position, (float2*)traj_velocity, frame_counter, done_counter, frames, radius};
    if (N >= mlapm_wg16_min())
        hipLaunchKernelGGL((mlapm_fwd_kernel<16, true>), dim3((N + 15) / 16), dim3(1024), 0, as_stream(stream), nullptr, nullptr,
                           desired_speed, (const float2*)destination, N, P, dt, nullptr, nullptr, Rr);
    else
        hipLaunchKernelGGL((mlapm_fwd_kernel<4, true>), dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), nullptr, nullptr,
                           desired_speed, (const float2*)destination, N, P, dt, nullptr, nullptr, Rr);
    return hipGetLastError();
}

PIML_API int piml_mlapm_step_bwd(const float* g_action, const float* position, const float* velocity,
                                 const float* desired_speed, const float* destination, int N, int variant,
                                 float tau, float A, float B, float C, float D, float theta_deg, float radius,
                                 float dt, float* g_position, float* g_velocity, float* g_desired_speed,
                                 float* g_destination, void* stream) {
    if (N < 0 || variant < 0 || variant > 2) return hipErrorInvalidValue;
    if (N == 0) return hipSuccess;
    if (!g_action || !position || !velocity || !desired_speed || !destination || !g_position || !g_velocity ||
        !g_desired_speed || !g_destination)
        return hipErrorInvalidValue;
    const MlapmParams P = make_params(variant, tau, A, B, C, D, theta_deg, radius);
    if (N >= mlapm_wg16_min())
        hipLaunchKernelGGL(mlapm_bwd_kernel<16>, dim3((N + 15) / 16), dim3(1024), 0, as_stream(stream),
                           (const float2*)g_action, (const float2*)position, (const float2*)velocity, desired_speed,
                           (const float2*)destination, N, P, dt, (float2*)g_position, (float2*)g_velocity,
                           g_desired_speed, (float2*)g_destination);
    else
        hipLaunchKernelGGL(mlapm_bwd_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream),
                           (const float2*)g_action, (const float2*)position, (const float2*)velocity, desired_speed,
                           (const float2*)destination, N, P, dt, (float2*)g_position, (float2*)g_velocity,
                           g_desired_speed, (float2*)g_destination);
    return hipGetLastError();
}

PIML_API long long piml_mlapm_bwd_workspace_floats(int N, int variant) {
    if (N <= 0 || variant < 0 || variant > 2) return 0;
    const MlapmSysGeom G = mlapm_sys_geom(N, variant);
    return (long long)G.rows * G.npad * 4;
}

PIML_API int piml_mlapm_step_bwd_ws(const float* g_action, const float* position, const float* velocity,
                                    const float* desired_speed, const float* destination, int N, int variant,
                                    float tau, float A, float B, float C, float D, float theta_deg, float radius,
                                    float dt, float* g_position, float* g_velocity, float* g_desired_speed,
                                    float* g_destination, float* workspace, long long workspace_floats, void* stream) {
    if (N < 0 || variant < 0 || variant > 2 || workspace_floats < 0) return hipErrorInvalidValue;
    const MlapmSysGeom G = N > 0 ? mlapm_sys_geom(N, variant) : MlapmSysGeom{};
    const long long need = (long long)G.rows * G.npad * 4;
    if (need == 0 || !workspace || workspace_floats < need)     // small scenes or no workspace: the two-role kernel
        return need && workspace ? hipErrorInvalidValue
                                 : piml_mlapm_step_bwd(g_action, position, velocity, desired_speed, destination, N, variant, tau, A,
                                                       B, C, D, theta_deg, radius, dt, g_position, g_velocity, g_desired_speed,
                                                       g_destination, stream);
    if (!g_action || !position || !velocity || !desired_speed || !destination || !g_position || !g_velocity ||
        !g_desired_speed || !g_destination)
        return hipErrorInvalidValue;
    const MlapmParams P = make_params(variant, tau, A, B, C, D, theta_deg, radius);
    const int units = G.nob * G.nxg * G.split;
#define PIML_SYS_LAUNCH(V)                                                                                                      \
    hipLaunchKernelGGL(mlapm_bwd_sys_kernel<V>, dim3((units + 3) / 4), dim3(256), 0, as_stream(stream), (const float2*)g_action, \
                       (const float2*)position, (const float2*)velocity, (const float2*)destination, N, P, dt, G,               \
                       (float4*)workspace)
#define PIML_SYS_FINISH(KERNEL, GRID, BLOCK)                                                                                    \
    hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, as_stream(stream), (const float4*)workspace, G.rows, G.npad,         \
                       (const float2*)g_action, (const float2*)position, (const float2*)velocity, desired_speed,                \
                       (const float2*)destination, N, P, dt, (float2*)g_position, (float2*)g_velocity, g_desired_speed,         \
                       (float2*)g_destination)
    if (variant == 0) PIML_SYS_LAUNCH(0);
    else if (variant == 1) PIML_SYS_LAUNCH(1);
    else PIML_SYS_LAUNCH(2);
    if (variant != 2) PIML_SYS_FINISH(mlapm_bwd_sys_reduce_kernel, G.npad / 64, 256);
    else if (N >= mlapm_wg16_min()) PIML_SYS_FINISH(mlapm_bwd_ucy_fix_kernel<16>, (N + 15) / 16, 1024);
    else PIML_SYS_FINISH(mlapm_bwd_ucy_fix_kernel<4>, (N + 3) / 4, 256);
#undef PIML_SYS_LAUNCH
#undef PIML_SYS_FINISH
    return hipGetLastError();
}

PIML_API int piml_collision_matrix(const float* position, int S, int N, float threshold, int minus_identity,
                                   float* coll, void* stream) {
    if (S < 0 || N < 0 || N > 65535 || S > 65535) return hipErrorInvalidValue;
    if ((long)S * N == 0) return hipSuccess;
    if (!position || !coll) return hipErrorInvalidValue;
    if (N % 4 == 0)
        hipLaunchKernelGGL(collision_pairs4_kernel, dim3((N / 4 + 255) / 256, (N + kPairRows - 1) / kPairRows, S),
                           dim3(256), 0, as_stream(stream), (const float2*)position, S, N, threshold,
                           minus_identity, coll);
    else
        hipLaunchKernelGGL(collision_pairs_kernel, dim3((N + 255) / 256, N, S), dim3(256), 0, as_stream(stream),
                           (const float2*)position, S, N, threshold, minus_identity, coll);
    return hipGetLastError();
}

PIML_API int piml_collision_friends(float* coll, const float* base, int C, int T, int S_base, int N, void* stream) {
    if (C < 0 || T < 0 || N < 0 || S_base < 0 || C > 65535) return hipErrorInvalidValue;
    const size_t NN = (size_t)N * N;
    if ((size_t)C * T * NN == 0) return hipSuccess;
    if (!coll) return hipErrorInvalidValue;
    const unsigned gx = (unsigned)((NN + 255) / 256);
    if (C == 0 || T == 0) return hipSuccess;
    if (base) {   // 3-D rule: C == 1, T slices of coll; base has S_base slices
        hipLaunchKernelGGL(collision_friends3_kernel, dim3(gx), dim3(256), 0, as_stream(stream), coll, base, T,
                           S_base, NN);
    } else {      // 4-D rule per channel
        hipLaunchKernelGGL(collision_friends4_kernel, dim3(gx, C), dim3(256), 0, as_stream(stream), coll, C, T, NN);
    }
    return hipGetLastError();
}

PIML_API int piml_collision_counts(const float* position, int S, int N, const float* thresholds, int n_thresholds,
                                   float* counts, void* stream) {
    if (S < 0 || N < 0 || n_thresholds < 0) return hipErrorInvalidValue;
    if ((long)S * N * n_thresholds == 0) return hipSuccess;
    if (!position || !thresholds || !counts) return hipErrorInvalidValue;
    if (S <= 25 && n_thresholds <= kCollMaxThr) {          // independent slices: streaming fast path
        const int waves = (long)S * N >= 4096 ? 16 : 4;
        const unsigned grid = (unsigned)(S * ((N + waves - 1) / waves));
#define PIML_CC_LAUNCH(W, T)                                                                                   \
    hipLaunchKernelGGL((collision_counts_fast_kernel<W, T>), dim3(grid), dim3(W * 64), 0, as_stream(stream),   \
                       (const float2*)position, S, N, thresholds, n_thresholds, counts)
#define PIML_CC_BY_T(W)                                                                                        \
    switch (n_thresholds) { case 1: PIML_CC_LAUNCH(W, 1); break; case 2: PIML_CC_LAUNCH(W, 2); break;         \
                            case 3: PIML_CC_LAUNCH(W, 3); break; default: PIML_CC_LAUNCH(W, 4); break; }
        if (waves == 16) { PIML_CC_BY_T(16) } else { PIML_CC_BY_T(4) }
#undef PIML_CC_BY_T
#undef PIML_CC_LAUNCH
        return hipGetLastError();
    }
    // wave-private per-slice counters in LDS: 4 agents per block up to 4096 slices (64 KiB), one beyond (40 960 at most)
    if (S <= 4096) {
        hipLaunchKernelGGL(collision_counts_kernel<4>, dim3((N + 3) / 4), dim3(256), sizeof(float) * 4 * (size_t)S,
                           as_stream(stream), (const float2*)position, S, N, thresholds, n_thresholds, counts);
    } else if (S <= 40960) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(collision_counts_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 40960 * 4);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(collision_counts_kernel<1>, dim3(N), dim3(64), sizeof(float) * (size_t)S, as_stream(stream),
                           (const float2*)position, S, N, thresholds, n_thresholds, counts);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

PIML_API int piml_collision_counts_frames(const float* const* frames, int nframes, int S, int N, const float* thresholds,
                                         int n_thresholds, float* counts, void* stream) {
    if (nframes < 0 || nframes > 32 || S < 0 || S > 25 || N < 0 || n_thresholds < 1 || n_thresholds > kCollMaxThr) return hipErrorInvalidValue;
    if ((long)nframes * S * N == 0) return hipSuccess;
    if (!frames || !thresholds || !counts) return hipErrorInvalidValue;
    CcFrames F = {};
    F.nframes = nframes;
    for (int f = 0; f < nframes; ++f) {
        if (!frames[f]) return hipErrorInvalidValue;
        F.p[f] = (const float2*)frames[f];
    }
    const int waves = (long)nframes * S * N >= 4096 ? 16 : 4;
    const unsigned grid = (unsigned)(nframes * S * ((N + waves - 1) / waves));
#define PIML_CCF_LAUNCH(W, T)                                                                                        \
    hipLaunchKernelGGL((collision_counts_frames_kernel<W, T>), dim3(grid), dim3(W * 64), 0, as_stream(stream), F, S, N, \
                       thresholds, counts)
#define PIML_CCF_BY_T(W)                                                                                             \
    switch (n_thresholds) { case 1: PIML_CCF_LAUNCH(W, 1); break; case 2: PIML_CCF_LAUNCH(W, 2); break;              \
                            case 3: PIML_CCF_LAUNCH(W, 3); break; default: PIML_CCF_LAUNCH(W, 4); break; }
    if (waves == 16) { PIML_CCF_BY_T(16) } else { PIML_CCF_BY_T(4) }
#undef PIML_CCF_BY_T
#undef PIML_CCF_LAUNCH
    return hipGetLastError();
}

PIML_API int piml_collision_counts_scratch(const float* position, int S, int N, const float* thresholds,
                                           int n_thresholds, int* totals_zeroed, float* counts, void* stream) {
    if (S < 0 || N < 0 || n_thresholds < 0 || n_thresholds > kCollMaxThr) return hipErrorInvalidValue;
    if ((long)S * N * n_thresholds == 0) return hipSuccess;
    if (!position || !thresholds || !counts || !totals_zeroed) return hipErrorInvalidValue;
    const dim3 g1((N + 63) / 64, (N + 4 * CC_IB - 1) / (4 * CC_IB), (S + CC_CHUNK - 1) / CC_CHUNK);
    const long long waves = (long long)S * ((N + CC_IB - 1) / CC_IB);
    const dim3 g2((unsigned)((waves + 3) / 4));
    const float2* pp = (const float2*)position;
#define PIML_CC2(T)                                                                                                              \
    hipLaunchKernelGGL(collision_totals_kernel<T>, g1, dim3(256), 0, as_stream(stream), pp, S, N, thresholds, totals_zeroed);   \
    hipLaunchKernelGGL(collision_counts_from_totals_kernel<T>, g2, dim3(256), 0, as_stream(stream), pp, S, N, thresholds,       \
                       totals_zeroed, counts)
    switch (n_thresholds) {
        case 1: PIML_CC2(1); break;
        case 2: PIML_CC2(2); break;
        case 3: PIML_CC2(3); break;
        default: PIML_CC2(4); break;
    }
#undef PIML_CC2
    return hipGetLastError();
}

PIML_API int piml_collision_counts_grid(const float* position, int S, int N, const float* thresholds, int n_thresholds,
                                        int* totals_zeroed, float* counts, void* stream) {
    if (S < 0 || N < 0 || n_thresholds < 0 || n_thresholds > kCollMaxThr || N > 8192) return hipErrorInvalidValue;
    if ((long)S * N * n_thresholds == 0) return hipSuccess;
    if (!position || !thresholds || !counts || !totals_zeroed) return hipErrorInvalidValue;
    const size_t lds = (2 * piml::CG_CELLS + 8) * 4 + (size_t)N * 10 + 16;
    const float2* pp = (const float2*)position;
#define PIML_CG(T)                                                                                                                \
    {                                                                                                                             \
        static int attr = -1;                                                                                                     \
        if (attr < 0) {                                                                                                           \
            attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(piml::collision_grid_kernel<T, 0>),                     \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (2 * piml::CG_CELLS + 8) * 4 + 8192 * 10 + 16); \
            if (!attr)                                                                                                            \
                attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(piml::collision_grid_kernel<T, 1>),                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (2 * piml::CG_CELLS + 8) * 4 + 8192 * 10 + 16); \
        }                                                                                                                         \
        if (attr) return attr;                                                                                                    \
        hipLaunchKernelGGL((piml::collision_grid_kernel<T, 0>), dim3(S), dim3(256), lds, as_stream(stream), pp, S, N, thresholds, \
                           totals_zeroed, counts);                                                                                \
        hipLaunchKernelGGL((piml::collision_grid_kernel<T, 1>), dim3(S), dim3(256), lds, as_stream(stream), pp, S, N, thresholds, \
                           totals_zeroed, counts);                                                                                \
    }
    switch (n_thresholds) {
        case 1: PIML_CG(1) break;
        case 2: PIML_CG(2) break;
        case 3: PIML_CG(3) break;
        default: PIML_CG(4) break;
    }
#undef PIML_CG
    return hipGetLastError();
}

PIML_API int piml_collision_label(const float* ped_features, size_t rows, int row_stride, float* label, void* stream) {
    if (row_stride < 4) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!ped_features || !label) return hipErrorInvalidValue;
    hipLaunchKernelGGL(collision_label_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream),
                       ped_features, rows, row_stride, label);
    return hipGetLastError();
}

PIML_API int piml_calc_acceleration(const float* relative_data, size_t rows, int row_stride, int version, float A,
                                    float B, float C, float D, float theta, float eps, float* acc, void* stream) {
    if (row_stride < 2 || version < 0 || version > 2) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!relative_data || !acc) return hipErrorInvalidValue;
    hipLaunchKernelGGL(calc_acceleration_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       as_stream(stream), relative_data, rows, row_stride, version, A, B, C, D,
                       (float)cos((double)theta), (float)sin((double)theta), eps, (float2*)acc);
    return hipGetLastError();
}

// ---- HIP-event timer usable inside stream capture (bench.py's live kernel timing) ----
PIML_API int piml_timer_create(void** event) {
    if (!event) return hipErrorInvalidValue;
    hipEvent_t e;
    hipError_t err = hipEventCreate(&e);
    *event = (void*)e;
    return err;
}

PIML_API int piml_timer_record(void* event, void* stream) {
    if (!event) return hipErrorInvalidValue;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipError_t err = hipStreamIsCapturing(as_stream(stream), &st);
    if (err != hipSuccess) return err;
    // while capturing, the record must become an external event-record node of the graph
    return hipEventRecordWithFlags((hipEvent_t)event, as_stream(stream),
                                   st == hipStreamCaptureStatusActive ? hipEventRecordExternal : hipEventRecordDefault);
}

PIML_API int piml_timer_elapsed_ms(void* start, void* stop, float* ms) {
    if (!start || !stop || !ms) return hipErrorInvalidValue;
    hipError_t err = hipEventSynchronize((hipEvent_t)stop);
    if (err != hipSuccess) return err;
    return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}

PIML_API int piml_timer_destroy(void* event) {
    return event ? hipEventDestroy((hipEvent_t)event) : hipSuccess;
}

// ---- fused per-frame integrator epilogue of the inference rollout (simulators.py:596-639) ----
namespace piml {

struct RolloutArgs {
    float2 *p, *v, *a, *dest; long long* dest_idx; float* hist; int hist_w;
    const float2* a_next; const float2* waypoints; int D, wp_per_slice; const long long* dest_num;
    const float2 *pos_s, *vel_s, *acc_s, *dest_s; const long long* dest_idx_s; const float* selff_s; int F;
    const unsigned char* new_flag;
    float2 *p_res, *v_res, *a_res; float* mask_new;
    float* selff_out; const float* desired_speed;
    const long long* t_dev; int C, T, N; float dt; int remove_arrived;
};

// One thread per (slice, agent): record frame t, lagged explicit Euler (quirk Q6), waypoint switch
// at 0.5 m, leave-scene NaN, injection of agents entering at frame t+1 from the ground truth,
// history shift, and the (hist, a, v0) columns of the next self_features row.
__device__ __forceinline__ void rollout_step_agent(const RolloutArgs& A, long long g, float2 an) {
    const int c = (int)(g / A.N), i = (int)(g - (long long)c * A.N);
    const long long t = *A.t_dev;
    const long long tn = t + 1, tc = tn < A.T ? tn : A.T - 1;
    const size_t ft = ((size_t)c * A.T + t) * A.N + i;           // (c, t, i) in the time series
    const size_t fn = ((size_t)c * A.T + tc) * A.N + i;          // (c, t+1, i), clamped
    const float2 p = A.p[g], v = A.v[g], a = A.a[g], d = A.dest[g];
    A.p_res[ft] = p; A.v_res[ft] = v; A.a_res[ft] = a;           // :596-600
    if (!(p.x != p.x)) A.mask_new[ft] = 1.f;

    float2 vn = make_float2(__fadd_rn(v.x, __fmul_rn(a.x, A.dt)), __fadd_rn(v.y, __fmul_rn(a.y, A.dt)));   // :603
    float2 pn = make_float2(__fadd_rn(p.x, __fmul_rn(v.x, A.dt)), __fadd_rn(p.y, __fmul_rn(v.y, A.dt)));   // :604
    long long idx = A.dest_idx[g];
    if (norm2(p.x - d.x, p.y - d.y) < 0.5f) idx += 1;            // :608-609
    const bool gone = idx > A.dest_num[i] - 1;
    if (gone) {
        if (A.remove_arrived) { const float qn = __uint_as_float(0x7fc00000u); pn = make_float2(qn, qn); }   // :611
        idx -= 1;                                                // :613
    }
    float2 dn = A.waypoints[((size_t)(A.wp_per_slice ? c : 0) * A.D + idx) * A.N + i];   // :614-616

    const int hw = A.hist_w;
    float* h = A.hist + (size_t)g * hw;
    float* so = A.selff_out + (size_t)g * A.F;
    const bool enter = A.new_flag[((size_t)c * (A.T + 1) + tn) * A.N + i] != 0;        // :629-639
    if (enter) {
        pn = A.pos_s[fn]; vn = A.vel_s[fn]; an = A.acc_s[fn]; dn = A.dest_s[fn]; idx = A.dest_idx_s[fn];
        const float* hs = A.selff_s + fn * A.F + 2;
        for (int q = 0; q < hw; ++q) { h[q] = hs[q]; so[2 + q] = hs[q]; }
    } else {
        for (int q = 0; q + 2 < hw; ++q) h[q] = h[q + 2];        // :624-626
        h[hw - 2] = vn.x; h[hw - 1] = vn.y;
        for (int q = 0; q < hw; ++q) so[2 + q] = h[q];
    }
    so[2 + hw] = an.x; so[3 + hw] = an.y; so[4 + hw] = A.desired_speed[g];             // :651
    A.p[g] = pn; A.v[g] = vn; A.a[g] = an; A.dest[g] = dn; A.dest_idx[g] = idx;
}

__global__ void rollout_step_kernel(const RolloutArgs A) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)A.C * A.N) return;
    rollout_step_agent(A, g, A.a_next[g]);
}

// The same step with the bottleneck variants' network epilogue in front of it (the arithmetic of pinnsf_epilogue_ksum_fwd_kernel,
// mlpglue.hip: sum of the agent's kp / ko per-neighbour predictions + the desired-force term of its self_features row): an
// inference frame of `pinnsf_bm` / `pinnsf_bottleneck` has one launch for both.  `sf` IS the buffer the step rewrites
// (A.selff_out): a thread reads its own row before it writes it.
__global__ __launch_bounds__(256) void rollout_step_ksum_kernel(const RolloutArgs A, const float2* __restrict__ pred_ped, int kp,
                                                                const float2* __restrict__ pred_obs, int ko, const float* sf, float tau) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)A.C * A.N) return;
    const float* s = sf + g * 7;
    const float dx = s[0], dy = s[1], vx = s[2], vy = s[3], v0 = s[6];
    float t = norm2(dx, dy);
    t = (t == 0.f) ? t + 0.1f : t;
    float2 a = make_float2(0.f, 0.f), o = make_float2(0.f, 0.f);
    for (int i = 0; i < kp; ++i) { const float2 v = pred_ped[g * kp + i]; a.x += v.x; a.y += v.y; }
    if (pred_obs) {
        for (int i = 0; i < ko; ++i) { const float2 v = pred_obs[g * ko + i]; o.x += v.x; o.y += v.y; }
        a.x += o.x;
        a.y += o.y;
    }
    rollout_step_agent(A, g, make_float2(a.x + (v0 * (dx / t) - vx) / tau, a.y + (v0 * (dy / t) - vy) / tau));
}

// ---- differentiable frame step of the fine-tuning rollout (src/models/simulators.py:741-769) ----
struct TrainStepArgs {
    const float2 *p, *v, *a, *a_pred, *dest; const long long* dest_idx;
    const float2* waypoints; int D, wp_per_slice; const long long* dest_num;
    const unsigned char* new_flag;                       // (C, T, N) or NULL
    const float2 *pos_s, *vel_s, *acc_s, *dest_s; const long long* dest_idx_s;
    float2 *p_out, *v_out, *a_out, *dest_out; long long* dest_idx_out; int* nan_flag; unsigned char* zero_mask;
    int C, T, N, t_next; float dt;
    float2* p_copy; long long p_copy_cstride;            // optional: the INPUT position once more, slices p_copy_cstride float2 apart
};

// One thread per (slice, agent): lagged explicit Euler, waypoint switch at 0.5 m (nobody is removed
// in the training rollout), then agents entering at frame t_next are re-initialised from the series.
__device__ __forceinline__ void train_step_agent(const TrainStepArgs& A, long long g, int c, int i, float2 an) {
    const float2 p = A.p[g], v = A.v[g], a = A.a[g], d = A.dest[g];
    if (A.p_copy) A.p_copy[(size_t)c * A.p_copy_cstride + i] = p;      // frame t of the (C, T, N, 2) positions the rollout loss reads
    if (A.nan_flag && (an.x != an.x || an.y != an.y)) atomicOr(A.nan_flag, 1);           // :745
    float2 vn = make_float2(__fadd_rn(v.x, __fmul_rn(a.x, A.dt)), __fadd_rn(v.y, __fmul_rn(a.y, A.dt)));   // :741
    float2 pn = make_float2(__fadd_rn(p.x, __fmul_rn(v.x, A.dt)), __fadd_rn(p.y, __fmul_rn(v.y, A.dt)));   // :742
    long long idx = A.dest_idx[g];
    if (norm2(p.x - d.x, p.y - d.y) < 0.5f) idx += 1;                                    // :748-750
    if (idx > A.dest_num[i] - 1) idx -= 1;                                               // :751-752
    float2 dn = A.waypoints[((size_t)(A.wp_per_slice ? c : 0) * A.D + idx) * A.N + i];   // :753-754
    if (A.new_flag && A.t_next < A.T) {
        const size_t fn = ((size_t)c * A.T + A.t_next) * A.N + i;
        if (A.new_flag[fn]) {                                                            // :762-769
            pn = A.pos_s[fn]; vn = A.vel_s[fn]; an = A.acc_s[fn]; dn = A.dest_s[fn]; idx = A.dest_idx_s[fn];
        }
    }
    if (A.zero_mask) {       // the NaN -> 0 that get_relative_features applies to v, a in place (data.py:483-484)
        unsigned m = 0;
        if (vn.x != vn.x) { vn.x = 0.f; m |= 1u; }
        if (vn.y != vn.y) { vn.y = 0.f; m |= 2u; }
        if (an.x != an.x) { an.x = 0.f; m |= 4u; }
        if (an.y != an.y) { an.y = 0.f; m |= 8u; }
        A.zero_mask[g] = (unsigned char)m;
    }
    A.p_out[g] = pn; A.v_out[g] = vn; A.a_out[g] = an; A.dest_out[g] = dn; A.dest_idx_out[g] = idx;
}

__global__ void train_step_fwd_kernel(const TrainStepArgs A) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)A.C * A.N) return;
    const int c = (int)(g / A.N), i = (int)(g - (long long)c * A.N);
    train_step_agent(A, g, c, i, A.a_pred[g]);
}

// ---- the step with the model's TAIL in front of it (round 6): in the training rollout the network's output is channelled,
// (C, N, .), and the reference's desired-force term takes its norm over the AGENT axis (quirk Q2, src/models/model.py:1290 as
// evaluated at src/models/simulators.py:701) -- a reduction over a slice's agents, which kept this tail a launch of its own
// (pinnsf_epilogue_agentnorm_fwd_kernel, mlpglue.hip) between the network and the step.  One workgroup per slice does both: the
// two norms (the SAME block reduction, so the same bits), then per agent the prediction = sum of the kp (+ ko) per-neighbour
// outputs + (v0 d / t - v) / tau and the step on it.  Backward likewise: the step's gradients, then the tail's (sum_n g_e d over
// the slice, g_self, the broadcast to the neighbour rows). ----
struct TailArgs {
    const float2 *acc_ped, *acc_obs;     // (C, N, kp, 2) / (C, N, ko, 2) or NULL
    int kp, ko;
    const float* sf;                     // (C, N, 7)
    float tau;
};

__device__ __forceinline__ float2 tail_block_sum2(float2 v, float2* sh) {      // = block_sum2 of mlpglue.hip (256 threads)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float2 s = make_float2(0.f, 0.f);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        s.x += sh[w].x;
        s.y += sh[w].y;
    }
    return s;
}

__global__ __launch_bounds__(256) void train_step_tail_fwd_kernel(const TrainStepArgs A, const TailArgs E) {
    __shared__ float2 sh[4];
    const int c = blockIdx.x, N = A.N;
    const size_t base = (size_t)c * N;
    float2 sq = make_float2(0.f, 0.f);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = E.sf + (base + n) * 7;
        sq.x += s[0] * s[0];
        sq.y += s[1] * s[1];
    }
    sq = tail_block_sum2(sq, sh);
    float tx = sqrtf(sq.x), ty = sqrtf(sq.y);
    tx = (tx == 0.f) ? tx + 0.1f : tx;
    ty = (ty == 0.f) ? ty + 0.1f : ty;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = E.sf + (base + n) * 7;
        float2 a = make_float2(0.f, 0.f);
        for (int i = 0; i < E.kp; ++i) {
            const float2 q = E.acc_ped[(base + n) * E.kp + i];
            a.x += q.x; a.y += q.y;
        }
        if (E.acc_obs) {
            float2 o = make_float2(0.f, 0.f);
            for (int i = 0; i < E.ko; ++i) {
                const float2 q = E.acc_obs[(base + n) * E.ko + i];
                o.x += q.x; o.y += q.y;
            }
            a.x += o.x;
            a.y += o.y;
        }
        train_step_agent(A, (long long)(base + n), c, n,
                         make_float2(a.x + (s[6] * (s[0] / tx) - s[2]) / E.tau, a.y + (s[6] * (s[1] / ty) - s[3]) / E.tau));
    }
}

// keep = agent not re-initialised at t_next:  g_p = keep g_p',  g_v = keep (g_v' + dt g_p'),
// g_a = keep dt g_v',  g_a_pred = keep g_a'.
struct TrainBwdArgs {
    const float2 *gp_o, *gv_o, *ga_o, *g6;
    const unsigned char *new_flag, *zero_mask;
    int C, T, N, t_next;
    float dt;
    float2 *gp, *gv, *ga, *ga_pred;
    const float2* gp_in;
    long long gp_in_cstride, gp_o_cstride;
};

// one agent of train_step_bwd_kernel; returns d/d(a_pred)
__device__ __forceinline__ float2 train_step_bwd_agent(const float2* __restrict__ gp_o, const float2* __restrict__ gv_o,
                                                       const float2* __restrict__ ga_o, const float2* __restrict__ g6,
                                                       const unsigned char* __restrict__ new_flag,
                                                       const unsigned char* __restrict__ zero_mask, int T, int N, int t_next, float dt,
                                                       float2* __restrict__ gp, float2* __restrict__ gv, float2* __restrict__ ga,
                                                       float2* __restrict__ ga_pred, const float2* __restrict__ gp_in,
                                                       long long gp_in_cstride, long long gp_o_cstride, long long g, int c, int i) {
    bool keep = true;
    if (new_flag && t_next < T) keep = new_flag[((size_t)c * T + t_next) * N + i] == 0;
    const float2 z = make_float2(0.f, 0.f);
    // (gp_o_cstride != 0: g_position_out is C slices of (N, 2) that far apart -- a time slice of the loss's gradient as it stands)
    float2 a = (keep && gp_o) ? (gp_o_cstride ? gp_o[(size_t)c * gp_o_cstride + i] : gp_o[g]) : z;
    float2 b = (keep && gv_o) ? gv_o[g] : z, e = (keep && ga_o) ? ga_o[g] : z;
    if (keep && g6) {        // the features' share of d/d(p', v', a'), interleaved (C, N, 6)
        const float2 q0 = g6[3 * g], q1 = g6[3 * g + 1], q2 = g6[3 * g + 2];
        a.x += q0.x; a.y += q0.y; b.x += q1.x; b.y += q1.y; e.x += q2.x; e.y += q2.y;
    }
    if (zero_mask) {         // components that were NaN and zeroed in the forward pass pass no gradient
        const unsigned m = zero_mask[g];
        if (m & 1u) b.x = 0.f;
        if (m & 2u) b.y = 0.f;
        if (m & 4u) e.x = 0.f;
        if (m & 8u) e.y = 0.f;
    }
    // gp_in: a gradient that arrives on the INPUT position itself (the loss reads the frame's input through an alias output of
    // the frame's autograd node): added to g_p only -- slices (N, 2) contiguous, gp_in_cstride float2 between the slices
    float2 ain = z;
    if (gp_in) ain = gp_in[(size_t)c * gp_in_cstride + i];
    if (gp) gp[g] = make_float2(a.x + ain.x, a.y + ain.y);
    if (gv) gv[g] = make_float2(b.x + dt * a.x, b.y + dt * a.y);
    if (ga) ga[g] = make_float2(dt * b.x, dt * b.y);
    if (ga_pred) ga_pred[g] = e;
    return e;
}

__global__ void train_step_bwd_kernel(const float2* __restrict__ gp_o, const float2* __restrict__ gv_o,
                                      const float2* __restrict__ ga_o, const float2* __restrict__ g6,
                                      const unsigned char* __restrict__ new_flag,
                                      const unsigned char* __restrict__ zero_mask, int C, int T, int N, int t_next,
                                      float dt, float2* __restrict__ gp,
                                      float2* __restrict__ gv, float2* __restrict__ ga,
                                      float2* __restrict__ ga_pred, const float2* __restrict__ gp_in = nullptr,
                                      long long gp_in_cstride = 0, long long gp_o_cstride = 0) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)C * N) return;
    const int c = (int)(g / N), i = (int)(g - (long long)c * N);
    train_step_bwd_agent(gp_o, gv_o, ga_o, g6, new_flag, zero_mask, T, N, t_next, dt, gp, gv, ga, ga_pred, gp_in, gp_in_cstride,
                         gp_o_cstride, g, c, i);
}

// the step's backward and the tail's (pinnsf_epilogue_agentnorm_bwd_kernel's arithmetic + the broadcast to the neighbour rows),
// one workgroup per slice; B.ga_pred (C, N, 2) is written first and read back by the same thread: it is d/d(prediction), i.e.
// the gradient of every summand of the tail's sum
__global__ __launch_bounds__(256) void train_step_tail_bwd_kernel(const TrainBwdArgs B, const float* __restrict__ sf, float tau, int kp, int ko,
                                                                  float2* __restrict__ g_ped, float2* __restrict__ g_obs,
                                                                  float* __restrict__ g_self) {
    __shared__ float2 sh[4];
    const int c = blockIdx.x, N = B.N;
    const size_t base = (size_t)c * N;
    float2 sq = make_float2(0.f, 0.f), dot = make_float2(0.f, 0.f);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float2 g = train_step_bwd_agent(B.gp_o, B.gv_o, B.ga_o, B.g6, B.new_flag, B.zero_mask, B.T, N, B.t_next, B.dt, B.gp, B.gv, B.ga,
                                              B.ga_pred, B.gp_in, B.gp_in_cstride, B.gp_o_cstride, (long long)(base + n), c, n);
        const float* s = sf + (base + n) * 7;
        sq.x += s[0] * s[0];
        sq.y += s[1] * s[1];
        dot.x += (g.x * s[6] / tau) * s[0];
        dot.y += (g.y * s[6] / tau) * s[1];
    }
    sq = tail_block_sum2(sq, sh);
    dot = tail_block_sum2(dot, sh);
    const float nx = sqrtf(sq.x), ny = sqrtf(sq.y);
    const float tx = (nx == 0.f) ? nx + 0.1f : nx, ty = (ny == 0.f) ? ny + 0.1f : ny;
    const float gtx = -dot.x / (tx * tx), gty = -dot.y / (ty * ty);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = sf + (base + n) * 7;
        const float2 g = B.ga_pred[base + n];
        if (g_self) {
            const float gex = g.x * s[6] / tau, gey = g.y * s[6] / tau;
            float* o = g_self + (base + n) * 7;
            o[0] = gex / tx + (nx != 0.f ? gtx * (s[0] / nx) : 0.f);
            o[1] = gey / ty + (ny != 0.f ? gty * (s[1] / ny) : 0.f);
            o[2] = -g.x / tau;
            o[3] = -g.y / tau;
            o[4] = 0.f;
            o[5] = 0.f;
            o[6] = (g.x * (s[0] / tx) + g.y * (s[1] / ty)) / tau;
        }
        if (g_ped) for (int i = 0; i < kp; ++i) g_ped[(base + n) * kp + i] = g;
        if (g_obs) for (int i = 0; i < ko; ++i) g_obs[(base + n) * ko + i] = g;
    }
}

// ---- collision post-correction of PINNSF_polar_bottleneck_collision (src/models/model.py:1383-1444) ----
struct CorrSel {            // what one agent's pass over its k gathered neighbours selects
    int e, c;               // nearest head-on ("encounter") / chasing neighbour (0 when none)
    float me, mc;           // 1 when such a neighbour exists
};

// Classify the k neighbours (rows of `stride` floats: p_j - p_i, v_j - v_i, ...) and pick the nearest of each
// kind.  collision: reaction radius >= |p_ji| + 1e-6 > 1e-4; head-on when (v_i . p_ji)(v_j . -p_ji) > 0.
__device__ __forceinline__ CorrSel corr_select(const float* __restrict__ ped, int k, int stride, float2 vi,
                                               float radius) {
    CorrSel s;
    s.e = 0; s.c = 0; s.me = 0.f; s.mc = 0.f;
    float best_e = 0.f, best_c = 0.f;
    bool have_e = false, have_c = false;
    for (int j = 0; j < k; ++j) {
        const float* f = ped + (size_t)j * stride;
        const float px = nan_to_zero(f[0]), py = nan_to_zero(f[1]);
        const float nrm = norm2(px, py) + 1e-6f;
        if (!(radius >= nrm && nrm > 1e-4f)) continue;
        const float vjx = f[2] + vi.x, vjy = f[3] + vi.y;
        float inter = (vi.x * px + vi.y * py) * (vjx * (-px) + vjy * (-py));
        const bool head_on = inter > 0.f;                    // NaN -> false, as the reference's masking does
        // torch.min over (distance * flag, +100 where < 1e-4): the first strict minimum among flagged rows;
        // unflagged rows sit at 100 and only win (index 0) when nothing is flagged
        if (head_on) {
            if (!have_e || nrm < best_e) { best_e = nrm; s.e = j; have_e = true; }
        } else {
            if (!have_c || nrm < best_c) { best_c = nrm; s.c = j; have_c = true; }
        }
    }
    // a flagged distance >= 100 would lose against the 100 of an unflagged row 0; radius < 100 always
    s.me = have_e ? 1.f : 0.f;
    s.mc = have_c ? 1.f : 0.f;
    return s;
}

__device__ __forceinline__ float2 corr_normal(const float* f, float& r_out, float2& p_out) {
    const float px = nan_to_zero(f[0]), py = nan_to_zero(f[1]);
    const float r = norm2(px, py);
    r_out = r;
    p_out = make_float2(px, py);
    const float d = r + 1e-6f;
    return make_float2(px / d, py / d);
}

__global__ void collision_correction_fwd_kernel(const float2* __restrict__ pred, const float* __restrict__ ped,
                                                const float2* __restrict__ vel, size_t rows, int k, int stride,
                                                float radius, float dt, float2* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows) return;
    if (k == 0) { out[g] = pred[g]; return; }          // no gathered neighbours: nothing to correct
    const float* f = ped + g * (size_t)k * stride;
    const float2 vi = vel[g];
    float2 P = pred[g];
    const CorrSel s = corr_select(f, k, stride, vi, radius);
    float r; float2 p;
    // step 2: head-on -- remove the approaching normal component, add the acceleration that stops v_i . n within dt
    {
        const float2 n = corr_normal(f + (size_t)s.e * stride, r, p);
        const float u = vi.x * n.x + vi.y * n.y;
        const float2 ac = make_float2(-u * n.x / dt * s.me, -u * n.y / dt * s.me);
        float2 Pm = make_float2(P.x * s.me, P.y * s.me);
        float sn = Pm.x * n.x + Pm.y * n.y;
        sn = sn > 0.f ? sn : 0.f;
        Pm = make_float2((Pm.x - sn * n.x) + ac.x, (Pm.y - sn * n.y) + ac.y);
        P = make_float2(P.x + Pm.x, P.y + Pm.y);
    }
    // step 3: chasing -- only when i closes in on j (relative normal speed q < 0)
    {
        const float* fc = f + (size_t)s.c * stride;
        const float2 n = corr_normal(fc, r, p);
        const float q = fc[2] * n.x + fc[3] * n.y;
        const float h = q < 0.f ? 1.f : 0.f;
        const float2 a = make_float2(q * h * n.x / dt * s.mc, q * h * n.y / dt * s.mc);
        float2 Pm = make_float2(P.x * s.mc, P.y * s.mc);
        float sn = Pm.x * n.x + Pm.y * n.y;
        sn = (sn > 0.f ? sn : 0.f) * h;
        Pm = make_float2((Pm.x - sn * n.x) + a.x, (Pm.y - sn * n.y) + a.y);
        P = make_float2(P.x + Pm.x, P.y + Pm.y);
    }
    out[g] = P;
}

// gradient of n = p / (|p| + eps) pulled back to p
__device__ __forceinline__ float2 corr_normal_bwd(float2 gn, float2 p, float r) {
    if (r == 0.f) return make_float2(gn.x / 1e-6f, gn.y / 1e-6f);      // torch: d|p|/dp = 0 at p = 0
    const float d = r + 1e-6f;
    const float pg = p.x * gn.x + p.y * gn.y;
    const float c = pg / (r * d * d);
    return make_float2(gn.x / d - p.x * c, gn.y / d - p.y * c);
}

// Analytic backward: the flags and the selected neighbours are piecewise constant, gradients flow through the
// two selected normals (p_ji of those rows), v_i, the selected chasing row's v_ji and the predictions.
__global__ void collision_correction_bwd_kernel(const float2* __restrict__ g_out, const float2* __restrict__ pred,
                                                const float* __restrict__ ped, const float2* __restrict__ vel,
                                                size_t rows, int k, int stride, float radius, float dt,
                                                float2* __restrict__ g_pred, float* __restrict__ g_ped,
                                                float2* __restrict__ g_vel) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows) return;
    if (k == 0) {                                      // identity forward: the gradient passes straight through
        if (g_pred) g_pred[g] = g_out[g];
        if (g_vel) g_vel[g] = make_float2(0.f, 0.f);
        return;
    }
    const float* f = ped + g * (size_t)k * stride;
    const float2 vi = vel[g];
    const float2 P = pred[g];
    const float2 G = g_out[g];
    const CorrSel s = corr_select(f, k, stride, vi, radius);
    float r1, r2; float2 p1, p2;
    const float2 n1 = corr_normal(f + (size_t)s.e * stride, r1, p1);
    const float* fc = f + (size_t)s.c * stride;
    const float2 n2 = corr_normal(fc, r2, p2);
    // recompute the forward intermediates
    const float u = vi.x * n1.x + vi.y * n1.y;
    const float s1raw = s.me * (P.x * n1.x + P.y * n1.y);
    const float g1 = s1raw > 0.f ? 1.f : 0.f;
    const float2 P1 = make_float2((1.f + s.me) * P.x - g1 * s1raw * n1.x - s.me * u * n1.x / dt,
                                  (1.f + s.me) * P.y - g1 * s1raw * n1.y - s.me * u * n1.y / dt);
    const float2 w = make_float2(fc[2], fc[3]);
    const float q = w.x * n2.x + w.y * n2.y;
    const float h = q < 0.f ? 1.f : 0.f;
    const float s2raw = s.mc * (P1.x * n2.x + P1.y * n2.y);
    const float g2 = (s2raw > 0.f ? 1.f : 0.f) * h;
    // step 3 backward
    const float Gn2 = G.x * n2.x + G.y * n2.y;
    const float2 G1 = make_float2((1.f + s.mc) * G.x - g2 * s.mc * Gn2 * n2.x, (1.f + s.mc) * G.y - g2 * s.mc * Gn2 * n2.y);
    const float2 gw = make_float2(s.mc * h * Gn2 * n2.x / dt, s.mc * h * Gn2 * n2.y / dt);
    const float P1n2 = P1.x * n2.x + P1.y * n2.y;
    const float2 gn2 = make_float2(-g2 * s.mc * (Gn2 * P1.x + P1n2 * G.x) + s.mc * h / dt * (Gn2 * w.x + q * G.x),
                                   -g2 * s.mc * (Gn2 * P1.y + P1n2 * G.y) + s.mc * h / dt * (Gn2 * w.y + q * G.y));
    // step 2 backward
    const float G1n1 = G1.x * n1.x + G1.y * n1.y;
    const float Pn1 = P.x * n1.x + P.y * n1.y;
    const float2 gP = make_float2((1.f + s.me) * G1.x - g1 * s.me * G1n1 * n1.x, (1.f + s.me) * G1.y - g1 * s.me * G1n1 * n1.y);
    const float2 gvi = make_float2(-s.me * G1n1 * n1.x / dt, -s.me * G1n1 * n1.y / dt);
    const float2 gn1 = make_float2(-g1 * s.me * (G1n1 * P.x + Pn1 * G1.x) - s.me / dt * (G1n1 * vi.x + u * G1.x),
                                   -g1 * s.me * (G1n1 * P.y + Pn1 * G1.y) - s.me / dt * (G1n1 * vi.y + u * G1.y));
    if (g_pred) g_pred[g] = gP;
    if (g_vel) g_vel[g] = gvi;
    if (g_ped) {
        float* o = g_ped + g * (size_t)k * stride;
        for (int j = 0; j < k * stride; ++j) o[j] = 0.f;
        const bool nan1 = f[(size_t)s.e * stride] != f[(size_t)s.e * stride] || f[(size_t)s.e * stride + 1] != f[(size_t)s.e * stride + 1];
        if (s.me != 0.f && !nan1) {
            const float2 gp = corr_normal_bwd(gn1, p1, r1);
            o[(size_t)s.e * stride] += gp.x;
            o[(size_t)s.e * stride + 1] += gp.y;
        }
        if (s.mc != 0.f) {
            const float2 gp = corr_normal_bwd(gn2, p2, r2);
            o[(size_t)s.c * stride] += gp.x;
            o[(size_t)s.c * stride + 1] += gp.y;
            o[(size_t)s.c * stride + 2] += gw.x;
            o[(size_t)s.c * stride + 3] += gw.y;
        }
    }
}

}  // namespace piml

PIML_API int piml_rollout_step(float* position, float* velocity, float* acceleration, float* destination,
                               int64_t* dest_idx, float* hist_velocity, int hist_width, const float* a_next,
                               const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                               const float* position_series, const float* velocity_series,
                               const float* acceleration_series, const float* destination_series,
                               const int64_t* dest_idx_series, const float* self_features_series, int F,
                               const uint8_t* new_flag, float* position_out, float* velocity_out,
                               float* acceleration_out, float* mask_out, float* self_features_next,
                               const float* desired_speed, const int64_t* frame_counter, int C, int T, int N,
                               float dt, int remove_arrived, void* stream) {
    if (C < 0 || T <= 0 || N < 0 || hist_width < 2 || F != hist_width + 5 || D <= 0) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    piml::RolloutArgs A;
    A.p = (float2*)position; A.v = (float2*)velocity; A.a = (float2*)acceleration; A.dest = (float2*)destination;
    A.dest_idx = (long long*)dest_idx; A.hist = hist_velocity; A.hist_w = hist_width;
    A.a_next = (const float2*)a_next; A.waypoints = (const float2*)waypoints; A.D = D;
    A.wp_per_slice = waypoints_per_slice; A.dest_num = (const long long*)dest_num;
    A.pos_s = (const float2*)position_series; A.vel_s = (const float2*)velocity_series;
    A.acc_s = (const float2*)acceleration_series; A.dest_s = (const float2*)destination_series;
    A.dest_idx_s = (const long long*)dest_idx_series; A.selff_s = self_features_series; A.F = F;
    A.new_flag = new_flag; A.p_res = (float2*)position_out; A.v_res = (float2*)velocity_out;
    A.a_res = (float2*)acceleration_out; A.mask_new = mask_out; A.selff_out = self_features_next;
    A.desired_speed = desired_speed; A.t_dev = (const long long*)frame_counter;
    A.C = C; A.T = T; A.N = N; A.dt = dt; A.remove_arrived = remove_arrived;
    const long n = (long)C * N;
    hipLaunchKernelGGL(piml::rollout_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       piml::as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_rollout_step_ksum(const float* pred_ped, int kp, const float* pred_obs, int ko, float tau, float* position, float* velocity, float* acceleration, float* destination,
                               int64_t* dest_idx, float* hist_velocity, int hist_width, const float* a_next,
                               const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                               const float* position_series, const float* velocity_series,
                               const float* acceleration_series, const float* destination_series,
                               const int64_t* dest_idx_series, const float* self_features_series, int F,
                               const uint8_t* new_flag, float* position_out, float* velocity_out,
                               float* acceleration_out, float* mask_out, float* self_features_next,
                               const float* desired_speed, const int64_t* frame_counter, int C, int T, int N,
                               float dt, int remove_arrived, void* stream) {
    if (C < 0 || T <= 0 || N < 0 || hist_width < 2 || F != hist_width + 5 || D <= 0) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    piml::RolloutArgs A;
    A.p = (float2*)position; A.v = (float2*)velocity; A.a = (float2*)acceleration; A.dest = (float2*)destination;
    A.dest_idx = (long long*)dest_idx; A.hist = hist_velocity; A.hist_w = hist_width;
    A.a_next = (const float2*)a_next; A.waypoints = (const float2*)waypoints; A.D = D;
    A.wp_per_slice = waypoints_per_slice; A.dest_num = (const long long*)dest_num;
    A.pos_s = (const float2*)position_series; A.vel_s = (const float2*)velocity_series;
    A.acc_s = (const float2*)acceleration_series; A.dest_s = (const float2*)destination_series;
    A.dest_idx_s = (const long long*)dest_idx_series; A.selff_s = self_features_series; A.F = F;
    A.new_flag = new_flag; A.p_res = (float2*)position_out; A.v_res = (float2*)velocity_out;
    A.a_res = (float2*)acceleration_out; A.mask_new = mask_out; A.selff_out = self_features_next;
    A.desired_speed = desired_speed; A.t_dev = (const long long*)frame_counter;
    A.C = C; A.T = T; A.N = N; A.dt = dt; A.remove_arrived = remove_arrived;
    if (!pred_ped || kp < 1 || (pred_obs && ko < 1) || !(tau > 0.f) || F != 7 || !self_features_next) return hipErrorInvalidValue;
    const long n = (long)C * N;
    hipLaunchKernelGGL(piml::rollout_step_ksum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, piml::as_stream(stream), A,
                       (const float2*)pred_ped, kp, (const float2*)pred_obs, ko, (const float*)self_features_next, tau);
    return hipGetLastError();
}

static int train_step_fwd_impl(const float* position, const float* velocity, const float* acceleration,
                               const float* a_pred, const float* destination, const int64_t* dest_idx,
                               const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                               const uint8_t* new_flag, const float* position_series,
                               const float* velocity_series, const float* acceleration_series,
                               const float* destination_series, const int64_t* dest_idx_series, int C, int T,
                               int N, int t_next, float dt, float* position_out, float* velocity_out,
                               float* acceleration_out, float* destination_out, int64_t* dest_idx_out,
                               int* nan_flag, uint8_t* zero_mask, float* position_copy, long long position_copy_slice_stride, void* stream,
                               const piml::TailArgs* tail = nullptr);

PIML_API int piml_train_step_fwd(const float* position, const float* velocity, const float* acceleration,
                                 const float* a_pred, const float* destination, const int64_t* dest_idx,
                                 const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                                 const uint8_t* new_flag, const float* position_series,
                                 const float* velocity_series, const float* acceleration_series,
                                 const float* destination_series, const int64_t* dest_idx_series, int C, int T,
                                 int N, int t_next, float dt, float* position_out, float* velocity_out,
                                 float* acceleration_out, float* destination_out, int64_t* dest_idx_out,
                                 int* nan_flag, uint8_t* zero_mask, void* stream) {
    return train_step_fwd_impl(position, velocity, acceleration, a_pred, destination, dest_idx, waypoints, D, waypoints_per_slice, dest_num,
                               new_flag, position_series, velocity_series, acceleration_series, destination_series, dest_idx_series, C, T, N,
                               t_next, dt, position_out, velocity_out, acceleration_out, destination_out, dest_idx_out, nan_flag, zero_mask,
                               nullptr, 0, stream);
}

PIML_API int piml_train_step_fwd_copy(const float* position, const float* velocity, const float* acceleration,
                                      const float* a_pred, const float* destination, const int64_t* dest_idx,
                                      const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                                      const uint8_t* new_flag, const float* position_series,
                                      const float* velocity_series, const float* acceleration_series,
                                      const float* destination_series, const int64_t* dest_idx_series, int C, int T,
                                      int N, int t_next, float dt, float* position_out, float* velocity_out,
                                      float* acceleration_out, float* destination_out, int64_t* dest_idx_out,
                                      int* nan_flag, uint8_t* zero_mask, float* position_copy, long long position_copy_slice_stride,
                                      void* stream) {
    if (position_copy && (position_copy_slice_stride & 1)) return hipErrorInvalidValue;
    return train_step_fwd_impl(position, velocity, acceleration, a_pred, destination, dest_idx, waypoints, D, waypoints_per_slice, dest_num,
                               new_flag, position_series, velocity_series, acceleration_series, destination_series, dest_idx_series, C, T, N,
                               t_next, dt, position_out, velocity_out, acceleration_out, destination_out, dest_idx_out, nan_flag, zero_mask,
                               position_copy, position_copy_slice_stride, stream);
}

static int train_step_fwd_impl(const float* position, const float* velocity, const float* acceleration,
                               const float* a_pred, const float* destination, const int64_t* dest_idx,
                               const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                               const uint8_t* new_flag, const float* position_series,
                               const float* velocity_series, const float* acceleration_series,
                               const float* destination_series, const int64_t* dest_idx_series, int C, int T,
                               int N, int t_next, float dt, float* position_out, float* velocity_out,
                               float* acceleration_out, float* destination_out, int64_t* dest_idx_out,
                               int* nan_flag, uint8_t* zero_mask, float* position_copy, long long position_copy_slice_stride, void* stream,
                               const piml::TailArgs* tail) {
    if (C < 0 || T <= 0 || N < 0 || D <= 0 || t_next < 0) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    if (!position || !velocity || !acceleration || (!a_pred && !tail) || !destination || !dest_idx || !waypoints || !dest_num ||
        !position_out || !velocity_out || !acceleration_out || !destination_out || !dest_idx_out)
        return hipErrorInvalidValue;
    if (new_flag && t_next < T &&
        (!position_series || !velocity_series || !acceleration_series || !destination_series || !dest_idx_series))
        return hipErrorInvalidValue;
    piml::TrainStepArgs A;
    A.p = (const float2*)position; A.v = (const float2*)velocity; A.a = (const float2*)acceleration;
    A.a_pred = (const float2*)a_pred; A.dest = (const float2*)destination; A.dest_idx = (const long long*)dest_idx;
    A.waypoints = (const float2*)waypoints; A.D = D; A.wp_per_slice = waypoints_per_slice;
    A.dest_num = (const long long*)dest_num; A.new_flag = new_flag;
    A.pos_s = (const float2*)position_series; A.vel_s = (const float2*)velocity_series;
    A.acc_s = (const float2*)acceleration_series; A.dest_s = (const float2*)destination_series;
    A.dest_idx_s = (const long long*)dest_idx_series;
    A.p_out = (float2*)position_out; A.v_out = (float2*)velocity_out; A.a_out = (float2*)acceleration_out;
    A.dest_out = (float2*)destination_out; A.dest_idx_out = (long long*)dest_idx_out; A.nan_flag = nan_flag;
    A.zero_mask = zero_mask;
    A.C = C; A.T = T; A.N = N; A.t_next = t_next; A.dt = dt;
    A.p_copy = (float2*)position_copy; A.p_copy_cstride = position_copy_slice_stride / 2;
    const long n = (long)C * N;
    if (tail)
        hipLaunchKernelGGL(piml::train_step_tail_fwd_kernel, dim3((unsigned)C), dim3(256), 0, piml::as_stream(stream), A, *tail);
    else
        hipLaunchKernelGGL(piml::train_step_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           piml::as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_train_step_tail_fwd(const float* position, const float* velocity, const float* acceleration,
                                      const float* acc_ped, int kp, const float* acc_obs, int ko, const float* self_features, float tau,
                                      const float* destination, const int64_t* dest_idx,
                                      const float* waypoints, int D, int waypoints_per_slice, const int64_t* dest_num,
                                      const uint8_t* new_flag, const float* position_series,
                                      const float* velocity_series, const float* acceleration_series,
                                      const float* destination_series, const int64_t* dest_idx_series, int C, int T,
                                      int N, int t_next, float dt, float* position_out, float* velocity_out,
                                      float* acceleration_out, float* destination_out, int64_t* dest_idx_out,
                                      int* nan_flag, uint8_t* zero_mask, float* position_copy, long long position_copy_slice_stride,
                                      void* stream) {
    if (!acc_ped || kp < 1 || (acc_obs && ko < 1) || !self_features || (position_copy && (position_copy_slice_stride & 1)))
        return hipErrorInvalidValue;
    piml::TailArgs E;
    E.acc_ped = (const float2*)acc_ped; E.acc_obs = (const float2*)acc_obs; E.kp = kp; E.ko = acc_obs ? ko : 1;
    E.sf = self_features; E.tau = tau;
    return train_step_fwd_impl(position, velocity, acceleration, nullptr, destination, dest_idx, waypoints, D, waypoints_per_slice, dest_num,
                               new_flag, position_series, velocity_series, acceleration_series, destination_series, dest_idx_series, C, T, N,
                               t_next, dt, position_out, velocity_out, acceleration_out, destination_out, dest_idx_out, nan_flag, zero_mask,
                               position_copy, position_copy_slice_stride, stream, &E);
}

PIML_API int piml_train_step_bwd(const float* g_position_out, const float* g_velocity_out,
                                 const float* g_acceleration_out, const uint8_t* new_flag, const uint8_t* zero_mask,
                                 int C, int T, int N, int t_next, float dt, float* g_position, float* g_velocity,
                                 float* g_acceleration, float* g_a_pred, void* stream) {
    if (C < 0 || T <= 0 || N < 0 || t_next < 0) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    const long n = (long)C * N;
    hipLaunchKernelGGL(piml::train_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       piml::as_stream(stream), (const float2*)g_position_out, (const float2*)g_velocity_out,
                       (const float2*)g_acceleration_out, (const float2*)nullptr, new_flag, zero_mask, C, T, N, t_next, dt,
                       (float2*)g_position, (float2*)g_velocity, (float2*)g_acceleration, (float2*)g_a_pred);
    return hipGetLastError();
}

PIML_API int piml_train_step_bwd6(const float* g_position_out, const float* g_velocity_out, const float* g_acceleration_out,
                                  const float* g_state6, const unsigned char* new_flag, const unsigned char* zero_mask, int C, int T,
                                  int N, int t_next, float dt, float* g_position, float* g_velocity, float* g_acceleration,
                                  float* g_a_pred, void* stream) {
    if (C < 0 || N < 0 || T < 0) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    const long n = (long)C * N;
    hipLaunchKernelGGL(piml::train_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       piml::as_stream(stream), (const float2*)g_position_out, (const float2*)g_velocity_out,
                       (const float2*)g_acceleration_out, (const float2*)g_state6, new_flag, zero_mask, C, T, N, t_next, dt,
                       (float2*)g_position, (float2*)g_velocity, (float2*)g_acceleration, (float2*)g_a_pred);
    return hipGetLastError();
}

PIML_API int piml_train_step_bwd7(const float* g_position_out, long long g_position_out_slice_stride, const float* g_velocity_out,
                                  const float* g_acceleration_out,
                                  const float* g_state6, const float* g_position_in, long long g_position_in_slice_stride,
                                  const unsigned char* new_flag, const unsigned char* zero_mask, int C, int T, int N, int t_next,
                                  float dt, float* g_position, float* g_velocity, float* g_acceleration, float* g_a_pred, void* stream) {
    if (C < 0 || N < 0 || T < 0 || (g_position_in && (g_position_in_slice_stride & 1)) || (g_position_out_slice_stride & 1))
        return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    const long n = (long)C * N;
    hipLaunchKernelGGL(piml::train_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       piml::as_stream(stream), (const float2*)g_position_out, (const float2*)g_velocity_out,
                       (const float2*)g_acceleration_out, (const float2*)g_state6, new_flag, zero_mask, C, T, N, t_next, dt,
                       (float2*)g_position, (float2*)g_velocity, (float2*)g_acceleration, (float2*)g_a_pred,
                       (const float2*)g_position_in, g_position_in_slice_stride / 2, g_position_out_slice_stride / 2);
    return hipGetLastError();
}

PIML_API int piml_train_step_tail_bwd(const float* g_position_out, long long g_position_out_slice_stride, const float* g_velocity_out,
                                      const float* g_acceleration_out, const float* g_state6, const float* g_position_in,
                                      long long g_position_in_slice_stride, const unsigned char* new_flag, const unsigned char* zero_mask,
                                      int C, int T, int N, int t_next, float dt, float* g_position, float* g_velocity, float* g_acceleration,
                                      float* g_prediction, const float* self_features, float tau, int kp, int ko, float* g_acc_ped,
                                      float* g_acc_obs, float* g_self, void* stream) {
    if (C < 0 || N < 0 || T < 0 || (g_position_in && (g_position_in_slice_stride & 1)) || (g_position_out_slice_stride & 1) || kp < 1 ||
        (g_acc_obs && ko < 1))
        return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    if (!g_prediction || !self_features) return hipErrorInvalidValue;
    piml::TrainBwdArgs B;
    B.gp_o = (const float2*)g_position_out; B.gv_o = (const float2*)g_velocity_out; B.ga_o = (const float2*)g_acceleration_out;
    B.g6 = (const float2*)g_state6; B.new_flag = new_flag; B.zero_mask = zero_mask; B.C = C; B.T = T; B.N = N; B.t_next = t_next; B.dt = dt;
    B.gp = (float2*)g_position; B.gv = (float2*)g_velocity; B.ga = (float2*)g_acceleration; B.ga_pred = (float2*)g_prediction;
    B.gp_in = (const float2*)g_position_in; B.gp_in_cstride = g_position_in_slice_stride / 2; B.gp_o_cstride = g_position_out_slice_stride / 2;
    hipLaunchKernelGGL(piml::train_step_tail_bwd_kernel, dim3((unsigned)C), dim3(256), 0, piml::as_stream(stream), B, self_features, tau, kp,
                       g_acc_obs ? ko : 1, (float2*)g_acc_ped, (float2*)g_acc_obs, g_self);
    return hipGetLastError();
}

PIML_API int piml_collision_correction_fwd(const float* predictions, const float* ped_features, const float* velocity,
                                           size_t rows, int k, int row_stride, float collision_threshold,
                                           float time_unit, float* out, void* stream) {
    if (k < 0 || row_stride < 4 || !(time_unit > 0.f)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!predictions || !velocity || !out || (k > 0 && !ped_features)) return hipErrorInvalidValue;
    const float radius = (float)((double)collision_threshold + 1.34 * 2 * (double)time_unit);
    hipLaunchKernelGGL(piml::collision_correction_fwd_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0,
                       piml::as_stream(stream), (const float2*)predictions, ped_features, (const float2*)velocity, rows,
                       k, row_stride, radius, time_unit, (float2*)out);
    return hipGetLastError();
}

PIML_API int piml_collision_correction_bwd(const float* g_out, const float* predictions, const float* ped_features,
                                           const float* velocity, size_t rows, int k, int row_stride,
                                           float collision_threshold, float time_unit, float* g_predictions,
                                           float* g_ped_features, float* g_velocity, void* stream) {
    if (k < 0 || row_stride < 4 || !(time_unit > 0.f)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!g_out || !predictions || !velocity || (k > 0 && !ped_features)) return hipErrorInvalidValue;
    const float radius = (float)((double)collision_threshold + 1.34 * 2 * (double)time_unit);
    hipLaunchKernelGGL(piml::collision_correction_bwd_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0,
                       piml::as_stream(stream), (const float2*)g_out, (const float2*)predictions, ped_features,
                       (const float2*)velocity, rows, k, row_stride, radius, time_unit, (float2*)g_predictions,
                       g_ped_features, (float2*)g_velocity);
    return hipGetLastError();
}
