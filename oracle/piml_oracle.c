/*
 * piml_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the reference's per-timestep pairwise hot path
 * (tsinghua-fib-lab/PIML).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (piml_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked by tests/test_oracle_golden.py
 * against golden vectors captured by importing the real reference in the build
 * container (tests/golden/make_golden.py; the reference itself never ships).
 *
 * Float32 arithmetic is restated exactly as PyTorch's CPU kernels evaluate it
 * (measured, see DESIGN.md "pinned arithmetic"):
 *   norm2(x,y)  = sqrtf(fmaf(y, y, x*x))
 *   cosine(a,b) = (a0/max(|a|,1e-8))*(b0/max(|b|,1e-8)) + (a1/..)*(b1/..)   (no fma)
 * Build with -ffp-contract=off so the compiler introduces no other fused operations.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline float norm2f(float x, float y) { return sqrtf(fmaf(y, y, x * x)); }

ORACLE_API void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------
 * Heading direction.  src/data/data.py:350-395 (get_heading_direction).
 * velocity (C, T, N, 2) -> heading (C, T, N, 2): zero-velocity frames are filled from
 * the temporally nearest non-zero frame (backward sweep, then forward sweep), then
 * h / |h| with |h| == 0 -> divide by 0.1.  With T == 1 this is v/|v| (0 if v == 0).
 * The "is zero" test is torch.norm(h) == 0, i.e. norm2f(h) == 0.
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_heading(const float* vel, int C, int T, int N, float* out) {
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < N; ++i) {
            float tx = 0.f, ty = 0.f;
            for (int t = T - 1; t >= 0; --t) {
                size_t o = (((size_t)c * T + t) * N + i) * 2;
                float hx = vel[o], hy = vel[o + 1];
                if (norm2f(hx, hy) == 0.f) { hx = tx; hy = ty; } else { tx = hx; ty = hy; }
                out[o] = hx; out[o + 1] = hy;
            }
            for (int t = 0; t < T; ++t) {
                size_t o = (((size_t)c * T + t) * N + i) * 2;
                float hx = out[o], hy = out[o + 1];
                if (norm2f(hx, hy) == 0.f) { hx = tx; hy = ty; } else { tx = hx; ty = hy; }
                float n = norm2f(hx, hy);
                if (n == 0.f) n = n + 0.1f;
                out[o] = hx / n; out[o + 1] = hy / n;
            }
        }
}

/* ---------------------------------------------------------------------------------
 * Top-k in-view neighbours of one focal agent among `M` objects.
 * src/data/data.py:416-447 (get_nearby_obj_in_sight) + :449-464 (get_filtered_features).
 * Sort key is (distance, index): ties resolve to the lower index (torch.sort on equal
 * keys is unspecified).  Slots beyond the in-view, in-range set get
 * idx = -1, dist = +inf.
 * ------------------------------------------------------------------------------- */
static void topk_in_sight(float pix, float piy, float hx, float hy, const float* obj, int M,
                          int k, float cos_thr, float dist_thr, int* idx_out, float* dist_out) {
    /* cosine_similarity re-normalises the (already unit) heading: x2 / max(|x2|, eps) */
    float n2 = norm2f(hx, hy);
    float n2c = fmaxf(n2, 1e-8f);
    float h0 = hx / n2c, h1 = hy / n2c;
    for (int s = 0; s < k; ++s) { idx_out[s] = -1; dist_out[s] = INFINITY; }
    for (int j = 0; j < M; ++j) {
        float rx = obj[2 * j] - pix, ry = obj[2 * j + 1] - piy;
        if (isnan(rx)) rx = INFINITY;                      /* data.py:433 */
        if (isnan(ry)) ry = INFINITY;
        float d = norm2f(rx, ry);                          /* data.py:434 */
        float n1c = fmaxf(d, 1e-8f);
        float c0 = (rx / n1c) * h0;
        float c1 = (ry / n1c) * h1;
        float cs = c0 + c1;                                /* data.py:439-440 */
        if (isnan(cs)) cs = -1.f;                          /* data.py:441 */
        if (cs < cos_thr) d = INFINITY;                    /* data.py:442-443 */
        if (d > dist_thr) continue;                        /* data.py:461 (strict >) */
        /* insertion into the sorted (dist, idx) list; j ascends so ties keep lower idx */
        int s = k;
        while (s > 0 && dist_out[s - 1] > d) --s;
        if (s == k) continue;
        for (int q = k - 1; q > s; --q) { dist_out[q] = dist_out[q - 1]; idx_out[q] = idx_out[q - 1]; }
        dist_out[s] = d; idx_out[s] = j;
    }
}

/* ---------------------------------------------------------------------------------
 * Relative features, forward.  src/data/data.py:466-512 (get_relative_features).
 * Inputs are (C, N, 2) slices (C = product of all leading dims incl. time); `heading`
 * is the output of oracle_heading for the same slices; velocity / acceleration must
 * already have NaN -> 0 applied (data.py:483-484, done by the caller in place).
 * obstacles (M, 2) shared by all slices.  kp_eff = min(kp, N), ko_eff = min(ko, M).
 * Outputs: ped_feat (C,N,kp_eff,6), obs_feat (C,N,ko_eff,6), dest_feat (C,N,2),
 *          ped_idx/obs_idx (int32, -1 = empty slot), ped_dist/obs_dist.
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_relfeat_fwd(const float* p, const float* heading, const float* v,
                                   const float* a, const float* dest, const float* obs,
                                   int C, int N, int M, int kp, int ko,
                                   float cos_thr_p, float cos_thr_o, float dist_thr_p,
                                   float dist_thr_o, float* ped_feat, float* obs_feat,
                                   float* dest_feat, int* ped_idx, int* obs_idx,
                                   float* ped_dist, float* obs_dist) {
    int kpe = kp < N ? kp : N;
    int koe = ko < M ? ko : M;
#pragma omp parallel for schedule(dynamic, 16) collapse(2)
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < N; ++i) {
            size_t ci = (size_t)c * N + i;
            const float* pc = p + (size_t)c * N * 2;
            float pix = p[ci * 2], piy = p[ci * 2 + 1];
            float hx = heading[ci * 2], hy = heading[ci * 2 + 1];
            float vix = v[ci * 2], viy = v[ci * 2 + 1];
            float aix = a[ci * 2], aiy = a[ci * 2 + 1];
            int* pidx = ped_idx + ci * kpe;
            float* pdst = ped_dist + ci * kpe;
            topk_in_sight(pix, piy, hx, hy, pc, N, kpe, cos_thr_p, dist_thr_p, pidx, pdst);
            for (int s = 0; s < kpe; ++s) {
                float* f = ped_feat + (ci * kpe + s) * 6;
                int j = pidx[s];
                if (j < 0) { memset(f, 0, 6 * sizeof(float)); continue; }
                size_t cj = (size_t)c * N + j;
                f[0] = p[cj * 2] - pix;     f[1] = p[cj * 2 + 1] - piy;   /* data.py:491-492 */
                f[2] = v[cj * 2] - vix;     f[3] = v[cj * 2 + 1] - viy;
                f[4] = a[cj * 2] - aix;     f[5] = a[cj * 2 + 1] - aiy;
            }
            if (koe > 0) {
                int* oidx = obs_idx + ci * koe;
                float* odst = obs_dist + ci * koe;
                topk_in_sight(pix, piy, hx, hy, obs, M, koe, cos_thr_o, dist_thr_o, oidx, odst);
                for (int s = 0; s < koe; ++s) {
                    float* f = obs_feat + (ci * koe + s) * 6;
                    int j = oidx[s];
                    if (j < 0) { memset(f, 0, 6 * sizeof(float)); continue; }
                    f[0] = obs[2 * j] - pix; f[1] = obs[2 * j + 1] - piy;  /* data.py:506-508 */
                    f[2] = 0.f - vix;        f[3] = 0.f - viy;
                    f[4] = 0.f - aix;        f[5] = 0.f - aiy;
                }
            }
            float dx = dest[ci * 2] - pix, dy = dest[ci * 2 + 1] - piy;    /* data.py:496-497 */
            dest_feat[ci * 2] = isnan(dx) ? 0.f : dx;
            dest_feat[ci * 2 + 1] = isnan(dy) ? 0.f : dy;
        }
}

/* ---------------------------------------------------------------------------------
 * Relative features, backward (what autograd does through gather / repeat / masked
 * zeroing in data.py:397-414, 449-464, 491-510).  Accumulates in double, fixed order.
 * g_state (C,N,6) = grads w.r.t. (p,v,a) concatenated; g_dest (C,N,2).
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_relfeat_bwd(const float* g_ped, const float* g_obs, const float* g_destf,
                                   const int* ped_idx, const int* obs_idx, const float* p,
                                   const float* dest, int C, int N, int kpe, int koe,
                                   float* g_state, float* g_dest) {
    double* acc = (double*)calloc((size_t)C * N * 6, sizeof(double));
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < N; ++i) {
            size_t ci = (size_t)c * N + i;
            for (int s = 0; s < kpe; ++s) {
                int j = ped_idx[ci * kpe + s];
                if (j < 0) continue;
                size_t cj = (size_t)c * N + j;
                for (int q = 0; q < 6; ++q) {
                    double g = g_ped[(ci * kpe + s) * 6 + q];
                    acc[cj * 6 + q] += g;
                    acc[ci * 6 + q] -= g;
                }
            }
            for (int s = 0; s < koe; ++s) {
                if (obs_idx[ci * koe + s] < 0) continue;
                for (int q = 0; q < 6; ++q) acc[ci * 6 + q] -= g_obs[(ci * koe + s) * 6 + q];
            }
            for (int q = 0; q < 2; ++q) {
                float d = dest[ci * 2 + q] - p[ci * 2 + q];
                double g = isnan(d) ? 0.0 : (double)g_destf[ci * 2 + q];
                g_dest[ci * 2 + q] = (float)g;
                acc[ci * 6 + q] -= g;
            }
        }
    for (size_t t = 0; t < (size_t)C * N * 6; ++t) g_state[t] = (float)acc[t];
    free(acc);
}

/* ---------------------------------------------------------------------------------
 * Collision matrix for one stack of slices.  src/data/data.py:537-601
 * (collision_detection, pair part only: lines 549-564).  position (S, N, 2) ->
 * coll (S, N, N) in {0,1}: [|p_j - p_i| < thr] - I, NaN -> 0.
 * The self pair has distance 0 < thr -> 1 - 1 = 0; with a NaN position the row and
 * column are NaN -> 0 (including the diagonal, NaN - 1 = NaN -> 0).
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_collision_pairs(const float* p, int S, int N, float thr, float* coll) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < N; ++i) {
            const float* ps = p + (size_t)s * N * 2;
            float* row = coll + ((size_t)s * N + i) * N;
            for (int j = 0; j < N; ++j) {
                float rx = ps[2 * j] - ps[2 * i], ry = ps[2 * j + 1] - ps[2 * i + 1];
                float d = norm2f(rx, ry);
                float cval = isnan(d) ? NAN : (d < thr ? 1.f : 0.f);
                if (i == j) cval -= 1.f;
                row[j] = isnan(cval) ? 0.f : cval;
            }
        }
}

/* friends filter for 3-D input (data.py:587-591 / 573-585): friends_ij = [sum_s base_sij <= 25],
 * coll *= friends.  `base` is coll itself, or the (un-diagonal-corrected) collisions of
 * real_position when that is given (then the diagonal of base counts self pairs). */
ORACLE_API void oracle_collision_friends3(float* coll, const float* base, int S, int N) {
    size_t NN = (size_t)N * N;
    for (size_t q = 0; q < NN; ++q) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum += base[(size_t)s * NN + q];
        if (!(sum <= 25.f)) for (int s = 0; s < S; ++s) coll[(size_t)s * NN + q] = 0.f;
    }
}

/* real_position variant base matrix (data.py:576-581): [d < thr], NaN -> 0, no "- I". */
ORACLE_API void oracle_collision_pairs_raw(const float* p, int S, int N, float thr, float* coll) {
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < N; ++i) {
            const float* ps = p + (size_t)s * N * 2;
            float* row = coll + ((size_t)s * N + i) * N;
            for (int j = 0; j < N; ++j) {
                float d = norm2f(ps[2 * j] - ps[2 * i], ps[2 * j + 1] - ps[2 * i + 1]);
                row[j] = isnan(d) ? 0.f : (d < thr ? 1.f : 0.f);
            }
        }
}

/* friends filter for 4-D input (C,T,N,N) (data.py:592-598): pairs colliding in any of the
 * first 4 frames of a channel are dropped for every frame of that channel. */
ORACLE_API void oracle_collision_friends4(float* coll, int C, int T, int N) {
    size_t NN = (size_t)N * N;
    int T4 = T < 4 ? T : 4;
    for (int c = 0; c < C; ++c)
        for (size_t q = 0; q < NN; ++q) {
            float sum = 0.f;
            for (int t = 0; t < T4; ++t) sum += coll[((size_t)c * T + t) * NN + q];
            if (sum > 0.f) for (int t = 0; t < T; ++t) coll[((size_t)c * T + t) * NN + q] = 0.f;
        }
}

/* ---------------------------------------------------------------------------------
 * 1-second collision label.  src/data/data.py:514-535 (calculate_collision_label).
 * feat (R, >=4 cols with stride `ld`) -> label (R): any tau in {0,.1,...,.9} with
 * 0 != |dp + dv*tau| < 0.5.  tau = arange(10)*0.1 in float32.
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_collision_label(const float* feat, size_t R, int ld, float* label) {
    for (size_t r = 0; r < R; ++r) {
        const float* f = feat + r * ld;
        float hit = 0.f;
        for (int t = 0; t < 10; ++t) {
            float tau = (float)t * 0.1f;
            float x = f[0] + f[2] * tau, y = f[1] + f[3] * tau;
            float d = norm2f(x, y);
            if (d < 0.5f && d != 0.f) hit = 1.f;
        }
        label[r] = hit;
    }
}

/* ---------------------------------------------------------------------------------
 * Closed-form social force step.  src/models/mlapm.py:10-58 (MLAPM.step).
 * variant 0 = 'raw', 1 = 'GC', 2 = 'UCY' (with the one-line coll.unsqueeze(-1) fix,
 * SURVEY quirk Q8).  Pair terms are evaluated in float32 like the reference; the sum
 * over neighbours is accumulated in double (the reference's float32 .sum(dim=1) and
 * the HIP kernel's wave reduction are both compared against this within 1e-5).
 * Returns action = v + F*dt; `force` (optional) receives F.
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_mlapm_step(const float* p, const float* v, const float* v0,
                                  const float* dest, int N, int variant, float tau, float A,
                                  float B, float Cc, float D, float theta_deg, float radius,
                                  float dt, float* action, float* force) {
    const float PI_F = 3.14159265358979323846f;
    float th = theta_deg / 180.f * PI_F;                    /* mlapm.py:34 */
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        float pix = p[2 * i], piy = p[2 * i + 1], vix = v[2 * i], viy = v[2 * i + 1];
        float ex = dest[2 * i] - pix, ey = dest[2 * i + 1] - piy;
        float en = fmaxf(norm2f(ex, ey), 1e-12f);            /* F.normalize eps, mlapm.py:21 */
        ex /= en; ey /= en;
        float fx = (v0[i] * ex - vix) / tau, fy = (v0[i] * ey - viy) / tau;   /* :22 */
        double sx = 0.0, sy = 0.0;
        for (int j = 0; j < N; ++j) {
            float rx = p[2 * j] - pix, ry = p[2 * j + 1] - piy;               /* :25 */
            float r = norm2f(rx, ry);                                          /* :26 */
            float view = (vix * rx + viy * ry > 0.f) ? 1.f : 0.f;              /* :27 */
            float rn = fmaxf(r, 1e-12f);
            float nx = rx / rn, ny = ry / rn;
            float tx, ty;
            if (variant == 0) {
                float g = expf(B * r);                                         /* :29 */
                tx = view * A * g * nx; ty = view * A * g * ny;
            } else {
                float wx = v[2 * j] - vix, wy = v[2 * j + 1] - viy;            /* :31 / :42 */
                float cr = rx * ey - ry * ex;                                  /* :34 / :48 */
                float sg = (cr > 0.f) ? 1.f : ((cr < 0.f) ? -1.f : (cr == 0.f ? 0.f : NAN));
                float tij = -sg * theta_deg / 180.f * PI_F;
                if (tij == 0.f) tij = th;                                      /* :35 */
                float ct = cosf(tij), st = sinf(tij);
                float dx = ct * nx - st * ny, dy = st * nx + ct * ny;          /* :36-39 */
                float g;
                if (variant == 1) {
                    float wn = norm2f(wx, wy);
                    float r1 = fmaxf(r, 1e-8f), w1 = fmaxf(wn, 1e-8f);
                    float cs = (rx / r1) * (wx / w1) + (ry / r1) * (wy / w1);  /* :32 */
                    g = expf(B * r + Cc * cs + D * r * cs);                    /* :40 */
                } else {
                    float r2 = radius * 2.f;
                    int coll = norm2f(rx, ry) < r2;                            /* :43 */
                    coll |= norm2f(rx + wx * 1.0f, ry + wy * 1.0f) < r2;       /* :44 */
                    float rw = rx * wx + ry * wy, ww = wx * wx + wy * wy, rr = rx * rx + ry * ry;
                    float tmin = -rw / ww;                                     /* :45 */
                    float dmin = sqrtf(rr - rw * rw / ww);                     /* :46 */
                    coll |= (tmin > 0.f) && (tmin < 1.f) && (dmin < r2);       /* :47 */
                    float cf = coll ? 1.f : 0.f;
                    g = expf(B * r * cf + Cc * cf);                            /* :53 */
                }
                tx = view * A * g * dx; ty = view * A * g * dy;
            }
            sx += tx; sy += ty;
        }
        fx -= (float)sx; fy -= (float)sy;
        if (force) { force[2 * i] = fx; force[2 * i + 1] = fy; }
        action[2 * i] = vix + fx * dt;                                          /* :57 */
        action[2 * i + 1] = viy + fy * dt;
    }
}

/* ---------------------------------------------------------------------------------
 * Physics label on gathered neighbours.  src/utils/utils.py:31-100 (calc_acceleration).
 * version 0 = 'v0', 2 = 'v2' (v1 has C = 0 and is v0 with cos computed but unused
 * numerically: exp(B r + 0*cos)); rel (R, ld>=4) -> acc (R,2).  Reproduces quirk Q10:
 * "dv" is read from the position slice, so cos = |dr|^2/((r+eps)(r+eps)).
 * ------------------------------------------------------------------------------- */
ORACLE_API void oracle_calc_acceleration(const float* rel, size_t R, int ld, int version,
                                         float A, float B, float Cc, float D, float theta,
                                         float eps, float* acc) {
    float ct = (float)cos((double)theta), st = (float)sin((double)theta);
    for (size_t q = 0; q < R; ++q) {
        float dx = rel[q * ld], dy = rel[q * ld + 1];
        float r = norm2f(dx, dy) + eps;                       /* utils.py:54-55 */
        float ux = dx / r, uy = dy / r;
        if (version == 0) {
            float g = A * expf(B * r);
            acc[2 * q] = -g * ux; acc[2 * q + 1] = -g * uy;
        } else {
            float vn = norm2f(dx, dy) + eps;                  /* dv == dr, utils.py:84-88 */
            float cs = (dx * dx + dy * dy) / r / vn;
            float g = A * expf(B * r + Cc * cs + D * r * cs);
            if (version == 1) { acc[2 * q] = -g * ux; acc[2 * q + 1] = -g * uy; }
            else {
                float bx = ct * ux - st * uy, by = st * ux + ct * uy;
                acc[2 * q] = -g * bx; acc[2 * q + 1] = -g * by;
            }
        }
    }
}
