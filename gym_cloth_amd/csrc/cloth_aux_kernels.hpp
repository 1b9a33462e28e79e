// cloth_aux_kernels.hpp -- the small kernels around the stepper: Gripper.grab_top / grab (gripper.pyx:23-53), release, observation and
// metrics read-back, flat reset, self-tests.
#pragma once

#include "cloth_common.hpp"
#include "cloth_metrics.hpp"

namespace clothhip {

// ---- Gripper.grab_top / grab (gripper.pyx:23-53): one wave per env ---------------------------------
template <typename T> struct GrabArgs {
    const T *pos; uint8_t *cnt;
    const double *xy;        // [E][2]
    const double *radius;    // [E] or nullptr
    const uint8_t *active;   // [E] or nullptr
    int32_t *n_grabbed;      // [E]
    const double *levels;    // [n_levels] curZ table (double; cast per use)
    int32_t n_levels, P, Ppad, top;
    double default_radius, two_thickness;
};

template <typename T> __global__ __launch_bounds__(64) void k_grab(GrabArgs<T> A) {
    const int e = blockIdx.x, lane = threadIdx.x;
    if (A.active && !A.active[e]) { if (lane == 0) A.n_grabbed[e] = 0; return; }
    const T gx = (T)A.xy[2 * e], gy = (T)A.xy[2 * e + 1];
    const T rad = (T)(A.radius ? A.radius[e] : A.default_radius);
    const T tt = (T)A.two_thickness;
    const T *px = A.pos + (size_t)e * 3 * A.Ppad, *py = px + A.Ppad, *pz = py + A.Ppad;
    uint8_t *cnt = A.cnt + (size_t)e * A.Ppad;
    int best = 0x7fffffff;
    if (A.top) {
        // first level (scanning down from `height`) at which any in-cylinder point lies in the band
        for (int i = lane; i < A.P; i += 64) {
            const T dx = px[i] - gx, dy = py[i] - gy;
            if (dx * dx + dy * dy < rad) {                              // gripper.pyx:35 (radius not squared)
                const T z = pz[i];
                for (int l = 0; l < A.n_levels && l < best; l++) {
                    T d = z - (T)A.levels[l]; d = d < 0 ? -d : d;
                    if (d < tt) { best = l; break; }                    // gripper.pyx:36
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(best, o); best = v < best ? v : best; }
        if (best == 0x7fffffff) { if (lane == 0) A.n_grabbed[e] = 0; return; }
    }
    int n = 0;
    for (int i = lane; i < A.P; i += 64) {
        const T dx = px[i] - gx, dy = py[i] - gy;
        if (dx * dx + dy * dy < rad) {
            bool hit = true;
            if (A.top) { T d = pz[i] - (T)A.levels[best]; d = d < 0 ? -d : d; hit = d < tt; }
            if (hit) {                                                  // pinned = True ; grabbed_pts.append
                uint8_t c = cnt[i];
                if ((c & CNT_GRAB_MASK) < CNT_GRAB_MASK) c = (uint8_t)(c + 1);
                cnt[i] = c; n++;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) A.n_grabbed[e] = n;
}

__global__ void k_release(uint8_t *cnt, const uint8_t *active, int Ppad) {
    const int e = blockIdx.x;
    if (active && !active[e]) return;
    uint8_t *c = cnt + (size_t)e * Ppad;
    for (int i = threadIdx.x; i < Ppad; i += blockDim.x) if (c[i] & CNT_GRAB_MASK) c[i] = 0;
}

// '1d' observation (cloth_env.py:196-200) as float32 [E][3P], from SoA device state
template <typename T> __global__ void k_write_obs(const T *pos, float *out, int P, int Ppad) {
    const int e = blockIdx.x;
    const T *p = pos + (size_t)e * 3 * Ppad;
    float *o = out + (size_t)e * 3 * P;
    for (int t = threadIdx.x; t < 3 * P; t += blockDim.x) {
        const int i = t / 3, ax = t - 3 * i;
        o[t] = (float)p[ax * Ppad + i];
    }
}

// ---- per-env metrics kernel: one 256-thread workgroup per env over the SoA state in HBM (metrics_block above)
template <typename T>
__global__ __launch_bounds__(256) void k_metrics(const T *pos, int P, int Ppad, int NS, int NH, double *cov, double *vinv, uint8_t *oob,
                                                 int32_t *hcnt, double half_thick) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x;
    const T *px = pos + (size_t)e * 3 * Ppad, *py = px + Ppad, *pz = py + Ppad;
    auto src = [&](int i, double &x, double &y, double &z) { x = (double)px[i]; y = (double)py[i]; z = (double)pz[i]; };
    double out[4];
    metrics_block<256, T>(src, P, NS, NH, smem, (int)threadIdx.x, half_thick, out);
    if (threadIdx.x == 0) {
        cov[e] = out[0]; vinv[e] = out[1]; oob[e] = out[2] != 0.0 ? 1 : 0;
        if (hcnt) hcnt[e] = (int32_t)out[3];
    }
}

// Cloth(...) rebuilt on reset (cloth_env.py:737-746) for the flat tiers 1/3: masked envs <- the flat grid (pos = prev),
// nothing pinned, tear flag cleared; with per-env rest tables also the flat rest lengths.
template <typename T>
__global__ void k_reset_flat(T *pos, T *prev, uint8_t *cnt, int32_t *tear, const T *flat, const uint8_t *mask, int Ppad,
                             T *rest, const T *flat_rest, int rest_stride, int Spad) {
    const int e = blockIdx.x;
    if (mask && !mask[e]) return;
    T *p = pos + (size_t)e * 3 * Ppad, *q = prev + (size_t)e * 3 * Ppad;
    for (int i = threadIdx.x; i < 3 * Ppad; i += blockDim.x) { const T v = flat[i]; p[i] = v; q[i] = v; }
    for (int i = threadIdx.x; i < Ppad; i += blockDim.x) cnt[(size_t)e * Ppad + i] = 0;
    if (rest_stride)
        for (int i = threadIdx.x; i < Spad; i += blockDim.x) rest[(size_t)e * rest_stride + i] = flat_rest[i];
    if (threadIdx.x == 0) tear[e] = 0;
}

// A state change from outside the episode launches voids the operation a time slice left in flight -- for the envs it touches only:
// mask (or the schedules' active flags) selects them, nullptr = every env.
__global__ void k_clear_resume(EpResume *r, const uint8_t *mask, const ClothSchedule *sched, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (mask && !mask[e]) return;
    if (sched && !(sched[e].active && sched[e].n_total > 0)) return;
    r[e].valid = 0;
}

__global__ void k_selftest(int op, const double *a, const double *b, double *out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    if (op == 0) r = x / y;
    else if (op == 1) r = sqrt(x);
    else if (op == 2) r = x * y + y;
    else r = floor(x / y);
    out[i] = r;
}

}  // namespace clothhip
