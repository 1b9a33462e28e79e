// cloth_kernels.hpp -- gfx950 device code of the cloth stepper.
//
// One workgroup steps ONE cloth through a whole schedule (up to ~2000 substeps) with the particle state
// resident in LDS; HBM is touched once on entry and once on exit.  Every phase keeps the reference's
// evaluation order (cloth.pyx:169-214), so the double instantiation reproduces the reference bit for bit
// (compiled with -ffp-contract=off) and the float instantiation is the same algorithm in fp32.
//
//   phase            reference                parallelisation (exact-order preserving)
//   adjust/release   gripper.pyx:55-73        per point
//   gravity+Hooke    cloth.pyx:216-237        per-point gather of <=12 springs in ascending list index
//   Verlet           cloth.pyx:239-256        per point (fused with the gather)
//   spatial map      cloth.pyx:298-311        sort of (cell key, point index) -> cells are contiguous runs
//   self-collision   cloth.pyx:313-343        one lane per cell, Gauss-Seidel in ascending index inside it
//   plane            cloth.pyx:345-370        per point (done by the cell's lane after its sweep)
//   strain limit     cloth.pyx:258-296        dependency-level schedule, one wave, levels in order
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/clothhip.h"
#include "cloth_tables.hpp"

namespace clothhip {

template <typename T> struct DevConsts {
    T mg;              // mass * gravity                         cloth.pyx:179
    T ksK[2];          // ks * 1.0, ks * 0.2                     cloth.pyx:225-232
    T dsm;             // (dt*dt)/mass                           cloth.pyx:240
    T damp;            // 1 - damping/100                        cloth.pyx:241
    T cw, ch, ct;      // hash cell extents w, h, t              cloth.pyx:308-310
    T thresh;          // 2 * thickness                          cloth.pyx:317
    T sim_steps;       // simulation_steps as a real             cloth.pyx:338
    T min_z;           // minimum_z                              cloth.pyx:356
    T surf_off;        // 0.0001                                 cloth.pyx:185
    T one_m_fric;      // 1. - plane_friction                    cloth.pyx:368
    T tear_thresh;     //                                        cloth.pyx:272
    T c11;             // 1.1                                    cloth.pyx:275
};

template <typename T> struct StepArgs {
    T *pos;                  // [E][3][Ppad]
    T *prev;                 // [E][3][Ppad]
    uint8_t *cnt;            // [E][Ppad]  bits0..6 multiplicity in grabbed_pts, bit7 pinned from outside
    const T *rest;           // [E or 1][Spad] rest lengths in LEVEL order
    int32_t *tear;           // [E] sticky Cloth.cloth_have_tear
    int32_t *executed;       // [E]
    const ClothSchedule *sched;   // [E]
    const uint32_t *gather;  // [HK_SLOTS][Ppad]
    const uint32_t *lv_ent;  // [S]  ptA | ptB<<16, level order
    const int32_t *lv_off;   // [n_levels+1]
    int32_t n_levels;
    int32_t N, P, Ppad, S, Spad, Psort;
    int32_t rest_stride;     // 0: one shared table
    DevConsts<T> k;
};

constexpr uint32_t KEY_BIAS = 1u << 19;         // composite sort word = (key+bias) << 12 | point index
constexpr int KEY_SHIFT = 12;
constexpr uint8_t CNT_GRAB_MASK = 0x7F, CNT_EXT_PIN = 0x80;

template <typename T> __device__ __forceinline__ T dev_sqrt(T x);
template <> __device__ __forceinline__ double dev_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float dev_sqrt<float>(float x) { return sqrtf(x); }
template <typename T> __device__ __forceinline__ T dev_floor(T x);
template <> __device__ __forceinline__ double dev_floor<double>(double x) { return floor(x); }
template <> __device__ __forceinline__ float dev_floor<float>(float x) { return floorf(x); }

// cloth.pyx:17-18, association ((x*x + y*y) + z*z)
template <typename T> __device__ __forceinline__ T fastnorm(T x, T y, T z) { return dev_sqrt<T>(x * x + y * y + z * z); }

// cloth.pyx:307-311 -> biased, clamped cell key (exact for |coordinate| < ~60 cloth widths)
template <typename T> __device__ __forceinline__ uint32_t cell_key(const DevConsts<T> &k, T x, T y, T z) {
    T fx = dev_floor<T>(x / k.cw), fy = dev_floor<T>(y / k.ch), fz = dev_floor<T>(z / k.ct);
    const T lim = (T)4096;
    fx = fx < -lim ? -lim : (fx > lim ? lim : fx);   // NaN falls through the compares; handled below
    fy = fy < -lim ? -lim : (fy > lim ? lim : fy);
    fz = fz < -lim ? -lim : (fz > lim ? lim : fz);
    if (!(fx == fx) || !(fy == fy) || !(fz == fz)) return (1u << 20) - 1u;
    int key = 961 * (int)fx + 31 * (int)fy + (int)fz;
    int kb = key + (int)KEY_BIAS;
    kb = kb < 0 ? 0 : (kb > (1 << 20) - 2 ? (1 << 20) - 2 : kb);
    return (uint32_t)kb;
}

// LDS carve-up (dynamic shared memory), all offsets in bytes, 16-byte aligned
template <typename T> struct LdsLayout {
    int bufA, bufB, cnt, sortw, misc, total;
    __host__ __device__ LdsLayout(int Ppad, int Psort) {
        int o = 0;
        bufA = o; o += 3 * Ppad * (int)sizeof(T);
        bufB = o; o += 3 * Ppad * (int)sizeof(T);
        cnt = o; o += (Ppad + 15) / 16 * 16;
        sortw = o; o += Psort * 4;
        misc = o; o += 64;
        total = o;
    }
};

template <typename T, int NT>
__global__ __launch_bounds__(NT) void k_run_schedule(StepArgs<T> A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const ClothSchedule sc = A.sched[e];
    if (!sc.active || sc.n_total <= 0) {
        if (tid == 0) A.executed[e] = 0;
        return;
    }
    const int P = A.P, Ppad = A.Ppad, Psort = A.Psort;
    const LdsLayout<T> lay(Ppad, Psort);
    T *cur = reinterpret_cast<T *>(smem + lay.bufA);
    T *prv = reinterpret_cast<T *>(smem + lay.bufB);
    uint8_t *cnt = smem + lay.cnt;
    uint32_t *sw = reinterpret_cast<uint32_t *>(smem + lay.sortw);
    volatile int *misc = reinterpret_cast<volatile int *>(smem + lay.misc);   // [0] = tear

    const DevConsts<T> k = A.k;
    const T *rest = A.rest + (size_t)e * A.rest_stride;

    {   // HBM -> LDS, coalesced
        const T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        for (int i = tid; i < 3 * Ppad; i += NT) { cur[i] = gp[i]; prv[i] = gq[i]; }
        const uint8_t *gc = A.cnt + (size_t)e * Ppad;
        for (int i = tid; i < Ppad; i += NT) cnt[i] = gc[i];
        if (tid == 0) misc[0] = A.tear[e];
    }
    __syncthreads();

    const T dz_up = (T)sc.dz_up, dxp = (T)sc.dx_pull, dyp = (T)sc.dy_pull, dzp = (T)sc.dz_pull;
    int done = 0;
    for (int it = 0; it < sc.n_total; it++) {
        // ---- ClothEnv._pull (cloth_env.py:352-367): adjust / nothing / release -------------------
        int mode = 0; T ax = 0, ay = 0, az = 0;
        if (it < sc.n_up_end) { mode = 1; az = dz_up; }
        else if (it < sc.n_uprest_end) { }
        else if (it < sc.n_pull_end) { mode = 1; ax = dxp; ay = dyp; az = dzp; }
        else if (it < sc.n_griprest_end) { }
        else mode = 2;
        if (mode == 1) {
            for (int i = tid; i < P; i += NT) {
                int m = cnt[i] & CNT_GRAB_MASK;
                for (int q = 0; q < m; q++) {           // gripper.pyx:60-66: p <- x ; x <- delta + x
                    T x = cur[i], y = cur[Ppad + i], z = cur[2 * Ppad + i];
                    prv[i] = x; prv[Ppad + i] = y; prv[2 * Ppad + i] = z;
                    cur[i] = ax + x; cur[Ppad + i] = ay + y; cur[2 * Ppad + i] = az + z;
                }
            }
        } else if (mode == 2) {
            for (int i = tid; i < P; i += NT)          // gripper.pyx:68-73
                if (cnt[i] & CNT_GRAB_MASK) cnt[i] = 0;
        }
        __syncthreads();

        // ---- gravity + Hooke gather + Verlet (cloth.pyx:216-256) ----------------------------------
        // new position goes to the point's slot in `prv` (only its owner reads that slot); the two
        // buffers then swap roles, which is exactly p <- x ; x <- new for unpinned points.
        for (int i = tid; i < P; i += NT) {
            if (cnt[i]) continue;                       // pinned: force irrelevant (cloth.pyx:244)
            const T x = cur[i], y = cur[Ppad + i], z = cur[2 * Ppad + i];
            T fx = (T)0 + (T)0, fy = (T)0 + (T)0, fz = (T)0 + k.mg;
#pragma unroll
            for (int s = 0; s < HK_SLOTS; s++) {
                const uint32_t g = A.gather[s * Ppad + i];
                if (!(g & HK_VALID)) break;
                const int j = (int)(g & HK_NBR_MASK);
                const T r = rest[(g >> HK_POS_SHIFT) & HK_POS_MASK];
                const T kk = k.ksK[(g & HK_BEND) ? 1 : 0];
                const T xj = cur[j], yj = cur[Ppad + j], zj = cur[2 * Ppad + j];
                if (g & HK_ASB) {                       // this point is ptB: d = pb - pa = self - nbr
                    const T dx = x - xj, dy = y - yj, dz = z - zj;
                    const T l = fastnorm<T>(dx, dy, dz);
                    const T fm = kk * (l - r) / l;      // cloth.pyx:232
                    fx = fx + (-(fm * dx)); fy = fy + (-(fm * dy)); fz = fz + (-(fm * dz));   // :237
                } else {                                // this point is ptA: d = nbr - self
                    const T dx = xj - x, dy = yj - y, dz = zj - z;
                    const T l = fastnorm<T>(dx, dy, dz);
                    const T fm = kk * (l - r) / l;
                    fx = fx + fm * dx; fy = fy + fm * dy; fz = fz + fm * dz;                  // :236
                }
            }
            const T px = prv[i], py = prv[Ppad + i], pz = prv[2 * Ppad + i];
            prv[i] = x + (k.damp * (x - px)) + (fx * k.dsm);                                  // :249
            prv[Ppad + i] = y + (k.damp * (y - py)) + (fy * k.dsm);
            prv[2 * Ppad + i] = z + (k.damp * (z - pz)) + (fz * k.dsm);
        }
        __syncthreads();
        { T *t = cur; cur = prv; prv = t; }

        // ---- pinned fix-up (they did not move: undo the swap) + cell keys (cloth.pyx:298-311) ------
        for (int i = tid; i < Psort; i += NT) {
            uint32_t w = 0xFFFFFFFFu;
            if (i < P) {
                if (cnt[i]) {
                    T a0 = cur[i], a1 = cur[Ppad + i], a2 = cur[2 * Ppad + i];
                    cur[i] = prv[i]; cur[Ppad + i] = prv[Ppad + i]; cur[2 * Ppad + i] = prv[2 * Ppad + i];
                    prv[i] = a0; prv[Ppad + i] = a1; prv[2 * Ppad + i] = a2;
                }
                w = (cell_key<T>(k, cur[i], cur[Ppad + i], cur[2 * Ppad + i]) << KEY_SHIFT) | (uint32_t)i;
            }
            sw[i] = w;
        }
        __syncthreads();

        // ---- bitonic sort of (key, index): cells become contiguous runs in ascending point index ---
        for (int kk = 2; kk <= Psort; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (Psort >> 1); t += NT) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const uint32_t a = sw[i], b = sw[l];
                    const bool up = ((i & kk) == 0);
                    if ((a > b) == up) { sw[i] = b; sw[l] = a; }
                }
                __syncthreads();
            }
        }

        // ---- self-collision (cloth.pyx:313-343) + plane (cloth.pyx:345-370), one lane per cell -------
        for (int q = tid; q < P; q += NT) {
            const uint32_t w0 = sw[q];
            const uint32_t key = w0 >> KEY_SHIFT;
            if (q > 0 && (sw[q - 1] >> KEY_SHIFT) == key) continue;     // not the head of its cell
            int end = q + 1;
            while (end < P && (sw[end] >> KEY_SHIFT) == key) end++;
            if (end - q > 1) {
                for (int a = q; a < end; a++) {                         // ascending point index
                    const int i = (int)(sw[a] & HK_NBR_MASK);
                    if (cnt[i]) continue;                               // :314
                    const T xi = cur[i], yi = cur[Ppad + i], zi = cur[2 * Ppad + i];
                    T tx = (T)0, ty = (T)0, tz = (T)0;
                    int n = 0;
                    for (int b = q; b < end; b++) {
                        if (b == a) continue;                           // :325
                        const int j = (int)(sw[b] & HK_NBR_MASK);
                        const T dx = xi - cur[j], dy = yi - cur[Ppad + j], dz = zi - cur[2 * Ppad + j];
                        const T dist = fastnorm<T>(dx, dy, dz);         // :327
                        if (dist <= k.thresh) {                         // :330
                            const T factor = (k.thresh - dist) / dist;  // :331
                            tx += dx * factor; ty += dy * factor; tz += dz * factor;
                            n += 1;
                        }
                    }
                    if (n != 0) {                                       // :336-343
                        const T nf = (T)n;
                        cur[i] = xi + tx / nf / k.sim_steps;
                        cur[Ppad + i] = yi + ty / nf / k.sim_steps;
                        cur[2 * Ppad + i] = zi + tz / nf / k.sim_steps;
                    }
                }
            }
            for (int a = q; a < end; a++) {                             // plane, cloth.pyx:356-370
                const int i = (int)(sw[a] & HK_NBR_MASK);
                if (cnt[i] || cur[2 * Ppad + i] >= k.min_z) continue;
                const T px = prv[i], py = prv[Ppad + i], pz = prv[2 * Ppad + i];
                const T t = (k.min_z - pz) * (T)1.0;
                const T tgx = px + t * (T)(-0.0), tgy = py + t * (T)(-0.0), tgz = pz + t * (T)(-1.0);
                const T gx = tgx + k.surf_off * (T)0.0, gy = tgy + k.surf_off * (T)0.0, gz = tgz + k.surf_off * (T)1.0;
                const T cx = gx - px, cy = gy - py, cz = gz - pz;
                cur[i] = px + cx * k.one_m_fric;
                cur[Ppad + i] = py + cy * k.one_m_fric;
                cur[2 * Ppad + i] = pz + cz * k.one_m_fric;
            }
        }
        __syncthreads();

        // ---- strain limit + tear (cloth.pyx:258-296): wave 0 walks the dependency levels in order --
        if (tid < 64) {
            int tear = 0;
            int off = A.lv_off[0];
            for (int L = 0; L < A.n_levels; L++) {
                const int nxt = A.lv_off[L + 1];
                const int idx = off + tid;
                if (idx < nxt) {
                    const uint32_t en = A.lv_ent[idx];
                    const int a = (int)(en & 0xFFFFu), b = (int)(en >> 16);
                    const bool pa = cnt[a] != 0, pb = cnt[b] != 0;
                    if (!(pa && pb)) {                                                      // :268
                        const T r = rest[idx];
                        const T xa = cur[a], ya = cur[Ppad + a], za = cur[2 * Ppad + a];
                        const T xb = cur[b], yb = cur[Ppad + b], zb = cur[2 * Ppad + b];
                        const T dx = xa - xb, dy = ya - yb, dz = za - zb;
                        const T len = fastnorm<T>(dx, dy, dz);                              // :270
                        if (len > r * k.tear_thresh) tear = 1;                              // :272
                        if (len > (r * k.c11)) {                                            // :275
                            const T ux = dx / len, uy = dy / len, uz = dz / len;            // :276-278
                            const T extra = len - r * k.c11;                                // :279
                            if (pa) {
                                cur[b] = xb + ux * extra; cur[Ppad + b] = yb + uy * extra; cur[2 * Ppad + b] = zb + uz * extra;
                            } else if (pb) {
                                cur[a] = xa - ux * extra; cur[Ppad + a] = ya - uy * extra; cur[2 * Ppad + a] = za - uz * extra;
                            } else {
                                const T ed = extra * (T)0.5;
                                cur[a] = xa - ux * ed; cur[Ppad + a] = ya - uy * ed; cur[2 * Ppad + a] = za - uz * ed;
                                cur[b] = xb + ux * ed; cur[Ppad + b] = yb + uy * ed; cur[2 * Ppad + b] = zb + uz * ed;
                            }
                        }
                    }
                }
                off = nxt;
                // the next level's lanes read what this level's lanes wrote: same wave, LDS is in order;
                // the fence only stops the compiler from moving LDS accesses across the level boundary.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (__any(tear) && tid == 0) misc[0] = 1;
        }
        __syncthreads();
        done++;
        if (sc.break_on_tear && misc[0]) break;                                            // cloth_env.py:511-514
    }

    {   // LDS -> HBM
        T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        for (int i = tid; i < 3 * Ppad; i += NT) { gp[i] = cur[i]; gq[i] = prv[i]; }
        uint8_t *gc = A.cnt + (size_t)e * Ppad;
        for (int i = tid; i < Ppad; i += NT) gc[i] = cnt[i];
        if (tid == 0) { A.tear[e] = misc[0]; A.executed[e] = done; }
    }
}

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
