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
    const uint32_t *lv_ent;  // [Spad]  ptA | ptB<<16, level order
    const uint16_t *lv_off;  // [n_levels+1]
    const uint32_t *lv_rows; // [n_levels] bitmask of the grid-row groups a level touches
    int32_t n_levels;
    int32_t N, P, Ppad, S, Spad;
    int32_t HT, ht_bits;     // spatial hash table slots (power of two > P)
    int32_t lvw_shift;       // lanes per level in the parallel pre-pass = 1 << lvw_shift (16 or 32)
    int32_t rest_stride;     // 0: one shared table
    int32_t phase_mask;      // debug/ablation: bit0 hooke+verlet, bit1 collide, bit2 plane, bit3 strain, bit4 no-skip
    DevConsts<T> k;
};

constexpr int KEY_SHIFT = 12;
constexpr uint32_t KEY_BIAS = 1u << 19;
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr uint8_t CNT_GRAB_MASK = 0x7F, CNT_EXT_PIN = 0x80;
enum { PH_HOOKE = 1, PH_COLLIDE = 2, PH_PLANE = 4, PH_STRAIN = 8, PH_NOSKIP = 16 };

template <typename T> __device__ __forceinline__ T dev_sqrt(T x);
template <> __device__ __forceinline__ double dev_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float dev_sqrt<float>(float x) { return sqrtf(x); }
template <typename T> __device__ __forceinline__ T dev_floor(T x);
template <> __device__ __forceinline__ double dev_floor<double>(double x) { return floor(x); }
template <> __device__ __forceinline__ float dev_floor<float>(float x) { return floorf(x); }
// relative slack of the conservative "could this comparison against a sqrt be true" pre-filters
template <typename T> __device__ __forceinline__ T filt_slack();
template <> __device__ __forceinline__ double filt_slack<double>() { return 1e-9; }
template <> __device__ __forceinline__ float filt_slack<float>() { return 1e-5f; }

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
struct LdsLayout {
    int cur, cnt, ent, rest, off, rows, flag, hkey, hco, memb, slot, misc, total;
    __host__ __device__ LdsLayout(int tsz, int Ppad, int Spad, int nL, int HT, bool tab_lds) {
        int o = 0;
        auto take = [&](int bytes) { int r = o; o += (bytes + 15) / 16 * 16; return r; };
        cur = take(3 * Ppad * tsz);
        cnt = take(Ppad);
        ent = take(tab_lds ? Spad * 4 : 0);
        rest = take(tab_lds ? Spad * tsz : 0);
        off = take(tab_lds ? (nL + 1) * 2 : 0);
        rows = take(tab_lds ? nL * 4 : 0);
        flag = take(nL + 64);
        hkey = take(HT * 4);
        hco = take(HT * 4);          // (fill cursor << 16) | member count
        memb = take(Ppad * 2);
        slot = take(Ppad * 2);
        misc = take(256);
        total = o;
    }
};

// One spring of the strain limiter, exactly as cloth.pyx:265-296 evaluates it. Returns true if a correction
// was applied. `tear` is OR-ed.
template <typename T>
__device__ __forceinline__ bool strain_spring(T *cur, const uint8_t *cnt, int Ppad, uint32_t en, T r,
                                              const DevConsts<T> &k, int &tear) {
    const int a = (int)(en & 0xFFFFu), b = (int)(en >> 16);
    const bool pa = cnt[a] != 0, pb = cnt[b] != 0;
    if (pa && pb) return false;                                                         // :268
    const T xa = cur[a], ya = cur[Ppad + a], za = cur[2 * Ppad + a];
    const T xb = cur[b], yb = cur[Ppad + b], zb = cur[2 * Ppad + b];
    const T dx = xa - xb, dy = ya - yb, dz = za - zb;
    const T len2 = dx * dx + dy * dy + dz * dz;
    const T t11 = r * k.c11, tt = r * k.tear_thresh;
    const T tmin = t11 < tt ? t11 : tt;
    if (!(len2 > tmin * tmin * ((T)1 - filt_slack<T>()))) return false;   // certainly neither tear nor stretch
    const T len = dev_sqrt<T>(len2);                                                    // :270
    if (len > tt) tear = 1;                                                             // :272
    if (!(len > t11)) return false;                                                     // :275
    const T ux = dx / len, uy = dy / len, uz = dz / len;                                // :276-278
    const T extra = len - t11;                                                          // :279
    if (pa) {
        cur[b] = xb + ux * extra; cur[Ppad + b] = yb + uy * extra; cur[2 * Ppad + b] = zb + uz * extra;
    } else if (pb) {
        cur[a] = xa - ux * extra; cur[Ppad + a] = ya - uy * extra; cur[2 * Ppad + a] = za - uz * extra;
    } else {
        const T ed = extra * (T)0.5;
        cur[a] = xa - ux * ed; cur[Ppad + a] = ya - uy * ed; cur[2 * Ppad + a] = za - uz * ed;
        cur[b] = xb + ux * ed; cur[Ppad + b] = yb + uy * ed; cur[2 * Ppad + b] = zb + uz * ed;
    }
    return true;
}

// Particle i is owned by thread (i % NT); a thread owns PPT particles i = tid + k*NT. The previous
// position of a particle is only ever touched by its owner (adjust, Verlet, plane), so it lives in the
// owner's registers for the whole schedule; only the current positions are shared through LDS.
template <typename T, int NT, int PPT, bool TAB_LDS>
__global__ __launch_bounds__(NT) void k_run_schedule(StepArgs<T> A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const ClothSchedule sc = A.sched[e];
    if (!sc.active || sc.n_total <= 0) {
        if (tid == 0) A.executed[e] = 0;
        return;
    }
    const int P = A.P, Ppad = A.Ppad, nL = A.n_levels, HT = A.HT;
    const LdsLayout lay((int)sizeof(T), Ppad, A.Spad, nL, HT, TAB_LDS);
    T *cur = reinterpret_cast<T *>(smem + lay.cur);
    uint8_t *cnt = smem + lay.cnt;
    uint8_t *lvflag = smem + lay.flag;
    uint32_t *hkey = reinterpret_cast<uint32_t *>(smem + lay.hkey);
    uint32_t *hco = reinterpret_cast<uint32_t *>(smem + lay.hco);
    uint16_t *memb = reinterpret_cast<uint16_t *>(smem + lay.memb);
    uint16_t *slot = reinterpret_cast<uint16_t *>(smem + lay.slot);
    volatile int *misc = reinterpret_cast<volatile int *>(smem + lay.misc);   // [0] tear, [1] strain-active, [8..] scan
    const DevConsts<T> k = A.k;
    const T *g_rest = A.rest + (size_t)e * A.rest_stride;
    const uint32_t *ent = TAB_LDS ? reinterpret_cast<const uint32_t *>(smem + lay.ent) : A.lv_ent;
    const T *rest = TAB_LDS ? reinterpret_cast<const T *>(smem + lay.rest) : g_rest;
    const uint16_t *loff = TAB_LDS ? reinterpret_cast<const uint16_t *>(smem + lay.off) : A.lv_off;
    const uint32_t *lrows = TAB_LDS ? reinterpret_cast<const uint32_t *>(smem + lay.rows) : A.lv_rows;
    const int pm = A.phase_mask;

    T pvx[PPT], pvy[PPT], pvz[PPT];         // previous positions of the owned particles
    {   // HBM -> LDS / registers, coalesced
        const T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        for (int i = tid; i < 3 * Ppad; i += NT) cur[i] = gp[i];
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * NT;
            const bool ok = i < P;
            pvx[q] = ok ? gq[i] : (T)0; pvy[q] = ok ? gq[Ppad + i] : (T)0; pvz[q] = ok ? gq[2 * Ppad + i] : (T)0;
        }
        const uint8_t *gc = A.cnt + (size_t)e * Ppad;
        for (int i = tid; i < Ppad; i += NT) cnt[i] = gc[i];
        if (TAB_LDS) {
            uint32_t *d0 = reinterpret_cast<uint32_t *>(smem + lay.ent);
            T *d1 = reinterpret_cast<T *>(smem + lay.rest);
            for (int i = tid; i < A.Spad; i += NT) { d0[i] = A.lv_ent[i]; d1[i] = g_rest[i]; }
            uint16_t *d2 = reinterpret_cast<uint16_t *>(smem + lay.off);
            uint32_t *d3 = reinterpret_cast<uint32_t *>(smem + lay.rows);
            for (int i = tid; i <= nL; i += NT) d2[i] = A.lv_off[i];
            for (int i = tid; i < nL; i += NT) d3[i] = A.lv_rows[i];
        }
        for (int h = tid; h < HT; h += NT) { hkey[h] = KEY_EMPTY; hco[h] = 0; }
        if (tid == 0) { misc[0] = A.tear[e]; misc[1] = 0; }
    }
    __syncthreads();

    const T dz_up = (T)sc.dz_up, dxp = (T)sc.dx_pull, dyp = (T)sc.dy_pull, dzp = (T)sc.dz_pull;
    const int W = 1 << A.lvw_shift;
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
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const int m = cnt[i] & CNT_GRAB_MASK;
                for (int r = 0; r < m; r++) {           // gripper.pyx:60-66: p <- x ; x <- delta + x
                    const T x = cur[i], y = cur[Ppad + i], z = cur[2 * Ppad + i];
                    pvx[q] = x; pvy[q] = y; pvz[q] = z;
                    cur[i] = ax + x; cur[Ppad + i] = ay + y; cur[2 * Ppad + i] = az + z;
                }
            }
            __syncthreads();
        } else if (mode == 2) {
            bool any = false;
            for (int i = tid; i < P; i += NT)          // gripper.pyx:68-73
                if (cnt[i] & CNT_GRAB_MASK) { cnt[i] = 0; any = true; }
            (void)any;
            __syncthreads();
        }

        // ---- gravity + Hooke gather + Verlet (cloth.pyx:216-256) ----------------------------------
        if (pm & PH_HOOKE) {
            T nx[PPT], ny[PPT], nz[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                nx[q] = ny[q] = nz[q] = (T)0;
                if (i >= P || cnt[i]) continue;             // pinned: Verlet skips it (cloth.pyx:244)
                const T x = cur[i], y = cur[Ppad + i], z = cur[2 * Ppad + i];
                T fx = (T)0 + (T)0, fy = (T)0 + (T)0, fz = (T)0 + k.mg;
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
                nx[q] = x + (k.damp * (x - pvx[q])) + (fx * k.dsm);                               // :249
                ny[q] = y + (k.damp * (y - pvy[q])) + (fy * k.dsm);
                nz[q] = z + (k.damp * (z - pvz[q])) + (fz * k.dsm);
                pvx[q] = x; pvy[q] = y; pvz[q] = z;                                               // :256
            }
            __syncthreads();                                // every neighbour read of the old positions is done
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P || cnt[i]) continue;
                cur[i] = nx[q]; cur[Ppad + i] = ny[q]; cur[2 * Ppad + i] = nz[q];                 // :255
            }
        }

        // ---- spatial map (cloth.pyx:298-311): hash table in LDS keyed by the exact cell key, CSR member
        // lists; the order inside a cell is restored to ascending point index by the cell's lane.
        if (pm & PH_COLLIDE) {
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const uint32_t key = cell_key<T>(k, cur[i], cur[Ppad + i], cur[2 * Ppad + i]);    // own slots: no hazard
                uint32_t h = (key * 2654435761u) >> (32 - A.ht_bits);
                while (true) {
                    const uint32_t old = atomicCAS(&hkey[h], KEY_EMPTY, key);
                    if (old == KEY_EMPTY || old == key) break;
                    h = (h + 1) & (uint32_t)(HT - 1);
                }
                slot[i] = (uint16_t)h;
                atomicAdd(&hco[h], 1u);
            }
            __syncthreads();
            // exclusive prefix sum of the slot counts -> fill cursors (each thread owns HT/NT consecutive slots)
            const int per = HT / NT;
            uint32_t loc = 0;
            for (int q = 0; q < per; q++) loc += hco[tid * per + q];
            uint32_t inc = loc;
            for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
            if (lane == 63) misc[8 + (tid >> 6)] = (int)inc;
            __syncthreads();
            uint32_t base = inc - loc;
            for (int w = 0; w < (tid >> 6); w++) base += (uint32_t)misc[8 + w];
            for (int q = 0; q < per; q++) {
                const uint32_t c = hco[tid * per + q];
                hco[tid * per + q] = (base << 16) | c;
                base += c;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const uint32_t old = atomicAdd(&hco[slot[i]], 1u << 16);   // afterwards cursor = END of the cell
                memb[old >> 16] = (uint16_t)i;
            }
            __syncthreads();
            // ---- self-collision (cloth.pyx:313-343), one lane per cell, Gauss-Seidel in ascending index ----
            const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
            for (int h = tid; h < HT; h += NT) {
                const uint32_t co = hco[h];
                const int n = (int)(co & 0xFFFFu);
                hco[h] = 0;                                             // ready for the next substep
                if (n == 0) continue;
                hkey[h] = KEY_EMPTY;
                if (n == 1) continue;
                uint16_t *m = memb + ((int)(co >> 16) - n);
                for (int a = 1; a < n; a++) {                           // restore ascending point index
                    const uint16_t v = m[a];
                    int b = a - 1;
                    while (b >= 0 && m[b] > v) { m[b + 1] = m[b]; b--; }
                    m[b + 1] = v;
                }
                for (int a = 0; a < n; a++) {
                    const int i = (int)m[a];
                    if (cnt[i]) continue;                               // :314
                    const T xi = cur[i], yi = cur[Ppad + i], zi = cur[2 * Ppad + i];
                    T tx = (T)0, ty = (T)0, tz = (T)0;
                    int nh = 0;
                    for (int b = 0; b < n; b++) {
                        if (b == a) continue;                           // :325
                        const int j = (int)m[b];
                        const T dx = xi - cur[j], dy = yi - cur[Ppad + j], dz = zi - cur[2 * Ppad + j];
                        const T d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 > thr2) continue;                        // certainly dist > thresh
                        const T dist = dev_sqrt<T>(d2);                 // :327
                        if (dist <= k.thresh) {                         // :330
                            const T factor = (k.thresh - dist) / dist;  // :331
                            tx += dx * factor; ty += dy * factor; tz += dz * factor;
                            nh += 1;
                        }
                    }
                    if (nh != 0) {                                      // :336-343
                        const T nf = (T)nh;
                        cur[i] = xi + tx / nf / k.sim_steps;
                        cur[Ppad + i] = yi + ty / nf / k.sim_steps;
                        cur[2 * Ppad + i] = zi + tz / nf / k.sim_steps;
                    }
                }
            }
        }
        __syncthreads();

        // ---- plane (cloth.pyx:345-370), by the owner (it holds the previous position) --------------------
        if (pm & PH_PLANE) {
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P || cnt[i] || cur[2 * Ppad + i] >= k.min_z) continue;
                const T px = pvx[q], py = pvy[q], pz = pvz[q];
                const T t = (k.min_z - pz) * (T)1.0;
                const T tgx = px + t * (T)(-0.0), tgy = py + t * (T)(-0.0), tgz = pz + t * (T)(-1.0);
                const T gx = tgx + k.surf_off * (T)0.0, gy = tgy + k.surf_off * (T)0.0, gz = tgz + k.surf_off * (T)1.0;
                const T cx = gx - px, cy = gy - py, cz = gz - pz;
                cur[i] = px + cx * k.one_m_fric;
                cur[Ppad + i] = py + cy * k.one_m_fric;
                cur[2 * Ppad + i] = pz + cz * k.one_m_fric;
            }
            __syncthreads();
        }

        // ---- strain limit + tear (cloth.pyx:258-296) ---------------------------------------------------
        // (1) all threads: which levels hold a spring that would stretch/tear at the CURRENT positions?
        //     A spring untouched by earlier corrections of the sweep behaves exactly as evaluated here.
        // (2) wave 0 walks the dependency levels in order, executing only levels that are flagged or touch a
        //     grid row already modified by the sweep; everything it skips is provably a no-op.
        if (pm & PH_STRAIN) {
            {
                const int sub = tid & (W - 1), grp = tid >> A.lvw_shift, G = NT >> A.lvw_shift;
                const int gsh = (lane >> A.lvw_shift) << A.lvw_shift;
                const unsigned long long gm = (W == 64 ? ~0ull : ((1ull << W) - 1ull)) << gsh;
                int any_wave = 0;
                for (int L0 = 0; L0 < nL; L0 += G) {
                    const int L = L0 + grp;
                    bool act = false;
                    if (L < nL) {
                        const int idx = (int)loff[L] + sub;
                        if (idx < (int)loff[L + 1]) {
                            const uint32_t en = ent[idx];
                            const int a = (int)(en & 0xFFFFu), b = (int)(en >> 16);
                            if (!(cnt[a] && cnt[b])) {
                                const T r = rest[idx];
                                const T dx = cur[a] - cur[b], dy = cur[Ppad + a] - cur[Ppad + b], dz = cur[2 * Ppad + a] - cur[2 * Ppad + b];
                                const T len2 = dx * dx + dy * dy + dz * dz;
                                const T t11 = r * k.c11, tt = r * k.tear_thresh;
                                const T tmin = t11 < tt ? t11 : tt;
                                act = len2 > tmin * tmin * ((T)1 - filt_slack<T>());
                            }
                        }
                    }
                    const unsigned long long bal = __ballot(act);
                    if (sub == 0 && L < nL) lvflag[L] = (bal & gm) ? 1 : 0;
                    any_wave |= (bal != 0ull);
                }
                if (any_wave && lane == 0) misc[1] = 1;
            }
            __syncthreads();
            if (tid < 64 && (misc[1] || (pm & PH_NOSKIP))) {
                int tear = 0;
                uint32_t dirty = (pm & PH_NOSKIP) ? 0xFFFFFFFFu : 0u;
                for (int L0 = 0; L0 < nL; L0 += 64) {
                    const int Lm = L0 + lane;
                    const bool valid = Lm < nL;
                    const uint32_t myrows = valid ? lrows[Lm] : 0u;
                    const int myoff = valid ? (int)loff[Lm] : 0, myoff1 = valid ? (int)loff[Lm + 1] : 0;
                    const unsigned long long amask = __ballot(valid && lvflag[Lm] != 0);
                    int j = 0;
                    while (j < 64) {
                        const unsigned long long rmask = __ballot((myrows & dirty) != 0u);
                        const unsigned long long need = (amask | rmask) & (~0ull << j);
                        if (!need) break;
                        j = __builtin_amdgcn_readfirstlane(__ffsll((long long)need) - 1);
                        const int o0 = __builtin_amdgcn_readlane(myoff, j), o1 = __builtin_amdgcn_readlane(myoff1, j);
                        const int idx = o0 + lane;
                        bool trig = false;
                        if (idx < o1) trig = strain_spring<T>(cur, cnt, Ppad, ent[idx], rest[idx], k, tear);
                        if (__any(trig)) dirty |= (uint32_t)__builtin_amdgcn_readlane((int)myrows, j);
                        j++;
                        // the next level's lanes read what this level's lanes wrote: same wave, LDS is in order;
                        // the fences only stop the compiler from moving LDS accesses across the level boundary.
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
                if (__any(tear) && lane == 0) misc[0] = 1;
                if (lane == 0) misc[1] = 0;
            }
            __syncthreads();
        }
        done++;
        if (sc.break_on_tear && misc[0]) break;                                            // cloth_env.py:511-514
    }

    {   // LDS / registers -> HBM
        T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        for (int i = tid; i < 3 * Ppad; i += NT) gp[i] = cur[i];
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * NT;
            if (i < P) { gq[i] = pvx[q]; gq[Ppad + i] = pvy[q]; gq[2 * Ppad + i] = pvz[q]; }
        }
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
