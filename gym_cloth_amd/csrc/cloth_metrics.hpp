// cloth_metrics.hpp -- what the episode loop computes between substep runs: coverage (monotone-chain hull + shoelace, in double),
// z-variance, out-of-bounds, height count (cloth_env.py:1020-1098) and the MT19937 skip of the domain-randomisation draws.
#pragma once

#include "cloth_common.hpp"

namespace clothhip {

// Advance env's MT19937 stream (global memory, numpy layout) by n words with the whole workgroup: the twist of the 624-word
// state is done in its three dependency phases, one word per thread (mt19937_gen's sequential in-place semantics: phase A
// reads old words only, phases B / C read the new words of the previous phase). All threads must call it.
template <int NT>
__device__ __forceinline__ void mt_skip_block(uint32_t *mt, uint64_t n, int tid) {
    static_assert(NT >= 256, "one word per thread and phase");
    __syncthreads();
    uint32_t pos = mt[MT_N];
    while (n > 0) {
        if (pos >= (uint32_t)MT_N) {
            const int lo[3] = {0, MT_N - MT_M, 2 * (MT_N - MT_M)}, hi[3] = {MT_N - MT_M, 2 * (MT_N - MT_M), MT_N - 1};
            for (int ph = 0; ph < 3; ph++) {
                const int i = lo[ph] + tid;
                uint32_t v = 0;
                const bool on = i < hi[ph];
                if (on) v = mt_twist_word(mt[i], mt[i + 1], ph == 0 ? mt[i + MT_M] : mt[i + (MT_M - MT_N)]);
                __syncthreads();
                if (on) mt[i] = v;
                __syncthreads();
            }
            if (tid == 0) mt[MT_N - 1] = mt_twist_word(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
            __syncthreads();
            pos = 0;
        }
        const uint64_t take = n < (uint64_t)(MT_N - pos) ? n : (uint64_t)(MT_N - pos);
        pos += (uint32_t)take;
        n -= take;
    }
    if (tid == 0) mt[MT_N] = pos;
    __syncthreads();
}

// ---- per-env metrics (cloth_env.py:1020-1098): coverage = area of the convex hull of the clipped (x,y) (same
// monotone-chain + shoelace arithmetic, in double, as clothhip_hull_area on the host), variance_inv of z, out-of-bounds,
// #(z < thickness/2). ONE workgroup of NT threads; `src(i, x, y, z)` yields particle i as doubles. Scratch (LDS):
// sx/sy[NS] sort buffers (the handle's precision) + hx/hy[NH] hull stack of doubles (NH >= P + 2) + 64 doubles =
// 2 NS sizeof(K) + (2 NH + 64) * 8 bytes (HULL_IDX: see below).
// The reductions are done by the first 256 threads in a fixed tree, so the result does not depend on NT: the stand-alone
// kernel (256 threads) and the in-kernel call of the episode stepper give the same bits.
// Results: out[0] coverage, out[1] variance_inv, out[2] out-of-bounds (0/1), out[3] #(z < half_thick); valid for ALL
// threads on return (the function ends with a barrier).
// HULL_IDX: the hull stack holds u16 INDICES into the sorted, de-duplicated points instead of their coordinates as doubles -- the chain's
// arithmetic reads the same (double)sx / (double)sy values either way, at an eighth of the LDS: the variants whose LDS is tight take
// it (two large-grid cloths per CU, five / six 25x25 cloths per CU); scratch = 2 NS sizeof(K) + 512 + 2 NH bytes then.
template <int NT, typename K, typename Src, bool HULL_IDX = false>
__device__ __forceinline__ void metrics_block(const Src &src, int P, int NS, int NH, unsigned char *scr, int tid, double half_thick,
                                              double out[4]) {
    K *sx = reinterpret_cast<K *>(scr), *sy = sx + NS;
    double *hx = reinterpret_cast<double *>(sy + NS), *hy = hx + (HULL_IDX ? 0 : NH);
    double *red = HULL_IDX ? hx : hy + NH;                    // [64] reduction scratch
    uint16_t *hs = reinterpret_cast<uint16_t *>(red + 64);    // HULL_IDX: [NH] hull stack of indices
    const int lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    double mnx = INF, mxx = -INF, mny = INF, mxy = -INF, mnz = INF, mxz = -INF, sum = 0.0;
    int nlow = 0;                                               // compute_height (cloth_env.py:603-609): #(z < thickness/2)
    if (NT == 256 || tid < 256) {
        for (int i = tid; i < NS; i += 256) {
            double x = INF, y = INF;
            if (i < P) {
                double z;
                src(i, x, y, z);
                nlow += z < half_thick ? 1 : 0;
                mnx = fmin(mnx, x); mxx = fmax(mxx, x); mny = fmin(mny, y); mxy = fmax(mxy, y);
                mnz = fmin(mnz, z); mxz = fmax(mxz, z); sum += z;
                x = fmin(fmax(x, 0.0), 1.0); y = fmin(fmax(y, 0.0), 1.0);                         // cloth_env.py:629
            }
            sx[i] = (K)x; sy[i] = (K)y;
        }
    }
    // block reductions (min/max exact; the z-sum order differs from numpy's pairwise sum only in the last bits)
    auto wred = [&](double v, int op) {
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(v, o);
            v = op == 0 ? fmin(v, w) : (op == 1 ? fmax(v, w) : v + w);
        }
        return v;
    };
    double vals[7] = {mnx, mxx, mny, mxy, mnz, mxz, sum};
    const int ops[7] = {0, 1, 0, 1, 0, 1, 2};
    if (NT == 256 || tid < 256)
        for (int q = 0; q < 7; q++) { const double r = wred(vals[q], ops[q]); if (lane == 0) red[q * 4 + wave] = r; }
    __syncthreads();
    for (int q = 0; q < 7; q++) {
        double r = red[q * 4];
        for (int w = 1; w < 4; w++) r = ops[q] == 0 ? fmin(r, red[q * 4 + w]) : (ops[q] == 1 ? fmax(r, red[q * 4 + w]) : r + red[q * 4 + w]);
        vals[q] = r;
    }
    __syncthreads();
    const double mean = vals[6] / P;
    if (NT == 256 || tid < 256) {
        double acc = 0.0;
        for (int i = tid; i < P; i += 256) { double x, y, z; src(i, x, y, z); const double d = z - mean; acc += d * d; }
        acc = wred(acc, 2);
        if (lane == 0) red[wave] = acc;
        for (int o = 32; o > 0; o >>= 1) nlow += __shfl_xor(nlow, o);
        if (lane == 0) reinterpret_cast<int *>(red + 32)[wave] = nlow;
    }
    // bitonic sort of the clipped points, lexicographic (x, y); padding (+inf,+inf) sinks to the end
    for (int kk = 2; kk <= NS; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < (NS >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const K ax = sx[i], ay = sy[i], bx = sx[l], by = sy[l];
                const bool gt = ax > bx || (ax == bx && ay > by);
                if (gt == ((i & kk) == 0)) { sx[i] = bx; sy[i] = by; sx[l] = ax; sy[l] = ay; }
            }
        }
    __syncthreads();
    if (tid == 0) {
        const double var = (red[0] + red[1] + red[2] + red[3]) / P;                            // np.var
        red[41] = var < 0.000001 ? 1000.0 : 0.001 / var;                                       // cloth_env.py:1081-1084
        const int *nl = reinterpret_cast<const int *>(red + 32);
        red[43] = (double)(nl[0] + nl[1] + nl[2] + nl[3]);
        const double slack = 0.25;                                                             // cloth_env.py:1031-1036
        red[42] = (vals[1] >= 1.0 + slack || vals[0] < -slack || vals[3] >= 1.0 + slack || vals[2] < -slack ||
                   vals[5] >= 1.0 || vals[4] < 0) ? 1.0 : 0.0;
        // dedupe (in place), then Andrew's monotone chain exactly as clothhip_hull_area
        int m = 0;
        for (int i = 0; i < P; i++)
            if (m == 0 || sx[i] != sx[m - 1] || sy[i] != sy[m - 1]) { sx[m] = sx[i]; sy[m] = sy[i]; m++; }
        double area = 0.0;
        if (m >= 3) {
            auto cross = [](double ox, double oy, double ax, double ay, double bx, double by) {
                return (ax - ox) * (by - oy) - (ay - oy) * (bx - ox);
            };
            auto HX = [&](int q) -> double { if constexpr (HULL_IDX) return (double)sx[hs[q]]; else return hx[q]; };
            auto HY = [&](int q) -> double { if constexpr (HULL_IDX) return (double)sy[hs[q]]; else return hy[q]; };
            auto PUSH = [&](int q, int i) { if constexpr (HULL_IDX) hs[q] = (uint16_t)i; else { hx[q] = (double)sx[i]; hy[q] = (double)sy[i]; } };
            int k = 0;
            for (int i = 0; i < m; i++) {
                while (k >= 2 && cross(HX(k - 2), HY(k - 2), HX(k - 1), HY(k - 1), (double)sx[i], (double)sy[i]) <= 0) k--;
                PUSH(k, i); k++;
            }
            for (int i = m - 2, t = k + 1; i >= 0; i--) {
                while (k >= t && cross(HX(k - 2), HY(k - 2), HX(k - 1), HY(k - 1), (double)sx[i], (double)sy[i]) <= 0) k--;
                PUSH(k, i); k++;
            }
            k--;
            if (k >= 3) {
                double a2 = 0.0;
                for (int i = 0; i < k; i++) {
                    const int n = (i + 1) % k;
                    a2 += (HX(i) - HX(0)) * (HY(n) - HY(0)) - (HX(n) - HX(0)) * (HY(i) - HY(0));
                }
                area = 0.5 * fabs(a2);
            }
        }
        red[40] = area;
    }
    __syncthreads();
    out[0] = red[40]; out[1] = red[41]; out[2] = red[42]; out[3] = red[43];
    __syncthreads();
}

}  // namespace clothhip
