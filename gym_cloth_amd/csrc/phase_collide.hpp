// phase_collide.hpp -- self-collision of one spatial cell (cloth.pyx:313-343) in the reference's Gauss-Seidel order: a whole wave per large
// cell, four small cells per wave in 16-lane groups, one lane for cells of more than 64 members.
#pragma once

#include "cloth_common.hpp"

namespace clothhip {

// Self-collision of ONE spatial cell (cloth.pyx:313-343) by a whole wave, exact Gauss-Seidel order:
// lane b holds the cell's b-th member in ascending point index; members are visited serially in that order
// each against all lanes in parallel; the hits are summed in ascending member order. n <= 64.
template <typename T>
__device__ __forceinline__ int collide_cell_wave(Pt<T> *cur, uint16_t *m, const uint16_t *slot, int n,
                                                 const DevConsts<T> &k, int lane) {
    int visits_ = 0, hits_ = 0;                      // (profiling builds only read them)
    const bool in = lane < n;
    const int mine = in ? (int)m[lane] : 0x7fff;
    int rank = 0;
    // (four members per trip: the lanes behind the last member hold 0x7fff, which is below nobody; n <= 64)
    for (int t = 0; t < n; t += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) rank += (__builtin_amdgcn_readlane(mine, t + u) < mine) ? 1 : 0;
    }
    // lane r takes the member of rank r: one pass through the LDS crossbar (the list in LDS stays as the fill left it: nobody
    // reads it after the sweeps)
    const int srt_ = __builtin_amdgcn_ds_permute((in ? rank : lane) << 2, mine);
    const int i = in ? srt_ : 0;
    const Pt<T> me = cur[i];
    T x = me.x, y = me.y, z = me.z;
    // Members to visit, in ascending order: the SEEDS (unpinned members that have a hit at the positions the phase
    // started from, flagged by the parallel pre-check) and, dynamically, every later unpinned member that is within
    // the candidate radius of a member that actually MOVED: a move displaces a particle by at most thresh/steps, so
    // anyone farther than thresh*(1+2/steps) from the mover's old position cannot be hit by it. All other members
    // provably collect no hit at their turn (cloth.pyx:330 never true) and are skipped without changing the result.
    const bool free_ = in && w_cnt(me.w) == 0;
    unsigned long long todo = ballot64(free_ && (slot[i] & 0x8000u) != 0);
    const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
    const T cfac = (T)1 + (T)2 / k.sim_steps;
    const T thr2c = thr2 * cfac * cfac;
    bool moved = false;
#ifndef CLOTHHIP_SERIAL_HITSUM
    if constexpr (sizeof(T) == 4) {
        // fp32: the lane predicates of a visit as wave masks in scalar registers (one compare each; the conjunctions, "not the visited
        // member", "later than it" are scalar bit operations), the exact test without a branch around it (a big cell nearly always
        // has a candidate), the selects straight from the masks. Same arithmetic per lane, same visiting order.
        const unsigned long long inm = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        const unsigned long long freem = ballot64(free_);
        unsigned long long movedm = 0ull;
        while (todo) {
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 2
            // MUTANT 2 (tools/run_mutants.sh; never a product build): the first visit of a cell goes to the SECOND member due, the first one
            // follows -- cloth.pyx:324-343 visits in ascending point index
            const bool mut2_ = visits_ == 0 && (todo & (todo - 1ull)) != 0ull;
            const unsigned long long pick_ = mut2_ ? (todo & (todo - 1ull)) : todo;
            const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)pick_) - 1);
            todo &= ~(1ull << a);
#else
            const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1ull;
#endif
            visits_++;
            const T xa = bcast(x, a), ya = bcast(y, a), za = bcast(z, a);
            const T dx = xa - x, dy = ya - y, dz = za - z;
            const T d2 = sumsq<T>(dx, dy, dz);
            // (round 5, measured: ONE scalar decision per visit -- the sqrt for every lane, no pre-filter ballot -- 21.09 vs 21.15 M/s: not kept)
            const unsigned long long hm0 = ballot64(!(d2 > thr2)) & inm & ~(1ull << a);
            if (!hm0) continue;
            const T dist = dev_sqrt<T>(d2);                                             // :327
            const unsigned long long hm = ballot64(dist <= k.thresh) & hm0;             // :330
            if (!hm) continue;
            // (the PRODUCT is selected, not the factor: a member that is no hit may hold a non-finite coordinate -- a blown-up particle --
            //  and inf * 0 would carry it into the visited particle's sum; the reference reads the hits only, :330-334)
            const bool hl = __builtin_amdgcn_inverse_ballot_w64(hm);
            const T factor = dev_div<T>(k.thresh - dist, dist);                                                   // :331
            const T tx = wave_sum_f32(hl ? dx * factor : (T)0), ty = wave_sum_f32(hl ? dy * factor : (T)0), tz = wave_sum_f32(hl ? dz * factor : (T)0);
            const int nh = __builtin_popcount((uint32_t)hm) + __builtin_popcount((uint32_t)(hm >> 32));    // (two 32-bit counts: the 64-bit
                                                                         // one reached the float conversion as a 64-bit integer, seven instructions)
            hits_ += nh;
            const T nf = (T)nh;                                                         // :336-343
            const T nxa = xa + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps);
            const T nya = ya + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps);
            const T nza = za + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps);
            if (__builtin_amdgcn_inverse_ballot_w64(1ull << a)) { x = nxa; y = nya; z = nza; }
            movedm |= 1ull << a;
            todo |= ballot64(!(d2 > thr2c)) & freem & ~((2ull << a) - 1ull);            // a moved: later neighbours must look
        }
        moved = __builtin_amdgcn_inverse_ballot_w64(movedm);
    } else
#endif
    {
    while (todo) {
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 2
        const bool mut2_ = visits_ == 0 && (todo & (todo - 1ull)) != 0ull;      // MUTANT 2 (see the fp32 loop above)
        const unsigned long long pick_ = mut2_ ? (todo & (todo - 1ull)) : todo;
        const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)pick_) - 1);
        todo &= ~(1ull << a);
#else
        const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        todo &= todo - 1ull;
#endif
        visits_++;
        const T xa = bcast(x, a), ya = bcast(y, a), za = bcast(z, a);
        const T dx = xa - x, dy = ya - y, dz = za - z;
        const T d2 = sumsq<T>(dx, dy, dz);
        bool hit = in && lane != a && !(d2 > thr2);
        T fx = (T)0, fy = (T)0, fz = (T)0;
        if (hit) {
            const T dist = dev_sqrt<T>(d2);                                             // :327
            hit = dist <= k.thresh;                                                     // :330
            const T factor = hit ? dev_div<T>(k.thresh - dist, dist) : (T)0;            // :331
            fx = dx * factor; fy = dy * factor; fz = dz * factor;
        }
        unsigned long long hm = ballot64(hit);
        if (!hm) continue;
        T tx = (T)0, ty = (T)0, tz = (T)0;
        int nh = 0;
#ifndef CLOTHHIP_SERIAL_HITSUM
        if constexpr (sizeof(T) == 4) {
            // fp32 (parity is a tolerance): the hits' contributions (zero in the other lanes) summed by a DPP tree instead of one by
            // one in ascending order -- the Gauss-Seidel visiting order is untouched, only the association of this one sum differs
            tx = wave_sum_f32(fx); ty = wave_sum_f32(fy); tz = wave_sum_f32(fz);
            nh = __builtin_popcount((uint32_t)hm) + __builtin_popcount((uint32_t)(hm >> 32));    // (two 32-bit counts: the 64-bit one reached
                                                                                                   //  the float conversion as a 64-bit integer, seven instructions)
        } else
#endif
        while (hm) {                                                                    // ascending candidate order
            const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)hm) - 1);
            tx += bcast(fx, b); ty += bcast(fy, b); tz += bcast(fz, b);
            nh++;
            hm &= hm - 1ull;
        }
        hits_ += nh;
        const T nf = (T)nh;                                                             // :336-343
        const T nxa = xa + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps);
        const T nya = ya + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps);
        const T nza = za + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps);
        if (lane == a) { x = nxa; y = nya; z = nza; moved = true; }
        todo |= ballot64(free_ && lane > a && !(d2 > thr2c));                           // a moved: later neighbours must look
    }
    }
    if (moved) cur[i] = Pt<T>{x, y, z, me.w};
    return visits_ | (hits_ << 16);
}

// 64/GSZ cells of at most GSZ (16 or 32) members each at once, one per GSZ-lane group of the wave; same exact
// Gauss-Seidel semantics as collide_cell_wave, with group-local broadcasts through ds_bpermute. `hs` = this lane's
// group's hash slot (or -1: no cell for this group).
template <typename T, int GSZ>
__device__ __forceinline__ void collide_cells_group(Pt<T> *cur, uint16_t *memb, const uint16_t *slot,
                                                      const uint32_t *hco, int hs, const DevConsts<T> &k, int lane) {
    const int sub = lane & (GSZ - 1), base = lane & ~(GSZ - 1), gsh = base;   // my group's lanes are [base, base+GSZ)
    constexpr unsigned long long GM = GSZ == 32 ? 0xFFFFFFFFull : 0xFFFFull;
    const bool gvalid = hs >= 0;
    const uint32_t co = gvalid ? hco[hs] : 0u;
    const int n = (int)(co & 0xFFFFu);
    const int start = (int)(co >> 16);
    const bool in = gvalid && sub < n;
    const int mine = in ? (int)memb[start + sub] : 0x7fff;
    int rank = 0;
#pragma unroll 4
    for (int t = 0; t < GSZ; t++) rank += (lane_pull(mine, base + t) < mine) ? 1 : 0;
    // members get ranks 0..n-1 (ascending index); the other lanes of the group keep their own position (>= n)
    const int i = __builtin_amdgcn_ds_permute((base + (in ? rank : sub)) << 2, in ? mine : 0);
    const bool ins = gvalid && sub < n;                                  // after the permutation lane sub < n holds rank sub
    const Pt<T> me = cur[ins ? i : 0];
    T x = me.x, y = me.y, z = me.z;
    const bool free_ = ins && w_cnt(me.w) == 0;
    const bool want = free_ && (slot[ins ? i : 0] & 0x8000u) != 0;
    unsigned int todo = (unsigned int)((ballot64(want) >> gsh) & GM);          // my group's seeds
    const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
    const T cfac = (T)1 + (T)2 / k.sim_steps;
    const T thr2c = thr2 * cfac * cfac;
    bool moved = false;
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 2
    bool first_ = true;                       // MUTANT 2 (see collide_cell_wave): the small cells' first visit too
#endif
    while (__any(todo != 0u)) {
        const bool act = todo != 0u;
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 2
        const bool mut2_ = first_ && (todo & (todo - 1u)) != 0u;
        const unsigned int pick_ = mut2_ ? (todo & (todo - 1u)) : todo;
        const int a = act ? __ffs((int)pick_) - 1 : 0;
        todo &= ~(act ? (1u << a) : 0u);
        first_ = false;
#else
        const int a = act ? __ffs((int)todo) - 1 : 0;
        todo &= todo - 1u;
#endif
        const T xa = lane_pull(x, base + a), ya = lane_pull(y, base + a), za = lane_pull(z, base + a);
        const T dx = xa - x, dy = ya - y, dz = za - z;
        const T d2 = sumsq<T>(dx, dy, dz);
        bool hit = act && ins && sub != a && !(d2 > thr2);
        T fx = (T)0, fy = (T)0, fz = (T)0;
        if (hit) {
            const T dist = dev_sqrt<T>(d2);                                             // :327
            hit = dist <= k.thresh;                                                     // :330
            const T factor = hit ? dev_div<T>(k.thresh - dist, dist) : (T)0;            // :331
            fx = dx * factor; fy = dy * factor; fz = dz * factor;
        }
        unsigned int hm = (unsigned int)((ballot64(hit) >> gsh) & GM);
        if (!__any(hm != 0u)) continue;
        T tx = (T)0, ty = (T)0, tz = (T)0;
        int nh = 0;
#ifndef CLOTHHIP_SERIAL_HITSUM
        if constexpr (sizeof(T) == 4 && GSZ == 16) {
            tx = row_allsum_f32(fx); ty = row_allsum_f32(fy); tz = row_allsum_f32(fz);      // (see collide_cell_wave)
            nh = __popc(hm);
        } else
#endif
        while (__any(hm != 0u)) {               // ascending candidate order; four hits are fetched per LDS round trip
            bool has[4]; T vx[4], vy[4], vz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                has[u] = hm != 0u;
                const int b = has[u] ? __ffs((int)hm) - 1 : 0;
                hm &= hm - 1u;
                vx[u] = lane_pull(fx, base + b); vy[u] = lane_pull(fy, base + b); vz[u] = lane_pull(fz, base + b);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (has[u]) { tx += vx[u]; ty += vy[u]; tz += vz[u]; nh++; }
        }
        if (nh != 0 && sub == a && act) {                                               // :336-343
            const T nf = (T)nh;
            x = xa + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps);
            y = ya + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps);
            z = za + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps);
            moved = true;
        }
        // my group's `a` moved: its later neighbours within the candidate radius must look too
        const bool wake = act && nh != 0 && free_ && sub > a && !(d2 > thr2c);
        todo |= (unsigned int)((ballot64(wake) >> gsh) & GM);
    }
    if (moved) cur[i] = Pt<T>{x, y, z, me.w};
}

// Same, by a single lane (cells with more than 64 members; not expected in practice).
template <typename T>
__device__ __forceinline__ void collide_cell_serial(Pt<T> *cur, uint16_t *m, int n, const DevConsts<T> &k) {
    for (int a = 1; a < n; a++) {                           // restore ascending point index
        const uint16_t v = m[a];
        int b = a - 1;
        while (b >= 0 && m[b] > v) { m[b + 1] = m[b]; b--; }
        m[b + 1] = v;
    }
    const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
    for (int a = 0; a < n; a++) {
        const int i = (int)m[a];
        const Pt<T> I = cur[i];
        if (w_cnt(I.w)) continue;
        T tx = (T)0, ty = (T)0, tz = (T)0;
        int nh = 0;
        for (int b = 0; b < n; b++) {
            if (b == a) continue;
            const Pt<T> J = cur[(int)m[b]];
            const T dx = I.x - J.x, dy = I.y - J.y, dz = I.z - J.z;
            const T d2 = sumsq<T>(dx, dy, dz);
            if (d2 > thr2) continue;
            const T dist = dev_sqrt<T>(d2);
            if (dist <= k.thresh) {
                const T factor = dev_div<T>(k.thresh - dist, dist);
                tx += dx * factor; ty += dy * factor; tz += dz * factor;
                nh += 1;
            }
        }
        if (nh != 0) {
            const T nf = (T)nh;
            cur[i] = Pt<T>{I.x + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps), I.y + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps),
                           I.z + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps), I.w};
        }
    }
}

}  // namespace clothhip
