// cloth_tables.hpp -- host-side static tables of the cloth stepper (depend only on n_side).
//
// The reference builds a list of Spring objects in row-major owner order (cloth.pyx:134-146) and walks
// that list twice per substep: _hookes (cloth.pyx:221-237, a sum whose rounding follows list order) and
// _limit_spring_changes (cloth.pyx:258-296, an in-place Gauss-Seidel sweep).  The device never sees the
// list; it sees two tables derived from it here:
//
//   * gather table: for every point the <=12 incident springs in ascending list index, so a per-point
//     gather adds the Hooke forces in exactly the order the reference's scatter does;
//   * window table: springs grouped into dependency levels (level = 1 + max(level of the previous
//     spring touching either endpoint)); springs inside a level share no endpoint, and executing the
//     levels in order reproduces the sequential sweep exactly.  Consecutive
//     levels are packed into WINDOWS of 64 slots -- one slot per lane of the wave that walks the sweep --
//     so that lane order == level order inside a window; per slot the set of earlier lanes of its window the spring
//     depends on (shares a particle with, transitively) lets one pass of the sweep finish several levels at once.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace clothhip {

enum : uint8_t { SPRING_STRUCTURAL = 0, SPRING_SHEARING = 1, SPRING_BENDING = 2 };

// gather-table entry layout (uint32)
constexpr uint32_t HK_NBR_MASK = 0xFFFu;        // bits 0..11  neighbour point index (P <= 4096)
constexpr int HK_POS_SHIFT = 12;                // bits 12..27 slot of the spring in the window table
constexpr uint32_t HK_POS_MASK = 0xFFFFu;
constexpr uint32_t HK_ASB = 1u << 28;           // this point is the spring's ptB (owner): f += -(F)
constexpr uint32_t HK_BEND = 1u << 29;          // BENDING spring: ks * 0.2
constexpr uint32_t HK_VALID = 1u << 31;
constexpr int HK_SLOTS = 12;
constexpr int MAX_SIDE = 64;                    // 12-bit point index

struct Topology {
    int N = 0, P = 0, S = 0;
    std::vector<int32_t> a, b;       // ptA (earlier point), ptB (owning point)   cloth.pyx:136-146
    std::vector<uint8_t> type;
};

inline int spring_count(int N) { return 2 * N * (N - 1) + 2 * (N - 1) * (N - 1) + 2 * N * (N - 2); }

// cloth.pyx:134-146: six springs per owning point (r,c), in this order.
inline Topology build_topology(int N) {
    Topology t;
    t.N = N; t.P = N * N; t.S = spring_count(N);
    t.a.reserve(t.S); t.b.reserve(t.S); t.type.reserve(t.S);
    auto add = [&](int ai, int bi, uint8_t ty) { t.a.push_back(ai); t.b.push_back(bi); t.type.push_back(ty); };
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++) {
            int i = r * N + c;
            if (r > 0) add((r - 1) * N + c, i, SPRING_STRUCTURAL);
            if (c > 0) add(r * N + c - 1, i, SPRING_STRUCTURAL);
            if (r > 0 && c > 0) add((r - 1) * N + c - 1, i, SPRING_SHEARING);
            if (r > 0 && c + 1 < N) add((r - 1) * N + c + 1, i, SPRING_SHEARING);
            if (r > 1) add((r - 2) * N + c, i, SPRING_BENDING);
            if (c > 1) add(r * N + c - 2, i, SPRING_BENDING);
        }
    return t;
}

struct LevelSchedule {
    int n_levels = 0, max_width = 0;
    std::vector<int32_t> off;        // [n_levels+1] offsets into the level-ordered spring arrays
    std::vector<int32_t> order;      // level-ordered position -> spring list index
    std::vector<int32_t> pos_of;     // spring list index -> level-ordered position
    std::vector<uint32_t> ent;       // level-ordered: ptA | ptB << 16
};

inline LevelSchedule build_levels(const Topology &t) {
    LevelSchedule L;
    std::vector<int> last(t.P, 0), lvl(t.S, 0);
    int nl = 0;
    for (int s = 0; s < t.S; s++) {
        int l = 1 + std::max(last[t.a[s]], last[t.b[s]]);
        lvl[s] = l; last[t.a[s]] = l; last[t.b[s]] = l;
        nl = std::max(nl, l);
    }
    L.n_levels = nl;
    L.off.assign(nl + 1, 0);
    for (int s = 0; s < t.S; s++) L.off[lvl[s]]++;          // count per level (1-based) ...
    for (int l = 1; l <= nl; l++) { L.max_width = std::max(L.max_width, L.off[l]); }
    {   // ... exclusive prefix: off[l-1] = start of level l
        int run = 0;
        for (int l = 1; l <= nl; l++) { int cnt = L.off[l]; L.off[l - 1] = run; run += cnt; }
        L.off[nl] = run;
    }
    L.order.assign(t.S, 0); L.pos_of.assign(t.S, 0); L.ent.assign(t.S, 0);
    std::vector<int> fill(L.off.begin(), L.off.end() - 1);
    for (int s = 0; s < t.S; s++) {                           // ascending list index inside a level
        int p = fill[lvl[s] - 1]++;
        L.order[p] = s; L.pos_of[s] = p;
        L.ent[p] = (uint32_t)t.a[s] | ((uint32_t)t.b[s] << 16);
    }
    return L;
}

// Window table of the strain-limit sweep. Slot i = window (i >> 6), lane (i & 63). Entry (0 = empty slot, ptA == ptB == 0,
// which can never stretch):
//   bits  0..11  ptA           bits 12..23  ptB
//   bits 24..27  which of the window's levels (in order, saturating at 15) the spring belongs to (informative)
//   bits 28..31  reach: how many windows past its own the last spring incident to a particle of THIS WINDOW's springs sits, in
//                units of 2^reach_shift windows, rounded up (the same value in every entry of a window: a pass may correct
//                springs of several levels at once); no correction made in this window can influence anything behind that window
// Beside it, per slot, `dep`: the lanes of the same window whose springs this one depends on -- every earlier lane that shares a
// particle with it, transitively closed.
// The rest-length arrays of the device (one per env for tier 2, else one shared) are kept in slot order too.
constexpr int WT_IDX_BITS = 12, WT_GROUP_SHIFT = 24, WT_REACH_SHIFT = 28;
constexpr uint32_t WT_IDX_MASK = 0xFFFu;
constexpr int WT_MAX_GROUPS = 16, WT_MAX_REACH = 15;
constexpr int WT_PAD_WINDOWS = 4;                // empty windows behind the last one: the entry stream reads ahead

struct WindowTable {
    int nW = 0, n_slots = 0;         // windows; slots incl. the padding windows
    int max_reach = 0, reach_shift = 0;
    std::vector<uint32_t> ent;       // [n_slots]
    std::vector<uint64_t> dep;       // [n_slots] lanes of the SAME window (all earlier ones) the slot's spring transitively depends on
    std::vector<int32_t> slot_of;    // spring list index -> slot
    std::vector<int32_t> spring_at;  // slot -> spring list index, -1 empty
};

inline WindowTable build_windows(const Topology &t, const LevelSchedule &L) {
    WindowTable W;
    W.slot_of.assign(t.S, -1);
    std::vector<int> group_of(t.S, 0);
    int w = 0, used = 0, groups = 0;
    for (int l = 0; l < L.n_levels; l++) {
        int p = L.off[l];
        while (p < L.off[l + 1]) {                                // a level may continue in the next window: still an antichain
            if (used == 64) { w++; used = 0; groups = 0; }
            const int take = std::min(L.off[l + 1] - p, 64 - used);
            for (int q = 0; q < take; q++) {
                const int s = L.order[p + q];
                W.slot_of[s] = w * 64 + used + q;
                group_of[s] = groups;
            }
            p += take; used += take; groups++;
        }
    }
    W.nW = w + 1;
#ifndef CLOTHHIP_WINDOW_LEVEL_ORDER
    // Lane order inside a window (round 5): any linear extension of "an earlier spring that shares a particle comes first" serves the sweep
    // (its pass rule reads the dependency masks, not the lane numbers). Level order is one; this one is chosen for the LDS banks: the sweep
    // reads the two particle records of every lane (16-byte records: two particles whose indices agree mod 16 share their banks), sixteen lanes
    // to a bank cycle, so within each group of sixteen lanes the ptA indices -- and the ptB indices -- should differ mod 16 (same particle =
    // same address = a broadcast, no conflict). Greedy list scheduling over the window's springs: 25x25 per window, summed over its four
    // groups of sixteen, max records per bank group 12.7 (ptA) / 9.9 (ptB) in level order -> 7.7 / 4.5 (4 = conflict-free).
    {
        std::vector<std::vector<int>> members((size_t)W.nW);
        for (int s = 0; s < t.S; s++) members[W.slot_of[s] >> 6].push_back(s);          // (ascending list index)
        std::vector<int> last_in_win(t.P, -1), npred(t.S, 0);
        std::vector<std::vector<int>> succ(t.S);
        for (int ws = 0; ws < W.nW; ws++) {
            const std::vector<int> &m = members[ws];
            for (int s : m) {                                                            // in-window predecessors: the previous spring of either particle
                const int pa = last_in_win[t.a[s]], pb = last_in_win[t.b[s]];
                if (pa >= 0) { succ[pa].push_back(s); npred[s]++; }
                if (pb >= 0 && pb != pa) { succ[pb].push_back(s); npred[s]++; }
                last_in_win[t.a[s]] = s; last_in_win[t.b[s]] = s;
            }
            for (int s : m) { last_in_win[t.a[s]] = -1; last_in_win[t.b[s]] = -1; }
            std::vector<int> ready, placed;
            for (int s : m) if (npred[s] == 0) ready.push_back(s);
            while (!ready.empty()) {
                const int g0 = (int)placed.size() / 16 * 16;                             // the sixteen-lane group being filled
                int best = -1, best_key = 0;
                for (int s : ready) {
                    int ca = 0, cb = 0;                                                  // other particles of the group in the same bank group
                    for (size_t q = g0; q < placed.size(); q++) {
                        const int o = placed[q];
                        if (t.a[o] != t.a[s] && (t.a[o] & 15) == (t.a[s] & 15)) ca++;
                        if (t.b[o] != t.b[s] && (t.b[o] & 15) == (t.b[s] & 15)) cb++;
                    }
                    const int key = (std::max(ca, cb) << 20) | ((ca + cb) << 12) | 0;     // then the earliest spring (ready is kept in list order)
                    if (best < 0 || key < best_key) { best = s; best_key = key; }
                }
                placed.push_back(best);
                ready.erase(std::find(ready.begin(), ready.end(), best));
                for (int n : succ[best]) if (--npred[n] == 0) ready.insert(std::upper_bound(ready.begin(), ready.end(), n), n);
            }
            for (size_t q = 0; q < placed.size(); q++) W.slot_of[placed[q]] = ws * 64 + (int)q;
        }
    }
#endif

    std::vector<int> last_win(t.P, 0);                            // window of the last spring incident to a point
    for (int s = 0; s < t.S; s++) {
        const int ws = W.slot_of[s] >> 6;
        last_win[t.a[s]] = std::max(last_win[t.a[s]], ws);
        last_win[t.b[s]] = std::max(last_win[t.b[s]], ws);
    }
    std::vector<int> wreach((size_t)W.nW, 0);                     // per window: the farthest reach of its springs (a pass may commit
    for (int s = 0; s < t.S; s++) {                               // springs of several groups at once)
        const int ws = W.slot_of[s] >> 6;
        const int r = std::max(last_win[t.a[s]], last_win[t.b[s]]) - ws;
        wreach[ws] = std::max(wreach[ws], r);
        W.max_reach = std::max(W.max_reach, r);
    }
    while (((W.max_reach + (1 << W.reach_shift) - 1) >> W.reach_shift) > WT_MAX_REACH) W.reach_shift++;
    // padding: the entry stream reads WT_PAD_WINDOWS ahead, and a reach rounded up to its unit may point up to 2^reach_shift - 1
    // windows past the last one (they are empty: walking them changes nothing)
    W.n_slots = (W.nW + WT_PAD_WINDOWS + (1 << W.reach_shift) - 1) * 64;
    W.ent.assign(W.n_slots, 0u);
    W.spring_at.assign(W.n_slots, -1);
    for (int s = 0; s < t.S; s++) {
        const int i = W.slot_of[s], ws = i >> 6;
        const int r = (wreach[ws] + (1 << W.reach_shift) - 1) >> W.reach_shift;
        W.ent[i] = (uint32_t)t.a[s] | ((uint32_t)t.b[s] << WT_IDX_BITS) | ((uint32_t)std::min(group_of[s], WT_MAX_GROUPS - 1) << WT_GROUP_SHIFT) |
                   ((uint32_t)r << WT_REACH_SHIFT);
        W.spring_at[i] = s;
    }
    // Dependencies inside a window: lane l depends on every earlier lane that shares a particle with it, transitively. A pass of
    // the sweep commits every over-stretched spring none of whose (transitive) predecessors is over-stretched itself: those
    // springs see exactly the state the sequential sweep shows them; everything behind an over-stretched spring is evaluated
    // again by the next pass.
    W.dep.assign(W.n_slots, 0ull);
    {
        std::vector<uint64_t> touch(t.P, 0ull);
        std::vector<int> touched;
        for (int ws = 0; ws < W.nW; ws++) {
            touched.clear();
            for (int l = 0; l < 64; l++) {
                const int s = W.spring_at[ws * 64 + l];
                if (s < 0) continue;
                const uint64_t direct = touch[t.a[s]] | touch[t.b[s]];
                uint64_t all = direct;
                for (uint64_t m = direct; m; m &= m - 1) all |= W.dep[ws * 64 + __builtin_ctzll(m)];
                W.dep[ws * 64 + l] = all;
                touch[t.a[s]] |= 1ull << l; touch[t.b[s]] |= 1ull << l;
                touched.push_back(t.a[s]); touched.push_back(t.b[s]);
            }
            for (int p : touched) touch[p] = 0ull;
        }
    }
    return W;
}

// gather table [HK_SLOTS][Ppad]: slot k of point i = its k-th incident spring in ascending list index.
inline std::vector<uint32_t> build_gather(const Topology &t, const WindowTable &W, int Ppad) {
    std::vector<uint32_t> tab((size_t)HK_SLOTS * Ppad, 0u);
    std::vector<int> n(t.P, 0);
    for (int s = 0; s < t.S; s++) {
        uint32_t common = ((uint32_t)W.slot_of[s] << HK_POS_SHIFT) | HK_VALID |
                          (t.type[s] == SPRING_BENDING ? HK_BEND : 0u);
        int a = t.a[s], b = t.b[s];
        tab[(size_t)n[a]++ * Ppad + a] = common | (uint32_t)b;            // point is ptA: f += F
        tab[(size_t)n[b]++ * Ppad + b] = common | (uint32_t)a | HK_ASB;   // point is ptB: f += -F
    }
    return tab;
}

// Gripper.grab_top level table (gripper.pyx:31-41): curZ = height; while curZ > 0: ...; curZ -= thickness
inline std::vector<double> build_grab_levels(double height, double thickness) {
    std::vector<double> lv;
    if (!(thickness > 0)) return lv;
    double z = height;
    while (z > 0 && lv.size() < 100000) { lv.push_back(z); z -= thickness; }
    return lv;
}

}  // namespace clothhip
