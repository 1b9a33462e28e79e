// phase_strain.hpp -- strain limit + tear (cloth.pyx:258-296): the strictly ordered sweep over the window table, three ways of walking it
// (strain_sweep: one wave, instrumented builds and tear_thresh < 1.1; strain_sweep_lean: one wave, the production walk; strain_sweep_mw: all
// waves of the cloth, A/B only). DESIGN.md 4.1 has the exactness argument, tests/test_sweep_rule.py pins the pass rule on the CPU.
#pragma once

#include "cloth_common.hpp"

namespace clothhip {

// Strain limit + tear (cloth.pyx:258-296) by ONE wave, exactly in the reference's order.
//
// The springs sit in the window table (cloth_tables.hpp): window = 64 slots = one spring per lane, consecutive dependency
// levels in lane order. A PASS evaluates every not yet finished spring of the window against the same particle state. A spring
// is VALID in that pass when none of the earlier springs of the window it depends on -- shares a particle with, transitively: a
// static 64-bit lane mask per table slot -- is over-stretched now: every predecessor that touches one of its particles then
// leaves it alone, so the spring sees exactly what the sequential sweep shows it. All valid springs are finished by the pass, the
// over-stretched ones corrected at once (two valid over-stretched springs share no particle, or the later one would not be
// valid); the others are evaluated again by the next pass. The first over-stretched spring in table order is always valid, so
// every pass with work makes progress; a window without a correction costs one pass. tests/test_sweep_rule.py pins this rule,
// on the tables the library exports (clothhip_selftest_windows), to the reference's sequential loop bit for bit (CPU).
// The walk starts at the window of the first spring the pre-pass flagged (nothing before it is over-stretched and nothing has
// moved yet) and ends behind the last window that can hold work: the last flagged spring, pushed out by every correction to the
// last window that holds a spring of one of the two moved particles (the entry's static `reach`). Everything outside
// [w0, w_end] provably evaluates to "no correction, no tear".
// Entry stream: lane-private, coalesced, read PF windows ahead (LDS or, for the large grids, L2).
template <typename T, bool LDS_TAB, bool TIMED, bool STATS, bool TIC>
__device__ __forceinline__ int strain_sweep(Pt<T> *cur, const WEnt<T> *wt, const uint32_t *g_ent, const T *g_rest,
                                            const unsigned long long *g_dep, int w0, int w_end,
                                            int nW, int rshift, const DevConsts<T> &k, int lane, int &st_windows, int &st_passes, int &st_commits,
                                            TphT tph, unsigned long long fmask = 0ull) {
    constexpr int PF = LDS_TAB ? 1 : 3;          // entry stream: windows read ahead
    // dependency words (always from L2 / L1: one table for all cloths). The queue's rotation needs the NEWEST word, so whatever its
    // depth the stream runs one window ahead: fp32 keeps two words (three and four measured the same, with more moves per window)
    constexpr int PD = sizeof(T) == 4 ? 1 : 2;
    static_assert(PF + 1 <= WT_PAD_WINDOWS && PD + 1 <= WT_PAD_WINDOWS, "the table is padded by the read-ahead distance");
    int tear = 0;
#ifdef CLOTHHIP_SWEEP_OUTER
    unsigned long long so0_, so1_, so2_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(so0_)::"memory");
#endif
    DevConsts<T> kl = k;                         // spring-test constants pinned in VGPRs
    asm volatile("" : "+v"(kl.c11), "+v"(kl.tear_thresh));
    constexpr bool tic = TIC;                    // tear_thresh >= 1.1: tear implies stretch (the usual case; the caller tests it once)
    const T INF_ = sizeof(T) == 4 ? (T)__builtin_huge_valf() : (T)__builtin_huge_val();
    uint32_t eab[PF + 1]; T erest[PF + 1];
    unsigned long long edep[PD + 1];
    auto load = [&](int wi, uint32_t &ab_, T &r_) {
        if (LDS_TAB) { const WEnt<T> e_ = wt[wi * 64 + lane]; ab_ = e_.ab; r_ = e_.rest; }
        else { const uint32_t ix = (uint32_t)(wi * 64 + lane); ab_ = g_ent[ix]; r_ = g_rest[ix]; }    // (unsigned: scalar base + 32-bit offset addressing)
    };
#pragma unroll
    for (int j = 0; j <= PD; j++) edep[j] = g_dep[(uint32_t)((w0 + j) * 64 + lane)];
#pragma unroll
    for (int j = 0; j <= PF; j++) load(w0 + j, eab[j], erest[j]);
    // Both loops are single-exit do-whiles with wave-uniform conditions (ballots), so they compile to plain scalar branches; the
    // particle state carried from pass to pass is the six coordinates only (12-byte LDS reads / writes: the pin word never changes
    // during a sweep and is read once per window).
    struct __attribute__((aligned(16))) P3 { T x, y, z; };
    int w = w0;
    if (w > w_end) return tear;
#ifdef CLOTHHIP_CELL_COUNTERS
    int corr_end_ = w0 - 1;                      // census: the last window a correction made so far can reach
#endif
    // fp32: the particle records of the NEXT window are read while this window's passes run; they are good unless this window
    // corrected something (then they are read again): most windows of a walk correct nothing. (fp64: the sixteen registers
    // this costs are spilled, measured -2 %; there the records are read when the window starts.)
    constexpr bool NEXT_AHEAD = sizeof(T) == 4;
    Pt<T> NA, NB;
    int an = (int)(eab[0] & WT_IDX_MASK), bn = (int)__builtin_amdgcn_ubfe(eab[0], WT_IDX_BITS, WT_IDX_BITS);
    NA = cur[an]; NB = cur[bn];
#ifdef CLOTHHIP_SWEEP_OUTER
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(so1_)::"memory");
#endif
    do {
        const uint32_t ab = eab[0];
        const T rest = erest[0];
        const unsigned long long dep = edep[0];
#pragma unroll
        for (int j = 0; j < PF; j++) { eab[j] = eab[j + 1]; erest[j] = erest[j + 1]; }
#pragma unroll
        for (int j = 0; j < PD; j++) edep[j] = edep[j + 1];
        load(w + PF + 1, eab[PF], erest[PF]);
        edep[PD] = g_dep[(uint32_t)((w + PD + 1) * 64 + lane)];
        const int a = an, b = bn;                 // (decoded once, as the next window's, by the window before: +1.5 %)
        P3 *const pa = reinterpret_cast<P3 *>(cur + a), *const pb = reinterpret_cast<P3 *>(cur + b);
        T ax, ay, az, bx, by, bz;
        uint32_t ca, cb;
        ax = NA.x; ay = NA.y; az = NA.z; bx = NB.x; by = NB.y; bz = NB.z;
        ca = w_cnt(NA.w); cb = w_cnt(NB.w);                         // pins do not change during a sweep
        an = (int)(eab[0] & WT_IDX_MASK); bn = (int)__builtin_amdgcn_ubfe(eab[0], WT_IDX_BITS, WT_IDX_BITS);
        if (NEXT_AHEAD) { NA = cur[an]; NB = cur[bn]; }
#ifdef CLOTHHIP_WINDOW_STAMPS          // dev measurement (sweep-stamps build): how long the read-ahead's two 16-byte reads take when waited for
        if (TIMED) {                   // at once ([1], count [2]) against two stamps back to back ([3]): the LDS latency the sweep sees
            unsigned long long s0_, s1_, s2_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0_)::"memory");
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1_)::"memory");
            const Pt<T> xa_ = cur[an], xb_ = cur[bn];
            asm volatile("s_memtime %0" : "=s"(s2_)::"memory");
            T keep_ = xa_.x + xb_.x; asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(keep_)::"memory");
            unsigned long long s3_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s3_)::"memory");
            tph[3] += s1_ - s0_; tph[1] += s3_ - s1_; tph[2] += 64;
        }
#endif
        bool dirty = !NEXT_AHEAD;
        const T t11 = rest * kl.c11;
        // both ends pinned: skipped by the reference (:268) -- by a limit no length exceeds: ONE compare per pass then
        const T tlim = ((ca != 0) & (cb != 0)) ? INF_ : t11;
        const uint32_t dlo = (uint32_t)dep, dhi = (uint32_t)(dep >> 32);
        bool pl = true;                                             // this lane's spring is not finished
        // a finished spring's limit becomes +inf, so that the compare alone yields the wave's mask of over-stretched UNFINISHED
        // springs (the ballot of a conjunction costs a select and a compare more per pass); the unfinished lanes as a scalar mask
        constexpr bool V1 = true;
        T tl = tlim;
        T tl2 = tlim * tlim * ((T)1 - filt_slack<T>());           // fp64: the squared pre-filter of the limit
        unsigned long long plm = ~0ull;
        if (STATS) st_windows++;
#ifdef CLOTHHIP_CELL_COUNTERS
        if (w > corr_end_ && !((fmask >> (w & 63)) & 1ull)) tph[3] += 64;   // census: no flagged spring, beyond every correction's reach
#endif
        bool more;
        do {
            unsigned long long td0 = 0;
            if (TIMED) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td0)::"memory"); }
            const T dx = ax - bx, dy = ay - by, dz = az - bz;
            const T len2 = sumsq<T>(dx, dy, dz);
            bool trig;
            T len;
            if constexpr (sizeof(T) == 4) {      // one v_sqrt: cheaper than a branch around it
                len = dev_sqrt<T>(len2);                                        // :270
                trig = len > (V1 ? tl : tlim);                                  // :275
            } else {
                trig = false; len = (T)0;
                if (len2 > (V1 ? tl2 : tlim * tlim * ((T)1 - filt_slack<T>()))) {
                    len = dev_sqrt<T>(len2);
                    trig = len > (V1 ? tl : tlim);
                }
            }
            // the over-stretched unfinished springs of the window, as this state shows them
            const unsigned long long tb = V1 ? ballot64(trig) : ballot64(trig & pl);
            if (STATS) st_passes++;
            // A spring is VALID when none of its (transitive) predecessors in the window is over-stretched now: then every
            // predecessor that shares a particle with it leaves the particle alone, and the spring sees what the sequential sweep
            // shows it. All valid springs are finished by this pass (the over-stretched ones corrected, all at once: two valid
            // over-stretched springs share no particle, or the later one would not be valid); the others are evaluated again.
            // The first over-stretched spring in table order is always valid.
            bool bad = false;
            if (tb) bad = ((dlo & (uint32_t)tb) | (dhi & (uint32_t)(tb >> 32))) != 0u;      // (a quiet pass skips this)
            if (!tic) {          // tear_thresh < 1.1: a spring can tear without stretching, so every finished spring is tested (:272)
                const bool mine = pl & !bad;
                if (mine && !((ca != 0) & (cb != 0)) && dev_sqrt<T>(len2) > rest * kl.tear_thresh) tear = 1;
            }
            more = false;
            if (tb) {
                // every correction of the window may move particles whose springs sit as far as the window's reach
                const int reach = w + ((int)((uint32_t)__builtin_amdgcn_readfirstlane((int)ab) >> WT_REACH_SHIFT) << rshift);
                w_end = reach > w_end ? reach : w_end;
#ifdef CLOTHHIP_CELL_COUNTERS
                corr_end_ = reach > corr_end_ ? reach : corr_end_;
#endif
                dirty = true;
                if (STATS) st_commits++;
                if (V1 ? (trig & !bad) : (trig & pl & !bad)) {
                    if (tic && len > rest * kl.tear_thresh) tear = 1;               // :272
                    const T ux = dev_div<T>(dx, len), uy = dev_div<T>(dy, len), uz = dev_div<T>(dz, len);   // :276-278
                    const T extra = len - t11;                                      // :279
                    // A pinned: B += dir*extra ; B pinned: A -= dir*extra ; else A -= dir*(extra*0.5), B += dir*(extra*0.5)
                    // (extra * 1.0 == extra exactly, so one weighted form covers the three reference branches, :281-296)
                    const T wa = ca != 0 ? (T)0 : (cb != 0 ? (T)1 : (T)0.5);
                    const T wb = cb != 0 ? (T)0 : (ca != 0 ? (T)1 : (T)0.5);
                    const T ea = extra * wa, eb = extra * wb;
                    // branch-free: a pinned end has weight 0 and x - u*0 == x exactly (u is finite: len > 0 here), so writing it
                    // back unchanged equals the reference's skipped assignment; the springs corrected together share no
                    // particle, so nobody else writes these two records in this pass
                    *pa = P3{mad<T>(-ux, ea, ax), mad<T>(-uy, ea, ay), mad<T>(-uz, ea, az)};
                    *pb = P3{mad<T>(ux, eb, bx), mad<T>(uy, eb, by), mad<T>(uz, eb, bz)};
                }
                pl = pl & bad;
                if (V1) { tl = pl ? tl : INF_; if (sizeof(T) == 8) tl2 = pl ? tl2 : INF_; plm &= ballot64(bad); more = plm != 0ull; }
                else more = ballot64(pl) != 0ull;
                if (TIMED) { unsigned long long td1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td1)::"memory"); tph[11] += td1 - td0; }
                if (more) {
                    // same-wave LDS operations execute in program order: the reads below see the writes above without waiting
                    // for them; the barrier only pins the compiler's ordering
                    __builtin_amdgcn_wave_barrier();
                    const P3 na = *pa, nb = *pb;
                    ax = na.x; ay = na.y; az = na.z; bx = nb.x; by = nb.y; bz = nb.z;
                }
            } else if (TIMED) { unsigned long long td1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td1)::"memory"); tph[10] += td1 - td0; tph[9] += 64; }
        } while (more);
        w++;
        if (dirty) { NA = cur[an]; NB = cur[bn]; }
    } while (w <= w_end);
#ifdef CLOTHHIP_SWEEP_OUTER
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(so2_)::"memory");
    tph[10] += so1_ - so0_; tph[11] += so2_ - so1_;
#endif
    return tear;
}

// The one-wave sweep again, shaped for how a LONE wave issues (round 5; tools/micro/lone_wave_issue.hip, MI355X): 4.5 cycles per
// instruction whatever its kind, but ~40 cycles for a branch on a value a vector instruction has just produced (v_cmp -> vcc / SGPR
// -> s_cbranch) and ~22 for any taken branch -- strain_sweep's quiet window (~75 instructions) runs through five taken branches and
// two such dependencies, ~550 cycles of which its arithmetic is 60. Here a QUIET window is straight-line code with ONE conditional
// branch (not taken) and the loop's back-edge every second window:
//   * two register sets (the walk is unrolled by two, the sets swap roles by name: no queue moves): entry {ab, rest, dep}, decoded
//     addresses, the two particle records;
//   * per window: decode the NEXT window's entry and issue its two 16-byte particle reads (speculative: good unless this window
//     corrects something, then they are read again), evaluate THIS window against t11 = rest * 1.1 -- no look at the pins: a
//     both-pinned spring (skipped by the reference, :268) can only make the window take the exact path below for nothing --, branch,
//     prefetch the entry of the window after next into the set this window has just released;
//   * everything a correction needs -- pins and weights, the dependency word, the exact limit with both-pinned springs sorted out,
//     the reach, the pass loop of strain_sweep (same rule, same arithmetic) -- sits behind that one branch.
// Same walk, same passes, same results as strain_sweep (tear_thresh >= 1.1 only: the caller keeps strain_sweep for the other case).
// (AHEAD false -- the high-residency builds, whose other waves hide the latency: a window's particle records are read when the window starts,
//  not a window ahead: sixteen registers fewer across the walk)
template <typename T, bool LDS_TAB, bool STATS, bool AHEAD = true>
__device__ __forceinline__ int strain_sweep_lean(Pt<T> *cur, const WEnt<T> *wt, const uint32_t *g_ent, const T *g_rest,
                                                 const unsigned long long *g_dep, int w0, int w_end, int rshift, const DevConsts<T> &k,
                                                 int lane, int &st_windows, int &st_passes, int &st_commits) {
    static_assert(WT_PAD_WINDOWS >= 3, "the entry stream reads two windows ahead, the particle reads one");
    int tear = 0;
    T c11 = k.c11, tth = k.tear_thresh;              // spring-test constants pinned in VGPRs
    asm volatile("" : "+v"(c11), "+v"(tth));
    const T INF_ = sizeof(T) == 4 ? (T)__builtin_huge_valf() : (T)__builtin_huge_val();
    struct __attribute__((aligned(16))) P3 { T x, y, z; };
    struct Set { uint32_t ab; T rest; unsigned long long dep; int a, b; Pt<T> A, B; };
    auto load = [&](int wi, Set &s) {
        const uint32_t ix = (uint32_t)(wi * 64 + lane);
        if (LDS_TAB) { const WEnt<T> e_ = wt[ix]; s.ab = e_.ab; s.rest = e_.rest; }
        else { s.ab = g_ent[ix]; s.rest = g_rest[ix]; }
        s.dep = g_dep[ix];
    };
    auto decode_read = [&](Set &s) {
        s.a = (int)(s.ab & WT_IDX_MASK); s.b = (int)__builtin_amdgcn_ubfe(s.ab, WT_IDX_BITS, WT_IDX_BITS);
        s.A = cur[s.a]; s.B = cur[s.b];
    };
    int w = w0;
    // one window: `c` holds it (entry decoded, particle records read or in flight), `n` the next one's entry
    auto step = [&](Set &c, Set &n) {
        if (AHEAD) decode_read(n);                                  // speculative: valid unless this window corrects something
        else decode_read(c);
        T ax = c.A.x, ay = c.A.y, az = c.A.z, bx = c.B.x, by = c.B.y, bz = c.B.z;
        T dx = ax - bx, dy = ay - by, dz = az - bz;
        T len2 = sumsq<T>(dx, dy, dz);
        const T t11 = c.rest * c11;
        T len = (T)0; bool trig;
        if constexpr (sizeof(T) == 4) { len = dev_sqrt<T>(len2); trig = len > t11; }
        else trig = len2 > t11 * t11 * ((T)1 - filt_slack<T>());    // fp64: the squared pre-filter decides whether anybody looks closer
        if (STATS) { st_windows++; st_passes++; }
        if (ballot64(trig) != 0ull) {                               // (no "unlikely" hint: with the exact path out of line the pulled states ran 2.5 % slower)
            // ---- the exact path (strain_sweep's pass loop; its first pass is the evaluation above) ----
            // (measured and rejected, round 5: the commit computed by every lane with the stores of the lanes that must not write sent
            //  to a per-lane sink record -- no exec-mask detour, one branch per pass --: -4 %; the window's "nobody left" exit dropped
            //  in favour of the next pass's "nobody over-stretched": -2.5 %; the pass hand-ordered so that every scalar instruction that waits for
            //  a vector compare's mask has independent vector work in front of it -- the correction computed before the exec mask is formed, the
            //  re-reads issued before the "nobody left" decision --: -2.5 %; the next window's records re-read together with every pass's own
            //  re-read, so that a window ending in a quiet pass leaves them fresh: -5 % -- two more scattered 16-byte reads per pass cost ~180 cycles;
            //  the re-read under an exec mask of the lanes still pending: -4 %)
            const uint32_t ca = w_cnt(c.A.w), cb = w_cnt(c.B.w);    // pins do not change during a sweep
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 4
            T tl = t11;             // MUTANT 4 (tools/run_mutants.sh; never a product build): the both-pinned skip of cloth.pyx:268 dropped
#else
            T tl = ((ca != 0) & (cb != 0)) ? INF_ : t11;           // both ends pinned: skipped by the reference (:268)
#endif
            T tl2 = tl * tl * ((T)1 - filt_slack<T>());
            auto test = [&]() {
                if constexpr (sizeof(T) == 4) { trig = len > tl; }
                else { trig = false; if (len2 > tl2) { len = dev_sqrt<T>(len2); trig = len > tl; } }
            };
            test();
            unsigned long long tb = ballot64(trig);
            if (tb) {
                P3 *const pa = reinterpret_cast<P3 *>(cur + c.a), *const pb = reinterpret_cast<P3 *>(cur + c.b);
                const uint32_t dlo = (uint32_t)c.dep, dhi = (uint32_t)(c.dep >> 32);
                // every correction of the window may move particles whose springs sit as far as the window's reach
                const int reach = w + ((int)((uint32_t)__builtin_amdgcn_readfirstlane((int)c.ab) >> WT_REACH_SHIFT) << rshift);
                w_end = reach > w_end ? reach : w_end;
                bool pl = true;
                unsigned long long plm = ~0ull;
                for (;;) {
                    // a spring is VALID when none of its (transitive) predecessors in the window is over-stretched now (strain_sweep)
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 3
                    // MUTANT 3 (tools/run_mutants.sh; never a product build): the SECOND over-stretched spring of the pass is committed together
                    // with the first whether or not it depends on it -- cloth.pyx:265-296 shows a later spring the earlier one's correction
                    const unsigned long long tb2_ = tb & (tb - 1ull);
                    const bool bad = (((dlo & (uint32_t)tb) | (dhi & (uint32_t)(tb >> 32))) != 0u) && !(tb2_ != 0ull && lane == __ffsll((long long)tb2_) - 1);
#else
                    const bool bad = ((dlo & (uint32_t)tb) | (dhi & (uint32_t)(tb >> 32))) != 0u;
#endif
#ifdef CLOTHHIP_COUNT_SPRINGS        // dev (profiling builds): count the SPRINGS a pass corrects instead of the correcting passes
                    if (STATS) st_commits += __builtin_popcountll(ballot64(trig & !bad));
#else
                    if (STATS) st_commits++;
#endif
                    if (trig & !bad) {
                        if (len > c.rest * tth) tear = 1;                               // :272 (tear implies stretch here)
                        const T ux = dev_div<T>(dx, len), uy = dev_div<T>(dy, len), uz = dev_div<T>(dz, len);   // :276-278
                        const T extra = len - t11;                                      // :279
                        const T wa = ca != 0 ? (T)0 : (cb != 0 ? (T)1 : (T)0.5);        // :281-296 as weights (strain_sweep)
                        const T wb = cb != 0 ? (T)0 : (ca != 0 ? (T)1 : (T)0.5);
                        const T ea = extra * wa, eb = extra * wb;
                        *pa = P3{mad<T>(-ux, ea, ax), mad<T>(-uy, ea, ay), mad<T>(-uz, ea, az)};
                        *pb = P3{mad<T>(ux, eb, bx), mad<T>(uy, eb, by), mad<T>(uz, eb, bz)};
                    }
                    pl = pl & bad;
                    tl = pl ? tl : INF_; if (sizeof(T) == 8) tl2 = pl ? tl2 : INF_;
                    plm &= ballot64(bad);
                    if (__builtin_expect(plm == 0ull, 0)) break;    // ("another pass" as the loop's fall-through: pulled states -1 %, bench +0.5 %)
                    __builtin_amdgcn_wave_barrier();                // same-wave LDS operations execute in program order
                    const P3 na = *pa, nb = *pb;
                    ax = na.x; ay = na.y; az = na.z; bx = nb.x; by = nb.y; bz = nb.z;
                    dx = ax - bx; dy = ay - by; dz = az - bz;
                    len2 = sumsq<T>(dx, dy, dz);
                    if constexpr (sizeof(T) == 4) len = dev_sqrt<T>(len2);
                    test();
                    tb = ballot64(trig);
                    if (STATS) st_passes++;
                    if (__builtin_expect(tb == 0ull, 0)) break;     // a quiet pass ends the window
                }
                if (AHEAD) { n.A = cur[n.a]; n.B = cur[n.b]; }      // the speculative records are stale now
            }
        }
        load(w + 2, c);                                             // this set is free: the entry of the window after next
    };
    Set S0, S1;
    load(w, S0); load(w + 1, S1);
    if (AHEAD) decode_read(S0);
    for (;;) {
        step(S0, S1);
        if (++w > w_end) break;
        step(S1, S0);
        if (++w > w_end) break;
    }
    return tear;
}

// The same sweep by ALL NW waves of the cloth (round 5): speculative look-ahead over the next NW windows.
//
// Invariant at the head of a ROUND: every window before `wb` is finished and the particle state is the sequential sweep's state
// at that point. Wave j holds the one window w of [wb, wb + NW) with w == j (mod NW) and evaluates its FIRST pass against that
// state; whether the window holds an over-stretched spring goes to an LDS flag at position w - wb. After the workgroup barrier
// every wave knows f, the first flagged window of the round. The windows before it are QUIET at the very state the sequential
// sweep shows them (nothing before them in the round moved anything): they are finished -- no correction, and, a tear implying
// a stretch (TIC), no tear. Window wb + f is then run to completion by its wave, exactly as strain_sweep's pass loop does
// (its first pass is the one already evaluated: the state has not changed since), while the others wait at a second barrier;
// it also publishes the end of the walk its corrections pushed out. The windows behind f were evaluated against a state that
// f's corrections have since changed: their waves KEEP them (entry decoded, pins read) and evaluate them again in the next
// round, wb' = wb + f + 1; the waves whose windows were finished move on to w + NW, whose table entry they read a round ahead.
// A round without a flagged window finishes NW windows for one barrier. The pass rule inside a window, the reach rule and the
// arithmetic are strain_sweep's; only WHO evaluates a window's first pass, and when, differs -- never against which state a
// finished window was evaluated. tests/test_sweep_rule.py models the rounds on the CPU against the sequential loop.
// `sw`: LDS ints, [0, 2 NW) the round's flags (double-buffered: a round's writes cannot meet the previous round's readers),
// [2 NW] the end of the walk as the correcting wave left it.
template <typename T, bool LDS_TAB, int NW, bool STATS, bool TIC>
__device__ __forceinline__ int strain_sweep_mw(Pt<T> *cur, const WEnt<T> *wt, const uint32_t *g_ent, const T *g_rest,
                                               const unsigned long long *g_dep, int w0, int w_end, int w_last, int rshift,
                                               const DevConsts<T> &k, int lane, int wave, int *sw, int *st) {
    static_assert((NW & (NW - 1)) == 0 && NW >= 2 && NW <= 16, "waves per cloth: a power of two");
    int tear = 0;
    DevConsts<T> kl = k;                         // spring-test constants pinned in VGPRs
    asm volatile("" : "+v"(kl.c11), "+v"(kl.tear_thresh));
    const T INF_ = sizeof(T) == 4 ? (T)__builtin_huge_valf() : (T)__builtin_huge_val();
    struct __attribute__((aligned(16))) P3 { T x, y, z; };
    // (windows past the table's padding are never active: a clamped read gives them an empty window's entries)
    auto load = [&](int wi, uint32_t &ab_, T &r_, unsigned long long &d_) {
        const uint32_t ix = (uint32_t)((wi < w_last ? wi : w_last) * 64 + lane);
        if (LDS_TAB) { const WEnt<T> e_ = wt[ix]; ab_ = e_.ab; r_ = e_.rest; }
        else { ab_ = g_ent[ix]; r_ = g_rest[ix]; }
        d_ = g_dep[ix];
    };
    // LDS traffic of this wave visible to the others, then the workgroup barrier; the table stream's global loads stay in flight
    auto wg_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    int wb = w0;
    int w = wb + ((wave - wb) & (NW - 1));
    uint32_t ab, abn; T rest, restn; unsigned long long dep, depn;
    load(w, ab, rest, dep);
    load(w + NW, abn, restn, depn);
    int bank = 0;
    while (wb <= w_end) {
        const int a = (int)(ab & WT_IDX_MASK), b = (int)__builtin_amdgcn_ubfe(ab, WT_IDX_BITS, WT_IDX_BITS);
        const Pt<T> RA = cur[a], RB = cur[b];
        P3 *const pa = reinterpret_cast<P3 *>(cur + a), *const pb = reinterpret_cast<P3 *>(cur + b);
        T ax = RA.x, ay = RA.y, az = RA.z, bx = RB.x, by = RB.y, bz = RB.z;
        const uint32_t ca = w_cnt(RA.w), cb = w_cnt(RB.w);          // pins do not change during a sweep
        const bool both = (ca != 0) & (cb != 0);
        const T t11 = rest * kl.c11;
        // both ends pinned: skipped by the reference (:268) -- by a limit no length exceeds: ONE compare per pass then
        T tl = both ? INF_ : t11;
        T tl2 = tl * tl * ((T)1 - filt_slack<T>());                 // fp64: the squared pre-filter of the limit
        T dx = ax - bx, dy = ay - by, dz = az - bz;
        T len2 = sumsq<T>(dx, dy, dz);
        T len; bool trig;
        auto test = [&]() {
            if constexpr (sizeof(T) == 4) {      // one v_sqrt: cheaper than a branch around it
                len = dev_sqrt<T>(len2);                                        // :270
                trig = len > tl;                                                // :275
            } else {
                trig = false; len = (T)0;
                if (len2 > tl2) { len = dev_sqrt<T>(len2); trig = len > tl; }
            }
        };
        test();
        unsigned long long tb = ballot64(trig);
        const int rel = w - wb;                                     // wave-uniform, in [0, NW)
        const bool hot = (w <= w_end) & (tb != 0ull);
        if (lane == 0) sw[bank + rel] = hot ? 1 : 0;
        wg_barrier();
        const int fv = sw[bank + (lane & (NW - 1))];
        const uint32_t fm = (uint32_t)ballot64(fv != 0) & ((1u << NW) - 1u);
        bank ^= NW;
        // tear_thresh < 1.1: a spring can tear without stretching, so every finished spring is tested (:272)
        auto tear_test = [&](bool mine) { if (mine && !both && dev_sqrt<T>(len2) > rest * kl.tear_thresh) tear = 1; };
        int adv = NW;                                               // windows this round finishes
        if (fm != 0u) {
            const int f = __builtin_ctz(fm);
            adv = f + 1;
            if (!TIC && rel < f && w <= w_end) tear_test(true);
            if (rel == f) {
                const uint32_t dlo = (uint32_t)dep, dhi = (uint32_t)(dep >> 32);
                bool pl = true;                                     // this lane's spring is not finished
                unsigned long long plm = ~0ull;
                // every correction of the window may move particles whose springs sit as far as the window's reach
                const int reach = w + ((int)((uint32_t)__builtin_amdgcn_readfirstlane((int)ab) >> WT_REACH_SHIFT) << rshift);
                w_end = reach > w_end ? reach : w_end;
                if (lane == 0) sw[2 * NW] = w_end;
                for (;;) {
                    // A spring is VALID when none of its (transitive) predecessors in the window is over-stretched now (see
                    // strain_sweep): all valid springs are finished by this pass, the over-stretched ones corrected at once
                    const bool bad = ((dlo & (uint32_t)tb) | (dhi & (uint32_t)(tb >> 32))) != 0u;
                    if (!TIC) tear_test(pl & !bad);
                    if (STATS && lane == 0) { atomicAdd(&st[0], 1); atomicAdd(&st[1], 1); }
                    if (trig & !bad) {
                        if (TIC && len > rest * kl.tear_thresh) tear = 1;               // :272
                        const T ux = dev_div<T>(dx, len), uy = dev_div<T>(dy, len), uz = dev_div<T>(dz, len);   // :276-278
                        const T extra = len - t11;                                      // :279
                        // A pinned: B += dir*extra ; B pinned: A -= dir*extra ; else A -= dir*(extra*0.5), B += dir*(extra*0.5)
                        // (extra * 1.0 == extra exactly, so one weighted form covers the three reference branches, :281-296)
                        const T wa = ca != 0 ? (T)0 : (cb != 0 ? (T)1 : (T)0.5);
                        const T wb_ = cb != 0 ? (T)0 : (ca != 0 ? (T)1 : (T)0.5);
                        const T ea = extra * wa, eb = extra * wb_;
                        // branch-free: a pinned end has weight 0 and x - u*0 == x exactly (see strain_sweep)
                        *pa = P3{mad<T>(-ux, ea, ax), mad<T>(-uy, ea, ay), mad<T>(-uz, ea, az)};
                        *pb = P3{mad<T>(ux, eb, bx), mad<T>(uy, eb, by), mad<T>(uz, eb, bz)};
                    }
                    pl = pl & bad;
                    tl = pl ? tl : INF_; if (sizeof(T) == 8) tl2 = pl ? tl2 : INF_;
                    plm &= ballot64(bad);
                    if (plm == 0ull) break;
                    // same-wave LDS operations execute in program order: the reads below see the writes above
                    __builtin_amdgcn_wave_barrier();
                    const P3 na = *pa, nb = *pb;
                    ax = na.x; ay = na.y; az = na.z; bx = nb.x; by = nb.y; bz = nb.z;
                    dx = ax - bx; dy = ay - by; dz = az - bz;
                    len2 = sumsq<T>(dx, dy, dz);
                    test();
                    tb = ballot64(trig);
                    if (tb == 0ull) {                                // a quiet pass ends the window
                        if (!TIC) tear_test(pl);
                        if (STATS && lane == 0) atomicAdd(&st[0], 1);
                        break;
                    }
                }
            }
            wg_barrier();
            w_end = __builtin_amdgcn_readfirstlane(sw[2 * NW]);
        } else if (!TIC) {
            tear_test(w <= w_end);
        }
        if (STATS && lane == 0 && wave == 0) {                      // windows walked; first passes of the quiet ones among them
            const int nw_ = fm != 0u ? adv : (w_end - wb + 1 < NW ? w_end - wb + 1 : NW);
            atomicAdd(&st[2], nw_); atomicAdd(&st[0], fm != 0u ? nw_ - 1 : nw_);
#ifdef CLOTHHIP_MW_ROUNDS               // dev measurement: rounds instead of windows, correcting rounds instead of correcting passes
            atomicAdd(&st[2], 1 - nw_); atomicAdd(&st[3], fm != 0u ? 1 : 0);
#endif
        }
        wb += adv;
        if (rel < adv) {
            w += NW; ab = abn; rest = restn; dep = depn;
            load(w + NW, abn, restn, depn);
        }
    }
    return tear;
}

}  // namespace clothhip
