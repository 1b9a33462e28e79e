// episode_loop.hpp -- k_run_schedule: one workgroup steps ONE cloth through a whole schedule or through whole episodes (action decode,
// grab, the substep loop, metrics, terminal test, resets) with the particle state resident in LDS. The ordered phases of a substep live in
// phase_strain.hpp / phase_collide.hpp; the variant table (threads x particles x table mode x arithmetic) is at the top of this file.
#pragma once

#include "cloth_common.hpp"
#include "phase_strain.hpp"
#include "phase_collide.hpp"
#include "cloth_metrics.hpp"

namespace clothhip {

// Particle i is owned by thread (i % NT); a thread owns PPT particles i = tid + q*NT. The previous position
// of a particle is only ever touched by its owner (adjust, Verlet, plane), so it lives in the owner's
// registers for the whole schedule, as do the particle's static gather entries (and, with REST_REG, the rest
// lengths of its incident springs). Only the current positions are shared, through LDS.
//   TAB: 0 static tables in global memory, 1 ent/rest/offsets in LDS, 2 also the per-point level table.
//   FUSED: 0 = one externally decoded schedule per env (clothhip_run); 1 = whole episodes per launch (clothhip_run_actions)
//          with the resets of the flat tiers 1 and 3; 2 = also tier-2 resets. (The tier-2 reset code is cold, but its presence
//          costs the substep loop registers: -7 % on the headline workload, so it is compiled in only where it is asked for.)
// LEAN variant (TAB <= 0 with REST_REG, fp32): the 12-slot gather stencil of a particle is recomputed from its grid position
// instead of being held in 36 registers, and rest lengths come from a three-value palette instead of 36 more: the stepper is then
// compiled for 168 VGPRs (TAB 0: three cloths share a CU) or 128 (TAB -1: four). Position k of the stencil = the k-th incident spring in ascending list index when
// all twelve exist (cloth.pyx:134-146: the six springs the point owns, then those its later neighbours own):
//   k      0    1    2      3      4     5    6   7    8      9    10     11
//   nbr   -N   -1   -N-1   -N+1   -2N   -2   +1  +2   +N-1   +N   +N+1   +2N      (index i = r*N + c)
//   type   S    S    Sh     Sh     B     B    S   B    Sh     S    Sh     B
// (the host checks this against the gather table it builds from the reference's spring list before choosing the variant).
__device__ __forceinline__ int lean_off(int k, int N) {
    switch (k) {
        case 0: return -N; case 1: return -1; case 2: return -N - 1; case 3: return -N + 1; case 4: return -2 * N; case 5: return -2;
        case 6: return 1; case 7: return 2; case 8: return N - 1; case 9: return N; case 10: return N + 1; default: return 2 * N;
    }
}
__host__ __device__ constexpr bool lean_bend(int k) { return k == 4 || k == 5 || k == 7 || k == 11; }
__host__ __device__ constexpr bool lean_shear(int k) { return k == 2 || k == 3 || k == 8 || k == 10; }
__host__ __device__ inline uint32_t lean_valid_mask(int r, int c, int N) {
    const bool u1 = r >= 1, u2 = r >= 2, d1 = r + 1 < N, d2 = r + 2 < N, l1 = c >= 1, l2 = c >= 2, r1 = c + 1 < N, r2 = c + 2 < N;
    return (u1 ? 1u : 0u) | (l1 ? 2u : 0u) | ((u1 && l1) ? 4u : 0u) | ((u1 && r1) ? 8u : 0u) | (u2 ? 16u : 0u) | (l2 ? 32u : 0u) |
           (r1 ? 64u : 0u) | (r2 ? 128u : 0u) | ((d1 && l1) ? 256u : 0u) | (d1 ? 512u : 0u) | ((d1 && r1) ? 1024u : 0u) | (d2 ? 2048u : 0u);
}

// How a (TAB, REST_REG, precision) triple is compiled:
//   standard arithmetic   TAB 1: window table + rest lengths resident in LDS; TAB 0: streamed from L2
//   LEAN arithmetic       (REST_REG, fp32) TAB 0: built for three cloths per CU (168 VGPRs), -1: for four (128), 3: the whole CU for one cloth
//                         (the large grids), 4: two large-grid cloths per CU -- the table streamed from L2 in these --; 2: table in LDS, two per CU
//                         (with 512 threads x 2 particles: eight waves per cloth at 128 VGPRs, the headline variant)
//   LEAN arithmetic, fp64 (round 6) TAB 0 only: the stencil recomputed as above; a spring's rest length = the bit pattern of its type's palette value + a per-spring
//                         offset in ulps (one byte, from an LDS-resident table): bit-identical to the table read it replaces, without the 24 L2 loads per particle
constexpr bool v_lean(int TAB, bool RR, int tsz) { return RR && (tsz == 4 ? (TAB <= 0 || TAB == 2 || TAB == 3 || TAB == 4) : TAB == 0); }
constexpr bool v_ldstab(int TAB) { return TAB == 1 || TAB == 2; }
// the in-kernel metrics' hull stack as u16 indices (same arithmetic, an eighth of the LDS): the variants whose LDS is tight -- two large-grid
// cloths per CU, five / six 25x25 cloths per CU, the fp64 instantiation of the large grids (50x50: 71 KB of scratch instead of 107 KB,
// which is what lets its episode launches exist at all) and the 1024 x 4 variants (64x64)
constexpr bool v_hull_idx(int TAB, int tsz = 4, int NT = 0, int PPT = 0) { return TAB == 4 || TAB <= -2 || (tsz == 8 && NT * PPT > 1024) || NT * PPT >= 4096; }
constexpr int v_waves_per_eu(int NT, int TAB, bool lean, int PPT = 0) {      // __launch_bounds__' second argument: waves per SIMD
    if (!lean && NT == 512 && PPT == 2) return 4;                // eight waves per cloth, two cloths per CU (standard arithmetic)
    if (lean && NT == 512 && PPT == 2 && TAB == 0) return 4;    // the fp64 LEAN build: eight waves per cloth, two cloths per CU (fp32 TAB 0 is a 256-thread build)
    if (!lean || TAB == 3) return NT <= 512 ? 2 : NT / 256;
    if (TAB == 2 || TAB == 4) return NT / 128;                   // two cloths per CU (4: the large grids, table streamed)
    return TAB < 0 ? 3 - TAB : 3;                                // TAB 0, -1, -2, -3: three, four, five, six cloths per CU
}
// NS: 0 = any grid (sizes from the kernel arguments); 25 / 50 = a BASELINE grid known at compile time (cloth_common.hpp: spec_*)
template <typename T, int NT, int PPT, int TAB, bool REST_REG, int FUSED, int NS = 0>
__global__ __launch_bounds__(NT, v_waves_per_eu(NT, TAB, v_lean(TAB, REST_REG, (int)sizeof(T)), PPT)) void k_run_schedule(StepArgs<T> A) {
    static_assert((NS == 0 || NS == 25 || NS == 50) && (NS == 0 || FUSED != 3), "grid-specialised builds exist for 25x25 and 50x50 (stepper_variants.hpp: CLOTH_SPEC_*)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x + A.e0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // fz == nullptr: ONE externally decoded schedule per env (clothhip_run). Otherwise: nT whole ClothEnv.step calls per env with
    // action decoding, grab_top, metrics, terminal test and episode resets in the kernel (clothhip_run_actions).
    const FusedArgs<T> *const Fp = A.fz;
    constexpr bool fused = FUSED != 0;
    constexpr bool with_tier2 = FUSED == 2;
    // FUSED 3: like 1, with the two ORDERED phases relaxed (SURVEY 7-H4's labelled, non-parity data point): self-collision in Jacobi
    // order (every particle corrected against the phase's start positions), strain limit in coloured order (twelve classes of springs
    // that share no particle, each class in parallel). Different trajectories from the reference's: never a product path, bench only.
    constexpr bool RELAXED = FUSED == 3;
    ClothSchedule sc;
    if (!fused) {
        sc = A.sched[e];
        if (!sc.active || sc.n_total <= 0) {
            if (tid == 0) A.executed[e] = 0;
            return;
        }
    } else {
        sc.n_up_end = sc.n_uprest_end = sc.n_pull_end = sc.n_griprest_end = sc.n_total = 0;
        sc.break_on_tear = 1; sc.active = 1; sc._pad = 0;
        sc.dz_up = sc.dx_pull = sc.dy_pull = sc.dz_pull = 0.0;
    }
    const int P = NS > 0 ? spec_p(NS) : A.P, Ppad = NS > 0 ? spec_ppad(NS) : A.Ppad, HT = NS > 0 ? spec_ht(NS, TAB) : A.HT;
    constexpr bool LEAN64 = v_lean(TAB, REST_REG, (int)sizeof(T)) && sizeof(T) == 8;
    const LdsLayout lay((int)sizeof(T), Ppad, KA_SPAD(&A), HT, TAB == 2 ? 2 : (v_ldstab(TAB) ? 1 : 0), KA_CELLCOPY(&A), LEAN64 ? 1 : 0);
    Pt<T> *cur = reinterpret_cast<Pt<T> *>(smem + lay.cur);
    uint32_t *hkey = reinterpret_cast<uint32_t *>(smem + lay.hkey);
    uint32_t *hco = reinterpret_cast<uint32_t *>(smem + lay.hco);
    uint16_t *memb = reinterpret_cast<uint16_t *>(smem + lay.memb);
    uint16_t *slot = reinterpret_cast<uint16_t *>(smem + lay.slot);
    int *misc = reinterpret_cast<int *>(smem + lay.misc);   // [0] tear, [1] #springs flagged by the pre-pass, [2] #active cells, [3] #occupied cells, [4] member cursor, [5],[6] cell tickets, [10],[11] first / last flagged slot
    uint16_t *olist = reinterpret_cast<uint16_t *>(smem + lay.olist);
    uint16_t *alist_end = olist + (Ppad - 1);            // active list grows downwards: entry k = alist_end[-k]
    Pt<T> *cpos = reinterpret_cast<Pt<T> *>(smem + lay.cpos);
    const DevConsts<T> &k = A.k;                         // (every phase of the substep loop shadows this by its own freshly loaded copy: CLOTH_PHASE_ARGS)
    (void)k;
    const T *g_rest = A.rest + (size_t)e * A.rest_stride;
    const WEnt<T> *wtab = reinterpret_cast<const WEnt<T> *>(smem + lay.wtab);    // TAB >= 1 only
    // rest length of the spring in window-table slot i (Hooke, pre-pass; the sweep streams its own)
    auto rest_at = [&](uint32_t i) -> T { return v_ldstab(TAB) ? wtab[i].rest : g_rest[i]; };
#ifdef CLOTHHIP_FORCE_PM            // register-pressure bisection (dev): the phase mask as a compile-time constant
    const int pm = CLOTHHIP_FORCE_PM;
#else
    const int pm = NS > 0 ? (PH_HOOKE | PH_COLLIDE | PH_PLANE | PH_STRAIN) : A.phase_mask;     // (the specialised builds run every phase: debug masks take the generic build)
#endif

    T pvx[PPT], pvy[PPT], pvz[PPT];         // previous positions of the owned particles
    // their incident-spring gather entries (static): in registers for fp32; the fp64 instantiation has no room
    // (they ended up in scratch, reloaded one by one) and re-reads the L2-resident table, 12 loads in flight
    constexpr bool LEAN = v_lean(TAB, REST_REG, (int)sizeof(T));      // (the variants: see v_lean above)
    constexpr bool GT_REG = sizeof(T) == 4 && !LEAN;
    constexpr bool REST_R = REST_REG && !LEAN;
    uint32_t gt[GT_REG ? PPT : 1][HK_SLOTS];
    T rr[REST_R ? PPT : 1][HK_SLOTS];     // and those springs' rest lengths
    uint32_t vm[(LEAN && !LEAN64) ? PPT : 1];   // LEAN: which of the twelve stencil positions exist for the particle (fp64: in the LDS table below)
    // fp64 LEAN: per particle {mask, 12 offset bytes} in LDS (one 16-byte read at the head of the Hooke gather and of the strain pre-pass)
    const uint4 *const lstc = reinterpret_cast<const uint4 *>(smem + lay.lstc);
    auto lean_rest64 = [&](int sl, const uint4 &lw) -> T {
        const uint32_t wsel = sl < 4 ? lw.y : (sl < 8 ? lw.z : lw.w);
        const uint32_t off = (wsel >> (8 * (sl & 3))) & 0xFFu;
        const double base = (double)(lean_bend(sl) ? A.pal_bend : (lean_shear(sl) ? A.pal_shear : A.pal_struct));
        return (T)__longlong_as_double(__double_as_longlong(base) + (long long)off);
    };
    (void)lean_rest64; (void)lstc;
    uint32_t rc[RELAXED ? PPT : 1];         // RELAXED: the particle's grid position, r | c << 8 (parities of the colour classes)
    auto lean_entry = [&](int i, uint32_t vmq, int sl) -> uint32_t {      // a gather entry without its table-slot field
        const bool ok = ((vmq >> sl) & 1u) != 0u;
        return (uint32_t)(ok ? i + lean_off(sl, KA_N(&A)) : i) | (ok ? HK_VALID : 0u) | (sl < HK_SLOTS / 2 ? HK_ASB : 0u) |
               (lean_bend(sl) ? HK_BEND : 0u);
    };
    auto lean_rest = [&](int sl) -> T {
        if constexpr (NS > 0 && sizeof(T) == 4)        // (specialised fp32 builds: the palette as literals, cloth_common.hpp spec_pal -- held as kernel arguments the three
            return (T)(lean_bend(sl) ? spec_pal(NS, 2) : (lean_shear(sl) ? spec_pal(NS, 1) : spec_pal(NS, 0)));   //  values were spilled and restored at every use)
        else return lean_bend(sl) ? A.pal_bend : (lean_shear(sl) ? A.pal_shear : A.pal_struct);
    };
    {   // HBM -> LDS / registers, coalesced
        const T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        const uint8_t *gc = A.cnt + (size_t)e * Ppad;
        for (int i = tid; i < Ppad; i += NT)
            cur[i] = Pt<T>{gp[i], gp[Ppad + i], gp[2 * Ppad + i], w_make<T>(gc[i])};
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * NT;
            const bool ok = i < P;
            pvx[q] = ok ? gq[i] : (T)0; pvy[q] = ok ? gq[Ppad + i] : (T)0; pvz[q] = ok ? gq[2 * Ppad + i] : (T)0;
            if (LEAN64) { }
            else if (LEAN) { const int r_ = i / KA_N(&A); vm[(LEAN && !LEAN64) ? q : 0] = ok ? lean_valid_mask(r_, i - r_ * KA_N(&A), KA_N(&A)) : 0u; if (RELAXED) rc[RELAXED ? q : 0] = (uint32_t)r_ | ((uint32_t)(i - r_ * KA_N(&A)) << 8); }
            else
#pragma unroll
            for (int sl = 0; sl < HK_SLOTS; sl++) {
                const uint32_t g0 = ok ? A.gather[sl * Ppad + i] : 0u;
                if (GT_REG) gt[GT_REG ? q : 0][sl] = g0;
                if (REST_R) rr[REST_R ? q : 0][sl] = g_rest[(g0 >> HK_POS_SHIFT) & HK_POS_MASK];
            }
        }
    }
    // everything in LDS behind the particle records: static tables, hash table, sweep flags (also re-run after the in-kernel
    // metrics, which borrow that region as scratch)
    auto init_lds = [&](int tear_flag, const uint32_t *s_ent, const T *s_rest) {
        if (v_ldstab(TAB) && s_ent != nullptr) {         // (nullptr: the table in LDS is intact, only the scratch behind it is rebuilt)
            WEnt<T> *d0 = reinterpret_cast<WEnt<T> *>(smem + lay.wtab);
            for (int i = tid; i < KA_SPAD(&A); i += NT) { WEnt<T> w_; w_.ab = s_ent[i]; w_.rest = s_rest[i]; d0[i] = w_; }
        }
        for (int h = tid; h < HT; h += NT) { hkey[h] = KEY_EMPTY; hco[h] = 0; }
        if (tid == 0) { misc[0] = tear_flag; misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[4] = 0; misc[5] = 0; misc[6] = 0; misc[10] = 0x7fffffff; misc[11] = -1; misc[12] = 0; misc[13] = 0; misc[14] = 0; misc[20] = 0; misc[21] = 0; misc[22] = 0; misc[23] = 0; }
    };
    if (tid == 0) misc[15] = 0;
    init_lds(A.tear[e], A.wt_ent, g_rest);
    if (LEAN64) {
        uint4 *d_ = reinterpret_cast<uint4 *>(smem + lay.lstc);
        for (int i = tid; i < Ppad; i += NT) d_[i] = A.lstc[i];
    }
    uint16_t *pslot = reinterpret_cast<uint16_t *>(smem + lay.pslot);       // TAB 2 only
    if (TAB == 2) {
        for (int i = tid; i < Ppad; i += NT) {
            const int r_ = i / KA_N(&A);
            const uint32_t vmi = i < P ? lean_valid_mask(r_, i - r_ * KA_N(&A), KA_N(&A)) : 0u;
#pragma unroll
            for (int sl = 0; sl < HK_SLOTS / 2; sl++)      // the sl-th stencil position = the popcount(valid below sl)-th entry of the compacted table
                pslot[sl * Ppad + i] = ((vmi >> sl) & 1u) ? (uint16_t)((A.gather[__popc(vmi & ((1u << sl) - 1u)) * Ppad + i] >> HK_POS_SHIFT) & HK_POS_MASK) : (uint16_t)0;
        }
    }
    __syncthreads();

    int st_windows = 0, st_passes = 0, st_commits = 0;      // wave 0 only (uniform); the number of sweeps run lives in misc[15]
#ifdef CLOTHHIP_TPH_LDS
    const TphLds tph{reinterpret_cast<unsigned long long *>(smem + lay.tphs), tid == 0};
    if (tid < 12) tph.base[tid] = 0ull;
    unsigned long long tlast = 0, tstart = 0;
#else
    unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0;
#endif
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart)::"memory");   // whole-schedule clock of this cloth (stats[15])
#ifdef CLOTHHIP_DIAG_PLACEMENT
    const unsigned long long diag_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef CLOTHHIP_PHASE_STAMPS                 // profiling build (make stamps): phase mask bit 32 turns the stamps on
    const bool timing = (pm & PH_TIME) != 0;
#else
    constexpr bool timing = false;          // the stamp accumulators would cost the hot loops two dozen SGPRs
#endif
#ifdef CLOTHHIP_SWEEP_STAMPS            // profiling build of the strain sweep: cycles per quiet / correcting pass
    constexpr bool SWEEP_TIMED = true;
#else
    constexpr bool SWEEP_TIMED = false;
#endif
#ifndef CLOTHHIP_SWEEP_LEAN
#define CLOTHHIP_SWEEP_LEAN 1           // A/B: 0 = strain_sweep everywhere, 2 = the lean walk for fp32 only
#endif
#if !defined(CLOTHHIP_SWEEP_STAMPS) && !defined(CLOTHHIP_SWEEP_OUTER) && !defined(CLOTHHIP_CELL_COUNTERS)
    constexpr bool SWEEP_LEAN = CLOTHHIP_SWEEP_LEAN != 0 && (CLOTHHIP_SWEEP_LEAN != 2 || sizeof(T) == 4);
#else
    constexpr bool SWEEP_LEAN = false;  // (the sweep-stamps and census builds instrument strain_sweep)
#endif
#ifndef CLOTHHIP_SWEEP_AHEAD_MIN_TAB
#define CLOTHHIP_SWEEP_AHEAD_MIN_TAB 1      // TAB below this (the four-wave builds for three to six cloths per CU: 768 cloths 24.9 -> 25.2 M/s, 1 024: 31.4 -> 31.7, 1 280: +-0,
                                            // 1 536: 33.8 -> 34.2): the lean walk without its read-ahead; the eight-wave headline build loses 2.8 % without it
#endif
    constexpr bool SWEEP_AHEAD = TAB >= CLOTHHIP_SWEEP_AHEAD_MIN_TAB;
#ifndef CLOTHHIP_PRECHECK2_MAX_TAB
#define CLOTHHIP_PRECHECK2_MAX_TAB -3       // TAB at or below this: the collision pre-check takes two members per trip instead of four (substep_collision.inc.hpp)
#endif
    (void)SWEEP_AHEAD;
#if defined(CLOTHHIP_SWEEP_MW) && !defined(CLOTHHIP_SWEEP_STAMPS) && !defined(CLOTHHIP_SWEEP_OUTER)
    constexpr bool SWEEP_MW = true;     // A/B build (round 5): every wave of the cloth looks ahead one window each (strain_sweep_mw);
                                        // bit-identical, measured -8 % on the headline workload: not the production path
#else
    constexpr bool SWEEP_MW = false;    // the one-wave walk (strain_sweep)
#endif
#if defined(CLOTHHIP_PHASE_STAMPS) || defined(CLOTHHIP_CELL_COUNTERS)   // the sweep's window / pass / correction counters cost its loop three instructions per pass:
    constexpr bool SWEEP_STATS = true;  // profiling builds only (the production build counts sweeps)
#else
    constexpr bool SWEEP_STATS = false;
#endif
#define TSTAMP(slot_)                                                          \
    if (timing) {                                                              \
        unsigned long long tn_;                                                \
        __builtin_amdgcn_sched_barrier(0);                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                     \
        tph[slot_] += tn_ - tlast; tlast = tn_;                                \
    }
    if (timing) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory"); }

    // ---- episode state machine (fused mode). Every trip of the loop below is ONE operation, so that a single copy of the
    // action decoding, of the grab, of the substep loop and of the metrics serves actions, reset pulls and settling:
    //   OP_SCHED        the externally decoded schedule of clothhip_run (not fused): run, then leave
    //   OP_ACTION       ClothEnv.step: action -> decode -> grab_top -> run -> metrics -> record, terminal test
    //   OP_RESET_COND   tier-1 reset: "third pull only if coverage >= 0.90" (cloth_env.py:866): metrics, then decide
    //   OP_RESET_PULL   step(action, initialize=True) of a scripted reset pull: decode -> grab_top -> run
    //   OP_RESET_SETTLE bare update() calls after the pulls (tier 3)
    //   OP_RESET_END    start coverage / variance of the new episode (cloth_env.py:780-782)
    // All control decisions derive from values every thread holds identically (kernel arguments, global tables, LDS
    // broadcasts), so the whole workgroup takes the same path through every barrier.
    enum { OP_SCHED = 0, OP_ACTION, OP_RESET_COND, OP_RESET_PULL, OP_RESET_SETTLE, OP_RESET_END };
    EpState *const eps = reinterpret_cast<EpState *>(smem + lay.eps);
    // thread 0, at the end of an operation (or where a time slice cuts it): everything since the last boundary goes to its class
    auto account = [&](int op_, int n_sub) {
        const int cls = op_ == OP_ACTION ? 0 : ((op_ == OP_RESET_PULL || op_ == OP_RESET_COND) ? 1 : (op_ == OP_RESET_SETTLE ? 2 : 3));
        const uint64_t now_ = __builtin_amdgcn_s_memrealtime();
        eps->ticks[cls] += now_ - eps->t_mark; eps->t_mark = now_; eps->subs[cls] += (uint32_t)n_sub;
    };
    if (fused) {
        if (tid == 0) {
            eps->t_slot = 0; eps->rp = -1; eps->n_resets = 0; eps->chain_ok = 1; eps->rs_pulls = 0; eps->reset_mark = 0;
            eps->ep_steps = Fp->num_steps[e]; eps->ep_done = Fp->done[e] ? 1 : 0; eps->done_total = 0; eps->stop = 0;
            misc[7] = 0;
            eps->t_mark = __builtin_amdgcn_s_memrealtime();
            for (int q = 0; q < 4; q++) { eps->ticks[q] = 0; eps->subs[q] = 0; }
            eps->last_cov = __longlong_as_double(0x7ff8000000000000LL); eps->n_ran = 0;
            eps->swap = Fp->policy_arg != nullptr ? Fp->policy_arg[e] : 0; eps->choice = 0;   // 0 flat tiers, 1 / 2 tier 2 with init_side False / True
            if (Fp->resume != nullptr && Fp->resume[e].valid) {        // continue the operation the previous time slice cut
                const EpResume *rs_ = Fp->resume + e;
                const EpState &o = rs_->eps;
                eps->rp = o.rp; eps->chain_ok = o.chain_ok; eps->rs_pulls = o.rs_pulls; eps->ep_steps = o.ep_steps;
                eps->ep_done = o.ep_done; eps->op = o.op; eps->n_grab = o.n_grab; eps->iters_pull = o.iters_pull;
                eps->decode_err = o.decode_err; eps->side = o.side; eps->pull = o.pull; eps->choice = o.choice; eps->swap = o.swap;
                eps->act[0] = o.act[0]; eps->act[1] = o.act[1]; eps->act[2] = o.act[2]; eps->act[3] = o.act[3];
                if (o.rp >= 0 && Fp->resets != nullptr) Fp->resets[(size_t)e * Fp->n_scripts] = rs_->rr;   // its record, now slot 0
            }
        }
        __syncthreads();
    }
    // resume_it >= 0: the first trip of the loop below continues an interrupted run instead of planning an operation
    int resume_it = -1, resume_done = 0;
    if (FUSED) {
        if (Fp->resume != nullptr && Fp->resume[e].valid) {
            const EpResume *rs_ = Fp->resume + e;
            resume_it = rs_->it; resume_done = rs_->done_partial;
            if (resume_it >= 0) sc = rs_->sc;
        }
        __syncthreads();
        if (tid == 0 && Fp->resume != nullptr) Fp->resume[e].valid = 0;
    }
    if (FUSED) {
        if (tid == 0) {
            eps->t_launch = __builtin_amdgcn_s_memrealtime();   // 100 MHz, constant rate (thread 0 is the only reader)
            // the time slice's DEADLINE beside it (EpState, LDS -- not in `misc`, which the in-kernel metrics overwrite): round 5 read Fp->budget_ticks --
            // a global load on wave 0's path to a barrier -- in EVERY substep (found in the ISA while taking the pipelined loop apart, round 6)
            eps->t_deadline = Fp->budget_ticks != 0 ? eps->t_launch + Fp->budget_ticks : ~0ull;
        }
    }
    int done_nf = 0;                   // executed substeps of the external schedule (not fused)
    for (;;) {
        bool do_run = true;
        const bool resumed_run = FUSED && resume_it >= 0;
        if (fused && resumed_run) {
            sc.n_up_end = __builtin_amdgcn_readfirstlane(sc.n_up_end);
            sc.n_uprest_end = __builtin_amdgcn_readfirstlane(sc.n_uprest_end);
            sc.n_pull_end = __builtin_amdgcn_readfirstlane(sc.n_pull_end);
            sc.n_griprest_end = __builtin_amdgcn_readfirstlane(sc.n_griprest_end);
            sc.n_total = __builtin_amdgcn_readfirstlane(sc.n_total);
            sc.break_on_tear = __builtin_amdgcn_readfirstlane(sc.break_on_tear);
        }
        if (fused && !resumed_run) {
#include "episode_plan.inc.hpp"
        }
        int done = resumed_run ? resume_done : 0;
        int it_next = -1;                  // >= 0: the time slice ended inside this run, which continues there in the next launch
        {
        // (wave-uniform by construction: kept in SGPRs -- as four VGPRs they were spilled and reloaded at the head of every substep)
        auto uni = [](T v) -> T {
            if constexpr (sizeof(T) == 4) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)v)));
            else return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint((double)v)), __builtin_amdgcn_readfirstlane(__double2loint((double)v)));
        };
        const T dz_up = uni((T)sc.dz_up), dxp = uni((T)sc.dx_pull), dyp = uni((T)sc.dy_pull), dzp = uni((T)sc.dz_pull);
        const bool sliced = FUSED && Fp->budget_ticks != 0 && Fp->resume != nullptr;
#ifndef CLOTHHIP_UNIFORM_SCHED
#define CLOTHHIP_UNIFORM_SCHED 1
#endif
        // (round 6, from the ISA: the schedule's phase bounds and the loop counter are the same in every lane, but the compiler does not know --
        //  held in VGPRs they turned the "which phase is substep `it` in" ladder at the head of every substep and the loop's exit test into
        //  vector compares + exec-mask branches, ~40 cycles each, and occupied seven VGPRs across the whole loop: scalar from here on)
        const int n_up_end_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.n_up_end) : sc.n_up_end;
        const int n_uprest_end_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.n_uprest_end) : sc.n_uprest_end;
        const int n_pull_end_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.n_pull_end) : sc.n_pull_end;
        const int n_griprest_end_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.n_griprest_end) : sc.n_griprest_end;
        const int n_total_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.n_total) : sc.n_total;
        const int break_on_tear_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(sc.break_on_tear) : sc.break_on_tear;
        const int it0_ = CLOTHHIP_UNIFORM_SCHED ? __builtin_amdgcn_readfirstlane(resumed_run ? resume_it : 0) : (resumed_run ? resume_it : 0);
    const int tid_outer_ = tid;
#ifdef CLOTHHIP_CELL_COUNTERS
    bool frozen_prev_ = false; (void)frozen_prev_;
#endif
    for (int it = it0_; it < n_total_; it++) {
        // LEAN and fp64: everything derived from the thread index (LDS addresses of the owned particles, table offsets) is formed anew in
        // every substep instead of being hoisted out of the loop and held -- or spilled -- for the whole schedule
        int tid = tid_outer_;
        if (LEAN || sizeof(T) == 8 || NT >= 512) asm volatile("" : "+v"(tid));     // (fp64: 65 -> 0 spilled registers; 50x50: +3 %)
        const int lane = tid & 63;
        KArgsC<T> *Ak_ = (KArgsC<T> *)__builtin_amdgcn_kernarg_segment_ptr();

#include "substep_pull.inc.hpp"
        TSTAMP(0)
#include "substep_hooke_verlet.inc.hpp"
        TSTAMP(1)
#include "substep_collision.inc.hpp"
#include "substep_plane.inc.hpp"
        if (FUSED && sliced) {             // thread 0 looks at the clock here, between two barriers that every thread passes in
                                           // every substep; everyone reads its verdict at the end of the substep
            if (tid == 0) misc[7] = (__builtin_amdgcn_s_memrealtime() >= eps->t_deadline) ? 1 : 0;
        }
        __syncthreads();

        TSTAMP(7)
#include "substep_strain.inc.hpp"
        if (timing) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory"); }
        done++;
        if (break_on_tear_ && misc[0]) break;                                            // cloth_env.py:511-514
        if (FUSED && sliced && misc[7] && it + 1 < n_total_) { it_next = it + 1; break; }
    }
        }   // the run
        resume_it = -1;
        if (!fused) { done_nf = done; break; }
#include "episode_finish.inc.hpp"
    }
    const int done = fused ? eps->done_total : done_nf;

#undef TSTAMP
    {   // LDS / registers -> HBM
        __syncthreads();
        T *gp = A.pos + (size_t)e * 3 * Ppad, *gq = A.prev + (size_t)e * 3 * Ppad;
        uint8_t *gc = A.cnt + (size_t)e * Ppad;
        for (int i = tid; i < Ppad; i += NT) {
            const Pt<T> c = cur[i];
            gp[i] = c.x; gp[Ppad + i] = c.y; gp[2 * Ppad + i] = c.z; gc[i] = (uint8_t)w_cnt(c.w);
        }
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * NT;
            if (i < P) { gq[i] = pvx[q]; gq[Ppad + i] = pvy[q]; gq[2 * Ppad + i] = pvz[q]; }
        }
        if (tid == 0 && fused) {
            Fp->num_steps[e] = eps->ep_steps; Fp->done[e] = (uint8_t)eps->ep_done;
            if (Fp->summary != nullptr) {
                double *sm_ = Fp->summary + 4 * (size_t)e;
                sm_[0] = (double)eps->n_ran; sm_[1] = eps->ep_done ? 1.0 : 0.0; sm_[2] = eps->last_cov; sm_[3] = (double)eps->subs[0];
            }
            if (Fp->op_ticks != nullptr) {
                eps->ticks[3] += __builtin_amdgcn_s_memrealtime() - eps->t_mark;     // what is left: rebuilds, idling out of action slots
                for (int q = 0; q < 4; q++) { Fp->op_ticks[8 * e + q] = eps->ticks[q]; Fp->op_ticks[8 * e + 4 + q] = eps->subs[q]; }
            }
        }
        if (tid == 0) {
            A.tear[e] = misc[0]; A.executed[e] = done;
            if (A.stats) {
                A.stats[16 * e] = misc[15]; A.stats[16 * e + 1] = st_windows; A.stats[16 * e + 2] = st_passes; A.stats[16 * e + 3] = st_commits;
                for (int q = 0; q < 12; q++) A.stats[16 * e + 4 + q] = (int)((unsigned long long)tph[q] >> 6);
#ifndef CLOTHHIP_PHASE_STAMPS
                unsigned long long tend;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend)::"memory");
                A.stats[16 * e + 15] = (int)((tend - tstart) >> 10);   // shader clocks / 1024 this cloth's schedule took
#endif
#ifdef CLOTHHIP_DIAG_PLACEMENT          // dev: where and when this workgroup ran (tools/placement.py)
                unsigned hw_, xcc_;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
                A.stats[16 * e + 12] = (int)xcc_; A.stats[16 * e + 13] = (int)hw_;
                A.stats[16 * e + 14] = (int)(diag_t0_ & 0x7fffffffull); A.stats[16 * e + 11] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffffull);
#endif
            }
        }
    }
}

}  // namespace clothhip
