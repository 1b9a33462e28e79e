// cloth_kernels.hpp -- gfx950 device code of the cloth stepper.
//
// One workgroup steps ONE cloth through a whole schedule (up to ~2000 substeps) with the particle state
// resident in LDS; HBM is touched once on entry and once on exit.  Every phase keeps the reference's
// evaluation order (cloth.pyx:169-214), so the double instantiation reproduces the reference bit for bit
// (compiled with -ffp-contract=off) and the float instantiation is the same algorithm in fp32.
//
//   phase            reference                parallelisation (exact-order preserving)
//   adjust/release   gripper.pyx:55-73        per point (owner thread)
//   gravity+Hooke    cloth.pyx:216-237        per-point gather of <=12 springs in ascending list index
//   Verlet           cloth.pyx:239-256        per point (fused with the gather)
//   spatial map      cloth.pyx:298-311        LDS hash table keyed by the exact cell key + occupied-cell list; members
//                                             of a cell stored contiguously (rank from the counting atomic)
//   self-collision   cloth.pyx:313-343        parallel seed test, then an exact Gauss-Seidel sweep of the cells that
//                                             have a seed: to-visit set = seeds + later neighbours of members that moved;
//                                             four small cells per wave / one wave per large cell, LDS tickets
//   plane            cloth.pyx:345-370        per point (owner thread: it holds the previous position)
//   strain limit     cloth.pyx:258-296        dependency levels packed into 64-slot windows (one spring per lane, lane order ==
//                                             level order) walked by ONE wave from the first over-stretched spring to the last
//                                             window its corrections can reach; a pass finishes every spring none of whose
//                                             predecessors in the window (static per-slot dependency masks) is over-stretched
// DESIGN.md section 4 has the exactness argument of every phase.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/clothhip.h"
#include "cloth_rng.hpp"
#include "cloth_tables.hpp"

namespace clothhip {

template <typename T> struct DevConsts {
    T mg;              // mass * gravity                         cloth.pyx:179
    T ks_str, ks_bend; // ks * 1.0, ks * 0.2 (no array: a dynamic index would push the struct to scratch)  cloth.pyx:225-232
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

struct EpResume;

template <typename T> struct FusedArgs {
    int32_t nT, policy, NS, NH;       // action slots per launch, CLOTHHIP_POLICY_*, metrics sort / hull buffer sizes
    const double *actions;            // [nT][E][4]
    const int32_t *policy_arg;        // [E] or nullptr
    const ClothResetScript *scripts;  // [E][n_scripts] or nullptr: the env's next resets, in order (see clothhip.h)
    int32_t *num_steps;               // [E]
    uint8_t *done;                    // [E]
    ClothStepRecord *records;         // [nT][E]
    ClothResetRecord *resets;         // [E][n_scripts] or nullptr
    float *obs;                       // [nT][E][3P] or nullptr
    float *reset_obs;                 // [E][n_scripts][3P] or nullptr
    const T *flat;                    // [3][Ppad] flat grid
    const double *levels;             // Gripper.grab_top curZ table
    int32_t n_glevels, E;
    int32_t n_scripts, _pad;
    // copies of StepArgs' static-table pointers: the LDS re-initialisation after the in-kernel metrics loads them from
    // here (plain global loads at the point of use) instead of keeping the kernel arguments alive across the substep loop
    const uint32_t *wt_ent; const T *rest; int32_t rest_stride, _pad3;
    T *rest_rw;                       // the same table, writable: a tier-2 reset rebuilds the env's rest lengths (cloth.pyx:417)
    double grid_dx, grid_dy;          // width / (N - 1), height / (N - 1) (cloth.pyx:55-56)
    uint32_t *mt;                     // [E][MT_WORDS] numpy RandomState of every env, or nullptr (resets come from `scripts`)
    uint64_t domrand_words;           // 32-bit words the domain-randomisation draws after a reset consume (cloth_env.py:786-789), or 0
    int32_t rng_tier, _pad2;          // with mt: 1 or 3, the reset procedure to draw (cloth_env.py:843-891, :951-982)
    EpResume *resume;                 // [E] or nullptr: operations cut by the previous launch's time slice / to be cut by this one
    double *summary;                  // [E][4] or nullptr: per env {actions executed by this launch, episode over (0/1), coverage after its last
                                      // action or reset of this launch (NaN: none), Cloth.update() calls of its actions}: what the multi-GPU driver gathers
    uint64_t *op_ticks;               // [E][8] or nullptr: per env, 100 MHz ticks of this launch spent in {actions, reset pulls, reset settling, the
                                      // rest (episode rebuild, idling out of slots)} and the Cloth.update() calls executed in each
    uint64_t budget_ticks;            // 0 = none; else no new action / reset starts once the launch has run this many 100 MHz ticks
    double two_thickness, half_thickness;
    ClothEpisodeParams ep;
};

// episode state of one cloth between the operations of the fused loop: kept in LDS, not in registers, so that nothing of it
// is live across the substep loop
struct EpState {
    int32_t t_slot;        // next action slot of this launch
    int32_t rp;            // reset stage: -1 none; 2p = coverage condition of pull p, 2p+1 = pull p, 6 = settle, 7 = end
    int32_t n_resets;      // resets done in this launch
    int32_t chain_ok;      // 1 while every reset of this launch ran its unconditional pulls only: the next script is valid
    int32_t rs_pulls;      // pulls run by the reset in progress
    int32_t reset_mark;    // the next executed action record gets reset_before = this
    int32_t ep_steps, ep_done;
    int32_t op, n_grab, iters_pull, decode_err;
    int32_t done_total;
    int32_t stop;          // the launch's time slice is used up: no new action or reset starts
    int32_t side;          // device-RNG resets: Cloth.init_side of the reset in progress (cloth.pyx:75)
    int32_t choice;        // tier-2 reset: the corner picked for the first pull (-25 or -1, cloth_env.py:907)
    int32_t swap, n_ran;   // n_ran: actions executed by this launch (per-launch, not carried over). swap: how the cloth was built, for the policies: 0 flat tiers, 1 tier 2 with init_side False (the oracle-corner
                           // policy swaps its corner indices, analytic.py:108-114), 2 tier 2 with init_side True
    double act[4];
    ClothResetPull pull;   // device-RNG resets: the draws of the pull being executed
    uint64_t t_mark;       // per-operation accounting of this launch (not carried across launches): last boundary,
    uint64_t ticks[4];     //   ticks per class (0 action, 1 reset pull incl. its coverage test, 2 reset settling, 3 other),
    uint32_t subs[4];      //   update() calls per class
    double last_cov;       // coverage after the last action / reset of this launch (NaN: none yet)
    uint64_t t_launch;     // 100 MHz clock when this cloth's workgroup started (the time slice counts from here); LDS, not a register pair:
                           // held in registers it was spilled, and its reload sat on every substep's path
};

// An operation cut by the end of a time slice (clothhip_run_actions with a time budget): everything needed to continue it in
// the next launch. The particle state itself goes through pos / prev / cnt / tear as for any launch end; a substep
// boundary is a complete state (the hash table and sweep flags are rebuilt every substep).
struct EpResume {
    int32_t valid;             // 0 none; 1 an operation of this env is in flight
    int32_t it;                // >= 0: the substep loop of `sc` continues at this iteration; -1: between two operations of a reset
    int32_t done_partial;      // update() calls the interrupted run had executed
    int32_t _pad;
    ClothSchedule sc;
    EpState eps;
    ClothResetRecord rr;       // the partly filled record of the reset in flight (eps.rp >= 0)
};

template <typename T> struct StepArgs {
    T *pos;                  // [E][3][Ppad]   (HBM layout: SoA, coalesced)
    T *prev;                 // [E][3][Ppad]
    uint8_t *cnt;            // [E][Ppad]  bits0..6 multiplicity in grabbed_pts, bit7 pinned from outside
    const T *rest;           // [E or 1][Spad] rest lengths in window-table SLOT order (0 in empty slots)
    int32_t *tear;           // [E] sticky Cloth.cloth_have_tear
    int32_t *executed;       // [E]
    int32_t *stats;          // [E][16] or nullptr: [0] sweeps run, [1] windows walked, [2] passes, [3] passes that corrected;
                             // [4..15] with PH_TIME: shader cycles/64 spent per phase (wave 0's view)
    const ClothSchedule *sched;   // [E]
    const uint32_t *gather;  // [HK_SLOTS][Ppad]
    const uint32_t *wt_ent;  // [Spad] window table of the strain sweep (cloth_tables.hpp), Spad = (nW + padding windows) * 64
    const unsigned long long *wt_dep;   // [Spad] per slot: the lanes of its window the spring transitively depends on
    T pal_struct, pal_shear, pal_bend;  // LEAN variant: the rest length of every structural / shearing / bending spring (one shared table
                                        // whose fp32 values are one per type: checked by the host before the variant is chosen)
    int32_t nW, wt_rshift;   // windows that hold springs; unit (log2 windows) of the entries' reach field
    int32_t N, P, Ppad, S, Spad;
    int32_t HT, ht_bits;     // spatial hash table slots (> P) and log2 of it (0: not a power of two)
    int32_t rest_stride;     // 0: one shared table
    int32_t cell_copy;       // 1: LDS holds a cell-ordered copy of the particle records for the collision pre-check
    int32_t phase_mask;      // debug/ablation: bit0 hooke+verlet, bit1 collide, bit2 plane, bit3 strain, bit4 no-skip
    DevConsts<T> k;
    // whole episodes on the device (clothhip_run_actions): a DEVICE pointer to the episode arguments, or nullptr = one
    // externally decoded schedule per env (clothhip_run). By pointer, not by value: kernel arguments are invariant loads
    // that the compiler hoists to the kernel entry and keeps in SGPRs across the substep loop, which has none to spare.
    const struct FusedArgs<T> *fz;
};

constexpr int KEY_SHIFT = 12;
constexpr uint32_t KEY_BIAS = 1u << 19;
constexpr uint32_t KEY_FLOOR = 4096u;         // stored keys are >= KEY_FLOOR so a slot can later hold a point index (< 4096)
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr uint8_t CNT_GRAB_MASK = 0x7F, CNT_EXT_PIN = 0x80;
enum { PH_HOOKE = 1, PH_COLLIDE = 2, PH_PLANE = 4, PH_STRAIN = 8, PH_NOSKIP = 16, PH_TIME = 32 };

// double: correctly rounded IEEE sqrt / division (bit parity with the reference's CPython doubles).
// float : the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (the fp32 instantiation is the throughput mode; its
//         parity is a tolerance, not bits).
template <typename T> __device__ __forceinline__ T dev_sqrt(T x);
template <> __device__ __forceinline__ double dev_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float dev_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <typename T> __device__ __forceinline__ T dev_div(T a, T b);
template <> __device__ __forceinline__ double dev_div<double>(double a, double b) { return a / b; }
template <> __device__ __forceinline__ float dev_div<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
template <typename T> __device__ __forceinline__ T dev_floor(T x);
template <> __device__ __forceinline__ double dev_floor<double>(double x) { return floor(x); }
template <> __device__ __forceinline__ float dev_floor<float>(float x) { return floorf(x); }
// wave-uniform broadcast of lane `l`'s value (l must be wave-uniform)
__device__ __forceinline__ float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double bcast(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// inclusive OR-scan inside each row of 16 lanes (DPP row_shr 1,2,4,8); lane 16r+15 ends up with row r's OR
__device__ __forceinline__ uint32_t row_or_scan(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// value of lane `src` (per-lane varying) through the LDS crossbar
__device__ __forceinline__ int lane_pull(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ float lane_pull(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
__device__ __forceinline__ double lane_pull(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)));
}
// inclusive +scan over the 64 lanes of the wave (DPP: row_shr 1,2,4,8, then row_bcast 15 and 31)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);     // lane 15 of rows 0,2 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);     // lane 31 -> rows 2,3
    return v;
}
// inclusive min-scan over the 64 lanes (same DPP steps; lanes without a source keep their own value); lane 63 = the wave's min
__device__ __forceinline__ int wave_incl_min(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false); v = t < v ? t : v;     // lane 15 of rows 0,2 -> rows 1,3
    t = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false); v = t < v ? t : v;     // lane 31 -> rows 2,3
    return v;
}
// fp32 sums over lanes by DPP (no LDS round trips): the whole wave's total (uniform), and the total of each row of 16 lanes
// in every lane of the row (rotations: row_ror 8, 4, 2, 1)
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));
    // (all rows enabled in the two broadcast steps: only lane 63's value is used, and it comes out with the same association as with
    //  the rows masked -- (R3 + R2) + (R1 + R0) -- while the unmasked form fuses into one v_add_f32_dpp per step)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xF, 0xF, true));     // lane 15 of every row -> the next row
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xF, 0xF, true));     // lane 31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float row_allsum_f32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));     // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, true));     // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, true));     // row_ror:1
    return v;
}
// relative slack of the conservative "could this comparison against a sqrt be true" pre-filters
template <typename T> __device__ __forceinline__ T filt_slack();
template <> __device__ __forceinline__ double filt_slack<double>() { return 1e-9; }
template <> __device__ __forceinline__ float filt_slack<float>() { return 1e-5f; }

// Particle record in LDS: position + the pin/grab count in the 4th slot, so ONE 16-byte (fp32) LDS read
// brings everything a phase needs to know about a particle.
template <typename T> struct __attribute__((aligned(16))) Pt { T x, y, z, w; };
__device__ __forceinline__ uint32_t w_cnt(float w) { return __float_as_uint(w); }
__device__ __forceinline__ uint32_t w_cnt(double w) { return (uint32_t)__double2loint(w); }
template <typename T> __device__ __forceinline__ T w_make(uint32_t c);
template <> __device__ __forceinline__ float w_make<float>(uint32_t c) { return __uint_as_float(c); }
template <> __device__ __forceinline__ double w_make<double>(uint32_t c) { return __hiloint2double(0, (int)c); }

// cloth.pyx:17-18, association ((x*x + y*y) + z*z)
// a * b + c: for double two roundings, as the reference's C doubles compute it (the file is built with -ffp-contract=off);
// for float ONE fused multiply-add -- the fp32 instantiation is the throughput mode, its parity a tolerance
template <typename T> __device__ __forceinline__ T mad(T a, T b, T c);
template <> __device__ __forceinline__ double mad<double>(double a, double b, double c) { return a * b + c; }
template <> __device__ __forceinline__ float mad<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// (x*x + y*y) + z*z in the reference's association
template <typename T> __device__ __forceinline__ T sumsq(T x, T y, T z) { return mad<T>(z, z, mad<T>(y, y, x * x)); }
template <typename T> __device__ __forceinline__ T fastnorm(T x, T y, T z) { return dev_sqrt<T>(sumsq<T>(x, y, z)); }

// cloth.pyx:307-311 -> biased, clamped cell key (exact for |coordinate| < ~60 cloth widths)
template <typename T> __device__ __forceinline__ uint32_t cell_key(const DevConsts<T> &k, T x, T y, T z) {
    T fx = dev_floor<T>(x / k.cw), fy = dev_floor<T>(y / k.ch), fz = dev_floor<T>(z / k.ct);
    const T lim = (T)4096;
    fx = fx < -lim ? -lim : (fx > lim ? lim : fx);   // NaN falls through the compares; handled below
    fy = fy < -lim ? -lim : (fy > lim ? lim : fy);
    fz = fz < -lim ? -lim : (fz > lim ? lim : fz);
    if (!(fx == fx) || !(fy == fy) || !(fz == fz)) return (1u << 20) - 1u + KEY_FLOOR;
    int key = 961 * (int)fx + 31 * (int)fy + (int)fz;
    int kb = key + (int)KEY_BIAS;
    kb = kb < 0 ? 0 : (kb > (1 << 20) - 2 ? (1 << 20) - 2 : kb);
    return (uint32_t)kb + KEY_FLOOR;
}

// Window-table entry as the sweep wave streams it: the static word of cloth_tables.hpp + the spring's rest length.
template <typename T> struct WEnt;
template <> struct __attribute__((aligned(8))) WEnt<float> { uint32_t ab; float rest; };
template <> struct __attribute__((aligned(16))) WEnt<double> { uint32_t ab; uint32_t _pad; double rest; };

// The stepper's constants re-read from the kernel-argument block (constant address space: scalar loads) at the head of a phase of the
// substep loop, through a pointer made opaque there: loaded once at the kernel's entry they would occupy SGPRs for the whole launch --
// the hot loop has none to spare, they were spilled (to VGPR lanes, some on to scratch) and reloaded all over the loop.
template <typename T> using KArgsC = const __attribute__((address_space(4))) StepArgs<T>;
template <typename T> __device__ __forceinline__ DevConsts<T> load_consts(KArgsC<T> *p) {
    DevConsts<T> k;
    k.mg = p->k.mg; k.ks_str = p->k.ks_str; k.ks_bend = p->k.ks_bend; k.dsm = p->k.dsm; k.damp = p->k.damp;
    k.cw = p->k.cw; k.ch = p->k.ch; k.ct = p->k.ct; k.thresh = p->k.thresh; k.sim_steps = p->k.sim_steps;
    k.min_z = p->k.min_z; k.surf_off = p->k.surf_off; k.one_m_fric = p->k.one_m_fric; k.tear_thresh = p->k.tear_thresh; k.c11 = p->k.c11;
    return k;
}
// (in a phase's scope: shadows the kernel's `k`, `P`, `Ppad`, `HT` by freshly loaded copies)
#define CLOTH_PHASE_ARGS()                                                        \
    asm volatile("" : "+s"(Ak_));                                                 \
    const DevConsts<T> k = load_consts<T>(Ak_);                                   \
    const int P = Ak_->P, Ppad = Ak_->Ppad, HT = Ak_->HT;                         \
    (void)k; (void)P; (void)Ppad; (void)HT;

constexpr int EPSTATE_LDS_BYTES = 240;
static_assert(sizeof(EpState) <= EPSTATE_LDS_BYTES, "EpState outgrew its LDS slot (LdsLayout::eps): the window table / hash region follows it");
static_assert(WT_IDX_BITS == 12 && HK_NBR_MASK == WT_IDX_MASK, "point indices are 12 bits in the gather entries and in the window table alike");
// LDS carve-up (dynamic shared memory), all offsets in bytes, 16-byte aligned.
// tab: 0 = the window table stays in global memory (L2), 1 = table + rest lengths resident in LDS
#if defined(CLOTHHIP_PHASE_STAMPS) || defined(CLOTHHIP_CELL_COUNTERS) || defined(CLOTHHIP_SWEEP_STAMPS)
#define CLOTHHIP_TPH_LDS 1
#endif
struct LdsLayout {
    int lkey;        // census build: every particle's cell key of the previous substep
    int tphs;        // profiling / census builds: their twelve 64-bit accumulators (in front of the region the in-kernel metrics borrow)
    int cur, eps, wtab, pslot, hkey, hco, memb, slot, misc, alist, olist, cpos, total;
    // tab 2 (the eight-wave LEAN build): like 1, plus the table slots of every particle's six own springs (u16 [6][Ppad]): the strain
    // pre-pass of the LEAN arithmetic needs the slot of a flagged spring, and read it from the L2-resident gather table otherwise
    __host__ __device__ LdsLayout(int tsz, int Ppad, int Spad, int HT, int tab, int cp) {
        int o = 0;
        auto take = [&](int bytes) { int r = o; o += (bytes + 15) / 16 * 16; return r; };
        cur = take(4 * Ppad * tsz);
        eps = take(EPSTATE_LDS_BYTES);   // EpState (fused episodes)
        wtab = take(tab >= 1 ? Spad * (tsz == 8 ? 16 : 8) : 0);   // WEnt<T>[Spad]
        pslot = take(tab == 2 ? (HK_SLOTS / 2) * Ppad * 2 : 0);
#ifdef CLOTHHIP_TPH_LDS
        tphs = take(96);
#else
        tphs = 0;
#endif
#ifdef CLOTHHIP_CELL_COUNTERS
        lkey = take(4 * Ppad);
#else
        lkey = 0;
#endif
        hkey = take(HT * 4);         // everything from here on doubles as scratch of the in-kernel metrics and is rebuilt afterwards
        hco = take(HT * 4);          // (fill cursor << 16) | member count
        memb = take(Ppad * 2);
        slot = take(Ppad * 2);
        misc = take(256);            // flags and scan scratch (64 ints)
        olist = take(2 * Ppad);       // u16 hash slots: occupied cells from the front, cells with a seed from the back
        alist = olist;                //   (an active cell has >= 2 members, so #occupied + #active <= P)
        cpos = take(cp ? 4 * (Ppad + 32) * tsz : 0);   // particle records in cell (CSR) order for the pre-check; the
                                                       // unclamped member loop may read up to a cell's width past the end
        total = o;
    }
};

// Accumulators of the profiling / census builds (per-phase cycles, counters): in LDS, written by thread 0 alone -- as twelve 64-bit
// registers per wave they cost the VGPR-capped variants two dozen SGPRs and turned the profile into one of the spills they caused.
struct TphRef {
    unsigned long long *a; bool w;
    __device__ __forceinline__ void operator+=(unsigned long long v) const { if (w) *a += v; }
    __device__ __forceinline__ operator unsigned long long() const { return *a; }
};
struct TphLds {
    unsigned long long *base; bool w;
    __device__ __forceinline__ TphRef operator[](int i) const { return TphRef{base + i, w}; }
};
#ifdef CLOTHHIP_TPH_LDS
typedef TphLds TphT;
#else
typedef unsigned long long *TphT;
#endif

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
template <typename T, bool LDS_TAB, bool STATS>
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
        decode_read(n);                                             // speculative: valid unless this window corrects something
        T ax = c.A.x, ay = c.A.y, az = c.A.z, bx = c.B.x, by = c.B.y, bz = c.B.z;
        T dx = ax - bx, dy = ay - by, dz = az - bz;
        T len2 = sumsq<T>(dx, dy, dz);
        const T t11 = c.rest * c11;
        T len = (T)0; bool trig;
        if constexpr (sizeof(T) == 4) { len = dev_sqrt<T>(len2); trig = len > t11; }
        else trig = len2 > t11 * t11 * ((T)1 - filt_slack<T>());    // fp64: the squared pre-filter decides whether anybody looks closer
        if (STATS) { st_windows++; st_passes++; }
        if (__builtin_expect(ballot64(trig) != 0ull, 0)) {
            // ---- the exact path (strain_sweep's pass loop; its first pass is the evaluation above) ----
            // (measured and rejected, round 5: the commit computed by every lane with the stores of the lanes that must not write sent
            //  to a per-lane sink record -- no exec-mask detour, one branch per pass --: -4 %; the window's "nobody left" exit dropped
            //  in favour of the next pass's "nobody over-stretched": -2.5 %)
            const uint32_t ca = w_cnt(c.A.w), cb = w_cnt(c.B.w);    // pins do not change during a sweep
            T tl = ((ca != 0) & (cb != 0)) ? INF_ : t11;           // both ends pinned: skipped by the reference (:268)
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
                    const bool bad = ((dlo & (uint32_t)tb) | (dhi & (uint32_t)(tb >> 32))) != 0u;
                    if (STATS) st_commits++;
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
                    if (plm == 0ull) break;
                    __builtin_amdgcn_wave_barrier();                // same-wave LDS operations execute in program order
                    const P3 na = *pa, nb = *pb;
                    ax = na.x; ay = na.y; az = na.z; bx = nb.x; by = nb.y; bz = nb.z;
                    dx = ax - bx; dy = ay - by; dz = az - bz;
                    len2 = sumsq<T>(dx, dy, dz);
                    if constexpr (sizeof(T) == 4) len = dev_sqrt<T>(len2);
                    test();
                    tb = ballot64(trig);
                    if (STATS) st_passes++;
                    if (tb == 0ull) break;                          // a quiet pass ends the window
                }
                n.A = cur[n.a]; n.B = cur[n.b];                     // the speculative records are stale now
            }
        }
        load(w + 2, c);                                             // this set is free: the entry of the window after next
    };
    Set S0, S1;
    load(w, S0); load(w + 1, S1);
    decode_read(S0);
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
            const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1ull;
            visits_++;
            const T xa = bcast(x, a), ya = bcast(y, a), za = bcast(z, a);
            const T dx = xa - x, dy = ya - y, dz = za - z;
            const T d2 = sumsq<T>(dx, dy, dz);
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
        const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        todo &= todo - 1ull;
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
    while (__any(todo != 0u)) {
        const bool act = todo != 0u;
        const int a = act ? __ffs((int)todo) - 1 : 0;
        todo &= todo - 1u;
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
constexpr bool v_lean(int TAB, bool RR, int tsz) { return (TAB <= 0 || TAB == 2 || TAB == 3 || TAB == 4) && RR && tsz == 4; }
constexpr bool v_ldstab(int TAB) { return TAB == 1 || TAB == 2; }
constexpr bool v_hull_idx(int TAB) { return TAB == 4 || TAB <= -2; }      // the in-kernel metrics' hull stack as u16 indices (tight LDS)
constexpr int v_waves_per_eu(int NT, int TAB, bool lean, int PPT = 0) {      // __launch_bounds__' second argument: waves per SIMD
    if (!lean && NT == 512 && PPT == 2) return 4;                // eight waves per cloth, two cloths per CU (standard arithmetic)
    if (!lean || TAB == 3) return NT <= 512 ? 2 : NT / 256;
    if (TAB == 2 || TAB == 4) return NT / 128;                   // two cloths per CU (4: the large grids, table streamed)
    return TAB < 0 ? 3 - TAB : 3;                                // TAB 0, -1, -2, -3: three, four, five, six cloths per CU
}
template <typename T, int NT, int PPT, int TAB, bool REST_REG, int FUSED>
__global__ __launch_bounds__(NT, v_waves_per_eu(NT, TAB, v_lean(TAB, REST_REG, (int)sizeof(T)), PPT)) void k_run_schedule(StepArgs<T> A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x;
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
    const int P = A.P, Ppad = A.Ppad, HT = A.HT;
    const LdsLayout lay((int)sizeof(T), Ppad, A.Spad, HT, TAB == 2 ? 2 : (v_ldstab(TAB) ? 1 : 0), A.cell_copy);
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
    const int pm = A.phase_mask;
#endif

    T pvx[PPT], pvy[PPT], pvz[PPT];         // previous positions of the owned particles
    // their incident-spring gather entries (static): in registers for fp32; the fp64 instantiation has no room
    // (they ended up in scratch, reloaded one by one) and re-reads the L2-resident table, 12 loads in flight
    constexpr bool LEAN = v_lean(TAB, REST_REG, (int)sizeof(T));      // (the variants: see v_lean above)
    constexpr bool GT_REG = sizeof(T) == 4 && !LEAN;
    constexpr bool REST_R = REST_REG && !LEAN;
    uint32_t gt[GT_REG ? PPT : 1][HK_SLOTS];
    T rr[REST_R ? PPT : 1][HK_SLOTS];     // and those springs' rest lengths
    uint32_t vm[LEAN ? PPT : 1];            // LEAN: which of the twelve stencil positions exist for the particle
    uint32_t rc[RELAXED ? PPT : 1];         // RELAXED: the particle's grid position, r | c << 8 (parities of the colour classes)
    auto lean_entry = [&](int i, uint32_t vmq, int sl) -> uint32_t {      // a gather entry without its table-slot field
        const bool ok = ((vmq >> sl) & 1u) != 0u;
        return (uint32_t)(ok ? i + lean_off(sl, A.N) : i) | (ok ? HK_VALID : 0u) | (sl < HK_SLOTS / 2 ? HK_ASB : 0u) |
               (lean_bend(sl) ? HK_BEND : 0u);
    };
    auto lean_rest = [&](int sl) -> T { return lean_bend(sl) ? A.pal_bend : (lean_shear(sl) ? A.pal_shear : A.pal_struct); };
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
            if (LEAN) { const int r_ = i / A.N; vm[LEAN ? q : 0] = ok ? lean_valid_mask(r_, i - r_ * A.N, A.N) : 0u; if (RELAXED) rc[RELAXED ? q : 0] = (uint32_t)r_ | ((uint32_t)(i - r_ * A.N) << 8); }
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
            for (int i = tid; i < A.Spad; i += NT) { WEnt<T> w_; w_.ab = s_ent[i]; w_.rest = s_rest[i]; d0[i] = w_; }
        }
        for (int h = tid; h < HT; h += NT) { hkey[h] = KEY_EMPTY; hco[h] = 0; }
        if (tid == 0) { misc[0] = tear_flag; misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[4] = 0; misc[5] = 0; misc[6] = 0; misc[10] = 0x7fffffff; misc[11] = -1; misc[12] = 0; misc[13] = 0; misc[14] = 0; misc[20] = 0; misc[21] = 0; misc[22] = 0; misc[23] = 0; }
    };
    if (tid == 0) misc[15] = 0;
    init_lds(A.tear[e], A.wt_ent, g_rest);
    uint16_t *pslot = reinterpret_cast<uint16_t *>(smem + lay.pslot);       // TAB 2 only
    if (TAB == 2) {
        for (int i = tid; i < Ppad; i += NT) {
            const int r_ = i / A.N;
            const uint32_t vmi = i < P ? lean_valid_mask(r_, i - r_ * A.N, A.N) : 0u;
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
    if (FUSED) { if (tid == 0) eps->t_launch = __builtin_amdgcn_s_memrealtime(); }   // 100 MHz, constant rate (thread 0 is the only reader)
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
            // ---- plan the next operation. Every thread evaluates the same transitions on the same LDS-resident state.
            const FusedArgs<T> &F = *Fp;
            int t_slot = eps->t_slot, rp = eps->rp;
            const int n_resets = eps->n_resets;
            // rp >= 0: the script of the reset in progress; else the env's next one, valid only while the chain is intact
            const bool have_scr = F.scripts != nullptr && n_resets < F.n_scripts && (rp >= 0 || eps->chain_ok);
            const ClothResetScript *scr = have_scr ? F.scripts + ((size_t)e * F.n_scripts + n_resets) : nullptr;
            // device-RNG resets (F.mt): the script is not read from a table but drawn from the env's numpy stream as the reset
            // proceeds, in the reference's order; its shape depends on the tier only
            uint32_t *const mt = F.mt ? F.mt + (size_t)e * MT_WORDS : nullptr;
            const bool rngm = mt != nullptr;
            const int tier = with_tier2 ? F.rng_tier : (F.rng_tier == 3 ? 3 : 1);
            auto s_n_pulls = [&]() { return rngm ? (tier == 1 ? 3 : (tier == 2 ? 2 : 1)) : scr->n_pulls; };
            auto s_settle = [&]() { return rngm ? (tier == 3 ? 800 : (tier == 2 ? 500 : 0)) : scr->settle_after; };
            auto s_need_cov = [&](int p_) { return rngm ? (tier == 1 && p_ == 2) : ((scr->pull[p_].need_coverage & 1) != 0); };
            int op = OP_ACTION;
            bool do_decode = false;
            double act[4] = {0.0, 0.0, 0.0, 0.0};
            double run_iters_up = F.ep.iters_up;
            do_run = false;
            if (rp < 0) {
                if (t_slot >= F.nT) break;
                // time slice: envs advance at their own pace, so a launch ends when its time budget is used up rather than when
                // the slowest env has finished a fixed number of actions. Decided by thread 0 between operations (also between a
                // reset and the first action of the new episode: the reset record tells the host). Which launch executes an
                // action never changes its result.
                if (eps->stop) break;
                if (eps->ep_done) {
                    __syncthreads();                     // everyone has read the state
                    if (rngm ? (n_resets < F.n_scripts) : (scr != nullptr && scr->valid)) {
                        // the Cloth(...) rebuild of ClothEnv.reset (cloth_env.py:737-746): nothing pinned, no tear
                        int side_ = 0;
                        bool t2_ = false;
                        if constexpr (with_tier2) t2_ = rngm && tier == 2;
                        if constexpr (with_tier2) if (t2_) {
                            // tier 2 (cloth.pyx:94-116): a vertical sheet at x = |noise| (init_side) or 1 - |noise|, one rand()
                            // per point in r-major order (row 0 draws too, its noise is zeroed), and rest lengths measured on
                            // these positions (cloth.pyx:417) -- in double, as the host's clothhip_init_grid does, through a
                            // scratch copy behind the particle records
                            double *dpos = reinterpret_cast<double *>(smem + lay.wtab);
                            if (tid == 0) {
                                side_ = mt_double(mt) > 0.5 ? 1 : 0;                             // cloth.pyx:75
                                const int N_ = A.N;
                                for (int r_ = 0; r_ < N_; r_++)
                                    for (int c_ = 0; c_ < N_; c_++) {
                                        double noise = mt_double(mt) * 0.01 - 0.005;             // :101
                                        if (r_ == 0) noise = 0;                                  // :102-103
                                        const int i = r_ * N_ + c_;
                                        dpos[3 * i] = side_ ? 0.0 + fabs(noise) : 1.0 - fabs(noise);   // :104-107
                                        dpos[3 * i + 1] = F.grid_dx * c_; dpos[3 * i + 2] = F.grid_dy * r_;   // :109-110
                                    }
                                eps->side = side_;
                            }
                            __syncthreads();
                            side_ = eps->side;
                            for (int i = tid; i < Ppad; i += NT)
                                cur[i] = i < P ? Pt<T>{(T)dpos[3 * i], (T)dpos[3 * i + 1], (T)dpos[3 * i + 2], w_make<T>(0u)}
                                               : Pt<T>{(T)0, (T)0, (T)0, w_make<T>(0u)};
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int i = tid + q * NT;
                                if (i < P) { pvx[q] = (T)dpos[3 * i]; pvy[q] = (T)dpos[3 * i + 1]; pvz[q] = (T)dpos[3 * i + 2]; }
                            }
                            T *rw = F.rest_rw + (size_t)e * F.rest_stride;
                            for (int p_ = tid; p_ < A.Spad; p_ += NT) {
                                const uint32_t en = F.wt_ent[p_];                                 // empty slots: ptA == ptB == 0 -> 0
                                const double *PA = dpos + 3 * (en & WT_IDX_MASK), *PB = dpos + 3 * ((en >> WT_IDX_BITS) & WT_IDX_MASK);
                                const double ux = PA[0] - PB[0], uy = PA[1] - PB[1], uz = PA[2] - PB[2];
                                rw[p_] = (T)sqrt(ux * ux + uy * uy + uz * uz);                    // cloth.pyx:417 via :17-18
                            }
                            __syncthreads();
                            init_lds(0, F.wt_ent, F.rest + (size_t)e * F.rest_stride);
                            if (REST_R) {
#pragma unroll
                                for (int q = 0; q < PPT; q++)
#pragma unroll
                                    for (int sl = 0; sl < HK_SLOTS; sl++) {
                                        const uint32_t g0 = GT_REG ? gt[GT_REG ? q : 0][sl] : 0u;
                                        rr[REST_R ? q : 0][sl] = rw[(g0 >> HK_POS_SHIFT) & HK_POS_MASK];
                                    }
                            }
                        }
                        if (!t2_) {
                            for (int i = tid; i < Ppad; i += NT)
                                cur[i] = Pt<T>{F.flat[i], F.flat[Ppad + i], F.flat[2 * Ppad + i], w_make<T>(0u)};
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int i = tid + q * NT;
                                if (i < P) { pvx[q] = F.flat[i]; pvy[q] = F.flat[Ppad + i]; pvz[q] = F.flat[2 * Ppad + i]; }
                            }
                        }
                        if (tid == 0) {
                            misc[0] = 0;
                            eps->rp = t2_ ? 8 : 0; eps->rs_pulls = 0; eps->ep_steps = 0; eps->ep_done = 0;
                            if (rngm && !t2_) side_ = mt_double(mt) > 0.5 ? 1 : 0;               // cloth.pyx:75
                            eps->side = side_;
                            if (t2_) eps->swap = side_ ? 2 : 1;
                            if (F.resets) {
                                ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + n_resets);
                                rr_->init_side = side_;
                                rr_->consumed = 1; rr_->pulls_run = 0; rr_->executed[0] = rr_->executed[1] = rr_->executed[2] = 0;
                                rr_->settle_executed = 0; rr_->tear = 0;
                            }
                        }
                    } else if (tid == 0) {               // episode over and no script left: the slot stays empty
                        ClothStepRecord *r_ = F.records + ((size_t)t_slot * F.E + e);
                        r_->ran = 0; r_->executed = 0; r_->n_grabbed = 0; r_->done = 1; r_->reset_before = 0;
                        eps->t_slot = t_slot + 1;
                    }
                    __syncthreads();
                    continue;
                }
                do_decode = true;
                if (F.policy == CLOTHHIP_POLICY_ORACLE_CORNER) {
                    // examples/analytic.py:105-155 ('distance' method, delta actions): pull the inset corner that is
                    // farthest from its plane corner; candidates in the order ur, lr, ll, ul, the first maximum wins
                    const bool sw = eps->swap == 1;                                       // tier 2, init_side False (:108-114)
                    double best = -1.0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int ci = c == 0 ? (sw ? 48 : 598) : (c == 1 ? (sw ? 26 : 576) : (c == 2 ? (sw ? 576 : 26) : (sw ? 598 : 48)));
                        const double tgx = c < 2 ? 1.0 : 0.0, tgy = (c == 0 || c == 3) ? 1.0 : 0.0;
                        const Pt<T> pc = cur[ci];
                        const double x = (double)pc.x, y = (double)pc.y;
                        const double cx = (x - 0.5) * 2.0, cy = (y - 0.5) * 2.0;                 // analytic.py:53-54
                        double dx = tgx - x, dy = tgy - y;                                        // :55-56
                        const double dist = sqrt((x - tgx) * (x - tgx) + (y - tgy) * (y - tgy)); // :57
                        dx = dx * 0.90; dy = dy * 0.90;                                           // :64-66
                        if (dist > best) {
                            best = dist;
                            act[0] = F.ep.clip_act_space ? cx : x; act[1] = F.ep.clip_act_space ? cy : y;   // :151-154
                            act[2] = dx; act[3] = dy;
                        }
                    }
                } else if (with_tier2 && F.policy == CLOTHHIP_POLICY_HIGHEST_POINT) {
                    // examples/analytic.py:792-808: sorted(pts, key=z, reverse=True)[k] -- a stable sort, so equal heights keep
                    // their index order -- with k (the reference: np.random.randint(top_k)) from the caller's table, pulled to
                    // where that point sits on the flat cloth (:742-789). k + 1 rounds of a workgroup arg-max over (z, -index),
                    // each excluding what the earlier rounds took; the per-wave results go through the member list (scratch
                    // between substeps).
                    struct Cand { T z; int i; int pad; };
                    Cand *red = reinterpret_cast<Cand *>(memb);
                    int kc = F.policy_arg[(size_t)(1 + t_slot) * F.E + e];
                    kc = kc < 0 ? 0 : (kc > P - 1 ? P - 1 : kc);
                    T lastz = (T)0; int lasti = -1;
                    const auto better = [](T z1, int i1, T z0, int i0) { return i1 != 0x7fffffff && (i0 == 0x7fffffff || z1 > z0 || (z1 == z0 && i1 < i0)); };
                    for (int round = 0; round <= kc; round++) {
                        T bz = (T)0; int bi = 0x7fffffff;
#pragma unroll
                        for (int q = 0; q < PPT; q++) {
                            const int i = tid + q * NT;
                            if (i < P) {
                                const T z = cur[i].z;
                                const bool ok = lasti < 0 || z < lastz || (z == lastz && i > lasti);
                                if (ok && better(z, i, bz, bi)) { bz = z; bi = i; }
                            }
                        }
                        for (int o = 32; o > 0; o >>= 1) {
                            const T oz = __shfl_xor(bz, o); const int oi = __shfl_xor(bi, o);
                            if (better(oz, oi, bz, bi)) { bz = oz; bi = oi; }
                        }
                        if (lane == 0) { red[tid >> 6].z = bz; red[tid >> 6].i = bi; }
                        __syncthreads();
                        bz = red[0].z; bi = red[0].i;
                        for (int w = 1; w < NT / 64; w++) { const T oz = red[w].z; const int oi = red[w].i; if (better(oz, oi, bz, bi)) { bz = oz; bi = oi; } }
                        lastz = bz; lasti = bi;
                        __syncthreads();
                    }
                    const int pr = lasti / A.N, pc_ = lasti - pr * A.N;
                    const Pt<T> pp = cur[lasti];
                    const double x = (double)pp.x, y = (double)pp.y;
                    double tgx, tgy;
                    if (eps->swap == 0) { tgx = F.grid_dx * pr; tgy = F.grid_dy * pc_; }                   // pt.orig_x, pt.orig_y of the flat grid (cloth.pyx:122-124)
                    else { tgx = eps->swap == 2 ? F.grid_dy * pr : 1.0 - F.grid_dy * pr; tgy = F.grid_dx * pc_; }   // :781-788 (orig_z, orig_y)
                    const double cx = (x - 0.5) * 2.0, cy = (y - 0.5) * 2.0;                     // analytic.py:53-54
                    const double dx = (tgx - x) * 0.90, dy = (tgy - y) * 0.90;                    // :55-56, :64-66
                    act[0] = F.ep.clip_act_space ? cx : x; act[1] = F.ep.clip_act_space ? cy : y; // :803-806
                    act[2] = dx; act[3] = dy;
                } else {
                    const double *ap = F.actions + ((size_t)t_slot * F.E + e) * 4;
                    act[0] = ap[0]; act[1] = ap[1]; act[2] = ap[2]; act[3] = ap[3];
                }
            } else {
                if (eps->stop && F.resume != nullptr) {  // the slice ends between two operations of a reset
                    if (tid == 0) {
                        EpResume *rs_ = F.resume + e;
                        rs_->valid = 1; rs_->it = -1; rs_->done_partial = 0; rs_->eps = *eps;
                        if (F.resets != nullptr) {
                            ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + n_resets);
                            rs_->rr = *rr_; rr_->consumed = 2;
                        }
                    }
                    break;
                }
                for (;;) {                               // skip the stages this script does not have
                    if (with_tier2 && rp == 8) break;    // tier 2: 1500 updates before the pulls (cloth_env.py:902-903)
                    if (rp < 6) {
                        const int p_ = rp >> 1;
                        if (p_ >= s_n_pulls()) { rp = 6; continue; }
                        if (!(rp & 1) && !s_need_cov(p_)) { rp++; continue; }
                    }
                    if (rp == 6 && s_settle() <= 0) { rp = 7; continue; }
                    break;
                }
                if (with_tier2 && rp == 8) {
                    op = OP_RESET_SETTLE; do_run = true;
                    sc.n_up_end = sc.n_uprest_end = sc.n_pull_end = 0;
                    sc.n_griprest_end = sc.n_total = 1500;
                    sc.break_on_tear = 0;
                } else if (rp < 6 && !(rp & 1)) op = OP_RESET_COND;
                else if (rp < 6) {
                    op = OP_RESET_PULL; do_decode = true;
                    if (rngm) {                          // draw this pull now (cloth_env.py:851-877 tier 1, :959-972 tier 3)
                        if (tid == 0) {
                            ClothResetPull d_;
                            d_.need_coverage = 0; d_.coverage_min = 0.0;
                            if (tier == 1) {
                                d_.point = (int32_t)mt_randint(mt, (uint32_t)P);
                                d_.x = d_.y = 0.0;
                                d_.dx = mt_randval_minabs(mt, -0.20, 0.20, 0.08);
                                d_.dy = mt_randval_minabs(mt, -0.20, 0.20, 0.08);
                                d_.iters_up = F.ep.iters_up;
                            } else if (with_tier2 && tier == 2) {   // cloth_env.py:905-947: hard-coded corner points, no _prevent_oob
                                const double sd = eps->side ? 1.0 : -1.0;
                                d_.need_coverage = 2;     // bit 1: no _prevent_oob
                                d_.x = d_.y = 0.0;
                                d_.iters_up = F.ep.iters_up;
                                if ((rp >> 1) == 0) {
                                    const int ch = mt_double(mt) < 0.5 ? -25 : -1;               // :907
                                    eps->choice = ch;
                                    d_.point = P + ch;
                                    d_.dx = mt_uniform(mt, 0.30, 0.50) * sd;
                                    d_.dy = ch == -25 ? mt_uniform(mt, 0.30, 0.60) : mt_uniform(mt, -0.60, -0.30);
                                } else {
                                    const bool c25 = eps->choice == -25;
                                    d_.point = P + (c25 ? -19 : -7);
                                    d_.dx = mt_uniform(mt, 0.30, 0.60) * sd;
                                    d_.dy = c25 ? mt_uniform(mt, -0.30, -0.60) : mt_uniform(mt, 0.30, 0.60);
                                }
                            } else {
                                d_.iters_up = mt_uniform(mt, 200.0, 280.0);
                                d_.point = -1;
                                d_.x = mt_randval_minabs(mt, 0.30, 0.70, 0.0);
                                d_.y = mt_randval_minabs(mt, 0.30, 0.70, 0.0);
                                d_.dx = mt_randval_minabs(mt, -0.25, 0.25, 0.10);
                                d_.dy = mt_randval_minabs(mt, -0.25, 0.25, 0.10);
                            }
                            eps->pull = d_;
                        }
                        __syncthreads();
                    }
                    const ClothResetPull *pl = rngm ? &eps->pull : &scr->pull[rp >> 1];
                    double px_ = pl->x, py_ = pl->y;
                    const int pt_ = pl->point;
                    if (pt_ >= 0) { const Pt<T> pp = cur[pt_ < P ? pt_ : 0]; px_ = (double)pp.x; py_ = (double)pp.y; }
                    // _prevent_oob (cloth_env.py:834-840)
                    double dx0 = pl->dx, dy0 = pl->dy;
                    if (!(pl->need_coverage & 2)) {
                        if (px_ + dx0 < 0.0) dx0 = 0.0 - px_; else if (px_ + dx0 > 1.0) dx0 = 1.0 - px_;
                        if (py_ + dy0 < 0.0) dy0 = 0.0 - py_; else if (py_ + dy0 > 1.0) dy0 = 1.0 - py_;
                    }
                    // _convert_action_to_clip_space (cloth_env.py:1207-1215), delta actions
                    act[0] = F.ep.clip_act_space ? (px_ - 0.5) * 2 : px_;
                    act[1] = F.ep.clip_act_space ? (py_ - 0.5) * 2 : py_;
                    act[2] = dx0; act[3] = dy0;
                    run_iters_up = pl->iters_up;
                } else if (rp == 6) {
                    op = OP_RESET_SETTLE; do_run = true;
                    sc.n_up_end = sc.n_uprest_end = sc.n_pull_end = 0;
                    sc.n_griprest_end = sc.n_total = s_settle();
                    sc.break_on_tear = 0;
                } else {
                    op = OP_RESET_END;
                }
            }
            int n_grab = 0, iters_pull = 0, decode_err = 0;
            if (do_decode) {
                // ---- action -> schedule (cloth_env.py:396-475), in double, every thread the same arithmetic
                const ClothEpisodeParams &ep = F.ep;
                double a0 = fmax(fmin(act[0], ep.act_high[0]), ep.act_low[0]);                    // :402-415
                double a1 = fmax(fmin(act[1], ep.act_high[1]), ep.act_low[1]);
                const double c2 = fmax(fmin(act[2], ep.act_high[2]), ep.act_low[2]);
                const double c3 = fmax(fmin(act[3], ep.act_high[3]), ep.act_low[3]);
                if (ep.clip_act_space) { a0 = (a0 / 2.0) + 0.5; a1 = (a1 / 2.0) + 0.5; }          // :417-426
                const double tl = sqrt(c2 * c2 + c3 * c3);                                        // :449
                const double xd = c2 / (tl + 1e-5), yd = c3 / (tl + 1e-5);                        // :450-451
                const double xr = xd * ep.reduce_factor, yr = yd * ep.reduce_factor;              // :455-456
                const double stp = sqrt(xr * xr + yr * yr);
                double cl = 0.0;
                int ii = 0;
                for (;;) {                                                                        // :461-468
                    cl = cl + stp;
                    if (cl >= tl) break;
                    ii++;
                    if (ii >= 200000) { decode_err = 1; break; }      // non-finite action: the host wrapper raises
                }
                iters_pull = ii;
                const double iu = run_iters_up;                                                   // :472-475, left to right
                const double b1 = iu, b2 = iu + ep.iters_up_rest, b3 = iu + ep.iters_up_rest + ii;
                const double b4 = iu + ep.iters_up_rest + ii + ep.iters_grip_rest;
                const double b5 = iu + ep.iters_up_rest + ii + ep.iters_grip_rest + ep.iters_rest;
                sc.n_up_end = (int)ceil(b1); sc.n_uprest_end = (int)ceil(b2); sc.n_pull_end = (int)ceil(b3);
                sc.n_griprest_end = (int)ceil(b4); sc.n_total = (int)ceil(b5);
                sc.break_on_tear = 1;
                sc.dz_up = ep.dz_up; sc.dx_pull = xr; sc.dy_pull = yr; sc.dz_pull = 0.0;
                // ---- Gripper.grab_top (gripper.pyx:23-42) on the LDS-resident state, + force_grab (cloth_env.py:434-444)
                const T gx = (T)a0, gy = (T)a1, tt = (T)F.two_thickness;
                double radius = ep.grip_radius;
                for (int tries = 0;; tries++) {
                    const T rad = (T)radius;
                    __syncthreads();
                    if (tid == 0) { misc[8] = 0x7fffffff; misc[9] = 0; }
                    __syncthreads();
                    int best = 0x7fffffff;
                    bool incyl[PPT];
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        const int i = tid + q * NT;
                        incyl[q] = false;
                        if (i < P) {
                            const Pt<T> c = cur[i];
                            const T dx = c.x - gx, dy = c.y - gy;
                            if (dx * dx + dy * dy < rad) {                                        // gripper.pyx:35 (radius not squared)
                                incyl[q] = true;
                                for (int l = 0; l < F.n_glevels && l < best; l++) {
                                    T d = c.z - (T)F.levels[l]; d = d < 0 ? -d : d;
                                    if (d < tt) { best = l; break; }                              // gripper.pyx:36
                                }
                            }
                        }
                    }
                    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(best, o); best = v < best ? v : best; }
                    if (lane == 0 && best != 0x7fffffff) atomicMin(&misc[8], best);
                    __syncthreads();
                    best = misc[8];
                    int n = 0;
                    if (best != 0x7fffffff) {
                        const T lz = (T)F.levels[best];
#pragma unroll
                        for (int q = 0; q < PPT; q++) {
                            if (incyl[q]) {
                                const int i = tid + q * NT;
                                Pt<T> c = cur[i];
                                T d = c.z - lz; d = d < 0 ? -d : d;
                                if (d < tt) {                                                     // pinned = True ; grabbed_pts.append
                                    uint32_t w = w_cnt(c.w);
                                    if ((w & CNT_GRAB_MASK) < CNT_GRAB_MASK) w++;
                                    c.w = w_make<T>(w); cur[i] = c; n++;
                                }
                            }
                        }
                        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
                        if (lane == 0 && n) atomicAdd(&misc[9], n);
                    }
                    __syncthreads();
                    n_grab = misc[9];
                    if (n_grab > 0 || !ep.force_grab || tries >= 10000) break;
                    radius += ep.radius_inc;                                                      // cloth_env.py:439
                }
                do_run = n_grab > 0 && !decode_err;                                               // cloth_env.py:490-493
            }
            // park the plan in LDS: nothing of it stays in registers across the substep loop
            if (tid == 0) {
                eps->rp = rp; eps->op = op; eps->n_grab = n_grab; eps->iters_pull = iters_pull; eps->decode_err = decode_err;
                eps->act[0] = act[0]; eps->act[1] = act[1]; eps->act[2] = act[2]; eps->act[3] = act[3];
            }
            // uniform copies of the schedule for the loop's phase tests
            sc.n_up_end = __builtin_amdgcn_readfirstlane(sc.n_up_end);
            sc.n_uprest_end = __builtin_amdgcn_readfirstlane(sc.n_uprest_end);
            sc.n_pull_end = __builtin_amdgcn_readfirstlane(sc.n_pull_end);
            sc.n_griprest_end = __builtin_amdgcn_readfirstlane(sc.n_griprest_end);
            sc.n_total = __builtin_amdgcn_readfirstlane(do_run ? sc.n_total : 0);
            sc.break_on_tear = __builtin_amdgcn_readfirstlane(sc.break_on_tear);
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
    const int tid_outer_ = tid;
#ifdef CLOTHHIP_CELL_COUNTERS
    bool frozen_prev_ = false;
#endif
    for (int it = resumed_run ? resume_it : 0; it < sc.n_total; it++) {
        // LEAN and fp64: everything derived from the thread index (LDS addresses of the owned particles, table offsets) is formed anew in
        // every substep instead of being hoisted out of the loop and held -- or spilled -- for the whole schedule
        int tid = tid_outer_;
        if (LEAN || sizeof(T) == 8 || NT >= 512) asm volatile("" : "+v"(tid));     // (fp64: 65 -> 0 spilled registers; 50x50: +3 %)
        const int lane = tid & 63;
        KArgsC<T> *Ak_ = (KArgsC<T> *)__builtin_amdgcn_kernarg_segment_ptr();

        // ---- ClothEnv._pull (cloth_env.py:352-367): adjust / nothing / release -------------------
        int mode = 0; T ax = 0, ay = 0, az = 0;
        if (it < sc.n_up_end) { mode = 1; az = dz_up; }
        else if (it < sc.n_uprest_end) { }
        else if (it < sc.n_pull_end) { mode = 1; ax = dxp; ay = dyp; az = dzp; }
        else if (it < sc.n_griprest_end) { }
        else mode = 2;
        if (mode == 1) {
            Pt<T> cq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) cq[q] = cur[tid + q * NT < P ? tid + q * NT : 0];     // batched: one LDS latency, not PPT
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                Pt<T> c = cq[q];
                const int m = (int)(w_cnt(c.w) & CNT_GRAB_MASK);
                if (m) {
                    for (int r = 0; r < m; r++) {       // gripper.pyx:60-66: p <- x ; x <- delta + x
                        pvx[q] = c.x; pvy[q] = c.y; pvz[q] = c.z;
                        c.x = ax + c.x; c.y = ay + c.y; c.z = az + c.z;
                    }
                    cur[i] = c;
                }
            }
            __syncthreads();
        } else if (mode == 2 && it == sc.n_griprest_end) {      // release() is idempotent: only its first call acts
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const uint32_t c = w_cnt(cur[i].w);
                if (c & CNT_GRAB_MASK) cur[i].w = w_make<T>(0u);    // gripper.pyx:68-73
            }
            __syncthreads();
        }

        TSTAMP(0)
        // ---- gravity + Hooke gather + Verlet (cloth.pyx:216-256) ----------------------------------
        if (pm & PH_HOOKE) {
            CLOTH_PHASE_ARGS()
            // Per particle: f = (0,0,m*g) + sum over its incident springs in ascending list index of fm * (nbr - self).
            // (For the spring's ptB the reference adds -(fm * (self - nbr)), which is the same IEEE value.)
            // Branch-free: absent slots (grid border) and pinned particles are computed and discarded.
            T nx[PPT], ny[PPT], nz[PPT];
            uint32_t wme[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                nx[q] = ny[q] = nz[q] = (T)0; wme[q] = 1u;
                // a real branch per particle: each particle's 12 springs form their own scheduling region, which
                // keeps the register allocator from interleaving all PPT*12 spring evaluations at once
                if (tid + q * NT < P) {
                    const Pt<T> me = cur[tid + q * NT];
                    wme[q] = w_cnt(me.w);
                    T fx = (T)0 + (T)0, fy = (T)0 + (T)0, fz = (T)0 + k.mg;
                    uint32_t gl[HK_SLOTS];
                    int iq_ = tid + q * NT; uint32_t vq_ = vm[LEAN ? q : 0];
                    if (LEAN) asm volatile("" : "+v"(iq_), "+v"(vq_));     // opaque: the stencil is recomputed every substep, not hoisted and held
#pragma unroll
                    for (int sl = 0; sl < HK_SLOTS; sl++)
                        gl[sl] = LEAN ? lean_entry(iq_, vq_, sl) : (GT_REG ? gt[GT_REG ? q : 0][sl] : Ak_->gather[sl * Ppad + tid + q * NT]);
                    // software pipeline: the neighbour records of the next springs are in flight while spring sl is
                    // evaluated (left to itself the scheduler, which minimises live registers at this kernel's pressure, issues
                    // each 16-byte read right before its use and waits out the whole LDS latency 12 times per particle)
                    constexpr int HK_AHEAD = 2;
                    Pt<T> nbq[HK_AHEAD];
#pragma unroll
                    for (int sl = 0; sl < HK_AHEAD; sl++) {
                        uint32_t g = gl[sl];
                        asm volatile("" : "+v"(g));         // opaque: keeps the address math inside the substep loop
                        gl[sl] = g;
                        nbq[sl] = cur[g & HK_NBR_MASK];
                    }
#pragma unroll
                    for (int sl = 0; sl < HK_SLOTS; sl++) {
                        const uint32_t g = gl[sl];
                        const Pt<T> nb = nbq[sl % HK_AHEAD];
                        if (sl + HK_AHEAD < HK_SLOTS) {
                            uint32_t gn = gl[sl + HK_AHEAD];
                            asm volatile("" : "+v"(gn));
                            gl[sl + HK_AHEAD] = gn;
                            nbq[sl % HK_AHEAD] = cur[gn & HK_NBR_MASK];
                        }
                        __builtin_amdgcn_sched_barrier(0);  // the reads above stay above the arithmetic below
                        const T r = LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK));
                        const T kk = (LEAN ? lean_bend(sl) : (g & HK_BEND) != 0u) ? k.ks_bend : k.ks_str;
                        const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;
                        const T l = fastnorm<T>(dx, dy, dz);                                      // :231
                        const T fm = dev_div<T>(kk * (l - r), l);                                 // :232
                        const bool valid = (g & HK_VALID) != 0u;
                        fx = valid ? mad<T>(fm, dx, fx) : fx; fy = valid ? mad<T>(fm, dy, fy) : fy; fz = valid ? mad<T>(fm, dz, fz) : fz;   // :236-237
                    }
                    nx[q] = mad<T>(fx, k.dsm, mad<T>(k.damp, me.x - pvx[q], me.x));               // :249
                    ny[q] = mad<T>(fy, k.dsm, mad<T>(k.damp, me.y - pvy[q], me.y));
                    nz[q] = mad<T>(fz, k.dsm, mad<T>(k.damp, me.z - pvz[q], me.z));
                    if (wme[q] == 0) { pvx[q] = me.x; pvy[q] = me.y; pvz[q] = me.z; }             // :256
                }
            }
            __syncthreads();                                // every neighbour read of the old positions is done
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (wme[q]) continue;                       // pinned (or no particle): Verlet skips it (cloth.pyx:244)
                cur[i] = Pt<T>{nx[q], ny[q], nz[q], w_make<T>(0u)};                               // :255
            }
        }

        TSTAMP(1)
        // ---- spatial map (cloth.pyx:298-311): hash table in LDS keyed by the exact cell key + a list of the occupied
        // slots; members of a cell are stored contiguously (CSR); ascending point index is restored by the sweep.
        if (pm & PH_COLLIDE) {
            CLOTH_PHASE_ARGS()
            uint32_t ch[PPT], rank[PPT];
            {
                uint32_t ckey[PPT];
                bool pend[PPT], made[PPT];
                bool anyp = false;
#pragma unroll
                for (int q = 0; q < PPT; q++) {             // batched: the PPT particles' LDS traffic overlaps
                    const int i = tid + q * NT;
                    const Pt<T> c = cur[i < P ? i : 0];                                           // own slot: no hazard
                    ckey[q] = cell_key<T>(k, c.x, c.y, c.z);
#ifdef CLOTHHIP_CELL_COUNTERS
                    if (i < P) {      // census: did any particle change its cell since the previous substep?
                        uint32_t *lk_ = reinterpret_cast<uint32_t *>(smem + lay.lkey);
                        if (lk_[i] != ckey[q]) atomicOr(&misc[16], 1);
                        lk_[i] = ckey[q];
                    }
#endif
                    // (ht_bits 0: a table whose size is not a power of two -- the two-per-CU layout of the large grids -- is indexed by the
                    //  high half of hash x size; which slot a cell gets never shows in the results)
                    ch[q] = Ak_->ht_bits ? (ckey[q] * 2654435761u) >> (32 - Ak_->ht_bits) : __umulhi(ckey[q] * 2654435761u, (uint32_t)HT);
                    pend[q] = i < P; made[q] = false; anyp |= pend[q];
                }
                // linear probing; the table has >= 1.5 P slots, so a free one always exists -- the probe bound only
                // guarantees termination should LDS ever be corrupted
                for (int probe = 0; anyp && probe < HT; probe++) {
                    anyp = false;
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        if (pend[q]) {
                            const uint32_t old = atomicCAS(&hkey[ch[q]], KEY_EMPTY, ckey[q]);
                            if (old == KEY_EMPTY || old == ckey[q]) { pend[q] = false; made[q] = old == KEY_EMPTY; }
                            else { ch[q] = ch[q] + 1u >= (uint32_t)HT ? 0u : ch[q] + 1u; anyp = true; }
                        }
                    }
                }
                int nmade = 0;
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    rank[q] = 0;
                    if (i < P) { slot[i] = (uint16_t)ch[q]; rank[q] = atomicAdd(&hco[ch[q]], 1u); }   // my place in the cell
                    nmade += made[q] ? 1 : 0;
                }
                // whoever created a slot lists it: one LDS atomic per wave
                const int inc = wave_incl_scan(nmade);
                const int tot = __builtin_amdgcn_readlane(inc, 63);
                if (tot) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&misc[3], tot);
                    int o = __builtin_amdgcn_readfirstlane(base) + inc - nmade;
#pragma unroll
                    for (int q = 0; q < PPT; q++)
                        if (made[q]) olist[o++] = (uint16_t)ch[q];
                }
            }
            __syncthreads();
            TSTAMP(2)
            const int nocc = __builtin_amdgcn_readfirstlane(misc[3]);
            for (int t0 = 0; t0 < nocc; t0 += NT) {         // member range of every occupied cell (any order)
                const int t = t0 + tid;
                const int h = t < nocc ? (int)olist[t] : 0;
                const int c = t < nocc ? (int)hco[h] : 0;
                const int inc = wave_incl_scan(c);
                int base = 0;
                if (lane == 63) base = atomicAdd(&misc[4], inc);
                base = __builtin_amdgcn_readlane(base, 63);
                if (t < nocc) hco[h] = ((uint32_t)(base + inc - c) << 16) | (uint32_t)c;          // (start << 16) | count
            }
            __syncthreads();
            TSTAMP(3)
            int cn[PPT], cstart[PPT];
            Pt<T> cme[PPT];
            int nmax = 0;
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                const uint32_t co = hco[ch[q]];
                cme[q] = cur[i < P ? i : 0];
                cstart[q] = (int)(co >> 16);
                if (i < P) {
                    memb[cstart[q] + (int)rank[q]] = (uint16_t)i;
                    if (Ak_->cell_copy) cpos[cstart[q] + (int)rank[q]] = Pt<T>{cme[q].x, cme[q].y, cme[q].z, w_make<T>((uint32_t)i)};
                }
                const bool use = i < P && w_cnt(cme[q].w) == 0;
                cn[q] = use ? (int)(co & 0xFFFFu) : 0;
                if (cn[q] < 2) cn[q] = 0;
                nmax = cn[q] > nmax ? cn[q] : nmax;
            }
            __syncthreads();
            TSTAMP(4)
            if constexpr (RELAXED) {
                // Jacobi order (NOT the reference's Gauss-Seidel order, cloth.pyx:313-343): every unpinned particle collects its hits
                // against the cell-ordered snapshot of the phase's start positions and moves by itself; no seeds, no cell sweeps
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    if (cn[q] >= 2) {
                        const Pt<T> me_ = cme[q];
                        T tx = (T)0, ty = (T)0, tz = (T)0; int nh = 0;
#pragma unroll 1
                        for (int b = 0; b < cn[q]; b++) {
                            const Pt<T> o = cpos[cstart[q] + b];
                            const T dx = me_.x - o.x, dy = me_.y - o.y, dz = me_.z - o.z;
                            const T dist = dev_sqrt<T>(sumsq<T>(dx, dy, dz));
                            const bool hit_ = ((int)w_cnt(o.w) != i) & (dist <= k.thresh);
                            const T factor = hit_ ? dev_div<T>(k.thresh - dist, dist) : (T)0;
                            tx = hit_ ? mad<T>(dx, factor, tx) : tx; ty = hit_ ? mad<T>(dy, factor, ty) : ty; tz = hit_ ? mad<T>(dz, factor, tz) : tz;
                            nh += hit_ ? 1 : 0;
                        }
                        if (nh) {
                            const T nf = (T)nh;
                            cur[i] = Pt<T>{me_.x + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps), me_.y + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps),
                                           me_.z + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps), me_.w};
                        }
                    }
                }
            } else {
            // ---- self-collision (cloth.pyx:313-343) ------------------------------------------------------
            // (1) seeds: every unpinned particle, in parallel: does it have a hit (a same-cell member within
            //     2*thickness) at the CURRENT positions? A seed gets the flag bit of its slot word; the first seed of
            //     a cell (whoever wins) puts the cell on the active list. Conservative by the filter slack; the sweep
            //     re-tests exactly. With the cell-ordered record copy a pair costs ONE LDS read.
            {
                const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
                bool hit[PPT];
#pragma unroll
                for (int q = 0; q < PPT; q++) hit[q] = false;
                // (left to itself the compiler unrolls the member loops several times: fine at 256 VGPRs, 500 spilled registers at
                //  the LEAN variant's 168 -- that variant gets its own copy of the loops, not unrolled)
#ifndef CLOTHHIP_BISECT_NOPRECHECK
                if constexpr (LEAN || (NT == 512 && PPT == 2)) {
                    // register-lean form (the builds with a VGPR cap: LEAN, eight waves per cloth): one owned particle after the other (a real branch each: a scheduling region of its own),
                    // the member loop not unrolled; the trip count is the wave's largest member count for THAT particle
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        const int nq = -__builtin_amdgcn_readlane(wave_incl_min(-cn[q]), 63);
                        if (nq > 0) {
                            int iq_ = tid + q * NT;
                            asm volatile("" : "+v"(iq_));
                            const Pt<T> me_ = cur[iq_ < P ? iq_ : 0];
                            const int cs_ = cstart[q], cn_ = cn[q];
                            bool h_ = false;
                            if (Ak_->cell_copy) {
                                // a read past the cell's range (another cell's record or the padding behind the array) is masked out
                                // by the member count; the trip base is clamped so that no read leaves the padded array
#ifndef CLOTHHIP_PRECHECK2              // (round 5: four members per trip -- half the loop branches and LDS waits per member: +0.25 %; -DCLOTHHIP_PRECHECK2: two)
#pragma unroll 1
                                for (int b = 0; b < nq; b += 4) {
                                    const int base = cs_ + b < Ppad + 28 ? cs_ + b : Ppad + 28;
                                    const Pt<T> o0 = cpos[base], o1 = cpos[base + 1], o2 = cpos[base + 2], o3 = cpos[base + 3];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    const T dx2 = me_.x - o2.x, dy2 = me_.y - o2.y, dz2 = me_.z - o2.z;
                                    const T dx3 = me_.x - o3.x, dy3 = me_.y - o3.y, dz3 = me_.z - o3.z;
                                    h_ |= (b < cn_) & ((int)w_cnt(o0.w) != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);
                                    h_ |= (b + 1 < cn_) & ((int)w_cnt(o1.w) != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                    h_ |= (b + 2 < cn_) & ((int)w_cnt(o2.w) != iq_) & !(sumsq<T>(dx2, dy2, dz2) > thr2);
                                    h_ |= (b + 3 < cn_) & ((int)w_cnt(o3.w) != iq_) & !(sumsq<T>(dx3, dy3, dz3) > thr2);
                                }
#else
#pragma unroll 1
                                for (int b = 0; b < nq; b += 2) {
                                    const int base = cs_ + b < Ppad + 30 ? cs_ + b : Ppad + 30;
                                    const Pt<T> o0 = cpos[base], o1 = cpos[base + 1];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    h_ |= (b < cn_) & ((int)w_cnt(o0.w) != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);       // branch-free on purpose (& not &&)
                                    h_ |= (b + 1 < cn_) & ((int)w_cnt(o1.w) != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                }
#endif
                            } else {
#pragma unroll 1
                                for (int b = 0; b < nq; b += 2) {
                                    const int j0 = (int)memb[cn_ ? cs_ + (b < cn_ ? b : 0) : 0], j1 = (int)memb[cn_ ? cs_ + (b + 1 < cn_ ? b + 1 : 0) : 0];
                                    const Pt<T> o0 = cur[j0], o1 = cur[j1];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    h_ |= (b < cn_) & (j0 != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);
                                    h_ |= (b + 1 < cn_) & (j1 != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                }
                            }
                            hit[q] = h_;
                        }
                    }
                } else {
                    if (Ak_->cell_copy) {
                        constexpr int CU = 2;
                        // a read past the cell's range (another cell's record or the padding behind the array) is masked out
                        // by the member count; the trip base is clamped so that no read leaves the padded array
                        for (int b = 0; b < nmax; b += CU) {     // CU members x PPT particles per trip: their LDS reads overlap
                            Pt<T> o[PPT][CU];
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int base = cstart[q] + b < Ppad + 32 - CU ? cstart[q] + b : Ppad + 32 - CU;
#pragma unroll
                                for (int u = 0; u < CU; u++) o[q][u] = cpos[base + u];
                            }
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < CU; u++) {                              // branch-free on purpose (& not &&)
                                    const T dx = cme[q].x - o[q][u].x, dy = cme[q].y - o[q][u].y, dz = cme[q].z - o[q][u].z;
                                    const bool other = (b + u < cn[q]) & ((int)w_cnt(o[q][u].w) != tid + q * NT);
                                    hit[q] |= other & !(sumsq<T>(dx, dy, dz) > thr2);
                                }
                        }
                    } else {
                        for (int b = 0; b < nmax; b += 4) {
                            int jj[PPT][4];
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const int bb = b + u < cn[q] ? b + u : 0;
                                    jj[q][u] = (int)memb[cn[q] ? cstart[q] + bb : 0];
                                }
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const Pt<T> o = cur[jj[q][u]];
                                    const T dx = cme[q].x - o.x, dy = cme[q].y - o.y, dz = cme[q].z - o.z;
                                    const bool other = (b + u < cn[q]) & (jj[q][u] != tid + q * NT);
                                    hit[q] |= other & !(sumsq<T>(dx, dy, dz) > thr2);
                                }
                        }
                    }
                }
#endif
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    if (hit[q]) {
                        slot[tid + q * NT] = (uint16_t)(ch[q] | 0x8000u);
                        if (atomicMin(&hkey[ch[q]], (uint32_t)(tid + q * NT)) >= KEY_FLOOR)
                            alist_end[-atomicAdd(&misc[2], 1)] = (uint16_t)ch[q];
                    }
                }
            }
            __syncthreads();
            TSTAMP(5)
            // (2) the active cells (those with a seed): exact Gauss-Seidel sweep (cells are independent: each particle
            // sits in exactly one). Every wave reads the whole list; work is handed out by LDS tickets so that the waves
            // finish together: first the cells with more than 16 members, one per wave at a time, then the small
            // cells four at a time, one per 16-lane group (two larger cells per wave in 32-lane groups was measured:
            // the bpermute broadcasts cost what the pairing saves).
            {
                const int na = __builtin_amdgcn_readfirstlane(misc[2]);
#ifdef CLOTHHIP_CELL_COUNTERS
                tph[9] += 64 * na; tph[10] += 64 * nocc; tph[0] += na == 0 ? 64 : 0;
                tph[7] += 64 * (-__builtin_amdgcn_readlane(wave_incl_min(-nmax), 63));
#endif
                if (na) __builtin_amdgcn_s_setprio(2);     // serial per-cell sweeps: latency-critical like the strain sweep
                int tkb = -1, tks = -1, bbase = 0, sbase = 0;       // outstanding tickets, tickets used up by earlier chunks
                for (int c0 = 0; c0 < na; c0 += 64) {
                    const int ei = c0 + lane;
                    const bool ev = ei < na;
                    const int hs_l = ev ? (int)alist_end[-ei] : 0;
                    const uint32_t co_l = ev ? hco[hs_l] : 0u;
                    const int n_l = (int)(co_l & 0xFFFFu);
                    unsigned long long big = ballot64(ev && n_l > 16);
                    unsigned long long sm = ballot64(ev && n_l <= 16);
                    const int nbig = (int)__popcll(big), nsb = ((int)__popcll(sm) + 3) >> 2;
                    for (int used = 0;;) {
                        if (tkb < 0) { int t = 0; if (lane == 0) t = atomicAdd(&misc[5], 1); tkb = __builtin_amdgcn_readfirstlane(t); }
                        if (tkb >= bbase + nbig) break;             // that ticket is for a later chunk (or nothing)
                        for (; used < tkb - bbase; used++) big &= big - 1ull;
                        tkb = -1;
                        const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)big) - 1);
                        const uint32_t co = (uint32_t)__builtin_amdgcn_readlane((int)co_l, b);
                        const int n = (int)(co & 0xFFFFu);
                        uint16_t *m = memb + (int)(co >> 16);
                        if (n <= 64) {
#ifndef CLOTHHIP_BISECT_NOWAVE
                            const int nv_ = collide_cell_wave<T>(cur, m, slot, n, k, lane);
#else
                            const int nv_ = 0;
#endif
#ifdef CLOTHHIP_CELL_COUNTERS
                            tph[4] += 64; tph[5] += 64 * n; tph[6] += 64 * (nv_ & 0xffff); tph[11] += 64 * (nv_ >> 16);
#else
                            (void)nv_;
#endif
                        } else if (lane == 0) collide_cell_serial<T>(cur, m, n, k);
                    }
                    bbase += nbig;
#ifdef CLOTHHIP_CELL_STAMPS
                    TSTAMP(10)
#endif
                    for (int used = 0;;) {                        // up to four small cells per ticket
                        if (tks < 0) { int t = 0; if (lane == 0) t = atomicAdd(&misc[6], 1); tks = __builtin_amdgcn_readfirstlane(t); }
                        if (tks >= sbase + nsb) break;
                        for (; used < 4 * (tks - sbase); used++) sm &= sm - 1ull;
                        tks = -1;
                        int hs = -1;
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            if (sm) {
                                const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)sm) - 1);
                                sm &= sm - 1ull;
                                const int v = __builtin_amdgcn_readlane(hs_l, b);
                                hs = (lane >> 4) == g ? v : hs;
                            }
                        }
                        used += 4;
#ifdef CLOTHHIP_CELL_COUNTERS
                        tph[8] += 64;
#endif
#ifndef CLOTHHIP_BISECT_NOGROUP
                        collide_cells_group<T, 16>(cur, memb, slot, hco, hs, k, lane);
#endif
                    }
                    sbase += nsb;
#ifdef CLOTHHIP_CELL_STAMPS
                    TSTAMP(11)
#endif
                }
            }
            __builtin_amdgcn_s_setprio(0);
            }   // (exact order)
            __syncthreads();
            TSTAMP(6)
            for (int t = tid; t < nocc; t += NT) { const int h = (int)olist[t]; hkey[h] = KEY_EMPTY; hco[h] = 0; }   // ready for the next substep
            if (tid == 0) { misc[2] = 0; misc[3] = 0; misc[4] = 0; misc[5] = 0; misc[6] = 0; }
        } else {
            __syncthreads();
        }

        // ---- plane (cloth.pyx:345-370), by the owner (it holds the previous position) --------------------
        if (pm & PH_PLANE) {
            CLOTH_PHASE_ARGS()
            const T k_min_z = k.min_z, k_surf_off = k.surf_off, k_one_m_fric = k.one_m_fric;
            Pt<T> mq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) mq[q] = cur[tid + q * NT < P ? tid + q * NT : 0];     // batched: one LDS latency, not PPT
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const Pt<T> me = mq[q];
#ifdef CLOTHHIP_CELL_COUNTERS
                if (!w_cnt(me.w) && me.z >= k_min_z) atomicOr(&misc[12], 1);     // census: an unpinned particle the plane did not restore
#endif
                if (w_cnt(me.w) || me.z >= k_min_z) continue;
                const T px = pvx[q], py = pvy[q], pz = pvz[q];
                const T t = (k_min_z - pz) * (T)1.0;
                const T tgx = px + t * (T)(-0.0), tgy = py + t * (T)(-0.0), tgz = pz + t * (T)(-1.0);
                const T gx = tgx + k_surf_off * (T)0.0, gy = tgy + k_surf_off * (T)0.0, gz = tgz + k_surf_off * (T)1.0;
                const T cx = gx - px, cy = gy - py, cz = gz - pz;
                cur[i] = Pt<T>{mad<T>(cx, k_one_m_fric, px), mad<T>(cy, k_one_m_fric, py), mad<T>(cz, k_one_m_fric, pz), me.w};
            }
        }
        if (FUSED && sliced) {             // thread 0 looks at the clock here, between two barriers that every thread passes in
                                           // every substep; everyone reads its verdict at the end of the substep
            if (tid == 0) misc[7] = (__builtin_amdgcn_s_memrealtime() - eps->t_launch >= Fp->budget_ticks) ? 1 : 0;
        }
        __syncthreads();

        TSTAMP(7)
        // ---- strain limit + tear (cloth.pyx:258-296) ---------------------------------------------------
        // (1) all threads: which springs would stretch/tear at the CURRENT positions? Only the first and the last of them
        //     (in window-table order) are kept: a spring untouched by earlier corrections of the sweep behaves exactly as
        //     evaluated here, so nothing before the first needs a look, and nothing behind the last unless a correction
        //     reaches it. No spring flagged: the sweep is skipped (a cloth at rest).
        // (2) wave 0 walks the windows in between (strain_sweep above).
        if constexpr (RELAXED) {
            // Coloured order (NOT the reference's list order, cloth.pyx:258-296): the six springs a particle owns (to r-1, c-1, the two
            // diagonals, r-2, c-2) in two parity classes each -- twelve classes whose springs share no particle --, one class after the
            // other, every class in parallel by the owners of its springs. Same test, same correction per spring.
            CLOTH_PHASE_ARGS()
            static_assert(!RELAXED || LEAN, "the relaxed-order companion exists for the LEAN arithmetic (stencil from the grid position, palette rest lengths)");
            int tear_ = 0;
#pragma unroll 1
            for (int col = 0; col < 12; col++) {
                const int kind = col >> 1, par = col & 1;
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    const int r_ = (int)(rc[RELAXED ? q : 0] & 0xFFu), c_ = (int)(rc[RELAXED ? q : 0] >> 8);
                    const int key = (kind == 1 || kind == 5) ? c_ : r_;
                    const bool on = i < P && ((vm[LEAN ? q : 0] >> kind) & 1u) && (((kind >= 4 ? key >> 1 : key) & 1) == par);
                    if (on) {
                        const int j = i + (kind == 0 ? -Ak_->N : kind == 1 ? -1 : kind == 2 ? -Ak_->N - 1 : kind == 3 ? -Ak_->N + 1 : kind == 4 ? -2 * Ak_->N : -2);
                        const Pt<T> a_ = cur[j], b_ = cur[i];                                  // ptA (the earlier point), ptB (the owner)
                        const uint32_t ca = w_cnt(a_.w), cb = w_cnt(b_.w);
                        const T rest = kind >= 4 ? Ak_->pal_bend : (kind >= 2 ? Ak_->pal_shear : Ak_->pal_struct);
                        const T dx = a_.x - b_.x, dy = a_.y - b_.y, dz = a_.z - b_.z;
                        const T len = fastnorm<T>(dx, dy, dz);
                        if (!((ca != 0) & (cb != 0))) {
                            if (len > rest * k.tear_thresh) tear_ = 1;
                            const T t11 = rest * k.c11;
                            if (len > t11) {
                                const T ux = dev_div<T>(dx, len), uy = dev_div<T>(dy, len), uz = dev_div<T>(dz, len);
                                const T extra = len - t11;
                                const T wa = ca != 0 ? (T)0 : (cb != 0 ? (T)1 : (T)0.5), wb = cb != 0 ? (T)0 : (ca != 0 ? (T)1 : (T)0.5);
                                const T ea = extra * wa, eb = extra * wb;
                                if (ca == 0) cur[j] = Pt<T>{mad<T>(-ux, ea, a_.x), mad<T>(-uy, ea, a_.y), mad<T>(-uz, ea, a_.z), a_.w};
                                if (cb == 0) cur[i] = Pt<T>{mad<T>(ux, eb, b_.x), mad<T>(uy, eb, b_.y), mad<T>(uz, eb, b_.z), b_.w};
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (__any(tear_) && lane == 0) misc[0] = 1;
            __syncthreads();
        } else
        if (pm & PH_STRAIN) {
            CLOTH_PHASE_ARGS()
            {
                // Every spring is tested once, by the owner of its ptB (the particle the reference appended it for):
                // the owner holds the spring's gather entry (neighbour = ptA, table slot) and, with
                // REST_REG, its rest length in registers, so the pre-pass needs one 16-byte LDS read per spring.
                int nact = 0, pmin = 0x7fffffff, pmax = -1;     // flagged springs; the first / last of them in table order
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    if (tid + q * NT < P) {
                        const Pt<T> me = cur[tid + q * NT];
                        const uint32_t cme_ = w_cnt(me.w);
                        uint32_t gl[HK_SLOTS / 2];
                        int iq_ = tid + q * NT; uint32_t vq_ = vm[LEAN ? q : 0];
                        if (LEAN) asm volatile("" : "+v"(iq_), "+v"(vq_));
#pragma unroll
                        for (int sl = 0; sl < HK_SLOTS / 2; sl++)
                            gl[sl] = LEAN ? lean_entry(iq_, vq_, sl) : (GT_REG ? gt[GT_REG ? q : 0][sl] : Ak_->gather[sl * Ppad + tid + q * NT]);
                        // software pipeline, as in the Hooke phase: two neighbour reads in flight ahead of the test
                        constexpr int PP_AHEAD = 2;
                        Pt<T> nbq[PP_AHEAD];
#pragma unroll
                        for (int sl = 0; sl < PP_AHEAD; sl++) {
                            uint32_t g = gl[sl];
                            asm volatile("" : "+v"(g));
                            gl[sl] = g;
                            nbq[sl] = cur[g & HK_NBR_MASK];
                        }
                        // (a) branch-free: which of the six springs come within the slack band of their limit at all?
                        uint32_t cand = 0u;
                        T l2s[HK_SLOTS / 2];
#pragma unroll
                        for (int sl = 0; sl < HK_SLOTS / 2; sl++) {       // own springs come first in ascending list order
                            const uint32_t g = gl[sl];
                            const Pt<T> nb = nbq[sl % PP_AHEAD];
                            if (sl + PP_AHEAD < HK_SLOTS / 2) {
                                uint32_t gn = gl[sl + PP_AHEAD];
                                asm volatile("" : "+v"(gn));
                                gl[sl + PP_AHEAD] = gn;
                                nbq[sl % PP_AHEAD] = cur[gn & HK_NBR_MASK];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            T r = LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK));
                            asm volatile("" : "+v"(r));     // or the thresholds below are hoisted out of the substep loop
                                                            // for all 18 springs and live in scratch
                            const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;   // (ptA - ptB), as :270
                            const T len2 = sumsq<T>(dx, dy, dz);
                            l2s[sl] = len2;
                            const T t11 = r * k.c11, tt = r * k.tear_thresh;
                            const T tmin = t11 < tt ? t11 : tt;
                            const bool pre = ((g & (HK_VALID | HK_ASB)) == (HK_VALID | HK_ASB)) &
                                             !((cme_ != 0) & (w_cnt(nb.w) != 0)) &
                                             (len2 > tmin * tmin * ((T)1 - filt_slack<T>()));
                            cand |= pre ? (1u << sl) : 0u;
                        }
                        // (b) those few: inside the slack band around the limit the sweep's exact test (:270-275) decides: a
                        // spring that sits exactly ON its limit (left there by an earlier substep's correction) is then not
                        // flagged, and a cloth at rest skips the sweep altogether
                        if (cand) {
#pragma unroll
                            for (int sl = 0; sl < HK_SLOTS / 2; sl++) {
                                if (cand & (1u << sl)) {
                                    // (LEAN: the spring's table slot is read from the gather table only now that it is needed: the table
                                    //  is compacted, the sl-th stencil position is the particle's popcount(valid below sl)-th entry)
                                    const uint32_t pos_ = TAB == 2 ? (uint32_t)pslot[sl * Ppad + iq_] : ((LEAN ? Ak_->gather[__popc(vq_ & ((1u << sl) - 1u)) * Ppad + iq_] : gl[sl])      // (the opaque copies: nothing of this is hoisted out of the substep loop and held)
                                                           >> HK_POS_SHIFT) & HK_POS_MASK;
                                    T r = LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at(pos_));
                                    asm volatile("" : "+v"(r));
                                    const T len2 = l2s[sl];
                                    const T t11 = r * k.c11, tt = r * k.tear_thresh;
                                    const T tmin = t11 < tt ? t11 : tt;
                                    bool flag = len2 > tmin * tmin * ((T)1 + filt_slack<T>());
                                    if (!flag) { const T len = dev_sqrt<T>(len2); flag = len > t11 || len > tt; }
                                    if (flag) { nact++; pmin = (int)pos_ < pmin ? (int)pos_ : pmin; pmax = (int)pos_ > pmax ? (int)pos_ : pmax; }
#ifdef CLOTHHIP_CELL_COUNTERS
                                    if (flag) atomicOr(&misc[13 + (((int)pos_ >> 6) >> 5 & 1)], 1 << (((int)pos_ >> 6) & 31));
#endif
                                }
                            }
                        }
                    }
                }
                if (__any(nact)) {
                    const int min_ = __builtin_amdgcn_readlane(wave_incl_min(pmin), 63);         // DPP: no LDS round trips
                    const int max_ = -__builtin_amdgcn_readlane(wave_incl_min(-pmax), 63);
                    if (lane == 0) { misc[1] = 1; atomicMin(&misc[10], min_); atomicMax(&misc[11], max_); }
                }
            }
            __syncthreads();
            TSTAMP(8)
#ifdef CLOTHHIP_CELL_COUNTERS
            int swept_ = 0;
#endif
            if (SWEEP_MW) {
                // every wave of the cloth takes part (strain_sweep_mw): the decision and the walk's bounds are read by all of them
                // before the sweep's first barrier and reset by wave 0 behind its last
                if (misc[1] || (pm & PH_NOSKIP)) {
                    const bool all_ = (pm & PH_NOSKIP) != 0;
                    const int w0 = __builtin_amdgcn_readfirstlane(all_ ? 0 : (misc[10] >> 6));
                    const int w1 = __builtin_amdgcn_readfirstlane(all_ ? Ak_->nW - 1 : (misc[11] >> 6));
                    const bool tic = __builtin_amdgcn_readfirstlane(!(k.tear_thresh < k.c11) ? 1 : 0) != 0;
                    const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
                    const int wl_ = (Ak_->Spad >> 6) - 1;                   // the table's last (padding, empty) window
                    int *const sw_ = misc + 24, *const st_ = misc + 20;
#ifdef CLOTHHIP_MW_PRIO
                    __builtin_amdgcn_s_setprio(CLOTHHIP_MW_PRIO);
#endif
                    const int tear = tic ? strain_sweep_mw<T, v_ldstab(TAB), NT / 64, SWEEP_STATS, true>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, wl_, Ak_->wt_rshift, k, lane, wave_, sw_, st_)
                                         : strain_sweep_mw<T, v_ldstab(TAB), NT / 64, SWEEP_STATS, false>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, wl_, Ak_->wt_rshift, k, lane, wave_, sw_, st_);
#ifdef CLOTHHIP_MW_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    if (__any(tear) && lane == 0) misc[0] = 1;
#ifdef CLOTHHIP_CELL_COUNTERS
                    swept_ = 1;
#endif
                    if (tid == 0) {
                        misc[15]++; misc[1] = 0; misc[10] = 0x7fffffff; misc[11] = -1;
                        if (SWEEP_STATS) {
                            st_passes += st_[0]; st_commits += st_[1]; st_windows += st_[2];
#ifdef CLOTHHIP_MW_ROUNDS
                            st_commits += st_[3] - st_[1]; st_[3] = 0;
#endif
                            st_[0] = 0; st_[1] = 0; st_[2] = 0;
                        }
                    }
                }
            } else
            if (tid < 64 && (misc[1] || (pm & PH_NOSKIP))) {
                __builtin_amdgcn_s_setprio(3);            // the serial sweep is the critical path of the whole cloth
                const bool all_ = (pm & PH_NOSKIP) != 0;
                const int w0 = __builtin_amdgcn_readfirstlane(all_ ? 0 : (misc[10] >> 6));
                const int w1 = __builtin_amdgcn_readfirstlane(all_ ? Ak_->nW - 1 : (misc[11] >> 6));
                if (lane == 0) misc[15]++;               // sweeps run (clothhip_debug_stats): in LDS -- as a register it was spilled, reloaded and stored by every sweep
                // tear_thresh >= 1.1 (every shipped configuration): only a stretching spring can tear, the test sits in the commit
                const bool tic = __builtin_amdgcn_readfirstlane(!(k.tear_thresh < k.c11) ? 1 : 0) != 0;
#ifdef CLOTHHIP_CELL_COUNTERS
                const unsigned long long fmask_ = (unsigned long long)(uint32_t)misc[13] | ((unsigned long long)(uint32_t)misc[14] << 32);
                swept_ = 1;
#else
                const unsigned long long fmask_ = 0ull;
#endif
                const int tear = tic ? (SWEEP_LEAN ? strain_sweep_lean<T, v_ldstab(TAB), SWEEP_STATS>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, Ak_->wt_rshift, k,
                                                                                                  lane, st_windows, st_passes, st_commits)
                                                   : strain_sweep<T, v_ldstab(TAB), SWEEP_TIMED, SWEEP_STATS, true>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, Ak_->nW, Ak_->wt_rshift, k,
                                                                                                 lane, st_windows, st_passes, st_commits, tph, fmask_))
                                     : strain_sweep<T, v_ldstab(TAB), SWEEP_TIMED, SWEEP_STATS, false>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, Ak_->nW, Ak_->wt_rshift, k,
                                                                                                  lane, st_windows, st_passes, st_commits, tph, fmask_);
                if (__any(tear) && lane == 0) misc[0] = 1;
                if (lane == 0) { misc[1] = 0; misc[10] = 0x7fffffff; misc[11] = -1; }
#ifdef CLOTHHIP_CELL_COUNTERS
                if (lane == 0) { misc[13] = 0; misc[14] = 0; }
#endif
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
#ifndef CLOTHHIP_SWEEP_STAMPS          // (that build uses slots 9-11 for the sweep's passes)
            TSTAMP(9)
#endif
#ifdef CLOTHHIP_CELL_COUNTERS
            // census (wave 0): a substep in which nothing was adjusted, the plane restored every unpinned particle to its old
            // position (friction 1) and no spring was over-stretched leaves the positions as they were: tph[1] counts those,
            // tph[2] those whose predecessor was one too (state(t+1) == state(t): a fixed point)
            if (tid < 64) {
                const bool frozen_ = mode != 1 && misc[12] == 0 && !swept_ && k.one_m_fric == (T)0;
                tph[1] += frozen_ ? 64 : 0; tph[2] += misc[16] == 0 ? 64 : 0;     // [2]: no particle changed its collision cell in this substep
                frozen_prev_ = frozen_;
            }
            __syncthreads();
            if (tid == 0) { misc[12] = 0; misc[16] = 0; }
#endif
        }
        if (timing) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory"); }
        done++;
        if (sc.break_on_tear && misc[0]) break;                                            // cloth_env.py:511-514
        if (FUSED && sliced && misc[7] && it + 1 < sc.n_total) { it_next = it + 1; break; }
    }
        }   // the run
        resume_it = -1;
        if (!fused) { done_nf = done; break; }
        // ---- after the run: everything is re-read from LDS -----------------------------------------------------------
        {
            const FusedArgs<T> &F = *Fp;
            __syncthreads();
            if (it_next >= 0) {                          // cut by the time slice: park the run and leave
                if (tid == 0) {
                    eps->done_total += done - (resumed_run ? resume_done : 0);
                    account(eps->op, done - (resumed_run ? resume_done : 0));
                    EpResume *rs_ = F.resume + e;
                    rs_->valid = 1; rs_->it = it_next; rs_->done_partial = done; rs_->sc = sc; rs_->eps = *eps;
                    if (eps->rp >= 0 && F.resets != nullptr) {
                        ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + eps->n_resets);
                        rs_->rr = *rr_; rr_->consumed = 2;
                    }
                }
                __syncthreads();
                break;
            }
            const int tear_now = __builtin_amdgcn_readfirstlane(misc[0]);
            const int op = eps->op, rp = eps->rp, t_slot = eps->t_slot, n_resets = eps->n_resets;
            double mo[4] = {0.0, 0.0, 0.0, 0.0};
            if (op == OP_ACTION || op == OP_RESET_COND || op == OP_RESET_END) {
                // cloth_env.py:1020-1098 on the LDS-resident state; the sort buffers borrow the LDS behind the particle records
                auto src = [&](int i, double &x, double &y, double &z) { const Pt<T> c = cur[i]; x = (double)c.x; y = (double)c.y; z = (double)c.z; };
                // (the sort buffers and the hull stack live BEHIND the window table -- hash table, member lists, cell-ordered copy: all
                //  rebuilt below --, so the table itself stays in LDS for the whole launch and is not re-read from L2 after every action)
                metrics_block<NT, T, decltype(src), v_hull_idx(TAB)>(src, P, F.NS, F.NH, smem + lay.hkey, tid, F.half_thickness, mo);
                init_lds(tear_now, nullptr, nullptr);
                __syncthreads();
            }
            if (op == OP_ACTION && F.obs) {                                                       // '1d' observation, cloth_env.py:196-200
                float *o_ = F.obs + ((size_t)t_slot * F.E + e) * 3 * P;
                for (int i = tid; i < P; i += NT) { const Pt<T> c = cur[i]; o_[3 * i] = (float)c.x; o_[3 * i + 1] = (float)c.y; o_[3 * i + 2] = (float)c.z; }
            }
            if (op == OP_RESET_END && F.reset_obs) {                                              // what env.reset() returns
                float *o_ = F.reset_obs + ((size_t)e * F.n_scripts + n_resets) * 3 * P;
                for (int i = tid; i < P; i += NT) { const Pt<T> c = cur[i]; o_[3 * i] = (float)c.x; o_[3 * i + 1] = (float)c.y; o_[3 * i + 2] = (float)c.z; }
            }
            if (tid == 0) {
                if (F.budget_ticks != 0 && __builtin_amdgcn_s_memrealtime() - eps->t_launch >= F.budget_ticks) eps->stop = 1;
                eps->done_total += done - (resumed_run ? resume_done : 0);
                account(op, done - (resumed_run ? resume_done : 0));
                if (op == OP_ACTION) {
                    const int ep_steps = eps->ep_steps + 1;
                    const bool oob_ = mo[2] != 0.0;
                    // _terminal (cloth_env.py:684-715)
                    const bool dn = ep_steps >= F.ep.max_actions || tear_now != 0 || oob_ || mo[0] > F.ep.coverage_done;
                    ClothStepRecord *r_ = F.records + ((size_t)t_slot * F.E + e);
                    r_->action[0] = eps->act[0]; r_->action[1] = eps->act[1]; r_->action[2] = eps->act[2]; r_->action[3] = eps->act[3];
                    r_->coverage = mo[0]; r_->variance_inv = mo[1]; eps->last_cov = mo[0];
                    r_->executed = done; r_->n_grabbed = eps->n_grab; r_->iters_pull = eps->iters_pull;
                    r_->n_below_half_thickness = (int32_t)mo[3];
                    r_->ran = eps->decode_err ? 2 : 1; r_->oob = oob_ ? 1 : 0; r_->tear = tear_now ? 1 : 0; r_->done = dn ? 1 : 0;
                    r_->reset_before = (uint8_t)eps->reset_mark;
                    eps->reset_mark = 0; eps->ep_steps = ep_steps; eps->ep_done = dn ? 1 : 0; eps->t_slot = t_slot + 1; eps->n_ran += 1;
                } else {
                    const bool rngm = F.mt != nullptr;
                    const ClothResetScript *scr = rngm ? nullptr : F.scripts + ((size_t)e * F.n_scripts + n_resets);
                    ClothResetRecord *rr_ = F.resets ? F.resets + ((size_t)e * F.n_scripts + n_resets) : nullptr;
                    if (op == OP_RESET_COND) {
                        const double cmin = rngm ? 0.90 : scr->pull[rp >> 1].coverage_min;
                        eps->rp = mo[0] >= cmin ? rp + 1 : 6;                                     // cloth_env.py:866
                    } else if (op == OP_RESET_PULL) {
                        const int p_ = rp >> 1;
                        if (rr_) {
                            rr_->executed[p_] = done; rr_->pulls_run = eps->rs_pulls + 1;
                            rr_->action[p_][0] = eps->act[0]; rr_->action[p_][1] = eps->act[1];
                            rr_->action[p_][2] = eps->act[2]; rr_->action[p_][3] = eps->act[3];
                        }
                        eps->rs_pulls += 1; eps->rp = rp + 1;
                    } else if (op == OP_RESET_SETTLE) {
                        if (rr_) rr_->settle_executed += done;
                        eps->rp = (with_tier2 && rp == 8) ? 0 : 7;
                    } else {                                                                      // OP_RESET_END
                        if (rr_) { rr_->start_coverage = mo[0]; rr_->start_variance_inv = mo[1]; rr_->tear = tear_now; }
                        eps->last_cov = mo[0];
                        // a conditional pull that ran consumed RNG draws the later scripts were drawn without (clothhip.h)
                        if (!rngm) {
                            int n_uncond = 0;
                            for (int p_ = 0; p_ < scr->n_pulls; p_++) n_uncond += scr->pull[p_].need_coverage ? 0 : 1;
                            if (eps->rs_pulls > n_uncond) eps->chain_ok = 0;
                        }
                        eps->n_resets = n_resets + 1; eps->reset_mark = n_resets + 1; eps->rp = -1;
                    }
                }
            }
            __syncthreads();
            if (op == OP_RESET_END && F.mt != nullptr && F.domrand_words != 0)                    // cloth_env.py:786-789
                mt_skip_block<NT>(F.mt + (size_t)e * MT_WORDS, F.domrand_words, tid);
        }
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
            }
        }
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
