// cloth_common.hpp -- what every phase of the stepper shares: kernel-argument blocks, the episode state, arithmetic helpers (exact fp64 /
// fast fp32), wave primitives (DPP scans and sums, ballots, lane broadcasts), the particle record, the LDS carve-up.
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
    uint64_t t_deadline;   // t_launch + the launch's time budget (~0: none): what thread 0 compares the clock with in every substep of a time-sliced launch
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
    int32_t e0;              // env of workgroup 0: a time-sliced episode launch over more cloths than are resident goes out as one launch per generation (launch_run)
    const uint4 *lstc;       // fp64 LEAN build: [Ppad] per particle {stencil mask, 12 rest-length offsets in ulps (one byte each)}: the flat tiers' fp64 rest
                             // lengths (cloth.pyx:417 on the grid of :117-130) are one value per spring type up to a few dozen ulps -- rest = bits(pal_type) + offset
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
// The physics parameters a grid-specialised build (NS = 25 / 50) has compiled in: the shipped configuration's (cfg/t1_rgbd.yaml:5-24; for 50x50 the
// thickness of cfg/_json_files/default184.json:19). The host selects such a build only for a handle whose ClothParams equal these (spec_ns).
struct SpecPhys { double width, height, density, ks, damping, thickness, plane_friction, tear_thresh, gravity, minimum_z; int frames_per_sec, simulation_steps; };
constexpr SpecPhys spec_phys(int ns) { return SpecPhys{1.0, 1.0, 200.0, 10000.0, 2.0, ns == 50 ? 0.0095 : 0.02, 1.0, 2.0, -9.8, 0.0, 30, 30}; }
// ... and the stepper's constants derived from them, by the very expressions (same doubles, same roundings) make_consts evaluates on the host
template <typename T> constexpr DevConsts<T> spec_consts(int ns) {
    const SpecPhys p = spec_phys(ns);
    const double dx = p.width * 1.0 / (ns - 1), dy = p.height * 1.0 / (ns - 1);
    const double mass = p.density / ns / ns;
    const double delta_t = 1.0 / p.frames_per_sec / p.simulation_steps;
    const double w = 3 * dx, h = 3 * dy, t = (w > h) ? w : h;
    DevConsts<T> k{};
    k.mg = (T)(mass * p.gravity);
    k.ks_str = (T)(p.ks * 1.0); k.ks_bend = (T)(p.ks * 0.2);
    k.dsm = (T)((delta_t * delta_t) / mass);
    k.damp = (T)(1.0 - p.damping / 100.0);
    k.cw = (T)w; k.ch = (T)h; k.ct = (T)t;
    k.thresh = (T)(2.0 * p.thickness);
    k.sim_steps = (T)p.simulation_steps;
    k.min_z = (T)p.minimum_z;
    k.surf_off = (T)0.0001;
    k.one_m_fric = (T)(1. - p.plane_friction);
    k.tear_thresh = (T)p.tear_thresh;
    k.c11 = (T)1.1;
    return k;
}
// the fp32 palette of the flat tiers' rest lengths at grid ns (cloth.pyx:117-146, :417: structural dx, shearing sqrt(dx^2 + dy^2), bending 2 dx,
// rounded to float): what the LEAN arithmetic of a specialised fp32 build uses as literals; the host compares them with the palette it read
// back from the device's rest table (spec_ns) -- a mismatch selects the generic build
constexpr float spec_pal(int ns, int type) {
    const double dx = 1.0 / (ns - 1);
    return type == 0 ? (float)dx : (type == 1 ? (float)__builtin_sqrt(dx * dx + dx * dx) : (float)(2.0 * dx));
}
// NS > 0: the constants as LITERALS (no scalar loads at the head of every phase, no SGPRs held, constant subexpressions folded: +1 % on the
// headline) -- all but sim_steps: the fp32 arithmetic divides by it through v_rcp_f32, which the compiler would fold to the correctly rounded
// reciprocal, and the specialised build must stay bit-identical to the generic one (every other use of a constant is an exactly rounded operation).
template <typename T, int NS = 0> __device__ __forceinline__ DevConsts<T> load_consts(KArgsC<T> *p) {
    DevConsts<T> k;
    if constexpr (NS > 0) {
        constexpr DevConsts<T> c = spec_consts<T>(NS);
        k = c;
        k.sim_steps = p->k.sim_steps;
        return k;
    }
    k.mg = p->k.mg; k.ks_str = p->k.ks_str; k.ks_bend = p->k.ks_bend; k.dsm = p->k.dsm; k.damp = p->k.damp;
    k.cw = p->k.cw; k.ch = p->k.ch; k.ct = p->k.ct; k.thresh = p->k.thresh; k.sim_steps = p->k.sim_steps;
    k.min_z = p->k.min_z; k.surf_off = p->k.surf_off; k.one_m_fric = p->k.one_m_fric; k.tear_thresh = p->k.tear_thresh; k.c11 = p->k.c11;
    return k;
}
// (in a phase's scope: shadows the kernel's `k`, `P`, `Ppad`, `HT` by freshly loaded copies)
// Grid-specialised instantiations (round 6): k_run_schedule<..., NS> with NS = 25 / 50 knows a BASELINE grid at compile time -- particle count, padded
// count, hash-table size, window-table size, the whole LDS carve-up, "all phases on", whether the cell-ordered copy exists -- so none of it occupies
// registers across the substep loop (the generic build holds ~30 loop-invariant VGPRs of LDS base addresses and constants: headline +1.9 %, the
// six-per-CU build +3.3 %). NS = 0 is the generic build (any grid, debug phase masks). The host picks the specialised kernel only when every constant
// below equals what it computed for the handle (clothhip_api.hip::spec_ok); the names are used inside the kernel body, where NS and TAB are in scope.
constexpr int spec_p(int ns) { return ns * ns; }
constexpr int spec_ppad(int ns) { return (ns * ns + 63) / 64 * 64; }
// hash-table slots: the smallest power of two above 1.5 P -- except the two-cloths-per-CU layout of 50x50 (TAB 4), whose table is sized to the LDS left
constexpr int spec_ht(int ns, int tab) { if (ns == 50 && tab == 4) return 2880; int h = 64; while (h <= ns * ns + ns * ns / 2) h <<= 1; return h; }
constexpr int spec_htbits(int ns, int tab) { if (ns == 50 && tab == 4) return 0; int b = 0; while ((1 << b) < spec_ht(ns, tab)) b++; return b; }
constexpr int spec_nw(int ns) { return ns == 25 ? 55 : (ns == 50 ? 227 : 0); }             // windows of the strain sweep's table (cloth_tables.hpp)
constexpr int spec_spad(int ns) { return ns == 25 ? 3776 : (ns == 50 ? 14976 : 0); }       // its slots incl. padding
constexpr int spec_rshift(int ns) { return ns == 50 ? 2 : 0; }                             // unit of the entries' reach field
// the cell-ordered record copy: every specialised 25x25 layout but the LEAN builds for five / six cloths per CU has it; 50x50 at two per CU has not
constexpr int spec_cell_copy(int ns, int tab) { return ns == 25 ? (tab > -2 ? 1 : 0) : 0; }
#define KA_N(p_) (NS > 0 ? NS : (p_)->N)
#define KA_HTBITS(p_) (NS > 0 ? spec_htbits(NS, TAB) : (p_)->ht_bits)
#define KA_NW(p_) (NS > 0 ? spec_nw(NS) : (p_)->nW)
#define KA_SPAD(p_) (NS > 0 ? spec_spad(NS) : (p_)->Spad)
#define KA_RSHIFT(p_) (NS > 0 ? spec_rshift(NS) : (p_)->wt_rshift)
#define KA_CELLCOPY(p_) (NS > 0 ? spec_cell_copy(NS, TAB) : (p_)->cell_copy)
#define CLOTH_PHASE_DIMS() const int P = NS > 0 ? spec_p(NS) : Ak_->P, Ppad = NS > 0 ? spec_ppad(NS) : Ak_->Ppad, HT = NS > 0 ? spec_ht(NS, TAB) : Ak_->HT;
#define CLOTH_PHASE_ARGS()                                                        \
    asm volatile("" : "+s"(Ak_));                                                 \
    const DevConsts<T> k = load_consts<T, NS>(Ak_);                               \
    CLOTH_PHASE_DIMS()                                                            \
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
    int cur, eps, wtab, pslot, lstc, hkey, hco, memb, slot, misc, alist, olist, cpos, total;
    // tab 2 (the eight-wave LEAN build): like 1, plus the table slots of every particle's six own springs (u16 [6][Ppad]): the strain
    // pre-pass of the LEAN arithmetic needs the slot of a flagged spring, and read it from the L2-resident gather table otherwise
    // lst 1 (the fp64 LEAN build): the per-particle stencil constants (StepArgs::lstc, 16 bytes each) resident in LDS
    __host__ __device__ LdsLayout(int tsz, int Ppad, int Spad, int HT, int tab, int cp, int lst = 0) {
        int o = 0;
        auto take = [&](int bytes) { int r = o; o += (bytes + 15) / 16 * 16; return r; };
        cur = take(4 * Ppad * tsz);
        eps = take(EPSTATE_LDS_BYTES);   // EpState (fused episodes)
        wtab = take(tab >= 1 ? Spad * (tsz == 8 ? 16 : 8) : 0);   // WEnt<T>[Spad]
        pslot = take(tab == 2 ? (HK_SLOTS / 2) * Ppad * 2 : 0);
        lstc = take(lst ? Ppad * 16 : 0);
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

}  // namespace clothhip
