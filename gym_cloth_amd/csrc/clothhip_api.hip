// clothhip_api.hip -- C-ABI implementation of libclothhip.so (see include/clothhip.h).
// Host side only orchestrates: tables, uploads, launches. All physics runs in cloth_kernels.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "cloth_kernels.hpp"
#include "stepper_variants.hpp"
#include "lean_rates.hpp"
#include "cloth_render.hpp"

using namespace clothhip;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t err_ = (expr);                                                               \
        if (err_ != hipSuccess)                                                                 \
            return fail(CLOTHHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)


// LDS scratch of metrics_block (cloth_kernels.hpp): two sort buffers of NS values in the handle's precision, 64 doubles, NH u16 hull indices
static inline int metrics_scratch_bytes(int NS, int NH, int tsz, bool hull_idx = false) {
    return hull_idx ? 2 * NS * tsz + 64 * 8 + ((2 * NH + 15) / 16) * 16 : 2 * NS * tsz + (2 * NH + 64) * 8;
}

struct clothhip_handle {
    ClothParams prm{};
    int E = 0, N = 0, P = 0, Ppad = 0, S = 0, Spad = 0, precision = 0, device = 0;
    size_t tsz = 8;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_timing = false, pending_exec = false;
    void *d_pos = nullptr, *d_prev = nullptr, *d_rest = nullptr;
    void *d_flat = nullptr, *d_flat_rest = nullptr;   // flat tier-1 grid [3][Ppad] and its rest table [Spad] (window-table slot order), handle precision
    uint8_t *d_cnt = nullptr, *d_active = nullptr;
    int rest_stride = 0;
    int32_t *d_tear = nullptr, *d_exec = nullptr, *d_ngrab = nullptr, *d_stats = nullptr;
    ClothSchedule *d_sched = nullptr, *h_sched = nullptr;   // h_sched: pinned staging
    uint32_t *d_gather = nullptr, *d_wt_ent = nullptr;
    unsigned long long *d_wt_dep = nullptr;
    int cell_copy = 0;
    int HT = 0, ht_bits = 0, lds_bytes = 0, phase_mask = 15, nt = 256, ppt = 3;
    int tab = 0;            // the strain sweep's window table (+ rest lengths) resident in LDS: 0 no (streamed from L2), 1 yes
    bool rest_reg = false;
    // LEAN stepper (fp32, n_side <= 27, batches of >= 1024 cloths): 168 VGPRs and 33 KB of LDS per cloth -> three cloths per CU.
    // It needs ONE shared rest table whose fp32 values are one per spring type (checked on the device's table whenever that table
    // may have changed) and the regular gather stencil (checked once); otherwise the (0, false) variant runs on the same layout.
    bool lean = false, lean_dirty = true, lean_ok = false, lean_stencil_ok = false;
    bool relaxed = false;   // clothhip_set_relaxed_order(h, 1): THIS handle's episode launches run the relaxed-order companion kernel (bench only, no parity)
    int last_dispatches = 0; // kernel dispatches the last stepper launch was issued as (clothhip_last_dispatches)
    int spec_now = 0;        // 25 / 50: the layout in use runs that grid-specialised build (decided by lean_refresh per launch: spec_ns); 0: the generic build
    int last_spec = 0;       // what the last launch ran (clothhip_last_specialised)
    int lean_r = 3;         // cloths per CU the chosen LEAN build is compiled for (3: 168 VGPRs, 4: 128 VGPRs; 2: eight waves per cloth, table in LDS; 1: the large grids)
    float pal[3] = {0, 0, 0};
    double pal64[3] = {0, 0, 0};     // fp64 LEAN build: the smallest rest length of each spring type (the others are it + a few ulps: StepArgs::lstc)
    uint4 *d_lstc = nullptr;         // [Ppad] fp64 LEAN build: per particle {stencil mask, 12 offset bytes}
    int32_t last_variant[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // what the last launch ran (clothhip_last_variant)
    bool have_variant = false;
    int n_cus = 0;
    // (scratch_have / scratch_need: the LDS behind the hash table that the in-kernel metrics of the episode launches borrow, and what they need)
    struct Layout { int nt, ppt, tab; bool rest_reg; int cell_copy; int lds_bytes; int HT, ht_bits; int scratch_have, scratch_need; } lay_std = {256, 3, 0, false, 0, 0, 0, 0, 0, 0}, lay_lean = {256, 3, 0, true, 0, 0, 0, 0, 0, 0};
    struct OccKey { const void *fn; int lds; int occ; } occ_cache[8] = {};   // hipOccupancyMaxActiveBlocksPerMultiprocessor per (kernel, LDS bytes)
    double *d_levels = nullptr, *d_xy = nullptr, *d_radius = nullptr, *d_cov = nullptr, *d_vinv = nullptr;
    uint8_t *d_oob = nullptr;
    int32_t *d_hcnt = nullptr;      // per env: #points with z < thickness/2 (height reward, cloth_env.py:1047-1073)
    int n_grab_levels = 0;
    // clothhip_run_actions staging (device), grown on demand
    void *d_fz = nullptr, *d_fact = nullptr, *d_fscr = nullptr, *d_frec = nullptr, *d_frst = nullptr, *d_fobs = nullptr, *d_frobs = nullptr;
    int32_t *d_fsteps = nullptr, *d_fparg = nullptr;
    EpResume *d_resume = nullptr;   // [E] operations cut by a time slice (clothhip_run_actions), continued by the next launch
    uint32_t *d_fmt = nullptr;      // [E][MT_WORDS] numpy RandomState of every env (device-drawn resets)
    uint8_t *d_fdone = nullptr;
    double *d_fsum = nullptr;       // [E][4] per-env summary of the last episode launch (what the multi-GPU driver all-gathers)
    uint64_t *d_fticks = nullptr;   // [E][8] per-operation-class ticks and update() counts of the last episode launch
    int f_T = 0; size_t f_nscr = 0; bool f_pending = false, f_resets = false, f_obs = false, f_robs = false, f_mt = false;
    size_t cap_fact = 0, cap_frec = 0, cap_fobs = 0, cap_fscr = 0, cap_frst = 0, cap_frobs = 0, cap_fparg = 0;
    Topology topo;
    WindowTable wt;
    std::vector<unsigned char> stage;   // host staging for layout conversion
    std::vector<double> flat_rest;
};

extern "C" const char *clothhip_last_error(void) { return g_err.c_str(); }
extern "C" int clothhip_abi_version(void) { return CLOTHHIP_ABI_VERSION; }

extern "C" int clothhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static int check_params(const ClothParams *p) {
    if (!p) return fail(CLOTHHIP_EINVAL, "params is NULL");
    if (p->n_side < 3 || p->n_side > MAX_SIDE) return fail(CLOTHHIP_EINVAL, "n_side %d outside [3,%d]", p->n_side, MAX_SIDE);
    if (!(p->width > 0) || !(p->height > 0)) return fail(CLOTHHIP_EINVAL, "width/height must be > 0");
    if (p->height != p->width) return fail(CLOTHHIP_EINVAL, "height must equal width (cloth.pyx:91)");
    if (p->frames_per_sec <= 0 || p->simulation_steps <= 0) return fail(CLOTHHIP_EINVAL, "frames_per_sec/simulation_steps must be > 0");
    if (!(p->density > 0) || !(p->thickness > 0)) return fail(CLOTHHIP_EINVAL, "density/thickness must be > 0");
    return 0;
}

// ---- host restatement of Cloth.__init__ grid + rest lengths (cloth.pyx:92-146, :411-417) -----------
extern "C" int clothhip_init_grid(const ClothParams *p, int32_t tier, int32_t init_side,
                                  const double *rand_draws, double *pos, double *rest) {
    if (int rc = check_params(p)) return rc;
    if (tier < 1 || tier > 3) return fail(CLOTHHIP_EINVAL, "init tier %d (ValueError, cloth.pyx:131-132)", tier);
    if (tier == 2 && !rand_draws) return fail(CLOTHHIP_EINVAL, "tier 2 needs the P rand() draws");
    if (!pos) return fail(CLOTHHIP_EINVAL, "pos is NULL");
    const int N = p->n_side;
    const double dx = p->width * 1.0 / (N - 1), dy = p->height * 1.0 / (N - 1);   // cloth.pyx:55-56
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++) {
            const int i = r * N + c;
            double x, y, z;
            if (tier == 2) {
                double noise = rand_draws[i] * 0.01 - 0.005;           // cloth.pyx:101
                if (r == 0) noise = 0;                                 // :102-103
                x = init_side ? 0.0 + std::fabs(noise) : 1.0 - std::fabs(noise);   // :104-107
                y = dx * c; z = dy * r;                                // :109-110
            } else {
                x = dx * r; y = dy * c; z = 0.0;                       // :122-124
            }
            pos[3 * i] = x; pos[3 * i + 1] = y; pos[3 * i + 2] = z;
        }
    if (rest) {
        Topology t = build_topology(N);
        for (int s = 0; s < t.S; s++) {
            const double *A = pos + 3 * t.a[s], *B = pos + 3 * t.b[s];
            const double ux = A[0] - B[0], uy = A[1] - B[1], uz = A[2] - B[2];
            rest[s] = std::sqrt(ux * ux + uy * uy + uz * uz);         // cloth.pyx:417 via :17-18
        }
    }
    return 0;
}

extern "C" int clothhip_spring_topology(const ClothParams *p, int32_t *a, int32_t *b, uint8_t *type) {
    if (int rc = check_params(p)) return rc;
    Topology t = build_topology(p->n_side);
    if (a) memcpy(a, t.a.data(), sizeof(int32_t) * t.S);
    if (b) memcpy(b, t.b.data(), sizeof(int32_t) * t.S);
    if (type) memcpy(type, t.type.data(), t.S);
    return 0;
}

extern "C" int clothhip_selftest_windows(const ClothParams *p, int32_t *n_windows, int32_t *n_slots, int32_t *reach_shift,
                                         int32_t *spring_at, uint32_t *ent, uint64_t *dep, int32_t capacity) {
    if (int rc = check_params(p)) return rc;
    const Topology t = build_topology(p->n_side);
    const WindowTable W = build_windows(t, build_levels(t));
    if (n_windows) *n_windows = W.nW;
    if (n_slots) *n_slots = W.n_slots;
    if (reach_shift) *reach_shift = W.reach_shift;
    if ((spring_at || ent || dep) && capacity < W.n_slots) return fail(CLOTHHIP_EINVAL, "capacity below the table's slot count");
    if (spring_at) memcpy(spring_at, W.spring_at.data(), sizeof(int32_t) * W.n_slots);
    if (ent) memcpy(ent, W.ent.data(), sizeof(uint32_t) * W.n_slots);
    if (dep) memcpy(dep, W.dep.data(), sizeof(uint64_t) * W.n_slots);
    return 0;
}

template <typename T> static DevConsts<T> make_consts(const ClothParams &p) {
    const int N = p.n_side;
    const double dx = p.width * 1.0 / (N - 1), dy = p.height * 1.0 / (N - 1);
    const double mass = p.density / N / N;                              // cloth.pyx:178
    const double delta_t = 1.0 / p.frames_per_sec / p.simulation_steps; // :180
    const double w = 3 * dx, h = 3 * dy, t = (w > h) ? w : h;           // :308-310
    DevConsts<T> k;
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

static void free_handle(clothhip_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void *ptrs[] = {h->d_pos, h->d_prev, h->d_rest, h->d_cnt, h->d_active, h->d_tear, h->d_exec, h->d_ngrab, h->d_stats,
                    h->d_sched, h->d_flat, h->d_flat_rest, h->d_hcnt, h->d_fz, h->d_fact, h->d_fscr, h->d_frec, h->d_frst, h->d_fobs, h->d_frobs, h->d_fsteps, h->d_fparg, h->d_fdone, h->d_fticks, h->d_fsum, h->d_fmt, h->d_resume, h->d_gather, h->d_wt_ent, h->d_wt_dep, h->d_lstc, h->d_levels, h->d_xy, h->d_radius, h->d_cov, h->d_vinv, h->d_oob};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->h_sched) (void)hipHostFree(h->h_sched);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static const void *stepper_fn(const clothhip_handle *h, int fused);
static int spec_ns(const clothhip_handle *h, bool with_palette = true);

// The LDS a layout leaves the in-kernel metrics (from the hash table to the end of the allocation) against what they need; the
// allocation is padded behind the layout's end when that fits the budget (the kernel addresses LDS by the layout's offsets: bytes
// behind `total` are free). False: the episode launches cannot run on this layout.
static bool fit_scratch(clothhip_handle::Layout &L, int tsz, int Ppad, int Spad, int P, int budget) {
    int NS = 1; while (NS < P) NS <<= 1;
    const int lst = (tsz == 8 && v_lean(L.tab, L.rest_reg, tsz)) ? 1 : 0;
    const LdsLayout lay(tsz, Ppad, Spad, L.HT, L.tab == 2 ? 2 : (v_ldstab(L.tab) ? 1 : 0), L.cell_copy, lst);
    L.scratch_need = metrics_scratch_bytes(NS, Ppad + 8, tsz, v_hull_idx(L.tab, tsz, L.nt, L.ppt));
    if (L.lds_bytes < lay.total) L.lds_bytes = lay.total;
    if (L.lds_bytes - lay.hkey < L.scratch_need && lay.hkey + L.scratch_need <= budget) L.lds_bytes = (lay.hkey + L.scratch_need + 15) / 16 * 16;
    L.scratch_have = L.lds_bytes - lay.hkey;
    return L.scratch_have >= L.scratch_need;
}

// LDS a workgroup may use when r workgroups are to share a CU: LDS is allocated in granules of 1 280 bytes on gfx950 (128 granules = the CU's
// 160 KiB), so r cloths fit when each takes at most floor(128 / r) granules -- 160 KiB / r overstates that for r = 3, 5, 6.
static constexpr int LDS_GRANULE = 1280;
static constexpr int lds_budget(int r) { return (128 / (r < 1 ? 1 : r)) * LDS_GRANULE; }
static_assert(lds_budget(1) == 160 * 1024 && lds_budget(2) == 80 * 1024 && lds_budget(4) == 40 * 1024, "granule arithmetic");

// Which stepper variant and which LDS layout a handle runs: pure host logic (no HIP call), so that the CPU test suite can sweep it
// over grid sizes and precisions (clothhip_selftest_layout). Fills nt / ppt / HT / tab / rest_reg / cell_copy / lds_bytes, the
// standard layout lay_std and, where the LEAN arithmetic applies, lay_lean + lean_r.
// max_r: the highest residency the pick may choose (clothhip_create lowers it when the device's occupancy query grants the chosen LEAN
// build fewer workgroups per CU than it was planned for).
static void plan_layouts(clothhip_handle *h, int cus, const std::vector<uint32_t> &gather, int max_r = 6) {
    // threads per cloth x particles per thread (compile-time variants of the stepper)
    // P <= 768 (the 25x25 class, two cloths per CU): EIGHT waves per cloth -- 512 threads x 2 particles, compiled for 128 VGPRs: the cell
    // sweeps have eight ticket takers and the parallel phases two waves per SIMD to hide their LDS latency (+4 % fp32 standard
    // arithmetic, +9 % fp64, +13 % tier 2 over the four-wave 256 x 3 variants, bit-identical; CLOTHHIP_DEBUG_W8=0 selects those)
    const bool small_grid = h->P <= 768;
    const bool w8 = !(getenv("CLOTHHIP_DEBUG_W8") && atoi(getenv("CLOTHHIP_DEBUG_W8")) == 0);
    if (small_grid) { h->nt = w8 ? 512 : 256; h->ppt = w8 ? 2 : 3; }
    else if (h->P <= 2560 && !getenv("CLOTHHIP_DEBUG_NT1024")) { h->nt = 512; h->ppt = 5; }
    else if (h->P <= 3072) { h->nt = 1024; h->ppt = 3; } else { h->nt = 1024; h->ppt = 4; }
    h->HT = 64; h->ht_bits = 0;
    while (h->HT <= h->P + h->P / 2) h->HT <<= 1;
    while ((1 << h->ht_bits) < h->HT) h->ht_bits++;
    // large dynamic LDS (up to the CU's 160 KiB) for the stepper kernels. The static tables ride in LDS too
    // as long as TWO cloths still fit per CU (512 cloths = 2 per CU on the 256 CUs of an MI355X).
    {
        const int tsz = (int)h->tsz;
        const int precision = h->precision;
        // 256-thread variants: two cloths per CU (<= 80 KiB each); the larger ones own the CU (<= 160 KiB)
        const int budget = small_grid ? lds_budget(2) : lds_budget(1);
        const int tmax = h->nt <= 512 ? 1 : 0;
        h->tab = (tmax >= 1 && LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 1, 0).total <= budget) ? 1 : 0;
        if (const char *t = getenv("CLOTHHIP_DEBUG_TAB_LDS")) h->tab = std::min(h->tab, atoi(t));
        h->rest_reg = (h->nt == 256 && precision == CLOTHHIP_F32 && h->tab == 1);
        if (const char *t = getenv("CLOTHHIP_DEBUG_REST_REG")) h->rest_reg = h->rest_reg && atoi(t);
        // LEAN variants: three to six cloths per CU instead of two, each stepping at a lower rate (lean_rates.hpp, measured by
        // tools/measure_pick_table.py). A launch runs its cloths in generations of what is resident, so the batch size decides:
        // the largest rate_r / ceil(E / (r * CUs)) wins.
        {
            h->n_cus = cus;
            // substeps/s of ONE resident cloth at 2 (standard), 3 and 4 cloths per CU, relative to the standard variant's: measured
            // by tools/measure_pick_table.py on the bench workload and written to lean_rates.hpp (its output: profiles/)
            // r = 2: the EIGHT-WAVE LEAN build (512 threads x 2 particles, window table in LDS) when the flat palette holds, else the
            // standard variant; r = 3 .. 6: the four-wave LEAN builds with the table streamed from L2 (168 / 128 / 96 / 80 VGPRs; from
            // five per CU on without the cell-ordered record copy: 22.6 KB of LDS per cloth)
            const bool lean_able = small_grid && precision == CLOTHHIP_F32;
            const double rate[5] = {lean_able ? LEAN_RATE_2_PER_CU_8W : 1.0, LEAN_RATE_3_PER_CU, LEAN_RATE_4_PER_CU, LEAN_RATE_5_PER_CU, LEAN_RATE_6_PER_CU};
            double best = 0.0; int best_r = 2;
            for (int r = 2; r <= std::max(2, std::min(6, max_r)); r++) {
                // (r >= 3: the four-wave LEAN layout, table streamed, must fit r times in the CU's LDS -- 27x27 does not at five per CU)
                if (r >= 3 && (!lean_able || LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, 0).total > lds_budget(r))) continue;
                const double v = rate[r - 2] / (double)((h->E + r * cus - 1) / (r * cus));
                if (v > best * 1.02) { best = v; best_r = r; }
            }
            h->lean = lean_able;
            h->lean_r = best_r;
            // the large grids (one cloth per CU): the LEAN arithmetic frees the registers of the gather entries and takes the rest lengths
            // off the L2 path, which lets SIXTEEN waves step a cloth at 128 VGPRs (1024 threads x 3 or 4 particles; 50x50: 2.90 M/s
            // standard 512 x 5 -> 3.05 LEAN 512 x 5 -> 3.26 LEAN 1024 x 3); the standard variant stays as the fallback (per-env rest tables)
            if (!small_grid && precision == CLOTHHIP_F32) { h->lean = true; h->lean_r = 1; }
            // fp64, 25x25 class, eight waves per cloth (round 6): the LEAN arithmetic with per-spring ulp offsets (StepArgs::lstc) -- no gather-table and no
            // rest-length loads from L2 in the Hooke gather and the strain pre-pass; same layout class as the standard fp64 variant (two cloths per CU)
            if (small_grid && precision == CLOTHHIP_F64 && h->nt == 512) { h->lean = true; h->lean_r = 2; }
        }
        if (const char *t = getenv("CLOTHHIP_DEBUG_LEAN")) {      // 0: never; 8 (or 2): the eight-wave build; 3 (or 1) / 4 / 5 / 6: the LEAN build for that many cloths per CU, whatever the batch size
            const int v = atoi(t);
            if (v == 0) h->lean = false;
            else if (small_grid && precision == CLOTHHIP_F32) { h->lean = true; h->lean_r = (v == 8 || v == 2) ? 2 : ((v >= 4 && v <= 6) ? v : 3); }
        }
        if (h->lean) {
            // the arithmetic stencil of the LEAN kernel against the gather table built from the reference's spring list
            h->lean_stencil_ok = true;
            for (int i = 0; i < h->P && h->lean_stencil_ok; i++) {
                const uint32_t vm = lean_valid_mask(i / h->N, i % h->N, h->N);
                int slot = 0;
                for (int k = 0; k < HK_SLOTS; k++) {
                    if (!((vm >> k) & 1u)) continue;
                    const int off[12] = {-h->N, -1, -h->N - 1, -h->N + 1, -2 * h->N, -2, 1, 2, h->N - 1, h->N, h->N + 1, 2 * h->N};
                    const uint32_t want = (uint32_t)(i + off[k]) | HK_VALID | (k < HK_SLOTS / 2 ? HK_ASB : 0u) | (lean_bend(k) ? HK_BEND : 0u);
                    const uint32_t g = gather[(size_t)slot * h->Ppad + i];
                    const uint32_t have = g & (HK_NBR_MASK | HK_VALID | HK_ASB | HK_BEND);
                    const int sp = h->wt.spring_at[(g >> HK_POS_SHIFT) & HK_POS_MASK];
                    const int ty = sp >= 0 ? h->topo.type[sp] : -1;
                    const int want_ty = lean_bend(k) ? SPRING_BENDING : (lean_shear(k) ? SPRING_SHEARING : SPRING_STRUCTURAL);
                    if (have != want || ty != want_ty) h->lean_stencil_ok = false;
                    slot++;
                }
                if (slot < HK_SLOTS && h->lean_stencil_ok && (gather[(size_t)slot * h->Ppad + i] & HK_VALID)) h->lean_stencil_ok = false;
            }
            if (!h->lean_stencil_ok) h->lean = false;
        }
        // the cell-ordered record copy for the collision pre-check is taken only if it does not cost the table its place
        h->cell_copy = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, h->tab, 1).total <= budget ? 1 : 0;
        if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) h->cell_copy = h->cell_copy && atoi(t);
        h->lds_bytes = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, h->tab, h->cell_copy).total;
        h->lay_std = {h->nt, h->ppt, h->tab, h->rest_reg, h->cell_copy, h->lds_bytes, h->HT, h->ht_bits, 0, 0};
        // the in-kernel metrics of the episode launches borrow the LDS from the hash table on (the window table in front of it stays
        // resident). With the table in LDS but no room for the cell-ordered copy that region can be too small (fp64 21, 22, 30-32;
        // fp32 41-43): the allocation is then padded behind the layout's end, or, if the budget forbids that, the table leaves LDS
        if (!fit_scratch(h->lay_std, tsz, h->Ppad, h->Spad, h->P, budget) && h->tab == 1) {
            h->tab = 0; h->rest_reg = false;
            h->cell_copy = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, 1).total <= budget ? 1 : 0;
            if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) h->cell_copy = h->cell_copy && atoi(t);
            h->lay_std = {h->nt, h->ppt, 0, false, h->cell_copy, LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, h->cell_copy).total, h->HT, h->ht_bits, 0, 0};
            fit_scratch(h->lay_std, tsz, h->Ppad, h->Spad, h->P, budget);
        }
        h->lds_bytes = h->lay_std.lds_bytes;
        if (h->lean) {                                   // the lean layout: window table streamed from L2, 33 KB of LDS
            int cc = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, 1).total <= lds_budget(std::max(h->lean_r, 3)) ? 1 : 0;
            if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) cc = cc && atoi(t);
            h->lay_lean = {256, 3, h->lean_r >= 4 ? 3 - h->lean_r : 0, true, cc, LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, cc).total, h->HT, h->ht_bits};   // (r = 3, 4: four waves per cloth)
            if (precision == CLOTHHIP_F64) {             // fp64 LEAN: table streamed (TAB 0), the stencil constants in LDS
                cc = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, 1, 1).total <= lds_budget(2) ? 1 : 0;
                if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) cc = cc && atoi(t);
                h->lay_lean = {512, 2, 0, true, cc, LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, cc, 1).total, h->HT, h->ht_bits};
                if (h->lay_lean.lds_bytes > lds_budget(2)) h->lean = false;
            } else
            if (h->lean_r == 2 && small_grid) {          // eight waves per cloth, two cloths per CU: the standard variant's LDS budget
                cc = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 2, 1).total <= 80 * 1024 ? 1 : 0;
                if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) cc = cc && atoi(t);
                h->lay_lean = {512, 2, 2, true, cc, LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 2, cc).total, h->HT, h->ht_bits};
                if (h->lay_lean.lds_bytes > 80 * 1024 || h->P > 1024) h->lean = false;      // (the table must fit beside a second cloth)
            }
            if (h->lean_r == 1) {                        // the whole CU: same LDS budget as the standard variant of these grids
                cc = LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, 1).total <= 160 * 1024 ? 1 : 0;
                if (const char *t = getenv("CLOTHHIP_DEBUG_CELL_COPY")) cc = cc && atoi(t);
                h->lay_lean = {1024, h->P <= 3072 ? 3 : 4, 3, true, cc, LdsLayout(tsz, h->Ppad, h->Spad, h->HT, 0, cc).total, h->HT, h->ht_bits};
                // TWO large-grid cloths per CU (eight waves each, 128 VGPRs) when the batch has more cloths than the device has CUs and it
                // pays by the measured rates: <= 80 KB of LDS per cloth -- no cell-ordered copy, and a hash table of just enough slots
                // (not a power of two: > P, so that a free slot always exists, and large enough that the in-kernel metrics' scratch fits)
                int NSb = 1; while (NSb < h->P) NSb <<= 1;
                int ht2 = (h->P / 64 + 2) * 64;
                while (LdsLayout(tsz, h->Ppad, h->Spad, ht2, 0, 0).total - LdsLayout(tsz, h->Ppad, h->Spad, ht2, 0, 0).hkey < metrics_scratch_bytes(NSb, h->Ppad + 8, tsz, true)) ht2 += 64;
                const int lds2 = LdsLayout(tsz, h->Ppad, h->Spad, ht2, 0, 0).total;
                const int gens1 = (h->E + h->n_cus - 1) / h->n_cus, gens2 = (h->E + 2 * h->n_cus - 1) / (2 * h->n_cus);
                const bool two = h->P <= 2560 && lds2 <= 80 * 1024 && LEAN_RATE_LARGE_2_PER_CU / gens2 > 1.02 / gens1;
                int want2 = two ? 1 : 0;
                if (const char *t = getenv("CLOTHHIP_DEBUG_LARGE2")) want2 = atoi(t) && h->P <= 2560 && lds2 <= 80 * 1024;
                if (want2) { h->lay_lean = {512, 5, 4, true, 0, lds2, ht2, 0}; h->lean_r = 2; }
            }
            // the in-kernel metrics borrow the region behind the hash table (clothhip_fused_supported): it must hold them here too
            const int lean_budget = lds_budget(h->lean_r);
            if (!fit_scratch(h->lay_lean, tsz, h->Ppad, h->Spad, h->P, lean_budget)) h->lean = false;
        }
    }
}

extern "C" int clothhip_create(const ClothParams *params, int32_t n_envs, int32_t device, int32_t precision,
                               clothhip_handle **out) {
    if (!out) return fail(CLOTHHIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (int rc = check_params(params)) return rc;
    if (n_envs < 1) return fail(CLOTHHIP_EINVAL, "n_envs must be >= 1");
    if (precision != CLOTHHIP_F64 && precision != CLOTHHIP_F32) return fail(CLOTHHIP_EINVAL, "precision must be 0 (f64) or 1 (f32)");
    int ndev = clothhip_device_count();
    if (ndev <= 0) return fail(CLOTHHIP_ENODEV, "no HIP device visible: libclothhip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(CLOTHHIP_EINVAL, "device %d outside [0,%d)", device, ndev);
    HIPCHECK(hipSetDevice(device));
    clothhip_handle *h = new (std::nothrow) clothhip_handle();
    if (!h) return fail(CLOTHHIP_ENOMEM, "out of host memory");
    h->prm = *params; h->E = n_envs; h->device = device; h->precision = precision;
    h->N = params->n_side; h->P = h->N * h->N; h->Ppad = (h->P + 63) / 64 * 64;
    h->tsz = precision == CLOTHHIP_F64 ? 8 : 4;
    h->topo = build_topology(h->N);
    h->wt = build_windows(h->topo, build_levels(h->topo));
    h->S = h->topo.S; h->Spad = h->wt.n_slots;               // rest-length arrays are kept in window-table slot order
    if (const char *pmk = getenv("CLOTHHIP_DEBUG_PHASES")) h->phase_mask = atoi(pmk);
    std::vector<uint32_t> gather = build_gather(h->topo, h->wt, h->Ppad);
    std::vector<double> levels = build_grab_levels(params->height, params->thickness);
    h->n_grab_levels = (int)levels.size();

#define HC(expr)                                                                                      \
    do {                                                                                              \
        hipError_t err_ = (expr);                                                                     \
        if (err_ != hipSuccess) {                                                                     \
            int rc_ = fail(err_ == hipErrorOutOfMemory ? CLOTHHIP_ENOMEM : CLOTHHIP_EHIP,             \
                           "%s failed: %s", #expr, hipGetErrorString(err_));                          \
            free_handle(h);                                                                           \
            return rc_;                                                                               \
        }                                                                                             \
    } while (0)
    HC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HC(hipEventCreate(&h->ev0));
    HC(hipEventCreate(&h->ev1));
    const size_t E = h->E;
    HC(hipMalloc(&h->d_pos, E * 3 * h->Ppad * h->tsz));
    HC(hipMalloc(&h->d_prev, E * 3 * h->Ppad * h->tsz));
    HC(hipMalloc(&h->d_rest, E * h->Spad * h->tsz));
    HC(hipMalloc(&h->d_cnt, E * h->Ppad));
    HC(hipMalloc(&h->d_active, E));
    HC(hipMalloc(&h->d_tear, E * 4));
    HC(hipMalloc(&h->d_exec, E * 4));
    HC(hipMalloc(&h->d_ngrab, E * 4));
    HC(hipMalloc(&h->d_stats, E * 64));
    HC(hipMemset(h->d_stats, 0, E * 64));
    HC(hipMalloc(&h->d_sched, E * sizeof(ClothSchedule)));
    HC(hipHostMalloc((void **)&h->h_sched, E * sizeof(ClothSchedule), hipHostMallocDefault));
    HC(hipMalloc(&h->d_gather, gather.size() * 4));
    HC(hipMalloc(&h->d_wt_ent, (size_t)h->Spad * 4));
    HC(hipMalloc(&h->d_wt_dep, (size_t)h->Spad * 8));
    HC(hipMalloc(&h->d_lstc, (size_t)h->Ppad * 16));
    HC(hipMemset(h->d_lstc, 0, (size_t)h->Ppad * 16));
    HC(hipMalloc(&h->d_levels, (levels.size() + 1) * 8));
    HC(hipMalloc(&h->d_xy, E * 2 * 8));
    HC(hipMalloc(&h->d_radius, E * 8));
    HC(hipMalloc(&h->d_cov, E * 8));
    HC(hipMalloc(&h->d_vinv, E * 8));
    HC(hipMalloc(&h->d_oob, E));
    HC(hipMalloc(&h->d_hcnt, E * 4));
    HC(hipMalloc(&h->d_resume, E * sizeof(EpResume)));
    HC(hipMemset(h->d_resume, 0, E * sizeof(EpResume)));
    HC(hipMalloc(&h->d_flat, (size_t)3 * h->Ppad * h->tsz));
    HC(hipMalloc(&h->d_flat_rest, (size_t)h->Spad * h->tsz));
    HC(hipMemcpy(h->d_gather, gather.data(), gather.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(h->d_wt_ent, h->wt.ent.data(), (size_t)h->Spad * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(h->d_wt_dep, h->wt.dep.data(), (size_t)h->Spad * 8, hipMemcpyHostToDevice));
    HC(hipMemset(h->d_rest, 0, E * h->Spad * h->tsz));
    if (!levels.empty()) HC(hipMemcpy(h->d_levels, levels.data(), levels.size() * 8, hipMemcpyHostToDevice));
    HC(hipMemset(h->d_exec, 0, E * 4));
    // large dynamic LDS (up to the CU's 160 KiB) for the stepper kernels: which variant, which layout (plan_layouts)
    {
        hipDeviceProp_t dp;
        int cus = 256;
        if (hipGetDeviceProperties(&dp, device) == hipSuccess && dp.multiProcessorCount > 0) cus = dp.multiProcessorCount;
        plan_layouts(h, cus, gather);
        // the pick assumed lean_r resident cloths per CU: ask the device (registers, LDS granules, what else it counts) and fall back to the
        // best residency it does grant -- a build planned for r that runs at r - 1 would be slower than the build meant for r - 1
        for (int guard = 0; guard < 5 && h->lean && h->lean_r >= 3 && !getenv("CLOTHHIP_DEBUG_LEAN"); guard++) {
            const clothhip_handle::Layout keep = {h->nt, h->ppt, h->tab, h->rest_reg, h->cell_copy, h->lds_bytes};
            h->nt = h->lay_lean.nt; h->ppt = h->lay_lean.ppt; h->tab = h->lay_lean.tab; h->rest_reg = h->lay_lean.rest_reg;
            const void *fl = stepper_fn(h, 1);
            h->nt = keep.nt; h->ppt = keep.ppt; h->tab = keep.tab; h->rest_reg = keep.rest_reg;
            int occ = 0;
            if (!fl || hipFuncSetAttribute(fl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fl, h->lay_lean.nt, (size_t)h->lay_lean.lds_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            if (occ >= h->lean_r) break;
            plan_layouts(h, cus, gather, std::max(2, occ));
        }
        if (h->lds_bytes > 160 * 1024) { free_handle(h); return fail(CLOTHHIP_EINVAL, "n_side %d needs %d B of LDS (> 160 KiB)", h->N, h->lds_bytes); }
        if (h->lean) {                                   // the lean kernels too (which layout runs is decided per launch)
            const clothhip_handle::Layout keep = {h->nt, h->ppt, h->tab, h->rest_reg, h->cell_copy, h->lds_bytes};
            const int keep_ht = h->HT, keep_hb = h->ht_bits;
            h->nt = h->lay_lean.nt; h->ppt = h->lay_lean.ppt; h->tab = h->lay_lean.tab; h->rest_reg = h->lay_lean.rest_reg;
            h->cell_copy = h->lay_lean.cell_copy; h->HT = h->lay_lean.HT; h->ht_bits = h->lay_lean.ht_bits;
            for (int sp = 0; sp < 2; sp++) {             // the generic build of the layout and, where it exists for it, the grid-specialised one
                h->spec_now = sp == 1 ? spec_ns(h, false) : 0;
                if (sp == 1 && !h->spec_now) break;
                for (int f = 0; f < 3; f++) {
                    const void *fl = stepper_fn(h, f);
                    if (!fl) { free_handle(h); return fail(CLOTHHIP_EINVAL, "no lean stepper variant for n_side %d", h->N); }
                    HC(hipFuncSetAttribute(fl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                }
            }
            h->spec_now = 0; h->cell_copy = keep.cell_copy; h->HT = keep_ht; h->ht_bits = keep_hb;
            h->nt = keep.nt; h->ppt = keep.ppt; h->tab = keep.tab; h->rest_reg = keep.rest_reg;
        }
        const void *fn = stepper_fn(h, 0), *fnf = stepper_fn(h, 1), *fnf2 = stepper_fn(h, 2);
        if (!fn || !fnf || !fnf2) { free_handle(h); return fail(CLOTHHIP_EINVAL, "no stepper variant for n_side %d", h->N); }
        HC(hipFuncSetAttribute(fnf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HC(hipFuncSetAttribute(fnf2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        // the attribute is per kernel function and process-global: always the CU's full 160 KiB, so that a later handle
        // with a smaller footprint can never lower it under an earlier one
        HC(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->spec_now = spec_ns(h, false);                 // the grid-specialised build of the standard layout, where one exists (tier 2 at 25x25)
        if (h->spec_now) {
            for (int f = 0; f < 3; f++) {
                const void *fs = stepper_fn(h, f);
                if (fs) HC(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
        }
        h->spec_now = 0;
    }
#undef HC
    // initial state: flat tier-1 grid for every env, shared rest table
    std::vector<double> pos((size_t)h->P * 3), rest(h->S);
    int rc = clothhip_init_grid(params, 1, 0, nullptr, pos.data(), rest.data());
    if (rc) { free_handle(h); return rc; }
    h->flat_rest = rest;
    std::vector<double> all((size_t)h->E * h->P * 3);
    for (int e = 0; e < h->E; e++) memcpy(all.data() + (size_t)e * h->P * 3, pos.data(), sizeof(double) * h->P * 3);
    std::vector<uint8_t> pin((size_t)h->E * h->P, 0);
    rc = clothhip_set_state(h, 0, h->E, all.data(), all.data(), pin.data(), rest.data(), CLOTHHIP_REST_SHARED);
    if (rc) { free_handle(h); return rc; }
    // the flat grid and its rest table stay on the device for clothhip_reset_flat / the in-kernel episode reset
    if (hipMemcpy(h->d_flat, h->d_pos, (size_t)3 * h->Ppad * h->tsz, hipMemcpyDeviceToDevice) != hipSuccess ||
        hipMemcpy(h->d_flat_rest, h->d_rest, (size_t)h->Spad * h->tsz, hipMemcpyDeviceToDevice) != hipSuccess) {
        free_handle(h);
        return fail(CLOTHHIP_EHIP, "copying the flat-grid template failed");
    }
    *out = h;
    return 0;
}

extern "C" int clothhip_destroy(clothhip_handle *h) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_handle(h);
    return 0;
}

extern "C" int clothhip_num_points(const clothhip_handle *h) { return h ? h->P : fail(CLOTHHIP_EINVAL, "handle is NULL"); }
extern "C" int clothhip_num_springs(const clothhip_handle *h) { return h ? h->S : fail(CLOTHHIP_EINVAL, "handle is NULL"); }
extern "C" int clothhip_num_envs(const clothhip_handle *h) { return h ? h->E : fail(CLOTHHIP_EINVAL, "handle is NULL"); }
extern "C" int clothhip_precision(const clothhip_handle *h) { return h ? h->precision : fail(CLOTHHIP_EINVAL, "handle is NULL"); }
extern "C" void *clothhip_stream(clothhip_handle *h) { return h ? (void *)h->stream : nullptr; }

// Any state change from outside the episode launches (uploads, resets, grabs, raw schedules) voids the operation a time slice
// left in flight -- for the envs that call touches, and only for them: the parked operations of the others continue in the next
// episode launch. d_mask: device mask [E] (nullptr = all); d_sched: device schedules whose active flag selects (or nullptr).
static int drop_in_flight(clothhip_handle *h, const uint8_t *d_mask, const ClothSchedule *d_sched) {
    if (!h->d_resume) return 0;
    hipLaunchKernelGGL(k_clear_resume, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, h->d_resume, d_mask, d_sched, h->E);
    HIPCHECK(hipGetLastError());
    return 0;
}
static int drop_in_flight_range(clothhip_handle *h, int env0, int n) {
    if (h->d_resume && n > 0) HIPCHECK(hipMemsetAsync(h->d_resume + env0, 0, (size_t)n * sizeof(EpResume), h->stream));
    return 0;
}

static int check_range(const clothhip_handle *h, int env0, int n) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    if (env0 < 0 || n < 0 || env0 + n > h->E) return fail(CLOTHHIP_EINVAL, "env range [%d,%d) outside [0,%d)", env0, env0 + n, h->E);
    return 0;
}

// [n][P][3] double  ->  [n][3][Ppad] T
template <typename T> static void aos_to_soa(const double *src, T *dst, int n, int P, int Ppad) {
    for (int e = 0; e < n; e++) {
        const double *s = src + (size_t)e * P * 3;
        T *d = dst + (size_t)e * 3 * Ppad;
        for (int i = 0; i < P; i++) { d[i] = (T)s[3 * i]; d[Ppad + i] = (T)s[3 * i + 1]; d[2 * Ppad + i] = (T)s[3 * i + 2]; }
        for (int i = P; i < Ppad; i++) { d[i] = 0; d[Ppad + i] = 0; d[2 * Ppad + i] = 0; }
    }
}
template <typename T> static void soa_to_aos(const T *src, double *dst, int n, int P, int Ppad) {
    for (int e = 0; e < n; e++) {
        const T *s = src + (size_t)e * 3 * Ppad;
        double *d = dst + (size_t)e * P * 3;
        for (int i = 0; i < P; i++) { d[3 * i] = (double)s[i]; d[3 * i + 1] = (double)s[Ppad + i]; d[3 * i + 2] = (double)s[2 * Ppad + i]; }
    }
}

extern "C" int clothhip_set_state(clothhip_handle *h, int32_t env0, int32_t n, const double *pos, const double *prev,
                                  const uint8_t *pinned, const double *rest, int32_t flags) {
    if (int rc = check_range(h, env0, n)) return rc;
    if (int rc = drop_in_flight_range(h, env0, n)) return rc;
    const bool rest_shared = (flags & CLOTHHIP_REST_SHARED) != 0;
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    const size_t per = (size_t)3 * h->Ppad * h->tsz;
    for (int pass = 0; pass < 2; pass++) {
        const double *src = pass == 0 ? pos : prev;
        if (!src) continue;
        h->stage.resize(per * n);
        if (h->precision == CLOTHHIP_F64) aos_to_soa<double>(src, (double *)h->stage.data(), n, h->P, h->Ppad);
        else aos_to_soa<float>(src, (float *)h->stage.data(), n, h->P, h->Ppad);
        char *dst = (char *)(pass == 0 ? h->d_pos : h->d_prev) + per * env0;
        HIPCHECK(hipMemcpy(dst, h->stage.data(), per * n, hipMemcpyHostToDevice));
    }
    if (pos && !(flags & CLOTHHIP_KEEP_TEAR)) HIPCHECK(hipMemset(h->d_tear + env0, 0, (size_t)n * 4));
    if (pinned) {
        std::vector<uint8_t> c((size_t)n * h->Ppad, 0);
        for (int e = 0; e < n; e++)
            for (int i = 0; i < h->P; i++) c[(size_t)e * h->Ppad + i] = pinned[(size_t)e * h->P + i] ? 1 : 0;
        HIPCHECK(hipMemcpy(h->d_cnt + (size_t)env0 * h->Ppad, c.data(), c.size(), hipMemcpyHostToDevice));
    }
    if (rest) {
        if (!rest_shared && h->rest_stride == 0 && !(env0 == 0 && n == h->E)) {
            // switching from the shared table to per-env tables: replicate the shared one first
            std::vector<unsigned char> one((size_t)h->Spad * h->tsz);
            HIPCHECK(hipMemcpy(one.data(), h->d_rest, one.size(), hipMemcpyDeviceToHost));
            for (int e = 1; e < h->E; e++)
                HIPCHECK(hipMemcpy((char *)h->d_rest + (size_t)e * one.size(), one.data(), one.size(), hipMemcpyHostToDevice));
        }
        const int nt = rest_shared ? 1 : n;
        std::vector<unsigned char> buf((size_t)nt * h->Spad * h->tsz, 0);
        for (int e = 0; e < nt; e++)
            for (int p = 0; p < h->S; p++) {
                const int i = h->wt.slot_of[p];                               // list order -> table slot (empty slots stay 0)
                const double v = rest[(size_t)e * h->S + p];
                if (h->precision == CLOTHHIP_F64) ((double *)buf.data())[(size_t)e * h->Spad + i] = v;
                else ((float *)buf.data())[(size_t)e * h->Spad + i] = (float)v;
            }
        char *dst = (char *)h->d_rest + (rest_shared ? 0 : (size_t)env0 * h->Spad * h->tsz);
        HIPCHECK(hipMemcpy(dst, buf.data(), buf.size(), hipMemcpyHostToDevice));
        h->rest_stride = rest_shared ? 0 : h->Spad;
        h->lean_dirty = true;
    }
    return 0;
}

extern "C" int clothhip_get_state(clothhip_handle *h, int32_t env0, int32_t n, double *pos, double *prev, uint8_t *pinned) {
    if (int rc = check_range(h, env0, n)) return rc;
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    const size_t per = (size_t)3 * h->Ppad * h->tsz;
    for (int pass = 0; pass < 2; pass++) {
        double *dst = pass == 0 ? pos : prev;
        if (!dst) continue;
        h->stage.resize(per * n);
        const char *src = (const char *)(pass == 0 ? h->d_pos : h->d_prev) + per * env0;
        HIPCHECK(hipMemcpy(h->stage.data(), src, per * n, hipMemcpyDeviceToHost));
        if (h->precision == CLOTHHIP_F64) soa_to_aos<double>((const double *)h->stage.data(), dst, n, h->P, h->Ppad);
        else soa_to_aos<float>((const float *)h->stage.data(), dst, n, h->P, h->Ppad);
    }
    if (pinned) {
        std::vector<uint8_t> c((size_t)n * h->Ppad);
        HIPCHECK(hipMemcpy(c.data(), h->d_cnt + (size_t)env0 * h->Ppad, c.size(), hipMemcpyDeviceToHost));
        for (int e = 0; e < n; e++)
            for (int i = 0; i < h->P; i++) pinned[(size_t)e * h->P + i] = c[(size_t)e * h->Ppad + i] ? 1 : 0;
    }
    return 0;
}

extern "C" int clothhip_get_rest(clothhip_handle *h, int32_t env0, int32_t n, double *rest) {
    if (int rc = check_range(h, env0, n)) return rc;
    if (!rest) return fail(CLOTHHIP_EINVAL, "rest is NULL");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    std::vector<unsigned char> buf((size_t)h->Spad * h->tsz);
    for (int e = 0; e < n; e++) {
        const char *src = (const char *)h->d_rest + (size_t)(env0 + e) * h->rest_stride * h->tsz;   // stride 0: the shared table
        if (e == 0 || h->rest_stride) HIPCHECK(hipMemcpy(buf.data(), src, buf.size(), hipMemcpyDeviceToHost));
        for (int p = 0; p < h->S; p++) {        // table slot -> list order (Spring.rest_length of cloth.springs[p])
            const int i = h->wt.slot_of[p];
            rest[(size_t)e * h->S + p] = h->precision == CLOTHHIP_F64 ? ((const double *)buf.data())[i] : (double)((const float *)buf.data())[i];
        }
    }
    return 0;
}

extern "C" int clothhip_reset_flat(clothhip_handle *h, const uint8_t *mask) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    HIPCHECK(hipSetDevice(h->device));
    if (mask) HIPCHECK(hipMemcpyAsync(h->d_active, mask, (size_t)h->E, hipMemcpyHostToDevice, h->stream));
    if (int rc = drop_in_flight(h, mask ? h->d_active : nullptr, nullptr)) return rc;
    if (h->precision == CLOTHHIP_F64)
        hipLaunchKernelGGL(k_reset_flat<double>, dim3(h->E), dim3(256), 0, h->stream, (double *)h->d_pos, (double *)h->d_prev, h->d_cnt,
                           h->d_tear, (const double *)h->d_flat, mask ? h->d_active : nullptr, h->Ppad, (double *)h->d_rest,
                           (const double *)h->d_flat_rest, h->rest_stride, h->Spad);
    else
        hipLaunchKernelGGL(k_reset_flat<float>, dim3(h->E), dim3(256), 0, h->stream, (float *)h->d_pos, (float *)h->d_prev, h->d_cnt,
                           h->d_tear, (const float *)h->d_flat, mask ? h->d_active : nullptr, h->Ppad, (float *)h->d_rest,
                           (const float *)h->d_flat_rest, h->rest_stride, h->Spad);
    // (the LEAN palette verdict stands: a shared rest table is not touched here, and per-env tables rule the variant out anyway)
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_get_tear(clothhip_handle *h, uint8_t *tear) {
    if (!h || !tear) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    std::vector<int32_t> t(h->E);
    HIPCHECK(hipMemcpy(t.data(), h->d_tear, (size_t)h->E * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < h->E; e++) tear[e] = t[e] ? 1 : 0;
    return 0;
}

extern "C" int clothhip_set_tear(clothhip_handle *h, const uint8_t *tear) {
    if (!h || !tear) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    std::vector<int32_t> t(h->E);
    for (int e = 0; e < h->E; e++) t[e] = tear[e] ? 1 : 0;
    HIPCHECK(hipMemcpy(h->d_tear, t.data(), (size_t)h->E * 4, hipMemcpyHostToDevice));
    return 0;
}

static int do_grab(clothhip_handle *h, const double *xy, const double *radius, const uint8_t *active,
                   int32_t *n_grabbed, int top) {
    if (!h || !xy) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipMemcpyAsync(h->d_xy, xy, (size_t)h->E * 16, hipMemcpyHostToDevice, h->stream));
    if (radius) HIPCHECK(hipMemcpyAsync(h->d_radius, radius, (size_t)h->E * 8, hipMemcpyHostToDevice, h->stream));
    if (active) HIPCHECK(hipMemcpyAsync(h->d_active, active, (size_t)h->E, hipMemcpyHostToDevice, h->stream));
    if (int rc = drop_in_flight(h, active ? h->d_active : nullptr, nullptr)) return rc;
    if (h->precision == CLOTHHIP_F64) {
        GrabArgs<double> a{(const double *)h->d_pos, h->d_cnt, h->d_xy, radius ? h->d_radius : nullptr,
                           active ? h->d_active : nullptr, h->d_ngrab, h->d_levels, h->n_grab_levels, h->P, h->Ppad, top,
                           h->prm.grip_radius, 2 * h->prm.thickness};
        hipLaunchKernelGGL(k_grab<double>, dim3(h->E), dim3(64), 0, h->stream, a);
    } else {
        GrabArgs<float> a{(const float *)h->d_pos, h->d_cnt, h->d_xy, radius ? h->d_radius : nullptr,
                          active ? h->d_active : nullptr, h->d_ngrab, h->d_levels, h->n_grab_levels, h->P, h->Ppad, top,
                          h->prm.grip_radius, 2 * h->prm.thickness};
        hipLaunchKernelGGL(k_grab<float>, dim3(h->E), dim3(64), 0, h->stream, a);
    }
    HIPCHECK(hipGetLastError());
    if (n_grabbed) HIPCHECK(hipMemcpyAsync(n_grabbed, h->d_ngrab, (size_t)h->E * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_grab_top(clothhip_handle *h, const double *xy, const double *radius, const uint8_t *active, int32_t *n_grabbed) {
    return do_grab(h, xy, radius, active, n_grabbed, 1);
}
extern "C" int clothhip_grab(clothhip_handle *h, const double *xy, const double *radius, const uint8_t *active, int32_t *n_grabbed) {
    return do_grab(h, xy, radius, active, n_grabbed, 0);
}

extern "C" int clothhip_release(clothhip_handle *h, const uint8_t *active) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    HIPCHECK(hipSetDevice(h->device));
    if (active) HIPCHECK(hipMemcpyAsync(h->d_active, active, (size_t)h->E, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_release, dim3(h->E), dim3(64), 0, h->stream, h->d_cnt, active ? h->d_active : nullptr, h->Ppad);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_pin_points(clothhip_handle *h, int32_t env, const int32_t *idx, int32_t n) {
    if (int rc = check_range(h, env, 1)) return rc;
    if (n < 0 || (n > 0 && !idx)) return fail(CLOTHHIP_EINVAL, "bad idx/n");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    std::vector<uint8_t> c(h->Ppad);
    HIPCHECK(hipMemcpy(c.data(), h->d_cnt + (size_t)env * h->Ppad, c.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= h->P) return fail(CLOTHHIP_EINVAL, "point index %d outside [0,%d)", idx[k], h->P);
        c[idx[k]] |= CNT_EXT_PIN;
    }
    HIPCHECK(hipMemcpy(h->d_cnt + (size_t)env * h->Ppad, c.data(), c.size(), hipMemcpyHostToDevice));
    return 0;
}

template <typename T> static StepArgs<T> make_args(clothhip_handle *h, const ClothSchedule *d_sched) {
    StepArgs<T> a;
    a.e0 = 0;
    a.pos = (T *)h->d_pos; a.prev = (T *)h->d_prev; a.cnt = h->d_cnt; a.rest = (const T *)h->d_rest;
    a.tear = h->d_tear; a.executed = h->d_exec; a.stats = h->d_stats; a.sched = d_sched;
    a.gather = h->d_gather; a.wt_ent = h->d_wt_ent; a.wt_dep = h->d_wt_dep; a.nW = h->wt.nW; a.wt_rshift = h->wt.reach_shift; a.cell_copy = h->cell_copy;
    a.N = h->N; a.P = h->P; a.Ppad = h->Ppad; a.S = h->S; a.Spad = h->Spad;
    a.HT = h->HT; a.ht_bits = h->ht_bits;
    a.rest_stride = h->rest_stride; a.phase_mask = h->phase_mask;
    a.k = make_consts<T>(h->prm);
    if (sizeof(T) == 8) { a.pal_struct = (T)h->pal64[SPRING_STRUCTURAL]; a.pal_shear = (T)h->pal64[SPRING_SHEARING]; a.pal_bend = (T)h->pal64[SPRING_BENDING]; }
    else { a.pal_struct = (T)h->pal[SPRING_STRUCTURAL]; a.pal_shear = (T)h->pal[SPRING_SHEARING]; a.pal_bend = (T)h->pal[SPRING_BENDING]; }
    a.lstc = h->d_lstc;
    a.fz = nullptr;
    return a;
}

// Which stepper runs the next launch: the LEAN variant (three cloths per CU) when this handle wants it and the device's shared
// rest table is a three-value palette (re-checked whenever the table may have changed: per-env tables, i.e. tier 2, or odd
// rest lengths uploaded by the caller switch back), else the standard variant. LDS is rebuilt by every launch, so the layout
// may change from one launch to the next.
// Which grid-specialised kernel (k_run_schedule<..., NS>, NS = 25 or 50) may run the layout the handle's fields describe NOW -- 0: none, the generic
// build. Only if the variant is one of the specialised ones (stepper_variants.hpp: CLOTH_SPEC_*) AND every constant that build has compiled in
// (cloth_common.hpp: spec_*) is what this handle computed: grid, window table, hash-table size, whether the cell-ordered copy exists, all phases
// on (debug masks take the generic build, as does CLOTHHIP_DEBUG_NOSPEC=1 -- the A/B and the bit-identity test of the two).
// (with_palette false: clothhip_create, which prepares every kernel the handle may launch before any rest table has been read back)
static int spec_ns(const clothhip_handle *h, bool with_palette) {
    if (getenv("CLOTHHIP_DEBUG_NOSPEC") && atoi(getenv("CLOTHHIP_DEBUG_NOSPEC"))) return 0;
    if (h->phase_mask != 15 || (h->N != 25 && h->N != 50)) return 0;
    const int ns = h->N;
    {   // the physics constants the build has compiled in (cloth_common.hpp: spec_phys) must be this handle's
        const SpecPhys q = spec_phys(ns);
        const ClothParams &p = h->prm;
        if (p.width != q.width || p.height != q.height || p.density != q.density || p.ks != q.ks || p.damping != q.damping || p.thickness != q.thickness ||
            p.plane_friction != q.plane_friction || p.tear_thresh != q.tear_thresh || p.gravity != q.gravity || p.minimum_z != q.minimum_z ||
            p.frames_per_sec != q.frames_per_sec || p.simulation_steps != q.simulation_steps) return 0;
        // (belt and braces: the literals the kernel holds are what make_consts gives the generic build, bit for bit)
        if (h->precision == CLOTHHIP_F32) { const DevConsts<float> a = make_consts<float>(p), b = spec_consts<float>(ns); if (memcmp(&a, &b, sizeof(a)) != 0) return 0; }
        else { const DevConsts<double> a = make_consts<double>(p), b = spec_consts<double>(ns); if (memcmp(&a, &b, sizeof(a)) != 0) return 0; }
    }
    bool listed = false;
#define XS(T_, NT, PPT, TAB, RR, NS_) \
    if (NS_ == ns && (sizeof(T_) == 4) == (h->precision == CLOTHHIP_F32) && h->nt == NT && h->ppt == PPT && h->tab == TAB && h->rest_reg == RR) listed = true;
    CLOTH_SPEC_F32(XS) CLOTH_SPEC_F64(XS)
#undef XS
    if (!listed) return 0;
    const bool same = h->P == spec_p(ns) && h->Ppad == spec_ppad(ns) && h->HT == spec_ht(ns, h->tab) && h->ht_bits == spec_htbits(ns, h->tab) &&
                      h->Spad == spec_spad(ns) && h->wt.nW == spec_nw(ns) && h->wt.reach_shift == spec_rshift(ns) && h->cell_copy == spec_cell_copy(ns, h->tab);
    if (!same) return 0;
    // the LEAN fp32 builds hold the rest-length palette as literals: it must be what lean_refresh read back from the device's table
    if (with_palette && h->precision == CLOTHHIP_F32 && h->rest_reg) {
        for (int t = 0; t < 3; t++) { const float v = spec_pal(ns, t); if (memcmp(&v, &h->pal[t], 4) != 0) return 0; }
    }
    return ns;
}

static int lean_refresh(clothhip_handle *h) {
    if (!h->lean) { h->spec_now = spec_ns(h); return 0; }
    if (h->lean_dirty) {
        h->lean_dirty = false; h->lean_ok = false;
        if (h->rest_stride == 0 && h->precision == CLOTHHIP_F64) {
            // fp64: every spring's rest length must be its type's smallest value + at most 255 ulps (the flat tiers: <= 46 at 50x50); the offsets go to
            // the per-particle stencil table, slot k of particle i = its k-th stencil position (lean_off), i.e. the popcount(valid below k)-th gather entry
            std::vector<double> r((size_t)h->Spad);
            HIPCHECK(hipStreamSynchronize(h->stream));
            HIPCHECK(hipMemcpy(r.data(), h->d_rest, r.size() * 8, hipMemcpyDeviceToHost));
            long long base[3] = {0, 0, 0}; bool have[3] = {false, false, false}, ok = true;
            auto bits = [](double v) { long long b; memcpy(&b, &v, 8); return b; };
            for (int sp = 0; sp < h->S; sp++) {
                const int ty = h->topo.type[sp];
                const double v = r[h->wt.slot_of[sp]];
                if (!(v > 0.0) || !std::isfinite(v)) { ok = false; break; }
                if (!have[ty] || bits(v) < base[ty]) { base[ty] = bits(v); have[ty] = true; }
            }
            ok = ok && have[0] && have[1] && have[2];
            std::vector<uint32_t> tab((size_t)h->Ppad * 4, 0u);
            std::vector<uint32_t> gather = build_gather(h->topo, h->wt, h->Ppad);
            for (int i = 0; i < h->P && ok; i++) {
                const uint32_t vm = lean_valid_mask(i / h->N, i % h->N, h->N);
                tab[(size_t)4 * i] = vm;
                int slot = 0;
                for (int k = 0; k < HK_SLOTS; k++) {
                    if (!((vm >> k) & 1u)) continue;
                    const uint32_t g = gather[(size_t)slot * h->Ppad + i];
                    const int pos = (int)((g >> HK_POS_SHIFT) & HK_POS_MASK);
                    const int sp = h->wt.spring_at[pos];
                    const long long off = sp >= 0 ? bits(r[pos]) - base[h->topo.type[sp]] : -1;
                    if (off < 0 || off > 255) { ok = false; break; }
                    tab[(size_t)4 * i + 1 + (k >> 2)] |= (uint32_t)off << (8 * (k & 3));
                    slot++;
                }
            }
            if (ok) {
                for (int t = 0; t < 3; t++) memcpy(&h->pal64[t], &base[t], 8);
                HIPCHECK(hipMemcpy(h->d_lstc, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            }
            h->lean_ok = ok;
        } else
        if (h->rest_stride == 0) {
            std::vector<float> r((size_t)h->Spad);
            HIPCHECK(hipStreamSynchronize(h->stream));
            HIPCHECK(hipMemcpy(r.data(), h->d_rest, r.size() * 4, hipMemcpyDeviceToHost));
            bool have[3] = {false, false, false}, ok = true;
            for (int sp = 0; sp < h->S && ok; sp++) {
                const int ty = h->topo.type[sp];
                const float v = r[h->wt.slot_of[sp]];
                if (!have[ty]) { h->pal[ty] = v; have[ty] = true; }
                else if (memcmp(&h->pal[ty], &v, 4) != 0) ok = false;
            }
            h->lean_ok = ok && have[0] && have[1] && have[2];
        }
    }
    const clothhip_handle::Layout &L = (h->lean_ok && h->rest_stride == 0) ? h->lay_lean : h->lay_std;
    h->nt = L.nt; h->ppt = L.ppt; h->tab = L.tab; h->rest_reg = L.rest_reg; h->cell_copy = L.cell_copy; h->lds_bytes = L.lds_bytes;
    h->HT = L.HT; h->ht_bits = L.ht_bits;
    h->spec_now = spec_ns(h);
    return 0;
}

// (the compile-time variants -- CLOTH_VARIANTS, CLOTH_VARIANTS_LEAN -- and the object file each is compiled in: stepper_variants.hpp)
template <typename T, int FUSED> static const void *stepper_fn_t(const clothhip_handle *h) {
#define X(T_, NT, PPT, TAB, RR) \
    if (h->nt == NT && h->ppt == PPT && h->tab == TAB && h->rest_reg == RR) return (const void *)k_run_schedule<T_, NT, PPT, TAB, RR, FUSED>;
    if (h->spec_now) {
#define XS(T_, NT, PPT, TAB, RR, NS_) \
        if constexpr (sizeof(T_) == sizeof(T)) { if (h->spec_now == NS_ && h->nt == NT && h->ppt == PPT && h->tab == TAB && h->rest_reg == RR) return (const void *)k_run_schedule<T, NT, PPT, TAB, RR, FUSED, NS_>; }
        CLOTH_SPEC_F32(XS) CLOTH_SPEC_F64(XS)
#undef XS
    }
    CLOTH_VARIANTS(X, T)
    if constexpr (sizeof(T) == 4) { CLOTH_VARIANTS_LEAN(X, T) } else { CLOTH_VARIANTS_LEAN64(X, T) }
#undef X
    return nullptr;
}
static const void *stepper_fn(const clothhip_handle *h, int fused) {
    if (fused == 2) return h->precision == CLOTHHIP_F64 ? stepper_fn_t<double, 2>(h) : stepper_fn_t<float, 2>(h);
    if (fused == 1) return h->precision == CLOTHHIP_F64 ? stepper_fn_t<double, 1>(h) : stepper_fn_t<float, 1>(h);
    return h->precision == CLOTHHIP_F64 ? stepper_fn_t<double, 0>(h) : stepper_fn_t<float, 0>(h);
}

// resident workgroups per CU of a stepper kernel at the handle's LDS footprint (for clothhip_last_variant): asked once per (kernel,
// LDS bytes), not on every launch -- the step mode launches once per env step
static int cached_occupancy(clothhip_handle *h, const void *fn, int nt) {
    for (auto &c : h->occ_cache) if (c.fn == fn && c.lds == h->lds_bytes) return c.occ;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, nt, (size_t)h->lds_bytes) != hipSuccess) { (void)hipGetLastError(); occ = 0; }
    for (auto &c : h->occ_cache) if (c.fn == nullptr) { c = {fn, h->lds_bytes, occ}; return occ; }
    h->occ_cache[0] = {fn, h->lds_bytes, occ};
    return occ;
}

// (the caller has run lean_refresh(h) -- which of the handle's two layouts may run now -- BEFORE recording its start event)
// `by_generation` (the time-sliced episode launches): every workgroup runs for the same time slice, counted from its own start, so a batch of
// more cloths than are resident runs in generations -- which go out as ONE LAUNCH EACH, in stream order. Left to the hardware's
// dispatcher the generations of a single launch change hands on every CU within a few dozen microseconds, and now and then a CU
// that has just lost both of its workgroups takes only one new one for the whole slice (measured on 1 024 cloths of 50x50, two per
// CU at 79.9 KB of LDS and 4 x 128 VGPRs per SIMD: in 3 launches of 8 one workgroup of the 1 024 started only when the second
// generation had ended, 2 400 instead of 1 600 ms -- tools/placement.py, profiles/r05_placement.txt). A fresh launch finds every CU empty.
template <typename T, int FUSED> static void launch_run(clothhip_handle *h, const ClothSchedule *d_sched, const void *d_fz, bool by_generation = false) {
    StepArgs<T> a = make_args<T>(h, d_sched);
    a.fz = (const FusedArgs<T> *)d_fz;
    a.e0 = 0;
#define XN(T_, NT, PPT, TAB, RR, NS_)                                                                   \
    if (h->nt == NT && h->ppt == PPT && h->tab == TAB && h->rest_reg == RR) {                           \
        const int occ_ = cached_occupancy(h, (const void *)k_run_schedule<T_, NT, PPT, TAB, RR, FUSED, NS_>, NT);    \
        const int cap_ = by_generation && occ_ > 0 && h->n_cus > 0 && !getenv("CLOTHHIP_DEBUG_ONE_LAUNCH") ? occ_ * h->n_cus : h->E;   \
        h->last_dispatches = 0;                                                                         \
        for (int e0_ = 0; e0_ < h->E; e0_ += cap_) {                                                    \
            a.e0 = e0_; h->last_dispatches++;                                                           \
            hipLaunchKernelGGL((k_run_schedule<T_, NT, PPT, TAB, RR, FUSED, NS_>), dim3(std::min(cap_, h->E - e0_)), dim3(NT), h->lds_bytes, h->stream, a); \
        }                                                                                               \
        const int32_t v_[10] = {NT, PPT, TAB, RR ? 1 : 0, v_lean(TAB, RR, (int)sizeof(T_)) ? 1 : 0, FUSED, h->lds_bytes, occ_, h->n_cus, sizeof(T_) == 4 ? 1 : 0}; \
        memcpy(h->last_variant, v_, sizeof(v_)); h->have_variant = true; h->last_spec = NS_;            \
        return;                                                                                         \
    }
#define X(T_, NT, PPT, TAB, RR) XN(T_, NT, PPT, TAB, RR, 0)
#define XS(T_, NT, PPT, TAB, RR, NS_) if constexpr (sizeof(T_) == sizeof(T)) { if (h->spec_now == NS_) { XN(T, NT, PPT, TAB, RR, NS_) } }
    if (h->spec_now) { CLOTH_SPEC_F32(XS) CLOTH_SPEC_F64(XS) }
    CLOTH_VARIANTS(X, T)
    if constexpr (sizeof(T) == 4) { CLOTH_VARIANTS_LEAN(X, T) } else { CLOTH_VARIANTS_LEAN64(X, T) }
#undef X
#undef XS
#undef XN
}

// The relaxed-order companion (k_run_schedule<float, 512, 2, 2, true, 3>: Jacobi self-collision, coloured strain limit): ONE instantiation,
// the headline variant's layout. Results differ from the reference's by construction -- a labelled measurement of what the exact order
// costs (bench.py's companion record "exact_order": false), never a product path.
static void launch_relaxed(clothhip_handle *h, const void *d_fz) {
    StepArgs<float> a = make_args<float>(h, h->d_sched);
    a.fz = (const FusedArgs<float> *)d_fz;
    hipLaunchKernelGGL((k_run_schedule<float, 512, 2, 2, true, 3>), dim3(h->E), dim3(512), h->lds_bytes, h->stream, a);
    h->last_dispatches = 1; h->last_spec = 0;
    const int occ_ = cached_occupancy(h, (const void *)k_run_schedule<float, 512, 2, 2, true, 3>, 512);
    const int32_t v_[10] = {512, 2, 2, 1, 1, 3, h->lds_bytes, occ_, h->n_cus, 1};
    memcpy(h->last_variant, v_, sizeof(v_)); h->have_variant = true;
}

static int run_common(clothhip_handle *h, const ClothSchedule *d_sched) {
    if (int rc = drop_in_flight(h, nullptr, d_sched)) return rc;
    if (int rc = lean_refresh(h)) return rc;         // (may synchronise and read the rest table back: outside the timed events)
    HIPCHECK(hipEventRecord(h->ev0, h->stream));
    if (h->precision == CLOTHHIP_F64) launch_run<double, 0>(h, d_sched, nullptr);
    else launch_run<float, 0>(h, d_sched, nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(h->ev1, h->stream));
    h->have_timing = true;
    h->pending_exec = true;
    return 0;
}

extern "C" int clothhip_run_async(clothhip_handle *h, const ClothSchedule *sched) {
    if (!h || !sched) return fail(CLOTHHIP_EINVAL, "NULL argument");
    for (int e = 0; e < h->E; e++) {
        const ClothSchedule &s = sched[e];
        if (s.n_total < 0 || s.n_up_end < 0 || s.n_uprest_end < s.n_up_end || s.n_pull_end < s.n_uprest_end ||
            s.n_griprest_end < s.n_pull_end || s.n_total < s.n_griprest_end)
            return fail(CLOTHHIP_EINVAL, "env %d: phase boundaries must be non-decreasing and <= n_total", e);
    }
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));      // h_sched may still be in flight from a previous _async call
    memcpy(h->h_sched, sched, (size_t)h->E * sizeof(ClothSchedule));
    HIPCHECK(hipMemcpyAsync(h->d_sched, h->h_sched, (size_t)h->E * sizeof(ClothSchedule), hipMemcpyHostToDevice, h->stream));
    return run_common(h, h->d_sched);
}

extern "C" int clothhip_run_device_sched_async(clothhip_handle *h, const void *d_sched) {
    if (!h || !d_sched) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    return run_common(h, (const ClothSchedule *)d_sched);
}

extern "C" int clothhip_sync(clothhip_handle *h, int32_t *executed) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    HIPCHECK(hipSetDevice(h->device));
    if (executed && h->pending_exec)
        HIPCHECK(hipMemcpyAsync(executed, h->d_exec, (size_t)h->E * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_run(clothhip_handle *h, const ClothSchedule *sched, int32_t *executed) {
    if (int rc = clothhip_run_async(h, sched)) return rc;
    return clothhip_sync(h, executed);
}


// ---- whole episodes on the device ---------------------------------------------------------------------------------
template <typename T> static void fill_fused(clothhip_handle *h, FusedArgs<T> &f, const ClothEpisodeParams *ep, int T_, int policy,
                                             const double *d_actions, bool have_parg, bool have_scripts, bool have_resets, bool have_obs,
                                             bool have_robs, int n_scripts, uint64_t budget_ticks, bool have_mt, int rng_tier,
                                             uint64_t domrand_words, int NS, int NH) {
    memset(&f, 0, sizeof(f));
    f.nT = T_; f.policy = policy; f.NS = NS; f.NH = NH;
    f.actions = d_actions;
    f.policy_arg = have_parg ? h->d_fparg : nullptr;
    f.scripts = have_scripts ? (const ClothResetScript *)h->d_fscr : nullptr;
    f.num_steps = h->d_fsteps; f.done = h->d_fdone;
    f.records = (ClothStepRecord *)h->d_frec;
    f.resets = have_resets ? (ClothResetRecord *)h->d_frst : nullptr;
    f.obs = have_obs ? (float *)h->d_fobs : nullptr;
    f.reset_obs = have_robs ? (float *)h->d_frobs : nullptr;
    f.flat = (const T *)h->d_flat;
    f.wt_ent = h->d_wt_ent;
    f.rest = (const T *)h->d_rest; f.rest_rw = (T *)h->d_rest; f.rest_stride = h->rest_stride;
    f.grid_dx = h->prm.width * 1.0 / (h->N - 1); f.grid_dy = h->prm.height * 1.0 / (h->N - 1);
    f.levels = h->d_levels; f.n_glevels = h->n_grab_levels; f.E = h->E; f.n_scripts = n_scripts; f.budget_ticks = budget_ticks;
    f.resume = h->d_resume;
    f.op_ticks = h->d_fticks;
    f.summary = h->d_fsum;
    f.mt = have_mt ? h->d_fmt : nullptr; f.rng_tier = rng_tier; f.domrand_words = domrand_words;
    f.two_thickness = 2 * h->prm.thickness; f.half_thickness = h->prm.thickness / 2.0;
    f.ep = *ep;
}

static int grow(void **p, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    if (*p) HIPCHECK(hipFree(*p));
    *p = nullptr; *cap = 0;
    HIPCHECK(hipMalloc(p, need));
    *cap = need;
    return 0;
}

// (of the layout the handle's fields describe NOW: call lean_refresh first -- the LEAN and the standard layout differ in hash-table
//  size, cell copy and hull-stack format)
static int fused_scratch(const clothhip_handle *h, int *need_out) {
    int NS = 1; while (NS < h->P) NS <<= 1;
    const int NH = h->Ppad + 8;
    *need_out = metrics_scratch_bytes(NS, NH, (int)h->tsz, v_hull_idx(h->tab, (int)h->tsz, h->nt, h->ppt));
    const LdsLayout lay((int)h->tsz, h->Ppad, h->Spad, h->HT, h->tab == 2 ? 2 : (v_ldstab(h->tab) ? 1 : 0), h->cell_copy);
    return h->lds_bytes - lay.hkey;
}

// Every layout the handle may run (the standard one always; the LEAN one while its palette holds) was sized in clothhip_create
extern "C" int clothhip_fused_supported(const clothhip_handle *h) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    const bool std_ok = h->lay_std.scratch_have >= h->lay_std.scratch_need;
    const bool lean_ok = !h->lean || h->lay_lean.scratch_have >= h->lay_lean.scratch_need;
    return std_ok && lean_ok ? 1 : 0;
}

extern "C" int clothhip_selftest_layout(const ClothParams *p, int32_t precision, int32_t n_envs, int32_t n_cus, int32_t *out, int32_t capacity) {
    if (int rc = check_params(p)) return rc;
    if (!out || capacity < 24) return fail(CLOTHHIP_EINVAL, "out needs 24 entries");
    if (precision != CLOTHHIP_F64 && precision != CLOTHHIP_F32) return fail(CLOTHHIP_EINVAL, "precision must be 0 (f64) or 1 (f32)");
    if (n_envs < 1 || n_cus < 1) return fail(CLOTHHIP_EINVAL, "n_envs and n_cus must be >= 1");
    clothhip_handle h;                                       // host fields only: nothing here touches a device
    h.prm = *p; h.E = n_envs; h.precision = precision;
    h.N = p->n_side; h.P = h.N * h.N; h.Ppad = (h.P + 63) / 64 * 64;
    h.tsz = precision == CLOTHHIP_F64 ? 8 : 4;
    h.topo = build_topology(h.N);
    h.wt = build_windows(h.topo, build_levels(h.topo));
    h.S = h.topo.S; h.Spad = h.wt.n_slots;
    plan_layouts(&h, n_cus, build_gather(h.topo, h.wt, h.Ppad));
    auto put = [&](int o, const clothhip_handle::Layout &L) {
        out[o] = L.nt; out[o + 1] = L.ppt; out[o + 2] = L.tab; out[o + 3] = L.rest_reg ? 1 : 0; out[o + 4] = L.cell_copy;
        out[o + 5] = L.lds_bytes; out[o + 6] = L.HT; out[o + 7] = L.scratch_have; out[o + 8] = L.scratch_need;
        out[o + 9] = L.scratch_have >= L.scratch_need ? 1 : 0;
    };
    put(0, h.lay_std);
    out[10] = h.lean ? 1 : 0; out[11] = h.lean_r;
    put(12, h.lay_lean);
    out[22] = clothhip_fused_supported(&h); out[23] = h.lds_bytes <= 160 * 1024 ? 1 : 0;
    return 0;
}

extern "C" int clothhip_run_actions_begin(clothhip_handle *h, const ClothEpisodeParams *ep, int32_t T_, int32_t policy,
                                          const double *actions, int32_t actions_on_device, const int32_t *policy_arg,
                                          const ClothResetScript *scripts, int32_t n_scripts, const int32_t *num_steps,
                                          const uint8_t *done, const uint32_t *rng_states, int32_t rng_tier, uint64_t domrand_words,
                                          int32_t want_resets, int32_t want_obs, int32_t want_reset_obs,
                                          double time_budget_ms) {
    if (!h || !ep || !num_steps || !done) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (h->f_pending) return fail(CLOTHHIP_ESTATE, "a clothhip_run_actions_begin is already in flight");
    const bool resets = want_resets != 0, obs = want_obs != 0, reset_obs = want_reset_obs != 0;
    if (T_ < 1 || T_ > 4096) return fail(CLOTHHIP_EINVAL, "T must be in [1, 4096]");
    if (policy != CLOTHHIP_POLICY_TABLE && policy != CLOTHHIP_POLICY_ORACLE_CORNER && policy != CLOTHHIP_POLICY_HIGHEST_POINT)
        return fail(CLOTHHIP_EINVAL, "unknown policy %d", policy);
    if (policy == CLOTHHIP_POLICY_HIGHEST_POINT && !policy_arg)
        return fail(CLOTHHIP_EINVAL, "the highest-point policy needs policy_arg[1 + T][E] (construction codes + which of the highest points per slot)");
    if (policy == CLOTHHIP_POLICY_TABLE && !actions) return fail(CLOTHHIP_EINVAL, "the table policy needs actions[T][E][4]");
    if (policy == CLOTHHIP_POLICY_ORACLE_CORNER && h->N != 25)
        return fail(CLOTHHIP_ESTATE, "the oracle-corner policy is defined for 25x25 cloths only (analytic.py:106)");
    if ((scripts || rng_states) && (n_scripts < 1 || n_scripts > 255)) return fail(CLOTHHIP_EINVAL, "n_scripts must be in [1, 255]");
    if (scripts && rng_states) return fail(CLOTHHIP_EINVAL, "resets come either from scripts or from the device-side RNG streams, not both");
    if (rng_states && (rng_tier < 1 || rng_tier > 3)) return fail(CLOTHHIP_EINVAL, "rng_tier must be 1, 2 or 3");
    if (!scripts && !rng_states) n_scripts = 0;
    const bool tier2 = rng_states && rng_tier == 2;
    if ((scripts || rng_states) && !tier2 && h->rest_stride != 0)
        return fail(CLOTHHIP_ESTATE, "in-kernel resets of the flat tiers need the shared flat rest table; this handle has per-env rest lengths");
    if (tier2 && h->rest_stride == 0)
        return fail(CLOTHHIP_ESTATE, "in-kernel tier-2 resets rebuild per-env rest lengths; upload per-env rest tables first (clothhip_set_state without CLOTHHIP_REST_SHARED)");
    if (tier2 && (size_t)3 * h->P * 8 > (size_t)160 * 1024) return fail(CLOTHHIP_ESTATE, "grid too large for the tier-2 reset scratch");
    if (!(ep->reduce_factor > 0) || ep->max_actions < 1) return fail(CLOTHHIP_EINVAL, "bad episode parameters");
    int NS = 1; while (NS < h->P) NS <<= 1;
    const int NH = h->Ppad + 8;
    HIPCHECK(hipSetDevice(h->device));
    // which of the handle's two layouts runs now (may synchronise and read the rest table back: long before the timed events) -- the
    // scratch check below is against THAT layout, not the previous launch's
    if (int rc = lean_refresh(h)) return rc;
    int need = 0;
    const int have = fused_scratch(h, &need);
    if (have < need)
        return fail(CLOTHHIP_ESTATE, "n_side %d: the in-kernel metrics need %d B of LDS scratch, this variant has %d", h->N, need, have);
    HIPCHECK(hipStreamSynchronize(h->stream));
    const size_t E = h->E, nrec = (size_t)T_ * E;
    if (!h->d_fz) {
        HIPCHECK(hipMalloc(&h->d_fz, 1024));
        HIPCHECK(hipMalloc(&h->d_fsteps, E * 4));
        HIPCHECK(hipMalloc(&h->d_fdone, E));
        HIPCHECK(hipMalloc(&h->d_fticks, E * 64));
        HIPCHECK(hipMalloc(&h->d_fsum, E * 32));
    }
    HIPCHECK(hipMemsetAsync(h->d_fticks, 0, E * 64, h->stream));
    if (int rc = grow(&h->d_frec, &h->cap_frec, nrec * sizeof(ClothStepRecord))) return rc;
    const size_t nscr = E * (size_t)(n_scripts > 0 ? n_scripts : 1);
    if (int rc = grow(&h->d_fscr, &h->cap_fscr, nscr * sizeof(ClothResetScript))) return rc;
    if (int rc = grow(&h->d_frst, &h->cap_frst, nscr * sizeof(ClothResetRecord))) return rc;
    const double *d_actions = nullptr;
    if (policy == CLOTHHIP_POLICY_TABLE) {
        if (actions_on_device) d_actions = actions;
        else {
            if (int rc = grow(&h->d_fact, &h->cap_fact, nrec * 4 * 8)) return rc;
            HIPCHECK(hipMemcpyAsync(h->d_fact, actions, nrec * 4 * 8, hipMemcpyHostToDevice, h->stream));
            d_actions = (const double *)h->d_fact;
        }
    }
    if (obs) if (int rc = grow(&h->d_fobs, &h->cap_fobs, nrec * 3 * h->P * 4)) return rc;
    if ((reset_obs || resets) && !scripts && !rng_states) return fail(CLOTHHIP_EINVAL, "reset outputs without a reset source");
    if (reset_obs) {
        if (int rc = grow(&h->d_frobs, &h->cap_frobs, nscr * 3 * h->P * 4)) return rc;
        HIPCHECK(hipMemsetAsync(h->d_frobs, 0, nscr * 3 * h->P * 4, h->stream));
    }
    if (rng_states) {
        if (!h->d_fmt) HIPCHECK(hipMalloc(&h->d_fmt, E * MT_WORDS * 4));
        HIPCHECK(hipMemcpyAsync(h->d_fmt, rng_states, E * MT_WORDS * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (policy_arg) {
        const size_t nb = (policy == CLOTHHIP_POLICY_HIGHEST_POINT ? (size_t)(1 + T_) : (size_t)1) * E * 4;
        if (int rc = grow((void **)&h->d_fparg, &h->cap_fparg, nb)) return rc;
        HIPCHECK(hipMemcpyAsync(h->d_fparg, policy_arg, nb, hipMemcpyHostToDevice, h->stream));
    }
    if (scripts) HIPCHECK(hipMemcpyAsync(h->d_fscr, scripts, nscr * sizeof(ClothResetScript), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->d_fsteps, num_steps, E * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->d_fdone, done, E, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemsetAsync(h->d_frec, 0, nrec * sizeof(ClothStepRecord), h->stream));
    if (resets) HIPCHECK(hipMemsetAsync(h->d_frst, 0, nscr * sizeof(ClothResetRecord), h->stream));
    const uint64_t budget_ticks = time_budget_ms > 0 ? (uint64_t)(time_budget_ms * 1e5) : 0;      // s_memrealtime: 100 MHz
    static_assert(sizeof(FusedArgs<double>) <= 1024 && sizeof(FusedArgs<float>) <= 1024, "fused argument block");
    unsigned char fzbuf[1024];
    if (h->precision == CLOTHHIP_F64)
        fill_fused<double>(h, *reinterpret_cast<FusedArgs<double> *>(fzbuf), ep, T_, policy, d_actions, policy_arg != nullptr, scripts != nullptr, resets, obs, reset_obs, n_scripts, budget_ticks,
                           rng_states != nullptr, rng_tier, domrand_words, NS, NH);
    else
        fill_fused<float>(h, *reinterpret_cast<FusedArgs<float> *>(fzbuf), ep, T_, policy, d_actions, policy_arg != nullptr, scripts != nullptr, resets, obs, reset_obs, n_scripts, budget_ticks,
                           rng_states != nullptr, rng_tier, domrand_words, NS, NH);
    HIPCHECK(hipMemcpyAsync(h->d_fz, fzbuf, 1024, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));          // fzbuf is on this stack frame
    if (h->relaxed && !(h->precision == CLOTHHIP_F32 && h->nt == 512 && h->ppt == 2 && h->tab == 2 && h->rest_reg && h->cell_copy && !tier2 &&
                        policy != CLOTHHIP_POLICY_HIGHEST_POINT))
        return fail(CLOTHHIP_ESTATE, "clothhip_set_relaxed_order: the relaxed-order companion exists for the eight-wave LEAN layout only (fp32, flat tiers, 25x25 class, <= 512 cloths)");
    HIPCHECK(hipEventRecord(h->ev0, h->stream));
    if (h->relaxed) launch_relaxed(h, h->d_fz);
    else
    if (tier2 || policy == CLOTHHIP_POLICY_HIGHEST_POINT) {   // the variant that also carries the tier-2 reset code and the cold policies
        if (h->precision == CLOTHHIP_F64) launch_run<double, 2>(h, h->d_sched, h->d_fz, budget_ticks != 0);
        else launch_run<float, 2>(h, h->d_sched, h->d_fz, budget_ticks != 0);
    } else if (h->precision == CLOTHHIP_F64) launch_run<double, 1>(h, h->d_sched, h->d_fz, budget_ticks != 0);
    else launch_run<float, 1>(h, h->d_sched, h->d_fz, budget_ticks != 0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(h->ev1, h->stream));
    h->have_timing = true;
    h->pending_exec = true;
    h->f_T = T_; h->f_nscr = nscr; h->f_resets = resets; h->f_obs = obs; h->f_robs = reset_obs; h->f_mt = rng_states != nullptr;
    h->f_pending = true;
    return 0;
}

extern "C" int clothhip_run_actions_end(clothhip_handle *h, int32_t *num_steps, uint8_t *done, ClothStepRecord *records,
                                        ClothResetRecord *resets, float *obs, float *reset_obs, uint32_t *rng_states) {
    if (!h || !num_steps || !done || !records) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (!h->f_pending) return fail(CLOTHHIP_ESTATE, "no clothhip_run_actions_begin in flight");
    if ((resets != nullptr) != h->f_resets || (obs != nullptr) != h->f_obs || (reset_obs != nullptr) != h->f_robs)
        return fail(CLOTHHIP_EINVAL, "the output buffers must match the ones announced to clothhip_run_actions_begin");
    if ((rng_states != nullptr) != h->f_mt) return fail(CLOTHHIP_EINVAL, "rng_states must be given to both halves or to neither");
    HIPCHECK(hipSetDevice(h->device));
    const size_t E = h->E, nrec = (size_t)h->f_T * E, nscr = h->f_nscr;
    HIPCHECK(hipMemcpyAsync(records, h->d_frec, nrec * sizeof(ClothStepRecord), hipMemcpyDeviceToHost, h->stream));
    if (resets) HIPCHECK(hipMemcpyAsync(resets, h->d_frst, nscr * sizeof(ClothResetRecord), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipMemcpyAsync(num_steps, h->d_fsteps, E * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipMemcpyAsync(done, h->d_fdone, E, hipMemcpyDeviceToHost, h->stream));
    if (obs) HIPCHECK(hipMemcpyAsync(obs, h->d_fobs, nrec * 3 * h->P * 4, hipMemcpyDeviceToHost, h->stream));
    if (reset_obs) HIPCHECK(hipMemcpyAsync(reset_obs, h->d_frobs, nscr * 3 * h->P * 4, hipMemcpyDeviceToHost, h->stream));
    if (rng_states) HIPCHECK(hipMemcpyAsync(rng_states, h->d_fmt, E * MT_WORDS * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->f_pending = false;
    return 0;
}

extern "C" int clothhip_run_actions_summary(clothhip_handle *h, double *summary, void **d_summary) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    if (!h->d_fsum) return fail(CLOTHHIP_ESTATE, "no clothhip_run_actions launch yet");
    if (d_summary) *d_summary = h->d_fsum;
    if (summary) {
        HIPCHECK(hipSetDevice(h->device));
        HIPCHECK(hipMemcpyAsync(summary, h->d_fsum, (size_t)h->E * 32, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int clothhip_run_actions_op_ticks(clothhip_handle *h, uint64_t *ticks) {
    if (!h || !ticks) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (!h->d_fticks) return fail(CLOTHHIP_ESTATE, "no clothhip_run_actions launch yet");
    if (h->f_pending) return fail(CLOTHHIP_ESTATE, "clothhip_run_actions_begin still in flight: call clothhip_run_actions_end first");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipMemcpyAsync(ticks, h->d_fticks, (size_t)h->E * 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_run_actions(clothhip_handle *h, const ClothEpisodeParams *ep, int32_t T_, int32_t policy,
                                    const double *actions, int32_t actions_on_device, const int32_t *policy_arg,
                                    const ClothResetScript *scripts, int32_t n_scripts, int32_t *num_steps, uint8_t *done,
                                    ClothStepRecord *records, ClothResetRecord *resets, float *obs, float *reset_obs,
                                    double time_budget_ms) {
    if (!records) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (int rc = clothhip_run_actions_begin(h, ep, T_, policy, actions, actions_on_device, policy_arg, scripts, n_scripts,
                                            num_steps, done, nullptr, 0, 0, resets != nullptr, obs != nullptr,
                                            reset_obs != nullptr, time_budget_ms))
        return rc;
    return clothhip_run_actions_end(h, num_steps, done, records, resets, obs, reset_obs, nullptr);
}

extern "C" int clothhip_update(clothhip_handle *h, int32_t n_sub, const double *delta) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    if (n_sub < 0) return fail(CLOTHHIP_EINVAL, "n_sub < 0");
    std::vector<ClothSchedule> s(h->E);
    for (auto &x : s) {
        memset(&x, 0, sizeof(x));
        x.active = 1; x.break_on_tear = 0; x.n_total = n_sub; x.n_griprest_end = n_sub;
        if (delta) {   // n x { adjust(delta) ; update }: the whole run is one "pull" phase
            x.n_pull_end = n_sub;
            x.dx_pull = delta[0]; x.dy_pull = delta[1]; x.dz_pull = delta[2];
        }
    }
    return clothhip_run(h, s.data(), nullptr);
}

// ---- metrics (host, double): cloth_env.py:1020-1098 ------------------------------------------------------
// Convex-hull area by Andrew's monotone chain + shoelace. Collinear and duplicate points (plenty after the
// clip to [0,1]^2) are dropped from the chain; they do not change the area.
extern "C" double clothhip_hull_area(const double *xy, int32_t n) {
    if (!xy || n < 3) return 0.0;
    std::vector<std::pair<double, double>> p(n);
    for (int i = 0; i < n; i++) p[i] = {xy[2 * i], xy[2 * i + 1]};
    std::sort(p.begin(), p.end());
    p.erase(std::unique(p.begin(), p.end()), p.end());
    const int m = (int)p.size();
    if (m < 3) return 0.0;
    auto cross = [](const std::pair<double, double> &o, const std::pair<double, double> &a, const std::pair<double, double> &b) {
        return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
    };
    std::vector<std::pair<double, double>> hull(2 * m);
    int k = 0;
    for (int i = 0; i < m; i++) { while (k >= 2 && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    for (int i = m - 2, t = k + 1; i >= 0; i--) { while (k >= t && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    k--;   // last point == first point
    if (k < 3) return 0.0;
    double a2 = 0.0;
    for (int i = 0; i < k; i++) {
        const auto &u = hull[i], &v = hull[(i + 1) % k];
        a2 += (u.first - hull[0].first) * (v.second - hull[0].second) - (v.first - hull[0].first) * (u.second - hull[0].second);
    }
    return 0.5 * std::fabs(a2);
}

static int launch_metrics(clothhip_handle *h) {
    int NS = 1; while (NS < h->P) NS <<= 1;
    const int NH = h->Ppad + 8;                 // the monotone chain holds at most m + 1 <= P + 1 points
    const int lds = metrics_scratch_bytes(NS, NH, (int)h->tsz);
    const double half_thick = h->prm.thickness / 2.0;                                   // cloth_env.py:604
    if (h->precision == CLOTHHIP_F64) {
        HIPCHECK(hipFuncSetAttribute((const void *)k_metrics<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_metrics<double>, dim3(h->E), dim3(256), lds, h->stream, (const double *)h->d_pos, h->P, h->Ppad, NS, NH, h->d_cov, h->d_vinv, h->d_oob, h->d_hcnt, half_thick);
    } else {
        HIPCHECK(hipFuncSetAttribute((const void *)k_metrics<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_metrics<float>, dim3(h->E), dim3(256), lds, h->stream, (const float *)h->d_pos, h->P, h->Ppad, NS, NH, h->d_cov, h->d_vinv, h->d_oob, h->d_hcnt, half_thick);
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int clothhip_metrics_ex(clothhip_handle *h, double *coverage, double *variance_inv, uint8_t *oob, uint8_t *tear,
                                   int32_t *n_below_half_thickness) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    if (tear) if (int rc = clothhip_get_tear(h, tear)) return rc;
    if (!coverage && !variance_inv && !oob && !n_below_half_thickness) return 0;
    HIPCHECK(hipSetDevice(h->device));
    if (int rc = launch_metrics(h)) return rc;
    if (coverage) HIPCHECK(hipMemcpyAsync(coverage, h->d_cov, (size_t)h->E * 8, hipMemcpyDeviceToHost, h->stream));
    if (variance_inv) HIPCHECK(hipMemcpyAsync(variance_inv, h->d_vinv, (size_t)h->E * 8, hipMemcpyDeviceToHost, h->stream));
    if (oob) HIPCHECK(hipMemcpyAsync(oob, h->d_oob, (size_t)h->E, hipMemcpyDeviceToHost, h->stream));
    if (n_below_half_thickness) HIPCHECK(hipMemcpyAsync(n_below_half_thickness, h->d_hcnt, (size_t)h->E * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_metrics(clothhip_handle *h, double *coverage, double *variance_inv, uint8_t *oob, uint8_t *tear) {
    return clothhip_metrics_ex(h, coverage, variance_inv, oob, tear, nullptr);
}

extern "C" int clothhip_write_obs_f32_device(clothhip_handle *h, void *d_out) {
    if (!h || !d_out) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    if (h->precision == CLOTHHIP_F64)
        hipLaunchKernelGGL(k_write_obs<double>, dim3(h->E), dim3(256), 0, h->stream, (const double *)h->d_pos, (float *)d_out, h->P, h->Ppad);
    else
        hipLaunchKernelGGL(k_write_obs<float>, dim3(h->E), dim3(256), 0, h->stream, (const float *)h->d_pos, (float *)d_out, h->P, h->Ppad);
    HIPCHECK(hipGetLastError());
    return 0;
}

// ---- headless rendering ------------------------------------------------------------------------------------
extern "C" int clothhip_render(clothhip_handle *h, const ClothRenderParams *p, const uint8_t *swap_sides, uint8_t *rgb, float *depth) {
    if (!h || !p) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (p->width < 1 || p->height < 1 || p->width > 4096 || p->height > 4096) return fail(CLOTHHIP_EINVAL, "image size outside [1, 4096]");
    if (!(p->lens_mm > 0) || !(p->sensor_mm > 0)) return fail(CLOTHHIP_EINVAL, "lens / sensor must be > 0");
    if (!rgb && !depth) return 0;
    HIPCHECK(hipSetDevice(h->device));
    const size_t npx = (size_t)p->width * p->height, E = h->E;
    unsigned long long *d_z = nullptr; uint8_t *d_rgb = nullptr, *d_sw = nullptr; float *d_dep = nullptr;
    auto cleanup = [&]() { if (d_z) (void)hipFree(d_z); if (d_rgb) (void)hipFree(d_rgb); if (d_sw) (void)hipFree(d_sw); if (d_dep) (void)hipFree(d_dep); };
#define RC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(CLOTHHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    RC(hipMalloc(&d_z, E * npx * 8));
    if (rgb) RC(hipMalloc(&d_rgb, E * npx * 3));
    if (depth) RC(hipMalloc(&d_dep, E * npx * 4));
    if (swap_sides) { RC(hipMalloc(&d_sw, E)); RC(hipMemcpyAsync(d_sw, swap_sides, E, hipMemcpyHostToDevice, h->stream)); }
    RenderArgs a;
    a.N = h->N; a.P = h->P; a.Ppad = h->Ppad; a.W = p->width; a.H = p->height; a.E = h->E;
    for (int k = 0; k < 9; k++) a.R[k] = p->world_to_cam[k];
    for (int k = 0; k < 3; k++) { a.cam[k] = p->cam_pos[k]; a.front[k] = p->front[k]; a.back[k] = p->back[k]; a.bg[k] = p->background[k]; a.light[k] = p->light_dir[k]; }
    a.fx = (p->lens_mm / p->sensor_mm) * (float)p->width; a.fy = a.fx;           // square pixels, horizontal sensor fit
    a.cx = 0.5f * (float)p->width; a.cy = 0.5f * (float)p->height;
    a.ambient = p->ambient; a.energy = p->energy;
    a.swap = d_sw; a.zbuf = d_z; a.rgb = d_rgb; a.depth = d_dep;
    const int lds = 7 * h->Ppad * 4;
    if (h->precision == CLOTHHIP_F64) {
        RC(hipFuncSetAttribute((const void *)k_render<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_render<double>, dim3(h->E), dim3(256), lds, h->stream, (const double *)h->d_pos, a);
    } else {
        RC(hipFuncSetAttribute((const void *)k_render<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_render<float>, dim3(h->E), dim3(256), lds, h->stream, (const float *)h->d_pos, a);
    }
    RC(hipGetLastError());
    if (rgb) RC(hipMemcpyAsync(rgb, d_rgb, E * npx * 3, hipMemcpyDeviceToHost, h->stream));
    if (depth) RC(hipMemcpyAsync(depth, d_dep, E * npx * 4, hipMemcpyDeviceToHost, h->stream));
    RC(hipStreamSynchronize(h->stream));
#undef RC
    cleanup();
    return 0;
}

// ---- raw device buffers on the handle's device (collective staging of the multi-GPU driver) ------------------
extern "C" int clothhip_device_alloc(clothhip_handle *h, uint64_t nbytes, void **d_out) {
    if (!h || !d_out || nbytes == 0) return fail(CLOTHHIP_EINVAL, "bad argument");
    HIPCHECK(hipSetDevice(h->device));
    hipError_t err = hipMalloc(d_out, (size_t)nbytes);
    if (err != hipSuccess) return fail(err == hipErrorOutOfMemory ? CLOTHHIP_ENOMEM : CLOTHHIP_EHIP, "hipMalloc(%llu) failed: %s",
                                       (unsigned long long)nbytes, hipGetErrorString(err));
    return 0;
}
extern "C" int clothhip_device_free(clothhip_handle *h, void *d) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (d) HIPCHECK(hipFree(d));
    return 0;
}
extern "C" int clothhip_device_upload(clothhip_handle *h, void *d_dst, const void *src, uint64_t nbytes) {
    if (!h || !d_dst || !src) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipMemcpyAsync(d_dst, src, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));      // the host buffer is never retained
    return 0;
}
extern "C" int clothhip_device_download(clothhip_handle *h, void *dst, const void *d_src, uint64_t nbytes) {
    if (!h || !dst || !d_src) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipMemcpyAsync(dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int clothhip_debug_stats(clothhip_handle *h, int32_t *stats) {
    if (!h || !stats) return fail(CLOTHHIP_EINVAL, "NULL argument");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(stats, h->d_stats, (size_t)h->E * 64, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int clothhip_set_relaxed_order(clothhip_handle *h, int32_t on) {
    if (!h) return fail(CLOTHHIP_EINVAL, "handle is NULL");
    if (on) {
        HIPCHECK(hipSetDevice(h->device));
        HIPCHECK(hipFuncSetAttribute((const void *)k_run_schedule<float, 512, 2, 2, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    h->relaxed = on != 0;
    return 0;
}

extern "C" int clothhip_last_specialised(clothhip_handle *h, int32_t *n_side) {
    if (!h || !n_side) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (!h->have_variant) return fail(CLOTHHIP_ESTATE, "no stepper launch on this handle yet");
    *n_side = h->last_spec;
    return 0;
}

extern "C" int clothhip_last_dispatches(clothhip_handle *h, int32_t *n) {
    if (!h || !n) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (!h->have_variant) return fail(CLOTHHIP_ESTATE, "no stepper launch on this handle yet");
    *n = h->last_dispatches;
    return 0;
}

extern "C" int clothhip_last_variant(clothhip_handle *h, int32_t v[10]) {
    if (!h || !v) return fail(CLOTHHIP_EINVAL, "NULL argument");
    if (!h->have_variant) return fail(CLOTHHIP_ESTATE, "no stepper launch on this handle yet");
    memcpy(v, h->last_variant, sizeof(h->last_variant));
    return 0;
}

extern "C" double clothhip_last_kernel_ms(clothhip_handle *h) {
    if (!h || !h->have_timing) return -1.0;
    if (hipSetDevice(h->device) != hipSuccess) return -1.0;
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

extern "C" int clothhip_selftest_rng(uint32_t *state, int32_t kind, int32_t n, double a, double b, double c, double *out) {
    if (!state || n < 0 || (n > 0 && !out && kind != 5)) return fail(CLOTHHIP_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = (double)mt_next32(state); break;
        case 1: out[i] = mt_double(state); break;
        case 2: out[i] = mt_uniform(state, a, b); break;
        case 3: out[i] = (double)mt_randint(state, (uint32_t)a); break;
        case 4: out[i] = mt_randval_minabs(state, a, b, c); break;
        default: break;
        }
    }
    if (kind == 5) mt_skip_serial(state, (uint64_t)a);
    return 0;
}

extern "C" int clothhip_selftest_arith(int32_t device, int32_t op, const double *a, const double *b, double *out, int64_t n) {
    if (!a || !out || n <= 0) return fail(CLOTHHIP_EINVAL, "bad argument");
    if (clothhip_device_count() <= 0) return fail(CLOTHHIP_ENODEV, "no HIP device visible");
    HIPCHECK(hipSetDevice(device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHECK(hipMalloc(&da, n * 8));
    HIPCHECK(hipMalloc(&dout, n * 8));
    if (b) { HIPCHECK(hipMalloc(&db, n * 8)); HIPCHECK(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice)); }
    HIPCHECK(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, da, db, dout, (long long)n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(dout); if (db) (void)hipFree(db);
    return 0;
}
