/*
 * clothhip.h -- C ABI of libclothhip.so, the MI355X (gfx950) cloth stepper.
 *
 * Drop-in boundary for the hot path of DanielTakeshi/gym-cloth: the Python object API of the three
 * Cython modules gym_cloth/physics/{cloth,point,gripper}.pyx that ClothEnv imports at
 * gym_cloth/envs/cloth_env.py:25-27.  The reference has no FFI layer of its own; these entry points
 * are what a ctypes binding for that path binds (INTEGRATION.md shows the stub).  Every entry point
 * names the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain C types only; host arrays are caller-owned, passed as pointer + explicit extents, never
 *     retained after the call returns.
 *   - E = n_envs of the handle, P = n_side*n_side points, S = clothhip_num_springs() springs.
 *   - host position arrays are [E][P][3] doubles (x,y,z interleaved) = E stacked Cloth.allpts_arr
 *     (cloth.pyx:395); the device keeps SoA [E][3][Ppad] in the handle's precision.
 *   - every call returns 0 on success or a negative CLOTHHIP_E* code; clothhip_last_error() gives a
 *     thread-local message.  No C++ exception and no abort() crosses the ABI.
 *   - calls on one handle are not re-entrant.  Calls enqueue on the handle's HIP stream and
 *     synchronise before returning unless the name ends in _async.
 *   - there is NO CPU fallback: without a HIP device clothhip_create fails with CLOTHHIP_ENODEV.
 */
#ifndef CLOTHHIP_H
#define CLOTHHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLOTHHIP_ABI_VERSION 7

enum {
    CLOTHHIP_OK = 0,
    CLOTHHIP_EINVAL = -1,   /* bad argument (ValueError / AssertionError in the reference, cloth.pyx:85,91,132) */
    CLOTHHIP_ENODEV = -2,   /* no usable HIP device */
    CLOTHHIP_EHIP = -3,     /* a HIP runtime call failed; message has the hipError string */
    CLOTHHIP_ENOMEM = -4,
    CLOTHHIP_ESTATE = -5    /* call not valid in the handle's current state */
};

enum { CLOTHHIP_F64 = 0, CLOTHHIP_F32 = 1 };

/* clothhip_set_state flags */
enum {
    CLOTHHIP_REST_SHARED = 1,   /* `rest` is ONE [S] table used by all envs of the handle (tiers 1 and 3) */
    CLOTHHIP_KEEP_TEAR = 2      /* do not clear Cloth.cloth_have_tear (it is sticky in the reference, cloth.pyx:272-273):
                                   for position writes into a live cloth (pts[i].x = ..., Gripper.adjust) as opposed to
                                   the Cloth(...) rebuild of a reset */
};

/* Physics constants: the keys Cloth.__init__/Cloth.update read from the cfg dict
 * (cloth.pyx:53-56, :175-186) plus the Cloth() constructor defaults (cloth.pyx:24-26) and the
 * Gripper constructor arguments (gripper.pyx:10; cloth_env.py:752-753). */
typedef struct ClothParams {
    int32_t n_side;            /* cloth.num_width_points == cloth.num_height_points (asserted, cloth.pyx:91) */
    int32_t frames_per_sec;    /* cfg frames_per_sec */
    int32_t simulation_steps;  /* cfg simulation_steps */
    int32_t _pad;
    double width, height;      /* cfg cloth.width / cloth.height */
    double density, ks, damping, thickness, plane_friction, tear_thresh;  /* cfg cloth.* */
    double gravity;            /* Cloth(gravity=-9.8) */
    double minimum_z;          /* Cloth(minimum_z=0)  */
    double grip_radius;        /* cfg env.grip_radius (default Gripper.grip_radius) */
} ClothParams;

/* One pick-and-place schedule = the hot loop of ClothEnv.step + ClothEnv._pull
 * (cloth_env.py:352-367, :472-515):
 *   for i in [0, n_total):
 *       i <  n_up_end       : gripper.adjust(0, 0, dz_up)
 *       i <  n_uprest_end   : -
 *       i <  n_pull_end     : gripper.adjust(dx_pull, dy_pull, dz_pull)   (dz_pull = 0 in the reference)
 *       i <  n_griprest_end : -
 *       else                : gripper.release()
 *       cloth.update()
 *       if break_on_tear and cloth.have_tear: break
 * The n_* are the integer ceilings of the reference's cumulative (possibly fractional: tier 3 draws a
 * float iters_up, cloth_env.py:960) phase boundaries, so `i < boundary` is unchanged.
 * A raw `for _ in range(n): cloth.update()` is {0,0,0,n,n}, break_on_tear=0. */
typedef struct ClothSchedule {
    int32_t n_up_end, n_uprest_end, n_pull_end, n_griprest_end, n_total;
    int32_t break_on_tear;
    int32_t active;            /* 0: this env is skipped entirely (cloth_env.py:490-493 "nothing grabbed") */
    int32_t _pad;
    double dz_up;              /* 0.0025 in the reference (cloth_env.py:359) */
    double dx_pull, dy_pull;   /* x_dir_r, y_dir_r (cloth_env.py:455-456) */
    double dz_pull;            /* 0 in the reference (cloth_env.py:363); lets clothhip_update pass any delta */
} ClothSchedule;

typedef struct clothhip_handle clothhip_handle;

const char *clothhip_last_error(void);
int clothhip_abi_version(void);
/* number of visible HIP devices (0 if none / runtime unusable); never fails */
int clothhip_device_count(void);

/* Replaces Cloth(...) + Gripper(...) construction for a batch of E independent cloths
 * (cloth.pyx:23-167, gripper.pyx:10-21; cloth_env.py:737-753). State starts as the flat tier-1 grid. */
int clothhip_create(const ClothParams *params, int32_t n_envs, int32_t device, int32_t precision,
                    clothhip_handle **out);
int clothhip_destroy(clothhip_handle *h);
int clothhip_num_points(const clothhip_handle *h);
int clothhip_num_springs(const clothhip_handle *h);
int clothhip_num_envs(const clothhip_handle *h);
int clothhip_precision(const clothhip_handle *h);

/* Host-side restatement of the grid/spring construction of Cloth.__init__ (cloth.pyx:92-146, :411-417),
 * in double, for one cloth: fills pos[P][3] and rest[S]. tier 1/3 = flat grid, tier 2 = vertical sheet
 * with the P np_random.rand() draws passed in rand_draws (r-major order). Pure host function. */
int clothhip_init_grid(const ClothParams *params, int32_t tier, int32_t init_side,
                       const double *rand_draws, double *pos, double *rest);
/* Spring list (ptA index, ptB index, type 0/1/2 = STRUCTURAL/SHEARING/BENDING) in reference list order. */
int clothhip_spring_topology(const ClothParams *params, int32_t *a, int32_t *b, uint8_t *type);

/* State upload/download for envs [env0, env0+n): the replacement for writing/reading
 * pts[i].x/.y/.z/.px/.py/.pz/.pinned (point.pyx:34-48; read at cloth_env.py:196-200, :629, :854-937).
 * Any pointer may be NULL (= leave untouched / do not fetch).  rest is [n][S] (Spring.rest_length), or with
 * CLOTHHIP_REST_SHARED in `flags` a single [S] table used by ALL envs of the handle (tier 1/3).
 * pinned != 0 marks the point pinned and a member of gripper.grabbed_pts. set_state clears the tear flag
 * of the envs it touches when `pos` is given, unless CLOTHHIP_KEEP_TEAR is set. */
int clothhip_set_state(clothhip_handle *h, int32_t env0, int32_t n, const double *pos, const double *prev,
                       const uint8_t *pinned, const double *rest, int32_t flags);
int clothhip_get_state(clothhip_handle *h, int32_t env0, int32_t n, double *pos, double *prev,
                       uint8_t *pinned);
/* Spring.rest_length (cloth.pyx:417) of envs [env0, env0+n) in reference list order: rest[n][S]
 * (what the reference's save_state pickle keeps per spring, cloth_env.py:343-350). */
int clothhip_get_rest(clothhip_handle *h, int32_t env0, int32_t n, double *rest);
/* The Cloth(...) rebuild of ClothEnv.reset (cloth_env.py:737-746) for the flat tiers 1 and 3, on the device:
 * every env with mask[e] != 0 (mask NULL = all) gets the flat grid (cloth.pyx:117-130) as position and previous
 * position, nothing pinned, tear flag cleared, flat rest lengths. No host upload. */
int clothhip_reset_flat(clothhip_handle *h, const uint8_t *mask);
/* Cloth.have_tear (cloth.pyx:390-392) for every env: tear[E] (0/1). set: overwrite (Cloth() is rebuilt
 * per reset in the reference, which clears it). */
int clothhip_get_tear(clothhip_handle *h, uint8_t *tear);
int clothhip_set_tear(clothhip_handle *h, const uint8_t *tear);

/* Gripper.grab_top(x, y) (gripper.pyx:23-42) for every active env: xy[E][2]; radius[E] or NULL
 * (= params.grip_radius; cloth_env.py:436-442 mutates it for force_grab); active[E] or NULL (= all).
 * n_grabbed[E] receives the number of points appended to grabbed_pts by THIS call (0 = nothing grabbed). */
int clothhip_grab_top(clothhip_handle *h, const double *xy, const double *radius, const uint8_t *active,
                      int32_t *n_grabbed);
/* Gripper.grab(x, y) (gripper.pyx:44-53) */
int clothhip_grab(clothhip_handle *h, const double *xy, const double *radius, const uint8_t *active,
                  int32_t *n_grabbed);
/* Gripper.release() (gripper.pyx:68-73) */
int clothhip_release(clothhip_handle *h, const uint8_t *active);
/* pt.pinned = True from outside the gripper (point.pyx:48 is a plain writable attribute): idx[n] point
 * indices of env `env`; the points are pinned but NOT members of grabbed_pts (adjust does not move them). */
int clothhip_pin_points(clothhip_handle *h, int32_t env, const int32_t *idx, int32_t n);

/* The hot path. Runs sched[e] for every env e (see ClothSchedule) entirely on the device:
 * n_total x { Gripper.adjust / release ; Cloth.update (cloth.pyx:169-214) ; tear break }.
 * executed[E] (may be NULL) receives the number of update() calls each env performed. */
int clothhip_run(clothhip_handle *h, const ClothSchedule *sched, int32_t *executed);
int clothhip_run_async(clothhip_handle *h, const ClothSchedule *sched);
/* wait for the handle's stream; after an _async call also fetches executed[E] (may be NULL) */
int clothhip_sync(clothhip_handle *h, int32_t *executed);

/* ---- Whole episodes on the device -------------------------------------------------------------------------------
 * clothhip_run_actions runs T consecutive ClothEnv.step calls (cloth_env.py:369-534) for every env in ONE launch, with
 * the particle state resident on the CU: action source (table or scripted policy), the action -> schedule arithmetic
 * (:396-475), Gripper.grab_top (+ force_grab, :434-444), the hot loop (:495-515), the per-action metrics
 * (:1020-1098), the terminal test (:684-715) and -- when a reset script is supplied -- the ClothEnv.reset of an env whose
 * episode ended (:717-987, tiers 1 and 3), i.e. the reference's episode loop `while not done: step` / `env.reset()`
 * (examples/analytic.py:872-882) without a host round trip. Envs never wait for each other inside the launch.
 * Reward and info bookkeeping stay on the host (gym_cloth_amd/envs.py::ClothVecEnv.step_many) and are computed from the
 * records below. */

/* constants of ClothEnv.__init__ / step / _terminal the device needs (cloth_env.py:87-186) */
typedef struct ClothEpisodeParams {
    int32_t max_actions;             /* env.max_actions */
    int32_t iters_up_rest, iters_grip_rest, iters_rest;
    int32_t clip_act_space;          /* env.clip_act_space */
    int32_t force_grab;              /* env.force_grab */
    int32_t _pad[2];
    double iters_up;                 /* env.iters_up (tier 3 overrides it per reset pull with a float, cloth_env.py:960) */
    double reduce_factor, grip_radius;
    double radius_inc;               /* 0.02, cloth_env.py:131 */
    double dz_up;                    /* 0.0025, cloth_env.py:359 */
    double act_low[4], act_high[4];  /* action_space.low / .high (cloth_env.py:162-181) */
    double coverage_done;            /* _REWARD_THRESHOLDS[reward_type] (cloth_env.py:42-50, :706) */
} ClothEpisodeParams;

enum { CLOTHHIP_POLICY_TABLE = 0,          /* actions[t][e][4] given by the caller (any policy run on the host, or random) */
       CLOTHHIP_POLICY_ORACLE_CORNER = 1,  /* examples/analytic.py:105-155, 'distance' method, delta actions, 25x25 only */
       CLOTHHIP_POLICY_HIGHEST_POINT = 2   /* examples/analytic.py:723-808: the k-th highest point (stable order), pulled to where it
                                              sits on the flat cloth; k per slot and env from policy_arg (the reference draws it
                                              with np.random.randint(top_k = 5)) */ };

/* One scripted pull of a reset (cloth_env.py:851-877 tier 1, :959-978 tier 3): the raw RNG draws; everything that
 * depends on the particle state (the picked point's position, _prevent_oob) is evaluated on the device. */
typedef struct ClothResetPull {
    int32_t point;                   /* >= 0: pick at pts[point] (tier 1: np_random.randint(P)); < 0: pick at (x, y) */
    int32_t need_coverage;           /* bit 0: run this pull (and the later ones) only if coverage >= coverage_min (tier 1's 3rd);
                                        bit 1: do not apply _prevent_oob (tier 2's pulls, cloth_env.py:905-947) */
    double x, y;                     /* pick point when point < 0 (tier 3: p0x, p0y) */
    double dx, dy;                   /* drawn deltas BEFORE _prevent_oob (cloth_env.py:834-840, applied on the device) */
    double iters_up;                 /* iters_up of this pull (tier 3: uniform(200, 280); else env.iters_up) */
    double coverage_min;             /* 0.90 (cloth_env.py:866) */
} ClothResetPull;

typedef struct ClothResetScript {
    int32_t valid;                   /* 0: no script in this slot */
    int32_t n_pulls;                 /* <= 3 */
    int32_t settle_after;            /* bare update() calls after the pulls (tier 3: 800, cloth_env.py:980-981) */
    int32_t _pad;
    ClothResetPull pull[3];
} ClothResetScript;

/* what happened in action slot t of env e */
typedef struct ClothStepRecord {
    double action[4];                /* the action taken, as passed to step() (clip space if clip_act_space) */
    double coverage, variance_inv;   /* after the action (cloth_env.py:1075-1098) */
    int32_t executed;                /* update() calls of this action (0: nothing grabbed) */
    int32_t n_grabbed;               /* len(gripper.grabbed_pts) */
    int32_t iters_pull;
    int32_t n_below_half_thickness;  /* compute_height numerator (cloth_env.py:603-609) */
    uint8_t ran;                     /* 0: slot not executed (episode over and no reset script left) */
    uint8_t oob, tear, done;
    uint8_t reset_before;            /* k > 0: the env was reset right before this action, by its k-th script of this launch */
    uint8_t _pad[3];
} ClothStepRecord;

/* one consumed reset script */
typedef struct ClothResetRecord {
    int32_t consumed;                /* 0 no; 1 yes (complete record); 2 cut by the time slice, completed by the next launch */
    int32_t pulls_run;               /* scripted pulls executed (tier 1: 2 or 3) */
    int32_t executed[3];             /* update() calls of each pull */
    int32_t settle_executed;
    int32_t tear;                    /* Cloth.have_tear after the reset */
    int32_t init_side;               /* device-drawn resets: Cloth.init_side (cloth.pyx:75: np_random.rand() > 0.5) */
    double start_coverage, start_variance_inv;   /* cloth_env.py:780-782 */
    double action[3][4];             /* the reset actions in clip space, as the reference passes them to step(initialize=True) */
} ClothResetRecord;

/* T action slots for every env. policy: CLOTHHIP_POLICY_*. actions: [T][E][4] (TABLE), a HOST pointer unless
 * actions_on_device != 0 (a device table, e.g. after an RCCL broadcast). policy_arg: NULL, or how every cloth was built, int32[E]: 0 = the
 * flat tiers, 1 = tier 2 with init_side False (ORACLE_CORNER then swaps its corner indices, analytic.py:108-114), 2 = tier 2
 * with init_side True; a tier-2 reset inside the launch updates it. HIGHEST_POINT needs it and T more rows, int32[1 + T][E]:
 * row 1 + t = which of the highest points (0 = the highest) the env pulls in its t-th slot. scripts: [E][n_scripts] or NULL -- the
 * env's next resets in order. Script k+1 is drawn (by the host) from the RNG state script k leaves when only its
 * unconditional pulls run; if a conditional pull (tier 1's third, cloth_env.py:866) does run, it consumes further draws,
 * the later scripts of that env are void, and the env idles once its next episode ends (the host re-draws them for the
 * next launch). num_steps[E], done[E]: in/out episode state (ClothEnv.num_steps; episode over). records: [T][E].
 * resets: [E][n_scripts] or NULL. obs: [T][E][3P] float32 '1d' observation after each executed slot, or NULL. reset_obs:
 * [E][n_scripts][3P] float32 first observation of each episode started inside the launch (what env.reset() returns), or
 * NULL.
 * time_budget_ms > 0 makes the launch a TIME SLICE: an env starts no further action once the launch has run that long
 * (constant-rate 100 MHz clock), so envs advance at their own pace and the launch does not wait for the env with the most
 * work; the unused slots of an env stay `ran == 0` at the END of its column and the caller passes those actions again in
 * the next launch. The slice cuts operations in the middle (at a substep boundary): the handle keeps what is needed to
 * continue them, the next launch does so first, and the action's record appears in the launch that completes it (slot 0
 * of that env). A reset cut that way has consumed == 2 in this launch's record and its complete record (consumed == 1) in the
 * next one's slot 0; a slice may also end right after a reset (consumed == 1, no record with that reset_before). Any
 * other state-changing call on the handle (set_state, reset_flat, grab*, run*, update) drops the operations in flight.
 * Which launch executes an action never changes its result (envs are independent); only the partition
 * of an env's action sequence into launches depends on timing. 0 = every env executes all T slots.
 * The slice counts from each cloth's own start: a batch of more cloths than the device holds at once runs in generations,
 * each of them a kernel dispatch of its own on the handle's stream (the call is still one call, and
 * clothhip_last_kernel_ms spans all of them).
 * Returns CLOTHHIP_ESTATE when the handle's variant cannot run fused (per-env rest tables with reset scripts,
 * non-25x25 oracle policy, grid too large for the in-kernel metrics). Synchronous. */
/* Resets drawn ON THE DEVICE (the _begin/_end form only): instead of `scripts`, pass rng_states[E][626] = every env's
 * numpy RandomState (get_state(): key[624], pos, one pad word). The kernel then draws each reset exactly as ClothEnv.reset
 * does from np_random (cloth.pyx:75; cloth_env.py:851-877 tier 1 incl. the coverage-conditional third pull; :893-949 +
 * cloth.pyx:94-116 tier 2: the noisy vertical sheet, its rest lengths, 1500 + 500 settling updates, the two corner pulls --
 * the handle must hold per-env rest tables; :959-972 tier 3; rng_tier = 1, 2 or 3), bit for bit numpy's MT19937 / rand / uniform / randint stream (csrc/cloth_rng.hpp), skips
 * domrand_words 32-bit words after each reset (the domain-randomisation draws of cloth_env.py:786-789; 0 = none), and
 * _end returns the advanced states. n_scripts is then the capacity of resets[E][n_scripts] / reset_obs per env; any number
 * of resets per env and launch up to that capacity, no void scripts.
 * The same in two halves, so that the caller can work (e.g. draw the next reset scripts) while the launch runs:
 * _begin uploads the inputs and launches (the host input arrays are not retained), _end waits and downloads. The want_*
 * flags of _begin announce which of the optional output buffers _end will be given. One launch in flight per handle. */
int clothhip_run_actions_begin(clothhip_handle *h, const ClothEpisodeParams *ep, int32_t T, int32_t policy,
                               const double *actions, int32_t actions_on_device, const int32_t *policy_arg,
                               const ClothResetScript *scripts, int32_t n_scripts, const int32_t *num_steps,
                               const uint8_t *done, const uint32_t *rng_states, int32_t rng_tier, uint64_t domrand_words,
                               int32_t want_resets, int32_t want_obs, int32_t want_reset_obs, double time_budget_ms);
int clothhip_run_actions_end(clothhip_handle *h, int32_t *num_steps, uint8_t *done, ClothStepRecord *records,
                             ClothResetRecord *resets, float *obs, float *reset_obs, uint32_t *rng_states);
/* Per-env summary of the last clothhip_run_actions launch, written by the kernel: summary[E][4] = {actions the env executed (slots with ran != 0),
 * 1 if its episode is over, the coverage after its last action or reset of the launch (NaN: it did neither), the Cloth.update()
 * calls of its actions}. `summary` (host, may be NULL) receives a copy; `d_summary` (may be NULL) receives the DEVICE address of the
 * table, valid for the handle's life and ordered on the handle's stream: the multi-GPU driver all-gathers it in place
 * (ncclAllGather) without staging through the host. Call after clothhip_run_actions_begin (the device address) or after _end (the
 * copy). No reference counterpart. */
int clothhip_run_actions_summary(clothhip_handle *h, double *summary, void **d_summary);
/* Where the time of the last clothhip_run_actions launch went, per env: ticks[E][8] = 100 MHz ticks (s_memrealtime) the env's
 * workgroup spent in {0: actions (ClothEnv.step incl. decode, grab_top, metrics), 1: scripted reset pulls (cloth_env.py:851-982) incl.
 * the coverage test of tier 1's third pull, 2: reset settling (bare update() calls), 3: the rest (Cloth() rebuild, idling once its
 * action slots are used up)}, then the Cloth.update() calls executed in each of the four classes. The benchmark derives the
 * action-only rate of SURVEY 8d from it. Call after clothhip_run_actions / _end. No reference counterpart. */
int clothhip_run_actions_op_ticks(clothhip_handle *h, uint64_t *ticks);
/* 1 if this handle's kernel variant has the LDS room for the in-kernel metrics of clothhip_run_actions, else 0 */
int clothhip_fused_supported(const clothhip_handle *h);
int clothhip_run_actions(clothhip_handle *h, const ClothEpisodeParams *ep, int32_t T, int32_t policy,
                         const double *actions, int32_t actions_on_device, const int32_t *policy_arg,
                         const ClothResetScript *scripts, int32_t n_scripts, int32_t *num_steps, uint8_t *done,
                         ClothStepRecord *records, ClothResetRecord *resets, float *obs, float *reset_obs,
                         double time_budget_ms);

/* Convenience: n x Cloth.update() on every env (cloth_env.py:902-903, :948-949, :980-981), optionally
 * preceded each time by Gripper.adjust(delta) when delta != NULL ([3] doubles, same for all envs). */
int clothhip_update(clothhip_handle *h, int32_t n_sub, const double *delta);

/* Per-env quantities ClothEnv derives from the particle positions after every action:
 *   coverage[E]      area of the 2-D convex hull of (clip(x,0,1), clip(y,0,1))  (cloth_env.py:628-638, :1086-1098;
 *                    the reference calls scipy.spatial.ConvexHull(points).volume; 0.0 for a degenerate hull)
 *   variance_inv[E]  1000 if var(z) < 1e-6 else 0.001/var(z)                     (cloth_env.py:1075-1084)
 *   oob[E]           out-of-bounds test with slack 0.25 on x,y and [0,1) on z    (cloth_env.py:1020-1045)
 *   tear[E]          Cloth.have_tear
 * Any pointer may be NULL. */
int clothhip_metrics(clothhip_handle *h, double *coverage, double *variance_inv, uint8_t *oob, uint8_t *tear);
/* the same plus n_below_half_thickness[E] = #points with z < thickness/2, the numerator of the 'height' reward's
 * compute_height (cloth_env.py:603-609) */
int clothhip_metrics_ex(clothhip_handle *h, double *coverage, double *variance_inv, uint8_t *oob, uint8_t *tear,
                        int32_t *n_below_half_thickness);
/* the same hull-area routine on caller-supplied points xy[n][2] (pure host function; used by tests to pin it
 * against scipy's Qhull on the golden states) */
double clothhip_hull_area(const double *xy, int32_t n);

/* Device-resident observation path for the multi-GPU driver: writes the '1d' observation
 * (cloth_env.py:196-200: [x0,y0,z0,x1,...] per env) of every env as float32 into a DEVICE buffer
 * d_out[E][3P] (e.g. a torch/RCCL gather buffer). Asynchronous on the handle's stream. */
int clothhip_write_obs_f32_device(clothhip_handle *h, void *d_out);
/* Same, schedules read from a DEVICE array of ClothSchedule[E] (e.g. after an RCCL broadcast). */
int clothhip_run_device_sched_async(clothhip_handle *h, const void *d_sched);
/* Headless RGB / depth rendering of every env's cloth mesh: what the reference obtains by exporting the
 * particle grid as a triangle mesh (cloth_env.py:218-229) and calling a Blender subprocess
 * (gym_cloth/blender/get_image_rep_279.py; cloth_env.py:212-330). The scene follows that script -- pinhole camera
 * (default: at (0.5, 0.5, 1.45) looking straight down, lens 40 mm on a 36 mm sensor), the two cloth sides in different
 * colours, a shadow-less lamp, a white bed plane -- but the pixels are this library's own rasterisation rules
 * (csrc/cloth_render.hpp), pinned to a numpy restatement (oracle/render_oracle.py), not to Blender. */
typedef struct ClothRenderParams {
    int32_t width, height;           /* cfg: 224 x 224 (cloth_env.py:152-157) */
    float cam_pos[3];                /* 0.5, 0.5, 1.45 (+ dom_rand camera_pos) */
    float world_to_cam[9];           /* row-major rotation; the camera looks along -z_cam with +y_cam up. Identity = straight
                                        down, image +x = world +x, image up = world +y. (Blender's rotation_euler XYZ of
                                        get_image_rep_279.py:119-122 is camera-to-world = Rz Ry Rx: pass its transpose.) */
    float lens_mm, sensor_mm;        /* 40, 36 */
    float front[3], back[3];         /* (0.070, 0.050, 0.600), (0.070, 0.300, 0.900) */
    float background[3];             /* bed plane, (1, 1, 1) */
    float light_dir[3];              /* unit vector towards the lamp */
    float ambient, energy;
} ClothRenderParams;
/* rgb: [E][height][width][3] uint8 or NULL; depth: [E][height][width] float32 camera-space depth (distance along the view
 * axis; the bed plane where no cloth is) or NULL; swap_sides[E] or NULL: != 0 swaps the two side colours (a tier-2 cloth
 * with init_side False, get_image_rep_279.py:235-239). */
int clothhip_render(clothhip_handle *h, const ClothRenderParams *params, const uint8_t *swap_sides, uint8_t *rgb,
                    float *depth);

/* Raw device buffers on the handle's device, for the multi-GPU driver's RCCL staging (action tables in, result /
 * observation tables out; gym_cloth_amd/dist.py). upload/download run on the handle's stream and synchronise it, so
 * they are ordered with the stepper launches. */
int clothhip_device_alloc(clothhip_handle *h, uint64_t nbytes, void **d_out);
int clothhip_device_free(clothhip_handle *h, void *d);
int clothhip_device_upload(clothhip_handle *h, void *d_dst, const void *src, uint64_t nbytes);
int clothhip_device_download(clothhip_handle *h, void *dst, const void *d_src, uint64_t nbytes);
/* the handle's hipStream_t as an opaque pointer (for event timing / stream ordering by the caller) */
void *clothhip_stream(clothhip_handle *h);

/* Timing of the last clothhip_run/_async/_update launch measured with HIP events recorded on the
 * handle's stream around the stepper kernel: milliseconds, or a negative value if none. */
double clothhip_last_kernel_ms(clothhip_handle *h);

/* (new, ABI 5) Which compiled stepper variant the handle's LAST launch ran (clothhip_run*, clothhip_update, clothhip_run_actions*),
 * so that tests and the benchmark can tell the variants apart instead of inferring them from the batch size:
 *   v[0] threads per cloth, v[1] particles per thread, v[2] table mode (1: strain-sweep window table resident in LDS, 0: streamed
 *   from L2; LEAN builds: 0 = compiled for three cloths per CU, -1 = for four, 2 = LEAN arithmetic with the table in LDS),
 *   v[3] rest lengths in registers / LEAN flag as compiled (0/1), v[4] 1 when the LEAN arithmetic ran (gather stencil recomputed,
 *   rest lengths from the three-value palette), v[5] episode-loop flavour (0 plain schedule, 1 flat-tier episodes, 2 + tier-2 /
 *   highest-point code), v[6] dynamic LDS bytes per cloth, v[7] cloths resident per CU for that kernel and LDS size
 *   (hipOccupancyMaxActiveBlocksPerMultiprocessor), v[8] compute units of the device, v[9] precision (0 f64, 1 f32).
 * Returns CLOTHHIP_ESTATE when the handle has not launched a stepper yet. */
int clothhip_last_variant(clothhip_handle *h, int32_t v[10]);

/* (new, ABI 7) Kernel dispatches the handle's last stepper launch was issued as: 1, or -- a time-sliced clothhip_run_actions over more
 * cloths than are resident -- one per generation of resident cloths (rocprofv3 lists them one by one; clothhip_last_kernel_ms spans
 * them all). CLOTHHIP_ESTATE before the first launch. */
int clothhip_last_dispatches(clothhip_handle *h, int32_t *n);

/* (new, ABI 7) 25 or 50 when the handle's last stepper launch ran a GRID-SPECIALISED build of the variant clothhip_last_variant names -- the
 * kernel compiled with the 25x25 grid of the shipped configurations (cfg/t1_rgbd.yaml:15-16) or the 50x50 grid of BASELINE configs[4] as
 * compile-time constants: particle count, LDS carve-up, hash- and window-table sizes --, 0 for the generic build (any grid). Same arithmetic,
 * same results bit for bit (tests/test_gpu_lean.py, tests/test_gpu_fullsize.py); CLOTHHIP_DEBUG_NOSPEC=1 at launch time forces the generic build. */
int clothhip_last_specialised(clothhip_handle *h, int32_t *n_side);

/* (new, ABI 7; measurement only, NOT a reference path) on != 0: THIS handle's clothhip_run_actions launches run the relaxed-order
 * companion kernel -- self-collision in Jacobi order, strain limit in coloured order (cloth.pyx:258-296 and :313-343 keep list order;
 * this does not) -- so that bench.py can state what the reference's Gauss-Seidel orders cost. Its trajectories are not the
 * reference's; clothhip_last_variant reports flavour v[5] == 3 for such a launch. Per handle (round 5 read an environment variable at
 * create time, which leaked into every handle created meanwhile). Only the eight-wave LEAN layout has the companion (fp32, flat tiers,
 * 25x25 class, <= 512 cloths): other handles get CLOTHHIP_ESTATE from clothhip_run_actions*. update() / step() on the handle stay exact. */
int clothhip_set_relaxed_order(clothhip_handle *h, int32_t on);

/* Diagnostics of the last clothhip_run*: stats[E][16]: [0..3] = {strain sweeps run, 64-spring windows walked, passes
 * over a window, passes in which a correction was applied}; [15] = shader clocks/1024 the env's whole schedule
 * took; in the profiling build of the library (make -C gym_cloth_amd/csrc stamps) with CLOTHHIP_DEBUG_PHASES bit 32
 * set, [4..15] = shader cycles/64 per kernel phase instead. Not part of the reference surface. */
int clothhip_debug_stats(clothhip_handle *h, int32_t *stats);

/* Arithmetic self-test used by the parity tests: evaluates out[i] = op(a[i], b[i]) in double ON THE
 * DEVICE with the same compiler flags as the stepper (op 0: a/b, 1: sqrt(a), 2: a*b+c unfused = (a*b)+b,
 * 3: floor(a/b)).  Lets tests assert IEEE-correct rounding of the device's sqrt/div bit-for-bit. */
/* (new, host only) The static tables of the strain sweep for a grid, for the CPU test that pins the sweep's pass rule to the
 * sequential loop of cloth.pyx:258-296: `spring_at[slot]` = reference list index of the spring in window-table slot `slot`
 * (-1: empty), `ent[slot]` = packed entry (ptA | ptB << 12 | level-in-window << 24 | reach << 28), `dep[slot]` = the lanes of
 * the slot's 64-slot window whose springs it transitively depends on. Arrays hold `capacity` slots; returns CLOTHHIP_EINVAL
 * when that is too small (n_slots is set either way). Any array may be NULL. */
int clothhip_selftest_windows(const ClothParams *params, int32_t *n_windows, int32_t *n_slots, int32_t *reach_shift,
                              int32_t *spring_at, uint32_t *ent, uint64_t *dep, int32_t capacity);

/* Host-side view of the variant / LDS-layout decisions clothhip_create takes for (params, precision, n_envs) on a device of n_cus
 * compute units -- no device needed (ABI 6). out[24]: [0..9] the standard layout {threads per cloth, particles per thread, table mode,
 * rest lengths in registers, cell-ordered copy, LDS bytes, hash-table slots, LDS the in-kernel metrics of clothhip_run_actions can
 * borrow, LDS they need, 1 if that fits}; [10] 1 if a LEAN layout exists beside it, [11] the cloths per CU it is built for;
 * [12..21] the LEAN layout, same fields; [22] what clothhip_fused_supported would return; [23] 1 if the layout fits the CU's LDS. */
int clothhip_selftest_layout(const ClothParams *params, int32_t precision, int32_t n_envs, int32_t n_cus, int32_t *out, int32_t capacity);

/* Host-side self-test of csrc/cloth_rng.hpp (the same functions the kernel runs): n draws of kind 0 next32, 1 rand(),
 * 2 uniform(a, b), 3 randint((uint32)a), 4 _randval_minabs(a, b, minabs = c) into out[n]; kind 5 skips (uint64)a words.
 * state[625] = key[624], pos: numpy RandomState.get_state()[1:3], advanced in place. No device needed. */
int clothhip_selftest_rng(uint32_t *state, int32_t kind, int32_t n, double a, double b, double c, double *out);
int clothhip_selftest_arith(int32_t device, int32_t op, const double *a, const double *b, double *out,
                            int64_t n);

#ifdef __cplusplus
}
#endif
#endif
