/*
 * cloth_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE double, exact reference operation order) of the
 * gym-cloth physics hot path: gym_cloth/physics/{cloth,point,gripper}.pyx of
 * DanielTakeshi/gym-cloth.  Every function cites the reference lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline -- never as the product path.
 *
 * Parity pin: the restatement is checked bit-for-bit against fixtures produced by
 * importing the real reference (tests/golden/make_golden.py -> tests/golden/ npz files);
 * see tests/test_oracle_golden.py.  The reference ships no tests or golden vectors of
 * its own for this path.
 */
#ifndef CLOTH_ORACLE_H
#define CLOTH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleParams {
    int32_t n_side;            /* cfg cloth.num_width_points == num_height_points (cloth.pyx:53-54,91) */
    int32_t frames_per_sec;    /* cfg frames_per_sec   (cloth.pyx:176) */
    int32_t simulation_steps;  /* cfg simulation_steps (cloth.pyx:177) */
    int32_t _pad;
    double width, height;      /* cfg cloth.width/height (cloth.pyx:55-56) */
    double density, ks, damping, thickness, plane_friction, tear_thresh; /* cloth.pyx:178-186 */
    double gravity;            /* Cloth(gravity=-9.8)  (cloth.pyx:24) */
    double minimum_z;          /* Cloth(minimum_z=0)   (cloth.pyx:26) */
} OracleParams;

typedef struct OracleCloth OracleCloth;

OracleCloth *oracle_create(const OracleParams *p);
void oracle_destroy(OracleCloth *c);
int oracle_num_points(const OracleCloth *c);
int oracle_num_springs(const OracleCloth *c);

/* cloth.pyx:92-146. tier: 1,2,3. rand_draws: the P values np_random.rand() returned in the
 * r-major loop (tier 2 only, may be NULL otherwise). Resets pinned/grabbed/tear. */
int oracle_init_grid(OracleCloth *c, int tier, int init_side, const double *rand_draws);

/* pos/prev are [P][3] (x,y,z interleaved, the layout of Cloth.allpts_arr, cloth.pyx:395).
 * rest may be NULL (keep current). pinned!=0 -> pinned AND member of gripper.grabbed_pts. */
void oracle_set_state(OracleCloth *c, const double *pos, const double *prev, const uint8_t *pinned,
                      const double *rest);
void oracle_get_state(const OracleCloth *c, double *pos, double *prev, uint8_t *pinned);
void oracle_get_rest(const OracleCloth *c, double *rest);
void oracle_get_springs(const OracleCloth *c, int32_t *a, int32_t *b, uint8_t *type);
/* pin without adding to grabbed_pts (pt.pinned = True from outside) */
void oracle_pin(OracleCloth *c, int idx);

void oracle_update(OracleCloth *c, int n);                          /* Cloth.update x n, cloth.pyx:169-214 */
int oracle_grab_top(OracleCloth *c, double x, double y, double grip_radius);  /* gripper.pyx:23-42 */
int oracle_grab(OracleCloth *c, double x, double y, double grip_radius);      /* gripper.pyx:44-53 */
void oracle_adjust(OracleCloth *c, double dx, double dy, double dz);          /* gripper.pyx:55-66 */
void oracle_release(OracleCloth *c);                                          /* gripper.pyx:68-73 */
int oracle_num_grabbed(const OracleCloth *c);
void oracle_get_grabbed(const OracleCloth *c, int32_t *idx);
int oracle_have_tear(const OracleCloth *c);                                   /* cloth.pyx:391 */
void oracle_set_tear(OracleCloth *c, int tear);
/* census of the spatial map left by the last update (number of cells, max occupancy) */
void oracle_cell_census(const OracleCloth *c, int32_t *n_cells, int32_t *max_occ);
/* debug: #springs corrected by the strain limiter / #points moved by self-collision in the last update */
void oracle_last_stats(const OracleCloth *c, int32_t *n_strain, int32_t *n_collide);

/* The per-action hot loop of ClothEnv.step + _pull (cloth_env.py:352-367, :495-515):
 *   for i in [0, n_total): phase(i) -> adjust / nothing / release ; update ; break on tear.
 * n_* are the integer ceilings of the (possibly fractional, tier 3) phase boundaries.
 * Returns the number of update() calls executed. */
int oracle_run_schedule(OracleCloth *c, int n_up_end, int n_uprest_end, int n_pull_end,
                        int n_griprest_end, int n_total, double dz_up, double dx_pull,
                        double dy_pull, int break_on_tear);

/* Batch helper for the CPU baseline: run_schedule over n cloths, OpenMP over cloths
 * (one cloth per thread). sched is [n][5] ints, delta is [n][3] doubles (dz_up, dx, dy).
 * executed[n] receives the update counts. */
void oracle_batch_run_schedule(OracleCloth **cs, int n, const int32_t *sched, const double *delta,
                               int break_on_tear, int32_t *executed, int n_threads);
void oracle_batch_update(OracleCloth **cs, int n, int n_sub, int n_threads);
int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
