/*
 * cloth_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cloth_oracle.h).
 *
 * Plain-C, IEEE-double, exact-order restatement of gym_cloth/physics of
 * DanielTakeshi/gym-cloth.  Build with -O2 -ffp-contract=off (no FMA fusion, no
 * fast-math) so every expression rounds exactly like CPython/Cython evaluates it.
 * Citations are file:line in /root/reference/gym_cloth/physics/.
 */
#include "cloth_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { STRUCTURAL = 0, SHEARING = 1, BENDING = 2 };

struct OracleCloth {
    OracleParams prm;
    int N, P, S;
    double dx, dy;                 /* cloth.pyx:55-56 */
    double *x, *y, *z;             /* Point.x/y/z     point.pyx:34-36 */
    double *px, *py, *pz;          /* Point.px/py/pz  point.pyx:37-39 */
    double *fx, *fy, *fz;          /* Point.fx/fy/fz  point.pyx:40-42 */
    uint8_t *pinned;               /* Point.pinned    point.pyx:48 */
    int32_t *sa, *sb;              /* Spring.ptA / ptB  cloth.pyx:414-415 */
    uint8_t *stype;                /* Spring.type       cloth.pyx:416 */
    double *rest;                  /* Spring.rest_length cloth.pyx:417 */
    int tear;                      /* Cloth.cloth_have_tear cloth.pyx:66 */
    /* Gripper.grabbed_pts (ordered, duplicates possible)  gripper.pyx:19 */
    int32_t *grabbed;
    int n_grabbed, cap_grabbed;
    /* spatial map (Cloth.map: dict hash -> list of points in index order) cloth.pyx:298-305 */
    int tsize;                     /* power of two >= 2P */
    long long *tkey;
    int32_t *thead, *ttail;        /* -1 = empty slot */
    int32_t *next;                 /* chain in ascending point index */
    int32_t *slot_of;              /* table slot of each point (from build_spatial_map) */
    /* debug census of the last update(): springs corrected by the strain limiter, points moved by
     * self-collision (used to characterise workloads; not part of the reference's state) */
    int stat_strain, stat_collide;
};

/* cloth.pyx:17-18  math.sqrt(x*x + y*y + z*z), association ((x*x + y*y) + z*z) */
static inline double fastnorm(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) abort();
    return p;
}

int oracle_num_points(const OracleCloth *c) { return c->P; }
int oracle_num_springs(const OracleCloth *c) { return c->S; }
int oracle_have_tear(const OracleCloth *c) { return c->tear; }
void oracle_set_tear(OracleCloth *c, int t) { c->tear = t ? 1 : 0; }
int oracle_num_grabbed(const OracleCloth *c) { return c->n_grabbed; }
void oracle_get_grabbed(const OracleCloth *c, int32_t *idx) {
    memcpy(idx, c->grabbed, sizeof(int32_t) * (size_t)c->n_grabbed);
}

/* Spring census for N x N: 2N(N-1) structural + 2(N-1)^2 shear + 2N(N-2) bending (cloth.pyx:135-146) */
static int spring_count(int N) { return 2 * N * (N - 1) + 2 * (N - 1) * (N - 1) + 2 * N * (N - 2); }

OracleCloth *oracle_create(const OracleParams *p) {
    if (!p || p->n_side < 3) return NULL;
    OracleCloth *c = (OracleCloth *)xcalloc(1, sizeof(*c));
    c->prm = *p;
    int N = c->N = p->n_side;
    int P = c->P = N * N;
    int S = c->S = spring_count(N);
    c->dx = p->width * 1.0 / (N - 1);   /* cloth.pyx:55 */
    c->dy = p->height * 1.0 / (N - 1);  /* cloth.pyx:56 */
    c->x = xcalloc(P, 8); c->y = xcalloc(P, 8); c->z = xcalloc(P, 8);
    c->px = xcalloc(P, 8); c->py = xcalloc(P, 8); c->pz = xcalloc(P, 8);
    c->fx = xcalloc(P, 8); c->fy = xcalloc(P, 8); c->fz = xcalloc(P, 8);
    c->pinned = xcalloc(P, 1);
    c->sa = xcalloc(S, 4); c->sb = xcalloc(S, 4); c->stype = xcalloc(S, 1); c->rest = xcalloc(S, 8);
    c->cap_grabbed = P; c->grabbed = xcalloc(P, 4);
    int t = 1; while (t < 2 * P) t <<= 1;
    c->tsize = t;
    c->tkey = xcalloc(t, sizeof(long long)); c->thead = xcalloc(t, 4); c->ttail = xcalloc(t, 4);
    c->next = xcalloc(P, 4); c->slot_of = xcalloc(P, 4);
    for (int i = 0; i < t; i++) c->thead[i] = -1;
    oracle_init_grid(c, 1, 0, NULL);
    return c;
}

void oracle_destroy(OracleCloth *c) {
    if (!c) return;
    free(c->x); free(c->y); free(c->z); free(c->px); free(c->py); free(c->pz);
    free(c->fx); free(c->fy); free(c->fz); free(c->pinned);
    free(c->sa); free(c->sb); free(c->stype); free(c->rest); free(c->grabbed);
    free(c->tkey); free(c->thead); free(c->ttail); free(c->next); free(c->slot_of);
    free(c);
}

/* cloth.pyx:92-146: point grid + spring list in row-major owner order, six springs per point in
 * the order S(r-1,c), S(r,c-1), Sh(r-1,c-1), Sh(r-1,c+1), B(r-2,c), B(r,c-2); ptA = earlier point,
 * ptB = owning point; rest_length = |A-B| of the initial positions (cloth.pyx:417). */
int oracle_init_grid(OracleCloth *c, int tier, int init_side, const double *rand_draws) {
    const int N = c->N;
    const double dx = c->dx, dy = c->dy;
    if (tier < 1 || tier > 3) return -1;
    if (tier == 2 && !rand_draws) return -2;
    int s = 0;
    for (int r = 0; r < N; r++) {
        for (int col = 0; col < N; col++) {
            int i = r * N + col;
            double X, Y, Z;
            if (tier == 2) {
                double noise = rand_draws[i] * 0.01 - 0.005;        /* cloth.pyx:101 */
                if (r == 0) noise = 0;                              /* cloth.pyx:102-103 */
                X = init_side ? 0.0 + fabs(noise) : 1.0 - fabs(noise); /* cloth.pyx:104-107 */
                Y = dx * col;                                       /* cloth.pyx:109 */
                Z = dy * r;                                         /* cloth.pyx:110 */
            } else {
                X = dx * r; Y = dy * col; Z = 0.0;                  /* cloth.pyx:122-124 */
            }
            c->x[i] = c->px[i] = X; c->y[i] = c->py[i] = Y; c->z[i] = c->pz[i] = Z;  /* point.pyx:34-39 */
            c->fx[i] = c->fy[i] = c->fz[i] = 0.0;
            c->pinned[i] = 0;
#define ADD_SPRING(AIDX, TYPE)                                                             \
    do {                                                                                   \
        int a_ = (AIDX);                                                                   \
        c->sa[s] = a_; c->sb[s] = i; c->stype[s] = (TYPE);                                 \
        c->rest[s] = fastnorm(c->x[a_] - X, c->y[a_] - Y, c->z[a_] - Z); /* cloth.pyx:417 */ \
        s++;                                                                               \
    } while (0)
            if (r > 0) ADD_SPRING((r - 1) * N + col, STRUCTURAL);                  /* :135-136 */
            if (col > 0) ADD_SPRING(r * N + col - 1, STRUCTURAL);                  /* :137-138 */
            if (r > 0 && col > 0) ADD_SPRING((r - 1) * N + col - 1, SHEARING);     /* :139-140 */
            if (r > 0 && col + 1 < N) ADD_SPRING((r - 1) * N + col + 1, SHEARING); /* :141-142 */
            if (r > 1) ADD_SPRING((r - 2) * N + col, BENDING);                     /* :143-144 */
            if (col > 1) ADD_SPRING(r * N + col - 2, BENDING);                     /* :145-146 */
#undef ADD_SPRING
        }
    }
    if (s != c->S) return -3;
    c->tear = 0;
    c->n_grabbed = 0;
    return 0;
}

void oracle_set_state(OracleCloth *c, const double *pos, const double *prev, const uint8_t *pinned,
                      const double *rest) {
    for (int i = 0; i < c->P; i++) {
        if (pos) { c->x[i] = pos[3 * i]; c->y[i] = pos[3 * i + 1]; c->z[i] = pos[3 * i + 2]; }
        if (prev) { c->px[i] = prev[3 * i]; c->py[i] = prev[3 * i + 1]; c->pz[i] = prev[3 * i + 2]; }
    }
    if (pinned) {
        c->n_grabbed = 0;
        for (int i = 0; i < c->P; i++) {
            c->pinned[i] = pinned[i] ? 1 : 0;
            if (pinned[i]) c->grabbed[c->n_grabbed++] = i;
        }
    }
    if (rest) memcpy(c->rest, rest, sizeof(double) * (size_t)c->S);
}

void oracle_get_state(const OracleCloth *c, double *pos, double *prev, uint8_t *pinned) {
    for (int i = 0; i < c->P; i++) {
        if (pos) { pos[3 * i] = c->x[i]; pos[3 * i + 1] = c->y[i]; pos[3 * i + 2] = c->z[i]; }
        if (prev) { prev[3 * i] = c->px[i]; prev[3 * i + 1] = c->py[i]; prev[3 * i + 2] = c->pz[i]; }
        if (pinned) pinned[i] = c->pinned[i];
    }
}

void oracle_get_rest(const OracleCloth *c, double *rest) { memcpy(rest, c->rest, 8 * (size_t)c->S); }

void oracle_get_springs(const OracleCloth *c, int32_t *a, int32_t *b, uint8_t *type) {
    memcpy(a, c->sa, 4 * (size_t)c->S);
    memcpy(b, c->sb, 4 * (size_t)c->S);
    memcpy(type, c->stype, (size_t)c->S);
}

void oracle_pin(OracleCloth *c, int idx) {
    if (idx >= 0 && idx < c->P) c->pinned[idx] = 1;
}

/* ---------------------------------------------------------------------------------------------- */
/* Cloth.update phases                                                                            */
/* ---------------------------------------------------------------------------------------------- */

/* cloth.pyx:216-219 + point.pyx:60-71,83-86: f = 0 ; f = f + (0, 0, m*g) */
static void reset_gravity(OracleCloth *c, double mass_times_g) {
    for (int i = 0; i < c->P; i++) {
        c->fx[i] = 0.0; c->fy[i] = 0.0; c->fz[i] = 0.0;
        c->fx[i] = c->fx[i] + 0; c->fy[i] = c->fy[i] + 0; c->fz[i] = c->fz[i] + mass_times_g;
    }
}

/* cloth.pyx:221-237 */
static void hookes(OracleCloth *c, double cp_ks) {
    for (int s = 0; s < c->S; s++) {
        double K = (c->stype[s] == BENDING) ? 0.2 : 1.0;          /* :225-228 */
        int a = c->sa[s], b = c->sb[s];
        double l = fastnorm(c->x[b] - c->x[a], c->y[b] - c->y[a], c->z[b] - c->z[a]);  /* :231 */
        double fm = cp_ks * K * (l - c->rest[s]) / l;             /* :232, left-to-right */
        double f0 = fm * (c->x[b] - c->x[a]);                     /* :233 */
        double f1 = fm * (c->y[b] - c->y[a]);
        double f2 = fm * (c->z[b] - c->z[a]);
        c->fx[a] = c->fx[a] + f0; c->fy[a] = c->fy[a] + f1; c->fz[a] = c->fz[a] + f2;        /* :236 */
        c->fx[b] = c->fx[b] + (-f0); c->fy[b] = c->fy[b] + (-f1); c->fz[b] = c->fz[b] + (-f2); /* :237 */
    }
}

/* cloth.pyx:239-256 */
static void verlet(OracleCloth *c, double mass, double delta_t, double cp_damping) {
    double dsm = (delta_t * delta_t) / mass;      /* :240 */
    double damping = (1.0 - cp_damping / 100.0);  /* :241 */
    for (int i = 0; i < c->P; i++) {
        if (c->pinned[i]) continue;               /* :244 */
        double cx = c->x[i], cy = c->y[i], cz = c->z[i];
        double nx = c->x[i] + (damping * (c->x[i] - c->px[i])) + (c->fx[i] * dsm);  /* :249 */
        double ny = c->y[i] + (damping * (c->y[i] - c->py[i])) + (c->fy[i] * dsm);
        double nz = c->z[i] + (damping * (c->z[i] - c->pz[i])) + (c->fz[i] * dsm);
        c->x[i] = nx; c->y[i] = ny; c->z[i] = nz;                                    /* :255 */
        c->px[i] = cx; c->py[i] = cy; c->pz[i] = cz;                                 /* :256 */
    }
}

/* cloth.pyx:307-311: 961*floor(x/w) + 31*floor(y/h) + floor(z/t), w=3dx, h=3dy, t=max(w,h).
 * Python computes this with unbounded ints; long long is exact for every finite position a cloth
 * can reach before |x| ~ 1e15. Non-finite coordinates (Python would raise) map to a sentinel. */
static long long hash_position(const OracleCloth *c, double x, double y, double z) {
    double w = 3 * c->dx, h = 3 * c->dy;
    double t = (w > h) ? w : h;   /* max(w, h) returns w when equal */
    double a = floor(x / w), b = floor(y / h), d = floor(z / t);
    if (!(isfinite(a) && isfinite(b) && isfinite(d))) return (long long)0x7fffffffffffff00LL;
    return 961LL * (long long)a + 31LL * (long long)b + (long long)d;
}

static int table_slot(const OracleCloth *c, long long key) {
    unsigned long long h = (unsigned long long)key * 0x9E3779B97F4A7C15ULL;
    int m = c->tsize - 1;
    int s = (int)(h >> 40) & m;
    while (c->thead[s] != -1 && c->tkey[s] != key) s = (s + 1) & m;
    return s;
}

/* cloth.pyx:298-305: cell -> list of points in ascending index order (pinned points included) */
static void build_spatial_map(OracleCloth *c) {
    for (int i = 0; i < c->tsize; i++) c->thead[i] = -1;
    for (int i = 0; i < c->P; i++) {
        long long key = hash_position(c, c->x[i], c->y[i], c->z[i]);
        int s = table_slot(c, key);
        if (c->thead[s] == -1) { c->tkey[s] = key; c->thead[s] = i; } else { c->next[c->ttail[s]] = i; }
        c->ttail[s] = i;
        c->next[i] = -1;
        c->slot_of[i] = s;
    }
}

/* cloth.pyx:313-343. The point's cell is re-hashed from its current position; it has not moved since
 * build_spatial_map (only self_collide(pt) moves pt), so the cell is the one it was filed under. */
static void self_collide(OracleCloth *c, int i, int simulation_steps, double thickness) {
    if (c->pinned[i]) return;                                     /* :314 */
    long long key = hash_position(c, c->x[i], c->y[i], c->z[i]);  /* :316 */
    double thresh = 2.0 * thickness;                              /* :317 */
    int s = table_slot(c, key);
    if (c->thead[s] == -1) return;                                /* :319 */
    double tx = 0.0, ty = 0.0, tz = 0.0;
    int n = 0;
    for (int j = c->thead[s]; j != -1; j = c->next[j]) {          /* :324 */
        if (j == i) continue;                                     /* :325 */
        double dist = fastnorm(c->x[i] - c->x[j], c->y[i] - c->y[j], c->z[i] - c->z[j]);  /* :327 */
        if (dist <= thresh) {                                     /* :330 */
            double factor = (thresh - dist) / dist;               /* :331 */
            tx += (c->x[i] - c->x[j]) * factor;                   /* :332 */
            ty += (c->y[i] - c->y[j]) * factor;
            tz += (c->z[i] - c->z[j]) * factor;
            n += 1;
        }
    }
    if (n != 0) {                                                 /* :336 */
        c->stat_collide++;
        double nf = (double)n;
        double cx = tx / nf / simulation_steps;                   /* :338 */
        double cy = ty / nf / simulation_steps;
        double cz = tz / nf / simulation_steps;
        c->x[i] = c->x[i] + cx; c->y[i] = c->y[i] + cy; c->z[i] = c->z[i] + cz;  /* :341 */
    }
}

/* cloth.pyx:345-370, plane normal (0,0,1) */
static void plane_collision(OracleCloth *c, int i, double p_friction, double surface_offset) {
    double minimum_z = c->prm.minimum_z;
    if (c->pinned[i] || c->z[i] >= minimum_z) return;             /* :356 */
    double t = (minimum_z - c->pz[i]) * 1.0;                      /* :358 */
    double tangent_x = c->px[i] + t * (-0.0);                     /* :359 */
    double tangent_y = c->py[i] + t * (-0.0);
    double tangent_z = c->pz[i] + t * (-1.0);
    double goal_x = tangent_x + surface_offset * 0.0;             /* :362 */
    double goal_y = tangent_y + surface_offset * 0.0;
    double goal_z = tangent_z + surface_offset * 1.0;
    double corr_x = goal_x - c->px[i];                            /* :365 */
    double corr_y = goal_y - c->py[i];
    double corr_z = goal_z - c->pz[i];
    c->x[i] = c->px[i] + corr_x * (1. - p_friction);              /* :368 */
    c->y[i] = c->py[i] + corr_y * (1. - p_friction);
    c->z[i] = c->pz[i] + corr_z * (1. - p_friction);
}

/* cloth.pyx:258-296: single in-place pass in spring-list order (Provot), + sticky tear flag */
static void limit_spring_changes(OracleCloth *c, double tear_thresh) {
    for (int s = 0; s < c->S; s++) {
        int a = c->sa[s], b = c->sb[s];
        if (c->pinned[a] && c->pinned[b]) continue;                                       /* :268 */
        double len = fastnorm(c->x[a] - c->x[b], c->y[a] - c->y[b], c->z[a] - c->z[b]);   /* :270 */
        if (len > c->rest[s] * tear_thresh) c->tear = 1;                                  /* :272 */
        if (len > (c->rest[s] * 1.1)) {                                                   /* :275 */
            c->stat_strain++;
            double dirx = (c->x[a] - c->x[b]) / len;                                      /* :276 */
            double diry = (c->y[a] - c->y[b]) / len;
            double dirz = (c->z[a] - c->z[b]) / len;
            double extra = len - c->rest[s] * 1.1;                                        /* :279 */
            if (c->pinned[a]) {                                                           /* :281 */
                c->x[b] = c->x[b] + dirx * extra; c->y[b] = c->y[b] + diry * extra; c->z[b] = c->z[b] + dirz * extra;
            } else if (c->pinned[b]) {                                                    /* :285 */
                c->x[a] = c->x[a] - dirx * extra; c->y[a] = c->y[a] - diry * extra; c->z[a] = c->z[a] - dirz * extra;
            } else {                                                                      /* :289 */
                double ed = extra * 0.5;
                c->x[a] = c->x[a] - dirx * ed; c->y[a] = c->y[a] - diry * ed; c->z[a] = c->z[a] - dirz * ed;
                c->x[b] = c->x[b] + dirx * ed; c->y[b] = c->y[b] + diry * ed; c->z[b] = c->z[b] + dirz * ed;
            }
        }
    }
}

/* cloth.pyx:169-214 (render/zmq branch omitted: out of scope) */
static void update_once(OracleCloth *c) {
    const OracleParams *p = &c->prm;
    int simulation_steps = p->simulation_steps;
    double mass = p->density / c->N / c->N;                       /* :178 */
    double mass_times_g = mass * p->gravity;                      /* :179 */
    double delta_t = 1.0 / p->frames_per_sec / simulation_steps;  /* :180 */
    c->stat_strain = 0; c->stat_collide = 0;
    reset_gravity(c, mass_times_g);                               /* :189 */
    hookes(c, p->ks);                                             /* :192 */
    verlet(c, mass, delta_t, p->damping);                         /* :195 */
    build_spatial_map(c);                                         /* :198 */
    for (int i = 0; i < c->P; i++) self_collide(c, i, simulation_steps, p->thickness);   /* :199-200 */
    for (int i = 0; i < c->P; i++) plane_collision(c, i, p->plane_friction, 0.0001);     /* :203-204 */
    limit_spring_changes(c, p->tear_thresh);                      /* :207 */
}

void oracle_update(OracleCloth *c, int n) {
    for (int k = 0; k < n; k++) update_once(c);
}

void oracle_last_stats(const OracleCloth *c, int32_t *n_strain, int32_t *n_collide) {
    *n_strain = c->stat_strain; *n_collide = c->stat_collide;
}

void oracle_cell_census(const OracleCloth *c, int32_t *n_cells, int32_t *max_occ) {
    int nc = 0, mo = 0;
    for (int s = 0; s < c->tsize; s++) {
        if (c->thead[s] == -1) continue;
        int occ = 0;
        for (int j = c->thead[s]; j != -1; j = c->next[j]) occ++;
        nc++;
        if (occ > mo) mo = occ;
    }
    *n_cells = nc; *max_occ = mo;
}

/* ---------------------------------------------------------------------------------------------- */
/* Gripper                                                                                        */
/* ---------------------------------------------------------------------------------------------- */

static void push_grabbed(OracleCloth *c, int i) {
    if (c->n_grabbed == c->cap_grabbed) {
        c->cap_grabbed *= 2;
        c->grabbed = (int32_t *)realloc(c->grabbed, sizeof(int32_t) * (size_t)c->cap_grabbed);
        if (!c->grabbed) abort();
    }
    c->grabbed[c->n_grabbed++] = i;
}

/* gripper.pyx:23-42: scan levels curZ = height, height - thickness, ... while curZ > 0; at the first
 * level where any point has (dx^2 + dy^2 < grip_radius) [radius NOT squared] and |z - curZ| < 2*thickness,
 * pin all such points. Returns the number of points appended to grabbed_pts. */
int oracle_grab_top(OracleCloth *c, double x, double y, double grip_radius) {
    double curZ = c->prm.height;                                  /* :31, Gripper(height=cfg cloth.height) */
    double thickness = c->prm.thickness;
    int n_new = 0;
    while (curZ > 0) {                                            /* :33 */
        for (int i = 0; i < c->P; i++) {
            if ((c->x[i] - x) * (c->x[i] - x) + (c->y[i] - y) * (c->y[i] - y) < grip_radius &&
                fabs(c->z[i] - curZ) < 2 * thickness) {           /* :35-36 */
                c->pinned[i] = 1;                                 /* :37 */
                push_grabbed(c, i);
                n_new++;
            }
        }
        if (n_new) break;                                         /* :39-40 */
        curZ -= thickness;                                        /* :41 */
    }
    return n_new;
}

/* gripper.pyx:44-53 */
int oracle_grab(OracleCloth *c, double x, double y, double grip_radius) {
    int n_new = 0;
    for (int i = 0; i < c->P; i++) {
        if ((c->x[i] - x) * (c->x[i] - x) + (c->y[i] - y) * (c->y[i] - y) < grip_radius) {
            c->pinned[i] = 1;
            push_grabbed(c, i);
            n_new++;
        }
    }
    return n_new;
}

/* gripper.pyx:55-66: p <- x ; x <- delta + x */
void oracle_adjust(OracleCloth *c, double dx, double dy, double dz) {
    for (int k = 0; k < c->n_grabbed; k++) {
        int i = c->grabbed[k];
        c->px[i] = c->x[i]; c->py[i] = c->y[i]; c->pz[i] = c->z[i];
        c->x[i] = dx + c->x[i]; c->y[i] = dy + c->y[i]; c->z[i] = dz + c->z[i];
    }
}

/* gripper.pyx:68-73 */
void oracle_release(OracleCloth *c) {
    for (int k = 0; k < c->n_grabbed; k++) c->pinned[c->grabbed[k]] = 0;
    c->n_grabbed = 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* ClothEnv.step hot loop (cloth_env.py:352-367, :495-515)                                        */
/* ---------------------------------------------------------------------------------------------- */

int oracle_run_schedule(OracleCloth *c, int n_up_end, int n_uprest_end, int n_pull_end,
                        int n_griprest_end, int n_total, double dz_up, double dx_pull,
                        double dy_pull, int break_on_tear) {
    int done = 0;
    for (int i = 0; i < n_total; i++) {
        if (i < n_up_end) oracle_adjust(c, 0.0, 0.0, dz_up);             /* cloth_env.py:358-359 */
        else if (i < n_uprest_end) { }                                   /* :360-361 */
        else if (i < n_pull_end) oracle_adjust(c, dx_pull, dy_pull, 0.0);/* :362-363 */
        else if (i < n_griprest_end) { }                                 /* :364-365 */
        else oracle_release(c);                                          /* :366-367 */
        update_once(c);                                                  /* :498 */
        done++;
        if (break_on_tear && c->tear) break;                             /* :511-514 */
    }
    return done;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_batch_run_schedule(OracleCloth **cs, int n, const int32_t *sched, const double *delta,
                               int break_on_tear, int32_t *executed, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
    for (int e = 0; e < n; e++) {
        const int32_t *s = sched + 5 * e;
        const double *d = delta + 3 * e;
        executed[e] = oracle_run_schedule(cs[e], s[0], s[1], s[2], s[3], s[4], d[0], d[1], d[2], break_on_tear);
    }
}

void oracle_batch_update(OracleCloth **cs, int n, int n_sub, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
    for (int e = 0; e < n; e++) oracle_update(cs[e], n_sub);
}
