/* eph_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See eph_oracle.h.
 *
 * Plain-C restatement of the reference algorithm, in the reference's loop order, AoS double[3] vectors,
 * compiled with -ffp-contract=off (Rust never fuses a*b+c). Each function cites the reference file:line
 * (relative to /root/reference) that it follows.
 */
#include "eph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* Ratio  (integration/src/ratio.rs)                                                                */
/* ------------------------------------------------------------------------------------------------ */
typedef __int128 i128;
typedef unsigned __int128 u128;
typedef struct { int64_t hi; uint64_t lo; } EPH_I128;
typedef struct { EPH_I128 n, d; } EPH_RATIO;
#include "coeff_tables.inc"

static i128 i128_of(EPH_I128 v) { return (i128)(((u128)(uint64_t)v.hi << 64) | (u128)v.lo); }

/* Mul<Ratio> for f64: `self * (rhs.numer as f64 / rhs.denom as f64)`  ratio.rs:221-228.
 * `as f64` on i128 rounds to nearest-even; so does gcc's __floattidf. */
static double ratio_f64(EPH_RATIO r) { return (double)i128_of(r.n) / (double)i128_of(r.d); }
static double int_f64(EPH_I128 v) { return (double)i128_of(v); }

static u128 uabs(i128 v) { return v < 0 ? (u128)0 - (u128)v : (u128)v; }
static int ctz128(u128 v) {
    uint64_t lo = (uint64_t)v;
    if (lo) return __builtin_ctzll(lo);
    return 64 + __builtin_ctzll((uint64_t)(v >> 64));
}
/* Stein's gcd, ratio.rs:246-271 */
static u128 gcd_u128(u128 a, u128 b) {
    u128 m = a, n = b;
    if (m == 0 || n == 0) return m | n;
    int shift = ctz128(m | n);
    m >>= ctz128(m);
    n >>= ctz128(n);
    while (m != n) {
        if (m > n) { m -= n; m >>= ctz128(m); }
        else { n -= m; n >>= ctz128(n); }
    }
    return m << shift;
}
/* normalize, ratio.rs:153-177 */
static void ratio_normalize(i128 *numer, i128 *denom) {
    if (*denom == 0) return;
    if (*numer == 0) { *denom = 1; return; }
    if (*numer == *denom) { *numer = 1; *denom = 1; return; }
    i128 g = (i128)gcd_u128(uabs(*numer), uabs(*denom));
    *numer /= g;
    *denom /= g;
    if (*denom < 0) { *numer = -*numer; *denom = -*denom; }
}
static u128 pow10_u128(unsigned p) { u128 r = 1; while (p--) r *= 10; return r; }

/* Ratio::from_f64, ratio.rs:75-103 (finite inputs only) */
double orc_ratio_from_f64(double val, int64_t out_nd[4]) {
    unsigned p = 0;
    double new_val = val;
    for (;;) {
        double a = fabs(new_val);
        /* `(a as u64) as f64 == a` with Rust's saturating float->int cast */
        uint64_t au = a >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)a;
        if ((double)au == a) break;
        p += 1;
        new_val = val * (double)pow10_u128(p);
    }
    i128 n = (i128)new_val, d = (i128)pow10_u128(p);
    ratio_normalize(&n, &d);
    if (out_nd) {
        out_nd[0] = (int64_t)(n >> 64); out_nd[1] = (int64_t)(uint64_t)n;
        out_nd[2] = (int64_t)(d >> 64); out_nd[3] = (int64_t)(uint64_t)d;
    }
    return (double)n / (double)d;
}
double orc_ratio_to_f64(int64_t n_hi, uint64_t n_lo, int64_t d_hi, uint64_t d_lo) {
    EPH_RATIO r = {{n_hi, n_lo}, {d_hi, d_lo}};
    return ratio_f64(r);
}

/* ------------------------------------------------------------------------------------------------ */
/* coefficient lookup                                                                               */
/* ------------------------------------------------------------------------------------------------ */
static const EPH_SRKN_TABLE *find_srkn(const char *name) {
    for (int i = 0; i < EPH_N_SRKN_TABLES; ++i)
        if (!strcmp(eph_srkn_tables[i].name, name)) return &eph_srkn_tables[i];
    return NULL;
}
static const EPH_ELM2_TABLE *find_elm2(const char *name) {
    for (int i = 0; i < EPH_N_ELM2_TABLES; ++i)
        if (!strcmp(eph_elm2_tables[i].name, name)) return &eph_elm2_tables[i];
    return NULL;
}
static const EPH_ERK_TABLE *find_erk(const char *name) {
    for (int i = 0; i < EPH_N_ERK_TABLES; ++i)
        if (!strcmp(eph_erk_tables[i].name, name)) return &eph_erk_tables[i];
    return NULL;
}

int orc_srkn_coeffs(const char *name, int *stages, int *fsal, double *A, double *B) {
    const EPH_SRKN_TABLE *t = find_srkn(name);
    if (!t) return ORC_BAD_ARGUMENT;
    *stages = t->stages; *fsal = t->fsal;
    for (int s = 0; s < t->stages; ++s) { A[s] = ratio_f64(t->A[s]); B[s] = ratio_f64(t->B[s]); }
    return ORC_OK;
}
int orc_elm2_coeffs(const char *name, int *order, double *wa, double *wb, double *inv_beta_d, double *cw,
                    double *inv_cowell_d) {
    const EPH_ELM2_TABLE *t = find_elm2(name);
    if (!t) return ORC_BAD_ARGUMENT;
    *order = t->order;
    for (int j = 0; j < t->order; ++j) {
        /* P::Time::one() * Ratio::from_int(-C::ALPHA[j+1])  second_order/mod.rs:107 */
        wa[j] = 1.0 * ((double)(-i128_of(t->ALPHA[j + 1])) / 1.0);
        wb[j] = 1.0 * (int_f64(t->BETA_N[j + 1]) / 1.0);
        cw[j] = 1.0 * (int_f64(t->COWELL_N[j]) / 1.0);
    }
    *inv_beta_d = 1.0 / int_f64(t->BETA_D);      /* Ratio::from_recip -> 1 as f64 / D as f64 */
    *inv_cowell_d = 1.0 / int_f64(t->COWELL_D);
    return ORC_OK;
}
int orc_erk_coeffs(const char *name, int *stages, int *order, int *order_embedded, int *fsal, double *A,
                   double *B, double *C, double *E) {
    const EPH_ERK_TABLE *t = find_erk(name);
    if (!t) return ORC_BAD_ARGUMENT;
    int s = t->stages;
    *stages = s; *order = t->order; *order_embedded = t->order_embedded; *fsal = t->fsal;
    for (int i = 0; i < s * (s - 1) / 2; ++i) A[i] = ratio_f64(t->A[i]);
    for (int i = 0; i < s; ++i) {
        B[i] = ratio_f64(t->B[i]);
        C[i] = ratio_f64(t->C[i]);
        E[i] = t->E ? ratio_f64(t->E[i]) : 0.0;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* NewtonianGravity::eval  ephemeris/src/propagators/nbody.rs:16-39                                 */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;

static uint64_t g_pair_counter = 0;
uint64_t orc_pair_counter(void) { return g_pair_counter; }

/* (V, f64)::acceleration_paired(&other, &softening=0.0) -- crate `particular` 0.8.0-dev @ d490707a, source
 * NOT on disk (Cargo.lock:4277-4285): PARITY UNPINNED. Restated in the form of the published crate
 * (<= 0.7): dir = p_other - p_self ; n2 = dir.length_squared() (+ softening^2 = 0) ;
 * inv = 1 / (n2 * sqrt(n2)) ; (dir * (mu_other*inv), -dir * (mu_self*inv)).
 * glam DVec3::length_squared = x*x + y*y + z*z evaluated left to right (glam 0.30.10). */
/* Sensitivity switch (tests only): other orders a point-mass routine could plausibly use. 0 = the pinned restatement
 * `dir * (mu * (1 / (n2 * sqrt(n2))))`; 1 = 1/(r*r*r), r = sqrt(n2); 2 = s*s*s, s = 1/sqrt(n2); 3 = (1/n2) * (1/sqrt(n2))
 * -- all of the shape "one reciprocal, then dir * (mu * inv)". The DIVISION forms, with p = n2 * sqrt(n2) and glam's
 * component-wise `DVec3 / f64` (three true divisions):
 *     4 = (dir * mu) / p      -- Rust `dir * mu / (mag_2 * mag_2.sqrt())`, the published crate's documented scalar form
 *     5 = dir * (mu / p)
 *     6 = (dir / p) * mu      -- a paired routine that shares `dir / p` between the two masses
 * tests/test_oracle.py::test_pair_formula_variants_stay_bounded bounds what the unpinned choice can cost. */
static int g_pair_variant = 0;
void orc_set_pair_variant(int v) { g_pair_variant = v; }
static inline double inv_r3(double n2) {
    switch (g_pair_variant) {
        case 1: { const double r = sqrt(n2); return 1.0 / (r * r * r); }
        case 2: { const double s = 1.0 / sqrt(n2); return s * s * s; }
        case 3: return (1.0 / n2) * (1.0 / sqrt(n2));
        default: return 1.0 / (n2 * sqrt(n2));
    }
}
/* the acceleration of a point mass mu seen along dir (n2 = |dir|^2), in the selected operation order */
static inline v3 point_mass_term(v3 d, double n2, double mu) {
    v3 a;
    switch (g_pair_variant) {
        case 4: { const double p = n2 * sqrt(n2); a.x = (d.x * mu) / p; a.y = (d.y * mu) / p; a.z = (d.z * mu) / p; break; }
        case 5: { const double s = mu / (n2 * sqrt(n2)); a.x = d.x * s; a.y = d.y * s; a.z = d.z * s; break; }
        case 6: { const double p = n2 * sqrt(n2); a.x = (d.x / p) * mu; a.y = (d.y / p) * mu; a.z = (d.z / p) * mu; break; }
        default: { const double s = mu * inv_r3(n2); a.x = d.x * s; a.y = d.y * s; a.z = d.z * s; }
    }
    return a;
}
static inline void acceleration_paired(v3 pi, double mui, v3 pj, double muj, v3 *ai, v3 *aj) {
    const v3 d = {pj.x - pi.x, pj.y - pi.y, pj.z - pi.z};
    const double n2 = d.x * d.x + d.y * d.y + d.z * d.z;
    const v3 nd = {-d.x, -d.y, -d.z};
    *ai = point_mass_term(d, n2, muj);
    *aj = point_mass_term(nd, n2, mui);            /* -dir * ... : the negation is exact wherever it is applied */
}

static void gravity_eval(int n, const v3 *y, const double *mu, v3 *ddy) {
    for (int i = 0; i < n; ++i) {
        v3 out = {0.0, 0.0, 0.0};                       /* let mut output_i = V::default() */
        for (int j = i + 1; j < n; ++j) {
            v3 ai, aj;
            acceleration_paired(y[i], mu[i], y[j], mu[j], &ai, &aj);
            out.x += ai.x; out.y += ai.y; out.z += ai.z;     /* output_i += computed.0 */
            ddy[j].x += aj.x; ddy[j].y += aj.y; ddy[j].z += aj.z; /* ddy[j] += computed.1 */
        }
        ddy[i].x += out.x; ddy[i].y += out.y; ddy[i].z += out.z;  /* ddy[i] += output_i */
    }
    g_pair_counter += (uint64_t)n * (uint64_t)(n - 1) / 2;
}
/* "What a parallel CPU could do" (SURVEY 8(d)): the same sums partitioned by TARGET body over OpenMP threads. Each
 * thread evaluates whole rows -- all N^2 directed interactions instead of N(N-1)/2 pairs -- in the reference's
 * per-body order, so the result has the same bits as gravity_eval: for j < i the term is the `computed.1` half of
 * pair (j, i), -(p_i - p_j) * (mu_j * inv). Off (0 threads) unless orc_set_gravity_threads is called. */
static int g_gravity_threads = 0;
void orc_set_gravity_threads(int t) { g_gravity_threads = t; }
static void gravity_eval_rows(int n, const v3 *y, const double *mu, v3 *ddy) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_gravity_threads)
    for (int i = 0; i < n; ++i) {
        v3 acc = ddy[i];
        for (int j = 0; j < i; ++j) {
            v3 aj_, ai_;
            acceleration_paired(y[j], mu[j], y[i], mu[i], &aj_, &ai_);   /* pair (j, i): second half acts on i */
            acc.x += ai_.x; acc.y += ai_.y; acc.z += ai_.z;
        }
        v3 out = {0.0, 0.0, 0.0};
        for (int j = i + 1; j < n; ++j) {
            v3 ai, aj;
            acceleration_paired(y[i], mu[i], y[j], mu[j], &ai, &aj);
            out.x += ai.x; out.y += ai.y; out.z += ai.z;
        }
        acc.x += out.x; acc.y += out.y; acc.z += out.z;
        ddy[i] = acc;
    }
}
void orc_newtonian_gravity_eval(int n, const double *y, const double *mu, double *ddy) {
    if (g_gravity_threads > 1) gravity_eval_rows(n, (const v3 *)y, mu, (v3 *)ddy);
    else gravity_eval(n, (const v3 *)y, mu, (v3 *)ddy);
}

/* ------------------------------------------------------------------------------------------------ */
/* NBodyProblem = ODEProblem<f64, SecondOrderState<Vec<V>>, NewtonianGravity>  nbody.rs:41           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    double time, bound;
    int n;
    v3 *y, *dy;      /* state */
    double *mu;      /* ode.gravitational_parameters */
    uint64_t evals;
} problem_t;

static v3 *v3_alloc(int n) { return (v3 *)calloc((size_t)(n > 0 ? n : 1), sizeof(v3)); }
static v3 *v3_dup(const v3 *s, int n) { v3 *d = v3_alloc(n); memcpy(d, s, sizeof(v3) * (size_t)n); return d; }
static void v3_zero(v3 *v, int n) { for (int i = 0; i < n; ++i) v[i].x = v[i].y = v[i].z = 0.0; }

static void ode_eval(problem_t *p, const v3 *y, v3 *ddy_zeroed) {
    if (g_gravity_threads > 1) gravity_eval_rows(p->n, y, p->mu, ddy_zeroed);
    else gravity_eval(p->n, y, p->mu, ddy_zeroed);
    p->evals++;
}

/* ---- SRKN<C, V>  integration/src/runge_kutta/nystrom/symplectic.rs:36-102 ------------------------- */
typedef struct {
    int stages, fsal;
    double A[32], B[32];
    uint32_t i;
    v3 *ddy;
} srkn_t;

static int srkn_init(srkn_t *k, const char *name, int n) {
    const EPH_SRKN_TABLE *t = find_srkn(name);
    if (!t || t->stages > 32) return ORC_BAD_ARGUMENT;
    k->stages = t->stages; k->fsal = t->fsal; k->i = 0;
    for (int s = 0; s < t->stages; ++s) { k->A[s] = ratio_f64(t->A[s]); k->B[s] = ratio_f64(t->B[s]); }
    k->ddy = v3_alloc(n);
    return ORC_OK;
}
/* symplectic.rs:69-102 */
static void srkn_advance(srkn_t *k, double h, problem_t *p) {
    for (int s = 0; s < k->stages; ++s) {
        if (!k->fsal || s > 0 || k->i == 0) {
            v3_zero(k->ddy, p->n);                 /* self.ddy.zero() */
            ode_eval(p, p->y, k->ddy);             /* t_stage unused by NewtonianGravity */
        }
        double hb = h * k->B[s], ha = h * k->A[s];
        for (int b = 0; b < p->n; ++b) {
            p->dy[b].x = p->dy[b].x + k->ddy[b].x * hb;
            p->dy[b].y = p->dy[b].y + k->ddy[b].y * hb;
            p->dy[b].z = p->dy[b].z + k->ddy[b].z * hb;
            p->y[b].x = p->y[b].x + p->dy[b].x * ha;
            p->y[b].y = p->y[b].y + p->dy[b].y * ha;
            p->y[b].z = p->y[b].z + p->dy[b].z * ha;
        }
    }
    p->time = p->time + h;
    k->i += 1;
}
/* FixedRungeKuttaIntegrator::advance  runge_kutta/mod.rs:112-125 */
static int frk_advance(srkn_t *k, double h, problem_t *p) {
    if (p->time >= p->bound) return ORC_BOUND_REACHED;
    if (p->time + h == p->time) return ORC_STEP_SIZE_UNDERFLOW;
    srkn_advance(k, h, p);
    return ORC_OK;
}

/* ---- ELM2 + Cowell + LMBuffer ------------------------------------------------------------------ */
#define ELM_MAX_ORDER 16
typedef struct {
    int order;                 /* ORDER; ring has ORDER-1 slots */
    double wa[ELM_MAX_ORDER], wb[ELM_MAX_ORDER], cw[ELM_MAX_ORDER];
    double inv_beta_d, inv_cowell_d;
    uint32_t i;
    v3 *current_ddy;
    /* LMBuffer<StepOrder2>  multistep/buffer.rs:1-66 */
    int head, len;
    v3 **sy, **sdy, **sddy;    /* steps[k].state.y / .state.dy / .ddy */
    v3 *sum1, *sum2;
} elm2_t;

static int elm2_init(elm2_t *e, const char *name, const problem_t *p) {
    int order;
    if (orc_elm2_coeffs(name, &order, e->wa, e->wb, &e->inv_beta_d, e->cw, &e->inv_cowell_d)) return ORC_BAD_ARGUMENT;
    e->order = order; e->i = 0;
    e->len = order - 1;
    e->head = e->len;                                   /* buffer.rs:12-15: head = data.len() */
    e->current_ddy = v3_alloc(p->n);
    e->sy = calloc((size_t)e->len, sizeof(v3 *));
    e->sdy = calloc((size_t)e->len, sizeof(v3 *));
    e->sddy = calloc((size_t)e->len, sizeof(v3 *));
    for (int k = 0; k < e->len; ++k) {                  /* second_order/mod.rs:73-88 */
        e->sy[k] = v3_dup(p->y, p->n);
        e->sdy[k] = v3_dup(p->dy, p->n);
        e->sddy[k] = v3_alloc(p->n);
    }
    e->sum1 = v3_alloc(p->n);
    e->sum2 = v3_alloc(p->n);
    return ORC_OK;
}
#define SWAPP(a, b) do { v3 *t_ = (a); (a) = (b); (b) = t_; } while (0)
/* prepare_next_step  second_order/mod.rs:41-45 */
static void elm2_prepare_next_step(elm2_t *e) {
    e->head = (e->head + e->len - 1) % e->len;          /* rotate_right buffer.rs:34-36 */
    SWAPP(e->sddy[e->head], e->current_ddy);
}
/* LMBuffer::iter item `idx` (front -> oldest)  buffer.rs:39-66 */
static inline int ring_at(const elm2_t *e, int idx) { return (e->head + idx) % e->len; }

/* Cowell::<ORDER>::update_velocity  cowell.rs:17-53 */
static void cowell_update_velocity(elm2_t *e, problem_t *p, double h) {
    int n = p->n;
    v3_zero(e->sum1, n);
    for (int j = 0; j < e->order; ++j) {
        const v3 *ddy = j == 0 ? e->current_ddy : e->sddy[ring_at(e, j - 1)];
        double c = e->cw[j];
        for (int b = 0; b < n; ++b) {
            e->sum1[b].x = e->sum1[b].x + ddy[b].x * c;
            e->sum1[b].y = e->sum1[b].y + ddy[b].y * c;
            e->sum1[b].z = e->sum1[b].z + ddy[b].z * c;
        }
    }
    const v3 *ym1 = e->sy[e->head];                      /* lm.steps.front().state.y */
    double hc = h * e->inv_cowell_d;                     /* h * Ratio::from_recip(BETA_D) */
    for (int b = 0; b < n; ++b) {
        p->dy[b].x = (p->y[b].x - ym1[b].x) / h + e->sum1[b].x * hc;
        p->dy[b].y = (p->y[b].y - ym1[b].y) / h + e->sum1[b].y * hc;
        p->dy[b].z = (p->y[b].z - ym1[b].z) / h + e->sum1[b].z * hc;
    }
}
/* ELM2::advance  second_order/mod.rs:90-131 */
static void elm2_advance(elm2_t *e, double h, problem_t *p) {
    int n = p->n;
    v3_zero(e->sum1, n);
    v3_zero(e->sum2, n);
    for (int j = 0; j < e->order; ++j) {
        const v3 *y = j == 0 ? p->y : e->sy[ring_at(e, j - 1)];
        const v3 *ddy = j == 0 ? e->current_ddy : e->sddy[ring_at(e, j - 1)];
        double a = e->wa[j], bb = e->wb[j];
        for (int b = 0; b < n; ++b) {
            e->sum1[b].x = e->sum1[b].x + y[b].x * a;
            e->sum1[b].y = e->sum1[b].y + y[b].y * a;
            e->sum1[b].z = e->sum1[b].z + y[b].z * a;
            e->sum2[b].x = e->sum2[b].x + ddy[b].x * bb;
            e->sum2[b].y = e->sum2[b].y + ddy[b].y * bb;
            e->sum2[b].z = e->sum2[b].z + ddy[b].z * bb;
        }
    }
    elm2_prepare_next_step(e);
    SWAPP(e->sy[e->head], p->y);                         /* mem::swap(front.state, problem.state) */
    SWAPP(e->sdy[e->head], p->dy);
    double hh = h * h * e->inv_beta_d;                   /* h * h * Ratio::from_recip(BETA_D) */
    for (int b = 0; b < n; ++b) {
        p->y[b].x = e->sum1[b].x + e->sum2[b].x * hh;
        p->y[b].y = e->sum1[b].y + e->sum2[b].y * hh;
        p->y[b].z = e->sum1[b].z + e->sum2[b].z * hh;
    }
    p->time = p->time + h;
    v3_zero(e->current_ddy, n);
    ode_eval(p, p->y, e->current_ddy);
    cowell_update_velocity(e, p, h);
    e->i += 1;
}

/* ---- the integrator object --------------------------------------------------------------------- */
struct orc_nbody {
    problem_t p;
    double h;
    int is_multistep;
    elm2_t lm;
    srkn_t starter;      /* Substepper<4, FixedRungeKutta<BlanesMoan6B>> inner, or the plain SRKN method */
    int substeps;
    double h_sub;
};

static void problem_free(problem_t *p) { free(p->y); free(p->dy); free(p->mu); }

orc_nbody *orc_nbody_new(int n, const double *pos, const double *vel, const double *mu, double t0, double h,
                         const char *method) {
    orc_nbody *o = calloc(1, sizeof(*o));
    o->p.time = t0; o->p.bound = INFINITY; o->p.n = n;     /* nbody.rs:110-113 */
    o->p.y = v3_dup((const v3 *)pos, n);
    o->p.dy = v3_dup((const v3 *)vel, n);
    o->p.mu = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(o->p.mu, mu, sizeof(double) * (size_t)n);
    o->h = h;
    if (find_elm2(method)) {
        /* LinearMultistep::new + Substepper::new  multistep/mod.rs:54-57,120-128 ; methods.rs:37-40 */
        o->is_multistep = 1;
        o->substeps = 4;
        o->h_sub = h * (1.0 / 4.0);                     /* params.h * Ratio::from_recip(SUBSTEPS) */
        if (elm2_init(&o->lm, method, &o->p) || srkn_init(&o->starter, "BlanesMoan6B", n)) { free(o); return NULL; }
    } else if (find_srkn(method)) {
        o->is_multistep = 0;
        o->substeps = 1;
        o->h_sub = h;
        if (srkn_init(&o->starter, method, n)) { free(o); return NULL; }
    } else {
        problem_free(&o->p); free(o); return NULL;
    }
    return o;
}

orc_nbody *orc_nbody_clone(const orc_nbody *s) {
    orc_nbody *o = malloc(sizeof(*o));
    *o = *s;
    int n = s->p.n;
    o->p.y = v3_dup(s->p.y, n); o->p.dy = v3_dup(s->p.dy, n);
    o->p.mu = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(o->p.mu, s->p.mu, sizeof(double) * (size_t)n);
    o->starter.ddy = v3_dup(s->starter.ddy, n);
    if (s->is_multistep) {
        const elm2_t *e = &s->lm;
        elm2_t *d = &o->lm;
        d->current_ddy = v3_dup(e->current_ddy, n);
        d->sum1 = v3_dup(e->sum1, n); d->sum2 = v3_dup(e->sum2, n);
        d->sy = calloc((size_t)e->len, sizeof(v3 *));
        d->sdy = calloc((size_t)e->len, sizeof(v3 *));
        d->sddy = calloc((size_t)e->len, sizeof(v3 *));
        for (int k = 0; k < e->len; ++k) {
            d->sy[k] = v3_dup(e->sy[k], n); d->sdy[k] = v3_dup(e->sdy[k], n); d->sddy[k] = v3_dup(e->sddy[k], n);
        }
    }
    return o;
}

void orc_nbody_free(orc_nbody *o) {
    if (!o) return;
    problem_free(&o->p);
    free(o->starter.ddy);
    if (o->is_multistep) {
        elm2_t *e = &o->lm;
        for (int k = 0; k < e->len; ++k) { free(e->sy[k]); free(e->sdy[k]); free(e->sddy[k]); }
        free(e->sy); free(e->sdy); free(e->sddy); free(e->current_ddy); free(e->sum1); free(e->sum2);
    }
    free(o);
}

/* SubstepperIntegrator::advance  multistep/mod.rs:101-107 */
static int substepper_advance(orc_nbody *o) {
    for (int s = 0; s < o->substeps; ++s) {
        int st = frk_advance(&o->starter, o->h_sub, &o->p);
        if (st) return st;
    }
    return ORC_OK;
}
/* ELM2::advance_with  second_order/mod.rs:133-153 */
static int elm2_advance_with(orc_nbody *o, int run_starter) {
    elm2_t *e = &o->lm;
    problem_t *p = &o->p;
    elm2_prepare_next_step(e);
    memcpy(e->sy[e->head], p->y, sizeof(v3) * (size_t)p->n);    /* front.state.clone_from(problem.state) */
    memcpy(e->sdy[e->head], p->dy, sizeof(v3) * (size_t)p->n);
    if (run_starter) {
        int st = substepper_advance(o);
        if (st) return st;
    }
    v3_zero(e->current_ddy, p->n);
    ode_eval(p, p->y, e->current_ddy);
    return ORC_OK;
}
/* LinearMultistepIntegrator::advance  multistep/mod.rs:201-224 */
static int integrator_advance(orc_nbody *o) {
    problem_t *p = &o->p;
    if (!o->is_multistep) return frk_advance(&o->starter, o->h, p);
    if (p->time >= p->bound) return ORC_BOUND_REACHED;
    if (p->time + o->h == p->time) return ORC_STEP_SIZE_UNDERFLOW;
    uint32_t starter_count = o->starter.i / (uint32_t)o->substeps;   /* multistep/mod.rs:93-95 */
    if (starter_count < (uint32_t)o->lm.order) {
        if (starter_count == 0) {
            int st = elm2_advance_with(o, 0);
            if (st) return st;
        }
        return elm2_advance_with(o, 1);
    }
    elm2_advance(&o->lm, o->h, p);
    return ORC_OK;
}

int orc_nbody_advance(orc_nbody *o, int64_t nsteps) {
    for (int64_t s = 0; s < nsteps; ++s) {
        int st = integrator_advance(o);
        if (st) return st;
    }
    return ORC_OK;
}
static uint32_t integrator_step_count(const orc_nbody *o) {
    if (!o->is_multistep) return o->starter.i;
    return o->starter.i / (uint32_t)o->substeps + o->lm.i;       /* multistep/mod.rs:170-172 */
}
void orc_nbody_get_state(const orc_nbody *o, double *pos, double *vel, double *t, uint32_t *step_count) {
    if (pos) memcpy(pos, o->p.y, sizeof(v3) * (size_t)o->p.n);
    if (vel) memcpy(vel, o->p.dy, sizeof(v3) * (size_t)o->p.n);
    if (t) *t = o->p.time;
    if (step_count) *step_count = integrator_step_count(o);
}
void orc_nbody_get_acc(const orc_nbody *o, double *acc) {
    memcpy(acc, o->is_multistep ? o->lm.current_ddy : o->starter.ddy, sizeof(v3) * (size_t)o->p.n);
}
void orc_nbody_set_bound(orc_nbody *o, double bound) { o->p.bound = bound; }
uint64_t orc_nbody_eval_count(const orc_nbody *o) { return o->p.evals; }

/* ------------------------------------------------------------------------------------------------ */
/* Polynomial / UniformSpline  ephemeris/src/trajectory.rs:337-633                                  */
/* ------------------------------------------------------------------------------------------------ */
#define DIV 8                              /* trajectory.rs:335 */
typedef struct { int ncoef; v3 c[DIV]; } poly_t;    /* SmallVec<[V; 8]> */

/* eval_slice_horner trajectory.rs:398-410 */
static v3 poly_eval(const poly_t *p, double t) {
    v3 r = {0.0, 0.0, 0.0};
    for (int k = p->ncoef - 1; k >= 0; --k) {
        r.x = r.x * t + p->c[k].x; r.y = r.y * t + p->c[k].y; r.z = r.z * t + p->c[k].z;
    }
    return r;
}
/* Polynomial::eval_and_deriv trajectory.rs:368-385 */
static void poly_eval_and_deriv(const poly_t *p, double t, v3 *eval, v3 *deriv) {
    v3 zero = {0.0, 0.0, 0.0};
    v3 first = p->ncoef ? p->c[0] : zero;
    v3 last = p->ncoef ? p->c[p->ncoef - 1] : zero;
    v3 e = last, d = last;
    /* self.0.iter().skip(1).rev().skip(1): indices ncoef-2 .. 1 */
    for (int k = p->ncoef - 2; k >= 1; --k) {
        e.x = e.x * t + p->c[k].x; e.y = e.y * t + p->c[k].y; e.z = e.z * t + p->c[k].z;
        d.x = d.x * t + e.x; d.y = d.y * t + e.y; d.z = d.z * t + e.z;
    }
    e.x = e.x * t + first.x; e.y = e.y * t + first.y; e.z = e.z * t + first.z;
    *eval = e; *deriv = d;
}
void orc_poly_eval_and_deriv(int ncoef, const double *coeffs, double tau, double *val, double *deriv) {
    poly_t p; p.ncoef = ncoef;
    memcpy(p.c, coeffs, sizeof(v3) * DIV);
    v3 e, d;
    poly_eval_and_deriv(&p, tau, &e, &d);
    memcpy(val, &e, sizeof e); memcpy(deriv, &d, sizeof d);
}

typedef struct {
    double start, interval;
    /* VecDeque<Polynomial>: stored with a movable front */
    poly_t *buf; int64_t cap, off, len;
} spline_t;

static void spline_init(spline_t *s, double start, double interval) {
    s->start = start; s->interval = interval; s->buf = NULL; s->cap = s->off = s->len = 0;
}
static void spline_reserve(spline_t *s, int64_t front, int64_t back) {
    if (s->off >= front && s->off + s->len + back <= s->cap) return;
    int64_t ncap = (s->len + front + back) * 2 + 16;
    poly_t *nb = malloc(sizeof(poly_t) * (size_t)ncap);
    int64_t noff = (ncap - s->len) / 2;
    if (noff < front) noff = front;
    if (s->len) memcpy(nb + noff, s->buf + s->off, sizeof(poly_t) * (size_t)s->len);
    free(s->buf); s->buf = nb; s->cap = ncap; s->off = noff;
}
static void spline_push_back(spline_t *s, const poly_t *p) {      /* trajectory.rs:510-513 */
    spline_reserve(s, 0, 1);
    s->buf[s->off + s->len++] = *p;
}
static void spline_push_front(spline_t *s, const poly_t *p) {     /* trajectory.rs:504-508 */
    spline_reserve(s, 1, 0);
    s->buf[--s->off] = *p; s->len++;
    s->start -= s->interval;
}
static double spline_span(const spline_t *s) { return s->interval * (double)s->len; }  /* :624-627 scaled */
static double spline_end(const spline_t *s) { return s->start + spline_span(s); }      /* :437-440 */
/* get_polynomial trajectory.rs:551-561 with get_index_local_exclusive :600-607, index_local_exclusive :614-617 */
static const poly_t *spline_get_polynomial(const spline_t *s, double at, double *tau) {
    double local = at - s->start;
    if (signbit(local) || local > spline_span(s)) return NULL;    /* Duration::is_negative = sign bit */
    double q = ceil(local / s->interval);
    /* `as usize` saturating cast, then saturating_sub(1) */
    uint64_t qi = q <= 0.0 ? 0 : (q >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)q);
    uint64_t idx = qi == 0 ? 0 : qi - 1;
    double local_polynomial = local - s->interval * (double)idx;
    *tau = local_polynomial / s->interval;
    if (idx >= (uint64_t)s->len) return NULL;                     /* self.polynomials.get(local_index)? */
    return &s->buf[s->off + (int64_t)idx];
}

struct orc_solution { int n; spline_t *s; };

static orc_solution *solution_alloc(int n) {
    orc_solution *so = malloc(sizeof(*so));
    so->n = n; so->s = calloc((size_t)(n > 0 ? n : 1), sizeof(spline_t));
    return so;
}
void orc_solution_free(orc_solution *so) {
    if (!so) return;
    for (int b = 0; b < so->n; ++b) free(so->s[b].buf);
    free(so->s); free(so);
}
static orc_solution *solution_clone(const orc_solution *src) {
    orc_solution *so = solution_alloc(src->n);
    for (int b = 0; b < src->n; ++b) {
        so->s[b] = src->s[b];
        so->s[b].buf = NULL; so->s[b].cap = so->s[b].off = 0;
        if (src->s[b].len) {
            so->s[b].buf = malloc(sizeof(poly_t) * (size_t)src->s[b].len);
            memcpy(so->s[b].buf, src->s[b].buf + src->s[b].off, sizeof(poly_t) * (size_t)src->s[b].len);
            so->s[b].cap = src->s[b].len;
        }
    }
    return so;
}
int orc_solution_bodies(const orc_solution *so) { return so->n; }
void orc_solution_info(const orc_solution *so, int body, double *start, double *interval, int64_t *npoly) {
    *start = so->s[body].start; *interval = so->s[body].interval; *npoly = so->s[body].len;
}
void orc_solution_coeffs(const orc_solution *so, int body, double *coeffs, int32_t *ncoef) {
    const spline_t *s = &so->s[body];
    for (int64_t p = 0; p < s->len; ++p) {
        const poly_t *q = &s->buf[s->off + p];
        ncoef[p] = q->ncoef;
        for (int k = 0; k < DIV; ++k) {
            v3 c = k < q->ncoef ? q->c[k] : (v3){0.0, 0.0, 0.0};
            coeffs[(p * DIV + k) * 3 + 0] = c.x; coeffs[(p * DIV + k) * 3 + 1] = c.y; coeffs[(p * DIV + k) * 3 + 2] = c.z;
        }
    }
}
/* EvaluateTrajectory for UniformSpline  trajectory.rs:450-471 */
int orc_solution_eval(const orc_solution *so, int body, double at, double *pos, double *vel) {
    const spline_t *s = &so->s[body];
    double tau;
    const poly_t *p = spline_get_polynomial(s, at, &tau);
    if (!p) return 0;
    if (vel) {
        v3 e, d;
        poly_eval_and_deriv(p, tau, &e, &d);
        pos[0] = e.x; pos[1] = e.y; pos[2] = e.z;
        vel[0] = d.x / s->interval; vel[1] = d.y / s->interval; vel[2] = d.z / s->interval;
    } else {
        v3 e = poly_eval(p, tau);
        pos[0] = e.x; pos[1] = e.y; pos[2] = e.z;
    }
    return 1;
}
/* UniformSpline::append (direction>0) / prepend (direction<0)  trajectory.rs:515-539 */
int orc_solution_append(orc_solution *a, const orc_solution *b, int direction) {
    if (a->n != b->n) return 0;
    for (int k = 0; k < a->n; ++k) {
        spline_t *x = &a->s[k];
        const spline_t *y = &b->s[k];
        if (x->interval != y->interval) return 0;
        if (direction > 0) {
            if (spline_end(x) != y->start) return 0;
            spline_reserve(x, 0, y->len);
            for (int64_t p = 0; p < y->len; ++p) x->buf[x->off + x->len++] = y->buf[y->off + p];
        } else {
            if (x->start != spline_end(y)) return 0;
            x->start = y->start;
            spline_reserve(x, y->len, 0);
            for (int64_t p = y->len - 1; p >= 0; --p) { x->buf[--x->off] = y->buf[y->off + p]; x->len++; }
        }
    }
    return 1;
}
/* UniformSpline::clear_before (after = 0, trajectory.rs:536-542) / clear_after (after != 0, :544-549) with get_index_local
 * (:591-598: None when negative or time >= span) and get_index_local_exclusive (:600-607: None when negative or time > span);
 * body < 0: every spline */
static uint64_t as_usize(double x) { return !(x > 0.0) ? 0 : (x >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)x); }
void orc_solution_clear(orc_solution *so, int body, double at, int after) {
    for (int k = 0; k < so->n; ++k) {
        if (body >= 0 && body != k) continue;
        spline_t *s = &so->s[k];
        if (after) {
            const double time = at - s->start;                       /* get_index(at) */
            if (signbit(time) || time >= spline_span(s)) continue;
            const uint64_t idx = as_usize(time / s->interval);       /* index_local :609-612 */
            if (idx < (uint64_t)s->len) s->len = (int64_t)idx;       /* truncate(idx) */
        } else {
            const double time = (at + s->interval) - s->start;       /* get_index_exclusive(at + self.interval) */
            if (signbit(time) || time > spline_span(s)) continue;
            const uint64_t c = as_usize(ceil(time / s->interval));   /* index_local_exclusive :614-617 */
            const uint64_t idx = c == 0 ? 0 : c - 1;
            s->start += s->interval * (double)idx;                   /* self.start += self.interval.scaled(idx as f64) */
            const int64_t drop = idx < (uint64_t)s->len ? (int64_t)idx : s->len;   /* drain(0..idx) */
            s->off += drop; s->len -= drop;
        }
    }
}
orc_solution *orc_solution_clone(const orc_solution *src) { return solution_clone(src); }

/* ------------------------------------------------------------------------------------------------ */
/* LeastSquaresFit::interpolate  ephemeris_explorer/src/dynamics/celestial.rs:24-135                */
/* DVec3 quantities whose three components are always identical (gamma, b, c, p_k, px) are carried  */
/* as scalars: every component undergoes the same f64 operations, so the bits are the same.          */
/* ------------------------------------------------------------------------------------------------ */
static int least_squares_fit(int degree_in, int m, const double *ts, const v3 *xs, poly_t *out) {
    v3 d0 = {0, 0, 0};
    double gamma0 = 0.0, b0 = 0.0;
    for (int k = 0; k < m; ++k) {                        /* :32-40 */
        d0.x += xs[k].x; d0.y += xs[k].y; d0.z += xs[k].z;
        gamma0 += 1.0;
        b0 += ts[k];
    }
    if (gamma0 == 0.0) return -1;                        /* :42-44 */
    int degree = degree_in < m - 1 ? degree_in : m - 1;  /* :46 */
    b0 /= gamma0;
    d0.x /= gamma0; d0.y /= gamma0; d0.z /= gamma0;
    memset(out, 0, sizeof(*out));
    if (degree == 0) { out->ncoef = 1; out->c[0] = d0; return 1; }   /* :53-56 */
    if (degree + 1 > DIV) return -2;                     /* SmallVec<[V;8]> would spill; never in the app */
    int nco = degree + 1;
    v3 pdata[DIV + 1];
    double pa[DIV + 2], pb[DIV + 2];
    double *p_km1 = pa, *p_k = pb;
    for (int i = 0; i < nco; ++i) { pdata[i] = (v3){0, 0, 0}; pa[i] = 0.0; pb[i] = 0.0; }
    pdata[0] = d0;
    p_k[0] = 1.0;
    double gamma_k = gamma0, b_k = b0, minus_c_k = 0.0;
    int kp1 = 1;
    for (;;) {
        for (int i = 0; i < kp1; ++i) p_km1[i] = minus_c_k * p_km1[i] - b_k * p_k[i];   /* :76-80 */
        for (int im1 = 0; im1 < kp1; ++im1) p_km1[im1 + 1] += p_k[im1];                 /* :82-86 */
        v3 d = {0, 0, 0};
        double g = 0.0, bsum = 0.0;
        for (int k = 0; k < m; ++k) {                    /* :88-103 */
            double px = 0.0;
            for (int c = kp1; c >= 0; --c) px = px * ts[k] + p_km1[c];   /* eval_slice_horner */
            double wipx = px;
            d.x += xs[k].x * wipx; d.y += xs[k].y * wipx; d.z += xs[k].z * wipx;
            double wipxpx = wipx * px;
            g += wipxpx;
            bsum += ts[k] * wipxpx;
        }
        if (g == 0.0) break;                             /* :105-107 */
        d.x /= g; d.y /= g; d.z /= g;
        for (int i = 0; i < kp1 + 1; ++i) {              /* :111-115 */
            pdata[i].x += d.x * p_km1[i]; pdata[i].y += d.y * p_km1[i]; pdata[i].z += d.z * p_km1[i];
        }
        if (kp1 == degree) break;
        bsum /= g;
        kp1 += 1;
        b_k = bsum;
        minus_c_k = -(g / gamma_k);
        gamma_k = g;
        double *t = p_k; p_k = p_km1; p_km1 = t;
    }
    int ncoef = nco;                                     /* Polynomial::trim trajectory.rs:387-395 */
    while (ncoef > 0 && pdata[ncoef - 1].x == 0.0 && pdata[ncoef - 1].y == 0.0 && pdata[ncoef - 1].z == 0.0) ncoef--;
    out->ncoef = ncoef;
    for (int i = 0; i < ncoef; ++i) out->c[i] = pdata[i];
    return ncoef;
}
int orc_least_squares_fit(int degree, int m, const double *ts, const double *xs, double *coeffs) {
    poly_t p;
    int r = least_squares_fit(degree, m, ts, (const v3 *)xs, &p);
    if (r < 0) return r;
    for (int k = 0; k < DIV; ++k) {
        v3 c = k < p.ncoef ? p.c[k] : (v3){0, 0, 0};
        coeffs[k * 3] = c.x; coeffs[k * 3 + 1] = c.y; coeffs[k * 3 + 2] = c.z;
    }
    return p.ncoef;
}

/* ------------------------------------------------------------------------------------------------ */
/* SplineInterpolators solout + NBodyPropagator  ephemeris/src/propagators/nbody.rs:243-517         */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    double last_sample_time, sample_period;   /* Durations */
    int index;                                /* PolyonmialInterpolator.index */
    v3 samples[DIV + 1];
    int degree;                               /* LeastSquaresFit.degree */
} interp_t;

struct orc_prop {
    orc_nbody *integ;       /* Integration.problem + .integrator */
    int direction;
    double delta;           /* SplineInterpolators.delta (positive) */
    interp_t *interp;
    orc_solution *solution; /* Integration.solution */
};

/* D::offset  propagators/mod.rs:49-51,84-86 */
static double dir_offset(int direction, double to, double duration) {
    return direction > 0 ? to + duration : to - duration;
}
/* D::distance :54-56,89-91 */
static double dir_distance(int direction, double from, double to) { return direction > 0 ? to - from : from - to; }
/* SplineInterpolator::time  nbody.rs:317-323 */
static double interp_time(const interp_t *it) {
    int len1 = it->index > 0 ? it->index - 1 : 0;
    return it->last_sample_time + it->sample_period * (double)len1;
}
/* Solout::new_solution  nbody.rs:454-469 */
static orc_solution *prop_new_solution(const orc_prop *pr) {
    int n = pr->integ->p.n;
    orc_solution *so = solution_alloc(n);
    for (int b = 0; b < n; ++b)
        spline_init(&so->s[b], dir_offset(pr->direction, pr->integ->p.time, -interp_time(&pr->interp[b])),
                    pr->interp[b].sample_period * (double)DIV);
    return so;
}
static double spline_bound(const spline_t *s, int direction) { return direction > 0 ? spline_end(s) : s->start; }

orc_prop *orc_prop_new(int n, const double *pos, const double *vel, const double *mu, double t0, double dt,
                       int direction, const char *method, const uint32_t *count, const uint32_t *degree) {
    orc_prop *pr = calloc(1, sizeof(*pr));
    double delta = fabs(dt);                                     /* Forward::new / Backward::new: delta.abs() */
    double h = direction > 0 ? delta : -delta;                   /* signed_delta */
    pr->integ = orc_nbody_new(n, pos, vel, mu, t0, h, method);
    if (!pr->integ) { free(pr); return NULL; }
    pr->direction = direction > 0 ? 1 : -1;
    pr->delta = dt;                                              /* SplineInterpolators::new(delta, ..) celestial.rs:183 */
    pr->interp = calloc((size_t)(n > 0 ? n : 1), sizeof(interp_t));
    for (int b = 0; b < n; ++b) {
        interp_t *it = &pr->interp[b];
        it->last_sample_time = 0.0;
        it->sample_period = dt * (double)count[b];               /* load/mod.rs:325 */
        it->index = 1;                                           /* PolyonmialInterpolator::new nbody.rs:251-259 */
        for (int k = 0; k <= DIV; ++k) it->samples[k] = ((const v3 *)pos)[b];
        it->degree = (int)degree[b];
    }
    pr->solution = prop_new_solution(pr);                        /* with_solout lib.rs:441-451 */
    return pr;
}
orc_prop *orc_prop_clone(const orc_prop *s) {
    orc_prop *pr = malloc(sizeof(*pr));
    *pr = *s;
    pr->integ = orc_nbody_clone(s->integ);
    int n = s->integ->p.n;
    pr->interp = malloc(sizeof(interp_t) * (size_t)(n > 0 ? n : 1));
    memcpy(pr->interp, s->interp, sizeof(interp_t) * (size_t)n);
    pr->solution = solution_clone(s->solution);
    return pr;
}
void orc_prop_free(orc_prop *pr) {
    if (!pr) return;
    orc_nbody_free(pr->integ); free(pr->interp); orc_solution_free(pr->solution); free(pr);
}
/* SplineInterpolators::solout  nbody.rs:371-400,471-489 */
static int prop_solout(orc_prop *pr) {
    const problem_t *p = &pr->integ->p;
    for (int b = 0; b < p->n; ++b) {
        interp_t *it = &pr->interp[b];
        spline_t *traj = &pr->solution->s[b];
        it->last_sample_time += pr->delta;
        if (it->last_sample_time == it->sample_period) {
            it->last_sample_time = 0.0;
            if (it->index >= DIV + 1) abort();                   /* assert!(self.index < LEN) nbody.rs:266 */
            it->samples[it->index++] = p->y[b];
            if (it->index == DIV + 1) {                          /* try_to_polynomial: is_full */
                double ts[DIV + 1];
                for (int i = 0; i <= DIV; ++i)                   /* SplineBound::samples nbody.rs:422-424,439-441 */
                    ts[i] = pr->direction > 0 ? (double)i / (double)DIV : 1.0 - (double)i / (double)DIV;
                poly_t poly;
                if (least_squares_fit(it->degree, DIV + 1, ts, it->samples, &poly) < 0) return 0;
                if (pr->direction > 0) spline_push_back(traj, &poly); else spline_push_front(traj, &poly);
                v3 t0 = it->samples[0];                          /* finish(): swap(0, index-1); index = 1 */
                it->samples[0] = it->samples[it->index - 1];
                it->samples[it->index - 1] = t0;
                it->index = 1;
            }
        }
    }
    return 1;
}
int orc_prop_step(orc_prop *pr) {
    int st = integrator_advance(pr->integ);                      /* Integration::advance lib.rs:496-503 */
    if (st) return st;
    if (!prop_solout(pr)) return ORC_SOLOUT_EXIT;                /* nbody.rs:201-204 */
    return ORC_OK;
}
/* n x IncrementalPropagator::step, stopping at the first error (a loop for callers that time the restatement: one call from
 * Python per step would be half of what is measured) */
int orc_prop_step_n(orc_prop *pr, int64_t n) {
    for (int64_t k = 0; k < n; ++k) {
        int st = orc_prop_step(pr);
        if (st) return st;
    }
    return ORC_OK;
}
/* DirectionalSolout::solution_time  nbody.rs:501-508 */
double orc_prop_time(const orc_prop *pr) {
    int n = pr->solution->n;
    if (n == 0) return dir_offset(pr->direction, 0.0, -1.7976931348623157e308);
    double best = spline_bound(&pr->solution->s[0], pr->direction);
    for (int b = 1; b < n; ++b) {                                /* min_by(D::cmp): keeps the first minimum */
        double x = spline_bound(&pr->solution->s[b], pr->direction);
        /* D::cmp(x, best) = 0.cmp(distance(x, best)); Less  <=>  0 < distance(x,best) */
        if (0.0 < dir_distance(pr->direction, x, best)) best = x;
    }
    return best;
}
/* DirectionalSolout::has_reached  nbody.rs:510-516 */
int orc_prop_has_reached(const orc_prop *pr, double t) {
    for (int b = 0; b < pr->solution->n; ++b) {
        double bound = spline_bound(&pr->solution->s[b], pr->direction);
        /* D::cmp(&bound, &time).is_ge()  <=>  !(0 < distance(bound, time)) */
        if (0.0 < dir_distance(pr->direction, bound, t)) return 0;
    }
    return 1;
}
int orc_prop_step_to(orc_prop *pr, double t) {
    for (;;) {
        if (orc_prop_has_reached(pr, t)) return ORC_OK;
        int st = orc_prop_step(pr);
        if (st) return st;
    }
}
double orc_prop_integrator_time(const orc_prop *pr) { return pr->integ->p.time; }
void orc_prop_get_state(const orc_prop *pr, double *pos, double *vel, double *t, uint32_t *sc) {
    orc_nbody_get_state(pr->integ, pos, vel, t, sc);
}
/* Propagator::take_solution  nbody.rs:182-189 */
orc_solution *orc_prop_take_solution(orc_prop *pr) {
    orc_solution *old = pr->solution;
    pr->solution = prop_new_solution(pr);
    return old;
}

/* ================================================================================================== */
/* Massless path                                                                                      */
/* ================================================================================================== */
#define ERK_MAX_STAGES 16
#define ERK_MAX_DIM 6

/* ODE right-hand side over a flat state of `dim` doubles: returns 0 or ORC_EVAL_FAILED */
typedef int (*ode_fn)(void *ctx, double t, const double *y, double *dy);

/* ERK<C, [State; STAGES]> + EERK  integration/src/runge_kutta/explicit.rs:40-141 */
typedef struct {
    int stages, order, order_embedded, fsal, dim;
    /* nystrom = 0: ERK (A, B, C, E). nystrom = 1: ERKNG on a SecondOrderState (y = state[0..2], dy = state[3..5]):
     * A = AP, A2 = AV, B = BP, B2 = BV, E = EP, E2 = EV; k[s][0..2] = dk[s]. nystrom = 2: ERKN (nystrom/explicit.rs),
     * the same without AV */
    int nystrom;
    double A[ERK_MAX_STAGES][ERK_MAX_STAGES], B[ERK_MAX_STAGES], C[ERK_MAX_STAGES], E[ERK_MAX_STAGES];
    double A2[ERK_MAX_STAGES][ERK_MAX_STAGES], B2[ERK_MAX_STAGES], E2[ERK_MAX_STAGES];
    uint32_t i;
    double k[ERK_MAX_STAGES][ERK_MAX_DIM];
} erk_t;
static const EPH_ERKNG_TABLE *find_erkng(const char *name) {
    for (int i = 0; i < EPH_N_ERKNG_TABLES; ++i)
        if (!strcmp(eph_erkng_tables[i].name, name)) return &eph_erkng_tables[i];
    return NULL;
}

static const EPH_ERKN_TABLE *find_erkn(const char *name) {
    for (int i = 0; i < EPH_N_ERKN_TABLES; ++i)
        if (!strcmp(eph_erkn_tables[i].name, name)) return &eph_erkn_tables[i];
    return NULL;
}

static int erk_init(erk_t *r, const char *name, int dim, const double *state) {
    const EPH_ERKN_TABLE *n = find_erkn(name);
    if (n) {                                                  /* ERKN<C, [V; STAGES], V>::from_problem  nystrom/explicit.rs:64-72 */
        if (n->stages > ERK_MAX_STAGES || dim != 6) return ORC_BAD_ARGUMENT;
        memset(r, 0, sizeof(*r));
        r->nystrom = 2;                                       /* y'' = f(t, y): no stage velocities, no AV */
        r->stages = n->stages; r->order = n->order; r->order_embedded = n->order_embedded; r->fsal = n->fsal; r->dim = 6;
        int idx = 0;
        for (int s = 0; s < n->stages; ++s) {
            for (int j = 0; j < s; ++j) r->A[s][j] = ratio_f64(n->A[idx++]);
            r->B[s] = ratio_f64(n->BP[s]); r->B2[s] = ratio_f64(n->BV[s]); r->C[s] = ratio_f64(n->C[s]);
            r->E[s] = ratio_f64(n->EP[s]); r->E2[s] = ratio_f64(n->EV[s]);
        }
        for (int s = 0; s < n->stages; ++s)                   /* dk = [state.dy.clone(); STAGES] */
            for (int d = 0; d < 3; ++d) r->k[s][d] = state[3 + d];
        return ORC_OK;
    }
    const EPH_ERKNG_TABLE *g = find_erkng(name);
    if (g) {                                                  /* ERKNG<C, [V; STAGES], V>::from_problem  explicit_generalized.rs:87-95 */
        if (g->stages > ERK_MAX_STAGES || dim != 6) return ORC_BAD_ARGUMENT;
        memset(r, 0, sizeof(*r));
        r->nystrom = 1;
        r->stages = g->stages; r->order = g->order; r->order_embedded = g->order_embedded; r->fsal = g->fsal; r->dim = 6;
        int idx = 0;
        for (int s = 0; s < g->stages; ++s) {
            for (int j = 0; j < s; ++j) { r->A[s][j] = ratio_f64(g->AP[idx]); r->A2[s][j] = ratio_f64(g->AV[idx]); ++idx; }
            r->B[s] = ratio_f64(g->BP[s]); r->B2[s] = ratio_f64(g->BV[s]); r->C[s] = ratio_f64(g->C[s]);
            r->E[s] = ratio_f64(g->EP[s]); r->E2[s] = ratio_f64(g->EV[s]);
        }
        for (int s = 0; s < g->stages; ++s)                   /* dk = [state.dy.clone(); STAGES] */
            for (int d = 0; d < 3; ++d) r->k[s][d] = state[3 + d];
        return ORC_OK;
    }
    const EPH_ERK_TABLE *t = find_erk(name);
    if (!t || t->stages > ERK_MAX_STAGES || dim > ERK_MAX_DIM) return ORC_BAD_ARGUMENT;
    memset(r, 0, sizeof(*r));
    r->stages = t->stages; r->order = t->order; r->order_embedded = t->order_embedded; r->fsal = t->fsal; r->dim = dim;
    int idx = 0;
    for (int s = 0; s < t->stages; ++s) {
        for (int j = 0; j < s; ++j) r->A[s][j] = ratio_f64(t->A[idx++]);
        r->B[s] = ratio_f64(t->B[s]);
        r->C[s] = ratio_f64(t->C[s]);
        r->E[s] = t->E ? ratio_f64(t->E[s]) : 0.0;
    }
    r->i = 0;
    for (int s = 0; s < t->stages; ++s)                       /* from_problem: k = [state.clone(); STAGES] :65-69 */
        for (int d = 0; d < dim; ++d) r->k[s][d] = state[d];
    return ORC_OK;
}
/* ERK::advance  explicit.rs:72-106 */
/* ERKNG::advance  integration/src/runge_kutta/nystrom/explicit_generalized.rs:97-143 (SecondOrderODEGeneral) */
static int erkng_advance(erk_t *r, double h, double *time, double *state, ode_fn f, void *ctx, uint64_t *evals) {
    const int S = r->stages;
    double *y = state, *dy = state + 3;
    for (int s = 0; s < S; ++s) {
        if (r->fsal && s == 0 && r->i > 0) {                  /* self.dk.swap(s, STAGES - 1); continue */
            for (int d = 0; d < 3; ++d) { double t = r->k[0][d]; r->k[0][d] = r->k[S - 1][d]; r->k[S - 1][d] = t; }
            continue;
        }
        const double ti = *time + h * r->C[s];
        double sv[6], out[6];
        const double hc = h * r->C[s];
        for (int d = 0; d < 3; ++d) { sv[d] = y[d]; sv[d] = sv[d] + dy[d] * hc; sv[3 + d] = dy[d]; }
        for (int j = 0; j < s; ++j) {
            const double hhap = h * h * r->A[s][j], hav = h * r->A2[s][j];
            for (int d = 0; d < 3; ++d) {
                sv[d] = sv[d] + r->k[j][d] * hhap;
                sv[3 + d] = sv[3 + d] + r->k[j][d] * hav;
            }
        }
        if (evals) (*evals)++;
        int st = f(ctx, ti, sv, out);                         /* ddy = context + manoeuvre (spacecraft.rs:319-331) */
        if (st) {                                             /* `self.dk[s].zero()` ran before eval returned Err  :109-111 */
            for (int d = 0; d < 3; ++d) r->k[s][d] = 0.0;
            return st;
        }
        for (int d = 0; d < 3; ++d) r->k[s][d] = out[3 + d];
    }
    for (int d = 0; d < 3; ++d) y[d] = y[d] + dy[d] * h;
    for (int i = 0; i < S; ++i) {
        const double hhbp = h * h * r->B[i], hbv = h * r->B2[i];
        for (int d = 0; d < 3; ++d) {
            y[d] = y[d] + r->k[i][d] * hhbp;
            dy[d] = dy[d] + r->k[i][d] * hbv;
        }
    }
    *time = *time + h;
    r->i += 1;
    return ORC_OK;
}
/* ERKN::advance  integration/src/runge_kutta/nystrom/explicit.rs:74-121 (SecondOrderODE: y'' = f(t, y)). The
 * right-hand side is handed the UNCHANGED current velocity in sv[3..5] only because ode_fn takes a flat 6-vector; a
 * SecondOrderODE never reads it (craft_rhs does only for frame-relative burns, which orc_craft_new refuses here). */
static int erkn_advance(erk_t *r, double h, double *time, double *state, ode_fn f, void *ctx, uint64_t *evals) {
    const int S = r->stages;
    double *y = state, *dy = state + 3;
    for (int s = 0; s < S; ++s) {
        if (r->fsal && s == 0 && r->i > 0) {                  /* self.dk.swap(s, STAGES - 1); continue  :79-82 */
            for (int d = 0; d < 3; ++d) { double t = r->k[0][d]; r->k[0][d] = r->k[S - 1][d]; r->k[S - 1][d] = t; }
            continue;
        }
        const double ti = *time + h * r->C[s];
        double sv[6], out[6];
        const double hc = h * r->C[s];
        for (int d = 0; d < 3; ++d) { sv[d] = y[d]; sv[d] = sv[d] + dy[d] * hc; sv[3 + d] = dy[d]; }   /* :87-90 */
        for (int j = 0; j < s; ++j) {
            const double hha = h * h * r->A[s][j];            /* :91-95 */
            for (int d = 0; d < 3; ++d) sv[d] = sv[d] + r->k[j][d] * hha;
        }
        if (evals) (*evals)++;
        int st = f(ctx, ti, sv, out);                         /* problem.ode.eval(ti, &self.yi, self.dk[s].zero())  :97 */
        if (st) {                                             /* the zeroed dk[s] stays */
            for (int d = 0; d < 3; ++d) r->k[s][d] = 0.0;
            return st;
        }
        for (int d = 0; d < 3; ++d) r->k[s][d] = out[3 + d];
    }
    for (int d = 0; d < 3; ++d) y[d] = y[d] + dy[d] * h;      /* :102-104 */
    for (int i = 0; i < S; ++i) {                             /* :105-116 */
        const double hhbp = h * h * r->B[i], hbv = h * r->B2[i];
        for (int d = 0; d < 3; ++d) {
            y[d] = y[d] + r->k[i][d] * hhbp;
            dy[d] = dy[d] + r->k[i][d] * hbv;
        }
    }
    *time = *time + h;
    r->i += 1;
    return ORC_OK;
}
static int erk_advance(erk_t *r, double h, double *time, double *state, ode_fn f, void *ctx, uint64_t *evals) {
    if (r->nystrom == 2) return erkn_advance(r, h, time, state, f, ctx, evals);
    if (r->nystrom) return erkng_advance(r, h, time, state, f, ctx, evals);
    double yi[ERK_MAX_DIM];
    const int S = r->stages, D = r->dim;
    for (int s = 0; s < S; ++s) {
        if (r->fsal && s == 0 && r->i > 0) {                  /* self.k.swap(s, STAGES - 1); continue */
            for (int d = 0; d < D; ++d) { double t = r->k[0][d]; r->k[0][d] = r->k[S - 1][d]; r->k[S - 1][d] = t; }
            continue;
        }
        const double ti = *time + h * r->C[s];
        for (int d = 0; d < D; ++d) yi[d] = state[d];
        for (int j = 0; j < s; ++j) {
            const double ha = h * r->A[s][j];
            for (int d = 0; d < D; ++d) yi[d] = yi[d] + r->k[j][d] * ha;
        }
        for (int d = 0; d < D; ++d) r->k[s][d] = 0.0;         /* self.k[s].zero() */
        if (evals) (*evals)++;
        int st = f(ctx, ti, yi, r->k[s]);
        if (st) return st;
    }
    for (int i = 0; i < S; ++i) {
        const double hb = h * r->B[i];
        for (int d = 0; d < D; ++d) state[d] = state[d] + r->k[i][d] * hb;
    }
    *time = *time + h;
    r->i += 1;
    return ORC_OK;
}
/* RKEmbedded::error  explicit.rs:123-132 */
static void erk_error(const erk_t *r, double h, double *err) {
    for (int d = 0; d < r->dim; ++d) err[d] = 0.0;
    if (r->nystrom) {                                         /* explicit_generalized.rs:153-170 = explicit.rs:137-157 */
        for (int i = 0; i < r->stages; ++i) {
            const double hhep = h * h * r->E[i], hev = h * r->E2[i];
            for (int d = 0; d < 3; ++d) {
                err[d] = err[d] + r->k[i][d] * hhep;
                err[3 + d] = err[3 + d] + r->k[i][d] * hev;
            }
        }
        return;
    }
    for (int i = 0; i < r->stages; ++i) {
        const double he = h * r->E[i];
        for (int d = 0; d < r->dim; ++d) err[d] = err[d] + r->k[i][d] * he;
    }
}

/* Tolerance models */
typedef double (*tol_fn)(void *ctx, const double *state, const double *err);

/* AdaptiveRungeKuttaIntegrator + IController + PreviousStep  runge_kutta/mod.rs:188-285,396-440 */
typedef struct {
    erk_t rk;                /* frk.rk */
    double frk_h, next_h;
    double error[ERK_MAX_DIM];
    /* PreviousStep */
    double prev_t, prev_y[ERK_MAX_DIM], prev_klast[ERK_MAX_DIM];
    uint32_t prev_i;
    double fac_min, fac_max, fac, h_max;
    uint32_t n, n_max;
} adaptive_t;

static int adaptive_init(adaptive_t *a, const char *method, int dim, const double *state, double t, double h_init,
                         double h_max, double fac_min, double fac_max, double fac, uint32_t n_max) {
    int st = erk_init(&a->rk, method, dim, state);           /* FixedRungeKutta::new(h_init).init(problem) */
    if (st) return st;
    a->frk_h = h_init; a->next_h = h_init;
    for (int d = 0; d < dim; ++d) { a->error[d] = state[d]; a->prev_y[d] = state[d]; a->prev_klast[d] = state[d]; }
    a->prev_t = t; a->prev_i = 0;
    a->fac_min = fac_min; a->fac_max = fac_max; a->fac = fac; a->h_max = h_max;
    a->n = 0; a->n_max = n_max;
    return ORC_OK;
}

/* ---- powf in the step-size controller -------------------------------------------------------------
 * `err.pow(-k.inv())` is f64::powf, i.e. the platform libm's pow: the one operation on the hot path whose bits
 * depend on the platform (glibc picks an FMA or non-FMA variant at run time and is documented as < 0.52 ULP, not
 * correctly rounded). The oracle pins it to the correctly rounded value, computed in double-double arithmetic
 * (log by the atanh series, exp by Taylor series, ~100 bits); the product evaluates the identical sequence on the
 * device. tests/test_oracle.py checks it against this host's libm pow (equal on every sampled input) and that the
 * reference scenarios are unchanged when libm's pow is used instead (orc_set_pow_mode). */
typedef struct { double hi, lo; } dd_t;
static inline dd_t dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return (dd_t){s, (a - (s - bb)) + (b - bb)};
}
static inline dd_t dd_quick(double a, double b) { const double s = a + b; return (dd_t){s, b - (s - a)}; }
static inline dd_t dd_two_prod(double a, double b) { const double p = a * b; return (dd_t){p, fma(a, b, -p)}; }
static inline dd_t dd_add(dd_t a, dd_t b) {
    dd_t s = dd_two_sum(a.hi, b.hi);
    const dd_t t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick(s.hi, s.lo);
}
static inline dd_t dd_mul(dd_t a, dd_t b) {
    dd_t p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick(p.hi, p.lo);
}
static inline dd_t dd_mul_d(dd_t a, double b) {
    dd_t p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick(p.hi, p.lo);
}
static inline dd_t dd_div(dd_t a, dd_t b) {
    const double q1 = a.hi / b.hi;
    dd_t r = dd_add(a, (dd_t){-dd_mul_d(b, q1).hi, -dd_mul_d(b, q1).lo});
    const double q2 = r.hi / b.hi;
    r = dd_add(r, (dd_t){-dd_mul_d(b, q2).hi, -dd_mul_d(b, q2).lo});
    const double q3 = r.hi / b.hi;
    dd_t q = dd_quick(q1, q2);
    return dd_add(q, (dd_t){q3, 0.0});
}
static const dd_t DD_LN2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
#define EPH_POW_CONST const
typedef struct { double hi, lo; } EPH_POW_DD;
#include "cr_pow_tables.inc"

static double cr_pow(double x, double y) {
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0.0) return y < 0.0 ? INFINITY : 0.0;
    if (isinf(x)) return y < 0.0 ? 0.0 : INFINITY;
    if (x < 0.0) return NAN;
    /* log(x) = e*ln2 + 2*atanh(s), s = (m-1)/(m+1), m in [sqrt(1/2), sqrt(2)) */
    int e;
    double m = frexp(x, &e);
    if (m < 0x1.6a09e667f3bcdp-1) { m *= 2.0; e -= 1; }
    const dd_t s = dd_div((dd_t){m - 1.0, 0.0}, dd_two_sum(m, 1.0));
    const dd_t s2 = dd_mul(s, s);
    /* atanh(s)/s = sum_k s2^k / (2k+1), Horner over the double-double table (remainder < 2^-120) */
    dd_t sum = {eph_pow_atanh[EPH_POW_TERMS - 1].hi, eph_pow_atanh[EPH_POW_TERMS - 1].lo};
    for (int k = EPH_POW_TERMS - 2; k >= 0; --k)
        sum = dd_add(dd_mul(sum, s2), (dd_t){eph_pow_atanh[k].hi, eph_pow_atanh[k].lo});
    dd_t lg = dd_mul(dd_mul_d(s, 2.0), sum);
    lg = dd_add(dd_mul_d(DD_LN2, (double)e), lg);
    /* z = y*log(x); exp(z) = 2^k * exp(r), r = z - k*ln2 */
    const dd_t z = dd_mul_d(lg, y);
    if (z.hi > 709.0) return INFINITY;
    if (z.hi < -745.0) return 0.0;
    const double kf = nearbyint(z.hi / DD_LN2.hi);
    const dd_t r = dd_add(z, (dd_t){-dd_mul_d(DD_LN2, kf).hi, -dd_mul_d(DD_LN2, kf).lo});
    /* exp(r) = sum_n r^n / n!, Horner over the double-double table, |r| <= ln2/2 */
    dd_t ex = {eph_pow_invfact[EPH_POW_TERMS - 1].hi, eph_pow_invfact[EPH_POW_TERMS - 1].lo};
    for (int n = EPH_POW_TERMS - 2; n >= 0; --n)
        ex = dd_add(dd_mul(ex, r), (dd_t){eph_pow_invfact[n].hi, eph_pow_invfact[n].lo});
    return ldexp(ex.hi + ex.lo, (int)kf);
}
static int g_pow_mode = 0;   /* 0: correctly rounded (the pinned definition), 1: this host's libm pow */
void orc_set_pow_mode(int mode) { g_pow_mode = mode; }
double orc_cr_pow(double x, double y) { return cr_pow(x, y); }
static inline double controller_pow(double x, double y) { return g_pow_mode ? pow(x, y) : cr_pow(x, y); }

/* IController::step  mod.rs:225-243 ; num_traits::clamp / clamp_max */
static int controller_step(const adaptive_t *a, double err, double *h, int order) {
    const double k = (double)order;
    const double m = a->fac * controller_pow(err, -(1.0 / k));   /* err.pow(-k.inv()): f64::powf */
    const double c = m < a->fac_min ? a->fac_min : (m > a->fac_max ? a->fac_max : m);
    const double nh = *h * c;
    *h = nh > a->h_max ? a->h_max : nh;
    return err <= 1.0;
}
/* AdaptiveRungeKuttaIntegrator::advance  mod.rs:414-439 */
static int adaptive_advance(adaptive_t *a, double *time, double bound, double *state, ode_fn f, void *fctx,
                            tol_fn tol, void *tctx, uint64_t *evals) {
    const int D = a->rk.dim, S = a->rk.stages;
    /* prev.store: t, y, rk.undo_step(rk) (i and, if FSAL, the last stage) */
    a->prev_t = *time;
    for (int d = 0; d < D; ++d) a->prev_y[d] = state[d];
    a->prev_i = a->rk.i;
    if (a->rk.fsal) for (int d = 0; d < D; ++d) a->prev_klast[d] = a->rk.k[S - 1][d];
    const int lower = a->rk.order < a->rk.order_embedded ? a->rk.order : a->rk.order_embedded;
    for (;;) {
        if (a->n > a->n_max) return ORC_MAX_ITERATIONS;
        if (*time + a->next_h > bound) a->next_h = bound - *time;
        a->frk_h = a->next_h;
        /* FixedRungeKuttaIntegrator::advance  mod.rs:112-125 */
        if (*time >= bound) return ORC_BOUND_REACHED;
        if (*time + a->frk_h == *time) return ORC_STEP_SIZE_UNDERFLOW;
        int st = erk_advance(&a->rk, a->frk_h, time, state, f, fctx, evals);
        if (st) return st;
        a->n += 1;
        erk_error(&a->rk, a->frk_h, a->error);
        const double err = tol(tctx, state, a->error);
        if (controller_step(a, err, &a->next_h, lower)) break;
        /* prev.restore */
        *time = a->prev_t;
        for (int d = 0; d < D; ++d) state[d] = a->prev_y[d];
        a->rk.i = a->prev_i;
        if (a->rk.fsal) for (int d = 0; d < D; ++d) a->rk.k[S - 1][d] = a->prev_klast[d];
    }
    return ORC_OK;
}

/* ---- doc-test known answers: y' = -y  (integration/src/lib.rs:32-93) ---------------------------- */
static int decay_rhs(void *ctx, double t, const double *y, double *dy) { (void)ctx; (void)t; dy[0] = -y[0]; return 0; }
typedef struct { double atol, rtol; } scalar_tol_t;
static double scalar_tol(void *ctx, const double *state, const double *err) {
    const scalar_tol_t *t = ctx;
    return fabs(err[0]) / (t->atol + t->rtol * fabs(state[0]));
}
double orc_doc_test_decay(const char *method, int adaptive, double h, double h_max, double atol, double rtol,
                          double t_end, uint32_t *steps) {
    double time = 0.0, y[1] = {1.0};
    uint32_t nsteps = 0;
    if (!adaptive) {
        erk_t rk;
        if (erk_init(&rk, method, 1, y)) return NAN;
        for (;;) {                                           /* Integrator::advance_to_bound lib.rs:362-376 */
            if (time >= t_end) break;                        /* BoundReached */
            if (time + h == time) break;
            if (erk_advance(&rk, h, &time, y, decay_rhs, NULL, NULL)) return NAN;
            nsteps++;
            if (time >= t_end) break;
        }
    } else {
        adaptive_t a;
        /* AdaptiveMethodParams::new(tol, 10_000).h_init(..).h_max(..): fac_min 1/5, fac_max 5, fac 9/10 lib.rs:184-186,230-246 */
        if (adaptive_init(&a, method, 1, y, 0.0, h, h_max, 1.0 / 5.0, 5.0 / 1.0, 9.0 / 10.0, 10000)) return NAN;
        scalar_tol_t tol = {atol, rtol};
        for (;;) {
            int st = adaptive_advance(&a, &time, t_end, y, decay_rhs, NULL, scalar_tol, &tol, NULL);
            if (st) break;
            nsteps++;
            if (time >= t_end) break;
        }
    }
    if (steps) *steps = nsteps;
    return y[0];
}

/* ---- glam::DVec3 pieces the burn frame uses (crate glam 0.30.10, Cargo.lock:2890-2892; source not on disk) */
static v3 v3_sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static v3 v3_add(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static v3 v3_scale(v3 a, double s) { return (v3){a.x * s, a.y * s, a.z * s}; }
static double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 v3_cross(v3 a, v3 b) {
    return (v3){a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static double v3_length_recip(v3 a) { return 1.0 / sqrt(v3_dot(a, a)); }   /* self.length().recip() */
static int v3_try_normalize(v3 a, v3 *out) {                                /* rcp.is_finite() && rcp > 0.0 */
    const double rcp = v3_length_recip(a);
    if (isfinite(rcp) && rcp > 0.0) { *out = v3_scale(a, rcp); return 1; }
    return 0;
}

/* ---- Timeline  spacecraft.rs:58-222 ------------------------------------------------------------- */
typedef struct { double start, end; int is_burn; v3 acc; int ref; } segment_t;

struct orc_craft {
    const orc_solution *eph;
    double *mu;
    int32_t *body_order;           /* NULL = file order; else the order Bodies::acceleration visits the bodies in */
    int nseg, current_segment;
    segment_t *seg;
    /* problem */
    double time, bound, state[6];
    /* method parameters (kept: reset_integrator re-creates the integrator from them, spacecraft.rs:479-485) */
    char method[32];
    double h_init, h_max, fac_min, fac_max, fac, tol_pos, tol_vel;
    uint32_t n_max;
    adaptive_t integ;
    uint64_t evals;
    uint32_t steps;
    /* solution: CubicHermiteSpline */
    int64_t nk, capk;
    double *kt, *kp, *kv;
    /* SpacecraftSolout (the app's solout): SoiTransitions + Apsides  dynamics/spacecraft.rs:296-451,514-587 */
    double *soi_radius;            /* NULL = CubicHermiteSplineSolout only */
    int64_t ntr, captr, nap, capap;
    struct transition { double time; int32_t body; } *tr;
    struct apsis { double time, distance; int32_t body, kind; } *ap;   /* kind 0 = Periapsis, 1 = Apoapsis */
};

static int timeline_idx_at(const orc_craft *c, double time) {   /* partition_point(|seg| seg.end() <= time) :157-160 */
    int lo = 0, hi = c->nseg;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (c->seg[mid].end <= time) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* GravitationalBody::acceleration_at  dynamics/spacecraft.rs:70-74 with `particular` acceleration_at::<false>
 * (source absent, same restated formula as acceleration_paired: dir = body - at). */
static int craft_rhs(void *ctx, double t, const double *y, double *dy) {
    orc_craft *c = ctx;
    const v3 pos = {y[0], y[1], y[2]}, vel = {y[3], y[4], y[5]};
    /* Bodies::acceleration  dynamics/spacecraft.rs:222-228 */
    v3 acc = {0.0, 0.0, 0.0};
    /* (the app iterates an EntityHashMap, dynamics/spacecraft.rs:164-165,222-228: its order is unspecified upstream; file order
     * by default -- the library test's IndexMap -- or the permutation given to orc_craft_set_body_order) */
    for (int q = 0; q < c->eph->n; ++q) {
        const int b = c->body_order ? c->body_order[q] : q;
        double tau;
        const poly_t *p = spline_get_polynomial(&c->eph->s[b], t, &tau);
        if (!p) return ORC_EVAL_FAILED;
        const v3 bp = poly_eval(p, tau);
        const v3 d = v3_sub(bp, pos);
        const double n2 = v3_dot(d, d);
        acc = v3_add(acc, point_mass_term(d, n2, c->mu[b])); /* acceleration_at::<false>: same crate routine as the pairs */
    }
    /* manoeuvre_acceleration: Segment::acceleration  spacecraft.rs:102-116,272-281 */
    v3 man = {0.0, 0.0, 0.0};
    const segment_t *sg = &c->seg[c->current_segment];
    if (sg->is_burn) {
        if (sg->ref >= 0) {
            /* ReferenceFrame::Relative: TNB::try_new(sv - reference.state_vector(t))  dynamics/spacecraft.rs:281-293 */
            double tau;
            const spline_t *rs = &c->eph->s[sg->ref];
            const poly_t *p = spline_get_polynomial(rs, t, &tau);
            if (!p) return ORC_EVAL_FAILED;
            v3 rp, rd;
            poly_eval_and_deriv(p, tau, &rp, &rd);
            const v3 rv = {rd.x / rs->interval, rd.y / rs->interval, rd.z / rs->interval};
            const v3 rel_p = v3_sub(pos, rp), rel_v = v3_sub(vel, rv);
            v3 x, yv, z;
            if (!v3_try_normalize(rel_v, &x)) return ORC_EVAL_FAILED;             /* TNB::try_new :246-252 */
            if (!v3_try_normalize(v3_cross(rel_p, rel_v), &yv)) return ORC_EVAL_FAILED;
            const v3 xy = v3_cross(x, yv);
            z = v3_scale(xy, v3_length_recip(xy));                                /* normalize() */
            /* DMat3::from_cols(x, z, y).mul_vec3(a) = x*a.x + z*a.y + y*a.z */
            v3 r = v3_scale(x, sg->acc.x);
            r = v3_add(r, v3_scale(z, sg->acc.y));
            r = v3_add(r, v3_scale(yv, sg->acc.z));
            man = r;
        } else {                                                                  /* TNB::IDENTITY.mul_vec3 */
            v3 r = v3_scale((v3){1.0, 0.0, 0.0}, sg->acc.x);
            r = v3_add(r, v3_scale((v3){0.0, 1.0, 0.0}, sg->acc.y));
            r = v3_add(r, v3_scale((v3){0.0, 0.0, 1.0}, sg->acc.z));
            man = r;
        }
    }
    /* dy.velocity = context + manoeuvre ; dy.position = y.velocity   spacecraft.rs:303-306 */
    const v3 a = v3_add(acc, man);
    dy[0] = vel.x; dy[1] = vel.y; dy[2] = vel.z;
    dy[3] = a.x; dy[4] = a.y; dy[5] = a.z;
    return ORC_OK;
}
/* AbsTol::err_over_tol  dynamics/spacecraft.rs:615-624: max(|e_r / tol_r|.max_element(), |e_v / tol_v|.max_element()) */
static double abs_tol(void *ctx, const double *state, const double *e) {
    (void)state;
    const orc_craft *c = ctx;
    const double px = fabs(e[0] / c->tol_pos), py = fabs(e[1] / c->tol_pos), pz = fabs(e[2] / c->tol_pos);
    const double vx = fabs(e[3] / c->tol_vel), vy = fabs(e[4] / c->tol_vel), vz = fabs(e[5] / c->tol_vel);
    const double pm = fmax(px, fmax(py, pz)), vm = fmax(vx, fmax(vy, vz));   /* max_element: x.max(y.max(z)) */
    return fmax(pm, vm);
}
static void craft_push_knot(orc_craft *c) {                     /* CubicHermiteSpline::push  trajectory.rs:806-809 */
    if (c->nk == c->capk) {
        c->capk = c->capk ? c->capk * 2 : 256;
        c->kt = realloc(c->kt, sizeof(double) * (size_t)c->capk);
        c->kp = realloc(c->kp, sizeof(double) * 3 * (size_t)c->capk);
        c->kv = realloc(c->kv, sizeof(double) * 3 * (size_t)c->capk);
    }
    c->kt[c->nk] = c->time;
    memcpy(c->kp + 3 * c->nk, c->state, sizeof(double) * 3);
    memcpy(c->kv + 3 * c->nk, c->state + 3, sizeof(double) * 3);
    c->nk++;
}
static int craft_reset_integrator(orc_craft *c) {               /* spacecraft.rs:479-485 */
    return adaptive_init(&c->integ, c->method, 6, c->state, c->time, c->h_init, c->h_max, c->fac_min, c->fac_max,
                         c->fac, c->n_max);
}

/* ---- SpacecraftSolout events -------------------------------------------------------------------------------- */
typedef struct { double b0; v3 a0, a1, a2, a3; } hermite_t;
static hermite_t hermite_new(double t0, double t1, v3 p0, v3 p1, v3 d0, v3 d1) {   /* CubicHermite::new trajectory.rs:645-679 */
    hermite_t h;
    h.b0 = t0; h.a0 = p0; h.a1 = d0;
    const double dt = t1 - t0;
    if (dt == 0.0 && p0.x == p1.x && p0.y == p1.y && p0.z == p1.z && d0.x == d1.x && d0.y == d1.y && d0.z == d1.z) {
        h.a2 = (v3){0, 0, 0}; h.a3 = (v3){0, 0, 0};
        return h;
    }
    const double dt_recip = 1.0 / dt;
    const double dt_recip_2 = dt_recip * dt_recip;
    const double dt_recip_3 = dt_recip * dt_recip_2;
    const v3 dt_val = v3_sub(p1, p0);
    /* a2 = dt_val * dt_recip_2 * 3.0 - (d0 * 2.0 + d1) * dt_recip ; a3 = dt_val * dt_recip_3 * -2.0 + (d0 + d1) * dt_recip_2 */
    h.a2 = v3_sub(v3_scale(v3_scale(dt_val, dt_recip_2), 3.0), v3_scale(v3_add(v3_scale(d0, 2.0), d1), dt_recip));
    h.a3 = v3_add(v3_scale(v3_scale(dt_val, dt_recip_3), -2.0), v3_scale(v3_add(d0, d1), dt_recip_2));
    return h;
}
static v3 hermite_pos(const hermite_t *h, double t) {            /* eval :681-688 */
    const double dt = t - h->b0;
    return v3_add(v3_scale(v3_add(v3_scale(v3_add(v3_scale(h->a3, dt), h->a2), dt), h->a1), dt), h->a0);
}
static v3 hermite_vel(const hermite_t *h, double t) {            /* eval_derivative :690-697 */
    const double dt = t - h->b0;
    return v3_add(v3_scale(v3_add(v3_scale(v3_scale(h->a3, dt), 3.0), v3_scale(h->a2, 2.0)), dt), h->a1);
}
static int body_position(const orc_craft *c, int b, double t, v3 *out) {
    double tau;
    const poly_t *p = spline_get_polynomial(&c->eph->s[b], t, &tau);
    if (!p) return 0;
    *out = poly_eval(p, tau);
    return 1;
}
static int body_state(const orc_craft *c, int b, double t, v3 *pos, v3 *vel) {
    double tau;
    const spline_t *s = &c->eph->s[b];
    const poly_t *p = spline_get_polynomial(s, t, &tau);
    if (!p) return 0;
    v3 d;
    poly_eval_and_deriv(p, tau, pos, &d);
    *vel = (v3){d.x / s->interval, d.y / s->interval, d.z / s->interval};
    return 1;
}
/* soi_distance_squared_at  dynamics/spacecraft.rs:77-83 ; radial_velocity_at :85-89 */
typedef struct { const orc_craft *c; const hermite_t *h; int body, radial; } event_fn_t;
static int event_f(const event_fn_t *e, double t, double *out) {
    if (!e->radial) {
        v3 bp;
        if (!body_position(e->c, e->body, t, &bp)) return 0;
        const v3 d = v3_sub(hermite_pos(e->h, t), bp);          /* position.distance_squared(body) */
        const double r = e->c->soi_radius[e->body];
        *out = v3_dot(d, d) - r * r;
        return 1;
    }
    v3 bp, bv;
    if (!body_state(e->c, e->body, t, &bp, &bv)) return 0;
    const v3 rp = v3_sub(hermite_pos(e->h, t), bp), rv = v3_sub(hermite_vel(e->h, t), bv);   /* sv - body sv */
    *out = v3_dot(rp, rv);
    return 1;
}
static double f64_signum(double x) { return x != x ? x : copysign(1.0, x); }
/* find_zero_crossing + find_root_bisection  dynamics/spacecraft.rs:112-162. Returns 1 and (time, ascending). */
static int find_zero_crossing(const event_fn_t *e, double t0, double t1, double *time, int *ascending) {
    double f0, f1;
    if (!event_f(e, t0, &f0) || !event_f(e, t1, &f1)) return 0;
    if (f64_signum(f0) == f64_signum(f1)) return 0;
    double x0 = t0, x1 = t1, g0 = f0;
    for (int it = 0; it < 100; ++it) {
        const double mid = x0 + (x1 - x0) / 2.0;
        double f_mid = 0.0;
        event_f(e, mid, &f_mid);                                 /* f(t).unwrap() */
        if (f64_signum(g0) != f64_signum(f_mid)) x1 = mid;
        else { x0 = mid; g0 = f_mid; }
        if (fabs(x1 - x0) < 1e-3) {
            *time = x0;
            *ascending = signbit(f0) ? 1 : 0;                    /* f0.is_sign_negative() => Ascending */
            return 1;
        }
    }
    return 0;
}
/* find_soi over Bodies in body order (the reference iterates an EntityHashMap, whose order is unspecified)
 * dynamics/spacecraft.rs:172-185,208-221: inside iff d2 < r*r, the closest wins (first on ties: Iterator::min_by) */
static int soi_at_except(const orc_craft *c, double t, v3 position, int except) {
    int best = -1;
    double best_d2 = 0.0;
    for (int b = 0; b < c->eph->n; ++b) {
        if (b == except) continue;
        v3 bp;
        if (!body_position(c, b, t, &bp)) continue;              /* filter_map */
        const v3 d = v3_sub(position, bp);
        const double d2 = v3_dot(d, d), r = c->soi_radius[b];
        if (!(d2 < r * r)) continue;
        if (best < 0 || d2 < best_d2 /* total_cmp on finite d2 */) { best = b; best_d2 = d2; }
    }
    return best;
}
static int64_t tr_search(const orc_craft *c, double time, int *found) {   /* binary_search_by(|(t, ..)| t.cmp(&time)) */
    int64_t lo = 0, hi = c->ntr;
    *found = 0;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (c->tr[mid].time == time) { *found = 1; return mid; }
        if (c->tr[mid].time < time) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static void tr_insert(orc_craft *c, double time, int body) {     /* SoiTransitions::insert :332-339 */
    int found;
    const int64_t i = tr_search(c, time, &found);
    if (found) { c->tr[i].time = time; c->tr[i].body = body; return; }
    if (i > 0 && c->tr[i - 1].body == body) return;
    if (c->ntr == c->captr) {
        c->captr = c->captr ? 2 * c->captr : 16;
        c->tr = realloc(c->tr, sizeof(*c->tr) * (size_t)c->captr);
    }
    memmove(c->tr + i + 1, c->tr + i, sizeof(*c->tr) * (size_t)(c->ntr - i));
    c->tr[i].time = time; c->tr[i].body = body;
    c->ntr++;
}
static void ap_insert(orc_craft *c, double time, double distance, int body, int kind) {   /* Apsides::insert :420-426 */
    int64_t lo = 0, hi = c->nap;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (c->ap[mid].time == time) { c->ap[mid] = (struct apsis){time, distance, body, kind}; return; }
        if (c->ap[mid].time < time) lo = mid + 1; else hi = mid;
    }
    if (c->nap == c->capap) {
        c->capap = c->capap ? 2 * c->capap : 16;
        c->ap = realloc(c->ap, sizeof(*c->ap) * (size_t)c->capap);
    }
    memmove(c->ap + lo + 1, c->ap + lo, sizeof(*c->ap) * (size_t)(c->nap - lo));
    c->ap[lo] = (struct apsis){time, distance, body, kind};
    c->nap++;
}
static v3 knot_v3(const double *a, int64_t i) { return (v3){a[3 * i], a[3 * i + 1], a[3 * i + 2]}; }
/* SpacecraftSolout::solout after the push  dynamics/spacecraft.rs:539-586 */
static void craft_events(orc_craft *c) {
    const int64_t k = c->nk - 2;
    const double t0 = c->kt[k], t1 = c->kt[k + 1];
    const hermite_t h = hermite_new(t0, t1, knot_v3(c->kp, k), knot_v3(c->kp, k + 1), knot_v3(c->kv, k),
                                    knot_v3(c->kv, k + 1));
    for (int b = 0; b < c->eph->n; ++b) {
        const event_fn_t e = {c, &h, b, 0};
        double time; int asc;
        if (!find_zero_crossing(&e, t0, t1, &time, &asc)) continue;
        if (!asc) tr_insert(c, time, b);                         /* Descending: entered b's sphere */
        else {
            const int entered = soi_at_except(c, time, hermite_pos(&h, time), b);
            if (entered >= 0) tr_insert(c, time, entered);
        }
    }
    /* transitions.starting_at(t0) :326-329 */
    int found;
    int64_t i0 = tr_search(c, t0, &found);
    if (!found) i0 = i0 == 0 ? 0 : i0 - 1;
    for (int64_t i = i0; i < c->ntr; ++i) {
        const double t = c->tr[i].time;
        const int soi = c->tr[i].body;
        const double ta = t0 > t ? t0 : t;                       /* t.max(t0) */
        const double tb = i + 1 < c->ntr ? c->tr[i + 1].time : t1;
        const event_fn_t e = {c, &h, soi, 1};
        double time; int asc;
        if (!find_zero_crossing(&e, ta, tb, &time, &asc)) continue;
        v3 bp;
        if (!body_position(c, soi, time, &bp)) continue;         /* distance_at  dynamics/mod.rs:141-146 */
        const v3 d = v3_sub(bp, hermite_pos(&h, time));
        ap_insert(c, time, sqrt(v3_dot(d, d)), soi, asc ? 0 : 1);
    }
}
/* Enable the app's SpacecraftSolout: soi_radius[b] per body (INFINITY for the root). new_solution :525-537 */
void orc_craft_enable_events(orc_craft *c, const double *soi_radius) {
    c->soi_radius = malloc(sizeof(double) * (size_t)(c->eph->n > 0 ? c->eph->n : 1));
    memcpy(c->soi_radius, soi_radius, sizeof(double) * (size_t)c->eph->n);
    c->ntr = c->nap = 0;
    const int cur = soi_at_except(c, c->kt[0], knot_v3(c->kp, 0), -1);
    if (cur >= 0) tr_insert(c, c->kt[0], cur);
}
int64_t orc_craft_transitions(const orc_craft *c, double *time, int32_t *body) {
    for (int64_t i = 0; time && i < c->ntr; ++i) { time[i] = c->tr[i].time; body[i] = c->tr[i].body; }
    return c->ntr;
}
int64_t orc_craft_apsides(const orc_craft *c, double *time, double *distance, int32_t *body, int32_t *kind) {
    for (int64_t i = 0; time && i < c->nap; ++i) {
        time[i] = c->ap[i].time; distance[i] = c->ap[i].distance; body[i] = c->ap[i].body; kind[i] = c->ap[i].kind;
    }
    return c->nap;
}

orc_craft *orc_craft_new(const orc_solution *eph, const double *mu, double t0, const double *pos, const double *vel,
                         const char *method, double h_init, double h_max, double tol_pos, double tol_vel,
                         double fac_min, double fac_max, double fac, uint32_t n_max, int nburns,
                         const double *burn_start, const double *burn_end, const double *burn_acc,
                         const int32_t *burn_ref) {
    const EPH_ERK_TABLE *tab = find_erk(method);
    if (!((tab && tab->E) || find_erkng(method) || find_erkn(method)) || strlen(method) >= 32) return NULL;
    /* ERKN needs P::ODE: SecondOrderODE (nystrom/explicit.rs:60); the spacecraft model is one only while no burn's
     * frame depends on the velocity (ReferenceFrame::Relative -> TNB of the relative STATE, dynamics/spacecraft.rs:281-293) */
    if (find_erkn(method))
        for (int i = 0; i < nburns; ++i) if (burn_ref[i] >= 0) return NULL;
    orc_craft *c = calloc(1, sizeof(*c));
    c->eph = eph;
    c->mu = malloc(sizeof(double) * (size_t)(eph->n > 0 ? eph->n : 1));
    memcpy(c->mu, mu, sizeof(double) * (size_t)eph->n);
    /* Timeline::new  spacecraft.rs:129-152: stable sort by start, coast segments in the gaps */
    int *order = malloc(sizeof(int) * (size_t)(nburns > 0 ? nburns : 1));
    for (int i = 0; i < nburns; ++i) order[i] = i;
    for (int i = 1; i < nburns; ++i) {                          /* insertion sort = stable */
        int k = order[i], j = i - 1;
        while (j >= 0 && burn_start[order[j]] > burn_start[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    c->seg = calloc((size_t)(2 * nburns + 1), sizeof(segment_t));
    const double EMIN = -1.7976931348623157e308, EMAX = 1.7976931348623157e308;   /* Epoch::MIN / MAX */
    double cursor = EMIN;
    for (int q = 0; q < nburns; ++q) {
        const int i = order[q];
        if (burn_start[i] > cursor) c->seg[c->nseg++] = (segment_t){cursor, burn_start[i], 0, {0, 0, 0}, -1};
        cursor = burn_end[i];
        c->seg[c->nseg++] = (segment_t){burn_start[i], burn_end[i], 1,
                                        {burn_acc[3 * i], burn_acc[3 * i + 1], burn_acc[3 * i + 2]}, burn_ref[i]};
    }
    if (cursor < EMAX) c->seg[c->nseg++] = (segment_t){cursor, EMAX, 0, {0, 0, 0}, -1};
    free(order);
    /* SpacecraftPropagator::new  spacecraft.rs:446-476 */
    c->time = t0;
    memcpy(c->state, pos, sizeof(double) * 3);
    memcpy(c->state + 3, vel, sizeof(double) * 3);
    c->current_segment = timeline_idx_at(c, t0);
    c->bound = c->seg[c->current_segment].end;
    strcpy(c->method, method);
    c->h_init = h_init; c->h_max = h_max; c->fac_min = fac_min; c->fac_max = fac_max; c->fac = fac;
    c->tol_pos = tol_pos; c->tol_vel = tol_vel; c->n_max = n_max;
    craft_reset_integrator(c);
    craft_push_knot(c);                                          /* CubicHermiteSplineSolout::new_solution :654-661 */
    return c;
}
int orc_craft_set_body_order(orc_craft *c, const int32_t *order) {
    free(c->body_order);
    c->body_order = NULL;
    if (!order) return ORC_OK;
    const int n = c->eph->n;
    int32_t *o = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    char *seen = calloc((size_t)(n > 0 ? n : 1), 1);
    int ok = o && seen;
    for (int q = 0; ok && q < n; ++q) {
        ok = order[q] >= 0 && order[q] < n && !seen[order[q]];
        if (ok) { seen[order[q]] = 1; o[q] = order[q]; }
    }
    free(seen);
    if (!ok) { free(o); return ORC_BAD_ARGUMENT; }
    c->body_order = o;
    return ORC_OK;
}
void orc_craft_free(orc_craft *c) {
    if (c) free(c->body_order);
    if (!c) return;
    free(c->mu); free(c->seg); free(c->kt); free(c->kp); free(c->kv); free(c->soi_radius); free(c->tr); free(c->ap);
    free(c);
}
int orc_craft_step(orc_craft *c) {                               /* spacecraft.rs:598-615 */
    /* advance_timeline(self.time())  :250-256 */
    if (c->time >= c->seg[c->current_segment].end) {
        c->current_segment += 1;
        c->bound = c->seg[c->current_segment].end;              /* set_bound(new_end) */
        craft_reset_integrator(c);
    }
    int st = adaptive_advance(&c->integ, &c->time, c->bound, c->state, craft_rhs, c, abs_tol, c, &c->evals);
    if (st) return st;
    c->steps++;
    craft_push_knot(c);                                          /* solout :663-676 */
    if (c->soi_radius) craft_events(c);
    return ORC_OK;
}
int orc_craft_step_to(orc_craft *c, double t) {
    for (;;) {
        if (c->kt[c->nk - 1] >= t) return ORC_OK;                /* has_reached: solution.end() >= time :691-693 */
        int st = orc_craft_step(c);
        if (st) return st;
    }
}
int64_t orc_craft_knots(const orc_craft *c) { return c->nk; }
void orc_craft_get_knots(const orc_craft *c, double *t, double *pos, double *vel) {
    memcpy(t, c->kt, sizeof(double) * (size_t)c->nk);
    memcpy(pos, c->kp, sizeof(double) * 3 * (size_t)c->nk);
    memcpy(vel, c->kv, sizeof(double) * 3 * (size_t)c->nk);
}
void orc_craft_state(const orc_craft *c, double *t, double *pos, double *vel, double *next_h, uint32_t *n,
                     uint32_t *steps) {
    if (t) *t = c->time;
    if (pos) memcpy(pos, c->state, sizeof(double) * 3);
    if (vel) memcpy(vel, c->state + 3, sizeof(double) * 3);
    if (next_h) *next_h = c->integ.next_h;
    if (n) *n = c->integ.n;
    if (steps) *steps = c->steps;
}
uint64_t orc_craft_evals(const orc_craft *c) { return c->evals; }

/* CubicHermiteSpline::state_vector  trajectory.rs:766-797 with CubicHermite::new/eval/eval_derivative :645-696 */
int orc_hermite_eval(int64_t n, const double *t, const double *pos, const double *vel, double at, double *p, double *v) {
    /* binary_search_by(|(t, _)| t.cmp(&at)) */
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (t[mid] == at) {
            memcpy(p, pos + 3 * mid, sizeof(double) * 3);
            if (v) memcpy(v, vel + 3 * mid, sizeof(double) * 3);
            return 1;
        }
        if (t[mid] < at) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return 0;                                       /* i.checked_sub(1)? */
    const int64_t i = lo - 1;
    if (i + 1 >= n) return 0;                                    /* self.0.get(i + 1)? */
    const double b0 = t[i], b1 = t[i + 1];
    const double dt = b1 - b0;
    for (int c = 0; c < 3; ++c) {
        const double v0 = pos[3 * i + c], v1 = pos[3 * (i + 1) + c], d0 = vel[3 * i + c], d1 = vel[3 * (i + 1) + c];
        double a2, a3;
        /* the degenerate test is on whole vectors; dt == 0 cannot happen between distinct knots found by the search */
        const double dt_recip = 1.0 / dt;
        const double dt_recip_2 = dt_recip * dt_recip;
        const double dt_recip_3 = dt_recip * dt_recip_2;
        const double dt_val = v1 - v0;
        a2 = dt_val * dt_recip_2 * 3.0 - (d0 * 2.0 + d1) * dt_recip;
        a3 = dt_val * dt_recip_3 * -2.0 + (d0 + d1) * dt_recip_2;
        const double x = at - b0;
        p[c] = (((a3 * x + a2) * x) + d0) * x + v0;
        if (v) v[c] = ((a3 * x * 3.0 + a2 * 2.0) * x) + d0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* The reference's own convergence test on Double<DVec3> (ephemeris/tests/solar_system_convergence.rs) */
/* ------------------------------------------------------------------------------------------------ */
#include "convergence_double.inc"
