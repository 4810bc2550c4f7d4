/* examples/craft.c -- the spacecraft seam used from plain C, the way INTEGRATION.md section 4b's GpuSpacecraftPropagator
 * uses it: ephemeris of the massive bodies -> eph_ephemeris -> a BATCH OF ONE spacecraft (SpacecraftPropagator::new,
 * ephemeris_explorer/src/dynamics/spacecraft.rs:661-677) with the app's solout (SpacecraftSolout: knots + SOI
 * transitions + apsides) -> the prediction task's loop `step(); if has_reached(end) break` (prediction.rs:422-443) in
 * chunks of steps -> take_solution = knots + events (SpacecraftSolution, dynamics/spacecraft.rs:453-458).
 *
 *   gcc -std=c99 -Iinclude examples/craft.c -Lephemeris_explorer_amd -lephemeris_amd -Wl,-rpath,$PWD/ephemeris_explorer_amd -o craft
 *   ./craft                (needs an MI355X; without a device every compute call returns EPH_ERR_NO_DEVICE)
 *
 * Sun, Earth, Moon at 1950-01-01 (systems/sun_earth_moon_2433282.5/state.json) and the "Earth Station" ship of that system
 * (ships/Earth Station.json: a 7000 km orbit around the Earth) with one prograde burn added: 0.5 m/s^2 for 60 s in the
 * TNB frame of the Earth, two hours in. Verner87, tolerance 1e-3 km, three days.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ephemeris_amd.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int32_t st_ = (call);                                                                         \
        if (st_ != EPH_OK) {                                                                          \
            fprintf(stderr, "%s -> %d (%s) %s\n", #call, (int)st_, eph_status_string(st_), eph_last_error()); \
            return st_ == EPH_ERR_NO_DEVICE ? 77 : 1;                                                 \
        }                                                                                             \
    } while (0)

int main(void) {
    const double mu[3] = {132712440041.27942, 398600.43550702266, 4902.80011845755};
    const double pos[9] = {130800.7436285839, 344339.3116943656, 136496.914202216,
                           -27204249.66910069, 132940582.438431, 57641619.74238631,
                           -27017766.52877057, 133253431.1006455, 57806029.23241135};
    const double vel[9] = {-0.007799748521575531, -0.005561934613704532, -0.00225317087714714,
                           -29.75359910616436, -5.189518219844614, -2.251561710555783,
                           -30.64009897505477, -4.820684674596127, -2.032529075882219};
    const uint32_t count[3] = {12, 3, 1}, degree[3] = {6, 7, 6};
    const double t0 = -252460800.0 /* 1950-01-01 00:00:00 TAI */, dt = 21600.0, day = 86400.0;

    /* 1. the massive bodies' Vec<UniformSpline> (what Bodies holds per GravitationalBody), a little past the ship's end */
    eph_prop *p = NULL;
    CHECK(eph_prop_create(3, pos, vel, mu, t0, dt, EPH_FORWARD, "QuinlanTremaine12", count, degree, &p));
    eph_solution *sol = NULL;
    CHECK(eph_prop_propagate(p, t0 + 40.0 * day, &sol));
    eph_ephemeris *eph = NULL;
    CHECK(eph_ephemeris_create(sol, mu, &eph));

    /* 2. SpacecraftPropagator::new: a batch of one */
    const double ship_pos[3] = {-27204249.668775786, 132947582.43848978, 57641619.74241204};
    const double ship_vel[3] = {-22.207539106181895, -5.189518219791726, -2.2515617105336263};
    const eph_adaptive_params params = {60.0, 1.7976931348623157e308, 1e-3, 1e-3, 1.0 / 5.0, 5.0, 9.0 / 10.0, 1000000u};
    const int64_t burn_offset[2] = {0, 1};
    const double burn_start[1] = {t0 + 7200.0}, burn_end[1] = {t0 + 7260.0}, burn_acc[3] = {5e-4, 0.0, 0.0};
    const int32_t burn_ref[1] = {1};                                  /* TNB frame of the Earth */
    const int32_t max_knots = 512;
    eph_craft_batch *ship = NULL;
    CHECK(eph_craft_batch_create(eph, 1, &t0, ship_pos, ship_vel, "Verner87", &params, burn_offset, burn_start, burn_end,
                                 burn_acc, burn_ref, max_knots, &ship));
    /* SphereOfInfluence::approximate over Sun > Earth > Moon (load/mod.rs:283-307): a (m / M)^(2/5) */
    const double soi[3] = {INFINITY, 909153.0740387321, 68800.06030265747};
    CHECK(eph_craft_batch_enable_events(ship, soi, 64, 256));

    /* 3. the task loop: step until the solution has reached `end` (time() = solution.trajectory.end() = the newest
     * knot = the problem time after an accepted step), 64 steps per call; the slab is drained when it fills */
    const double end = t0 + 3.0 * day;
    double t = t0, r[3], v[3], h;
    long total_knots = 1, calls = 0;
    for (;;) {
        int32_t status = 0, nknots = 0;
        CHECK(eph_craft_batch_step_n(ship, 64));
        CHECK(eph_craft_batch_status(ship, &status, &nknots, NULL, NULL));
        CHECK(eph_craft_batch_state(ship, &t, r, v, &h));
        ++calls;
        if (status == EPH_KNOTS_FULL) {                                /* keep what was read, continue from the newest knot */
            total_knots += nknots - 1;
            CHECK(eph_craft_batch_reset_knots(ship));
        } else if (status != EPH_OK) {
            fprintf(stderr, "StepError %d\n", (int)status);
            return 1;
        }
        if (t >= end) break;                                           /* DirectionalPropagator::has_reached */
    }
    int32_t status = 0, nknots = 0, ntr = 0, nap = 0, est = 0;
    uint32_t steps = 0;
    CHECK(eph_craft_batch_status(ship, &status, &nknots, NULL, &steps));
    total_knots += nknots - 1;

    /* 4. take_solution: the knots still in the slab + the event lists */
    double *kt = malloc(sizeof(double) * (size_t)nknots), *kp = malloc(sizeof(double) * 3 * (size_t)nknots),
           *kv = malloc(sizeof(double) * 3 * (size_t)nknots);
    CHECK(eph_craft_batch_knots(ship, 0, kt, kp, kv));
    CHECK(eph_craft_batch_event_counts(ship, &ntr, &nap, &est));
    double *tr_t = malloc(sizeof(double) * (size_t)(ntr + 1)), *ap_t = malloc(sizeof(double) * (size_t)(nap + 1)),
           *ap_d = malloc(sizeof(double) * (size_t)(nap + 1));
    int32_t *tr_b = malloc(sizeof(int32_t) * (size_t)(ntr + 1)), *ap_b = malloc(sizeof(int32_t) * (size_t)(nap + 1)),
            *ap_k = malloc(sizeof(int32_t) * (size_t)(nap + 1));
    CHECK(eph_craft_batch_events(ship, 0, tr_t, tr_b, ap_t, ap_d, ap_b, ap_k));
    printf("steps %u in %ld calls, knots %ld, t - t0 = %.6f s, r = (%.6f, %.6f, %.6f) km\n", (unsigned)steps, calls, total_knots,
           t - t0, r[0], r[1], r[2]);
    printf("transitions %d (first: body %d at %.3f s), apsides %d (event status %d)\n", (int)ntr, ntr ? (int)tr_b[0] : -1,
           ntr ? tr_t[0] - t0 : 0.0, (int)nap, (int)est);
    if (nap)
        printf("first apsis: %s of body %d at %.3f s, %.6f km\n", ap_k[0] ? "apoapsis" : "periapsis", (int)ap_b[0],
               ap_t[0] - t0, ap_d[0]);
    free(kt); free(kp); free(kv); free(tr_t); free(tr_b); free(ap_t); free(ap_d); free(ap_b); free(ap_k);
    eph_craft_batch_destroy(ship);
    eph_ephemeris_destroy(eph);
    eph_solution_destroy(sol);
    eph_prop_destroy(p);
    return 0;
}
