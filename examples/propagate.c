/* examples/propagate.c -- the drop-in boundary used from plain C, the way the reference-side binding
 * would use it: build the massive bodies' ephemeris with the propagator seam, then read a body's state from it.
 *
 *   gcc -std=c99 -Iinclude examples/propagate.c -Lephemeris_explorer_amd -lephemeris_amd -Wl,-rpath,$PWD/ephemeris_explorer_amd -o propagate
 *   ./propagate            (needs an MI355X; without a device every compute call returns EPH_ERR_NO_DEVICE)
 *
 * Three bodies (Sun, Earth, Moon at 1950-01-01, the values of systems/sun_earth_moon_2433282.5/state.json), dt = 6 h,
 * QuinlanTremaine12, 30 days forward; then the Earth's position at day 10 from the Vec<UniformSpline>.
 */
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
    const uint32_t count[3] = {12, 3, 1}, degree[3] = {6, 7, 6};   /* ephemeris.json: samples every `count` steps, fit degree */
    const double t0 = -252460800.0 /* 1950-01-01 00:00:00 TAI, seconds since 1958-01-01 */, dt = 21600.0;
    int32_t ndev = 0;
    CHECK(eph_device_count(&ndev));
    printf("abi %d, pair variant %d, %d device(s)\n", (int)eph_abi_version(), (int)eph_pair_variant(), (int)ndev);

    eph_prop *p = NULL;
    CHECK(eph_prop_create(3, pos, vel, mu, t0, dt, EPH_FORWARD, "QuinlanTremaine12", count, degree, &p));
    CHECK(eph_prop_step_to(p, t0 + 30.0 * 86400.0));               /* IncrementalPropagator::step_to */
    double reached = 0.0;
    CHECK(eph_prop_time(p, &reached));
    eph_solution *sol = NULL;
    CHECK(eph_prop_take_solution(p, &sol));                        /* Propagator::take_solution */
    double start, interval;
    int64_t npoly;
    CHECK(eph_solution_info(sol, 1, &start, &interval, &npoly));
    const double at = t0 + 10.0 * 86400.0;
    double r[3], v[3];
    uint8_t inside = 0;
    CHECK(eph_solution_eval(sol, 1, 1, &at, r, v, &inside));        /* EvaluateTrajectory::state_vector */
    printf("reached %.1f s; Earth spline: %lld polynomials of %.0f s; r(day 10) = (%.3f, %.3f, %.3f) km inside=%d\n",
           reached - t0, (long long)npoly, interval, r[0], r[1], r[2], (int)inside);
    eph_solution_destroy(sol);
    eph_prop_destroy(p);
    return 0;
}
