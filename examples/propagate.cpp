// examples/propagate.cpp -- the reference's operator surface from C++ (include/ephemeris_amd.hpp over the C ABI): the calls read like the
// reference's own tests (ephemeris/tests/spacecraft_propagation.rs): build the massive bodies' ephemeris with a bounded propagator,
// hand it to a spacecraft propagator, step that to a time, read both trajectories.
//
//   g++ -std=c++17 -Iinclude examples/propagate.cpp -Lephemeris_explorer_amd -lephemeris_amd -Wl,-rpath,$PWD/ephemeris_explorer_amd -o propagate_cpp
//   ./propagate_cpp        (needs an MI355X; without a device the first compute call throws Error{EPH_ERR_NO_DEVICE}: exit 77)
//
// Values are printed as hex floats: the GPU test compares them bit for bit with the CPU restatement.
#include <cstdio>
#include <limits>

#include "ephemeris_amd.hpp"

namespace ea = ephemeris_amd;

int main() try {
    const std::vector<double> mu = {132712440041.27942, 398600.43550702266, 4902.80011845755};
    const std::vector<ea::DVec3> y = {{130800.7436285839, 344339.3116943656, 136496.914202216},
                                      {-27204249.66910069, 132940582.438431, 57641619.74238631},
                                      {-27017766.52877057, 133253431.1006455, 57806029.23241135}};
    const std::vector<ea::DVec3> dy = {{-0.007799748521575531, -0.005561934613704532, -0.00225317087714714},
                                       {-29.75359910616436, -5.189518219844614, -2.251561710555783},
                                       {-30.64009897505477, -4.820684674596127, -2.032529075882219}};
    const double t0 = -252460800.0, dt = 21600.0, day = 86400.0;
    std::printf("abi %d, %d device(s)\n", (int)eph_abi_version(), (int)ea::device_count());

    // SecondOrderODE::eval of NewtonianGravity at the initial state
    std::vector<ea::DVec3> ddy(3, ea::DVec3{0.0, 0.0, 0.0});
    ea::newtonian_gravity_eval(y, mu, ddy);
    std::printf("ddy[2] = %a %a %a\n", ddy[2][0], ddy[2][1], ddy[2][2]);

    // NBodyPropagator::new(.., Forward, QuinlanTremaine12 h = dt, SplineInterpolators{count, degree}).propagate(t0 + 40 d)
    ea::NBodyPropagator massive(y, dy, mu, t0, dt, ea::Direction::Forward, {12, 3, 1}, {6, 7, 6});
    ea::StepError err = ea::StepError::None;
    ea::Solution splines = massive.propagate(t0 + 40.0 * day, &err);
    if (err != ea::StepError::None) { std::fprintf(stderr, "propagate: %s\n", ea::to_string(err)); return 1; }
    ea::StateVector earth{};
    const bool inside = splines.state_vector(1, t0 + 10.0 * day, earth);
    std::printf("reached %a; Earth spline: %lld polynomials of %.0f s; inside=%d\n", massive.time(), (long long)splines.len(1), splines.interval(1), (int)inside);
    std::printf("earth(day 10) = %a %a %a | %a %a %a\n", earth.position[0], earth.position[1], earth.position[2], earth.velocity[0],
                earth.velocity[1], earth.velocity[2]);

    // SpacecraftPropagator over those bodies: Verner87, the app's adaptive parameters, one burn in the Earth's TNB frame
    ea::Ephemeris bodies(splines, mu);
    const ea::StateVector craft0{{-27204249.668775786, 132947582.43848978, 57641619.74241204}, {-22.207539106181895, -5.189518219791726, -2.2515617105336263}};
    ea::SpacecraftBatch craft(bodies, t0, {craft0}, "Verner87", ea::AdaptiveParams(1e-3), {{ea::Burn{t0 + 7200.0, t0 + 7260.0, {5e-4, 0.0, 0.0}, 1}}});
    craft.step_to(t0 + 3.0 * day);
    std::vector<int32_t> nknots;
    const std::vector<int32_t> status = craft.status(&nknots);
    std::vector<double> kt;
    std::vector<ea::DVec3> kp, kv;
    craft.knots(0, nknots[0], kt, kp, kv);
    std::printf("craft: status %d (%s), knots %d, last knot t = %a r = %a %a %a\n", (int)status[0], eph_status_string(status[0]), (int)nknots[0], kt.back(),
                kp.back()[0], kp.back()[1], kp.back()[2]);

    // the app's flow (prediction.rs:422-443): SpacecraftSolout events, a snapshot (Clone) resumed later, a long propagation drained in
    // pieces and stitched with SpacecraftPropagator::join
    ea::SpacecraftBatch ship(bodies, t0, {craft0}, "Verner87", ea::AdaptiveParams(1e-3), {{ea::Burn{t0 + 7200.0, t0 + 7260.0, {5e-4, 0.0, 0.0}, 1}}});
    ship.enable_events({std::numeric_limits<double>::infinity(), 909153.0740387321, 68800.06030265747});
    ea::SpacecraftBatch snapshot = ship.clone();
    ship.step_to(t0 + 1.5 * day);
    snapshot.step_to(t0 + 1.5 * day);                       // the resumed snapshot takes the same steps
    ea::CubicHermiteSpline whole = snapshot.trajectory(0);
    snapshot.reset_knots();                                 // newest knot becomes knot 0 of an empty slab
    snapshot.step_to(t0 + 3.0 * day);
    whole.join(snapshot.trajectory(0));
    ea::StateVector at2{};
    const bool in2 = whole.state_vector(t0 + 2.0 * day, at2);
    ea::SoiTransitions tr;
    ea::Apsides ap;
    const int32_t ev = snapshot.events(0, tr, ap);
    std::printf("joined: knots %zu (first leg %zu), inside=%d, r(day 2) = %a %a %a\n", whole.len(), ship.trajectory(0).len(), (int)in2, at2.position[0],
                at2.position[1], at2.position[2]);
    std::printf("events: status %d, transitions %zu, apsides %zu, first apsis at %a\n", (int)ev, tr.time.size(), ap.time.size(), ap.time.empty() ? 0.0 : ap.time.front());
    const double restart = ea::divergence_time_before({ea::Burn{t0 + 7200.0, t0 + 7260.0, {5e-4, 0.0, 0.0}, 1}},
                                                      {ea::Burn{t0 + 7200.0, t0 + 7260.0, {5e-4, 0.0, 0.0}, 1}, ea::Burn{t0 + 2.0 * day, t0 + 2.0 * day + 30.0, {0.0, 1e-4, 0.0}, -1}},
                                                      t0 + 3.0 * day);
    std::printf("flight plan edit restarts at t0 + %.1f s\n", restart - t0);

    // auto-extend (ephemeris_explorer/src/auto_extend.rs:182-202): the bodies' propagator keeps going and every snapshot is merged into
    // the LIVE table (dynamics/celestial.rs:198-204); a stored ship propagator that ran off the table's end (EvalFailed,
    // spacecraft.rs:264-281) resumes against the same context once it has grown (prediction.rs:378, flight_plan.rs:363-395)
    ea::SpacecraftBatch runner(bodies, t0, {craft0}, "DormandPrince54", ea::AdaptiveParams(1e-3), {}, 65536);
    const double target = t0 + 50.0 * day;
    runner.step_to(target);
    std::vector<int32_t> nk0;
    const int32_t off_the_end = runner.status(&nk0)[0];
    const bool valid_before = bodies.is_valid_at(target);
    ea::Solution extension = massive.propagate(t0 + 60.0 * day, &err);
    bodies.merge(extension);
    runner.retry_failed();
    runner.step_to(target);
    std::vector<int32_t> nk1;
    const int32_t resumed = runner.status(&nk1)[0];
    const ea::CubicHermiteSpline path = runner.trajectory(0);
    std::printf("auto-extend: %s after %d knots (table valid at target: %d -> %d, revision %llu); resumed: %s, knots %d, last knot t = %a r = %a %a %a\n",
                eph_status_string(off_the_end), (int)nk0[0], (int)valid_before, (int)bodies.is_valid_at(target), (unsigned long long)bodies.revision(),
                eph_status_string(resumed), (int)nk1[0], path.t.back(), path.position.back()[0], path.position.back()[1], path.position.back()[2]);
    return 0;
} catch (const ea::Error &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return e.status == EPH_ERR_NO_DEVICE ? 77 : 1;
}
