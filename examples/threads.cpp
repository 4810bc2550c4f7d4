// examples/threads.cpp -- the header's threading contract, executed: "a handle is not thread-safe; distinct handles may be used from
// distinct threads; eph_ephemeris is the one handle that may be shared" (include/ephemeris_amd.h). The reference relies on exactly
// that: one async task per propagator on Bevy's compute pool -- forward and backward N-body propagators and one task per ship, each
// `loop { step(); if ready { take_solution(); clone(); send } }` (ephemeris_explorer/src/prediction.rs:385-391,422-443,
// load/mod.rs:673-687) -- the ships evaluating the bodies' LIVE trajectories (Arc<RwLock<..>>, dynamics/mod.rs:84-85) while merged
// snapshots grow them, and propagators that are MOVED between the pool's threads.
//
//   g++ -std=c++17 -pthread -Iinclude examples/threads.cpp -Lephemeris_explorer_amd -lephemeris_amd -Wl,-rpath,$PWD/ephemeris_explorer_amd -o threads
//   ./threads [repetitions = 20]      (needs an MI355X; exit 77 without a device)
//
// Every thread's results are digested (FNV-1a over the bytes of every number handed back) and compared with the same work done
// serially first: bit-identical or the program fails. The serial digests are printed too, so the GPU test can pin them to the CPU
// restatement's.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include "ephemeris_amd.hpp"

namespace ea = ephemeris_amd;

namespace {
struct Digest {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    }
    void add(double x) { bytes(&x, sizeof(x)); }
    void add(int64_t x) { bytes(&x, sizeof(x)); }
    void add(const std::vector<double> &v) { if (!v.empty()) bytes(v.data(), v.size() * sizeof(double)); }
    void add(const std::vector<int32_t> &v) { if (!v.empty()) bytes(v.data(), v.size() * sizeof(int32_t)); }
    void add(const std::vector<ea::DVec3> &v) { if (!v.empty()) bytes(v.data(), v.size() * sizeof(ea::DVec3)); }
    void add(const ea::Solution &s) {
        const int32_t nb = s.bodies();
        for (int32_t b = 0; b < nb; ++b) {
            add(s.start(b)); add(s.interval(b)); add(s.len(b));
            std::vector<double> c;
            std::vector<int32_t> nc;
            s.polynomials(b, c, nc);
            add(c); add(nc);
        }
    }
};

const std::vector<double> kMu = {132712440041.27942, 398600.43550702266, 4902.80011845755};
const std::vector<ea::DVec3> kY = {{130800.7436285839, 344339.3116943656, 136496.914202216},
                                   {-27204249.66910069, 132940582.438431, 57641619.74238631},
                                   {-27017766.52877057, 133253431.1006455, 57806029.23241135}};
const std::vector<ea::DVec3> kDy = {{-0.007799748521575531, -0.005561934613704532, -0.00225317087714714},
                                    {-29.75359910616436, -5.189518219844614, -2.251561710555783},
                                    {-30.64009897505477, -4.820684674596127, -2.032529075882219}};
const ea::StateVector kCraft{{-27204249.668775786, 132947582.43848978, 57641619.74241204}, {-22.207539106181895, -5.189518219791726, -2.2515617105336263}};
constexpr double kT0 = -252460800.0, kDt = 21600.0, kDay = 86400.0;

// A / B: the N-body task -- step, take_solution + clone every few steps (prediction.rs:422-443); the clone carries on (prediction.rs:378)
uint64_t nbody_task(ea::Direction dir) {
    Digest d;
    ea::NBodyPropagator p(kY, kDy, kMu, kT0, kDt, dir, {2, 2, 1}, {6, 7, 6});
    for (int k = 0; k < 24; ++k) {
        const ea::StepError e = p.step_n(5);
        if (e != ea::StepError::None) throw std::runtime_error("nbody_task: step error");
        ea::Solution s = p.take_solution();
        d.add(s);
        d.add(p.time());
        if (k % 4 == 3) {
            ea::NBodyPropagator c = p.clone();
            p = std::move(c);                        // the stored propagator is replaced by its snapshot
        }
    }
    return d.h;
}

// the bodies' table a ship task evaluates: the forward propagator run to 40 days, as a Solution (built on the calling thread)
ea::Solution bodies_to(double days) {
    ea::NBodyPropagator p(kY, kDy, kMu, kT0, kDt, ea::Direction::Forward, {2, 2, 1}, {6, 7, 6});
    ea::StepError e = ea::StepError::None;
    ea::Solution s = p.propagate(kT0 + days * kDay, &e);
    if (e != ea::StepError::None) throw std::runtime_error("bodies_to: step error");
    return s;
}

void add_batch(Digest &d, const ea::SpacecraftBatch &b) {
    std::vector<int32_t> nk;
    d.add(b.status(&nk));
    d.add(nk);
    for (int64_t i = 0; i < b.len(); ++i) {
        const ea::CubicHermiteSpline sp = b.trajectory(i);
        d.add(sp.t); d.add(sp.position); d.add(sp.velocity);
    }
}

std::vector<ea::StateVector> fleet(int n) {
    std::vector<ea::StateVector> v;
    for (int i = 0; i < n; ++i) {
        ea::StateVector s = kCraft;
        s.position[0] += 7.0 * i; s.position[2] -= 3.0 * i; s.velocity[1] += 1e-4 * i;
        v.push_back(s);
    }
    return v;
}

// C: a ship task against its OWN table -- step_to in pieces, a snapshot (clone) per piece that replaces the stored propagator
uint64_t ship_task(const ea::Ephemeris &table, const char *method, int n) {
    Digest d;
    ea::SpacecraftBatch b(table, kT0, fleet(n), method, ea::AdaptiveParams(1e-3), {}, 8192);
    for (int k = 1; k <= 6; ++k) {
        b.step_to(kT0 + 0.5 * k * kDay);
        if (k % 2 == 0) {
            ea::SpacecraftBatch c = b.clone();
            add_batch(d, c);
            b = std::move(c);
        }
    }
    add_batch(d, b);
    return d.h;
}

// D: seams 1 and 2 in a loop (their staging buffers were process-wide until round 6): eval + advance + read the state back, every step
uint64_t seam_task(int n, int loops) {
    std::vector<ea::DVec3> y(static_cast<size_t>(n)), dy(static_cast<size_t>(n));
    std::vector<double> mu(static_cast<size_t>(n));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&s] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return static_cast<double>(s >> 11) * (1.0 / 9007199254740992.0); };
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) { y[i][c] = (rnd() - 0.5) * 2e6; dy[i][c] = (rnd() - 0.5) * 1e-1; }
        mu[i] = 1.0 + 10.0 * rnd();
    }
    Digest d;
    ea::NBodyIntegration g(y, dy, mu, 0.0, 10.0);
    std::vector<ea::DVec3> py, pdy, ddy(static_cast<size_t>(n));
    for (int k = 0; k < loops; ++k) {
        if (g.advance(1) != ea::StepError::None) throw std::runtime_error("seam_task: step error");
        uint32_t sc = 0;
        d.add(g.state(py, pdy, &sc));
        d.add(py); d.add(pdy);
        for (ea::DVec3 &a : ddy) a = {0.0, 0.0, 0.0};
        ea::newtonian_gravity_eval(py, mu, ddy);
        d.add(ddy);
    }
    return d.h;
}

// E: the LIVE table. The writer merges the bodies' snapshots as the N-body task would send them (dynamics/celestial.rs:198-204); the
// ship task restarts its stored propagator whenever the context has become valid at its next target (flight_plan.rs:363-395). The
// margin keeps every evaluation of a leg inside the table as it was when the leg started, so the knots do not depend on how far the
// writer has got -- the digest equals the serial run's, whatever the interleaving.
struct Live {
    std::vector<ea::Solution> pieces;            // piece 0 seeds the table; the rest are merged in order
};
Live live_pieces() {
    Live l;
    ea::NBodyPropagator p(kY, kDy, kMu, kT0, kDt, ea::Direction::Forward, {2, 2, 1}, {6, 7, 6});
    for (int k = 1; k <= 12; ++k) {
        if (p.step_to(kT0 + 2.0 * k * kDay) != ea::StepError::None) throw std::runtime_error("live_pieces: step error");
        l.pieces.push_back(p.take_solution());
    }
    return l;
}
uint64_t live_task(const Live &l, bool threaded, const char *method) {
    auto table = std::make_shared<ea::Ephemeris>(l.pieces[0], kMu);
    std::atomic<bool> failed{false};
    auto writer = [&] {
        try {
            for (size_t k = 1; k < l.pieces.size(); ++k) {
                table->merge(l.pieces[k]);
                if (threaded) std::this_thread::sleep_for(std::chrono::microseconds(300));
            }
        } catch (...) { failed = true; }
    };
    Digest d;
    auto reader = [&] {
        try {
            ea::SpacecraftBatch b(*table, kT0, fleet(3), method, ea::AdaptiveParams(1e-3), {}, 32768);
            const double margin = 1.0 * kDay;
            for (int leg = 1; leg <= 10; ++leg) {
                const double target = kT0 + 2.0 * leg * kDay;
                while (!table->is_valid_at(target + margin)) {
                    if (failed) return;
                    std::this_thread::yield();
                }
                b.step_to(target);
            }
            add_batch(d, b);
        } catch (...) { failed = true; }
    };
    if (threaded) {
        std::thread w(writer), r(reader);
        w.join(); r.join();
    } else {
        writer(); reader();
    }
    if (failed) throw std::runtime_error("live_task failed");
    d.add(static_cast<int64_t>(table->revision()));
    return d.h;
}

// F: a propagator and a batch created on one thread, used on a second, destroyed on a third (the task pool moves them)
uint64_t moved_task(const ea::Ephemeris &table, bool threaded) {
    std::unique_ptr<ea::NBodyPropagator> p;
    std::unique_ptr<ea::SpacecraftBatch> b;
    Digest d;
    auto make = [&] {
        p.reset(new ea::NBodyPropagator(kY, kDy, kMu, kT0, kDt, ea::Direction::Forward, {2, 2, 1}, {6, 7, 6}));
        b.reset(new ea::SpacecraftBatch(table, kT0, fleet(2), "Verner87", ea::AdaptiveParams(1e-3), {}, 4096));
    };
    auto use = [&] {
        p->step_n(40);
        d.add(p->take_solution());
        b->step_to(kT0 + 1.0 * kDay);
        add_batch(d, *b);
    };
    auto drop = [&] { p.reset(); b.reset(); };
    if (threaded) {
        std::thread(make).join();
        std::thread(use).join();
        std::thread(drop).join();
    } else {
        make(); use(); drop();
    }
    return d.h;
}
}  // namespace

int main(int argc, char **argv) try {
    const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
    std::printf("abi %d, %d device(s), %u hardware threads\n", (int)eph_abi_version(), (int)ea::device_count(), std::thread::hardware_concurrency());
    const ea::Solution sol40 = bodies_to(40.0);
    const ea::Ephemeris table_c(sol40, kMu), table_c2(sol40, kMu), table_f(sol40, kMu);
    const Live live = live_pieces();

    // the serial run: what every thread must reproduce
    const uint64_t want_a = nbody_task(ea::Direction::Forward), want_b = nbody_task(ea::Direction::Backward);
    const uint64_t want_c = ship_task(table_c, "DormandPrince54", 5), want_c2 = ship_task(table_c2, "Verner87", 130);
    const uint64_t want_d = seam_task(512, 40), want_d2 = seam_task(33, 60);
    const uint64_t want_e = live_task(live, false, "Verner87"), want_e2 = live_task(live, false, "DormandPrince54");
    const uint64_t want_f = moved_task(table_f, false);
    std::printf("serial: forward %016llx backward %016llx ship %016llx fleet %016llx seams %016llx %016llx live %016llx %016llx moved %016llx\n",
                (unsigned long long)want_a, (unsigned long long)want_b, (unsigned long long)want_c, (unsigned long long)want_c2,
                (unsigned long long)want_d, (unsigned long long)want_d2, (unsigned long long)want_e, (unsigned long long)want_e2,
                (unsigned long long)want_f);

    int bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        uint64_t got[9] = {};
        std::atomic<int> errors{0};
        auto run = [&errors](uint64_t *out, std::function<uint64_t()> f) {
            return std::thread([out, f, &errors] {
                try { *out = f(); } catch (const std::exception &e) { std::fprintf(stderr, "thread: %s\n", e.what()); errors += 1; }
            });
        };
        std::vector<std::thread> th;
        th.push_back(run(&got[0], [] { return nbody_task(ea::Direction::Forward); }));
        th.push_back(run(&got[1], [] { return nbody_task(ea::Direction::Backward); }));
        th.push_back(run(&got[2], [&] { return ship_task(table_c, "DormandPrince54", 5); }));
        th.push_back(run(&got[3], [&] { return ship_task(table_c2, "Verner87", 130); }));
        th.push_back(run(&got[4], [] { return seam_task(512, 40); }));
        th.push_back(run(&got[5], [] { return seam_task(33, 60); }));
        th.push_back(run(&got[6], [&] { return live_task(live, true, "Verner87"); }));
        th.push_back(run(&got[7], [&] { return live_task(live, true, "DormandPrince54"); }));
        th.push_back(run(&got[8], [&] { return moved_task(table_f, true); }));
        for (std::thread &t : th) t.join();
        const uint64_t want[9] = {want_a, want_b, want_c, want_c2, want_d, want_d2, want_e, want_e2, want_f};
        static const char *name[9] = {"forward", "backward", "ship", "fleet", "seams-512", "seams-33", "live-V87", "live-DP54", "moved"};
        for (int i = 0; i < 9; ++i)
            if (got[i] != want[i]) {
                std::printf("repetition %d: %s differs from the serial run (%016llx vs %016llx)\n", rep, name[i], (unsigned long long)got[i], (unsigned long long)want[i]);
                bad += 1;
            }
        bad += errors;
    }
    std::printf("%d repetitions x 9 concurrent tasks (13 threads): %s\n", reps, bad ? "MISMATCH" : "every result bit-identical to the serial run");
    return bad ? 1 : 0;
} catch (const ea::Error &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return e.status == EPH_ERR_NO_DEVICE ? 77 : 1;
} catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
}
