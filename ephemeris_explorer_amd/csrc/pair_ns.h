// pair_ns.h -- the translation units that contain the point-mass term (step_wave.hip, step_wg.hip, step_small.hip, fast.hip,
// craft_sweep.hip) are compiled ONCE PER EVALUATION ORDER of that term (-DEPH_PAIR_VARIANT=k, k = 0..6; pair_term.h), every
// symbol inside namespace eph::pv<k>, and all seven sets are linked into the one library: the order is a compile-time constant
// in every kernel and a run-time choice of the caller (eph_set_pair_variant; dispatch.cpp routes a handle's launches to its
// namespace's table).
#pragma once
#ifndef EPH_PAIR_VARIANT
#error "this file is compiled once per evaluation order of the point-mass term: -DEPH_PAIR_VARIANT=k (build.py does it)"
#endif
#define EPH_PV_CAT2(a, b) a##b
#define EPH_PV_CAT(a, b) EPH_PV_CAT2(a, b)
#define EPH_PV_NS EPH_PV_CAT(pv, EPH_PAIR_VARIANT)
#include "eph_internal.h"
#include "ieee_seq.h"
#include "force_common.h"
namespace eph {
namespace EPH_PV_NS {
#include "pair_term.h"
#include "pair_launchers.h"
}  // namespace EPH_PV_NS
}  // namespace eph
