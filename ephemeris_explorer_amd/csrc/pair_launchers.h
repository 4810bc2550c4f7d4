// pair_launchers.h -- what one evaluation order's translation units export to each other and, through kTable, to dispatch.cpp.
// INCLUDED INSIDE namespace eph::pv<k> (pair_ns.h).
int accel_wave(hipStream_t s, int bpw, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out, int lo, int hi,
               const KickDrift &kd);                                                                        // step_wave.hip
int accel_wg(hipStream_t s, int wb, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out, int lo, int hi,
             const KickDrift &kd);                                                                          // step_wg.hip
int lm_step_wave(hipStream_t s, int bpw, const LmArgs &a);                                                 // step_wave.hip
int lm_step_wg(hipStream_t s, int wb, const LmArgs &a);                                                    // step_wg.hip
int lm_persistent(hipStream_t s, const LmArgs &a, int64_t nsteps);                                         // step_wave.hip
int lm_small(hipStream_t s, const LmArgs &a, int64_t nsteps);                                              // step_small.hip
int lm_small_many(hipStream_t s, const LmArgs *argv_dev, int count, int L, int64_t nsteps);                // step_small.hip
int lm_step_fast(hipStream_t s, const LmArgs &a, double *partial, int S, int unroll, bool approx, float *posf, int f32_stage,
                 int conv_lo, int conv_cnt, unsigned *tickets);                                             // fast.hip
int craft_launch(hipStream_t s, const CraftArgs &a, const CraftLaunch &how);                               // craft_sweep.hip
int debug_inv_r3(hipStream_t s, int64_t n, const double *n2, double *fast, double *ieee);                  // step_wave.hip
int debug_inv_r3_sweep(hipStream_t s, uint64_t seed, int64_t n, unsigned long long *out2);
int debug_quot(hipStream_t s, int64_t n, const double *x, const double *a, double *fast, double *ieee);
int debug_wg_cycles(long long *out8);                                                                      // step_small.hip
const PairKernels *pair_table();                                                                           // step_wave.hip
