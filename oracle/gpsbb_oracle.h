/*
 * gpsbb_oracle.h — CPU restatement of the reference's IQ fill loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product path (include/, pluto-gps-sim_amd/) may include, link or call this.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker /
 * the timed CPU baseline.
 *
 * Pinning status: the restatement is checked bit-for-bit against the reference's own statements —
 * the loop body plutogpssim.c:2690-2756 and the front-end functions plutogpssim.c:93-1989 compiled
 * verbatim from /root/reference by oracle/ref/build_ref.sh into oracle/_ref/ (see oracle/ref/README.md
 * for exactly which lines are the reference's and which are harness glue) — by tests/test_oracle.py and tests/test_frontend.py
 * in the build container, and against the golden vectors generated from that build and committed under
 * tests/golden/ everywhere else.  The reference ships no tests or golden vectors of its own, and its
 * whole-program build needs libiio/libad9361/libcurl headers this image lacks, so no whole-program
 * run is used.
 */
#ifndef GPSBB_ORACLE_H
#define GPSBB_ORACLE_H

#include "../include/gpsbb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sinTable512 / cosTable512 of plutogpssim.c:93-161, regenerated from their closed form
 * trunc(511*f(2*pi*i/512) + 1.0) (checked entry by entry against the reference's arrays). */
void gpsbb_oracle_tables(int sin512[512], int cos512[512]);

/* codegen(), plutogpssim.c:207-244: G1/G2 LFSRs + per-PRN G2 delay -> ca[1023] of 0/1. */
void gpsbb_oracle_codegen(int *ca, int prn);

/*
 * The sample loop, plutogpssim.c:2690-2756, over descriptors instead of channel_t.
 * Sequential, IEEE double NCOs, same operation order.  Returns 0, or -1 on a descriptor outside the
 * contract of gpsbb_chan_t.  end_state / hz may be NULL.
 */
int gpsbb_oracle_fill(const gpsbb_chan_t *ch, int nch, double delt, int nsamp, int16_t *iq,
                      gpsbb_chan_state_t *end_state, gpsbb_hazards_t *hz);

/* Consecutive blocks with the carrier phase carried from block to block (what iterating
 * plutogpssim.c:2655 does); block-major descriptors, iq = nblocks*nsamp*2 int16 (may be NULL to only
 * obtain end states), end_state = nblocks*nch. chain=0 treats blocks as independent. */
int gpsbb_oracle_fill_blocks(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                             int chain, int16_t *iq, gpsbb_chan_state_t *end_state,
                             gpsbb_hazards_t *hz);

/* The reference's other carrier NCO (the `#ifndef FLOAT_CARR_PHASE` code, compiled out as shipped:
 * h:12, 157-162; c:1966-1967, 2675, 2699, 2748): a 32-bit accumulator, step
 * (int)round(2^25 * f_carr * delt) per block, table index (phase >> 16) & 0x1ff.  ch[].carr_phase holds
 * the accumulator's value (an integer in [0, 2^32)); end_state[].carr_phase likewise. */
int gpsbb_oracle_fill_blocks_fixed(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                                   int chain, int16_t *iq, gpsbb_chan_state_t *end_state,
                                   gpsbb_hazards_t *hz);

#ifdef __cplusplus
}
#endif
#endif
