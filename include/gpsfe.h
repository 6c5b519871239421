/*
 * gpsfe.h — host front end for libgpsbb: from {RINEX-2 navigation file, receiver position or motion,
 * start time} to the per-block channel descriptors (gpsbb_chan_t) the IQ fill consumes.
 *
 * This is the part of the reference that stays on the host.  It reproduces,
 * bit for bit, what the reference's main() leaves in chan[] / gain[] before every run of the sample loop:
 *   ephemeris reader           readRinex2 / readRinex3   plutogpssim.c:874-1233, 1241-1610
 *   scenario start / eph set   main()                plutogpssim.c:2497-2597
 *   orbit + clock              satpos                plutogpssim.c:443-546
 *   range, az/el, Klobuchar    computeRange          plutogpssim.c:1691-1747, 1612-1683
 *   nav message                eph2sbf, generateNavMsg, computeChecksum   c:552-723, 1820-1894, 751-814
 *   channel allocation         allocateChannel       plutogpssim.c:1918-1989
 *   per-block seeding          computeCodePhase + gain   plutogpssim.c:1754-1787, 2656-2687
 *   30 s maintenance           main()                plutogpssim.c:2764-2805
 * Plain C, scalar doubles, built -ffp-contract=off -fno-builtin so every libm call and rounding is the
 * reference's.  Checked field by field (bitwise) against descriptor dumps of the reference's own code.
 */
#ifndef GPSFE_H
#define GPSFE_H

#include "gpsbb.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GPSFE_OK 0
#define GPSFE_E_BADARG (-1)
#define GPSFE_E_NAVFILE (-2)   /* cannot open / not a RINEX-2 GPS nav file / no ephemeris          */
#define GPSFE_E_MOTION (-3)    /* cannot open / empty user-motion file                              */
#define GPSFE_E_TIME (-4)      /* start time outside the ephemeris window (c:2555-2564)             */
#define GPSFE_E_NOEPH (-5)     /* no ephemeris set within an hour of the start time (c:2593-2596)   */
#define GPSFE_E_NOMEM (-6)

typedef struct gpsfe gpsfe_t;

typedef struct gpsfe_config {
    const char *navfile;     /* -e : RINEX navigation file (plain or gzip)                          */
    int rinex3;              /* -3 : the file is RINEX 3 (readRinex3 c:1241-1610) instead of 2      */
    const char *motion_file; /* -u : "t,x,y,z" ECEF at 10 Hz; NULL = static position               */
    int use_ecef;            /* static position given as ECEF (-c) instead of lat,lon,height (-l)  */
    double pos[3];           /* -l: degrees, degrees, metres   /   -c: metres ECEF                  */
    int have_start;          /* -t given                                                            */
    int y, m, d, hh, mm;     /* -t YYYY/MM/DD,hh:mm:ss                                              */
    double sec;
    int time_overwrite;      /* -T : overwrite TOC/TOE to the scenario start time                   */
    int iono_disable;        /* -i                                                                  */
    int max_chan;            /* channels to allocate: 12 in the reference (h:21), up to 16 here     */
    int fixed_carrier;       /* descriptors for GPSBB_FIXED_CARRIER: carr_phase is the 32-bit accumulator
                                value the `#ifndef FLOAT_CARR_PHASE` code initialises at c:1966-1967 */
} gpsfe_config_t;

int gpsfe_open(const gpsfe_config_t *cfg, gpsfe_t **out);
void gpsfe_close(gpsfe_t *fe);
const char *gpsfe_strerror(int err);

/* number of channels (cfg.max_chan) */
int gpsfe_max_chan(const gpsfe_t *fe);

/*
 * Descriptors of the next 0.1 s block (what c:2656-2687 computes) into ch[max_chan], then advance the
 * scenario by one block including the 30-second maintenance (c:2764-2805).
 * ch[i].carr_phase is the value the front end knows: the initial phase of a channel when it was
 * allocated (c:1956-1964) or whatever was fed back with gpsfe_feed_back().  Callers that render the
 * blocks in time order on one handle use GPSBB_CHAIN_CARRIER and never feed back; callers that shard the
 * stream use gpsbb_chain_carrier_host() on the descriptor sequence.
 */
int gpsfe_next_block(gpsfe_t *fe, gpsbb_chan_t *ch);

/* Optional: give the front end the carrier phases the fill left behind (end_state of the block just
 * rendered), exactly as the reference's loop updates chan[i].carr_phase in place. */
int gpsfe_feed_back(gpsfe_t *fe, const gpsbb_chan_state_t *end_state);

/* nblocks consecutive blocks, block-major into ch[nblocks*max_chan] (no feedback).  The blocks between two 30 s
 * maintenances (c:2764-2798) are independent given the ranges at their ends, so they are spread over threads: the same
 * descriptors, bit for bit, as nblocks calls of gpsfe_next_block. */
int gpsfe_generate(gpsfe_t *fe, int nblocks, gpsbb_chan_t *ch);

/* threads gpsfe_generate uses: 0 = the machine's cores up to 16 (default), 1 = none (the reference's one generator thread,
 * c:2286-2289), up to 32 */
int gpsfe_set_threads(gpsfe_t *fe, int nthreads);

/* introspection for tests / logging: current receiver GPS time and the visible-satellite table */
int gpsfe_time(const gpsfe_t *fe, int *week, double *sec);
int gpsfe_channel_info(const gpsfe_t *fe, int i, int *prn, double *az_deg, double *el_deg, double *range_m,
                       double *iono_m);

#ifdef __cplusplus
}
#endif
#endif
