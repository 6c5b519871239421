/*
 * ref_harness.c — HARNESS GLUE around verbatim slices of the reference.  TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code.  Every `#include "slice_*.inc"` below pulls in a line range of
 * /root/reference/plutogpssim.c that oracle/ref/build_ref.sh cuts out at build time into a temporary
 * directory (never into the repo).  What is glue (written here) and what is the reference's own text:
 *
 *   reference text (compiled unchanged)           glue (this file)
 *   ------------------------------------------    -------------------------------------------------
 *   c:93-1989   tables, codegen, geodesy,         the #include block the reference has at c:12-31
 *               satpos, eph2sbf, checksum,          minus <curl/curl.h> <iio.h> <ad9361.h>;
 *               RINEX readers, computeRange,        `rinex_date` (c:86); NUM_SAMPLES as a run-time
 *               computeCodePhase, generateNavMsg,   variable instead of c:43-44's constant; MAX_CHAN
 *               allocateChannel                     optionally raised from 12 (h:21) to 16
 *   c:2497-2569 start-time window                 declarations of main()'s locals (c:2204-2253) so the
 *   c:2576-2597 ephemeris-set selection             slices compile; option handling for -e -l -c -u
 *   c:2620-2632 channel init, allocateChannel       -t -T -s -i restated from c:2296-2390; the
 *   c:2645-2646 antenna pattern                     `for (blk...)` that stands in for
 *   c:2653      grx += 0.1                          `while (!plutotx.exit)` (c:2655); the dump hooks
 *   c:2656-2687 per-block seeding                   between the slices; conversion channel_t ->
 *   c:2690-2756 THE SAMPLE LOOP                     gpsbb_chan_t for the dumps
 *   c:2764-2805 30-s maintenance, time update
 *
 * Not built: the libiio TX thread (c:2058-2190), FTP fetch (c:2428-2474), signal/affinity code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <errno.h>
#include <unistd.h>
#include <stdbool.h>
#include <stdint.h>
#include <limits.h>
#include <sys/types.h>
#include <zlib.h>

#include "plutogpssim.h" /* the reference's own header, found via -I/root/reference */
#include "gpsbb.h"       /* descriptor structs used for the dumps */

#ifdef REF_MAX_CHAN
#undef MAX_CHAN
#define MAX_CHAN (REF_MAX_CHAN)
#endif

/* c:43-44 make the block length a compile-time constant; here it is a variable so one build serves
 * every block length.  The sample loop slice only uses it as the loop bound. */
static int ref_nsamp = 300000;
#define NUM_SAMPLES ref_nsamp

static char rinex_date[21]; /* c:86 */
#define NOTUSED(V) ((void)V) /* c:40 */

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wunused-function"
#include "slice_front.inc" /* c:93-1989 */
#pragma GCC diagnostic pop

/* ---- exports for ctypes ------------------------------------------------------------------------- */

void ref_tables(const int **sin512, const int **cos512)
{
    *sin512 = sinTable512;
    *cos512 = cosTable512;
}

void ref_codegen(int *ca, int prn) { codegen(ca, prn); }

int ref_max_chan(void) { return MAX_CHAN; }

static void chan_to_desc(gpsbb_chan_t *d, const channel_t *c, double gain)
{
    memset(d, 0, sizeof *d);
    d->prn = c->prn;
    if (c->prn <= 0)
        return;
    d->iword = c->iword;
    d->ibit = c->ibit;
    d->icode = c->icode;
    d->f_carr = c->f_carr;
    d->f_code = c->f_code;
    d->carr_phase = c->carr_phase;
    d->code_phase = c->code_phase;
    d->gain = gain;
    for (int k = 0; k < N_DWRD; k++)
        d->dwrd[k] = (uint32_t)c->dwrd[k];
}

static void chan_to_state(gpsbb_chan_state_t *s, const channel_t *c)
{
    memset(s, 0, sizeof *s);
    if (c->prn <= 0)
        return;
    s->carr_phase = c->carr_phase;
    s->code_phase = c->code_phase;
    s->iword = c->iword;
    s->ibit = c->ibit;
    s->icode = c->icode;
    s->dataBit = c->dataBit;
    s->codeCA = c->codeCA;
}

/*
 * Run the reference's sample loop (c:2690-2756, verbatim) on caller-supplied descriptors.
 * The channel_t array is filled the way the reference's front end fills it: ca[] by codegen()
 * (c:1944), codeCA/dataBit by the two statements at c:1780-1781.
 */
int ref_loop_fill(const gpsbb_chan_t *ch, int nch, double delt_in, int nsamp, int16_t *iq_out,
                  gpsbb_chan_state_t *end_state)
{
    /* names below are the ones the slice refers to (main()'s locals, c:2211-2241, and c:84) */
    static channel_t chan[MAX_CHAN];
    double gain[MAX_CHAN];
    double delt = delt_in;
    int isamp, i, iTable;
    int ip, qp;
    short *iq_buff = (short *)iq_out;

    if (nch > MAX_CHAN || nch < 0)
        return -1;
    memset(chan, 0, sizeof chan);
    for (i = 0; i < MAX_CHAN; i++) {
        chan[i].prn = 0;
        gain[i] = 0.0;
    }
    for (i = 0; i < nch; i++) {
        if (ch[i].prn <= 0)
            continue;
        chan[i].prn = ch[i].prn;
        codegen(chan[i].ca, chan[i].prn);
        chan[i].f_carr = ch[i].f_carr;
        chan[i].f_code = ch[i].f_code;
        chan[i].carr_phase = ch[i].carr_phase;
        chan[i].code_phase = ch[i].code_phase;
        for (int k = 0; k < N_DWRD; k++)
            chan[i].dwrd[k] = ch[i].dwrd[k];
        chan[i].iword = ch[i].iword;
        chan[i].ibit = ch[i].ibit;
        chan[i].icode = ch[i].icode;
        /* as computeCodePhase leaves them (c:1780-1781) */
        chan[i].codeCA = chan[i].ca[(int)chan[i].code_phase] * 2 - 1;
        chan[i].dataBit = (int)((chan[i].dwrd[chan[i].iword] >> (29 - chan[i].ibit)) & 0x1UL) * 2 - 1;
        gain[i] = ch[i].gain;
#ifndef FLOAT_CARR_PHASE /* built against the header copy without h:12: the step main() sets at c:2675 */
        chan[i].carr_phasestep = (int)round(512.0 * 65536.0 * chan[i].f_carr * delt);
#endif
    }
    ref_nsamp = nsamp;

#include "slice_loop.inc" /* c:2690-2756 */

    if (end_state)
        for (i = 0; i < nch; i++)
            chan_to_state(&end_state[i], &chan[i]);
    return 0;
}

/* Front-end functions exposed for descriptor-level parity tests of a from-scratch front end. */
void ref_llh2xyz(const double *llh, double *xyz) { llh2xyz(llh, xyz); }
void ref_xyz2llh(const double *xyz, double *llh) { xyz2llh(xyz, llh); }
unsigned long ref_computeChecksum(unsigned long source, int nib) { return computeChecksum(source, nib); }

#ifdef REF_BUILD_MAIN
/* ---- scenario runner: main()'s control flow with the device I/O removed -------------------------- */

static void die(const char *m)
{
    fprintf(stderr, "ref_sim: %s\n", m);
    exit(2);
}

int main(int argc, char *argv[])
{
    /* main()'s locals (c:2204-2253), same names and types, so that the slices compile unchanged */
    int sv;
    int neph, ieph;
    static ephem_t eph[EPHEM_ARRAY_SIZE][MAX_SAT];
    gpstime_t g0;
    double llh[3];
    int i;
    static channel_t chan[MAX_CHAN];
    double elvmask = 0.0;
    int ip, qp;
    int iTable;
    gpstime_t grx;
    double delt;
    int isamp;
    int numd = 0, iumd = 0;
    static double xyz[USER_MOTION_SIZE][3];
    int staticLocationMode = true;
    const char *navfile = NULL;
    const char *umfile = NULL;
    double gain[MAX_CHAN];
    double path_loss;
    double ant_gain;
    double ant_pat[37];
    int ibs;
    datetime_t t0, tmin, tmax;
    gpstime_t gmin, gmax;
    double dt;
    int igrx;
    bool timeoverwrite = false;
    ionoutc_t ionoutc;
    short *iq_buff = NULL; /* c:84 */

    /* harness-only */
    long long fs_hz = 3000000; /* TX_SAMPLE_FREQ, c:43, 2271 */
    int nblocks = 1, result, blk, use_rinex3 = 0;
    const char *iq_path = NULL, *desc_path = NULL, *state_path = NULL;
    FILE *fiq = NULL, *fdesc = NULL, *fstate = NULL;

    memset(&tmin, 0, sizeof tmin);
    memset(&gmin, 0, sizeof gmin);
    memset(&t0, 0, sizeof t0);

    /* defaults, c:2261-2268 */
    g0.week = -1;
    ionoutc.enable = true;
    llh[0] = 35.681298 / R2D;
    llh[1] = 139.766247 / R2D;
    llh[2] = 10.0;
    llh2xyz(llh, xyz[0]);

    while ((result = getopt(argc, argv, "e:u:c:l:s:Tt:in:b:o:d:S:3")) != -1) {
        switch (result) {
        case 'e':
            navfile = optarg;
            break;
        case 'u': /* c:2301-2304 */
            umfile = optarg;
            staticLocationMode = false;
            break;
        case 'c': /* c:2312-2315 */
            sscanf(optarg, "%lf,%lf,%lf", &xyz[0][0], &xyz[0][1], &xyz[0][2]);
            break;
        case 'l': /* c:2316-2323 */
            sscanf(optarg, "%lf,%lf,%lf", &llh[0], &llh[1], &llh[2]);
            llh[0] = llh[0] / R2D;
            llh[1] = llh[1] / R2D;
            llh2xyz(llh, xyz[0]);
            break;
        case 's': /* c:2324-2330 */
            fs_hz = (long long)atoi(optarg);
            if (fs_hz < 1000000)
                die("invalid sampling frequency");
            break;
        case 'T': /* c:2331 (only the flag; "-T now" is not supported by the harness) */
            timeoverwrite = true;
            break;
        case 't': /* c:2350-2359 */
            sscanf(optarg, "%d/%d/%d,%d:%d:%lf", &t0.y, &t0.m, &t0.d, &t0.hh, &t0.mm, &t0.sec);
            if (t0.y <= 1980 || t0.m < 1 || t0.m > 12 || t0.d < 1 || t0.d > 31 || t0.hh < 0 ||
                t0.hh > 23 || t0.mm < 0 || t0.mm > 59 || t0.sec < 0.0 || t0.sec >= 60.0)
                die("invalid date and time");
            t0.sec = floor(t0.sec);
            date2gps(&t0, &g0);
            break;
        case '3': /* c:2305-2308 (the reference's getopt string gives -3 an argument; a flag here) */
            use_rinex3 = 1;
            break;
        case 'i': /* c:2360-2362 */
            ionoutc.enable = false;
            break;
        case 'n':
            ref_nsamp = atoi(optarg);
            break;
        case 'b':
            nblocks = atoi(optarg);
            break;
        case 'o':
            iq_path = optarg;
            break;
        case 'd':
            desc_path = optarg;
            break;
        case 'S':
            state_path = optarg;
            break;
        default:
            die("usage: ref_sim -e nav [-l lat,lon,h|-c x,y,z|-u motion.csv] [-t date] [-T] [-i] "
                "-s fs -n nsamp -b nblocks [-o iq.bin] [-d desc.bin] [-S state.bin]");
        }
    }
    if (navfile == NULL)
        die("no ephemeris file");

    delt = 1.0 / fs_hz; /* c:2397 (fs_hz is long long there too) */

    if (!staticLocationMode) { /* c:2403-2415 */
        numd = readUserMotion(xyz, umfile);
        if (numd == -1)
            die("failed to open user motion file");
        else if (numd == 0)
            die("failed to read user motion data");
    }

    neph = use_rinex3 ? readRinex3(eph, &ionoutc, navfile) : readRinex2(eph, &ionoutc, navfile); /* c:2476-2480 */
    if (neph == 0)
        die("no ephemeris available");

#include "slice_timewin.inc" /* c:2497-2569 */
#include "slice_ephsel.inc"  /* c:2576-2597 */

    iq_buff = calloc((size_t)NUM_SAMPLES, 4); /* c:2604 */
    if (!iq_buff)
        die("calloc");
    if (iq_path && !(fiq = fopen(iq_path, "wb")))
        die("cannot open iq output");
    if (desc_path && !(fdesc = fopen(desc_path, "wb")))
        die("cannot open descriptor output");
    if (state_path && !(fstate = fopen(state_path, "wb")))
        die("cannot open state output");

#include "slice_chaninit.inc" /* c:2620-2632 */

    for (i = 0; i < MAX_CHAN; i++) /* the table the reference prints at c:2634-2639 */
        if (chan[i].prn > 0)
            fprintf(stderr, "%02d %6.1f %5.1f %11.1f %5.1f\n", chan[i].prn, chan[i].azel[0] * R2D,
                    chan[i].azel[1] * R2D, chan[i].rho0.d, chan[i].rho0.iono_delay);

#include "slice_antpat.inc" /* c:2645-2646 */
#include "slice_grx0.inc"   /* c:2653 */

    for (blk = 0; blk < nblocks; blk++) { /* stands in for `while (!plutotx.exit)` c:2655 */
#include "slice_seed.inc" /* c:2656-2687 */

        if (fdesc) {
            gpsbb_chan_t d;
            for (i = 0; i < MAX_CHAN; i++) {
                chan_to_desc(&d, &chan[i], chan[i].prn > 0 ? gain[i] : 0.0);
                fwrite(&d, sizeof d, 1, fdesc);
            }
        }

#include "slice_loop.inc" /* c:2690-2756 */

        if (fiq)
            fwrite(iq_buff, 4, (size_t)NUM_SAMPLES, fiq);
        if (fstate) {
            gpsbb_chan_state_t s;
            for (i = 0; i < MAX_CHAN; i++) {
                chan_to_state(&s, &chan[i]);
                fwrite(&s, sizeof s, 1, fstate);
            }
        }

#include "slice_maint.inc" /* c:2764-2805 */
    }

    if (fiq)
        fclose(fiq);
    if (fdesc)
        fclose(fdesc);
    if (fstate)
        fclose(fstate);
    free(iq_buff);
    (void)ip;
    (void)qp;
    (void)iTable;
    (void)isamp;
    (void)tmax;
    return 0;
}
#endif /* REF_BUILD_MAIN */
