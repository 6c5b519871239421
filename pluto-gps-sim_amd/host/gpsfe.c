/*
 * gpsfe.c — host front end for libgpsbb (see include/gpsfe.h).
 *
 * A from-scratch restatement of the scalar host code of pluto-gps-sim that produces the channel state the
 * IQ fill consumes.  Structure, names and I/O are ours; the ARITHMETIC follows the reference statement by
 * statement (operation order, intermediate roundings, integer truncations, libm calls), because the
 * descriptors feed threshold functions (chip index, table index, data bit) and a 1-ulp difference in a
 * frequency or phase flips output samples.  Build: -std=c11 -O2 -ffp-contract=off -fno-builtin (no FMA,
 * no sin/cos -> sincos merging, no compile-time folding of libm calls).
 *
 * Every function cites the reference lines it follows (plutogpssim.c = c:, plutogpssim.h = h:).
 */
#include "gpsfe.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

/* ---- constants (h:40-76) --------------------------------------------------------------------------- */
#define N_SAT 32       /* MAX_SAT h:18 */
#define N_EPH_SETS 13  /* EPHEM_ARRAY_SIZE h:78 */
#define N_MOTION 3000  /* USER_MOTION_SIZE h:24-26 */
#define K_PI 3.1415926535898
#define K_GM 3.986005e14
#define K_OMEGA_E 7.2921151467e-5
#define K_C 2.99792458e8
#define K_LAMBDA 0.190293672798365
#define K_R2D 57.2957795131
#define K_WGS_A 6378137.0
#define K_WGS_E 0.0818191908426
#define K_WEEK 604800.0
#define K_HALF_WEEK 302400.0
#define K_DAY 86400.0
#define K_HOUR 3600.0
#define P2_5 0.03125
#define P2_19 1.907348632812500e-6
#define P2_29 1.862645149230957e-9
#define P2_31 4.656612873077393e-10
#define P2_33 1.164153218269348e-10
#define P2_43 1.136868377216160e-13
#define P2_55 2.775557561562891e-17
#define P2_50 8.881784197001252e-016
#define P2_30 9.313225746154785e-010
#define P2_27 7.450580596923828e-009
#define P2_24 5.960464477539063e-008

typedef struct { int week; double sec; } gtime_t;                 /* gpstime_t h:81-84 */
typedef struct { int y, m, d, hh, mm; double sec; } caltime_t;    /* datetime_t h:87-94 */

typedef struct {  /* ephem_t h:97-130 */
    int valid;
    caltime_t t;
    gtime_t toc, toe;
    int iodc, iode;
    double deltan, cuc, cus, cic, cis, crc, crs, ecc, sqrta, m0, omg0, inc0, aop, omgdot, idot;
    double af0, af1, af2, tgd;
    int svhlth, codeL2;
    double n, sq1e2, A, omgkdot; /* working values c:1217-1221 */
} eph_t;

typedef struct {  /* ionoutc_t h:132-140 */
    int enable, valid;
    double alpha[4], beta[4], A0, A1;
    int dtls, tot, wnt;
} iono_t;

typedef struct {  /* range_t h:142-149 */
    gtime_t g;
    double range, rate, d, azel[2], iono_delay;
} range_t;

typedef struct {  /* the host-side part of channel_t h:152-174 */
    int prn;
    double f_carr, f_code, carr_phase, code_phase;
    gtime_t g0;
    uint32_t sbf[5][10];
    uint32_t dwrd[GPSBB_N_DWRD];
    int iword, ibit, icode;
    double azel[2];
    range_t rho0;
    double gain;
} chan_t;

struct gpsfe {
    int max_chan;
    int emitted_prn[GPSBB_MAX_CHAN]; /* prn of each channel in the descriptors gpsfe_next_block last handed out */
    int fixed_carrier; /* the reference's `#ifndef FLOAT_CARR_PHASE` variant: 32-bit phase accumulator */
    eph_t eph[N_EPH_SETS + 1][N_SAT]; /* one spare, always-invalid set: the reference peeks at set ieph+1 (c:2777) */
    int neph, ieph;
    iono_t iono;
    gtime_t grx;
    int static_mode;
    double (*xyz)[3];
    int numd, iumd;
    chan_t chan[GPSBB_MAX_CHAN];
    int sat_chan[N_SAT]; /* allocatedSat c:171 */
    double ant_pat[37];
    int nthreads;          /* gpsfe_generate: threads the blocks of a span are spread over (gpsfe_set_threads) */
    struct fe_pool *pool;  /* ... started on first use */
    gtime_t *span_grx;     /* gpsfe_generate's scratch of a span (block times, position indices, ranges): kept between calls */
    int *span_ipos;
    range_t *span_rho;
};

/* receiver antenna attenuation in dB for boresight angle 0:5:180 deg (c:164-169) */
static const double k_ant_pat_db[37] = {
    0.00, 0.00, 0.22, 0.44, 0.67, 1.11, 1.56, 2.00, 2.44, 2.89, 3.56, 4.22, 4.89, 5.56, 6.22, 6.89,
    7.56, 8.22, 8.89, 9.78, 10.67, 11.56, 12.44, 13.33, 14.44, 15.56, 16.67, 17.78, 18.89, 20.00, 21.33,
    22.67, 24.00, 25.56, 27.33, 29.33, 31.56};

/* ---- time (c:250-290, 838-866) ---------------------------------------------------------------------- */

static gtime_t cal_to_gps(const caltime_t *t) /* date2gps c:250-272 */
{
    static const int cum_days[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const int ye = t->y - 1980;
    int leap_days = ye / 4 + 1;
    if ((ye % 4) == 0 && t->m <= 2)
        leap_days--;
    const int de = ye * 365 + cum_days[t->m - 1] + t->d + leap_days - 6;
    gtime_t g;
    g.week = de / 7;
    g.sec = (double)(de % 7) * K_DAY + t->hh * K_HOUR + t->mm * 60.0 + t->sec;
    return g;
}

static caltime_t gps_to_cal(const gtime_t *g) /* gps2date c:274-290 */
{
    caltime_t t;
    const int c = (int)(7 * g->week + floor(g->sec / 86400.0) + 2444245.0) + 1537;
    const int d = (int)((c - 122.1) / 365.25);
    const int e = 365 * d + d / 4;
    const int f = (int)((c - e) / 30.6001);
    t.d = c - e - (int)(30.6001 * f);
    t.m = f - 1 - 12 * (f / 14);
    t.y = d - 4715 - ((7 + t.m) / 10);
    t.hh = ((int)(g->sec / 3600.0)) % 24;
    t.mm = ((int)(g->sec / 60.0)) % 60;
    t.sec = g->sec - 60.0 * floor(g->sec / 60.0);
    return t;
}

static double gps_diff(gtime_t a, gtime_t b) /* subGpsTime c:838-845 */
{
    double dt = a.sec - b.sec;
    dt += (double)(a.week - b.week) * K_WEEK;
    return dt;
}

static gtime_t gps_add(gtime_t g, double dt) /* incGpsTime c:847-866: rounds to the millisecond */
{
    g.sec = g.sec + dt;
    g.sec = round(g.sec * 1000.0) / 1000.0;
    while (g.sec >= K_WEEK) {
        g.sec -= K_WEEK;
        g.week++;
    }
    while (g.sec < 0.0) {
        g.sec += K_WEEK;
        g.week--;
    }
    return g;
}

/* ---- geodesy (c:178-201, 296-434) ---------------------------------------------------------------------- */

static double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

static void ecef_to_llh(const double *xyz, double *llh) /* xyz2llh c:296-341 */
{
    const double a = K_WGS_A, eps = 1.0e-3, e2 = K_WGS_E * K_WGS_E;
    if (norm3(xyz) < eps) { /* degenerate: the reference returns (0, 0, -a) */
        llh[0] = 0.0;
        llh[1] = 0.0;
        llh[2] = -a;
        return;
    }
    const double x = xyz[0], y = xyz[1], z = xyz[2];
    const double rho2 = x * x + y * y;
    double dz = e2 * z, zdz, nh, n;
    for (;;) {
        zdz = z + dz;
        nh = sqrt(rho2 + zdz * zdz);
        const double slat = zdz / nh;
        n = a / sqrt(1.0 - e2 * slat * slat);
        const double dz_new = n * e2 * slat;
        if (fabs(dz - dz_new) < eps)
            break;
        dz = dz_new;
    }
    llh[0] = atan2(zdz, sqrt(rho2));
    llh[1] = atan2(y, x);
    llh[2] = nh - n;
}

static void llh_to_ecef(const double *llh, double *xyz) /* llh2xyz c:347-378 */
{
    const double a = K_WGS_A, e = K_WGS_E, e2 = e * e;
    const double clat = cos(llh[0]), slat = sin(llh[0]), clon = cos(llh[1]), slon = sin(llh[1]);
    const double d = e * slat;
    const double n = a / sqrt(1.0 - d * d);
    const double nph = n + llh[2];
    const double tmp = nph * clat;
    xyz[0] = tmp * clon;
    xyz[1] = tmp * slon;
    xyz[2] = ((1.0 - e2) * n + llh[2]) * slat;
}

static void local_frame(const double *llh, double t[3][3]) /* ltcmat c:384-405 */
{
    const double slat = sin(llh[0]), clat = cos(llh[0]), slon = sin(llh[1]), clon = cos(llh[1]);
    t[0][0] = -slat * clon;
    t[0][1] = -slat * slon;
    t[0][2] = clat;
    t[1][0] = -slon;
    t[1][1] = clon;
    t[1][2] = 0.0;
    t[2][0] = clat * clon;
    t[2][1] = clat * slon;
    t[2][2] = slat;
}

static void los_to_azel(const double *los, double t[3][3], double *azel) /* ecef2neu + neu2azel c:411-434 */
{
    double neu[3];
    for (int k = 0; k < 3; k++)
        neu[k] = t[k][0] * los[0] + t[k][1] * los[1] + t[k][2] * los[2];
    azel[0] = atan2(neu[1], neu[0]);
    if (azel[0] < 0.0)
        azel[0] += (2.0 * K_PI);
    const double ne = sqrt(neu[0] * neu[0] + neu[1] * neu[1]);
    azel[1] = atan2(neu[2], ne);
}

/* ---- orbit and clock (satpos c:443-546) ---------------------------------------------------------------- */

static void sat_state(const eph_t *e, gtime_t g, double *pos, double *vel, double *clk)
{
    double tk = g.sec - e->toe.sec;
    if (tk > K_HALF_WEEK)
        tk -= K_WEEK;
    else if (tk < -K_HALF_WEEK)
        tk += K_WEEK;

    /* Kepler's equation by Newton iteration to 1e-14 (c:478-487) */
    const double mk = e->m0 + e->n * tk;
    double ek = mk, ek_old = ek + 1.0, one_m_ecos = 0;
    while (fabs(ek - ek_old) > 1.0E-14) {
        ek_old = ek;
        one_m_ecos = 1.0 - e->ecc * cos(ek_old);
        ek = ek + (mk - ek_old + e->ecc * sin(ek_old)) / one_m_ecos;
    }
    const double sek = sin(ek), cek = cos(ek);
    const double ekdot = e->n / one_m_ecos;
    const double relativistic = -4.442807633E-10 * e->ecc * e->sqrta * sek;

    const double pk = atan2(e->sq1e2 * sek, cek - e->ecc) + e->aop;
    const double pkdot = e->sq1e2 * ekdot / one_m_ecos;
    const double s2pk = sin(2.0 * pk), c2pk = cos(2.0 * pk);

    const double uk = pk + e->cus * s2pk + e->cuc * c2pk;
    const double suk = sin(uk), cuk = cos(uk);
    const double ukdot = pkdot * (1.0 + 2.0 * (e->cus * c2pk - e->cuc * s2pk));

    const double rk = e->A * one_m_ecos + e->crc * c2pk + e->crs * s2pk;
    const double rkdot = e->A * e->ecc * sek * ekdot + 2.0 * pkdot * (e->crs * c2pk - e->crc * s2pk);

    const double ik = e->inc0 + e->idot * tk + e->cic * c2pk + e->cis * s2pk;
    const double sik = sin(ik), cik = cos(ik);
    const double ikdot = e->idot + 2.0 * pkdot * (e->cis * c2pk - e->cic * s2pk);

    const double xpk = rk * cuk, ypk = rk * suk;
    const double xpkdot = rkdot * cuk - ypk * ukdot;
    const double ypkdot = rkdot * suk + xpk * ukdot;

    const double ok = e->omg0 + tk * e->omgkdot - K_OMEGA_E * e->toe.sec;
    const double sok = sin(ok), cok = cos(ok);

    pos[0] = xpk * cok - ypk * cik * sok;
    pos[1] = xpk * sok + ypk * cik * cok;
    pos[2] = ypk * sik;

    const double tmp = ypkdot * cik - ypk * sik * ikdot;
    vel[0] = -e->omgkdot * pos[1] + xpkdot * cok - tmp * sok;
    vel[1] = e->omgkdot * pos[0] + xpkdot * sok + tmp * cok;
    vel[2] = ypk * cik * ikdot + ypkdot * sik;

    /* clock (c:533-543) */
    tk = g.sec - e->toc.sec;
    if (tk > K_HALF_WEEK)
        tk -= K_WEEK;
    else if (tk < -K_HALF_WEEK)
        tk += K_WEEK;
    clk[0] = e->af0 + tk * (e->af1 + tk * e->af2) + relativistic - e->tgd;
    clk[1] = e->af1 + 2.0 * tk * e->af2;
}

/* ---- Klobuchar ionosphere (ionosphericDelay c:1612-1683) ------------------------------------------------- */

static double iono_delay_m(const iono_t *io, gtime_t g, const double *llh, const double *azel)
{
    if (!io->enable)
        return 0.0;
    const double E = azel[1] / K_PI, phi_u = llh[0] / K_PI, lam_u = llh[1] / K_PI;
    const double F = 1.0 + 16.0 * pow((0.53 - E), 3.0); /* obliquity */
    if (!io->valid)
        return F * 5.0e-9 * K_C;

    const double psi = 0.0137 / (E + 0.11) - 0.022;
    double phi_i = phi_u + psi * cos(azel[0]);
    if (phi_i > 0.416)
        phi_i = 0.416;
    else if (phi_i < -0.416)
        phi_i = -0.416;
    const double lam_i = lam_u + psi * sin(azel[0]) / cos(phi_i * K_PI);
    const double phi_m = phi_i + 0.064 * cos((lam_i - 1.617) * K_PI);
    const double phi_m2 = phi_m * phi_m, phi_m3 = phi_m2 * phi_m;

    double amp = io->alpha[0] + io->alpha[1] * phi_m + io->alpha[2] * phi_m2 + io->alpha[3] * phi_m3;
    if (amp < 0.0)
        amp = 0.0;
    double per = io->beta[0] + io->beta[1] * phi_m + io->beta[2] * phi_m2 + io->beta[3] * phi_m3;
    if (per < 72000.0)
        per = 72000.0;

    double t = K_DAY / 2.0 * lam_i + g.sec; /* local time */
    while (t >= K_DAY)
        t -= K_DAY;
    while (t < 0)
        t += K_DAY;
    const double X = 2.0 * K_PI * (t - 50400.0) / per;
    if (fabs(X) < 1.57) {
        const double X2 = X * X, X4 = X2 * X2;
        return F * (5.0e-9 + amp * (1.0 - X2 / 2.0 + X4 / 24.0)) * K_C;
    }
    return F * 5.0e-9 * K_C;
}

/* ---- pseudorange (computeRange c:1691-1747) ------------------------------------------------------------- */

static void pseudorange(range_t *rho, const eph_t *e, const iono_t *io, gtime_t g, const double *xyz)
{
    double pos[3], vel[3], clk[2], los[3], llh[3], tmat[3][3];
    sat_state(e, g, pos, vel, clk);

    for (int k = 0; k < 3; k++)
        los[k] = pos[k] - xyz[k];
    const double tau = norm3(los) / K_C; /* light time */

    for (int k = 0; k < 3; k++) /* back to the transmission time */
        pos[k] -= vel[k] * tau;

    const double xrot = pos[0] + pos[1] * K_OMEGA_E * tau; /* Earth rotation during the flight */
    const double yrot = pos[1] - pos[0] * K_OMEGA_E * tau;
    pos[0] = xrot;
    pos[1] = yrot;

    for (int k = 0; k < 3; k++)
        los[k] = pos[k] - xyz[k];
    const double range = norm3(los);
    rho->d = range;
    rho->range = range - K_C * clk[0];
    rho->rate = (vel[0] * los[0] + vel[1] * los[1] + vel[2] * los[2]) / range;
    rho->g = g;

    ecef_to_llh(xyz, llh);
    local_frame(llh, tmat);
    los_to_azel(los, tmat, rho->azel);

    rho->iono_delay = iono_delay_m(io, g, llh, rho->azel);
    rho->range += rho->iono_delay;
}

/* ---- navigation message ------------------------------------------------------------------------------------ */

static uint32_t popcnt32(uint32_t v) /* countBits c:729-744 (32-bit masks there too) */
{
    v = ((v >> 1) & 0x55555555u) + (v & 0x55555555u);
    v = ((v >> 2) & 0x33333333u) + (v & 0x33333333u);
    v = ((v >> 4) & 0x0F0F0F0Fu) + (v & 0x0F0F0F0Fu);
    v = ((v >> 8) & 0x00FF00FFu) + (v & 0x00FF00FFu);
    v = ((v >> 16) & 0x0000FFFFu) + (v & 0x0000FFFFu);
    return v;
}

/*
 * The (32,26) Hamming code of the navigation message (IS-GPS-200 20.3.5.2; computeChecksum c:751-814).  Word layout here
 * as in the reference: bits 31..30 = D29*, D30* of the previous word, 29..6 = d1..d24, 5..0 = D25..D30.  Parity bit
 * D(25+k) = (D29* or D30*) xor the parity of the data bits selected by row k of the generator below; the data bits go
 * out inverted when D30* is set.  nib: words 2 and 10 carry two non-information-bearing bits (d23, d24) that are
 * solved for so that the word's own D29 and D30 come out zero.
 */
static const struct {
    uint32_t taps; /* d1..d24 entering this parity bit, in word position */
    int prev;      /* which bit of the previous word enters: 31 = D29*, 30 = D30* */
} k_parity_row[6] = {
    {0x3B1F3480u, 31}, {0x1D8F9A40u, 30}, {0x2EC7CD00u, 31}, {0x1763E680u, 30}, {0x2BB1F340u, 30}, {0x0B7A89C0u, 31}};

static uint32_t parity_bit(uint32_t source, uint32_t d, int k)
{
    return ((source >> k_parity_row[k].prev) ^ popcnt32(k_parity_row[k].taps & d)) & 1u;
}

static uint32_t nav_parity(uint32_t source, int nib)
{
    uint32_t d = source & 0x3FFFFFC0u;
    if (nib) {
        /* d24 (bit 6) so that D29 = 0, then d23 (bit 7) so that D30 = 0 — in that order: row 5 taps d24 */
        d ^= parity_bit(source, d, 4) << 6;
        d ^= parity_bit(source, d, 5) << 7;
    }
    uint32_t word = (source >> 30) & 1u ? d ^ 0x3FFFFFC0u : d;
    for (int k = 0; k < 6; k++)
        word |= parity_bit(source, d, k) << (5 - k);
    return word & 0x3FFFFFFFu;
}

/*
 * Ephemeris + iono/UTC -> the 24 data bits of each of 5 x 10 words, left-justified in 30 (eph2sbf c:552-723).
 *
 * Two steps.  (1) The broadcast quantities as scaled integers — FLOATING POINT, so the statements keep the reference's
 * operation order (x / 2^-n / pi, truncation toward zero by the cast; only the iono/UTC terms are rounded): a different
 * order can change the last bit of a navigation word.  (2) Where each integer goes in the frame — pure bit layout, taken
 * from IS-GPS-200 figures 20-1 (subframes 1-3) and 40-1 (pages 18 and 25) as a table: {subframe, word, position of the
 * field's lowest bit in the 30-bit word, width, which quantity, how many of its low bits were sent elsewhere}.
 */
enum nav_quantity {
    Q_PREAMBLE, Q_SF_ID_1, Q_SF_ID_2, Q_SF_ID_3, Q_SF_ID_4, Q_SF_ID_5,
    Q_WN, Q_CODE_L2, Q_URA, Q_SV_HEALTH, Q_IODC, Q_TGD, Q_TOC, Q_AF2, Q_AF1, Q_AF0,
    Q_IODE, Q_CRS, Q_DELTA_N, Q_M0, Q_CUC, Q_ECC, Q_CUS, Q_SQRT_A, Q_TOE,
    Q_CIC, Q_OMEGA0, Q_CIS, Q_I0, Q_CRC, Q_OMEGA, Q_OMEGA_DOT, Q_IDOT,
    Q_DATA_ID, Q_PAGE_18, Q_PAGE_25_SF4, Q_PAGE_25_SF5,
    Q_ALPHA0, Q_ALPHA1, Q_ALPHA2, Q_ALPHA3, Q_BETA0, Q_BETA1, Q_BETA2, Q_BETA3,
    Q_A1, Q_A0, Q_TOT, Q_WNT, Q_DT_LS, Q_WN_LSF, Q_DN, Q_DT_LSF, Q_TOA, Q_WNA,
    Q_COUNT
};

typedef struct {
    uint8_t sf, word, lsb, width, what, sent_below;
} nav_field_t;

#define EVERY_SUBFRAME(sf, id) {sf, 0, 22, 8, Q_PREAMBLE, 0}, {sf, 1, 8, 3, id, 0}

static const nav_field_t k_frame_layout[] = {
    /* subframe 1: clock */
    EVERY_SUBFRAME(0, Q_SF_ID_1),
    {0, 2, 20, 10, Q_WN, 0}, {0, 2, 18, 2, Q_CODE_L2, 0}, {0, 2, 14, 4, Q_URA, 0}, {0, 2, 8, 6, Q_SV_HEALTH, 0}, {0, 2, 6, 2, Q_IODC, 8},
    {0, 6, 6, 8, Q_TGD, 0},
    {0, 7, 22, 8, Q_IODC, 0}, {0, 7, 6, 16, Q_TOC, 0},
    {0, 8, 22, 8, Q_AF2, 0}, {0, 8, 6, 16, Q_AF1, 0},
    {0, 9, 8, 22, Q_AF0, 0},
    /* subframe 2: orbit, first half */
    EVERY_SUBFRAME(1, Q_SF_ID_2),
    {1, 2, 22, 8, Q_IODE, 0}, {1, 2, 6, 16, Q_CRS, 0},
    {1, 3, 14, 16, Q_DELTA_N, 0}, {1, 3, 6, 8, Q_M0, 24},
    {1, 4, 6, 24, Q_M0, 0},
    {1, 5, 14, 16, Q_CUC, 0}, {1, 5, 6, 8, Q_ECC, 24},
    {1, 6, 6, 24, Q_ECC, 0},
    {1, 7, 14, 16, Q_CUS, 0}, {1, 7, 6, 8, Q_SQRT_A, 24},
    {1, 8, 6, 24, Q_SQRT_A, 0},
    {1, 9, 14, 16, Q_TOE, 0},
    /* subframe 3: orbit, second half */
    EVERY_SUBFRAME(2, Q_SF_ID_3),
    {2, 2, 14, 16, Q_CIC, 0}, {2, 2, 6, 8, Q_OMEGA0, 24},
    {2, 3, 6, 24, Q_OMEGA0, 0},
    {2, 4, 14, 16, Q_CIS, 0}, {2, 4, 6, 8, Q_I0, 24},
    {2, 5, 6, 24, Q_I0, 0},
    {2, 6, 14, 16, Q_CRC, 0}, {2, 6, 6, 8, Q_OMEGA, 24},
    {2, 7, 6, 24, Q_OMEGA, 0},
    {2, 8, 6, 24, Q_OMEGA_DOT, 0},
    {2, 9, 22, 8, Q_IODE, 0}, {2, 9, 8, 14, Q_IDOT, 0},
    /* subframe 5, page 25: almanac reference time and week */
    EVERY_SUBFRAME(4, Q_SF_ID_5),
    {4, 2, 28, 2, Q_DATA_ID, 0}, {4, 2, 22, 6, Q_PAGE_25_SF5, 0}, {4, 2, 14, 8, Q_TOA, 0}, {4, 2, 6, 8, Q_WNA, 0},
    /* subframe 4 */
    EVERY_SUBFRAME(3, Q_SF_ID_4),
    {3, 2, 28, 2, Q_DATA_ID, 0},
};
/* ... page 18 (ionosphere, UTC) when the file's header had both, */
static const nav_field_t k_page_18[] = {
    {3, 2, 22, 6, Q_PAGE_18, 0}, {3, 2, 14, 8, Q_ALPHA0, 0}, {3, 2, 6, 8, Q_ALPHA1, 0},
    {3, 3, 22, 8, Q_ALPHA2, 0}, {3, 3, 14, 8, Q_ALPHA3, 0}, {3, 3, 6, 8, Q_BETA0, 0},
    {3, 4, 22, 8, Q_BETA1, 0}, {3, 4, 14, 8, Q_BETA2, 0}, {3, 4, 6, 8, Q_BETA3, 0},
    {3, 5, 6, 24, Q_A1, 0},
    {3, 6, 6, 24, Q_A0, 8},
    {3, 7, 22, 8, Q_A0, 0}, {3, 7, 14, 8, Q_TOT, 0}, {3, 7, 6, 8, Q_WNT, 0},
    {3, 8, 22, 8, Q_DT_LS, 0}, {3, 8, 14, 8, Q_WN_LSF, 0}, {3, 8, 6, 8, Q_DN, 0},
    {3, 9, 22, 8, Q_DT_LSF, 0},
};
/* ... else the empty page 25 */
static const nav_field_t k_page_25_sf4[] = {{3, 2, 22, 6, Q_PAGE_25_SF4, 0}};
#undef EVERY_SUBFRAME

static void place_fields(uint32_t sbf[5][10], const nav_field_t *f, size_t n, const int64_t *q)
{
    for (size_t k = 0; k < n; k++) {
        /* two's complement, low `width` bits of what is left after the bits sent in another word (>> of a negative
         * int64_t is arithmetic with gcc, as the reference's casts assume) */
        const uint64_t bits = (uint64_t)(q[f[k].what] >> f[k].sent_below) & ((1ull << f[k].width) - 1ull);
        sbf[f[k].sf][f[k].word] |= (uint32_t)(bits << f[k].lsb);
    }
}

static void build_subframes(const eph_t *e, const iono_t *io, uint32_t sbf[5][10])
{
    int64_t q[Q_COUNT];
    /* ---- (1) scaling: the reference's statements (c:580-645), operation for operation ---- */
    q[Q_PREAMBLE] = 0x8B;
    q[Q_SF_ID_1] = 1, q[Q_SF_ID_2] = 2, q[Q_SF_ID_3] = 3, q[Q_SF_ID_4] = 4, q[Q_SF_ID_5] = 5;
    q[Q_WN] = 0; /* the transmission week is put in per frame (build_nav_words; c:1877-1878) */
    q[Q_TOE] = (int64_t)(uint64_t)(e->toe.sec / 16.0);
    q[Q_TOC] = (int64_t)(uint64_t)(e->toc.sec / 16.0);
    q[Q_IODE] = (int64_t)(uint64_t)(e->iode);
    q[Q_IODC] = (int64_t)(uint64_t)(e->iodc);
    q[Q_DELTA_N] = (int64_t)(e->deltan / P2_43 / K_PI);
    q[Q_CUC] = (int64_t)(e->cuc / P2_29);
    q[Q_CUS] = (int64_t)(e->cus / P2_29);
    q[Q_CIC] = (int64_t)(e->cic / P2_29);
    q[Q_CIS] = (int64_t)(e->cis / P2_29);
    q[Q_CRC] = (int64_t)(e->crc / P2_5);
    q[Q_CRS] = (int64_t)(e->crs / P2_5);
    q[Q_ECC] = (int64_t)(uint64_t)(e->ecc / P2_33);
    q[Q_SQRT_A] = (int64_t)(uint64_t)(e->sqrta / P2_19);
    q[Q_M0] = (int64_t)(e->m0 / P2_31 / K_PI);
    q[Q_OMEGA0] = (int64_t)(e->omg0 / P2_31 / K_PI);
    q[Q_I0] = (int64_t)(e->inc0 / P2_31 / K_PI);
    q[Q_OMEGA] = (int64_t)(e->aop / P2_31 / K_PI);
    q[Q_OMEGA_DOT] = (int64_t)(e->omgdot / P2_43 / K_PI);
    q[Q_IDOT] = (int64_t)(e->idot / P2_43 / K_PI);
    q[Q_AF0] = (int64_t)(e->af0 / P2_31);
    q[Q_AF1] = (int64_t)(e->af1 / P2_43);
    q[Q_AF2] = (int64_t)(e->af2 / P2_55);
    q[Q_TGD] = (int64_t)(e->tgd / P2_31);
    q[Q_SV_HEALTH] = (int)(uint64_t)(e->svhlth);
    q[Q_CODE_L2] = (int)(uint64_t)(e->codeL2);
    q[Q_WNA] = (int64_t)(uint64_t)(e->toe.week % 256);
    q[Q_TOA] = (int64_t)(uint64_t)(e->toe.sec / 4096.0);
    q[Q_URA] = 0;
    q[Q_DATA_ID] = 1;
    q[Q_PAGE_25_SF4] = 63, q[Q_PAGE_25_SF5] = 51, q[Q_PAGE_18] = 56; /* the SV (page) IDs of c:576-578 */
    q[Q_ALPHA0] = (int64_t)round(io->alpha[0] / P2_30);
    q[Q_ALPHA1] = (int64_t)round(io->alpha[1] / P2_27);
    q[Q_ALPHA2] = (int64_t)round(io->alpha[2] / P2_24);
    q[Q_ALPHA3] = (int64_t)round(io->alpha[3] / P2_24);
    q[Q_BETA0] = (int64_t)round(io->beta[0] / 2048.0);
    q[Q_BETA1] = (int64_t)round(io->beta[1] / 16384.0);
    q[Q_BETA2] = (int64_t)round(io->beta[2] / 65536.0);
    q[Q_BETA3] = (int64_t)round(io->beta[3] / 65536.0);
    q[Q_A0] = (int64_t)round(io->A0 / P2_30);
    q[Q_A1] = (int64_t)round(io->A1 / P2_50);
    q[Q_DT_LS] = (int64_t)(io->dtls);
    q[Q_DT_LSF] = 18; /* fixed leap-second schedule c:643-645 */
    q[Q_TOT] = (int64_t)(uint64_t)(io->tot / 4096);
    q[Q_WNT] = (int64_t)(uint64_t)(io->wnt % 256);
    q[Q_WN_LSF] = 1929 % 256;
    q[Q_DN] = 7;
    /* ---- (2) layout ---- */
    memset(sbf, 0, 50 * sizeof(uint32_t));
    place_fields(sbf, k_frame_layout, sizeof k_frame_layout / sizeof k_frame_layout[0], q);
    if (io->valid)
        place_fields(sbf, k_page_18, sizeof k_page_18 / sizeof k_page_18[0], q);
    else
        place_fields(sbf, k_page_25_sf4, sizeof k_page_25_sf4 / sizeof k_page_25_sf4[0], q);
}

/* one 30 s frame (plus the previous subframe 5 in front) with TOW, week and parity (generateNavMsg c:1820-1894) */
static void build_nav_words(gtime_t g, chan_t *c, int init)
{
    gtime_t g0;
    g0.week = g.week;
    g0.sec = (double)(((unsigned long)(g.sec + 0.5)) / 30UL) * 30.0; /* align to the frame */
    c->g0 = g0;
    const unsigned long wn = (unsigned long)(g0.week % 1024);
    unsigned long tow = ((unsigned long)g0.sec) / 6UL;
    uint32_t prev = 0;

    if (init == 1) { /* words 0-9: subframe 5 of the frame before */
        for (int k = 0; k < 10; k++) {
            uint32_t w = c->sbf[4][k];
            if (k == 1)
                w |= (uint32_t)((tow & 0x1FFFFUL) << 13);
            w |= (prev << 30) & 0xC0000000u;
            c->dwrd[k] = nav_parity(w, (k == 1) || (k == 9));
            prev = c->dwrd[k];
        }
    } else {
        for (int k = 0; k < 10; k++) {
            c->dwrd[k] = c->dwrd[50 + k];
            prev = c->dwrd[k];
        }
    }
    for (int s = 0; s < 5; s++) {
        tow++;
        for (int k = 0; k < 10; k++) {
            uint32_t w = c->sbf[s][k];
            if (s == 0 && k == 2)
                w |= (uint32_t)((wn & 0x3FFUL) << 20);
            if (k == 1)
                w |= (uint32_t)((tow & 0x1FFFFUL) << 13);
            w |= (prev << 30) & 0xC0000000u;
            c->dwrd[(s + 1) * 10 + k] = nav_parity(w, (k == 1) || (k == 9));
            prev = c->dwrd[(s + 1) * 10 + k];
        }
    }
}

/* ---- channels ------------------------------------------------------------------------------------------------ */

static int sat_visible(const eph_t *e, gtime_t g, const double *xyz, double mask_deg, double *azel) /* c:1896-1916 */
{
    if (!e->valid)
        return -1;
    double llh[3], tmat[3][3], pos[3], vel[3], clk[3], los[3];
    ecef_to_llh(xyz, llh);
    local_frame(llh, tmat);
    sat_state(e, g, pos, vel, clk);
    for (int k = 0; k < 3; k++)
        los[k] = pos[k] - xyz[k];
    los_to_azel(los, tmat, azel);
    return (azel[1] * K_R2D > mask_deg) ? 1 : 0;
}

/* allocateChannel c:1918-1989 (the elevation mask is hard-wired to 0 there) */
static int allocate_channels(gpsfe_t *fe, const eph_t *eph, gtime_t grx, const double *xyz)
{
    int nsat = 0;
    double azel[2];
    const double origin[3] = {0.0, 0.0, 0.0};
    for (int sv = 0; sv < N_SAT; sv++) {
        if (sat_visible(&eph[sv], grx, xyz, 0.0, azel) == 1) {
            nsat++;
            if (fe->sat_chan[sv] == -1) {
                int i;
                for (i = 0; i < fe->max_chan; i++) {
                    chan_t *c = &fe->chan[i];
                    if (c->prn != 0)
                        continue;
                    c->prn = sv + 1;
                    c->azel[0] = azel[0];
                    c->azel[1] = azel[1];
                    build_subframes(&eph[sv], &fe->iono, c->sbf);
                    build_nav_words(grx, c, 1);
                    range_t rho;
                    pseudorange(&rho, &eph[sv], &fe->iono, grx, xyz);
                    c->rho0 = rho;
                    /* initial carrier phase from the range to the receiver and to the geocentre (c:1956-1964) */
                    const double r_xyz = rho.range;
                    pseudorange(&rho, &eph[sv], &fe->iono, grx, origin);
                    const double r_ref = rho.range;
                    double phase_ini = (2.0 * r_ref - r_xyz) / K_LAMBDA;
                    if (!fe->fixed_carrier) {
                        c->carr_phase = phase_ini - floor(phase_ini);
                    } else { /* c:1966-1967: the accumulator's value, kept here as a double */
                        phase_ini -= floor(phase_ini);
                        c->carr_phase = (double)(unsigned int)(512.0 * 65536.0 * phase_ini);
                    }
                    break;
                }
                if (i < fe->max_chan)
                    fe->sat_chan[sv] = i;
            }
        } else if (fe->sat_chan[sv] >= 0) { /* set: free its channel */
            fe->chan[fe->sat_chan[sv]].prn = 0;
            fe->sat_chan[sv] = -1;
        }
    }
    return nsat;
}

/* computeCodePhase c:1754-1787: what a block's descriptor holds of the code and carrier NCOs, from the ranges at the
 * block's start (rho0) and end (rho1) and the frame the nav words were built for (g0).  A pure function: the blocks of a
 * span between two 30 s maintenances can be seeded in any order (gpsfe_generate does so on several threads). */
typedef struct {
    double f_carr, f_code, code_phase;
    int iword, ibit, icode;
} nco_seed_t;

static nco_seed_t seed_ncos(const range_t *rho0, const range_t *rho1, gtime_t g0, double dt)
{
    nco_seed_t s;
    const double rhorate = (rho1->range - rho0->range) / dt;
    s.f_carr = -rhorate / K_LAMBDA;
    s.f_code = 1.023e6 + s.f_carr * (1.0 / 1540.0);
    const double ms = ((gps_diff(rho0->g, g0) + 6.0) - rho0->range / K_C) * 1000.0;
    int ims = (int)ms;
    s.code_phase = (ms - (double)ims) * GPSBB_CA_LEN;
    s.iword = ims / 600;
    ims -= s.iword * 600;
    s.ibit = ims / 20;
    ims -= s.ibit * 20;
    s.icode = ims;
    return s;
}

static void seed_code_phase(chan_t *c, const range_t *rho1, double dt)
{
    const nco_seed_t s = seed_ncos(&c->rho0, rho1, c->g0, dt);
    c->f_carr = s.f_carr;
    c->f_code = s.f_code;
    c->code_phase = s.code_phase;
    c->iword = s.iword;
    c->ibit = s.ibit;
    c->icode = s.icode;
    c->rho0 = *rho1;
}

/* gain[i] of c:2677-2685 */
static double channel_gain(const gpsfe_t *fe, const range_t *rho)
{
    const double path_loss = 20200000.0 / rho->d;
    const int ibs = (int)((90.0 - rho->azel[1] * K_R2D) / 5.0); /* elevation -> boresight bin */
    return (double)(path_loss * fe->ant_pat[ibs]);
}

/* ---- RINEX-2 navigation reader (readRinex2 c:874-1233) ----------------------------------------------------- */

/* fixed-column field: copy, turn Fortran 'D' exponents into 'E' (c:821-836), atof */
static double field(const char *line, size_t len, int col, int width)
{
    char tmp[24];
    int n = 0;
    for (int k = 0; k < width && (size_t)(col + k) < len && line[col + k] != 0; k++)
        tmp[n++] = (line[col + k] == 'D' || line[col + k] == 'd') ? 'E' : line[col + k];
    tmp[n] = 0;
    return atof(tmp);
}

static int field_int(const char *line, size_t len, int col, int width)
{
    char tmp[24];
    int n = 0;
    for (int k = 0; k < width && (size_t)(col + k) < len && line[col + k] != 0; k++)
        tmp[n++] = line[col + k];
    tmp[n] = 0;
    return atoi(tmp);
}

/* the v3 reader runs the D->E replacement over this integer field before atoi (c:1345-1348) */
static long strtol_field(const char *line, size_t len, int col, int width)
{
    char tmp[24];
    int n = 0;
    for (int k = 0; k < width && (size_t)(col + k) < len && line[col + k] != 0; k++)
        tmp[n++] = (line[col + k] == 'D' || line[col + k] == 'd') ? 'E' : line[col + k];
    tmp[n] = 0;
    return atoi(tmp);
}

static int label_is(const char *line, size_t len, const char *label)
{
    const size_t l = strlen(label);
    return len >= 60 + l && strncmp(line + 60, label, l) == 0;
}

/* RINEX-2 (readRinex2 c:874-1233) and RINEX-3 (readRinex3 c:1241-1610) GPS navigation files: same
 * content, the v3 columns are shifted by one and the header carries the iono/UTC terms under other labels */
static int read_rinex(gpsfe_t *fe, const char *path, int v3)
{
    gzFile fp = gzopen(path, "rt");
    if (!fp)
        return -1;
    char line[100];
    int flags = 0;
    iono_t *io = &fe->iono;
    for (int s = 0; s <= N_EPH_SETS; s++)
        for (int sv = 0; sv < N_SAT; sv++)
            fe->eph[s][sv].valid = 0;

    /* header: labels start at column 60 (c:899-1000 / c:1266-1362) */
    while (gzgets(fp, line, sizeof line)) {
        const size_t len = strlen(line);
        if (label_is(line, len, "COMMENT"))
            continue;
        if (label_is(line, len, "END OF HEADER"))
            break;
        if (label_is(line, len, "RINEX VERSION / TYPE")) {
            const double ver = field(line, len, 0, 9);
            const int bad = v3 ? (ver < 3.0 || (line[20] != 'N' && line[40] != 'G')) : (ver > 3.0 || line[20] != 'N');
            if (bad) {
                gzclose(fp);
                return -2;
            }
        } else if (!v3 && label_is(line, len, "ION ALPHA")) {
            for (int k = 0; k < 4; k++)
                io->alpha[k] = field(line, len, 2 + 12 * k, 12);
            flags |= 1;
        } else if (!v3 && label_is(line, len, "ION BETA")) {
            for (int k = 0; k < 4; k++)
                io->beta[k] = field(line, len, 2 + 12 * k, 12);
            flags |= 2;
        } else if (!v3 && label_is(line, len, "DELTA-UTC")) {
            io->A0 = field(line, len, 3, 19);
            io->A1 = field(line, len, 22, 19);
            io->tot = field_int(line, len, 41, 9);
            io->wnt = field_int(line, len, 50, 9);
            if (io->tot % 4096 == 0)
                flags |= 4;
        } else if (v3 && label_is(line, len, "IONOSPHERIC CORR")) {
            if (strncmp(line, "GPSA", 4) == 0) {
                for (int k = 0; k < 4; k++)
                    io->alpha[k] = field(line, len, 5 + 12 * k, 12);
                flags |= 1;
            } else if (strncmp(line, "GPSB", 4) == 0) {
                for (int k = 0; k < 4; k++)
                    io->beta[k] = field(line, len, 5 + 12 * k, 12);
                flags |= 2;
            }
        } else if (v3 && label_is(line, len, "TIME SYSTEM CORR") && strncmp(line, "GPUT", 4) == 0) {
            io->A0 = field(line, len, 5, 17);
            io->A1 = field(line, len, 22, 16);
            io->tot = (int)strtol_field(line, len, 38, 7);
            io->wnt = field_int(line, len, 45, 6);
            if (io->tot % 4096 == 0)
                flags |= 4;
        } else if (label_is(line, len, "LEAP SECONDS")) {
            io->dtls = field_int(line, len, 0, 6);
            flags |= 8;
        }
    }
    io->valid = (flags == 0xF);

    /* records: 8 lines per satellite; a new set starts when TOC jumps by more than an hour (c:1046-1054) */
    const int o = v3 ? 1 : 0; /* column shift of the four 19-character fields */
    gtime_t g_set = {-1, 0.0};
    int ieph = 0;
    while (gzgets(fp, line, sizeof line)) {
        size_t len = strlen(line);
        int sv;
        caltime_t t;
        if (v3) {
            if (line[0] != 'G') /* other constellations (c:1380-1382) */
                continue;
            sv = field_int(line, len, 1, 2) - 1;
            t.y = field_int(line, len, 4, 4);
            t.m = field_int(line, len, 9, 2);
            t.d = field_int(line, len, 12, 2);
            t.hh = field_int(line, len, 15, 2);
            t.mm = field_int(line, len, 18, 2);
            t.sec = (double)field_int(line, len, 21, 2);
        } else {
            sv = field_int(line, len, 0, 2) - 1;
            t.y = field_int(line, len, 3, 2) + 2000;
            t.m = field_int(line, len, 6, 2);
            t.d = field_int(line, len, 9, 2);
            t.hh = field_int(line, len, 12, 2);
            t.mm = field_int(line, len, 15, 2);
            t.sec = field(line, len, 18, 2); /* two characters of the seconds field (c:1036-1038) */
        }
        const gtime_t g = cal_to_gps(&t);
        if (g_set.week == -1)
            g_set = g;
        if (gps_diff(g, g_set) > K_HOUR) {
            g_set = g;
            if (++ieph >= N_EPH_SETS)
                break;
        }
        if (sv < 0 || sv >= N_SAT)
            break; /* malformed record: the reference would index out of bounds here */
        eph_t *e = &fe->eph[ieph][sv];
        e->t = t;
        e->toc = g;
        e->af0 = field(line, len, 22 + o, 19);
        e->af1 = field(line, len, 41 + o, 19);
        e->af2 = field(line, len, 60 + o, 19);
#define NEXT_LINE()                                  \
    if (!gzgets(fp, line, sizeof line))              \
        break;                                       \
    len = strlen(line)
        NEXT_LINE(); /* orbit 1 */
        e->iode = (int)field(line, len, 3 + o, 19);
        e->crs = field(line, len, 22 + o, 19);
        e->deltan = field(line, len, 41 + o, 19);
        e->m0 = field(line, len, 60 + o, 19);
        NEXT_LINE(); /* orbit 2 */
        e->cuc = field(line, len, 3 + o, 19);
        e->ecc = field(line, len, 22 + o, 19);
        e->cus = field(line, len, 41 + o, 19);
        e->sqrta = field(line, len, 60 + o, 19);
        NEXT_LINE(); /* orbit 3 */
        e->toe.sec = field(line, len, 3 + o, 19);
        e->cic = field(line, len, 22 + o, 19);
        e->omg0 = field(line, len, 41 + o, 19);
        e->cis = field(line, len, 60 + o, 19);
        NEXT_LINE(); /* orbit 4 */
        e->inc0 = field(line, len, 3 + o, 19);
        e->crc = field(line, len, 22 + o, 19);
        e->aop = field(line, len, 41 + o, 19);
        e->omgdot = field(line, len, 60 + o, 19);
        NEXT_LINE(); /* orbit 5 */
        e->idot = field(line, len, 3 + o, 19);
        e->codeL2 = (int)field(line, len, 22 + o, 19);
        e->toe.week = (int)field(line, len, 41 + o, 19);
        NEXT_LINE(); /* orbit 6 */
        e->svhlth = (int)field(line, len, 22 + o, 19);
        if (e->svhlth > 0 && e->svhlth < 32)
            e->svhlth += 32;
        e->tgd = field(line, len, 41 + o, 19);
        e->iodc = (int)field(line, len, 60 + o, 19);
        NEXT_LINE(); /* orbit 7: not used */
#undef NEXT_LINE
        e->valid = 1;
        e->A = e->sqrta * e->sqrta;
        e->n = sqrt(K_GM / (e->A * e->A * e->A)) + e->deltan;
        e->sq1e2 = sqrt(1.0 - e->ecc * e->ecc);
        e->omgkdot = e->omgdot - K_OMEGA_E;
    }
    gzclose(fp);
    if (g_set.week >= 0)
        ieph += 1;
    return ieph;
}

/* user motion "t,x,y,z" per line (readUserMotion c:1794-1818) */
static int read_motion(gpsfe_t *fe, const char *path)
{
    FILE *fp = fopen(path, "rt");
    if (!fp)
        return -1;
    char line[100];
    int n;
    for (n = 0; n < N_MOTION; n++) {
        double t, x, y, z;
        if (!fgets(line, sizeof line, fp))
            break;
        if (EOF == sscanf(line, "%lf,%lf,%lf,%lf", &t, &x, &y, &z))
            break;
        fe->xyz[n][0] = x;
        fe->xyz[n][1] = y;
        fe->xyz[n][2] = z;
    }
    fclose(fp);
    return n;
}

/* ---- public API ------------------------------------------------------------------------------------------------ */

static void fe_pool_stop(struct fe_pool *p);

const char *gpsfe_strerror(int err)
{
    switch (err) {
    case GPSFE_OK: return "ok";
    case GPSFE_E_BADARG: return "bad argument";
    case GPSFE_E_NAVFILE: return "cannot read the RINEX-2 navigation file";
    case GPSFE_E_MOTION: return "cannot read the user-motion file";
    case GPSFE_E_TIME: return "start time outside the ephemeris window";
    case GPSFE_E_NOEPH: return "no current set of ephemerides";
    case GPSFE_E_NOMEM: return "out of memory";
    default: return "unknown error";
    }
}

void gpsfe_close(gpsfe_t *fe)
{
    if (!fe)
        return;
    fe_pool_stop(fe->pool);
    free(fe->span_grx);
    free(fe->span_ipos);
    free(fe->span_rho);
    free(fe->xyz);
    free(fe);
}

int gpsfe_max_chan(const gpsfe_t *fe) { return fe ? fe->max_chan : 0; }

int gpsfe_open(const gpsfe_config_t *cfg, gpsfe_t **out)
{
    if (!cfg || !out || !cfg->navfile || cfg->max_chan < 1 || cfg->max_chan > GPSBB_MAX_CHAN)
        return GPSFE_E_BADARG;
    *out = NULL;
    gpsfe_t *fe = calloc(1, sizeof *fe);
    if (!fe)
        return GPSFE_E_NOMEM;
    fe->xyz = calloc(N_MOTION, sizeof *fe->xyz);
    if (!fe->xyz) {
        free(fe);
        return GPSFE_E_NOMEM;
    }
    fe->max_chan = cfg->max_chan;
    fe->fixed_carrier = cfg->fixed_carrier;
    fe->iono.enable = !cfg->iono_disable;

    /* receiver position (c:2312-2322, 2403-2415) */
    fe->static_mode = cfg->motion_file == NULL;
    if (!fe->static_mode) {
        fe->numd = read_motion(fe, cfg->motion_file);
        if (fe->numd <= 0) {
            gpsfe_close(fe);
            return GPSFE_E_MOTION;
        }
    } else if (cfg->use_ecef) {
        memcpy(fe->xyz[0], cfg->pos, sizeof fe->xyz[0]);
    } else {
        double llh[3] = {cfg->pos[0] / K_R2D, cfg->pos[1] / K_R2D, cfg->pos[2]};
        llh_to_ecef(llh, fe->xyz[0]);
    }

    fe->neph = read_rinex(fe, cfg->navfile, cfg->rinex3);
    if (fe->neph <= 0) {
        gpsfe_close(fe);
        return GPSFE_E_NAVFILE;
    }

    /* scenario start time within the ephemeris window (c:2497-2569) */
    gtime_t g0 = {-1, 0.0}, gmin = {0, 0.0}, gmax = {0, 0.0};
    if (cfg->have_start) {
        caltime_t t0 = {cfg->y, cfg->m, cfg->d, cfg->hh, cfg->mm, floor(cfg->sec)};
        if (t0.y <= 1980 || t0.m < 1 || t0.m > 12 || t0.d < 1 || t0.d > 31 || t0.hh < 0 || t0.hh > 23 ||
            t0.mm < 0 || t0.mm > 59 || cfg->sec < 0.0 || cfg->sec >= 60.0) {
            gpsfe_close(fe);
            return GPSFE_E_BADARG;
        }
        g0 = cal_to_gps(&t0);
    }
    for (int sv = 0; sv < N_SAT; sv++)
        if (fe->eph[0][sv].valid) {
            gmin = fe->eph[0][sv].toc;
            break;
        }
    for (int sv = 0; sv < N_SAT; sv++)
        if (fe->eph[fe->neph - 1][sv].valid) {
            gmax = fe->eph[fe->neph - 1][sv].toc;
            break;
        }
    if (g0.week >= 0) {
        if (cfg->time_overwrite) { /* -T: shift every TOC/TOE so the file covers the requested start (c:2523-2553) */
            gtime_t gt;
            gt.week = g0.week;
            gt.sec = (double)(((int)(g0.sec)) / 7200) * 7200.0;
            const double dsec = gps_diff(gt, gmin);
            fe->iono.wnt = gt.week;
            fe->iono.tot = (int)gt.sec;
            for (int sv = 0; sv < N_SAT; sv++)
                for (int i = 0; i < fe->neph; i++) {
                    eph_t *e = &fe->eph[i][sv];
                    if (!e->valid)
                        continue;
                    e->toc = gps_add(e->toc, dsec);
                    e->t = gps_to_cal(&e->toc);
                    e->toe = gps_add(e->toe, dsec);
                }
        } else if (gps_diff(g0, gmin) < 0.0 || gps_diff(gmax, g0) < 0.0) {
            gpsfe_close(fe);
            return GPSFE_E_TIME;
        }
    } else {
        g0 = gmin;
    }

    /* current ephemeris set: first one with a TOC within an hour of the start (c:2576-2597) */
    fe->ieph = -1;
    for (int i = 0; i < fe->neph && fe->ieph < 0; i++)
        for (int sv = 0; sv < N_SAT; sv++)
            if (fe->eph[i][sv].valid) {
                const double dt = gps_diff(g0, fe->eph[i][sv].toc);
                if (dt >= -K_HOUR && dt < K_HOUR) {
                    fe->ieph = i;
                    break;
                }
            }
    if (fe->ieph == -1) {
        gpsfe_close(fe);
        return GPSFE_E_NOEPH;
    }

    /* channels (c:2620-2632), antenna pattern (c:2645-2646), first block time (c:2653) */
    for (int i = 0; i < GPSBB_MAX_CHAN; i++)
        fe->chan[i].prn = 0;
    for (int sv = 0; sv < N_SAT; sv++)
        fe->sat_chan[sv] = -1;
    fe->grx = gps_add(g0, 0.0);
    allocate_channels(fe, fe->eph[fe->ieph], fe->grx, fe->xyz[0]);
    for (int i = 0; i < 37; i++)
        fe->ant_pat[i] = pow(10.0, -k_ant_pat_db[i] / 20.0);
    fe->grx = gps_add(fe->grx, 0.1);
    *out = fe;
    return GPSFE_OK;
}

/* every 30 s: next nav frame, ephemeris roll-over, channel re-allocation (c:2764-2798); then on to the next block's time
 * and position (c:2800-2805) */
static void end_of_block(gpsfe_t *fe, const double *xyz)
{
    const int igrx = (int)(fe->grx.sec * 10.0 + 0.5);
    if (igrx % 300 == 0) {
        for (int i = 0; i < fe->max_chan; i++)
            if (fe->chan[i].prn > 0)
                build_nav_words(fe->grx, &fe->chan[i], 0);
        for (int sv = 0; sv < N_SAT; sv++)
            if (fe->eph[fe->ieph + 1][sv].valid) {
                if (gps_diff(fe->eph[fe->ieph + 1][sv].toc, fe->grx) < K_HOUR) {
                    fe->ieph++;
                    for (int i = 0; i < fe->max_chan; i++)
                        if (fe->chan[i].prn != 0)
                            build_subframes(&fe->eph[fe->ieph][fe->chan[i].prn - 1], &fe->iono, fe->chan[i].sbf);
                }
                break; /* only the first valid satellite of the next set is looked at */
            }
        allocate_channels(fe, fe->eph[fe->ieph], fe->grx, xyz);
    }
    fe->grx = gps_add(fe->grx, 0.1); /* c:2800 */
    if (++fe->iumd >= fe->numd)      /* c:2802-2805 */
        fe->iumd = 0;
}

static void fill_descriptor(gpsbb_chan_t *d, const chan_t *c, const nco_seed_t *s, double gain)
{
    d->prn = c->prn;
    d->iword = s->iword;
    d->ibit = s->ibit;
    d->icode = s->icode;
    d->f_carr = s->f_carr;
    d->f_code = s->f_code;
    d->carr_phase = c->carr_phase;
    d->code_phase = s->code_phase;
    d->gain = gain;
    memcpy(d->dwrd, c->dwrd, sizeof d->dwrd);
}

int gpsfe_next_block(gpsfe_t *fe, gpsbb_chan_t *ch)
{
    if (!fe || !ch)
        return GPSFE_E_BADARG;
    const double *xyz = fe->static_mode ? fe->xyz[0] : fe->xyz[fe->iumd];

    /* refresh code phase, counters, frequencies and gain of every allocated channel (c:2656-2687) */
    for (int i = 0; i < fe->max_chan; i++) {
        chan_t *c = &fe->chan[i];
        gpsbb_chan_t *d = &ch[i];
        memset(d, 0, sizeof *d);
        fe->emitted_prn[i] = c->prn > 0 ? c->prn : 0;
        if (c->prn <= 0)
            continue;
        range_t rho;
        pseudorange(&rho, &fe->eph[fe->ieph][c->prn - 1], &fe->iono, fe->grx, xyz);
        c->azel[0] = rho.azel[0];
        c->azel[1] = rho.azel[1];
        seed_code_phase(c, &rho, 0.1);
        c->gain = channel_gain(fe, &rho);
        const nco_seed_t s = {c->f_carr, c->f_code, c->code_phase, c->iword, c->ibit, c->icode};
        fill_descriptor(d, c, &s, c->gain);
    }
    end_of_block(fe, xyz);
    return GPSFE_OK;
}

int gpsfe_feed_back(gpsfe_t *fe, const gpsbb_chan_state_t *end_state)
{
    if (!fe || !end_state)
        return GPSFE_E_BADARG;
    /* Only a channel that still carries the satellite it carried in the block just rendered keeps that block's end
     * phase.  gpsfe_next_block has already run the block's 30 s maintenance (the reference runs it after the fill,
     * c:2764-2798): a slot freed and given to a newly risen satellite in the same pass holds allocateChannel's
     * phase (c:1956-1964) and must not inherit the departed satellite's. */
    for (int i = 0; i < fe->max_chan; i++)
        if (fe->chan[i].prn > 0 && fe->chan[i].prn == fe->emitted_prn[i] && end_state[i].dataBit != 0)
            fe->chan[i].carr_phase = end_state[i].carr_phase;
    return GPSFE_OK;
}

/* ---- gpsfe_generate on several threads ------------------------------------------------------------------------
 * Between two 30 s maintenances (c:2764-2798) nothing of the scenario changes but the time: channel allocation,
 * ephemeris set, nav words and the frame time g0 stand still, and a block's descriptor is a pure function of the
 * ranges at its start and its end.  So a span of up to 300 blocks is done in two parallel sweeps — (1) the range to
 * every allocated satellite at every block time (orbit, clock, Klobuchar: where the time goes), (2) each block's
 * descriptor from range k-1 and range k — and the state the sequential loop would have left is put back before the
 * maintenance of the span's last block runs.  Same arithmetic per value, so the same bits (tests/test_frontend.py). */
#define FE_MAX_THREADS 32
#define FE_SPAN_MAX 512

typedef struct fe_pool {
    int n; /* parties: n - 1 workers and the caller */
    pthread_t th[FE_MAX_THREADS];
    int tid_of[FE_MAX_THREADS];
    pthread_mutex_t m;
    pthread_cond_t cv;
    pthread_barrier_t bar;
    unsigned long gen;
    int quit;
    void (*fn)(void *arg, int tid, int n, pthread_barrier_t *bar);
    void *arg;
    struct fe_pool_arg { struct fe_pool *p; int tid; } args[FE_MAX_THREADS];
} fe_pool_t;

static void *fe_worker(void *v)
{
    struct fe_pool_arg *a = v;
    fe_pool_t *p = a->p;
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->m);
        while (p->gen == seen && !p->quit)
            pthread_cond_wait(&p->cv, &p->m);
        if (p->quit) {
            pthread_mutex_unlock(&p->m);
            return NULL;
        }
        seen = p->gen;
        pthread_mutex_unlock(&p->m);
        p->fn(p->arg, a->tid, p->n, &p->bar);
        pthread_barrier_wait(&p->bar);
    }
}

static fe_pool_t *fe_pool_start(int n)
{
    fe_pool_t *p = calloc(1, sizeof *p);
    if (!p)
        return NULL;
    /* (a pool that cannot be set up is no pool: the caller then generates block by block on its own thread) */
    if (pthread_mutex_init(&p->m, NULL) != 0) {
        free(p);
        return NULL;
    }
    if (pthread_cond_init(&p->cv, NULL) != 0) {
        pthread_mutex_destroy(&p->m);
        free(p);
        return NULL;
    }
    int started = 1;
    for (int t = 1; t < n; t++) {
        p->args[t].p = p;
        p->args[t].tid = started;
        if (pthread_create(&p->th[started], NULL, fe_worker, &p->args[t]) != 0)
            break; /* fewer threads than asked for: still correct */
        started++;
    }
    p->n = started;
    if (pthread_barrier_init(&p->bar, NULL, (unsigned)p->n) != 0) {
        /* the workers wait on the condition variable, none has touched the barrier: send them home */
        pthread_mutex_lock(&p->m);
        p->quit = 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->m);
        for (int t = 1; t < p->n; t++)
            pthread_join(p->th[t], NULL);
        pthread_mutex_destroy(&p->m);
        pthread_cond_destroy(&p->cv);
        free(p);
        return NULL;
    }
    return p;
}

static void fe_pool_run(fe_pool_t *p, void (*fn)(void *, int, int, pthread_barrier_t *), void *arg)
{
    pthread_mutex_lock(&p->m);
    p->fn = fn;
    p->arg = arg;
    p->gen++;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->m);
    fn(arg, 0, p->n, &p->bar);
    pthread_barrier_wait(&p->bar);
}

static void fe_pool_stop(fe_pool_t *p)
{
    if (!p)
        return;
    pthread_mutex_lock(&p->m);
    p->quit = 1;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->m);
    for (int t = 1; t < p->n; t++)
        pthread_join(p->th[t], NULL);
    pthread_barrier_destroy(&p->bar);
    pthread_mutex_destroy(&p->m);
    pthread_cond_destroy(&p->cv);
    free(p);
}

typedef struct {
    const gpsfe_t *fe;
    int n;                    /* blocks in the span */
    const gtime_t *grx;       /* [n] block times */
    const int *ipos;          /* [n] index into fe->xyz */
    range_t *rho;             /* [n][max_chan] */
    gpsbb_chan_t *ch;         /* [n][max_chan] out */
    nco_seed_t last[GPSBB_MAX_CHAN]; /* what the last block left in chan[] */
    double last_gain[GPSBB_MAX_CHAN];
} span_job_t;

static void span_work(void *v, int tid, int nthr, pthread_barrier_t *bar)
{
    span_job_t *j = v;
    const gpsfe_t *fe = j->fe;
    const int mc = fe->max_chan;
    const int k0 = (int)((long)j->n * tid / nthr), k1 = (int)((long)j->n * (tid + 1) / nthr);
    for (int k = k0; k < k1; k++)
        for (int i = 0; i < mc; i++) {
            const chan_t *c = &fe->chan[i];
            if (c->prn > 0)
                pseudorange(&j->rho[(size_t)k * mc + i], &fe->eph[fe->ieph][c->prn - 1], &fe->iono, j->grx[k], fe->xyz[j->ipos[k]]);
        }
    pthread_barrier_wait(bar); /* block k needs the range of block k - 1, which another thread may have computed */
    for (int k = k0; k < k1; k++)
        for (int i = 0; i < mc; i++) {
            const chan_t *c = &fe->chan[i];
            gpsbb_chan_t *d = &j->ch[(size_t)k * mc + i];
            memset(d, 0, sizeof *d);
            if (c->prn <= 0)
                continue;
            const range_t *r1 = &j->rho[(size_t)k * mc + i];
            const range_t *r0 = k ? &j->rho[(size_t)(k - 1) * mc + i] : &c->rho0;
            const nco_seed_t s = seed_ncos(r0, r1, c->g0, 0.1);
            const double gain = channel_gain(fe, r1);
            fill_descriptor(d, c, &s, gain);
            if (k == j->n - 1) {
                j->last[i] = s;
                j->last_gain[i] = gain;
            }
        }
}

int gpsfe_set_threads(gpsfe_t *fe, int nthreads)
{
    if (!fe || nthreads < 0 || nthreads > FE_MAX_THREADS)
        return GPSFE_E_BADARG;
    if (fe->pool && fe->pool->n != nthreads) {
        fe_pool_stop(fe->pool);
        fe->pool = NULL;
    }
    fe->nthreads = nthreads;
    return GPSFE_OK;
}

int gpsfe_generate(gpsfe_t *fe, int nblocks, gpsbb_chan_t *ch)
{
    if (!fe || !ch || nblocks < 0)
        return GPSFE_E_BADARG;
    int nthr = fe->nthreads;
    if (nthr == 0) { /* default: the machine's cores, up to 16 */
        const long on = sysconf(_SC_NPROCESSORS_ONLN);
        nthr = on < 1 ? 1 : (on > 16 ? 16 : (int)on);
    }
    /* the span's scratch stays with the front end between calls (freed by gpsfe_destroy): a producer that generates one push
     * ahead of a ring calls this thousands of times */
    if (nthr > 1 && nblocks >= 64) {
        if (!fe->span_grx)
            fe->span_grx = malloc(FE_SPAN_MAX * sizeof(gtime_t));
        if (!fe->span_ipos)
            fe->span_ipos = malloc(FE_SPAN_MAX * sizeof(int));
        if (!fe->span_rho)
            fe->span_rho = malloc((size_t)FE_SPAN_MAX * GPSBB_MAX_CHAN * sizeof(range_t));
        if (!fe->pool)
            fe->pool = fe_pool_start(nthr);
    }
    gtime_t *grx = fe->span_grx;
    int *ipos = fe->span_ipos;
    range_t *rho = fe->span_rho;
    const int parallel = nthr > 1 && nblocks >= 64 && grx && ipos && rho && fe->pool && fe->pool->n > 1;
    int b = 0;
    while (b < nblocks) {
        if (!parallel) {
            const int rc = gpsfe_next_block(fe, ch + (size_t)b * fe->max_chan);
            if (rc != GPSFE_OK)
                return rc;
            b++;
            continue;
        }
        /* the span: up to and including the block whose end runs the 30 s maintenance */
        int n = 0, iumd = fe->iumd;
        gtime_t g = fe->grx;
        while (b + n < nblocks && n < FE_SPAN_MAX) {
            grx[n] = g;
            ipos[n] = iumd; /* (0 throughout for a static position: numd is 0 there) */
            n++;
            const int igrx = (int)(g.sec * 10.0 + 0.5);
            if (igrx % 300 == 0)
                break;
            g = gps_add(g, 0.1);
            if (++iumd >= fe->numd)
                iumd = 0;
        }
        span_job_t job;
        memset(&job, 0, sizeof job);
        job.fe = fe;
        job.n = n;
        job.grx = grx;
        job.ipos = ipos;
        job.rho = rho;
        job.ch = ch + (size_t)b * fe->max_chan;
        fe_pool_run(fe->pool, span_work, &job);
        /* the state the sequential loop would be in before the maintenance of the span's last block */
        for (int i = 0; i < fe->max_chan; i++) {
            chan_t *c = &fe->chan[i];
            fe->emitted_prn[i] = c->prn > 0 ? c->prn : 0;
            if (c->prn <= 0)
                continue;
            const range_t *r = &rho[(size_t)(n - 1) * fe->max_chan + i];
            c->azel[0] = r->azel[0];
            c->azel[1] = r->azel[1];
            c->f_carr = job.last[i].f_carr;
            c->f_code = job.last[i].f_code;
            c->code_phase = job.last[i].code_phase;
            c->iword = job.last[i].iword;
            c->ibit = job.last[i].ibit;
            c->icode = job.last[i].icode;
            c->rho0 = *r;
            c->gain = job.last_gain[i];
        }
        fe->grx = grx[n - 1];
        fe->iumd = ipos[n - 1];
        end_of_block(fe, fe->xyz[ipos[n - 1]]);
        b += n;
    }
    return GPSFE_OK;
}

int gpsfe_time(const gpsfe_t *fe, int *week, double *sec)
{
    if (!fe)
        return GPSFE_E_BADARG;
    if (week)
        *week = fe->grx.week;
    if (sec)
        *sec = fe->grx.sec;
    return GPSFE_OK;
}

int gpsfe_channel_info(const gpsfe_t *fe, int i, int *prn, double *az_deg, double *el_deg, double *range_m,
                       double *iono_m)
{
    if (!fe || i < 0 || i >= fe->max_chan)
        return GPSFE_E_BADARG;
    const chan_t *c = &fe->chan[i];
    if (prn) *prn = c->prn;
    if (az_deg) *az_deg = c->azel[0] * K_R2D;
    if (el_deg) *el_deg = c->azel[1] * K_R2D;
    if (range_m) *range_m = c->rho0.d;
    if (iono_m) *iono_m = c->rho0.iono_delay;
    return GPSFE_OK;
}
