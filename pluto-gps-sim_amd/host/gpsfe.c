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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

/* GPS (32,26) Hamming parity; nib = word carries the two non-information-bearing bits (words 2 and 10)
 * that are solved so that D29 = D30 = 0 (computeChecksum c:751-814) */
static uint32_t nav_parity(uint32_t source, int nib)
{
    static const uint32_t mask[6] = {0x3B1F3480u, 0x1D8F9A40u, 0x2EC7CD00u, 0x1763E680u, 0x2BB1F340u, 0x0B7A89C0u};
    uint32_t d = source & 0x3FFFFFC0u;
    const uint32_t D29 = (source >> 31) & 1u, D30 = (source >> 30) & 1u;
    if (nib) {
        if ((D30 + popcnt32(mask[4] & d)) % 2)
            d ^= (1u << 6);
        if ((D29 + popcnt32(mask[5] & d)) % 2)
            d ^= (1u << 7);
    }
    uint32_t D = d;
    if (D30)
        D ^= 0x3FFFFFC0u;
    D |= ((D29 + popcnt32(mask[0] & d)) % 2) << 5;
    D |= ((D30 + popcnt32(mask[1] & d)) % 2) << 4;
    D |= ((D29 + popcnt32(mask[2] & d)) % 2) << 3;
    D |= ((D30 + popcnt32(mask[3] & d)) % 2) << 2;
    D |= ((D30 + popcnt32(mask[4] & d)) % 2) << 1;
    D |= ((D29 + popcnt32(mask[5] & d)) % 2);
    return D & 0x3FFFFFFFu;
}

/* ephemeris + iono/UTC -> the 24 data bits of each of 5 x 10 words, left-justified in 30 (eph2sbf c:552-723).
 * The scaled integers are truncated toward zero (casts to long), only the iono/UTC ones are rounded. */
static void build_subframes(const eph_t *e, const iono_t *io, uint32_t sbf[5][10])
{
    const uint64_t wn = 0; /* transmission week is inserted per frame (c:1877-1878) */
    const uint64_t toe = (uint64_t)(e->toe.sec / 16.0), toc = (uint64_t)(e->toc.sec / 16.0);
    const uint64_t iode = (uint64_t)(e->iode), iodc = (uint64_t)(e->iodc);
    const int64_t deltan = (int64_t)(e->deltan / P2_43 / K_PI);
    const int64_t cuc = (int64_t)(e->cuc / P2_29), cus = (int64_t)(e->cus / P2_29);
    const int64_t cic = (int64_t)(e->cic / P2_29), cis = (int64_t)(e->cis / P2_29);
    const int64_t crc = (int64_t)(e->crc / P2_5), crs = (int64_t)(e->crs / P2_5);
    const uint64_t ecc = (uint64_t)(e->ecc / P2_33), sqrta = (uint64_t)(e->sqrta / P2_19);
    const int64_t m0 = (int64_t)(e->m0 / P2_31 / K_PI), omg0 = (int64_t)(e->omg0 / P2_31 / K_PI);
    const int64_t inc0 = (int64_t)(e->inc0 / P2_31 / K_PI), aop = (int64_t)(e->aop / P2_31 / K_PI);
    const int64_t omgdot = (int64_t)(e->omgdot / P2_43 / K_PI), idot = (int64_t)(e->idot / P2_43 / K_PI);
    const int64_t af0 = (int64_t)(e->af0 / P2_31), af1 = (int64_t)(e->af1 / P2_43), af2 = (int64_t)(e->af2 / P2_55);
    const int64_t tgd = (int64_t)(e->tgd / P2_31);
    const int svhlth = (int)(uint64_t)(e->svhlth), codeL2 = (int)(uint64_t)(e->codeL2);
    const uint64_t wna = (uint64_t)(e->toe.week % 256), toa = (uint64_t)(e->toe.sec / 4096.0);
    const uint64_t ura = 0, data_id = 1, sv_p25_sf4 = 63, sv_p25_sf5 = 51, sv_p18 = 56;

    const int64_t alpha0 = (int64_t)round(io->alpha[0] / P2_30), alpha1 = (int64_t)round(io->alpha[1] / P2_27);
    const int64_t alpha2 = (int64_t)round(io->alpha[2] / P2_24), alpha3 = (int64_t)round(io->alpha[3] / P2_24);
    const int64_t beta0 = (int64_t)round(io->beta[0] / 2048.0), beta1 = (int64_t)round(io->beta[1] / 16384.0);
    const int64_t beta2 = (int64_t)round(io->beta[2] / 65536.0), beta3 = (int64_t)round(io->beta[3] / 65536.0);
    const int64_t A0 = (int64_t)round(io->A0 / P2_30), A1 = (int64_t)round(io->A1 / P2_50);
    const int64_t dtls = (int64_t)(io->dtls), dtlsf = 18; /* fixed leap-second schedule c:643-645 */
    const uint64_t tot = (uint64_t)(io->tot / 4096), wnt = (uint64_t)(io->wnt % 256);
    const uint64_t wnlsf = 1929 % 256, dn = 7;

    const uint64_t preamble = 0x8B0000ull << 6;
#define W(s, k, v) sbf[s][k] = (uint32_t)(v)
    /* subframe 1 */
    W(0, 0, preamble);
    W(0, 1, 0x1ull << 8);
    W(0, 2, ((wn & 0x3FFull) << 20) | (((uint64_t)codeL2 & 0x3ull) << 18) | ((ura & 0xFull) << 14) |
                (((uint64_t)svhlth & 0x3Full) << 8) | (((iodc >> 8) & 0x3ull) << 6));
    W(0, 3, 0);
    W(0, 4, 0);
    W(0, 5, 0);
    W(0, 6, ((uint64_t)tgd & 0xFFull) << 6);
    W(0, 7, ((iodc & 0xFFull) << 22) | ((toc & 0xFFFFull) << 6));
    W(0, 8, (((uint64_t)af2 & 0xFFull) << 22) | (((uint64_t)af1 & 0xFFFFull) << 6));
    W(0, 9, ((uint64_t)af0 & 0x3FFFFFull) << 8);
    /* subframe 2 */
    W(1, 0, preamble);
    W(1, 1, 0x2ull << 8);
    W(1, 2, ((iode & 0xFFull) << 22) | (((uint64_t)crs & 0xFFFFull) << 6));
    W(1, 3, (((uint64_t)deltan & 0xFFFFull) << 14) | ((((uint64_t)(m0 >> 24)) & 0xFFull) << 6));
    W(1, 4, ((uint64_t)m0 & 0xFFFFFFull) << 6);
    W(1, 5, (((uint64_t)cuc & 0xFFFFull) << 14) | (((ecc >> 24) & 0xFFull) << 6));
    W(1, 6, (ecc & 0xFFFFFFull) << 6);
    W(1, 7, (((uint64_t)cus & 0xFFFFull) << 14) | (((sqrta >> 24) & 0xFFull) << 6));
    W(1, 8, (sqrta & 0xFFFFFFull) << 6);
    W(1, 9, (toe & 0xFFFFull) << 14);
    /* subframe 3 */
    W(2, 0, preamble);
    W(2, 1, 0x3ull << 8);
    W(2, 2, (((uint64_t)cic & 0xFFFFull) << 14) | ((((uint64_t)(omg0 >> 24)) & 0xFFull) << 6));
    W(2, 3, ((uint64_t)omg0 & 0xFFFFFFull) << 6);
    W(2, 4, (((uint64_t)cis & 0xFFFFull) << 14) | ((((uint64_t)(inc0 >> 24)) & 0xFFull) << 6));
    W(2, 5, ((uint64_t)inc0 & 0xFFFFFFull) << 6);
    W(2, 6, (((uint64_t)crc & 0xFFFFull) << 14) | ((((uint64_t)(aop >> 24)) & 0xFFull) << 6));
    W(2, 7, ((uint64_t)aop & 0xFFFFFFull) << 6);
    W(2, 8, ((uint64_t)omgdot & 0xFFFFFFull) << 6);
    W(2, 9, ((iode & 0xFFull) << 22) | (((uint64_t)idot & 0x3FFFull) << 8));
    /* subframe 4: page 18 (iono/UTC) when the header had them, else page 25 */
    W(3, 0, preamble);
    W(3, 1, 0x4ull << 8);
    if (io->valid) {
        W(3, 2, (data_id << 28) | (sv_p18 << 22) | (((uint64_t)alpha0 & 0xFFull) << 14) | (((uint64_t)alpha1 & 0xFFull) << 6));
        W(3, 3, (((uint64_t)alpha2 & 0xFFull) << 22) | (((uint64_t)alpha3 & 0xFFull) << 14) | (((uint64_t)beta0 & 0xFFull) << 6));
        W(3, 4, (((uint64_t)beta1 & 0xFFull) << 22) | (((uint64_t)beta2 & 0xFFull) << 14) | (((uint64_t)beta3 & 0xFFull) << 6));
        W(3, 5, ((uint64_t)A1 & 0xFFFFFFull) << 6);
        W(3, 6, (((uint64_t)(A0 >> 8)) & 0xFFFFFFull) << 6);
        W(3, 7, (((uint64_t)A0 & 0xFFull) << 22) | ((tot & 0xFFull) << 14) | ((wnt & 0xFFull) << 6));
        W(3, 8, (((uint64_t)dtls & 0xFFull) << 22) | ((wnlsf & 0xFFull) << 14) | ((dn & 0xFFull) << 6));
        W(3, 9, ((uint64_t)dtlsf & 0xFFull) << 22);
    } else {
        W(3, 2, (data_id << 28) | (sv_p25_sf4 << 22));
        for (int k = 3; k < 10; k++)
            W(3, k, 0);
    }
    /* subframe 5, page 25 */
    W(4, 0, preamble);
    W(4, 1, 0x5ull << 8);
    W(4, 2, (data_id << 28) | (sv_p25_sf5 << 22) | ((toa & 0xFFull) << 14) | ((wna & 0xFFull) << 6));
    for (int k = 3; k < 10; k++)
        W(4, k, 0);
#undef W
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

/* computeCodePhase c:1754-1787 */
static void seed_code_phase(chan_t *c, const range_t *rho1, double dt)
{
    const double rhorate = (rho1->range - c->rho0.range) / dt;
    c->f_carr = -rhorate / K_LAMBDA;
    c->f_code = 1.023e6 + c->f_carr * (1.0 / 1540.0);
    const double ms = ((gps_diff(c->rho0.g, c->g0) + 6.0) - c->rho0.range / K_C) * 1000.0;
    int ims = (int)ms;
    c->code_phase = (ms - (double)ims) * GPSBB_CA_LEN;
    c->iword = ims / 600;
    ims -= c->iword * 600;
    c->ibit = ims / 20;
    ims -= c->ibit * 20;
    c->icode = ims;
    c->rho0 = *rho1;
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
        const double path_loss = 20200000.0 / rho.d;
        const int ibs = (int)((90.0 - rho.azel[1] * K_R2D) / 5.0); /* elevation -> boresight bin */
        c->gain = (double)(path_loss * fe->ant_pat[ibs]);

        d->prn = c->prn;
        d->iword = c->iword;
        d->ibit = c->ibit;
        d->icode = c->icode;
        d->f_carr = c->f_carr;
        d->f_code = c->f_code;
        d->carr_phase = c->carr_phase;
        d->code_phase = c->code_phase;
        d->gain = c->gain;
        memcpy(d->dwrd, c->dwrd, sizeof d->dwrd);
    }

    /* every 30 s: next nav frame, ephemeris roll-over, channel re-allocation (c:2764-2798) */
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

int gpsfe_generate(gpsfe_t *fe, int nblocks, gpsbb_chan_t *ch)
{
    if (!fe || !ch || nblocks < 0)
        return GPSFE_E_BADARG;
    for (int b = 0; b < nblocks; b++) {
        int rc = gpsfe_next_block(fe, ch + (size_t)b * fe->max_chan);
        if (rc != GPSFE_OK)
            return rc;
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
