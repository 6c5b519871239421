/*
 * gpsbb_oracle.c — CPU restatement of the reference's IQ fill loop.  TEST INFRASTRUCTURE ONLY
 * (see gpsbb_oracle.h for who may use it and how it is pinned to the reference).
 *
 * Build with -std=c11 -ffp-contract=off (never -ffast-math): the reference is compiled -std=c11, i.e.
 * every double product/sum below is rounded on its own, no FMA (Makefile:1-2 of the reference).
 */
#include "gpsbb_oracle.h"

#include <math.h>
#include <string.h>

/* ---- tables (plutogpssim.c:93-161) ------------------------------------------------------------- */

void gpsbb_oracle_tables(int sin512[512], int cos512[512])
{
    /* The reference's arrays are literal ints; their closed form is trunc(511*f(2*pi*i/512)+1.0)
     * evaluated in double (which also reproduces cos[384]==0: cos(3*pi/2) is -1.8e-16 in double,
     * so 511*cos+1.0 rounds to 1-1.1e-13 and truncates to 0).  tests/ compare all 1024 entries with
     * the reference's arrays and with the committed golden copy. */
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < 512; i++) {
        double a = two_pi * (double)i / 512.0;
        sin512[i] = (int)(511.0 * sin(a) + 1.0);
        cos512[i] = (int)(511.0 * cos(a) + 1.0);
    }
}

/* ---- C/A code (plutogpssim.c:207-244) ---------------------------------------------------------- */

void gpsbb_oracle_codegen(int *ca, int prn)
{
    /* G2 delay in chips per PRN 1..32 (plutogpssim.c:208-213) */
    static const short g2_delay[32] = {5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252,
                                       254, 255, 256, 257, 258, 469, 470, 471, 472, 473, 474,
                                       509, 512, 513, 514, 515, 516, 859, 860, 861, 862};
    signed char g1[GPSBB_CA_LEN], g2[GPSBB_CA_LEN];
    signed char r1[10], r2[10]; /* +-1 valued shift registers, all "ones" (-1) at start (c:224-225) */

    if (prn < 1 || prn > 32)
        return; /* c:220-221: leaves ca untouched */

    for (int k = 0; k < 10; k++)
        r1[k] = r2[k] = -1;

    for (int i = 0; i < GPSBB_CA_LEN; i++) {
        g1[i] = r1[9];
        g2[i] = r2[9];
        /* feedback: G1 = x^3 + x^10, G2 = x^2+x^3+x^6+x^8+x^9+x^10, as products of +-1 (c:229-230) */
        signed char f1 = (signed char)(r1[2] * r1[9]);
        signed char f2 = (signed char)(r2[1] * r2[2] * r2[5] * r2[7] * r2[8] * r2[9]);
        memmove(r1 + 1, r1, 9);
        memmove(r2 + 1, r2, 9);
        r1[0] = f1;
        r2[0] = f2;
    }

    /* chip i = (1 - G1[i]*G2[i - delay]) / 2  -> 0/1   (c:240-241) */
    int j = GPSBB_CA_LEN - g2_delay[prn - 1];
    for (int i = 0; i < GPSBB_CA_LEN; i++, j++)
        ca[i] = (1 - g1[i] * g2[j % GPSBB_CA_LEN]) / 2;
}

/* ---- descriptor contract (include/gpsbb.h, gpsbb_chan_t) ---------------------------------------- */

static int chan_ok(const gpsbb_chan_t *c, double delt, int fixed)
{
    if (c->prn == 0)
        return 1;
    if (c->prn < 0 || c->prn > 32)
        return 0;
    if (!isfinite(c->f_carr) || !isfinite(c->f_code) || !isfinite(c->carr_phase) ||
        !isfinite(c->code_phase) || !isfinite(c->gain))
        return 0;
    if (!fixed && (signbit(c->carr_phase) || c->carr_phase > 1.0))
        return 0;
    if (fixed && (signbit(c->carr_phase) || c->carr_phase >= 4294967296.0 || c->carr_phase != floor(c->carr_phase)))
        return 0; /* fixed-point variant: the 32-bit accumulator's value (h:160) */
    if (signbit(c->code_phase) || !(c->code_phase < 1023.0))
        return 0;
    double sc = c->f_code * delt, sk = c->f_carr * delt;
    if (!(sc > 0.0 && sc <= 1.5) || !(fabs(sk) <= 0.125))
        return 0;
    if (!(fabs(c->gain) < 2097152.0))
        return 0;
    if (c->iword < 0 || c->iword > 59 || c->ibit < 0 || c->ibit > 29 || c->icode < 0 || c->icode > 19)
        return 0;
    for (int k = 0; k < GPSBB_N_DWRD; k++)
        if (c->dwrd[k] >> 30)
            return 0;
    return 1;
}

/* working copy of one channel: the hot fields of channel_t (plutogpssim.h:152-174) */
typedef struct {
    int prn;
    int ca[GPSBB_CA_LEN];
    double f_carr, f_code, carr_phase, code_phase, gain;
    const uint32_t *dwrd;
    int iword, ibit, icode, dataBit, codeCA;
    unsigned int carr_acc; /* fixed-point variant: unsigned int carr_phase (h:160) */
    int carr_step;         /* int carr_phasestep (h:161) */
} ochan_t;

static int nav_bit(const ochan_t *c, gpsbb_hazards_t *hz)
{
    int w = c->iword;
    if (w >= GPSBB_N_DWRD) { /* latent out-of-bounds read of the reference (c:2732): defined + counted */
        w = GPSBB_N_DWRD - 1;
        if (hz)
            hz->dwrd_oob++;
    }
    return (int)((c->dwrd[w] >> (29 - c->ibit)) & 1u) * 2 - 1;
}

static void load_chan(ochan_t *o, const gpsbb_chan_t *c, gpsbb_hazards_t *hz)
{
    o->prn = c->prn;
    if (c->prn <= 0)
        return;
    gpsbb_oracle_codegen(o->ca, c->prn);
    o->f_carr = c->f_carr;
    o->f_code = c->f_code;
    o->carr_phase = c->carr_phase;
    o->code_phase = c->code_phase;
    o->gain = c->gain;
    o->dwrd = c->dwrd;
    o->iword = c->iword;
    o->ibit = c->ibit;
    o->icode = c->icode;
    /* computeCodePhase's last two assignments (c:1780-1781) */
    o->codeCA = o->ca[(int)o->code_phase] * 2 - 1;
    o->dataBit = nav_bit(o, hz);
}

static void fill_core(ochan_t *oc, int nch, double delt, int nsamp, int16_t *iq, const int *sinT,
                      const int *cosT, gpsbb_hazards_t *hz, int fixed)
{
    for (int n = 0; n < nsamp; n++) {
        int64_t i_acc = 0, q_acc = 0; /* c:2691-2692 */
        for (int i = 0; i < nch; i++) {
            ochan_t *c = &oc[i];
            if (c->prn <= 0) /* c:2695 */
                continue;

            int it;
            if (fixed) {
                it = (c->carr_acc >> 16) & 0x1ff; /* 9-bit index, c:2699 */
            } else {
                /* carrier table index (c:2697); ==512 only if carr_phase==1.0 exactly (latent OOB) */
                it = (int)floor(c->carr_phase * 512.0);
                if (it > 511) {
                    it &= 511;
                    if (hz)
                        hz->itable_512++;
                }
            }

            /* int*int*int -> double product with gain -> truncation toward zero (c:2701-2702) */
            int ip = (int)(c->dataBit * c->codeCA * cosT[it] * c->gain);
            int qp = (int)(c->dataBit * c->codeCA * sinT[it] * c->gain);
            i_acc += ip; /* c:2705-2706 */
            q_acc += qp;

            /* code NCO (c:2709-2734) */
            c->code_phase += c->f_code * delt;
            if (c->code_phase >= GPSBB_CA_LEN) {
                c->code_phase -= GPSBB_CA_LEN;
                if (++c->icode >= 20) { /* 20 code periods per data bit */
                    c->icode = 0;
                    if (++c->ibit >= 30) { /* 30 bits per word */
                        c->ibit = 0;
                        c->iword++;
                    }
                    c->dataBit = nav_bit(c, hz);
                }
            }
            c->codeCA = c->ca[(int)c->code_phase] * 2 - 1; /* c:2737 */

            if (fixed) {
                c->carr_acc += (unsigned int)c->carr_step; /* c:2748 */
            } else {
                /* carrier NCO, FLOAT_CARR_PHASE variant (c:2741-2746) */
                c->carr_phase += c->f_carr * delt;
                if (c->carr_phase >= 1.0)
                    c->carr_phase -= 1.0;
                else if (c->carr_phase < 0.0)
                    c->carr_phase += 1.0;
            }
        }
        if (iq) {
            iq[2 * n] = (int16_t)i_acc; /* (short) cast, c:2754-2755 */
            iq[2 * n + 1] = (int16_t)q_acc;
        }
    }
}

static void store_state(gpsbb_chan_state_t *s, const ochan_t *c)
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

static int fill_blocks(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp, int chain,
                       int16_t *iq, gpsbb_chan_state_t *end_state, gpsbb_hazards_t *hz, int fixed)
{
    static int sinT[512], cosT[512];
    static int have_tables;
    if (!ch || nblocks < 0 || nch < 0 || nch > GPSBB_MAX_CHAN || nsamp < 0 || !(delt > 0.0))
        return -1;
    for (int k = 0; k < nblocks * nch; k++)
        if (!chan_ok(&ch[k], delt, fixed))
            return -1;
    if (!have_tables) {
        gpsbb_oracle_tables(sinT, cosT);
        have_tables = 1;
    }

    ochan_t oc[GPSBB_MAX_CHAN];
    int prev_prn[GPSBB_MAX_CHAN];
    double prev_phase[GPSBB_MAX_CHAN];
    for (int i = 0; i < GPSBB_MAX_CHAN; i++)
        prev_prn[i] = 0;

    for (int b = 0; b < nblocks; b++) {
        const gpsbb_chan_t *cb = ch + (size_t)b * nch;
        for (int i = 0; i < nch; i++) {
            load_chan(&oc[i], &cb[i], hz);
            /* carr_phase is never re-seeded while a channel stays allocated (c:2741-2746; born c:1964) */
            if (chain && b > 0 && oc[i].prn > 0 && prev_prn[i] == oc[i].prn)
                oc[i].carr_phase = prev_phase[i];
            if (fixed && oc[i].prn > 0) {
                oc[i].carr_acc = (unsigned int)oc[i].carr_phase;
                /* per-block phase step of the fixed-point variant (c:2675) */
                oc[i].carr_step = (int)round(512.0 * 65536.0 * oc[i].f_carr * delt);
            }
        }
        fill_core(oc, nch, delt, nsamp, iq ? iq + (size_t)b * 2 * nsamp : NULL, sinT, cosT, hz, fixed);
        for (int i = 0; i < nch; i++) {
            if (fixed && oc[i].prn > 0)
                oc[i].carr_phase = (double)oc[i].carr_acc;
            prev_prn[i] = oc[i].prn > 0 ? oc[i].prn : 0;
            prev_phase[i] = oc[i].carr_phase;
            if (end_state)
                store_state(&end_state[(size_t)b * nch + i], &oc[i]);
        }
    }
    return 0;
}

int gpsbb_oracle_fill_blocks(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                             int chain, int16_t *iq, gpsbb_chan_state_t *end_state,
                             gpsbb_hazards_t *hz)
{
    return fill_blocks(ch, nblocks, nch, delt, nsamp, chain, iq, end_state, hz, 0);
}

int gpsbb_oracle_fill_blocks_fixed(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                                   int chain, int16_t *iq, gpsbb_chan_state_t *end_state,
                                   gpsbb_hazards_t *hz)
{
    return fill_blocks(ch, nblocks, nch, delt, nsamp, chain, iq, end_state, hz, 1);
}

int gpsbb_oracle_fill(const gpsbb_chan_t *ch, int nch, double delt, int nsamp, int16_t *iq,
                      gpsbb_chan_state_t *end_state, gpsbb_hazards_t *hz)
{
    return gpsbb_oracle_fill_blocks(ch, 1, nch, delt, nsamp, 0, iq, end_state, hz);
}
