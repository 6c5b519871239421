/*
 * gpsbb_nco.h — exact jump-ahead for the reference's two sequential IEEE-double NCOs.
 *
 * Shared by the device seeding pre-pass (gpsbb_kernels.hip) and the host carrier chaining helper
 * (gpsbb_host.cpp); it compiles as plain C++ too, which is how tests/test_nco_host.py checks it against
 * brute-force stepping without a GPU.
 *
 * The recurrences (reference: plutogpssim.c:2709-2712 code, 2741-2746 carrier, FLOAT_CARR_PHASE on):
 *     code   : x = fl(x + s);  if (x >= 1023.0) { x = fl(x - 1023.0); <advance nav counters> }
 *     carrier: x = fl(x + s);  if (x >= 1.0) x = fl(x - 1.0); else if (x < 0.0) x = fl(x + 1.0);
 * with s = fl(f * delt) computed once (no FMA: the reference is built -std=c11, Makefile:1-2).
 * Sample n's chip / table index depend on n successive roundings, so x0 + n*s is NOT what the
 * reference computes.  What makes the sequence jumpable:
 *
 *   While x stays inside one binade [2^e, 2^(e+1)) and no wrap fires, x = M * 2^(e-52) with an integer
 *   mantissa M in [2^52, 2^53), and fl(x + s) = (M + inc) * 2^(e-52) where inc = RN(s / 2^(e-52)) is a
 *   constant of (s, e): the exact sum is (M + sigma)*ulp with sigma = s/ulp real, M integer, so
 *   round-to-nearest gives M + RN(sigma) unless sigma is exactly half-way, in which case ties-to-even
 *   makes M even after one step and from then on the increment is the even one of {floor, ceil}.
 *   Hence the raw bit pattern of x advances by the integer `inc` per step: bits(x_j) = bits(x_0) + j*inc,
 *   for as many steps as M + j*inc provably stays inside the binade and below the wrap threshold
 *   ("regular run").  Everything else — binade crossings, wraps, x below s's binade, x == 0 — is taken
 *   as ONE explicit step with genuine IEEE adds, after which the next regular run starts.
 *
 * A block is thereby cut into "rows" {n0, bits(x_n0), inc}: row k covers samples n0_k .. n0_{k+1}-1 and
 * the state at any sample n in it is bits(x_n0) + (n - n0)*inc, reinterpreted as a double.
 */
#ifndef GPSBB_NCO_H
#define GPSBB_NCO_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPSBB_HD __host__ __device__ __forceinline__
#else
#define GPSBB_HD inline
#endif

namespace gpsbb_impl {

enum NcoKind { NCO_CODE = 0, NCO_CARR = 1 };

static constexpr uint64_t F64_HID = 1ull << 52;          /* hidden bit            */
static constexpr uint64_t F64_MANT = F64_HID - 1;         /* mantissa field        */
static constexpr uint64_t F64_ABS = ~(1ull << 63);
static constexpr uint64_t CODE_WRAP_M = 1023ull << 43;    /* 1023.0 = (1023*2^43) * 2^(9-52) */
static constexpr int64_t NCO_KINF = INT64_MAX;

GPSBB_HD uint64_t f64_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
GPSBB_HD double bits_f64(uint64_t u)
{
    double x;
    memcpy(&x, &u, 8);
    return x;
}

/* individually rounded IEEE operations (never contracted into an FMA) */
GPSBB_HD double add_rn(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(a, b);
#else
    volatile double r = a + b; /* host: built with -ffp-contract=off as well; volatile pins the rounding */
    return r;
#endif
}
GPSBB_HD double mul_rn(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dmul_rn(a, b);
#else
    volatile double r = a * b;
    return r;
#endif
}

/* One genuine step of the code NCO (plutogpssim.c:2709-2712).  Returns true when the 1023 wrap fired. */
GPSBB_HD bool code_step(double &x, double s)
{
    x = add_rn(x, s);
    if (x >= 1023.0) {
        x = add_rn(x, -1023.0);
        return true;
    }
    return false;
}

/* One genuine step of the carrier NCO (plutogpssim.c:2741-2746).  Returns true when a wrap fired. */
GPSBB_HD bool carr_step(double &x, double s)
{
    x = add_rn(x, s);
    if (x >= 1.0) {
        x = add_rn(x, -1.0);
        return true;
    }
    if (x < 0.0) {
        x = add_rn(x, 1.0);
        return true;
    }
    return false;
}

/* Nav-message counters of one channel (plutogpssim.h:166-169) packed for the rows:
 * bits 0-4 icode (0..19), 5-9 ibit (0..29), 10-25 iword. */
GPSBB_HD uint32_t nav_pack(int icode, int ibit, int iword) { return (uint32_t)icode | ((uint32_t)ibit << 5) | ((uint32_t)iword << 10); }
GPSBB_HD int nav_icode(uint32_t p) { return (int)(p & 31u); }
GPSBB_HD int nav_ibit(uint32_t p) { return (int)((p >> 5) & 31u); }
GPSBB_HD int nav_iword(uint32_t p) { return (int)(p >> 10); }

/* One code-period roll-over (plutogpssim.c:2714-2733), on the packed counters. */
GPSBB_HD uint32_t nav_advance(uint32_t p)
{
    int icode = nav_icode(p) + 1, ibit = nav_ibit(p), iword = nav_iword(p);
    if (icode >= 20) {
        icode = 0;
        if (++ibit >= 30) {
            ibit = 0;
            iword++;
        }
    }
    return nav_pack(icode, ibit, iword);
}

/* floor(a / q) for 0 <= a < 2^53, 0 < q < 2^53, without a 64-bit integer division (which is a ~200
 * instruction software routine on the GPU): reciprocal estimate, one Newton step, exact fix-up by the
 * remainder.  Quotients >= 2^40 are only ever compared with a sample count, so they saturate. */
GPSBB_HD uint64_t div_floor_53(uint64_t a, uint64_t q)
{
    const double qd = (double)q;
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(qd);
#else
    double r = 1.0 / qd;
#endif
    r = r * (2.0 - qd * r); /* any rounding/contraction is fine here: the result is corrected below */
    const double kd = (double)a * r;
    if (kd >= 1099511627776.0)
        return 1ull << 40;
    uint64_t k = (uint64_t)kd;
    int64_t rem = (int64_t)a - (int64_t)(k * q);
    while (rem < 0) {
        k--;
        rem += (int64_t)q;
    }
    while (rem >= (int64_t)q) {
        k++;
        rem -= (int64_t)q;
    }
    return k;
}

/*
 * Regular run starting from state x (raw bits xb, x >= 0) with step s (raw bits sb).
 * Returns k >= 0, the number of consecutive steps (capped at kcap) for which
 *     bits(x_j) = xb + j*inc   (j = 0..k)   and no wrap fires,
 * and sets inc.  k == 0 means "take one explicit step".
 */
template <int KIND>
GPSBB_HD int64_t regular_run(uint64_t xb, uint64_t sb, int64_t kcap, int64_t &inc)
{
    inc = 0;
    const int ex = (int)((xb >> 52) & 0x7ff);
    if (ex == 0 || (xb >> 63))
        return 0; /* zero / subnormal / negative: explicit step */
    if (KIND == NCO_CARR) {
        if (ex >= 1023)
            return 0; /* x >= 1.0 (only the latent x == 1.0 case): the >= 1.0 wrap fires */
    } else {
        if (ex >= 1023 + 10)
            return 0; /* x >= 1024: outside the contract */
    }
    const uint64_t sabs = sb & F64_ABS;
    const bool sneg = (sb >> 63) != 0;
    const uint64_t M = (xb & F64_MANT) | F64_HID;
    if (sabs == 0)
        return kcap; /* x + 0 == x for ever */

    int es = (int)(sabs >> 52);
    uint64_t Ms = sabs & F64_MANT;
    if (es == 0)
        es = 1; /* subnormal step: no hidden bit */
    else
        Ms |= F64_HID;

    const int d = ex - es; /* sigma = s/ulp(x) = Ms / 2^d */
    if (d <= 0)
        return 0; /* x not above s's binade: the sum leaves x's binade at once */

    uint64_t q;
    bool tie = false;
    if (d >= 64) {
        q = 0; /* |sigma| < 2^-11 */
    } else {
        q = Ms >> d;
        const uint64_t r = Ms & ((1ull << d) - 1);
        const uint64_t half = 1ull << (d - 1);
        if (r > half)
            q++;
        else if (r == half)
            tie = true;
    }
    if (tie) {
        if (M & 1)
            return 0; /* odd mantissa on a tie: one explicit step makes it even */
        q += (q & 1); /* even mantissa: ties-to-even keeps it even -> the even neighbour */
    }
    if (q == 0) {
        if (sneg && M == F64_HID)
            return 0; /* power of two and a negative step: spacing below is finer */
        return kcap;  /* |s| < ulp/2: x + s rounds back to x for ever */
    }

    /* distance (in mantissa units) to the edge the run moves towards, written without a branch on the
     * sign of s: lanes with opposite Doppler signs then stay convergent on the GPU */
    uint64_t Mlim = (F64_HID << 1) - 1;
    if (KIND == NCO_CODE && ex == 1023 + 9)
        Mlim = CODE_WRAP_M - 1; /* stay strictly below 1023.0 */
    const uint64_t Mmin = F64_HID + 1; /* stay strictly above the binade's lower edge */
    const uint64_t room = sneg ? (M >= Mmin ? M - Mmin : 0) : (Mlim >= M ? Mlim - M : 0);
    if ((sneg && M < Mmin) || (!sneg && M > Mlim) || room < q)
        return 0;
    inc = sneg ? -(int64_t)q : (int64_t)q;
    const int64_t k = (int64_t)div_floor_53(room, q);
    return k < kcap ? k : kcap;
}

GPSBB_HD double fma_rn(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(a, b, c);
#else
    return __builtin_fma(a, b, c);
#endif
}

/* The step of a regular run as a double: S = inc * ulp(x), exact (|inc| < 2^53 goes to double in two exact
 * 32-bit halves; ulp may be subnormal).  The state j steps into the run is then fma(j, S, x) exactly: the
 * product is exact, the sum is a representable number, so the single rounding does nothing. */
GPSBB_HD double step_of_inc(uint64_t xb, int64_t inc)
{
    uint32_t ex = (uint32_t)(xb >> 52) & 0x7ffu;
    ex = ex ? ex : 1u;
    const double u = bits_f64(ex > 52u ? (uint64_t)(ex - 52u) << 52 : 1ull << (ex - 1u));
    const uint64_t ia = (uint64_t)(inc < 0 ? -inc : inc);
    const double sa = fma_rn((double)(uint32_t)(ia >> 32), mul_rn(u, 4294967296.0), mul_rn((double)(uint32_t)ia, u));
    return inc < 0 ? -sa : sa;
}

/*
 * regular_run() in double arithmetic, for x and s normal and not tiny (>= 2^-900).  Same contract — k
 * consecutive steps that add exactly S = RN(s/ulp)*ulp and stay strictly inside x's binade (below 1023 for
 * the code NCO) — but S directly and a fraction of the instructions (the device pre-pass is one lane per
 * chain and bound by the length of this dependent chain).  Returns -1 when the case is not covered: the
 * caller then uses regular_run().
 *   S:   adding and subtracting C = 1.5*2^e rounds s to a multiple of ulp(x) with ties to even — exactly the
 *        increment an IEEE add applies to an even mantissa; a tie with an odd mantissa is an explicit step;
 *   k:   floor(room / |S|), room = exact distance to the last state inside the binade: reciprocal estimate
 *        + Newton step (error far below 1 for quotients < 2^32; larger ones are capped anyway), settled by
 *        the exact remainder fma(-k, |S|, room).
 */
template <int KIND>
GPSBB_HD int32_t regular_run_f64(double x, double s, int32_t kcap, double &S)
{
    const uint64_t xb = f64_bits(x), sb = f64_bits(s);
    const int ex = (int)((xb >> 52) & 0x7ff), es = (int)((sb >> 52) & 0x7ff);
    const int d = ex - es;
    S = 0.0; /* a row that is a single explicit step still gets a finite S: its state is fma(0, S, x) */
    /* explicit step, as in regular_run(): negative / zero / subnormal x, x at or above the wrap threshold's
     * binade limit, x not above s's binade — and, here, also x only one binade above s (two or three steps,
     * which regular_run() would take as a run): that keeps 1.5*2^e + s inside x's binade below */
    if ((xb >> 63) || ex == 0 || (KIND == NCO_CARR ? ex >= 1023 : ex >= 1023 + 10) || (es >= 123 && d < 2))
        return 0;
    /* rare: tiny x or s, s == 0, steps far below ulp(x) — every lane of a wavefront that needs this costs
     * all of its lanes the long way, so it must stay rare */
    if (ex < 123 || es < 123 || d > 50)
        return -1;
    const double lo = bits_f64((uint64_t)ex << 52);               /* 2^e              */
    const double u = bits_f64((uint64_t)(ex - 52) << 52);          /* ulp of the binade */
    const double C = bits_f64(((uint64_t)ex << 52) | (1ull << 51)); /* 1.5 * 2^e, even mantissa */
    S = add_rn(add_rn(s, C), -C);
    const double r = add_rn(s, -S); /* exact: the bits of s below ulp(x) */
    if ((f64_bits(r) & F64_ABS) == f64_bits(u) - F64_HID && (xb & 1))
        return 0; /* half-way case on an odd mantissa: one explicit step makes it even */
    if (S == 0.0)
        return -1; /* |s| <= ulp/2: x does not move (or the power-of-two corner); left to regular_run() */
    double room;
    if (S > 0.0) {
        const double top = (KIND == NCO_CODE && ex == 1023 + 9) ? 1023.0 : add_rn(lo, lo);
        room = add_rn(add_rn(top, -u), -x); /* stay at or below top - ulp */
    } else {
        room = add_rn(x, -add_rn(lo, u)); /* stay at or above 2^e + ulp */
    }
    const double Sa = bits_f64(f64_bits(S) & F64_ABS);
    if (!(room >= Sa))
        return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    double rs = __builtin_amdgcn_rcp(Sa);
    rs = fma_rn(fma_rn(-Sa, rs, 1.0), rs, rs);
#else
    const double rs = 1.0 / Sa;
#endif
    const double kq = room * rs;
    if (kq >= 4294967296.0)
        return kcap; /* sample counts are below 2^31 */
    double kf = __builtin_floor(kq);
    const double rem = fma_rn(-kf, Sa, room); /* exact: |rem| < 2|S| */
    if (rem < 0.0)
        kf -= 1.0;
    else if (rem >= Sa)
        kf += 1.0;
    if (kf >= (double)kcap)
        return kcap;
    return (int32_t)kf;
}

/*
 * build_rows() for the device pool: rows as {n0, nav, x, S} (see step_of_inc), the fast regular run first.
 * `sink.row(n0, nav, x, S, after_wrap)` once per row in increasing n0 — after_wrap: the step that led to
 * the row's first sample wrapped (code: the 1023 roll-over; carrier: either wrap), or the row starts at the
 * carrier's latent 1.0 (table index 512, defined as 0): that is what decides whether a wavefront walking
 * these samples needs the wrap-capable update; `sink.table_index_512()` for every sample whose carrier phase
 * is exactly 1.0 (always the first sample of a row); `sink.nav_fetch(nav)` as in build_rows().
 */
template <int KIND, class Sink>
GPSBB_HD double build_rows_f64(double x, double s, uint32_t &nav, int nsamp, Sink &sink)
{
    const uint64_t sb = f64_bits(s);
    int32_t n = 0; /* sample counts fit 32 bits: no 64-bit integer arithmetic on the chain's critical path */
    bool after_wrap = false;
    while (n < nsamp) {
        double S;
        int32_t k = regular_run_f64<KIND>(x, s, nsamp - n, S);
        if (KIND == NCO_CARR && !(x < 1.0)) {
            after_wrap = true;
            sink.table_index_512(); /* the reference's latent out-of-bounds table read, c:2697-2702 */
        }
        if (k >= 0) {
            sink.row(n, nav, x, S, after_wrap);
            if (k > 0)
                x = fma_rn((double)k, S, x);
        } else {
            int64_t inc;
            const uint64_t xb = f64_bits(x);
            k = (int32_t)regular_run<KIND>(xb, sb, (int64_t)(nsamp - n), inc);
            sink.row(n, nav, x, step_of_inc(xb, inc), after_wrap);
            if (k > 0)
                x = bits_f64(xb + (uint64_t)((int64_t)k * inc));
        }
        if (k > 0) {
            n += k;
            if (n >= nsamp)
                break;
        }
        /* one explicit step, sample n -> n+1 */
        const uint64_t before = f64_bits(x);
        bool wrapped = false;
        if (KIND == NCO_CODE) {
            wrapped = code_step(x, s);
            if (wrapped) {
                nav = nav_advance(nav);
                if (nav_icode(nav) == 0)
                    sink.nav_fetch(nav);
            }
        } else {
            wrapped = carr_step(x, s);
        }
        after_wrap = wrapped;
        n += 1;
        if (!wrapped && f64_bits(x) == before) {
            /* x + s rounds back to x and nothing wrapped: constant from here on */
            if (n < nsamp)
                sink.row(n, nav, x, 0.0, false);
            break;
        }
    }
    return x;
}

/*
 * Advance a carrier NCO by n steps exactly (no rows emitted): used by the host chaining helper and by
 * tests.  O(number of regular runs) instead of O(n).
 */
GPSBB_HD double carr_jump(double x, double s, int64_t n)
{
    const uint64_t sb = f64_bits(s);
    while (n > 0) {
        int64_t inc;
        uint64_t xb = f64_bits(x);
        int64_t k = regular_run<NCO_CARR>(xb, sb, n, inc);
        if (k > 0) {
            xb += (uint64_t)(k * inc);
            x = bits_f64(xb);
            n -= k;
        } else {
            const double before = x;
            carr_step(x, s);
            n -= 1;
            if (x == before && f64_bits(x) == f64_bits(before)) {
                /* x + s rounds back to x and no wrap fired: it will do so for ever */
                break;
            }
        }
    }
    return x;
}

/* Same for the code NCO; *wraps receives the number of 1023 wraps (code periods completed). */
GPSBB_HD double code_jump(double x, double s, int64_t n, int64_t *wraps)
{
    const uint64_t sb = f64_bits(s);
    int64_t w = 0;
    while (n > 0) {
        int64_t inc;
        uint64_t xb = f64_bits(x);
        int64_t k = regular_run<NCO_CODE>(xb, sb, n, inc);
        if (k > 0) {
            xb += (uint64_t)(k * inc);
            x = bits_f64(xb);
            n -= k;
        } else {
            if (code_step(x, s))
                w++;
            n -= 1;
        }
    }
    if (wraps)
        *wraps = w;
    return x;
}

/* One row of the per-block NCO segment table (see the file comment). 24 bytes. */
struct NcoRow {
    int32_t n0;   /* first sample of the row */
    uint32_t nav; /* code NCO: packed nav counters valid for the whole row (they only change at a wrap,
                     which is always an explicit step, i.e. a row boundary); carrier NCO: 0 */
    uint64_t xb;  /* raw bits of the phase at sample n0 */
    int64_t inc;  /* raw-bit increment per sample inside the row */
};

/*
 * Cut nsamp steps of one NCO into rows.  `sink.row(n0, nav, xb, inc)` is called once per row in
 * increasing n0 (first row at n0 = 0); `sink.nav_fetch(nav)` is called for the code NCO each time a
 * data bit boundary is crossed (icode rolled over to 0), mirroring the reference's dataBit fetch at
 * plutogpssim.c:2732.  Returns the state after nsamp steps.
 */
template <int KIND, class Sink>
GPSBB_HD double build_rows(double x, double s, uint32_t &nav, int nsamp, Sink &sink)
{
    const uint64_t sb = f64_bits(s);
    int64_t n = 0;
    while (n < nsamp) {
        int64_t inc;
        uint64_t xb = f64_bits(x);
        const int64_t k = regular_run<KIND>(xb, sb, (int64_t)nsamp - n, inc);
        sink.row((int32_t)n, nav, xb, inc);
        if (k > 0) {
            xb += (uint64_t)(k * inc);
            x = bits_f64(xb);
            n += k;
            if (n >= nsamp)
                break;
        }
        /* one explicit step, sample n -> n+1 */
        const uint64_t before = f64_bits(x);
        bool wrapped = false;
        if (KIND == NCO_CODE) {
            wrapped = code_step(x, s);
            if (wrapped) {
                nav = nav_advance(nav);
                if (nav_icode(nav) == 0)
                    sink.nav_fetch(nav);
            }
        } else {
            carr_step(x, s);
        }
        n += 1;
        if (!wrapped && f64_bits(x) == before) {
            /* x + s rounds back to x and nothing wrapped: constant from here on */
            if (n < nsamp)
                sink.row((int32_t)n, nav, before, 0);
            break;
        }
    }
    return x;
}

} /* namespace gpsbb_impl */
#endif
