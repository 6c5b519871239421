/*
 * gpsbb_walk.hip.h — the exact NCO pre-pass of the breakpoint kernel, in two kernels (gfx950):
 *
 *   k_walk   One lane per NCO chain (block x channel x {code, carrier}), as k_seed, but written so that the
 *            64 lanes of a wavefront stay in LOCKSTEP: every turn of the loop is, for every lane, one regular
 *            run of the exact jump-ahead (gpsbb_nco.h: possibly of zero steps) followed by one genuine IEEE
 *            step (plutogpssim.c:2709-2712 / 2741-2746), all branch-free; only the rare cases (tiny or zero
 *            steps, states far below the step) leave the common path, behind a wave-uniform test.  k_seed
 *            spends ~1500 cycles per row because lanes that are in different phases of the walk serialise;
 *            here a turn costs its instruction count.  It emits one row {n0, bits, x, S} per turn into the
 *            chain's region of the row pool — inside a row the state at sample n is exactly
 *            fma(n - n0, S, x) — and the end-of-block state.  Nothing per tile happens here.
 *
 *   k_tiles  Fully parallel: one lane per row.  The tiles whose first sample lies in the row get their exact
 *            state (BatchDev::tile_x / tile_nav), which is all k_synth_ev reads.
 */
#ifndef GPSBB_WALK_HIP_H
#define GPSBB_WALK_HIP_H

#include "gpsbb_kernels.hip.h"

namespace gpsbb_impl {

#ifndef GPSBB_WALK_WG
#define GPSBB_WALK_WG 64
#endif
constexpr int WALK_ROW_MAX = 4 * TILE - 1; /* steps of one regular run (see walk_lockstep) */

/* what a lane carries through the walk */
template <int KIND>
struct WalkLane {
    double x, s;
    int32_t n;
    uint32_t nav;  /* code: packed nav counters */
    uint32_t bits; /* code: bit 0 = data bit in force is -1, bit 1 = the one after the next roll-over is -1 */
    const uint32_t *dwrd;
    SynRow *rows;
    uint32_t cap, cnt;
    bool active;
    bool stuck; /* x + s rounded back to x and nothing wrapped: the state is constant from here on */
};

/* code chains: the data bits of a row from the nav counters (c:2717-2733) */
__device__ __forceinline__ uint32_t walk_dbits(const uint32_t *dwrd, uint32_t nav, uint32_t cur)
{
    const uint32_t nav1 = nav_advance(nav);
    const uint32_t nxt = nav_icode(nav1) == 0 ? (nav_bit(dwrd, nav1) < 0 ? 2u : 0u) : (cur ? 2u : 0u);
    return (cur ? 1u : 0u) | nxt;
}

/* bit d (2 <= d <= 50) set: rounding s to a multiple of 2^d units of its own last place is a tie — the low d
 * bits of its mantissa are exactly 1 followed by zeros.  A state d binades above s then only takes a regular
 * run from an even mantissa (gpsbb_nco.h: "half-way case on an odd mantissa"). */
__device__ __forceinline__ uint64_t walk_tiemask(uint64_t sb)
{
    const uint64_t Ms = (sb & F64_MANT) | F64_HID;
    uint64_t m = 0;
    for (int d = 2; d <= 50; d++)
        if ((Ms & ((1ull << d) - 1)) == (1ull << (d - 1)))
            m |= 1ull << d;
    return m;
}

/*
 * One chain per lane, lanes in lockstep; SNEG: the step is negative (the caller masks the lanes by the sign of
 * their step, so that everything that depends on the direction is straight-line code).  One turn of the loop:
 *
 *   regular run   With x in binade e (ulp u), at least two and at most 50 binades above the step, a step adds
 *                 exactly S = s rounded to a multiple of u, ties to even = (s + 1.5*2^e) - 1.5*2^e, for as long
 *                 as the state stays inside the binade (below 1023 for the code): k = floor(room / |S|) more
 *                 steps, room = the distance to the last state inside.  A row is never longer than WALK_ROW_MAX
 *                 steps, so a reciprocal good to 2^-26 settles k to within one, and the exact remainder
 *                 fma(-k, |S|, room) decides.  k = 0 where no regular run applies (state zero, negative,
 *                 subnormal, beyond the wrap threshold's binade, less than two binades above the step, or a
 *                 half-way step on an odd mantissa).
 *   one step      x + s with the reference's wrap (c:2709-2712 / 2741-2746), genuine IEEE adds.
 *
 * The rare cases — tiny or zero steps (es < 123), states more than 50 binades above the step, a state that
 * no longer moves — go through the integer version of the regular run (gpsbb_nco.h) behind a wave-uniform test.
 */
template <int KIND, bool SNEG>
__device__ __forceinline__ void walk_lockstep(WalkLane<KIND> &w, int nsamp, unsigned long long *hz, uint32_t *status)
{
    constexpr int TOPEX = KIND == NCO_CARR ? 1023 : 1023 + 10;
    const double s = w.s;
    const uint64_t sb = f64_bits(s);
    const int es = (int)((sb >> 52) & 0x7ff);
    const bool generic = es < 123; /* tiny or zero step: every run through the integer version */
    const uint64_t tiemask = generic ? 0ull : walk_tiemask(sb);
    const bool any_tie = __ballot(w.active && tiemask != 0ull) != 0ull;
    while (__ballot(w.active)) {
        const double x = w.x;
        const uint32_t hi = (uint32_t)__double2hiint(x);
        const int ex = (int)(hi >> 20); /* sign bit included: a negative state (-0.0) counts as beyond the range */
        const int d = ex - es;
        const bool weird = (unsigned)(ex - 1) >= (unsigned)(TOPEX - 1); /* zero, subnormal, negative, beyond the top */
        const bool rare = w.active && !weird && (generic || d > 50 || w.stuck);
        bool expl = weird || d < 2;
        if (any_tie)
            expl |= ((tiemask >> (d & 63)) & 1ull) != 0ull && (__double2loint(x) & 1);
        /* S = s rounded to a multiple of ulp(x), ties to even: adding and subtracting 1.5 * 2^e */
        const double C = __hiloint2double((int)((hi & 0xfff00000u) | 0x80000u), 0);
        double S = add_rn(add_rn(s, C), -C);
        /* the last state inside the binade in the direction of the step, and the distance to it */
        double room;
        if (!SNEG) {
            double lim = __hiloint2double((int)(hi | 0xfffffu), -1); /* 2^(e+1) - ulp */
            if (KIND == NCO_CODE)
                lim = ex == 1023 + 9 ? 0x1.ff7ffffffffffp+9 /* 1023 - ulp */ : lim;
            room = add_rn(lim, -x);
        } else {
            room = add_rn(x, -__hiloint2double((int)(hi & 0xfff00000u), 1)); /* 2^e + ulp */
        }
        const double Sa = SNEG ? -S : S;
        const double kq = fmin(room * __builtin_amdgcn_rcp(Sa), 4096.0);
        int ki = (int)kq;
        const double rem = __fma_rn(-(double)ki, Sa, room); /* exact: |rem| < 2|S| */
        ki += (rem < 0.0 ? -1 : 0) + (rem >= Sa ? 1 : 0);
        /* a row never runs past the end of the block, nor past WALK_ROW_MAX samples: k_tiles gives every row one
         * lane, which then has at most five tiles to write (only the rows of slow chains are ever cut) */
        const int kleft = nsamp - w.n;
        const int kcap = kleft < WALK_ROW_MAX ? kleft : WALK_ROW_MAX;
        int k = (expl || !(room >= Sa)) ? 0 : (ki < kcap ? ki : kcap);
        double x1 = __fma_rn((double)k, S, x);
        if (__builtin_expect(__ballot(rare || (w.active && weird)) != 0ull, 0)) {
            if (KIND == NCO_CARR && w.active && ex >= TOPEX && !(hi >> 31))
                atomicAdd(hz, 1ull); /* carr_phase == 1.0: table index 512, one past the reference's tables */
            if (rare && w.stuck) {
                /* constant from here on: one row to the end of the block (cut like any other) */
                k = kcap;
                S = 0.0;
                x1 = x;
            } else if (rare) {
                /* the integer version of the regular run */
                int64_t inc;
                const uint64_t xb = f64_bits(x);
                k = (int)regular_run<KIND>(xb, sb, (int64_t)kcap, inc);
                S = step_of_inc(xb, inc);
                x1 = bits_f64(xb + (uint64_t)((int64_t)k * inc));
            }
        }
        /* the row of this turn: samples n .. n + k */
        if (w.active) {
            if (w.cnt < w.cap) {
                SynRow row;
                row.n0 = w.n;
                row.nav = w.bits;
                row.x = x;
                row.S = S;
                w.rows[w.cnt] = row;
            } else {
                atomicOr(status, ST_ROW_OVERFLOW);
            }
            w.cnt++;
        }
        /* one genuine step, sample n + k -> n + k + 1 (unless the block ends with the run) */
        const int n1 = w.n + k;
        const bool step = w.active && n1 < nsamp;
        double x2 = add_rn(x1, s);
        bool wrapped;
        if (KIND == NCO_CARR) {
            /* c:2743-2746; a rising phase can only pass 1.0, a falling one only 0.0 */
            wrapped = SNEG ? x2 < 0.0 : x2 >= 1.0;
            const double xw = add_rn(x2, SNEG ? 1.0 : -1.0);
            x2 = wrapped ? xw : x2;
        } else {
            wrapped = x2 >= 1023.0;
            x2 = wrapped ? add_rn(x2, -1023.0) : x2; /* c:2711-2712 */
            if (__ballot(step && wrapped)) {
                if (step && wrapped) {
                    w.nav = nav_advance(w.nav); /* c:2714-2733 */
                    uint32_t cur = w.bits & 1u;
                    if (nav_icode(w.nav) == 0) {
                        if (nav_iword(w.nav) >= GPSBB_N_DWRD)
                            atomicAdd(hz + 1, 1ull);
                        cur = nav_bit(w.dwrd, w.nav) < 0 ? 1u : 0u;
                    }
                    w.bits = walk_dbits(w.dwrd, w.nav, cur);
                }
            }
        }
        if (__builtin_expect(__ballot(rare) != 0ull, 0)) /* only a state far above the step can stop moving */
            w.stuck = rare && step && !wrapped && f64_bits(x2) == f64_bits(x1);
        w.x = step ? x2 : (w.active ? x1 : w.x);
        w.n = step ? n1 + 1 : (w.active ? n1 : w.n); /* lanes waiting for the other direction's loop keep theirs */
        w.active = step && w.n < nsamp;
    }
}

/* the lanes of a wavefront by the sign of their step, each group in its own straight-line loop (the host's plan
 * keeps the signs apart, so a wavefront normally runs only one of the two) */
template <int KIND>
__device__ __forceinline__ void walk_both_signs(WalkLane<KIND> &w, int nsamp, unsigned long long *hz, uint32_t *status)
{
    const bool on = w.active;
    const bool neg = w.s < 0.0;
    if (__ballot(on && !neg)) {
        w.active = on && !neg;
        walk_lockstep<KIND, false>(w, nsamp, hz, status);
    }
    if (KIND == NCO_CARR && __ballot(on && neg)) {
        w.active = on && neg;
        walk_lockstep<KIND, true>(w, nsamp, hz, status);
    }
}

/* Lane -> chain as planned in BatchDev::seed_order (code and carrier chains never share a wavefront). */
__global__ __launch_bounds__(GPSBB_WALK_WG) void k_walk(BatchDev p)
{
    __builtin_amdgcn_s_setprio(GPSBB_SEED_PRIO);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = gid < p.seed_lanes ? p.seed_order[gid] : -1;
    const int nbc = p.nblocks * p.nch;
    const bool is_code = c >= 0 && c < nbc, is_carr = c >= nbc;
    if (__ballot(is_code)) {
        WalkLane<NCO_CODE> w;
        const int k = is_code ? c : 0;
        const gpsbb_chan_t &ch = p.ch[k];
        const bool on = is_code && ch.prn > 0;
        w.x = ch.code_phase;
        w.s = mul_rn(ch.f_code, p.delt); /* plutogpssim.c:2709: f_code * delt, rounded on its own */
        w.n = 0;
        w.nav = nav_pack(ch.icode, ch.ibit, ch.iword);
        w.dwrd = ch.dwrd;
        w.bits = on ? walk_dbits(ch.dwrd, w.nav, nav_bit(ch.dwrd, w.nav) < 0 ? 1u : 0u) : 0u;
        const uint64_t o0 = p.row_off[k], o1 = p.row_off[k + 1];
        w.rows = p.rows + o0;
        w.cap = (uint32_t)(o1 - o0);
        w.cnt = 0;
        w.active = on;
        w.stuck = false;
        walk_both_signs<NCO_CODE>(w, p.nsamp, p.hazards, p.status);
        if (is_code) {
            gpsbb_chan_state_t &e = p.end[k];
            p.row_cnt[k] = on ? (int32_t)(w.cnt < w.cap ? w.cnt : w.cap) : 0;
            if (on) {
                e.code_phase = w.x;
                e.iword = nav_iword(w.nav);
                e.ibit = nav_ibit(w.nav);
                e.icode = nav_icode(w.nav);
                e.dataBit = nav_bit(ch.dwrd, w.nav);
                const int ci = (int)w.x;
                e.codeCA = (int)((p.ca_bits[ch.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1; /* c:2737 */
            } else {
                e.code_phase = 0.0;
                e.iword = e.ibit = e.icode = e.dataBit = e.codeCA = 0;
            }
            e._pad = 0;
        }
    }
    if (__ballot(is_carr)) {
        WalkLane<NCO_CARR> w;
        const int k = is_carr ? c - nbc : 0;
        const gpsbb_chan_t &ch = p.ch[k];
        const bool on = is_carr && ch.prn > 0;
        w.x = ch.carr_phase;
        w.s = mul_rn(ch.f_carr, p.delt); /* plutogpssim.c:2741 */
        w.n = 0;
        w.nav = 0;
        w.bits = 0;
        w.dwrd = nullptr;
        const uint64_t o0 = p.row_off[nbc + k], o1 = p.row_off[nbc + k + 1];
        w.rows = p.rows + o0;
        w.cap = (uint32_t)(o1 - o0);
        w.cnt = 0;
        w.active = on;
        w.stuck = false;
        walk_both_signs<NCO_CARR>(w, p.nsamp, p.hazards, p.status);
        if (is_carr) {
            p.row_cnt[nbc + k] = on ? (int32_t)(w.cnt < w.cap ? w.cnt : w.cap) : 0;
            p.end[k].carr_phase = on ? w.x : 0.0;
        }
    }
}

/*
 * Rows -> tile states.  Workgroup = one chain; its lanes stride over the chain's rows.  Row r holds samples
 * n0[r] .. n0[r+1]-1 (the last one: to the end of the block); every tile whose first sample is one of them
 * gets fma(tile start - n0, S, x), exactly the chain's state there.
 */
#ifndef GPSBB_TILES_WG
#define GPSBB_TILES_WG 256
#endif
__global__ __launch_bounds__(GPSBB_TILES_WG) void k_tiles(BatchDev p)
{
    const int chain = blockIdx.x;
    const int nbc = p.nblocks * p.nch;
    const int kind = chain >= nbc ? 1 : 0, bi = chain - kind * nbc;
    const int cnt = p.row_cnt[chain];
    if (cnt <= 0)
        return;
    const SynRow *__restrict__ rows = p.rows + p.row_off[chain];
    const int b = bi / p.nch, i = bi % p.nch;
    double *__restrict__ tx = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + kind) * (size_t)p.ntiles;
    uint32_t *__restrict__ tn = p.tile_nav + ((size_t)b * (size_t)p.nch + i) * (size_t)p.ntiles;
    /* four rows per lane and turn, their loads issued together: the kernel is bound by the latency of these
     * loads, not by their number */
    constexpr int U = 4;
    for (int r0 = threadIdx.x; r0 < cnt; r0 += U * GPSBB_TILES_WG) {
        SynRow row[U];
        int n_next[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int r = r0 + j * GPSBB_TILES_WG;
            const int rc = r < cnt ? r : cnt - 1;
            row[j] = rows[rc];
            n_next[j] = rc + 1 < cnt ? rows[rc + 1].n0 : INT32_MAX - TILE;
            if (r >= cnt)
                n_next[j] = row[j].n0; /* past the chain's last row: no tiles */
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            int t = (int)(((uint32_t)row[j].n0 + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            int t_end = (int)(((uint32_t)n_next[j] + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            t_end = t_end < p.ntiles ? t_end : p.ntiles;
            for (; t < t_end; t++) {
                const double v = __fma_rn((double)(t * TILE - row[j].n0), row[j].S, row[j].x);
                tx[t] = kind ? mul_rn(v, 512.0) : v;
                if (!kind)
                    tn[t] = row[j].nav;
            }
        }
    }
}

} /* namespace gpsbb_impl */
#endif
