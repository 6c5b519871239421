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

/* A row as this pre-pass keeps it: 16 bytes, one store per turn of the walk.  The increment S of SynRow is not stored:
 * it is a function of the row's first state and the chain's step — s rounded to a multiple of ulp(x), ties to even
 * (walk_row_step) — and k_tiles recomputes it.  (The scattered row stores of the walk are what the kernels running
 * beside it feel most; the chain's region of the pool is addressed in units of these rows.) */
struct WalkRow {
    int32_t n0;   /* first sample of the row */
    uint32_t nav; /* code chains: the data bits in force (walk_dbits) */
    double x;     /* state at n0 */
};
static_assert(sizeof(WalkRow) == 16 && sizeof(WalkRow) <= sizeof(SynRow), "rows of the walk live in the SynRow pool");

/* the increment of a regular run that starts at x: s rounded to a multiple of ulp(x), ties to even — adding and
 * subtracting 1.5 * 2^e (exact for every x, s of the NCOs; 0 where the state does not move) */
__device__ __forceinline__ double walk_row_step(double x, double s)
{
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const double C = __hiloint2double((int)((hi & 0xfff00000u) | 0x80000u), 0);
    return add_rn(add_rn(s, C), -C);
}

/* what a lane carries through the walk */
template <int KIND>
struct WalkLane {
    double x, s;
    int32_t n;
    uint32_t nav;  /* code: packed nav counters */
    uint32_t bits; /* code: bit 0 = data bit in force is -1, bit 1 = the one after the next roll-over is -1 */
    const uint32_t *dwrd;
    WalkRow *rows;
    uint32_t cap, cnt;
    bool active;
    bool stuck; /* x + s rounded back to x and nothing wrapped: the state is constant from here on */
    /* carrier chains of a batch whose carrier is chained on the device (see k_chain_fix) */
    ChainAux *aux;     /* pass B: where the crossings go */
    int32_t ncross;    /* crossings recorded so far; -1: too many */
    int32_t prev_ex;   /* biased exponent of the previous row's states */
    double prev_x1;    /* the state before the last step taken (the last sample of the previous row) */
    bool prev_wrapped; /* the step before this row wrapped */
    bool prev_tie;     /* ... and its "+ 1.0" was an exact tie (falling phase) */
    bool tie_done;     /* a tie after the first wrap has been recorded: both trajectories left it with an even mantissa, the
                          offset is an even number of grid steps from there on and no later tie can change it */
    bool wrap_seen;    /* the first wrap is behind: the offset is settled */
    double margin;     /* smallest distance of a row's first or last state to an edge of its binade */
    uint32_t hz512;    /* carrier: samples whose phase is exactly 1.0 (gpsbb_hazards_t.itable_512) */
};

/* code chains: the data bits of a row from the nav counters (c:2717-2733) */
__device__ __forceinline__ uint32_t walk_dbits(const uint32_t *dwrd, uint32_t nav, uint32_t cur)
{
    const uint32_t nav1 = nav_advance(nav);
    const uint32_t nxt = nav_icode(nav1) == 0 ? (nav_bit(dwrd, nav1) < 0 ? 2u : 0u) : (cur ? 2u : 0u);
    return (cur ? 1u : 0u) | nxt;
}

/* The one d for which rounding s to a multiple of 2^d units of its own last place is a tie — the low d bits of
 * its mantissa are exactly 1 followed by zeros: d = (trailing zeros) + 1.  A state d binades above s only takes
 * a regular run from an even mantissa (gpsbb_nco.h: "half-way case on an odd mantissa"). */
__device__ __forceinline__ int walk_tie_d(uint64_t sb)
{
    const uint64_t Ms = (sb & F64_MANT) | F64_HID;
    return __builtin_ctzll(Ms) + 1;
}
/* as a mask over d = 2..50 (bit d) */
__device__ __forceinline__ uint64_t walk_tiemask(uint64_t sb)
{
    const int d = walk_tie_d(sb);
    return d >= 2 && d <= 50 ? 1ull << d : 0ull;
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
template <int KIND, bool SNEG, bool TRACK, bool STORE = true>
__device__ __forceinline__ void walk_lockstep(WalkLane<KIND> &w, int nsamp, unsigned long long *hz, uint32_t *status)
{
    constexpr int TOPEX = KIND == NCO_CARR ? 1023 : 1023 + 10;
    const double s = w.s;
    const uint64_t sb = f64_bits(s);
    const int es = (int)((sb >> 52) & 0x7ff);
    const bool generic = es < 123; /* tiny or zero step: every run through the integer version */
    const uint64_t tiemask = generic ? 0ull : walk_tiemask(sb);
    const bool any_tie = __ballot(w.active && tiemask != 0ull) != 0ull;
    /* pass B, falling phase whose step ties on the top binade's grid (its low bits are exactly half of 2^-53): there
     * EVERY step in [0.5, 1) is a tie, taken from an even mantissa by a regular run and from an odd one by one
     * explicit step first.  A true trajectory an odd number of grid steps away does the other of the two at the
     * first sample after the first wrap — and is an even number away ever after.  That sample gets a row of its
     * own and the row after it is recorded like a tie, so that k_chain_fix steps through it. */
    const bool ttf = TRACK && SNEG && KIND == NCO_CARR && !generic && walk_tie_d(sb) == 1022 - es;
    while (__ballot(w.active)) {
        const double x = w.x;
        const uint32_t hi = (uint32_t)__double2hiint(x);
        const int ex = (int)(hi >> 20); /* sign bit included: a negative state (-0.0) counts as beyond the range */
        const int d = ex - es;
        const bool weird = (unsigned)(ex - 1) >= (unsigned)(TOPEX - 1); /* zero, subnormal, negative, beyond the top */
        const bool rare = w.active && (w.stuck || (!weird && (generic || d > 50)));
        bool expl = weird || d < 2;
        const bool first_wrap_row = TRACK && KIND == NCO_CARR && w.active && w.prev_wrapped && !w.wrap_seen;
        if (TRACK && SNEG)
            expl |= ttf && first_wrap_row;
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
        if (TRACK) {
            /* how close the row's first and last state come to the edges of their binade: the trajectory of a
             * start phase that differs by less than that takes every rounding on the same grid (k_chain_fix) */
            const double lo = __hiloint2double((int)(hi & 0xfff00000u), 0);
            const double dl = add_rn(SNEG ? x1 : x, -lo), dh = add_rn(add_rn(lo, lo), -(SNEG ? x : x1));
            w.margin = w.active ? fmin(w.margin, (weird || rare) ? 0.0 : fmin(dl, dh)) : w.margin;
            if (KIND == NCO_CARR) {
                /* this row's states lie on a coarser grid than the previous row's (or the step wrapped): the
                 * offset to the true trajectory may change here.  Few per block: only until the first wrap. */
                /* ... and, for a falling phase, wherever the "+ 1.0" of a wrap was an exact tie: the sum then goes to
                 * the even neighbour, which for an offset of an odd number of grid steps is the other one */
                const bool cross = w.active && ((!w.wrap_seen && (ex > w.prev_ex || w.prev_wrapped)) || (w.prev_tie && !w.tie_done));
                if (__builtin_expect(__ballot(cross) != 0ull, 0)) {
                    if (cross && w.prev_tie) {
                        atomicAdd(hz + 5, 1ull);
                        /* A tie settles the parity of the offset only if BOTH trajectories tied, i.e. if the offset
                         * already was a multiple of the coarsest grid: true from the first wrap on.  At the first
                         * wrap itself the offset may still be half a step of that grid — pass B ties, the true
                         * trajectory lands on a grid point (odd or even) — so later ties still have to be looked at. */
                        w.tie_done = w.wrap_seen;
                    }
                    if (cross) {
                        /* what k_chain_fix needs of it: where, and pass B's states either side of the step */
                        if (w.ncross >= 0 && w.ncross < CHAIN_MAX_CROSS) {
                            w.aux->cross[w.ncross] = w.n;
                            w.aux->pre[w.ncross] = w.prev_x1;
                            w.aux->post[w.ncross] = x;
                            w.ncross++;
                        } else {
                            w.ncross = -1;
                        }
                        if (w.prev_wrapped && !w.wrap_seen) {
                            w.aux->wrap_row = w.n;
                            w.aux->wrap_x = x;
                        }
                        w.wrap_seen = w.wrap_seen || w.prev_wrapped;
                    }
                }
                w.prev_ex = w.active ? ex : w.prev_ex;
            }
        }
        if (__builtin_expect(__ballot(rare || (w.active && weird)) != 0ull, 0)) {
            if (KIND == NCO_CARR && w.active && ex >= TOPEX && !(hi >> 31))
                w.hz512++; /* carr_phase == 1.0: table index 512, one past the reference's tables */
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
        /* Only rows that hold the first sample of a tile are kept: k_tiles reads nothing else of them (most rows
         * of a fast chain are the few-sample ones of the low binades after every wrap), and the scattered row
         * stores are what the kernels running beside the walk feel most.  Pass A keeps none. */
        const bool keeps = w.n == 0 || ((uint32_t)(w.n + k) / (uint32_t)TILE) != ((uint32_t)(w.n - 1) / (uint32_t)TILE);
        if (STORE && w.active && keeps) {
            if (w.cnt < w.cap) {
                WalkRow row;
                row.n0 = w.n;
                row.nav = w.bits;
                row.x = x;
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
        bool wrapped, tie = false;
        if (KIND == NCO_CARR) {
            /* c:2743-2746; a rising phase can only pass 1.0, a falling one only 0.0 */
            wrapped = SNEG ? x2 < 0.0 : x2 >= 1.0;
            const double xw = add_rn(x2, SNEG ? 1.0 : -1.0);
            /* was the sum rounded on the coarsest grid exactly half-way?  (Fast2Sum: both differences are exact.)
             * Falling: x2 + 1.0 on the 2^-53 grid; rising: x1 + s on the 2^-52 grid of [1, 2). */
            if (SNEG && TRACK)
                tie = (wrapped && fabs(add_rn(add_rn(xw, -1.0), -x2)) == 0x1p-54) || (ttf && first_wrap_row);
            if (!SNEG && TRACK)
                tie = wrapped && fabs(add_rn(add_rn(x2, -x1), -s)) == 0x1p-53;
#ifdef GPSBB_EXP_NOTIE /* experiment: what the test suite says when the ties are not looked for */
            tie = false;
#endif
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
        /* only a state far above the step, or a zero one with a zero step, can stop moving */
        if (__builtin_expect(__ballot(rare || (w.active && weird)) != 0ull, 0))
            w.stuck = (rare || weird) && step && !wrapped && f64_bits(x2) == f64_bits(x1);
        w.x = step ? x2 : (w.active ? x1 : w.x);
        w.n = step ? n1 + 1 : (w.active ? n1 : w.n); /* lanes waiting for the other direction's loop keep theirs */
        if (KIND == NCO_CARR && TRACK) {
            w.prev_wrapped = w.active ? (step && wrapped) : w.prev_wrapped;
            w.prev_tie = w.active ? (step && tie) : w.prev_tie;
            /* the block's last step has no row after it: if it crossed upwards or wrapped, the crossing is
             * recorded here, its "row" being the end state */
            const bool last_cross = step && !(w.n < nsamp) &&
                                    ((!w.wrap_seen && (wrapped || (int)((uint32_t)__double2hiint(x2) >> 20) > ex)) || (tie && !w.tie_done));
            if (__builtin_expect(__ballot(last_cross) != 0ull, 0)) {
                if (last_cross) {
                    if (w.ncross >= 0 && w.ncross < CHAIN_MAX_CROSS) {
                        w.aux->cross[w.ncross] = nsamp;
                        w.aux->pre[w.ncross] = x1;
                        w.aux->post[w.ncross] = x2;
                        w.ncross++;
                    } else {
                        w.ncross = -1;
                    }
                }
            }
            w.prev_x1 = w.active ? x1 : w.prev_x1;
        }
        w.active = step && w.n < nsamp;
    }
}

/* the lanes of a wavefront by the sign of their step, each group in its own straight-line loop (the host's plan
 * keeps the signs apart, so a wavefront normally runs only one of the two) */
template <int KIND, bool TRACK, bool STORE = true>
__device__ __forceinline__ void walk_both_signs(WalkLane<KIND> &w, int nsamp, unsigned long long *hz, uint32_t *status)
{
    const bool on = w.active;
    const bool neg = w.s < 0.0;
    if (__ballot(on && !neg)) {
        w.active = on && !neg;
        walk_lockstep<KIND, false, TRACK, STORE>(w, nsamp, hz, status);
    }
    if (KIND == NCO_CARR && __ballot(on && neg)) {
        w.active = on && neg;
        walk_lockstep<KIND, true, TRACK, STORE>(w, nsamp, hz, status);
    }
}

/* a lane set up for chain k of its kind: block-channel k, rows region `chain` of the pool */
template <int KIND>
__device__ __forceinline__ WalkLane<KIND> walk_lane(const BatchDev &p, int chain, double x0, double s, bool on)
{
    WalkLane<KIND> w;
    w.x = x0;
    w.s = s;
    w.n = 0;
    w.nav = 0;
    w.bits = 0;
    w.dwrd = nullptr;
    const uint64_t o0 = p.row_off[chain], o1 = p.row_off[chain + 1];
    w.rows = reinterpret_cast<WalkRow *>(p.rows) + o0;
    w.cap = (uint32_t)(o1 - o0);
    w.cnt = 0;
    w.active = on;
    w.stuck = false;
    w.aux = nullptr;
    w.ncross = 0;
    w.prev_ex = 0x7fff;
    w.prev_x1 = 0.0;
    w.prev_wrapped = false;
    w.prev_tie = false;
    w.tie_done = false;
    w.wrap_seen = false;
    w.margin = 1.0;
    w.hz512 = 0;
    return w;
}

/*
 * Lane -> chain as planned in BatchDev::seed_order (code and carrier chains never share a wavefront).
 * PASS: 0 = every block starts from its descriptor's carr_phase (independent blocks, or seeds resolved by the
 * host); 1 = pass A of the device-side carrier chain: carrier chains only, from the rough start phases, end
 * states only (ChainAux::endA); 2 = pass B: all chains, carriers from the refined start phases, rows, margins
 * and the place of the first wrap (see k_chain_fix); 3 = pass B where only the blocks' exact start phases are wanted
 * (the per-sample kernel's pre-pass follows and sees independent blocks): carrier chains only, no rows.
 */
template <int PASS>
__global__ __launch_bounds__(GPSBB_WALK_WG) void k_walk(BatchDev p)
{
    __builtin_amdgcn_s_setprio(GPSBB_SEED_PRIO);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = gid < p.seed_lanes ? p.seed_order[gid] : -1;
    const int nbc = p.nblocks * p.nch;
    const bool is_code = c >= 0 && c < nbc, is_carr = c >= nbc;
    if (PASS != 1 && PASS != 3 && __ballot(is_code)) {
        const int k = is_code ? c : 0;
        const gpsbb_chan_t &ch = p.ch[k];
        const bool on = is_code && ch.prn > 0;
        /* plutogpssim.c:2709: f_code * delt, rounded on its own */
        WalkLane<NCO_CODE> w = walk_lane<NCO_CODE>(p, k, ch.code_phase, mul_rn(ch.f_code, p.delt), on);
        w.nav = nav_pack(ch.icode, ch.ibit, ch.iword);
        w.dwrd = ch.dwrd;
        w.bits = on ? walk_dbits(ch.dwrd, w.nav, nav_bit(ch.dwrd, w.nav) < 0 ? 1u : 0u) : 0u;
        walk_both_signs<NCO_CODE, false>(w, p.nsamp, p.hazards, p.status);
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
        const int k = is_carr ? c - nbc : 0;
        const gpsbb_chan_t &ch = p.ch[k];
        const bool on = is_carr && ch.prn > 0;
        const double x0 = PASS == 1 ? p.aux[k].start0 : (PASS >= 2 ? p.aux[k].start1 : ch.carr_phase);
        WalkLane<NCO_CARR> w = walk_lane<NCO_CARR>(p, nbc + k, x0, mul_rn(ch.f_carr, p.delt) /* c:2741 */, on);
        w.aux = PASS >= 2 ? &p.aux[k] : nullptr;
        if (PASS >= 2)
            walk_both_signs<NCO_CARR, true, PASS == 2>(w, p.nsamp, p.hazards, p.status);
        else
            walk_both_signs<NCO_CARR, false, PASS != 1>(w, p.nsamp, p.hazards, p.status);
        if (is_carr) {
            if (PASS == 1) {
                p.aux[k].endA = on ? w.x : 0.0;
            } else {
                if (PASS != 3) {
                    p.row_cnt[nbc + k] = on ? (int32_t)(w.cnt < w.cap ? w.cnt : w.cap) : 0;
                    p.end[k].carr_phase = on ? w.x : 0.0;
                }
                if (PASS >= 2) {
                    /* the trajectory walked here is not final: k_chain_fix decides what counts */
                    ChainAux &a = p.aux[k];
                    a.margin = w.margin;
                    a.ncross = on ? w.ncross : 0;
                    if (!w.wrap_seen)
                        a.wrap_row = -1;
                    a.prefix_cnt = 0;
                    a.endB = on ? w.x : 0.0;
                    p.aux[k].hz512 = on ? w.hz512 : 0u;
                } else if (on && w.hz512) {
                    atomicAdd(p.hazards, (unsigned long long)w.hz512);
                }
            }
        }
    }
}

/* does block b of channel i continue block b-1's carrier (GPSBB_CHAIN_CARRIER: same channel index, same prn)? */
__device__ __forceinline__ bool chain_continues(const BatchDev &p, int b, int i)
{
    const int prn = p.ch[(size_t)b * p.nch + i].prn;
    if (b == 0) /* a stream's push continues the push before it */
        return p.carry && prn > 0 && ((p.cont0_mask >> i) & 1u);
    return prn > 0 && prn == p.ch[(size_t)(b - 1) * p.nch + i].prn;
}

/*
 * Device-side carrier chain, step 2 of 4: start phases good to a few units in the last place from pass A.
 * The walk of block b from the rough start0 ended at endA; from a start that is e = start1 - start0 away it
 * ends, up to a handful of roundings, e away from there (a trajectory is translated by a small change of its
 * start phase: see k_chain_fix), and there the next block begins: e[b+1] = e[b] + (endA[b] - start0[b+1]),
 * restarting from 0 where a block does not continue the one before.  One wavefront per channel: a segmented
 * inclusive scan over the blocks, 64 at a time.
 */
__global__ void k_chain_prefix(BatchDev p)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= p.nch)
        return;
    double carry = 0.0; /* e of the block before the chunk's first */
    for (int b0 = 0; b0 < p.nblocks; b0 += 64) {
        const int b = b0 + lane;
        const bool in = b < p.nblocks;
        const size_t k = (size_t)(in ? b : 0) * p.nch + i;
        const bool cont = in && chain_continues(p, b, i);
        /* the increment this block adds to the correction of the block before it: the phase lost between the end
         * of pass A's walk of block b-1 and the rough start of block b (a wrap of the unit interval apart at most) */
        double c = 0.0;
        if (cont) {
            c = (b == 0 ? p.carry->approx_end[i] : p.aux[k - p.nch].endA) - p.aux[k].start0;
            c = c > 0.5 ? c - 1.0 : (c < -0.5 ? c + 1.0 : c);
        }
        /* segmented inclusive scan: (flag, value) pairs, flag = the sum restarts here */
        double v = c;
        bool f = !cont;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double vu = __shfl_up(v, o);
            const int fu = __shfl_up((int)f, o);
            if (lane >= o) {
                v = f ? v : v + vu;
                f = f || fu;
            }
        }
        const double e = f ? v : v + carry;
        if (in) {
            ChainAux &a = p.aux[k];
            double st = cont ? a.start0 + e : p.ch[k].carr_phase;
            st = st >= 1.0 ? st - 1.0 : (st < 0.0 ? st + 1.0 : st);
            a.start1 = st;
            if (p.carry && b == p.nblocks - 1) {
                /* where the stream's next push will start, as far as pass A can tell */
                double en = a.endA + (st - a.start0);
                en = en >= 1.0 ? en - 1.0 : (en < 0.0 ? en + 1.0 : en);
                p.carry->approx_end[i] = en;
            }
        }
        carry = __shfl(e, 63);
    }
}

/* an exact walk of which only the end state and the hazard count are wanted */
struct FixNullSink {
    uint32_t hz512;
    __device__ __forceinline__ void row(int32_t, uint32_t, double, double, bool) {}
    __device__ __forceinline__ void table_index_512() { hz512++; }
    __device__ __forceinline__ void nav_fetch(uint32_t) {}
};

/* rows of one exact walk of a whole block, as k_walk writes them (build_rows_f64 drives it) */
struct FixRowSink {
    WalkRow *rows;
    uint32_t cap, cnt;
    bool overflow;
    uint32_t hz512;
    __device__ __forceinline__ void row(int32_t n0, uint32_t, double x, double, bool)
    {
        if (cnt < cap) {
            WalkRow r;
            r.n0 = n0;
            r.nav = 0;
            r.x = x;
            rows[cnt] = r;
        } else {
            overflow = true;
        }
        cnt++;
    }
    __device__ __forceinline__ void table_index_512() { hz512++; }
    __device__ __forceinline__ void nav_fetch(uint32_t) {}
};

/*
 * Device-side carrier chain, step 4 of 4: make it exact.  One lane per channel, blocks in order.
 *
 * Pass B walked block b from start1, a few units in the last place away from the true start phase (the end of
 * block b-1, known exactly only now).  Inside one of pass B's rows all states lie in one binade, on one grid,
 * and every step adds the same multiple of it: the true trajectory is pass B's plus an offset d that is a
 * multiple of that grid, for as long as the two stay in the same binade for the same steps — which pass B's
 * margin (how close its rows' first and last states come to a binade edge) guarantees when |d| is smaller.
 * d can only change where a sum is rounded on a COARSER grid: a binade crossed upwards, or a wrap.  Once a sum
 * has been rounded on the coarsest grid there is (the one before a wrap: 2^-52 in [1,2) for a rising phase,
 * 2^-53 in [0.5,1) for a falling one) d is a multiple of every grid the phase will ever meet and stays put —
 * except at a later rounding on that coarsest grid that is an exact TIE while d is an odd number of its steps
 * (see below: a rising phase's step decides whether its wrap sums can tie at all; a falling phase's "+ 1.0"
 * ties or not depending on the state's low bits, so pass B tests every wrap and records the ties as well).  Pass B recorded those few rows; here, for each of them in turn: pass B's state
 * at the last sample of the row before, plus d, is the true state there; one genuine IEEE step (c:2741-2746)
 * gives the true first state of the row; minus pass B's, that is the new d.  k_tiles adds the offsets to the
 * rows' states, the end state gets the last one.  A block whose step can tie on the coarsest grid, is tiny,
 * has more crossings than the record holds, or whose margin is not larger than its offsets is walked exactly
 * by the lane on its own (rare, slow).
 */
/* what one turn of k_chain_fix's loop reads: fetched a block ahead, because the loop itself is one long chain of
 * dependent arithmetic (each block starts where the one before ended) and must not wait for memory as well */
constexpr int FIX_PREFETCH_CROSS = 6;
struct FixIn {
    int prn, prn_prev, ncross, wrap_row;
    uint32_t hz512;
    double carr_phase, f_carr, start1, margin, endB, wrap_x;
    double pre[FIX_PREFETCH_CROSS], post[FIX_PREFETCH_CROSS]; /* the first crossings (a block has about five) */
};
__device__ __forceinline__ FixIn fix_load(const BatchDev &p, int b, int i)
{
    const size_t k = (size_t)b * p.nch + i;
    const gpsbb_chan_t &ch = p.ch[k];
    const ChainAux &a = p.aux[k];
    FixIn f;
    f.prn = ch.prn;
    f.prn_prev = b > 0 ? p.ch[k - p.nch].prn : 0;
    f.carr_phase = ch.carr_phase;
    f.f_carr = ch.f_carr;
    f.start1 = a.start1;
    f.margin = a.margin;
    f.endB = a.endB;
    f.ncross = a.ncross;
    f.wrap_row = a.wrap_row;
    f.wrap_x = a.wrap_x;
    f.hz512 = a.hz512;
#pragma unroll
    for (int j = 0; j < FIX_PREFETCH_CROSS; j++) {
        f.pre[j] = a.pre[j];
        f.post[j] = a.post[j];
    }
    return f;
}

__global__ __launch_bounds__(64) void k_chain_fix(BatchDev p)
{
#ifndef GPSBB_EXP_NOPRIO
    /* one wavefront on whose latency every later push of the stream waits: let it win the issue arbitration
     * against the synthesis wavefronts it shares its SIMD with */
    __builtin_amdgcn_s_setprio(3);
#endif
    const int i = threadIdx.x;
    const bool lane_on = i < p.nch;
    const int il = lane_on ? i : 0;
    const int nbc = p.nblocks * p.nch;
    double prev_end = p.carry ? p.carry->exact_end[il] : 0.0; /* a stream: where the push before this one ended */
    unsigned long long n_fallback = 0, n_hz = 0;
    /* two blocks ahead: a turn of this loop is a few hundred cycles of dependent arithmetic, a load from HBM beside
     * the other kernels takes longer than that */
    FixIn nxt = fix_load(p, 0, il);
    FixIn nxt2 = fix_load(p, p.nblocks > 1 ? 1 : 0, il);
    for (int b = 0; b < p.nblocks; b++) {
        const FixIn in = nxt;
        nxt = nxt2;
        nxt2 = fix_load(p, b + 2 < p.nblocks ? b + 2 : p.nblocks - 1, il);
        const size_t k = (size_t)b * p.nch + il;
        const bool on = lane_on && in.prn > 0;
        ChainAux &a = p.aux[k];
        const bool cont = lane_on && in.prn > 0 && (b > 0 ? in.prn == in.prn_prev : (p.carry && ((p.cont0_mask >> il) & 1u)));
        const double x = cont ? prev_end : in.carr_phase;
        const double s = mul_rn(in.f_carr, p.delt);
        const uint64_t sb = f64_bits(s);
        const int es = (int)((sb >> 52) & 0x7ff);
        const double margin = in.margin;
        const int ncross = in.ncross;
        double end = in.endB;
        uint32_t hz512 = in.hz512; /* pass B's trajectory is the true one, or a translate that met no edge */
        const double d0 = x - in.start1; /* exact: both in the same binade, or the margin test below fails */
        /* Ties.  A sum exactly half-way between two grid points goes to the even one, so a tie commutes with the
         * shift only if the shift is an even number of steps of that grid.  It always is on grids finer than the
         * one the offset was last rounded on.  That leaves (1) the binades the phase visits before its offset has
         * been through the coarsest grid — from its start binade upwards if it rises, its start binade only if it
         * falls: if the step can tie there (walk_tiemask) the lane walks the block's first lap on its own;
         * (2) the coarsest grid itself: pass B finds the first tie there after the first wrap (a wrap sum exactly
         * half-way; for a falling phase whose step ties on the top binade's grid, the first step after the wrap) and
         * records it as a crossing: both trajectories leave a tie with an even mantissa, so the offset is an even
         * number of grid steps from there on and no later tie can change it. */
        const bool fall = s < 0.0;
        bool tie_top = true;
        {
            const int dt = (fall ? 1022 : 1023) - es; /* the step's last place is 2^dt times finer than that grid */
            if (dt >= 1 && dt <= 52) {
                const uint64_t low = ((sb & F64_MANT) | F64_HID) & ((1ull << dt) - 1);
                tie_top = low == 0ull || low == (1ull << (dt - 1));
            } else if (dt <= 0) {
                tie_top = false; /* the step is a multiple of the grid: sums are never between grid points */
            }
        }
        bool tie_asc = false;
        if (on && d0 != 0.0 && es >= 123) {
            const int dstart = (int)((f64_bits(x) >> 52) & 0x7ff) - es;
            const int dtie = walk_tie_d(sb); /* the one binade (above the step's) in which the step ties */
            tie_asc = dtie >= 2 && dtie <= 50 && (fall ? dtie == dstart : dtie >= dstart);
        }
        const double gtop = fall ? 0x1p-53 : 0x1p-52;
        const bool base = on && es >= 123 && ncross >= 0 && fabs(d0) < margin - 0x1p-51;
        bool ok = on && d0 == 0.0;
        double d = d0;
        if (on)
            a.seg[0] = d0;
        if (base && !ok && !tie_asc) {
            /* the usual way: one genuine step per recorded crossing */
            ok = true;
            for (int j = 0; ok && j < ncross; j++) {
                /* pass B's state at the last sample before the crossing, the true one, one genuine step */
                double pre_j = 0.0, post_j = 0.0;
                if (j >= FIX_PREFETCH_CROSS) { /* rare: beyond what was fetched ahead */
                    pre_j = a.pre[j];
                    post_j = a.post[j];
                } else {
#pragma unroll
                    for (int q = 0; q < FIX_PREFETCH_CROSS; q++) {
                        pre_j = j == q ? in.pre[q] : pre_j;
                        post_j = j == q ? in.post[q] : post_j;
                    }
                }
                double xt = pre_j + d;
                carr_step(xt, s);
                d = xt - post_j;
                ok = fabs(d) < margin - 0x1p-51;
                a.seg[j + 1] = d;
            }
        } else if (base && !ok && in.wrap_row >= 0) {
            /* a tie-prone binade on the way up: the first lap exactly, on its own rows; after the wrap that ends it
             * the offset is a multiple of every grid and the rest of the block is pass B's plus it */
            const int nstar = in.wrap_row; /* a wrap always starts a row */
            FixRowSink sink;
            sink.rows = p.chain_starts ? nullptr : reinterpret_cast<WalkRow *>(p.prefix_rows) + k * CHAIN_PREFIX_CAP;
            sink.cap = p.chain_starts ? 0 : CHAIN_PREFIX_CAP;
            sink.cnt = 0;
            sink.overflow = false;
            sink.hz512 = 0;
            uint32_t nav = 0;
            double xs;
            if (p.chain_starts) { /* only the state at the wrap is wanted */
                FixNullSink ns;
                ns.hz512 = 0;
                xs = build_rows_f64<NCO_CARR>(x, s, nav, nstar, ns);
                sink.hz512 = ns.hz512;
            } else {
                xs = build_rows_f64<NCO_CARR>(x, s, nav, nstar, sink);
            }
            d = xs - in.wrap_x;
            /* (a tie-prone coarsest grid and an odd offset there: pass B's recorded tie is not part of this path) */
            ok = !sink.overflow && sink.hz512 == 0 && fabs(d) < margin - 0x1p-51 &&
                 !(tie_top && fmod(fabs(d) * (fall ? 0x1p+53 : 0x1p+52), 2.0) != 0.0);
            if (ok) {
                /* pass B's rows from the wrap on: its offset there, then — as on the usual way — one genuine step
                 * through every crossing pass B recorded after it (ties at later wraps, the block's last step);
                 * the list is compacted in place (entry m is written after entry j >= m has been read) */
                const double d_wrap = d;
                int m = 1;
                for (int j = 0; ok && j < ncross; j++) {
                    const int cj = a.cross[j];
                    if (cj <= nstar)
                        continue;
                    double xt = a.pre[j] + d;
                    carr_step(xt, s);
                    d = xt - a.post[j];
                    ok = fabs(d) < margin - 0x1p-51;
                    a.cross[m] = cj;
                    a.seg[m + 1] = d;
                    m++;
                }
                if (ok) {
                    a.prefix_cnt = (int32_t)sink.cnt;
                    a.prefix_end = nstar;
                    a.ncross = m;
                    a.cross[0] = nstar;
                    a.seg[0] = 0.0;
                    a.seg[1] = d_wrap;
                }
            }
        }
        (void)gtop;
        if (on && ok && d0 != 0.0)
            end = end + d; /* exact: the true end state is a double */
#ifdef GPSBB_CHAIN_DEBUG
        if (on && p.nch == 1) {
            printf("fix b %d x %.17g start1 %.17g d0 %.3e margin %.3e ncross %d wrap_row %d tie_top %d tie_asc %d ok %d endB %.17g end %.17g s %.17g\n", b, x,
                   in.start1, d0, margin, ncross, in.wrap_row, (int)tie_top, (int)tie_asc, (int)ok, in.endB, end, s);
            for (int j = 0; j < ncross && j < CHAIN_MAX_CROSS; j++)
                printf("    cross %d sample %d pre %.17g post %.17g seg %.3e\n", j, a.cross[j], a.pre[j], a.post[j], a.seg[j + 1]);
        }
        if (on && !ok)
            printf("chain fallback: block %d ch %d d0 %.3e d %.3e margin %.3e ncross %d es %d tie_top %d tie_asc %d wrap_row %d\n", b,
                   i, d0, d, margin, ncross, es, (int)tie_top, (int)tie_asc, in.wrap_row);
#endif
        if (on && !ok && p.chain_starts) {
            /* on its own: the whole block exactly; only its end state is wanted */
            FixNullSink ns;
            ns.hz512 = 0;
            uint32_t nav = 0;
            end = build_rows_f64<NCO_CARR>(x, s, nav, p.nsamp, ns);
            n_fallback++;
        } else if (on && !ok) {
            /* on its own: the whole block exactly, rows in place of pass B's, no offsets */
            FixRowSink sink;
            sink.rows = reinterpret_cast<WalkRow *>(p.rows) + p.row_off[nbc + k];
            sink.cap = (uint32_t)(p.row_off[nbc + k + 1] - p.row_off[nbc + k]);
            sink.cnt = 0;
            sink.overflow = false;
            sink.hz512 = 0;
            uint32_t nav = 0;
            end = build_rows_f64<NCO_CARR>(x, s, nav, p.nsamp, sink);
            hz512 = sink.hz512;
            if (sink.overflow)
                atomicOr(p.status, ST_ROW_OVERFLOW);
            p.row_cnt[nbc + k] = (int32_t)(sink.cnt < sink.cap ? sink.cnt : sink.cap);
            a.ncross = 0;
            a.seg[0] = 0.0;
            a.prefix_cnt = 0;
            n_fallback++;
        }
        if (lane_on && p.chain_starts) {
            /* the per-sample kernel's pre-pass comes next: all it needs is where this block starts (it counts the
             * hazards and writes the end states itself) */
            if (on && cont)
                const_cast<gpsbb_chan_t *>(p.ch)[k].carr_phase = x;
            prev_end = end;
        } else if (lane_on) {
            p.end[k].carr_phase = on ? end : 0.0;
            prev_end = end;
            if (on)
                n_hz += hz512;
        }
    }
    if (p.carry && lane_on)
        p.carry->exact_end[i] = prev_end;
    if (n_hz)
        atomicAdd(p.hazards, n_hz);
    if (n_fallback)
        atomicAdd(p.hazards + 4, n_fallback);
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
    /* the tile counters of this table set, for the synthesis kernel that follows (the one that last used them has
     * finished: the pre-pass waited for it): saves a memset and its launch gap on the synthesis stream */
    if (chain < p.nblocks && threadIdx.x == 0)
        p.tile_ctr[chain] = 0;
    const int cnt = p.row_cnt[chain];
    if (cnt <= 0)
        return;
    const WalkRow *__restrict__ rows = reinterpret_cast<const WalkRow *>(p.rows) + p.row_off[chain];
    const int b = bi / p.nch, i = bi % p.nch;
    /* the chain's step (c:2709 / c:2741), from which every row's increment follows (walk_row_step) */
    const double s = kind ? mul_rn(p.ch[bi].f_carr, p.delt) : mul_rn(p.ch[bi].f_code, p.delt);
    double *__restrict__ tx = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + kind) * (size_t)p.ntiles;
    uint32_t *__restrict__ tn = p.tile_nav + ((size_t)b * (size_t)p.nch + i) * (size_t)p.ntiles;
    /* carrier chained on the device: the true states are pass B's plus an offset per stretch of rows (k_chain_fix) */
    const bool shifted = kind && p.aux && p.chain_dev;
    const ChainAux *aux = shifted ? &p.aux[bi] : nullptr;
    const int ncross = shifted ? (aux->ncross > 0 ? aux->ncross : 0) : 0;
    /* k_chain_fix walked the first lap on its own: those rows (in the chain's prefix region) hold the samples
     * before prefix_end, pass B's rows before wrap_row are void */
    const int prefix_cnt = shifted ? aux->prefix_cnt : 0;
    const int void_before = prefix_cnt > 0 ? aux->prefix_end : 0; /* pass B's rows before this sample are void */
    if (prefix_cnt > 0) {
        const WalkRow *pr = reinterpret_cast<const WalkRow *>(p.prefix_rows) + (size_t)bi * CHAIN_PREFIX_CAP;
        const int pend = aux->prefix_end;
        for (int r = threadIdx.x; r < prefix_cnt; r += blockDim.x) {
            const WalkRow row = pr[r];
            const int n_next = r + 1 < prefix_cnt ? pr[r + 1].n0 : pend;
            int t = (int)(((uint32_t)row.n0 + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            int t_end = (int)(((uint32_t)n_next + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            t_end = t_end < p.ntiles ? t_end : p.ntiles;
            for (; t < t_end; t++)
                tx[t] = mul_rn(__fma_rn((double)(t * TILE - row.n0), walk_row_step(row.x, s), row.x), 512.0);
        }
    }
    /* four rows per lane and turn, their loads issued together: the kernel is bound by the latency of these
     * loads, not by their number */
    constexpr int U = 4;
    for (int r0 = threadIdx.x; r0 < cnt; r0 += U * GPSBB_TILES_WG) {
        WalkRow row[U];
        int n_next[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int r = r0 + j * GPSBB_TILES_WG;
            const int rc = r < cnt ? r : cnt - 1;
            row[j] = rows[rc];
            n_next[j] = rc + 1 < cnt ? rows[rc + 1].n0 : INT32_MAX - TILE;
            if (r >= cnt || row[j].n0 < void_before)
                n_next[j] = row[j].n0; /* past the chain's last row, or replaced by the prefix rows: no tiles */
        }
        double off[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            off[j] = 0.0;
            if (shifted) {
                int g = 0; /* the row's segment: the crossings at or before its first sample */
                while (g < ncross && aux->cross[g] <= row[j].n0)
                    g++;
                off[j] = aux->seg[g];
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            int t = (int)(((uint32_t)row[j].n0 + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            int t_end = (int)(((uint32_t)n_next[j] + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            t_end = t_end < p.ntiles ? t_end : p.ntiles;
            const double S = walk_row_step(row[j].x, s);
            for (; t < t_end; t++) {
                double v = __fma_rn((double)(t * TILE - row[j].n0), S, row[j].x);
                v = shifted ? v + off[j] : v; /* exact: the sum is the true state, a double */
                tx[t] = kind ? mul_rn(v, 512.0) : v;
                if (!kind)
                    tn[t] = row[j].nav;
            }
        }
    }
}

} /* namespace gpsbb_impl */
#endif
