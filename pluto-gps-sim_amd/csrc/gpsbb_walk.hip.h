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
template <int KIND, bool STORE = true>
__device__ __forceinline__ WalkLane<KIND> walk_lane(const BatchDev &p, int chain, double x0, double s, bool on)
{
    WalkLane<KIND> w;
    w.x = x0;
    w.s = s;
    w.n = 0;
    w.nav = 0;
    w.bits = 0;
    w.dwrd = nullptr;
    w.rows = nullptr;
    w.cap = 0;
    if (STORE) { /* walks that keep no rows (pass A, the chain-only pass B) have no region of the pool */
        const uint64_t o0 = p.row_off[chain], o1 = p.row_off[chain + 1];
        w.rows = reinterpret_cast<WalkRow *>(p.rows) + o0;
        w.cap = (uint32_t)(o1 - o0);
    }
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
    const int nbc = p.nblocks * p.nch;
    int c = -1;
    if (gid < p.seed_lanes) {
        if (p.seed_order) {
            c = p.seed_order[gid];
        } else {
            /* no plan (chain-only runs over very many blocks): the carrier chains channel by channel, blocks in order — 64
             * consecutive blocks of one channel share a wavefront, and a stream continuous in time gives them the same
             * direction and almost the same |f_carr|, which is all the plan's sort is for */
            const int nbp = (p.nvb + 63) & ~63;
            const int i = gid / nbp, b = gid - i * nbp;
            c = (i < p.nch && b < p.nvb) ? nbc + b * p.nch + i : -1;
        }
    }
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
        /* k: the chain's segment (PASS >= 1: virtual block * nch + channel) or block (PASS 0: blocks are not cut) */
        const int k = is_carr ? c - nbc : 0;
        /* the passes of the device-side chain read the 24-byte chain descriptors, the others the descriptors themselves */
        const int prn = PASS >= 1 ? p.cd[k].prn : p.ch[k].prn;
        const double f_carr = PASS >= 1 ? p.cd[k].f_carr : p.ch[k].f_carr;
        /* fixed-point carrier (only ever with PASS 0): nothing to walk, the phase is linear in the sample number */
        const bool fx = PASS == 0 && p.kph0 != nullptr;
        const bool on = is_carr && prn > 0 && !fx;
        const double x0 = PASS == 1 ? p.start0[k] : (PASS >= 2 ? (p.model_start ? p.start0[k] : p.aux[k].start1) : p.ch[k].carr_phase);
        const int vb = k / p.nch, sgi = PASS >= 1 ? vb % p.nseg : 0;
        const int ns = PASS >= 1 ? seg_nsamp(p, sgi) : p.nsamp; /* lanes of a wavefront may walk segments of different length */
        WalkLane<NCO_CARR> w = walk_lane<NCO_CARR, PASS == 0 || PASS == 2>(p, nbc + k, x0, mul_rn(f_carr, p.delt) /* c:2741 */, on);
        w.aux = PASS >= 2 ? &p.aux[k] : nullptr;
        if (PASS >= 2)
            walk_both_signs<NCO_CARR, true, PASS == 2>(w, ns, p.hazards, p.status);
        else
            walk_both_signs<NCO_CARR, false, PASS != 1>(w, ns, p.hazards, p.status);
        if (is_carr) {
            if (PASS == 1) {
                p.aux[k].endA = on ? w.x : 0.0;
            } else {
                if (PASS != 3) {
                    p.row_cnt[nbc + k] = on ? (int32_t)(w.cnt < w.cap ? w.cnt : w.cap) : 0;
                    if (PASS == 0)
                        p.end[k].carr_phase = fx ? (prn > 0 ? (double)(uint32_t)(p.kph0[k] + (uint32_t)p.nsamp * (uint32_t)p.kstep[k]) : 0.0)
                                                 : (on ? w.x : 0.0); /* (chained: the fix-up writes the blocks' true end phases) */
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
    const ChainDesc &d = p.cd[(size_t)b * p.nch + i];
    if (b == 0) /* a stream's push continues the push before it */
        return p.carry && d.prn > 0 && ((p.cont0_mask >> i) & 1u);
    return d.prn > 0 && !d.start && d.prn == p.cd[(size_t)(b - 1) * p.nch + i].prn;
}

/*
 * Device-side carrier chain, step 2 of 4: start phases good to a few units in the last place from pass A.
 * The walk of block b from the rough start0 ended at endA; from a start that is e = start1 - start0 away it
 * ends, up to a handful of roundings, e away from there (a trajectory is translated by a small change of its
 * start phase: see k_chain_fix), and there the next block begins: e[b+1] = e[b] + (endA[b] - start0[b+1]),
 * restarting from 0 where a block does not continue the one before.  One wavefront per channel: a segmented
 * inclusive scan over the blocks, 64 at a time.
 */
constexpr int PREFIX_WG = 256;
__global__ __launch_bounds__(PREFIX_WG) void k_chain_prefix(BatchDev p)
{
    __shared__ double wv[PREFIX_WG / 64];
    __shared__ int wf[PREFIX_WG / 64];
    __shared__ double carry_s;
    __builtin_amdgcn_s_setprio(3); /* see k_chain_fix_par */
    const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (i >= p.nch)
        return;
    double carry = 0.0; /* e of the block before the chunk's first */
    for (int b0 = 0; b0 < p.nvb; b0 += PREFIX_WG) { /* b: a virtual block (segment) */
        const int b = b0 + t;
        const bool in = b < p.nvb;
        const size_t k = (size_t)(in ? b : 0) * p.nch + i;
        const bool cont = in && chain_continues(p, b, i);
        /* the increment this block adds to the correction of the block before it: the phase lost between the end
         * of pass A's walk of block b-1 and the rough start of block b (a wrap of the unit interval apart at most) */
        const double start0 = in ? p.start0[k] : 0.0;
        double c = 0.0;
        if (cont) {
            c = (b == 0 ? p.carry->approx_end[i] : p.aux[k - p.nch].endA) - start0;
            c = c > 0.5 ? c - 1.0 : (c < -0.5 ? c + 1.0 : c);
        }
        /* segmented inclusive scan: (flag, value) pairs, flag = the sum restarts here; inside the wavefront ... */
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
        /* ... and across the wavefronts of the workgroup */
        if (lane == 63) {
            wv[wave] = v;
            wf[wave] = f ? 1 : 0;
        }
        __syncthreads();
        for (int w = wave - 1; w >= 0 && !f; w--) {
            v += wv[w];
            f = wf[w] != 0;
        }
        const double e = f ? v : v + carry;
        if (in) {
            ChainAux &a = p.aux[k];
            double st = cont ? start0 + e : p.cd[k].carr_phase;
            st = st >= 1.0 ? st - 1.0 : (st < 0.0 ? st + 1.0 : st);
            a.start1 = st;
            if (p.carry && b == p.nvb - 1) {
                /* where the stream's next push will start, as far as pass A can tell */
                double en = a.endA + (st - start0);
                en = en >= 1.0 ? en - 1.0 : (en < 0.0 ? en + 1.0 : en);
                p.carry->approx_end[i] = en;
            }
        }
        if (t == PREFIX_WG - 1)
            carry_s = e;
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}

/* the rows of an exact walk as k_walk writes them (build_rows_f64 drives it); STORE = false: the walk is only wanted
 * for its end state, its row count and its hazards (a trial evaluation, or a batch that keeps no rows) */
template <bool STORE>
struct FixRowSink {
    WalkRow *rows;
    uint32_t cap, cnt;
    bool overflow;
    uint32_t hz512;
    __device__ __forceinline__ void row(int32_t n0, uint32_t, double x, double, bool)
    {
        if (cnt < cap) {
            if (STORE) {
                WalkRow r;
                r.n0 = n0;
                r.nav = 0;
                r.x = x;
                rows[cnt] = r;
            }
        } else {
            overflow = true;
        }
        cnt++;
    }
    __device__ __forceinline__ void table_index_512() { hz512++; }
    __device__ __forceinline__ void nav_fetch(uint32_t) {}
};
/*
 * Device-side carrier chain, step 4 of 4: make it exact.
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
 * (see below: a rising phase's step decides whether its wrap sums can tie at all; a falling phase's "+ 1.0" ties or
 * not depending on the state's low bits, so pass B tests every wrap and records the ties as well).  Pass B recorded
 * those few rows; fix_block, for each of them in turn: pass B's state at the last sample of the row before, plus d,
 * is the true state there; one genuine IEEE step (c:2741-2746) gives the true first state of the row; minus pass
 * B's, that is the new d.  k_tiles adds the offsets to the rows' states, the end state gets the last one.  A block
 * whose step can tie on the coarsest grid, is tiny, has more crossings than the record holds, or whose margin is
 * not larger than its offsets is walked exactly by the lane on its own (rare, slow).
 *
 * fix_block is that computation for ONE block given its true start phase x: a pure function of (pass B's record of
 * the block, x).  Two kernels drive it: k_chain_fix (one lane per channel, the blocks in order — each block's x is
 * the end the block before just returned) and k_chain_fix_par (one workgroup per channel, one lane per block — the
 * x of all blocks guessed at once, then checked by the same function; see there).
 */
/* what fix_block reads of a block: fetched ahead of the arithmetic */
constexpr int FIX_PREFETCH_CROSS = 6;
struct FixIn {
    int prn, prn_prev, ncross, wrap_row; /* prn_prev: 0 where the block starts a chain of its own (ChainDesc::start) */
    uint32_t hz512;
    double carr_phase, f_carr, start1, margin, endB, wrap_x;
    double pre[FIX_PREFETCH_CROSS], post[FIX_PREFETCH_CROSS]; /* the first crossings (a block has about five) */
};
__device__ __forceinline__ FixIn fix_load(const BatchDev &p, int b, int i)
{
    const size_t k = (size_t)b * p.nch + i;
    const ChainDesc &ch = p.cd[k];
    const ChainAux &a = p.aux[k];
    FixIn f;
    f.prn = ch.prn;
    f.prn_prev = (b > 0 && !ch.start) ? p.cd[k - p.nch].prn : 0;
    f.carr_phase = ch.carr_phase;
    f.f_carr = ch.f_carr;
    f.start1 = p.model_start ? p.start0[k] : a.start1; /* what pass B walked from */
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

/* The exact walks of the rare paths, out of line: fix_block is instantiated many times over (k_chain_fix_par tries
 * start phases with it) and must stay small.  STORE: rows are written (room for `cap` of them at `rows`), else only
 * counted.  Returns the state after n steps. */
struct FixWalkOut {
    double x;
    uint32_t cnt, hz512;
    bool overflow;
};
template <bool STORE>
__device__ __noinline__ FixWalkOut fix_walk(double x, double s, int n, WalkRow *rows, uint32_t cap)
{
    FixRowSink<STORE> sink;
    sink.rows = rows;
    sink.cap = cap;
    sink.cnt = 0;
    sink.overflow = false;
    sink.hz512 = 0;
    uint32_t nav = 0;
    FixWalkOut o;
    o.x = build_rows_f64<NCO_CARR>(x, s, nav, n, sink);
    o.cnt = sink.cnt;
    o.hz512 = sink.hz512;
    o.overflow = sink.overflow;
    return o;
}

constexpr int FIX_OK = 0;   /* FixOut::end is the block's true end phase */
constexpr int FIX_SLOW = 1; /* only a walk of the whole block tells (and the caller did not ask for one) */
struct FixOut {
    double end;
    int kind;
    uint32_t hz512;  /* samples of the block whose phase is exactly 1.0 */
    bool walked;     /* the whole block was walked on its own (GPSBB_INFO_CHAIN_FALLBACKS) */
};

/*
 * One block (k = block * nch + channel, `on`: the channel is active in it) from its true start phase x.
 * COMMIT: leave what k_tiles needs (the offsets per stretch of rows, the rows of a lap or a block walked here) —
 * without it nothing is written, so that a start phase that is only a guess can be tried.  run_slow: walk the whole
 * block if nothing cheaper settles it (else FIX_SLOW comes back and the caller calls again once x is known to be true).
 */
template <bool COMMIT>
__device__ __forceinline__ FixOut fix_block(const BatchDev &p, const FixIn &in, size_t k, bool on, double x, bool run_slow)
{
    /* k = virtual block * nch + channel: a "block" here is a segment (BatchDev::nseg) */
    const int nsamp_seg = seg_nsamp(p, (int)((k / (size_t)p.nch) % (size_t)p.nseg));
    FixOut out;
    out.kind = FIX_OK;
    out.walked = false;
    ChainAux &a = p.aux[k];
    const int nbc = p.nblocks * p.nch;
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
     * falls: if the step can tie there (walk_tiemask) the block's first lap is walked here;
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
        /* (dtie == 1 included: one binade above the step's every sum is a tie — the walk takes those steps one by one, but
         * an offset that is an odd number of that binade's last places still flips them: found by the parity soak once the
         * start phases came from the drift model, tests/golden/chain_model_big_step_desc.npy) */
        tie_asc = dtie >= 1 && dtie <= 50 && (fall ? dtie == dstart : dtie >= dstart);
    }
    const bool base = on && es >= 123 && ncross >= 0 && fabs(d0) < margin - 0x1p-51;
    bool ok = on && d0 == 0.0;
    double d = d0;
    if (COMMIT && on) {
        a.seg[0] = d0;
        if (ok) /* pass B walked the true trajectory: no offset anywhere (the record is not zeroed between runs) */
            for (int j = 0; j < ncross; j++)
                a.seg[j + 1] = 0.0;
    }
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
            if (COMMIT)
                a.seg[j + 1] = d;
        }
    } else if (base && !ok && in.wrap_row >= 0) {
        /* a tie-prone binade on the way up: the first lap exactly, on its own rows; after the wrap that ends it
         * the offset is a multiple of every grid and the rest of the block is pass B's plus it */
        const int nstar = in.wrap_row; /* a wrap always starts a row */
        /* (a batch that keeps no rows, chain_starts: only the state at the wrap is wanted) */
        const FixWalkOut lap = (COMMIT && !p.chain_starts)
                                   ? fix_walk<true>(x, s, nstar, reinterpret_cast<WalkRow *>(p.prefix_rows) + k * CHAIN_PREFIX_CAP, CHAIN_PREFIX_CAP)
                                   : fix_walk<false>(x, s, nstar, nullptr, p.chain_starts ? 0x7fffffffu : (uint32_t)CHAIN_PREFIX_CAP);
        const bool overflow = lap.overflow;
        const uint32_t lap_rows = lap.cnt, lap_hz = lap.hz512;
        d = lap.x - in.wrap_x;
        /* (a tie-prone coarsest grid and an odd offset there: pass B's recorded tie is not part of this path) */
        ok = !overflow && lap_hz == 0 && fabs(d) < margin - 0x1p-51 &&
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
                if (COMMIT) {
                    a.cross[m] = cj;
                    a.seg[m + 1] = d;
                }
                m++;
            }
            if (ok && COMMIT) {
                a.prefix_cnt = (int32_t)lap_rows;
                a.prefix_end = nstar;
                a.ncross = m;
                a.cross[0] = nstar;
                a.seg[0] = 0.0;
                a.seg[1] = d_wrap;
            }
        }
    }
    if (on && ok && d0 != 0.0)
        end = end + d; /* exact: the true end state is a double */
#ifdef GPSBB_CHAIN_DEBUG
    if (COMMIT && on && p.nch == 1) {
        printf("fix k %d x %.17g start1 %.17g d0 %.3e margin %.3e ncross %d wrap_row %d tie_top %d tie_asc %d ok %d endB %.17g end %.17g s %.17g\n",
               (int)k, x, in.start1, d0, margin, ncross, in.wrap_row, (int)tie_top, (int)tie_asc, (int)ok, in.endB, end, s);
        for (int j = 0; j < ncross && j < CHAIN_MAX_CROSS; j++)
            printf("    cross %d sample %d pre %.17g post %.17g seg %.3e\n", j, a.cross[j], a.pre[j], a.post[j], a.seg[j + 1]);
    }
#endif
    if (on && !ok) {
        if (!run_slow) {
            out.kind = FIX_SLOW;
            out.end = end;
            out.hz512 = 0;
            return out;
        }
        out.walked = true;
        if (p.chain_starts) {
            /* on its own: the whole block exactly; only its end state is wanted */
            end = fix_walk<false>(x, s, nsamp_seg, nullptr, 0x7fffffffu).x;
        } else {
            /* on its own: the whole block exactly, rows in place of pass B's, no offsets */
            const uint32_t cap = (uint32_t)(p.row_off[nbc + k + 1] - p.row_off[nbc + k]);
            WalkRow *rows = reinterpret_cast<WalkRow *>(p.rows) + p.row_off[nbc + k];
            const FixWalkOut w = COMMIT ? fix_walk<true>(x, s, nsamp_seg, rows, cap) : fix_walk<false>(x, s, nsamp_seg, nullptr, cap);
            end = w.x;
            hz512 = w.hz512;
            if (COMMIT) {
                if (w.overflow)
                    atomicOr(p.status, ST_ROW_OVERFLOW);
                p.row_cnt[nbc + k] = (int32_t)(w.cnt < cap ? w.cnt : cap);
                a.ncross = 0;
                a.seg[0] = 0.0;
                a.prefix_cnt = 0;
            }
        }
    }
    out.end = end;
    out.hz512 = hz512;
    return out;
}

/* what a block leaves behind once its true start phase x and end are known (after fix_block<true>) */
__device__ __forceinline__ void fix_publish(const BatchDev &p, size_t k, bool on, bool cont, double x, const FixOut &r,
                                            unsigned long long &n_hz)
{
    const size_t vb = k / (size_t)p.nch, i = k % (size_t)p.nch;
    const size_t kb = (vb / (size_t)p.nseg) * (size_t)p.nch + i; /* the block the segment belongs to */
    const int sgi = (int)(vb % (size_t)p.nseg);
    if (p.chain_starts) {
        /* the per-sample kernel's pre-pass comes next (or nothing, gpsbb_chain_carrier): all that is wanted is where
         * every block starts (k_seed counts the hazards and writes the end states itself) */
        if (on && cont) {
            p.cd[k].carr_phase = x;
            if (p.ch && sgi == 0)
                const_cast<gpsbb_chan_t *>(p.ch)[kb].carr_phase = x;
        }
    } else {
        if (sgi == p.nseg - 1)
            p.end[kb].carr_phase = on ? r.end : 0.0;
        if (on)
            n_hz += r.hz512;
    }
}

/* The blocks in order, one lane per channel: each block starts where the one before ended.  Kept beside
 * k_chain_fix_par as the plain statement of the chain (GPSBB_OPT_CHAIN_WHERE 2; the tests run both). */
__global__ __launch_bounds__(64) void k_chain_fix(BatchDev p)
{
    /* one wavefront on whose latency every later push of the stream waits: let it win the issue arbitration
     * against the synthesis wavefronts it shares its SIMD with */
    __builtin_amdgcn_s_setprio(3);
    const int i = threadIdx.x;
    const bool lane_on = i < p.nch;
    const int il = lane_on ? i : 0;
    double prev_end = p.carry ? p.carry->exact_end[il] : 0.0; /* a stream: where the push before this one ended */
    unsigned long long n_fallback = 0, n_hz = 0;
    /* two blocks ahead: a turn of this loop is a few hundred cycles of dependent arithmetic, a load from HBM beside
     * the other kernels takes longer than that */
    FixIn nxt = fix_load(p, 0, il);
    FixIn nxt2 = fix_load(p, p.nvb > 1 ? 1 : 0, il);
    for (int b = 0; b < p.nvb; b++) { /* b: a virtual block (segment) */
        const FixIn in = nxt;
        nxt = nxt2;
        nxt2 = fix_load(p, b + 2 < p.nvb ? b + 2 : p.nvb - 1, il);
        const size_t k = (size_t)b * p.nch + il;
        const bool on = lane_on && in.prn > 0;
        const bool cont = on && (b > 0 ? in.prn == in.prn_prev : (p.carry && ((p.cont0_mask >> il) & 1u)));
        const double x = cont ? prev_end : in.carr_phase;
        const FixOut r = fix_block<true>(p, in, k, on, x, true);
        n_fallback += r.walked ? 1u : 0u;
        if (lane_on)
            fix_publish(p, k, on, cont, x, r, n_hz);
        prev_end = r.end;
    }
    if (p.carry && lane_on)
        p.carry->exact_end[i] = prev_end;
    if (n_hz)
        atomicAdd(p.hazards, n_hz);
    if (n_fallback)
        atomicAdd(p.hazards + 4, n_fallback);
}

/*
 * The same chain with the blocks in PARALLEL: one workgroup per channel, one lane per block (chunks of FIXP_WG blocks).
 *
 * The end of block b is fix_block(b, x_b) and x_(b+1) is that end: sequential as written.  But what a block does to
 * a start phase is almost a translation.  With u = 2^-53 (every grid a phase meets after a wrap is a multiple of it)
 * write the true end of block b as endB_b + o_b * u (endB: pass B's end).  For a block that wraps, shifting the start
 * by 4u shifts the end by exactly 4u (all roundings, ties included, commute with a shift by an even number of steps
 * of every grid — the coarsest is 2^-52), so
 *     o_b = o_(b-1) + k_b[o_(b-1) mod 4],
 * four small integers per block that every lane finds for its own block by running fix_block (no side effects) from the
 * four start phases endB_(b-1) + j*u.  Maps of that form compose to maps of that form: an inclusive scan over the
 * blocks gives every o_b at once.  A block that does not fit the form (no wrap, a start phase of its own, a lap or a
 * whole block to be walked) enters the scan as a constant — a guess.
 *
 * Nothing rests on that argument.  The guessed start phases are CHECKED by induction: every lane runs fix_block from
 * its guessed x_b and compares the end with the guess it handed to block b+1, bit for bit.  Up to the first block b*
 * where that fails (or that needs a whole-block walk) every x is the true one, hence x_(b*) is, hence its end is
 * (walked now if it has to be); that end enters the scan as a known constant and the blocks after b* are guessed and
 * checked again.  When no block fails, every x_b is what k_chain_fix would have passed on, and every lane commits
 * its block with fix_block<true>.  In a time-continuous stream a round fails for about one block-channel in 10^4.
 */
/* lanes (= blocks) per workgroup: 128 for batches and pushes — its two wavefronts have to find room beside the synthesis
 * kernel's (measured: 256 -> 1.03 ms per push beside the rest, 128 -> 0.57, 64 -> 1.08: more hand-offs) —, 256 for the
 * chain alone (gpsbb_chain_carrier: nothing else runs, fewer hand-offs) */
constexpr int FIXP_WG_BATCH = 128, FIXP_WG_ALONE = 256;

struct FixMap {
    int isconst; /* 1: o_b = v[0] whatever came before */
    double v[4]; /* 0: o_b = o_(b-1) + v[o_(b-1) mod 4] */
};
/* the map "first B, then A" */
__device__ __forceinline__ FixMap fix_compose(const FixMap &A, const FixMap &B)
{
    FixMap R;
    if (A.isconst)
        return A;
    if (B.isconst) {
        const double o = B.v[0];
        const int cls = (int)fmax(fmin(o, 0x1p+30), -0x1p+30) & 3;
        double add = A.v[0];
        add = cls == 1 ? A.v[1] : add;
        add = cls == 2 ? A.v[2] : add;
        add = cls == 3 ? A.v[3] : add;
        R.isconst = 1;
        R.v[0] = o + add;
        R.v[1] = R.v[2] = R.v[3] = 0.0;
        return R;
    }
    R.isconst = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int cls = (r + (int)fmax(fmin(B.v[r], 0x1p+30), -0x1p+30)) & 3;
        double add = A.v[0];
        add = cls == 1 ? A.v[1] : add;
        add = cls == 2 ? A.v[2] : add;
        add = cls == 3 ? A.v[3] : add;
        R.v[r] = B.v[r] + add;
    }
    return R;
}

/* LDS of k_chain_fix_par: a few hundred bytes (the kernel must fit on a CU beside a workgroup of the synthesis kernel,
 * which leaves 10 KB of LDS and 128 VGPRs per SIMD: anything bigger waits for a synthesis workgroup to leave) */
template <int FIXP_WG>
struct FixParLds {
    static constexpr int FIXP_WAVES = FIXP_WG / 64;
    int wc[FIXP_WAVES];        /* per wavefront: the composition of its lanes' maps */
    double wv[FIXP_WAVES][4];
    double wend[FIXP_WAVES];   /* per wavefront: its last lane's guessed end */
    double head_end;           /* the true end of the block before the head (or before the chunk) */
    int first_bad;
};

__device__ __forceinline__ FixMap fix_map_shfl_up(const FixMap &m, int delta)
{
    FixMap r;
    r.isconst = __shfl_up(m.isconst, delta);
#pragma unroll
    for (int j = 0; j < 4; j++)
        r.v[j] = __shfl_up(m.v[j], delta);
    return r;
}

template <int FIXP_WG>
__global__ __launch_bounds__(FIXP_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_chain_fix_par(BatchDev p)
{
    __shared__ FixParLds<FIXP_WG> L;
    /* a few wavefronts on whose latency the rest of the pre-pass (and every later push of a stream) waits, on SIMDs
     * they share with four older, VALU-bound wavefronts of the synthesis kernel: let them win the issue arbitration */
    __builtin_amdgcn_s_setprio(3);
    const int i = blockIdx.x;     /* channel */
    const int chunk = blockIdx.y; /* FIXP_WG consecutive "blocks" — virtual blocks (segments) from here on */
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c0 = chunk * FIXP_WG;
    if (i >= p.nch || c0 >= p.nvb)
        return;
    unsigned long long n_fallback = 0, n_hz = 0, n_rounds = 0;
    const int n = p.nvb - c0 < FIXP_WG ? p.nvb - c0 : FIXP_WG; /* blocks of this chunk */
    const int b = c0 + t;
    const bool mine = t < n;
    const size_t k = (size_t)(mine ? b : c0) * p.nch + i;
    const FixIn in = fix_load(p, mine ? b : c0, i);
    const bool on = mine && in.prn > 0;
    const bool cont = on && (b > 0 ? in.prn == in.prn_prev : (p.carry && ((p.cont0_mask >> i) & 1u)));
    const double endB_prev = (mine && b > 0) ? p.aux[k - p.nch].endB : 0.0;
    /* ---- this block as a map of the scan (nothing here depends on the chunks before this one) ---- */
    FixMap my;
    my.isconst = 0;
    my.v[0] = my.v[1] = my.v[2] = my.v[3] = 0.0; /* lanes past the chunk: the identity */
    bool have = false; /* x_last / r_last hold an evaluation of this lane's block */
    double x_last = 0.0, my_end = 0.0;
    FixOut r_last;
    r_last.end = 0.0;
    r_last.kind = FIX_OK;
    r_last.hz512 = 0;
    r_last.walked = false;
    if (mine && !on) {
        my.isconst = 1; /* an idle channel ends at 0.0 = endB */
    } else if (on) {
        /* a block whose start phase is known — one that starts a chain of its own, and the very first one (a stream: it
         * starts where the push before ended) — is a constant from the start; any other is tried from the four start
         * phases endB_(b-1) + j*u */
        const bool known = !cont || b == 0;
        const double xk = (cont && b == 0) ? p.carry->exact_end[i] : in.carr_phase; /* (b == 0 continues only in a stream) */
        bool regular = true;
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int j = 0; j < (known ? 1 : 4) && regular; j++) {
            const double xj = known ? xk : endB_prev + (double)j * 0x1p-53;
            const FixOut r = fix_block<false>(p, in, k, on, xj, false);
            const double o = (r.end - in.endB) * 0x1p+53;
            if (known) {
                have = true;
                x_last = xj;
                r_last = r;
                kk[0] = r.kind == FIX_OK ? o : 0.0; /* (a whole-block walk to come: a guess until then) */
            } else {
                regular = r.kind == FIX_OK && fabs(o) < 0x1p+30 && o == __builtin_rint(o);
                const double kj = o - (double)j;
                kk[0] = j == 0 ? kj : kk[0];
                kk[1] = j == 1 ? kj : kk[1];
                kk[2] = j == 2 ? kj : kk[2];
                kk[3] = j == 3 ? kj : kk[3];
            }
        }
        if (known || !regular) {
            my.isconst = 1; /* not regular: a guess, pass B's own end */
            my.v[0] = known ? kk[0] : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                my.v[j] = kk[j];
        }
    }
    /* ---- the maps of the chunk composed, lane t: blocks c0 .. c0 + t (still nothing that depends on other chunks) ---- */
    FixMap acc = my;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const FixMap prev = fix_map_shfl_up(acc, off);
        if (lane >= off)
            acc = fix_compose(acc, prev);
    }
    if (lane == 63) {
        L.wc[wave] = acc.isconst;
#pragma unroll
        for (int j = 0; j < 4; j++)
            L.wv[wave][j] = acc.v[j];
    }
    if (t == 0) {
        L.first_bad = FIXP_WG;
        L.head_end = endB_prev; /* pass B's end of the block before the chunk: what the incoming offset is counted from */
    }
    __syncthreads();
    for (int w = wave - 1; w >= 0 && !acc.isconst; w--) { /* (a constant absorbs everything before it) */
        FixMap pw;
        pw.isconst = L.wc[w];
#pragma unroll
        for (int j = 0; j < 4; j++)
            pw.v[j] = L.wv[w][j];
        acc = fix_compose(acc, pw);
    }
    const double endB_before = L.head_end;
    __syncthreads();
    /* ---- where the chunk before this one ended: the only thing chunks wait for each other for ---- */
    if (t == 0) {
        double cs = p.carry ? p.carry->exact_end[i] : 0.0; /* a stream: where the push before this one ended */
        if (chunk > 0) {
            const size_t at = (size_t)i * p.fix_chunks + (chunk - 1);
            __builtin_amdgcn_s_setprio(0); /* waiting must not cost the wavefronts it shares the SIMD with anything */
            /* Chunk c - 1 is a workgroup of the same launch with a lower blockIdx: the hardware dispatches those first, but HIP
             * does not promise it.  Should that order ever change (CU masks, another dispatcher) the wait is bounded — about a
             * second — and ends in the status word (GPSBB_E_INTERNAL at the next sync / pop) instead of a hung stream. */
            unsigned long long spins = 0;
            while (__hip_atomic_load(&p.fix_flag[at], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.fix_epoch) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > (1ull << 22)) {
                    atomicOr(p.status, ST_CHAIN_STALL);
                    break;
                }
            }
            __builtin_amdgcn_s_setprio(3);
            cs = bits_f64(__hip_atomic_load(&p.fix_end[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        L.head_end = cs;
    }
    __syncthreads();
    const double chunk_start = L.head_end;
    __syncthreads();
    if (chunk > 0) {
        /* everything before the chunk as one constant: the offset its first block starts with */
        FixMap inc;
        inc.isconst = 1;
        inc.v[0] = (chunk_start - endB_before) * 0x1p+53;
        inc.v[1] = inc.v[2] = inc.v[3] = 0.0;
        acc = fix_compose(acc, inc);
    }
    /* ---- guess, check, move the head: until every block's start phase is the true one, then commit ---- */
    int lo = -1; /* the head: the last block whose end is known to be true (-1: the block before the chunk) */
    double lo_start = chunk_start; /* the true end of block lo - 1 (wave-uniform) */
    bool committed = false, final = false, pre = true; /* pre: first round, the maps are composed already */
    double prev_end = 0.0;
    for (;;) {
        const bool head = t == lo && mine && !final;
        if (head) { /* from its true start phase (a block that became the head in the round before has just been there) */
            const double x = cont ? lo_start : in.carr_phase;
            if (!have || f64_bits(x) != f64_bits(x_last)) {
                r_last = fix_block<false>(p, in, k, on, x, false);
                x_last = x;
                have = true;
            }
        }
        /* Whoever has to leave its results now: the head if only a walk of the whole block tells where it ends, and —
         * last round — every block. */
        const bool commit_now = mine && !committed && (final || (head && r_last.kind == FIX_SLOW));
        if (commit_now) {
            const double x = cont ? (final ? prev_end : lo_start) : in.carr_phase;
            const FixOut r = fix_block<true>(p, in, k, on, x, true);
            committed = true;
            n_fallback += r.walked ? 1u : 0u;
            fix_publish(p, k, on, cont, x, r, n_hz);
            x_last = x;
            r_last = r;
            have = true;
        }
        if (final)
            break;
        if (!pre) {
            if (head) {
                my.isconst = 1;
                my.v[0] = (r_last.end - in.endB) * 0x1p+53;
            }
            /* inclusive scan of the maps over [lo, n): inside the wavefront by shuffles ... */
            acc = my;
            if (t < lo) {
                acc.isconst = 0;
                acc.v[0] = acc.v[1] = acc.v[2] = acc.v[3] = 0.0;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const FixMap prev = fix_map_shfl_up(acc, off);
                if (lane >= off)
                    acc = fix_compose(acc, prev);
            }
            /* ... across the wavefronts through LDS */
            if (lane == 63) {
                L.wc[wave] = acc.isconst;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    L.wv[wave][j] = acc.v[j];
            }
            if (t == 0)
                L.first_bad = FIXP_WG;
            __syncthreads();
            for (int w = wave - 1; w >= 0 && !acc.isconst; w--) { /* (a constant absorbs everything before it) */
                FixMap pw;
                pw.isconst = L.wc[w];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    pw.v[j] = L.wv[w][j];
                acc = fix_compose(acc, pw);
            }
        }
        pre = false;
        /* every map from the head on is a constant now: the guessed ends */
        if (mine && t >= lo)
            my_end = t == lo ? r_last.end : in.endB + acc.v[0] * 0x1p-53;
        if (lane == 63)
            L.wend[wave] = my_end;
        __syncthreads();
        prev_end = __shfl_up(my_end, 1);
        if (lane == 0)
            prev_end = wave > 0 ? L.wend[wave - 1] : chunk_start;
        /* check: from the guessed start, does this block end where the next one was told it would? */
        if (mine && t > lo) {
            const double x = cont ? prev_end : in.carr_phase;
            if (!have || f64_bits(x) != f64_bits(x_last)) {
                r_last = fix_block<false>(p, in, k, on, x, false);
                x_last = x;
                have = true;
            }
            if (r_last.kind != FIX_OK || f64_bits(r_last.end) != f64_bits(my_end))
                atomicMin(&L.first_bad, t);
        }
        __syncthreads();
        const int bad = L.first_bad;
        if (bad >= FIXP_WG) {
            /* every x is the true one.  The next chunk can go on (it waits for nothing else); then everybody commits. */
            if (t == n - 1) {
                const size_t at = (size_t)i * p.fix_chunks + chunk;
                __hip_atomic_store(&p.fix_end[at], f64_bits(my_end), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&p.fix_flag[at], p.fix_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (p.carry && c0 + n >= p.nvb)
                    p.carry->exact_end[i] = my_end;
            }
            final = true;
            continue;
        }
        /* every block before `bad` is settled, so the guess handed to it is its true start phase (and the evaluation
         * it has just made from it stands) */
        if (t == bad)
            L.head_end = prev_end;
        __syncthreads();
        lo_start = L.head_end;
        lo = bad;
        n_rounds++;
    }
    if (t == 0 && n_rounds)
        atomicAdd(p.hazards + 6, n_rounds); /* rounds of guess-and-check beyond the first (GPSBB_INFO_CHAIN_REPAIRS) */
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
/* One wavefront per chain, two rows per lane and turn.  A chain keeps a few hundred rows (those that hold a tile's first
 * sample): with 256 lanes x 4 rows a turn's 1 024 slots were 60 % full and the kernel issued 2.81e7 vector instructions per
 * 400-block push; 64 x 2: 1.92e7, in less time (tools/tiles_geom.sh: exact counters, 64 / 128 / 192 / 256 lanes x 1..6 rows).
 * The pre-pass's instructions are what the synthesis kernel beside it pays for. */
#ifndef GPSBB_TILES_WG
#define GPSBB_TILES_WG 64
#endif
#ifndef GPSBB_TILES_U
#define GPSBB_TILES_U 2 /* rows per lane and turn */
#endif
__global__ __launch_bounds__(GPSBB_TILES_WG) void k_tiles(BatchDev p)
{
    __builtin_amdgcn_s_setprio(GPSBB_SEED_PRIO); /* latency-bound and short, like the walks it follows */
    const int chain = blockIdx.x;
    const int nbc = p.nblocks * p.nch;
    /* code chains: one per (block, channel); carrier chains: one per (segment, channel) where the carrier is chained on
     * the device (BatchDev::nseg), else per (block, channel) */
    const int kind = chain >= nbc ? 1 : 0, kv = chain - kind * nbc;
    const int nseg = (kind && p.chain_dev) ? p.nseg : 1;
    const int vb = kv / p.nch, sgi = vb % nseg;
    const int bi = (vb / nseg) * p.nch + kv % p.nch;                   /* block * nch + channel */
    const int t0 = sgi * p.seg_tiles;                                   /* the segment's first tile ... */
    const int ntl = nseg > 1 ? ((p.ntiles - t0 < p.seg_tiles) ? p.ntiles - t0 : p.seg_tiles) : p.ntiles; /* ... and how many it has */
    /* the tile counters of this table set, for the synthesis kernel that follows (the one that last used them has
     * finished: the pre-pass waited for it): saves a memset and its launch gap on the synthesis stream */
    if (chain <= p.nblocks && threadIdx.x == 0) /* ([nblocks]: the helpers' ticket counter, ev_pick_block; the grid has at least 2 * nblocks * nch workgroups) */
        p.tile_ctr[chain] = 0;
    const int b = bi / p.nch, i = bi % p.nch;
    if (kind && p.kph0) {
        /* fixed-point carrier: the table index at every tile start in closed form (no rows) */
        if (p.ch[bi].prn > 0) {
            double *__restrict__ txf = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + 1) * (size_t)p.ntiles;
            for (int t = threadIdx.x; t < p.ntiles; t += blockDim.x)
                txf[t] = fixed_tile_index(p.kph0[bi], p.kstep[bi], t);
        }
        return;
    }
    const int cnt = p.row_cnt[chain];
    if (cnt <= 0)
        return;
    const WalkRow *__restrict__ rows = reinterpret_cast<const WalkRow *>(p.rows) + p.row_off[chain];
    /* the chain's step (c:2709 / c:2741), from which every row's increment follows (walk_row_step) */
    const double s = kind ? mul_rn(p.ch[bi].f_carr, p.delt) : mul_rn(p.ch[bi].f_code, p.delt);
    double *__restrict__ tx = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + kind) * (size_t)p.ntiles + t0;
    uint32_t *__restrict__ tn = p.tile_nav + ((size_t)b * (size_t)p.nch + i) * (size_t)p.ntiles;
    /* carrier chained on the device: the true states are pass B's plus an offset per stretch of rows (k_chain_fix) */
    const bool shifted = kind && p.aux && p.chain_dev;
    const ChainAux *aux = shifted ? &p.aux[kv] : nullptr;
    const int ncross = shifted ? (aux->ncross > 0 ? aux->ncross : 0) : 0;
    /* k_chain_fix walked the first lap on its own: those rows (in the chain's prefix region) hold the samples
     * before prefix_end, pass B's rows before wrap_row are void */
    const int prefix_cnt = shifted ? aux->prefix_cnt : 0;
    const int void_before = prefix_cnt > 0 ? aux->prefix_end : 0; /* pass B's rows before this sample are void */
    if (prefix_cnt > 0) {
        const WalkRow *pr = reinterpret_cast<const WalkRow *>(p.prefix_rows) + (size_t)kv * CHAIN_PREFIX_CAP;
        const int pend = aux->prefix_end;
        for (int r = threadIdx.x; r < prefix_cnt; r += blockDim.x) {
            const WalkRow row = pr[r];
            const int n_next = r + 1 < prefix_cnt ? pr[r + 1].n0 : pend;
            int t = (int)(((uint32_t)row.n0 + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            int t_end = (int)(((uint32_t)n_next + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            t_end = t_end < ntl ? t_end : ntl;
            for (; t < t_end; t++)
                tx[t] = mul_rn(__fma_rn((double)(t * TILE - row.n0), walk_row_step(row.x, s), row.x), 512.0);
        }
    }
    /* a few rows per lane and turn, their loads issued together */
    constexpr int U = GPSBB_TILES_U;
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
            t_end = t_end < ntl ? t_end : ntl;
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
