/*
 * gpsbb_laps.hip.h — the exact NCO pre-pass of the model kernels, LAP-PARALLEL (gfx950; round 5).
 *
 * What it replaces: k_walk (one lane per chain or segment: ~1 750 dependent rows in a row), the chain kernels that
 * stitch segments together (pass A, k_chain_prefix, k_chain_fix*) and k_tiles (rows -> tile states).  What it
 * computes is the same: the reference's NCO states (plutogpssim.c:2709-2712 code, 2741-2746 carrier) at the first
 * sample of every 1024-sample tile, the end-of-block states, the hazard counts — bit for bit.
 *
 * The idea.  A LAP is the stretch of a chain from one wrap to the next (carrier: 1 / |f_carr * delt| samples;
 * code: 1023 / (f_code * delt)).  The state right after a wrap lies on the COARSEST grid the phase ever meets
 * (rising carrier: x2 - 1.0 with x2 in [1, 2), a multiple of 2^-52; falling: x2 + 1.0 rounded into [0.5, 1), a
 * multiple of 2^-53; code: x2 - 1023.0 with x2 in [1023, 1024), a multiple of 2^-43).  Every rounding of the
 * recurrence commutes with a shift of the state by a multiple of the grid it rounds on, so two trajectories whose
 * post-wrap states differ by d (a multiple of that coarsest grid u) stay exactly d apart for as long as they spend
 * the same steps in the same binades — through the whole lap, through the next wrap, for ever.  So:
 *
 *   k_lap_plan   a model of the recurrence (step + rounding drift: per binade and unit of phase, integrated over the part of a
 *                lap a block covers: lap_R) says where every lap of every chain starts — sample number and a reference
 *                state A on the grid u — without walking anything.  One lane per block and channel, two scans over the blocks.
 *   k_lap_pass1  ONE LANE PER LAP (per LapDev::unit consecutive laps: 4 carrier, 2 code; 1 in small batches) walks its lap exactly from A (the same turn as k_walk: a regular run of the
 *                exact jump-ahead, gpsbb_nco.h, then one genuine IEEE step; ~15 turns per lap) and finds
 *                where the reference trajectory ends: y'.  The next lap's reference start is A_next, so the offset of
 *                the true trajectory to the reference one changes by (y' - A_next) / u from lap to lap: an exact
 *                integer.  (Where a sum on the coarsest grid is an exact tie the change depends on the offset's residue
 *                modulo 4: the link is a map m -> m + g + o[m mod 4], see LapMap.)  The links are composed by a scan
 *                over each workgroup's 256 laps,
 *   k_lap_scan   ... and over the workgroups of a chain: the offset m of every lap.  A chain's first lap (a block that
 *                starts a chain, or the first block of a stream's push: the exact phase the push before left in
 *                device memory) starts from its exact state: m = 0.
 *   k_lap_pass2  one lane per lap again, now from the TRUE start A + m*u: the same walk, this time leaving the tile
 *                states, the end-of-block states and the hazard counts.  NOTHING RESTS ON THE ARGUMENT ABOVE: every lane
 *                compares where its walk ended (position and state, bit for bit) with where the next lap was told
 *                to start.  A chain's first lap starts from the truth; if every link holds, every lap did.  A lane
 *                never walks (or writes) beyond its own territory — up to the next lap's planned start —, so a wrong
 *                guess spoils nothing but its own territory.
 *   k_lap_repair one wavefront per chain kind and channel; does nothing unless a link failed (a reference lap
 *                that spends a step more or less in some binade than the true one: about one lap in 10^8; a wrap
 *                within the model's error of a lap's planned first sample).  Then: from the failed lap's true end it
 *                walks on sequentially until a wrap falls on a planned lap start, and from there — unless the state there is
 *                what pass 2 started that lap from, and nothing has been written over its output since — re-does pass 1,
 *                the scan and pass 2 for the rest of the chain, 64 laps at a time, checking as it goes.
 *
 * Where runs are short (the first steps of a rising lap, the last of a falling one) a lane takes LAP_BURST plain steps of the
 * recurrence instead of turns (lap_run).
 *
 * Cost: a lap is walked twice (~13 turns of ~60 vector instructions each) whatever the length of the chain, and all
 * laps are independent: the pre-pass of a 400-block push is a few hundred thousand wavefront-turns spread over the
 * machine instead of 1 750 turns in a row on 500 wavefronts.  The walks are exact (genuine IEEE adds, the jump-ahead
 * of gpsbb_nco.h); the model decides only how often the repair kernel has work.
 *
 * Eligibility (lap_eligible, host): carrier steps below 2^-50 and zero steps (the turn of the walk covers states up to
 * 50 binades above the step) stay with the row walks.  Steps whose sums tie on the coarsest grid — a property of the
 * step's low bits, one block-channel in 2^12 — are walked like any other; what they do to the offset is part of the link
 * (LapMap: the change depends on the offset modulo 4).
 */
#ifndef GPSBB_LAPS_HIP_H
#define GPSBB_LAPS_HIP_H

#include "gpsbb_walk.hip.h"

namespace gpsbb_impl {

/* (measurement: GPSBB_LAP_WAVES = n caps the two passes' registers at what n wavefronts per SIMD leave each) */
#ifdef GPSBB_LAP_WAVES
#define GPSBB_LAP_OCC __attribute__((amdgpu_waves_per_eu(GPSBB_LAP_WAVES, GPSBB_LAP_WAVES)))
#else
#define GPSBB_LAP_OCC
#endif
constexpr int LAP_WG = 256; /* lanes (= laps) per workgroup of the two passes: one chunk of the scan */
constexpr uint32_t LAPF_ACTIVE = 1u; /* the channel is on in this block */
constexpr uint32_t LAPF_HEAD = 2u;   /* the block starts a chain: its first lap starts at sample 0 from an exactly known state */
constexpr uint32_t LAPF_CONT = 4u;   /* the chain goes on into the next block */
constexpr uint32_t LAPF_CONST = 8u;  /* LapRec: the prefix map is a constant */
constexpr uint32_t LAPF_BAD = 16u;   /* LapRec: pass 2 did not end where the next lap starts */
constexpr uint32_t ST_LAP_PLAN = 8u; /* status word: more laps than the host planned room for (lap_bound) */

/* planner output per kind and (channel, block), channel-major */
struct LapBC {
    double phi;     /* model state at the block's first sample (a head: the exact start state) */
    double s;       /* the step, fl(f * delt) (c:2709 / c:2741) */
    double R1;      /* the model's rounding drift over one whole lap with this step (lap_R at the top of the range) */
    double Rphi;    /* ... and from 0 up to phi */
    uint32_t flags; /* LAPF_* */
    uint32_t c0;    /* code: code periods at the block's first sample, icode + 20*ibit + 600*iword */
};
static_assert(sizeof(LapBC) == 40, "LapBC layout");

/* one lap, written by pass 1, read by pass 2 and the repair */
struct LapRec {
    double A;       /* reference start state (a head: the exact start) */
    double g;       /* pass 1: the composition of the links from the chunk's first lap up to and including the link that leaves
                       this lap, as a LapMap {g, ov, CONST}: the offset of the lap after this one from the chunk's first.
                       pass 2: the offset m this lap was walked from */
    int32_t b, n0;  /* where it starts: block, sample in the block (0 <= n0 < nsamp) */
    uint32_t flags;
    uint32_t hz;    /* hazards pass 2 counted in this lap */
    uint32_t ov;    /* pass 1: LapMap::o1..o3 as three signed bytes */
    uint32_t _pad;
};
static_assert(sizeof(LapRec) == 40, "LapRec layout");

struct LapAgg {
    double g;
    uint32_t ov, isconst;
};

struct LapDev {
    LapBC *bc;           /* [2][nch * nblocks] (kind, then channel-major) */
    uint32_t *lane0;     /* [2][nch * (nblocks + 1)]: first lap of every block within its channel's range */
    uint32_t *nlaps;     /* [2][GPSBB_MAX_CHAN] laps of each channel (planner) */
    uint32_t *nbad;      /* [2][GPSBB_MAX_CHAN] links pass 2 found broken */
    LapRec *rec;         /* [chunks * LAP_WG] */
    LapAgg *agg;         /* [chunks] */
    double *chunk_m;     /* [chunks] the offset of each chunk's first lap (k_lap_scan) */
    uint32_t *chunk_bad; /* [chunks] */
    uint32_t chunk0[2][GPSBB_MAX_CHAN + 1]; /* first chunk of each channel's range (host plan: lap_bound) */
    int chained;         /* GPSBB_CHAIN_CARRIER is in force (blocks continue each other) */
    uint32_t jitter;     /* experiments: reference states are pushed off by up to this many grid steps (exercises the repair) */
    int burst;           /* plain steps where runs are short (lap_run): for at least 1 / burst of the lanes still walking (0: never) */
    int unit[2];         /* laps per lane, per kind (0: a block's whole chain where its first state is known — code chains, which are a
                            block long: no reference walk at all, 100 laps in a row per lane): a lane walks `unit` consecutive laps of its chain (up to the next lane's first
                            sample): what a lane costs besides its walk — finding its lap, the model, the scan, its record — is as
                            much as one lap's walk, so several laps share it (1: a lane per lap) */
};

template <int KIND>
struct LapK {
    static constexpr int TOPEX = KIND == NCO_CARR ? 1023 : 1023 + 10;
};

/* the coarsest grid of a chain's states: what offsets are counted in */
template <int KIND>
__device__ __forceinline__ double lap_unit() { return KIND == NCO_CARR ? 0x1p-53 : 0x1p-43; }
template <int KIND>
__device__ __forceinline__ double lap_runit() { return KIND == NCO_CARR ? 0x1p+53 : 0x1p+43; }

/* ---- link maps ------------------------------------------------------------------------------------------------
 * How the offset of the next lap follows from this lap's: m -> CONST ? g : m + g + o[m mod 4], o[0] = 0.  A plain
 * translation is {g, 0, 0, 0}.  The residue matters where a sum is rounded on the coarsest grid exactly half-way (or, for an
 * offset that is not a whole number of steps of that grid, anywhere near it): the "+ 1.0" of a falling carrier's wrap (odd
 * offsets), the sum that passes 1.0 on the two-unit grid of [1, 2) (offsets that are 2 mod 4 where it ties — steps whose low
 * bits make every other wrap tie exist: one block-channel in 2^12 —, odd ones always), the first step of a lap whose step
 * ties on the grid of [0.5, 1).  Maps of this form compose to maps of this form; the offsets stay small integers. */
struct LapMap {
    double g;
    int o1, o2, o3;
    int isconst;
};
__device__ __forceinline__ int lap_res(double m) { return (int)((long long)m & 3ll); }
__device__ __forceinline__ int lap_o(const LapMap &A, int r) { return r == 1 ? A.o1 : (r == 2 ? A.o2 : (r == 3 ? A.o3 : 0)); }
__device__ __forceinline__ double lap_apply(const LapMap &A, double m)
{
    return A.isconst ? A.g : m + A.g + (double)lap_o(A, lap_res(m));
}
/* "first B, then A" */
__device__ __forceinline__ LapMap lap_compose(const LapMap &A, const LapMap &B)
{
    LapMap R;
    if (A.isconst)
        return A;
    if (B.isconst) {
        R.isconst = 1;
        R.g = lap_apply(A, B.g);
        R.o1 = R.o2 = R.o3 = 0;
        return R;
    }
    R.isconst = 0;
    const int gb = lap_res(B.g);
    const int a0 = lap_o(A, gb & 3);
    R.g = B.g + A.g + (double)a0;
    R.o1 = B.o1 + lap_o(A, (1 + gb + B.o1) & 3) - a0;
    R.o2 = B.o2 + lap_o(A, (2 + gb + B.o2) & 3) - a0;
    R.o3 = B.o3 + lap_o(A, (3 + gb + B.o3) & 3) - a0;
    return R;
}
__device__ __forceinline__ uint32_t lap_pack_o(const LapMap &m)
{
    return ((uint32_t)m.o1 & 0xffu) | (((uint32_t)m.o2 & 0xffu) << 8) | (((uint32_t)m.o3 & 0xffu) << 16);
}
__device__ __forceinline__ LapMap lap_unpack(double g, uint32_t ov, int isconst)
{
    LapMap m;
    m.g = g;
    m.o1 = (int)(int8_t)(ov & 0xffu);
    m.o2 = (int)(int8_t)((ov >> 8) & 0xffu);
    m.o3 = (int)(int8_t)((ov >> 16) & 0xffu);
    m.isconst = isconst;
    return m;
}
__device__ __forceinline__ LapMap lap_map_shfl_up(const LapMap &m, int delta)
{
    LapMap r;
    r.isconst = __shfl_up(m.isconst, delta);
    r.g = __shfl_up(m.g, delta);
    r.o1 = __shfl_up(m.o1, delta);
    r.o2 = __shfl_up(m.o2, delta);
    r.o3 = __shfl_up(m.o3, delta);
    return r;
}
__device__ __forceinline__ LapMap lap_map_shfl(const LapMap &m, int src)
{
    LapMap r;
    r.isconst = __shfl(m.isconst, src);
    r.g = __shfl(m.g, src);
    r.o1 = __shfl(m.o1, src);
    r.o2 = __shfl(m.o2, src);
    r.o3 = __shfl(m.o3, src);
    return r;
}
__device__ __forceinline__ LapMap lap_identity()
{
    LapMap r;
    r.isconst = 0;
    r.g = 0.0;
    r.o1 = r.o2 = r.o3 = 0;
    return r;
}
__device__ __forceinline__ LapMap lap_const(double g)
{
    LapMap r = lap_identity();
    r.isconst = 1;
    r.g = g;
    return r;
}
/* inclusive scan over the lanes of a wavefront */
__device__ __forceinline__ LapMap lap_wave_scan(LapMap acc, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const LapMap prev = lap_map_shfl_up(acc, off);
        if (lane >= off)
            acc = lap_compose(acc, prev);
    }
    return acc;
}

/* ---- the model ---------------------------------------------------------------------------------------------- */

/* The model: rounding drift of the recurrence x = fl(x + s).  While x is in binade e a step adds not s but s rounded to a
 * multiple of that binade's last place, S_e (gpsbb_nco.h): a drift of S_e - s per step, i.e. of (S_e - s) / |s| per unit of phase
 * travelled there.  R(x) is that density integrated from 0 to x (piecewise linear, one piece per binade from the step's own up
 * to the top one; nothing below: those steps are exact), R1 = R(range) the drift of a whole lap.  n steps from x0 then end at
 *     x0 + n*s + D,   D = wraps * R1 + R(end) - R(x0)  (rising)  /  wraps * R1 + R(x0) - R(end)  (falling)
 * — the host's CarrDrift (gpsbb.hip) is the same model.  What it leaves out (the roundings of the steps that cross a binade
 * edge, that a binade holds a whole number of steps) averages out: a few 1e-15 per block.  The walks are exact whatever the
 * model says; a poor model only makes reference laps start further from the truth, and links break more often. */
template <int KIND>
__device__ __forceinline__ double lap_R(double s, double x, double &R1)
{
    const int es = (int)((f64_bits(s) >> 52) & 0x7ff);
    const int etop = KIND == NCO_CARR ? 1022 : 1023 + 9; /* the top binade: [0.5, 1) / [512, 1024) */
    double R = 0.0;
    R1 = 0.0;
    if (es < 1023 - 60 || es > etop)
        return 0.0;
    const double rs = 1.0 / fabs(s);
    for (int e = es; e <= etop; e++) {
        const double C = bits_f64(((uint64_t)e << 52) | (1ull << 51)); /* 1.5 * 2^e */
        const double S = add_rn(add_rn(s, C), -C);
        const double dens = add_rn(S, -s) * rs;
        const double lo = bits_f64((uint64_t)e << 52); /* the binade: [2^e, 2^(e+1)), the code's top one: [512, 1023) */
        const double w = KIND == NCO_CODE && e == etop ? 511.0 : lo;
        R1 += w * dens;
        const double in = x - lo;
        R += (in <= 0.0 ? 0.0 : (in < w ? in : w)) * dens;
    }
    return R;
}

/* where a lap starts, by the model */
struct LapStart {
    double A;
    int32_t b, n0;
    uint32_t head;  /* the lap is its chain's first: A is exact, the offset 0 */
    uint32_t jc;    /* code: code periods completed since the block's first sample when the lap starts */
};

/* block of lane r of channel i's range: the largest b with lane0[b] <= r */
__device__ __forceinline__ int lap_block_of(const uint32_t *lane0, int nblocks, uint32_t r)
{
    int lo = 0, hi = nblocks; /* lane0[nblocks] = the channel's lap count > r */
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (lane0[mid] <= r)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

template <int KIND>
__device__ __forceinline__ LapStart lap_start(const BatchDev &p, const LapDev &L, int i, uint32_t r)
{
    const uint32_t *lane0 = L.lane0 + ((size_t)KIND * p.nch + i) * ((size_t)p.nblocks + 1);
    const int b = lap_block_of(lane0, p.nblocks, r);
    const LapBC bc = L.bc[((size_t)KIND * p.nch + i) * (size_t)p.nblocks + b];
    const uint32_t j = r - lane0[b];
    LapStart st;
    st.b = b;
    st.head = 0;
    st.jc = 0;
    if ((bc.flags & LAPF_HEAD) && j == 0) {
        st.head = 1;
        st.n0 = 0;
        st.A = bc.phi;
        return st;
    }
    /* lane j of the block (after its head) starts at the block's wrap number jw = unit * (j - head), 0-based */
    const uint32_t jw = (j - ((bc.flags & LAPF_HEAD) ? 1u : 0u)) * (uint32_t)(L.unit[KIND] > 0 ? L.unit[KIND] : 1);
    const double s = bc.s;
    const double range_ = KIND == NCO_CARR ? 1.0 : 1023.0;
    const double sbar = s + fabs(s) * bc.R1 * (1.0 / range_); /* the mean step, drift included: for the estimate of n0 only */
    const bool neg = s < 0.0;
    const double range = KIND == NCO_CARR ? 1.0 : 1023.0;
    const double level = (double)(jw + 1) * range; /* exact */
    /* the first sample n with phi + n*sbar >= level (rising) / < -(level - range) (falling: the post-wrap state is that + 1) */
    const double target = neg ? bc.phi + (level - range) : level - bc.phi;
    const double sa = fabs(sbar);
    int n0 = neg ? (int)floor(target / sa) + 1 : (int)ceil(target / sa);
    double A = 0.0;
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
        /* A = phi + n0*sbar -/+ level, with the product in two pieces (the model has to be good to ~1e-15) */
        const double nd = (double)n0;
        const double hi = nd * s;
        /* ... and the drift up to there: a post-wrap state is the end of jw + 1 whole laps but for the stretch from 0 up to
         * phi (rising: not travelled) / from phi down to 0 (falling: travelled on top of jw whole laps) */
        const double D = neg ? (double)jw * bc.R1 + bc.Rphi : (double)(jw + 1u) * bc.R1 - bc.Rphi;
        const double lo = __fma_rn(nd, s, -hi) + D;
        A = ((neg ? hi + level : hi - level) + bc.phi) + lo;
        if (it)
            break;
        /* a post-wrap state lies in [0, |s|) (rising) / [range - |s|, range) (falling) */
        if (!neg) {
            if (A < 0.0)
                n0++;
            else if (A >= sa)
                n0--;
            else
                break;
        } else {
            if (A >= 1.0)
                n0++;
            else if (A < 1.0 - sa)
                n0--;
            else
                break;
        }
    }
    {
        /* a wrap on the block's last step starts a lap only if the chain goes on into the next block */
        const int n0max = (KIND == NCO_CARR && (bc.flags & LAPF_CONT)) ? p.nsamp : p.nsamp - 1;
        n0 = n0 > n0max ? n0max : n0;
        n0 = n0 < 1 ? 1 : n0;
    }
    /* onto the grid of post-wrap states (the true ones are on it: the offset is then a whole number of steps) */
    if (KIND == NCO_CARR) {
        if (!neg) {
            A = A < 0.0 ? 0.0 : A;
            A = add_rn(add_rn(A, 1.0), -1.0); /* a multiple of 2^-52 */
        } else {
            A = A < 0.5 ? 0.5 : (A >= 1.0 ? 0x1.fffffffffffffp-1 : A); /* any double in [0.5, 1) is a multiple of 2^-53 */
        }
    } else {
        A = A < 0.0 ? 0.0 : A;
        A = add_rn(add_rn(A, 512.0), -512.0); /* a multiple of 2^-43 */
    }
    if (L.jitter) {
        /* experiments: push the reference off the model by a pseudo-random whole number of grid steps (always even for a rising
         * carrier, whose states are multiples of 2^-52) — the walks do not care where they start, the links carry the difference,
         * and large offsets make laps cross binade edges at other samples than the truth: work for the repair */
        uint32_t hsh = (r + 0x9e3779b9u * (uint32_t)(i + 1)) * 2654435761u;
        hsh ^= hsh >> 15;
        const double k = (double)(hsh % L.jitter) * (KIND == NCO_CARR && !neg ? 2.0 : 1.0);
        const double Aj = KIND == NCO_CARR && neg ? A - k * lap_unit<KIND>() : A + k * lap_unit<KIND>();
        if (Aj >= 0.0 && Aj < range)
            A = Aj;
    }
    st.n0 = n0;
    st.A = A;
    st.jc = jw + 1;
    if (n0 >= p.nsamp) { /* the wrap of the block's last step: the lap starts with the next block */
        st.b = b + 1;
        st.n0 = 0;
        st.jc = 0; /* (carrier only: a code chain ends with its block, the planner plans no lap there) */
    }
    return st;
}

/* ---- the walk ------------------------------------------------------------------------------------------------ */

constexpr int LAP_OUT_WRAP = 1;  /* the walk ended with a wrap: (b, n) is the first sample of the next lap, x its state */
constexpr int LAP_OUT_LATE = 2;  /* ... at the end of its territory without a wrap */
constexpr int LAP_OUT_CHAIN = 3; /* ... with its chain (the block's last sample; the chain does not go on) */

template <int KIND>
struct LapLane {
    double x, s;
    int32_t b, n;
    int32_t bt, nt;    /* the territory's end: the next lap's planned first sample (bt = INT32_MAX: the chain's end) */
    int32_t nmax;      /* in this block: samples up to here */
    int32_t es;
    uint64_t tiemask;
    uint32_t jc;       /* code: code periods completed since the block's first sample */
    uint32_t c0;       /* code: ... and before it */
    uint32_t bits;     /* code: data bits in force (walk_dbits) */
    uint32_t hz;       /* hazards met: carrier, samples whose phase is exactly 1.0; code, data-bit fetches past dwrd[59] */
    uint32_t bcflags;
    int outcome;
    int so;            /* pass 1: what the first step in the top binade adds to an odd offset, where the step ties on that binade's grid */
    bool tt;           /* the block's step ties on the grid of the top binade ([0.5, 1) / [512, 1024)): every sum there is half-way */
    bool active, neg;
    bool fresh;        /* code: the walk starts on the first sample after a roll-over (whose data-bit fetch is this lap's to count) */
    bool one_lap;      /* stop at the first wrap wherever it falls (the repair's walks; a lane otherwise goes on through the wraps
                          inside its territory: it holds several laps, LapDev::unit) */
    LapMap acc;        /* pass 1: what the wraps and top-binade entries so far do to an offset, besides carrying it along */
    int32_t navt;      /* code, pass 2: the first tile of the block whose data bits this lane has not written yet (lap_nav_out) */
};

/* the data bits of code period c of a channel (c:2717-2733): bit 0: the bit in force is -1; bit 1: the one in force after the
 * next roll-over is */
__device__ __forceinline__ uint32_t lap_nav_of(uint32_t c)
{
    const uint32_t iword = c / 600u, rem = c - iword * 600u, ibit = rem / 20u, icode = rem - ibit * 20u;
    return nav_pack((int)icode, (int)ibit, (int)iword);
}
__device__ __forceinline__ uint32_t lap_code_bits(const uint32_t *dwrd, uint32_t c)
{
    const uint32_t cur = nav_bit(dwrd, lap_nav_of(c)) < 0 ? 1u : 0u;
    const uint32_t nxt = nav_bit(dwrd, lap_nav_of(c + 1u)) < 0 ? 2u : 0u;
    return cur | nxt;
}

/* a lane enters block w.b at sample w.n (its own first block, or the next one of its chain): the block's step and limits */
template <int KIND>
__device__ __forceinline__ void lap_enter_block(const BatchDev &p, const LapDev &L, int i, LapLane<KIND> &w)
{
    const LapBC bc = L.bc[((size_t)KIND * p.nch + i) * (size_t)p.nblocks + w.b];
    w.s = bc.s;
    w.bcflags = bc.flags;
    w.c0 = bc.c0;
    const uint64_t sb = f64_bits(bc.s);
    w.es = (int)((sb >> 52) & 0x7ff);
    /* (a step of exactly zero — a carrier without Doppler: a bench-top scenario — ties nowhere: the state stands still, lap_run) */
    w.tiemask = (sb << 1) == 0ull ? 0ull : walk_tiemask(sb);
    w.neg = bc.s < 0.0;
    w.tt = ((w.tiemask >> (((KIND == NCO_CARR ? 1022 : 1023 + 9) - w.es) & 63)) & 1ull) != 0ull;
    w.nmax = (w.b == w.bt) ? w.nt : p.nsamp;
}

/* tile states of a row: samples n .. n + k, state x at n, increment S */
/* The data bits of the tiles of a code period (bit 0: the bit in force is -1, bit 1: the one after the next roll-over is): the same
 * for every tile whose first sample lies in the period, so they go out when the period is over (or the lane's walk is) — two dozen
 * consecutive words, in 32- and 16-byte pieces — instead of one word with every tile state: stored one by one, four bytes at a
 * time, they were more than half of what the pre-pass writes to HBM (a sector of 32 bytes per word: WRITE_SIZE).  Tiles
 * [w.navt, the first tile whose first sample is n_excl or later). */
template <int KIND>
__device__ __forceinline__ void lap_nav_out(const BatchDev &p, int i, LapLane<KIND> &w, bool on, int n_excl, uint32_t bits)
{
    typedef uint32_t lap_u4 __attribute__((ext_vector_type(4)));
    const int t = w.navt;
    int t_end = (int)(((uint32_t)n_excl + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
    t_end = t_end < p.ntiles ? t_end : p.ntiles;
    int cnt = on && t_end > t ? t_end - t : 0;
    if (cnt > 0)
        w.navt = t_end;
    if (!__ballot(cnt > 0))
        return;
    uint32_t *pn = p.tile_nav + ((size_t)w.b * (size_t)p.nch + i) * (size_t)p.ntiles + t;
    const lap_u4 b4 = lap_u4{bits, bits, bits, bits};
    while (__ballot(cnt > 0)) {
        const bool oct = cnt >= 8 && (reinterpret_cast<uintptr_t>(pn) & 31u) == 0;
        const bool quad = !oct && cnt >= 4 && (reinterpret_cast<uintptr_t>(pn) & 15u) == 0;
        if (__ballot(oct)) {
            if (oct) {
                reinterpret_cast<lap_u4 *>(pn)[0] = b4;
                reinterpret_cast<lap_u4 *>(pn)[1] = b4;
            }
        }
        if (quad)
            *reinterpret_cast<lap_u4 *>(pn) = b4;
        if (cnt > 0 && !oct && !quad)
            *pn = bits;
        const int adv = oct ? 8 : (quad ? 4 : (cnt > 0 ? 1 : 0));
        pn += adv;
        cnt -= adv;
    }
}

template <int KIND, bool WIDE>
__device__ __forceinline__ void lap_emit_row(const BatchDev &p, int i, bool on, int b, int n, int k, double x, double S, uint32_t bits)
{
    int t = (int)(((uint32_t)n + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
    int t_end = (int)((uint32_t)(n + k) / (uint32_t)TILE) + 1; /* one past the last tile whose first sample is in the row */
    t_end = t_end < p.ntiles ? t_end : p.ntiles;
    int cnt = on ? t_end - t : 0;
    cnt = cnt < 0 ? 0 : cnt;
    if (!__ballot(cnt > 0))
        return;
    double *__restrict__ tx = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + KIND) * (size_t)p.ntiles;
    /* A turn of the loop below costs the wavefront the same whether one lane or all of them have a tile to write: fine where the
     * lanes' rows are alike (neighbouring laps of one chain: each has its dozen tiles).  A FEW lanes with very long rows — a slow
     * chain, hundreds of tiles in one row, beside lanes that have none — are written by the whole wavefront instead, a lane per
     * tile (which costs ~40 instructions per such row to set up: not worth it for rows every lane has). */
    unsigned long long big = __ballot(cnt > 24);
    if (big && __popcll(big) <= 6) {
        const int lane = (int)__lane_id();
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const int t0 = __builtin_amdgcn_readlane(t, src), c = __builtin_amdgcn_readlane(cnt, src);
            const int nn = __builtin_amdgcn_readlane(n, src);
            const double xx = bits_f64(readlane_u64(f64_bits(x), src)), SS = bits_f64(readlane_u64(f64_bits(S), src));
            const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)bits, src);
            double *txs = (double *)readlane_u64((uint64_t)tx, src);
            uint32_t *tns = (uint32_t *)readlane_u64((uint64_t)(p.tile_nav + ((size_t)b * (size_t)p.nch + i) * (size_t)p.ntiles), src);
            for (int q = lane; q < c; q += 64) {
                const int tt = t0 + q;
                const double v = __fma_rn((double)(tt * TILE - nn), SS, xx);
                txs[tt] = KIND == NCO_CARR ? mul_rn(v, 512.0) : v;
                if (!WIDE && KIND == NCO_CODE)
                    tns[tt] = bb;
            }
            if (lane == src)
                cnt = 0;
        }
    }
    /* the state at the row's first tile start, then 1024 steps further per tile: both exact (states of the row: fma(j, S, x) is) */
    double v = __fma_rn((double)(t * TILE - n), S, x);
    const double dv = mul_rn(S, (double)TILE); /* exact: a power of two */
    double *px = tx + t;
    if (!WIDE) {
        /* (geometries whose rows hold a tile at most — the reference's 2.6 MS/s: a code period is 2.5 tiles — keep the plain loop:
         * there the pieces below find nothing to join and cost pass 2 its eighth wavefront per SIMD: - 1.8 % on that leg) */
        uint32_t *pn = p.tile_nav + ((size_t)b * (size_t)p.nch + i) * (size_t)p.ntiles + t;
        for (; __ballot(cnt > 0); cnt--) {
            if (cnt > 0) {
                *px = KIND == NCO_CARR ? mul_rn(v, 512.0) : v;
                if (KIND == NCO_CODE)
                    *pn = bits;
            }
            v = add_rn(v, dv);
            px++;
            pn++;
        }
        return;
    }
    /* Where a row holds a whole 32-byte sector of the block's row of tile states (four tiles from a 32-byte boundary on: the long
     * rows of the upper binades), the four go out in one piece.  Stored one by one as they come, every state leaves a sector dirty
     * that is evicted long before its neighbours arrive (the open lines of the wavefronts in flight are ten times the L2):
     * WRITE_SIZE 2 - 4 x the bytes — and what the pre-pass's write traffic costs shows when it is doubled
     * (profiles/r06_sweeps.txt, GPSBB_X_TILE_DUP: the pre-pass of a push 3.8 -> 7.8 ms, the synthesis beside it 1.73 -> 1.93 ms). */
    typedef double lap_d2 __attribute__((ext_vector_type(2)));
    while (__ballot(cnt > 0)) {
        const bool quad = cnt >= 4 && (reinterpret_cast<uintptr_t>(px) & 31u) == 0;
        const bool pair = !quad && cnt >= 2 && (reinterpret_cast<uintptr_t>(px) & 15u) == 0;
        const bool one = cnt > 0 && !quad && !pair;
        double v3 = v;
        if (__ballot(pair)) {
            if (pair) {
                v3 = add_rn(v, dv);
                lap_d2 *q = reinterpret_cast<lap_d2 *>(px);
                q[0] = KIND == NCO_CARR ? lap_d2{mul_rn(v, 512.0), mul_rn(v3, 512.0)} : lap_d2{v, v3};
            }
        }
        if (__ballot(quad)) {
            if (quad) {
                const double v1 = add_rn(v, dv), v2 = add_rn(v1, dv);
                v3 = add_rn(v2, dv);
                lap_d2 *q = reinterpret_cast<lap_d2 *>(px);
                if (KIND == NCO_CARR) {
                    q[0] = lap_d2{mul_rn(v, 512.0), mul_rn(v1, 512.0)};
                    q[1] = lap_d2{mul_rn(v2, 512.0), mul_rn(v3, 512.0)};
                } else {
                    q[0] = lap_d2{v, v1};
                    q[1] = lap_d2{v2, v3};
                }
            }
        }
        if (one)
            *px = KIND == NCO_CARR ? mul_rn(v, 512.0) : v;
        const int adv = quad ? 4 : (pair ? 2 : (one ? 1 : 0));
        v = add_rn(quad || pair ? v3 : v, dv);
        px += adv;
        cnt -= adv;
    }
}

/*
 * The lanes in `go` — all of one direction — walk in lockstep, as walk_lockstep: every turn is a regular run of the exact
 * jump-ahead (possibly of no steps) followed by one genuine IEEE step, until each lane has wrapped (outcome LAP_OUT_WRAP: (n, x)
 * is the first sample of the next lap and its state), stands at the end of its territory without having wrapped
 * (LAP_OUT_LATE), or stands at its block's last sample + 1 (outcome 0, n == nsamp: the caller's).
 */
constexpr int LAP_BURST = 16;
template <int KIND, bool SNEG, bool EMIT, bool TIES, bool WIDE>
__device__ __forceinline__ void lap_run(const BatchDev &p, int i, LapLane<KIND> &w, const bool was, const int L_burst)
{
    constexpr int TOPEX = LapK<KIND>::TOPEX;
    constexpr int TOP = KIND == NCO_CARR ? 1022 : 1023 + 9; /* the top binade: [0.5, 1) / [512, 1024) */
    const double s = w.s;
    const int es = w.es, nmax = w.nmax;
    const bool any_tie = __ballot(was && w.tiemask != 0ull) != 0ull;
    const bool any_tt = TIES && __ballot(was && w.tt) != 0ull;
    /* (bursts only where every lane's step is small enough for them — the lanes of a wavefront share a channel, mostly a block) */
    const bool burst_ok = L_burst != 0 && !__ballot(was && es > TOP - 8);
    double x = w.x;
    int n = w.n;
    bool go = was, last_wrapped = false;
    /* A step of exactly zero (round 6; until then one such channel sent its whole batch to the row walks): c:2741 adds nothing, the
     * phase stands still to the end of the block or of the territory and every tile of the stretch starts from it — but for a phase
     * of exactly 1.0, which the first step takes to 0.0 like any other (c:2743: one ordinary turn below, then it stands). */
    const bool any_still = KIND == NCO_CARR && !SNEG && __ballot(was && s == 0.0) != 0ull;
    while (__ballot(go)) {
        if (any_still) {
            const bool still = go && s == 0.0 && x < 1.0;
            if (__ballot(still)) {
                const int k = still && nmax > n ? nmax - n : 0;
                if (EMIT && p.tile_x) {
                    const int t0 = (int)(((uint32_t)n + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
                    if (__ballot(still && t0 * TILE <= n + k && t0 < p.ntiles))
                        lap_emit_row<KIND, WIDE>(p, i, still, w.b, n, k, x, 0.0, w.bits);
                }
                if (still) {
                    n += k;
                    last_wrapped = k > 0 ? false : last_wrapped;
                    go = false;
                }
                continue;
            }
        }
#ifdef GPSBB_LAP_DEBUG /* (how many turns does a wavefront take?  hazards[3] is scratch) */
        if (__lane_id() == (unsigned)__builtin_ctzll(__ballot(go)))
            atomicAdd(p.hazards + 3, 1ull);
#endif
        const uint32_t hi = (uint32_t)__double2hiint(x);
        const int ex = (int)(hi >> 20);
        const int d = ex - es;
        /* Where the state is within a few binades of the step's own — the start of a rising lap, the end of a falling one —
         * regular runs are a step or two long and a turn buys next to nothing: LAP_BURST plain steps of the recurrence itself
         * instead (genuine IEEE adds: nothing to prove), for the lanes that have that many before anything happens — no wrap (a
         * rising state below 16 steps ends below 32 of them, short of the top binade: the links' bookkeeping up there never sees
         * a burst; a falling one stops before the step that would take it below zero), no tile's first sample, no end of block or
         * territory.  The lanes of a wavefront are neighbouring laps of one chain and start theirs together; a burst for a few
         * stragglers costs the others more than it saves them: it takes 1 / L_burst of the lanes still walking.  (A negative
         * state has its sign in ex: d is then large.) */
        if (burst_ok && __ballot(go && d < 4)) {
            const int r = n & (TILE - 1);
            const bool clear = n + LAP_BURST <= nmax && (!(EMIT && p.tile_x) || (r != 0 && r + LAP_BURST <= TILE));
            bool bq = go && d < 4 && clear;
            if (SNEG)
                bq = bq && x >= -s; /* (a state below one step wraps with the next: the turn's business) */
            const unsigned long long bm = __ballot(bq), gm = __ballot(go);
            if (bm && L_burst * __popcll(bm) >= __popcll(gm)) {
                if (!SNEG) {
                    if (bq) {
#pragma unroll
                        for (int j = 0; j < LAP_BURST; j++)
                            x = add_rn(x, s);
                        n += LAP_BURST;
                        last_wrapped = false;
                    }
                } else {
                    bool on = bq;
#pragma unroll
                    for (int j = 0; j < LAP_BURST; j++) {
                        const double x2 = add_rn(x, s);
                        on = on && x2 >= 0.0;
                        x = on ? x2 : x;
                        n += on ? 1 : 0;
                    }
                    if (bq)
                        last_wrapped = false; /* (it took at least one step: x >= |s|) */
                }
                go = go && n < nmax;
                continue;
            }
        }
        const bool weird = (unsigned)(ex - 1) >= (unsigned)(TOPEX - 1); /* zero, subnormal, negative, at or beyond the top */
        bool expl = weird || d < 2;
        if (any_tie)
            expl |= ((w.tiemask >> (d & 63)) & 1ull) != 0ull && (__double2loint(x) & 1);
        /* S = s rounded to a multiple of ulp(x), ties to even: adding and subtracting 1.5 * 2^e */
        const double C = __hiloint2double((int)((hi & 0xfff00000u) | 0x80000u), 0);
        const double S = add_rn(add_rn(s, C), -C);
        double room;
        if (!SNEG) {
            double lim = __hiloint2double((int)(hi | 0xfffffu), -1); /* 2^(e+1) - ulp */
            if (KIND == NCO_CODE)
                lim = ex == 1023 + 9 ? 0x1.ff7ffffffffffp+9 /* 1023 - ulp */ : lim;
            room = add_rn(lim, -x);
        } else {
            room = add_rn(x, -__hiloint2double((int)(hi & 0xfff00000u), 1)); /* 2^e + ulp */
        }
        if (any_tt) {
            /* A step that lies exactly half-way between two multiples of the top binade's last place (the coarsest grid: the unit
             * offsets are counted in): every sum up there is a tie.  From an even mantissa a step adds the even neighbour S of the
             * two, from an odd one (once: the sum is even) the other, 2s - S.  A trajectory an odd number of units away has the other
             * parity, so its first step up here differs by 2(s - S): one unit — and from then on the two are an even number apart. */
            if (go && w.tt && !w.so && ex == TOP) {
                const bool pos = add_rn(s, -S) > 0.0, even = !(__double2loint(x) & 1);
                w.so = (even == pos) ? 1 : -1;
            }
        }
        const double Sa = SNEG ? -S : S;
        double rs = __builtin_amdgcn_rcp(Sa);
        rs = __fma_rn(__fma_rn(-Sa, rs, 1.0), rs, rs);
        const double kq = fmin(room * rs, 2147483000.0);
        int ki = (int)kq;
        const double rem = __fma_rn(-(double)ki, Sa, room); /* exact: |rem| < 2|S| */
        ki += (rem < 0.0 ? -1 : 0) + (rem >= Sa ? 1 : 0);
        const int kcap = nmax - n;
        const int k = (expl || !(room >= Sa)) ? 0 : (ki < kcap ? ki : kcap);
        const double x1 = __fma_rn((double)k, S, x);
        if (KIND == NCO_CARR && __ballot(go && weird)) {
            if (go && ex >= TOPEX && !(hi >> 31))
                w.hz++; /* carr_phase == 1.0 at this sample: table index 512 (gpsbb_hazards_t.itable_512) */
        }
        const int n1 = n + k;
        if (EMIT && p.tile_x) { /* (no tile states where only the chain is wanted: gpsbb_chain_carrier) */
            /* does a tile start inside the row (samples n .. n1)? */
            const int t0 = (int)(((uint32_t)n + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
            if (__ballot(go && t0 * TILE <= n1 && t0 < p.ntiles))
                lap_emit_row<KIND, WIDE>(p, i, go, w.b, n, k, x, S, w.bits);
        }
        const bool step = go && n1 < nmax;
        double x2 = add_rn(x1, s);
        bool wrapped;
        double xw;
        if (KIND == NCO_CARR) {
            wrapped = SNEG ? x2 < 0.0 : x2 >= 1.0; /* c:2743-2746 */
            xw = add_rn(x2, SNEG ? 1.0 : -1.0);
        } else {
            wrapped = x2 >= 1023.0;
            xw = add_rn(x2, -1023.0); /* c:2711-2712 */
        }
        wrapped = wrapped && step;
        if ((TIES || KIND == NCO_CODE) && __ballot(wrapped)) {
            int eo1 = 0, eo2 = 0, eo3 = 0; /* what this wrap adds to an offset that is 1, 2, 3 mod 4, on top of carrying it along */
            if (TIES && KIND == NCO_CARR && wrapped && SNEG) {
                /* was x2 + 1.0 exactly half-way between two multiples of 2^-53?  (Fast2Sum: both differences are exact.)  It then went
                 * to the even one; a trajectory an odd number of grid steps away goes to the other side of ITS half-way point */
                const double err = add_rn(add_rn(xw, -1.0), -x2);
                if (fabs(err) == 0x1p-54)
                    eo1 = eo3 = err < 0.0 ? 1 : -1;
            }
            if (TIES && KIND == NCO_CARR && wrapped && !SNEG) {
                /* The sum that passes 1.0 is rounded on the grid of [1, 2): 2^-52, two units.  A rising chain's offsets are even — its
                 * post-wrap states are multiples of 2^-52 — except in the lap in which a falling phase turned round (the step changed
                 * sign with the block): an odd offset then comes out one unit further or nearer, by the side of the grid point the
                 * exact sum lies on.  And steps exist (one block-channel in 2^12) whose sums here are exactly half-way every other
                 * wrap: it went to the even grid point; two units further on (an offset that is 2 mod 4) the even one is the other. */
                const double err = add_rn(add_rn(x2, -x1), -s); /* the sum as rounded minus the exact sum, within a unit (2^-53) */
                if (err != 0.0) {
                    eo1 = eo3 = err < 0.0 ? 1 : -1;
                    eo2 = fabs(err) == 0x1p-53 ? (err < 0.0 ? 2 : -2) : 0;
                } else {
                    /* the sum is a grid point a: an odd offset lands half-way between two and takes the even one */
                    const bool a_even = !(__double2loint(x2) & 1);
                    eo1 = a_even ? -1 : 1;
                    eo3 = a_even ? 1 : -1;
                }
            }
            if (TIES && wrapped && (eo1 | eo2 | eo3 | w.so)) {
                /* this lap's part of the link: first the step into / in the top binade (so), then the wrap */
                LapMap F = lap_identity(), E = lap_identity();
                F.o1 = F.o3 = w.so;
                E.o1 = eo1;
                E.o2 = eo2;
                E.o3 = eo3;
                w.acc = lap_compose(E, lap_compose(F, w.acc));
                w.so = 0;
            }
            if (KIND == NCO_CODE && EMIT && WIDE && p.tile_x)
                lap_nav_out<KIND>(p, i, w, wrapped, n1 + 1, w.bits); /* the period's tiles: up to the one the next period's first sample starts */
            if (KIND == NCO_CODE && wrapped) {
                /* a code period is over (c:2714-2733): the data bits of the next one, should the lane go on into it — and its
                 * roll-over's data-bit fetch (past dwrd[59]?  the fetch of the roll-over that ENDS the lane's walk is the next lane's
                 * to count, or — on a block's last step — lap_walk's) */
                w.jc++;
                if (n1 + 1 < nmax && !w.one_lap) {
                    const uint32_t c = w.c0 + w.jc;
                    w.bits = lap_code_bits(p.ch[(size_t)w.b * p.nch + i].dwrd, c);
                    if (c % 20u == 0u && c / 600u >= (uint32_t)GPSBB_N_DWRD)
                        w.hz++;
                }
            }
        }
        if (any_tt) {
            /* ... or the step INTO the top binade from below is the tie (a state on the finer grid below plus a half-way step is
             * half-way exactly when that state is a multiple of the top grid): both trajectories then arrive with even mantissas, an
             * odd offset has already moved by one unit, to the side opposite the one the reference was rounded to */
            if (step && w.tt && !w.so && !wrapped && ex < TOP && (int)((uint32_t)__double2hiint(x2) >> 20) == TOP) {
                const double err = add_rn(add_rn(x2, -x1), -s);
                if (fabs(err) == (KIND == NCO_CARR ? 0x1p-54 : 0x1p-44))
                    w.so = err < 0.0 ? 1 : -1;
            }
        }
        if (go) {
            x = step ? (wrapped ? xw : x2) : x1;
            n = step ? n1 + 1 : n1;
        }
        last_wrapped = go ? wrapped : last_wrapped;
        /* a lane holds several laps (LapDev::unit): a wrap inside its territory is just another step */
        go = go && n < nmax && !(wrapped && w.one_lap);
    }
    if (was) {
        w.x = x;
        w.n = n;
        if (last_wrapped)
            w.outcome = LAP_OUT_WRAP;
        else if (w.b == w.bt)
            w.outcome = LAP_OUT_LATE;
    }
}

/* the end-of-block state a lane leaves when it arrives at the block's last sample + 1 (c:2709-2746's live-out) */
template <int KIND>
__device__ __forceinline__ void lap_block_end(const BatchDev &p, int i, LapLane<KIND> &w, bool emit)
{
    const size_t k = (size_t)w.b * p.nch + i;
    if (KIND == NCO_CARR) {
        if (emit) {
            if (p.lap_end)
                p.lap_end[k] = w.x; /* (the chain alone: gpsbb_chain_carrier) */
            else
                p.end[k].carr_phase = w.x;
        }
    } else {
        const gpsbb_chan_t &ch = p.ch[k];
        const uint32_t c = w.c0 + w.jc;
        const uint32_t nav = lap_nav_of(c);
        if (emit) {
            gpsbb_chan_state_t &e = p.end[k];
            e.code_phase = w.x;
            e.iword = nav_iword(nav);
            e.ibit = nav_ibit(nav);
            e.icode = nav_icode(nav);
            e.dataBit = nav_bit(ch.dwrd, nav);
            const int ci = (int)w.x;
            e.codeCA = (int)((p.ca_bits[ch.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1; /* c:2737 */
            e._pad = 0;
        }
    }
}

/*
 * Walk every active lane from (b, n, x) to the end of its lap or of its territory, whichever comes first.  The lanes of the
 * wavefront belong to one channel i.  EMIT: leave tile states, end-of-block states; count hazards either way (w.hz).
 */
template <int KIND, bool EMIT, bool TIES, bool WIDE = false>
__device__ __forceinline__ void lap_walk(const BatchDev &p, const LapDev &L, int i, LapLane<KIND> &w)
{
    w.outcome = 0;
    w.so = 0;
    w.acc = lap_identity();
    w.hz = 0;
    if (w.active && w.b >= p.nblocks) { /* (a plan that put a lap past the last block: its link will not hold) */
        w.outcome = LAP_OUT_LATE;
        w.active = false;
    }
    if (w.active) {
        lap_enter_block<KIND>(p, L, i, w);
        if (KIND == NCO_CODE) {
            const uint32_t c = w.c0 + w.jc;
            w.bits = lap_code_bits(p.ch[(size_t)w.b * p.nch + i].dwrd, c);
            /* the roll-over that started this lap fetched a data bit (c:2732) if it started a bit: past dwrd[59]? */
            if (w.fresh && w.jc > 0 && c % 20u == 0u && c / 600u >= (uint32_t)GPSBB_N_DWRD)
                w.hz++;
            w.navt = (int32_t)(((uint32_t)w.n + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
        }
        if (w.n >= w.nmax && w.b == w.bt) { /* an empty territory (the plan put two laps on one sample) */
            w.outcome = LAP_OUT_LATE;
            w.active = false;
        }
    }
    const bool walked = w.active;
    while (__ballot(w.active)) {
        /* the lanes by the sign of their step, each group in its own straight-line loop (a wavefront's laps are neighbours in time:
         * it normally runs only one of the two) */
        const bool rise = w.active && !w.neg, fall = w.active && w.neg;
        if (__ballot(rise))
            lap_run<KIND, false, EMIT, TIES, WIDE>(p, i, w, rise, L.burst);
        if (KIND == NCO_CARR && __ballot(fall))
            lap_run<KIND, true, EMIT, TIES, WIDE>(p, i, w, fall, L.burst);
        /* lanes at the last sample + 1 of their block: the end state; the chain's next block, or the walk ends */
        const bool at_end = w.active && w.n >= p.nsamp;
        if (__ballot(at_end)) {
            if (at_end) {
                if (KIND == NCO_CODE && w.outcome == LAP_OUT_WRAP) {
                    /* a roll-over on the block's last step: its data-bit fetch belongs to this lap (no lap starts there) */
                    const uint32_t c = w.c0 + w.jc;
                    if (c % 20u == 0u && c / 600u >= (uint32_t)GPSBB_N_DWRD)
                        w.hz++;
                }
                lap_block_end<KIND>(p, i, w, EMIT);
                const bool goes_on = KIND == NCO_CARR && (w.bcflags & LAPF_CONT) && w.b + 1 < p.nblocks;
                if (w.outcome == 0) {
                    if (goes_on) {
                        w.b++;
                        w.n = 0;
                        lap_enter_block<KIND>(p, L, i, w);
                        if (w.b == w.bt && w.nt <= 0)
                            w.outcome = LAP_OUT_LATE; /* the territory ended with the block */
                    } else {
                        w.outcome = LAP_OUT_CHAIN;
                    }
                } else if (w.outcome == LAP_OUT_WRAP) {
                    if (goes_on) {
                        w.b++; /* canonical position of the next lap's first sample: (b + 1, 0) */
                        w.n = 0;
                        if (!(w.b == w.bt && w.nt == 0) && !w.one_lap) {
                            /* a wrap on a block's last step INSIDE the lane's territory (a lane holds several laps): on it goes */
                            w.outcome = 0;
                            lap_enter_block<KIND>(p, L, i, w);
                        }
                    } else {
                        w.outcome = LAP_OUT_CHAIN; /* the chain ends with a wrap on its last step: no lap follows */
                    }
                }
            }
        }
        w.active = w.active && w.outcome == 0;
    }
    if (KIND == NCO_CODE && EMIT && WIDE && p.tile_x)
        lap_nav_out<KIND>(p, i, w, walked, w.n, w.bits); /* what is left of the lane's last period */
}

/* ---- k_lap_plan ---------------------------------------------------------------------------------------------- */

struct LapScanEl {
    int reset;
    int K;
    double phi;
};
__device__ __forceinline__ LapScanEl lap_scan_combine(const LapScanEl &left, const LapScanEl &right, double range)
{
    if (right.reset)
        return right;
    LapScanEl r;
    r.reset = left.reset;
    double f = left.phi + right.phi;
    int K = left.K + right.K;
    if (f >= range) {
        f -= range;
        K++;
    }
    r.K = K;
    r.phi = f;
    return r;
}

/* the model's advance over n steps of step s with a drift of ds per step, as whole laps and a fraction in [0, range) */
__device__ __forceinline__ void lap_advance(double s, double ds, int n, double range, int &K, double &f)
{
    const double nd = (double)n;
    const double hi = nd * s;
    const double lo = __fma_rn(nd, s, -hi) + nd * ds;
    double q = floor(hi / range);
    double fr = __fma_rn(-q, range, hi) + lo; /* exact product for range 1 and for 1023 * (an integer below 2^40) */
    if (fr < 0.0) {
        fr += range;
        q -= 1.0;
    }
    if (fr >= range) {
        fr -= range;
        q += 1.0;
    }
    K = (int)q;
    f = fr;
}

/* One wavefront per kind and channel: the model state at every block's first sample (a segmented scan over the blocks: a
 * chain's first block starts from its exact state), the laps that start in every block, their lanes.  Also what idle
 * channels leave (end states) and, kind 0, the tile counters of the table set (as k_tiles did). */
template <int KIND>
__device__ __forceinline__ void lap_plan_body(const BatchDev &p, const LapDev &L, const uint32_t bx)
{
    const int i = (int)bx, lane = threadIdx.x;
    if (i >= p.nch)
        return;
    const double range = KIND == NCO_CARR ? 1.0 : 1023.0;
    const bool fixed = p.kph0 != nullptr;
    LapBC *bcs = L.bc + ((size_t)KIND * p.nch + i) * (size_t)p.nblocks;
    uint32_t *lane0 = L.lane0 + ((size_t)KIND * p.nch + i) * ((size_t)p.nblocks + 1);
    LapScanEl carry; /* the state at the first sample of block b0 (exclusive of b0's own element) */
    carry.reset = 1;
    carry.K = 0;
    carry.phi = 0.0;
    uint32_t lanes_before = 0;
    double carry_c = 0.0; /* the chain's drift corrections up to the first block of the group (see c_own below) */
    if (KIND == NCO_CODE && i == 0)
        for (int b = lane; b <= p.nblocks; b += 64) /* ([nblocks]: the helpers' ticket counter, ev_pick_block) */
            p.tile_ctr[b] = 0;
    for (int b0 = 0; b0 < p.nblocks; b0 += 64) {
        const int b = b0 + lane;
        const bool in = b < p.nblocks;
        const size_t k = (size_t)(in ? b : 0) * p.nch + i;
        /* the chain alone (gpsbb_chain_carrier, carriers only): 24-byte chain descriptors instead of the 296-byte ones */
        const bool slim = KIND == NCO_CARR && p.ch == nullptr;
        int prn = 0;
        double f_of = 0.0, phase_of = 0.0;
        uint32_t c0_of = 0;
        if (slim) {
            prn = in ? p.cd[k].prn : 0;
            f_of = p.cd[k].f_carr;
            phase_of = p.cd[k].carr_phase;
        } else {
            const gpsbb_chan_t &ch = p.ch[k];
            prn = in ? ch.prn : 0;
            f_of = KIND == NCO_CARR ? ch.f_carr : ch.f_code;
            phase_of = KIND == NCO_CARR ? ch.carr_phase : ch.code_phase;
            c0_of = (uint32_t)ch.icode + 20u * (uint32_t)ch.ibit + 600u * (uint32_t)ch.iword;
        }
        const bool act = prn > 0 && !(KIND == NCO_CARR && fixed);
        const double s = act ? mul_rn(f_of, p.delt) : 0.0;
        /* the drift of one whole lap with this step, and — first — the model as if a block's drift were its share of that by the
         * samples it holds (linear in the sample count: what the scan below can add up); the stretch of a lap a block starts and
         * ends with is put right after it */
        double R1 = 0.0;
        if (act)
            (void)lap_R<KIND>(s, 0.0, R1);
        const double ds = fabs(s) * R1 * (1.0 / range);
        /* does this block continue the one before / go on into the next? */
        bool cont_in = false, cont_out = false;
        if (KIND == NCO_CARR && act && L.chained) {
            if (b > 0)
                cont_in = (slim ? p.cd[k - p.nch].prn : p.ch[k - p.nch].prn) == prn;
            else
                cont_in = p.carry && ((p.cont0_mask >> i) & 1u);
            cont_out = b + 1 < p.nblocks && (slim ? p.cd[k + p.nch].prn : p.ch[k + p.nch].prn) == prn;
        }
        const bool head = act && !cont_in;
        double start = 0.0;
        if (head)
            start = phase_of;
        if (KIND == NCO_CARR && act && b == 0 && cont_in)
            start = p.carry->exact_end[i]; /* a stream: where the push before this one ended, exactly */
        const bool known = act && (head || (b == 0 && cont_in)); /* the block's first state is known exactly */
        /* this block's advance, for the block after it */
        int aK = 0;
        double af = 0.0;
        if (act)
            lap_advance(s, ds, p.nsamp, range, aK, af);
        /* element of block b = what turns the state at block b-1's first sample into the one at b's: b-1's advance, or a reset */
        LapScanEl el;
        const int aK_prev = __shfl_up(aK, 1);
        const double af_prev = __shfl_up(af, 1);
        el.reset = known || !act ? 1 : 0;
        el.K = 0;
        el.phi = known ? (start >= range ? 0.0 : start) : 0.0;
        /* (the advance of the last block of the group before this one travels in `carry`) */
        if (!el.reset) {
            el.K = lane > 0 ? aK_prev : 0;
            el.phi = lane > 0 ? af_prev : 0.0;
        }
        LapScanEl acc = el;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            LapScanEl prev;
            prev.reset = __shfl_up(acc.reset, off);
            prev.K = __shfl_up(acc.K, off);
            prev.phi = __shfl_up(acc.phi, off);
            if (lane >= off)
                acc = lap_scan_combine(prev, acc, range);
        }
        acc = lap_scan_combine(carry, acc, range);
        /* acc = the model state at block b's first sample, as (laps since the chain's first sample, fraction) — to first order.
         * What a block really drifts by is not its share of whole laps' drift: it starts and ends somewhere INSIDE a lap, and the
         * drift per step depends on the binade the phase is in (a slow carrier's block covers a fraction of one lap, all of it
         * in one or two binades).  c = the block's drift by the model (lap_R) minus its linear share, from the first-order start
         * phase (the difference this makes to c is of second order); a second scan sums the c of the chain's blocks so far. */
        const double phi0 = known ? start : acc.phi;
        double c_own = 0.0;
        if (act) {
            int k0 = 0;
            double f0 = 0.0, r1 = 0.0;
            lap_advance(s, ds, p.nsamp, range, k0, f0);
            double xe = (phi0 >= range ? 0.0 : phi0) + f0; /* where the block ends, first order */
            int wraps = s < 0.0 ? -k0 : k0;
            if (xe >= range) {
                xe -= range;
                wraps += s < 0.0 ? -1 : 1;
            }
            const double Rs = lap_R<KIND>(s, phi0 >= range ? range : phi0, r1), Re = lap_R<KIND>(s, xe, r1);
            const double Dtrue = (double)(wraps < 0 ? -wraps : wraps) * R1 + (s < 0.0 ? Rs - Re : Re - Rs);
            c_own = Dtrue - (double)p.nsamp * ds;
        }
        /* exclusive segmented sum of c over the chain's blocks before this one (lane - 1's c: shifted in) */
        double cs = __shfl_up(c_own, 1);
        int cf = el.reset; /* a block whose start is known starts the sum afresh */
        cs = (lane == 0 || el.reset) ? 0.0 : cs;
        /* (the c of the last block of the group before this one travels in carry_c) */
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double vu = __shfl_up(cs, off);
            const int fu = __shfl_up(cf, off);
            if (lane >= off) {
                cs = cf ? cs : cs + vu;
                cf = cf | fu;
            }
        }
        const double corr = cf ? cs : cs + carry_c;
        double phi = known ? start : acc.phi + corr;
        int dK = 0; /* (a correction across a lap boundary moves the block's lap count by one) */
        if (!known && phi >= range) {
            phi -= range;
            dK = 1;
        }
        if (!known && phi < 0.0) {
            phi += range;
            dK = -1;
        }
        (void)dK;
        /* where the block ends — or, for a chain's last block, where its last step starts: a wrap on that step starts no lap */
        int eK = 0;
        double ef = 0.0;
        uint32_t nl = 0;
        if (act) {
            const bool to_end = KIND == NCO_CARR && cont_out;
            lap_advance(s, ds, to_end ? p.nsamp : p.nsamp - 1, range, eK, ef);
            double f = (known && start >= range ? 0.0 : phi) + ef + c_own;
            if (f >= range) {
                f -= range;
                eK++;
            }
            if (f < 0.0) {
                f += range;
                eK--;
            }
            /* rising: the levels reached; falling: the levels passed (floor of the unwrapped phase either way) */
            int W = s < 0.0 ? -eK : eK;
            W = W < 0 ? 0 : W;
            /* (never more laps than samples they could start at: a one-sample block that does not go on has none but its head's,
             * whatever the model makes of a phase of exactly 0 that falls) */
            const int wmax = to_end ? p.nsamp : p.nsamp - 1;
            W = W > wmax ? wmax : W;
            /* (unit 0: the whole chain of a block whose first state is known is its head's — no lap of its own starts in it) */
            if (L.unit[KIND] == 0 && (head || known))
                W = 0;
            const uint32_t un = L.unit[KIND] > 0 ? (uint32_t)L.unit[KIND] : 1u;
            nl = ((uint32_t)W + un - 1u) / un + (head || known ? 1u : 0u);
        }
        /* lanes: an exclusive sum over the blocks */
        uint32_t incl = nl;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off)
                incl += v;
        }
        if (in) {
            LapBC o;
            o.phi = phi;
            o.s = s;
            o.R1 = R1;
            {
                double r1 = 0.0;
                o.Rphi = act ? lap_R<KIND>(s, phi >= range ? range : phi, r1) : 0.0;
            }
            o.flags = (act ? LAPF_ACTIVE : 0u) | (known ? LAPF_HEAD : 0u) | (cont_out ? LAPF_CONT : 0u);
            o.c0 = KIND == NCO_CODE && act ? c0_of : 0u;
            bcs[b] = o;
            lane0[b] = lanes_before + incl - nl;
            if (!act && slim) {
                p.lap_end[k] = 0.0;
            } else if (!act) {
                /* what an idle channel leaves behind (as k_walk) */
                gpsbb_chan_state_t &e = p.end[k];
                if (KIND == NCO_CARR) {
                    e.carr_phase = fixed && prn > 0 ? (double)(uint32_t)(p.kph0[k] + (uint32_t)p.nsamp * (uint32_t)p.kstep[k]) : 0.0;
                } else {
                    e.code_phase = 0.0;
                    e.iword = e.ibit = e.icode = e.dataBit = e.codeCA = 0;
                    e._pad = 0;
                }
            }
        }
        lanes_before += (uint32_t)__shfl((int)incl, 63);
        /* the state at the first sample of the next group's first block */
        LapScanEl last;
        last.reset = __shfl(acc.reset, 63);
        last.K = __shfl(acc.K, 63);
        last.phi = __shfl(known ? (start >= range ? 0.0 : start) : acc.phi, 63);
        LapScanEl adv;
        adv.reset = 0;
        adv.K = __shfl(aK, 63);
        adv.phi = __shfl(af, 63);
        carry = lap_scan_combine(last, adv, range);
        carry_c = __shfl(corr + c_own, 63);
    }
    if (lane == 0) {
        lane0[p.nblocks] = lanes_before;
        L.nlaps[KIND * GPSBB_MAX_CHAN + i] = lanes_before;
        L.nbad[KIND * GPSBB_MAX_CHAN + i] = 0;
        const uint32_t room = (L.chunk0[KIND][i + 1] - L.chunk0[KIND][i]) * (uint32_t)LAP_WG;
        if (lanes_before > room) {
            atomicOr(p.status, ST_LAP_PLAN);
            L.nlaps[KIND * GPSBB_MAX_CHAN + i] = 0; /* nothing is walked: the status word says why the output is void */
        }
    }
}

/* fixed-point carrier: nothing to walk — the table index at every tile start in closed form (as k_tiles did) */
__global__ __launch_bounds__(256) void k_lap_fixed_tiles(BatchDev p)
{
    const int bi = blockIdx.x;
    const int b = bi / p.nch, i = bi % p.nch;
    if (p.ch[bi].prn <= 0)
        return;
    double *__restrict__ txf = p.tile_x + ((size_t)b * (2 * (size_t)p.nch) + 2 * i + 1) * (size_t)p.ntiles;
    for (int t = threadIdx.x; t < p.ntiles; t += blockDim.x)
        txf[t] = fixed_tile_index(p.kph0[bi], p.kstep[bi], t);
}

/* ---- the two passes ------------------------------------------------------------------------------------------ */

/* the channel whose range holds chunk c (wave-uniform) */
template <int KIND>
__device__ __forceinline__ int lap_channel_of(const LapDev &L, int nch, uint32_t c)
{
    int i = 0;
    while (i + 1 < nch && c >= L.chunk0[KIND][i + 1])
        i++;
    return i;
}

/* what the wavefronts of a pass hand each other: a few hundred bytes (the passes have to fit on a CU beside a workgroup of the
 * synthesis kernel, which leaves 10 KB of LDS: anything bigger waits for one to leave) */
constexpr int LAP_WAVES = LAP_WG / 64;
struct LapPassLds {
    double A[LAP_WAVES + 1]; /* the first lap of every wavefront, and of the next chunk: where the lap before it ends */
    double m[LAP_WAVES + 1];
    int32_t b[LAP_WAVES + 1], n0[LAP_WAVES + 1];
    uint32_t head[LAP_WAVES + 1];
    double wg[LAP_WAVES];
    uint32_t wo[LAP_WAVES];
    int wc[LAP_WAVES];
};
/* lane + 1's value; the last lane of a wavefront takes `edge` */
__device__ __forceinline__ double lap_next(double v, double edge, int lane)
{
    const double d = __shfl_down(v, 1);
    return lane == 63 ? edge : d;
}
__device__ __forceinline__ int32_t lap_next(int32_t v, int32_t edge, int lane)
{
    const int32_t d = __shfl_down(v, 1);
    return lane == 63 ? edge : d;
}

/* a lane set up for the lap that starts at st, its territory ending where nx starts (has_next) or with its chain */
template <int KIND>
__device__ __forceinline__ LapLane<KIND> lap_lane(bool on, double x, int32_t b, int32_t n0, uint32_t jc, bool has_next, int32_t nb, int32_t nn0)
{
    LapLane<KIND> w;
    w.x = x;
    w.s = 0.0;
    w.b = b;
    w.n = n0;
    w.bt = has_next ? nb : INT32_MAX;
    w.nt = has_next ? nn0 : 0;
    w.nmax = 0;
    w.es = 0;
    w.tiemask = 0ull;
    w.jc = jc;
    w.c0 = 0;
    w.navt = 0;
    w.bits = 0;
    w.hz = 0;
    w.bcflags = 0;
    w.outcome = 0;
    w.so = 0;
    w.tt = false;
    w.one_lap = false;
    w.acc = lap_identity();
    w.active = on;
    w.neg = false;
    w.fresh = true;
    return w;
}

/* code: code periods completed since the block's first sample when lap r of the channel starts (its wrap's number, a head: 0) */
/* (LapStart::jc carries it; pass 2 recomputes it from the plan) */
template <int KIND>
__device__ __forceinline__ uint32_t lap_jc_of(const BatchDev &p, const LapDev &L, int i, uint32_t r, int32_t b_start)
{
    if (KIND != NCO_CODE)
        return 0u;
    const uint32_t *lane0 = L.lane0 + ((size_t)KIND * p.nch + i) * ((size_t)p.nblocks + 1);
    /* every code block starts a chain: lane 0 of the block is its head (no period completed yet), lane j >= 1 starts with the
     * block's wrap number unit * (j - 1), 0-based: that many + 1 periods are over */
    const uint32_t j = r - lane0[b_start];
    return j == 0 ? 0u : (j - 1u) * (uint32_t)(L.unit[KIND] > 0 ? L.unit[KIND] : 1) + 1u;
}

/* the link a reference walk leaves: how the offset of the NEXT lap follows from this lap's */
template <int KIND>
__device__ __forceinline__ LapMap lap_link(const LapLane<KIND> &w, bool mine, bool has_next, double A_next, int32_t nb, int32_t nn0)
{
    if (!mine)
        return lap_identity();
    if (has_next && w.outcome == LAP_OUT_WRAP && w.b == nb && w.n == nn0) {
        const double g = (w.x - A_next) * lap_runit<KIND>(); /* exact: both on the grid, close together */
        if (fabs(g) < 0x1p+50 && g == __builtin_rint(g)) {
            LapMap T = lap_identity(); /* carried along through the lane's laps, and what their wraps and top-binade entries add */
            T.g = g;
            return lap_compose(T, w.acc);
        }
    }
    return lap_const(0.0); /* (a head follows, the chain ended, or the walk did not end where the plan says: the next lap's offset is a guess, 0) */
}

template <int KIND>
__device__ __forceinline__ void lap_pass1_body(const BatchDev &p, const LapDev &L, const uint32_t bx)
{
    __shared__ LapPassLds sh;
    const uint32_t chunk = bx + L.chunk0[KIND][0];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = lap_channel_of<KIND>(L, p.nch, chunk);
    const uint32_t c = chunk - L.chunk0[KIND][i];
    const uint32_t nl = L.nlaps[KIND * GPSBB_MAX_CHAN + i];
    if (c * (uint32_t)LAP_WG >= nl)
        return;
#ifdef GPSBB_WG_TRACE /* measurement build: see wg_trace_leave */
    const WgTrace wgt = wg_trace_enter();
#endif
    const uint32_t r = c * (uint32_t)LAP_WG + (uint32_t)t;
    const bool mine = r < nl;
    LapStart st;
    st.A = 0.0;
    st.b = 0;
    st.n0 = 0;
    st.head = 1;
    st.jc = 0;
    if (mine)
        st = lap_start<KIND>(p, L, i, r);
    const int32_t myhead = mine ? (int32_t)st.head : 1;
    if (lane == 0) {
        sh.A[wave] = st.A;
        sh.b[wave] = st.b;
        sh.n0[wave] = st.n0;
        sh.head[wave] = (uint32_t)myhead;
    }
    if (t == LAP_WG - 1) {
        LapStart nx;
        nx.A = 0.0;
        nx.b = 0;
        nx.n0 = 0;
        nx.head = 1;
        if (r + 1 < nl)
            nx = lap_start<KIND>(p, L, i, r + 1);
        sh.A[LAP_WAVES] = nx.A;
        sh.b[LAP_WAVES] = nx.b;
        sh.n0[LAP_WAVES] = nx.n0;
        sh.head[LAP_WAVES] = nx.head;
    }
    __syncthreads();
    const int32_t next_head = lap_next(myhead, (int32_t)sh.head[wave + 1], lane); /* (every lane takes part in the shuffle) */
    const bool has_next = mine && !next_head;
    const double A_next = lap_next(st.A, sh.A[wave + 1], lane);
    const int32_t nb = lap_next(st.b, sh.b[wave + 1], lane), nn0 = lap_next(st.n0, sh.n0[wave + 1], lane);
    /* (a lane whose chain ends with it — a head follows, or nothing — has no link to find: the next lap's offset is 0 whatever this
     * one does.  Code chains of big batches are such lanes only: one lane per block, LapDev::unit 0) */
    LapLane<KIND> w = lap_lane<KIND>(has_next, st.A, st.b, st.n0, st.jc, has_next, nb, nn0);
    lap_walk<KIND, false, true>(p, L, i, w);
    const LapMap link = lap_link<KIND>(w, mine, has_next, A_next, nb, nn0);
    /* inclusive scan of the links over the chunk */
    LapMap acc = lap_wave_scan(link, lane);
    if (lane == 63) {
        sh.wc[wave] = acc.isconst;
        sh.wg[wave] = acc.g;
        sh.wo[wave] = lap_pack_o(acc);
    }
    __syncthreads();
    for (int q = wave - 1; q >= 0 && !acc.isconst; q--)
        acc = lap_compose(acc, lap_unpack(sh.wg[q], sh.wo[q], sh.wc[q]));
    if (mine) {
        LapRec rec;
        rec.A = st.A;
        rec.g = acc.g;
        rec.ov = lap_pack_o(acc);
        rec._pad = 0;
        rec.b = st.b;
        rec.n0 = st.n0;
        rec.flags = (st.head ? LAPF_HEAD : 0u) | (acc.isconst ? LAPF_CONST : 0u);
        rec.hz = 0;
        L.rec[(size_t)chunk * LAP_WG + t] = rec;
    }
    if (t == LAP_WG - 1) {
        LapAgg a;
        a.g = acc.g;
        a.ov = lap_pack_o(acc);
        a.isconst = (uint32_t)acc.isconst;
        L.agg[chunk] = a;
        L.chunk_bad[chunk] = 0;
    }
#ifdef GPSBB_WG_TRACE
    wg_trace_leave(wgt, wgt.wall0, wgt.clk0, 0u, 3u + (unsigned)KIND, true);
#endif
}

/* the offset of every chunk's first lap: one wavefront per kind and channel composes the chunks' links in order */
template <int KIND>
__device__ __forceinline__ void lap_scan_body(const BatchDev &p, const LapDev &L, const uint32_t bx)
{
    const int i = (int)bx, lane = threadIdx.x;
    if (i >= p.nch)
        return;
    const uint32_t nl = L.nlaps[KIND * GPSBB_MAX_CHAN + i];
    const uint32_t nchunks = (nl + LAP_WG - 1) / LAP_WG;
    const uint32_t base = L.chunk0[KIND][i];
#ifdef GPSBB_LAP_DEBUG
    if (i == 0 && lane == 0)
        printf("kind %d: wave-turns so far %llu; laps of channel 0: %u\n", KIND, p.hazards[3], nl);
#endif
    double m = 0.0; /* the offset of the first lap of chunk q0 (a channel's first lap starts a chain: 0) */
    for (uint32_t q0 = 0; q0 < nchunks; q0 += 64) {
        const uint32_t q = q0 + (uint32_t)lane;
        LapMap my = lap_identity();
        if (q < nchunks) {
            const LapAgg a = L.agg[base + q];
            my = lap_unpack(a.g, a.ov, (int)a.isconst);
        }
        const LapMap acc = lap_wave_scan(my, lane);
        /* chunk q starts where the links of chunks q0 .. q-1 take m */
        const LapMap before = lap_map_shfl_up(acc, 1);
        const double mq = lane == 0 ? m : lap_apply(before, m);
        if (q < nchunks)
            L.chunk_m[base + q] = mq;
        m = lap_apply(lap_map_shfl(acc, 63), m);
    }
}

template <int KIND, bool WIDE = false>
__device__ __forceinline__ void lap_pass2_body(const BatchDev &p, const LapDev &L, const uint32_t bx)
{
    __shared__ LapPassLds sh;
    const uint32_t chunk = bx + L.chunk0[KIND][0];
    const int t = threadIdx.x;
    const int i = lap_channel_of<KIND>(L, p.nch, chunk);
    const uint32_t c = chunk - L.chunk0[KIND][i];
    const uint32_t nl = L.nlaps[KIND * GPSBB_MAX_CHAN + i];
    if (c * (uint32_t)LAP_WG >= nl)
        return;
#ifdef GPSBB_WG_TRACE
    const WgTrace wgt = wg_trace_enter();
#endif
    const uint32_t r = c * (uint32_t)LAP_WG + (uint32_t)t;
    const bool mine = r < nl;
    const double m_first = L.chunk_m[chunk];
    LapRec rec;
    rec.A = 0.0;
    rec.g = 0.0;
    rec.ov = 0;
    rec._pad = 0;
    rec.b = rec.n0 = 0;
    rec.flags = LAPF_HEAD;
    rec.hz = 0;
    if (mine)
        rec = L.rec[(size_t)chunk * LAP_WG + t];
    const LapMap P = lap_unpack(rec.g, rec.ov, (rec.flags & LAPF_CONST) ? 1 : 0);
    const int lane = t & 63, wave = t >> 6;
    const int32_t myhead = mine ? (int32_t)(rec.flags & LAPF_HEAD) : (int32_t)LAPF_HEAD;
    const double m_after = lap_apply(P, m_first); /* the offset of the lap after this one */
    if (lane == 0) {
        sh.A[wave] = rec.A;
        sh.b[wave] = rec.b;
        sh.n0[wave] = rec.n0;
        sh.head[wave] = (uint32_t)myhead;
    }
    if (lane == 63)
        sh.m[wave + 1] = m_after;
    if (t == 0)
        sh.m[0] = m_first;
    if (t == LAP_WG - 1) {
        LapRec nx;
        nx.A = 0.0;
        nx.b = nx.n0 = 0;
        nx.flags = LAPF_HEAD;
        if (r + 1 < nl)
            nx = L.rec[(size_t)(chunk + 1) * LAP_WG];
        sh.A[LAP_WAVES] = nx.A;
        sh.b[LAP_WAVES] = nx.b;
        sh.n0[LAP_WAVES] = nx.n0;
        sh.head[LAP_WAVES] = nx.flags & LAPF_HEAD;
    }
    __syncthreads();
    const bool head = (rec.flags & LAPF_HEAD) != 0;
    const double m_before = __shfl_up(m_after, 1);
    const double m = head ? 0.0 : (lane == 0 ? sh.m[wave] : m_before);
    const int32_t next_head = lap_next(myhead, (int32_t)sh.head[wave + 1], lane); /* (every lane takes part in the shuffle) */
    const bool has_next = mine && !next_head;
    const int32_t nb = lap_next(rec.b, sh.b[wave + 1], lane), nn0 = lap_next(rec.n0, sh.n0[wave + 1], lane);
    /* the true start: exact (the sum is the state itself, a double on the grid) */
    const double x0 = head ? rec.A : __fma_rn(m, lap_unit<KIND>(), rec.A);
    const double x_next = __fma_rn(m_after, lap_unit<KIND>(), lap_next(rec.A, sh.A[wave + 1], lane));
    LapLane<KIND> w = lap_lane<KIND>(mine, x0, rec.b, rec.n0, lap_jc_of<KIND>(p, L, i, r, rec.b), has_next, nb, nn0);
    lap_walk<KIND, true, false, WIDE>(p, L, i, w);
    if (mine) {
        bool ok;
        if (has_next)
            ok = w.outcome == LAP_OUT_WRAP && w.b == nb && w.n == nn0 && f64_bits(w.x) == f64_bits(x_next);
        else
            ok = w.outcome == LAP_OUT_CHAIN;
        LapRec &o = L.rec[(size_t)chunk * LAP_WG + t];
        o.g = m;
        o.hz = w.hz;
#ifdef GPSBB_LAP_DEBUG
        if (!ok)
            printf("bad link kind %d ch %d lap %u/%u head %d has_next %d outcome %d end (%d,%d) want (%d,%d) x %.17g want %.17g m %.0f m_after %.0f start (%d,%d) A %.17g s %.6g\n",
                   KIND, i, r, nl, (int)head, (int)has_next, w.outcome, w.b, w.n, nb, nn0, w.x, x_next, m, m_after, rec.b, rec.n0, rec.A, w.s);
#endif
        if (!ok) {
            o.flags = rec.flags | LAPF_BAD;
            atomicAdd(&L.nbad[KIND * GPSBB_MAX_CHAN + i], 1u);
            L.chunk_bad[chunk] = 1;
        }
        if (w.hz)
            atomicAdd(p.hazards + (KIND == NCO_CARR ? 0 : 1), (unsigned long long)w.hz);
    }
#ifdef GPSBB_WG_TRACE
    wg_trace_leave(wgt, wgt.wall0, wgt.clk0, 0u, 5u + (unsigned)KIND, true);
#endif
}

/* ---- k_lap_repair -------------------------------------------------------------------------------------------- */

/*
 * One wavefront per kind and channel.  Nothing to do unless pass 2 found a broken link in the channel's range.  Then, for the
 * first broken link (every lap before it was walked from its true start, so was the lap that leaves it: its end is the truth):
 *   1. lane 0 walks on from there alone, lap by lap, writing as it goes (these are true states), until a wrap lands on the
 *      first sample of a planned lap q — or the chain ends;
 *   2. if the state there is what pass 2 started lap q from, everything from q on stands: on to the next broken link;
 *   3. else the laps from q on are done again 64 at a time: reference walks, links, a scan from the known offset of lap q,
 *      walks from the true starts (writing), the same check; a link that breaks sends the loop back to 1.
 * At the end (a stream) the channel's exact end phase is what the last block's end state says.
 */
template <int KIND>
__device__ __forceinline__ void lap_repair_body(const BatchDev &p, const LapDev &L, const uint32_t bx)
{
    const int i = (int)bx, lane = threadIdx.x;
    if (i >= p.nch)
        return;
    const uint32_t nl = L.nlaps[KIND * GPSBB_MAX_CHAN + i];
    const uint32_t base = L.chunk0[KIND][i];
    LapRec *recs = L.rec + (size_t)base * LAP_WG;
    unsigned long long n_rewalked = 0, n_links = 0;
    long long hz_delta = 0;
    if (L.nbad[KIND * GPSBB_MAX_CHAN + i] != 0) {
        const uint32_t nchunks = (nl + LAP_WG - 1) / LAP_WG;
        uint32_t from = 0; /* laps before this one are settled */
        /* laps below this one may hold what a group of step 3 wrote BEYOND a link that broke inside the group (walks from starts
         * that were not the truth, into their own territories): what pass 2 left there is gone, so "as pass 2 had it" settles
         * nothing for them — they are done again whatever the state they start from */
        uint32_t dirty_end = 0;
        for (;;) {
            /* the next broken link at or after `from` */
            uint32_t bad = nl;
            for (uint32_t q = from / LAP_WG; q < nchunks && bad == nl; q++) {
                if (!L.chunk_bad[base + q])
                    continue;
                for (uint32_t r0 = q * LAP_WG; r0 < (q + 1) * LAP_WG && bad == nl; r0 += 64) {
                    const uint32_t r = r0 + (uint32_t)lane;
                    const bool isbad = r < nl && r >= from && (recs[r].flags & LAPF_BAD);
                    const unsigned long long mask = __ballot(isbad);
                    if (mask)
                        bad = r0 + (uint32_t)__builtin_ctzll(mask);
                }
            }
            if (bad >= nl)
                break;
            n_links++;
            /* the lap that leaves the broken link, again, from its true start (it wrote the truth already: no need to write) */
            LapRec rb = recs[bad];
#ifdef GPSBB_LAP_DEBUG
            if (lane == 0)
                printf("repair kind %d ch %d: bad link %u of %u (from %u) start (%d,%d) A %.17g g %.0f flags %x\n", KIND, i, bad, nl, from, rb.b, rb.n0, rb.A, rb.g, rb.flags);
#endif
            bool hn = bad + 1 < nl && !(recs[bad + 1 < nl ? bad + 1 : bad].flags & LAPF_HEAD);
            LapRec rn = recs[bad + 1 < nl ? bad + 1 : bad];
            LapLane<KIND> w = lap_lane<KIND>(lane == 0, (rb.flags & LAPF_HEAD) ? rb.A : __fma_rn(rb.g, lap_unit<KIND>(), rb.A), rb.b, rb.n0,
                                             lap_jc_of<KIND>(p, L, i, bad, rb.b), hn, rn.b, rn.n0);
            lap_walk<KIND, false, false>(p, L, i, w);
            /* cur: the true trajectory at the end of that walk (lane 0's copy is the one that counts) */
            double cx = bits_f64(readlane_u64(f64_bits(w.x), 0));
            int32_t cb = __builtin_amdgcn_readlane(w.b, 0), cn = __builtin_amdgcn_readlane(w.n, 0);
            int cout = __builtin_amdgcn_readlane(w.outcome, 0);
            uint32_t cjc = (uint32_t)__builtin_amdgcn_readlane((int)w.jc, 0);
            uint32_t q = bad + 1;
            bool chain_done = cout == LAP_OUT_CHAIN;
            for (;;) {
                /* 1. alone, until a wrap lands on a planned lap's first sample */
                bool synced = false;
                while (!chain_done) {
                    /* planned laps the truth has passed are void (what they counted does not count) */
                    while (q < nl && !(recs[q].flags & LAPF_HEAD) && (recs[q].b < cb || (recs[q].b == cb && recs[q].n0 < cn))) {
                        hz_delta -= (long long)recs[q].hz;
                        if (lane == 0)
                            recs[q].hz = 0;
                        q++;
                    }
                    if (cout == LAP_OUT_WRAP && q < nl && !(recs[q].flags & LAPF_HEAD) && recs[q].b == cb && recs[q].n0 == cn) {
                        synced = true;
                        break;
                    }
                    /* one more lap (or what is left of the territory after a late end), from the truth, to its end wherever that is */
                    LapLane<KIND> v = lap_lane<KIND>(lane == 0, cx, cb, cn, cjc, false, 0, 0);
                    v.fresh = cout == LAP_OUT_WRAP;
                    v.one_lap = true;
                    lap_walk<KIND, true, false>(p, L, i, v);
                    n_rewalked++;
                    hz_delta += (long long)__builtin_amdgcn_readlane((int)v.hz, 0);
                    cx = bits_f64(readlane_u64(f64_bits(v.x), 0));
                    cb = __builtin_amdgcn_readlane(v.b, 0);
                    cn = __builtin_amdgcn_readlane(v.n, 0);
                    cout = __builtin_amdgcn_readlane(v.outcome, 0);
                    cjc = KIND == NCO_CODE ? (uint32_t)__builtin_amdgcn_readlane((int)v.jc, 0) : 0u;
                    chain_done = cout == LAP_OUT_CHAIN;
#ifdef GPSBB_LAP_DEBUG
                    if (lane == 0)
                        printf("   alone: now at (%d,%d) x %.17g outcome %d; next planned q %u (%d,%d)\n", cb, cn, cx, cout, q, q < nl ? recs[q].b : -1, q < nl ? recs[q].n0 : -1);
#endif
                }
                if (chain_done) {
                    /* whatever was planned up to the chain's end is void */
                    while (q < nl && !(recs[q].flags & LAPF_HEAD)) {
                        hz_delta -= (long long)recs[q].hz;
                        if (lane == 0)
                            recs[q].hz = 0;
                        q++;
                    }
                    from = q;
                    break;
                }
                (void)synced;
                /* 2. lap q starts at (cb, cn) from cx: is that what pass 2 walked it from? */
                {
                    const LapRec rq = recs[q];
                    const double xq = __fma_rn(rq.g, lap_unit<KIND>(), rq.A);
#ifdef GPSBB_LAP_DEBUG
                    if (lane == 0)
                        printf("   synced at lap %u (%d,%d): truth %.17g, pass 2 had %.17g (A %.17g g %.0f flags %x)\n", q, cb, cn, cx, xq, rq.A, rq.g, rq.flags);
#endif
                    if (q >= dirty_end && f64_bits(xq) == f64_bits(cx)) {
                        from = q;
                        break;
                    }
                }
                /* 3. the laps from q on, 64 at a time */
                double mq = (cx - recs[q].A) * lap_runit<KIND>();
                bool back_to_1 = false;
                for (;;) {
                    const uint32_t r = q + (uint32_t)lane;
                    /* the group ends with the chain (the next head) */
                    const bool in_range = r < nl;
                    const LapRec rr = recs[in_range ? r : q];
                    const unsigned long long heads = __ballot(in_range && (rr.flags & LAPF_HEAD));
                    const int glen0 = heads ? __builtin_ctzll(heads) : 64;
                    const int glen = (int)((nl - q) < (uint32_t)glen0 ? (nl - q) : (uint32_t)glen0);
                    /* (glen >= 1: lap q is not a head) */
                    const bool mine = lane < glen;
                    const LapRec rx = recs[(r + 1 < nl) ? r + 1 : r];
                    const bool has_next = mine && r + 1 < nl && !(rx.flags & LAPF_HEAD);
                    /* reference walks and their links */
                    LapLane<KIND> w1 = lap_lane<KIND>(mine, rr.A, rr.b, rr.n0, lap_jc_of<KIND>(p, L, i, r, rr.b), has_next, rx.b, rx.n0);
                    lap_walk<KIND, false, true>(p, L, i, w1);
                    const LapMap link = lap_link<KIND>(w1, mine, has_next, rx.A, rx.b, rx.n0);
                    const LapMap acc = lap_wave_scan(link, lane);
                    const LapMap before = lap_map_shfl_up(acc, 1);
                    const double m = lane == 0 ? mq : lap_apply(before, mq);
                    const double m_next = lap_apply(acc, mq);
                    /* the first lane starts from the truth itself (it may be off the reference's grid after a guess that failed) */
                    const double x0 = lane == 0 ? cx : __fma_rn(m, lap_unit<KIND>(), rr.A);
                    const double x_next = __fma_rn(m_next, lap_unit<KIND>(), rx.A);
                    LapLane<KIND> w2 = lap_lane<KIND>(mine, x0, rr.b, rr.n0, lap_jc_of<KIND>(p, L, i, r, rr.b), has_next, rx.b, rx.n0);
                    lap_walk<KIND, true, false>(p, L, i, w2);
                    bool ok = true;
                    if (mine) {
                        if (has_next)
                            ok = w2.outcome == LAP_OUT_WRAP && w2.b == rx.b && w2.n == rx.n0 && f64_bits(w2.x) == f64_bits(x_next);
                        else
                            ok = w2.outcome == LAP_OUT_CHAIN;
                    }
                    const unsigned long long badm = __ballot(mine && !ok);
                    const int nok = badm ? __builtin_ctzll(badm) + 1 : glen; /* lanes 0 .. nok-1 walked the truth */
#ifdef GPSBB_LAP_DEBUG
                    if (mine)
                        printf("   redo lap %u: start (%d,%d) x0 %.17g (A %.17g m %.0f) -> end (%d,%d) x %.17g outcome %d; next (%d,%d) x_next %.17g has_next %d ok %d glen %d\n",
                               r, rr.b, rr.n0, x0, rr.A, m, w2.b, w2.n, w2.x, w2.outcome, rx.b, rx.n0, x_next, (int)has_next, (int)ok, glen);
#endif
                    if (lane < nok) {
                        hz_delta += (long long)w2.hz - (long long)rr.hz;
                        recs[r].g = m;
                        recs[r].hz = w2.hz;
                        recs[r].flags = rr.flags & ~LAPF_BAD;
                    }
                    n_rewalked += (unsigned long long)nok;
                    /* (lanes beyond nok wrote into their own territories only: the laps are done again below) */
                    const int last = nok - 1;
                    cx = bits_f64(readlane_u64(f64_bits(w2.x), last));
                    cb = __builtin_amdgcn_readlane(w2.b, last);
                    cn = __builtin_amdgcn_readlane(w2.n, last);
                    cout = __builtin_amdgcn_readlane(w2.outcome, last);
                    cjc = KIND == NCO_CODE ? (uint32_t)__builtin_amdgcn_readlane((int)w2.jc, last) : 0u;
                    if (badm && q + (uint32_t)glen > dirty_end)
                        dirty_end = q + (uint32_t)glen;
                    q += (uint32_t)nok;
                    if (badm) {
                        n_links++;
                        chain_done = cout == LAP_OUT_CHAIN;
                        back_to_1 = true;
                        break;
                    }
                    if (cout == LAP_OUT_CHAIN || q >= nl || (recs[q].flags & LAPF_HEAD)) {
                        from = q; /* the chain is done */
                        break;
                    }
                    /* the next group's first lap starts at (cb, cn) from cx: as pass 2 had it? */
                    {
                        const LapRec rq = recs[q];
                        const double xq = __fma_rn(rq.g, lap_unit<KIND>(), rq.A);
                        if (q >= dirty_end && !(rq.flags & LAPF_BAD) && f64_bits(xq) == f64_bits(cx)) {
                            from = q;
                            break;
                        }
                        mq = (cx - rq.A) * lap_runit<KIND>();
                    }
                }
                if (!back_to_1)
                    break;
            }
        }
    }
    if (lane == 0) {
        const size_t klast = (size_t)(p.nblocks - 1) * p.nch + i;
        if (KIND == NCO_CARR && p.carry && !p.kph0 && (p.ch ? p.ch[klast].prn : p.cd[klast].prn) > 0)
        {
            /* (approx_end: what the row walks' chain — k_chain_prefix — of a later push starts from, should the stream change sides) */
            p.carry->exact_end[i] = p.lap_end ? p.lap_end[klast] : p.end[klast].carr_phase;
            p.carry->approx_end[i] = p.carry->exact_end[i];
        }
        if (hz_delta)
            atomicAdd(p.hazards + (KIND == NCO_CARR ? 0 : 1), (unsigned long long)hz_delta);
        if (n_rewalked)
            atomicAdd(p.hazards + 4, n_rewalked); /* GPSBB_INFO_CHAIN_FALLBACKS: laps walked again */
        if (n_links)
            atomicAdd(p.hazards + 6, n_links); /* GPSBB_INFO_CHAIN_REPAIRS: links that did not hold */
    }
}

/* ---- the kernels ----------------------------------------------------------------------------------------------------
 * One kind per launch (k_lap_*<KIND>: a stream's pushes, whose code chains need not wait for the push before while their
 * carriers do; the chain alone, gpsbb_chain_carrier), or BOTH kinds in one grid (k_lap_*2: batches that continue nothing —
 * the drop-in call's one block, resident batches — where the ten launches in a row were the pre-pass's latency: five, and the
 * two plan kernels' 16 wavefronts each run side by side). */
template <int KIND>
__global__ __launch_bounds__(64) void k_lap_plan(BatchDev p, LapDev L) { lap_plan_body<KIND>(p, L, blockIdx.x); }
template <int KIND>
__global__ __launch_bounds__(LAP_WG) GPSBB_LAP_OCC void k_lap_pass1(BatchDev p, LapDev L) { lap_pass1_body<KIND>(p, L, blockIdx.x); }
template <int KIND>
__global__ __launch_bounds__(64) void k_lap_scan(BatchDev p, LapDev L) { lap_scan_body<KIND>(p, L, blockIdx.x); }
/* (WIDE: the tile states go out in 32- / 16-byte pieces where a row holds them, a code period's data bits when it is over —
 * lap_emit_row, lap_nav_out: geometries with several tiles per lap, >= 8 MS/s; the tables are the same bits either way) */
template <int KIND, bool WIDE = false>
__global__ __launch_bounds__(LAP_WG) GPSBB_LAP_OCC void k_lap_pass2(BatchDev p, LapDev L) { lap_pass2_body<KIND, WIDE>(p, L, blockIdx.x); }
template <int KIND>
__global__ __launch_bounds__(64) void k_lap_repair(BatchDev p, LapDev L) { lap_repair_body<KIND>(p, L, blockIdx.x); }

/* both kinds: workgroups [0, nch) / [0, code chunks) are the code chains', the rest the carriers' */
__global__ __launch_bounds__(64) void k_lap_plan2(BatchDev p, LapDev L)
{
    if ((int)blockIdx.x < p.nch)
        lap_plan_body<NCO_CODE>(p, L, blockIdx.x);
    else
        lap_plan_body<NCO_CARR>(p, L, blockIdx.x - (uint32_t)p.nch);
}
__global__ __launch_bounds__(LAP_WG) GPSBB_LAP_OCC void k_lap_pass1_2(BatchDev p, LapDev L)
{
    const uint32_t cc = L.chunk0[NCO_CODE][p.nch] - L.chunk0[NCO_CODE][0];
    if (blockIdx.x < cc)
        lap_pass1_body<NCO_CODE>(p, L, blockIdx.x);
    else
        lap_pass1_body<NCO_CARR>(p, L, blockIdx.x - cc);
}
__global__ __launch_bounds__(64) void k_lap_scan2(BatchDev p, LapDev L)
{
    if ((int)blockIdx.x < p.nch)
        lap_scan_body<NCO_CODE>(p, L, blockIdx.x);
    else
        lap_scan_body<NCO_CARR>(p, L, blockIdx.x - (uint32_t)p.nch);
}
template <bool WIDE = false>
__global__ __launch_bounds__(LAP_WG) GPSBB_LAP_OCC void k_lap_pass2_2(BatchDev p, LapDev L)
{
    const uint32_t cc = L.chunk0[NCO_CODE][p.nch] - L.chunk0[NCO_CODE][0];
    if (blockIdx.x < cc)
        lap_pass2_body<NCO_CODE, WIDE>(p, L, blockIdx.x);
    else
        lap_pass2_body<NCO_CARR, WIDE>(p, L, blockIdx.x - cc);
}
__global__ __launch_bounds__(64) void k_lap_repair2(BatchDev p, LapDev L)
{
    if ((int)blockIdx.x < p.nch)
        lap_repair_body<NCO_CODE>(p, L, blockIdx.x);
    else
        lap_repair_body<NCO_CARR>(p, L, blockIdx.x - (uint32_t)p.nch);
}


} /* namespace gpsbb_impl */
#endif
