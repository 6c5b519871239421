/*
 * gpsbb_events.hip.h — k_synth_ev: the sample loop (plutogpssim.c:2690-2756) evaluated per BREAKPOINT
 * instead of per sample, for sample rates at which a run of 16 consecutive samples of one channel holds
 * only a few table-index changes (|f_carr|*delt*512*15.5 < 4) and at most one chip change
 * (f_code*delt*15.5 < 1: sample rates above ~15.9 MS/s).  Hand-written HIP for gfx950.
 *
 * What a channel contributes to a run of SPT samples is piecewise constant: sign(j) * amp(j) with
 * amp = the (cos, sin) amplitude pair of table index floor(512*carr_phase) and sign = codeCA*dataBit.
 * Instead of stepping both NCOs through every sample, a lane locates the few samples at which its run's
 * table index or chip changes and adds the CHANGE of the contribution there: a difference array D[j] per
 * lane (LDS, ds_add_u32, one 256-byte row per j: conflict-free), D[0] in a register; one prefix sum per
 * tile over j then gives the 16 int16 I/Q pairs of all channels at once.
 *
 * Where the breakpoints are.  The reference's NCOs are sequences of rounded IEEE additions, so the state
 * at sample n is not x0 + n*s.  The pre-pass walks every chain exactly (gpsbb_nco.h) and leaves the exact state
 * at the first sample of every 1024-sample tile.  Inside a tile a lane uses the linear model
 *     state(n) ~= state(tile start) + (n - tile start) * step
 * which differs from the true sequence by the roundings of at most 1039 additions (each at most half an
 * ulp of a number below 512 resp. 1024: 2^-45 resp. 2^-44): less than 2^-33.9 table-index units / chips (MEASURED over
 * every tile of 19 corner workloads against the reference's own recurrence, tools/model_err.py: 0.125 resp. 0.25 units of
 * 2^-32).  What a lane actually tests is that model in guard format (below), with the format's own roundings on top:
 * EV_MODEL_ERR bounds the total at the first sample of a run.  floor(model) equals floor(truth) at every sample unless
 * the model passes within that distance of an integer at a sample; a
 * lane tests exactly that for each of its breakpoints (and for its first sample) and, if it cannot rule
 * it out, recomputes its run exactly: jump-ahead from the tile's exact state with the genuine IEEE steps
 * (ev_exact_run).  At 25 MS/s that happens for about one lane-run-channel in 10^6, so the cost is nil,
 * and the output is bit-exact by construction, not by luck.  Wraps need no special case on the fast
 * path: the carrier's is index 511 -> 0 (the model runs on unwrapped, the index is taken modulo 512), the
 * code's is chip 1022 -> 0 with the next period's data bit.
 *
 * I and Q travel as one 32-bit integer P = Q*65536 + I: sums of such integers are exact modulo 2^32, the
 * low half is the I sum modulo 2^16 (what the reference's (short) cast keeps) and the high half is the Q
 * sum plus the borrow of the low half.  The sum starts at 2^15, so that as long as |I sum| < 2^15 the low half
 * is I + 2^15 >= 0 and never borrows: one v_xor with 0x8000 per sample gives the int16 pair.  The
 * host only selects this kernel when the gains guarantee that (sum of |gain| < 63); other batches, low
 * sample rates and the fixed-point carrier run on k_synth.
 */
#ifndef GPSBB_EVENTS_HIP_H
#define GPSBB_EVENTS_HIP_H

#include <cstddef>
#include <type_traits>

#include "gpsbb_kernels.hip.h"

namespace gpsbb_impl {

#ifndef GPSBB_EV_WG
#define GPSBB_EV_WG 1024
#endif
constexpr int EV_WG = GPSBB_EV_WG;
constexpr int EV_WAVES = EV_WG / 64;
constexpr int EV_KC_MAX = 4;                       /* carrier breakpoints a run may hold */
constexpr int EV_AMP_PAD = EV_KC_MAX;              /* table entries repeated after [511] */
constexpr int EV_AMP_STRIDE = 512 + EV_AMP_PAD + 4;
constexpr int EV_CHIP_LEN = 1024 + 544;            /* chips 0 .. 1567: a tile's model phase never passes 1023 + 1040*sc; the host admits
                                                      sc up to (EV_CHIP_LEN - 1026) / 1040 = 0.52 chips per sample (1.96 MS/s) */
constexpr int EV_KC_DENSE = EV_KC_MAX + 1;         /* EvConst::kc of a channel that is evaluated sample by sample (ev_dense) */
#ifndef GPSBB_EV_CHUNK
#define GPSBB_EV_CHUNK 4
#endif
constexpr int EV_CHUNK = GPSBB_EV_CHUNK;           /* consecutive tiles a wavefront of the breakpoint kernels takes at a time (default of
                                                      BatchDev::ev_chunk; measured beside the lap-parallel pre-pass, round 5: 2 -> 4 tiles
                                                      + 1.6 % for k_synth_ev, alone and in the stream; 6 and 8 lose it again) */
constexpr int PD_CHUNK = 2;                        /* ... of k_synth_pd (3 and 4 tiles: - 1.3 %, - 2 %) */
constexpr int EV_ROW_DISCARD = 15;                 /* D row of changes that fall behind the run's last sample */

/* Bound on |guard-format model - truth| at the first sample of a run, in table-index units / chips: the linear model's own
 * error (< 2^-33.9 = 0.27 units of 2^-32; measured 0.25) PLUS the roundings of the format on the way there — the tile state
 * put into guard format (half a unit) and the fma that takes it to the run's first sample (half a unit): 1.27 units, measured
 * 1.1 (tools/model_err.py, profiles/r04_model_err.json).  Rounds 2 and 3 had 2^-32 here — the model's error alone — and let
 * EV_T_EPS absorb the format's roundings; but the position of a change is (integer - first-sample model) / step, so EVERYTHING
 * in the first-sample model is amplified by 1 / step: measured, the change positions were off by up to 1.48 W (index) and
 * 1.41 W (chip) with the old constants — outside the band the danger test covers, i.e. a truth landing within 0.4 units above
 * an integer at a sample could be placed one sample late without being flagged (about 1e-13 per channel-sample; no soak had
 * hit it).  Now 4 units, the channel's bias W is a whole number of units (so that the bias the format carries is W exactly and
 * not W rounded), and the test suite asserts realised <= W / 2 for everything tested. */
#ifndef EV_MODEL_ERR
#define EV_MODEL_ERR 0x1p-30
#endif
/* The quantities a lane tests travel in GUARD FORMAT: 2^20 + value, so that one unit in the last place is 2^-32, the
 * double's low word IS the fraction (in units of 2^-32) and the low bits of its high word ARE the integer part — index,
 * chip and row come out with one v_and, and "within the error of an integer" is a comparison of low words.  Every
 * tested quantity also carries the channel's bias +W (EvConst::W >= its error bound): floor(value + W) = floor(truth)
 * unless the low word of the biased value is below 2W (EvConst::danger), and the minimum of the low words of everything
 * a lane tests for a channel is compared once.  EV_T_EPS: the roundings of a change position's own arithmetic (the constant
 * tK0 / tC0: one unit; the fma: half a unit; up to three additions of 1 / step: half a unit each — 3 units at most) on top of
 * EV_MODEL_ERR / step. */
#ifndef EV_T_EPS
#define EV_T_EPS 0x1p-29
#endif
#define EV_GUARD 0x1p+20

/* A wavefront claims its next chunk of tiles with a returning atomic.  Rounds 2 and 3 issued it in inline assembly and did not
 * wait for it until the chunk's last tile — invisible to the compiler, which is free to copy or spill a register it believes
 * to be idle: round 3 lost chunks that way once (patched at the call sites with this macro) and round 4 again, as a hang, when
 * one more variable changed the allocation of k_synth_pd.  The claim is an ordinary atomic now: the compiler waits for it where
 * it issues it (it aggregates the wavefront's lanes and needs the value at once), which costs 1.2 % of the kernel
 * (tools/bound_hunt.sh CLAIM_ASM) and depends on nobody's register allocator.  The old way, for measuring: tools/experiments/bound_hunt_variants.patch. */
#define GPSBB_EV_SETTLE_CLAIM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

/* LDS image of one workgroup.  Two shapes: the mixed kernel (k_synth_ev_dense) needs the long chip table of the channels it
 * evaluates per sample (code steps up to 0.52 chips per sample) and takes its amplitude index modulo 512; the breakpoint
 * kernels proper (at most one chip change per run: steps below 0.0646, chip index below 1093) spend what the short chip table
 * saves on an amplitude table that simply goes on past 511 — 512 + 1040 * 4 / 15.5 entries cover a tile's unwrapped model —
 * so that the table index needs no masking instruction (see ev_first). */
constexpr int EV_CHIP_LEN_SHORT = 1104;
constexpr int EV_AMP_STRIDE_LONG = 512 + 272 + EV_AMP_PAD; /* 788 */
template <int AMP_STRIDE, int CHIP_LEN>
struct EvLdsT {
    static constexpr int AMP = AMP_STRIDE, CHIPS = CHIP_LEN;
    static constexpr bool UNMASKED = AMP_STRIDE >= EV_AMP_STRIDE_LONG; /* the amplitude index is not reduced modulo 512 */
    uint32_t amp[GPSBB_MAX_CHAN][AMP_STRIDE];    /* P = Q*65536 + I of table index k mod 512 at [k];
                                                    channels with a falling carrier: of index 511 - k */
    uint32_t D[EV_WAVES][16][64];                /* difference arrays: row j-1 holds the change at sample j of the lane's
                                                    run (row 15 = discard), one column per lane */
    double tstate[EV_WAVES][2][2 * GPSBB_MAX_CHAN]; /* per wavefront.  [0]: the tile's exact states in guard format with the
                                                    channel's bias (2^20 + W + state), column 2*channel = code phase,
                                                    2*channel+1 = carrier phase*512 (512 - that for a falling carrier): written
                                                    at the top of every tile from registers — one buffer is enough, a
                                                    wavefront's LDS operations execute in order.  [1]: EvConst::tK0, tC0 of
                                                    the block's channels, the addends of the two position fmas, 256 bytes
                                                    behind the states of the same channel.  A VALU instruction of gfx9 reads
                                                    ONE scalar register pair, and fma(-fraction, 1 / step, constant) has two
                                                    scalar constants: out of scalar registers the compiler had to copy one
                                                    into a vector pair first (a v_mov_b64 per fma); out of LDS — a second
                                                    broadcast ds_read_b128 off the same address register — it arrives where
                                                    the fma wants it */
    uint16_t chip2[GPSBB_MAX_CHAN][CHIP_LEN];    /* low byte: 0 where codeCA of chip c mod 1023 is +1, 0xff where -1;
                                                    high byte: the same for chip c+1 */
};
typedef EvLdsT<EV_AMP_STRIDE, EV_CHIP_LEN> EvLds;                   /* k_synth_ev_dense */
typedef EvLdsT<EV_AMP_STRIDE_LONG, EV_CHIP_LEN_SHORT> EvLdsLean;    /* k_synth_ev, k_synth_ev_fixed */
static_assert(sizeof(EvLds) <= 160 * 1024 && sizeof(EvLdsLean) <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
static_assert(offsetof(EvLdsLean, D) < 65536 && offsetof(EvLds, D) < 65536, "the difference arrays are addressed with a 16-bit offset");

__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
template <class T>
__device__ __forceinline__ T lds_read_at(uint32_t a)
{
    return *(__attribute__((address_space(3))) const T *)(uintptr_t)a;
}

/* (The measurement variants of this kernel — deliberately WRONG builds that take one resource out of the picture each, to see
 * what its time is made of: DESIGN.md 3.1 — are not in this file: tools/experiments/bound_hunt_variants.patch puts them into a
 * scratch copy, tools/bound_hunt.sh builds and times them.) */

typedef uint32_t ev_u32x4 __attribute__((ext_vector_type(4))); /* 16 bytes: one ds_read_b128 / ds_write_b128 / global_store_dwordx4 */

template <class T>
__device__ __forceinline__ void lds_write_at(uint32_t a, T v)
{
    *(__attribute__((address_space(3))) T *)(uintptr_t)a = v;
}

constexpr uint32_t EV_SAT_HI_C = 0x4130000fu; /* the high word of a position clamped to 15.5 (row 15: discard) */
#define EV_SAT_HI EV_SAT_HI_C

/*
 * One difference into this lane's column of the wavefront's arrays, at the row a position's high word names.  The arrays'
 * byte offset inside the 64 KB they occupy is  wavefront * 4096 + row * 256 + lane * 4: byte 0 is the lane, byte 1 is
 * (wavefront << 4) + row.  `drow` holds that offset with bytes 0, 2, 3 constant; one SDWA add puts
 * (wavefront << 4) + (low byte of the position's high word) into byte 1 — an address in ONE instruction, where the
 * compiler's own code masks the row out (v_and) and shifts it in (v_lshl_add).  The row is below 16: positions are
 * clamped to 15.5.  The arrays themselves sit at a compile-time offset of the image (the instruction's offset field).
 */
template <class LDS>
__device__ __forceinline__ void ev_d_add(uint32_t &drow, int wave, uint32_t pos_hi, uint32_t v)
{
    /* volatile, no memory clobber: the differences keep their order among themselves and stay before the fence the
     * tile's epilogue starts with, but the compiler may still issue the next channel's table reads ahead of them */
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
                 "ds_add_u32 %0, %3 offset:%4"
                 : "+v"(drow)
                 : "s"(wave << 4), "v"(pos_hi), "v"(v), "n"(offsetof(LDS, D))
                 );
}

/* (x ^ m) - m: x where m = 0, -x where m = -1 */
__device__ __forceinline__ uint32_t signed_by(uint32_t x, uint32_t m) { return (x ^ m) - m; }

/* wave-uniform loads through the scalar cache: the tables the host wrote before this kernel started are
 * read-only here, so they may be read as constant-address-space data (s_load: the values arrive in scalar
 * registers and cost no vector-memory or LDS slot) */
template <class T>
__device__ __forceinline__ T scalar_load(const T *p)
{
    typedef const __attribute__((address_space(4))) T *cptr_t;
    return *(cptr_t)(uintptr_t)p;
}

/*
 * The exact recomputation of one lane's run of one channel (rare): advance both NCOs from the tile's exact
 * state by n_off genuine steps with the jump-ahead of gpsbb_nco.h, then walk the run sample by sample as the
 * reference does (c:2697-2746) and add the differences of its contributions.  Returns the contribution at
 * the run's first sample.
 */
template <bool FIXED, class LDS>
__device__ __forceinline__ uint32_t ev_exact_run_body(LDS &L, int wave, int lane, int i, const EvConst *kbi, const double *tile_x,
                                                      int ntiles, uint32_t nb, int n_off, uint32_t fx_phase, int32_t fx_step)
{
    constexpr bool fixed = FIXED;
    const bool down = kbi->down != 0;
    const double S = down ? -kbi->S : kbi->S, sc = kbi->sc;
    const double xt = tile_x[(size_t)(2 * i) * ntiles], yt = tile_x[(size_t)(2 * i + 1) * ntiles];
    /* code NCO: at most one roll-over between the tile start and the end of the run (checked by the host) */
    int64_t wraps = 0;
    double x = code_jump(xt, sc, (int64_t)n_off, &wraps);
    uint32_t dbm = (wraps > 0 ? (nb >> 1) & 1u : nb & 1u) ? 0xffffffffu : 0u;
    const uint32_t dbm_next = ((nb >> 1) & 1u) ? 0xffffffffu : 0u;
    /* carrier NCO in cycles (the tile state is stored scaled by 512, exactly) */
    const double s = S * (1.0 / 512.0);
    double cp = carr_jump(yt * (1.0 / 512.0), s, (int64_t)n_off);
    uint32_t prev = 0, first = 0;
    uint32_t ph = fx_phase + (uint32_t)n_off * (uint32_t)fx_step; /* fixed-point carrier: the accumulator at the run's first sample */
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        const int it = fixed ? (int)((ph >> 16) & 0x1ffu) /* c:2699 */
                             : ((int)(cp * 512.0) & 511); /* c:2697; carr_phase == 1.0: index 512 defined as 0 */
        const int ci = (int)x;                  /* c:2737 */
        const uint32_t m = (uint32_t)(int32_t)(int8_t)(L.chip2[i][ci] & 0xffu) ^ dbm;
        const uint32_t v = signed_by(L.amp[i][down ? 511 - it : it], m);
        if (j == 0)
            first = v;
        else
            __hip_atomic_fetch_add(&L.D[wave][j - 1][lane], v - prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = v;
        if (code_step(x, sc)) /* c:2709-2734: the data bit of the next period */
            dbm = dbm_next;
        carr_step(cp, s); /* c:2741-2746 */
        ph += (uint32_t)fx_step; /* c:2748 */
    }
    return first;
}

/* fixed-point carrier: the block's start phases and steps per channel and the tile's first sample (ev_exact_run's way to the
 * accumulator; only k_synth_ev_fixed fills it in) */
struct EvFixed {
    const uint32_t *ph;
    const int32_t *st;
    int n0;
};

template <class LDS>
__device__ __noinline__ uint32_t ev_exact_run(LDS &L, int wave, int lane, int i, const EvConst *kbi, const double *tile_x,
                                              int ntiles, uint32_t nb, int n_off)
{
    return ev_exact_run_body<false>(L, wave, lane, i, kbi, tile_x, ntiles, nb, n_off, 0u, 0);
}
/* ... with the fixed-point carrier: the accumulator at the tile's first sample and its step take the carrier NCO's place */
template <class LDS>
__device__ __noinline__ uint32_t ev_exact_run_fixed(LDS &L, int wave, int lane, int i, const EvConst *kbi, const double *tile_x,
                                                    int ntiles, uint32_t nb, int n_off, uint32_t fx_phase, int32_t fx_step)
{
    return ev_exact_run_body<true>(L, wave, lane, i, kbi, tile_x, ntiles, nb, n_off, fx_phase, fx_step);
}

/* per-channel constants of the fast path (scalar registers) */
struct EvK {
    double S, rS, sc, rsc;
    uint32_t danger;    /* EvConst::danger_le: a lane whose smallest low word is <= this goes to the exact path */
    uint32_t chip_base; /* LDS address of the channel's chip table, minus what the exponent bits of a guard-format high word
                           contribute when it is shifted into a byte offset (see ev_first) */
    uint32_t amp_base;  /* ... of its amplitude table, likewise (EvLdsLean) */
};
/* the high word of a double in [2^20, 2^21) is 0x41300000 + its integer part */
constexpr uint32_t EV_GUARD_HI = 0x41300000u;
template <class LDS>
__device__ __forceinline__ EvK ev_load_k(const LDS &L, const EvConst *kb, int i)
{
    EvK k;
    /* the record's address as base + (i << 7): one shift, which the two loads take as their scalar offset; the fields in
     * the order of the record (gpsbb_kernels.hip.h), 32 + 16 bytes */
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const char *ki = reinterpret_cast<const char *>(kb) + ((uint32_t)i << 7);
    const u32x8 a = scalar_load(reinterpret_cast<const u32x8 *>(ki));
    const u32x4 c = scalar_load(reinterpret_cast<const u32x4 *>(ki + offsetof(EvConst, danger_le)));
    k.S = __hiloint2double((int)a[1], (int)a[0]);
    k.rS = __hiloint2double((int)a[3], (int)a[2]);
    k.sc = __hiloint2double((int)a[5], (int)a[4]);
    k.rsc = __hiloint2double((int)a[7], (int)a[6]);
    k.danger = c[0];
    if (LDS::UNMASKED) { /* (EvLdsLean: the host put the two table addresses into the record, next to the threshold) */
        k.chip_base = c[1];
        k.amp_base = c[2];
    } else {
        k.chip_base = lds_addr_of(&L.chip2[i][0]) - (EV_GUARD_HI << 1);
        k.amp_base = lds_addr_of(&L.amp[i][0]) - (EV_GUARD_HI << 2);
    }
    return k;
}

/*
 * First half of one channel's work on this lane's run: where the table index and the chip change, and the
 * LDS reads of the values involved (issued, not waited for).  KC: carrier breakpoints a run can hold
 * (wave-uniform, from the host).  A channel whose carrier falls is walked mirrored (phase 512 - y, step |S|,
 * amplitude table stored back to front: floor(y) = 511 - floor(512 - y) wherever y is not an integer, and a
 * lane that comes within the model error of one recomputes exactly anyway), so this code only knows rising
 * phases.  The code phase is not reduced modulo 1023: the chip table simply continues past 1023.
 */
template <int KC>
struct EvHalf {
    uint32_t hk[KC]; /* the k-th index change as the HIGH WORD of its position in guard format (clamped to 15.5): the low byte
                        is the row of D (= sample - 1), 15 = not in this run; high words of one binade compare like the positions */
    uint32_t hc;     /* ... of the chip change */
    int c0;     /* chip of the first sample, not reduced modulo 1023 */
    unsigned long long um; /* lanes that cannot rule out a disagreement between the model and the reference (wave mask:
                              the comparisons land in scalar registers and are combined there) */
    uint32_t A[KC + 1];
    uint32_t ch2;
};

/* FIXED: the fixed-point carrier (GPSBB_FIXED_CARRIER).  Its table index (phase mod 2^25) / 2^16 and its step are multiples of
 * 2^-16 and every product and sum here is exact, so the model IS the truth on the carrier side: no bias, nothing to test.
 * The k-th index change lies (1 - fraction + k) / step samples on; with one half of 2^-16 taken off the numerator
 * (EvConst::tK0) that is (2a - 1) / (2b) for integers a, b = |step| < 2^15 — never an integer and at least 1 / (2b) away
 * from one, against the 2^-32 this format resolves — so floor() is right even where a change falls exactly on a sample. */
template <int KC, bool FIXED, class LDS>
__device__ __forceinline__ EvHalf<KC> ev_first(const LDS &L, int i, const EvK &K, double xt, double yt, double tK0, double tC0, double off)
{
    EvHalf<KC> h;
    const double sat = 15.5 + EV_GUARD; /* a change past the run's last sample: row 15, fraction one half */
    /* ---- carrier: table index of the first sample and the samples at which it changes ---- */
    const double y0 = __fma_rn(off, K.S, yt); /* guard format, biased: 2^20 + W + phase */
    const double fr = __builtin_amdgcn_fract(y0);
    const int it0 = __double2hiint(y0) & 511;
    uint32_t m = FIXED ? 0xffffffffu : (uint32_t)__double2loint(y0); /* the previous change lies within the model error of sample 0: low word below 2W */
    double t = __fma_rn(-fr, K.rS, tK0);       /* 2^20 + W + (1 - fraction) / |step|: samples until the next index change */
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const double tq = fmin(t, sat);
        if (!FIXED)
            m = min(m, (uint32_t)__double2loint(tq));
        h.hk[k] = (uint32_t)__double2hiint(tq); /* the change shows at sample floor + 1: rows 0..14, or 15 */
        t += K.rS;
    }
    if (LDS::UNMASKED) {
        /* the table goes on past 511: the entry's address straight from the high word, like the chip pair's below */
        uint32_t amp_at;
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(amp_at) : "v"(__double2hiint(y0)), "s"(K.amp_base));
#pragma unroll
        for (int k = 0; k <= KC; k++)
            h.A[k] = lds_read_at<uint32_t>(amp_at + 4u * (uint32_t)k);
    } else {
        const uint32_t *ampi = &L.amp[i][it0];
#pragma unroll
        for (int k = 0; k <= KC; k++)
            h.A[k] = ampi[k];
    }
    /* ---- code: chip of the first sample and the sample at which it changes ---- */
    const double x0 = __fma_rn(off, K.sc, xt);
    const double frc = __builtin_amdgcn_fract(x0);
    h.c0 = __double2hiint(x0) & 2047;          /* (only the tiles in which the data bit changes look at it) */
    const double tc = fmin(__fma_rn(-frc, K.rsc, tC0), sat);
    m = min(m, min((uint32_t)__double2loint(x0), (uint32_t)__double2loint(tc)));
    h.hc = (uint32_t)__double2hiint(tc);
    /* the chip pair's address straight from the high word: shifted left by one its integer part is the byte offset and its
     * exponent bits a constant that the channel's base already has taken off — no masking instruction */
    uint32_t chip_at;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(chip_at) : "v"(__double2hiint(x0)), "s"(K.chip_base));
    h.ch2 = lds_read_at<uint16_t>(chip_at);
    /* lanes that cannot rule out a disagreement between the model and the reference: one comparison for everything tested */
    h.um = __builtin_amdgcn_uicmp(m, K.danger, 37 /* ule */);
    return h;
}

/*
 * Second half: signs, the contribution at sample 0 (into acc0) and its changes (into D).  db / db_next: the
 * data bit in force at the tile start / after the next code roll-over as masks (0 = +1, -1 = -1; wave-uniform);
 * DF: they differ, so lanes past the roll-over (chip index >= 1023) take the other one.
 */
template <int KC, bool DF, bool FIXED, class LDS>
__device__ __forceinline__ void ev_second(LDS &L, uint32_t &drow, int wave, int lane, int i, EvHalf<KC> &h, uint32_t db, uint32_t db_next,
                                          unsigned long long live_mask, const EvConst *kb, const double *tile_x,
                                          int ntiles, uint32_t nb, double off, uint32_t &acc0, unsigned long long *n_exact,
                                          const EvFixed &fx)
{
    const uint32_t ma = (uint32_t)(int32_t)(int8_t)(h.ch2 & 0xffu), mb = (uint32_t)(int32_t)(int8_t)(h.ch2 >> 8);
    uint32_t m0, m1;
    if (DF) {
        m0 = ma ^ (h.c0 >= 1023 ? db_next : db);
        m1 = mb ^ (h.c0 >= 1022 ? db_next : db);
    } else {
        m0 = ma ^ db;
        m1 = mb ^ db;
    }
    uint32_t hc = m0 == m1 ? EV_SAT_HI : h.hc; /* equal neighbours: nothing changes at the chip boundary */

    /* ---- rare: this lane cannot rule out that the model and the reference disagree (a channel that is always recomputed
     * exactly has a threshold every low word is below: EvConst::danger_le).  The branch is on the comparison as it comes
     * out of the vector unit; that the lane holds samples of the block at all is only looked at behind it ---- */
    if (__builtin_expect(h.um != 0ull, 0)) {
        const unsigned long long um = h.um & live_mask;
        GPSBB_EV_SETTLE_CLAIM();
        if ((um >> lane) & 1ull) {
            /* its fast-path contribution becomes nothing ... */
#pragma unroll
            for (int k = 0; k < KC; k++)
                h.hk[k] = EV_SAT_HI;
            hc = EV_SAT_HI;
            h.A[0] = 0;
            /* ... and the exact one takes its place */
            if (FIXED) /* the accumulator at the tile's first sample */
                acc0 += ev_exact_run_fixed(L, wave, lane, i, kb + i, tile_x, ntiles, nb, (int)off,
                                           fx.ph[i] + (uint32_t)fx.n0 * (uint32_t)fx.st[i], fx.st[i]);
            else
                acc0 += ev_exact_run(L, wave, lane, i, kb + i, tile_x, ntiles, nb, (int)off);
            atomicAdd(n_exact, 1ull);
        }
    }

    /* ---- the contribution at sample 0 and its changes ---- */
    {
        /* += signed_by(A[0], m0) as (A ^ m) + (acc - m): two instructions (the compiler's own choice is xor, sub, add) */
        const uint32_t c = acc0 - m0;
        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(acc0) : "v"(h.A[0]), "v"(m0), "v"(c));
    }
    uint32_t Ax = h.A[KC];
#pragma unroll
    for (int k = KC - 1; k >= 0; k--) {
        const bool before = h.hk[k] < hc; /* the index change comes before the chip change */
        const uint32_t mk = before ? m0 : m1;
        const uint32_t dk = signed_by(h.A[k + 1] - h.A[k], mk);
        ev_d_add<LDS>(drow, wave, h.hk[k], dk);
        Ax = before ? Ax : h.A[k]; /* amplitude in force just before the chip change */
    }
    /* the chip change flips the sign: -s0*A -> s1*A = 2*s1*A more */
    ev_d_add<LDS>(drow, wave, hc, signed_by(Ax << 1, m1));
}

/* what a wavefront knows about the tile it is working on */
struct EvTile {
    const double *ts;     /* LDS: the tile's states (mirrored where the carrier falls) */
    const double *tile_x; /* global: the same, exact and not mirrored (for the exact recomputation): chain c's is
                             tile_x[c * ntiles] */
    int ntiles;
    uint32_t dbits, dnext; /* bit i: channel i's data bit in force / after the next roll-over is -1 */
};

/*
 * A channel whose run holds more index or chip changes than the breakpoint path takes (low sample rates: at 2.6 MS/s
 * the table index changes at almost every sample): the same linear in-tile model, evaluated at every sample of the
 * lane's run — index and chip are floor(model), trusted wherever the model stays EV_MODEL_ERR away from an
 * integer, else the lane's run is recomputed exactly as on the other path — and accumulated per sample in registers
 * (accd), which the prefix sum of the difference arrays is added to at the end.  No rows, no NCO stepping.
 */
template <bool DF, class LDS>
__device__ __forceinline__ void ev_dense(LDS &L, int wave, int lane, int i, const EvConst *kb, const EvTile &T, double off,
                                         unsigned long long live_mask, uint32_t &acc0, uint32_t (&accd)[SPT],
                                         unsigned long long *n_exact)
{
    const double S = scalar_load(&kb[i].S), sc = scalar_load(&kb[i].sc);
    const uint32_t danger = scalar_load(&kb[i].danger);
    /* Both models in guard format with the channel's bias (see EV_T_EPS): index, chip and the test "within the error of an
     * integer" are integer operations on the two words. */
    const double y0 = __fma_rn(off, S, T.ts[2 * i + 1]), x0 = __fma_rn(off, sc, T.ts[2 * i]);
    const uint32_t db = 0u - ((T.dbits >> i) & 1u), dn = 0u - ((T.dnext >> i) & 1u);
    unsigned long long um = 0ull;
    uint32_t v[SPT];
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        const double yj = __fma_rn((double)j, S, y0), xj = __fma_rn((double)j, sc, x0);
        const uint32_t ylo = (uint32_t)__double2loint(yj), xlo = (uint32_t)__double2loint(xj);
        um |= __builtin_amdgcn_uicmp(min(ylo, xlo), danger, 36 /* ult */);
        const int it = __double2hiint(yj) & 511, ci = __double2hiint(xj) & 2047;
        uint32_t m = (uint32_t)(int32_t)(int8_t)(L.chip2[i][ci] & 0xffu);
        m ^= DF ? (ci >= 1023 ? dn : db) : db;
        v[j] = signed_by(L.amp[i][it], m);
    }
    um &= live_mask;
    if (__builtin_expect(um != 0ull, 0)) {
        GPSBB_EV_SETTLE_CLAIM();
        if ((um >> lane) & 1ull) {
#pragma unroll
            for (int j = 0; j < SPT; j++)
                v[j] = 0;
            const uint32_t nb = ((T.dbits >> i) & 1u) | (((T.dnext >> i) & 1u) << 1);
            acc0 += ev_exact_run(L, wave, lane, i, kb + i, T.tile_x, T.ntiles, nb, (int)off);
            atomicAdd(n_exact, 1ull);
        }
    }
#pragma unroll
    for (int j = 0; j < SPT; j++)
        accd[j] += v[j];
}

/* the channels of `mask` (bit i = channel i), all with KC breakpoints; two at a time, so that one channel's
 * arithmetic covers the other's LDS latency */
template <int KC, bool DF, bool FIXED, class LDS>
__device__ __forceinline__ void ev_channels(LDS &L, uint32_t &drow, int wave, int lane, uint32_t mask, const EvConst *kb, const EvTile &T,
                                            double off, unsigned long long live_mask, uint32_t &acc0, unsigned long long *n_exact,
                                            const EvFixed &fx)
{
#define GPSBB_EV_STATES(i)                                                                                             \
    const double xt##i = T.ts[2 * i], yt##i = T.ts[2 * i + 1];                                                         \
    const double tc##i = T.ts[2 * GPSBB_MAX_CHAN + 2 * i], tk##i = T.ts[2 * GPSBB_MAX_CHAN + 2 * i + 1];
#define GPSBB_EV_KIDX(i) i
#define GPSBB_EV_IN(i)                                                                                                 \
    const EvK K##i = ev_load_k(L, kb, GPSBB_EV_KIDX(i));                                                               \
    GPSBB_EV_STATES(i)
#define GPSBB_EV_OUT(i, h)                                                                                             \
    {                                                                                                                  \
        const uint32_t db_ = 0u - ((T.dbits >> i) & 1u), dn_ = 0u - ((T.dnext >> i) & 1u);                             \
        const uint32_t nb_ = ((T.dbits >> i) & 1u) | (((T.dnext >> i) & 1u) << 1);                                     \
        ev_second<KC, DF, FIXED>(L, drow, wave, lane, i, h, db_, dn_, live_mask, kb, T.tile_x,                 \
                          T.ntiles, nb_,                                                                               \
                          off, acc0, n_exact, fx);                                                                     \
    }
    while (mask & (mask - 1)) { /* at least two channels left */
        const int i0 = __builtin_ctz(mask);
        mask &= mask - 1;
        const int i1 = __builtin_ctz(mask);
        mask &= mask - 1;
        GPSBB_EV_IN(i0)
        GPSBB_EV_IN(i1)
        EvHalf<KC> h0 = ev_first<KC, FIXED>(L, i0, Ki0, xti0, yti0, tki0, tci0, off);
        EvHalf<KC> h1 = ev_first<KC, FIXED>(L, i1, Ki1, xti1, yti1, tki1, tci1, off);
        GPSBB_EV_OUT(i0, h0)
        GPSBB_EV_OUT(i1, h1)
    }
    if (mask) {
        const int i0 = __builtin_ctz(mask);
        GPSBB_EV_IN(i0)
        EvHalf<KC> h0 = ev_first<KC, FIXED>(L, i0, Ki0, xti0, yti0, tki0, tci0, off);
        GPSBB_EV_OUT(i0, h0)
    }
#undef GPSBB_EV_IN
#undef GPSBB_EV_OUT
#undef GPSBB_EV_STATES
}

/*
 * Which block a workgroup of the model kernels works on (k_synth_ev*, k_synth_pd).  The grid is one-dimensional:
 * workgroups 0 .. nblocks-1 are the blocks' PRIMARIES (dispatched first, in order, one per CU as CUs come free); the
 * EV_HELPERS(...) workgroups behind them are HELPERS, dispatched once every primary has been placed, i.e. when the launch is
 * running out of whole blocks.  A helper takes a ticket and joins the (ticket mod L)-th of the L blocks that still have at
 * least a chunk of tiles per wavefront left to hand out, counted from the LAST block — the blocks placed last have most left —
 * so that the helpers that arrive together spread evenly over what is still running, and the ones that arrive when a round
 * of blocks has finished go where the work is.  No live block: the helper leaves.
 *
 * Until round 6 the grid was (blocks, workgroups per block) with a block's helpers fixed in advance.  The workgroup trace of
 * round 6 (tools/corun_diag.py, profiles/r06_corun_diag.txt) showed what that cost: helpers of blocks long finished had to
 * be placed on a whole CU each and leave again before the first useful one was reached (1 941 of 3 000 workgroups of the
 * reference geometry's launch, 624 of 1 200 of the headline's), the last blocks of a launch got their helpers late or not at
 * all (the 32 blocks left at the end of the headline's launch had three workgroups each while 160 CUs idled), and the CUs
 * drained over the last 9 % (k_synth_ev) and 18 % (k_synth_pd) of a launch.
 *
 * slot: LDS byte address of a word outside the kernel's LDS image (the launch allocates EV_PICK_LDS bytes behind it).
 */
constexpr int EV_PICK_LDS = 16;
static_assert(sizeof(EvLds) + EV_PICK_LDS <= 160 * 1024 && sizeof(EvLdsLean) + EV_PICK_LDS <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
constexpr int EV_HELP_SCAN = 64 * 64; /* a helper looks at the last 4 096 blocks of the launch (one ballot per 64) */
__device__ __forceinline__ int ev_pick_block(const BatchDev &p, uint32_t slot)
{
    const int wg = (int)blockIdx.x;
    if (wg < p.nblocks)
        return wg;
    int b = -1;
    if (threadIdx.x < 64) { /* wavefront 0 decides for the workgroup */
        const int lane = (int)threadIdx.x;
        int ticket = 0;
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(p.tile_ctr + p.nblocks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        /* live: what is left to hand out is at least a chunk per wavefront (what a workgroup's staging is worth) */
        const int live_upto = p.ntiles - EV_WAVES * p.ev_chunk;
        const int nscan = p.nblocks < EV_HELP_SCAN ? p.nblocks : EV_HELP_SCAN;
        unsigned long long mymask = 0ull; /* lane c: which of the blocks at positions 64 c .. 64 c + 63 from the end are live */
        int nlive = 0;
        for (int c = 0; c * 64 < nscan; c++) {
            const int pos = c * 64 + lane;
            const bool live = pos < nscan &&
                              __hip_atomic_load(p.tile_ctr + (p.nblocks - 1 - (pos < nscan ? pos : 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= live_upto;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
            if (lane == c)
                mymask = m;
            nlive += __popcll(m);
        }
        if (nlive > 0) {
            const int r = ticket % nlive;
            const int cnt = __popcll(mymask);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d)
                    incl += v;
            }
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(r >= incl - cnt && r < incl); /* exactly one lane */
            const int c = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hm) - 1);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)mymask, c), hi = (uint32_t)__shfl((int)(uint32_t)(mymask >> 32), c);
            unsigned long long cm = ((unsigned long long)hi << 32) | lo;
            int k = __builtin_amdgcn_readfirstlane(r - __shfl(incl - cnt, c));
            for (; k > 0; k--)
                cm &= cm - 1ull; /* drop the k lowest set bits */
            b = p.nblocks - 1 - (c * 64 + (int)__ffsll((long long)cm) - 1);
        }
        if (lane == 0)
            lds_write_at<int>(slot, b);
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(lds_read_at<int>(slot));
}

/* DENSE: the batch has channels that are evaluated per sample (ev_dense): a kernel of its own (k_synth_ev_dense), so
 * that the common one keeps its register count: at <= 104 VGPRs four of its wavefronts leave room on a SIMD for a
 * wavefront of the pre-pass of the next push; at 120 they do not, and a CU that hosts walk wavefronts cannot take a
 * synthesis workgroup at all (measured: 2.2 -> 2.8 ms per launch beside the pre-passes). */
template <bool DENSE, bool FIXED = false, bool DIGEST = false>
__device__ __forceinline__ void synth_ev_body(const BatchDev &p, int16_t *__restrict__ iq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef typename std::conditional<DENSE, EvLds, EvLdsLean>::type LDS;
    LDS &L = *reinterpret_cast<LDS *>(smem_raw);

    const int tid = threadIdx.x;
#ifdef GPSBB_EV_PRIO /* (measurement: the synthesis wavefronts' issue priority against the pre-pass's, GPSBB_SEED_PRIO) */
    __builtin_amdgcn_s_setprio(GPSBB_EV_PRIO);
#endif
    if (lds_addr_of(smem_raw) != 0u) { /* ev_d_add addresses the image from LDS address 0 (the kernel has no static LDS) */
        if (tid == 0)
            atomicOr(p.status, ST_LDS_LAYOUT);
        return;
    }
#ifdef GPSBB_WG_TRACE /* measurement build (make trace; tools/corun_diag.py): when, where and at what clock every workgroup ran */
    WgTrace wgt = wg_trace_enter();
#endif
    const int b = ev_pick_block(p, (uint32_t)sizeof(LDS)); /* a primary's own block, or the block a helper joins */
#ifdef GPSBB_WG_TRACE
    wgt.block = b;
    wgt.nblocks = p.nblocks;
#endif
    if (b < 0) {
#ifdef GPSBB_WG_TRACE
        wg_trace_leave(wgt, wgt.wall0, wgt.clk0, 0u, 1u, false);
#endif
        return;
    }
    const gpsbb_chan_t *__restrict__ cb = p.ch + (size_t)b * p.nch;
    const EvConst *__restrict__ kb = p.evc + (size_t)b * p.nch;
    /* ---- stage the block's per-channel tables in LDS (once per workgroup) ---- */
    /* wavefront w stages channels w, w + 16, ...: what depends on the channel alone is wave-uniform */
    for (int i = __builtin_amdgcn_readfirstlane(tid >> 6); i < p.nch; i += EV_WAVES) {
        const int prn = cb[i].prn;
        const bool down = kb[i].down != 0; /* falling carrier: the table back to front (see ev_first) */
        const double g = cb[i].gain;
        /* fully unrolled: all table loads of a wavefront are in flight at once (a workgroup that joins a block late
         * idles its CU for as long as this staging takes) */
        int32_t tc[(LDS::AMP + 63) / 64], ts[(LDS::AMP + 63) / 64];
#pragma unroll
        for (int j = 0; j < (LDS::AMP + 63) / 64; j++) {
            const int e = (tid & 63) + 64 * j;
            const int k = down ? 511 - (e & 511) : (e & 511);
            tc[j] = p.tabs[k];
            ts[j] = p.tabs[512 + k];
        }
        /* the PRN's 1023 chips are 32 words: one per lane, fetched once, handed round with ds_bpermute */
        const uint32_t my_word = p.ca_bits[(prn > 0 ? prn : 0) * 32 + (tid & 31)];
#pragma unroll
        for (int j = 0; j < (LDS::AMP + 63) / 64; j++) {
            const int e = (tid & 63) + 64 * j;
            uint32_t v = 0;
            if (prn > 0) {
                /* (int)(table * gain): one IEEE multiply, truncation toward zero (plutogpssim.c:2701-2702) */
                const int ip = (int)mul_rn((double)tc[j], g);
                const int qp = (int)mul_rn((double)ts[j], g);
                v = ((uint32_t)qp << 16) + (uint32_t)ip;
            }
            if (e < LDS::AMP)
                L.amp[i][e] = v;
        }
#pragma unroll 4
        for (int j = 0; j < (LDS::CHIPS + 63) / 64; j++) {
            const int c = (tid & 63) + 64 * j;
            const int ca = c >= GPSBB_CA_LEN ? c - GPSBB_CA_LEN : c; /* c < 2 * 1023 */
            const int cb1 = c + 1 >= GPSBB_CA_LEN ? c + 1 - GPSBB_CA_LEN : c + 1;
            const uint32_t w0 = (uint32_t)__shfl((int)my_word, (ca >> 5) & 31), w1 = (uint32_t)__shfl((int)my_word, (cb1 >> 5) & 31);
            uint32_t v = 0;
            if (prn > 0) {
                const uint32_t b0 = (w0 >> (ca & 31)) & 1u;
                const uint32_t b1 = (w1 >> (cb1 & 31)) & 1u;
                v = (b0 ? 0x00u : 0xffu) | (b1 ? 0x0000u : 0xff00u);
            }
            if (c < LDS::CHIPS)
                L.chip2[i][c] = (uint16_t)v;
        }
    }
    for (int e = tid; e < EV_WAVES * 16 * 64; e += EV_WG)
        (&L.D[0][0][0])[e] = 0u;
    {
        const int l = tid & 63;
        if (l < p.nch) { /* every wavefront's own copy of the position constants (see EvLds::tstate) */
            L.tstate[tid >> 6][1][2 * l] = kb[l].tC0;     /* beside the code state */
            L.tstate[tid >> 6][1][2 * l + 1] = kb[l].tK0; /* beside the carrier state */
        }
    }
    __syncthreads();
#ifdef GPSBB_WG_TRACE
    const unsigned long long t_staged = wall_clock64(), c_staged = clock64();
#endif

    /* ---- from here on every wavefront works alone ---- */
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ntw = p.ntiles;
    const int nch2 = 2 * p.nch;
    /* channels by the number of carrier breakpoints a run can hold (bit i = channel i) */
    uint32_t mk[EV_KC_MAX];
    uint32_t mkd; /* evaluated per sample (channels that are always recomputed exactly are in mk[0]: EvConst::danger_le) */
    {
        const bool act = lane < p.nch && cb[lane < p.nch ? lane : 0].prn > 0;
        const int kc = act ? kb[lane].kc : 0;
#pragma unroll
        for (int k = 0; k < EV_KC_MAX; k++)
            mk[k] = (uint32_t)__ballot(act && (kc == k + 1 || (k == 0 && kc < 1)));
        mkd = DENSE ? (uint32_t)__ballot(act && kc == EV_KC_DENSE) : 0u;
    }
    /* lane c < 2*nch holds chain c = (channel c >> 1, kind c & 1) of the tile being staged */
    const bool chain_lane = lane < nch2;
    const bool mirror = chain_lane && (lane & 1) && kb[lane >> 1].down != 0;
    /* this lane's chain (tile-contiguous) in the tile arrays */
    const double *__restrict__ txb = p.tile_x + (size_t)b * ntw * nch2;
    /* (a scalar base and a 32-bit element offset per lane, not a 64-bit pointer per lane: the kernel is short of registers — what
     * it spills goes to HBM, DESIGN.md 3 — and a running pointer per lane is what the compiler spilled first) */
    const uint32_t tx_off = (uint32_t)(chain_lane ? lane : 0) * (uint32_t)ntw;
    const uint32_t *__restrict__ tnb = p.tile_nav + (size_t)b * p.nch * ntw;
    const uint32_t tn_off = (uint32_t)(lane < p.nch ? lane : 0) * (uint32_t)ntw;
    const double off = (double)(lane * SPT);
    /* this lane's chain in guard format, biased (the fixed-point carrier's index is exact: no bias, see ev_first) */
    const double guard_w = EV_GUARD + ((chain_lane && !(FIXED && (lane & 1))) ? kb[lane >> 1].W : 0.0);
    const double mirror_at = FIXED ? 512.0 - 0x1p-16 : 512.0; /* a falling fixed-point phase is mirrored bit by bit: (2^25 - 1 - p) / 2^16 */
    unsigned long long *n_exact = p.hazards + 2;
    /* this wavefront's difference arrays as raw LDS addresses (see the store at the end of a tile) */
    const uint32_t dwave = lds_addr_of(&L.D[wave][0][0]);
    const uint32_t d_col = dwave + (uint32_t)lane * 4u;                                    /* row j of this lane's column: + 256 j */
    const uint32_t t_wr = (dwave + (uint32_t)lane * 64u) ^ ((((uint32_t)lane >> 1) & 3u) << 4); /* piece q of this lane's 64 bytes: ^ 16 q */
    const uint32_t t_rd = dwave + ((uint32_t)lane >> 2) * 64u + ((((uint32_t)lane & 3u) ^ (((uint32_t)lane >> 3) & 3u)) << 4); /* + 1024 k */
    const uint32_t t_zero = dwave + (uint32_t)lane * 16u;
    uint32_t drow = (uint32_t)lane * 4u; /* ev_d_add's address register: byte 0 = this lane's column, byte 1 rewritten per difference */

    /* chunks of EV_CHUNK consecutive tiles from a per-block counter; the next chunk is asked for while the
     * current one is worked on, and a tile's states are fetched while the previous tile is worked on */
    int base = 0;
    if (lane == 0)
        base = atomicAdd(&p.tile_ctr[b], p.ev_chunk);
    base = __builtin_amdgcn_readfirstlane(base);
    int pos = 0;
    int pending = 0; /* lane 0: the next chunk, asked for at the first tile of the current one */
    unsigned tiles_rendered = 0; /* this wavefront's; summed into hazards[7] on the way out: every tile of every block exactly
                                    once is what the claim protocol (chunks handed out by an atomic per block) has to deliver,
                                    and the host can check it (GPSBB_INFO_TILES_RENDERED) */
    double ts_v = 0.0;
    uint32_t nav_v = 0;
    if (base < ntw) {
        ts_v = chain_lane ? txb[tx_off + (uint32_t)base] : 0.0;
        nav_v = lane < p.nch ? tnb[tn_off + (uint32_t)base] : 0u;
    }
    while (base < ntw) {
        const int wt = base + pos;
        /* the tile's states -> this wavefront's LDS slot, its data bits -> scalar masks */
        if (chain_lane)
            L.tstate[wave][0][lane] = (mirror ? mirror_at - ts_v : ts_v) + guard_w; /* one rounding, half a unit of 2^-32 */
        EvTile T;
        T.ts = L.tstate[wave][0];
        T.tile_x = txb + wt;
        T.ntiles = ntw;
        T.dbits = (uint32_t)__ballot(nav_v & 1u);
        T.dnext = (uint32_t)__ballot(nav_v & 2u);
        EvFixed fx;
        fx.ph = FIXED ? p.kph0 + (size_t)b * p.nch : nullptr;
        fx.st = FIXED ? p.kstep + (size_t)b * p.nch : nullptr;
        fx.n0 = wt * TILE;
        const uint32_t dflip = T.dbits ^ T.dnext;
        /* which tile comes next, and its states on their way */
        if (pos == 0 && lane == 0) {
            /* the next chunk (see GPSBB_EV_SETTLE_CLAIM) */
            pending = __hip_atomic_fetch_add(p.tile_ctr + b, p.ev_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool last_of_chunk = pos + 1 >= p.ev_chunk || wt + 1 >= ntw;
        int next_base = base, next_pos = pos + 1;
        if (last_of_chunk) {
            next_base = __builtin_amdgcn_readfirstlane(pending);
            next_pos = 0;
        }
        const int wt_next = next_base + next_pos;
        if (wt_next < ntw) {
            ts_v = chain_lane ? txb[tx_off + (uint32_t)wt_next] : 0.0;
            nav_v = lane < p.nch ? tnb[tn_off + (uint32_t)wt_next] : 0u;
        }

        const int n0 = wt * TILE + lane * SPT;
        const int nvalid = p.nsamp - n0 < SPT ? p.nsamp - n0 : SPT;
        const bool lane_live = nvalid > 0;
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(lane_live);
        uint32_t acc0 = 0x8000u; /* the I sum travels biased by 2^15: never negative, so the low half never borrows from the Q sum */
        ev_channels<1, false, FIXED>(L, drow, wave, lane, mk[0] & ~dflip, kb, T, off, live_mask, acc0, n_exact, fx);
        ev_channels<2, false, FIXED>(L, drow, wave, lane, mk[1] & ~dflip, kb, T, off, live_mask, acc0, n_exact, fx);
        if (__builtin_expect((mk[2] | mk[3] | dflip) != 0u, 0)) {
            ev_channels<3, false, FIXED>(L, drow, wave, lane, mk[2] & ~dflip, kb, T, off, live_mask, acc0, n_exact, fx);
            ev_channels<4, false, FIXED>(L, drow, wave, lane, mk[3] & ~dflip, kb, T, off, live_mask, acc0, n_exact, fx);
            ev_channels<1, true, FIXED>(L, drow, wave, lane, mk[0] & dflip, kb, T, off, live_mask, acc0, n_exact, fx);
            ev_channels<2, true, FIXED>(L, drow, wave, lane, mk[1] & dflip, kb, T, off, live_mask, acc0, n_exact, fx);
            ev_channels<3, true, FIXED>(L, drow, wave, lane, mk[2] & dflip, kb, T, off, live_mask, acc0, n_exact, fx);
            ev_channels<4, true, FIXED>(L, drow, wave, lane, mk[3] & dflip, kb, T, off, live_mask, acc0, n_exact, fx);
        }
        /* ---- prefix sum over the run, back to int16 pairs (c:2754-2755) ---- */
        uint32_t o[SPT];
        uint32_t P = acc0;
        uint32_t accd[DENSE ? SPT : 1];
        if (DENSE) {
#pragma unroll
            for (int j = 0; j < SPT; j++)
                accd[j] = 0u;
            /* channels evaluated per sample: their sums per sample, on top of the prefix sums of the others */
            for (uint32_t m = mkd; m; m &= m - 1) {
                const int i = __builtin_ctz(m);
                if ((dflip >> i) & 1u)
                    ev_dense<true>(L, wave, lane, i, kb, T, off, live_mask, P, *reinterpret_cast<uint32_t (*)[SPT]>(&accd[0]), n_exact);
                else
                    ev_dense<false>(L, wave, lane, i, kb, T, off, live_mask, P, *reinterpret_cast<uint32_t (*)[SPT]>(&accd[0]), n_exact);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            if (j)
                P += lds_read_at<uint32_t>(d_col + (uint32_t)(j - 1) * 256u); /* this lane's column of the difference arrays */
            o[j] = (DENSE ? P + accd[j] : P) ^ 0x8000u; /* the bias off again: the low half is I + 2^15 in [0, 2^16), the high half Q */
        }
#ifdef GPSBB_SABOTAGE /* a deliberately wrong build (make broken): bench.py's parity check must refuse it (tests/test_bench_shards.py) */
        /* (1: a block bench.py's oracle legs visit; 2: block 6 of every push, which — with two spot checks per shard — they do not:
         * only the cross-check of every block against the per-sample kernel sees it) */
        if (b == (GPSBB_SABOTAGE == 2 ? 6 : 1) && wt == 3 && lane == 5)
            o[7] ^= 1u;
#endif
        if (DIGEST) {
            /* the block's digest as it is rendered (gpsbb_device_digest's number: the sum of pair_j * m_j, k_block_digest): the
             * lane's 16 samples are in registers here, and reading the 4 GB of a push back for it costs the synthesis beside it a
             * third of its rate */
            uint32_t m = digest_weight((uint32_t)n0);
            unsigned long long dg = 0ull;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                if (j < nvalid)
                    dg += (unsigned long long)o[j] * m;
                m += DIGEST_STEP;
            }
#pragma unroll
            for (int off2 = 32; off2 > 0; off2 >>= 1)
                dg += (unsigned long long)__shfl_down((long long)dg, off2);
            if (lane == 0 && dg)
                atomicAdd(p.digest + b, dg);
        }
        /* ---- store.  A lane holds 16 consecutive samples = 64 bytes, and a store instruction moves 16 bytes per lane: written
         * straight from the registers, every instruction touches 64 different 64-byte pieces a quarter full, four times the
         * write requests the data needs — measured, that pattern alone cost 0.27 ms of the kernel's 1.86 (tools/bound_hunt.sh:
         * the same bytes stored 1 KB-contiguous per instruction, 1.61 ms).  So the tile is transposed on the way out, through
         * the wavefront's own 4 KB of difference arrays — the tile's 1024 int16 pairs are exactly that big, and the arrays have
         * just been read and have to be zeroed for the next tile anyway: every lane writes its 64 bytes as four 16-byte pieces
         * (piece q at slot q ^ (lane / 2 & 3): conflict-free), reads back the piece of the run 16 k + lane / 4 that instruction k
         * stores, and the region is zeroed with four 16-byte stores per lane instead of fifteen exchanges.  A wavefront's LDS
         * operations execute in order and the region is its own: no barrier. ---- */
        uint32_t *const tile_out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + (size_t)wt * TILE;
        const bool whole = (wt + 1) * TILE <= p.nsamp && (reinterpret_cast<uintptr_t>(tile_out) & 15u) == 0; /* wave-uniform */
        asm volatile("" ::: "memory");
        if (__builtin_expect(whole, 1)) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                lds_write_at<ev_u32x4>(t_wr ^ ((uint32_t)q << 4), ev_u32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]});
            asm volatile("" ::: "memory");
            ev_u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                v[k] = lds_read_at<ev_u32x4>(t_rd + (uint32_t)k * 1024u);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; k++)
                lds_write_at<ev_u32x4>(t_zero + (uint32_t)k * 1024u, ev_u32x4{0u, 0u, 0u, 0u});
            /* (the lane's offset made opaque here: hoisted out of the tile loop it is one more 64-bit value to keep — and the
             * compiler kept it in scratch, a reload and a store with every tile that went to HBM: DESIGN.md 3) */
            uint32_t lane_now = (uint32_t)lane;
            asm volatile("" : "+v"(lane_now));
            ev_u32x4 *g = reinterpret_cast<ev_u32x4 *>(tile_out) + lane_now;
#pragma unroll
            for (int k = 0; k < 4; k++)
                g[k * 64] = v[k]; /* 1 KB of consecutive addresses per instruction */
        } else {
            /* the block's last, partial tile (or a block that does not start on a 16-byte boundary): from the registers */
#pragma unroll
            for (int j = 1; j < SPT; j++)
                lds_write_at<uint32_t>(d_col + (uint32_t)(j - 1) * 256u, 0u);
            uint32_t *out = tile_out + lane * SPT;
#pragma unroll
            for (int j = 0; j < SPT; j++)
                if (j < nvalid)
                    out[j] = o[j];
        }
        asm volatile("" ::: "memory");
        base = next_base;
        pos = next_pos;
        tiles_rendered++;
    }
    if (lane == 0 && tiles_rendered)
        atomicAdd(p.hazards + 7, (unsigned long long)tiles_rendered);
#ifdef GPSBB_WG_TRACE
    wg_trace_leave(wgt, t_staged, c_staged, tiles_rendered, 1u, true); /* wavefront 0's tiles and its own clocks: it need not be the last to leave */
#endif
}

/* The common kernel is held to the register budget of five wavefronts per SIMD (96 VGPRs; the allocator's count
 * wanders between 97 and 117 with the size of the chip table): four of them then leave room for a walk wavefront. */
#ifdef GPSBB_EV_VGPRS /* (measurement: a register budget instead of five wavefronts per SIMD's 96) */
#define GPSBB_EV_BUDGET __attribute__((amdgpu_num_vgpr(GPSBB_EV_VGPRS)))
#else
#define GPSBB_EV_BUDGET __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
__global__ __launch_bounds__(EV_WG) GPSBB_EV_BUDGET void k_synth_ev(BatchDev p, int16_t *__restrict__ iq)
{
    synth_ev_body<false>(p, iq);
}
/* ... that also leaves every block's digest (BatchDev::digest, zeroed by the host before the launch) */
__global__ __launch_bounds__(EV_WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_synth_ev_digest(BatchDev p, int16_t *__restrict__ iq)
{
    synth_ev_body<false, false, true>(p, iq);
}
__global__ __launch_bounds__(EV_WG) void k_synth_ev_dense(BatchDev p, int16_t *__restrict__ iq)
{
    synth_ev_body<true>(p, iq);
}
/* the fixed-point carrier (GPSBB_FIXED_CARRIER) above ~15.9 MS/s: the same kernel with an exact carrier model (ev_first) */
__global__ __launch_bounds__(EV_WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_synth_ev_fixed(BatchDev p, int16_t *__restrict__ iq)
{
    synth_ev_body<false, true>(p, iq);
}

} /* namespace gpsbb_impl */
#endif
