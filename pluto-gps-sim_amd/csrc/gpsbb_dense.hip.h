/*
 * gpsbb_dense.hip.h — k_synth_pd: the sample loop (plutogpssim.c:2690-2756) for batches in which EVERY channel is
 * evaluated per sample (EvConst::kc == EV_KC_DENSE): sample rates below ~15.9 MS/s, where a run of 16 samples holds
 * more than one chip change — the reference's own operating point (2.6 / 3 MS/s, 300 000-sample blocks, c:43-45)
 * above all.  Same in-tile model as k_synth_ev / ev_dense (gpsbb_events.hip.h: the exact tile-start states of the
 * lockstep pre-pass plus n * step, trusted wherever it stays clear of an integer, else that lane's samples of that
 * channel are recomputed with genuine IEEE steps), arranged for what the per-sample work costs on gfx950:
 *
 *   lane = sample mod 64.  A lane's 16 samples of a tile are 64 apart, so the 64 lanes of a wavefront look at 64
 *       CONSECUTIVE samples: their table indices are consecutive or equal (|step| <= 1 entry per sample, 0.4 chips), and
 *       the two LDS reads per channel-sample are conflict-free (random reads cost three times as much, and with 16
 *       consecutive samples per lane — the other kernels' arrangement — the LDS, not the VALU, would bound this one).
 *       The stores are 16 dword stores of 256 contiguous bytes per wavefront.
 *   the model in "guard format": 2^20 + (8 * table index | 2 * chip + table address).  One unit in the last place is
 *       2^-32, so the double's low word IS the fraction and the low bits of its high word ARE the byte address of the
 *       entry: one v_and(_or) per lookup, and the test "did the model come within its error of an integer" is the minimum
 *       of the low words over the run (v_min3_u32, one per channel-sample for both NCOs), compared once.  From one
 *       sample of a lane to its next the models advance by one v_add_f64 each (the sums' roundings, half a unit of 2^-32
 *       apiece, are part of PD_BAND).
 *   amplitudes as float pairs, chips as +-1.0: a channel-sample's contribution is ONE packed FMA (I and Q together);
 *       sums of at most 16 integers below 2^15 are exact in binary32.  No sign extraction, no xor / sub / add chain: the
 *       chips are the upper halves of binary32 +-1.0, read with ds_read_u16_d16_hi into registers whose lower halves
 *       stay zero, and the data bit picks one of two chip tables (the second one negated) through the model's address.
 *   the 16 samples of a channel in ONE block of assembly (pd_channel_fast_wide / _narrow): six VALU instructions per channel-sample
 *       (2 v_add_f64, v_and_or, v_and, v_min3, v_pk_fma) and the LDS reads of four samples in flight behind counted
 *       waits — the compiler's own schedule of the same source waited for every pair of reads on the spot, copied the
 *       32 accumulators once per channel and moved both models through a register pair per sample: 12 instructions.
 *
 * About 27 issue cycles per channel-sample against the 58 of ev_dense (tools/ubench/valu_rates.hip); the two LDS reads
 * (12 bytes per lane) cost the CU about as much, so the kernel sits where VALU and LDS meet.
 */
#ifndef GPSBB_DENSE_HIP_H
#define GPSBB_DENSE_HIP_H

#include <type_traits>

#include "gpsbb_events.hip.h"

namespace gpsbb_impl {

typedef float v2f __attribute__((ext_vector_type(2)));

/* |model - truth| in units of 2^-32 of the scaled models, roundings of the guard format included.  Summed by hand: carrier
 * 8 * 2^-33.9 * 2^32 = 2.2, code 2 * 0.27; the tile state's and the first sample's fma half a unit each, the 15 additions
 * after it 7.5 (NOT random: the fraction of dy / dx the format drops is the same at every addition, so the half units pile
 * up): 11.24.  MEASURED over every tile of the corner workloads (tools/model_err.py): 8.55.  Round 3 had 12 here — 7 % above
 * the sum; the band is now more than twice what is realised. */
#ifndef GPSBB_PD_BAND
#define GPSBB_PD_BAND 20
#endif
constexpr uint32_t PD_BAND = GPSBB_PD_BAND;

/* WIDE: up to PD_WIDE_CHAN channels (the reference's MAX_CHAN, h:21): two chip tables, the second one negated, so that
 * the data bit in force is an address offset; else up to GPSBB_MAX_CHAN with one table and the data bit as a sign
 * modifier of the packed FMA (160 KB of LDS do not hold 16 channels' tables twice) */
constexpr int PD_WIDE_CHAN = 12;
constexpr float PD_ACC0 = 12582912.0f; /* 1.5 * 2^23: where the accumulators start (see k_synth_pd) */
template <bool WIDE>
struct PdLds {
    static constexpr int NCH = WIDE ? PD_WIDE_CHAN : GPSBB_MAX_CHAN;
    static constexpr int NTAB = WIDE ? 2 : 1;
    v2f amp[NCH][512];                     /* ((float)(int)(cos*gain), (float)(int)(sin*gain)) of table index k (a falling
                                              carrier: of 511 - k, see ev_first) */
    uint16_t chipf[NTAB][NCH][EV_CHIP_LEN]; /* upper half of binary32 +1.0 / -1.0: codeCA of chip c mod 1023; [1]: negated */
    double tstate[EV_WAVES][2][2 * GPSBB_MAX_CHAN]; /* per wavefront, two tiles deep: the tile's models at sample 0 in guard
                                              format: column 2*channel = 2^20 + band + address of chipf[0][channel] + 2 * code
                                              phase, 2*channel + 1 = 2^20 + band + 8 * carrier phase (mirrored) */
};

/* what the fast path added for sample j*64 + lane of a channel: the models exactly as pd_channel_fast advances them (one
 * fma, then j additions), index and chip as floor(model), the data bit before / after the code's roll-over */
struct PdModel {
    double ytg, xtg, S8, sc2, dy, dx; /* the tile's models (xtg: the plain chip table's) and their steps */
    uint32_t amp_base, roll_addr;
    uint32_t neg, neg_next;           /* the data bit in force at the tile start / after the roll-over is -1 */
};
__device__ __forceinline__ v2f pd_model_sample(const PdModel &M, int lane, int j)
{
    double y = __fma_rn((double)lane, M.S8, M.ytg), x = __fma_rn((double)lane, M.sc2, M.xtg);
#pragma unroll 1
    for (int q = 0; q < j; q++) {
        y = __dadd_rn(y, M.dy);
        x = __dadd_rn(x, M.dx);
    }
    const uint32_t ia = ((uint32_t)__double2hiint(y) & 0xff8u) | M.amp_base;
    const uint32_t ic = (uint32_t)__double2hiint(x) & 0xffffeu;
    const uint32_t ng = ic >= M.roll_addr ? M.neg_next : M.neg;
    const float sg = __uint_as_float(((uint32_t)lds_read_at<uint16_t>(ic) << 16) ^ (ng ? 0x80000000u : 0u));
    const v2f a = lds_read_at<v2f>(ia);
    v2f t;
    t.x = sg * a.x;
    t.y = sg * a.y;
    return t;
}

/*
 * Rare: a lane cannot rule out that the model and the reference disagree somewhere in its 16 samples of channel i.
 * For each of them it takes back what the model added and takes the true contribution instead: from the tile's exact
 * state to sample n with the exact jump-ahead (gpsbb_nco.h), then index, chip and data bit as the reference has them
 * there (c:2697-2737).  Out of line, and with nothing but scalars in and out, so that the accumulators of the fast path
 * stay in registers.  Returns (true contribution) - (the model's).
 */
template <bool WIDE>
__device__ __noinline__ v2f pd_fix_sample(const PdLds<WIDE> &L, int i, const EvConst *kbi, const double *tile_x, int ntiles, double ytg,
                                          double xtg, uint32_t amp_base, uint32_t roll_addr, uint32_t negs, int lane, int j, int fixed,
                                          uint32_t fx_phase, int32_t fx_step)
{
    PdModel M;
    M.ytg = ytg;
    M.xtg = xtg;
    M.S8 = kbi->pd_S8;
    M.sc2 = kbi->pd_sc2;
    M.dy = kbi->pd_dy;
    M.dx = kbi->pd_dx;
    M.amp_base = amp_base;
    M.roll_addr = roll_addr;
    M.neg = negs & 1u;
    M.neg_next = (negs >> 1) & 1u;
    const int n = j * 64 + lane;
    const bool down = kbi->down != 0;
    const double S = down ? -kbi->S : kbi->S, sc = kbi->sc;
    const double xt = tile_x[(size_t)(2 * i) * ntiles], yt = tile_x[(size_t)(2 * i + 1) * ntiles];
    int64_t wraps = 0;
    const double x = code_jump(xt, sc, (int64_t)n, &wraps);
    const bool neg = wraps > 0 ? M.neg_next != 0 : M.neg != 0; /* at most one roll-over per tile (checked by the host) */
    int it;
    if (fixed) {
        it = (int)(((fx_phase + (uint32_t)n * (uint32_t)fx_step) >> 16) & 0x1ffu); /* c:2699: the 32-bit accumulator */
    } else {
        const double cp = carr_jump(yt * (1.0 / 512.0), S * (1.0 / 512.0), (int64_t)n);
        it = (int)(cp * 512.0) & 511; /* c:2697; carr_phase == 1.0: index 512 defined as 0 */
    }
    const int ci = (int)x;                  /* c:2737 */
    const float sg = __uint_as_float(((uint32_t)L.chipf[0][i][ci] << 16) ^ (neg ? 0x80000000u : 0u));
    const v2f a = L.amp[i][down ? 511 - it : it];
    const v2f mdl = pd_model_sample(M, lane, j);
    v2f t;
    t.x = sg * a.x - mdl.x; /* exact: integers below 2^15 */
    t.y = sg * a.y - mdl.y;
    return t;
}

/*
 * One channel of one tile, SPT samples per lane, 64 apart: ONE statement of assembly, so that the 16 accumulators have one
 * producer per channel and stay where they are (two producers — this block and a C++ path for the tiles in which the data
 * bit changes — cost 64 register copies per channel-tile at the join).  y = v[72:73], x = v[74:75]: the models; four
 * samples' LDS reads in flight: amplitude pairs in v[76:83], chips (d16_hi, lower halves zero) in v84 / v86 / v88 / v90;
 * v92, v93: addresses.  LDS returns in order, so "lgkmcnt(6)" = all but the last three samples' reads have landed
 * (whatever else may be outstanding from before the block only makes the waits longer).  The data bit in force at the
 * tile start is in xtg's table address; where it changes inside the tile (df) the second body moves the addresses past the
 * code's roll-over (>= roll) by delta, into the other table.
 */
#define GPSBB_PD_ADDR_F                                                                                                \
    "v_and_or_b32 v92, v73, %[msk], %[ab]\n"                                                                           \
    "v_and_b32 v93, 0xffffe, v75\n"                                                                                    \
    "v_min3_u32 %[m], %[m], v72, v74\n"
/* fixed-point carrier: its model is exact (multiples of 2^-13, sums without rounding): only the code model is tested */
#define GPSBB_PD_ADDR_X                                                                                                \
    "v_and_or_b32 v92, v73, %[msk], %[ab]\n"                                                                           \
    "v_and_b32 v93, 0xffffe, v75\n"                                                                                    \
    "v_min_u32 %[m], %[m], v74\n"
#define GPSBB_PD_ISSUE(A0, A1, C, C1) GPSBB_PD_ISSUE_(GPSBB_PD_ADDR_F, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_X(A0, A1, C, C1) GPSBB_PD_ISSUE_(GPSBB_PD_ADDR_X, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_DFW(A0, A1, C, C1) GPSBB_PD_ISSUE_DFW_(GPSBB_PD_ADDR_F, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_DFW_X(A0, A1, C, C1) GPSBB_PD_ISSUE_DFW_(GPSBB_PD_ADDR_X, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_DFN(A0, A1, C, C1) GPSBB_PD_ISSUE_DFN_(GPSBB_PD_ADDR_F, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_DFN_X(A0, A1, C, C1) GPSBB_PD_ISSUE_DFN_(GPSBB_PD_ADDR_X, A0, A1, C, C1)
#define GPSBB_PD_ISSUE_(GPSBB_PD_ADDR, A0, A1, C, C1)                                                                  \
    GPSBB_PD_ADDR                                                                                                      \
    "ds_read_b64 v[" #A0 ":" #A1 "], v92\n"                                                                            \
    "ds_read_u16_d16_hi v" #C ", v93\n"
/* WIDE, the data bit changes: past the roll-over the other table */
#define GPSBB_PD_ISSUE_DFW_(GPSBB_PD_ADDR, A0, A1, C, C1)                                                              \
    GPSBB_PD_ADDR                                                                                                      \
    "ds_read_b64 v[" #A0 ":" #A1 "], v92\n"                                                                            \
    "v_cmp_le_u32 vcc, %[roll], v93\n"                                                                                 \
    "v_cndmask_b32 v92, 0, %[delta], vcc\n"                                                                            \
    "v_add_u32 v93, v93, v92\n"                                                                                        \
    "ds_read_u16_d16_hi v" #C ", v93\n"
/* one table, the data bit changes: the sample's sign bit next to its chip (the upper register of the chip's pair) */
#define GPSBB_PD_ISSUE_DFN_(GPSBB_PD_ADDR, A0, A1, C, C1)                                                              \
    GPSBB_PD_ADDR                                                                                                      \
    "ds_read_b64 v[" #A0 ":" #A1 "], v92\n"                                                                            \
    "v_cmp_le_u32 vcc, %[roll], v93\n"                                                                                 \
    "v_cndmask_b32 v" #C1 ", %[sgn0], %[sgn1], vcc\n"                                                                  \
    "ds_read_u16_d16_hi v" #C ", v93\n"
#define GPSBB_PD_STEP                                                                                                  \
    "v_add_f64 v[72:73], v[72:73], %[dy]\n"                                                                            \
    "v_add_f64 v[74:75], v[74:75], %[dx]\n"
#define GPSBB_PD_FMA(ACC, A0, A1, C0, C1, CNT)                                                                         \
    "s_waitcnt lgkmcnt(" #CNT ")\n"                                                                                    \
    "v_pk_fma_f32 %[" #ACC "], v[" #C0 ":" #C1 "], v[" #A0 ":" #A1 "], %[" #ACC "] op_sel_hi:[0,1,1]\n"
#define GPSBB_PD_FMA_NEG(ACC, A0, A1, C0, C1, CNT)                                                                     \
    "s_waitcnt lgkmcnt(" #CNT ")\n"                                                                                    \
    "v_pk_fma_f32 %[" #ACC "], v[" #C0 ":" #C1 "], v[" #A0 ":" #A1 "], %[" #ACC "] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
#define GPSBB_PD_FMA_SGN(ACC, A0, A1, C0, C1, CNT)                                                                     \
    "s_waitcnt lgkmcnt(" #CNT ")\n"                                                                                    \
    "v_xor_b32 v" #C0 ", v" #C0 ", v" #C1 "\n"                                                                         \
    "v_pk_fma_f32 %[" #ACC "], v[" #C0 ":" #C1 "], v[" #A0 ":" #A1 "], %[" #ACC "] op_sel_hi:[0,1,1]\n"
#define GPSBB_PD_BODY(ISSUE, FMA)                                                                                      \
    ISSUE(76, 77, 84, 85) GPSBB_PD_STEP                                                                                \
    ISSUE(78, 79, 86, 87) GPSBB_PD_STEP                                                                                \
    ISSUE(80, 81, 88, 89) GPSBB_PD_STEP                                                                                \
    ISSUE(82, 83, 90, 91) GPSBB_PD_STEP                                                                                \
    FMA(a0, 76, 77, 84, 85, 6) ISSUE(76, 77, 84, 85) GPSBB_PD_STEP                                                     \
    FMA(a1, 78, 79, 86, 87, 6) ISSUE(78, 79, 86, 87) GPSBB_PD_STEP                                                     \
    FMA(a2, 80, 81, 88, 89, 6) ISSUE(80, 81, 88, 89) GPSBB_PD_STEP                                                     \
    FMA(a3, 82, 83, 90, 91, 6) ISSUE(82, 83, 90, 91) GPSBB_PD_STEP                                                     \
    FMA(a4, 76, 77, 84, 85, 6) ISSUE(76, 77, 84, 85) GPSBB_PD_STEP                                                     \
    FMA(a5, 78, 79, 86, 87, 6) ISSUE(78, 79, 86, 87) GPSBB_PD_STEP                                                     \
    FMA(a6, 80, 81, 88, 89, 6) ISSUE(80, 81, 88, 89) GPSBB_PD_STEP                                                     \
    FMA(a7, 82, 83, 90, 91, 6) ISSUE(82, 83, 90, 91) GPSBB_PD_STEP                                                     \
    FMA(a8, 76, 77, 84, 85, 6) ISSUE(76, 77, 84, 85) GPSBB_PD_STEP                                                     \
    FMA(a9, 78, 79, 86, 87, 6) ISSUE(78, 79, 86, 87) GPSBB_PD_STEP                                                     \
    FMA(a10, 80, 81, 88, 89, 6) ISSUE(80, 81, 88, 89) GPSBB_PD_STEP                                                    \
    FMA(a11, 82, 83, 90, 91, 6) ISSUE(82, 83, 90, 91)                                                                  \
    FMA(a12, 76, 77, 84, 85, 6)                                                                                        \
    FMA(a13, 78, 79, 86, 87, 4)                                                                                        \
    FMA(a14, 80, 81, 88, 89, 2)                                                                                        \
    FMA(a15, 82, 83, 90, 91, 0)
#define GPSBB_PD_HEAD                                                                                                  \
    "v_fma_f64 v[72:73], %[lf], %[s8], %[ytg]\n"                                                                       \
    "v_fma_f64 v[74:75], %[lf], %[sc2], %[xtg]\n"                                                                      \
    "v_mov_b32 v84, 0\n"                                                                                               \
    "v_mov_b32 v86, 0\n"                                                                                               \
    "v_mov_b32 v88, 0\n"                                                                                               \
    "v_mov_b32 v90, 0\n"
#define GPSBB_PD_ACCS                                                                                                  \
    [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]),   \
        [a6] "+v"(acc[6]), [a7] "+v"(acc[7]), [a8] "+v"(acc[8]), [a9] "+v"(acc[9]), [a10] "+v"(acc[10]),               \
        [a11] "+v"(acc[11]), [a12] "+v"(acc[12]), [a13] "+v"(acc[13]), [a14] "+v"(acc[14]), [a15] "+v"(acc[15]),       \
        [m] "+v"(m)
#define GPSBB_PD_CLOBBERS                                                                                              \
    "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",     \
        "v88", "v89", "v90", "v91", "v92", "v93", "vcc", "scc"
/* WIDE: two chip tables.  The data bit in force at the tile start is in xtg's table address; where it changes inside the
 * tile (df) the second body moves the addresses past the code's roll-over (>= roll) by delta, into the other table. */
__device__ __forceinline__ uint32_t pd_channel_fast_wide(const PdModel &M, double xtg, double lf, uint32_t sel, uint32_t roll, int32_t delta,
                                                         v2f (&acc)[SPT])
{
    static_assert(SPT == 16, "the block below is written for 16 samples per lane");
    uint32_t m = 0xffffffffu;
    /* sel: bit 0 = the data bit changes inside the tile, bit 1 = fixed-point carrier (no test of the carrier model) */
    asm volatile(GPSBB_PD_HEAD
                 "s_bitcmp1_b32 %[sel], 1\n"
                 "s_cbranch_scc1 4f\n"
                 "s_bitcmp1_b32 %[sel], 0\n"
                 "s_cbranch_scc1 1f\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE, GPSBB_PD_FMA)
                 "s_branch 2f\n"
                 "1:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_DFW, GPSBB_PD_FMA)
                 "s_branch 2f\n"
                 "4:\n"
                 "s_bitcmp1_b32 %[sel], 0\n"
                 "s_cbranch_scc1 5f\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_X, GPSBB_PD_FMA)
                 "s_branch 2f\n"
                 "5:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_DFW_X, GPSBB_PD_FMA)
                 "2:\n"
                 : GPSBB_PD_ACCS
                 : [lf] "v"(lf), [ytg] "v"(M.ytg), [xtg] "v"(xtg), [s8] "s"(M.S8), [sc2] "s"(M.sc2), [dy] "s"(M.dy), [dx] "s"(M.dx),
                   [ab] "v"(M.amp_base), [msk] "s"(0xff8u), [sel] "s"(sel), [roll] "s"(roll), [delta] "v"(delta)
                 : GPSBB_PD_CLOBBERS);
    return m;
}
/* One chip table (13 to 16 channels): the data bit as a sign modifier of the packed FMA; where it changes inside the tile
 * every sample carries its own sign bit (sgn0 before the roll-over, sgn1 past it). */
__device__ __forceinline__ uint32_t pd_channel_fast_narrow(const PdModel &M, double lf, uint32_t fixed, v2f (&acc)[SPT])
{
    uint32_t m = 0xffffffffu;
    const uint32_t sgn0 = M.neg ? 0x80000000u : 0u, sgn1 = M.neg_next ? 0x80000000u : 0u;
    /* sel: bit 0 = the data bit changes inside the tile, bit 1 = fixed-point carrier, bit 2 = the data bit in force is -1 */
    const uint32_t sel = (M.neg ^ M.neg_next) | (fixed << 1) | (M.neg << 2);
    asm volatile(GPSBB_PD_HEAD
                 "s_bitcmp1_b32 %[sel], 1\n"
                 "s_cbranch_scc1 4f\n"
                 "s_bitcmp1_b32 %[sel], 0\n"
                 "s_cbranch_scc1 1f\n"
                 "s_bitcmp1_b32 %[sel], 2\n"
                 "s_cbranch_scc1 3f\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE, GPSBB_PD_FMA)
                 "s_branch 2f\n"
                 "3:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE, GPSBB_PD_FMA_NEG)
                 "s_branch 2f\n"
                 "1:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_DFN, GPSBB_PD_FMA_SGN)
                 "s_branch 2f\n"
                 "4:\n"
                 "s_bitcmp1_b32 %[sel], 0\n"
                 "s_cbranch_scc1 5f\n"
                 "s_bitcmp1_b32 %[sel], 2\n"
                 "s_cbranch_scc1 6f\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_X, GPSBB_PD_FMA)
                 "s_branch 2f\n"
                 "6:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_X, GPSBB_PD_FMA_NEG)
                 "s_branch 2f\n"
                 "5:\n"
                 GPSBB_PD_BODY(GPSBB_PD_ISSUE_DFN_X, GPSBB_PD_FMA_SGN)
                 "2:\n"
                 : GPSBB_PD_ACCS
                 : [lf] "v"(lf), [ytg] "v"(M.ytg), [xtg] "v"(M.xtg), [s8] "s"(M.S8), [sc2] "s"(M.sc2), [dy] "s"(M.dy), [dx] "s"(M.dx),
                   [ab] "v"(M.amp_base), [msk] "s"(0xff8u), [sel] "s"(sel), [roll] "s"(M.roll_addr), [sgn0] "v"(sgn0), [sgn1] "v"(sgn1)
                 : GPSBB_PD_CLOBBERS);
    return m;
}
#undef GPSBB_PD_ADDR_F
#undef GPSBB_PD_ADDR_X
#undef GPSBB_PD_ISSUE
#undef GPSBB_PD_ISSUE_X
#undef GPSBB_PD_ISSUE_
#undef GPSBB_PD_ISSUE_DFW
#undef GPSBB_PD_ISSUE_DFW_X
#undef GPSBB_PD_ISSUE_DFW_
#undef GPSBB_PD_ISSUE_DFN
#undef GPSBB_PD_ISSUE_DFN_X
#undef GPSBB_PD_ISSUE_DFN_
#undef GPSBB_PD_STEP
#undef GPSBB_PD_FMA
#undef GPSBB_PD_FMA_NEG
#undef GPSBB_PD_FMA_SGN
#undef GPSBB_PD_BODY
#undef GPSBB_PD_HEAD
#undef GPSBB_PD_ACCS
#undef GPSBB_PD_CLOBBERS

/* which of a lane's 16 samples of a channel the fast path cannot vouch for: bit j = the low word of one of the models at sample
 * j * 64 + lane, computed exactly as the fast path computes it, is below the threshold (for the lanes that have to be looked at
 * again: the fast path itself only keeps the minimum over the 16).  The test is per sample — index and chip of a sample are
 * floor() of that sample's models and of nothing else — so only these samples are recomputed: until round 6 a flagged lane
 * recomputed all 16, sixteen jump-aheads from the tile start where one was needed, and the wavefront it sat in was busy for
 * 200 - 400 us: in one block out of fifteen of the reference's geometry, and whenever that block was among the last of a
 * launch the whole chip waited for it (profiles/r06_corun_diag.txt: 1.07 ms per launch of which the last 0.14 were 3 blocks). */
__device__ __forceinline__ uint32_t pd_model_flagged(const PdModel &M, int lane, bool fixed, uint32_t danger)
{
    double y = __fma_rn((double)lane, M.S8, M.ytg), x = __fma_rn((double)lane, M.sc2, M.xtg);
    uint32_t bad = 0u;
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        const uint32_t m = min(fixed ? 0xffffffffu : (uint32_t)__double2loint(y), (uint32_t)__double2loint(x));
        bad |= m < danger ? 1u << j : 0u;
        y = __dadd_rn(y, M.dy);
        x = __dadd_rn(x, M.dx);
    }
    return bad;
}

/* channel i's models for the tile whose states are at ts */
template <bool WIDE>
__device__ __forceinline__ void pd_model_of(const PdLds<WIDE> &L, const EvConst *kb, const double *ts, int i, uint32_t dbits, uint32_t dnext, PdModel &M)
{
    M.S8 = scalar_load(&kb[i].pd_S8);
    M.dy = scalar_load(&kb[i].pd_dy);
    M.sc2 = scalar_load(&kb[i].pd_sc2);
    M.dx = scalar_load(&kb[i].pd_dx);
    M.xtg = ts[2 * i];
    M.ytg = ts[2 * i + 1];
    M.amp_base = lds_addr_of(&L.amp[i][0]);
    M.roll_addr = lds_addr_of(&L.chipf[0][i][GPSBB_CA_LEN]);
    M.neg = (dbits >> i) & 1u;
    M.neg_next = (dnext >> i) & 1u;
}

/* (DIGEST: the kernel also leaves every block's digest — BatchDev::digest, GPSBB_PUSH_DIGEST) */
template <bool WIDE, bool DIGEST = false>
__global__ __launch_bounds__(EV_WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_synth_pd(BatchDev p, int16_t *__restrict__ iq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PdLds<WIDE> &L = *reinterpret_cast<PdLds<WIDE> *>(smem_raw);
#ifdef GPSBB_EV_PRIO /* (measurement, as in k_synth_ev) */
    __builtin_amdgcn_s_setprio(GPSBB_EV_PRIO);
#endif
    const int tid = threadIdx.x;
#ifdef GPSBB_WG_TRACE /* measurement build: see wg_trace_leave */
    WgTrace wgt = wg_trace_enter();
#endif
    const int b = ev_pick_block(p, (uint32_t)sizeof(PdLds<WIDE>)); /* a primary's own block, or the block a helper joins (gpsbb_events.hip.h) */
#ifdef GPSBB_WG_TRACE
    wgt.block = b;
    wgt.nblocks = p.nblocks;
#endif
    if (b < 0) {
#ifdef GPSBB_WG_TRACE
        wg_trace_leave(wgt, wgt.wall0, wgt.clk0, 0u, 2u, false);
#endif
        return;
    }
    const gpsbb_chan_t *__restrict__ cb = p.ch + (size_t)b * p.nch;
    const EvConst *__restrict__ kb = p.evc + (size_t)b * p.nch;
    /* ---- stage the block's per-channel tables in LDS: wavefront w takes channels w, w + 16, ... ---- */
    for (int i = __builtin_amdgcn_readfirstlane(tid >> 6); i < p.nch; i += EV_WAVES) {
        const int prn = cb[i].prn;
        const bool down = kb[i].down != 0;
        const double g = cb[i].gain;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = (tid & 63) + 64 * j;
            const int k = down ? 511 - e : e;
            v2f v;
            v.x = v.y = 0.0f;
            if (prn > 0) {
                /* (int)(table * gain): one IEEE multiply, truncation toward zero (plutogpssim.c:2701-2702) */
                v.x = (float)(int)mul_rn((double)p.tabs[k], g);
                v.y = (float)(int)mul_rn((double)p.tabs[512 + k], g);
            }
            L.amp[i][e] = v;
        }
        const uint32_t my_word = p.ca_bits[(prn > 0 ? prn : 0) * 32 + (tid & 31)];
#pragma unroll 4
        for (int j = 0; j < (EV_CHIP_LEN + 63) / 64; j++) {
            const int c = (tid & 63) + 64 * j;
            const int ca = c >= GPSBB_CA_LEN ? c - GPSBB_CA_LEN : c; /* c < 2 * 1023 */
            const uint32_t w0 = (uint32_t)__shfl((int)my_word, (ca >> 5) & 31);
            const uint32_t bit = (w0 >> (ca & 31)) & 1u;
            if (c < EV_CHIP_LEN) {
                /* codeCA = chip * 2 - 1 (c:2737) as the upper half of a binary32; idle: 0.0 */
                L.chipf[0][i][c] = (uint16_t)(prn > 0 ? (bit ? 0x3f80u : 0xbf80u) : 0u);
                if (WIDE)
                    L.chipf[PdLds<WIDE>::NTAB - 1][i][c] = (uint16_t)(prn > 0 ? (bit ? 0xbf80u : 0x3f80u) : 0u);
            }
        }
    }
    __syncthreads();
#ifdef GPSBB_WG_TRACE
    const unsigned long long t_staged = wall_clock64(), c_staged = clock64();
#endif

    /* ---- from here on every wavefront works alone ---- */
    const int wave = tid >> 6, lane = tid & 63;
    const int ntw = p.ntiles;
    const int nch2 = 2 * p.nch;
    uint32_t act_mask, exact_mask;
    {
        const bool act = lane < p.nch && cb[lane < p.nch ? lane : 0].prn > 0;
        act_mask = (uint32_t)__ballot(act);
        exact_mask = (uint32_t)__ballot(act && kb[lane < p.nch ? lane : 0].kc != EV_KC_DENSE); /* (the host only sends all-dense batches) */
    }
    const bool chain_lane = lane < nch2;
    const bool mirror = chain_lane && (lane & 1) && kb[lane >> 1].down != 0;
    const double *__restrict__ txb = p.tile_x + (size_t)b * ntw * nch2;
    const double *__restrict__ tx = txb + (size_t)(chain_lane ? lane : 0) * ntw;
    const uint32_t *__restrict__ tn = p.tile_nav + ((size_t)b * p.nch + (lane < p.nch ? lane : 0)) * ntw;
    /* what turns a tile state into its model in guard format (see PdLds::tstate) */
    const double guard = 0x1p+20 + (double)PD_BAND * 0x1p-32;
    const double g_scale = (lane & 1) ? 8.0 : 2.0;
    const double g_add = guard + ((lane & 1) ? 0.0 : (double)lds_addr_of(&L.chipf[0][chain_lane ? lane >> 1 : 0][0]));
    constexpr uint32_t neg_table_bytes = (uint32_t)(sizeof(uint16_t) * PdLds<WIDE>::NCH * EV_CHIP_LEN); /* from chipf[0] to the negated table */
    const double neg_table = (double)neg_table_bytes;
    const double lf = (double)lane;
    /* fixed-point carrier (the reference without FLOAT_CARR_PHASE): the tile states are (phase mod 2^25) / 2^16, the steps
     * multiples of 2^-16, every sum exact; a falling phase is mirrored bit by bit: (2^25 - 1 - p) / 2^16 */
    const uint32_t fixed = p.kph0 != nullptr ? 1u : 0u;
    const double mirror_at = fixed ? 512.0 - 0x1p-16 : 512.0;
    unsigned long long *n_exact = p.hazards + 2;

    int base = 0;
    if (lane == 0)
        base = atomicAdd(&p.tile_ctr[b], p.ev_chunk);
    base = __builtin_amdgcn_readfirstlane(base);
    int pos = 0, buf = 0;
    int pending = 0;
    unsigned tiles_rendered = 0;
    double ts_v = 0.0;
    uint32_t nav_v = 0;
    if (base < ntw) {
        ts_v = chain_lane ? tx[base] : 0.0;
        nav_v = lane < p.nch ? tn[base] : 0u;
    }
    while (base < ntw) {
        const int wt = base + pos;
        if (chain_lane)
            L.tstate[wave][buf][lane] = __fma_rn(mirror ? mirror_at - ts_v : ts_v, g_scale, g_add);
        const double *ts = L.tstate[wave][buf];
        const uint32_t dbits = (uint32_t)__ballot(nav_v & 1u), dnext = (uint32_t)__ballot(nav_v & 2u);
        if (pos == 0 && lane == 0)
            pending = __hip_atomic_fetch_add(p.tile_ctr + b, p.ev_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last_of_chunk = pos + 1 >= p.ev_chunk || wt + 1 >= ntw;
        int next_base = base, next_pos = pos + 1;
        if (last_of_chunk) {
            next_base = __builtin_amdgcn_readfirstlane(pending);
            next_pos = 0;
        }
        const int wt_next = next_base + next_pos;
        if (wt_next < ntw) {
            ts_v = chain_lane ? tx[wt_next] : 0.0;
            nav_v = lane < p.nch ? tn[wt_next] : 0u;
        }

        /* the sums start at 1.5 * 2^23: every partial sum is then an integer below 2^24 in magnitude (exact in binary32) whose
         * low 16 bits ARE the int16 the reference's (short) cast keeps (c:2754-2755) */
        v2f acc[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++)
            acc[j].x = acc[j].y = PD_ACC0;
        uint32_t fixmask = 0u; /* channels in which some lane has to look again */
        for (uint32_t mk = act_mask; mk; mk &= mk - 1) {
            const int i = __builtin_ctz(mk);
            PdModel M;
            pd_model_of(L, kb, ts, i, dbits, dnext, M);
            uint32_t m;
            if (WIDE) {
                /* the table of the data bit in force at the tile start; past the roll-over, the other bit's */
                const double xtg = M.xtg + (M.neg ? neg_table : 0.0); /* exact: an integer number of bytes */
                const uint32_t roll = M.roll_addr + (M.neg ? neg_table_bytes : 0u);
                const int32_t delta = ((int32_t)M.neg_next - (int32_t)M.neg) * (int32_t)neg_table_bytes;
                m = pd_channel_fast_wide(M, xtg, lf, (M.neg ^ M.neg_next) | (fixed << 1), roll, delta, acc);
            } else {
                m = pd_channel_fast_narrow(M, lf, fixed, acc);
            }
            const unsigned long long um = __builtin_amdgcn_uicmp(m, p.pd_danger, 36 /* ult */);
            fixmask |= (um != 0ull || ((exact_mask >> i) & 1u)) ? 1u << i : 0u;
        }
        /* ---- rare: lanes that cannot rule out that the model and the reference disagree somewhere in their 16 samples of a
         * channel take the model's contributions out and put the exact ones in (after the channel loop: a second producer
         * of the accumulators inside it would cost register copies on every pass) ---- */
        if (__builtin_expect(fixmask != 0u, 0)) {
            GPSBB_EV_SETTLE_CLAIM(); /* the calls below make the compiler save registers: the chunk claim must have landed */
            for (uint32_t mk = fixmask; mk; mk &= mk - 1) {
                const int i = __builtin_ctz(mk);
                PdModel M;
                pd_model_of(L, kb, ts, i, dbits, dnext, M);
                const uint32_t bad = ((exact_mask >> i) & 1u) ? (1u << SPT) - 1u : pd_model_flagged(M, lane, fixed != 0u, p.pd_danger);
                if (bad) {
                    const size_t kk = (size_t)b * p.nch + i;
                    const int32_t fx_step = fixed ? p.kstep[kk] : 0;
                    const uint32_t fx_phase = fixed ? p.kph0[kk] + (uint32_t)wt * (uint32_t)TILE * (uint32_t)fx_step : 0u;
#pragma unroll 1
                    for (int j = 0; j < SPT; j++) {
                        if (!((bad >> j) & 1u))
                            continue;
                        const v2f t = pd_fix_sample(L, i, kb + i, txb + wt, ntw, M.ytg, M.xtg, M.amp_base, M.roll_addr, M.neg | (M.neg_next << 1), lane, j,
                                                    (int)fixed, fx_phase, fx_step);
#pragma unroll
                        for (int q = 0; q < SPT; q++) {
                            acc[q].x += q == j ? t.x : 0.0f;
                            acc[q].y += q == j ? t.y : 0.0f;
                        }
                    }
                    atomicAdd(n_exact, 1ull);
                }
            }
        }
        /* ---- the low halves of the two sums side by side, store (c:2754-2755): sample wt*TILE + j*64 + lane ---- */
        uint32_t *out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + (size_t)wt * TILE + lane;
        const int left = p.nsamp - wt * TILE - lane; /* samples j*64 < left exist */
        if (DIGEST) {
            /* the block's digest as it is rendered (see synth_ev_body): this lane's samples are wt*TILE + j*64 + lane */
            uint32_t m = digest_weight((uint32_t)(wt * TILE + lane));
            unsigned long long dg = 0ull;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                if (j * 64 < left)
                    dg += (unsigned long long)__builtin_amdgcn_perm(__float_as_uint(acc[j].y), __float_as_uint(acc[j].x), 0x05040100u) * m;
                m += 64u * DIGEST_STEP;
            }
#pragma unroll
            for (int off2 = 32; off2 > 0; off2 >>= 1)
                dg += (unsigned long long)__shfl_down((long long)dg, off2);
            if (lane == 0 && dg)
                atomicAdd(p.digest + b, dg);
        }
        if (__builtin_expect(p.nsamp - wt * TILE >= TILE, 1)) {
#pragma unroll
            for (int j = 0; j < SPT; j++)
                out[j * 64] = __builtin_amdgcn_perm(__float_as_uint(acc[j].y), __float_as_uint(acc[j].x), 0x05040100u);
        } else {
#pragma unroll
            for (int j = 0; j < SPT; j++)
                if (j * 64 < left)
                    out[j * 64] = __builtin_amdgcn_perm(__float_as_uint(acc[j].y), __float_as_uint(acc[j].x), 0x05040100u);
        }
        base = next_base;
        pos = next_pos;
        buf ^= 1;
        tiles_rendered++;
    }
    if (lane == 0 && tiles_rendered)
        atomicAdd(p.hazards + 7, (unsigned long long)tiles_rendered); /* see synth_ev_body */
#ifdef GPSBB_WG_TRACE
    wg_trace_leave(wgt, t_staged, c_staged, tiles_rendered, 2u, true);
#endif
}

} /* namespace gpsbb_impl */
#endif
