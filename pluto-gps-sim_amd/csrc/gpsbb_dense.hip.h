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
 *       entry: one v_and per lookup, and the test "did the model come within its error of an integer" is the minimum of
 *       the low words over the run (v_min3_u32, one per channel-sample for both NCOs), compared once.
 *   amplitudes as float pairs, chips as +-1.0: a channel-sample's contribution is ONE packed FMA (I and Q together);
 *       sums of at most 16 integers below 2^15 are exact in binary32.  No sign extraction, no xor / sub / add chain.
 *
 * Per channel-sample: 2 v_fma_f64, 2 v_and (+1 or), 1 v_min3_u32, 1 shift, 1 v_pk_fma_f32, 2 conflict-free LDS reads —
 * about 31 issue cycles against the 58 of ev_dense (tools/ubench/valu_rates.hip).
 */
#ifndef GPSBB_DENSE_HIP_H
#define GPSBB_DENSE_HIP_H

#include <type_traits>

#include "gpsbb_events.hip.h"

namespace gpsbb_impl {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr uint32_t PD_BAND = 4; /* |model - truth| in units of 2^-32 of the scaled models, roundings of the guard format included:
                                   carrier 8 * 2^-33.9 * 2^32 = 2.2 + 1.5, code (4 or 2) * 0.27 + 1.5 */

/* WIDE: up to PD_WIDE_CHAN channels (the reference's MAX_CHAN, h:21): the chips as whole binary32 +-1.0 (no shift per
 * channel-sample); else up to GPSBB_MAX_CHAN with the chips as the upper halves (160 KB of LDS do not hold 16 wide tables) */
constexpr int PD_WIDE_CHAN = 12;
template <bool WIDE>
struct PdLds {
    static constexpr int NCH = WIDE ? PD_WIDE_CHAN : GPSBB_MAX_CHAN;
    typedef typename std::conditional<WIDE, uint32_t, uint16_t>::type chip_t;
    v2f amp[NCH][512];               /* ((float)(int)(cos*gain), (float)(int)(sin*gain)) of table index k (a falling
                                        carrier: of 511 - k, see ev_first) */
    chip_t chipf[NCH][EV_CHIP_LEN];  /* binary32 +1.0 / -1.0 (its upper half): codeCA of chip c mod 1023 */
    double tstate[EV_WAVES][2][2 * GPSBB_MAX_CHAN]; /* per wavefront, two tiles deep: the tile's models at sample 0 in guard
                                        format: column 2*channel = 2^20 + band + address of chipf[channel] + sizeof(chip_t) *
                                        code phase, 2*channel + 1 = 2^20 + band + 8 * carrier phase (mirrored) */
};

__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
template <class T>
__device__ __forceinline__ T lds_read_at(uint32_t a)
{
    return *(__attribute__((address_space(3))) const T *)(uintptr_t)a;
}

/*
 * Rare: a lane cannot rule out that the model and the reference disagree somewhere in its 16 samples of channel i.
 * It takes back what it added for the channel (the fast path run again with the sign turned) and takes the true
 * contributions instead, one sample at a time: from the tile's exact state to sample n with the exact jump-ahead
 * (gpsbb_nco.h), then index, chip and data bit as the reference has them there (c:2697-2737).  Out of line, and with
 * nothing but scalars in and out, so that the accumulators of the fast path stay in registers.
 */
template <bool WIDE>
__device__ __noinline__ v2f pd_exact_sample(const PdLds<WIDE> &L, int i, const EvConst *kbi, const double *tile_x, int ntiles, uint32_t dbits_i,
                                            uint32_t dnext_i, int n)
{
    const bool down = kbi->down != 0;
    const double S = down ? -kbi->S : kbi->S, sc = kbi->sc;
    const double xt = tile_x[(size_t)(2 * i) * ntiles], yt = tile_x[(size_t)(2 * i + 1) * ntiles];
    int64_t wraps = 0;
    const double x = code_jump(xt, sc, (int64_t)n, &wraps);
    const bool neg = wraps > 0 ? dnext_i != 0 : dbits_i != 0; /* at most one roll-over per tile (checked by the host) */
    const double cp = carr_jump(yt * (1.0 / 512.0), S * (1.0 / 512.0), (int64_t)n);
    const int it = (int)(cp * 512.0) & 511; /* c:2697; carr_phase == 1.0: index 512 defined as 0 */
    const int ci = (int)x;                  /* c:2737 */
    const float sg = __uint_as_float(((uint32_t)L.chipf[i][ci] << (WIDE ? 0 : 16)) ^ (neg ? 0x80000000u : 0u));
    const v2f a = L.amp[i][down ? 511 - it : it];
    v2f t;
    t.x = sg * a.x;
    t.y = sg * a.y;
    return t;
}

/* one channel of one tile on the fast path: SPT samples per lane, 64 apart.  NEG: the data bit in force is -1; DF: it
 * changes inside the tile (samples past the code's roll-over, chip index >= 1023, take the other one) */
template <bool WIDE, bool NEG, bool DF, bool UNDO>
__device__ __forceinline__ uint32_t pd_channel(const PdLds<WIDE> &L, int lane, uint32_t amp_base, double S8, double sc2, double ytg, double xtg,
                                               uint32_t roll_addr, v2f (&acc)[SPT])
{
    const double lf = (double)lane;
    const double y0 = __fma_rn(lf, S8, ytg), x0 = __fma_rn(lf, sc2, xtg);
    const double dy = S8 * 64.0, dx = sc2 * 64.0; /* exact */
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        const double yj = j ? __fma_rn((double)j, dy, y0) : y0, xj = j ? __fma_rn((double)j, dx, x0) : x0;
        const uint32_t ylo = (uint32_t)__double2loint(yj), xlo = (uint32_t)__double2loint(xj);
        const uint32_t ia = ((uint32_t)__double2hiint(yj) & 0xff8u) | amp_base; /* 8 bytes per table entry, index modulo 512 */
        const uint32_t ic = (uint32_t)__double2hiint(xj) & (WIDE ? 0xffffcu : 0xffffeu); /* 4 / 2 bytes per chip, the table's address included */
        /* both fractions stay PD_BAND units away from an integer (the models carry +PD_BAND: safe iff low word >= 2*PD_BAND);
         * the low word of the carrier model misses its top three bits (the model is scaled by 8): a conservative test */
        m = min(m, min(ylo, xlo));
        const v2f a = lds_read_at<v2f>(ia);
        uint32_t sgb = WIDE ? lds_read_at<uint32_t>(ic) : (uint32_t)lds_read_at<uint16_t>(ic) << 16;
        if (DF)
            sgb ^= (ic >= roll_addr) != NEG ? 0x80000000u : 0u; /* NEG here: the data bit BEFORE the roll-over is -1; after it, the other */
        const float sg = __uint_as_float(sgb);
        v2f sv;
        sv.x = (NEG && !DF) != UNDO ? -sg : sg;
        sv.y = sv.x;
        acc[j] = __builtin_elementwise_fma(sv, a, acc[j]);
    }
    return m;
}

template <bool WIDE>
__global__ __launch_bounds__(EV_WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_synth_pd(BatchDev p, int16_t *__restrict__ iq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PdLds<WIDE> &L = *reinterpret_cast<PdLds<WIDE> *>(smem_raw);
    const int tid = threadIdx.x;
    const int b = blockIdx.x; /* the block is the fast grid dimension, helpers join blocks still in flight (see k_synth_ev) */
    if (blockIdx.y > 0 && __hip_atomic_load(&p.tile_ctr[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.ntiles)
        return;
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
            if (c < EV_CHIP_LEN)
                L.chipf[i][c] = (typename PdLds<WIDE>::chip_t)((prn > 0 ? (bit ? 0x3f800000u : 0xbf800000u) : 0u) >> (WIDE ? 0 : 16)); /* codeCA = chip * 2 - 1 (c:2737); idle: 0.0 */
        }
    }
    __syncthreads();

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
    const double g_scale = (lane & 1) ? 8.0 : (WIDE ? 4.0 : 2.0);
    const double g_add = guard + ((lane & 1) ? 0.0 : (double)lds_addr_of(&L.chipf[chain_lane ? lane >> 1 : 0][0]));
    unsigned long long *n_exact = p.hazards + 2;

    int base = 0;
    if (lane == 0)
        base = atomicAdd(&p.tile_ctr[b], p.ev_chunk);
    base = __builtin_amdgcn_readfirstlane(base);
    int pos = 0, buf = 0;
    int pending = 0;
    double ts_v = 0.0;
    uint32_t nav_v = 0;
    if (base < ntw) {
        ts_v = chain_lane ? tx[base] : 0.0;
        nav_v = lane < p.nch ? tn[base] : 0u;
    }
    while (base < ntw) {
        const int wt = base + pos;
        if (chain_lane)
            L.tstate[wave][buf][lane] = __fma_rn(mirror ? 512.0 - ts_v : ts_v, g_scale, g_add);
        const double *ts = L.tstate[wave][buf];
        const uint32_t dbits = (uint32_t)__ballot(nav_v & 1u), dnext = (uint32_t)__ballot(nav_v & 2u);
        const uint32_t dflip = dbits ^ dnext;
        if (pos == 0 && lane == 0)
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(pending) : "v"(p.tile_ctr + b), "v"(p.ev_chunk) : "memory");
        const bool last_of_chunk = pos + 1 >= p.ev_chunk || wt + 1 >= ntw;
        int next_base = base, next_pos = pos + 1;
        if (last_of_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pending)::"memory");
            next_base = __builtin_amdgcn_readfirstlane(pending);
            next_pos = 0;
        }
        const int wt_next = next_base + next_pos;
        if (wt_next < ntw) {
            ts_v = chain_lane ? tx[wt_next] : 0.0;
            nav_v = lane < p.nch ? tn[wt_next] : 0u;
        }

        v2f acc[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++)
            acc[j].x = acc[j].y = 0.0f;
        for (uint32_t mk = act_mask; mk; mk &= mk - 1) {
            const int i = __builtin_ctz(mk);
            const double S8 = scalar_load(&kb[i].S) * 8.0, sc2 = scalar_load(&kb[i].sc) * (WIDE ? 4.0 : 2.0);
            const double xtg = ts[2 * i], ytg = ts[2 * i + 1];
            const uint32_t amp_base = lds_addr_of(&L.amp[i][0]);
            const uint32_t roll_addr = lds_addr_of(&L.chipf[i][GPSBB_CA_LEN]);
            const bool neg = (dbits >> i) & 1u, df = (dflip >> i) & 1u;
            uint32_t m;
            if (__builtin_expect(df, 0)) {
                m = neg ? pd_channel<WIDE, true, true, false>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc)
                        : pd_channel<WIDE, false, true, false>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
            } else {
                m = neg ? pd_channel<WIDE, true, false, false>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc)
                        : pd_channel<WIDE, false, false, false>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
            }
            const unsigned long long um = ((exact_mask >> i) & 1u) ? ~0ull : __builtin_amdgcn_uicmp(m, 2u * PD_BAND, 36 /* ult */);
            if (__builtin_expect(um != 0ull, 0)) {
                if ((um >> lane) & 1ull) {
                    /* take the model's contributions back, put the exact ones in their place */
                    if (df) {
                        if (neg)
                            (void)pd_channel<WIDE, true, true, true>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
                        else
                            (void)pd_channel<WIDE, false, true, true>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
                    } else {
                        if (neg)
                            (void)pd_channel<WIDE, true, false, true>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
                        else
                            (void)pd_channel<WIDE, false, false, true>(L, lane, amp_base, S8, sc2, ytg, xtg, roll_addr, acc);
                    }
#pragma unroll 1
                    for (int j = 0; j < SPT; j++) {
                        const v2f t = pd_exact_sample(L, i, kb + i, txb + wt, ntw, (dbits >> i) & 1u, (dnext >> i) & 1u, j * 64 + lane);
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
        /* ---- back to int16 pairs, store (c:2754-2755): sample wt*TILE + j*64 + lane ---- */
        uint32_t *out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + (size_t)wt * TILE + lane;
        const int left = p.nsamp - wt * TILE - lane; /* samples j*64 < left exist */
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int ii = (int)acc[j].x, qq = (int)acc[j].y; /* exact: sums of integers below 2^15 */
            uint32_t o;
            asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(o) : "v"(ii), "v"(qq));
            if (j * 64 < left)
                out[j * 64] = o;
        }
        base = next_base;
        pos = next_pos;
        buf ^= 1;
    }
}

} /* namespace gpsbb_impl */
#endif
