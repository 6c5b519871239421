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
 * at sample n is not x0 + n*s.  k_seed walks every chain exactly (gpsbb_nco.h) and leaves the exact state
 * at the first sample of every 1024-sample tile.  Inside a tile a lane uses the linear model
 *     state(n) ~= state(tile start) + (n - tile start) * step
 * which differs from the true sequence by the roundings of at most 1039 additions (each at most half an
 * ulp of a number below 512 resp. 1024: 2^-45 resp. 2^-44) plus the model's own arithmetic: less than
 * 2^-33.9 table-index units / chips in total (EV_MODEL_ERR below is 2^-32).  floor(model) equals
 * floor(truth) at every sample unless the model passes within that distance of an integer at a sample; a
 * lane tests exactly that for each of its breakpoints (and for its first sample) and, if it cannot rule
 * it out, recomputes its run exactly: jump-ahead from the tile's exact state with the genuine IEEE steps
 * (ev_exact_run).  At 25 MS/s that happens for about one lane-run-channel in 10^5, so the cost is nil,
 * and the output is bit-exact by construction, not by luck.  Wraps need no special case on the fast
 * path: the carrier's is index 511 -> 0 (the model runs on unwrapped, the index is taken modulo 512), the
 * code's is chip 1022 -> 0 with the next period's data bit.
 *
 * I and Q travel as one 32-bit integer P = Q*65536 + I: sums of such integers are exact modulo 2^32, the
 * low half is the I sum modulo 2^16 (what the reference's (short) cast keeps) and the high half is the Q
 * sum plus the borrow of the low half, which one add of 0x8000 undoes as long as |I sum| < 2^15.  The
 * host only selects this kernel when the gains guarantee that (sum of |gain| < 63); other batches, low
 * sample rates and the fixed-point carrier run on k_synth.
 */
#ifndef GPSBB_EVENTS_HIP_H
#define GPSBB_EVENTS_HIP_H

#include "gpsbb_kernels.hip.h"

namespace gpsbb_impl {

#ifndef GPSBB_EV_WG
#define GPSBB_EV_WG 1024
#endif
constexpr int EV_WG = GPSBB_EV_WG;
constexpr int EV_WAVES = EV_WG / 64;
constexpr int EV_KC_MAX = 4;                       /* carrier breakpoints a run may hold */
constexpr int EV_AMP_PAD = EV_KC_MAX;              /* table entries repeated before [0] and after [511] */
constexpr int EV_AMP_STRIDE = 512 + 2 * EV_AMP_PAD;
#ifndef GPSBB_EV_CHUNK
#define GPSBB_EV_CHUNK 2
#endif
constexpr int EV_CHUNK = GPSBB_EV_CHUNK;

/* bound used for |model - truth| (table-index units / chips); the derivation above gives < 2^-33.9 */
#define EV_MODEL_ERR 0x1p-32
/* a breakpoint estimate t = g * (1/|step|) carries a few roundings on top of the model error */
#define EV_T_EPS 0x1p-44

/* LDS image of one workgroup */
struct EvLds {
    uint32_t amp[GPSBB_MAX_CHAN][EV_AMP_STRIDE]; /* P = Q*65536 + I of table index k at [EV_AMP_PAD + k], k = -PAD .. 511+PAD (mod 512) */
    int8_t chipm[GPSBB_MAX_CHAN][1024];          /* 0 where codeCA = +1, -1 where codeCA = -1; [1023] = [0] */
    uint32_t D[EV_WAVES][16][64];                /* difference arrays: row j & 15 (row 0 = discard), lane */
    int32_t act[GPSBB_MAX_CHAN];
    int32_t nact;
};

/* (x ^ m) - m: x where m = 0, -x where m = -1 */
__device__ __forceinline__ uint32_t signed_by(uint32_t x, uint32_t m) { return (x ^ m) - m; }

/*
 * The exact recomputation of one lane's run of one channel (rare): advance both NCOs from the tile's exact
 * state by n_off genuine steps with the jump-ahead of gpsbb_nco.h, then walk the run sample by sample as the
 * reference does (c:2697-2746) and add the differences of its contributions.
 */
__device__ __noinline__ void ev_exact_run(EvLds &L, int wave, int lane, int i, const EvConst K, double xt, double yt,
                                          uint32_t nb, int n_off, uint32_t *acc0)
{
    /* code NCO: at most one roll-over between the tile start and the end of the run (checked by the host) */
    int64_t wraps = 0;
    double x = code_jump(xt, K.sc, (int64_t)n_off, &wraps);
    uint32_t dbm = (wraps > 0 ? (nb >> 1) & 1u : nb & 1u) ? 0xffffffffu : 0u;
    const uint32_t dbm_next = ((nb >> 1) & 1u) ? 0xffffffffu : 0u;
    /* carrier NCO in cycles (the tile state is stored scaled by 512, exactly) */
    const double s = K.S * (1.0 / 512.0);
    double cp = carr_jump(yt * (1.0 / 512.0), s, (int64_t)n_off);
    uint32_t prev = 0;
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        const int it = (int)(cp * 512.0) & 511; /* c:2697; carr_phase == 1.0: index 512 defined as 0 */
        const int ci = (int)x;                  /* c:2737 */
        const uint32_t m = (uint32_t)(int32_t)L.chipm[i][ci] ^ dbm;
        const uint32_t v = signed_by(L.amp[i][EV_AMP_PAD + it], m);
        if (j == 0)
            *acc0 += v;
        else
            __hip_atomic_fetch_add(&L.D[wave][j][lane], v - prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = v;
        if (code_step(x, K.sc)) /* c:2709-2734: the data bit of the next period */
            dbm = dbm_next;
        carr_step(cp, s); /* c:2741-2746 */
    }
}

/*
 * One channel's contribution to this lane's run.  DOWN: the carrier step is negative; KC: carrier
 * breakpoints a run can hold (wave-uniform, from the host).  Returns nothing: D / acc0 are updated.
 */
template <bool DOWN, int KC>
__device__ __forceinline__ void ev_channel(EvLds &L, int wave, int lane, int i, const EvConst K, double xt, double yt,
                                           uint32_t nb, double off, bool lane_live, uint32_t &acc0,
                                           unsigned long long *n_exact)
{
    const double one_minus_b = 1.0 - EV_MODEL_ERR;
    /* ---- carrier: table index of the first sample and the samples at which it changes ---- */
    const double y0 = __fma_rn(off, K.S, yt);
    const double yf = floor(y0);
    const double fr = y0 - yf; /* exact */
    const int it0 = (int)yf & 511;
    const double g = DOWN ? fr : 1.0 - fr; /* distance to the next index change */
    bool unsafe = g > one_minus_b;         /* the previous change lies within the model error of sample 0 */
    int jk[KC];
    {
        double t = g * K.rS;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const double tq = fmin(t, 15.5);
            const double fq = __builtin_amdgcn_fract(tq);
            unsafe |= fabs(fq - 0.5) > K.thrK;
            jk[k] = (int)tq + 1; /* 1..16 */
            t += K.rS;
        }
    }
    const uint32_t *ampi = &L.amp[i][EV_AMP_PAD + it0];
    uint32_t A[KC + 1];
#pragma unroll
    for (int k = 0; k <= KC; k++)
        A[k] = DOWN ? ampi[-k] : ampi[k];

    /* ---- code: chip of the first sample and the sample at which it changes ---- */
    double x0 = __fma_rn(off, K.sc, xt);
    const bool wb = x0 >= 1023.0; /* rolled over since the tile start */
    x0 = wb ? x0 - 1023.0 : x0;
    const double xf = floor(x0);
    const int c0 = (int)xf;
    const double gc = 1.0 - (x0 - xf);
    unsafe |= gc > one_minus_b;
    const double tc = fmin(gc * K.rsc, 15.5);
    unsafe |= fabs(__builtin_amdgcn_fract(tc) - 0.5) > K.thrC;
    int jc = (int)tc + 1;
    const int8_t *chp = &L.chipm[i][c0];
    const uint32_t ma = (uint32_t)(int32_t)chp[0], mb = (uint32_t)(int32_t)chp[1];
    const uint32_t db_cur = (nb & 1u) ? 0xffffffffu : 0u, db_next = (nb & 2u) ? 0xffffffffu : 0u;
    const uint32_t dbA = wb ? db_next : db_cur;
    const uint32_t dbB = c0 == 1022 ? db_next : dbA;
    const uint32_t m0 = ma ^ dbA, m1 = mb ^ dbB;
    jc = m0 == m1 ? 16 : jc; /* equal neighbours: nothing changes at the chip boundary */

    /* ---- rare: this lane cannot rule out that the model and the reference disagree ---- */
    unsafe = (unsafe || K.kc < 0) && lane_live;
    if (__builtin_expect(__ballot(unsafe) != 0ull, 0)) {
        if (unsafe) {
            /* its fast-path contribution becomes nothing (row 0 is the discard row) ... */
#pragma unroll
            for (int k = 0; k < KC; k++)
                jk[k] = 16;
            jc = 16;
            A[0] = 0;
            /* ... and the exact one takes its place */
            ev_exact_run(L, wave, lane, i, K, xt, yt, nb, (int)off, &acc0);
            atomicAdd(n_exact, 1ull);
        }
    }

    /* ---- the contribution at sample 0 and its changes ---- */
    acc0 += signed_by(A[0], m0);
    uint32_t Ax = A[KC];
#pragma unroll
    for (int k = KC - 1; k >= 0; k--) {
        const bool before = jk[k] < jc; /* the index change comes before the chip change */
        const uint32_t mk = before ? m0 : m1;
        const uint32_t dk = signed_by(A[k + 1] - A[k], mk);
        __hip_atomic_fetch_add(&L.D[wave][jk[k] & 15][lane], dk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        Ax = before ? Ax : A[k]; /* amplitude in force just before the chip change */
    }
    /* the chip change flips the sign: -s0*A -> s1*A = 2*s1*A more */
    __hip_atomic_fetch_add(&L.D[wave][jc & 15][lane], signed_by(Ax << 1, m1), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(EV_WG) void k_synth_ev(BatchDev p, int16_t *__restrict__ iq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EvLds &L = *reinterpret_cast<EvLds *>(smem_raw);

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const gpsbb_chan_t *__restrict__ cb = p.ch + (size_t)b * p.nch;
    /* ---- stage the block's per-channel tables in LDS (once per workgroup) ---- */
    if (tid == 0) {
        int na = 0;
        for (int i = 0; i < p.nch; i++)
            if (cb[i].prn > 0)
                L.act[na++] = i;
        L.nact = na;
    }
    for (int e = tid; e < p.nch * EV_AMP_STRIDE; e += EV_WG) {
        const int i = e / EV_AMP_STRIDE, k = (e % EV_AMP_STRIDE - EV_AMP_PAD) & 511;
        uint32_t v = 0;
        if (cb[i].prn > 0) {
            const double g = cb[i].gain;
            /* (int)(table * gain): one IEEE multiply, truncation toward zero (plutogpssim.c:2701-2702) */
            const int ip = (int)mul_rn((double)p.tabs[k], g);
            const int qp = (int)mul_rn((double)p.tabs[512 + k], g);
            v = ((uint32_t)qp << 16) + (uint32_t)ip;
        }
        L.amp[i][e % EV_AMP_STRIDE] = v;
    }
    for (int e = tid; e < p.nch * 256; e += EV_WG) { /* four chips per lane */
        const int i = e >> 8, q = e & 255;
        const int prn = cb[i].prn;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = (4 * q + k) % GPSBB_CA_LEN; /* [1023] = [0] */
            const uint32_t bit = prn > 0 ? (p.ca_bits[prn * 32 + (c >> 5)] >> (c & 31)) & 1u : 1u;
            v |= (bit ? 0x00u : 0xffu) << (8 * k);
        }
        reinterpret_cast<uint32_t *>(&L.chipm[i][0])[q] = v;
    }
    for (int e = tid; e < EV_WAVES * 16 * 64; e += EV_WG)
        (&L.D[0][0][0])[e] = 0u;
    __syncthreads();
    const int nact = L.nact;

    /* ---- from here on every wavefront works alone ---- */
    const int wave = tid >> 6, lane = tid & 63;
    const int ntw = p.ntiles;
    const int nch2 = 2 * p.nch;
    const EvConst *__restrict__ kb = p.evc + (size_t)b * p.nch;
    const double *__restrict__ tx = p.tile_x + (size_t)b * ntw * nch2;
    const uint32_t *__restrict__ tn = p.tile_nav + (size_t)b * ntw * p.nch;
    const double off = (double)(lane * SPT);

    for (;;) {
        int chunk = 0;
        if (lane == 0)
            chunk = atomicAdd(&p.tile_ctr[b], EV_CHUNK);
        const int wt_begin = __builtin_amdgcn_readfirstlane(chunk);
        if (wt_begin >= ntw)
            break;
        const int wt_end = wt_begin + EV_CHUNK < ntw ? wt_begin + EV_CHUNK : ntw;
        /* exact states of all chains at the start of the tile: lane c holds chain c (2*channel + kind) */
        double ts = lane < nch2 ? tx[(size_t)wt_begin * nch2 + lane] : 0.0;
        uint32_t tnav = lane < p.nch ? tn[(size_t)wt_begin * p.nch + lane] : 0u;
        for (int wt = wt_begin; wt < wt_end; wt++) {
            double ts_next = 0.0;
            uint32_t tnav_next = 0u;
            if (wt + 1 < wt_end) {
                ts_next = lane < nch2 ? tx[(size_t)(wt + 1) * nch2 + lane] : 0.0;
                tnav_next = lane < p.nch ? tn[(size_t)(wt + 1) * p.nch + lane] : 0u;
            }
            const int n0 = wt * TILE + lane * SPT;
            const int nvalid = p.nsamp - n0 < SPT ? p.nsamp - n0 : SPT;
            const bool lane_live = nvalid > 0;
            uint32_t acc0 = 0;
            for (int a = 0; a < nact; a++) {
                const int i = __builtin_amdgcn_readfirstlane(L.act[a]);
                const EvConst K = kb[i];
                const double xt = hi_lo_f64(__builtin_amdgcn_readlane(__double2hiint(ts), 2 * i),
                                            __builtin_amdgcn_readlane(__double2loint(ts), 2 * i));
                const double yt = hi_lo_f64(__builtin_amdgcn_readlane(__double2hiint(ts), 2 * i + 1),
                                            __builtin_amdgcn_readlane(__double2loint(ts), 2 * i + 1));
                const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)tnav, i);
                const int kc = K.kc < 1 ? 1 : K.kc;
#define GPSBB_EV_CASE(DOWN, KC) ev_channel<DOWN, KC>(L, wave, lane, i, K, xt, yt, nb, off, lane_live, acc0, p.hazards + 2)
                if (K.down) {
                    switch (kc) {
                    case 1: GPSBB_EV_CASE(true, 1); break;
                    case 2: GPSBB_EV_CASE(true, 2); break;
                    case 3: GPSBB_EV_CASE(true, 3); break;
                    default: GPSBB_EV_CASE(true, 4); break;
                    }
                } else {
                    switch (kc) {
                    case 1: GPSBB_EV_CASE(false, 1); break;
                    case 2: GPSBB_EV_CASE(false, 2); break;
                    case 3: GPSBB_EV_CASE(false, 3); break;
                    default: GPSBB_EV_CASE(false, 4); break;
                    }
                }
#undef GPSBB_EV_CASE
            }
            /* ---- prefix sum over the run, back to int16 pairs, store (c:2754-2755) ---- */
            uint32_t o[SPT];
            uint32_t P = acc0;
            o[0] = ((P + 0x8000u) & 0xffff0000u) | (P & 0xffffu);
#pragma unroll
            for (int j = 1; j < SPT; j++) {
                P += __hip_atomic_exchange(&L.D[wave][j][lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                o[j] = ((P + 0x8000u) & 0xffff0000u) | (P & 0xffffu);
            }
            uint32_t *out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + n0;
            if (nvalid == SPT && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
                uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
                for (int j = 0; j < SPT; j += 4)
                    o4[j >> 2] = make_uint4(o[j], o[j + 1], o[j + 2], o[j + 3]);
            } else {
#pragma unroll
                for (int j = 0; j < SPT; j++)
                    if (j < nvalid)
                        out[j] = o[j];
            }
            ts = ts_next;
            tnav = tnav_next;
        }
    }
}

} /* namespace gpsbb_impl */
#endif
