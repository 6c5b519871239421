/*
 * gpsbb_modelerr.hip.h — experiments build only (libgpsbb_exp.so): MEASURES the error budgets the model kernels'
 * bit-exactness rests on, instead of summing them by hand.
 *
 * k_synth_ev / k_synth_ev_dense / k_synth_pd evaluate both NCOs of the sample loop (plutogpssim.c:2697, 2709-2746) on a
 * linear model inside a 1024-sample tile and trust floor(model) wherever the model, in "guard format" with the channel's
 * bias W, stays 2W clear of an integer; everything else is recomputed with genuine IEEE steps.  That is exact by
 * construction IF |model - truth| <= W wherever the model is trusted.  W is derived (EV_MODEL_ERR, EV_T_EPS, PD_BAND:
 * gpsbb_events.hip.h, gpsbb_dense.hip.h, ev_plan in gpsbb.hip); this kernel replays, for every (block, channel, tile) of
 * a batch that has just run, the very arithmetic of the fast paths (same operations in the same order on the same
 * tile states and per-channel constants) next to THE REFERENCE'S OWN RECURRENCE — both NCOs stepped sample by sample
 * from the tile's exact state with code_step / carr_step (c:2709-2712, 2741-2746) — and reports
 *   - the realised |biased quantity - W - truth| of everything the kernels test: first-sample index and chip models,
 *     positions of index and chip changes (k_synth_ev), the per-sample models (ev_dense, k_synth_pd), in units of 2^-32
 *     and as a fraction of the channel's W (the budget);
 *   - the realised error of the plain linear model state(tile) + n*step against the truth (EV_MODEL_ERR's 2^-32);
 *   - how many lane-runs the danger test sent to the exact path (must equal the kernel's own GPSBB_INFO_EXACT_RUNS count
 *     for the same launch: the cross-check that this replay computes what the kernel computes);
 *   - how many decisions (table index, chip, the sample a change shows at) of lanes the danger test did NOT flag differ
 *     from the truth: must be zero.
 * One thread per (block, channel, tile), stepping the 1024 samples of the tile in order.
 */
#ifndef GPSBB_MODELERR_HIP_H
#define GPSBB_MODELERR_HIP_H

#include "gpsbb_dense.hip.h"

namespace gpsbb_impl {

/* maxima per channel index (doubles, >= 0) */
enum {
    ME_Y0 = 0,  /* carrier model at a tested sample (first sample of a run; every sample for ev_dense / k_synth_pd): units of 2^-32 */
    ME_Y0_W,    /* ... as a fraction of the budget (W resp. PD_BAND) */
    ME_TK,      /* position of an index change (k_synth_ev): units of 2^-32 samples */
    ME_TK_W,
    ME_X0,      /* code model at a tested sample */
    ME_X0_W,
    ME_TC,      /* position of the chip change */
    ME_TC_W,
    ME_PURE_Y,  /* |tile state + n*S - truth| in table-index units * 2^32 (EV_MODEL_ERR = 1; derived: 2^-33.9 = 0.27) */
    ME_PURE_X,  /* ... chips */
    ME_W_UNITS, /* the largest W seen, in units of 2^-32 */
    ME_NQ
};
static_assert(ME_NQ == 11, "GPSBB_TEST_ME_NQ");
/* counts per channel index */
enum { MEC_BAD = 0, MEC_DANGER, MEC_LANES, MEC_ALWAYS, MEC_NQ };

__device__ __forceinline__ void me_max(double *slot, double v)
{
    atomicMax(reinterpret_cast<unsigned long long *>(slot), (unsigned long long)__double_as_longlong(v));
}

/* d modulo `period`, centred */
__device__ __forceinline__ double me_centre(double d, double period) { return d - period * rint(d / period); }

struct MeAcc {
    double mx[ME_NQ];
    unsigned long long bad, danger, lanes, always;
    __device__ void up(int q, double v) { mx[q] = v > mx[q] ? v : mx[q]; }
};

/* the truth: both NCOs of one channel stepped as the reference steps them */
struct MeTruth {
    double cp, s;   /* carrier phase in cycles, its step (signed) */
    double x, sc;   /* code phase in chips, its step */
    int laps, wraps;
    bool down;
    bool fixed;      /* GPSBB_FIXED_CARRIER: the carrier is the 32-bit accumulator (c:2699, 2748) */
    uint32_t ph, st;
    __device__ void step()
    {
        if (code_step(x, sc))
            wraps++;
        if (carr_step(cp, s))
            laps++;
        ph += st;
    }
    __device__ int it() const { return fixed ? (int)((ph >> 16) & 0x1ffu) : ((int)(cp * 512.0) & 511); } /* c:2699 / c:2697 (index 512 defined as 0) */
    __device__ int ci() const { return (int)x + GPSBB_CA_LEN * wraps; } /* c:2737, not reduced: the chip tables go on past 1023 */
    /* the carrier in the kernels' rising, unwrapped, mirrored coordinate (table-index units) */
    __device__ double u() const { return (down ? 512.0 - cp * 512.0 : cp * 512.0) + 512.0 * (double)laps; }
    __device__ double ux() const { return x + 1023.0 * (double)wraps; }
};

/* ---- k_synth_ev / k_synth_ev_dense ---------------------------------------------------------------------------- */
template <int KC, bool FIXED>
__device__ void me_ev_run(const EvConst &kb, MeTruth &T, double ytg, double xtg, double ts_y_m, double ts_x, int lane, int nvalid,
                          MeAcc &A)
{
    const double sat = 15.5 + EV_GUARD;
    const double off = (double)(lane * SPT);
    const double Wu = kb.W * 0x1p+32;
    /* ---- exactly ev_first<KC, FIXED> ---- */
    const double y0 = __fma_rn(off, kb.S, ytg);
    const double fr = __builtin_amdgcn_fract(y0);
    const int it0 = __double2hiint(y0) & 511;
    uint32_t m = FIXED ? 0xffffffffu : (uint32_t)__double2loint(y0);
    double t = __fma_rn(-fr, kb.rS, kb.tK0);
    int jk[KC];
    double tqv[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const double tq = fmin(t, sat);
        if (!FIXED)
            m = min(m, (uint32_t)__double2loint(tq));
        jk[k] = __double2hiint(tq) & 15;
        tqv[k] = tq;
        t += kb.rS;
    }
    const double x0 = __fma_rn(off, kb.sc, xtg);
    const double frc = __builtin_amdgcn_fract(x0);
    const int c0 = __double2hiint(x0) & 2047;
    const double tc = fmin(__fma_rn(-frc, kb.rsc, kb.tC0), sat);
    m = min(m, min((uint32_t)__double2loint(x0), (uint32_t)__double2loint(tc)));
    const int jc = __double2hiint(tc) & 15;
    const bool dang = m < kb.danger;
    /* ---- against the truth ---- */
    bool bad = false;
    uint32_t pred_k = 0, pred_c = 0, true_k = 0, true_c = 0;
#pragma unroll
    for (int k = 0; k < KC; k++)
        if (jk[k] != EV_ROW_DISCARD)
            pred_k |= 1u << (jk[k] + 1);
    if (jc != EV_ROW_DISCARD)
        pred_c |= 1u << (jc + 1);
    {
        /* first sample: index and chip, and the two models there */
        const int itk = T.down ? 511 - it0 : it0;
        bad = bad || itk != T.it() || c0 != T.ci();
        const double ex = fabs(((x0 - EV_GUARD) - T.ux()) - kb.W) * 0x1p+32;
        A.up(ME_X0, ex);
        A.up(ME_X0_W, ex / Wu);
        const double n = off;
        if (!FIXED) { /* (the accumulator's model is exact: nothing to measure on the carrier side) */
            const double ey = fabs(me_centre((y0 - EV_GUARD) - T.u(), 512.0) - kb.W) * 0x1p+32;
            A.up(ME_Y0, ey);
            A.up(ME_Y0_W, ey / Wu);
            A.up(ME_PURE_Y, fabs(me_centre(__fma_rn(n, kb.S, ts_y_m) - T.u(), 512.0)) * 0x1p+32);
        }
        A.up(ME_PURE_X, fabs(__fma_rn(n, kb.sc, ts_x) - T.ux()) * 0x1p+32);
    }
    int kth = 0;
    int it_prev = T.it(), ci_prev = T.ci();
    double u_prev = T.u(), ux_prev = T.ux();
    for (int j = 1; j < SPT; j++) {
        T.step();
        const int it = T.it(), ci = T.ci();
        const double u = T.u(), ux = T.ux();
        if (j < nvalid) {
            const double n = off + (double)j;
            if (!FIXED)
                A.up(ME_PURE_Y, fabs(me_centre(__fma_rn(n, kb.S, ts_y_m) - u, 512.0)) * 0x1p+32);
            A.up(ME_PURE_X, fabs(__fma_rn(n, kb.sc, ts_x) - ux) * 0x1p+32);
            if (it != it_prev) {
                true_k |= 1u << j;
                /* where between samples j-1 and j the (mirrored, rising) truth passes the integer: the continuous position the
                 * model's t estimates */
                if (!FIXED && !dang && kth < KC && tqv[kth] < sat && u > u_prev) { /* (a flagged lane does not use its positions) */
                    const double frac = (floor(u) - u_prev) / (u - u_prev);
                    const double e = fabs(((tqv[kth] - EV_GUARD) - (double)(j - 1) - frac) - kb.W) * 0x1p+32;
                    A.up(ME_TK, e);
                    A.up(ME_TK_W, e / Wu);
                }
                kth++;
            }
            if (ci != ci_prev) {
                true_c |= 1u << j;
                if (!dang && tc < sat && ux > ux_prev) {
                    const double frac = (floor(ux) - ux_prev) / (ux - ux_prev);
                    const double e = fabs(((tc - EV_GUARD) - (double)(j - 1) - frac) - kb.W) * 0x1p+32;
                    A.up(ME_TC, e);
                    A.up(ME_TC_W, e / Wu);
                }
            }
        }
        it_prev = it;
        ci_prev = ci;
        u_prev = u;
        ux_prev = ux;
    }
    T.step(); /* on to the next run's first sample */
    const uint32_t valid = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    bad = bad || ((pred_k ^ true_k) & valid) != 0u || ((pred_c ^ true_c) & valid) != 0u;
    A.lanes++;
    if (dang)
        A.danger++;
    else if (bad)
        A.bad++;
}

/* a channel the mixed kernel evaluates per sample (ev_dense) */
__device__ void me_ev_dense_run(const EvConst &kb, MeTruth &T, double ytg, double xtg, double ts_y_m, double ts_x, int lane, int nvalid,
                                MeAcc &A)
{
    const double off = (double)(lane * SPT);
    const double Wu = kb.W * 0x1p+32;
    const double y0 = __fma_rn(off, kb.S, ytg), x0 = __fma_rn(off, kb.sc, xtg);
    bool dang = false, bad = false;
    for (int j = 0; j < SPT; j++) {
        const double yj = __fma_rn((double)j, kb.S, y0), xj = __fma_rn((double)j, kb.sc, x0);
        const uint32_t ylo = (uint32_t)__double2loint(yj), xlo = (uint32_t)__double2loint(xj);
        dang = dang || min(ylo, xlo) < kb.danger;
        if (j < nvalid) {
            const int it = __double2hiint(yj) & 511, ci = __double2hiint(xj) & 2047;
            bad = bad || (T.down ? 511 - it : it) != T.it() || ci != T.ci();
            const double ey = fabs(me_centre((yj - EV_GUARD) - T.u(), 512.0) - kb.W) * 0x1p+32;
            const double ex = fabs(((xj - EV_GUARD) - T.ux()) - kb.W) * 0x1p+32;
            A.up(ME_Y0, ey);
            A.up(ME_Y0_W, ey / Wu);
            A.up(ME_X0, ex);
            A.up(ME_X0_W, ex / Wu);
            const double n = off + (double)j;
            A.up(ME_PURE_Y, fabs(me_centre(__fma_rn(n, kb.S, ts_y_m) - T.u(), 512.0)) * 0x1p+32);
            A.up(ME_PURE_X, fabs(__fma_rn(n, kb.sc, ts_x) - T.ux()) * 0x1p+32);
        }
        T.step();
    }
    A.lanes++;
    if (dang)
        A.danger++;
    else if (bad)
        A.bad++;
}

__global__ __launch_bounds__(256) void k_model_err_ev(BatchDev p, double *mx, unsigned long long *cnt)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wt = (int)(gid % (size_t)p.ntiles);
    const int i = (int)((gid / (size_t)p.ntiles) % (size_t)p.nch);
    const size_t b = gid / ((size_t)p.ntiles * p.nch);
    if (b >= (size_t)p.nblocks)
        return;
    if (p.ch[b * p.nch + i].prn <= 0)
        return;
    const EvConst kb = p.evc[b * p.nch + i];
    const double *txb = p.tile_x + b * (size_t)p.ntiles * 2 * p.nch;
    const double ts_x = txb[(size_t)(2 * i) * p.ntiles + wt], ts_y = txb[(size_t)(2 * i + 1) * p.ntiles + wt];
    const bool down = kb.down != 0;
    /* exactly what synth_ev_body puts into EvLds::tstate (IEEE carrier) */
    const bool fixed = p.kph0 != nullptr; /* k_synth_ev_fixed: the carrier lanes carry no bias, a falling phase is mirrored bit by bit */
    const double guard_w = EV_GUARD + kb.W;
    const double ts_y_m = down ? (fixed ? 512.0 - 0x1p-16 : 512.0) - ts_y : ts_y;
    const double ytg = ts_y_m + (fixed ? EV_GUARD : guard_w), xtg = ts_x + guard_w;
    MeTruth T;
    T.cp = fixed ? 0.0 : ts_y * (1.0 / 512.0);
    T.s = fixed ? 0.0 : (down ? -kb.S : kb.S) * (1.0 / 512.0);
    T.x = ts_x;
    T.sc = kb.sc;
    T.laps = T.wraps = 0;
    T.down = down;
    T.fixed = fixed;
    T.st = fixed ? (uint32_t)p.kstep[b * p.nch + i] : 0u;
    T.ph = fixed ? p.kph0[b * p.nch + i] + (uint32_t)wt * (uint32_t)TILE * T.st : 0u;
    MeAcc A;
    for (int q = 0; q < ME_NQ; q++)
        A.mx[q] = 0.0;
    A.bad = A.danger = A.lanes = A.always = 0;
    A.up(ME_W_UNITS, kb.W * 0x1p+32);
    for (int lane = 0; lane < 64; lane++) {
        const int n0 = wt * TILE + lane * SPT;
        const int nvalid = p.nsamp - n0 < SPT ? p.nsamp - n0 : SPT;
        if (nvalid <= 0)
            break;
        if (kb.kc < 0) { /* always recomputed exactly */
            A.always++;
            for (int j = 0; j < SPT; j++)
                T.step();
            continue;
        }
        if (fixed) {
            switch (kb.kc) {
            case 2: me_ev_run<2, true>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
            case 3: me_ev_run<3, true>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
            case 4: me_ev_run<4, true>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
            default: me_ev_run<1, true>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
            }
            continue;
        }
        switch (kb.kc) {
        case 2: me_ev_run<2, false>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
        case 3: me_ev_run<3, false>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
        case 4: me_ev_run<4, false>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
        case EV_KC_DENSE: me_ev_dense_run(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
        default: me_ev_run<1, false>(kb, T, ytg, xtg, ts_y_m, ts_x, lane, nvalid, A); break;
        }
    }
    for (int q = 0; q < ME_NQ; q++)
        if (A.mx[q] > 0.0)
            me_max(&mx[i * ME_NQ + q], A.mx[q]);
    if (A.bad)
        atomicAdd(&cnt[i * MEC_NQ + MEC_BAD], A.bad);
    if (A.danger)
        atomicAdd(&cnt[i * MEC_NQ + MEC_DANGER], A.danger);
    atomicAdd(&cnt[i * MEC_NQ + MEC_LANES], A.lanes);
    if (A.always)
        atomicAdd(&cnt[i * MEC_NQ + MEC_ALWAYS], A.always);
}

/* ---- k_synth_pd ------------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_model_err_pd(BatchDev p, double *mx, unsigned long long *cnt)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wt = (int)(gid % (size_t)p.ntiles);
    const int i = (int)((gid / (size_t)p.ntiles) % (size_t)p.nch);
    const size_t b = gid / ((size_t)p.ntiles * p.nch);
    if (b >= (size_t)p.nblocks)
        return;
    if (p.ch[b * p.nch + i].prn <= 0)
        return;
    const EvConst kb = p.evc[b * p.nch + i];
    const double *txb = p.tile_x + b * (size_t)p.ntiles * 2 * p.nch;
    const double ts_x = txb[(size_t)(2 * i) * p.ntiles + wt], ts_y = txb[(size_t)(2 * i + 1) * p.ntiles + wt];
    const bool down = kb.down != 0;
    const bool fixed = p.kph0 != nullptr;
    /* exactly what k_synth_pd puts into PdLds::tstate; the chip table's LDS address it adds to the code model is an integer
     * below 2^18: it changes neither the binade nor a rounding, and is left out here */
    const double guard = 0x1p+20 + (double)PD_BAND * 0x1p-32;
    const double mirror_at = fixed ? 512.0 - 0x1p-16 : 512.0;
    const double ts_y_m = down ? mirror_at - ts_y : ts_y;
    const double ytg = __fma_rn(ts_y_m, 8.0, guard), xtg = __fma_rn(ts_x, 2.0, guard);
    const double band = (double)PD_BAND * 0x1p-32;
    MeTruth T;
    T.cp = ts_y * (1.0 / 512.0);
    T.s = (down ? -kb.S : kb.S) * (1.0 / 512.0);
    T.x = ts_x;
    T.sc = kb.sc;
    T.laps = T.wraps = 0;
    T.down = down;
    T.fixed = fixed; /* the accumulator itself (c:2699, 2748) */
    T.st = fixed ? (uint32_t)p.kstep[b * p.nch + i] : 0u;
    T.ph = fixed ? p.kph0[b * p.nch + i] + (uint32_t)wt * (uint32_t)TILE * T.st : 0u;
    MeAcc A;
    for (int q = 0; q < ME_NQ; q++)
        A.mx[q] = 0.0;
    A.bad = A.danger = A.lanes = A.always = 0;
    A.up(ME_W_UNITS, (double)PD_BAND);
    unsigned long long dang = 0ull, bad = 0ull;
    for (int n = 0; n < TILE; n++) {
        const int lane = n & 63, j = n >> 6;
        /* exactly pd_channel_fast_*: one fma, then j additions */
        double y = __fma_rn((double)lane, kb.pd_S8, ytg), x = __fma_rn((double)lane, kb.pd_sc2, xtg);
        for (int q = 0; q < j; q++) {
            y = __dadd_rn(y, kb.pd_dy);
            x = __dadd_rn(x, kb.pd_dx);
        }
        const uint32_t lo = min(fixed ? 0xffffffffu : (uint32_t)__double2loint(y), (uint32_t)__double2loint(x));
        if (lo < p.pd_danger)
            dang |= 1ull << lane;
        if (wt * TILE + n < p.nsamp) {
            const int it = (int)(((uint32_t)__double2hiint(y) & 0xff8u) >> 3), ci = (int)(((uint32_t)__double2hiint(x) & 0xffffeu) >> 1);
            if ((down ? 511 - it : it) != T.it() || ci != T.ci())
                bad |= 1ull << lane;
            if (!fixed) {
                const double ey = fabs(me_centre((y - 0x1p+20) - 8.0 * T.u(), 4096.0) - band) * 0x1p+32;
                A.up(ME_Y0, ey);
                A.up(ME_Y0_W, ey / (double)PD_BAND);
                A.up(ME_PURE_Y, fabs(me_centre(__fma_rn((double)n, kb.S, ts_y_m) - T.u(), 512.0)) * 0x1p+32);
            }
            const double ex = fabs(((x - 0x1p+20) - 2.0 * T.ux()) - band) * 0x1p+32;
            A.up(ME_X0, ex);
            A.up(ME_X0_W, ex / (double)PD_BAND);
            A.up(ME_PURE_X, fabs(__fma_rn((double)n, kb.sc, ts_x) - T.ux()) * 0x1p+32);
        }
        T.step();
    }
    const bool always = kb.kc != EV_KC_DENSE;
    for (int q = 0; q < ME_NQ; q++)
        if (A.mx[q] > 0.0)
            me_max(&mx[i * ME_NQ + q], A.mx[q]);
    const unsigned long long nbad = (unsigned long long)__popcll(bad & ~dang);
    if (nbad && !always)
        atomicAdd(&cnt[i * MEC_NQ + MEC_BAD], nbad);
    atomicAdd(&cnt[i * MEC_NQ + MEC_DANGER], always ? 0ull : (unsigned long long)__popcll(dang));
    atomicAdd(&cnt[i * MEC_NQ + MEC_LANES], 64ull);
    if (always)
        atomicAdd(&cnt[i * MEC_NQ + MEC_ALWAYS], 64ull);
}

} /* namespace gpsbb_impl */
#endif
