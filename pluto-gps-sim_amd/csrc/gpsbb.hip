/*
 * gpsbb.hip — host side of libgpsbb: the C ABI of include/gpsbb.h over the HIP kernels in
 * gpsbb_kernels.hip.h.  Plain HIP runtime (streams, events, pinned memory); no torch, no CPU fallback:
 * every fill entry point fails with GPSBB_E_NODEVICE / GPSBB_E_HIP when there is no gfx950 device.
 *
 * Reference interface replaced: the inline sample loop plutogpssim.c:2689-2759 and its producer/consumer
 * contract with pluto_tx_thread_ep (plutogpssim.c:2146-2158).  See include/gpsbb.h.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

#include "gpsbb.h"
#include "gpsbb_kernels.hip.h"
#include "gpsbb_events.hip.h"
#include "gpsbb_dense.hip.h"
#include "gpsbb_walk.hip.h"
#include "gpsbb_laps.hip.h"
#include "gpsbb_nco.h"
#include "gpsbb_testhooks.h"
#ifdef GPSBB_EXPERIMENTS
#include "gpsbb_modelerr.hip.h"
#endif

using namespace gpsbb_impl;

/* Measurement knobs.  The library as shipped reads NO environment variable: a drop-in must not change its behaviour with
 * what happens to be in its host's environment.  The experiments build (make exp: -DGPSBB_EXPERIMENTS ->
 * libgpsbb_exp.so, which the tuning scripts under tools/ and the NCO unit tests load) turns the constants below into
 * getenv look-ups and exports the gpsbb_test_* hooks. */
#ifdef GPSBB_EXPERIMENTS
#define GPSBB_KNOB_LONG(name, dflt) ([](long d) -> long { static const char *const e = getenv(name); return e ? atol(e) : d; }((long)(dflt)))
#define GPSBB_KNOB_SET(name) ([]() -> bool { static const bool v = getenv(name) != nullptr; return v; }())
#else
#define GPSBB_KNOB_LONG(name, dflt) ((long)(dflt))
#define GPSBB_KNOB_SET(name) (false)
#endif

/* ================================================================================================== */
/* host-side tables                                                                                   */
/* ================================================================================================== */

namespace {

/* sinTable512 / cosTable512 (plutogpssim.c:93-161) from their closed form trunc(511*f(2*pi*i/512)+1.0);
 * guarded by a checksum of the 1024 values so that a libm that rounds differently fails loudly instead
 * of silently changing the output. */
constexpr uint64_t kSinCosFnv1a = 0x1c99a5cf84551314ull;

bool make_sincos(int32_t sin512[512], int32_t cos512[512])
{
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < 512; i++) {
        const double a = two_pi * (double)i / 512.0;
        sin512[i] = (int32_t)(511.0 * std::sin(a) + 1.0);
        cos512[i] = (int32_t)(511.0 * std::cos(a) + 1.0);
    }
    uint64_t h = 0xcbf29ce484222325ull;
    auto eat = [&h](const int32_t *t) {
        for (int i = 0; i < 512; i++)
            for (int k = 0; k < 4; k++) {
                h ^= (uint8_t)((uint32_t)t[i] >> (8 * k));
                h *= 0x100000001b3ull;
            }
    };
    eat(sin512);
    eat(cos512);
    return h == kSinCosFnv1a;
}

/* C/A code of one PRN (the table codegen() builds, plutogpssim.c:207-244), generated the ICD way:
 * two 10-stage LFSRs, G1 = 1+x^3+x^10, G2 = 1+x^2+x^3+x^6+x^8+x^9+x^10, both preset to all ones, and
 * the PRN picked by XOR-ing two G2 stages (the "phase selector"), which is equivalent to the G2 delay
 * table the reference uses (c:208-213). */
const uint8_t kG2Taps[32][2] = {
    {2, 6}, {3, 7}, {4, 8}, {5, 9}, {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4},
    {5, 6}, {6, 7}, {7, 8}, {8, 9}, {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {6, 9},
    {1, 3}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {8, 10}, {1, 6}, {2, 7}, {3, 8}, {4, 9}};

void make_ca(int prn, uint8_t ca[GPSBB_CA_LEN])
{
    /* bit k-1 of g = stage k */
    uint32_t g1 = 0x3ff, g2 = 0x3ff;
    const int t1 = kG2Taps[prn - 1][0], t2 = kG2Taps[prn - 1][1];
    for (int i = 0; i < GPSBB_CA_LEN; i++) {
        const uint32_t o1 = (g1 >> 9) & 1u;
        const uint32_t o2 = ((g2 >> (t1 - 1)) ^ (g2 >> (t2 - 1))) & 1u;
        ca[i] = (uint8_t)(o1 ^ o2);
        const uint32_t f1 = ((g1 >> 2) ^ (g1 >> 9)) & 1u;
        const uint32_t f2 = ((g2 >> 1) ^ (g2 >> 2) ^ (g2 >> 5) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 9)) & 1u;
        g1 = ((g1 << 1) | f1) & 0x3ff;
        g2 = ((g2 << 1) | f2) & 0x3ff;
    }
}

/* descriptor contract of gpsbb_chan_t (include/gpsbb.h) */
bool chan_ok(const gpsbb_chan_t &c, double delt, bool fixed = false)
{
    if (c.prn == 0)
        return true;
    if (c.prn < 0 || c.prn > 32)
        return false;
    if (!std::isfinite(c.f_carr) || !std::isfinite(c.f_code) || !std::isfinite(c.carr_phase) ||
        !std::isfinite(c.code_phase) || !std::isfinite(c.gain))
        return false;
    if (!fixed && (std::signbit(c.carr_phase) || c.carr_phase > 1.0))
        return false;
    if (fixed && (std::signbit(c.carr_phase) || c.carr_phase >= 4294967296.0 || c.carr_phase != std::floor(c.carr_phase)))
        return false; /* fixed-point variant: the value of the 32-bit accumulator */
    if (std::signbit(c.code_phase) || !(c.code_phase < 1023.0))
        return false;
    const double sc = c.f_code * delt, sk = c.f_carr * delt;
    if (!(sc > 0.0 && sc <= 1.5) || !(std::fabs(sk) <= 0.125))
        return false;
    if (!(std::fabs(c.gain) < 2097152.0))
        return false;
    if (c.iword < 0 || c.iword > 59 || c.ibit < 0 || c.ibit > 29 || c.icode < 0 || c.icode > 19)
        return false;
    for (int k = 0; k < GPSBB_N_DWRD; k++)
        if (c.dwrd[k] >> 30)
            return false;
    return true;
}

/* Upper bound on the rows build_rows() emits for one chain (see the derivation in DESIGN.md):
 * every lap (wrap to wrap) visits at most (top_e - e_s + 2) binades, each costing at most two rows,
 * plus a handful of explicit steps around the wrap. */
uint64_t row_bound(double s_abs, double range, int top_e, int nsamp)
{
    if (!(s_abs > 0.0))
        return 4;
    int es;
    std::frexp(s_abs, &es); /* s_abs = m * 2^es, m in [0.5,1)  ->  binade exponent es-1 */
    es -= 1;
    const double laps = std::floor((double)nsamp * s_abs / range) + 2.0;
    int binades = top_e - es + 3;
    if (binades < 3)
        binades = 3;
    const double per_lap = 2.0 * binades + 6.0;
    return (uint64_t)(laps * per_lap) + 16;
}

/*
 * Where a carrier will be n steps on, to ~1e-14 cycles, WITHOUT walking it: the phase pass B of the device-side chain
 * starts a segment from.  The reference's recurrence x = fl(x + s) does not advance by s per step but,
 * while x is in binade e, by s rounded to a multiple of that binade's last place (gpsbb_nco.h): a drift of
 * delta_e = RN(s / ulp_e) * ulp_e - s per step, i.e. of delta_e / |s| per unit of phase travelled there.  R(x) is that
 * density integrated from 0 to x (piecewise linear, one piece per binade from s's own up to [0.5, 1)); a path of
 * n steps from x0 covers whole laps R(1) each plus the two ends.  What is left out — the roundings of the steps that
 * cross a binade edge or wrap, and that a binade holds a whole number of steps — averages out: 6e-15 rms per 625 000
 * samples at 25 MS/s against 1e-11 for x0 + n*s (measured against the exact jump-ahead).  The chain's exactness does
 * not rest on this: fix_block takes any start phase within pass B's margin of the truth; a poor prediction only costs a
 * walk of the segment.
 */
struct CarrDrift {
    int e_lo = 0, n = 0;
    double sa = 0.0, R1 = 0.0;
    bool neg = false;
    double dens[64], cum[65];
    explicit CarrDrift(double s)
    {
        sa = std::fabs(s);
        neg = s < 0.0;
        if (!(sa >= 0x1p-60) || !(sa < 0.25))
            return; /* no model: plain arithmetic */
        int es;
        std::frexp(sa, &es);
        es -= 1; /* sa in [2^es, 2^(es+1)) */
        e_lo = es - 1 < -1 ? es - 1 : -1;
        cum[0] = 0.0;
        for (int e = e_lo; e <= -1; e++) {
            const double ulp = std::ldexp(1.0, e - 52);
            const double q = s / ulp; /* exact: a power of two */
            const double delta = (std::nearbyint(q) - q) * ulp;
            dens[n] = delta / sa;
            cum[n + 1] = cum[n] + dens[n] * std::ldexp(1.0, e); /* the binade is 2^e wide */
            n++;
        }
        R1 = cum[n];
    }
    double R(double x) const
    {
        if (n == 0 || !(x > std::ldexp(1.0, e_lo)))
            return 0.0;
        if (x >= 1.0)
            return R1;
        const int e = std::ilogb(x);
        const int k = e - e_lo;
        return cum[k] + dens[k] * (x - std::ldexp(1.0, e));
    }
    /* the phase n steps after x0 (both in [0, 1)) */
    double advance(double x0, int nsteps, double s) const
    {
        const double u = x0 + (double)nsteps * s;
        const double fl = std::floor(u), end = u - fl;
        const double drift = neg ? -fl * R1 + R(x0) - R(end) : fl * R1 + R(end) - R(x0);
        double v = end + drift;
        v -= std::floor(v);
        return v;
    }
};

/* A channel's bias W as a whole number of units of 2^-32 (rounded up): the guard format then carries exactly W — 2^20 + W is
 * representable — and not W rounded to the format's grid, half a unit of error that 1 / step would amplify. */
double ev_bias_on_grid(double w)
{
#ifdef GPSBB_W_OFF_GRID /* (measurement: rounds 2 and 3) */
    return w;
#else
    return w < 0.25 ? std::ceil(w * 4294967296.0) * 0x1p-32 : w;
#endif
}

/*
 * Can the breakpoint kernel render these blocks, and with which per-channel constants (EvConst)?  Eligible
 * when, for every active channel, a run of SPT samples holds at most one chip change (sc*15.5 < 1) and at
 * most EV_KC_MAX table-index changes, and, per block, the I sums cannot reach 2^15 (sum of 512*|gain|+1: the
 * packed I/Q arithmetic of that kernel needs it; the reference's (short) wrap-around is then unreachable
 * too).  Steps are the individually rounded products the kernels and the reference use (c:2709, 2741).
 */
bool ev_plan(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, std::vector<EvConst> &out, bool fixed = false)
{
    const size_t nbc = (size_t)nblocks * nch;
    out.resize(nbc);
    const double reach = (double)SPT - 0.5;
    for (int blk = 0; blk < nblocks; blk++) {
        double amp_sum = 0.0;
        for (int i = 0; i < nch; i++) {
            const gpsbb_chan_t &c = ch[(size_t)blk * nch + i];
            EvConst &K = out[(size_t)blk * nch + i];
            memset(&K, 0, sizeof K);
            if (c.prn <= 0)
                continue;
            amp_sum += 512.0 * std::fabs(c.gain) + 1.0;
            const volatile double sc = c.f_code * delt, sk = c.f_carr * delt;
            double S = sk * 512.0;
            if (fixed) {
                /* the 32-bit accumulator (c:2675, 2699): the table index is phase / 2^16 modulo 512, its step exactly
                 * step / 2^16.  Only k_synth_pd takes it (every channel evaluated per sample: checked below) */
                const volatile double scaled = 512.0 * 65536.0 * c.f_carr * delt;
                S = (double)(int)std::round(scaled) * 0x1p-16;
            }
            const double aS = std::fabs(S);
            /* more than one chip change per run: the channel is evaluated per sample (ev_dense), as long as the chip
             * table reaches past what a tile covers (which also leaves at most one code roll-over per tile) */
            const bool dense_code = !(sc * reach < 1.0);
            if (!(sc >= 0x1p-20) || !(1023.0 + 1040.0 * sc + 2.0 <= (double)EV_CHIP_LEN))
                return false;
            K.S = aS; /* a falling carrier is walked mirrored: phase -y, step |S| */
            K.sc = sc;
            K.rsc = 1.0 / sc;
            K.down = S < 0.0;
            /* how far an estimated change position may be off (in samples): the model's error over the step, plus the
             * roundings of the guard format (one unit in the last place is 2^-32 there) */
            const double wC = EV_MODEL_ERR * K.rsc + EV_T_EPS;
            double wK = 0.0;
            if (aS == 0.0) {
                K.rS = 0x1p+1000; /* the index never changes */
                K.kc = 1;
            } else if (aS < 8.0 * EV_MODEL_ERR) {
                K.rS = 0x1p+1000; /* the model error exceeds an eighth of a step (W would pass 1/8 sample): every run is recomputed exactly */
                K.kc = -1;
            } else {
                K.rS = 1.0 / aS;
                wK = EV_MODEL_ERR * K.rS + EV_T_EPS;
                const double kc = std::floor((reach + wK) * aS) + 1.0;
                K.kc = kc > (double)EV_KC_MAX ? EV_KC_DENSE : (int)kc; /* too many index changes per run: per sample */
            }
            /* one bias W for everything tested in the channel (first-sample fractions: W >= the model error in index units /
             * chips; change positions: W >= wK, wC): the tests are then all "low word of the biased quantity < 2W" */
            K.W = std::max(std::max(wK, wC), EV_T_EPS);
            K.W = ev_bias_on_grid(K.W);
            K.danger = K.W >= 0.25 ? 0x80000000u : (uint32_t)std::ceil(2.0 * K.W * 4294967296.0) + 1u;
            {
                /* (test aid, experiments build: a larger threshold sends more lane-runs to the exact path; never a smaller one) */
                const long floor_ = GPSBB_KNOB_LONG("GPSBB_EV_DANGER", 0);
                if (floor_ > 0 && (uint32_t)floor_ > K.danger)
                    K.danger = (uint32_t)floor_;
            }
            K.tK0 = K.rS * (1.0 + K.W) + 0x1p+20 + K.W;
            K.tC0 = K.rsc * (1.0 + K.W) + 0x1p+20 + K.W;
            K.pd_S8 = aS * 8.0;
            K.pd_dy = aS * 512.0;
            K.pd_sc2 = sc * 2.0;
            K.pd_dx = sc * 128.0;
            if (dense_code && K.kc > 0)
                K.kc = EV_KC_DENSE;
            if (fixed) {
                /* The accumulator's index model is exact (gpsbb_events.hip.h, ev_first<KC, true>): nothing on the carrier side is
                 * biased or tested, the change positions come from (1 - fraction - 2^-17) / |step|.  Per sample (k_synth_pd) where
                 * a run holds more than one chip change, per breakpoint otherwise; what neither takes goes to the stepped kernel. */
                if (dense_code) {
                    if (!(aS < 64.0))
                        return false;
                    K.kc = EV_KC_DENSE;
                } else {
                    const double kc = std::floor(reach * aS) + 1.0;
                    if (kc > (double)EV_KC_MAX)
                        return false;
                    K.kc = (int)kc;
                    K.rS = aS > 0.0 ? 1.0 / aS : 0x1p+1000;
                    K.W = ev_bias_on_grid(std::max(wC, EV_T_EPS));
                    K.danger = K.W >= 0.25 ? 0x80000000u : (uint32_t)std::ceil(2.0 * K.W * 4294967296.0) + 1u;
                    K.tK0 = K.rS * (1.0 - 0x1p-17) + 0x1p+20;
                    K.tC0 = K.rsc * (1.0 + K.W) + 0x1p+20 + K.W;
                }
            }
            /* what k_synth_ev's channel loop would otherwise work out per channel and tile in scalar instructions */
            K.danger_le = K.kc < 0 ? 0xffffffffu : K.danger - 1u; /* danger >= 1 */
            K.chip_at = (uint32_t)(offsetof(EvLdsLean, chip2) + (size_t)i * sizeof(uint16_t) * EvLdsLean::CHIPS) - (EV_GUARD_HI << 1);
            K.amp_at = (uint32_t)(offsetof(EvLdsLean, amp) + (size_t)i * sizeof(uint32_t) * EvLdsLean::AMP) - (EV_GUARD_HI << 2);
        }
        if (!(amp_sum < 32768.0))
            return false;
    }
    return true;
}

/* Does the lap-parallel pre-pass take these blocks (gpsbb_laps.hip.h, Eligibility)?  It is exact for any step it walks; what is
 * excluded is what its turn of the walk does not cover: steps more than 50 binades below the state (0 < |step| < 2^-50: below
 * 2e-8 Hz at 25 MS/s).  A step of exactly zero — a carrier without Doppler — is taken: the phase stands still (round 6). */
bool lap_eligible(const gpsbb_chan_t *ch, size_t nbc, double delt, bool fixed)
{
    for (size_t k = 0; k < nbc; k++) {
        const gpsbb_chan_t &c = ch[k];
        if (c.prn <= 0)
            continue;
        const volatile double sc = c.f_code * delt, sk = c.f_carr * delt;
        if (!(sc >= 0x1p-20))
            return false;
        if (!fixed && !(std::fabs(sk) >= 0x1p-50) && sk != 0.0) /* (exactly zero stands still: lap_run takes it) */
            return false;
    }
    return true;
}

/* room for the laps of every channel, in chunks of LAP_WG lanes: a chain of n steps of size s wraps at most floor(n * s / range)
 * + 1 times, a block may start a chain (one more lap), and the model's step differs from s by parts in 10^12 */
/* laps a lane walks (LapDev::unit): what a lane costs besides its walk is about one lap's walk (measured: 17 turns of ~55
 * vector instructions against ~900 for finding the lap, the model, the scan and the record), so a lane takes a few */
/* (GPSBB_LAP_UNIT_CODE = 0: a block's whole code chain in one lane — it starts from a known state, so nothing but the walk itself
 * is needed: 14 % fewer instructions, and measured SLOWER, 5.22e11 against 5.39e11: what a neighbour costs the synthesis kernel
 * is the time its wavefronts sit on a SIMD, not what they issue; a hundred laps in a row sit 0.4 ms) */
constexpr int LAP_UNIT_CARR = 4, LAP_UNIT_CODE = 2;
/* a burst of plain steps (lap_run) for the lanes that are due one when they are at least 1 / LAP_BURST_SHARE of the lanes still walking */
constexpr int LAP_BURST_SHARE = 4;
/* ... where there are lanes to spare: a batch of a few block-channels (one block: the drop-in call, whose LATENCY is the point)
 * has a few thousand laps for 1024 SIMDs — a lane per lap there (gpsbb_fill_block of the reference's geometry: 0.25 against
 * 0.31 ms) */
int lap_unit(int kind, size_t nbc)
{
    long u = kind == NCO_CARR ? GPSBB_KNOB_LONG("GPSBB_LAP_UNIT_CARR", LAP_UNIT_CARR) : GPSBB_KNOB_LONG("GPSBB_LAP_UNIT_CODE", LAP_UNIT_CODE);
    if (nbc <= 256)
        u = 1;
    else if (nbc <= 1024)
        u = kind == NCO_CARR ? std::min(u, 2L) : 1L;
    if (kind == NCO_CODE && u == 0)
        return 0; /* one lane per block: a code chain is a block long and starts from its descriptor's code_phase (c:2673) */
    return u < 1 ? 1 : (u > 64 ? 64 : (int)u);
}

void lap_bound(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp, bool fixed, uint32_t chunk0[2][GPSBB_MAX_CHAN + 1],
               bool carr_only = false)
{
    for (int kind = 0; kind < 2; kind++) {
        chunk0[kind][0] = kind == 0 ? 0u : chunk0[0][GPSBB_MAX_CHAN]; /* one array of chunks: the code chains', then the carriers' */
        for (int i = 0; i < nch; i++) {
            double laps = 0.0;
            for (int blk = 0; blk < nblocks; blk++) {
                const gpsbb_chan_t &c = ch[(size_t)blk * nch + i];
                if (c.prn <= 0 || (kind == NCO_CARR && fixed) || (kind == NCO_CODE && carr_only))
                    continue;
                const double s = kind == NCO_CARR ? std::fabs(c.f_carr * delt) : c.f_code * delt * (1.0 / 1023.0);
                /* (wraps, in lanes of `unit` laps, + a head and the rounding) */
                laps += std::floor((std::floor((double)nsamp * s * (1.0 + 0x1p-30)) + 1.0) / (double)std::max(1, lap_unit(kind, (size_t)nblocks * nch))) + 3.0;
            }
            const uint32_t chunks = (uint32_t)((laps + (double)(LAP_WG - 1)) / (double)LAP_WG) + 1u;
            chunk0[kind][i + 1] = chunk0[kind][i] + chunks;
        }
        for (int i = nch; i < GPSBB_MAX_CHAN; i++)
            chunk0[kind][i + 1] = chunk0[kind][i];
    }
}

} /* namespace */

/* ================================================================================================== */
/* handle / batch                                                                                     */
/* ================================================================================================== */

/* A few host threads that stay around between calls (host-side seeding of small batches): starting two
 * dozen threads costs more than the work they are given. */
struct WorkPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::function<void(size_t)> job;
    size_t njobs = 0, next = 0, running = 0;
    unsigned long gen = 0;
    bool stop = false;

    explicit WorkPool(size_t nworkers)
    {
        for (size_t t = 0; t < nworkers; t++)
            th.emplace_back([this] { loop(); });
    }
    ~WorkPool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th)
            t.join();
    }
    void loop()
    {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || gen != seen; });
            if (stop)
                return;
            seen = gen;
            drain(lk);
        }
    }
    /* take jobs until none is left; called with the lock held */
    void drain(std::unique_lock<std::mutex> &lk)
    {
        running++;
        while (next < njobs) {
            const size_t j = next++;
            lk.unlock();
            job(j);
            lk.lock();
        }
        if (--running == 0)
            cv_done.notify_all();
    }
    /* run f(0..n-1) on the workers and the calling thread; returns when all are done */
    void run(size_t n, std::function<void(size_t)> f)
    {
        std::unique_lock<std::mutex> lk(m);
        job = std::move(f);
        njobs = n;
        next = 0;
        gen++;
        cv_work.notify_all();
        drain(lk);
        cv_done.wait(lk, [&] { return running == 0 && next >= njobs; });
        njobs = 0;
    }
};

constexpr int SEED_STREAMS_MAX = 8;
constexpr int CHAIN_SEG_ROWS = 1750;    /* rows of a carrier chain per segment of the device-side chain, about (see batch_setup) */
constexpr int CHAIN_SEG_MIN_TILES = 16; /* ... but no segment shorter than this many tiles */
constexpr int CHAIN_SEG_MAX = 8;        /* segments per block at most */
constexpr int CHAIN_INDEP_MIN_TILES = 1024; /* independent blocks of at least this many tiles are cut into segments as well */
constexpr long CHAIN_MODEL_MAX_SEGS = 4096; /* segments per channel up to which pass B starts from the host's drift model */
constexpr unsigned STREAM_SEED_STREAMS = 4; /* pre-passes of a stream's pushes in flight.  Round 6, the lap-parallel pre-pass beside a synthesis
                                               kernel that keeps every CU to the end of its launch: for ONE handle 2, 3 and 4 give the same rate
                                               (5.43 - 5.60e11, tools/sweep_seed_streams.sh; 3.2 / 3.1 / 3.6 - 4.3 ms of pre-pass per push: what is
                                               in flight shares the slots the synthesis leaves) — but a second handle in the process (the node
                                               driver's shard beside the host's own handle: bench.py's node_driver.one_shard) ran at 0.73 of the
                                               headline with 3 and at 0.86 - 0.96 with 4 (tools/node_leg_exp.sh): which streams end up sharing a
                                               hardware queue depends on how many each handle creates */

struct gpsbb {
    int device = 0;
    hipStream_t s_seed = nullptr;    /* NCO seeding pre-pass (k_seed) and descriptor uploads            */
    hipStream_t s_more[SEED_STREAMS_MAX - 1] = {}; /* ... further ones, created on first use: every other batch, batches
                                                      that keep several pre-passes in flight, the pushes of a stream
                                                      (see batch_launch and seed_stream_at) */
    hipStream_t s_upload = nullptr;  /* descriptors and plans of a set-up: a stream of their own, so that they never queue
                                        behind an older push's pre-pass */
    unsigned batches_created = 0;
    hipStream_t s_compute = nullptr; /* synthesis kernel (k_synth)                                      */
    hipStream_t s_compute2 = nullptr; /* ... of every other launch: consecutive synthesis kernels work on different table sets and
                                         output ranges, so the head of one may fill the CUs the tail of the other leaves idle */
    unsigned compute_turn = 0;
    hipStream_t s_copy = nullptr;    /* device-to-host gather                                            */
    hipStream_t s_digest = nullptr;  /* gpsbb_slot_digest, created on first use: a stream nothing else waits on (the copy stream holds a
                                        wait for every push in flight: a digest queued there ran when the whole ring had drained) */
    std::vector<uint32_t> h_ca;      /* host copy of the C/A chips (seeding of small batches on the host)  */
    unsigned long long host_dwrd_oob = 0, host_itable_512 = 0; /* hazards counted by host-side seeding      */
    WorkPool *pool = nullptr;        /* host threads for seeding small batches (created on first use)      */
    int32_t *d_tabs = nullptr;
    uint32_t *d_ca = nullptr;
    uint32_t *d_status = nullptr;
    unsigned long long *d_hz = nullptr;
    int last_hip = 0;
    gpsbb_batch *scratch = nullptr;
    unsigned char *h_bounce = nullptr; /* pinned: a fill whose iq_out lies partly in a registered range is copied through here */
    size_t bounce_cap = 0;
    unsigned char *h_fill = nullptr; /* pinned: what the drop-in call brings back besides the IQ — the end states and the status word
                                        (fill_block_finish) */
    struct HostReg { char *host; char *dev; size_t bytes; };
    std::vector<HostReg> host_regs;         /* gpsbb_host_register: host ranges the device writes straight into */
    struct ChainOnly *chain_only = nullptr; /* device scratch of gpsbb_chain_carrier, kept between calls */
    int sm_count = 0;
    /* per-handle options (gpsbb_set_option) */
    int opt_seed_where = 0;   /* 0 by size (on the device: lap-parallel where eligible), 1 always the row walks (k_seed / k_walk), 2 always
                                 host threads, 3 always on the device, lap-parallel where eligible */
    int opt_synth_kernel = 0; /* 0 automatic, 1 always the per-sample kernel */
    int opt_skip_seed = 0;    /* measurement: re-use the tables of the first two runs of a batch */
    int opt_chain_where = 0;  /* GPSBB_CHAIN_CARRIER: 0 automatic (on the device wherever the pre-pass runs there), 1 host threads,
                                 2 as 0 with the fix-up walking the blocks in order (k_chain_fix instead of k_chain_fix_par) */
    int last_kernel = 0;      /* synthesis kernel of the last launch: 1 per-sample, 2 breakpoint */
    int last_chain_dev = 0;   /* the last launch resolved GPSBB_CHAIN_CARRIER on the device */
    int last_prepass = 0;     /* pre-pass of the last launch: 1 row walks on the device, 2 host threads, 3 lap-parallel on the device */
    struct DigestBuf { /* gpsbb_device_digest's scratch, kept between calls */
        unsigned long long *p = nullptr;
        size_t cap = 0;
        int reserve(size_t n)
        {
            if (n <= cap)
                return hipSuccess;
            if (p)
                (void)hipFree(p);
            p = nullptr;
            cap = 0;
            const hipError_t e = hipMalloc((void **)&p, n * sizeof(unsigned long long));
            if (e == hipSuccess)
                cap = n;
            return e;
        }
    } d_digest;
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; /* elements */
    /* room: a first allocation that expects to be outgrown (the row pool of a ring slot, whose pushes see different
     * Dopplers) takes that much more than asked, so that the stream does not stall on re-allocations later */
    int reserve(size_t n, size_t room = 0)
    {
        if (n <= cap)
            return hipSuccess;
        n += room;
        if (p) {
            (void)hipFree(p); /* synchronises the device: grow with head-room so that a ring whose slots see
                                 slightly different row counts stops re-allocating after a few pushes */
            n += n / 4;
        }
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess)
            return e;
        cap = n;
        return hipSuccess;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

/* Table sets of a batch.  Run k uses set k % nsets and its pre-pass may start as soon as the synthesis kernel
 * that last read that set has finished.  Three sets, and consecutive runs seed on alternating streams: a
 * pre-pass is as long as its longest chain whatever the batch size — longer than the synthesis it feeds — so
 * two of them have to be in flight for the synthesis kernel to set the pace; four, three pre-passes in flight on
 * three streams, where the carrier is chained on the device (two walks and the fix-up per run). */
constexpr int NSETS = 6;

struct gpsbb_batch {
    gpsbb *h = nullptr;
    /* the handle's seeding stream, or (odd slots of a ring) its second one: the pre-passes of two consecutive
     * slots — a few wavefronts each, as long as one chain takes — then run side by side (16-block slots:
     * 6.6e9 -> 7.3e9 samples/s at depth 3, 8.4e9 at depth 4).  Two, not one per slot: with the compute and
     * copy streams that makes four, and streams beyond the hardware queues share them. */
    hipStream_t seed_stream = nullptr;
    hipStream_t last_cs = nullptr; /* the synthesis stream of the last launch */
    bool want_digest = false;      /* this launch also leaves every block's digest in d_dig (GPSBB_PUSH_DIGEST): by the synthesis kernel
                                      itself where there is a variant that does (k_synth_ev_digest), by k_block_digest behind it otherwise */
    DevBuf<unsigned long long> d_dig;
    bool one_stream = false;       /* the drop-in call's scratch batch: upload, pre-pass and synthesis on the synthesis stream (a
                                      hop from stream to stream is 16 us of nothing for a call that takes 150: gpsbb_fill_block_ex) */
    int nblocks = 0, nch = 0, nsamp = 0, ntiles = 0;
    double delt = 0.0;
    unsigned flags = 0;
    uint64_t total_rows = 0;
    DevBuf<gpsbb_chan_t> d_ch;
    DevBuf<uint64_t> d_row_off;
    /* row tables / tile index / end states exist twice: run k uses set k&1, so that the seeding
     * pre-pass of run k+1 overlaps the synthesis kernel of run k (different streams) */
    DevBuf<NcoRow> d_rows[NSETS];
    DevBuf<int32_t> d_tile_row[NSETS];
    DevBuf<int32_t> d_row_cnt[NSETS];
    DevBuf<int32_t> d_tile_ctr;
    DevBuf<int32_t> d_seed_order; /* lane -> chain plan of k_seed (see BatchDev) */
    DevBuf<uint32_t> d_kph0; /* fixed-point carrier variant: start phase and step per (block, channel) */
    DevBuf<int32_t> d_kstep;
    std::vector<int32_t> h_seed_order;
    std::vector<uint32_t> h_kph0;
    std::vector<int32_t> h_kstep;
    const int *fixed_prev_prn = nullptr;      /* stream chaining of the fixed-point carrier (host side) */
    const uint32_t *fixed_prev_phase = nullptr;
    DevBuf<gpsbb_chan_state_t> d_end[NSETS];
    /* breakpoint kernel (ev): exact tile-start states instead of rows + tile index, and per-channel constants */
    bool ev = false;
    DevBuf<double> d_tile_x[NSETS];
    DevBuf<uint32_t> d_tile_nav[NSETS];
    DevBuf<EvConst> d_evc;
    std::vector<EvConst> h_evc;
    double *hs_tile_x = nullptr;
    uint32_t *hs_tile_nav = nullptr;
    size_t hs_tx_cap = 0, hs_tn_cap = 0;
    /* GPSBB_CHAIN_CARRIER resolved on the device (gpsbb_walk.hip.h: k_chain_prefix / k_chain_fix) */
    bool ev_dense = false; /* some channel is evaluated per sample: k_synth_ev<true> */
    bool ev_all_dense = false; /* every active channel is: k_synth_pd */
    bool chain_dev = false;
    bool chain_starts = false; /* ... with the per-sample kernel: the chain kernels only fix the blocks' start phases */
    DevBuf<int32_t> d_chain_order; /* chain_starts: the carrier chains, as k_walk's passes take them */
    int chain_lanes = 0;
    DevBuf<ChainAux> d_aux[NSETS];
    DevBuf<SynRow> d_prefix[NSETS];
    DevBuf<ChainDesc> d_cd;      /* what the chain kernels read of the descriptors (24 B per block-channel) */
    DevBuf<double> d_start0;     /* rough start phases: where pass A walks from */
    std::vector<ChainDesc> h_cd;
    std::vector<double> h_start0;
    int nseg = 1, seg_tiles = 0; /* the device-side chain cuts every block into nseg segments (BatchDev::nseg) */
    DevBuf<unsigned long long> d_fix_end; /* k_chain_fix_par: the hand-off between its chunks (BatchDev::fix_end) */
    DevBuf<int> d_fix_flag;
    int fix_epoch = 0, fix_chunks = 0, fix_wg = FIXP_WG_BATCH;
    bool fix_flags_zeroed = false; /* d_fix_flag has been cleared since it was last (re)allocated */
    bool chain_indep = false;    /* the chain machinery runs on a batch whose blocks are independent: only the segments of a block are chained */
    bool chain_model = false;    /* pass B starts from the host's drift model of the carrier (no pass A, no k_chain_prefix) */
    bool chain_fix_seq = false;  /* k_chain_fix (blocks in order) instead of k_chain_fix_par: GPSBB_OPT_CHAIN_WHERE 2 */
    bool host_seed = false;      /* the NCO tables of this batch are built on host threads: decided at set-up, like the
                                    chain (a run never re-reads the handle's options) */
    int carr_lanes = 0; /* lanes of the seed plan that walk carrier chains (they come first) */
    /* the lap-parallel pre-pass (gpsbb_laps.hip.h): one lane per lap of every chain; scratch per table set */
    bool laps = false;
    uint32_t lap_chunk0[2][GPSBB_MAX_CHAN + 1] = {};
    DevBuf<LapBC> d_lap_bc[NSETS];
    DevBuf<uint32_t> d_lap_lane0[NSETS], d_lap_cnt[NSETS], d_lap_chunk_bad[NSETS];
    DevBuf<LapRec> d_lap_rec[NSETS];
    DevBuf<LapAgg> d_lap_agg[NSETS];
    DevBuf<double> d_lap_chunk_m[NSETS];
    /* a stream's slot: the carrier continues from the push before (set by gpsbb_stream_push around set-up / launch) */
    ChainCarryDev *d_carry = nullptr;
    const int *carry_prn = nullptr;        /* in: prn per channel in the last block pushed before */
    double *carry_phase = nullptr;         /* in/out: the host's rough idea of the phase there / after this push */
    uint32_t cont0_mask = 0;
    hipEvent_t ev_prefix = nullptr, ev_fix = nullptr; /* the stream's: order k_chain_prefix / k_chain_fix across pushes */
    int max_sets = NSETS;
    /* pinned staging arena of the uploads of one set-up: hipMemcpyAsync from pageable memory is not asynchronous
     * (it waits for the stream's earlier work — the previous push's pre-pass — before it returns) */
    char *stage = nullptr;
    size_t stage_cap = 0, stage_used = 0;
    unsigned stream_turn = 0; /* the stream's push count */
    hipEvent_t synth_done_ref[NSETS] = {}; /* not owned: the run's ev[3] */
    hipEvent_t last_done = nullptr;
    bool synth_pending[NSETS] = {};
    hipEvent_t upload_done = nullptr; /* descriptors and plans of the last set-up are on the device */
    int nsets = 2;                    /* table sets in use: run k works on set k % nsets */
    unsigned run_count = 0;
    int last_set = 0;
    DevBuf<int16_t> d_iq;
    /* seeding on the host (small batches): pinned images of the row pool, tile index and end states */
    SynRow *hs_rows = nullptr;
    int32_t *hs_tile_row = nullptr;
    gpsbb_chan_state_t *hs_end = nullptr;
    size_t hs_rows_cap = 0, hs_tr_cap = 0, hs_end_cap = 0;
    std::vector<uint64_t> row_off;
    std::vector<gpsbb_chan_t> h_ch; /* library-owned copy: the caller's array may go away after the call */
    struct Ev4 { hipEvent_t e[4]; }; /* seed start/end (seed stream), synth start/end (compute stream) */
    std::vector<Ev4> evs; /* one set per run since the last timing reset */
    size_t ev_used = 0;
    bool ran = false;
    int16_t *last_iq = nullptr;
    int16_t *last_ext_iq = nullptr; /* the caller's device buffer of the last run, if it used one */
};

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (h)->last_hip = (int)e__;                                                              \
            return e__ == hipErrorOutOfMemory ? GPSBB_E_NOMEM : GPSBB_E_HIP;                       \
        }                                                                                          \
    } while (0)

extern "C" int gpsbb_version(void) { return GPSBB_VERSION; }

extern "C" const char *gpsbb_strerror(int err)
{
    switch (err) {
    case GPSBB_OK: return "ok";
    case GPSBB_E_BADARG: return "bad argument";
    case GPSBB_E_BADCHAN: return "channel descriptor outside the contract";
    case GPSBB_E_HIP: return "HIP runtime error";
    case GPSBB_E_NOMEM: return "out of memory";
    case GPSBB_E_INTERNAL: return "device self-check failed (row pool overflow)";
    case GPSBB_E_NODEVICE: return "no usable HIP device";
    case GPSBB_E_STATE: return "call sequence violation";
    default: return "unknown error";
    }
}

extern "C" int gpsbb_last_hip_error(const gpsbb_t *h) { return h ? h->last_hip : 0; }

extern "C" int gpsbb_set_option(gpsbb_t *h, int option, long value)
{
    if (!h)
        return GPSBB_E_BADARG;
    switch (option) {
    case GPSBB_OPT_SEED_WHERE:
        if (value < 0 || value > 3)
            return GPSBB_E_BADARG;
        h->opt_seed_where = (int)value;
        return GPSBB_OK;
    case GPSBB_OPT_SYNTH_KERNEL:
        if (value < 0 || value > 1)
            return GPSBB_E_BADARG;
        h->opt_synth_kernel = (int)value;
        return GPSBB_OK;
    case GPSBB_OPT_SKIP_SEED:
        h->opt_skip_seed = value != 0;
        return GPSBB_OK;
    case GPSBB_OPT_CHAIN_WHERE:
        if (value < 0 || value > 3)
            return GPSBB_E_BADARG;
        h->opt_chain_where = (int)value;
        return GPSBB_OK;
    default:
        return GPSBB_E_BADARG;
    }
}

extern "C" int gpsbb_get_info(gpsbb_t *h, int what, uint64_t *out)
{
    if (!h || !out)
        return GPSBB_E_BADARG;
    switch (what) {
    case GPSBB_INFO_LAST_KERNEL:
        *out = (uint64_t)h->last_kernel;
        return GPSBB_OK;
    case GPSBB_INFO_EXACT_RUNS:
    case GPSBB_INFO_CHAIN_FALLBACKS:
    case GPSBB_INFO_CHAIN_TIES:
    case GPSBB_INFO_CHAIN_REPAIRS:
    case GPSBB_INFO_TILES_RENDERED: {
        HIPCHK(h, hipSetDevice(h->device));
        unsigned long long v = 0;
        HIPCHK(h, hipMemcpy(&v, h->d_hz + (what == GPSBB_INFO_EXACT_RUNS ? 2 : (what == GPSBB_INFO_CHAIN_FALLBACKS ? 4 : (what == GPSBB_INFO_CHAIN_TIES ? 5 : (what == GPSBB_INFO_TILES_RENDERED ? 7 : 6)))), 8,
                            hipMemcpyDeviceToHost));
        *out = v;
        return GPSBB_OK;
    }
    case GPSBB_INFO_CHAIN_ON_DEVICE:
        *out = (uint64_t)h->last_chain_dev;
        return GPSBB_OK;
    case GPSBB_INFO_PREPASS:
        *out = (uint64_t)h->last_prepass;
        return GPSBB_OK;
    case GPSBB_INFO_STREAMS: {
        uint64_t n = 0;
        for (hipStream_t st : {h->s_seed, h->s_upload, h->s_compute, h->s_compute2, h->s_copy, h->s_digest})
            n += st != nullptr;
        for (hipStream_t st : h->s_more)
            n += st != nullptr;
        *out = n;
        return GPSBB_OK;
    }
    case GPSBB_INFO_HW_QUEUES: {
        /* reported, never acted on: what the runtime was (or will be) told when it maps streams onto hardware queues */
        const char *e = getenv("GPU_MAX_HW_QUEUES");
        const long v = e ? atol(e) : 4;
        *out = (uint64_t)(v > 0 ? v : 4);
        return GPSBB_OK;
    }
    default:
        return GPSBB_E_BADARG;
    }
}

extern "C" int gpsbb_codegen(int prn, uint8_t ca[GPSBB_CA_LEN])
{
    if (!ca || prn < 1 || prn > 32)
        return GPSBB_E_BADARG;
    make_ca(prn, ca);
    return GPSBB_OK;
}

extern "C" int gpsbb_sincos_tables(int32_t sin512[512], int32_t cos512[512])
{
    if (!sin512 || !cos512)
        return GPSBB_E_BADARG;
    return make_sincos(sin512, cos512) ? GPSBB_OK : GPSBB_E_INTERNAL;
}

static void chain_only_free(gpsbb *h);

extern "C" void gpsbb_destroy(gpsbb_t *h)
{
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    if (h->scratch)
        gpsbb_batch_destroy(h->scratch);
    if (h->h_fill)
        (void)hipHostFree(h->h_fill);
    if (h->h_bounce)
        (void)hipHostFree(h->h_bounce);
    for (const gpsbb::HostReg &r : h->host_regs)
        (void)hipHostUnregister(r.host);
    h->host_regs.clear();
    chain_only_free(h);
    if (h->s_seed)
        (void)hipStreamSynchronize(h->s_seed);
    for (hipStream_t st : h->s_more)
        if (st)
            (void)hipStreamSynchronize(st);
    if (h->s_upload)
        (void)hipStreamSynchronize(h->s_upload);
    if (h->s_compute)
        (void)hipStreamSynchronize(h->s_compute);
    if (h->s_compute2)
        (void)hipStreamSynchronize(h->s_compute2);
    if (h->s_copy)
        (void)hipStreamSynchronize(h->s_copy);
    if (h->s_digest)
        (void)hipStreamSynchronize(h->s_digest);
    if (h->d_tabs)
        (void)hipFree(h->d_tabs);
    if (h->d_ca)
        (void)hipFree(h->d_ca);
    if (h->d_status)
        (void)hipFree(h->d_status);
    if (h->d_hz)
        (void)hipFree(h->d_hz);
    if (h->d_digest.p)
        (void)hipFree(h->d_digest.p);
    delete h->pool;
    h->pool = nullptr;
    if (h->s_seed)
        (void)hipStreamDestroy(h->s_seed);
    for (hipStream_t st : h->s_more)
        if (st)
            (void)hipStreamDestroy(st);
    if (h->s_upload)
        (void)hipStreamDestroy(h->s_upload);
    if (h->s_compute)
        (void)hipStreamDestroy(h->s_compute);
    if (h->s_compute2)
        (void)hipStreamDestroy(h->s_compute2);
    if (h->s_copy)
        (void)hipStreamDestroy(h->s_copy);
    if (h->s_digest)
        (void)hipStreamDestroy(h->s_digest);
    delete h;
}

static hipError_t create_seed_stream(hipStream_t *st);

/* Zero device memory NOW.  The library's streams are non-blocking ones: nothing orders them behind the null stream, and a
 * hipMemset of device memory returns before it has happened — a kernel launched afterwards on one of those streams can run
 * first and have its result wiped (found by the node driver's stress, tools/stress_node.py: four handles on one GPU, the
 * cleared carry of a fresh stream landing after the first push's fix-up had written it — every later block of the shard off
 * by that push's phase, once in a hundred runs).  So: on a stream of the handle, and waited for. */
#ifdef GPSBB_EXPERIMENTS
/* the ordering test's helper: one lane that keeps its stream busy for `ticks` of the 100 MHz wall clock */
__global__ void k_park(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(64);
}
#endif

static hipError_t zero_now(gpsbb *h, void *ptr, size_t bytes)
{
#ifdef GPSBB_EXPERIMENTS
    {
        /* The deterministic version of the race of round 4 (tests/test_laps_gpu.py::test_the_null_stream_owns_nothing):
         * GPSBB_X_PARK_NULL_MS=<ms> parks a kernel on the NULL stream before every zeroing — whatever the library still put on the
         * null stream, or ordered behind it, then lands that many milliseconds later than the code around it assumes, every time
         * instead of once in a hundred runs under load; GPSBB_X_NULL_MEMSET brings round 3's bug back (the zeroing as a null-stream
         * memset nobody waits for), so that the test can be seen to fail on it. */
        const long park = GPSBB_KNOB_LONG("GPSBB_X_PARK_NULL_MS", 0);
        if (park > 0)
            hipLaunchKernelGGL(k_park, dim3(1), dim3(1), 0, nullptr, (unsigned long long)park * 100000ull);
        if (GPSBB_KNOB_SET("GPSBB_X_NULL_MEMSET"))
            return hipMemsetAsync(ptr, 0, bytes, nullptr);
    }
#endif
    const hipError_t e = hipMemsetAsync(ptr, 0, bytes, h->s_upload);
    return e != hipSuccess ? e : hipStreamSynchronize(h->s_upload);
}

extern "C" int gpsbb_create(gpsbb_t **out, int device)
{
    if (!out)
        return GPSBB_E_BADARG;
    *out = nullptr;
    /* Up to nine streams carry work at the same time (six pre-pass streams, upload, compute, gather).  The HIP runtime
     * maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one run one after the
     * other: a host that wants the stream rates of DESIGN.md exports GPU_MAX_HW_QUEUES=12 before its first HIP call
     * (gpsbb-sim and bench.py do).  The library itself neither reads nor writes the environment. */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return GPSBB_E_NODEVICE;

    int32_t tabs[1024];
    if (!make_sincos(tabs + 512, tabs)) /* device layout: cos first, then sin */
        return GPSBB_E_INTERNAL;
    std::vector<uint32_t> ca(33 * 32, 0u);
    for (int prn = 1; prn <= 32; prn++) {
        uint8_t chips[GPSBB_CA_LEN];
        make_ca(prn, chips);
        for (int i = 0; i < GPSBB_CA_LEN; i++)
            if (chips[i])
                ca[prn * 32 + (i >> 5)] |= 1u << (i & 31);
    }

    gpsbb *h = new (std::nothrow) gpsbb;
    if (!h)
        return GPSBB_E_NOMEM;
    h->device = device;
    auto fail = [&](hipError_t e) {
        h->last_hip = (int)e;
        gpsbb_destroy(h);
        return e == hipErrorOutOfMemory ? GPSBB_E_NOMEM : GPSBB_E_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail(e);
    h->sm_count = prop.multiProcessorCount;
    if ((e = create_seed_stream(&h->s_seed)) != hipSuccess) return fail(e);
    {
        /* (experiments: GPSBB_SYNTH_STREAM_PRIO = p gives the synthesis streams queue priority p — HIP: -1 high, 0 normal, 1 low —,
         * GPSBB_SEED_STREAM_PRIO the pre-pass streams: whose workgroup gets a CU that has just come free) */
        const long sp = GPSBB_KNOB_LONG("GPSBB_SYNTH_STREAM_PRIO", 0);
        if (sp != 0) {
            if ((e = hipStreamCreateWithPriority(&h->s_compute, hipStreamNonBlocking, (int)sp)) != hipSuccess) return fail(e);
            if ((e = hipStreamCreateWithPriority(&h->s_compute2, hipStreamNonBlocking, (int)sp)) != hipSuccess) return fail(e);
        } else {
            if ((e = hipStreamCreateWithFlags(&h->s_compute, hipStreamNonBlocking)) != hipSuccess) return fail(e);
            if ((e = hipStreamCreateWithFlags(&h->s_compute2, hipStreamNonBlocking)) != hipSuccess) return fail(e);
        }
    }
    if ((e = hipStreamCreateWithFlags(&h->s_upload, hipStreamNonBlocking)) != hipSuccess) return fail(e);
    if ((e = hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&h->d_tabs, sizeof tabs)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&h->d_ca, ca.size() * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&h->d_status, 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&h->d_hz, 64)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h->d_tabs, tabs, sizeof tabs, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h->d_ca, ca.data(), ca.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail(e); /* (the tables are there before any non-blocking stream runs) */
    h->h_ca = ca;
    if ((e = zero_now(h, h->d_status, 4)) != hipSuccess) return fail(e);
    if ((e = zero_now(h, h->d_hz, 64)) != hipSuccess) return fail(e);
    /* k_synth carves ~76 KB of dynamic LDS per workgroup: above the 64 KB default limit */
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(SynthLds))) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_ev), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(EvLdsLean) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_ev_digest), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(EvLdsLean) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_ev_dense), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(EvLds) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_ev_fixed), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(EvLdsLean) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_pd<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PdLds<true>) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_pd<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PdLds<false>) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_pd<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PdLds<true>) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_synth_pd<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PdLds<false>) + EV_PICK_LDS)) != hipSuccess) return fail(e);
    *out = h;
    return GPSBB_OK;
}

/* carry[i] = {prn, phase} of channel i after the previous call (in) / after this one (out); may be NULL */
struct ChainCarry {
    int prn[GPSBB_MAX_CHAN];
    double phase[GPSBB_MAX_CHAN];
};
static void chain_carrier_host(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp, double *seed,
                               int nthreads, ChainCarry *carry);

static bool host_seeding_wanted(const gpsbb_batch *b);

/* upload `bytes` from pageable `src` through the batch's pinned arena (grown at the start of a set-up) */
/* GPSBB_PUSH_TRACE=<ms>: where the host time of a stream push goes, printed for pushes that take longer than <ms> */
struct PushTrace {
    double limit_ms = -1.0;
    int n = 0;
    const char *what[32];
    std::chrono::steady_clock::time_point t[32];
    PushTrace()
    {
#ifdef GPSBB_EXPERIMENTS
        const char *e = getenv("GPSBB_PUSH_TRACE");
        if (e)
            limit_ms = atof(e);
#endif
    }
    void start() { n = 0; mark("start"); }
    void mark(const char *w)
    {
        if (limit_ms < 0.0 || n >= 32)
            return;
        what[n] = w;
        t[n++] = std::chrono::steady_clock::now();
    }
    void end()
    {
        if (limit_ms < 0.0 || n < 2)
            return;
        mark("end");
        const double tot = std::chrono::duration<double, std::milli>(t[n - 1] - t[0]).count();
        if (tot < limit_ms)
            return;
        fprintf(stderr, "[gpsbb push %.3f ms]", tot);
        for (int i = 1; i < n; i++)
            fprintf(stderr, " %s %.3f", what[i], std::chrono::duration<double, std::milli>(t[i] - t[i - 1]).count());
        fprintf(stderr, "\n");
    }
};
static thread_local PushTrace g_push_trace;
#define PUSH_MARK(w) g_push_trace.mark(w)

static hipError_t stage_upload(gpsbb_batch *b, void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    const size_t at = (b->stage_used + 63) & ~(size_t)63;
    if (at + bytes > b->stage_cap) /* cannot happen: the arena was sized for this set-up */
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    memcpy(b->stage + at, src, bytes);
    b->stage_used = at + bytes;
    return hipMemcpyAsync(dst, b->stage + at, bytes, hipMemcpyHostToDevice, stream);
}

/* ---- batch planning -------------------------------------------------------------------------------- */

static int batch_setup(gpsbb_batch *b, const gpsbb_chan_t *ch, int nblocks, int nch, double delt,
                       int nsamp, unsigned flags, hipStream_t upload_stream)
{
    gpsbb *h = b->h;
    if (!ch || nblocks < 1 || nblocks > 65535 || nch < 1 || nch > GPSBB_MAX_CHAN || nsamp < 1 ||
        !(delt > 0.0) || !std::isfinite(delt) || (flags & ~(GPSBB_CHAIN_CARRIER | GPSBB_FIXED_CARRIER)))
        return GPSBB_E_BADARG;
    const size_t nbc = (size_t)nblocks * nch;
    const bool fixed = (flags & GPSBB_FIXED_CARRIER) != 0;
    for (size_t k = 0; k < nbc; k++)
        if (!chan_ok(ch[k], delt, fixed))
            return GPSBB_E_BADCHAN;

    {
        /* everything a set-up uploads, with room to spare: descriptors, plans, per-channel constants, chain scratch */
        const size_t need = nbc * (sizeof(gpsbb_chan_t) + sizeof(EvConst) + 8 + 2 * 4 + 5 * 4 +
                                   (size_t)CHAIN_SEG_MAX * (sizeof(ChainDesc) + 8 + 8 + 2 * 4)) + 64 * 1024;
        if (b->upload_done) /* the previous set-up's copies out of the arena are long done; make sure */
            HIPCHK(h, hipEventSynchronize(b->upload_done));
        PUSH_MARK("arena");
        if (need > b->stage_cap) {
            if (b->stage)
                (void)hipHostFree(b->stage);
            b->stage = nullptr;
            b->stage_cap = 0;
            HIPCHK(h, hipHostMalloc((void **)&b->stage, need + need / 4, hipHostMallocDefault));
            b->stage_cap = need + need / 4;
        }
        b->stage_used = 0;
    }
    b->nblocks = nblocks;
    b->nch = nch;
    b->nsamp = nsamp;
    b->delt = delt;
    b->flags = flags;
    b->ntiles = (nsamp + TILE - 1) / TILE;
    if (2ull * nbc * ((unsigned long long)b->ntiles + 1) >= (1ull << 32))
        return GPSBB_E_NOMEM; /* the tile index is addressed with 32-bit element offsets (16 GiB of it) */

    /* Which synthesis kernel: the breakpoint kernel (gpsbb_events.hip.h) where every run of SPT samples holds
     * at most one chip change and at most EV_KC_MAX table-index changes and the I sums stay below 2^15. */
    b->ev = h->opt_synth_kernel != 1 && ev_plan(ch, nblocks, nch, delt, b->h_evc, fixed);
    b->ev_dense = false;
    b->ev_all_dense = b->ev;
    if (b->ev)
        for (size_t k = 0; k < nbc; k++) {
            b->ev_dense = b->ev_dense || b->h_evc[k].kc == EV_KC_DENSE;
            b->ev_all_dense = b->ev_all_dense && (ch[k].prn <= 0 || b->h_evc[k].kc == EV_KC_DENSE);
        }
    b->ev_all_dense = b->ev_all_dense && b->ev_dense;
    if (fixed && b->ev_dense && !b->ev_all_dense) {
        /* The fixed-point carrier has no mixed kernel: k_synth_ev_dense is the IEEE body (a falling phase mirrored as 512 - y,
         * biased change positions), not the accumulator's (512 - 2^-16 - y, exact).  A batch whose channels straddle the
         * one-chip-change-per-run limit (fs within ~50 Hz of 15.5 * 1.023e6 once the code Doppler is in) goes to the stepped
         * kernel. */
        b->ev = false;
        b->ev_dense = b->ev_all_dense = false;
    }
    PUSH_MARK("ev_plan");

    /* where the pre-pass runs and where the carrier chain is resolved: decided here, once, for all runs of the batch */
    /* on the device: lap-parallel (gpsbb_laps.hip.h) wherever the model kernels render and no step is tiny; the row walks
     * (k_walk and the chain kernels) for the rest and where GPSBB_OPT_SEED_WHERE / _CHAIN_WHERE ask for them.  The lap-parallel
     * pre-pass takes 0.1 ms whatever the size of the batch — less than host threads need for one block (tools/fill_latency.py:
     * 0.25 against 0.33 ms per gpsbb_fill_block of the reference's geometry) — so where it is eligible the size decides nothing. */
    const bool lap_ok = b->ev && h->opt_seed_where != 1 && h->opt_seed_where != 2 && h->opt_chain_where != 2 && h->opt_chain_where != 3 &&
                        !GPSBB_KNOB_SET("GPSBB_NO_LAPS") && lap_eligible(ch, nbc, delt, fixed);
    /* (a stream's push that was promised the device-side chain — b->d_carry: decided in gpsbb_stream_push on what it can see of
     * the descriptors, before the kernel plan exists — stays on the device whatever the size: the row walks where the laps
     * decline, e.g. a rate only the per-sample kernel renders) */
    b->host_seed = !lap_ok && !b->d_carry && host_seeding_wanted(b);
    b->laps = lap_ok;
    const bool chained = !fixed && (flags & GPSBB_CHAIN_CARRIER) && (nblocks > 1 || b->d_carry);
    b->chain_dev = chained && h->opt_chain_where != 1 && !b->host_seed;
    b->chain_fix_seq = h->opt_chain_where == 2;
    b->chain_starts = b->chain_dev && !b->ev;
    b->chain_model = false; /* decided below, once the number of segments is known */
    b->chain_indep = false;
    /* The device-side chain cuts blocks into SEGMENTS that are chained like blocks: a walk takes as long as its chain
     * whatever the batch (0.47 us per row; a 5 kHz carrier has 7 000 rows per 0.1 s of signal, at any sample rate), so
     * segments of about CHAIN_SEG_ROWS rows make the two walks of a pre-pass that many times shorter.  (Tried for batches
     * of independent blocks as well, every block's first segment starting a chain: the five dependent kernels of the
     * chain cost more than the shorter walks save — M1 geometry 1.77e11 -> 1.45e11 samples/s — so those keep k_walk<0>.) */
    b->nseg = 1;
    /* Batches of INDEPENDENT blocks go through the same machinery where the model of the carrier serves (no pass A): every
     * block's first segment starts a chain from its descriptor's phase, the walks are as many times shorter, and the three
     * kernels that follow cost less than a walk of whole blocks — for long blocks (25 MS/s, 2.5 M samples: 4.08e11 ->
     * 4.26e11 samples/s on a resident batch); for the reference's 300 000-sample blocks the fix-up over four thousand short
     * segments costs more than the walks save (1.72e11 -> 1.61e11): those keep k_walk<0>. */
    const bool indep_ok = !chained && !b->d_carry && b->ev && !fixed && !b->host_seed && h->opt_chain_where == 0 &&
                          b->ntiles >= (int)GPSBB_KNOB_LONG("GPSBB_INDEP_MIN_TILES", CHAIN_INDEP_MIN_TILES);
    if (!b->laps && ((b->chain_dev && !b->chain_starts) || indep_ok)) { /* (k_seed, the per-sample kernel's pre-pass, walks whole blocks) */
        double rows_max = 0.0;
        for (size_t k = 0; k < nbc; k++)
            if (ch[k].prn > 0) {
                const double sa = std::fabs(ch[k].f_carr * delt);
                const double r = sa > 0.0 ? ((double)nsamp * sa + 1.0) * (2.0 - std::log2(sa)) : 1.0;
                rows_max = r > rows_max ? r : rows_max;
            }
        int n = (int)(rows_max / (double)GPSBB_KNOB_LONG("GPSBB_SEG_ROWS", CHAIN_SEG_ROWS) + 0.5);
        const int n_cap = b->ntiles / CHAIN_SEG_MIN_TILES;
        n = n > CHAIN_SEG_MAX ? CHAIN_SEG_MAX : n;
        n = n > n_cap ? n_cap : n;
        n = n < 1 ? 1 : n;
        if (b->chain_dev) {
            b->nseg = n;
        } else if (n > 1 && (long)nblocks * n <= CHAIN_MODEL_MAX_SEGS) {
            b->chain_dev = true;
            b->chain_indep = true;
            b->nseg = n;
        }
    }
    b->seg_tiles = (b->ntiles + b->nseg - 1) / b->nseg;
    b->nseg = (b->ntiles + b->seg_tiles - 1) / b->seg_tiles; /* no empty last segment */
    /* Pass B's start phases from the host's drift model of the carrier (CarrDrift) instead of a first walk — where the
     * model's error cannot pile up: it is ~6e-15 cycles per segment (partly systematic), and a start phase further than
     * pass B's margin from the truth costs a walk of the segment (about one segment-channel in 10^5 per 1e-12 of error).  So:
     * batches of up to CHAIN_MODEL_MAX_SEGS segments.  Not the pushes of a stream — the belief can only be re-anchored on
     * end states that are a ring's depth of pushes old (measured: 3.3 segment walks per 400-block push, stream 4.48e11 ->
     * 4.26e11 samples/s), and not chains over tens of thousands of blocks (gpsbb_chain_carrier): those keep pass A. */
    b->chain_model = !b->laps && b->chain_dev && !b->chain_starts && !b->d_carry && h->opt_chain_where != 3 &&
                     (long)nblocks * b->nseg <= CHAIN_MODEL_MAX_SEGS;
    const size_t nvbc = nbc * (size_t)b->nseg; /* carrier chains: one per (segment, channel) */

    /* row pool plan: the code chains (block*nch + channel), then the carrier chains ((block*nseg + segment)*nch + channel) */
    b->row_off.assign(nbc + nvbc + 1, 0);
    uint64_t off = 0;
    for (size_t k = 0; k < nbc && !b->laps; k++) { /* (the lap-parallel pre-pass keeps no rows) */
        b->row_off[k] = off;
        if (ch[k].prn > 0) {
            off += row_bound(ch[k].f_code * delt, 1023.0, 9, nsamp) + 1;
            if (b->ev)
                off += (uint64_t)nsamp / (uint64_t)WALK_ROW_MAX + 1; /* k_walk cuts long rows */
        } else {
            off += 1;
        }
    }
    if (!b->laps) {
        /* a segment's bound depends on the block-channel's step and the segment's length only: one evaluation per
         * block-channel for the full segments, one for the (shorter) last */
        const int full = b->seg_tiles * TILE, last = nsamp - (b->nseg - 1) * full;
        const int ns_full = b->nseg == 1 ? nsamp : full, ns_last = b->nseg == 1 ? nsamp : last;
        for (int blk = 0; blk < nblocks; blk++)
            for (int sgi = 0; sgi < b->nseg; sgi++) {
                uint64_t *ro = &b->row_off[nbc + ((size_t)blk * b->nseg + sgi) * nch];
                const int ns = sgi == b->nseg - 1 ? ns_last : ns_full;
                for (int i = 0; i < nch; i++) {
                    const gpsbb_chan_t &c = ch[(size_t)blk * nch + i];
                    ro[i] = 0; /* count first, offsets below */
                    if (c.prn > 0 && !fixed) {
                        if (sgi == 0 || sgi == b->nseg - 1)
                            ro[i] = row_bound(std::fabs(c.f_carr * delt), 1.0, -1, ns) + 1 + (b->ev ? (uint64_t)ns / (uint64_t)WALK_ROW_MAX + 1 : 0);
                        else
                            ro[i] = b->row_off[nbc + ((size_t)blk * b->nseg) * nch + i]; /* as the block's first segment */
                    } else {
                        ro[i] = 1;
                    }
                }
            }
        /* (the first segments' entries are read above while later ones are filled: turn counts into offsets afterwards) */
        for (size_t kv = 0; kv < nvbc; kv++) {
            const uint64_t cnt = b->row_off[nbc + kv];
            b->row_off[nbc + kv] = off;
            off += cnt;
        }
    }
    b->row_off[nbc + nvbc] = off;
    b->total_rows = off;

    /* device scratch of both table sets, sized here so that a launch never allocates (growing frees, and
     * a free synchronises the device) */
    HIPCHK(h, (hipError_t)b->d_ch.reserve(nbc));
    HIPCHK(h, (hipError_t)b->d_row_off.reserve(nbc + nvbc + 1));
    HIPCHK(h, (hipError_t)b->d_tile_ctr.reserve((size_t)NSETS * ((size_t)nblocks + 1))); /* one set of counters per table set: a block's next tile, [nblocks] the helpers' tickets (ev_pick_block) */
    /* table sets = pre-passes in flight + 1: the pre-pass of either kernel takes longer than the synthesis it feeds
     * (M1 geometry, per-sample kernel: two sets 6.6e10, three 7.7e10 samples/s), the chained ones longer still */
    b->nsets = (int)GPSBB_KNOB_LONG("GPSBB_NSETS", b->chain_dev ? 4 : 3);
    b->nsets = b->nsets < 2 ? 2 : (b->nsets > NSETS ? NSETS : b->nsets);
    b->nsets = b->nsets > b->max_sets ? b->max_sets : b->nsets;
    for (int set = 0; set < b->nsets; set++) {
        HIPCHK(h, (hipError_t)b->d_end[set].reserve(nbc));
        if (b->ev) {
            HIPCHK(h, (hipError_t)b->d_tile_x[set].reserve(2 * nbc * (size_t)b->ntiles));
            HIPCHK(h, (hipError_t)b->d_tile_nav[set].reserve(nbc * (size_t)b->ntiles));
            HIPCHK(h, (hipError_t)b->d_rows[set].reserve(b->total_rows + 4, b->max_sets == 1 ? (size_t)(b->total_rows / 2) : 0));
            HIPCHK(h, (hipError_t)b->d_row_cnt[set].reserve(nbc + nvbc));
        } else {
            HIPCHK(h, (hipError_t)b->d_rows[set].reserve(b->total_rows + 4)); /* + slack: k_synth prefetches one row past a chain */
            HIPCHK(h, (hipError_t)b->d_tile_row[set].reserve(2 * nbc * ((size_t)b->ntiles + 1)));
            HIPCHK(h, (hipError_t)b->d_row_cnt[set].reserve(2 * nbc));
        }
    }
    if (b->ev) {
        HIPCHK(h, (hipError_t)b->d_evc.reserve(nbc));
        HIPCHK(h, stage_upload(b, b->d_evc.p, b->h_evc.data(), nbc * sizeof(EvConst), upload_stream));
    }
    if (fixed) {
        /* start phase and step of the 32-bit accumulator per (block, channel); the chain across blocks is
         * plain modular arithmetic, resolved here (c:2675, 2748) */
        b->h_kph0.assign(nbc, 0u);
        b->h_kstep.assign(nbc, 0);
        for (int i = 0; i < nch; i++) {
            int prev_prn = b->fixed_prev_prn ? b->fixed_prev_prn[i] : 0;
            uint32_t prev_ph = b->fixed_prev_phase ? b->fixed_prev_phase[i] : 0u;
            for (int blk = 0; blk < nblocks; blk++) {
                const gpsbb_chan_t &c = ch[(size_t)blk * nch + i];
                const size_t k = (size_t)blk * nch + i;
                if (c.prn <= 0) {
                    prev_prn = 0;
                    continue;
                }
                const volatile double scaled = 512.0 * 65536.0 * c.f_carr * delt;
                b->h_kstep[k] = (int)std::round(scaled);
                const bool cont = (flags & GPSBB_CHAIN_CARRIER) && c.prn == prev_prn;
                b->h_kph0[k] = cont ? prev_ph : (uint32_t)c.carr_phase;
                prev_ph = b->h_kph0[k] + (uint32_t)nsamp * (uint32_t)b->h_kstep[k];
                prev_prn = c.prn;
            }
        }
        HIPCHK(h, (hipError_t)b->d_kph0.reserve(nbc));
        HIPCHK(h, (hipError_t)b->d_kstep.reserve(nbc));
        HIPCHK(h, stage_upload(b, b->d_kph0.p, b->h_kph0.data(), nbc * 4, upload_stream));
        HIPCHK(h, stage_upload(b, b->d_kstep.p, b->h_kstep.data(), nbc * 4, upload_stream));
    }
    b->h_ch.assign(ch, ch + nbc);
    b->cont0_mask = 0;
    if (b->laps) {
        /* the lap-parallel pre-pass: room for the laps of every channel, scratch per table set; a stream's push: which
         * channels of its first block go on from the push before (the exact phase is on the device) */
        lap_bound(ch, nblocks, nch, delt, nsamp, fixed, b->lap_chunk0);
        const size_t chunks = (size_t)b->lap_chunk0[1][nch];
        for (int set = 0; set < b->nsets; set++) {
            HIPCHK(h, (hipError_t)b->d_lap_bc[set].reserve(2 * nbc));
            HIPCHK(h, (hipError_t)b->d_lap_lane0[set].reserve(2 * (size_t)nch * ((size_t)nblocks + 1)));
            HIPCHK(h, (hipError_t)b->d_lap_cnt[set].reserve(4 * GPSBB_MAX_CHAN));
            HIPCHK(h, (hipError_t)b->d_lap_rec[set].reserve(chunks * LAP_WG, b->max_sets == 1 ? chunks * LAP_WG / 4 : 0));
            HIPCHK(h, (hipError_t)b->d_lap_agg[set].reserve(chunks, b->max_sets == 1 ? chunks / 4 : 0));
            HIPCHK(h, (hipError_t)b->d_lap_chunk_m[set].reserve(chunks, b->max_sets == 1 ? chunks / 4 : 0));
            HIPCHK(h, (hipError_t)b->d_lap_chunk_bad[set].reserve(chunks, b->max_sets == 1 ? chunks / 4 : 0));
        }
        if (b->chain_dev && b->d_carry && b->carry_prn)
            for (int i = 0; i < nch; i++)
                if (ch[i].prn > 0 && ch[i].prn == b->carry_prn[i])
                    b->cont0_mask |= 1u << i;
    }
    if (b->chain_dev && !b->laps) {
        /* The carrier chain is resolved exactly on the device, in parallel over the blocks (k_walk pass A,
         * k_chain_prefix, k_walk pass B, k_chain_fix).  All the host contributes is a rough start phase per block:
         * the descriptor's phase carried forward by nsamp*step in plain double arithmetic (good to ~1e-7 cycles
         * after a few hundred blocks; pass A takes it from there). */
        b->h_cd.resize(nvbc);
        b->h_start0.resize(nvbc);
        for (int i = 0; i < nch; i++) {
            double x = b->d_carry && b->carry_phase ? b->carry_phase[i] : 0.0;
            int prev_prn = b->d_carry && b->carry_prn ? b->carry_prn[i] : 0;
            for (int blk = 0; blk < nblocks; blk++) {
                const gpsbb_chan_t &c = ch[(size_t)blk * nch + i];
                const volatile double sk = c.f_carr * delt;
                const CarrDrift drift(b->chain_model && c.prn > 0 ? (double)sk : 0.0);
                if (c.prn > 0) {
                    if (c.prn != prev_prn || b->chain_indep)
                        x = c.carr_phase;
                    else if (blk == 0)
                        b->cont0_mask |= 1u << i;
                }
                for (int sgi = 0; sgi < b->nseg; sgi++) {
                    const size_t kv = ((size_t)blk * b->nseg + sgi) * nch + i;
                    ChainDesc &cd = b->h_cd[kv];
                    cd.f_carr = c.f_carr;
                    cd.carr_phase = c.carr_phase; /* read for a block's first segment only (one that starts a chain) */
                    cd.prn = c.prn;
                    cd.start = (b->chain_indep && sgi == 0) ? 1 : 0;
                    b->h_start0[kv] = c.prn > 0 ? x : 0.0;
                    if (c.prn > 0) {
                        const int left = nsamp - sgi * b->seg_tiles * TILE, full = b->seg_tiles * TILE;
                        const int ns = b->nseg == 1 ? nsamp : (left < full ? left : full);
                        if (b->chain_model && x < 1.0) {
                            x = drift.advance(x, ns, sk);
                        } else {
                            x = x + (double)ns * sk;
                            x -= std::floor(x);
                        }
                    }
                }
                prev_prn = c.prn > 0 ? c.prn : 0;
            }
            if (b->d_carry && b->carry_phase)
                b->carry_phase[i] = x;
        }
        /* the chain's scratch (ChainAux) needs no initial image: every field is written by the pass that owns it */
        for (int set = 0; set < b->nsets; set++) {
            HIPCHK(h, (hipError_t)b->d_aux[set].reserve(nvbc));
            if (!b->chain_starts)
                HIPCHK(h, (hipError_t)b->d_prefix[set].reserve(nvbc * (size_t)CHAIN_PREFIX_CAP));
        }
        HIPCHK(h, (hipError_t)b->d_cd.reserve(nvbc));
        HIPCHK(h, (hipError_t)b->d_start0.reserve(nvbc));
        {
            /* flags are compared with a launch number, never cleared: zeroed once, when the buffer is (re)allocated */
            /* long chains (thousands of segments per channel): fewer, larger chunks — fewer hand-offs */
            b->fix_wg = nblocks * b->nseg >= 2048 ? FIXP_WG_ALONE : FIXP_WG_BATCH;
            b->fix_chunks = (nblocks * b->nseg + b->fix_wg - 1) / b->fix_wg;
            /* one set of flags per table set: runs of a resident batch overlap, each on its own table set */
            const size_t nf = (size_t)NSETS * GPSBB_MAX_CHAN * b->fix_chunks;
            if (nf > b->d_fix_flag.cap || b->d_fix_end.cap < b->d_fix_flag.cap || !b->fix_flags_zeroed) {
                /* (all three steps or none: a set-up that failed half-way must not leave flags that were never zeroed, or no
                 * buffer for the end phases, behind a capacity that says "nothing to do") */
                b->fix_flags_zeroed = false;
                HIPCHK(h, (hipError_t)b->d_fix_flag.reserve(nf));
                HIPCHK(h, (hipError_t)b->d_fix_end.reserve(b->d_fix_flag.cap));
                HIPCHK(h, hipMemsetAsync(b->d_fix_flag.p, 0, b->d_fix_flag.cap * sizeof(int), upload_stream));
                b->fix_epoch = 0;
                b->fix_flags_zeroed = true;
            }
        }
        HIPCHK(h, stage_upload(b, b->d_cd.p, b->h_cd.data(), nvbc * sizeof(ChainDesc), upload_stream));
        HIPCHK(h, stage_upload(b, b->d_start0.p, b->h_start0.data(), nvbc * sizeof(double), upload_stream));
    }
    if ((flags & GPSBB_CHAIN_CARRIER) && !fixed && nblocks > 1 && !b->chain_dev) {
        /* blocks consecutive in time: resolve the carrier phase at the start of every block here, exactly
         * (same jump-ahead as the device, one host thread per channel), so that the device's chains are all
         * independent.  Walking the blocks in order on the device would serialise the whole pre-pass. */
        std::vector<double> seeds(nbc);
        chain_carrier_host(ch, nblocks, nch, delt, nsamp, seeds.data(), 0, nullptr);
        for (size_t k = 0; k < nbc; k++)
            if (b->h_ch[k].prn > 0)
                b->h_ch[k].carr_phase = seeds[k];
    }
    PUSH_MARK("aux");
    HIPCHK(h, stage_upload(b, b->d_ch.p, b->h_ch.data(), nbc * sizeof(gpsbb_chan_t), upload_stream));
    PUSH_MARK("up_ch");
    if (!b->laps)
        HIPCHK(h, stage_upload(b, b->d_row_off.p, b->row_off.data(), (nbc + nvbc + 1) * 8, upload_stream));
    if (b->laps) {
        b->h_seed_order.clear();
        b->carr_lanes = 0;
    } else {
        /* which chain each lane of k_seed walks (BatchDev::seed_order).  k_seed takes as long as its slowest
         * wavefront: rows of its longest chain x the time of one turn of the loop, which grows with the
         * number of lanes that are out of step.  Measured (400 x 16 chains, |f_carr| uniform up to 5 kHz):
         * 6.3 ms in block order, 6.1 ms with the carrier chains by descending |f_carr|, 5.3 ms with the
         * longest of them in wavefronts of few lanes.  (16 chains per wavefront throughout does not help
         * small batches: 16-block ring slots 6.0e9 vs 6.6e9 samples/s.) */
        /* carrier chains: one per (segment, channel), kv = (block*nseg + segment)*nch + channel (nseg = 1: per block) */
        const size_t nvbc = nbc * (size_t)b->nseg;
        const int nseg = b->nseg;
        std::vector<int32_t> carr(nvbc);
        const gpsbb_chan_t *hc = b->h_ch.data();
        const bool by_sign = b->ev; /* k_walk runs the two directions in separate loops: keep them in separate wavefronts */
        {
            /* By direction (rising first), then by descending |f_carr| — a wavefront runs as long as its longest chain —
             * in CARR_BUCKETS classes of |f_carr| (a counting sort: the plan of a 400-block push with four segments per
             * block orders 25 600 chains, and a comparison sort of them cost more than everything else in the push). */
            constexpr int CARR_BUCKETS = 512;
            double fmax = 0.0;
            for (size_t k = 0; k < nbc; k++)
                if (hc[k].prn > 0 && std::fabs(hc[k].f_carr) > fmax)
                    fmax = std::fabs(hc[k].f_carr);
            const double scale = fmax > 0.0 ? (CARR_BUCKETS - 1) / fmax : 0.0;
            std::vector<uint16_t> key(nbc);
            std::vector<uint32_t> head(2 * CARR_BUCKETS + 2, 0u);
            for (size_t k = 0; k < nbc; k++) {
                unsigned kk;
                if (hc[k].prn <= 0) {
                    kk = 2 * CARR_BUCKETS; /* idle channels last */
                } else {
                    const unsigned q = (unsigned)(CARR_BUCKETS - 1) - (unsigned)(std::fabs(hc[k].f_carr) * scale);
                    kk = (by_sign && std::signbit(hc[k].f_carr) ? CARR_BUCKETS : 0) + (q < (unsigned)CARR_BUCKETS ? q : CARR_BUCKETS - 1);
                }
                key[k] = (uint16_t)kk;
                head[kk + 1] += (uint32_t)nseg;
            }
            for (size_t j = 1; j < head.size(); j++)
                head[j] += head[j - 1];
            for (size_t vb = 0; vb < (size_t)nblocks * nseg; vb++)
                for (int i = 0; i < nch; i++)
                    carr[head[key[(vb / nseg) * nch + i]]++] = (int32_t)(vb * nch + i);
        }
        std::vector<int32_t> &order = b->h_seed_order;
        order.clear();
        auto waves_of = [&](const int32_t *chains, size_t n, size_t per_wave, int32_t add) {
            for (size_t c = 0; c < n; c += per_wave)
                for (size_t l = 0; l < 64; l++)
                    order.push_back(l < per_wave && c + l < n ? chains[c + l] + add : -1);
        };
        std::vector<int32_t> code(nbc);
        for (size_t k = 0; k < nbc; k++)
            code[k] = (int32_t)k;
        if (b->ev) {
            /* k_walk keeps the lanes of a wavefront in lockstep: a turn of its loop costs the same however many
             * lanes take part, so wavefronts are full, the carrier chains by descending |f_carr| (a wavefront runs
             * as long as its longest chain) and the longest ones first */
            const size_t lanes_per_wave = (size_t)GPSBB_KNOB_LONG("GPSBB_WALK_LANES", 64);
            waves_of(carr.data(), nvbc, lanes_per_wave, (int32_t)nbc);
            b->carr_lanes = (int)order.size();
            waves_of(code.data(), nbc, 64, 0);
        } else {
            waves_of(code.data(), nbc, 64, 0);
            /* the longest 8 % in wavefronts of 8, the next 16 % in wavefronts of 16, the next 32 % in wavefronts of 32 */
            const size_t n8 = nbc * 8 / 100 / 8 * 8, n16 = nbc * 16 / 100 / 16 * 16, n32 = nbc * 32 / 100 / 32 * 32;
            waves_of(carr.data(), n8, 8, (int32_t)nbc);
            waves_of(carr.data() + n8, n16, 16, (int32_t)nbc);
            waves_of(carr.data() + n8 + n16, n32, 32, (int32_t)nbc);
            waves_of(carr.data() + n8 + n16 + n32, nbc - n8 - n16 - n32, 64, (int32_t)nbc);
        }
        HIPCHK(h, (hipError_t)b->d_seed_order.reserve(order.size()));
        PUSH_MARK("order");
        HIPCHK(h, stage_upload(b, b->d_seed_order.p, order.data(), order.size() * 4, upload_stream));
        if (b->chain_starts) {
            /* the chain's two walks take the carrier chains alone, in lockstep: by direction, then by |f_carr| (nseg = 1
             * here: chains are blocks) */
            std::stable_sort(carr.begin(), carr.end(), [hc](int32_t x, int32_t y) {
                const bool nx = hc[x].prn > 0 && std::signbit(hc[x].f_carr), ny = hc[y].prn > 0 && std::signbit(hc[y].f_carr);
                return nx != ny && ny;
            });
            std::vector<int32_t> co;
            for (size_t c = 0; c < nbc; c += 64)
                for (size_t l = 0; l < 64; l++)
                    co.push_back(c + l < nbc ? carr[c + l] + (int32_t)nbc : -1);
            b->chain_lanes = (int)co.size();
            HIPCHK(h, (hipError_t)b->d_chain_order.reserve(co.size()));
            HIPCHK(h, stage_upload(b, b->d_chain_order.p, co.data(), co.size() * 4, upload_stream));
        }
    }
    if (!b->upload_done)
        HIPCHK(h, hipEventCreateWithFlags(&b->upload_done, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(b->upload_done, upload_stream));
    b->ran = false;
    return GPSBB_OK;
}

static gpsbb_batch *batch_new(gpsbb *h)
{
    gpsbb_batch *b = new (std::nothrow) gpsbb_batch;
    if (!b)
        return nullptr;
    b->h = h;
    b->seed_stream = h->s_seed;
    return b;
}

/* A stream for pre-pass kernels.  Experiments build: GPSBB_SEED_CUS = n confines it to n compute units (every
 * GPSBB_SEED_CU_STRIDE-th bit of the CU mask, default 1), to see what the pre-pass kernels' presence on a CU costs the
 * synthesis kernel there. */
static hipError_t create_seed_stream(hipStream_t *st)
{
    const long ncu = GPSBB_KNOB_LONG("GPSBB_SEED_CUS", 0);
    if (ncu > 0) {
        const long stride = std::max(1L, GPSBB_KNOB_LONG("GPSBB_SEED_CU_STRIDE", 1));
        const long first = GPSBB_KNOB_LONG("GPSBB_SEED_CU_FIRST", 0);
        uint32_t mask[16] = {0};
        for (long k = 0; k < ncu; k++) {
            const long bit = first + k * stride;
            if (bit < 512)
                mask[bit >> 5] |= 1u << (bit & 31);
        }
        return hipExtStreamCreateWithCUMask(st, 16, mask);
    }
    {
        const long pp = GPSBB_KNOB_LONG("GPSBB_SEED_STREAM_PRIO", 0);
        if (pp != 0)
            return hipStreamCreateWithPriority(st, hipStreamNonBlocking, (int)pp);
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

/* seeding stream k of the handle (0 = s_seed), created on first use */
static hipError_t seed_stream_at(gpsbb *h, unsigned k, hipStream_t *out)
{
    if (k == 0 || k >= (unsigned)SEED_STREAMS_MAX) {
        *out = h->s_seed;
        return hipSuccess;
    }
    hipStream_t &st = h->s_more[k - 1];
    if (!st) {
        hipError_t e = create_seed_stream(&st);
        if (e != hipSuccess)
            return e;
    }
    *out = st;
    return hipSuccess;
}

/* every other batch (and every other slot of a ring) seeds on the handle's second stream */
static hipError_t use_second_seed_stream(gpsbb_batch *b)
{
    return seed_stream_at(b->h, 1, &b->seed_stream);
}

extern "C" void gpsbb_batch_destroy(gpsbb_batch_t *b)
{
    if (!b)
        return;
    (void)hipSetDevice(b->h->device);
    (void)hipStreamSynchronize(b->h->s_seed);
    for (hipStream_t st : b->h->s_more)
        if (st)
            (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(b->h->s_compute);
    (void)hipStreamSynchronize(b->h->s_compute2);
    b->d_ch.release();
    b->d_row_off.release();
    if (b->upload_done)
        (void)hipEventDestroy(b->upload_done);
    for (int k = 0; k < NSETS; k++) {
        b->d_rows[k].release();
        b->d_tile_row[k].release();
        b->d_row_cnt[k].release();
        b->d_tile_ctr.release();
        b->d_seed_order.release();
        b->d_kph0.release();
        b->d_kstep.release();
        b->d_end[k].release();
        b->d_tile_x[k].release();
        b->d_tile_nav[k].release();
        b->d_aux[k].release();
        b->d_cd.release();
        b->d_start0.release();
        b->d_fix_end.release();
        b->d_fix_flag.release();
        b->d_prefix[k].release();
        b->d_lap_bc[k].release();
        b->d_lap_lane0[k].release();
        b->d_lap_cnt[k].release();
        b->d_lap_chunk_bad[k].release();
        b->d_lap_rec[k].release();
        b->d_lap_agg[k].release();
        b->d_lap_chunk_m[k].release();
        b->d_chain_order.release();
        b->d_evc.release();
    }
    if (b->hs_rows)
        (void)hipHostFree(b->hs_rows);
    if (b->hs_tile_row)
        (void)hipHostFree(b->hs_tile_row);
    if (b->hs_end)
        (void)hipHostFree(b->hs_end);
    if (b->stage)
        (void)hipHostFree(b->stage);
    if (b->hs_tile_x)
        (void)hipHostFree(b->hs_tile_x);
    if (b->hs_tile_nav)
        (void)hipHostFree(b->hs_tile_nav);
    b->d_iq.release();
    b->d_dig.release();
    for (auto &t : b->evs)
        for (auto &e : t.e)
            if (e)
                (void)hipEventDestroy(e);
    delete b;
}

extern "C" int gpsbb_batch_create(gpsbb_t *h, const gpsbb_chan_t *ch, int nblocks, int nch, double delt,
                                  int nsamp, unsigned flags, gpsbb_batch_t **out)
{
    if (!h || !out)
        return GPSBB_E_BADARG;
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    gpsbb_batch *b = batch_new(h);
    if (!b)
        return GPSBB_E_NOMEM;
    if (h->batches_created++ & 1) { /* batches created one after the other seed side by side */
        const hipError_t e2 = use_second_seed_stream(b);
        if (e2 != hipSuccess) {
            gpsbb_batch_destroy(b);
            h->last_hip = (int)e2;
            return e2 == hipErrorOutOfMemory ? GPSBB_E_NOMEM : GPSBB_E_HIP;
        }
    }
    int rc = batch_setup(b, ch, nblocks, nch, delt, nsamp, flags, h->s_upload);
    if (rc != GPSBB_OK) {
        gpsbb_batch_destroy(b);
        return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->s_upload));
    *out = b;
    return GPSBB_OK;
}

extern "C" size_t gpsbb_batch_iq_bytes(const gpsbb_batch_t *b)
{
    return b ? (size_t)b->nblocks * (size_t)b->nsamp * 4 : 0;
}

/* ---- seeding on the host --------------------------------------------------------------------------
 * k_seed takes as long as its longest chain (one lane walks one chain, ~2 us per row), whatever the number
 * of chains.  For a handful of blocks — the drop-in single-block call above all — a few host threads walk
 * the same chains with the same code (gpsbb_nco.h) an order of magnitude faster per row, and the tables
 * (a few MB) are uploaded instead.  Same rows, same tile index, same end states as the kernel writes. */
namespace {

constexpr size_t HOST_SEED_MAX_CHANNELS = 64; /* blocks x channels up to which the host seeds */

struct HostRowSink {
    SynRow *rows;
    uint32_t cap, cnt;
    bool overflow;
    unsigned long long dwrd_oob, itable_512;
    const uint32_t *dwrd;
    uint32_t dbit;
    int32_t *tr;
    size_t tstride;
    int32_t tile_t, ntiles, wrap_pend;

    void row(int32_t n0, uint32_t nav, double x, double S, bool after_wrap)
    {
        const int32_t nt = (int32_t)(((int64_t)n0 + TILE - 1) / TILE);
        const int32_t lim = nt < ntiles ? nt : ntiles;
        const int32_t here = (int32_t)(cnt < cap ? cnt : cap);
        for (; tile_t < lim; tile_t++, tr += tstride) {
            *tr = (here - 1) | wrap_pend;
            wrap_pend = 0;
        }
        if (after_wrap) {
            if ((n0 & (TILE - 1)) == 0 && tile_t < ntiles) {
                *tr = here | wrap_pend;
                tr += tstride;
                tile_t++;
            }
            wrap_pend = (int32_t)0x80000000;
        }
        if (cnt < cap) {
            SynRow r;
            r.n0 = n0;
            if (dwrd) {
                r.nav = nav | dbit;
                r.x = x;
                r.S = S;
            } else {
                r.nav = 0;
                r.x = mul_rn(x, 512.0);
                r.S = mul_rn(S, 512.0);
            }
            rows[cnt] = r;
        } else {
            overflow = true;
        }
        cnt++;
    }
    void table_index_512() { itable_512++; }
    void nav_fetch(uint32_t nav)
    {
        if (nav_iword(nav) >= GPSBB_N_DWRD)
            dwrd_oob++;
        dbit = nav_bit(dwrd, nav) < 0 ? 0x80000000u : 0u;
    }
    void finish()
    {
        if (cnt > cap)
            cnt = cap;
        for (; tile_t <= ntiles; tile_t++, tr += tstride) {
            *tr = ((int32_t)cnt - 1) | wrap_pend;
            wrap_pend = 0;
        }
        SynRow r;
        r.n0 = INT32_MAX;
        r.nav = 0;
        r.x = 0.0;
        r.S = 0.0;
        rows[cnt] = r;
    }
};

/* one chain (kind 0 = code, 1 = carrier) of channel k = block*nch + i: what seed_code_chain /
 * seed_carr_chain / seed_carr_fixed do on the device.  Returns false on a row-pool overflow. */
bool host_seed_chain(const gpsbb_batch *b, int kind, size_t k, unsigned long long *dwrd_oob, unsigned long long *itable_512)
{
    const gpsbb_chan_t &c = b->h_ch[k];
    gpsbb_chan_state_t &e = b->hs_end[k];
    const size_t nbc = (size_t)b->nblocks * b->nch;
    const bool fixed = (b->flags & GPSBB_FIXED_CARRIER) != 0;
    if (c.prn <= 0) {
        if (kind == 0) {
            e.code_phase = 0.0;
            e.iword = e.ibit = e.icode = e.dataBit = e.codeCA = 0;
            e._pad = 0;
        } else {
            e.carr_phase = 0.0;
        }
        return true;
    }
    const size_t blk = k / (size_t)b->nch, i = k % (size_t)b->nch;
    if (kind == 1 && fixed) {
        e.carr_phase = (double)(uint32_t)(b->h_kph0[k] + (uint32_t)b->nsamp * (uint32_t)b->h_kstep[k]);
        if (b->ev) {
            /* k_synth_pd: the table index at every tile start, in closed form (what k_tiles writes on the device) */
            double *tx = b->hs_tile_x + (blk * (2 * (size_t)b->nch) + 2 * i + 1) * (size_t)b->ntiles;
            for (int t = 0; t < b->ntiles; t++)
                tx[t] = fixed_tile_index(b->h_kph0[k], b->h_kstep[k], t);
        }
        return true;
    }
    if (b->ev) {
        /* breakpoint kernel: tile-start states instead of rows (what k_seed<true> writes) */
        uint32_t nav = kind == 0 ? nav_pack(c.icode, c.ibit, c.iword) : 0u;
        TileSink sink = make_tile_sink(b->hs_tile_x, b->hs_tile_nav, b->nch, b->ntiles, (int)blk, (int)i, kind,
                                       kind == 0 ? c.dwrd : nullptr, nav, nullptr);
        if (kind == 0) {
            const double s = mul_rn(c.f_code, b->delt);
            const double x = build_rows_f64<NCO_CODE>(c.code_phase, s, nav, b->nsamp, sink);
            sink.finish();
            e.code_phase = x;
            e.iword = nav_iword(nav);
            e.ibit = nav_ibit(nav);
            e.icode = nav_icode(nav);
            e.dataBit = nav_bit(c.dwrd, nav);
            const int ci = (int)x;
            e.codeCA = (int)((b->h->h_ca[(size_t)c.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1;
            e._pad = 0;
        } else {
            const double s = mul_rn(c.f_carr, b->delt);
            e.carr_phase = build_rows_f64<NCO_CARR>(c.carr_phase, s, nav, b->nsamp, sink);
            sink.finish();
        }
        *itable_512 += sink.hz_local[0];
        *dwrd_oob += sink.hz_local[1];
        return true;
    }
    const size_t chain = (size_t)kind * nbc + k;
    HostRowSink sink;
    sink.rows = b->hs_rows + b->row_off[chain];
    sink.cap = (uint32_t)(b->row_off[chain + 1] - b->row_off[chain] - 1);
    sink.cnt = 0;
    sink.overflow = false;
    sink.dwrd_oob = 0;
    sink.itable_512 = 0;
    sink.dwrd = kind == 0 ? c.dwrd : nullptr;
    uint32_t nav = kind == 0 ? nav_pack(c.icode, c.ibit, c.iword) : 0u;
    sink.dbit = kind == 0 && nav_bit(c.dwrd, nav) < 0 ? 0x80000000u : 0u;
    sink.tr = b->hs_tile_row + (blk * ((size_t)b->ntiles + 1)) * (2 * (size_t)b->nch) + 2 * i + (size_t)kind;
    sink.tstride = 2 * (size_t)b->nch;
    sink.tile_t = 0;
    sink.ntiles = b->ntiles;
    sink.wrap_pend = 0;
    if (kind == 0) {
        const double s = mul_rn(c.f_code, b->delt);
        const double x = build_rows_f64<NCO_CODE>(c.code_phase, s, nav, b->nsamp, sink);
        sink.finish();
        e.code_phase = x;
        e.iword = nav_iword(nav);
        e.ibit = nav_ibit(nav);
        e.icode = nav_icode(nav);
        e.dataBit = nav_bit(c.dwrd, nav);
        const int ci = (int)x;
        e.codeCA = (int)((b->h->h_ca[(size_t)c.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1;
        e._pad = 0;
    } else {
        const double s = mul_rn(c.f_carr, b->delt);
        e.carr_phase = build_rows_f64<NCO_CARR>(c.carr_phase, s, nav, b->nsamp, sink);
        sink.finish();
    }
    *dwrd_oob += sink.dwrd_oob;
    *itable_512 += sink.itable_512;
    return !sink.overflow;
}

int host_pinned_reserve(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap)
        return hipSuccess;
    if (*p)
        (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    bytes += bytes / 4;
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess)
        *cap = bytes;
    return e;
}

} /* namespace */

/* where the NCO tables of a run are built: by size (default), or as GPSBB_OPT_SEED_WHERE says (tests run both ways) */
static bool host_seeding_wanted(const gpsbb_batch *b)
{
    const bool off = GPSBB_KNOB_SET("GPSBB_DEVICE_SEED_ONLY");
    const size_t lim = (size_t)GPSBB_KNOB_LONG("GPSBB_HOST_SEED_MAX", HOST_SEED_MAX_CHANNELS);
    if (b->h->opt_seed_where)
        return b->h->opt_seed_where == 2;
    return !off && (size_t)b->nblocks * b->nch <= lim;
}

/* Build the tables of one run on host threads and queue their upload on the seeding stream. */
static int host_seed_run(gpsbb_batch *b, int set, hipStream_t stream)
{
    gpsbb *h = b->h;
    const size_t nbc = (size_t)b->nblocks * b->nch;
    const size_t tr_n = 2 * nbc * ((size_t)b->ntiles + 1);
    const size_t tx_n = 2 * nbc * (size_t)b->ntiles, tn_n = nbc * (size_t)b->ntiles;
    if (b->ev) {
        HIPCHK(h, (hipError_t)host_pinned_reserve((void **)&b->hs_tile_x, &b->hs_tx_cap, tx_n * sizeof(double)));
        HIPCHK(h, (hipError_t)host_pinned_reserve((void **)&b->hs_tile_nav, &b->hs_tn_cap, tn_n * sizeof(uint32_t)));
    } else {
        HIPCHK(h, (hipError_t)host_pinned_reserve((void **)&b->hs_rows, &b->hs_rows_cap, (b->total_rows + 4) * sizeof(SynRow)));
        HIPCHK(h, (hipError_t)host_pinned_reserve((void **)&b->hs_tile_row, &b->hs_tr_cap, tr_n * sizeof(int32_t)));
    }
    HIPCHK(h, (hipError_t)host_pinned_reserve((void **)&b->hs_end, &b->hs_end_cap, nbc * sizeof(gpsbb_chan_state_t)));
    const size_t nchains = 2 * nbc;
    if (!h->pool) {
        const unsigned hw = std::thread::hardware_concurrency();
        size_t n = hw ? hw : 4;
        n = n > 32 ? 32 : n;
        h->pool = new (std::nothrow) WorkPool(n - 1);
        if (!h->pool)
            return GPSBB_E_NOMEM;
    }
    const size_t nthr = nchains; /* one slot of results per job */
    std::vector<unsigned long long> oob(nthr, 0ull), i512(nthr, 0ull);
    std::vector<char> ok(nthr, 1);
    /* carrier chains first: they are the long ones */
    h->pool->run(nchains, [&](size_t j) {
        const int kind = j < nbc ? 1 : 0;
        if (!host_seed_chain(b, kind, j < nbc ? j : j - nbc, &oob[j], &i512[j]))
            ok[j] = 0;
    });
    for (size_t t = 0; t < nthr; t++) {
        h->host_dwrd_oob += oob[t];
        h->host_itable_512 += i512[t];
        if (!ok[t])
            return GPSBB_E_INTERNAL;
    }
    if (b->ev) {
        HIPCHK(h, hipMemcpyAsync(b->d_tile_x[set].p, b->hs_tile_x, tx_n * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(h, hipMemcpyAsync(b->d_tile_nav[set].p, b->hs_tile_nav, tn_n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(b->d_rows[set].p, b->hs_rows, b->total_rows * sizeof(SynRow), hipMemcpyHostToDevice, stream));
        HIPCHK(h, hipMemcpyAsync(b->d_tile_row[set].p, b->hs_tile_row, tr_n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    }
    HIPCHK(h, hipMemcpyAsync(b->d_end[set].p, b->hs_end, nbc * sizeof(gpsbb_chan_state_t), hipMemcpyHostToDevice, stream));
    return GPSBB_OK;
}

static BatchDev batch_dev(const gpsbb_batch *b, int set)
{
    BatchDev p;
    memset(&p, 0, sizeof p); /* (every field has a value, also the ones a later round adds) */
    p.ch = b->d_ch.p;
    p.nblocks = b->nblocks;
    p.nch = b->nch;
    p.nsamp = b->nsamp;
    p.ntiles = b->ntiles;
    p.delt = b->delt;
    p.flags = b->flags;
    p.tabs = b->h->d_tabs;
    p.ca_bits = b->h->d_ca;
    p.rows = reinterpret_cast<SynRow *>(b->d_rows[set].p);
    p.row_off = b->d_row_off.p;
    p.tile_row = b->d_tile_row[set].p;
    p.row_cnt = b->d_row_cnt[set].p;
    p.tile_ctr = b->d_tile_ctr.p + (size_t)set * ((size_t)b->nblocks + 1);
    p.kph0 = (b->flags & GPSBB_FIXED_CARRIER) ? b->d_kph0.p : nullptr;
    p.kstep = (b->flags & GPSBB_FIXED_CARRIER) ? b->d_kstep.p : nullptr;
    p.end = b->d_end[set].p;
    p.status = b->h->d_status;
    p.hazards = b->h->d_hz;
    p.digest = b->want_digest ? b->d_dig.p : nullptr;
    p.seed_order = b->d_seed_order.p;
    p.seed_lanes = (int)b->h_seed_order.size();
    p.ev = b->ev ? 1 : 0;
    int ev_chunk = b->ev_all_dense ? (int)GPSBB_KNOB_LONG("GPSBB_PD_CHUNK", PD_CHUNK) : (int)GPSBB_KNOB_LONG("GPSBB_EV_CHUNK", EV_CHUNK);
    /* a batch too small to give every CU a workgroup's worth of chunks (the drop-in call's one block: 293 tiles) hands its tiles out
     * in smaller chunks, down to one at a time: twice the workgroups, half the tiles each (0.116 -> 0.110 ms for the reference's
     * block rendered into a registered buffer) */
    const long cus = b->h->sm_count > 0 ? b->h->sm_count : 256;
    while (ev_chunk > 1 && GPSBB_KNOB_LONG("GPSBB_SMALL_CHUNKS", 1) != 0 &&
           (long)b->nblocks * (((long)b->ntiles + ev_chunk - 1) / ev_chunk) < cus * EV_WAVES)
        ev_chunk--;
    p.ev_chunk = ev_chunk < 1 ? 1 : ev_chunk;
    p.pd_danger = (uint32_t)GPSBB_KNOB_LONG("GPSBB_PD_DANGER", 2u * PD_BAND); /* (larger: more lanes take the exact path; a test aid) */
    p.tile_x = b->d_tile_x[set].p;
    p.tile_nav = b->d_tile_nav[set].p;
    p.evc = b->d_evc.p;
    p.chain_dev = b->chain_dev ? 1 : 0;
    p.chain_starts = b->chain_starts ? 1 : 0;
    p.aux = b->chain_dev ? b->d_aux[set].p : nullptr;
    p.nseg = b->nseg;
    p.seg_tiles = b->seg_tiles;
    p.nvb = b->nblocks * b->nseg;
    p.fix_end = b->d_fix_end.p ? b->d_fix_end.p + (size_t)set * GPSBB_MAX_CHAN * b->fix_chunks : nullptr;
    p.fix_flag = b->d_fix_flag.p ? b->d_fix_flag.p + (size_t)set * GPSBB_MAX_CHAN * b->fix_chunks : nullptr;
    p.fix_epoch = b->fix_epoch;
    p.fix_chunks = b->fix_chunks;
    p.model_start = b->chain_model ? 1 : 0;
    p.cd = b->chain_dev ? b->d_cd.p : nullptr;
    p.start0 = b->chain_dev ? b->d_start0.p : nullptr;
    p.prefix_rows = b->chain_dev && !b->chain_starts ? b->d_prefix[set].p : nullptr;
    p.carry = b->chain_dev ? b->d_carry : nullptr;
    p.cont0_mask = b->cont0_mask;
    p.lap_end = nullptr;
    return p;
}

static LapDev lap_dev(const gpsbb_batch *b, int set)
{
    LapDev L;
    L.bc = b->d_lap_bc[set].p;
    L.lane0 = b->d_lap_lane0[set].p;
    L.nlaps = b->d_lap_cnt[set].p;
    L.nbad = b->d_lap_cnt[set].p + 2 * GPSBB_MAX_CHAN;
    L.rec = b->d_lap_rec[set].p;
    L.agg = b->d_lap_agg[set].p;
    L.chunk_m = b->d_lap_chunk_m[set].p;
    L.chunk_bad = b->d_lap_chunk_bad[set].p;
    memcpy(L.chunk0, b->lap_chunk0, sizeof L.chunk0);
    L.chained = b->chain_dev ? 1 : 0;
    L.jitter = (uint32_t)GPSBB_KNOB_LONG("GPSBB_LAP_JITTER", 0);
    L.burst = GPSBB_KNOB_SET("GPSBB_LAP_NO_BURST") ? 0 : (int)GPSBB_KNOB_LONG("GPSBB_LAP_BURST_SHARE", LAP_BURST_SHARE);
    L.unit[NCO_CODE] = lap_unit(NCO_CODE, (size_t)b->nblocks * b->nch);
    L.unit[NCO_CARR] = lap_unit(NCO_CARR, (size_t)b->nblocks * b->nch);
    return L;
}

static int batch_launch(gpsbb_batch *b, int16_t *d_iq)
{
    gpsbb *h = b->h;
    const int set = (int)(b->run_count % (unsigned)b->nsets);
    b->fix_epoch++; /* a number no earlier launch of this batch handed to k_chain_fix_par */
    if (b->want_digest)
        HIPCHK(h, (hipError_t)b->d_dig.reserve((size_t)b->nblocks));
    const BatchDev p = batch_dev(b, set);
    const int lanes = (int)b->h_seed_order.size();
    /* (the drop-in call's scratch batch, everything on one stream and waited for before the call returns: no events — each record
     * is a packet between two kernels, 5 us of nothing on a call of 140) */
    const bool timed = !b->one_stream;
    if (!timed && b->evs.empty()) {
        gpsbb_batch::Ev4 t = {{nullptr, nullptr, nullptr, nullptr}};
        b->evs.push_back(t);
    }
    if (!timed) {
        b->ev_used = 0;
    } else if (b->ev_used == b->evs.size()) {
        if (b->evs.size() >= 4096) {
            b->ev_used = 0; /* wrap: only the most recent runs are kept */
        } else {
            gpsbb_batch::Ev4 t = {{nullptr, nullptr, nullptr, nullptr}};
            for (auto &e : t.e)
                HIPCHK(h, hipEventCreate(&e));
            b->evs.push_back(t);
        }
    }
    hipEvent_t *ev = timed ? b->evs[b->ev_used++].e : b->evs[0].e;
    PUSH_MARK("l_ev");

    /* The pre-pass runs on a seeding stream of its own: it may start as soon as the synthesis kernel that last
     * read this table set has finished, i.e. it overlaps the synthesis of the runs before it.  With three sets
     * consecutive runs take the handle's two seeding streams in turn, so that two pre-passes are in flight. */
    hipStream_t ss = b->one_stream ? h->s_compute : b->seed_stream;
    if (b->one_stream) {
    } else if (b->nsets > 2) {
        const unsigned base = b->seed_stream == h->s_seed ? 0u : 1u;
        HIPCHK(h, seed_stream_at(h, (base + b->run_count) % (unsigned)(b->nsets - 1), &ss));
    } else if (b->d_carry && b->chain_dev) {
        /* a stream's slot (one table set): consecutive pushes take the seeding streams in turn, so that as many
         * pre-passes are in flight (a pre-pass is a chain of latency-bound kernels: ~12 ms whatever the size of the
         * push, and the ring delivers one push per (that / streams)) */
        const unsigned nseed = (unsigned)GPSBB_KNOB_LONG("GPSBB_STREAM_SEED_STREAMS", STREAM_SEED_STREAMS);
        HIPCHK(h, seed_stream_at(h, b->stream_turn % (nseed >= 1 && nseed <= (unsigned)SEED_STREAMS_MAX ? nseed : STREAM_SEED_STREAMS), &ss));
    }
    if (b->upload_done)
        HIPCHK(h, hipStreamWaitEvent(ss, b->upload_done, 0));
    if (b->synth_pending[set])
        HIPCHK(h, hipStreamWaitEvent(ss, b->synth_done_ref[set], 0));
    if (timed)
        HIPCHK(h, hipEventRecord(ev[0], ss));
    bool ctr_reset_by_prepass = false;
    if (h->opt_skip_seed && b->run_count >= (unsigned)b->nsets) {
        /* measurement hook: time the synthesis kernel alone on tables already built */
    } else if (b->host_seed) {
        /* the previous user of the pinned images (this batch's last run) has been copied out: its upload was
         * followed by the synthesis kernel, which synth_done[] of that set covers */
        const int prev = (set + b->nsets - 1) % b->nsets;
        if (b->synth_pending[prev])
            HIPCHK(h, hipEventSynchronize(b->synth_done_ref[prev]));
        const int rc = host_seed_run(b, set, ss);
        if (rc != GPSBB_OK)
            return rc;
    } else if (b->ev && b->laps) {
        /* the lap-parallel pre-pass (gpsbb_laps.hip.h): plan, reference walks, scan, true walks, repair — the code chains
         * first (nothing of theirs waits for another push), then the carriers: a stream's push starts from the exact phase the
         * push before it left on the device, so its plan follows that push's repair kernel */
        const LapDev L = lap_dev(b, set);
        const unsigned cc = b->lap_chunk0[NCO_CODE][b->nch] - b->lap_chunk0[NCO_CODE][0];
        const unsigned ck = b->lap_chunk0[NCO_CARR][b->nch] - b->lap_chunk0[NCO_CARR][0];
        /* a batch that continues nothing (no stream carry: the drop-in call's block, resident batches): both kinds in one grid
         * per step, five launches instead of ten — the chain of small launches IS the latency of a small batch's pre-pass (one
         * block of the reference's geometry: 98 -> 50 us), and a big batch's two plan kernels (16 wavefronts each, 0.1 - 0.2 ms)
         * run side by side.  A stream's pushes keep the kinds apart: their code chains wait for nobody, their carriers for the
         * push before (GPSBB_LAP_MERGE=1, experiments build: merged there too). */
        /* pass 2 writes the tile states in 32- / 16-byte pieces where the geometry has them (several tiles per lap: a code period is
         * 1023 chips / (1.023e6 * delt) samples) — k_lap_pass2<., true>; at the reference's 2.6 MS/s a period is 2.5 tiles and the
         * plain loop is the faster one */
        const bool wide = GPSBB_KNOB_LONG("GPSBB_LAP_WIDE", b->delt <= 1.0 / 8.0e6 ? 1 : 0) != 0;
        const bool merged = !p.kph0 && (!(b->d_carry && b->ev_fix) || GPSBB_KNOB_LONG("GPSBB_LAP_MERGE", 0) == 1) &&
                            GPSBB_KNOB_LONG("GPSBB_LAP_MERGE", 0) != 2;
        /* a stream's carry (ChainCarryDev) is read by this push's carrier plan and written by its repair (exact_end AND approx_end):
         * ordered behind every writer of the push before — lap-parallel or row walks — and ahead of every reader of the next, whichever
         * pre-pass that push takes: both events waited for, both recorded */
        const auto carry_wait = [&]() -> hipError_t {
            hipError_t e = hipSuccess;
            if (b->d_carry && b->ev_fix)
                e = hipStreamWaitEvent(ss, b->ev_fix, 0);
            if (e == hipSuccess && b->d_carry && b->ev_prefix)
                e = hipStreamWaitEvent(ss, b->ev_prefix, 0);
            return e;
        };
        const auto carry_done = [&]() -> hipError_t {
            hipError_t e = hipSuccess;
            if (b->d_carry && b->ev_fix)
                e = hipEventRecord(b->ev_fix, ss);
            if (e == hipSuccess && b->d_carry && b->ev_prefix)
                e = hipEventRecord(b->ev_prefix, ss);
            return e;
        };
        if (merged) {
            HIPCHK(h, carry_wait());
            hipLaunchKernelGGL(k_lap_plan2, dim3(2 * b->nch), dim3(64), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_pass1_2, dim3(cc + ck), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_scan2, dim3(2 * b->nch), dim3(64), 0, ss, p, L);
            if (wide)
                hipLaunchKernelGGL(k_lap_pass2_2<true>, dim3(cc + ck), dim3(LAP_WG), 0, ss, p, L);
            else
                hipLaunchKernelGGL(k_lap_pass2_2<false>, dim3(cc + ck), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_repair2, dim3(2 * b->nch), dim3(64), 0, ss, p, L);
            HIPCHK(h, carry_done());
        } else {
            hipLaunchKernelGGL(k_lap_plan<NCO_CODE>, dim3(b->nch), dim3(64), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_pass1<NCO_CODE>, dim3(cc), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_scan<NCO_CODE>, dim3(b->nch), dim3(64), 0, ss, p, L);
            if (wide)
                hipLaunchKernelGGL((k_lap_pass2<NCO_CODE, true>), dim3(cc), dim3(LAP_WG), 0, ss, p, L);
            else
                hipLaunchKernelGGL((k_lap_pass2<NCO_CODE, false>), dim3(cc), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_repair<NCO_CODE>, dim3(b->nch), dim3(64), 0, ss, p, L);
            if (p.kph0) {
                /* fixed-point carrier: no chain to walk; the plan kernel leaves the end states, the tile states are a closed form */
                hipLaunchKernelGGL(k_lap_plan<NCO_CARR>, dim3(b->nch), dim3(64), 0, ss, p, L);
                hipLaunchKernelGGL(k_lap_fixed_tiles, dim3(b->nblocks * b->nch), dim3(256), 0, ss, p);
            } else {
                HIPCHK(h, carry_wait());
                hipLaunchKernelGGL(k_lap_plan<NCO_CARR>, dim3(b->nch), dim3(64), 0, ss, p, L);
                hipLaunchKernelGGL(k_lap_pass1<NCO_CARR>, dim3(ck), dim3(LAP_WG), 0, ss, p, L);
                hipLaunchKernelGGL(k_lap_scan<NCO_CARR>, dim3(b->nch), dim3(64), 0, ss, p, L);
                if (wide)
                    hipLaunchKernelGGL((k_lap_pass2<NCO_CARR, true>), dim3(ck), dim3(LAP_WG), 0, ss, p, L);
                else
                    hipLaunchKernelGGL((k_lap_pass2<NCO_CARR, false>), dim3(ck), dim3(LAP_WG), 0, ss, p, L);
                hipLaunchKernelGGL(k_lap_repair<NCO_CARR>, dim3(b->nch), dim3(64), 0, ss, p, L);
                HIPCHK(h, carry_done());
            }
        }
        ctr_reset_by_prepass = true; /* k_lap_plan zeroes the set's tile counters */
    } else if (b->ev) {
        const bool old_seed = GPSBB_KNOB_SET("GPSBB_EV_KSEED"); /* experiment: the one-kernel pre-pass */
        if (old_seed) {
            hipLaunchKernelGGL(k_seed<true>, dim3((lanes + GPSBB_SEED_WG - 1) / GPSBB_SEED_WG), dim3(GPSBB_SEED_WG), 0, ss, p);
        } else {
            const dim3 wg_all((lanes + GPSBB_WALK_WG - 1) / GPSBB_WALK_WG);
            if (b->chain_dev) {
                if (!b->chain_model) {
                    BatchDev pa = p; /* pass A: the carrier chains only (they come first in the plan) */
                    pa.seed_lanes = b->carr_lanes;
                    hipLaunchKernelGGL(k_walk<1>, dim3((b->carr_lanes + GPSBB_WALK_WG - 1) / GPSBB_WALK_WG), dim3(GPSBB_WALK_WG), 0, ss, pa);
                    /* a stream: this push's prefix / fix-up follow the ones of the push before (other seeding stream) */
                    if (b->d_carry && b->ev_prefix)
                        HIPCHK(h, hipStreamWaitEvent(ss, b->ev_prefix, 0));
                    hipLaunchKernelGGL(k_chain_prefix, dim3(b->nch), dim3(PREFIX_WG), 0, ss, p);
                    if (b->d_carry && b->ev_prefix)
                        HIPCHK(h, hipEventRecord(b->ev_prefix, ss));
                } /* else: pass B walks from the host's drift model of every segment's start (batch_setup) */
                hipLaunchKernelGGL(k_walk<2>, wg_all, dim3(GPSBB_WALK_WG), 0, ss, p);
                if (b->d_carry && b->ev_fix)
                    HIPCHK(h, hipStreamWaitEvent(ss, b->ev_fix, 0));
                if (b->chain_fix_seq)
                    hipLaunchKernelGGL(k_chain_fix, dim3(1), dim3(64), 0, ss, p);
                else
                    if (b->fix_wg == FIXP_WG_ALONE)
                        hipLaunchKernelGGL(k_chain_fix_par<FIXP_WG_ALONE>, dim3(b->nch, b->fix_chunks), dim3(FIXP_WG_ALONE), 0, ss, p);
                    else
                        hipLaunchKernelGGL(k_chain_fix_par<FIXP_WG_BATCH>, dim3(b->nch, b->fix_chunks), dim3(FIXP_WG_BATCH), 0, ss, p);
                if (b->d_carry && b->ev_fix)
                    HIPCHK(h, hipEventRecord(b->ev_fix, ss));
            } else {
                hipLaunchKernelGGL(k_walk<0>, wg_all, dim3(GPSBB_WALK_WG), 0, ss, p);
            }
            hipLaunchKernelGGL(k_tiles, dim3((1 + b->nseg) * b->nblocks * b->nch), dim3(GPSBB_TILES_WG), 0, ss, p);
            ctr_reset_by_prepass = true; /* k_tiles zeroes the set's tile counters */
        }
    } else {
        if (b->chain_starts) {
            /* the carrier chained on the device for the per-sample kernel: pass A, prefix, pass B without rows and the
             * fix-up put the exact start phase of every block into its descriptor; k_seed then sees independent blocks */
            BatchDev pc = p;
            pc.seed_order = b->d_chain_order.p;
            pc.seed_lanes = b->chain_lanes;
            const dim3 wg_c((b->chain_lanes + GPSBB_WALK_WG - 1) / GPSBB_WALK_WG);
            hipLaunchKernelGGL(k_walk<1>, wg_c, dim3(GPSBB_WALK_WG), 0, ss, pc);
            if (b->d_carry && b->ev_prefix)
                HIPCHK(h, hipStreamWaitEvent(ss, b->ev_prefix, 0));
            hipLaunchKernelGGL(k_chain_prefix, dim3(b->nch), dim3(PREFIX_WG), 0, ss, pc);
            if (b->d_carry && b->ev_prefix)
                HIPCHK(h, hipEventRecord(b->ev_prefix, ss));
            hipLaunchKernelGGL(k_walk<3>, wg_c, dim3(GPSBB_WALK_WG), 0, ss, pc);
            if (b->d_carry && b->ev_fix)
                HIPCHK(h, hipStreamWaitEvent(ss, b->ev_fix, 0));
            if (b->chain_fix_seq)
                hipLaunchKernelGGL(k_chain_fix, dim3(1), dim3(64), 0, ss, pc);
            else
                if (b->fix_wg == FIXP_WG_ALONE)
                    hipLaunchKernelGGL(k_chain_fix_par<FIXP_WG_ALONE>, dim3(b->nch, b->fix_chunks), dim3(FIXP_WG_ALONE), 0, ss, pc);
                else
                    hipLaunchKernelGGL(k_chain_fix_par<FIXP_WG_BATCH>, dim3(b->nch, b->fix_chunks), dim3(FIXP_WG_BATCH), 0, ss, pc);
            if (b->d_carry && b->ev_fix)
                HIPCHK(h, hipEventRecord(b->ev_fix, ss));
        }
        hipLaunchKernelGGL(k_seed<false>, dim3((lanes + GPSBB_SEED_WG - 1) / GPSBB_SEED_WG), dim3(GPSBB_SEED_WG), 0, ss, p);
    }
    HIPCHK(h, hipGetLastError());
    if (timed)
        HIPCHK(h, hipEventRecord(ev[1], ss));
    h->last_prepass = b->host_seed ? 2 : (b->ev && b->laps ? 3 : 1);
    PUSH_MARK("l_pre");

    /* off by default: +2 % on a stream of pushes, but overlapping kernels make the per-launch time (the roofline figure)
     * meaningless and re-runs of a resident batch get slower */
    const bool one_cs = !GPSBB_KNOB_SET("GPSBB_TWO_COMPUTE_STREAMS");
    /* consecutive launches take the two synthesis streams in turn — they work on different table sets (or, slots of
     * a ring, different batches) — except re-runs of a batch that has a single table set */
    hipStream_t sc = h->s_compute;
    if (!one_cs && !b->one_stream && (b->nsets >= 2 || b->max_sets == 1) && ((h->compute_turn++) & 1u))
        sc = h->s_compute2;
    b->last_cs = sc;
    if (sc != ss)
        HIPCHK(h, hipStreamWaitEvent(sc, ev[1], 0));
    if (!ctr_reset_by_prepass)
        HIPCHK(h, hipMemsetAsync(p.tile_ctr, 0, ((size_t)b->nblocks + 1) * sizeof(int32_t), sc));
    if (timed)
        HIPCHK(h, hipEventRecord(ev[2], sc));
    if (b->want_digest)
        HIPCHK(h, hipMemsetAsync(b->d_dig.p, 0, (size_t)b->nblocks * sizeof(unsigned long long), sc));
    bool digest_fused = false;
    if (b->ev) {
        /* One workgroup of EV_WG lanes fits a CU (its LDS image takes ~140 - 156 KB).  Grid = the blocks' primaries, then the
         * helpers (ev_pick_block: a helper joins one of the blocks that still have tiles to hand out, chosen when it starts):
         * as many as can be useful when there are few blocks (never more workgroups per block than there are chunks of tiles
         * per wavefront), else enough to keep every CU busy through the end of the launch — the last round of blocks and then
         * what is left of it take the CUs twice over. */
        const long wg_slots = (long)(h->sm_count > 0 ? h->sm_count : 256);
        const long chunks = ((long)b->ntiles + p.ev_chunk - 1) / p.ev_chunk;
        const long max_useful = (chunks + EV_WAVES - 1) / EV_WAVES;
        const long oversub = GPSBB_KNOB_LONG("GPSBB_EV_HELPERS", 2);
        long helpers = (long)b->nblocks * (max_useful - 1);
        if (helpers > wg_slots * oversub)
            helpers = wg_slots * oversub;
        if (helpers < 0)
            helpers = 0;
        const dim3 grid((unsigned)(b->nblocks + helpers));
        if (b->ev_all_dense && b->want_digest) {
            if (b->nch <= PD_WIDE_CHAN)
                hipLaunchKernelGGL((k_synth_pd<true, true>), grid, dim3(EV_WG), sizeof(PdLds<true>) + EV_PICK_LDS, sc, p, d_iq);
            else
                hipLaunchKernelGGL((k_synth_pd<false, true>), grid, dim3(EV_WG), sizeof(PdLds<false>) + EV_PICK_LDS, sc, p, d_iq);
            digest_fused = true;
        } else if (b->ev_all_dense && b->nch <= PD_WIDE_CHAN)
            hipLaunchKernelGGL(k_synth_pd<true>, grid, dim3(EV_WG), sizeof(PdLds<true>) + EV_PICK_LDS, sc, p, d_iq);
        else if (b->ev_all_dense)
            hipLaunchKernelGGL(k_synth_pd<false>, grid, dim3(EV_WG), sizeof(PdLds<false>) + EV_PICK_LDS, sc, p, d_iq);
        else if (b->ev_dense)
            hipLaunchKernelGGL(k_synth_ev_dense, grid, dim3(EV_WG), sizeof(EvLds) + EV_PICK_LDS, sc, p, d_iq);
        else if (p.kph0)
            hipLaunchKernelGGL(k_synth_ev_fixed, grid, dim3(EV_WG), sizeof(EvLdsLean) + EV_PICK_LDS, sc, p, d_iq);
        else if (b->want_digest) {
            hipLaunchKernelGGL(k_synth_ev_digest, grid, dim3(EV_WG), sizeof(EvLdsLean) + EV_PICK_LDS, sc, p, d_iq);
            digest_fused = true;
        } else
            hipLaunchKernelGGL(k_synth_ev, grid, dim3(EV_WG), sizeof(EvLdsLean) + EV_PICK_LDS, sc, p, d_iq);
        h->last_kernel = 2;
        h->last_chain_dev = b->chain_dev && !b->chain_indep ? 1 : 0;
    } else {
        h->last_kernel = 1;
        h->last_chain_dev = b->chain_dev && !b->chain_indep ? 1 : 0;
        /* Workgroups per block: enough of them to oversubscribe the chip ~3x (tiles are handed out
         * dynamically in chunks, so the tail is short), never more than there are chunks; the per-block
         * LDS tables (amplitude LUT, chips, nav words) are then built few times per block. */
        const long wg_slots = (long)(h->sm_count > 0 ? h->sm_count : 256) * 2;
        const long chunks = ((long)b->ntiles + TILE_CHUNK - 1) / TILE_CHUNK;
        const long max_useful = (chunks + WAVES_PER_WG - 1) / WAVES_PER_WG;
        const long oversub = GPSBB_KNOB_LONG("GPSBB_OVERSUB", 12);
        long want = (wg_slots * oversub + b->nblocks - 1) / b->nblocks;
        want = want < 1 ? 1 : (want > max_useful ? max_useful : want);
        const int gx = (int)want;
        hipLaunchKernelGGL(k_synth, dim3(gx, b->nblocks), dim3(TILE_THREADS), sizeof(SynthLds), sc, p, d_iq);
    }
    if (b->want_digest && !digest_fused) {
        /* a synthesis kernel without a digesting variant: the blocks read back behind it, on its stream */
        long chunks = (2048 + b->nblocks - 1) / b->nblocks;
        const long max_chunks = ((long)b->nsamp + 1023) / 1024;
        chunks = chunks > max_chunks ? max_chunks : (chunks < 1 ? 1 : chunks);
        hipLaunchKernelGGL(k_block_digest, dim3((unsigned)chunks, (unsigned)b->nblocks), dim3(256), 0, sc, (const uint32_t *)d_iq, b->nsamp, b->d_dig.p);
    }
    HIPCHK(h, hipGetLastError());
    if (timed)
        HIPCHK(h, hipEventRecord(ev[3], sc));
    /* the run's end-of-synthesis event doubles as "this table set is free again" and as what a stream's copy stream
     * waits for: every further record on the synthesis stream is another packet between two kernels */
    b->synth_done_ref[set] = ev[3];
    b->last_done = ev[3];
    b->synth_pending[set] = timed;
    b->last_set = set;
    b->run_count++;
    b->ran = true;
    return GPSBB_OK;
}

extern "C" int gpsbb_batch_run(gpsbb_batch_t *b, int16_t *d_iq)
{
    if (!b)
        return GPSBB_E_BADARG;
    gpsbb *h = b->h;
    HIPCHK(h, hipSetDevice(h->device));
    if (!d_iq) {
        HIPCHK(h, (hipError_t)b->d_iq.reserve((size_t)b->nblocks * b->nsamp * 2));
        d_iq = b->d_iq.p;
        b->last_iq = d_iq;
    } else {
        b->last_iq = nullptr;
        b->last_ext_iq = d_iq; /* gpsbb_batch_read copies from it for as long as the caller keeps it alive */
    }
    return batch_launch(b, d_iq);
}

extern "C" int16_t *gpsbb_batch_device_iq(gpsbb_batch_t *b) { return b ? b->last_iq : nullptr; }

extern "C" int gpsbb_sync(gpsbb_t *h)
{
    if (!h)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->s_upload));
    HIPCHK(h, hipStreamSynchronize(h->s_seed));
    for (hipStream_t st : h->s_more)
        if (st)
            HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipStreamSynchronize(h->s_compute));
    HIPCHK(h, hipStreamSynchronize(h->s_compute2));
    HIPCHK(h, hipStreamSynchronize(h->s_copy));
    uint32_t st = 0;
    HIPCHK(h, hipMemcpy(&st, h->d_status, 4, hipMemcpyDeviceToHost));
    if (st) {
        HIPCHK(h, zero_now(h, h->d_status, 4));
        return GPSBB_E_INTERNAL;
    }
    return GPSBB_OK;
}

extern "C" int gpsbb_batch_read(gpsbb_batch_t *b, int16_t *iq_out, gpsbb_chan_state_t *end_state)
{
    if (!b || !b->ran)
        return GPSBB_E_STATE;
    gpsbb *h = b->h;
    HIPCHK(h, hipSetDevice(h->device));
    if (iq_out) {
        const int16_t *src = b->last_iq ? b->last_iq : b->last_ext_iq;
        if (!src)
            return GPSBB_E_STATE;
        HIPCHK(h, hipMemcpy(iq_out, src, gpsbb_batch_iq_bytes(b), hipMemcpyDeviceToHost));
    }
    if (end_state)
        HIPCHK(h, hipMemcpy(end_state, b->d_end[b->last_set].p, (size_t)b->nblocks * b->nch * sizeof(gpsbb_chan_state_t),
                            hipMemcpyDeviceToHost));
    return GPSBB_OK;
}

extern "C" int gpsbb_device_read(gpsbb_t *h, void *host_dst, const void *device_src, size_t bytes)
{
    if (!h || !host_dst || !device_src)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
    return GPSBB_OK;
}

extern "C" int gpsbb_device_digest(gpsbb_t *h, const int16_t *d_iq, long nblocks, int nsamp, uint64_t *digest_out)
{
    if (!h || !d_iq || !digest_out || nblocks < 1 || nblocks > 65535 || nsamp < 1)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    /* the work of every launch the caller may be reading the output of is on this handle's streams: wait for it, then digest on
     * the synthesis stream (ordered behind everything the handle rendered) */
    const int rc = gpsbb_sync(h);
    if (rc != GPSBB_OK)
        return rc;
    HIPCHK(h, (hipError_t)h->d_digest.reserve((size_t)nblocks));
    HIPCHK(h, hipMemsetAsync(h->d_digest.p, 0, (size_t)nblocks * sizeof(unsigned long long), h->s_compute));
    /* enough workgroups per block to fill the chip however few blocks there are, never pieces below 4 KB */
    long chunks = (2048 + nblocks - 1) / nblocks;
    const long max_chunks = ((long)nsamp + 1023) / 1024;
    chunks = chunks > max_chunks ? max_chunks : (chunks < 1 ? 1 : chunks);
    hipLaunchKernelGGL(k_block_digest, dim3((unsigned)chunks, (unsigned)nblocks), dim3(256), 0, h->s_compute, (const uint32_t *)d_iq, nsamp, h->d_digest.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(digest_out, h->d_digest.p, (size_t)nblocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->s_compute));
    HIPCHK(h, hipStreamSynchronize(h->s_compute));
    return GPSBB_OK;
}

extern "C" int gpsbb_slot_digest(gpsbb_t *h, const int16_t *d_iq, long nblocks, int nsamp, uint64_t *digest_out)
{
    if (!h || !d_iq || !digest_out || nblocks < 1 || nblocks > 65535 || nsamp < 1)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->s_digest)
        HIPCHK(h, hipStreamCreateWithFlags(&h->s_digest, hipStreamNonBlocking));
    hipStream_t cs = h->s_digest;
    HIPCHK(h, (hipError_t)h->d_digest.reserve((size_t)nblocks));
    HIPCHK(h, hipMemsetAsync(h->d_digest.p, 0, (size_t)nblocks * sizeof(unsigned long long), cs));
    long chunks = (GPSBB_KNOB_LONG("GPSBB_SLOT_DIGEST_WGS", 2048) + nblocks - 1) / nblocks;
    const long max_chunks = ((long)nsamp + 1023) / 1024;
    chunks = chunks > max_chunks ? max_chunks : (chunks < 1 ? 1 : chunks);
    hipLaunchKernelGGL(k_block_digest, dim3((unsigned)chunks, (unsigned)nblocks), dim3(256), 0, cs, (const uint32_t *)d_iq, nsamp, h->d_digest.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(digest_out, h->d_digest.p, (size_t)nblocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    HIPCHK(h, hipStreamSynchronize(cs));
    return GPSBB_OK;
}

extern "C" int gpsbb_get_hazards(gpsbb_t *h, gpsbb_hazards_t *out, int reset)
{
    if (!h || !out)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long v[2];
    HIPCHK(h, hipMemcpy(v, h->d_hz, 16, hipMemcpyDeviceToHost));
    out->itable_512 = v[0];
    out->itable_512 += h->host_itable_512;
    out->dwrd_oob = v[1] + h->host_dwrd_oob;
    if (reset) {
        HIPCHK(h, zero_now(h, h->d_hz, 16));
        h->host_dwrd_oob = 0;
        h->host_itable_512 = 0;
    }
    return GPSBB_OK;
}

extern "C" int gpsbb_batch_last_timing(gpsbb_batch_t *b, float *ms_seed, float *ms_synth, float *ms_total)
{
    if (!b || !b->ran || b->ev_used == 0)
        return GPSBB_E_STATE;
    gpsbb *h = b->h;
    hipEvent_t *ev = b->evs[b->ev_used - 1].e;
    float a = 0, c = 0, t = 0;
    HIPCHK(h, hipEventElapsedTime(&a, ev[0], ev[1]));
    HIPCHK(h, hipEventElapsedTime(&c, ev[2], ev[3]));
    HIPCHK(h, hipEventElapsedTime(&t, ev[0], ev[3]));
    if (ms_seed) *ms_seed = a;
    if (ms_synth) *ms_synth = c;
    if (ms_total) *ms_total = t;
    return GPSBB_OK;
}

extern "C" int gpsbb_batch_timing_stats(gpsbb_batch_t *b, int *nruns, float *ms_seed_sum, float *ms_synth_sum,
                                        float *ms_total_sum, int reset)
{
    if (!b)
        return GPSBB_E_BADARG;
    gpsbb *h = b->h;
    float sa = 0, sc = 0, stt = 0;
    for (size_t k = 0; k < b->ev_used; k++) {
        hipEvent_t *ev = b->evs[k].e;
        float a = 0, c = 0, t = 0;
        HIPCHK(h, hipEventElapsedTime(&a, ev[0], ev[1]));
        HIPCHK(h, hipEventElapsedTime(&c, ev[2], ev[3]));
        HIPCHK(h, hipEventElapsedTime(&t, ev[0], ev[3]));
        sa += a;
        sc += c;
        stt += t;
    }
    if (nruns) *nruns = (int)b->ev_used;
    if (ms_seed_sum) *ms_seed_sum = sa;
    if (ms_synth_sum) *ms_synth_sum = sc;
    if (ms_total_sum) *ms_total_sum = stt;
    if (reset)
        b->ev_used = 0;
    return GPSBB_OK;
}

extern "C" int gpsbb_fill_ceiling(gpsbb_t *h, void *d_dst, size_t bytes, int iters, float *ms)
{
    if (!h || !d_dst || bytes < 16 || iters < 1 || !ms)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    const int grid = h->sm_count > 0 ? h->sm_count * 8 : 2048;
    hipLaunchKernelGGL(k_fill_ceiling, dim3(grid), dim3(256), 0, h->s_compute, (uint4 *)d_dst, n16, 1u);
    HIPCHK(h, hipEventRecord(e0, h->s_compute));
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(k_fill_ceiling, dim3(grid), dim3(256), 0, h->s_compute, (uint4 *)d_dst, n16, (uint32_t)i);
    HIPCHK(h, hipEventRecord(e1, h->s_compute));
    HIPCHK(h, hipEventSynchronize(e1));
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return GPSBB_OK;
}

/* ---- the synchronous single-block surface ------------------------------------------------------------ */

/* What the drop-in call waits for.  Everything of the call is ordered in front of the synthesis stream's tail (upload -> pre-pass
 * -> synthesis by events), so ONE stream is waited for, not the handle's ten; the end states and the status word come back
 * through pinned memory behind the IQ on that stream instead of as two blocking copies of their own (each a round trip of
 * 25 - 30 us: with the ten-kernel pre-pass they were half of the call's 0.26 ms for the reference's block). */
/* end states and status word into the handle's pinned page, written by the device: one launch behind the synthesis kernel
 * instead of two copies (each a packet of its own and 4 us of blit kernel) */
__global__ void k_fill_tail(const gpsbb_chan_state_t *end, const uint32_t *status, unsigned char *host, int nch, uint32_t *host_status)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(end);
    uint32_t *dst = reinterpret_cast<uint32_t *>(host);
    const uint32_t words = end ? (uint32_t)nch * (uint32_t)(sizeof(gpsbb_chan_state_t) / 4) : 0u;
    for (uint32_t k = threadIdx.x; k < words; k += blockDim.x)
        dst[k] = src[k];
    if (threadIdx.x == 0)
        *host_status = *status;
}

static int fill_block_finish(gpsbb_t *h, gpsbb_batch *b, int nch, int nsamp, int16_t *iq_out, gpsbb_chan_state_t *end_state)
{
    const size_t end_bytes = (size_t)GPSBB_MAX_CHAN * sizeof(gpsbb_chan_state_t);
    if (!h->h_fill)
        HIPCHK(h, hipHostMalloc((void **)&h->h_fill, end_bytes + 64, hipHostMallocDefault));
    hipStream_t cs = b->last_cs;
    uint32_t *st = reinterpret_cast<uint32_t *>(h->h_fill + end_bytes);
    *st = 0xffffffffu;
    if (GPSBB_KNOB_LONG("GPSBB_FILL_TAIL_KERNEL", 1) != 0) {
        static_assert(sizeof(gpsbb_chan_state_t) % 4 == 0, "copied as 32-bit words");
        hipLaunchKernelGGL(k_fill_tail, dim3(1), dim3(256), 0, cs, end_state ? b->d_end[b->last_set].p : nullptr, h->d_status, h->h_fill, nch, st);
        HIPCHK(h, hipGetLastError());
    } else {
        if (end_state)
            HIPCHK(h, hipMemcpyAsync(h->h_fill, b->d_end[b->last_set].p, (size_t)nch * sizeof(gpsbb_chan_state_t), hipMemcpyDeviceToHost, cs));
        HIPCHK(h, hipMemcpyAsync(st, h->d_status, 4, hipMemcpyDeviceToHost, cs));
    }
    PUSH_MARK("tail");
    /* (iq_out null: the synthesis kernel wrote into the caller's registered buffer.)  A destination that lies partly in pages a
     * registration pinned is one the runtime's copy refuses (hipErrorInvalidValue): through a pinned buffer of the handle's then */
    bool bounce = false;
    if (iq_out) {
        const uintptr_t page = 4096, a = (uintptr_t)iq_out, e = a + (size_t)nsamp * 4;
        for (const gpsbb::HostReg &r : h->host_regs) {
            const uintptr_t ra = (uintptr_t)r.host & ~(page - 1), re = ((uintptr_t)r.host + r.bytes + page - 1) & ~(page - 1);
            bounce = bounce || (a < re && ra < e);
        }
    }
    if (bounce) {
        if (h->bounce_cap < (size_t)nsamp * 4) {
            if (h->h_bounce)
                (void)hipHostFree(h->h_bounce);
            h->h_bounce = nullptr;
            h->bounce_cap = 0;
            HIPCHK(h, hipHostMalloc((void **)&h->h_bounce, (size_t)nsamp * 4, hipHostMallocDefault));
            h->bounce_cap = (size_t)nsamp * 4;
        }
        HIPCHK(h, hipMemcpyAsync(h->h_bounce, b->last_iq, (size_t)nsamp * 4, hipMemcpyDeviceToHost, cs));
    } else if (iq_out) {
        HIPCHK(h, hipMemcpyAsync(iq_out, b->last_iq, (size_t)nsamp * 4, hipMemcpyDeviceToHost, cs));
    }
    PUSH_MARK("iq copy");
    HIPCHK(h, hipStreamSynchronize(cs));
    if (bounce)
        memcpy(iq_out, h->h_bounce, (size_t)nsamp * 4);
    if (end_state)
        memcpy(end_state, h->h_fill, (size_t)nch * sizeof(gpsbb_chan_state_t));
    if (*st) {
        HIPCHK(h, zero_now(h, h->d_status, 4));
        return GPSBB_E_INTERNAL;
    }
    return GPSBB_OK;
}

extern "C" int gpsbb_host_register(gpsbb_t *h, void *ptr, size_t bytes)
{
    if (!h || !ptr || !bytes)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    char *q = static_cast<char *>(ptr);
    for (const gpsbb::HostReg &r : h->host_regs)
        if (q < r.host + r.bytes && r.host < q + bytes)
            return GPSBB_E_STATE; /* overlaps a range that is registered already */
    HIPCHK(h, hipHostRegister(ptr, bytes, hipHostRegisterMapped));
    void *dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess || !dev) {
        (void)hipHostUnregister(ptr);
        h->last_hip = (int)e;
        return GPSBB_E_HIP;
    }
    h->host_regs.push_back({q, static_cast<char *>(dev), bytes});
    return GPSBB_OK;
}

extern "C" int gpsbb_host_unregister(gpsbb_t *h, void *ptr)
{
    if (!h || !ptr)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    for (size_t k = 0; k < h->host_regs.size(); k++)
        if (h->host_regs[k].host == static_cast<char *>(ptr)) {
            HIPCHK(h, hipStreamSynchronize(h->s_compute)); /* nothing of a fill is in flight once the call has returned; make sure */
            HIPCHK(h, hipHostUnregister(ptr));
            h->host_regs.erase(h->host_regs.begin() + (long)k);
            return GPSBB_OK;
        }
    return GPSBB_E_STATE;
}

extern "C" int gpsbb_fill_block(gpsbb_t *h, const gpsbb_chan_t *ch, int nch, double delt, int nsamp,
                                int16_t *iq_out, gpsbb_chan_state_t *end_state)
{
    return gpsbb_fill_block_ex(h, ch, nch, delt, nsamp, 0u, iq_out, end_state);
}

extern "C" int gpsbb_fill_block_ex(gpsbb_t *h, const gpsbb_chan_t *ch, int nch, double delt, int nsamp, unsigned flags,
                                   int16_t *iq_out, gpsbb_chan_state_t *end_state)
{
    if (!h || !ch || !iq_out)
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->scratch) {
        h->scratch = batch_new(h);
        if (!h->scratch)
            return GPSBB_E_NOMEM;
    }
    gpsbb_batch *b = h->scratch;
    if (flags & ~GPSBB_FIXED_CARRIER & ~GPSBB_CHAIN_CARRIER)
        return GPSBB_E_BADARG;
    g_push_trace.start();
    b->one_stream = GPSBB_KNOB_LONG("GPSBB_FILL_ONE_STREAM", 1) != 0;
    int rc = batch_setup(b, ch, 1, nch, delt, nsamp, flags & GPSBB_FIXED_CARRIER, b->one_stream ? h->s_compute : h->s_seed);
    if (rc != GPSBB_OK)
        return rc;
    PUSH_MARK("set-up");
    /* iq_out inside a range the caller registered (gpsbb_host_register): the synthesis kernel's stores go there over the bus
     * while it computes, and there is no copy to wait for afterwards */
    int16_t *direct = nullptr;
    for (const gpsbb::HostReg &r : h->host_regs) {
        const char *q = reinterpret_cast<const char *>(iq_out);
        if (q >= r.host && (size_t)(q - r.host) <= r.bytes && (size_t)nsamp * 4 <= r.bytes - (size_t)(q - r.host)) {
            direct = reinterpret_cast<int16_t *>(r.dev + (q - r.host));
            break;
        }
    }
    rc = gpsbb_batch_run(b, direct);
    if (rc != GPSBB_OK)
        return rc;
    PUSH_MARK("launches");
    rc = fill_block_finish(h, b, nch, nsamp, direct ? nullptr : iq_out, end_state);
    g_push_trace.end();
    return rc;
}

/* the reference's own channel_t[] / gain[] in, rendered, updated in place as its loop leaves them; fixed: the build without
 * FLOAT_CARR_PHASE (h:160-161: a 32-bit accumulator and its step instead of the double) */
static int fill_block_ref(gpsbb_t *h, void *chan, const gpsbb_refchan_layout_t *L, bool fixed, size_t off_step, int max_chan,
                          const double *gain, double delt, int nsamp, int16_t *iq_buff)
{
    if (!h || !chan || !L || !gain || !iq_buff || max_chan < 1 || max_chan > GPSBB_MAX_CHAN ||
        (L->sizeof_dwrd_elem != 4 && L->sizeof_dwrd_elem != 8))
        return GPSBB_E_BADARG;
    gpsbb_chan_t d[GPSBB_MAX_CHAN];
    gpsbb_chan_state_t st[GPSBB_MAX_CHAN];
    char *base = static_cast<char *>(chan);
    auto ld_i = [](const char *p) { int v; memcpy(&v, p, sizeof v); return v; };
    auto ld_u = [](const char *p) { unsigned v; memcpy(&v, p, sizeof v); return v; };
    auto ld_d = [](const char *p) { double v; memcpy(&v, p, sizeof v); return v; };
    for (int i = 0; i < max_chan; i++) {
        const char *c = base + (size_t)i * L->stride;
        memset(&d[i], 0, sizeof d[i]);
        d[i].prn = ld_i(c + L->off_prn);
        if (d[i].prn <= 0) {
            d[i].prn = 0;
            continue;
        }
        d[i].f_carr = ld_d(c + L->off_f_carr);
        d[i].f_code = ld_d(c + L->off_f_code);
        d[i].carr_phase = fixed ? (double)ld_u(c + L->off_carr_phase) : ld_d(c + L->off_carr_phase);
        d[i].code_phase = ld_d(c + L->off_code_phase);
        d[i].iword = ld_i(c + L->off_iword);
        d[i].ibit = ld_i(c + L->off_ibit);
        d[i].icode = ld_i(c + L->off_icode);
        d[i].gain = gain[i];
        for (int k = 0; k < GPSBB_N_DWRD; k++) {
            uint64_t w = 0;
            memcpy(&w, c + L->off_dwrd + (size_t)k * L->sizeof_dwrd_elem, L->sizeof_dwrd_elem);
            d[i].dwrd[k] = (uint32_t)w;
        }
        if (fixed && off_step != (size_t)-1 && std::isfinite(d[i].f_carr) && std::fabs(d[i].f_carr * delt) <= 0.125) {
            /* the step the host computed (c:2675) must be the one the accumulator is advanced by here (c:2748): a host that
             * put anything else into carr_phasestep would get different samples from its own loop */
            const volatile double scaled = 512.0 * 65536.0 * d[i].f_carr * delt;
            if (ld_i(c + off_step) != (int)std::round(scaled))
                return GPSBB_E_BADCHAN;
        }
    }
    int rc = gpsbb_fill_block_ex(h, d, max_chan, delt, nsamp, fixed ? GPSBB_FIXED_CARRIER : 0u, iq_buff, st);
    if (rc != GPSBB_OK)
        return rc;
    for (int i = 0; i < max_chan; i++) {
        if (d[i].prn <= 0)
            continue;
        char *c = base + (size_t)i * L->stride;
        if (fixed) {
            const unsigned ph = (unsigned)st[i].carr_phase; /* the accumulator's value, an integer below 2^32 */
            memcpy(c + L->off_carr_phase, &ph, 4);
        } else {
            memcpy(c + L->off_carr_phase, &st[i].carr_phase, 8);
        }
        memcpy(c + L->off_code_phase, &st[i].code_phase, 8);
        memcpy(c + L->off_iword, &st[i].iword, 4);
        memcpy(c + L->off_ibit, &st[i].ibit, 4);
        memcpy(c + L->off_icode, &st[i].icode, 4);
        memcpy(c + L->off_dataBit, &st[i].dataBit, 4);
        memcpy(c + L->off_codeCA, &st[i].codeCA, 4);
    }
    return GPSBB_OK;
}

extern "C" int gpsbb_fill_block_ref(gpsbb_t *h, void *chan, const gpsbb_refchan_layout_t *L, int max_chan,
                                    const double *gain, double delt, int nsamp, int16_t *iq_buff)
{
    return fill_block_ref(h, chan, L, false, (size_t)-1, max_chan, gain, delt, nsamp, iq_buff);
}

extern "C" int gpsbb_fill_block_ref_fixed(gpsbb_t *h, void *chan, const gpsbb_refchan_layout_t *L, size_t off_carr_phasestep,
                                          int max_chan, const double *gain, double delt, int nsamp, int16_t *iq_buff)
{
    return fill_block_ref(h, chan, L, true, off_carr_phasestep, max_chan, gain, delt, nsamp, iq_buff);
}

/* ================================================================================================== */
/* time-sharded streaming with pinned host gather                                                     */
/* ================================================================================================== */

struct gpsbb_stream {
    gpsbb *h = nullptr;
    int nch = 0, nsamp = 0, bps = 0, depth = 0;
    double delt = 0.0;
    unsigned flags = 0;
    struct Slot {
        gpsbb_batch *batch = nullptr;
        int16_t *h_iq = nullptr;            /* pinned */
        gpsbb_chan_state_t *h_end = nullptr; /* pinned */
        unsigned long long *h_dig = nullptr; /* pinned, on the first GPSBB_PUSH_DIGEST: the push's block digests */
        bool has_dig = false;
        hipEvent_t computed = nullptr, copied = nullptr;
    };
    std::vector<Slot> slots;
    uint64_t head = 0, tail = 0; /* pushes / pops so far */
    ChainCarry *carry = nullptr;               /* IEEE carrier chained on the host across pushes */
    /* ... or on the device (gpsbb_walk.hip.h): the exact phase never leaves it */
    ChainCarryDev *d_carry = nullptr;
    hipEvent_t ev_prefix = nullptr, ev_fix = nullptr;
    int last_prn[GPSBB_MAX_CHAN] = {0};        /* prn per channel in the last block pushed */
    double rough_phase[GPSBB_MAX_CHAN] = {0};  /* the host's rough idea of the carrier phase after it */
    bool carry_on_device = false;              /* where the authoritative carry is right now */
    std::vector<gpsbb_chan_t> seeded;
    std::vector<double> seeds;
    int fx_prn[GPSBB_MAX_CHAN] = {0};          /* fixed-point carrier: channel state after the last push */
    uint32_t fx_phase[GPSBB_MAX_CHAN] = {0};
    bool poisoned = false; /* a push failed after part of it had been enqueued: the chain's state on the device has
                              moved on without the host's; nothing more can be pushed or popped */
};

extern "C" void gpsbb_stream_destroy(gpsbb_stream_t *s)
{
    if (!s)
        return;
    (void)hipSetDevice(s->h->device);
    (void)hipStreamSynchronize(s->h->s_seed);
    for (hipStream_t st : s->h->s_more)
        if (st)
            (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(s->h->s_compute);
    (void)hipStreamSynchronize(s->h->s_compute2);
    (void)hipStreamSynchronize(s->h->s_copy);
    delete s->carry;
    if (s->d_carry)
        (void)hipFree(s->d_carry);
    if (s->ev_prefix)
        (void)hipEventDestroy(s->ev_prefix);
    if (s->ev_fix)
        (void)hipEventDestroy(s->ev_fix);
    for (auto &sl : s->slots) {
        if (sl.batch) gpsbb_batch_destroy(sl.batch);
        if (sl.h_iq) (void)hipHostFree(sl.h_iq);
        if (sl.h_dig) (void)hipHostFree(sl.h_dig);
        if (sl.h_end) (void)hipHostFree(sl.h_end);
        if (sl.computed) (void)hipEventDestroy(sl.computed);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
    }
    delete s;
}

extern "C" int gpsbb_stream_create(gpsbb_t *h, int nch, double delt, int nsamp, int blocks_per_slot,
                                   int depth, unsigned flags, gpsbb_stream_t **out)
{
    if (!h || !out || nch < 1 || nch > GPSBB_MAX_CHAN || nsamp < 1 || blocks_per_slot < 1 || depth < 2 ||
        depth > 64 || !(delt > 0.0) || (flags & ~(GPSBB_CHAIN_CARRIER | GPSBB_FIXED_CARRIER | GPSBB_STREAM_DEVICE_ONLY)))
        return GPSBB_E_BADARG;
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    gpsbb_stream *s = new (std::nothrow) gpsbb_stream;
    if (!s)
        return GPSBB_E_NOMEM;
    s->h = h;
    s->nch = nch;
    s->nsamp = nsamp;
    s->bps = blocks_per_slot;
    s->depth = depth;
    s->delt = delt;
    s->flags = flags;
    s->slots.resize(depth);
    const size_t iq_bytes = (size_t)blocks_per_slot * nsamp * 4;
    const size_t end_bytes = (size_t)blocks_per_slot * nch * sizeof(gpsbb_chan_state_t);
    for (auto &sl : s->slots) {
        sl.batch = batch_new(h);
        hipError_t e = sl.batch ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess && ((&sl - &s->slots[0]) & 1))
            e = use_second_seed_stream(sl.batch);
        if (e == hipSuccess) sl.batch->max_sets = 1; /* consecutive pushes use different slots: one table set each */
        if (e == hipSuccess) e = (hipError_t)sl.batch->d_iq.reserve(iq_bytes / 2);
        if (e == hipSuccess && !(flags & GPSBB_STREAM_DEVICE_ONLY))
            e = hipHostMalloc((void **)&sl.h_iq, iq_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_end, end_bytes + 32, hipHostMallocDefault); /* + the status word */
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming);
        if (e != hipSuccess) {
            h->last_hip = (int)e;
            gpsbb_stream_destroy(s);
            return e == hipErrorOutOfMemory ? GPSBB_E_NOMEM : GPSBB_E_HIP;
        }
    }
    *out = s;
    return GPSBB_OK;
}

extern "C" int gpsbb_stream_pending(const gpsbb_stream_t *s) { return s ? (int)(s->head - s->tail) : 0; }

extern "C" int gpsbb_stream_reset(gpsbb_stream_t *s)
{
    if (!s)
        return GPSBB_E_BADARG;
    if (s->poisoned || s->head != s->tail)
        return GPSBB_E_STATE;
    gpsbb *h = s->h;
    HIPCHK(h, hipSetDevice(h->device));
    /* everything the last stream left in the queues has completed (its slots were popped), but the chain kernels of its
     * last push may still be writing the carry: wait for the handle before touching it */
    const int rc = gpsbb_sync(h);
    if (rc != GPSBB_OK)
        return rc;
    if (s->d_carry)
        HIPCHK(h, zero_now(h, s->d_carry, sizeof(ChainCarryDev)));
    if (s->carry)
        memset(s->carry, 0, sizeof *s->carry);
    s->carry_on_device = false;
    for (int i = 0; i < GPSBB_MAX_CHAN; i++) {
        s->last_prn[i] = 0;
        s->rough_phase[i] = 0.0;
        s->fx_prn[i] = 0;
        s->fx_phase[i] = 0;
    }
    s->head = s->tail = 0;
    return GPSBB_OK;
}

extern "C" int gpsbb_stream_timing_stats(gpsbb_stream_t *s, int *nruns, float *ms_seed_sum, float *ms_synth_sum, int reset)
{
    if (!s)
        return GPSBB_E_BADARG;
    int n = 0;
    float a = 0, c = 0;
    for (auto &sl : s->slots) {
        int k = 0;
        float x = 0, y = 0, z = 0;
        const int rc = gpsbb_batch_timing_stats(sl.batch, &k, &x, &y, &z, reset);
        if (rc != GPSBB_OK)
            return rc;
        n += k;
        a += x;
        c += y;
    }
    if (nruns) *nruns = n;
    if (ms_seed_sum) *ms_seed_sum = a;
    if (ms_synth_sum) *ms_synth_sum = c;
    return GPSBB_OK;
}

static int stream_push(gpsbb_stream_t *s, const gpsbb_chan_t *ch, bool new_chain, bool want_digest);

extern "C" int gpsbb_stream_push(gpsbb_stream_t *s, const gpsbb_chan_t *ch) { return stream_push(s, ch, false, false); }

extern "C" int gpsbb_stream_push_ex(gpsbb_stream_t *s, const gpsbb_chan_t *ch, unsigned flags)
{
    if (flags & ~(GPSBB_PUSH_NEW_CHAIN | GPSBB_PUSH_DIGEST))
        return GPSBB_E_BADARG;
    return stream_push(s, ch, (flags & GPSBB_PUSH_NEW_CHAIN) != 0, (flags & GPSBB_PUSH_DIGEST) != 0);
}

static int stream_push(gpsbb_stream_t *s, const gpsbb_chan_t *ch, bool new_chain, bool want_digest)
{
    if (!s || !ch)
        return GPSBB_E_BADARG;
    if (s->poisoned || s->head - s->tail >= (uint64_t)s->depth)
        return GPSBB_E_STATE; /* ring full: pop first */
    if (new_chain) {
        /* this push does not continue the one before: every channel of its first block starts from its descriptor's phase,
         * as if it had just been allocated (c:1956-1964) — "no satellite was here before" is all the chain has to be told */
        for (int i = 0; i < GPSBB_MAX_CHAN; i++) {
            s->last_prn[i] = 0;
            s->fx_prn[i] = 0;
            if (s->carry)
                s->carry->prn[i] = 0;
        }
    }
    gpsbb *h = s->h;
    g_push_trace.start();
    HIPCHK(h, hipSetDevice(h->device));
    auto &sl = s->slots[s->head % s->depth];
    gpsbb_batch *b = sl.batch;
    /* Blocks consecutive in time with the IEEE carrier: the carrier phase carries over from block to block and from
     * push to push, exactly.  Where that is resolved is decided per push below (on the device: gpsbb_walk.hip.h). */
    unsigned run_flags = s->flags & (GPSBB_CHAIN_CARRIER | GPSBB_FIXED_CARRIER);
    const size_t nbc = (size_t)s->bps * s->nch;
    b->d_carry = nullptr;
    /* the host's chaining state only moves on once every enqueue of this push has succeeded */
    ChainCarry carry_next;
    bool carry_host = false;
    double rough_next[GPSBB_MAX_CHAN]; /* taken from s->rough_phase below, once the carry is where this push wants it */
    int fx_prn_next[GPSBB_MAX_CHAN];
    uint32_t fx_phase_next[GPSBB_MAX_CHAN];
    memcpy(fx_prn_next, s->fx_prn, sizeof fx_prn_next);
    memcpy(fx_phase_next, s->fx_phase, sizeof fx_phase_next);
    if ((s->flags & GPSBB_CHAIN_CARRIER) && !(s->flags & GPSBB_FIXED_CARRIER)) {
        for (size_t k = 0; k < nbc; k++)
            if (!chan_ok(ch[k], s->delt))
                return GPSBB_E_BADCHAN;
        /* Where the carrier is chained: on the device wherever the pre-pass runs there (exactly, in parallel over the
         * blocks, the phase carried from push to push in device memory; for either synthesis kernel), on host
         * threads — sequential per channel — for pushes small enough to be seeded on the host.  A stream may
         * change sides between pushes: the carry then moves across, which costs a synchronisation. */
        const size_t host_lim = (size_t)GPSBB_KNOB_LONG("GPSBB_HOST_SEED_MAX", HOST_SEED_MAX_CHANNELS);
        const bool dev_only = GPSBB_KNOB_SET("GPSBB_DEVICE_SEED_ONLY");
        /* (small pushes too where the lap-parallel pre-pass will take them: it costs less than the host threads) */
        /* (... which it only does for blocks the model kernels render: the kernel plan's own test, here, before the push is promised
         * the device-side chain — a small push of a 1 MS/s stream, which they decline, stays with the host threads instead of the
         * row walks' milliseconds) */
        bool small_on_dev = nbc <= host_lim && h->opt_seed_where == 0 && h->opt_synth_kernel != 1 && h->opt_chain_where == 0 &&
                            !GPSBB_KNOB_SET("GPSBB_NO_LAPS") && lap_eligible(ch, nbc, s->delt, false);
        if (small_on_dev) {
            std::vector<EvConst> probe;
            small_on_dev = ev_plan(ch, s->bps, s->nch, s->delt, probe);
        }
        const bool dev = h->opt_chain_where != 1 &&
                         (h->opt_seed_where == 1 || h->opt_seed_where == 3 || (h->opt_seed_where == 0 && (dev_only || nbc > host_lim || small_on_dev)));
        if (!s->carry) {
            s->carry = new (std::nothrow) ChainCarry();
            if (!s->carry)
                return GPSBB_E_NOMEM;
            memset(s->carry, 0, sizeof *s->carry);
        }
        if (dev) {
            if (!s->d_carry) {
                HIPCHK(h, hipMalloc((void **)&s->d_carry, sizeof(ChainCarryDev)));
                HIPCHK(h, zero_now(h, s->d_carry, sizeof(ChainCarryDev)));
                HIPCHK(h, hipEventCreateWithFlags(&s->ev_prefix, hipEventDisableTiming));
                HIPCHK(h, hipEventCreateWithFlags(&s->ev_fix, hipEventDisableTiming));
            }
            if (!s->carry_on_device && s->head > 0) {
                /* host -> device: the exact phases as both the prediction and the truth */
                ChainCarryDev c;
                for (int i = 0; i < GPSBB_MAX_CHAN; i++)
                    c.approx_end[i] = c.exact_end[i] = s->carry->phase[i];
                const int rc_ = gpsbb_sync(h);
                if (rc_ != GPSBB_OK)
                    return rc_;
                HIPCHK(h, hipMemcpy(s->d_carry, &c, sizeof c, hipMemcpyHostToDevice));
                HIPCHK(h, hipStreamSynchronize(nullptr)); /* (see zero_now) */
                for (int i = 0; i < s->nch; i++) {
                    s->last_prn[i] = s->carry->prn[i];
                    s->rough_phase[i] = s->carry->phase[i];
                }
            }
            s->carry_on_device = true;
            memcpy(rough_next, s->rough_phase, sizeof rough_next);
            b->d_carry = s->d_carry;
            b->carry_prn = s->last_prn;
            b->carry_phase = rough_next;
            b->ev_prefix = s->ev_prefix;
            b->ev_fix = s->ev_fix;
            b->stream_turn = (unsigned)s->head;
        } else {
            if (s->carry_on_device) {
                /* device -> host */
                ChainCarryDev c;
                const int rc_ = gpsbb_sync(h);
                if (rc_ != GPSBB_OK)
                    return rc_;
                HIPCHK(h, hipMemcpy(&c, s->d_carry, sizeof c, hipMemcpyDeviceToHost));
                for (int i = 0; i < s->nch; i++) {
                    s->carry->prn[i] = s->last_prn[i];
                    s->carry->phase[i] = c.exact_end[i];
                }
                s->carry_on_device = false;
            }
            s->seeded.assign(ch, ch + nbc);
            s->seeds.resize(nbc);
            carry_next = *s->carry; /* committed only when the push has been enqueued in full */
            carry_host = true;
            chain_carrier_host(ch, s->bps, s->nch, s->delt, s->nsamp, s->seeds.data(), 0, &carry_next);
            for (size_t k = 0; k < nbc; k++)
                s->seeded[k].carr_phase = s->seeds[k];
            ch = s->seeded.data();
            run_flags &= ~GPSBB_CHAIN_CARRIER;
        }
    }
    if (!b->d_carry)
        memcpy(rough_next, s->rough_phase, sizeof rough_next);
    /* the slot's previous D2H copy was waited for by the pop that freed it */
    const bool fx_chain = (s->flags & GPSBB_FIXED_CARRIER) && (s->flags & GPSBB_CHAIN_CARRIER) && s->head > 0;
    b->fixed_prev_prn = fx_chain ? s->fx_prn : nullptr;
    b->fixed_prev_phase = fx_chain ? s->fx_phase : nullptr;
    PUSH_MARK("plan");
    int rc = batch_setup(b, ch, s->bps, s->nch, s->delt, s->nsamp, run_flags, h->s_upload);
    PUSH_MARK("setup");
    b->fixed_prev_prn = nullptr;
    b->fixed_prev_phase = nullptr;
    if (rc != GPSBB_OK) {
        b->d_carry = nullptr;
        return rc;
    }
    if (b->d_carry && !b->chain_dev) { /* the push was promised a device-side chain */
        b->d_carry = nullptr;
        return GPSBB_E_INTERNAL;
    }
    if (s->flags & GPSBB_FIXED_CARRIER)
        for (int i = 0; i < s->nch; i++) {
            const size_t k = (size_t)(s->bps - 1) * s->nch + i;
            fx_prn_next[i] = ch[k].prn > 0 ? ch[k].prn : 0;
            fx_phase_next[i] = b->h_kph0[k] + (uint32_t)s->nsamp * (uint32_t)b->h_kstep[k];
        }
    b->last_iq = b->d_iq.p;
    /* From here on kernels of this push may be in the queues (with the device-side chain they advance the carry in
     * device memory): a failure leaves the host's and the device's view of the stream apart, so the stream is closed
     * instead of letting a retry chain from the wrong phase. */
    struct Poison {
        gpsbb_stream *s;
        bool armed = true;
        ~Poison() { if (armed) s->poisoned = true; }
    } poison{s};
    b->want_digest = want_digest;
    if (want_digest && !sl.h_dig)
        HIPCHK(h, hipHostMalloc((void **)&sl.h_dig, (size_t)s->bps * sizeof(unsigned long long), hipHostMallocDefault));
    sl.has_dig = false;
    rc = batch_launch(b, b->d_iq.p);
    PUSH_MARK("launch");
    b->d_carry = nullptr;
    if (rc != GPSBB_OK)
        return rc;
    PUSH_MARK("rec");
    /* gather on the side stream: pinned, asynchronous, overlaps the next push's kernels */
    hipStream_t cs = h->s_copy;
    HIPCHK(h, hipStreamWaitEvent(cs, b->last_done, 0));
    PUSH_MARK("wait");
    if (sl.h_iq) {
        const bool sdma = GPSBB_KNOB_SET("GPSBB_GATHER_SDMA"); /* experiment: the runtime's copy instead */
        const int gwg = (int)GPSBB_KNOB_LONG("GPSBB_GATHER_WGS", 32);
        if (sdma) {
            HIPCHK(h, hipMemcpyAsync(sl.h_iq, b->d_iq.p, (size_t)s->bps * s->nsamp * 4, hipMemcpyDeviceToHost, cs));
        } else {
            hipLaunchKernelGGL(k_gather_to_host, dim3(gwg), dim3(256), 0, cs, (const gather_u32x4 *)b->d_iq.p, (gather_u32x4 *)sl.h_iq,
                               ((size_t)s->bps * s->nsamp * 4 + 15) / 16);
            HIPCHK(h, hipGetLastError());
        }
    }
    PUSH_MARK("iq");
    {
        /* end states (40 B each: a multiple of 16 bytes for any even count; odd counts are rounded up into the
         * allocation's slack) + the self-check word, by a small kernel: see k_end_states_to_host */
        const size_t bytes = (size_t)s->bps * s->nch * sizeof(gpsbb_chan_state_t);
        hipLaunchKernelGGL(k_end_states_to_host, dim3(64), dim3(256), 0, cs, (const uint4 *)b->d_end[b->last_set].p,
                           (uint4 *)sl.h_end, (bytes + 15) / 16, h->d_status, (uint32_t *)((char *)sl.h_end + ((bytes + 15) & ~(size_t)15)));
        HIPCHK(h, hipGetLastError());
    }
    if (want_digest) {
        HIPCHK(h, hipMemcpyAsync(sl.h_dig, b->d_dig.p, (size_t)s->bps * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
        sl.has_dig = true;
    }
    PUSH_MARK("endst");
    HIPCHK(h, hipEventRecord(sl.copied, cs));
    /* commit */
    if (carry_host)
        *s->carry = carry_next;
    if (s->carry_on_device && (s->flags & GPSBB_CHAIN_CARRIER) && !(s->flags & GPSBB_FIXED_CARRIER))
        for (int i = 0; i < s->nch; i++) {
            const int prn = ch[(size_t)(s->bps - 1) * s->nch + i].prn;
            s->last_prn[i] = prn > 0 ? prn : 0;
        }
    memcpy(s->rough_phase, rough_next, sizeof rough_next);
    memcpy(s->fx_prn, fx_prn_next, sizeof fx_prn_next);
    memcpy(s->fx_phase, fx_phase_next, sizeof fx_phase_next);
    s->head++;
    poison.armed = false;
    g_push_trace.end();
    return GPSBB_OK;
}

extern "C" int gpsbb_stream_pop(gpsbb_stream_t *s, const int16_t **iq, gpsbb_chan_state_t *end_state)
{
    return gpsbb_stream_pop_digest(s, iq, end_state, nullptr);
}

extern "C" int gpsbb_stream_pop_digest(gpsbb_stream_t *s, const int16_t **iq, gpsbb_chan_state_t *end_state, uint64_t *digests)
{
    if (!s || !iq)
        return GPSBB_E_BADARG;
    if (s->poisoned || s->head == s->tail)
        return GPSBB_E_STATE;
    if (digests && !s->slots[s->tail % s->depth].has_dig)
        return GPSBB_E_STATE; /* the slot was not pushed with GPSBB_PUSH_DIGEST */
    gpsbb *h = s->h;
    HIPCHK(h, hipSetDevice(h->device));
    auto &sl = s->slots[s->tail % s->depth];
    HIPCHK(h, hipEventSynchronize(sl.copied));
    *iq = sl.h_iq ? sl.h_iq : sl.batch->d_iq.p; /* GPSBB_STREAM_DEVICE_ONLY: the slot's buffer in HBM */
    if (end_state)
        memcpy(end_state, sl.h_end, (size_t)s->bps * s->nch * sizeof(gpsbb_chan_state_t));
    if (digests)
        memcpy(digests, sl.h_dig, (size_t)s->bps * sizeof(unsigned long long));
    s->tail++;
    uint32_t st = 0;
    memcpy(&st, (const char *)sl.h_end + (((size_t)s->bps * s->nch * sizeof(gpsbb_chan_state_t) + 15) & ~(size_t)15), 4);
    if (st) {
        HIPCHK(h, zero_now(h, h->d_status, 4));
        return GPSBB_E_INTERNAL;
    }
    return GPSBB_OK;
}

/* ================================================================================================== */
/* host helpers                                                                                       */
/* ================================================================================================== */

extern "C" int gpsbb_device_affinity(int device, int *numa_node, char *cpulist, size_t cpulist_cap)
{
    if (device < 0)
        return GPSBB_E_BADARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count)
        return GPSBB_E_NODEVICE;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess)
        return GPSBB_E_HIP;
    for (char *c = bdf; *c; c++)
        *c = (char)tolower((unsigned char)*c); /* sysfs spells the address in lower case */
    if (numa_node)
        *numa_node = -1;
    if (cpulist && cpulist_cap)
        cpulist[0] = 0;
    char path[160];
    if (numa_node) {
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (FILE *f = fopen(path, "r")) {
            int v = -1;
            if (fscanf(f, "%d", &v) == 1)
                *numa_node = v;
            fclose(f);
        }
    }
    if (cpulist && cpulist_cap) {
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
        if (FILE *f = fopen(path, "r")) {
            if (fgets(cpulist, (int)cpulist_cap, f)) {
                size_t n = strlen(cpulist);
                while (n && (cpulist[n - 1] == '\n' || cpulist[n - 1] == ' '))
                    cpulist[--n] = 0;
            } else {
                cpulist[0] = 0;
            }
            fclose(f);
        }
    }
    return GPSBB_OK;
}

static void chain_carrier_host(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp, double *seed,
                               int nthreads, ChainCarry *carry)
{
    auto work = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            int prev_prn = carry ? carry->prn[i] : 0;
            double prev_x = carry ? carry->phase[i] : 0.0;
            for (int b = 0; b < nblocks; b++) {
                const gpsbb_chan_t &c = ch[(size_t)b * nch + i];
                double x0 = (c.prn > 0 && c.prn == prev_prn) ? prev_x : c.carr_phase;
                seed[(size_t)b * nch + i] = c.prn > 0 ? x0 : 0.0;
                if (c.prn > 0) {
                    volatile double s = c.f_carr * delt; /* rounded on its own, as in the loop */
                    prev_x = carr_jump(x0, s, nsamp);
                }
                prev_prn = c.prn > 0 ? c.prn : 0;
            }
            if (carry) {
                carry->prn[i] = prev_prn;
                carry->phase[i] = prev_x;
            }
        }
    };
    if (nthreads <= 0 || nthreads > nch)
        nthreads = nch;
    if (nthreads == 1) {
        work(0, nch);
    } else {
        /* a thread that cannot be started (no exception may cross the C boundary) leaves its share to the caller */
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) {
            const int i0 = (int)((long)nch * t / nthreads), i1 = (int)((long)nch * (t + 1) / nthreads);
            try {
                th.emplace_back(work, i0, i1);
            } catch (...) {
                work(i0, i1);
            }
        }
        for (auto &t : th)
            t.join();
    }
}

extern "C" int gpsbb_chain_carrier_host(const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                                        double *seed, int nthreads)
{
    if (!ch || !seed || nblocks < 1 || nch < 1 || nch > GPSBB_MAX_CHAN || nsamp < 1 || !(delt > 0.0))
        return GPSBB_E_BADARG;
    for (size_t k = 0; k < (size_t)nblocks * nch; k++)
        if (!chan_ok(ch[k], delt))
            return GPSBB_E_BADCHAN;
    chain_carrier_host(ch, nblocks, nch, delt, nsamp, seed, nthreads, nullptr);
    return GPSBB_OK;
}

/* ---- the carrier chain alone, on the device ---------------------------------------------------------- */

/* Device scratch of gpsbb_chain_carrier: 24-byte chain descriptors, rough start phases, the chain's per-block record
 * and the carry from one sub-batch to the next.  Nothing of a synthesis batch (rows, tile states, IQ) exists here. */
struct ChainOnly {
    DevBuf<ChainDesc> d_cd;
    DevBuf<double> d_start0;
    DevBuf<ChainAux> d_aux;
    ChainCarryDev *d_carry = nullptr;
    uint32_t *d_status = nullptr; /* a self-check word of its own: the handle's may belong to a push still in flight */
    DevBuf<unsigned long long> d_fix_end;
    DevBuf<int> d_fix_flag;
    bool fix_flags_zeroed = false;
    int fix_epoch = 0;
    std::vector<ChainDesc> h_cd;
    std::vector<double> h_start0;
    /* the lap-parallel chain (gpsbb_laps.hip.h with nothing to emit): the phase after every block, scratch of the lap kernels */
    DevBuf<double> d_lap_end;
    DevBuf<LapBC> d_lap_bc;
    DevBuf<uint32_t> d_lap_lane0, d_lap_cnt, d_lap_chunk_bad;
    DevBuf<LapRec> d_lap_rec;
    DevBuf<LapAgg> d_lap_agg;
    DevBuf<double> d_lap_chunk_m;
    unsigned long long *d_hz_scratch = nullptr; /* (a chain alone counts no hazards: the render of those blocks does) */
    std::vector<double> h_lap_end;
};
constexpr int CHAIN_ONLY_BLOCKS = 16384;
constexpr double CHAIN_ONLY_LAPS = 6.0e6; /* laps per sub-batch of the lap-parallel chain (40 bytes of scratch each) */ /* blocks per sub-batch: 168 MB of ChainAux at 16 channels */

static void chain_only_free(gpsbb *h)
{
    ChainOnly *c = h->chain_only;
    if (!c)
        return;
    c->d_cd.release();
    c->d_start0.release();
    c->d_aux.release();
    c->d_lap_end.release();
    c->d_lap_bc.release();
    c->d_lap_lane0.release();
    c->d_lap_cnt.release();
    c->d_lap_chunk_bad.release();
    c->d_lap_rec.release();
    c->d_lap_agg.release();
    c->d_lap_chunk_m.release();
    if (c->d_hz_scratch)
        (void)hipFree(c->d_hz_scratch);
    c->d_fix_end.release();
    c->d_fix_flag.release();
    if (c->d_carry)
        (void)hipFree(c->d_carry);
    if (c->d_status)
        (void)hipFree(c->d_status);
    delete c;
    h->chain_only = nullptr;
}

extern "C" int gpsbb_chain_carrier(gpsbb_t *h, const gpsbb_chan_t *ch, int nblocks, int nch, double delt, int nsamp,
                                   double *carr_phase_seed, double *carr_phase_end)
{
    if (!h || !ch || nblocks < 1 || nch < 1 || nch > GPSBB_MAX_CHAN || nsamp < 1 || !(delt > 0.0) || !std::isfinite(delt))
        return GPSBB_E_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->chain_only) {
        h->chain_only = new (std::nothrow) ChainOnly;
        if (!h->chain_only)
            return GPSBB_E_NOMEM;
    }
    ChainOnly *c = h->chain_only;
    if (!c->d_carry)
        HIPCHK(h, hipMalloc((void **)&c->d_carry, sizeof(ChainCarryDev)));
    if (!c->d_status)
        HIPCHK(h, hipMalloc((void **)&c->d_status, 4));
    HIPCHK(h, hipMemsetAsync(c->d_carry, 0, sizeof(ChainCarryDev), h->s_seed));
    HIPCHK(h, hipMemsetAsync(c->d_status, 0, 4, h->s_seed));
    /* what the chain reads of a descriptor, and the rough start phases (plain double arithmetic; pass A takes it from
     * there): one host thread per channel, the blocks in order */
    const size_t nbc_all = (size_t)nblocks * nch;
    c->h_cd.resize(nbc_all);
    c->h_start0.resize(nbc_all);
    std::vector<char> bad(nch, 0);
    auto work = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            double x = 0.0;
            int prev_prn = 0;
            for (int blk = 0; blk < nblocks; blk++) {
                const size_t k = (size_t)blk * nch + i;
                const gpsbb_chan_t &d = ch[k];
                ChainDesc &cd = c->h_cd[k];
                cd.f_carr = d.f_carr;
                cd.carr_phase = d.carr_phase;
                cd.prn = d.prn;
                cd.start = 0;
                double start0 = 0.0;
                if (d.prn != 0) {
                    /* the part of the descriptor contract the carrier chain depends on */
                    if (d.prn < 0 || d.prn > 32 || !std::isfinite(d.f_carr) || !std::isfinite(d.carr_phase) ||
                        std::signbit(d.carr_phase) || d.carr_phase > 1.0 || !(std::fabs(d.f_carr * delt) <= 0.125)) {
                        bad[i] = 1;
                        cd.prn = 0;
                    } else {
                        if (d.prn != prev_prn)
                            x = d.carr_phase;
                        start0 = x;
                        const volatile double sk = d.f_carr * delt;
                        x = x + (double)nsamp * sk;
                        x -= std::floor(x);
                    }
                }
                c->h_start0[k] = start0;
                prev_prn = cd.prn > 0 ? cd.prn : 0;
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < nch; i++) {
            try {
                th.emplace_back(work, i, i + 1);
            } catch (...) {
                work(i, i + 1);
            }
        }
        for (auto &t : th)
            t.join();
    }
    for (int i = 0; i < nch; i++)
        if (bad[i])
            return GPSBB_E_BADCHAN;

    hipStream_t ss = h->s_seed;
    /* Lap-parallel (gpsbb_laps.hip.h: plan, reference walks, scan, true walks with nothing to emit but the phase after every
     * block, repair) wherever no carrier step is below 2^-50: a fraction of a millisecond per sub-batch whatever its blocks are —
     * the row walks below take as long as their longest chain, twice (a feeder that chains every few slots of a stream, the
     * node driver's incremental run, could not live with 8 ms per call). */
    bool use_laps = h->opt_seed_where != 1 && !GPSBB_KNOB_SET("GPSBB_NO_LAPS");
    for (size_t k = 0; k < nbc_all && use_laps; k++)
        if (c->h_cd[k].prn > 0) {
            const volatile double sk = c->h_cd[k].f_carr * delt;
            use_laps = std::fabs(sk) >= 0x1p-50 || sk == 0.0;
        }
    if (use_laps) {
        if (!c->d_hz_scratch)
            HIPCHK(h, hipMalloc((void **)&c->d_hz_scratch, 64));
        HIPCHK(h, hipMemsetAsync(c->d_hz_scratch, 0, 64, ss));
        if (carr_phase_seed)
            c->h_lap_end.resize(nbc_all);
        int b0 = 0;
        while (b0 < nblocks) {
            /* as many blocks as fit the scratch: by the laps they hold */
            double laps = 0.0;
            int nb = 0;
            while (b0 + nb < nblocks && nb < CHAIN_ONLY_BLOCKS) {
                double l = 0.0;
                /* lanes by the laps per lane a sub-batch of nb + 1 blocks would get (lap_unit grows with the batch: a block counted
                 * with the smaller unit of a smaller batch is over-counted, never under: the scratch stays within CHAIN_ONLY_LAPS) */
                const double unit = (double)lap_unit(NCO_CARR, (size_t)(nb + 1) * nch);
                for (int i = 0; i < nch; i++) {
                    const ChainDesc &d = c->h_cd[(size_t)(b0 + nb) * nch + i];
                    if (d.prn > 0)
                        l += std::floor((std::floor((double)nsamp * std::fabs(d.f_carr * delt)) + 1.0) / unit) + 3.0;
                }
                if (nb > 0 && laps + l > CHAIN_ONLY_LAPS)
                    break;
                laps += l;
                nb++;
            }
            const size_t nbc = (size_t)nb * nch, k0 = (size_t)b0 * nch;
            LapDev L;
            memset(&L, 0, sizeof L);
            lap_bound(ch + k0, nb, nch, delt, nsamp, false, L.chunk0, true);
            const size_t chunks = L.chunk0[1][nch];
            HIPCHK(h, (hipError_t)c->d_cd.reserve(nbc));
            HIPCHK(h, (hipError_t)c->d_lap_end.reserve(nbc));
            HIPCHK(h, (hipError_t)c->d_lap_bc.reserve(2 * nbc));
            HIPCHK(h, (hipError_t)c->d_lap_lane0.reserve(2 * (size_t)nch * ((size_t)nb + 1)));
            HIPCHK(h, (hipError_t)c->d_lap_cnt.reserve(4 * GPSBB_MAX_CHAN));
            HIPCHK(h, (hipError_t)c->d_lap_rec.reserve(chunks * LAP_WG));
            HIPCHK(h, (hipError_t)c->d_lap_agg.reserve(chunks));
            HIPCHK(h, (hipError_t)c->d_lap_chunk_m.reserve(chunks));
            HIPCHK(h, (hipError_t)c->d_lap_chunk_bad.reserve(chunks));
            HIPCHK(h, hipMemcpyAsync(c->d_cd.p, c->h_cd.data() + k0, nbc * sizeof(ChainDesc), hipMemcpyHostToDevice, ss));
            L.bc = c->d_lap_bc.p;
            L.lane0 = c->d_lap_lane0.p;
            L.nlaps = c->d_lap_cnt.p;
            L.nbad = c->d_lap_cnt.p + 2 * GPSBB_MAX_CHAN;
            L.rec = c->d_lap_rec.p;
            L.agg = c->d_lap_agg.p;
            L.chunk_m = c->d_lap_chunk_m.p;
            L.chunk_bad = c->d_lap_chunk_bad.p;
            L.chained = 1;
            L.jitter = (uint32_t)GPSBB_KNOB_LONG("GPSBB_LAP_JITTER", 0);
            L.burst = GPSBB_KNOB_SET("GPSBB_LAP_NO_BURST") ? 0 : (int)GPSBB_KNOB_LONG("GPSBB_LAP_BURST_SHARE", LAP_BURST_SHARE);
            L.unit[NCO_CODE] = lap_unit(NCO_CODE, (size_t)nb * nch);
            L.unit[NCO_CARR] = lap_unit(NCO_CARR, (size_t)nb * nch);
            BatchDev p;
            memset(&p, 0, sizeof p);
            p.nblocks = nb;
            p.nch = nch;
            p.nsamp = nsamp;
            p.ntiles = (nsamp + TILE - 1) / TILE;
            p.delt = delt;
            p.flags = GPSBB_CHAIN_CARRIER;
            p.status = c->d_status;
            p.hazards = c->d_hz_scratch;
            p.chain_dev = 1;
            p.nseg = 1;
            p.nvb = nb;
            p.cd = c->d_cd.p;
            p.carry = c->d_carry;
            p.lap_end = c->d_lap_end.p;
            p.cont0_mask = 0;
            if (b0 > 0)
                for (int i = 0; i < nch; i++) {
                    const int prn = c->h_cd[k0 + i].prn;
                    if (prn > 0 && prn == c->h_cd[k0 - nch + i].prn)
                        p.cont0_mask |= 1u << i;
                }
            const unsigned ck = L.chunk0[NCO_CARR][nch] - L.chunk0[NCO_CARR][0];
            hipLaunchKernelGGL(k_lap_plan<NCO_CARR>, dim3(nch), dim3(64), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_pass1<NCO_CARR>, dim3(ck), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_scan<NCO_CARR>, dim3(nch), dim3(64), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_pass2<NCO_CARR>, dim3(ck), dim3(LAP_WG), 0, ss, p, L);
            hipLaunchKernelGGL(k_lap_repair<NCO_CARR>, dim3(nch), dim3(64), 0, ss, p, L);
            HIPCHK(h, hipGetLastError());
            if (carr_phase_seed)
                HIPCHK(h, hipMemcpyAsync(c->h_lap_end.data() + k0, c->d_lap_end.p, nbc * sizeof(double), hipMemcpyDeviceToHost, ss));
            HIPCHK(h, hipStreamSynchronize(ss));
            b0 += nb;
        }
        if (carr_phase_seed)
            /* a block starts where the one before ended, or — a channel that was idle or had another prn there — from its own phase */
            for (int i = 0; i < nch; i++)
                for (int blk = 0; blk < nblocks; blk++) {
                    const size_t k = (size_t)blk * nch + i;
                    const ChainDesc &d = c->h_cd[k];
                    double v = 0.0;
                    if (d.prn > 0)
                        v = (blk > 0 && c->h_cd[k - nch].prn == d.prn) ? c->h_lap_end[k - nch] : d.carr_phase;
                    carr_phase_seed[k] = v;
                }
        if (carr_phase_end) {
            ChainCarryDev cc;
            HIPCHK(h, hipMemcpy(&cc, c->d_carry, sizeof cc, hipMemcpyDeviceToHost));
            for (int i = 0; i < nch; i++)
                carr_phase_end[i] = c->h_cd[nbc_all - nch + i].prn > 0 ? cc.exact_end[i] : 0.0;
        }
        h->last_chain_dev = 1;
        uint32_t st = 0;
        HIPCHK(h, hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
        return st ? GPSBB_E_INTERNAL : GPSBB_OK;
    }
    for (int b0 = 0; b0 < nblocks; b0 += CHAIN_ONLY_BLOCKS) {
        const int nb = nblocks - b0 < CHAIN_ONLY_BLOCKS ? nblocks - b0 : CHAIN_ONLY_BLOCKS;
        const size_t nbc = (size_t)nb * nch, k0 = (size_t)b0 * nch;
        HIPCHK(h, (hipError_t)c->d_cd.reserve(nbc));
        HIPCHK(h, (hipError_t)c->d_start0.reserve(nbc));
        HIPCHK(h, (hipError_t)c->d_aux.reserve(nbc));
        const int fix_chunks = (nb + FIXP_WG_ALONE - 1) / FIXP_WG_ALONE;
        {
            const size_t nf = (size_t)GPSBB_MAX_CHAN * ((CHAIN_ONLY_BLOCKS + FIXP_WG_ALONE - 1) / FIXP_WG_ALONE);
            if (nf > c->d_fix_flag.cap || c->d_fix_end.cap < c->d_fix_flag.cap || !c->fix_flags_zeroed) {
                c->fix_flags_zeroed = false;
                HIPCHK(h, (hipError_t)c->d_fix_flag.reserve(nf));
                HIPCHK(h, (hipError_t)c->d_fix_end.reserve(c->d_fix_flag.cap));
                HIPCHK(h, hipMemsetAsync(c->d_fix_flag.p, 0, c->d_fix_flag.cap * sizeof(int), ss));
                c->fix_epoch = 0;
                c->fix_flags_zeroed = true;
            }
        }
        HIPCHK(h, hipMemcpyAsync(c->d_cd.p, c->h_cd.data() + k0, nbc * sizeof(ChainDesc), hipMemcpyHostToDevice, ss));
        HIPCHK(h, hipMemcpyAsync(c->d_start0.p, c->h_start0.data() + k0, nbc * sizeof(double), hipMemcpyHostToDevice, ss));
        BatchDev p;
        memset(&p, 0, sizeof p);
        p.nblocks = nb;
        p.nch = nch;
        p.nsamp = nsamp;
        p.ntiles = (nsamp + TILE - 1) / TILE;
        p.delt = delt;
        p.flags = GPSBB_CHAIN_CARRIER;
        p.status = c->d_status;
        p.hazards = h->d_hz;
        p.chain_dev = 1;
        p.chain_starts = 1;
        p.model_start = 0; /* over tens of thousands of blocks a prediction drifts too far: pass A and the prefix stay */
        p.nseg = 1;
        p.seg_tiles = p.ntiles;
        p.nvb = nb;
        p.aux = c->d_aux.p;
        p.cd = c->d_cd.p;
        p.start0 = c->d_start0.p;
        p.carry = c->d_carry;
        p.fix_end = c->d_fix_end.p;
        p.fix_flag = c->d_fix_flag.p;
        p.fix_epoch = ++c->fix_epoch;
        p.fix_chunks = fix_chunks;
        p.cont0_mask = 0;
        if (b0 > 0)
            for (int i = 0; i < nch; i++) {
                const int prn = c->h_cd[k0 + i].prn;
                if (prn > 0 && prn == c->h_cd[k0 - nch + i].prn)
                    p.cont0_mask |= 1u << i;
            }
        p.seed_order = nullptr; /* channel by channel, blocks in order (k_walk) */
        p.seed_lanes = nch * ((nb + 63) & ~63);
        const dim3 wg((p.seed_lanes + GPSBB_WALK_WG - 1) / GPSBB_WALK_WG);
        hipLaunchKernelGGL(k_walk<1>, wg, dim3(GPSBB_WALK_WG), 0, ss, p);
        hipLaunchKernelGGL(k_chain_prefix, dim3(nch), dim3(PREFIX_WG), 0, ss, p);
        hipLaunchKernelGGL(k_walk<3>, wg, dim3(GPSBB_WALK_WG), 0, ss, p);
        hipLaunchKernelGGL(k_chain_fix_par<FIXP_WG_ALONE>, dim3(nch, fix_chunks), dim3(FIXP_WG_ALONE), 0, ss, p);
        HIPCHK(h, hipGetLastError());
        if (carr_phase_seed) {
            /* the exact start phase of every block, as k_chain_fix_par left it in the chain descriptors */
            HIPCHK(h, hipMemcpyAsync(c->h_cd.data() + k0, c->d_cd.p, nbc * sizeof(ChainDesc), hipMemcpyDeviceToHost, ss));
        }
        HIPCHK(h, hipStreamSynchronize(ss)); /* the host image of the next sub-batch's uploads is re-used scratch */
    }
    if (carr_phase_seed)
        for (size_t k = 0; k < nbc_all; k++)
            carr_phase_seed[k] = c->h_cd[k].prn > 0 ? c->h_cd[k].carr_phase : 0.0;
    if (carr_phase_end) {
        ChainCarryDev cc;
        HIPCHK(h, hipMemcpy(&cc, c->d_carry, sizeof cc, hipMemcpyDeviceToHost));
        for (int i = 0; i < nch; i++)
            carr_phase_end[i] = c->h_cd[nbc_all - nch + i].prn > 0 ? cc.exact_end[i] : 0.0;
    }
    h->last_chain_dev = 1;
    uint32_t st = 0;
    HIPCHK(h, hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    return st ? GPSBB_E_INTERNAL : GPSBB_OK;
}

#ifdef GPSBB_EXPERIMENTS
/* ---- test hooks (gpsbb_testhooks.h): the shared NCO code, compiled for the host -------------------- */

extern "C" double gpsbb_test_carr_jump(double x, double s, long long n) { return carr_jump(x, s, n); }

extern "C" double gpsbb_test_code_jump(double x, double s, long long n, long long *wraps)
{
    int64_t w = 0;
    double r = code_jump(x, s, n, &w);
    if (wraps)
        *wraps = w;
    return r;
}

namespace {
struct HostSink {
    gpsbb_test_row_t *rows;
    int cap, cnt;
    unsigned long long fetches;
    void row(int32_t n0, uint32_t nav, uint64_t xb, int64_t inc)
    {
        if (cnt < cap) {
            rows[cnt].n0 = n0;
            rows[cnt].nav = nav;
            rows[cnt].xb = xb;
            rows[cnt].inc = inc;
        }
        cnt++;
    }
    void nav_fetch(uint32_t) { fetches++; }
};
} /* namespace */

extern "C" int gpsbb_test_build_rows(int kind, double x0, double s, unsigned nav0, int nsamp,
                                     gpsbb_test_row_t *rows, int cap, double *x_end, unsigned *nav_end)
{
    HostSink sink{rows, cap, 0, 0};
    uint32_t nav = nav0;
    double x = kind == NCO_CODE ? build_rows<NCO_CODE>(x0, s, nav, nsamp, sink)
                                : build_rows<NCO_CARR>(x0, s, nav, nsamp, sink);
    if (x_end)
        *x_end = x;
    if (nav_end)
        *nav_end = nav;
    return sink.cnt;
}

namespace {
struct HostSinkF64 {
    gpsbb_test_row_t *rows;
    int cap, cnt;
    void row(int32_t n0, uint32_t nav, double x, double S, bool after_wrap)
    {
        if (cnt < cap) {
            rows[cnt].n0 = n0;
            rows[cnt].nav = nav | (after_wrap ? 0x40000000u : 0u); /* bit 30: the row follows a wrap */
            rows[cnt].xb = f64_bits(x);
            rows[cnt].inc = (int64_t)f64_bits(S);
        }
        cnt++;
    }
    void nav_fetch(uint32_t) {}
    void table_index_512() {}
};
} /* namespace */

/* the row builder the device pre-pass runs (build_rows_f64): rows as {n0, nav, bits(x), bits(S)} */
extern "C" int gpsbb_test_build_rows_f64(int kind, double x0, double s, unsigned nav0, int nsamp,
                                         gpsbb_test_row_t *rows, int cap, double *x_end, unsigned *nav_end)
{
    HostSinkF64 sink{rows, cap, 0};
    uint32_t nav = nav0;
    double x = kind == NCO_CODE ? build_rows_f64<NCO_CODE>(x0, s, nav, nsamp, sink)
                                : build_rows_f64<NCO_CARR>(x0, s, nav, nsamp, sink);
    if (x_end)
        *x_end = x;
    if (nav_end)
        *nav_end = nav;
    return sink.cnt;
}

/* the host's prediction of a carrier n steps on (CarrDrift: where pass B of the device-side chain starts a segment) */
extern "C" double gpsbb_test_carr_predict(double x0, double s, int n)
{
    const CarrDrift d(s);
    return d.advance(x0, n, s);
}

/* the fixed-point carrier's table index at the first sample of tile t, as the model kernels' pre-pass writes it */
extern "C" double gpsbb_test_fixed_tile_index(uint32_t ph0, int32_t step, int t)
{
    return fixed_tile_index(ph0, step, t);
}

/* The realised error of the model kernels' in-tile models against the reference's own recurrence, over every tile of the
 * batch's LAST run (gpsbb_modelerr.hip.h).  maxima: [GPSBB_MAX_CHAN][GPSBB_TEST_ME_NQ] doubles, counts:
 * [GPSBB_MAX_CHAN][GPSBB_TEST_MEC_NQ]; *which = 1 for k_synth_ev / k_synth_ev_dense / k_synth_ev_fixed, 2 for k_synth_pd. */
extern "C" int gpsbb_test_model_err(gpsbb_batch_t *b, double *maxima, unsigned long long *counts, int *which)
{
    if (!b || !maxima || !counts)
        return GPSBB_E_BADARG;
    gpsbb *h = b->h;
    if (!b->ran || !b->ev)
        return GPSBB_E_STATE; /* only the model kernels have a model */
    const int rc = gpsbb_sync(h);
    if (rc != GPSBB_OK)
        return rc;
    const BatchDev p = batch_dev(b, b->last_set);
    double *d_mx = nullptr;
    unsigned long long *d_cnt = nullptr;
    const size_t mx_bytes = sizeof(double) * GPSBB_MAX_CHAN * ME_NQ, cnt_bytes = sizeof(unsigned long long) * GPSBB_MAX_CHAN * MEC_NQ;
    HIPCHK(h, hipMalloc((void **)&d_mx, mx_bytes));
    HIPCHK(h, hipMalloc((void **)&d_cnt, cnt_bytes));
    HIPCHK(h, hipMemsetAsync(d_mx, 0, mx_bytes, h->s_compute));
    HIPCHK(h, hipMemsetAsync(d_cnt, 0, cnt_bytes, h->s_compute));
    const size_t threads = (size_t)b->nblocks * b->nch * b->ntiles;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (b->ev_all_dense)
        hipLaunchKernelGGL(k_model_err_pd, grid, dim3(256), 0, h->s_compute, p, d_mx, d_cnt);
    else
        hipLaunchKernelGGL(k_model_err_ev, grid, dim3(256), 0, h->s_compute, p, d_mx, d_cnt);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->s_compute));
    HIPCHK(h, hipMemcpy(maxima, d_mx, mx_bytes, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(counts, d_cnt, cnt_bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_mx);
    (void)hipFree(d_cnt);
    if (which)
        *which = b->ev_all_dense ? 2 : 1;
    return GPSBB_OK;
}

/* What the pre-pass of the batch's LAST run left for the model kernels — every tile state (bit patterns), every tile's data bits,
 * every end-of-block state — as three 64-bit sums of mixed words: the lap-parallel pre-pass (its reference
 * states on the model or pushed far off it: GPSBB_LAP_JITTER) and the row walks must leave the same bits, whatever the IQ makes of
 * them (tools/table_check.py). */
extern "C" int gpsbb_test_table_digest(gpsbb_batch_t *b, unsigned long long out[3])
{
    if (!b || !out || !b->ran || !b->ev)
        return GPSBB_E_STATE;
    gpsbb *h = b->h;
    const int rc = gpsbb_sync(h);
    if (rc != GPSBB_OK)
        return rc;
    const BatchDev p = batch_dev(b, b->last_set);
    auto mix = [](unsigned long long z) { z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29; return z; };
    const size_t nx = (size_t)b->nblocks * 2 * b->nch * b->ntiles, nn = (size_t)b->nblocks * b->nch * b->ntiles, ne = (size_t)b->nblocks * b->nch;
    std::vector<unsigned long long> hx(nx);
    std::vector<uint32_t> hn(nn);
    std::vector<gpsbb_chan_state_t> he(ne);
    HIPCHK(h, hipMemcpy(hx.data(), p.tile_x, nx * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(hn.data(), p.tile_nav, nn * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(he.data(), p.end, ne * sizeof(gpsbb_chan_state_t), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = 0;
    /* (idle channels' entries are whatever the allocation held: only active block-channels count) */
    std::vector<gpsbb_chan_t> hc(ne);
    HIPCHK(h, hipMemcpy(hc.data(), p.ch, ne * sizeof(gpsbb_chan_t), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < ne; k++) {
        if (hc[k].prn <= 0)
            continue;
        const size_t blk = k / (size_t)b->nch, i = k % (size_t)b->nch;
        for (int kind = 0; kind < 2; kind++)
            for (int t = 0; t < b->ntiles; t++) {
                const size_t at = (blk * 2 * b->nch + 2 * i + kind) * (size_t)b->ntiles + t;
                out[0] += mix(hx[at] + 0x9E3779B97F4A7C15ull * (at + 1));
            }
        for (int t = 0; t < b->ntiles; t++) {
            const size_t at = k * (size_t)b->ntiles + t;
            out[1] += mix((unsigned long long)(hn[at] & 3u) + 0x9E3779B97F4A7C15ull * (at + 1));
        }
        unsigned long long w[5];
        memcpy(w, &he[k], sizeof w);
        for (int q = 0; q < 4; q++) /* carr_phase, code_phase, (iword, ibit), (icode, dataBit); codeCA with the padding */
            out[2] += mix(w[q] + 0x9E3779B97F4A7C15ull * (k * 5 + q + 1));
        out[2] += mix((unsigned long long)(uint32_t)he[k].codeCA + 0x9E3779B97F4A7C15ull * (k * 5 + 5));
    }
    return GPSBB_OK;
}

/* the derived budgets this build was compiled with, in units of 2^-32: EV_MODEL_ERR, EV_T_EPS, PD_BAND */
extern "C" void gpsbb_test_budgets(double out[3])
{
    out[0] = EV_MODEL_ERR * 0x1p+32;
    out[1] = EV_T_EPS * 0x1p+32;
    out[2] = (double)PD_BAND;
}

extern "C" unsigned long long gpsbb_test_row_bound(int kind, double s_abs, int nsamp)
{
    return kind == NCO_CODE ? row_bound(s_abs, 1023.0, 9, nsamp) : row_bound(s_abs, 1.0, -1, nsamp);
}

/* FETCH_SIZE calibration: `bytes` bytes of d_src (device memory, at least that big) read once with the tile states' pattern
 * (k_read_pattern: 32 rows of 2442 doubles per wavefront, what a wavefront of k_synth_ev reads of one block); returns the bytes
 * actually read */
extern "C" long long gpsbb_test_read_pattern(gpsbb_t *h, const void *d_src, size_t bytes)
{
    if (!h || !d_src)
        return GPSBB_E_BADARG;
    const int rows = 32, cols = 2442;
    const size_t per_wave = (size_t)rows * cols * 8;
    size_t waves = bytes / per_wave;
    waves -= waves % 4;
    if (waves < 4)
        return GPSBB_E_BADARG;
    double *d_sink = nullptr;
    if (hipMalloc((void **)&d_sink, 8) != hipSuccess)
        return GPSBB_E_NOMEM;
    hipLaunchKernelGGL(k_read_pattern, dim3((unsigned)(waves / 4)), dim3(256), 0, h->s_compute, (const double *)d_src, rows, cols, d_sink);
    const hipError_t e = hipStreamSynchronize(h->s_compute);
    (void)hipFree(d_sink);
    return e == hipSuccess ? (long long)(waves * per_wave) : (long long)GPSBB_E_HIP;
}

#ifdef GPSBB_WG_TRACE
/* The workgroup trace of the measurement build (gpsbb_kernels.hip.h: wg_trace_leave; make trace; tools/corun_diag.py).
 * _begin: (re)arm the trace with room for `cap` records; _read: wait for the device, copy out up to `cap` records of
 * WG_TRACE_WORDS 64-bit words, return how many were written (the device counts on past the capacity). */
namespace {
unsigned long long *g_trace_buf = nullptr;
unsigned g_trace_cap = 0;
}
extern "C" int gpsbb_test_wg_trace_begin(unsigned cap)
{
    if (hipDeviceSynchronize() != hipSuccess)
        return GPSBB_E_HIP;
    if (cap > g_trace_cap) {
        if (g_trace_buf)
            (void)hipFree(g_trace_buf);
        g_trace_buf = nullptr;
        g_trace_cap = 0;
        if (hipMalloc((void **)&g_trace_buf, (size_t)cap * WG_TRACE_WORDS * 8) != hipSuccess)
            return GPSBB_E_NOMEM;
        g_trace_cap = cap;
    }
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &g_trace_buf, sizeof g_trace_buf) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_cap), &cap, sizeof cap) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_n), &zero, sizeof zero) != hipSuccess)
        return GPSBB_E_HIP;
    return hipDeviceSynchronize() == hipSuccess ? GPSBB_OK : GPSBB_E_HIP;
}
extern "C" long gpsbb_test_wg_trace_read(unsigned long long *out, unsigned cap)
{
    if (hipDeviceSynchronize() != hipSuccess)
        return GPSBB_E_HIP;
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wg_trace_n), sizeof n) != hipSuccess)
        return GPSBB_E_HIP;
    const unsigned have = n < g_trace_cap ? n : g_trace_cap, take = have < cap ? have : cap;
    if (take && hipMemcpy(out, g_trace_buf, (size_t)take * WG_TRACE_WORDS * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return GPSBB_E_HIP;
    return (long)n;
}
#endif
#endif /* GPSBB_EXPERIMENTS */
