/*
 * gpsbb_kernels.hip.h — the device side of libgpsbb: hand-written HIP for gfx950 (CDNA4).
 *
 * Two kernels per batch of blocks:
 *
 *   k_seed   NCO seeding pre-pass.  One lane per NCO chain (block x channel x {code, carrier}).  Walks
 *            the chain with the exact jump-ahead of gpsbb_nco.h — O(#binade crossings + #wraps), not
 *            O(#samples) — and writes the chain's row table {n0, bits(x), inc}, the row index of every
 *            tile start, and the end-of-block state (the reference's live-out, plutogpssim.c:2741-2746).
 *            This replaces the sample-to-sample dependency of plutogpssim.c:2709/2741 with a table any
 *            lane can index.
 *
 *   k_synth  The sample loop itself (plutogpssim.c:2690-2756), one lane per run of SPT consecutive
 *            output samples.  Per workgroup the per-channel tables are staged in LDS: the amplitude LUT
 *            (int)(cosTable512[k]*gain), (int)(sinTable512[k]*gain) packed as int16x2 — the product
 *            dataBit*codeCA*table*gain of c:2701-2702 factorises into sign * that LUT because IEEE
 *            multiply and truncation are odd-symmetric — the 1023 C/A chips bit-packed (32 dwords per
 *            PRN) and the 60 nav words.  A lane looks up its start state in the row tables, then steps
 *            both NCOs with genuine IEEE double adds (__dadd_rn, never an FMA), accumulates all
 *            channels in packed int16x2 (wrap-around == the reference's (short) cast, c:2754-2755) and
 *            stores 16-byte vectors.
 *
 * No MFMA anywhere: this is table-driven fixed-point work.  The roofline that bounds the output is the
 * HBM write stream (4 bytes per IQ sample); the unit that actually saturates is the VALU (two FP64 adds
 * and ~25 integer ops per channel-sample).
 */
#ifndef GPSBB_KERNELS_HIP_H
#define GPSBB_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpsbb.h"
#include "gpsbb_nco.h"

namespace gpsbb_impl {

constexpr int TILE_THREADS = 256;           /* 4 wave64 per workgroup */
constexpr int SPT = 16;                     /* consecutive samples per lane: 64 bytes of output */
constexpr int TILE = TILE_THREADS * SPT;    /* samples per workgroup */

constexpr uint32_t ST_ROW_OVERFLOW = 1u;

/* Everything the kernels need about one batch; passed by value as the kernel argument. */
struct BatchDev {
    const gpsbb_chan_t *ch;         /* [nblocks*nch] descriptors, block-major                        */
    const gpsbb_chan_t *prev_ch;    /* [nch] last block of the previous push of a stream, or NULL   */
    const gpsbb_chan_state_t *prev_end; /* [nch] its end state, or NULL                              */
    int nblocks, nch, nsamp, ntiles;
    double delt;
    unsigned flags;
    const int32_t *tabs;            /* cos512[512] then sin512[512] (plutogpssim.c:93-161)           */
    const uint32_t *ca_bits;        /* [33][32] C/A chips per PRN, bit i of dword i>>5 = chip i      */
    NcoRow *rows;                   /* row pool                                                      */
    const uint64_t *row_off;        /* [2*nblocks*nch + 1] first row of each chain in the pool       */
    int32_t *tile_row;              /* [2*nblocks*nch][ntiles] row holding each tile's first sample  */
    gpsbb_chan_state_t *end;        /* [nblocks*nch] end-of-block state                              */
    uint32_t *status;               /* self-check word                                               */
    unsigned long long *hazards;    /* [0] itable_512, [1] dwrd_oob                                  */
};

__device__ __forceinline__ int chain_code(const BatchDev &p, int b, int i) { return b * p.nch + i; }
__device__ __forceinline__ int chain_carr(const BatchDev &p, int b, int i) { return p.nblocks * p.nch + b * p.nch + i; }

/* nav data bit (+1/-1) for packed counters, from 60 words at `dwrd` (plutogpssim.c:1781, 2732) */
template <class P>
__device__ __forceinline__ int nav_bit(const P dwrd, uint32_t nav)
{
    int w = nav_iword(nav);
    w = w < GPSBB_N_DWRD ? w : GPSBB_N_DWRD - 1; /* latent OOB of the reference: defined as dwrd[59] */
    return (int)((dwrd[w] >> (29 - nav_ibit(nav))) & 1u) * 2 - 1;
}

/* ---- k_seed -------------------------------------------------------------------------------------- */

struct RowSink {
    NcoRow *rows;
    uint32_t cap;  /* rows available, not counting the sentinel slot */
    uint32_t cnt;
    int32_t *tile_row;
    int ntiles, next_tile;
    bool overflow;
    unsigned long long *hz;

    __device__ __forceinline__ void row(int32_t n0, uint32_t nav, uint64_t xb, int64_t inc)
    {
        /* tiles that start before this row belong to the previous one */
        while (next_tile < ntiles && (int64_t)next_tile * TILE < (int64_t)n0)
            tile_row[next_tile++] = (int32_t)cnt - 1;
        if (cnt < cap) {
            NcoRow r;
            r.n0 = n0;
            r.nav = nav;
            r.xb = xb;
            r.inc = inc;
            rows[cnt] = r;
        } else {
            overflow = true;
        }
        cnt++;
    }
    __device__ __forceinline__ void nav_fetch(uint32_t nav)
    {
        if (nav_iword(nav) >= GPSBB_N_DWRD)
            atomicAdd(hz + 1, 1ull);
    }
    __device__ __forceinline__ void finish()
    {
        while (next_tile < ntiles)
            tile_row[next_tile++] = (int32_t)cnt - 1;
        const uint32_t at = cnt < cap ? cnt : cap;
        NcoRow r;
        r.n0 = INT32_MAX; /* sentinel: terminates every forward scan */
        r.nav = 0;
        r.xb = 0;
        r.inc = 0;
        rows[at] = r;
    }
};

__device__ __forceinline__ RowSink make_sink(const BatchDev &p, int chain)
{
    RowSink s;
    const uint64_t o0 = p.row_off[chain], o1 = p.row_off[chain + 1];
    s.rows = p.rows + o0;
    s.cap = (uint32_t)(o1 - o0 - 1);
    s.cnt = 0;
    s.tile_row = p.tile_row + (size_t)chain * p.ntiles;
    s.ntiles = p.ntiles;
    s.next_tile = 0;
    s.overflow = false;
    s.hz = p.hazards;
    return s;
}

__device__ inline void seed_code_chain(const BatchDev &p, int b, int i)
{
    const gpsbb_chan_t &c = p.ch[(size_t)b * p.nch + i];
    gpsbb_chan_state_t &e = p.end[(size_t)b * p.nch + i];
    if (c.prn <= 0) {
        e.code_phase = 0.0;
        e.iword = e.ibit = e.icode = e.dataBit = e.codeCA = 0;
        e._pad = 0;
        return;
    }
    RowSink sink = make_sink(p, chain_code(p, b, i));
    uint32_t nav = nav_pack(c.icode, c.ibit, c.iword);
    const double s = mul_rn(c.f_code, p.delt); /* plutogpssim.c:2709: f_code * delt, rounded on its own */
    const double x = build_rows<NCO_CODE>(c.code_phase, s, nav, p.nsamp, sink);
    sink.finish();
    if (sink.overflow)
        atomicOr(p.status, ST_ROW_OVERFLOW);
    e.code_phase = x;
    e.iword = nav_iword(nav);
    e.ibit = nav_ibit(nav);
    e.icode = nav_icode(nav);
    e.dataBit = nav_bit(c.dwrd, nav);
    const int ci = (int)x;
    e.codeCA = (int)((p.ca_bits[c.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1; /* c:2737 */
    e._pad = 0;
}

__device__ inline double seed_carr_chain(const BatchDev &p, int b, int i, double x0)
{
    const gpsbb_chan_t &c = p.ch[(size_t)b * p.nch + i];
    gpsbb_chan_state_t &e = p.end[(size_t)b * p.nch + i];
    if (c.prn <= 0) {
        e.carr_phase = 0.0;
        return 0.0;
    }
    RowSink sink = make_sink(p, chain_carr(p, b, i));
    uint32_t nav = 0;
    const double s = mul_rn(c.f_carr, p.delt); /* plutogpssim.c:2741 */
    const double x = build_rows<NCO_CARR>(x0, s, nav, p.nsamp, sink);
    sink.finish();
    if (sink.overflow)
        atomicOr(p.status, ST_ROW_OVERFLOW);
    e.carr_phase = x;
    return x;
}

/* grid: lanes [0, nbc) = code chains; lanes [cbase, ...) = carrier chains (cbase = nbc rounded up to a
 * wave so that the two kinds of chain never share a wavefront). */
__global__ __launch_bounds__(64) void k_seed(BatchDev p, int cbase)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nbc = p.nblocks * p.nch;
    if (gid < nbc) {
        seed_code_chain(p, gid / p.nch, gid % p.nch);
        return;
    }
    const int g = gid - cbase;
    if (g < 0)
        return;
    if (p.flags & GPSBB_CHAIN_CARRIER) {
        /* one lane per channel walks the blocks in time order: block b starts where b-1 ended, unless
         * the channel was (re)allocated, in which case the descriptor's own carr_phase applies
         * (allocateChannel, plutogpssim.c:1956-1964) */
        if (g >= p.nch)
            return;
        int prev_prn = 0;
        double prev_x = 0.0;
        if (p.prev_ch && p.prev_end) {
            prev_prn = p.prev_ch[g].prn;
            prev_x = p.prev_end[g].carr_phase;
        }
        for (int b = 0; b < p.nblocks; b++) {
            const gpsbb_chan_t &c = p.ch[(size_t)b * p.nch + g];
            const double x0 = (c.prn > 0 && c.prn == prev_prn) ? prev_x : c.carr_phase;
            prev_x = seed_carr_chain(p, b, g, x0);
            prev_prn = c.prn > 0 ? c.prn : 0;
        }
    } else {
        if (g >= nbc)
            return;
        const int b = g / p.nch, i = g % p.nch;
        seed_carr_chain(p, b, i, p.ch[(size_t)b * p.nch + i].carr_phase);
    }
}

/* ---- k_synth ------------------------------------------------------------------------------------- */

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s u32_v2s(uint32_t u)
{
    v2s v;
    __builtin_memcpy(&v, &u, 4);
    return v;
}
__device__ __forceinline__ uint32_t v2s_u32(v2s v)
{
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    return u;
}

/* state of one NCO at sample n, from the chain's row table */
__device__ __forceinline__ uint64_t row_state(const NcoRow *__restrict__ rows, int r, int n, uint32_t *nav)
{
    while (rows[r + 1].n0 <= n)
        r++;
    const NcoRow row = rows[r];
    if (nav)
        *nav = row.nav;
    return row.xb + (uint64_t)((int64_t)(n - row.n0) * row.inc);
}

__global__ __launch_bounds__(TILE_THREADS) void k_synth(BatchDev p, int16_t *__restrict__ iq)
{
    __shared__ uint32_t s_amp[GPSBB_MAX_CHAN][512]; /* int16x2: lo = I (cos), hi = Q (sin) */
    __shared__ uint32_t s_ca[GPSBB_MAX_CHAN][32];
    __shared__ uint32_t s_dwrd[GPSBB_MAX_CHAN][GPSBB_N_DWRD];
    __shared__ double s_sc[GPSBB_MAX_CHAN], s_sk[GPSBB_MAX_CHAN];
    __shared__ int s_act[GPSBB_MAX_CHAN];
    __shared__ int s_nact;

    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int b = blockIdx.y;
    const gpsbb_chan_t *__restrict__ cb = p.ch + (size_t)b * p.nch;

    /* ---- stage the block's per-channel tables in LDS ---- */
    if (tid == 0) {
        int na = 0;
        for (int i = 0; i < p.nch; i++)
            if (cb[i].prn > 0)
                s_act[na++] = i;
        s_nact = na;
    }
    if (tid < p.nch) {
        s_sc[tid] = mul_rn(cb[tid].f_code, p.delt);
        s_sk[tid] = mul_rn(cb[tid].f_carr, p.delt);
    }
    for (int e = tid; e < p.nch * 512; e += TILE_THREADS) {
        const int i = e >> 9, k = e & 511;
        uint32_t v = 0;
        if (cb[i].prn > 0) {
            const double g = cb[i].gain;
            /* (int)(table * gain): one IEEE multiply, truncation toward zero (plutogpssim.c:2701-2702) */
            const int ip = (int)mul_rn((double)p.tabs[k], g);
            const int qp = (int)mul_rn((double)p.tabs[512 + k], g);
            v = ((uint32_t)ip & 0xffffu) | ((uint32_t)qp << 16);
        }
        s_amp[i][k] = v;
    }
    for (int e = tid; e < p.nch * 32; e += TILE_THREADS) {
        const int i = e >> 5, w = e & 31;
        const int prn = cb[i].prn;
        s_ca[i][w] = prn > 0 ? p.ca_bits[prn * 32 + w] : 0u;
    }
    for (int e = tid; e < p.nch * GPSBB_N_DWRD; e += TILE_THREADS) {
        const int i = e / GPSBB_N_DWRD, w = e % GPSBB_N_DWRD;
        s_dwrd[i][w] = cb[i].dwrd[w];
    }
    __syncthreads();

    const int n0 = tile * TILE + tid * SPT;
    if (n0 >= p.nsamp)
        return;

    v2s acc[SPT];
#pragma unroll
    for (int j = 0; j < SPT; j++)
        acc[j] = v2s{0, 0};

    const int nact = s_nact;
    unsigned long long hz_itable = 0;
    for (int a = 0; a < nact; a++) {
        const int i = s_act[a];
        const int cc = chain_code(p, b, i), ck = chain_carr(p, b, i);
        uint32_t nav;
        double xc = bits_f64(row_state(p.rows + p.row_off[cc], p.tile_row[(size_t)cc * p.ntiles + tile], n0, &nav));
        double xk = bits_f64(row_state(p.rows + p.row_off[ck], p.tile_row[(size_t)ck * p.ntiles + tile], n0, nullptr));
        const double sc = s_sc[i], sk = s_sk[i];
        int db = nav_bit(s_dwrd[i], nav);

#pragma unroll
        for (int j = 0; j < SPT; j++) {
            /* carrier table index: floor(carr_phase*512) (c:2697); x*512 is exact, x >= 0 */
            int it = (int)(xk * 512.0);
            if (it > 511) { /* carr_phase == 1.0 exactly: latent OOB of the reference, defined as &511 */
                it &= 511;
                if (n0 + j < p.nsamp)
                    hz_itable++;
            }
            const int ci = (int)xc; /* chip index (c:2737) */
            const int chip = (int)((s_ca[i][ci >> 5] >> (ci & 31)) & 1u);
            const short sg = (short)((2 * chip - 1) * db); /* codeCA * dataBit */
            acc[j] += u32_v2s(s_amp[i][it]) * v2s{sg, sg};

            /* code NCO (c:2709-2734) */
            xc = add_rn(xc, sc);
            if (xc >= 1023.0) {
                xc = add_rn(xc, -1023.0);
                nav = nav_advance(nav);
                if (nav_icode(nav) == 0)
                    db = nav_bit(s_dwrd[i], nav);
            }
            /* carrier NCO (c:2741-2746) */
            xk = add_rn(xk, sk);
            if (xk >= 1.0)
                xk = add_rn(xk, -1.0);
            else if (xk < 0.0)
                xk = add_rn(xk, 1.0);
        }
    }
    if (hz_itable)
        atomicAdd(p.hazards, hz_itable);

    /* ---- store: int16 I,Q interleaved (c:2754-2755) ---- */
    uint32_t *out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + n0;
    const bool full = n0 + SPT <= p.nsamp;
    if (full && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
        for (int j = 0; j < SPT; j += 4)
            o4[j >> 2] = make_uint4(v2s_u32(acc[j]), v2s_u32(acc[j + 1]), v2s_u32(acc[j + 2]), v2s_u32(acc[j + 3]));
    } else {
#pragma unroll
        for (int j = 0; j < SPT; j++)
            if (n0 + j < p.nsamp)
                out[j] = v2s_u32(acc[j]);
    }
}

/* pure write stream of the same shape as k_synth's output: the empirical int16x2 write ceiling */
__global__ __launch_bounds__(256) void k_fill_ceiling(uint4 *__restrict__ dst, size_t n16, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint32_t v = seed + (uint32_t)i;
        dst[i] = make_uint4(v, v ^ 0x00010001u, v + 0x00020002u, v ^ 0x7fff7fffu);
    }
}

} /* namespace gpsbb_impl */
#endif
