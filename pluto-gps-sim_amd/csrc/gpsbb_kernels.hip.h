/*
 * gpsbb_kernels.hip.h — the device side of libgpsbb: hand-written HIP for gfx950 (CDNA4).
 *
 * Two kernels per batch of blocks:
 *
 *   k_seed        NCO seeding pre-pass.  One lane per NCO chain (block x channel x {code, carrier}).
 *                 Walks the chain with the exact jump-ahead of gpsbb_nco.h — O(#binade crossings +
 *                 #wraps), not O(#samples) — and writes the chain's row table and the end-of-block state
 *                 (the reference's live-out, plutogpssim.c:2741-2746).  A row is {n0, nav, x, S}: inside it
 *                 the state at sample n is exactly fma(n - n0, S, x).  This replaces the sample-to-sample
 *                 dependency of plutogpssim.c:2709/2741 with a table any lane can index.  While emitting
 *                 rows it also fills the tile index: which row holds the first sample of every 1024-sample
 *                 tile, laid out [block][tile][chain] so one tile's 32 entries share a line.  Sequential per
 *                 chain, so it runs on its own stream into double-buffered tables and overlaps the previous
 *                 run's k_synth.
 *
 *   k_synth       The sample loop itself (plutogpssim.c:2690-2756), one lane per run of SPT consecutive
 *                 output samples, one wavefront per tile.  Per workgroup the per-channel tables are staged
 *                 in LDS once: the amplitude LUT (int)(cosTable512[k]*gain), (int)(sinTable512[k]*gain)
 *                 packed as int16x2 — the product dataBit*codeCA*table*gain of c:2701-2702 factorises into
 *                 sign * that LUT because IEEE multiply and truncation are odd-symmetric — the 1023 C/A
 *                 chips as +-1 bytes and the 60 nav words.  After that every wavefront works alone: it
 *                 takes chunks of tiles from a per-block counter; the rows of all 2*nch chains that overlap
 *                 a tile (about 70 at 25 MS/s: binade crossings cluster after every wrap) are copied into
 *                 the wavefront's LDS slice by LDS-DMA, one row per lane, while the previous tile's last
 *                 channel is still being walked; each lane finds its row per chain by counting row starts
 *                 through scalar registers and gets its start state with one FMA; then it steps both NCOs
 *                 with genuine IEEE double adds (__dadd_rn, never an FMA), accumulates all channels in
 *                 packed int16x2 (wrap-around == the reference's (short) cast, c:2754-2755) and stores
 *                 16-byte vectors.
 *
 * No MFMA anywhere: this is table-driven fixed-point work.  The roofline that bounds the output is the
 * HBM write stream (4 bytes per IQ sample); the unit that actually saturates is the VALU (per
 * channel-sample: 2 FP64 adds, 2 FP64->int conversions, 4 integer/packed ops, 2 LDS reads).
 */
#ifndef GPSBB_KERNELS_HIP_H
#define GPSBB_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpsbb.h"
#include "gpsbb_nco.h"

namespace gpsbb_impl {

#ifndef GPSBB_WG
#define GPSBB_WG 512
#endif
#ifndef GPSBB_WAVES_PER_SIMD
#define GPSBB_WAVES_PER_SIMD 4
#endif
constexpr int TILE_THREADS = GPSBB_WG;      /* wave64 x (GPSBB_WG/64) per workgroup */
#ifndef GPSBB_SPT
#define GPSBB_SPT 16
#endif
constexpr int SPT = GPSBB_SPT;              /* consecutive samples per lane (16 -> 64 bytes of output) */
constexpr int TILE = 64 * SPT;              /* samples per tile = one pass of one wavefront (the row-index granule) */
constexpr int WAVES_PER_WG = TILE_THREADS / 64;
#ifndef GPSBB_TILE_CHUNK
#define GPSBB_TILE_CHUNK 2
#endif
constexpr int TILE_CHUNK = GPSBB_TILE_CHUNK;               /* consecutive tiles a wavefront takes at a time */
#ifndef GPSBB_ROW_CAP
#define GPSBB_ROW_CAP 128
#endif
constexpr int WAVE_ROW_CAP = GPSBB_ROW_CAP;           /* rows of all chains of one tile staged in a wavefront's LDS slice */

constexpr uint32_t ST_ROW_OVERFLOW = 1u;
constexpr uint32_t ST_LDS_LAYOUT = 4u;  /* k_synth_ev's LDS image does not start at LDS address 0 (see ev_d_add) */
constexpr uint32_t ST_CHAIN_STALL = 2u; /* k_chain_fix_par gave up waiting for the chunk before it (see there) */

/*
 * One row of a chain's table as the device pool holds it (24 bytes, the same slots as gpsbb_nco.h's NcoRow
 * {n0, nav, bits, inc}).  Inside a row every step adds the same whole number of units in the last place
 * and never leaves the binade, so the state at sample n is x + (n - n0)*S with S = inc * ulp: one FMA
 * gives it exactly (the product is exact, the sum is a representable number, so the single rounding does
 * nothing).  Code rows carry the data bit that is valid for the whole row in nav bit 31 (1 = dataBit -1);
 * carrier rows are stored scaled by 512 (exact), which is how the walk indexes the 512-entry tables.
 */
struct SynRow {
    int32_t n0;
    uint32_t nav;
    double x;
    double S;
};
static_assert(sizeof(SynRow) == sizeof(NcoRow), "row pool is sized for NcoRow");

/* Per (block, channel) constants of the breakpoint kernel (gpsbb_events.hip.h), built on the host at batch
 * set-up from the descriptors with the same individually rounded products the kernels use. */
struct EvConst {
    /* what k_synth_ev's channel loop takes into scalar registers comes first, in the order it is loaded: 32 + 16 bytes,
     * two scalar loads per channel, and records of 128 bytes so that a channel's address is a shift (gpsbb_events.hip.h:
     * the loop's time follows its instruction count, scalar ones included) */
    double S;     /* |carrier step| per sample in table-index units: 512 * |fl(f_carr*delt)|, exact        */
    double rS;    /* 1/|S| (2^1000 where S == 0: no index change is ever in reach)                        */
    double sc;    /* code step per sample in chips: fl(f_code*delt)                                       */
    double rsc;   /* 1/sc                                                                                  */
    uint32_t danger_le; /* danger - 1, and 0xffffffff where every run is recomputed exactly (kc < 0): "low word <= this"    */
    uint32_t chip_at;   /* k_synth_ev / k_synth_ev_fixed (EvLdsLean): LDS address of the channel's chip table, minus what the
                           exponent bits of a guard-format high word contribute when it is shifted into a byte offset    */
    uint32_t amp_at;    /* ... of its amplitude table, likewise                                                          */
    uint32_t danger; /* a low word (fraction in units of 2^-32) below this: the model cannot be trusted (= 2W)  */
    double tK0;   /* rS*(1 + W) + 2^20 + W: what turns the fraction of the (biased) first-sample model into the position
                     of the next index change, in guard format (gpsbb_events.hip.h)                         */
    double tC0;   /* as tK0, for the chip change                                                           */
    double W;     /* the channel's bias: every tested quantity carries +W, W >= its model error             */
    int32_t kc;   /* carrier breakpoints a run of SPT samples can hold (1..4); -1: always recompute exactly */
    int32_t down; /* the carrier step is negative                                                         */
    /* k_synth_pd's steps, scaled to LDS byte addresses (exact: powers of two): per sample and per 64 samples */
    double pd_S8;   /* 8 * S: the amplitude table has 8 bytes per entry   */
    double pd_dy;   /* 64 * pd_S8                                          */
    double pd_sc2;  /* 2 * sc: the chip tables have 2 bytes per chip      */
    double pd_dx;   /* 64 * pd_sc2                                         */
    uint32_t _pad[4];
};
static_assert(sizeof(EvConst) == 128 && offsetof(EvConst, danger_le) == 32, "EvConst layout");

/* Per (block, channel) scratch of the device-side carrier chain (gpsbb_walk.hip.h, k_chain_fix). */
constexpr int CHAIN_MAX_CROSS = 20;
constexpr int CHAIN_PREFIX_CAP = 64; /* rows of one lap walked by k_chain_fix on its own */
struct ChainAux {
    double start1; /* start phase pass B walks from: good to a few units in the last place                  */
    double endA;   /* end phase of pass A's walk from the rough start phase (BatchDev::start0)             */
    double margin; /* pass B: smallest distance of a row's first or last state to an edge of its binade    */
    /* pass B: the rows, up to and including the one after the first wrap, whose first state came out of a sum
     * rounded on a coarser grid than the states of the row before (a binade crossed upwards, or a wrap): only
     * there can the offset between pass B's trajectory and the true one change */
    int32_t ncross;                 /* -1: more than CHAIN_MAX_CROSS of them                              */
    int32_t cross[CHAIN_MAX_CROSS]; /* the first sample of each such row, ascending (nsamp: the block's last step) */
    uint32_t hz512;                 /* pass B: samples whose phase was exactly 1.0                         */
    int32_t wrap_row;               /* pass B: the first sample after the first wrap (-1: no wrap in the block) */
    /* k_chain_fix, when it walked the block's first lap on its own: rows 0 .. prefix_cnt-1 of the chain's
     * prefix region hold the samples before prefix_end (= wrap_row) and pass B's rows before it are void */
    int32_t prefix_cnt, prefix_end;
    int32_t _pad;
    /* pass B, for crossing j: its state at sample cross[j]-1 and at sample cross[j] */
    double pre[CHAIN_MAX_CROSS], post[CHAIN_MAX_CROSS];
    double endB;   /* pass B's end state */
    /* k_chain_fix: true state minus pass B's state, for samples cross[j-1] <= n < cross[j] (seg[0]: from sample 0,
     * seg[ncross]: to the end of the block) */
    double seg[CHAIN_MAX_CROSS + 1];
    double wrap_x; /* pass B's state at sample wrap_row */
};

/* What the chain kernels (k_walk passes A / B, k_chain_prefix, k_chain_fix*) read of a (block, channel): 24 bytes
 * instead of the 296 of a descriptor, so that a chain over tens of thousands of blocks (the seed of a time shard,
 * gpsbb_chain_carrier) uploads and reads little.  carr_phase: in: the descriptor's; out (chain_starts): the exact
 * phase at the block's first sample. */
struct ChainDesc {
    double f_carr;
    double carr_phase;
    int32_t prn;
    int32_t start; /* 1: this segment starts a chain from its own carr_phase whatever came before (the first segment of a
                      block of a batch whose blocks are independent); 0: it continues the segment before it if the prn is
                      the same */
};
static_assert(sizeof(ChainDesc) == 24, "ChainDesc layout");

/* The carrier of a stream (gpsbb_stream_*) from one push to the next, on the device. */
struct ChainCarryDev {
    double approx_end[GPSBB_MAX_CHAN];    /* channel i's carrier phase after the last block pushed so far: as
                                             k_chain_prefix of that push predicts it ...                          */
    double exact_end[GPSBB_MAX_CHAN];     /* ... and exactly, once its k_chain_fix has run                         */
};

#ifdef GPSBB_WG_TRACE
/* Measurement build only (make trace -> libgpsbb_trace.so; tools/corun_diag.py): every workgroup of the synthesis kernels leaves
 * one record — when it entered, had its tables staged, finished wavefront 0's tiles and left, in ticks of the 100 MHz
 * reference counter (s_memrealtime) AND in shader-clock cycles (s_memtime): cycles / ticks is the clock the chip ran at,
 * cycles / tile is what a wavefront's work cost in issue slots whatever the clock — and where it ran (HW_ID, XCC_ID).  The
 * buffer lives in device globals so that no kernel argument changes; gpsbb_test_wg_trace_begin / _read are the host side. */
constexpr int WG_TRACE_WORDS = 12;
__device__ unsigned long long *g_wg_trace;
__device__ unsigned g_wg_trace_cap, g_wg_trace_n;
struct WgTrace {
    unsigned long long wall0, clk0;
    int block;   /* the block the workgroup worked on (-1: a helper that found none) */
    int nblocks; /* blocks of the launch (the model kernels: workgroups from this number on are helpers); 0: not a synthesis kernel */
};
__device__ __forceinline__ WgTrace wg_trace_enter()
{
    WgTrace t;
    t.wall0 = wall_clock64();
    t.clk0 = clock64();
    t.block = (int)blockIdx.x;
    t.nblocks = 0;
    return t;
}
/* every lane of the workgroup calls it (a barrier: the workgroup leaves when its last wavefront does), except on the helpers'
 * early way out (!worked), where nobody waits for anybody */
__device__ __forceinline__ void wg_trace_leave(const WgTrace &t, unsigned long long wall_staged, unsigned long long clk_staged, unsigned tiles,
                                               unsigned kind, bool worked)
{
    const unsigned long long wall_loop = wall_clock64(), clk_loop = clock64();
    if (worked)
        __syncthreads();
    if (threadIdx.x == 0 && g_wg_trace) {
        const unsigned k = atomicAdd(&g_wg_trace_n, 1u);
        if (k < g_wg_trace_cap) {
            unsigned long long *r = g_wg_trace + (size_t)k * WG_TRACE_WORDS;
            r[0] = t.wall0;
            r[1] = wall_staged;
            r[2] = wall_loop;
            r[3] = wall_clock64();
            r[4] = t.clk0;
            r[5] = clk_staged;
            r[6] = clk_loop;
            r[7] = clock64();
            r[8] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) /* HW_REG_HW_ID */ |
                   ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) /* HW_REG_XCC_ID */ << 32);
            r[9] = (unsigned long long)(uint32_t)(t.block & 0xffffff) | ((unsigned long long)(t.nblocks && (int)blockIdx.x >= t.nblocks ? 1u : 0u) << 24) |
                   ((unsigned long long)tiles << 40);
            r[10] = kind | (worked ? 0x100u : 0u);
            r[11] = (unsigned long long)(uint32_t)t.nblocks | ((unsigned long long)gridDim.x << 32);
        }
    }
}
#endif

/* Everything the kernels need about one batch; passed by value as the kernel argument. */
struct BatchDev {
    const gpsbb_chan_t *ch;         /* [nblocks*nch] descriptors, block-major                        */
    int nblocks, nch, nsamp, ntiles;
    double delt;
    unsigned flags;
    const int32_t *tabs;            /* cos512[512] then sin512[512] (plutogpssim.c:93-161)           */
    const uint32_t *ca_bits;        /* [33][32] C/A chips per PRN, bit i of dword i>>5 = chip i      */
    SynRow *rows;                   /* row pool                                                      */
    const uint64_t *row_off;        /* [nblocks*nch + nvb*nch + 1] first row of each chain in the pool: the code chains
                                       (block*nch + channel), then the carrier chains (vb*nch + channel)  */
    int32_t *tile_row;              /* [nblocks][ntiles+1][2*nch]: row (relative to its chain) holding each
                                       tile's first sample, column 2*channel + kind; the 2*nch entries of
                                       one tile share a cache line; entry [ntiles] = the chain's last row */
    int32_t *row_cnt;               /* [nblocks*nch + nvb*nch] rows each chain produced (0 = inactive) */
    gpsbb_chan_state_t *end;        /* [nblocks*nch] end-of-block state                              */
    int32_t *tile_ctr;              /* [nblocks] next tile to hand out (zeroed before every k_synth)  */
    const uint32_t *kph0;           /* fixed-point carrier variant (GPSBB_FIXED_CARRIER): [nblocks*nch] 32-bit
                                       phase accumulator at the start of each block, else NULL          */
    const int32_t *kstep;           /* ... and its per-sample step (int)round(2^25*f_carr*delt), c:2675 */
    int seed_lanes;                 /* entries of seed_order */
    const int32_t *seed_order;      /* the chain each lane of k_seed walks: kind*nblocks*nch + block*nch + channel,
                                       -1 = idle lane (NULL, chain-only runs: carrier chains channel by channel, blocks
                                       in order, 64 consecutive blocks of one channel per wavefront).  Planned by the host: code and carrier chains never share
                                       a wavefront, carrier chains go by descending |f_carr| and the longest get
                                       wavefronts with few lanes (a wavefront runs as long as its longest chain
                                       and every extra lane adds turns of the loops its lanes do not share)   */
    uint32_t *status;               /* self-check word                                               */
    unsigned long long *digest;     /* [nblocks] or null: k_synth_ev_digest adds every block's digest as it renders (gpsbb_device_digest's
                                       number: GPSBB_PUSH_DIGEST) */
    unsigned long long *hazards;    /* [0] itable_512, [1] dwrd_oob, [2] lane-runs k_synth_ev recomputed exactly,
                                       [3] scratch (hazards seen by walks whose rows are not the final ones),
                                       [4] blocks k_chain_fix walked on its own, [5] wraps of falling phases whose
                                       "+ 1.0" was an exact tie (pass B)                                  */
    /* breakpoint kernel (k_synth_ev): instead of rows and a tile index, k_seed leaves the exact state of every
     * chain at the first sample of every tile */
    int ev;                         /* 1: this batch runs on k_synth_ev                               */
    int ev_chunk;                   /* consecutive tiles a wavefront of k_synth_ev takes at a time     */
    uint32_t pd_danger;             /* k_synth_pd: a model's low word below this sends the lane to the exact path (2 * PD_BAND) */
    double *tile_x;                 /* [nblocks][2*nch][ntiles]: row 2*channel = code phase (chips), 2*channel+1 =
                                       carrier phase * 512, at sample tile*TILE (tile-contiguous per chain: the
                                       pre-pass writes it coalesced, k_synth_ev reads it through the L2)  */
    uint32_t *tile_nav;             /* [nblocks][nch][ntiles]: bit 0 = the data bit in force is -1, bit 1 = the data
                                       bit after the next code roll-over is -1                         */
    const EvConst *evc;             /* [nblocks*nch]                                                  */
    int chain_dev;                  /* 1: GPSBB_CHAIN_CARRIER is resolved on the device (k_chain_prefix / k_chain_fix) */
    int chain_starts;               /* ... for the per-sample kernel: the chain kernels only put the exact start phase of every
                                       block into its descriptor (no rows, no offsets); k_seed then sees independent blocks */
    ChainAux *aux;                  /* [nvb*nch]                                                      */
    /* The device-side chain works on SEGMENTS: every block is cut into nseg pieces of seg_tiles tiles (the last one
     * shorter) that are chained like blocks, so that no walk is longer than a segment (a walk takes as long as its
     * chain, whatever the batch).  Segment sgi of block b is "virtual block" vb = b*nseg + sgi; aux / cd / start0 and the
     * carrier chains' regions of the row pool are indexed by vb*nch + channel.  nseg = 1: segments are blocks. */
    int nseg, seg_tiles, nvb;       /* nvb = nblocks*nseg */
    int model_start;                /* 1: pass B walks every segment from start0 (the host's drift model of the carrier): no
                                       pass A, no k_chain_prefix; 0: from ChainAux::start1                 */
    ChainDesc *cd;                  /* [nvb*nch] what the chain kernels read of the descriptors (and, chain_starts, where
                                       k_chain_fix* leaves the exact start phase of every block)        */
    const double *start0;           /* [nvb*nch] rough start phases (host: descriptor phase + sum of nsamp*step in plain
                                       double arithmetic): where pass A walks from                     */
    SynRow *prefix_rows;            /* [nvb*nch][CHAIN_PREFIX_CAP]: see ChainAux::prefix_cnt           */
    ChainCarryDev *carry;           /* stream: block 0 continues the previous push's last block; else NULL */
    /* k_chain_fix_par: one workgroup per (channel, chunk of FIXP_WG segments); chunk c hands the true end phase of its
     * last segment to chunk c + 1 through fix_end[channel * fix_chunks + c], published by fix_flag[...] = fix_epoch (a
     * number no earlier launch on these buffers used: nothing is cleared between launches) */
    unsigned long long *fix_end;
    int *fix_flag;
    int fix_epoch, fix_chunks;
    uint32_t cont0_mask;            /* ... for the channels of this mask (same prn as in that block: the host knows) */
    double *lap_end;                /* the lap-parallel chain alone (gpsbb_chain_carrier): [nblocks*nch] the carrier phase after every
                                       block, instead of `end`; `ch` is then NULL and the chain's descriptors are `cd` */
};

/* samples of segment sgi of a block (see BatchDev::nseg) */
__device__ __forceinline__ int seg_nsamp(const BatchDev &p, int sgi)
{
    const int n0 = sgi * p.seg_tiles * TILE;
    const int left = p.nsamp - n0;
    const int full = p.seg_tiles * TILE;
    return p.nseg <= 1 ? p.nsamp : (left < full ? left : full);
}

__device__ __forceinline__ size_t tile_row_at(const BatchDev &p, int b, int t, int i, int kind)
{
    return ((size_t)b * (p.ntiles + 1) + t) * (2 * p.nch) + 2 * i + kind;
}
__device__ __forceinline__ int chain_code(const BatchDev &p, int b, int i) { return b * p.nch + i; }
__device__ __forceinline__ int chain_carr(const BatchDev &p, int b, int i) { return p.nblocks * p.nch + b * p.nch + i; }

/* nav data bit (+1/-1) for packed counters, from 60 words at `dwrd` (plutogpssim.c:1781, 2732) */
template <class P>
__host__ __device__ __forceinline__ int nav_bit(const P dwrd, uint32_t nav)
{
    int w = nav_iword(nav);
    w = w < GPSBB_N_DWRD ? w : GPSBB_N_DWRD - 1; /* latent OOB of the reference: defined as dwrd[59] */
    return (int)((dwrd[w] >> (29 - nav_ibit(nav))) & 1u) * 2 - 1;
}

/* lane `src`'s 64-bit value, broadcast through scalar registers (src is wave-uniform) */
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

/* ---- k_seed -------------------------------------------------------------------------------------- */

struct RowSink {
    SynRow *rows;
    uint32_t cap; /* rows available, not counting the sentinel slot */
    uint32_t cnt;
    bool overflow;
    unsigned long long *hz;
    const uint32_t *dwrd; /* code chains: the channel's nav words; carrier chains: nullptr */
    uint32_t dbit;        /* code chains: 0x80000000 while the current data bit is -1 (refreshed by nav_fetch) */
    int32_t *tile_row;    /* the batch's tile index */
    uint32_t tr;          /* this chain's column in it: element of the next tile's entry (32-bit: the index is */
    uint32_t tstride;     /* sized well below 2^32 entries) ... and the distance to the one after             */
    int32_t tile_t;       /* that tile's number */
    int32_t ntiles;
    int32_t wrap_pend;    /* 0x80000000 while a row that follows a wrap has started in the tile before tile_t */

    __device__ __forceinline__ void row(int32_t n0, uint32_t nav, double x, double S, bool after_wrap)
    {
        {
            /* tiles that start before this row belong to the previous one (a tile that starts with it gets its
             * entry when the next row arrives).  Lanes run in lockstep, rows do not: a long row leaves several
             * tiles to fill in at once.  Entry e also carries, in bit 31, whether a row that follows a wrap
             * started in tile e-1. */
            const int32_t nt = (int32_t)(((uint32_t)n0 + (uint32_t)(TILE - 1)) / (uint32_t)TILE); /* tiles that start before n0 */
            const int32_t lim = nt < ntiles ? nt : ntiles;
            const int32_t here = (int32_t)(cnt < cap ? cnt : cap);
            while (tile_t < lim) {
                tile_row[tr] = (here - 1) | wrap_pend; /* the first entry written after a wrap row is its tile's successor */
                wrap_pend = 0;
                tr += tstride;
                tile_t++;
            }
            if (after_wrap) {
                if ((n0 & (TILE - 1)) == 0 && tile_t < ntiles) { /* the row opens a tile: its entry now, so
                                                                    that the flag below goes to the next one */
                    tile_row[tr] = here | wrap_pend;
                    tr += tstride;
                    tile_t++;
                }
                wrap_pend = (int32_t)0x80000000;
            }
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
    /* a sample whose carrier phase is exactly 1.0: table index 512, one past the reference's tables */
    __device__ __forceinline__ void table_index_512() { atomicAdd(hz, 1ull); }
    /* a data-bit boundary (c:2717-2733): the rows from here on carry the new bit */
    __device__ __forceinline__ void nav_fetch(uint32_t nav)
    {
        if (nav_iword(nav) >= GPSBB_N_DWRD)
            atomicAdd(hz + 1, 1ull);
        dbit = nav_bit(dwrd, nav) < 0 ? 0x80000000u : 0u;
    }
    __device__ __forceinline__ void finish()
    {
        if (cnt > cap)
            cnt = cap;
        for (; tile_t <= ntiles; tile_t++, tr += tstride) /* the remaining tiles and entry [ntiles] */
        {
            tile_row[tr] = ((int32_t)cnt - 1) | wrap_pend;
            wrap_pend = 0;
        }
        SynRow r;
        r.n0 = INT32_MAX; /* sentinel: terminates every forward scan */
        r.nav = 0;
        r.x = 0.0;
        r.S = 0.0;
        rows[cnt] = r;
    }
};

/*
 * What k_seed leaves for the breakpoint kernel: no rows, only the exact state of the chain at the first
 * sample of every tile — fma(tile start - n0, S, x) of the row that holds it (exact, see SynRow) — and, for
 * code chains, the data bit in force there and the one the next roll-over brings.  Same call sequence as
 * RowSink (build_rows_f64 drives either).
 */
struct TileSink {
    double *tx;         /* this chain's column of tile_x, at the next tile to write */
    uint32_t *tn;       /* code chains: this channel's column of tile_nav; carrier chains: nullptr */
    int32_t tile_t, ntiles;
    /* the row being walked */
    int32_t pn0;
    double px, pS;
    uint32_t pbits;
    uint32_t cnt;
    bool overflow; /* never: kept so that the chain drivers treat both sinks alike */
    unsigned long long *hz;
    unsigned long long hz_local[2];
    const uint32_t *dwrd;
    uint32_t dbit;

    GPSBB_HD void flush(int32_t upto) /* tiles that start before sample `upto` lie in the current row */
    {
        const int32_t nt = (int32_t)(((uint32_t)upto + (uint32_t)(TILE - 1)) / (uint32_t)TILE);
        const int32_t lim = nt < ntiles ? nt : ntiles;
#ifdef GPSBB_EXP_NOFLUSH
        tile_t = lim;
#endif
        while (tile_t < lim) {
            *tx++ = fma_rn((double)(tile_t * TILE - pn0), pS, px);
            if (tn)
                *tn++ = pbits;
            tile_t++;
        }
    }
    GPSBB_HD void row(int32_t n0, uint32_t nav, double x, double S, bool)
    {
        if (cnt)
            flush(n0);
        pn0 = n0;
        if (dwrd) {
            px = x;
            pS = S;
            /* the data bit after the next roll-over (c:2717-2733): a new one only when icode rolls over too */
            const uint32_t nav1 = nav_advance(nav);
            const uint32_t nxt = nav_icode(nav1) == 0 ? (nav_bit(dwrd, nav1) < 0 ? 2u : 0u) : (dbit ? 2u : 0u);
            pbits = (dbit ? 1u : 0u) | nxt;
        } else {
            px = mul_rn(x, 512.0);
            pS = mul_rn(S, 512.0);
            pbits = 0;
        }
        cnt++;
    }
    GPSBB_HD void table_index_512()
    {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(hz, 1ull);
#else
        hz_local[0]++;
#endif
    }
    GPSBB_HD void nav_fetch(uint32_t nav)
    {
        if (nav_iword(nav) >= GPSBB_N_DWRD) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicAdd(hz + 1, 1ull);
#else
            hz_local[1]++;
#endif
        }
        dbit = nav_bit(dwrd, nav) < 0 ? 0x80000000u : 0u;
    }
    GPSBB_HD void finish() { flush(INT32_MAX - TILE); }
};

/* a TileSink for chain (b, i, kind) of a batch whose tile arrays start at tile_x / tile_nav */
GPSBB_HD TileSink make_tile_sink(double *tile_x, uint32_t *tile_nav, int nch, int ntiles, int b, int i, int kind,
                                 const uint32_t *dwrd, uint32_t nav0, unsigned long long *hz)
{
    TileSink s;
    s.tx = tile_x + ((size_t)b * (2 * (size_t)nch) + 2 * i + kind) * (size_t)ntiles;
    s.tn = kind == 0 ? tile_nav + ((size_t)b * (size_t)nch + i) * (size_t)ntiles : nullptr;
    s.tile_t = 0;
    s.ntiles = ntiles;
    s.pn0 = 0;
    s.px = s.pS = 0.0;
    s.pbits = 0;
    s.cnt = 0;
    s.overflow = false;
    s.hz = hz;
    s.hz_local[0] = s.hz_local[1] = 0;
    s.dwrd = dwrd;
    s.dbit = dwrd && nav_bit(dwrd, nav0) < 0 ? 0x80000000u : 0u;
    return s;
}

/* what phase 2 of k_seed needs to know about the chain a lane has just built */
struct ChainDone {
    const SynRow *rows;
    int cnt; /* 0 = this lane built nothing */
};

__device__ __forceinline__ RowSink make_sink(const BatchDev &p, int chain, const uint32_t *dwrd, uint32_t nav0)
{
    RowSink s;
    const uint64_t o0 = p.row_off[chain], o1 = p.row_off[chain + 1];
    s.rows = p.rows + o0;
    s.cap = (uint32_t)(o1 - o0 - 1);
    s.cnt = 0;
    s.overflow = false;
    s.hz = p.hazards;
    s.dwrd = dwrd;
    s.dbit = dwrd && nav_bit(dwrd, nav0) < 0 ? 0x80000000u : 0u;
    const int nbc = p.nblocks * p.nch;
    const int kind = chain >= nbc ? 1 : 0, bi = chain - kind * nbc;
    s.tile_row = p.tile_row;
    s.tr = (uint32_t)tile_row_at(p, bi / p.nch, 0, bi % p.nch, kind);
    s.tstride = 2u * (uint32_t)p.nch;
    s.tile_t = 0;
    s.ntiles = p.ntiles;
    s.wrap_pend = 0;
    return s;
}

/* EV: the batch runs on the breakpoint kernel — tile states (TileSink) instead of rows + tile index (RowSink) */
template <bool EV>
__device__ inline ChainDone seed_code_chain(const BatchDev &p, int b, int i)
{
    const gpsbb_chan_t &c = p.ch[(size_t)b * p.nch + i];
    gpsbb_chan_state_t &e = p.end[(size_t)b * p.nch + i];
    ChainDone d = {nullptr, 0};
    if (c.prn <= 0) {
        e.code_phase = 0.0;
        e.iword = e.ibit = e.icode = e.dataBit = e.codeCA = 0;
        e._pad = 0;
        return d;
    }
    const int chain = chain_code(p, b, i);
    uint32_t nav = nav_pack(c.icode, c.ibit, c.iword);
    const double s = mul_rn(c.f_code, p.delt); /* plutogpssim.c:2709: f_code * delt, rounded on its own */
    double x;
    if (EV) {
        TileSink sink = make_tile_sink(p.tile_x, p.tile_nav, p.nch, p.ntiles, b, i, 0, c.dwrd, nav, p.hazards);
        x = build_rows_f64<NCO_CODE>(c.code_phase, s, nav, p.nsamp, sink);
        sink.finish();
        d.cnt = (int)sink.cnt;
    } else {
        RowSink sink = make_sink(p, chain, c.dwrd, nav);
        x = build_rows_f64<NCO_CODE>(c.code_phase, s, nav, p.nsamp, sink);
        sink.finish();
        if (sink.overflow)
            atomicOr(p.status, ST_ROW_OVERFLOW);
        d.rows = sink.rows;
        d.cnt = (int)sink.cnt;
    }
    e.code_phase = x;
    e.iword = nav_iword(nav);
    e.ibit = nav_ibit(nav);
    e.icode = nav_icode(nav);
    e.dataBit = nav_bit(c.dwrd, nav);
    const int ci = (int)x;
    e.codeCA = (int)((p.ca_bits[c.prn * 32 + (ci >> 5)] >> (ci & 31)) & 1u) * 2 - 1; /* c:2737 */
    e._pad = 0;
    return d;
}

template <bool EV>
__device__ inline ChainDone seed_carr_chain(const BatchDev &p, int b, int i, double x0, double *x_end)
{
    const gpsbb_chan_t &c = p.ch[(size_t)b * p.nch + i];
    gpsbb_chan_state_t &e = p.end[(size_t)b * p.nch + i];
    ChainDone d = {nullptr, 0};
    if (c.prn <= 0) {
        e.carr_phase = 0.0;
        *x_end = 0.0;
        return d;
    }
    const int chain = chain_carr(p, b, i);
    uint32_t nav = 0;
    const double s = mul_rn(c.f_carr, p.delt); /* plutogpssim.c:2741 */
    double x;
    if (EV) {
        TileSink sink = make_tile_sink(p.tile_x, p.tile_nav, p.nch, p.ntiles, b, i, 1, nullptr, 0u, p.hazards);
        x = build_rows_f64<NCO_CARR>(x0, s, nav, p.nsamp, sink);
        sink.finish();
        d.cnt = (int)sink.cnt;
    } else {
        RowSink sink = make_sink(p, chain, nullptr, 0u);
        x = build_rows_f64<NCO_CARR>(x0, s, nav, p.nsamp, sink);
        sink.finish();
        if (sink.overflow)
            atomicOr(p.status, ST_ROW_OVERFLOW);
        d.rows = sink.rows;
        d.cnt = (int)sink.cnt;
    }
    e.carr_phase = x;
    *x_end = x;
    return d;
}

/* fixed-point carrier variant on the model kernel (k_synth_pd): the table index at the first sample of tile t as a
 * double, fraction included — (phase / 2^16) modulo 512 with phase = start + t*TILE*step modulo 2^32 (c:2699, 2748) */
GPSBB_HD double fixed_tile_index(uint32_t ph0, int32_t step, int t)
{
    const uint32_t ph = ph0 + (uint32_t)t * (uint32_t)TILE * (uint32_t)step;
    return (double)(ph & 0x1ffffffu) * 0x1p-16;
}

/* fixed-point carrier variant: the phase after the block is start + nsamp*step modulo 2^32 (c:2748) */
__device__ inline void seed_carr_fixed(const BatchDev &p, int b, int i)
{
    const size_t k = (size_t)b * p.nch + i;
    p.row_cnt[chain_carr(p, b, i)] = 0;
    p.end[k].carr_phase = p.ch[k].prn > 0 ? (double)(uint32_t)(p.kph0[k] + (uint32_t)p.nsamp * (uint32_t)p.kstep[k]) : 0.0;
}

/* One lane per chain as planned in seed_order.  Writes each chain's rows, end state and row count, and — as
 * the rows are emitted — the tile index: tile_row[t] = the row holding sample t*TILE (t = 0..ntiles-1),
 * [ntiles] = the last row. */
#ifndef GPSBB_SEED_WG
#define GPSBB_SEED_WG 256
#endif
#ifndef GPSBB_SEED_PRIO
#define GPSBB_SEED_PRIO 3
#endif
/* Four wavefronts per workgroup (one per SIMD of a CU): measured best trade between the pre-pass's own
 * speed (chains sharing a SIMD slow each other by ~1/3) and how much it disturbs the previous run's k_synth,
 * which it overlaps. */
template <bool EV>
__global__ __launch_bounds__(GPSBB_SEED_WG) void k_seed(BatchDev p)
{
    /* the chain walk is a long dependent instruction stream: let it issue whenever it is ready (it uses a
     * small fraction of the issue slots, so the co-resident synthesis wavefronts hardly notice) */
    __builtin_amdgcn_s_setprio(GPSBB_SEED_PRIO);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.seed_lanes)
        return;
    const int c = p.seed_order[gid];
    if (c < 0)
        return;
    const int nbc = p.nblocks * p.nch;
    if (c < nbc) {
        const ChainDone d = seed_code_chain<EV>(p, c / p.nch, c % p.nch);
        if (!EV)
            p.row_cnt[chain_code(p, c / p.nch, c % p.nch)] = d.cnt;
        return;
    }
    /* carrier chains: every block starts from its descriptor's carr_phase.  Blocks that continue each
     * other (GPSBB_CHAIN_CARRIER) had their start phases resolved exactly on the host when the batch was
     * set up, so all chains are independent here. */
    const int g = c - nbc, b = g / p.nch, i = g % p.nch;
    if (p.kph0) {
        seed_carr_fixed(p, b, i);
        return;
    }
    double unused;
    const ChainDone d = seed_carr_chain<EV>(p, b, i, p.ch[(size_t)b * p.nch + i].carr_phase, &unused);
    if (!EV)
        p.row_cnt[chain_carr(p, b, i)] = d.cnt;
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

/* a wavefront's private slice of LDS: the rows of every chain that overlap its current tile, one slot per
 * row, chain after chain.  Filled by LDS-DMA (global_load_lds_dwordx4 / _dword): lane k of a load owns
 * slot k, nothing passes through registers and the copy of the next tile's rows runs under the current
 * tile's last channel. */
struct __attribute__((aligned(16))) WaveRows {
    uint4 a[WAVE_ROW_CAP];       /* {n0, nav, x.lo, x.hi} */
    uint32_t s_lo[WAVE_ROW_CAP]; /* S */
    uint32_t s_hi[WAVE_ROW_CAP];
};

/* LDS image of one workgroup (dynamic shared memory, 16-byte aligned carve) */
struct SynthLds {
    uint32_t amp[GPSBB_MAX_CHAN][512];          /* int16x2: lo = I (cos*gain), hi = Q (sin*gain)      */
    int8_t chip[GPSBB_MAX_CHAN][1024];          /* codeCA as +1/-1 per chip (plutogpssim.c:2737)       */
    uint32_t dwrd[GPSBB_MAX_CHAN][GPSBB_N_DWRD]; /* nav words                                          */
    double sc[GPSBB_MAX_CHAN];                  /* f_code*delt                                         */
    double rsc[GPSBB_MAX_CHAN];                 /* 1/sc where a run of SPT samples holds at most one chip boundary
                                                   (sc*(SPT-1) < 1, i.e. sample rates above ~15.4 MS/s), else 0 */
    double sk512[GPSBB_MAX_CHAN];               /* f_carr*delt*512 (carrier phase is walked scaled by 512: exact) */
    WaveRows wr[WAVES_PER_WG];
    uint64_t roff[2 * GPSBB_MAX_CHAN];          /* first pool row of chain (a, kind) of this block     */
    int32_t act[GPSBB_MAX_CHAN];
    int32_t nact;
};

/* state of one NCO at sample n, scanning forward from the first of `rows` (global-memory fallback) */
__device__ __forceinline__ double row_state_global(const SynRow *__restrict__ rows, int n, uint32_t *nav)
{
    int r = 0;
    while (rows[r + 1].n0 <= n)
        r++;
    const SynRow row = rows[r];
    *nav = row.nav;
    return __fma_rn((double)(n - row.n0), row.S, row.x);
}

__device__ __forceinline__ double hi_lo_f64(int hi, int lo) { return __hiloint2double(hi, lo); }

/* a 64-bit value known to be equal in all lanes, moved to scalar registers */
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

/* ---- wavefront-level helpers -------------------------------------------------------------------- */

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

/* inclusive prefix sum over the 64 lanes, all in the VALU's data-parallel-primitive path: four shifts inside
 * each row of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3 */
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += dpp_or_zero<0x111, 0xf>(v); /* row_shr:1 */
    v += dpp_or_zero<0x112, 0xf>(v); /* row_shr:2 */
    v += dpp_or_zero<0x114, 0xf>(v); /* row_shr:4 */
    v += dpp_or_zero<0x118, 0xf>(v); /* row_shr:8 */
    v += dpp_or_zero<0x142, 0xa>(v); /* row_bcast:15 into rows 1 and 3 */
    v += dpp_or_zero<0x143, 0xc>(v); /* row_bcast:31 into rows 2 and 3 */
    return v;
}

/* lane `src` (per-lane index) of a 64-bit value */
__device__ __forceinline__ uint64_t bpermute_u64(uint64_t v, int src)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

/* one row per lane, HBM -> LDS without touching registers: the lane's row lands in slot (first + lane) */
__device__ __forceinline__ void dma_row(uint64_t src, WaveRows &W, int first)
{
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)src, (lptr_t)&W.a[first], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + 16), (lptr_t)&W.s_lo[first], 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + 20), (lptr_t)&W.s_hi[first], 4, 0, 0);
}

/* The chain that owns slot k: the number of chains whose slot range ends at or before k, i.e. the number of
 * lanes c with incl[c] <= k.  incl is non-decreasing over the lanes, so a branch-free binary search does it
 * in six steps, each fetching another lane's value with ds_bpermute. */
__device__ __forceinline__ int chain_of_slot(int incl, int k)
{
    int lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
        const int t = __builtin_amdgcn_ds_bpermute((lo + step - 1) << 2, incl);
        lo += t <= k ? step : 0;
    }
    return lo;
}

/* what plan_tile() tells about a tile: to chain lane c its slot range in the slice (base, cnt), to slot lane k
 * the address of its row (a0 for slot k, a1 for slot 64+k), to all the number of rows R; ok = 0: the tile
 * has more rows than the slice holds */
struct Staged {
    int base, cnt, ok, R;
    uint64_t a0, a1;
};

/*
 * Plan the copy of one tile's rows into the wavefront's slice.  Lane c is chain c of the block; its rows
 * trA..trB overlap the tile.  The chains' slot ranges come from a prefix sum over the lanes, slot k's chain
 * from chain_of_slot().  Not inlined (it is needed in every variant of the channel loop, once per tile) —
 * and for that reason it only plans: a function waits for its outstanding memory operations before it
 * returns, so the copy itself (issue_tile) is issued by the caller and stays asynchronous.
 */
__device__ __noinline__ Staged plan_tile(int lane, bool has_chain, uint64_t row0_addr, int trA, int trB)
{
    Staged st;
    st.cnt = has_chain ? trB - trA + 1 : 0;
    const int incl = wave_incl_scan(st.cnt);
    st.base = incl - st.cnt;
    st.R = __builtin_amdgcn_readlane(incl, 63);
    st.ok = st.R <= WAVE_ROW_CAP;
    st.a0 = st.a1 = 0;
    if (!st.ok)
        return st;
    /* slot k holds row trA + (k - base) of its chain, i.e. address [row 0 + (trA - base) rows] + k rows */
    const uint64_t addr_c = row0_addr + (uint64_t)((int64_t)(trA - st.base) * (int64_t)sizeof(SynRow));
    st.a0 = bpermute_u64(addr_c, chain_of_slot(incl, lane)) + (uint64_t)lane * sizeof(SynRow);
    if (st.R > 64)
        st.a1 = bpermute_u64(addr_c, chain_of_slot(incl, lane + 64)) + (uint64_t)(lane + 64) * sizeof(SynRow);
    return st;
}

/* start the planned copy: one row per lane, HBM -> LDS, nothing waits */
__device__ __forceinline__ void issue_tile(WaveRows &W, int lane, const Staged &st)
{
    if (st.ok) {
        if (lane < st.R)
            dma_row(st.a0, W, 0);
        if (st.R > 64 && lane + 64 < st.R)
            dma_row(st.a1, W, 64);
    }
}

/* A tile whose rows do not fit the slice (dense rows: high Doppler at a low sample rate): copy just one
 * channel's two chains (nc + nk <= WAVE_ROW_CAP rows from ac / ak) and wait for them.  Returns the first
 * samples of slots lane and 64+lane. */
struct SlotN0 {
    int a, b;
};
__device__ __noinline__ SlotN0 stage_channel(WaveRows &W, int lane, uint64_t ac, uint64_t ak, int nc, int nk)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the previous channel's lookups are done */
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int k = lane + 64 * h;
        if (k < nc + nk)
            dma_row(k < nc ? ac + (uint64_t)k * sizeof(SynRow) : ak + (uint64_t)(k - nc) * sizeof(SynRow), W, 64 * h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SlotN0 r;
    r.a = (int)W.a[lane].x;
    r.b = (int)W.a[64 + lane].x;
    return r;
}

/* ... and the same copy into one half of the slice (slots first .. first+63, nc + nk <= 64) without waiting:
 * in such tiles the next channel's rows are fetched while the current channel is walked.  Inlined: see
 * plan_tile() */
__device__ __forceinline__ void prefetch_channel(WaveRows &W, int lane, uint64_t ac, uint64_t ak, int nc, int nk, int first)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* lookups in this half are done */
    if (lane < nc + nk)
        dma_row(lane < nc ? ac + (uint64_t)lane * sizeof(SynRow) : ak + (uint64_t)(lane - nc) * sizeof(SynRow), W, first);
}

/*
 * State of one chain at sample n (this lane's first sample).  The chain's rows sit in slots
 * sbase .. sbase+scnt-1 and their first samples in slot_n0a / slot_n0b (lane k = slot k / 64+k): the
 * lane's row is found by counting the rows that start at or before n, each compared through a scalar
 * register, then fetched with one 16-byte and two 4-byte LDS reads.
 */
__device__ __forceinline__ double chain_state(const WaveRows &W, int sbase, int scnt, int slot_n0a, int slot_n0b, int n,
                                              uint32_t *nav, double *S_out)
{
    int slot = sbase;
    for (int j = 1; j < scnt; j++) {
        const int k = sbase + j;
        const int t = k < 64 ? __builtin_amdgcn_readlane(slot_n0a, k) : __builtin_amdgcn_readlane(slot_n0b, k - 64);
        slot += t <= n;
    }
    const uint4 r = W.a[slot];
    /* &s_lo[slot], written so that the compiler does not derive it from &a[slot] with a 64-bit multiply-add */
    uint32_t sa;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(sa) : "v"(slot), "v"((uint32_t)(uintptr_t)(lptr_t)&W.s_lo[0]));
    typedef __attribute__((address_space(3))) const uint32_t *lds_u32_t;
    const double S = hi_lo_f64((int)((lds_u32_t)(uintptr_t)sa)[WAVE_ROW_CAP], (int)((lds_u32_t)(uintptr_t)sa)[0]);
    *nav = r.y;
    *S_out = S;
    return __fma_rn((double)(n - (int)r.x), S, hi_lo_f64((int)r.w, (int)r.z));
}

/*
 * SPT consecutive samples of one channel.  CODEW = false / CARR = 0 are the straight-line versions used when
 * no lane of the wavefront can reach a code / carrier wrap inside its run (decided by the caller): that
 * NCO's update is then a single IEEE add.  With CODEW / CARR = 1 it is the reference's full update
 * (plutogpssim.c:2709-2746) with the comparisons done on the high dword of the double.  CARR = 2 is the
 * reference's fixed-point carrier (32-bit accumulator, c:2699/2748).
 */
#ifndef GPSBB_WALK_G
#define GPSBB_WALK_G 8
#endif
constexpr int WALK_G = GPSBB_WALK_G; /* samples whose LDS lookups are in flight together */

/* dataBit handling inside one run: a run of SPT samples crosses at most one code-period boundary
 * (a period is >= 666 samples under the contract f_code*delt <= 1.5), hence at most one data-bit change:
 * samples j < jw use dbx0, the others dbx1 (XOR masks, see walk_channel). */
struct RunNav {
    uint32_t nav;
    int dbx0, dbx1, jw;
};

/* phase 1 of a group: table indices of WALK_G samples, both NCOs advanced (two chains of IEEE adds).
 * CODE: 0 straight line, 1 with the 1023 wrap, 2 no per-sample code work at all (see walk_channel). */
template <int CODE, int CARR>
__device__ __forceinline__ void walk_indices(const SynthLds &L, int i, double sc, double sk, double &xc, double &yk,
                                             uint32_t &ph, uint32_t kstep, RunNav &rn, int (&it)[WALK_G],
                                             int (&ci)[WALK_G], int jbase, uint32_t wlim, int wadj)
{
    constexpr bool CARRW = CARR == 1;
#pragma unroll
    for (int u = 0; u < WALK_G; u++) {
        if (CARR == 2) {
            it[u] = (int)((ph >> 16) & 0x1ffu); /* fixed-point variant: 9-bit index, c:2699 */
            ph += kstep;                       /* c:2748 */
        } else {
            it[u] = (int)yk; /* floor(carr_phase*512), c:2697 (yk = carr_phase*512 >= 0) */
        }
        if (CARRW)
            it[u] &= 511; /* carr_phase == 1.0: latent OOB of the reference, defined as &511 (counted where the
                             rows are built: such a state is always the first sample of a row) */
        if (CODE != 2) {
            ci[u] = (int)xc;     /* c:2737 */
            xc = add_rn(xc, sc); /* c:2709 */
        }
        if (CARR != 2)
            yk = add_rn(yk, sk); /* c:2741, scaled by 512 */
        if (CODE == 1) {
            /* c:2711-2734 without a branch: a run holds at most one roll-over (a period is >= 666 samples), the
             * data bit that follows it (rn.dbx1) was looked up before the walk; here only where it happens */
            const bool w = __double2hiint(xc) >= 0x408FF800; /* xc >= 1023.0 (xc >= 0) */
            xc = add_rn(xc, hi_lo_f64(w ? (int)0xC08FF800 : 0, 0)); /* -1023.0 or +0.0 */
            rn.jw = w ? jbase + u + 1 : rn.jw;
        }
        if (CARRW) {
            /* c:2743-2746.  Only one of the two wraps can fire for a given sign of the step, so one unsigned
             * compare of the high dword does it: step >= 0: yk >= 512.0 (wlim 0x40800000, then -512);
             * step < 0: yk < 0 (sign bit set, wlim 0x80000000, then +512).  Adding +0.0 is exact. */
            const uint32_t h = (uint32_t)__double2hiint(yk);
            yk = add_rn(yk, hi_lo_f64(h >= wlim ? wadj : 0, 0));
        }
    }
}

/*
 * SPT consecutive samples of one channel.  CODE = 0 / CARR = 0 are the straight-line versions used when
 * no lane of the wavefront can reach a code / carrier wrap inside its run (decided by the caller): that
 * NCO's update is then a single IEEE add.  With CODE / CARR = 1 it is the reference's full update
 * (plutogpssim.c:2709-2746) with the comparisons done on the high dword of the double.  CARR = 2 is the
 * reference's fixed-point carrier (32-bit accumulator, c:2699/2748).
 *
 * CODE = 2: the whole tile lies in one row of the code chain (state at step j of the run = fma(j, Sc, xc)
 * exactly) and a run advances by less than one chip, so it sees at most one chip boundary.  The code NCO is
 * then not stepped at all: the boundary's position jc (first j with floor(state) > floor(xc)) comes from
 * one approximate division, biased upward and settled by one exact FMA + compare, and every sample selects
 * one of two pre-signed chips with a compare against its (compile-time) position.
 *
 * Software-pipelined by hand: the LDS reads of group k are issued, then the indices of group k+1 are
 * computed (pure VALU, covers the LDS latency), then group k is accumulated.
 */
template <int CODE, int CARR>
__device__ __forceinline__ void walk_channel(const SynthLds &L, int i, double xc, double Sc, double yk, uint32_t ph,
                                             uint32_t kstep, uint32_t nav, int dbx0, v2s (&acc)[SPT])
{
    constexpr int G = WALK_G;
    const double sc = L.sc[i], sk = L.sk512[i];
    /* wrap-capable carrier update: which wrap the sign of the step allows (wave-uniform) */
    const bool down = __builtin_amdgcn_readfirstlane(__double2hiint(sk)) < 0;
    const uint32_t wlim = down ? 0x80000000u : 0x40800000u;
    const int wadj = down ? 0x40800000 : (int)0xC0800000;
    const uint32_t *__restrict__ amp = L.amp[i];
    const int8_t *__restrict__ chip = L.chip[i];
    /* codeCA*dataBit: chip sign (+1/-1) XOR-ed with 0xfffe when dataBit = -1 flips +-1 in 16 bits */
    RunNav rn;
    rn.nav = nav;
    rn.dbx0 = rn.dbx1 = dbx0;
    rn.jw = SPT;
    if (CODE == 1) {
        const uint32_t nav1 = nav_advance(nav); /* counters after the run's one possible roll-over */
        if (nav_icode(nav1) == 0)               /* c:2717-2733: a new data bit from the next sample on */
            rn.dbx1 = nav_bit(L.dwrd[i], nav1) < 0 ? 0xfffe : 0;
    }
    int jc = 0;
    int sg_a = 0, sg_b = 0;
    if (CODE == 2) {
        const int c0 = (int)xc;                 /* chip index of the run's first sample, c:2737 */
        const double cn = (double)(c0 + 1);      /* the boundary the run may cross */
        /* r = (cn - xc)/Sc steps to the boundary; q >= r by the bias (the quotient is good to ~2^-35:
         * 1/sc instead of 1/Sc, three roundings), so k = floor(q) is floor(r) or, just below an integer,
         * floor(r)+1 = ceil(r); the exact state at step k decides */
        const double kf = floor(__fma_rn(add_rn(cn, -xc), L.rsc[i], 0x1p-30));
        jc = (int)kf + (__fma_rn(kf, Sc, xc) >= cn ? 0 : 1);
        sg_a = (int)chip[c0] ^ dbx0;
        sg_b = (int)chip[c0 + 1] ^ dbx0;
        /* keep the two signed chips as they are: the compiler would otherwise select first and XOR per sample */
        asm volatile("" : "+v"(sg_a), "+v"(sg_b));
    }
    int it[G], ci[G];
    walk_indices<CODE, CARR>(L, i, sc, sk, xc, yk, ph, kstep, rn, it, ci, 0, wlim, wadj);
#pragma unroll
    for (int j0 = 0; j0 < SPT; j0 += G) {
        __builtin_amdgcn_sched_barrier(0);
        /* phase 2: 2*G LDS reads issued back to back */
        uint32_t av[G];
        int cv[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
            av[u] = amp[it[u]];
            if (CODE != 2)
                cv[u] = (int)chip[ci[u]];
        }
        __builtin_amdgcn_sched_barrier(0);
        /* phase 1 of the next group while the reads are in flight */
        if (j0 + G < SPT)
            walk_indices<CODE, CARR>(L, i, sc, sk, xc, yk, ph, kstep, rn, it, ci, j0 + G, wlim, wadj);
        __builtin_amdgcn_sched_barrier(0);
        /* phase 3: acc += amp * (codeCA*dataBit), packed int16x2 (c:2701-2706) */
#pragma unroll
        for (int u = 0; u < G; u++) {
            short sg;
            if (CODE == 2) {
                sg = (short)(j0 + u < jc ? sg_a : sg_b);
            } else {
                const int dbx = CODE == 1 ? (j0 + u < rn.jw ? rn.dbx0 : rn.dbx1) : rn.dbx0;
                sg = (short)(cv[u] ^ dbx);
            }
            acc[j0 + u] += u32_v2s(av[u]) * v2s{sg, sg};
        }
    }
}

__global__ __launch_bounds__(TILE_THREADS, GPSBB_WAVES_PER_SIMD) void k_synth(BatchDev p, int16_t *__restrict__ iq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    SynthLds &L = *reinterpret_cast<SynthLds *>(smem_raw);

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
    if (tid < p.nch) {
        const double sc = mul_rn(cb[tid].f_code, p.delt);
        const double sk = mul_rn(mul_rn(cb[tid].f_carr, p.delt), 512.0);
        L.sc[tid] = sc;
        L.rsc[tid] = (sc >= 0x1p-10 && sc * (double)(SPT - 1) < 1.0) ? 1.0 / sc : 0.0;
        L.sk512[tid] = sk;
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
        L.amp[i][k] = v;
    }
    for (int e = tid; e < p.nch * 32; e += TILE_THREADS) { /* one dword of 32 chips per lane */
        const int i = e >> 5, wd = e & 31;
        const int prn = cb[i].prn;
        const uint32_t w = prn > 0 ? p.ca_bits[prn * 32 + wd] : 0u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(&L.chip[i][wd * 32]);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                v |= (((w >> (4 * q + k)) & 1u) ? 0x01u : 0xffu) << (8 * k);
            dst[q] = v;
        }
    }
    for (int e = tid; e < p.nch * GPSBB_N_DWRD; e += TILE_THREADS) {
        const int i = e / GPSBB_N_DWRD, w = e % GPSBB_N_DWRD;
        L.dwrd[i][w] = cb[i].dwrd[w];
    }
    if (tid < 2 * p.nch) {
        const int i = tid >> 1;
        L.roff[tid] = p.row_off[(tid & 1) ? chain_carr(p, b, i) : chain_code(p, b, i)];
    }
    __syncthreads();
    const int nact = L.nact;

    /* ---- from here on every wavefront works alone: chunks of consecutive tiles, no workgroup barrier ---- */
    const int wave = tid >> 6, lane = tid & 63;
    WaveRows &W = L.wr[wave];
    const int ntw = p.ntiles;
    /* lane c serves chain c = (channel c>>1, kind c&1) of this block */
    const bool fixed_carr = p.kph0 != nullptr; /* fixed-point carrier variant: carrier chains have no rows */
    const bool has_chain = lane < 2 * nact && !(fixed_carr && (lane & 1));
    const int my_chan = has_chain ? L.act[lane >> 1] : 0;
    const int32_t *__restrict__ lane_tr = p.tile_row + tile_row_at(p, b, 0, my_chan, lane & 1);
    const size_t tstride = 2 * (size_t)p.nch; /* the 2*nch entries of one tile are contiguous: one cache line */
    const uint64_t row0_addr = (uint64_t)(uintptr_t)(p.rows + L.roff[has_chain ? 2 * my_chan + (lane & 1) : 0]);
    /* code chain lanes: a run of this channel holds at most one chip boundary (see walk_channel, CODE = 2) */
    const bool one_chip = has_chain && !(lane & 1) && L.rsc[my_chan] != 0.0;
    constexpr int TR_ROW = 0x7fffffff; /* tile index entry: row number; bit 31: a wrap in the previous tile */

    /* Tiles are handed out dynamically in chunks of TILE_CHUNK consecutive tiles from a per-block counter:
     * a wavefront that shares its SIMD with another kernel (the next run's seeding pre-pass runs
     * concurrently) simply takes fewer chunks instead of stretching the whole launch. */
    for (;;) {
        int chunk = 0;
        if (lane == 0)
            chunk = atomicAdd(&p.tile_ctr[b], TILE_CHUNK);
        const int wt_begin = __builtin_amdgcn_readfirstlane(chunk);
        if (wt_begin >= ntw)
            break;
        const int wt_end = wt_begin + TILE_CHUNK < ntw ? wt_begin + TILE_CHUNK : ntw;

        /* tile index entries of this chain: the row holding the first sample of tile wt, wt+1, wt+2 */
        int trA = 0, trB = 0, trC = 0;
        if (has_chain) {
            trA = lane_tr[wt_begin * tstride];
            trB = lane_tr[(wt_begin + 1) * tstride];
            trC = lane_tr[(wt_begin + 2 <= ntw ? wt_begin + 2 : ntw) * tstride];
        }
        /* slots of this chain's rows in the slice for the current tile; ovf: the tile's rows do not fit;
         * cwrap: one of the chain's rows that start in the tile follows a wrap */
        Staged cur = plan_tile(lane, has_chain, row0_addr, trA & TR_ROW, trB & TR_ROW);
        issue_tile(W, lane, cur);
        bool cwrap = trB < 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int slot_n0a = (int)W.a[lane].x, slot_n0b = (int)W.a[64 + lane].x;

        for (int wt = wt_begin; wt < wt_end; wt++) {
            const bool more = wt + 1 < wt_end;
            const int wn0 = wt * TILE;       /* first sample of the tile (wave-uniform) */
            const int n0 = wn0 + lane * SPT; /* this lane's run; lanes past the block end compute and store nothing */
            const int nvalid = p.nsamp - n0 < SPT ? p.nsamp - n0 : SPT;
            const int cbase = cur.base, ccnt = cur.cnt;
            const bool ovf = !cur.ok;
            Staged nxt = {0, 0, 1, 0, 0, 0};
            bool nwrap = false;
            v2s acc[SPT];
#pragma unroll
            for (int j = 0; j < SPT; j++)
                acc[j] = v2s{0, 0};

            /* Which walk each channel needs follows from its two chains' rows alone: no row starts inside the
             * tile -> no wrap (and, for the code, the one-boundary walk where it applies); a row that follows a
             * wrap starts inside it (the tile index says so) -> the wrap-capable update; else straight line.
             * Channels are taken variant by variant, so that inside each loop the accumulators stay where they
             * are (sums modulo 2^16 do not care about the order). */
            const int kcode = ccnt == 1 ? (one_chip ? 2 : 0) : (cwrap ? 1 : 0);
            const bool code_lane = has_chain && !(lane & 1);
            const uint64_t m_c0 = __ballot(code_lane && kcode == 0), m_c1 = __ballot(code_lane && kcode == 1),
                           m_c2 = __ballot(code_lane && kcode == 2);
            const uint64_t m_k1 = __ballot(has_chain && (lane & 1) && cwrap) >> 1; /* on the channel's even bit */
            int remaining = nact;

            /* one channel: its start states from the rows in the slice (sb_* / sc_*: first slot and number of
             * rows of its code / carrier chain), then the walk */
#define GPSBB_STAGE_NEXT_TILE                                                                                          \
    {                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this tile's lookups have returned */                     \
        nxt = plan_tile(lane, has_chain, row0_addr, trB & TR_ROW, trC & TR_ROW);                                       \
        issue_tile(W, lane, nxt);                                                                                      \
        nwrap = trC < 0;                                                                                               \
        trA = trB;                                                                                                     \
        trB = trC;                                                                                                     \
        if (has_chain && wt + 3 <= ntw)                                                                                \
            trC = lane_tr[(wt + 3) * tstride];                                                                         \
    }
            /* BEFORE_WALK: what to start between the lookups and the walk (the next copy into the slice) */
#define GPSBB_LOOKUP_AND_WALK(CODE, CARR, BEFORE_WALK)                                                                 \
    {                                                                                                                  \
        uint32_t nav_raw, nav_unused;                                                                                  \
        double xc, Sc = 0.0, S_unused, yk = 0.0;                                                                       \
        uint32_t ph = 0, kstep = 0;                                                                                    \
        if (!glob) {                                                                                                   \
            xc = chain_state(W, sb_c, sc_c, slot_n0a, slot_n0b, n0, &nav_raw, &Sc);                                    \
        } else {                                                                                                       \
            const uint64_t ta_ = row0_addr + (uint64_t)(trA & TR_ROW) * sizeof(SynRow);                                \
            xc = row_state_global((const SynRow *)(uintptr_t)readlane_u64(ta_, 2 * a), n0, &nav_raw);                  \
        }                                                                                                              \
        const int dbx = (nav_raw >> 31) ? 0xfffe : 0; /* dataBit -1: flips the chip's +-1 in 16 bits */               \
        const uint32_t nav = nav_raw & 0x3fffffffu;                                                                    \
        if (CARR == 2) {                                                                                               \
            kstep = (uint32_t)p.kstep[(size_t)b * p.nch + i];                                                          \
            ph = p.kph0[(size_t)b * p.nch + i] + (uint32_t)n0 * kstep;                                                 \
        } else if (!glob) {                                                                                            \
            yk = chain_state(W, sb_k, sc_k, slot_n0a, slot_n0b, n0, &nav_unused, &S_unused);                           \
        } else {                                                                                                       \
            const uint64_t ta_ = row0_addr + (uint64_t)(trA & TR_ROW) * sizeof(SynRow);                                \
            yk = row_state_global((const SynRow *)(uintptr_t)readlane_u64(ta_, 2 * a + 1), n0, &nav_unused);           \
        }                                                                                                              \
        BEFORE_WALK                                                                                                    \
        walk_channel<CODE, CARR>(L, i, xc, Sc, yk, ph, kstep, nav, dbx, acc);                                          \
    }

            /* the channels of one walk variant (bit 2a of MASK = active channel a), rows of the whole tile staged */
#define GPSBB_CHANNEL(CODE, CARR, MASK)                                                                                \
    for (uint64_t m_ = (MASK); m_; m_ &= m_ - 1) {                                                                     \
        const int a = (int)(__builtin_ctzll(m_) >> 1);                                                                 \
        const int i = __builtin_amdgcn_readfirstlane(L.act[a]); /* scalar: the table addresses become SALU work */    \
        const int sc_c = __builtin_amdgcn_readlane(ccnt, 2 * a), sc_k = __builtin_amdgcn_readlane(ccnt, 2 * a + 1);    \
        const int sb_c = __builtin_amdgcn_readlane(cbase, 2 * a), sb_k = __builtin_amdgcn_readlane(cbase, 2 * a + 1);  \
        constexpr bool glob = false;                                                                                   \
        --remaining;                                                                                                   \
        /* before the tile's last walk: start the copy of the next tile's rows */                                     \
        GPSBB_LOOKUP_AND_WALK(CODE, CARR, if (remaining == 0 && more) GPSBB_STAGE_NEXT_TILE)                           \
    }

            /* A tile whose rows do not fit the slice (dense rows: high Doppler at a low sample rate, wraps in
             * nearly every tile): channel by channel, wrap-capable walks throughout; the rows of channel a+1 are
             * copied into the other half of the slice while channel a is walked.  A channel with more than 64
             * rows takes the whole slice and waits; with more than the slice holds, lanes scan the pool in HBM. */
#define GPSBB_DENSE_TILE(CARR)                                                                                         \
    {                                                                                                                  \
        const uint64_t tile_addr = row0_addr + (uint64_t)(trA & TR_ROW) * sizeof(SynRow);                              \
        int nc = __builtin_amdgcn_readlane(ccnt, 0), nk = __builtin_amdgcn_readlane(ccnt, 1);                          \
        bool half_ok = nact > 0 && nc + nk <= 64;                                                                      \
        if (half_ok)                                                                                                   \
            prefetch_channel(W, lane, readlane_u64(tile_addr, 0), readlane_u64(tile_addr, 1), nc, nk, 0);              \
        for (int a = 0; a < nact; a++) {                                                                               \
            const int i = __builtin_amdgcn_readfirstlane(L.act[a]);                                                    \
            const int sc_c = nc, sc_k = nk;                                                                            \
            int sb_c = 0, sb_k = sc_c;                                                                                 \
            bool glob = false;                                                                                         \
            if (half_ok) {                                                                                             \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
                if (a & 1)                                                                                             \
                    slot_n0b = (int)W.a[64 + lane].x;                                                                  \
                else                                                                                                   \
                    slot_n0a = (int)W.a[lane].x;                                                                       \
                sb_c = 64 * (a & 1);                                                                                   \
                sb_k = sb_c + sc_c;                                                                                    \
            } else if (sc_c + sc_k <= WAVE_ROW_CAP) {                                                                  \
                const SlotN0 sn = stage_channel(W, lane, readlane_u64(tile_addr, 2 * a), readlane_u64(tile_addr, 2 * a + 1), \
                                                sc_c, sc_k);                                                           \
                slot_n0a = sn.a;                                                                                       \
                slot_n0b = sn.b;                                                                                       \
            } else {                                                                                                   \
                glob = true;                                                                                           \
            }                                                                                                          \
            const bool last = a + 1 == nact;                                                                           \
            if (!last) {                                                                                               \
                nc = __builtin_amdgcn_readlane(ccnt, 2 * a + 2);                                                       \
                nk = __builtin_amdgcn_readlane(ccnt, 2 * a + 3);                                                       \
            }                                                                                                          \
            /* between this channel's lookups and its walk: the next channel's rows, or the next tile's */           \
            GPSBB_LOOKUP_AND_WALK(                                                                                     \
                1, CARR, if (!last) {                                                                                  \
                    half_ok = nc + nk <= 64;                                                                           \
                    if (half_ok)                                                                                       \
                        prefetch_channel(W, lane, readlane_u64(tile_addr, 2 * a + 2), readlane_u64(tile_addr, 2 * a + 3), \
                                         nc, nk, 64 * ((a + 1) & 1));                                                  \
                } else if (more) GPSBB_STAGE_NEXT_TILE)                                                                \
        }                                                                                                              \
        remaining = 0;                                                                                                 \
    }

            if (ovf) {
                if (fixed_carr)
                    GPSBB_DENSE_TILE(2)
                else
                    GPSBB_DENSE_TILE(1)
            } else if (fixed_carr) {
                GPSBB_CHANNEL(2, 2, m_c2)
                GPSBB_CHANNEL(0, 2, m_c0)
                GPSBB_CHANNEL(1, 2, m_c1)
            } else {
                GPSBB_CHANNEL(2, 0, m_c2 & ~m_k1)
                GPSBB_CHANNEL(0, 0, m_c0 & ~m_k1)
                GPSBB_CHANNEL(2, 1, m_c2 & m_k1)
                GPSBB_CHANNEL(0, 1, m_c0 & m_k1)
                GPSBB_CHANNEL(1, 0, m_c1 & ~m_k1)
                GPSBB_CHANNEL(1, 1, m_c1 & m_k1)
            }
            if (more && nact == 0) /* no active channel: nothing above ran */
                GPSBB_STAGE_NEXT_TILE
#undef GPSBB_CHANNEL
#undef GPSBB_DENSE_TILE
#undef GPSBB_LOOKUP_AND_WALK
#undef GPSBB_STAGE_NEXT_TILE
            if (more) {
                /* the next tile's rows have had the last walk to arrive; waiting here, before this tile's
                 * stores are issued, keeps the wait from covering the stores as well */
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                slot_n0a = (int)W.a[lane].x;
                slot_n0b = (int)W.a[64 + lane].x;
                cur = nxt;
                cwrap = nwrap;
            }

            /* ---- store: int16 I,Q interleaved (c:2754-2755) ---- */
            uint32_t *out = reinterpret_cast<uint32_t *>(iq) + (size_t)b * p.nsamp + n0;
            if (nvalid == SPT && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
                uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
                for (int j = 0; j < SPT; j += 4)
                    o4[j >> 2] = make_uint4(v2s_u32(acc[j]), v2s_u32(acc[j + 1]), v2s_u32(acc[j + 2]), v2s_u32(acc[j + 3]));
            } else {
#pragma unroll
                for (int j = 0; j < SPT; j++)
                    if (j < nvalid)
                        out[j] = v2s_u32(acc[j]);
            }
        }
    } /* chunk loop */
}

/* pure write stream of the same shape as k_synth's output: the empirical int16x2 write ceiling */
/* Counter calibration for READS (experiments build, tools/kbench.py --read-cal): a known number of bytes fetched with the
 * synthesis kernels' own access pattern for the tile states — the first `rows` lanes of a wavefront each read one double of a row of
 * their own, the same column, column after column (k_synth_ev: tx[chain * ntiles + tile]) — so that FETCH_SIZE can be priced on
 * this pattern instead of with the factor the guide gives for wide coalesced reads.  Reads rows * cols * 8 bytes per wavefront. */
__global__ __launch_bounds__(256) void k_read_pattern(const double *__restrict__ src, int rows, int cols, double *__restrict__ sink)
{
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
    const double *__restrict__ p = src + ((size_t)wave * rows + (size_t)(lane < rows ? lane : 0)) * (size_t)cols;
    double acc = 0.0;
    for (int t = 0; t < cols; t++)
        acc += lane < rows ? p[t] : 0.0;
    if (acc == 0x1.23456789abcdep+700)
        sink[0] = acc; /* (never: keeps the loads) */
}

__global__ __launch_bounds__(256) void k_fill_ceiling(uint4 *__restrict__ dst, size_t n16, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint32_t v = seed + (uint32_t)i;
        dst[i] = make_uint4(v, v ^ 0x00010001u, v + 0x00020002u, v ^ 0x7fff7fffu);
    }
}

/* A stream slot's end-of-block states and the handle's self-check word, written straight into the slot's pinned
 * host memory (device-visible): a few hundred KB per push.  A hipMemcpyAsync of that size stalls the calling host
 * thread for milliseconds whenever the copy stream still waits for its event; a kernel launch never does. */
__global__ __launch_bounds__(256) void k_end_states_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst_host, size_t n16,
                                                            const uint32_t *__restrict__ status, uint32_t *__restrict__ status_host)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        dst_host[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *status_host = *status;
}

/* The gather of a stream slot's IQ into its pinned host buffer as a kernel (stores that leave over PCIe): unlike
 * hipMemcpyAsync — which makes the calling thread wait until the copy stream's pending event wait is satisfied, i.e.
 * for the whole pre-pass and synthesis of the push — a launch returns at once, so the host keeps the ring full.
 * A workgroup moves 16 KB at a time, contiguous, so that the stores of a wavefront form full PCIe write bursts. */
typedef unsigned int gather_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gather_to_host(const gather_u32x4 *__restrict__ src, gather_u32x4 *__restrict__ dst_host, size_t n16)
{
    const size_t per = 256 * 4;
    const size_t nchunk = (n16 + per - 1) / per;
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const size_t base = c * per + threadIdx.x;
        gather_u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (base + (size_t)u * 256 < n16)
                v[u] = __builtin_nontemporal_load(src + base + (size_t)u * 256);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (base + (size_t)u * 256 < n16)
                __builtin_nontemporal_store(v[u], dst_host + base + (size_t)u * 256);
    }
}

/* One 64-bit digest per block of int16 I/Q pairs in device memory (gpsbb_device_digest): the sum over the block's samples j of
 * pair_j * m_j modulo 2^64, m_j = (j * DIGEST_STEP + DIGEST_ODD) mod 2^32 — an odd weight per position, so one changed sample
 * changes the sum and two samples swapped do; order-independent as a sum, so that any partition of the block over lanes gives
 * the same number.  Grid (chunks, blocks): every workgroup digests a contiguous piece of one block — 16-byte loads between the
 * ragged ends (a block starts on a 4-byte boundary only: nsamp is any number), one 32 x 32 + 64 multiply-add and one add per
 * sample — and adds its part with one atomic.  (Rounds 4-5 mixed every sample through a 64-bit multiply-xorshift round: 15 vector
 * instructions per sample, and the node driver's digest sink was bound by them — 2.2e11 samples/s beside a synthesis that renders
 * 5.6e11: DESIGN.md 4.  This one is 3.) */
constexpr uint32_t DIGEST_STEP = 0x9E3779BAu, DIGEST_ODD = 0x85EBCA6Bu;
__device__ __forceinline__ uint32_t digest_weight(uint32_t j) { return j * DIGEST_STEP + DIGEST_ODD; }
__global__ __launch_bounds__(256) void k_block_digest(const uint32_t *__restrict__ iq, int nsamp, unsigned long long *__restrict__ out)
{
    const int b = blockIdx.y;
    const uint32_t *__restrict__ p = iq + (size_t)b * (size_t)nsamp;
    const int per = (((nsamp + (int)gridDim.x - 1) / (int)gridDim.x) + 3) & ~3;
    const int j0 = (int)blockIdx.x * per;
    if (j0 >= nsamp)
        return;
    const int j1 = j0 + per < nsamp ? j0 + per : nsamp;
    unsigned long long acc = 0ull;
    /* up to three samples before the first 16-byte boundary, whole vectors, up to three samples after the last one */
    const int mis = (int)((reinterpret_cast<uintptr_t>(p + j0) >> 2) & 3u);
    const int ja = j0 + ((4 - mis) & 3) < j1 ? j0 + ((4 - mis) & 3) : j1;
    if ((int)threadIdx.x < ja - j0) {
        const int j = j0 + (int)threadIdx.x;
        acc += (unsigned long long)p[j] * digest_weight((uint32_t)j);
    }
    const int nvec = (j1 - ja) >> 2;
    const gather_u32x4 *__restrict__ pv = reinterpret_cast<const gather_u32x4 *>(p + ja);
    uint32_t m = digest_weight((uint32_t)(ja + 4 * (int)threadIdx.x));
#pragma unroll 2
    for (int v = (int)threadIdx.x; v < nvec; v += 256) {
        const gather_u32x4 q = pv[v];
        acc += (unsigned long long)q.x * m;
        acc += (unsigned long long)q.y * (m + DIGEST_STEP);
        acc += (unsigned long long)q.z * (m + 2u * DIGEST_STEP);
        acc += (unsigned long long)q.w * (m + 3u * DIGEST_STEP);
        m += 1024u * DIGEST_STEP;
    }
    const int jt = ja + 4 * nvec;
    if ((int)threadIdx.x < j1 - jt) {
        const int j = jt + (int)threadIdx.x;
        acc += (unsigned long long)p[j] * digest_weight((uint32_t)j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc += (unsigned long long)__shfl_down((long long)acc, off);
    if ((threadIdx.x & 63) == 0 && acc)
        atomicAdd(out + b, acc);
}

} /* namespace gpsbb_impl */
#endif
