/*
 * gpsbb_node.h — one process, N GPUs, ONE output stream: the node-level time-shard driver of libgpsbb.
 *
 * The reference has exactly one consumer of one IQ stream: pluto_tx_thread_ep() memcpy()s every block the
 * generator loop produced and hands it to iio_buffer_push (plutogpssim.c:2146-2158), with both threads pinned to
 * cores of their own (plutogpssim.c:2045-2056, 2069, 2289).  This driver keeps that shape at node scale
 * (BASELINE.json configs[4]: a 16-channel 25 MS/s stream time-sharded across the 8 GPUs of a node, host gather):
 *
 *   - the block sequence [0, nblocks) is cut into `nshards` CONTIGUOUS time shards, shard g = blocks
 *     [first_g, first_(g+1)), boundaries on multiples of blocks_per_slot;
 *   - one producer thread per shard owns one gpsbb_t handle on its GPU and one gpsbb_stream ring (pinned host
 *     slots, or device slots with GPSBB_NODE_DEVICE_ONLY).  Before it creates either, the thread binds itself to
 *     the CPUs that are local to its GPU (sysfs local_cpulist of the GPU's PCI function), so the ring's pinned
 *     pages, the descriptor staging and the thread that touches them sit on the GPU's NUMA node;
 *   - a shard that does not start at block 0 gets the exact carrier phase of its first block from the device-side
 *     carrier chain over everything before it (gpsbb_chain_carrier on its own GPU: plutogpssim.c:2741-2746 never
 *     re-seeds carr_phase, so it is a function of all earlier blocks); the 32-bit accumulator of
 *     GPSBB_NODE_FIXED_CARRIER is carried forward in integer arithmetic.  No GPU talks to another one;
 *   - ONE sink receives the blocks.  By default strictly in stream order — block 0 first, every block once, the
 *     contract of the reference's single consumer — whichever GPU rendered them; a shard whose turn has not come
 *     fills its ring and waits.  A sink that can place blocks itself (a file written with pwrite, a host buffer)
 *     asks for GPSBB_NODE_INDEXED and gets every slot as soon as it is complete, with its block index: that is
 *     what lets N GPUs deliver N times the blocks per second into one output from contiguous shards.  A consumer that
 *     needs the order AND the rate (a radio fed faster than real time, a pipe) asks for GPSBB_NODE_INTERLEAVED: the
 *     slots go round the GPUs, so the next ones are always being rendered while the current one is consumed.
 *
 * Plain C ABI like gpsbb.h; lives in libgpsbb.so.  A node object is driven by one thread at a time.
 */
#ifndef GPSBB_NODE_H
#define GPSBB_NODE_H

#include "gpsbb.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GPSBB_NODE_MAX_SHARDS 64

#define GPSBB_NODE_INDEXED 1u       /* the sink takes slots in completion order (in order within a shard), told by block index */
#define GPSBB_NODE_CONCURRENT 2u    /* with INDEXED: the sink may be entered by several producer threads at once (it is
                                       thread-safe and the ranges are disjoint); otherwise calls are serialised */
#define GPSBB_NODE_DEVICE_ONLY 4u   /* rings of GPSBB_STREAM_DEVICE_ONLY slots: `iq` handed to the sink is DEVICE memory of
                                       the shard's GPU, for a consumer there */
#define GPSBB_NODE_NO_AFFINITY 8u   /* do not bind the producer threads (a host that manages placement itself) */
#define GPSBB_NODE_FIXED_CARRIER 16u /* GPSBB_FIXED_CARRIER streams (plutogpssim.h:160-161, c:2675 / 2699 / 2748) */
#define GPSBB_NODE_INTERLEAVED 32u  /* slot k of the stream (blocks [k * blocks_per_slot, (k + 1) * blocks_per_slot)) goes to shard
                                       k mod nshards instead of contiguous shards: every slot starts a chain of its own from the
                                       exact phase of its first block (each producer chains the whole stream once on its GPU,
                                       0.1 s per 100 000 blocks, and pushes with GPSBB_PUSH_NEW_CHAIN).  With this layout the
                                       ORDERED sink scales too: while the consumer takes slot k from one GPU the others are
                                       rendering k + 1 ... k + nshards * depth - 1 */
#define GPSBB_NODE_DIGESTS 64u      /* every push is rendered WITH the digests of its blocks (GPSBB_PUSH_DIGEST: include/gpsbb.h): a sink
                                       that wants them calls gpsbb_node_slot_digests from inside its callback instead of reading the
                                       slot back (gpsbb_slot_digest).  Costs the synthesis ~7 %, where reading back costs it a third */

/*
 * The one consumer.  `iq` = nblocks consecutive blocks (nblocks * nsamp int16 I/Q pairs, interleaved), the first of them
 * block `first_block` of the stream; valid only during the call (the slot is rendered into again afterwards).  `shard` says
 * which producer / GPU made them.  Return < 0 to stop the run (the reference leaves its loop on a negative
 * iio_buffer_push, plutogpssim.c:2153-2157): gpsbb_node_run then returns GPSBB_E_STATE after the producers have wound down.
 */
typedef int (*gpsbb_node_sink_fn)(void *user, const int16_t *iq, long first_block, int nblocks, int shard);

typedef struct gpsbb_node_config {
    int nshards;          /* producer threads = handles = time shards, 1 .. GPSBB_NODE_MAX_SHARDS */
    const int *devices;   /* nshards HIP device ordinals (they may repeat: several shards on one GPU); NULL = 0, 1, 2 ... */
    int nch;              /* channels per block */
    double delt;          /* 1 / fs (plutogpssim.c:2397) */
    int nsamp;            /* samples per block */
    int blocks_per_slot;  /* blocks per push of a producer's ring */
    int depth;            /* ring slots per producer (>= 2) */
    unsigned flags;       /* GPSBB_NODE_* */
} gpsbb_node_config_t;

typedef struct gpsbb_node_shard_stats {
    long first_block, nblocks;
    int device;
    int numa_node;        /* of the GPU (-1: unknown) */
    int cpus_bound;       /* CPUs in the producer thread's affinity mask after binding (0: not bound) */
    double seed_seconds;  /* gpsbb_chain_carrier over the blocks before the shard */
    double busy_seconds;  /* first push to last pop */
    double wait_seconds;  /* of those, spent waiting for the sink's turn (ordered mode) or inside the sink */
} gpsbb_node_shard_stats_t;

typedef struct gpsbb_node_stats {
    double seconds;       /* the whole run, wall clock */
    long blocks;          /* blocks delivered to the sink */
    int nshards;
    gpsbb_node_shard_stats_t shard[GPSBB_NODE_MAX_SHARDS];
} gpsbb_node_stats_t;

typedef struct gpsbb_node gpsbb_node_t;

/* Starts the producer threads; each binds itself, creates its handle and its ring.  Fails as a whole if any of them fails
 * (GPSBB_E_NODEVICE without a GPU: there is no CPU path). */
int gpsbb_node_create(gpsbb_node_t **out, const gpsbb_node_config_t *cfg);

/*
 * Render blocks [0, nblocks) of the stream described by ch[nblocks * nch] (block-major, as gpsbb_batch_create; consecutive
 * in time: the carrier is chained from block to block as the reference's loop does, a channel whose prn changes restarts
 * from its descriptor's carr_phase) and deliver them to `sink`.  Returns when every block has been delivered or the sink
 * stopped the run.  Can be called again (another stream): the rings are kept — also after a run that FAILED (a descriptor
 * outside the contract: GPSBB_E_BADCHAN; out of memory; a HIP error): every shard drains its ring or, where the library has
 * closed the stream, gets a new handle and ring before the call returns.  Only if that fails too do later runs return the error
 * of the re-creation, and the node is good for gpsbb_node_destroy alone.
 */
int gpsbb_node_run(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks, gpsbb_node_sink_fn sink, void *user,
                   gpsbb_node_stats_t *stats);

/*
 * gpsbb_node_run with the DRIVER'S OWN sink: digests[b] = the 64-bit digest of block b (gpsbb_device_digest's number), taken on
 * the GPU that rendered the block, every shard on its own — what a host that keeps the IQ in HBM compares instead of the bytes,
 * and a consumer that costs the driver next to nothing.  Rings in HBM (GPSBB_NODE_DEVICE_ONLY) are rendered WITH their digests
 * (GPSBB_PUSH_DIGEST: the synthesis kernel adds them up as it renders, gpsbb_stream_pop_digest hands them to the shard's
 * producer thread; nothing is read back): 0.93 x the rate of a sink that does nothing.  Rings in host memory
 * (no GPSBB_NODE_DEVICE_ONLY) are digested by the producer threads on the host, the same number.  The node's flags decide the
 * layout (contiguous / GPSBB_NODE_INTERLEAVED) as for gpsbb_node_run; the order of delivery does not matter to this sink: it is
 * entered as GPSBB_NODE_INDEXED | GPSBB_NODE_CONCURRENT whatever the flags say.
 */
int gpsbb_node_run_digest(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks, uint64_t *digests, gpsbb_node_stats_t *stats);

/* From INSIDE a sink of a GPSBB_NODE_DIGESTS node: the digests of the `nblocks` blocks the sink has just been handed by `shard`
 * (gpsbb_device_digest's numbers, computed as the blocks were rendered).  GPSBB_E_STATE anywhere else. */
int gpsbb_node_slot_digests(gpsbb_node_t *n, int shard, uint64_t *digests, int nblocks);

/*
 * The same stream, INCREMENTALLY — what the reference's loop does: it makes the descriptors of one block, renders it, and goes
 * round again, for as long as it runs (plutogpssim.c:2655-2687, the 30 s maintenance c:2764-2805).
 *   gpsbb_node_begin  starts a run with nothing to render yet;
 *   gpsbb_node_feed   the next `nblocks` blocks of the stream (copied: the caller's array is free on return; any number — the
 *                     driver cuts the stream into slots of blocks_per_slot itself and keeps what does not fill one).  Slot k of the
 *                     stream goes to shard k mod nshards, a chain of its own from the exact carrier phases (the layout of
 *                     GPSBB_NODE_INTERLEAVED, whatever the node's flags say: contiguous shards need the length of a stream that
 *                     has none); the feeder chains the carrier across everything it is fed on a handle of its own
 *                     (gpsbb_chain_carrier: a microsecond per block).  Returns once the blocks are queued: it WAITS while the
 *                     queue of the shard whose turn it is holds depth + 1 slots — the front end runs that far ahead of the rings and
 *                     no further, memory does not grow with the duration.  GPSBB_E_STATE once the sink has stopped the run;
 *   gpsbb_node_end    no more blocks: a short last slot goes out padded, the rings drain, statistics as gpsbb_node_run
 *                     (seed_seconds: the feeder's chain).  Must be called after begin, also after a failed feed.
 * The sink is called as in gpsbb_node_run (ordered by default: block 0 first, every block once).  One feeder thread.
 */
int gpsbb_node_begin(gpsbb_node_t *n, gpsbb_node_sink_fn sink, void *user);
int gpsbb_node_feed(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks);
int gpsbb_node_end(gpsbb_node_t *n, gpsbb_node_stats_t *stats);

void gpsbb_node_destroy(gpsbb_node_t *n);

/* which shard renders block b of an nblocks-long stream, and where the shards begin: first[0 .. nshards] (first[nshards] =
 * nblocks); pure arithmetic, no GPU */
int gpsbb_node_plan(long nblocks, int nshards, int blocks_per_slot, long *first);

#ifdef __cplusplus
}
#endif
#endif
