/*
 * gpsbb_node.cpp — include/gpsbb_node.h: one process, one producer thread + handle + ring per time shard, one sink.
 *
 * Host code only, on top of the public C ABI of gpsbb.h (nothing here knows a kernel): what the reference's main loop and
 * pluto_tx_thread_ep() are to one CPU (plutogpssim.c:2655-2806, 2146-2158, pinned at 2045-2056 / 2069 / 2289), this is
 * to the GPUs of a node.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <pthread.h>
#include <sched.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "gpsbb_node.h"

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* "0-31,128-159" -> cpu_set_t; returns the number of CPUs */
int parse_cpulist(const char *s, cpu_set_t *set)
{
    CPU_ZERO(set);
    int n = 0;
    while (*s) {
        char *end = nullptr;
        const long a = strtol(s, &end, 10);
        if (end == s)
            break;
        long b = a;
        s = end;
        if (*s == '-') {
            b = strtol(s + 1, &end, 10);
            s = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && !CPU_ISSET((int)c, set)) {
                CPU_SET((int)c, set);
                n++;
            }
        if (*s == ',')
            s++;
    }
    return n;
}

struct Shard {
    gpsbb_node *node = nullptr;
    int index = 0, device = 0;
    std::thread th;
    gpsbb_t *h = nullptr;
    gpsbb_stream_t *st = nullptr;
    int numa_node = -1, cpus_bound = 0;
    int create_rc = GPSBB_OK;
    std::vector<gpsbb_chan_t> slot_desc; /* a push's descriptors where they have to be edited: the shard's first slot (seed) and a padded tail */
    std::vector<uint64_t> slot_dig;      /* gpsbb_node_run_digest over rings in HBM: the digests the slot just popped was rendered with */
    bool slot_dig_valid = false;
    /* the job of the current run */
    long first = 0, count = 0;
    int rc = GPSBB_OK;
    gpsbb_node_shard_stats_t stats{};
    /* an incremental run (gpsbb_node_begin / _feed / _end): the slots waiting for this shard, oldest first */
    struct FedSlot {
        long slot;  /* its number in the stream: blocks [slot * bps, slot * bps + nb) */
        int nb;     /* blocks it really holds (the stream's last slot may be short: padded with idle blocks) */
        std::vector<gpsbb_chan_t> desc; /* bps * nch descriptors, the first block's carrier phases exact */
    };
    std::deque<FedSlot> fed;
};

} /* namespace */

struct gpsbb_node {
    gpsbb_node_config_t cfg{};
    std::vector<int> devices;
    std::vector<Shard> shards;
    std::mutex m;
    std::condition_variable cv;
    /* life cycle of the producer threads */
    int created = 0;       /* threads that have finished setting up (successfully or not) */
    unsigned long job = 0; /* run number; a thread works when it sees a new one */
    int done = 0;          /* threads that have finished the current job */
    bool quit = false;
    /* the current run */
    const gpsbb_chan_t *ch = nullptr;
    long nblocks = 0;
    gpsbb_node_sink_fn sink = nullptr;
    void *user = nullptr;
    long next_block = 0;   /* ordered mode: the block the sink gets next */
    long delivered = 0;
    bool stop = false;     /* the sink asked to stop, or a shard failed */
    std::mutex sink_m;     /* indexed, not concurrent: one sink call at a time */
    uint64_t *digest_out = nullptr; /* gpsbb_node_run_digest: the driver's own sink writes here; the sink is entered indexed + concurrent */
    /* an incremental run: the feeder (the caller's thread) cuts what it is fed into slots, chains the carrier across them on a
     * handle of its own and deals the slots round the shards' queues; bounded: a feed waits while its shard's queue is full */
    bool feeding = false, fed_eof = false;
    gpsbb_t *feed_h = nullptr;           /* the feeder's handle (the chain of the fed blocks: gpsbb_chain_carrier) */
    std::vector<gpsbb_chan_t> pend;      /* fed blocks that do not fill a slot yet */
    std::vector<gpsbb_chan_t> work;      /* pend + the blocks of the current feed */
    std::vector<double> work_seed;
    long fed_slots = 0, fed_blocks = 0;
    bool have_carry = false;
    int carry_prn[GPSBB_MAX_CHAN] = {};
    double carry_phase[GPSBB_MAX_CHAN] = {}; /* the exact phase (the accumulator, fixed-point carrier) at pend's first block */
    double feed_t0 = 0.0, feed_chain_s = 0.0;
};

extern "C" int gpsbb_node_plan(long nblocks, int nshards, int blocks_per_slot, long *first)
{
    if (nblocks < 0 || nshards < 1 || nshards > GPSBB_NODE_MAX_SHARDS || blocks_per_slot < 1 || !first)
        return GPSBB_E_BADARG;
    /* shard g = slots [g * S / N, (g + 1) * S / N) of the S = ceil(B / bps) slots of the stream: contiguous in time
     * (BASELINE.json configs[4]: GPU g gets blocks [g * B / G, (g + 1) * B / G)), boundaries on whole pushes */
    const long slots = (nblocks + blocks_per_slot - 1) / blocks_per_slot;
    for (int g = 0; g <= nshards; g++) {
        long b = (slots * g / nshards) * blocks_per_slot;
        first[g] = b > nblocks ? nblocks : b;
    }
    first[nshards] = nblocks;
    return GPSBB_OK;
}

namespace {

/* the 32-bit accumulator of the fixed-point carrier at the start of block `first`, per channel (c:2675, 2748): the same
 * integer recurrence gpsbb_stream_* carries from push to push */
void fixed_carrier_seed(const gpsbb_chan_t *ch, long first, int nch, double delt, int nsamp, double *seed)
{
    for (int i = 0; i < nch; i++) {
        int prev_prn = 0;
        uint32_t ph = 0;
        for (long b = 0; b < first; b++) {
            const gpsbb_chan_t &c = ch[(size_t)b * nch + i];
            if (c.prn > 0) {
                if (c.prn != prev_prn)
                    ph = (uint32_t)c.carr_phase;
                const volatile double scaled = 512.0 * 65536.0 * c.f_carr * delt;
                const int32_t step = (int32_t)std::round(scaled);
                ph += (uint32_t)nsamp * (uint32_t)step;
            }
            prev_prn = c.prn > 0 ? c.prn : 0;
        }
        const gpsbb_chan_t &c = ch[(size_t)first * nch + i];
        seed[i] = (c.prn > 0 && c.prn == prev_prn) ? (double)ph : c.carr_phase;
    }
}

/* hand one popped slot to the sink; returns false when the run is to stop */
bool deliver(gpsbb_node *n, Shard &s, const int16_t *iq, long first_block, int nb)
{
    const bool indexed = (n->cfg.flags & GPSBB_NODE_INDEXED) != 0 || n->digest_out != nullptr;
    const bool concurrent = (indexed && (n->cfg.flags & GPSBB_NODE_CONCURRENT)) || n->digest_out != nullptr;
    const double t0 = now_s();
    int rc = 0;
    if (!indexed) {
        /* one ordered stream: wait until every earlier block has been delivered (plutogpssim.c:2146-2158: one consumer) */
        std::unique_lock<std::mutex> lk(n->m);
        n->cv.wait(lk, [&] { return n->stop || n->next_block == first_block; });
        if (n->stop)
            return false;
        lk.unlock();
        rc = n->sink(n->user, iq, first_block, nb, s.index); /* nobody else can be here: it is this block's turn only */
        lk.lock();
        n->next_block = first_block + nb;
        n->delivered += nb;
        if (rc < 0)
            n->stop = true;
        n->cv.notify_all();
    } else {
        {
            std::lock_guard<std::mutex> lk(n->m);
            if (n->stop)
                return false;
        }
        if (concurrent) {
            rc = n->sink(n->user, iq, first_block, nb, s.index);
        } else {
            std::lock_guard<std::mutex> g(n->sink_m);
            rc = n->sink(n->user, iq, first_block, nb, s.index);
        }
        std::lock_guard<std::mutex> lk(n->m);
        n->delivered += nb;
        if (rc < 0) {
            n->stop = true;
            n->cv.notify_all();
        }
    }
    s.stats.wait_seconds += now_s() - t0;
    s.slot_dig_valid = false; /* (gpsbb_node_slot_digests: from inside the sink only) */
    return rc >= 0;
}

/* the driver's own digest sink over rings in HBM: the pushes are rendered WITH their digests (GPSBB_PUSH_DIGEST: the synthesis
 * kernel adds them up as it renders) and pop hands them out — nothing reads the slot back */
inline bool digest_at_render(const gpsbb_node *n)
{
    return (n->digest_out != nullptr && (n->cfg.flags & GPSBB_NODE_DEVICE_ONLY) != 0) || (n->cfg.flags & GPSBB_NODE_DIGESTS) != 0;
}
inline int shard_push(gpsbb_node *n, Shard &s, const gpsbb_chan_t *desc, unsigned flags)
{
    if (digest_at_render(n))
        flags |= GPSBB_PUSH_DIGEST;
    return flags ? gpsbb_stream_push_ex(s.st, desc, flags) : gpsbb_stream_push(s.st, desc);
}
inline int shard_pop(gpsbb_node *n, Shard &s, const int16_t **iq)
{
    s.slot_dig_valid = false;
    if (!digest_at_render(n))
        return gpsbb_stream_pop(s.st, iq, nullptr);
    s.slot_dig.resize((size_t)n->cfg.blocks_per_slot);
    const int rc = gpsbb_stream_pop_digest(s.st, iq, nullptr, s.slot_dig.data());
    s.slot_dig_valid = rc == GPSBB_OK;
    return rc;
}

/* GPSBB_NODE_INTERLEAVED: shard g renders slots g, g + N, g + 2N ... of the stream, each a chain of its own */
int run_shard_interleaved(gpsbb_node *n, Shard &s)
{
    const gpsbb_node_config_t &c = n->cfg;
    const int bps = c.blocks_per_slot, nch = c.nch, N = c.nshards;
    const bool fixed = (c.flags & GPSBB_NODE_FIXED_CARRIER) != 0;
    const long B = n->nblocks, nslots_all = (B + bps - 1) / bps;
    const long mine = s.index < nslots_all ? (nslots_all - s.index + N - 1) / N : 0; /* slots g, g + N, ... */
    s.stats.first_block = (long)s.index * bps;
    s.stats.nblocks = 0;
    s.stats.seed_seconds = s.stats.busy_seconds = s.stats.wait_seconds = 0.0;
    if (mine <= 0)
        return GPSBB_OK;
    int rc = gpsbb_stream_reset(s.st);
    if (rc != GPSBB_OK)
        return rc;
    /* the exact start phase of every block of the stream, on this shard's own GPU (every shard does: no GPU talks to
     * another one); what it needs of it are the first blocks of its slots */
    const double t0 = now_s();
    std::vector<double> seeds((size_t)B * nch);
    if (fixed) {
        for (int i = 0; i < nch; i++) {
            int prev_prn = 0;
            uint32_t ph = 0;
            for (long b = 0; b < B; b++) {
                const gpsbb_chan_t &d = n->ch[(size_t)b * nch + i];
                if (d.prn > 0) {
                    if (d.prn != prev_prn)
                        ph = (uint32_t)d.carr_phase;
                    seeds[(size_t)b * nch + i] = (double)ph;
                    const volatile double scaled = 512.0 * 65536.0 * d.f_carr * c.delt;
                    ph += (uint32_t)c.nsamp * (uint32_t)(int32_t)std::round(scaled);
                } else {
                    seeds[(size_t)b * nch + i] = 0.0;
                }
                prev_prn = d.prn > 0 ? d.prn : 0;
            }
        }
    } else {
        double end[GPSBB_MAX_CHAN];
        long done = 0;
        std::vector<gpsbb_chan_t> patched;
        while (done < B) {
            const long piece = B - done > 32768 ? 32768 : B - done;
            const gpsbb_chan_t *src = n->ch + (size_t)done * nch;
            if (done > 0) { /* carry the end phases of the piece before into this piece's first block */
                patched.assign(src, src + (size_t)piece * nch);
                for (int i = 0; i < nch; i++) {
                    const gpsbb_chan_t &prev = n->ch[(size_t)(done - 1) * nch + i];
                    if (patched[i].prn > 0 && patched[i].prn == prev.prn)
                        patched[i].carr_phase = end[i];
                }
                src = patched.data();
            }
            rc = gpsbb_chain_carrier(s.h, src, (int)piece, nch, c.delt, c.nsamp, seeds.data() + (size_t)done * nch, end);
            if (rc != GPSBB_OK)
                return rc;
            done += piece;
        }
    }
    s.stats.seed_seconds = now_s() - t0;
    const unsigned depth = (unsigned)c.depth;
    const double t_busy = now_s();
    long pushed = 0, popped = 0;
    bool going = true;
    while (going && popped < mine) {
        while (going && pushed < mine && (unsigned)gpsbb_stream_pending(s.st) < depth) {
            const long b0 = ((long)s.index + pushed * N) * bps;
            const long nb = B - b0 < bps ? B - b0 : bps;
            s.slot_desc.assign((size_t)bps * nch, gpsbb_chan_t{}); /* (a short last slot is padded with idle blocks) */
            memcpy(s.slot_desc.data(), n->ch + (size_t)b0 * nch, (size_t)nb * nch * sizeof(gpsbb_chan_t));
            for (int i = 0; i < nch; i++)
                if (s.slot_desc[i].prn > 0)
                    s.slot_desc[i].carr_phase = seeds[(size_t)b0 * nch + i];
            rc = shard_push(n, s, s.slot_desc.data(), GPSBB_PUSH_NEW_CHAIN);
            if (rc != GPSBB_OK)
                return rc;
            pushed++;
        }
        const int16_t *iq = nullptr;
        rc = shard_pop(n, s, &iq);
        if (rc != GPSBB_OK)
            return rc;
        const long b0 = ((long)s.index + popped * N) * bps;
        const long nb = B - b0 < bps ? B - b0 : bps;
        popped++;
        s.stats.nblocks += nb;
        going = deliver(n, s, iq, b0, (int)nb);
    }
    while (gpsbb_stream_pending(s.st) > 0) {
        const int16_t *iq = nullptr;
        const int r2 = gpsbb_stream_pop(s.st, &iq, nullptr);
        if (r2 != GPSBB_OK)
            return r2;
    }
    s.stats.busy_seconds = now_s() - t_busy;
    return GPSBB_OK;
}

/* An incremental run: this shard renders the slots the feeder queues for it (slot k of the stream goes to shard k mod N, as
 * GPSBB_NODE_INTERLEAVED: every slot a chain of its own from the exact phases the feeder put into its first block) until the
 * feeder says the stream is over and the queue is empty. */
int run_shard_feed(gpsbb_node *n, Shard &s)
{
    const gpsbb_node_config_t &c = n->cfg;
    const int bps = c.blocks_per_slot;
    s.stats.first_block = (long)s.index * bps;
    s.stats.nblocks = 0;
    s.stats.seed_seconds = s.stats.busy_seconds = s.stats.wait_seconds = 0.0;
    int rc = gpsbb_stream_reset(s.st);
    if (rc != GPSBB_OK)
        return rc;
    const unsigned depth = (unsigned)c.depth;
    const double t_busy = now_s();
    std::deque<std::pair<long, int>> flying; /* (first block, blocks) of the slots in the ring, oldest first */
    bool going = true, eof = false;
    while (going && !(eof && flying.empty())) {
        /* push what is waiting, as far as the ring takes it; wait for the feeder only when there is nothing to pop */
        while (going && !eof && (unsigned)flying.size() < depth) {
            Shard::FedSlot slot;
            {
                std::unique_lock<std::mutex> lk(n->m);
                if (flying.empty())
                    n->cv.wait(lk, [&] { return n->stop || !s.fed.empty() || n->fed_eof; });
                if (n->stop) {
                    going = false;
                    break;
                }
                if (s.fed.empty()) {
                    eof = n->fed_eof;
                    break;
                }
                slot = std::move(s.fed.front());
                s.fed.pop_front();
                n->cv.notify_all(); /* room for the feeder */
            }
            /* (one shard: consecutive slots of one stream, the ring's own chain carries the phase across them) */
            rc = shard_push(n, s, slot.desc.data(), c.nshards == 1 ? 0u : GPSBB_PUSH_NEW_CHAIN);
            if (rc != GPSBB_OK)
                return rc;
            flying.emplace_back(slot.slot * bps, slot.nb);
        }
        if (!going || flying.empty())
            continue;
        const int16_t *iq = nullptr;
        rc = shard_pop(n, s, &iq);
        if (rc != GPSBB_OK)
            return rc;
        const std::pair<long, int> f = flying.front();
        flying.pop_front();
        s.stats.nblocks += f.second;
        going = deliver(n, s, iq, f.first, f.second);
    }
    while (gpsbb_stream_pending(s.st) > 0) {
        const int16_t *iq = nullptr;
        const int r2 = gpsbb_stream_pop(s.st, &iq, nullptr);
        if (r2 != GPSBB_OK)
            return r2;
    }
    s.stats.busy_seconds = now_s() - t_busy;
    return GPSBB_OK;
}

int run_shard(gpsbb_node *n, Shard &s)
{
    if (n->feeding)
        return run_shard_feed(n, s);
    if (n->cfg.flags & GPSBB_NODE_INTERLEAVED)
        return run_shard_interleaved(n, s);
    const gpsbb_node_config_t &c = n->cfg;
    const int bps = c.blocks_per_slot, nch = c.nch;
    const bool fixed = (c.flags & GPSBB_NODE_FIXED_CARRIER) != 0;
    s.stats.first_block = s.first;
    s.stats.nblocks = s.count;
    s.stats.seed_seconds = s.stats.busy_seconds = s.stats.wait_seconds = 0.0;
    if (s.count <= 0)
        return GPSBB_OK;
    int rc = gpsbb_stream_reset(s.st);
    if (rc != GPSBB_OK)
        return rc;
    /* the exact carrier phase this shard starts from: the chain over everything before it, on this shard's own GPU */
    double seed[GPSBB_MAX_CHAN];
    bool seeded = false;
    if (s.first > 0) {
        const double t0 = now_s();
        if (fixed) {
            fixed_carrier_seed(n->ch, s.first, nch, c.delt, c.nsamp, seed);
        } else {
            /* gpsbb_chain_carrier takes int block counts: chain in pieces, carrying the end phases across */
            double end[GPSBB_MAX_CHAN];
            long done = 0;
            bool have_end = false;
            while (done < s.first) {
                const long piece = s.first - done > 32768 ? 32768 : s.first - done;
                const gpsbb_chan_t *src = n->ch + (size_t)done * nch;
                std::vector<gpsbb_chan_t> patched;
                if (have_end) {
                    patched.assign(src, src + (size_t)piece * nch);
                    for (int i = 0; i < nch; i++) {
                        const gpsbb_chan_t &prev = n->ch[(size_t)(done - 1) * nch + i];
                        if (patched[i].prn > 0 && patched[i].prn == prev.prn)
                            patched[i].carr_phase = end[i];
                    }
                    src = patched.data();
                }
                rc = gpsbb_chain_carrier(s.h, src, (int)piece, nch, c.delt, c.nsamp, nullptr, end);
                if (rc != GPSBB_OK)
                    return rc;
                have_end = true;
                done += piece;
            }
            for (int i = 0; i < nch; i++) {
                const gpsbb_chan_t &cur = n->ch[(size_t)s.first * nch + i], &prev = n->ch[(size_t)(s.first - 1) * nch + i];
                /* a channel that changes satellite at the boundary keeps its own phase (allocateChannel, c:1956-1964) */
                seed[i] = (cur.prn > 0 && cur.prn == prev.prn) ? end[i] : cur.carr_phase;
            }
        }
        seeded = true;
        s.stats.seed_seconds = now_s() - t0;
    }
    const unsigned depth = (unsigned)c.depth;
    const long nslots = (s.count + bps - 1) / bps;
    const double t_busy = now_s();
    long pushed = 0, popped = 0;
    bool going = true;
    while (going && popped < nslots) {
        while (going && pushed < nslots && (unsigned)gpsbb_stream_pending(s.st) < depth) {
            const long b0 = s.first + pushed * bps;
            const long nb = s.first + s.count - b0 < bps ? s.first + s.count - b0 : bps;
            const gpsbb_chan_t *src = n->ch + (size_t)b0 * nch;
            if ((pushed == 0 && seeded) || nb < bps) {
                s.slot_desc.assign((size_t)bps * nch, gpsbb_chan_t{}); /* a short last slot is padded with idle blocks (prn 0) */
                memcpy(s.slot_desc.data(), src, (size_t)nb * nch * sizeof(gpsbb_chan_t));
                if (pushed == 0 && seeded)
                    for (int i = 0; i < nch; i++)
                        s.slot_desc[i].carr_phase = seed[i];
                src = s.slot_desc.data();
            }
            rc = shard_push(n, s, src, 0u);
            if (rc != GPSBB_OK)
                return rc;
            pushed++;
        }
        const int16_t *iq = nullptr;
        rc = shard_pop(n, s, &iq);
        if (rc != GPSBB_OK)
            return rc;
        const long b0 = s.first + popped * bps;
        const long nb = s.first + s.count - b0 < bps ? s.first + s.count - b0 : bps;
        popped++;
        going = deliver(n, s, iq, b0, (int)nb);
    }
    /* a stopped run leaves slots in the ring: drain them so that the next run can reset the stream */
    while (gpsbb_stream_pending(s.st) > 0) {
        const int16_t *iq = nullptr;
        const int r2 = gpsbb_stream_pop(s.st, &iq, nullptr);
        if (r2 != GPSBB_OK)
            return r2;
    }
    s.stats.busy_seconds = now_s() - t_busy;
    return GPSBB_OK;
}

void shard_main(gpsbb_node *n, Shard *sp)
{
    Shard &s = *sp;
    const gpsbb_node_config_t &c = n->cfg;
    /* placement first: the thread, and with it everything it allocates and touches from here on — the handle's staging
     * arenas, the ring's pinned slots — goes next to the GPU */
    if (!(c.flags & GPSBB_NODE_NO_AFFINITY)) {
        char cpus[512];
        int node = -1;
        if (gpsbb_device_affinity(s.device, &node, cpus, sizeof cpus) == GPSBB_OK) {
            s.numa_node = node;
            cpu_set_t set;
            if (cpus[0] && parse_cpulist(cpus, &set) > 0 && pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0) {
                cpu_set_t got;
                if (pthread_getaffinity_np(pthread_self(), sizeof got, &got) == 0)
                    s.cpus_bound = CPU_COUNT(&got);
            }
        }
    } else {
        (void)gpsbb_device_affinity(s.device, &s.numa_node, nullptr, 0);
    }
    int rc = gpsbb_create(&s.h, s.device);
    if (rc == GPSBB_OK) {
        const unsigned sf = GPSBB_CHAIN_CARRIER | ((c.flags & GPSBB_NODE_FIXED_CARRIER) ? GPSBB_FIXED_CARRIER : 0u) |
                            ((c.flags & GPSBB_NODE_DEVICE_ONLY) ? GPSBB_STREAM_DEVICE_ONLY : 0u);
        rc = gpsbb_stream_create(s.h, c.nch, c.delt, c.nsamp, c.blocks_per_slot, c.depth, sf, &s.st);
    }
    s.create_rc = rc;
    unsigned long seen = 0;
    {
        std::unique_lock<std::mutex> lk(n->m);
        n->created++;
        n->cv.notify_all();
    }
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(n->m);
            n->cv.wait(lk, [&] { return n->quit || n->job != seen; });
            if (n->quit)
                break;
            seen = n->job;
        }
        if (s.create_rc == GPSBB_OK) {
            /* nothing may leave this thread but a return code: an exception through a std::thread is std::terminate, and this is
             * a C library (the vectors of run_shard can throw bad_alloc on streams of millions of blocks) */
            try {
                s.rc = run_shard(n, s);
            } catch (const std::bad_alloc &) {
                s.rc = GPSBB_E_NOMEM;
            } catch (...) {
                s.rc = GPSBB_E_INTERNAL;
            }
            if (s.rc != GPSBB_OK) {
                /* A run that failed half-way (a descriptor outside the contract in push 17, a HIP error) leaves slots in the ring, or
                 * a stream the library has closed: the node must still be good for another run (gpsbb_node.h).  Drain what pops;
                 * if that fails too, this shard gets a new handle and a new ring. */
                bool fresh = false;
                while (gpsbb_stream_pending(s.st) > 0) {
                    const int16_t *iq = nullptr;
                    if (gpsbb_stream_pop(s.st, &iq, nullptr) != GPSBB_OK) {
                        fresh = true;
                        break;
                    }
                }
                if (!fresh && gpsbb_stream_reset(s.st) != GPSBB_OK)
                    fresh = true;
                if (fresh) {
                    gpsbb_stream_destroy(s.st);
                    s.st = nullptr;
                    gpsbb_destroy(s.h);
                    s.h = nullptr;
                    int rc2 = gpsbb_create(&s.h, s.device);
                    if (rc2 == GPSBB_OK) {
                        const unsigned sf = GPSBB_CHAIN_CARRIER | ((c.flags & GPSBB_NODE_FIXED_CARRIER) ? GPSBB_FIXED_CARRIER : 0u) |
                                            ((c.flags & GPSBB_NODE_DEVICE_ONLY) ? GPSBB_STREAM_DEVICE_ONLY : 0u);
                        rc2 = gpsbb_stream_create(s.h, c.nch, c.delt, c.nsamp, c.blocks_per_slot, c.depth, sf, &s.st);
                    }
                    s.create_rc = rc2; /* (not GPSBB_OK: every later run reports it — the node is then only good for gpsbb_node_destroy) */
                }
            }
        } else {
            s.rc = s.create_rc;
        }
        std::unique_lock<std::mutex> lk(n->m);
        if (s.rc != GPSBB_OK)
            n->stop = true; /* the others wind down: an ordered stream with a hole is no stream */
        n->done++;
        n->cv.notify_all();
    }
    if (s.st)
        gpsbb_stream_destroy(s.st);
    if (s.h)
        gpsbb_destroy(s.h);
}

} /* namespace */

extern "C" int gpsbb_node_create(gpsbb_node_t **out, const gpsbb_node_config_t *cfg)
{
    if (!out || !cfg || cfg->nshards < 1 || cfg->nshards > GPSBB_NODE_MAX_SHARDS || cfg->nch < 1 || cfg->nch > GPSBB_MAX_CHAN ||
        !(cfg->delt > 0.0) || cfg->nsamp < 1 || cfg->blocks_per_slot < 1 || cfg->depth < 2 ||
        (cfg->flags & ~(GPSBB_NODE_INDEXED | GPSBB_NODE_CONCURRENT | GPSBB_NODE_DEVICE_ONLY | GPSBB_NODE_NO_AFFINITY | GPSBB_NODE_FIXED_CARRIER | GPSBB_NODE_INTERLEAVED | GPSBB_NODE_DIGESTS)))
        return GPSBB_E_BADARG;
    *out = nullptr;
    gpsbb_node *n = new (std::nothrow) gpsbb_node;
    if (!n)
        return GPSBB_E_NOMEM;
    n->cfg = *cfg;
    n->devices.resize(cfg->nshards);
    for (int g = 0; g < cfg->nshards; g++)
        n->devices[g] = cfg->devices ? cfg->devices[g] : g;
    n->cfg.devices = n->devices.data();
    n->shards.resize(cfg->nshards);
    for (int g = 0; g < cfg->nshards; g++) {
        Shard &s = n->shards[g];
        s.node = n;
        s.index = g;
        s.device = n->devices[g];
    }
    int started = 0;
    try {
        for (int g = 0; g < cfg->nshards; g++) {
            n->shards[g].th = std::thread(shard_main, n, &n->shards[g]);
            started++;
        }
    } catch (...) {
        /* fall through: wait for the ones that did start, then tear down */
    }
    {
        std::unique_lock<std::mutex> lk(n->m);
        n->cv.wait(lk, [&] { return n->created == started; });
    }
    int rc = started == cfg->nshards ? GPSBB_OK : GPSBB_E_NOMEM;
    for (int g = 0; g < started && rc == GPSBB_OK; g++)
        rc = n->shards[g].create_rc;
    if (rc != GPSBB_OK) {
        gpsbb_node_destroy(n);
        return rc;
    }
    *out = n;
    return GPSBB_OK;
}

extern "C" void gpsbb_node_destroy(gpsbb_node_t *n)
{
    if (!n)
        return;
    {
        std::lock_guard<std::mutex> lk(n->m);
        n->quit = true;
        n->cv.notify_all();
    }
    for (auto &s : n->shards)
        if (s.th.joinable())
            s.th.join();
    if (n->feed_h)
        gpsbb_destroy(n->feed_h);
    delete n;
}

/* ---- the incremental run ------------------------------------------------------------------------------------------
 * The reference generates its descriptors block by block inside a loop that has no end (plutogpssim.c:2655-2687, maintenance
 * c:2764-2805): a driver that wants all of them up front (gpsbb_node_run) holds 296 bytes x channels per 0.1 s of signal — 4 GB
 * per simulated day — and cannot start rendering before the front end has finished.  begin / feed / end take the stream as it
 * comes: memory is what the rings and a bounded queue per shard hold, whatever the duration. */
namespace {

/* collect return codes and statistics of a finished run (both kinds) */
int finish_run(gpsbb_node *n, long expected_blocks, double t0, gpsbb_node_stats_t *stats)
{
    const int N = n->cfg.nshards;
    int rc = GPSBB_OK;
    for (int g = 0; g < N && rc == GPSBB_OK; g++)
        rc = n->shards[g].rc;
    if (rc == GPSBB_OK && n->delivered != expected_blocks)
        rc = GPSBB_E_STATE; /* the sink stopped the run */
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->seconds = now_s() - t0;
        stats->blocks = n->delivered;
        stats->nshards = N;
        for (int g = 0; g < N; g++) {
            stats->shard[g] = n->shards[g].stats;
            stats->shard[g].device = n->shards[g].device;
            stats->shard[g].numa_node = n->shards[g].numa_node;
            stats->shard[g].cpus_bound = n->shards[g].cpus_bound;
        }
    }
    return rc;
}

/* queue the first `nslots` slots of n->work (slot j holds blocks [j * bps, j * bps + nb)), their first blocks seeded */
int feed_slots(gpsbb_node *n, long nslots, long nblocks_in_work)
{
    const gpsbb_node_config_t &c = n->cfg;
    const int bps = c.blocks_per_slot, nch = c.nch, N = c.nshards;
    for (long j = 0; j < nslots; j++) {
        const long b0 = j * bps;
        const long nb = nblocks_in_work - b0 < bps ? nblocks_in_work - b0 : bps;
        Shard::FedSlot slot;
        slot.slot = n->fed_slots;
        slot.nb = (int)nb;
        slot.desc.assign((size_t)bps * nch, gpsbb_chan_t{}); /* (a short last slot is padded with idle blocks) */
        memcpy(slot.desc.data(), n->work.data() + (size_t)b0 * nch, (size_t)nb * nch * sizeof(gpsbb_chan_t));
        if (N > 1)
            for (int i = 0; i < nch; i++)
                if (slot.desc[i].prn > 0)
                    slot.desc[i].carr_phase = n->work_seed[(size_t)b0 * nch + i];
        Shard &s = n->shards[(size_t)(n->fed_slots % N)];
        std::unique_lock<std::mutex> lk(n->m);
        n->cv.wait(lk, [&] { return n->stop || s.fed.size() < (size_t)c.depth + 1; });
        if (n->stop)
            return GPSBB_E_STATE;
        s.fed.push_back(std::move(slot));
        n->fed_slots++;
        n->fed_blocks += nb;
        n->cv.notify_all();
    }
    return GPSBB_OK;
}

/* exact carrier phase at the first sample of every block of n->work (nblocks of them), continuing n->carry_*; leaves the
 * phase after the last block in `end` */
int feed_chain(gpsbb_node *n, long nblocks, double *end)
{
    const gpsbb_node_config_t &c = n->cfg;
    const int nch = c.nch;
    const bool fixed = (c.flags & GPSBB_NODE_FIXED_CARRIER) != 0;
    if (c.nshards == 1) {
        /* one shard: its ring chains the carrier from push to push itself (the slots are pushed in order, as a continuing
         * stream): nothing to chain here */
        for (int i = 0; i < nch; i++)
            end[i] = 0.0;
        return GPSBB_OK;
    }
    n->work_seed.resize((size_t)nblocks * nch);
    const double t0 = now_s();
    if (fixed) {
        for (int i = 0; i < nch; i++) {
            int prev_prn = n->have_carry ? n->carry_prn[i] : 0;
            uint32_t ph = n->have_carry ? (uint32_t)n->carry_phase[i] : 0u;
            for (long b = 0; b < nblocks; b++) {
                const gpsbb_chan_t &d = n->work[(size_t)b * nch + i];
                if (d.prn > 0) {
                    if (d.prn != prev_prn)
                        ph = (uint32_t)d.carr_phase;
                    n->work_seed[(size_t)b * nch + i] = (double)ph;
                    const volatile double scaled = 512.0 * 65536.0 * d.f_carr * c.delt;
                    ph += (uint32_t)c.nsamp * (uint32_t)(int32_t)std::round(scaled);
                } else {
                    n->work_seed[(size_t)b * nch + i] = 0.0;
                }
                prev_prn = d.prn > 0 ? d.prn : 0;
            }
            end[i] = (double)ph;
        }
    } else {
        /* the first block continues the stream: its descriptors with the carried phases in place of their own */
        if (n->have_carry)
            for (int i = 0; i < nch; i++)
                if (n->work[i].prn > 0 && n->work[i].prn == n->carry_prn[i])
                    n->work[i].carr_phase = n->carry_phase[i];
        long done = 0;
        while (done < nblocks) {
            const long piece = nblocks - done > 32768 ? 32768 : nblocks - done;
            if (done > 0)
                for (int i = 0; i < nch; i++) {
                    gpsbb_chan_t &d = n->work[(size_t)done * nch + i];
                    if (d.prn > 0 && d.prn == n->work[(size_t)(done - 1) * nch + i].prn)
                        d.carr_phase = end[i];
                }
            const int rc = gpsbb_chain_carrier(n->feed_h, n->work.data() + (size_t)done * nch, (int)piece, nch, c.delt, c.nsamp,
                                               n->work_seed.data() + (size_t)done * nch, end);
            if (rc != GPSBB_OK)
                return rc;
            done += piece;
        }
    }
    n->feed_chain_s += now_s() - t0;
    return GPSBB_OK;
}

} /* namespace */

extern "C" int gpsbb_node_begin(gpsbb_node_t *n, gpsbb_node_sink_fn sink, void *user)
{
    if (!n || !sink)
        return GPSBB_E_BADARG;
    if (n->feeding)
        return GPSBB_E_STATE;
    if (!n->feed_h && !(n->cfg.flags & GPSBB_NODE_FIXED_CARRIER)) {
        const int rc = gpsbb_create(&n->feed_h, n->devices[0]);
        if (rc != GPSBB_OK)
            return rc;
    }
    std::lock_guard<std::mutex> lk(n->m);
    n->ch = nullptr;
    n->nblocks = 0;
    n->sink = sink;
    n->user = user;
    n->next_block = 0;
    n->delivered = 0;
    n->stop = false;
    n->done = 0;
    n->feeding = true;
    n->fed_eof = false;
    n->pend.clear();
    n->fed_slots = n->fed_blocks = 0;
    n->have_carry = false;
    n->feed_chain_s = 0.0;
    n->feed_t0 = now_s();
    for (auto &s : n->shards) {
        s.fed.clear();
        s.rc = GPSBB_OK;
        s.first = 0;
        s.count = 0;
    }
    n->job++;
    n->cv.notify_all();
    return GPSBB_OK;
}

extern "C" int gpsbb_node_feed(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks)
{
    if (!n || !ch || nblocks < 1)
        return GPSBB_E_BADARG;
    if (!n->feeding)
        return GPSBB_E_STATE;
    const gpsbb_node_config_t &c = n->cfg;
    const int bps = c.blocks_per_slot, nch = c.nch;
    /* what is fed goes behind what is kept, in place; nothing is chained or copied again until a slot is complete (fed block by
     * block, a 400-block slot used to be copied and re-chained 400 times) */
    try {
        n->pend.insert(n->pend.end(), ch, ch + (size_t)nblocks * nch);
    } catch (const std::bad_alloc &) {
        return GPSBB_E_NOMEM;
    }
    const long have = (long)(n->pend.size() / (size_t)nch);
    const long nslots = have / bps;
    if (nslots == 0)
        return GPSBB_OK;
    n->work.swap(n->pend);
    n->pend.clear();
    /* the chain over everything at hand (cheap: 24 bytes per block and channel go to the device, a microsecond per block), the
     * whole slots out, the rest kept with the exact phase at its first block */
    double end[GPSBB_MAX_CHAN];
    int rc = GPSBB_OK;
    try {
        rc = feed_chain(n, have, end);
        if (rc == GPSBB_OK)
            rc = feed_slots(n, nslots, have);
    } catch (const std::bad_alloc &) {
        rc = GPSBB_E_NOMEM;
    }
    if (rc != GPSBB_OK)
        return rc;
    const long used = nslots * bps;
    for (int i = 0; i < nch && c.nshards > 1; i++) { /* (one shard: the ring carries the phase, feed_chain computed nothing) */
        if (used < have) {
            const gpsbb_chan_t &d = n->work[(size_t)used * nch + i];
            n->carry_prn[i] = d.prn > 0 ? d.prn : 0;
            n->carry_phase[i] = n->work_seed[(size_t)used * nch + i];
            /* (the kept blocks are chained again with the next feed: from the phase at their first block, like now) */
        } else {
            const gpsbb_chan_t &d = n->work[(size_t)(have - 1) * nch + i];
            n->carry_prn[i] = d.prn > 0 ? d.prn : 0;
            n->carry_phase[i] = end[i];
        }
    }
    n->have_carry = true;
    n->pend.assign(n->work.begin() + (size_t)used * nch, n->work.end());
    if (used < have && c.nshards > 1) /* the kept first block carries its exact phase itself: it must not be taken for a continuation twice */
        for (int i = 0; i < nch; i++)
            if (n->pend[i].prn > 0)
                n->pend[i].carr_phase = n->carry_phase[i];
    return GPSBB_OK;
}

extern "C" int gpsbb_node_end(gpsbb_node_t *n, gpsbb_node_stats_t *stats)
{
    if (!n)
        return GPSBB_E_BADARG;
    if (!n->feeding)
        return GPSBB_E_STATE;
    int rc = GPSBB_OK;
    const long left = (long)(n->pend.size() / (size_t)n->cfg.nch);
    bool stopped;
    {
        std::lock_guard<std::mutex> lk(n->m);
        stopped = n->stop;
    }
    if (left > 0 && !stopped) {
        /* the stream's last, short slot */
        double end[GPSBB_MAX_CHAN];
        try {
            n->work = n->pend;
            rc = feed_chain(n, left, end);
            if (rc == GPSBB_OK)
                rc = feed_slots(n, 1, left);
        } catch (const std::bad_alloc &) {
            rc = GPSBB_E_NOMEM;
        }
        n->pend.clear();
    }
    const int N = n->cfg.nshards;
    {
        std::unique_lock<std::mutex> lk(n->m);
        if (rc != GPSBB_OK && rc != GPSBB_E_STATE)
            n->stop = true;
        n->fed_eof = true;
        n->cv.notify_all();
        n->cv.wait(lk, [&] { return n->done == N; });
        n->feeding = false;
    }
    const int rc2 = finish_run(n, n->fed_blocks, n->feed_t0, stats);
    if (stats) /* the feeder's chain, as every shard's seed time */
        for (int g = 0; g < N; g++)
            stats->shard[g].seed_seconds = n->feed_chain_s;
    return rc != GPSBB_OK && rc != GPSBB_E_STATE ? rc : rc2;
}

namespace {
/* the driver's own sink (gpsbb_node_run_digest): called on the producer thread of the shard that rendered the slot */
int digest_sink(void *user, const int16_t *iq, long first_block, int nblocks, int shard)
{
    gpsbb_node *n = static_cast<gpsbb_node *>(user);
    Shard &s = n->shards[shard];
    uint64_t *out = n->digest_out + first_block;
    if (n->cfg.flags & GPSBB_NODE_DEVICE_ONLY) {
        if (s.slot_dig_valid) { /* rendered with its digests (shard_push / shard_pop) */
            memcpy(out, s.slot_dig.data(), (size_t)nblocks * sizeof(uint64_t));
            return 0;
        }
        return gpsbb_slot_digest(s.h, iq, nblocks, n->cfg.nsamp, out);
    }
    /* a ring in pinned host memory: the same number on the host (include/gpsbb.h, gpsbb_device_digest) */
    const uint32_t *w = reinterpret_cast<const uint32_t *>(iq);
    const size_t ns = (size_t)n->cfg.nsamp;
    for (int b = 0; b < nblocks; b++) {
        uint64_t acc = 0;
        for (size_t j = 0; j < ns; j++)
            acc += (uint64_t)w[(size_t)b * ns + j] * (uint64_t)(uint32_t)((uint32_t)j * 0x9E3779BAu + 0x85EBCA6Bu);
        out[b] = acc;
    }
    return 0;
}
} /* namespace */

extern "C" int gpsbb_node_slot_digests(gpsbb_node_t *n, int shard, uint64_t *digests, int nblocks)
{
    if (!n || !digests || shard < 0 || shard >= (int)n->shards.size() || nblocks < 1 || nblocks > n->cfg.blocks_per_slot)
        return GPSBB_E_BADARG;
    const Shard &s = n->shards[(size_t)shard];
    if (!s.slot_dig_valid)
        return GPSBB_E_STATE; /* not a GPSBB_NODE_DIGESTS node, or not called from inside the sink */
    memcpy(digests, s.slot_dig.data(), (size_t)nblocks * sizeof(uint64_t));
    return GPSBB_OK;
}

extern "C" int gpsbb_node_run_digest(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks, uint64_t *digests, gpsbb_node_stats_t *stats)
{
    if (!n || !digests)
        return GPSBB_E_BADARG;
    n->digest_out = digests;
    const int rc = gpsbb_node_run(n, ch, nblocks, digest_sink, n, stats);
    n->digest_out = nullptr;
    return rc;
}

extern "C" int gpsbb_node_run(gpsbb_node_t *n, const gpsbb_chan_t *ch, long nblocks, gpsbb_node_sink_fn sink, void *user,
                              gpsbb_node_stats_t *stats)
{
    if (!n || !ch || nblocks < 1 || !sink)
        return GPSBB_E_BADARG;
    if (n->feeding)
        return GPSBB_E_STATE; /* between gpsbb_node_begin and gpsbb_node_end */
    const int N = n->cfg.nshards;
    long first[GPSBB_NODE_MAX_SHARDS + 1];
    int rc = gpsbb_node_plan(nblocks, N, n->cfg.blocks_per_slot, first);
    if (rc != GPSBB_OK)
        return rc;
    const double t0 = now_s();
    {
        std::lock_guard<std::mutex> lk(n->m);
        n->ch = ch;
        n->nblocks = nblocks;
        n->sink = sink;
        n->user = user;
        n->next_block = 0;
        n->delivered = 0;
        n->stop = false;
        n->done = 0;
        for (int g = 0; g < N; g++) {
            n->shards[g].first = first[g];
            n->shards[g].count = first[g + 1] - first[g];
            n->shards[g].rc = GPSBB_OK;
        }
        n->job++;
        n->cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(n->m);
        n->cv.wait(lk, [&] { return n->done == N; });
    }
    return finish_run(n, nblocks, t0, stats);
}
