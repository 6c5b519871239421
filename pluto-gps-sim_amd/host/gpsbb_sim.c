/*
 * gpsbb-sim — end-to-end generator: RINEX-2 navigation file + position/motion -> int16 I/Q file, with the
 * reference's own program structure (front end -> fill -> TX hand-off) and the fill done on an MI355X.
 *
 *   gpsbb-sim -e nav.14n [-l lat,lon,h | -c x,y,z | -u motion.csv] [-t Y/M/D,h:m:s] [-T] [-i] [-3]
 *             [-s fs_hz] [-d seconds] [-n samples_per_block] [-N channels] [-g gpu] [-F] -o out.bin
 *
 * Options mirror the reference's (plutogpssim.c:1991-2012, 2296-2390) where they concern the signal; the
 * Pluto-specific ones (-A -B -U -N host) have no meaning here.  -n defaults to 300000, the reference's
 * fixed block (plutogpssim.c:43-44); "-n 0" means fs/10 (time-continuous blocks, gps-sdr-sim semantics).
 * The main loop below is the reference's (c:2655-2806) with the inline sample loop replaced by
 * gpsbb_fill_block(): same mutex/condvar hand-off, same per-block front-end update, carrier phase fed back.
 * -F ("fast") produces the same bytes faster than real time's structure allows: the front end runs ahead,
 * blocks go through the streaming ring (gpsbb_stream_*, carrier chained exactly on host threads, pinned
 * device-to-host gather on a side stream) and are written as they pop.
 * -G N renders through the node driver (include/gpsbb_node.h): one producer thread + handle + ring per shard on the GPUs
 * "-g a,b,c" names (default 0 .. N-1; an ordinal may repeat), ONE output: a regular file is written with pwrite() as slots
 * complete (GPSBB_NODE_INDEXED), a pipe in stream order.  The stream is fed INCREMENTALLY (gpsbb_node_begin / _feed / _end):
 * the front end makes the descriptors of a few slots, feeds them, and goes round again like the reference's loop — the slots go
 * round the GPUs, each from the exact carrier phase the feeder chained, and the memory used does not grow with -d.
 * -G N -C: N CONTIGUOUS time shards instead (BASELINE configs[4]'s layout; gpsbb_node_run: the whole descriptor sequence up
 * front, each shard seeded by the device-side chain over everything before it); -C -I: the whole sequence up front, slots round
 * the GPUs (GPSBB_NODE_INTERLEAVED).
 * -P usec paces the consumer like the radio does: the TX surface's sink takes one block every `usec` microseconds (the
 * reference's iio_buffer_push blocks until the hardware has room, c:2152; 100000 = real time, less = compressed time), counts
 * the blocks that were not there when their turn came (under-runs) and, with -S file, writes the latency distribution of
 * the drop-in call gpsbb_fill_block (p50 / p99 / max over all blocks) as JSON.  -Q n: blocks the device queues (libiio's
 * kernel buffers: 4 by default; a push only blocks when all are taken).  -R: do not register iq_buff with the library
 * (gpsbb_host_register: by default the drop-in call renders straight into it; with -R it ends with a copy, as for any buffer).  -k a,b,c keeps only those blocks in the
 * output file (a soak of hours of signal need not write them all).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "gpsbb.h"
#include "gpsbb_node.h"
#include "gpsbb_tx.h"
#include "gpsfe.h"

/* the node driver's one consumer: a file (random access) or a pipe (ordered) */
struct node_out {
    FILE *f;
    int fd;         /* >= 0: pwrite at the block's offset */
    size_t nsamp;
};

static int node_sink(void *user, const int16_t *iq, long first_block, int nblocks, int shard)
{
    struct node_out *o = user;
    (void)shard;
    const size_t bytes = (size_t)nblocks * o->nsamp * 4;
    if (o->fd >= 0) {
        const char *p = (const char *)iq;
        off_t at = (off_t)first_block * (off_t)o->nsamp * 4;
        size_t left = bytes;
        while (left) {
            const ssize_t w = pwrite(o->fd, p, left, at);
            if (w <= 0)
                return -1;
            p += w;
            at += w;
            left -= (size_t)w;
        }
        return 0;
    }
    return fwrite(iq, 1, bytes, o->f) == bytes ? 0 : -1;
}

/* the paced consumer of the soak: what sits behind the TX surface instead of the Pluto */
struct paced_sink {
    FILE *f;
    long period_ns;         /* 0: not paced */
    int queue;              /* blocks the device holds: libiio hands a pushed buffer to the kernel and blocks only when all
                               of its kernel buffers (4 by default) are queued, so the generator may run that far ahead */
    long keep[64];
    int nkeep;              /* 0: keep everything */
    long seen;              /* blocks taken so far */
    long underruns;         /* blocks that arrived after their slot had begun */
    double worst_late_ms;
    struct timespec t0;     /* the first block's arrival: slot k begins at t0 + k * period */
};

static double ts_ms(const struct timespec *a, const struct timespec *b)
{
    return (double)(a->tv_sec - b->tv_sec) * 1e3 + (double)(a->tv_nsec - b->tv_nsec) * 1e-6;
}

static int paced_push(void *user, const int16_t *iq, size_t nsamp)
{
    struct paced_sink *p = user;
    const long k = p->seen++;
    if (p->period_ns > 0) {
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (k == 0)
            p->t0 = now;
        /* block k goes on the air at t0 + k * period; the push returns when the device has room again, i.e. when block
         * k - (queue - 1) has started playing */
        struct timespec due = p->t0, room = p->t0;
        const long long ns = (long long)p->t0.tv_nsec + (long long)k * p->period_ns;
        due.tv_sec += (time_t)(ns / 1000000000LL);
        due.tv_nsec = (long)(ns % 1000000000LL);
        const long long kr = k - (p->queue - 1) > 0 ? k - (p->queue - 1) : 0;
        const long long nr = (long long)p->t0.tv_nsec + kr * p->period_ns;
        room.tv_sec += (time_t)(nr / 1000000000LL);
        room.tv_nsec = (long)(nr % 1000000000LL);
        const double late = ts_ms(&now, &due);
        if (late > 0.0) { /* the radio had nothing to send when this block's slot began */
            p->underruns++;
            if (late > p->worst_late_ms)
                p->worst_late_ms = late;
        }
        if (ts_ms(&now, &room) < 0.0)
            clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &room, NULL); /* iio_buffer_push blocks until there is room (c:2152) */
    }
    int want = p->nkeep == 0;
    for (int i = 0; i < p->nkeep; i++)
        want = want || p->keep[i] == k;
    if (want && fwrite(iq, 4, nsamp, p->f) != nsamp)
        return -1;
    return 0;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static void usage(void)
{
    fprintf(stderr, "usage: gpsbb-sim -e nav [-l lat,lon,h|-c x,y,z|-u motion.csv] [-t Y/M/D,h:m:s] [-T] [-i] [-3]\n"
                    "                 [-s fs_hz] [-d seconds] [-n samples_per_block] [-N channels] [-g gpu[,gpu...]] [-F] [-G shards [-C [-I]]]\n"
                    "                 [-P usec_per_block] [-Q device_queue_blocks] [-S stats.json] [-k keep,blocks] -o out.bin\n");
}

int main(int argc, char **argv)
{
    /* libgpsbb keeps up to nine HIP streams busy; the runtime's default of four hardware queues would make unrelated
     * streams share one.  This is the host's call (before its first HIP call); the library does not touch the environment. */
    setenv("GPU_MAX_HW_QUEUES", "12", 0);
    gpsfe_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.pos[0] = 35.681298; /* default static location: Tokyo (c:2266-2268) */
    cfg.pos[1] = 139.766247;
    cfg.pos[2] = 10.0;
    cfg.max_chan = 12; /* MAX_CHAN h:21 */
    long fs_hz = 3000000; /* TX_SAMPLE_FREQ c:43 */
    long nsamp = 300000;  /* NUM_SAMPLES c:44 */
    double duration = 1.0;
    int gpu = 0, opt, fast = 0, no_register = 0, nshards = 0, ndev = 0, interleaved = 0, contiguous = 0;
    int16_t *iq_registered = NULL;
    int devs[GPSBB_NODE_MAX_SHARDS];
    struct paced_sink paced;
    memset(&paced, 0, sizeof paced);
    paced.queue = 4; /* libiio's default number of kernel buffers */
    const char *stats_path = NULL;
    const char *out_path = NULL;

    while ((opt = getopt(argc, argv, "e:u:c:l:s:Tt:in:N:d:o:g:3FG:P:S:k:Q:ICR")) != -1) {
        switch (opt) {
        case 'e': cfg.navfile = optarg; break;
        case 'u': cfg.motion_file = optarg; break;
        case 'c':
        case 'l':
            cfg.use_ecef = opt == 'c';
            if (sscanf(optarg, "%lf,%lf,%lf", &cfg.pos[0], &cfg.pos[1], &cfg.pos[2]) != 3) {
                fprintf(stderr, "ERROR: -%c wants three comma-separated numbers\n", opt);
                return 1;
            }
            break;
        case 's':
            fs_hz = atol(optarg);
            if (fs_hz < 1000000) { /* c:2326 */
                fprintf(stderr, "ERROR: Invalid sampling frequency.\n");
                return 1;
            }
            break;
        case 'T': cfg.time_overwrite = 1; break;
        case 't':
            cfg.have_start = 1;
            if (sscanf(optarg, "%d/%d/%d,%d:%d:%lf", &cfg.y, &cfg.m, &cfg.d, &cfg.hh, &cfg.mm, &cfg.sec) != 6) {
                fprintf(stderr, "ERROR: -t wants YYYY/MM/DD,hh:mm:ss\n");
                return 1;
            }
            break;
        case 'i': cfg.iono_disable = 1; break;
        case '3': cfg.rinex3 = 1; break; /* the reference declares -3 with an argument (c:2296); here it is a flag */
        case 'n': nsamp = atol(optarg); break;
        case 'N': cfg.max_chan = atoi(optarg); break;
        case 'd': duration = atof(optarg); break;
        case 'o': out_path = optarg; break;
        case 'g': /* one ordinal, or the list of the node driver's shards */
            ndev = 0;
            for (char *t = strtok(optarg, ","); t && ndev < GPSBB_NODE_MAX_SHARDS; t = strtok(NULL, ","))
                devs[ndev++] = atoi(t);
            gpu = ndev ? devs[0] : 0;
            break;
        case 'G': nshards = atoi(optarg); break;
        case 'I': interleaved = 1; break;
        case 'C': contiguous = 1; break; /* (with -I alone: the whole sequence up front as well: GPSBB_NODE_INTERLEAVED) */
        case 'P': paced.period_ns = atol(optarg) * 1000L; break;
        case 'S': stats_path = optarg; break;
        case 'Q': paced.queue = atoi(optarg) > 0 ? atoi(optarg) : 1; break;
        case 'k':
            for (char *t = strtok(optarg, ","); t && paced.nkeep < 64; t = strtok(NULL, ","))
                paced.keep[paced.nkeep++] = atol(t);
            break;
        case 'F': fast = 1; break;
        case 'R': no_register = 1; break; /* the drop-in call copies into iq_buff instead of rendering straight into it */
        default: usage(); return 1;
        }
    }
    if (!cfg.navfile || !out_path) {
        usage();
        return 1;
    }
    if (nsamp == 0)
        nsamp = fs_hz / 10;
    const double delt = 1.0 / (double)fs_hz; /* c:2397 */
    const long nblocks = (long)(duration * 10.0 + 0.5);

    gpsfe_t *fe = NULL;
    int rc = gpsfe_open(&cfg, &fe);
    if (rc != GPSFE_OK) {
        fprintf(stderr, "ERROR: %s\n", gpsfe_strerror(rc));
        return 1;
    }
    if (nshards > 0) {
        /* the whole descriptor sequence first (the host range solver runs ahead of everything: 296 bytes per block-channel),
         * then N time shards on N handles into one output */
        if (nshards > GPSBB_NODE_MAX_SHARDS || (ndev > 1 && ndev != nshards) || nblocks < 1) {
            fprintf(stderr, "ERROR: -G wants 1..%d shards and as many -g ordinals\n", GPSBB_NODE_MAX_SHARDS);
            return 1;
        }
        for (int g = 0; g < nshards; g++)
            devs[g] = ndev > 1 ? devs[g] : (ndev == 1 && nshards == 1 ? devs[0] : (ndev == 1 ? devs[0] + g : g));
        /* Default: the stream as the reference makes it, incrementally (gpsbb_node_begin / _feed / _end): the front end generates
         * a few slots' worth of descriptors, feeds them, and goes round again while the GPUs render — it runs one queue ahead of
         * the rings, and the memory this takes does not depend on -d.  -C: contiguous time shards (BASELINE configs[4]'s layout),
         * which need the whole descriptor sequence up front (296 bytes per block and channel). */
        const int bps = nblocks < 16 ? (int)nblocks : 16;
        contiguous = contiguous || interleaved;
        const long chunk = contiguous ? nblocks : (long)bps * nshards * 2;
        gpsbb_chan_t *all = malloc((size_t)chunk * cfg.max_chan * sizeof *all);
        FILE *fo = strcmp(out_path, "-") ? fopen(out_path, "wb") : stdout;
        if (!all || !fo) {
            fprintf(stderr, "ERROR: cannot allocate the descriptors / open %s\n", out_path);
            return 1;
        }
        struct node_out o = {fo, -1, (size_t)nsamp};
        unsigned nflags = interleaved ? GPSBB_NODE_INTERLEAVED : 0u; /* -C -I: the slots go round the GPUs (an ordered output scales) */
        if (fo != stdout && ftruncate(fileno(fo), (off_t)nblocks * nsamp * 4) == 0) {
            o.fd = fileno(fo); /* a regular file: blocks are placed by index as they complete, from every shard at once */
            nflags |= GPSBB_NODE_INDEXED | GPSBB_NODE_CONCURRENT;
        }
        gpsbb_node_config_t nc = {nshards, devs, cfg.max_chan, delt, (int)nsamp, bps, 3, nflags};
        gpsbb_node_t *node = NULL;
        gpsbb_node_stats_t ns;
        rc = gpsbb_node_create(&node, &nc);
        if (rc == GPSBB_OK && contiguous) {
            gpsfe_generate(fe, (int)nblocks, all);
            rc = gpsbb_node_run(node, all, nblocks, node_sink, &o, &ns);
        } else if (rc == GPSBB_OK) {
            rc = gpsbb_node_begin(node, node_sink, &o);
            for (long done = 0; rc == GPSBB_OK && done < nblocks; done += chunk) { /* while (!plutotx.exit), c:2655 */
                const long nb = nblocks - done < chunk ? nblocks - done : chunk;
                gpsfe_generate(fe, (int)nb, all);                  /* c:2656-2687 (+ c:2764-2805) */
                rc = gpsbb_node_feed(node, all, nb);
            }
            const int rc_end = gpsbb_node_end(node, &ns);
            rc = rc == GPSBB_OK || rc == GPSBB_E_STATE ? rc_end : rc;
        }
        if (rc != GPSBB_OK)
            fprintf(stderr, "ERROR: node driver: %s\n", gpsbb_strerror(rc));
        else
            for (int g = 0; g < ns.nshards; g++)
                fprintf(stderr, "shard %d: gpu %d (numa node %d, %d cpus), blocks %ld..%ld, seed %.3f s, busy %.3f s\n", g,
                        ns.shard[g].device, ns.shard[g].numa_node, ns.shard[g].cpus_bound, ns.shard[g].first_block,
                        ns.shard[g].first_block + ns.shard[g].nblocks, ns.shard[g].seed_seconds, ns.shard[g].busy_seconds);
        gpsbb_node_destroy(node);
        free(all);
        if (fo != stdout)
            fclose(fo);
        else
            fflush(fo);
        gpsfe_close(fe);
        if (rc == GPSBB_OK)
            fprintf(stderr, "%ld blocks of %ld samples written\n", nblocks, nsamp);
        return rc == GPSBB_OK ? 0 : 1;
    }
    gpsbb_t *bb = NULL;
    rc = gpsbb_create(&bb, gpu);
    if (rc != GPSBB_OK) {
        fprintf(stderr, "ERROR: gpsbb_create: %s\n", gpsbb_strerror(rc));
        return 1;
    }
    FILE *fout = strcmp(out_path, "-") ? fopen(out_path, "wb") : stdout;
    if (!fout) {
        fprintf(stderr, "ERROR: cannot open %s\n", out_path);
        return 1;
    }
    if (fast) {
        /* offline generation: slots of up to 16 blocks through the ring, three in flight */
        const int bps = nblocks < 16 ? (int)(nblocks > 0 ? nblocks : 1) : 16, depth = 3;
        gpsbb_stream_t *st = NULL;
        gpsbb_chan_t *slot = malloc((size_t)bps * cfg.max_chan * sizeof *slot);
        rc = slot ? gpsbb_stream_create(bb, cfg.max_chan, delt, (int)nsamp, bps, depth, GPSBB_CHAIN_CARRIER, &st) : GPSBB_E_NOMEM;
        long pushed = 0, written = 0;
        while (rc == GPSBB_OK && slot && written < nblocks) {
            while (rc == GPSBB_OK && pushed < nblocks && gpsbb_stream_pending(st) < depth) {
                /* a short last slot is padded with blocks that are generated but not written */
                gpsfe_generate(fe, bps, slot);
                rc = gpsbb_stream_push(st, slot);
                pushed += bps;
            }
            const int16_t *iq = NULL;
            if (rc == GPSBB_OK)
                rc = gpsbb_stream_pop(st, &iq, NULL);
            if (rc == GPSBB_OK) {
                const long n = nblocks - written < bps ? nblocks - written : bps;
                if (fwrite(iq, 4, (size_t)n * (size_t)nsamp, fout) != (size_t)n * (size_t)nsamp)
                    rc = GPSBB_E_STATE;
                written += n;
            }
        }
        if (rc != GPSBB_OK)
            fprintf(stderr, "ERROR: streaming: %s\n", gpsbb_strerror(rc));
        if (st)
            gpsbb_stream_destroy(st);
        free(slot);
        if (fout != stdout)
            fclose(fout);
        gpsbb_destroy(bb);
        gpsfe_close(fe);
        fprintf(stderr, "%ld blocks of %ld samples written\n", written, nsamp);
        return rc == GPSBB_OK ? 0 : 1;
    }

    gpsbb_tx_t *tx = NULL;
    paced.f = fout;
    double *lat_ms = calloc((size_t)(nblocks > 0 ? nblocks : 1), sizeof *lat_ms);
    if (gpsbb_tx_create(&tx, (size_t)nsamp, paced_push, &paced) != 0 || !lat_ms) {
        fprintf(stderr, "ERROR: cannot start the TX surface\n");
        return 1;
    }

    fprintf(stderr, "PRN   Az    El     Range     Iono\n"); /* the table the reference prints, c:2634-2639 */
    for (int i = 0; i < cfg.max_chan; i++) {
        int prn;
        double az, el, range, iono;
        gpsfe_channel_info(fe, i, &prn, &az, &el, &range, &iono);
        if (prn > 0)
            fprintf(stderr, "%02d %6.1f %5.1f %11.1f %5.1f\n", prn, az, el, range, iono);
    }

    gpsbb_chan_t ch[GPSBB_MAX_CHAN];
    gpsbb_chan_state_t st[GPSBB_MAX_CHAN];
    long blk;
    for (blk = 0; blk < nblocks; blk++) { /* while (!plutotx.exit), c:2655 */
        gpsfe_next_block(fe, ch);                                   /* c:2656-2687 (+ c:2764-2805) */
        if (blk == 0) {
            /* The first two calls on a handle pay for what every later one finds in place (code objects loaded, scratch
             * and both table sets allocated: 18 and 10 ms against 0.3): render the first block twice into a scratch buffer
             * before the consumer's clock starts.  A fill has no side effects besides its outputs ... */
            int16_t *scratch = malloc((size_t)nsamp * 4);
            for (int w = 0; scratch && w < 2; w++)
                (void)gpsbb_fill_block(bb, ch, cfg.max_chan, delt, (int)nsamp, scratch, NULL);
            free(scratch);
            /* ... besides the handle's hazard counters: block 0 must count once in the note at the end, not three times */
            gpsbb_hazards_t warm;
            (void)gpsbb_get_hazards(bb, &warm, 1);
        }
        int16_t *iq = gpsbb_tx_begin(tx);                           /* c:2689 */
        if (blk == 0 && !no_register && gpsbb_host_register(bb, iq, (size_t)nsamp * 4) == GPSBB_OK)
            iq_registered = iq; /* iq_buff is one allocation for the run (c:2604): rendered into directly from here on */
        struct timespec ta, tb;
        clock_gettime(CLOCK_MONOTONIC, &ta);
        rc = gpsbb_fill_block(bb, ch, cfg.max_chan, delt, (int)nsamp, iq, st); /* replaces c:2690-2756 */
        clock_gettime(CLOCK_MONOTONIC, &tb);
        lat_ms[blk] = ts_ms(&tb, &ta);
        if (rc != GPSBB_OK) { /* the buffer holds no valid block: it must not reach the sink */
            gpsbb_tx_cancel(tx);
            fprintf(stderr, "ERROR: gpsbb_fill_block: %s\n", gpsbb_strerror(rc));
            break;
        }
        const int stopped = gpsbb_tx_end(tx);                       /* c:2757-2759 */
        gpsfe_feed_back(fe, st); /* the loop's in-place update of chan[i].carr_phase */
        if (stopped)
            break;
    }
    if (iq_registered)
        (void)gpsbb_host_unregister(bb, iq_registered); /* before the buffer is freed (c:2815) */
    gpsbb_tx_destroy(tx);
    if (fout != stdout)
        fclose(fout);
    if (stats_path && blk > 0) {
        /* the drop-in call's latency over the whole run, and what the paced consumer saw */
        /* which blocks were slow (before the sort loses the order): the first 32 above 2 ms */
        char slow[2048];
        size_t so = 0;
        int nslow = 0;
        slow[0] = 0;
        for (long i = 0; i < blk; i++)
            if (lat_ms[i] > 2.0) {
                if (nslow < 32 && so + 40 < sizeof slow)
                    so += (size_t)snprintf(slow + so, sizeof slow - so, "%s[%ld, %.2f]", nslow ? ", " : "", i, lat_ms[i]);
                nslow++;
            }
        qsort(lat_ms, (size_t)blk, sizeof *lat_ms, cmp_double);
        double sum = 0.0;
        for (long i = 0; i < blk; i++)
            sum += lat_ms[i];
        FILE *sf = fopen(stats_path, "w");
        if (sf) {
            fprintf(sf, "{\"blocks\": %ld, \"nsamp\": %ld, \"channels\": %d, \"fs_hz\": %ld, \"period_us\": %.1f, \"device_queue_blocks\": %d, "
                        "\"fill_block_ms\": {\"mean\": %.4f, \"p50\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"p999\": %.4f, \"max\": %.4f}, "
                        "\"underruns\": %ld, \"worst_late_ms\": %.4f, \"delivered\": %ld, \"blocks_over_2ms\": %d, \"first_slow_blocks\": [%s]}\n",
                    blk, nsamp, cfg.max_chan, fs_hz, (double)paced.period_ns / 1e3, paced.queue, sum / (double)blk, lat_ms[blk / 2],
                    lat_ms[(long)((double)blk * 0.9)], lat_ms[(long)((double)blk * 0.99)], lat_ms[(long)((double)blk * 0.999)],
                    lat_ms[blk - 1], paced.underruns, paced.worst_late_ms, paced.seen, nslow, slow);
            fclose(sf);
        }
    }
    free(lat_ms);
    gpsbb_hazards_t hz;
    if (gpsbb_get_hazards(bb, &hz, 0) == GPSBB_OK && (hz.itable_512 || hz.dwrd_oob))
        fprintf(stderr, "note: latent out-of-bounds cases of the reference hit: table %llu, nav words %llu\n",
                (unsigned long long)hz.itable_512, (unsigned long long)hz.dwrd_oob);
    gpsbb_destroy(bb);
    gpsfe_close(fe);
    fprintf(stderr, "%ld blocks of %ld samples written\n", blk, nsamp);
    return rc == GPSBB_OK ? 0 : 1;
}
