/*
 * gpsbb_tx.h — the TX hand-off surface of pluto-gps-sim, kept as it is, with the producer side fed by
 * libgpsbb.
 *
 * In the reference the generator thread and the libiio TX thread share ONE buffer under one mutex and one
 * condition variable:
 *   generator (main loop)   lock; fill iq_buff; signal; wait; unlock          plutogpssim.c:2689, 2757-2759
 *   pluto_tx_thread_ep      lock; memcpy(dev_buf, iq_buff); signal; unlock; iio_buffer_push(dev_buf)
 *                                                                            plutogpssim.c:2146-2158
 * This module is that protocol with the device call abstracted to a `push` callback (libiio's
 * iio_buffer_push in a real deployment, a file writer here), and with one defect of the original removed:
 * the reference's TX thread does not wait for fresh data, so it may send a block twice or send the zeroed
 * buffer before the first fill (harmless on the air, wrong in a file); here every submitted block is
 * delivered exactly once, in order.
 */
#ifndef GPSBB_TX_H
#define GPSBB_TX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* consumer side: gets one complete block (2*nsamp int16, interleaved I,Q); return < 0 to stop the stream
 * (the reference leaves its loop on a negative iio_buffer_push, c:2153-2157) */
typedef int (*gpsbb_tx_push_fn)(void *user, const int16_t *iq, size_t nsamp);

typedef struct gpsbb_tx gpsbb_tx_t;

/* allocates the shared buffer (calloc(nsamp, 4), c:2604) and starts the TX thread */
int gpsbb_tx_create(gpsbb_tx_t **out, size_t nsamp, gpsbb_tx_push_fn push, void *user);

/* generator side, in the order the reference uses them:
 *   iq = gpsbb_tx_begin(tx);      -- pthread_mutex_lock(&data_mutex)            c:2689
 *   ... fill iq[0 .. 2*nsamp) ... -- e.g. gpsbb_fill_block(..., iq, ...)
 *   rc = gpsbb_tx_end(tx);        -- cond_signal; cond_wait; unlock            c:2757-2759
 * gpsbb_tx_end returns 0, or 1 once the consumer has stopped (push returned < 0): plutotx.exit. */
int16_t *gpsbb_tx_begin(gpsbb_tx_t *tx);
int gpsbb_tx_end(gpsbb_tx_t *tx);
/* instead of gpsbb_tx_end when the fill failed: releases the buffer WITHOUT handing it to the consumer (it holds
 * the previous block, or half of a new one) */
void gpsbb_tx_cancel(gpsbb_tx_t *tx);

/* stop the TX thread (after it has delivered everything submitted) and free the surface */
void gpsbb_tx_destroy(gpsbb_tx_t *tx);

/* blocks delivered to push() so far */
unsigned long gpsbb_tx_delivered(gpsbb_tx_t *tx);

/* a ready-made consumer: appends every block to a FILE* (int16 interleaved I,Q: the .bin format of
 * gps-sdr-sim); user = FILE*.  Stops the stream on a short write. */
int gpsbb_tx_push_to_file(void *user, const int16_t *iq, size_t nsamp);

#ifdef __cplusplus
}
#endif
#endif
