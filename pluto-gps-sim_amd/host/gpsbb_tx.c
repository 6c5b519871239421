/*
 * gpsbb_tx.c — the one-buffer, one-mutex, one-condvar TX hand-off of pluto-gps-sim
 * (plutogpssim.c:2146-2158 consumer, 2689 / 2757-2759 producer) with the device call abstracted.
 * See include/gpsbb_tx.h.
 */
#include "gpsbb_tx.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct gpsbb_tx {
    size_t nsamp;
    int16_t *iq_buff;  /* the shared buffer (iq_buff, c:84) */
    int16_t *dev_buf;  /* stands for the libiio buffer the TX thread copies into (ptx_buffer, c:2144) */
    pthread_mutex_t data_mutex;
    pthread_cond_t data_cond;
    pthread_t thread;
    unsigned long submitted, copied, delivered;
    int exit_flag; /* plutotx.exit */
    gpsbb_tx_push_fn push;
    void *user;
};

static void *tx_thread(void *arg)
{
    gpsbb_tx_t *tx = arg;
    for (;;) {
        pthread_mutex_lock(&tx->data_mutex);
        while (tx->copied == tx->submitted && !tx->exit_flag) /* wait for a block that has not been sent */
            pthread_cond_wait(&tx->data_cond, &tx->data_mutex);
        if (tx->copied == tx->submitted) { /* exit requested and nothing pending */
            pthread_mutex_unlock(&tx->data_mutex);
            break;
        }
        memcpy(tx->dev_buf, tx->iq_buff, tx->nsamp * 4); /* c:2148 */
        tx->copied++;
        pthread_cond_broadcast(&tx->data_cond); /* c:2149: the generator may refill */
        pthread_mutex_unlock(&tx->data_mutex);

        const int rc = tx->push(tx->user, tx->dev_buf, tx->nsamp); /* c:2152: outside the lock, paces the stream */
        pthread_mutex_lock(&tx->data_mutex);
        tx->delivered++;
        if (rc < 0)
            tx->exit_flag = 1; /* c:2153-2157 -> 2181-2184 */
        pthread_cond_broadcast(&tx->data_cond);
        const int stop = tx->exit_flag && rc < 0;
        pthread_mutex_unlock(&tx->data_mutex);
        if (stop)
            break;
    }
    return NULL;
}

int gpsbb_tx_create(gpsbb_tx_t **out, size_t nsamp, gpsbb_tx_push_fn push, void *user)
{
    if (!out || !push || nsamp == 0)
        return -1;
    gpsbb_tx_t *tx = calloc(1, sizeof *tx);
    if (!tx)
        return -4;
    tx->nsamp = nsamp;
    tx->push = push;
    tx->user = user;
    tx->iq_buff = calloc(nsamp, 4);
    tx->dev_buf = calloc(nsamp, 4);
    if (!tx->iq_buff || !tx->dev_buf) {
        free(tx->iq_buff);
        free(tx->dev_buf);
        free(tx);
        return -4;
    }
    pthread_mutex_init(&tx->data_mutex, NULL);
    pthread_cond_init(&tx->data_cond, NULL);
    if (pthread_create(&tx->thread, NULL, tx_thread, tx) != 0) {
        free(tx->iq_buff);
        free(tx->dev_buf);
        free(tx);
        return -3;
    }
    *out = tx;
    return 0;
}

int16_t *gpsbb_tx_begin(gpsbb_tx_t *tx)
{
    pthread_mutex_lock(&tx->data_mutex); /* c:2689 */
    return tx->iq_buff;
}

int gpsbb_tx_end(gpsbb_tx_t *tx)
{
    tx->submitted++;
    pthread_cond_broadcast(&tx->data_cond); /* c:2757 */
    while (tx->copied != tx->submitted && !tx->exit_flag) /* c:2758: until the TX thread has taken the block */
        pthread_cond_wait(&tx->data_cond, &tx->data_mutex);
    const int stopped = tx->exit_flag;
    pthread_mutex_unlock(&tx->data_mutex); /* c:2759 */
    return stopped ? 1 : 0;
}

void gpsbb_tx_cancel(gpsbb_tx_t *tx)
{
    pthread_mutex_unlock(&tx->data_mutex); /* nothing submitted: the TX thread keeps waiting */
}

unsigned long gpsbb_tx_delivered(gpsbb_tx_t *tx)
{
    pthread_mutex_lock(&tx->data_mutex);
    const unsigned long n = tx->delivered;
    pthread_mutex_unlock(&tx->data_mutex);
    return n;
}

void gpsbb_tx_destroy(gpsbb_tx_t *tx)
{
    if (!tx)
        return;
    pthread_mutex_lock(&tx->data_mutex);
    while (tx->delivered != tx->submitted && !tx->exit_flag) /* drain */
        pthread_cond_wait(&tx->data_cond, &tx->data_mutex);
    tx->exit_flag = 1;
    pthread_cond_broadcast(&tx->data_cond);
    pthread_mutex_unlock(&tx->data_mutex);
    pthread_join(tx->thread, NULL);
    pthread_mutex_destroy(&tx->data_mutex);
    pthread_cond_destroy(&tx->data_cond);
    free(tx->iq_buff);
    free(tx->dev_buf);
    free(tx);
}

int gpsbb_tx_push_to_file(void *user, const int16_t *iq, size_t nsamp)
{
    FILE *f = user;
    return fwrite(iq, 4, nsamp, f) == nsamp ? 0 : -1;
}
