/*
 * gpsbb_testhooks.h — exported only so that tests can exercise, on the host and without a GPU, the very
 * same exact-jump-ahead code (gpsbb_nco.h) the device pre-pass runs.  Pure functions: no state, no handle.
 * Not part of the drop-in ABI.
 */
#ifndef GPSBB_TESTHOOKS_H
#define GPSBB_TESTHOOKS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpsbb_test_row {
    int32_t n0;
    uint32_t nav;
    uint64_t xb;
    int64_t inc;
} gpsbb_test_row_t;

/* kind: 0 = code NCO, 1 = carrier NCO */
double gpsbb_test_carr_jump(double x, double s, long long n);
double gpsbb_test_code_jump(double x, double s, long long n, long long *wraps);
int gpsbb_test_build_rows(int kind, double x0, double s, unsigned nav0, int nsamp, gpsbb_test_row_t *rows,
                          int cap, double *x_end, unsigned *nav_end);
/* the builder the device pre-pass uses: xb = bits of x at n0, inc = bits of the double step S; the state at
 * sample n of a row is fma(n - n0, S, x) */
int gpsbb_test_build_rows_f64(int kind, double x0, double s, unsigned nav0, int nsamp, gpsbb_test_row_t *rows,
                              int cap, double *x_end, unsigned *nav_end);
unsigned long long gpsbb_test_row_bound(int kind, double s_abs, int nsamp);
/* the host's drift model of the carrier recurrence: the predicted phase n steps after x0 (not exact: ~1e-14) */
double gpsbb_test_carr_predict(double x0, double s, int n);
/* the fixed-point carrier's table index (fraction included) at the first sample of tile t: what k_tiles / the host hand the
 * model kernels with GPSBB_FIXED_CARRIER */
double gpsbb_test_fixed_tile_index(unsigned ph0, int step, int t);

/* The error budgets of the model kernels, measured (gpsbb_modelerr.hip.h): the realised |model - truth| of everything
 * k_synth_ev / k_synth_pd test, over every tile of the batch's last run, per channel index.  Needs a GPU. */
#define GPSBB_TEST_ME_NQ 11  /* maxima per channel: y0, y0/W, tk, tk/W, x0, x0/W, tc, tc/W, pure_y, pure_x, W (units of 2^-32) */
#define GPSBB_TEST_MEC_NQ 4  /* counts per channel: wrong unflagged decisions, lanes flagged, lane-runs looked at, always-exact */
struct gpsbb_batch;
int gpsbb_test_model_err(struct gpsbb_batch *b, double *maxima, unsigned long long *counts, int *which);
void gpsbb_test_budgets(double out[3]); /* EV_MODEL_ERR, EV_T_EPS, PD_BAND of this build, in units of 2^-32 */

#ifdef __cplusplus
}
#endif
#endif
