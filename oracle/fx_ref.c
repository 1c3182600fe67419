/* ORACLE (test infrastructure, NOT the product): plain-C restatement of the two serial FX recursions,
 * so that full-size (131072-sample) checks and bench.py's cpu_baseline finish in seconds
 * (the numpy/python loop in fx_ref.py needs ~1 s per channel).
 *
 *   ref_compressor      <- mixing_manipulator/common_audioeffects.py:529-587 (compressor_process)
 *                          called per channel with makeup 0 (:637-649), float32 in / float32 out
 *   ref_biquad_cascade  <- Equaliser.process (:500-525): per band, zero state, whole-signal
 *                          transposed-direct-form-II recursion (scipy.signal.lfilter semantics;
 *                          coefficients from pymixconsole==0.0.1 RBJ formulas, see fx_ref.py;
 *                          PARITY UNPINNED for the coefficients' origin), float64, float32 result.
 * Layout [L][C] interleaved, like the reference processors.  Checked against fx_ref.py by
 * tests/test_oracle_fx.py.
 */
#include <math.h>
#include <stdlib.h>

void ref_compressor(const float *x, float *y, long L, int C, double threshold, double attack_ms,
                    double release_ms, double ratio, double makeup, double sample_rate)
{
    const double a_att = exp(-1.0 / (0.001 * sample_rate * attack_ms));
    const double a_rel = exp(-1.0 / (0.001 * sample_rate * release_ms));
    for (int c = 0; c < C; ++c) {
        double prev = 0.0;
        for (long i = 0; i < L; ++i) {
            const double xv = (double)x[i * C + c];
            const double ax = fabs(xv);
            const double xg = (ax < 0.000001) ? -120.0 : 20.0 * log10(ax);
            double yg = 0.0;
            if (ratio > 1.0)
                yg = (xg >= threshold) ? threshold + (xg - threshold) / ratio : xg;
            else if (ratio < 1.0)
                yg = (xg <= threshold) ? threshold + (xg - threshold) / (1.0 / ratio) : xg;
            const double xl = xg - yg;
            if (xl > prev)
                prev = a_att * prev + (1.0 - a_att) * xl;
            else
                prev = a_rel * prev + (1.0 - a_rel) * xl;
            y[i * C + c] = (float)(xv * pow(10.0, (makeup - prev) / 20.0));
        }
    }
}

/* coef: n_bands rows of (b0,b1,b2,a0=1,a1,a2) */
void ref_biquad_cascade(const float *x, float *y, long L, int C, const double *coef, int n_bands)
{
    double *buf = (double *)malloc(sizeof(double) * (size_t)L);
    for (int c = 0; c < C; ++c) {
        for (long i = 0; i < L; ++i) buf[i] = (double)x[i * C + c];
        for (int b = 0; b < n_bands; ++b) {
            const double b0 = coef[6 * b], b1 = coef[6 * b + 1], b2 = coef[6 * b + 2];
            const double a1 = coef[6 * b + 4], a2 = coef[6 * b + 5];
            double z1 = 0.0, z2 = 0.0;
            for (long i = 0; i < L; ++i) {
                const double xn = buf[i];
                const double yn = b0 * xn + z1;
                z1 = b1 * xn - a1 * yn + z2;
                z2 = b2 * xn - a2 * yn;
                buf[i] = yn;
            }
        }
        for (long i = 0; i < L; ++i) y[i * C + c] = (float)buf[i];
    }
    free(buf);
}
