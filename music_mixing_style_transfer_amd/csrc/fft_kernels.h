// Power-of-two real FFTs for the FX rows that convolve or analyse in the frequency domain (reference scipy.signal.oaconvolve / filtfilt in
// common_audioeffects.py:727-764 and utils_data_normalization.py:93-102, librosa.stft in common_miscellaneous.py:50-77): batched R2C and C2R
// transforms of length n = 2^k with hipFFT's conventions (unnormalised, n / 2 + 1 bins per sequence) - hand-written, so that the library has
// no run-time dependency (hipFFT's kernels are compiled at first use: 4 s on the first song of a process) and rows f-3 / f-4 run on this
// repository's own gfx950 code like everything else.
//
// A real sequence of n samples is read as m = n / 2 complex numbers z[k] = x[2k] + i x[2k + 1] (a reinterpretation, no pass), transformed by
// a complex FFT of size m, and un-mixed: X[k] = (Z[k] + conj Z[m - k]) / 2 - i w^k (Z[k] - conj Z[m - k]) / 2, w = exp(-2 pi i / n).  The
// complex FFT is the Stockham autosort form, radix 4 (one radix-2 pass first when log2(m) is odd): out-of-place passes between two buffers,
// every pass reads contiguous runs and writes runs of Ns elements - no bit reversal, no LDS, HBM- / cache-bound (a 65536-point frame is 256 KB: batches of 64 live in the
// 256 MB infinity cache).  Twiddles come from a table exp(-2 pi i j / m), j < m / 2, computed once per plan in float64.
// Accuracy: float32 butterflies, ~log2(n) ulp - the same class as the library it replaces (tests/test_fft.py against numpy.fft).
#pragma once
#include "mst_dev.h"

// tw[j] = exp(-2 pi i j / len), j < count (float64 sine / cosine, rounded once)
__global__ __launch_bounds__(256) void fft_twiddle_kernel(float2 *tw, long len, long count) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    const double a = -2.0 * 3.14159265358979323846 * (double)j / (double)len;
    tw[j] = make_float2((float)cos(a), (float)sin(a));
}

// one radix-2 Stockham pass of a size-m complex FFT over a batch: thread j < m / 2 of sequence blockIdx.y
//   k = j mod Ns;  a = in[j], b = w(k) in[j + m / 2];  out[(j / Ns) 2 Ns + k] = a + b, out[... + Ns] = a - b;  w(k) = exp(-+ 2 pi i k / (2 Ns))
__global__ __launch_bounds__(256) void fft_stockham2_kernel(const float2 *in, float2 *out, const float2 *tw, long m, long Ns, long stride_in,
                                                            long stride_out, int inverse) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= m / 2) return;
    const float2 *src = in + (size_t)blockIdx.y * stride_in;
    float2 *dst = out + (size_t)blockIdx.y * stride_out;
    const long k = j & (Ns - 1);
    float2 w = tw[k * (m / (2 * Ns))];
    if (inverse) w.y = -w.y;
    const float2 a = src[j], b0 = src[j + m / 2];
    const float2 b = make_float2(w.x * b0.x - w.y * b0.y, w.x * b0.y + w.y * b0.x);
    const long j0 = ((j - k) << 1) + k;
    dst[j0] = make_float2(a.x + b.x, a.y + b.y);
    dst[j0 + Ns] = make_float2(a.x - b.x, a.y - b.y);
}

// one radix-4 Stockham pass: thread j < m / 4;  k = j mod Ns;  a_t = w(t k) in[j + t m / 4], w(u) = exp(-+ 2 pi i u / (4 Ns));  a 4-point DFT;
// out[(j / Ns) 4 Ns + k + t Ns].  Twiddles from the size-m table: exp(-2 pi i u / (4 Ns)) = T[u m / (4 Ns)], T[i >= m / 2] = -T[i - m / 2].
__device__ __forceinline__ float2 fft_tw(const float2 *tw, long idx, long half, int inverse) {
    float2 w = idx < half ? tw[idx] : tw[idx - half];
    if (idx >= half) { w.x = -w.x; w.y = -w.y; }
    if (inverse) w.y = -w.y;
    return w;
}
__device__ __forceinline__ float2 fft_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__global__ __launch_bounds__(256) void fft_stockham4_kernel(const float2 *in, float2 *out, const float2 *tw, long m, long Ns, long stride_in,
                                                            long stride_out, int inverse) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long q = m / 4;
    if (j >= q) return;
    const float2 *src = in + (size_t)blockIdx.y * stride_in;
    float2 *dst = out + (size_t)blockIdx.y * stride_out;
    const long k = j & (Ns - 1);
    const long step = k * (m / (4 * Ns));                  // table index of w(k)
    const float2 a0 = src[j];
    const float2 a1 = fft_cmul(fft_tw(tw, step, m / 2, inverse), src[j + q]);
    const float2 a2 = fft_cmul(fft_tw(tw, 2 * step, m / 2, inverse), src[j + 2 * q]);
    const float2 a3 = fft_cmul(fft_tw(tw, 3 * step, m / 2, inverse), src[j + 3 * q]);
    const float2 s02 = make_float2(a0.x + a2.x, a0.y + a2.y), d02 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 s13 = make_float2(a1.x + a3.x, a1.y + a3.y), d13 = make_float2(a1.x - a3.x, a1.y - a3.y);
    // forward: -i d13 = (d13.y, -d13.x); inverse: +i d13 = (-d13.y, d13.x)
    const float2 r = inverse ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
    const long j0 = ((j - k) << 2) + k;
    dst[j0] = make_float2(s02.x + s13.x, s02.y + s13.y);
    dst[j0 + Ns] = make_float2(d02.x + r.x, d02.y + r.y);
    dst[j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
    dst[j0 + 3 * Ns] = make_float2(d02.x - r.x, d02.y - r.y);
}

// ------------------------------------------------------------------------------------------------
// Four-step form for m >= 256 (m = m1 * m2, m1 = 2^ceil(k/2) <= 1024, m2 = 2^floor(k/2) >= 16): two kernels instead of log4(m) global passes -
//   fft_cols_kernel   for 16 consecutive columns n2: the m1-point FFTs over n1 of X[n1][n2] = z[n1 m2 + n2] in LDS (in-place radix-2 on
//                     bit-reversed positions), times W_m^(n2 k1), stored as T[k1][n2];
//   fft_rows_kernel   for 16 consecutive rows k1: the m2-point FFTs over n2 in LDS, stored transposed: Z[k1 + m1 k2].
// Every global access is a run of 16 complex numbers (128 B) or longer; the array is read and written twice in all (a 2^17-point transform
// took nine global passes in the Stockham form).  Sub-FFT twiddles come from the size-m table: W_m1^j = T[j m2], W_m2^j = T[j m1].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fft_bitrev(unsigned v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

// the in-LDS transform of 16 sequences of len = 2^lg points, sequence c at buf + c * pitch (bit-reversed order on entry); wl[j] = W_len^j,
// j < len / 2, in LDS too (a global twiddle load per butterfly is an exposed L2 round trip per loop iteration: measured 4 x the kernel time)
__device__ __forceinline__ void fft_lds16(float2 *buf, int pitch, int lg, const float2 *wl, int inverse, int tid) {
    const int len = 1 << lg, half = len >> 1;
    for (int h = 1; h < len; h <<= 1) {
        const int ws = half / h;
#pragma unroll 4
        for (int e = tid; e < 16 * half; e += 256) {
            const int c = e & 15, b = e >> 4;
            const int k = b & (h - 1);
            const int p0 = ((b - k) << 1) + k, p1 = p0 + h;
            float2 w = wl[k * ws];
            if (inverse) w.y = -w.y;
            float2 *q = buf + c * pitch;
            const float2 u = q[p0], v = fft_cmul(w, q[p1]);
            q[p0] = make_float2(u.x + v.x, u.y + v.y);
            q[p1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

template <int LOGMAX>
__global__ __launch_bounds__(256) void fft_cols_kernel(const float2 *in, float2 *out, const float2 *tw, long m, int l1, int l2, long stride_in,
                                                       long stride_out, int inverse) {
    __shared__ float2 buf[16 * ((1 << LOGMAX) + 1)];
    __shared__ float2 wl[(1 << LOGMAX) / 2];
    const int tid = threadIdx.x, m1 = 1 << l1, pitch = m1 + 1;
    const long m2 = 1L << l2, n2base = (long)blockIdx.x * 16;
    const float2 *src = in + (size_t)blockIdx.y * stride_in;
    float2 *dst = out + (size_t)blockIdx.y * stride_out;
    for (int j = tid; j < m1 / 2; j += 256) wl[j] = tw[(long)j * m2];          // W_m1^j = T[j m2]
#pragma unroll 8
    for (int e = tid; e < 16 * m1; e += 256) {
        const int c = e & 15, n1 = e >> 4;
        buf[c * pitch + fft_bitrev((unsigned)n1, l1)] = src[(long)n1 * m2 + n2base + c];
    }
    __syncthreads();
    fft_lds16(buf, pitch, l1, wl, inverse, tid);
#pragma unroll 8
    for (int e = tid; e < 16 * m1; e += 256) {
        const int c = e & 15, k1 = e >> 4;
        const long idx = ((n2base + c) * (long)k1) & (m - 1);
        dst[(long)k1 * m2 + n2base + c] = fft_cmul(fft_tw(tw, idx, m / 2, inverse), buf[c * pitch + k1]);
    }
}

template <int LOGMAX>
__global__ __launch_bounds__(256) void fft_rows_kernel(const float2 *in, float2 *out, const float2 *tw, long m, int l1, int l2, long stride_in,
                                                       long stride_out, int inverse) {
    __shared__ float2 buf[16 * ((1 << LOGMAX) + 1)];
    __shared__ float2 wl[(1 << LOGMAX) / 2];
    const int tid = threadIdx.x, m2 = 1 << l2, pitch = m2 + 1;
    const long m1 = 1L << l1, k1base = (long)blockIdx.x * 16;
    const float2 *src = in + (size_t)blockIdx.y * stride_in;
    float2 *dst = out + (size_t)blockIdx.y * stride_out;
    for (int j = tid; j < m2 / 2; j += 256) wl[j] = tw[(long)j * m1];          // W_m2^j = T[j m1]
#pragma unroll 8
    for (int e = tid; e < 16 * m2; e += 256) {
        const int rr = e >> l2, n2 = e & (m2 - 1);                  // consecutive threads read consecutive elements of a row
        buf[rr * pitch + fft_bitrev((unsigned)n2, l2)] = src[(k1base + rr) * m2 + n2];
    }
    __syncthreads();
    fft_lds16(buf, pitch, l2, wl, inverse, tid);
#pragma unroll 8
    for (int e = tid; e < 16 * m2; e += 256) {
        const int c = e & 15, k2 = e >> 4;                           // 16 consecutive k1 of one k2: a 128-byte run
        dst[k1base + c + m1 * (long)k2] = buf[c * pitch + k2];
    }
}

// R2C un-mix: Z (m complex per sequence, stride sz) -> X (m + 1 bins per sequence, stride sx); thread k <= m / 2 handles the pair (k, m - k),
// so Z and X may be the same buffer (sz == sx).  twn[k] = exp(-2 pi i k / n), k <= m / 2.
__global__ __launch_bounds__(256) void fft_r2c_post_kernel(const float2 *Z, float2 *X, const float2 *twn, long m, long sz, long sx) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k > m / 2) return;
    const float2 *z = Z + (size_t)blockIdx.y * sz;
    float2 *x = X + (size_t)blockIdx.y * sx;
    if (k == 0) {
        const float2 z0 = z[0];
        x[0] = make_float2(z0.x + z0.y, 0.0f);
        x[m] = make_float2(z0.x - z0.y, 0.0f);
        return;
    }
    const long kc = m - k;
    const float2 a = z[k], b = z[kc];                       // Z[k], Z[m - k]
    // E = (a + conj b) / 2, D = (a - conj b) / 2;  X[k] = E - i w D;  X[m - k] = conj(E) - i conj(w) (-conj D) ... written out below
    const float ex = 0.5f * (a.x + b.x), ey = 0.5f * (a.y - b.y);
    const float dx = 0.5f * (a.x - b.x), dy = 0.5f * (a.y + b.y);
    const float2 w = twn[k];
    // -i w D = (w.y dx + w.x dy) - i (w.x dx - w.y dy)
    const float px = w.y * dx + w.x * dy, py = -(w.x * dx - w.y * dy);
    x[k] = make_float2(ex + px, ey + py);
    // X[m - k] = conj(X_even[k]) - i w(m - k) * (-conj D) with w(m - k) = -conj(w):  = conj(E) - conj(-i w D) = (ex - px) + i (-ey + py)
    if (kc != k) x[kc] = make_float2(ex - px, -ey + py);
}

// C2R mix: X (m + 1 bins, stride sx) -> Z' (m complex, stride sz) with Z'[k] = (X[k] + conj X[m - k]) + i conj(w)^k... = E2 + i e^{+2 pi i k / n} D2;
// the unnormalised inverse complex FFT of Z' read as interleaved reals is n * irfft(X).  Pairs (k, m - k) per thread: in place when sz == sx.
__global__ __launch_bounds__(256) void fft_c2r_pre_kernel(const float2 *X, float2 *Z, const float2 *twn, long m, long sx, long sz) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k > m / 2) return;
    const float2 *x = X + (size_t)blockIdx.y * sx;
    float2 *z = Z + (size_t)blockIdx.y * sz;
    const long kc = m - k;
    const float2 a = x[k], b = x[kc];                       // X[k], X[m - k]  (k = 0: X[0], X[m])
    const float ex = a.x + b.x, ey = a.y - b.y;             // a + conj b
    const float dx = a.x - b.x, dy = a.y + b.y;             // a - conj b
    const float2 w = twn[k];                                // exp(-2 pi i k / n); its conjugate is needed
    // i conj(w) D = i (w.x - i w.y)(dx + i dy) = i [(w.x dx + w.y dy) + i (w.x dy - w.y dx)] = -(w.x dy - w.y dx) + i (w.x dx + w.y dy)
    const float qx = -(w.x * dy - w.y * dx), qy = w.x * dx + w.y * dy;
    const float2 zk = make_float2(ex + qx, ey + qy);
    // Z'[m - k] = conj(E2) + i conj(w(m - k)) (-conj D) with conj(w(m - k)) = -w:  = conj(E2) - conj(i conj(w) D) ... = (ex - qx) + i (-ey + qy)
    const float2 zc = make_float2(ex - qx, -ey + qy);
    if (k == 0) {
        z[0] = zk;                                          // (X0 + Xm) + i (X0 - Xm) for real X0, Xm
        return;
    }
    z[k] = zk;
    if (kc != k) z[kc] = zc;
}
