// qh_snba.hpp -- WDSP's SNBA (spectral noise blanker, wdsp/snb.c:31-577 with asolve / median / trI / dR of wdsp/lmath.c:29-186
// and the two polyphase resamplers of wdsp/resample.c:120-157) on gfx950.
//
// One wavefront per channel: the block is a chain of short, strictly ordered recurrences (an order-64 Levinson solve per
// 256-sample frame, a median, a run scanner, then per corrupt run another solve, a Toeplitz inverse and a banded
// least-squares interpolation), so the parallelism is the channels; inside a channel the 64 lanes take one output each of
// every sum that has independent outputs (autocorrelation lags, residue samples, polyphase outputs, matrix rows), each lane
// adding its terms in the reference's order, and the scalar recurrences run redundantly in all lanes from LDS broadcasts.
// Every sum therefore has the reference's operation order (fp contraction off), which keeps the threshold decisions of the
// detector identical.  The structured matrices A1 / A2 of xHat (snb.c:265-304) are never built: their entries are read off
// the predictor `a` while P1 = A1'A2 is formed; the Toeplitz inverse (up to xsize x xsize) lives in a per-channel scratch.
#pragma once
#include <hip/hip_runtime.h>
#include "qh_wave.hpp"

namespace qh {

constexpr int kSnbX = 256;          // xsize, RXA.c:245
constexpr int kSnbMaxImp = 256;     // MAXIMP, snb.c:29
constexpr int kSnbMaxDsp = 1024;    // dsp_size limit of this kernel
constexpr int kSnbMaxCppIn = 561;   // 140 * (dsp_rate / 12000) + 1 with dsp_rate <= 48000

struct SnbaParam {
    int ratio, isize, incr, iasize, oasize, init_oaoutidx, cpp_in, cpp_out;     // calc_snba, snb.c:31-66
    int asize, npasses, b, pre, post, pad;
    double k1, k2, pmultmin;
    // state layout (doubles per channel)
    int off_inacc, off_outacc, off_rin, off_rout, state_doubles, pad2;
};
struct SnbaIdx { int iainidx, iaoutidx, nsamps, oainidx, oaoutidx, pad[3]; };
// what the tuning setters change, per channel (SetRXASNBAasize / npasses / k1 / k2 / bridge / presamps / postsamps / pmultmin,
// snb.c:604-658); the defaults are create_rxa's (RXA.c:183-202)
struct SnbaTune { int asize, npasses, b, pre, post, pad; double k1, k2, pmultmin; };

struct SnbaLds {
    double xb[2 * kSnbX];           // xbase | xaux
    double savex[kSnbX], v[kSnbX], vpwr[kSnbX], a[kSnbX], r[kSnbX + 1], z[kSnbX + 1], p2[kSnbX], xout[kSnbX], y[kSnbX], vv[kSnbX];
    double merit[kSnbMaxImp];
    double sin_[kSnbMaxCppIn - 1 + kSnbMaxDsp];     // input resampler: history | block (real parts)
    double sob[140 + kSnbMaxDsp];                   // output resampler: history | 12 kHz block
    double scal[4];
    int det[kSnbX], unfixed[kSnbX], bimp[kSnbMaxImp], limp[kSnbMaxImp], bef[kSnbMaxImp + 1], aft[kSnbMaxImp], popt[kSnbMaxImp],
        nextl[kSnbMaxImp];
    int nimp, next;
};

// asolve, lmath.c:96-127: x has at least asize valid samples in front of it
static __device__ void snba_asolve(SnbaLds &s, int xsize, int asize, const double *x, int lane)
{
#pragma clang fp contract(off)
    {   // lags lane and lane + 64 side by side (asize <= 64: the second one is lag 64 in lane 0); each sum in the order j = 0 .. xsize - 1
        double acc = 0.0, acc2 = 0.0;
        const double *xa = x - lane, *xb2 = x - lane - 64;
#pragma unroll 8
        for (int j = 0; j < xsize; j++) {
            const double xj = x[j];
            acc += xj * xa[j];
            acc2 += xj * xb2[j];
        }
        if (lane <= asize) { s.r[lane] = acc; s.z[lane] = lane == 0 ? 1.0 : 0.0; }
        if (lane + 64 <= asize) { s.r[lane + 64] = acc2; s.z[lane + 64] = 0.0; }
    }
    __syncthreads();
    double beta = s.r[0];
    for (int k = 0; k < asize; k++) {
        // the products in parallel, their sum in the reference's order j = 0 .. k (one readlane + one subtract per term)
        const double prod = lane <= k ? s.z[lane] * s.r[k + 1 - lane] : 0.0;
        double alpha = 0.0;
        for (int j = 0; j <= k; j++) alpha -= lane_bcast(prod, j);
        alpha /= beta;
        const int half = (k + 1) / 2;
        double zi = 0.0, zo = 0.0;
        if (lane <= half) { zi = s.z[lane]; zo = s.z[k + 1 - lane]; }
        __syncthreads();
        if (lane <= half) {
            const double t = zo + alpha * zi;
            s.z[lane] = zi + alpha * zo;
            s.z[k + 1 - lane] = t;
        }
        __syncthreads();
        beta *= 1.0 - alpha * alpha;
    }
    for (int i = lane; i < asize; i += 64) {
        double t = -s.z[i + 1];
        if (t != t) t = 0.0;
        s.a[i] = t;
    }
    __syncthreads();
}

// invf, snb.c:306-322
static __device__ void snba_invf(SnbaLds &s, int xsize, int asize, const double *x, int lane)
{
#pragma clang fp contract(off)
    for (int i = lane; i < xsize; i += 64) {
        double acc = 0.0;
        if (i >= asize && i < xsize - asize) {
            for (int j = 0; j < asize; j++) acc += s.a[j] * (x[i - 1 - j] + x[i + 1 + j]);
            acc = x[i] - 0.5 * acc;
        } else if (i >= xsize - asize) {
            for (int j = 0; j < asize; j++) acc += s.a[j] * x[i - 1 - j];
            acc = x[i] - acc;
        }
        s.v[i] = acc;
    }
    __syncthreads();
}

// det, snb.c:324-402
static __device__ void snba_det(SnbaLds &s, const SnbaParam &q, int asize, int lane)
{
#pragma clang fp contract(off)
    const int xs = kSnbX, n = xs - asize, kth = n / 2;
    for (int i = lane; i < xs; i += 64) s.vpwr[i] = i >= asize ? s.v[i] * s.v[i] : 0.0;
    __syncthreads();
    // median (lmath.c:129-186 selects the element of rank n / 2): rank by counting
    for (int i = asize + lane; i < xs; i += 64) {
        const double e = s.vpwr[i];
        int rank = 0;
        for (int j = asize; j < xs; j++) {
            const double o = s.vpwr[j];
            rank += (o < e || (o == e && j < i)) ? 1 : 0;
        }
        if (rank == kth) s.scal[0] = e;
    }
    __syncthreads();
    const double t1 = q.k1 * s.scal[0];
    double t2 = 0.0;
    for (int base = asize; base < xs; base += 64) {        // contributions in parallel (0.0 where the reference adds nothing), summed in order
        const int i = base + lane;
        const double p = i < xs ? s.vpwr[i] : 0.0;
        double c = 0.0;
        if (i < xs) { if (p <= t1) c = p; else if (p <= 2.0 * t1) c = 2.0 * t1 - p; }
        const int cnt = xs - base < 64 ? xs - base : 64;
        for (int j = 0; j < cnt; j++) t2 += lane_bcast(c, j);
    }
    t2 *= q.k2 / (double)n;
    for (int i = lane; i < xs; i += 64) s.det[i] = (i >= asize && s.vpwr[i] > t2) ? 1 : 0;
    __syncthreads();
    if (lane == 0) {
        int bstate = 0, bcount = 0, bsamp = 0;
        for (int i = asize; i < xs; i++) {
            const int d = s.det[i];
            if (bstate == 0) { if (d == 1) bstate = 1; }
            else if (bstate == 1) { if (d == 0) { bstate = 2; bsamp = i; bcount = 1; } }
            else {
                ++bcount;
                if (bcount > q.b) bstate = d == 1 ? 1 : 0;
                else if (d == 1) {
                    for (int j = bsamp; j < bsamp + bcount - 1; j++) s.det[j] = 1;
                    bstate = 1;
                }
            }
        }
        for (int i = asize; i < xs; i++)
            if (s.det[i] == 1)
                for (int j = i - 1; j > i - 1 - q.pre; j--) if (j >= asize) s.det[j] = 1;
        for (int i = xs - 1; i >= asize; i--)
            if (s.det[i] == 1)
                for (int j = i + 1; j < i + 1 + q.post; j++) if (j < xs) s.det[j] = 1;
    }
    __syncthreads();
}

// scanFrame, snb.c:404-490 (lane 0; results in LDS)
static __device__ void snba_scan(SnbaLds &s, int xsize, int pval, double pmultmin, const int *det, int lane)
{
#pragma clang fp contract(off)
    if (lane == 0) {
        int inflag = 0, nimp = 0;
        for (int i = 0; i <= kSnbMaxImp; i++) s.bef[i] = 0;
        for (int i = 0; i < kSnbMaxImp; i++) s.aft[i] = 0;
        for (int i = 0; i < xsize && nimp < kSnbMaxImp; i++) {
            const int d = det[i];
            if (d == 1 && inflag == 0) { inflag = 1; s.bimp[nimp] = i; s.limp[nimp] = 1; nimp++; }
            else if (d == 1) s.limp[nimp - 1]++;
            else {
                inflag = 0;
                s.bef[nimp]++;
                if (nimp > 0) s.aft[nimp - 1]++;
            }
        }
        for (int i = 0; i < nimp; i++) {
            int p = s.bef[i] < s.aft[i] ? s.bef[i] : s.aft[i];
            if (p > pval) p = pval;
            if (p < (int)(pmultmin * s.limp[i])) p = -1;
            s.popt[i] = p;
            s.merit[i] = (double)p / (double)s.limp[i];
            s.nextl[i] = i;
        }
        for (int j = 0; j < nimp - 1; j++)
            for (int k = 0; k < nimp - j - 1; k++)
                if (s.merit[k] < s.merit[k + 1]) {
                    const double td = s.merit[k]; const int ti = s.nextl[k];
                    s.merit[k] = s.merit[k + 1]; s.nextl[k] = s.nextl[k + 1];
                    s.merit[k + 1] = td; s.nextl[k + 1] = ti;
                }
        int i = 1;
        if (nimp > 0) while (i < nimp && s.merit[i] == s.merit[0]) i++;
        for (int j = 0; j < i - 1; j++)
            for (int k = 0; k < i - j - 1; k++)
                if (s.limp[s.nextl[k]] < s.limp[s.nextl[k + 1]]) {
                    const double td = s.merit[k]; const int ti = s.nextl[k];
                    s.merit[k] = s.merit[k + 1]; s.nextl[k] = s.nextl[k + 1];
                    s.merit[k + 1] = td; s.nextl[k + 1] = ti;
                }
        s.nimp = nimp;
        s.next = nimp > 0 ? s.nextl[0] : 0;
    }
    __syncthreads();
}

// xHat, snb.c:265-304: the n = xusize unknown samples from p known ones either side (xk = first known sample)
static __device__ void snba_xhat(SnbaLds &s, int n, int p, const double *xk, double *B, int lane)
{
#pragma clang fp contract(off)
    const double *a = s.a;
    // ATAc0 (snb.c:209-216): r[i] = sum_j A1[j][i] A1[j][0]; column 0 of A1 is 1, -a[0], ..., -a[p-1]
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        for (int j = 0; j <= p; j++) {
            const double c0 = j == 0 ? 1.0 : -a[j - 1];
            double ci = 0.0;
            if (j == i) ci = 1.0;
            else if (j > i && j <= i + p) ci = -a[j - i - 1];
            acc += ci * c0;
        }
        s.r[i] = acc;
    }
    __syncthreads();
    // trI (lmath.c:52-94)
    const double scale = 1.0 / s.r[0];
    __syncthreads();
    for (int i = lane; i < n; i += 64) s.r[i] *= scale;
    for (int i = lane; i < n; i += 64) { s.y[i] = 0.0; s.vv[i] = 0.0; }
    __syncthreads();
    {   // dR (lmath.c:29-50) on n - 1 unknowns
        const int m = n - 1;
        if (lane == 0) s.y[0] = -s.r[1];
        __syncthreads();
        double alpha = -s.r[1], beta = 1.0;
        for (int k = 0; k < m - 1; k++) {
            beta *= 1.0 - alpha * alpha;
            double gamma = 0.0;
            for (int j = 0; j <= k; j++) gamma += s.r[k + 1 - j] * s.y[j];
            alpha = -(s.r[k + 2] + gamma) / beta;
            __syncthreads();
            for (int i = lane; i <= k; i += 64) s.z[i] = s.y[i] + alpha * s.y[k - i];
            __syncthreads();
            for (int i = lane; i <= k; i += 64) s.y[i] = s.z[i];
            if (lane == 0) s.y[k + 1] = alpha;
            __syncthreads();
        }
    }
    double t = 0.0;
    for (int i = 0; i < n - 1; i++) t += s.r[i + 1] * s.y[i];
    const double gamma = 1.0 / (1.0 + t);
    for (int i = lane; i < n - 1; i += 64) s.vv[i] = gamma * s.y[n - 2 - i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) B[i] = i == 0 ? gamma : s.vv[n - 1 - i];
    __syncthreads();
    for (int i = 1; i <= (n - 1) / 2; i++) {
        for (int j = i + lane; j < n - i; j += 64)
            B[i * n + j] = B[(i - 1) * n + (j - 1)] + (s.vv[n - j - 1] * s.vv[n - i - 1] - s.vv[i - 1] * s.vv[j - 1]) / gamma;
        __syncthreads();
    }
    for (int i = 0; i <= (n - 1) / 2; i++)
        for (int j = i + lane; j < n - i; j += 64) {
            const double b = B[i * n + j] * scale;
            const int ni = n - i - 1, nj = n - j - 1;
            B[i * n + j] = b; B[j * n + i] = b; B[ni * n + nj] = b; B[nj * n + ni] = b;
        }
    // P2 = (A1'A2 restricted to the known columns) xk: multA1TA2 + multXKE, snb.c:218-252
    const int q2 = n + 2 * p;
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        for (int col = i; col < p; col++) {                 // known samples in front: A2[k][col] = a[p-1-col+k] for k <= col
            double e = 0.0;
            const int kmax = i + p < col ? i + p : col;
            for (int k = i; k <= kmax; k++) {
                const double a1 = k == i ? 1.0 : -a[k - i - 1];
                e += a1 * a[p - 1 - col + k];
            }
            acc += e * xk[col];
        }
        for (int col = q2 - p; col <= q2 - n + i; col++) {  // known samples behind: A2[c][col] = -1, A2[k][col] = a[k-c-1] below it, c = col - p
            double e = 0.0;
            const int c = col - p;
            for (int k = (i > c ? i : c); k <= i + p; k++) {
                const double a1 = k == i ? 1.0 : -a[k - i - 1];
                const double a2 = k == c ? -1.0 : a[k - c - 1];
                e += a1 * a2;
            }
            acc += e * xk[col];
        }
        s.p2[i] = acc;
    }
    __syncthreads();
    // multAv, snb.c:254-263 (B is symmetric: read the transposed element for coalescing)
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        for (int k = 0; k < n; k++) acc += B[k * n + i] * s.p2[k];
        s.xout[i] = acc;
    }
    __syncthreads();
}

// execFrame, snb.c:492-537; x = s.xb + xsize
static __device__ void snba_frame(SnbaLds &s, const SnbaParam &q, double *B, int lane)
{
#pragma clang fp contract(off)
    const int xs = kSnbX;
    double *x = s.xb + xs;
    for (int i = lane; i < xs; i += 64) s.savex[i] = x[i];
    __syncthreads();
    snba_asolve(s, xs, q.asize, x, lane);
    snba_invf(s, xs, q.asize, x, lane);
    snba_det(s, q, q.asize, lane);
    for (int i = lane; i < xs; i += 64) if (s.det[i] != 0) x[i] = 0.0;
    __syncthreads();
    snba_scan(s, xs, q.asize, q.pmultmin, s.det, lane);
    const int nimp = s.nimp;
    for (int pass = 0; pass < q.npasses; pass++) {
        for (int i = lane; i < xs; i += 64) s.unfixed[i] = s.det[i];
        __syncthreads();
        for (int k = 0; k < nimp; k++) {
            if (k > 0) snba_scan(s, xs, q.asize, q.pmultmin, s.unfixed, lane);
            const int next = s.next, p = s.popt[next], b0 = s.bimp[next], len = s.limp[next];
            __syncthreads();
            if (p > 0) {
                snba_asolve(s, xs, p, x, lane);
                snba_xhat(s, len, p, x + b0 - p, B, lane);
                for (int i = lane; i < len; i += 64) { x[b0 + i] = s.xout[i]; s.unfixed[b0 + i] = 0; }
            } else
                for (int i = lane; i < len; i += 64) x[b0 + i] = s.savex[b0 + i];
            __syncthreads();
        }
    }
}

// rows of the listed channels from one chain buffer to the other
static __global__ __launch_bounds__(NT) void copy_rows_kernel(const double2 *src, double2 *dst, long long stride, int n, const int *chan_list)
{
    const long long o = (long long)chan_list[blockIdx.y] * stride;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) dst[o + i] = src[o + i];
}

// xsnba (snb.c:539-571) over nblk blocks of dsp_size samples, in place on the channel's row of buf
static __global__ __launch_bounds__(64) void snba_kernel(double2 *buf, long long stride, int nblk, int dsp_size, const int *chan_list,
                                                         SnbaParam q, const double *h_in, const double *h_out, double *state,
                                                         SnbaIdx *idx, double *scratch, const SnbaTune *tune)
{
#pragma clang fp contract(off)
    __shared__ SnbaLds s;
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    {
        const SnbaTune t = tune[ch];
        q.asize = t.asize; q.npasses = t.npasses; q.b = t.b; q.pre = t.pre; q.post = t.post; q.k1 = t.k1; q.k2 = t.k2; q.pmultmin = t.pmultmin;
    }
    double2 *row = buf + (long long)ch * stride;
    double *st = state + (size_t)ch * q.state_doubles;
    double *B = scratch + (size_t)ch * kSnbX * kSnbX;
    const double *ho = h_out + (size_t)ch * q.ratio * q.cpp_out;
    const int hin = q.cpp_in - 1, hout = q.cpp_out - 1;
    SnbaIdx ix = idx[ch];
    for (int i = lane; i < 2 * kSnbX; i += 64) s.xb[i] = st[i];
    for (int i = lane; i < hin; i += 64) s.sin_[i] = st[q.off_rin + i];
    for (int i = lane; i < hout; i += 64) s.sob[i] = st[q.off_rout + i];
    __syncthreads();
    double *inacc = st + q.off_inacc, *outacc = st + q.off_outacc;
    for (int blk = 0; blk < nblk; blk++) {
        double2 *p = row + (long long)blk * dsp_size;
        // xresample (inresamp): L = 1, M = ratio; out[m] = sum_j h[j] x[m * ratio - j]
        for (int i = lane; i < dsp_size; i += 64) s.sin_[hin + i] = p[i].x;
        __syncthreads();
        for (int m = lane; m < q.isize; m += 64) {
            double acc;
            if (q.ratio == 1) acc = s.sin_[hin + m];
            else {
                acc = 0.0;
                const double *xp = s.sin_ + hin + m * q.ratio;
                for (int j = 0; j < q.cpp_in; j++) acc += h_in[j] * xp[-j];
            }
            inacc[(ix.iainidx + m) % q.iasize] = acc;
        }
        __syncthreads();
        for (int base = 0; base < hin; base += 64) {         // slide the input history down by one block
            const double keep = base + lane < hin ? s.sin_[dsp_size + base + lane] : 0.0;
            __syncthreads();
            if (base + lane < hin) s.sin_[base + lane] = keep;
            __syncthreads();
        }
        ix.iainidx = (ix.iainidx + q.isize) % q.iasize;
        ix.nsamps += q.isize;
        __syncthreads();
        while (ix.nsamps >= q.incr) {
            for (int i = lane; i < q.incr; i += 64) s.xb[2 * kSnbX - q.incr + i] = inacc[ix.iaoutidx + i];
            __syncthreads();
            snba_frame(s, q, B, lane);
            ix.iaoutidx = (ix.iaoutidx + q.incr) % q.iasize;
            ix.nsamps -= q.incr;
            for (int i = lane; i < q.incr; i += 64) outacc[ix.oainidx + i] = s.xb[kSnbX + i];
            ix.oainidx = (ix.oainidx + q.incr) % q.oasize;
            __syncthreads();
            for (int base = 0; base < 2 * kSnbX - q.incr; base += 64) {      // memmove(xbase, xbase + incr, ...), snb.c:556
                const double keep = base + lane < 2 * kSnbX - q.incr ? s.xb[q.incr + base + lane] : 0.0;
                __syncthreads();
                if (base + lane < 2 * kSnbX - q.incr) s.xb[base + lane] = keep;
                __syncthreads();
            }
        }
        for (int i = lane; i < q.isize; i += 64) s.sob[hout + i] = outacc[(ix.oaoutidx + i) % q.oasize];
        ix.oaoutidx = (ix.oaoutidx + q.isize) % q.oasize;
        __syncthreads();
        // xresample (outresamp): L = ratio, M = 1; out[i * ratio + ph] = sum_j h[ph][j] o[i - j]
        for (int n = lane; n < dsp_size; n += 64) {
            double acc;
            if (q.ratio == 1) acc = s.sob[hout + n];
            else {
                const int i = n / q.ratio, ph = n - i * q.ratio;
                const double *hp = ho + ph * q.cpp_out, *op = s.sob + hout + i;
                acc = 0.0;
                for (int j = 0; j < q.cpp_out; j++) acc += hp[j] * op[-j];
            }
            p[n] = make_double2(acc, 0.0);
        }
        __syncthreads();
        for (int base = 0; base < hout; base += 64) {
            const double keep = base + lane < hout ? s.sob[q.isize + base + lane] : 0.0;
            __syncthreads();
            if (base + lane < hout) s.sob[base + lane] = keep;
            __syncthreads();
        }
    }
    for (int i = lane; i < 2 * kSnbX; i += 64) st[i] = s.xb[i];
    for (int i = lane; i < hin; i += 64) st[q.off_rin + i] = s.sin_[i];
    for (int i = lane; i < hout; i += 64) st[q.off_rout + i] = s.sob[i];
    if (lane == 0) idx[ch] = ix;
}

}  // namespace qh
