// qh_pan.hip -- batched panadapter (include/quiskhip.h group 5): the spectrum-display path of Quisk,
// FFT ring producer in quisk_process_samples (quisk.c:2454-2475) + get_graph job 1 (quisk.c:5142-5331),
// for `nch` receivers at once and for EVERY completed block (the reference drops blocks when its 4-deep ring
// is full).
//
// fft_size N = R * M with M in {1024, 2048, 4096} (an LDS-resident FFT, qh_fft.hpp) and R in {1, 2, 4}:
//   pan_fft_kernel    one workgroup per (block, channel): for r = 0..R-1 the decimated sequence x[R*m + r]
//                     times the Hanning window (quisk.c:6008) goes through FFT-M; Y_r is parked in a scratch
//                     buffer (L2 resident).  The R strided passes re-read the block from L2, not from HBM.
//   pan_accum_kernel  one thread per (channel, bin k < M): X[k + M*q] = sum_r W_R^(r*q) * (W_N^(r*k) * Y_r[k]),
//                     then fft_avg[(bin + N/2) mod N] += |X| and the RMS S-meter sum over the passband bins
//                     (quisk.c:5218-5244), looping over the blocks so that no atomics touch fft_avg.
//   pan_graph_kernel  the refresh branch (quisk.c:5279-5327): n-bin box sums per pixel done IN PLACE on fft_avg
//                     in pixel order like the reference (a later pixel can see an earlier pixel's sum when
//                     zoomed in), dB scale, clamp to [-200, 0], S-meter in dB.  Tiny: one thread per channel.
#include <cmath>
#include <vector>
#include "qh_design.hpp"
#include "qh_internal.hpp"
#include "qh_fft.hpp"

namespace qh {

template <int M>
__global__ __launch_bounds__(NT) void pan_fft_kernel(const double2 *in, long long in_stride, long long blk_stride, int R,
                                                     const double *window, const double2 *tw, double2 *scratch,
                                                     long long scr_chan_stride)
{
    using C = double2;
    constexpr int E = M / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    C *lds = reinterpret_cast<C *>(smem);
    const int t = threadIdx.x, blk = blockIdx.x, ch = blockIdx.y;
    const C *x = in + (long long)ch * in_stride + (long long)blk * blk_stride;
    C *y = scratch + (long long)ch * scr_chan_stride + (long long)blk * R * M;
    const typename FftRR<M, false, C>::Tw twf = FftRR<M, false, C>::load(tw);
    for (int r = 0; r < R; r++) {
        C v[E];
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int n = R * (t + NT * i) + r;
            const double w = window[n];
            C s = x[n];
            v[i].x = s.x * w; v[i].y = s.y * w;
        }
        FftRR<M, false, C>::first(v, lds);
        FftRR<M, false, C>::rest(lds, v, twf);
#pragma unroll
        for (int i = 0; i < E; i++) y[(long long)r * M + t + NT * i] = v[i];
        __syncthreads();
    }
}

struct PanBand { int first, nwhole; double frac; int valid, pad; };     // S-meter passband in bins, quisk.c:5223-5244

__global__ __launch_bounds__(NT) void pan_accum_kernel(const double2 *scratch, long long scr_chan_stride, int nblk, int M,
                                                       int R, double *avg, double *meter, const PanBand *band)
{
    const int ch = blockIdx.y;
    const int k = blockIdx.x * NT + threadIdx.x;
    if (k >= M) return;
    const int N = M * R;
    const double2 *y = scratch + (long long)ch * scr_chan_stride;
    const PanBand pb = band[ch];
    // W_N^(r*k), r = 1..R-1
    double2 w[4];
    w[0] = make_double2(1.0, 0.0);
    for (int r = 1; r < R; r++) {
        double s, c;
        sincospi(-2.0 * (double)r * (double)k / (double)N, &s, &c);
        w[r] = make_double2(c, s);
    }
    double acc[4] = { 0, 0, 0, 0 }, m2 = 0.0;
    // weight of bin b in the S-meter sum
    double wt[4];
    for (int q = 0; q < R; q++) {
        const int b = k + M * q;
        const int sb = b >= N / 2 ? b - N : b;              // signed bin
        double v = 0.0;
        if (pb.valid) {
            if (sb >= pb.first && sb < pb.first + pb.nwhole) v = 1.0;
            else if (sb == pb.first + pb.nwhole) v = pb.frac;
        }
        wt[q] = v;
    }
    for (int blk = 0; blk < nblk; blk++) {
        const double2 *yb = y + (long long)blk * N;
        double2 z[4];
        for (int r = 0; r < R; r++) z[r] = cmul(yb[(long long)r * M + k], w[r]);
        double2 X[4];
        if (R == 1) {
            X[0] = z[0];
        } else if (R == 2) {
            X[0] = cadd(z[0], z[1]); X[1] = csub(z[0], z[1]);
        } else {
            const double2 a = cadd(z[0], z[2]), b = csub(z[0], z[2]), c = cadd(z[1], z[3]);
            const double2 d = mul_mi<false>(csub(z[1], z[3]));      // * (-i)
            X[0] = cadd(a, c); X[2] = csub(a, c); X[1] = cadd(b, d); X[3] = csub(b, d);
        }
        for (int q = 0; q < R; q++) {
            acc[q] += hypot(X[q].x, X[q].y);
            m2 += wt[q] * (X[q].x * X[q].x + X[q].y * X[q].y);
        }
    }
    for (int q = 0; q < R; q++) {
        const int b = k + M * q;
        avg[(long long)ch * N + ((b + N / 2) % N)] += acc[q];
    }
    if (m2 != 0.0) atomicAdd(meter + ch, m2);
}

__global__ void pan_graph_kernel(double *avg, double *meter, int nch, int N, int data_width, double rate, double zoom,
                                 double deltaf, int count, double *pixels, double *smeter)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    double *a = avg + (long long)ch * N;
    double *p = pixels + (long long)ch * data_width;
    const double scale = 20.0 * (log10((double)count) + log10((double)N) + 31.0 * log10(2.0));
    int n = (int)(zoom * (double)N / data_width + 0.5);
    if (n < 1) n = 1;
    for (int i = 0; i < data_width; i++) {
        int k = (int)(N * (deltaf / rate + zoom * ((double)i / data_width - 0.5) + 0.5) + 0.1);
        double d2 = 0.0;
        for (int j = 0; j < n; j++, k++)
            if (k >= 0 && k < N) d2 += a[k];
        a[i] = d2;
    }
    const double ss = 1.0 / 2147483647.0 / N;
    double sm = meter[ch] * ss * ss / count;
    sm = sm > 1E-16 ? 10.0 * log10(sm) : -160.0;
    smeter[ch] = sm + 4.25969;
    meter[ch] = 0.0;
    for (int i = 0; i < data_width; i++) {
        double d2 = 20.0 * log10(a[i]) - scale;
        if (d2 < -200) d2 = -200; else if (d2 > 0) d2 = 0;
        p[i] = d2;
    }
    for (int i = 0; i < N; i++) a[i] = 0.0;
}

// dst[ch][dst_off + i] = src[ch][src_off + i], i < n
__global__ void pan_copy_kernel(const double2 *src, long long src_stride, long long src_off, double2 *dst, long long dst_stride,
                                long long dst_off, int n)
{
    const int ch = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[(long long)ch * dst_stride + dst_off + i] = src[(long long)ch * src_stride + src_off + i];
}

struct Pan {
    int device = 0, nch = 0, N = 0, M = 0, R = 1, data_width = 0;
    double rate = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *window = nullptr, *avg = nullptr, *meter = nullptr, *pixels = nullptr, *smeter = nullptr;
    double2 *tw = nullptr, *carry = nullptr, *scratch = nullptr;
    PanBand *band = nullptr;
    std::vector<PanBand> hband;
    int fill = 0, count = 0, chunk_blocks = 0;

    ~Pan()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(window); (void)hipFree(avg); (void)hipFree(meter); (void)hipFree(pixels); (void)hipFree(smeter);
        (void)hipFree(tw); (void)hipFree(carry); (void)hipFree(scratch); (void)hipFree(band);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    int run_blocks(const double2 *src, long long src_stride, long long off, int nblk)
    {
        const int lds = (M + M / 16) * (int)sizeof(double2);
        for (int done = 0; done < nblk; done += chunk_blocks) {
            const int nb = nblk - done < chunk_blocks ? nblk - done : chunk_blocks;
            const double2 *p = src + off + (long long)done * N;
            dim3 g((unsigned)nb, (unsigned)nch);
            const long long scs = (long long)chunk_blocks * N;
            switch (M) {
            case 1024: hipLaunchKernelGGL(pan_fft_kernel<1024>, g, dim3(NT), lds, stream, p, src_stride, (long long)N, R, window, tw, scratch, scs); break;
            case 2048: hipLaunchKernelGGL(pan_fft_kernel<2048>, g, dim3(NT), lds, stream, p, src_stride, (long long)N, R, window, tw, scratch, scs); break;
            default:   hipLaunchKernelGGL(pan_fft_kernel<4096>, g, dim3(NT), lds, stream, p, src_stride, (long long)N, R, window, tw, scratch, scs); break;
            }
            hipLaunchKernelGGL(pan_accum_kernel, dim3((unsigned)((M + NT - 1) / NT), (unsigned)nch), dim3(NT), 0, stream, scratch,
                               scs, nb, M, R, avg, meter, band);
        }
        count += nblk;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
};

}  // namespace qh

using namespace qh;
struct qh_pan { Pan p; };

extern "C" {

qh_pan *qh_pan_create(int device, int nch, int fft_size, int data_width, double sample_rate, void *stream)
{
    int M = 0, R = 0;
    for (int r : { 1, 2, 4 })
        for (int m : { 4096, 2048, 1024 })
            if (!M && r * m == fft_size) { M = m; R = r; }
    if (nch <= 0 || data_width <= 0 || sample_rate <= 0 || !M) {
        set_error(M ? QH_ERR_INVALID : QH_ERR_UNSUPPORTED, "qh_pan_create: fft_size must be 1024 .. 16384, a power of two (got %d)", fft_size);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_pan *h = new qh_pan();
    Pan &p = h->p;
    p.device = device; p.nch = nch; p.N = fft_size; p.M = M; p.R = R; p.data_width = data_width; p.rate = sample_rate;
    p.stream = (hipStream_t)stream;
    auto fail = [&](const char *what) -> qh_pan * { set_error(QH_ERR_HIP, "qh_pan_create: %s failed", what); delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    if (!p.stream) { if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) return fail("stream"); p.own_stream = true; }
    // scratch for at most ~256 MiB of sub-FFT results per pass
    p.chunk_blocks = (int)((256ll << 20) / ((long long)nch * fft_size * 16));
    if (p.chunk_blocks < 1) p.chunk_blocks = 1;
    if (p.chunk_blocks > 64) p.chunk_blocks = 64;
    std::vector<double> win((size_t)fft_size);
    for (int i = 0, j = -fft_size / 2; i < fft_size; i++, j++)      // Hanning, quisk.c:6008
        win[(size_t)i] = 0.5 + 0.5 * std::cos(2. * M_PI * j / fft_size);
    std::vector<cd> tw = fft_twiddle_table(M);
    p.hband.assign((size_t)nch, PanBand{ 0, 0, 0.0, 0, 0 });
    if (hipMalloc((void **)&p.window, win.size() * 8) != hipSuccess || hipMalloc((void **)&p.tw, tw.size() * 16) != hipSuccess ||
        hipMalloc((void **)&p.avg, (size_t)nch * fft_size * 8) != hipSuccess || hipMalloc((void **)&p.meter, (size_t)nch * 8) != hipSuccess ||
        hipMalloc((void **)&p.pixels, (size_t)nch * data_width * 8) != hipSuccess || hipMalloc((void **)&p.smeter, (size_t)nch * 8) != hipSuccess ||
        hipMalloc((void **)&p.carry, (size_t)nch * fft_size * 16) != hipSuccess ||
        hipMalloc((void **)&p.scratch, (size_t)nch * p.chunk_blocks * fft_size * 16) != hipSuccess ||
        hipMalloc((void **)&p.band, (size_t)nch * sizeof(PanBand)) != hipSuccess)
        return fail("hipMalloc");
    if (hipMemcpy(p.window, win.data(), win.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p.tw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p.band, p.hband.data(), (size_t)nch * sizeof(PanBand), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(p.avg, 0, (size_t)nch * fft_size * 8) != hipSuccess || hipMemset(p.meter, 0, (size_t)nch * 8) != hipSuccess)
        return fail("initial copies");
    const int lds = (M + M / 16) * (int)sizeof(double2);
    hipError_t e = hipSuccess;
    switch (M) {
    case 1024: e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pan_fft_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); break;
    case 2048: e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pan_fft_kernel<2048>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); break;
    default:   e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pan_fft_kernel<4096>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); break;
    }
    if (e != hipSuccess) return fail("hipFuncSetAttribute");
    return h;
}

void qh_pan_destroy(qh_pan *h) { delete h; }

// The S-meter passband: first bin from (rx_tune_freq + filter_start_offset), width filter_bandwidth (quisk.c:5223-5229)
int qh_pan_set_smeter_band(qh_pan *h, int ch, double f_start, double bandwidth)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    Pan &p = h->p;
    if (ch < -1 || ch >= p.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    QH_HIP(hipSetDevice(p.device));
    const double d2 = bandwidth * p.N / p.rate;
    const int i = (int)(f_start * p.N / p.rate + 0.5);
    const int n = (int)(std::floor(d2) + 0.01);
    PanBand b{ i, n, d2 - n, (i > -p.N / 2 && i + n + 1 < p.N / 2) ? 1 : 0, 0 };
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? p.nch : ch + 1); c++) p.hband[(size_t)c] = b;
    QH_HIP(hipMemcpyAsync(p.band, p.hband.data(), (size_t)p.nch * sizeof(PanBand), hipMemcpyHostToDevice, p.stream));
    QH_HIP(hipStreamSynchronize(p.stream));
    return QH_OK;
}

int qh_pan_feed(qh_pan *h, const double *d_in, long long in_stride, int n)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    if (n <= 0) return QH_OK;
    if (!d_in || in_stride < n) return set_error(QH_ERR_INVALID, "bad input");
    Pan &p = h->p;
    QH_HIP(hipSetDevice(p.device));
    const double2 *in = reinterpret_cast<const double2 *>(d_in);
    long long pos = 0;
    auto copy = [&](const double2 *src, long long ss, long long so, double2 *dst, long long ds, long long dofs, int cnt) {
        int gx = (cnt + 255) / 256; if (gx > 64) gx = 64;
        hipLaunchKernelGGL(pan_copy_kernel, dim3((unsigned)gx, (unsigned)p.nch), dim3(256), 0, p.stream, src, ss, so, dst, ds, dofs, cnt);
    };
    if (p.fill > 0) {
        const int need = p.N - p.fill;
        const int take = n < need ? n : need;
        copy(in, in_stride, 0, p.carry, p.N, p.fill, take);
        p.fill += take; pos = take;
        if (p.fill == p.N) {
            if (int rc = p.run_blocks(p.carry, p.N, 0, 1)) return rc;
            p.fill = 0;
        }
    }
    const int whole = (int)((n - pos) / p.N);
    if (whole > 0) {
        if (int rc = p.run_blocks(in, in_stride, pos, whole)) return rc;
        pos += (long long)whole * p.N;
    }
    const int rest = (int)(n - pos);
    if (rest > 0) { copy(in, in_stride, pos, p.carry, p.N, 0, rest); p.fill = rest; }
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_pan_count(const qh_pan *h) { return h ? h->p.count : 0; }

int qh_pan_graph(qh_pan *h, double zoom, double deltaf, double *h_pixels, double *h_smeter, int *count)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    Pan &p = h->p;
    if (count) *count = p.count;
    if (p.count <= 0) return QH_OK;                             // get_graph returns None until an FFT has run
    QH_HIP(hipSetDevice(p.device));
    hipLaunchKernelGGL(pan_graph_kernel, dim3((unsigned)((p.nch + 63) / 64)), dim3(64), 0, p.stream, p.avg, p.meter, p.nch, p.N,
                       p.data_width, p.rate, zoom, deltaf, p.count, p.pixels, p.smeter);
    if (h_pixels) QH_HIP(hipMemcpyAsync(h_pixels, p.pixels, (size_t)p.nch * p.data_width * 8, hipMemcpyDeviceToHost, p.stream));
    if (h_smeter) QH_HIP(hipMemcpyAsync(h_smeter, p.smeter, (size_t)p.nch * 8, hipMemcpyDeviceToHost, p.stream));
    QH_HIP(hipStreamSynchronize(p.stream));
    p.count = 0;
    return QH_OK;
}

int qh_pan_feed_host(qh_pan *h, const double *h_in, long long in_stride, int n)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    if (n <= 0) return QH_OK;
    Pan &p = h->p;
    QH_HIP(hipSetDevice(p.device));
    double2 *d = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)p.nch * n * 16));
    hipError_t e = hipMemcpy2DAsync(d, (size_t)n * 16, h_in, (size_t)in_stride * 16, (size_t)n * 16, (size_t)p.nch, hipMemcpyHostToDevice, p.stream);
    int rc = e == hipSuccess ? qh_pan_feed(h, reinterpret_cast<const double *>(d), n, n) : QH_ERR_HIP;
    hipError_t e2 = hipStreamSynchronize(p.stream);
    (void)hipFree(d);
    if (rc) return rc == QH_ERR_HIP ? set_error(QH_ERR_HIP, "qh_pan_feed_host: copy failed") : rc;
    if (e2 != hipSuccess) return set_error(QH_ERR_HIP, "qh_pan_feed_host: synchronize failed");
    return QH_OK;
}

}  // extern "C"
