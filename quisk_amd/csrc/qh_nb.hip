// qh_nb.hip -- Quisk's noise blanker (include/quiskhip.h group 10): NoiseBlanker, quisk.c:680-784, for `nch` streams
// at the receiver's input rate (quisk_process_samples calls it before the panadapter ring
// and the tune, quisk.c:2448-2449).
//
// The reference steps a delay line of S = 3 * hw samples (hw = 500 us) one sample at a time: sample j is a pulse,
// p[j], when |x[j]| > limit * mean(|x| over the last S samples); a state machine then tapers the hw samples before
// the first pulse of a run to zero, zeroes samples while pulses last, and ramps the gain back over hw samples.
// Unrolled, the state is s[j] = p[j-1] and the gain of sample m is a function of the p's in (m - hw, m + hw) alone:
//     d = m - (last j <= m with p[j])        d <= 1: 0        2 <= d <= hw: (d - 1) / hw        else 1
//     times (t - m) / hw for every run start t (p[t] && !p[t-1]) with 1 <= t - m <= hw - 1, in time order
// (the order the reference multiplies in).  That makes the blanker a sliding-window function of the input, so it
// tiles over time as well as over channels: one workgroup produces L outputs from L + 5 hw - 1 inputs, with the |x|
// prefix sum, the p bits and the run-start bits of the tile in LDS.  State between calls is the last 7 hw raw input
// samples per stream (ping-pong), nothing else; a tile without a single pulse -- the normal case -- copies.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "qh_internal.hpp"
#include "qh_kernels.hpp"

using namespace qh;

namespace {

struct NbArgs {
    const double2 *in;      // [nch][in_stride]   new samples of this call
    const double2 *hist;    // [nch][H]           the H samples before in[0]
    double2 *out;           // [nch][out_stride]  out[i] = g * x[i - S]
    long long in_stride, out_stride;
    int n, H, hw, S, ntiles;
    double limit, limit_before;     // limit / S in force from call-relative sample g_change on / before it
    long long g_change;
};

__device__ __forceinline__ double2 nb_fetch(const NbArgs &a, int ch, long long g)
{
    if (g >= a.n) return make_double2(0.0, 0.0);
    return g >= 0 ? a.in[(long long)ch * a.in_stride + g] : a.hist[(long long)ch * a.H + (a.H + g)];
}

// |z| as cabs gives it for every magnitude a receiver sample can have; hypot's scaling only for the far ends
__device__ __forceinline__ double nb_mag(double2 z)
{
    const double m2 = __builtin_fma(z.x, z.x, z.y * z.y);
    if (__builtin_expect(!(m2 > 1e-280 && m2 < 1e280), 0)) return hypot(z.x, z.y);
    return __builtin_sqrt(m2);
}

// dynamic LDS: double mag[W], double E[W + 1], u64 pw[W/64 + 2], u64 ow[W/64 + 2]   (W = L + 5 hw - 1)
// One workgroup = L outputs of one stream; the L samples behind them stay in registers from the first load.
template <int L>
__global__ __launch_bounds__(NT) void nb_kernel(NbArgs a)
{
    constexpr int R = L / NT;
    extern __shared__ double nb_lds[];
    const int W = L + 5 * a.hw - 1, nw = (W >> 6) + 2;
    double *mag = nb_lds, *E = nb_lds + W;
    unsigned long long *pw = reinterpret_cast<unsigned long long *>(E + W + 1), *ow = pw + nw;
    __shared__ double wsum[NT / 64];
    int ch, tile;
    xcd_tile_map(a.ntiles, ch, tile);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = tile * L;                                    // first output of the tile
    const long long B = (long long)i0 - 2 * a.S - a.hw + 1;     // call-relative time of local index 0
    const int nout = a.n - i0 < L ? a.n - i0 : L;
    const int kout = a.S + a.hw - 1;                            // local index of the sample behind output i0

    // |x| of the tile and its look-back / look-ahead, coalesced; tiles inside the call's samples skip the seam logic
    double2 z[R];
    if (B >= 0 && B + W <= a.n) {
        const double2 *src = a.in + (long long)ch * a.in_stride + B;
#pragma unroll
        for (int j = 0; j < R; j++) z[j] = src[kout + t + j * NT];
#pragma unroll
        for (int j = 0; j < R; j++) mag[kout + t + j * NT] = nb_mag(z[j]);
        for (int k = t; k < kout; k += NT) mag[k] = nb_mag(src[k]);
        for (int k = kout + L + t; k < W; k += NT) mag[k] = nb_mag(src[k]);
    } else {
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int k = kout + t + j * NT;
            z[j] = nb_fetch(a, ch, B + k);
            mag[k] = nb_mag(z[j]);
        }
        for (int k = t; k < kout; k += NT) mag[k] = nb_mag(nb_fetch(a, ch, B + k));
        for (int k = kout + L + t; k < W; k += NT) mag[k] = nb_mag(nb_fetch(a, ch, B + k));
    }
    __syncthreads();
    // exclusive prefix sum E[k] = mag[0] + .. + mag[k-1]: a contiguous chunk per thread, then across threads
    const int C = ((W + NT - 1) / NT) | 1;                      // odd: chunk starts spread over the LDS banks
    const int k0 = t * C, k1 = k0 + C < W ? k0 + C : W;
    double part = 0.0;
    for (int k = k0; k < k1; k++) part += mag[k];
    double inc = part;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    double base = inc - part;
    for (int w = 0; w < wave; w++) base += wsum[w];
    for (int k = k0; k < k1; k++) { E[k] = base; base += mag[k]; }
    if (k1 == W && k0 < W) E[W] = base;
    __syncthreads();
    // pulse bits: is_pulse = !(mag <= save_sum / save_size * limit), quisk.c:744-747 (limit / save_size folded into one
    // factor); 64 consecutive samples per wave
    int any = 0;
    for (int kb = 0; kb < nw * 64; kb += NT) {
        const int k = kb + t;
        bool p = false;
        if (k < W && k >= a.S - 1) {
            const double sum = E[k + 1] - E[k + 1 - a.S];
            const double lim = B + k >= a.g_change ? a.limit : a.limit_before;
            p = !(mag[k] <= sum * lim);
        }
        const unsigned long long bits = __ballot(p);
        if (lane == 0 && (k >> 6) < nw) pw[k >> 6] = bits;
        any |= bits != 0;
    }
    any = __syncthreads_or(any);
    double2 *o = a.out + (long long)ch * a.out_stride + i0;
    if (!any) {
#pragma unroll
        for (int j = 0; j < R; j++) if (t + j * NT < nout) o[t + j * NT] = z[j];
        return;
    }
    // run starts: p[k] && !p[k-1]
    for (int w = t; w < nw; w += NT) {
        const unsigned long long cur = pw[w], prev = w ? pw[w - 1] : 0ull;
        ow[w] = cur & ~((cur << 1) | (prev >> 63));
    }
    __syncthreads();
    const double hwd = (double)a.hw;
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int i = t + j * NT;
        if (i >= nout) break;
        const int k = kout + i;
        double2 v = z[j];
        // distance back to the last pulse, looking hw samples back
        int w = k >> 6;
        unsigned long long bits = pw[w] & (~0ull >> (63 - (k & 63)));
        const int wlo = (k - a.hw) >> 6;
        while (!bits && w > wlo) bits = pw[--w];
        int d = a.hw + 1;
        if (bits) d = k - ((w << 6) + 63 - __clzll((long long)bits));
        if (d <= 1) {
            v = make_double2(0.0, 0.0);                         // the pulse itself and the sample after it
        } else {
            if (d <= a.hw) { const double f = (double)(d - 1) / hwd; v.x *= f; v.y *= f; }     // ramp up, quisk.c:758-762
            // tapers of the runs that start within the next hw - 1 samples, quisk.c:751-756
            const int kend = k + a.hw - 1;
            int wo = (k + 1) >> 6;
            unsigned long long ob = ow[wo] & (~0ull << ((k + 1) & 63));
            const int whi = kend >> 6;
            for (;;) {
                while (ob) {
                    const int tpos = (wo << 6) + __ffsll((long long)ob) - 1;
                    ob &= ob - 1;
                    if (tpos > kend) { wo = whi; ob = 0; break; }
                    const double f = (double)(tpos - k) / hwd;
                    v.x *= f; v.y *= f;
                }
                if (wo >= whi) break;
                ob = ow[++wo];
            }
        }
        o[i] = v;
    }
}

__global__ __launch_bounds__(NT) void nb_copy_kernel(const double2 *in, long long in_stride, double2 *out, long long out_stride, int n)
{
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i < n) out[(long long)blockIdx.y * out_stride + i] = in[(long long)blockIdx.y * in_stride + i];
}

double nb_limit(int level) { return level == 2 ? 4.0 : level == 3 ? 2.5 : 6.0; }     // quisk.c:716-728

template <int L> void nb_launch(const NbArgs &a, int nch, int lds, hipStream_t s)
{
    hipLaunchKernelGGL((nb_kernel<L>), dim3((unsigned)a.ntiles * (unsigned)nch), dim3(NT), (size_t)lds, s, a);
}
template <int L> hipError_t nb_attr(int lds)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(nb_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

struct qh_nb {
    int device = 0, nch = 0, sample_rate = 0, hw = 0, S = 0, H = 0, L = 0, lds = 0;
    int level = 0, level_before = 0;
    long long g_change = -(1ll << 60);
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double2 *hist[2] = { nullptr, nullptr };
    int cur = 0;
    ~qh_nb()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(hist[0]); (void)hipFree(hist[1]);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

qh_nb *qh_nb_create(int device, int nch, int sample_rate, void *stream)
{
    if (nch <= 0 || sample_rate < 8000) { set_error(QH_ERR_INVALID, "qh_nb_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_nb *h = new qh_nb();
    h->device = device; h->nch = nch; h->sample_rate = sample_rate;
    h->hw = (int)(sample_rate * 500.E-6 + 0.5);             // QUISK_NB_HWINDOW_SECS, quisk.c:679,702
    h->S = 3 * h->hw;                                        // save_size, quisk.c:703
    h->H = 7 * h->hw;                                        // look-back of the first output: 2 S + hw - 1
    // tile: as many outputs as keep two workgroups on a CU, at least 1024; 16 B of LDS per sample of the tile
    const int halo = 5 * h->hw - 1;
    // Outputs per tile.  A tile reads L + halo samples and keeps 16 B of LDS for each.  Measured on MI355X
    // (tools/nb_bench.py, profiles/r01_notes.md): the kernel is latency-bound, workgroups per CU count for more than
    // the re-read halo -- 1024 beats 512, 2048 and 4096 at 48 k .. 1.536 M (5.1 TB/s of algorithmic traffic at 192 k,
    // 1.0 TB/s at 1.536 M where the halo is 3.7 tiles long).  QH_NB_TILE overrides for experiments.
    int L = 1024;
    if (const char *e = std::getenv("QH_NB_TILE")) L = std::atoi(e) >= 4096 ? 4096 : std::atoi(e) >= 2048 ? 2048 : std::atoi(e) >= 1024 ? 1024 : 512;
    while (L > 512 && (L + halo) * 16 + 1024 > 158 * 1024) L >>= 1;
    h->L = L;
    h->lds = (L + halo) * 16 + 8 + 2 * 8 * (((L + halo) >> 6) + 2);
    if (h->lds > 160 * 1024 - 64) {
        set_error(QH_ERR_UNSUPPORTED, "qh_nb_create: sample rate %d needs a %d-sample window, more than one LDS tile holds", sample_rate, h->hw);
        delete h;
        return nullptr;
    }
    auto fail = [&](const char *what) -> qh_nb * { set_error(QH_ERR_HIP, "qh_nb_create: %s failed", what); delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
        h->own_stream = true;
    }
    h->stream = s;
    for (int i = 0; i < 2; i++) {
        if (hipMalloc((void **)&h->hist[i], (size_t)nch * (size_t)h->H * sizeof(double2)) != hipSuccess) return fail("hipMalloc");
        if (hipMemsetAsync(h->hist[i], 0, (size_t)nch * (size_t)h->H * sizeof(double2), s) != hipSuccess) return fail("hipMemset");
    }
    if ((L == 4096 ? nb_attr<4096>(h->lds) : L == 2048 ? nb_attr<2048>(h->lds) : L == 1024 ? nb_attr<1024>(h->lds) : nb_attr<512>(h->lds)) != hipSuccess)
        return fail("hipFuncSetAttribute");
    return h;
}

void qh_nb_destroy(qh_nb *h) { delete h; }

int qh_nb_delay(const qh_nb *h) { return h ? h->S : 0; }

// set_noise_blanker (quisk.c:4605): 0 off, 1..3 = limit 6.0 / 4.0 / 2.5, from the next processed sample on
int qh_nb_set_level(qh_nb *h, int level)
{
    if (!h || level < 0) return set_error(QH_ERR_INVALID, "qh_nb_set_level: bad arguments");
    if (level == h->level) return QH_OK;
    if (level > 0) {
        // samples judged so far keep their verdict: the kernel uses the earlier limit for times before g_change.
        // (Two changes within 3.5 ms of signal: the older of the two limits is forgotten.)
        if (h->level > 0) h->level_before = h->level;
        else if (h->level_before == 0) h->level_before = level;
        h->g_change = 0;
    } else {
        h->level_before = h->level;      // switched off: delay line and verdicts freeze (quisk.c:695 returns first)
    }
    h->level = level;
    return QH_OK;
}

int qh_nb_reset(qh_nb *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_nb_reset: null handle");
    QH_HIP(hipSetDevice(h->device));
    for (int i = 0; i < 2; i++) QH_HIP(hipMemsetAsync(h->hist[i], 0, (size_t)h->nch * (size_t)h->H * sizeof(double2), h->stream));
    h->g_change = -(1ll << 60);
    return QH_OK;
}

int qh_nb_process(qh_nb *h, const void *d_in, long long in_stride, void *d_out, long long out_stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!d_in || !d_out || in_stride < n || out_stride < n)))
        return set_error(QH_ERR_INVALID, "qh_nb_process: bad arguments");
    if (n == 0) return QH_OK;
    if (d_in == d_out) return set_error(QH_ERR_INVALID, "qh_nb_process: in place is not supported (tiles read their neighbours' input)");
    QH_HIP(hipSetDevice(h->device));
    const double2 *in = static_cast<const double2 *>(d_in);
    double2 *out = static_cast<double2 *>(d_out);
    if (h->level <= 0) {        // off: samples pass undelayed and the delay line stands still
        hipLaunchKernelGGL(nb_copy_kernel, dim3((unsigned)((n + NT - 1) / NT), (unsigned)h->nch), dim3(NT), 0, h->stream, in, in_stride,
                           out, out_stride, n);
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
    NbArgs a{};
    a.in = in; a.hist = h->hist[h->cur]; a.out = out;
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.n = n; a.H = h->H; a.hw = h->hw; a.S = h->S;
    a.ntiles = (n + h->L - 1) / h->L;
    a.limit = nb_limit(h->level) / (double)h->S;
    a.limit_before = nb_limit(h->level_before > 0 ? h->level_before : h->level) / (double)h->S;
    a.g_change = h->g_change;
    if (h->L == 4096) nb_launch<4096>(a, h->nch, h->lds, h->stream);
    else if (h->L == 2048) nb_launch<2048>(a, h->nch, h->lds, h->stream);
    else if (h->L == 1024) nb_launch<1024>(a, h->nch, h->lds, h->stream);
    else nb_launch<512>(a, h->nch, h->lds, h->stream);
    hipLaunchKernelGGL((hist_update_kernel<double, false>), dim3((unsigned)((h->H + NT - 1) / NT), (unsigned)h->nch), dim3(NT), 0, h->stream,
                       in, in_stride, n, h->hist[h->cur], h->hist[h->cur ^ 1], h->H, (const unsigned long long *)nullptr,
                       (const unsigned long long *)nullptr, (const int *)nullptr, (const unsigned char *)nullptr, PackedFmt{});
    h->cur ^= 1;
    if (h->g_change > -(1ll << 59)) h->g_change -= n;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_nb_process_host(qh_nb *h, const void *h_in, long long in_stride, void *h_out, long long out_stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!h_in || !h_out || in_stride < n || out_stride < n)))
        return set_error(QH_ERR_INVALID, "qh_nb_process_host: bad arguments");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    double2 *d = nullptr, *o = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)h->nch * (size_t)n * sizeof(double2)));
    if (hipMalloc((void **)&o, (size_t)h->nch * (size_t)n * sizeof(double2)) != hipSuccess) { (void)hipFree(d); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    int rc = QH_OK;
    if (hipMemcpy2DAsync(d, (size_t)n * 16, h_in, (size_t)in_stride * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyHostToDevice, h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "upload failed");
    if (rc == QH_OK) rc = qh_nb_process(h, d, n, o, n, n);
    if (rc == QH_OK && hipMemcpy2DAsync(h_out, (size_t)out_stride * 16, o, (size_t)n * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyDeviceToHost,
                                         h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "download failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "synchronize failed");
    (void)hipFree(d); (void)hipFree(o);
    return rc;
}

int qh_nb_synchronize(qh_nb *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_nb_synchronize: null handle");
    QH_HIP(hipSetDevice(h->device));
    QH_HIP(hipStreamSynchronize(h->stream));
    return QH_OK;
}

}  // extern "C"
