// qh_polyphase.hip -- batched time-domain polyphase resampler (include/quiskhip.h group 3c).
//
// GPU form of quisk_cInterpDecim (filter.c:287-324) and, as special cases, quisk_cInterpolate (filter.c:131-165,
// decim = 1), quisk_dInterpolate (:167-201) / quisk_cInterp2HB45 / quisk_dInterp2HB45 (:420-488) -- real streams
// ride as the real and imaginary parts of a complex one, the taps being real.  With the upsampled grid position
// p_m = phase + m decim,  i = p_m / interp,  ph = p_m % interp:
//
//     y[m] = interp * sum_{k < K} taps[ph + k interp] * x[i - k],      K = ceil(ntaps / interp)
//
// `phase` is the reference's decim_index and carries over between calls together with the last K-1 inputs.
// This is the path for rational ratios (Quisk's 6/5 and 4/5 stages, quisk.c:1834-1838) and audio-rate
// interpolators, where an FFT tile would be mostly padding: one lane per output, taps phase-major so a lane
// walks contiguous memory, inputs through L1/L2.  The high-rate decimators use qh_fir / qh_hbc instead.
#include <cstdlib>
#include <vector>
#include "qh_fft.hpp"
#include "qh_internal.hpp"

using namespace qh;

namespace {

template <typename T>
__global__ __launch_bounds__(NT) void polyphase_kernel(const cplx<T> *in, long long in_stride, const cplx<T> *hist, int n_in,
                                                       cplx<T> *out, long long out_stride, int n_out, const T *hp, int K, int U,
                                                       int D, int p0, T gain)
{
    const int m = blockIdx.x * NT + threadIdx.x;
    const int ch = blockIdx.y;
    if (m >= n_out) return;
    const long long p = (long long)p0 + (long long)m * D;
    const int i = (int)(p / U), ph = (int)(p - (long long)i * U);
    const cplx<T> *x = in + (long long)ch * in_stride;
    const cplx<T> *h = hist + (long long)ch * (K - 1) + (K - 1);        // h[-d] = the d-th sample before in[0]
    const T *c = hp + (long long)ph * K;
    T ar = 0, ai = 0;
    for (int k = 0; k < K; k++) {
        const int idx = i - k;
        const cplx<T> v = idx >= 0 ? x[idx] : h[idx];
        ar += v.x * c[k];
        ai += v.y * c[k];
    }
    out[(long long)ch * out_stride + m] = mk<T>(ar * gain, ai * gain);
}

// hist_new[j] = stream sample (n_in - len + j) counted from in[0]; negative positions come from the old history
template <typename T>
__global__ __launch_bounds__(NT) void polyphase_hist_kernel(const cplx<T> *in, long long in_stride, int n_in, const cplx<T> *hist_old,
                                                            cplx<T> *hist_new, int len)
{
    const int ch = blockIdx.y;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= len) return;
    const long long p = (long long)n_in - len + j;
    hist_new[(long long)ch * len + j] = p >= 0 ? in[(long long)ch * in_stride + p] : hist_old[(long long)ch * len + len + p];
}

}  // namespace

struct qh_rat {
    int device = 0, nch = 0, ntaps = 0, U = 1, D = 1, K = 1, dtype = QH_F64, phase = 0;
    size_t esize = 16;
    void *hp = nullptr;             // [U][K] taps, phase-major, zero padded
    void *hist[2] = { nullptr, nullptr };
    int cur = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ~qh_rat()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(hp);
        for (auto &h : hist) (void)hipFree(h);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    size_t hist_bytes() const { return (size_t)nch * (size_t)(K > 1 ? K - 1 : 1) * esize; }
};

namespace {

int count_for(const qh_rat *h, int n_in)
{
    const long long span = (long long)n_in * h->U;
    if (h->phase >= span) return 0;
    return (int)((span - h->phase + h->D - 1) / h->D);
}

template <typename T>
int run(qh_rat *h, const void *in, long long in_stride, int n_in, void *out, long long out_stride, int n_out)
{
    if (n_out > 0) {
        hipLaunchKernelGGL(polyphase_kernel<T>, dim3((unsigned)((n_out + NT - 1) / NT), (unsigned)h->nch), dim3(NT), 0, h->stream,
                           (const cplx<T> *)in, in_stride, (const cplx<T> *)h->hist[h->cur], n_in, (cplx<T> *)out, out_stride, n_out,
                           (const T *)h->hp, h->K, h->U, h->D, h->phase, (T)h->U);
        QH_HIP(hipGetLastError());
    }
    if (h->K > 1) {
        hipLaunchKernelGGL(polyphase_hist_kernel<T>, dim3((unsigned)((h->K - 1 + NT - 1) / NT), (unsigned)h->nch), dim3(NT), 0, h->stream,
                           (const cplx<T> *)in, in_stride, n_in, (const cplx<T> *)h->hist[h->cur], (cplx<T> *)h->hist[h->cur ^ 1], h->K - 1);
        QH_HIP(hipGetLastError());
        h->cur ^= 1;
    }
    return QH_OK;
}

}  // namespace

extern "C" {

qh_rat *qh_rat_create(int device, int nch, const double *taps, int ntaps, int interp, int decim, int dtype, void *stream)
{
    if (nch <= 0 || !taps || ntaps <= 0 || interp <= 0 || decim <= 0 || (dtype != QH_F64 && dtype != QH_F32)) {
        set_error(QH_ERR_INVALID, "qh_rat_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_rat *h = new qh_rat();
    h->device = device; h->nch = nch; h->ntaps = ntaps; h->U = interp; h->D = decim; h->dtype = dtype;
    h->esize = dtype == QH_F64 ? 16 : 8;
    // quisk_cInterpDecim uses nTaps / interp taps per phase (integer division, filter.c:308); a remainder would
    // drop the last taps there.  The half-band interpolators have 45 taps for 2 phases (23 + 22): callers who
    // want those pass ntaps as is and get K = ceil, the missing tap being zero.
    h->K = (ntaps + interp - 1) / interp;
    auto fail = [&](const char *what) -> qh_rat * { set_error(QH_ERR_HIP, "qh_rat_create: %s failed", what); delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
        h->own_stream = true;
    }
    h->stream = s;
    const size_t nt = (size_t)h->U * (size_t)h->K;
    if (hipMalloc(&h->hp, nt * (h->esize / 2)) != hipSuccess) return fail("hipMalloc");
    if (dtype == QH_F64) {
        std::vector<double> t(nt, 0.0);
        for (int i = 0; i < ntaps; i++) t[(size_t)(i % interp) * h->K + (size_t)(i / interp)] = taps[i];
        if (hipMemcpy(h->hp, t.data(), nt * 8, hipMemcpyHostToDevice) != hipSuccess) return fail("tap upload");
    } else {
        std::vector<float> t(nt, 0.0f);
        for (int i = 0; i < ntaps; i++) t[(size_t)(i % interp) * h->K + (size_t)(i / interp)] = (float)taps[i];
        if (hipMemcpy(h->hp, t.data(), nt * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("tap upload");
    }
    for (auto &p : h->hist)
        if (hipMalloc(&p, h->hist_bytes()) != hipSuccess || hipMemsetAsync(p, 0, h->hist_bytes(), s) != hipSuccess) return fail("history allocation");
    return h;
}

void qh_rat_destroy(qh_rat *h) { delete h; }

int qh_rat_reset(qh_rat *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_rat_reset: null handle");
    QH_HIP(hipSetDevice(h->device));
    for (auto &p : h->hist) QH_HIP(hipMemsetAsync(p, 0, h->hist_bytes(), h->stream));
    h->phase = 0;
    return QH_OK;
}

int qh_rat_set_state(qh_rat *h, const void *hist, int phase)
{
    if (!h || phase < 0) return set_error(QH_ERR_INVALID, "qh_rat_set_state: bad arguments");
    QH_HIP(hipSetDevice(h->device));
    if (hist && h->K > 1) QH_HIP(hipMemcpyAsync(h->hist[h->cur], hist, h->hist_bytes(), hipMemcpyHostToDevice, h->stream));
    else QH_HIP(hipMemsetAsync(h->hist[h->cur], 0, h->hist_bytes(), h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    h->phase = phase;
    return QH_OK;
}

int qh_rat_phase(const qh_rat *h) { return h ? h->phase : 0; }
int qh_rat_out_count(const qh_rat *h, int n_in) { return h && n_in > 0 ? count_for(h, n_in) : 0; }

int qh_rat_process(qh_rat *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride, int *n_out)
{
    if (n_out) *n_out = 0;
    if (!h || n_in < 0) return set_error(QH_ERR_INVALID, "qh_rat_process: bad arguments");
    if (n_in == 0) return QH_OK;
    const int m = count_for(h, n_in);
    if (!d_in || in_stride < n_in || (m > 0 && (!d_out || out_stride < m))) return set_error(QH_ERR_INVALID, "qh_rat_process: bad buffers");
    QH_HIP(hipSetDevice(h->device));
    const int rc = h->dtype == QH_F64 ? run<double>(h, d_in, in_stride, n_in, d_out, out_stride, m)
                                      : run<float>(h, d_in, in_stride, n_in, d_out, out_stride, m);
    if (rc) return rc;
    h->phase = (int)((long long)h->phase + (long long)m * h->D - (long long)n_in * h->U);
    if (n_out) *n_out = m;
    return QH_OK;
}

int qh_rat_process_host(qh_rat *h, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride, int *n_out)
{
    if (n_out) *n_out = 0;
    if (!h || n_in < 0) return set_error(QH_ERR_INVALID, "qh_rat_process_host: bad arguments");
    if (n_in == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    const int m = count_for(h, n_in);
    void *di = nullptr, *dout = nullptr;
    QH_HIP(hipMalloc(&di, (size_t)h->nch * (size_t)n_in * h->esize));
    if (hipMalloc(&dout, (size_t)h->nch * (size_t)(m > 0 ? m : 1) * h->esize) != hipSuccess) { (void)hipFree(di); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    int rc = QH_OK, got = 0;
    hipError_t e = hipMemcpy2DAsync(di, (size_t)n_in * h->esize, h_in, (size_t)in_stride * h->esize, (size_t)n_in * h->esize, (size_t)h->nch,
                                    hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) rc = set_error(QH_ERR_HIP, "qh_rat_process_host: upload failed");
    if (rc == QH_OK) rc = qh_rat_process(h, di, n_in, n_in, dout, m > 0 ? m : 1, &got);
    if (rc == QH_OK && got > 0) {
        e = hipMemcpy2DAsync(h_out, (size_t)out_stride * h->esize, dout, (size_t)m * h->esize, (size_t)got * h->esize, (size_t)h->nch,
                             hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) rc = set_error(QH_ERR_HIP, "qh_rat_process_host: download failed");
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "qh_rat_process_host: synchronize failed");
    (void)hipFree(di); (void)hipFree(dout);
    if (rc == QH_OK && n_out) *n_out = got;
    return rc;
}

int qh_rat_synchronize(qh_rat *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_rat_synchronize: null handle");
    QH_HIP(hipSetDevice(h->device));
    QH_HIP(hipStreamSynchronize(h->stream));
    return QH_OK;
}

}  // extern "C"
