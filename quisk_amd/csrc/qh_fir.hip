// qh_fir.hip -- batched FIR decimator bank (include/quiskhip.h group 3).
//
// GPU form of Quisk's quisk_cDecimate / quisk_cCDecimate / quisk_cFilter (filter.c:203-257,372-375) and,
// with the 43 non-trivial taps of the 45-tap half-band, quisk_cDecim2HB45 (filter.c:377-417), applied to
// `nch` independent complex streams at once:
//
//     y[m] = sum_k h[k] * x[decim*m + (decim - 1 - phase) - k]        phase = decim_index carried between calls
//
// computed by qh::osfir_kernel (overlap-save; spectral fold for the power-of-two part of `decim`, keep-
// every-k-th for the rest).  fp64 or fp32 arithmetic.
#include <cmath>
#include <cstring>
#include <vector>
#include "qh_design.hpp"
#include "qh_internal.hpp"
#include "qh_kernels.hpp"

namespace qh {

static constexpr int kFirNfft = 4096;

struct FirBank {
    int device = 0, nch = 0, ntaps = 0, decim = 1, dtype = QH_F64;
    int fold = 1, pick = 1, P = 0, Lf = 0;      // Lf: folded outputs per tile (multiple of pick)
    int phase = 0;                              // decim_index: samples consumed since the last output
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void *mask = nullptr, *tw_fwd = nullptr, *tw_inv = nullptr;
    void *hist[2] = { nullptr, nullptr };
    int cur = 0;
    size_t esize = 16;                          // bytes per complex sample

    ~FirBank()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(mask); (void)hipFree(tw_fwd); (void)hipFree(tw_inv); (void)hipFree(hist[0]); (void)hipFree(hist[1]);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

static int upload_cplx(void **dst, const std::vector<cd> &v, int dtype, hipStream_t s)
{
    if (dtype == QH_F64) {
        QH_HIP(hipMalloc(dst, v.size() * sizeof(cd)));
        QH_HIP(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(cd), hipMemcpyHostToDevice, s));
        QH_HIP(hipStreamSynchronize(s));
    } else {
        std::vector<float> f(v.size() * 2);
        for (size_t i = 0; i < v.size(); i++) { f[2 * i] = (float)v[i].real(); f[2 * i + 1] = (float)v[i].imag(); }
        QH_HIP(hipMalloc(dst, f.size() * sizeof(float)));
        QH_HIP(hipMemcpyAsync(*dst, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice, s));
        QH_HIP(hipStreamSynchronize(s));
    }
    return QH_OK;
}

template <typename T, int FOLD> static int set_lds_attr()
{
    const int lds = lds_elems<kFirNfft>() * (int)sizeof(cplx<T>);
    QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<T, kFirNfft, FOLD, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return QH_OK;
}

template <typename T, int FOLD>
static void launch(const FirBank &b, const void *in, long long in_stride, int n_in, void *out, long long out_stride,
                   int n_out, int off)
{
    OsfirArgs<T> a{};
    a.in = static_cast<const cplx<T> *>(in); a.in_stride = in_stride;
    a.hist = static_cast<const cplx<T> *>(b.hist[b.cur]); a.hist_stride = b.P; a.hist_len = b.P;
    a.out = static_cast<cplx<T> *>(out); a.out_stride = out_stride; a.out_offset = 0;
    a.mask = static_cast<const cplx<T> *>(b.mask); a.mask_stride = 0;
    a.tw_fwd = static_cast<const cplx<T> *>(b.tw_fwd); a.tw_inv = static_cast<const cplx<T> *>(b.tw_inv);
    a.n_in = n_in; a.n_out = n_out; a.off = off; a.P = b.P; a.Lout = b.Lf; a.pick = b.pick;
    const int per_tile = b.Lf / b.pick;
    a.ntiles = (n_out + per_tile - 1) / per_tile;
    dim3 grid((unsigned)a.ntiles, (unsigned)b.nch), block(NT);
    hipLaunchKernelGGL((osfir_kernel<T, kFirNfft, FOLD, false>), grid, block, lds_elems<kFirNfft>() * sizeof(cplx<T>),
                       b.stream, a);
    dim3 g((unsigned)((b.P + NT - 1) / NT), (unsigned)b.nch);
    hipLaunchKernelGGL((hist_update_kernel<T, false>), g, dim3(NT), 0, b.stream, static_cast<const cplx<T> *>(in), in_stride,
                       n_in, static_cast<const cplx<T> *>(b.hist[b.cur]), static_cast<cplx<T> *>(b.hist[b.cur ^ 1]), b.P,
                       (const unsigned long long *)nullptr, (const unsigned long long *)nullptr);
}

template <typename T>
static int dispatch(FirBank &b, const void *in, long long in_stride, int n_in, void *out, long long out_stride, int n_out, int off)
{
    if (n_out > 0) {
        switch (b.fold) {
        case 1: launch<T, 1>(b, in, in_stride, n_in, out, out_stride, n_out, off); break;
        case 2: launch<T, 2>(b, in, in_stride, n_in, out, out_stride, n_out, off); break;
        case 4: launch<T, 4>(b, in, in_stride, n_in, out, out_stride, n_out, off); break;
        case 8: launch<T, 8>(b, in, in_stride, n_in, out, out_stride, n_out, off); break;
        default: return set_error(QH_ERR_INVALID, "bad fold");
        }
    } else {
        // fewer than `decim` samples: only the history moves
        dim3 g((unsigned)((b.P + NT - 1) / NT), (unsigned)b.nch);
        hipLaunchKernelGGL((hist_update_kernel<T, false>), g, dim3(NT), 0, b.stream, static_cast<const cplx<T> *>(in),
                           in_stride, n_in, static_cast<const cplx<T> *>(b.hist[b.cur]),
                           static_cast<cplx<T> *>(b.hist[b.cur ^ 1]), b.P, (const unsigned long long *)nullptr,
                           (const unsigned long long *)nullptr);
    }
    b.cur ^= 1;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

}  // namespace qh

using namespace qh;

struct qh_fir { FirBank b; };

extern "C" {

// The 43 taps at delays 0..42 of Quisk's 45-tap half-band (the outer two taps of the 45 are zero):
// h[2i] = h[42-2i] = coef[i] (i = 0..10), h[21] = 0.5  (filter.c:382-385,401-413).
void qh_hb45_taps(double *taps43)
{
    static const double coef[11] = { 0.000018566625444266, -0.000118469698701817, 0.000457318798253456,
        -0.001347840471412094, 0.003321838571445455, -0.007198422696929033, 0.014211106939802483,
        -0.026424776824073383, 0.048414810444971007, -0.096214669073304823, 0.314881034738348550 };
    for (int i = 0; i < 43; i++) taps43[i] = 0.0;
    for (int i = 0; i < 11; i++) { taps43[2 * i] = coef[i]; taps43[42 - 2 * i] = coef[i]; }
    taps43[21] = 0.5;
}

qh_fir *qh_fir_create(int device, int nch, const double *taps_re, const double *taps_im, int ntaps, int decim, int dtype,
                      void *stream)
{
    if (nch <= 0 || !taps_re || ntaps <= 0 || decim <= 0 || (dtype != QH_F64 && dtype != QH_F32)) {
        set_error(QH_ERR_INVALID, "qh_fir_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_fir *h = new qh_fir();
    FirBank &b = h->b;
    b.device = device; b.nch = nch; b.ntaps = ntaps; b.decim = decim; b.dtype = dtype;
    b.esize = dtype == QH_F64 ? 16 : 8;
    b.fold = (decim % 8 == 0) ? 8 : (decim % 4 == 0) ? 4 : (decim % 2 == 0) ? 2 : 1;
    b.pick = decim / b.fold;
    b.P = ((ntaps - 1 + b.fold - 1) / b.fold) * b.fold;
    if (b.P < b.fold) b.P = b.fold;             // keep at least one history row so the buffers exist
    const int lf_max = (kFirNfft - b.P) / b.fold;
    b.Lf = (lf_max / b.pick) * b.pick;
    if (b.Lf < b.pick || b.Lf <= 0) {
        set_error(QH_ERR_UNSUPPORTED, "qh_fir_create: %d taps / decimation %d do not fit a %d-point tile", ntaps, decim, kFirNfft);
        delete h;
        return nullptr;
    }
    auto fail = [&](const char *what) -> qh_fir * {
        if (g_last_error.empty()) set_error(QH_ERR_HIP, "qh_fir_create: %s failed", what);
        delete h;
        return nullptr;
    };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    b.stream = (hipStream_t)stream;
    if (!b.stream) {
        if (hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
        b.own_stream = true;
    }
    std::vector<cd> taps((size_t)ntaps);
    for (int i = 0; i < ntaps; i++) taps[(size_t)i] = cd(taps_re[i], taps_im ? taps_im[i] : 0.0);
    if (upload_cplx(&b.mask, make_mask(taps, kFirNfft), dtype, b.stream)) return fail("mask upload");
    if (upload_cplx(&b.tw_fwd, fft_twiddle_table(kFirNfft), dtype, b.stream)) return fail("twiddle upload");
    if (upload_cplx(&b.tw_inv, fft_twiddle_table(kFirNfft / b.fold), dtype, b.stream)) return fail("twiddle upload");
    for (int i = 0; i < 2; i++) {
        if (hipMalloc(&b.hist[i], (size_t)nch * b.P * b.esize) != hipSuccess) return fail("hipMalloc");
        if (hipMemsetAsync(b.hist[i], 0, (size_t)nch * b.P * b.esize, b.stream) != hipSuccess) return fail("hipMemset");
    }
    int rc = QH_OK;
    if (dtype == QH_F64) {
        switch (b.fold) { case 1: rc = set_lds_attr<double, 1>(); break; case 2: rc = set_lds_attr<double, 2>(); break;
                          case 4: rc = set_lds_attr<double, 4>(); break; default: rc = set_lds_attr<double, 8>(); }
    } else {
        switch (b.fold) { case 1: rc = set_lds_attr<float, 1>(); break; case 2: rc = set_lds_attr<float, 2>(); break;
                          case 4: rc = set_lds_attr<float, 4>(); break; default: rc = set_lds_attr<float, 8>(); }
    }
    if (rc) return fail("hipFuncSetAttribute");
    if (hipStreamSynchronize(b.stream) != hipSuccess) return fail("synchronize");
    return h;
}

void qh_fir_destroy(qh_fir *h) { delete h; }

int qh_fir_out_count(const qh_fir *h, int n_in)
{
    if (!h || n_in < 0) return 0;
    return (h->b.phase + n_in) / h->b.decim;
}

int qh_fir_reset(qh_fir *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    FirBank &b = h->b;
    QH_HIP(hipSetDevice(b.device));
    for (int i = 0; i < 2; i++) QH_HIP(hipMemsetAsync(b.hist[i], 0, (size_t)b.nch * b.P * b.esize, b.stream));
    b.phase = 0;
    return QH_OK;
}

// Load the filter state from host memory: `hist` = the ntaps-1 most recent input samples of every channel,
// oldest first, [nch][ntaps-1] in the bank's sample type (NULL = zeros); phase = decim_index.
int qh_fir_set_state(qh_fir *h, const void *hist, int phase)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    FirBank &b = h->b;
    if (phase < 0 || phase >= b.decim) return set_error(QH_ERR_INVALID, "phase out of range");
    QH_HIP(hipSetDevice(b.device));
    QH_HIP(hipMemsetAsync(b.hist[b.cur], 0, (size_t)b.nch * b.P * b.esize, b.stream));
    const int nh = b.ntaps - 1;
    if (hist && nh > 0)
        QH_HIP(hipMemcpy2DAsync(static_cast<char *>(b.hist[b.cur]) + (size_t)(b.P - nh) * b.esize, (size_t)b.P * b.esize, hist,
                                (size_t)nh * b.esize, (size_t)nh * b.esize, (size_t)b.nch, hipMemcpyHostToDevice, b.stream));
    QH_HIP(hipStreamSynchronize(b.stream));
    b.phase = phase;
    return QH_OK;
}

int qh_fir_process(qh_fir *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;                // quisk_cDecimate with count <= 0 produces nothing
    if (!d_in || !d_out) return set_error(QH_ERR_INVALID, "null buffer");
    FirBank &b = h->b;
    QH_HIP(hipSetDevice(b.device));
    const int nout = (b.phase + n_in) / b.decim;
    if (in_stride < n_in || out_stride < nout) return set_error(QH_ERR_INVALID, "stride shorter than the data");
    const int off = b.decim - 1 - b.phase;
    int rc = b.dtype == QH_F64 ? dispatch<double>(b, d_in, in_stride, n_in, d_out, out_stride, nout, off)
                               : dispatch<float>(b, d_in, in_stride, n_in, d_out, out_stride, nout, off);
    if (rc) return rc;
    b.phase = (b.phase + n_in) % b.decim;
    if (n_out) *n_out = nout;
    return QH_OK;
}

int qh_fir_synchronize(qh_fir *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    QH_HIP(hipSetDevice(h->b.device));
    QH_HIP(hipStreamSynchronize(h->b.stream));
    return QH_OK;
}

int qh_fir_process_host(qh_fir *h, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;
    FirBank &b = h->b;
    QH_HIP(hipSetDevice(b.device));
    const int nout = (b.phase + n_in) / b.decim;
    void *din = nullptr, *dout = nullptr;
    QH_HIP(hipMalloc(&din, (size_t)b.nch * n_in * b.esize));
    QH_HIP(hipMalloc(&dout, (size_t)b.nch * (nout > 0 ? nout : 1) * b.esize));
    hipError_t e = hipMemcpy2DAsync(din, (size_t)n_in * b.esize, h_in, (size_t)in_stride * b.esize, (size_t)n_in * b.esize,
                                    (size_t)b.nch, hipMemcpyHostToDevice, b.stream);
    int rc = QH_OK, got = 0;
    if (e == hipSuccess) rc = qh_fir_process(h, din, n_in, n_in, dout, nout > 0 ? nout : 1, &got);
    if (e == hipSuccess && rc == QH_OK && got > 0)
        e = hipMemcpy2DAsync(h_out, (size_t)out_stride * b.esize, dout, (size_t)got * b.esize, (size_t)got * b.esize,
                             (size_t)b.nch, hipMemcpyDeviceToHost, b.stream);
    hipError_t e2 = hipStreamSynchronize(b.stream);
    (void)hipFree(din); (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return set_error(QH_ERR_HIP, "qh_fir_process_host: copy failed");
    if (n_out) *n_out = got;
    return QH_OK;
}

}  // extern "C"
