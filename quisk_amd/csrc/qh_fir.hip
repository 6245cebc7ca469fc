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
#include "qh_stage.hpp"

using namespace qh;

struct qh_fir {
    Stage b;
    bool own_stream = false;
    ~qh_fir()
    {
        (void)hipSetDevice(b.device);
        if (b.stream) (void)hipStreamSynchronize(b.stream);
        b.destroy();
        if (own_stream && b.stream) (void)hipStreamDestroy(b.stream);
    }
};

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
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); delete h; return nullptr; }
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { set_error(QH_ERR_HIP, "stream creation failed"); delete h; return nullptr; }
        h->own_stream = true;
    }
    h->b.stream = s;
    if (h->b.init(device, nch, ntaps, decim, 1, dtype, false, false, false, s)) { delete h; return nullptr; }
    std::vector<cd> taps((size_t)ntaps);
    for (int i = 0; i < ntaps; i++) taps[(size_t)i] = cd(taps_re[i], taps_im ? taps_im[i] : 0.0);
    if (h->b.set_taps(-1, taps)) { delete h; return nullptr; }
    return h;
}

void qh_fir_destroy(qh_fir *h) { delete h; }

int qh_fir_out_count(const qh_fir *h, int n_in)
{
    if (!h || n_in < 0) return 0;
    return h->b.out_count(n_in);
}

int qh_fir_reset(qh_fir *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    return h->b.reset();
}

// Load the filter state from host memory: `hist` = the ntaps-1 most recent input samples of every channel,
// oldest first, [nch][ntaps-1] in the bank's sample type (NULL = zeros); phase = decim_index.
int qh_fir_set_state(qh_fir *h, const void *hist, int phase)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    return h->b.set_state(hist, phase);
}

int qh_fir_process(qh_fir *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null filter");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;                // quisk_cDecimate with count <= 0 produces nothing
    if (!d_in || !d_out) return set_error(QH_ERR_INVALID, "null buffer");
    if (in_stride < n_in || out_stride < h->b.out_count(n_in)) return set_error(QH_ERR_INVALID, "stride shorter than the data");
    return h->b.process(d_in, in_stride, n_in, d_out, out_stride, n_out);
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
    Stage &b = h->b;
    QH_HIP(hipSetDevice(b.device));
    const int nout = b.out_count(n_in);
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
