// qh_qrx.hip -- bank of Quisk-native receivers (include/quiskhip.h group 6): the receive path of
// quisk_process_samples (quisk.c:2289-2742) for `nch` receivers in one mode.
//
// The reference runs, per block and per receiver, a chain of small FIR stages:
//   tune (quisk.c:2477-2488) -> HB45 / 144D3 / 240D5 / 48dec24 decimators (quisk_process_decimate, quisk.c:1769-1833)
//   -> mode front end (HB45s, 48dec24) -> Rx filter cRxFilterOut / dRxFilterOut (quisk.c:1182-1256) -> detector
//   -> audio interpolators back to 48 ksps (quisk.c:1906-2068) -> mono to both channels (quisk.c:2622-2627).
// Consecutive linear stages compose into ONE equivalent FIR (h1 * up_D1(h2) * up_D1D2(h3) ..., total decimation
// D1 D2 ...; likewise for the interpolators), so the GPU runs 3 overlap-save launches for SSB / CW
//   [tune + equivalent decimator]  ->  [Rx filter, complex taps, real part]  ->  [equivalent interpolator, (d, d)]
// plus a detector kernel for AM (envelope, DC remover) and FM (phase difference, de-emphasis).  Results equal the
// staged computation to rounding.  Stops before process_agc.
#include <cmath>
#include <vector>
#include "qh_stage.hpp"
#include "qh_demod.hpp"

namespace qh {

enum { Q_CWL = 0, Q_CWU, Q_LSB, Q_USB, Q_AM, Q_FM };    // rx_mode_type, quisk.h:55-70
static constexpr int kMaxEqTaps = 2049;

struct FirStageSpec { std::vector<double> h; int decim; };

static std::vector<double> conv(const std::vector<double> &a, const std::vector<double> &b)
{
    std::vector<double> r(a.size() + b.size() - 1, 0.0);
    for (size_t i = 0; i < a.size(); i++)
        for (size_t j = 0; j < b.size(); j++) r[i + j] += a[i] * b[j];
    return r;
}

static std::vector<double> upsample(const std::vector<double> &h, int f)
{
    if (f == 1) return h;
    std::vector<double> r((h.size() - 1) * (size_t)f + 1, 0.0);
    for (size_t i = 0; i < h.size(); i++) r[i * (size_t)f] = h[i];
    return r;
}

static std::vector<double> hb45_dec_taps()      // 43 taps at delays 0..42, filter.c:382-385,401-413
{
    double t[43];
    qh_hb45_taps(t);
    return std::vector<double>(t, t + 43);
}

static std::vector<double> hb45_interp_taps()   // 45 taps, gain 2 (filter.c:420-453): g[2k+1] = g[43-2k] = coef[k], g[22] = 0.5
{
    double t[43];
    qh_hb45_taps(t);                            // t[2k] = coef[k]
    std::vector<double> g(45, 0.0);
    for (int k = 0; k < 11; k++) { g[(size_t)(2 * k + 1)] = 2.0 * t[2 * k]; g[(size_t)(43 - 2 * k)] = 2.0 * t[2 * k]; }
    g[22] = 2.0 * 0.5;
    return g;
}

// quisk_dInterpolate (filter.c:167-201): phases j use taps j + k*interp for k < ntaps/interp, gain interp
static std::vector<double> dinterp_taps(const double *h, int ntaps, int interp)
{
    const int used = (ntaps / interp) * interp;
    std::vector<double> g((size_t)used);
    for (int i = 0; i < used; i++) g[(size_t)i] = h[i] * interp;
    return g;
}

struct Qrx {
    int device = 0, nch = 0, sample_rate = 0, mode = Q_USB, decim_srate = 0, filter_srate = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<Stage *> dec;       // equivalent decimators; dec[0] carries the NCO
    Stage *rxf = nullptr;           // Rx filter (per-channel taps)
    Stage *post = nullptr;          // FM only: 186-tap /4 + 309-tap high-pass as one decimator
    Stage *up = nullptr;            // equivalent interpolator to 48 ksps, output (d, d)
    std::vector<int> rx_size;       // sizeFilter per channel (0 = pass through like the reference)
    double *dc_state = nullptr;     // AM
    double4 *fm_state = nullptr;    // FM
    QFmParam fm_prm{};
    double2 *buf[2] = { nullptr, nullptr };
    long long buf_cap = 0;

    ~Qrx()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (Stage *s : dec) { s->destroy(); delete s; }
        for (Stage *s : { rxf, post, up }) if (s) { s->destroy(); delete s; }
        (void)hipFree(dc_state); (void)hipFree(fm_state); (void)hipFree(buf[0]); (void)hipFree(buf[1]);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace qh

using namespace qh;
struct qh_qrx { Qrx q; };

extern "C" {

qh_qrx *qh_qrx_create(int device, int nch, int sample_rate, int mode, const double *f48dec24, const double *f144d3,
                      const double *f240d5, const double *audio24p4, const double *audio24p6, const double *lp48,
                      const double *fmhp, void *stream)
{
    if (nch <= 0 || sample_rate <= 0 || mode < Q_CWL || mode > Q_FM || !f48dec24 || !f144d3 || !f240d5 || !audio24p4 ||
        !audio24p6 || !lp48 || !fmhp) {
        set_error(QH_ERR_INVALID, "qh_qrx_create: bad arguments");
        return nullptr;
    }
    // PlanDecimation, quisk.c:1633-1671
    int best = sample_rate, d2 = 0, d3 = 0, d5 = 0;
    for (int i2 = 0; i2 <= 6; i2++)
        for (int i3 = 0; i3 <= 3; i3++)
            for (int i5 = 0; i5 <= 3; i5++) {
                int t = sample_rate;
                for (int i = 0; i < i2; i++) t /= 2;
                for (int i = 0; i < i3; i++) t /= 3;
                for (int i = 0; i < i5; i++) t /= 5;
                if (t >= 48000 && t < best) { d2 = i2; d3 = i3; d5 = i5; best = t; }
            }
    {
        int t = sample_rate;
        for (int i = 0; i < d2; i++) t /= 2;
        for (int i = 0; i < d3; i++) t /= 3;
        for (int i = 0; i < d5; i++) t /= 5;
        if (t != 48000) {
            set_error(QH_ERR_UNSUPPORTED, "sample rate %d does not decimate to 48000 by 2, 3 and 5 (the 6/5 x 4/5 stage is not provided)", sample_rate);
            return nullptr;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_qrx *h = new qh_qrx();
    Qrx &q = h->q;
    q.device = device; q.nch = nch; q.sample_rate = sample_rate; q.mode = mode; q.decim_srate = 48000;
    q.rx_size.assign((size_t)nch, 0);
    auto fail = [&]() -> qh_qrx * { delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); return fail(); }
    q.stream = (hipStream_t)stream;
    if (!q.stream) {
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess) { set_error(QH_ERR_HIP, "stream creation failed"); return fail(); }
        q.own_stream = true;
    }
    // ---- the reference's decimating stages in order (quisk.c:1769-1833, then the mode's front end 1906-2030)
    std::vector<FirStageSpec> st;
    const std::vector<double> hb = hb45_dec_taps();
    const std::vector<double> v48(f48dec24, f48dec24 + 98), v3(f144d3, f144d3 + 147), v5(f240d5, f240d5 + 245);
    for (int i = 0; i < d2 - 1 && i < 5; i++) st.push_back({ hb, 2 });
    for (int i = 0; i < d3; i++) st.push_back({ v3, 3 });
    for (int i = 0; i < d5; i++) st.push_back({ v5, 5 });
    if (d2 > 0) st.push_back({ v48, 2 });
    switch (mode) {
    case Q_CWL: case Q_CWU: q.filter_srate = 6000;  st.push_back({ hb, 2 }); st.push_back({ hb, 2 }); st.push_back({ v48, 2 }); break;
    case Q_LSB: case Q_USB: q.filter_srate = 12000; st.push_back({ hb, 2 }); st.push_back({ v48, 2 }); break;
    case Q_AM:              q.filter_srate = 24000; st.push_back({ v48, 2 }); break;
    default:                q.filter_srate = 48000; break;
    }
    // ---- greedy grouping into equivalent decimators of at most kMaxEqTaps taps
    std::vector<FirStageSpec> groups;
    {
        std::vector<double> heq(1, 1.0);
        int deq = 1;
        for (const FirStageSpec &s : st) {
            const size_t len = heq.size() + (size_t)deq * (s.h.size() - 1);
            if (len > (size_t)kMaxEqTaps && deq > 1) {
                groups.push_back({ heq, deq });
                heq.assign(1, 1.0); deq = 1;
            }
            heq = conv(heq, upsample(s.h, deq));
            deq *= s.decim;
        }
        groups.push_back({ heq, deq });         // possibly the identity (FM at 48 ksps): still carries the NCO
    }
    for (size_t g = 0; g < groups.size(); g++) {
        Stage *s = new Stage();
        q.dec.push_back(s);
        if (s->init(device, nch, (int)groups[g].h.size(), groups[g].decim, 1, QH_F64, g == 0, false, false, q.stream)) return fail();
        std::vector<cd> taps(groups[g].h.size());
        for (size_t i = 0; i < taps.size(); i++) taps[i] = cd(groups[g].h[i], 0.0);
        if (s->set_taps(-1, taps)) return fail();
    }
    // ---- Rx filter: per-channel taps, up to 2048; identity until set_filters is called (sizeFilter == 0)
    q.rxf = new Stage();
    const bool real_out = mode <= Q_USB;
    if (q.rxf->init(device, nch, 2048, 1, 1, QH_F64, false, true, real_out, q.stream)) return fail();
    // sizeFilter == 0: c/dRxFilterOut return the sample itself (quisk.c:1201,1239), so SSB/CW give re -+ im
    {
        cd id(1.0, 0.0);
        if (mode == Q_CWU || mode == Q_USB) id = cd(1.0, 1.0);
        if (mode == Q_CWL || mode == Q_LSB) id = cd(1.0, -1.0);
        if (q.rxf->set_taps(-1, std::vector<cd>(1, id))) return fail();
    }
    // ---- back to 48 ksps
    const std::vector<double> g45 = hb45_interp_taps();
    std::vector<double> ueq;
    int U = 1;
    if (mode == Q_CWL || mode == Q_CWU) {           // dInterpolate(Audio24p4 table, 2), HB45, HB45 (quisk.c:1930-1932)
        ueq = conv(conv(upsample(dinterp_taps(audio24p4, 50, 2), 4), upsample(g45, 2)), g45); U = 8;
    } else if (mode == Q_LSB || mode == Q_USB) {    // dInterpolate(Audio24p4, 2), HB45 (quisk.c:1975-1976)
        ueq = conv(upsample(dinterp_taps(audio24p4, 50, 2), 2), g45); U = 4;
    } else if (mode == Q_AM) {                      // dFilter(Audio24p6), HB45 (quisk.c:2017,2024)
        ueq = conv(upsample(std::vector<double>(audio24p6, audio24p6 + 36), 2), g45); U = 2;
    } else {                                        // FM: HB45, HB45 after the /4 (quisk.c:2067-2068)
        ueq = conv(upsample(g45, 2), g45); U = 4;
        // dDecimate(LpFilt48, 4) then dFilter(AudioFmHp) (quisk.c:2065-2066) as one decimator
        std::vector<double> p = conv(std::vector<double>(lp48, lp48 + 186), upsample(std::vector<double>(fmhp, fmhp + 309), 4));
        q.post = new Stage();
        if (q.post->init(device, nch, (int)p.size(), 4, 1, QH_F64, false, false, false, q.stream)) return fail();
        std::vector<cd> taps(p.size());
        for (size_t i = 0; i < p.size(); i++) taps[i] = cd(p[i], 0.0);
        if (q.post->set_taps(-1, taps)) return fail();
    }
    q.up = new Stage();
    if (q.up->init(device, nch, (int)ueq.size(), 1, U, QH_F64, false, false, true, q.stream)) return fail();
    {
        std::vector<cd> taps(ueq.size());
        for (size_t i = 0; i < taps.size(); i++) taps[i] = cd(ueq[i], 0.0);
        if (q.up->set_taps(-1, taps)) return fail();
        for (int c = 0; c < nch; c++) {
            if (q.up->set_epi(c, EpiParam{ 1, 0, 1, 0 })) return fail();                    // d + I*d, quisk.c:2625
            if (real_out && q.rxf->set_epi(c, EpiParam{ 1, 0, 0, 0 })) return fail();       // re -+ im collapses to the real part
        }
    }
    if (mode == Q_AM) {
        if (hipMalloc((void **)&q.dc_state, (size_t)nch * 8) != hipSuccess || hipMemset(q.dc_state, 0, (size_t)nch * 8) != hipSuccess) {
            set_error(QH_ERR_HIP, "allocation failed"); return fail();
        }
    }
    if (mode == Q_FM) {
        std::vector<double4> init((size_t)nch, make_double4(10.0, 0.0, 0.0, 0.0));          // fm_1 = 10, quisk.c:1893
        if (hipMalloc((void **)&q.fm_state, (size_t)nch * sizeof(double4)) != hipSuccess ||
            hipMemcpy(q.fm_state, init.data(), (size_t)nch * sizeof(double4), hipMemcpyHostToDevice) != hipSuccess) {
            set_error(QH_ERR_HIP, "allocation failed"); return fail();
        }
        const double www = std::tan(M_PI * 300.0 / 48000);                                  // quisk.c:1894-1898
        const double nnn = 1.0 / (1.0 + www);
        q.fm_prm.a0 = www * nnn; q.fm_prm.a1 = q.fm_prm.a0; q.fm_prm.b1 = nnn * (www - 1.0);
    }
    return h;
}

void qh_qrx_destroy(qh_qrx *h) { delete h; }
int qh_qrx_filter_rate(const qh_qrx *h) { return h ? h->q.filter_srate : 0; }

// set_tune (quisk.c:4702): the stream is multiplied by exp(-j 2 pi tune n / sample_rate)
int qh_qrx_set_tune(qh_qrx *h, int ch, int rx_tune_freq)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (ch < -1 || ch >= q.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? q.nch : ch + 1); c++)
        if (int rc = q.dec[0]->set_nco(c, -(double)rx_tune_freq, (double)q.sample_rate)) return rc;
    return QH_OK;
}

// set_filters (quisk.c:4551): taps as MakeFilterCoef designs them.  cRxFilterOut's ring walk (quisk.c:1246-1253)
// pairs tap 0 with the newest sample and taps 1..N-1 with the oldest..second newest: as a convolution
// g[0] = h[0], g[d] = h[N-d].  SSB/CW keep re -+ im = Re{(gI +- j gQ) * x}; AM/FM use filtI on both parts.
int qh_qrx_set_filters(qh_qrx *h, int ch, const double *filtI, const double *filtQ, int size)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (ch < -1 || ch >= q.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    if (size < 0 || size > 2048 || (size > 0 && (!filtI || !filtQ)))
        return set_error(QH_ERR_UNSUPPORTED, "Rx filter size must be 0..2048 (got %d)", size);
    cd id(1.0, 0.0);
    if (q.mode == Q_CWU || q.mode == Q_USB) id = cd(1.0, 1.0);
    if (q.mode == Q_CWL || q.mode == Q_LSB) id = cd(1.0, -1.0);
    std::vector<cd> g((size_t)(size > 0 ? size : 1), id);
    for (int d = 0; d < size; d++) {
        const int k = d == 0 ? 0 : size - d;
        const double gi = filtI[k], gq = filtQ[k];
        switch (q.mode) {
        case Q_CWU: case Q_USB: g[(size_t)d] = cd(gi, gq); break;      // re - im
        case Q_CWL: case Q_LSB: g[(size_t)d] = cd(gi, -gq); break;     // re + im
        default: g[(size_t)d] = cd(gi, 0.0); break;                    // dRxFilterOut
        }
    }
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? q.nch : ch + 1); c++) {
        if (int rc = q.rxf->set_taps(c, g)) return rc;
        q.rx_size[(size_t)c] = size;
    }
    return QH_OK;
}

int qh_qrx_out_count(const qh_qrx *h, int n_in)
{
    if (!h) return 0;
    const Qrx &q = h->q;
    int n = n_in;
    for (const Stage *s : q.dec) n = s->out_count(n);
    if (q.post) n = q.post->out_count(n);
    return q.up->out_count(n);
}

int qh_qrx_process(qh_qrx *h, const double *d_in, long long in_stride, int n_in, double *d_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;
    if (!d_in || !d_out || in_stride < n_in) return set_error(QH_ERR_INVALID, "bad buffers");
    Qrx &q = h->q;
    QH_HIP(hipSetDevice(q.device));
    const int total = qh_qrx_out_count(h, n_in);
    if (out_stride < total) return set_error(QH_ERR_INVALID, "output stride %lld shorter than %d samples", out_stride, total);
    // intermediate buffers: no stage after the first produces more than max(n_in, total) samples
    const long long need = (long long)(n_in > total ? n_in : total) + 8;
    if (need > q.buf_cap) {
        QH_HIP(hipStreamSynchronize(q.stream));
        for (int i = 0; i < 2; i++) { (void)hipFree(q.buf[i]); q.buf[i] = nullptr; }
        for (int i = 0; i < 2; i++) QH_HIP(hipMalloc((void **)&q.buf[i], (size_t)q.nch * (size_t)need * 16));
        q.buf_cap = need;
    }
    const void *cur = d_in;
    long long cur_stride = in_stride;
    int n = n_in, w = 0;
    for (Stage *s : q.dec) {
        int m = 0;
        if (int rc = s->process(cur, cur_stride, n, q.buf[w], q.buf_cap, &m)) return rc;
        cur = q.buf[w]; cur_stride = q.buf_cap; n = m; w ^= 1;
    }
    {
        int m = 0;
        if (int rc = q.rxf->process(cur, cur_stride, n, q.buf[w], q.buf_cap, &m)) return rc;
        cur = q.buf[w]; n = m; w ^= 1;
    }
    if (q.mode == Q_AM && n > 0)
        hipLaunchKernelGGL(q_am_env_kernel, dim3((unsigned)q.nch), dim3(64), 0, q.stream, const_cast<double2 *>(static_cast<const double2 *>(cur)),
                           q.buf_cap, n, q.dc_state);
    if (q.mode == Q_FM && n > 0) {
        hipLaunchKernelGGL(q_fm_disc_kernel, dim3((unsigned)q.nch), dim3(64), 0, q.stream, const_cast<double2 *>(static_cast<const double2 *>(cur)),
                           q.buf_cap, n, q.fm_state, q.fm_prm);
        int m = 0;
        if (int rc = q.post->process(cur, q.buf_cap, n, q.buf[w], q.buf_cap, &m)) return rc;
        cur = q.buf[w]; n = m; w ^= 1;
    }
    int m = 0;
    if (int rc = q.up->process(cur, q.buf_cap, n, d_out, out_stride, &m)) return rc;
    if (n_out) *n_out = m;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_qrx_process_host(qh_qrx *h, const double *h_in, long long in_stride, int n_in, double *h_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;
    Qrx &q = h->q;
    QH_HIP(hipSetDevice(q.device));
    const int total = qh_qrx_out_count(h, n_in);
    double2 *din = nullptr, *dout = nullptr;
    QH_HIP(hipMalloc((void **)&din, (size_t)q.nch * n_in * 16));
    QH_HIP(hipMalloc((void **)&dout, (size_t)q.nch * (size_t)(total > 0 ? total : 1) * 16));
    hipError_t e = hipMemcpy2DAsync(din, (size_t)n_in * 16, h_in, (size_t)in_stride * 16, (size_t)n_in * 16, (size_t)q.nch, hipMemcpyHostToDevice, q.stream);
    int rc = QH_OK, got = 0;
    if (e == hipSuccess) rc = qh_qrx_process(h, reinterpret_cast<const double *>(din), n_in, n_in, reinterpret_cast<double *>(dout), total > 0 ? total : 1, &got);
    if (e == hipSuccess && rc == QH_OK && got > 0)
        e = hipMemcpy2DAsync(h_out, (size_t)out_stride * 16, dout, (size_t)(total > 0 ? total : 1) * 16, (size_t)got * 16, (size_t)q.nch, hipMemcpyDeviceToHost, q.stream);
    hipError_t e2 = hipStreamSynchronize(q.stream);
    (void)hipFree(din); (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return set_error(QH_ERR_HIP, "qh_qrx_process_host: copy failed");
    if (n_out) *n_out = got;
    return QH_OK;
}

int qh_qrx_synchronize(qh_qrx *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    QH_HIP(hipSetDevice(h->q.device));
    QH_HIP(hipStreamSynchronize(h->q.stream));
    return QH_OK;
}

}  // extern "C"
