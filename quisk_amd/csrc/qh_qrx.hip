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
#include <mutex>
#include <vector>
#include "qh_stage.hpp"
#include "qh_qdemod.hpp"

static constexpr int kQTiledMin = 8192;     // samples per call from which the detectors run over time segments

namespace qh {

// rx_mode_type, quisk.h:55-70
enum { Q_CWL = 0, Q_CWU, Q_LSB, Q_USB, Q_AM, Q_FM, Q_EXT, Q_DGT_U, Q_DGT_L, Q_DGT_IQ, Q_IMD, Q_FDV_U, Q_FDV_L, Q_DGT_FM };
static constexpr int kMaxEqTaps = 2049;
static constexpr int kDgtNarrowFreq = 3000;     // DGT_NARROW_FREQ, quisk.c:52

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

// ---- Rx filters longer than one 4096-point tile allows (2048 taps): K partitions of 2048 taps, partition k on the stream delayed by
// 2048 k samples, their outputs added.  [delay line | block] is put together once per call.
static __global__ void rxp_gather_kernel(const double2 *dly, int D, const double2 *in, long long in_stride, int n, double2 *x, long long x_stride)
{
    const long long row = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D + n; i += gridDim.x * blockDim.x)
        x[row * x_stride + i] = i < D ? dly[row * D + i] : in[row * in_stride + (i - D)];
}
static __global__ void rxp_keep_kernel(const double2 *x, long long x_stride, int D, int n, double2 *dly)
{
    const long long row = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) dly[row * D + i] = x[row * x_stride + n + i];
}
static __global__ void rxp_add_kernel(double2 *y, long long y_stride, const double2 *t, long long t_stride, int n)
{
    const long long row = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double2 a = y[row * y_stride + i];
        const double2 b = t[row * t_stride + i];
        a.x += b.x; a.y += b.y;
        y[row * y_stride + i] = a;
    }
}

// quisk_dInterpolate (filter.c:167-201): phases j use taps j + k*interp for k < ntaps/interp, gain interp
static std::vector<double> dinterp_taps(const double *h, int ntaps, int interp)
{
    const int used = (ntaps / interp) * interp;
    std::vector<double> g((size_t)used);
    for (int i = 0; i < used; i++) g[(size_t)i] = h[i] * interp;
    return g;
}

// mode classes of quisk_process_demodulate (quisk.c:1906-2153); modes without a case of their own take the SSB path
static bool is_cw(int m) { return m == Q_CWL || m == Q_CWU; }
static bool is_am(int m) { return m == Q_AM; }
static bool is_fm(int m) { return m == Q_FM || m == Q_DGT_FM; }
static bool is_dgt(int m) { return m == Q_DGT_U || m == Q_DGT_L || m == Q_FDV_U || m == Q_FDV_L; }
static bool is_iq(int m) { return m == Q_DGT_IQ; }
static bool is_ssb(int m) { return !is_cw(m) && !is_am(m) && !is_fm(m) && !is_dgt(m) && !is_iq(m); }
static bool lower(int m) { return m == Q_CWL || m == Q_LSB || m == Q_DGT_L || m == Q_FDV_L; }     // re + im
static bool sideband(int m) { return is_cw(m) || is_ssb(m) || is_dgt(m); }                          // cRxFilterOut, re -+ im

struct Step {
    enum Kind { FIR, RAT, AM_ENV, FM_DISC, SSB_SQ, DELAY, NOTCH } kind;
    Stage *st = nullptr;
    qh_rat *rat = nullptr;
    bool dup = false;               // NOTCH: the stream is already (d, d) here, keep both parts equal
};

struct Qrx {
    int device = 0, nch = 0, sample_rate = 0, mode = Q_USB, bandwidth = 2700, decim_srate = 0, filter_srate = 0;
    int ssb_sq_bandwidth = 2700;            // filter_bandwidth[0], whatever bank or filter set this receiver is (quisk.c:1120)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<Step> steps;        // in order; steps[0] is an overlap-save stage that carries the NCO
    Stage *rxf = nullptr;           // Rx filter (per-channel taps); owned by its step
    // filters of more than kRxPart taps: partitions 1 .. K - 1 (taps [kRxPart k, kRxPart (k + 1)) on the stream delayed by kRxPart k)
    static constexpr int kRxPart = 2048, kRxMaxTaps = 10000;       // MAX_FILTER_SIZE - 1 (quisk.h:10, quisk.c:4575)
    std::vector<Stage *> rxf_more;
    bool rx_sideband_epi = false, rx_direct_out = false;
    double2 *rx_dly[2] = { nullptr, nullptr }, *rx_x = nullptr, *rx_tmp = nullptr;
    int rx_dly_cur = 0;
    long long rx_x_cap = 0;
    int rx_D() const { return (int)rxf_more.size() * kRxPart; }
    double *dc_state = nullptr;     // AM
    double4 *fm_state = nullptr;    // FM
    double4 *fm_state_new = nullptr;
    double *fm_part = nullptr;      // the squelch's partial sums of a long call, one per time segment
    long long fm_part_cap = 0;
    QFmParam fm_prm{};
    double2 *buf[2] = { nullptr, nullptr };
    long long buf_cap = 0;
    // SSB squelch (quisk.c:1086-1180) + its 512-sample audio delay (quisk.c:1057-1084)
    bool has_ssb_sq = false, ssb_sq_on = false, ssb_sq_inited = false;
    int ssb_sq_level = 0;
    QSsbSqState *ssq_state = nullptr;
    double *ssq_ring = nullptr;
    double2 *ssq_delay[2] = { nullptr, nullptr };
    int ssq_cur = 0;
    QSquelchState *sq_state = nullptr;  // squelch flag (FM squelch state, quisk.c:2076-2085)
    bool sq_defer = false;              // the calls are pieces of one quisk_process_samples call (qh_qrx_squelch_pieces)
    bool mute_deferred = false;         // the caller applies the squelch itself, behind its AGC (qh_quisk_process_samples)
    double *sq_level = nullptr;
    std::vector<double> h_sq_level;
    bool sq_dirty = false;
    // c/dRxFilterOut keep their samples in a ring of sizeFilter entries with a running write index (quisk.c:1225-1253);
    // a set_filters call with another size re-reads the same storage as a ring of the new size.  The linear history of
    // the overlap-save stage is rebuilt to hold exactly what the reference's taps will then see: rx_size / rx_index are
    // the reference's sizeFilter / indexFilter per channel, rx_ring what its buffer holds outside the live ring.
    std::vector<int> rx_size, rx_index;
    std::vector<std::vector<cd>> rx_ring;
    std::vector<char> rx_ring_fresh;        // rx_ring[c] is what the reference's buffer holds right now (no sample since rx_resize brought it up to date)
    qh_qagc *agc = nullptr;         // process_agc on the output (quisk.c:2686-2702); null = never switched on
    bool agc_on = false;
    double agc_gain = 80.0;
    // dAutoNotch (quisk.c:786-963): a NOTCH step sits where the mode calls it, idle until qh_qrx_set_auto_notch
    QNotchState *notch_state = nullptr;
    double2 *tw2048 = nullptr;
    bool notch_on = false;
    int rit_freq = 0;
    qh_nb *nb = nullptr;            // NoiseBlanker ahead of the tune (quisk.c:2448-2449); created by the first non-zero level
    int nb_level = 0;
    double2 *nb_buf = nullptr;
    long long nb_cap = 0;

    ~Qrx()
    {
        if (agc) qh_qagc_destroy(agc);
        if (nb) qh_nb_destroy(nb);
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (Step &s : steps) {
            if (s.st) { s.st->destroy(); delete s.st; }
            if (s.rat) qh_rat_destroy(s.rat);
        }
        (void)hipFree(dc_state); (void)hipFree(fm_state); (void)hipFree(fm_state_new); (void)hipFree(fm_part); (void)hipFree(buf[0]); (void)hipFree(buf[1]);
        (void)hipFree(sq_state); (void)hipFree(sq_level); (void)hipFree(nb_buf); (void)hipFree(notch_state); (void)hipFree(tw2048);
        (void)hipFree(ssq_state); (void)hipFree(ssq_ring); (void)hipFree(ssq_delay[0]); (void)hipFree(ssq_delay[1]);
        for (Stage *p : rxf_more) { p->destroy(); delete p; }
        (void)hipFree(rx_dly[0]); (void)hipFree(rx_dly[1]); (void)hipFree(rx_x); (void)hipFree(rx_tmp);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    // greedy grouping of consecutive FIR decimators into equivalent ones of at most kMaxEqTaps taps
    int add_groups(const std::vector<FirStageSpec> &st, bool nco, bool even_if_empty)
    {
        if (st.empty() && !even_if_empty) return QH_OK;
        std::vector<FirStageSpec> groups;
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
        groups.push_back({ heq, deq });         // possibly the identity: it still carries the NCO
        for (size_t g = 0; g < groups.size(); g++) {
            Step step;
            step.kind = Step::FIR;
            step.st = new Stage();
            steps.push_back(step);
            if (int rc = step.st->init(device, nch, (int)groups[g].h.size(), groups[g].decim, 1, QH_F64, nco && g == 0, false, false, stream)) return rc;
            std::vector<cd> taps(groups[g].h.size());
            for (size_t i = 0; i < taps.size(); i++) taps[i] = cd(groups[g].h[i], 0.0);
            if (int rc = step.st->set_taps(-1, taps)) return rc;
        }
        return QH_OK;
    }

    int add_rat(const double *taps, int ntaps, int interp, int decim)
    {
        Step step;
        step.kind = Step::RAT;
        step.rat = qh_rat_create(device, nch, taps, (ntaps / interp) * interp, interp, decim, QH_F64, stream);   // filter.c:308
        if (!step.rat) return QH_ERR_HIP;
        steps.push_back(step);
        return QH_OK;
    }

    // The Rx filter's history of channel c, ages 1 (the newest sample) .. rx_hist_len(): the first partition's history rows when the
    // filter fits one partition; else the delay line (ages 1 .. D) and the last partition's rows behind it.
    int rx_hist_len() const { return rx_D() + rxf->hist_len; }
    int rx_hist_read(int c, std::vector<cd> &v)         // v[age - 1]
    {
        const int D = rx_D(), H = rxf->hist_len;
        v.assign((size_t)(D + H), cd(0.0, 0.0));
        QH_HIP(hipStreamSynchronize(stream));
        std::vector<cd> row((size_t)(D > H ? D : H));
        if (D > 0) {
            QH_HIP(hipMemcpy(row.data(), rx_dly[rx_dly_cur] + (size_t)c * D, (size_t)D * sizeof(cd), hipMemcpyDeviceToHost));
            for (int a = 1; a <= D; a++) v[(size_t)a - 1] = row[(size_t)(D - a)];
        }
        Stage *lastp = rxf_more.empty() ? rxf : rxf_more.back();
        QH_HIP(hipMemcpy(row.data(), static_cast<char *>(lastp->hist[lastp->cur]) + (size_t)c * H * sizeof(cd), (size_t)H * sizeof(cd), hipMemcpyDeviceToHost));
        for (int a = 1; a <= H; a++) v[(size_t)(D + a) - 1] = row[(size_t)(H - a)];
        return QH_OK;
    }
    int rx_hist_write(int c, const std::vector<cd> &v)
    {
        const int D = rx_D(), H = rxf->hist_len;
        QH_HIP(hipStreamSynchronize(stream));
        std::vector<cd> row((size_t)(D > H ? D : H));
        auto at = [&](int age) { return age >= 1 && (size_t)age <= v.size() ? v[(size_t)age - 1] : cd(0.0, 0.0); };
        if (D > 0) {
            for (int a = 1; a <= D; a++) row[(size_t)(D - a)] = at(a);
            QH_HIP(hipMemcpy(rx_dly[rx_dly_cur] + (size_t)c * D, row.data(), (size_t)D * sizeof(cd), hipMemcpyHostToDevice));
        }
        for (size_t k = 0; k <= rxf_more.size(); k++) {     // partition k sees the stream delayed by kRxPart k
            Stage *p = k == 0 ? rxf : rxf_more[k - 1];
            for (int a = 1; a <= H; a++) row[(size_t)(H - a)] = at((int)k * kRxPart + a);
            QH_HIP(hipMemcpy(static_cast<char *>(p->hist[p->cur]) + (size_t)c * H * sizeof(cd), row.data(), (size_t)H * sizeof(cd), hipMemcpyHostToDevice));
        }
        return QH_OK;
    }
    // partitions for filters of up to `size` taps: Stages like the first one (per-channel taps, the same output matrix), the delay line
    int rx_ensure_parts(int size)
    {
        const int K = size > kRxPart ? (size + kRxPart - 1) / kRxPart : 1;
        if ((int)rxf_more.size() + 1 >= K) return QH_OK;
        QH_HIP(hipSetDevice(device));
        // what the receivers hold so far, to be laid out again over the longer delay line
        std::vector<std::vector<cd>> keep((size_t)nch);
        for (int c = 0; c < nch; c++) if (int rc = rx_hist_read(c, keep[(size_t)c])) return rc;
        while ((int)rxf_more.size() + 1 < K) {
            Stage *p = new Stage();
            rxf_more.push_back(p);
            if (int rc = p->init(device, nch, kRxPart, 1, 1, QH_F64, false, true, rx_sideband_epi, stream)) return rc;
            if (int rc = p->set_taps(-1, std::vector<cd>(1, cd(0.0, 0.0)))) return rc;
            if (rx_sideband_epi)
                for (int c = 0; c < nch; c++)
                    if (int rc = p->set_epi(c, rx_direct_out ? EpiParam{ 1, 0, 1, 0 } : EpiParam{ 1, 0, 0, 0 })) return rc;
        }
        const int D = rx_D();
        for (int i = 0; i < 2; i++) {
            (void)hipFree(rx_dly[i]); rx_dly[i] = nullptr;
            QH_HIP(hipMalloc((void **)&rx_dly[i], (size_t)nch * (size_t)D * sizeof(cd)));
            QH_HIP(qh::dev_zero(rx_dly[i], (size_t)nch * (size_t)D * sizeof(cd)));
        }
        for (int c = 0; c < nch; c++) if (int rc = rx_hist_write(c, keep[(size_t)c])) return rc;
        return QH_OK;
    }

    // The reference's ring of N = rx_size entries (position p was written when indexFilter was p; the tap loop reads
    // (index + k) mod N) becomes a ring of M entries over the same storage.
    int rx_resize(int c, int M)
    {
        const int N = rx_size[(size_t)c], H = rx_hist_len();
        QH_HIP(hipSetDevice(device));
        std::vector<cd> hist;
        std::vector<cd> &ring = rx_ring[(size_t)c];
        if ((int)ring.size() < (N > M ? N : M)) ring.resize((size_t)(N > M ? N : M), cd(0.0, 0.0));
        if (rx_ring_fresh.size() != rx_size.size()) rx_ring_fresh.assign(rx_size.size(), 0);
        // (only when samples went through since the storage was last brought up to date: a second set_filters ahead of the next block finds
        // the storage as the first one left it, and indexFilter possibly still beyond the size in between -- it is clamped by the next
        // SAMPLE, quisk.c:1246 -- which as a "last written" position would scatter the history over the ring: bank walk 920306)
        if (N > 0 && !rx_ring_fresh[(size_t)c]) {
            if (int rc = rx_hist_read(c, hist)) return rc;
            // what the live ring holds: the sample written `age` calls ago sits at (last - age) mod N, last = index - 1
            const int last = rx_index[(size_t)c] - 1;
            if (last >= 0)
                for (int age = 0; age < N; age++) {
                    const int p = ((last - age) % N + N) % N;
                    ring[(size_t)p] = age < H ? hist[(size_t)age] : cd(0.0, 0.0);      // (the oldest sample of a ring as long as the history is not kept)
                }
        }
        if (M > 0) {
            // the next sample is written at w; tap k >= 1 reads (w + k) mod M = the history sample of age M - k
            const int i = rx_index[(size_t)c], w = i >= M ? 0 : i;
            hist.assign((size_t)H, cd(0.0, 0.0));
            for (int d = 1; d < M && d <= H; d++) hist[(size_t)d - 1] = ring[(size_t)(((w - d) % M + M) % M)];
            if (int rc = rx_hist_write(c, hist)) return rc;
        }
        rx_size[(size_t)c] = M;
        rx_ring_fresh[(size_t)c] = 1;
        return QH_OK;
    }
    // indexFilter after m more samples: "if (index >= size) index = 0; ...; index++" per sample
    void rx_advance(int m)
    {
        for (size_t c = 0; c < rx_size.size(); c++) {
            const int N = rx_size[c];
            if (N <= 0 || m <= 0) continue;
            if (c < rx_ring_fresh.size()) rx_ring_fresh[c] = 0;
            const int i0 = rx_index[c] >= N ? 0 : rx_index[c];
            rx_index[c] = (int)(((long long)i0 + m - 1) % N) + 1;
        }
    }

    cd rx_identity() const      // sizeFilter == 0: c/dRxFilterOut return the sample itself (quisk.c:1201,1239)
    {
        if (!sideband(mode)) return cd(1.0, 0.0);
        return lower(mode) ? cd(1.0, -1.0) : cd(1.0, 1.0);
    }
};

}  // namespace qh

using namespace qh;
// one lock per bank: setters (GUI thread) against the thread that runs the blocks
struct qh_qrx { Qrx q; std::recursive_mutex mtx; };
#define QH_QRX_LOCK(h) std::unique_lock<std::recursive_mutex> _lk; if (h) _lk = std::unique_lock<std::recursive_mutex>((h)->mtx)

extern "C" {

qh_qrx *qh_qrx_create_ex(int device, int nch, int sample_rate, int mode, int bandwidth, const qh_qrx_tables *t, void *stream)
{
    if (nch <= 0 || sample_rate <= 0 || mode < Q_CWL || mode > Q_DGT_FM || !t || !t->f48dec24 || !t->f144d3 || !t->f240d5 ||
        !t->audio24p4 || !t->audio24p6 || !t->lp48 || !t->fmhp) {
        set_error(QH_ERR_INVALID, "qh_qrx_create: bad arguments");
        return nullptr;
    }
    if (mode == Q_EXT) {
        set_error(QH_ERR_UNSUPPORTED, "mode EXT hands the samples to a user plugin (quisk_extern_demod, quisk.c:2490) and has no GPU form");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_qrx *h = new qh_qrx();
    Qrx &q = h->q;
    q.device = device; q.nch = nch; q.sample_rate = sample_rate; q.mode = mode; q.bandwidth = bandwidth; q.ssb_sq_bandwidth = bandwidth;
    auto fail = [&]() -> qh_qrx * { delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); return fail(); }
    q.stream = (hipStream_t)stream;
    if (!q.stream) {
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess) { set_error(QH_ERR_HIP, "stream creation failed"); return fail(); }
        q.own_stream = true;
    }
    // ---- quisk_process_decimate (quisk.c:1729-1843)
    std::vector<FirStageSpec> st;
    const std::vector<double> hb = hb45_dec_taps();
    const std::vector<double> v48(t->f48dec24, t->f48dec24 + 98), v3(t->f144d3, t->f144d3 + 147), v5(t->f240d5, t->f240d5 + 245);
    bool rational = false;
    auto need = [&](const double *p, const char *name) {
        if (!p) set_error(QH_ERR_INVALID, "sample rate %d needs the table %s", sample_rate, name);
        return p != nullptr;
    };
    switch ((sample_rate + 100) / 1000) {
    case 41: q.decim_srate = 48000; break;
    case 53:
        if (!need(t->sdriq53, "quiskFilt53D1Coefs")) return fail();
        q.decim_srate = sample_rate; st.push_back({ std::vector<double>(t->sdriq53, t->sdriq53 + 55), 1 }); break;
    case 111:
        if (!need(t->sdriq111, "quiskFilt111D2Coefs")) return fail();
        q.decim_srate = sample_rate / 2; st.push_back({ std::vector<double>(t->sdriq111, t->sdriq111 + 114), 2 }); break;
    case 133:
        if (!need(t->sdriq133, "quiskFilt133D2Coefs")) return fail();
        q.decim_srate = sample_rate / 2; st.push_back({ std::vector<double>(t->sdriq133, t->sdriq133 + 136), 2 }); break;
    case 185: case 370: case 740:
        if (!need(t->sdriq185, "quiskFilt185D3Coefs")) return fail();
        q.decim_srate = sample_rate / 3;
        for (int k = (sample_rate + 100) / 1000; k > 185; k /= 2) { st.push_back({ hb, 2 }); q.decim_srate /= 2; }
        st.push_back({ std::vector<double>(t->sdriq185, t->sdriq185 + 189), 3 }); break;
    case 1333:
        if (!need(t->sdriq167, "quiskFilt167D3Coefs")) return fail();
        q.decim_srate = sample_rate / 24;
        for (int k = 0; k < 3; k++) st.push_back({ hb, 2 });
        st.push_back({ std::vector<double>(t->sdriq167, t->sdriq167 + 174), 3 }); break;
    default: {
        // PlanDecimation, quisk.c:1633-1671
        int best = sample_rate, d2 = 0, d3 = 0, d5 = 0;
        for (int i2 = 0; i2 <= 6; i2++)
            for (int i3 = 0; i3 <= 3; i3++)
                for (int i5 = 0; i5 <= 3; i5++) {
                    int r = sample_rate;
                    for (int i = 0; i < i2; i++) r /= 2;
                    for (int i = 0; i < i3; i++) r /= 3;
                    for (int i = 0; i < i5; i++) r /= 5;
                    if (r >= 48000 && r < best) { d2 = i2; d3 = i3; d5 = i5; best = r; }
                }
        q.decim_srate = best;
        for (int i = 0; i < d2 - 1 && i < 5; i++) st.push_back({ hb, 2 });
        for (int i = 0; i < d3; i++) st.push_back({ v3, 3 });
        for (int i = 0; i < d5; i++) st.push_back({ v5, 5 });
        if (d2 > 0) st.push_back({ v48, 2 });
        if (q.decim_srate >= 50000) {           // 6/5 then 4/5, quisk.c:1834-1838
            if (!need(t->f300d5, "quiskFilt300D5Coefs")) return fail();
            rational = true;
            q.decim_srate = q.decim_srate * 24 / 25;
        }
        break;
    }
    }
    // ---- the mode's own decimators in front of the Rx filter (quisk.c:1906-2030,2087-2126)
    std::vector<FirStageSpec> fe;
    const bool dgt_narrow = is_dgt(mode) && bandwidth < kDgtNarrowFreq;
    if (is_cw(mode) || dgt_narrow) { q.filter_srate = q.decim_srate / 8; fe.push_back({ hb, 2 }); fe.push_back({ hb, 2 }); fe.push_back({ v48, 2 }); }
    else if (is_ssb(mode))         { q.filter_srate = q.decim_srate / 4; fe.push_back({ hb, 2 }); fe.push_back({ v48, 2 }); }
    else if (is_am(mode))          { q.filter_srate = q.decim_srate / 2; fe.push_back({ v48, 2 }); }
    else                           { q.filter_srate = q.decim_srate; }
    if (rational) {
        if (q.add_groups(st, true, true)) return fail();
        if (q.add_rat(t->f300d5, 125, 6, 5) || q.add_rat(t->f240d5, 245, 4, 5)) return fail();
        if (q.add_groups(fe, false, false)) return fail();
    } else {
        st.insert(st.end(), fe.begin(), fe.end());
        if (q.add_groups(st, true, true)) return fail();
    }
    bool notch_after_rxf = false;
    // ---- Rx filter: per-channel taps, up to 2048; identity until set_filters is called (sizeFilter == 0)
    const bool direct_out = (is_dgt(mode) && !dgt_narrow) || is_iq(mode);       // nothing follows the Rx filter
    {
        Step step;
        step.kind = Step::FIR;
        step.st = q.rxf = new Stage();
        q.steps.push_back(step);
        q.rx_sideband_epi = sideband(mode); q.rx_direct_out = direct_out;
        if (q.rxf->init(device, nch, Qrx::kRxPart, 1, 1, QH_F64, false, true, sideband(mode), q.stream)) return fail();
        if (q.rxf->set_taps(-1, std::vector<cd>(1, q.rx_identity()))) return fail();
        if (sideband(mode) && !is_iq(mode)) notch_after_rxf = true;     // quisk.c:1923,1946,1968,1992,2106,2133 (not DGT-IQ)
        if (sideband(mode))
            for (int c = 0; c < nch; c++)       // re -+ im collapses to the real part; (d, d) when it is the last stage (quisk.c:2625)
                if (q.rxf->set_epi(c, direct_out ? EpiParam{ 1, 0, 1, 0 } : EpiParam{ 1, 0, 0, 0 })) return fail();
    }
    if (notch_after_rxf) { Step nt; nt.kind = Step::NOTCH; nt.dup = direct_out; q.steps.push_back(nt); }
    // ---- detector and the way back to decim_srate
    const std::vector<double> g45 = hb45_interp_taps();
    std::vector<double> ueq;
    int U = 1;
    if (is_cw(mode) || dgt_narrow) {                // dInterpolate(Audio24p4 table, 2), HB45, HB45 (quisk.c:1930-1932,2108-2112)
        ueq = conv(conv(upsample(dinterp_taps(t->audio24p4, 50, 2), 4), upsample(g45, 2)), g45); U = 8;
    } else if (is_ssb(mode)) {                      // dInterpolate(Audio24p4, 2), HB45 (quisk.c:1975-1976)
        ueq = conv(upsample(dinterp_taps(t->audio24p4, 50, 2), 2), g45); U = 4;
    } else if (is_am(mode)) {                       // dFilter(Audio24p6), HB45 (quisk.c:2017,2024)
        Step det; det.kind = Step::AM_ENV; q.steps.push_back(det);
        // dFilter(Audio24p6) is its own stage: ssb_squelch and d_delay sit between it and the interpolator (quisk.c:2017-2024)
        std::vector<FirStageSpec> a6;
        a6.push_back({ std::vector<double>(t->audio24p6, t->audio24p6 + 36), 1 });
        if (q.add_groups(a6, false, false)) return fail();
        if (q.steps.back().st->set_pair(2)) return fail();                                  // real audio, real taps: (d, d) rows, two receivers per tile
        { Step nt; nt.kind = Step::NOTCH; q.steps.push_back(nt); }                          // quisk.c:2018-2019
        ueq = g45; U = 2;
    } else if (is_fm(mode)) {                       // HB45, HB45 after the /4 (quisk.c:2067-2068)
        Step det; det.kind = Step::FM_DISC; q.steps.push_back(det);
        ueq = conv(upsample(g45, 2), g45); U = 4;
        // dDecimate(LpFilt48, 4) then dFilter(AudioFmHp) (quisk.c:2065-2066) as one decimator
        std::vector<FirStageSpec> post;
        post.push_back({ std::vector<double>(t->lp48, t->lp48 + 186), 4 });
        post.push_back({ std::vector<double>(t->fmhp, t->fmhp + 309), 1 });
        if (q.add_groups(post, false, false)) return fail();
        for (size_t k = q.steps.size(); k-- > 0 && q.steps[k].kind == Step::FIR;)          // (the group may have been cut in two)
            if (q.steps[k].st->set_pair(1)) return fail();                                  // the discriminator's rows are (y, 0)
    }
    if ((is_cw(mode) || is_ssb(mode) || is_am(mode)) && !dgt_narrow) {
        // ssb_squelch + d_delay (quisk.c:1925-1928,1970-1973,2020-2023): present in the step list, idle until enabled
        Step a; a.kind = Step::SSB_SQ; q.steps.push_back(a);
        Step d; d.kind = Step::DELAY; q.steps.push_back(d);
        q.has_ssb_sq = true;
    }
    if (U > 1) {
        Step step;
        step.kind = Step::FIR;
        step.st = new Stage();
        q.steps.push_back(step);
        if (step.st->init(device, nch, (int)ueq.size(), 1, U, QH_F64, false, false, true, q.stream)) return fail();
        std::vector<cd> taps(ueq.size());
        for (size_t i = 0; i < taps.size(); i++) taps[i] = cd(ueq[i], 0.0);
        if (step.st->set_taps(-1, taps)) return fail();
        for (int c = 0; c < nch; c++)
            if (step.st->set_epi(c, EpiParam{ 1, 0, 1, 0 })) return fail();                 // d + I*d, quisk.c:2625
        if (step.st->set_pair(1)) return fail();                                            // only the real part counts: two receivers per tile
    }
    if (is_fm(mode)) { Step nt; nt.kind = Step::NOTCH; nt.dup = true; q.steps.push_back(nt); }     // after the interpolators, quisk.c:2069-2070
    if (is_am(mode)) {
        if (hipMalloc((void **)&q.dc_state, (size_t)nch * 8) != hipSuccess || qh::dev_zero(q.dc_state, (size_t)nch * 8) != hipSuccess) {
            set_error(QH_ERR_HIP, "allocation failed"); return fail();
        }
    }
    if (is_fm(mode)) {
        std::vector<double4> init((size_t)nch, make_double4(10.0, 0.0, 0.0, 0.0));          // fm_1 = 10, quisk.c:1893
        if (hipMalloc((void **)&q.fm_state, (size_t)nch * sizeof(double4)) != hipSuccess ||
            hipMalloc((void **)&q.fm_state_new, (size_t)nch * sizeof(double4)) != hipSuccess ||
            hipMemcpy(q.fm_state, init.data(), (size_t)nch * sizeof(double4), hipMemcpyHostToDevice) != hipSuccess) {
            set_error(QH_ERR_HIP, "allocation failed"); return fail();
        }
        q.h_sq_level.assign((size_t)nch, -999.0);                                           // squelch_level, quisk.c:193
        if (hipMalloc((void **)&q.sq_state, (size_t)nch * sizeof(QSquelchState)) != hipSuccess ||
            qh::dev_zero(q.sq_state, (size_t)nch * sizeof(QSquelchState)) != hipSuccess ||
            hipMalloc((void **)&q.sq_level, (size_t)nch * 8) != hipSuccess ||
            hipMemcpy(q.sq_level, q.h_sq_level.data(), (size_t)nch * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_error(QH_ERR_HIP, "allocation failed"); return fail();
        }
        const double www = std::tan(M_PI * 300.0 / 48000);                                  // quisk.c:1894-1898
        const double nnn = 1.0 / (1.0 + www);
        q.fm_prm.a0 = www * nnn; q.fm_prm.a1 = q.fm_prm.a0; q.fm_prm.b1 = nnn * (www - 1.0);
    }
    return h;
}

qh_qrx *qh_qrx_create(int device, int nch, int sample_rate, int mode, const double *f48dec24, const double *f144d3,
                      const double *f240d5, const double *audio24p4, const double *audio24p6, const double *lp48,
                      const double *fmhp, void *stream)
{
    qh_qrx_tables t{};
    t.f48dec24 = f48dec24; t.f144d3 = f144d3; t.f240d5 = f240d5; t.audio24p4 = audio24p4; t.audio24p6 = audio24p6;
    t.lp48 = lp48; t.fmhp = fmhp;
    return qh_qrx_create_ex(device, nch, sample_rate, mode, 2700, &t, stream);
}

void qh_qrx_destroy(qh_qrx *h) { delete h; }
int qh_qrx_filter_rate(const qh_qrx *h) { return h ? h->q.filter_srate : 0; }

// set_tune (quisk.c:4702): the stream is multiplied by exp(-j 2 pi tune n / sample_rate)
int qh_qrx_set_tune(qh_qrx *h, int ch, int rx_tune_freq)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (ch < -1 || ch >= q.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? q.nch : ch + 1); c++)
        if (int rc = q.steps[0].st->set_nco(c, -(double)rx_tune_freq, (double)q.sample_rate)) return rc;
    return QH_OK;
}

// set_tune for every receiver of the bank at once: rx_tune_freq[nch] (Hz), one launch per table instead of one per receiver
int qh_qrx_set_tune_all(qh_qrx *h, const int *rx_tune_freq)
{
    QH_QRX_LOCK(h);
    if (!h || !rx_tune_freq) return set_error(QH_ERR_INVALID, "qh_qrx_set_tune_all: bad arguments");
    Qrx &q = h->q;
    std::vector<double> f((size_t)q.nch);
    for (int c = 0; c < q.nch; c++) f[(size_t)c] = -(double)rx_tune_freq[c];
    return q.steps[0].st->set_nco_all(f.data(), (double)q.sample_rate);
}

// set_filters (quisk.c:4551): taps as MakeFilterCoef designs them.  cRxFilterOut's ring walk (quisk.c:1246-1253)
// pairs tap 0 with the newest sample and taps 1..N-1 with the oldest..second newest: as a convolution
// g[0] = h[0], g[d] = h[N-d].  SSB/CW keep re -+ im = Re{(gI +- j gQ) * x}; AM/FM use filtI on both parts.
int qh_qrx_set_filters(qh_qrx *h, int ch, const double *filtI, const double *filtQ, int size)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (ch < -1 || ch >= q.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    if (size < 0 || size > Qrx::kRxMaxTaps || (size > 0 && (!filtI || !filtQ)))
        return set_error(QH_ERR_INVALID, "Filter size must be less than 10001 (MAX_FILTER_SIZE, quisk.c:4575; got %d)", size);
    const bool bypass = is_iq(q.mode) && q.bandwidth >= 19000;        // "No filtering for wide bandwidth", quisk.c:2143
    if (bypass) size = 0;
    std::vector<cd> g((size_t)(size > 0 ? size : 1), q.rx_identity());
    for (int d = 0; d < size; d++) {
        const int k = d == 0 ? 0 : size - d;
        const double gi = filtI[k], gq = filtQ[k];
        if (!sideband(q.mode)) g[(size_t)d] = cd(gi, 0.0);              // dRxFilterOut
        else g[(size_t)d] = lower(q.mode) ? cd(gi, -gq) : cd(gi, gq);   // re + im : re - im
    }
    if (q.rx_size.empty()) { q.rx_size.assign((size_t)q.nch, 0); q.rx_index.assign((size_t)q.nch, 0); q.rx_ring.resize((size_t)q.nch); }
    if (int rc = q.rx_ensure_parts(size)) return rc;
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? q.nch : ch + 1); c++) {
        for (size_t k = 0; k <= q.rxf_more.size(); k++) {           // partition k: taps [kRxPart k, kRxPart (k + 1))
            Stage *p = k == 0 ? q.rxf : q.rxf_more[k - 1];
            const size_t lo = k * (size_t)Qrx::kRxPart, hi = lo + (size_t)Qrx::kRxPart < g.size() ? lo + (size_t)Qrx::kRxPart : g.size();
            std::vector<cd> part = lo < g.size() ? std::vector<cd>(g.begin() + (long)lo, g.begin() + (long)hi) : std::vector<cd>(1, cd(0.0, 0.0));
            if (int rc = p->set_taps(c, part)) return rc;
        }
        if (size != q.rx_size[(size_t)c]) if (int rc = q.rx_resize(c, size)) return rc;
    }
    return QH_OK;
}

int qh_qrx_out_count(const qh_qrx *h, int n_in)
{
    if (!h) return 0;
    int n = n_in;
    for (const Step &s : h->q.steps) {
        if (s.kind == Step::FIR) n = s.st->out_count(n);
        else if (s.kind == Step::RAT) n = qh_rat_out_count(s.rat, n);
    }
    return n;
}

int qh_qrx_decim_rate(const qh_qrx *h) { return h ? h->q.decim_srate : 0; }

int qh_qrx_process(qh_qrx *h, const double *d_in, long long in_stride, int n_in, double *d_out, long long out_stride, int *n_out)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    if (n_out) *n_out = 0;
    if (n_in <= 0) return QH_OK;
    if (!d_in || !d_out || in_stride < n_in) return set_error(QH_ERR_INVALID, "bad buffers");
    Qrx &q = h->q;
    QH_HIP(hipSetDevice(q.device));
    const int total = qh_qrx_out_count(h, n_in);
    if (out_stride < total) return set_error(QH_ERR_INVALID, "output stride %lld shorter than %d samples", out_stride, total);
    // intermediate buffers: the 6/5 stage is the only one that grows the count before the last step
    const long long need = (long long)(n_in > total ? n_in : total) * 5 / 4 + 64;
    if (need > q.buf_cap) {
        QH_HIP(hipStreamSynchronize(q.stream));
        for (int i = 0; i < 2; i++) { (void)hipFree(q.buf[i]); q.buf[i] = nullptr; }
        for (int i = 0; i < 2; i++) QH_HIP(hipMalloc((void **)&q.buf[i], (size_t)q.nch * (size_t)need * 16));
        q.buf_cap = need;
    }
    const void *cur = d_in;
    long long cur_stride = in_stride;
    if (q.nb && q.nb_level > 0) {
        // NoiseBlanker(cSamples, nSamples) on the raw samples, quisk.c:2448-2449
        if (n_in > q.nb_cap) {
            QH_HIP(hipStreamSynchronize(q.stream));
            (void)hipFree(q.nb_buf); q.nb_buf = nullptr;
            QH_HIP(hipMalloc((void **)&q.nb_buf, (size_t)q.nch * (size_t)n_in * 16));
            q.nb_cap = n_in;
        }
        if (int rc = qh_nb_process(q.nb, d_in, in_stride, q.nb_buf, q.nb_cap, n_in)) return rc;
        cur = q.nb_buf; cur_stride = q.nb_cap;
    }
    int n = n_in, w = 0;
    size_t last = 0;
    for (size_t i = 0; i < q.steps.size(); i++)
        if (q.steps[i].kind == Step::FIR || q.steps[i].kind == Step::RAT) last = i;     // (a DELAY is always followed by the interpolator)
    for (size_t i = 0; i < q.steps.size(); i++) {
        const Step &s = q.steps[i];
        void *dst = i == last ? (void *)d_out : (void *)q.buf[w];
        const long long dst_stride = i == last ? out_stride : q.buf_cap;
        int m = 0;
        switch (s.kind) {
        case Step::FIR:
            if (s.st == q.rxf && !q.rxf_more.empty() && n > 0) {
                // [delay line | block]; partition k filters the block as it was kRxPart k samples ago, the outputs add up
                const int D = q.rx_D();
                const long long xs = (long long)D + n;
                if ((long long)q.nch * xs > q.rx_x_cap) {
                    QH_HIP(hipStreamSynchronize(q.stream));
                    (void)hipFree(q.rx_x); (void)hipFree(q.rx_tmp); q.rx_x = q.rx_tmp = nullptr;
                    q.rx_x_cap = (long long)q.nch * xs * 5 / 4;
                    QH_HIP(hipMalloc((void **)&q.rx_x, (size_t)q.rx_x_cap * 16));
                    QH_HIP(hipMalloc((void **)&q.rx_tmp, (size_t)q.rx_x_cap * 16));
                }
                const unsigned gx = (unsigned)((xs + 255) / 256 < 512 ? (xs + 255) / 256 : 512);
                hipLaunchKernelGGL(rxp_gather_kernel, dim3(gx, (unsigned)q.nch), dim3(256), 0, q.stream, (const double2 *)q.rx_dly[q.rx_dly_cur], D,
                                   static_cast<const double2 *>(cur), cur_stride, n, q.rx_x, xs);
                if (int rc = q.rxf->process(q.rx_x + D, xs, n, dst, dst_stride, &m)) return rc;
                for (size_t k = 0; k < q.rxf_more.size(); k++) {
                    int mk = 0;
                    if (int rc = q.rxf_more[k]->process(q.rx_x + D - (long long)(k + 1) * Qrx::kRxPart, xs, n, q.rx_tmp, (long long)n, &mk)) return rc;
                    hipLaunchKernelGGL(rxp_add_kernel, dim3(gx, (unsigned)q.nch), dim3(256), 0, q.stream, static_cast<double2 *>(dst), dst_stride,
                                       (const double2 *)q.rx_tmp, (long long)n, n);
                }
                hipLaunchKernelGGL(rxp_keep_kernel, dim3(gx, (unsigned)q.nch), dim3(256), 0, q.stream, (const double2 *)q.rx_x, xs, D, n, q.rx_dly[q.rx_dly_cur ^ 1]);
                q.rx_dly_cur ^= 1;
                q.rx_advance(n);
                break;
            }
            if (int rc = s.st->process(cur, cur_stride, n, dst, dst_stride, &m)) return rc;
            if (s.st == q.rxf) q.rx_advance(n);
            break;
        case Step::RAT:
            if (int rc = qh_rat_process(s.rat, cur, cur_stride, n, dst, dst_stride, &m)) return rc;
            break;
        case Step::AM_ENV:
            if (n >= kQTiledMin)
                hipLaunchKernelGGL(q_am_env_tiled_kernel, dim3((unsigned)q.nch), dim3(kSegThreads), 0, q.stream,
                                   const_cast<double2 *>(static_cast<const double2 *>(cur)), cur_stride, n, q.dc_state);
            else if (n > 0)
                hipLaunchKernelGGL(q_am_env_kernel, dim3((unsigned)q.nch), dim3(64), 0, q.stream,
                                   const_cast<double2 *>(static_cast<const double2 *>(cur)), cur_stride, n, q.dc_state);
            continue;
        case Step::SSB_SQ:
            if (q.ssb_sq_on && n > 0) {
                if (!q.ssb_sq_inited) { q.ssb_sq_inited = true; continue; }      // "if (!plan) { ...; return; }", quisk.c:1104-1112
                int bw = q.ssb_sq_bandwidth > 3000 ? 3000 : q.ssb_sq_bandwidth;
                QSsbSqParam sp;
                sp.samp_rate = q.filter_srate;
                sp.bw1 = 300 * 512 / q.filter_srate;
                sp.bw2 = (bw + 300) * 512 / q.filter_srate;
                if (sp.bw2 > 257) sp.bw2 = 257;                                     // out_fft holds N/2 + 1 bins
                sp.thresh = q.ssb_sq_level * 0.005;
                hipLaunchKernelGGL(q_ssb_squelch_kernel, dim3((unsigned)q.nch), dim3(256), 0, q.stream, static_cast<const double2 *>(cur),
                                   cur_stride, n, q.ssq_state, q.ssq_ring, q.sq_state, sp);
            }
            continue;
        case Step::NOTCH:
            if (q.notch_on && n > 0) {
                constexpr size_t lds = TileFft<2048, false, double2>::kLdsBytes;
                hipLaunchKernelGGL(q_autonotch_kernel, dim3((unsigned)q.nch), dim3(256), lds, q.stream, const_cast<double2 *>(static_cast<const double2 *>(cur)), cur_stride, n, q.notch_state,
                                   q.tw2048, is_cw(q.mode) ? q.rit_freq : 0, q.filter_srate, s.dup ? 1 : 0);
            }
            continue;
        case Step::DELAY:
            if (!q.ssb_sq_on || n <= 0) continue;
            {
                int gx = (n + 512 + 255) / 256;
                if (gx > 256) gx = 256;
                hipLaunchKernelGGL(q_delay_kernel, dim3((unsigned)gx, (unsigned)q.nch), dim3(256), 0, q.stream,
                                   static_cast<const double2 *>(cur), cur_stride, static_cast<double2 *>(dst), dst_stride, n,
                                   q.ssq_delay[q.ssq_cur], q.ssq_delay[q.ssq_cur ^ 1]);
                q.ssq_cur ^= 1;
                m = n;
            }
            break;
        case Step::FM_DISC:
            if (n > 0) {
                if (q.sq_dirty) {
                    QH_HIP(hipMemcpyAsync(q.sq_level, q.h_sq_level.data(), (size_t)q.nch * 8, hipMemcpyHostToDevice, q.stream));
                    QH_HIP(hipStreamSynchronize(q.stream));
                    q.sq_dirty = false;
                }
                if (n < kQTiledMin)      // long calls: the sum rides on the tiled detector below
                    hipLaunchKernelGGL(q_fm_squelch_kernel, dim3((unsigned)q.nch), dim3(64), 0, q.stream,
                                       static_cast<const double2 *>(cur), cur_stride, n, q.sq_state, q.sq_level, q.sq_defer ? 1 : 0);
            }
            if (n >= kQTiledMin) {       // long calls: the detector over a grid of time segments (qh_qdemod.hpp), into the other buffer
                const int tail_b = seg_tail_batches(-q.fm_prm.b1);
                const int seg_b = n >= (1 << 17) ? 128 : 64;
                const int nseg = (n + seg_b * 64 - 1) / (seg_b * 64);
                if ((long long)q.nch * nseg > q.fm_part_cap) {
                    QH_HIP(hipStreamSynchronize(q.stream));
                    (void)hipFree(q.fm_part); q.fm_part = nullptr;
                    QH_HIP(hipMalloc((void **)&q.fm_part, (size_t)q.nch * (size_t)nseg * 8));
                    q.fm_part_cap = (long long)q.nch * nseg;
                }
                hipLaunchKernelGGL(q_fm_disc_grid_kernel, dim3((unsigned)((nseg + 3) / 4), (unsigned)q.nch), dim3(256), 0, q.stream,
                                   static_cast<const double2 *>(cur), cur_stride, static_cast<double2 *>(dst), dst_stride, n,
                                   (const double4 *)q.fm_state, q.fm_state_new, q.fm_prm, seg_b, tail_b, q.fm_part, nseg);
                hipLaunchKernelGGL(q_fm_disc_finish_kernel, dim3((unsigned)((q.nch + 63) / 64)), dim3(64), 0, q.stream, q.nch, q.fm_state,
                                   (const double4 *)q.fm_state_new, (const double *)q.fm_part, nseg, n, q.sq_state, (const double *)q.sq_level, q.sq_defer ? 1 : 0);
                m = n;
                break;
            }
            if (n > 0)
                hipLaunchKernelGGL(q_fm_disc_kernel, dim3((unsigned)q.nch), dim3(64), 0, q.stream,
                                   const_cast<double2 *>(static_cast<const double2 *>(cur)), cur_stride, n, q.fm_state, q.fm_prm);
            continue;
        }
        cur = dst; cur_stride = dst_stride; n = m;
        if (i != last) w ^= 1;
    }
    if (q.agc && q.agc_on && n > 0)
        if (int rc = qh_qagc_process(q.agc, d_out, out_stride, n)) return rc;
    if (q.sq_state && !q.mute_deferred && n > 0) {
        int gx = (n + 255) / 256;
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(q_mute_kernel, dim3((unsigned)gx, (unsigned)q.nch), dim3(256), 0, q.stream, reinterpret_cast<double2 *>(d_out),
                           out_stride, n, q.sq_state);
    }
    if (n_out) *n_out = n;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

// set_ssb_squelch(enabled, level) (quisk.c:4729): CW / SSB / AM banks.  Enabling also puts d_delay's 512 samples of
// audio delay in the path, as in the reference.
int qh_qrx_set_ssb_squelch(qh_qrx *h, int enabled, int level)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (!q.has_ssb_sq) return set_error(QH_ERR_UNSUPPORTED, "ssb_squelch belongs to the CW, SSB and AM modes (quisk.c:1925,1970,2020)");
    QH_HIP(hipSetDevice(q.device));
    if (enabled && !q.ssq_state) {
        QH_HIP(hipMalloc((void **)&q.ssq_state, (size_t)q.nch * sizeof(QSsbSqState)));
        QH_HIP(hipMalloc((void **)&q.ssq_ring, (size_t)q.nch * 512 * sizeof(double)));
        for (int i = 0; i < 2; i++) QH_HIP(hipMalloc((void **)&q.ssq_delay[i], (size_t)q.nch * 512 * sizeof(double2)));
        QH_HIP(hipMemsetAsync(q.ssq_state, 0, (size_t)q.nch * sizeof(QSsbSqState), q.stream));
        QH_HIP(hipMemsetAsync(q.ssq_ring, 0, (size_t)q.nch * 512 * sizeof(double), q.stream));
        for (int i = 0; i < 2; i++) QH_HIP(hipMemsetAsync(q.ssq_delay[i], 0, (size_t)q.nch * 512 * sizeof(double2), q.stream));
        if (!q.sq_state) {
            QH_HIP(hipMalloc((void **)&q.sq_state, (size_t)q.nch * sizeof(QSquelchState)));
            QH_HIP(hipMemsetAsync(q.sq_state, 0, (size_t)q.nch * sizeof(QSquelchState), q.stream));
        }
        QH_HIP(hipStreamSynchronize(q.stream));
    }
    if (!enabled && q.ssb_sq_on && q.sq_state)          // MeasureSquelch[bank].squelch_active = 0 at the top of every call, quisk.c:1908
        QH_HIP(hipMemsetAsync(q.sq_state, 0, (size_t)q.nch * sizeof(QSquelchState), q.stream));
    q.ssb_sq_on = enabled != 0;
    q.ssb_sq_level = level;
    return QH_OK;
}

// The FM squelch's level is looked at once per call of quisk_process_samples (quisk.c:2076-2085): a caller that hands one such call over in
// PIECES (qh_qps.hip) says so -- pieces = 1: the following process calls only add to the sums -- and closes the call with pieces = 0,
// which takes the look the reference takes.  (The measurement runs whether a threshold is set or not, so the windows must line up even
// while the squelch cannot act.)
int qh_qrx_squelch_pieces(qh_qrx *h, int pieces)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (!q.sq_state || !is_fm(q.mode)) return QH_OK;
    if (!pieces && q.sq_defer) {
        if (q.sq_dirty) {
            QH_HIP(hipMemcpyAsync(q.sq_level, q.h_sq_level.data(), (size_t)q.nch * 8, hipMemcpyHostToDevice, q.stream));
            QH_HIP(hipStreamSynchronize(q.stream));
            q.sq_dirty = false;
        }
        hipLaunchKernelGGL(q_squelch_close_kernel, dim3((unsigned)((q.nch + 63) / 64)), dim3(64), 0, q.stream, q.nch, q.sq_state, (const double *)q.sq_level);
    }
    q.sq_defer = pieces != 0;
    return QH_OK;
}

// set_squelch (quisk.c:4721-4727): the FM squelch threshold in dB re full scale; -999 (the default) never mutes
int qh_qrx_set_squelch(qh_qrx *h, int ch, double level)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    if (ch < -1 || ch >= q.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    if (!is_fm(q.mode)) return set_error(QH_ERR_UNSUPPORTED, "the FM squelch belongs to modes FM and DGT-FM (quisk.c:2026-2085)");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? q.nch : ch + 1); c++) q.h_sq_level[(size_t)c] = level;
    q.sq_dirty = true;
    return QH_OK;
}

// process_agc as quisk_process_samples runs it on the playback stream: Agc1 = {0.7, ...} (quisk.c:2321), release
// time 1.0 s (quisk.c:192), |z| for DGT-IQ and |Re z| otherwise (quisk.c:2686-2702), playback rate = decim rate
int qh_qrx_set_agc(qh_qrx *h, int on, double release_gain)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    q.agc_on = on != 0;
    if (!on) return QH_OK;          // the AGC's state stays as it is (Agc1 is static in the reference) and carries on when switched back on
    if (!q.agc) {
        q.agc = qh_qagc_create(q.device, q.nch, q.decim_srate, 0.7, 1.0, is_iq(q.mode) ? 1 : 0, q.stream);
        if (!q.agc) return QH_ERR_HIP;
    }
    q.agc_gain = release_gain;
    return qh_qagc_set_gain(q.agc, -1, release_gain);
}

// set_auto_notch (quisk.c:4596-4603) with set_sidetone's rit_freq (quisk.c:4712; the CW modes keep the notch off the
// sidetone): stores the flag and starts the notch state over, like dAutoNotch(NULL, 0, 0, 0)
int qh_qrx_set_auto_notch(qh_qrx *h, int on, int rit_freq)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    Qrx &q = h->q;
    bool has = false;
    for (const Step &s : q.steps) has = has || s.kind == Step::NOTCH;
    if (!has) { q.notch_on = false; return on ? set_error(QH_ERR_UNSUPPORTED, "this mode has no auto-notch (DGT-IQ)") : QH_OK; }
    QH_HIP(hipSetDevice(q.device));
    if (!q.notch_state) {
        if (!on) return QH_OK;
        QH_HIP(hipMalloc((void **)&q.notch_state, (size_t)q.nch * sizeof(QNotchState)));
        QH_HIP(hipMemsetAsync(q.notch_state, 0, (size_t)q.nch * sizeof(QNotchState), q.stream));
        const std::vector<cd> tw = fft_twiddle_table(2048);
        QH_HIP(hipMalloc((void **)&q.tw2048, tw.size() * sizeof(cd)));
        QH_HIP(hipMemcpyAsync(q.tw2048, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice, q.stream));
        QH_HIP(hipStreamSynchronize(q.stream));
    }
    hipLaunchKernelGGL(q_autonotch_init_kernel, dim3((unsigned)q.nch), dim3(256), 0, q.stream, q.notch_state);
    QH_HIP(hipGetLastError());
    q.notch_on = on != 0;
    q.rit_freq = rit_freq;
    return QH_OK;
}

// set_noise_blanker (quisk.c:4605): the blanker runs on the raw samples, ahead of the tune
int qh_qrx_set_noise_blanker(qh_qrx *h, int level)
{
    QH_QRX_LOCK(h);
    if (!h || level < 0) return set_error(QH_ERR_INVALID, "qh_qrx_set_noise_blanker: bad arguments");
    Qrx &q = h->q;
    if (!q.nb) {
        if (level == 0) return QH_OK;
        q.nb = qh_nb_create(q.device, q.nch, q.sample_rate, q.stream);
        if (!q.nb) return QH_ERR_HIP;
    }
    q.nb_level = level;
    return qh_nb_set_level(q.nb, level);
}

// ---- hooks for quisk_process_samples' orchestration (qh_quisk_rx_compat.cpp) ------------------------------------------------
// The reference applies the squelch flags of its banks at the very end of the block, behind the interpolation and the AGC
// (quisk.c:2712-2728): the bank leaves its output alone and hands out where its flag lives.
int qh_qrx_set_mute_deferred(qh_qrx *h, int on)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    h->q.mute_deferred = on != 0;
    return QH_OK;
}
// device address of MeasureSquelch[bank].squelch_active of channel ch (an int), or null while the bank has no squelch
const int *qh_qrx_squelch_flag(qh_qrx *h, int ch)
{
    QH_QRX_LOCK(h);
    if (!h || ch < 0 || ch >= h->q.nch || !h->q.sq_state) return nullptr;
    return &h->q.sq_state[ch].active;
}
// ssb_squelch looks at the bins of filter_bandwidth[0] -- the bandwidth set_filters was given for filter set 0 -- in EVERY bank, also in
// one that runs filter set 1 or 2 (quisk.c:1120).  A receiver on its own is bank 0 with set 0: the bandwidth it was created with.
int qh_qrx_set_ssb_squelch_bandwidth(qh_qrx *h, int bandwidth)
{
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    h->q.ssb_sq_bandwidth = bandwidth;
    return QH_OK;
}
// ssb_squelch's FFT plan is ONE function static for all banks (quisk.c:1091,1104): the first call of any bank creates it and
// returns without looking at its samples.  A caller that runs several banks as the reference's bank 0 / 1 / 2 passes the fact on.
int qh_qrx_ssb_squelch_planned(qh_qrx *h, int set)
{
    QH_QRX_LOCK(h);
    if (!h) return 0;
    if (set > 0) h->q.ssb_sq_inited = true;
    return h->q.ssb_sq_inited ? 1 : 0;
}
// the tuning oscillator's phase (2^-64 turns) at the next input sample: one vector per purpose in the reference (quisk.c:2308-2311)
int qh_qrx_get_nco_phase(qh_qrx *h, int ch, unsigned long long *phase)
{
    QH_QRX_LOCK(h);
    if (!h || !phase || ch < 0 || ch >= h->q.nch) return set_error(QH_ERR_INVALID, "qh_qrx_get_nco_phase: bad arguments");
    return h->q.steps[0].st->get_nco_phase(ch, phase);
}
int qh_qrx_set_nco_phase(qh_qrx *h, int ch, unsigned long long phase)
{
    QH_QRX_LOCK(h);
    if (!h || ch < 0 || ch >= h->q.nch) return set_error(QH_ERR_INVALID, "qh_qrx_set_nco_phase: bad arguments");
    return h->q.steps[0].st->set_nco_phase(ch, phase);
}

int qh_qrx_process_host(qh_qrx *h, const double *h_in, long long in_stride, int n_in, double *h_out, long long out_stride, int *n_out)
{
    QH_QRX_LOCK(h);
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
    QH_QRX_LOCK(h);
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    QH_HIP(hipSetDevice(h->q.device));
    QH_HIP(hipStreamSynchronize(h->q.stream));
    return QH_OK;
}

}  // extern "C"
