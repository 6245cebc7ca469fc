// qh_qagc.hip -- Quisk's audio AGC (include/quiskhip.h group 8): process_agc, quisk.c:2162-2287, for `nch` streams.
//
// A 15 ms FIFO (AGC_DELAY, quisk.c:47) delays the audio while a five-branch state machine moves the gain: ramp
// down linearly over the FIFO length when a sample would exceed max_out, otherwise relax exponentially towards
// min(agcReleaseGain, max_out * CLIP32 / largest sample of the last FIFO cycle).  The recurrence is non-linear and
// sequential: one wavefront per stream, lanes hold 64 consecutive samples, every lane steps the same scalar state
// through the 64 magnitudes (broadcast by __shfl) and lane i keeps the gain that applied to sample i; the FIFO
// lives in global memory in the reference's ring order, so a call leaves exactly the reference's state.
#include <cmath>
#include <vector>
#include "qh_internal.hpp"
#include "qh_wave.hpp"

using namespace qh;

namespace {

constexpr double kClip32 = 2147483647.0;      // CLIP32, quisk.h:13

struct QAgcParam { double limit /* max_out * CLIP32 */, time_release; int buf_size, is_cpx; };
struct QAgcState { int index_read, index_start, is_clipping, pad; double themax, gain, delta, target_gain; };

__global__ __launch_bounds__(64) void q_agc_kernel(double2 *buf, long long stride, int n, QAgcState *state, double2 *ring,
                                                   const double *release_gain, QAgcParam q)
{
    // The overload ramp is built to END on a comparison that is exact in real arithmetic (gain - B * delta ==
    // target, quisk.c:2219,2257): which step leaves the ramp is decided by the last bit.  No FMA contraction
    // here, same operation order as the C source, so the state machine takes the reference's branches.
#pragma clang fp contract(off)
    const int ch = blockIdx.x, lane = threadIdx.x;
    double2 *p = buf + (long long)ch * stride;
    double2 *rb = ring + (long long)ch * q.buf_size;
    QAgcState st = state[ch];
    const double rg = release_gain[ch];
    const int B = q.buf_size;
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        double2 z = make_double2(0, 0), d = make_double2(0, 0);
        int ri = st.index_read + lane;
        if (ri >= B) ri -= B;
        if (lane < cnt) {
            z = p[base + lane];
            d = rb[ri];                              // FIFO output: the sample written B steps ago
            rb[ri] = z;                              // "write new sample at read index"
        }
        const double bm = q.is_cpx ? hypot(z.x, z.y) : fabs(z.x);
        double mygain = 0.0;
        for (int i = 0; i < cnt; i++) {              // uniform: every lane steps the same state
            const double b = lane_bcast(bm, i);
            if (lane == i) mygain = st.gain;
            int ir = st.index_read + i;
            if (ir >= B) ir -= B;
            if (st.is_clipping == 0) {
                if (b * st.gain > q.limit) {
                    st.target_gain = q.limit / b;
                    st.delta = (st.gain - st.target_gain) / B;
                    st.is_clipping = 1;
                    st.themax = b;
                    st.gain -= st.delta;
                } else if (ir == st.index_start) {
                    const double clip_gain = q.limit / st.themax;
                    st.target_gain = rg > clip_gain ? clip_gain : rg;
                    st.themax = b;
                    st.gain = st.gain * (1.0 - q.time_release) + st.target_gain * q.time_release;
                } else {
                    if (st.themax < b) st.themax = b;
                    st.gain = st.gain * (1.0 - q.time_release) + st.target_gain * q.time_release;
                }
            } else {
                if (b > st.themax) {
                    st.themax = b;
                    st.target_gain = q.limit / b;
                    const double dtmp = (st.gain - st.target_gain) / B;
                    if (dtmp > st.delta) st.delta = dtmp;
                }
                st.gain -= st.delta;
                if (st.gain <= st.target_gain) {
                    st.is_clipping = 0;
                    st.gain = st.target_gain;
                    st.themax = b;
                    st.index_start = ir;
                }
            }
        }
        st.index_read += cnt;
        if (st.index_read >= B) st.index_read -= B;
        if (lane < cnt) {
            double2 o = make_double2(d.x * mygain, d.y * mygain);
            const double om = q.is_cpx ? hypot(o.x, o.y) : fabs(o.x);
            if (om > kClip32) { o.x /= om; o.y /= om; }     // quisk.c:2204-2205
            p[base + lane] = o;
        }
    }
    if (lane == 0) state[ch] = st;
}

}  // namespace

struct qh_qagc {
    int device = 0, nch = 0, sample_rate = 0;
    bool inited = false;            // the reference's first call only initialises (quisk.c:2173-2190)
    QAgcParam prm{};
    QAgcState *state = nullptr;
    double2 *ring = nullptr;
    double *gain = nullptr;
    std::vector<double> h_gain;
    bool gain_dirty = true;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ~qh_qagc()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(state); (void)hipFree(ring); (void)hipFree(gain);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

static int qagc_init_state(qh_qagc *h)
{
    std::vector<QAgcState> st((size_t)h->nch);
    for (auto &s : st) { s.index_read = 0; s.index_start = 0; s.is_clipping = 0; s.pad = 0; s.themax = 1.0; s.gain = 100; s.delta = 0; s.target_gain = 100; }
    QH_HIP(hipMemcpyAsync(h->state, st.data(), st.size() * sizeof(QAgcState), hipMemcpyHostToDevice, h->stream));
    QH_HIP(hipMemsetAsync(h->ring, 0, (size_t)h->nch * (size_t)h->prm.buf_size * sizeof(double2), h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    return QH_OK;
}

extern "C" {

qh_qagc *qh_qagc_create(int device, int nch, int sample_rate, double max_out, double release_time, int is_cpx, void *stream)
{
    if (nch <= 0 || sample_rate < 1000 || !(max_out > 0.0) || !(release_time > 0.0)) {
        set_error(QH_ERR_INVALID, "qh_qagc_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_qagc *h = new qh_qagc();
    h->device = device; h->nch = nch; h->sample_rate = sample_rate;
    h->prm.limit = max_out * kClip32;
    h->prm.time_release = 1.0 - std::exp(-1.0 / sample_rate / release_time);      // quisk.c:2185
    h->prm.buf_size = sample_rate * 15 / 1000;                                     // AGC_DELAY, quisk.c:47,2176
    h->prm.is_cpx = is_cpx ? 1 : 0;
    h->h_gain.assign((size_t)nch, 80.0);                                           // agcReleaseGain, quisk.c:191
    auto fail = [&](const char *what) -> qh_qagc * { set_error(QH_ERR_HIP, "qh_qagc_create: %s failed", what); delete h; return nullptr; };
    if (h->prm.buf_size < 64) { set_error(QH_ERR_INVALID, "qh_qagc_create: sample rate too low"); delete h; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
        h->own_stream = true;
    }
    h->stream = s;
    if (hipMalloc((void **)&h->state, (size_t)nch * sizeof(QAgcState)) != hipSuccess ||
        hipMalloc((void **)&h->ring, (size_t)nch * (size_t)h->prm.buf_size * sizeof(double2)) != hipSuccess ||
        hipMalloc((void **)&h->gain, (size_t)nch * sizeof(double)) != hipSuccess) return fail("hipMalloc");
    if (qagc_init_state(h)) { delete h; return nullptr; }
    return h;
}

void qh_qagc_destroy(qh_qagc *h) { delete h; }

// set_agc (quisk.c:4543): the AGC's maximum gain
int qh_qagc_set_gain(qh_qagc *h, int ch, double release_gain)
{
    if (!h || ch < -1 || ch >= h->nch) return set_error(QH_ERR_INVALID, "qh_qagc_set_gain: bad arguments");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? h->nch : ch + 1); c++) h->h_gain[(size_t)c] = release_gain;
    h->gain_dirty = true;
    return QH_OK;
}

// process_agc takes is_cpx per call (quisk.c:2162: |z| for the DGT-IQ stream, |Re z| otherwise); the state carries over
int qh_qagc_set_cpx(qh_qagc *h, int is_cpx)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_qagc_set_cpx: null handle");
    h->prm.is_cpx = is_cpx ? 1 : 0;
    return QH_OK;
}

int qh_qagc_reset(qh_qagc *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_qagc_reset: null handle");
    QH_HIP(hipSetDevice(h->device));
    h->inited = false;
    return qagc_init_state(h);
}

int qh_qagc_process(qh_qagc *h, void *d_buf, long long stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!d_buf || stride < n))) return set_error(QH_ERR_INVALID, "qh_qagc_process: bad arguments");
    if (n == 0) return QH_OK;
    if (!h->inited) { h->inited = true; return QH_OK; }            // first call: state set up, samples untouched
    QH_HIP(hipSetDevice(h->device));
    if (h->gain_dirty) {
        QH_HIP(hipMemcpyAsync(h->gain, h->h_gain.data(), (size_t)h->nch * sizeof(double), hipMemcpyHostToDevice, h->stream));
        QH_HIP(hipStreamSynchronize(h->stream));
        h->gain_dirty = false;
    }
    hipLaunchKernelGGL(q_agc_kernel, dim3((unsigned)h->nch), dim3(64), 0, h->stream, (double2 *)d_buf, stride, n, h->state, h->ring,
                       h->gain, h->prm);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_qagc_process_host(qh_qagc *h, void *h_buf, long long stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!h_buf || stride < n))) return set_error(QH_ERR_INVALID, "qh_qagc_process_host: bad arguments");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    double2 *d = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)h->nch * (size_t)n * sizeof(double2)));
    int rc = QH_OK;
    if (hipMemcpy2DAsync(d, (size_t)n * 16, h_buf, (size_t)stride * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyHostToDevice, h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "upload failed");
    if (rc == QH_OK) rc = qh_qagc_process(h, d, n, n);
    if (rc == QH_OK && hipMemcpy2DAsync(h_buf, (size_t)stride * 16, d, (size_t)n * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyDeviceToHost,
                                         h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "download failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "synchronize failed");
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
