// qh_wdsp_compat.cpp -- the WDSP exports Quisk binds (include/quiskhip.h group 2) on top of the
// MI355X engine.  quisk_wdsp.py loads the library by name and calls these through ctypes
// (quisk_wdsp.py:27-41,79-91); quisk_wdsp.c calls fexchange0 through a function pointer
// (quisk_wdsp.c:22,57).
//
// Each open channel = a one-channel qh_rxa engine + the host-side double ring of wdsp/iobuffs.c.
// The reference runs xrxa() on a DSP thread that is woken by fexchange0 and publishes the PREVIOUS
// block's result before it processes the next one (dexchange before xrxa, wdsp/main.c:48-49); with
// bfo = 1 the caller blocks until output is available, so in sample terms the behaviour is
// deterministic: output lags input by (DSP_MULT-1)*r2_size + dsp_outsize samples.  The same
// sequence is executed here synchronously on the calling thread.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"
#include "qh_emnr_tables.hpp"

extern "C" int qh_rxa_flush(qh_rxa *e);
extern "C" void qh_wdsp_shim_release_device(int channel);

namespace {

constexpr int kMaxChannels = 32;    // wdsp/comm.h:117
constexpr int kDspMult = 2;         // wdsp/comm.h:118

// The up / down slews of wdsp/iobuffs.c:47-160,226-300 are gain ENVELOPES over a sample count here, not the reference's
// per-sample state machines: g(p) with p = 0 at the trigger (up: the first non-zero input sample after the channel is
// switched on, which itself is muted; down: the first output sample after SetChannelState(ch, 0, 0), which still passes).
//   up:   0 for p <= lead,  0.5 (1 - cos(pi k / ramp)) for k = p - lead - 1 = 0 .. ramp,  1 beyond
//   down: 1 for p <= lead,  0.5 (1 + cos(pi k / ramp)) for k = 0 .. ramp,                 0 beyond
// lead = delay + 1 when there is a delay (the reference's counters run down to zero inclusive), else 0.  The host keeps
// the positions; the multiplication is a small kernel on the block that is on its way to / from the GPU anyway.
struct Slew {
    int delay = 0, ramp = 0;        // samples
    bool armed = false;             // upflag / downflag
    long long p = -1;               // up: envelope position of stream sample `origin`; down: position of the next output sample
    long long origin = 0;
    int lead() const { return delay > 0 ? delay + 1 : 0; }
    long long level_from() const { return (long long)lead() + (ramp > 0 ? ramp + 1 : 0) + 1; }     // first position at the end level
    void reset() { armed = false; p = -1; origin = 0; }
};

__global__ void slew_kernel(double2 *buf, int n, long long p0, int lead, int ramp, int rising)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long p = p0 + i;
    double g;
    if (p <= lead) g = rising ? 0.0 : 1.0;
    else if (ramp > 0 && p - lead - 1 <= ramp) {
        const double c = cospi((double)(p - lead - 1) / (double)ramp);
        g = rising ? 0.5 * (1.0 - c) : 0.5 * (1.0 + c);
    } else g = rising ? 1.0 : 0.0;
    buf[i].x *= g; buf[i].y *= g;
}

// ---- the exchange with every buffer on the GPU (qh_wdsp_fexchange0_device): the rings, the block buffers and the up-slew's trigger
// live in device memory, the indices stay on the host (they do not depend on the data).  What does depend on the data -- the
// up-slew starts at the first non-zero input sample, iobuffs.c:104-113 -- is found by the kernel that writes the ring.
struct DevSlew { long long p, origin; int armed, pad; };

// One launch per fexchange0 call in the common case: every copy of the exchange that precedes the DSP block, as the phases of one
// workgroup (a phase with n = 0 is skipped; the phases are ordered by the workgroup's barriers, which also order their global
// memory accesses within the workgroup):
//   A  in_size samples into r1 + the up-slew's bookkeeping (fexchange0, iobuffs.c:98-160,478-480)
//   B  dexchange (iobuffs.c:583-604): the previous DSP block's output into r2; the next DSP block out of r1 into the engine's
//      input buffer, under the up-slew's envelope where it still rises
//   C  out_size samples out of r2 (iobuffs.c:497-511), under the down-slew's envelope, or zeros
struct XchgArgs {
    const double2 *in; double2 *r1_in; int n_in; DevSlew *sl; long long in_count, level_from;
    const double2 *dout; double2 *r2_in; int n_r2;
    const double2 *r1_out; double2 *blk; int n_blk; long long blk0; int up_lead, up_ramp;
    const double2 *r2_out; double2 *out; int n_out, out_mode;      // out_mode 0 copy, 1 zeros, 2 copy under the down-slew from position down_p0
    long long down_p0; int down_lead, down_ramp;
};
__global__ __launch_bounds__(256) void exchange_kernel(XchgArgs a)
{
    __shared__ int first;
    const int t = threadIdx.x;
    if (a.n_in > 0) {
        if (t == 0) first = a.n_in;
        const int armed = a.sl->armed;
        const long long p_was = a.sl->p;
        __syncthreads();
        int mine = a.n_in;
        for (int i = t; i < a.n_in; i += 256) {
            const double2 v = a.in[i];
            a.r1_in[i] = v;
            if (mine == a.n_in && (v.x != 0.0 || v.y != 0.0)) mine = i;
        }
        if (armed && p_was < 0 && mine < a.n_in) atomicMin(&first, mine);
        __syncthreads();
        if (armed && t == 0) {
            long long p = p_was, origin = a.sl->origin;
            if (p < 0 && first < a.n_in) { origin = a.in_count + first; p = 0; a.sl->origin = origin; a.sl->p = 0; }
            if (p >= 0 && a.in_count + a.n_in - 1 - origin + p >= a.level_from) a.sl->armed = 0;
        }
        __syncthreads();
    }
    for (int i = t; i < a.n_r2; i += 256) a.r2_in[i] = a.dout[i];
    if (a.n_blk > 0) {
        const long long p = a.sl->p, origin = a.sl->origin;
        for (int i = t; i < a.n_blk; i += 256) {
            double2 v = a.r1_out[i];
            if (p >= 0) {
                const long long q = a.blk0 + i - origin + p;
                double g = 1.0;
                if (q <= a.up_lead) g = 0.0;
                else if (a.up_ramp > 0 && q - a.up_lead - 1 <= a.up_ramp) g = 0.5 * (1.0 - cospi((double)(q - a.up_lead - 1) / (double)a.up_ramp));
                v.x *= g; v.y *= g;
            }
            a.blk[i] = v;
        }
    }
    if (a.n_out > 0) {
        __syncthreads();
        for (int i = t; i < a.n_out; i += 256) {
            double2 v = make_double2(0.0, 0.0);
            if (a.out_mode != 1) v = a.r2_out[i];
            if (a.out_mode == 2) {
                const long long q = a.down_p0 + i;
                double g = 0.0;
                if (q <= a.down_lead) g = 1.0;
                else if (a.down_ramp > 0 && q - a.down_lead - 1 <= a.down_ramp) g = 0.5 * (1.0 + cospi((double)(q - a.down_lead - 1) / (double)a.down_ramp));
                v.x *= g; v.y *= g;
            }
            a.out[i] = v;
        }
    }
}
__global__ void slew_set_kernel(DevSlew *sl, long long p, long long origin, int armed) { sl->p = p; sl->origin = origin; sl->armed = armed; sl->pad = 0; }
// quisk_wdsp.c:43-49 / :63-66: into the shim's ring divided by CLIP32; the results times CLIP32
__global__ void shim_in_kernel(const double2 *x, int n, double2 *ring, int W, int size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int w = W + i;
    if (w >= size) w -= size;
    ring[w] = make_double2(x[i].x / 2147483647.0, x[i].y / 2147483647.0);
}
__global__ void shim_scale_kernel(double2 *x, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i].x *= 2147483647.0; x[i].y *= 2147483647.0; }
}

struct Chan {
    bool open = false, emnr_tables = false;
    qh_rxa *eng = nullptr;
    int in_size = 0, dsp_size = 0, in_rate = 0, dsp_rate = 0, out_rate = 0;
    int dsp_insize = 0, dsp_outsize = 0, out_size = 0;
    int state = 0, exchange = 0, bfo = 1;
    double tdelayup = 0, tslewup = 0, tdelaydown = 0, tslewdown = 0;
    // iobuffs (wdsp/iobuffs.h)
    int r1_outsize = 0, r1_size = 0, r2_insize = 0, r2_size = 0, r1_active = 0, r2_active = 0;
    std::vector<double> r1, r2, outbuff;
    int r1_inidx = 0, r1_outidx = 0, r1_unqueued = 0, r2_inidx = 0, r2_outidx = 0, r2_havesamps = 0, r2_unqueued = 0;
    int sem_buffready = 0, sem_outready = 0;
    Slew up, down;
    long long in_count = 0, dsp_count = 0;          // input samples written to r1 / taken out of it since the rings were reset
    // staging: pinned host + device block buffers
    double *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    // device mode (qh_wdsp_fexchange0_device): r1 / r2 in device memory, outbuff = d_out, the up-slew's trigger in *d_up
    bool dev_mode = false;
    double *d_r1 = nullptr, *d_r2 = nullptr;
    DevSlew *d_up = nullptr;
};

Chan g_ch[kMaxChannels];
std::recursive_mutex g_mtx[kMaxChannels];   // csEXCH + csDSP: setters may come from another thread
thread_local int g_status = QH_OK;

bool valid(int channel)
{
    if (channel < 0 || channel >= kMaxChannels) {
        g_status = qh::set_error(QH_ERR_INVALID, "WDSP channel %d out of range", channel);
        return false;
    }
    return true;
}

void create_slews(Chan &c)          // the sample counts of wdsp/iobuffs.c:47-68: up-slew at the input rate, down-slew at the output rate
{
    c.up = Slew(); c.down = Slew();
    c.up.delay = (int)(c.tdelayup * c.in_rate); c.up.ramp = (int)(c.tslewup * c.in_rate);
    c.down.delay = (int)(c.tdelaydown * c.out_rate); c.down.ramp = (int)(c.tslewdown * c.out_rate);
}

// device mode: the up-slew's state as the host has just set it
void push_up(Chan &c)
{
    if (!c.dev_mode) return;
    hipLaunchKernelGGL(slew_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)qh_rxa_stream(c.eng), c.d_up, c.up.p, c.up.origin, c.up.armed ? 1 : 0);
}
void flush_slews(Chan &c) { c.up.reset(); c.down.reset(); push_up(c); }

void init_rings(Chan &c)            // create_iobuffs / flush_iobuffs, wdsp/iobuffs.c:384-455
{
    c.r1.assign((size_t)c.r1_active * 2, 0.0);
    c.r2.assign((size_t)c.r2_active * 2, 0.0);
    c.outbuff.assign((size_t)c.dsp_outsize * 2, 0.0);
    c.r1_inidx = 0; c.r1_outidx = 0; c.r1_unqueued = 0;
    c.r2_inidx = (kDspMult - 1) * c.r2_size;
    c.r2_outidx = 0;
    c.r2_havesamps = (kDspMult - 1) * c.r2_size;
    const int n = c.r2_havesamps / c.out_size;
    c.r2_unqueued = c.r2_havesamps - n * c.out_size;
    c.sem_buffready = 0;
    c.sem_outready = n;
    c.in_count = 0; c.dsp_count = 0;
    if (c.dev_mode) {
        hipStream_t s = (hipStream_t)qh_rxa_stream(c.eng);
        (void)hipMemsetAsync(c.d_r1, 0, (size_t)c.r1_active * 16, s);
        (void)hipMemsetAsync(c.d_r2, 0, (size_t)c.r2_active * 16, s);
        (void)hipMemsetAsync(c.d_out, 0, (size_t)c.dsp_outsize * 16, s);
    }
}

// Multiplies the n samples at `buf` (device-visible) by the envelope, the first one at position p0; on the engine's stream.
int apply_slew(Chan &c, const Slew &sl, double *buf, int n, long long p0, int rising)
{
    hipStream_t st = (hipStream_t)qh_rxa_stream(c.eng);
    hipLaunchKernelGGL(slew_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<double2 *>(buf), n, p0, sl.lead(),
                       sl.ramp, rising);
    return hipGetLastError() == hipSuccess ? QH_OK : qh::set_error(QH_ERR_HIP, "slew kernel launch failed");
}

long long g_graph_launches = 0;      // blocks replayed by engines that have since been closed

// one DSP-thread iteration: dexchange (wdsp/iobuffs.c:583-604) then xrxa on the GPU
int dsp_iteration(Chan &c)
{
    c.r2_havesamps += c.r2_insize;
    std::memcpy(c.r2.data() + 2 * c.r2_inidx, c.outbuff.data(), (size_t)c.r2_insize * 2 * sizeof(double));
    if ((c.r2_inidx += c.r2_insize) == c.r2_active) c.r2_inidx = 0;
    if (c.bfo && (c.r2_unqueued += c.r2_insize) >= c.out_size) {
        const int n = c.r2_unqueued / c.out_size;
        c.sem_outready += n;
        c.r2_unqueued -= n * c.out_size;
    }
    std::memcpy(c.h_in, c.r1.data() + 2 * c.r1_outidx, (size_t)c.r1_outsize * 2 * sizeof(double));
    if ((c.r1_outidx += c.r1_outsize) == c.r1_active) c.r1_outidx = 0;
    // the up-slew: this block holds stream samples dsp_count ...; while any of them sits below the envelope's end level the
    // block takes the envelope on its way in (before the trigger the stream is zeros: nothing to do)
    const long long blk0 = c.dsp_count;
    c.dsp_count += c.r1_outsize;
    const bool slew_in = c.up.p >= 0 && blk0 - c.up.origin + c.up.p < c.up.level_from();
    // xrxa: one block through the engine.  The kernels read the input block from and write the output block to the
    // pinned staging buffers directly (a few KB over PCIe: no copy-engine hop); QH_WDSP_IO=copy stages both through
    // device buffers with two async copies instead.  The launch sequence itself is replayed from hipGraphs by the
    // engine once the parameters stand still (qh_rxa_set_graph_replay, switched on in OpenChannel).
    static const bool staged = [] { const char *e = std::getenv("QH_WDSP_IO"); return e && std::strcmp(e, "copy") == 0; }();
    int rc = QH_OK;
    if (staged) {
        hipStream_t s = (hipStream_t)qh_rxa_stream(c.eng);
        if (hipMemcpyAsync(c.d_in, c.h_in, (size_t)c.dsp_insize * 2 * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "fexchange0: host to device copy failed");
        if (slew_in) if (int e = apply_slew(c, c.up, c.d_in, c.dsp_insize, blk0 - c.up.origin + c.up.p, 1)) return e;
        rc = qh_rxa_process(c.eng, c.d_in, c.dsp_insize, c.d_out, c.dsp_outsize, 1);
        if (rc == QH_OK && hipMemcpyAsync(c.h_out, c.d_out, (size_t)c.dsp_outsize * 2 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "fexchange0: device to host copy failed");
    } else {
        if (slew_in) if (int e = apply_slew(c, c.up, c.h_in, c.dsp_insize, blk0 - c.up.origin + c.up.p, 1)) return e;
        rc = qh_rxa_process(c.eng, c.h_in, c.dsp_insize, c.h_out, c.dsp_outsize, 1);
    }
    if (rc) return rc;
    rc = qh_rxa_synchronize(c.eng);
    if (rc) return rc;
    std::memcpy(c.outbuff.data(), c.h_out, (size_t)c.dsp_outsize * 2 * sizeof(double));
    return QH_OK;
}

void free_staging(Chan &c)
{
    if (c.d_in) (void)hipFree(c.d_in);
    if (c.d_out) (void)hipFree(c.d_out);
    if (c.h_in) (void)hipHostFree(c.h_in);
    if (c.h_out) (void)hipHostFree(c.h_out);
    if (c.d_r1) (void)hipFree(c.d_r1);
    if (c.d_r2) (void)hipFree(c.d_r2);
    if (c.d_up) (void)hipFree(c.d_up);
    c.d_in = c.d_out = c.h_in = c.h_out = c.d_r1 = c.d_r2 = nullptr;
    c.d_up = nullptr;
    c.dev_mode = false;
}

// Host rings <-> device rings: a channel is driven through fexchange0 (host pointers) or through qh_wdsp_fexchange0_device; a caller
// that changes sides in mid-stream takes the rings, the pending output block and the up-slew's state along.
int set_mode(Chan &c, bool dev)
{
    if (c.dev_mode == dev) return QH_OK;
    hipStream_t s = (hipStream_t)qh_rxa_stream(c.eng);
    if (hipStreamSynchronize(s) != hipSuccess) return qh::set_error(QH_ERR_HIP, "fexchange0: synchronize failed");
    if (dev) {
        if (!c.d_r1 && (hipMalloc((void **)&c.d_r1, (size_t)c.r1_active * 16) != hipSuccess || hipMalloc((void **)&c.d_r2, (size_t)c.r2_active * 16) != hipSuccess ||
                        hipMalloc((void **)&c.d_up, sizeof(DevSlew)) != hipSuccess))
            return qh::set_error(QH_ERR_HIP, "fexchange0: device ring allocation failed");
        const DevSlew h{c.up.p, c.up.origin, c.up.armed ? 1 : 0, 0};
        if (hipMemcpy(c.d_r1, c.r1.data(), (size_t)c.r1_active * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c.d_r2, c.r2.data(), (size_t)c.r2_active * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c.d_out, c.outbuff.data(), (size_t)c.dsp_outsize * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c.d_up, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "fexchange0: ring upload failed");
    } else {
        DevSlew h{};
        if (hipMemcpy(c.r1.data(), c.d_r1, (size_t)c.r1_active * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(c.r2.data(), c.d_r2, (size_t)c.r2_active * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(c.outbuff.data(), c.d_out, (size_t)c.dsp_outsize * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&h, c.d_up, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "fexchange0: ring download failed");
        c.up.p = h.p; c.up.origin = h.origin; c.up.armed = h.armed != 0;
    }
    c.dev_mode = dev;
    return QH_OK;
}

struct Locked {
    Chan *c = nullptr;
    std::unique_lock<std::recursive_mutex> lk;
    explicit Locked(int channel)
    {
        if (!valid(channel)) return;
        lk = std::unique_lock<std::recursive_mutex>(g_mtx[channel]);
        if (!g_ch[channel].open) {
            g_status = qh::set_error(QH_ERR_INVALID, "WDSP channel %d is not open", channel);
            return;
        }
        c = &g_ch[channel];
    }
};

}  // namespace

extern "C" {

int qh_wdsp_status(void) { return g_status; }
long long qh_wdsp_graph_launches(void)       // DSP blocks replayed from a captured hipGraph so far, over all channels
{
    long long n = g_graph_launches;
    for (int ch = 0; ch < kMaxChannels; ch++) {
        std::unique_lock<std::recursive_mutex> lk(g_mtx[ch]);
        if (g_ch[ch].open) n += qh_rxa_graph_launches(g_ch[ch].eng);
    }
    return n;
}

int GetWDSPVersion(void) { return 125; }    // the WDSP release Quisk 4.2.52 bundles

void OpenChannel(int channel, int in_size, int dsp_size, int input_samplerate, int dsp_rate, int output_samplerate,
                 int type, int state, double tdelayup, double tslewup, double tdelaydown, double tslewdown, int bfo)
{
    g_status = QH_OK;
    if (!valid(channel)) return;
    std::unique_lock<std::recursive_mutex> lk(g_mtx[channel]);
    Chan &c = g_ch[channel];
    if (c.open) { g_status = qh::set_error(QH_ERR_INVALID, "WDSP channel %d is already open", channel); return; }
    if (type != 0) { g_status = qh::set_error(QH_ERR_UNSUPPORTED, "only RXA channels (type 0) are provided"); return; }
    if (in_size <= 0 || dsp_size <= 0) { g_status = qh::set_error(QH_ERR_INVALID, "bad buffer sizes"); return; }
    c.eng = qh_rxa_create(0, 1, dsp_size, input_samplerate, dsp_rate, output_samplerate, nullptr);
    if (!c.eng) { g_status = QH_ERR_NO_DEVICE; return; }       // message already set
    c.in_size = in_size; c.dsp_size = dsp_size;
    c.in_rate = input_samplerate; c.dsp_rate = dsp_rate; c.out_rate = output_samplerate;
    c.tdelayup = tdelayup; c.tslewup = tslewup; c.tdelaydown = tdelaydown; c.tslewdown = tslewdown;
    c.bfo = bfo; c.state = state;
    // pre_main_build, wdsp/channel.c:39-52
    c.dsp_insize = qh_rxa_dsp_insize(c.eng);
    c.dsp_outsize = qh_rxa_dsp_outsize(c.eng);
    if (c.in_rate >= c.out_rate) c.out_size = in_size / (c.in_rate / c.out_rate);
    else c.out_size = in_size * (c.out_rate / c.in_rate);
    // create_iobuffs, wdsp/iobuffs.c:384-423
    c.r1_outsize = c.dsp_insize;
    c.r1_size = c.r1_outsize > in_size ? c.r1_outsize : in_size;
    c.r2_insize = c.dsp_outsize;
    c.r2_size = c.out_size > c.r2_insize ? c.out_size : c.r2_insize;
    c.r1_active = kDspMult * c.r1_size;
    c.r2_active = kDspMult * c.r2_size;
    init_rings(c);
    create_slews(c);
    if (hipMalloc((void **)&c.d_in, (size_t)c.dsp_insize * 2 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c.d_out, (size_t)c.dsp_outsize * 2 * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&c.h_in, (size_t)c.dsp_insize * 2 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c.h_out, (size_t)c.dsp_outsize * 2 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        g_status = qh::set_error(QH_ERR_HIP, "OpenChannel: staging allocation failed");
        free_staging(c);
        qh_rxa_destroy(c.eng);
        c.eng = nullptr;
        return;
    }
    c.exchange = 0;
    c.open = true;
    if (state) { c.up.armed = true; c.exchange = 1; }      // wdsp/channel.c:92-98
    (void)qh_rxa_enable_meters(c.eng, 1);           // WDSP's meters always run (RXA.c:69-82)
    // the per-block sequence is launch-bound: replay it from hipGraphs (QH_WDSP_NO_GRAPHS=1 keeps plain launches)
    if (!std::getenv("QH_WDSP_NO_GRAPHS")) (void)qh_rxa_set_graph_replay(c.eng, 1);
}

void CloseChannel(int channel)
{
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return;
    Chan &c = *L.c;
    free_staging(c);
    qh_wdsp_shim_release_device(channel);       // the re-blocking ring keeps its samples (the reference's statics outlive the channel), on the host
    g_graph_launches += qh_rxa_graph_launches(c.eng);
    qh_rxa_destroy(c.eng);
    c.eng = nullptr;
    c.open = false;
    c.emnr_tables = false;
}

int SetChannelState(int channel, int state, int dmode)
{
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return 0;
    Chan &c = *L.c;
    const int prior = c.state;
    if (c.state != state) {
        c.state = state;
        if (state == 0) {
            // wdsp/channel.c:269-288.  With dmode the reference waits up to 100 ms for another thread's
            // fexchange0 calls to finish the down-slew; when none arrive (the single-threaded use Quisk
            // makes of it, quisk_wdsp.py:117-139) it times out: exchange off, no flush.  That outcome is
            // produced here at once.
            if (dmode) { c.exchange = 0; c.down.reset(); }
            else { c.down.armed = true; c.down.p = 0; }
        } else {
            c.up.armed = true; c.up.p = -1;
            push_up(c);
            c.exchange = 1;
        }
    }
    return prior;
}

void fexchange0(int channel, double *in, double *out, int *error)
{
    g_status = QH_OK;
    *error = 0;
    Locked L(channel);
    if (!L.c) { *error = -1; return; }
    Chan &c = *L.c;
    if (!c.exchange) return;                        // wdsp/iobuffs.c:471: `out` is left untouched
    if (int rc = set_mode(c, false)) { g_status = rc; *error = -1; return; }
    std::memcpy(c.r1.data() + 2 * c.r1_inidx, in, (size_t)c.in_size * 2 * sizeof(double));
    if (c.up.armed) {
        // iobuffs.c:98-160 as bookkeeping: the envelope is triggered by the first non-zero sample; the flag drops at the end of
        // the first call whose last sample has reached the end level
        if (c.up.p < 0) {
            for (int i = 0; i < c.in_size; i++)
                if (in[2 * i] != 0.0 || in[2 * i + 1] != 0.0) { c.up.origin = c.in_count + i; c.up.p = 0; break; }
        }
        if (c.up.p >= 0 && c.in_count + c.in_size - 1 - c.up.origin + c.up.p >= c.up.level_from()) c.up.armed = false;
    }
    c.in_count += c.in_size;
    if ((c.r1_unqueued += c.in_size) >= c.r1_outsize) {
        const int n = c.r1_unqueued / c.r1_outsize;
        c.sem_buffready += n;
        c.r1_unqueued -= n * c.r1_outsize;
    }
    if ((c.r1_inidx += c.in_size) == c.r1_active) c.r1_inidx = 0;
    while (c.sem_buffready > 0) {                   // the DSP thread's loop, wdsp/main.c:40-58
        c.sem_buffready--;
        int rc = dsp_iteration(c);
        if (rc) { g_status = rc; *error = -1; std::memset(out, 0, (size_t)c.out_size * 2 * sizeof(double)); return; }
    }
    int doit = c.r2_havesamps >= c.out_size;
    if ((c.r2_havesamps -= c.out_size) < 0) c.r2_havesamps = 0;
    int ready = 0;
    if (c.bfo) { if (c.sem_outready > 0) { c.sem_outready--; ready = 1; } }
    else ready = doit;
    if (ready) {
        std::memcpy(out, c.r2.data() + 2 * c.r2_outidx, (size_t)c.out_size * 2 * sizeof(double));
        if (c.down.armed) {
            // iobuffs.c:226-300: this call's samples sit at envelope positions p ... p + out_size - 1; behind the ramp come
            // out_size + 1 zeros, and the call that ends beyond those stops the channel and flushes it (iobuffs.c:499-503)
            if (c.down.p < c.down.level_from()) {
                for (int o = 0; o < c.out_size; o += c.dsp_outsize) {       // through the pinned block buffer, a DSP block at a time
                    const int m = c.out_size - o < c.dsp_outsize ? c.out_size - o : c.dsp_outsize;
                    std::memcpy(c.h_out, out + 2 * (size_t)o, (size_t)m * 2 * sizeof(double));
                    int rc = apply_slew(c, c.down, c.h_out, m, c.down.p + o, 0);
                    if (!rc) rc = qh_rxa_synchronize(c.eng);
                    if (rc) { g_status = rc; *error = -1; return; }
                    std::memcpy(out + 2 * (size_t)o, c.h_out, (size_t)m * 2 * sizeof(double));
                }
            } else {
                std::memset(out, 0, (size_t)c.out_size * 2 * sizeof(double));
            }
            c.down.p += c.out_size;
            if (c.down.p - 1 >= c.down.level_from() + c.out_size + 1) {
                c.exchange = 0;
                init_rings(c);
                flush_slews(c);
                (void)qh_rxa_flush(c.eng);
                return;
            }
        }
    } else {
        std::memset(out, 0, (size_t)c.out_size * 2 * sizeof(double));
        *error += -2;
    }
    if ((c.r2_outidx += c.out_size) == c.r2_active) c.r2_outidx = 0;
}

}  // extern "C"

namespace {
// fexchange0 with `in` (in_size samples) and `out` (out_size samples) in device memory: the same bookkeeping on the host, the copies
// and the slews as phases of exchange_kernel on the engine's stream, nothing waited for.  A call that runs one DSP block (in_size =
// dsp_insize, Quisk's case) is two launches: the exchange and the block (a replayed hipGraph); the output copy (out of r2, which the
// block does not touch) rides in the exchange ahead of the block.
int fexchange0_dev(Chan &c, const double *d_in, double *d_out, int *error)
{
    *error = 0;
    if (!c.exchange) return QH_OK;
    hipStream_t s = (hipStream_t)qh_rxa_stream(c.eng);
    XchgArgs a{};
    a.sl = c.d_up;
    auto launch = [&]() {
        hipLaunchKernelGGL(exchange_kernel, dim3(1), dim3(256), 0, s, a);
        a = XchgArgs{};
        a.sl = c.d_up;
    };
    a.in = reinterpret_cast<const double2 *>(d_in); a.r1_in = reinterpret_cast<double2 *>(c.d_r1) + c.r1_inidx; a.n_in = c.in_size;
    a.in_count = c.in_count; a.level_from = c.up.level_from();
    c.in_count += c.in_size;
    if ((c.r1_unqueued += c.in_size) >= c.r1_outsize) {
        const int n = c.r1_unqueued / c.r1_outsize;
        c.sem_buffready += n;
        c.r1_unqueued -= n * c.r1_outsize;
    }
    if ((c.r1_inidx += c.in_size) == c.r1_active) c.r1_inidx = 0;
    bool out_done = false, stop = false;
    auto output = [&]() {      // iobuffs.c:487-515 (and the down-slew, :226-300, :499-503)
        out_done = true;
        const int doit = c.r2_havesamps >= c.out_size;
        if ((c.r2_havesamps -= c.out_size) < 0) c.r2_havesamps = 0;
        int ready = 0;
        if (c.bfo) { if (c.sem_outready > 0) { c.sem_outready--; ready = 1; } }
        else ready = doit;
        a.out = reinterpret_cast<double2 *>(d_out); a.n_out = c.out_size;
        a.r2_out = reinterpret_cast<const double2 *>(c.d_r2) + c.r2_outidx;
        if (ready) {
            a.out_mode = 0;
            if (c.down.armed) {
                a.out_mode = c.down.p < c.down.level_from() ? 2 : 1;
                a.down_p0 = c.down.p; a.down_lead = c.down.lead(); a.down_ramp = c.down.ramp;
                c.down.p += c.out_size;
                if (c.down.p - 1 >= c.down.level_from() + c.out_size + 1) stop = true;
            }
        } else {
            a.out_mode = 1;
            *error += -2;
        }
        if (!stop && (c.r2_outidx += c.out_size) == c.r2_active) c.r2_outidx = 0;
    };
    while (c.sem_buffready > 0) {           // the DSP thread's loop (wdsp/main.c:40-58): dexchange, then xrxa
        c.sem_buffready--;
        c.r2_havesamps += c.r2_insize;
        a.dout = reinterpret_cast<const double2 *>(c.d_out); a.r2_in = reinterpret_cast<double2 *>(c.d_r2) + c.r2_inidx; a.n_r2 = c.r2_insize;
        if ((c.r2_inidx += c.r2_insize) == c.r2_active) c.r2_inidx = 0;
        if (c.bfo && (c.r2_unqueued += c.r2_insize) >= c.out_size) {
            const int n = c.r2_unqueued / c.out_size;
            c.sem_outready += n;
            c.r2_unqueued -= n * c.out_size;
        }
        a.r1_out = reinterpret_cast<const double2 *>(c.d_r1) + c.r1_outidx; a.blk = reinterpret_cast<double2 *>(c.d_in); a.n_blk = c.r1_outsize;
        a.blk0 = c.dsp_count; a.up_lead = c.up.lead(); a.up_ramp = c.up.ramp;
        if ((c.r1_outidx += c.r1_outsize) == c.r1_active) c.r1_outidx = 0;
        c.dsp_count += c.r1_outsize;
        if (c.sem_buffready == 0) output();
        launch();
        if (int rc = qh_rxa_process(c.eng, c.d_in, c.dsp_insize, c.d_out, c.dsp_outsize, 1)) {
            *error = -1;
            (void)hipMemsetAsync(d_out, 0, (size_t)c.out_size * 16, s);
            return rc;
        }
    }
    if (!out_done) { output(); launch(); }
    if (hipGetLastError() != hipSuccess) return qh::set_error(QH_ERR_HIP, "fexchange0: launch failed");
    if (stop) {
        c.exchange = 0;
        init_rings(c);
        flush_slews(c);
        (void)qh_rxa_flush(c.eng);
    }
    return QH_OK;
}
}  // namespace

extern "C" {

#define WDSP_SETTER(call)                                   \
    do {                                                    \
        g_status = QH_OK;                                   \
        Locked L(channel);                                  \
        if (!L.c) return;                                   \
        int rc = (call);                                    \
        if (rc) g_status = rc;                              \
    } while (0)

void SetRXAMode(int channel, int mode) { WDSP_SETTER(qh_rxa_SetRXAMode(L.c->eng, 0, mode)); }
void RXASetPassband(int channel, double f_low, double f_high) { WDSP_SETTER(qh_rxa_RXASetPassband(L.c->eng, 0, f_low, f_high)); }
void RXASetNC(int channel, int nc) { WDSP_SETTER(qh_rxa_RXASetNC(L.c->eng, 0, nc)); }
void SetRXAShiftRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXAShiftRun(L.c->eng, 0, run)); }
void SetRXAShiftFreq(int channel, double fshift) { WDSP_SETTER(qh_rxa_SetRXAShiftFreq(L.c->eng, 0, fshift)); }
void RXANBPSetRun(int channel, int run) { WDSP_SETTER(qh_rxa_RXANBPSetRun(L.c->eng, 0, run)); }
void RXANBPSetFreqs(int channel, double flow, double fhigh) { WDSP_SETTER(qh_rxa_RXANBPSetFreqs(L.c->eng, 0, flow, fhigh)); }
void SetRXABandpassRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXABandpassRun(L.c->eng, 0, run)); }
void SetRXABandpassFreqs(int channel, double f_low, double f_high) { WDSP_SETTER(qh_rxa_SetRXABandpassFreqs(L.c->eng, 0, f_low, f_high)); }
void SetRXAAGCMode(int channel, int mode) { WDSP_SETTER(qh_rxa_SetRXAAGCMode(L.c->eng, 0, mode)); }
void SetRXAAGCFixed(int channel, double fixed_agc) { WDSP_SETTER(qh_rxa_SetRXAAGCFixed(L.c->eng, 0, fixed_agc)); }
void SetRXAAGCAttack(int channel, int attack) { WDSP_SETTER(qh_rxa_SetRXAAGCAttack(L.c->eng, 0, attack)); }
void SetRXAAGCDecay(int channel, int decay) { WDSP_SETTER(qh_rxa_SetRXAAGCDecay(L.c->eng, 0, decay)); }
void SetRXAAGCHang(int channel, int hang) { WDSP_SETTER(qh_rxa_SetRXAAGCHang(L.c->eng, 0, hang)); }
void SetRXAAGCTop(int channel, double max_agc) { WDSP_SETTER(qh_rxa_SetRXAAGCTop(L.c->eng, 0, max_agc)); }
void SetRXAAGCSlope(int channel, int slope) { WDSP_SETTER(qh_rxa_SetRXAAGCSlope(L.c->eng, 0, slope)); }
void SetRXAAGCHangThreshold(int channel, int t) { WDSP_SETTER(qh_rxa_SetRXAAGCHangThreshold(L.c->eng, 0, t)); }
void SetRXAPanelGain1(int channel, double gain) { WDSP_SETTER(qh_rxa_SetRXAPanelGain1(L.c->eng, 0, gain)); }
void SetRXAPanelGain2(int channel, double gainI, double gainQ) { WDSP_SETTER(qh_rxa_SetRXAPanelGain2(L.c->eng, 0, gainI, gainQ)); }
void SetRXAPanelSelect(int channel, int select) { WDSP_SETTER(qh_rxa_SetRXAPanelSelect(L.c->eng, 0, select)); }
void SetRXAPanelCopy(int channel, int copy) { WDSP_SETTER(qh_rxa_SetRXAPanelCopy(L.c->eng, 0, copy)); }
void SetRXAAMDSBMode(int channel, int sbmode) { WDSP_SETTER(qh_rxa_SetRXAAMDSBMode(L.c->eng, 0, sbmode)); }
void SetRXAAMDRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXAAMDRun(L.c->eng, 0, run)); }
void SetRXAFMLimRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXAFMLimRun(L.c->eng, 0, run)); }
void SetRXAFMLimGain(int channel, double gaindB) { WDSP_SETTER(qh_rxa_SetRXAFMLimGain(L.c->eng, 0, gaindB)); }
// EMNR, wdsp/emnr.c:1096-1143.  Its tables come from where WDSP takes them when a channel is created: GG / GGS from the file `calculus`
// in the working directory, else the arrays compiled in (emnr.c:317-328); zetaHat from `zetaHat.bin` in the working directory, else the
// compiled-in one (readZetaHat, emnr.c:207-238) -- each file on its own.  (Quisk runs from its own directory and the files lie in wdsp/:
// it gets the compiled-in tables from WDSP, and the same numbers from here.)  $QH_WDSP_DATA, a directory looked into first, is this
// library's addition.
static bool read_file_into(const char *name, const std::function<bool(FILE *)> &take)
{
    const char *dir = std::getenv("QH_WDSP_DATA");
    for (int pass = dir ? 0 : 1; pass < 2; pass++) {
        const std::string path = pass == 0 ? std::string(dir) + "/" + name : std::string(name);
        if (FILE *f = std::fopen(path.c_str(), "rb")) {
            const bool ok = take(f);
            std::fclose(f);
            if (ok) return true;
        }
    }
    return false;
}
static bool load_emnr_tables(Chan &c)
{
    std::vector<double> gg(2 * 241 * 241), zeta(3600);
    std::vector<int> valid(3600);
    double range[4];
    if (!read_file_into("calculus", [&](FILE *f) { return std::fread(gg.data(), 8, gg.size(), f) == gg.size(); })) {
        std::memcpy(gg.data(), qh::kEmnrDefaultGG, 241 * 241 * 8);
        std::memcpy(gg.data() + 241 * 241, qh::kEmnrDefaultGGS, 241 * 241 * 8);
    }
    if (!read_file_into("zetaHat.bin", [&](FILE *f) {
            int dims[2];
            return std::fread(dims, 4, 2, f) == 2 && dims[0] == 60 && dims[1] == 60 && std::fread(range, 8, 4, f) == 4 &&
                   std::fread(zeta.data(), 8, 3600, f) == 3600 && std::fread(valid.data(), 4, 3600, f) == 3600; })) {
        std::memcpy(range, qh::kEmnrDefaultRange, sizeof range);
        std::memcpy(zeta.data(), qh::kEmnrDefaultZeta, 3600 * 8);
        std::memcpy(valid.data(), qh::kEmnrDefaultValid, 3600 * 4);
    }
    return qh_rxa_SetEMNRTables(c.eng, gg.data(), gg.data() + 241 * 241, zeta.data(), valid.data(), range[0], range[1], range[2], range[3]) == QH_OK;
}
void SetRXAEMNRRun(int channel, int run)
{
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return;
    if (run && !L.c->emnr_tables) {
        L.c->emnr_tables = load_emnr_tables(*L.c);
        if (!L.c->emnr_tables) { g_status = QH_ERR_INVALID; return; }       // (qh_rxa_SetEMNRTables has said why)
    }
    const int rc = qh_rxa_SetRXAEMNRRun(L.c->eng, 0, run);
    if (rc) g_status = rc;
}
void SetRXAEMNRnpeMethod(int channel, int method) { WDSP_SETTER(qh_rxa_SetRXAEMNRnpeMethod(L.c->eng, 0, method)); }
void SetRXAEMNRaeRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXAEMNRaeRun(L.c->eng, 0, run)); }
void SetRXAEMNRPosition(int channel, int position) { WDSP_SETTER(qh_rxa_SetRXAEMNRPosition(L.c->eng, 0, position)); }
void SetRXAEMNRaeZetaThresh(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAEMNRaeZetaThresh(L.c->eng, 0, v)); }
void SetRXAEMNRaePsi(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAEMNRaePsi(L.c->eng, 0, v)); }
void SetRXAEMNRtrainZetaThresh(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAEMNRtrainZetaThresh(L.c->eng, 0, v)); }
void SetRXAEMNRtrainT2(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAEMNRtrainT2(L.c->eng, 0, v)); }
// the AM squelch, wdsp/amsq.c:216-243
void SetRXAAMSQRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXAAMSQRun(L.c->eng, 0, run)); }
void SetRXAAMSQThreshold(int channel, double threshold) { WDSP_SETTER(qh_rxa_SetRXAAMSQThreshold(L.c->eng, 0, threshold)); }
void SetRXAAMSQMaxTail(int channel, double tail) { WDSP_SETTER(qh_rxa_SetRXAAMSQMaxTail(L.c->eng, 0, tail)); }
// the LMS auto-notch / noise reduction, wdsp/anf.c:175-239, anr.c:175-238
void SetRXAANFRun(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANFRun(L.c->eng, 0, v)); }
void SetRXAANFTaps(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANFTaps(L.c->eng, 0, v)); }
void SetRXAANFDelay(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANFDelay(L.c->eng, 0, v)); }
void SetRXAANFPosition(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANFPosition(L.c->eng, 0, v)); }
void SetRXAANFGain(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAANFGain(L.c->eng, 0, v)); }
void SetRXAANFLeakage(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAANFLeakage(L.c->eng, 0, v)); }
void SetRXAANFVals(int channel, int taps, int delay, double gain, double leakage) { WDSP_SETTER(qh_rxa_SetRXAANFVals(L.c->eng, 0, taps, delay, gain, leakage)); }
void SetRXAANRRun(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANRRun(L.c->eng, 0, v)); }
void SetRXAANRTaps(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANRTaps(L.c->eng, 0, v)); }
void SetRXAANRDelay(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANRDelay(L.c->eng, 0, v)); }
void SetRXAANRPosition(int channel, int v) { WDSP_SETTER(qh_rxa_SetRXAANRPosition(L.c->eng, 0, v)); }
void SetRXAANRGain(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAANRGain(L.c->eng, 0, v)); }
void SetRXAANRLeakage(int channel, double v) { WDSP_SETTER(qh_rxa_SetRXAANRLeakage(L.c->eng, 0, v)); }
void SetRXAANRVals(int channel, int taps, int delay, double gain, double leakage) { WDSP_SETTER(qh_rxa_SetRXAANRVals(L.c->eng, 0, taps, delay, gain, leakage)); }

// the notch database, wdsp/nbp.c:358-525: int results are the reference's (0 / -1)
int RXANBPAddNotch(int channel, int notch, double fcenter, double fwidth, int active)
{
    int rval = -1;
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return -1;
    const int rc = qh_rxa_RXANBPAddNotch(L.c->eng, 0, notch, fcenter, fwidth, active, &rval);
    if (rc) g_status = rc;
    return rval;
}
int RXANBPGetNotch(int channel, int notch, double *fcenter, double *fwidth, int *active)
{
    int rval = -1;
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return -1;
    const int rc = qh_rxa_RXANBPGetNotch(L.c->eng, 0, notch, fcenter, fwidth, active, &rval);
    if (rc) g_status = rc;
    return rval;
}
int RXANBPDeleteNotch(int channel, int notch)
{
    int rval = -1;
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return -1;
    const int rc = qh_rxa_RXANBPDeleteNotch(L.c->eng, 0, notch, &rval);
    if (rc) g_status = rc;
    return rval;
}
int RXANBPEditNotch(int channel, int notch, double fcenter, double fwidth, int active)
{
    int rval = -1;
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return -1;
    const int rc = qh_rxa_RXANBPEditNotch(L.c->eng, 0, notch, fcenter, fwidth, active, &rval);
    if (rc) g_status = rc;
    return rval;
}
void RXANBPGetNumNotches(int channel, int *nnotches) { WDSP_SETTER(qh_rxa_RXANBPGetNumNotches(L.c->eng, 0, nnotches)); }
void RXANBPGetMinNotchWidth(int channel, double *minwidth) { WDSP_SETTER(qh_rxa_RXANBPGetMinNotchWidth(L.c->eng, 0, minwidth)); }
void RXANBPSetTuneFrequency(int channel, double tunefreq) { WDSP_SETTER(qh_rxa_RXANBPSetTuneFrequency(L.c->eng, 0, tunefreq)); }
void RXANBPSetShiftFrequency(int channel, double shift) { WDSP_SETTER(qh_rxa_RXANBPSetShiftFrequency(L.c->eng, 0, shift)); }
void RXANBPSetNotchesRun(int channel, int run) { WDSP_SETTER(qh_rxa_RXANBPSetNotchesRun(L.c->eng, 0, run)); }
void RXANBPSetWindow(int channel, int wintype) { WDSP_SETTER(qh_rxa_RXANBPSetWindow(L.c->eng, 0, wintype)); }
void RXANBPSetAutoIncrease(int channel, int autoincr) { WDSP_SETTER(qh_rxa_RXANBPSetAutoIncrease(L.c->eng, 0, autoincr)); }
void SetRXAAMDFadeLevel(int channel, int levelfade) { WDSP_SETTER(qh_rxa_SetRXAAMDFadeLevel(L.c->eng, 0, levelfade)); }
void SetRXAFMDeviation(int channel, double deviation) { WDSP_SETTER(qh_rxa_SetRXAFMDeviation(L.c->eng, 0, deviation)); }
void SetRXACTCSSFreq(int channel, double freq) { WDSP_SETTER(qh_rxa_SetRXACTCSSFreq(L.c->eng, 0, freq)); }
void SetRXACTCSSRun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXACTCSSRun(L.c->eng, 0, run)); }

double GetRXAMeter(int channel, int mt)
{
    g_status = QH_OK;
    Locked L(channel);
    if (!L.c) return -400.0;
    double v = -400.0;
    int rc = qh_rxa_GetRXAMeter(L.c->eng, 0, mt, &v);
    if (rc) g_status = rc;
    return v;
}

// ---- quisk_wdsp.c:12-69: re-block an arbitrary count into in_size blocks through a ring, CLIP32 scaling
namespace {
struct Shim {
    std::vector<double> buf; int sizeBuf = 0, nBuf = 0, in_size = 0, in_use = 0, W = 0, R = 0;
    // qh_wdsp_fexchange0_device: the ring in device memory (d_cap samples; `dev`: it holds the ring's contents, not `buf`)
    double *d_buf = nullptr; int d_cap = 0; bool dev = false;
    int device = -1;                    // where d_buf and the events live (the device that was current when the ring went there)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
};
Shim g_shim[kMaxChannels];

// the shim's ring to the side that is going to use it (sizeBuf samples of it are in use), grown to `want` samples
int shim_side(Shim &s, bool dev, int want)
{
    if (dev) {
        if (s.device < 0) (void)hipGetDevice(&s.device);
        if (want > s.d_cap) {
            double *nb = nullptr;
            if (hipMalloc((void **)&nb, (size_t)want * 16) != hipSuccess) return qh::set_error(QH_ERR_HIP, "wdspFexchange0: ring allocation failed");
            (void)hipDeviceSynchronize();
            (void)qh::dev_zero(nb, (size_t)want * 16);
            if (s.dev && s.sizeBuf > 0) { (void)hipMemcpy(nb, s.d_buf, (size_t)s.sizeBuf * 16, hipMemcpyDeviceToDevice); (void)hipDeviceSynchronize(); }
            if (s.d_buf) (void)hipFree(s.d_buf);
            s.d_buf = nb; s.d_cap = want;
        }
        if (!s.dev && s.sizeBuf > 0 && hipMemcpy(s.d_buf, s.buf.data(), (size_t)s.sizeBuf * 16, hipMemcpyHostToDevice) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "wdspFexchange0: ring upload failed");
    } else if (s.dev && s.sizeBuf > 0) {
        if (s.buf.size() < (size_t)s.sizeBuf * 2) s.buf.resize((size_t)s.sizeBuf * 2);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(s.buf.data(), s.d_buf, (size_t)s.sizeBuf * 16, hipMemcpyDeviceToHost) != hipSuccess)
            return qh::set_error(QH_ERR_HIP, "wdspFexchange0: ring download failed");
    }
    s.dev = dev;
    return QH_OK;
}
}

// CloseChannel: the shim's ring goes back to host memory (quisk_wdsp.c's statics outlive a channel: the ring keeps its samples for the
// next OpenChannel on that id), its device buffer and events are released
extern "C" void qh_wdsp_shim_release_device(int channel)
{
    if (!valid(channel)) return;
    Shim &s = g_shim[channel];
    // (the ring's device, not whatever is current in the caller's thread)
    int cur = -1;
    const bool moved = s.device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != s.device && hipSetDevice(s.device) == hipSuccess;
    // a ring that cannot be brought back starts again empty: `dev` may not stay set over a freed buffer
    if (s.dev && shim_side(s, false, 0) != QH_OK) { s.sizeBuf = 0; s.W = 0; s.R = 0; s.nBuf = 0; }
    s.dev = false;
    if (s.d_buf) { (void)hipDeviceSynchronize(); (void)hipFree(s.d_buf); s.d_buf = nullptr; s.d_cap = 0; }
    if (s.ev_in) { (void)hipEventDestroy(s.ev_in); s.ev_in = nullptr; }
    if (s.ev_out) { (void)hipEventDestroy(s.ev_out); s.ev_out = nullptr; }
    s.device = -1;
    if (moved) (void)hipSetDevice(cur);
}

void qh_wdsp_set_parameter(int channel, int in_size, int in_use)
{
    if (!valid(channel)) return;
    Shim &s = g_shim[channel];
    // The reference keeps its ring's length across a change of in_size (quisk_wdsp.c:44-49 only ever grows it): a length that is not a
    // multiple of the new block lets Rindex step past the end and fexchange0 read beyond the allocation.  Here the ring starts again.
    if (in_size > 0 && in_size != s.in_size) { s.in_size = in_size; s.sizeBuf = 0; s.W = 0; s.R = 0; s.nBuf = 0; }
    if (in_use >= 0) s.in_use = in_use;
}

// for qh_quisk_process_samples: the block size of the channel while quisk_wdsp.c's hand-off is in use, else 0
int qh_wdsp_shim_in_size(int channel)
{
    if (channel < 0 || channel >= kMaxChannels) return 0;
    return g_shim[channel].in_use && g_shim[channel].in_size > 0 ? g_shim[channel].in_size : 0;
}

int wdspFexchange0(int channel, double *cSamples, int nSamples)
{
    const double CLIP32 = 2147483647.0;
    if (!valid(channel)) return nSamples;
    Shim &s = g_shim[channel];
    if (!s.in_use) { s.W = 0; s.R = 0; s.nBuf = 0; return nSamples; }
    if (nSamples <= 0 || s.in_size <= 0) return nSamples;
    const int in_size = s.in_size;
    if (s.dev && shim_side(s, false, 0)) return 0;
    int i = nSamples / in_size + 3;                 // blocks needed for the samples plus a partial block
    if (i * in_size > s.sizeBuf) { s.sizeBuf = i * in_size; s.buf.resize((size_t)s.sizeBuf * 2); }
    for (i = 0; i < nSamples; i++) {
        s.buf[2 * (size_t)s.W] = cSamples[2 * i] / CLIP32;
        s.buf[2 * (size_t)s.W + 1] = cSamples[2 * i + 1] / CLIP32;
        if (++s.W >= s.sizeBuf) s.W = 0;
    }
    s.nBuf += nSamples;
    int nout = 0, error = 0;
    while (s.nBuf >= in_size) {
        fexchange0(channel, s.buf.data() + 2 * (size_t)s.R, cSamples + 2 * (size_t)nout, &error);
        s.R += in_size;
        if (s.R >= s.sizeBuf) s.R = 0;
        nout += in_size;
        s.nBuf -= in_size;
    }
    for (i = 0; i < nout; i++) { cSamples[2 * i] *= CLIP32; cSamples[2 * i + 1] *= CLIP32; }
    return nout;
}

// The same hand-off for a caller whose samples are on the GPU (qh_quisk_process_samples, quisk.c:2660-2661): cSamples -> d_samples
// (room for nSamples + in_size samples), everything enqueued: the shim's ring, fexchange0's rings and the DSP blocks in device
// memory on the channel's stream, which is ordered behind `stream` on entry; `stream` is ordered behind it on return.  Nothing
// is waited for.  A channel may change between this and the host-pointer calls in mid-stream (the rings move with it).
int qh_wdsp_fexchange0_device(int channel, void *d_samples, int nSamples, void *stream)
{
    g_status = QH_OK;
    if (!valid(channel)) return nSamples;
    Shim &s = g_shim[channel];
    if (!s.in_use) { s.W = 0; s.R = 0; s.nBuf = 0; return nSamples; }
    if (nSamples <= 0 || s.in_size <= 0) return nSamples;
    if (!d_samples) { g_status = qh::set_error(QH_ERR_INVALID, "qh_wdsp_fexchange0_device: null buffer"); return 0; }
    const int in_size = s.in_size;
    Locked L(channel);
    hipStream_t cs = (hipStream_t)stream;
    double2 *x = reinterpret_cast<double2 *>(d_samples);
    if (!L.c || L.c->in_size != in_size) {
        // no such channel: fexchange0 reports -1 and leaves `out` alone, the shim scales what is there (quisk_wdsp.c:57-66)
        if (L.c) g_status = qh::set_error(QH_ERR_INVALID, "WDSP channel %d: in_size %d, the shim's %d", channel, L.c->in_size, in_size);
        s.nBuf += nSamples;
        const int nout = s.nBuf / in_size * in_size;
        s.nBuf -= nout;
        if (nout > 0) hipLaunchKernelGGL(shim_scale_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, cs, x, nout);
        return nout;
    }
    Chan &c = *L.c;
    if (int rc = set_mode(c, true)) { g_status = rc; return 0; }
    int blocks = nSamples / in_size + 3;
    const int want = blocks * in_size > s.sizeBuf ? blocks * in_size : s.sizeBuf;
    if (int rc = shim_side(s, true, want)) { g_status = rc; return 0; }
    s.sizeBuf = want;
    hipStream_t es = (hipStream_t)qh_rxa_stream(c.eng);
    if (!s.ev_in && (hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming) != hipSuccess)) {
        g_status = qh::set_error(QH_ERR_HIP, "qh_wdsp_fexchange0_device: event creation failed");
        return 0;
    }
    if (cs != es) { (void)hipEventRecord(s.ev_in, cs); (void)hipStreamWaitEvent(es, s.ev_in, 0); }
    hipLaunchKernelGGL(shim_in_kernel, dim3((unsigned)((nSamples + 255) / 256)), dim3(256), 0, es, (const double2 *)x, nSamples,
                       reinterpret_cast<double2 *>(s.d_buf), s.W, s.sizeBuf);         // (the ring is longer than the call: one wrap at most)
    s.W = (s.W + nSamples) % s.sizeBuf;
    s.nBuf += nSamples;
    int nout = 0, error = 0;
    while (s.nBuf >= in_size) {
        if (int rc = fexchange0_dev(c, s.d_buf + 2 * (size_t)s.R, reinterpret_cast<double *>(x + nout), &error)) g_status = rc;
        s.R += in_size;
        if (s.R >= s.sizeBuf) s.R = 0;
        nout += in_size;
        s.nBuf -= in_size;
    }
    if (nout > 0) hipLaunchKernelGGL(shim_scale_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, es, x, nout);
    if (hipGetLastError() != hipSuccess) g_status = qh::set_error(QH_ERR_HIP, "qh_wdsp_fexchange0_device: launch failed");
    if (cs != es) { (void)hipEventRecord(s.ev_out, es); (void)hipStreamWaitEvent(cs, s.ev_out, 0); }
    return nout;
}

// xpanel never looks at its run flag (wdsp/patchpanel.c:55-101): accepted, no effect on the data.
void SetRXAPanelRun(int channel, int run) { (void)run; g_status = QH_OK; (void)valid(channel); }
void RXASetMP(int channel, int mp) { WDSP_SETTER(qh_rxa_RXASetMP(L.c->eng, 0, mp)); }     // wdsp/RXA.c:948-958
void SetRXAEMNRgainMethod(int channel, int method) { WDSP_SETTER(qh_rxa_SetRXAEMNRgainMethod(L.c->eng, 0, method)); }      // emnr.c:1112

// fexchange2 (wdsp/iobuffs.c:518-582): the same exchange with separate float I and Q buffers (INREAL / OUTREAL are
// float, wdsp/comm.h:119-120); upslew2 / downslew2 are the slews of fexchange0 on that layout
void fexchange2(int channel, float *Iin, float *Qin, float *Iout, float *Qout, int *error)
{
    g_status = QH_OK;
    *error = 0;
    int in_size = 0, out_size = 0;
    {
        Locked L(channel);
        if (!L.c) { *error = -1; return; }
        if (!L.c->exchange) return;                 // outputs are left untouched, like fexchange0
        in_size = L.c->in_size; out_size = L.c->out_size;
    }
    std::vector<double> in((size_t)in_size * 2), out((size_t)out_size * 2);
    for (int i = 0; i < in_size; i++) { in[2 * (size_t)i] = (double)Iin[i]; in[2 * (size_t)i + 1] = (double)Qin[i]; }
    fexchange0(channel, in.data(), out.data(), error);
    for (int i = 0; i < out_size; i++) { Iout[i] = (float)out[2 * (size_t)i]; Qout[i] = (float)out[2 * (size_t)i + 1]; }
}

void SetRXASNBARun(int channel, int run) { WDSP_SETTER(qh_rxa_SetRXASNBARun(L.c->eng, 0, run)); }          // wdsp/snb.c:579-593
void SetRXASNBAOutputBandwidth(int channel, double flow, double fhigh) { WDSP_SETTER(qh_rxa_SetRXASNBAOutputBandwidth(L.c->eng, 0, flow, fhigh)); }  // snb.c:660-694
void SetRXASNBAasize(int channel, int size) { WDSP_SETTER(qh_rxa_SetRXASNBAasize(L.c->eng, 0, size)); }                  // snb.c:604
void SetRXASNBAnpasses(int channel, int npasses) { WDSP_SETTER(qh_rxa_SetRXASNBAnpasses(L.c->eng, 0, npasses)); }        // snb.c:611
void SetRXASNBAk1(int channel, double k1) { WDSP_SETTER(qh_rxa_SetRXASNBAk1(L.c->eng, 0, k1)); }                          // snb.c:618
void SetRXASNBAk2(int channel, double k2) { WDSP_SETTER(qh_rxa_SetRXASNBAk2(L.c->eng, 0, k2)); }                          // snb.c:625
void SetRXASNBAbridge(int channel, int bridge) { WDSP_SETTER(qh_rxa_SetRXASNBAbridge(L.c->eng, 0, bridge)); }             // snb.c:632
void SetRXASNBApresamps(int channel, int presamps) { WDSP_SETTER(qh_rxa_SetRXASNBApresamps(L.c->eng, 0, presamps)); }     // snb.c:639
void SetRXASNBApostsamps(int channel, int postsamps) { WDSP_SETTER(qh_rxa_SetRXASNBApostsamps(L.c->eng, 0, postsamps)); } // snb.c:646
void SetRXASNBAovrlp(int channel, int ovrlp) { WDSP_SETTER(qh_rxa_SetRXASNBAovrlp(L.c->eng, 0, ovrlp)); }                    // snb.c:595
void SetRXASNBApmultmin(int channel, double pmultmin) { WDSP_SETTER(qh_rxa_SetRXASNBApmultmin(L.c->eng, 0, pmultmin)); }  // snb.c:653

}  // extern "C"
