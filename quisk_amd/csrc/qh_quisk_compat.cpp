// qh_quisk_compat.cpp -- filter.h drop-in exports (include/quiskhip.h group 4).
//
// quisk.c keeps every filter's state in a caller-owned struct (struct quisk_cFilter, filter.h:1-10) and calls
// the primitives on small blocks from the sound thread.  These wrappers keep that contract: state in, block
// through the GPU FIR bank (qh_fir.hip), state out in the reference's ring format.  They exist for link
// compatibility and parity testing; throughput work should use the batched qh_fir_* / qh_rxa_* API.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"

namespace {

struct Key {
    const void *taps; int ntaps, decim, cplx; unsigned long long sum;
    bool operator<(const Key &o) const
    {
        if (taps != o.taps) return taps < o.taps;
        if (ntaps != o.ntaps) return ntaps < o.ntaps;
        if (decim != o.decim) return decim < o.decim;
        if (cplx != o.cplx) return cplx < o.cplx;
        return sum < o.sum;
    }
};

std::mutex g_mtx;
std::map<Key, qh_fir *> g_banks;            // one single-channel bank per distinct (taps, decimation)

unsigned long long checksum(const double *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        unsigned long long v;
        std::memcpy(&v, p + i, 8);
        h = (h ^ v) * 1099511628211ull;
    }
    return h;
}

qh_fir *bank_for(const double *re_or_interleaved, int ntaps, int decim, bool cplx)
{
    Key k{ re_or_interleaved, ntaps, decim, cplx ? 1 : 0, checksum(re_or_interleaved, (size_t)ntaps * (cplx ? 2 : 1)) };
    auto it = g_banks.find(k);
    if (it != g_banks.end()) return it->second;
    qh_fir *b;
    if (cplx) {
        std::vector<double> re((size_t)ntaps), im((size_t)ntaps);
        for (int i = 0; i < ntaps; i++) { re[(size_t)i] = re_or_interleaved[2 * i]; im[(size_t)i] = re_or_interleaved[2 * i + 1]; }
        b = qh_fir_create(0, 1, re.data(), im.data(), ntaps, decim, QH_F64, nullptr);
    } else {
        b = qh_fir_create(0, 1, re_or_interleaved, nullptr, ntaps, decim, QH_F64, nullptr);
    }
    if (b) g_banks[k] = b;
    return b;
}

// run `count` samples through a bank whose state is (hist oldest-first [ntaps-1], phase); returns outputs in x
int run_block(qh_fir *b, const double *hist, int phase, double *x, int count, int decim)
{
    if (qh_fir_set_state(b, hist, phase)) return 0;
    int nout = 0;
    const int cap = (phase + count) / decim;
    std::vector<double> out((size_t)(cap > 0 ? cap : 1) * 2);
    if (qh_fir_process_host(b, x, count, count, out.data(), cap > 0 ? cap : 1, &nout)) return 0;
    std::memcpy(x, out.data(), (size_t)nout * 2 * sizeof(double));
    return nout;
}

int decimate(double *x, int count, struct quisk_cFilter *f, int decim, bool cplx)
{
    if (count <= 0 || !f || decim <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mtx);
    const double *taps = cplx ? f->cpxCoefs : f->dCoefs;
    if (!taps) { qh::set_error(QH_ERR_INVALID, "filter has no %s coefficients", cplx ? "complex" : "real"); return 0; }
    qh_fir *b = bank_for(taps, f->nTaps, decim, cplx);
    if (!b) return 0;
    const int nt = f->nTaps;
    // history, oldest first: the ring holds the last nTaps samples; ptcSamp is the next write slot, so the
    // newest sample is at ptcSamp - 1 (filter.c:214-226)
    const int pos = (int)((f->ptcSamp - f->cSamples) / 2);
    std::vector<double> hist((size_t)(nt > 1 ? nt - 1 : 1) * 2);
    for (int k = 0; k < nt - 1; k++) {
        int idx = pos - (nt - 1) + k;                   // k = 0 is the oldest of the nt-1 we need
        idx %= nt; if (idx < 0) idx += nt;
        hist[2 * (size_t)k] = f->cSamples[2 * idx];
        hist[2 * (size_t)k + 1] = f->cSamples[2 * idx + 1];
    }
    // the reference's ring after the call: every input sample written in order from ptcSamp
    int w = pos;
    for (int i = 0; i < count; i++) {
        f->cSamples[2 * w] = x[2 * i];
        f->cSamples[2 * w + 1] = x[2 * i + 1];
        if (++w >= nt) w = 0;
    }
    const int phase = f->decim_index;
    const int nout = run_block(b, nt > 1 ? hist.data() : nullptr, phase, x, count, decim);
    f->ptcSamp = f->cSamples + 2 * w;
    f->decim_index = (phase + count) % decim;
    return nout;
}

}  // namespace

extern "C" {

void quisk_filt_cInit(struct quisk_cFilter *filter, double *coefs, int taps)
{
    filter->dCoefs = coefs;
    filter->cpxCoefs = nullptr;
    filter->cSamples = (double *)std::calloc((size_t)taps * 2, sizeof(double));
    filter->ptcSamp = filter->cSamples;
    filter->nTaps = taps;
    filter->decim_index = 0;
    filter->cBuf = nullptr;
    filter->nBuf = 0;
}

void quisk_filt_tune(struct quisk_cFilter *filter, double freq, int ssb_upper)
{
    if (!filter->cpxCoefs) filter->cpxCoefs = (double *)std::malloc((size_t)filter->nTaps * 2 * sizeof(double));
    const double w = 2.0 * M_PI * freq;
    const double D = (filter->nTaps - 1.0) / 2.0;
    for (int i = 0; i < filter->nTaps; i++) {
        const double a = w * (i - D);
        const double re = std::cos(a) * filter->dCoefs[i], im = std::sin(a) * filter->dCoefs[i];
        if (ssb_upper) { filter->cpxCoefs[2 * i] = re; filter->cpxCoefs[2 * i + 1] = im; }
        else           { filter->cpxCoefs[2 * i] = im; filter->cpxCoefs[2 * i + 1] = re; }
    }
}

int quisk_cDecimate(double *cSamples, int count, struct quisk_cFilter *filter, int decim)
{
    return decimate(cSamples, count, filter, decim, false);
}

int quisk_cCDecimate(double *cSamples, int count, struct quisk_cFilter *filter, int decim)
{
    return decimate(cSamples, count, filter, decim, true);
}

int quisk_cFilter(double *cSamples, int count, struct quisk_cFilter *filter)
{
    return decimate(cSamples, count, filter, 1, false);
}

int quisk_cDecim2HB45(double *x, int count, struct quisk_cHB45Filter *f)
{
    if (count <= 0 || !f) return 0;
    std::lock_guard<std::mutex> lk(g_mtx);
    static double taps[43];
    static bool have = false;
    if (!have) { qh_hb45_taps(taps); have = true; }
    qh_fir *b = bank_for(taps, 43, 2, false);
    if (!b) return 0;
    // Chronological history from the two delay lines (filter.c:391-399): samples[] holds the inputs that
    // arrived with toggle == 1 (newest first), center[] those with toggle == 0.  center[] keeps 11 entries;
    // older even-phase inputs only ever meet zero taps, so they are filled with zeros.
    double hist[42 * 2];
    for (int d = 0; d < 42; d++) {                      // d = delay of the history sample (0 = newest)
        double re = 0.0, im = 0.0;
        const bool in_samples = f->toggle == 0 ? (d % 2 == 0) : (d % 2 == 1);
        const int j = d / 2;
        if (in_samples) { if (j < 22) { re = f->samples[2 * j]; im = f->samples[2 * j + 1]; } }
        else if (j < 11) { re = f->center[2 * j]; im = f->center[2 * j + 1]; }
        hist[2 * (41 - d)] = re; hist[2 * (41 - d) + 1] = im;
    }
    // the reference's delay lines after the call
    int toggle = f->toggle;
    for (int i = 0; i < count; i++) {
        if (toggle == 0) {
            toggle = 1;
            std::memmove(f->center + 2, f->center, sizeof(double) * 2 * 10);
            f->center[0] = x[2 * i]; f->center[1] = x[2 * i + 1];
        } else {
            toggle = 0;
            std::memmove(f->samples + 2, f->samples, sizeof(double) * 2 * 21);
            f->samples[0] = x[2 * i]; f->samples[1] = x[2 * i + 1];
        }
    }
    const int phase = f->toggle;                        // toggle 1 == one sample consumed since the last output
    f->toggle = toggle;
    return run_block(b, hist, phase, x, count, 2);
}

}  // extern "C"
