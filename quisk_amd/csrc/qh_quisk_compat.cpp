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
// one single-channel bank per distinct (taps, decimation), at most kCacheCap of them: quisk_filt_tune rewrites cpxCoefs in place,
// so every retune is a new checksum; the entry used longest ago makes room
constexpr size_t kCacheCap = 32;
unsigned long long g_tick = 0;
struct FirEntry { qh_fir *bank; unsigned long long used; };
std::map<Key, FirEntry> g_banks;

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
    if (it != g_banks.end()) { it->second.used = ++g_tick; return it->second.bank; }
    if (g_banks.size() >= kCacheCap) {
        auto old = g_banks.begin();
        for (auto e = g_banks.begin(); e != g_banks.end(); ++e) if (e->second.used < old->second.used) old = e;
        qh_fir_destroy(old->second.bank);
        g_banks.erase(old);
    }
    qh_fir *b;
    if (cplx) {
        std::vector<double> re((size_t)ntaps), im((size_t)ntaps);
        for (int i = 0; i < ntaps; i++) { re[(size_t)i] = re_or_interleaved[2 * i]; im[(size_t)i] = re_or_interleaved[2 * i + 1]; }
        b = qh_fir_create(0, 1, re.data(), im.data(), ntaps, decim, QH_F64, nullptr);
    } else {
        b = qh_fir_create(0, 1, re_or_interleaved, nullptr, ntaps, decim, QH_F64, nullptr);
    }
    if (b) g_banks[k] = FirEntry{ b, ++g_tick };
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
    // a decim_index left behind by a LARGER factor (the same struct used with another decim): the reference's `++decim_index >= decim`
    // then fires at the first sample (filter.c:213,241,269), which is what the last phase does
    const int phase = f->decim_index < 0 ? 0 : f->decim_index >= decim ? decim - 1 : f->decim_index;
    const int nout = run_block(b, nt > 1 ? hist.data() : nullptr, phase, x, count, decim);
    f->ptcSamp = f->cSamples + 2 * w;
    f->decim_index = (phase + count) % decim;
    return nout;
}


// ---- the rest of filter.h rides on the polyphase resampler (qh_polyphase.hip)

constexpr int kOutCap = 66000 * 8 / 10;     // SAMP_BUFFER_SIZE * 8 / 10, quisk.h:15, filter.c:158

struct RatKey {
    const void *taps; int ntaps, interp, decim; unsigned long long sum;
    bool operator<(const RatKey &o) const
    {
        if (taps != o.taps) return taps < o.taps;
        if (ntaps != o.ntaps) return ntaps < o.ntaps;
        if (interp != o.interp) return interp < o.interp;
        if (decim != o.decim) return decim < o.decim;
        return sum < o.sum;
    }
};
struct RatEntry { qh_rat *rat; unsigned long long used; };
std::map<RatKey, RatEntry> g_rats;

qh_rat *rat_for(const double *taps, int ntaps, int interp, int decim)
{
    RatKey k{ taps, ntaps, interp, decim, checksum(taps, (size_t)ntaps) };
    auto it = g_rats.find(k);
    if (it != g_rats.end()) { it->second.used = ++g_tick; return it->second.rat; }
    if (g_rats.size() >= kCacheCap) {
        auto old = g_rats.begin();
        for (auto e = g_rats.begin(); e != g_rats.end(); ++e) if (e->second.used < old->second.used) old = e;
        qh_rat_destroy(old->second.rat);
        g_rats.erase(old);
    }
    qh_rat *b = qh_rat_create(0, 1, taps, ntaps, interp, decim, QH_F64, nullptr);
    if (b) g_rats[k] = RatEntry{ b, ++g_tick };
    return b;
}

// A view of the reference's circular history: `width` doubles per entry (2 = complex, 1 = real).
struct Ring {
    double *base; int width, n, pos;
    void history(int need, std::vector<double> &hist) const     // the `need` newest entries, oldest first, as complex
    {
        hist.assign((size_t)(need > 0 ? need : 1) * 2, 0.0);
        for (int k = 0; k < need; k++) {
            int idx = (pos - need + k) % n;
            if (idx < 0) idx += n;
            hist[2 * (size_t)k] = base[(size_t)width * idx];
            if (width == 2) hist[2 * (size_t)k + 1] = base[2 * (size_t)idx + 1];
        }
    }
    void push(const double *x, int xwidth, int count)           // every input in order from the write slot
    {
        for (int i = 0; i < count; i++) {
            base[(size_t)width * pos] = x[(size_t)xwidth * i];
            if (width == 2) base[2 * (size_t)pos + 1] = xwidth == 2 ? x[2 * (size_t)i + 1] : 0.0;
            if (++pos >= n) pos = 0;
        }
    }
};

// x holds `count` samples of `width` doubles; returns the outputs (complex, interleaved) of one polyphase call
int rat_block(qh_rat *b, const std::vector<double> &hist, int phase, const double *x, int width, int count, std::vector<double> &out)
{
    if (qh_rat_set_state(b, hist.data(), phase)) return 0;
    std::vector<double> in((size_t)count * 2);
    for (int i = 0; i < count; i++) { in[2 * (size_t)i] = x[(size_t)width * i]; in[2 * (size_t)i + 1] = width == 2 ? x[2 * (size_t)i + 1] : 0.0; }
    const int cap = qh_rat_out_count(b, count);
    out.assign((size_t)(cap > 0 ? cap : 1) * 2, 0.0);
    int nout = 0;
    if (qh_rat_process_host(b, in.data(), count, count, out.data(), cap > 0 ? cap : 1, &nout)) return 0;
    return nout;
}

// quisk_cInterpolate / quisk_dInterpolate / quisk_cInterpDecim on a caller-owned filter struct
int interp_decim(double *x, int width, int count, struct quisk_cFilter *f, int interp, int decim, bool keep_phase)
{
    if (count <= 0 || !f || interp <= 0 || decim <= 0 || !f->dCoefs) return 0;
    std::lock_guard<std::mutex> lk(g_mtx);
    const int used = (f->nTaps / interp) * interp;              // filter.c:149,308: nTaps / interp taps per phase
    if (used <= 0) {
        // fewer taps than phases: the reference's inner loop runs nTaps / interp = 0 times and every output is 0.0 -- their count, the
        // ring and the phase move on all the same (filter.c:146-163, 303-322)
        if (f->nTaps <= 0) return 0;
        Ring r{ f->cSamples, width, f->nTaps, (int)((f->ptcSamp - f->cSamples) / width) };
        r.push(x, width, count);
        f->ptcSamp = f->cSamples + (size_t)width * r.pos;
        long long nout = 0;
        if (keep_phase) {
            int di = f->decim_index;
            for (int i = 0; i < count; i++) { while (di < interp) { nout++; di += decim; } di -= interp; }
            f->decim_index = di;
        } else nout = (long long)count * interp;
        if (nout > kOutCap) nout = kOutCap;
        std::memset(x, 0, (size_t)nout * width * sizeof(double));
        return (int)nout;
    }
    qh_rat *b = rat_for(f->dCoefs, used, interp, decim);
    if (!b) return 0;
    Ring r{ f->cSamples, width, f->nTaps, (int)((f->ptcSamp - f->cSamples) / width) };
    std::vector<double> hist, out;
    r.history(used / interp - 1, hist);
    const int phase = keep_phase ? f->decim_index : 0;
    int nout = rat_block(b, hist, phase, x, width, count, out);
    r.push(x, width, count);
    f->ptcSamp = f->cSamples + (size_t)width * r.pos;
    if (keep_phase) f->decim_index = qh_rat_phase(b);
    if (nout > kOutCap) nout = kOutCap;                         // filter.c:158,315
    for (int i = 0; i < nout; i++) { x[(size_t)width * i] = out[2 * (size_t)i]; if (width == 2) x[2 * (size_t)i + 1] = out[2 * (size_t)i + 1]; }
    return nout;
}

// quisk_dDecimate / quisk_dFilter / quisk_dD_out / quisk_dC_out: real history ring, FIR bank as for the complex ones
int real_decimate(double *x, int count, struct quisk_cFilter *f, int decim, bool cplx_taps, double *cplx_out)
{
    if (count <= 0 || !f || decim <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mtx);
    const double *taps = cplx_taps ? f->cpxCoefs : f->dCoefs;
    if (!taps) { qh::set_error(QH_ERR_INVALID, "filter has no %s coefficients", cplx_taps ? "complex" : "real"); return 0; }
    qh_fir *b = bank_for(taps, f->nTaps, decim, cplx_taps);
    if (!b) return 0;
    Ring r{ f->cSamples, 1, f->nTaps, (int)(f->ptcSamp - f->cSamples) };
    std::vector<double> hist;
    r.history(f->nTaps - 1, hist);
    std::vector<double> in((size_t)count * 2, 0.0);
    for (int i = 0; i < count; i++) in[2 * (size_t)i] = x[i];
    r.push(x, 1, count);
    const int phase = f->decim_index < 0 ? 0 : f->decim_index >= decim ? decim - 1 : f->decim_index;     // (as in decimate() above)
    const int nout = run_block(b, f->nTaps > 1 ? hist.data() : nullptr, phase, in.data(), count, decim);
    f->ptcSamp = f->cSamples + r.pos;
    f->decim_index = (phase + count) % decim;
    for (int i = 0; i < nout; i++) {
        if (cplx_out) { cplx_out[2 * i] = in[2 * (size_t)i]; cplx_out[2 * i + 1] = in[2 * (size_t)i + 1]; }
        else x[i] = in[2 * (size_t)i];
    }
    return nout;
}

// quisk_cInterp2HB45 / quisk_dInterp2HB45: samples[22] newest first is the whole state (filter.c:437-452)
int interp2_hb45(double *x, int width, int count, double *samples)
{
    if (count <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mtx);
    static double taps[45];
    static bool have = false;
    if (!have) {
        double t[43];
        qh_hb45_taps(t);
        for (double &v : taps) v = 0.0;
        for (int k = 0; k < 11; k++) { taps[2 * k + 1] = t[2 * k]; taps[43 - 2 * k] = t[2 * k]; }
        taps[22] = 0.5;
        have = true;
    }
    qh_rat *b = rat_for(taps, 45, 2, 1);
    if (!b) return 0;
    std::vector<double> hist(22 * 2, 0.0), out;
    for (int d = 0; d < 22; d++) {                              // samples[d] has delay d + 1 at the next input
        hist[2 * (size_t)(21 - d)] = samples[(size_t)width * d];
        if (width == 2) hist[2 * (size_t)(21 - d) + 1] = samples[2 * (size_t)d + 1];
    }
    int nout = rat_block(b, hist, 0, x, width, count, out);
    // the delay line after the call: the 22 newest of (old line, inputs)
    std::vector<double> line((size_t)(22 + count) * width);
    for (int d = 0; d < 22; d++)
        for (int w = 0; w < width; w++) line[(size_t)(21 - d) * width + w] = samples[(size_t)width * d + w];
    std::memcpy(line.data() + (size_t)22 * width, x, (size_t)count * width * sizeof(double));
    for (int d = 0; d < 22; d++)
        for (int w = 0; w < width; w++) samples[(size_t)width * d + w] = line[(size_t)(22 + count - 1 - d) * width + w];
    // filter.c:444-446: a pair is written while nOut <= cap
    const int cap_pairs = kOutCap / 2 + 1;
    if (nout > 2 * cap_pairs) nout = 2 * cap_pairs;
    for (int i = 0; i < nout; i++) { x[(size_t)width * i] = out[2 * (size_t)i]; if (width == 2) x[2 * (size_t)i + 1] = out[2 * (size_t)i + 1]; }
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

// struct quisk_dFilter (filter.h:12-21) has the layout of struct quisk_cFilter with a real history ring.
void quisk_filt_dInit(struct quisk_cFilter *filter, double *coefs, int taps)      // filter.c:22-33
{
    filter->dCoefs = coefs;
    filter->cpxCoefs = nullptr;
    filter->cSamples = (double *)std::calloc((size_t)taps, sizeof(double));
    filter->ptcSamp = filter->cSamples;
    filter->nTaps = taps;
    filter->decim_index = 0;
    filter->cBuf = nullptr;
    filter->nBuf = 0;
}

void quisk_filt_differInit(struct quisk_cFilter *filter, int taps)                // filter.c:35-56 (without its printf)
{
    double *c = (double *)std::malloc((size_t)taps * sizeof(double));
    for (int k = -(taps - 1) / 2; k <= (taps - 1) / 2; k++)
        c[(taps - 1) / 2 + k] = k == 0 ? 0.0 : std::pow(-1.0, k) / k;
    quisk_filt_dInit(filter, c, taps);
}

int quisk_cInterpolate(double *cSamples, int count, struct quisk_cFilter *filter, int interp)
{
    return interp_decim(cSamples, 2, count, filter, interp, 1, false);
}

int quisk_dInterpolate(double *dSamples, int count, struct quisk_cFilter *filter, int interp)
{
    return interp_decim(dSamples, 1, count, filter, interp, 1, false);
}

int quisk_cInterpDecim(double *cSamples, int count, struct quisk_cFilter *filter, int interp, int decim)
{
    return interp_decim(cSamples, 2, count, filter, interp, decim, true);
}

int quisk_dDecimate(double *dSamples, int count, struct quisk_cFilter *filter, int decim)
{
    return real_decimate(dSamples, count, filter, decim, false, nullptr);
}

int quisk_dFilter(double *dSamples, int count, struct quisk_cFilter *filter)
{
    if (!filter) return 0;
    const int keep = filter->decim_index;           // quisk_dFilter does not touch decim_index (filter.c:347-370)
    const int n = real_decimate(dSamples, count, filter, 1, false, nullptr);
    filter->decim_index = keep;
    return n;
}

double quisk_dD_out(double sample, struct quisk_cFilter *filter)                  // filter.c:326-345
{
    if (!filter) return 0.0;
    const int keep = filter->decim_index;
    double v = sample;
    real_decimate(&v, 1, filter, 1, false, nullptr);
    filter->decim_index = keep;
    return v;
}

void qh_quisk_dC_out(double sample, struct quisk_cFilter *filter, double *out_re_im)   // filter.c:83-104, result by pointer
{
    out_re_im[0] = out_re_im[1] = 0.0;
    if (!filter) return;
    const int keep = filter->decim_index;
    double v = sample;
    real_decimate(&v, 1, filter, 1, true, out_re_im);
    filter->decim_index = keep;
}

// the reference's own name and return type (filter.c:83): clang spells C's complex double in C++ as an extension, and the C ABI
// returns it in two floating-point registers either way
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wreturn-type-c-linkage"
double _Complex quisk_dC_out(double sample, struct quisk_cFilter *filter)
{
    double o[2];
    qh_quisk_dC_out(sample, filter, o);
    double _Complex r;
    __real__ r = o[0];
    __imag__ r = o[1];
    return r;
}
#pragma clang diagnostic pop

int quisk_cInterp2HB45(double *cSamples, int count, struct quisk_cHB45Filter *filter)
{
    return filter ? interp2_hb45(cSamples, 2, count, filter->samples) : 0;
}

int quisk_dInterp2HB45(double *dSamples, int count, struct quisk_dHB45Filter *filter)
{
    return filter ? interp2_hb45(dSamples, 1, count, filter->samples) : 0;
}

}  // extern "C"
