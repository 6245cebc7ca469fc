// qh_design.cpp -- see qh_design.hpp.
#include "qh_design.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace qh {

static const double kPi = 3.1415926535897932;      // wdsp/comm.h:146 (the reference's literals)
static const double kTwoPi = 6.2831853071795864;   // wdsp/comm.h:147

static double bh_window(int wintype, double cosphi)
{
    if (wintype == 0)       // 4-term Blackman-Harris, wdsp/fir.c:220-225
        return 0.21747 + cosphi * (-0.45325 + cosphi * (0.28256 + cosphi * (-0.04672)));
    // 7-term Blackman-Harris as a polynomial in cos, wdsp/fir.c:227-236
    return 6.3964424114390378e-02
         + cosphi * (-2.3993864599352804e-01
         + cosphi * (3.5015956323820469e-01
         + cosphi * (-2.4774111897080783e-01
         + cosphi * (8.5438256055858031e-02
         + cosphi * (-1.2320203369293225e-02
         + cosphi * (4.3778825791773474e-04))))));
}

// Windowed-sinc band-pass (what wdsp/fir.c:187-254 designs).  In closed form, with x = n - (N - 1) / 2 the tap's distance from the centre,
//
//     h[n] = scale * w[n] * sin(2 pi B x) / (pi x) * exp(-j w0 x),      B = (f_high - f_low) / (2 fs),   w0 = pi (f_high + f_low) / fs
//
// i.e. a low-pass of half-width B under a Blackman-Harris window, moved to the band's centre (the real part alone for rtype 0).  The
// low-pass times the window is real and even in x: the upper half is evaluated, the lower half mirrored, and one pass turns the
// oscillator on.  (The window is evaluated from cos(pi n / m) at the UPPER index of each mirrored pair, as the reference does; the
// window is symmetric, so this only fixes which of two equal-to-rounding values is used.)
std::vector<cd> fir_bandpass(int N, double f_low, double f_high, double samplerate, int wintype, int rtype, double scale)
{
    const double half_width = (f_high - f_low) / (2.0 * samplerate);
    const double centre = kPi * (f_high + f_low) / samplerate;
    const double mid = 0.5 * (double)(N - 1);
    std::vector<double> lowpass((size_t)N, 0.0);
    if (N & 1) lowpass[(size_t)(N >> 1)] = scale * 2.0 * half_width;               // the limit of sin(2 pi B x) / (pi x) at x = 0; the window is 1 there
    for (int n = (N + 1) / 2; n < N; n++) {
        const double x = (double)n - mid;
        const double v = scale * (std::sin(kTwoPi * half_width * x) / (kPi * x)) * bh_window(wintype, std::cos((kPi / mid) * n));
        lowpass[(size_t)n] = v;
        lowpass[(size_t)(N - 1 - n)] = v;
    }
    std::vector<cd> h((size_t)N);
    for (int n = 0; n < N; n++) {
        const double x = (double)n - mid, v = lowpass[(size_t)n];
        h[(size_t)n] = rtype == 0 ? cd(v * std::cos(x * centre), 0.0) : cd(v * std::cos(x * centre), -v * std::sin(x * centre));
    }
    return h;
}

ResamplerDesign design_resampler(int in_rate, int out_rate, double fc, int ncoef, double gain)
{
    ResamplerDesign d;
    int x = in_rate, y = out_rate;
    while (y != 0) { int z = y; y = x % y; x = z; }
    d.L = out_rate / x;
    d.M = in_rate / x;
    const int min_rate = in_rate < out_rate ? in_rate : out_rate;
    if (fc == 0.0) fc = 0.45 * (double)min_rate;
    const double full_rate = (double)in_rate * d.L;
    const double fc_norm_high = fc / full_rate;
    const double fc_norm_low = -fc_norm_high;
    if (ncoef == 0) ncoef = (int)(140.0 * full_rate / min_rate);
    ncoef = (ncoef / d.L + 1) * d.L;
    d.ncoef = ncoef;
    d.cpp = ncoef / d.L;
    std::vector<cd> imp = fir_bandpass(ncoef, fc_norm_low, fc_norm_high, 1.0, 1, 0, gain * (double)d.L);
    d.h.resize((size_t)ncoef);
    for (int i = 0; i < ncoef; i++) d.h[i] = imp[i].real();
    return d;
}

static std::vector<double> fsamp_window(int N, int wintype)
{
    std::vector<double> w((size_t)N, 1.0);
    if (wintype == 0 || wintype == 1) {
        const double arg0 = 2.0 * kPi / ((double)N - 1.0);
        for (int i = 0; i < N; i++) w[i] = bh_window(wintype, std::cos(arg0 * (double)i));
    }
    return w;
}

// The FM de-emphasis / pre-emphasis curve as an FIR (what wdsp/fcurve.c:29-145 builds through fir_fsamp, wdsp/fir.c:129-185), even nc.
//
// Wanted magnitude at the nc / 2 bin centres f_i = (i + 1/2) / (nc / 2) * fs / 2: a 6 dB / octave line through (f0, g0 dB), rising
// (curve 0) or falling (curve 1).  With ctfmode 0 the line holds between f0 and f1 only; outside, bin by bin, the magnitude is the
// one inside times (f_k / f_edge)^4 resp. (f_edge / f_k)^4 with f_k = k / (nc / 2) -- a running product, so the skirts fall faster
// than a fourth-order slope (the reference's own shape, kept; floor 1e-100).
// The taps are the linear-phase frequency-sampling design: h[n] = (A_0 + 2 sum_{k=1}^{nc/2-1} A_k cos(2 pi (n - M) k / nc)) / nc with
// M = (nc - 1) / 2, even about M (one half evaluated), under a Blackman-Harris window.
std::vector<cd> fc_impulse(int nc, double f0, double f1, double g0, double /*g1*/, int curve, double samplerate,
                           double scale, int ctfmode, int wintype)
{
    if (nc & 1) throw std::runtime_error("fc_impulse: odd nc is not used by the RXA chain");
    const int bins = nc / 2;
    const double nyquist = samplerate / 2.0, line = scale * std::pow(10.0, g0 / 20.0);
    std::vector<double> mag((size_t)bins + 1, 0.0);
    for (int i = 0; i < bins; i++) {
        const double f = ((double)i + 0.5) / (double)bins * nyquist;
        mag[(size_t)i] = curve == 0 ? (f0 > 0.0 ? line * f / f0 : 0.0) : (f > 0.0 ? line * f0 / f : 0.0);
    }
    if (ctfmode == 0) {
        const int lo = (int)(f0 / nyquist * bins - 0.5), hi = (int)(f1 / nyquist * bins - 0.5);
        auto pow4 = [](double v) { return v * v * v * v; };
        const double lo4 = std::pow((double)lo / (double)bins, 4.0), hi4 = std::pow((double)hi / (double)bins, 4.0);
        double skirt = mag[(size_t)lo];
        for (int k = lo - 1; k >= 0; k--) {                  // below the band: each bin the one above it times (f_k / f_lo)^4
            skirt = std::max(skirt * (pow4((double)k / (double)bins) / lo4), 1.0e-100);
            mag[(size_t)k] = skirt;
        }
        skirt = mag[(size_t)hi];
        for (int k = hi + 1; k < bins; k++) {                // above it: each bin the one below it times (f_hi / f_k)^4
            skirt = std::max(skirt * (hi4 / pow4((double)k / (double)bins)), 1.0e-100);
            mag[(size_t)k] = skirt;
        }
    }
    const double centre = (double)(nc - 1) / 2.0;
    const std::vector<double> w = fsamp_window(nc, wintype);
    std::vector<cd> h((size_t)nc);
    for (int n = 0; n < bins; n++) {
        double acc = 0.0;
        for (int k = 1; k < bins; k++) acc += 2.0 * mag[(size_t)k] * std::cos(kTwoPi * (n - centre) * k / nc);
        const double tap = (1.0 / nc) * (mag[0] + acc);
        h[(size_t)n] = cd(tap * w[(size_t)n], 0.0);
        h[(size_t)(nc - 1 - n)] = cd(tap * w[(size_t)(nc - 1 - n)], 0.0);
    }
    return h;
}

void host_fft(std::vector<cd> &x, int sign)
{
    const int n = (int)x.size();
    if (n <= 1) return;
    if (n & (n - 1)) throw std::runtime_error("host_fft: size must be a power of two");
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    std::vector<cd> tw((size_t)n / 2);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < n / 2; k++) {
        long double a = (sign < 0 ? -2.0L : 2.0L) * pi * (long double)k / (long double)n;
        tw[k] = cd((double)cosl(a), (double)sinl(a));
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < half; k++) {
                cd t = x[i + k + half] * tw[(size_t)k * step];
                cd u = x[i + k];
                x[i + k] = u + t;
                x[i + k + half] = u - t;
            }
    }
}

// make_nbp (wdsp/nbp.c:97-179): the passband [flow, fhigh] minus the active notches, as a list of sub-bands.
// A notch narrower than minwidth is widened about its centre when autoincr is set.  Sub-bands are produced in the
// reference's order (a split appends the upper part at the end), which fixes the summation order of fir_mbandpass.
std::vector<std::pair<double, double>> make_nbp(const std::vector<Notch> &notches, double minwidth, int autoincr, double flow,
                                                double fhigh, bool *havnotch)
{
    std::vector<std::pair<double, double>> bp;
    if (havnotch) *havnotch = false;
    if (!(fhigh > flow)) return bp;
    bp.emplace_back(flow, fhigh);
    for (const Notch &n : notches) {
        double nl = n.fcenter - 0.5 * n.fwidth, nh = n.fcenter + 0.5 * n.fwidth;       // nlow / nhigh, nbp.c:374-375
        if (autoincr && n.fwidth < minwidth) { nl = n.fcenter - 0.5 * minwidth; nh = n.fcenter + 0.5 * minwidth; }
        if (!(n.active && nh > flow && nl < fhigh)) continue;
        if (havnotch) *havnotch = true;
        const size_t nbp = bp.size();
        std::vector<char> del(nbp, 0);
        for (size_t i = 0; i < nbp; i++) {
            if (!(nh > bp[i].first && nl < bp[i].second)) continue;
            if (nl <= bp[i].first && nh >= bp[i].second) del[i] = 1;
            else if (nl > bp[i].first && nh < bp[i].second) { bp.emplace_back(nh, bp[i].second); bp[i].second = nl; }
            else if (nl <= bp[i].first && nh > bp[i].first) bp[i].first = nh;
            else if (nl < bp[i].second && nh >= bp[i].second) bp[i].second = nl;
        }
        // the reference compacts in place while it scans (nbp.c:161-172): an entry that slides into a slot already
        // visited is not looked at again, and del[] belongs to positions, not entries -- restated as is
        del.resize(bp.size(), 0);
        size_t total = bp.size(), nnbp = total;
        for (size_t i = 0; i < total; i++) {
            if (del[i] == 1) {
                nnbp--;
                for (size_t j = i; j < nnbp; j++) bp[j] = bp[j + 1];
                del[i] = 0;
            }
        }
        bp.resize(nnbp);
    }
    return bp;
}

// fir_mbandpass (wdsp/nbp.c:64-80): sum of the band-pass impulse responses of the sub-bands
std::vector<cd> fir_mbandpass(int N, const std::vector<std::pair<double, double>> &bands, double rate, double scale, int wintype)
{
    std::vector<cd> h((size_t)N, cd(0.0, 0.0));
    for (const auto &b : bands) {
        const std::vector<cd> imp = fir_bandpass(N, b.first, b.second, rate, wintype, 1, scale);
        for (int i = 0; i < N; i++) h[(size_t)i] += imp[(size_t)i];
    }
    return h;
}

// mp_imp (wdsp/fir.c:319-368) with analytic() (fir.c:292-317): the minimum-phase impulse response with the
// magnitude response of `fir`, by the cepstral method on a grid of pfactor * N points (N * pfactor a power of two).
std::vector<cd> mp_imp(const std::vector<cd> &fir, int pfactor, int polarity)
{
    const int N = (int)fir.size(), size = N * pfactor;
    const double inv_PN = 1.0 / (double)size;
    std::vector<cd> freq((size_t)size, cd(0.0, 0.0));
    for (int i = 0; i < N; i++) freq[(size_t)i] = fir[(size_t)i];
    host_fft(freq, -1);
    std::vector<double> mag((size_t)size);
    std::vector<cd> ana((size_t)size);
    for (int i = 0; i < size; i++) {
        mag[(size_t)i] = std::sqrt(freq[(size_t)i].real() * freq[(size_t)i].real() + freq[(size_t)i].imag() * freq[(size_t)i].imag()) * inv_PN;
        ana[(size_t)i] = cd(mag[(size_t)i] > 0.0 ? std::log(mag[(size_t)i]) : std::log(1.0e-300), 0.0);
    }
    // analytic(): forward FFT, keep DC and Nyquist once, double the positive bins, zero the negative ones, inverse
    host_fft(ana, -1);
    const double inv_N = 1.0 / (double)size, two_inv_N = 2.0 * inv_N;
    ana[0] *= inv_N;
    for (int i = 1; i < size / 2; i++) ana[(size_t)i] *= two_inv_N;
    ana[(size_t)size / 2] *= inv_N;
    for (int i = size / 2 + 1; i < size; i++) ana[(size_t)i] = cd(0.0, 0.0);
    host_fft(ana, +1);
    std::vector<cd> nf((size_t)size);
    for (int i = 0; i < size; i++) {
        const double ph = ana[(size_t)i].imag();
        nf[(size_t)i] = cd(mag[(size_t)i] * std::cos(ph), (polarity ? 1.0 : -1.0) * mag[(size_t)i] * std::sin(ph));
    }
    host_fft(nf, +1);
    std::vector<cd> out((size_t)N);
    const size_t off = polarity ? (size_t)(pfactor - 1) * (size_t)N : 0;
    for (int i = 0; i < N; i++) out[(size_t)i] = nf[off + (size_t)i];
    return out;
}

// forward DFT of 3 * 2^k points: the three decimated sequences x[3 m + a] by host_fft, then X[k + M q] = sum_a W_N^(a (k + M q)) F_a[k]
static void host_fft_3x(std::vector<cd> &x)
{
    const int N = (int)x.size(), M = N / 3;
    std::vector<cd> f[3];
    for (int a = 0; a < 3; a++) {
        f[a].resize((size_t)M);
        for (int m = 0; m < M; m++) f[a][(size_t)m] = x[(size_t)(3 * m + a)];
        host_fft(f[a], -1);
    }
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < N; k++) {
        std::complex<long double> acc(0, 0);
        for (int a = 0; a < 3; a++) {
            const long double ang = -2.0L * pi * (long double)(((long long)a * k) % N) / (long double)N;
            acc += std::complex<long double>(f[a][(size_t)(k % M)]) * std::complex<long double>(cosl(ang), sinl(ang));
        }
        x[(size_t)k] = cd((double)acc.real(), (double)acc.imag());
    }
}

std::vector<cd> make_mask(const std::vector<cd> &h, int nfft)
{
    if ((int)h.size() > nfft) throw std::runtime_error("make_mask: impulse longer than the FFT");
    std::vector<cd> m((size_t)nfft, cd(0, 0));
    for (size_t i = 0; i < h.size(); i++) m[i] = h[i];
    if (nfft % 3 == 0) host_fft_3x(m);
    else host_fft(m, -1);
    const double s = 1.0 / (double)nfft;
    for (auto &v : m) v *= s;
    return m;
}

std::vector<cd> fft_twiddle_table(int n)
{
    // (Ns, R) of every pass after the first, in order; must match qh::FftRR<N>
    std::vector<std::pair<int, int>> passes;
    switch (n) {
    case 4096: passes = { {16, 16}, {256, 16} }; break;
    case 2048: passes = { {8, 4}, {32, 8}, {256, 8} }; break;
    case 1024: passes = { {4, 4}, {16, 4}, {64, 4}, {256, 4} }; break;
    case 512:  passes = { {2, 16}, {32, 8}, {256, 2} }; break;
    case 8192: passes = { {32, 8}, {256, 32} }; break;
    default: throw std::runtime_error("fft_twiddle_table: unsupported size");
    }
    std::vector<cd> t;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (auto &p : passes) {
        const int Ns = p.first, R = p.second;
        for (int k = 0; k < Ns; k++) {
            long double a = -2.0L * pi * (long double)k / ((long double)Ns * (long double)R);
            t.emplace_back((double)cosl(a), (double)sinl(a));
        }
    }
    return t;
}

}  // namespace qh
