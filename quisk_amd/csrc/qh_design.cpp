// qh_design.cpp -- see qh_design.hpp.
#include "qh_design.hpp"
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

std::vector<cd> fir_bandpass(int N, double f_low, double f_high, double samplerate, int wintype, int rtype, double scale)
{
    std::vector<cd> h((size_t)N, cd(0, 0));
    const double ft = (f_high - f_low) / (2.0 * samplerate);
    const double ft_rad = kTwoPi * ft;
    const double w_osc = kPi * (f_high + f_low) / samplerate;
    const double m = 0.5 * (double)(N - 1);
    const double delta = kPi / m;
    if (N & 1) h[N >> 1] = cd(scale * 2.0 * ft, 0.0);
    for (int i = (N + 1) / 2, j = N / 2 - 1; i < N; i++, j--) {
        const double posi = (double)i - m, posj = (double)j - m;
        const double sinc = std::sin(ft_rad * posi) / (kPi * posi);
        const double coef = scale * sinc * bh_window(wintype, std::cos(delta * i));
        if (rtype == 0) {
            h[i] = cd(coef * std::cos(posi * w_osc), 0.0);
            h[j] = cd(coef * std::cos(posj * w_osc), 0.0);
        } else {
            h[i] = cd(coef * std::cos(posi * w_osc), -coef * std::sin(posi * w_osc));
            h[j] = cd(coef * std::cos(posj * w_osc), -coef * std::sin(posj * w_osc));
        }
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

std::vector<cd> fc_impulse(int nc, double f0, double f1, double g0, double /*g1*/, int curve, double samplerate,
                           double scale, int ctfmode, int wintype)
{
    if (nc & 1) throw std::runtime_error("fc_impulse: odd nc is not used by the RXA chain");
    const int mid = nc / 2;
    std::vector<double> A((size_t)mid + 1, 0.0);
    const double g0_lin = std::pow(10.0, g0 / 20.0);
    for (int i = 0; i < mid; i++) {
        const double fn = ((double)i + 0.5) / (double)mid;
        const double f = fn * samplerate / 2.0;
        if (curve == 0) A[i] = (f0 > 0.0) ? scale * (g0_lin * f / f0) : 0.0;
        else            A[i] = (f > 0.0) ? scale * (g0_lin * f0 / f) : 0.0;
    }
    if (ctfmode == 0) {
        const int low = (int)(2.0 * f0 / samplerate * mid - 0.5);
        const int high = (int)(2.0 * f1 / samplerate * mid - 0.5);
        double lowmag = A[low], highmag = A[high];
        const double flow4 = std::pow((double)low / (double)mid, 4.0);
        const double fhigh4 = std::pow((double)high / (double)mid, 4.0);
        int k = low;
        while (--k >= 0) {
            const double f = (double)k / (double)mid;
            lowmag *= (f * f * f * f) / flow4;
            if (lowmag < 1.0e-100) lowmag = 1.0e-100;
            A[k] = lowmag;
        }
        k = high;
        while (++k < mid) {
            const double f = (double)k / (double)mid;
            highmag *= fhigh4 / (f * f * f * f);
            if (highmag < 1.0e-100) highmag = 1.0e-100;
            A[k] = highmag;
        }
    }
    // fir_fsamp, even N, rtype 1, scale 1 (wdsp/fir.c:129-185)
    const int N = nc;
    std::vector<cd> h((size_t)N, cd(0, 0));
    const double M = (double)(N - 1) / 2.0;
    for (int n = 0; n < N / 2; n++) {
        double sum = 0.0;
        for (int k = 1; k < N / 2; k++) sum += 2.0 * A[k] * std::cos(kTwoPi * (n - M) * k / N);
        h[n] = cd((1.0 / N) * (A[0] + sum), 0.0);
    }
    for (int n = N / 2, j = 1; n < N; n++, j++) h[n] = cd(h[N / 2 - j].real(), 0.0);
    std::vector<double> w = fsamp_window(N, wintype);
    for (int i = 0; i < N; i++) h[i] = cd(h[i].real() * (1.0 * w[i]), 0.0);
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
