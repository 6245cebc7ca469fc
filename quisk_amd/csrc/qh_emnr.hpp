// qh_emnr.hpp -- WDSP's EMNR ("NR2", wdsp/emnr.c) for many channels: one workgroup per channel.
//
// EMNR is an overlap-add STFT (4096-point frames every 1024 samples, sqrt-Hamming analysis and synthesis windows) whose work
// per frame is a set of recursions that are independent per frequency bin -- minimum-statistics noise power (LambdaD,
// emnr.c:604-739) or the speech-presence estimator (LambdaDs, :741-754), decision-directed a-priori SNR and the gain rule
// (calc_gain, :885-1013: Gaussian-amplitude MMSE, log-MMSE via E1, the 241 x 241 gamma-speech tables, or the trained zeta
// tables), the artifact-elimination post-filter (aepf, :777-816: a moving average over bins) -- tied together by six sums
// over all bins.  The 256 lanes of a workgroup walk the 2049 bins (block sums through LDS), the two 4096-point transforms are
// TileFft<4096> on (x, 0) pairs, and the frame / block bookkeeping of xemnr (emnr.c:1015-1068) is kept literally, so a call
// may hold any number of DSP blocks.  Per-channel state lives in global memory in the reference's arrays.
#pragma once
#include "qh_fft.hpp"
#include "qh_wave.hpp"

namespace qh {

constexpr int kEmnrF = 4096, kEmnrM = 2049, kEmnrIncr = 1024, kEmnrU = 8;

struct EmnrParam {      // calc_emnr (emnr.c:240-497) for one sample rate; the same for every channel of an engine
    double gain, gf1p5, alpha, eps_floor, gamma_max, xi_min, q, gmax, z_gamma_min, z_gamma_max, z_xihat_min, z_xihat_max;
    double alphaCsmooth, alphaMax, alphaCmin, alphaMin_max_value, snrq, betamax, invQeqMax, av, MofD, MofV, invQbar_points[4], nsmax[4];
    double alpha_pow, alpha_Pbar, epsH1, epsH1r;
    double l_eta, l_gamma, l_beta, l_alpha_d, l_alpha_p, delta_LF, delta_MF;        // npl, emnr.c:458-489
    int U, V, D, dim_zeta, bsize, oasize, init_oainidx, pad;
};
struct EmnrChan { int gain_method, npe_method, ae_run, pad; double zetaThresh, psi, zeta_thresh, t2; };      // per channel (emnr.c:1112-1174)
struct EmnrScalars { int iainidx, iaoutidx, oainidx, oaoutidx, nsamps, saveidx, subwc, amb_idx; double alphaC; };

// per-channel arrays, all doubles, in one block of kEmnrStateDoubles
constexpr int kEmnrPad = 2052;          // msize rounded up
enum EmnrOff {
    EO_INACC = 0, EO_OUTACC = EO_INACC + kEmnrF, EO_SAVE = EO_OUTACC + kEmnrIncr, EO_PREVG = EO_SAVE + 4 * kEmnrF, EO_PREVM = EO_PREVG + kEmnrPad,
    EO_P = EO_PREVM + kEmnrPad, EO_SIG = EO_P + kEmnrPad, EO_PBAR = EO_SIG + kEmnrPad, EO_P2BAR = EO_PBAR + kEmnrPad, EO_ACTMIN = EO_P2BAR + kEmnrPad,
    EO_ACTSUB = EO_ACTMIN + kEmnrPad, EO_PMINU = EO_ACTSUB + kEmnrPad, EO_LMIN = EO_PMINU + kEmnrPad, EO_AMB = EO_LMIN + kEmnrPad,
    EO_SSIG = EO_AMB + kEmnrU * kEmnrPad, EO_SPBAR = EO_SSIG + kEmnrPad, EO_LP = EO_SPBAR + kEmnrPad, EO_LPMIN = EO_LP + kEmnrPad,
    EO_LPP = EO_LPMIN + kEmnrPad, EO_LD = EO_LPP + kEmnrPad, EO_END = EO_LD + kEmnrPad
};
constexpr int kEmnrStateDoubles = EO_END;

__device__ __forceinline__ double emnr_bessI0(double x)         // emnr.c:43-82
{
    if (x == 0.0) return 1.0;
    if (x < 0.0) x = -x;
    if (x <= 3.75) {
        double p = x / 3.75; p = p * p;
        return ((((( 0.0045813 * p + 0.0360768) * p + 0.2659732) * p + 1.2067492) * p + 3.0899424) * p + 3.5156229) * p + 1.0;
    }
    const double p = 3.75 / x;
    return exp(x) / sqrt(x) * (((((((( + 0.00392377 * p - 0.01647633) * p + 0.02635537) * p - 0.02057706) * p + 0.00916281) * p
           - 0.00157565) * p + 0.00225319) * p + 0.01328592) * p + 0.39894228);
}
__device__ __forceinline__ double emnr_bessI1(double x)         // emnr.c:84-124
{
    if (x == 0.0) return 0.0;
    if (x < 0.0) x = -x;
    if (x <= 3.75) {
        double p = x / 3.75; p = p * p;
        return x * (((((( 0.00032411 * p + 0.00301532) * p + 0.02658733) * p + 0.15084934) * p + 0.51498869) * p + 0.87890594) * p + 0.5);
    }
    const double p = 3.75 / x;
    return exp(x) / sqrt(x) * (((((((( - 0.00420059 * p + 0.01787654) * p - 0.02895312) * p + 0.02282967) * p - 0.01031555) * p
           + 0.00163801) * p - 0.00362018) * p - 0.03988024) * p + 0.39894228);
}
__device__ __forceinline__ double emnr_e1xb(double x)           // emnr.c:132-165
{
    if (x == 0.0) return 1.0e300;
    if (x <= 1.0) {
        double e1 = 1.0, r = 1.0;
        for (int k = 1; k <= 25; k++) {
            r = -r * k * x / ((k + 1.0) * (k + 1.0));
            e1 = e1 + r;
            if (fabs(r) <= fabs(e1) * 1.0e-15) break;
        }
        return -0.5772156649015328 - log(x) + x * e1;
    }
    const int m = 20 + (int)(80.0 / x);
    double t0 = 0.0;
    for (int k = m; k >= 1; k--) t0 = (double)k / (1.0 + k / (x + t0));
    return exp(-x) * (1.0 / (x + t0));
}
__device__ __forceinline__ double emnr_getKey(const double *__restrict__ type, double gamma, double xi)   // emnr.c:818-862
{
    int ngamma1, ngamma2, nxi1, nxi2;
    double tg, tx;
    if (gamma <= 0.001) { ngamma1 = ngamma2 = 0; tg = 0.0; }
    else if (gamma >= 1000.0) { ngamma1 = ngamma2 = 240; tg = 60.0; }
    else { tg = 10.0 * log10(gamma / 0.001); ngamma1 = (int)(4.0 * tg); ngamma2 = ngamma1 + 1; }
    if (xi <= 0.001) { nxi1 = nxi2 = 0; tx = 0.0; }
    else if (xi >= 1000.0) { nxi1 = nxi2 = 240; tx = 60.0; }
    else { tx = 10.0 * log10(xi / 0.001); nxi1 = (int)(4.0 * tx); nxi2 = nxi1 + 1; }
    const double dg = (tg - 0.25 * ngamma1) / 0.25, dx = (tx - 0.25 * nxi1) / 0.25;
    return (1.0 - dg) * (1.0 - dx) * type[241 * nxi1 + ngamma1] + (1.0 - dg) * dx * type[241 * nxi2 + ngamma1]
         + dg * (1.0 - dx) * type[241 * nxi1 + ngamma2] + dg * dx * type[241 * nxi2 + ngamma2];
}
__device__ __forceinline__ double emnr_mlog10(double val)       // wdsp/meterlog10.c:29-32,547-554
{
    const unsigned long long N = (unsigned long long)__double_as_longlong(val);
    const int e = (int)((N >> 52) & 2047) - 1023, m = (int)((N >> (52 - 11)) & 2047);
    return 0.301029995663981 * (e + log2(1.0 + m / 2048.0));
}

// sum over the workgroup, the same value in every lane (fixed order: lanes by xor tree, waves 0..3)
__device__ __forceinline__ double emnr_block_sum(double v, double *red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// dynamic LDS: FFT image, then 4 arrays of kEmnrPad doubles (lambda_y, lambda_d, Qeq / nmask, mask)
constexpr int emnr_lds_bytes() { return TileFft<kEmnrF, false, double2>::kLdsBytes + 4 * kEmnrPad * 8; }

static __global__ __launch_bounds__(NT) void emnr_kernel(double2 *buf, long long stride, int nblk, const int *chan_list, EmnrParam q,
                                                         const EmnrChan *chan, EmnrScalars *scal, double *state, const double *window,
                                                         const double2 *tw4096, const double *GG, const double *GGS, const double *zeta_hat,
                                                         const int *zeta_true)
{
    using Fwd = TileFft<kEmnrF, false, double2>;
    using Inv = TileFft<kEmnrF, true, double2>;
    extern __shared__ __align__(16) unsigned char emnr_smem[];
    __shared__ double red[4];
    double *ly = reinterpret_cast<double *>(emnr_smem + Fwd::kLdsBytes), *ld = ly + kEmnrPad, *qa = ld + kEmnrPad, *mk_ = qa + kEmnrPad;
    const int ch = chan_list[blockIdx.x], t = threadIdx.x;
    const EmnrChan cc = chan[ch];
    EmnrScalars sc = scal[ch];
    double *S = state + (long long)ch * kEmnrStateDoubles;
    double2 *p = buf + (long long)ch * stride;
    const int bs = q.bsize, M = kEmnrM;
    for (int b = 0; b < nblk; b++) {
        // in: the block's real parts into the input accumulator (emnr.c:1021-1026)
        for (int i = t; i < bs; i += NT) S[EO_INACC + ((sc.iainidx + i) & (kEmnrF - 1))] = p[(long long)b * bs + i].x;
        sc.iainidx = (sc.iainidx + bs) & (kEmnrF - 1);
        sc.nsamps += bs;
        __syncthreads();
        while (sc.nsamps >= kEmnrF) {
            // ---- analysis window + forward transform
            double2 x[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = t + NT * r;
                x[r] = make_double2(window[i] * S[EO_INACC + ((sc.iaoutidx + i) & (kEmnrF - 1))], 0.0);
            }
            sc.iaoutidx = (sc.iaoutidx + kEmnrIncr) & (kEmnrF - 1);
            sc.nsamps -= kEmnrIncr;
            __syncthreads();
            Fwd::run(x, emnr_smem, Fwd::load(tw4096));
            // lane t holds bins t + 256 r; the noise / gain work is on bins 0 .. 2048: r < 8, and bin 2048 in lane 0 (r = 8)
            // ---- lambda_y and the three sums of LambdaD
            double s_p = 0.0, s_y = 0.0, s_n = 0.0;
#pragma unroll
            for (int r = 0; r < 9; r++) {
                const int k = t + NT * r;
                if (k < M) {
                    const double v = x[r].x * x[r].x + x[r].y * x[r].y;
                    ly[k] = v;
                    s_y += v; s_p += S[EO_P + k]; s_n += S[EO_SIG + k];
                }
            }
            if (cc.npe_method == 0) {
                // ---- LambdaD, emnr.c:604-739
                const double sum_prev_p = emnr_block_sum(s_p, red), sum_lambda_y = emnr_block_sum(s_y, red),
                             sum_prev_sigma2N = emnr_block_sum(s_n, red);
                const double SNR = sum_prev_p / sum_prev_sigma2N;
                const double alphaMin = fmin(q.alphaMin_max_value, pow(SNR, q.snrq));
                const double f1 = sum_prev_p / sum_lambda_y - 1.0;
                const double alphaCtilda = 1.0 / (1.0 + f1 * f1);
                sc.alphaC = q.alphaCsmooth * sc.alphaC + (1.0 - q.alphaCsmooth) * fmax(alphaCtilda, q.alphaCmin);
                const double f2 = q.alphaMax * sc.alphaC;
                double s_iq = 0.0;
                for (int k = t; k < M; k += NT) {
                    const double sig = S[EO_SIG + k];
                    double pk = S[EO_P + k];
                    const double f0 = pk / sig - 1.0;
                    double aopt = 1.0 / (1.0 + f0 * f0);
                    if (aopt < alphaMin) aopt = alphaMin;
                    const double ah = f2 * aopt;
                    pk = ah * pk + (1.0 - ah) * ly[k];
                    S[EO_P + k] = pk;
                    const double beta = fmin(q.betamax, ah * ah);
                    const double pb = beta * S[EO_PBAR + k] + (1.0 - beta) * pk;
                    const double p2 = beta * S[EO_P2BAR + k] + (1.0 - beta) * pk * pk;
                    S[EO_PBAR + k] = pb; S[EO_P2BAR + k] = p2;
                    const double varHat = p2 - pb * pb;
                    double invQeq = varHat / (2.0 * sig * sig);
                    if (invQeq > q.invQeqMax) invQeq = q.invQeqMax;
                    qa[k] = 1.0 / invQeq;
                    s_iq += invQeq;
                }
                const double invQbar = emnr_block_sum(s_iq, red) / (double)M;
                const double bc = 1.0 + q.av * sqrt(invQbar);
                double noise_slope_max = q.nsmax[3];
                if (invQbar < q.invQbar_points[0]) noise_slope_max = q.nsmax[0];
                else if (invQbar < q.invQbar_points[1]) noise_slope_max = q.nsmax[1];
                else if (invQbar < q.invQbar_points[2]) noise_slope_max = q.nsmax[2];
                for (int k = t; k < M; k += NT) {
                    const double Qeq = qa[k], pk = S[EO_P + k];
                    const double QeqTilda = (Qeq - 2.0 * q.MofD) / (1.0 - q.MofD), QeqTildaSub = (Qeq - 2.0 * q.MofV) / (1.0 - q.MofV);
                    const double bmin = 1.0 + 2.0 * (q.D - 1.0) / QeqTilda, bmin_sub = 1.0 + 2.0 * (q.V - 1.0) / QeqTildaSub;
                    double actmin = S[EO_ACTMIN + k], actmin_sub = S[EO_ACTSUB + k], pmin_u = S[EO_PMINU + k], sig = S[EO_SIG + k];
                    int lmin = (int)S[EO_LMIN + k], k_mod = 0;
                    const double f3 = pk * bmin * bc;
                    if (f3 < actmin) { actmin = f3; actmin_sub = pk * bmin_sub * bc; k_mod = 1; }
                    if (sc.subwc == q.V) {
                        if (k_mod) lmin = 0;
                        S[EO_AMB + sc.amb_idx * kEmnrPad + k] = actmin;
                        double mn = 1.0e300;
                        for (int ku = 0; ku < q.U; ku++) { const double v = S[EO_AMB + ku * kEmnrPad + k]; if (v < mn) mn = v; }
                        pmin_u = mn;
                        if (lmin == 1 && actmin_sub < noise_slope_max * pmin_u && actmin_sub > pmin_u) {
                            pmin_u = actmin_sub;
                            for (int ku = 0; ku < q.U; ku++) S[EO_AMB + ku * kEmnrPad + k] = actmin_sub;
                        }
                        lmin = 0; actmin = 1.0e300; actmin_sub = 1.0e300;
                    } else if (sc.subwc > 1 && k_mod) {
                        lmin = 1;
                        sig = fmin(actmin_sub, pmin_u);
                        pmin_u = sig;
                    }
                    S[EO_ACTMIN + k] = actmin; S[EO_ACTSUB + k] = actmin_sub; S[EO_PMINU + k] = pmin_u; S[EO_SIG + k] = sig;
                    S[EO_LMIN + k] = (double)lmin;
                    ld[k] = sig;
                }
                if (sc.subwc == q.V) { if (++sc.amb_idx == q.U) sc.amb_idx = 0; sc.subwc = 1; }
                else ++sc.subwc;
            } else if (cc.npe_method == 2) {
                // ---- LambdaDl, emnr.c:756-775
                const double c = (1.0 - q.l_gamma) / (1.0 - q.l_beta);
                for (int k = t; k < M; k += NT) {
                    const double P_old = S[EO_LP + k];
                    const double P = q.l_eta * P_old + (1.0 - q.l_eta) * ly[k];
                    double Pmin = S[EO_LPMIN + k];
                    if (Pmin < P) Pmin = q.l_gamma * Pmin + c * (P - q.l_beta * P_old); else Pmin = P;
                    const double Sr = P / Pmin;
                    const double delta = (double)k <= q.delta_LF ? 2.0 : (double)k <= q.delta_MF ? 2.0 : 5.0;
                    const double I = Sr > delta ? 1.0 : 0.0;
                    const double pp = q.l_alpha_p * S[EO_LPP + k] + (1.0 - q.l_alpha_p) * I;
                    const double alpha_s = q.l_alpha_d + (1.0 - q.l_alpha_d) * pp;
                    const double Dk = alpha_s * S[EO_LD + k] + (1.0 - alpha_s) * ly[k];
                    S[EO_LP + k] = P; S[EO_LPMIN + k] = Pmin; S[EO_LPP + k] = pp; S[EO_LD + k] = Dk;
                    ld[k] = Dk;
                }
            } else {
                // ---- LambdaDs, emnr.c:741-754
                for (int k = t; k < M; k += NT) {
                    double sig = S[EO_SSIG + k];
                    double PH1y = 1.0 / (1.0 + (1.0 + q.epsH1) * exp(-q.epsH1r * ly[k] / sig));
                    const double Pbar = q.alpha_Pbar * S[EO_SPBAR + k] + (1.0 - q.alpha_Pbar) * PH1y;
                    S[EO_SPBAR + k] = Pbar;
                    if (Pbar > 0.99) PH1y = fmin(PH1y, 0.99);
                    const double EN2y = (1.0 - PH1y) * ly[k] + PH1y * sig;
                    sig = q.alpha_pow * sig + (1.0 - q.alpha_pow) * EN2y;
                    S[EO_SSIG + k] = sig;
                    ld[k] = sig;
                }
            }
            // ---- gain, emnr.c:905-1011
            double s_pre = 0.0, s_post = 0.0;
            for (int k = t; k < M; k += NT) {
                const double lam_y = ly[k], lam_d = ld[k];
                const double gamma = fmin(lam_y / lam_d, q.gamma_max);
                const double pm = S[EO_PREVM + k];
                double eps_hat = q.alpha * pm * pm * S[EO_PREVG + k] + (1.0 - q.alpha) * fmax(gamma - 1.0, q.eps_floor);
                double m;
                if (cc.gain_method == 2) {
                    const double eps_p = eps_hat / (1.0 - q.q);
                    m = emnr_getKey(GG, gamma, eps_hat) * emnr_getKey(GGS, gamma, eps_p);
                    S[EO_PREVM + k] = m;
                } else if (cc.gain_method == 1) {
                    const double ehr = eps_hat / (1.0 + eps_hat), v = ehr * gamma;
                    m = ehr * exp(fmin(700.0, 0.5 * emnr_e1xb(v)));
                    if (m > q.gmax) m = q.gmax;
                    if (m != m) m = 0.01;
                    S[EO_PREVM + k] = m;
                } else {
                    eps_hat = fmax(eps_hat, q.xi_min);
                    const double v = (eps_hat / (1.0 + eps_hat)) * gamma;
                    m = q.gf1p5 * sqrt(v) / gamma * exp(-0.5 * v) * ((1.0 + v) * emnr_bessI0(0.5 * v) + v * emnr_bessI1(0.5 * v));
                    const double v2 = fmin(v, 700.0);
                    {
                        const double eta = m * m * lam_y / lam_d, eps = eta / (1.0 - q.q);
                        const double witchHat = (1.0 - q.q) / q.q * exp(v2) / (1.0 + eps);
                        m *= witchHat / (1.0 + witchHat);
                    }
                    if (m > q.gmax) m = q.gmax;
                    if (m != m) m = 0.01;
                    S[EO_PREVM + k] = m;
                    if (cc.gain_method == 3) {
                        double xi_ts = m * m * gamma;
                        xi_ts = fmax(xi_ts, q.xi_min);
                        const double v_ts = (xi_ts / (1.0 + xi_ts)) * gamma;
                        m = q.gf1p5 * sqrt(v_ts) / gamma * exp(-0.5 * v_ts) * ((1.0 + v_ts) * emnr_bessI0(0.5 * v_ts) + v_ts * emnr_bessI1(0.5 * v_ts));
                        const double eta = m * m * lam_y / lam_d, eps = eta / (1.0 - q.q);
                        const double witchHat = (1.0 - q.q) / q.q * exp(v2) / (1.0 + eps);
                        m *= witchHat / (1.0 + witchHat);
                        // getZeta, emnr.c:864-883 (with its `xi_dB >= dim_zeta` test)
                        const double gamma_dB = 10.0 * emnr_mlog10(gamma), xi_dB = 10.0 * emnr_mlog10(xi_ts);
                        const double gpc = (q.z_gamma_max - q.z_gamma_min) / q.dim_zeta, xpc = (q.z_xihat_max - q.z_xihat_min) / q.dim_zeta;
                        const int i_gamma = (int)floor((gamma_dB - q.z_gamma_min) / gpc), i_xi = (int)floor((xi_dB - q.z_xihat_min) / xpc);
                        if (!(i_gamma < 0 || i_gamma >= q.dim_zeta || i_xi < 0 || xi_dB >= q.dim_zeta)) {
                            const int index = i_gamma * q.dim_zeta + i_xi;
                            if (zeta_true[index] > 0) m = zeta_hat[index] > cc.zeta_thresh ? 1.0 : 0.0;
                        }
                    }
                }
                S[EO_PREVG + k] = gamma;
                mk_[k] = m;
                s_pre += lam_y; s_post += m * m * lam_y;
            }
            if (cc.ae_run) {
                // ---- aepf, emnr.c:777-816
                const double sumPre = emnr_block_sum(s_pre, red), sumPost = emnr_block_sum(s_post, red);
                const double zeta = sumPost / sumPre;
                const double zetaT = zeta >= cc.zetaThresh ? 1.0 : zeta;
                const int N = zetaT == 1.0 ? 1 : 1 + 2 * (int)(0.5 + cc.psi * (1.0 - zetaT / cc.zetaThresh));
                const int n = N / 2;
                const double tail = (cc.gain_method == 3 && zetaT < cc.t2) ? 0.05 : 1.0;
                for (int k = t; k < M; k += NT) {
                    double acc = 0.0, div;
                    if (k < n) { for (int m2 = 0; m2 <= 2 * k; m2++) acc += mk_[m2]; div = (double)(2 * k + 1); }
                    else if (k < M - n) { for (int m2 = k - n; m2 <= k + n; m2++) acc += mk_[m2]; div = (double)N; }
                    else { for (int m2 = M - 1; m2 >= -M + 2 * k + 1; m2--) acc += mk_[m2]; div = (double)(2 * (M - k) - 1); }
                    qa[k] = acc / div * tail;
                }
                __syncthreads();
                for (int k = t; k < M; k += NT) mk_[k] = qa[k];
            }
            __syncthreads();
            // ---- gain * mask on the spectrum (Hermitian: bins above 2048 are this lane's own mirrors), inverse transform
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = t + NT * r, km = k <= 2048 ? k : kEmnrF - k;
                const double g1 = q.gain * mk_[km];
                x[r] = make_double2(g1 * x[r].x, (k == 0 || k == 2048) ? 0.0 : g1 * x[r].y);
            }
            __syncthreads();
            Inv::run(x, emnr_smem, Inv::load(tw4096));
            // ---- synthesis window, save, overlap-add (emnr.c:1042-1058)
            double *sv = S + EO_SAVE + sc.saveidx * kEmnrF;
#pragma unroll
            for (int r = 0; r < 16; r++) { const int i = t + NT * r; sv[i] = window[i] * x[r].x; }
            __syncthreads();
            for (int j = t; j < kEmnrIncr; j += NT) {
                const int kk = (sc.oainidx + j) % q.oasize;
                double acc = 0.0;
                for (int i = 4; i > 0; i--) {
                    const int sbuff = (sc.saveidx + i) & 3, sbegin = kEmnrIncr * (4 - i);
                    const double v = S[EO_SAVE + sbuff * kEmnrF + sbegin + j];
                    acc = i == 4 ? v : acc + v;
                }
                S[EO_OUTACC + kk] = acc;
            }
            sc.saveidx = (sc.saveidx + 1) & 3;
            sc.oainidx = (sc.oainidx + kEmnrIncr) % q.oasize;
            __syncthreads();
        }
        // out (emnr.c:1059-1065)
        for (int i = t; i < bs; i += NT) p[(long long)b * bs + i] = make_double2(S[EO_OUTACC + (sc.oaoutidx + i) % q.oasize], 0.0);
        sc.oaoutidx = (sc.oaoutidx + bs) % q.oasize;
        __syncthreads();
    }
    if (t == 0) scal[ch] = sc;
}

}  // namespace qh
