// qh_stage.hpp -- one overlap-save FIR stage over `nch` streams with its state, shared by the FIR bank
// (qh_fir.hip) and the Quisk receiver bank (qh_qrx.hip).
//
//   decimating   y[m] = sum_k h[k] x'[decim*m + (decim-1-phase) - k]      (x' = x * NCO when mix)
//   interpolating y[n] = sum_m h[m] u[n-m], u[interp*i] = x[i]
// Taps: one set for all channels or one set per channel; optional per-channel 2x2 real output matrix.
//
// A mixing fp64 stage that decimates by an even factor with REAL taps keeps its oscillator BEHIND the filter (`outmix`, the
// RXA front stage's form, qh_osfir.hpp OUTMIX): with theta(n) = phi + delta n,
//      sum_k h[k] x[g - k] exp(j theta(g - k)) = exp(j theta(g)) sum_k (h[k] exp(-j delta k)) x[g - k],
// so every channel gets its own mask (the taps modulated down by its delta, front_mask_kernel), the tile kernel rotates the
// 4096 / fold folded outputs instead of the 4096 inputs, and the history rows hold raw samples -- rewritten for the new phase
// law when a channel is retuned (nco_retune_hist_kernel), because the reference mixed them sample by sample with the old one.
#pragma once
#include <cmath>
#include <vector>
#include "qh_design.hpp"
#include "qh_internal.hpp"
#include "qh_kernels.hpp"

namespace qh {

static constexpr int kStageNfft = 4096;

struct Stage {
    int device = 0, nch = 0, ntaps = 0, decim = 1, interp = 1, dtype = QH_F64;
    bool mix = false, per_channel = false;
    bool outmix = false;                        // oscillator behind the filter (see above); decided by the first set_taps
    bool taps_set = false;
    std::vector<int> taps_len;                  // taps actually set, per channel (per_channel) or [0]: the tile plan follows their maximum
    double *taps_re = nullptr;                  // outmix: the real taps on the device, for front_mask_kernel
    double2 *lane_rot = nullptr, *out_step = nullptr, *tile_rot = nullptr;
    int tile_cap = 0;
    int *d_list = nullptr;
    unsigned long long *d_law = nullptr, *d_park = nullptr;
    int fold = 1, pick = 1, P = 0, Lf = 0;      // decimating: Lf folded outputs per tile; interpolating: Lf high-rate outputs per tile
    int hist_len = 0;                           // history rows (input-rate samples)
    int phase = 0;                              // decim_index
    hipStream_t stream = nullptr;
    void *mask = nullptr, *tw_fwd = nullptr, *tw_inv = nullptr, *hist[2] = { nullptr, nullptr };
    unsigned long long *nco_phase = nullptr, *nco_dphase = nullptr;
    double2 *nco_step = nullptr;
    EpiParam *epi = nullptr;
    int cur = 0;
    size_t esize = 16;
    // Real signals through real taps shared by all channels (Quisk's audio stages): channels 2p and 2p + 1 ride in the real and the
    // imaginary part of one tile (osfir_kernel PAIR).  pair: 0 off, 1 the stage's rows hold (y, 0), 2 they hold (y, y).
    int pair = 0, npairs = 0;
    int *d_pairs = nullptr;

    void destroy()
    {
        (void)hipFree(mask); (void)hipFree(tw_fwd); (void)hipFree(tw_inv); (void)hipFree(hist[0]); (void)hipFree(hist[1]);
        (void)hipFree(nco_phase); (void)hipFree(nco_dphase); (void)hipFree(nco_step); (void)hipFree(epi);
        (void)hipFree(taps_re); (void)hipFree(lane_rot); (void)hipFree(out_step); (void)hipFree(tile_rot); (void)hipFree(d_list); (void)hipFree(d_law); (void)hipFree(d_park);
        (void)hipFree(d_pairs); d_pairs = nullptr;
        mask = tw_fwd = tw_inv = hist[0] = hist[1] = nullptr;
        nco_phase = nco_dphase = nullptr; nco_step = nullptr; epi = nullptr;
        taps_re = nullptr; lane_rot = out_step = tile_rot = nullptr; d_list = nullptr; d_law = nullptr; d_park = nullptr; tile_cap = 0;
    }

    int upload_cplx(void *dst, const std::vector<cd> &v)
    {
        if (dtype == QH_F64) {
            QH_HIP(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(cd), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        } else {
            std::vector<float> f(v.size() * 2);
            for (size_t i = 0; i < v.size(); i++) { f[2 * i] = (float)v[i].real(); f[2 * i + 1] = (float)v[i].imag(); }
            QH_HIP(hipMemcpyAsync(dst, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        return QH_OK;
    }

    // geometry + buffers; taps are set afterwards with set_taps
    int init(int device_, int nch_, int ntaps_, int decim_, int interp_, int dtype_, bool mix_, bool per_channel_, bool with_epi,
             hipStream_t s)
    {
        device = device_; nch = nch_; ntaps = ntaps_; decim = decim_; interp = interp_; dtype = dtype_; mix = mix_;
        per_channel = per_channel_; stream = s;
        esize = dtype == QH_F64 ? 16 : 8;
        if (interp > 1) {
            if (decim != 1 || (interp != 2 && interp != 4 && interp != 8))
                return set_error(QH_ERR_UNSUPPORTED, "interpolation must be 2, 4 or 8 without decimation");
            P = ((ntaps - 1 + interp - 1) / interp) * interp;
            if (P < interp) P = interp;
            Lf = ((kStageNfft - P) / interp) * interp;
            hist_len = P / interp;
            fold = 1; pick = 1;
        } else {
            fold = (decim % 8 == 0) ? 8 : (decim % 4 == 0) ? 4 : (decim % 2 == 0) ? 2 : 1;
            pick = decim / fold;
            P = ((ntaps - 1 + fold - 1) / fold) * fold;
            if (P < fold) P = fold;
            Lf = (((kStageNfft - P) / fold) / pick) * pick;
            hist_len = P;
        }
        if (Lf <= 0) return set_error(QH_ERR_UNSUPPORTED, "%d taps with decimation %d / interpolation %d do not fit a %d-point tile",
                                      ntaps, decim, interp, kStageNfft);
        QH_HIP(hipSetDevice(device));
        outmix = mix && dtype == QH_F64 && interp == 1 && fold > 1;      // taken back by set_taps if the taps are not real
        taps_set = false;
        taps_len.assign((size_t)(per_channel ? nch : 1), 0);
        const size_t nmask = (size_t)(per_channel || outmix ? nch : 1) * kStageNfft;
        QH_HIP(hipMalloc(&mask, nmask * esize));
        QH_HIP(hipMemsetAsync(mask, 0, nmask * esize, stream));
        const int nfwd = interp > 1 ? kStageNfft / interp : kStageNfft;
        const int ninv = interp > 1 ? kStageNfft : kStageNfft / fold;
        std::vector<cd> t1 = fft_twiddle_table(nfwd), t2 = fft_twiddle_table(ninv);
        QH_HIP(hipMalloc(&tw_fwd, t1.size() * esize));
        QH_HIP(hipMalloc(&tw_inv, t2.size() * esize));
        if (int rc = upload_cplx(tw_fwd, t1)) return rc;
        if (int rc = upload_cplx(tw_inv, t2)) return rc;
        for (int i = 0; i < 2; i++) {
            QH_HIP(hipMalloc(&hist[i], (size_t)nch * hist_len * esize));
            QH_HIP(hipMemsetAsync(hist[i], 0, (size_t)nch * hist_len * esize, stream));
        }
        if (mix) {
            QH_HIP(dev_alloc(&nco_phase, (size_t)nch));
            QH_HIP(dev_alloc(&nco_dphase, (size_t)nch));
            QH_HIP(dev_alloc(&nco_step, (size_t)nch));
            QH_HIP(hipMemsetAsync(nco_phase, 0, (size_t)nch * 8, stream));
            QH_HIP(hipMemsetAsync(nco_dphase, 0, (size_t)nch * 8, stream));
            std::vector<double2> one((size_t)nch, make_double2(1.0, 0.0));
            QH_HIP(hipMemcpyAsync(nco_step, one.data(), (size_t)nch * 16, hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        if (outmix) {
            QH_HIP(dev_alloc(&taps_re, (size_t)ntaps));
            QH_HIP(dev_alloc(&lane_rot, (size_t)nch * NT));
            QH_HIP(dev_alloc(&out_step, (size_t)nch));
            QH_HIP(dev_alloc(&d_list, (size_t)nch));
            QH_HIP(dev_alloc(&d_law, (size_t)2 * (size_t)nch));        // (action, new step) per channel: one retune of the whole bank is one launch
            std::vector<int> all((size_t)nch);
            for (int c = 0; c < nch; c++) all[(size_t)c] = c;
            QH_HIP(hipMemcpyAsync(d_list, all.data(), (size_t)nch * sizeof(int), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        if (with_epi) {
            QH_HIP(dev_alloc(&epi, (size_t)nch));
            std::vector<EpiParam> id((size_t)nch, EpiParam{ 1, 0, 0, 1 });
            QH_HIP(hipMemcpyAsync(epi, id.data(), (size_t)nch * sizeof(EpiParam), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        if (int rc = set_attr()) return rc;
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }

    // ch = -1 (shared taps) or a channel index (per-channel stages).  `taps` are the convolution taps h[k].
    int set_taps(int ch, const std::vector<cd> &taps)
    {
        if ((int)taps.size() > ntaps) return set_error(QH_ERR_INVALID, "more taps than the stage was created for");
        QH_HIP(hipSetDevice(device));
        if (outmix) {                           // decided before anything is touched: a refused call leaves the stage as it was
            bool real = true;
            for (const cd &v : taps) if (v.imag() != 0.0) { real = false; break; }
            if (!real) {
                if (taps_set) return set_error(QH_ERR_UNSUPPORTED, "a mixing stage that ran with real taps cannot take complex ones");
                outmix = false;                 // complex taps: the oscillator stays at the input (MIX); nco_step is kept current by set_nco
            }
        }
        if (interp == 1) {
            // The stage was created for up to `ntaps` taps (the Rx filter: 2048) and keeps that much history, but a tile only has to
            // overlap its neighbour by the longest filter actually set: 153 taps leave 3841 useful outputs of 4096, not 2049.
            if (per_channel && ch >= 0) taps_len[(size_t)ch] = (int)taps.size();
            else for (int &v : taps_len) v = (int)taps.size();
            int need = 1;
            for (int v : taps_len) need = v > need ? v : need;
            P = ((need - 1 + fold - 1) / fold) * fold;
            if (P < fold) P = fold;
            Lf = (((kStageNfft - P) / fold) / pick) * pick;
        }
        taps_set = true;
        if (outmix) {
            std::vector<double> re((size_t)ntaps, 0.0);
            for (size_t i = 0; i < taps.size(); i++) re[i] = taps[i].real();
            QH_HIP(hipMemcpyAsync(taps_re, re.data(), (size_t)ntaps * 8, hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            return build_outmix_tables(0, nch);
        }
        std::vector<cd> m = make_mask(taps, kStageNfft);
        if (poly()) {
            // the decimating fp64 kernel runs the polyphase form of its transform (FftSplit4096::run_poly): the fold then reads
            // G_a[k] = W_N^(a k) sum_q W_D^(a q) M[k + (N / D) q] at index (N / D) a + k
            const int N = kStageNfft, D = fold, S = N / D;
            const long double pi = 3.14159265358979323846264338327950288L;
            std::vector<cd> g((size_t)N);
            for (int a = 0; a < D; a++)
                for (int k = 0; k < S; k++) {
                    std::complex<long double> acc(0, 0);
                    for (int q = 0; q < D; q++) {
                        const long double ang = -2.0L * pi * (long double)((a * q) % D) / (long double)D;
                        acc += std::complex<long double>(m[(size_t)(k + S * q)]) * std::complex<long double>(cosl(ang), sinl(ang));
                    }
                    const long double ang = -2.0L * pi * (long double)(((long long)a * k) % N) / (long double)N;
                    acc *= std::complex<long double>(cosl(ang), sinl(ang));
                    g[(size_t)(S * a + k)] = cd((double)acc.real(), (double)acc.imag());
                }
            m.swap(g);
        }
        char *dst = static_cast<char *>(mask);
        if (per_channel) {
            if (ch < 0) {
                for (int c = 0; c < nch; c++)
                    if (int rc = upload_cplx(dst + (size_t)c * kStageNfft * esize, m)) return rc;
                return QH_OK;
            }
            return upload_cplx(dst + (size_t)ch * kStageNfft * esize, m);
        }
        return upload_cplx(dst, m);
    }

    // outmix: masks (modulated taps, polyphase form), lane phasors and per-register steps of channels [c0, c0 + n)
    int build_outmix_tables(int c0, int n)
    {
        hipLaunchKernelGGL((front_mask_kernel<kStageNfft>), dim3((unsigned)n), dim3(NT), (size_t)(TileFft<kStageNfft, false, double2>::kLdsBytes),
                           stream, (const double *)taps_re, ntaps, (const unsigned long long *)nco_dphase, (const int *)(d_list + c0), fold,
                           static_cast<const double2 *>(tw_fwd), static_cast<double2 *>(mask), lane_rot, out_step, 1);
        QH_HIP(hipGetLastError());
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }

    // NCO of channel ch: frequency ratio f/rate in turns per input sample (negative = tune down)
    int set_nco(int ch, double freq, double rate)
    {
        if (!mix) return set_error(QH_ERR_INVALID, "stage has no NCO");
        QH_HIP(hipSetDevice(device));
        long double t = (long double)freq / (long double)rate;
        t -= floorl(t);
        long double sc = t * 18446744073709551616.0L;
        unsigned long long d = sc >= 18446744073709551616.0L ? 0ull : (unsigned long long)sc;
        {       // the input-side oscillator's per-register step: kept current in both forms (outmix may still be withdrawn by set_taps)
            long double ang = 2.0L * 3.14159265358979323846264338327950288L * ((long double)(d * (unsigned long long)NT) / 18446744073709551616.0L);
            double2 st = make_double2((double)cosl(ang), (double)sinl(ang));
            QH_HIP(hipMemcpyAsync(nco_step + ch, &st, 16, hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        if (outmix) {
            // the raw history was going to be seen through the old phase law: re-express it for the new one (the kernel reads the
            // old law from the device arrays, so it runs ahead of their update), then rebuild the channel's tables
            const unsigned long long law[2] = { 0ull, d };
            QH_HIP(hipMemcpyAsync(d_law, law, sizeof(law), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(nco_retune_hist_kernel, dim3((unsigned)((hist_len + NT - 1) / NT), 1u), dim3(NT), 0, stream,
                               static_cast<double2 *>(hist[cur]), hist_len, (const unsigned long long *)nco_phase,
                               (const unsigned long long *)nco_dphase, (const unsigned long long *)nullptr, (const int *)(d_list + ch),
                               (const unsigned long long *)d_law);
            QH_HIP(hipMemcpyAsync(nco_dphase + ch, &d, 8, hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            return taps_set ? build_outmix_tables(ch, 1) : QH_OK;
        }
        QH_HIP(hipMemcpyAsync(nco_dphase + ch, &d, 8, hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }

    // Every channel's NCO in one go: freq[nch].  The same arithmetic as set_nco channel by channel (bit-identical tables), but one
    // upload per array, ONE nco_retune_hist_kernel grid and ONE front_mask_kernel grid over the bank instead of a launch and two
    // stream synchronisations per receiver (a 256-receiver bank: 2 x 256 launches and ~95 ms, profiles/r04_j_quisk_kernel_stats.csv).
    int set_nco_all(const double *freq, double rate)
    {
        if (!mix) return set_error(QH_ERR_INVALID, "stage has no NCO");
        QH_HIP(hipSetDevice(device));
        std::vector<unsigned long long> d((size_t)nch), law((size_t)2 * (size_t)nch);
        std::vector<double2> st((size_t)nch);
        for (int c = 0; c < nch; c++) {
            long double t = (long double)freq[c] / (long double)rate;
            t -= floorl(t);
            long double sc = t * 18446744073709551616.0L;
            d[(size_t)c] = sc >= 18446744073709551616.0L ? 0ull : (unsigned long long)sc;
            long double ang = 2.0L * 3.14159265358979323846264338327950288L * ((long double)(d[(size_t)c] * (unsigned long long)NT) / 18446744073709551616.0L);
            st[(size_t)c] = make_double2((double)cosl(ang), (double)sinl(ang));
            law[(size_t)2 * c] = 0ull; law[(size_t)2 * c + 1] = d[(size_t)c];
        }
        QH_HIP(hipMemcpyAsync(nco_step, st.data(), (size_t)nch * 16, hipMemcpyHostToDevice, stream));
        if (outmix) {
            QH_HIP(hipMemcpyAsync(d_law, law.data(), law.size() * 8, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(nco_retune_hist_kernel, dim3((unsigned)((hist_len + NT - 1) / NT), (unsigned)nch), dim3(NT), 0, stream,
                               static_cast<double2 *>(hist[cur]), hist_len, (const unsigned long long *)nco_phase,
                               (const unsigned long long *)nco_dphase, (const unsigned long long *)nullptr, (const int *)d_list,
                               (const unsigned long long *)d_law);
        }
        QH_HIP(hipMemcpyAsync(nco_dphase, d.data(), (size_t)nch * 8, hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));               // (the host vectors above are pageable)
        if (outmix && taps_set) return build_outmix_tables(0, nch);
        return QH_OK;
    }

    // The oscillator's phase (2^-64 turns) at the next input sample of channel ch.  quisk_process_samples keeps one tune vector
    // per PURPOSE (rxTuneVector, txTuneVector, aux1TuneVector, aux2TuneVector; quisk.c:2308-2311) while the banks' filter storage is
    // per bank: a bank that changes purpose takes the other vector's phase and keeps its filter history.
    int get_nco_phase(int ch, unsigned long long *p)
    {
        if (!mix) return set_error(QH_ERR_INVALID, "stage has no NCO");
        QH_HIP(hipSetDevice(device));
        QH_HIP(hipMemcpyAsync(p, nco_phase + ch, 8, hipMemcpyDeviceToHost, stream));
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }
    int set_nco_phase(int ch, unsigned long long p)
    {
        if (!mix) return set_error(QH_ERR_INVALID, "stage has no NCO");
        QH_HIP(hipSetDevice(device));
        if (outmix) {
            // the raw history was going to be seen through the old phase; re-express it for the new one (same frequency)
            if (!d_park) QH_HIP(dev_alloc(&d_park, (size_t)nch));
            unsigned long long d = 0;
            QH_HIP(hipMemcpyAsync(&d, nco_dphase + ch, 8, hipMemcpyDeviceToHost, stream));
            QH_HIP(hipStreamSynchronize(stream));
            const unsigned long long law[2] = { 2ull, d };
            QH_HIP(hipMemcpyAsync(d_law, law, sizeof(law), hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemcpyAsync(d_park + ch, &p, 8, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(nco_retune_hist_kernel, dim3((unsigned)((hist_len + NT - 1) / NT), 1u), dim3(NT), 0, stream,
                               static_cast<double2 *>(hist[cur]), hist_len, (const unsigned long long *)nco_phase,
                               (const unsigned long long *)nco_dphase, (const unsigned long long *)d_park, (const int *)(d_list + ch),
                               (const unsigned long long *)d_law);
            QH_HIP(hipGetLastError());
        }
        QH_HIP(hipMemcpyAsync(nco_phase + ch, &p, 8, hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }

    int set_epi(int ch, EpiParam e)
    {
        if (!epi) return set_error(QH_ERR_INVALID, "stage has no output matrix");
        QH_HIP(hipSetDevice(device));
        QH_HIP(hipMemcpyAsync(epi + ch, &e, sizeof(e), hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        return QH_OK;
    }

    int reset()
    {
        QH_HIP(hipSetDevice(device));
        for (int i = 0; i < 2; i++) QH_HIP(hipMemsetAsync(hist[i], 0, (size_t)nch * hist_len * esize, stream));
        if (mix) QH_HIP(hipMemsetAsync(nco_phase, 0, (size_t)nch * 8, stream));
        phase = 0;
        return QH_OK;
    }

    int out_count(int n_in) const { return interp > 1 ? n_in * interp : (phase + n_in) / decim; }

    // fp64 decimating stages run the polyphase form of the forward transform (masks turned into G in set_taps)
    bool poly() const { return dtype == QH_F64 && interp == 1 && fold > 1; }
    template <typename T, int FOLD> static constexpr bool kPoly = sizeof(T) == 8 && FOLD > 1;
    template <typename T, int FOLD, bool MIX> int attr_one()
    {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<T, kStageNfft, FOLD, MIX, false, false, false, false, kPoly<T, FOLD>>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_lds_bytes<T, kStageNfft, FOLD>())));
        return QH_OK;
    }
    template <typename T, int U> int attr_up()
    {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_interp_kernel<T, kStageNfft, U>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_interp_lds_bytes<T, kStageNfft, U>())));
        return QH_OK;
    }
    template <typename T> int set_attr_t()
    {
        if (interp > 1) return interp == 2 ? attr_up<T, 2>() : interp == 4 ? attr_up<T, 4>() : attr_up<T, 8>();
        if (mix) return fold == 1 ? attr_one<T, 1, true>() : fold == 2 ? attr_one<T, 2, true>() : fold == 4 ? attr_one<T, 4, true>() : attr_one<T, 8, true>();
        return fold == 1 ? attr_one<T, 1, false>() : fold == 2 ? attr_one<T, 2, false>() : fold == 4 ? attr_one<T, 4, false>() : attr_one<T, 8, false>();
    }
    template <int FOLD> int attr_outmix()
    {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<double, kStageNfft, FOLD, false, false, false, true, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_lds_bytes<double, kStageNfft, FOLD>())));
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&front_mask_kernel<kStageNfft>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   TileFft<kStageNfft, false, double2>::kLdsBytes));
        return QH_OK;
    }
    int set_attr()
    {
        if (outmix) if (int rc = fold == 2 ? attr_outmix<2>() : fold == 4 ? attr_outmix<4>() : attr_outmix<8>()) return rc;
        return dtype == QH_F64 ? set_attr_t<double>() : set_attr_t<float>();
    }
    template <int FOLD> void launch_outmix(const OsfirArgs<double> &a)
    {
        dim3 grid((unsigned)a.ntiles * (unsigned)nch), block(NT);
        constexpr int lds = osfir_lds_bytes<double, kStageNfft, FOLD>();
        hipLaunchKernelGGL((osfir_kernel<double, kStageNfft, FOLD, false, false, false, true, false, true>), grid, block, lds, stream, a);
    }

    template <typename T, int FOLD, bool MIX> void launch_dec(const OsfirArgs<T> &a)
    {
        dim3 grid((unsigned)a.ntiles * (unsigned)nch), block(NT);      // 1-D: the kernel maps ids to (channel, tile)
        constexpr int lds = osfir_lds_bytes<T, kStageNfft, FOLD>();
        hipLaunchKernelGGL((osfir_kernel<T, kStageNfft, FOLD, MIX, false, false, false, false, kPoly<T, FOLD>>), grid, block, lds, stream, a);
    }
    template <typename T, int FOLD> void launch_dec_pair(const OsfirArgs<T> &a)
    {
        dim3 grid((unsigned)a.ntiles * (unsigned)npairs), block(NT);
        constexpr int lds = osfir_lds_bytes<T, kStageNfft, FOLD>();
        hipLaunchKernelGGL((osfir_kernel<T, kStageNfft, FOLD, false, false, false, false, false, kPoly<T, FOLD>, 0, true>), grid, block, lds, stream, a);
    }
    template <typename T, int U> void launch_up_pair(const OsfirArgs<T> &a)
    {
        dim3 grid((unsigned)a.ntiles * (unsigned)npairs), block(NT);
        constexpr int lds = osfir_interp_lds_bytes<T, kStageNfft, U>();
        hipLaunchKernelGGL((osfir_interp_kernel<T, kStageNfft, U, true>), grid, block, lds, stream, a);
    }
    template <typename T, int FOLD> int attr_pair_one()
    {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<T, kStageNfft, FOLD, false, false, false, false, false, kPoly<T, FOLD>, 0, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_lds_bytes<T, kStageNfft, FOLD>())));
        return QH_OK;
    }
    template <typename T, int U> int attr_pair_up()
    {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_interp_kernel<T, kStageNfft, U, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_interp_lds_bytes<T, kStageNfft, U>())));
        return QH_OK;
    }
    template <typename T> int set_attr_pair_t()
    {
        if (interp > 1) return interp == 2 ? attr_pair_up<T, 2>() : interp == 4 ? attr_pair_up<T, 4>() : attr_pair_up<T, 8>();
        return fold == 1 ? attr_pair_one<T, 1>() : fold == 2 ? attr_pair_one<T, 2>() : fold == 4 ? attr_pair_one<T, 4>() : attr_pair_one<T, 8>();
    }
    // mode 1: the rows hold (y, 0); 2: (y, y); 0: off.  Real taps shared by all channels only.
    int set_pair(int mode)
    {
        if (mode && (per_channel || mix || outmix || nch < 2)) { pair = 0; return QH_OK; }
        if (mode) {
            QH_HIP(hipSetDevice(device));
            if (int rc = dtype == QH_F64 ? set_attr_pair_t<double>() : set_attr_pair_t<float>()) return rc;
        }
        pair = mode;
        if (mode && !d_pairs) {
            npairs = (nch + 1) / 2;
            std::vector<int> l((size_t)npairs * 2);
            for (int p = 0; p < npairs; p++) { l[(size_t)2 * p] = 2 * p; l[(size_t)2 * p + 1] = 2 * p + 1 < nch ? 2 * p + 1 : 2 * p; }
            QH_HIP(hipSetDevice(device));
            QH_HIP(hipMalloc((void **)&d_pairs, l.size() * sizeof(int)));
            QH_HIP(hipMemcpyAsync(d_pairs, l.data(), l.size() * sizeof(int), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        return QH_OK;
    }
    template <typename T, int U> void launch_up(const OsfirArgs<T> &a)
    {
        dim3 grid((unsigned)a.ntiles * (unsigned)nch), block(NT);      // 1-D: the kernel maps ids to (channel, tile)
        constexpr int lds = osfir_interp_lds_bytes<T, kStageNfft, U>();
        hipLaunchKernelGGL((osfir_interp_kernel<T, kStageNfft, U>), grid, block, lds, stream, a);
    }

    template <typename T> int process_t(const void *in, long long in_stride, int n_in, void *out, long long out_stride, int *n_out)
    {
        QH_HIP(hipSetDevice(device));
        const int nout = out_count(n_in);
        if (n_out) *n_out = nout;
        if (n_in <= 0) return QH_OK;
        OsfirArgs<T> a{};
        a.in = static_cast<const cplx<T> *>(in); a.in_stride = in_stride;
        a.hist = static_cast<const cplx<T> *>(hist[cur]); a.hist_stride = hist_len; a.hist_len = hist_len;
        a.out = static_cast<cplx<T> *>(out); a.out_stride = out_stride; a.out_offset = 0;
        a.mask = static_cast<const cplx<T> *>(mask); a.mask_stride = per_channel || outmix ? kStageNfft : 0;
        a.tw_fwd = static_cast<const cplx<T> *>(tw_fwd); a.tw_inv = static_cast<const cplx<T> *>(tw_inv);
        a.nco_phase = nco_phase; a.nco_dphase = nco_dphase; a.nco_step = nco_step;
        a.epi = epi;
        a.n_in = n_in; a.n_out = nout; a.P = P; a.Lout = Lf;
        // the kernel's own tiles leave the next call's delay line (OsfirArgs::hist_next): decimators / plain filters with raw history, a call
        // that is at least one delay line long (the tiles' loads cover [off - P, off + ntiles * fold * Lf), which reaches n_in: off + decim * nout >= n_in)
        const bool paired_now = pair && !per_channel && !mix && !outmix && pick <= 1;
        const bool hist_in_kernel = nout > 0 && interp == 1 && !mix && !paired_now && n_in >= hist_len;
        if (hist_in_kernel) a.hist_next = static_cast<cplx<T> *>(hist[cur ^ 1]);
        if (nout > 0) {
            const bool paired = pair && !per_channel && !mix && !outmix && pick <= 1;
            if (paired) { a.chan_list = d_pairs; a.pair_im0 = pair == 1 ? 1 : 0; }
            if (interp > 1) {
                a.ntiles = (nout + Lf - 1) / Lf;
                if (paired) switch (interp) { case 2: launch_up_pair<T, 2>(a); break; case 4: launch_up_pair<T, 4>(a); break; default: launch_up_pair<T, 8>(a); }
                else switch (interp) { case 2: launch_up<T, 2>(a); break; case 4: launch_up<T, 4>(a); break; default: launch_up<T, 8>(a); }
            } else if (paired) {
                a.off = decim - 1 - phase; a.pick = pick;
                a.ntiles = (nout + Lf - 1) / Lf;
                switch (fold) { case 1: launch_dec_pair<T, 1>(a); break; case 2: launch_dec_pair<T, 2>(a); break;
                                case 4: launch_dec_pair<T, 4>(a); break; default: launch_dec_pair<T, 8>(a); }
            } else {
                a.off = decim - 1 - phase; a.pick = pick;
                const int per_tile = Lf / pick;
                a.ntiles = (nout + per_tile - 1) / per_tile;
                if constexpr (sizeof(T) == 8) if (outmix) {
                    if (a.ntiles > tile_cap) {
                        QH_HIP(hipStreamSynchronize(stream));
                        (void)hipFree(tile_rot); tile_rot = nullptr; tile_cap = 0;
                        QH_HIP(dev_alloc(&tile_rot, (size_t)nch * (size_t)a.ntiles));
                        tile_cap = a.ntiles;
                    }
                    // oscillator phase at the first input index of every tile: g0 = off - P + tile * fold * Lf
                    hipLaunchKernelGGL(nco_tile_kernel, dim3((unsigned)((a.ntiles + 255) / 256), (unsigned)nch), dim3(256), 0, stream,
                                       (const unsigned long long *)nco_phase, (const unsigned long long *)nco_dphase, tile_rot, a.ntiles,
                                       (long long)(a.off - a.P), (long long)fold * Lf);
                    a.tile_rot = tile_rot; a.lane_rot = lane_rot; a.nco_step = out_step;
                    switch (fold) { case 2: launch_outmix<2>(a); break; case 4: launch_outmix<4>(a); break; default: launch_outmix<8>(a); }
                }
                if (outmix) { }
                else if (mix) switch (fold) { case 1: launch_dec<T, 1, true>(a); break; case 2: launch_dec<T, 2, true>(a); break;
                                         case 4: launch_dec<T, 4, true>(a); break; default: launch_dec<T, 8, true>(a); }
                else switch (fold) { case 1: launch_dec<T, 1, false>(a); break; case 2: launch_dec<T, 2, false>(a); break;
                                     case 4: launch_dec<T, 4, false>(a); break; default: launch_dec<T, 8, false>(a); }
            }
        }
        dim3 g((unsigned)((hist_len + NT - 1) / NT), (unsigned)nch);
        if (outmix) {           // raw history; the phase advances all the same
            if (!hist_in_kernel)
            hipLaunchKernelGGL((hist_update_kernel<T, false>), g, dim3(NT), 0, stream, a.in, in_stride, n_in, a.hist,
                               static_cast<cplx<T> *>(hist[cur ^ 1]), hist_len, (const unsigned long long *)nullptr,
                               (const unsigned long long *)nullptr, (const int *)nullptr);
            hipLaunchKernelGGL(nco_advance_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, stream, nco_phase, nco_dphase,
                               nch, (long long)n_in);
        } else if (mix) {
            hipLaunchKernelGGL((hist_update_kernel<T, true>), g, dim3(NT), 0, stream, a.in, in_stride, n_in, a.hist,
                               static_cast<cplx<T> *>(hist[cur ^ 1]), hist_len, nco_phase, nco_dphase, (const int *)nullptr);
            hipLaunchKernelGGL(nco_advance_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, stream, nco_phase, nco_dphase,
                               nch, (long long)n_in);
        } else if (!hist_in_kernel) {
            hipLaunchKernelGGL((hist_update_kernel<T, false>), g, dim3(NT), 0, stream, a.in, in_stride, n_in, a.hist,
                               static_cast<cplx<T> *>(hist[cur ^ 1]), hist_len, (const unsigned long long *)nullptr,
                               (const unsigned long long *)nullptr, (const int *)nullptr);
        }
        cur ^= 1;
        if (interp == 1) phase = (phase + n_in) % decim;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }

    int process(const void *in, long long in_stride, int n_in, void *out, long long out_stride, int *n_out)
    {
        return dtype == QH_F64 ? process_t<double>(in, in_stride, n_in, out, out_stride, n_out)
                               : process_t<float>(in, in_stride, n_in, out, out_stride, n_out);
    }

    // state from the host: hist = ntaps-1 most recent samples per channel, oldest first
    int set_state(const void *h, int phase_)
    {
        if (phase_ < 0 || phase_ >= decim) return set_error(QH_ERR_INVALID, "phase out of range");
        QH_HIP(hipSetDevice(device));
        QH_HIP(hipMemsetAsync(hist[cur], 0, (size_t)nch * hist_len * esize, stream));
        const int nh = ntaps - 1;
        if (h && nh > 0)
            QH_HIP(hipMemcpy2DAsync(static_cast<char *>(hist[cur]) + (size_t)(hist_len - nh) * esize, (size_t)hist_len * esize, h,
                                    (size_t)nh * esize, (size_t)nh * esize, (size_t)nch, hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        phase = phase_;
        return QH_OK;
    }
};

}  // namespace qh
