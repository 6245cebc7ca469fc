// qh_demod.hpp -- WDSP demodulators for gfx950: AM envelope + fade leveler, SAM / FM phase-locked loops,
// CTCSS notch.  One 64-lane wavefront per receiver channel, samples taken 64 at a time (coalesced).
//
//   am_detect_kernel   xamd mode 0 (wdsp/amd.c:131-146): |z| and two one-pole averages.  The averages are
//                      linear recurrences, solved inside the wavefront by a Kogge-Stone scan with constant
//                      coefficients (v_i += m^d * v_{i-d}), the carry passing from one 64-sample group to the next.
//   fm_pll_kernel,     the discriminator of xfmd (fmd.c:151-172) and xamd mode 1 (amd.c:148-232).  A PLL is a
//   sam_pll_kernel     non-linear recurrence: sequential per channel (SURVEY.md hard part 3).  Its phase detector
//                      atan2(corr1, corr0) equals arg(z) - phs, so arg(z) is taken by all lanes at once and only
//                      the loop filter is stepped sample by sample (pll_run64); every lane steps the same state,
//                      lane i keeps the i-th result, loads and stores stay coalesced.  Parallelism = channels.
//   snotch_kernel      xsnotch (wdsp/iir.c:76-95): bi-quad on the I component only; 2x2 constant-matrix scan.
#pragma once
#include "qh_fft.hpp"
#include "qh_wave.hpp"

namespace qh {

struct AmParam {            // per engine (depends on the DSP rate only), init_amd wdsp/amd.c:86-89
    double mtauR, onem_mtauR, mtauI, onem_mtauI;
};

struct AmState { double dc, dc_insert; };

// ------------------------------------------------------------------------------------------------ PLLs
struct PllParam {           // calc_fmd wdsp/fmd.c:29-44 ; init_amd wdsp/amd.c:72-89
    double omega_min, omega_max, g1, g2;
    double mtau, onem_mtau, again;                  // FM: dc removal and audio gain
    double mtauR, onem_mtauR, mtauI, onem_mtauI;    // SAM: fade leveler
};

struct PllState {
    double phs, fil_out, omega;
    double fmdc;                                    // FM
    double dc, dc_insert;                           // SAM
    double dsI, dsQ;                                // SAM sideband separation
    double a[24], b[24], c[24], d[24];              // SAM all-pass chains (3*STAGES + 3, wdsp/amd.h:64-67)
};

struct SamChanParam { int sbmode, levelfade; };

// The second-order loop of both PLL detectors (fmd.c:151-172, amd.c:222-232), 64 samples per call.
// The reference forms det = atan2(c1, c0) of the sample rotated back by the VCO phase; that is arg(z) - phs wrapped
// to (-pi, pi], and arg(z) does not depend on the loop: the lanes take the atan2 of their samples at once
// (theta_t, in turns) and the sequential part carries the loop filter alone -- 9 dependent VALU operations per
// sample with the phase kept in turns (wrap = x - rint(x), v_fract for the VCO) where the literal form needs a
// sincos and an atan2.  A PLL in lock is a contraction, so the few-ulp difference in det does not accumulate
// (tests compare after lock, like for every FFT-noise-driven acquisition).  Lane i receives the VCO phase that
// sample i saw (turns) and the loop-filter output after sample i.
struct PllLoop { double pt, fil_out, omega; };
// `out` is 128 doubles of LDS: the loop's uniform state after / before each sample is written there (the same value from every
// lane) and picked up per lane afterwards -- two LDS stores per sample instead of four predicated register moves.
template <bool ZEROS>
__device__ __forceinline__ void pll_steps(PllLoop &s, double theta_t, unsigned long long zero, int cnt, const PllParam &q, double *out)
{
    const double g1t = q.g1 * kTwoPiRef, g2t = q.g2 * kTwoPiRef, inv = 1.0 / kTwoPiRef;
    const double lo = q.omega_min, hi = q.omega_max;
    for (int i = 0; i < cnt; i++) {
        out[i] = s.pt;
        double d = lane_bcast(theta_t, i) - s.pt;           // (-1.5, 0.5] turns
        d -= rint(d);
        if constexpr (ZEROS) if ((zero >> i) & 1ull) d = 0.0;   // "if both are zero, corr[0] = 1.0": det = 0
        const double del_out = s.fil_out;
        s.omega = fmin(fmax(__builtin_fma(g2t, d, s.omega), lo), hi);
        s.fil_out = __builtin_fma(g1t, d, s.omega);
        s.pt = __builtin_amdgcn_fract(__builtin_fma(del_out, inv, s.pt));
        out[64 + i] = s.fil_out;
    }
}
__device__ __forceinline__ void pll_run64(PllLoop &s, double theta_t, unsigned long long zero, int cnt, const PllParam &q,
                                          int lane, double &my_pt, double &my_fil, double *out)
{
    if (zero == 0ull) pll_steps<false>(s, theta_t, zero, cnt, q, out);      // wave-uniform: no all-zero sample in the batch
    else pll_steps<true>(s, theta_t, zero, cnt, q, out);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    my_pt = lane < cnt ? out[lane] : 0.0;
    my_fil = lane < cnt ? out[64 + lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
}

// all-pass coefficients of the SAM sideband separator, wdsp/amd.c:91-106
static __constant__ double kSamC0[7] = { -0.328201924180698, -0.744171491539427, -0.923022915444215, -0.978490468768238,
                                  -0.994128272402075, -0.998458978159551, -0.999790306259206 };
static __constant__ double kSamC1[7] = { -0.0991227952747244, -0.565619728761389, -0.857467122550052, -0.959123933111275,
                                  -0.988739372718090, -0.996959189310611, -0.999282492800792 };

// ---- the sideband separator's all-pass chains (amd.c:160-186), 64 samples at a time ------------------------------------------
// A chain is seven sections y[n] = c (x[n] - y[n-2]) + x[n-2], the output of one the input of the next.  Per section and batch:
// u[n] = c x[n] + x[n-2] (x[n-2] by a shift of two lanes, the first two from the carried state), then y[n] = -c y[n-2] + u[n] as a
// lag-two Kogge-Stone scan over the 64 lanes (shifts of 2, 4, 8, 16, 32 with multipliers (-c)^1, ^2, ^4, ^8, ^16), then the carried
// y[-2] (even lanes) / y[-1] (odd lanes) times (-c)^(lane / 2 + 1).  The state is the reference's: x_j[n-1] = arr[3 j + 1],
// x_j[n-2] = arr[3 j + 2] for j = 0 .. 7 (x_7 = the chain's output), the other words of the arrays are scratch there.
__device__ __forceinline__ double shfl_up0(double v, int d, int lane)
{
    const double t = __shfl_up(v, d, 64);
    return lane >= d ? t : 0.0;
}
struct ApChain { double x1[8], x2[8]; };            // x_j[n-1], x_j[n-2]
__device__ __forceinline__ void ap_load(ApChain &s, const double *arr)
{
#pragma unroll
    for (int j = 0; j < 8; j++) { s.x1[j] = arr[3 * j + 1]; s.x2[j] = arr[3 * j + 2]; }
}
__device__ __forceinline__ void ap_store(const ApChain &s, double *arr)
{
#pragma unroll
    for (int j = 0; j < 8; j++) { arr[3 * j + 1] = s.x1[j]; arr[3 * j + 2] = s.x2[j]; }
}
// x: the chain's input of sample lane (lanes >= cnt: 0); pw[j] = (-c_j)^(lane / 2 + 1); returns the chain's output per lane
__device__ __forceinline__ double ap_chain64(ApChain &s, double x, const double *c, const double (&pw)[7], int cnt, int lane)
{
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double xs = shfl_up0(x, 2, lane);
        if (lane == 0) xs = s.x2[j];
        if (lane == 1) xs = s.x1[j];
        const double cj = c[j], p1 = -cj, p2 = p1 * p1, p4 = p2 * p2, p8 = p4 * p4, p16 = p8 * p8;
        double y = __builtin_fma(cj, x, xs);
        y = __builtin_fma(p1, shfl_up0(y, 2, lane), y);
        y = __builtin_fma(p2, shfl_up0(y, 4, lane), y);
        y = __builtin_fma(p4, shfl_up0(y, 8, lane), y);
        y = __builtin_fma(p8, shfl_up0(y, 16, lane), y);
        y = __builtin_fma(p16, shfl_up0(y, 32, lane), y);
        y = __builtin_fma(pw[j], (lane & 1) ? s.x1[j + 1] : s.x2[j + 1], y);
        // this section's input history for the next batch (cnt >= 1)
        const double l1 = lane_bcast(x, cnt - 1), l2 = cnt >= 2 ? lane_bcast(x, cnt - 2) : s.x1[j];
        s.x1[j] = l1; s.x2[j] = l2;
        x = lane < cnt ? y : 0.0;
    }
    const double l1 = lane_bcast(x, cnt - 1), l2 = cnt >= 2 ? lane_bcast(x, cnt - 2) : s.x1[7];
    s.x1[7] = l1; s.x2[7] = l2;
    return x;
}

// The same for FULL batches with the samples dealt to the two halves of the wavefront -- lane k < 32 holds sample 2 k, lane 32 + k
// sample 2 k + 1 -- so that the recurrences' lag of two samples is a lag of one lane inside a half: Kogge-Stone by DPP row shifts
// and one row broadcast per half (scan_pole_dpp without its last step), no LDS crossbar at all (the shuffles of ap_chain64 are 336
// ds_bpermute per batch of the four chains and bound it).  pa[j] = (-c_j)^((lane & 15) + 1), pw[j] = (-c_j)^((lane & 31) + 1).
__device__ __forceinline__ double ap_chain64s(ApChain &s, double x, const double *c, const double (&pa)[7], const double (&pw)[7], int lane)
{
    const bool first = (lane & 31) == 0, odd = lane >= 32;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double xs = wave_shr1(x);                               // x[n-2]: the element before it in its half
        if (first) xs = odd ? s.x1[j] : s.x2[j];
        const double cj = c[j], p1 = -cj, p2 = p1 * p1, p4 = p2 * p2, p8 = p4 * p4;
        double y = __builtin_fma(cj, x, xs);
        y = __builtin_fma(p1, dpp_fetch_d<0x111, 0xf>(y), y);
        y = __builtin_fma(p2, dpp_fetch_d<0x112, 0xf>(y), y);
        y = __builtin_fma(p4, dpp_fetch_d<0x114, 0xf>(y), y);
        y = __builtin_fma(p8, dpp_fetch_d<0x118, 0xf>(y), y);
        y = __builtin_fma(pa[j], dpp_fetch_d<0x142, 0xa>(y), y);        // lane 15 of rows 0 and 2 into rows 1 and 3
        y = __builtin_fma(pw[j], odd ? s.x1[j + 1] : s.x2[j + 1], y);
        s.x1[j] = lane_bcast(x, 63); s.x2[j] = lane_bcast(x, 31);
        x = y;
    }
    s.x1[7] = lane_bcast(x, 63); s.x2[7] = lane_bcast(x, 31);
    return x;
}

// Synchronous AM: in place.
static __global__ __launch_bounds__(64) void sam_pll_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                     PllState *state, const SamChanParam *cprm, PllParam q, AmState *fade)
{
    const double *c0 = kSamC0, *c1 = kSamC1;
    __shared__ double pll_out[128];
    const int ch = chan_list[blockIdx.x];
    const int lane = threadIdx.x;
    double2 *p = buf + (long long)ch * stride;
    PllState *sp = state + ch;
    const int sbmode = cprm[ch].sbmode, levelfade = cprm[ch].levelfade;
    PllLoop L{ sp->phs * (1.0 / kTwoPiRef), sp->fil_out, sp->omega };
    // the fade leveller's two averages belong to the amd block, not to a mode (amd.h:58-59): AM and SAM share them
    double dc = fade[ch].dc, dc_insert = fade[ch].dc_insert;
    double dsI = sp->dsI, dsQ = sp->dsQ;
    ApChain ca, cb, cc, cd;
    ap_load(ca, sp->a); ap_load(cb, sp->b); ap_load(cc, sp->c); ap_load(cd, sp->d);
    double pw0[7], pw1[7], pa0[7], pa1[7], ps0[7], ps1[7];      // powers for the shuffle form and for the DPP form (ap_chain64s)
#pragma unroll
    for (int j = 0; j < 7; j++) {
        pw0[j] = ipow_d(-c0[j], lane / 2 + 1); pw1[j] = ipow_d(-c1[j], lane / 2 + 1);
        pa0[j] = ipow_d(-c0[j], (lane & 15) + 1); pa1[j] = ipow_d(-c1[j], (lane & 15) + 1);
        ps0[j] = ipow_d(-c0[j], (lane & 31) + 1); ps1[j] = ipow_d(-c1[j], (lane & 31) + 1);
    }
    // Same split as in fm_pll_kernel: det = atan2(corr1, corr0) (amd.c:222-223) is arg(z) - phs, so the sequential
    // part only carries the loop filter; each lane then forms the VCO products of ITS sample from the phase the
    // loop had at that sample.  The fade leveler (two one-pole averages, amd.c:211-216) is solved by scans; only
    // the sideband separator's all-pass chains (sbmode 1, 2) remain a per-sample loop.
    const double pwR = lane_pow(q.mtauR, lane + 1), pwI = lane_pow(q.mtauI, lane + 1);
    double2 znext = make_double2(0, 0);
    if (lane < n) znext = p[lane];
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        const double2 z = znext;
        znext = make_double2(0, 0);
        if (base + 64 + lane < n) znext = p[base + 64 + lane];
        const double theta_t = atan2(z.y, z.x) * (1.0 / kTwoPiRef);
        const unsigned long long zero = __ballot(z.x == 0.0 && z.y == 0.0);
        double my_pt, my_fil;
        pll_run64(L, theta_t, zero, cnt, q, lane, my_pt, my_fil, pll_out);
        const double myphs = my_pt * kTwoPiRef;
        double sn, cs;
        sincos(myphs, &sn, &cs);
        const double ai = z.x * cs, bi = z.x * sn, aq = z.y * cs, bq = z.y * sn;
        const double corr0 = ai + bq;
        double audio = corr0;
        if (sbmode != 0) {
            // chain inputs: a <- ai one sample late (dsI), b <- bi, c <- bq one sample late (dsQ), d <- aq   (amd.c:162-167)
            const bool live = lane < cnt;
            double ai_d = wave_shr1(ai), bq_d = wave_shr1(bq);
            if (lane == 0) { ai_d = dsI; bq_d = dsQ; }
            dsI = lane_bcast(ai, cnt - 1); dsQ = lane_bcast(bq, cnt - 1);
            if (cnt == 64) {
                // full batch: deal the samples to the halves of the wavefront (lane k: sample 2 k, lane 32 + k: sample 2 k + 1), run the
                // chains by DPP scans (ap_chain64s), bring the result back
                const int src = 2 * (lane & 31) + (lane >> 5), back = (lane >> 1) + 32 * (lane & 1);
                const double ai_ps = ap_chain64s(ca, __shfl(ai_d, src, 64), c0, pa0, ps0, lane), bi_ps = ap_chain64s(cb, __shfl(bi, src, 64), c1, pa1, ps1, lane);
                const double bq_ps = ap_chain64s(cc, __shfl(bq_d, src, 64), c0, pa0, ps0, lane), aq_ps = ap_chain64s(cd, __shfl(aq, src, 64), c1, pa1, ps1, lane);
                audio = __shfl(sbmode == 1 ? (ai_ps - bi_ps) + (aq_ps + bq_ps) : (ai_ps + bi_ps) - (aq_ps - bq_ps), back, 64);
            } else {
                const double ai_ps = ap_chain64(ca, live ? ai_d : 0.0, c0, pw0, cnt, lane), bi_ps = ap_chain64(cb, live ? bi : 0.0, c1, pw1, cnt, lane);
                const double bq_ps = ap_chain64(cc, live ? bq_d : 0.0, c0, pw0, cnt, lane), aq_ps = ap_chain64(cd, live ? aq : 0.0, c1, pw1, cnt, lane);
                audio = sbmode == 1 ? (ai_ps - bi_ps) + (aq_ps + bq_ps) : (ai_ps + bi_ps) - (aq_ps - bq_ps);
            }
        }
        if (levelfade) {
            // dc_i = mtauR dc_{i-1} + onem_mtauR audio_i ; dc_insert_i likewise on corr0 ; audio += dc_insert - dc
            const bool live = lane < cnt;
            const double dcs = scan_pole(live ? q.onem_mtauR * audio : 0.0, q.mtauR, lane) + pwR * dc;
            const double dis = scan_pole(live ? q.onem_mtauI * corr0 : 0.0, q.mtauI, lane) + pwI * dc_insert;
            audio += dis - dcs;
            dc = lane_bcast(dcs, cnt - 1);
            dc_insert = lane_bcast(dis, cnt - 1);
        }
        if (lane < cnt) p[base + lane] = make_double2(audio, audio);
    }
    __syncthreads();
    if (lane == 0) {
        sp->phs = L.pt * kTwoPiRef; sp->fil_out = L.fil_out; sp->omega = L.omega; fade[ch].dc = dc; fade[ch].dc_insert = dc_insert;
        sp->dsI = dsI; sp->dsQ = dsQ;
    }
    if (lane == 0 && sbmode != 0) { ap_store(ca, sp->a); ap_store(cb, sp->b); ap_store(cc, sp->c); ap_store(cd, sp->d); }
}

// ------------------------------------------------------------------------------------------------ CTCSS notch
struct SnotchParam { double a0, a1, a2, b1, b2; int run; int pad; };     // calc_snotch wdsp/iir.c:35-49
struct SnotchState { double x1, x2, y1, y2; };

struct M2 { double a, b, c, d; };   // [[a, b], [c, d]]
__device__ __forceinline__ M2 mmul(M2 x, M2 y)
{
    M2 r;
    r.a = x.a * y.a + x.b * y.c; r.b = x.a * y.b + x.b * y.d;
    r.c = x.c * y.a + x.d * y.c; r.d = x.c * y.b + x.d * y.d;
    return r;
}

// ------------------------------------------------------------------------------------------------ WDSP meters
// xmeter (wdsp/meter.c:75-108) over nblk DSP blocks: running average of |z|^2 (one pole per sample) and a peak that
// decays per sample and is topped up with the block maximum at the end of every block.  result[0] = average,
// result[1] = peak, both in dB through mlog10 (wdsp/meterlog10.c:547-554: log10(2) * (exponent + log2 of the
// mantissa truncated to 11 bits)).  One wave per channel.
struct MeterParam { double mult_average, mult_peak; };
struct MeterState { double avg, peak, res_av, res_pk; };

static __global__ __launch_bounds__(64) void meter_kernel(const double2 *buf, long long stride, int nblk, int size,
                                                          MeterState *state, MeterParam q, const int *chan_list,
                                                          const double *gain2 = nullptr)
{
    const int ch = chan_list ? chan_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    const double2 *p = buf + (long long)ch * stride;
    MeterState st = state[ch];
    // the buffer holds the signal ahead of a fixed AGC gain g that the output stage applies: |g z|^2 = g^2 |z|^2
    const double g2 = gain2 ? gain2[ch] : 1.0;
    const double pw = lane_pow(q.mult_average, lane + 1);
    const double pk_blk = pow(q.mult_peak, (double)size);
    for (int b = 0; b < nblk; b++) {
        double np = 0.0;
        for (int base = 0; base < size; base += 64) {
            const int cnt = size - base < 64 ? size - base : 64;
            double smag = 0.0;
            if (lane < cnt) { const double2 z = p[(long long)b * size + base + lane]; smag = (z.x * z.x + z.y * z.y) * g2; }
            const double a = scan_pole((1.0 - q.mult_average) * smag, q.mult_average, lane) + pw * st.avg;
            st.avg = lane_bcast(a, cnt - 1);
            np = fmax(np, wave_max_d(smag));
        }
        st.peak *= pk_blk;
        if (np > st.peak) st.peak = np;
        st.res_av = 10.0 * mlog10_dev(st.avg + 1.0e-40);
        st.res_pk = 10.0 * mlog10_dev(st.peak + 1.0e-40);
    }
    if (lane == 0) state[ch] = st;
}

// The same meters from the per-chunk partials the METER overlap-save kernel leaves behind (qh_osfir.hpp: x = sum over
// the chunk's 64 samples of (1 - m) m^(63 - i) |z_i|^2, y = max |z_i|^2): the one-pole average advances a chunk at a time,
// avg <- m^64 avg + x, and the peak a DSP block (cpb chunks) at a time, peak <- max(peak * mp^(64 cpb), block max).  Both
// recurrences are linear (the second over (max, *)), so their value at the end of the call is a weighted sum / maximum of
// independent pieces: every chunk's partial carries its own decay to the end of the call, and one reduction joins them
// with the carried state.  One workgroup per channel serves the three meters: adc meter (partials of the stage input), S meter and agc
// meter (both on the stage output, the agc meter behind the fixed gain g: gain2 = g^2).
// wdsp/RXA.c:566,569,589.  Storage: tile-major, inside a tile [wave][register] (cpt chunks per tile; layout 1: the
// two-group tile's [A: wave][8], [B: wave][16]).
static constexpr int kMeterFinishThreads = 1024;
// m^k for the meters' decay factors m = exp(-1 / (rate tau)): exp(k ln m) with ln m exact by construction
__device__ __forceinline__ double meter_decay(double ln_m, double k) { return exp(k * ln_m); }

static __global__ __launch_bounds__(kMeterFinishThreads) void meter_finish_kernel(const double2 *part_in, const double2 *part_out,
                                                                 long long stride, int nchunks, int cpb, int cpt, int layout, MeterState *m_adc,
                                                                 MeterState *m_s, MeterState *m_agc, double ln_avg, double ln_pk,
                                                                 const double *gain2)
{
    __shared__ double s_red[kMeterFinishThreads / 64][4];
    const int ch = blockIdx.x, T = threadIdx.x, lane = T & 63, wave = T >> 6;
    const double2 *pi = part_in + (long long)ch * stride, *po = part_out + (long long)ch * stride;
    const int nseg = cpt >> 2, nblocks = nchunks / cpb;
    // Both recurrences unrolled: chunk c of the call weighs m^(64 (nchunks - 1 - c)) in the average, and the peak of DSP
    // block b weighs mp^(size (nblocks - 1 - b)) in the peak.  Lanes read consecutive partials (storage order) and weigh
    // each by its own place in time.
    double avg_i = 0.0, peak_i = 0.0, avg_o = 0.0, peak_o = 0.0;        // adc meter (stage input), S meter (stage output)
    const int nslots = (nchunks + cpt - 1) / cpt * cpt;                 // the last tile may hold fewer chunks than slots
    for (int s0 = T; s0 < nslots; s0 += kMeterFinishThreads) {
        const int tile = s0 / cpt, in = s0 - tile * cpt;
        int c;
        if (layout == 2) {                                              // 6144-point tile on six wavefronts (osfir6k_kernel): chunk 6 r + w - 32
            const int w = in < 20 ? in / 10 : 2 + (in - 20) / 11, r = in < 20 ? 6 + in % 10 : 5 + (in - 20) % 11;
            c = tile * cpt + 6 * r + w - 32;
        } else if (layout == 0) c = tile * cpt + 4 * (in % nseg) + in / nseg;  // [wave][register] -> chunk 4 register + wave
        else if (in < 32) c = tile * cpt + 32 + 4 * (in & 7) + (in >> 3);      // two-group tile (osfir8k_kernel): group A, registers 8 .. 15
        else {                                                                  // group B: [wave][16 registers], the upper eight 4096 samples on
            const int w = (in - 32) >> 4, k = (in - 32) & 15;
            c = tile * cpt + (k < 8 ? 4 * k + w : 64 + 4 * (k - 8) + w);
        }
        if (c >= nchunks) continue;
        const double2 vi = pi[s0], vo = po[s0];
        const double wa = meter_decay(ln_avg, 64.0 * (double)(nchunks - 1 - c));
        const double wp = meter_decay(ln_pk, 64.0 * (double)cpb * (double)(nblocks - 1 - c / cpb));
        avg_i = __builtin_fma(vi.x, wa, avg_i); peak_i = fmax(peak_i, vi.y * wp);
        avg_o = __builtin_fma(vo.x, wa, avg_o); peak_o = fmax(peak_o, vo.y * wp);
    }
    avg_i = wave_sum_d(avg_i); avg_o = wave_sum_d(avg_o);
    peak_i = wave_max_d(peak_i); peak_o = wave_max_d(peak_o);
    if (lane == 0) { s_red[wave][0] = avg_i; s_red[wave][1] = peak_i; s_red[wave][2] = avg_o; s_red[wave][3] = peak_o; }
    __syncthreads();
    if (T < 3) {            // thread 0: adc meter, 1: S meter, 2: agc meter (the S meter's sums behind the fixed gain g: g^2)
        double a = 0.0, pk = 0.0;
        const int col = T == 0 ? 0 : 2;
        for (int w = 0; w < kMeterFinishThreads / 64; w++) { a += s_red[w][col]; pk = fmax(pk, s_red[w][col + 1]); }
        MeterState *state = T == 0 ? m_adc : T == 1 ? m_s : m_agc;
        const double g2 = (T == 2 && gain2) ? gain2[ch] : 1.0;
        MeterState st = state[ch];
        st.avg = st.avg * meter_decay(ln_avg, 64.0 * (double)nchunks) + g2 * a;
        st.peak = fmax(st.peak * meter_decay(ln_pk, 64.0 * (double)nchunks), g2 * pk);
        st.res_av = 10.0 * mlog10_dev(st.avg + 1.0e-40);
        st.res_pk = 10.0 * mlog10_dev(st.peak + 1.0e-40);
        state[ch] = st;
    }
}

// ------------------------------------------------------------------------------------------------ WDSP AGC
// xwcpagc modes 1-5 (wdsp/wcpAGC.c:177-338): look-ahead ring of attack_buffsize samples, running maximum over the
// ring (re-scanned, here by the 64 lanes in parallel, when the outgoing sample was the maximum), five-state
// attack / fast-decay / hang / decay machine on `volts`, log-slope gain.  A data-dependent recurrence: sequential
// per channel, one wave per channel, every lane runs the same scalar flow and lane i keeps output i.
struct AgcParam {                   // loadWcpAGC, wcpAGC.c:115-146
    double attack_mult, decay_mult, fast_decay_mult, fast_backmult, onemfast_backmult, out_target, min_volts, inv_out_target;
    double slope_constant, inv_max_input, hang_level, hang_backmult, onemhang_backmult, hang_decay_mult, pop_ratio;
    int attack_buffsize, hang_count_init, hang_enable, pmode;
};
static constexpr int kAgcRing = 2048;       // LDS ring entries (>= attack_buffsize + 2)
struct AgcState {
    double ring_max, volts, save_volts, fast_backaverage, hang_backaverage, gain;
    int out_index, hang_counter, decay_type, state, attack_buffsize, pad;
    double2 ring[kAgcRing];
    double abs_ring[kAgcRing];
};

// The reference's ring in full.  xwcpagc's ring has RB_SIZE = 30721 entries (wcpAGC.h:30-33, wcpAGC.c:38-50) of which the kernels here
// keep the 2048 behind out_index -- all that a constant attack window ever reads.  When SetRXAAGCAttack lengthens the window in
// mid-stream, in_index jumps ahead (loadWcpAGC, wcpAGC.c:119-120) and the entries it jumps over come out later as they are: samples
// written a lap (30721 samples) ago, or zeros.  So the full ring is kept beside the state, written by every call (only the call's
// last RB_SIZE inputs can survive it), and read when the window moves (agc_rewindow_kernel).
static constexpr int kAgcLongRing = 30721;
// grid (x, listed channels); lout[ch] = out_index in the full ring (-1 at start, calc_wcpagc)
static __global__ void agc_long_mirror_kernel(const double2 *buf, long long stride, int n, const int *chan_list, const AgcParam *prm, double2 *lring,
                                              double *labs, const int *lout)
{
    const int ch = chan_list[blockIdx.y];
    const AgcParam q = prm[ch];
    const long long o = lout[ch];
    const double2 *p = buf + (long long)ch * stride;
    const int first = n > kAgcLongRing ? n - kAgcLongRing : 0;
    for (int i = first + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int pos = (int)((o + 1 + q.attack_buffsize + i) % kAgcLongRing);     // sample i goes in at in_index = out_index + attack_buffsize
        const double2 z = p[i];
        lring[(long long)ch * kAgcLongRing + pos] = z;
        labs[(long long)ch * kAgcLongRing + pos] = q.pmode == 0 ? fmax(fabs(z.x), fabs(z.y)) : sqrt(__builtin_fma(z.x, z.x, z.y * z.y));
    }
}
static __global__ void agc_long_advance_kernel(int n, const int *chan_list, int count, int *lout)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int ch = chan_list[k];
    lout[ch] = (int)(((long long)lout[ch] + n) % kAgcLongRing);
}
// the state's 2048 entries behind out_index, taken again from the full ring (one workgroup per listed channel)
static __global__ __launch_bounds__(256) void agc_rewindow_kernel(const int *chan_list, AgcState *state, const double2 *lring, const double *labs,
                                                                   const int *lout);

static __global__ __launch_bounds__(64) void wcpagc_seq_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                               const AgcParam *prm, AgcState *state, double pre_gain = 1.0)
{
    __shared__ double2 ring[kAgcRing];
    __shared__ double abs_ring[kAgcRing];
    const int ch = chan_list[blockIdx.x];
    const int lane = threadIdx.x;
    const AgcParam q = prm[ch];
    AgcState *sp = state + ch;
    for (int i = lane; i < kAgcRing; i += 64) { ring[i] = sp->ring[i]; abs_ring[i] = sp->abs_ring[i]; }
    double ring_max = sp->ring_max, volts = sp->volts, save_volts = sp->save_volts, fba = sp->fast_backaverage,
           hba = sp->hang_backaverage, gain = sp->gain;
    int out_index = sp->out_index, hang_counter = sp->hang_counter, decay_type = sp->decay_type, st = sp->state;
    // loadWcpAGC re-derives in_index from out_index whenever a parameter changes (wcpAGC.c:120); keeping only
    // out_index and the current attack_buffsize is equivalent
    const int A = q.attack_buffsize;
    __syncthreads();
    double2 *p = buf + (long long)ch * stride;
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        double2 z = make_double2(0, 0);
        if (lane < cnt) z = p[base + lane];
        z.x *= pre_gain; z.y *= pre_gain;           // the FM limiter's lim_pre_gain (fmd.c:181-182); 1 for the AGC proper
        double2 mine = make_double2(0, 0);
        for (int i = 0; i < cnt; i++) {
            const double I = lane_bcast(z.x, i), Q = lane_bcast(z.y, i);
            out_index = (out_index + 1) & (kAgcRing - 1);
            const int in_index = (out_index + A) & (kAgcRing - 1);
            const double2 o = ring[out_index];
            const double abs_out = abs_ring[out_index];
            const double abs_in = q.pmode == 0 ? fmax(fabs(I), fabs(Q)) : sqrt(__builtin_fma(I, I, Q * Q));
            ring[in_index] = make_double2(I, Q);
            abs_ring[in_index] = abs_in;
            fba = __builtin_fma(q.fast_backmult, abs_out, q.onemfast_backmult * fba);
            hba = __builtin_fma(q.hang_backmult, abs_out, q.onemhang_backmult * hba);
            if (abs_out >= ring_max && abs_out > 0.0) {
                double m = 0.0;
                for (int j = lane; j < A; j += 64) m = fmax(m, abs_ring[(out_index + 1 + j) & (kAgcRing - 1)]);
                ring_max = wave_max_d(m);
            }
            if (abs_in > ring_max) ring_max = abs_in;
            if (hang_counter > 0) --hang_counter;
            const bool up = ring_max >= volts;
            switch (st) {
            case 0:
                if (up) volts = __builtin_fma(ring_max - volts, q.attack_mult, volts);
                else if (volts > q.pop_ratio * fba) { st = 1; volts = __builtin_fma(ring_max - volts, q.fast_decay_mult, volts); }
                else if (q.hang_enable && hba > q.hang_level) { st = 2; hang_counter = q.hang_count_init; decay_type = 1; }
                else { st = 3; volts = __builtin_fma(ring_max - volts, q.decay_mult, volts); decay_type = 0; }
                break;
            case 1:
                if (up) { st = 0; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else if (volts > save_volts) volts = __builtin_fma(ring_max - volts, q.fast_decay_mult, volts);
                else if (hang_counter > 0) st = 2;
                else if (decay_type == 0) { st = 3; volts = __builtin_fma(ring_max - volts, q.decay_mult, volts); }
                else { st = 4; volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts); }
                break;
            case 2:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else if (hang_counter == 0) { st = 4; volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts); }
                break;
            case 3:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else volts = __builtin_fma(ring_max - volts, q.decay_mult, volts);
                break;
            default:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts);
                break;
            }
            if (volts < q.min_volts) volts = q.min_volts;
            gain = volts * q.inv_out_target;
            const double mult = __builtin_fma(-q.slope_constant, fmin(0.0, log10(q.inv_max_input * volts)), q.out_target) / volts;
            if (lane == i) mine = make_double2(o.x * mult, o.y * mult);
        }
        if (lane < cnt) p[base + lane] = mine;
    }
    __syncthreads();
    for (int i = lane; i < kAgcRing; i += 64) { sp->ring[i] = ring[i]; sp->abs_ring[i] = abs_ring[i]; }
    if (lane == 0) {
        sp->ring_max = ring_max; sp->volts = volts; sp->save_volts = save_volts; sp->fast_backaverage = fba;
        sp->hang_backaverage = hba; sp->gain = gain; sp->out_index = out_index; sp->hang_counter = hang_counter;
        sp->decay_type = decay_type; sp->state = st;
    }
}

static __global__ __launch_bounds__(256) void agc_rewindow_kernel(const int *chan_list, AgcState *state, const double2 *lring, const double *labs,
                                                                   const int *lout)
{
    const int ch = chan_list[blockIdx.x];
    AgcState *sp = state + ch;
    const int ol = sp->out_index;
    const long long og = lout[ch];
    // xwcpagc looks a moved window over only when the sample that leaves is `> 0.0` (wcpAGC.c:197).  Where the reference's per-block
    // transforms leave 0.0 (behind a stage just switched on, a muted input) an overlap-save tile that also holds signal leaves its
    // rounding floor, 1e-17 of that signal: entries below 1e-13 of the largest magnitude of the last RB_SIZE samples are taken for the
    // zeros they are in the reference (tests/test_gpu_wcpagc_batch.py::test_attack_window_moved_while_the_ring_holds_exact_zeros).
    __shared__ double red[4];
    double m = 0.0;
    for (int k = threadIdx.x; k < kAgcLongRing; k += blockDim.x) m = fmax(m, labs[(long long)ch * kAgcLongRing + k]);
    m = wave_max_d(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const double floor_ = 1e-13 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    for (int k = 1 + threadIdx.x; k < kAgcRing; k += blockDim.x) {
        const int src = (int)((og + k) % kAgcLongRing), dst = (ol + k) & (kAgcRing - 1);
        const double a = labs[(long long)ch * kAgcLongRing + src];
        sp->ring[dst] = a < floor_ ? make_double2(0.0, 0.0) : lring[(long long)ch * kAgcLongRing + src];
        sp->abs_ring[dst] = a < floor_ ? 0.0 : a;
    }
}

// The same loop, 64 samples per step of the wavefront.  Everything that does not depend on `volts` is done by all lanes at once:
// the batch's inputs go into the ring, the delayed samples and their magnitudes come out of it, and lane j takes the rescan result
// of sample j -- the maximum over the attack window (out_j, in_j] of the magnitude ring, which is what the reference's rescan would
// find -- with attack_buffsize LDS reads.  The sequential part then steps the level detector alone: per sample three values out of
// the lanes (v_readlane), the two back-averages, the incremental ring_max logic (kept literally, stale values and all: after
// SetRXAAGCAttack shrinks the window the reference's ring_max can outlive the samples it came from), the five-state switch on
// scalar branches, and the new `volts` into lane j.  (The switch written as selects instead, the same cost in every state: slower,
// 100 against 76 ms for 256 channels x 2^18 samples; the sample-by-sample form: 164 ms.)  The gain curve (a log10 and a division per sample) and the multiply are
// lane-parallel again.  Same state, same arithmetic in the same order as wcpagc_seq_kernel: the two are interchangeable between calls
// and bit-identical (tests/test_gpu_wcpagc_batch.py).
static __global__ __launch_bounds__(64) void wcpagc_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                           const AgcParam *prm, AgcState *state, double pre_gain = 1.0)
{
    __shared__ double2 ring[kAgcRing];
    __shared__ double abs_ring[kAgcRing];
    const int ch = chan_list[blockIdx.x];
    const int lane = threadIdx.x;
    const AgcParam q = prm[ch];
    AgcState *sp = state + ch;
    for (int i = lane; i < kAgcRing; i += 64) { ring[i] = sp->ring[i]; abs_ring[i] = sp->abs_ring[i]; }
    double ring_max = sp->ring_max, volts = sp->volts, save_volts = sp->save_volts, fba = sp->fast_backaverage,
           hba = sp->hang_backaverage;
    int out_index = sp->out_index, hang_counter = sp->hang_counter, decay_type = sp->decay_type, st = sp->state;
    const int A = q.attack_buffsize;
    const int BS = A + 64 < kAgcRing ? 64 : kAgcRing - A - 1;           // the batch's inputs must not land on its own outputs
    __syncthreads();
    double2 *p = buf + (long long)ch * stride;
    for (int base = 0; base < n; base += BS) {
        const int cnt = n - base < BS ? n - base : BS;
        double2 z = make_double2(0, 0);
        if (lane < cnt) z = p[base + lane];
        z.x *= pre_gain; z.y *= pre_gain;           // the FM limiter's lim_pre_gain (fmd.c:181-182); 1 for the AGC proper
        const double abs_in = q.pmode == 0 ? fmax(fabs(z.x), fabs(z.y)) : sqrt(__builtin_fma(z.x, z.x, z.y * z.y));
        const int so = (out_index + 1 + lane) & (kAgcRing - 1), si = (so + A) & (kAgcRing - 1);
        if (lane < cnt) { ring[si] = z; abs_ring[si] = abs_in; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double2 o = make_double2(0, 0);
        double abs_out = 0.0, scan = 0.0;
        if (lane < cnt) {
            o = ring[so]; abs_out = abs_ring[so];
            // what a rescan at this sample finds.  The slots behind in_j hold this batch's LATER inputs: they are not in the window
            for (int k = 1; k <= A; k++) scan = fmax(scan, abs_ring[(so + k) & (kAgcRing - 1)]);
        }
        double vj = 0.0;                            // volts after sample j, in lane j
        for (int i = 0; i < cnt; i++) {
            const double a_out = lane_bcast(abs_out, i), a_in = lane_bcast(abs_in, i);
            fba = __builtin_fma(q.fast_backmult, a_out, q.onemfast_backmult * fba);
            hba = __builtin_fma(q.hang_backmult, a_out, q.onemhang_backmult * hba);
            if (a_out >= ring_max && a_out > 0.0) ring_max = lane_bcast(scan, i);
            if (a_in > ring_max) ring_max = a_in;
            if (hang_counter > 0) --hang_counter;
            const bool up = ring_max >= volts;
            switch (st) {
            case 0:
                if (up) volts = __builtin_fma(ring_max - volts, q.attack_mult, volts);
                else if (volts > q.pop_ratio * fba) { st = 1; volts = __builtin_fma(ring_max - volts, q.fast_decay_mult, volts); }
                else if (q.hang_enable && hba > q.hang_level) { st = 2; hang_counter = q.hang_count_init; decay_type = 1; }
                else { st = 3; volts = __builtin_fma(ring_max - volts, q.decay_mult, volts); decay_type = 0; }
                break;
            case 1:
                if (up) { st = 0; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else if (volts > save_volts) volts = __builtin_fma(ring_max - volts, q.fast_decay_mult, volts);
                else if (hang_counter > 0) st = 2;
                else if (decay_type == 0) { st = 3; volts = __builtin_fma(ring_max - volts, q.decay_mult, volts); }
                else { st = 4; volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts); }
                break;
            case 2:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else if (hang_counter == 0) { st = 4; volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts); }
                break;
            case 3:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else volts = __builtin_fma(ring_max - volts, q.decay_mult, volts);
                break;
            default:
                if (up) { st = 0; save_volts = volts; volts = __builtin_fma(ring_max - volts, q.attack_mult, volts); }
                else volts = __builtin_fma(ring_max - volts, q.hang_decay_mult, volts);
                break;
            }
            if (volts < q.min_volts) volts = q.min_volts;
            if (lane == i) vj = volts;
        }
        if (lane < cnt) {
            const double mult = __builtin_fma(-q.slope_constant, fmin(0.0, log10(q.inv_max_input * vj)), q.out_target) / vj;
            p[base + lane] = make_double2(o.x * mult, o.y * mult);
        }
        out_index = (out_index + cnt) & (kAgcRing - 1);
        __builtin_amdgcn_wave_barrier();            // the next batch's inputs overwrite slots this one has read
    }
    __syncthreads();
    for (int i = lane; i < kAgcRing; i += 64) { sp->ring[i] = ring[i]; sp->abs_ring[i] = abs_ring[i]; }
    if (lane == 0) {
        sp->ring_max = ring_max; sp->volts = volts; sp->save_volts = save_volts; sp->fast_backaverage = fba;
        sp->hang_backaverage = hba; if (n > 0) sp->gain = volts * q.inv_out_target; sp->out_index = out_index;
        sp->hang_counter = hang_counter; sp->decay_type = decay_type; sp->state = st;
    }
}

// xwcpagc mode 0 (wcpAGC.c:167-175) in place, for the channels whose fixed gain cannot wait for the output matrix
static __global__ __launch_bounds__(NT) void scale_kernel(double2 *buf, long long stride, int n, const int *chan_list, const double *gain)
{
    const int ch = chan_list[blockIdx.y];
    const double g = gain[ch];
    double2 *p = buf + (long long)ch * stride;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) p[i] = make_double2(g * p[i].x, g * p[i].y);
}

// xanf (wdsp/anf.c:82-133) / xanr (wdsp/anr.c:82-133): leaky normalised-LMS line enhancer on the real part, up to 64 taps.
// The recurrence is sequential in time (the weight update needs the error of the full dot product), so one wavefront
// per channel with one tap per lane: lane j keeps w[j] and x[n - delay - j]; per sample the window moves one lane up
// (DPP wave_shr:1, lane 0 takes x[n - delay]), w.x and x.x are reduced across the wave (wave_sum_d), the step-size
// logic runs uniformly in every lane in the reference's operation order, and the weights update in place.  Lane i of
// a 64-sample batch keeps output i for one coalesced store.  State: the weights, the last 128 inputs, lidx, ngamma.
struct LmsParam { int taps, delay, is_anr, pad; double two_mu, gamma, lidx_min, lidx_max, den_mult, lincr, ldecr; };
struct LmsState { double w[64]; double hist[128]; double lidx, ngamma; };

static __global__ __launch_bounds__(64) void lms_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                        const LmsParam *prm, LmsState *state)
{
#pragma clang fp contract(off)
    __shared__ double s[192];
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    const LmsParam q = prm[ch];
    LmsState *sp = state + ch;
    double2 *p = buf + (long long)ch * stride;
    const bool tap = lane < q.taps;
    double w = tap ? sp->w[lane] : 0.0;
    double h0 = sp->hist[lane], h1 = sp->hist[64 + lane];            // the 128 samples before this call, oldest first
    double lidx = sp->lidx, ngamma = sp->ngamma;
    // window before the first sample: x[-1 - delay - lane]
    s[lane] = h0; s[64 + lane] = h1;
    __syncthreads();
    double xw = s[127 - q.delay - lane >= 0 ? 127 - q.delay - lane : 0];
    __syncthreads();
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        const double xin = lane < cnt ? p[base + lane].x : 0.0;
        // x[base + lane - delay] from the previous 64 samples and this batch
        s[lane] = h1; s[64 + lane] = xin;
        __syncthreads();
        const double xdel = s[64 + lane - q.delay];
        __syncthreads();
        double myout = 0.0;
        // sigma = sum of the window's squares: summed afresh once per batch, then moved along with the window (one sample
        // in, one out); what the running form drifts by in 64 steps is ~1e-16 of sigma, against the 1e-10 added to it
        double sigma = wave_sum_d(tap ? xw * xw : 0.0);
        for (int i = 0; i < cnt; i++) {
            const double xn = lane_bcast(xin, i), xd = lane_bcast(xdel, i);
            const double xold = lane_bcast(xw, q.taps - 1);
            xw = wave_shr1(xw);
            if (lane == 0) xw = xd;
            sigma = (sigma - xold * xold) + xd * xd;
            // 1 / (sigma + 1e-10): v_rcp_f64 and two Newton steps (the weights' dependency chain waits for nothing else)
            const double den = sigma + 1e-10;
            double inv_sigp = __builtin_amdgcn_rcp(den);
            inv_sigp = __builtin_fma(__builtin_fma(-den, inv_sigp, 1.0), inv_sigp, inv_sigp);
            inv_sigp = __builtin_fma(__builtin_fma(-den, inv_sigp, 1.0), inv_sigp, inv_sigp);
            const double y = wave_sum_d(w * xw);
            const double error = xn - y;
            if (lane == i) myout = q.is_anr ? y : error;
            double nel = error * (1.0 - q.two_mu * sigma * inv_sigp);
            if (nel < 0.0) nel = -nel;
            double nev = xn - (1.0 - q.two_mu * ngamma) * y - q.two_mu * error * sigma * inv_sigp;
            if (nev < 0.0) nev = -nev;
            if (nev < nel) { lidx += q.lincr; if (lidx > q.lidx_max) lidx = q.lidx_max; }
            else { lidx -= q.ldecr; if (lidx < q.lidx_min) lidx = q.lidx_min; }
            ngamma = q.gamma * (lidx * lidx) * (lidx * lidx) * q.den_mult;
            const double c0 = 1.0 - q.two_mu * ngamma, c1 = q.two_mu * error * inv_sigp;
            if (tap) w = c0 * w + c1 * xw;
        }
        if (lane < cnt) p[base + lane] = make_double2(myout, 0.0);
        // history moves on by cnt samples
        s[lane] = h0; s[64 + lane] = h1; s[128 + lane] = xin;
        __syncthreads();
        h0 = s[lane + cnt]; h1 = s[64 + lane + cnt];
        __syncthreads();
    }
    if (tap) sp->w[lane] = w;
    sp->hist[lane] = h0; sp->hist[64 + lane] = h1;
    if (lane == 0) { sp->lidx = lidx; sp->ngamma = ngamma; }
}

// AM squelch, wdsp/amsq.c: xamsqcap (:189-192) keeps the signal behind nbp0, xamsq (:119-187) at the end of the chain runs a
// five-state machine (muted / raised-cosine up / unmuted / tail / raised-cosine down) on the 10 ms average of its magnitude.
// The average is a linear recurrence (wave scan); the state machine is sequential: one wavefront per channel, uniform state,
// lane i keeps the gain of sample i.  cup / cdown are the reference's slew tables (compute_slews, amsq.c:28-46).
struct AmsqParam { double avm, onem_avm, tail_thresh, unmute_thresh, min_tail, max_tail, muted_gain, rate; int ntup, ntdown; };
struct AmsqState { double avsig; int state, count; };

static __global__ __launch_bounds__(NT) void amsq_cap_kernel(const double2 *buf, long long stride, int n, const int *chan_list, double *mag,
                                                             long long mag_stride)
{
    const int ch = chan_list[blockIdx.y];
    const double2 *p = buf + (long long)ch * stride;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT)
        mag[(long long)ch * mag_stride + i] = sqrt(p[i].x * p[i].x + p[i].y * p[i].y);
}

static __global__ __launch_bounds__(64) void amsq_apply_kernel(double2 *out, long long out_stride, int n, const int *chan_list, const double *mag,
                                                               long long mag_stride, const AmsqParam *prm, AmsqState *state,
                                                               const double *cup, const double *cdown)
{
    enum { MUTED, INCREASE, UNMUTED, TAIL, DECREASE };
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    const AmsqParam q = prm[ch];
    AmsqState st = state[ch];
    double2 *p = out + (long long)ch * out_stride;
    const double *m = mag + (long long)ch * mag_stride;
    const double pw = lane_pow(q.avm, lane + 1);
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        const double sig = lane < cnt ? m[base + lane] : 0.0;
        const double av = scan_pole(q.onem_avm * sig, q.avm, lane) + pw * st.avsig;       // avsig = avm avsig + onem_avm sig
        double myg = 1.0;
        for (int i = 0; i < cnt; i++) {
            const double a = lane_bcast(av, i);
            double g = 1.0;
            switch (st.state) {
            case MUTED:
                if (a > q.unmute_thresh) { st.state = INCREASE; st.count = q.ntup; }
                g = q.muted_gain;
                break;
            case INCREASE:
                g = cup[q.ntup - st.count];
                if (st.count-- == 0) st.state = UNMUTED;
                break;
            case UNMUTED:
                if (a < q.tail_thresh) {
                    st.state = TAIL;
                    const double siglimit = a > 1.0 ? 1.0 : a;
                    st.count = (int)((q.min_tail + (q.max_tail - q.min_tail) * (1.0 - siglimit)) * q.rate);
                }
                break;
            case TAIL:
                if (a > q.unmute_thresh) st.state = UNMUTED;
                else if (st.count-- == 0) { st.state = DECREASE; st.count = q.ntdown; }
                break;
            default:
                g = cdown[q.ntdown - st.count];
                if (st.count-- == 0) st.state = MUTED;
                break;
            }
            if (lane == i) myg = g;
        }
        st.avsig = lane_bcast(av, cnt - 1);
        if (lane < cnt && myg != 1.0) { const double2 v = p[base + lane]; p[base + lane] = make_double2(myg * v.x, myg * v.y); }
    }
    if (lane == 0) state[ch] = st;
}

}  // namespace qh
