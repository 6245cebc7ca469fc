// qh_osfir.hpp -- batched overlap-save complex FIR + integer decimation for gfx950.
//
// One workgroup (256 threads) computes one tile of one receiver channel:
//
//     y[m] = sum_k h[k] * x'[D*m + off - k],      x'[n] = x[n] * exp(j*(phase0 + n*delta))   (MIX)
//
// or, for OUTMIX, the same thing with the oscillator moved behind the filter:
//
//     y[m] = exp(j*(phase0 + (D*m + off)*delta)) * sum_k (h[k] exp(-j*k*delta)) * x[D*m + off - k]
//
// (an identity while delta is constant over the taps' span; the engine rewrites the short history when it changes):
// the mask is the spectrum of the modulated taps, one per channel, and the rotation costs one multiply per OUTPUT
// sample -- a D-th of the input-side form -- with the phasor taken from two tables, exp(j*Phi_tile) * exp(j*D*delta*t)
// (tile_rot, lane_rot), instead of a sincospi per lane and tile.
//
// which is, with D = 1: WDSP's partitioned overlap-save "fircore" (wdsp/firmin.c:409-430: the
// partitions sum to one causal linear convolution with the nc-tap complex impulse), and with
// real taps and D > 1: WDSP's polyphase resampler for L = 1 (wdsp/resample.c:120-157, preceded
// by xshift, wdsp/shift.c:60-85, when MIX) and Quisk's quisk_cDecimate / quisk_cCDecimate
// (filter.c:203-257; off = decim - 1 - decim_index).
//
// Per tile: NFFT inputs (P = pre-roll >= ntaps-1 samples of history + Lout*D new ones) are
// loaded straight from HBM into registers (16 B per lane, consecutive lanes), rotated by the
// NCO, transformed (qh_fft.hpp), multiplied by the frequency-domain mask H = FFT(h)/NFFT and
// folded D-fold (decimation in time == aliasing sum in frequency) without leaving registers,
// inverse transformed at NFFT/D points, and the Lout valid outputs are stored straight from
// registers through a per-channel 2x2 real epilogue (fixed AGC gain and the patch panel,
// wdsp/wcpAGC.c:167-175, wdsp/patchpanel.c:55-101).
//
// HBM traffic per tile = NFFT*16 B read (P/NFFT of it re-read history) + Lout*16 B written;
// masks and twiddles are L2 resident.
#pragma once
#include "qh_fft.hpp"
#include "qh_wave.hpp"
#include "qh_ingest.hpp"
#include "qh_egress.hpp"

namespace qh {

// NCO state, one entry per channel (struct of arrays so that the phase can live on the device):
//   phase  : phase at input index 0 of the call, in turns * 2^64 (wraps exactly like angle mod 2*pi)
//   dphase : phase step per input sample, turns * 2^64
//   step   : exp(j*2*pi*NT*dphase), the rotation for a jump of NT samples

struct EpiParam {                 // per channel 2x2 real output matrix: [re';im'] = [[a,b],[c,d]] [re;im]
    double a, b, c, d;
};

template <typename T> struct OsfirArgs {
    const cplx<T> *in;            // [nch][in_stride]; element 0 = first new sample of this call
    const cplx<T> *hist;          // [nch][hist_stride]; the hist_len samples that precede in[0] (already mixed)
    cplx<T> *hist_next;           // plain (not MIX / PACKED / PAIR) kernels: where the delay line for the next call goes, or null (hist_update_kernel does it)
    cplx<T> *out;                 // [nch][out_stride]; output m is written at out_offset + m
    const cplx<T> *mask;          // [nch or 1][NFFT]  FFT(h)/NFFT
    const cplx<T> *tw_fwd;        // pass tables for NFFT
    const cplx<T> *tw_inv;        // pass tables for NFFT/D
    const unsigned long long *nco_phase;   // [nch] (MIX only)
    const unsigned long long *nco_dphase;  // [nch]
    const double2 *nco_step;               // [nch]
    const EpiParam *epi;          // [nch] or null
    const int *chan_list;         // null: channel = slot; else channel = chan_list[slot] (sub-set launches), slot from xcd_tile_map
    long long in_stride, hist_stride, out_stride, mask_stride;
    long long out_offset;
    int hist_len;
    int n_in;                     // new samples available in `in`
    int n_out;                    // outputs to produce
    int off;                      // decimation phase
    int P;                        // pre-roll, multiple of D, >= ntaps - 1
    int Lout;                     // outputs per tile, <= (NFFT - P) / D
    int ntiles;                   // ceil(n_out / Lout)
    int pick;                     // 0/1: every folded sample is an output; k > 1: every k-th (n_out counts final outputs)
    const unsigned char *pk_src;  // PACKED kernels: the wire-format input (qh_ingest.hpp) instead of `in`
    PackedFmt pk;
    // METER kernels (D = 1, P and Lout multiples of 256): per 64-sample chunk of the stage's input (meter_in) and of its
    // output ahead of the epilogue (meter_out), x = sum_i w[i] |z_i|^2 and y = max_i |z_i|^2 -- what xmeter's one-pole
    // average and block peak (wdsp/meter.c:75-108) need from the chunk; meter_finish_kernel walks them in time order
    // OUTMIX kernels: phasor of the NCO at input index g0 of every tile, at D*t for lane t, at D*256 (nco_step)
    const double2 *tile_rot;            // [nch][ntiles]
    const double2 *lane_rot;            // [nch][NT]
    EgressFmt eg;                       // EGRESS kernels: the outputs leave as audio frames (qh_egress.hpp) instead of through `out`
    double2 *meter_in, *meter_out;      // [nch][meter_stride] chunk partials, chunk c = samples 64c .. 64c + 63 of this call
    long long meter_stride;
    const double *meter_w;              // [64]  (1 - m) m^(63 - i)
    const double2 *tw_r2;               // osfir8k_kernel: exp(-2 pi i k / 8192), k < 256
    // DET kernels: the stage feeds a detector only, and the detector's first step rides in the store.  One double per output leaves
    // (det_out [nch][det_stride]) instead of the complex sample:
    //   DET 1  arg z in turns (xfmd's loop takes nothing else: qh_tiled.hpp, pll_theta_kernel)
    //   DET 2  |z| (xamd's envelope, amd.c:131-133), and per tile the response of the fade leveller's two averages to the tile's own
    //          magnitudes from a zero state (amd.c:136-137): det_sum [nch][det_sum_stride][2] = g sum_i m^(Lout - 1 - i) |z_i|
    double *det_out;
    long long det_stride;
    double *det_sum;
    long long det_sum_stride;           // tiles per channel row
    double det_m[2], det_m256[2], det_g[2];
    int pair_im0;                       // PAIR kernels: the unpaired equivalent would see (y, 0) (Quisk's real chains) instead of (y, y)
    // PAIR kernels behind xfmd's loop in its local-dc form (qh_tiled.hpp: pll_lanes_kernel local_dc, fm_dc_chain_kernel): the stage's
    // input never exists as complex samples -- sample g of a channel is again (a_local[g] - cin[g >> shift] pw[g & (L - 1)]), the dc
    // removal and gain of fmd.c:169-171 taken in the load (one 8-byte array read instead of a pass that writes 16 and a load that reads them)
    const double *fmdc_a;               // [nch][fmdc_stride] doubles; null: the plain PAIR load
    long long fmdc_stride;
    const double *fmdc_cin;             // [nch][fmdc_cstride]: fmdc ahead of every tile of 2^shift samples
    long long fmdc_cstride;
    const double *fmdc_pw;              // mtau^(k + 1), k < 2^shift
    const double *fmdc_gain;            // [nch] again
    int fmdc_shift;
    // DET 3 (xamd's envelope AND the fade leveller's share of the tile, see the kernel) and the PAIR stage behind it (bp1):
    const int *det_lf;                  // [nch] levelfade flags (amd.c:134)
    const double *det_scan;             // [2][3][64]: the lanes' scan weights of the two averages (PoleScan pa, pb, pw), made on the host
    double det_mp[2][4];                // m, m^2, m^4, m^8 of the two averages
    double *det_last;                   // [nch][2]: the two averages' local values at the call's last sample (the tile the call's end cuts short)
    const double *amlv_a;               // PAIR: [nch][amlv_stride] doubles left by DET 3; null: the plain PAIR load
    long long amlv_stride;
    const double *amlv_cin;             // [nch][amlv_cstride][2]: the two averages ahead of every tile of 2^amlv_shift samples
    long long amlv_cstride;
    const double *amlv_pw;              // [2][2^amlv_shift]: mtauR^(k + 1), mtauI^(k + 1)
    int amlv_shift;
    double2 *stash;                     // osfir8s_kernel: [nch][4096] scratch for the tile that the end of the call cuts short
};


// the audio of sample g in xfmd's local-dc form (OsfirArgs::fmdc_*)
__device__ __forceinline__ double fm_audio_at(const double *a_local, const double *cin, const double *pw, int shift, double gain, long long g)
{
    const long long t = g >> shift;
    const int k = (int)(g - (t << shift));
    return gain * __builtin_fma(-cin[t], pw[k], a_local[g]);
}

template <typename T> __device__ __forceinline__ void sincos_turns(unsigned long long ph, T &c, T &s);
template <> __device__ __forceinline__ void sincos_turns<double>(unsigned long long ph, double &c, double &s)
{
    // top 53 bits -> turns in [0,1); sincospi(2*turns)
    double t = (double)(ph >> 11) * (1.0 / 9007199254740992.0);
    sincospi(2.0 * t, &s, &c);
}
template <> __device__ __forceinline__ void sincos_turns<float>(unsigned long long ph, float &c, float &s)
{
    double t = (double)(ph >> 11) * (1.0 / 9007199254740992.0);
    double sd, cd;
    sincospi(2.0 * t, &sd, &cd);           // keep the phase in double; only the product is fp32
    c = (float)cd; s = (float)sd;
}

// Issue the NFFT/NT loads of one tile for this lane.  Branch-free: every lane always issues all its
// loads back to back (clamped address; `okbits` says which values are real) so the requests are all in
// flight together and nothing waits until the values are used.
template <typename T, int E>
__device__ __forceinline__ unsigned load_tile(cplx<T> (&x)[E], const cplx<T> *__restrict__ in,
                                              const cplx<T> *__restrict__ hist, int hist_len, int n_in, int g0)
{
    using C = cplx<T>;
    const C *hsafe = hist ? hist : in;
    const int hlen = hist ? hist_len : 0;
    const int last = n_in - 1;
    unsigned okbits = 0;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int g = g0 + (int)threadIdx.x + r * NT;
        const int gi = g < 0 ? 0 : (g > last ? last : g);
        const int gh = g + hlen < 0 ? 0 : g + hlen;
        const C *p = g >= 0 ? in + gi : hsafe + (g < 0 ? gh : 0);
        const bool ok = g >= 0 ? g <= last : (g + hlen >= 0);
        okbits |= (ok ? 1u : 0u) << r;
        x[r] = *p;
    }
    return okbits;
}

// XCD-aware workgroup -> (channel, tile) map.  Workgroup ids go round the 8 XCDs of the chip, each with its own
// L2, and consecutive tiles of a channel share NFFT - D*Lout input samples (half of a tile for the nc = 2048
// band-pass).  With the plain (tile, channel) grid neighbouring tiles land on different XCDs and both fetch the
// shared half across the fabric (PMC: 2.1x the input size for the D = 1 kernel).  Here the launch is one-dimensional
// over G = tiles * channels workgroups; XCD c (= id % 8) walks ITS contiguous eighth of the channel-major tile list
// in dispatch order, so a tile's neighbour runs on the same XCD right after it and finds the overlap in L2.
#ifndef QH_XCD_SWIZZLE
#define QH_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ void xcd_tile_map(int ntiles, int &chan_slot, int &tile)
{
    const unsigned G = gridDim.x, lin = blockIdx.x;
#if QH_XCD_SWIZZLE
    const unsigned c = lin & 7u, k = lin >> 3, q = G >> 3, rem = G & 7u;
    const unsigned gt = c * q + (c < rem ? c : rem) + k;
#else
    const unsigned gt = lin;
    (void)G;
#endif
    chan_slot = (int)(gt / (unsigned)ntiles);
    tile = (int)(gt - (unsigned)chan_slot * (unsigned)ntiles);
}

// dynamic LDS of the two kernels below
template <typename T, int NFFT, int D, bool METER = false> constexpr int osfir_lds_bytes()
{
    constexpr int a = TileFft<NFFT, false, cplx<T>>::kLdsBytes, b = TileFft<NFFT / D, true, cplx<T>>::kLdsBytes;
    constexpr int m = METER ? NT / 64 * 2 * 64 * 9 * 8 : 0;        // the meter taps' per-wave blocks overlay the image (kMeterLdsDoublesPerWave)
    return (a > b ? a : b) > m ? (a > b ? a : b) : m;
}
template <typename T, int NFFT, int U> constexpr int osfir_interp_lds_bytes()
{
    constexpr int a = TileFft<NFFT / U, false, cplx<T>>::kLdsBytes, b = TileFft<NFFT, true, cplx<T>>::kLdsBytes;
    return a > b ? a : b;
}

// One workgroup = one tile of one channel (xcd_tile_map above).  Straight-line code: a persistent tile loop with
// register prefetch was tried and costs more in registers than it gains (tools/ab_bench.py, profiles/r01_notes.md);
// latency is hidden by the three or four workgroups a CU holds.
// Waves per SIMD the register allocator must leave room for: with the split LDS exchange three (or four) fp64
// workgroups fit a CU, provided each stays within 168 (128) VGPRs.
// Measured (tools/ab_bench.py, C2): the D = 1 kernel gains from four waves despite 2 spilled registers
// (3.41 -> 2.77 ms), the decimating ones are best at three (128 VGPRs cost them 14 spills).
#ifndef QH_MASK_BATCH
#define QH_MASK_BATCH 8
#endif
#ifndef QH_OSFIR_WAVES_F64_D1
#define QH_OSFIR_WAVES_F64_D1 4
#endif
#ifndef QH_OSFIR_WAVES_F64
#define QH_OSFIR_WAVES_F64 3
#endif
#ifndef QH_OSFIR_WAVES_F64_OUTMIX
#define QH_OSFIR_WAVES_F64_OUTMIX 3
#endif
template <typename T, int D, bool OUTMIX = false, int NFFT = 4096> constexpr int osfir_min_waves()
{
    if (NFFT >= 8192) return sizeof(T) == 8 ? 1 : 2;        // 32 elements per lane: one 139 KB (fp64) image per CU anyway
    return sizeof(T) == 8 ? (D == 1 ? QH_OSFIR_WAVES_F64_D1 : OUTMIX ? QH_OSFIR_WAVES_F64_OUTMIX : QH_OSFIR_WAVES_F64) : 4;
}

#ifdef QH_OSFIR_PROBE      // tools/ubench/osfir_phase.hip only: shader-clock stamps of one workgroup per phase
__device__ long long g_osfir_probe[16];
#define QH_OPROBE(slot) do { if (blockIdx.x == 7 * 8 && threadIdx.x == 0) g_osfir_probe[slot] = clock64(); } while (0)
#else
#define QH_OPROBE(slot) do { } while (0)
#endif


// ---- fused meters -----------------------------------------------------------------------------------------------
// Lane t of a tile holds samples t + 256 r in register r, so a wavefront holds one 64-sample chunk per register.  The
// eight chunks of a register group are reduced together: every lane writes its eight weighted magnitudes (and the
// eight plain ones) into the wave's own LDS block as [register][lane], reads back eight consecutive lanes of one
// register (segments of 8 doubles at a pitch of 9: conflict-free 64-bit accesses), adds / maxes them and finishes
// across the eight lanes that share a register with three DPP steps.  No workgroup barrier inside; the caller
// guarantees that nobody else uses the LDS image meanwhile.
template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int kMeterLdsDoublesPerWave = 2 * 64 * 9;
     // (sum set, max set) x 64 segments x pitch 9

// max of two numbers that are not NaN (squared magnitudes): fmax() would first re-quiet operands the compiler cannot
// prove canonical (everything that comes back from LDS or a DPP move) with a v_max_f64 x, x each
__device__ __forceinline__ double max_nn(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x[r0 + k], k = 0 .. nseg - 1 (nseg = E - r0), are the registers to meter.  The wave's nseg partials go out as one
// contiguous run: dst[wave * nseg + k] (meter_finish_kernel knows that chunk 4 k + wave of the tile sits there).
template <typename C, int E>
__device__ __forceinline__ void meter_tap(const C (&x)[E], int r0, double wlane, double *lds_wave, double2 *dst, int wave, int lane, int wstride = -1)
{
    const int ws = wstride < 0 ? E - r0 : wstride;      // a wave's run inside dst (callers that tap a tile's registers in several calls)
    const int wr = (lane >> 3) * 9 + (lane & 7);        // write slot inside a register's 8 segments
    const double *rd = lds_wave + lane * 9;
#pragma unroll
    for (int g = 0; g < E / 8; g++) {
        if (8 * (g + 1) <= r0) continue;                // workgroup-uniform
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const C v = x[8 * g + k];
            double m2 = (double)v.x * (double)v.x;
            m2 = __builtin_fma((double)v.y, (double)v.y, m2);       // wdsp/meter.c:90 (contracted)
            lds_wave[k * 72 + wr] = m2 * wlane;
            lds_wave[64 * 9 + k * 72 + wr] = m2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double sum = rd[0], mx = rd[64 * 9];
#pragma unroll
        for (int k = 1; k < 8; k++) { sum += rd[k]; mx = max_nn(mx, rd[64 * 9 + k]); }
        sum += dpp_mov_d<0xB1>(sum); mx = max_nn(mx, dpp_mov_d<0xB1>(mx));        // quad_perm [1,0,3,2]
        sum += dpp_mov_d<0x4E>(sum); mx = max_nn(mx, dpp_mov_d<0x4E>(mx));        // quad_perm [2,3,0,1]
        sum += dpp_mov_d<0x141>(sum); mx = max_nn(mx, dpp_mov_d<0x141>(mx));      // row_half_mirror
        const int r = 8 * g + (lane >> 3);
        if ((lane & 7) == 0 && r >= r0) dst[wave * ws + (r - r0)] = make_double2(sum, mx);
        __builtin_amdgcn_wave_barrier();                // the next group overwrites the block
    }
}

// POLY: the forward transform stops ahead of the stage that combines the D decimated sequences and the mask holds the
// polyphase spectra G (FftSplit4096::run_poly, front_mask_kernel): same fold below, one butterfly stage less.
// DET: see OsfirArgs (8 bytes per output instead of 16, and no pass of the detector's own over the stage's output).
// PAIR: a REAL filter (its mask is the transform of real taps) behind a detector, whose output is one real signal written to both
// components (xamd: amd.c:139-140, xfmd: fmd.c:170-171): two channels with the same mask share a tile -- channel A in the real part,
// channel B in the imaginary part, one transform pair for both, and (y, y) goes to each one's row.  chan_list holds the pairs
// (A0, B0, A1, B1, ...; a channel without a partner is paired with itself).
template <typename T, int NFFT, int D, bool MIX, bool PACKED = false, bool METER = false, bool OUTMIX = false, bool EGRESS = false, bool POLY = false,
          int DET = 0, bool PAIR = false>
__global__ __launch_bounds__(NT, (osfir_min_waves<T, D, OUTMIX, NFFT>())) void osfir_kernel(OsfirArgs<T> a)
{
    using C = cplx<T>;
    constexpr int E = NFFT / NT;            // elements per thread, forward
    constexpr int NOUT = NFFT / D;
    constexpr int EO = E / D;               // elements per thread, inverse
    static_assert(E % D == 0 && EO >= 1, "decimation must divide NFFT/256");
    static_assert(NOUT >= 2 * NT, "NFFT/D must be >= 512");
    static_assert(E <= 32, "validity bits are kept in one word");
    using Fwd = TileFft<NFFT, false, C>;
    using Inv = TileFft<NOUT, true, C>;
    extern __shared__ __align__(16) unsigned char smem[];
    void *lds = smem;

    const int t = threadIdx.x;
    int tile, slot;
    xcd_tile_map(a.ntiles, slot, tile);
    const int ch = PAIR ? a.chan_list[2 * slot] : a.chan_list ? a.chan_list[slot] : slot;
    const int ch_b = PAIR ? a.chan_list[2 * slot + 1] : ch;
    const C *in = a.in + (long long)ch * a.in_stride;
    const C *hist = a.hist ? a.hist + (long long)ch * a.hist_stride : nullptr;
    const int g0 = a.off - a.P + tile * (D * a.Lout);      // input index of element 0 of this tile (Lout counts folded samples)

    // Interior tiles (all NFFT inputs inside this call's buffer: every tile but the first and the last
    // one or two of a channel) take plain loads; edge tiles take the clamped, history-aware path.
    QH_OPROBE(0);
    C x[E];
    const bool interior = (g0 >= 0) && (g0 + NFFT <= a.n_in);       // workgroup-uniform
    if constexpr (PACKED) {
        // wire-format input: new samples are decoded from the packed bytes, the pre-roll comes from the (complex,
        // already mixed) history like in the plain path
        // contiguous records whose 8-byte load windows all end inside the buffer: pointer + constant stride
        const long long base = a.pk.first_offset + (long long)ch * a.pk.chan_stride + (long long)g0 * a.pk.record_stride;
        const bool fast = interior && a.pk.records_per_frame == 0 &&
                          base + (long long)(NFFT - 1) * a.pk.record_stride + 8 <= a.pk.total_bytes;    // workgroup-uniform
        if (fast) {
            const unsigned char *p = a.pk_src + base + (long long)t * a.pk.record_stride;
            const long long step = (long long)NT * a.pk.record_stride;
#pragma unroll
            for (int r = 0; r < E; r++) x[r] = decode_packed_at<T>(p + r * step, a.pk.sel_re, a.pk.sel_im, a.pk.gain);
        } else if (interior) {
#pragma unroll
            for (int r = 0; r < E; r++) x[r] = decode_packed<T>(a.pk_src, a.pk, ch, (long long)(g0 + t + r * NT));
        } else {
#pragma unroll
            for (int r = 0; r < E; r++) {
                const int g = g0 + t + r * NT;
                if (g >= 0) x[r] = g < a.n_in ? decode_packed<T>(a.pk_src, a.pk, ch, (long long)g) : mk<T>(0, 0);
                else x[r] = (hist && g + a.hist_len >= 0) ? hist[g + a.hist_len] : mk<T>(0, 0);
            }
        }
    } else if constexpr (PAIR) {
        static_assert(!MIX && !METER && !OUTMIX && !EGRESS && DET == 0, "pairs ride on a plain stage (any fold: a real filter acts on the two parts alike)");
        const C *in_b = a.in + (long long)ch_b * a.in_stride;
        const C *hist_b = a.hist ? a.hist + (long long)ch_b * a.hist_stride : nullptr;
        if (a.fmdc_a) {                     // workgroup-uniform
            const double *aa = a.fmdc_a + (long long)ch * a.fmdc_stride, *ab = a.fmdc_a + (long long)ch_b * a.fmdc_stride;
            const double *ca = a.fmdc_cin + (long long)ch * a.fmdc_cstride, *cb = a.fmdc_cin + (long long)ch_b * a.fmdc_cstride;
            const double ga = a.fmdc_gain[ch], gb = a.fmdc_gain[ch_b];
            if (interior) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const long long g = (long long)g0 + t + r * NT;
                    const long long tl = g >> a.fmdc_shift;
                    const double pw = a.fmdc_pw[(int)(g - (tl << a.fmdc_shift))];
                    x[r] = mk<T>((T)(ga * __builtin_fma(-ca[tl], pw, aa[g])), (T)(gb * __builtin_fma(-cb[tl], pw, ab[g])));
                }
            } else {
                auto fetch = [&](const double *src, const double *cin, double gain, const C *h, int g) -> T {
                    if (g >= 0) return g < a.n_in ? (T)fm_audio_at(src, cin, a.fmdc_pw, a.fmdc_shift, gain, g) : (T)0;
                    return (h && g + a.hist_len >= 0) ? h[g + a.hist_len].x : (T)0;
                };
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int g = g0 + t + r * NT;
                    x[r] = mk<T>(fetch(aa, ca, ga, hist, g), fetch(ab, cb, gb, hist_b, g));
                }
            }
        } else if (a.amlv_a) {              // workgroup-uniform: behind DET 3 -- audio = a_local + cI mI^(k + 1) - cR mR^(k + 1) (amd.c:136-138)
            const double *aa = a.amlv_a + (long long)ch * a.amlv_stride, *ab = a.amlv_a + (long long)ch_b * a.amlv_stride;
            const double2 *ca = reinterpret_cast<const double2 *>(a.amlv_cin) + (long long)ch * a.amlv_cstride;
            const double2 *cb = reinterpret_cast<const double2 *>(a.amlv_cin) + (long long)ch_b * a.amlv_cstride;
            const double *pwR = a.amlv_pw, *pwI = a.amlv_pw + (1 << a.amlv_shift);
            auto fetch = [&](const double *src, const double2 *cin, const C *h, int g) -> T {
                if (g >= 0) {
                    if (g >= a.n_in) return (T)0;
                    const int tl = g >> a.amlv_shift, k = g - (tl << a.amlv_shift);
                    const double2 c = cin[tl];
                    return (T)(src[g] + __builtin_fma(c.y, pwI[k], -c.x * pwR[k]));
                }
                return (h && g + a.hist_len >= 0) ? h[g + a.hist_len].x : (T)0;
            };
            if (interior) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int g = g0 + t + r * NT;
                    const int tl = g >> a.amlv_shift, k = g - (tl << a.amlv_shift);
                    const double2 c0 = ca[tl], c1 = cb[tl];
                    const double wR = pwR[k], wI = pwI[k];
                    x[r] = mk<T>((T)(aa[g] + __builtin_fma(c0.y, wI, -c0.x * wR)), (T)(ab[g] + __builtin_fma(c1.y, wI, -c1.x * wR)));
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int g = g0 + t + r * NT;
                    x[r] = mk<T>(fetch(aa, ca, hist, g), fetch(ab, cb, hist_b, g));
                }
            }
        } else if (interior) {
            const C *p = in + g0 + t, *pb = in_b + g0 + t;
#pragma unroll
            for (int r = 0; r < E; r++) x[r] = mk<T>(p[r * NT].x, pb[r * NT].x);
        } else {
            auto fetch = [&](const C *src, const C *h, int g) -> T {
                if (g >= 0) return g < a.n_in ? src[g].x : (T)0;
                return (h && g + a.hist_len >= 0) ? h[g + a.hist_len].x : (T)0;
            };
#pragma unroll
            for (int r = 0; r < E; r++) {
                const int g = g0 + t + r * NT;
                x[r] = mk<T>(fetch(in, hist, g), fetch(in_b, hist_b, g));
            }
        }
    } else if (interior) {
        const C *p = in + g0 + t;
#pragma unroll
        for (int r = 0; r < E; r++) x[r] = p[r * NT];
    } else {
        const unsigned ok = load_tile<T, E>(x, in, hist, a.hist_len, a.n_in, g0);
#pragma unroll
        for (int r = 0; r < E; r++)
            if (!((ok >> r) & 1u)) x[r] = mk<T>(0, 0);
    }
    if constexpr (MIX) {
        C rot;
        sincos_turns<T>(a.nco_phase[ch] + a.nco_dphase[ch] * (unsigned long long)(long long)(g0 + t), rot.x, rot.y);
        const double2 st = a.nco_step[ch];
        const C step = mk<T>((T)st.x, (T)st.y);
        if (interior) {
#pragma unroll
            for (int r = 0; r < E; r++) { x[r] = cmul(x[r], rot); rot = cmul(rot, step); }
        } else {
#pragma unroll
            for (int r = 0; r < E; r++) {
                if (g0 + t + r * NT >= 0) x[r] = cmul(x[r], rot);      // history is stored already mixed
                rot = cmul(rot, step);
            }
        }
    }

    if constexpr (METER) {
        static_assert(D == 1 && !MIX && !PACKED, "meters ride on a plain D = 1 stage");
        static_assert(NT / 64 * kMeterLdsDoublesPerWave * 8 <= osfir_lds_bytes<T, NFFT, D, true>(), "meter blocks overlay the exchange image");
        // the tile's new samples [tile * Lout, (tile + 1) * Lout) are chunks tile * Lout / 64 ...; register P / 256 holds the first
        meter_tap<C, E>(x, a.P >> 8, a.meter_w[t & 63], reinterpret_cast<double *>(lds) + (t >> 6) * kMeterLdsDoublesPerWave,
                        a.meter_in + (long long)ch * a.meter_stride + (long long)tile * (a.Lout >> 6), t >> 6, t & 63);
        __syncthreads();                    // the transform's exchange image overlays the waves' meter blocks
    }

    // ---- forward FFT, registers -> registers
    QH_OPROBE(1);
    if constexpr (POLY) Fwd::template run_poly<D>(x, lds, Fwd::load(a.tw_fwd));
    else Fwd::run(x, lds, Fwd::load(a.tw_fwd));
    QH_OPROBE(2);

    // ---- mask multiply + D-fold: lane holds bins t + NT*i; bins t + NT*(i' + EO*q) alias to t + NT*i'
    const C *mask = a.mask + (long long)ch * a.mask_stride;
    C z[EO];
#pragma unroll
    for (int i = 0; i < EO; i++) {
        C acc = cmul(x[i], mask[t + NT * i]);
#pragma unroll
        for (int q = 1; q < D; q++) acc = cadd(acc, cmul(x[i + EO * q], mask[t + NT * (i + EO * q)]));
        z[i] = acc;
        // at most QH_MASK_BATCH mask values in flight: all 16 at once cost the D = 1 kernel its spills (A/B: +0.8 %)
        if (((i + 1) * D) % QH_MASK_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
    }

    // ---- inverse FFT at NOUT points
    QH_OPROBE(3);
    __syncthreads();                        // every lane has finished reading LDS in the last forward pass
    Inv::run(z, lds, Inv::load(a.tw_inv));
    QH_OPROBE(4);
    if constexpr (METER) {
        __syncthreads();                    // other waves may still be reading the exchange image
        int t2 = t;
        asm volatile("" : "+v"(t2));        // the lane's weight and block address are derived again, not carried through both transforms
        meter_tap<C, EO>(z, a.P >> 8, a.meter_w[t2 & 63], reinterpret_cast<double *>(lds) + (t2 >> 6) * kMeterLdsDoublesPerWave,
                         a.meter_out + (long long)ch * a.meter_stride + (long long)tile * (a.Lout >> 6), t2 >> 6, t2 & 63);
    }

    if constexpr (OUTMIX) {
        static_assert(!MIX, "one oscillator");
        const double2 tr = a.tile_rot[(long long)ch * a.ntiles + tile];         // workgroup-uniform: a scalar load (rows by channel, also under a channel list)
        const double2 lr = a.lane_rot[(long long)ch * NT + t];
        const double2 st = a.nco_step[ch];
        C rot = cmul(mk<T>((T)tr.x, (T)tr.y), mk<T>((T)lr.x, (T)lr.y));
        const C step = mk<T>((T)st.x, (T)st.y);
#pragma unroll
        for (int i = 0; i < EO; i++) {
            z[i] = cmul(z[i], rot);
            if (i + 1 < EO) rot = cmul(rot, step);
        }
    }

    // ---- epilogue + store of the Lout valid outputs
    C *out = a.out + (long long)ch * a.out_stride + a.out_offset;
    const int j0 = a.P / D;
    EpiParam ep;
    if (a.epi) ep = a.epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
    if constexpr (PAIR) {
        C *out_b = a.out + (long long)ch_b * a.out_stride + a.out_offset;
        EpiParam eb = ep;
        if (a.epi) eb = a.epi[ch_b];
#pragma unroll
        for (int i = 0; i < EO; i++) {
            const int rel = t + NT * i - j0;
            const long long m = (long long)tile * a.Lout + rel;
            if (rel >= 0 && rel < a.Lout && m < a.n_out) {
                const T ya = z[i].x, yb = z[i].y;
                const T ia = a.pair_im0 ? (T)0 : ya, ib = a.pair_im0 ? (T)0 : yb;       // the imaginary part the unpaired stage would hold
                out[m] = mk<T>((T)ep.a * ya + (T)ep.b * ia, (T)ep.c * ya + (T)ep.d * ia);
                out_b[m] = mk<T>((T)eb.a * yb + (T)eb.b * ib, (T)eb.c * yb + (T)eb.d * ib);
            }
        }
    } else if constexpr (DET == 3) {
        // xamd's envelope (amd.c:131-133) and the fade leveller's two averages (amd.c:136-137) as far as the tile's own samples carry them:
        // what leaves, 8 bytes per output, is mag + (dcI_local - dcR_local) -- the averages' responses to the tile's magnitudes from a
        // zero state -- and per tile their values at the tile's end (det_sum: am_lv_chain_kernel chains them); the share of everything
        // ahead of the tile, cI mI^(k + 1) - cR mR^(k + 1), is added where bp1 loads the sample (the PAIR load above).  am_level_tiled_kernel's
        // pass (8 B read, 16 written per sample, a CU-filling grid of dependent scans) is gone.
        // Geometry fixed by the caller: P = Lout = 2048, so registers 8 .. 15 hold the outputs, register 8 + r samples 256 r + t of the tile.
        static_assert(D == 1 && NT == 256 && NFFT == 4096 && !EGRESS && !METER && !PAIR && !OUTMIX, "the leveller's tap rides on a plain 4096-point stage");
        constexpr int R0 = 8, NR = 8;
        const int lane = t & 63, wv = t >> 6;
        const bool lf = a.det_lf[ch] != 0;
        // Row by row, the transform's values dying as it goes: the magnitude, the two scans inside the wavefront, and ONE value per sample
        // parked in the exchange image (free behind the barrier) -- q = mag + (vI - vR), which is all the last step needs beside the
        // segments' carries; the segments' end values go to `red`.  (Keeping mag, vR, vI of eight rows in registers spilled a hundred.)
        double *sv = reinterpret_cast<double *>(lds);           // [row][t]
        double *red = sv + NR * NT;                             // [segment 4 r + wave][2], then [t][2] for the call's last sample
        static_assert((NR * NT + 2 * 4 * NR + 2) * 8 <= osfir_lds_bytes<T, NFFT, D, false>(), "the leveller's scans fit the exchange image");
        __syncthreads();                        // the inverse transform's last exchange has been read by everyone
        int lane2 = lane;
        asm volatile("" : "+v"(lane2));         // (the scans' weights are formed here, behind the transforms)
        PoleScan sR, sI;                        // (from the host's tables: lane_pow's temporaries beside the transform's 64 registers spilled)
        sR.m1 = a.det_mp[0][0]; sR.m2 = a.det_mp[0][1]; sR.m4 = a.det_mp[0][2]; sR.m8 = a.det_mp[0][3];
        sI.m1 = a.det_mp[1][0]; sI.m2 = a.det_mp[1][1]; sI.m4 = a.det_mp[1][2]; sI.m8 = a.det_mp[1][3];
        sR.pa = a.det_scan[lane2]; sR.pb = a.det_scan[64 + lane2]; sR.pw = a.det_scan[128 + lane2];
        sI.pa = a.det_scan[192 + lane2]; sI.pb = a.det_scan[256 + lane2]; sI.pw = a.det_scan[320 + lane2];
        const long long m0 = (long long)tile * a.Lout + t;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const long long m = m0 + NT * r;
            C v;
            v.x = (T)ep.a * z[R0 + r].x + (T)ep.b * z[R0 + r].y;
            v.y = (T)ep.c * z[R0 + r].x + (T)ep.d * z[R0 + r].y;
            const double mag = m < a.n_out ? sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y) : 0.0;
            const double vR = scan_pole_dpp(a.det_g[0] * mag, sR), vI = scan_pole_dpp(a.det_g[1] * mag, sI);
            sv[r * NT + t] = lf ? mag + (vI - vR) : mag;
            if (lane2 == 63) { red[(4 * r + wv) * 2] = vR; red[(4 * r + wv) * 2 + 1] = vI; }
            if (m == (long long)a.n_out - 1) { red[2 * 4 * NR] = vR; red[2 * 4 * NR + 1] = vI; }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // the 32 segments in order: what each starts from
        const double m64R = ipow_d(a.det_m[0], 64), m64I = ipow_d(a.det_m[1], 64);
        double cR = 0.0, cI = 0.0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            double kR = 0.0, kI = 0.0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (w == wv) { kR = cR; kI = cI; }              // wave-uniform
                cR = __builtin_fma(cR, m64R, red[(4 * r + w) * 2]);
                cI = __builtin_fma(cI, m64I, red[(4 * r + w) * 2 + 1]);
            }
            const long long m = m0 + NT * r;
            if (m < a.n_out) {
                const double cR_here = sR.pw * kR, cI_here = sI.pw * kI;        // the rows' and wavefronts' shares ahead of this sample
                a.det_out[(long long)ch * a.det_stride + a.out_offset + m] = lf ? sv[r * NT + t] + (cI_here - cR_here) : sv[r * NT + t];
                if (m == (long long)a.n_out - 1) { a.det_last[2 * ch] = red[2 * 4 * NR] + cR_here; a.det_last[2 * ch + 1] = red[2 * 4 * NR + 1] + cI_here; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t == 0) {
            double *o = a.det_sum + ((long long)ch * a.det_sum_stride + tile) * 2;
            o[0] = lf ? cR : 0.0; o[1] = lf ? cI : 0.0;
        }
    } else if (a.pick <= 1) {
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int i = 0; i < EO; i++) {
            const int rel = t + NT * i - j0;
            const long long m = (long long)tile * a.Lout + rel;
            double mag = 0.0;
            if (rel >= 0 && rel < a.Lout && m < a.n_out) {
                C v;
                v.x = (T)ep.a * z[i].x + (T)ep.b * z[i].y;
                v.y = (T)ep.c * z[i].x + (T)ep.d * z[i].y;
                if constexpr (DET == 1) {
                    double th = atan2((double)v.y, (double)v.x) * (1.0 / 6.2831853071795864);
                    if (v.x == (T)0 && v.y == (T)0) th = kThetaZeroMark;
                    a.det_out[(long long)ch * a.det_stride + a.out_offset + m] = th;
                } else if constexpr (DET == 2) {
                    mag = sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
                    a.det_out[(long long)ch * a.det_stride + a.out_offset + m] = mag;
                } else if constexpr (EGRESS) egress_store(a.eg, ch, a.out_offset + m, (double)v.x, (double)v.y);
                else out[m] = v;
            }
            if constexpr (DET == 2) {       // sample rel = t - j0 + NT i has weight m^(Lout - 1 - rel): Horner in m^NT over i, m^(NT - 1 - t) below
                acc0 = __builtin_fma(acc0, a.det_m256[0], mag);
                acc1 = __builtin_fma(acc1, a.det_m256[1], mag);
            }
        }
        if constexpr (DET == 2) {
            static_assert(D == 1 && NT == 256 && !EGRESS && !METER, "the envelope tap rides on a plain D = 1 stage");
            // Lout + P = NFFT: the exponent of element i is (NFFT - 1 - t) - NT i = (NT - 1 - t) + NT (EO - 1 - i)
            const double s0 = wave_sum_d(acc0 * ipow_d(a.det_m[0], NT - 1 - t)), s1 = wave_sum_d(acc1 * ipow_d(a.det_m[1], NT - 1 - t));
            double *red = reinterpret_cast<double *>(lds);
            __syncthreads();                    // the inverse transform's last exchange has been read by everyone
            if ((t & 63) == 0) { red[(t >> 6) * 2] = s0; red[(t >> 6) * 2 + 1] = s1; }
            __syncthreads();
            if (t == 0) {
                double *o = a.det_sum + ((long long)ch * a.det_sum_stride + tile) * 2;
                o[0] = a.det_g[0] * ((red[0] + red[2]) + (red[4] + red[6]));
                o[1] = a.det_g[1] * ((red[1] + red[3]) + (red[5] + red[7]));
            }
        }
    } else {
        // total decimation D * pick: keep every pick-th sample of the D-folded result (Lout % pick == 0)
        const int lpt = a.Lout / a.pick;    // final outputs per tile
#pragma unroll
        for (int i = 0; i < EO; i++) {
            const int rel = t + NT * i - j0;
            const int q = rel / a.pick;
            const long long m = (long long)tile * lpt + q;
            if (rel >= 0 && rel < a.Lout && q * a.pick == rel && m < a.n_out) {
                C v;
                v.x = (T)ep.a * z[i].x + (T)ep.b * z[i].y;
                v.y = (T)ep.c * z[i].x + (T)ep.d * z[i].y;
                if constexpr (EGRESS) egress_store(a.eg, ch, a.out_offset + m, (double)v.x, (double)v.y);
                else out[m] = v;
            }
        }
    }
    QH_OPROBE(5);
    if constexpr (!MIX && !PACKED && !PAIR) {
        // The stage's delay line for the next call -- the last hist_len RAW samples of this call's input (hist_update_kernel's job: a launch of its
        // own behind every stage, a dozen on config 4's critical path, four of config 5's 170 us) -- is copied by the tiles whose span holds those
        // samples: the last two or three of a channel (workgroup-uniform test; a sample two tiles share is written twice, the same value).  A copy
        // loop of its own BEHIND the tile's stores: taken from x[] after the loads it cost the /8 front kernel ten registers and DET 3 thirteen
        // spills, and in front of the loads it cost config 2's two kernels 1 % each (same-box A/B against the tree before, profiles/r06_notes.md).
        // The rows of `out` never lie over the rows of `in` when the host asks for this (it asks only when the call is at least hist_len long).
        if (a.hist_next && g0 + NFFT > a.n_in - a.hist_len) {
            C *hn = a.hist_next + (long long)ch * a.hist_stride;
            const int first = a.n_in - a.hist_len;
            const int tq = threadIdx.x;
#pragma unroll 1
            for (int g = max(g0, first) + tq; g < min(g0 + NFFT, a.n_in); g += NT) hn[g - first] = in[g];
        }
    }
}

// ---- D = 1 stage on 8192-point tiles shared by two 256-lane groups ---------------------------------------------------
// A filter of nc <= 2048 taps leaves a 4096-point tile 2049 useful outputs of 4096 (two transforms per 2049 samples); an
// 8192-point tile leaves 6144 of 8192.  A lane cannot hold 32 fp64 complex elements at four waves per SIMD (the one-wave
// osfir_kernel<8192> above is slower than the 4096-point tile), so the tile is shared by two groups of 256 lanes that hold 16
// elements each -- the register budget of the 4096-point kernel -- and split the transform by one radix-2 step in registers:
//   forward (decimation in frequency):  a[n] = x[n] + x[n + 4096],  b[n] = (x[n] - x[n + 4096]) W^n,  W = exp(-2 pi i / 8192)
//                                       X[2k] = FFT4096(a)[k]  (group A),   X[2k + 1] = FFT4096(b)[k]  (group B)
//   inverse (decimation in time):       y[n] = A'[n] + W^-n B'[n],  y[n + 4096] = A'[n] - W^-n B'[n],
//                                       A' = IFFT4096(X[2k] H[2k]),  B' = IFFT4096(X[2k + 1] H[2k + 1])
// Lane j of group g loads x[n] and x[n + 4096] for n = 2048 g + j + 256 s (s = 0 .. 7), so both operands of the forward
// butterflies are its own; it keeps the half its group transforms and hands the other half to lane j of the other group (8
// elements through LDS, conflict free), after which each group holds element j + 256 r of its sequence in register r: the
// layout FftSplit4096 takes.  The inverse mirrors it.  The mask is stored [even bins | odd bins].  P is 2048: the 6144 outputs
// are y[2048 .. 8191]; group A stores y[4096 + j + 256 s], group B y[2048 + j + 256 s] and y[6144 + j + 256 s].
//
// MEASURED (profiles/r02_notes.md): the VALU instruction count per output falls to 0.775 of the 4096-point kernel's as
// planned, but the kernel is SLOWER (2.8 against 2.54 ms for 256 x 2^20 samples): a CU holds two 8-wavefront workgroups
// instead of four 4-wavefront ones, the twenty barriers of a tile stop eight wavefronts at a time, and the VALU is busy 48 %
// of the time instead of 79 %.  (A four-wavefront barrier through an LDS counter, to let the groups drift apart between the
// hand-overs: 4.3 ms.)  The engine therefore runs 4096-point tiles unless qh_rxa_set_band_tile(e, 8192) asks for these.
// The group index must be SCALAR (readfirstlane): as a vector condition it put the meter taps' wave barriers inside
// exec-masked regions and the taps corrupted live registers of the lanes that do not store.
// component-wise select: `c ? a : b` on two lvalues of struct type is an lvalue (a select of ADDRESSES), which pins the
// register arrays to scratch memory
__device__ __forceinline__ double2 sel2(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }
constexpr int kOsfir8kThreads = 512, kOsfir8kP = 2048, kOsfir8kLout = 6144;
constexpr int kOsfir8kImage = FftSplit4096<false, double2>::kLdsBytes > 4 * 2 * 64 * 9 * 8 ? FftSplit4096<false, double2>::kLdsBytes : 4 * 2 * 64 * 9 * 8;   // image or meter blocks
constexpr int osfir8k_lds_bytes() { return 2 * kOsfir8kImage; }

template <bool METER, bool EGRESS>
__global__ __launch_bounds__(kOsfir8kThreads, 4) void osfir8k_kernel(OsfirArgs<double> a)
{
    using C = double2;
    using SF = FftSplit4096<false, C>;
    using SI = FftSplit4096<true, C>;
    constexpr int N = 8192, P = kOsfir8kP, L = kOsfir8kLout;
    extern __shared__ __align__(16) unsigned char smem8k[];
    const int T = threadIdx.x, j = T & 255;
    const int g = __builtin_amdgcn_readfirstlane(T >> 8);      // the group is the same for a whole wavefront: scalar branches
    unsigned char *image = smem8k + (size_t)g * kOsfir8kImage;              // this group's exchange image
    C *mine = reinterpret_cast<C *>(image), *theirs = reinterpret_cast<C *>(smem8k + (size_t)(g ^ 1) * kOsfir8kImage);
    int tile, slot;
    xcd_tile_map(a.ntiles, slot, tile);
    const int ch = a.chan_list ? a.chan_list[slot] : slot;
    const C *in = a.in + (long long)ch * a.in_stride;
    const C *hist = a.hist ? a.hist + (long long)ch * a.hist_stride : nullptr;
    const int g0 = a.off - P + tile * L;
    const int n0 = 2048 * g + j;                        // the lane's first sample inside the tile

    C x[16];                                            // x[s] = tile[n0 + 256 s], x[8 + s] = tile[n0 + 256 s + 4096]
    if (g0 >= 0 && g0 + N <= a.n_in) {                  // workgroup-uniform
        const C *p = in + g0 + n0;
#pragma unroll
        for (int s = 0; s < 8; s++) { x[s] = p[256 * s]; x[8 + s] = p[256 * s + 4096]; }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gi = g0 + n0 + 256 * (r & 7) + (r >> 3) * 4096;
            C v = make_double2(0.0, 0.0);
            if (gi >= 0) { if (gi < a.n_in) v = in[gi]; }
            else if (hist && gi + a.hist_len >= 0) v = hist[gi + a.hist_len];
            x[r] = v;
        }
    }
    if constexpr (METER) {
        // new samples: tile[2048 ..]: group A registers 8 .. 15 (chunks 32 ..), group B all sixteen (chunks 0 .. 31, 64 .. 95);
        // a tile's 96 partials are stored [A: wave][8] then [B: wave][16] (meter_finish_kernel, layout 1)
        double *blk = reinterpret_cast<double *>(image) + (j >> 6) * kMeterLdsDoublesPerWave;
        double2 *dst = a.meter_in + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) + (g ? 32 : 0);
        meter_tap<C, 16>(x, g ? 0 : 8, a.meter_w[j & 63], blk, dst, j >> 6, j & 63);
        __syncthreads();
    }

    // ---- forward radix-2 step and the hand-over
    {
        C w = a.tw_r2[j];                               // W^n0 = W^j (-i)^g
        if (g) w = make_double2(w.y, -w.x);
        const C wstep = make_double2(0.98078528040323044913, -0.19509032201612826785);      // exp(-i pi / 16) = W^256
        C give[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const C sum = cadd(x[s], x[8 + s]);
            const C dif = cmul(make_double2(x[s].x - x[8 + s].x, x[s].y - x[8 + s].y), w);
            if (s < 7) w = cmul(w, wstep);
            x[s] = sel2(g != 0, dif, sum);              // kept: a[n] in group A, b[n] in group B
            give[s] = sel2(g != 0, sum, dif);
        }
#pragma unroll
        for (int s = 0; s < 8; s++) theirs[256 * s + j] = give[s];
        __syncthreads();
        // group A: a[j + 256 r] = own (r < 8), B's sums (r >= 8);  group B: b[j + 256 r] = A's differences (r < 8), own (r >= 8)
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const C got = mine[256 * s + j], own = x[s];
            x[8 + s] = sel2(g != 0, own, got);
            x[s] = sel2(g != 0, got, own);
        }
        __syncthreads();
    }
    SF::run_at(x, image, FftRR<4096, false, C>::load_at(a.tw_fwd, j), j);

    // ---- mask: group A holds bins 2 (j + 256 r), group B the odd ones; the mask rows are [even | odd]
    const C *mask = a.mask + (long long)ch * a.mask_stride + 4096 * g;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        x[r] = cmul(x[r], mask[j + 256 * r]);
        if ((r + 1) % QH_MASK_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    SI::run_at(x, image, FftRR<4096, true, C>::load_at(a.tw_inv, j), j);

    // ---- inverse radix-2 step: B turns its half by W^-n, the halves change hands, sums and differences leave
    {
        if (g) {
            C w = a.tw_r2[j];
            w.y = -w.y;                                 // W^-j
            const C wstep = make_double2(0.98078528040323044913, 0.19509032201612826785);
#pragma unroll
            for (int r = 0; r < 16; r++) { x[r] = cmul(x[r], w); if (r < 15) w = cmul(w, wstep); }
        }
        __syncthreads();                                // the inverse transform's last LDS reads are done
#pragma unroll
        for (int s = 0; s < 8; s++) theirs[256 * s + j] = sel2(g != 0, x[s], x[8 + s]);    // B: W^-n B'[j + 256 s];  A: A'[2048 + j + 256 s]
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const C got = mine[256 * s + j];
            // B, n = 2048 + j + 256 s: y[n] = A'[n] + W^-n B'[n] and y[n + 4096] = A'[n] - W^-n B'[n], A' received
            // A, n = j + 256 s: only y[n + 4096] = A'[n] - (received) lies behind the pre-roll
            const C wb = x[8 + s], lo = x[s];
            const C minuend = sel2(g != 0, got, lo), subtrahend = sel2(g != 0, wb, got);
            x[s] = cadd(got, wb);                       // used by group B only
            x[8 + s] = make_double2(minuend.x - subtrahend.x, minuend.y - subtrahend.y);
        }
    }
    if constexpr (METER) {
        __syncthreads();
        int j2 = j;
        asm volatile("" : "+v"(j2));
        double *blk = reinterpret_cast<double *>(image) + (j2 >> 6) * kMeterLdsDoublesPerWave;
        double2 *dst = a.meter_out + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) + (g ? 32 : 0);
        meter_tap<C, 16>(x, g ? 0 : 8, a.meter_w[j2 & 63], blk, dst, j2 >> 6, j2 & 63);
    }

    // ---- epilogue + store: register r of group g is output (r < 8 ? 2048 g : 4096 + 2048 g) + j + 256 (r & 7) - P of the tile
    C *out = a.out + (long long)ch * a.out_stride + a.out_offset;
    EpiParam ep;
    if (a.epi) ep = a.epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r < 8 && g == 0) continue;                  // wave-uniform
        const int rel = (r < 8 ? 0 : 4096) + n0 + 256 * (r & 7) - P;
        const long long m = (long long)tile * L + rel;
        if (m < a.n_out) {
            C v;
            v.x = ep.a * x[r].x + ep.b * x[r].y;
            v.y = ep.c * x[r].x + ep.d * x[r].y;
            if constexpr (EGRESS) egress_store(a.eg, ch, a.out_offset + m, v.x, v.y);
            else out[m] = v;
        }
    }
}

// (-DQH_EXP_BAND8_SEQ: experiment builds, tools/ab_bench.py; QH_BAND8_FORM=seq in the environment then selects it)
#ifdef QH_EXP_BAND8_SEQ
// ---- D = 1 stage on 8192-point tiles, the two halves of the radix-2 split ONE AFTER THE OTHER on 256 lanes ------------------------
// The split of osfir8k_kernel (a = x[n] + x[n + 4096] -> even bins, b = (x[n] - x[n + 4096]) W^n -> odd bins) without its second
// lane group: one 256-lane workgroup -- the register, LDS and occupancy budget of the 4096-point kernel, four workgroups per CU --
// runs the 4096-point transform pair of the even bins, parks A'[n] (n = j + 256 s: every lane parks and later fetches its OWN sixteen
// values, so no fence beyond the lane's own store -> load order is needed), reads the tile's samples a second time (from L2), runs
// the pair of the odd bins and joins: y[n] = A'[n] + W^-n B'[n], y[n + 4096] = A'[n] - W^-n B'[n].  6144 outputs per four 4096-point
// transforms and two butterfly stages instead of 2049 per two: 0.66 of the fp64 instructions per output, HBM bytes per output 0.67.
// A' is parked in the tile's own stretch of the output rows (positions rel 2048 .. 6143, which the tile overwrites with y at its
// end); a tile that the call's end cuts short parks in a per-channel scratch row instead (a.stash).  Masks [even | odd], meter
// partials and tile geometry are osfir8k_kernel's (layout 1), so the engine's band2g plumbing serves both.  Not for EGRESS (the
// parked values would be narrowed): the two-group kernel keeps those calls.
template <bool METER>
__global__ __launch_bounds__(NT, 4) void osfir8s_kernel(OsfirArgs<double> a)
{
    using C = double2;
    using SF = FftSplit4096<false, C>;
    using SI = FftSplit4096<true, C>;
    constexpr int N = 8192, P = kOsfir8kP, L = kOsfir8kLout;
    static_assert(NT / 64 * kMeterLdsDoublesPerWave * 8 <= SF::kLdsBytes, "meter blocks overlay the exchange image");
    extern __shared__ __align__(16) unsigned char smem8s[];
    const int j_ = threadIdx.x;
    int tile, slot;
    xcd_tile_map(a.ntiles, slot, tile);
    // (the division of xcd_tile_map runs on the vector unit: said to be uniform, the tile and the channel -- and every row pointer formed
    // from them -- live in scalar registers)
    tile = __builtin_amdgcn_readfirstlane(tile);
    const int ch = __builtin_amdgcn_readfirstlane(a.chan_list ? a.chan_list[slot] : slot);
    const C *in = a.in + (long long)ch * a.in_stride;
    const int g0_ = a.off - P + tile * L;
    const bool interior = g0_ >= 0 && g0_ + N <= a.n_in;                  // workgroup-uniform
    C *out = a.out + (long long)ch * a.out_stride + a.out_offset;
    // where A' waits: the tile's own outputs rel 2048 + n (n < 4096) while all of them exist, the channel's scratch row otherwise
    const bool whole = (long long)(tile + 1) * L <= (long long)a.n_out;
    C *park = whole ? out + (long long)tile * L + 2048 : a.stash + (long long)ch * 4096;
    // eight samples tile[n0 + 256 s]: plain loads inside the call's buffer, clamped history-aware ones at its ends (branch free)
    auto load8_plain = [&](C (&v)[8], int g0, int n0) {
        const C *p = in + g0 + n0;
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = p[256 * s];
    };
    auto load8_edge = [&](C (&v)[8], int g0, int n0) {
        const C *hist = a.hist ? a.hist + (long long)ch * a.hist_stride : in;
        const int hlen = a.hist ? a.hist_len : 0, last = a.n_in - 1;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int gi = g0 + n0 + 256 * s;
            const int ii = gi > last ? last : gi, ih = gi + hlen < 0 ? 0 : gi + hlen;
            const bool now = gi >= 0, ok = now ? gi <= last : gi + hlen >= 0;
            // (one address from selected parts: a select between two finished pointers became a branch around every load)
            const unsigned long long base = now ? (unsigned long long)in : (unsigned long long)hist;
            const C w = *reinterpret_cast<const C *>(base + (unsigned long long)(unsigned)(now ? ii : ih) * sizeof(C));
            v[s] = make_double2(ok ? w.x : 0.0, ok ? w.y : 0.0);
        }
    };

#pragma nounroll
    for (int g = 0; g < 2; g++) {                                       // g = 0: sums, even bins; g = 1: differences, odd bins
        C x[16];
        const double sgn = g ? -1.0 : 1.0;
        // (the two phases share this code: what depends on the lane or the tile alone is formed again in each -- hoisted out of the
        // loop, three dozen addresses and flags would be carried through both transform pairs in registers the transforms need)
        int j = j_, g0 = g0_;
        asm volatile("" : "+v"(j));
        asm volatile("" : "+s"(g0));
        // ---- the tile's samples n = j + 256 s and n + 4096, eight at a time; one straight-line copy for the tiles inside the call's
        // buffer and one for those at its ends (a branch around every batch of loads made the register allocator park the batches in
        // scratch memory at the joins)
        auto gather = [&](auto load8) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // meter partials, layout 1 of meter_finish_kernel: 32 slots [wave][8] = chunks 32 + 4 s + wave (the upper samples of s < 8),
                // then 64 slots [wave][16]: chunks 4 k + wave (the lower samples of s = 8 + k) and 64 + 4 k + wave (their upper ones)
                double *blk = reinterpret_cast<double *>(smem8s) + (j >> 6) * kMeterLdsDoublesPerWave;
                double2 *mdst = METER ? a.meter_in + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) : nullptr;
                C hi[8];
                load8(hi, g0, j + 2048 * h + 4096);
                if constexpr (METER) {
                    if (g == 0) {
                        if (h == 0) meter_tap<C, 8>(hi, 0, a.meter_w[j & 63], blk, mdst, j >> 6, j & 63, 8);
                        else meter_tap<C, 8>(hi, 0, a.meter_w[j & 63], blk, mdst + 40, j >> 6, j & 63, 16);
                    }
                }
                C lo[8];
                load8(lo, g0, j + 2048 * h);
                if constexpr (METER) {
                    if (g == 0 && h == 1) meter_tap<C, 8>(lo, 0, a.meter_w[j & 63], blk, mdst + 32, j >> 6, j & 63, 16);
                }
#pragma unroll
                for (int s = 0; s < 8; s++)
                    x[8 * h + s] = make_double2(__builtin_fma(hi[s].x, sgn, lo[s].x), __builtin_fma(hi[s].y, sgn, lo[s].y));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (interior) gather(load8_plain); else gather(load8_edge);
        const C wj = a.tw_r2[j];                                        // W^j, W = exp(-2 pi i / 8192)
        if (g) {                                                        // b[n] = (x[n] - x[n + 4096]) W^n, W^n = W^j exp(-i pi s / 16)
#pragma unroll
            for (int s = 0; s < 16; s++) x[s] = cmul(x[s], wj);
            x[1] = mul_wconst<32, 1, false>(x[1]); x[2] = mul_wconst<32, 2, false>(x[2]); x[3] = mul_wconst<32, 3, false>(x[3]);
            x[4] = mul_wconst<32, 4, false>(x[4]); x[5] = mul_wconst<32, 5, false>(x[5]); x[6] = mul_wconst<32, 6, false>(x[6]);
            x[7] = mul_wconst<32, 7, false>(x[7]); x[8] = mul_wconst<32, 8, false>(x[8]); x[9] = mul_wconst<32, 9, false>(x[9]);
            x[10] = mul_wconst<32, 10, false>(x[10]); x[11] = mul_wconst<32, 11, false>(x[11]); x[12] = mul_wconst<32, 12, false>(x[12]);
            x[13] = mul_wconst<32, 13, false>(x[13]); x[14] = mul_wconst<32, 14, false>(x[14]); x[15] = mul_wconst<32, 15, false>(x[15]);
        }
        if constexpr (METER) __syncthreads();                           // the transform's image overlays the waves' meter blocks
        SF::run(x, smem8s, FftRR<4096, false, C>::load(a.tw_fwd));
        const C *mask = a.mask + (long long)ch * a.mask_stride + 4096 * g;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            x[r] = cmul(x[r], mask[j + 256 * r]);
            if ((r + 1) % QH_MASK_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        SI::run(x, smem8s, FftRR<4096, true, C>::load(a.tw_inv));
        if (g == 0) {
#pragma unroll
            for (int s = 0; s < 16; s++) park[j + 256 * s] = x[s];
            __syncthreads();                                            // everybody has read the image: the next phase's taps overlay it
        } else {
            // W^-n B'[n], then the halves: lower outputs y[n] (n >= 2048: s >= 8), upper outputs y[n + 4096] (every s)
            int j2 = j;
            asm volatile("" : "+v"(j2));                                // (lane-derived values formed again behind the transforms, not carried)
            const C wt = a.tw_r2[j2], wc = make_double2(wt.x, -wt.y);
#pragma unroll
            for (int s = 0; s < 16; s++) x[s] = cmul(x[s], wc);
            x[1] = mul_wconst<32, 1, true>(x[1]); x[2] = mul_wconst<32, 2, true>(x[2]); x[3] = mul_wconst<32, 3, true>(x[3]);
            x[4] = mul_wconst<32, 4, true>(x[4]); x[5] = mul_wconst<32, 5, true>(x[5]); x[6] = mul_wconst<32, 6, true>(x[6]);
            x[7] = mul_wconst<32, 7, true>(x[7]); x[8] = mul_wconst<32, 8, true>(x[8]); x[9] = mul_wconst<32, 9, true>(x[9]);
            x[10] = mul_wconst<32, 10, true>(x[10]); x[11] = mul_wconst<32, 11, true>(x[11]); x[12] = mul_wconst<32, 12, true>(x[12]);
            x[13] = mul_wconst<32, 13, true>(x[13]); x[14] = mul_wconst<32, 14, true>(x[14]); x[15] = mul_wconst<32, 15, true>(x[15]);
            if constexpr (METER) __syncthreads();                       // other waves may still be reading the exchange image
            EpiParam ep;
            if (a.epi) ep = a.epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
            double *blk = reinterpret_cast<double *>(smem8s) + (j2 >> 6) * kMeterLdsDoublesPerWave;
            double2 *mdst = METER ? a.meter_out + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) : nullptr;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                C up[8], dn[8];
#pragma unroll
                for (int s = 0; s < 8; s++) up[s] = park[j2 + 2048 * h + 256 * s];
#pragma unroll
                for (int s = 0; s < 8; s++) { dn[s] = cadd(up[s], x[8 * h + s]); up[s] = csub(up[s], x[8 * h + s]); }
                if constexpr (METER) {
                    if (h == 0) meter_tap<C, 8>(up, 0, a.meter_w[j2 & 63], blk, mdst, j2 >> 6, j2 & 63, 8);
                    else {
                        meter_tap<C, 8>(dn, 0, a.meter_w[j2 & 63], blk, mdst + 32, j2 >> 6, j2 & 63, 16);
                        meter_tap<C, 8>(up, 0, a.meter_w[j2 & 63], blk, mdst + 40, j2 >> 6, j2 & 63, 16);
                    }
                }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int n = j2 + 2048 * h + 256 * s;
                    const long long mu = (long long)tile * L + 2048 + n;        // y[n + 4096]: rel = n + 4096 - P
                    if (mu < a.n_out) out[mu] = make_double2(ep.a * up[s].x + ep.b * up[s].y, ep.c * up[s].x + ep.d * up[s].y);
                    if (h == 1) {                                               // y[n], n >= 2048: rel = n - P
                        const long long md = (long long)tile * L + (n - 2048);
                        if (md < a.n_out) out[md] = make_double2(ep.a * dn[s].x + ep.b * dn[s].y, ep.c * dn[s].x + ep.d * dn[s].y);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}
#endif

// ---- D = 1 stage on 6144-point tiles, 384 lanes ---------------------------------------------------------------------------
// The 16 x 24 x 16 plan of qh_fft.hpp (Fft6144): 16 elements per lane like the 4096-point kernel, 4096 outputs per pair of
// transforms instead of 2049 (P = 2048: whole 64-sample chunks, so the fused meters keep one chunk per wavefront and register).
// Lane t holds tile elements t + 384 r; the new samples begin at element 2048 = register 5 of wavefronts 2 .. 5, register 6 of
// wavefronts 0 and 1 (the meter taps take a per-wavefront first register).  Mask: the 6144-point spectrum in natural order.
constexpr int kOsfir6kThreads = kFft6kThreads, kOsfir6kP = 2048, kOsfir6kLout = 4096, kOsfir6kN = 6144;
constexpr int osfir6k_lds_bytes()
{
    constexpr int m = kOsfir6kThreads / 64 * kMeterLdsDoublesPerWave * 8;
    return kFft6kLdsBytes > m ? kFft6kLdsBytes : m;
}

template <bool METER, bool EGRESS>
__global__ __launch_bounds__(kOsfir6kThreads, 3) void osfir6k_kernel(OsfirArgs<double> a)
{
    using C = double2;
    constexpr int N = kOsfir6kN, P = kOsfir6kP, L = kOsfir6kLout, TH = kOsfir6kThreads;
    extern __shared__ __align__(16) unsigned char smem6k[];
    const int t = threadIdx.x;
    int tile, slot;
    xcd_tile_map(a.ntiles, slot, tile);
    const int ch = a.chan_list ? a.chan_list[slot] : slot;
    const C *in = a.in + (long long)ch * a.in_stride;
    const C *hist = a.hist ? a.hist + (long long)ch * a.hist_stride : nullptr;
    const int g0 = a.off - P + tile * L;

    C x[16];
    if (g0 >= 0 && g0 + N <= a.n_in) {                  // workgroup-uniform
        const C *p = in + g0 + t;
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = p[TH * r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gi = g0 + t + TH * r;
            C v = make_double2(0.0, 0.0);
            if (gi >= 0) { if (gi < a.n_in) v = in[gi]; }
            else if (hist && gi + a.hist_len >= 0) v = hist[gi + a.hist_len];
            x[r] = v;
        }
    }
    // meter taps: wavefront w's chunks are elements 64 w + 384 r; new from register 5 (w >= 2) or 6 (w < 2).  A tile's 64
    // partials are stored [w = 0: 10][w = 1: 10][w = 2: 11] .. [w = 5: 11] (meter_finish_kernel, layout 2)
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r0 = wv < 2 ? 6 : 5, moff = wv < 2 ? 10 * wv : 20 + 11 * (wv - 2);
    if constexpr (METER) {
        double *blk = reinterpret_cast<double *>(smem6k) + wv * kMeterLdsDoublesPerWave;
        meter_tap<C, 16>(x, r0, a.meter_w[t & 63], blk, a.meter_in + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) + moff, 0, t & 63);
        __syncthreads();
    }
    Fft6144<false>::run(x, smem6k, Fft6144<false>::load(t), t);

    const C *mask = a.mask + (long long)ch * a.mask_stride;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        x[r] = cmul(x[r], mask[t + TH * r]);
        if ((r + 1) % QH_MASK_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    Fft6144<true>::run(x, smem6k, Fft6144<true>::load(t), t);
    if constexpr (METER) {
        __syncthreads();
        int t2 = t;
        asm volatile("" : "+v"(t2));
        double *blk = reinterpret_cast<double *>(smem6k) + wv * kMeterLdsDoublesPerWave;
        meter_tap<C, 16>(x, r0, a.meter_w[t2 & 63], blk, a.meter_out + (long long)ch * a.meter_stride + (long long)tile * (L >> 6) + moff, 0, t2 & 63);
    }

    C *out = a.out + (long long)ch * a.out_stride + a.out_offset;
    EpiParam ep;
    if (a.epi) ep = a.epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
#pragma unroll
    for (int r = 5; r < 16; r++) {
        const int rel = t + TH * r - P;
        const long long m = (long long)tile * L + rel;
        if (rel >= 0 && m < a.n_out) {
            C v;
            v.x = ep.a * x[r].x + ep.b * x[r].y;
            v.y = ep.c * x[r].x + ep.d * x[r].y;
            if constexpr (EGRESS) egress_store(a.eg, ch, a.out_offset + m, v.x, v.y);
            else out[m] = v;
        }
    }
}

// Interpolating overlap-save FIR:  y[n] = sum_m h[m] * u[n - m],  u[U*i] = x[i], zero elsewhere  -- the audio-rate
// interpolators quisk_dInterpolate / quisk_cInterpolate / quisk_*Interp2HB45 (filter.c:131-201,420-488; the gain
// factor `interp` is folded into the taps).  Zero stuffing replicates the spectrum, so the tile takes an FFT of
// NFFT/U low-rate samples, reads it U times against the NFFT-point mask (the replicas of bin k all live in the lane
// that holds k) and inverse transforms at NFFT points.  Args: Lout / P / n_out in HIGH-rate samples (both multiples
// of U), n_in / hist in low-rate samples; off unused.
// PAIR: as in osfir_kernel -- two channels' real signals in the two parts of one tile (the interpolators' taps are real).
template <typename T, int NFFT, int U, bool PAIR = false>
__global__ __launch_bounds__(NT) void osfir_interp_kernel(OsfirArgs<T> a)
{
    using C = cplx<T>;
    constexpr int NF = NFFT / U;            // forward size
    constexpr int EF = NF / NT, E = NFFT / NT;
    static_assert(NF >= 2 * NT && U > 1, "NFFT/U must be >= 512");
    using Fwd = TileFft<NF, false, C>;
    using Inv = TileFft<NFFT, true, C>;
    extern __shared__ __align__(16) unsigned char smem[];
    void *lds = smem;

    const int t = threadIdx.x;
    int tile, slot;
    xcd_tile_map(a.ntiles, slot, tile);
    const int ch = PAIR ? a.chan_list[2 * slot] : a.chan_list ? a.chan_list[slot] : slot;
    const int ch_b = PAIR ? a.chan_list[2 * slot + 1] : ch;
    const C *in = a.in + (long long)ch * a.in_stride;
    const C *hist = a.hist ? a.hist + (long long)ch * a.hist_stride : nullptr;
    const int g0 = tile * (a.Lout / U) - a.P / U;           // low-rate index of tile element 0

    C x[EF];
    const unsigned ok = load_tile<T, EF>(x, in, hist, a.hist_len, a.n_in, g0);
#pragma unroll
    for (int r = 0; r < EF; r++)
        if (!((ok >> r) & 1u)) x[r] = mk<T>(0, 0);
    if constexpr (PAIR) {
        C xb[EF];
        const unsigned okb = load_tile<T, EF>(xb, a.in + (long long)ch_b * a.in_stride,
                                              a.hist ? a.hist + (long long)ch_b * a.hist_stride : nullptr, a.hist_len, a.n_in, g0);
#pragma unroll
        for (int r = 0; r < EF; r++) x[r].y = ((okb >> r) & 1u) ? xb[r].x : (T)0;
    }

    Fwd::run(x, lds, Fwd::load(a.tw_fwd));

    const C *mask = a.mask + (long long)ch * a.mask_stride;
    C z[E];
#pragma unroll
    for (int i = 0; i < E; i++) z[i] = cmul(x[i % EF], mask[t + NT * i]);

    __syncthreads();
    Inv::run(z, lds, Inv::load(a.tw_inv));

    C *out = a.out + (long long)ch * a.out_stride + a.out_offset;
    EpiParam ep;
    if (a.epi) ep = a.epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
    if constexpr (PAIR) {
        C *out_b = a.out + (long long)ch_b * a.out_stride + a.out_offset;
        EpiParam eb = ep;
        if (a.epi) eb = a.epi[ch_b];
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int rel = t + NT * i - a.P;
            const long long m = (long long)tile * a.Lout + rel;
            if (rel >= 0 && rel < a.Lout && m < a.n_out) {
                const T ya = z[i].x, yb = z[i].y;
                const T ia = a.pair_im0 ? (T)0 : ya, ib = a.pair_im0 ? (T)0 : yb;
                out[m] = mk<T>((T)ep.a * ya + (T)ep.b * ia, (T)ep.c * ya + (T)ep.d * ia);
                out_b[m] = mk<T>((T)eb.a * yb + (T)eb.b * ib, (T)eb.c * yb + (T)eb.d * ib);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int rel = t + NT * i - a.P;
            const long long m = (long long)tile * a.Lout + rel;
            if (rel >= 0 && rel < a.Lout && m < a.n_out) {
                C v;
                v.x = (T)ep.a * z[i].x + (T)ep.b * z[i].y;
                v.y = (T)ep.c * z[i].x + (T)ep.d * z[i].y;
                out[m] = v;
            }
        }
    }
}

}  // namespace qh
