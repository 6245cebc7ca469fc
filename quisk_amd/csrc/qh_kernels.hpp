// qh_kernels.hpp -- small bookkeeping kernels around qh::osfir_kernel.
#pragma once
#include "qh_osfir.hpp"

namespace qh {

// Carry a stage's input history to the next call (the reference keeps it in the resampler ring,
// wdsp/resample.c:133-134, and in fircore's fftin/fftout delay line, wdsp/firmin.c:412-429):
//   new_hist[j] <- sample at stream index  n_in - H + j   (j = 0..H-1; index -1 is the newest old sample)
// taken from `in` (mixed with the NCO when MIX, so that history is stored already rotated) or, for
// negative indices, from old_hist.  Ping-pong buffers, so reads never race the writes.
template <typename T, bool MIX, bool PACKED = false>
__global__ __launch_bounds__(NT) void hist_update_kernel(const cplx<T> *in, long long in_stride, int n_in,
                                                         const cplx<T> *old_hist, cplx<T> *new_hist, int H,
                                                         const unsigned long long *nco_phase,
                                                         const unsigned long long *nco_dphase,
                                                         const int *chan_list = nullptr,
                                                         const unsigned char *pk_src = nullptr, PackedFmt pk = PackedFmt{})
{
    using C = cplx<T>;
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= H) return;
    const long long g = (long long)n_in - H + j;
    C v;
    if (g >= 0) {
        if constexpr (PACKED) v = decode_packed<T>(pk_src, pk, ch, g);
        else v = in[(long long)ch * in_stride + g];
        if constexpr (MIX) {
            unsigned long long ph = nco_phase[ch] + nco_dphase[ch] * (unsigned long long)g;
            C rot;
            sincos_turns<T>(ph, rot.x, rot.y);
            v = cmul(v, rot);
        }
    } else {
        v = old_hist[(long long)ch * H + (H + g)];
    }
    new_hist[(long long)ch * H + j] = v;
}

// phase[ch] += dphase[ch] * n   (wdsp/shift.c:77-79 accumulates the same quantity in radians)
[[maybe_unused]] static __global__ void nco_advance_kernel(unsigned long long *phase, const unsigned long long *dphase, int nch, long long n)
{
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < nch) phase[ch] += dphase[ch] * (unsigned long long)n;
}

// xshift with run = 0 passes the samples through and leaves the phase where it is (wdsp/shift.c:60-85): park the phase
// of channel `ch` while the shift is off (phase 0, step 0 = no rotation) and bring it back when it is switched on.
[[maybe_unused]] static __global__ void nco_park_kernel(unsigned long long *phase, unsigned long long *parked, int ch, int run)
{
    if (run) phase[ch] = parked[ch];
    else { parked[ch] = phase[ch]; phase[ch] = 0; }
}

// ---- output-side NCO (osfir_kernel OUTMIX) ---------------------------------------------------------------------------
// Phasor of the oscillator at the first input index of every tile of the coming launch: g0(tile) = g00 + tile * gstep.
[[maybe_unused]] static __global__ void nco_tile_kernel(const unsigned long long *phase, const unsigned long long *dphase, double2 *tile_rot,
                                                        int ntiles, long long g00, long long gstep)
{
    const int tile = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (tile >= ntiles) return;
    const long long g0 = g00 + (long long)tile * gstep;
    double2 r;
    sincos_turns<double>(phase[ch] + dphase[ch] * (unsigned long long)g0, r.x, r.y);
    tile_rot[(long long)ch * ntiles + tile] = r;
}

// A channel's oscillator changes (frequency, run / parked phase) between two calls: the stored raw history x[n'], n' < 0,
// was going to be seen through the old phase law phi_old(n') = p_old + d_old n'; the kernel will apply the new one.
// x[n'] <- x[n'] exp(j (phi_old(n') - phi_new(n'))) keeps every product h[k] x[n'] exp(j phi(n')) what the reference's
// sample-by-sample xshift (wdsp/shift.c:60-85) made it.  Reads the old law from the device arrays (the update follows in
// stream order); list[] = channel, new_law[] = (action, d_new) per listed channel.
[[maybe_unused]] static __global__ void nco_retune_hist_kernel(double2 *hist, int H, const unsigned long long *phase,
                                                               const unsigned long long *dphase, const unsigned long long *parked,
                                                               const int *list, const unsigned long long *new_law)
{
    const int ch = list[blockIdx.y], j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    const long long np = (long long)j - H;                  // n' of history entry j
    // new_law: (action, d_new) -- action 0: the phase carries on, 1: the shift is switched off (phase parked, no rotation
    // from now on), 2: switched on again (the parked phase comes back)
    const unsigned long long act = new_law[2 * blockIdx.y], d_new = new_law[2 * blockIdx.y + 1];
    const unsigned long long p_new = act == 0 ? phase[ch] : act == 1 ? 0ull : parked[ch];
    const unsigned long long dphi = (phase[ch] - p_new) + (dphase[ch] - d_new) * (unsigned long long)np;
    double2 r;
    sincos_turns<double>(dphi, r.x, r.y);
    double2 *p = hist + (long long)ch * H + j;
    *p = cmul(*p, r);
}

// The front stage's per-channel tables for the output-side oscillator: mask = FFT_NFFT(h[k] exp(-j k delta)) / NFFT (the
// resampler's real taps of wdsp/resample.c:35-78 modulated down by the channel's shift), lane_rot[t] = exp(j D delta t),
// step = exp(j D NT delta).  One workgroup per listed channel; delta = dphase[ch] in 2^-64 turns, every angle exact in
// 64-bit wrap-around arithmetic before it is converted.
template <int NFFT>
__global__ __launch_bounds__(NT) void front_mask_kernel(const double *taps, int ntaps, const unsigned long long *dphase, const int *list,
                                                        int D, const double2 *tw, double2 *mask, double2 *lane_rot, double2 *step, int poly)
{
    constexpr int E = NFFT / NT;
    using Fwd = TileFft<NFFT, false, double2>;
    extern __shared__ __align__(16) unsigned char smem_mask[];
    const int ch = list[blockIdx.x], t = threadIdx.x;
    const unsigned long long d = dphase[ch];
    double2 x[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int k = t + NT * r;
        x[r] = make_double2(0.0, 0.0);
        if (k < ntaps) {
            double c, s;
            sincos_turns<double>(0ull - d * (unsigned long long)k, c, s);
            const double h = taps[k];
            x[r] = make_double2(h * c, h * s);
        }
    }
    Fwd::run(x, smem_mask, Fwd::load(tw));
    if (poly) {
        // polyphase form for the stage's shortened transform (FftSplit4096::run_poly): G_a[k] = W_NFFT^(a k) sum_q W_D^(a q) M[k + (NFFT/D) q],
        // k = t + NT i < NFFT / D; the lane holds M[t + NT (i + EO q)] in x[i + EO q]; G_a is stored where the fold reads its q = a
        const int EO = E / D;
        for (int i = 0; i < EO; i++)
            for (int a = 0; a < D; a++) {
                double2 acc = make_double2(0.0, 0.0);
                for (int q = 0; q < D; q++) {
                    double c, s;
                    sincospi(-2.0 * (double)((a * q) % D) / (double)D, &s, &c);
                    const double2 m = x[i + EO * q];
                    acc.x += m.x * c - m.y * s; acc.y += m.x * s + m.y * c;
                }
                double c, s;
                sincospi(-2.0 * (double)(((long long)a * (t + NT * i)) % NFFT) / (double)NFFT, &s, &c);
                mask[(long long)ch * NFFT + t + NT * (i + EO * a)] =
                    make_double2((acc.x * c - acc.y * s) * (1.0 / NFFT), (acc.x * s + acc.y * c) * (1.0 / NFFT));
            }
    } else {
#pragma unroll
        for (int r = 0; r < E; r++) mask[(long long)ch * NFFT + t + NT * r] = make_double2(x[r].x * (1.0 / NFFT), x[r].y * (1.0 / NFFT));
    }
    double2 lr;
    sincos_turns<double>(d * (unsigned long long)((long long)D * t), lr.x, lr.y);
    lane_rot[(long long)ch * NT + t] = lr;
    if (t == 0) {
        double2 st;
        sincos_turns<double>(d * (unsigned long long)((long long)D * NT), st.x, st.y);
        step[ch] = st;
    }
}

// Elementwise stage used when a chain has no FIR stage to fuse into:
//   out = epi * (in * nco)      (xshift, wdsp/shift.c:60-85; xwcpagc mode 0 + xpanel)
template <typename T, bool MIX, bool EGRESS = false>
__global__ __launch_bounds__(NT) void pointwise_kernel(const cplx<T> *in, long long in_stride, cplx<T> *out,
                                                       long long out_stride, int n,
                                                       const unsigned long long *nco_phase,
                                                       const unsigned long long *nco_dphase,
                                                       const EpiParam *epi, const int *chan_list = nullptr, EgressFmt eg = EgressFmt{})
{
    using C = cplx<T>;
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    EpiParam ep;
    if (epi) ep = epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
    for (long long g = (long long)blockIdx.x * NT + threadIdx.x; g < n; g += (long long)gridDim.x * NT) {
        C v = in[(long long)ch * in_stride + g];
        if constexpr (MIX) {
            unsigned long long ph = nco_phase[ch] + nco_dphase[ch] * (unsigned long long)g;
            C rot;
            sincos_turns<T>(ph, rot.x, rot.y);
            v = cmul(v, rot);
        }
        C o;
        o.x = (T)ep.a * v.x + (T)ep.b * v.y;
        o.y = (T)ep.c * v.x + (T)ep.d * v.y;
        if constexpr (EGRESS) egress_store(eg, ch, g, (double)o.x, (double)o.y);
        else out[(long long)ch * out_stride + g] = o;
    }
}

}  // namespace qh
