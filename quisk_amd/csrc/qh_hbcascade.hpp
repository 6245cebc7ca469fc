// qh_hbcascade.hpp -- fused cascade of NS Quisk 45-tap half-band decimators (quisk_cDecim2HB45, filter.c:377-417),
// total decimation 2^NS, time domain, one HBM pass: 8 (fp32) or 16 (fp64) bytes read per input sample and
// 1/2^NS of that written.  This is BASELINE config 5's front end (8 stages, 61.44 Msps -> 240 ksps) and the
// HB45 runs of quisk_process_decimate (quisk.c:1772-1796).
//
// A half-band output needs only odd-indexed inputs for its 22 symmetric taps and one even-indexed input for the
// centre tap:   v[m] = sum_{i<11} c_i (uo[m-i] + uo[m-21+i]) + 0.5 ue[m-10],   uo[i] = u[2i+1], ue[i] = u[2i],
// so every stage keeps two LDS rings (odd: 21 samples of history, even: 10).  A lane computes R consecutive
// outputs (4 in the first two stages, then 2) from a register window of (21+R) samples read as PAIRS -- 15
// 128-bit LDS reads per 4 outputs where one output alone needs 23 64-bit ones; LDS bandwidth, not HBM, is what
// a naive version of this kernel is bound by.  The rings are stored transposed, pair J at row J%(R/2), column
// J/(R/2), so lane u's window pair j sits at row j%(R/2), column u + j/(R/2): consecutive lanes read consecutive
// addresses with a compile-time offset.  Stages with fewer lanes than the workgroup are dealt round the four
// waves (HbGeom::lane0) so that no SIMD carries all of them.
//
// A workgroup owns a contiguous time segment of one channel and walks it in steps of 2048 input samples, all
// stages per step, so stage histories never leave LDS; the next step's input is prefetched into registers while
// the stages run.  Segments start with a warm-up of ceil(42 (2^NS - 1) / 2048) steps over the preceding input
// (the previous call's tail for the first segment) whose outputs are discarded -- FIR state is nothing but
// input history.
#pragma once
#include "qh_fft.hpp"

namespace qh {

#ifdef QH_HBC_PROBE      // tools/ubench/hbc_phase.hip only: per-phase shader-clock stamps of workgroup 0, lane 0
__device__ long long g_hbc_probe[64 * 16];
__device__ int g_hbc_probe_step;
#define QH_PROBE(slot) do { if (probe_on && threadIdx.x == 0) g_hbc_probe[probe_row * 16 + (slot)] = clock64(); } while (0)
#else
#define QH_PROBE(slot) do { } while (0)
#endif

template <typename T> struct alignas(2 * sizeof(cplx<T>)) HbPair { cplx<T> e[2]; };

// A pair out of LDS as 128-bit accesses: ds_read_b128 runs at 256 B/clk per CU, the ds_read2_b64 the compiler picks for a pair whose
// halves are used apart at 128 B/clk -- and with two-way bank conflicts at a lane pitch of 16 bytes (MI355X_MICROARCH.md, LDS).
typedef float hb_v4f __attribute__((ext_vector_type(4)));
typedef double hb_v2d __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ HbPair<T> hb_load_pair(const HbPair<T> *p)
{
    HbPair<T> r;
    if constexpr (sizeof(T) == 4) {
        const hb_v4f v = *reinterpret_cast<const hb_v4f *>(p);
        r.e[0].x = v.x; r.e[0].y = v.y; r.e[1].x = v.z; r.e[1].y = v.w;
    } else {
        const hb_v2d a = reinterpret_cast<const hb_v2d *>(p)[0], b = reinterpret_cast<const hb_v2d *>(p)[1];
        r.e[0].x = a.x; r.e[0].y = a.y; r.e[1].x = b.x; r.e[1].y = b.y;
    }
    return r;
}

// STEP_ / R0_: input samples per step and outputs per lane in the FIRST stage.  2048 / 4 is the form of rounds 2 - 5; 4096 / 8 (fp32 only:
// 63 KB of rings for four stages, two workgroups per CU) reads 19 window pairs per 8 outputs in the first stage instead of 15 per 4 and keeps
// twice the bytes in flight per workgroup (profiles/r06_notes.md).
template <int NS, int STEP_ = 2048, int R0_ = 4> struct HbGeom {
    static constexpr int STEP = STEP_;
    static constexpr int WSTEPS = (42 * ((1 << NS) - 1) + STEP - 1) / STEP;     // warm-up steps
    static constexpr int WARM = WSTEPS * STEP;                                  // input samples of history kept per channel
    // stage s: n(s) outputs per step (= new samples per ring per step), R(s) outputs per lane, lanes(s) lanes
    // starting at lane0(s) -- the stages with fewer lanes than the workgroup are dealt round the four waves so
    // that the packed-FMA work per wave is level (w0: S0+S1+S7, w1: S0+S1, w2: S0+S2+S3+S5, w3: S0+S2+S4+S6).
    static constexpr int n(int s) { return STEP >> (s + 1); }
    static constexpr int R(int s) { return s == 0 ? R0_ : s == 1 ? 4 : 2; }
    static constexpr int lanes(int s) { return n(s) / R(s); }
    static constexpr int lane0(int s)
    {
        if (STEP_ == 2048) return s < 2 ? 0 : s == 2 || s == 3 || s == 5 ? 128 : s == 4 || s == 6 ? 192 : 0;
        // 4096 / 8: stages 0 - 2 take every lane; 3 (128 lanes) waves 2 - 3, 4 (64) wave 0, 5 (32) wave 1, the rest wave 0
        return s < 3 ? 0 : s == 3 ? 128 : s == 5 ? 64 : 0;
    }
    // Rings hold PAIRS of consecutive samples (one 128-bit LDS access for fp32).  Odd ring: logical index HO + i
    // for new sample i, history below it; even ring: HE + i.  Pair J = L/2 lives at row J % RP, column J / RP,
    // RP = R/2, so lane u's window pair j sits at row j % RP, column u + j / RP.  PP = pairs per row, chosen so
    // that the two rows of an R = 4 ring start half a bank sweep apart.
    static constexpr int HO(int s) { return R(s) >= 4 ? 24 : 22; }
    static constexpr int HE(int s) { return R(s) >= 4 ? 12 : 10; }
    static constexpr int RP(int s) { return R(s) / 2; }
    static constexpr int PP(int s)
    {
        int p = ((HO(s) + n(s)) / 2 + RP(s) - 1) / RP(s);
        if (RP(s) > 1) while (p % 16 != 16 / RP(s)) p++;       // the RP rows start 64 / RP banks apart
        return p;
    }
    static constexpr int odd_off(int s)              // in pairs
    {
        int o = 0;
        for (int k = 0; k < s; k++) o += 2 * RP(k) * PP(k);
        return o;
    }
    static constexpr int even_off(int s) { return odd_off(s) + RP(s) * PP(s); }
    static constexpr int ring_pairs() { return odd_off(NS); }
    static constexpr int carry_pairs()
    {
        int c = 0;
        for (int k = 0; k < NS; k++) c += (HO(k) + HE(k)) / 2;
        return c;
    }
};

template <typename T, int NS, int S, typename G> struct HbStage {
    using C = cplx<T>;
    using PR = HbPair<T>;
    // One stage over one step: lane u = t - lane0 (0 <= u < lanes) produces outputs R u .. R u + R - 1.
    static __device__ __forceinline__ void run(PR *lds, int t, bool store, C *y, long long obase, long long olimit)
    {
        constexpr int R = G::R(S), RP = G::RP(S), PP = G::PP(S), HO = G::HO(S), HE = G::HE(S), K0 = HO - 21;
        constexpr int J0 = K0 / 2, J1 = (HO + R - 1) / 2;          // window pairs J0..J1 hold logical R u + 2 J0 .. R u + 2 J1 + 1
        const T cc[11] = { (T)0.000018566625444266, (T)-0.000118469698701817, (T)0.000457318798253456,
                           (T)-0.001347840471412094, (T)0.003321838571445455, (T)-0.007198422696929033,
                           (T)0.014211106939802483, (T)-0.026424776824073383, (T)0.048414810444971007,
                           (T)-0.096214669073304823, (T)0.314881034738348550 };         // filter.c:382-385
        const int u = t - G::lane0(S);
        if (u >= 0 && u < G::lanes(S)) {
            const PR *uo = lds + G::odd_off(S) + u, *ue = lds + G::even_off(S) + u;
            PR w[J1 - J0 + 1];
#pragma unroll
            for (int j = J0; j <= J1; j++) w[j - J0] = hb_load_pair(uo + (j % RP) * PP + j / RP);
            constexpr int JE0 = (HE - 10) / 2, JE1 = (HE - 10 + R - 1) / 2;      // even ring, logical HE + m - 10
            PR ce[JE1 - JE0 + 1];
#pragma unroll
            for (int j = JE0; j <= JE1; j++) ce[j - JE0] = hb_load_pair(ue + (j % RP) * PP + j / RP);
            C acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int ke = HE - 10 + r;
                const C ctr = ce[ke / 2 - JE0].e[ke % 2];
                acc[r].x = (T)0.5 * ctr.x; acc[r].y = (T)0.5 * ctr.y;
            }
            // taps outside, outputs inside: the R accumulation chains advance side by side (same order of additions per output)
#pragma unroll
            for (int i = 0; i < 11; i++) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int ka = HO + r - i, kb = K0 + r + i;
                    const C a = w[ka / 2 - J0].e[ka % 2], b = w[kb / 2 - J0].e[kb % 2];
                    acc[r].x += cc[i] * (a.x + b.x);
                    acc[r].y += cc[i] * (a.y + b.y);
                }
            }
            if constexpr (S + 1 < NS) {
                // output m = R u + r is sample m >> 1 of the next stage's even (m even) / odd (m odd) ring
                constexpr int RP2 = G::RP(S + 1), PP2 = G::PP(S + 1), HO2 = G::HO(S + 1), HE2 = G::HE(S + 1);
                PR *wo = lds + G::odd_off(S + 1), *we = lds + G::even_off(S + 1);
                if constexpr (R == 8) {                       // samples 4u .. 4u+3 of each ring: two aligned pairs
                    const int Je = HE2 / 2 + 2 * u, Jo = HO2 / 2 + 2 * u;
                    PR pe, po;
                    pe.e[0] = acc[0]; pe.e[1] = acc[2]; po.e[0] = acc[1]; po.e[1] = acc[3];
                    we[(Je % RP2) * PP2 + Je / RP2] = pe;
                    wo[(Jo % RP2) * PP2 + Jo / RP2] = po;
                    pe.e[0] = acc[4]; pe.e[1] = acc[6]; po.e[0] = acc[5]; po.e[1] = acc[7];
                    we[((Je + 1) % RP2) * PP2 + (Je + 1) / RP2] = pe;
                    wo[((Jo + 1) % RP2) * PP2 + (Jo + 1) / RP2] = po;
                } else if constexpr (R == 4) {                // samples 2u, 2u+1 of each ring: one aligned pair
                    const int Je = HE2 / 2 + u, Jo = HO2 / 2 + u;
                    PR pe, po;
                    pe.e[0] = acc[0]; pe.e[1] = acc[2]; po.e[0] = acc[1]; po.e[1] = acc[3];
                    we[(Je % RP2) * PP2 + Je / RP2] = pe;
                    wo[(Jo % RP2) * PP2 + Jo / RP2] = po;
                } else {                                      // sample u of each ring
                    const int Le = HE2 + u, Lo = HO2 + u;
                    we[((Le >> 1) % RP2) * PP2 + (Le >> 1) / RP2].e[Le & 1] = acc[0];
                    wo[((Lo >> 1) % RP2) * PP2 + (Lo >> 1) / RP2].e[Lo & 1] = acc[1];
                }
            } else if (store) {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (obase + R * u + r < olimit) y[obase + R * u + r] = acc[r];
            }
        }
    }
};

template <typename T, int NS, int S, typename G> struct HbStages {
    static __device__ __forceinline__ void run(HbPair<T> *lds, int t, bool store, cplx<T> *y, long long obase, long long olimit,
                                               bool probe_on = false, int probe_row = 0)
    {
        HbStage<T, NS, S, G>::run(lds, t, store, y, obase, olimit);
        __syncthreads();
        QH_PROBE(2 + S);
        if constexpr (S + 1 < NS) HbStages<T, NS, S + 1, G>::run(lds, t, store, y, obase, olimit, probe_on, probe_row);
    }
};

#ifndef QH_HBC_WAVES_F32
#define QH_HBC_WAVES_F32 3
#endif
// BIG: the 4096 / 8 geometry (fp32).  hlen: samples of history per channel row (at least G::WARM, the same for both geometries of a handle).
template <typename T, int NS, bool BIG = false>
__global__ __launch_bounds__(NT, BIG ? 2 : sizeof(T) == 4 ? QH_HBC_WAVES_F32 : 2) void hb45_cascade_kernel(const cplx<T> *in, long long in_stride, const cplx<T> *hist, int n_in,
                                                          cplx<T> *out, long long out_stride, int seg, cplx<T> *hist_next, int hlen)
{
    using C = cplx<T>;
    using PR = HbPair<T>;
    using G = HbGeom<NS, BIG ? 4096 : 2048, BIG ? 8 : 4>;
    constexpr int STEP = G::STEP, NQ = STEP / 4 / NT;
    extern __shared__ __align__(32) unsigned char smem[];
    PR *lds = reinterpret_cast<PR *>(smem);

    const int t = threadIdx.x;
    const int ch = blockIdx.y;
    const long long start = (long long)blockIdx.x * seg;
    if (start >= n_in) return;
    const int nsteps = (int)((((long long)n_in - start < seg ? (long long)n_in - start : (long long)seg) + STEP - 1) / STEP);
    const C *x = in + (long long)ch * in_stride;
    const C *h = hist + (long long)ch * hlen + hlen;            // h[-k] = the k-th sample before in[0]
    C *y = out + (long long)ch * out_stride;

    // the history the next call finds (hb45_hist_kernel's job, a launch of its own): the segments that hold the call's last WARM samples
    // copy their share, ahead of everything else (the host asks for this when the call is at least hlen samples long)
    if (hist_next && start + seg > (long long)n_in - hlen) {
        const long long first = (long long)n_in - hlen;
        const long long lo = start > first ? start : first, hi = start + seg < n_in ? start + seg : (long long)n_in;
        C *hn = hist_next + (long long)ch * hlen;
        for (long long g = lo + t; g < hi; g += NT) hn[g - first] = x[g];
    }

    {
        PR z;
        z.e[0] = mk<T>(0, 0); z.e[1] = z.e[0];
        for (int i = t; i < G::ring_pairs(); i += NT) lds[i] = z;
    }

    // history carry: the last HO/2 odd and HE/2 even pairs of ring s move down by n(s)/2 pairs = n(s)/R(s) columns
    static_assert(G::carry_pairs() <= NT, "one carried pair per lane");
    int cdst = -1, csrc = 0;
    {
        int id = t;
        for (int s = 0; s < NS; s++) {
            const int rp = G::RP(s), pp = G::PP(s), ho = G::HO(s) / 2, he = G::HE(s) / 2;
            if (id >= 0 && id < ho + he) {
                const int j = id < ho ? id : id - ho;
                cdst = (id < ho ? G::odd_off(s) : G::even_off(s)) + (j % rp) * pp + j / rp;
                csrc = cdst + G::n(s) / G::R(s);
            }
            id -= ho + he;
        }
    }

    // prefetch registers: raw loads only -- nothing may consume them before the next step's ring fill, or the
    // compiler parks a vmcnt(0) wait in front of the stage arithmetic.  A lane takes 4 consecutive samples.
    C pf[NQ][4];
    unsigned okmask = 0;
    auto fetch = [&](long long base) {
        okmask = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const long long g = base + 4 * (t + NT * q);
            const bool ok = g < n_in;                                       // ragged last step (n_in is a multiple of 2^NS only)
            const C *p = g >= 0 ? x + (ok ? g : 0) : h + g;
#pragma unroll
            for (int k = 0; k < 4; k++) pf[q][k] = p[k];
            okmask |= ok ? 1u << q : 0u;
        }
    };
    fetch(start - (long long)G::WSTEPS * STEP);
    __syncthreads();

    for (int step = -G::WSTEPS; step < nsteps; step++) {
        const long long base = start + (long long)step * STEP;
        // ---- stage-0 rings from the prefetched registers: samples 4i..4i+3 -> even pair HE/2 + i, odd pair HO/2 + i
        {
            constexpr int RP = G::RP(0), PP = G::PP(0), HO = G::HO(0), HE = G::HE(0);
            PR *uo = lds + G::odd_off(0), *ue = lds + G::even_off(0);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int i = t + NT * q;
                const bool ok = (okmask >> q) & 1u;
                PR pe, po;
                pe.e[0] = ok ? pf[q][0] : mk<T>(0, 0); po.e[0] = ok ? pf[q][1] : mk<T>(0, 0);
                pe.e[1] = ok ? pf[q][2] : mk<T>(0, 0); po.e[1] = ok ? pf[q][3] : mk<T>(0, 0);
                const int Je = HE / 2 + i, Jo = HO / 2 + i;
                ue[(Je % RP) * PP + Je / RP] = pe;
                uo[(Jo % RP) * PP + Jo / RP] = po;
            }
        }
#ifdef QH_HBC_PROBE
        const bool probe_on = blockIdx.x == 1 && blockIdx.y == 0 && step >= 8 && step < 8 + 60;
        const int probe_row = step - 8;
#else
        constexpr bool probe_on = false;
        constexpr int probe_row = 0;
#endif
        QH_PROBE(0);
        __syncthreads();
        QH_PROBE(1);
        if (step + 1 < nsteps) fetch(base + STEP);
        HbStages<T, NS, 0, G>::run(lds, t, step >= 0, y, base >> NS, (long long)(n_in >> NS), probe_on, probe_row);
        // every lane reads (lanes without a carried pair read pair 0 and drop it): a value defined under a condition on both sides of
        // the barrier went through scratch memory, and the scratch load's vmcnt(0) wait also waited for the prefetch above
        // (and as two scalars-of-complex rather than one aggregate: the aggregate copy was given a stack slot)
        const PR *cs = lds + (cdst >= 0 ? csrc : 0);
        const C carry0 = cs->e[0], carry1 = cs->e[1];
        __syncthreads();
        if (cdst >= 0) { lds[cdst].e[0] = carry0; lds[cdst].e[1] = carry1; }
        QH_PROBE(12);
        // no barrier here: the next step's ring-0 fill touches only the "new" columns, and is followed by one
    }
}

// hist_new[j] = stream sample (n_in - WARM + j) counted from in[0]; negative positions come from the old history.
template <typename T>
__global__ __launch_bounds__(NT) void hb45_hist_kernel(const cplx<T> *in, long long in_stride, int n_in, const cplx<T> *hist_old,
                                                       cplx<T> *hist_new, int warm)
{
    const int ch = blockIdx.y;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= warm) return;
    const long long p = (long long)n_in - warm + j;
    hist_new[(long long)ch * warm + j] = p >= 0 ? in[(long long)ch * in_stride + p] : hist_old[(long long)ch * warm + warm + p];
}

}  // namespace qh
