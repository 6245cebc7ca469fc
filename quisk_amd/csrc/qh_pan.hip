// qh_pan.hip -- batched panadapter (include/quiskhip.h group 5): the spectrum-display path of Quisk,
// FFT ring producer in quisk_process_samples (quisk.c:2454-2475) + get_graph job 1 (quisk.c:5142-5331),
// for `nch` receivers at once and for EVERY completed block (the reference drops blocks when its 4-deep ring
// is full).
//
// fft_size N = R * M with M in {1024, 2048, 4096} (an LDS-resident FFT, qh_fft.hpp) and R in {1, 2, 4}, split by
// decimation in FREQUENCY so that every load is contiguous:
//     X[R k + r] = FFT_M{ W_N^(m r) * sum_q W_R^(q r) x[m + M q] w[m + M q] }[k]
//   pan_spectrum_kernel  one workgroup per (r, block range, channel): per block the R strips x[m + M q] times the
//                     Hanning window (quisk.c:6008) are combined for this r, twiddled, put through FFT-M, and
//                     |X| is added to per-lane accumulators kept in registers over the workgroup's blocks; so is
//                     the RMS S-meter sum over the passband bins (quisk.c:5218-5244).  Each block is read by the R
//                     workgroups of its r values; the re-reads come from L2 / Infinity Cache.
//   pan_reduce_kernel the partial sums of the block ranges are added into fft_avg[(bin + N/2) mod N] in a fixed
//                     order (no atomics: results do not depend on scheduling).
//   pan_graph_kernel  the refresh branch (quisk.c:5279-5327): n-bin box sums per pixel done IN PLACE on fft_avg
//                     in pixel order like the reference (a later pixel can see an earlier pixel's sum when
//                     zoomed in), dB scale, clamp to [-200, 0], S-meter in dB.  Tiny: one thread per channel.
#include <cmath>
#include <cstring>
#include <vector>
#include "qh_design.hpp"
#include "qh_internal.hpp"
#include "qh_fft.hpp"

namespace qh {

struct PanBand { int first, nwhole; double frac; int valid, pad; };     // S-meter passband in bins, quisk.c:5223-5244

// sqrt of a power |X|^2 (an ordinary magnitude or exactly 0; never denormal, infinite or NaN): the hardware's 26-bit
// reciprocal square root and one Newton step in fused arithmetic -- s = x y, s += (y / 2) (x - s s) -- leave an error of a
// few 1e-16 relative; the library sqrt spends three times the instructions on scaling and special cases that cannot occur
__device__ __forceinline__ double sqrt_pow(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    const double r = __builtin_fma(-s, s, x);
    s = __builtin_fma(0.5 * y, r, s);
    return x == 0.0 ? 0.0 : s;
}

#ifndef QH_PAN_WAVES
#define QH_PAN_WAVES 2
#endif
template <int M, int R>
__global__ __launch_bounds__(NT, QH_PAN_WAVES) void pan_spectrum_kernel(const double2 *in, long long in_stride, int nblk, int nsplit,
                                                          const double2 *tw, double *partial, double *partial_m2,
                                                          const PanBand *band, int nch)
{
    using C = double2;
    using F = TileFft<M, false, C>;
    constexpr int E = M / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double wsum[NT / 64];
    // Workgroup ids go round the 8 XCDs (each with its own L2): the R workgroups that read the same blocks are
    // given ids 8 apart, so they land on ONE XCD at about the same time and share its L2 (id = 8 R g + 8 r + x,
    // unit = 8 g + x; units beyond nsplit * nch are padding).
    const int t = threadIdx.x;
    const int id = blockIdx.x, grp = id / (8 * R), r = (id / 8) % R, unit = grp * 8 + id % 8;
    if (unit >= nsplit * nch) return;
    const int split = unit % nsplit, ch = unit / nsplit;
    const int N = M * R;
    const int per = (nblk + nsplit - 1) / nsplit;
    const int b0 = split * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    const typename F::Tw twf = F::load(tw);
    // W_N^(r m), m = t + NT i:  w0 * step^i
    C w0, step;
    sincospi(-2.0 * (double)r * (double)t / (double)N, &w0.y, &w0.x);
    sincospi(-2.0 * (double)r * (double)NT / (double)N, &step.y, &step.x);
    C e0, estep;                                                // exp(-2 pi i m / N) for the window
    sincospi(-2.0 * (double)t / (double)N, &e0.y, &e0.x);
    sincospi(-2.0 * (double)NT / (double)N, &estep.y, &estep.x);
    const double sgn = (r & 1) ? -1.0 : 1.0;
    const C tau = r == 0 ? mk<double>(1, 0) : r == 1 ? mk<double>(0, -1) : r == 2 ? mk<double>(-1, 0) : mk<double>(0, 1);
    // weight of this lane's bins in the S-meter sum
    const PanBand pb = band[ch];
    // the |X| accumulators of this lane's E bins live in LDS behind the FFT image (32 more registers would spill)
    double *acc = reinterpret_cast<double *>(smem + F::kLdsBytes) + t;
    double m2 = 0.0;
    unsigned whole = 0, part = 0;           // bit i: bin i of this lane counts fully / with weight frac
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int bin = R * (t + NT * i) + r;
        const int sb = bin >= N / 2 ? bin - N : bin;            // signed bin
        if (pb.valid) {
            if (sb >= pb.first && sb < pb.first + pb.nwhole) whole |= 1u << i;
            else if (sb == pb.first + pb.nwhole) part |= 1u << i;
        }
        acc[NT * i] = 0.0;
    }
    for (int blk = b0; blk < b1; blk++) {
        const C *x = in + (long long)ch * in_stride + (long long)blk * N;
        asm volatile("" : "+s"(x));         // keeps the R E per-lane load addresses from being precomputed and spilled
        C u[E];
        C w = w0, e = e0;
        // ... and the window / twiddle recurrences from being hoisted out of the block loop as 5 E live values
        asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(w.x), "+v"(w.y));
#pragma unroll
        for (int i = 0; i < E; i++) {
            // Hanning window 0.5 - 0.5 cos(2 pi n / N) (quisk.c:6008) at n = m + M q from e = exp(-2 pi i m / N):
            // cos(a + pi q / 2) and cos(a + pi q) are +-cos a, +-sin a -- no table loads
            double g[4];
            if constexpr (R == 4) {
                g[0] = __builtin_fma(-0.5, e.x, 0.5); g[2] = __builtin_fma(0.5, e.x, 0.5);
                g[1] = __builtin_fma(-0.5, e.y, 0.5); g[3] = __builtin_fma(0.5, e.y, 0.5);
            } else {
                g[0] = __builtin_fma(-0.5, e.x, 0.5); g[1] = __builtin_fma(0.5, e.x, 0.5); g[2] = g[3] = 0.0;
            }
            C v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < R) {
                    const C sx = (x + (NT * i + M * q))[t];         // wave-uniform base + the lane index
                    v[q] = mk<double>(sx.x * g[q], sx.y * g[q]);
                } else {
                    v[q] = mk<double>(0, 0);
                }
            }
            // sum_q W_R^(q r) v_q with wave-uniform factors instead of branches on r:
            //   R = 4:  (v0 + s v2) + tau (v1 + s v3),  s = (-1)^r,  tau = (-i)^r;   R = 2:  v0 + s v1
            C o;
            if constexpr (R == 1) {
                o = v[0];
            } else if constexpr (R == 2) {
                o = mk<double>(__builtin_fma(sgn, v[1].x, v[0].x), __builtin_fma(sgn, v[1].y, v[0].y));
            } else {
                const C a = mk<double>(__builtin_fma(sgn, v[2].x, v[0].x), __builtin_fma(sgn, v[2].y, v[0].y));
                const C c = mk<double>(__builtin_fma(sgn, v[3].x, v[1].x), __builtin_fma(sgn, v[3].y, v[1].y));
                o = mk<double>(a.x + (c.x * tau.x - c.y * tau.y), a.y + (c.x * tau.y + c.y * tau.x));
            }
            u[i] = cmul(o, w);
            w = cmul(w, step);
            e = cmul(e, estep);
            // four elements' worth of loads (16 + 16) in flight at a time: hoisting all 128 costs 400 registers
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        F::run(u, smem, twf);
#pragma unroll
        for (int i = 0; i < E; i++) {
            const double pw2 = u[i].x * u[i].x + u[i].y * u[i].y;
            acc[NT * i] += sqrt(pw2);       // cabs(): no overflow / underflow concern at +-2^31 * N full scale
            m2 += ((whole >> i) & 1u) ? pw2 : (((part >> i) & 1u) ? pb.frac * pw2 : 0.0);
        }
        __syncthreads();                    // the LDS image is free for the next block
    }
    // partial[split][ch][(bin + N/2) mod N]
    double *pp = partial + ((long long)split * nch + ch) * N;
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int bin = R * (t + NT * i) + r;
        pp[(bin + N / 2) % N] = acc[NT * i];
    }
    // S-meter partial: lanes -> wave -> workgroup, fixed order
    for (int d = 32; d > 0; d >>= 1) m2 += __shfl_down(m2, d, 64);
    if ((t & 63) == 0) wsum[t >> 6] = m2;
    __syncthreads();
    if (t == 0) {
        double sm = 0.0;
        for (int k = 0; k < NT / 64; k++) sm += wsum[k];
        partial_m2[((long long)split * nch + ch) * R + r] = sm;
    }
}

// fft_size 16384 (BASELINE config 3), second form: a workgroup of G 256-thread groups (G = 2 or 4) computes G of the four
// r-transforms of the decimation in frequency side by side, and reads the block once for all of them: thread (s, t) loads
// the four strips x[m + 4096 q] for its 16 / G values m = t + 256 (16 s / G + j), windows them, forms the radix-4 combination
// (a 4-point DFT over q) for the workgroup's r values, applies W_N^(m r) -- W_N^m is the phasor the window already uses --
// and the sixteen results change hands inside {(0, t) .. (G - 1, t)} through LDS (real parts, then imaginary parts, laid
// over the G transform images).  Then every group runs the split-exchange FFT-4096 of qh_fft.hpp on its own image, the
// barriers being the workgroup's.  |X| accumulates over the range's blocks in sixteen registers per lane and is stored once, in
// the order [r][bin / 4] (pan_reduce_kernel undoes it).
// G = 2 (72 KB of LDS, two workgroups per CU that cover each other's load phases, the block read twice, the second time
// from L2: the two workgroups of a block sit 8 ids apart = on one XCD) or G = 4 (one workgroup per CU, one read).
// Against pan_spectrum_kernel<4096, 4>: a half / a quarter of the load instructions, 16 wavefronts per CU instead of 8,
// same summation order.
#ifndef QH_PAN16K_GROUPS
#define QH_PAN16K_GROUPS 4
#endif
#ifndef QH_PAN_LOADS_AHEAD
#define QH_PAN_LOADS_AHEAD 1
#endif
#ifndef QH_PAN_PIPE
#define QH_PAN_PIPE 1
#endif
#ifndef QH_PAN_PIPE_LAG
#define QH_PAN_PIPE_LAG 0
#endif
#ifndef QH_PAN_PIPE_AHEAD
#define QH_PAN_PIPE_AHEAD 2
#endif
// -DQH_PAN_TRACE=<thread> (experiment builds, tools/dbg/pan_trace.py): that thread of workgroup 0 leaves the shader clock at the phase
// boundaries of its blocks in qh_pan_trace[block][phase] (pan16k_kernel and panfir16k_kernel: whichever ran last).  (The stamps are stores: a wait that follows them counts their
// acknowledgement too -- the phases up to the transform are trustworthy, the loop edge is not.)
#ifdef QH_PAN_TRACE
__device__ unsigned long long qh_pan_trace[64 * 8];
#define PAN_STAMP(ph) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == QH_PAN_TRACE && blk - b0 < 64) qh_pan_trace[(blk - b0) * 8 + (ph)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PAN_STAMP(ph) do { } while (0)
#endif
// LDS of pan16k_kernel: the G transform images, then what waits there between blocks so that it does not ride through the transform in
// registers: exp(-2 pi i t / N) by t (the window's phasor, and by two squarings the last pass's twiddle), the middle pass's sixteen
// twiddles, and every lane's S-meter sum
template <int G> constexpr int pan16k_lds() { return G * FftSplit4096<false, double2>::kLdsBytes + 256 * 16 + 256 * G * 8; }      // G = 2: 80 KB, two per CU
constexpr int panfir16k_lds() { return pan16k_lds<4>() + 16 * 16; }
// a load from GLOBAL memory at a wave-uniform base plus the lane's 32-bit element offset (a pointer that went through an asm barrier is
// a generic one to the compiler: flat loads, which the LDS counters wait for as well)
__device__ __forceinline__ double2 gload(const double2 *base, unsigned off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const double __attribute__((address_space(1))) *gp;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef const v2d __attribute__((address_space(1))) *gp2;
    const v2d v = *(gp2)(gp)(reinterpret_cast<const double *>(base) + 2 * (size_t)off);
    double2 r; r.x = v.x; r.y = v.y;
    return r;
#else
    return base[off];
#endif
}
// a wave-uniform double as two scalar registers
__device__ __forceinline__ double uniform_f64(double v)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// The register budget (1024 lanes on a CU = 128 registers each) is the kernel: the block's 16384 samples are 64 registers per lane, the
// |X| sums of the workgroup's block range 32 more, and whatever else is live through the transform comes out of the last 32 -- round 5's
// form kept the window's phasor, both passes' twiddles, the S-meter sum and 64-bit load addresses there, the compiler spilled 28
// registers (16 of them |X| sums, stored and fetched again every block: 0.6 bytes of scratch traffic per input byte, profiles/r05_g_c3_pmc.json).
// Here those wait in LDS between their uses (the 16 KB the four images leave) or are formed again from `t`, and the loads take a scalar
// base per strip and one 32-bit lane offset.
template <int G>
__global__ __launch_bounds__(256 * G, 4) void pan16k_kernel(const double2 *in, long long in_stride, int nblk, int nsplit,
                                                         const double2 *tw, double *partial, double *partial_m2,
                                                         const PanBand *band, int nch)
{
    using C = double2;
    using S = FftSplit4096<false, C>;
    constexpr int M = 4096, N = 16384, E = 16, RP = 4 / G, MJ = 16 / G;      // RP workgroups per block, MJ values of m per thread
    extern __shared__ __align__(16) unsigned char smem[];
    const int T = threadIdx.x, t = T & 255;
    const int s = __builtin_amdgcn_readfirstlane(T >> 8);               // residue group: the same for a whole wavefront
    const int id = blockIdx.x, grp = id / (8 * RP), rp = (id / 8) % RP, unit = grp * 8 + id % 8;
    if (unit >= nsplit * nch) return;
    const int rbase = rp * G, r = rbase + s;
    const int split = unit % nsplit, ch = unit / nsplit;
    const int per = (nblk + nsplit - 1) / nsplit;
    const int b0 = split * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    void *image = smem + (size_t)s * S::kLdsBytes;                      // this group's transform image
    C *et = reinterpret_cast<C *>(smem + (size_t)G * S::kLdsBytes);     // exp(-2 pi i t / N), t < 256
    double *m2s = reinterpret_cast<double *>(et + 256) + T;             // this lane's S-meter sum
    // exp(-2 pi i k / 256), k < 16, the middle pass's twiddles: in the two pad scalars behind rows 240 .. 255 of the last image (the
    // transforms never touch the pads, and the groups' exchange area, which is laid over the images, ends before these rows)
    static_assert(S::kPad == 2 && (G * 8) * 256 * 16 <= (G - 1) * S::kLdsBytes + 240 * (16 + S::kPad) * 8, "the pads used lie behind the exchange area");
    double *tap = reinterpret_cast<double *>(smem + (size_t)(G - 1) * S::kLdsBytes) + 240 * (16 + S::kPad) + 16;
    if (T < 256) { C e; sincospi(-2.0 * (double)T / (double)N, &e.y, &e.x); et[T] = e; }
    if (T < 16) *reinterpret_cast<C *>(tap + (16 + S::kPad) * T) = tw[T];
    *m2s = 0.0;
    // e_j = exp(-2 pi i m_j / N), m_j = t + 256 (MJ s + j): the table's entry times the group's phasor (scalar registers), then steps of 256
    C es;
    sincospi(-2.0 * (double)(256 * MJ * s) / (double)N, &es.y, &es.x);
    es.x = uniform_f64(es.x); es.y = uniform_f64(es.y);
    C estep;
    estep.x = 0.99518472667219693; estep.y = -0.098017140329560604;            // exp(-2 pi i 256 / N) = exp(-i pi / 32): literals stay out of the vector registers
    const PanBand pb = band[ch];
    unsigned wp = 0;                        // bit i: bin i of this lane counts fully in the S-meter sum, bit 16 + i: with weight frac
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int bin = 4 * (t + NT * i) + r;
        const int sb = bin >= N / 2 ? bin - N : bin;
        if (pb.valid) {
            if (sb >= pb.first && sb < pb.first + pb.nwhole) wp |= 1u << i;
            else if (sb == pb.first + pb.nwhole) wp |= 0x10000u << i;
        }
    }
    // the S-meter passband is a few bins wide: almost every wavefront holds none of them and skips that sum
    const bool in_band = __ballot(wp != 0) != 0ull;
    // |X| sums of the workgroup's block range: sixteen registers per lane (read-modify-write of `partial` once per block, even
    // laid out so that a wavefront's accesses coalesce, was 0.19 of this kernel's 0.64 ms on config 3)
    double racc[E];
#pragma unroll
    for (int i = 0; i < E; i++) racc[i] = 0.0;
    __syncthreads();                        // the tables are written
    // Every load of a round -- with G = 4 of the block: sixteen, in the 64 registers the transform's values take later -- is asked for
    // before the first is used: one memory latency per block, not one per j (with one workgroup on the CU nothing else runs while it
    // waits).  (Round 5 measured this form slower: its sums were already spilling.)  PIPE (G = 4): and asked for a block AHEAD, four at a
    // time as the |X| loop at the end of a block lets go of the transform's values; the barriers in between order LDS only, so the loads
    // stay in flight across them and the latency lies under the end of one block and the top of the next.
    constexpr bool PIPE = QH_PAN_PIPE && QH_PAN_LOADS_AHEAD && G == 4;
    constexpr int LR = QH_PAN_LOADS_AHEAD ? (G == 4 ? 1 : 4) : MJ;     // rounds of loads per block (G = 2: eight loads a round beside half the transform's values)
    constexpr int LJ = MJ / LR;                                         // values of j per round of loads
    constexpr int PK = PIPE ? QH_PAN_PIPE_AHEAD : 0;                   // values of j whose loads are asked for a block ahead (the first round's)
    C xx[LJ][4];
    if (PIPE && b0 < b1) {
        const C *xb = in + (long long)ch * in_stride + (long long)b0 * N + 256 * MJ * s;
        asm volatile("" : "+s"(xb));
#pragma unroll
        for (int k = 0; k < PK; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * k + M * q, (unsigned)t);
    }
    for (int blk = b0; blk < b1; blk++) {
        // (a scalar base and the lane's 32-bit offset, not a 64-bit address per lane and strip; xn: the NEXT block's -- the last block
        // asks for itself again and drops it)
        const C *xb = in + (long long)ch * in_stride + (long long)blk * N + 256 * MJ * s;
        asm volatile("" : "+s"(xb));
        const C *xn = xb + (blk + 1 < b1 ? N : 0);
        asm volatile("" : "+s"(xn));
        // The sixteen results of a thread change hands in TWO rounds of eight, real and imaginary part together (128-bit accesses):
        // round h takes j = h MJ/2 .. h MJ/2 + MJ/2 - 1.  [group][ii = (MJ/2) s + jj][t] complex, laid over the transform images.  (Round
        // 4's form sent all sixteen real parts, then all sixteen imaginary parts, and carried the imaginary parts through the first
        // pass in 32 registers of a kernel that spills.)
        C *xc = reinterpret_cast<C *>(smem);
        int tt = t;
        asm volatile("" : "+v"(tt));        // (what is derived from t below is derived in the loop: addresses hoisted out of it are spilled)
        C e = cmul(et[tt], es);
        C u[E];
        PAN_STAMP(6);
        if (LR == 1) {                      // (ahead of the barrier: the loads do not touch LDS)
#pragma unroll
            for (int k = PK; k < LJ; k++)
#pragma unroll
                for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * k + M * q, (unsigned)tt);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            lds_barrier();                  // the area is free: the previous block's transform / the first round's values have been read out
            if (h == 0) PAN_STAMP(0);
#pragma unroll
            for (int jj = 0; jj < MJ / 2; jj++) {
                const int j = h * (MJ / 2) + jj;
                if (LR > 1 && j % LJ == 0) {
#pragma unroll
                    for (int k = 0; k < LJ; k++)
#pragma unroll
                        for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * (j + k) + M * q, (unsigned)tt);         // scalar base
                    __builtin_amdgcn_sched_barrier(0);
                }
                const C x0 = xx[j % LJ][0], x1 = xx[j % LJ][1], x2 = xx[j % LJ][2], x3 = xx[j % LJ][3];
                // Hanning window 0.5 - 0.5 cos(2 pi n / N) at n = m + M q (quisk.c:6008): cos(a + pi q / 2) = cos a, -sin a, -cos a, sin a
                // with e = (cos a, -sin a)
                const double g0 = __builtin_fma(-0.5, e.x, 0.5), g2 = __builtin_fma(0.5, e.x, 0.5);
                const double g1 = __builtin_fma(-0.5, e.y, 0.5), g3 = __builtin_fma(0.5, e.y, 0.5);
                const C v0 = mk<double>(x0.x * g0, x0.y * g0), v1 = mk<double>(x1.x * g1, x1.y * g1);
                const C v2 = mk<double>(x2.x * g2, x2.y * g2), v3 = mk<double>(x3.x * g3, x3.y * g3);
                // sum_q W_4^(q r) v_q, r = 0 .. 3 (W_4 = -i), then W_N^(m r) = e^r
                const C a0 = cadd(v0, v2), a1 = csub(v0, v2), c0 = cadd(v1, v3), c1 = csub(v1, v3);
                const C e2 = cmul(e, e);
                C o[G];
                if (G == 4 || rbase == 0) {     // workgroup-uniform
                    o[0] = cadd(a0, c0);
                    o[1] = cmul(mk<double>(a1.x + c1.y, a1.y - c1.x), e);           // a1 - i c1
                }
                if (G == 4 || rbase == 2) {
                    o[G - 2] = cmul(csub(a0, c0), e2);
                    o[G - 1] = cmul(mk<double>(a1.x - c1.y, a1.y + c1.x), cmul(e2, e));        // a1 + i c1
                }
#pragma unroll
                for (int g = 0; g < G; g++) xc[(g * 8 + (MJ / 2) * s + jj) * 256 + tt] = o[g];
                e = cmul(e, estep);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (h == 1) PAN_STAMP(1);
            lds_barrier();
            if (h == 1) PAN_STAMP(2);
            // group s takes element i = MJ s' + j of its transform from thread (s', t): slot ii = (MJ/2) s' + jj of its own plane
#pragma unroll
            for (int ii = 0; ii < 8; ii++) u[MJ * (ii / (MJ / 2)) + h * (MJ / 2) + ii % (MJ / 2)] = xc[(s * 8 + ii) * 256 + tt];
        }
        lds_barrier();                      // everybody has taken its elements: the images may be written
        PAN_STAMP(3);                       // the exchange between the residue groups
        // The 4096-point transform (FftSplit4096::run_at's three passes) with each pass's twiddle fetched where it is used: the middle
        // pass's from the sixteen-entry table, the last pass's exp(-2 pi i t / 4096) as the fourth power of the window table's entry.
        {
            double *lds = reinterpret_cast<double *>(image);
            Dft<16, false, C>::run(u);
            S::template exchange<1, 256 + 16 * S::kPad, true>(u, lds, (16 + S::kPad) * tt, S::sphys(tt));
            const int base = stockham_butterfly<4096, 16, 16, false, C, true>(u, tt, *reinterpret_cast<const C *>(tap + (16 + S::kPad) * (tt & 15)));
            lds_barrier();
            S::template exchange<16 + S::kPad, 256 + 16 * S::kPad, true>(u, lds, S::sphys(base), S::sphys(tt));
            C wb = et[tt];
            wb = cmul(wb, wb);
            wb = cmul(wb, wb);
            stockham_butterfly<4096, 16, 256, false, C, true>(u, tt, wb);
        }
        PAN_STAMP(4);                       // the 4096-point transform
        double m2 = 0.0;
        if (in_band) m2 = *m2s;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const double pw2 = u[i].x * u[i].x + u[i].y * u[i].y;
            racc[i] += sqrt_pow(pw2);       // cabs(): no overflow / underflow concern at +-2^31 * N full scale
            if (PIPE) asm volatile("" : "+v"(racc[i]));     // (pinned: left alone the sums sink below the loads and sixteen |X|^2 wait for them)
            if (in_band) m2 += ((wp >> i) & 1u) ? pw2 : (((wp >> (16 + i)) & 1u) ? pb.frac * pw2 : 0.0);
            if (PIPE && (i & 1) == 1) __builtin_amdgcn_sched_barrier(0);       // two square roots' worth of temporaries at a time
            if ((i & 3) == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (PIPE) {                 // values of the transform are done with: the next block's loads take their registers, QH_PAN_PIPE_LAG values behind
#pragma unroll
                    for (int k = 0; k < PK; k++)
                        if (i == (4 * k + 3 + QH_PAN_PIPE_LAG < E ? 4 * k + 3 + QH_PAN_PIPE_LAG : E - 1)) {
#pragma unroll
                            for (int q = 0; q < 4; q++) xx[k][q] = gload(xn + 256 * k + M * q, (unsigned)tt);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (in_band) *m2s = m2;
        PAN_STAMP(5);                       // |X| and the sums
    }
    // partial[split][ch][r][t + 256 i] for bin = 4 (t + 256 i) + r (pan_reduce_kernel undoes the order): coalesced stores; a block range
    // that came out empty (nblk not a multiple of the ranges) stores its zeros
    double *pp = partial + ((long long)split * nch + ch) * N + r * M + t;
#pragma unroll
    for (int i = 0; i < E; i++) pp[256 * i] = racc[i];
    // S-meter partial: lanes -> wave -> group of four waves (one r), fixed order, in the layout of pan_spectrum_kernel
    double m2 = *m2s;
    for (int d = 32; d > 0; d >>= 1) m2 += __shfl_down(m2, d, 64);
    __syncthreads();                        // (the last block's transform has left the images)
    double *wsum = reinterpret_cast<double *>(smem);
    if ((T & 63) == 0) wsum[T >> 6] = m2;
    __syncthreads();
    if (t == 0) {
        double sm = 0.0;
        for (int k = 0; k < 4; k++) sm += wsum[4 * s + k];
        partial_m2[((long long)split * nch + ch) * 4 + r] = sm;
    }
}

// ---- fft_size 16384 with a decimating FIR on the same read (BASELINE config 3: 1023-tap /32 beside the panadapter) -----------
// quisk_process_samples copies the raw block into the FFT ring (quisk.c:2454-2475) and decimates the same samples
// (quisk_cDecimate, filter.c:203-229): one stream, two consumers.  pan16k_kernel already holds the whole 16384-sample block of a
// channel in one workgroup; this variant makes its transform serve both:
//   * the transform is taken of the UNWINDOWED block; the Hanning window 0.5 - 0.5 cos(2 pi n / N) (quisk.c:6008) is applied in the
//     frequency domain, X_w[k] = X[k] / 2 - (X[k - 1] + X[k + 1]) / 4 -- the neighbours of bin 4 m + r are bins of the residue
//     groups r - 1 and r + 1 at the same m (one step along m at the two ends), fetched through the exchange area;
//   * the FIR is the circular convolution of the block with the taps, decimated by D = 32: Z[k] = X[k] H'[k] with
//     H'[k] = FFT_N(h)[k] exp(2 pi i (D - 1) k / N) / N (the output of index m is the sample of time D m + D - 1, filter.c:216),
//     folded over the 32 images of every bin class k mod 512 -- all of a lane's sixteen bins and those of lane t +- 128 of its
//     group are ONE class -- and handed to panfir_tail_kernel (panfir_finish_block) as 512 folded bins per block: a 512-point inverse transform, and
//     the circular wrap of the first (ntaps - 1) / D outputs repaired by direct sums over the difference between the previous
//     block's and this block's last ntaps - 1 samples.
// The input is read once: 16 B + 16 / 32 B per sample (16.5), where FIR bank + panadapter read it twice.
template <int DUMMY>
__global__ __launch_bounds__(1024, 4) void panfir16k_kernel(const double2 *in, long long in_stride, int nblk, int nsplit,
                                                            const double2 *tw, double *partial, double *partial_m2,
                                                            const PanBand *band, int nch, const double2 *firH, double2 *yfold)
{
    using C = double2;
    using S = FftSplit4096<false, C>;
    constexpr int M = 4096, N = 16384, E = 16, G = 4, MJ = 4;
    extern __shared__ __align__(16) unsigned char smem[];
    const int T = threadIdx.x, t = T & 255;
    const int s = __builtin_amdgcn_readfirstlane(T >> 8);               // residue group: the same for a whole wavefront
    const int id = blockIdx.x, grp = id / 8, unit = grp * 8 + id % 8;
    if (unit >= nsplit * nch) return;
    const int r = s;
    const int split = unit % nsplit, ch = unit / nsplit;
    const int per = (nblk + nsplit - 1) / nsplit;
    const int b0 = split * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    double *xch = reinterpret_cast<double *>(smem);                     // exchange area [group][i][t] scalars: 128 KB
    C *fold = reinterpret_cast<C *>(smem + 4 * 16 * 256 * 8);           // [group][t] complex: the last 16 KB of the four images
    void *image = smem + (size_t)s * S::kLdsBytes;                      // this group's transform image
    // (the tables behind the images and the register budget: see pan16k_kernel)
    C *et = reinterpret_cast<C *>(smem + (size_t)G * S::kLdsBytes);     // exp(-2 pi i t / N), t < 256
    C *ta = et + 256;                                                   // exp(-2 pi i k / 256), k < 16
    double *m2s = reinterpret_cast<double *>(ta + 16) + T;              // this lane's S-meter sum
    if (T < 256) { C e; sincospi(-2.0 * (double)T / (double)N, &e.y, &e.x); et[T] = e; }
    if (T < 16) ta[T] = tw[T];
    *m2s = 0.0;
    C es;
    sincospi(-2.0 * (double)(256 * MJ * s) / (double)N, &es.y, &es.x);
    es.x = uniform_f64(es.x); es.y = uniform_f64(es.y);
    C estep;
    estep.x = 0.99518472667219693; estep.y = -0.098017140329560604;    // exp(-2 pi i 256 / N)
    const PanBand pb = band[ch];
    unsigned wp = 0;
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int bin = 4 * (t + NT * i) + r;
        const int sb = bin >= N / 2 ? bin - N : bin;
        if (pb.valid) {
            if (sb >= pb.first && sb < pb.first + pb.nwhole) wp |= 1u << i;
            else if (sb == pb.first + pb.nwhole) wp |= 0x10000u << i;
        }
    }
    const bool in_band = __ballot(wp != 0) != 0ull;
    double racc[E];
#pragma unroll
    for (int i = 0; i < E; i++) racc[i] = 0.0;
    __syncthreads();                        // the tables are written
    // (loads a round ahead and barriers that order LDS only: see pan16k_kernel)
    constexpr bool PIPE = QH_PAN_PIPE && QH_PAN_LOADS_AHEAD;
    constexpr int LR = QH_PAN_LOADS_AHEAD ? 1 : MJ, LJ = MJ / LR;
    constexpr int PK = PIPE ? QH_PAN_PIPE_AHEAD : 0;
    C xx[LJ][4];
    if (PIPE && b0 < b1) {
        const C *xb = in + (long long)ch * in_stride + (long long)b0 * N + 256 * MJ * s;
        asm volatile("" : "+s"(xb));
#pragma unroll
        for (int k = 0; k < PK; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * k + M * q, (unsigned)t);
    }
    for (int blk = b0; blk < b1; blk++) {
        const C *xb = in + (long long)ch * in_stride + (long long)blk * N + 256 * MJ * s;
        asm volatile("" : "+s"(xb));
        const C *xn = xb + (blk + 1 < b1 ? N : 0);
        asm volatile("" : "+s"(xn));
        C *xc = reinterpret_cast<C *>(smem);
        int tt = t;
        asm volatile("" : "+v"(tt));
        C e = cmul(et[tt], es);
        C u[E];
        // the block's four strips, the radix-4 combination over them, W_N^(m r), and the change of hands between the residue groups in two
        // rounds of eight complex values (pan16k_kernel's, without the window)
        PAN_STAMP(6);
        if (LR == 1) {
#pragma unroll
            for (int k = PK; k < LJ; k++)
#pragma unroll
                for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * k + M * q, (unsigned)tt);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            lds_barrier();                  // the images, the fold rows and the first round's values are free
            if (h == 0) PAN_STAMP(0);
#pragma unroll
            for (int jj = 0; jj < MJ / 2; jj++) {
                const int j = h * (MJ / 2) + jj;
                if (LR > 1 && j % LJ == 0) {
#pragma unroll
                    for (int k = 0; k < LJ; k++)
#pragma unroll
                        for (int q = 0; q < 4; q++) xx[k][q] = gload(xb + 256 * (j + k) + M * q, (unsigned)tt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const C v0 = xx[j % LJ][0], v1 = xx[j % LJ][1], v2 = xx[j % LJ][2], v3 = xx[j % LJ][3];
                const C a0 = cadd(v0, v2), a1 = csub(v0, v2), c0 = cadd(v1, v3), c1 = csub(v1, v3);
                const C e2 = cmul(e, e);
                C o[G];
                o[0] = cadd(a0, c0);
                o[1] = cmul(mk<double>(a1.x + c1.y, a1.y - c1.x), e);
                o[2] = cmul(csub(a0, c0), e2);
                o[3] = cmul(mk<double>(a1.x - c1.y, a1.y + c1.x), cmul(e2, e));
#pragma unroll
                for (int g = 0; g < G; g++) xc[(g * 8 + (MJ / 2) * s + jj) * 256 + tt] = o[g];
                e = cmul(e, estep);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (h == 1) PAN_STAMP(1);
            lds_barrier();
            if (h == 1) PAN_STAMP(2);
#pragma unroll
            for (int ii = 0; ii < 8; ii++) u[MJ * (ii / (MJ / 2)) + h * (MJ / 2) + ii % (MJ / 2)] = xc[(s * 8 + ii) * 256 + tt];
        }
        lds_barrier();
        PAN_STAMP(3);
        {
            double *lds = reinterpret_cast<double *>(image);
            Dft<16, false, C>::run(u);
            S::template exchange<1, 256 + 16 * S::kPad, true>(u, lds, (16 + S::kPad) * tt, S::sphys(tt));
            const int base = stockham_butterfly<4096, 16, 16, false, C, true>(u, tt, ta[tt & 15]);
            lds_barrier();
            S::template exchange<16 + S::kPad, 256 + 16 * S::kPad, true>(u, lds, S::sphys(base), S::sphys(tt));
            C wb = et[tt];
            wb = cmul(wb, wb);
            wb = cmul(wb, wb);
            stockham_butterfly<4096, 16, 256, false, C, true>(u, tt, wb);       // u[i] = X[4 (t + 256 i) + r], unwindowed
        }
        PAN_STAMP(4);
        // ---- the FIR's share: sum of the lane's sixteen products (one class k mod 512), lanes t and t + 128 joined below
#ifndef QH_PANFIR_NOFIR        // (tools/ab_bench.py attribution builds: timing only)
        {
            const C *hh = firH + r * M;     // scalar
            asm volatile("" : "+s"(hh));
            C acc = cmul(u[0], gload(hh, (unsigned)tt));
#pragma unroll
            for (int i = 1; i < E; i++) {
                const C z = cmul(u[i], gload(hh + 256 * i, (unsigned)tt));
                acc.x += z.x; acc.y += z.y;
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();                // every group has read its transform out of its image
            fold[s * 256 + tt] = acc;
        }
#else
        lds_barrier();
#endif
        PAN_STAMP(7);
        // ---- the window in the frequency domain: real parts, then imaginary parts, through the exchange area
#ifndef QH_PANFIR_NOHANN
        // neighbours in the exchange area: bin 4 m + r + 1 is residue r + 1 at m (r = 3: residue 0 at m + 1), bin 4 m + r - 1 residue r - 1
        // at m (r = 0: residue 3 at m - 1); m = t + 256 i is linear in the [i][t] rows, so the two ends are one address step -- except
        // for the two bins where the spectrum wraps round
        const int up0 = s < 3 ? ((s + 1) * 16) * 256 + tt : tt + 1;
        const int dn0 = s > 0 ? ((s - 1) * 16) * 256 + tt : (48 * 256) + tt - 1;
        double *own = xch + (s * 16) * 256 + tt;
#pragma unroll
        for (int i = 0; i < E; i++) own[i * 256] = u[i].x;
        lds_barrier();
#ifndef QH_PANFIR_NOFIR
        if (tt < 128) {
            const C a = fold[s * 256 + tt], b = fold[s * 256 + tt + 128];
            yfold[((long long)ch * nblk + blk) * 512 + 4 * tt + r] = mk<double>(a.x + b.x, a.y + b.y);
        }
#endif
#pragma unroll
        for (int i = 0; i < E; i++) {
            int ua = up0 + i * 256, da = dn0 + i * 256;
            if (s == 3 && i == 15 && tt == 255) ua = 0;                 // X[16384] = X[0]
            if (s == 0 && i == 0 && tt == 0) da = 63 * 256 + 255;       // X[-1] = X[16383]
            u[i].x = __builtin_fma(0.5, u[i].x, -0.25 * (xch[ua] + xch[da]));
            // (pinned: left alone, the sums sink to where |X| is formed, behind the imaginary parts' exchange, and the thirty-two
            // neighbours wait for them in 64 registers)
            asm volatile("" : "+v"(u[i].x));
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < E; i++) own[i * 256] = u[i].y;
        lds_barrier();
        double m2 = 0.0;
        if (in_band) m2 = *m2s;
#pragma unroll
        for (int i = 0; i < E; i++) {
            int ua = up0 + i * 256, da = dn0 + i * 256;
            if (s == 3 && i == 15 && tt == 255) ua = 0;
            if (s == 0 && i == 0 && tt == 0) da = 63 * 256 + 255;
            const double uy = __builtin_fma(0.5, u[i].y, -0.25 * (xch[ua] + xch[da]));
            const double pw2 = u[i].x * u[i].x + uy * uy;
            racc[i] += sqrt_pow(pw2);
            asm volatile("" : "+v"(racc[i]));       // (pinned, like the real parts above)
            if (in_band) m2 += ((wp >> i) & 1u) ? pw2 : (((wp >> (16 + i)) & 1u) ? pb.frac * pw2 : 0.0);
            if ((i & 3) == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (PIPE) {                 // the next block's first loads in the registers of the values that are done with
#pragma unroll
                    for (int k = 0; k < PK; k++)
                        if (i == (4 * k + 3 + QH_PAN_PIPE_LAG < E ? 4 * k + 3 + QH_PAN_PIPE_LAG : E - 1)) {
#pragma unroll
                            for (int q = 0; q < 4; q++) xx[k][q] = gload(xn + 256 * k + M * q, (unsigned)tt);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#else
        double m2 = 0.0;
        if (in_band) m2 = *m2s;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const double pw2 = u[i].x * u[i].x + u[i].y * u[i].y;
            racc[i] += sqrt_pow(pw2);
            if (in_band) m2 += ((wp >> i) & 1u) ? pw2 : (((wp >> (16 + i)) & 1u) ? pb.frac * pw2 : 0.0);
        }
#endif
        if (in_band) *m2s = m2;
        PAN_STAMP(5);
    }
    double *pp = partial + ((long long)split * nch + ch) * N + r * M + t;
#pragma unroll
    for (int i = 0; i < E; i++) pp[256 * i] = racc[i];
    double m2 = *m2s;
    for (int d = 32; d > 0; d >>= 1) m2 += __shfl_down(m2, d, 64);
    __syncthreads();
    double *wsum = reinterpret_cast<double *>(smem);
    if ((T & 63) == 0) wsum[T >> 6] = m2;
    __syncthreads();
    if (t == 0) {
        double sm = 0.0;
        for (int k = 0; k < 4; k++) sm += wsum[4 * s + k];
        partial_m2[((long long)split * nch + ch) * 4 + r] = sm;
    }
}

// The FIR outputs of one 16384-sample block from its 512 folded bins: unnormalised inverse transform (the 1 / N sits in H'), then
// the circular wrap of outputs m < ncorr repaired: y[m] += sum_{k > n} h[k] (xprev - xcur)[N + n - k], n = D m + D - 1 -- the circular
// convolution took the block's own last samples where the previous block's belong (`hist`: the last `hl` samples of the call
// before, for block 0).  Thread 8 m + p takes every eighth tap of output m.
__device__ __forceinline__ void panfir_finish_block(int blk, int ch, const double2 *yfold, int nblk, const double2 *tw512, const double *taps, int ntaps,
                                                    int D, const double2 *in, long long in_stride, const double2 *hist, int hl,
                                                    double2 *out, long long out_stride)
{
    using C = double2;
    using F = TileFft<512, true, C>;
    __shared__ __align__(16) unsigned char fimg[F::kLdsBytes];
    __shared__ C corr[32];
    __shared__ C diff[1024];                // diff[d] = (previous block - this block)[N - d], d = 1 .. hl
    __shared__ double htap[1024];
    constexpr int N = 16384;
    const int t = threadIdx.x;
    const C *yf = yfold + ((long long)ch * nblk + blk) * 512;
    C z[2] = { yf[t], yf[t + 256] };
    const C *cur = in + (long long)ch * in_stride + (long long)blk * N + N;         // cur[-d] = the d-th last sample of this block
    const C *prv = blk > 0 ? cur - N : hist + (long long)ch * hl + hl;              // the same of the block before
    for (int d = 1 + t; d <= hl; d += NT) {
        const C a = prv[-d], b = cur[-d];
        diff[d] = mk<double>(a.x - b.x, a.y - b.y);
    }
    for (int k = t; k < ntaps; k += NT) htap[k] = taps[k];
    F::run(z, fimg, F::load(tw512));        // (its barriers also publish diff[] and htap[])
    // Output m needs the taps behind n = D m + D - 1: (ntaps - 1 - n) products, most for m = 0, none for the last.  Outputs m and
    // 31 - m together are the same length for every m: sixteen lanes take such a pair, every sixteenth tap each (62 products a lane
    // for 1023 taps instead of 124 in the longest of eight-lane rows), four sums in flight.
    static_assert(NT == 256, "sixteen pairs of outputs x sixteen lanes");
    const int q = t >> 4, l = t & 15;
    C part[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const int m = w == 0 ? q : 31 - q, n = D * m + D - 1;
        C a0 = mk<double>(0, 0), a1 = a0, a2 = a0, a3 = a0;
        int k = n + 1 + l;
        for (; k + 48 < ntaps; k += 64) {
            const C d0 = diff[k - n], d1 = diff[k + 16 - n], d2 = diff[k + 32 - n], d3 = diff[k + 48 - n];
            const double h0 = htap[k], h1 = htap[k + 16], h2 = htap[k + 32], h3 = htap[k + 48];
            a0.x = __builtin_fma(h0, d0.x, a0.x); a0.y = __builtin_fma(h0, d0.y, a0.y);
            a1.x = __builtin_fma(h1, d1.x, a1.x); a1.y = __builtin_fma(h1, d1.y, a1.y);
            a2.x = __builtin_fma(h2, d2.x, a2.x); a2.y = __builtin_fma(h2, d2.y, a2.y);
            a3.x = __builtin_fma(h3, d3.x, a3.x); a3.y = __builtin_fma(h3, d3.y, a3.y);
        }
        for (; k < ntaps; k += 16) {
            const C dv = diff[k - n];
            const double h = htap[k];
            a0.x = __builtin_fma(h, dv.x, a0.x); a0.y = __builtin_fma(h, dv.y, a0.y);
        }
        part[w] = mk<double>((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
    }
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) {
#pragma unroll
        for (int w = 0; w < 2; w++) { part[w].x += __shfl_xor(part[w].x, sft, 64); part[w].y += __shfl_xor(part[w].y, sft, 64); }
    }
    if (l == 0) { corr[q] = part[0]; corr[31 - q] = part[1]; }
    __syncthreads();
    if (t < 32) { z[0].x += corr[t].x; z[0].y += corr[t].y; }
    C *o = out + (long long)ch * out_stride + (long long)blk * (N / D);
    o[t] = z[0]; o[t + 256] = z[1];
}

// fft_avg += sum over the block ranges; meter += sum over ranges and r.  One thread per (channel, index).
// by_residue: the partial sums lie as [r][bin / 4] (pan16k_kernel) instead of in fft_avg's order.
__device__ __forceinline__ void pan_reduce_block(int bx, int ch, int nch, const double *partial, const double *partial_m2, int nsplit, int N, int R,
                                                 double *avg, double *meter, int by_residue)
{
    const int idx = bx * NT + threadIdx.x;
    if (idx < N) {
        double sacc = 0.0;
        int src = idx;
        if (by_residue) {
            const int bin = (idx + N / 2) % N;
            src = (bin & 3) * (N / 4) + (bin >> 2);
        }
        for (int sp = 0; sp < nsplit; sp++) sacc += partial[((long long)sp * nch + ch) * N + src];
        avg[(long long)ch * N + idx] += sacc;
    }
    if (idx == 0) {
        double sm = 0.0;
        for (int sp = 0; sp < nsplit; sp++)
            for (int r = 0; r < R; r++) sm += partial_m2[((long long)sp * nch + ch) * R + r];
        meter[ch] += sm;
    }
}
__global__ __launch_bounds__(NT) void pan_reduce_kernel(const double *partial, const double *partial_m2, int nsplit, int N, int R,
                                                        double *avg, double *meter, int by_residue)
{
    pan_reduce_block(blockIdx.x, blockIdx.y, gridDim.y, partial, partial_m2, nsplit, N, R, avg, meter, by_residue);
}

// Everything behind panfir16k_kernel in ONE launch (three small grids one after the other were 0.10 of config 3's 0.55 ms): workgroups
// x < nblk finish block x's FIR outputs, the next N / NT add the block ranges' |X| sums to fft_avg (pan_reduce_kernel's), the last
// ones keep the call's last samples for the next call's first block (they read `in` and write the OTHER history buffer).
struct PanfirTail {
    const double2 *yfold, *tw512, *in, *hist; const double *taps; double2 *out, *hist_next; const double *partial, *partial_m2; double *avg, *meter;
    long long in_stride, out_stride; int nblk, ntaps, D, hl, nsplit, N, R, n;
};
__global__ __launch_bounds__(NT) void panfir_tail_kernel(PanfirTail a)
{
    const int bx = blockIdx.x, ch = blockIdx.y;
    const int nred = (a.N + NT - 1) / NT;
    if (bx < a.nblk) {
        panfir_finish_block(bx, ch, a.yfold, a.nblk, a.tw512, a.taps, a.ntaps, a.D, a.in, a.in_stride, a.hist, a.hl, a.out, a.out_stride);
    } else if (bx < a.nblk + nred) {
        pan_reduce_block(bx - a.nblk, ch, gridDim.y, a.partial, a.partial_m2, a.nsplit, a.N, a.R, a.avg, a.meter, 1);
    } else {
        const int j = (bx - a.nblk - nred) * NT + threadIdx.x;
        if (j < a.hl) a.hist_next[(long long)ch * a.hl + j] = a.in[(long long)ch * a.in_stride + a.n - a.hl + j];
    }
}

__global__ void pan_graph_kernel(double *avg, double *meter, int nch, int N, int data_width, double rate, double zoom,
                                 double deltaf, int count, double *pixels, double *smeter)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    double *a = avg + (long long)ch * N;
    double *p = pixels + (long long)ch * data_width;
    const double scale = 20.0 * (log10((double)count) + log10((double)N) + 31.0 * log10(2.0));
    int n = (int)(zoom * (double)N / data_width + 0.5);
    if (n < 1) n = 1;
    for (int i = 0; i < data_width; i++) {
        int k = (int)(N * (deltaf / rate + zoom * ((double)i / data_width - 0.5) + 0.5) + 0.1);
        double d2 = 0.0;
        for (int j = 0; j < n; j++, k++)
            if (k >= 0 && k < N) d2 += a[k];
        a[i] = d2;
    }
    const double ss = 1.0 / 2147483647.0 / N;
    double sm = meter[ch] * ss * ss / count;
    sm = sm > 1E-16 ? 10.0 * log10(sm) : -160.0;
    smeter[ch] = sm + 4.25969;
    meter[ch] = 0.0;
    for (int i = 0; i < data_width; i++) {
        double d2 = 20.0 * log10(a[i]) - scale;
        if (d2 < -200) d2 = -200; else if (d2 > 0) d2 = 0;
        p[i] = d2;
    }
    for (int i = 0; i < N; i++) a[i] = 0.0;
}

// watfall_OnGraphData (quisk.c:5372-5421): one waterfall row per channel from the dB row of get_graph -- colour index
// l = (int)((dB - gain + yz) * (y_scale + 10) * 0.10 + 128) clamped to 0..255, yz = 40 + 0.69 y_zero, through the
// 256-entry red / green / blue tables; pixels past the dB row are black.  palette = red[256] green[256] blue[256].
__global__ __launch_bounds__(NT) void watfall_row_kernel(const double *db, long long db_stride, int size, int width,
                                                         const unsigned char *palette, int y_zero, int y_scale, double gain,
                                                         unsigned char *rgb)
{
    // the colour index is a truncation: a fused multiply-add would round the product differently right at an integer
#pragma clang fp contract(off)
    __shared__ unsigned char pal[768];
    for (int j = threadIdx.x; j < 768; j += NT) pal[j] = palette[j];
    __syncthreads();
    const int row = blockIdx.y;
    const double yz = 40.0 + y_zero * 0.69;
    unsigned char *o = rgb + (long long)row * width * 3;
    for (int i = blockIdx.x * NT + threadIdx.x; i < width; i += gridDim.x * NT) {
        unsigned char r = 0, g = 0, b = 0;
        if (i < size) {
            const double v = (db[(long long)row * db_stride + i] - gain + yz) * (y_scale + 10) * 0.10 + 128;
            int l = v < 0.0 ? 0 : v > 255.0 ? 255 : (int)v;        // (int) then the clamp, for every value the clamp leaves alone
            if (!(v == v)) l = 0;
            r = pal[l]; g = pal[256 + l]; b = pal[512 + l];
        }
        o[3 * i] = r; o[3 * i + 1] = g; o[3 * i + 2] = b;
    }
}

// ---- bandscope (get_bandscope, quisk.c:4957-5011): real ADC samples through the same windowed transform as (x, 0)
// out[ch][i] = (in[ch][i], 0); maxbits[ch] = max |in| (non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(NT) void bscope_convert_kernel(const double *in, long long in_stride, int n, double2 *out,
                                                            unsigned long long *maxbits)
{
    __shared__ double wmax[NT / 64];
    const int ch = blockIdx.y;
    double m = 0.0;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        const double v = in[(long long)ch * in_stride + i];
        out[(long long)ch * n + i] = make_double2(v, 0.0);
        m = fmax(m, fabs(v));
    }
    for (int d = 32; d > 0; d >>= 1) m = fmax(m, __shfl_down(m, d, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < NT / 64; k++) m = fmax(m, wmax[k]);
        atomicMax(maxbits + ch, (unsigned long long)__double_as_longlong(m));
    }
}

// copy2pixels (quisk.c:4932-4955) over bandscopeAverage[0 .. N/2] (+ the zero the reference appends "in case we run off
// the end"), then scale and dB (quisk.c:4986-4999).  avg is the panadapter's image: bin k sits at (k + N/2) % N.
__global__ __launch_bounds__(NT) void bscope_graph_kernel(const double *avg, int N, int width, double rate, double zoom, double deltaf,
                                                          int count, double *pixels)
{
    const int ch = blockIdx.y, L = N / 2 + 1;
    const double *a = avg + (long long)ch * N;
    auto bin = [&](int k) -> double { return k >= 0 && k < L ? a[(k + N / 2) % N] : 0.0; };
    const double frac = (double)L / width, scale = 1.0 / frac / count / N;
    const double f1 = deltaf + rate / 2.0 * (1.0 - zoom);
    for (int i = blockIdx.x * NT + threadIdx.x; i < width; i += gridDim.x * NT) {
        const double d1 = L / rate * (f1 + (double)i / width * zoom * rate);
        const double d2 = L / rate * (f1 + (double)(i + 1) / width * zoom * rate);
        const int j1 = (int)floor(d1), j2 = (int)floor(d2);
        double sample;
        if (j1 == j2) {
            sample = (d2 - d1) * bin(j1);
        } else {
            sample = (j1 + 1 - d1) * bin(j1);
            for (int j = j1 + 1; j < j2; j++) sample += bin(j);
            sample += (d2 - j2) * bin(j2);
        }
        sample *= scale;
        pixels[(long long)ch * width + i] = sample <= 1E-10 ? -200.0 : 20.0 * log10(sample);
    }
}

// ---- panadapter sizes that are not a power of two (Quisk's fft_size = data_width * fft_mult with data_width = 2^a * y * z,
// y, z odd <= 15, quisk.py:186-194,4179): Bluestein's chirp-z on the power-of-two transforms.  With w_n = exp(i pi n^2 / N):
//   X[k] = conj(w_k) * sum_n (x[n] conj(w_n)) w_(k-n)   -- a circular convolution of length BM >= 2 N - 1,
// and |X[k]| = |conv[k]| (the last chirp has unit modulus), which is all get_graph keeps.
// gfft_kernel: one forward transform of BM = R * M points per (f, r) workgroup pair, decimation in frequency like
// pan_spectrum_kernel (bins R k + r from an M-point LDS transform of the r-th combination), global -> global.
//   LOAD 0: src = the sample stream; element n of transform f = x[ch][blk N + n] * pre[n] for n < N, else 0
//   LOAD 1: src = transforms; element n = conj(src[f][n] * mul[n])   (inverse transform by conjugation)
//   STORE 0: dst[f][bin] = value;   STORE 1: mag[f][(bin + N/2) % N] = |value| * scale for bin < N
template <int M, int R, int LOAD, int STORE>
__global__ __launch_bounds__(NT) void gfft_kernel(const double2 *src, long long src_stride, int nblk_c, long long blk0, int N,
                                                  const double2 *aux, const double2 *tw, double2 *dst, double *mag, double scale)
{
    using C = double2;
    using F = TileFft<M, false, C>;
    constexpr int E = M / NT, BM = M * R;
    extern __shared__ __align__(16) unsigned char smem[];
    const int t = threadIdx.x, f = blockIdx.x / R, r = blockIdx.x % R;
    const typename F::Tw twf = F::load(tw);
    C u[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int m = t + NT * i;
        C sum = mk<double>(0, 0);
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int n = m + M * q;
            C v;
            if constexpr (LOAD == 0) {
                const int ch = f / nblk_c, b = f % nblk_c;
                v = mk<double>(0, 0);
                if (n < N) v = cmul(src[(long long)ch * src_stride + (blk0 + b) * N + n], aux[n]);
            } else {
                const C a = cmul(src[(long long)f * BM + n], aux[n]);
                v = mk<double>(a.x, -a.y);
            }
            if constexpr (R > 1) {
                C wq;                                           // W_R^(q r)
                sincospi(-2.0 * (double)((q * r) % R) / (double)R, &wq.y, &wq.x);
                v = cmul(v, wq);
            }
            sum = cadd(sum, v);
        }
        C wm;                                                   // W_BM^(m r)
        sincospi(-2.0 * (double)(((long long)m * r) % BM) / (double)BM, &wm.y, &wm.x);
        u[i] = R > 1 ? cmul(sum, wm) : sum;
    }
    F::run(u, smem, twf);
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int bin = R * (t + NT * i) + r;
        if constexpr (STORE == 0) dst[(long long)f * BM + bin] = u[i];
        else if (bin < N) mag[(long long)f * N + (bin + N / 2) % N] = sqrt(u[i].x * u[i].x + u[i].y * u[i].y) * scale;
    }
}

// fft_avg += the blocks' magnitudes (fixed order), S-meter += the passband's |X|^2 (quisk.c:5223-5244, 5271-5276)
__global__ __launch_bounds__(NT) void gfft_reduce_kernel(const double *mag, int nblk_c, int N, const PanBand *band, double *avg, double *meter)
{
    __shared__ double wsum[NT / 64];
    const int ch = blockIdx.y;
    const PanBand pb = band[ch];
    double m2 = 0.0;
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < N; idx += gridDim.x * NT) {
        double sacc = 0.0, p2 = 0.0;
        for (int b = 0; b < nblk_c; b++) { const double v = mag[((long long)ch * nblk_c + b) * N + idx]; sacc += v; p2 += v * v; }
        avg[(long long)ch * N + idx] += sacc;
        const int sb = idx - N / 2;                             // signed bin of this fftshifted index (N even)
        if (pb.valid) {
            if (sb >= pb.first && sb < pb.first + pb.nwhole) m2 += p2;
            else if (sb == pb.first + pb.nwhole) m2 += pb.frac * p2;
        }
    }
    // one workgroup per channel does the meter so that the sum has a fixed order
    if (gridDim.x == 1) {
        for (int d = 32; d > 0; d >>= 1) m2 += __shfl_down(m2, d, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = m2;
        __syncthreads();
        if (threadIdx.x == 0) { double sm = 0.0; for (int k = 0; k < NT / 64; k++) sm += wsum[k]; meter[ch] += sm; }
    }
}

// dst[ch][dst_off + i] = src[ch][src_off + i], i < n
__global__ void pan_copy_kernel(const double2 *src, long long src_stride, long long src_off, double2 *dst, long long dst_stride,
                                long long dst_off, int n)
{
    const int ch = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[(long long)ch * dst_stride + dst_off + i] = src[(long long)ch * src_stride + src_off + i];
}

struct Pan {
    int device = 0, nch = 0, N = 0, M = 0, R = 1, data_width = 0;
    double rate = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *avg = nullptr, *meter = nullptr, *pixels = nullptr, *smeter = nullptr;
    double2 *tw = nullptr, *carry = nullptr;
    double *partial = nullptr, *partial_m2 = nullptr;
    PanBand *band = nullptr;
    std::vector<PanBand> hband;
    int fill = 0, count = 0, max_split = 1;
    // Bluestein mode (fft_size not a power of two): transforms of BM = BR * M points
    bool blue = false;
    int BM = 0, BR = 1;
    double2 *b_pre = nullptr, *b_filt = nullptr, *wa = nullptr, *wb = nullptr;
    double *mag = nullptr;
    int chunk_blk = 0;          // blocks per channel the work buffers hold
    // a decimating FIR on the panadapter's read (qh_pan_attach_fir / qh_pan_feed_decimate; fft_size 16384)
    int fir_ntaps = 0, fir_decim = 0, fir_hl = 0;
    double2 *fir_H = nullptr, *fir_hist[2] = { nullptr, nullptr }, *fir_yf = nullptr, *fir_tw512 = nullptr;
    double *fir_taps = nullptr;
    long long fir_yf_cap = 0;
    int fir_cur = 0;

    ~Pan()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(avg); (void)hipFree(meter); (void)hipFree(pixels); (void)hipFree(smeter);
        (void)hipFree(tw); (void)hipFree(carry); (void)hipFree(partial); (void)hipFree(partial_m2); (void)hipFree(band);
        (void)hipFree(b_pre); (void)hipFree(b_filt); (void)hipFree(wa); (void)hipFree(wb); (void)hipFree(mag);
        (void)hipFree(fir_H); (void)hipFree(fir_hist[0]); (void)hipFree(fir_hist[1]); (void)hipFree(fir_yf); (void)hipFree(fir_tw512); (void)hipFree(fir_taps);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    template <int MM, int RR> void launch2(const double2 *p, long long src_stride, int nblk, int nsplit)
    {
        constexpr int lds = TileFft<MM, false, double2>::kLdsBytes + MM * 8;
        const int units = nsplit * nch, groups = (units + 7) / 8;
        hipLaunchKernelGGL((pan_spectrum_kernel<MM, RR>), dim3((unsigned)(groups * 8 * RR)), dim3(NT), lds, stream, p, src_stride, nblk,
                           nsplit, tw, partial, partial_m2, band, nch);
    }
    template <int MM> void launch(const double2 *p, long long src_stride, int nblk, int nsplit)
    {
        if (MM == 4096 && R == 4) {     // fft_size 16384: the one-read form
            constexpr int G = QH_PAN16K_GROUPS;
            const int units = nsplit * nch, groups = (units + 7) / 8;
            hipLaunchKernelGGL((pan16k_kernel<G>), dim3((unsigned)(groups * 8 * (4 / G))), dim3(256 * G), (size_t)pan16k_lds<G>(), stream, p,
                               src_stride, nblk, nsplit, tw, partial, partial_m2, band, nch);
            return;
        }
        if (R == 1) launch2<MM, 1>(p, src_stride, nblk, nsplit);
        else if (R == 2) launch2<MM, 2>(p, src_stride, nblk, nsplit);
        else launch2<MM, 4>(p, src_stride, nblk, nsplit);
    }

    template <int MM, int RR> void blue_launch(const double2 *src, long long src_stride, int nb, long long blk0)
    {
        constexpr size_t lds = TileFft<MM, false, double2>::kLdsBytes;
        const unsigned g = (unsigned)(nb * nch * RR);
        hipLaunchKernelGGL((gfft_kernel<MM, RR, 0, 0>), dim3(g), dim3(NT), lds, stream, src, src_stride, nb, blk0, N, b_pre, tw, wa,
                           (double *)nullptr, 1.0);
        hipLaunchKernelGGL((gfft_kernel<MM, RR, 1, 1>), dim3(g), dim3(NT), lds, stream, wa, 0, nb, 0, N, b_filt, tw, (double2 *)nullptr,
                           mag, 1.0 / (double)BM);
    }
    // fft_size that is not a power of two: Bluestein over chunks of blocks that fit the work buffers
    int run_blocks_blue(const double2 *src, long long src_stride, long long off, int nblk)
    {
        for (int done = 0; done < nblk; done += chunk_blk) {
            const int nb = nblk - done < chunk_blk ? nblk - done : chunk_blk;
            const double2 *p0 = src + off;
            switch (BR * 10000 + M) {
            case 10000 + 1024: blue_launch<1024, 1>(p0, src_stride, nb, done); break;
            case 10000 + 2048: blue_launch<2048, 1>(p0, src_stride, nb, done); break;
            case 10000 + 4096: blue_launch<4096, 1>(p0, src_stride, nb, done); break;
            case 20000 + 4096: blue_launch<4096, 2>(p0, src_stride, nb, done); break;
            case 40000 + 4096: blue_launch<4096, 4>(p0, src_stride, nb, done); break;
            default:           blue_launch<4096, 8>(p0, src_stride, nb, done); break;
            }
            hipLaunchKernelGGL(gfft_reduce_kernel, dim3(1, (unsigned)nch), dim3(NT), 0, stream, mag, nb, N, band, avg, meter);
        }
        count += nblk;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }

    int run_blocks(const double2 *src, long long src_stride, long long off, int nblk)
    {
        if (blue) return run_blocks_blue(src, src_stride, off, nblk);
        // enough workgroups to fill the chip (>= 1024) but no more block ranges than blocks
        int nsplit = (1024 + R * nch - 1) / (R * nch);
        if (nsplit > nblk) nsplit = nblk;
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
        const double2 *p = src + off;
        switch (M) {
        case 1024: launch<1024>(p, src_stride, nblk, nsplit); break;
        case 2048: launch<2048>(p, src_stride, nblk, nsplit); break;
        default:   launch<4096>(p, src_stride, nblk, nsplit); break;
        }
        hipLaunchKernelGGL(pan_reduce_kernel, dim3((unsigned)((N + NT - 1) / NT), (unsigned)nch), dim3(NT), 0, stream, partial, partial_m2,
                           nsplit, N, R, avg, meter, (M == 4096 && R == 4) ? 1 : 0);
        count += nblk;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
};

}  // namespace qh

using namespace qh;
struct qh_pan { Pan p; };

extern "C" {

qh_pan *qh_pan_create(int device, int nch, int fft_size, int data_width, double sample_rate, void *stream)
{
    int M = 0, R = 0, BM = 0;
    for (int r : { 1, 2, 4 })
        for (int m : { 4096, 2048, 1024 })
            if (!M && r * m == fft_size) { M = m; R = r; }
    if (!M && fft_size >= 16 && fft_size <= 16384 && fft_size % 2 == 0) {
        // any even size (FFT size must be an even number, quisk.py:186): Bluestein on BM = 2^k >= 2 N - 1 points
        BM = 1024;
        while (BM < 2 * fft_size - 1) BM *= 2;
        M = BM > 4096 ? 4096 : BM;
        R = BM / M;
    }
    if (nch <= 0 || data_width <= 0 || sample_rate <= 0 || !M) {
        set_error(M ? QH_ERR_INVALID : QH_ERR_UNSUPPORTED, "qh_pan_create: fft_size must be even, 16 .. 16384 (got %d)", fft_size);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_pan *h = new qh_pan();
    Pan &p = h->p;
    p.device = device; p.nch = nch; p.N = fft_size; p.M = M; p.R = R; p.data_width = data_width; p.rate = sample_rate;
    if (BM) { p.blue = true; p.BM = BM; p.BR = R; p.R = 1; }
    p.stream = (hipStream_t)stream;
    auto fail = [&](const char *what) -> qh_pan * { set_error(QH_ERR_HIP, "qh_pan_create: %s failed", what); delete h; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    if (!p.stream) { if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) return fail("stream"); p.own_stream = true; }
    // partial sums: one fft_avg image per block range of a call
    p.max_split = (1024 + R * nch - 1) / (R * nch);
    if (p.max_split < 1) p.max_split = 1;
    std::vector<cd> tw = fft_twiddle_table(M);
    p.hband.assign((size_t)nch, PanBand{ 0, 0, 0.0, 0, 0 });
    if (hipMalloc((void **)&p.tw, tw.size() * 16) != hipSuccess ||
        hipMalloc((void **)&p.avg, (size_t)nch * fft_size * 8) != hipSuccess || hipMalloc((void **)&p.meter, (size_t)nch * 8) != hipSuccess ||
        hipMalloc((void **)&p.pixels, (size_t)nch * data_width * 8) != hipSuccess || hipMalloc((void **)&p.smeter, (size_t)nch * 8) != hipSuccess ||
        hipMalloc((void **)&p.carry, (size_t)nch * fft_size * 16) != hipSuccess ||
        hipMalloc((void **)&p.partial, (size_t)p.max_split * nch * fft_size * 8) != hipSuccess ||
        hipMalloc((void **)&p.partial_m2, (size_t)p.max_split * nch * R * 8) != hipSuccess ||
        hipMalloc((void **)&p.band, (size_t)nch * sizeof(PanBand)) != hipSuccess)
        return fail("hipMalloc");
    if (hipMemcpy(p.tw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p.band, p.hband.data(), (size_t)nch * sizeof(PanBand), hipMemcpyHostToDevice) != hipSuccess ||
        qh::dev_zero(p.avg, (size_t)nch * fft_size * 8) != hipSuccess || qh::dev_zero(p.meter, (size_t)nch * 8) != hipSuccess)
        return fail("initial copies");
    if (p.blue) {
        const int N = fft_size;
        // w_n = exp(i pi n^2 / N), the phase reduced exactly: n^2 mod 2N in integers
        std::vector<cd> chirp((size_t)N), pre((size_t)N), filt((size_t)BM, cd(0.0, 0.0));
        for (int n = 0; n < N; n++) {
            const long long q = ((long long)n * n) % (2ll * N);
            const long double a = 3.14159265358979323846264338327950288L * (long double)q / (long double)N;
            chirp[(size_t)n] = cd((double)cosl(a), (double)sinl(a));
            // Hanning window of record_app (quisk.c:6008): 0.5 - 0.5 cos(2 pi n / N)
            const double win = 0.5 - 0.5 * (double)cosl(2.0L * 3.14159265358979323846264338327950288L * (long double)n / (long double)N);
            pre[(size_t)n] = std::conj(chirp[(size_t)n]) * win;
        }
        filt[0] = chirp[0];
        for (int n = 1; n < N; n++) { filt[(size_t)n] = chirp[(size_t)n]; filt[(size_t)(BM - n)] = chirp[(size_t)n]; }
        host_fft(filt, -1);
        long long per = 256ll * 1024 * 1024 / ((long long)nch * BM * 16);
        p.chunk_blk = (int)(per < 1 ? 1 : per > 64 ? 64 : per);
        if (hipMalloc((void **)&p.b_pre, (size_t)N * 16) != hipSuccess || hipMalloc((void **)&p.b_filt, (size_t)BM * 16) != hipSuccess ||
            hipMalloc((void **)&p.wa, (size_t)p.chunk_blk * nch * BM * 16) != hipSuccess ||
            hipMalloc((void **)&p.mag, (size_t)p.chunk_blk * nch * N * 8) != hipSuccess)
            return fail("hipMalloc");
        if (hipMemcpy(p.b_pre, pre.data(), (size_t)N * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p.b_filt, filt.data(), (size_t)BM * 16, hipMemcpyHostToDevice) != hipSuccess)
            return fail("initial copies");
    }
    hipError_t e = hipSuccess;
#define QH_PAN_ATTR(MM, RR) if (M == MM && R == RR) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pan_spectrum_kernel<MM, RR>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (TileFft<MM, false, double2>::kLdsBytes + MM * 8))
    QH_PAN_ATTR(1024, 1); QH_PAN_ATTR(1024, 2); QH_PAN_ATTR(1024, 4);
    QH_PAN_ATTR(2048, 1); QH_PAN_ATTR(2048, 2); QH_PAN_ATTR(2048, 4);
    QH_PAN_ATTR(4096, 1); QH_PAN_ATTR(4096, 2); QH_PAN_ATTR(4096, 4);
#undef QH_PAN_ATTR
    if (e == hipSuccess && M == 4096 && R == 4)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pan16k_kernel<QH_PAN16K_GROUPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                pan16k_lds<QH_PAN16K_GROUPS>());
    if (e != hipSuccess) return fail("hipFuncSetAttribute");
    return h;
}

void qh_pan_destroy(qh_pan *h) { delete h; }

// The S-meter passband: first bin from (rx_tune_freq + filter_start_offset), width filter_bandwidth (quisk.c:5223-5229)
int qh_pan_set_smeter_band(qh_pan *h, int ch, double f_start, double bandwidth)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    Pan &p = h->p;
    if (ch < -1 || ch >= p.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    QH_HIP(hipSetDevice(p.device));
    const double d2 = bandwidth * p.N / p.rate;
    const int i = (int)(f_start * p.N / p.rate + 0.5);
    const int n = (int)(std::floor(d2) + 0.01);
    PanBand b{ i, n, d2 - n, (i > -p.N / 2 && i + n + 1 < p.N / 2) ? 1 : 0, 0 };
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? p.nch : ch + 1); c++) p.hband[(size_t)c] = b;
    QH_HIP(hipMemcpyAsync(p.band, p.hband.data(), (size_t)p.nch * sizeof(PanBand), hipMemcpyHostToDevice, p.stream));
    QH_HIP(hipStreamSynchronize(p.stream));
    return QH_OK;
}

int qh_pan_feed(qh_pan *h, const double *d_in, long long in_stride, int n)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    if (n <= 0) return QH_OK;
    if (!d_in || in_stride < n) return set_error(QH_ERR_INVALID, "bad input");
    Pan &p = h->p;
    QH_HIP(hipSetDevice(p.device));
    const double2 *in = reinterpret_cast<const double2 *>(d_in);
    long long pos = 0;
    auto copy = [&](const double2 *src, long long ss, long long so, double2 *dst, long long ds, long long dofs, int cnt) {
        int gx = (cnt + 255) / 256; if (gx > 64) gx = 64;
        hipLaunchKernelGGL(pan_copy_kernel, dim3((unsigned)gx, (unsigned)p.nch), dim3(256), 0, p.stream, src, ss, so, dst, ds, dofs, cnt);
    };
    if (p.fill > 0) {
        const int need = p.N - p.fill;
        const int take = n < need ? n : need;
        copy(in, in_stride, 0, p.carry, p.N, p.fill, take);
        p.fill += take; pos = take;
        if (p.fill == p.N) {
            if (int rc = p.run_blocks(p.carry, p.N, 0, 1)) return rc;
            p.fill = 0;
        }
    }
    const int whole = (int)((n - pos) / p.N);
    if (whole > 0) {
        if (int rc = p.run_blocks(in, in_stride, pos, whole)) return rc;
        pos += (long long)whole * p.N;
    }
    const int rest = (int)(n - pos);
    if (rest > 0) { copy(in, in_stride, pos, p.carry, p.N, 0, rest); p.fill = rest; }
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_pan_count(const qh_pan *h) { return h ? h->p.count : 0; }

// ---- the decimating FIR that shares the panadapter's read (panfir16k_kernel) ----------------------------------------------------
// quisk_process_samples feeds the same cSamples to the FFT ring (quisk.c:2454-2475) and to quisk_cDecimate (filter.c:203-229).
// Shapes: fft_size 16384, decimation 32, up to 1024 real taps (BASELINE config 3: 1023 taps, /32); anything else is refused --
// qh_fir + qh_pan_feed do those.  State: the last ntaps - 1 samples of the call before.
int qh_pan_attach_fir(qh_pan *h, const double *taps, int ntaps, int decim)
{
    if (!h || !taps) return set_error(QH_ERR_INVALID, "qh_pan_attach_fir: bad arguments");
    Pan &p = h->p;
    if (p.blue || p.M != 4096 || p.R != 4 || decim != 32 || ntaps < 2 || ntaps > 1024)
        return set_error(QH_ERR_UNSUPPORTED, "qh_pan_attach_fir: fft_size 16384, decimation 32 and 2 .. 1024 taps (got %d, %d, %d): "
                         "use qh_fir beside qh_pan_feed for other shapes", p.N, decim, ntaps);
    QH_HIP(hipSetDevice(p.device));
    QH_HIP(hipStreamSynchronize(p.stream));
    const int N = p.N;
    // H'[k] = FFT_N(h)[k] exp(2 pi i (D - 1) k / N) / N, stored [r][m] for k = 4 m + r
    std::vector<cd> hp((size_t)N, cd(0, 0));
    for (int i = 0; i < ntaps; i++) hp[(size_t)i] = cd(taps[i], 0.0);
    host_fft(hp, -1);
    std::vector<cd> hr((size_t)N);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < N; k++) {
        const long double a = 2.0L * pi * (long double)(((long long)(decim - 1) * k) % N) / (long double)N;
        const cd v = hp[(size_t)k] * cd((double)cosl(a), (double)sinl(a)) / (double)N;
        hr[(size_t)((k & 3) * (N / 4) + (k >> 2))] = v;
    }
    const int hl = ntaps - 1;
    if (!p.fir_H) {
        QH_HIP(hipMalloc((void **)&p.fir_H, (size_t)N * 16));
        QH_HIP(hipMalloc((void **)&p.fir_taps, 1024 * 8));
        for (auto &q : p.fir_hist) QH_HIP(hipMalloc((void **)&q, (size_t)p.nch * 1024 * 16));
        const std::vector<cd> tw = fft_twiddle_table(512);
        QH_HIP(hipMalloc((void **)&p.fir_tw512, tw.size() * 16));
        QH_HIP(hipMemcpy(p.fir_tw512, tw.data(), tw.size() * 16, hipMemcpyHostToDevice));
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panfir16k_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, panfir16k_lds()));
    }
    QH_HIP(hipMemcpy(p.fir_H, hr.data(), (size_t)N * 16, hipMemcpyHostToDevice));
    QH_HIP(hipMemcpy(p.fir_taps, taps, (size_t)ntaps * 8, hipMemcpyHostToDevice));
    for (auto &q : p.fir_hist) QH_HIP(qh::dev_zero(q, (size_t)p.nch * 1024 * 16));
    p.fir_ntaps = ntaps; p.fir_decim = decim; p.fir_hl = hl; p.fir_cur = 0;
    return QH_OK;
}

// n (a multiple of fft_size, the panadapter at a block boundary) samples per channel: every block goes through the panadapter like
// qh_pan_feed AND comes out decimated, n / decim samples per channel, as quisk_cDecimate(..., decim) would leave them.
int qh_pan_feed_decimate(qh_pan *h, const double *d_in, long long in_stride, int n, double *d_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    if (n_out) *n_out = 0;
    Pan &p = h->p;
    if (!p.fir_H) return set_error(QH_ERR_INVALID, "qh_pan_feed_decimate: no filter attached (qh_pan_attach_fir)");
    if (n <= 0) return QH_OK;
    if (!d_in || !d_out || in_stride < n || p.fill != 0 || n % p.N) return set_error(QH_ERR_INVALID, "qh_pan_feed_decimate: whole blocks of %d samples at a block boundary", p.N);
    const int nblk = n / p.N, nout = n / p.fir_decim;
    if (out_stride < nout) return set_error(QH_ERR_INVALID, "qh_pan_feed_decimate: output stride shorter than %d samples", nout);
    QH_HIP(hipSetDevice(p.device));
    const long long need = (long long)p.nch * nblk * 512;
    if (need > p.fir_yf_cap) {
        QH_HIP(hipStreamSynchronize(p.stream));
        (void)hipFree(p.fir_yf); p.fir_yf = nullptr; p.fir_yf_cap = 0;
        QH_HIP(hipMalloc((void **)&p.fir_yf, (size_t)need * 16));
        p.fir_yf_cap = need;
    }
    const double2 *in = reinterpret_cast<const double2 *>(d_in);
    int nsplit = (1024 + p.R * p.nch - 1) / (p.R * p.nch);
    if (nsplit > nblk) nsplit = nblk;
    if (nsplit > p.max_split) nsplit = p.max_split;
    if (nsplit < 1) nsplit = 1;
    const int units = nsplit * p.nch, groups = (units + 7) / 8;
    hipLaunchKernelGGL((panfir16k_kernel<0>), dim3((unsigned)(groups * 8)), dim3(1024), (size_t)panfir16k_lds(), p.stream, in, in_stride, nblk, nsplit,
                       p.tw, p.partial, p.partial_m2, p.band, p.nch, (const double2 *)p.fir_H, p.fir_yf);
    p.count += nblk;
    PanfirTail a;
    a.yfold = (const double2 *)p.fir_yf; a.tw512 = (const double2 *)p.fir_tw512; a.in = in; a.hist = (const double2 *)p.fir_hist[p.fir_cur];
    a.taps = (const double *)p.fir_taps; a.out = reinterpret_cast<double2 *>(d_out); a.hist_next = p.fir_hist[p.fir_cur ^ 1];
    a.partial = p.partial; a.partial_m2 = p.partial_m2; a.avg = p.avg; a.meter = p.meter;
    a.in_stride = in_stride; a.out_stride = out_stride; a.nblk = nblk; a.ntaps = p.fir_ntaps; a.D = p.fir_decim; a.hl = p.fir_hl;
    a.nsplit = nsplit; a.N = p.N; a.R = p.R; a.n = n;
    hipLaunchKernelGGL(panfir_tail_kernel, dim3((unsigned)(nblk + (p.N + NT - 1) / NT + (p.fir_hl + NT - 1) / NT), (unsigned)p.nch), dim3(NT), 0, p.stream, a);
    p.fir_cur ^= 1;
    QH_HIP(hipGetLastError());
    if (n_out) *n_out = nout;
    return QH_OK;
}

// Another window in place of record_app's Hanning (measure_freq multiplies by 0.5 - 0.5 cos(2 pi i / (N - 1)), quisk.c:5600-5601).
// Sizes that run Bluestein's transform only: there the window is a table (folded into the pre-chirp); the power-of-two kernels
// form record_app's window from their twiddles.
int qh_pan_set_window(qh_pan *h, const double *window)
{
    if (!h || !window) return set_error(QH_ERR_INVALID, "qh_pan_set_window: bad arguments");
    Pan &p = h->p;
    if (!p.blue) return set_error(QH_ERR_UNSUPPORTED, "qh_pan_set_window: fft_size %d uses the fused power-of-two kernels (fixed Hanning window)", p.N);
    QH_HIP(hipSetDevice(p.device));
    const int N = p.N;
    std::vector<cd> pre((size_t)N);
    for (int n = 0; n < N; n++) {
        const long long q = ((long long)n * n) % (2ll * N);
        const long double a = 3.14159265358979323846264338327950288L * (long double)q / (long double)N;
        pre[(size_t)n] = cd((double)cosl(a), -(double)sinl(a)) * window[n];
    }
    QH_HIP(hipStreamSynchronize(p.stream));
    QH_HIP(hipMemcpy(p.b_pre, pre.data(), (size_t)N * 16, hipMemcpyHostToDevice));
    return QH_OK;
}

// The running sums of |X| in fftshift order (fft_avg of get_graph, fft_average of measure_freq), [nch][fft_size]; reset != 0
// starts the average over like a get_graph call does.
int qh_pan_read_avg(qh_pan *h, double *h_avg, int reset)
{
    if (!h || !h_avg) return set_error(QH_ERR_INVALID, "qh_pan_read_avg: bad arguments");
    Pan &p = h->p;
    QH_HIP(hipSetDevice(p.device));
    QH_HIP(hipMemcpyAsync(h_avg, p.avg, (size_t)p.nch * p.N * 8, hipMemcpyDeviceToHost, p.stream));
    if (reset) {
        QH_HIP(hipMemsetAsync(p.avg, 0, (size_t)p.nch * p.N * 8, p.stream));
        QH_HIP(hipMemsetAsync(p.meter, 0, (size_t)p.nch * 8, p.stream));
        p.count = 0;
    }
    QH_HIP(hipStreamSynchronize(p.stream));
    return QH_OK;
}
// forgets a partly filled block (measure_freq drops the rest of the call that completes a transform, quisk.c:5607-5610)
int qh_pan_drop_partial(qh_pan *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    h->p.fill = 0;
    return QH_OK;
}

int qh_pan_graph(qh_pan *h, double zoom, double deltaf, double *h_pixels, double *h_smeter, int *count)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    Pan &p = h->p;
    if (count) *count = p.count;
    if (p.count <= 0) return QH_OK;                             // get_graph returns None until an FFT has run
    QH_HIP(hipSetDevice(p.device));
    hipLaunchKernelGGL(pan_graph_kernel, dim3((unsigned)((p.nch + 63) / 64)), dim3(64), 0, p.stream, p.avg, p.meter, p.nch, p.N,
                       p.data_width, p.rate, zoom, deltaf, p.count, p.pixels, p.smeter);
    if (h_pixels) QH_HIP(hipMemcpyAsync(h_pixels, p.pixels, (size_t)p.nch * p.data_width * 8, hipMemcpyDeviceToHost, p.stream));
    if (h_smeter) QH_HIP(hipMemcpyAsync(h_smeter, p.smeter, (size_t)p.nch * 8, hipMemcpyDeviceToHost, p.stream));
    QH_HIP(hipStreamSynchronize(p.stream));
    p.count = 0;
    return QH_OK;
}

static int watfall_launch(int device, hipStream_t s, const double *d_db, long long db_stride, int nrows, int size, int width,
                          const unsigned char *palette, int y_zero, int y_scale, double gain, unsigned char *h_rgb)
{
    unsigned char *d_pal = nullptr, *d_rgb = nullptr;
    QH_HIP(hipSetDevice(device));
    QH_HIP(hipMalloc((void **)&d_pal, 768));
    if (hipMalloc((void **)&d_rgb, (size_t)nrows * width * 3) != hipSuccess) { (void)hipFree(d_pal); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(d_pal, palette, 768, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        int gx = (width + NT - 1) / NT;
        hipLaunchKernelGGL(watfall_row_kernel, dim3((unsigned)gx, (unsigned)nrows), dim3(NT), 0, s, d_db, db_stride, size < width ? size : width,
                           width, d_pal, y_zero, y_scale, gain, d_rgb);
        e = hipMemcpyAsync(h_rgb, d_rgb, (size_t)nrows * width * 3, hipMemcpyDeviceToHost, s);
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(d_pal); (void)hipFree(d_rgb);
    if (e != hipSuccess || e2 != hipSuccess) return set_error(QH_ERR_HIP, "waterfall row failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return QH_OK;
}

// get_graph + watfall_OnGraphData in one call: the dB row never leaves the device
int qh_pan_waterfall(qh_pan *h, double zoom, double deltaf, const unsigned char *red, const unsigned char *green,
                     const unsigned char *blue, int y_zero, int y_scale, double gain, int width, unsigned char *h_rgb,
                     double *h_smeter, int *count)
{
    if (!h || !red || !green || !blue || !h_rgb || width <= 0) return set_error(QH_ERR_INVALID, "qh_pan_waterfall: bad arguments");
    Pan &p = h->p;
    if (count) *count = p.count;
    if (p.count <= 0) return QH_OK;
    QH_HIP(hipSetDevice(p.device));
    hipLaunchKernelGGL(pan_graph_kernel, dim3((unsigned)((p.nch + 63) / 64)), dim3(64), 0, p.stream, p.avg, p.meter, p.nch, p.N,
                       p.data_width, p.rate, zoom, deltaf, p.count, p.pixels, p.smeter);
    if (h_smeter) QH_HIP(hipMemcpyAsync(h_smeter, p.smeter, (size_t)p.nch * 8, hipMemcpyDeviceToHost, p.stream));
    p.count = 0;
    unsigned char pal[768];
    std::memcpy(pal, red, 256); std::memcpy(pal + 256, green, 256); std::memcpy(pal + 512, blue, 256);
    return watfall_launch(p.device, p.stream, p.pixels, p.data_width, p.nch, p.data_width, width, pal, y_zero, y_scale, gain, h_rgb);
}

// watfall_OnGraphData for `nrows` dB rows that are already on the host (h_db [nrows][ncols])
int qh_watfall_rows_host(int device, const double *h_db, int nrows, int ncols, const unsigned char *red, const unsigned char *green,
                         const unsigned char *blue, int y_zero, int y_scale, double gain, int width, unsigned char *h_rgb)
{
    if (!h_db || !red || !green || !blue || !h_rgb || nrows <= 0 || ncols < 0 || width <= 0)
        return set_error(QH_ERR_INVALID, "qh_watfall_rows_host: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    QH_HIP(hipSetDevice(device));
    double *d_db = nullptr;
    QH_HIP(hipMalloc((void **)&d_db, (size_t)nrows * (ncols > 0 ? ncols : 1) * 8));
    if (ncols > 0 && hipMemcpy(d_db, h_db, (size_t)nrows * ncols * 8, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d_db);
        return set_error(QH_ERR_HIP, "upload failed");
    }
    unsigned char pal[768];
    std::memcpy(pal, red, 256); std::memcpy(pal + 256, green, 256); std::memcpy(pal + 512, blue, 256);
    const int rc = watfall_launch(device, nullptr, d_db, ncols, nrows, ncols, width, pal, y_zero, y_scale, gain, h_rgb);
    (void)hipFree(d_db);
    return rc;
}

int qh_pan_feed_host(qh_pan *h, const double *h_in, long long in_stride, int n)
{
    if (!h) return set_error(QH_ERR_INVALID, "null panadapter");
    if (n <= 0) return QH_OK;
    Pan &p = h->p;
    QH_HIP(hipSetDevice(p.device));
    double2 *d = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)p.nch * n * 16));
    hipError_t e = hipMemcpy2DAsync(d, (size_t)n * 16, h_in, (size_t)in_stride * 16, (size_t)n * 16, (size_t)p.nch, hipMemcpyHostToDevice, p.stream);
    int rc = e == hipSuccess ? qh_pan_feed(h, reinterpret_cast<const double *>(d), n, n) : QH_ERR_HIP;
    hipError_t e2 = hipStreamSynchronize(p.stream);
    (void)hipFree(d);
    if (rc) return rc == QH_ERR_HIP ? set_error(QH_ERR_HIP, "qh_pan_feed_host: copy failed") : rc;
    if (e2 != hipSuccess) return set_error(QH_ERR_HIP, "qh_pan_feed_host: synchronize failed");
    return QH_OK;
}

// ---- bandscope: qh_bscope_* -------------------------------------------------------------------------------
struct qh_bscope {
    qh_pan *pan = nullptr;
    int nch = 0, N = 0, width = 0;
    unsigned long long *maxbits = nullptr;
    double2 *cbuf = nullptr;
    long long ccap = 0;
    ~qh_bscope()
    {
        if (pan) {
            (void)hipSetDevice(pan->p.device);
            (void)hipStreamSynchronize(pan->p.stream);
            (void)hipFree(maxbits); (void)hipFree(cbuf);
            qh_pan_destroy(pan);
        }
    }
};

qh_bscope *qh_bscope_create(int device, int nch, int bandscope_size, int graph_width, void *stream)
{
    qh_pan *pan = qh_pan_create(device, nch, bandscope_size, graph_width, 1.0, stream);
    if (!pan) return nullptr;
    qh_bscope *b = new qh_bscope();
    b->pan = pan; b->nch = nch; b->N = bandscope_size; b->width = graph_width;
    if (hipMalloc((void **)&b->maxbits, (size_t)nch * 8) != hipSuccess ||
        hipMemsetAsync(b->maxbits, 0, (size_t)nch * 8, pan->p.stream) != hipSuccess) {
        set_error(QH_ERR_HIP, "qh_bscope_create: allocation failed");
        delete b;
        return nullptr;
    }
    return b;
}

void qh_bscope_destroy(qh_bscope *b) { delete b; }
int qh_bscope_count(const qh_bscope *b) { return b ? b->pan->p.count : 0; }

// n real samples per channel, device pointer [nch][in_stride] doubles, already divided by bandscopeScale (quisk.c:3596)
int qh_bscope_feed(qh_bscope *b, const double *d_in, long long in_stride, int n)
{
    if (!b) return set_error(QH_ERR_INVALID, "null bandscope");
    if (n <= 0) return QH_OK;
    if (!d_in || in_stride < n) return set_error(QH_ERR_INVALID, "bad input");
    Pan &p = b->pan->p;
    QH_HIP(hipSetDevice(p.device));
    if (n > b->ccap) {
        QH_HIP(hipStreamSynchronize(p.stream));
        (void)hipFree(b->cbuf); b->cbuf = nullptr;
        QH_HIP(hipMalloc((void **)&b->cbuf, (size_t)b->nch * (size_t)n * 16));
        b->ccap = n;
    }
    int gx = (n + NT - 1) / NT;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(bscope_convert_kernel, dim3((unsigned)gx, (unsigned)b->nch), dim3(NT), 0, p.stream, d_in, in_stride, n, b->cbuf, b->maxbits);
    return qh_pan_feed(b->pan, reinterpret_cast<const double *>(b->cbuf), n, n);
}

int qh_bscope_feed_host(qh_bscope *b, const double *h_in, long long in_stride, int n)
{
    if (!b) return set_error(QH_ERR_INVALID, "null bandscope");
    if (n <= 0) return QH_OK;
    if (!h_in || in_stride < n) return set_error(QH_ERR_INVALID, "bad input");
    Pan &p = b->pan->p;
    QH_HIP(hipSetDevice(p.device));
    double *d = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)b->nch * (size_t)n * 8));
    int rc = QH_OK;
    if (hipMemcpy2DAsync(d, (size_t)n * 8, h_in, (size_t)in_stride * 8, (size_t)n * 8, (size_t)b->nch, hipMemcpyHostToDevice, p.stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "qh_bscope_feed_host: copy failed");
    if (rc == QH_OK) rc = qh_bscope_feed(b, d, n, n);
    if (hipStreamSynchronize(p.stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "qh_bscope_feed_host: synchronize failed");
    (void)hipFree(d);
    return rc;
}

// The refresh branch of get_bandscope(clock, zoom, deltaf): h_pixels [nch][graph_width] dB, h_adc_level [nch] = the largest
// |sample| since the last call (hermes_adc_level); *count = blocks averaged (0: nothing written, like returning None)
int qh_bscope_graph(qh_bscope *b, int clock, double zoom, double deltaf, double *h_pixels, double *h_adc_level, int *count)
{
    if (!b || clock <= 0) return set_error(QH_ERR_INVALID, "qh_bscope_graph: bad arguments");
    Pan &p = b->pan->p;
    if (count) *count = p.count;
    if (p.count <= 0) return QH_OK;
    QH_HIP(hipSetDevice(p.device));
    hipLaunchKernelGGL(bscope_graph_kernel, dim3((unsigned)((b->width + NT - 1) / NT), (unsigned)b->nch), dim3(NT), 0, p.stream, p.avg, p.N,
                       b->width, clock / 2.0, zoom, deltaf, p.count, p.pixels);
    if (h_pixels) QH_HIP(hipMemcpyAsync(h_pixels, p.pixels, (size_t)b->nch * b->width * 8, hipMemcpyDeviceToHost, p.stream));
    if (h_adc_level) QH_HIP(hipMemcpyAsync(h_adc_level, b->maxbits, (size_t)b->nch * 8, hipMemcpyDeviceToHost, p.stream));
    QH_HIP(hipMemsetAsync(p.avg, 0, (size_t)b->nch * p.N * 8, p.stream));
    QH_HIP(hipMemsetAsync(p.meter, 0, (size_t)b->nch * 8, p.stream));
    QH_HIP(hipMemsetAsync(b->maxbits, 0, (size_t)b->nch * 8, p.stream));
    QH_HIP(hipStreamSynchronize(p.stream));
    p.count = 0;
    return QH_OK;
}

}  // extern "C"

#ifdef QH_PAN_TRACE
extern "C" int qh_dbg_pan_trace(unsigned long long *out, int n)
{
    if (n > 64 * 8) n = 64 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(qh::qh_pan_trace), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
