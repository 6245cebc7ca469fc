// qh_fft.hpp -- fp64 / fp32 complex FFT building blocks for gfx950 (CDNA4), LDS resident.
//
// Replaces the FFTW3 calls of the reference hot path (fftw_plan_dft_1d + fftw_execute at
// wdsp/firmin.c:313-318,412,428 and quisk.c:5215,5999): unnormalised DFT, sign -1 forward.
//
// Design (MI355X): one workgroup of NT = 256 threads (4 wavefronts of 64) transforms N points
// held in LDS as interleaved complex (ds_read_b128 / ds_write_b128 per element).  Stockham
// autosort passes with large radices (16 for N = 4096) keep the number of LDS round trips at
// log16(N) - 1; the first pass takes its inputs from registers and the last pass leaves its
// outputs in registers, both in the "strided register layout"
//        thread t holds elements  t + NT * i,   i = 0 .. N/NT - 1,
// so that global loads/stores are coalesced (consecutive lanes, 16 B each) and so that a
// frequency-domain mask multiply, a spectral fold (decimation) and the first inverse pass
// need no LDS traffic at all.  The LDS image is padded (row pitch 17 elements, lds_phys below).
#pragma once
#include <hip/hip_runtime.h>

namespace qh {

// angle arrays (turns, (-0.5, 0.5]) mark an all-zero sample with this value (pll_theta_kernel, THETA stores of osfir_kernel)
static constexpr double kThetaZeroMark = 8.0;


constexpr int NT = 256;     // threads per workgroup for every kernel in this file

template <typename T> struct cplx_of;
template <> struct cplx_of<double> { using type = double2; };
template <> struct cplx_of<float>  { using type = float2; };
template <typename T> using cplx = typename cplx_of<T>::type;

template <typename T> __device__ __forceinline__ cplx<T> mk(T x, T y) { cplx<T> r; r.x = x; r.y = y; return r; }
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { C r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { C r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b)
{
    C r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store the wavefront has in
// flight (s_waitcnt vmcnt(0)) -- a kernel that asks for the next block's samples early, or has stores on the way, stalls at each one.  Only
// for kernels whose wavefronts do not hand each other data through global memory.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename C> __device__ __forceinline__ C cconj(C a) { C r; r.x = a.x; r.y = -a.y; return r; }
// a * (-i) (forward) or a * (+i) (inverse)
template <bool INV, typename C> __device__ __forceinline__ C mul_mi(C a)
{
    C r;
    if (INV) { r.x = -a.y; r.y = a.x; } else { r.x = a.y; r.y = -a.x; }
    return r;
}

// cos(2*pi*a/32), a = 0..8
__device__ __forceinline__ constexpr double cos32(int a)
{
    constexpr double t[9] = { 1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                              0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                              0.19509032201612826785, 0.0 };
    return t[a];
}

// multiply by W_R^i = exp(-+ 2*pi*i*I/R), I a compile-time index in [0, R/2)
template <int R, int I, bool INV, typename C> __device__ __forceinline__ C mul_wconst(C a)
{
    using T = decltype(a.x);
    constexpr int A = I * (32 / R);            // angle in units of 2*pi/32, 0 <= A < 16
    if constexpr (A == 0) {
        return a;
    } else if constexpr (A == 8) {
        return mul_mi<INV>(a);
    } else if constexpr (A == 4) {             // (1 -+ i)/sqrt2
        const T s = (T)0.70710678118654752440;
        C r;
        if (INV) { r.x = (a.x - a.y) * s; r.y = (a.x + a.y) * s; }
        else     { r.x = (a.x + a.y) * s; r.y = (a.y - a.x) * s; }
        return r;
    } else if constexpr (A == 12) {            // (-1 -+ i)/sqrt2
        const T s = (T)0.70710678118654752440;
        C r;
        if (INV) { r.x = -(a.x + a.y) * s; r.y = (a.x - a.y) * s; }
        else     { r.x = (a.y - a.x) * s;  r.y = -(a.x + a.y) * s; }
        return r;
    } else {
        constexpr double c = (A <= 8) ? cos32(A) : -cos32(16 - A);
        constexpr double s = (A <= 8) ? cos32(8 - A) : cos32(A - 8);     // sin(2*pi*A/32) >= 0
        C w;
        w.x = (T)c;
        w.y = INV ? (T)s : (T)(-s);
        return cmul(a, w);
    }
}

// In-register DFT of R points (R = 1, 2, 4, 8, 16, 32), natural order in and out.
// Radix-2 decimation-in-frequency recursion, fully unrolled; twiddles are literals.
template <int R, bool INV, typename C> struct Dft {
    template <int I> static __device__ __forceinline__ void stage(C (&x)[R], C (&a)[R / 2], C (&b)[R / 2])
    {
        if constexpr (I < R / 2) {
            a[I] = cadd(x[I], x[I + R / 2]);
            b[I] = mul_wconst<R, I, INV>(csub(x[I], x[I + R / 2]));
            stage<I + 1>(x, a, b);
        }
    }
    static __device__ __forceinline__ void run(C (&x)[R])
    {
        C a[R / 2], b[R / 2];
        stage<0>(x, a, b);
        Dft<R / 2, INV, C>::run(a);
        Dft<R / 2, INV, C>::run(b);
#pragma unroll
        for (int q = 0; q < R / 2; q++) { x[2 * q] = a[q]; x[2 * q + 1] = b[q]; }
    }
};
template <bool INV, typename C> struct Dft<1, INV, C> { static __device__ __forceinline__ void run(C (&)[1]) {} };

// LDS image: element i lives at i + (i >> 4) (one 16-byte pad after every 16 elements = row pitch 17).
// The stride-R scatter of the early passes then hits distinct banks (ds_write_b128 is served 8 lanes at a
// time over 32 banks; pitch 17 elements = 68 dwords = 4 mod 32), and every access of a pass is
// "lane base + compile-time offset", so the address arithmetic is one value per pass, not one per access.
__device__ __forceinline__ int lds_phys(int i) { return i + (i >> 4); }
template <int N> constexpr int lds_elems() { return N + N / 16; }

// Powers w^1 .. w^(R-1) applied to x[1..R-1]; products formed by squaring/multiplying with
// depth <= log2(R) so the rounding error stays at a few ulp.
template <int R, typename C> __device__ __forceinline__ void apply_twiddle_powers(C (&x)[R], C w1)
{
    // w^r = w^hi * w^lo (hi = top set bit of r, w^hi by squaring).  Powers below R/2 are kept because the upper
    // half needs them as w^lo; those of the upper half are used once and dropped, so R/2 + 1 powers are live at
    // most, not R - 1 (the difference is what lets the decimating fp64 kernels fit 128 registers).
    constexpr int H = R / 2 > 1 ? R / 2 : 1;
    C w[H + 1];
    w[1] = w1;
    x[1] = cmul(x[1], w1);
#pragma unroll
    for (int r = 2; r < R; r++) {
        int hi = 1;
        while (hi * 2 <= r) hi *= 2;
        const int lo = r - hi;
        C wr;
        if (hi < H || (hi == H && lo == 0)) {
            wr = (lo == 0) ? cmul(w[hi / 2], w[hi / 2]) : cmul(w[hi], w[lo]);
            w[r < H + 1 ? r : H] = wr;
        } else {
            wr = cmul(w[H], w[lo]);         // hi == H, lo > 0: upper half
        }
        x[r] = cmul(x[r], wr);
    }
}

// The same with three powers live instead of R/2 + 1: w^2 by squaring, then the even and the odd powers as two chains w^(r+2) = w^r w^2
// (depth R/2, i.e. about R/2 roundings in the last power instead of log2 R -- 8e-16 for R = 16).  For kernels whose registers are full
// of something else (pan16k_kernel: the |X| sums of a block range ride through the transform): 28 registers fewer for R = 16.
template <int R, typename C> __device__ __forceinline__ void apply_twiddle_powers_lean(C (&x)[R], C w1)
{
    const C w2 = cmul(w1, w1);
    C we = w2, wo = w1;
    x[1] = cmul(x[1], wo);
#pragma unroll
    for (int r = 2; r < R; r++) {
        if (r & 1) { wo = cmul(wo, w2); x[r] = cmul(x[r], wo); }
        else { x[r] = cmul(x[r], we); if (r + 2 < R) we = cmul(we, w2); }
    }
}

// One Stockham pass of radix R over N points for butterfly j, data in registers x[r] = in[j + r*N/R].
// Applies the inter-pass twiddle w1 = exp(-+2*pi*i*k/(Ns*R)), k = j mod Ns (Ns = product of the radices of
// the earlier passes), then the R-point DFT, and returns the output position of x[0]; x[r] belongs at
// base + r*Ns.  w1 depends on the lane only, so callers load it once per kernel (pass_twiddle below).
template <int N, int R, int Ns, bool INV, typename C, bool LEAN = false>
__device__ __forceinline__ int stockham_butterfly(C (&x)[R], int j, C w1)
{
    int k = j & (Ns - 1);
    if constexpr (Ns > 1) { if constexpr (LEAN) apply_twiddle_powers_lean<R>(x, w1); else apply_twiddle_powers<R>(x, w1); }
    Dft<R, INV, C>::run(x);
    return (j - k) * R + k;
}

// the lane's twiddle for butterfly j of a pass whose table (Ns entries) starts at tw_pass
template <int Ns, bool INV, typename C>
__device__ __forceinline__ C pass_twiddle(const C *__restrict__ tw_pass, int j)
{
    C w = tw_pass[j & (Ns - 1)];
    return INV ? cconj(w) : w;
}

// pass: registers (strided layout, requires R == N/NT, Ns == 1) -> LDS
template <int N, int R, bool INV, typename C>
__device__ __forceinline__ void pass_regs_to_lds(C (&x)[R], C *lds)
{
    static_assert(R == N / NT, "first pass radix must equal N/NT");
    const int j = threadIdx.x;
    Dft<R, INV, C>::run(x);
    if constexpr (16 % R == 0) {
        C *p = lds + lds_phys(j * R);       // (j*R + r) >> 4 == (j*R) >> 4 for r < R when R divides 16
#pragma unroll
        for (int r = 0; r < R; r++) p[r] = x[r];
    } else {
        static_assert(R % 16 == 0, "first pass radix: a divisor or a multiple of 16");
        C *p = lds + lds_phys(j * R);       // rows of 16 elements, one pad element between them
#pragma unroll
        for (int r = 0; r < R; r++) p[r + r / 16] = x[r];
    }
}

template <int N, int R> struct PassGeom {
    static constexpr int NB = N / R;                    // butterflies in the pass
    static constexpr int PER = (NB + NT - 1) / NT;      // per thread
};

// pass: LDS -> LDS (in place; barrier between the read and the write phase)
template <int N, int R, int Ns, bool INV, typename C>
__device__ __forceinline__ void pass_lds_to_lds(C *lds, const C (&w1)[PassGeom<N, R>::PER])
{
    constexpr int NB = PassGeom<N, R>::NB;
    constexpr int PER = PassGeom<N, R>::PER;
    C x[PER][R];
    int base[PER];
    static_assert(NB % 16 == 0, "pass geometry");
#pragma unroll
    for (int p = 0; p < PER; p++) {
        int j = threadIdx.x + p * NT;
        if (NB >= NT || j < NB) {
            const C *q = lds + lds_phys(j);
#pragma unroll
            for (int r = 0; r < R; r++) x[p][r] = q[r * (NB + NB / 16)];
        }
    }
#pragma unroll
    for (int p = 0; p < PER; p++) {
        int j = threadIdx.x + p * NT;
        if (NB >= NT || j < NB) base[p] = stockham_butterfly<N, R, Ns, INV>(x[p], j, w1[p]);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PER; p++) {
        int j = threadIdx.x + p * NT;
        if (NB >= NT || j < NB) {
            if constexpr (Ns % 16 == 0) {
                C *q = lds + lds_phys(base[p]);
#pragma unroll
                for (int r = 0; r < R; r++) q[r * (Ns + Ns / 16)] = x[p][r];
            } else if constexpr (16 % (Ns * R) == 0) {
                C *q = lds + lds_phys(base[p]);     // the R outputs stay inside one 16-element row
#pragma unroll
                for (int r = 0; r < R; r++) q[r * Ns] = x[p][r];
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) lds[lds_phys(base[p] + r * Ns)] = x[p][r];
            }
        }
    }
}

// pass: LDS -> registers (strided layout; requires R == N/NT and Ns == N/R, the last pass)
template <int N, int R, bool INV, typename C>
__device__ __forceinline__ void pass_lds_to_regs(const C *lds, C (&x)[R], C w1)
{
    static_assert(R == N / NT, "last pass radix must equal N/NT");
    const int j = threadIdx.x;
    const C *q = lds + lds_phys(j);
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = q[r * (NT + NT / 16)];
    stockham_butterfly<N, R, N / R, INV>(x, j, w1);     // output of x[r] is element j + r*NT
}

// ---------------------------------------------------------------------------------------------
// FFT of N points: registers (strided layout) -> registers (strided layout), through LDS.
//   first(x, lds)      pass 1: consumes x (dead afterwards), leaves the data in LDS
//   rest(lds, x, tw)   barrier + remaining passes; result in x
// Radix plans (first == last == N/NT).  Twiddle tables: one per pass with Ns > 1, concatenated; entry k
// of a pass is exp(-2*pi*i*k/(Ns*R)).  Offsets must match qh::fft_twiddle_table() in qh_design.cpp.
//      N      radices        Ns per pass        table offsets
//      4096   16,16,16       1,16,256           [0,16) [16,272)
//      2048   8,4,8,8        1,8,32,256         [0,8) [8,40) [40,296)
//      1024   4,4,4,4,4      1,4,16,64,256      [0,4) [4,20) [20,84) [84,340)
//      512    2,16,8,2       1,2,32,256         [0,2) [2,34) [34,290)
//      8192   32,8,32        1,32,256           [0,32) [32,288)
template <int N, bool INV, typename C> struct FftRR;

template <bool INV, typename C> struct FftRR<4096, INV, C> {
    struct Tw { C a[1], b; };
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw) { return load_at(tw, threadIdx.x); }
    // j = index of the thread inside its 256-thread transform group (workgroups that run several transforms side by side)
    static __device__ __forceinline__ Tw load_at(const C *__restrict__ tw, int j)
    {
        Tw t;
        t.a[0] = pass_twiddle<16, INV>(tw, j);
        t.b = pass_twiddle<256, INV>(tw + 16, j);
        return t;
    }
    static __device__ __forceinline__ void first(C (&x)[16], C *lds) { pass_regs_to_lds<4096, 16, INV>(x, lds); }
    static __device__ __forceinline__ void rest(C *lds, C (&x)[16], const Tw &t)
    {
        __syncthreads();
        pass_lds_to_lds<4096, 16, 16, INV>(lds, t.a);
        __syncthreads();
        pass_lds_to_regs<4096, 16, INV>(lds, x, t.b);
    }
};

// The same 4096-point transform with the real and the imaginary parts exchanged through LDS one after the other:
// the LDS image holds 4096 scalars (34.8 KB in fp64 instead of 69.6 KB), so four workgroups fit a CU where two did,
// at the price of barriers (7 per transform instead of 2).  Registers -> registers; the caller guarantees that
// nobody still reads the LDS image when run() starts.
#ifndef QH_SPLIT_PAD
#define QH_SPLIT_PAD 2
#endif
// QH_SPLIT_SWIZZLE: no padding at all -- the image is exactly 4096 scalars = 32 KB in fp64, so FIVE workgroups fit the 160 KB
// of a CU -- and bank conflicts are avoided by an XOR swizzle instead: element i lives at (i & ~15) | ((i ^ (i >> 4)) & 15).
// A lane's 16 contiguous scalars of the first exchange (row j, column r) land in column r ^ (j & 15): one store instruction
// (fixed r) spreads 16 neighbouring lanes over the 16 bank pairs; the second exchange's stores (row 16 u + r, column k) land
// in column k ^ r: again all 16; every load is element j + 256 r = row 16 r + (j >> 4), column (j & 15) ^ ((j >> 4) & 15):
// lane base + compile-time offset, 32 consecutive lanes on every bank pair exactly twice (the floor for 64-bit accesses).
// MEASURED (profiles/r02_notes.md): correct, conflict free -- and no faster.  The per-access XOR costs the front kernel 24
// VGPRs (98 -> 122, so still four wavefronts per SIMD, not five; capped at 102 it spills 35 and takes 7.5 ms) and the band
// kernel its last free registers (14 spilled: 3.01 -> 3.33 ms).  Off by default; the padded image below is what ships.
#ifndef QH_SPLIT_SWIZZLE
#define QH_SPLIT_SWIZZLE 0
#endif
constexpr int split4096_lds_bytes(int scalar_bytes) { return (QH_SPLIT_SWIZZLE ? 4096 : 4096 + QH_SPLIT_PAD * 256) * scalar_bytes; }
template <bool INV, typename C> struct FftSplit4096 {
    using T = decltype(C{}.x);
    using Tw = typename FftRR<4096, INV, C>::Tw;
    static constexpr bool kSwizzle = QH_SPLIT_SWIZZLE != 0;
    // PHASE 1: the lane's elements are 16 j + r; PHASE 2: base + 16 r with base = 256 u + k (j = 16 u + k)
    template <int PHASE>
    static __device__ __forceinline__ void exchange_sw(C (&x)[16], T *lds, int j, int base)
    {
        const int m = j & 15;
        T *wb = PHASE == 1 ? lds + 16 * j : lds + (base - m);              // row start of r = 0 (phase 2: base & 15 == m)
        const T *rp = lds + 16 * (j >> 4) + (m ^ ((j >> 4) & 15));
#pragma unroll
        for (int r = 0; r < 16; r++) (PHASE == 1 ? wb : wb + 16 * r)[PHASE == 1 ? (r ^ m) : (m ^ r)] = x[r].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].x = rp[256 * r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) (PHASE == 1 ? wb : wb + 16 * r)[PHASE == 1 ? (r ^ m) : (m ^ r)] = x[r].y;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].y = rp[256 * r];
    }
    // scalar image, row pitch 18: the compiler pairs the 16 contiguous scalars a lane writes in the first exchange
    // into 128-bit stores, which are served 8 lanes at a time -- a lane stride of 18 scalars = 4 banks (mod 32)
    // keeps those 8 lanes on distinct banks (pitch 17 gave 2-way conflicts on a third of the LDS cycles)
    static constexpr int kPad = QH_SPLIT_PAD;
    static constexpr int kLdsBytes = split4096_lds_bytes((int)sizeof(T));
    static __device__ __forceinline__ int sphys(int i) { return i + (i >> 4) * kPad; }

    // LB: the barriers order LDS accesses only (lds_barrier): loads from / stores to global memory stay in flight across them
    template <int WS, int RS, bool LB = false>
    static __device__ __forceinline__ void exchange(C (&x)[16], T *lds, int wbase, int rbase)
    {
        T *wp = lds + wbase;
        const T *rp = lds + rbase;
#pragma unroll
        for (int r = 0; r < 16; r++) wp[r * WS] = x[r].x;
        if (LB) lds_barrier(); else __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].x = rp[r * RS];
        if (LB) lds_barrier(); else __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) wp[r * WS] = x[r].y;
        if (LB) lds_barrier(); else __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].y = rp[r * RS];
    }

    static __device__ __forceinline__ void run(C (&x)[16], void *lds_raw, const Tw &t) { run_at(x, lds_raw, t, threadIdx.x); }
    // the same for thread j of a 256-thread group inside a larger workgroup (the barriers are the workgroup's: every group of
    // the workgroup must run its transform at the same time)
    // LEAN: the inter-pass twiddle powers as two chains (apply_twiddle_powers_lean), for callers short of registers
    template <bool LEAN = false>
    static __device__ __forceinline__ void run_at(C (&x)[16], void *lds_raw, const Tw &t, int j)
    {
        T *lds = reinterpret_cast<T *>(lds_raw);
        Dft<16, INV, C>::run(x);                                    // pass 1: x[r] = in[j + 256 r]
        // element j*16 + r  ->  j + 256 r'   (sphys: 18 j + r, and sphys(j) + 288 r')
        if constexpr (kSwizzle) exchange_sw<1>(x, lds, j, 0);
        else exchange<1, 256 + 16 * kPad>(x, lds, (16 + kPad) * j, sphys(j));
        const int base = stockham_butterfly<4096, 16, 16, INV, C, LEAN>(x, j, t.a[0]);      // pass 2: outputs at base + 16 r
        __syncthreads();
        if constexpr (kSwizzle) exchange_sw<2>(x, lds, j, base);
        else exchange<16 + kPad, 256 + 16 * kPad>(x, lds, sphys(base), sphys(j));
        stockham_butterfly<4096, 16, 256, INV, C, LEAN>(x, j, t.b);          // pass 3: x[r] is element j + 256 r
    }

    // The transform WITHOUT the stage that would combine the D decimated sequences x[D m + a] (plan 16 x 16 x 16/D x [D], the last
    // radix-D pass left out): register i + (16/D) a ends up with bin j + 256 i of the (4096/D)-point transform of phase a.  A
    // decimating overlap-save stage wants exactly that: its D-fold aliasing sum over X[k + (4096/D) q] M[k + (4096/D) q] equals
    // sum_a X_a[k] G_a[k] with G_a[k] = W_4096^(a k) sum_q W_D^(a q) M[k + (4096/D) q] -- the same number of mask products,
    // one butterfly stage less (polyphase form of the decimator; the masks are turned into G when they are built).
    template <int D>
    static __device__ __forceinline__ void run_poly(C (&x)[16], void *lds_raw, const Tw &t)
    {
        static_assert(!INV && (D == 2 || D == 4 || D == 8), "forward transform of a decimating stage");
        constexpr int R3 = 16 / D;
        const int j = threadIdx.x;
        T *lds = reinterpret_cast<T *>(lds_raw);
        Dft<16, INV, C>::run(x);
        if constexpr (kSwizzle) exchange_sw<1>(x, lds, j, 0);
        else exchange<1, 256 + 16 * kPad>(x, lds, (16 + kPad) * j, sphys(j));
        const int base = stockham_butterfly<4096, 16, 16, INV>(x, j, t.a[0]);
        __syncthreads();
        if constexpr (kSwizzle) exchange_sw<2>(x, lds, j, base);
        else exchange<16 + kPad, 256 + 16 * kPad>(x, lds, sphys(base), sphys(j));
        // pass 3 of the shorter plan: R3-point butterflies with twiddle exp(-2 pi i j r / (256 R3)) = (t.b^D)^r; butterfly a takes
        // the elements j + 256 (a + D r), i.e. registers a + D r
        C w[R3];
        w[1] = t.b;
#pragma unroll
        for (int s = 1; s < D; s <<= 1) w[1] = cmul(w[1], w[1]);
#pragma unroll
        for (int r = 2; r < R3; r++) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
        C y[16];
#pragma unroll
        for (int a = 0; a < D; a++) {
            C b[R3];
            b[0] = x[a];
#pragma unroll
            for (int r = 1; r < R3; r++) b[r] = cmul(x[a + D * r], w[r]);
            Dft<R3, INV, C>::run(b);
#pragma unroll
            for (int r = 0; r < R3; r++) y[r + R3 * a] = b[r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = y[r];
    }
};

// ---------------------------------------------------------------------------------------------
// 6144 = 16 x 24 x 16 points on 384 lanes (six wavefronts), 16 elements per lane in the strided layout  t + 384 r.
// The outer passes are the radix-16 passes of the 4096-point plan (same registers, same code); the middle pass is 256 butterflies
// of 24 points (3 x 8: three-point transforms, constant twiddles W_24^(n2 k1), eight-point transforms), run by the lanes below 256.
// A filter of nc <= 2048 taps leaves a tile 4097 useful outputs of 6144 (4096 are taken: whole 64-sample chunks) where the
// 4096-point tile leaves 2049 of 4096: about a fifth fewer flops, a quarter less LDS traffic and half the barriers per output.
// Exchange through ONE scalar image (real parts, then imaginary parts), row pitch 18 doubles per 16 elements: 55 296 bytes.
constexpr int kFft6kThreads = 384;
constexpr int kFft6kPad = 2;
constexpr int kFft6kLdsBytes = (6144 + 6144 / 16 * kFft6kPad) * 8;

template <bool INV, typename C> __device__ __forceinline__ void dft3(C &a, C &b, C &c)
{
    // y0 = a + b + c, y1 = a + w b + w^2 c, y2 = a + w^2 b + w c, w = exp(-+2 pi i / 3) = -1/2 -+ i sqrt(3)/2
    const double h = 0.86602540378443864676;
    const C t1 = cadd(b, c);
    C t2; t2.x = a.x - 0.5 * t1.x; t2.y = a.y - 0.5 * t1.y;
    const C d = csub(b, c);
    C t3;                                    // (b - c) * (-+ i h)
    if (INV) { t3.x = -h * d.y; t3.y = h * d.x; } else { t3.x = h * d.y; t3.y = -h * d.x; }
    a = cadd(a, t1);
    b = cadd(t2, t3);
    c = csub(t2, t3);
}

// multiply by W_24^m = exp(-+2 pi i m / 24), m a compile-time constant in [0, 24)
template <int M, bool INV, typename C> __device__ __forceinline__ C mul_w24(C a)
{
    constexpr int m = M % 24;
    if constexpr (m == 0) return a;
    else if constexpr (m == 6) return mul_mi<INV>(a);
    else if constexpr (m == 12) { C r; r.x = -a.x; r.y = -a.y; return r; }
    else if constexpr (m == 18) return mul_mi<!INV>(a);
    else {
        // cos / sin of 2 pi m / 24 = m * 15 degrees
        constexpr double c15[7] = { 1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5,
                                    0.25881904510252076235, 0.0 };
        constexpr int q = m / 6, rem = m % 6;           // quadrant, angle inside it
        constexpr double cr = c15[rem], sr = c15[6 - rem];
        constexpr double co = q == 0 ? cr : q == 1 ? -sr : q == 2 ? -cr : sr;
        constexpr double si = q == 0 ? sr : q == 1 ? cr : q == 2 ? -sr : -cr;
        C w; w.x = co; w.y = INV ? si : -si;
        return cmul(a, w);
    }
}

// In-register DFT of 24 points, natural order in and out: n = 8 n1 + n2, k = k1 + 3 k2
template <bool INV, typename C> struct Dft24 {
    template <int N2> static __device__ __forceinline__ void tw(C (&y)[3][8])
    {
        if constexpr (N2 < 8) {
            y[1][N2] = mul_w24<N2, INV>(y[1][N2]);
            y[2][N2] = mul_w24<2 * N2, INV>(y[2][N2]);
            tw<N2 + 1>(y);
        }
    }
    static __device__ __forceinline__ void run(C (&x)[24])
    {
        C y[3][8];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) {
            C a = x[n2], b = x[8 + n2], c = x[16 + n2];
            dft3<INV>(a, b, c);
            y[0][n2] = a; y[1][n2] = b; y[2][n2] = c;
        }
        tw<1>(y);
#pragma unroll
        for (int k1 = 0; k1 < 3; k1++) Dft<8, INV, C>::run(y[k1]);
#pragma unroll
        for (int k1 = 0; k1 < 3; k1++)
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) x[k1 + 3 * k2] = y[k1][k2];
    }
};

// x[r] *= w^r, r = 1 .. 23, powers by products of depth <= 4
template <typename C> __device__ __forceinline__ void apply_twiddle_powers24(C (&x)[24], C w1)
{
    const C w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4), w16 = cmul(w8, w8);
    const C w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
    x[1] = cmul(x[1], w1); x[2] = cmul(x[2], w2); x[3] = cmul(x[3], w3); x[4] = cmul(x[4], w4);
    x[5] = cmul(x[5], w5); x[6] = cmul(x[6], w6); x[7] = cmul(x[7], w7); x[8] = cmul(x[8], w8);
    x[9] = cmul(x[9], cmul(w8, w1)); x[10] = cmul(x[10], cmul(w8, w2)); x[11] = cmul(x[11], cmul(w8, w3));
    x[12] = cmul(x[12], cmul(w8, w4)); x[13] = cmul(x[13], cmul(w8, w5)); x[14] = cmul(x[14], cmul(w8, w6));
    x[15] = cmul(x[15], cmul(w8, w7)); x[16] = cmul(x[16], w16);
    x[17] = cmul(x[17], cmul(w16, w1)); x[18] = cmul(x[18], cmul(w16, w2)); x[19] = cmul(x[19], cmul(w16, w3));
    x[20] = cmul(x[20], cmul(w16, w4)); x[21] = cmul(x[21], cmul(w16, w5)); x[22] = cmul(x[22], cmul(w16, w6));
    x[23] = cmul(x[23], cmul(w16, w7));
}

template <bool INV> struct Fft6144 {
    using C = double2;
    static constexpr int kPad = kFft6kPad;
    static __device__ __forceinline__ int sphys(int i) { return i + (i >> 4) * kPad; }
    struct Tw { C mid, last; };             // exp(-+2 pi i (t & 15) / 384), exp(-+2 pi i t / 6144)
    static __device__ __forceinline__ Tw load(int t)
    {
        Tw w;
        double s, c;
        sincospi(-2.0 * (double)(t & 15) / 384.0, &s, &c);
        w.mid = make_double2(c, INV ? -s : s);
        sincospi(-2.0 * (double)t / 6144.0, &s, &c);
        w.last = make_double2(c, INV ? -s : s);
        return w;
    }
    // registers (x[r] = element t + 384 r) -> registers (same layout); the caller guarantees that nobody still reads the image
    static __device__ __forceinline__ void run(C (&x)[16], void *lds_raw, const Tw &tw, int t)
    {
        double *lds = reinterpret_cast<double *>(lds_raw);
        const bool mid = t < 256;                                       // wave-uniform (lanes 0 .. 255 = wavefronts 0 .. 3)
        Dft<16, INV, C>::run(x);                                        // pass 1: output r of lane t is element 16 t + r
        C y[24];
        double *w1 = lds + (16 + kPad) * t;
        const double *r1 = lds + sphys(t);                              // element t + 256 r' sits at sphys(t) + (256 + 16 kPad) r'
#pragma unroll
        for (int r = 0; r < 16; r++) w1[r] = x[r].x;
        __syncthreads();
        if (mid) {
#pragma unroll
            for (int r = 0; r < 24; r++) y[r].x = r1[r * (256 + 16 * kPad)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) w1[r] = x[r].y;
        __syncthreads();
        const int k = t & 15, u = t >> 4;
        double *w2 = lds + (384 + 24 * kPad) * u + k;                   // output r' of butterfly t is element 384 u + k + 16 r'
        if (mid) {
#pragma unroll
            for (int r = 0; r < 24; r++) y[r].y = r1[r * (256 + 16 * kPad)];
            apply_twiddle_powers24(y, tw.mid);                          // pass 2: Ns = 16, radix 24
            Dft24<INV, C>::run(y);
        }
        __syncthreads();
        if (mid) {
#pragma unroll
            for (int r = 0; r < 24; r++) w2[r * (16 + kPad)] = y[r].x;
        }
        __syncthreads();
        const double *r2 = lds + sphys(t);                              // element t + 384 r sits at sphys(t) + (384 + 24 kPad) r
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].x = r2[r * (384 + 24 * kPad)];
        __syncthreads();
        if (mid) {
#pragma unroll
            for (int r = 0; r < 24; r++) w2[r * (16 + kPad)] = y[r].y;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[r].y = r2[r * (384 + 24 * kPad)];
        apply_twiddle_powers<16>(x, tw.last);                           // pass 3: Ns = 384, radix 16
        Dft<16, INV, C>::run(x);
    }
};

// What the overlap-save kernels call: registers (strided layout) -> registers, LDS image of lds_bytes.
template <int N, bool INV, typename C> struct TileFft {
    static constexpr bool kSplit = N == 4096 && sizeof(C) == 16;
    static constexpr int kLdsBytes = kSplit ? split4096_lds_bytes(8) : lds_elems<N>() * (int)sizeof(C);
    using Tw = typename FftRR<N, INV, C>::Tw;
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw) { return FftRR<N, INV, C>::load(tw); }
    static __device__ __forceinline__ void run(C (&x)[N / NT], void *lds, const Tw &t)
    {
        if constexpr (kSplit) {
            FftSplit4096<INV, C>::run(x, lds, t);
        } else {
            FftRR<N, INV, C>::first(x, reinterpret_cast<C *>(lds));
            FftRR<N, INV, C>::rest(reinterpret_cast<C *>(lds), x, t);
        }
    }
    template <int D> static __device__ __forceinline__ void run_poly(C (&x)[N / NT], void *lds, const Tw &t)
    {
        static_assert(kSplit, "polyphase plan: the fp64 4096-point transform");
        FftSplit4096<INV, C>::template run_poly<D>(x, lds, t);
    }
};

template <bool INV, typename C> struct FftRR<2048, INV, C> {
    struct Tw { C a[2], b[1], c; };
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw)
    {
        Tw t;
        t.a[0] = pass_twiddle<8, INV>(tw, threadIdx.x);
        t.a[1] = pass_twiddle<8, INV>(tw, threadIdx.x + NT);
        t.b[0] = pass_twiddle<32, INV>(tw + 8, threadIdx.x);
        t.c = pass_twiddle<256, INV>(tw + 40, threadIdx.x);
        return t;
    }
    static __device__ __forceinline__ void first(C (&x)[8], C *lds) { pass_regs_to_lds<2048, 8, INV>(x, lds); }
    static __device__ __forceinline__ void rest(C *lds, C (&x)[8], const Tw &t)
    {
        __syncthreads();
        pass_lds_to_lds<2048, 4, 8, INV>(lds, t.a);
        __syncthreads();
        pass_lds_to_lds<2048, 8, 32, INV>(lds, t.b);
        __syncthreads();
        pass_lds_to_regs<2048, 8, INV>(lds, x, t.c);
    }
};

template <bool INV, typename C> struct FftRR<1024, INV, C> {
    struct Tw { C a[1], b[1], c[1], d; };
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw)
    {
        Tw t;
        t.a[0] = pass_twiddle<4, INV>(tw, threadIdx.x);
        t.b[0] = pass_twiddle<16, INV>(tw + 4, threadIdx.x);
        t.c[0] = pass_twiddle<64, INV>(tw + 20, threadIdx.x);
        t.d = pass_twiddle<256, INV>(tw + 84, threadIdx.x);
        return t;
    }
    static __device__ __forceinline__ void first(C (&x)[4], C *lds) { pass_regs_to_lds<1024, 4, INV>(x, lds); }
    static __device__ __forceinline__ void rest(C *lds, C (&x)[4], const Tw &t)
    {
        __syncthreads();
        pass_lds_to_lds<1024, 4, 4, INV>(lds, t.a);
        __syncthreads();
        pass_lds_to_lds<1024, 4, 16, INV>(lds, t.b);
        __syncthreads();
        pass_lds_to_lds<1024, 4, 64, INV>(lds, t.c);
        __syncthreads();
        pass_lds_to_regs<1024, 4, INV>(lds, x, t.d);
    }
};

template <bool INV, typename C> struct FftRR<512, INV, C> {
    struct Tw { C a[1], b[1], c; };
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw)
    {
        Tw t;
        t.a[0] = pass_twiddle<2, INV>(tw, threadIdx.x);
        t.b[0] = pass_twiddle<32, INV>(tw + 2, threadIdx.x);
        t.c = pass_twiddle<256, INV>(tw + 34, threadIdx.x);
        return t;
    }
    static __device__ __forceinline__ void first(C (&x)[2], C *lds) { pass_regs_to_lds<512, 2, INV>(x, lds); }
    static __device__ __forceinline__ void rest(C *lds, C (&x)[2], const Tw &t)
    {
        __syncthreads();
        pass_lds_to_lds<512, 16, 2, INV>(lds, t.a);
        __syncthreads();
        pass_lds_to_lds<512, 8, 32, INV>(lds, t.b);
        __syncthreads();
        pass_lds_to_regs<512, 2, INV>(lds, x, t.c);
    }
};

template <bool INV, typename C> struct FftRR<8192, INV, C> {
    struct Tw { C a[4], b; };
    static __device__ __forceinline__ Tw load(const C *__restrict__ tw)
    {
        Tw t;
#pragma unroll
        for (int p = 0; p < 4; p++) t.a[p] = pass_twiddle<32, INV>(tw, threadIdx.x + p * NT);
        t.b = pass_twiddle<256, INV>(tw + 32, threadIdx.x);
        return t;
    }
    static __device__ __forceinline__ void first(C (&x)[32], C *lds) { pass_regs_to_lds<8192, 32, INV>(x, lds); }
    static __device__ __forceinline__ void rest(C *lds, C (&x)[32], const Tw &t)
    {
        __syncthreads();
        pass_lds_to_lds<8192, 8, 32, INV>(lds, t.a);
        __syncthreads();
        pass_lds_to_regs<8192, 32, INV>(lds, x, t.b);
    }
};

}  // namespace qh
