// qh_qagc.hip -- Quisk's audio AGC (include/quiskhip.h group 8): process_agc, quisk.c:2162-2287, for `nch` streams.
//
// A 15 ms FIFO (AGC_DELAY, quisk.c:47) delays the audio while a five-branch state machine moves the gain: ramp
// down linearly over the FIFO length when a sample would exceed max_out, otherwise relax exponentially towards
// min(agcReleaseGain, max_out * CLIP32 / largest sample of the last FIFO cycle).  The recurrence is non-linear and
// sequential: one wavefront per stream, lanes hold 64 consecutive samples and lane i keeps the gain that applied to
// sample i; the FIFO lives in global memory in the reference's ring order, so a call leaves exactly the reference's
// state.  Three kernels, bit-identical: q_agc_kernel steps the whole machine sample by sample (every lane the same
// scalar state; FIFOs shorter than 128 samples, and the diagnostic form); q_agc_chain_kernel takes the two regimes as
// chains of the one or two instructions that carry the gain from sample to sample and handles the turns between them
// with ballots (FIFOs of 128 .. 191 samples); q_agc_pair_kernel, the one that runs at 12.8 ksps and above, is the same
// with a second wavefront that moves the samples.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "qh_internal.hpp"
#include "qh_wave.hpp"

using namespace qh;

namespace {

constexpr double kClip32 = 2147483647.0;      // CLIP32, quisk.h:13

struct QAgcParam { double limit /* max_out * CLIP32 */, time_release; int buf_size, is_cpx; };
struct QAgcState { int index_read, index_start, is_clipping, pad; double themax, gain, delta, target_gain; };

struct QAgcLane { double g, T, d, mx; int clip, is; };

__device__ __forceinline__ double qagc_mag(double2 z, int is_cpx) { return is_cpx ? hypot(z.x, z.y) : fabs(z.x); }

// One sample of the state machine, quisk.c:2212-2272; b = its magnitude, ir = the FIFO index it is written to.
// No FMA contraction, the C source's operation order: the branches taken are the reference's.
__device__ __forceinline__ void qagc_step(QAgcLane &s, double b, int ir, double limit, double tr, double rg, int B)
{
#pragma clang fp contract(off)
    if (s.clip == 0) {
        if (b * s.g > limit) {
            s.T = limit / b;
            s.d = (s.g - s.T) / B;
            s.clip = 1;
            s.mx = b;
            s.g -= s.d;
        } else if (ir == s.is) {
            const double clip_gain = limit / s.mx;
            s.T = rg > clip_gain ? clip_gain : rg;
            s.mx = b;
            s.g = s.g * (1.0 - tr) + s.T * tr;
        } else {
            if (s.mx < b) s.mx = b;
            s.g = s.g * (1.0 - tr) + s.T * tr;
        }
    } else {
        if (b > s.mx) {
            s.mx = b;
            s.T = limit / b;
            const double dtmp = (s.g - s.T) / B;
            if (dtmp > s.d) s.d = dtmp;
        }
        s.g -= s.d;
        if (s.g <= s.T) {
            s.clip = 0;
            s.g = s.T;
            s.mx = b;
            s.is = ir;
        }
    }
}

__device__ __forceinline__ QAgcLane qagc_lane_of(const QAgcState &st)
{
    return QAgcLane{ st.gain, st.target_gain, st.delta, st.themax, st.is_clipping, st.index_start };
}

// src == dst or two buffers
__global__ __launch_bounds__(64) void q_agc_kernel(const double2 *src, long long sstride, double2 *dst, long long dstride, int n,
                                                   QAgcState *state, double2 *ring, const double *release_gain, QAgcParam q)
{
    // The overload ramp is built to END on a comparison that is exact in real arithmetic (gain - B * delta ==
    // target, quisk.c:2219,2257): which step leaves the ramp is decided by the last bit.  qagc_step has no FMA
    // contraction and the C source's operation order, so the state machine takes the reference's branches.
    const int ch = blockIdx.x, lane = threadIdx.x;
    const double2 *p = src + (long long)ch * sstride;
    double2 *o_ = dst + (long long)ch * dstride;
    double2 *rb = ring + (long long)ch * q.buf_size;
    const QAgcState st0 = state[ch];
    QAgcLane st = qagc_lane_of(st0);
    int index_read = st0.index_read;
    const double rg = release_gain[ch];
    const int B = q.buf_size;
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        double2 z = make_double2(0, 0), d = make_double2(0, 0);
        int ri = index_read + lane;
        if (ri >= B) ri -= B;
        if (lane < cnt) {
            z = p[base + lane];
            d = rb[ri];                              // FIFO output: the sample written B steps ago
            rb[ri] = z;                              // "write new sample at read index"
        }
        const double bm = qagc_mag(z, q.is_cpx);
        double mygain = 0.0;
        for (int i = 0; i < cnt; i++) {              // uniform: every lane steps the same state
            if (lane == i) mygain = st.g;
            int ir = index_read + i;
            if (ir >= B) ir -= B;
            qagc_step(st, lane_bcast(bm, i), ir, q.limit, q.time_release, rg, B);
        }
        index_read += cnt;
        if (index_read >= B) index_read -= B;
        if (lane < cnt) {
            double2 o = make_double2(d.x * mygain, d.y * mygain);
            const double om = q.is_cpx ? hypot(o.x, o.y) : fabs(o.x);
            if (om > kClip32) { o.x /= om; o.y /= om; }     // quisk.c:2204-2205
            o_[base + lane] = o;
        }
    }
    if (lane == 0) {
        QAgcState s2;
        s2.index_read = index_read; s2.index_start = st.is; s2.is_clipping = st.clip; s2.pad = 0;
        s2.themax = st.mx; s2.gain = st.g; s2.delta = st.d; s2.target_gain = st.T;
        state[ch] = s2;
    }
}

// ---- the same state machine, the two regimes as chains -----------------------------------------------------------------------------
// Outside an overload the gain follows g <- fl(fl(g (1 - r)) + fl(T r)) with T fixed until the FIFO cycle's first sample; inside
// one, g <- fl(g - delta).  Nothing else sits on the path from one sample's gain to the next, so a chunk of 64 samples is taken as
// a chain of those two (one) instructions with the EXEC mask shrinking by one lane per step: lane j ends up holding the gain AHEAD
// of sample j, every rounding the reference's.  The events -- a sample that would exceed the limit, a new largest sample inside a
// ramp, the ramp's end, the cycle's first sample -- are then found with one ballot over the lanes, everything ahead of the first
// one is accepted, and that sample is one qagc_step.  8 cycles per sample instead of the ~150 of stepping the whole machine.
//
// gl: per lane; lanes p + 1 .. get 1, 2, .. steps (the chain stops after `steps`).  p wave-uniform.
__device__ __forceinline__ double qagc_chain_relax(double g, double a, double c, int p, int steps)
{
    unsigned long long m = p >= 63 ? 0ull : ~0ull << (p + 1), sv;
    double gl = g;
    if (steps >= 56) {
        // (nearly) a whole chunk, no loop: lane p + 1 + r needs r + 1 = 4 q + k + 1 steps.  q blocks of four steps with EXEC moving up
        // FOUR lanes per block (one EXEC write per four steps instead of one per step: 13.5 against 18 clocks per step), then four
        // single steps on the lanes whose r mod 4 is at least 0, 1, 2, 3.  Every lane still runs its own steps one after the other.
        const int sh = p + 1;
        const unsigned long long k1 = 0xEEEEEEEEEEEEEEEEull << sh, k2 = 0xCCCCCCCCCCCCCCCCull << sh, k3 = 0x8888888888888888ull << sh;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_lshl_b64 exec, %[m], 4\n\t"
                     ".rept 15\n\t"
                     ".rept 4\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     ".endr\n\t"
                     "s_lshl_b64 exec, exec, 4\n\t"
                     ".endr\n\t"
                     "s_mov_b64 exec, %[m]\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     "s_mov_b64 exec, %[k1]\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     "s_mov_b64 exec, %[k2]\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     "s_mov_b64 exec, %[k3]\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [g] "+v"(gl), [sv] "=&s"(sv)
                     : [a] "v"(a), [c] "v"(c), [m] "s"(m), [k1] "s"(k1), [k2] "s"(k2), [k3] "s"(k3)
                     : "scc");
        return gl;
    }
    while (steps > 0 && m) {
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m]\n\t"
                     ".rept 8\n\t"
                     "v_mul_f64 %[g], %[g], %[a]\n\t"
                     "v_add_f64 %[g], %[g], %[c]\n\t"
                     "s_lshl_b64 exec, exec, 1\n\t"
                     ".endr\n\t"
                     "s_mov_b64 %[m], exec\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [g] "+v"(gl), [m] "+s"(m), [sv] "=&s"(sv)
                     : [a] "v"(a), [c] "v"(c)
                     : "scc");
        steps -= 8;
    }
    return gl;
}
__device__ __forceinline__ double qagc_chain_ramp(double g, double nd, int p, int steps)
{
    unsigned long long m = p >= 63 ? 0ull : ~0ull << (p + 1), sv;
    double gl = g;
    if (steps >= 56) {          // as in qagc_chain_relax: blocks of four steps, then the four remainders
        const int sh = p + 1;
        const unsigned long long k1 = 0xEEEEEEEEEEEEEEEEull << sh, k2 = 0xCCCCCCCCCCCCCCCCull << sh, k3 = 0x8888888888888888ull << sh;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_lshl_b64 exec, %[m], 4\n\t"
                     ".rept 15\n\t"
                     ".rept 4\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     ".endr\n\t"
                     "s_lshl_b64 exec, exec, 4\n\t"
                     ".endr\n\t"
                     "s_mov_b64 exec, %[m]\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     "s_mov_b64 exec, %[k1]\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     "s_mov_b64 exec, %[k2]\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     "s_mov_b64 exec, %[k3]\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [g] "+v"(gl), [sv] "=&s"(sv)
                     : [nd] "v"(nd), [m] "s"(m), [k1] "s"(k1), [k2] "s"(k2), [k3] "s"(k3)
                     : "scc");
        return gl;
    }
    while (steps > 0 && m) {
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m]\n\t"
                     ".rept 8\n\t"
                     "v_add_f64 %[g], %[g], %[nd]\n\t"
                     "s_lshl_b64 exec, exec, 1\n\t"
                     ".endr\n\t"
                     "s_mov_b64 %[m], exec\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [g] "+v"(gl), [m] "+s"(m), [sv] "=&s"(sv)
                     : [nd] "v"(nd)
                     : "scc");
        steps -= 8;
    }
    return gl;
}

// The largest of b over the lanes of `acc`, given that at least one of them is above `floor_` (records are few: a ballot per record
// instead of a six-round butterfly through the LDS crossbar).
__device__ __forceinline__ double qagc_max_above(double b, bool acc, double floor_)
{
    double m = floor_;
    unsigned long long above = __ballot(acc && b > m);
    while (above) {
        m = lane_bcast(b, __ffsll((long long)above) - 1);
        above = __ballot(acc && b > m);
    }
    return m;
}

struct QAgcChainPrm { double limit, tr, a, rg; int B; };

// One chunk: cnt <= 64 samples with magnitudes bm in the lanes, the first written to FIFO index index_read.  Returns, per lane, the
// gain its sample's FIFO output is multiplied by (the gain ahead of that sample's step).
__device__ __forceinline__ double qagc_chunk_exact(QAgcLane &st, double bm, int cnt, int index_read, int lane, const QAgcChainPrm &w)
{
#pragma clang fp contract(off)
    const int B = w.B;
    double mygain = 0.0;
    int p = 0;
    if (st.clip == 0 && cnt == 64) {
        // the common chunk: 64 samples of relaxing, the cycle's first sample not among them, no sample over the limit -- one chain,
        // two ballots, one branch
        int c = st.is - index_read;
        if (c < 0) c += B;
        if (__builtin_amdgcn_readfirstlane(c) >= 64) {
            const double cT = st.T * w.tr;
            const double gl = qagc_chain_relax(st.g, w.a, cT, 0, 63);
            if (!__ballot(bm * gl > w.limit)) {
                st.mx = qagc_max_above(bm, true, st.mx);
                st.g = lane_bcast(gl, 63) * w.a + cT;
                return gl;
            }
        }
    }
    if (st.clip == 1 && cnt == 64 && !__ballot(bm > st.mx)) {
        // a chunk inside a ramp with no new largest sample: one chain, one ballot for the ramp's end
        const double gl = qagc_chain_ramp(st.g, -st.d, 0, 63);
        const double after = gl - st.d;
        if (!__ballot(after <= st.T)) {
            st.g = lane_bcast(after, 63);
            return gl;
        }
    }
    while (p < cnt) {
        int irp = index_read + p;
        if (irp >= B) irp -= B;
        if (st.clip == 0) {
            int c = st.is - irp;                 // samples ahead of the cycle's first one
            if (c < 0) c += B;
            const int e = __builtin_amdgcn_readfirstlane(p + c < cnt ? p + c : cnt);
            if (e > p) {
                const double cT = st.T * w.tr;
                const double gl = qagc_chain_relax(st.g, w.a, cT, p, e - p - 1);
                const bool in = lane >= p && lane < e;
                const unsigned long long trig = __ballot(in && bm * gl > w.limit);
                const int k = trig ? __ffsll((long long)trig) - 1 : e;
                const bool acc = lane >= p && lane < k;
                if (acc || lane == k) mygain = gl;
                st.mx = qagc_max_above(bm, acc, st.mx);
                if (k < e) {
                    st.g = lane_bcast(gl, k);
                    int irk = index_read + k;
                    if (irk >= B) irk -= B;
                    qagc_step(st, lane_bcast(bm, k), irk, w.limit, w.tr, w.rg, B);       // the overload begins
                    p = k + 1;
                } else {
                    const double gq = lane_bcast(gl, e - 1);
                    st.g = gq * w.a + cT;
                    p = e;
                }
            } else {
                if (lane == p) mygain = st.g;
                qagc_step(st, lane_bcast(bm, p), irp, w.limit, w.tr, w.rg, B);           // the cycle's first sample: a new target
                p++;
            }
        } else {
            const unsigned long long nm = __ballot(lane >= p && lane < cnt && bm > st.mx);
            const int k1 = nm ? __ffsll((long long)nm) - 1 : cnt;
            if (k1 > p) {
                const double gl = qagc_chain_ramp(st.g, -st.d, p, k1 - p - 1);
                const double after = gl - st.d;
                const bool in = lane >= p && lane < k1;
                const unsigned long long ex = __ballot(in && after <= st.T);
                if (ex) {
                    const int jx = __ffsll((long long)ex) - 1;
                    if (lane >= p && lane <= jx) mygain = gl;
                    int irx = index_read + jx;
                    if (irx >= B) irx -= B;
                    st.clip = 0; st.g = st.T; st.mx = lane_bcast(bm, jx); st.is = irx;      // quisk.c:2257-2265
                    p = jx + 1;
                } else {
                    if (in) mygain = gl;
                    st.g = lane_bcast(after, k1 - 1);
                    p = k1;
                }
            } else {
                if (lane == p) mygain = st.g;
                qagc_step(st, lane_bcast(bm, p), irp, w.limit, w.tr, w.rg, B);           // a new largest sample inside the ramp
                p++;
            }
        }
        p = __builtin_amdgcn_readfirstlane(p);
        st.clip = __builtin_amdgcn_readfirstlane(st.clip);
        st.is = __builtin_amdgcn_readfirstlane(st.is);
    }
    return mygain;
}

// D chunks of input and FIFO output are in flight while one is stepped: chunk c + D's FIFO entries were written B - 64 D samples
// ahead of chunk c, so D <= B / 64 - 1 (the host runs D = 1 for FIFOs of 128 .. 191 samples; longer ones take q_agc_pair_kernel).
template <int D>
__global__ __launch_bounds__(64) void q_agc_chain_kernel(const double2 *src, long long sstride, double2 *dst, long long dstride, int n,
                                                         QAgcState *state, double2 *ring, const double *release_gain, QAgcParam q)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    const double2 *x = src + (long long)ch * sstride;
    double2 *y = dst + (long long)ch * dstride;
    double2 *rb = ring + (long long)ch * q.buf_size;
    const QAgcState st0 = state[ch];
    QAgcLane st = qagc_lane_of(st0);
    int index_read = st0.index_read;
    const QAgcChainPrm w{ q.limit, q.time_release, 1.0 - q.time_release, release_gain[ch], q.buf_size };
    const int B = q.buf_size;
    // (loads with clamped indices instead of predicates: a predicated load keeps its old register alive through a copy, and the copy
    // waits for every load in flight)
    double2 zq[D], dq[D];
    int rq = index_read + lane;                  // FIFO index of the chunk being loaded, this lane
    if (rq >= B) rq -= B;
#pragma unroll
    for (int u = 0; u < D; u++) {
        const int i = u * 64 + lane;
        zq[u] = x[i < n ? i : n - 1];
        dq[u] = rb[rq];
        rq += 64;
        if (rq >= B) rq -= B;
    }
    int ri = index_read + lane;
    if (ri >= B) ri -= B;
    for (int base0 = 0; base0 < n; base0 += 64 * D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int base = base0 + 64 * u;
            const int left = n - base, cnt = left < 64 ? left : 64;
            const double2 z = zq[u], d = dq[u];
            if (lane < cnt) rb[ri] = z;              // "write new sample at read index"
            ri += 64;
            if (ri >= B) ri -= B;
            if (cnt > 0) {
                const double bm = qagc_mag(z, q.is_cpx);
                const double mygain = qagc_chunk_exact(st, bm, cnt, index_read, lane, w);
                index_read += cnt;
                if (index_read >= B) index_read -= B;
                if (lane < cnt) {
                    double2 o = make_double2(d.x * mygain, d.y * mygain);
                    const double om = q.is_cpx ? hypot(o.x, o.y) : fabs(o.x);
                    if (om > kClip32) { o.x /= om; o.y /= om; }     // quisk.c:2204-2205
                    y[base + lane] = o;
                }
            }
            {   // chunk base + 64 D into the registers this chunk has just finished with (no copies at the loop's end)
                const int i = base + 64 * D + lane;
                zq[u] = x[i < n ? i : n - 1];
                dq[u] = rb[rq];
                rq += 64;
                if (rq >= B) rq -= B;
            }
        }
    }
    if (lane == 0) {
        QAgcState s2;
        s2.index_read = index_read; s2.index_start = st.is; s2.is_clipping = st.clip; s2.pad = 0;
        s2.themax = st.mx; s2.gain = st.g; s2.delta = st.d; s2.target_gain = st.T;
        state[ch] = s2;
    }
}

// Two wavefronts per stream: wave 0 steps (magnitudes in, gains out, through LDS); wave 1 moves -- while chunk c is stepped it
// writes chunk c + 1 to the FIFO and takes its magnitudes, and multiplies chunk c - 1's FIFO output by its gains and stores it.
// S register slots take turns: the slot of the chunk just stored receives the loads of chunk c + S - 1 (S = 8: six chunks, some
// 8 000 clocks, between a load and its use -- with S = 4 the mover waited for HBM and the stepper for the mover: 2.77 against
// 2.34 ms).  The FIFO entries of the call's first S - 1 chunks must be old ones, so B >= 64 (S - 1); S = 4 below 448 samples.
template <int S>
__global__ __launch_bounds__(128) void q_agc_pair_kernel(const double2 *src, long long sstride, double2 *dst, long long dstride, int n,
                                                         QAgcState *state, double2 *ring, const double *release_gain, QAgcParam q)
{
    __shared__ double s_bm[2][64], s_gain[2][64];
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = q.buf_size, nchunks = (n + 63) >> 6;
    const QAgcState st0 = state[ch];
    if (wave == 0) {
        // the stepper is one chain of dependent instructions: when other kernels' wavefronts share its SIMD (the receiver bank runs
        // the next piece's filters beside it) it should issue the moment its operand is ready -- highest wave priority
        __builtin_amdgcn_s_setprio(3);
        QAgcLane st = qagc_lane_of(st0);
        int index_read = st0.index_read;
        const QAgcChainPrm w{ q.limit, q.time_release, 1.0 - q.time_release, release_gain[ch], B };
        __syncthreads();                             // chunk 0's magnitudes are there
        for (int c = 0; c < nchunks; c++) {
            const int left = n - c * 64, cnt = left < 64 ? left : 64;
            const double bm = s_bm[c & 1][lane];
            s_gain[c & 1][lane] = qagc_chunk_exact(st, bm, cnt, index_read, lane, w);
            index_read += cnt;
            if (index_read >= B) index_read -= B;
            __syncthreads();
        }
        if (lane == 0) {
            QAgcState s2;
            s2.index_read = index_read; s2.index_start = st.is; s2.is_clipping = st.clip; s2.pad = 0;
            s2.themax = st.mx; s2.gain = st.g; s2.delta = st.d; s2.target_gain = st.T;
            state[ch] = s2;
        }
        return;
    }
    const double2 *x = src + (long long)ch * sstride;
    double2 *y = dst + (long long)ch * dstride;
    double2 *rb = ring + (long long)ch * B;
    double2 zq[S], dq[S];
    int rq = st0.index_read + lane;                  // FIFO index of the chunk being loaded, this lane
    if (rq >= B) rq -= B;
    int ri = rq;                                     // ... of the chunk being written
#pragma unroll
    for (int u = 0; u < S - 1; u++) {
        const int i = u * 64 + lane;
        zq[u] = x[i < n ? i : n - 1];
        dq[u] = rb[rq];
        rq += 64;
        if (rq >= B) rq -= B;
    }
    zq[S - 1] = dq[S - 1] = make_double2(0, 0);
    auto prepare = [&](int c, const double2 &z) {    // chunk c: into the FIFO, its magnitudes to the stepper
        if (c * 64 + lane < n) rb[ri] = z;           // "write new sample at read index"
        ri += 64;
        if (ri >= B) ri -= B;
        s_bm[c & 1][lane] = qagc_mag(z, q.is_cpx);
    };
    prepare(0, zq[0]);
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += S) {
#pragma unroll
        for (int u = 0; u < S; u++) {
            const int c = c0 + u;
            if (c < nchunks) {
                if (c + 1 < nchunks) prepare(c + 1, zq[(u + 1) & (S - 1)]);
                if (c >= 1) {                        // chunk c - 1: its FIFO output times the gains the stepper left
                    const int base = (c - 1) * 64;
                    const double g = s_gain[(c - 1) & 1][lane];
                    const double2 d = dq[(u + S - 1) & (S - 1)];
                    double2 o = make_double2(d.x * g, d.y * g);
                    const double om = q.is_cpx ? hypot(o.x, o.y) : fabs(o.x);
                    if (om > kClip32) { o.x /= om; o.y /= om; }     // quisk.c:2204-2205
                    y[base + lane] = o;              // (base + lane < n: chunk c - 1 is a full one)
                }
                {   // chunk c + S - 1 into the slot chunk c - 1 has just left
                    const int i = (c + S - 1) * 64 + lane;
                    zq[(u + S - 1) & (S - 1)] = x[i < n ? i : n - 1];
                    dq[(u + S - 1) & (S - 1)] = rb[rq];
                    rq += 64;
                    if (rq >= B) rq -= B;
                }
                __syncthreads();
            }
        }
    }
    {   // the last chunk
        const int c = nchunks - 1, base = c * 64;
        if (base + lane < n) {
            const double g = s_gain[c & 1][lane];
            double2 d = dq[0];                       // (slot c mod S, picked by value: an indexed register array would live in scratch)
#pragma unroll
            for (int q = 1; q < S; q++) if ((c & (S - 1)) == q) d = dq[q];
            double2 o = make_double2(d.x * g, d.y * g);
            const double om = q.is_cpx ? hypot(o.x, o.y) : fabs(o.x);
            if (om > kClip32) { o.x /= om; o.y /= om; }
            y[base + lane] = o;
        }
    }
}

}  // namespace

struct qh_qagc {
    int device = 0, nch = 0, sample_rate = 0;
    bool inited = false;            // the reference's first call only initialises (quisk.c:2173-2190)
    QAgcParam prm{};
    QAgcState *state = nullptr;
    double2 *ring = nullptr;
    double *gain = nullptr;
    std::vector<double> h_gain;
    bool gain_dirty = true;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int form = 0;                   // diagnostics, see qh_qagc_debug_form
    ~qh_qagc()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        (void)hipFree(state); (void)hipFree(ring); (void)hipFree(gain);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

static int qagc_init_state(qh_qagc *h)
{
    std::vector<QAgcState> st((size_t)h->nch);
    for (auto &s : st) { s.index_read = 0; s.index_start = 0; s.is_clipping = 0; s.pad = 0; s.themax = 1.0; s.gain = 100; s.delta = 0; s.target_gain = 100; }
    QH_HIP(hipMemcpyAsync(h->state, st.data(), st.size() * sizeof(QAgcState), hipMemcpyHostToDevice, h->stream));
    QH_HIP(hipMemsetAsync(h->ring, 0, (size_t)h->nch * (size_t)h->prm.buf_size * sizeof(double2), h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    return QH_OK;
}

extern "C" {

qh_qagc *qh_qagc_create(int device, int nch, int sample_rate, double max_out, double release_time, int is_cpx, void *stream)
{
    if (nch <= 0 || sample_rate < 1000 || !(max_out > 0.0) || !(release_time > 0.0)) {
        set_error(QH_ERR_INVALID, "qh_qagc_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_qagc *h = new qh_qagc();
    h->device = device; h->nch = nch; h->sample_rate = sample_rate;
    h->prm.limit = max_out * kClip32;
    h->prm.time_release = 1.0 - std::exp(-1.0 / sample_rate / release_time);      // quisk.c:2185
    h->prm.buf_size = sample_rate * 15 / 1000;                                     // AGC_DELAY, quisk.c:47,2176
    h->prm.is_cpx = is_cpx ? 1 : 0;
    h->h_gain.assign((size_t)nch, 80.0);                                           // agcReleaseGain, quisk.c:191
    auto fail = [&](const char *what) -> qh_qagc * { set_error(QH_ERR_HIP, "qh_qagc_create: %s failed", what); delete h; return nullptr; };
    if (h->prm.buf_size < 64) { set_error(QH_ERR_INVALID, "qh_qagc_create: sample rate too low"); delete h; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
        h->own_stream = true;
    }
    h->stream = s;
    if (hipMalloc((void **)&h->state, (size_t)nch * sizeof(QAgcState)) != hipSuccess ||
        hipMalloc((void **)&h->ring, (size_t)nch * (size_t)h->prm.buf_size * sizeof(double2)) != hipSuccess ||
        hipMalloc((void **)&h->gain, (size_t)nch * sizeof(double)) != hipSuccess) return fail("hipMalloc");
    if (qagc_init_state(h)) { delete h; return nullptr; }
    return h;
}

void qh_qagc_destroy(qh_qagc *h) { delete h; }

// set_agc (quisk.c:4543): the AGC's maximum gain
int qh_qagc_set_gain(qh_qagc *h, int ch, double release_gain)
{
    if (!h || ch < -1 || ch >= h->nch) return set_error(QH_ERR_INVALID, "qh_qagc_set_gain: bad arguments");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? h->nch : ch + 1); c++) h->h_gain[(size_t)c] = release_gain;
    h->gain_dirty = true;
    return QH_OK;
}

// process_agc takes is_cpx per call (quisk.c:2162: |z| for the DGT-IQ stream, |Re z| otherwise); the state carries over
int qh_qagc_set_cpx(qh_qagc *h, int is_cpx)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_qagc_set_cpx: null handle");
    h->prm.is_cpx = is_cpx ? 1 : 0;
    return QH_OK;
}

int qh_qagc_reset(qh_qagc *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_qagc_reset: null handle");
    QH_HIP(hipSetDevice(h->device));
    h->inited = false;
    return qagc_init_state(h);
}

// src -> dst (two buffers, or the same one): process_agc(dat, cSamples, count, is_cpx) for every stream
int qh_qagc_process2(qh_qagc *h, const void *d_src, long long src_stride, void *d_dst, long long dst_stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!d_src || !d_dst || src_stride < n || dst_stride < n)))
        return set_error(QH_ERR_INVALID, "qh_qagc_process: bad arguments");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    if (!h->inited) {               // first call: state set up, samples untouched
        h->inited = true;
        if (d_src != d_dst)
            QH_HIP(hipMemcpy2DAsync(d_dst, (size_t)dst_stride * 16, d_src, (size_t)src_stride * 16, (size_t)n * 16, (size_t)h->nch,
                                    hipMemcpyDeviceToDevice, h->stream));
        return QH_OK;
    }
    if (h->gain_dirty) {
        QH_HIP(hipMemcpyAsync(h->gain, h->h_gain.data(), (size_t)h->nch * sizeof(double), hipMemcpyHostToDevice, h->stream));
        QH_HIP(hipStreamSynchronize(h->stream));
        h->gain_dirty = false;
    }
    // chunks in flight: D <= B / 64 - 1 (q_agc_chain_kernel); a FIFO of 64 .. 127 samples takes the plain kernel
    const int B = h->prm.buf_size;
    auto *kern = h->form == 1 || B < 128 ? q_agc_kernel : B >= 448 ? q_agc_pair_kernel<8> : B >= 192 ? q_agc_pair_kernel<4> : q_agc_chain_kernel<1>;
    hipLaunchKernelGGL(kern, dim3((unsigned)h->nch), dim3(B >= 192 && h->form != 1 ? 128 : 64), 0, h->stream, (const double2 *)d_src, src_stride, (double2 *)d_dst, dst_stride, n,
                       h->state, h->ring, h->gain, h->prm);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_qagc_process(qh_qagc *h, void *d_buf, long long stride, int n) { return qh_qagc_process2(h, d_buf, stride, d_buf, stride, n); }

// diagnostics: 0 the two regimes as chains (default), 1 the whole machine sample by sample (bit-identical, ~9 times slower)
int qh_qagc_debug_form(qh_qagc *h, int form)
{
    if (!h || form < 0 || form > 1) return set_error(QH_ERR_INVALID, "qh_qagc_debug_form: bad arguments");
    h->form = form;
    return QH_OK;
}

int qh_qagc_process_host(qh_qagc *h, void *h_buf, long long stride, int n)
{
    if (!h || n < 0 || (n > 0 && (!h_buf || stride < n))) return set_error(QH_ERR_INVALID, "qh_qagc_process_host: bad arguments");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    double2 *d = nullptr;
    QH_HIP(hipMalloc((void **)&d, (size_t)h->nch * (size_t)n * sizeof(double2)));
    int rc = QH_OK;
    if (hipMemcpy2DAsync(d, (size_t)n * 16, h_buf, (size_t)stride * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyHostToDevice, h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "upload failed");
    if (rc == QH_OK) rc = qh_qagc_process(h, d, n, n);
    if (rc == QH_OK && hipMemcpy2DAsync(h_buf, (size_t)stride * 16, d, (size_t)n * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyDeviceToHost,
                                         h->stream) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "download failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "synchronize failed");
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
