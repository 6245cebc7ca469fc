// qh_agc_tiled.hpp -- xwcpagc modes 1-4 (wdsp/wcpAGC.c:177-338) over LONG calls, in time tiles.
//
// Per sample the reference (a) moves the delay line and its magnitudes, (b) updates two one-pole averages of the delayed
// magnitude, (c) keeps ring_max, the maximum of the magnitudes in the attack window, (d) steps the five-state level detector
// `volts`, (e) applies the gain curve to the delayed sample.  Only (d) is a recurrence that is not linear:
//   agc_prep_kernel        (a) + (c): RM_j = max |x| over samples (j - A, j] -- block prefix / suffix maxima in LDS; what the
//                          reference's rescans and single compares amount to while ring_max is not stale (it is stale only after
//                          SetRXAAGCAttack has moved in_index in mid-stream: the engine keeps such a channel on wcpagc_kernel);
//                          also keeps the A samples ahead of every 1024-sample tile for agc_apply_kernel
//   agc_avg_tiled_kernel   (b): two linear one-pole scans of |x_{j-A}| over time segments, the carries chained from the prep kernel's tile sums
//   agc_bounds_kernel      (d), coarse: the detector's state at every tile boundary.  One wavefront walks a stretch of the call jumping over
//                          the runs the detector spends in one regime in closed form.  The call is cut into K segments walked at once, each
//                          from the state the call began in; agc_bounds_round_kernel then walks every segment again from the end of the one
//                          before it until the new walk meets the old one (two runs of the detector meet at the rate of the attack steps
//                          they share -- one step in twenty on a steady noisy signal, ~10 000 samples -- and not at all while both decay
//                          after a peak, which is why a fixed warm-up ahead of every tile, as the PLL tiles use, does not work here);
//                          agc_bounds_fix_kernel catches, in order, what the rounds left
//   agc_lanes_kernel       (d), exact: ONE LANE PER TILE steps its tile sample by sample from its boundary state
//   agc_verify_kernel      checks every boundary state against the end of the tile before it and re-runs, in order, the tiles whose
//                          start was off (exact compare of the discrete state, 1e-9 on volts)
//   agc_apply_kernel       (e), in place: a workgroup stages its tile and the A samples ahead of it in LDS first
//   agc_finish_kernel      the state the next call (tiled or not) starts from: ring slots, indices, averages, detector state
// The step itself is agc_lane_step: the reference's switch as selects, `volts += (ring_max - volts) * mult` with mult = 0 where the
// reference leaves volts alone, every update an FMA as in wcpagc_kernel.
#pragma once
#include "qh_demod.hpp"

namespace qh {

constexpr int kAgcTile = 1024;          // samples per workgroup in the lane-parallel kernels
constexpr int kAgcEndsW = 8;            // a detector state: volts, save_volts, hang counter, decay type, state
constexpr int kAgcBatch = 16;           // steps per LDS round of the lanes kernel
constexpr int kAgcPitch = kAgcBatch + 1;

struct AgcLane { double volts, save_volts; int hc, decay_type, st; };

__device__ __forceinline__ void agc_lane_step(AgcLane &s, double rm, double fba, double hba, const AgcParam &q)
{
    int hc = s.hc;
    hc = hc > 0 ? hc - 1 : hc;
    const double volts = s.volts;
    const bool up = rm >= volts;
    const bool c_pop = volts > q.pop_ratio * fba, c_hang = q.hang_enable && hba > q.hang_level, c_sv = volts > s.save_volts;
    const bool hcpos = hc > 0, hc0 = hc == 0, dt0 = s.decay_type == 0;
    const int st = s.st;
    const int n0 = c_pop ? 1 : c_hang ? 2 : 3;
    const double m0 = c_pop ? q.fast_decay_mult : c_hang ? 0.0 : q.decay_mult;
    const int n1 = c_sv ? 1 : hcpos ? 2 : dt0 ? 3 : 4;
    const double m1 = c_sv ? q.fast_decay_mult : hcpos ? 0.0 : dt0 ? q.decay_mult : q.hang_decay_mult;
    const int n2 = hc0 ? 4 : 2;
    const double m2 = hc0 ? q.hang_decay_mult : 0.0;
    int nst = st == 0 ? n0 : st == 1 ? n1 : st == 2 ? n2 : st == 3 ? 3 : 4;
    double m = st == 0 ? m0 : st == 1 ? m1 : st == 2 ? m2 : st == 3 ? q.decay_mult : q.hang_decay_mult;
    nst = up ? 0 : nst;
    m = up ? q.attack_mult : m;
    const bool to_hang = !up && st == 0 && !c_pop && c_hang, to_decay = !up && st == 0 && !c_pop && !c_hang;
    hc = to_hang ? q.hang_count_init : hc;
    s.decay_type = to_hang ? 1 : to_decay ? 0 : s.decay_type;
    s.save_volts = (up && st >= 2) ? volts : s.save_volts;
    double v = __builtin_fma(rm - volts, m, volts);
    v = v < q.min_volts ? q.min_volts : v;
    s.volts = v; s.st = nst; s.hc = hc;
}

// sample j of the call as the ring would hold it (scaled by pre_gain), j >= -A: from the rows, or from the ring of the calls before
__device__ __forceinline__ double2 agc_sample(const double2 *x, const AgcState *sp, int A, int j, double pre_gain)
{
    if (j >= 0) { const double2 z = x[j]; return make_double2(z.x * pre_gain, z.y * pre_gain); }
    return sp->ring[(sp->out_index + A + 1 + j) & (kAgcRing - 1)];              // j = -1: the slot in_index points at
}
__device__ __forceinline__ double agc_mag_of(double2 z, int pmode)
{
    return pmode == 0 ? fmax(fabs(z.x), fabs(z.y)) : sqrt(__builtin_fma(z.x, z.x, z.y * z.y));
}
__device__ __forceinline__ double agc_mag(const double2 *x, const AgcState *sp, const AgcParam &q, int j, double pre_gain)
{
    if (j >= 0) { const double2 z = x[j]; return agc_mag_of(make_double2(z.x * pre_gain, z.y * pre_gain), q.pmode); }
    return sp->abs_ring[(sp->out_index + q.attack_buffsize + 1 + j) & (kAgcRing - 1)];
}

// scratch of channel slot k (index in the launch's list): four arrays of `arr` doubles: RM, fba, hba, volts
__device__ __forceinline__ double *agc_arr(double *scr, long long arr, int slot, int which) { return scr + ((long long)slot * 4 + which) * arr; }

// ---- (a) + (c): sliding maximum and the tiles' halos ---------------------------------------------------------------------
// grid (tiles of kAgcTile, channels), 256 threads.  LDS: magnitudes m[], prefix maxima P[], suffix maxima S[] over blocks of 64,
// for the samples [j0 - Ah, j0 + kAgcTile), Ah = A rounded up to 64.
static __global__ __launch_bounds__(256) void agc_prep_kernel(const double2 *buf, long long stride, int n, const int *chan_list,
                                                              const AgcParam *prm, const AgcState *state, double *scr, long long arr,
                                                              double2 *halo, int halo_pitch, double pre_gain, double *tsum)
{
    extern __shared__ double sm_prep[];
    __shared__ double red[8];
    const int slot = blockIdx.y, ch = chan_list[slot], t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const AgcParam q = prm[ch];
    const AgcState *sp = state + ch;
    const double2 *x = buf + (long long)ch * stride;
    const int A = q.attack_buffsize, Ah = (A + 63) & ~63, M = Ah + kAgcTile, j0 = blockIdx.x * kAgcTile, base = j0 - Ah;
    double *m = sm_prep, *P = m + M, *S = P + M;
    // the A samples ahead of the tile, as the ring would hold them, are kept for agc_apply_kernel, which works in place
    double2 *hl = halo + ((long long)slot * gridDim.x + blockIdx.x) * halo_pitch;
    if (base >= 0) {
        // every sample of the window is in the rows (all tiles but the call's first): five loads per thread in flight at a time -- one per
        // round of the loop is one HBM latency per round, and the kernel was bound by that -- and the halo taken from the same registers
        constexpr int U = 5;
        for (int i0 = 0; i0 < M; i0 += 256 * U) {
            double2 z[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * 256 + t, j = base + i;
                z[u] = x[(i < M && j < n) ? j : base];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * 256 + t, j = base + i;
                if (i >= M) continue;
                const double2 zz = make_double2(z[u].x * pre_gain, z[u].y * pre_gain);
                m[i] = j < n ? agc_mag_of(zz, q.pmode) : 0.0;
                if (i >= Ah - A && i < Ah) hl[i - (Ah - A)] = zz;                  // (j < j0 <= n - 1 there)
            }
        }
    } else {
        for (int i = t; i < M; i += 256) {
            const int j = base + i;
            m[i] = (j >= -A && j < n) ? agc_mag(x, sp, q, j, pre_gain) : 0.0;    // (samples further back than the window are not looked at)
        }
        for (int i = t; i < A; i += 256) hl[i] = agc_sample(x, sp, A, j0 - A + i, pre_gain);
    }
    __syncthreads();
    for (int b = wave; b < M / 64; b += 4) {
        // prefix and suffix maxima of the 64 magnitudes (all >= 0, so the 0 a DPP move reads where it has no source is the identity):
        // inside the rows of 16 by row shifts, across them by row broadcasts (prefix) and three lane reads (suffix)
        const double v = m[b * 64 + lane];
        double p = v, s = v;
        p = fmax(p, dpp_fetch_d<0x111, 0xf>(p)); p = fmax(p, dpp_fetch_d<0x112, 0xf>(p));
        p = fmax(p, dpp_fetch_d<0x114, 0xf>(p)); p = fmax(p, dpp_fetch_d<0x118, 0xf>(p));
        p = fmax(p, dpp_fetch_d<0x142, 0xa>(p)); p = fmax(p, dpp_fetch_d<0x143, 0xc>(p));
        s = fmax(s, dpp_fetch_d<0x101, 0xf>(s)); s = fmax(s, dpp_fetch_d<0x102, 0xf>(s));
        s = fmax(s, dpp_fetch_d<0x104, 0xf>(s)); s = fmax(s, dpp_fetch_d<0x108, 0xf>(s));
        {
            const double r1 = lane_bcast(s, 16), r2 = lane_bcast(s, 32), r3 = lane_bcast(s, 48);
            const double t2 = fmax(r2, r3), t1 = fmax(r1, t2);
            const int row = lane >> 4;
            s = fmax(s, row == 0 ? t1 : row == 1 ? t2 : row == 2 ? r3 : 0.0);
        }
        P[b * 64 + lane] = p; S[b * 64 + lane] = s;
    }
    __syncthreads();
    // the tile's own contribution to the two back-averages of the delayed magnitude a_j = m[Ah - A + k] (sample j - A), from a zero
    // state: sum_k mult (1 - mult)^(T - 1 - k) a_k for the whole tile -- agc_avg_tiled_kernel chains these instead of reading the rows
    // a first time.  Thread t takes k = t + 256 i: Horner in (1 - mult)^256 over i, then (1 - mult)^(255 - t).
    {
        const double mF = q.onemfast_backmult, mH = q.onemhang_backmult;
        const double sF = ipow_d(mF, 256), sH = ipow_d(mH, 256);
        double aF = 0.0, aH = 0.0;
#pragma unroll
        for (int i = 0; i < kAgcTile / 256; i++) {
            const int k = t + 256 * i;
            const double a = j0 + k < n ? m[Ah - A + k] : 0.0;
            aF = __builtin_fma(aF, sF, a); aH = __builtin_fma(aH, sH, a);
        }
        const double wF = wave_sum_d(aF * ipow_d(mF, 255 - t)), wH = wave_sum_d(aH * ipow_d(mH, 255 - t));
        if (lane == 0) { red[wave * 2] = wF; red[wave * 2 + 1] = wH; }
        __syncthreads();
        if (t == 0) {
            double *o = tsum + ((long long)slot * gridDim.x + blockIdx.x) * 2;
            o[0] = q.fast_backmult * ((red[0] + red[2]) + (red[4] + red[6]));
            o[1] = q.hang_backmult * ((red[1] + red[3]) + (red[5] + red[7]));
        }
    }
    double *rm = agc_arr(scr, arr, slot, 0);
    for (int k = t; k < kAgcTile; k += 256) {
        const int j = j0 + k;
        if (j >= n) break;
        const int hi = Ah + k, lo = hi - A + 1, blo = lo >> 6, bhi = hi >> 6;
        double r;
        if (blo == bhi) {
            r = m[lo];
            for (int i = lo + 1; i <= hi; i++) r = fmax(r, m[i]);
        } else {
            r = fmax(S[lo], P[hi]);
            for (int b = blo + 1; b < bhi; b++) r = fmax(r, P[b * 64 + 63]);
        }
        rm[j] = r;
    }
}

// ---- (b): the two back-averages of the delayed magnitude ---------------------------------------------------------------------
// One pass: the time segments are cut on the prep kernel's tile boundaries (kAgcTile samples) and a segment's carry-in is the chain over
// the tiles ahead of it (tsum [slot][ntile][2], agc_prep_kernel); a partial last tile is nobody's predecessor.
static __global__ __launch_bounds__(kSegThreads) void agc_avg_tiled_kernel(const double2 *buf, long long stride, int n, const int *chan_list,
                                                                           const AgcParam *prm, const AgcState *state, double *scr,
                                                                           long long arr, const double *tsum, int ntile, double pre_gain)
{
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = kSegWaves * (int)gridDim.y, sidx = (int)blockIdx.y * kSegWaves + wave;
    const AgcParam q = prm[ch];
    const AgcState *sp = state + ch;
    const double2 *x = buf + (long long)ch * stride;
    const int A = q.attack_buffsize;
    constexpr int lb = kAgcTile / 64;
    const int nb = (n + 63) >> 6, nt = (n + kAgcTile - 1) / kAgcTile;
    const int t0 = (int)((long long)sidx * nt / S), t1 = (int)((long long)(sidx + 1) * nt / S);
    const int b0 = t0 * lb, b1 = t1 * lb < nb ? t1 * lb : nb;
    const double mF = q.onemfast_backmult, mH = q.onemhang_backmult;
    double cF = sp->fast_backaverage, cH = sp->hang_backaverage;
    {
        // c = c_in mT^t0 + sum_{t < t0} mT^(t0 - 1 - t) s_t, 64 tiles per step
        const double tF = ipow_d(mF, kAgcTile), tH = ipow_d(mH, kAgcTile);
        const double *e = tsum + (long long)slot * ntile * 2;
        for (int blk = 0; blk < t0; blk += 64) {
            const int cnt = t0 - blk < 64 ? t0 - blk : 64;
            double vF = 0.0, vH = 0.0;
            if (lane < cnt) {
                const double2 sv = *reinterpret_cast<const double2 *>(e + (long long)(blk + lane) * 2);
                vF = sv.x * ipow_d(tF, cnt - 1 - lane); vH = sv.y * ipow_d(tH, cnt - 1 - lane);
            }
            cF = __builtin_fma(cF, ipow_d(tF, cnt), wave_sum_d(vF));
            cH = __builtin_fma(cH, ipow_d(tH, cnt), wave_sum_d(vH));
        }
    }
    const PoleScan sF = make_pole_scan(mF, lane), sH = make_pole_scan(mH, lane);
    double *fo = agc_arr(scr, arr, slot, 1), *ho = agc_arr(scr, arr, slot, 2);
    for (int b = b0; b < b1; b++) {
        const int base = b * 64, j = base + lane, cnt = n - base < 64 ? n - base : 64;
        const double a = j < n ? agc_mag(x, sp, q, j - A, pre_gain) : 0.0;
        const double f = scan_pole_dpp(q.fast_backmult * a, sF) + sF.pw * cF;
        const double h = scan_pole_dpp(q.hang_backmult * a, sH) + sH.pw * cH;
        if (lane < cnt) { fo[j] = f; ho[j] = h; }
        cF = lane_bcast(f, cnt - 1); cH = lane_bcast(h, cnt - 1);
    }
}

// ---- (d), first the detector's state at every tile boundary ---------------------------------------------------------------------
// The detector spends its time in a few regimes -- attacking towards ring_max from below, decaying towards it from above, holding
// while the hang counter runs -- and inside a regime it is a LINEAR one-pole recurrence driven by ring_max.  One wavefront per channel
// takes 64 samples of the three streams at a time and advances regime by regime (agc_chunk): a wave-wide scan where the state stays
// put, single reference steps (agc_lane_step) at the turns.  What comes out are the states at the tile boundaries, within rounding of
// what sample-by-sample stepping gives -- the lanes then step every tile exactly from there, and the check of every tile's end against
// the next boundary (agc_verify_kernel) catches what a scan got wrong.
// One chunk of cnt <= 64 samples (ring_max r in the lanes): the detector advances REGIME by regime.  While it attacks (state 0,
// ring_max >= volts), decays (state 3 or 4, ring_max < volts), decays fast (state 1, ring_max < volts, volts still above save_volts) or
// holds (state 2, counter running) it is a linear recurrence in volts with one multiplier whatever ring_max does, so the rest of the
// chunk is taken as a wave-wide scan,
//   v_j = (1 - m)^(j - p + 1) v + sum_{i = p .. j} m r_i (1 - m)^(j - i),
// a ballot finds the first sample at which the regime's own condition (on v_{j-1}) fails, everything ahead of it is accepted at once,
// and that sample is one reference step (agc_lane_step).  tab[w][k] = (1 - mult_w)^k for the attack, decay, hang-decay and fast-decay
// multipliers.  The two back-averages only matter to a step out of state 0 (the choice between fast decay, hang and decay): they are
// fetched there, f0 / h0 pointing at the chunk's first sample, and not streamed beside ring_max.
#ifdef QH_AGC_COUNT     // experiment builds: how the walk spends its rounds ([w]: scans of regime w, [6 + st]: single steps out of state st,
                        // [11 + w]: samples the scans of regime w covered)
static __device__ unsigned long long g_agc_count[20];
#define QH_AGC_CNT(i, v) do { if (lane == 0) atomicAdd(&g_agc_count[i], (unsigned long long)(v)); } while (0)
#else
#define QH_AGC_CNT(i, v) do { } while (0)
#endif
struct AgcScans { PoleScan a, d, hd, fd; };         // attack, decay, hang decay, fast decay
__device__ __forceinline__ AgcScans agc_scans_make(const AgcParam &q, int lane)
{
    return AgcScans{ make_pole_scan(1.0 - q.attack_mult, lane), make_pole_scan(1.0 - q.decay_mult, lane),
                     make_pole_scan(1.0 - q.hang_decay_mult, lane), make_pole_scan(1.0 - q.fast_decay_mult, lane) };
}
__device__ __forceinline__ void agc_chunk(AgcLane &s, double r, const double *f0, const double *h0, int cnt, int lane, const AgcParam &q,
                                          const AgcScans &sc4, const double (*tab)[65])
{
    int p = 0;
    while (p < cnt) {
        const double rp = lane_bcast(r, p);
        const bool up = rp >= s.volts;
        // (a decay that sits on the min_volts clamp holds there while ring_max stays below it: silence)
        // (state 1 holds there too while volts stays above save_volts: the fast decay of silence)
        const bool fast = s.st == 1 && !up && s.volts > s.save_volts;
        const bool floor = !up && s.volts <= q.min_volts && (s.st == 3 || s.st == 4 || fast);
        // 0 attack, 1 decay, 2 hang decay, 3 fast decay: scans; 4 hold, 5 floor: volts stays
        const int w = floor ? 5 : (s.st == 0 && up) ? 0 : (s.st == 3 && !up) ? 1 : (s.st == 4 && !up) ? 2 : fast ? 3 : (s.st == 2 && !up && s.hc > 1) ? 4 : -1;
        int k = p;                                              // first sample the regime does not cover
        if (w >= 0 && w < 4) {
            const double m = w == 0 ? q.attack_mult : w == 1 ? q.decay_mult : w == 2 ? q.hang_decay_mult : q.fast_decay_mult;
            const bool in = lane >= p && lane < cnt;
            // (the decay is 19 scans in 20: its weights straight from their registers; the others field by field -- an index or a select
            // between the structs themselves would be a select of addresses, i.e. scratch memory)
            double sc;
            if (w == 1) sc = scan_pole_dpp(in ? m * r : 0.0, sc4.d);
            else {
                PoleScan pq;
#define QH_SEL3(f) pq.f = w == 0 ? sc4.a.f : w == 2 ? sc4.hd.f : sc4.fd.f
                QH_SEL3(m1); QH_SEL3(m2); QH_SEL3(m4); QH_SEL3(m8); QH_SEL3(pa); QH_SEL3(pb); QH_SEL3(pw);
#undef QH_SEL3
                sc = scan_pole_dpp(in ? m * r : 0.0, pq);
            }
            const double v = __builtin_fma(tab[w][in ? lane - p + 1 : 0], s.volts, sc);
            double vb = wave_shr1(v);
            if (lane == p) vb = s.volts;
            // (the clamp is a turn: the reference applies it after every step; the fast decay ends where volts has come down to save_volts)
            const bool ok = (w == 0 ? r >= vb : r < vb) && v >= q.min_volts && (w != 3 || vb > s.save_volts);
            const unsigned long long bad = __ballot(in && !ok);
            k = bad ? __ffsll((long long)bad) - 1 : cnt;
            if (k > p) s.volts = lane_bcast(v, k - 1);
        } else if (w >= 4) {
            const bool in = lane >= p && lane < cnt;
            const unsigned long long bad = __ballot(in && r >= s.volts);
            k = bad ? __ffsll((long long)bad) - 1 : cnt;
            if (w == 4 && k - p > s.hc - 1) k = p + s.hc - 1;   // the counter runs out: that sample is a turn
        }
        if (w >= 0) { QH_AGC_CNT(w, 1); QH_AGC_CNT(11 + w, k - p); }
        if (k > p) { s.hc = s.hc > k - p ? s.hc - (k - p) : 0; p = k; }
        else {
            QH_AGC_CNT(6 + s.st, 1);
            double f = 0.0, h = 0.0;
            if (s.st == 0) { f = f0[p]; h = h0[p]; }            // (uniform: every lane the same word)
            agc_lane_step(s, rp, f, h, q);
            p++;
        }
    }
}

// bounds[slot][tile][0..4]: the state at the START of tile t (tile 0: the carried state).
struct AgcWalk {
    const double *in0, *in1, *in2;
    double *bo;
    int n, L, lane;
};
__device__ __forceinline__ void agc_put(double *o, const AgcLane &s)
{
    o[0] = s.volts; o[1] = s.save_volts; o[2] = (double)s.hc; o[3] = (double)s.decay_type; o[4] = (double)s.st;
}
__device__ __forceinline__ bool agc_state_differs(const double *a, const double *b);
// The walk over [j0, j1) (multiples of 64, j0 a tile boundary or a warm-up start), boundary states written from sample `from` on; ring_max
// travels kAgcAhead chunks ahead of the one being walked (a chunk is ~1000 cycles of work, a fetch from HBM under load two to four times
// that).  AGAIN: the segment has been walked before from another start state: at every tile boundary behind the first the state is held
// against the one that walk left there -- once they agree the two walks have met and everything further on stands as it is (true is
// returned: the segment's recorded end is still right).
constexpr int kAgcAhead = 4;
template <bool AGAIN>
__device__ __forceinline__ bool agc_walk(AgcLane &s, const AgcWalk &w, int j0, int from, int j1, const AgcParam &q, const AgcScans &sc4,
                                         const double (*tab)[65])
{
    const int lane = w.lane;
    double rq[kAgcAhead];
#pragma unroll
    for (int u = 0; u < kAgcAhead; u++) rq[u] = j0 + u * 64 + lane < w.n ? w.in0[j0 + u * 64 + lane] : 0.0;
    // the next tile boundary at or behind `from` (j0, from and L are multiples of 64: the chunks land on it), kept by addition -- a
    // division per chunk is forty instructions of a walk that is one wavefront's instruction stream
    int tile = ((from > j0 ? from : j0) + w.L - 1) / w.L, nextb = tile * w.L;
    for (int base0 = j0; base0 < j1; base0 += 64 * kAgcAhead) {
#pragma unroll
        for (int u = 0; u < kAgcAhead; u++) {
            const int base = base0 + u * 64;
            if (base >= j1) break;
            const int cnt = w.n - base < 64 ? w.n - base : 64;
            if (base == nextb) {
                double *o = w.bo + (long long)tile * 8;
                if (AGAIN && base > j0) {
                    const double mine[5] = { s.volts, s.save_volts, (double)s.hc, (double)s.decay_type, (double)s.st };
                    if (!agc_state_differs(mine, o)) return true;        // (uniform: every lane holds the same state and reads the same words)
                }
                if (lane == 0) agc_put(o, s);
                tile++; nextb += w.L;
            }
            const double r = rq[u];
            const int nx = base + 64 * kAgcAhead + lane;
            rq[u] = nx < w.n ? w.in0[nx] : 0.0;
            agc_chunk(s, r, w.in1 + base, w.in2 + base, cnt, lane, q, sc4, tab);
        }
    }
    return false;
}
__device__ __forceinline__ void agc_tab_init(double (*tab)[65], const AgcParam &q, int lane)
{
    const double lg0 = log1p(-q.attack_mult), lg1 = log1p(-q.decay_mult), lg2 = log1p(-q.hang_decay_mult), lg3 = log1p(-q.fast_decay_mult);
    tab[0][lane + 1] = exp((double)(lane + 1) * lg0); tab[1][lane + 1] = exp((double)(lane + 1) * lg1);
    tab[2][lane + 1] = exp((double)(lane + 1) * lg2); tab[3][lane + 1] = exp((double)(lane + 1) * lg3);
    if (lane == 0) tab[0][0] = tab[1][0] = tab[2][0] = tab[3][0] = 1.0;
    __syncthreads();
}

// One wavefront per channel walks the call alone at a few thousand cycles per 64 samples, so the call is cut into K super-segments
// (seg samples each, a multiple of the tile length) that K wavefronts walk at once: segment 0 from the carried state; segment k > 0
// from the state the CALL began in, W samples ahead of its own start (W = 0 unless QH_AGC_WARM asks for a warm-up).  Where the level is
// steady such a walk ends on the true trajectory some 10 000 samples in, so most segments END right although they began wrong; after a
// drop in level it does not while the decay lasts.  The repair rounds below put that right.
// sege[slot][k][0..4]: the state at the END of segment k.  grid (channels, K), one wavefront.
static __global__ __launch_bounds__(64) void agc_bounds_kernel(int n, const int *chan_list, const AgcParam *prm, const AgcState *state,
                                                              const double *scr, long long arr, double *bounds, long long bstride, int L,
                                                              int seg, int W, double *sege)
{
    const int slot = blockIdx.x, k = blockIdx.y, ch = chan_list[slot], lane = threadIdx.x;
    const int begin = k * seg;
    if (begin >= n) return;
    const AgcParam q = prm[ch];
    const AgcState *sp = state + ch;
    AgcWalk w{ scr + ((long long)slot * 4 + 0) * arr, scr + ((long long)slot * 4 + 1) * arr, scr + ((long long)slot * 4 + 2) * arr,
               bounds + (long long)slot * bstride, n, L, lane };
    __shared__ double tab[4][65];
    agc_tab_init(tab, q, lane);
    const AgcScans sc4 = agc_scans_make(q, lane);
    AgcLane s{ sp->volts, sp->save_volts, sp->hang_counter, sp->decay_type, sp->state };
    int j0 = begin - W;
    if (j0 <= 0) j0 = 0;
    else s.hc = s.hc > j0 ? s.hc - j0 : 0;                  // the hang counter has run down meanwhile
    const int end = begin + seg < n ? begin + seg : n;
    agc_walk<false>(s, w, j0, begin, end, q, sc4, tab);
    if (lane == 0) agc_put(sege + ((long long)slot * gridDim.y + k) * 8, s);
}

// One ROUND of repair over all segments at once, grid (channels, K), one wavefront: segment k began in bounds[tile of k seg]; where that is
// not the state segment k - 1 ended in AS OF THE ROUND BEFORE (sege_in), it is walked again from there until it meets its old self.  A
// round reads sege_in and writes sege_out, so that no block reads an end another is writing.  The ends of the round before may themselves
// be off (their segments are being walked again beside this one), so a chain of r segments that all depend on their start -- a decay
// that runs across r boundaries -- takes r rounds; agc_bounds_fix_kernel behind the rounds catches what is left, in order.
static __global__ __launch_bounds__(64) void agc_bounds_round_kernel(int n, const int *chan_list, const AgcParam *prm, const double *scr, long long arr,
                                                                    double *bounds, long long bstride, int L, int seg, int K, const double *sege_in,
                                                                    double *sege_out, int *nseg_fixed)
{
    const int slot = blockIdx.x, k = blockIdx.y, ch = chan_list[slot], lane = threadIdx.x;
    const int begin = k * seg;
    if (begin >= n) return;
    const double *mine_in = sege_in + ((long long)slot * K + k) * 8;
    double *mine_out = sege_out + ((long long)slot * K + k) * 8;
    double *bo = bounds + (long long)slot * bstride;
    bool again = false;
    if (k > 0) again = agc_state_differs(bo + (long long)(begin / L) * 8, mine_in - 8);
    if (!again) {
        if (lane < 5) mine_out[lane] = mine_in[lane];
        return;
    }
    const AgcParam q = prm[ch];
    AgcWalk w{ scr + ((long long)slot * 4 + 0) * arr, scr + ((long long)slot * 4 + 1) * arr, scr + ((long long)slot * 4 + 2) * arr, bo, n, L, lane };
    __shared__ double tab[4][65];
    agc_tab_init(tab, q, lane);
    const AgcScans sc4 = agc_scans_make(q, lane);
    const double *b = mine_in - 8;
    AgcLane s{ b[0], b[1], (int)b[2], (int)b[3], (int)b[4] };
    const int end = begin + seg < n ? begin + seg : n;
    const bool met = agc_walk<true>(s, w, begin, begin, end, q, sc4, tab);
    if (met) { if (lane < 5) mine_out[lane] = mine_in[lane]; }
    else if (lane == 0) agc_put(mine_out, s);
    if (lane == 0 && nseg_fixed) atomicAdd(nseg_fixed, 1);
}
// One wavefront per channel, in order: segment k began in bounds[tile of k seg]; where that is not the state segment k - 1 ended in, the
// segment is walked again from there (its tiles' boundary states and its own end rewritten, until it meets its old self).
// nseg_fixed counts them.
static __global__ __launch_bounds__(64) void agc_bounds_fix_kernel(int n, const int *chan_list, const AgcParam *prm, const double *scr, long long arr,
                                                                  double *bounds, long long bstride, int L, int seg, int K, double *sege, int *nseg_fixed)
{
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x;
    double *bo = bounds + (long long)slot * bstride;
    int fixed = 0;
    bool ready = false;
    __shared__ double tab[4][65];
    AgcParam q;
    AgcScans sc4;
    for (int k = 1; k < K && k * seg < n; k++) {
        const double *a = bo + (long long)(k * seg / L) * 8, *b = sege + ((long long)slot * K + k - 1) * 8;
        if (!agc_state_differs(a, b)) continue;
        if (!ready) {
            q = prm[ch];
            agc_tab_init(tab, q, lane);
            sc4 = agc_scans_make(q, lane);
            ready = true;
        }
        AgcWalk w{ scr + ((long long)slot * 4 + 0) * arr, scr + ((long long)slot * 4 + 1) * arr, scr + ((long long)slot * 4 + 2) * arr, bo, n, L, lane };
        AgcLane s{ b[0], b[1], (int)b[2], (int)b[3], (int)b[4] };
        const int begin = k * seg, end = begin + seg < n ? begin + seg : n;
        const bool met = agc_walk<true>(s, w, begin, begin, end, q, sc4, tab);
        if (!met && lane == 0) agc_put(sege + ((long long)slot * K + k) * 8, s);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        fixed++;
    }
    if (lane == 0 && fixed && nseg_fixed) atomicAdd(nseg_fixed, fixed);
}

// ---- (d): one lane per tile ------------------------------------------------------------------------------------------------
// grid (groups of 64 tiles, channels), one wavefront.  L = tile length, H = warm-up, both multiples of kAgcBatch.
static __global__ __launch_bounds__(64, 2) void agc_lanes_kernel(int n, const int *chan_list, const AgcParam *prm, double *scr, long long arr,
                                                                const double *bounds, long long bstride, double *ends, long long estride, int L)
{
    __shared__ double lds[3 * 64 * kAgcPitch];
    const int slot = blockIdx.y, ch = chan_list[slot], lane = threadIdx.x, group = blockIdx.x;
    const long long tile0 = (long long)group * 64 * L;
    if (tile0 >= n) return;
    const AgcParam q = prm[ch];
    const double *in0 = agc_arr(scr, arr, slot, 0), *in1 = agc_arr(scr, arr, slot, 1), *in2 = agc_arr(scr, arr, slot, 2);
    double *vo = agc_arr(scr, arr, slot, 3);
    const long long s0 = tile0 + (long long)lane * L;
    const bool live = s0 < n;
    AgcLane s{ 0.0, 0.0, 0, 0, 0 };
    if (live) {
        const double *b = bounds + (long long)slot * bstride + ((long long)group * 64 + lane) * 8;
        s.volts = b[0]; s.save_volts = b[1]; s.hc = (int)b[2]; s.decay_type = (int)b[3]; s.st = (int)b[4];
    }
    constexpr int B = kAgcBatch, RPI = 64 / B;
    const int frow = lane / B, fcol = lane % B;
    const int RL = RPI * L, loff0 = frow * L + fcol, gb0 = (int)tile0;
    double *e = ends + (long long)slot * estride + ((long long)group * 64 + lane) * 8;
    // the next batch's 3 x B values per lane travel while the current batch is stepped
    double t0n[B], t1n[B], t2n[B];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int off = loff0 + j * RL, g = gb0 + i0 + off;
            const bool ok = g < n;
            t0n[j] = ok ? in0[g] : 0.0; t1n[j] = ok ? in1[g] : 0.0; t2n[j] = ok ? in2[g] : 0.0;
        }
    };
    fetch(0);
    for (int i0 = 0; i0 < L; i0 += B) {
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int at = (RPI * j + frow) * kAgcPitch + fcol;
            lds[at] = t0n[j]; lds[64 * kAgcPitch + at] = t1n[j]; lds[2 * 64 * kAgcPitch + at] = t2n[j];
        }
        if (i0 + B < L) fetch(i0 + B);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double *row = lds + lane * kAgcPitch;
        double vv[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            if (live && s0 + i0 + k < n) agc_lane_step(s, row[k], row[64 * kAgcPitch + k], row[2 * 64 * kAgcPitch + k], q);
            vv[k] = s.volts;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < B; k++) row[k] = vv[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int off = loff0 + j * RL, g = gb0 + i0 + off;
            if (g < n) vo[g] = lds[(RPI * j + frow) * kAgcPitch + fcol];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live) { e[0] = s.volts; e[1] = s.save_volts; e[2] = (double)s.hc; e[3] = (double)s.decay_type; e[4] = (double)s.st; }
}

// A boundary state is accepted when it is the state the tile before it ended in: the discrete part exactly, volts to 1e-11 and
// save_volts -- volts as it was at the last turn to state 0 -- to 1e-9 (the closed-form jumps land within 1e-14 of the stepped values;
// the misses of a call show in qh_rxa_agc_repairs).
__device__ __forceinline__ bool agc_state_differs(const double *a, const double *b)
{
    return !(fabs(a[0] - b[0]) <= 1e-11 * fabs(b[0]) && fabs(a[1] - b[1]) <= 1e-9 * fabs(b[1]) && a[2] == b[2] && a[3] == b[3] && a[4] == b[4]);
}

// One wavefront per channel walks the tiles in order: tile t began in bounds[t]; it is right when that is the state tile t - 1 ended in
// (ends[t - 1], after any repair).  A tile that is not is stepped again from there, 64 samples per round, which also gives the end state
// its successor is judged against.  fin[slot][0..4] = the state after the call's last sample.
static __global__ __launch_bounds__(64) void agc_verify_kernel(int n, const int *chan_list, const AgcParam *prm, double *scr, long long arr,
                                                              const double *bounds, long long bstride, double *ends, long long estride, int L,
                                                              double *fin, int *nfixed, int check_only)
{
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x;
    const AgcParam q = prm[ch];
    const double *in0 = agc_arr(scr, arr, slot, 0), *in1 = agc_arr(scr, arr, slot, 1), *in2 = agc_arr(scr, arr, slot, 2);
    double *vo = agc_arr(scr, arr, slot, 3);
    double *e = ends + (long long)slot * estride;
    const double *bd = bounds + (long long)slot * bstride;
    const int ntiles = (n + L - 1) / L;
    int fixed = 0;
    for (int t = 1; t < ntiles; t++) {
        const double *a = bd + (long long)t * 8, *b = e + (long long)(t - 1) * 8;
        if (!agc_state_differs(a, b)) continue;                 // (uniform: every lane reads the same words)
        if (check_only) { fixed++; continue; }                  // diagnostics: count, leave everything as it is
        AgcLane s{ b[0], b[1], (int)b[2], (int)b[3], (int)b[4] };
        const long long s0 = (long long)t * L;
        const int len = (int)((long long)n - s0 < L ? (long long)n - s0 : L);
        for (int off = 0; off < len; off += 64) {
            const int cnt = len - off < 64 ? len - off : 64;
            double r = 0.0, f = 0.0, h = 0.0;
            if (lane < cnt) { r = in0[s0 + off + lane]; f = in1[s0 + off + lane]; h = in2[s0 + off + lane]; }
            double mine = 0.0;
            for (int i = 0; i < cnt; i++) {
                agc_lane_step(s, lane_bcast(r, i), lane_bcast(f, i), lane_bcast(h, i), q);
                if (lane == i) mine = s.volts;
            }
            if (lane < cnt) vo[s0 + off + lane] = mine;
        }
        if (lane == 0) {
            double *w = e + (long long)t * 8;
            w[0] = s.volts; w[1] = s.save_volts; w[2] = (double)s.hc; w[3] = (double)s.decay_type; w[4] = (double)s.st;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the next tile's compare reads what lane 0 has just written
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        fixed++;
    }
    if (lane == 0 && n > 0) {
        const double *b = e + (long long)(ntiles - 1) * 8;
        double *o = fin + (long long)slot * 8;
        for (int k = 0; k < 5; k++) o[k] = b[k];
        if (fixed && nfixed) atomicAdd(nfixed, fixed);
    }
}

// ---- (e): gain curve and multiply, in place ------------------------------------------------------------------------------------
// grid (tiles of kAgcTile, channels), 256 threads; LDS: the tile's samples behind the A samples ahead of it (agc_prep_kernel kept them)
// dst != null: the AGC is the channel's last stage -- the output matrix (xpanel) is applied here and the result goes to the caller's rows
static __global__ __launch_bounds__(256) void agc_apply_kernel(double2 *buf, long long stride, int n, const int *chan_list, const AgcParam *prm,
                                                               const double *scr, long long arr, const double2 *halo, int halo_pitch,
                                                               double pre_gain, double2 *dst = nullptr, long long dst_stride = 0,
                                                               const EpiParam *epi = nullptr)
{
    extern __shared__ double2 sm_apply[];
    const int slot = blockIdx.y, ch = chan_list[slot], t = threadIdx.x;
    const AgcParam q = prm[ch];
    double2 *x = buf + (long long)ch * stride;
    const int A = q.attack_buffsize, j0 = blockIdx.x * kAgcTile;
    const double2 *hl = halo + ((long long)slot * gridDim.x + blockIdx.x) * halo_pitch;
    for (int i = t; i < A; i += 256) sm_apply[i] = hl[i];
    for (int k = t; k < kAgcTile; k += 256) {
        const int j = j0 + k;
        double2 z = make_double2(0.0, 0.0);
        if (j < n) { z = x[j]; z.x *= pre_gain; z.y *= pre_gain; }
        sm_apply[A + k] = z;
    }
    __syncthreads();
    const double *vo = scr + ((long long)slot * 4 + 3) * arr;
    for (int k = t; k < kAgcTile; k += 256) {
        const int j = j0 + k;
        if (j >= n) break;
        const double v = vo[j];
        const double mult = __builtin_fma(-q.slope_constant, fmin(0.0, log10(q.inv_max_input * v)), q.out_target) / v;
        const double2 o = sm_apply[k];                          // sample j - A
        const double2 y = make_double2(o.x * mult, o.y * mult);
        if (dst) {
            EpiParam ep{ 1, 0, 0, 1 };
            if (epi) ep = epi[ch];
            dst[(long long)ch * dst_stride + j] = make_double2(ep.a * y.x + ep.b * y.y, ep.c * y.x + ep.d * y.y);
        } else x[j] = y;
    }
}

// The state the next call starts from.  One workgroup per channel, AFTER agc_apply_kernel -- which has overwritten the rows: the last A
// input samples come out of `tail` (kept by the caller ahead of the apply: the last A samples of every row, scaled).
static __global__ __launch_bounds__(256) void agc_tail_kernel(const double2 *buf, long long stride, int n, const int *chan_list,
                                                              const AgcParam *prm, double2 *tail, double pre_gain)
{
    const int slot = blockIdx.x, ch = chan_list[slot];
    const int A = prm[ch].attack_buffsize;
    const double2 *x = buf + (long long)ch * stride;
    for (int i = threadIdx.x; i < A; i += 256) {
        const double2 z = x[n - A + i];
        tail[(long long)slot * kAgcRing + i] = make_double2(z.x * pre_gain, z.y * pre_gain);
    }
}
static __global__ __launch_bounds__(256) void agc_finish_kernel(int n, const int *chan_list, const AgcParam *prm, AgcState *state,
                                                                const double *scr, long long arr, const double2 *tail, const double *fin)
{
    const int slot = blockIdx.x, ch = chan_list[slot], t = threadIdx.x;
    const AgcParam q = prm[ch];
    AgcState *sp = state + ch;
    const int A = q.attack_buffsize;
    const int oi = (sp->out_index + n) & (kAgcRing - 1);
    // the window (out, in]: samples n - A .. n - 1 at slots out + 1 .. out + A
    for (int i = t; i < A; i += 256) {
        const double2 z = tail[(long long)slot * kAgcRing + i];
        const int s = (oi + 1 + i) & (kAgcRing - 1);
        sp->ring[s] = z;
        sp->abs_ring[s] = agc_mag_of(z, q.pmode);
    }
    __syncthreads();
    if (t == 0) {
        const double *f = fin + (long long)slot * 8;
        sp->ring_max = scr[((long long)slot * 4 + 0) * arr + n - 1];
        sp->fast_backaverage = scr[((long long)slot * 4 + 1) * arr + n - 1];
        sp->hang_backaverage = scr[((long long)slot * 4 + 2) * arr + n - 1];
        sp->volts = f[0]; sp->save_volts = f[1]; sp->hang_counter = (int)f[2]; sp->decay_type = (int)f[3]; sp->state = (int)f[4];
        sp->gain = f[0] * q.inv_out_target;
        sp->out_index = oi;
    }
}

}  // namespace qh
