// qh_tiled.hpp -- the per-channel recurrences of the WDSP detectors, parallel in TIME as well as over channels.
//
// The reference steps every one of them sample by sample (xamd wdsp/amd.c:131-146, xfmd wdsp/fmd.c:151-172, xsnotch
// wdsp/iir.c:76-95).  On the GPU a channel's 2^16 ... 2^20 samples per call would leave one wavefront per channel busy for
// milliseconds while a thousand SIMDs idle (round 1: 85 FM channels = 85 wavefronts, 5.5 ms).  Two facts remove the
// serialisation:
//
//  * LINEAR recurrences (the AM fade leveller's two one-pole averages, the FM detector's DC removal, the CTCSS bi-quad)
//    have a state that is an affine function of the state at any earlier time: a workgroup of 16 wavefronts cuts the call
//    into 16 time segments, every wavefront first runs its segment from a ZERO state and keeps the end value, the end
//    values are chained (C <- T^len C + e, 15 steps), and a second pass over the segment starts from the true carry.  Exact:
//    the same additions in a different association.  Inside a wavefront the scan over 64 consecutive samples uses DPP row
//    shifts / row broadcasts (18 instructions) instead of six LDS-crossbar shuffles.
//
//  * The PLL of the FM detector is not linear (phase wrap, frequency clamp) but it is a CONTRACTION: WDSP's loop
//    (zeta 1, omega_N 20 000 rad/s at 48 kHz, RXA.c:199-204) has a double pole at 0.66 per sample, so two runs that start
//    from different states agree to 1e-16 after ~100 samples (a 2 pi slip of one of them only restarts that clock).  The call
//    is cut into tiles of L samples; EVERY LANE of a wavefront owns one tile and steps its own loop state through the tile's
//    samples, beginning H samples early from a zero state (the first tile begins at the carried state and is exact).  The
//    detector angle arg z is taken for all samples up front (pll_theta_kernel); a wavefront stages the angles of its 64 tiles
//    through LDS 64 steps at a time (rows loaded coalesced, read back transposed, pitch 65: conflict free).
#pragma once
#include "qh_demod.hpp"

namespace qh {

// (the segment helpers -- kSegWaves, scan_pole_dpp, seg_range, seg_load ... -- live in qh_wave.hpp: the Quisk-native detectors
// of qh_qdemod.hpp use them too)

// ---- AM envelope detector + fade leveller (xamd mode 0, wdsp/amd.c:131-146), in place, z -> (audio, audio) ------------
// SAM (xamd mode 1 without sideband separation, amd.c:148-232): the detected value is the sample mixed with the VCO phase it
// saw, corr0 = I cos(phs) + Q sin(phs) (amd.c:150-158,169), phs per sample from the time-tiled loop (pll_lanes_kernel<true>);
// the fade leveller behind it is the AM one.
template <bool SAM, int MODE = 0>
static __global__ __launch_bounds__(kSegThreads) void am_detect_tiled_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                                             const int *levelfade, AmState *state, AmParam prm,
                                                                             const double *pt = nullptr, long long ptstride = 0,
                                                                             double *gsum = nullptr, AmState *carry_out = nullptr)
{
    // carry_out: where the call's last segment leaves the new carry.  One workgroup per channel (MODE 0) reads the old carry ahead of
    // its barrier and may overwrite it; the grid forms (MODE 2) have no barrier between a workgroup that starts late and the one that
    // ends early, so their `state` stays read-only for the whole launch and commit_am_kernel moves the new carry in afterwards.
    __shared__ double s_sum[kSegWaves * kSegSumW];
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = MODE == 0 ? kSegWaves : kSegWaves * (int)gridDim.y, sidx = MODE == 0 ? wave : (int)blockIdx.y * kSegWaves + wave;
    double *sum = MODE == 0 ? s_sum : gsum + (long long)blockIdx.x * S * kSegSumW;
    double2 *p = buf + (long long)ch * stride;
    const double *ph = SAM ? pt + (long long)ch * ptstride : nullptr;
    const bool lf = levelfade[ch] != 0;
    int b0, b1;
    seg_range(n, sidx, b0, b1, S);
    const PoleScan sR = make_pole_scan(prm.mtauR, lane), sI = make_pole_scan(prm.mtauI, lane);
    // pass 1: the magnitudes (the output already when the leveller is off) and the segment's response to its own samples.
    // Only the END value is wanted here, so no scan: lane l keeps sum_b u[64 b + l] (m^64)^(B - 1 - b), one FMA per batch,
    // and one weighted wave sum closes the segment.  (The batch that is cut short by the end of the call belongs to the
    // last segment with samples, whose end value nobody reads.)
    const double m64R = lane_pow(prm.mtauR, 64), m64I = lane_pow(prm.mtauI, 64);
    if constexpr (MODE != 2) {
        double accR = 0.0, accI = 0.0;
        double2 zn[kSegGroup];
        seg_load(zn, b0, b1, n, lane, p);
        for (int b = b0; b < b1; b += kSegGroup) {
            double2 z[kSegGroup];
#pragma unroll
            for (int k = 0; k < kSegGroup; k++) z[k] = zn[k];
            seg_load(zn, b + kSegGroup, b1, n, lane, p);        // the next group is on its way while this one is worked on
#pragma unroll
            for (int k = 0; k < kSegGroup; k++) {
                if (b + k >= b1) break;                         // wave-uniform
                const int i = (b + k) * 64 + lane;
                double a;
                if constexpr (SAM) {
                    double sn, cs;
                    sincos((i < n ? ph[i] : 0.0) * kTwoPiRef, &sn, &cs);
                    a = z[k].x * cs + z[k].y * sn;
                } else a = sqrt(z[k].x * z[k].x + z[k].y * z[k].y);
                if (i < n) p[i] = make_double2(a, a);
                accR = __builtin_fma(accR, m64R, prm.onem_mtauR * a);
                accI = __builtin_fma(accI, m64I, prm.onem_mtauI * a);
            }
        }
        if (!lf) return;                                    // block-uniform
        const double eR = wave_sum_d(accR * lane_pow(prm.mtauR, 63 - lane)), eI = wave_sum_d(accI * lane_pow(prm.mtauI, 63 - lane));
        if (lane == 0) { sum[sidx * kSegSumW] = eR; sum[sidx * kSegSumW + 1] = eI; }
        if constexpr (MODE == 1) return;
    } else if (!lf) return;
    const double st_dc = state[ch].dc, st_dci = state[ch].dc_insert;     // read ahead of the barrier: the last wavefront stores the new carry at its end
    if constexpr (MODE == 0) __syncthreads();
    // the true state at the start of this segment: dc <- mtau^len dc + e over the segments before it (whole batches each: the
    // ragged batch at the end of the call belongs to the last segment with samples)
    double cR = st_dc, cI = st_dci;
    {
        const int q = ((n + 63) >> 6) / S;                  // a segment has q or q + 1 batches
        const double tR0 = pow(m64R, (double)q), tI0 = pow(m64I, (double)q), tR1 = tR0 * m64R, tI1 = tI0 * m64I;
        SegWalk walk(n, S);
        for (int w = 0; w < sidx; w++) {
            const int nbw = walk.next();
            if (nbw == 0) continue;
            cR = __builtin_fma(cR, nbw == q ? tR0 : tR1, sum[w * kSegSumW]);
            cI = __builtin_fma(cI, nbw == q ? tI0 : tI1, sum[w * kSegSumW + 1]);
        }
    }
    // pass 2
    double an[kSegGroup];
    seg_load_re(an, b0, b1, n, lane, p);
    for (int b = b0; b < b1; b += kSegGroup) {
        double av[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) av[k] = an[k];
        seg_load_re(an, b + kSegGroup, b1, n, lane, p);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
            const double a = av[k];
            const double dc = scan_pole_dpp(prm.onem_mtauR * a, sR) + sR.pw * cR;       // amd.c:136-137
            const double di = scan_pole_dpp(prm.onem_mtauI * a, sI) + sI.pw * cI;
            const double audio = a + (di - dc);                                          // amd.c:138
            if (lane < cnt) p[base + lane] = make_double2(audio, audio);
            cR = lane_bcast(dc, cnt - 1); cI = lane_bcast(di, cnt - 1);
        }
    }
    int last = S - 1;                                   // the segment that holds the last sample of the call
    while (last > 0 && seg_samples_of(n, last, S) == 0) last--;
    if (sidx == last && lane == 0 && n > 0) { AmState *o = carry_out ? carry_out : state; o[ch].dc = cR; o[ch].dc_insert = cI; }
}

// The carries that a grid-segmented launch left in slots of their own (see am_detect_tiled_kernel), moved into the state the next
// call reads: one thread per listed channel, the predicate the writer had.
static __global__ void commit_am_kernel(AmState *state, const AmState *next, const int *chan_list, int count, const int *levelfade)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int ch = chan_list[i];
    if (levelfade[ch] != 0) state[ch] = next[ch];
}

// c_(t+1) = mL c_t + s_t over the whole tiles 0 .. nfull - 1 of a channel by ONE wavefront: put(t + 1, c_(t+1)) for every tile, returns
// c_nfull.  A lane takes K consecutive tiles (their K loads in flight together), runs them from a zero state, the lanes' ends meet in one
// DPP scan with the pole mL^K, and the value carried into a lane goes onto its K local ones.  (The form before this one -- 64 tiles per
// round, a dependent load and a scan per round -- took 32 rounds for config 4's 2048 tiles: 0.5 ms on the AM channels' critical path
// with the chip busy, profiles/r06_d_c4_timeline_after.txt.)
template <int K, typename SF, typename PF>
__device__ __forceinline__ double chain_tiles(int nfull, double mL, double c, int lane, SF s_at, PF put)
{
    double pwj[K];                                  // mL^(j + 1)
    pwj[0] = mL;
#pragma unroll
    for (int j = 1; j < K; j++) pwj[j] = pwj[j - 1] * mL;
    const PoleScan sc = make_pole_scan(pwj[K - 1], lane);
    const double plane = lane_pow(pwj[K - 1], lane);     // (mL^K)^lane: what the value ahead of lane 0 has become ahead of this lane
    for (int blk = 0; blk < nfull; blk += 64 * K) {
        const int t0 = blk + lane * K;
        double v[K];
#pragma unroll
        for (int j = 0; j < K; j++) v[j] = t0 + j < nfull ? s_at(t0 + j) : 0.0;
#pragma unroll
        for (int j = 1; j < K; j++) v[j] = __builtin_fma(mL, v[j - 1], v[j]);
        const double incl = scan_pole_dpp(v[K - 1], sc);
        const double up = __shfl_up(incl, 1, 64);
        const double cin = __builtin_fma(plane, c, lane ? up : 0.0);
        const int last = (nfull - blk < 64 * K ? nfull - blk : 64 * K) - 1;         // the round's last whole tile, counted from blk
        double pick = 0.0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            v[j] = __builtin_fma(pwj[j], cin, v[j]);
            if (t0 + j < nfull) put(t0 + j + 1, v[j]);
            if (j == last % K) pick = v[j];
        }
        c = lane_bcast(pick, last / K);
    }
    return c;
}

// The fade leveller's two averages at every tile boundary, for osfir_kernel's DET 3 form (qh_osfir.hpp): cin[ch][t] = (dc, dc_insert)
// just ahead of tile t's first sample, c_(t+1) = m^L c_t + s_t with s_t the tile's own end values (tsum).  One wavefront per channel,
// 64 tiles per step; the values behind the call's last sample -- through the tile the call's end cuts short: its local values at that
// sample, `last` -- go straight into the state (the stage that adds the carried share reads cin).  A channel with the leveller off
// (amd.c:134) gets zeros and keeps its state.
static __global__ __launch_bounds__(64) void am_lv_chain_kernel(int n, int L, const int *chan_list, const int *levelfade, AmState *state, AmParam prm,
                                                                 const double *tsum, long long tstride, const double *last, double *cin, long long cstride)
{
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    const int nt = (n + L - 1) / L, nfull = n / L;
    double2 *co = reinterpret_cast<double2 *>(cin) + (long long)ch * cstride;
    if (levelfade[ch] == 0) {
        for (int t = lane; t < nt; t += 64) co[t] = make_double2(0.0, 0.0);
        return;
    }
    const double mR = pow(prm.mtauR, (double)L), mI = pow(prm.mtauI, (double)L);
    const double2 *e = reinterpret_cast<const double2 *>(tsum) + (long long)ch * tstride;
    double cR = state[ch].dc, cI = state[ch].dc_insert;
    if (lane == 0) co[0] = make_double2(cR, cI);
    cR = chain_tiles<16>(nfull, mR, cR, lane, [&](int t) { return e[t].x; }, [&](int t, double v) { if (t < nt) co[t].x = v; });
    cI = chain_tiles<16>(nfull, mI, cI, lane, [&](int t) { return e[t].y; }, [&](int t, double v) { if (t < nt) co[t].y = v; });
    if (nt > nfull) {
        const int len = n - nfull * L;
        cR = __builtin_fma(cR, pow(prm.mtauR, (double)len), last[2 * ch]);
        cI = __builtin_fma(cI, pow(prm.mtauI, (double)len), last[2 * ch + 1]);
    }
    if (lane == 0 && n > 0) { state[ch].dc = cR; state[ch].dc_insert = cI; }
}
// ... and the delay line of the stage behind it (bp1), whose input never exists as complex samples: new_hist[j] <- (audio, audio) of
// sample n - H + j, audio = a_local + cI mI^(k + 1) - cR mR^(k + 1), or the old line's sample where that index is negative.
static __global__ __launch_bounds__(256) void am_audio_hist_kernel(const double *a_local, long long astride, int n, const int *chan_list,
                                                                   const double *cin, long long cstride, const double *pw, int shift,
                                                                   const double2 *old_hist, double2 *new_hist, int H)
{
    const int ch = chan_list[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    const long long g = (long long)n - H + j;
    double2 v;
    if (g >= 0) {
        const int tl = (int)(g >> shift), k = (int)(g - ((long long)tl << shift));
        const double2 c = (reinterpret_cast<const double2 *>(cin) + (long long)ch * cstride)[tl];
        const double a = a_local[(long long)ch * astride + g] + __builtin_fma(c.y, pw[(1 << shift) + k], -c.x * pw[k]);
        v = make_double2(a, a);
    } else v = old_hist[(long long)ch * H + (H + g)];
    new_hist[(long long)ch * H + j] = v;
}

// The AM fade leveller behind an nbp0 stage that left the envelope itself (osfir_kernel DET 2): mag [ch][mstride] doubles, tsum
// [ch][tstride][2] every filter tile's contribution to the two averages (L samples per tile, a multiple of 64).  Segments are cut on
// tile boundaries, a segment's carry-in is the chain over the tiles ahead of it, and there is one pass: 8 bytes read, 16 written
// per sample, against 16 + 16 + 16 + 16 of am_detect_tiled_kernel's two.  out may not be the rows `mag` lies in.
static __global__ __launch_bounds__(kSegThreads) void am_level_tiled_kernel(const double *mag, long long mstride, double2 *out, long long ostride,
                                                                            int n, const int *chan_list, const int *levelfade, const AmState *state,
                                                                            AmParam prm, const double *tsum, long long tstride, int L, AmState *carry_out)
{
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = kSegWaves * (int)gridDim.y, sidx = (int)blockIdx.y * kSegWaves + wave;
    const double *a = mag + (long long)ch * mstride;
    double2 *p = out + (long long)ch * ostride;
    const bool lf = levelfade[ch] != 0;
    const int nb = (n + 63) >> 6, lb = L >> 6, nt = (n + L - 1) / L;
    const int t0 = (int)((long long)sidx * nt / S), t1 = (int)((long long)(sidx + 1) * nt / S);
    const int b0 = t0 * lb, b1 = t1 * lb < nb ? t1 * lb : nb;
    const bool is_last = t1 == nt && t0 < t1;
    const PoleScan sR = make_pole_scan(prm.mtauR, lane), sI = make_pole_scan(prm.mtauI, lane);
    double cR = state[ch].dc, cI = state[ch].dc_insert;
    if (lf) {
        // c = c_in mL^t0 + sum_{t < t0} mL^(t0 - 1 - t) s_t for both averages, 64 tiles per step
        const double mR = pow(prm.mtauR, (double)L), mI = pow(prm.mtauI, (double)L);
        const double *e = tsum + (long long)ch * tstride * 2;
        for (int blk = 0; blk < t0; blk += 64) {
            const int cnt = t0 - blk < 64 ? t0 - blk : 64;
            double vR = 0.0, vI = 0.0;
            if (lane < cnt) {
                const double2 s = *reinterpret_cast<const double2 *>(e + (long long)(blk + lane) * 2);
                vR = s.x * ipow_d(mR, cnt - 1 - lane);
                vI = s.y * ipow_d(mI, cnt - 1 - lane);
            }
            cR = __builtin_fma(cR, ipow_d(mR, cnt), wave_sum_d(vR));
            cI = __builtin_fma(cI, ipow_d(mI, cnt), wave_sum_d(vI));
        }
    }
    double an[kSegGroup];
    seg_load(an, b0, b1, n, lane, a);
    for (int b = b0; b < b1; b += kSegGroup) {
        double av[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) av[k] = an[k];
        seg_load(an, b + kSegGroup, b1, n, lane, a);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
            const double v = av[k];
            double audio = v;
            if (lf) {                                       // block-uniform
                const double dc = scan_pole_dpp(prm.onem_mtauR * v, sR) + sR.pw * cR;       // amd.c:136-137
                const double di = scan_pole_dpp(prm.onem_mtauI * v, sI) + sI.pw * cI;
                audio = v + (di - dc);                                                       // amd.c:138
                cR = lane_bcast(dc, cnt - 1); cI = lane_bcast(di, cnt - 1);
            }
            if (lane < cnt) p[base + lane] = make_double2(audio, audio);
        }
    }
    if (lf && is_last && lane == 0 && n > 0) { carry_out[ch].dc = cR; carry_out[ch].dc_insert = cI; }       // committed by commit_am_kernel
}

// ---- SAM with sideband selection (amd.c:150-208) over time segments ---------------------------------------------------------------
// Behind the time-tiled loop (pll_lanes_kernel<true>: the phase every sample saw) the four all-pass chains are linear and time
// invariant, with poles up to 0.9999 (no warm-up reaches that far).  Two passes over 16 G time segments per channel:
//   sam_sb_tiled_kernel<1>  every segment's chains from a ZERO state: the end state z_w, 4 chains x 17 words
//                           (ds, then x_j[n-1], x_j[n-2] for j = 0 .. 7: ap_chain64's state with the one-sample delay of chains a, c)
//   sam_sb_chain_kernel     start state of segment w: s_w = Phi(len_{w-1}) s_{w-1} + z_{w-1}, Phi = the chains' 17 x 17 transition
//                           over a segment of that length (two lengths occur; the host raises the one-sample matrix to the power)
//   sam_sb_tiled_kernel<2>  the chains again from the true start states; writes (audio, corr0): the fade leveller follows
//   sam_level_tiled_kernel  amd.c:211-216 with dc on audio and dc_insert on corr0 (two-pass segment scan as for AM)
constexpr int kSbW = 17, kSbSum = 4 * kSbW;         // words per chain, per segment summary
struct SbBatch { ApChain c[4]; double ds[2]; };
__device__ __forceinline__ void sb_zero(SbBatch &b)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) { b.c[k].x1[j] = 0.0; b.c[k].x2[j] = 0.0; }
    b.ds[0] = b.ds[1] = 0.0;
}
// word r of chain k: 0 = ds (chains 0 and 2: dsI, dsQ; unused in 1 and 3), 1 + 2 j = x_j[n-1], 2 + 2 j = x_j[n-2]
__device__ __forceinline__ void sb_store(const SbBatch &b, double *w)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[k * kSbW] = k == 0 ? b.ds[0] : k == 2 ? b.ds[1] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) { w[k * kSbW + 1 + 2 * j] = b.c[k].x1[j]; w[k * kSbW + 2 + 2 * j] = b.c[k].x2[j]; }
    }
}
__device__ __forceinline__ void sb_load(SbBatch &b, const double *w)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) { b.c[k].x1[j] = w[k * kSbW + 1 + 2 * j]; b.c[k].x2[j] = w[k * kSbW + 2 + 2 * j]; }
    b.ds[0] = w[0]; b.ds[1] = w[2 * kSbW];
}

// (workgroups of four wavefronts = four segments, 4 G workgroups per channel: the four chains' 64 state words and the 28 per-lane
// powers want ~220 registers, which a 16-wavefront workgroup does not have)
constexpr int kSbWaves = 4;
template <int MODE>
static __global__ __launch_bounds__(64 * kSbWaves, 2) void sam_sb_tiled_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                                               const SamChanParam *cprm, const double *pt, long long ptstride,
                                                                               PllState *state, double *sums, const double *starts)
{
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = kSbWaves * (int)gridDim.y, sidx = (int)blockIdx.y * kSbWaves + wave;
    double2 *p = buf + (long long)ch * stride;
    const double *ph = pt + (long long)ch * ptstride;
    const int sbmode = cprm[ch].sbmode;
    int b0, b1;
    seg_range(n, sidx, b0, b1, S);
    const double *c0 = kSamC0, *c1 = kSamC1;
    // (whole batches only: n is a multiple of 64 here; the samples dealt to the halves of the wavefront, see ap_chain64s)
    double pa0[7], pa1[7], pw0[7], pw1[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        pa0[j] = ipow_d(-c0[j], (lane & 15) + 1); pa1[j] = ipow_d(-c1[j], (lane & 15) + 1);
        pw0[j] = ipow_d(-c0[j], (lane & 31) + 1); pw1[j] = ipow_d(-c1[j], (lane & 31) + 1);
    }
    const int sl = 2 * (lane & 31) + (lane >> 5);           // this lane's sample inside a batch
    SbBatch B;
    if constexpr (MODE == 1) sb_zero(B);
    else sb_load(B, starts + ((long long)slot * S + sidx) * kSbSum);
    for (int b = b0; b < b1; b++) {
        const int i = b * 64 + sl;
        const double2 z = p[i];
        double sn, cs;
        sincos(ph[i] * kTwoPiRef, &sn, &cs);
        const double ai = z.x * cs, bi = z.x * sn, aq = z.y * cs, bq = z.y * sn;
        // one sample late: sample 2 k takes sample 2 k - 1 (lane 31 + k; the first: the carried one), sample 2 k + 1 takes sample 2 k (lane k)
        const int from = lane < 32 ? lane + 31 : lane - 32;
        double ai_d = __shfl(ai, from, 64), bq_d = __shfl(bq, from, 64);
        if (lane == 0) { ai_d = B.ds[0]; bq_d = B.ds[1]; }
        B.ds[0] = lane_bcast(ai, 63); B.ds[1] = lane_bcast(bq, 63);
        const double ai_ps = ap_chain64s(B.c[0], ai_d, c0, pa0, pw0, lane), bi_ps = ap_chain64s(B.c[1], bi, c1, pa1, pw1, lane);
        const double bq_ps = ap_chain64s(B.c[2], bq_d, c0, pa0, pw0, lane), aq_ps = ap_chain64s(B.c[3], aq, c1, pa1, pw1, lane);
        if constexpr (MODE == 2) {
            const double audio = sbmode == 1 ? (ai_ps - bi_ps) + (aq_ps + bq_ps) : (ai_ps + bi_ps) - (aq_ps - bq_ps);
            p[i] = make_double2(audio, ai + bq);
        }
    }
    if constexpr (MODE == 1) {
        if (lane == 0) sb_store(B, sums + ((long long)slot * S + sidx) * kSbSum);
    } else {
        int last = S - 1;
        while (last > 0 && seg_samples_of(n, last, S) == 0) last--;
        if (sidx == last && lane == 0 && n > 0) {
            PllState *sp = state + ch;
            ap_store(B.c[0], sp->a); ap_store(B.c[1], sp->b); ap_store(B.c[2], sp->c); ap_store(B.c[3], sp->d);
            sp->dsI = B.ds[0]; sp->dsQ = B.ds[1];
        }
    }
}

// phi: [2 lengths: q and q + 1 batches][2: chains a / c (coefficients c0, delayed input), chains b / d (c1)][17][17], row major
static __global__ __launch_bounds__(64) void sam_sb_chain_kernel(int n, int S, const int *chan_list, const PllState *state, const double *phi,
                                                                const double *sums, double *starts)
{
    __shared__ double sphi[2 * 2 * kSbW * kSbW];
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x;
    for (int i = lane; i < 2 * 2 * kSbW * kSbW; i += 64) sphi[i] = phi[i];
    __syncthreads();
    // lane r < 17 holds word r of the four chains' vectors
    const PllState *sp = state + ch;
    double v[4] = { 0.0, 0.0, 0.0, 0.0 };
    if (lane < kSbW) {
        const double *arr[4] = { sp->a, sp->b, sp->c, sp->d };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (lane == 0) v[k] = k == 0 ? sp->dsI : k == 2 ? sp->dsQ : 0.0;
            else { const int j = (lane - 1) >> 1; v[k] = arr[k][3 * j + 1 + ((lane - 1) & 1)]; }
        }
    }
    const int qb = ((n + 63) >> 6) / S;
    SegWalk walk(n, S);
    for (int w = 0; w < S; w++) {
        if (lane < kSbW) {
            double *o = starts + ((long long)slot * S + w) * kSbSum;
#pragma unroll
            for (int k = 0; k < 4; k++) o[k * kSbW + lane] = v[k];
        }
        const int nbw = walk.next();
        if (nbw == 0) continue;
        const double *P = sphi + (nbw == qb ? 0 : 1) * 2 * kSbW * kSbW;
        const double *z = sums + ((long long)slot * S + w) * kSbSum;
        double nv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) nv[k] = lane < kSbW ? z[k * kSbW + lane] : 0.0;
        for (int c = 0; c < kSbW; c++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double vk = lane_bcast(v[k], c);
                if (lane < kSbW) nv[k] = __builtin_fma(P[(k & 1) * kSbW * kSbW + lane * kSbW + c], vk, nv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = nv[k];
    }
}

// the fade leveller of the sideband modes: buf holds (audio, corr0); dc averages audio, dc_insert averages corr0 (amd.c:211-216)
template <int MODE>
static __global__ __launch_bounds__(kSegThreads) void sam_level_tiled_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                                             const int *levelfade, const AmState *state, AmParam prm, double *gsum,
                                                                             AmState *carry_out)
{
    const int slot = blockIdx.x, ch = chan_list[slot], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = kSegWaves * (int)gridDim.y, sidx = (int)blockIdx.y * kSegWaves + wave;
    double *sum = gsum + (long long)slot * S * kSegSumW;
    double2 *p = buf + (long long)ch * stride;
    const bool lf = levelfade[ch] != 0;
    int b0, b1;
    seg_range(n, sidx, b0, b1, S);
    const double m64R = lane_pow(prm.mtauR, 64), m64I = lane_pow(prm.mtauI, 64);
    if constexpr (MODE == 1) {
        if (!lf) return;
        double accR = 0.0, accI = 0.0;
        for (int b = b0; b < b1; b++) {
            const int i = b * 64 + lane;
            const double2 v = i < n ? p[i] : make_double2(0.0, 0.0);
            accR = __builtin_fma(accR, m64R, prm.onem_mtauR * v.x);
            accI = __builtin_fma(accI, m64I, prm.onem_mtauI * v.y);
        }
        const double eR = wave_sum_d(accR * lane_pow(prm.mtauR, 63 - lane)), eI = wave_sum_d(accI * lane_pow(prm.mtauI, 63 - lane));
        if (lane == 0) { sum[sidx * kSegSumW] = eR; sum[sidx * kSegSumW + 1] = eI; }
    } else {
        double cR = state[ch].dc, cI = state[ch].dc_insert;
        if (lf) {
            const int q = ((n + 63) >> 6) / S;
            const double tR0 = pow(m64R, (double)q), tI0 = pow(m64I, (double)q), tR1 = tR0 * m64R, tI1 = tI0 * m64I;
            SegWalk walk(n, S);
            for (int w = 0; w < sidx; w++) {
                const int nbw = walk.next();
                if (nbw == 0) continue;
                cR = __builtin_fma(cR, nbw == q ? tR0 : tR1, sum[w * kSegSumW]);
                cI = __builtin_fma(cI, nbw == q ? tI0 : tI1, sum[w * kSegSumW + 1]);
            }
        }
        const PoleScan sR = make_pole_scan(prm.mtauR, lane), sI = make_pole_scan(prm.mtauI, lane);
        for (int b = b0; b < b1; b++) {
            const int base = b * 64, cnt = n - base < 64 ? n - base : 64, i = base + lane;
            const double2 v = i < n ? p[i] : make_double2(0.0, 0.0);
            double audio = v.x;
            if (lf) {
                const double dc = scan_pole_dpp(prm.onem_mtauR * v.x, sR) + sR.pw * cR;
                const double di = scan_pole_dpp(prm.onem_mtauI * v.y, sI) + sI.pw * cI;
                audio = v.x + (di - dc);
                cR = lane_bcast(dc, cnt - 1); cI = lane_bcast(di, cnt - 1);
            }
            if (lane < cnt) p[i] = make_double2(audio, audio);
        }
        int last = S - 1;
        while (last > 0 && seg_samples_of(n, last, S) == 0) last--;
        if (lf && sidx == last && lane == 0 && n > 0) { carry_out[ch].dc = cR; carry_out[ch].dc_insert = cI; }   // committed by commit_am_kernel
    }
}

// ---- CTCSS notch (xsnotch, wdsp/iir.c:76-95): bi-quad on the I component, in place ------------------------------------
// state vector (y_i, y_{i-1}) driven by (f_i, 0), f_i = a0 x_i + a1 x_{i-1} + a2 x_{i-2}; transition A = [[b1, b2], [1, 0]]
__device__ __forceinline__ M2 m2_pow(M2 a, int e)
{
    M2 r; r.a = 1; r.b = 0; r.c = 0; r.d = 1;
    while (e > 0) {
        if (e & 1) r = mmul(a, r);
        a = mmul(a, a);
        e >>= 1;
    }
    return r;
}
struct BiquadScan { M2 a1, a2, a4, a8, pa, pb, pw; };
__device__ __forceinline__ BiquadScan make_biquad_scan(M2 A, int lane)
{
    BiquadScan s;
    s.a1 = A; s.a2 = mmul(A, A); s.a4 = mmul(s.a2, s.a2); s.a8 = mmul(s.a4, s.a4);
    s.pa = m2_pow(A, (lane & 15) + 1); s.pb = m2_pow(A, (lane & 31) + 1); s.pw = m2_pow(A, lane + 1);
    return s;
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ void biquad_step(double &u0, double &u1, const M2 &m)
{
    const double v0 = dpp_fetch_d<CTRL, ROWMASK>(u0), v1 = dpp_fetch_d<CTRL, ROWMASK>(u1);
    u0 += m.a * v0 + m.b * v1;
    u1 += m.c * v0 + m.d * v1;
}
__device__ __forceinline__ void scan_biquad_dpp(double &u0, double &u1, const BiquadScan &s)
{
    biquad_step<0x111, 0xf>(u0, u1, s.a1);
    biquad_step<0x112, 0xf>(u0, u1, s.a2);
    biquad_step<0x114, 0xf>(u0, u1, s.a4);
    biquad_step<0x118, 0xf>(u0, u1, s.a8);
    biquad_step<0x142, 0xa>(u0, u1, s.pa);
    biquad_step<0x143, 0xc>(u0, u1, s.pb);
}

template <int MODE = 0>
// dst != null: the stage is the channel's last -- pass 2 writes the output matrix applied to (y, Q) to dst instead of y in place (a
// channel whose notch does not run is copied through the matrix)
static __global__ __launch_bounds__(kSegThreads) void snotch_tiled_kernel(double2 *buf, long long stride, int n, const int *chan_list,
                                                                          const SnotchParam *prm, SnotchState *state, double *gsum = nullptr,
                                                                          double2 *dst = nullptr, long long dst_stride = 0,
                                                                          const EpiParam *epi = nullptr, SnotchState *carry_out = nullptr)
{
    // carry_out: as in am_detect_tiled_kernel -- the grid form (MODE 2) leaves the new state there for commit_snotch_kernel
    __shared__ double s_sum[kSegWaves * kSegSumW];
    const int ch = chan_list[blockIdx.x];
    const SnotchParam q = prm[ch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = MODE == 0 ? kSegWaves : kSegWaves * (int)gridDim.y, sidx = MODE == 0 ? wave : (int)blockIdx.y * kSegWaves + wave;
    EpiParam ep{ 1, 0, 0, 1 };
    if (dst && epi) ep = epi[ch];
    if (!q.run) {                                       // block-uniform
        if (dst && MODE != 1) {
            int c0, c1;
            seg_range(n, sidx, c0, c1, S);
            const double2 *src = buf + (long long)ch * stride;
            double2 *o = dst + (long long)ch * dst_stride;
            for (int i = c0 * 64 + lane; i < c1 * 64 && i < n; i += 64) {
                const double2 v = src[i];
                o[i] = make_double2(ep.a * v.x + ep.b * v.y, ep.c * v.x + ep.d * v.y);
            }
        }
        return;
    }
    double *sum = MODE == 0 ? s_sum : gsum + (long long)blockIdx.x * S * kSegSumW;     // row: e0, e1, x[end - 1], x[end - 2]
    double2 *p = buf + (long long)ch * stride;
    const SnotchState st0 = state[ch];
    int b0, b1;
    seg_range(n, sidx, b0, b1, S);
    M2 A; A.a = q.b1; A.b = q.b2; A.c = 1.0; A.d = 0.0;
    const BiquadScan sc = make_biquad_scan(A, lane);
    // the forcing term needs x two samples back: from the buffer (pass 1 leaves it untouched), from the carried state at
    // the very beginning; pass 2 overwrites x with y, so every segment's last two inputs are put aside for its successor
    auto x_at = [&](int i) -> double { return i >= 0 ? p[i].x : (i == -1 ? st0.x1 : st0.x2); };
    const M2 A64 = m2_pow(A, 64);
    double xm1, xm2;
    double xn[kSegGroup];
    if constexpr (MODE != 2) {
        // pass 1: response of the zero state to the segment's forcing.  End value only: lane l keeps
        // sum_b (A^64)^(B - 1 - b) (u[64 b + l], 0), and sum_l A^(63 - l) acc_l closes the segment.
        double acc0 = 0.0, acc1 = 0.0;
        xm1 = x_at(b0 * 64 - 1); xm2 = x_at(b0 * 64 - 2);
        seg_load_re(xn, b0, b1, n, lane, p);
        for (int b = b0; b < b1; b += kSegGroup) {
            double xv[kSegGroup];
#pragma unroll
            for (int k = 0; k < kSegGroup; k++) xv[k] = xn[k];
            seg_load_re(xn, b + kSegGroup, b1, n, lane, p);
#pragma unroll
            for (int k = 0; k < kSegGroup; k++) {
                if (b + k >= b1) break;
                const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
                const double x0 = xv[k];
                double x1 = dpp_fetch_d<0x138, 0xf>(x0), x2 = dpp_fetch_d<0x138, 0xf>(x1);    // wave_shr:1, twice
                if (lane == 0) { x1 = xm1; x2 = xm2; }
                if (lane == 1) x2 = xm1;
                const double u = lane < cnt ? q.a0 * x0 + q.a1 * x1 + q.a2 * x2 : 0.0;
                const double t0 = A64.a * acc0 + A64.b * acc1 + u, t1 = A64.c * acc0 + A64.d * acc1;
                acc0 = t0; acc1 = t1;
                const double prev1 = xm1;
                xm1 = lane_bcast(x0, cnt - 1);
                xm2 = cnt >= 2 ? lane_bcast(x0, cnt - 2) : prev1;
            }
        }
        const M2 W = m2_pow(A, 63 - lane);
        const double e0 = wave_sum_d(W.a * acc0 + W.b * acc1), e1 = wave_sum_d(W.c * acc0 + W.d * acc1);
        const int ns = seg_samples(n, b0, b1);
        if (lane == 0) {
            const int end = b0 * 64 + ns;                   // one past the segment's last sample
            double *row = sum + sidx * kSegSumW;
            row[0] = e0; row[1] = e1;
            row[2] = ns > 0 ? x_at(end - 1) : 0.0; row[3] = ns > 0 ? x_at(end - 2) : 0.0;
        }
        if constexpr (MODE == 1) return;
    }
    if constexpr (MODE == 0) __syncthreads();
    // true (y_{-1}, y_{-2}) and (x_{-1}, x_{-2}) at the start of this segment
    double c0 = st0.y1, c1 = st0.y2;
    xm1 = st0.x1; xm2 = st0.x2;
    {
        const int qb = ((n + 63) >> 6) / S;
        const M2 T0 = m2_pow(A64, qb), T1 = mmul(A64, T0);
        SegWalk walk(n, S);
        for (int w = 0; w < sidx; w++) {
            const int nbw = walk.next();
            if (nbw == 0) continue;
            const M2 T = nbw == qb ? T0 : T1;
            const double *row = sum + w * kSegSumW;
            const double n0 = T.a * c0 + T.b * c1 + row[0], n1 = T.c * c0 + T.d * c1 + row[1];
            c0 = n0; c1 = n1;
            xm1 = row[2]; xm2 = row[3];
        }
    }
    // pass 2 (every segment's boundary inputs were put aside before the barrier above: in-place writes are safe now)
    seg_load_re(xn, b0, b1, n, lane, p);
    for (int b = b0; b < b1; b += kSegGroup) {
        double xv[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) xv[k] = xn[k];
        seg_load_re(xn, b + kSegGroup, b1, n, lane, p);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
            const double x0 = xv[k];
            double x1 = dpp_fetch_d<0x138, 0xf>(x0), x2 = dpp_fetch_d<0x138, 0xf>(x1);    // wave_shr:1, twice
            if (lane == 0) { x1 = xm1; x2 = xm2; }
            if (lane == 1) x2 = xm1;
            double u0 = lane < cnt ? q.a0 * x0 + q.a1 * x1 + q.a2 * x2 : 0.0, u1 = 0.0;
            scan_biquad_dpp(u0, u1, sc);
            const double y0 = u0 + sc.pw.a * c0 + sc.pw.b * c1, y1 = u1 + sc.pw.c * c0 + sc.pw.d * c1;
            if (lane < cnt) {                                       // the Q component passes (iir.c:76-95 filters I only)
                if (dst) {
                    const double qv = p[base + lane].y;
                    dst[(long long)ch * dst_stride + base + lane] = make_double2(ep.a * y0 + ep.b * qv, ep.c * y0 + ep.d * qv);
                } else p[base + lane].x = y0;
            }
            c0 = lane_bcast(y0, cnt - 1); c1 = lane_bcast(y1, cnt - 1);
            const double prev1 = xm1;
            xm1 = lane_bcast(x0, cnt - 1);
            xm2 = cnt >= 2 ? lane_bcast(x0, cnt - 2) : prev1;
        }
    }
    int last = S - 1;
    while (last > 0 && seg_samples_of(n, last, S) == 0) last--;
    if (sidx == last && lane == 0 && n > 0) { SnotchState st; st.x1 = xm1; st.x2 = xm2; st.y1 = c0; st.y2 = c1; (carry_out ? carry_out : state)[ch] = st; }
}

static __global__ void commit_snotch_kernel(SnotchState *state, const SnotchState *next, const int *chan_list, int count, const SnotchParam *prm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int ch = chan_list[i];
    if (prm[ch].run) state[ch] = next[ch];
}

// ---- FM discriminator (xfmd's loop, wdsp/fmd.c:151-172) ---------------------------------------------------------------
// arg z in turns, (-0.5, 0.5], for every sample of the listed channels; an all-zero sample ("corr[0] = 1.0": det = 0) is
// marked by the value kThetaZero, which the loop turns into zero loop gains for that step
static constexpr double kThetaZero = kThetaZeroMark;     // qh_osfir.hpp: a THETA stage writes the same mark
static __global__ __launch_bounds__(NT) void pll_theta_kernel(const double2 *buf, long long stride, int n, const int *chan_list,
                                                              double *theta, long long tstride)
{
    const int ch = chan_list[blockIdx.y];
    const double2 *p = buf + (long long)ch * stride;
    double *th = theta + (long long)ch * tstride;
    for (long long g = (long long)blockIdx.x * NT + threadIdx.x; g < n; g += (long long)gridDim.x * NT) {
        const double2 z = p[g];
        double t = atan2(z.y, z.x) * (1.0 / kTwoPiRef);
        if (z.x == 0.0 && z.y == 0.0) t = kThetaZero;
        th[g] = t;
    }
}

__device__ __forceinline__ double min_nn(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_nn_d(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One wavefront = 64 consecutive tiles of one channel, one tile per lane.  theta: [ch][tstride] angles (turns) of the call's
// n samples; fil: [ch][fstride] the loop filter output per sample (what xfmd's dc removal and gain work on).  L = tile length
// (multiple of 64), H = warm-up (multiple of 64).  Lane l of group g owns outputs [q L, (q + 1) L), q = 64 g + l, and starts
// at sample max(0, q L - H): from the carried state when that is sample 0 -- exact -- else from the zero state.
// kPllBatch steps at a time: the 64 x kPllBatch angles of the next batch travel from HBM into registers (runs of kPllBatch
// consecutive samples per tile, coalesced) while the current batch is stepped (lane l takes row l out of LDS into registers first;
// odd pitch: conflict free both ways); the loop filter outputs go back through LDS and leave the same way.  32 steps per batch
// instead of 64 halves the registers (256 -> ~130) and the LDS block (33 -> 17 KB): the loop is a chain of dependent fp64
// operations, and only other wavefronts on the SIMD fill its latency.
// Speculation is checked, not trusted: every tile records its loop state where the warm-up ends (`ends` [ch][tile][0..2]) and
// where the tile ends ([3..5]); pll_verify_kernel compares neighbours and re-runs, in order, the tiles whose warm-up had
// not met the true trajectory yet (a loop that sits on noise, without a carrier, can take several hundred samples).
static constexpr int kPllBatch = 32;
static constexpr int kPllPitch = kPllBatch + 1;
// `ends` row of a tile: [0..2] loop state where the warm-up ended, [3..5] where the tile ended, [6] (FM) the tile's own
// contribution to the dc-removal average at its end: onem_mtau sum_i mtau^(L-1-i) fil_i -- what fm_dc_tiled_kernel's first pass
// would read `fil` again for.  The loop kernel has every fil_i in a register anyway.
static constexpr int kPllEndsW = 8;
struct PllLane { double pt, fil_out, omega; };

// EMIT_PT: the step's output is the VCO phase the sample SAW (turns; what the SAM detector mixes with, amd.c:150-158) instead
// of the loop filter's output after it (the FM discriminator's audio, fmd.c:165)
template <bool CHECKED, bool EMIT_PT>
__device__ __forceinline__ void pll_lane_steps(PllLane &s, double (&t)[kPllBatch], long long gb, int n, double g1t, double g2t, double inv,
                                               double lo, double hi)
{
#pragma unroll
    for (int k = 0; k < kPllBatch; k++) {
        const double th = t[k];
        const double seen = s.pt;
        double d = th - s.pt;                                   // (-1.5, 0.5] turns
        d -= rint(d);
        d = th < 4.0 ? d : 0.0;                                 // all-zero sample: det = 0 (fmd.c:156-157)
        const double del_out = s.fil_out;
        const double om = min_nn(max_nn_d(__builtin_fma(g2t, d, s.omega), lo), hi);     // fmd.c:159-161
        const double fo = __builtin_fma(g1t, d, om);
        const double np = __builtin_amdgcn_fract(__builtin_fma(del_out, inv, s.pt));
        if (CHECKED) {
            const long long g = gb + k;
            if (g >= 0 && g < n) { s.omega = om; s.fil_out = fo; s.pt = np; }        // outside the call the state stands still
        } else {
            s.omega = om; s.fil_out = fo; s.pt = np;
        }
        t[k] = EMIT_PT ? seen : s.fil_out;
    }
}

template <bool EMIT_PT>
static __global__ __launch_bounds__(64, 2) void pll_lanes_kernel(const double *theta, long long tstride, double *fil, long long fstride, int n,
                                                                 const int *chan_list, const PllState *state, double *ends, long long estride,
                                                                 PllParam q, int L, int H, int local_dc = 0)
{
    __shared__ double lds[64 * kPllPitch];
    const int ch = chan_list[blockIdx.y], lane = threadIdx.x, group = blockIdx.x;
    const long long tile0 = (long long)group * 64 * L;                  // first output sample of lane 0's tile
    if (tile0 >= n) return;                                             // no tile of this wavefront starts inside the call
    const double *th = theta + (long long)ch * tstride;
    double *fo = fil + (long long)ch * fstride;
    const long long s0 = tile0 + (long long)lane * L;                   // first output sample of this lane's tile
    const long long g_first = s0 - H;                                   // sample of step 0 (may be negative: those steps idle)
    const bool live = s0 < n;
    // loop state: carried for a lane whose run begins at (or before) sample 0, zero otherwise
    const PllState *sp = state + ch;
    const bool from_carry = g_first <= 0;
    PllLane s{ from_carry ? sp->phs * (1.0 / kTwoPiRef) : 0.0, from_carry ? sp->fil_out : 0.0, from_carry ? sp->omega : 0.0 };
    const double g1t = q.g1 * kTwoPiRef, g2t = q.g2 * kTwoPiRef, inv = 1.0 / kTwoPiRef, lo = q.omega_min, hi = q.omega_max;
    if (!from_carry && live) {
        // a warm-up has to end on the trajectory the loop is really on, and a loop with a wrapping detector can hold several
        // (false locks): start where a loop that has been tracking would be -- on the signal's own phase ...
        const double t0 = th[g_first], t1 = th[g_first + 1];
        if (t0 < 4.0 && t1 < 4.0) {
            s.pt = t0 - floor(t0);
            if constexpr (EMIT_PT) {
                // ... and, for SAM's narrow loop (40 Hz: it cannot pull a frequency guess in within a warm-up), at the frequency
                // the loop held when the call began: a carrier it has locked to stays put over a call.  While it is still
                // acquiring, the tiles fail the check and the verify pass steps the call in order, like the reference.
                s.omega = sp->omega;
            } else {
                // ... and phase step (the FM loop follows a single step's estimate within a few samples)
                double dt = t1 - t0;
                dt -= rint(dt);
                s.omega = min_nn(max_nn_d(dt * kTwoPiRef, lo), hi);
            }
            s.fil_out = s.omega;
        }
    }
    const int nsteps = H + L;
    double *e = ends + (long long)ch * estride + ((long long)group * 64 + lane) * kPllEndsW;
    double dcsum = 0.0;
    constexpr int B = kPllBatch, RPI = 64 / B;                          // rows (tiles) one load instruction of the wavefront covers
    const int frow = lane / B, fcol = lane % B;
    // addresses: a wave-uniform base (scalar registers) that moves with the batch, and one 32-bit lane offset plus a uniform
    // multiple of the row stride per load, formed when the load is issued -- kept out of the loop-invariant code the compiler would
    // otherwise park in 2 x 2 x kPllBatch registers (the asm statement below)
    const int RL = RPI * L, loff0 = frow * L + fcol, gb0 = (int)tile0 - H;
    double tn[B];                                                       // the next batch's angles: instruction j = rows RPI j .. RPI j + RPI - 1
    auto fetch = [&](int i0) {
        const double *pb = th + ((long long)gb0 + i0);                  // dereferenced only where the sample index is inside the call
        int loff = loff0;
        asm volatile("" : "+v"(loff));
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int off = loff + j * RL, g = gb0 + i0 + off;
            tn[j] = (g >= 0 && g < n) ? pb[(unsigned)off] : 0.0;
        }
    };
    fetch(0);
    for (int i0 = 0; i0 < nsteps; i0 += B) {
#pragma unroll
        for (int j = 0; j < B; j++) lds[(RPI * j + frow) * kPllPitch + fcol] = tn[j];
        if (i0 + B < nsteps) fetch(i0 + B);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double *row = lds + lane * kPllPitch;
        double t[B];
#pragma unroll
        for (int k = 0; k < B; k++) t[k] = row[k];
        const long long gb = g_first + i0;
        if (live) {
            if (gb >= 0 && gb + B <= n) pll_lane_steps<false, EMIT_PT>(s, t, gb, n, g1t, g2t, inv, lo, hi);
            else if (gb + B - 1 >= 0) pll_lane_steps<true, EMIT_PT>(s, t, gb, n, g1t, g2t, inv, lo, hi);
        }
        if (i0 + B == H && live) { e[0] = s.pt; e[1] = s.fil_out; e[2] = s.omega; }       // state where the warm-up ends
        if constexpr (!EMIT_PT) {
            if (i0 >= H) {                              // output steps (a tile that crosses the call's end is nobody's predecessor)
                // local_dc (FM, fm_dc_chain_kernel's form): what leaves is fil - (the dc average's response to this tile's OWN samples from a
                // zero state) -- onem_mtau times the running sum; the share of everything ahead of the tile, c_in mtau^(k + 1), is taken off
                // where the next stage loads the sample (osfir_kernel, PAIR with fmdc_a)
                if (local_dc) {
                    // (here the sum stops at the call's last sample: the tile the call's end cuts short hands its share to the chain too)
                    if (gb + B <= n) {
#pragma unroll
                        for (int k = 0; k < B; k++) { dcsum = __builtin_fma(dcsum, q.mtau, t[k]); t[k] = __builtin_fma(-q.onem_mtau, dcsum, t[k]); }
                    } else {
#pragma unroll
                        for (int k = 0; k < B; k++)
                            if (gb + k < n) { dcsum = __builtin_fma(dcsum, q.mtau, t[k]); t[k] = __builtin_fma(-q.onem_mtau, dcsum, t[k]); }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < B; k++) dcsum = __builtin_fma(dcsum, q.mtau, t[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < B; k++) row[k] = t[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // write the output steps out (warm-up steps i < H are dropped)
        if (i0 >= H) {
            double *ob = fo + ((long long)gb0 + i0);
            int loff = loff0;
            asm volatile("" : "+v"(loff));
#pragma unroll
            for (int j = 0; j < B; j++) {
                const int off = loff + j * RL;
                if (gb0 + i0 + off < n) ob[(unsigned)off] = lds[(RPI * j + frow) * kPllPitch + fcol];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live) {                                                           // state at the end of the tile (or of the call)
        e[3] = s.pt; e[4] = s.fil_out; e[5] = s.omega;
        if constexpr (!EMIT_PT) e[6] = q.onem_mtau * dcsum;
    }
}

// Checks the speculation of pll_lanes_kernel and repairs it.  Tile t is right when the state its warm-up reached equals
// the state tile t - 1 ended in and tile t - 1 is right (the loop is deterministic: equal states stay equal; tiles 0 .. H/L ran
// from the carried state and are right by construction).  One wavefront per channel walks the tiles in order; a tile that
// fails is re-run from its predecessor's end state the sequential way (64 samples per batch, the angles in the lanes, the
// loop stepped uniformly: pll_run64), which also gives the end state its successor is checked against.  A carrier the loop
// tracks costs nothing here; a loop that history has left in one of its false locks (the wrapping detector and the +-8 kHz
// clamp allow alias locks with the frequency pinned at the clamp, e.g. on an unmodulated carrier 1 kHz off tune after a
// noisy start) fails every check and the whole call is stepped sequentially, as the reference does.
// Then the loop state after the call's last sample goes into `state`.
__device__ __forceinline__ bool pll_state_differs(double pa, double fa, double oa, double pb, double fb, double ob)
{
    double dp = pa - pb;
    dp -= rint(dp);
    const double tol = 1e-12;
    return !(fabs(dp) < tol && fabs(fa - fb) < tol && fabs(oa - ob) < tol);
}

template <bool EMIT_PT>
static __global__ __launch_bounds__(64) void pll_verify_kernel(const double *theta, long long tstride, double *fil, long long fstride, int n,
                                                                  const int *chan_list, PllState *state, double *ends, long long estride,
                                                                  PllParam q, int L, int H, int *nfixed, int check_only, int local_dc = 0)
{
    __shared__ double pll_out[128];
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    const double *th = theta + (long long)ch * tstride;
    double *fo = fil + (long long)ch * fstride;
    double *e = ends + (long long)ch * estride;
    const int ntiles = (n + L - 1) / L;
    const int first = H / L + 1;                                        // tiles 0 .. H/L began at sample 0 from the carried state
    int fixed = 0;
    const PoleScan sc = make_pole_scan(q.mtau, lane);
    // the tile re-run last and its end state (uniform): whoever needs that state takes it from here, not from memory
    double rp = 0, rf = 0, ro = 0;
    int rt = -1;
    for (int base = first; base < ntiles; base += 64) {
        const int t = base + lane;
        bool bad = false;
        double wp = 0, wf = 0, wo = 0;
        if (t < ntiles) {
            const double *a = e + (long long)t * kPllEndsW, *b = e + (long long)(t - 1) * kPllEndsW + 3;
            wp = a[0]; wf = a[1]; wo = a[2];
            double bp = b[0], bf = b[1], bo = b[2];
            if (t - 1 == rt) { bp = rp; bf = rf; bo = ro; }             // the last tile of the batch before was re-run
            bad = pll_state_differs(wp, wf, wo, bp, bf, bo);
        }
        unsigned long long mask = __ballot(bad);
        if (check_only) { fixed += __popcll(mask); mask = 0; }         // diagnostics: count, leave the speculative result
        while (mask) {
            const int l0 = __ffsll((long long)mask) - 1, tq = base + l0;
            mask &= ~(1ull << l0);
            // re-run tile tq from the end state of tile tq - 1 (uniform values: read by every lane)
            const double *bq = e + (long long)(tq - 1) * kPllEndsW + 3;
            PllLoop Ls{ bq[0], bq[1], bq[2] };
            if (tq - 1 == rt) { Ls.pt = rp; Ls.fil_out = rf; Ls.omega = ro; }
            const long long s0 = (long long)tq * L;
            const int len = (int)((long long)n - s0 < L ? (long long)n - s0 : L);
            double dcsum = 0.0, dcl = 0.0;
            for (int off = 0; off < len; off += 64) {
                const int cnt = len - off < 64 ? len - off : 64;
                double tt = 0.0;
                if (lane < cnt) tt = th[s0 + off + lane];
                const unsigned long long zero = __ballot(tt >= 4.0);
                double my_pt, my_fil;
                pll_run64(Ls, tt, zero, cnt, q, lane, my_pt, my_fil, pll_out);
                double outv = EMIT_PT ? my_pt : my_fil;
                if constexpr (!EMIT_PT) {
                    if (local_dc) {         // (pll_lanes_kernel's local_dc: the tile's own share of the dc average comes off here)
                        const double dcs = scan_pole_dpp(lane < cnt ? q.onem_mtau * my_fil : 0.0, sc) + sc.pw * dcl;
                        outv = my_fil - dcs;
                        dcl = lane_bcast(dcs, cnt - 1);
                    }
                }
                if (lane < cnt) fo[s0 + off + lane] = outv;
                if constexpr (!EMIT_PT)
                    dcsum = __builtin_fma(dcsum, lane_pow(q.mtau, cnt), wave_sum_d(lane < cnt ? my_fil * lane_pow(q.mtau, cnt - 1 - lane) : 0.0));
            }
            if (lane == 0) {
                double *w = e + (long long)tq * kPllEndsW + 3;
                w[0] = Ls.pt; w[1] = Ls.fil_out; w[2] = Ls.omega;
                if constexpr (!EMIT_PT) w[3] = q.onem_mtau * dcsum;
            }
            rt = tq; rp = Ls.pt; rf = Ls.fil_out; ro = Ls.omega;
            fixed++;
            // the successor is judged against the repaired end state
            const double np_ = lane_bcast(wp, l0 + 1 < 64 ? l0 + 1 : 63), nf_ = lane_bcast(wf, l0 + 1 < 64 ? l0 + 1 : 63),
                         no_ = lane_bcast(wo, l0 + 1 < 64 ? l0 + 1 : 63);
            if (l0 + 1 < 64 && tq + 1 < ntiles) {
                if (pll_state_differs(np_, nf_, no_, Ls.pt, Ls.fil_out, Ls.omega)) mask |= 1ull << (l0 + 1);
                else mask &= ~(1ull << (l0 + 1));
            }
        }
    }
    if (lane == 0 && n > 0) {
        const double *b = e + (long long)(ntiles - 1) * kPllEndsW + 3;
        double b0 = b[0], b1 = b[1], b2 = b[2];
        if (rt == ntiles - 1) { b0 = rp; b1 = rf; b2 = ro; }
        state[ch].phs = b0 * kTwoPiRef; state[ch].fil_out = b1; state[ch].omega = b2;
        if (fixed && nfixed) atomicAdd(nfixed, fixed);
    }
}

// dc removal and gain of xfmd (fmd.c:169-171): fmdc <- mtau fmdc + onem_mtau fil ; audio = again (fil - fmdc), written as
// (audio, audio); fil is a real array, out the channel's complex row.  ONE pass: the loop kernels left every tile's contribution to
// the average in `ends` (kPllEndsW per tile, [6]); the time segments (16 per workgroup, gridDim.y workgroups per channel) are cut
// on tile boundaries (L samples, a multiple of 64) and a segment's carry-in is the chain over the tiles ahead of it.
static __global__ __launch_bounds__(kSegThreads) void fm_dc_tiled_kernel(const double *fil, long long fstride, double2 *out, long long stride,
                                                                         int n, const int *chan_list, const PllState *state, const double *again,
                                                                         PllParam q, const double *ends, long long estride, int L, double *fmdc_out)
{
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = kSegWaves * (int)gridDim.y, sidx = (int)blockIdx.y * kSegWaves + wave;
    const double *f = fil + (long long)ch * fstride;
    double2 *p = out + (long long)ch * stride;
    const PoleScan sc = make_pole_scan(q.mtau, lane);
    const int nb = (n + 63) >> 6, lb = L >> 6, nt = (n + L - 1) / L;
    const int t0 = (int)((long long)sidx * nt / S), t1 = (int)((long long)(sidx + 1) * nt / S);      // this segment's tiles
    const int b0 = t0 * lb, b1 = t1 * lb < nb ? t1 * lb : nb;
    const bool is_last = t1 == nt && t0 < t1;               // the segment that holds the call's last sample
    double c = state[ch].fmdc;
    {
        // c = c_in mL^t0 + sum_{t < t0} mL^(t0 - 1 - t) s_t, 64 tiles per step; tiles whose weight is below 1e-40 are left out
        const double mL = pow(q.mtau, (double)L), mL64 = lane_pow(mL, 64);
        const double *e = ends + (long long)ch * estride;
        int depth = 1;
        for (double w = mL64; w > 1e-40 && depth < 4096; w *= mL64) depth++;
        int first = t0 - 64 * depth;
        if (first <= 0) first = 0; else c = 0.0;
        for (int blk = first; blk < t0; blk += 64) {
            const int cnt = t0 - blk < 64 ? t0 - blk : 64;
            double v = 0.0;
            if (lane < cnt) v = e[(long long)(blk + lane) * kPllEndsW + 6] * lane_pow(mL, cnt - 1 - lane);
            c = __builtin_fma(c, lane_pow(mL, cnt), wave_sum_d(v));
        }
    }
    const double gain = again[ch];
    double fn[kSegGroup];
    seg_load(fn, b0, b1, n, lane, f);
    for (int b = b0; b < b1; b += kSegGroup) {
        double fv[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) fv[k] = fn[k];
        seg_load(fn, b + kSegGroup, b1, n, lane, f);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
            const double dcs = scan_pole_dpp(q.onem_mtau * fv[k], sc) + sc.pw * c;
            const double audio = gain * (fv[k] - dcs);
            if (lane < cnt) p[base + lane] = make_double2(audio, audio);
            c = lane_bcast(dcs, cnt - 1);
        }
    }
    // a slot of its own: another workgroup of this launch may not have read state[ch].fmdc yet (commit_fmdc_kernel moves it in)
    if (is_last && lane == 0 && n > 0) fmdc_out[ch] = c;
}

// The dc average of xfmd at every tile boundary (fmd.c:169), for the loop kernels' local_dc form: cin[ch][t] = fmdc just ahead of tile
// t's first sample = the carried value through every tile before it, c_(t+1) = mtau^L c_t + s_t with s_t the tile's own contribution
// (`ends` [6]).  One wavefront per channel, 64 tiles per step (a scan with the pole mtau^L); the value behind the call's last sample goes
// straight into the state (nobody else reads it in this call: the stage that takes the dc off reads cin).
static __global__ __launch_bounds__(64) void fm_dc_chain_kernel(int n, int L, const int *chan_list, PllState *state, PllParam q, const double *ends,
                                                                 long long estride, double *cin, long long cstride)
{
    const int ch = chan_list[blockIdx.x], lane = threadIdx.x;
    const int nt = (n + L - 1) / L, nfull = n / L;                     // tiles, whole tiles
    const double mL = pow(q.mtau, (double)L);
    const double *e = ends + (long long)ch * estride;
    double *co = cin + (long long)ch * cstride;
    double c = state[ch].fmdc;
    if (lane == 0) co[0] = c;
    // fmdc behind the last sample of every whole tile
    c = chain_tiles<8>(nfull, mL, c, lane, [&](int t) { return e[(long long)t * kPllEndsW + 6]; }, [&](int t, double v) { co[t] = v; });
    if (nt > nfull) c = __builtin_fma(c, pow(q.mtau, (double)(n - nfull * L)), e[(long long)nfull * kPllEndsW + 6]);        // the tile the call's end cuts short
    if (lane == 0 && n > 0) state[ch].fmdc = c;
}

// The delay line of the stage that loads the audio in that form (fm_audio_at, qh_osfir.hpp) (hist_update_kernel's job for a stage whose input never exists as complex
// samples): new_hist[j] <- (audio, audio) of sample n - H + j, or the old line's sample where that index is negative.
static __global__ __launch_bounds__(256) void fm_audio_hist_kernel(const double *a_local, long long astride, int n, const int *chan_list,
                                                                   const double *cin, long long cstride, const double *pw, int shift, const double *again,
                                                                   const double2 *old_hist, double2 *new_hist, int H)
{
    const int ch = chan_list[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    const long long g = (long long)n - H + j;
    double2 v;
    if (g >= 0) {
        const double a = fm_audio_at(a_local + (long long)ch * astride, cin + (long long)ch * cstride, pw, shift, again[ch], g);
        v = make_double2(a, a);
    } else v = old_hist[(long long)ch * H + (H + g)];
    new_hist[(long long)ch * H + j] = v;
}

static __global__ void commit_fmdc_kernel(PllState *state, const double *next, const int *chan_list, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) state[chan_list[i]].fmdc = next[chan_list[i]];
}

}  // namespace qh
