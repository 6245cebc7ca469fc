// qh_engine.hip -- batched RXA receive engine for MI355X and its C ABI (include/quiskhip.h, group 1).
//
// Mirrors create_rxa()/xrxa() of the reference (wdsp/RXA.c:31-598) for the blocks on the hot path.
// Per-channel differences are data (NCO step, masks, 2x2 output matrix), not control flow, so one
// launch per stage covers every channel:
//
//   front   : xshift + xresample(in)      -> qh::osfir_kernel<NFFT, D = in_rate/dsp_rate, MIX>
//   nbp0    : xnbp (fircore)              -> qh::osfir_kernel<NFFT, 1>
//   bp1     : xbandpass (fircore)         -> qh::osfir_kernel<NFFT, 1>      (only when some channel runs it)
//   epilogue: xwcpagc mode 0 + xpanel     -> fused into the last launch (2x2 real matrix per channel)
//
// State carried between calls (all device resident, right-aligned rows of the most recent samples):
//   hist_front [2][nch][HF]  raw input samples (the reference's resampler ring holds them behind xshift; here the
//                            oscillator sits behind the filter, qh_osfir.hpp OUTMIX, and nco_retune_hist_kernel
//                            rewrites the row when a channel's shift changes)
//   hist_nbp   [2][nch][HB]  nbp0 input samples  (the reference's fircore delay line)
//   hist_bp1   [2][nch][HB]
//   nco_phase  [nch]         64-bit fixed-point turns
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/quiskhip.h"
#include "qh_design.hpp"
#include "qh_kernels.hpp"
#include "qh_demod.hpp"
#include "qh_tiled.hpp"
#include "qh_agc_tiled.hpp"
#include "qh_emnr.hpp"
#include "qh_snba.hpp"
#include "qh_internal.hpp"

namespace qh {

thread_local std::string g_last_error;

int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static constexpr int kNfft = 4096;          // FFT size of the front stage and, for nc <= 2048, of the fircore stages
static constexpr int kBandNfftMax = 8192;   // fircore stages with 2048 < nc <= 4096 run 8192-point tiles (Engine::bnfft)
static constexpr int kHistBand = 4095;      // fircore history capacity: nc up to 4096
// nc = 8192 ... 65536 (RXASetNC, wdsp/RXA.c:934-946): the impulse response in partitions of 4096 taps, every partition an ordinary
// 8192-point tile pass over a view of the stream that starts 4096 p samples earlier, the passes added (Engine::run_band)
static constexpr int kLongPart = 4096, kLongNcMax = 65536, kLongParts = kLongNcMax / kLongPart, kLongHist = kLongNcMax - 1;
static constexpr int kHistFront = 2240;     // resampler history capacity: 140 * D taps, D <= 16
// FM PLL time tiles (qh_tiled.hpp).  On a carrier the loop (double pole at 0.66 per sample) forgets its start state in ~100
// samples; on noise alone two runs meet after ~135 samples on average with an exponential tail, so a 768-sample warm-up
// leaves a fraction of a percent of the tiles to the verify kernel's sequential re-run.
static constexpr int kFmTile = 256;        // shortest tile; long calls take up to 2048 samples per lane (Engine::process_chain)
static constexpr int kFmWarm = 768;
// SAM's loop (omega_N 250 rad/s, zeta 1, RXA.c:185-186) forgets a state in exp(-250 t): 1e-16 after 0.147 s = 7068 samples at 48 kHz
static constexpr int kSamTile = 4096;
static constexpr int kSamWarm = 8192;
static constexpr long long kSamTiledMin = 4 * 8192;     // shorter calls take the sequential kernel
static constexpr int kAgcSegs = 16;                     // super-segments of the AGC boundary pass, at most
static constexpr long long kAgcTiledMin = 16384;        // xwcpagc in time tiles from this many detector samples per call (qh_agc_tiled.hpp)

struct ChanCfg {
    int mode = QH_LSB;                                          // RXA.c:33
    int shift_run = 1; double shift_freq = 0.0;                 // RXA.c:39-45
    int shift_on_device = 1;                                    // whether the device phase is live or parked (nco_park_kernel)
    int nbp_run = 1, nbp_nc = 2048, nbp_wintype = 0;            // RXA.c:90-106
    int mp = 0;                                                 // RXASetMP, RXA.c:948
    // notch database (create_notchdb RXA.c:85-87) and nbp0's use of it (fnfrun 0, autoincr 1: RXA.c:92,104)
    std::vector<Notch> notches;
    double ndb_tunefreq = 0.0, ndb_shift = 0.0;
    int fnfrun = 0, autoincr = 1;
    double nbp_flow = -4150.0, nbp_fhigh = -150.0, nbp_gain = 1.0;
    int amd_run = 0, amd_mode = 0, fmd_run = 0;                 // RXA.c:175-212
    int agc_run = 1, agc_mode = 3; double agc_fixed = 1000.0;   // RXA.c:335-358
    double agc_tau_attack = 0.001, agc_tau_decay = 0.250, agc_max_gain = 10000.0, agc_var_gain = 1.5;
    double agc_hangtime = 0.250, agc_hang_thresh = 0.250;
    bool agc_dirty = true;
    bool agc_on() const { return agc_run && agc_mode != 0; }
    int agc_abuf = -1;                  // attack_buffsize last uploaded
    bool agc_rewindow = false;          // the attack window moved in mid-stream: the state's ring is taken again from the full one
    bool agc_ran = false, agc_stale = false;    // the window moved while the ring held samples: ring_max may be stale (qh_agc_tiled.hpp)
    int bp1_run = 1, bp1_nc = 2048, bp1_wintype = 1;            // RXA.c:377-389
    bool long_live[5] = { false, false, false, false, false };  // the channel holds a delay line of stage sid longer than 4095 samples (process_chain)
    double bp1_flow = -4150.0, bp1_fhigh = -150.0, bp1_gain = 1.0;
    double gain1 = 4.0, gain2I = 1.0, gain2Q = 1.0;             // RXA.c:464-474
    int inselect = 3, copy = 0;
    int levelfade = 1, sbmode = 0;                              // RXA.c:180-181
    double fm_dev = 5000.0, ctcss_freq = 254.1;                 // RXA.c:198,208
    int ctcss_run = 1, fm_nc = 2048;                            // RXA.c:207,209-212
    int lim_run = 0; double lim_gain = 2.5; bool lim_dirty = true;   // FM detector limiter, fmd.c:106-108
    // anf / anr (create_anf / create_anr of create_rxa, RXA.c:278-315): [0] = anf, [1] = anr
    struct Lms { int run = 0, position = 0, taps = 64, delay = 16; double two_mu = 0.0001, gamma = 0.1; bool dirty = true, flush = false; } lms[2];
    // emnr (create_emnr of create_rxa, RXA.c:319-332)
    // snba (wdsp/snb.c) and its bandpass bpsnba (snb.c:696-855; run / position follow the mode, RXA.c:883-917)
    int snba_run = 0, snb_hist_at = 0;
    int fm_hist_at = 0;                                         // ping-pong half that holds this channel's FM fircore delay lines
    int bp1_hist_at = 0;                                        // ... and bp1's (SetRXABandpassRun switches it on WITHOUT the flush of RXAbp1Set)
    bool snba_flush = false, snba_taps_dirty = true, snba_rout_flush = false, snb_dirty = true, snb_flush = false;
    double snba_f_low = 200.0, snba_f_high = 0.0;               // outresamp fc_low / fcin (snb.c:45-46, resample.c:195-204)
    int snb_pos() const {
        if (!snba_run) return -1;
        switch (mode) {
        case QH_LSB: case QH_CWL: case QH_DIGL: case QH_USB: case QH_CWU: case QH_DIGU: return 0;
        case QH_AM: case QH_SAM: case QH_DSB: case QH_FM: return 1;
        default: return -1;
        }
    }
    int emnr_run = 0, emnr_pos = 0, emnr_gain_method = 2, emnr_npe = 0, emnr_ae = 1; bool emnr_dirty = true, emnr_flush = false;
    double emnr_ae_zeta = 0.75, emnr_ae_psi = 20.0, emnr_train_zeta = -2.0, emnr_train_t2 = 0.20;       // emnr.c:332,491-493
    // amsq (create_amsq of create_rxa, RXA.c:158-172)
    int amsq_run = 0; double amsq_tail_thresh = 0.009, amsq_unmute_thresh = 0.010, amsq_max_tail = 1.5; bool amsq_dirty = true;
    int bp1_pos = 0;                                            // SetRXAANFPosition / SetRXAANRPosition set it too (anf.c:236)
    // xwcpagc mode 0 with a position-1 stage behind it: the gain is applied in place at the AGC's spot, not in the epilogue
    bool demod_dirty = true, ctcss_flush = false;
    bool nbp_dirty = true, bp1_dirty = true, nco_dirty = true, epi_dirty = true;
    bool nbp_flush = false, bp1_flush = false;
    bool fix_before() const
    {
        return agc_run && agc_mode == 0 && ((bp1_run && bp1_pos) || (lms[0].run && lms[0].position) || (lms[1].run && lms[1].position) ||
                                            (emnr_run && emnr_pos));
    }
};

struct Engine {
    int device = 0, nch = 0, dsp_size = 0, in_rate = 0, dsp_rate = 0, out_rate = 0;
    int D = 1, dsp_insize = 0, dsp_outsize = 0, front_fold = 1, front_pick = 1;
    unsigned long long epoch = 0;       // bumped by every setter / flush: a captured launch sequence is stale when it moves
    // Launch-sequence replay (qh_rxa_set_graph_replay): a process() call whose arguments and parameters repeat is
    // captured into a hipGraph, one per state of the ping-pong history flags, and replayed.  Everything the host
    // side of process() changes from call to call is those flags, so a slot also records the flags it leaves behind.
    struct GraphKey {
        const void *in = nullptr; void *out = nullptr; long long in_stride = 0, out_stride = 0; int nblk = 0;
        unsigned long long epoch = ~0ull;
        bool operator==(const GraphKey &o) const
        { return in == o.in && out == o.out && in_stride == o.in_stride && out_stride == o.out_stride && nblk == o.nblk && epoch == o.epoch; }
    };
    struct GraphSlot { hipGraphExec_t exec = nullptr; unsigned after = 0; };
    bool graph_on = false, graph_seen = false;
    GraphKey graph_key;
    GraphSlot graph_slot[64];
    long long graph_launches = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // the AM / SAM detectors of a call run beside the FM detector chain: other channels' rows, other state (process_chain)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<ChanCfg> cfg;
    // front (resampler) design
    int front_ntaps = 0, front_P = 0, front_L = 0;
    // device state
    double2 *mask_front = nullptr, *mask_nbp = nullptr, *mask_bp1 = nullptr;
    double2 *tw4096 = nullptr, *tw_inv_front = nullptr, *tw8192 = nullptr;
    int bnfft = kNfft;                      // tile size of the fircore stages: what their masks are built for (4096 or 8192)
    bool band2g = false;                    // 8192-point tiles shared by two lane groups (osfir8k_kernel): masks stored [even | odd]
    bool band6k = false;                    // 6144-point tiles on 384 lanes (osfir6k_kernel)
    int dbg_forms = [] { const char *e = std::getenv("QH_DBG_FORMS"); return e ? std::atoi(e) : 0; }();   // see process_chain
    int band_tile_pref = 0;                 // qh_rxa_set_band_tile: 0 / 4096: 4096-point tiles, 8192: the two-group tiles
    double2 *band_stash = nullptr;          // osfir8s_kernel: [nch][4096], where the tile cut short by the end of a call parks A'
    std::vector<cd> band_mask(const std::vector<cd> &h) const;
    unsigned long long *nco_phase = nullptr, *nco_dphase = nullptr, *nco_parked = nullptr;
    double2 *nco_step = nullptr;
    // output-side oscillator of the front stage (D > 1): per-channel lane table, per-launch tile table, the resampler taps,
    // and the scratch lists of refresh_params (channel list, new phase law)
    double2 *lane_rot = nullptr, *tile_rot = nullptr;
    long long tile_rot_cap = 0;             // tiles per channel
    double *front_taps = nullptr;
    int *retune_list = nullptr;
    unsigned long long *retune_law = nullptr;
    EpiParam *epi = nullptr;
    double2 *hist_front[2] = { nullptr, nullptr }, *hist_nbp[2] = { nullptr, nullptr }, *hist_bp1[2] = { nullptr, nullptr };
    int cur_front = 0, cur_nbp = 0, cur_bp1 = 0, cur_snb = 0;
    double2 *mask_snb = nullptr, *hist_snb[2] = { nullptr, nullptr };
    double2 *buf[2] = { nullptr, nullptr };
    long long buf_cap = 0;                  // complex samples per channel
    // long impulse responses (nc > 4096), per fircore stage s = 0 nbp0, 1 bp1, 2 FM de-emphasis, 3 FM audio filter, 4 bpsnba:
    // long_parts[s] partitions (1: the ordinary path), their masks lmask[s] ([nch or 1][kLongParts][8192]), the stage's last kLongHist
    // input samples lhist[s][ping-pong][nch][kLongHist]; lcat: history + block of the stage being run, ltmp: a partition's output
    int long_parts[5] = { 1, 1, 1, 1, 1 };
    double2 *lmask[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    double2 *lhist[5][2] = { { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr } };
    double2 *lcat = nullptr, *ltmp = nullptr;
    long long lcat_cap = 0;                 // the buf_cap they were made for
    int long_stage_alloc(int sid, bool shared_mask);
    int long_buffers();
    int long_masks_upload(int sid, long long row, const std::vector<cd> &h);
    long long dev_bytes = 0;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_cat;
    int ev_used = 0;
    double last_ms[3] = { 0, 0, 0 };

    // demodulators (allocated on first use)
    bool demod_alloc = false, lists_dirty = true;
    int *list_buf = nullptr, *list_am = nullptr, *list_sam = nullptr, *list_fm = nullptr, *list_bp1 = nullptr, *list_plain = nullptr;
    int n_am = 0, n_sam = 0, n_fm = 0, n_bp1 = 0, n_plain = 0, n_rest = 0, n_usb = 0, n_rb = 0;
    int *list_usb = nullptr, *list_rb = nullptr;       // the non-FM channels without / with a bp1 stage
    // channel pairs for the real filters behind the detectors (osfir_kernel PAIR): FM de-emphasis (one mask for all), bp1 of the AM
    // and of the SAM channels (partners have the same design; count 0 when a channel of the kind has complex taps)
    int *pairs_fm = nullptr, *pairs_am = nullptr, *pairs_sam = nullptr;
    int np_fm = 0, np_am = 0, np_sam = 0;
    bool de_real = false;
    bool all_nbp = false;
    int *list_rest = nullptr;                // the channels that are not FM (the mixed-mode path runs the two kinds on two streams)
    int n_sam0 = 0;                         // the first n_sam0 entries of list_sam have sbmode 0 (no all-pass chains): time-tiled in long calls
    // anf / anr: lists per (filter, position), parameters and state per filter; bp1 lists per position
    // [filter][0] = position 0 (always in `cur`); [filter][1 + b] = position 1 with the data in cur (b = 0: bp1 still to come
    // or not running) or in other (b = 1: bp1 ran at position 0)
    int *list_lms[2][3] = { { nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr } }, n_lms[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
    int *list_bp1p[2] = { nullptr, nullptr }, n_bp1p[2] = { 0, 0 };
    // xwcpagc mode 0 ahead of a position-1 anf / anr / bp1: the fixed gain does not commute with what follows when it
    // changes, so it is applied where the reference applies it; [b] = which buffer holds the channel at that point
    int *list_fix[2] = { nullptr, nullptr }, n_fix[2] = { 0, 0 };
    double *fix_gain = nullptr;
    // emnr: lists like the LMS filters' ([0] position 0; [1 + b] position 1 with the data in cur / other)
    int *list_emnr[3] = { nullptr, nullptr, nullptr }, n_emnr[3] = { 0, 0, 0 };
    // snba: bpsnba lists per position, the blanker's list, its parameters, taps, state and the Toeplitz-inverse scratch
    int *list_snb[2] = { nullptr, nullptr }, n_snb[2] = { 0, 0 }, *list_snba = nullptr, n_snba = 0;
    SnbaParam snba_prm{};
    double *snba_state = nullptr, *snba_hin = nullptr, *snba_hout = nullptr, *snba_scratch = nullptr;
    SnbaIdx *snba_idx = nullptr;
    SnbaTune *snba_tune = nullptr;          // [nch]; host copy below, uploaded when a tuning setter has run
    std::vector<SnbaTune> snba_tune_h;
    bool snba_tune_dirty = true;
    std::vector<char> snb_listed, fm_listed, bp1_listed;
    int snba_alloc();
    int snba_ovrlp = 4;                 // create_rxa's overlap (RXA.c:244): incr = xsize / 4
    void snba_plan(int ovrlp);
    int snba_set_ovrlp(int ovrlp);
    EmnrParam emnr_prm{};
    EmnrChan *emnr_chan = nullptr;
    EmnrScalars *emnr_scal = nullptr;
    double *emnr_state = nullptr, *emnr_window = nullptr, *emnr_GG = nullptr, *emnr_GGS = nullptr, *emnr_zeta = nullptr;
    int *emnr_zeta_true = nullptr;
    bool emnr_tables = false;
    std::vector<double> h_GG, h_GGS, h_zeta; std::vector<int> h_zeta_true; double h_zrange[4] = { 0, 0, 0, 0 };
    int *list_amsq = nullptr, n_amsq = 0;
    AmsqParam *amsq_prm = nullptr;
    AmsqState *amsq_state = nullptr;
    double *amsq_cup = nullptr, *amsq_cdown = nullptr, *amsq_mag = nullptr;
    long long amsq_mag_cap = 0;
    int amsq_ntup = 0, amsq_ntdown = 0;
    LmsParam *lms_prm[2] = { nullptr, nullptr };
    LmsState *lms_state[2] = { nullptr, nullptr };
    int *levelfade = nullptr;
    AmState *am_state = nullptr;
    // carries of the grid-segmented scans (qh_tiled.hpp: a launch reads the state of its channels and leaves the new one here; a
    // commit kernel behind it moves it in): [nch] each
    AmState *am_next = nullptr;
    SnotchState *sn_next = nullptr;
    double *fmdc_next = nullptr;
    double *fm_cin = nullptr, *fm_pw = nullptr;       // xfmd's dc removal taken in the de-emphasis stage's load: fmdc ahead of every tile, mtau^(k + 1)
    long long fm_cin_cap = 0;
    struct FmDcSrc { const double *a; long long stride; int shift; } ;
    const FmDcSrc *band_fmdc = nullptr;               // set around the run_band call of that stage
    double *am_cin = nullptr, *am_pw = nullptr, *am_last = nullptr;      // the fade leveller's carried share taken in bp1's load (osfir_kernel DET 3)
    long long am_cin_cap = 0;
    const FmDcSrc *band_amlv = nullptr;
    AmParam am_prm{};
    PllState *pll_state = nullptr;          // the SAM detector's loop (amd.c) ...
    PllState *fm_pll_state = nullptr;       // ... and the FM detector's (fmd.c): two objects in the reference, each keeps its state while the other runs
    double *fm_again = nullptr;
    // time-tiled FM loop (qh_tiled.hpp): per tile the loop state where its warm-up and where the tile ends, and the count of
    // tiles pll_verify_kernel had to re-run
    double *pll_ends = nullptr;
    long long pll_ends_cap = 0;             // tiles per channel
    double *am_tsum = nullptr;              // [nch][am_tsum_cap][2]: the AM nbp0 tiles' contributions to the fade leveller (osfir_kernel DET 2)
    long long am_tsum_cap = 0;
    double *seg_sum[3] = { nullptr, nullptr, nullptr };     // segment summaries of the multi-workgroup scans: AM / SAM, (unused), snotch
    int *pll_nfixed = nullptr;
    // xwcpagc in time tiles (qh_agc_tiled.hpp): streams RM / fba / hba / volts per listed channel, the tiles' halos, the last samples
    // of the rows, the lanes' states, the final states, tiles re-run
    double *agc_scr = nullptr, *agc_ends = nullptr, *agc_fin = nullptr, *agc_sege = nullptr, *agc_tsum = nullptr;
    double2 *agc_halo = nullptr, *agc_tail = nullptr;
    long long agc_arr = 0, agc_ends_cap = 0, agc_halo_cap = 0;
    int *agc_nfixed = nullptr;
    // SAM sideband modes over time segments (qh_tiled.hpp, sam_sb_*): the chains' transition matrices for the two segment lengths of
    // the current call shape, the segments' zero-state end states and their start states
    double *sb_phi = nullptr, *sb_sum = nullptr, *sb_start = nullptr;
    long long sb_phi_key = -1;
    int set_sb_phi(long long n, int S);
    int pll_check_only = 0;                 // diagnostics (qh_rxa_debug_pll): count unconverged tiles without re-running them
    int agc_form = 0;                       // diagnostics (qh_rxa_debug_agc): 1 = the sample-by-sample form of the wcpAGC loop
    SamChanParam *sam_prm = nullptr;
    PllParam sam_pll_prm{}, fm_pll_prm{};
    SnotchParam *sn_prm = nullptr;
    SnotchState *sn_state = nullptr;
    double2 *mask_de = nullptr, *mask_aud = nullptr, *hist_de[2] = { nullptr, nullptr }, *hist_aud[2] = { nullptr, nullptr };
    int cur_de = 0, cur_aud = 0, fm_nc_built = 0, fm_mp = 0, fm_mp_built = 0, fm_nfft_built = 0;
    unsigned flags() const { return (unsigned)(cur_front | cur_nbp << 1 | cur_bp1 << 2 | cur_de << 3 | cur_aud << 4 | cur_snb << 5); }
    void set_flags(unsigned f) { cur_front = f & 1; cur_nbp = f >> 1 & 1; cur_bp1 = f >> 2 & 1; cur_de = f >> 3 & 1; cur_aud = f >> 4 & 1; cur_snb = f >> 5 & 1; }
    void drop_graphs() { for (auto &g : graph_slot) if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; } }
    int process_replayed(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk);
    AgcParam *agc_prm = nullptr;
    AgcState *agc_state = nullptr;
    // xwcpagc's ring in full (RB_SIZE entries per channel, qh_demod.hpp: agc_long_mirror_kernel), made when a state machine first runs
    double2 *agc_lring = nullptr;
    double *agc_labs = nullptr;
    int *agc_lout = nullptr, *agc_rewin_list = nullptr;
    AgcParam *lim_prm = nullptr;        // FM detector limiter: a wcpAGC of its own (fmd.c:48-72)
    AgcState *lim_state = nullptr;
    int *list_lim = nullptr, n_lim = 0;
    bool meters_on = false;
    MeterState *m_adc = nullptr, *m_s = nullptr, *m_agc = nullptr;
    MeterParam m_prm{};
    // meters fused into the nbp0 launch of the linear fast path (qh_osfir.hpp METER): chunk partials of the stage's input and
    // output, the chunk weights, and g^2 of a fixed AGC gain that the output matrix applies behind the agc meter's tap
    double2 *m_part[2] = { nullptr, nullptr };
    long long m_part_cap = 0;               // chunks per channel
    double *m_w = nullptr, *m_g2 = nullptr;
    int meters_alloc();
    int *list_agc_cur = nullptr, *list_agc_other = nullptr;
    int n_agc_cur = 0, n_agc_other = 0;
    int agc_last_tiled = 0;             // channels whose xwcpagc took the time tiles in the last call (diagnostics)
    int n_agc_cur_stale = 0, n_agc_other_stale = 0;     // ... of which, at the lists' ends, channels whose attack window moved in mid-stream

    ~Engine();
    int init();
    int refresh_params();
    int refresh_demod();
    int run_front(const double2 *src, long long src_stride, double2 *dst, long long dst_stride, const EpiParam *ep,
                  long long n_in, long long n_mid, const int *list = nullptr, int nlist = 0, int part = 0);
    const unsigned char *pk_src = nullptr;      // set for the duration of a qh_rxa_process_packed call
    PackedFmt pk{};
    EgressFmt eg{};                             // set (kind != 0) for the duration of a qh_rxa_process_audio call
    double2 *abuf = nullptr;                    // complex-double staging of an audio call whose last stage cannot narrow in its store
    long long abuf_cap = 0;
    int ensure_abuf(long long n);
    void pack_audio(const double2 *src, long long src_stride, long long n);
    void run_band(const double2 *src, long long src_stride, double2 *dst, long long dst_stride, const EpiParam *ep,
                  long long n_mid, const double2 *mask, long long mask_stride, double2 **hist, int &hc, int P,
                  const int *list, int nlist, bool meter = false, bool egress = false, int det = 0, double *det_out = nullptr,
                  long long det_stride = 0, const int *pairs = nullptr, int npairs = 0);
    int ensure_buffers(long long n_mid);
    int ensure_meter_partials(long long n_mid, int lout);
    int emnr_alloc();
    int process(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk);
    int process_chain(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk);
    qh_rat *rsmpout = nullptr;          // xresample out (wdsp/RXA.c:596), only when out_rate != dsp_rate
    qh_rat *rsmpin = nullptr;           // xresample in for the rate ratios the overlap-save front stage does not cover (D == 0)
    double2 *fbuf = nullptr;            // its input: the shifted samples at in_rate
    long long fbuf_cap = 0;
    double2 *obuf = nullptr;
    long long obuf_cap = 0;
    void tick(int cat);
};

Engine::~Engine()
{
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    drop_graphs();
    if (rsmpout) qh_rat_destroy(rsmpout);
    if (rsmpin) qh_rat_destroy(rsmpin);
    (void)hipFree(fbuf);
    (void)hipFree(obuf); (void)hipFree(abuf);
    for (int i = 0; i < 5; i++) { (void)hipFree(lmask[i]); (void)hipFree(lhist[i][0]); (void)hipFree(lhist[i][1]); }
    (void)hipFree(lcat); (void)hipFree(ltmp);
    (void)hipFree(band_stash);
    (void)hipFree(mask_front); (void)hipFree(mask_nbp); (void)hipFree(mask_bp1); (void)hipFree(tw4096); (void)hipFree(tw_inv_front); (void)hipFree(tw8192);
    (void)hipFree(nco_phase); (void)hipFree(nco_dphase); (void)hipFree(nco_parked); (void)hipFree(nco_step); (void)hipFree(epi);
    (void)hipFree(lane_rot); (void)hipFree(tile_rot); (void)hipFree(front_taps); (void)hipFree(retune_list); (void)hipFree(retune_law);
    for (int i = 0; i < 2; i++) { (void)hipFree(hist_front[i]); (void)hipFree(hist_nbp[i]); (void)hipFree(hist_bp1[i]); (void)hipFree(buf[i]); }
    (void)hipFree(list_buf); (void)hipFree(levelfade); (void)hipFree(am_state); (void)hipFree(am_next); (void)hipFree(sn_next); (void)hipFree(fmdc_next); (void)hipFree(fm_cin); (void)hipFree(fm_pw); (void)hipFree(am_cin); (void)hipFree(am_pw); (void)hipFree(am_last); (void)hipFree(pll_state); (void)hipFree(fm_pll_state); (void)hipFree(fm_again); (void)hipFree(pll_ends); (void)hipFree(pll_nfixed); (void)hipFree(am_tsum);
    (void)hipFree(sb_phi); (void)hipFree(sb_sum); (void)hipFree(sb_start);
    (void)hipFree(agc_scr); (void)hipFree(agc_ends); (void)hipFree(agc_fin); (void)hipFree(agc_halo); (void)hipFree(agc_tail); (void)hipFree(agc_nfixed); (void)hipFree(agc_sege); (void)hipFree(agc_tsum);
    for (double *&q : seg_sum) { (void)hipFree(q); q = nullptr; }
    (void)hipFree(agc_prm); (void)hipFree(agc_state); (void)hipFree(agc_lring); (void)hipFree(agc_labs); (void)hipFree(agc_lout); (void)hipFree(agc_rewin_list); (void)hipFree(lim_prm); (void)hipFree(lim_state); (void)hipFree(list_lim); (void)hipFree(m_adc); (void)hipFree(m_s); (void)hipFree(m_agc); (void)hipFree(m_part[0]); (void)hipFree(m_part[1]); (void)hipFree(m_w); (void)hipFree(m_g2);
    (void)hipFree(mask_snb); (void)hipFree(hist_snb[0]); (void)hipFree(hist_snb[1]); (void)hipFree(snba_state); (void)hipFree(snba_hin);
    (void)hipFree(snba_hout); (void)hipFree(snba_scratch); (void)hipFree(snba_idx); (void)hipFree(snba_tune);
    (void)hipFree(emnr_chan); (void)hipFree(emnr_scal); (void)hipFree(emnr_state); (void)hipFree(emnr_window); (void)hipFree(emnr_GG);
    (void)hipFree(emnr_GGS); (void)hipFree(emnr_zeta); (void)hipFree(emnr_zeta_true);
    (void)hipFree(amsq_prm); (void)hipFree(amsq_state); (void)hipFree(amsq_cup); (void)hipFree(amsq_cdown); (void)hipFree(amsq_mag);
    (void)hipFree(fix_gain); (void)hipFree(lms_prm[0]); (void)hipFree(lms_prm[1]); (void)hipFree(lms_state[0]); (void)hipFree(lms_state[1]);
    (void)hipFree(sam_prm); (void)hipFree(sn_prm); (void)hipFree(sn_state); (void)hipFree(mask_de); (void)hipFree(mask_aud);
    for (int i = 0; i < 2; i++) { (void)hipFree(hist_de[i]); (void)hipFree(hist_aud[i]); }
    for (auto e : ev) (void)hipEventDestroy(e);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

static int upload(double2 *dst, const std::vector<cd> &v, hipStream_t s)
{
    QH_HIP(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(cd), hipMemcpyHostToDevice, s));
    QH_HIP(hipStreamSynchronize(s));        // the host vector dies with the caller's scope
    return QH_OK;
}

int Engine::init()
{
    QH_HIP(hipSetDevice(device));
    if (!stream) { QH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
    // pre_main_build, wdsp/channel.c:39-47
    dsp_insize = D > 0 ? dsp_size * D : (int)((long long)dsp_size * in_rate / dsp_rate);
    if (D == 0) {
        // create_resample(..., in_rate, dsp_rate, 0.0, 0, 1.0), wdsp/RXA.c:48-57
        const ResamplerDesign rd = design_resampler(in_rate, dsp_rate, 0.0, 0, 1.0);
        std::vector<double> taps(rd.h);
        for (double &v : taps) v /= (double)rd.L;           // qh_rat applies the gain `interp` itself (quisk_cInterpDecim's convention)
        rsmpin = qh_rat_create(device, nch, taps.data(), rd.ncoef, rd.L, rd.M, QH_F64, stream);
        if (!rsmpin) return QH_ERR_HIP;
    }
    dsp_outsize = out_rate >= dsp_rate ? dsp_size * (out_rate / dsp_rate) : dsp_size / (dsp_rate / out_rate);    // channel.c:47-50
    if (out_rate != dsp_rate) {
        // create_resample(..., dsp_rate, out_rate, 0.0, 0, 1.0), wdsp/RXA.c:474-484; the polyphase loop of xresample
        // (resample.c:120-157) is quisk_cInterpDecim's with the gain already in the taps
        const ResamplerDesign rd = design_resampler(dsp_rate, out_rate, 0.0, 0, 1.0);
        std::vector<double> taps(rd.h);
        for (double &v : taps) v /= (double)rd.L;
        rsmpout = qh_rat_create(device, nch, taps.data(), rd.ncoef, rd.L, rd.M, QH_F64, stream);
        if (!rsmpout) return QH_ERR_HIP;
    }
    cfg.assign((size_t)nch, ChanCfg());
    snba_tune_h.assign((size_t)nch, SnbaTune{ 64, 2, 10, 2, 2, 0, 8.0, 20.0, 0.5 });         // create_snba's arguments, RXA.c:183-202

    std::vector<cd> tw = fft_twiddle_table(kNfft);
    QH_HIP(dev_alloc(&tw4096, tw.size()));
    if (int rc = upload(tw4096, tw, stream)) return rc;
    dev_bytes += tw.size() * sizeof(cd);
    tw = fft_twiddle_table(kBandNfftMax);
    QH_HIP(dev_alloc(&tw8192, tw.size()));
    if (int rc = upload(tw8192, tw, stream)) return rc;
    dev_bytes += tw.size() * sizeof(cd);

    if (D > 1) {
        // calc_resample, wdsp/resample.c:35-72 (L = 1): y[m] = sum_j h[j] x[D*m - j]
        ResamplerDesign rd = design_resampler(in_rate, dsp_rate, 0.0, 0, 1.0);
        if (rd.L != 1 || rd.M != D) return set_error(QH_ERR_UNSUPPORTED, "resampler L/M = %d/%d not supported", rd.L, rd.M);
        front_ntaps = rd.ncoef;
        // spectral fold by min(D, 8); the rest of the decimation (D = 16) keeps every second folded sample
        front_fold = D > 8 ? 8 : D;
        front_pick = D / front_fold;
        front_P = ((front_ntaps - 1 + front_fold - 1) / front_fold) * front_fold;
        front_L = (((kNfft - front_P) / front_fold) / front_pick) * front_pick;
        if (front_P > kHistFront) return set_error(QH_ERR_UNSUPPORTED, "resampler history %d too long", front_P);
        // the masks are per channel (taps modulated by the channel's shift): front_mask_kernel builds them in refresh_params
        QH_HIP(dev_alloc(&mask_front, (size_t)nch * kNfft));
        QH_HIP(dev_alloc(&lane_rot, (size_t)nch * NT));
        QH_HIP(dev_alloc(&front_taps, (size_t)front_ntaps));
        QH_HIP(dev_alloc(&retune_list, (size_t)nch));
        QH_HIP(dev_alloc(&retune_law, (size_t)nch * 2));
        QH_HIP(hipMemcpyAsync(front_taps, rd.h.data(), (size_t)front_ntaps * sizeof(double), hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&front_mask_kernel<kNfft>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (TileFft<kNfft, false, double2>::kLdsBytes)));
        std::vector<cd> twi = fft_twiddle_table(kNfft / front_fold);
        QH_HIP(dev_alloc(&tw_inv_front, twi.size()));
        if (int rc = upload(tw_inv_front, twi, stream)) return rc;
        dev_bytes += twi.size() * sizeof(cd) + (size_t)nch * (kNfft + NT) * sizeof(double2) + (size_t)front_ntaps * 8 + (size_t)nch * 20;
        for (int i = 0; i < 2; i++) {
            QH_HIP(dev_alloc(&hist_front[i], (size_t)nch * kHistFront));
            QH_HIP(hipMemsetAsync(hist_front[i], 0, (size_t)nch * kHistFront * sizeof(double2), stream));
            dev_bytes += (size_t)nch * kHistFront * sizeof(double2);
        }
    }
    QH_HIP(dev_alloc(&mask_nbp, (size_t)nch * kBandNfftMax));
    QH_HIP(dev_alloc(&mask_bp1, (size_t)nch * kBandNfftMax));
    dev_bytes += 2ll * nch * kBandNfftMax * sizeof(double2);
    for (int i = 0; i < 2; i++) {
        QH_HIP(dev_alloc(&hist_nbp[i], (size_t)nch * kHistBand));
        QH_HIP(dev_alloc(&hist_bp1[i], (size_t)nch * kHistBand));
        QH_HIP(hipMemsetAsync(hist_nbp[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
        QH_HIP(hipMemsetAsync(hist_bp1[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
        dev_bytes += 2ll * nch * kHistBand * sizeof(double2);
    }
    QH_HIP(dev_alloc(&nco_phase, (size_t)nch));
    QH_HIP(dev_alloc(&nco_dphase, (size_t)nch));
    QH_HIP(dev_alloc(&nco_parked, (size_t)nch));
    QH_HIP(hipMemsetAsync(nco_parked, 0, (size_t)nch * sizeof(unsigned long long), stream));
    QH_HIP(hipMemsetAsync(nco_dphase, 0, (size_t)nch * sizeof(unsigned long long), stream));
    QH_HIP(dev_alloc(&nco_step, (size_t)nch));
    QH_HIP(dev_alloc(&epi, (size_t)nch));
    QH_HIP(hipMemsetAsync(nco_phase, 0, (size_t)nch * sizeof(unsigned long long), stream));
    dev_bytes += (size_t)nch * (16 + 16 + sizeof(EpiParam));

    // dynamic LDS of the overlap-save kernels
#define QH_SET_LDS(D, ...) QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<double, 4096, D, __VA_ARGS__>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_lds_bytes<double, 4096, D, true>())))
    QH_SET_LDS(1, false); QH_SET_LDS(1, false, false, true);
    QH_SET_LDS(1, false, false, false, false, true); QH_SET_LDS(1, false, false, true, false, true);
#define QH_SET_LDS8(...) QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<double, kBandNfftMax, 1, false, false, __VA_ARGS__>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (osfir_lds_bytes<double, kBandNfftMax, 1, true>())))
    QH_SET_LDS8(false, false, false); QH_SET_LDS8(true, false, false); QH_SET_LDS8(false, false, true); QH_SET_LDS8(true, false, true);
#undef QH_SET_LDS8
#define QH_SET_LDS6K(...) QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir6k_kernel<__VA_ARGS__>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, osfir6k_lds_bytes()))
    QH_SET_LDS6K(false, false); QH_SET_LDS6K(true, false); QH_SET_LDS6K(false, true); QH_SET_LDS6K(true, true);
#undef QH_SET_LDS6K
#define QH_SET_LDS2G(...) QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir8k_kernel<__VA_ARGS__>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, osfir8k_lds_bytes()))
    QH_SET_LDS2G(false, false); QH_SET_LDS2G(true, false); QH_SET_LDS2G(false, true); QH_SET_LDS2G(true, true);
#undef QH_SET_LDS2G
#ifdef QH_EXP_BAND8_SEQ
    QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir8s_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kOsfir8kImage));
    QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir8s_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kOsfir8kImage));
#endif
    QH_SET_LDS(2, false, false, false, true, false, true); QH_SET_LDS(4, false, false, false, true, false, true); QH_SET_LDS(8, false, false, false, true, false, true);
    QH_SET_LDS(2, false, true, false, true, false, true); QH_SET_LDS(4, false, true, false, true, false, true); QH_SET_LDS(8, false, true, false, true, false, true);
#undef QH_SET_LDS
    QH_HIP(hipStreamSynchronize(stream));
    return QH_OK;
}

// fixed-point turns for a frequency ratio f / rate
static unsigned long long turns_fx(double f, double rate)
{
    long double t = (long double)f / (long double)rate;
    t -= floorl(t);
    long double s = t * 18446744073709551616.0L;
    if (s >= 18446744073709551616.0L) s = 0;
    return (unsigned long long)s;
}

int Engine::refresh_params()
{
    // one pass over the channels; upload only what changed
    std::vector<cd> last_nbp, last_bp1, last_nbp_h, last_bp1_h;      // masks and impulse responses of the last design made
    const ChanCfg *last_nbp_cfg = nullptr, *last_bp1_cfg = nullptr;
    // Oscillator changes (SetRXAShiftFreq / SetRXAShiftRun).  With a front FIR stage (D > 1) the oscillator sits behind
    // the filter: first the stored raw history of every changed channel is re-expressed for its new phase law (the kernel
    // reads the old law from the device arrays, so it goes first), then the arrays are updated, then the channel's
    // modulated mask and phasor tables are rebuilt.
    std::vector<int> nco_list;
    if (D > 1) {
        std::vector<unsigned long long> law;
        for (int ch = 0; ch < nch; ch++) {
            const ChanCfg &c = cfg[(size_t)ch];
            if (!c.nco_dirty) continue;
            nco_list.push_back(ch);
            const bool flip = (c.shift_run != 0) != (c.shift_on_device != 0);
            law.push_back(flip ? (c.shift_run ? 2ull : 1ull) : 0ull);
            law.push_back(c.shift_run ? turns_fx(c.shift_freq, (double)in_rate) : 0ull);
        }
        if (!nco_list.empty()) {
            QH_HIP(hipMemcpyAsync(retune_list, nco_list.data(), nco_list.size() * sizeof(int), hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemcpyAsync(retune_law, law.data(), law.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(nco_retune_hist_kernel, dim3((kHistFront + NT - 1) / NT, (unsigned)nco_list.size()), dim3(NT), 0, stream,
                               hist_front[cur_front], kHistFront, nco_phase, nco_dphase, nco_parked, (const int *)retune_list,
                               (const unsigned long long *)retune_law);
            QH_HIP(hipStreamSynchronize(stream));           // the host vectors die with this scope
        }
    }
    for (int ch = 0; ch < nch; ch++) {
        ChanCfg &c = cfg[(size_t)ch];
        if (c.nco_dirty) {
            if ((c.shift_run != 0) != (c.shift_on_device != 0)) {
                hipLaunchKernelGGL(nco_park_kernel, dim3(1), dim3(1), 0, stream, nco_phase, nco_parked, ch, c.shift_run ? 1 : 0);
                c.shift_on_device = c.shift_run ? 1 : 0;
            }
            // calc_shift, wdsp/shift.c:29-34: delta = 2*pi*shift/rate per input sample
            unsigned long long d = c.shift_run ? turns_fx(c.shift_freq, (double)in_rate) : 0ull;
            QH_HIP(hipMemcpyAsync(nco_dphase + ch, &d, sizeof(d), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.nco_dirty = false;
        }
        if (c.epi_dirty) {
            // xwcpagc mode 0 (wcpAGC.c:167-175) then xpanel (patchpanel.c:55-101) as one 2x2 real matrix
            // (with a position-1 anf / anr / bp1 behind it the gain is applied at the AGC's own spot instead: fix_before)
            const double g = (c.agc_run && c.agc_mode == 0 && !c.fix_before()) ? c.agc_fixed : 1.0;
            const double g2 = g * g;            // the agc meter reads |g z|^2 off the signal ahead of the output matrix
            if (m_g2) QH_HIP(hipMemcpyAsync(m_g2 + ch, &g2, sizeof(double), hipMemcpyHostToDevice, stream));
            if (fix_gain) QH_HIP(hipMemcpyAsync(fix_gain + ch, &c.agc_fixed, sizeof(double), hipMemcpyHostToDevice, stream));
            const double gI = c.gain1 * c.gain2I, gQ = c.gain1 * c.gain2Q;
            const double sI = (double)(c.inselect >> 1), sQ = (double)(c.inselect & 1);
            EpiParam e;
            switch (c.copy) {
            default:
            case 0: e.a = gI * sI * g; e.b = 0; e.c = 0; e.d = gQ * sQ * g; break;
            case 1: e.a = gI * sI * g; e.b = 0; e.c = gQ * sI * g; e.d = 0; break;
            case 2: e.a = 0; e.b = gI * sQ * g; e.c = 0; e.d = gQ * sQ * g; break;
            case 3: e.a = 0; e.b = gI * sQ * g; e.c = gQ * sI * g; e.d = 0; break;
            }
            QH_HIP(hipMemcpyAsync(epi + ch, &e, sizeof(e), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.epi_dirty = false;
        }
        if (c.nbp_dirty) c.snb_dirty = true;        // bpsnba's nbp shares nc, window, auto-increase, mp and the notch database with nbp0
        if (c.snb_dirty && c.snba_run && mask_snb) {
            // recalc_bpsnba_filter (snb.c:807-822) with RXAbpsnbaCheck's frequencies (RXA.c:829-881): 250..5700 Hz on the mode's side
            double f_low = 0.0, f_high = 0.0;
            int run_notches = 0;
            switch (c.mode) {
            case QH_LSB: case QH_CWL: case QH_DIGL: f_low = -5700.0; f_high = -250.0; run_notches = c.fnfrun; break;
            case QH_USB: case QH_CWU: case QH_DIGU: f_low = 250.0; f_high = 5700.0; run_notches = c.fnfrun; break;
            case QH_AM: case QH_SAM: case QH_DSB: case QH_FM: f_low = 250.0; f_high = 5700.0; break;
            default: break;
            }
            std::vector<cd> h;
            const double scale = 1.0 / (double)(2 * dsp_size);
            if (long_parts[4] > 1) if (int rc = long_stage_alloc(4, false)) return rc;
            if (run_notches) {
                const double offset = c.ndb_tunefreq + c.ndb_shift;
                const double minwidth = (c.nbp_wintype == 1 ? 2200.0 : 1600.0) / (c.nbp_nc / 256) * ((double)dsp_rate / 48000);
                std::vector<std::pair<double, double>> bands = make_nbp(c.notches, minwidth, c.autoincr, f_low + offset, f_high + offset, nullptr);
                for (auto &b : bands) { b.first -= offset; b.second -= offset; }
                h = fir_mbandpass(c.nbp_nc, bands, (double)dsp_rate, scale, c.nbp_wintype);
            } else
                h = fir_bandpass(c.nbp_nc, f_low, f_high, (double)dsp_rate, c.nbp_wintype, 1, scale);
            if (c.mp) h = mp_imp(h, 16, 0);
            for (auto &v : h) v *= (double)(2 * dsp_size);
            if (long_parts[4] > 1) if (int rc = long_masks_upload(4, ch, h)) return rc;
            if ((int)h.size() > kLongPart) h.resize((size_t)kLongPart);      // (the one-tile mask is not used then)
            const std::vector<cd> m = band_mask(h);
            QH_HIP(hipMemcpyAsync(mask_snb + (size_t)ch * kBandNfftMax, m.data(), (size_t)bnfft * sizeof(cd), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.snb_dirty = false;
        }
        if (c.snb_flush && hist_snb[0]) {           // setNc_fircore zeroes the delay line
            for (int i = 0; i < 2; i++) {
                QH_HIP(hipMemsetAsync(hist_snb[i] + (size_t)ch * kHistBand, 0, kHistBand * sizeof(double2), stream));
                if (lhist[4][i]) QH_HIP(hipMemsetAsync(lhist[4][i] + (size_t)ch * kLongHist, 0, kLongHist * sizeof(double2), stream));
            }
        }
        c.snb_flush = false;
        if (c.nbp_dirty) {
            // calc_nbp_impulse without notches, wdsp/nbp.c:234-238; identity when the filter is off
            bool same = !c.fnfrun && last_nbp_cfg && !last_nbp_cfg->fnfrun && last_nbp_cfg->nbp_run == c.nbp_run && last_nbp_cfg->nbp_nc == c.nbp_nc &&
                        last_nbp_cfg->nbp_wintype == c.nbp_wintype && last_nbp_cfg->nbp_flow == c.nbp_flow &&
                        last_nbp_cfg->nbp_fhigh == c.nbp_fhigh && last_nbp_cfg->nbp_gain == c.nbp_gain && last_nbp_cfg->mp == c.mp;
            if (!same) {
                std::vector<cd> h;
                if (c.nbp_run && c.fnfrun) {
                    // calc_nbp_impulse with the notches, wdsp/nbp.c:221-232: bands in absolute frequency, filter in baseband
                    const double offset = c.ndb_tunefreq + c.ndb_shift;
                    const double minwidth = (c.nbp_wintype == 1 ? 2200.0 : 1600.0) / (c.nbp_nc / 256) * ((double)dsp_rate / 48000);
                    std::vector<std::pair<double, double>> bands =
                        make_nbp(c.notches, minwidth, c.autoincr, c.nbp_flow + offset, c.nbp_fhigh + offset, nullptr);
                    for (auto &b : bands) { b.first -= offset; b.second -= offset; }
                    h = fir_mbandpass(c.nbp_nc, bands, (double)dsp_rate, c.nbp_gain / (double)(2 * dsp_size), c.nbp_wintype);
                } else if (c.nbp_run)
                    h = fir_bandpass(c.nbp_nc, c.nbp_flow, c.nbp_fhigh, (double)dsp_rate, c.nbp_wintype, 1,
                                     c.nbp_gain / (double)(2 * dsp_size));
                else
                    h.assign(1, cd(1.0, 0.0));
                if (c.nbp_run && c.mp) h = mp_imp(h, 16, 0);            // calc_fircore, wdsp/firmin.c:327-328
                // the reference's unnormalised inverse FFT of 2*size points restores the 1/(2*size)
                if (c.nbp_run) for (auto &v : h) v *= (double)(2 * dsp_size);
                last_nbp_h = h;
                if ((int)h.size() > kLongPart) h.resize((size_t)kLongPart);      // (the one-tile mask is not used then)
                last_nbp = band_mask(h);
                last_nbp_cfg = &c;
            }
            if (long_parts[0] > 1) if (int rc = long_masks_upload(0, ch, last_nbp_h)) return rc;
            QH_HIP(hipMemcpyAsync(mask_nbp + (size_t)ch * kBandNfftMax, last_nbp.data(), (size_t)bnfft * sizeof(cd), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.nbp_dirty = false;
        }
        if (c.bp1_dirty) {
            bool same = last_bp1_cfg && last_bp1_cfg->bp1_run == c.bp1_run && last_bp1_cfg->bp1_nc == c.bp1_nc &&
                        last_bp1_cfg->bp1_wintype == c.bp1_wintype && last_bp1_cfg->bp1_flow == c.bp1_flow &&
                        last_bp1_cfg->bp1_fhigh == c.bp1_fhigh && last_bp1_cfg->bp1_gain == c.bp1_gain && last_bp1_cfg->mp == c.mp;
            if (!same) {
                std::vector<cd> h;
                if (c.bp1_run) {
                    h = fir_bandpass(c.bp1_nc, c.bp1_flow, c.bp1_fhigh, (double)dsp_rate, c.bp1_wintype, 1,
                                     c.bp1_gain / (double)(2 * dsp_size));     // wdsp/bandpass.c:302
                    if (c.mp) h = mp_imp(h, 16, 0);
                    for (auto &v : h) v *= (double)(2 * dsp_size);
                } else {
                    h.assign(1, cd(1.0, 0.0));
                }
                last_bp1_h = h;
                if ((int)h.size() > kLongPart) h.resize((size_t)kLongPart);
                last_bp1 = band_mask(h);
                last_bp1_cfg = &c;
            }
            if (long_parts[1] > 1) if (int rc = long_masks_upload(1, ch, last_bp1_h)) return rc;
            QH_HIP(hipMemcpyAsync(mask_bp1 + (size_t)ch * kBandNfftMax, last_bp1.data(), (size_t)bnfft * sizeof(cd), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.bp1_dirty = false;
            lists_dirty = true;             // the channel pairs of the real bp1 filters follow the designs
        }
        if (c.nbp_flush) {      // setNc_fircore re-plans and so zeroes the delay line, wdsp/firmin.c:454-466
            for (int i = 0; i < 2; i++) {
                QH_HIP(hipMemsetAsync(hist_nbp[i] + (size_t)ch * kHistBand, 0, kHistBand * sizeof(double2), stream));
                if (lhist[0][i]) QH_HIP(hipMemsetAsync(lhist[0][i] + (size_t)ch * kLongHist, 0, kLongHist * sizeof(double2), stream));
            }
            c.nbp_flush = false;
        }
        if (c.bp1_flush) {      // flush_bandpass on off->on (RXA.c:825) and setNc_fircore
            for (int i = 0; i < 2; i++) {
                QH_HIP(hipMemsetAsync(hist_bp1[i] + (size_t)ch * kHistBand, 0, kHistBand * sizeof(double2), stream));
                if (lhist[1][i]) QH_HIP(hipMemsetAsync(lhist[1][i] + (size_t)ch * kLongHist, 0, kLongHist * sizeof(double2), stream));
            }
            c.bp1_flush = false;
        }
    }
    if (!nco_list.empty())      // retune_list still holds the channels; the new dphase values are in place
        hipLaunchKernelGGL((front_mask_kernel<kNfft>), dim3((unsigned)nco_list.size()), dim3(NT), (size_t)(TileFft<kNfft, false, double2>::kLdsBytes),
                           stream, (const double *)front_taps, front_ntaps, (const unsigned long long *)nco_dphase, (const int *)retune_list, front_fold,
                           (const double2 *)tw4096, mask_front, lane_rot, nco_step, 1);
    return QH_OK;
}

// create_meter x3 (RXA.c:69-82,142-155,361-374): tau 0.1 s for average and peak decay; flush_meter -> -400 dB
int Engine::meters_alloc()
{
    if (m_adc) return QH_OK;
    const double rate = (double)dsp_rate;
    std::vector<MeterState> init((size_t)nch, MeterState{ 0.0, 0.0, -400.0, -400.0 });
    for (MeterState **pm : { &m_adc, &m_s, &m_agc }) {
        QH_HIP(dev_alloc(pm, (size_t)nch));
        QH_HIP(hipMemcpyAsync(*pm, init.data(), (size_t)nch * sizeof(MeterState), hipMemcpyHostToDevice, stream));
    }
    m_prm.mult_average = std::exp(-1.0 / (rate * 0.100));
    m_prm.mult_peak = std::exp(-1.0 / (rate * 0.100));
    std::vector<double> w(64), g2((size_t)nch);
    for (int i = 0; i < 64; i++) w[(size_t)i] = (1.0 - m_prm.mult_average) * std::pow(m_prm.mult_average, (double)(63 - i));
    for (int ch = 0; ch < nch; ch++) {
        const ChanCfg &c = cfg[(size_t)ch];
        g2[(size_t)ch] = (c.agc_run && c.agc_mode == 0 && !c.fix_before()) ? c.agc_fixed * c.agc_fixed : 1.0;
    }
    QH_HIP(dev_alloc(&m_w, (size_t)64));
    QH_HIP(dev_alloc(&m_g2, (size_t)nch));
    QH_HIP(hipMemcpyAsync(m_w, w.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
    QH_HIP(hipMemcpyAsync(m_g2, g2.data(), g2.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    QH_HIP(hipStreamSynchronize(stream));
    dev_bytes += (long long)nch * (3 * sizeof(MeterState) + 8) + 512;
    return QH_OK;
}

// Demodulator state, channel lists and FM filters (only engines that run AM/SAM/FM channels get here).
int Engine::refresh_demod()
{
    const double rate = (double)dsp_rate;
    if (!demod_alloc) {
        QH_HIP(dev_alloc(&list_buf, (size_t)nch * 33));
        list_rest = list_buf + 24 * nch; list_usb = list_buf + 25 * nch; list_rb = list_buf + 26 * nch;
        pairs_fm = list_buf + 27 * nch; pairs_am = list_buf + 29 * nch; pairs_sam = list_buf + 31 * nch;
        list_amsq = list_buf + 17 * nch;
        for (int k = 0; k < 3; k++) list_emnr[k] = list_buf + (18 + k) * nch;
        for (int f = 0; f < 2; f++) for (int k = 0; k < 3; k++) list_lms[f][k] = list_buf + (7 + 3 * f + k) * nch;
        list_snb[0] = list_buf + 21 * nch; list_snb[1] = list_buf + 22 * nch; list_snba = list_buf + 23 * nch;
        list_bp1p[0] = list_buf + 13 * nch; list_bp1p[1] = list_buf + 14 * nch;
        list_fix[0] = list_buf + 15 * nch; list_fix[1] = list_buf + 16 * nch;
        QH_HIP(dev_alloc(&fix_gain, (size_t)nch));
        list_am = list_buf; list_sam = list_buf + nch; list_fm = list_buf + 2 * nch; list_bp1 = list_buf + 3 * nch;
        list_plain = list_buf + 4 * nch; list_agc_cur = list_buf + 5 * nch; list_agc_other = list_buf + 6 * nch;
        if (int rc = meters_alloc()) return rc;
        QH_HIP(dev_alloc(&agc_prm, (size_t)nch));
        QH_HIP(dev_alloc(&agc_state, (size_t)nch));
        QH_HIP(hipMemsetAsync(agc_state, 0, (size_t)nch * sizeof(AgcState), stream));
        {
            std::vector<int> oi((size_t)nch, kAgcRing - 1);         // out_index = -1 (calc_wcpagc, wcpAGC.c:34)
            for (int c = 0; c < nch; c++)
                QH_HIP(hipMemcpyAsync(&agc_state[c].out_index, &oi[(size_t)c], sizeof(int), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        dev_bytes += (long long)nch * (sizeof(AgcParam) + sizeof(AgcState));
        QH_HIP(dev_alloc(&levelfade, (size_t)nch));
        QH_HIP(dev_alloc(&am_state, (size_t)nch));
        QH_HIP(dev_alloc(&am_next, (size_t)nch));
        QH_HIP(dev_alloc(&sn_next, (size_t)nch));
        QH_HIP(dev_alloc(&fmdc_next, (size_t)nch));
        QH_HIP(dev_alloc(&pll_state, (size_t)nch));
        QH_HIP(dev_alloc(&fm_pll_state, (size_t)nch));
        QH_HIP(dev_alloc(&fm_again, (size_t)nch));
        QH_HIP(dev_alloc(&pll_nfixed, (size_t)1));
        QH_HIP(hipMemsetAsync(pll_nfixed, 0, sizeof(int), stream));
        QH_HIP(dev_alloc(&sam_prm, (size_t)nch));
        QH_HIP(dev_alloc(&sn_prm, (size_t)nch));
        QH_HIP(dev_alloc(&sn_state, (size_t)nch));
        QH_HIP(hipMemsetAsync(am_state, 0, (size_t)nch * sizeof(AmState), stream));
        QH_HIP(hipMemsetAsync(pll_state, 0, (size_t)nch * sizeof(PllState), stream));
        QH_HIP(hipMemsetAsync(fm_pll_state, 0, (size_t)nch * sizeof(PllState), stream));
        QH_HIP(hipMemsetAsync(sn_state, 0, (size_t)nch * sizeof(SnotchState), stream));
        QH_HIP(dev_alloc(&mask_de, (size_t)kBandNfftMax));
        QH_HIP(dev_alloc(&mask_aud, (size_t)kBandNfftMax));
        for (int i = 0; i < 2; i++) {
            QH_HIP(dev_alloc(&hist_de[i], (size_t)nch * kHistBand));
            QH_HIP(dev_alloc(&hist_aud[i], (size_t)nch * kHistBand));
            QH_HIP(hipMemsetAsync(hist_de[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
            QH_HIP(hipMemsetAsync(hist_aud[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
        }
        dev_bytes += (long long)nch * (5 * 4 + 4 + sizeof(AmState) + sizeof(PllState) + 8 + sizeof(SamChanParam) +
                                      sizeof(SnotchParam) + sizeof(SnotchState) + 4 * kHistBand * sizeof(double2)) +
                     2ll * kBandNfftMax * sizeof(double2);
        // init_amd (wdsp/amd.c:72-89) with create_rxa's constants (RXA.c:183-189)
        {
            const double zeta = 1.0, omegaN = 250.0, tauR = 0.02, tauI = 1.4;
            PllParam &q = sam_pll_prm;
            q.omega_min = kTwoPiRef * -2000.0 / rate; q.omega_max = kTwoPiRef * 2000.0 / rate;
            q.g1 = 1.0 - std::exp(-2.0 * omegaN * zeta / rate);
            q.g2 = -q.g1 + 2.0 * (1 - std::exp(-omegaN * zeta / rate) * std::cos(omegaN / rate * std::sqrt(1.0 - zeta * zeta)));
            q.mtauR = std::exp(-1.0 / (rate * tauR)); q.onem_mtauR = 1.0 - q.mtauR;
            q.mtauI = std::exp(-1.0 / (rate * tauI)); q.onem_mtauI = 1.0 - q.mtauI;
            am_prm.mtauR = q.mtauR; am_prm.onem_mtauR = q.onem_mtauR; am_prm.mtauI = q.mtauI; am_prm.onem_mtauI = q.onem_mtauI;
            std::vector<double> pw(2 * 2048);           // mtauR^(k + 1), mtauI^(k + 1): the carried averages' weights at sample k of a 2048-sample tile
            for (int k = 0; k < 2048; k++) { pw[(size_t)k] = std::pow(q.mtauR, (double)(k + 1)); pw[(size_t)(2048 + k)] = std::pow(q.mtauI, (double)(k + 1)); }
            // ... and the lanes' scan weights of the two averages (qh_wave.hpp PoleScan: pa = m^((lane & 15) + 1), pb = m^((lane & 31) + 1), pw = m^(lane + 1))
            pw.resize(2 * 2048 + 6 * 64);
            for (int f = 0; f < 2; f++) {
                const double m = f ? q.mtauI : q.mtauR;
                for (int l = 0; l < 64; l++) {
                    pw[(size_t)(2 * 2048 + (3 * f + 0) * 64 + l)] = std::pow(m, (double)((l & 15) + 1));
                    pw[(size_t)(2 * 2048 + (3 * f + 1) * 64 + l)] = std::pow(m, (double)((l & 31) + 1));
                    pw[(size_t)(2 * 2048 + (3 * f + 2) * 64 + l)] = std::pow(m, (double)(l + 1));
                }
            }
            QH_HIP(dev_alloc(&am_pw, pw.size()));
            QH_HIP(dev_alloc(&am_last, (size_t)2 * nch));
            QH_HIP(hipMemcpyAsync(am_pw, pw.data(), pw.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemsetAsync(am_last, 0, (size_t)2 * nch * sizeof(double), stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        // calc_fmd (wdsp/fmd.c:29-44) with create_rxa's constants (RXA.c:199-204)
        {
            const double zeta = 1.0, omegaN = 20000.0, tau = 0.02;
            PllParam &q = fm_pll_prm;
            q.omega_min = kTwoPiRef * -8000.0 / rate; q.omega_max = kTwoPiRef * 8000.0 / rate;
            q.g1 = 1.0 - std::exp(-2.0 * omegaN * zeta / rate);
            q.g2 = -q.g1 + 2.0 * (1 - std::exp(-omegaN * zeta / rate) * std::cos(omegaN / rate * std::sqrt(1.0 - zeta * zeta)));
            q.mtau = std::exp(-1.0 / (rate * tau)); q.onem_mtau = 1.0 - q.mtau;
            std::vector<double> pw(2048);               // mtau^(k + 1): the carried dc's weight at sample k of a tile (fm_audio_at)
            for (int k = 0; k < 2048; k++) pw[(size_t)k] = std::pow(q.mtau, (double)(k + 1));
            QH_HIP(dev_alloc(&fm_pw, (size_t)2048));
            QH_HIP(hipMemcpyAsync(fm_pw, pw.data(), 2048 * sizeof(double), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
        }
        demod_alloc = true;
        lists_dirty = true;
        for (ChanCfg &c : cfg) c.demod_dirty = true;
    }
    // a channel whose attack window moves after its AGC has run keeps the reference's ring_max, which may then be a value the window
    // no longer holds (wcpAGC.c:197-210 only rescans when the sample that leaves equals it): the time tiles take the window's maximum,
    // so that channel stays on the kernel that steps the reference's bookkeeping -- it alone: the lists put such channels last
    for (ChanCfg &c : cfg) {
        if (!c.agc_dirty || !c.agc_ran) continue;
        const int abuf = (int)std::ceil(rate * 4.0 * c.agc_tau_attack);
        if (c.agc_abuf != abuf) { c.agc_rewindow = true; if (!c.agc_stale) { c.agc_stale = true; lists_dirty = true; } }
    }
    if (lists_dirty) {
        std::vector<int> la, ls, lf, lb, lp, lgc, lgo, lgc_s, lgo_s, ll, lms_l[2][3], lbp[2], lfix[2], lsq, lem[3], lsn[2], lsnba, lrest, lusb, lrb;
        int n_sam0_new = 0;
        for (int ch = 0; ch < nch; ch++) {
            const ChanCfg &c = cfg[(size_t)ch];
            if (c.amsq_run) lsq.push_back(ch);
            if (c.snba_run) lsnba.push_back(ch);
            if (c.snb_pos() >= 0) lsn[c.snb_pos()].push_back(ch);
            if (c.emnr_run) lem[c.emnr_pos ? 1 + ((c.bp1_run && !c.bp1_pos) ? 1 : 0) : 0].push_back(ch);
            const int at_agc = (c.bp1_run && !c.bp1_pos) ? 1 : 0;       // the buffer the channel is in when xwcpagc runs
            for (int f = 0; f < 2; f++) if (c.lms[f].run) lms_l[f][c.lms[f].position ? 1 + at_agc : 0].push_back(ch);
            if (c.bp1_run) lbp[c.bp1_pos ? 1 : 0].push_back(ch);
            if (c.fix_before()) lfix[at_agc].push_back(ch);
            if (c.fmd_run && c.lim_run) ll.push_back(ch);
            if (c.amd_run && c.amd_mode == 0) la.push_back(ch);
            if (c.amd_run && c.amd_mode == 1) { if (c.sbmode == 0) ls.insert(ls.begin() + n_sam0_new++, ch); else ls.push_back(ch); }
            if (c.fmd_run) lf.push_back(ch); else { lrest.push_back(ch); (c.bp1_run ? lrb : lusb).push_back(ch); }
            if (c.bp1_run) lb.push_back(ch); else lp.push_back(ch);
            // xwcpagc sits between the two bp1 positions (RXA.c:581-586): a position-1 channel is still in `cur` there
            if (c.agc_run && c.agc_mode != 0) (c.bp1_run && !c.bp1_pos ? (c.agc_stale ? lgo_s : lgo) : (c.agc_stale ? lgc_s : lgc)).push_back(ch);
        }
        bool any_lms = false;
        for (int f = 0; f < 2; f++) for (int k = 0; k < 3; k++) { n_lms[f][k] = (int)lms_l[f][k].size(); any_lms = any_lms || n_lms[f][k]; }
        n_bp1p[0] = (int)lbp[0].size(); n_bp1p[1] = (int)lbp[1].size();
        n_fix[0] = (int)lfix[0].size(); n_fix[1] = (int)lfix[1].size();
        n_amsq = (int)lsq.size();
        for (int k = 0; k < 3; k++) n_emnr[k] = (int)lem[k].size();
        if ((n_emnr[0] || n_emnr[1] || n_emnr[2]) && !emnr_state) if (int rc = emnr_alloc()) return rc;
        n_snb[0] = (int)lsn[0].size(); n_snb[1] = (int)lsn[1].size(); n_snba = (int)lsnba.size();
        if (n_snba && !snba_state) if (int rc = snba_alloc()) return rc;
        if (snba_state) {
            // bpsnba's fircore keeps its delay line while it does not run: a channel that (re)joins the list finds its rows in
            // the ping-pong half that was current when it left
            for (int ch = 0; ch < nch; ch++) if (snb_listed[(size_t)ch]) cfg[(size_t)ch].snb_hist_at = cur_snb;
            std::fill(snb_listed.begin(), snb_listed.end(), 0);
            for (int ps = 0; ps < 2; ps++)
                for (int ch : lsn[ps]) {
                    ChanCfg &c = cfg[(size_t)ch];
                    if (c.snb_hist_at != cur_snb) {
                        QH_HIP(hipMemcpyAsync(hist_snb[cur_snb] + (size_t)ch * kHistBand, hist_snb[c.snb_hist_at] + (size_t)ch * kHistBand,
                                              kHistBand * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                        if (lhist[4][0] && lhist[4][1])         // the partitioned form's 16383-sample delay line goes along (nc > 4096)
                            QH_HIP(hipMemcpyAsync(lhist[4][cur_snb] + (size_t)ch * kLongHist, lhist[4][c.snb_hist_at] + (size_t)ch * kLongHist,
                                                  kLongHist * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                    }
                    c.snb_hist_at = cur_snb;
                    snb_listed[(size_t)ch] = 1;
                }
        }
        {
            // bp1's fircore keeps its delay line while it does not run, and SetRXABandpassRun (bandpass.c:385-390) switches it on without
            // RXAbp1Set's flush (RXA.c:825): the rows are in the half that was current when the channel left the list
            if (bp1_listed.size() != (size_t)nch) bp1_listed.assign((size_t)nch, 0);
            for (int ch = 0; ch < nch; ch++) if (bp1_listed[(size_t)ch]) cfg[(size_t)ch].bp1_hist_at = cur_bp1;
            std::fill(bp1_listed.begin(), bp1_listed.end(), 0);
            for (int ch : lb) {
                ChanCfg &c = cfg[(size_t)ch];
                if (c.bp1_hist_at != cur_bp1) {
                    QH_HIP(hipMemcpyAsync(hist_bp1[cur_bp1] + (size_t)ch * kHistBand, hist_bp1[c.bp1_hist_at] + (size_t)ch * kHistBand,
                                          kHistBand * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                    if (lhist[1][0] && lhist[1][1])
                        QH_HIP(hipMemcpyAsync(lhist[1][cur_bp1] + (size_t)ch * kLongHist, lhist[1][c.bp1_hist_at] + (size_t)ch * kLongHist,
                                              kLongHist * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                }
                c.bp1_hist_at = cur_bp1;
                bp1_listed[(size_t)ch] = 1;
            }
        }
        {
            // the FM de-emphasis / audio fircores keep their delay lines while the channel is in another mode (SetRXAMode only
            // clears fmd's run flag, RXA.c:758-776): the ping-pong pair flips for the listed channels only, so a channel that
            // comes back finds its rows in the half that was current when it left
            if (fm_listed.size() != (size_t)nch) fm_listed.assign((size_t)nch, 0);
            for (int ch = 0; ch < nch; ch++) if (fm_listed[(size_t)ch]) cfg[(size_t)ch].fm_hist_at = cur_de;
            std::fill(fm_listed.begin(), fm_listed.end(), 0);
            for (int ch : lf) {
                ChanCfg &c = cfg[(size_t)ch];
                if (c.fm_hist_at != cur_de) {
                    QH_HIP(hipMemcpyAsync(hist_de[cur_de] + (size_t)ch * kHistBand, hist_de[c.fm_hist_at] + (size_t)ch * kHistBand,
                                          kHistBand * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                    QH_HIP(hipMemcpyAsync(hist_aud[cur_aud] + (size_t)ch * kHistBand, hist_aud[c.fm_hist_at] + (size_t)ch * kHistBand,
                                          kHistBand * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                    if (lhist[2][0] && lhist[2][1])             // ... and the partitioned forms' long delay lines (nc > 4096)
                        QH_HIP(hipMemcpyAsync(lhist[2][cur_de] + (size_t)ch * kLongHist, lhist[2][c.fm_hist_at] + (size_t)ch * kLongHist,
                                              kLongHist * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                    if (lhist[3][0] && lhist[3][1])
                        QH_HIP(hipMemcpyAsync(lhist[3][cur_aud] + (size_t)ch * kLongHist, lhist[3][c.fm_hist_at] + (size_t)ch * kLongHist,
                                              kLongHist * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                }
                c.fm_hist_at = cur_de;
                fm_listed[(size_t)ch] = 1;
            }
        }
        if (n_amsq && !amsq_prm) {
            QH_HIP(dev_alloc(&amsq_prm, (size_t)nch));
            QH_HIP(dev_alloc(&amsq_state, (size_t)nch));
            QH_HIP(hipMemsetAsync(amsq_state, 0, (size_t)nch * sizeof(AmsqState), stream));
            // compute_slews, amsq.c:28-46, with muted_gain 0 and 70 ms up / down (RXA.c:166-167,172): theta accumulates as there
            amsq_ntup = (int)(0.070 * rate); amsq_ntdown = (int)(0.070 * rate);
            std::vector<double> up((size_t)amsq_ntup + 1), down((size_t)amsq_ntdown + 1);
            double delta = kPiRef / (double)amsq_ntup, theta = 0.0;
            for (int i = 0; i <= amsq_ntup; i++) { up[(size_t)i] = 0.0 + (1.0 - 0.0) * 0.5 * (1.0 - std::cos(theta)); theta += delta; }
            delta = kPiRef / (double)amsq_ntdown; theta = 0.0;
            for (int i = 0; i <= amsq_ntdown; i++) { down[(size_t)i] = 0.0 + (1.0 - 0.0) * 0.5 * (1.0 + std::cos(theta)); theta += delta; }
            QH_HIP(dev_alloc(&amsq_cup, up.size()));
            QH_HIP(dev_alloc(&amsq_cdown, down.size()));
            QH_HIP(hipMemcpyAsync(amsq_cup, up.data(), up.size() * 8, hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemcpyAsync(amsq_cdown, down.data(), down.size() * 8, hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            dev_bytes += (long long)nch * (sizeof(AmsqParam) + sizeof(AmsqState)) + (long long)(up.size() + down.size()) * 8;
            for (ChanCfg &c : cfg) c.amsq_dirty = true;
        }
        if (any_lms && !lms_prm[0]) {
            for (int f = 0; f < 2; f++) {
                QH_HIP(dev_alloc(&lms_prm[f], (size_t)nch));
                QH_HIP(dev_alloc(&lms_state[f], (size_t)nch));
                // create_anf: lidx 1.0, ngamma 6.25e-12; create_anr: lidx 120.0, ngamma 0.001 (RXA.c:289-292,309-312)
                std::vector<LmsState> init((size_t)nch);
                std::memset(init.data(), 0, init.size() * sizeof(LmsState));
                for (LmsState &st : init) { st.lidx = f ? 120.0 : 1.0; st.ngamma = f ? 0.001 : 6.25e-12; }
                QH_HIP(hipMemcpyAsync(lms_state[f], init.data(), init.size() * sizeof(LmsState), hipMemcpyHostToDevice, stream));
                QH_HIP(hipStreamSynchronize(stream));
                dev_bytes += (long long)nch * (sizeof(LmsParam) + sizeof(LmsState));
            }
            for (ChanCfg &c : cfg) { c.lms[0].dirty = c.lms[1].dirty = true; c.lms[0].flush = c.lms[1].flush = false; }
        }
        n_sam0 = n_sam0_new;
        n_am = (int)la.size(); n_sam = (int)ls.size(); n_fm = (int)lf.size(); n_bp1 = (int)lb.size(); n_plain = (int)lp.size();
        n_agc_cur_stale = (int)lgc_s.size(); n_agc_other_stale = (int)lgo_s.size();
        lgc.insert(lgc.end(), lgc_s.begin(), lgc_s.end()); lgo.insert(lgo.end(), lgo_s.begin(), lgo_s.end());
        n_agc_cur = (int)lgc.size(); n_agc_other = (int)lgo.size();
        n_lim = (int)ll.size();
        if (n_lim) {
            if (!list_lim) {
                QH_HIP(dev_alloc(&list_lim, (size_t)nch));
                QH_HIP(dev_alloc(&lim_prm, (size_t)nch));
                QH_HIP(dev_alloc(&lim_state, (size_t)nch));
                dev_bytes += (long long)nch * (sizeof(AgcParam) + sizeof(AgcState) + sizeof(int));
                for (ChanCfg &c : cfg) c.lim_dirty = true;
            }
            QH_HIP(hipMemcpyAsync(list_lim, ll.data(), ll.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        }
        auto put = [&](int *dst, const std::vector<int> &v) -> hipError_t {
            return v.empty() ? hipSuccess : hipMemcpyAsync(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, stream);
        };
        QH_HIP(put(list_am, la)); QH_HIP(put(list_sam, ls)); QH_HIP(put(list_fm, lf)); QH_HIP(put(list_bp1, lb)); QH_HIP(put(list_plain, lp));
        QH_HIP(put(list_agc_cur, lgc)); QH_HIP(put(list_agc_other, lgo)); QH_HIP(put(list_rest, lrest)); QH_HIP(put(list_usb, lusb)); QH_HIP(put(list_rb, lrb));
        n_rest = (int)lrest.size(); n_usb = (int)lusb.size(); n_rb = (int)lrb.size();
        {
            // partners: neighbours in the list, ordered so that equal designs are neighbours; a channel left over is its own partner
            auto bp1_real = [&](int ch) { const ChanCfg &c = cfg[(size_t)ch]; return c.bp1_run && c.bp1_flow == -c.bp1_fhigh && !c.mp; };
            auto bp1_same = [&](int x, int y) {
                const ChanCfg &p = cfg[(size_t)x], &q = cfg[(size_t)y];
                return p.bp1_nc == q.bp1_nc && p.bp1_wintype == q.bp1_wintype && p.bp1_fhigh == q.bp1_fhigh && p.bp1_gain == q.bp1_gain;
            };
            auto bp1_pairs = [&](std::vector<int> v) {
                std::vector<int> pr;
                for (int ch : v) if (!bp1_real(ch)) return pr;
                std::stable_sort(v.begin(), v.end(), [&](int x, int y) {
                    const ChanCfg &p = cfg[(size_t)x], &q = cfg[(size_t)y];
                    if (p.bp1_fhigh != q.bp1_fhigh) return p.bp1_fhigh < q.bp1_fhigh;
                    if (p.bp1_nc != q.bp1_nc) return p.bp1_nc < q.bp1_nc;
                    if (p.bp1_wintype != q.bp1_wintype) return p.bp1_wintype < q.bp1_wintype;
                    return p.bp1_gain < q.bp1_gain;
                });
                for (size_t i = 0; i < v.size();) {
                    if (i + 1 < v.size() && bp1_same(v[i], v[i + 1])) { pr.push_back(v[i]); pr.push_back(v[i + 1]); i += 2; }
                    else { pr.push_back(v[i]); pr.push_back(v[i]); i += 1; }
                }
                return pr;
            };
            std::vector<int> pf, pa = bp1_pairs(la), ps = bp1_pairs(ls);
            for (size_t i = 0; i < lf.size(); i += 2) { pf.push_back(lf[i]); pf.push_back(i + 1 < lf.size() ? lf[i + 1] : lf[i]); }
            QH_HIP(put(pairs_fm, pf)); QH_HIP(put(pairs_am, pa)); QH_HIP(put(pairs_sam, ps));
            np_fm = (int)pf.size() / 2; np_am = (int)pa.size() / 2; np_sam = (int)ps.size() / 2;
        }
        for (int f = 0; f < 2; f++) for (int k = 0; k < 3; k++) QH_HIP(put(list_lms[f][k], lms_l[f][k]));
        QH_HIP(put(list_bp1p[0], lbp[0])); QH_HIP(put(list_bp1p[1], lbp[1]));
        QH_HIP(put(list_fix[0], lfix[0])); QH_HIP(put(list_fix[1], lfix[1])); QH_HIP(put(list_amsq, lsq));
        for (int k = 0; k < 3; k++) QH_HIP(put(list_emnr[k], lem[k]));
        QH_HIP(put(list_snb[0], lsn[0])); QH_HIP(put(list_snb[1], lsn[1])); QH_HIP(put(list_snba, lsnba));
        std::vector<double> fg((size_t)nch);
        for (int ch = 0; ch < nch; ch++) fg[(size_t)ch] = cfg[(size_t)ch].agc_fixed;
        QH_HIP(hipMemcpyAsync(fix_gain, fg.data(), fg.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        QH_HIP(hipStreamSynchronize(stream));
        lists_dirty = false;
    }
    int want_nc = 0, want_mp = -1;
    for (int ch = 0; ch < nch; ch++) {
        ChanCfg &c = cfg[(size_t)ch];
        if (c.fmd_run) {
            if (want_nc && want_nc != c.fm_nc) return set_error(QH_ERR_UNSUPPORTED, "FM channels with different nc in one engine");
            want_nc = c.fm_nc;
            // RXASetMP reaches the FM filters of ITS channel (SetRXAFMMPde / MPaud, wdsp/RXA.c:956-957): the one design the engine's FM
            // channels share follows them, not whichever channel was set last
            if (want_mp >= 0 && want_mp != c.mp) return set_error(QH_ERR_UNSUPPORTED, "FM channels with different RXASetMP in one engine");
            want_mp = c.mp;
        }
        if (c.agc_dirty) {
            // loadWcpAGC, wdsp/wcpAGC.c:115-146, with create_rxa's constants (RXA.c:335-358)
            const double n_tau = 4.0, max_input = 1.0, out_targ = 1.0, tau_fast_back = 0.250, tau_fast_decay = 0.005;
            const double tau_hang_backmult = 0.500, tau_hang_decay = 0.100;
            AgcParam q{};
            q.attack_buffsize = (int)std::ceil(rate * n_tau * c.agc_tau_attack);
            if (q.attack_buffsize + 2 > kAgcRing)
                return set_error(QH_ERR_UNSUPPORTED, "AGC attack of %g s needs a look-ahead of %d samples (limit %d)", c.agc_tau_attack,
                                 q.attack_buffsize, kAgcRing - 2);
            c.agc_abuf = q.attack_buffsize;
            q.attack_mult = 1.0 - std::exp(-1.0 / (rate * c.agc_tau_attack));
            q.decay_mult = 1.0 - std::exp(-1.0 / (rate * c.agc_tau_decay));
            q.fast_decay_mult = 1.0 - std::exp(-1.0 / (rate * tau_fast_decay));
            q.fast_backmult = 1.0 - std::exp(-1.0 / (rate * tau_fast_back));
            q.onemfast_backmult = 1.0 - q.fast_backmult;
            q.out_target = out_targ * (1.0 - std::exp(-n_tau)) * 0.9999;
            q.min_volts = q.out_target / (c.agc_var_gain * c.agc_max_gain);
            q.inv_out_target = 1.0 / q.out_target;
            double tmp = std::log10(q.out_target / (max_input * c.agc_var_gain * c.agc_max_gain));
            if (tmp == 0.0) tmp = 1e-16;
            q.slope_constant = (q.out_target * (1.0 - 1.0 / c.agc_var_gain)) / tmp;
            q.inv_max_input = 1.0 / max_input;
            tmp = std::pow(10.0, (c.agc_hang_thresh - 1.0) / 0.125);
            q.hang_level = (max_input * tmp + (q.out_target / (c.agc_var_gain * c.agc_max_gain)) * (1.0 - tmp)) * 0.637;
            q.hang_backmult = 1.0 - std::exp(-1.0 / (rate * tau_hang_backmult));
            q.onemhang_backmult = 1.0 - q.hang_backmult;
            q.hang_decay_mult = 1.0 - std::exp(-1.0 / (rate * tau_hang_decay));
            q.pop_ratio = 5.0;
            q.hang_count_init = (int)(c.agc_hangtime * rate);
            q.hang_enable = 1;
            q.pmode = 1;
            QH_HIP(hipMemcpyAsync(agc_prm + ch, &q, sizeof(q), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.agc_dirty = false;
        }
        if (emnr_chan && c.emnr_dirty) {
            if (c.emnr_run && (c.emnr_npe < 0 || c.emnr_npe > 2 || c.emnr_gain_method < 0 || c.emnr_gain_method > 3))
                return set_error(QH_ERR_UNSUPPORTED, "EMNR: gain methods 0..3 and noise estimators 0..2");
            const EmnrChan ec{ c.emnr_gain_method, c.emnr_npe, c.emnr_ae, 0, c.emnr_ae_zeta, c.emnr_ae_psi, c.emnr_train_zeta, c.emnr_train_t2 };
            QH_HIP(hipMemcpyAsync(emnr_chan + ch, &ec, sizeof(ec), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.emnr_dirty = false;
        }
        if (emnr_state && c.emnr_flush) {           // flush_emnr, emnr.c:583-596: the accumulators and their indices, not the estimators
            EmnrScalars sc;
            QH_HIP(hipMemcpyAsync(&sc, emnr_scal + ch, sizeof(sc), hipMemcpyDeviceToHost, stream));
            QH_HIP(hipStreamSynchronize(stream));
            sc.iainidx = sc.iaoutidx = sc.oaoutidx = sc.nsamps = sc.saveidx = 0; sc.oainidx = emnr_prm.init_oainidx;
            QH_HIP(hipMemcpyAsync(emnr_scal + ch, &sc, sizeof(sc), hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemsetAsync(emnr_state + (size_t)ch * kEmnrStateDoubles, 0, (size_t)EO_PREVG * sizeof(double), stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.emnr_flush = false;
        }
        if (snba_state && c.snba_taps_dirty) {
            // calc_resample for outresamp (12 kHz -> dsp_rate, gain 2, resample.c:35-79) with the channel's output bandwidth
            const SnbaParam &q = snba_prm;
            if (q.ratio > 1) {
                const int L = q.ratio, ncoef = q.cpp_out * L;
                const double full = (double)(12000 * L), fc = c.snba_f_high == 0.0 ? 0.45 * 12000.0 : c.snba_f_high;
                const double lo = c.snba_f_low < 0.0 ? -fc / full : c.snba_f_low / full;
                const std::vector<cd> imp = fir_bandpass(ncoef, lo, fc / full, 1.0, 1, 0, 2.0 * (double)L);
                std::vector<double> hp((size_t)ncoef);
                size_t i = 0;
                for (int j = 0; j < L; j++) for (int k = 0; k < ncoef; k += L) hp[i++] = imp[(size_t)(j + k)].real();
                QH_HIP(hipMemcpyAsync(snba_hout + (size_t)ch * ncoef, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, stream));
                QH_HIP(hipStreamSynchronize(stream));
            }
            c.snba_taps_dirty = false;
        }
        if (snba_state && (c.snba_flush || c.snba_rout_flush)) {
            const SnbaParam &q = snba_prm;
            double *st = snba_state + (size_t)ch * q.state_doubles;
            if (q.cpp_out > 1) QH_HIP(hipMemsetAsync(st + q.off_rout, 0, (size_t)(q.cpp_out - 1) * sizeof(double), stream));
            if (c.snba_flush) {         // flush_snba, snb.c:161-185: the frame half of xbase, the accumulators, both resamplers
                QH_HIP(hipMemsetAsync(st + kSnbX, 0, (size_t)kSnbX * sizeof(double), stream));
                QH_HIP(hipMemsetAsync(st + q.off_inacc, 0, (size_t)(q.state_doubles - q.off_inacc) * sizeof(double), stream));
                const SnbaIdx ix{ 0, 0, 0, 0, q.init_oaoutidx, { 0, 0, 0 } };
                QH_HIP(hipMemcpyAsync(snba_idx + ch, &ix, sizeof(ix), hipMemcpyHostToDevice, stream));
                QH_HIP(hipStreamSynchronize(stream));
            }
        }
        c.snba_flush = c.snba_rout_flush = false;
        if (amsq_prm && c.amsq_dirty) {
            // calc_amsq, amsq.c:48-64: 10 ms average (RXA.c:165)
            AmsqParam q{};
            q.avm = std::exp(-1.0 / (rate * 0.010)); q.onem_avm = 1.0 - q.avm;
            q.tail_thresh = c.amsq_tail_thresh; q.unmute_thresh = c.amsq_unmute_thresh; q.min_tail = 0.0; q.max_tail = c.amsq_max_tail;
            q.muted_gain = 0.0; q.rate = rate; q.ntup = amsq_ntup; q.ntdown = amsq_ntdown;
            QH_HIP(hipMemcpyAsync(amsq_prm + ch, &q, sizeof(q), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.amsq_dirty = false;
        }
        for (int f = 0; f < 2 && lms_prm[0]; f++) {
            ChanCfg::Lms &m = c.lms[f];
            if (m.dirty) {
                if (m.run && (m.taps < 1 || m.taps > 64 || m.delay < 1 || m.delay > 64))
                    return set_error(QH_ERR_UNSUPPORTED, "%s: taps %d / delay %d (1..64 each: one tap per lane)", f ? "ANR" : "ANF", m.taps, m.delay);
                // lidx_min, lidx_max, den_mult, lincr, ldecr of create_rxa (RXA.c:290-295,310-315)
                const LmsParam q{ m.taps, m.delay, f, 0, m.two_mu, m.gamma, f ? 120.0 : 0.0, 200.0, 6.25e-10, 1.0, 3.0 };
                QH_HIP(hipMemcpyAsync(lms_prm[f] + ch, &q, sizeof(q), hipMemcpyHostToDevice, stream));
                QH_HIP(hipStreamSynchronize(stream));
                m.dirty = false;
            }
            if (m.flush) {          // flush_anf (anf.c:135-140): delay line and weights; lidx / ngamma carry on
                QH_HIP(hipMemsetAsync(lms_state[f] + ch, 0, offsetof(LmsState, lidx), stream));
                m.flush = false;
            }
        }
        if (c.lim_dirty && lim_prm) {
            // calc_fmd's create_wcpagc(1, 5, 1, ..., 0.001, 0.008, 4, lim_gain, 1.0, 1.0, 1.0, 0.9, 0.250, 0.004, 4.0, 0,
            // 0.500, 0.500, 2.000, 0.100) (fmd.c:48-72) through loadWcpAGC (wcpAGC.c:115-146); a new limiter starts cleared
            const double tau_attack = 0.001, tau_decay = 0.008, n_tau = 4.0, max_gain = c.lim_gain, var_gain = 1.0, max_input = 1.0,
                         out_targ = 0.9, tau_fast_back = 0.250, tau_fast_decay = 0.004, tau_hang_backmult = 0.500, hangtime = 0.500,
                         hang_thresh = 2.000, tau_hang_decay = 0.100;
            AgcParam q{};
            q.attack_buffsize = (int)std::ceil(rate * n_tau * tau_attack);
            q.attack_mult = 1.0 - std::exp(-1.0 / (rate * tau_attack));
            q.decay_mult = 1.0 - std::exp(-1.0 / (rate * tau_decay));
            q.fast_decay_mult = 1.0 - std::exp(-1.0 / (rate * tau_fast_decay));
            q.fast_backmult = 1.0 - std::exp(-1.0 / (rate * tau_fast_back));
            q.onemfast_backmult = 1.0 - q.fast_backmult;
            q.out_target = out_targ * (1.0 - std::exp(-n_tau)) * 0.9999;
            q.min_volts = q.out_target / (var_gain * max_gain);
            q.inv_out_target = 1.0 / q.out_target;
            double tmp = std::log10(q.out_target / (max_input * var_gain * max_gain));
            if (tmp == 0.0) tmp = 1e-16;
            q.slope_constant = (q.out_target * (1.0 - 1.0 / var_gain)) / tmp;
            q.inv_max_input = 1.0 / max_input;
            tmp = std::pow(10.0, (hang_thresh - 1.0) / 0.125);
            q.hang_level = (max_input * tmp + (q.out_target / (var_gain * max_gain)) * (1.0 - tmp)) * 0.637;
            q.hang_backmult = 1.0 - std::exp(-1.0 / (rate * tau_hang_backmult));
            q.onemhang_backmult = 1.0 - q.hang_backmult;
            q.hang_decay_mult = 1.0 - std::exp(-1.0 / (rate * tau_hang_decay));
            q.pop_ratio = 4.0;
            q.hang_count_init = (int)(hangtime * rate);
            q.hang_enable = 0;
            q.pmode = 1;
            QH_HIP(hipMemcpyAsync(lim_prm + ch, &q, sizeof(q), hipMemcpyHostToDevice, stream));
            QH_HIP(hipMemsetAsync(lim_state + ch, 0, sizeof(AgcState), stream));
            const int oi = kAgcRing - 1;                            // out_index = -1 (calc_wcpagc, wcpAGC.c:34)
            QH_HIP(hipMemcpyAsync(&lim_state[ch].out_index, &oi, sizeof(int), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            c.lim_dirty = false;
        }
        if (!c.demod_dirty) continue;
        const int lf = c.levelfade;
        const double again = rate / (c.fm_dev * kTwoPiRef);                 // wdsp/fmd.c:44
        SamChanParam sp{ c.sbmode, c.levelfade };
        SnotchParam sn{};
        {   // calc_snotch, wdsp/iir.c:35-49 (bw 0.0002, fmd.c:47)
            const double fn = c.ctcss_freq / (double)dsp_rate, csn = std::cos(kTwoPiRef * fn), qr = 1.0 - 3.0 * 0.0002;
            const double qk = (1.0 - 2.0 * qr * csn + qr * qr) / (2.0 * (1.0 - csn));
            sn.a0 = qk; sn.a1 = -2.0 * qk * csn; sn.a2 = qk; sn.b1 = 2.0 * qr * csn; sn.b2 = -qr * qr; sn.run = c.ctcss_run;
        }
        QH_HIP(hipMemcpyAsync(levelfade + ch, &lf, sizeof(int), hipMemcpyHostToDevice, stream));
        QH_HIP(hipMemcpyAsync(fm_again + ch, &again, sizeof(double), hipMemcpyHostToDevice, stream));
        QH_HIP(hipMemcpyAsync(sam_prm + ch, &sp, sizeof(sp), hipMemcpyHostToDevice, stream));
        QH_HIP(hipMemcpyAsync(sn_prm + ch, &sn, sizeof(sn), hipMemcpyHostToDevice, stream));
        if (c.ctcss_flush) {                    // calc_snotch ends with flush_snotch, wdsp/iir.c:48
            QH_HIP(hipMemsetAsync(sn_state + ch, 0, sizeof(SnotchState), stream));
            c.ctcss_flush = false;
        }
        QH_HIP(hipStreamSynchronize(stream));
        c.demod_dirty = false;
    }
    if (want_mp >= 0) fm_mp = want_mp;
    if (want_nc && (want_nc != fm_nc_built || fm_mp != fm_mp_built || fm_nfft_built != 2 * bnfft + (band2g ? 1 : 0))) {
        // create_fmd, wdsp/fmd.c:108-116: de-emphasis by frequency sampling, audio band-pass 0.8*f_low .. 1.1*f_high
        const double f_low = 300.0, f_high = 3000.0, afgain = 0.5;
        std::vector<cd> de = fc_impulse(want_nc, f_low, f_high, +20.0 * std::log10(f_high / f_low), 0.0, 1, rate,
                                        1.0 / (2.0 * dsp_size), 0, 0);
        std::vector<cd> au = fir_bandpass(want_nc, 0.8 * f_low, 1.1 * f_high, rate, 0, 1, afgain / (2.0 * dsp_size));
        if (fm_mp) { de = mp_imp(de, 16, 0); au = mp_imp(au, 16, 0); }     // SetRXAFMMPde / MPaud, wdsp/RXA.c:956-957
        fm_mp_built = fm_mp;
        for (auto &v : de) v *= (double)(2 * dsp_size);
        for (auto &v : au) v *= (double)(2 * dsp_size);
        de_real = true;
        for (const cd &v : de) de_real = de_real && v.imag() == 0.0;
        if (long_parts[2] > 1) {
            if (int rc = long_masks_upload(2, 0, de)) return rc;
            if (int rc = long_masks_upload(3, 0, au)) return rc;
            de.resize((size_t)kLongPart); au.resize((size_t)kLongPart);      // (the one-tile masks are not used then)
        }
        if (int rc = upload(mask_de, band_mask(de), stream)) return rc;
        if (int rc = upload(mask_aud, band_mask(au), stream)) return rc;
        fm_nfft_built = 2 * bnfft + (band2g ? 1 : 0);
        if (fm_nc_built && fm_nc_built != want_nc) {      // setNc_fircore zeroes the delay lines, wdsp/firmin.c:454-466
            for (int i = 0; i < 2; i++) {
                if (lhist[2][i]) QH_HIP(hipMemsetAsync(lhist[2][i], 0, (size_t)nch * kLongHist * sizeof(double2), stream));
                if (lhist[3][i]) QH_HIP(hipMemsetAsync(lhist[3][i], 0, (size_t)nch * kLongHist * sizeof(double2), stream));
                QH_HIP(hipMemsetAsync(hist_de[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
                QH_HIP(hipMemsetAsync(hist_aud[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
            }
        }
        fm_nc_built = want_nc;
    }
    return QH_OK;
}

// the frame advance and the two accumulators' sizes for an overlap (calc_snba, snb.c:45-65)
void Engine::snba_plan(int ovrlp)
{
    SnbaParam &q = snba_prm;
    q.incr = kSnbX / ovrlp;
    q.iasize = q.incr > q.isize ? q.incr : q.isize;
    q.oasize = q.iasize;
    q.init_oaoutidx = q.incr > q.isize ? q.isize : 0;
    q.off_inacc = 2 * kSnbX; q.off_outacc = q.off_inacc + q.iasize; q.off_rin = q.off_outacc + q.oasize;
    q.off_rout = q.off_rin + (q.cpp_in - 1); q.state_doubles = q.off_rout + (q.cpp_out - 1);
}

// SetRXASNBAovrlp (snb.c:595-603): decalc_snba + calc_snba with the new overlap -- the frame memory (xbase, made by create_snba) stays,
// the accumulators, their indices and both resamplers start over
int Engine::snba_set_ovrlp(int ovrlp)
{
    if (ovrlp < 1 || ovrlp > kSnbX || kSnbX / ovrlp < 1) return set_error(QH_ERR_INVALID, "SetRXASNBAovrlp: 1 .. %d", kSnbX);
    snba_ovrlp = ovrlp;
    if (!snba_state) return QH_OK;                       // nothing built yet: snba_alloc plans with it
    QH_HIP(hipSetDevice(device));
    QH_HIP(hipStreamSynchronize(stream));
    const SnbaParam old = snba_prm;
    std::vector<double> frames((size_t)nch * 2 * kSnbX);
    QH_HIP(hipMemcpy2D(frames.data(), 2 * kSnbX * sizeof(double), snba_state, (size_t)old.state_doubles * sizeof(double), 2 * kSnbX * sizeof(double),
                       (size_t)nch, hipMemcpyDeviceToHost));
    snba_plan(ovrlp);
    const SnbaParam &q = snba_prm;
    (void)hipFree(snba_state); snba_state = nullptr;
    QH_HIP(dev_alloc(&snba_state, (size_t)nch * q.state_doubles));
    QH_HIP(qh::dev_zero(snba_state, (size_t)nch * q.state_doubles * sizeof(double)));
    QH_HIP(hipMemcpy2D(snba_state, (size_t)q.state_doubles * sizeof(double), frames.data(), 2 * kSnbX * sizeof(double), 2 * kSnbX * sizeof(double),
                       (size_t)nch, hipMemcpyHostToDevice));
    std::vector<SnbaIdx> ix((size_t)nch, SnbaIdx{ 0, 0, 0, 0, q.init_oaoutidx, { 0, 0, 0 } });
    QH_HIP(hipMemcpy(snba_idx, ix.data(), ix.size() * sizeof(SnbaIdx), hipMemcpyHostToDevice));
    drop_graphs(); epoch++;
    return QH_OK;
}

// calc_emnr (wdsp/emnr.c:240-497) with create_rxa's arguments (RXA.c:319-332): parameters, window, start values of every array
int Engine::snba_alloc()
{
    // calc_snba, snb.c:31-66, with create_rxa's arguments (RXA.c:237-255)
    SnbaParam &q = snba_prm;
    if (dsp_rate % 12000 || (dsp_rate / 12000 != 1 && dsp_rate / 12000 != 2 && dsp_rate / 12000 != 4) || dsp_size > kSnbMaxDsp ||
        dsp_size % (dsp_rate / 12000))
        return set_error(QH_ERR_UNSUPPORTED, "SNBA: dsp_rate 12000, 24000 or 48000 and dsp_size up to %d", kSnbMaxDsp);
    q.ratio = dsp_rate / 12000;
    q.isize = dsp_size / q.ratio;
    q.cpp_in = q.ratio > 1 ? 140 * q.ratio + 1 : 1;
    q.cpp_out = q.ratio > 1 ? 141 : 1;
    q.asize = 64; q.npasses = 2; q.b = 10; q.pre = 2; q.post = 2; q.k1 = 8.0; q.k2 = 20.0; q.pmultmin = 0.5;
    snba_plan(snba_ovrlp);
    QH_HIP(dev_alloc(&snba_state, (size_t)nch * q.state_doubles));
    QH_HIP(hipMemsetAsync(snba_state, 0, (size_t)nch * q.state_doubles * sizeof(double), stream));
    QH_HIP(dev_alloc(&snba_idx, (size_t)nch));
    std::vector<SnbaIdx> ix((size_t)nch, SnbaIdx{ 0, 0, 0, 0, q.init_oaoutidx, { 0, 0, 0 } });
    QH_HIP(hipMemcpyAsync(snba_idx, ix.data(), ix.size() * sizeof(SnbaIdx), hipMemcpyHostToDevice, stream));
    QH_HIP(dev_alloc(&snba_scratch, (size_t)nch * kSnbX * kSnbX));
    QH_HIP(dev_alloc(&snba_tune, (size_t)nch));
    snba_tune_dirty = true;
    QH_HIP(dev_alloc(&snba_hin, (size_t)q.cpp_in));
    QH_HIP(dev_alloc(&snba_hout, (size_t)nch * q.cpp_out * q.ratio));
    std::vector<double> hin((size_t)q.cpp_in, 1.0);
    if (q.ratio > 1) {      // inresamp: dsp_rate -> 12 kHz, 250 .. 5400 Hz, gain 2 (snb.c:43-44)
        const double full = (double)dsp_rate;
        const std::vector<cd> imp = fir_bandpass(q.cpp_in, 250.0 / full, 0.45 * 12000.0 / full, 1.0, 1, 0, 2.0);
        for (int i = 0; i < q.cpp_in; i++) hin[(size_t)i] = imp[(size_t)i].real();
    }
    QH_HIP(hipMemcpyAsync(snba_hin, hin.data(), hin.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    QH_HIP(dev_alloc(&mask_snb, (size_t)nch * kBandNfftMax));
    for (int i = 0; i < 2; i++) {
        QH_HIP(dev_alloc(&hist_snb[i], (size_t)nch * kHistBand));
        QH_HIP(hipMemsetAsync(hist_snb[i], 0, (size_t)nch * kHistBand * sizeof(double2), stream));
    }
    QH_HIP(hipStreamSynchronize(stream));
    dev_bytes += (long long)nch * ((long long)q.state_doubles * 8 + (long long)kSnbX * kSnbX * 8 + (long long)kBandNfftMax * 16 + 2LL * kHistBand * 16);
    snb_listed.assign((size_t)nch, 0);
    for (ChanCfg &c : cfg) { c.snba_taps_dirty = true; c.snb_dirty = true; c.snba_flush = c.snba_rout_flush = false; c.snb_flush = false; c.snb_hist_at = cur_snb; }
    return QH_OK;
}

int Engine::emnr_alloc()
{
    if (dsp_size > kEmnrIncr) return set_error(QH_ERR_UNSUPPORTED, "EMNR: dsp_size up to %d", kEmnrIncr);
    const double rate = (double)dsp_rate, incr = (double)kEmnrIncr;
    EmnrParam &q = emnr_prm;
    auto tc = [&](double base) { const double tau = -128.0 / 8000.0 / std::log(base); return std::exp(-incr / rate / tau); };
    q.gain = 1.0 / kEmnrF / 4.0;
    q.gf1p5 = std::sqrt(kPiRef) / 2.0;
    q.alpha = tc(0.985);
    q.eps_floor = 1.0e-300; q.gamma_max = 40.0; q.xi_min = std::pow(10.0, -40.0 / 10.0); q.q = 0.2; q.gmax = 10000.0;
    q.dim_zeta = 60;
    q.z_gamma_min = h_zrange[0]; q.z_gamma_max = h_zrange[1]; q.z_xihat_min = h_zrange[2]; q.z_xihat_max = h_zrange[3];
    q.alphaCsmooth = tc(0.7); q.alphaMax = tc(0.96); q.alphaCmin = tc(0.7); q.alphaMin_max_value = tc(0.3);
    q.snrq = -incr / (0.064 * rate);
    q.betamax = tc(0.8);
    q.invQeqMax = 0.5; q.av = 2.12;
    const double Dtime = 8.0 * 12.0 * 128.0 / 8000.0;
    q.U = 8;
    q.V = (int)(0.5 + (Dtime * rate / (q.U * incr)));
    if (q.V < 4) q.V = 4;
    if ((q.U = (int)(0.5 + (Dtime * rate / (q.V * incr)))) < 1) q.U = 1;
    if (q.U > kEmnrU) return set_error(QH_ERR_UNSUPPORTED, "EMNR: %d minimum sub-windows at this rate (up to %d)", q.U, kEmnrU);
    q.D = q.U * q.V;
    {
        static const double Dvals[18] = { 1.0, 2.0, 5.0, 8.0, 10.0, 15.0, 20.0, 30.0, 40.0, 60.0, 80.0, 120.0, 140.0, 160.0, 180.0, 220.0, 260.0, 300.0 };
        static const double Mvals[18] = { 0.000, 0.260, 0.480, 0.580, 0.610, 0.668, 0.705, 0.762, 0.800, 0.841, 0.865, 0.890, 0.900, 0.910,
                                          0.920, 0.930, 0.935, 0.940 };
        auto interpM = [&](double x) {              // emnr.c:185-202
            if (x <= Dvals[0]) return Mvals[0];
            if (x >= Dvals[17]) return Mvals[17];
            int idx = 0;
            while (x >= Dvals[idx]) idx++;
            const double xllow = std::log10(Dvals[idx - 1]), xlhigh = std::log10(Dvals[idx]);
            const double frac = (std::log10(x) - xllow) / (xlhigh - xllow);
            return Mvals[idx - 1] + frac * (Mvals[idx] - Mvals[idx - 1]);
        };
        q.MofD = interpM((double)q.D); q.MofV = interpM((double)q.V);
    }
    q.invQbar_points[0] = 0.03; q.invQbar_points[1] = 0.05; q.invQbar_points[2] = 0.06; q.invQbar_points[3] = 1.0e300;
    {
        const double f[4] = { 8.0, 4.0, 2.0, 1.2 };
        for (int i = 0; i < 4; i++) {
            const double db = 10.0 * std::log10(f[i]) / (12.0 * 128 / 8000);
            q.nsmax[i] = std::pow(10.0, db / 10.0 * q.V * incr / rate);
        }
    }
    q.alpha_pow = tc(0.8); q.alpha_Pbar = tc(0.9);
    q.epsH1 = std::pow(10.0, 15.0 / 10.0); q.epsH1r = q.epsH1 / (1.0 + q.epsH1);
    {   // npl, emnr.c:458-489
        auto tl = [&](double base) { const double tau = -256.0 / (20100.0 * std::log(base)); return std::exp(-incr / (rate * tau)); };
        q.l_eta = tl(0.7); q.l_gamma = tl(0.998); q.l_beta = tl(0.8); q.l_alpha_d = tl(0.85); q.l_alpha_p = tl(0.2);
        q.delta_LF = 1000.0 / (rate / 2) * kEmnrM; q.delta_MF = 3000.0 / (rate / 2) * kEmnrM;
    }
    q.bsize = dsp_size;
    q.oasize = dsp_size > kEmnrIncr ? dsp_size : kEmnrIncr;
    q.init_oainidx = (kEmnrF - dsp_size - kEmnrIncr) % q.oasize;
    // window (calc_window, wintype 0, emnr.c:160-183)
    std::vector<double> win(kEmnrF);
    {
        const double arg = 2.0 * kPiRef / (double)kEmnrF;
        double sum = 0.0;
        for (int i = 0; i < kEmnrF; i++) { win[(size_t)i] = std::sqrt(0.54 - 0.46 * std::cos((double)i * arg)); sum += win[(size_t)i]; }
        const double inv_coherent_gain = (double)kEmnrF / sum;
        for (double &w : win) w *= inv_coherent_gain;
    }
    // start values (emnr.c:309-313,409-426,452-456)
    std::vector<double> st((size_t)kEmnrStateDoubles, 0.0);
    for (int k = 0; k < kEmnrM; k++) {
        st[(size_t)(EO_PREVG + k)] = 1.0; st[(size_t)(EO_PREVM + k)] = 1.0;
        st[(size_t)(EO_P + k)] = 0.5; st[(size_t)(EO_SIG + k)] = 0.5; st[(size_t)(EO_PBAR + k)] = 0.5; st[(size_t)(EO_PMINU + k)] = 0.5;
        st[(size_t)(EO_P2BAR + k)] = 0.25;
        st[(size_t)(EO_ACTMIN + k)] = 1.0e300; st[(size_t)(EO_ACTSUB + k)] = 1.0e300;
        for (int ku = 0; ku < kEmnrU; ku++) st[(size_t)(EO_AMB + ku * kEmnrPad + k)] = 1.0e300;
        st[(size_t)(EO_SSIG + k)] = 0.5; st[(size_t)(EO_SPBAR + k)] = 0.5;
    }
    QH_HIP(dev_alloc(&emnr_state, (size_t)nch * kEmnrStateDoubles));
    QH_HIP(dev_alloc(&emnr_scal, (size_t)nch));
    QH_HIP(dev_alloc(&emnr_chan, (size_t)nch));
    QH_HIP(dev_alloc(&emnr_window, (size_t)kEmnrF));
    QH_HIP(dev_alloc(&emnr_GG, (size_t)241 * 241));
    QH_HIP(dev_alloc(&emnr_GGS, (size_t)241 * 241));
    QH_HIP(dev_alloc(&emnr_zeta, (size_t)3600));
    QH_HIP(dev_alloc(&emnr_zeta_true, (size_t)3600));
    const EmnrScalars sc0{ 0, 0, q.init_oainidx, 0, 0, 0, q.V, 0, 1.0 };
    for (int ch = 0; ch < nch; ch++) {
        QH_HIP(hipMemcpyAsync(emnr_state + (size_t)ch * kEmnrStateDoubles, st.data(), st.size() * 8, hipMemcpyHostToDevice, stream));
        QH_HIP(hipMemcpyAsync(emnr_scal + ch, &sc0, sizeof(sc0), hipMemcpyHostToDevice, stream));
    }
    QH_HIP(hipMemcpyAsync(emnr_window, win.data(), win.size() * 8, hipMemcpyHostToDevice, stream));
    QH_HIP(hipMemcpyAsync(emnr_GG, h_GG.data(), h_GG.size() * 8, hipMemcpyHostToDevice, stream));
    QH_HIP(hipMemcpyAsync(emnr_GGS, h_GGS.data(), h_GGS.size() * 8, hipMemcpyHostToDevice, stream));
    QH_HIP(hipMemcpyAsync(emnr_zeta, h_zeta.data(), h_zeta.size() * 8, hipMemcpyHostToDevice, stream));
    QH_HIP(hipMemcpyAsync(emnr_zeta_true, h_zeta_true.data(), h_zeta_true.size() * 4, hipMemcpyHostToDevice, stream));
    QH_HIP(hipStreamSynchronize(stream));
    QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&emnr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, emnr_lds_bytes()));
    dev_bytes += (long long)nch * (kEmnrStateDoubles * 8 + sizeof(EmnrScalars) + sizeof(EmnrChan)) + 2 * 241 * 241 * 8;
    for (ChanCfg &c : cfg) { c.emnr_dirty = true; c.emnr_flush = false; }
    return QH_OK;
}

int Engine::ensure_buffers(long long n_mid)
{
    if (n_mid <= buf_cap) return QH_OK;
    QH_HIP(hipStreamSynchronize(stream));
    // captured launch sequences hold the old buffer addresses: whatever entry point grows the buffers, they are stale now
    drop_graphs(); epoch++;
    for (int i = 0; i < 2; i++) {
        if (buf[i]) { QH_HIP(hipFree(buf[i])); dev_bytes -= buf_cap * nch * (long long)sizeof(double2); buf[i] = nullptr; }
    }
    for (int i = 0; i < 2; i++) {
        QH_HIP(dev_alloc(&buf[i], (size_t)nch * (size_t)n_mid));
        dev_bytes += n_mid * nch * (long long)sizeof(double2);
    }
    buf_cap = n_mid;
    return QH_OK;
}

// chunk partials of the fused meters: whole tiles per channel (a tile's store is not bounds-checked)
// the mask of a fircore stage for the tile in use; the two-group tile reads even bins in group A, odd bins in group B
std::vector<cd> Engine::band_mask(const std::vector<cd> &h) const
{
    std::vector<cd> m = make_mask(h, bnfft);
    if (!band2g) return m;
    std::vector<cd> p(m.size());
    const size_t half = m.size() / 2;
    for (size_t k = 0; k < half; k++) { p[k] = m[2 * k]; p[half + k] = m[2 * k + 1]; }
    return p;
}

int Engine::ensure_meter_partials(long long n_mid, int lout)
{
    const long long need = ((n_mid + lout - 1) / lout) * (lout / 64);
    if (need <= m_part_cap) return QH_OK;
    QH_HIP(hipStreamSynchronize(stream));
    drop_graphs(); epoch++;
    for (int i = 0; i < 2; i++) {
        if (m_part[i]) { QH_HIP(hipFree(m_part[i])); dev_bytes -= m_part_cap * nch * (long long)sizeof(double2); m_part[i] = nullptr; }
        QH_HIP(dev_alloc(&m_part[i], (size_t)nch * (size_t)need));
        dev_bytes += need * nch * (long long)sizeof(double2);
    }
    m_part_cap = need;
    return QH_OK;
}

int Engine::ensure_abuf(long long n)
{
    if (n <= abuf_cap) return QH_OK;
    QH_HIP(hipStreamSynchronize(stream));
    drop_graphs(); epoch++;
    if (abuf) { QH_HIP(hipFree(abuf)); dev_bytes -= abuf_cap * nch * (long long)sizeof(double2); abuf = nullptr; }
    QH_HIP(dev_alloc(&abuf, (size_t)nch * (size_t)n));
    abuf_cap = n;
    dev_bytes += n * nch * (long long)sizeof(double2);
    return QH_OK;
}

void Engine::pack_audio(const double2 *src, long long src_stride, long long n)
{
    const long long per = (n + 255) / 256;
    hipLaunchKernelGGL(egress_pack_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)nch), dim3(256), 0, stream, src, src_stride,
                       (int)n, eg);
}

void Engine::tick(int cat)
{
    if (!timing) return;
    if (ev_used >= (int)ev.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        ev.push_back(e);
        ev_cat.push_back(0);
    }
    ev_cat[(size_t)ev_used] = cat;
    (void)hipEventRecord(ev[(size_t)ev_used], stream);
    ev_used++;
}

template <int D, bool MIX, bool PACKED = false, bool METER = false, bool OUTMIX = false, bool EGRESS = false, int NFFT = kNfft, bool POLY = false,
          int DET = 0, bool PAIR = false>
static void launch_osfir(OsfirArgs<double> a, int ntiles, int nch, hipStream_t s)
{
    a.ntiles = ntiles;
    dim3 grid((unsigned)ntiles * (unsigned)nch), block(NT);      // 1-D: the kernel maps ids to (channel, tile), qh_osfir.hpp
    constexpr int lds = osfir_lds_bytes<double, NFFT, D, METER>();
    hipLaunchKernelGGL((osfir_kernel<double, NFFT, D, MIX, PACKED, METER, OUTMIX, EGRESS, POLY, DET, PAIR>), grid, block, lds, s, a);
}
template <int NFFT>
static void launch_band(OsfirArgs<double> a, int ntiles, int nch, hipStream_t s, bool meter, bool egress)
{
    if (meter && egress) launch_osfir<1, false, false, true, false, true, NFFT>(a, ntiles, nch, s);
    else if (meter) launch_osfir<1, false, false, true, false, false, NFFT>(a, ntiles, nch, s);
    else if (egress) launch_osfir<1, false, false, false, false, true, NFFT>(a, ntiles, nch, s);
    else launch_osfir<1, false, false, false, false, false, NFFT>(a, ntiles, nch, s);
}

static void launch_band6k(OsfirArgs<double> a, int ntiles, int nch, hipStream_t s, bool meter, bool egress)
{
    a.ntiles = ntiles;
    dim3 grid((unsigned)ntiles * (unsigned)nch), block(kOsfir6kThreads);
    constexpr int lds = osfir6k_lds_bytes();
    if (meter && egress) hipLaunchKernelGGL((osfir6k_kernel<true, true>), grid, block, lds, s, a);
    else if (meter) hipLaunchKernelGGL((osfir6k_kernel<true, false>), grid, block, lds, s, a);
    else if (egress) hipLaunchKernelGGL((osfir6k_kernel<false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((osfir6k_kernel<false, false>), grid, block, lds, s, a);
}

static void launch_band2g(OsfirArgs<double> a, int ntiles, int nch, hipStream_t s, bool meter, bool egress)
{
    a.ntiles = ntiles;
    // QH_BAND8_FORM=seq: the two halves one after the other on 256 lanes (osfir8s_kernel) instead of the two lane groups side by side
    // (osfir8k_kernel): same masks, same meter partials, same tile geometry.  Measured slower still (profiles/r05_notes.md: the second
    // read of the tile and the parked half go through memory), so it is there for experiments only; never for narrowed outputs.
    // The kernel is in experiment builds only (-DQH_EXP_BAND8_SEQ, tools/ab_bench.py).
#ifdef QH_EXP_BAND8_SEQ
    static const bool seq = [] { const char *e = std::getenv("QH_BAND8_FORM"); return e && std::strcmp(e, "seq") == 0; }();
    if (!egress && seq && a.stash) {
        dim3 g1((unsigned)ntiles * (unsigned)nch);
        if (meter) hipLaunchKernelGGL((osfir8s_kernel<true>), g1, dim3(NT), kOsfir8kImage, s, a);
        else hipLaunchKernelGGL((osfir8s_kernel<false>), g1, dim3(NT), kOsfir8kImage, s, a);
        return;
    }
#endif
    dim3 grid((unsigned)ntiles * (unsigned)nch), block(kOsfir8kThreads);
    constexpr int lds = osfir8k_lds_bytes();
    if (meter && egress) hipLaunchKernelGGL((osfir8k_kernel<true, true>), grid, block, lds, s, a);
    else if (meter) hipLaunchKernelGGL((osfir8k_kernel<true, false>), grid, block, lds, s, a);
    else if (egress) hipLaunchKernelGGL((osfir8k_kernel<false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((osfir8k_kernel<false, false>), grid, block, lds, s, a);
}

// ---- stage helpers ---------------------------------------------------------------------------
// front: xshift + xresample(in) over all channels
// part 0: the whole stage.  The mixed-mode path runs the FM channels and the others on two streams: part 1 = the oscillator's tile
// table for every channel, part 2 = the tile kernel for the listed channels (on whatever `stream` is at the moment), part 3 = the
// history rows and the oscillator phases of every channel.  Parts 1 - 3 exist for the overlap-save form (D > 1) only.
int Engine::run_front(const double2 *src, long long src_stride, double2 *dst, long long dst_stride, const EpiParam *ep,
                      long long n_in, long long n_mid, const int *list, int nlist, int part)
{
    if (part == 0) tick(0);
    if (D > 1) {
        OsfirArgs<double> a{};
        a.in = src; a.in_stride = src_stride;
        a.hist = hist_front[cur_front]; a.hist_stride = kHistFront; a.hist_len = kHistFront;
        a.out = dst; a.out_stride = dst_stride; a.out_offset = 0;
        a.mask = mask_front; a.mask_stride = kNfft;
        a.tw_fwd = tw4096; a.tw_inv = tw_inv_front;
        a.nco_step = nco_step; a.lane_rot = lane_rot;
        a.epi = ep;
        a.n_in = (int)n_in; a.n_out = (int)n_mid; a.off = 0; a.P = front_P; a.Lout = front_L; a.pick = front_pick;
        const int per_tile = front_L / front_pick;
        const int ntiles = (int)((n_mid + per_tile - 1) / per_tile);
        a.chan_list = list;
        const int nl = list ? nlist : nch;
        if (part <= 1 && ntiles > tile_rot_cap) {
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            if (tile_rot) { QH_HIP(hipFree(tile_rot)); dev_bytes -= tile_rot_cap * nch * (long long)sizeof(double2); tile_rot = nullptr; }
            QH_HIP(dev_alloc(&tile_rot, (size_t)nch * (size_t)ntiles));
            tile_rot_cap = ntiles;
            dev_bytes += (long long)ntiles * nch * (long long)sizeof(double2);
        }
        // oscillator phasor at the first input index of every tile: g0 = off - P + tile * fold * Lout (qh_osfir.hpp)
        if (part <= 1) hipLaunchKernelGGL(nco_tile_kernel, dim3((unsigned)((ntiles + 255) / 256), (unsigned)nch), dim3(256), 0, stream,
                           (const unsigned long long *)nco_phase, (const unsigned long long *)nco_dphase, tile_rot, ntiles,
                           (long long)(a.off - a.P), (long long)front_fold * front_L);
        a.tile_rot = tile_rot;
        a.pk_src = pk_src; a.pk = pk;
        // a call at least one delay line long: the tiles that hold its tail leave the next call's line (OsfirArgs::hist_next), every channel
        // in the pass of its own list
        const bool hist_in_kernel = !pk_src && n_in >= kHistFront && n_mid > 0;
        if (hist_in_kernel) a.hist_next = hist_front[cur_front ^ 1];
        if (part == 1 || part == 3) { }
        else if (pk_src) {
            switch (front_fold) {
            case 2: launch_osfir<2, false, true, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            case 4: launch_osfir<4, false, true, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            case 8: launch_osfir<8, false, true, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            default: return set_error(QH_ERR_UNSUPPORTED, "decimation %d not supported", D);
            }
        } else {
            switch (front_fold) {
            case 2: launch_osfir<2, false, false, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            case 4: launch_osfir<4, false, false, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            case 8: launch_osfir<8, false, false, false, true, false, kNfft, true>(a, ntiles, nl, stream); break;
            default: return set_error(QH_ERR_UNSUPPORTED, "decimation %d not supported", D);
            }
        }
        if (part == 0) tick(2);
        if (part == 1 || part == 2) return QH_OK;
        dim3 g((kHistFront + NT - 1) / NT, (unsigned)nch);
        if (pk_src)
            hipLaunchKernelGGL((hist_update_kernel<double, false, true>), g, dim3(NT), 0, stream, src, src_stride, (int)n_in,
                               hist_front[cur_front], hist_front[cur_front ^ 1], kHistFront, (const unsigned long long *)nullptr,
                               (const unsigned long long *)nullptr, (const int *)nullptr, pk_src, pk);
        else if (!hist_in_kernel)
            hipLaunchKernelGGL((hist_update_kernel<double, false>), g, dim3(NT), 0, stream, src, src_stride, (int)n_in,
                               hist_front[cur_front], hist_front[cur_front ^ 1], kHistFront, (const unsigned long long *)nullptr,
                               (const unsigned long long *)nullptr, (const int *)nullptr, (const unsigned char *)nullptr, PackedFmt{});
        cur_front ^= 1;
    } else if (D == 0) {
        // any other ratio: xshift into the staging rows, then the polyphase resampler (its ring and phase live in qh_rat)
        if (pk_src) return set_error(QH_ERR_UNSUPPORTED, "packed input needs in_rate / dsp_rate in 2, 4, 8, 16");
        if (n_in > fbuf_cap) {
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            if (fbuf) { QH_HIP(hipFree(fbuf)); dev_bytes -= fbuf_cap * nch * (long long)sizeof(double2); fbuf = nullptr; }
            QH_HIP(dev_alloc(&fbuf, (size_t)nch * (size_t)n_in));
            fbuf_cap = n_in;
            dev_bytes += n_in * nch * (long long)sizeof(double2);
        }
        long long per = (n_in + NT - 1) / NT;
        dim3 g((unsigned)(per < 4096 ? per : 4096), (unsigned)nch);
        hipLaunchKernelGGL((pointwise_kernel<double, true>), g, dim3(NT), 0, stream, src, src_stride, fbuf, fbuf_cap,
                           (int)n_in, nco_phase, nco_dphase, (const EpiParam *)nullptr, (const int *)nullptr);
        int got = 0;
        if (int rc = qh_rat_process(rsmpin, fbuf, fbuf_cap, (int)n_in, dst, dst_stride, &got)) return rc;
        if (got != (int)n_mid) return set_error(QH_ERR_HIP, "input resampler produced %d samples, expected %lld", got, n_mid);
        if (ep) {       // the front stage is the chain's last: fixed AGC gain and panel, in place
            long long pm = (n_mid + NT - 1) / NT;
            hipLaunchKernelGGL((pointwise_kernel<double, false>), dim3((unsigned)(pm < 1024 ? pm : 1024), (unsigned)nch), dim3(NT), 0, stream,
                               (const double2 *)dst, dst_stride, dst, dst_stride, (int)n_mid, (const unsigned long long *)nullptr,
                               (const unsigned long long *)nullptr, ep, (const int *)nullptr);
        }
        tick(2);
    } else {
        if (pk_src) return set_error(QH_ERR_UNSUPPORTED, "packed input needs in_rate > dsp_rate (unpack with qh_unpack_iq first)");
        long long per = (n_in + NT - 1) / NT;
        dim3 g((unsigned)(per < 4096 ? per : 4096), (unsigned)nch);
        hipLaunchKernelGGL((pointwise_kernel<double, true>), g, dim3(NT), 0, stream, src, src_stride, dst, dst_stride,
                           (int)n_in, nco_phase, nco_dphase, ep, (const int *)nullptr);
        tick(2);
    }
    hipLaunchKernelGGL(nco_advance_kernel, dim3((nch + 255) / 256), dim3(256), 0, stream, nco_phase, nco_dphase, nch, n_in);
    return QH_OK;
}

// ---- impulse responses longer than 4096 taps (kLongPart) -------------------------------------------------------------------
// lcat[ch] = [the stage's last kLongHist input samples | this call's block]
// (lh = the samples of history the stage's partitions look back over, 4096 K - 1: the rows hold kLongHist, what lies further back is not moved)
static __global__ __launch_bounds__(NT) void long_gather_kernel(const double2 *hist, const double2 *src, long long src_stride, int n, double2 *cat,
                                                                long long cat_stride, const int *chan_list, int lh)
{
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const double2 *h = hist + (long long)ch * kLongHist, *x = src + (long long)ch * src_stride;
    double2 *c = cat + (long long)ch * cat_stride;
    const long long tot = (long long)kLongHist + n;
    for (long long i = (long long)(kLongHist - lh) + (long long)blockIdx.x * NT + threadIdx.x; i < tot; i += (long long)gridDim.x * NT) c[i] = i < kLongHist ? h[i] : x[i - kLongHist];
}
// the history the next call finds: the last kLongHist samples of lcat
static __global__ __launch_bounds__(NT) void long_hist_kernel(const double2 *cat, long long cat_stride, int n, double2 *hist, const int *chan_list, int lh)
{
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const double2 *c = cat + (long long)ch * cat_stride + n;
    double2 *h = hist + (long long)ch * kLongHist;
    for (int i = kLongHist - lh + blockIdx.x * NT + threadIdx.x; i < kLongHist; i += gridDim.x * NT) h[i] = c[i];
}
static __global__ __launch_bounds__(NT) void long_add_kernel(double2 *dst, long long dst_stride, const double2 *add, long long add_stride, int n, const int *chan_list)
{
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    double2 *d = dst + (long long)ch * dst_stride;
    const double2 *a = add + (long long)ch * add_stride;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) { d[i].x += a[i].x; d[i].y += a[i].y; }
}

// a stage changes between the one-tile form and the partitioned one because ONE channel's nc moved: the other channels' delay lines go
// along (the 4095 samples of the short history are the tail of the long one; the channel whose nc moved is flushed afterwards, as
// setNc_fircore does)
static __global__ __launch_bounds__(NT) void long_migrate_kernel(double2 *hist, double2 *lhist, int to_long)
{
    const int ch = blockIdx.y;
    double2 *h = hist + (long long)ch * kHistBand, *l = lhist + (long long)ch * kLongHist;
    for (int i = blockIdx.x * NT + threadIdx.x; i < kLongHist; i += gridDim.x * NT) {
        const int j = i - (kLongHist - kHistBand);
        if (to_long) l[i] = j >= 0 ? h[j] : make_double2(0.0, 0.0);
        else if (j >= 0) h[j] = l[i];
    }
}

int Engine::long_stage_alloc(int sid, bool shared_mask)
{
    if (lmask[sid]) return QH_OK;
    QH_HIP(hipStreamSynchronize(stream));
    drop_graphs(); epoch++;
    const size_t rows = shared_mask ? 1 : (size_t)nch;
    QH_HIP(dev_alloc(&lmask[sid], rows * kLongParts * kBandNfftMax));
    QH_HIP(hipMemsetAsync(lmask[sid], 0, rows * kLongParts * kBandNfftMax * sizeof(double2), stream));
    for (int i = 0; i < 2; i++) {
        QH_HIP(dev_alloc(&lhist[sid][i], (size_t)nch * kLongHist));
        QH_HIP(hipMemsetAsync(lhist[sid][i], 0, (size_t)nch * kLongHist * sizeof(double2), stream));
    }
    dev_bytes += (long long)(rows * kLongParts * kBandNfftMax + 2 * (size_t)nch * kLongHist) * (long long)sizeof(double2);
    return QH_OK;
}
int Engine::long_buffers()
{
    if (lcat && lcat_cap == buf_cap) return QH_OK;
    QH_HIP(hipStreamSynchronize(stream));
    if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
    drop_graphs(); epoch++;
    if (lcat) dev_bytes -= (long long)nch * (2 * lcat_cap + kLongHist) * (long long)sizeof(double2);
    (void)hipFree(lcat); (void)hipFree(ltmp); lcat = ltmp = nullptr;
    QH_HIP(dev_alloc(&lcat, (size_t)nch * (size_t)(kLongHist + buf_cap)));
    QH_HIP(dev_alloc(&ltmp, (size_t)nch * (size_t)buf_cap));
    lcat_cap = buf_cap;
    dev_bytes += (long long)nch * (2 * lcat_cap + kLongHist) * (long long)sizeof(double2);
    return QH_OK;
}
// the kLongParts partition masks of the impulse response h (8192-point spectra of its 4096-tap slices; the slices past its end zero)
int Engine::long_masks_upload(int sid, long long row, const std::vector<cd> &h)
{
    std::vector<cd> all((size_t)kLongParts * kBandNfftMax, cd(0.0, 0.0));
    for (int p = 0; p < kLongParts && (size_t)p * kLongPart < h.size(); p++) {
        const size_t a = (size_t)p * kLongPart, b = std::min(h.size(), a + (size_t)kLongPart);
        const std::vector<cd> m = make_mask(std::vector<cd>(h.begin() + (long)a, h.begin() + (long)b), kBandNfftMax);
        std::copy(m.begin(), m.end(), all.begin() + (long)((size_t)p * kBandNfftMax));
    }
    QH_HIP(hipMemcpyAsync(lmask[sid] + (size_t)row * kLongParts * kBandNfftMax, all.data(), all.size() * sizeof(cd), hipMemcpyHostToDevice, stream));
    QH_HIP(hipStreamSynchronize(stream));
    return QH_OK;
}

// one fircore stage (overlap-save, D = 1) over all channels (list == nullptr) or a sub-set
void Engine::run_band(const double2 *src, long long src_stride, double2 *dst, long long dst_stride, const EpiParam *ep,
                      long long n_mid, const double2 *mask, long long mask_stride, double2 **hist, int &hc, int P,
                      const int *list, int nlist, bool meter, bool egress, int det, double *det_out, long long det_stride,
                      const int *pairs, int npairs)
{
    const int sid = hist == hist_nbp ? 0 : hist == hist_bp1 ? 1 : hist == hist_de ? 2 : hist == hist_aud ? 3 : 4;
    if (long_parts[sid] > 1) {
        // nc > 4096: y = sum_p h_p * (x delayed by 4096 p).  lcat holds the stage's last 16383 samples and the block in one row per
        // channel, so partition p is the ordinary tile pass (4096 taps, 8192 points, 4097 outputs per tile) over a row that begins
        // 4096 p samples further back; the partitions' outputs are added, the epilogue follows as a pass of its own.  (The callers
        // switch every fusion off for such a call: no meters, audio frames or detector steps in this stage's stores.)
        const int K = long_parts[sid], Pk = kLongPart - 1, Lk = kBandNfftMax - Pk, nt = (int)((n_mid + Lk - 1) / Lk);
        const int nl = list ? nlist : nch;
        const long long cat_stride = kLongHist + lcat_cap;
        // what the K partitions look back over; a channel's own nc may be shorter (its further masks are zero) and what lies beyond in its row is then
        // whatever an earlier, longer form of the stage left there -- finite samples, times zero
        const int lh = K * kLongPart - 1;
        const long long per = (lh + n_mid + NT - 1) / NT;
        tick(1);
        hipLaunchKernelGGL(long_gather_kernel, dim3((unsigned)(per < 2048 ? per : 2048), (unsigned)nl), dim3(NT), 0, stream, (const double2 *)lhist[sid][hc], src,
                           src_stride, (int)n_mid, lcat, cat_stride, list, lh);
        for (int p = 0; p < K; p++) {
            OsfirArgs<double> a{};
            a.in = lcat + kLongHist - (long long)kLongPart * p; a.in_stride = cat_stride;
            a.hist = a.in - Pk; a.hist_stride = cat_stride; a.hist_len = Pk;
            a.out = p ? ltmp : dst; a.out_stride = p ? lcat_cap : dst_stride; a.out_offset = 0;
            a.mask = lmask[sid] + (size_t)p * kBandNfftMax; a.mask_stride = mask_stride ? (long long)kLongParts * kBandNfftMax : 0;
            a.tw_fwd = a.tw_inv = tw8192;
            a.tw_r2 = tw8192 + 32;
            a.chan_list = list;
            a.n_in = (int)n_mid; a.n_out = (int)n_mid; a.off = 0; a.P = Pk; a.Lout = Lk;
            launch_band<kBandNfftMax>(a, nt, nl, stream, false, false);
            const long long pn = (n_mid + NT - 1) / NT;
            if (p) hipLaunchKernelGGL(long_add_kernel, dim3((unsigned)(pn < 1024 ? pn : 1024), (unsigned)nl), dim3(NT), 0, stream, dst, dst_stride,
                                      (const double2 *)ltmp, lcat_cap, (int)n_mid, list);
        }
        const long long pn = (n_mid + NT - 1) / NT;
        if (ep) hipLaunchKernelGGL((pointwise_kernel<double, false>), dim3((unsigned)(pn < 1024 ? pn : 1024), (unsigned)nl), dim3(NT), 0, stream,
                                   (const double2 *)dst, dst_stride, dst, dst_stride, (int)n_mid, (const unsigned long long *)nullptr,
                                   (const unsigned long long *)nullptr, ep, list);
        tick(2);
        hipLaunchKernelGGL(long_hist_kernel, dim3(64, (unsigned)nl), dim3(NT), 0, stream, (const double2 *)lcat, cat_stride, (int)n_mid, lhist[sid][hc ^ 1], list, lh);
        hc ^= 1;
        return;
    }
    const int Lout = bnfft - P;
    const int ntiles = (int)((n_mid + Lout - 1) / Lout);
    OsfirArgs<double> a{};
    a.in = src; a.in_stride = src_stride;
    a.hist = hist[hc]; a.hist_stride = kHistBand; a.hist_len = kHistBand;
    a.out = dst; a.out_stride = dst_stride; a.out_offset = 0;
    a.mask = mask; a.mask_stride = mask_stride;
    a.tw_fwd = a.tw_inv = (bnfft == kNfft || band2g) ? tw4096 : tw8192;
    a.tw_r2 = tw8192 + 32;                  // second pass table of the 8192-point plan: exp(-2 pi i k / 8192), k < 256 (qh_design.cpp)
    a.stash = band_stash;
    a.epi = ep;
    a.chan_list = list;
    a.n_in = (int)n_mid; a.n_out = (int)n_mid; a.off = 0; a.P = P; a.Lout = Lout;
    tick(1);
    if (meter) { a.meter_in = m_part[0]; a.meter_out = m_part[1]; a.meter_stride = m_part_cap; a.meter_w = m_w; }
    if (egress) a.eg = eg;
    const int nl = list ? nlist : nch;
    // the one-group tile kernel writes the next call's delay line itself when the call is at least one line long (OsfirArgs::hist_next)
    const bool hist_in_kernel = !pairs && (det || (!band6k && !band2g)) && n_mid >= kHistBand;
    if (hist_in_kernel) a.hist_next = hist[hc ^ 1];
    if (pairs) {            // the caller has checked: real taps, one mask per pair, 4096-point tiles, no meters, no egress
        a.chan_list = pairs;
        if (band_fmdc) {    // behind xfmd's loop in its local-dc form: the samples are made in the load (OsfirArgs::fmdc_*)
            a.fmdc_a = band_fmdc->a; a.fmdc_stride = band_fmdc->stride; a.fmdc_shift = band_fmdc->shift;
            a.fmdc_cin = fm_cin; a.fmdc_cstride = fm_cin_cap; a.fmdc_pw = fm_pw; a.fmdc_gain = fm_again;
        }
        if (band_amlv) {    // behind an nbp0 stage that left the envelope and the leveller's local share (DET 3)
            a.amlv_a = band_amlv->a; a.amlv_stride = band_amlv->stride; a.amlv_shift = band_amlv->shift;
            a.amlv_cin = am_cin; a.amlv_cstride = am_cin_cap; a.amlv_pw = am_pw;
        }
        launch_osfir<1, false, false, false, false, false, kNfft, false, 0, true>(a, ntiles, npairs, stream);
    } else if (det) {       // the caller has checked: 4096-point tiles, no meters, no egress
        a.det_out = det_out; a.det_stride = det_stride;
        if (det == 2 || det == 3) {
            a.det_sum = am_tsum; a.det_sum_stride = am_tsum_cap;
            a.det_m[0] = am_prm.mtauR; a.det_m[1] = am_prm.mtauI;
            a.det_m256[0] = std::pow(am_prm.mtauR, 256.0); a.det_m256[1] = std::pow(am_prm.mtauI, 256.0);
            a.det_g[0] = am_prm.onem_mtauR; a.det_g[1] = am_prm.onem_mtauI;
            a.det_lf = levelfade; a.det_last = am_last; a.det_scan = am_pw + 2 * 2048;
            for (int f = 0; f < 2; f++) {
                const double m = a.det_m[f];
                a.det_mp[f][0] = m; a.det_mp[f][1] = m * m; a.det_mp[f][2] = (m * m) * (m * m); a.det_mp[f][3] = ((m * m) * (m * m)) * ((m * m) * (m * m));
            }
            if (det == 3) launch_osfir<1, false, false, false, false, false, kNfft, false, 3>(a, ntiles, nl, stream);
            else launch_osfir<1, false, false, false, false, false, kNfft, false, 2>(a, ntiles, nl, stream);
        } else launch_osfir<1, false, false, false, false, false, kNfft, false, 1>(a, ntiles, nl, stream);
    } else if (band6k) launch_band6k(a, ntiles, nl, stream, meter, egress);
    else if (band2g) launch_band2g(a, ntiles, nl, stream, meter, egress);
    else if (bnfft == kNfft) launch_band<kNfft>(a, ntiles, nl, stream, meter, egress);
    else launch_band<kBandNfftMax>(a, ntiles, nl, stream, meter, egress);
    tick(2);
    dim3 g((kHistBand + NT - 1) / NT, (unsigned)(list ? nlist : nch));
    if (pairs && band_amlv)
        hipLaunchKernelGGL(am_audio_hist_kernel, g, dim3(256), 0, stream, band_amlv->a, band_amlv->stride, (int)n_mid, list, (const double *)am_cin, am_cin_cap,
                           (const double *)am_pw, band_amlv->shift, (const double2 *)hist[hc], hist[hc ^ 1], kHistBand);
    else if (pairs && band_fmdc)
        hipLaunchKernelGGL(fm_audio_hist_kernel, g, dim3(256), 0, stream, band_fmdc->a, band_fmdc->stride, (int)n_mid, list, (const double *)fm_cin, fm_cin_cap,
                           (const double *)fm_pw, band_fmdc->shift, (const double *)fm_again, (const double2 *)hist[hc], hist[hc ^ 1], kHistBand);
    else if (!hist_in_kernel)
        hipLaunchKernelGGL((hist_update_kernel<double, false>), g, dim3(NT), 0, stream, src, src_stride, (int)n_mid,
                           hist[hc], hist[hc ^ 1], kHistBand, (const unsigned long long *)nullptr,
                           (const unsigned long long *)nullptr, list);
    // (a listed stage reads and writes the history rows of its own channels only)
    hc ^= 1;
}

// xrxa's last step (wdsp/RXA.c:596): rsmpout runs when out_rate != dsp_rate (RXAResCheck, RXA.c:789-798)
int Engine::process(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk)
{
    if (!rsmpout) return process_chain(d_in, in_stride, d_out, out_stride, nblk);
    if (nblk <= 0) return QH_OK;
    QH_HIP(hipSetDevice(device));
    if (eg.kind) {      // audio frames behind the output resampler: resample into the staging rows, then narrow
        const long long n_out = (long long)nblk * dsp_outsize;
        if (int rc = ensure_abuf(n_out)) return rc;
        const EgressFmt keep = eg;
        eg = EgressFmt{};
        const int rc = process(d_in, in_stride, reinterpret_cast<double *>(abuf), abuf_cap, nblk);
        eg = keep;
        if (rc) return rc;
        pack_audio(abuf, abuf_cap, n_out);
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
    const long long n_mid = (long long)nblk * dsp_size;
    if (n_mid > obuf_cap) {
        QH_HIP(hipStreamSynchronize(stream));
        drop_graphs(); epoch++;
        if (obuf) { QH_HIP(hipFree(obuf)); dev_bytes -= obuf_cap * nch * (long long)sizeof(double2); obuf = nullptr; }
        QH_HIP(dev_alloc(&obuf, (size_t)nch * (size_t)n_mid));
        obuf_cap = n_mid;
        dev_bytes += n_mid * nch * (long long)sizeof(double2);
    }
    if (int rc = process_chain(d_in, in_stride, reinterpret_cast<double *>(obuf), obuf_cap, nblk)) return rc;
    int got = 0;
    if (int rc = qh_rat_process(rsmpout, obuf, obuf_cap, (int)n_mid, d_out, out_stride, &got)) return rc;
    if (got != nblk * dsp_outsize) return set_error(QH_ERR_HIP, "output resampler produced %d samples, expected %d", got, nblk * dsp_outsize);
    return QH_OK;
}

// The chains' transition over a segment: the one-sample map of the 17 words (ds, x_j[n-1], x_j[n-2]; input 0) raised to the segment's
// length, for the two lengths a call of n samples in S segments has (q and q + 1 batches of 64), chains a / c (coefficients c0, input
// one sample late through ds) and b / d (c1).  Long double on the host; kept until the call shape changes.
int Engine::set_sb_phi(long long n, int S)
{
    const long long key = n * 1024 + S;
    if (key == sb_phi_key) return QH_OK;
    static const long double c0[7] = { -0.328201924180698L, -0.744171491539427L, -0.923022915444215L, -0.978490468768238L,
                                       -0.994128272402075L, -0.998458978159551L, -0.999790306259206L };
    static const long double c1[7] = { -0.0991227952747244L, -0.565619728761389L, -0.857467122550052L, -0.959123933111275L,
                                       -0.988739372718090L, -0.996959189310611L, -0.999282492800792L };
    constexpr int W = 17;
    typedef std::vector<long double> Mat;
    auto mul = [&](const Mat &a, const Mat &b) {
        Mat r((size_t)W * W, 0.0L);
        for (int i = 0; i < W; i++)
            for (int k = 0; k < W; k++) {
                const long double v = a[(size_t)i * W + k];
                if (v != 0.0L) for (int j = 0; j < W; j++) r[(size_t)i * W + j] += v * b[(size_t)k * W + j];
            }
        return r;
    };
    auto one_step = [&](const long double *c, bool delayed) {
        Mat m((size_t)W * W, 0.0L);
        for (int col = 0; col < W; col++) {
            long double v[W] = { 0 }, x[8];
            v[col] = 1.0L;
            x[0] = delayed ? v[0] : 0.0L;                               // the chain's input: ds (a, c) or the external input, 0 here
            for (int j = 0; j < 7; j++) x[j + 1] = c[j] * (x[j] - v[2 + 2 * (j + 1)]) + v[2 + 2 * j];       // amd.c:172-175
            long double nv[W];
            nv[0] = 0.0L;
            for (int j = 0; j < 8; j++) { nv[1 + 2 * j] = x[j]; nv[2 + 2 * j] = v[1 + 2 * j]; }
            for (int r = 0; r < W; r++) m[(size_t)r * W + col] = nv[r];
        }
        return m;
    };
    auto power = [&](Mat b, long long e) {
        Mat r((size_t)W * W, 0.0L);
        for (int i = 0; i < W; i++) r[(size_t)i * W + i] = 1.0L;
        while (e > 0) { if (e & 1) r = mul(b, r); b = mul(b, b); e >>= 1; }
        return r;
    };
    const long long q = ((n + 63) / 64) / S;
    std::vector<double> h((size_t)2 * 2 * W * W);
    for (int li = 0; li < 2; li++)
        for (int set = 0; set < 2; set++) {
            const Mat p = power(one_step(set ? c1 : c0, set == 0), 64 * (q + li));
            for (int i = 0; i < W * W; i++) h[((size_t)li * 2 + set) * W * W + i] = (double)p[(size_t)i];
        }
    if (!sb_phi) {
        QH_HIP(dev_alloc(&sb_phi, h.size()));
        QH_HIP(dev_alloc(&sb_sum, (size_t)nch * kSegWaves * kSegMaxGroups * kSbSum));
        QH_HIP(dev_alloc(&sb_start, (size_t)nch * kSegWaves * kSegMaxGroups * kSbSum));
    }
    QH_HIP(hipStreamSynchronize(stream));
    if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
    drop_graphs(); epoch++;
    QH_HIP(hipMemcpy(sb_phi, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    sb_phi_key = key;
    return QH_OK;
}

int Engine::process_chain(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk)
{
    if (nblk <= 0) return QH_OK;
    QH_HIP(hipSetDevice(device));
    // what the chain of every channel needs
    bool any_nbp = false, any_bp1 = false, mixed = false, every_nbp = true;
    int nc_max = 1;
    for (const ChanCfg &c : cfg) {
        if (c.agc_run && c.agc_mode > 4)
            return set_error(QH_ERR_UNSUPPORTED, "AGC mode %d is not provided (0 fixed, 1-4 long/slow/med/fast)", c.agc_mode);
        // (SetRXAAMDRun can switch the AM detector on beside the FM one, RXA.c:594-595 then runs both in a row: not provided, and said so)
        if (c.amd_run && c.fmd_run) return set_error(QH_ERR_UNSUPPORTED, "channel %d: the AM and the FM detector both switched on", (int)(&c - cfg.data()));
        if (c.amd_run || c.fmd_run || (c.agc_run && c.agc_mode != 0) || c.lms[0].run || c.lms[1].run || c.amsq_run || c.emnr_run || c.snba_run) mixed = true;
        if (c.emnr_run && !emnr_tables) return set_error(QH_ERR_INVALID, "EMNR needs its gain tables first (qh_rxa_SetEMNRTables: WDSP's `calculus` and `zetaHat.bin` data)");
        if (c.nbp_run) { any_nbp = true; if (c.nbp_nc > nc_max) nc_max = c.nbp_nc; } else every_nbp = false;
        if (c.bp1_run) { any_bp1 = true; if (c.bp1_nc > nc_max) nc_max = c.bp1_nc; }
        if (c.fmd_run && c.fm_nc > nc_max) nc_max = c.fm_nc;
    }
    if (nc_max > kLongNcMax) return set_error(QH_ERR_UNSUPPORTED, "nc = %d exceeds %d", nc_max, kLongNcMax);
    // stages whose impulse response is longer than 4096 taps run in partitions (run_band): how many, per stage
    bool long_mode = false;
    {
        auto parts = [](int nc) { return nc > kLongPart ? (nc + kLongPart - 1) / kLongPart : 1; };
        int lp[5] = { 1, 1, 1, 1, 1 };
        // Over every channel that runs the stage OR still holds a long delay line of it: a fircore keeps its delay line while it does not run (xbandpass / xnbp with run = 0
        // only copy; SetRXABandpassRun, a mode change back to AM / FM, RXANBPSetRun switch it on again without a flush), so a channel with
        // nc > 4096 that sits out holds 16383 samples the one-tile form has no room for.  Had the form followed the RUNNING channels,
        // the only long channel leaving took the stage to the short form (its line cut to 4095 samples) and came back to zeros behind
        // them: one long call 0.65 off (walk rxa_long 900190, found by round 6's seed sweep; in the suite since).
        // (long_live: the channel has run the stage with such an nc since RXASetNC last zeroed its lines.)
        for (ChanCfg &c : cfg) {
            const int pn = parts(c.nbp_nc), pb = parts(c.bp1_nc), pf = parts(c.fm_nc);
            if (c.nbp_run && pn > 1) c.long_live[0] = true;
            if (c.bp1_run && pb > 1) c.long_live[1] = true;
            if (c.fmd_run && pf > 1) c.long_live[2] = true;
            if (c.snba_run && pn > 1) c.long_live[4] = true;
            if (c.nbp_run || c.long_live[0]) lp[0] = std::max(lp[0], pn);
            if (c.bp1_run || c.long_live[1]) lp[1] = std::max(lp[1], pb);
            if (c.fmd_run || c.long_live[2]) lp[2] = lp[3] = std::max(lp[2], pf);
            if (c.snba_run || c.long_live[4]) lp[4] = std::max(lp[4], pn);
        }
        for (int sid = 0; sid < 5; sid++) {
            if (lp[sid] > 1) { long_mode = true; if (int rc = long_stage_alloc(sid, sid == 2 || sid == 3)) return rc; }
            if (lp[sid] == long_parts[sid]) continue;
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            if ((lp[sid] > 1) != (long_parts[sid] > 1)) {       // the delay lines move with the form
                double2 **hs = sid == 0 ? hist_nbp : sid == 1 ? hist_bp1 : sid == 2 ? hist_de : sid == 3 ? hist_aud : hist_snb;
                // BOTH ping-pong halves: a channel that has left the stage's list (another mode, bp1 or SNBA switched off) keeps its rows in
                // the half that was current when it left (bp1_hist_at, fm_hist_at, snb_hist_at), which need not be the current one
                for (int half = 0; half < 2; half++)
                    if (hs[half] && lhist[sid][half])
                        hipLaunchKernelGGL(long_migrate_kernel, dim3(16, (unsigned)nch), dim3(NT), 0, stream, hs[half], lhist[sid][half], lp[sid] > 1 ? 1 : 0);
            }
            long_parts[sid] = lp[sid];
            for (ChanCfg &c : cfg) {            // the stage's masks are laid out for another form now: all of them again
                if (sid == 0) c.nbp_dirty = true;
                if (sid == 1) c.bp1_dirty = true;
                if (sid == 4) c.snb_dirty = true;
            }
            if (sid == 2 || sid == 3) fm_nc_built = 0;
        }
    }
    {   // The fircore tile.  Impulse responses longer than 2048 taps need 8192 points (osfir_kernel<8192>, one wave per SIMD).
        // Shorter ones run 4096-point tiles; the two-group 8192-point tile (osfir8k_kernel, 6144 instead of 2049 outputs per
        // pair of transforms) is there on request (qh_rxa_set_band_tile) -- it measured slower, see qh_osfir.hpp.  The masks are
        // spectra of the tile size, so a change rebuilds every one of them (the delay lines, kept 4095 samples deep, carry over).
        static const bool force8k = [] { const char *e = std::getenv("QH_BAND_NFFT"); return e && std::atoi(e) == 8192; }();  // tuning experiments
        const bool two_group = nc_max <= 2048 && band_tile_pref == 8192 && !force8k;
        const bool six_k = nc_max <= 2048 && band_tile_pref == 6144 && !force8k;
        const int want = six_k ? kOsfir6kN : (nc_max > 2048 || force8k || two_group) ? kBandNfftMax : kNfft;
#ifdef QH_EXP_BAND8_SEQ
        if (two_group && !band_stash) {         // (osfir8s_kernel's scratch rows: the experiment builds' kernel only)
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            QH_HIP(dev_alloc(&band_stash, (size_t)nch * 4096));
            dev_bytes += (long long)nch * 4096 * (long long)sizeof(double2);
        }
#endif
        if (want != bnfft || two_group != band2g || six_k != band6k) {
            bnfft = want; band2g = two_group; band6k = six_k;
            for (ChanCfg &c : cfg) { c.nbp_dirty = c.bp1_dirty = true; c.snb_dirty = true; }
        }
    }
    // The three meters of xrxa (adc, S, agc: RXA.c:566,569,589) ride on the nbp0 launch when the chain is linear (nbp0 runs,
    // bp1 does not, fixed AGC gain): the band tile then starts on a multiple of 256 samples so that a register holds one
    // 64-sample chunk per wavefront.  Any other chain takes the per-mode path with the stand-alone meter kernel.
    const bool meters_fused = meters_on && !mixed && any_nbp && !any_bp1 && dsp_size >= 64 && dsp_size <= 2048 && !long_mode;
    if (meters_on && !meters_fused) mixed = true;
    if (meters_on) if (int rc = meters_alloc()) return rc;
    if (int rc = refresh_params()) return rc;
    if (mixed) if (int rc = refresh_demod()) return rc;
    if (n_snba) if (int rc = refresh_params()) return rc;       // bpsnba's mask needs the buffers the line above may just have made

    const long long n_in = (long long)nblk * dsp_insize;
    const long long n_mid = (long long)nblk * dsp_size;
    if (n_in > 0x7fffffffLL) return set_error(QH_ERR_INVALID, "too many samples in one call");
    if (int rc = ensure_buffers(n_mid)) return rc;
    if (long_mode) if (int rc = long_buffers()) return rc;
    ev_used = 0;

    const double2 *in = reinterpret_cast<const double2 *>(d_in);
    double2 *out = reinterpret_cast<double2 *>(d_out);
    // (with a partitioned stage in the call the others run 4096-tap tiles: their own nc is at most that)
    const int P = band6k ? kOsfir6kP : band2g ? kOsfir8kP : meters_fused ? ((nc_max - 1 + 255) / 256) * 256 : long_mode ? kLongPart - 1 : nc_max - 1;

    // audio egress (qh_rxa_process_audio): the narrowing rides in the store of the last kernel when that is an overlap-save
    // band stage or the per-mode path's output pass; other endings write complex doubles to the staging rows and narrow after
    const bool eg_fused = eg.kind && (mixed ? n_amsq == 0 : ((any_nbp || any_bp1) && !long_mode));
    if (eg.kind && !eg_fused) {
        if (int rc = ensure_abuf(n_mid)) return rc;
        out = abuf; out_stride = abuf_cap;
    }
    if (!mixed) {
        // ---- every channel is a linear chain: the epilogue rides on the last stage, no extra pass
        const int nstage = 1 + (any_nbp ? 1 : 0) + (any_bp1 ? 1 : 0);
        int stage = 0, which = 0;
        const double2 *cur = in;
        long long cur_stride = in_stride;
        auto dst_of = [&](int st, long long &stride) -> double2 * {
            if (st == nstage - 1) { stride = out_stride; return out; }
            stride = buf_cap;
            double2 *p = buf[which];
            which ^= 1;
            return p;
        };
        {
            long long dst_stride;
            double2 *dst = dst_of(stage, dst_stride);
            // A chain that is the front stage alone stores to the caller's rows while other tiles -- and the history pass behind the kernel --
            // still read the input: when the rows of the two matrices overlap (include/quiskhip.h: the output may lie over the input) the
            // stage goes to the engine's own rows and the output is written in a last pass.  (The rows' own extent, as in the mixed path.)
            const bool over = nstage == 1 && !((const char *)(out + (size_t)(nch - 1) * (size_t)out_stride + (size_t)n_mid) <= (const char *)in ||
                                              (const char *)in + ((size_t)(nch - 1) * (size_t)in_stride + (size_t)n_in) * sizeof(double2) <= (const char *)out);
            if (over) { dst = buf[0]; dst_stride = buf_cap; }
            if (int rc = run_front(cur, cur_stride, dst, dst_stride, stage == nstage - 1 ? epi : nullptr, n_in, n_mid)) return rc;
            if (over) QH_HIP(hipMemcpy2DAsync(out, (size_t)out_stride * sizeof(double2), dst, (size_t)dst_stride * sizeof(double2),
                                              (size_t)n_mid * sizeof(double2), (size_t)nch, hipMemcpyDeviceToDevice, stream));
            cur = dst; cur_stride = dst_stride; stage++;
        }
        for (int f = 0; f < 2; f++) {
            if (f == 0 ? !any_nbp : !any_bp1) continue;
            long long dst_stride;
            double2 *dst = dst_of(stage, dst_stride);
            if (meters_fused) if (int rc = ensure_meter_partials(n_mid, bnfft - P)) return rc;
            run_band(cur, cur_stride, dst, dst_stride, stage == nstage - 1 ? epi : nullptr, n_mid,
                     f == 0 ? mask_nbp : mask_bp1, kBandNfftMax, f == 0 ? hist_nbp : hist_bp1, f == 0 ? cur_nbp : cur_bp1, P,
                     nullptr, 0, meters_fused, eg_fused && stage == nstage - 1);
            cur = dst; cur_stride = dst_stride; stage++;
        }
        if (meters_fused)
            hipLaunchKernelGGL(meter_finish_kernel, dim3((unsigned)nch), dim3(kMeterFinishThreads), 0, stream, m_part[0], m_part[1],
                               m_part_cap, (int)(n_mid / 64), dsp_size / 64, (bnfft - P) / 64, band6k ? 2 : band2g ? 1 : 0, m_adc, m_s, m_agc,
                               -1.0 / ((double)dsp_rate * 0.100), -1.0 / ((double)dsp_rate * 0.100), (const double *)m_g2);
        if (eg.kind && !eg_fused) pack_audio(out, out_stride, n_mid);
        tick(3);
        QH_HIP(hipGetLastError());
        return QH_OK;
    }

    // ---- mixed modes: per-mode stages run on channel lists; gains/panel in a final pointwise pass
    double2 *cur = buf[0], *other = buf[1];
    // FM channels have a long detector chain of kernels that fill a fraction of the chip (one lane per 256-sample tile of the loop,
    // a few workgroups per channel in the scans) while the other channels' work is dense filtering.  With both kinds in the call
    // the FM channels' front and nbp0 stages are launched first and their detector chain follows on the main stream; the other
    // channels' front, nbp0 and AM detectors run beside it on a second stream (fork / join by events, which a launch-sequence
    // capture records as graph edges).  BASELINE config 4: 1.20 -> 0.8 ms per call.
    // dbg_forms (QH_DBG_FORMS in the environment when the engine is made; diagnostics: tools/dbg/determinism_stress2.py and the failure
    // branch of tests/test_gpu_properties_fullsize.py): bit 0 no second stream for the filters, 1 no stores straight to the caller's rows,
    // 2 no envelope in nbp0's store, 3 no angles in nbp0's store, 4 no paired real filters, 5 no second stream at all, 6-7 where the second
    // stream forks (counted down from behind the FM channels' nbp0), 8 xfmd's dc removal as a pass of its own (fm_dc_tiled_kernel), 9 the AM
    // fade leveller as a pass of its own (am_level_tiled_kernel)
    const bool split = n_fm > 0 && n_rest > 0 && D > 1 && !meters_on && !n_amsq && !n_snb[0] && !timing && !(dbg_forms & 1);
    // ... and when nothing sits between a channel's last filter and the output matrix (no AGC state machine, LMS, EMNR, SNBA,
    // limiter, squelch or position-1 stage anywhere), that last stage -- nbp0 for the plain channels, bp1 for AM / SAM, the CTCSS
    // notch for FM -- applies the matrix in its store and writes the caller's buffer: the output pass (32 B per output sample) goes.
    const bool fm_theta_fused = split && any_nbp && !band6k && !band2g && bnfft == kNfft && !(dbg_forms & 8);
    bool no_lms = true;
    for (int f = 0; f < 2; f++) for (int k = 0; k < 3; k++) no_lms = no_lms && !n_lms[f][k];
    bool agc_direct = false;            // set where xwcpagc runs: its gain multiply writes the caller's rows (see there)
    const bool direct = split && !(dbg_forms & 2) && every_nbp && !eg.kind && !n_lim && !n_agc_cur && !n_agc_other && !n_snba && !n_snb[1] && no_lms &&
                        !n_emnr[0] && !n_emnr[1] && !n_emnr[2] && !n_fix[0] && !n_fix[1] && !n_bp1p[1] && n_bp1p[0] == n_bp1 && n_rb == n_bp1 &&
                        n_usb + n_fm == n_plain &&
                        // the first stores to `out` come while other channels' input is still being read: not for a caller that works in place.
                        // (The extents are the rows' own: first sample of the first row to last sample of the last.  nch * stride from a
                        // pointer INTO a matrix -- a caller walking along its rows call by call -- reaches past the matrix's end by the
                        // offset, and whether that touches the input depended on where the allocator had put the two: the same calls took
                        // this form or the other -- AM channels' last bits, tests/test_gpu_properties_fullsize.py -- by address.)
                        ((const char *)(out + (size_t)(nch - 1) * (size_t)out_stride + (size_t)n_mid) <= (const char *)in ||
                         (const char *)in + ((size_t)(nch - 1) * (size_t)in_stride + (size_t)n_in) * sizeof(double2) <= (const char *)out);
    // ... and the AM channels' nbp0 leaves the envelope and every tile's share of the fade leveller's averages: one pass does the rest
    const int P_am = ((P + 63) / 64) * 64;
    const bool am_fused = direct && !(dbg_forms & 4) && n_am > 0 && n_rb == n_am + n_sam && !band6k && !band2g && bnfft == kNfft && P_am < bnfft;
    // ... or, when the tiles are 2048 outputs behind 2048 samples of pre-roll and bp1 takes its channels two a tile, no pass at all: the
    // leveller's local share in nbp0's store (DET 3), the carried share in bp1's load
    const bool am_lv_fused = am_fused && P_am == 2048 && bnfft - P_am == 2048 && np_am > 0 && n_bp1p[0] && long_parts[1] <= 1 && !(dbg_forms & (16 | 512));
    if (am_fused) {
        const long long nt = (n_mid + (bnfft - P_am) - 1) / (bnfft - P_am);
        if (nt > am_tsum_cap) {
            QH_HIP(hipStreamSynchronize(stream));
            if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
            drop_graphs(); epoch++;
            (void)hipFree(am_tsum); am_tsum = nullptr;
            QH_HIP(dev_alloc(&am_tsum, (size_t)nch * (size_t)nt * 2));
            am_tsum_cap = nt;
        }
        if (am_lv_fused && am_tsum_cap + 1 > am_cin_cap) {
            QH_HIP(hipStreamSynchronize(stream));
            if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
            drop_graphs(); epoch++;
            (void)hipFree(am_cin); am_cin = nullptr;
            QH_HIP(dev_alloc(&am_cin, (size_t)nch * (size_t)(am_tsum_cap + 1) * 2));
            am_cin_cap = am_tsum_cap + 1;
        }
    }
    if (split) {
        if (!side_stream) {
            QH_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            QH_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            QH_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        if (int rc = run_front(in, in_stride, cur, buf_cap, nullptr, n_in, n_mid, nullptr, 0, 1)) return rc;      // tile table, every channel
        // Where the second stream starts: behind the FM channels' nbp0 (2; the default), behind their front (1) or at once (0) -- QH_DBG_FORMS
        // bits 6-7 count DOWN from 2 (experiments).  The FM channels' chain is the longer one and ends in kernels that cannot fill the chip
        // (the loop's lanes, the CTCSS notch's scans): with their front and nbp0 alone on the chip first, all of them run beside the other
        // channels' dense filters (config 4, one box: 10.45 ms forked at once, 10.31 forked here).
        const int fork_at = 2 - (((dbg_forms >> 6) & 3) > 2 ? 2 : ((dbg_forms >> 6) & 3));
        if (fork_at == 0) { QH_HIP(hipEventRecord(ev_fork, stream)); QH_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0)); }
        if (int rc = run_front(in, in_stride, cur, buf_cap, nullptr, n_in, n_mid, list_fm, n_fm, 2)) return rc;
        if (fork_at == 1) { QH_HIP(hipEventRecord(ev_fork, stream)); QH_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0)); }
        int hc = cur_nbp;
        // the FM channels' nbp0 feeds the loop's phase detector and nothing else: its store takes the angles (first half of the rows)
        if (any_nbp) run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P, list_fm, n_fm, false, false,
                              fm_theta_fused ? 1 : 0, reinterpret_cast<double *>(other), 2 * buf_cap);
        if (fork_at >= 2) { QH_HIP(hipEventRecord(ev_fork, stream)); QH_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0)); }
        std::swap(stream, side_stream);
        int rc2 = run_front(in, in_stride, cur, buf_cap, nullptr, n_in, n_mid, list_rest, n_rest, 2);
        hc = cur_nbp;
        if (!rc2 && any_nbp) {
            if (direct) {       // the plain channels end here: output matrix in the store, straight to the caller's buffer
                if (n_usb) run_band(cur, buf_cap, out, out_stride, epi, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P, list_usb, n_usb);
                hc = cur_nbp;
                if (am_fused) {
                    if (n_sam) run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P, list_sam, n_sam);
                    hc = cur_nbp;
                    run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P_am, list_am, n_am, false, false,
                             am_lv_fused ? 3 : 2, reinterpret_cast<double *>(other), 2 * buf_cap);
                } else if (n_rb) run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P, list_rb, n_rb);
            } else run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, hc, P, list_rest, n_rest);
        }
        std::swap(stream, side_stream);
        if (rc2) return rc2;
        if (any_nbp) { cur_nbp ^= 1; std::swap(cur, other); }
        if (int rc = run_front(in, in_stride, cur, buf_cap, nullptr, n_in, n_mid, nullptr, 0, 3)) return rc;          // histories, oscillator phases
    } else {
    if (int rc = run_front(in, in_stride, cur, buf_cap, nullptr, n_in, n_mid)) return rc;
    if (meters_on) hipLaunchKernelGGL(meter_kernel, dim3((unsigned)nch), dim3(64), 0, stream, cur, buf_cap, nblk, dsp_size, m_adc,
                                      m_prm, (const int *)nullptr);
    if (any_nbp) {
        run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_nbp, kBandNfftMax, hist_nbp, cur_nbp, P, nullptr, 0);
        std::swap(cur, other);
    }
    }
    if (meters_on) hipLaunchKernelGGL(meter_kernel, dim3((unsigned)nch), dim3(64), 0, stream, cur, buf_cap, nblk, dsp_size, m_s,
                                      m_prm, (const int *)nullptr);
    if (n_amsq) {           // xamsqcap (RXA.c:571): the magnitudes of the signal behind nbp0, for xamsq at the end of the chain
        if (buf_cap > amsq_mag_cap) {
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            (void)hipFree(amsq_mag); amsq_mag = nullptr;
            QH_HIP(dev_alloc(&amsq_mag, (size_t)nch * (size_t)buf_cap));
            amsq_mag_cap = buf_cap;
        }
        long long per = (n_mid + NT - 1) / NT;
        hipLaunchKernelGGL(amsq_cap_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)n_amsq), dim3(NT), 0, stream, cur, buf_cap,
                           (int)n_mid, list_amsq, amsq_mag, amsq_mag_cap);
    }
    // xbpsnbaout at position 0 (RXA.c:572): the 250..5700 Hz filter of the signal ahead of nbp0 replaces nbp0's output
    auto snb_inplace = [&](const int *list, int n) {
        int hc = cur_snb;
        run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_snb, kBandNfftMax, hist_snb, hc, P, list, n);
        long long per = (n_mid + NT - 1) / NT;
        hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)n), dim3(NT), 0, stream, other, cur, buf_cap,
                           (int)n_mid, list);
    };
    if (n_snb[0]) {
        if (any_nbp) { int hc = cur_snb; run_band(other, buf_cap, cur, buf_cap, nullptr, n_mid, mask_snb, kBandNfftMax, hist_snb, hc, P, list_snb[0], n_snb[0]); }
        else snb_inplace(list_snb[0], n_snb[0]);
    }
    tick(1);
    // The AM / SAM detectors and the FM detector chain touch disjoint channel rows and disjoint state, and neither fills the
    // chip (one workgroup or wavefront per channel): with both kinds of channel in the call the AM side runs on a second
    // stream, forked and joined by events (which a launch-sequence capture records as graph edges).
    const bool side = split || ((n_am || n_sam) && n_fm && !(dbg_forms & 32));
    hipStream_t am_stream = stream;
    if (split) am_stream = side_stream;         // forked already: the AM detectors follow the other channels' filters there
    else if (side) {
        if (!side_stream) {
            QH_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            QH_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            QH_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        QH_HIP(hipEventRecord(ev_fork, stream));
        QH_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
        am_stream = side_stream;
    }
    // Segment scans: one 16-wavefront workgroup per channel fills one CU.  With fewer channels of a kind than the chip has CUs the
    // call is cut into 16 G segments, G workgroups per channel, pass 1 and pass 2 as two launches (qh_wave.hpp, MODE 1 / 2).
    auto seg_groups = [&](int count) {
        int G = count > 0 ? 256 / count : 1;
        while (G > 1 && n_mid / (64LL * kSegWaves * G) < 8) G--;        // at least 8 batches of 64 samples per segment
        return G < 1 ? 1 : G > kSegMaxGroups ? kSegMaxGroups : G;
    };
    for (double *&q : seg_sum)
        if (!q) QH_HIP(dev_alloc(&q, (size_t)nch * kSegWaves * kSegMaxGroups * kSegSumW));
    FmDcSrc amlv_src{ nullptr, 0, 0 };
    if (n_am && am_lv_fused) {      // the envelope + the leveller's local share lie in the channels' own rows (first half); bp1 loads them from there
        hipLaunchKernelGGL(am_lv_chain_kernel, dim3((unsigned)n_am), dim3(64), 0, am_stream, (int)n_mid, bnfft - P_am, list_am, (const int *)levelfade, am_state,
                           am_prm, (const double *)am_tsum, am_tsum_cap, (const double *)am_last, am_cin, am_cin_cap);
        amlv_src = FmDcSrc{ reinterpret_cast<double *>(cur), 2 * buf_cap, 11 };
    } else if (n_am && am_fused) {  // envelopes in the channels' own rows (first half), audio to the rows of `other`
        const int G = seg_groups(n_am + (n_mid >= kSamTiledMin ? n_sam0 : 0));
        hipLaunchKernelGGL(am_level_tiled_kernel, dim3((unsigned)n_am, (unsigned)G), dim3(kSegThreads), 0, am_stream,
                           (const double *)reinterpret_cast<double *>(cur), 2 * buf_cap, other, buf_cap, (int)n_mid, list_am, levelfade, (const AmState *)am_state,
                           am_prm, (const double *)am_tsum, am_tsum_cap, bnfft - P_am, am_next);
        hipLaunchKernelGGL(commit_am_kernel, dim3((unsigned)((n_am + 255) / 256)), dim3(256), 0, am_stream, am_state, (const AmState *)am_next, list_am, n_am, levelfade);
    } else if (n_am) {
        const int G = seg_groups(n_am + (n_mid >= kSamTiledMin ? n_sam0 : 0));
        if (G > 1) {
            hipLaunchKernelGGL((am_detect_tiled_kernel<false, 1>), dim3((unsigned)n_am, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur, buf_cap,
                               (int)n_mid, list_am, levelfade, am_state, am_prm, (const double *)nullptr, 0LL, seg_sum[0]);
            hipLaunchKernelGGL((am_detect_tiled_kernel<false, 2>), dim3((unsigned)n_am, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur, buf_cap,
                               (int)n_mid, list_am, levelfade, am_state, am_prm, (const double *)nullptr, 0LL, seg_sum[0], am_next);
            hipLaunchKernelGGL(commit_am_kernel, dim3((unsigned)((n_am + 255) / 256)), dim3(256), 0, am_stream, am_state, (const AmState *)am_next, list_am, n_am, levelfade);
        } else
            hipLaunchKernelGGL((am_detect_tiled_kernel<false, 0>), dim3((unsigned)n_am), dim3(kSegThreads), 0, am_stream, cur, buf_cap, (int)n_mid,
                               list_am, levelfade, am_state, am_prm, (const double *)nullptr, 0LL, (double *)nullptr);
    }
    {
        // SAM without sideband separation in a long call: angles, the loop one tile per lane with a warm-up, verify / repair,
        // then the mix with the phase each sample saw and the fade leveller over time segments (qh_tiled.hpp).  The channels'
        // rows of `other` are free here: first half = angles, second half = phases.  Short calls and the all-pass modes
        // (SAM-L / SAM-U) take the sequential kernel.
        // (the channels with a sideband selected follow the others in list_sam: the loop is the same, the chains come behind it)
        const int nt = n_mid >= kSamTiledMin ? n_sam : 0, nt0 = nt ? n_sam0 : 0, ntsb = nt - nt0;
        if (nt) {
            double *theta = reinterpret_cast<double *>(other), *pts = theta + buf_cap;
            const long long per = (n_mid + NT - 1) / NT;
            hipLaunchKernelGGL(pll_theta_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)nt), dim3(NT), 0, am_stream, cur, buf_cap,
                               (int)n_mid, list_sam, theta, 2 * buf_cap);
            const long long ntl = (n_mid + kSamTile - 1) / kSamTile;
            const int ngroups = (int)((ntl + 63) / 64);
            if ((long long)ngroups * 64 > pll_ends_cap) {
                QH_HIP(hipStreamSynchronize(stream));
                if (side) QH_HIP(hipStreamSynchronize(side_stream));
                drop_graphs(); epoch++;
                (void)hipFree(pll_ends); pll_ends = nullptr;
                QH_HIP(dev_alloc(&pll_ends, (size_t)nch * (size_t)ngroups * 64 * kPllEndsW));
                pll_ends_cap = (long long)ngroups * 64;
            }
            hipLaunchKernelGGL((pll_lanes_kernel<true>), dim3((unsigned)ngroups, (unsigned)nt), dim3(64), 0, am_stream, (const double *)theta,
                               2 * buf_cap, pts, 2 * buf_cap, (int)n_mid, list_sam, (const PllState *)pll_state, pll_ends, pll_ends_cap * kPllEndsW,
                               sam_pll_prm, kSamTile, kSamWarm);
            hipLaunchKernelGGL((pll_verify_kernel<true>), dim3((unsigned)nt), dim3(64), 0, am_stream, (const double *)theta, 2 * buf_cap, pts,
                               2 * buf_cap, (int)n_mid, list_sam, pll_state, pll_ends, pll_ends_cap * kPllEndsW, sam_pll_prm, kSamTile, kSamWarm,
                               pll_nfixed, pll_check_only);
            if (nt0) {
                const int G = seg_groups(n_am + nt);
                double *gs = seg_sum[0] + (size_t)n_am * kSegWaves * kSegMaxGroups * kSegSumW;       // behind the AM channels' rows
                if (G > 1) {
                    hipLaunchKernelGGL((am_detect_tiled_kernel<true, 1>), dim3((unsigned)nt0, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur,
                                       buf_cap, (int)n_mid, list_sam, levelfade, am_state, am_prm, (const double *)pts, 2 * buf_cap, gs);
                    hipLaunchKernelGGL((am_detect_tiled_kernel<true, 2>), dim3((unsigned)nt0, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur,
                                       buf_cap, (int)n_mid, list_sam, levelfade, am_state, am_prm, (const double *)pts, 2 * buf_cap, gs, am_next);
                    hipLaunchKernelGGL(commit_am_kernel, dim3((unsigned)((nt0 + 255) / 256)), dim3(256), 0, am_stream, am_state, (const AmState *)am_next, list_sam, nt0, levelfade);
                } else
                    hipLaunchKernelGGL((am_detect_tiled_kernel<true, 0>), dim3((unsigned)nt0), dim3(kSegThreads), 0, am_stream, cur, buf_cap,
                                       (int)n_mid, list_sam, levelfade, am_state, am_prm, (const double *)pts, 2 * buf_cap, (double *)nullptr);
            }
            if (ntsb) {
                const int G = seg_groups(n_am + nt), S = kSegWaves * G;
                if (int rc = set_sb_phi(n_mid, S)) return rc;
                const int *lst = list_sam + nt0;
                double *gs = seg_sum[0] + (size_t)(n_am + nt0) * kSegWaves * kSegMaxGroups * kSegSumW;
                hipLaunchKernelGGL((sam_sb_tiled_kernel<1>), dim3((unsigned)ntsb, (unsigned)(S / kSbWaves)), dim3(64 * kSbWaves), 0, am_stream, cur, buf_cap, (int)n_mid,
                                   lst, (const SamChanParam *)sam_prm, (const double *)pts, 2 * buf_cap, pll_state, sb_sum, (const double *)sb_start);
                hipLaunchKernelGGL(sam_sb_chain_kernel, dim3((unsigned)ntsb), dim3(64), 0, am_stream, (int)n_mid, S, lst, (const PllState *)pll_state,
                                   (const double *)sb_phi, (const double *)sb_sum, sb_start);
                hipLaunchKernelGGL((sam_sb_tiled_kernel<2>), dim3((unsigned)ntsb, (unsigned)(S / kSbWaves)), dim3(64 * kSbWaves), 0, am_stream, cur, buf_cap, (int)n_mid,
                                   lst, (const SamChanParam *)sam_prm, (const double *)pts, 2 * buf_cap, pll_state, sb_sum, (const double *)sb_start);
                hipLaunchKernelGGL((sam_level_tiled_kernel<1>), dim3((unsigned)ntsb, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur, buf_cap, (int)n_mid,
                                   lst, levelfade, (const AmState *)am_state, am_prm, gs, am_next);
                hipLaunchKernelGGL((sam_level_tiled_kernel<2>), dim3((unsigned)ntsb, (unsigned)G), dim3(kSegThreads), 0, am_stream, cur, buf_cap, (int)n_mid,
                                   lst, levelfade, (const AmState *)am_state, am_prm, gs, am_next);
                hipLaunchKernelGGL(commit_am_kernel, dim3((unsigned)((ntsb + 255) / 256)), dim3(256), 0, am_stream, am_state, (const AmState *)am_next, lst, ntsb, levelfade);
            }
        }
        if (n_sam - nt) hipLaunchKernelGGL(sam_pll_kernel, dim3((unsigned)(n_sam - nt)), dim3(64), 0, am_stream, cur, buf_cap, (int)n_mid,
                                           list_sam + nt, pll_state, sam_prm, sam_pll_prm, am_state);
    }
    if (direct && n_bp1p[0]) {      // bp1 is the AM / SAM channels' last stage: it follows their detectors on the second stream
        std::swap(stream, side_stream);
        int hc = cur_bp1;
        if (am_fused) {
            if (amlv_src.a) band_amlv = &amlv_src;
            run_band(other, buf_cap, out, out_stride, epi, n_mid, mask_bp1, kBandNfftMax, hist_bp1, hc, P, list_am, n_am, false, false, 0, nullptr, 0,
                     np_am && !(dbg_forms & 16) ? pairs_am : nullptr, np_am);
            band_amlv = nullptr;
            hc = cur_bp1;
            if (n_sam) run_band(cur, buf_cap, out, out_stride, epi, n_mid, mask_bp1, kBandNfftMax, hist_bp1, hc, P, list_sam, n_sam, false, false, 0,
                                nullptr, 0, np_sam && !(dbg_forms & 16) ? pairs_sam : nullptr, np_sam);
        } else run_band(cur, buf_cap, out, out_stride, epi, n_mid, mask_bp1, kBandNfftMax, hist_bp1, hc, P, list_bp1p[0], n_bp1p[0]);
        std::swap(stream, side_stream);
    }
    if (side) QH_HIP(hipEventRecord(ev_join, side_stream));
    if (n_fm) {
        // xfmd's loop (fmd.c:151-172), time-tiled (qh_tiled.hpp): angles, then one loop per lane and tile, then dc removal + gain.
        // The FM channels' rows of `other` are free here: first half = angles, second half = loop filter output.
        const bool pair = de_real && np_fm && !band6k && !band2g && bnfft == kNfft && !(dbg_forms & 16);      // the de-emphasis stage two channels a tile
        const bool fmdc_fused = pair && long_parts[2] <= 1 && !(dbg_forms & 256);
        FmDcSrc fmdc_src{ nullptr, 0, 0 };
        {
            // fused: nbp0 left the angles in the channels' own rows (first half) and the loop output goes to the rows of `other`
            double *theta = reinterpret_cast<double *>(fm_theta_fused ? cur : other), *fil = reinterpret_cast<double *>(other) + buf_cap;
            if (fmdc_fused) fil = reinterpret_cast<double *>(cur) + buf_cap;     // (the de-emphasis stage reads it while it writes the rows of `other`)
            const long long per = (n_mid + NT - 1) / NT;
            if (!fm_theta_fused)
                hipLaunchKernelGGL(pll_theta_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)n_fm), dim3(NT), 0, stream, cur, buf_cap,
                                   (int)n_mid, list_fm, theta, 2 * buf_cap);
            static const int fm_tile_env = getenv("QH_FM_TILE") ? atoi(getenv("QH_FM_TILE")) : 0;
            // the longest tile that still gives the chip 512 wavefronts of 64 tiles: the 768-sample warm-up is 3/4 of a 256-sample
            // tile's steps and 3/11 of a 2048-sample tile's
            int fm_tile = kFmTile;
            while (fm_tile < 2048 && (long long)n_fm * n_mid / (64LL * 2 * fm_tile) >= 512) fm_tile *= 2;
            if (fm_tile_env > 0) fm_tile = fm_tile_env;
            const long long ntl = (n_mid + fm_tile - 1) / fm_tile;
            const int ngroups = (int)((ntl + 63) / 64);
            if ((long long)ngroups * 64 > pll_ends_cap) {
                QH_HIP(hipStreamSynchronize(stream));
                drop_graphs(); epoch++;
                (void)hipFree(pll_ends); pll_ends = nullptr;
                QH_HIP(dev_alloc(&pll_ends, (size_t)nch * (size_t)ngroups * 64 * kPllEndsW));
                pll_ends_cap = (long long)ngroups * 64;
            }
            // fmdc_fused: the dc removal and gain (fmd.c:169-171) do not get a pass of their own -- the loop kernels take the tile's own
            // share of the average off (local_dc), a chain over the tiles' contributions gives the average ahead of every tile, and the
            // de-emphasis stage's load takes the rest off and applies the gain (OsfirArgs::fmdc_*): 8 bytes per sample read there instead
            // of 8 read + 16 written here and 16 read there
            const bool fused_now = fmdc_fused && (fm_tile & (fm_tile - 1)) == 0 && fm_tile <= 2048;
            if (fused_now && pll_ends_cap + 1 > fm_cin_cap) {
                QH_HIP(hipStreamSynchronize(stream));
                if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
                drop_graphs(); epoch++;
                (void)hipFree(fm_cin); fm_cin = nullptr;
                QH_HIP(dev_alloc(&fm_cin, (size_t)nch * (size_t)(pll_ends_cap + 1)));
                fm_cin_cap = pll_ends_cap + 1;
            }
            hipLaunchKernelGGL((pll_lanes_kernel<false>), dim3((unsigned)ngroups, (unsigned)n_fm), dim3(64), 0, stream, (const double *)theta,
                               2 * buf_cap, fil, 2 * buf_cap, (int)n_mid, list_fm, (const PllState *)fm_pll_state, pll_ends, pll_ends_cap * kPllEndsW,
                               fm_pll_prm, fm_tile, kFmWarm, fused_now ? 1 : 0);
            hipLaunchKernelGGL((pll_verify_kernel<false>), dim3((unsigned)n_fm), dim3(64), 0, stream, (const double *)theta, 2 * buf_cap, fil,
                               2 * buf_cap, (int)n_mid, list_fm, fm_pll_state, pll_ends, pll_ends_cap * kPllEndsW, fm_pll_prm, fm_tile, kFmWarm,
                               pll_nfixed, pll_check_only, fused_now ? 1 : 0);
            if (fused_now) {
                hipLaunchKernelGGL(fm_dc_chain_kernel, dim3((unsigned)n_fm), dim3(64), 0, stream, (int)n_mid, fm_tile, list_fm, fm_pll_state, fm_pll_prm,
                                   (const double *)pll_ends, pll_ends_cap * kPllEndsW, fm_cin, fm_cin_cap);
                int sh = 0;
                while ((1 << sh) < fm_tile) sh++;
                fmdc_src = FmDcSrc{ fil, 2 * buf_cap, sh };
            } else {
                // dc removal + gain: the tiles' contributions are in `ends` already, one pass over `fil`
                const int G = seg_groups(n_fm);
                hipLaunchKernelGGL(fm_dc_tiled_kernel, dim3((unsigned)n_fm, (unsigned)G), dim3(kSegThreads), 0, stream, (const double *)fil,
                                   2 * buf_cap, cur, buf_cap, (int)n_mid, list_fm, (const PllState *)fm_pll_state, (const double *)fm_again, fm_pll_prm,
                                   (const double *)pll_ends, pll_ends_cap * kPllEndsW, fm_tile, fmdc_next);
                hipLaunchKernelGGL(commit_fmdc_kernel, dim3((unsigned)((n_fm + 255) / 256)), dim3(256), 0, stream, fm_pll_state, (const double *)fmdc_next, list_fm, n_fm);
            }
        }
        {   // de-emphasis: real taps on a real signal, two channels per tile
            if (fmdc_src.a) band_fmdc = &fmdc_src;
            run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_de, 0, hist_de, cur_de, P, list_fm, n_fm, false, false, 0, nullptr, 0,
                     pair ? pairs_fm : nullptr, np_fm);
            band_fmdc = nullptr;
        }
        run_band(other, buf_cap, cur, buf_cap, nullptr, n_mid, mask_aud, 0, hist_aud, cur_aud, P, list_fm, n_fm);   // audio filter
        tick(1);
        {
            const int G = seg_groups(n_fm);
            if (G > 1) {
                hipLaunchKernelGGL((snotch_tiled_kernel<1>), dim3((unsigned)n_fm, (unsigned)G), dim3(kSegThreads), 0, stream, cur, buf_cap, (int)n_mid,
                                   list_fm, sn_prm, sn_state, seg_sum[2]);
                hipLaunchKernelGGL((snotch_tiled_kernel<2>), dim3((unsigned)n_fm, (unsigned)G), dim3(kSegThreads), 0, stream, cur, buf_cap, (int)n_mid,
                                   list_fm, sn_prm, sn_state, seg_sum[2], direct ? out : (double2 *)nullptr, out_stride, (const EpiParam *)epi, sn_next);
                hipLaunchKernelGGL(commit_snotch_kernel, dim3((unsigned)((n_fm + 255) / 256)), dim3(256), 0, stream, sn_state, (const SnotchState *)sn_next, list_fm, n_fm,
                                   (const SnotchParam *)sn_prm);
            } else
                hipLaunchKernelGGL((snotch_tiled_kernel<0>), dim3((unsigned)n_fm), dim3(kSegThreads), 0, stream, cur, buf_cap, (int)n_mid, list_fm,
                                   sn_prm, sn_state, (double *)nullptr, direct ? out : (double2 *)nullptr, out_stride, (const EpiParam *)epi);
        }
        if (n_lim)      // detector limiter: lim_pre_gain 0.4, then its own wcpAGC (fmd.c:179-184)
            hipLaunchKernelGGL(agc_form == 1 ? wcpagc_seq_kernel : wcpagc_kernel, dim3((unsigned)n_lim), dim3(64), 0, stream, cur, buf_cap,
                               (int)n_mid, list_lim, lim_prm, lim_state, 0.4);
    }
    if (side) QH_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    if (n_snb[1]) snb_inplace(list_snb[1], n_snb[1]);       // xbpsnbain / xbpsnbaout at position 1 (RXA.c:576-577)
    if (n_snb[0] || n_snb[1]) cur_snb ^= 1;
    if (n_snba) {                                           // xsnba, RXA.c:578
        if (snba_tune_dirty) {
            QH_HIP(hipMemcpyAsync(snba_tune, snba_tune_h.data(), snba_tune_h.size() * sizeof(SnbaTune), hipMemcpyHostToDevice, stream));
            QH_HIP(hipStreamSynchronize(stream));
            snba_tune_dirty = false;
        }
        hipLaunchKernelGGL(snba_kernel, dim3((unsigned)n_snba), dim3(64), 0, stream, cur, buf_cap, nblk, dsp_size, list_snba, snba_prm,
                           snba_hin, snba_hout, snba_state, snba_idx, snba_scratch, (const SnbaTune *)snba_tune);
    }
    // xanf, xanr, xbandpass(bp1) at position 0, xwcpagc, then the same three at position 1 (RXA.c:579-586).  The two bp1
    // launches work on disjoint channel rows of one ping-pong history pair, so the pair flips once for both.
    auto lms_on = [&](int k, double2 *b) {
        for (int f = 0; f < 2; f++)
            if (n_lms[f][k]) hipLaunchKernelGGL(lms_kernel, dim3((unsigned)n_lms[f][k]), dim3(64), 0, stream, b, buf_cap, (int)n_mid,
                                                list_lms[f][k], lms_prm[f], lms_state[f]);
        if (n_emnr[k])          // xemnr follows xanf and xanr at either position (RXA.c:581,585)
            hipLaunchKernelGGL(emnr_kernel, dim3((unsigned)n_emnr[k]), dim3(NT), (size_t)emnr_lds_bytes(), stream, b, buf_cap, nblk, list_emnr[k],
                               emnr_prm, emnr_chan, emnr_scal, emnr_state, emnr_window, tw4096, emnr_GG, emnr_GGS, emnr_zeta, emnr_zeta_true);
    };
    auto bp1_at = [&](int ps) {
        int hc = cur_bp1;
        if (n_bp1p[ps] && !direct) run_band(cur, buf_cap, other, buf_cap, nullptr, n_mid, mask_bp1, kBandNfftMax, hist_bp1, hc, P, list_bp1p[ps], n_bp1p[ps]);
    };
    lms_on(0, cur);
    bp1_at(0);
    // xwcpagc modes 1-4 (sequential per channel); mode 0 rides in the output matrix below unless a position-1 stage follows
    tick(1);
    if (n_agc_cur || n_agc_other) {
        // Long calls: the level detector in time tiles, everything around it lane-parallel (qh_agc_tiled.hpp).  Short calls (the drop-in's
        // blocks), a channel whose attack window moved in mid-stream, and the diagnostic forms: one wavefront per channel.
        bool tiled = (agc_form == 0 || agc_form == 3) && n_mid >= kAgcTiledMin;      // (3: diagnostics, the tiles' check counts and repairs nothing)
        int a_max = 0;
        for (int ch = 0; ch < nch && tiled; ch++) {
            ChanCfg &c = cfg[(size_t)ch];
            if (!c.agc_on() || c.agc_stale) continue;
            a_max = c.agc_abuf > a_max ? c.agc_abuf : a_max;
        }
        // the tiles take the channels at the head of each list, the stepping kernel the ones behind them (all of them in a short call)
        const int nt_cur = tiled ? n_agc_cur - n_agc_cur_stale : 0, nt_other = tiled ? n_agc_other - n_agc_other_stale : 0;
        if (nt_cur + nt_other == 0) tiled = false;
        agc_last_tiled = nt_cur + nt_other;
        for (ChanCfg &c : cfg) if (c.agc_on()) c.agc_ran = true;
        // the reference's full ring (RB_SIZE entries): this call's last inputs go in where xwcpagc writes them; a channel whose attack
        // window moved since its last call first takes its 2048-entry ring again from it (the entries the longer window jumped over)
        if (!agc_lring) {
            QH_HIP(hipStreamSynchronize(stream));
            drop_graphs(); epoch++;
            QH_HIP(dev_alloc(&agc_lring, (size_t)nch * kAgcLongRing));
            QH_HIP(dev_alloc(&agc_labs, (size_t)nch * kAgcLongRing));
            QH_HIP(dev_alloc(&agc_lout, (size_t)nch));
            QH_HIP(dev_alloc(&agc_rewin_list, (size_t)nch));
            QH_HIP(hipMemsetAsync(agc_lring, 0, (size_t)nch * kAgcLongRing * sizeof(double2), stream));
            QH_HIP(hipMemsetAsync(agc_labs, 0, (size_t)nch * kAgcLongRing * sizeof(double), stream));
            QH_HIP(hipMemsetAsync(agc_lout, 0xff, (size_t)nch * sizeof(int), stream));          // out_index = -1 (calc_wcpagc, wcpAGC.c:34)
            dev_bytes += (long long)nch * kAgcLongRing * 24;
        }
        {
            std::vector<int> rw;
            for (int ch = 0; ch < nch; ch++) {
                ChanCfg &c = cfg[(size_t)ch];
                if (c.agc_on() && c.agc_rewindow) rw.push_back(ch);
                if (c.agc_on()) c.agc_rewindow = false;
            }
            if (!rw.empty()) {
                QH_HIP(hipMemcpyAsync(agc_rewin_list, rw.data(), rw.size() * sizeof(int), hipMemcpyHostToDevice, stream));
                QH_HIP(hipStreamSynchronize(stream));
                hipLaunchKernelGGL(agc_rewindow_kernel, dim3((unsigned)rw.size()), dim3(256), 0, stream, (const int *)agc_rewin_list, agc_state,
                                   (const double2 *)agc_lring, (const double *)agc_labs, (const int *)agc_lout);
            }
        }
        {
            const long long span = n_mid < kAgcLongRing ? n_mid : kAgcLongRing;
            const unsigned gx = (unsigned)((span + 255) / 256 < 120 ? (span + 255) / 256 : 120);
            auto mirror = [&](const double2 *b, const int *lst, int cnt) {
                if (!cnt) return;
                hipLaunchKernelGGL(agc_long_mirror_kernel, dim3(gx, (unsigned)cnt), dim3(256), 0, stream, b, buf_cap, (int)n_mid, lst, (const AgcParam *)agc_prm,
                                   agc_lring, agc_labs, (const int *)agc_lout);
                hipLaunchKernelGGL(agc_long_advance_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, (int)n_mid, lst, cnt, agc_lout);
            };
            mirror(cur, list_agc_cur, n_agc_cur);
            mirror(other, list_agc_other, n_agc_other);
        }
        // ... and when every channel has the AGC as its last stage (nothing at position 1, no meters, squelch or audio frames), the gain
        // multiply applies the output matrix and writes the caller's rows: the output pass goes
        bool no_p1 = !n_bp1p[1] && !n_fix[0] && !n_fix[1] && !n_emnr[1] && !n_emnr[2] && !n_amsq && !meters_on && !eg.kind;
        for (int f = 0; f < 2; f++) for (int k = 1; k < 3; k++) no_p1 = no_p1 && !n_lms[f][k];
        agc_direct = tiled && no_p1 && nt_cur == n_plain && nt_other == n_bp1;
        if (tiled) {
            const int nl = nt_cur > nt_other ? nt_cur : nt_other;
            const int ntile = (int)((n_mid + kAgcTile - 1) / kAgcTile), hp = (a_max + 15) & ~15;
            // tiles short enough for one to two wavefronts of 64 tiles per SIMD
            int L = 256;
            while (L < 16384 && (long long)nl * n_mid / (64LL * 2 * L) >= 1024) L *= 2;
            if (const char *e = getenv("QH_AGC_TILE")) { const int v = atoi(e); if (v > 0) L = (v + 63) / 64 * 64; }
            const long long nt_l = (n_mid + L - 1) / L, ngroups = (nt_l + 63) / 64;
            if (n_mid > agc_arr || ngroups * 64 > agc_ends_cap || (long long)ntile * hp > agc_halo_cap) {
                QH_HIP(hipStreamSynchronize(stream));
                if (side_stream) QH_HIP(hipStreamSynchronize(side_stream));
                drop_graphs(); epoch++;
                (void)hipFree(agc_scr); (void)hipFree(agc_ends); (void)hipFree(agc_halo); (void)hipFree(agc_tsum);
                agc_scr = agc_ends = agc_tsum = nullptr; agc_halo = nullptr;
                agc_arr = n_mid > agc_arr ? n_mid : agc_arr;
                agc_ends_cap = ngroups * 64 > agc_ends_cap ? ngroups * 64 : agc_ends_cap;
                agc_halo_cap = (long long)ntile * hp > agc_halo_cap ? (long long)ntile * hp : agc_halo_cap;
                QH_HIP(dev_alloc(&agc_scr, (size_t)nch * 4 * (size_t)agc_arr));
                QH_HIP(dev_alloc(&agc_ends, (size_t)nch * (size_t)agc_ends_cap * kAgcEndsW * 2));        // boundary states, then end states
                QH_HIP(dev_alloc(&agc_halo, (size_t)nch * (size_t)agc_halo_cap));
                QH_HIP(dev_alloc(&agc_tsum, (size_t)nch * (size_t)((agc_arr + kAgcTile - 1) / kAgcTile) * 2));
                if (!agc_fin) {
                    QH_HIP(dev_alloc(&agc_fin, (size_t)nch * 8));
                    QH_HIP(dev_alloc(&agc_tail, (size_t)nch * kAgcRing));
                    QH_HIP(dev_alloc(&agc_nfixed, (size_t)2));
                    QH_HIP(hipMemsetAsync(agc_nfixed, 0, 2 * sizeof(int), stream));
                    QH_HIP(dev_alloc(&agc_sege, (size_t)2 * nch * kAgcSegs * 8));        // two copies: a repair round reads one and writes the other
                }
            }
            static bool agc_attr = false;
            if (!agc_attr) {        // attack windows of up to kAgcRing samples: more dynamic LDS than the default limit
                QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(agc_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           3 * (kAgcRing + kAgcTile) * (int)sizeof(double)));
                QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(agc_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (kAgcRing + kAgcTile) * (int)sizeof(double2)));
                agc_attr = true;
            }
            auto run = [&](double2 *b, const int *lst, int cnt) {
                if (!cnt) return;
                const int n = (int)n_mid;
                int G = 256 / cnt;
                while (G > 1 && n_mid / (64LL * kSegWaves * G) < 8) G--;
                G = G < 1 ? 1 : G > kSegMaxGroups ? kSegMaxGroups : G;
                const size_t lds_prep = (size_t)3 * (((size_t)a_max + 63) / 64 * 64 + kAgcTile) * sizeof(double);
                const size_t lds_apply = ((size_t)a_max + kAgcTile) * sizeof(double2);
                hipLaunchKernelGGL(agc_prep_kernel, dim3((unsigned)ntile, (unsigned)cnt), dim3(256), lds_prep, stream, (const double2 *)b, buf_cap, n,
                                   lst, (const AgcParam *)agc_prm, (const AgcState *)agc_state, agc_scr, agc_arr, agc_halo, hp, 1.0, agc_tsum);
                hipLaunchKernelGGL(agc_avg_tiled_kernel, dim3((unsigned)cnt, (unsigned)G), dim3(kSegThreads), 0, stream, (const double2 *)b,
                                   buf_cap, n, lst, (const AgcParam *)agc_prm, (const AgcState *)agc_state, agc_scr, agc_arr, (const double *)agc_tsum,
                                   ntile, 1.0);
                double *bnd = agc_ends, *end = agc_ends + (size_t)nch * (size_t)agc_ends_cap * kAgcEndsW;
                // the boundary pass over K super-segments per channel at once (a multiple of the tile length each)
                int K = 1;
                while (K < kAgcSegs && (long long)cnt * K < 4096 && n_mid / (2 * K) >= 8 * L && n_mid / (2 * K) >= 32768) K *= 2;
                if (const char *e = getenv("QH_AGC_SEGS")) { const int v = atoi(e); if (v >= 1 && v <= kAgcSegs) K = v; }
                const int seg = (int)(((n_mid + K - 1) / K + L - 1) / L) * L;
                // No warm-up ahead of a segment by default: every segment starts from the state the call began in, and the repair rounds
                // walk each one again from the end of the one before it until the two walks meet (agc_bounds_round_kernel) -- the work a
                // warm-up long enough for every channel (400 attack windows on the bench input) spends on all of them, spent only where
                // and for as long as the walks differ.
                int wmul = 0, rounds = 3;
                if (const char *e = getenv("QH_AGC_WARM")) { const int v = atoi(e); if (v >= 0) wmul = v; }
                if (const char *e = getenv("QH_AGC_ROUNDS")) { const int v = atoi(e); if (v >= 0 && v <= 16) rounds = v; }
                const int Wm = ((wmul * a_max + L - 1) / L) * L;
                double *sg[2] = { agc_sege, agc_sege + (size_t)nch * kAgcSegs * 8 };
                hipLaunchKernelGGL(agc_bounds_kernel, dim3((unsigned)cnt, (unsigned)K), dim3(64), 0, stream, n, lst, (const AgcParam *)agc_prm,
                                   (const AgcState *)agc_state, (const double *)agc_scr, agc_arr, bnd, agc_ends_cap * kAgcEndsW, L, seg, Wm, sg[0]);
                if (K > 1) {
                    int at = 0;
                    for (int r = 0; r < rounds; r++, at ^= 1)
                        hipLaunchKernelGGL(agc_bounds_round_kernel, dim3((unsigned)cnt, (unsigned)K), dim3(64), 0, stream, n, lst,
                                           (const AgcParam *)agc_prm, (const double *)agc_scr, agc_arr, bnd, agc_ends_cap * kAgcEndsW, L, seg, K,
                                           (const double *)sg[at], sg[at ^ 1], agc_nfixed + 1);
                    hipLaunchKernelGGL(agc_bounds_fix_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, n, lst, (const AgcParam *)agc_prm,
                                       (const double *)agc_scr, agc_arr, bnd, agc_ends_cap * kAgcEndsW, L, seg, K, sg[at], agc_nfixed + 1);
                }
                hipLaunchKernelGGL(agc_lanes_kernel, dim3((unsigned)ngroups, (unsigned)cnt), dim3(64), 0, stream, n, lst, (const AgcParam *)agc_prm,
                                   agc_scr, agc_arr, (const double *)bnd, agc_ends_cap * kAgcEndsW, end, agc_ends_cap * kAgcEndsW, L);
                hipLaunchKernelGGL(agc_verify_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, n, lst, (const AgcParam *)agc_prm, agc_scr, agc_arr,
                                   (const double *)bnd, agc_ends_cap * kAgcEndsW, end, agc_ends_cap * kAgcEndsW, L, agc_fin, agc_nfixed, agc_form == 3 ? 1 : 0);
                hipLaunchKernelGGL(agc_tail_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, (const double2 *)b, buf_cap, n, lst,
                                   (const AgcParam *)agc_prm, agc_tail, 1.0);
                hipLaunchKernelGGL(agc_apply_kernel, dim3((unsigned)ntile, (unsigned)cnt), dim3(256), lds_apply, stream, b, buf_cap, n, lst,
                                   (const AgcParam *)agc_prm, (const double *)agc_scr, agc_arr, (const double2 *)agc_halo, hp, 1.0,
                                   agc_direct ? out : (double2 *)nullptr, out_stride, (const EpiParam *)epi);
                hipLaunchKernelGGL(agc_finish_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, n, lst, (const AgcParam *)agc_prm, agc_state,
                                   (const double *)agc_scr, agc_arr, (const double2 *)agc_tail, (const double *)agc_fin);
            };
            run(cur, list_agc_cur, nt_cur);
            run(other, list_agc_other, nt_other);
        }
        if (const int ns = n_agc_cur - nt_cur)
            hipLaunchKernelGGL(agc_form == 1 ? wcpagc_seq_kernel : wcpagc_kernel, dim3((unsigned)ns), dim3(64), 0, stream, cur, buf_cap, (int)n_mid,
                               (const int *)(list_agc_cur + nt_cur), agc_prm, agc_state, 1.0);
        if (const int ns = n_agc_other - nt_other)
            hipLaunchKernelGGL(agc_form == 1 ? wcpagc_seq_kernel : wcpagc_kernel, dim3((unsigned)ns), dim3(64), 0, stream, other, buf_cap, (int)n_mid,
                               (const int *)(list_agc_other + nt_other), agc_prm, agc_state, 1.0);
    }
    {
        long long per = (n_mid + NT - 1) / NT;
        const unsigned gx = (unsigned)(per < 1024 ? per : 1024);
        for (int b = 0; b < 2; b++)
            if (n_fix[b]) hipLaunchKernelGGL(scale_kernel, dim3(gx, (unsigned)n_fix[b]), dim3(NT), 0, stream, b ? other : cur, buf_cap,
                                             (int)n_mid, list_fix[b], fix_gain);
    }
    lms_on(1, cur);
    lms_on(2, other);
    bp1_at(1);
    if (n_bp1) cur_bp1 ^= 1;
    if (meters_on) {    // agcmeter sits after xwcpagc (RXA.c:589); mode 0's gain multiply is applied below, so its
                        // level reading is taken on the fixed-gain input times g^2 (m_g2)
        if (n_plain) hipLaunchKernelGGL(meter_kernel, dim3((unsigned)n_plain), dim3(64), 0, stream, cur, buf_cap, nblk, dsp_size,
                                        m_agc, m_prm, list_plain, (const double *)m_g2);
        if (n_bp1) hipLaunchKernelGGL(meter_kernel, dim3((unsigned)n_bp1), dim3(64), 0, stream, other, buf_cap, nblk, dsp_size,
                                      m_agc, m_prm, list_bp1, (const double *)m_g2);
    }
    tick(2);
    // xwcpagc mode 0 + xpanel
    long long per = (n_mid + NT - 1) / NT;
    const unsigned gx = (unsigned)(per < 1024 ? per : 1024);
    if (direct || agc_direct) {
        // every channel's last stage has written the caller's buffer
    } else if (eg_fused) {
        if (n_plain) hipLaunchKernelGGL((pointwise_kernel<double, false, true>), dim3(gx, (unsigned)n_plain), dim3(NT), 0, stream, cur,
                                        buf_cap, out, out_stride, (int)n_mid, (const unsigned long long *)nullptr,
                                        (const unsigned long long *)nullptr, epi, list_plain, eg);
        if (n_bp1) hipLaunchKernelGGL((pointwise_kernel<double, false, true>), dim3(gx, (unsigned)n_bp1), dim3(NT), 0, stream, other,
                                      buf_cap, out, out_stride, (int)n_mid, (const unsigned long long *)nullptr,
                                      (const unsigned long long *)nullptr, epi, list_bp1, eg);
    } else {
    if (n_plain) hipLaunchKernelGGL((pointwise_kernel<double, false>), dim3(gx, (unsigned)n_plain), dim3(NT), 0, stream, cur,
                                    buf_cap, out, out_stride, (int)n_mid, (const unsigned long long *)nullptr,
                                    (const unsigned long long *)nullptr, epi, list_plain);
    if (n_bp1) hipLaunchKernelGGL((pointwise_kernel<double, false>), dim3(gx, (unsigned)n_bp1), dim3(NT), 0, stream, other,
                                  buf_cap, out, out_stride, (int)n_mid, (const unsigned long long *)nullptr,
                                  (const unsigned long long *)nullptr, epi, list_bp1);
    }
    if (n_amsq) hipLaunchKernelGGL(amsq_apply_kernel, dim3((unsigned)n_amsq), dim3(64), 0, stream, out, out_stride, (int)n_mid, list_amsq,
                                   amsq_mag, amsq_mag_cap, amsq_prm, amsq_state, amsq_cup, amsq_cdown);       // xamsq, RXA.c:596
    if (eg.kind && !eg_fused) pack_audio(out, out_stride, n_mid);
    tick(3);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

}  // namespace qh

// ------------------------------------------------------------------------------------------ C ABI
using namespace qh;

// One lock per engine: setters may come from another thread than the one that runs the blocks (Quisk's GUI thread against its
// sound thread; WDSP's setters take csDSP).  A setter only edits the host-side configuration and marks it dirty; the next
// process call uploads what changed before it enqueues the block, so parameters swap on a block boundary.
struct qh_rxa { Engine e; std::recursive_mutex mtx; };
#define QH_RXA_LOCK(h) std::lock_guard<std::recursive_mutex> _lk((h)->mtx)

extern "C" {

int qh_version(void) { return 100; }
const char *qh_last_error(void) { return g_last_error.c_str(); }

int qh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

qh_rxa *qh_rxa_create(int device, int nch, int dsp_size, int in_rate, int dsp_rate, int out_rate, void *stream)
{
    if (nch <= 0 || dsp_size <= 0 || (dsp_size & (dsp_size - 1)) || in_rate <= 0 || dsp_rate <= 0) {
        set_error(QH_ERR_INVALID, "qh_rxa_create: bad arguments");
        return nullptr;
    }
    if (out_rate <= 0 || (out_rate % dsp_rate && dsp_rate % out_rate) || (out_rate < dsp_rate && dsp_size % (dsp_rate / out_rate))) {
        set_error(QH_ERR_UNSUPPORTED, "out_rate must be an integer multiple or fraction of dsp_rate (wdsp/channel.c:47-52)");
        return nullptr;
    }
    // in_rate / dsp_rate 1, 2, 4, 8, 16: the overlap-save front stage.  Any other whole ratio, up or down (3, 5, 6 ...; 1/2,
    // 1/4 ...): xshift as a pointwise pass + the polyphase form of xresample (wdsp/resample.c:35-157) -- D = 0 marks it.  The
    // reference sizes its blocks with integer divisions of the two rates (pre_main_build, wdsp/channel.c:39-42): a ratio that is
    // not whole one way or the other does not give it consistent block sizes, and is refused here.
    int D = (in_rate % dsp_rate) ? 0 : in_rate / dsp_rate;
    if (D != 1 && D != 2 && D != 4 && D != 8 && D != 16) D = 0;
    if (D == 0 && !((in_rate > dsp_rate && in_rate % dsp_rate == 0) ||
                    (in_rate < dsp_rate && dsp_rate % in_rate == 0 && dsp_size % (dsp_rate / in_rate) == 0))) {
        // (a CPU test of the reference's arithmetic, test_rates_that_are_whole_in_neither_direction_recycle_stale_buffer_tails, shows what it
        // does with 96 k -> 64 k -> 48 k: every block's tail is what the block before left in the buffer)
        set_error(QH_ERR_UNSUPPORTED, "in_rate / dsp_rate must be a whole number or the reciprocal of one (wdsp/channel.c:39-42)");
        return nullptr;
    }
    if (qh_device_count() <= device || device < 0) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_rxa *h = new qh_rxa();
    h->e.device = device; h->e.nch = nch; h->e.dsp_size = dsp_size;
    h->e.in_rate = in_rate; h->e.dsp_rate = dsp_rate; h->e.out_rate = out_rate; h->e.D = D;
    h->e.stream = (hipStream_t)stream;
    if (h->e.init() != QH_OK) { delete h; return nullptr; }
    return h;
}

void qh_rxa_destroy(qh_rxa *h) { delete h; }
int qh_rxa_nch(const qh_rxa *h) { return h->e.nch; }
int qh_rxa_dsp_insize(const qh_rxa *h) { return h->e.dsp_insize; }
int qh_rxa_dsp_outsize(const qh_rxa *h) { return h->e.dsp_outsize; }
void *qh_rxa_stream(const qh_rxa *h) { return h ? (void *)h->e.stream : nullptr; }
long long qh_rxa_device_bytes(const qh_rxa *h) { return h->e.dev_bytes; }

#define FOR_CH(h, ch, body)                                                                       \
    do {                                                                                          \
        if (!(h)) return set_error(QH_ERR_INVALID, "null engine");                                \
        QH_RXA_LOCK(h);                                                                           \
        if ((ch) < -1 || (ch) >= (h)->e.nch) return set_error(QH_ERR_INVALID, "channel %d out of range", (ch)); \
        int _lo = (ch) < 0 ? 0 : (ch), _hi = (ch) < 0 ? (h)->e.nch : (ch) + 1;                    \
        (h)->e.epoch++;                                                                           \
        for (int _i = _lo; _i < _hi; _i++) { ChanCfg &c = (h)->e.cfg[(size_t)_i]; body }         \
        return QH_OK;                                                                             \
    } while (0)

// RXAbp1Check + RXAbp1Set, wdsp/RXA.c:800-827 (snba/emnr/anf/anr never run here)
static void bp1_check_set(ChanCfg &c, int amd_run, int anf_run, int anr_run, int emnr_run = -1, int snba_run = -1)
{
    if (emnr_run < 0) emnr_run = c.emnr_run;
    if (snba_run < 0) snba_run = c.snba_run;
    const double gain = (amd_run || anf_run || anr_run || emnr_run || snba_run) ? 2.0 : 1.0;
    if (c.bp1_gain != gain) { c.bp1_gain = gain; c.bp1_dirty = true; }
}
static void bp1_set(ChanCfg &c)
{
    const int old = c.bp1_run;
    c.bp1_run = (c.amd_run || c.lms[0].run || c.lms[1].run || c.emnr_run || c.snba_run) ? 1 : 0;
    if (old != c.bp1_run) c.bp1_dirty = true;
    if (!old && c.bp1_run) c.bp1_flush = true;
}

int qh_rxa_SetRXAMode(qh_rxa *h, int ch, int mode)
{
    FOR_CH(h, ch, {
        if (c.mode != mode) {       // wdsp/RXA.c:748-787
            const int amd_run = (mode == QH_AM) || (mode == QH_SAM);
            bp1_check_set(c, amd_run, c.lms[0].run, c.lms[1].run);
            c.mode = mode;
            c.amd_run = 0; c.fmd_run = 0; c.agc_run = 1;
            if (mode == QH_AM) { c.amd_run = 1; c.amd_mode = 0; }
            else if (mode == QH_SAM) { c.amd_run = 1; c.amd_mode = 1; }
            else if (mode == QH_FM) { c.fmd_run = 1; c.agc_run = 0; }
            bp1_set(c);
            c.snb_dirty = true;
            c.epi_dirty = true;
            h->e.lists_dirty = true;
        }
    });
}

// SetRXAAMDRun (wdsp/amd.c:264-277): the AM demodulator's run flag on its own (SetRXAMode sets it from the mode)
int qh_rxa_SetRXAAMDRun(qh_rxa *h, int ch, int run)
{
    FOR_CH(h, ch, {
        run = run ? 1 : 0;
        if (c.amd_run != run) {
            bp1_check_set(c, run, c.lms[0].run, c.lms[1].run);
            c.amd_run = run;
            bp1_set(c);
            c.epi_dirty = true;
            h->e.lists_dirty = true;
        }
    });
}

int qh_rxa_SetRXABandpassFreqs(qh_rxa *h, int ch, double f_low, double f_high)
{
    FOR_CH(h, ch, {
        if (f_low != c.bp1_flow || f_high != c.bp1_fhigh) { c.bp1_flow = f_low; c.bp1_fhigh = f_high; c.bp1_dirty = true; }
    });
}

int qh_rxa_RXANBPSetFreqs(qh_rxa *h, int ch, double flow, double fhigh)
{
    FOR_CH(h, ch, {
        if (flow != c.nbp_flow || fhigh != c.nbp_fhigh) { c.nbp_flow = flow; c.nbp_fhigh = fhigh; c.nbp_dirty = true; }
    });
}

// SetRXASNBAOutputBandwidth, wdsp/snb.c:660-694: the pass band of the blanker's 12 kHz -> dsp_rate resampler
int qh_rxa_SetRXASNBAOutputBandwidth(qh_rxa *h, int ch, double flow, double fhigh)
{
    FOR_CH(h, ch, {
        const double lc = 200.0;        // out_low_cut / out_high_cut, RXA.c:254-255
        const double hc = 5400.0;
        double lo = flow;
        double hi = fhigh;
        double f_low = c.snba_f_low;
        double f_high = c.snba_f_high;
        if (lo >= 0 && hi >= 0) {
            if (hi < lc) hi = lc;
            if (lo > hc) lo = hc;
            f_low = lc > lo ? lc : lo;
            f_high = hc < hi ? hc : hi;
        } else if (lo <= 0 && hi <= 0) {
            if (lo > -lc) lo = -lc;
            if (hi < -hc) hi = -hc;
            f_low = lc > -hi ? lc : -hi;
            f_high = hc < -lo ? hc : -lo;
        } else if (lo < 0 && hi > 0) {
            double absmax = -lo > hi ? -lo : hi;
            if (absmax < lc) absmax = lc;
            f_low = lc;
            f_high = hc < absmax ? hc : absmax;
        }
        if (f_low != c.snba_f_low || f_high != c.snba_f_high) {     // setBandwidth_resample rebuilds the filter and clears its ring
            c.snba_f_low = f_low; c.snba_f_high = f_high;
            c.snba_taps_dirty = true; c.snba_rout_flush = true;
            c.epi_dirty = true;
        }
    });
}

// The blanker's tuning setters, wdsp/snb.c:604-658.  They act on the next block, as under csDSP.
static int snba_tune_set(qh_rxa *h, int ch, const char *who, bool ok, void (*apply)(SnbaTune &, double), double v)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    if (!ok) return set_error(QH_ERR_INVALID, "%s: value out of range", who);
    QH_RXA_LOCK(h);
    if (ch < -1 || ch >= h->e.nch) return set_error(QH_ERR_INVALID, "channel out of range");
    for (int c = ch < 0 ? 0 : ch; c < (ch < 0 ? h->e.nch : ch + 1); c++) apply(h->e.snba_tune_h[(size_t)c], v);
    h->e.snba_tune_dirty = true;
    h->e.drop_graphs(); h->e.epoch++;
    return QH_OK;
}
int qh_rxa_SetRXASNBAasize(qh_rxa *h, int ch, int size)
{ return snba_tune_set(h, ch, "SetRXASNBAasize (1 .. 64)", size >= 1 && size <= 64, [](SnbaTune &t, double v) { t.asize = (int)v; }, size); }
int qh_rxa_SetRXASNBAnpasses(qh_rxa *h, int ch, int npasses)
{ return snba_tune_set(h, ch, "SetRXASNBAnpasses (0 .. 8)", npasses >= 0 && npasses <= 8, [](SnbaTune &t, double v) { t.npasses = (int)v; }, npasses); }
int qh_rxa_SetRXASNBAk1(qh_rxa *h, int ch, double k1)
{ return snba_tune_set(h, ch, "SetRXASNBAk1", k1 > 0.0, [](SnbaTune &t, double v) { t.k1 = v; }, k1); }
int qh_rxa_SetRXASNBAk2(qh_rxa *h, int ch, double k2)
{ return snba_tune_set(h, ch, "SetRXASNBAk2", k2 > 0.0, [](SnbaTune &t, double v) { t.k2 = v; }, k2); }
int qh_rxa_SetRXASNBAbridge(qh_rxa *h, int ch, int bridge)
{ return snba_tune_set(h, ch, "SetRXASNBAbridge (0 .. 64)", bridge >= 0 && bridge <= 64, [](SnbaTune &t, double v) { t.b = (int)v; }, bridge); }
int qh_rxa_SetRXASNBApresamps(qh_rxa *h, int ch, int presamps)
{ return snba_tune_set(h, ch, "SetRXASNBApresamps (0 .. 64)", presamps >= 0 && presamps <= 64, [](SnbaTune &t, double v) { t.pre = (int)v; }, presamps); }
int qh_rxa_SetRXASNBApostsamps(qh_rxa *h, int ch, int postsamps)
{ return snba_tune_set(h, ch, "SetRXASNBApostsamps (0 .. 64)", postsamps >= 0 && postsamps <= 64, [](SnbaTune &t, double v) { t.post = (int)v; }, postsamps); }
int qh_rxa_SetRXASNBApmultmin(qh_rxa *h, int ch, double pmultmin)
{ return snba_tune_set(h, ch, "SetRXASNBApmultmin", pmultmin >= 0.0, [](SnbaTune &t, double v) { t.pmultmin = v; }, pmultmin); }

// SetRXASNBAovrlp, wdsp/snb.c:595-603.  The frame advance sizes the blanker's state, which the engine lays out once for all its
// channels: ch = -1 (or the only channel).  The WDSP-named layer keeps one engine per channel, so there it is per channel as in WDSP.
int qh_rxa_SetRXASNBAovrlp(qh_rxa *h, int ch, int ovrlp)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!(ch == -1 || (ch == 0 && h->e.nch == 1)))
        return set_error(QH_ERR_UNSUPPORTED, "SetRXASNBAovrlp re-plans the blanker's accumulators for the whole engine: pass channel -1");
    if (int rc = h->e.snba_set_ovrlp(ovrlp)) return rc;
    // calc_snba makes the output resampler anew with its creation arguments: fc_low 200, the default cut-off (snb.c:45-46) -- what
    // SetRXASNBAOutputBandwidth had set is gone, as in WDSP
    for (ChanCfg &c : h->e.cfg) { c.snba_f_low = 200.0; c.snba_f_high = 0.0; c.snba_taps_dirty = true; }
    return QH_OK;
}

// SetRXASNBARun, wdsp/snb.c:579-593
int qh_rxa_SetRXASNBARun(qh_rxa *h, int ch, int run)
{
    FOR_CH(h, ch, {
        run = run ? 1 : 0;
        if (c.snba_run != run) {
            bp1_check_set(c, c.amd_run, c.lms[0].run, c.lms[1].run, -1, run);
            c.snba_run = run;
            bp1_set(c);
            c.snb_dirty = true;
            c.epi_dirty = true;
            h->e.lists_dirty = true;
        }
    });
}

int qh_rxa_RXASetPassband(qh_rxa *h, int ch, double f_low, double f_high)
{
    int rc = qh_rxa_SetRXABandpassFreqs(h, ch, f_low, f_high);
    if (rc) return rc;
    if ((rc = qh_rxa_SetRXASNBAOutputBandwidth(h, ch, f_low, f_high))) return rc;
    return qh_rxa_RXANBPSetFreqs(h, ch, f_low, f_high);
}

int qh_rxa_RXASetNC(qh_rxa *h, int ch, int nc)
{
    if (nc < 1 || (nc & (nc - 1)) || nc > kLongNcMax || (h && nc < h->e.dsp_size))
        return set_error(QH_ERR_UNSUPPORTED, "nc must be a power of two in [dsp_size, %d]", kLongNcMax);
    FOR_CH(h, ch, {
        if (c.nbp_nc != nc) { c.nbp_nc = nc; c.nbp_dirty = true; c.nbp_flush = true; c.snb_flush = true; c.long_live[0] = c.long_live[4] = false; }
        if (c.bp1_nc != nc) { c.bp1_nc = nc; c.bp1_dirty = true; c.bp1_flush = true; c.long_live[1] = false; }
        if (c.fm_nc != nc) c.long_live[2] = false;     // (setNc_fircore zeroes the delay lines, firmin.c:454-466: nothing long is held any more)
        c.fm_nc = nc;                           // SetRXAFMNCde / SetRXAFMNCaud, wdsp/RXA.c:942-943
    });
}

static Notch mk_notch(double fcenter, double fwidth, int active)
{
    Notch n;
    n.fcenter = fcenter; n.fwidth = fwidth; n.active = active;
    return n;
}

// ---- the notch database (wdsp/nbp.c:358-525).  Return values of Add / Delete / Edit / Get follow the reference:
// 0, or -1 for an index out of range (reported through *rval; the function result stays the library's status).
int qh_rxa_RXANBPAddNotch(qh_rxa *h, int ch, int notch, double fcenter, double fwidth, int active, int *rval)
{
    if (rval) *rval = -1;
    FOR_CH(h, ch, {
        if (notch >= 0 && notch <= (int)c.notches.size() && c.notches.size() < 1024) {
            c.notches.insert(c.notches.begin() + notch, mk_notch(fcenter, fwidth, active));
            if (c.fnfrun) c.nbp_dirty = true;
            if (rval) *rval = 0;
        } else if (rval) *rval = -1;
    });
}

int qh_rxa_RXANBPDeleteNotch(qh_rxa *h, int ch, int notch, int *rval)
{
    if (rval) *rval = -1;
    FOR_CH(h, ch, {
        if (notch >= 0 && notch < (int)c.notches.size()) {
            c.notches.erase(c.notches.begin() + notch);
            if (c.fnfrun) c.nbp_dirty = true;
            if (rval) *rval = 0;
        } else if (rval) *rval = -1;
    });
}

int qh_rxa_RXANBPEditNotch(qh_rxa *h, int ch, int notch, double fcenter, double fwidth, int active, int *rval)
{
    if (rval) *rval = -1;
    FOR_CH(h, ch, {
        if (notch >= 0 && notch < (int)c.notches.size()) {
            c.notches[(size_t)notch] = mk_notch(fcenter, fwidth, active);
            if (c.fnfrun) c.nbp_dirty = true;
            if (rval) *rval = 0;
        } else if (rval) *rval = -1;
    });
}

int qh_rxa_RXANBPGetNotch(qh_rxa *h, int ch, int notch, double *fcenter, double *fwidth, int *active, int *rval)
{
    if (!h || ch < 0 || ch >= h->e.nch || !fcenter || !fwidth || !active) return set_error(QH_ERR_INVALID, "RXANBPGetNotch: bad arguments");
    const ChanCfg &c = h->e.cfg[(size_t)ch];
    if (notch >= 0 && notch < (int)c.notches.size()) {
        *fcenter = c.notches[(size_t)notch].fcenter; *fwidth = c.notches[(size_t)notch].fwidth; *active = c.notches[(size_t)notch].active;
        if (rval) *rval = 0;
    } else {
        *fcenter = -1.0; *fwidth = 0.0; *active = -1;
        if (rval) *rval = -1;
    }
    return QH_OK;
}

int qh_rxa_RXANBPGetNumNotches(qh_rxa *h, int ch, int *nnotches)
{
    if (!h || ch < 0 || ch >= h->e.nch || !nnotches) return set_error(QH_ERR_INVALID, "RXANBPGetNumNotches: bad arguments");
    *nnotches = (int)h->e.cfg[(size_t)ch].notches.size();
    return QH_OK;
}

int qh_rxa_RXANBPGetMinNotchWidth(qh_rxa *h, int ch, double *minwidth)
{
    if (!h || ch < 0 || ch >= h->e.nch || !minwidth) return set_error(QH_ERR_INVALID, "RXANBPGetMinNotchWidth: bad arguments");
    const ChanCfg &c = h->e.cfg[(size_t)ch];
    *minwidth = (c.nbp_wintype == 1 ? 2200.0 : 1600.0) / (c.nbp_nc / 256) * ((double)h->e.dsp_rate / 48000);      // nbp.c:82-95
    return QH_OK;
}

int qh_rxa_RXANBPSetTuneFrequency(qh_rxa *h, int ch, double f) { FOR_CH(h, ch, { if (f != c.ndb_tunefreq) { c.ndb_tunefreq = f; if (c.fnfrun) c.nbp_dirty = true; } }); }
int qh_rxa_RXANBPSetShiftFrequency(qh_rxa *h, int ch, double f) { FOR_CH(h, ch, { if (f != c.ndb_shift) { c.ndb_shift = f; if (c.fnfrun) c.nbp_dirty = true; } }); }
int qh_rxa_RXANBPSetNotchesRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { run = run ? 1 : 0; if (run != c.fnfrun) { c.fnfrun = run; c.nbp_dirty = true; } }); }
int qh_rxa_RXANBPSetWindow(qh_rxa *h, int ch, int wintype) { FOR_CH(h, ch, { if (c.nbp_wintype != wintype) { c.nbp_wintype = wintype; c.nbp_dirty = true; } }); }
int qh_rxa_RXANBPSetAutoIncrease(qh_rxa *h, int ch, int autoincr) { FOR_CH(h, ch, { if (c.autoincr != autoincr) { c.autoincr = autoincr; if (c.fnfrun) c.nbp_dirty = true; } }); }

// RXASetMP (wdsp/RXA.c:948-958): minimum-phase impulse responses in every fircore of the chain.  nbp0 and bp1 have
// per-channel masks; the FM de-emphasis / audio masks are shared by the channels of an engine and follow the
// most recent call.
int qh_rxa_RXASetMP(qh_rxa *h, int ch, int mp)
{
    mp = mp ? 1 : 0;
    FOR_CH(h, ch, {
        if (c.mp != mp) { c.mp = mp; c.nbp_dirty = true; c.bp1_dirty = true; c.demod_dirty = true; }
    });
}

int qh_rxa_SetRXAShiftRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { c.shift_run = run; c.nco_dirty = true; }); }
int qh_rxa_SetRXAShiftFreq(qh_rxa *h, int ch, double f) { FOR_CH(h, ch, { c.shift_freq = f; c.nco_dirty = true; }); }
int qh_rxa_RXANBPSetRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { if (c.nbp_run != run) { c.nbp_run = run; c.nbp_dirty = true; } }); }
// (bandpass.c:385-390 writes the flag and nothing else; where a fixed AGC gain is applied -- at the AGC's own spot or in the output matrix --
// depends on it: fix_before)
int qh_rxa_SetRXABandpassRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { if (c.bp1_run != run) { c.bp1_run = run; c.bp1_dirty = true; c.epi_dirty = true; h->e.lists_dirty = true; } }); }
int qh_rxa_SetRXAAMDSBMode(qh_rxa *h, int ch, int sbmode) { FOR_CH(h, ch, { c.sbmode = sbmode; c.demod_dirty = true; }); }
int qh_rxa_SetRXAAMDFadeLevel(qh_rxa *h, int ch, int levelfade) { FOR_CH(h, ch, { c.levelfade = levelfade; c.demod_dirty = true; }); }
int qh_rxa_SetRXAFMDeviation(qh_rxa *h, int ch, double deviation) { FOR_CH(h, ch, { c.fm_dev = deviation; c.demod_dirty = true; }); }
int qh_rxa_SetRXACTCSSFreq(qh_rxa *h, int ch, double freq) { FOR_CH(h, ch, { c.ctcss_freq = freq; c.demod_dirty = true; c.ctcss_flush = true; }); }
int qh_rxa_SetRXACTCSSRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { c.ctcss_run = run; c.demod_dirty = true; }); }
// SetRXAFMLimRun / SetRXAFMLimGain (wdsp/fmd.c:336-362): the FM detector's limiter
int qh_rxa_SetRXAFMLimRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { run = run ? 1 : 0; if (c.lim_run != run) { c.lim_run = run; h->e.lists_dirty = true; } }); }
int qh_rxa_SetRXAFMLimGain(qh_rxa *h, int ch, double gaindB)
{
    const double gain = std::pow(10.0, gaindB / 20.0);
    FOR_CH(h, ch, { if (c.lim_gain != gain) { c.lim_gain = gain; c.lim_dirty = true; } });
}

// SetRXAEMNRRun ... SetRXAEMNRPosition, wdsp/emnr.c:1096-1143
int qh_rxa_SetRXAEMNRRun(qh_rxa *h, int ch, int run)
{
    if (h && run && !h->e.emnr_tables)
        return set_error(QH_ERR_INVALID, "EMNR needs its gain tables first (qh_rxa_SetEMNRTables: WDSP's `calculus` and `zetaHat.bin` data)");
    if (h && run && h->e.dsp_size > kEmnrIncr) return set_error(QH_ERR_UNSUPPORTED, "EMNR: dsp_size up to %d", kEmnrIncr);
    FOR_CH(h, ch, {
        run = run ? 1 : 0;
        if (c.emnr_run != run) {
            bp1_check_set(c, c.amd_run, c.lms[0].run, c.lms[1].run, run);
            c.emnr_run = run;
            bp1_set(c);
            c.epi_dirty = true;
            h->e.lists_dirty = true;
        }
    });
}
int qh_rxa_SetRXAEMNRgainMethod(qh_rxa *h, int ch, int method) { FOR_CH(h, ch, { c.emnr_gain_method = method; c.emnr_dirty = true; }); }
int qh_rxa_SetRXAEMNRnpeMethod(qh_rxa *h, int ch, int method) { FOR_CH(h, ch, { c.emnr_npe = method; c.emnr_dirty = true; }); }
int qh_rxa_SetRXAEMNRaeRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { c.emnr_ae = run ? 1 : 0; c.emnr_dirty = true; }); }
int qh_rxa_SetRXAEMNRaeZetaThresh(qh_rxa *h, int ch, double v) { FOR_CH(h, ch, { c.emnr_ae_zeta = v; c.emnr_dirty = true; }); }       // emnr.c:1145
int qh_rxa_SetRXAEMNRaePsi(qh_rxa *h, int ch, double v) { FOR_CH(h, ch, { c.emnr_ae_psi = v; c.emnr_dirty = true; }); }               // emnr.c:1153
int qh_rxa_SetRXAEMNRtrainZetaThresh(qh_rxa *h, int ch, double v) { FOR_CH(h, ch, { c.emnr_train_zeta = v; c.emnr_dirty = true; }); }  // emnr.c:1161
int qh_rxa_SetRXAEMNRtrainT2(qh_rxa *h, int ch, double v) { FOR_CH(h, ch, { c.emnr_train_t2 = v; c.emnr_dirty = true; }); }            // emnr.c:1169
int qh_rxa_SetRXAEMNRPosition(qh_rxa *h, int ch, int position)
{
    FOR_CH(h, ch, { c.emnr_pos = position ? 1 : 0; c.bp1_pos = position ? 1 : 0; c.epi_dirty = true; h->e.lists_dirty = true; });
}
// The data WDSP reads at create time from the files `calculus` (GG, GGS: 241 x 241 each) and `zetaHat.bin` (60 x 60 values, validity
// flags and their gamma / xi ranges in dB), emnr.c:206-238,317-334
int qh_rxa_SetEMNRTables(qh_rxa *h, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_true, double gamma_min,
                         double gamma_max, double xi_min, double xi_max)
{
    if (!h || !GG || !GGS || !zeta_hat || !zeta_true) return set_error(QH_ERR_INVALID, "qh_rxa_SetEMNRTables: null table");
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    e.epoch++;
    e.h_GG.assign(GG, GG + 241 * 241); e.h_GGS.assign(GGS, GGS + 241 * 241);
    e.h_zeta.assign(zeta_hat, zeta_hat + 3600); e.h_zeta_true.assign(zeta_true, zeta_true + 3600);
    e.h_zrange[0] = gamma_min; e.h_zrange[1] = gamma_max; e.h_zrange[2] = xi_min; e.h_zrange[3] = xi_max;
    e.emnr_tables = true;
    if (e.emnr_GG) {            // already on the device: refresh
        QH_HIP(hipSetDevice(e.device));
        QH_HIP(hipMemcpyAsync(e.emnr_GG, GG, 241 * 241 * 8, hipMemcpyHostToDevice, e.stream));
        QH_HIP(hipMemcpyAsync(e.emnr_GGS, GGS, 241 * 241 * 8, hipMemcpyHostToDevice, e.stream));
        QH_HIP(hipMemcpyAsync(e.emnr_zeta, zeta_hat, 3600 * 8, hipMemcpyHostToDevice, e.stream));
        QH_HIP(hipMemcpyAsync(e.emnr_zeta_true, zeta_true, 3600 * 4, hipMemcpyHostToDevice, e.stream));
        QH_HIP(hipStreamSynchronize(e.stream));
        e.emnr_prm.z_gamma_min = gamma_min; e.emnr_prm.z_gamma_max = gamma_max; e.emnr_prm.z_xihat_min = xi_min; e.emnr_prm.z_xihat_max = xi_max;
    }
    return QH_OK;
}

// SetRXAANFRun ... SetRXAANFPosition (wdsp/anf.c:175-239) and the ANR twins (wdsp/anr.c:175-238); which = 0 anf, 1 anr
static int lms_run(qh_rxa *h, int ch, int which, int run)
{
    FOR_CH(h, ch, {
        run = run ? 1 : 0;
        ChanCfg::Lms &m = c.lms[which];
        if (m.run != run) {
            bp1_check_set(c, c.amd_run, which == 0 ? run : c.lms[0].run, which == 1 ? run : c.lms[1].run);
            m.run = run;
            bp1_set(c);
            m.flush = true;
            c.lms[0].dirty = c.lms[1].dirty = true; c.epi_dirty = true;
            h->e.lists_dirty = true;
        }
    });
}
static int lms_vals(qh_rxa *h, int ch, int which, const int *taps, const int *delay, const double *gain, const double *leakage)
{
    FOR_CH(h, ch, {
        ChanCfg::Lms &m = c.lms[which];
        if (taps) m.taps = *taps;
        if (delay) m.delay = *delay;
        if (gain) m.two_mu = *gain;
        if (leakage) m.gamma = *leakage;
        m.flush = true; m.dirty = true;
    });
}
static int lms_position(qh_rxa *h, int ch, int which, int position)
{
    FOR_CH(h, ch, {
        c.lms[which].position = position ? 1 : 0;
        c.bp1_pos = position ? 1 : 0;                 // "rxa[channel].bp1.p->position = position", anf.c:236
        c.lms[which].flush = true;
        c.lms[0].dirty = c.lms[1].dirty = true; c.epi_dirty = true;
        h->e.lists_dirty = true;
    });
}
// SetRXAAMSQRun / Threshold / MaxTail, wdsp/amsq.c:216-243
int qh_rxa_SetRXAAMSQRun(qh_rxa *h, int ch, int run) { FOR_CH(h, ch, { c.amsq_run = run ? 1 : 0; h->e.lists_dirty = true; }); }
int qh_rxa_SetRXAAMSQThreshold(qh_rxa *h, int ch, double threshold)
{
    FOR_CH(h, ch, { const double t = std::pow(10.0, threshold / 20.0); c.amsq_tail_thresh = 0.9 * t; c.amsq_unmute_thresh = t; c.amsq_dirty = true; });
}
int qh_rxa_SetRXAAMSQMaxTail(qh_rxa *h, int ch, double tail) { FOR_CH(h, ch, { c.amsq_max_tail = tail < 0.0 ? 0.0 : tail; c.amsq_dirty = true; }); }
int qh_rxa_SetRXAANFRun(qh_rxa *h, int ch, int run) { return lms_run(h, ch, 0, run); }
int qh_rxa_SetRXAANRRun(qh_rxa *h, int ch, int run) { return lms_run(h, ch, 1, run); }
int qh_rxa_SetRXAANFVals(qh_rxa *h, int ch, int taps, int delay, double gain, double leakage) { return lms_vals(h, ch, 0, &taps, &delay, &gain, &leakage); }
int qh_rxa_SetRXAANRVals(qh_rxa *h, int ch, int taps, int delay, double gain, double leakage) { return lms_vals(h, ch, 1, &taps, &delay, &gain, &leakage); }
int qh_rxa_SetRXAANFTaps(qh_rxa *h, int ch, int taps) { return lms_vals(h, ch, 0, &taps, nullptr, nullptr, nullptr); }
int qh_rxa_SetRXAANRTaps(qh_rxa *h, int ch, int taps) { return lms_vals(h, ch, 1, &taps, nullptr, nullptr, nullptr); }
int qh_rxa_SetRXAANFDelay(qh_rxa *h, int ch, int delay) { return lms_vals(h, ch, 0, nullptr, &delay, nullptr, nullptr); }
int qh_rxa_SetRXAANRDelay(qh_rxa *h, int ch, int delay) { return lms_vals(h, ch, 1, nullptr, &delay, nullptr, nullptr); }
int qh_rxa_SetRXAANFGain(qh_rxa *h, int ch, double gain) { return lms_vals(h, ch, 0, nullptr, nullptr, &gain, nullptr); }
int qh_rxa_SetRXAANRGain(qh_rxa *h, int ch, double gain) { return lms_vals(h, ch, 1, nullptr, nullptr, &gain, nullptr); }
int qh_rxa_SetRXAANFLeakage(qh_rxa *h, int ch, double leakage) { return lms_vals(h, ch, 0, nullptr, nullptr, nullptr, &leakage); }
int qh_rxa_SetRXAANRLeakage(qh_rxa *h, int ch, double leakage) { return lms_vals(h, ch, 1, nullptr, nullptr, nullptr, &leakage); }
int qh_rxa_SetRXAANFPosition(qh_rxa *h, int ch, int position) { return lms_position(h, ch, 0, position); }
int qh_rxa_SetRXAANRPosition(qh_rxa *h, int ch, int position) { return lms_position(h, ch, 1, position); }

int qh_rxa_SetRXAAGCMode(qh_rxa *h, int ch, int mode)
{
    FOR_CH(h, ch, {                 // wdsp/wcpAGC.c:369-411
        switch (mode) {
        case 0: c.agc_mode = 0; break;
        case 1: c.agc_mode = 1; c.agc_hangtime = 2.000; c.agc_tau_decay = 2.000; break;
        case 2: c.agc_mode = 2; c.agc_hangtime = 1.000; c.agc_tau_decay = 0.500; break;
        case 3: c.agc_mode = 3; c.agc_hang_thresh = 1.0; c.agc_hangtime = 0.000; c.agc_tau_decay = 0.250; break;
        case 4: c.agc_mode = 4; c.agc_hang_thresh = 1.0; c.agc_hangtime = 0.000; c.agc_tau_decay = 0.050; break;
        default: c.agc_mode = 5; break;
        }
        c.epi_dirty = true; c.agc_dirty = true; c.lms[0].dirty = c.lms[1].dirty = true; h->e.lists_dirty = true;
    });
}
int qh_rxa_SetRXAAGCAttack(qh_rxa *h, int ch, int attack_ms) { FOR_CH(h, ch, { c.agc_tau_attack = (double)attack_ms / 1000.0; c.agc_dirty = true; }); }
int qh_rxa_SetRXAAGCDecay(qh_rxa *h, int ch, int decay_ms) { FOR_CH(h, ch, { c.agc_tau_decay = (double)decay_ms / 1000.0; c.agc_dirty = true; }); }
int qh_rxa_SetRXAAGCHang(qh_rxa *h, int ch, int hang_ms) { FOR_CH(h, ch, { c.agc_hangtime = (double)hang_ms / 1000.0; c.agc_dirty = true; }); }
int qh_rxa_SetRXAAGCTop(qh_rxa *h, int ch, double max_agc_db) { FOR_CH(h, ch, { c.agc_max_gain = std::pow(10.0, max_agc_db / 20.0); c.agc_dirty = true; }); }
int qh_rxa_SetRXAAGCSlope(qh_rxa *h, int ch, int slope) { FOR_CH(h, ch, { c.agc_var_gain = std::pow(10.0, (double)slope / 20.0 / 10.0); c.agc_dirty = true; }); }
int qh_rxa_SetRXAAGCHangThreshold(qh_rxa *h, int ch, int t) { FOR_CH(h, ch, { c.agc_hang_thresh = (double)t / 100.0; c.agc_dirty = true; }); }

int qh_rxa_SetRXAAGCFixed(qh_rxa *h, int ch, double db)
{
    FOR_CH(h, ch, { c.agc_fixed = std::pow(10.0, db / 20.0); c.epi_dirty = true; c.lms[0].dirty = c.lms[1].dirty = true; });
}

int qh_rxa_SetRXAPanelGain1(qh_rxa *h, int ch, double g) { FOR_CH(h, ch, { c.gain1 = g; c.epi_dirty = true; }); }
int qh_rxa_SetRXAPanelGain2(qh_rxa *h, int ch, double gI, double gQ) { FOR_CH(h, ch, { c.gain2I = gI; c.gain2Q = gQ; c.epi_dirty = true; }); }
int qh_rxa_SetRXAPanelSelect(qh_rxa *h, int ch, int s) { FOR_CH(h, ch, { c.inselect = s; c.epi_dirty = true; }); }
int qh_rxa_SetRXAPanelCopy(qh_rxa *h, int ch, int cp) { FOR_CH(h, ch, { c.copy = cp; c.epi_dirty = true; }); }

int qh_rxa_process(qh_rxa *h, const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!d_in || !d_out) return set_error(QH_ERR_INVALID, "null buffer");
    if (in_stride < (long long)nblk * h->e.dsp_insize || out_stride < (long long)nblk * h->e.dsp_outsize)
        return set_error(QH_ERR_INVALID, "stride shorter than nblk blocks");
    if (h->e.graph_on) return h->e.process_replayed(d_in, in_stride, d_out, out_stride, nblk);
    return h->e.process(d_in, in_stride, d_out, out_stride, nblk);
}

// ---- audio egress ---------------------------------------------------------------------------------------------------
static int make_egress(const qh_audio_format *fmt, void *d_out, long long out_stride_bytes, long long frames, EgressFmt *f)
{
    if (!fmt || !d_out) return set_error(QH_ERR_INVALID, "null audio format or buffer");
    if (fmt->kind < QH_AUDIO_I16 || fmt->kind > QH_AUDIO_F32) return set_error(QH_ERR_INVALID, "audio kind %d", fmt->kind);
    if (fmt->num_channels < 1 || fmt->channel_I < 0 || fmt->channel_Q < 0 || fmt->channel_I >= fmt->num_channels ||
        fmt->channel_Q >= fmt->num_channels)
        return set_error(QH_ERR_INVALID, "audio channel slots outside the frame");
    const int bytes = fmt->kind == QH_AUDIO_I16 ? 2 : fmt->kind == QH_AUDIO_I24 ? 3 : 4;
    if (out_stride_bytes < frames * fmt->num_channels * bytes) return set_error(QH_ERR_INVALID, "audio row stride shorter than the frames");
    if (fmt->kind != QH_AUDIO_I24 && out_stride_bytes % bytes) return set_error(QH_ERR_INVALID, "audio row stride not a multiple of the sample size");
    f->kind = fmt->kind; f->nchan = fmt->num_channels; f->ch_i = fmt->channel_I; f->ch_q = fmt->channel_Q;
    f->volume = fmt->volume; f->prescale = fmt->prescale == 0.0 ? 1.0 : fmt->prescale;
    f->out = static_cast<unsigned char *>(d_out); f->stride = out_stride_bytes;
    return QH_OK;
}

int qh_rxa_process_audio(qh_rxa *h, const double *d_in, long long in_stride, void *d_out, long long out_stride_bytes, int nblk,
                         const qh_audio_format *fmt)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!d_in) return set_error(QH_ERR_INVALID, "null buffer");
    if (in_stride < (long long)nblk * h->e.dsp_insize) return set_error(QH_ERR_INVALID, "stride shorter than nblk blocks");
    EgressFmt f{};
    if (int rc = make_egress(fmt, d_out, out_stride_bytes, (long long)nblk * h->e.dsp_outsize, &f)) return rc;
    h->e.eg = f;
    const int rc = h->e.process(d_in, in_stride, nullptr, 0, nblk);
    h->e.eg = EgressFmt{};
    return rc;
}

int qh_audio_pack(int device, void *stream, const double *d_src, long long src_stride, int nch, int n, const qh_audio_format *fmt,
                  void *d_dst, long long dst_stride_bytes)
{
    if (!d_src || nch <= 0 || n < 0) return set_error(QH_ERR_INVALID, "qh_audio_pack: bad arguments");
    if (qh_device_count() <= device || device < 0) return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    EgressFmt f{};
    if (int rc = make_egress(fmt, d_dst, dst_stride_bytes, n, &f)) return rc;
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(device));
    const long long per = ((long long)n + 255) / 256;
    hipLaunchKernelGGL(egress_pack_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)nch), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const double2 *>(d_src), src_stride, n, f);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_rxa_set_graph_replay(qh_rxa *h, int on)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    h->e.graph_on = on != 0;
    if (!on) { h->e.drop_graphs(); h->e.graph_key = Engine::GraphKey{}; }
    return QH_OK;
}
long long qh_rxa_graph_launches(const qh_rxa *h) { return h ? h->e.graph_launches : 0; }

int qh_rxa_set_band_tile(qh_rxa *h, int nfft)
{
    if (!h || (nfft != 0 && nfft != 4096 && nfft != 6144 && nfft != 8192)) return set_error(QH_ERR_INVALID, "qh_rxa_set_band_tile: 0 (default), 4096, 6144 or 8192");
    QH_RXA_LOCK(h);
    h->e.band_tile_pref = nfft;
    return QH_OK;
}
int qh_rxa_band_tile(const qh_rxa *h) { return h ? h->e.bnfft : 0; }

int Engine::process_replayed(const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk)
{
    // the resamplers (the output one; the input one of the rate ratios that are not 1, 2, 4, 8 or 16) keep a host-side phase and their
    // own ping-pong of delay lines per call, event timing records per call: all of them stay on the plain path.  (The input resampler was
    // missing here until a seeded walk through the WDSP names at 144 ksps found it: a sequence captured after a setter replayed the
    // other parity of its delay lines, tests/test_gpu_wdsp_names_fuzz.py.)
    if (rsmpin || rsmpout || timing || nblk <= 0) return process(d_in, in_stride, d_out, out_stride, nblk);
    const GraphKey key{d_in, d_out, in_stride, out_stride, nblk, epoch};
    if (!(key == graph_key)) { drop_graphs(); graph_key = key; graph_seen = false; }
    const unsigned before = flags();
    GraphSlot &slot = graph_slot[before];
    if (slot.exec) {
        QH_HIP(hipSetDevice(device));
        QH_HIP(hipGraphLaunch(slot.exec, stream));
        set_flags(slot.after);
        graph_launches++;
        return QH_OK;
    }
    if (!graph_seen) {
        // the first call under a new key uploads dirty parameters and grows buffers (synchronising): not capturable
        const int rc = process(d_in, in_stride, d_out, out_stride, nblk);
        graph_seen = rc == QH_OK;
        graph_key.epoch = epoch;        // buffers that this call grew are in place now
        return rc;
    }
    QH_HIP(hipSetDevice(device));
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        graph_on = false;
        return process(d_in, in_stride, d_out, out_stride, nblk);
    }
    const int rc = process(d_in, in_stride, d_out, out_stride, nblk);
    hipGraph_t g = nullptr;
    const hipError_t e_end = hipStreamEndCapture(stream, &g);
    bool ok = rc == QH_OK && e_end == hipSuccess && g && hipGraphInstantiate(&slot.exec, g, nullptr, nullptr, 0) == hipSuccess;
    if (g) (void)hipGraphDestroy(g);
    if (ok) {
        slot.after = flags();
        QH_HIP(hipGraphLaunch(slot.exec, stream));
        graph_launches++;
        return QH_OK;
    }
    // nothing ran: put the flags back, stop capturing for this engine and run the block the plain way
    (void)hipGetLastError();
    slot.exec = nullptr;
    set_flags(before);
    graph_on = false;
    return process(d_in, in_stride, d_out, out_stride, nblk);
}

// The same chain fed with wire-format samples: the front kernel decodes them in its load.
int qh_rxa_process_packed(qh_rxa *h, const void *d_src, long long src_bytes, const qh_iq_format *fmt, long long chan_stride,
                          double *d_out, long long out_stride, int nblk)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!d_src || !d_out || !fmt) return set_error(QH_ERR_INVALID, "null buffer");
    if (out_stride < (long long)nblk * h->e.dsp_outsize) return set_error(QH_ERR_INVALID, "stride shorter than nblk blocks");
    PackedFmt pk;
    if (int rc = qh::make_packed_fmt(fmt, chan_stride, src_bytes, (long long)nblk * h->e.dsp_insize, h->e.nch, &pk)) return rc;
    h->e.pk_src = static_cast<const unsigned char *>(d_src);
    h->e.pk = pk;
    // process() wants an input pointer; the packed kernels never touch it
    const int rc = h->e.process(reinterpret_cast<const double *>(d_src), (long long)nblk * h->e.dsp_insize, d_out, out_stride, nblk);
    h->e.pk_src = nullptr;
    return rc;
}

// flush_rxa (wdsp/RXA.c:527-559): NCO phase, resampler ring and fircore delay lines back to zero
int qh_rxa_flush(qh_rxa *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    e.epoch++;
    QH_HIP(hipSetDevice(e.device));
    QH_HIP(hipMemsetAsync(e.nco_phase, 0, (size_t)e.nch * sizeof(unsigned long long), e.stream));
    QH_HIP(hipMemsetAsync(e.nco_parked, 0, (size_t)e.nch * sizeof(unsigned long long), e.stream));
    if (e.rsmpout) if (int rc = qh_rat_reset(e.rsmpout)) return rc;        // flush_resample, wdsp/resample.c:159-165
    if (e.rsmpin) if (int rc = qh_rat_reset(e.rsmpin)) return rc;
    for (int i = 0; i < 2; i++) {
        if (e.hist_front[i]) QH_HIP(hipMemsetAsync(e.hist_front[i], 0, (size_t)e.nch * kHistFront * sizeof(double2), e.stream));
        QH_HIP(hipMemsetAsync(e.hist_nbp[i], 0, (size_t)e.nch * kHistBand * sizeof(double2), e.stream));
        QH_HIP(hipMemsetAsync(e.hist_bp1[i], 0, (size_t)e.nch * kHistBand * sizeof(double2), e.stream));
        if (e.demod_alloc) {
            QH_HIP(hipMemsetAsync(e.hist_de[i], 0, (size_t)e.nch * kHistBand * sizeof(double2), e.stream));
            QH_HIP(hipMemsetAsync(e.hist_aud[i], 0, (size_t)e.nch * kHistBand * sizeof(double2), e.stream));
        }
        for (int sid = 0; sid < 5; sid++)
            if (e.lhist[sid][i]) QH_HIP(hipMemsetAsync(e.lhist[sid][i], 0, (size_t)e.nch * kLongHist * sizeof(double2), e.stream));
    }
    if (e.demod_alloc) {                        // flush_wcpagc zeroes the ring (wcpAGC.c:154-159)
        for (int c = 0; c < e.nch; c++) {
            QH_HIP(hipMemsetAsync(e.agc_state[c].ring, 0, sizeof(e.agc_state[c].ring), e.stream));
            QH_HIP(hipMemsetAsync(e.agc_state[c].abs_ring, 0, sizeof(e.agc_state[c].abs_ring), e.stream));
            QH_HIP(hipMemsetAsync(&e.agc_state[c].ring_max, 0, sizeof(double), e.stream));
            if (e.cfg[(size_t)c].agc_stale) e.lists_dirty = true;
            e.cfg[(size_t)c].agc_stale = false;     // an empty ring and ring_max = 0: nothing stale (qh_agc_tiled.hpp)
        }
        if (e.agc_lring) {
            QH_HIP(hipMemsetAsync(e.agc_lring, 0, (size_t)e.nch * kAgcLongRing * sizeof(double2), e.stream));
            QH_HIP(hipMemsetAsync(e.agc_labs, 0, (size_t)e.nch * kAgcLongRing * sizeof(double), e.stream));
        }
    }
    for (ChanCfg &c : e.cfg) { c.lms[0].flush = c.lms[1].flush = true; c.emnr_flush = true; c.snba_flush = true; c.snb_flush = true; }    // flush_anf / flush_anr / flush_emnr, RXA.c:541-543
    if (e.amsq_state) QH_HIP(hipMemsetAsync(e.amsq_state, 0, (size_t)e.nch * sizeof(AmsqState), e.stream));     // flush_amsq
    if (e.demod_alloc) {                        // flush_amd / flush_fmd / flush_snotch
        QH_HIP(hipMemsetAsync(e.am_state, 0, (size_t)e.nch * sizeof(AmState), e.stream));
        QH_HIP(hipMemsetAsync(e.pll_state, 0, (size_t)e.nch * sizeof(PllState), e.stream));
        QH_HIP(hipMemsetAsync(e.fm_pll_state, 0, (size_t)e.nch * sizeof(PllState), e.stream));
        QH_HIP(hipMemsetAsync(e.sn_state, 0, (size_t)e.nch * sizeof(SnotchState), e.stream));
    }
    return QH_OK;
}

// Meters (wdsp/meter.c): enable != 0 makes later process calls maintain the ADC, S and AGC meters.
int qh_rxa_enable_meters(qh_rxa *h, int enable)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    h->e.meters_on = enable != 0;
    h->e.epoch++;                   // the meter launches join / leave the sequence
    return QH_OK;
}

// GetRXAMeter (wdsp/meter.c:133-142); mt as wdsp/RXA.h:47-57: 0 S_PK, 1 S_AV, 2 ADC_PK, 3 ADC_AV, 4 AGC_GAIN, 5 AGC_PK, 6 AGC_AV
int qh_rxa_GetRXAMeter(qh_rxa *h, int ch, int mt, double *value)
{
    if (!h || !value) return set_error(QH_ERR_INVALID, "null argument");
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    if (ch < 0 || ch >= e.nch || mt < 0 || mt > 6) return set_error(QH_ERR_INVALID, "channel or meter index out of range");
    if (!e.meters_on || !e.m_adc) { *value = -400.0; return QH_OK; }             // flush_meter's initial reading
    QH_HIP(hipSetDevice(e.device));
    QH_HIP(hipStreamSynchronize(e.stream));
    MeterState st;
    const MeterState *src = mt <= 1 ? e.m_s : mt <= 3 ? e.m_adc : e.m_agc;
    QH_HIP(hipMemcpy(&st, src + ch, sizeof(st), hipMemcpyDeviceToHost));
    if (mt == 4) {
        double g = 0.0;
        // RXA_AGC_GAIN: xwcpagc's `gain` (wcpAGC.c:334), which only the modes 1-5 write; 0 from create_wcpagc's calloc until then
        if (e.agc_state) QH_HIP(hipMemcpy(&g, &e.agc_state[ch].gain, sizeof(double), hipMemcpyDeviceToHost));
        const double v = g + 1.0e-40;
        unsigned long long N; std::memcpy(&N, &v, 8);
        const int ex = (int)((N >> 52) & 2047) - 1023, m = (int)((N >> 41) & 2047);
        *value = 20.0 * 0.301029995663981 * ((double)ex + std::log2(1.0 + (double)m / 2048.0));
        return QH_OK;
    }
    *value = (mt == 0 || mt == 2 || mt == 5) ? st.res_pk : st.res_av;
    return QH_OK;
}

// Diagnostics of the time-tiled FM loop: tiles whose speculative warm-up had not converged and were re-run in order.
long long qh_rxa_pll_repairs(qh_rxa *h)
{
    if (!h || !h->e.pll_nfixed) return 0;
    int v = 0;
    if (hipSetDevice(h->e.device) != hipSuccess || hipStreamSynchronize(h->e.stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, h->e.pll_nfixed, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

// Diagnostics: check_only >= 0 sets the verify pass to count-only (1) or repair (0); then copies up to `max` doubles of
// channel ch's per-tile loop states of the last call ([tile][kPllEndsW]: pt, fil_out, omega where the warm-up ended / the tile ended).
int qh_rxa_debug_pll(qh_rxa *h, int check_only, int ch, double *out, int max)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    if (check_only >= 0) e.pll_check_only = check_only;
    if (!out || max <= 0 || !e.pll_ends) return 0;
    QH_HIP(hipSetDevice(e.device));
    QH_HIP(hipStreamSynchronize(e.stream));
    long long n = e.pll_ends_cap * kPllEndsW;
    if (n > max) n = max;
    QH_HIP(hipMemcpy(out, e.pll_ends + (long long)ch * e.pll_ends_cap * kPllEndsW, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return (int)n;
}

// Diagnostics: the lanes' states of the last time-tiled wcpAGC call, list slot `slot`: [tile][kAgcEndsW]
int qh_rxa_debug_agc_ends(qh_rxa *h, int slot, double *out, int max)
{
    if (!h || !out || max <= 0 || !h->e.agc_ends) return 0;
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    QH_HIP(hipSetDevice(e.device));
    QH_HIP(hipStreamSynchronize(e.stream));
    // [tile][8] boundary states (what the run-jumping pass found), then [tile][8] the states the exact tiles ended in
    long long n = e.agc_ends_cap * kAgcEndsW;
    if (2 * n > max) n = max / 2;
    QH_HIP(hipMemcpy(out, e.agc_ends + (long long)slot * e.agc_ends_cap * kAgcEndsW, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    QH_HIP(hipMemcpy(out + n, e.agc_ends + ((long long)e.nch + slot) * e.agc_ends_cap * kAgcEndsW, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    n *= 2;
    return (int)n;
}

// tiles of the time-tiled wcpAGC that the verify pass re-ran in order, over all calls so far
long long qh_rxa_agc_repairs(qh_rxa *h)
{
    if (!h || !h->e.agc_nfixed) return 0;
    QH_RXA_LOCK(h);
    int v = 0;
    if (hipSetDevice(h->e.device) != hipSuccess || hipStreamSynchronize(h->e.stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, h->e.agc_nfixed, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

// super-segments of the AGC boundary pass that were walked again (their warm-up had not ended on the true trajectory), over all calls
#ifdef QH_AGC_COUNT
extern "C" int qh_dbg_agc_counts(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qh::g_agc_count), 20 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[20] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(qh::g_agc_count), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
long long qh_rxa_agc_segments_rerun(qh_rxa *h)
{
    if (!h || !h->e.agc_nfixed) return 0;
    QH_RXA_LOCK(h);
    int v = 0;
    if (hipSetDevice(h->e.device) != hipSuccess || hipStreamSynchronize(h->e.stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, h->e.agc_nfixed + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

// channels whose xwcpagc ran in time tiles in the last call (the others, if any: one wavefront per channel)
int qh_rxa_agc_tiled_channels(qh_rxa *h)
{
    if (!h) return 0;
    QH_RXA_LOCK(h);
    return h->e.agc_last_tiled;
}

// Diagnostics: which form of the wcpAGC loop runs (0: time tiles for long calls, else 64 samples per step of the wavefront; 1: sample by
// sample; 2: 64 samples per step whatever the call length).  Same state; 1 and 2 give the same bits, 0 the same to rounding.
int qh_rxa_debug_agc(qh_rxa *h, int form)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (h->e.agc_form != form) { h->e.agc_form = form; h->e.drop_graphs(); h->e.epoch++; }
    return QH_OK;
}

int qh_rxa_synchronize(qh_rxa *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    QH_HIP(hipSetDevice(h->e.device));
    QH_HIP(hipStreamSynchronize(h->e.stream));
    return QH_OK;
}

int qh_rxa_process_host(qh_rxa *h, const double *h_in, long long in_stride, double *h_out, long long out_stride, int nblk)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!h_in || !h_out) return set_error(QH_ERR_INVALID, "null buffer");
    Engine &e = h->e;
    QH_HIP(hipSetDevice(e.device));
    const long long n_in = (long long)nblk * e.dsp_insize, n_out = (long long)nblk * e.dsp_outsize;
    double2 *din = nullptr, *dout = nullptr;
    QH_HIP(dev_alloc(&din, (size_t)e.nch * (size_t)n_in));
    QH_HIP(dev_alloc(&dout, (size_t)e.nch * (size_t)n_out));
    int rc = QH_OK;
    hipError_t err = hipMemcpy2DAsync(din, (size_t)n_in * sizeof(double2), h_in, (size_t)in_stride * sizeof(double2),
                                      (size_t)n_in * sizeof(double2), (size_t)e.nch, hipMemcpyHostToDevice, e.stream);
    if (err == hipSuccess) {
        rc = e.process(reinterpret_cast<const double *>(din), n_in, reinterpret_cast<double *>(dout), n_out, nblk);
        if (rc == QH_OK)
            err = hipMemcpy2DAsync(h_out, (size_t)out_stride * sizeof(double2), dout, (size_t)n_out * sizeof(double2),
                                   (size_t)n_out * sizeof(double2), (size_t)e.nch, hipMemcpyDeviceToHost, e.stream);
    }
    hipError_t err2 = hipStreamSynchronize(e.stream);
    (void)hipFree(din); (void)hipFree(dout);
    if (rc != QH_OK) return rc;
    if (err != hipSuccess) return set_error(QH_ERR_HIP, "copy failed: %s", hipGetErrorString(err));
    if (err2 != hipSuccess) return set_error(QH_ERR_HIP, "synchronize failed: %s", hipGetErrorString(err2));
    return QH_OK;
}

int qh_rxa_process_packed_host(qh_rxa *h, const void *h_src, long long src_bytes, const qh_iq_format *fmt, long long chan_stride,
                               double *h_out, long long out_stride, int nblk)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    if (!h_src || !h_out || src_bytes <= 0) return set_error(QH_ERR_INVALID, "null buffer");
    Engine &e = h->e;
    QH_HIP(hipSetDevice(e.device));
    const long long n_out = (long long)nblk * e.dsp_outsize;
    unsigned char *dsrc = nullptr;
    double2 *dout = nullptr;
    QH_HIP(dev_alloc(&dsrc, (size_t)src_bytes));
    if (dev_alloc(&dout, (size_t)e.nch * (size_t)n_out) != hipSuccess) { (void)hipFree(dsrc); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    int rc = QH_OK;
    hipError_t err = hipMemcpyAsync(dsrc, h_src, (size_t)src_bytes, hipMemcpyHostToDevice, e.stream);
    if (err == hipSuccess) {
        rc = qh_rxa_process_packed(h, dsrc, src_bytes, fmt, chan_stride, reinterpret_cast<double *>(dout), n_out, nblk);
        if (rc == QH_OK)
            err = hipMemcpy2DAsync(h_out, (size_t)out_stride * sizeof(double2), dout, (size_t)n_out * sizeof(double2),
                                   (size_t)n_out * sizeof(double2), (size_t)e.nch, hipMemcpyDeviceToHost, e.stream);
    }
    hipError_t err2 = hipStreamSynchronize(e.stream);
    (void)hipFree(dsrc); (void)hipFree(dout);
    if (rc != QH_OK) return rc;
    if (err != hipSuccess) return set_error(QH_ERR_HIP, "copy failed: %s", hipGetErrorString(err));
    if (err2 != hipSuccess) return set_error(QH_ERR_HIP, "synchronize failed: %s", hipGetErrorString(err2));
    return QH_OK;
}

int qh_rxa_enable_timing(qh_rxa *h, int enable)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    h->e.timing = enable != 0;
    return QH_OK;
}

int qh_rxa_timing(qh_rxa *h, double *ms, int n)
{
    if (!h) return set_error(QH_ERR_INVALID, "null engine");
    QH_RXA_LOCK(h);
    Engine &e = h->e;
    QH_HIP(hipSetDevice(e.device));
    QH_HIP(hipStreamSynchronize(e.stream));
    double acc[3] = { 0, 0, 0 };
    for (int i = 0; i + 1 < e.ev_used; i++) {
        float t = 0;
        QH_HIP(hipEventElapsedTime(&t, e.ev[(size_t)i], e.ev[(size_t)i + 1]));
        int cat = e.ev_cat[(size_t)i];
        if (cat >= 0 && cat < 3) acc[cat] += t;
    }
    for (int k = 0; k < 3; k++) { e.last_ms[k] = acc[k]; if (k < n) ms[k] = acc[k]; }
    return n < 3 ? n : 3;
}

}  // extern "C"
