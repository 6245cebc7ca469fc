// qh_hbcascade.hip -- batched cascade of 45-tap half-band decimators in one HBM pass (include/quiskhip.h group 3b).
//
// GPU form of `nstage` consecutive quisk_cDecim2HB45 calls (filter.c:377-417) as quisk_process_decimate chains
// them (quisk.c:1772-1796: HalfBand1..5) and as BASELINE config 5 chains eight of them (61.44 Msps -> 240 ksps).
// See qh_hbcascade.hpp for the kernel.
#include <cstdlib>
#include <vector>
#include "qh_hbcascade.hpp"
#include "qh_internal.hpp"

using namespace qh;

struct qh_hbc {
    int device = 0, nch = 0, nstage = 0, dtype = QH_F64;
    int warm = 0;
    size_t esize = 16;
    void *hist[2] = { nullptr, nullptr };
    int cur = 0;
    unsigned attr_set = 0;          // bit 0 / 1: the dynamic LDS limit of the 2048 / 4096-sample-step instantiation is set
    int big = 0;                    // experiment builds (QH_EXP_HBC_BIG): -1 = calls of at least kBigMin samples take the 4096-sample steps, 0 / 1 forced
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Six and more stages run as TWO launches: the first four stages (15/16 of the arithmetic, 5 barriers per 2048-sample step) write
    // the /16 stream to `mid`, a second cascade of the remaining stages reads it.  In the one-launch form the stages behind the third
    // occupy 32 .. 8 lanes of a wavefront and still cost every wavefront a barrier each per step, and a segment's warm-up is 6 steps
    // instead of 1; the extra 2 x 1/16 of HBM traffic is small beside that (61.44 Msps fp32, 2^26 samples, tools/dbg/hbc_head.sh (a one-off script, in git history):
    // 0.178 ms as one launch, 0.182 as 3 + 5, 0.152 as 4 + 4, 0.153 as 5 + 3).
    int head = 0;                   // stages of this handle's own launch (== nstage when there is no tail)
    qh_hbc *tail = nullptr;
    void *mid = nullptr;
    long long mid_cap = 0;
    ~qh_hbc()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        delete tail;
        for (auto &h : hist) if (h) (void)hipFree(h);
        if (mid) (void)hipFree(mid);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

[[maybe_unused]] constexpr int kBigMin = 1 << 23;

template <typename T, int NS, bool BIG>
int launch(qh_hbc *h, const void *in, long long in_stride, int n_in, void *out, long long out_stride)
{
    using G = HbGeom<NS, BIG ? 4096 : 2048, BIG ? 8 : 4>;
    // segments: long enough that the warm-up is a few per cent, short enough to fill 256 CUs x 4 workgroups
    long long seg = 64LL * G::STEP;
    constexpr int kResident = BIG ? 512 : 1024;         // workgroups the chip holds at once: 256 CUs x 2 (63 KB of rings) or x 4
    while (seg > 16 * G::STEP && (long long)h->nch * ((n_in + seg - 1) / seg) < kResident) seg >>= 1;
    // a short input (the second launch of a long cascade): a workgroup's walk through its steps is a chain of barriers and LDS round
    // trips, so more and shorter segments finish sooner although each repeats the warm-up
    while (seg > 4 * G::STEP && seg > 4 * G::WARM && (long long)h->nch * ((n_in + seg - 1) / seg) < 512) seg >>= 1;
    if (const char *e = getenv("QH_HBC_SEG_STEPS")) { const int v = atoi(e); if (v > 0) seg = (long long)v * G::STEP; }
    const int nseg = (int)((n_in + seg - 1) / seg);
    const size_t lds = (size_t)G::ring_pairs() * sizeof(HbPair<T>);
    auto k = hb45_cascade_kernel<T, NS, BIG>;
    if (!(h->attr_set & (BIG ? 2u : 1u))) {     // once per handle: the dynamic LDS limit of this instantiation
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->attr_set |= BIG ? 2u : 1u;
    }
    static_assert(G::WARM <= 4096 * ((42 * ((1 << NS) - 1) + 4095) / 4096), "the handle's history rows hold either geometry's warm-up");
    const bool hist_in_kernel = n_in >= h->warm;       // nothing of the old history survives: the segments at the call's end write the new one
    hipLaunchKernelGGL(k, dim3((unsigned)nseg, (unsigned)h->nch), dim3(NT), lds, h->stream, (const cplx<T> *)in, in_stride,
                       (const cplx<T> *)h->hist[h->cur], n_in, (cplx<T> *)out, out_stride, (int)seg,
                       hist_in_kernel ? (cplx<T> *)h->hist[h->cur ^ 1] : (cplx<T> *)nullptr, h->warm);
    QH_HIP(hipGetLastError());
    if (!hist_in_kernel)
    hipLaunchKernelGGL(hb45_hist_kernel<T>, dim3((unsigned)((h->warm + NT - 1) / NT), (unsigned)h->nch), dim3(NT), 0, h->stream,
                       (const cplx<T> *)in, in_stride, n_in, (const cplx<T> *)h->hist[h->cur], (cplx<T> *)h->hist[h->cur ^ 1],
                       h->warm);
    QH_HIP(hipGetLastError());
    h->cur ^= 1;
    return QH_OK;
}

template <typename T>
int dispatch(qh_hbc *h, const void *in, long long is, int n, void *out, long long os)
{
#ifdef QH_EXP_HBC_BIG
    // Experiment builds only (tools/ab_bench.py build big:QH_EXP_HBC_BIG=1; QH_HBC_BIG=1 in the environment takes it for every call, unset: for
    // calls of at least 2^23 samples): fp32, three to five stages in this launch, 4096-sample steps with eight outputs a lane in the first
    // stage.  Parity-green (the suite with QH_HBC_BIG=1), LDS work -17 %, bank conflicts -34 % (profiles/r06_g_c5_pmc_step*.json) -- and 2 - 4 %
    // SLOWER: the kernel does not wait for the LDS pipe (profiles/r06_notes.md).
    if constexpr (sizeof(T) == 4) {
        const bool big = h->big == 1 || (h->big < 0 && n >= kBigMin);
        if (big) switch (h->head) {
        case 3: return launch<T, 3, true>(h, in, is, n, out, os);
        case 4: return launch<T, 4, true>(h, in, is, n, out, os);
        case 5: return launch<T, 5, true>(h, in, is, n, out, os);
        }
    }
#endif
    switch (h->head) {
    case 1: return launch<T, 1, false>(h, in, is, n, out, os);
    case 2: return launch<T, 2, false>(h, in, is, n, out, os);
    case 3: return launch<T, 3, false>(h, in, is, n, out, os);
    case 4: return launch<T, 4, false>(h, in, is, n, out, os);
    case 5: return launch<T, 5, false>(h, in, is, n, out, os);
    case 6: return launch<T, 6, false>(h, in, is, n, out, os);
    case 7: return launch<T, 7, false>(h, in, is, n, out, os);
    case 8: return launch<T, 8, false>(h, in, is, n, out, os);
    }
    return set_error(QH_ERR_INVALID, "qh_hbc: nstage out of range");
}

// history samples per channel row: the longer of the two geometries' warm-ups (whole steps of 2048 / 4096 samples)
int warm_of(int ns)
{
    const int need = 42 * ((1 << ns) - 1);
    return (need + 4095) / 4096 * 4096;
}

}  // namespace

extern "C" {

qh_hbc *qh_hbc_create(int device, int nch, int nstage, int dtype, void *stream)
{
    if (nch <= 0 || nstage < 1 || nstage > 8 || (dtype != QH_F64 && dtype != QH_F32)) {
        set_error(QH_ERR_INVALID, "qh_hbc_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_hbc *h = new qh_hbc();
    h->device = device; h->nch = nch; h->nstage = nstage; h->dtype = dtype;
    h->esize = dtype == QH_F64 ? 16 : 8;
    h->head = nstage >= 6 ? 4 : nstage;
    if (const char *e = getenv("QH_HBC_HEAD")) { const int v = atoi(e); h->head = v > 0 && v < nstage ? v : nstage; }      // experiments (tools/dbg/hbc_head.sh (a one-off script, in git history))
    h->warm = warm_of(h->head);
#ifdef QH_EXP_HBC_BIG
    h->big = -1;
    if (const char *e = getenv("QH_HBC_BIG")) h->big = atoi(e) != 0;
#endif
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); delete h; return nullptr; }
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { set_error(QH_ERR_HIP, "stream creation failed"); delete h; return nullptr; }
        h->own_stream = true;
    }
    h->stream = s;
    if (h->head < nstage) {
        h->tail = qh_hbc_create(device, nch, nstage - h->head, dtype, s);
        if (!h->tail) { delete h; return nullptr; }
    }
    const size_t bytes = (size_t)nch * (size_t)h->warm * h->esize;
    for (auto &p : h->hist) {
        if (hipMalloc(&p, bytes) != hipSuccess || hipMemsetAsync(p, 0, bytes, s) != hipSuccess) {
            set_error(QH_ERR_HIP, "qh_hbc_create: history allocation failed");
            delete h;
            return nullptr;
        }
    }
    return h;
}

void qh_hbc_destroy(qh_hbc *h) { delete h; }

int qh_hbc_reset(qh_hbc *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_hbc_reset: null handle");
    QH_HIP(hipSetDevice(h->device));
    for (auto &p : h->hist) QH_HIP(hipMemsetAsync(p, 0, (size_t)h->nch * (size_t)h->warm * h->esize, h->stream));
    return h->tail ? qh_hbc_reset(h->tail) : QH_OK;
}

int qh_hbc_process(qh_hbc *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride)
{
    if (!h || !d_in || !d_out || n_in < 0) return set_error(QH_ERR_INVALID, "qh_hbc_process: bad arguments");
    if (n_in % (1 << h->nstage)) return set_error(QH_ERR_INVALID, "qh_hbc_process: n_in must be a multiple of 2^nstage = %d", 1 << h->nstage);
    if (in_stride < n_in || out_stride < (n_in >> h->nstage)) return set_error(QH_ERR_INVALID, "qh_hbc_process: stride shorter than the data");
    if (n_in == 0) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    if (h->tail) {
        const long long n_mid = n_in >> h->head;
        if (n_mid > h->mid_cap) {
            QH_HIP(hipStreamSynchronize(h->stream));
            if (h->mid) (void)hipFree(h->mid);
            h->mid = nullptr; h->mid_cap = 0;
            QH_HIP(hipMalloc(&h->mid, (size_t)h->nch * (size_t)n_mid * h->esize));
            h->mid_cap = n_mid;
        }
        if (int rc = h->dtype == QH_F64 ? dispatch<double>(h, d_in, in_stride, n_in, h->mid, h->mid_cap)
                                        : dispatch<float>(h, d_in, in_stride, n_in, h->mid, h->mid_cap)) return rc;
        return qh_hbc_process(h->tail, h->mid, h->mid_cap, (int)n_mid, d_out, out_stride);
    }
    return h->dtype == QH_F64 ? dispatch<double>(h, d_in, in_stride, n_in, d_out, out_stride)
                              : dispatch<float>(h, d_in, in_stride, n_in, d_out, out_stride);
}

int qh_hbc_process_host(qh_hbc *h, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride)
{
    if (!h || !h_in || !h_out || n_in < 0) return set_error(QH_ERR_INVALID, "qh_hbc_process_host: bad arguments");
    QH_HIP(hipSetDevice(h->device));
    const int n_out = n_in >> h->nstage;
    void *di = nullptr, *dout = nullptr;
    const size_t ib = (size_t)h->nch * (size_t)n_in * h->esize, ob = (size_t)h->nch * (size_t)(n_out > 0 ? n_out : 1) * h->esize;
    QH_HIP(hipMalloc(&di, ib ? ib : 16));
    if (hipMalloc(&dout, ob) != hipSuccess) { (void)hipFree(di); return set_error(QH_ERR_HIP, "qh_hbc_process_host: hipMalloc failed"); }
    int rc = QH_OK;
    hipError_t e = hipMemcpy2DAsync(di, (size_t)n_in * h->esize, h_in, (size_t)in_stride * h->esize, (size_t)n_in * h->esize,
                                    (size_t)h->nch, hipMemcpyHostToDevice, h->stream);
    if (n_in > 0 && e != hipSuccess) rc = set_error(QH_ERR_HIP, "qh_hbc_process_host: upload failed");
    if (rc == QH_OK) rc = qh_hbc_process(h, di, n_in, n_in, dout, n_out > 0 ? n_out : 1);
    if (rc == QH_OK && n_out > 0) {
        e = hipMemcpy2DAsync(h_out, (size_t)out_stride * h->esize, dout, (size_t)n_out * h->esize, (size_t)n_out * h->esize,
                             (size_t)h->nch, hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) rc = set_error(QH_ERR_HIP, "qh_hbc_process_host: download failed");
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == QH_OK) rc = set_error(QH_ERR_HIP, "qh_hbc_process_host: synchronize failed");
    (void)hipFree(di); (void)hipFree(dout);
    return rc;
}

int qh_hbc_synchronize(qh_hbc *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "qh_hbc_synchronize: null handle");
    QH_HIP(hipSetDevice(h->device));
    QH_HIP(hipStreamSynchronize(h->stream));
    return QH_OK;
}

}  // extern "C"
