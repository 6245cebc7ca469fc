// qh_ingest.hip -- wire-format IQ to complex samples (include/quiskhip.h group 7); see qh_ingest.hpp.
#include "qh_ingest.hpp"
#include "qh_internal.hpp"

using namespace qh;

namespace {

template <typename T>
__global__ __launch_bounds__(NT) void unpack_kernel(const unsigned char *src, PackedFmt f, int n, cplx<T> *dst, long long dst_stride)
{
    const int ch = blockIdx.y;
    for (long long g = (long long)blockIdx.x * NT + threadIdx.x; g < n; g += (long long)gridDim.x * NT)
        dst[(long long)ch * dst_stride + g] = decode_packed<T>(src, f, ch, g);
}

// read_rx_udp17 (quisk.c:3821-3999): packets of 2 header bytes + records of 6 (24-bit little-endian I, then Q, left-justified in
// an int32).  The least significant bit of I says which of two interleaved streams the sample belongs to -- set: the
// panadapter stream ("channel 1"), clear: the receiver's -- and on the panadapter stream a clear LSB of Q marks the first
// sample of a scan's first block.  Which output slot a sample takes depends on the flags of all samples before it: a stream
// compaction.  One workgroup walks the records in tiles of 1024; inside a tile every wavefront counts its three kinds of
// sample with ballots, the sixteen counts are scanned through LDS, and the running offsets carry from tile to tile.  The
// panadapter samples leave conjugated when the spectrum is inverted and with the DC estimate removed, as the reference's loop
// does; their plain sum goes back to the caller, who owns the estimate (the reference renews it once a second of wall time).
constexpr int kUdp17Threads = 1024;
__global__ __launch_bounds__(kUdp17Threads) void udp17_kernel(const unsigned char *src, int npackets, int packet_bytes, double gain,
                                                              int invert, double dc_re, double dc_im, double2 *ch0, double2 *ch1,
                                                              int *marks, long long *counts, double *dc_sum)
{
    __shared__ int s_cnt[3][kUdp17Threads / 64];
    __shared__ int s_base[3];
    __shared__ double s_sum[2][kUdp17Threads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (packet_bytes - 2) / 6;             // records per packet
    const long long total = (long long)npackets * per;
    if (t < 3) s_base[t] = 0;
    double sum_re = 0.0, sum_im = 0.0;
    int overrange = 0;
    for (int p = t; p < npackets; p += kUdp17Threads) overrange += (src[(long long)p * packet_bytes + 1] & 0x02) ? 1 : 0;     // quisk.c:3912
    __syncthreads();
    for (long long g0 = 0; g0 < total; g0 += kUdp17Threads) {
        const long long g = g0 + t;
        bool live = g < total, is1 = false, mark = false;
        double2 v = make_double2(0.0, 0.0);
        if (live) {
            const unsigned char *r = src + (g / per) * packet_bytes + 2 + (g % per) * 6;
            const int xr = (int)(((unsigned)r[0] << 8) | ((unsigned)r[1] << 16) | ((unsigned)r[2] << 24));      // memcpy(ptxr + 1, ., 3)
            const int xi = (int)(((unsigned)r[3] << 8) | ((unsigned)r[4] << 16) | ((unsigned)r[5] << 24));
            v = make_double2((double)xr * gain, (double)xi * gain);                 // (xr + xi I) * rx_udp_gain_correct
            is1 = (xr & 0x100) != 0;
            mark = is1 && !(xi & 0x100);
        }
        const unsigned long long b1 = __ballot(live && is1), b0 = __ballot(live && !is1), bm = __ballot(mark);
        if (lane == 0) { s_cnt[0][wave] = __popcll(b0); s_cnt[1][wave] = __popcll(b1); s_cnt[2][wave] = __popcll(bm); }
        __syncthreads();
        int off0 = s_base[0], off1 = s_base[1], offm = s_base[2];
        for (int w = 0; w < wave; w++) { off0 += s_cnt[0][w]; off1 += s_cnt[1][w]; offm += s_cnt[2][w]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (live) {
            if (is1) {
                const int slot = off1 + __popcll(b1 & below);
                if (invert) v.y = -v.y;                 // conj(sample), quisk.c:3941-3942
                sum_re += v.x; sum_im += v.y;           // dc_sum += sample
                ch1[slot] = make_double2(v.x - dc_re, v.y - dc_im);
                if (mark) marks[offm + __popcll(bm & below)] = slot;
            } else ch0[off0 + __popcll(b0 & below)] = v;
        }
        __syncthreads();
        if (t == 0)
            for (int k = 0; k < 3; k++) { int a = 0; for (int w = 0; w < kUdp17Threads / 64; w++) a += s_cnt[k][w]; s_base[k] += a; }
        __syncthreads();
    }
    // sums in a fixed order: lanes, then wavefronts
    for (int d = 32; d >= 1; d >>= 1) { sum_re += __shfl_xor(sum_re, d, 64); sum_im += __shfl_xor(sum_im, d, 64); overrange += __shfl_xor(overrange, d, 64); }
    if (lane == 0) { s_sum[0][wave] = sum_re; s_sum[1][wave] = sum_im; s_cnt[0][wave] = overrange; }
    __syncthreads();
    if (t == 0) {
        double a = 0.0, b = 0.0;
        int o = 0;
        for (int w = 0; w < kUdp17Threads / 64; w++) { a += s_sum[0][w]; b += s_sum[1][w]; o += s_cnt[0][w]; }
        dc_sum[0] = a; dc_sum[1] = b;
        counts[0] = s_base[0]; counts[1] = s_base[1]; counts[2] = s_base[2]; counts[3] = o;
    }
}

}  // namespace

int qh::make_packed_fmt(const qh_iq_format *f, long long chan_stride, long long total_bytes, long long n, int nch, PackedFmt *out)
{
    if (!f || f->sample_bytes < 1 || f->sample_bytes > 4 || f->record_stride < 2 * f->sample_bytes || f->first_offset < 0 ||
        f->records_per_frame < 0 || (f->records_per_frame > 0 && f->frame_stride <= 0) || chan_stride < 0 || n < 0 || nch <= 0)
        return set_error(QH_ERR_INVALID, "bad packed sample format");
    if (f->records_per_frame >= (1 << 20)) return set_error(QH_ERR_INVALID, "records_per_frame too large");
    // last byte touched by the last sample of the last channel
    long long last = f->first_offset + (nch - 1) * chan_stride;
    if (n > 0) {
        const long long g = n - 1;
        if (f->records_per_frame > 0) last += (g / f->records_per_frame) * f->frame_stride + (g % f->records_per_frame) * f->record_stride;
        else last += g * f->record_stride;
        last += 2 * f->sample_bytes;
        if (last > total_bytes) return set_error(QH_ERR_INVALID, "packed buffer of %lld bytes is shorter than the %lld the samples span", total_bytes, last);
    }
    out->first_offset = f->first_offset; out->record_stride = f->record_stride; out->frame_stride = f->frame_stride;
    out->chan_stride = chan_stride; out->total_bytes = total_bytes;
    out->sample_bytes = f->sample_bytes; out->big_endian = f->big_endian ? 1 : 0; out->q_first = f->q_first ? 1 : 0;
    out->records_per_frame = f->records_per_frame;
    out->inv_rpf = f->records_per_frame > 0 ? 1.0 / f->records_per_frame : 0.0;
    out->gain = f->gain;
    const int p_re = f->q_first ? f->sample_bytes : 0, p_im = f->q_first ? 0 : f->sample_bytes;
    out->sel_re = perm_selector(p_re, f->sample_bytes, f->big_endian != 0);
    out->sel_im = perm_selector(p_im, f->sample_bytes, f->big_endian != 0);
    return QH_OK;
}

extern "C" {

// quisk_read_rx_udp on a little-endian host (quisk.c:3378-3392): 3-byte little-endian I then Q, back to back
void qh_iq_format_le24(qh_iq_format *f, double gain)
{
    f->sample_bytes = 3; f->big_endian = 0; f->q_first = 0; f->first_offset = 0; f->record_stride = 6;
    f->records_per_frame = 0; f->frame_stride = 0; f->gain = gain;
}

// read_rx_udp10 (quisk.c:3745-3760): 512-byte frames, 8 header bytes, records of 6 bytes per receiver + 2
// microphone bytes, big-endian, first triple = imaginary part; receiver r of nrx sits 6 r bytes into the record
void qh_iq_format_hermes(qh_iq_format *f, int nrx, double gain)
{
    if (nrx < 1) nrx = 1;
    f->sample_bytes = 3; f->big_endian = 1; f->q_first = 1; f->first_offset = 8; f->record_stride = 6 * nrx + 2;
    f->records_per_frame = 504 / (6 * nrx + 2); f->frame_stride = 512; f->gain = gain;
}

int qh_unpack_iq(int device, void *stream, const void *d_src, long long src_bytes, const qh_iq_format *fmt, int nch,
                 long long chan_stride, int n, void *d_dst, long long dst_stride, int dtype)
{
    if (!d_src || !d_dst || (dtype != QH_F64 && dtype != QH_F32) || dst_stride < n)
        return set_error(QH_ERR_INVALID, "qh_unpack_iq: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    PackedFmt pk;
    if (int rc = make_packed_fmt(fmt, chan_stride, src_bytes, n, nch, &pk)) return rc;
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(device));
    long long blocks = ((long long)n + NT - 1) / NT;
    if (blocks > 65535) blocks = 65535;
    dim3 grid((unsigned)blocks, (unsigned)nch);
    if (dtype == QH_F64)
        hipLaunchKernelGGL(unpack_kernel<double>, grid, dim3(NT), 0, (hipStream_t)stream, (const unsigned char *)d_src, pk, n,
                           (double2 *)d_dst, dst_stride);
    else
        hipLaunchKernelGGL(unpack_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const unsigned char *)d_src, pk, n,
                           (float2 *)d_dst, dst_stride);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_unpack_iq_host(int device, const void *h_src, long long src_bytes, const qh_iq_format *fmt, int nch, long long chan_stride,
                      int n, void *h_dst, long long dst_stride, int dtype)
{
    if (!h_src || !h_dst || src_bytes <= 0 || n < 0) return set_error(QH_ERR_INVALID, "qh_unpack_iq_host: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    QH_HIP(hipSetDevice(device));
    const size_t es = dtype == QH_F64 ? 16 : 8;
    void *dsrc = nullptr, *ddst = nullptr;
    QH_HIP(hipMalloc(&dsrc, (size_t)src_bytes));
    if (hipMalloc(&ddst, (size_t)nch * (size_t)(n > 0 ? n : 1) * es) != hipSuccess) { (void)hipFree(dsrc); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    int rc = QH_OK;
    if (hipMemcpy(dsrc, h_src, (size_t)src_bytes, hipMemcpyHostToDevice) != hipSuccess) rc = set_error(QH_ERR_HIP, "upload failed");
    if (rc == QH_OK) rc = qh_unpack_iq(device, nullptr, dsrc, src_bytes, fmt, nch, chan_stride, n, ddst, n > 0 ? n : 1, dtype);
    if (rc == QH_OK && n > 0 &&
        hipMemcpy2D(h_dst, (size_t)dst_stride * es, ddst, (size_t)n * es, (size_t)n * es, (size_t)nch, hipMemcpyDeviceToHost) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "download failed");
    (void)hipFree(dsrc); (void)hipFree(ddst);
    return rc;
}


// read_rx_udp17's sample loop (quisk.c:3917-3996) on `npackets` packets of `packet_bytes` (RX_UDP_SIZE = 1442, quisk.c:204) that
// lie back to back on the device.  d_ch0 / d_ch1: room for npackets * (packet_bytes - 2) / 6 complex doubles each; d_marks:
// as many ints (slots of d_ch1 where a scan's first block starts); d_counts: 4 long longs = samples on channel 0, on channel 1,
// marks, packets with the ADC overrange bit; d_dc_sum: 2 doubles = sum of the channel-1 samples before the DC estimate
// (dc_re, dc_im) came off.
int qh_unpack_udp17(int device, void *stream, const void *d_src, int npackets, int packet_bytes, double gain, int invert_spectrum,
                    double dc_re, double dc_im, void *d_ch0, void *d_ch1, int *d_marks, long long *d_counts, double *d_dc_sum)
{
    if (!d_src || npackets < 0 || packet_bytes < 8 || (packet_bytes - 2) % 6 || !d_ch0 || !d_ch1 || !d_marks || !d_counts || !d_dc_sum)
        return set_error(QH_ERR_INVALID, "qh_unpack_udp17: bad arguments (packets of 2 + 6 k bytes)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    QH_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(udp17_kernel, dim3(1), dim3(kUdp17Threads), 0, (hipStream_t)stream, (const unsigned char *)d_src, npackets, packet_bytes,
                       gain, invert_spectrum ? 1 : 0, dc_re, dc_im, (double2 *)d_ch0, (double2 *)d_ch1, d_marks, d_counts, d_dc_sum);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_unpack_udp17_host(int device, const void *h_src, int npackets, int packet_bytes, double gain, int invert_spectrum, double dc_re,
                         double dc_im, void *h_ch0, void *h_ch1, int *h_marks, long long *h_counts, double *h_dc_sum)
{
    if (!h_src || npackets <= 0 || packet_bytes < 8 || (packet_bytes - 2) % 6 || !h_ch0 || !h_ch1 || !h_marks || !h_counts || !h_dc_sum)
        return set_error(QH_ERR_INVALID, "qh_unpack_udp17_host: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
    QH_HIP(hipSetDevice(device));
    const size_t nrec = (size_t)npackets * (size_t)((packet_bytes - 2) / 6), nbytes = (size_t)npackets * packet_bytes;
    unsigned char *dsrc = nullptr;
    double2 *d0 = nullptr, *d1 = nullptr;
    int *dm = nullptr;
    long long *dc = nullptr;
    double *ds = nullptr;
    int rc = QH_OK;
    if (hipMalloc((void **)&dsrc, nbytes) != hipSuccess || hipMalloc((void **)&d0, nrec * 16) != hipSuccess || hipMalloc((void **)&d1, nrec * 16) != hipSuccess ||
        hipMalloc((void **)&dm, nrec * 4) != hipSuccess || hipMalloc((void **)&dc, 32) != hipSuccess || hipMalloc((void **)&ds, 16) != hipSuccess)
        rc = set_error(QH_ERR_HIP, "hipMalloc failed");
    if (rc == QH_OK && hipMemcpy(dsrc, h_src, nbytes, hipMemcpyHostToDevice) != hipSuccess) rc = set_error(QH_ERR_HIP, "upload failed");
    if (rc == QH_OK) rc = qh_unpack_udp17(device, nullptr, dsrc, npackets, packet_bytes, gain, invert_spectrum, dc_re, dc_im, d0, d1, dm, dc, ds);
    if (rc == QH_OK && (hipMemcpy(h_counts, dc, 32, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h_dc_sum, ds, 16, hipMemcpyDeviceToHost) != hipSuccess))
        rc = set_error(QH_ERR_HIP, "download failed");
    if (rc == QH_OK) {
        const size_t n0 = (size_t)h_counts[0], n1 = (size_t)h_counts[1], nm = (size_t)h_counts[2];
        if ((n0 && hipMemcpy(h_ch0, d0, n0 * 16, hipMemcpyDeviceToHost) != hipSuccess) || (n1 && hipMemcpy(h_ch1, d1, n1 * 16, hipMemcpyDeviceToHost) != hipSuccess) ||
            (nm && hipMemcpy(h_marks, dm, nm * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = set_error(QH_ERR_HIP, "download failed");
    }
    (void)hipFree(dsrc); (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dm); (void)hipFree(dc); (void)hipFree(ds);
    return rc;
}

}  // extern "C"
