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

}  // extern "C"
