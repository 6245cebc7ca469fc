// qh_ingest.hpp -- wire-format IQ samples straight from the receive buffers.
//
// Quisk's sample sources deliver integer IQ of 1-4 bytes per part and left-justify it in an int32 (so that full
// scale is +-2^31 whatever the width) before the complex double stream starts:
//   quisk_read_rx_udp   little-endian, I then Q, 3 bytes:   quisk.c:3378-3392  (x rx_udp_gain_correct)
//   add_rx_samples      little/big-endian, I then Q:        quisk.c:2923-2952
//   read_rx_udp10       Hermes / HPSDR frames of 512 bytes: 8 header bytes, then records of 6 bytes per receiver
//                       (big-endian, the first triple lands in the IMAGINARY part) + 2 microphone bytes:
//                       quisk.c:3745-3760
// decode_packed() reads one sample of any of these from the packed bytes, so the first filter stage can take the
// wire format as it is (6 bytes per sample instead of 16) and no fp64 staging copy of the input ever exists.
#pragma once
#include "qh_fft.hpp"

struct qh_iq_format;

namespace qh {

struct PackedFmt {
    long long first_offset;     // byte offset of sample 0 of channel 0 inside its frame
    long long record_stride;    // bytes from one sample to the next inside a frame
    long long frame_stride;     // bytes from frame to frame (unused when records_per_frame == 0)
    long long chan_stride;      // bytes from channel to channel (own buffers: the buffer pitch; Hermes multi-rx: 6)
    long long total_bytes;      // size of the packed buffer; no load touches a byte at or beyond it
    int sample_bytes;           // 1..4 per part
    int big_endian;
    int q_first;                // the first part is the imaginary one
    int records_per_frame;      // 0: one endless frame
    double inv_rpf;             // 1 / records_per_frame
    double gain;
    unsigned sel_re, sel_im;    // v_perm_b32 selectors: the 8 loaded bytes -> left-justified int32 of each part
};

// Byte selector that lifts the part starting `p` bytes into the loaded window into a left-justified int32:
// little-endian wire bytes keep their order under the zero low bytes (memcpy to ptxr + 4 - sample_bytes,
// quisk.c:3380), big-endian ones are reversed (xi = b0 << 24 | b1 << 16 | b2 << 8, quisk.c:3748).  0x0c selects 0x00.
inline unsigned perm_selector(int p, int sample_bytes, bool big_endian)
{
    unsigned sel = 0;
    for (int j = 0; j < 4; j++) {
        unsigned b = 0x0c;
        if (big_endian) { if (3 - j < sample_bytes) b = (unsigned)(p + 3 - j); }
        else if (j >= 4 - sample_bytes) b = (unsigned)(p + j - (4 - sample_bytes));
        sel |= b << (8 * j);
    }
    return sel;
}

// Fast form for windows that lie wholly inside the buffer: one unaligned 64-bit load, one byte permute per part.
template <typename T>
__device__ __forceinline__ cplx<T> decode_packed_at(const unsigned char *__restrict__ p, unsigned sel_re, unsigned sel_im, double gain)
{
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
    const int ir = (int)__builtin_amdgcn_perm(hi, lo, sel_re), ii = (int)__builtin_amdgcn_perm(hi, lo, sel_im);
    return mk<T>((T)((double)ir * gain), (T)((double)ii * gain));
}

// Sample g of channel ch.  One unaligned 64-bit load covers both parts (2 x 4 bytes at most); near the end of the
// buffer the address is pulled back and the word shifted instead, so nothing past total_bytes is read.
template <typename T>
__device__ __forceinline__ cplx<T> decode_packed(const unsigned char *__restrict__ src, const PackedFmt &f, int ch, long long g)
{
    long long off = f.first_offset + (long long)ch * f.chan_stride;
    if (f.records_per_frame > 0) {
        const long long fr = (long long)(((double)g + 0.5) * f.inv_rpf);        // exact: |g| < 2^31, rpf < 2^20
        off += fr * f.frame_stride + (g - fr * f.records_per_frame) * f.record_stride;
    } else {
        off += g * f.record_stride;
    }
    const long long lim = f.total_bytes - 8;
    unsigned long long w = 0;
    if (lim >= 0) {
        const long long o2 = off < lim ? off : lim;
        __builtin_memcpy(&w, src + o2, 8);
        w >>= 8 * (unsigned)(off - o2);
    } else {                                    // a buffer of fewer than 8 bytes: byte by byte
        for (int k = 0; k < 8; k++)
            if (off + k < f.total_bytes) w |= (unsigned long long)src[off + k] << (8 * k);
    }
    const unsigned bits = 8u * (unsigned)f.sample_bytes;
    const unsigned mask = bits >= 32u ? 0xffffffffu : ((1u << bits) - 1u);
    const unsigned a = (unsigned)w & mask, b = (unsigned)(w >> bits) & mask;
    int ia, ib;
    if (f.big_endian) { ia = (int)__builtin_bswap32(a); ib = (int)__builtin_bswap32(b); }
    else { ia = (int)(a << (32u - bits)); ib = (int)(b << (32u - bits)); }
    const double re = (double)(f.q_first ? ib : ia) * f.gain, im = (double)(f.q_first ? ia : ib) * f.gain;
    return mk<T>((T)re, (T)im);
}

// Validates a host-side format description (include/quiskhip.h) against the buffer and turns it into the device form.
int make_packed_fmt(const struct ::qh_iq_format *f, long long chan_stride, long long total_bytes, long long n, int nch, PackedFmt *out);

}  // namespace qh
