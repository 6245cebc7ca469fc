// qh_emnr_tables.hpp -- the EMNR tables the library holds itself, as the reference does: wdsp/emnr.c:317-328 reads GG / GGS from the file
// `calculus` and falls back to the arrays compiled into it (calculus.c), readZetaHat (emnr.c:207-225) does the same with `zetaHat.bin` /
// zetahat.c.  The definitions are generated at build time (quisk_amd/build.py, _emnr_tables_source) from
// quisk_amd/data/wdsp_emnr_tables.npz: the doubles as their 64-bit patterns.
#pragma once
namespace qh {
extern const unsigned long long kEmnrDefaultGG[241 * 241], kEmnrDefaultGGS[241 * 241];
extern const unsigned long long kEmnrDefaultZeta[3600];          // 60 x 60; 0 where the cell is not valid
extern const unsigned long long kEmnrDefaultRange[4];            // gamma min / max, xi_hat min / max (dB)
extern const int kEmnrDefaultValid[3600];
}
