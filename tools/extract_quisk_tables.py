#!/usr/bin/env python3
"""Extracts the FIR coefficient tables the Quisk-native receive chain uses (quisk.c:1700-1723,1877-1893) from the
reference build oracle/_ref/libquisk_filter_ref.so (filter.c compiled where it lies; it defines the arrays of
filters.h) into quisk_amd/data/quisk_filter_tables.npz.  Numeric data only; runs only in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po      # noqa: E402

TABLES = {  # name: length (filter.h:57-86)
    "quiskFilt48dec24Coefs": 98, "quiskFilt144D3Coefs": 147, "quiskFilt240D5CoefsSharp": 245,
    "quiskFilt300D5Coefs": 125, "quiskAudio24p4Coefs": 50, "quiskAudio24p6Coefs": 36, "quiskAudio24p3Coefs": 100,
    "quiskLpFilt48Coefs": 186, "quiskAudioFmHpCoefs": 309, "quiskFilt16dec8Coefs": 62, "quiskAudio48p6Coefs": 71,
    "quiskAudio96Coefs": 11,
    # SDR-IQ rates (quisk.c:1706-1710,1732-1768)
    "quiskFilt53D1Coefs": 55, "quiskFilt111D2Coefs": 114, "quiskFilt133D2Coefs": 136, "quiskFilt167D3Coefs": 174,
    "quiskFilt185D3Coefs": 189,
}


def main():
    po.build(ref=True)
    out = {k: po.ref_table(k, n) for k, n in TABLES.items()}
    path = os.path.join(ROOT, "quisk_amd", "data", "quisk_filter_tables.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
