"""Synthetic IQ input shared by bench.py and the tests.

Per channel c:  x[n] = A1*exp(j*2*pi*f1*n/fs) + A2*exp(j*2*pi*f2*n/fs) + sigma*(g_re + j*g_im)
WDSP scale (+-1.0): A1 = 0.1 in the passband after the shift, A2 = 0.05 out of band, sigma = 0.01.
The shift is +10 kHz + 37*c Hz, so the in-band tone sits at (-1000 - shift) Hz before the shift
(WDSP's "USB 300..3000" passes conventional -3000..-300 Hz: wdsp/fir.c:246-249).
"""
import numpy as np


def shift_freq(c):
    return 10000.0 + 37.0 * c


def channel_tones(c, fs):
    f1 = -1000.0 - shift_freq(c)        # lands at -1000 Hz after the +shift: inside the USB passband
    f2 = 30000.0 - shift_freq(c)        # far out of band
    return f1, f2


def make_input_numpy(nch, n, fs=192000.0, first_channel=0, a1=0.1, a2=0.05, sigma=0.01):
    """complex128 [nch, n]; noise from numpy.random.default_rng(1000 + c)."""
    t = np.arange(n, dtype=np.float64)
    x = np.empty((nch, n), dtype=np.complex128)
    for i in range(nch):
        c = first_channel + i
        f1, f2 = channel_tones(c, fs)
        rng = np.random.default_rng(1000 + c)
        g = rng.standard_normal((n, 2))
        x[i] = (a1 * np.exp(2j * np.pi * ((f1 / fs) * t % 1.0)) + a2 * np.exp(2j * np.pi * ((f2 / fs) * t % 1.0))
                + sigma * (g[:, 0] + 1j * g[:, 1]))
    return x


def make_input_torch(nch, n, device, fs=192000.0, first_channel=0, a1=0.1, a2=0.05, sigma=0.01, chunk=16):
    """Same signal model generated on the GPU (torch generator seeded 1000 + c); complex128 [nch, n]."""
    import torch
    x = torch.empty((nch, n), dtype=torch.complex128, device=device)
    t = torch.arange(n, dtype=torch.float64, device=device)
    for i0 in range(0, nch, chunk):
        i1 = min(nch, i0 + chunk)
        for i in range(i0, i1):
            c = first_channel + i
            f1, f2 = channel_tones(c, fs)
            gen = torch.Generator(device=device)
            gen.manual_seed(1000 + c)
            ph1 = torch.remainder(t * (f1 / fs), 1.0) * (2.0 * np.pi)
            ph2 = torch.remainder(t * (f2 / fs), 1.0) * (2.0 * np.pi)
            re = a1 * torch.cos(ph1) + a2 * torch.cos(ph2)
            im = a1 * torch.sin(ph1) + a2 * torch.sin(ph2)
            re += sigma * torch.randn(n, dtype=torch.float64, device=device, generator=gen)
            im += sigma * torch.randn(n, dtype=torch.float64, device=device, generator=gen)
            x[i] = torch.complex(re, im)
            del ph1, ph2, re, im
    return x


def make_mode_input_numpy(mode, c, n, fs=192000.0, sigma=0.01):
    """BASELINE config 4 inputs: 'usb' as make_input_numpy; 'am': carrier with m = 0.5, 1 kHz
    modulation; 'fm': carrier with 1 kHz tone at +-3 kHz deviation; all at -shift_freq(c) so that the shift
    brings the carrier to 0 Hz.  complex128 [n], noise from default_rng(1000 + c)."""
    t = np.arange(n, dtype=np.float64)
    rng = np.random.default_rng(1000 + c)
    g = rng.standard_normal((n, 2))
    noise = sigma * (g[:, 0] + 1j * g[:, 1])
    fc = -shift_freq(c)
    car = np.exp(2j * np.pi * ((fc / fs) * t % 1.0))
    if mode == "usb":
        return make_input_numpy(1, n, fs=fs, first_channel=c, sigma=sigma)[0]
    if mode == "am":
        return 0.1 * (1.0 + 0.5 * np.cos(2 * np.pi * 1000.0 / fs * t)) * car + noise
    if mode == "fm":
        return 0.1 * car * np.exp(1j * (3000.0 / 1000.0) * np.sin(2 * np.pi * 1000.0 / fs * t)) + noise
    raise ValueError(mode)


def make_mode_input_torch(modes, n, device, fs=192000.0, sigma=0.01, first_channel=0, periodic=False):
    """make_mode_input_numpy's signal model generated on the GPU: modes = one of 'usb' / 'am' / 'fm' per channel; complex128
    [len(modes), n]; noise from a torch generator seeded 1000 + c (not the numpy stream: bench input, not a parity vector).
    periodic: the 'usb' tones moved (by less than fs / 2n) onto multiples of fs / n, so that a buffer fed again and again is ONE
    continuous stream instead of a phase jump per call."""
    import torch
    nch = len(modes)
    x = torch.empty((nch, n), dtype=torch.complex128, device=device)
    t = torch.arange(n, dtype=torch.float64, device=device)
    two_pi = 2.0 * np.pi
    for i, mode in enumerate(modes):
        c = first_channel + i
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + c)
        if mode == "usb":
            f1, f2 = channel_tones(c, fs)
            if periodic:
                f1, f2 = round(f1 * n / fs) * fs / n, round(f2 * n / fs) * fs / n
            ph1 = torch.remainder(t * (f1 / fs), 1.0) * two_pi
            ph2 = torch.remainder(t * (f2 / fs), 1.0) * two_pi
            re = 0.1 * torch.cos(ph1) + 0.05 * torch.cos(ph2)
            im = 0.1 * torch.sin(ph1) + 0.05 * torch.sin(ph2)
        else:
            ph = torch.remainder(t * (-shift_freq(c) / fs), 1.0) * two_pi
            aud = t * (two_pi * 1000.0 / fs)
            if mode == "am":
                amp = 0.1 * (1.0 + 0.5 * torch.cos(aud))
            elif mode == "fm":
                amp = 0.1
                ph = ph + (3000.0 / 1000.0) * torch.sin(aud)
            else:
                raise ValueError(mode)
            re = amp * torch.cos(ph)
            im = amp * torch.sin(ph)
        re = re + sigma * torch.randn(n, dtype=torch.float64, device=device, generator=gen)
        im = im + sigma * torch.randn(n, dtype=torch.float64, device=device, generator=gen)
        x[i] = torch.complex(re, im)
    return x


def impulsive_input(nch, n, seed=7, scale=1e6):
    """Gaussian noise at `scale` with spikes, short bursts and pairs of bursts 20-40 dB above it (noise-blanker input)."""
    out = np.empty((nch, n), dtype=np.complex128)
    for c in range(nch):
        rng = np.random.default_rng(seed + c)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale
        for p in rng.integers(0, n, size=max(4, n // 3000)):
            m = min(int(rng.integers(1, 6)), n - int(p))
            x[p:p + m] += (rng.standard_normal(m) + 1j * rng.standard_normal(m)) * 50.0 * scale
            q = int(p) + int(rng.integers(3, 40))              # often a second run inside the first one's recovery ramp
            if rng.random() < 0.5 and q < n:
                x[q] += 80.0 * scale
        out[c] = x
    return out
