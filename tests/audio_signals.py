"""Seeded 48 kHz I/Q test signals shared by the audio oracle tests and the GPU
parity tests (data only; no reference code involved)."""
import numpy as np

FS = 48000.0


def tone_iq(n, f_hz, amp, phase=0.0, quadrature=+1):
    """int16 [n][2]: I = amp cos, Q = quadrature * amp sin."""
    t = np.arange(n) / FS
    i = amp * np.cos(2 * np.pi * f_hz * t + phase)
    q = quadrature * amp * np.sin(2 * np.pi * f_hz * t + phase)
    return np.stack([i, q], axis=1)


def quantise(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def noisy_two_tone(n, seed, amp=300.0):
    """Two audio-band complex tones plus noise, different per seed."""
    r = np.random.default_rng(seed)
    f1, f2 = r.uniform(300, 2500), r.uniform(-2500, 6000)
    x = tone_iq(n, f1, amp, r.uniform(0, 6.28)) + tone_iq(n, f2, amp * 0.4, r.uniform(0, 6.28))
    x += r.normal(0, amp * 0.05, size=x.shape)
    # slow fading so that the AGC keeps moving
    x *= (1.0 + 0.5 * np.sin(2 * np.pi * 1.7 * np.arange(n) / FS + seed))[:, None]
    return quantise(x)


def ssb_pair(n, f_hz, amp, upper=True):
    """What the FPGA VOICE path delivers for a single tone f_hz above (upper) or
    below the carrier: after the Hilbert transformer I and Q are in phase for one
    sideband and in antiphase for the other."""
    t = np.arange(n) / FS
    i = amp * np.cos(2 * np.pi * f_hz * t)
    q = i if upper else -i
    return quantise(np.stack([i, q], axis=1))


def fm_iq(n, f_mod, deviation_hz, amp=2000.0, seed=0):
    t = np.arange(n) / FS
    ph = (deviation_hz / f_mod) * np.sin(2 * np.pi * f_mod * t)
    r = np.random.default_rng(seed)
    x = amp * np.stack([np.cos(ph), np.sin(ph)], axis=1) + r.normal(0, 2.0, size=(n, 2))
    return quantise(x)


MORSE = {"C": "-.-.", "Q": "--.-", "D": "-..", "E": ".", "T": "-", "S": "...", "O": "---", "R": ".-.",
         "A": ".-", "N": "-.", "U": "..-", "3": "...--"}


def cw_keyed(text, wpm=20, f_hz=250.0, amp=400.0, lead_ms=400, tail_ms=1200, seed=1):
    """Keyed carrier, as the VOICE path would deliver it in CW_U (I == Q)."""
    dot = 1.2 / wpm
    seq = [(0, lead_ms / 1000.0)]
    for wi, word in enumerate(text.split(" ")):
        if wi:
            seq.append((0, 7 * dot))
        for ci, ch in enumerate(word):
            if ci:
                seq.append((0, 3 * dot))
            for si, sym in enumerate(MORSE[ch]):
                if si:
                    seq.append((0, dot))
                seq.append((1, dot if sym == "." else 3 * dot))
    seq.append((0, tail_ms / 1000.0))
    env = np.concatenate([np.full(int(round(d * FS)), float(k)) for k, d in seq])
    n = (len(env) + 191) // 192 * 192
    env = np.pad(env, (0, n - len(env)))
    t = np.arange(n) / FS
    r = np.random.default_rng(seed)
    i = amp * env * np.cos(2 * np.pi * f_hz * t) + r.normal(0, 1.0, n)
    return quantise(np.stack([i, i], axis=1))
