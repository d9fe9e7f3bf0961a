#!/usr/bin/env python3
"""Writes tests/golden/ddc_golden.npz: small seeded input / expected-output vectors
for every stage of the chain, produced by the CPU oracle (oracle/).  The
reference ships no vectors of its own; these pin the oracle against drift and
let the GPU tests check the HIP path against stored data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
import audio_signals as sig  # noqa: E402


def lfsr_adc(n, seed=0xACE1):
    """12-bit words from a 16-bit Fibonacci LFSR (taps 16,14,13,11)."""
    out = np.zeros(n, np.int16)
    s = seed
    for i in range(n):
        for _ in range(12):
            bit = ((s >> 0) ^ (s >> 2) ^ (s >> 3) ^ (s >> 5)) & 1
            s = (s >> 1) | (bit << 15)
        v = s & 0xFFF
        out[i] = v - 4096 if v & 0x800 else v
    return out


def adc_cases(n):
    k = np.arange(n)
    imp = np.zeros(n, np.int16)
    imp[100] = 2047
    step = np.zeros(n, np.int16)
    step[n // 3:] = 1500
    return {
        "impulse": imp,
        "step": step,
        "dc_pos_fullscale": np.full(n, 2047, np.int16),
        "dc_neg_fullscale": np.full(n, -2048, np.int16),
        "tone_fs4": np.array([0, 2047, 0, -2047] * (n // 4), np.int16),
        "tone_near_lo": np.rint(1800 * np.sin(2 * np.pi * (7100000 + 1500) * k / 49152000.0)).astype(np.int16),
        "lfsr": lfsr_adc(n),
    }


def main():
    oracle.build()
    g = {}
    n = 1024 * 12
    phis = {"impulse": 0, "step": 0, "dc_pos_fullscale": 1, "dc_neg_fullscale": 0x3FFFFF, "tone_fs4": 1 << 20,
            "tone_near_lo": 605867, "lfsr": 2796203}
    for name, adc in adc_cases(n).items():
        phi = phis[name]
        front = oracle.RxFront(phi)
        ci, cq = front.run(adc)
        cic = np.stack([ci, cq], axis=1)
        spec, voice = oracle.RxChain(phi).run(adc)
        g["rx_%s_adc" % name] = adc
        g["rx_%s_phi" % name] = np.uint32(phi)
        g["rx_%s_cic" % name] = cic
        g["rx_%s_spec" % name] = spec
        g["rx_%s_voice" % name] = voice
    # rx_ciccomp / rx_hilb on their own: impulse (replays the taps), worst-case sign
    # pattern (wrap path), values that land the sum exactly on .5 (round half to even)
    rng = np.random.default_rng(2024)
    x = np.zeros(160, np.int16)
    x[0] = 16384
    g["ciccomp_impulse_in"] = x
    g["ciccomp_impulse_out"] = oracle.RxCicComp().run(x)
    x = rng.integers(-32768, 32768, 400).astype(np.int16)
    g["ciccomp_random_in"] = x
    g["ciccomp_random_out"] = oracle.RxCicComp().run(x)
    x = np.full(200, -32768, np.int16)  # every product at its extreme: the wrap path
    g["ciccomp_negfs_in"] = x
    g["ciccomp_negfs_out"] = oracle.RxCicComp().run(x)
    x = np.zeros(700, np.int16)
    x[0] = 16384
    g["hilb_impulse_in"] = x
    g["hilb_impulse_out"] = oracle.RxHilb().run(x)
    x = rng.integers(-32768, 32768, (700,)).astype(np.int16)
    g["hilb_random_in"] = x
    g["hilb_random_out"] = oracle.RxHilb().run(x)
    # TX: impulse, full-scale DC, random
    for name, iq in {
        "impulse": np.concatenate([np.array([[20000, -20000]], np.int16), np.zeros((11, 2), np.int16)]),
        "dc_fullscale": np.full((12, 2), 32767, np.int16),
        "neg_fullscale": np.full((12, 2), -32768, np.int16),
        "random": rng.integers(-32768, 32768, (12, 2)).astype(np.int16),
    }.items():
        phi = 605867 if name != "impulse" else 0
        g["tx_%s_iq" % name] = iq
        g["tx_%s_phi" % name] = np.uint32(phi)
        g["tx_%s_dac" % name] = oracle.TxChain(phi).run(iq)
    # tuning words (getPhraseFromFrequency incl. Nyquist folding)
    freqs = np.array([0, 1, 100000, 1840000, 7100000, 14200000, 24576000, 24576001, 28500000, 30000000, 49152000,
                      50000000, 60000000, 98304000, 144000000], np.uint32)
    ph = [oracle.phrase_from_freq(int(f)) for f in freqs]
    g["phrase_freq"] = freqs
    g["phrase_word"] = np.array([p[0] for p in ph], np.uint32)
    g["phrase_swap"] = np.array([p[1] for p in ph], np.uint8)
    # bus frames
    spec = rng.integers(-32768, 32768, (16, 2)).astype(np.int16)
    voice = rng.integers(-32768, 32768, (16, 2)).astype(np.int16)
    g["frames_spec"], g["frames_voice"] = spec, voice
    g["frames_rx_bytes"] = np.frombuffer(b"".join(
        oracle.pack_rx_frame(int(spec[k, 0]), int(spec[k, 1]), int(voice[k, 0]), int(voice[k, 1]))
        for k in range(16)), np.uint8)
    # audio: one config per demodulator family, 10 blocks each
    na = 192 * 10
    for i, kw in enumerate([dict(mode="USB"), dict(mode="LSB", notch=1, dnr=1), dict(mode="AM", filter_width=6000),
                            dict(mode="CW_U", filter_width=500), dict(mode="IQ")]):
        sp, vo = sig.noisy_two_tone(na, 900 + 2 * i), sig.noisy_two_tone(na, 901 + 2 * i)
        a = oracle.Audio(oracle.audio_params(**kw))
        g["audio%d_spec" % i], g["audio%d_voice" % i] = sp, vo
        g["audio%d_out" % i] = a.rx(sp, vo)
        g["audio%d_kw" % i] = np.array(repr(sorted(kw.items())))
    # TX audio
    mic = sig.quantise(np.stack([2000 * np.sin(2 * np.pi * 1000 * np.arange(na) / 48000.0)] * 2, axis=1))
    for i, kw in enumerate([dict(mode="USB", rf_power=50), dict(mode="AM", rf_power=80, filter_width=6000)]):
        a = oracle.Audio(oracle.audio_params(**kw))
        g["txaudio%d_mic" % i] = mic
        g["txaudio%d_iq" % i] = a.tx(mic)[0]
        g["txaudio%d_kw" % i] = np.array(repr(sorted(kw.items())))
    # panadapter: 6 frames of a carrier in noise
    x = sig.quantise(sig.tone_iq(512 * 6, 6000.0, 500.0) + rng.normal(0, 20, (512 * 6, 2)))
    a = oracle.Audio(oracle.audio_params(fft_averaging=2))
    bins, rows = [], []
    for f in range(6):
        b, r, _ = a.fft(x[512 * f:512 * (f + 1)])
        bins.append(b)
        rows.append(r)
    g["fft_spec"], g["fft_bins"], g["fft_rows"] = x, np.stack(bins), np.stack(rows)
    out = os.path.join(HERE, "ddc_golden.npz")
    np.savez_compressed(out, **g)
    print("wrote %s (%d arrays, %d bytes)" % (out, len(g), os.path.getsize(out)))


if __name__ == "__main__":
    main()
