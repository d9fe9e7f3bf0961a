#!/usr/bin/env python3
"""Build-container script: reads the constant tables the hot path needs out of the
reference's HDL / C text and its four NCO ROM images and stores them, as plain
numeric arrays, in tests/golden/ref_tables.npz.

The fixture is DATA (numbers), not source text.  It is the only thing the
reference holds that the CPU oracle and the HIP library can be pinned to: the
reference ships no test vectors.  tests/test_ref_tables.py compares the tables
compiled into the library and into the oracle with it.

Sources (paths under the reference checkout):
  FPGA/rx_ciccomp.vhd:81-146     coeffphase1_1..33, coeffphase2_1..33  (Q15)
  FPGA/rx_hilb.vhd:77-332        coeff1..256                          (Q15)
  FPGA/tx_ciccomp.vhd:81-128     coeffphase1_1..24, coeffphase2_1..24  (Q14)
  FPGA/nco/synthesis/submodules/nco_nco_ii_0_{sin,cos}_{c,f}.hex       (2048 x 14 bit)
  STM32/Src/audio_filters.c:41-126   TX Hilbert pair, HPF / LPF / squelch lattice K, V
  STM32/Src/fft.c:72-183             zoom-FFT biquads and decimator taps

Run:  python tests/golden/make_ref_tables.py [/root/reference]
"""
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tables.npz")


def vhdl_consts(path, prefix):
    pat = re.compile(r"CONSTANT\s+%s(\d+)\s*:.*to_signed\((-?\d+),\s*16\)" % prefix)
    got = {}
    with open(path) as f:
        for line in f:
            m = pat.search(line)
            if m:
                got[int(m.group(1))] = int(m.group(2))
    return np.array([got[i] for i in range(1, len(got) + 1)], np.int16)


def intel_hex_s14(path):
    words = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line.startswith(":") or int(line[7:9], 16) != 0:
                continue
            assert int(line[1:3], 16) == 2
            v = int(line[9:13], 16) & 0x3FFF
            words[int(line[3:7], 16)] = v - 0x4000 if v & 0x2000 else v
    return np.array([words[i] for i in range(len(words))], np.int16)


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return re.sub(r"//[^\n]*", " ", txt)


def c_float_tables(path):
    txt = strip_comments(open(path, encoding="latin-1").read())
    out = {}
    for m in re.finditer(r"static const float32_t\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", txt):
        out[m.group(1)] = np.array([float(v) for v in m.group(2).split(",") if v.strip()], np.float64)
    return out


def anonymous_arrays(txt, cast):
    """Every `(const <cast>[]) { ... }` literal of the text, in order, as float64 arrays."""
    out = []
    for m in re.finditer(r"\(const\s+%s\s*\[\s*\]\s*\)\s*\{([^}]*)\}" % cast, txt):
        out.append(np.array([float(v) for v in m.group(1).split(",") if v.strip()], np.float64))
    return out


def main():
    t = {}
    f = os.path.join(REF, "FPGA")
    t["rx_ciccomp_phase1"] = vhdl_consts(os.path.join(f, "rx_ciccomp.vhd"), "coeffphase1_")
    t["rx_ciccomp_phase2"] = vhdl_consts(os.path.join(f, "rx_ciccomp.vhd"), "coeffphase2_")
    t["rx_hilb_coeff"] = vhdl_consts(os.path.join(f, "rx_hilb.vhd"), "coeff")
    t["tx_ciccomp_phase1"] = vhdl_consts(os.path.join(f, "tx_ciccomp.vhd"), "coeffphase1_")
    t["tx_ciccomp_phase2"] = vhdl_consts(os.path.join(f, "tx_ciccomp.vhd"), "coeffphase2_")
    assert [len(t[k]) for k in ("rx_ciccomp_phase1", "rx_ciccomp_phase2", "rx_hilb_coeff", "tx_ciccomp_phase1",
                                "tx_ciccomp_phase2")] == [33, 33, 256, 24, 24]
    sub = os.path.join(f, "nco/synthesis/submodules")
    for name in ("sin_c", "cos_c", "sin_f", "cos_f"):
        t["nco_" + name] = intel_hex_s14(os.path.join(sub, "nco_nco_ii_0_%s.hex" % name))
        assert len(t["nco_" + name]) == 2048
    # float tables: kept as float64 values of the source literals (the firmware narrows them to float32)
    for name, v in c_float_tables(os.path.join(REF, "STM32/Src/audio_filters.c")).items():
        t["af_" + name] = v
    assert len(t["af_i_tx_coeffs"]) == 201 and len(t["af_q_tx_coeffs"]) == 201
    fft = strip_comments(open(os.path.join(REF, "STM32/Src/fft.c"), encoding="latin-1").read())
    mag = [a for a in anonymous_arrays(fft, "float32_t") if len(a) == 20]
    fir = [a for a in anonymous_arrays(fft, "float") if len(a) == 4]
    assert len(mag) == 4 and len(fir) == 4, (len(mag), len(fir))
    t["fft_zoom_biquad"] = np.stack(mag)      # zoom 2, 4, 8, 16: 4 stages x {b0, b1, b2, a1, a2}
    t["fft_zoom_fir"] = np.stack(fir)         # zoom 2, 4, 8, 16: 4 taps
    np.savez_compressed(OUT, **t)
    print("wrote %s: %d tables" % (OUT, len(t)))
    for k in sorted(t):
        print("  %-28s %s %s" % (k, t[k].dtype, t[k].shape))


if __name__ == "__main__":
    main()
