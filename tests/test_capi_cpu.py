"""CPU-side checks of the C ABI: the shared library loads, exports every symbol
the header declares, and the host-only helpers agree with the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.build_native()
    return ua3reo_ddc


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ua3reo_ddc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(ddc):
    L = ddc.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export " + n


def test_pack_unpack_adc12_matches_oracle(ddc, orc):
    rng = np.random.default_rng(0)
    x = rng.integers(-2048, 2048, 1000).astype(np.int16)
    p = ddc.pack_adc12(x)
    assert np.array_equal(p, orc.pack12(x))
    assert np.array_equal(ddc.unpack_adc12(p, 1000), x)
    assert np.array_equal(orc.unpack12(p, 1000), x)


def test_phrase_math_matches_oracle(ddc, orc):
    for f in (0, 1, 1000000, 7100000, 14200000, 24576000, 24576001, 30000000, 49152000, 50000000,
              145000000, 433000000):
        assert ddc.phrase_from_frequency(f) == orc.phrase_from_freq(f)
    for ph in (0, 1, 605867, 2097152, 4194303):
        assert ddc.frequency_from_phrase(ph) == orc.freq_from_phrase(ph)
    assert ddc.phrase_from_frequency(7100000, 65000000) == orc.phrase_from_freq(7100000, 65000000)


def test_bus_frames_match_oracle(ddc, orc):
    rng = np.random.default_rng(1)
    spec = rng.integers(-32768, 32768, (5, 2)).astype(np.int16)
    voice = rng.integers(-32768, 32768, (5, 2)).astype(np.int16)
    got = ddc.pack_rx_frames(spec, voice).reshape(5, 8)
    for n in range(5):
        want = orc.pack_rx_frame(int(spec[n, 0]), int(spec[n, 1]), int(voice[n, 0]), int(voice[n, 1]))
        assert bytes(got[n]) == want
    tx = rng.integers(0, 256, 12).astype(np.uint8)
    iq = ddc.unpack_tx_frames(tx)
    for n in range(3):
        assert tuple(iq[n]) == orc.unpack_tx_frame(bytes(tx[4 * n:4 * n + 4]))
    assert ddc.parse_params(bytes([0b1100, 0x09, 0x3E, 0xAB])) == orc.parse_params(bytes([0b1100, 0x09, 0x3E, 0xAB]))


def test_synthetic_signal_definition(ddc):
    x = ddc.synth_adc(3, 4096)
    assert x.dtype == np.int16 and abs(int(x.max())) <= 2047 and abs(int(x.min())) <= 2047
    # two tones near 1.144 MHz dominate; noise is +-64
    n = np.arange(4096)
    f1 = 1.0e6 + 3 * 48e3
    tone = 1400 * np.sin(2 * np.pi * f1 * n / 49152000) + 300 * np.sin(2 * np.pi * (f1 + 1e3) * n / 49152000)
    assert np.max(np.abs(x - np.round(tone))) <= 64
    assert np.array_equal(x, ddc.synth_adc(3, 4096))
    assert not np.array_equal(x, ddc.synth_adc(4, 4096))


def test_parse_status_frame():
    import ua3reo_ddc
    # min = -1234 (0xB2E), max = 1999 (0x7CF), ADC over-range set
    frame = bytes([0x01, (0xB << 4) | 0x7, 0x2E, 0xCF, 0])
    assert ua3reo_ddc.parse_status(frame) == (1, 0, -1234, 1999)
    # the reset value of ADC_MAX (-2000) reads back unsigned, as the firmware takes it
    frame = bytes([0x02, (0x7 << 4) | 0x8, 0xD0, 0x30, 0])
    assert ua3reo_ddc.parse_status(frame) == (0, 1, 2000, 0x830)


def legacy_symbols():
    """Functions and extern objects ua3reo_legacy.h declares."""
    txt = open(os.path.join(ROOT, "include", "ua3reo_legacy.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct \{.*?\} VFO;", "", txt, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt))
    names -= {"defined", "FPGA_AUDIO_BUFFER_SIZE", "FPGA_AUDIO_BUFFER_HALF_SIZE"}
    for decl in re.findall(r"extern\s+(?!\"C\")[^;{]*;", txt):
        for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*(,|;)", decl):
            names.add(m.group(1))
    names.add("TRX")
    return sorted(n for n in names if not n.isupper() or n == "TRX")


def test_library_exports_the_firmware_interface(ddc):
    L = ddc.lib()
    names = legacy_symbols()
    for must in ("processRxAudio", "processTxAudio", "initAudioProcessor", "FFT_Init", "FFT_doFFT", "InitAudioFilters",
                 "ReinitAudioFilters", "InitNotchFilter", "dc_filter", "InitAGC", "DoAGC", "InitNoiseReduction",
                 "processNoiseReduction", "CWDecoder_Init", "CWDecoder_Process", "getPhraseFromFrequency",
                 "getFrequencyFromPhrase", "FPGA_Init", "FPGA_fpgadata_iqclock", "FPGA_fpgadata_stuffclock",
                 "FPGA_Audio_Buffer_SPEC_I", "FPGA_Audio_Buffer_VOICE_Q", "FPGA_Audio_SendBuffer_I",
                 "FPGA_Audio_Buffer_Index", "FPGA_Audio_Buffer_I_tmp", "Processor_AudioBuffer_A", "Processor_AudioBuffer_B",
                 "FFTInput_I", "FFTInput_Q", "TRX"):
        assert must in names, must
    for n in names:
        assert hasattr(L, n), "missing export " + n


def test_autogain_search_matches_oracle_and_the_firmware_story(ddc, orc):
    """ddc_autogain_step against the restated TRX_DoAutoGain on random and scripted
    amplitude sequences, and the walk the firmware comments describe."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        g, ref = ddc.AutoGain(), [0, 0, 0, 0]
        amps = rng.choice([0, 30, 100, 250, 280, 400, 1000, 1099, 1100, 1101, 1500, 2047, -5], 200)
        for a in amps:
            before = ref[2:4]
            orc.autogain_step(ref, a)
            changed = g.step(a)
            st = g.state
            assert [st["stage"], st["wait_reaction"], int(st["preamp"]), int(st["att"])] == ref
            assert changed == (before != ref[2:4])
    # a quiet band: ATT first, then 7 calm seconds per step up to PREAMP without ATT
    g = ddc.AutoGain()
    seen = []
    for _ in range(40):
        g.step(20)
        s = g.state
        seen.append((s["preamp"], s["att"]))
    assert seen[0] == (False, True) and seen[-1] == (True, False)
    assert seen.index((False, False)) == 8 and seen.index((True, True)) == 16 and seen.index((True, False)) == 24
    # a strong signal appears: one setting back
    g.step(1500)
    assert g.state["stage"] == 4
    g.step(1500)
    assert (g.state["preamp"], g.state["att"]) == (True, True)
    # 276 * 10^(12/20) = 1098.8 <= 1100 counts as calm, 277 does not (stage 1)
    g = ddc.AutoGain()
    g.step(0)
    for _ in range(6):
        g.step(276)
    assert g.state["wait_reaction"] == 6
    g.step(277)
    assert g.state["wait_reaction"] == 0 and g.state["stage"] == 1


def test_legacy_autogain_drives_the_trx_fields(ddc):
    """TRX_DoAutoGain of the firmware-style shim: host-only, works on TRX and TRX_ADC_MAXAMPLITUDE."""
    import ctypes as C
    from tests.legacy_binding import Legacy
    fw = Legacy(ddc.lib())
    amp = fw.var("TRX_ADC_MAXAMPLITUDE", C.c_int16)
    fw.TRX.AutoGain = False
    fw.TRX.Preamp, fw.TRX.ATT = True, False
    fw.L.TRX_DoAutoGain()
    assert (fw.TRX.Preamp, fw.TRX.ATT) == (True, False)  # feature off: untouched
    fw.TRX.AutoGain = True
    amp.value = 10
    fw.L.TRX_DoAutoGain()
    assert (fw.TRX.Preamp, fw.TRX.ATT, fw.TRX.LPF, fw.TRX.BPF) == (False, True, True, True)
    for _ in range(8):
        fw.L.TRX_DoAutoGain()
    assert (fw.TRX.Preamp, fw.TRX.ATT) == (False, False)
    fw.TRX.AutoGain = False


def test_mode_policy_is_trx_setmode(ddc, orc):
    """ddc_audio_params_for_mode against TRX_setMode (trx_manager.c:193-220) for every mode:
    SSB / DIGI / AM take the SSB width, CW the CW width, NFM the FM width, WFM 0, the rest
    keep what the VFO had; no other field moves."""
    import ctypes as C
    L = ddc.lib()
    for mode in range(12):
        p = ddc.audio_params(mode="USB", filter_width=1234, volume=77, notch_fc=900)
        before = bytes(p)
        assert L.ddc_audio_params_for_mode(mode, 2700, 500, 15000, C.byref(p)) == 0
        assert p.mode == mode
        assert p.filter_width == orc.params_for_mode(mode, 2700, 500, 15000, 1234)
        p.mode, p.filter_width = 1, 1234
        assert bytes(p) == before
    want = {0: 2700, 1: 2700, 2: 1234, 3: 500, 4: 500, 5: 2700, 6: 2700, 7: 1234, 8: 15000, 9: 0, 10: 2700, 11: 1234}
    assert {m: orc.params_for_mode(m, 2700, 500, 15000, 1234) for m in range(12)} == want
    p = ddc.audio_params()
    assert L.ddc_audio_params_for_mode(12, 2700, 500, 15000, C.byref(p)) != 0


def test_oracle_usb_frames_known_answers(orc):
    """audio_processor.c:415-432 by hand: the factor is (1 / Volume) * 100 in float32, the
    product is cut towards zero, 32-bit saturation comes before the cut to 16 bits."""
    x = np.array([0, 1000, -1000, 3, -3, 16383, 40000, -40000, 2 ** 31 - 1, -2 ** 31], np.int32)
    assert list(orc.usb_frames(x, 50)) == [0, 2000, -2000, 6, -6, 32766, 80000 - 65536, 65536 - 80000, -1, 0]
    assert list(orc.usb_frames(x, 100)[:6]) == [0, 1000, -1000, 3, -3, 16383]
    g = np.float32(np.float32(1.0) / np.float32(30)) * np.float32(100)
    assert list(orc.usb_frames(x[:6], 30)) == [((int(np.float32(v) * g) + 32768) % 65536) - 32768 for v in x[:6]]
    assert list(orc.usb_frames(np.array([0, 5, -5], np.int32), 0)) == [0, -1, 0]   # 0 * inf = NaN -> 0; +-inf saturate


def test_shard_range_is_the_host_logic_of_the_gloo_tests(ddc):
    """ddc_shard_range (what a C host partitions with) = ua3reo_ddc.shard.channel_range (what the
    gloo tests and the benchmark partition with): contiguous, covering, first ranks one larger."""
    from ua3reo_ddc import shard
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 8, 1023, 8192):
            at = 0
            for r in range(world):
                lo, hi = ddc.shard_range(r, world, total)
                assert (lo, hi) == shard.channel_range(r, world, total)
                assert lo == at
                at = hi
            assert at == total
    assert ddc.shard_range(3, 3, 10) == (0, 0)  # rank out of range: empty
