/*
 * The firmware's global-buffer interface (include/ua3reo_legacy.h) on top of the
 * batch C ABI: channel 0 of a one-channel ctx.  Host code only; every GPU call goes
 * through include/ua3reo_ddc.h.
 */
#include "../../include/ua3reo_legacy.h"
#include "../../include/ua3reo_ddc.h"

#include <cstring>
#include <deque>
#include <string>
#include <vector>

extern "C" {

struct TRX_SETTINGS TRX = {
    false, { 7100000, TRX_MODE_LSB, 2700 }, { 14150000, TRX_MODE_USB, 2700 },
    true, false, false, 3, 50, false, 100, 1, 50, 1, false, 1000, false, true, 500, 4, 300,
    false, false, false, false, 500, 2700, 15000 }; // widths: settings.c:54-56
volatile bool TRX_ptt_hard = false, TRX_ptt_cat = false, TRX_key_serial = false, TRX_key_hard = false, TRX_tune = false,
              TRX_IQ_swap = false;
volatile bool TRX_ADC_OTR = false, TRX_DAC_OTR = false;
volatile int16_t TRX_ADC_MINAMPLITUDE = 0, TRX_ADC_MAXAMPLITUDE = 0;

volatile bool FPGA_busy = false, FPGA_Buffer_underrun = false, FPGA_NeedSendParams = false, FPGA_NeedGetParams = false;
volatile uint32_t FPGA_samples = 0;
float32_t FPGA_Audio_Buffer_SPEC_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_Buffer_SPEC_I[FPGA_AUDIO_BUFFER_SIZE];
float32_t FPGA_Audio_Buffer_VOICE_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_Buffer_VOICE_I[FPGA_AUDIO_BUFFER_SIZE];
float32_t FPGA_Audio_SendBuffer_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_SendBuffer_I[FPGA_AUDIO_BUFFER_SIZE];
uint16_t FPGA_Audio_Buffer_Index = 0;
bool FPGA_Audio_Buffer_State = true;

volatile uint32_t AUDIOPROC_samples = 0, AUDIOPROC_TXA_samples = 0, AUDIOPROC_TXB_samples = 0;
int32_t Processor_AudioBuffer_A[FPGA_AUDIO_BUFFER_SIZE], Processor_AudioBuffer_B[FPGA_AUDIO_BUFFER_SIZE];
volatile uint8_t Processor_AudioBuffer_ReadyBuffer = 0;
volatile bool Processor_NeedRXBuffer = false, Processor_NeedTXBuffer = false;
volatile float32_t Processor_TX_MAX_amplitude = 0.0f, ALC_need_gain = 1.0f;
volatile float32_t Processor_RX_Audio_Samples_MAX_value = 0.0f, Processor_RX_Audio_Samples_MIN_value = 0.0f;
float32_t FPGA_Audio_Buffer_Q_tmp[FPGA_AUDIO_BUFFER_HALF_SIZE], FPGA_Audio_Buffer_I_tmp[FPGA_AUDIO_BUFFER_HALF_SIZE];
int16_t USB_AUDIO_rx_buffer_a[USB_AUDIO_RX_BUFFER_SIZE / 2], USB_AUDIO_rx_buffer_b[USB_AUDIO_RX_BUFFER_SIZE / 2];
volatile bool USB_AUDIO_current_rx_buffer = false, USB_AUDIO_need_rx_buffer = false;

volatile uint32_t FFT_buff_index = 0;
bool NeedFFTInputBuffer = false, FFT_need_fft = true;
float32_t FFTInput_I[FFT_SIZE], FFTInput_Q[FFT_SIZE];
float32_t FFTOutput_mean[FFT_PRINT_SIZE];
uint16_t FFT_waterfall_row[FFT_PRINT_SIZE];

volatile uint16_t CW_Decoder_WPM = 0;
char CW_Decoder_Text[CWDECODER_STRLEN + 1] = "               ";
volatile uint32_t ddc_legacy_tick_ms = 0;

} // extern "C"

namespace {

ddc_ctx *g_ctx = nullptr;
std::string g_err;
std::deque<int16_t> g_adc;
std::deque<uint16_t> g_dac;
ddc_audio_params g_cfg;
bool g_cfg_valid = false;
bool g_last_tx = false;

bool check(int rc)
{
    if (rc == 0) return true;
    g_err = ddc_last_error(g_ctx);
    return false;
}

bool ensure_ctx()
{
    if (g_ctx) return true;
    ddc_config cfg = { 1, 0, 0, 0 };
    if (ddc_create(&cfg, &g_ctx) != 0) {
        g_err = ddc_last_error(nullptr);
        g_ctx = nullptr;
        return false;
    }
    return true;
}

// The audio-path view of TRX and the transceiver state (what processRxAudio /
// processTxAudio / FFT_doFFT read).  The I/Q swap is applied where the firmware
// applies it, when the frames are stored (fpga.c:305-385), not here.
ddc_audio_params params_now()
{
    ddc_audio_params p;
    memset(&p, 0, sizeof p);
    p.mode = TRX_getMode();
    p.agc = TRX.AGC;
    p.dnr = TRX.DNR;
    p.notch = TRX.NotchFilter;
    p.mute = TRX.Mute;
    p.cw_decoder = TRX.CWDecoder;
    p.fm_sql_threshold = TRX.FM_SQL_threshold;
    p.volume = TRX.Volume;
    p.iq_swap = 0;
    p.fft_averaging = TRX.FFT_Averaging ? TRX.FFT_Averaging : 1;
    p.fft_zoom = TRX.FFT_Zoom > 1 ? TRX.FFT_Zoom : 1;
    p.rf_power = TRX.RF_Power;
    p.tune = TRX_tune;
    p.key = TRX_key_serial || TRX_ptt_hard || TRX_key_hard;
    p.filter_width = CurrentVFO()->Filter_Width;
    p.hpf_pass = TRX.SSB_HPF_pass;
    p.notch_fc = TRX.NotchFC;
    p.agc_speed = TRX.Agc_speed ? TRX.Agc_speed : 1;
    p.rf_gain = (float)TRX.RF_Gain;
    return p;
}

bool selects_same_filters(const ddc_audio_params &a, const ddc_audio_params &b)
{
    return a.mode == b.mode && a.filter_width == b.filter_width && a.hpf_pass == b.hpf_pass && a.notch_fc == b.notch_fc &&
           a.fft_zoom == b.fft_zoom;
}

bool configure(bool force)
{
    if (!ensure_ctx()) return false;
    const ddc_audio_params p = params_now();
    if (force || !g_cfg_valid || !selects_same_filters(p, g_cfg)) {
        if (!check(ddc_audio_configure(g_ctx, 0, &p))) return false;
    } else if (memcmp(&p, &g_cfg, sizeof p) != 0) {
        if (!check(ddc_audio_update(g_ctx, 0, &p))) return false;
    }
    g_cfg = p;
    g_cfg_valid = true;
    return true;
}

// readHalfFromCircleBuffer32 (functions.c:21-34): the half buffer that ends at index
void read_half(const float *src, float *dst, uint32_t index)
{
    const uint32_t len = FPGA_AUDIO_BUFFER_SIZE, half = len / 2;
    if (index >= half)
        memcpy(dst, src + index - half, half * sizeof(float));
    else {
        const uint32_t prev = half - index;
        memcpy(dst, src + len - prev, prev * sizeof(float));
        memcpy(dst + prev, src, (half - prev) * sizeof(float));
    }
}

void pull_status()
{
    ddc_audio_status st;
    if (!check(ddc_audio_get_status(g_ctx, 0, &st))) return;
    Processor_RX_Audio_Samples_MAX_value = st.smeter_max;
    Processor_RX_Audio_Samples_MIN_value = st.smeter_min;
    ALC_need_gain = st.alc_gain;
    Processor_TX_MAX_amplitude = st.tx_max_amplitude;
    CW_Decoder_WPM = (uint16_t)st.cw_wpm;
    memcpy(CW_Decoder_Text, st.cw_text, 16);
}

} // namespace

extern "C" {

const char *ddc_legacy_last_error(void) { return g_err.c_str(); }

void ddc_legacy_shutdown(void)
{
    if (g_ctx) ddc_destroy(g_ctx);
    g_ctx = nullptr;
    g_cfg_valid = false;
    g_adc.clear();
    g_dac.clear();
}

size_t ddc_legacy_adc_push(const int16_t *adc, size_t n)
{
    g_adc.insert(g_adc.end(), adc, adc + n);
    return n;
}

size_t ddc_legacy_dac_pop(uint16_t *dac, size_t n)
{
    size_t k = 0;
    while (k < n && !g_dac.empty()) {
        dac[k++] = g_dac.front();
        g_dac.pop_front();
    }
    return k;
}

VFO *CurrentVFO(void) { return TRX.current_vfo ? &TRX.VFO_B : &TRX.VFO_A; }
uint8_t TRX_getMode(void) { return CurrentVFO()->Mode; }

// trx_manager.c:193-220 (the loopback start, LCD and settings-save flags are the firmware's own business)
void TRX_setMode(uint8_t _mode)
{
    CurrentVFO()->Mode = _mode;
    ddc_audio_params p;
    memset(&p, 0, sizeof p);
    p.filter_width = CurrentVFO()->Filter_Width;
    if (ddc_audio_params_for_mode(_mode, TRX.SSB_Filter, TRX.CW_Filter, TRX.FM_Filter, &p) == 0)
        CurrentVFO()->Filter_Width = p.filter_width;
    ReinitAudioFilters();
}

static ddc_autogain g_autogain;
void TRX_DoAutoGain(void)
{
    if (!TRX.AutoGain) return;
    TRX.LPF = true;
    TRX.BPF = true;
    g_autogain.preamp = TRX.Preamp;
    g_autogain.att = TRX.ATT;
    ddc_autogain_step(&g_autogain, TRX_ADC_MAXAMPLITUDE);
    TRX.Preamp = g_autogain.preamp;
    TRX.ATT = g_autogain.att;
}
bool TRX_on_TX(void) { return TRX_ptt_hard || TRX_ptt_cat || TRX_tune || TRX_getMode() == TRX_MODE_LOOPBACK; }

uint32_t getPhraseFromFrequency(uint32_t freq)
{
    int sw = 0;
    const uint32_t v = ddc_phrase_from_frequency(freq, 0, &sw);
    TRX_IQ_swap = sw != 0;
    return v;
}

uint32_t getFrequencyFromPhrase(uint32_t phrase) { return ddc_frequency_from_phrase(phrase, 0); }

void FPGA_Init(void)
{
    ddc_legacy_shutdown();
    ensure_ctx();
    FPGA_Audio_Buffer_Index = 0;
    FPGA_NeedSendParams = true;
}

// FPGA_fpgadata_sendparam / _getparam (fpga.c:173-284) as one call each
void FPGA_fpgadata_stuffclock(void)
{
    if (!ensure_ctx()) return;
    FPGA_busy = true;
    if (FPGA_NeedSendParams) {
        uint32_t f = CurrentVFO()->Freq;
        if (!TRX_on_TX()) {
            if (TRX_getMode() == TRX_MODE_CW_L) f += TRX.CW_GENERATOR_SHIFT_HZ;
            if (TRX_getMode() == TRX_MODE_CW_U) f -= TRX.CW_GENERATOR_SHIFT_HZ;
        }
        const uint32_t phrase = getPhraseFromFrequency(f);
        uint8_t frame[4] = { 0, (uint8_t)((phrase >> 16) & 0xFF), (uint8_t)((phrase >> 8) & 0xFF), (uint8_t)(phrase & 0xFF) };
        const bool tx = TRX_on_TX() && TRX_getMode() != TRX_MODE_LOOPBACK;
        if (tx) frame[0] |= 1u << 3;
        if (!TRX_on_TX() && TRX.Preamp) frame[0] |= 1u << 2;
        check(ddc_set_params(g_ctx, 0, frame));
        FPGA_NeedSendParams = false;
    } else if (FPGA_NeedGetParams) {
        uint8_t st[5];
        if (check(ddc_get_status(g_ctx, 0, st))) {
            int a, d, lo, hi;
            ddc_parse_status(st, &a, &d, &lo, &hi);
            TRX_ADC_OTR = a != 0;
            TRX_DAC_OTR = d != 0;
            TRX_ADC_MINAMPLITUDE = (int16_t)lo;
            TRX_ADC_MAXAMPLITUDE = (int16_t)hi;
        }
        FPGA_NeedGetParams = false;
    }
    FPGA_busy = false;
}

// One 48 kHz tick: command 4 (RX frame in) or command 3 (TX frame out), fpga.c:148-171
void FPGA_fpgadata_iqclock(void)
{
    if (!ensure_ctx()) return;
    FPGA_busy = true;
    const bool tx = TRX_on_TX() && TRX_getMode() != TRX_MODE_LOOPBACK;
    if (tx != g_last_tx) { // the FPGA holds the idle direction in reset
        if (tx) check(ddc_tx_reset(g_ctx, 0));
        else check(ddc_rx_reset(g_ctx, 0));
        g_last_tx = tx;
    }
    FPGA_samples = FPGA_samples + 1;
    if (!tx) { // FPGA_fpgadata_getiq, fpga.c:286-401
        if (g_adc.size() < DDC_DECIMATION) { FPGA_Buffer_underrun = true; FPGA_busy = false; return; }
        int16_t adc[DDC_DECIMATION];
        for (unsigned i = 0; i < DDC_DECIMATION; i++) { adc[i] = g_adc.front(); g_adc.pop_front(); }
        uint8_t packed[DDC_DECIMATION / 2 * 3];
        ddc_pack_adc12(adc, DDC_DECIMATION, packed);
        int16_t spec[2], voice[2];
        if (!check(ddc_rx_process(g_ctx, packed, sizeof packed, DDC_DECIMATION, spec, voice))) { FPGA_busy = false; return; }
        const uint16_t k = FPGA_Audio_Buffer_Index;
        const bool sw = TRX_IQ_swap;
        (sw ? FPGA_Audio_Buffer_SPEC_I : FPGA_Audio_Buffer_SPEC_Q)[k] = spec[1];
        (sw ? FPGA_Audio_Buffer_SPEC_Q : FPGA_Audio_Buffer_SPEC_I)[k] = spec[0];
        (sw ? FPGA_Audio_Buffer_VOICE_I : FPGA_Audio_Buffer_VOICE_Q)[k] = voice[1];
        (sw ? FPGA_Audio_Buffer_VOICE_Q : FPGA_Audio_Buffer_VOICE_I)[k] = voice[0];
        if (NeedFFTInputBuffer) {
            (sw ? FFTInput_I : FFTInput_Q)[FFT_buff_index] = spec[1];
            (sw ? FFTInput_Q : FFTInput_I)[FFT_buff_index] = spec[0];
        }
        FPGA_Audio_Buffer_Index++;
        if (FPGA_Audio_Buffer_Index == FPGA_AUDIO_BUFFER_SIZE) FPGA_Audio_Buffer_Index = 0;
        if (NeedFFTInputBuffer) {
            FFT_buff_index = FFT_buff_index + 1;
            if (FFT_buff_index == FFT_SIZE) { FFT_buff_index = 0; NeedFFTInputBuffer = false; }
        }
    } else { // FPGA_fpgadata_sendiq, fpga.c:403-466
        float fi = FPGA_Audio_SendBuffer_I[FPGA_Audio_Buffer_Index], fq = FPGA_Audio_SendBuffer_Q[FPGA_Audio_Buffer_Index];
        // float -> int16 as the Cortex-M4 does it: saturating convert to 32 bit, keep the low half
        auto narrow = [](float v) -> int16_t {
            if (!(v == v)) return 0;
            const int32_t w = v >= 2147483648.0f ? INT32_MAX : v <= -2147483648.0f ? INT32_MIN : (int32_t)v;
            return (int16_t)w;
        };
        int16_t iq[2] = { narrow(fi), narrow(fq) };
        std::vector<uint16_t> dac(DDC_TX_INTERPOLATION);
        if (check(ddc_tx_process(g_ctx, iq, 1, dac.data()))) g_dac.insert(g_dac.end(), dac.begin(), dac.end());
        FPGA_Audio_Buffer_Index++;
        if (FPGA_Audio_Buffer_Index == FPGA_AUDIO_BUFFER_SIZE) {
            if (Processor_NeedTXBuffer) { FPGA_Buffer_underrun = true; FPGA_Audio_Buffer_Index--; }
            else { FPGA_Audio_Buffer_Index = 0; FPGA_Audio_Buffer_State = true; Processor_NeedTXBuffer = true; }
        } else if (FPGA_Audio_Buffer_Index == FPGA_AUDIO_BUFFER_SIZE / 2) {
            if (Processor_NeedTXBuffer) { FPGA_Buffer_underrun = true; FPGA_Audio_Buffer_Index--; }
            else { FPGA_Audio_Buffer_State = false; Processor_NeedTXBuffer = true; }
        }
    }
    FPGA_busy = false;
}

void initAudioProcessor(void) { configure(true); }
void InitAudioFilters(void) { configure(true); }
void ReinitAudioFilters(void) { configure(true); }
void InitNotchFilter(void) { configure(true); }
void InitAGC(void) { configure(true); }
void InitNoiseReduction(void) { configure(true); }
void CWDecoder_Init(void) { configure(true); }
void FFT_Init(void) { configure(true); }

// audio_processor.c:275-435
void processRxAudio(void)
{
    if (!Processor_NeedRXBuffer) return;
    if (!configure(false)) return;
    AUDIOPROC_samples = AUDIOPROC_samples + 1;
    uint32_t idx = FPGA_Audio_Buffer_Index;
    idx = idx == 0 ? FPGA_AUDIO_BUFFER_SIZE : idx - 1;
    float si[192], sq[192], vi[192], vq[192];
    read_half(FPGA_Audio_Buffer_SPEC_I, si, idx);
    read_half(FPGA_Audio_Buffer_SPEC_Q, sq, idx);
    read_half(FPGA_Audio_Buffer_VOICE_I, vi, idx);
    read_half(FPGA_Audio_Buffer_VOICE_Q, vq, idx);
    int16_t spec[192][2], voice[192][2];
    for (int i = 0; i < 192; i++) {
        spec[i][0] = (int16_t)si[i]; spec[i][1] = (int16_t)sq[i];
        voice[i][0] = (int16_t)vi[i]; voice[i][1] = (int16_t)vq[i];
    }
    int32_t out[192][2];
    if (!check(ddc_audio_rx_process(g_ctx, &spec[0][0], &voice[0][0], 192, &out[0][0]))) return;
    int32_t *dst = Processor_AudioBuffer_ReadyBuffer == 0 ? Processor_AudioBuffer_B : Processor_AudioBuffer_A;
    memcpy(dst, out, sizeof out);
    Processor_AudioBuffer_ReadyBuffer = Processor_AudioBuffer_ReadyBuffer == 0 ? 1 : 0;
    for (int i = 0; i < 192; i++) { FPGA_Audio_Buffer_I_tmp[i] = (float)out[i][0]; FPGA_Audio_Buffer_Q_tmp[i] = (float)out[i][1]; }
    if (USB_AUDIO_need_rx_buffer) { // audio_processor.c:415-432
        int16_t *usb = USB_AUDIO_current_rx_buffer ? USB_AUDIO_rx_buffer_b : USB_AUDIO_rx_buffer_a;
        if (check(ddc_audio_usb_frames(g_ctx, &out[0][0], 192, usb))) USB_AUDIO_need_rx_buffer = false;
    }
    pull_status();
    ddc_legacy_tick_ms = ddc_legacy_tick_ms + 4;
    Processor_NeedRXBuffer = false;
}

// audio_processor.c:61-273; the host puts the 192 stereo capture frames into
// Processor_AudioBuffer_A, where the firmware's codec / USB read leaves them
void processTxAudio(void)
{
    if (!Processor_NeedTXBuffer) return;
    if (!configure(false)) return;
    AUDIOPROC_samples = AUDIOPROC_samples + 1;
    int16_t mic[192][2], iq[192][2];
    int32_t loop[192][2];
    for (int i = 0; i < 192; i++) {
        mic[i][0] = (int16_t)Processor_AudioBuffer_A[2 * i];
        mic[i][1] = (int16_t)Processor_AudioBuffer_A[2 * i + 1];
    }
    if (!check(ddc_audio_tx_process(g_ctx, &mic[0][0], 192, &iq[0][0], &loop[0][0]))) return;
    for (int i = 0; i < 192; i++) { FPGA_Audio_Buffer_I_tmp[i] = iq[i][0]; FPGA_Audio_Buffer_Q_tmp[i] = iq[i][1]; }
    if (TRX_getMode() == TRX_MODE_LOOPBACK && !TRX_tune) {
        memcpy(Processor_AudioBuffer_A, loop, sizeof loop);
    } else {
        const int off = FPGA_Audio_Buffer_State ? FPGA_AUDIO_BUFFER_HALF_SIZE : 0;
        for (int i = 0; i < 192; i++) {
            FPGA_Audio_SendBuffer_I[off + i] = iq[i][0];
            FPGA_Audio_SendBuffer_Q[off + i] = iq[i][1];
        }
        if (FPGA_Audio_Buffer_State) AUDIOPROC_TXA_samples = AUDIOPROC_TXA_samples + 1;
        else AUDIOPROC_TXB_samples = AUDIOPROC_TXB_samples + 1;
    }
    pull_status();
    Processor_NeedTXBuffer = false;
    Processor_NeedRXBuffer = false;
}

// fft.c:212-331 + the colour row of :361-379
void FFT_doFFT(void)
{
    if (!TRX.FFT_Enabled) return;
    if (!FFT_need_fft) return;
    if (NeedFFTInputBuffer) return;
    if (!configure(false)) return;
    int16_t spec[FFT_SIZE][2];
    for (int i = 0; i < FFT_SIZE; i++) { spec[i][0] = (int16_t)FFTInput_I[i]; spec[i][1] = (int16_t)FFTInput_Q[i]; }
    NeedFFTInputBuffer = true;
    if (!check(ddc_fft_process(g_ctx, &spec[0][0], FFT_SIZE, FFTOutput_mean, FFT_waterfall_row))) return;
    FFT_need_fft = false;
}

void dc_filter(float32_t *agcBuffer, int16_t blockSize, uint8_t stateNum)
{
    if (!configure(false)) return;
    check(ddc_audio_stage(g_ctx, 0, DDC_STAGE_DC, stateNum, agcBuffer, blockSize));
}

void DoAGC(float32_t *agcBuffer, int16_t blockSize)
{
    if (!configure(false)) return;
    check(ddc_audio_stage(g_ctx, 0, DDC_STAGE_AGC, 0, agcBuffer, blockSize));
}

void processNoiseReduction(float32_t *bufferIn, float32_t *bufferOut)
{
    if (!configure(false)) return;
    float tmp[APROCESSOR_BLOCK_SIZE];
    memcpy(tmp, bufferIn, sizeof tmp);
    if (check(ddc_audio_stage(g_ctx, 0, DDC_STAGE_DNR, 0, tmp, APROCESSOR_BLOCK_SIZE))) memcpy(bufferOut, tmp, sizeof tmp);
}

void CWDecoder_Process(float32_t *bufferIn)
{
    if (!configure(false)) return;
    float tmp[192];
    memcpy(tmp, bufferIn, sizeof tmp);
    if (check(ddc_audio_stage(g_ctx, 0, DDC_STAGE_CW, (int)ddc_legacy_tick_ms, tmp, 192))) pull_status();
}

} // extern "C"
