/*
 * ua3reo_legacy.h -- the firmware's own global-buffer, void(void) processing
 * interface, kept for hosts that were written against STM32/Src, implemented on
 * channel 0 of a one-channel ddc_ctx (include/ua3reo_ddc.h).  Names, argument
 * meaning and buffer shapes are the firmware's; every declaration cites it.
 *
 * The FPGA used to see the ADC and DAC pins itself; here the host pushes ADC
 * words in and pops DAC words out (ddc_legacy_adc_push / ddc_legacy_dac_pop).
 * Everything else is driven exactly as the interrupt handlers of
 * STM32/Src/stm32f4xx_it.c drive it: FPGA_fpgadata_iqclock once per 48 kHz tick,
 * processRxAudio / processTxAudio when the Need*Buffer flags rise, FFT_doFFT when
 * 512 samples have been collected.
 *
 * This is a compatibility surface, not a fast path: one tick moves one I/Q frame
 * through the GPU.  Throughput users call the batch API.
 */
#ifndef UA3REO_LEGACY_H
#define UA3REO_LEGACY_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float float32_t;

#define FPGA_AUDIO_BUFFER_SIZE (192 * 2)                        /* audio_processor.h:10 */
#define FPGA_AUDIO_BUFFER_HALF_SIZE (FPGA_AUDIO_BUFFER_SIZE / 2) /* audio_processor.h:11 */
#define APROCESSOR_BLOCK_SIZE 64                                /* audio_processor.h:12 */
#define FFT_SIZE 512                                            /* fft.h:10 */
#define FFT_PRINT_SIZE 256                                      /* fft.h:12 */
#define CWDECODER_STRLEN 15                                     /* cw_decoder.h:15 */

/* trx_manager.h:11-22 */
#define TRX_MODE_LSB 0
#define TRX_MODE_USB 1
#define TRX_MODE_IQ 2
#define TRX_MODE_CW_L 3
#define TRX_MODE_CW_U 4
#define TRX_MODE_DIGI_L 5
#define TRX_MODE_DIGI_U 6
#define TRX_MODE_NO_TX 7
#define TRX_MODE_NFM 8
#define TRX_MODE_WFM 9
#define TRX_MODE_AM 10
#define TRX_MODE_LOOPBACK 11

/* The fields of TRX_SETTINGS (settings.h:57-123) that the sample path reads, under
 * their firmware names.  Source-compatible for those fields; the display, touch
 * pad and band-memory fields of the firmware's struct are not carried. */
typedef struct {
    uint32_t Freq;
    uint8_t Mode;
    uint16_t Filter_Width;
} VFO;

extern struct TRX_SETTINGS {
    bool current_vfo; /* false - A; true - B */
    VFO VFO_A;
    VFO VFO_B;
    bool AGC;
    bool Preamp;
    bool DNR;
    uint8_t Agc_speed;
    uint8_t Volume;
    bool Mute;
    uint8_t RF_Power;
    uint8_t FM_SQL_threshold;
    uint8_t RF_Gain;
    uint8_t FFT_Zoom;
    bool NotchFilter;
    uint16_t NotchFC;
    bool CWDecoder;
    bool FFT_Enabled;
    uint16_t CW_GENERATOR_SHIFT_HZ;
    uint8_t FFT_Averaging;
    uint16_t SSB_HPF_pass;
    bool AutoGain; /* settings.h:90 */
    bool ATT, LPF, BPF;
    uint16_t CW_Filter, SSB_Filter, FM_Filter; /* settings.h:82-84: the widths TRX_setMode hands to the VFO */
} TRX;

/* trx_manager.h: transceiver state the path reads */
extern volatile bool TRX_ptt_hard, TRX_ptt_cat, TRX_key_serial, TRX_key_hard, TRX_tune, TRX_IQ_swap;
extern volatile bool TRX_ADC_OTR, TRX_DAC_OTR;
extern volatile int16_t TRX_ADC_MINAMPLITUDE, TRX_ADC_MAXAMPLITUDE;
VFO *CurrentVFO(void);    /* settings.c */
uint8_t TRX_getMode(void); /* trx_manager.c */
void TRX_setMode(uint8_t _mode); /* trx_manager.c:193-220: mode -> Filter_Width, then ReinitAudioFilters */
void TRX_DoAutoGain(void);  /* trx_manager.c:268: once a second, on TRX_ADC_MAXAMPLITUDE; sets TRX.Preamp / TRX.ATT */
bool TRX_on_TX(void);      /* trx_manager.c:  ptt | tune | CAT ptt | key */

/* fpga.h:12-31 */
void FPGA_Init(void);
void FPGA_fpgadata_iqclock(void);
void FPGA_fpgadata_stuffclock(void);
extern volatile bool FPGA_busy, FPGA_Buffer_underrun, FPGA_NeedSendParams, FPGA_NeedGetParams;
extern volatile uint32_t FPGA_samples;
extern float32_t FPGA_Audio_Buffer_SPEC_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_Buffer_SPEC_I[FPGA_AUDIO_BUFFER_SIZE];
extern float32_t FPGA_Audio_Buffer_VOICE_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_Buffer_VOICE_I[FPGA_AUDIO_BUFFER_SIZE];
extern float32_t FPGA_Audio_SendBuffer_Q[FPGA_AUDIO_BUFFER_SIZE], FPGA_Audio_SendBuffer_I[FPGA_AUDIO_BUFFER_SIZE];
extern uint16_t FPGA_Audio_Buffer_Index;
extern bool FPGA_Audio_Buffer_State;

/* audio_processor.h:26-43 */
void initAudioProcessor(void);
void processRxAudio(void);
void processTxAudio(void);
extern volatile uint32_t AUDIOPROC_samples, AUDIOPROC_TXA_samples, AUDIOPROC_TXB_samples;
extern int32_t Processor_AudioBuffer_A[FPGA_AUDIO_BUFFER_SIZE], Processor_AudioBuffer_B[FPGA_AUDIO_BUFFER_SIZE];
extern volatile uint8_t Processor_AudioBuffer_ReadyBuffer;
extern volatile bool Processor_NeedRXBuffer, Processor_NeedTXBuffer;
extern volatile float32_t Processor_TX_MAX_amplitude, ALC_need_gain;
extern volatile float32_t Processor_RX_Audio_Samples_MAX_value, Processor_RX_Audio_Samples_MIN_value;
extern float32_t FPGA_Audio_Buffer_Q_tmp[FPGA_AUDIO_BUFFER_HALF_SIZE], FPGA_Audio_Buffer_I_tmp[FPGA_AUDIO_BUFFER_HALF_SIZE];

/* usbd_audio_if.h:14-18, usbd_ua3reo.h:70: the USB-audio side of processRxAudio
 * (audio_processor.c:415-432).  While USB_AUDIO_need_rx_buffer is set, processRxAudio
 * fills buffer a (USB_AUDIO_current_rx_buffer false) or b with the int16 frames of the
 * block and clears the flag. */
#define USB_AUDIO_RX_BUFFER_SIZE (FPGA_AUDIO_BUFFER_SIZE * 2)
extern int16_t USB_AUDIO_rx_buffer_a[USB_AUDIO_RX_BUFFER_SIZE / 2], USB_AUDIO_rx_buffer_b[USB_AUDIO_RX_BUFFER_SIZE / 2];
extern volatile bool USB_AUDIO_current_rx_buffer, USB_AUDIO_need_rx_buffer;

/* fft.h:25-35.  FFT_doFFT leaves the smoothed bins in FFTOutput_mean and the
 * newest waterfall row (RGB565, fft-shifted) in FFT_waterfall_row, where the
 * firmware's FFT_printFFT would have put it into wtf_buffer[0]. */
void FFT_Init(void);
void FFT_doFFT(void);
extern volatile uint32_t FFT_buff_index;
extern bool NeedFFTInputBuffer, FFT_need_fft;
extern float32_t FFTInput_I[FFT_SIZE], FFTInput_Q[FFT_SIZE];
extern float32_t FFTOutput_mean[FFT_PRINT_SIZE];
extern uint16_t FFT_waterfall_row[FFT_PRINT_SIZE];

/* audio_filters.h:48-51, agc.h:9-10, noise_reduction.h:15-16, cw_decoder.h:17-21.
 * The Init / Reinit functions all re-derive the channel's configuration from TRX
 * and clear the audio state (the firmware clears only the state each one owns). */
void InitAudioFilters(void);
void ReinitAudioFilters(void);
void InitNotchFilter(void);
void dc_filter(float32_t *agcBuffer, int16_t blockSize, uint8_t stateNum);
void InitAGC(void);
void DoAGC(float32_t *agcBuffer, int16_t blockSize);
void InitNoiseReduction(void);
void processNoiseReduction(float32_t *bufferIn, float32_t *bufferOut);
void CWDecoder_Init(void);
void CWDecoder_Process(float32_t *bufferIn);
extern volatile uint16_t CW_Decoder_WPM;
extern char CW_Decoder_Text[CWDECODER_STRLEN + 1];

/* functions.h:42-43 */
uint32_t getPhraseFromFrequency(uint32_t freq);
uint32_t getFrequencyFromPhrase(uint32_t phrase);

/* What the FPGA's pins were: ADC words in (12-bit two's complement in int16, 1024
 * per 48 kHz tick), DAC words out (14-bit offset binary, 1024 per tick in TX).
 * push returns the number of words accepted, pop the number delivered. */
size_t ddc_legacy_adc_push(const int16_t *adc, size_t n);
size_t ddc_legacy_dac_pop(uint16_t *dac, size_t n);
/* The millisecond clock CWDecoder_Process reads (HAL_GetTick() in the firmware);
 * the shim advances it by 4 per processRxAudio call unless the host sets it. */
extern volatile uint32_t ddc_legacy_tick_ms;
/* Last error text of the shim's ctx (empty when none). */
const char *ddc_legacy_last_error(void);
void ddc_legacy_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
