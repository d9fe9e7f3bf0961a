/*
 * ua3reo_ddc.h -- C ABI of the MI355X DDC/DUC signal chain.
 *
 * One ddc_ctx owns the per-channel state of N independent tuned channels on one
 * GPU: NCO phase, CIC window moments, FIR delay lines.  Every entry point
 * returns 0 on success or a negative errno-style code; ddc_last_error() gives
 * the text.  A ctx is single-owner (not thread-safe) and runs on one HIP
 * stream; the *_dev calls are asynchronous on that stream, the host-pointer
 * calls return when the results are in the caller's buffers.
 *
 * The reference has no FFI layer: its boundary is the FPGA<->MCU byte contract
 * (FPGA/stm32_interface.v) and the global-buffer C functions of STM32/Src.
 * Each declaration below names the reference interface it stands in for.
 */
#ifndef UA3REO_DDC_H
#define UA3REO_DDC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDC_ADC_CLOCK_DEFAULT 49152000u /* STM32/Src/settings.h:10 ADCDAC_CLOCK */
#define DDC_DECIMATION 1024u            /* rx_cic 512 x rx_ciccomp 2 */
#define DDC_TX_INTERPOLATION 1024u      /* tx_ciccomp 2 x tx_cic 512 */

typedef struct ddc_ctx ddc_ctx;

typedef struct ddc_config {
    uint32_t n_channels; /* independent tuned channels held by this ctx */
    uint32_t fs_adc;     /* ADC/DAC clock in Hz; 0 = DDC_ADC_CLOCK_DEFAULT */
    int32_t device;      /* HIP device ordinal */
    uint32_t reserved;
} ddc_config;

/* Library / device life cycle.  Replaces FPGA_Init (STM32/Src/fpga.c:29-55):
 * allocates the per-channel state in HBM, zeroed as the FPGA is after reset. */
int ddc_create(const ddc_config *cfg, ddc_ctx **out);
void ddc_destroy(ddc_ctx *ctx);
const char *ddc_last_error(const ddc_ctx *ctx);
const char *ddc_version(void);

/* Tuning.  ddc_tune writes the 22-bit phase increment the way command 1 of the
 * bus does (FPGA/stm32_interface.v:157-171 -> NCO phi_inc_i).  ddc_tune_hz is
 * getPhraseFromFrequency (STM32/Src/functions.c:206-226): folds the frequency
 * into the first Nyquist zone and reports TRX_IQ_swap through *iq_swap.
 * channel = -1 applies to all channels. */
int ddc_tune(ddc_ctx *ctx, int32_t channel, uint32_t phi_inc22);
int ddc_tune_hz(ddc_ctx *ctx, int32_t channel, uint32_t freq_hz, int *iq_swap);
/* One 22-bit word per channel in one transfer (n must equal n_channels). */
int ddc_tune_many(ddc_ctx *ctx, const uint32_t *phi_inc22, uint32_t n);
uint32_t ddc_phrase_from_frequency(uint32_t freq_hz, uint32_t fs_adc, int *iq_swap);
uint32_t ddc_frequency_from_phrase(uint32_t phrase, uint32_t fs_adc); /* functions.c:199-204 */

/* Zero one channel's (or, with -1, every channel's) RX datapath state the
 * way the FPGA's reset = RX_N does on every TX->RX switch
 * (FPGA/UA3REO.bdf clk_enable/reset wiring); the tuning word is kept. */
int ddc_rx_reset(ddc_ctx *ctx, int32_t channel);
/* Direct access to the NCO phase accumulator (22 bit) for state save/restore. */
int ddc_rx_set_phase(ddc_ctx *ctx, int32_t channel, uint32_t phase22);
int ddc_rx_get_phase(ddc_ctx *ctx, int32_t channel, uint32_t *phase22);

/* RX: ADC words -> SPEC and VOICE I/Q at fs/1024.
 * Replaces the FPGA RX datapath nco + mixer.v + rx_cic.vhd + rx_ciccomp.vhd +
 * rx_hilb.vhd + data_delay.v as wired in FPGA/UA3REO.bdf, i.e. what command 4
 * of the bus (FPGA/stm32_interface.v:228-271) hands to
 * FPGA_fpgadata_getiq (STM32/Src/fpga.c:286-401) once per 48 kHz tick.
 *
 *   adc12          packed signed 12-bit words, two per three bytes (b0=s0[7:0],
 *                  b1=s0[11:8]|s1[3:0]<<4, b2=s1[11:4]); one row per channel,
 *                  rows row_bytes apart (row_bytes a multiple of 16)
 *                  row_bytes = 0: one ADC stream shared by every channel (the
 *                  reference's case: one converter, here fanned out to N tunings)
 *   samples        ADC samples per channel in this call, a multiple of 1024
 *   spec_iq        [channel][samples/1024][2] int16: SPEC_I, SPEC_Q
 *   voice_iq       [channel][samples/1024][2] int16: VOICE_I, VOICE_Q
 * State carries over from call to call exactly as if the samples had been
 * clocked through in one piece.  _dev takes device pointers and queues the work
 * on the ctx stream; the plain form takes host pointers and waits. */
int ddc_rx_process_dev(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                       int16_t *spec_iq, int16_t *voice_iq);
int ddc_rx_process(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                   int16_t *spec_iq, int16_t *voice_iq);
/* ddc_rx_process_dev for callers that must not allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): same
 * arguments, same results, same carried state, but the work buffers of the call
 * (window moments, rx_cic and compensator outputs) live in one fixed work area
 * of ddc_rx_static_scratch_bytes() = 32 MiB per GPU inside the library instead
 * of buffers that grow with the largest call.  No counterpart in the reference.
 * The call allocates and frees nothing and records no timing events.
 * -E2BIG (with the byte counts in ddc_last_error) when n_channels x samples needs
 * more than the area holds: ddc_rx_static_scratch_need() tells beforehand
 * (86 bytes per channel per 512 samples plus 1680 bytes per channel, each of the
 * three buffers rounded up to 256 bytes).
 * Every ctx on a GPU shares that GPU's area: calls of this entry point that
 * overlap in time on one GPU (two ctxs on two streams) race; calls on one stream,
 * or ordered by the caller, do not.  Mixing it with ddc_rx_process_dev on the
 * same ctx is fine: the carried state lives in the ctx, not in the area.
 * Every piece of carried state (NCO phase, mixer carries, filter histories) is
 * advanced by the kernels on the device, none by the host: one call captured in a
 * HIP graph on the stream given to ddc_set_stream can be replayed for chunk after
 * chunk of the same shape, with fresh input written into the same buffers. */
int ddc_rx_process_dev_static(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                              int16_t *spec_iq, int16_t *voice_iq);
size_t ddc_rx_static_scratch_bytes(void);
size_t ddc_rx_static_scratch_need(uint32_t n_channels, size_t samples);
/* Debug tap: the raw rx_cic outputs of the last ddc_rx_process* call,
 * [channel][samples/512][2] int16 (host pointer).  After ddc_rx_process_dev_static
 * the outputs sit in the GPU's shared work area: read them before another ctx makes
 * a static call on that GPU. */
int ddc_rx_read_cic(ddc_ctx *ctx, int16_t *cic_iq, size_t samples);
/* Debug switch.  An RX call whose samples / 512 is a multiple of 256 runs as two
 * kernels (the ingest kernel forms the window sums of rx_cic on the matrix cores),
 * a smaller call as six smaller kernels, and any other size as both: the largest
 * multiple of 131072 samples the first way, the remainder the second.  The results
 * are the same bit for bit and the carried state is common, so calls (and parts of
 * calls) of both kinds may alternate.
 * plan = 1 makes every call of this ctx use the six kernels, plan = 2 keeps the
 * choice by size but runs the two-kernel parts with the ingest kernel that stays on
 * the vector ALU (both for A/B timing and for tests), plan = 0 restores the default. */
int ddc_debug_rx_plan(ddc_ctx *ctx, int plan);

/* TX: 48 kHz I/Q -> DAC words at 1024x the rate.
 * Replaces the FPGA TX datapath tx_ciccomp.vhd + tx_cic.vhd + tx_mixer.v +
 * tx_summator.v + DAC_corrector.v as wired in FPGA/UA3REO.bdf, fed by command 3
 * of the bus (FPGA/stm32_interface.v:206-227 <- FPGA_fpgadata_sendiq,
 * STM32/Src/fpga.c:403-466) and driving DAC_OUTPUT[13..0].
 *
 *   iq48k   [channel][n_iq][2] int16: TX_I, TX_Q, one pair per 48 kHz tick
 *   dac14   [channel][1024 * n_iq] uint16: 14-bit offset-binary DAC words
 *           (8191 = mid scale), DAC word t of the call at index t
 * The carrier is the channel's tuning word (ddc_tune); state carries over from
 * call to call.  The reference's DAC_OTR flag (adder overflow) can never rise:
 * |tx_cic| <= 8192 and |NCO| <= 8191 keep the 28-bit sum in range. */
int ddc_tx_process_dev(ddc_ctx *ctx, const int16_t *iq48k, size_t n_iq, uint16_t *dac14);
int ddc_tx_process(ddc_ctx *ctx, const int16_t *iq48k, size_t n_iq, uint16_t *dac14);
/* Zero the TX datapath state (reset = TX_N on every RX->TX switch). */
int ddc_tx_reset(ddc_ctx *ctx, int32_t channel);
int ddc_tx_set_phase(ddc_ctx *ctx, int32_t channel, uint32_t phase22);

/* Multi-GPU.  Channels are independent, so a job shards them over GPUs (one process and one
 * ctx per GPU, contiguous channel ranges) and nothing is exchanged inside the chain; the one
 * collective is the all-gather of the decimated outputs, the stream the reference sends over its
 * 8-bit bus to the consumer of STM32/Src/fpga.c:286-401, here delivered to every rank.  No
 * counterpart in the reference (one FPGA, one MCU).
 *
 * ddc_shard_range: the channels [*lo, *hi) of `rank` out of `world` ranks; the first
 * total_channels % world ranks hold one more.
 * ddc_comm_unique_id / ddc_comm_init: an RCCL communicator over `world` ranks for this ctx's
 * GPU (ncclGetUniqueId on one rank, the 128 bytes carried to the others by the host, then
 * ncclCommInitRank on every rank: a collective call).  ddc_comm_attach uses a communicator the
 * host already owns (an ncclComm_t of the RCCL instance loaded in this process; the ctx does not
 * destroy it).  RCCL is bound when one of these is first called (librccl.so.1); -ENOSYS when it
 * cannot be loaded.
 * ddc_allgather_decimated_dev: all-gather of this rank's decimated outputs, local =
 * [2 (SPEC, VOICE)][n_channels][n_frames][2] int16 in one piece (e.g. the two output buffers of
 * ddc_rx_process_dev allocated back to back), into gathered = [world][2][n_channels][n_frames][2]
 * on every rank (device pointers; every rank must hold the same n_channels and pass the same
 * n_frames).  One ncclAllGather of bytes on a stream of the ctx's own, started after the work
 * queued on the ctx stream so far, so it overlaps whatever the ctx is given next.  Two slots for
 * double buffering: step k computes into buffer k & 1 and gathers with slot k & 1;
 * ddc_allgather_reuse(slot) makes the ctx stream wait for that slot's last gather, call it
 * before queueing the work that overwrites the slot's `local` buffer; ddc_allgather_wait(slot)
 * waits on the host until `gathered` of that slot is complete (-1: both) and reports the
 * duration of the last gather waited for in milliseconds (gather_ms may be NULL). */
#define DDC_COMM_ID_BYTES 128
void ddc_shard_range(uint32_t rank, uint32_t world, uint32_t total_channels, uint32_t *lo, uint32_t *hi);
int ddc_comm_unique_id(uint8_t id[DDC_COMM_ID_BYTES]);
int ddc_comm_init(ddc_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t id[DDC_COMM_ID_BYTES]);
int ddc_comm_attach(ddc_ctx *ctx, void *rccl_comm, uint32_t world, uint32_t rank);
int ddc_comm_destroy(ddc_ctx *ctx);
int ddc_allgather_decimated_dev(ddc_ctx *ctx, int slot, const int16_t *local, size_t n_frames, int16_t *gathered);
int ddc_allgather_reuse(ddc_ctx *ctx, int slot);
int ddc_allgather_wait(ddc_ctx *ctx, int slot, float *gather_ms);

/* Wait for everything queued on the ctx stream. */
int ddc_sync(ddc_ctx *ctx);
/* The HIP stream of the ctx (a hipStream_t), for callers that time or chain work. */
void *ddc_stream(ddc_ctx *ctx);
/* Run the ctx on a stream of the caller's (a hipStream_t; NULL = back to the ctx's own
 * stream): every later call enqueues there, so work the caller queued on that stream
 * before (filling the input) and after (reading the output) is ordered without an event.
 * Waits for what the ctx still has in flight on its previous stream.  The caller keeps
 * ownership of its stream and must not destroy it while the ctx uses it.  No counterpart
 * in the reference (the firmware has one flow of control). */
int ddc_set_stream(ddc_ctx *ctx, void *hip_stream);

/* Timing taps (the reference's analogue is STM32/Src/profiler.c:21-88): when
 * enabled, every ddc_rx_process_dev call records HIP events on the ctx stream
 * around its ADC-ingest kernel and around the rest of the call.
 * ddc_profile_collect waits for the stream and returns the summed durations in
 * milliseconds of the (at most 256) calls since the last collect. */
int ddc_profile_enable(ddc_ctx *ctx, int on);
int ddc_profile_collect(ddc_ctx *ctx, double *front_ms, double *tail_ms, uint32_t *n_calls);

/* Bus frames, byte-identical to the FPGA side of FPGA/stm32_interface.v.
 * ddc_pack_rx_frames: command 4 reply, 8 bytes per frame {SPEC_Q hi,lo, SPEC_I
 * hi,lo, VOICE_Q hi,lo, VOICE_I hi,lo} (:228-271).  ddc_unpack_tx_frames:
 * command 3 payload, 4 bytes {Q hi,lo, I hi,lo} (:206-227).  ddc_parse_params:
 * command 1 payload {flags(bit3 TX, bit2 preamp), freq[21:16], [15:8], [7:0]}
 * (:142-171).  Host memory, no ctx needed. */
void ddc_pack_rx_frames(const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames, uint8_t *out);
void ddc_unpack_tx_frames(const uint8_t *in, size_t n_frames, int16_t *tx_iq);
void ddc_parse_params(const uint8_t in[4], int *tx, int *preamp, uint32_t *phi_inc22);
/* Applies a command-1 payload to a channel (-1 = all): the tuning word, and the RX/TX
 * flag as the FPGA uses it (clock enable of one direction, reset of the other): when
 * the flag differs from the channel's last one, the direction that is switched on is
 * reset to zero state (ddc_tx_reset / ddc_rx_reset).  A ctx starts in RX.  The preamp
 * bit belongs to the RF unit and is ignored.  The flag gates nothing else: a
 * ddc_rx_process* or ddc_tx_process* call runs whatever the channel's flag says (in the
 * FPGA the idle direction has no clock), and ddc_tune / ddc_rx_reset / ddc_tx_reset do
 * not touch the flag; a host that mixes directions per channel skips the calls itself. */
int ddc_set_params(ddc_ctx *ctx, int32_t channel, const uint8_t in[4]);

/* Status frame: command 2 reply of the bus (FPGA/stm32_interface.v:172-205), read
 * by FPGA_fpgadata_getparam (STM32/Src/fpga.c:222-284): {bit0 ADC_OTR, bit1
 * DAC_OTR, bits 2-5 keys}, {ADC_MIN[11:8] << 4 | ADC_MAX[11:8]}, ADC_MIN[7:0],
 * ADC_MAX[7:0], {encoder}.  ADC_MIN / ADC_MAX (:384-397) are the extremes of the
 * channel's ADC words since the last status read (reset to 2000 / -2000 by the
 * read); they are gathered by an extra pass over the rows of every
 * ddc_rx_process* call while ddc_adc_stats_enable is on (off by default).  Keys
 * and encoder read 0; DAC_OTR cannot rise (see ddc_tx_process). */
int ddc_adc_stats_enable(ddc_ctx *ctx, int on);
int ddc_get_status(ddc_ctx *ctx, int32_t channel, uint8_t out[5]);
void ddc_parse_status(const uint8_t in[5], int *adc_otr, int *dac_otr, int *adc_min, int *adc_max);

/* Whole-ctx state (every channel's NCO phases, tuning words, filter histories,
 * audio / FFT / TX state) as one host blob, for checkpoint / migration between
 * GPUs.  The reference keeps this state in FPGA registers and MCU statics and has
 * no counterpart.  A blob only restores into a ctx with the same channel count;
 * work queued on the ctx stream is waited for first. */
size_t ddc_state_size(ddc_ctx *ctx);
int ddc_state_save(ddc_ctx *ctx, void *buf, size_t size);
int ddc_state_restore(ddc_ctx *ctx, const void *buf, size_t size);

/* ---- 48 kHz audio path and panadapter (the STM32 half of the reference) ------
 *
 * Demodulator modes, numbered as STM32/Src/trx_manager.h:11-22. */
enum ddc_mode {
    DDC_MODE_LSB = 0, DDC_MODE_USB, DDC_MODE_IQ, DDC_MODE_CW_L, DDC_MODE_CW_U, DDC_MODE_DIGI_L,
    DDC_MODE_DIGI_U, DDC_MODE_NO_TX, DDC_MODE_NFM, DDC_MODE_WFM, DDC_MODE_AM, DDC_MODE_LOOPBACK
};

/* Per-channel operator settings: the fields of TRX / TRX.VFO the audio path reads
 * (STM32/Src/settings.h:38-101).  filter_width is the LPF corner in Hz (0 = LPF
 * off), hpf_pass the SSB high-pass corner (100..500, used in LSB/USB only),
 * notch_fc the manual notch frequency, agc_speed 1..n, rf_gain the linear gain
 * the firmware multiplies in (TRX.RF_Gain), volume 0..100, fft_averaging >= 1,
 * fft_zoom 1 (off), 2, 4, 8 or 16 (TRX.FFT_Zoom), rf_power 0..100 (TRX.RF_Power),
 * tune = TRX_tune, key = CW key / PTT down (TRX_key_serial | TRX_ptt_hard |
 * TRX_key_hard). */
typedef struct ddc_audio_params {
    uint8_t mode, agc, dnr, notch, mute, cw_decoder, fm_sql_threshold, volume;
    uint8_t iq_swap, fft_averaging, fft_zoom, rf_power;
    uint8_t tune, key, pad0, pad1;
    uint16_t filter_width, hpf_pass, notch_fc, agc_speed;
    float rf_gain;
} ddc_audio_params;

typedef struct ddc_audio_status {
    float smeter_max, smeter_min; /* Processor_RX_Audio_Samples_MAX/MIN_value, audio_processor.c:396-421 */
    float agc_gain;               /* AGC_need_gain, agc.c */
    float fft_max_value;          /* maxValueFFT, fft.c */
    uint32_t cw_wpm;              /* CW_Decoder_WPM */
    uint32_t squelched;           /* FM squelch state */
    uint32_t blocks;              /* 192-frame RX blocks processed since configure */
    float alc_gain;               /* ALC_need_gain, audio_processor.c:34 */
    float tx_max_amplitude;       /* Processor_TX_MAX_amplitude */
    char cw_text[16];             /* CW_Decoder_Text: last 15 decoded characters */
} ddc_audio_status;

/* TRX_setMode (STM32/Src/trx_manager.c:193-220): the mode -> Filter_Width policy of
 * the transceiver.  Sets p->mode and p->filter_width from the three widths the
 * operator keeps (TRX.SSB_Filter for LSB / USB / DIGI_L / DIGI_U / AM, TRX.CW_Filter
 * for CW_L / CW_U, TRX.FM_Filter for NFM, 0 for WFM; IQ, NO_TX and LOOPBACK leave the
 * width as it is) and touches no other field; follow it with ddc_audio_configure,
 * as the firmware follows it with ReinitAudioFilters.  -EINVAL for a mode > 11. */
int ddc_audio_params_for_mode(uint8_t mode, uint16_t ssb_filter, uint16_t cw_filter, uint16_t fm_filter,
                              ddc_audio_params *p);

/* initAudioProcessor + ReinitAudioFilters + InitNotchFilter + InitAGC +
 * InitNoiseReduction + CWDecoder_Init for `channel` (-1 = all): filter selection
 * (audio_filters.c:141-339), notch design (calcBiquad :375-494), state zeroed.
 * Must be called before the first audio / FFT call.  -EINVAL when filter_width or
 * hpf_pass name no table of the firmware. */
int ddc_audio_configure(ddc_ctx *ctx, int32_t channel, const ddc_audio_params *p);
/* The waterfall of one channel (wtf_buffer, fft.c:28, :353-376): the colour rows of
 * the last 50 frames, newest first, [50][256] RGB565 (host pointer; waits for the
 * stream).  ddc_fft_move_waterfall is FFT_moveWaterfall (fft.c:453-503): after a
 * retune by freq_diff_hz the picture and the smoothed bins move sideways by
 * (freq_diff_hz / 187) * fft_zoom pixels, the bins in the firmware's in-place order. */
int ddc_fft_get_waterfall(ddc_ctx *ctx, int32_t channel, uint16_t *rows);
int ddc_fft_move_waterfall(ddc_ctx *ctx, int32_t channel, int32_t freq_diff_hz);
/* Changes the settings that select no filter (agc, dnr, notch on/off, mute,
 * cw_decoder, fm_sql_threshold, volume, iq_swap, fft_averaging, rf_power, tune, key,
 * agc_speed, rf_gain) and keeps all running state, as the firmware does when it
 * reads TRX.* on the next block.  mode, filter_width, hpf_pass, notch_fc and
 * fft_zoom must equal the configured ones (-EINVAL otherwise: use
 * ddc_audio_configure). */
int ddc_audio_update(ddc_ctx *ctx, int32_t channel, const ddc_audio_params *p);
/* processRxAudio (audio_processor.c:275-435) for all channels.  spec_iq/voice_iq:
 * [n_channels][n_frames][2] int16 as ddc_rx_process_dev writes them; n_frames a
 * multiple of 192 (the firmware's half-buffer); audio_lr: [n_channels][n_frames][2]
 * int32, left then right, the codec buffer contents before the 16-bit word swap
 * of the I2S DMA.  The _dev form takes device pointers and is asynchronous on the
 * ctx stream. */
int ddc_audio_rx_process_dev(ddc_ctx *ctx, const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames,
                             int32_t *audio_lr);
int ddc_audio_rx_process(ddc_ctx *ctx, const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames,
                         int32_t *audio_lr);
/* The USB-audio frames of the RX path (audio_processor.c:415-432, USB_AUDIO_rx_buffer_a/b):
 * one int16 per int32 entry of audio_lr, (int16_t)((int32_t)v * (1.0f / volume * 100.0f)),
 * i.e. the codec samples with the volume setting taken out again; the cast is the
 * Cortex-M4F's (towards zero and saturating in 32 bits, then the low 16 bits; volume 0
 * gives an infinite factor: 0 stays 0, other samples become -1 / 0).  audio_lr as
 * ddc_audio_rx_process* writes it, [n_channels][n_frames][2]; usb_lr the same shape in
 * int16; n_frames even.  Uses the volume of the channel's current audio parameters. */
int ddc_audio_usb_frames_dev(ddc_ctx *ctx, const int32_t *audio_lr, size_t n_frames, int16_t *usb_lr);
int ddc_audio_usb_frames(ddc_ctx *ctx, const int32_t *audio_lr, size_t n_frames, int16_t *usb_lr);
/* FFT_doFFT (fft.c:212-331, zoom 1) + the colour row of FFT_printFFT (:361-379,
 * :506-539) over consecutive 512-sample frames of SPEC.  n_frames a multiple of
 * 512; bins: [n_channels][n_frames/512][256] float (FFTOutput_mean after each
 * frame); rows: [n_channels][n_frames/512][256] RGB565 waterfall rows, already
 * fft-shifted.  Either output may be NULL.  With fft_zoom = M > 1 every frame is
 * low-passed, decimated by M and slid into the last 512 decimated samples first
 * (fft.c:236-261), so one row still comes out per 512 input samples. */
int ddc_fft_process_dev(ddc_ctx *ctx, const int16_t *spec_iq, size_t n_frames, float *bins, uint16_t *rows);
int ddc_fft_process(ddc_ctx *ctx, const int16_t *spec_iq, size_t n_frames, float *bins, uint16_t *rows);
/* processTxAudio (audio_processor.c:61-273) + ModulateFM (:590-619) for all
 * channels.  mic_lr: [n_channels][n_frames][2] int16, the codec / USB-audio
 * capture (left, right); n_frames a multiple of 192; iq48k: [n_channels][n_frames][2]
 * int16 (I, Q) as FPGA_fpgadata_sendiq narrows them for the bus (fpga.c:403-436),
 * ready for ddc_tx_process; loop_lr: [n_channels][n_frames][2] int32, the codec
 * output of LOOPBACK mode (zero in other modes; iq48k is zero in LOOPBACK), may
 * be NULL.  Shares the HPF / LPF lattice state and DC-blocker slots with nothing
 * on the RX side except the filter instances the firmware itself shares
 * (IIR_HPF_I, IIR_LPF_I). */
int ddc_audio_tx_process_dev(ddc_ctx *ctx, const int16_t *mic_lr, size_t n_frames, int16_t *iq48k, int32_t *loop_lr);
int ddc_audio_tx_process(ddc_ctx *ctx, const int16_t *mic_lr, size_t n_frames, int16_t *iq48k, int32_t *loop_lr);
/* One stage of the RX audio path on one channel's state, for hosts that still call
 * the firmware's fine-grained functions: dc_filter (audio_filters.h:51, slot in
 * arg), DoAGC (agc.h:10, n = 192), processNoiseReduction (noise_reduction.h:16,
 * n = 64, in place), CWDecoder_Process (cw_decoder.h:21, n = 192, arg = the
 * millisecond clock HAL_GetTick() would return).  buf is host memory, processed
 * in place; the call waits for the result. */
enum ddc_audio_stage_id { DDC_STAGE_DC = 0, DDC_STAGE_AGC = 1, DDC_STAGE_DNR = 2, DDC_STAGE_CW = 3 };
int ddc_audio_stage(ddc_ctx *ctx, int32_t channel, int stage, int arg, float *buf, int n);
/* The S-meter reading the firmware takes every 100 ms (stm32f4xx_it.c:398-411):
 * peak-to-peak of the audio samples since the last reading -> dBm at the antenna
 * input (TRX_RX_dBm), with the firmware's log10f_fast; clears the extremes. */
int ddc_audio_smeter_dbm(ddc_ctx *ctx, int32_t channel, float *dbm);
/* The once-a-second front-end gain search of the firmware, TRX_DoAutoGain
 * (trx_manager.c:268-356; AUTOGAIN_MAX_AMPLITUDE 1100 settings.h:25, ATT_DB 12 and
 * PREAMP_GAIN_DB 20 settings.h:23-24, AUTOGAIN_CORRECTOR_WAITSTEP 7 trx_manager.h:9):
 * from the ADC's largest amplitude since the last status read (TRX_ADC_MAXAMPLITUDE,
 * byte 3-4 of ddc_get_status) it walks ATT -> none -> PREAMP+ATT -> PREAMP, waiting
 * seven quiet calls before every step up and falling back when the ADC comes close
 * to its rail.  Host-side state machine, one per channel, zero-initialised; the
 * caller switches its relays (or scales its ADC rows) after `preamp` and `att`.
 * Returns 1 when the call changed preamp / att, else 0. */
typedef struct ddc_autogain {
    uint8_t stage, wait_reaction; /* autogain_stage, autogain_wait_reaction (trx_manager.c:35-36) */
    uint8_t preamp, att;          /* TRX.Preamp, TRX.ATT as the search leaves them */
} ddc_autogain;
int ddc_autogain_step(ddc_autogain *st, int16_t adc_max_amplitude);
/* S-meter extremes, AGC / ALC gain, CW decoder text (waits for the ctx stream). */
int ddc_audio_get_status(ddc_ctx *ctx, int32_t channel, ddc_audio_status *out);

/* Read-out of a constant table as compiled into this library (host memory, no GPU
 * needed), so that a test can compare what ships with the reference's numbers:
 * "rx_ciccomp_h" int16[65] (FPGA/rx_ciccomp.vhd:81-146 interleaved), "rx_hilb_c"
 * int16[256] (rx_hilb.vhd:77-332), "tx_ciccomp_p1" / "tx_ciccomp_p2" int16[24]
 * (tx_ciccomp.vhd:81-128), "tx_hilbert_i" / "tx_hilbert_q" float[201], "hpf_k"
 * float[6][6], "hpf_v" float[6][7], "lpf_k" float[32][11], "lpf_v" float[32][12],
 * "lpf_stages" / "hpf_hz" / "lpf_hz" int[], "sql_hpf_k" float[6], "sql_hpf_v" float[7]
 * (STM32/Src/audio_filters.c:41-126), "zoom_biquad" double[4][20], "zoom_fir"
 * double[4][4] (fft.c:72-183), "nco_coarse" uint32[2048][2] ({2 cc | 2 sc << 16,
 * -2 sc | 2 cc << 16} of the coarse ROMs), "nco_fine_sine" int16[2048] (the kernels'
 * arithmetic form of the fine sine ROM).  out = NULL only reports the size. */
int ddc_debug_table(const char *name, void *out, size_t cap, size_t *bytes);
/* Debug tap: the NCO exactly as the kernels evaluate it (coarse table, arithmetic fine sine,
 * angle sums with the rounding of nco_nco_ii_0.v:370-391), for `count` consecutive 22-bit
 * phase-accumulator values from first_phase (wrapping), into host arrays: sin14 / cos14 by the TX
 * kernel's code (all 14 bits, FPGA/tx_mixer.v), sin12 / cos12 (may be NULL) by the RX kernel's
 * code (14 bits >> 2, FPGA/nco_shift.v:9). */
int ddc_debug_nco(ddc_ctx *ctx, uint32_t first_phase, uint32_t count, int16_t *sin14, int16_t *cos14,
                  int16_t *sin12, int16_t *cos12);

/* Packed-ADC helpers and the synthetic test signal of the benchmark
 * (two tones + uniform noise, xorshift64* seeded 0x5EED0000 + channel). */
void ddc_pack_adc12(const int16_t *samples, size_t n, uint8_t *packed);
void ddc_unpack_adc12(const uint8_t *packed, size_t n, int16_t *samples);
void ddc_synth_adc(uint32_t channel, uint32_t fs_adc, size_t n, int16_t *samples);
/* The same signal generated on the GPU, packed: rows[c] (device memory of the current
 * device, rows row_bytes apart) receives ddc_pack_adc12(ddc_synth_adc(first_channel + c))
 * for c < n_channels; samples a multiple of 512.  Allocates a small scratch buffer and
 * waits for hip_stream (a hipStream_t, NULL = the default stream).  Bench / test helper. */
int ddc_synth_adc_dev(uint32_t first_channel, uint32_t n_channels, uint32_t fs_adc, size_t samples, uint8_t *rows,
                      size_t row_bytes, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
