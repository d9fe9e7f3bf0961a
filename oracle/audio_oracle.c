/*
 * audio_oracle.c -- CPU restatement of the UA3REO STM32 48 kHz RX audio path and
 * FFT panadapter, one channel, block by block as the firmware runs it.
 *
 * TEST INFRASTRUCTURE ONLY (see ddc_oracle.c).  PARITY UNPINNED BY THE REFERENCE:
 * the firmware has no tests and all its filter/FFT arithmetic lives in the
 * CMSIS-DSP 1.6.0 library (Keil pack ARM.CMSIS 5.5.1, not in the reference
 * tree).  The arm_* routines below restate the published CMSIS-DSP algorithms
 * (lattice-ladder IIR, transposed direct form II biquad, normalised LMS,
 * complex FFT + magnitude, table-interpolated arm_sin/cos_f32) in their documented
 * operation order; the complex FFT is a radix-2 one in double precision, not the
 * library's radix-8 in float, which is why FFT results carry a tolerance instead of
 * bit equality.
 *
 * Build with -ffp-contract=off: every float operation rounds on its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_AUDIO_TABLES
#include "oracle_tables.h"

#define BLOCK 192          /* FPGA_AUDIO_BUFFER_HALF_SIZE, audio_processor.h:10-11 */
#define SUB 64             /* APROCESSOR_BLOCK_SIZE, audio_processor.h:12 */
#define FFT_SIZE 512       /* fft.h:10 */
#define FFT_PRINT 256      /* fft.h:12 */
#define PI_F 3.14159265358979f /* arm_math.h PI */

enum { MODE_LSB, MODE_USB, MODE_IQ, MODE_CW_L, MODE_CW_U, MODE_DIGI_L, MODE_DIGI_U, MODE_NO_TX,
       MODE_NFM, MODE_WFM, MODE_AM, MODE_LOOPBACK }; /* trx_manager.h:11-22 */

typedef struct {
    uint8_t mode, agc, dnr, notch, mute, cw_decoder, fm_sql_threshold, volume;
    uint8_t iq_swap, fft_averaging, fft_zoom, rf_power;
    uint8_t tune, key, pad0, pad1;
    uint16_t filter_width, hpf_pass, notch_fc, agc_speed;
    float rf_gain;
} orc_audio_params_t;

/* ---- CMSIS-DSP restatements ------------------------------------------------ */

/* arm_sin_f32 / arm_cos_f32 (CMSIS-DSP 1.6.0 FastMathFunctions): x in radians is scaled
 * to turns, the fraction of a turn times 512 gives a table index and a remainder, and the
 * result is the linear interpolation of two neighbouring entries of sinTable_f32[513]. */
static float arm_table_lookup(float in)
{
    int32_t n = (int32_t)in;
    if (in < 0.0f) n--;              /* floor */
    in = in - (float)n;
    float findex = 512.0f * in;
    uint16_t index = (uint16_t)findex;
    if (index >= 512) { index = 0; findex -= 512.0f; }
    const float fract = findex - (float)index;
    const float a = arm_sin_table[index], b = arm_sin_table[index + 1];
    return (1.0f - fract) * a + fract * b;
}
float orc_arm_sin_f32(float x) { return arm_table_lookup(x * 0.159154943092f); }
float orc_arm_cos_f32(float x) { return arm_table_lookup(x * 0.159154943092f + 0.25f); }
const float *orc_arm_sin_table(void) { return arm_sin_table; }

/* arm_iir_lattice_f32: N stages, reversed reflection (k) and ladder (v) tables,
 * state g[0..N-1] = backward residuals of the previous sample, newest stage first. */
typedef struct { int n; const float *k, *v; float g[12]; } lattice_t;

static void lattice_init(lattice_t *s, int n, const float *k, const float *v)
{
    s->n = n; s->k = k; s->v = v;
    memset(s->g, 0, sizeof s->g);
}

static void lattice_run(lattice_t *s, float *buf, int len)
{
    const int n = s->n;
    for (int t = 0; t < len; t++) {
        float fcurr = buf[t], fnext = fcurr, acc = 0.0f;
        float ng[12];
        for (int i = 0; i < n; i++) {
            const float gcurr = s->g[i];
            fnext = fcurr - s->k[i] * gcurr;
            const float gnext = fnext * s->k[i] + gcurr;
            acc += gnext * s->v[i];
            ng[i] = gnext;
            fcurr = fnext;
        }
        acc += fnext * s->v[n];
        /* the state window slides by one: g'[i] = gnext of stage i+1, last = f0 */
        for (int i = 0; i + 1 < n; i++) s->g[i] = ng[i + 1];
        s->g[n - 1] = fnext;
        buf[t] = acc;
    }
}

/* arm_biquad_cascade_df2T_f32, one stage, coefficients {b0,b1,b2,a1,a2} */
typedef struct { float c[5], d1, d2; } biquad_t;
static void biquad_run(biquad_t *s, float *buf, int len)
{
    for (int t = 0; t < len; t++) {
        const float x = buf[t];
        const float acc = s->c[0] * x + s->d1;
        s->d1 = (s->c[1] * x + s->c[3] * acc) + s->d2;
        s->d2 = s->c[2] * x + s->c[4] * acc;
        buf[t] = acc;
    }
}

/* arm_lms_norm_f32: 16 taps, mu = 1e-6; hist = the 15 samples before the block */
typedef struct { float w[16], hist[15], energy, x0; } lms_t;
static void lms_run(lms_t *s, const float *src, const float *ref, float *out, int len)
{
    float st[15 + SUB];
    memcpy(st, s->hist, sizeof s->hist);
    float energy = s->energy, x0 = s->x0;
    for (int t = 0; t < len; t++) {
        st[15 + t] = src[t];
        const float *px = st + t; /* oldest of the 16-sample window */
        const float in = src[t];
        energy -= x0 * x0;
        energy += in * in;
        float acc = 0.0f;
        for (int i = 0; i < 16; i++) acc += px[i] * s->w[i];
        out[t] = acc;
        const float e = ref[t] - acc;
        const float w = (e * 0.000001f) / (energy + 0.000000119209289f);
        for (int i = 0; i < 16; i++) s->w[i] += w * px[i];
        x0 = px[0];
    }
    s->energy = energy; s->x0 = x0;
    memcpy(s->hist, st + len, sizeof s->hist);
}

/* ---- per-channel state ------------------------------------------------------ */
typedef struct {
    orc_audio_params_t p;
    float dc_x[6], dc_y[6];                 /* audio_filters.c:348-373 */
    lattice_t lpf_i, lpf_q, hpf_i, hpf_q;   /* audio_filters.c:141-339 */
    biquad_t notch, notch_fft_i, notch_fft_q;
    lms_t lms;                              /* noise_reduction.c */
    float lms_ref[128];
    uint16_t ref_old, ref_new;
    float agc_gain, agc_gain_old, agc_up, agc_down; /* agc.c:9-19 */
    /* FM demodulator statics, audio_processor.c:509-588 */
    float fm_lpf_prev, fm_hpf_a, fm_hpf_b, fm_i_prev, fm_q_prev, fm_sql_avg;
    uint8_t fm_sql_count, squelched;
    float smeter_max, smeter_min;
    /* CW decoder, cw_decoder.c */
    float cw_coeff, cw_limit, cw_limit_low;
    uint8_t cw_real, cw_real_before, cw_filt, cw_filt_before, cw_stop;
    long cw_laststart, cw_starthigh, cw_highdur, cw_startlow, cw_lowdur, cw_highavg;
    uint16_t cw_wpm;
    char cw_code[20], cw_text[16];
    uint32_t blocks; /* RX blocks processed since init: the millisecond clock is 4 * blocks */
    /* FFT, fft.c */
    float fft_mean[FFT_PRINT], fft_max_value;
    uint16_t fft_max_errors;
    uint16_t wtf[50][FFT_PRINT]; /* wtf_buffer, fft.c:28: row 0 is the newest */
    /* zoom FFT (fft.c:44-209,236-261): 4-stage DF1 biquad, 4-tap decimator, sliding buffer */
    float zoom_c[20], zoom_fir[4];
    float zoom_bq_i[16], zoom_bq_q[16];       /* per stage x1,x2,y1,y2 */
    float zoom_dec_i[3], zoom_dec_q[3];       /* the numTaps-1 samples before the block */
    float zoom_buf[2 * FFT_SIZE];             /* FFTInput_ZOOMFFT, interleaved */
    /* TX (audio_processor.c:61-273,590-619) */
    float tx_fir_i[200], tx_fir_q[200];       /* Fir_Tx_Hilbert_State_I/Q: the 200 samples before the block */
    float alc_gain;                           /* ALC_need_gain */
    float txfm_hpf_a, txfm_hpf_b, txfm_index;
    uint32_t txfm_accum;
    float tx_max_amplitude;
} orc_audio_t;

size_t orc_audio_sizeof(void) { return sizeof(orc_audio_t); }

static int lpf_index(int hz)
{
    for (int i = 0; i < DDC_N_LPF; i++) if (ddc_lpf_hz[i] == hz) return i;
    return -1;
}

/* calcBiquad(BIQUAD_notch, Fc, 48000, 0.5, 0), audio_filters.c:375-494, with the
 * firmware's mix of float variables and double literals */
static void notch_coeffs(uint32_t fc, float c[5])
{
    const uint32_t fs = 48000;
    const float q = 0.5f;
    const float k = (float)tan(PI_F * fc / fs);
    const float norm = (float)(1.0 / (1.0 + k / q + k * k));
    const float a0 = (float)((1.0 + k * k) * norm);
    const float a1 = (float)(2.0 * (k * k - 1.0) * norm);
    const float a2 = a0;
    const float b1 = a1;
    const float b2 = (float)((1.0 - k / q + k * k) * norm);
    c[0] = a0; c[1] = a1; c[2] = a2; c[3] = -b1; c[4] = -b2;
}

/* Zoom-FFT coefficient sets (fft.c:76-186; double literals in the firmware, narrowed to
 * float32 when stored): zoom_biquad / zoom_fir_taps in oracle_tables.h */

/* initAudioProcessor + ReinitAudioFilters + InitNotchFilter + InitAGC +
 * InitNoiseReduction + CWDecoder_Init + FFT state, all from power-on zeros */
int orc_audio_configure(orc_audio_t *s, const orc_audio_params_t *p)
{
    memset(s, 0, sizeof *s);
    s->p = *p;
    if (p->filter_width) {
        const int li = lpf_index(p->filter_width);
        if (li < 0) return -1;
        lattice_init(&s->lpf_i, ddc_lpf_stages[li], ddc_lpf_k[li], ddc_lpf_v[li]);
        lattice_init(&s->lpf_q, ddc_lpf_stages[li], ddc_lpf_k[li], ddc_lpf_v[li]);
    }
    int hi = 0; /* 60 Hz unless USB/LSB, audio_filters.c:306-338 */
    if (p->mode == MODE_USB || p->mode == MODE_LSB) {
        hi = -1;
        for (int i = 1; i < DDC_N_HPF; i++) if (ddc_hpf_hz[i] == p->hpf_pass) hi = i;
        if (hi < 0) return -1;
    }
    lattice_init(&s->hpf_i, 6, ddc_hpf_k[hi], ddc_hpf_v[hi]);
    lattice_init(&s->hpf_q, 6, ddc_hpf_k[hi], ddc_hpf_v[hi]);
    notch_coeffs(p->notch_fc, s->notch.c);
    memcpy(s->notch_fft_i.c, s->notch.c, sizeof s->notch.c);
    memcpy(s->notch_fft_q.c, s->notch.c, sizeof s->notch.c);
    s->agc_gain = 0.0f; s->agc_gain_old = 1.0f;
    s->agc_up = 500.0f / (float)p->agc_speed;
    s->agc_down = s->agc_up / 10.0f;
    /* CWDecoder_Init: k = (int)(0.5 + 192*350/48000) = 1 */
    const float omega = (float)((2.0 * PI_F * 1) / 192);
    s->cw_coeff = (float)(2.0 * orc_arm_cos_f32(omega));
    s->cw_limit = 50; s->cw_limit_low = 50;
    memset(s->cw_text, ' ', 15);
    s->cw_text[15] = 0;
    if (p->fft_zoom > 1) { /* FFT_Init, fft.c:188-209 */
        int zi;
        switch (p->fft_zoom) { case 2: zi = 0; break; case 4: zi = 1; break; case 8: zi = 2; break;
                               case 16: zi = 3; break; default: return -1; }
        for (int i = 0; i < 20; i++) s->zoom_c[i] = (float)zoom_biquad[zi][i];
        for (int i = 0; i < 4; i++) s->zoom_fir[i] = (float)zoom_fir_taps[zi][i];
    }
    s->alc_gain = 1.0f;
    s->txfm_index = 2.0f; /* ModulateFM, audio_processor.c:597-606 */
    switch (p->filter_width) {
    case 5000: s->txfm_index = 2.0f; break; case 6000: s->txfm_index = 3.0f; break;
    case 7000: s->txfm_index = 4.0f; break; case 8000: s->txfm_index = 5.0f; break;
    case 9000: s->txfm_index = 6.0f; break; case 9500: s->txfm_index = 7.0f; break;
    case 10000: s->txfm_index = 8.0f; break; case 15000: s->txfm_index = 9.0f; break;
    case 0: s->txfm_index = 10.0f; break; default: break; }
    return 0;
}

static void dc_filter(orc_audio_t *s, float *buf, int n, int slot)
{
    const float A1 = (float)(1.0 - pow(2.0, -11.0));
    for (int i = 0; i < n; i++) {
        const float in = buf[i];
        const float delta = in - s->dc_x[slot];
        const float a1y = A1 * s->dc_y[slot];
        const float out = delta + a1y;
        s->dc_x[slot] = in;
        s->dc_y[slot] = out;
        buf[i] = out;
    }
}

static void do_agc(orc_audio_t *s, float *buf)
{
    float mx = buf[0];
    for (int i = 1; i < BLOCK; i++) if (buf[i] > mx) mx = buf[i]; /* arm_max_f32: signed maximum */
    if (mx == 0.0f) mx = 0.001f;
    const float target = 7000 / mx;
    if (target > s->agc_gain) {
        float step = (target - s->agc_gain) / s->agc_up;
        if (step > 1.0f) step = 1.0f;
        s->agc_gain += step;
    } else
        s->agc_gain -= (s->agc_gain - target) / s->agc_down;
    if (s->agc_gain < 0.0f) s->agc_gain = 0.0f;
    if ((s->agc_gain * mx) > 10000) s->agc_gain = target;
    if (!s->p.agc || s->p.mode == MODE_DIGI_L || s->p.mode == MODE_DIGI_U) s->agc_gain = 1.0f;
    if (s->agc_gain_old != s->agc_gain) {
        float step = 0;
        if (s->agc_gain_old > s->agc_gain) step = -(s->agc_gain_old - s->agc_gain) / BLOCK;
        if (s->agc_gain_old < s->agc_gain) step = (s->agc_gain - s->agc_gain_old) / BLOCK;
        for (int i = 0; i < BLOCK; i++) {
            s->agc_gain_old += step;
            buf[i] = buf[i] * s->agc_gain_old;
        }
    } else
        for (int i = 0; i < BLOCK; i++) buf[i] = buf[i] * s->agc_gain;
}

static void do_dnr(orc_audio_t *s, float *buf)
{
    if (!s->p.dnr) return;
    for (int b = 0; b < BLOCK / SUB; b++) {
        float *x = buf + b * SUB, out[SUB];
        memcpy(&s->lms_ref[s->ref_new], x, SUB * sizeof(float));
        lms_run(&s->lms, x, &s->lms_ref[s->ref_old], out, SUB);
        memcpy(x, out, sizeof out);
        s->ref_old += SUB;
        if (s->ref_old >= 128) s->ref_old = 0;
        s->ref_new = s->ref_old + SUB;
        if (s->ref_new >= 128) s->ref_new = 0;
    }
}

static void demod_fm(orc_audio_t *s, float *bi, float *bq)
{
    float first_angle = 0;
    for (int i = 0; i < BLOCK; i++) {
        const float y = (bq[i] * s->fm_i_prev) - (bi[i] * s->fm_q_prev);
        const float x = (bi[i] * s->fm_i_prev) + (bq[i] * s->fm_q_prev);
        const float angle = atan2f(y, x);
        if (i == 0) first_angle = angle;
        const float a = s->fm_lpf_prev + (0.05f * (angle - s->fm_lpf_prev));
        s->fm_lpf_prev = a;
        s->fm_q_prev = bq[i];
        s->fm_i_prev = bi[i];
        if (!s->squelched || !s->p.fm_sql_threshold) {
            if (s->p.mode == MODE_WFM)
                bi[i] = (float)(angle / PI_F * (1 << 14));
            else {
                const float b = 0.96f * (s->fm_hpf_b + a - s->fm_hpf_a);
                s->fm_hpf_a = a;
                s->fm_hpf_b = b;
                bi[i] = b * 30000.0f;
            }
        } else if (s->squelched)
            bi[i] = 0;
    }
    s->fm_sql_avg = ((1.0f - 0.005f) * s->fm_sql_avg) + (0.005f * sqrtf(fabsf(first_angle)));
    if (s->fm_sql_count == 0) {
        if (s->fm_sql_avg > 0.7f) s->fm_sql_avg = 0.7f;
        const float b = s->fm_sql_avg * 10.0f;
        const int thr = s->p.fm_sql_threshold;
        if (!thr) s->squelched = 0;
        else if (s->squelched) {
            if (b <= (float)((10 - thr) - 0.3f)) s->squelched = 0;
        } else {
            if ((10.0f - thr) > 0.3f) {
                if (b > (float)((10 - thr) + 0.3f)) s->squelched = 1;
            } else if (b > (10.0f - (float)thr)) s->squelched = 1;
        }
        /* the firmware bumps the counter inside this branch only, so the
         * threshold is evaluated on the very first block and never again */
        s->fm_sql_count++;
        if (s->fm_sql_count >= 50) s->fm_sql_count = 0;
    }
}

static void cw_print(orc_audio_t *s, char c)
{
    if (strlen(s->cw_text) >= 15) memmove(s->cw_text, s->cw_text + 1, 15); /* shiftTextLeft */
    const size_t n = strlen(s->cw_text);
    s->cw_text[n] = c;
    s->cw_text[n + 1] = 0;
}

static void cw_decode(orc_audio_t *s)
{
    static const char *tab[][2] = {
        {".-", "A"}, {"-...", "B"}, {"-.-.", "C"}, {"-..", "D"}, {".", "E"}, {"..-.", "F"}, {"--.", "G"},
        {"....", "H"}, {"..", "I"}, {".---", "J"}, {"-.-", "K"}, {".-..", "L"}, {"--", "M"}, {"-.", "N"},
        {"---", "O"}, {".--.", "P"}, {"--.-", "Q"}, {".-.", "R"}, {"...", "S"}, {"-", "T"}, {"..-", "U"},
        {"...-", "V"}, {".--", "W"}, {"-..-", "X"}, {"-.--", "Y"}, {"--..", "Z"}, {".----", "1"},
        {"..---", "2"}, {"...--", "3"}, {"....-", "4"}, {".....", "5"}, {"-....", "6"}, {"--...", "7"},
        {"---..", "8"}, {"----.", "9"}, {"-----", "0"}, {"..--..", "?"}, {".-.-.-", "."}, {"--..--", ","},
        {"-.-.--", "!"}, {".--.-.", "@"}, {"---...", ":"}, {"-....-", "-"}, {"-..-.", "/"}, {"-.--.", "("},
        {"-.--.-", ")"}, {".-...", "_"}, {"...-..-", "$"}, {"...-.-", ">"}, {".-.-.", "<"}, {"...-.", "~"} };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
        if (strcmp(s->cw_code, tab[i][0]) == 0) { cw_print(s, tab[i][1][0]); return; }
}

static void cw_append(orc_audio_t *s, char c)
{
    const size_t n = strlen(s->cw_code);
    if (n < 19) { s->cw_code[n] = c; s->cw_code[n + 1] = 0; }
}

/* CWDecoder_Process (cw_decoder.c:56-174); HAL_GetTick() = now_ms */
static void cw_process(orc_audio_t *s, const float *buf, long now_ms)
{
    float q1 = 0, q2 = 0;
    for (int i = 0; i < BLOCK; i++) {
        const float q0 = s->cw_coeff * q1 - q2 + buf[i];
        q2 = q1;
        q1 = q0;
    }
    const float mag2 = (q1 * q1) + (q2 * q2) - q1 * q2 * s->cw_coeff;
    const float magnitude = (float)sqrt(mag2);
    if (magnitude > s->cw_limit_low) s->cw_limit = (s->cw_limit + ((magnitude - s->cw_limit) / 100));
    s->cw_limit_low = (s->cw_limit_low + ((magnitude - s->cw_limit_low) / 1000));
    if (s->cw_limit_low > magnitude) s->cw_limit_low = magnitude;
    s->cw_real = magnitude > (s->cw_limit_low + (s->cw_limit - s->cw_limit_low) * 0.8f);
    if (s->cw_real != s->cw_real_before) s->cw_laststart = now_ms;
    if ((now_ms - s->cw_laststart) > 6)
        if (s->cw_real != s->cw_filt) s->cw_filt = s->cw_real;
    if (s->cw_filt != s->cw_filt_before) {
        if (s->cw_filt) {
            s->cw_starthigh = now_ms;
            s->cw_lowdur = now_ms - s->cw_startlow;
        } else {
            s->cw_startlow = now_ms;
            s->cw_highdur = now_ms - s->cw_starthigh;
            if (s->cw_highdur < (2 * s->cw_highavg) || s->cw_highavg == 0)
                s->cw_highavg = (s->cw_highdur + s->cw_highavg + s->cw_highavg) / 3;
            if (s->cw_highdur > (5 * s->cw_highavg)) s->cw_highavg = s->cw_highdur + s->cw_highavg;
        }
    }
    if (s->cw_filt != s->cw_filt_before) {
        s->cw_stop = 0;
        if (!s->cw_filt) {
            if (s->cw_highdur < (s->cw_highavg * 2) && s->cw_highdur > (s->cw_highavg * 0.6)) cw_append(s, '.');
            if (s->cw_highdur > (s->cw_highavg * 2) && s->cw_highdur < (s->cw_highavg * 6)) {
                cw_append(s, '-');
                s->cw_wpm = (uint16_t)((s->cw_wpm + (1200 / ((s->cw_highdur) / 3))) / 2);
            }
        } else {
            float lack = 1.0f;
            if (s->cw_wpm > 25) lack = 1.0f;
            if (s->cw_wpm > 30) lack = 1.2f;
            if (s->cw_wpm > 35) lack = 1.5f;
            if (s->cw_lowdur > (s->cw_highavg * (2 * lack)) && s->cw_lowdur < s->cw_highavg * (5 * lack)) {
                cw_decode(s);
                s->cw_code[0] = 0;
            }
            if (s->cw_lowdur >= s->cw_highavg * (5 * lack)) {
                cw_decode(s);
                s->cw_code[0] = 0;
                cw_print(s, ' ');
            }
        }
    }
    if ((now_ms - s->cw_startlow) > (s->cw_highdur * 6) && !s->cw_stop) {
        cw_decode(s);
        s->cw_code[0] = 0;
        s->cw_stop = 1;
    }
    s->cw_real_before = s->cw_real;
    s->cw_filt_before = s->cw_filt;
}

static int32_t f2i(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

/* processRxAudio (audio_processor.c:275-435) on one 192-frame block.
 * spec/voice: interleaved (I,Q) int16 as the FPGA delivers them;
 * out: 192 stereo int32 pairs (left = I, right = Q). */
void orc_audio_rx_block(orc_audio_t *s, const int16_t *spec_iq, const int16_t *voice_iq, int32_t *out)
{
    float bi[BLOCK], bq[BLOCK];
    const int m = s->p.mode;
    const int use_spec = (m == MODE_IQ || m == MODE_NFM || m == MODE_WFM || m == MODE_AM);
    const int16_t *src = use_spec ? spec_iq : voice_iq;
    for (int i = 0; i < BLOCK; i++) { /* fpga.c:305-385: int16 -> float, optional I/Q swap */
        bi[i] = (float)src[2 * i + (s->p.iq_swap ? 1 : 0)];
        bq[i] = (float)src[2 * i + (s->p.iq_swap ? 0 : 1)];
    }
    dc_filter(s, bi, BLOCK, 0);
    dc_filter(s, bq, BLOCK, 1);
    for (int i = 0; i < BLOCK; i++) { bi[i] = bi[i] * s->p.rf_gain; bq[i] = bq[i] * s->p.rf_gain; }
    const int ssb_l = (m == MODE_LSB || m == MODE_CW_L || m == MODE_DIGI_L);
    const int ssb_u = (m == MODE_USB || m == MODE_CW_U || m == MODE_DIGI_U);
    const int hpf = (m == MODE_LSB || m == MODE_CW_L || m == MODE_USB || m == MODE_CW_U);
    int copy = 0;
    if (ssb_l || ssb_u || m == MODE_AM || m == MODE_NFM || m == MODE_WFM) {
        if (hpf)
            for (int b = 0; b < BLOCK / SUB; b++) {
                lattice_run(&s->hpf_i, bi + b * SUB, SUB);
                lattice_run(&s->hpf_q, bq + b * SUB, SUB);
            }
        if (s->p.filter_width > 0)
            for (int b = 0; b < BLOCK / SUB; b++) {
                lattice_run(&s->lpf_i, bi + b * SUB, SUB);
                lattice_run(&s->lpf_q, bq + b * SUB, SUB);
            }
        if (ssb_l) for (int i = 0; i < BLOCK; i++) bi[i] = bi[i] - bq[i];
        else if (ssb_u) for (int i = 0; i < BLOCK; i++) bi[i] = bi[i] + bq[i];
        else if (m == MODE_AM)
            for (int i = 0; i < BLOCK; i++) {
                const float p = bi[i] * bi[i] + bq[i] * bq[i]; /* two arm_mult + arm_add */
                bi[i] = p >= 0.0f ? sqrtf(p) : 0.0f;           /* arm_sqrt_f32 */
            }
        else demod_fm(s, bi, bq);
        if (m != MODE_NFM && m != MODE_WFM && s->p.notch)
            for (int b = 0; b < BLOCK / SUB; b++) biquad_run(&s->notch, bi + b * SUB, SUB);
        copy = 1;
    }
    for (int i = 0; i < BLOCK; i++) { /* doRX_SMETER */
        if (bi[i] > s->smeter_max) s->smeter_max = bi[i];
        if (bq[i] > s->smeter_max) s->smeter_max = bq[i];
        if (bi[i] < s->smeter_min) s->smeter_min = bi[i];
        if (bq[i] < s->smeter_min) s->smeter_min = bq[i];
    }
    if (copy) {
        do_dnr(s, bi);
        do_agc(s, bi);
        if ((ssb_l || ssb_u) && s->p.cw_decoder && (m == MODE_CW_L || m == MODE_CW_U))
            cw_process(s, bi, 4L * (long)s->blocks);
        memcpy(bq, bi, sizeof bi); /* doRX_COPYCHANNEL */
    }
    const float vol = s->p.mute ? 0.0f : (float)s->p.volume / 100.0f;
    for (int i = 0; i < BLOCK; i++) {
        out[2 * i] = f2i(bi[i] * vol);
        out[2 * i + 1] = f2i(bq[i] * vol);
    }
    s->blocks++;
}

void orc_audio_rx_run(orc_audio_t *s, const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames, int32_t *out)
{
    for (size_t b = 0; b + BLOCK <= n_frames; b += BLOCK)
        orc_audio_rx_block(s, spec_iq + 2 * b, voice_iq + 2 * b, out + 2 * b);
}

/* USB-audio frames (audio_processor.c:415-432): (int16_t)((int32_t)v * (1.0f / Volume * 100.0f)),
 * the cast as the Cortex-M4F performs it (vcvt.s32.f32: towards zero, saturating, NaN -> 0; then
 * the low 16 bits). */
void orc_audio_usb_frames(const int32_t *in, size_t n, uint8_t volume, int16_t *out)
{
    const float gain = (1.0f / (float)volume) * 100.0f;
    for (size_t i = 0; i < n; i++) {
        const float f = (float)in[i] * gain;
        int32_t v;
        if (f != f) v = 0;
        else if (f >= 2147483648.0f) v = 2147483647;
        else if (f <= -2147483648.0f) v = (-2147483647 - 1);
        else v = (int32_t)f;
        out[i] = (int16_t)(uint16_t)((uint32_t)v & 0xFFFFu);
    }
}

void orc_audio_status(const orc_audio_t *s, float *smeter_max, float *smeter_min, float *agc_gain,
                      char text[16], uint16_t *wpm, int *squelched)
{
    *smeter_max = s->smeter_max; *smeter_min = s->smeter_min; *agc_gain = s->agc_gain;
    memcpy(text, s->cw_text, 16);
    *wpm = s->cw_wpm; *squelched = s->squelched;
}

/* ---- TX audio ------------------------------------------------------------------ */

/* arm_fir_f32, 201 taps: y[n] = sum_k c[k] * x[n - 200 + k], one accumulator, k ascending */
static void tx_fir(const float *c, float *hist, float *buf, int n)
{
    float ext[200 + SUB];
    memcpy(ext, hist, 200 * sizeof(float));
    memcpy(ext + 200, buf, n * sizeof(float));
    for (int t = 0; t < n; t++) {
        float acc = 0.0f;
        for (int k = 0; k < 201; k++) acc += ext[t + k] * c[k];
        buf[t] = acc;
    }
    memcpy(hist, ext + n, 200 * sizeof(float));
}

static uint32_t f2u_sat(float v)
{
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

/* processTxAudio (audio_processor.c:61-273) on one 192-frame block.  mic_lr: the
 * codec / USB samples (left, right) int16; iq_out: what FPGA_fpgadata_sendiq
 * (fpga.c:403-436) puts on the bus, (I, Q) int16; loop_lr: the codec output of
 * LOOPBACK mode (left, right) int32, zero otherwise.  The HPF/LPF lattices and
 * their state are the RX ones, as in the firmware. */
void orc_audio_tx_block(orc_audio_t *s, const int16_t *mic_lr, int16_t *iq_out, int32_t *loop_lr)
{
    float bi[BLOCK], bq[BLOCK];
    const int m = s->p.mode, tune = s->p.tune;
    float amp = s->p.rf_power / 100.0f * 31000.0f; /* MAX_TX_AMPLITUDE, settings.h:13 */
    for (int i = 0; i < BLOCK; i++) {
        if (tune) { bi[i] = amp; bq[i] = amp; }
        else { bi[i] = (float)mic_lr[2 * i]; bq[i] = (float)mic_lr[2 * i + 1]; }
    }
    if (!tune) { dc_filter(s, bi, BLOCK, 2); dc_filter(s, bq, BLOCK, 3); }
    if (m != MODE_IQ && !tune) {
        for (int b = 0; b < BLOCK / SUB; b++) lattice_run(&s->hpf_i, bi + b * SUB, SUB);
        if (s->p.filter_width > 0)
            for (int b = 0; b < BLOCK / SUB; b++) lattice_run(&s->lpf_i, bi + b * SUB, SUB);
        memcpy(bq, bi, sizeof bi);
        float mx = 0;
        for (int i = 0; i < BLOCK; i++) {
            const float ai = fabsf(bi[i]), aq = fabsf(bq[i]);
            if (ai > mx) mx = ai;
            if (aq > mx) mx = aq;
        }
        if (mx == 0.0f) mx = 0.001f;
        s->tx_max_amplitude = mx;
        const float target = amp / mx;
        if (target > s->alc_gain) s->alc_gain += (target - s->alc_gain) / 500.0f;
        else s->alc_gain -= (s->alc_gain - target) / 500.0f;
        if (target < s->alc_gain) s->alc_gain = target;
        if (s->alc_gain < 0.0f) s->alc_gain = 0.0f;
        if ((s->alc_gain * mx) > (amp * 1.1f)) s->alc_gain = target;
        if (s->alc_gain > 500.0f) s->alc_gain = 500.0f;
        if (mx < 15.0f) s->alc_gain = 0.0f;
        if ((s->alc_gain > 1.0f) && (m == MODE_DIGI_L || m == MODE_DIGI_U || m == MODE_IQ || m == MODE_LOOPBACK))
            s->alc_gain = 1.0f;
        for (int i = 0; i < BLOCK; i++) { bi[i] = bi[i] * s->alc_gain; bq[i] = bq[i] * s->alc_gain; }
        switch (m) {
        case MODE_CW_L: case MODE_CW_U:
            if (!s->p.key) amp = 0;
            for (int i = 0; i < BLOCK; i++) { bq[i] = amp; bi[i] = amp; }
            break;
        case MODE_USB: case MODE_DIGI_U:
            for (int b = 0; b < BLOCK / SUB; b++) {
                tx_fir(tx_hilbert_q, s->tx_fir_q, bi + b * SUB, SUB);
                tx_fir(tx_hilbert_i, s->tx_fir_i, bq + b * SUB, SUB);
            }
            break;
        case MODE_LSB: case MODE_DIGI_L: case MODE_AM:
            for (int b = 0; b < BLOCK / SUB; b++) {
                tx_fir(tx_hilbert_i, s->tx_fir_i, bi + b * SUB, SUB);
                tx_fir(tx_hilbert_q, s->tx_fir_q, bq + b * SUB, SUB);
            }
            if (m == MODE_AM)
                for (int i = 0; i < BLOCK; i++) {
                    const float i_am = ((bi[i] - bq[i]) + amp) / 2.0f;
                    const float q_am = ((bq[i] - bi[i]) - amp) / 2.0f;
                    bi[i] = i_am; bq[i] = q_am;
                }
            break;
        case MODE_NFM: case MODE_WFM:
            for (int i = 0; i < BLOCK; i++) { /* ModulateFM */
                const float a = bi[i];
                s->txfm_hpf_b = 0.95f * (s->txfm_hpf_b + a - s->txfm_hpf_a);
                s->txfm_hpf_a = a;
                s->txfm_accum = f2u_sat((float)s->txfm_accum + s->txfm_hpf_b);
                s->txfm_accum %= 48000u;
                const float sd = (s->txfm_accum / (float)48000) * PI_F * s->txfm_index;
                bi[i] = amp * orc_arm_sin_f32(sd);
                bq[i] = amp * orc_arm_cos_f32(sd);
            }
            break;
        default: break;
        }
    }
    if (s->p.mute && !tune)
        for (int i = 0; i < BLOCK; i++) { bi[i] = bi[i] * 0; bq[i] = bq[i] * 0; }
    if (m == MODE_LOOPBACK && !tune) {
        const float vol = (float)s->p.volume / 50.0f;
        for (int i = 0; i < BLOCK; i++) {
            const int32_t v = f2i(bi[i] * vol);
            loop_lr[2 * i] = v; loop_lr[2 * i + 1] = v;
            iq_out[2 * i] = 0; iq_out[2 * i + 1] = 0;
        }
    } else
        for (int i = 0; i < BLOCK; i++) {
            iq_out[2 * i] = (int16_t)f2i(bi[i]);
            iq_out[2 * i + 1] = (int16_t)f2i(bq[i]);
            loop_lr[2 * i] = 0; loop_lr[2 * i + 1] = 0;
        }
}

void orc_audio_tx_run(orc_audio_t *s, const int16_t *mic_lr, size_t n_frames, int16_t *iq_out, int32_t *loop_lr)
{
    for (size_t b = 0; b + BLOCK <= n_frames; b += BLOCK)
        orc_audio_tx_block(s, mic_lr + 2 * b, iq_out + 2 * b, loop_lr + 2 * b);
}

float orc_audio_alc_gain(const orc_audio_t *s) { return s->alc_gain; }

/* S-meter, stm32f4xx_it.c:398-411 (every 100 ms): peak-to-peak of the audio samples
 * since the last reading -> volts at the ADC -> dBm into 50 ohm, then the extremes
 * are cleared.  log10f_fast is the firmware's own cubic (functions.c:247-260). */
static float log10f_fast(float X)
{
    float Y, F;
    int E;
    F = frexpf(fabsf(X), &E);
    Y = 1.23149591368684f;
    Y *= F;
    Y += -4.11852516267426f;
    Y *= F;
    Y += 6.02197014179219f;
    Y *= F;
    Y += -3.13396450166353f;
    Y += E;
    return (Y * 0.3010299956639812f);
}

float orc_audio_smeter_dbm(orc_audio_t *s)
{
    float vpp = (s->smeter_max / (float)s->p.rf_gain) - (s->smeter_min / (float)s->p.rf_gain);
    for (int i = 0; i < (16 - 12); i++) vpp = vpp / 2;
    const float adc_vpp = vpp * 1.0f / ((float)pow(2.0, 12) - 1);
    const float vrms = adc_vpp * 0.3535f;
    float rf_in = (vrms / 4) * 0.2f;
    if (rf_in < 0.0000001f) rf_in = 0.0000001f;
    const float dbm = 10 * log10f_fast((rf_in * rf_in) / (50.0f * 0.001f));
    s->smeter_max = 0;
    s->smeter_min = 0;
    return dbm;
}

/* ---- FFT panadapter ---------------------------------------------------------- */
static void fft512(double *re, double *im)
{
    const int n = FFT_SIZE;
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * 3.14159265358979323846 / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

/* arm_biquad_cascade_df1_f32, 4 stages, coefficients {b0,b1,b2,a1,a2} per stage,
 * state {x[n-1], x[n-2], y[n-1], y[n-2]} per stage */
static void zoom_biquad_run(const float *c, float *st, float *buf, int n)
{
    for (int sg = 0; sg < 4; sg++) {
        const float b0 = c[5 * sg], b1 = c[5 * sg + 1], b2 = c[5 * sg + 2], a1 = c[5 * sg + 3], a2 = c[5 * sg + 4];
        float x1 = st[4 * sg], x2 = st[4 * sg + 1], y1 = st[4 * sg + 2], y2 = st[4 * sg + 3];
        for (int i = 0; i < n; i++) {
            const float x = buf[i];
            const float acc = (b0 * x) + (b1 * x1) + (b2 * x2) + (a1 * y1) + (a2 * y2);
            x2 = x1; x1 = x; y2 = y1; y1 = acc;
            buf[i] = acc;
        }
        st[4 * sg] = x1; st[4 * sg + 1] = x2; st[4 * sg + 2] = y1; st[4 * sg + 3] = y2;
    }
}

/* arm_fir_decimate_f32, 4 taps, factor M, in place: out[m] = sum_k c[k] * x[m*M - 3 + k]
 * (the state window of output m ends at the first of its M new samples) */
static void zoom_decimate(const float *c, float *hist, float *buf, int n, int M)
{
    float ext[3 + FFT_SIZE];
    memcpy(ext, hist, 3 * sizeof(float));
    memcpy(ext + 3, buf, n * sizeof(float));
    for (int m = 0; m < n / M; m++) {
        const float *x = ext + m * M;
        float acc = 0.0f;
        for (int k = 0; k < 4; k++) acc += x[k] * c[k];
        buf[m] = acc;
    }
    memcpy(hist, ext + n, 3 * sizeof(float));
}

/* getFFTColor (fft.c:506-539) + rgb888torgb565 */
static uint16_t fft_color(uint8_t height)
{
    uint8_t red = 0, green = 0, blue = 0;
    const float c1 = 0.1f, c2 = 0.4f, c3 = 0.5f;
    if (height < 30 * c1) blue = (uint8_t)(height * 255 / (30 * c1));
    else if (height < 30 * (c1 + c2)) {
        green = (uint8_t)((height - 30 * c1) * 255 / ((30 - 30 * c1) * (c1 + c2)));
        red = green;
        blue = (uint8_t)(255 - green);
    } else {
        red = 255;
        blue = 0;
        green = (uint8_t)(255 - (height - (30 * (c1 + c2))) * 255 / ((30 - (30 * (c1 + c2))) * (c1 + c2 + c3)));
    }
    return (uint16_t)(((red & 0xF8) << 8) | ((green & 0xFC) << 3) | (blue >> 3));
}

/* FFT_doFFT (fft.c:212-331, zoom = 1) followed by the row mapping of FFT_printFFT
 * (fft.c:361-379).  in: 512 SPEC (I,Q) int16 pairs; bins: the 256 smoothed,
 * normalised magnitudes (FFTOutput_mean); row: 256 RGB565 waterfall pixels,
 * already fft-shifted; heights: the 256 column heights 0..30. */
void orc_fft_frame(orc_audio_t *s, const int16_t *spec_iq, float *bins, uint16_t *row, uint8_t *heights)
{
    float fi[FFT_SIZE], fq[FFT_SIZE];
    for (int i = 0; i < FFT_SIZE; i++) {
        fi[i] = (float)spec_iq[2 * i + (s->p.iq_swap ? 1 : 0)];
        fq[i] = (float)spec_iq[2 * i + (s->p.iq_swap ? 0 : 1)];
    }
    dc_filter(s, fi, FFT_SIZE, 4);
    dc_filter(s, fq, FFT_SIZE, 5);
    if (s->p.notch) { biquad_run(&s->notch_fft_i, fi, FFT_SIZE); biquad_run(&s->notch_fft_q, fq, FFT_SIZE); }
    if (s->p.fft_zoom > 1) { /* fft.c:236-261 */
        const int M = s->p.fft_zoom, zw = FFT_SIZE / M;
        zoom_biquad_run(s->zoom_c, s->zoom_bq_i, fi, FFT_SIZE);
        zoom_biquad_run(s->zoom_c, s->zoom_bq_q, fq, FFT_SIZE);
        zoom_decimate(s->zoom_fir, s->zoom_dec_i, fi, FFT_SIZE, M);
        zoom_decimate(s->zoom_fir, s->zoom_dec_q, fq, FFT_SIZE, M);
        memmove(s->zoom_buf, s->zoom_buf + 2 * zw, sizeof(float) * 2 * (FFT_SIZE - zw));
        for (int i = 0; i < zw; i++) {
            s->zoom_buf[2 * (FFT_SIZE - zw + i)] = fi[i];
            s->zoom_buf[2 * (FFT_SIZE - zw + i) + 1] = fq[i];
        }
        for (int i = 0; i < FFT_SIZE; i++) { fi[i] = s->zoom_buf[2 * i]; fq[i] = s->zoom_buf[2 * i + 1]; }
    }
    double re[FFT_SIZE], im[FFT_SIZE];
    for (int i = 0; i < FFT_SIZE; i++) {
        const float w = 0.54f - 0.46f * orc_arm_cos_f32((2.0f * PI_F * i) / ((float)FFT_SIZE - 1.0f));
        re[i] = (double)(w * fi[i]);
        im[i] = (double)(w * fq[i]);
    }
    fft512(re, im);
    float mag[FFT_SIZE];
    for (int i = 0; i < FFT_SIZE; i++) mag[i] = (float)sqrt(re[i] * re[i] + im[i] * im[i]);
    float v[FFT_PRINT];
    for (int i = 0; i < FFT_PRINT; i++) v[i] = (mag[2 * i] + mag[2 * i + 1]) / 2;
    v[0] = v[1];
    float mx = v[0];
    for (int i = 1; i < FFT_PRINT; i++) if (v[i] > mx) mx = v[i];
    const float diff = (mx - s->fft_max_value) / 10.0f;
    const uint16_t e = s->fft_max_errors;
    if (e >= 6 && diff > 0) s->fft_max_value += diff;
    else if (e <= 1 && diff < 0 && diff < -10.0f) s->fft_max_value += diff;
    else if (e <= 1 && s->fft_max_value > 10.0f) s->fft_max_value -= 10.0f;
    else if (e <= 1 && diff < 0 && diff < -1.0f) s->fft_max_value += diff;
    else if (e <= 1 && s->fft_max_value > 1.0f) s->fft_max_value -= 1.0f;
    s->fft_max_errors = 0;
    if (s->fft_max_value < 100.0f) s->fft_max_value = 100.0f;
    if (s->p.mode == MODE_LOOPBACK) s->fft_max_value = 60000;
    const float inv = 1.0f / s->fft_max_value;
    for (int i = 0; i < FFT_PRINT; i++) {
        const float x = v[i] * inv;
        if (s->fft_mean[i] < x) s->fft_mean[i] += (x - s->fft_mean[i]) / s->p.fft_averaging;
        else s->fft_mean[i] -= (s->fft_mean[i] - x) / s->p.fft_averaging;
        bins[i] = s->fft_mean[i];
    }
    for (int x = 0; x < FFT_PRINT; x++) {
        const int nx = x < FFT_PRINT / 2 ? x + FFT_PRINT / 2 : x - FFT_PRINT / 2;
        uint16_t h = (uint16_t)f2i(s->fft_mean[x] * 30);
        uint16_t c;
        if (h > 29) { h = 30; c = 0xF800; s->fft_max_errors++; }
        else c = fft_color((uint8_t)h);
        row[nx] = c;
        heights[nx] = (uint8_t)h;
    }
    /* FFT_printFFT scrolls the waterfall down by one row and puts the new one on top (fft.c:353-376) */
    memmove(s->wtf[1], s->wtf[0], sizeof(uint16_t) * 49 * FFT_PRINT);
    memcpy(s->wtf[0], row, sizeof(uint16_t) * FFT_PRINT);
}

void orc_fft_get_waterfall(const orc_audio_t *s, uint16_t *out) { memcpy(out, s->wtf, sizeof s->wtf); }

/* FFT_moveWaterfall (fft.c:453-503): on a retune by freq_diff Hz the picture moves
 * sideways by (freq_diff / 187) * zoom pixels; the smoothed bins are moved in place
 * in the same loop, in their own (unshifted) bin order, exactly as written. */
void orc_fft_move_waterfall(orc_audio_t *s, int16_t freq_diff)
{
    const int zoom = s->p.fft_zoom > 1 ? s->p.fft_zoom : 1;
    freq_diff = (int16_t)((freq_diff / (48000 / FFT_PRINT)) * zoom);
    for (int y = 0; y < 50; y++) {
        if (freq_diff > 0) {
            for (int x = 0; x < FFT_PRINT; x++) {
                const int new_x = x + freq_diff;
                if (y == 0) {
                    int fx = new_x;
                    if (fx >= FFT_PRINT) fx -= FFT_PRINT;
                    if (fx >= 0 && fx < FFT_PRINT) s->fft_mean[x] = s->fft_mean[fx];
                }
                if (new_x < 0 || new_x >= FFT_PRINT) { s->wtf[y][x] = 0; continue; }
                s->wtf[y][x] = s->wtf[y][new_x];
            }
        } else if (freq_diff < 0) {
            for (int x = FFT_PRINT - 1; x >= 0; x--) {
                const int new_x = x + freq_diff;
                if (y == 0) {
                    int fx = new_x;
                    if (fx < 0) fx += FFT_PRINT;
                    if (fx >= 0 && fx < FFT_PRINT) s->fft_mean[x] = s->fft_mean[fx];
                }
                if (new_x < 0) { s->wtf[y][x] = 0; continue; }
                s->wtf[y][x] = s->wtf[y][new_x];
            }
        }
    }
}

/* ---- auto-gain search ------------------------------------------------------------
 * TRX_DoAutoGain, STM32/Src/trx_manager.c:268-356 (stage / wait counters :35-36),
 * constants STM32/Src/settings.h:23-25 and trx_manager.h:9, db2rateV functions.c:262-265.
 * st: {stage, wait_reaction, preamp, att}; written as the firmware writes it, case by case. */
void orc_autogain_step(uint8_t *st, int16_t adc_max)
{
    uint8_t stage = st[0], wait = st[1];
    const double att_rate = pow(10.0, (12 / 20.0)), pre_rate = pow(10.0, (20 / 20.0));
    switch (stage) {
    case 0:
        st[2] = 0; st[3] = 1;
        stage++; wait = 0;
        break;
    case 1:
        if ((adc_max * att_rate) <= 1100) wait++;
        else wait = 0;
        if (wait >= 7) { stage++; wait = 0; }
        break;
    case 2:
        st[2] = 0; st[3] = 0;
        stage++; wait = 0;
        break;
    case 3:
        if (adc_max > 1100) stage -= 3;
        if ((adc_max * pre_rate / att_rate) <= 1100) wait++;
        else wait = 0;
        if (wait >= 7) { stage++; wait = 0; }
        break;
    case 4:
        st[2] = 1; st[3] = 1;
        stage++; wait = 0;
        break;
    case 5:
        if (adc_max > 1100) stage -= 3;
        if ((adc_max * att_rate) <= 1100) wait++;
        else wait = 0;
        if (wait >= 7) { stage++; wait = 0; }
        break;
    case 6:
        st[2] = 1; st[3] = 0;
        stage++; wait = 0;
        break;
    case 7:
        if (adc_max > 1100) stage -= 3;
        break;
    default:
        stage = 0;
        break;
    }
    st[0] = stage; st[1] = wait;
}
