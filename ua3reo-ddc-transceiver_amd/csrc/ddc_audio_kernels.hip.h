/*
 * 48 kHz RX audio path and FFT panadapter of the MI355X DDC.
 *
 * Replaces the STM32-side processing of the reference (float32, CMSIS-DSP):
 *   dc_filter                      STM32/Src/audio_filters.c:358-373
 *   doRX_HPF / doRX_LPF            STM32/Src/audio_processor.c:445-464 (arm_iir_lattice_f32,
 *                                  tables audio_filters.c:45-122, selection :141-339)
 *   SSB / AM / FM demodulation     STM32/Src/audio_processor.c:308-362, :509-588
 *   doRX_NOTCH + calcBiquad        STM32/Src/audio_processor.c:466-473, audio_filters.c:375-494
 *   processNoiseReduction          STM32/Src/noise_reduction.c:19-37 (arm_lms_norm_f32)
 *   DoAGC                          STM32/Src/agc.c:21-67
 *   CWDecoder_Process              STM32/Src/cw_decoder.c:56-174 (millisecond clock = 4 ms per block)
 *   output stage                   STM32/Src/audio_processor.c:364-394
 *   FFT_doFFT + row mapping        STM32/Src/fft.c:212-331, :361-379, :506-539 (zoom branch :236-261)
 *   processTxAudio / ModulateFM    STM32/Src/audio_processor.c:61-273, :590-619
 *
 * All of it is recursion in time (IIR lattices, LMS, AGC, squelch).  The RX path is cut
 * where the recursions allow it: the lattice filters run as a systolic pipeline over the
 * lanes of a wave (ddc_audio_rx_filters), the rest as one wave per channel with the
 * element-wise work spread over the lanes (ddc_audio_rx_demod); the TX path and the
 * panadapter prefilter walk one channel per lane (pair).  In every form a sample sees the
 * firmware's operations in the firmware's order; floating-point contraction is off so
 * that every operation rounds where the firmware's does.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/ua3reo_ddc.h"
#include "ddc_audio_coeffs.h"

namespace ddc {

#pragma clang fp contract(off)

constexpr int kABlock = 192;
constexpr int kASub = 64;
constexpr float kPiF = 3.14159265358979f;

/* arm_sin_f32 / arm_cos_f32 of CMSIS-DSP 1.6.0 (the library the firmware links; its source is
 * not part of the reference tree, the algorithm is the published one): the angle scaled to turns,
 * its fraction times 512 split into a table index and a remainder, linear interpolation between
 * two entries of a 513-entry sine table.  The table is built by the host (ddc_lib.hip,
 * arm_sin_table) and uploaded when a ctx is created.  T is that table. */
template <typename T>
__host__ __device__ __forceinline__ float arm_sin_interp(const T &table, float in)
{
    int32_t n = (int32_t)in;
    if (in < 0.0f) n--;
    in = in - (float)n;
    float findex = 512.0f * in;
    uint32_t index = (uint32_t)(uint16_t)findex;
    if (index >= 512u) { index = 0; findex -= 512.0f; }
    const float fract = findex - (float)index;
    const float a = table[index], b = table[index + 1];
    return (1.0f - fract) * a + fract * b;
}
template <typename T> __host__ __device__ __forceinline__ float arm_sin_with(const T &table, float x)
{
    return arm_sin_interp(table, x * 0.159154943092f);
}
template <typename T> __host__ __device__ __forceinline__ float arm_cos_with(const T &table, float x)
{
    return arm_sin_interp(table, x * 0.159154943092f + 0.25f);
}
__device__ float g_arm_sin_table[513]; // written once per device, before any kernel that reads it

struct Lattice {
    float g[12];
};

struct AudioState {
    ddc_audio_params p;
    int32_t lpf_idx, hpf_idx, lpf_n, pad;
    float dc_x[6], dc_y[6];
    Lattice lpf_i, lpf_q, hpf_i, hpf_q;
    float notch_c[5];
    float notch_d[2], notch_fi[2], notch_fq[2];
    float lms_w[16], lms_hist[15], lms_energy, lms_x0;
    float lms_ref[128];
    uint32_t ref_old, ref_new;
    float agc_gain, agc_gain_old, agc_up, agc_down;
    float fm_lpf_prev, fm_hpf_a, fm_hpf_b, fm_i_prev, fm_q_prev, fm_sql_avg;
    uint32_t fm_sql_count, squelched;
    float smeter_max, smeter_min;
    float cw_coeff, cw_limit, cw_limit_low;
    uint32_t cw_real, cw_real_before, cw_filt, cw_filt_before, cw_stop;
    int64_t cw_laststart, cw_starthigh, cw_highdur, cw_startlow, cw_lowdur, cw_highavg;
    uint32_t cw_wpm;
    char cw_code[20], cw_text[16];
    uint32_t blocks;
    float fft_mean[256], fft_max_value;
    uint32_t fft_max_errors;
};

/* Zoom-FFT state of one channel (fft.c:44-209): 4-stage DF1 biquad low-pass,
 * 4-tap FIR decimator, and the tail of the decimated stream that the next
 * frame's sliding 512-sample window starts with. */
struct ZoomState {
    float c[20], fir[4];
    float bq_i[16], bq_q[16];
    float dec_i[3], dec_q[3];
    uint32_t zoom, pad;
    float2 hist[512];
};

/* TX-only state of one channel: Hilbert FIR delay lines, ALC gain, FM modulator. */
struct TxAudioState {
    float fir_i[200], fir_q[200];
    float alc_gain, fm_hpf_a, fm_hpf_b, fm_index, max_amplitude;
    uint32_t fm_accum;
};

__device__ __forceinline__ float dc_step(AudioState &s, int slot, float in)
{
    const float A1 = 0.99951171875f; // 1 - 2^-11
    const float delta = in - s.dc_x[slot];
    const float a1y = A1 * s.dc_y[slot];
    const float out = delta + a1y;
    s.dc_x[slot] = in;
    s.dc_y[slot] = out;
    return out;
}

__device__ __forceinline__ float biquad_step(const float *c, float *d, float x)
{
    const float acc = c[0] * x + d[0];
    d[0] = (c[1] * x + c[3] * acc) + d[1];
    d[1] = c[2] * x + c[4] * acc;
    return acc;
}

__device__ inline void lms_block(AudioState &s, float *x /*64, in place*/)
{
    float st[15 + kASub];
    for (int i = 0; i < 15; i++) st[i] = s.lms_hist[i];
    const float *ref = &s.lms_ref[s.ref_old];
    float energy = s.lms_energy, x0 = s.lms_x0;
    for (int t = 0; t < kASub; t++) {
        const float in = x[t];
        st[15 + t] = in;
        const float *px = st + t;
        energy -= x0 * x0;
        energy += in * in;
        float acc = 0.0f;
        for (int i = 0; i < 16; i++) acc += px[i] * s.lms_w[i];
        const float e = ref[t] - acc;
        const float w = (e * 0.000001f) / (energy + 0.000000119209289f);
        for (int i = 0; i < 16; i++) s.lms_w[i] += w * px[i];
        x0 = px[0];
        x[t] = acc;
    }
    s.lms_energy = energy;
    s.lms_x0 = x0;
    for (int i = 0; i < 15; i++) s.lms_hist[i] = st[kASub + i];
}

__device__ inline void agc_block(AudioState &s, float *buf)
{
    float mx = buf[0];
    for (int i = 1; i < kABlock; i++) if (buf[i] > mx) mx = buf[i];
    if (mx == 0.0f) mx = 0.001f;
    const float target = 7000 / mx;
    if (target > s.agc_gain) {
        float step = (target - s.agc_gain) / s.agc_up;
        if (step > 1.0f) step = 1.0f;
        s.agc_gain += step;
    } else
        s.agc_gain -= (s.agc_gain - target) / s.agc_down;
    if (s.agc_gain < 0.0f) s.agc_gain = 0.0f;
    if ((s.agc_gain * mx) > 10000) s.agc_gain = target;
    const int m = s.p.mode;
    if (!s.p.agc || m == DDC_MODE_DIGI_L || m == DDC_MODE_DIGI_U) s.agc_gain = 1.0f;
    if (s.agc_gain_old != s.agc_gain) {
        float step = 0;
        if (s.agc_gain_old > s.agc_gain) step = -(s.agc_gain_old - s.agc_gain) / kABlock;
        if (s.agc_gain_old < s.agc_gain) step = (s.agc_gain - s.agc_gain_old) / kABlock;
        for (int i = 0; i < kABlock; i++) {
            s.agc_gain_old += step;
            buf[i] = buf[i] * s.agc_gain_old;
        }
    } else
        for (int i = 0; i < kABlock; i++) buf[i] = buf[i] * s.agc_gain;
}

__device__ inline int dev_strlen(const char *s) { int n = 0; while (s[n]) n++; return n; }

__device__ inline void cw_print(AudioState &s, char c)
{
    int n = dev_strlen(s.cw_text);
    if (n >= 15) {
        for (int i = 0; i < 15; i++) s.cw_text[i] = s.cw_text[i + 1];
        n = dev_strlen(s.cw_text);
    }
    s.cw_text[n] = c;
    s.cw_text[n + 1] = 0;
}

// Morse table as (length, bit pattern: 1 = dash, first symbol in the highest used bit)
__device__ inline void cw_decode(AudioState &s)
{
    const int n = dev_strlen(s.cw_code);
    if (n == 0 || n > 7) return;
    unsigned pat = 0;
    for (int i = 0; i < n; i++) pat = (pat << 1) | (s.cw_code[i] == '-' ? 1u : 0u);
    const unsigned key = ((unsigned)n << 8) | pat;
    char c = 0;
    switch (key) {
    case 0x201: c = 'A'; break; case 0x408: c = 'B'; break; case 0x40A: c = 'C'; break; case 0x304: c = 'D'; break;
    case 0x100: c = 'E'; break; case 0x402: c = 'F'; break; case 0x306: c = 'G'; break; case 0x400: c = 'H'; break;
    case 0x200: c = 'I'; break; case 0x407: c = 'J'; break; case 0x305: c = 'K'; break; case 0x404: c = 'L'; break;
    case 0x203: c = 'M'; break; case 0x202: c = 'N'; break; case 0x307: c = 'O'; break; case 0x406: c = 'P'; break;
    case 0x40D: c = 'Q'; break; case 0x302: c = 'R'; break; case 0x300: c = 'S'; break; case 0x101: c = 'T'; break;
    case 0x301: c = 'U'; break; case 0x401: c = 'V'; break; case 0x303: c = 'W'; break; case 0x409: c = 'X'; break;
    case 0x40B: c = 'Y'; break; case 0x40C: c = 'Z'; break;
    case 0x50F: c = '1'; break; case 0x507: c = '2'; break; case 0x503: c = '3'; break; case 0x501: c = '4'; break;
    case 0x500: c = '5'; break; case 0x510: c = '6'; break; case 0x518: c = '7'; break; case 0x51C: c = '8'; break;
    case 0x51E: c = '9'; break; case 0x51F: c = '0'; break;
    case 0x60C: c = '?'; break; case 0x615: c = '.'; break; case 0x633: c = ','; break; case 0x62B: c = '!'; break;
    case 0x61A: c = '@'; break; case 0x638: c = ':'; break; case 0x621: c = '-'; break; case 0x512: c = '/'; break;
    case 0x516: c = '('; break; case 0x62D: c = ')'; break; case 0x508: c = '_'; break; case 0x709: c = '$'; break;
    case 0x605: c = '>'; break; case 0x50A: c = '<'; break; case 0x502: c = '~'; break;
    default: break;
    }
    if (c) cw_print(s, c);
}

__device__ inline void cw_append(AudioState &s, char c)
{
    const int n = dev_strlen(s.cw_code);
    if (n < 19) { s.cw_code[n] = c; s.cw_code[n + 1] = 0; }
}

__device__ inline void cw_block(AudioState &s, const float *buf, int64_t now_ms)
{
    float q1 = 0, q2 = 0;
    for (int i = 0; i < kABlock; i++) {
        const float q0 = s.cw_coeff * q1 - q2 + buf[i];
        q2 = q1;
        q1 = q0;
    }
    const float mag2 = (q1 * q1) + (q2 * q2) - q1 * q2 * s.cw_coeff;
    const float magnitude = (float)sqrt((double)mag2);
    if (magnitude > s.cw_limit_low) s.cw_limit = (s.cw_limit + ((magnitude - s.cw_limit) / 100));
    s.cw_limit_low = (s.cw_limit_low + ((magnitude - s.cw_limit_low) / 1000));
    if (s.cw_limit_low > magnitude) s.cw_limit_low = magnitude;
    s.cw_real = magnitude > (s.cw_limit_low + (s.cw_limit - s.cw_limit_low) * 0.8f);
    if (s.cw_real != s.cw_real_before) s.cw_laststart = now_ms;
    if ((now_ms - s.cw_laststart) > 6)
        if (s.cw_real != s.cw_filt) s.cw_filt = s.cw_real;
    if (s.cw_filt != s.cw_filt_before) {
        if (s.cw_filt) {
            s.cw_starthigh = now_ms;
            s.cw_lowdur = now_ms - s.cw_startlow;
        } else {
            s.cw_startlow = now_ms;
            s.cw_highdur = now_ms - s.cw_starthigh;
            if (s.cw_highdur < (2 * s.cw_highavg) || s.cw_highavg == 0)
                s.cw_highavg = (s.cw_highdur + s.cw_highavg + s.cw_highavg) / 3;
            if (s.cw_highdur > (5 * s.cw_highavg)) s.cw_highavg = s.cw_highdur + s.cw_highavg;
        }
        s.cw_stop = 0;
        if (!s.cw_filt) {
            if ((double)s.cw_highdur < (double)(s.cw_highavg * 2) && (double)s.cw_highdur > (s.cw_highavg * 0.6)) cw_append(s, '.');
            if (s.cw_highdur > (s.cw_highavg * 2) && s.cw_highdur < (s.cw_highavg * 6)) {
                cw_append(s, '-');
                s.cw_wpm = (uint32_t)(uint16_t)((s.cw_wpm + (1200 / ((s.cw_highdur) / 3))) / 2);
            }
        } else {
            float lack = 1.0f;
            if (s.cw_wpm > 30) lack = 1.2f;
            if (s.cw_wpm > 35) lack = 1.5f;
            if ((float)s.cw_lowdur > ((float)s.cw_highavg * (2 * lack)) && (float)s.cw_lowdur < (float)s.cw_highavg * (5 * lack)) {
                cw_decode(s);
                s.cw_code[0] = 0;
            }
            if ((float)s.cw_lowdur >= (float)s.cw_highavg * (5 * lack)) {
                cw_decode(s);
                s.cw_code[0] = 0;
                cw_print(s, ' ');
            }
        }
    }
    if ((now_ms - s.cw_startlow) > (s.cw_highdur * 6) && !s.cw_stop) {
        cw_decode(s);
        s.cw_code[0] = 0;
        s.cw_stop = 1;
    }
    s.cw_real_before = s.cw_real;
    s.cw_filt_before = s.cw_filt;
}

__device__ __forceinline__ int32_t f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)v;
}

// arm_iir_lattice_f32 step with the state and the coefficient tables in registers
// (N is 6, 7 or 11; the arrays are sized for the longest filter): the TX path's form
template <int N>
__device__ __forceinline__ float lattice_reg(float (&g)[11], const float (&k)[11], const float (&v)[12], float x)
{
    float fcurr = x, fnext = x, acc = 0.0f, ng[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float gcurr = g[i];
        fnext = fcurr - k[i] * gcurr;
        const float gnext = fnext * k[i] + gcurr;
        acc += gnext * v[i];
        ng[i] = gnext;
        fcurr = fnext;
    }
    acc += fnext * v[N];
#pragma unroll
    for (int i = 0; i + 1 < N; i++) g[i] = ng[i + 1];
    g[N - 1] = fnext;
    return acc;
}

constexpr int kABiPitch = kABlock + 1; // LDS row pitch in floats: rows of neighbouring lanes start one bank apart

/*
 * The filter half of processRxAudio as a systolic pipeline: dc_filter, RF gain and the
 * HPF / LPF lattices (audio_processor.c:299-362, :445-464) for every channel, results to
 * filt[channel][frame] = {I, Q} floats for ddc_audio_rx_demod.
 *
 * An IIR lattice is a chain of stages: stage i of sample n needs f from stage i - 1 of the
 * same sample and the g that stage i + 1 produced for sample n - 1.  So every stage gets a
 * lane of its own, f and the running tap sum acc travel down the lanes (wave_shr:1), g travels
 * up (wave_shl:1), and lane L works on item j = tick - L.  Items alternate between the two
 * branches (j = 2 n + branch), which fills the tick a stage would otherwise wait for its g:
 * every lane does useful work on every tick, and every sample sees exactly the operations of
 * arm_iir_lattice_f32 in exactly their order.
 *
 * 20 lanes per channel, three channels per wave:
 *   lane 0       source: int16 -> float, dc_filter (slots 0 / 1), RF gain
 *   lanes 1..6   HPF stages 0..5, lane 7 its output tap (acc + f v[6]) -> the next filter's input
 *   lanes 8..18  LPF stages (a 7-stage filter sits in lanes 12..18), lane 19 its output tap
 * A filter that is off, and the unused lanes of a short one, pass f through.  Roles are data,
 * not code: a pass lane is a stage with k = v = 0, an output tap a stage with k = 1 and no g
 * from its neighbour, so one instruction stream serves all lanes.  The loop is unrolled over
 * 32 ticks (16 frames): the source lane's input registers and lane 19's output registers are
 * indexed statically.
 */
constexpr int kFiltSeg = 20, kFiltPerWave = 3, kFiltTicks = 32;

__device__ __forceinline__ float dpp_from_lower(float v)   // lane i <- lane i - 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_upper(float v)   // lane i <- lane i + 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

struct FiltLane {
    float k, v;            // this lane's stage coefficients
    bool use_nb;           // g comes from the lane above (a real stage)
    bool is_fin, is_src, zero_acc;
    float rf_gain;
    uint32_t half_i, half_q; // bit offset of the I / Q half of a packed input word
    uint32_t L;
};

// MODE 0: first iteration (lanes start one after the other), 1: steady state, 2: the last two
// iterations (lanes stop one after the other; the g of the last I item is kept beside that of the last Q item)
template <int MODE>
__device__ __forceinline__ void filt_iteration(const FiltLane &ln, const uint4 (&in4)[4], uint32_t tick0, uint32_t items,
                                               float &f_out, float &a_out, float &g_out, float &g_prev, float g_q_init,
                                               float (&dx)[2], float (&dy)[2], float (&o)[32], float2 *dst, uint32_t frame0,
                                               bool store_lane, bool do_store)
{
    const uint32_t in[16] = { in4[0].x, in4[0].y, in4[0].z, in4[0].w, in4[1].x, in4[1].y, in4[1].z, in4[1].w,
                              in4[2].x, in4[2].y, in4[2].z, in4[2].w, in4[3].x, in4[3].y, in4[3].z, in4[3].w };
    const float A1 = 0.99951171875f; // 1 - 2^-11
#pragma unroll
    for (int k = 0; k < kFiltTicks; k++) {
        const float f_in = dpp_from_lower(f_out), a_in = dpp_from_lower(a_out), g_in = dpp_from_upper(g_out);
        // one lattice stage (arm_iir_lattice_f32): fnext = f - k g; gnext = fnext k + g; acc += gnext v
        const float g_use = ln.use_nb ? g_in : 0.0f;
        const float kg = ln.k * g_use;
        const float t = f_in - kg;
        const float tk = t * ln.k;
        const float gn = tk + g_use;
        const float av = gn * ln.v;
        const float as = a_in + av;
        float nf = ln.is_fin ? as : t;
        const float na = ln.zero_acc ? 0.0f : as;
        // the source lane: item k of this iteration is frame k / 2, branch k & 1
        const int br = k & 1;
        const float x = (float)(int)__builtin_amdgcn_sbfe(in[k >> 1], br ? ln.half_q : ln.half_i, 16u);
        const float d = x - dx[br];
        const float a1y = A1 * dy[br];
        const float y = d + a1y;
        const float sv = y * ln.rf_gain;
        nf = ln.is_src ? sv : nf;
        bool act = true;
        if (MODE == 0) act = (uint32_t)k >= ln.L;
        if (MODE == 2) act = tick0 + (uint32_t)k - ln.L < items;
        if (MODE == 1) {
            dx[br] = x; dy[br] = y;
            f_out = nf; a_out = na; g_out = gn;
        } else {
            if (act) {
                dx[br] = x; dy[br] = y;
                if (MODE == 2) g_prev = g_out;
                f_out = nf; a_out = na; g_out = gn;
            }
            if (MODE == 0 && (uint32_t)(k + 1) == ln.L) g_out = g_q_init; // the lane above starts its Q item next
        }
        o[(k + 13) & 31] = f_out;
        if (k == 18 && do_store && store_lane) {
            // lane 19 has just finished the 32 items (16 frames) that started 32 ticks ago
#pragma unroll
            for (int j = 0; j < 8; j++)
                reinterpret_cast<float4 *>(dst + frame0)[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
        }
    }
}

__global__ __launch_bounds__(64) void ddc_audio_rx_filters(AudioState *__restrict__ states, uint32_t n_ch,
                                                           const int16_t *__restrict__ spec_iq,
                                                           const int16_t *__restrict__ voice_iq, uint32_t n_frames,
                                                           float2 *__restrict__ filt)
{
    const uint32_t lane = threadIdx.x, seg = lane / kFiltSeg, L = lane - seg * kFiltSeg;
    const uint32_t ch_raw = blockIdx.x * kFiltPerWave + seg;
    const bool live = seg < (uint32_t)kFiltPerWave && ch_raw < n_ch;
    const uint32_t ch = live ? ch_raw : 0;
    AudioState &s = states[ch];
    const ddc_audio_params p = s.p;
    const int m = p.mode;
    const bool use_spec = (m == DDC_MODE_IQ || m == DDC_MODE_NFM || m == DDC_MODE_WFM || m == DDC_MODE_AM);
    const bool ssb = (m == DDC_MODE_LSB || m == DDC_MODE_CW_L || m == DDC_MODE_DIGI_L || m == DDC_MODE_USB ||
                      m == DDC_MODE_CW_U || m == DDC_MODE_DIGI_U);
    const bool hpf = (m == DDC_MODE_LSB || m == DDC_MODE_CW_L || m == DDC_MODE_USB || m == DDC_MODE_CW_U);
    const bool chain = ssb || m == DDC_MODE_NFM || m == DDC_MODE_WFM || m == DDC_MODE_AM;
    const bool lpf = chain && p.filter_width > 0;
    const int lpf_n = s.lpf_n;

    FiltLane ln;
    ln.L = L;
    ln.k = 0.0f; ln.v = 0.0f;
    ln.use_nb = false; ln.is_fin = false;
    ln.is_src = L == 0;
    ln.rf_gain = p.rf_gain;
    ln.half_i = p.iq_swap ? 16u : 0u;
    ln.half_q = p.iq_swap ? 0u : 16u;
    // which state slot this lane's g output feeds (the stage above it), -1: none
    int provides = -1;
    const float *g_i = nullptr, *g_q = nullptr;
    if (live && hpf && L >= 1 && L <= 7) {
        const int i = (int)L - 1; // 0..5: stage, 6: output tap
        if (i < 6) { ln.k = ddc_hpf_k[s.hpf_idx][i]; ln.v = ddc_hpf_v[s.hpf_idx][i]; ln.use_nb = true; provides = i - 1; }
        else { ln.k = 1.0f; ln.v = ddc_hpf_v[s.hpf_idx][6]; ln.is_fin = true; provides = 5; }
        g_i = s.hpf_i.g; g_q = s.hpf_q.g;
    }
    if (live && lpf && L >= 8) {
        const int i = (int)L - (19 - lpf_n); // 0..lpf_n-1: stage, lpf_n: output tap
        if (i >= 0 && i < lpf_n) { ln.k = ddc_lpf_k[s.lpf_idx][i]; ln.v = ddc_lpf_v[s.lpf_idx][i]; ln.use_nb = true; provides = i - 1; }
        else if (i == lpf_n) { ln.k = 1.0f; ln.v = ddc_lpf_v[s.lpf_idx][lpf_n]; ln.is_fin = true; provides = lpf_n - 1; }
        g_i = s.lpf_i.g; g_q = s.lpf_q.g;
    }
    ln.zero_acc = ln.is_fin || ln.is_src;
    float dx[2] = { s.dc_x[0], s.dc_x[1] }, dy[2] = { s.dc_y[0], s.dc_y[1] };
    float g_out = provides >= 0 ? g_i[provides] : 0.0f;
    const float g_q_init = provides >= 0 ? g_q[provides] : 0.0f;
    float f_out = 0.0f, a_out = 0.0f, g_prev = 0.0f;
    float o[32];
#pragma unroll
    for (int i = 0; i < 32; i++) o[i] = 0.0f;

    const uint32_t *src = reinterpret_cast<const uint32_t *>(use_spec ? spec_iq : voice_iq) + (size_t)ch * n_frames;
    float2 *dst = filt + (size_t)ch * n_frames;
    const bool src_lane = live && L == 0, store_lane = live && L == (uint32_t)kFiltSeg - 1;
    const uint32_t G = n_frames / 16, items = 2 * n_frames; // n_frames is a multiple of 192
    uint4 cur[4] = {}, nxt[4] = {};
    if (src_lane) {
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = reinterpret_cast<const uint4 *>(src)[k];
    }
    for (uint32_t c = 0; c <= G; c++) {
        if (src_lane && c + 1 < G) {
#pragma unroll
            for (int k = 0; k < 4; k++) nxt[k] = reinterpret_cast<const uint4 *>(src + 16 * (c + 1))[k];
        }
        const uint32_t frame0 = c ? 16 * (c - 1) : 0;
        if (c == 0)
            filt_iteration<0>(ln, cur, 0, items, f_out, a_out, g_out, g_prev, g_q_init, dx, dy, o, dst, frame0, store_lane, false);
        else if (c + 1 < G)
            filt_iteration<1>(ln, cur, 32 * c, items, f_out, a_out, g_out, g_prev, g_q_init, dx, dy, o, dst, frame0, store_lane, true);
        else
            filt_iteration<2>(ln, cur, 32 * c, items, f_out, a_out, g_out, g_prev, g_q_init, dx, dy, o, dst, frame0, store_lane, true);
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    if (!live) return;
    if (L == 0) { s.dc_x[0] = dx[0]; s.dc_x[1] = dx[1]; s.dc_y[0] = dy[0]; s.dc_y[1] = dy[1]; }
    if (provides >= 0) {
        float *wi = L <= 7 ? s.hpf_i.g : s.lpf_i.g, *wq = L <= 7 ? s.hpf_q.g : s.lpf_q.g;
        wi[provides] = g_prev;
        wq[provides] = g_out;
    }
}

/*
 * arm_lms_norm_f32 (noise_reduction.c:19-37 through lms_block above) for one 64-sample
 * sub-block with the sixteen taps in lanes 0..15: lane i holds w[i] and the sample px[i]
 * the tap multiplies; the tap sum runs up the lanes one add at a time (lane i takes the sum
 * of lane i - 1 and adds its own product, sixteen times, so it is the same left-to-right sum
 * as the loop's), everything scalar in the firmware's loop is computed by every lane alike.
 * x_reg / ref_reg: lane l holds input sample l / reference sample l of the sub-block;
 * returns the filter output of sample l in lane l.  px of lanes 0..14 carries the fifteen
 * past samples from sub-block to sub-block.
 */
__device__ __forceinline__ float lms_lanes(float x_reg, float ref_reg, float &w, float &px, float &energy, float &x0,
                                           uint32_t lane)
{
    float out = 0.0f;
    for (int t = 0; t < kASub; t++) {
        const float in = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x_reg), t));
        if (lane == 15) px = in;
        energy -= x0 * x0;
        energy += in * in;
        const float prod = px * w;
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) // row_shr:1, lane 0 keeps 0: a(lane i) = a(lane i - 1) + prod(lane i)
            a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xF, 0xF, false)) + prod;
        const float acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 15));
        const float ref = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ref_reg), t));
        const float e = ref - acc;
        const float ws = (e * 0.000001f) / (energy + 0.000000119209289f);
        const float wp = ws * px;
        w += wp;
        x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), 0));
        if (lane == (uint32_t)t) out = acc;
        px = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, px), 0x101, 0xF, 0xF, false)); // lane i <- lane i + 1
    }
    return out;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}

/*
 * The rest of processRxAudio (audio_processor.c:308-435) on the filtered samples of
 * ddc_audio_rx_filters, one wave per channel, block after block of 192 frames (three frames
 * per lane).  Everything that is the same operation on every sample runs across the lanes
 * (SSB sum / AM envelope / FM discriminator angle, S-meter and AGC peak searches, gain
 * multiply, volume and integer conversion); the recursions (FM filters, notch biquad, AGC
 * ramp) run as one loop over the block that every lane executes on the same values, each
 * lane keeping the three results that are its own, so every sample still sees the firmware's
 * operations in the firmware's order.  Noise reduction and the CW decoder keep their
 * one-lane form.  filt: [channel][n_frames] {I, Q}; audio: [channel][n_frames][2] int32.
 */
__global__ __launch_bounds__(64) void ddc_audio_rx_demod(AudioState *__restrict__ states, uint32_t n_ch,
                                                         const float2 *__restrict__ filt, uint32_t n_frames,
                                                         int32_t *__restrict__ audio)
{
    __shared__ float bi[kABlock], bq[kABlock];
    const uint32_t ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= n_ch) return;
    AudioState &s = states[ch];
    const ddc_audio_params p = s.p;
    const int m = p.mode;
    const float2 *src = filt + (size_t)ch * n_frames;
    int2 *dst = reinterpret_cast<int2 *>(audio) + (size_t)ch * n_frames;
    const bool ssb_l = (m == DDC_MODE_LSB || m == DDC_MODE_CW_L || m == DDC_MODE_DIGI_L);
    const bool ssb_u = (m == DDC_MODE_USB || m == DDC_MODE_CW_U || m == DDC_MODE_DIGI_U);
    const bool fm = (m == DDC_MODE_NFM || m == DDC_MODE_WFM);
    const bool am = m == DDC_MODE_AM;
    const bool chain = ssb_l || ssb_u || fm || am;
    const bool notch = chain && !fm && p.notch;
    const float vol = p.mute ? 0.0f : (float)p.volume / 100.0f;
    // state that every lane carries along identically
    float nc[5], nd[2] = { s.notch_d[0], s.notch_d[1] };
#pragma unroll
    for (int i = 0; i < 5; i++) nc[i] = s.notch_c[i];
    float sm_max = s.smeter_max, sm_min = s.smeter_min;
    float fm_lpf = s.fm_lpf_prev, fm_ha = s.fm_hpf_a, fm_hb = s.fm_hpf_b, fm_ip = s.fm_i_prev, fm_qp = s.fm_q_prev;
    float fm_avg = s.fm_sql_avg;
    uint32_t fm_cnt = s.fm_sql_count, squelched = s.squelched, blocks = s.blocks;
    float agc_gain = s.agc_gain, agc_gain_old = s.agc_gain_old;
    const float agc_up = s.agc_up, agc_down = s.agc_down;
    const bool agc_on = p.agc && m != DDC_MODE_DIGI_L && m != DDC_MODE_DIGI_U;
    // noise reduction: taps and past samples in lanes 0..15, the two halves of the reference ring in all lanes
    const bool dnr = chain && p.dnr;
    float lms_w = 0, lms_px = 0, lms_energy = 0, lms_x0 = 0, ref_a = 0, ref_b = 0;
    uint32_t ref_old = 0, ref_new = 0;
    if (dnr) {
        lms_w = lane < 16 ? s.lms_w[lane] : 0.0f;
        lms_px = lane < 15 ? s.lms_hist[lane] : 0.0f;
        lms_energy = s.lms_energy; lms_x0 = s.lms_x0;
        ref_a = s.lms_ref[lane]; ref_b = s.lms_ref[64 + lane];
        ref_old = s.ref_old; ref_new = s.ref_new;
    }

    for (uint32_t b0 = 0; b0 + kABlock <= n_frames; b0 += kABlock) {
        float fi[3], fq[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float2 v = src[b0 + lane + 64 * j];
            fi[j] = v.x;
            fq[j] = v.y;
        }
        if (chain) {
            if (ssb_l) {
#pragma unroll
                for (int j = 0; j < 3; j++) fi[j] = fi[j] - fq[j];
            } else if (ssb_u) {
#pragma unroll
                for (int j = 0; j < 3; j++) fi[j] = fi[j] + fq[j];
            } else if (am) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float pw = fi[j] * fi[j] + fq[j] * fq[j];
                    fi[j] = pw >= 0.0f ? sqrtf(pw) : 0.0f;
                }
            } else { // DemodulateFM, audio_processor.c:509-559: the angle against the sample before, then the recursions
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 3; j++) { bi[lane + 64 * j] = fi[j]; bq[lane + 64 * j] = fq[j]; }
                __syncthreads();
                float ang[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const uint32_t e = lane + 64 * j;
                    const float ip = e ? bi[e - 1] : fm_ip, qp = e ? bq[e - 1] : fm_qp;
                    const float y = (fq[j] * ip) - (fi[j] * qp);
                    const float x = (fi[j] * ip) + (fq[j] * qp);
                    ang[j] = atan2f(y, x);
                }
                fm_ip = bi[kABlock - 1];
                fm_qp = bq[kABlock - 1];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 3; j++) bi[lane + 64 * j] = ang[j];
                __syncthreads();
                const bool open = !squelched || !p.fm_sql_threshold;
#pragma unroll
                for (int j = 0; j < 3; j++)
                for (int ii = 0; ii < 64; ii++) {
                    const float angle = bi[64 * j + ii];
                    const float a = fm_lpf + (0.05f * (angle - fm_lpf));
                    fm_lpf = a;
                    float o = 0;
                    if (open) {
                        if (m == DDC_MODE_WFM)
                            o = angle / kPiF * 16384.0f;
                        else {
                            const float bb = 0.96f * (fm_hb + a - fm_ha);
                            fm_ha = a;
                            fm_hb = bb;
                            o = bb * 30000.0f;
                        }
                    }
                    if ((uint32_t)ii == lane) fi[j] = o;
                }
                const float first_angle = bi[0];
                // squelch, audio_processor.c:560-588
                fm_avg = ((1.0f - 0.005f) * fm_avg) + (0.005f * sqrtf(fabsf(first_angle)));
                if (fm_cnt == 0) {
                    if (fm_avg > 0.7f) fm_avg = 0.7f;
                    const float bsq = fm_avg * 10.0f;
                    const int thr = p.fm_sql_threshold;
                    if (!thr) squelched = 0;
                    else if (squelched) {
                        if (bsq <= (float)((10 - thr) - 0.3f)) squelched = 0;
                    } else {
                        if ((10.0f - thr) > 0.3f) {
                            if (bsq > (float)((10 - thr) + 0.3f)) squelched = 1;
                        } else if (bsq > (10.0f - (float)thr)) squelched = 1;
                    }
                    fm_cnt++; // only ever bumped here, as in the firmware
                    if (fm_cnt >= 50) fm_cnt = 0;
                }
            }
            if (notch) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 3; j++) bi[lane + 64 * j] = fi[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 3; j++)
                for (int ii = 0; ii < 64; ii++) {
                    const float o = biquad_step(nc, nd, bi[64 * j + ii]);
                    if ((uint32_t)ii == lane) fi[j] = o;
                }
            }
        }
        {   // S-meter extremes over both outputs of every frame
            float hi = fmaxf(fmaxf(fi[0], fi[1]), fi[2]), lo = fminf(fminf(fi[0], fi[1]), fi[2]);
            hi = fmaxf(hi, fmaxf(fmaxf(fq[0], fq[1]), fq[2]));
            lo = fminf(lo, fminf(fminf(fq[0], fq[1]), fq[2]));
            hi = wave_max(hi);
            lo = wave_min(lo);
            if (hi > sm_max) sm_max = hi;
            if (lo < sm_min) sm_min = lo;
        }
        if (!chain) {
#pragma unroll
            for (int j = 0; j < 3; j++) dst[b0 + lane + 64 * j] = make_int2(f2i_sat(fi[j] * vol), f2i_sat(fq[j] * vol));
            blocks++;
            continue;
        }
        if (dnr) {
#pragma unroll
            for (int j = 0; j < 3; j++) { // processNoiseReduction on each third of the block: the reference is the sub-block before
                if (ref_new) ref_b = fi[j]; else ref_a = fi[j];
                fi[j] = lms_lanes(fi[j], ref_old ? ref_b : ref_a, lms_w, lms_px, lms_energy, lms_x0, lane);
                ref_old += kASub;
                if (ref_old >= 128) ref_old = 0;
                ref_new = ref_old + kASub;
                if (ref_new >= 128) ref_new = 0;
            }
        }
        {   // DoAGC, agc.c:21-67
            float mx = wave_max(fmaxf(fmaxf(fi[0], fi[1]), fi[2]));
            if (mx == 0.0f) mx = 0.001f;
            const float target = 7000 / mx;
            if (target > agc_gain) {
                float step = (target - agc_gain) / agc_up;
                if (step > 1.0f) step = 1.0f;
                agc_gain += step;
            } else
                agc_gain -= (agc_gain - target) / agc_down;
            if (agc_gain < 0.0f) agc_gain = 0.0f;
            if ((agc_gain * mx) > 10000) agc_gain = target;
            if (!agc_on) agc_gain = 1.0f;
            if (agc_gain_old != agc_gain) {
                float step = 0;
                if (agc_gain_old > agc_gain) step = -(agc_gain_old - agc_gain) / kABlock;
                if (agc_gain_old < agc_gain) step = (agc_gain - agc_gain_old) / kABlock;
                float g[3] = {0, 0, 0};
#pragma unroll
                for (int j = 0; j < 3; j++)
                for (int ii = 0; ii < 64; ii++) { // the running gain is a sum of 192 steps, one after the other
                    agc_gain_old += step;
                    if ((uint32_t)ii == lane) g[j] = agc_gain_old;
                }
#pragma unroll
                for (int j = 0; j < 3; j++) fi[j] = fi[j] * g[j];
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) fi[j] = fi[j] * agc_gain;
            }
        }
        if (p.cw_decoder && (m == DDC_MODE_CW_L || m == DDC_MODE_CW_U)) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; j++) bi[lane + 64 * j] = fi[j];
            __syncthreads();
            if (lane == 0) cw_block(s, bi, 4ll * (int64_t)blocks);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) { // doRX_COPYCHANNEL + volume: left = right
            const int32_t a = f2i_sat(fi[j] * vol);
            dst[b0 + lane + 64 * j] = make_int2(a, a);
        }
        blocks++;
    }
    if (dnr) {
        if (lane < 16) s.lms_w[lane] = lms_w;
        if (lane < 15) s.lms_hist[lane] = lms_px;
        s.lms_ref[lane] = ref_a;
        s.lms_ref[64 + lane] = ref_b;
        if (lane == 0) { s.lms_energy = lms_energy; s.lms_x0 = lms_x0; s.ref_old = ref_old; s.ref_new = ref_new; }
    }
    if (lane) return;
    s.notch_d[0] = nd[0]; s.notch_d[1] = nd[1];
    s.smeter_max = sm_max; s.smeter_min = sm_min;
    s.fm_lpf_prev = fm_lpf; s.fm_hpf_a = fm_ha; s.fm_hpf_b = fm_hb; s.fm_i_prev = fm_ip; s.fm_q_prev = fm_qp;
    s.fm_sql_avg = fm_avg; s.fm_sql_count = fm_cnt; s.squelched = squelched; s.blocks = blocks;
    s.agc_gain = agc_gain; s.agc_gain_old = agc_gain_old;
}

/* ---- FFT panadapter ---------------------------------------------------------- */

/* Pass A with zoom (lane = one branch, I or Q, of one channel): DC filter and optional
 * notch over the call's frames of 512 SPEC samples, in sequence; the two branches never
 * meet before the FFT, so they run in neighbouring lanes.  The work buffer is
 * planar: work[channel][branch][work_pitch] floats.  The stream is low-passed
 * (4 biquads), decimated by M (4-tap FIR) and appended behind the 512 - 512/M samples
 * kept from earlier frames: [512 - 512/M + n/M], where FFT frame f is the 512 samples
 * starting at f * 512/M (the firmware's sliding FFTInput_ZOOMFFT buffer).  Without zoom
 * the samples go to [n]: ddc_fft_prefilter_tiled below. */
template <uint32_t M>
__global__ __launch_bounds__(64) void ddc_fft_prefilter(AudioState *__restrict__ states, ZoomState *__restrict__ zooms,
                                                        uint32_t n_ch, const int16_t *__restrict__ spec_iq, uint32_t n,
                                                        float *__restrict__ work, uint32_t work_pitch)
{
    static_assert(M > 1, "zoom factors only");
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ch = id >> 1, q = id & 1u; // q = 0: I branch, 1: Q branch
    if (ch >= n_ch) return;
    AudioState &s = states[ch];
    ZoomState &z = zooms[ch];
    float dx = s.dc_x[4 + q], dy = s.dc_y[4 + q];
    float *nsave = q ? s.notch_fq : s.notch_fi;
    float nd[2] = { nsave[0], nsave[1] };
    float c[5];
    for (int i = 0; i < 5; i++) c[i] = s.notch_c[i];
    const bool notch = s.p.notch;
    // the half of a packed SPEC word this lane filters (I is the low half)
    const uint32_t half_off = ((q ^ (s.p.iq_swap ? 1u : 0u)) & 1u) * 16u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(spec_iq) + (size_t)ch * n;
    float *dst = work + ((size_t)ch * 2 + q) * work_pitch;
    const float A1 = 0.99951171875f;
    float zc[20], zf[4], bs[16], w[4]; // w: the newest four low-passed samples
    constexpr uint32_t keep = 512 - 512 / M;
    {
        float *bq = q ? z.bq_q : z.bq_i, *dec = q ? z.dec_q : z.dec_i;
        for (int i = 0; i < 20; i++) zc[i] = z.c[i];
        for (int i = 0; i < 4; i++) zf[i] = z.fir[i];
        for (int i = 0; i < 16; i++) bs[i] = bq[i];
        for (int i = 0; i < 3; i++) w[i + 1] = dec[i];
        for (uint32_t i = 0; i < keep; i++) dst[i] = q ? z.hist[i].y : z.hist[i].x;
    }
    // 16 samples (64 bytes) per trip.  A lane pair walks one channel, so a trip's
    // loads touch memory nobody else has warmed; four trips are kept in flight in a
    // register ring so that their latency hides behind the recursion of the three
    // trips before (n is a multiple of 512, rows are 16-byte aligned).
    constexpr int kRing = 4;
    uint4 ring[kRing][4];
#pragma unroll
    for (int r = 0; r < kRing; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) ring[r][k] = reinterpret_cast<const uint4 *>(src + 16 * r)[k];
    // the notch is a per-channel setting: when no lane of the wave has it on (the usual case) the
    // sample loop runs without the biquad and without a branch around it
    const bool any_notch = __any(notch);
    auto trip = [&](const uint4 (&in4)[4], const uint32_t i0, auto with_notch) {
        const uint32_t in[16] = { in4[0].x, in4[0].y, in4[0].z, in4[0].w, in4[1].x, in4[1].y, in4[1].z, in4[1].w,
                                  in4[2].x, in4[2].y, in4[2].z, in4[2].w, in4[3].x, in4[3].y, in4[3].z, in4[3].w };
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float f = (float)(int)__builtin_amdgcn_sbfe(in[k], half_off, 16u); // the builtin is typed unsigned
            { const float d = f - dx; const float a = A1 * dy; const float o = d + a; dx = f; dy = o; f = o; }
            if (decltype(with_notch)::value) { if (notch) f = biquad_step(c, nd, f); }
#pragma unroll
            for (int sg = 0; sg < 4; sg++) { // arm_biquad_cascade_df1_f32: b0 x + b1 x1 + b2 x2 + a1 y1 + a2 y2
                const float b0 = zc[5 * sg], b1 = zc[5 * sg + 1], b2 = zc[5 * sg + 2], a1 = zc[5 * sg + 3], a2 = zc[5 * sg + 4];
                const float y = (b0 * f) + (b1 * bs[4 * sg]) + (b2 * bs[4 * sg + 1]) + (a1 * bs[4 * sg + 2]) + (a2 * bs[4 * sg + 3]);
                bs[4 * sg + 1] = bs[4 * sg]; bs[4 * sg] = f; bs[4 * sg + 3] = bs[4 * sg + 2]; bs[4 * sg + 2] = y;
                f = y;
            }
            w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = f;
            if (k % M == 0) { // arm_fir_decimate_f32: the window of output m ends at sample m * M (i0 is a multiple of 16)
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc += w[kk] * zf[kk];
                dst[keep + (i0 + k) / M] = acc;
            }
        }
    };
    for (uint32_t i0 = 0; i0 < n; i0 += 16 * kRing) {
#pragma unroll
        for (int r = 0; r < kRing; r++) {
            uint4 cur[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = ring[r][k];
            const uint32_t ahead = i0 + 16 * (r + kRing);
            if (ahead < n) {
#pragma unroll
                for (int k = 0; k < 4; k++) ring[r][k] = reinterpret_cast<const uint4 *>(src + ahead)[k];
            }
            if (any_notch) trip(cur, i0 + 16 * r, std::true_type{});
            else trip(cur, i0 + 16 * r, std::false_type{});
        }
    }
    s.dc_x[4 + q] = dx; s.dc_y[4 + q] = dy;
    if (notch) { nsave[0] = nd[0]; nsave[1] = nd[1]; }
    {
        float *bq = q ? z.bq_q : z.bq_i, *dec = q ? z.dec_q : z.dec_i;
        for (int i = 0; i < 16; i++) bq[i] = bs[i];
        for (int i = 0; i < 3; i++) dec[i] = w[i + 1];
        const uint32_t produced = n / M;
        for (uint32_t i = 0; i < keep; i++) {
            if (q) z.hist[i].y = dst[produced + i]; else z.hist[i].x = dst[produced + i];
        }
    }
}

/* Pass A without zoom: the same DC filter (and optional notch) as ddc_fft_prefilter<1>, laid
 * out so that the recursion is all its lanes do.  A workgroup of four waves takes 16 channels
 * (few per workgroup: the work is one dependent chain per channel and branch, so it wants many
 * compute units, not full ones), tile after tile of 64 samples:
 *   waves 1 and 2 (loaders) fetch the packed SPEC words, whole 128-byte lines and several tiles
 *     ahead, and turn them into the filter's first difference x[n] - x[n-1] per branch (both
 *     are 16-bit integers, so the difference is exact in integers as it is in floats), written
 *     to an LDS tile as floats;
 *   wave 0 (lanes 0..15 the I branches, 16..31 the Q branches of the 16 channels) runs
 *     y = d + A y (and the notch biquad where it is on) over its row of the tile, in place;
 *   wave 3 writes the finished tile of the step before to the work buffer, whole lines again.
 * Four tiles rotate through these roles (filled two steps ahead, so that the recursion wave can
 * fetch its next row while it works on the current one); rows are 68 floats apart, which
 * spreads the lanes' 16-byte accesses over all banks. */
constexpr int kPfCh = 16, kPfTile = 64, kPfPitch = 68, kPfAhead = 4, kPfThreads = 4 * 64;

__global__ __launch_bounds__(kPfThreads) void ddc_fft_prefilter_tiled(AudioState *__restrict__ states, uint32_t n_ch,
                                                                      const int16_t *__restrict__ spec_iq, uint32_t n,
                                                                      float *__restrict__ work, uint32_t work_pitch)
{
    constexpr int kP4 = kPfPitch / 4;  // row pitch in 16-byte units
    constexpr int kPieces = kPfTile / 4; // 16-byte pieces per row of a tile: 16, one DPP row of lanes
    static_assert(kPieces == 16 && kPfCh == 16, "lane maps below");
    __shared__ float4 tile[4][2][kPfCh * kP4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t ch0 = blockIdx.x * kPfCh;
    const uint32_t n_tiles = n / kPfTile; // n is a multiple of 512, so n_tiles is one of 8
    const uint32_t mrow = lane >> 4, mpiece = lane & 15u; // loaders and the writer: 16 bytes of a row each

    if (wave == 1 || wave == 2) {
        // ---- loaders: 8 rows each, two 16-byte fetches per lane and tile
        const uint4 *src[2];
        uint32_t row[2], swap[2], last_w[2] = { 0, 0 }, final_w[2] = { 0, 0 };
#pragma unroll
        for (int i = 0; i < 2; i++) {
            row[i] = (wave - 1u) * 8u + (uint32_t)i * 4u + mrow;
            const uint32_t chan = ch0 + row[i] < n_ch ? ch0 + row[i] : n_ch - 1u; // rows past the end repeat the last channel
            src[i] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(spec_iq) + (size_t)chan * n) + mpiece;
            swap[i] = states[chan].p.iq_swap ? 16u : 0u;
        }
        uint4 ring[kPfAhead][2]; // slot u % kPfAhead holds tile u; tiles past the end repeat the last one
        auto fetch = [&](uint32_t t, uint4 (&v)[2]) {
            const uint32_t tt = t < n_tiles ? t : n_tiles - 1u;
#pragma unroll
            for (int i = 0; i < 2; i++) v[i] = src[i][(size_t)tt * kPieces];
        };
        auto emit = [&](const uint4 (&v)[2], uint32_t t, uint32_t buf) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                // I in the low half whatever the channel's I/Q swap
                const uint32_t w[4] = { __builtin_amdgcn_alignbit(v[i].x, v[i].x, swap[i]), __builtin_amdgcn_alignbit(v[i].y, v[i].y, swap[i]),
                                        __builtin_amdgcn_alignbit(v[i].z, v[i].z, swap[i]), __builtin_amdgcn_alignbit(v[i].w, v[i].w, swap[i]) };
                // the word before this lane's four: the neighbour's last one, or for the first piece of a row
                // the last word of the row's tile before (held by the lane fifteen up)
                const uint32_t from_nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x111, 0xF, 0xF, false);       // row_shr:1
                const uint32_t from_tile = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last_w[i], 0x10F, 0xF, 0xF, false); // row_shl:15
                const uint32_t before = mpiece == 0u ? from_tile : from_nb;
                last_w[i] = w[3];
                if (t == n_tiles - 1u) final_w[i] = w[3];
                const uint32_t p[5] = { before, w[0], w[1], w[2], w[3] };
                float di[4], dq[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    di[e] = (float)((int)(int16_t)(p[e + 1] & 0xFFFFu) - (int)(int16_t)(p[e] & 0xFFFFu));
                    dq[e] = (float)(((int)p[e + 1] >> 16) - ((int)p[e] >> 16));
                }
                tile[buf][0][row[i] * kP4 + mpiece] = make_float4(di[0], di[1], di[2], di[3]);
                tile[buf][1][row[i] * kP4 + mpiece] = make_float4(dq[0], dq[1], dq[2], dq[3]);
            }
        };
        fetch(0, ring[0]);
        fetch(1, ring[1]);
        emit(ring[0], 0, 0);
        emit(ring[1], 1, 1);
#pragma unroll
        for (int u = 2; u < 2 + kPfAhead; u++) fetch((uint32_t)u, ring[u % kPfAhead]);
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += kPfAhead) {
#pragma unroll
            for (int u = 0; u < kPfAhead; u++) {
                const uint32_t t = t0 + (uint32_t)u;
                const int slot = (u + 2) % kPfAhead; // tile t + 2 (a constant once the loop is unrolled)
                emit(ring[slot], t + 2u, (t + 2u) & 3u);
                fetch(t + 2u + kPfAhead, ring[slot]);
                __syncthreads();
            }
        }
        // the new x[n-1] of both branches: the last word of a row sits with the lane of its last piece
        if (mpiece == kPieces - 1) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (ch0 + row[i] >= n_ch) continue;
                AudioState &s = states[ch0 + row[i]];
                s.dc_x[4] = (float)(int)(int16_t)(final_w[i] & 0xFFFFu);
                s.dc_x[5] = (float)((int)final_w[i] >> 16);
            }
        }
        return;
    }

    if (wave == 3) {
        // ---- writer: the tile finished in the step before, 2 x 16 rows x 256 bytes
        __syncthreads();
        for (uint32_t t = 0; t <= n_tiles; t++) {
            const uint32_t buf = (t - 1u) & 3u;
            if (t > 0) {
                float4 v[8]; // all LDS reads first, so that their latency is paid once
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t q = (uint32_t)i >> 2, r = ((uint32_t)i & 3u) * 4u + mrow;
                    v[i] = tile[buf][q][r * kP4 + mpiece];
                }
                // (the values are pinned to registers here: otherwise each read sinks into its store's
                // conditional block and the LDS latencies line up one behind the other)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t q = (uint32_t)i >> 2, r = ((uint32_t)i & 3u) * 4u + mrow;
                    if (ch0 + r < n_ch)
                        reinterpret_cast<float4 *>(work + ((size_t)(ch0 + r) * 2 + q) * work_pitch + (size_t)(t - 1u) * kPfTile)[mpiece] = v[i];
                }
            }
            if (t < n_tiles) __syncthreads();
        }
        return;
    }

    // ---- the recursion: lanes 0..15 = I branches, 16..31 = Q branches of the 16 channels
    const uint32_t q = (lane >> 4) & 1u, crow = lane & 15u;
    const uint32_t ch = ch0 + crow;
    const bool live = lane < 32u && ch < n_ch;
    AudioState &s = states[ch < n_ch ? ch : n_ch - 1u];
    const float A1 = 0.99951171875f;
    const float dx = s.dc_x[4 + q];
    float dy = s.dc_y[4 + q];
    float *nsave = q ? s.notch_fq : s.notch_fi;
    float nd[2] = { nsave[0], nsave[1] };
    float c[5];
    for (int i = 0; i < 5; i++) c[i] = s.notch_c[i];
    const bool notch = live && s.p.notch;
    const bool any_notch = __any(notch);
    __syncthreads();
    // the loaders took the sample before the call as 0: put the filter's own x[n-1] in its place
    const bool work_lane = lane < 32u;
    float4 rows[2][kPieces]; // the row being filtered and the next one, fetched while the first is worked on
    if (work_lane) {
        float4 &first = tile[0][q][crow * kP4];
        first.x = first.x - dx;
#pragma unroll
        for (int k4 = 0; k4 < kPieces; k4++) rows[0][k4] = tile[0][q][crow * kP4 + k4];
    }
    auto step = [&](float4 (&cur)[kPieces], float4 (&nxt)[kPieces], uint32_t t, auto with_notch) {
        if (work_lane) {
            float4 *rowp = &tile[t & 3u][q][crow * kP4];
            const float4 *nextp = &tile[(t + 1u) & 3u][q][crow * kP4]; // filled in the step before
#pragma unroll
            for (int k4 = 0; k4 < kPieces; k4++) nxt[k4] = nextp[k4];
#pragma unroll
            for (int k4 = 0; k4 < kPieces; k4++) {
                const float d[4] = { cur[k4].x, cur[k4].y, cur[k4].z, cur[k4].w };
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = A1 * dy;
                    dy = d[e] + a;
                    o[e] = dy;
                    if (decltype(with_notch)::value) { if (notch) o[e] = biquad_step(c, nd, o[e]); }
                }
                rowp[k4] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
    };
    for (uint32_t t = 0; t < n_tiles; t += 2) { // n_tiles is even
        if (any_notch) {
            step(rows[0], rows[1], t, std::true_type{});
            step(rows[1], rows[0], t + 1u, std::true_type{});
        } else {
            step(rows[0], rows[1], t, std::false_type{});
            step(rows[1], rows[0], t + 1u, std::false_type{});
        }
    }
    if (live) {
        s.dc_y[4 + q] = dy;
        if (notch) { nsave[0] = nd[0]; nsave[1] = nd[1]; }
    }
}

__constant__ float c_fft_window[512]; // 0.54 - 0.46 cos(2 pi i / 511)
__constant__ float2 c_fft_twiddle[256]; // exp(-2 pi i k / 512)

/* Pass B (workgroup of 64 = one frame of one channel): Hamming window, 512-point
 * radix-2 FFT in LDS, magnitude, 2:1 bin average, bin 0 := bin 1.
 * mags: [channel][frames][256]. */
__global__ __launch_bounds__(64) void ddc_fft_frames(const float *__restrict__ work, uint32_t frames_per_ch,
                                                     uint32_t work_pitch, uint32_t frame_step,
                                                     float *__restrict__ mags)
{
    __shared__ float re[512], im[512];
    const uint32_t fr = blockIdx.x, t = threadIdx.x;
    const uint32_t ch = fr / frames_per_ch, f = fr - ch * frames_per_ch;
    const float *xr = work + (size_t)ch * 2 * work_pitch + (size_t)f * frame_step, *xi = xr + work_pitch;
    for (int j = 0; j < 8; j++) {
        const int i = t + 64 * j;
        const int r = (int)(__brev((unsigned)i) >> 23); // 9-bit reversal
        const float w = c_fft_window[i];
        re[r] = w * xr[i];
        im[r] = w * xi[i];
    }
    __syncthreads();
    for (int len = 2, lg = 1; len <= 512; len <<= 1, lg++) {
        const int half = len >> 1;
        for (int j = 0; j < 4; j++) {
            const int bf = t + 64 * j;           // butterfly 0..255
            const int k = bf & (half - 1);
            const int a = ((bf >> (lg - 1)) << lg) + k, b = a + half;
            const float2 w = c_fft_twiddle[k << (9 - lg)];
            const float xr = re[b] * w.x - im[b] * w.y, xi = re[b] * w.y + im[b] * w.x;
            const float ar = re[a], ai = im[a];
            re[b] = ar - xr; im[b] = ai - xi;
            re[a] = ar + xr; im[a] = ai + xi;
        }
        __syncthreads();
    }
    float *out = mags + (size_t)fr * 256;
    for (int j = 0; j < 4; j++) {
        const int i = t + 64 * j;
        const float m0 = sqrtf(re[2 * i] * re[2 * i] + im[2 * i] * im[2 * i]);
        const float m1 = sqrtf(re[2 * i + 1] * re[2 * i + 1] + im[2 * i + 1] * im[2 * i + 1]);
        out[i] = (m0 + m1) / 2;
    }
    __syncthreads();
    if (t == 0) out[0] = out[1];
}

__device__ inline uint16_t fft_color(int height)
{
    int red = 0, green = 0, blue = 0;
    const float c1 = 0.1f, c2 = 0.4f, c3 = 0.5f;
    if (height < 30 * c1) blue = (int)(uint8_t)(int)(height * 255 / (30 * c1));
    else if (height < 30 * (c1 + c2)) {
        green = (int)(uint8_t)(int)((height - 30 * c1) * 255 / ((30 - 30 * c1) * (c1 + c2)));
        red = green;
        blue = (255 - green) & 255;
    } else {
        red = 255;
        green = (int)(uint8_t)(int)(255 - (height - (30 * (c1 + c2))) * 255 / ((30 - (30 * (c1 + c2))) * (c1 + c2 + c3)));
    }
    return (uint16_t)(((red & 0xF8) << 8) | ((green & 0xFC) << 3) | (blue >> 3));
}

/* Pass C (one wave = one channel, sequential over its frames; a lane owns four neighbouring
 * bins): auto-level, normalise, temporal smoothing, then the waterfall row (fft-shifted
 * RGB565) and the over-range count that feeds the next frame's auto-level.  The frame
 * maximum and the count are wave reductions, so a frame costs no barrier, and the waterfall
 * is updated in the same pass. */
__global__ __launch_bounds__(64) void ddc_fft_level(AudioState *__restrict__ states, const float *__restrict__ mags,
                                                    uint32_t frames_per_ch, float *__restrict__ bins,
                                                    uint16_t *__restrict__ rows, uint16_t *__restrict__ wtf)
{
    __shared__ uint16_t colour[32]; // fft_color of the 30 heights a pixel can have
    const uint32_t ch = blockIdx.x, x0 = threadIdx.x * 4u;
    if (threadIdx.x < 32u) colour[threadIdx.x] = fft_color((int)threadIdx.x);
    __syncthreads();
    AudioState &s = states[ch];
    float mean[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mean[k] = s.fft_mean[x0 + k];
    float max_value = s.fft_max_value;
    uint32_t max_errors = s.fft_max_errors;
    const float avg = (float)s.p.fft_averaging;
    const bool loopback = s.p.mode == DDC_MODE_LOOPBACK;
    // the caller's buffers take 16-byte / 8-byte stores when they are aligned for them
    const bool wide = ((reinterpret_cast<uintptr_t>(bins) & 15u) | (reinterpret_cast<uintptr_t>(rows) & 7u)) == 0;
    const uint32_t nx0 = x0 < 128u ? x0 + 128u : x0 - 128u; // fft shift; four neighbours stay neighbours
    const size_t fr0 = (size_t)ch * frames_per_ch;
    // the magnitudes are fetched four frames ahead of the frame being levelled
    float4 ahead[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
        ahead[j] = j < frames_per_ch ? reinterpret_cast<const float4 *>(mags + (fr0 + j) * 256 + x0)[0] : make_float4(0, 0, 0, 0);
    // waterfall (wtf_buffer, fft.c:353-376): every frame scrolls the picture down by one row and lands
    // on top, so after this call's F frames the old rows sit F lower.  They move first (all loads, then
    // all stores: a lane only ever touches its own four columns); the new rows are stored as they are made.
    const uint32_t F = frames_per_ch < 50u ? frames_per_ch : 50u;
    uint16_t *w = wtf + (size_t)ch * 50 * 256;
    {
        uint2 old[50];
#pragma unroll
        for (uint32_t y = 0; y < 50; y++)
            if (y + F < 50u) old[y] = reinterpret_cast<const uint2 *>(w + y * 256 + x0)[0];
#pragma unroll
        for (uint32_t y = 0; y < 50; y++)
            if (y + F < 50u) reinterpret_cast<uint2 *>(w + (y + F) * 256 + x0)[0] = old[y];
    }
    for (uint32_t f0 = 0; f0 < frames_per_ch; f0 += 4)
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t f = f0 + j;
        if (f >= frames_per_ch) break;
        const size_t fr = fr0 + f;
        const float v[4] = { ahead[j].x, ahead[j].y, ahead[j].z, ahead[j].w };
        if (f + 4 < frames_per_ch) ahead[j] = reinterpret_cast<const float4 *>(mags + (fr + 4) * 256 + x0)[0];
        const float mx = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        {
            const float diff = (mx - max_value) / 10.0f;
            const uint32_t e = max_errors;
            float mv = max_value;
            if (e >= 6 && diff > 0) mv += diff;
            else if (e <= 1 && diff < 0 && diff < -10.0f) mv += diff;
            else if (e <= 1 && mv > 10.0f) mv -= 10.0f;
            else if (e <= 1 && diff < 0 && diff < -1.0f) mv += diff;
            else if (e <= 1 && mv > 1.0f) mv -= 1.0f;
            if (mv < 100.0f) mv = 100.0f;
            if (loopback) mv = 60000;
            max_value = mv;
        }
        const float inv = 1.0f / max_value;
        uint16_t c[4];
        uint32_t errs = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float xn = v[k] * inv;
            if (mean[k] < xn) mean[k] += (xn - mean[k]) / avg;
            else mean[k] -= (mean[k] - xn) / avg;
            const int h = (int)(uint16_t)f2i_sat(mean[k] * 30);
            const bool over = h > 29;
            errs += (uint32_t)__popcll(__ballot(over));
            c[k] = over ? (uint16_t)0xF800 : colour[h & 31];
        }
        max_errors = errs;
        const uint2 px = make_uint2((uint32_t)c[0] | ((uint32_t)c[1] << 16), (uint32_t)c[2] | ((uint32_t)c[3] << 16));
        if (wide) {
            if (bins) reinterpret_cast<float4 *>(bins + fr * 256 + x0)[0] = make_float4(mean[0], mean[1], mean[2], mean[3]);
            reinterpret_cast<uint2 *>(rows + fr * 256 + nx0)[0] = px;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (bins) bins[fr * 256 + x0 + k] = mean[k];
                rows[fr * 256 + nx0 + k] = c[k];
            }
        }
        // the pixels of this lane's own columns in the shifted row were made by the lane half a wave away
        const uint2 mine = make_uint2((uint32_t)__shfl_xor((int)px.x, 32, 64), (uint32_t)__shfl_xor((int)px.y, 32, 64));
        const uint32_t y = frames_per_ch - 1u - f; // the last frame is row 0
        if (y < 50u) reinterpret_cast<uint2 *>(w + y * 256 + x0)[0] = mine;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) s.fft_mean[x0 + k] = mean[k];
    if (threadIdx.x == 0) { s.fft_max_value = max_value; s.fft_max_errors = max_errors; }
}

/* FFT_moveWaterfall (fft.c:453-503): on a retune the waterfall moves sideways by
 * `shift` pixels (zero fill) and the smoothed bins are moved in place in the same
 * loop order as the firmware's, which is not a clean rotation.  One workgroup;
 * thread y < 50 walks row y, thread 50 the bins. */
__global__ void ddc_fft_move_waterfall_k(AudioState *__restrict__ states, uint16_t *__restrict__ wtf, uint32_t ch, int shift)
{
    const int y = threadIdx.x;
    if (shift == 0) return;
    if (y < 50) {
        uint16_t *row = wtf + ((size_t)ch * 50 + y) * 256;
        if (shift > 0)
            for (int x = 0; x < 256; x++) { const int nx = x + shift; row[x] = nx >= 256 ? (uint16_t)0 : row[nx]; }
        else
            for (int x = 255; x >= 0; x--) { const int nx = x + shift; row[x] = nx < 0 ? (uint16_t)0 : row[nx]; }
    } else if (y == 50) {
        float *mean = states[ch].fft_mean;
        if (shift > 0)
            for (int x = 0; x < 256; x++) { int fx = x + shift; if (fx >= 256) fx -= 256; if (fx >= 0 && fx < 256) mean[x] = mean[fx]; }
        else
            for (int x = 255; x >= 0; x--) { int fx = x + shift; if (fx < 0) fx += 256; if (fx >= 0 && fx < 256) mean[x] = mean[fx]; }
    }
}

/* ---- TX audio -------------------------------------------------------------------- */

__device__ __forceinline__ uint32_t f2u_sat(float v)
{
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

__constant__ float c_tx_hilbert[2][201]; // [0] = +45 degree (I) taps, [1] = -45 degree (Q) taps

/*
 * processTxAudio for every channel, first half: thread = channel, sequential over
 * blocks of 192 frames.  mic_lr: [channel][n_frames][2] int16 (left, right).
 * Everything that is recursion in time runs here: DC blockers (slots 2, 3), the
 * HPF / LPF lattices (the firmware's shared instances IIR_HPF_I, IIR_LPF_I), the
 * ALC compressor (one gain per block from the block's peak) and the FM phase
 * accumulator.  For the modes that end in the +-45 degree FIR pair (SSB, DIGI, AM)
 * the gain-scaled samples go to xs[channel][200 + n_frames] (the 200 samples the
 * FIR delay lines held come first) and the second kernel finishes the block; the
 * other modes (CW, FM, IQ, LOOPBACK, tune) write their output here.
 */
__device__ __forceinline__ bool tx_mode_has_fir(int m)
{
    return m == DDC_MODE_USB || m == DDC_MODE_DIGI_U || m == DDC_MODE_LSB || m == DDC_MODE_DIGI_L || m == DDC_MODE_AM;
}

__global__ __launch_bounds__(64) void ddc_audio_tx(AudioState *__restrict__ states, TxAudioState *__restrict__ txs,
                                                   uint32_t n_ch, const int16_t *__restrict__ mic_lr, uint32_t n_frames,
                                                   int16_t *__restrict__ iq_out, int32_t *__restrict__ loop_lr,
                                                   float *__restrict__ xs)
{
    __shared__ float blockbuf[64 * kABiPitch];
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_ch) return;
    AudioState &s = states[ch];
    TxAudioState &x = txs[ch];
    float *bi = &blockbuf[threadIdx.x * kABiPitch];
    const ddc_audio_params p = s.p;
    const int m = p.mode;
    const bool tune = p.tune != 0;
    const bool fir = tx_mode_has_fir(m) && !tune;
    const bool lpf = p.filter_width > 0;
    const float amp0 = p.rf_power / 100.0f * 31000.0f; // MAX_TX_AMPLITUDE, settings.h:13
    const int lpf_n = s.lpf_n;
    float lk[11], lv[12], hk[11], hv[12], gh[11], gl[11];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        lk[i] = ddc_lpf_k[s.lpf_idx][i]; hk[i] = i < 6 ? ddc_hpf_k[s.hpf_idx][i] : 0.0f;
        gh[i] = s.hpf_i.g[i]; gl[i] = s.lpf_i.g[i];
    }
#pragma unroll
    for (int i = 0; i < 12; i++) { lv[i] = ddc_lpf_v[s.lpf_idx][i]; hv[i] = i < 7 ? ddc_hpf_v[s.hpf_idx][i] : 0.0f; }
    float dx2 = s.dc_x[2], dy2 = s.dc_y[2], dx3 = s.dc_x[3], dy3 = s.dc_y[3];
    float alc = x.alc_gain, fa = x.fm_hpf_a, fb = x.fm_hpf_b, mxa = x.max_amplitude;
    const float fm_index = x.fm_index;
    uint32_t facc = x.fm_accum;
    const float A1 = 0.99951171875f;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(mic_lr) + (size_t)ch * n_frames;
    uint32_t *dst = reinterpret_cast<uint32_t *>(iq_out) + (size_t)ch * n_frames;
    int32_t *lp = loop_lr ? loop_lr + (size_t)ch * n_frames * 2 : nullptr;
    float *xrow = xs + (size_t)ch * (200 + (size_t)n_frames);
    if (fir)
        for (int i = 0; i < 200; i++) xrow[i] = x.fir_i[i];
    for (uint32_t b0 = 0; b0 + kABlock <= n_frames; b0 += kABlock) {
        float amp = amp0;
        if (tune) {
            const uint32_t w = (uint32_t)(uint16_t)(int16_t)f2i_sat(amp);
            for (int i = 0; i < kABlock; i++) {
                dst[b0 + i] = w | (w << 16);
                if (lp) { lp[2 * (b0 + i)] = 0; lp[2 * (b0 + i) + 1] = 0; }
            }
            continue;
        }
        float mx = 0;
        for (int i0 = 0; i0 < kABlock; i0 += 16) {
            uint4 in4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) in4[k] = reinterpret_cast<const uint4 *>(src + b0 + i0)[k];
            const uint32_t in[16] = { in4[0].x, in4[0].y, in4[0].z, in4[0].w, in4[1].x, in4[1].y, in4[1].z, in4[1].w,
                                      in4[2].x, in4[2].y, in4[2].z, in4[2].w, in4[3].x, in4[3].y, in4[3].z, in4[3].w };
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t v = in[k];
                float fi = (float)(int16_t)(v & 0xFFFFu), fq = (float)((int32_t)v >> 16);
                { const float d = fi - dx2; const float a = A1 * dy2; const float o = d + a; dx2 = fi; dy2 = o; fi = o; }
                { const float d = fq - dx3; const float a = A1 * dy3; const float o = d + a; dx3 = fq; dy3 = o; fq = o; }
                if (m == DDC_MODE_IQ) { // raw stereo through the DC blockers
                    float oi = fi, oq = fq;
                    if (p.mute) { oi = oi * 0; oq = oq * 0; }
                    dst[b0 + i0 + k] = (uint32_t)(uint16_t)(int16_t)f2i_sat(oi) | ((uint32_t)(uint16_t)(int16_t)f2i_sat(oq) << 16);
                    if (lp) { lp[2 * (b0 + i0 + k)] = 0; lp[2 * (b0 + i0 + k) + 1] = 0; }
                    continue;
                }
                fi = lattice_reg<6>(gh, hk, hv, fi);
                if (lpf) fi = lpf_n == 11 ? lattice_reg<11>(gl, lk, lv, fi) : lattice_reg<7>(gl, lk, lv, fi);
                const float a = fabsf(fi);
                if (a > mx) mx = a;
                bi[i0 + k] = fi;
            }
        }
        if (m == DDC_MODE_IQ) continue;
        // RF power control / compressor, audio_processor.c:112-145
        if (mx == 0.0f) mx = 0.001f;
        mxa = mx;
        const float target = amp / mx;
        if (target > alc) alc += (target - alc) / 500.0f;
        else alc -= (alc - target) / 500.0f;
        if (target < alc) alc = target;
        if (alc < 0.0f) alc = 0.0f;
        if ((alc * mx) > (amp * 1.1f)) alc = target;
        if (alc > 500.0f) alc = 500.0f;
        if (mx < 15.0f) alc = 0.0f;
        if ((alc > 1.0f) && (m == DDC_MODE_DIGI_L || m == DDC_MODE_DIGI_U || m == DDC_MODE_LOOPBACK)) alc = 1.0f;
        if (fir) {
            for (int i = 0; i < kABlock; i++) xrow[200 + b0 + i] = bi[i] * alc;
            continue;
        }
        const bool cw = m == DDC_MODE_CW_L || m == DDC_MODE_CW_U;
        const bool fmm = m == DDC_MODE_NFM || m == DDC_MODE_WFM;
        if (cw && !p.key) amp = 0;
        const float vol = (float)p.volume / 50.0f;
        for (int i = 0; i < kABlock; i++) {
            float oi = bi[i] * alc, oq = oi;
            if (cw) { oi = amp; oq = amp; }
            else if (fmm) { // ModulateFM, audio_processor.c:590-619
                fb = 0.95f * (fb + oi - fa);
                fa = oi;
                facc = f2u_sat((float)facc + fb) % 48000u;
                const float sd = (facc / (float)48000) * kPiF * fm_index;
                oi = amp * arm_sin_with(g_arm_sin_table, sd);
                oq = amp * arm_cos_with(g_arm_sin_table, sd);
            }
            if (p.mute) { oi = oi * 0; oq = oq * 0; }
            if (m == DDC_MODE_LOOPBACK) {
                const int32_t v = f2i_sat(oi * vol);
                if (lp) { lp[2 * (b0 + i)] = v; lp[2 * (b0 + i) + 1] = v; }
                dst[b0 + i] = 0;
            } else {
                dst[b0 + i] = (uint32_t)(uint16_t)(int16_t)f2i_sat(oi) | ((uint32_t)(uint16_t)(int16_t)f2i_sat(oq) << 16);
                if (lp) { lp[2 * (b0 + i)] = 0; lp[2 * (b0 + i) + 1] = 0; }
            }
        }
    }
    if (fir && n_frames >= kABlock) { // what the FIR delay lines hold after the call
        const uint32_t done = n_frames / kABlock * kABlock;
        for (int i = 0; i < 200; i++) { const float v = xrow[done + i]; x.fir_i[i] = v; x.fir_q[i] = v; }
    }
#pragma unroll
    for (int i = 0; i < 11; i++) { s.hpf_i.g[i] = gh[i]; s.lpf_i.g[i] = gl[i]; }
    s.dc_x[2] = dx2; s.dc_y[2] = dy2; s.dc_x[3] = dx3; s.dc_y[3] = dy3;
    x.alc_gain = alc; x.fm_hpf_a = fa; x.fm_hpf_b = fb; x.fm_accum = facc; x.max_amplitude = mxa;
}

/*
 * processTxAudio, second half, for the channels in an SSB / DIGI / AM mode: the
 * +-45 degree Hilbert pair (arm_fir_f32, 201 taps each, audio_filters.c:41-42) over
 * the gain-scaled samples, then sideband selection / AM carrier, mute and the
 * narrowing to the 16-bit bus words.  One thread per output frame; a workgroup
 * covers 256 consecutive frames of one channel from 456 staged samples.  Every
 * output is one accumulator fed tap by tap in ascending order, as the CMSIS
 * routine does it, so the result does not depend on the parallel split.
 */
__global__ __launch_bounds__(256) void ddc_audio_tx_fir(const AudioState *__restrict__ states, uint32_t tiles,
                                                        const float *__restrict__ xs, uint32_t n_frames,
                                                        int16_t *__restrict__ iq_out, int32_t *__restrict__ loop_lr)
{
    __shared__ float win[256 + 200];
    const uint32_t ch = blockIdx.x / tiles, n0 = (blockIdx.x - ch * tiles) * 256u, t = threadIdx.x;
    const ddc_audio_params p = states[ch].p;
    const int m = p.mode;
    if (!tx_mode_has_fir(m) || p.tune) return; // uniform over the workgroup
    const uint32_t done = n_frames / kABlock * kABlock;
    const float *xrow = xs + (size_t)ch * (200 + (size_t)n_frames);
    for (uint32_t j = t; j < 456; j += 256) win[j] = n0 + j < 200 + done ? xrow[n0 + j] : 0.0f;
    __syncthreads();
    const uint32_t n = n0 + t;
    if (n >= done) return;
    float ai = 0.0f, aq = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 201; k++) {
        const float v = win[t + k];
        ai += v * c_tx_hilbert[0][k];
        aq += v * c_tx_hilbert[1][k];
    }
    float oi, oq;
    if (m == DDC_MODE_USB || m == DDC_MODE_DIGI_U) { oi = aq; oq = ai; }
    else { oi = ai; oq = aq; }
    if (m == DDC_MODE_AM) {
        const float amp = p.rf_power / 100.0f * 31000.0f;
        const float i_am = ((oi - oq) + amp) / 2.0f;
        const float q_am = ((oq - oi) - amp) / 2.0f;
        oi = i_am; oq = q_am;
    }
    if (p.mute) { oi = oi * 0; oq = oq * 0; }
    reinterpret_cast<uint32_t *>(iq_out)[(size_t)ch * n_frames + n] =
        (uint32_t)(uint16_t)(int16_t)f2i_sat(oi) | ((uint32_t)(uint16_t)(int16_t)f2i_sat(oq) << 16);
    if (loop_lr) { loop_lr[((size_t)ch * n_frames + n) * 2] = 0; loop_lr[((size_t)ch * n_frames + n) * 2 + 1] = 0; }
}

/* Operator settings that do not select a filter, changed without touching the
 * running state (the firmware just reads TRX.* on the next block). */
__global__ void ddc_audio_update_k(AudioState *__restrict__ states, uint32_t lo, uint32_t hi, ddc_audio_params p)
{
    const uint32_t ch = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= hi) return;
    AudioState &s = states[ch];
    s.p = p;
    s.agc_up = 500.0f / (float)p.agc_speed;
    s.agc_down = s.agc_up / 10.0f;
}

// clears the S-meter extremes of one channel once they have been read
__global__ void ddc_audio_smeter_clear_k(AudioState *__restrict__ states, uint32_t ch)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { states[ch].smeter_max = 0; states[ch].smeter_min = 0; }
}

/* ---- single stages, for callers that drive the firmware's fine-grained functions ---- */

/* One stage of the RX audio path on one channel's state, one thread:
 * DDC_STAGE_DC   dc_filter(buf, n, slot = arg)              audio_filters.c:358-373
 * DDC_STAGE_AGC  DoAGC(buf, 192)                            agc.c:21-67
 * DDC_STAGE_DNR  processNoiseReduction(buf -> buf), 64      noise_reduction.c:19-37
 * DDC_STAGE_CW   CWDecoder_Process(buf), 192, clock arg ms  cw_decoder.c:56-174 */
__global__ void ddc_audio_stage_k(AudioState *__restrict__ states, uint32_t ch, int stage, int arg,
                                  float *__restrict__ buf, int n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    AudioState &s = states[ch];
    if (stage == DDC_STAGE_DC) {
        for (int i = 0; i < n; i++) buf[i] = dc_step(s, arg, buf[i]);
    } else if (stage == DDC_STAGE_AGC) {
        agc_block(s, buf);
    } else if (stage == DDC_STAGE_DNR) {
        for (int i = 0; i < kASub; i++) s.lms_ref[s.ref_new + i] = buf[i];
        lms_block(s, buf);
        s.ref_old += kASub;
        if (s.ref_old >= 128) s.ref_old = 0;
        s.ref_new = s.ref_old + kASub;
        if (s.ref_new >= 128) s.ref_new = 0;
    } else if (stage == DDC_STAGE_CW) {
        cw_block(s, buf, (int64_t)arg);
    }
}


/*
 * USB-audio frames of the RX path (STM32/Src/audio_processor.c:415-432): every
 * int32 entry of the codec buffer becomes one int16,
 *     (int16_t)((int32_t)Processor_AudioBuffer[i] * (1.0f / TRX.Volume * 100.0f)),
 * i.e. the volume scaling of :364-374 is taken out again for the USB host.  The
 * float -> int16 cast is what the Cortex-M4F does with it: vcvt.s32.f32 (towards
 * zero, saturating, NaN -> 0) and the low 16 bits of that (Volume 0 makes the
 * factor infinite: 0 stays 0, anything else saturates to 0xFFFF / 0x0000).
 * audio_lr / usb_lr: [channel][n_frames][2].
 */
__device__ __forceinline__ int16_t usb_word(int32_t v, float gain)
{
#pragma clang fp contract(off)
    const float f = (float)v * gain;
    int32_t i;
    if (f != f) i = 0;
    else if (f >= 2147483648.0f) i = 2147483647;
    else if (f <= -2147483648.0f) i = (-2147483647 - 1);
    else i = (int32_t)f;
    return (int16_t)(uint16_t)((uint32_t)i & 0xFFFFu);
}

__global__ __launch_bounds__(256) void ddc_audio_usb_frames(const AudioState *__restrict__ st, uint32_t n_ch,
                                                           const int32_t *__restrict__ audio_lr, uint32_t n_frames,
                                                           int16_t *__restrict__ usb_lr)
{
    // one thread per two frames (16 bytes in, 8 bytes out)
    const uint32_t per_ch = n_frames / 2;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n_ch * per_ch) return;
    const uint32_t ch = (uint32_t)(gid / per_ch);
    const float gain = (1.0f / (float)st[ch].p.volume) * 100.0f;
    const int4 v = reinterpret_cast<const int4 *>(audio_lr)[gid];
    uint2 o;
    o.x = (uint32_t)(uint16_t)usb_word(v.x, gain) | ((uint32_t)(uint16_t)usb_word(v.y, gain) << 16);
    o.y = (uint32_t)(uint16_t)usb_word(v.z, gain) | ((uint32_t)(uint16_t)usb_word(v.w, gain) << 16);
    reinterpret_cast<uint2 *>(usb_lr)[gid] = o;
}

} // namespace ddc
