/*
 * TX (DUC) kernels of the MI355X DDC: 48 kHz I/Q -> 14-bit DAC words at
 * 1024x the rate.
 *
 * Replaces, bit for bit, the FPGA TX datapath of the reference:
 *   tx_ciccomp (x2)  FPGA/tx_ciccomp.vhd:81-128,224-240,332-469
 *   tx_cic (x2)      FPGA/tx_cic.vhd:155-416
 *   tx_mixer (x2)    FPGA/tx_mixer.v:55-71 (full 14-bit NCO outputs)
 *   tx_summator      FPGA/tx_summator.v:59-80
 *   DAC_corrector    FPGA/DAC_corrector.v:15-21
 *
 * tx_cic's register widths equal the worst-case growth of every stage, so none
 * of its 60-bit wraps can trigger and all its in-chain shifts drop zero bits
 * only: the interpolator is exactly the linear filter (1,...,1)_512^{*5} on the
 * zero-stuffed 96 kHz stream.  Between two input words its output is therefore
 * one degree-4 polynomial in the sample index; a thread owns one such 512-sample
 * segment, gets the polynomial's coefficients from five input words and walks
 * it with four running sums per sample (no recursion across segments).
 *
 * The kernel writes 2 bytes per DAC sample and reads 4 bytes per 1024 samples;
 * like the RX ingest it is bound by vector-ALU issue (NCO + two exact
 * polynomial walks per sample on 64-bit integers).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddc_rx_kernels.hip.h"

namespace ddc {

constexpr int kTxXHist = 24; // tx_ciccomp reaches back 23 inputs
constexpr int kTxYHist = 6;  // a DAC sample reaches back 5 (+1 for the segment before the call)

__constant__ int16_t c_tx_p1[24];
__constant__ int16_t c_tx_p2[24];

/*
 * tx_ciccomp: y[2n] = cround(sum_k cp1[k] x[n-k], 14), y[2n+1] likewise with
 * cp2 (low 30 bits of the sum, wrap16).  x: this call's [channel][n48][2] plus
 * the saved 24-word history for negative indices.  ybuf: [channel][6 + 2 n48][2].
 */
__global__ void ddc_tx_ciccomp(const int16_t *__restrict__ iq48k, const int16_t *__restrict__ hist_x,
                               uint32_t n48, uint32_t n_ch, int16_t *__restrict__ ybuf)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n_ch * n48) return;
    const uint32_t ch = (uint32_t)(gid / n48), n = (uint32_t)(gid - (uint64_t)ch * n48);
    const uint32_t *x = reinterpret_cast<const uint32_t *>(iq48k) + (size_t)ch * n48;
    const uint32_t *hx = reinterpret_cast<const uint32_t *>(hist_x) + (size_t)ch * kTxXHist;
    uint32_t a1i = 0, a1q = 0, a2i = 0, a2q = 0;
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const int idx = (int)n - k;
        const uint32_t v = idx >= 0 ? x[idx] : hx[kTxXHist + idx];
        const int xi = (int)(int16_t)(v & 0xFFFFu), xq = (int)v >> 16;
        const int p1 = c_tx_p1[k], p2 = c_tx_p2[k];
        a1i += (uint32_t)(p1 * xi); a1q += (uint32_t)(p1 * xq);
        a2i += (uint32_t)(p2 * xi); a2q += (uint32_t)(p2 * xq);
    }
    uint32_t *y = reinterpret_cast<uint32_t *>(ybuf) + (size_t)ch * (kTxYHist + 2 * (size_t)n48) + kTxYHist + 2 * (size_t)n;
    y[0] = (uint32_t)(uint16_t)cround32(a1i, 14, 30) | ((uint32_t)(uint16_t)cround32(a1q, 14, 30) << 16);
    y[1] = (uint32_t)(uint16_t)cround32(a2i, 14, 30) | ((uint32_t)(uint16_t)cround32(a2q, 14, 30) << 16);
}

// 14-bit NCO outputs (no saturation can occur: see tests).  tab2 holds
// the coarse entries doubled ({4 cc | 4 sc << 16, -4 sc | 4 cc << 16}, still 16-bit) and the
// {fine sine, 8191} operand is doubled too, so the angle sums come out times four
// and the 14-bit values sit in the upper halves, where a multiply selects them for
// free; s14 / c14 are returned in that form (value << 16 plus rounding remainder).
// Both phases of an even / odd pair (one 64-bit phase sum) at a time: the two fine-sine
// multiply-adds are one packed instruction and the two doublings one 64-bit shift (bit 31 of a
// {fine sine, 8191} operand is zero, so nothing crosses between the dwords).
__device__ __forceinline__ void nco14x_pair(const uint2 *tab2, unsigned long long php, int (&s14x)[2], int (&c14x)[2])
{
    uint2 e0, e1;
    uint32_t w0, w1;
    nco_fetch_pair(tab2, php, e0, e1, w0, w1);
    const unsigned long long w = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
    unsigned long long w2;
    asm("v_lshl_add_u64 %0, %1, 1, 0" : "=v"(w2) : "v"(w));
    s14x[0] = dot2_sconst(e0.x, (uint32_t)w2, 32768);
    c14x[0] = dot2_sconst(e0.y, (uint32_t)w2, 32768);
    s14x[1] = dot2_sconst(e1.x, (uint32_t)(w2 >> 32), 32768);
    c14x[1] = dot2_sconst(e1.y, (uint32_t)(w2 >> 32), 32768);
}
__device__ __forceinline__ uint2 nco_entry_doubled(uint2 e)
{
    // each 16-bit half doubled (|2 cc|, |2 sc| <= 16382)
    return make_uint2(((e.x & 0x7FFF7FFFu) << 1), ((e.y & 0x7FFF7FFFu) << 1));
}

// Running-sum walk of the interpolated tx_cic output over one segment,
//   P(u) = sum_q p_q(u) y[G-1-q],  p_q(u) = sum_e COEF[q][e] C(u+e, e) = h[512 q + u]  (u = 0..511),
// by its backward differences: one step is four 64-bit additions.
//
// p_q(u) are the taps of (1,...,1)_512^{*5}: non-negative, and the five of one u
// sum to 2^36.  So |P| <= 2^51 for 16-bit y.  The start values are sums of at
// most 25 products of a constant and a 16-bit word and lie in [-2^51, 2^51)
// (checked at compile time): they are formed in double precision (exact, 25
// v_fma_f64 per branch), turned into integers by adding 1.5 * 2^52 and
// subtracting that number's bit pattern, and scaled by 2^10.  The walk itself
// runs on those 64-bit integers (one v_lshl_add_u64 per addition), and the
// tx_cic output register, bits 51..38 of P, is the signed upper half of the high
// dword of 2^10 P: the multiply that consumes it selects that half for free.
__host__ __device__ constexpr long long tx_start_const(int u, int k, int q)
{
    // nabla^k p_q(u) = sum_{e>=k} COEF[q][e] C(u + e - k, e - k)
    long long v = 0;
    for (int e = k; e < 5; e++) v += COEF[q][e] * binom_gen(u + e - k, e - k);
    return v;
}
constexpr bool tx_start_consts_exact()
{
    // every start value (and every partial sum on the way to it) must lie in [-2^51, 2^51)
    // for words in [-32768, 32767]
    for (int u : { 0, 504 })
        for (int k = 0; k < 5; k++) {
            long long pos = 0, neg = 0;
            for (int q = 0; q < 5; q++) {
                const long long v = tx_start_const(u, k, q);
                if (v < 0) neg -= v; else pos += v;
            }
            if (pos * 32767 + neg * 32768 >= (1ll << 51)) return false;
            if (pos * 32768 + neg * 32767 > (1ll << 51)) return false;
        }
    return true;
}
static_assert(tx_start_consts_exact(), "a start value of the TX walk would leave [-2^51, 2^51)");

struct PolyWalk {
    unsigned long long q3, q2, q1, p, c4; // 2^10 times the polynomial and its differences, mod 2^64
    // state at u = U (0 or 504) of the segment whose newest input word is y[0]
    // (y[q] = q words older); B = 0 takes the I halves, 1 the Q halves
    template <int U, int B>
    __device__ __forceinline__ void start(const uint32_t (&y)[5])
    {
        double v[5];
#pragma unroll
        for (int q = 0; q < 5; q++) v[q] = (double)(B ? (int)y[q] >> 16 : (int)(int16_t)(y[q] & 0xFFFFu));
        constexpr double kToInt = 0x1.8p+52; // 1.5 * 2^52: an integer below 2^51 in magnitude lands in the mantissa
        double a[5] = { kToInt, kToInt, kToInt, kToInt, kToInt };
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const double c = (double)tx_start_const(U, k, q);
                if (c != 0.0) a[k] = __builtin_fma(c, v[q], a[k]);
            }
        unsigned long long w[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            w[k] = (__builtin_bit_cast(unsigned long long, a[k]) - __builtin_bit_cast(unsigned long long, kToInt)) << 10;
        p = w[0]; q1 = w[1]; q2 = w[2]; q3 = w[3]; c4 = w[4];
    }
    __device__ __forceinline__ void next()
    {
        q3 = pair_add(q3, c4);
        q2 = pair_add(q2, q3);
        q1 = pair_add(q1, q2);
        p = pair_add(p, q1);
    }
    // tx_cic output register: bits 59..46 of the 2^8-scaled sum = P >> 38 = bits 63..48 of 2^10 P
    __device__ __forceinline__ int out14() const { return (int)(uint32_t)(p >> 32) >> 16; }
};

// tx_mixer + tx_summator + the DAC offset: I * sin14 + Q * cos14 + (8191 << 14) as two multiply-adds.  All four
// factors sit in upper halves — the tx_cic outputs in bits 63..48 of the walks, the NCO values in bits 31..16 of
// the angle sums — and v_mad_i32_i16 picks that half of each source by itself (two SDWA multiplies and a
// three-operand add before).
__device__ __forceinline__ uint32_t tx_mix(unsigned long long p_i, unsigned long long p_q, int s14x, int c14x)
{
    int t, r;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(t) : "v"((uint32_t)(p_i >> 32)), "v"(s14x), "s"(8191 << 14));
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"((uint32_t)(p_q >> 32)), "v"(c14x), "v"(t));
    return (uint32_t)r;
}

/*
 * One thread writes one aligned block of 512 DAC samples (1 KB) of one channel:
 * block g (1..n96) is t = 512 (g - 1) + k, k = 0..511.  The interpolated tx_cic
 * output is piecewise polynomial with pieces ("segments") that start 8 samples
 * into a block:
 *   segment G covers t = 512 G + 8 + u, u = 0..511,
 *   cic(t-3) = P_G(u) >> 38 with the input words y[G-1-q], q = 0..4
 *   dac[t]   = (((I*sin14 + Q*cos14)(phase(t-2)) >> 14) + 8191) & 0x3FFF
 * so the first 8 samples of block g come from segment g - 2 at u = 504..511 and
 * the other 504 from segment g - 1 at u = 0..503.
 *
 * The 16 bytes a thread produces per group of 8 samples are collected in LDS for
 * 8 groups; then 8 adjacent lanes write one whole 128-byte line (the blocks are
 * line-aligned), instead of 64 lanes writing 16 bytes each 1 KB apart.
 * ybuf: [channel][6 + n96][2]; dac: [channel][512 n96] uint16.
 */
__global__ __launch_bounds__(256) void ddc_tx_duc(const int16_t *__restrict__ ybuf, uint32_t n96,
                                                  uint32_t total, const uint32_t *__restrict__ acc0,
                                                  const uint32_t *__restrict__ phi32v,
                                                  uint16_t *__restrict__ dac,
                                                  const uint2 *__restrict__ nco_tab)
{
    __shared__ uint2 tab[2048];
    __shared__ uint4 obuf[8 * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave_base = tid & ~63u;
    for (uint32_t i = tid; i < 2048; i += blockDim.x) tab[i] = nco_entry_doubled(nco_tab[i]);

    // flush targets: store j of a wave covers rows wave_base + 8 j .. + 7, lane l
    // writing, for row 8 j + (l >> 3), the piece held in slot l & 7; offsets in
    // 16-byte units from dac, ~0 = no such block
    uint32_t dst16[8];
    {
        const uint32_t g0 = blockIdx.x * 256u + wave_base + (lane >> 3);
        uint32_t gid = g0 < total ? g0 : total - 1;
        uint32_t cho = gid / n96, go = gid - cho * n96;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = wave_base + 8u * j + (lane >> 3);
            const uint32_t piece = (lane & 7u) ^ line_swz(row);
            const bool ok = blockIdx.x * 256u + row < total;
            dst16[j] = ok ? (uint32_t)(((size_t)cho * n96 + go) * 64u) + piece : 0xFFFFFFFFu;
            if (gid + 8u < total) {
                gid += 8u;
                go += 8u;
                while (go >= n96) { go -= n96; cho++; }
            }
        }
    }

    const uint32_t gid0 = blockIdx.x * 256u + tid;
    const bool active = gid0 < total;
    const uint32_t gidc = active ? gid0 : total - 1;
    const uint32_t ch = gidc / n96, g = gidc - ch * n96 + 1; // block g = 1..n96
    // buffer index of y[G-1] for segment G is G + 5 (six words of history in front)
    const uint32_t *ybase = reinterpret_cast<const uint32_t *>(ybuf) + (size_t)ch * (kTxYHist + n96);
    PolyWalk wi, wq;
    uint32_t yw[6]; // yw[j] = y[g + 4 - j]: words 1..5 feed segment g - 2, words 0..4 segment g - 1
#pragma unroll
    for (int j = 0; j < 6; j++) yw[j] = ybase[4 + g - j];
    {
        const uint32_t y5[5] = { yw[1], yw[2], yw[3], yw[4], yw[5] };
        wi.start<504, 0>(y5);
        wq.start<504, 1>(y5);
    }
    // stored phase / increment are << 10; the sample loop runs on the << 8 form of nco_fetch
    const uint32_t phi_in = phi32v[ch];
    const uint32_t phi32 = phi_in >> 2;
    const uint32_t ph0 = (acc0[ch] + (512u * (g - 1) - 2u) * phi_in) >> 2;
    // even / odd sample phases as the two dwords of one 64-bit sum (see group8 of the RX kernel)
    unsigned long long php = (unsigned long long)ph0 | ((unsigned long long)(ph0 + phi32) << 32);
    const unsigned long long phi2 = (unsigned long long)(2u * phi32) | ((unsigned long long)(2u * phi32) << 32);
    const uint32_t mine = tid * 8u, swz = line_swz(tid);
    __syncthreads(); // tab is ready
#pragma unroll 1
    for (int grp = 0; grp < 64; grp++) {
        if (grp == 1) { // the block's own segment starts here
            const uint32_t y5[5] = { yw[0], yw[1], yw[2], yw[3], yw[4] };
            wi.start<0, 0>(y5);
            wq.start<0, 1>(y5);
        }
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t sum[2];
            int s14x[2], c14x[2];
            nco14x_pair(tab, php, s14x, c14x);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                sum[h] = tx_mix(wi.p, wq.p, s14x[h], c14x[h]);
                wi.next();
                wq.next();
            }
            php = pair_add(php, phi2);
            // a DAC word is bits 27..14 of its sum; two per dword
            uint32_t lo, hi;
            asm("v_bfe_u32 %0, %1, 14, 14" : "=v"(lo) : "v"(sum[0]));
            hi = sum[1] & 0x0FFFC000u;
            asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(w[k]) : "v"(hi), "v"(lo));
        }
        obuf[mine + (((uint32_t)grp & 7u) ^ swz)] = make_uint4(w[0], w[1], w[2], w[3]);
        if ((grp & 7) == 7) {
            // the rows a wave flushes are its own lanes' rows: LDS operations of one wave complete in
            // order, so the exchange needs no workgroup barrier, only that the compiler keeps the order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t line = (uint32_t)grp >> 3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 v = obuf[(wave_base + 8u * j) * 8u + lane];
                if (dst16[j] != 0xFFFFFFFFu)
                    reinterpret_cast<uint4 *>(dac)[(size_t)dst16[j] + line * 8u] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

__global__ void ddc_tx_load_history(const int16_t *hist_y, int16_t *ybuf, uint32_t n96)
{
    const uint32_t ch = blockIdx.x, t = threadIdx.x;
    if (t < kTxYHist)
        reinterpret_cast<uint32_t *>(ybuf)[(size_t)ch * (kTxYHist + n96) + t] =
            reinterpret_cast<const uint32_t *>(hist_y)[(size_t)ch * kTxYHist + t];
}

// one workgroup of 64 threads per channel
__global__ void ddc_tx_save_history(int16_t *hist_x, int16_t *hist_y, const int16_t *iq48k,
                                    const int16_t *ybuf, uint32_t *acc0, const uint32_t *phi32v,
                                    uint32_t n48)
{
    const uint32_t ch = blockIdx.x, t = threadIdx.x, n96 = 2 * n48;
    uint32_t *hx = reinterpret_cast<uint32_t *>(hist_x) + (size_t)ch * kTxXHist;
    uint32_t xv = 0;
    if (t < kTxXHist) {
        const long long idx = (long long)n48 - kTxXHist + t; // stream index of the new history word t
        xv = idx >= 0 ? reinterpret_cast<const uint32_t *>(iq48k)[(size_t)ch * n48 + idx] : hx[kTxXHist + idx];
    }
    __syncthreads();
    if (t < kTxXHist) hx[t] = xv;
    if (t < kTxYHist)
        reinterpret_cast<uint32_t *>(hist_y)[(size_t)ch * kTxYHist + t] =
            reinterpret_cast<const uint32_t *>(ybuf)[(size_t)ch * (kTxYHist + n96) + n96 + t];
    if (t == 0) acc0[ch] += (n48 * 1024u) * phi32v[ch];
}

} // namespace ddc

struct ddc_tx_state {
    uint32_t n_channels = 0;
    uint32_t *d_acc = nullptr;    // NCO phase << 10 at the first DAC sample of the next call
    int16_t *d_hist_x = nullptr;  // [ch][24][2]
    int16_t *d_hist_y = nullptr;  // [ch][6][2]
    int16_t *d_ybuf = nullptr;    // [ch][6 + 2 cap48][2]
    uint32_t cap48 = 0;
    // staging for the host-pointer entry point
    int16_t *d_in = nullptr;
    size_t in_cap = 0;
    uint16_t *d_out = nullptr;
    size_t out_cap = 0;
};

// upload_tables: also write the compensator taps into this device's __constant__ memory
inline int ddc_tx_state_create(ddc_tx_state *s, uint32_t n_channels, hipStream_t st, bool upload_tables)
{
    s->n_channels = n_channels;
    const size_t n = n_channels;
    if (hipMalloc(&s->d_acc, n * 4) != hipSuccess) return -12;
    if (hipMalloc(&s->d_hist_x, n * ddc::kTxXHist * 4) != hipSuccess) return -12;
    if (hipMalloc(&s->d_hist_y, n * ddc::kTxYHist * 4) != hipSuccess) return -12;
    if (hipMemsetAsync(s->d_acc, 0, n * 4, st) != hipSuccess) return -5;
    if (hipMemsetAsync(s->d_hist_x, 0, n * ddc::kTxXHist * 4, st) != hipSuccess) return -5;
    if (hipMemsetAsync(s->d_hist_y, 0, n * ddc::kTxYHist * 4, st) != hipSuccess) return -5;
    if (!upload_tables) return 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(ddc::c_tx_p1), tx_ciccomp_p1, sizeof tx_ciccomp_p1, 0,
                               hipMemcpyHostToDevice, st) != hipSuccess) return -5;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(ddc::c_tx_p2), tx_ciccomp_p2, sizeof tx_ciccomp_p2, 0,
                               hipMemcpyHostToDevice, st) != hipSuccess) return -5;
    return 0;
}

inline void ddc_tx_state_destroy(ddc_tx_state *s)
{
    void *p[] = { s->d_acc, s->d_hist_x, s->d_hist_y, s->d_ybuf, s->d_in, s->d_out };
    for (void *q : p)
        if (q) (void)hipFree(q);
    *s = ddc_tx_state();
}
