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
 * like the RX ingest it is bound by vector-ALU issue (NCO + two 64-bit
 * polynomial walks per sample).
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

// 14-bit NCO outputs for one phase (no saturation can occur: see tests)
__device__ __forceinline__ void nco14(const uint2 *tab, uint32_t acc32, int &s14, int &c14)
{
    uint2 e;
    uint32_t w;
    nco_fetch(tab, acc32, e, w);
    s14 = dot2_sconst(e.x, w, 8192) >> 14;
    c14 = dot2_sconst(e.y, w, 8192) >> 14;
}

// running-sum walk of sum_e c[e] * C(u+e, e), u = 0, 1, 2, ...
struct PolyWalk {
    long long q3, q2, q1, p, c4;
    __device__ __forceinline__ void init(const long long (&c)[5])
    {
        c4 = c[4];
        q3 = c[3] + c[4];
        q2 = c[2] + q3;
        q1 = c[1] + q2;
        p = c[0] + q1;
    }
    // the same walk entered at u = 504: p and its backward differences there,
    // nabla^k p(504) = sum_{e>=k} c[e] C(504 + e - k, e - k)
    __device__ __forceinline__ void init_at504(const long long (&c)[5])
    {
        const long long k1 = 505, k2 = 127765, k3 = 21592285, k4 = 2742220195ll; // C(504 + m, m)
        c4 = c[4];
        q3 = c[3] + c[4] * k1;
        q2 = c[2] + c[3] * k1 + c[4] * k2;
        q1 = c[1] + c[2] * k1 + c[3] * k2 + c[4] * k3;
        p = c[0] + c[1] * k1 + c[2] * k2 + c[3] * k3 + c[4] * k4;
    }
    __device__ __forceinline__ void next()
    {
        q3 += c4;
        q2 += q3;
        q1 += q2;
        p += q1;
    }
    // tx_cic output register: bits 59..46 of the 2^8-scaled sum
    __device__ __forceinline__ int out14() const { return (int)(p >> 32) >> 6; }
};

/*
 * One thread writes one aligned block of 512 DAC samples (1 KB) of one channel:
 * block g (1..n96) is t = 512 (g - 1) + k, k = 0..511.  The interpolated tx_cic
 * output is piecewise polynomial with pieces ("segments") that start 8 samples
 * into a block:
 *   segment G covers t = 512 G + 8 + u, u = 0..511,
 *   cic(t-3) = (sum_q p_q(u) y[G-1-q]) >> 38,  p_q(u) = sum_e COEF[q][e] C(u+e, e)
 *   dac[t]   = (((I*sin14 + Q*cos14)(phase(t-2)) >> 14) + 8191) & 0x3FFF
 * so the first 8 samples of block g come from segment g - 2 at u = 504..511 and
 * the other 504 from segment g - 1 at u = 0..503.
 *
 * The 16 bytes a thread produces per group of 8 samples are collected in LDS for
 * 8 groups; then 8 adjacent lanes write one whole 128-byte line (the blocks are
 * line-aligned), instead of 64 lanes writing 16 bytes each 1 KB apart.
 * ybuf: [channel][6 + n96][2]; dac: [channel][512 n96] uint16.
 */
__device__ __forceinline__ void tx_segment_coeffs(const uint32_t *y, long long (&ci)[5], long long (&cq)[5])
{
#pragma unroll
    for (int e = 0; e < 5; e++) { ci[e] = 0; cq[e] = 0; }
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const uint32_t v = y[-q];
        const long long yi = (int16_t)(v & 0xFFFFu), yq = (int)v >> 16;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            ci[e] += COEF[q][e] * yi;
            cq[e] += COEF[q][e] * yq;
        }
    }
}

__global__ __launch_bounds__(256) void ddc_tx_duc(const int16_t *__restrict__ ybuf, uint32_t n96,
                                                  uint32_t total, const uint32_t *__restrict__ acc0,
                                                  const uint32_t *__restrict__ phi32v,
                                                  uint16_t *__restrict__ dac,
                                                  const uint2 *__restrict__ nco_tab)
{
    __shared__ uint2 tab[2048];
    __shared__ uint4 obuf[8 * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave_base = tid & ~63u;
    for (uint32_t i = tid; i < 2048; i += blockDim.x) tab[i] = nco_tab[i];

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
    {
        long long ci[5], cq[5];
        tx_segment_coeffs(ybase + 3 + g, ci, cq); // segment g - 2
        wi.init_at504(ci);
        wq.init_at504(cq);
    }
    // stored phase / increment are << 10; the sample loop runs on the << 5 form of nco_fetch
    const uint32_t phi_in = phi32v[ch];
    const uint32_t phi32 = phi_in >> 5;
    uint32_t ph = (acc0[ch] + (512u * (g - 1) - 2u) * phi_in) >> 5;
    const uint32_t mine = tid * 8u, swz = line_swz(tid);
    __syncthreads(); // tab is ready
#pragma unroll 1
    for (int grp = 0; grp < 64; grp++) {
        if (grp == 1) { // the block's own segment starts here
            long long ci[5], cq[5];
            tx_segment_coeffs(ybase + 4 + g, ci, cq); // segment g - 1
            wi.init(ci);
            wq.init(cq);
        }
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int s14, c14;
            nco14(tab, ph, s14, c14);
            const int sum = __mul24(wi.out14(), s14) + __mul24(wq.out14(), c14) + (8191 << 14);
            const uint32_t d = __builtin_amdgcn_ubfe((uint32_t)sum, 14, 14);
            if (k & 1) w[k / 2] |= d << 16; else w[k / 2] = d;
            ph += phi32;
            wi.next();
            wq.next();
        }
        obuf[mine + (((uint32_t)grp & 7u) ^ swz)] = make_uint4(w[0], w[1], w[2], w[3]);
        if ((grp & 7) == 7) {
            __syncthreads();
            const uint32_t line = (uint32_t)grp >> 3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 v = obuf[(wave_base + 8u * j) * 8u + lane];
                if (dst16[j] != 0xFFFFFFFFu)
                    reinterpret_cast<uint4 *>(dac)[(size_t)dst16[j] + line * 8u] = v;
            }
            __syncthreads();
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

inline int ddc_tx_state_create(ddc_tx_state *s, uint32_t n_channels, hipStream_t st)
{
    s->n_channels = n_channels;
    const size_t n = n_channels;
    if (hipMalloc(&s->d_acc, n * 4) != hipSuccess) return -12;
    if (hipMalloc(&s->d_hist_x, n * ddc::kTxXHist * 4) != hipSuccess) return -12;
    if (hipMalloc(&s->d_hist_y, n * ddc::kTxYHist * 4) != hipSuccess) return -12;
    (void)hipMemsetAsync(s->d_acc, 0, n * 4, st);
    (void)hipMemsetAsync(s->d_hist_x, 0, n * ddc::kTxXHist * 4, st);
    (void)hipMemsetAsync(s->d_hist_y, 0, n * ddc::kTxYHist * 4, st);
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
