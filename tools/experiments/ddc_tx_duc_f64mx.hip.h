// Experiment, not part of the library: ddc_tx_duc with the interpolated tx_cic output from
// v_mfma_f64_16x16x4_f64 instead of the difference walk.  Bit-exact against the oracle and the walk
// kernel (all TX tests), but 0.912 ms against 0.66 ms at 256 channels x 2^22 DAC words on MI355X: the
// double-precision matrix instruction takes 64 cycles per SIMD (tools/experiments/mfma_probe_f64.hip)
// and does not run beside the vector ALU the way the int8 one does, so four of them per 256 words
// cost what sixteen vector instructions per word would, twice the eight the walk takes.
// To build it: include this file behind ddc_tx_kernels.hip.h in ddc_lib.hip, build the table with
// build_tx_mx_table at ddc_create, launch <<<(total + 63) / 64, 256>>> in place of ddc_tx_duc.
#pragma once
namespace ddc {
/*
 * The same 512-sample blocks with the interpolated tx_cic output from the double-precision matrix
 * instruction instead of the difference walk.  P(u) = sum_q p_q(u) y[q] is a product of the
 * constant taps (rows: 16 samples, K: the five input words) and the blocks' input words (columns:
 * 16 blocks), v_mfma_f64_16x16x4_f64 twice per branch (K = 4 + 1).  Every partial sum is an integer
 * below 2^53 (the taps are non-negative and the five of one u sum to 2^36; inputs are 16-bit), so the
 * result is exact whatever order the unit adds in.  The table holds 2 p_q(u) and the accumulators
 * start at 1 - 2^38: D = 2 P - 2^38 + 1 is odd, D / 2^39 = P / 2^38 - 1/2 + 2^-39 is never half
 * way between two integers, and ONE v_fma_f64 (D, 2^-39, 1.5 * 2^52) leaves floor(P / 2^38), the
 * tx_cic output register, in the low dword of its result.
 *
 * A wave takes 16 blocks (lane mod 16); a step makes 16 DAC words of each: matrix row i is word
 * 4 (i mod 4) + i / 4 of the step, so that lane group q = lane / 16, which receives rows q, 4 + q,
 * 8 + q, 12 + q, holds the four consecutive words 4 q .. 4 q + 3.  Words 16 s + w of a block are
 * u = 16 s + w - 8 of its own segment; the first eight of step 0 belong to the segment before
 * (u = 504 .. 511): two more matrix steps with that segment's input words and a table whose other
 * rows are zero.  The NCO, the mixer and the DAC words are those of ddc_tx_duc.  Four steps make
 * one 128-byte line per block: through a wave-private LDS buffer, then eight lanes write a line.
 * mx_tab: [33][2][64] doubles (build_tx_mx_table); step s, matrix step m, lane.
 */
constexpr int kTxMxSteps = 32;
constexpr int kTxMxTabDoubles = (kTxMxSteps + 1) * 2 * 64;
constexpr uint32_t kTxMxRowPitch = 144; // bytes per block row of a wave's line buffer (128 + 16: rows on different banks)
constexpr uint32_t kTxMxBlocksPerWg = 64;

inline void build_tx_mx_table(double *tab /* kTxMxTabDoubles */)
{
    for (int s = 0; s <= kTxMxSteps; s++)
        for (int m = 0; m < 2; m++)
            for (int l = 0; l < 64; l++) {
                const int i = l % 16, k = l / 16;           // matrix row, K slot
                const int w = 4 * (i % 4) + i / 4;          // word of the step
                const int u = (s < kTxMxSteps ? 16 * s : 0) - 8 + w;
                const int q = m == 0 ? k : (k == 0 ? 4 : -1); // input word this K slot multiplies
                double v = 0.0;
                if (q >= 0) {
                    if (s < kTxMxSteps) {
                        if (u >= 0) v = 2.0 * (double)tx_start_const(u, 0, q);
                    } else if (u < 0) {
                        v = 2.0 * (double)tx_start_const(512 + u, 0, q); // step 0, the segment before
                    }
                }
                tab[(s * 2 + m) * 64 + l] = v;
            }
}

typedef double tx_v4d __attribute__((ext_vector_type(4)));

// floor(P / 2^38) from D = 2 P - 2^38 + 1: the low dword of fma(D, 2^-39, 1.5 * 2^52) (the compiler's own
// choice, the accumulate-in-place form, costs two moves per call to seed the addend)
__device__ __forceinline__ int tx_mx_out14(double d, double k39, double magic)
{
    double f;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(f) : "v"(d), "s"(k39), "v"(magic));
    return (int)(uint32_t)__builtin_bit_cast(unsigned long long, f);
}

__global__ __launch_bounds__(256, 3) void ddc_tx_duc_mx(const int16_t *__restrict__ ybuf, uint32_t n96, uint32_t total,
                                                     const uint32_t *__restrict__ acc0, const uint32_t *__restrict__ phi32v,
                                                     uint16_t *__restrict__ dac, const uint2 *__restrict__ nco_tab,
                                                     const double *__restrict__ mx_tab)
{
    __shared__ uint2 tab[2048];
    __shared__ __align__(16) char obuf[4][16 * kTxMxRowPitch];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, n = lane & 15u, q = lane >> 4;
    for (uint32_t i = tid; i < 2048; i += blockDim.x) tab[i] = nco_entry_doubled(nco_tab[i]);

    const uint32_t wave_gid0 = (blockIdx.x * 4u + wv) * 16u; // first block of the wave
    const uint32_t gid0 = wave_gid0 + n;
    const uint32_t gidc = gid0 < total ? gid0 : total - 1;   // lanes past the end compute a valid block and store nothing
    const uint32_t ch = gidc / n96, g = gidc - ch * n96 + 1;
    const uint32_t *ybase = reinterpret_cast<const uint32_t *>(ybuf) + (size_t)ch * (kTxYHist + n96);
    // input words: y[k] of the block's own segment is ybase[4 + g - k], of the segment before ybase[3 + g - k]
    const uint32_t w_own = ybase[4 + g - q], w_prev = ybase[3 + g - q], w_own4 = ybase[g], w_prev4 = ybase[g - 1];
    const double b1i = (double)(int)(int16_t)(w_own & 0xFFFFu), b1q = (double)((int)w_own >> 16);
    const double p1i = (double)(int)(int16_t)(w_prev & 0xFFFFu), p1q = (double)((int)w_prev >> 16);
    const double b2i = q == 0 ? (double)(int)(int16_t)(w_own4 & 0xFFFFu) : 0.0, b2q = q == 0 ? (double)((int)w_own4 >> 16) : 0.0;
    const double p2i = q == 0 ? (double)(int)(int16_t)(w_prev4 & 0xFFFFu) : 0.0, p2q = q == 0 ? (double)((int)w_prev4 >> 16) : 0.0;

    // phases as in ddc_tx_duc; the lane starts at word 4 q of the block and skips 12 words behind its four
    const uint32_t phi_in = phi32v[ch], phi32 = phi_in >> 2;
    const uint32_t ph0 = ((acc0[ch] + (512u * (g - 1) - 2u) * phi_in) >> 2) + 4u * q * phi32;
    unsigned long long php = (unsigned long long)ph0 | ((unsigned long long)(ph0 + phi32) << 32);
    const unsigned long long phi2 = (unsigned long long)(2u * phi32) | ((unsigned long long)(2u * phi32) << 32);
    const unsigned long long phi14 = (unsigned long long)(14u * phi32) | ((unsigned long long)(14u * phi32) << 32);

    char *wbuf = obuf[wv];
    char *wrow = wbuf + n * kTxMxRowPitch + 8u * q;
    const double *ta = mx_tab + lane;
    const double cinit = 1.0 - 0x1p38, k39 = 0x1p-39;
    double magic = 0x1.8p+52;
    asm volatile("" : "+v"(magic)); // kept in a register pair
    tx_v4d cconst = { cinit, cinit, cinit, cinit };
    asm volatile("" : "+v"(cconst));
    double a1 = ta[0], a2 = ta[64];
    const double pa1 = ta[(2 * kTxMxSteps) * 64], pa2 = ta[(2 * kTxMxSteps + 1) * 64];
    __syncthreads(); // tab is ready
#pragma unroll 1
    for (int s = 0; s < kTxMxSteps; s++) {
        // the accumulators start from the constant operand: no moves to seed them
        tx_v4d di, dq;
        if (s == 0) { // the first eight words come from the segment before
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1, p1i, cconst, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1, p1q, cconst, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2, p2i, di, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2, p2q, dq, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1i, di, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1q, dq, 0, 0, 0);
        } else {
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1i, cconst, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1q, cconst, 0, 0, 0);
        }
        di = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2i, di, 0, 0, 0);
        dq = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2q, dq, 0, 0, 0);
        if (s + 1 < kTxMxSteps) { // the next step's rows of the table
            a1 = ta[(2 * (s + 1)) * 64];
            a2 = ta[(2 * (s + 1) + 1) * 64];
        }
        uint32_t w[2];
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            int s14x[2], c14x[2];
            nco14x_pair(tab, php, s14x, c14x);
            php = pair_add(php, pr == 0 ? phi2 : phi14);
            uint32_t sum[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int ci = tx_mx_out14(di[2 * pr + h], k39, magic);
                const int cq = tx_mx_out14(dq[2 * pr + h], k39, magic);
                sum[h] = (uint32_t)(__mul24(ci, s14x[h] >> 16) + __mul24(cq, c14x[h] >> 16) + (8191 << 14));
            }
            uint32_t lo, hi;
            asm("v_bfe_u32 %0, %1, 14, 14" : "=v"(lo) : "v"(sum[0]));
            hi = sum[1] & 0x0FFFC000u;
            asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(w[pr]) : "v"(hi), "v"(lo));
        }
        *reinterpret_cast<uint2 *>(wrow + 32u * ((uint32_t)s & 3u)) = make_uint2(w[0], w[1]);
        if ((s & 3) == 3) {
            // the rows a wave flushes are its own: LDS operations of one wave complete in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t line = (uint32_t)s >> 2;
#pragma unroll
            for (uint32_t i = 0; i < 2; i++) {
                const uint32_t piece = lane + 64u * i, row = piece >> 3, col = piece & 7u;
                const uint4 v = *reinterpret_cast<const uint4 *>(wbuf + row * kTxMxRowPitch + col * 16u);
                if (wave_gid0 + row < total)
                    reinterpret_cast<uint4 *>(dac)[(size_t)(wave_gid0 + row) * 64u + line * 8u + col] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

} // namespace ddc
