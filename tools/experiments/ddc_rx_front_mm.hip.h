/*
 * ddc_rx_front_mm: the RX ingest kernel (NCO + mixer + rx_cic integrate/decimate,
 * FPGA/nco.qsys, FPGA/mixer.v, FPGA/rx_mixer_shift.v, FPGA/rx_cic.vhd:147-404)
 * with the window moments accumulated by the matrix cores.
 *
 * A thread still owns one natural block of 512 samples and mixes it sample by
 * sample exactly as ddc_rx_front does.  What changes is the reduction
 *     B[e] = sum_d C(d + e, e) * x[d],   d = distance of a sample from the window end,
 * which is a product of the sample vector with a constant 512 x 5 matrix.  The
 * weights reach 2^32, the samples 16 bits, so both are cut into signed bytes:
 * five weights -> 1 + 2 + 3 + 4 + 5 = 15 byte planes (columns), samples -> a high
 * and a low plane.  One v_mfma_i32_32x32x32_i8 then multiplies 16 samples of all
 * 64 windows of a wave with the 15 columns:
 *
 *   lane l supplies, as the A operand, 16 consecutive sample bytes of its window;
 *   the instruction reads lanes 0..31 / 32..63 as the K halves of rows l % 32.
 *   The B operand carries the columns in lanes 0..15 (first K half) and 48..63
 *   (second K half) and zeros elsewhere, so output column j < 16 collects the
 *   windows of lanes 0..31 and column 16 + j those of lanes 32..63:
 *       C[i][j] = column j % 16 of window i + 32 (j / 16).
 *
 * The 32-bit accumulators take all 32 steps of a block without overflow
 * (512 * 128 * 128 = 2^23); at the end they are transposed through LDS to the
 * window's own lane and recombined into the five 64-bit moments.  The running
 * sums, the local dot products and the 5x5 basis change of ddc_rx_front are gone;
 * per sample the vector ALU is left with the NCO, the mixer and three byte
 * shuffles.  The five samples before a block (they open its window, d = 507..511)
 * arrive from the neighbouring thread at the end and take one more step of the
 * matrix instruction; the last five of the block carry zero weight.
 *
 * Output records, carries and every bit of the result are those of ddc_rx_front.
 *
 * Opt-in (UA3REO_DDC_FRONT=mm when a ctx is created): the default ingest kernel is
 * ddc_rx_front, which keeps the whole path on the vector ALU.  This form measures
 * about 6 % faster; it is kept to document what the matrix cores buy here.
 */
#pragma once
#include "ddc_rx_kernels.hip.h"

namespace ddc {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMmThreads = kFrontThreads;
constexpr int kMmSteps = 32;             // 16-sample steps per block
constexpr int kMmTabRows = kMmSteps + 1; // one row of 16 columns x 16 bytes per step (column 15 is zero in every row)
                                         // + the row of the five samples before the block
constexpr int kMmLastD = 506;            // distance of the block's first sample from its window's end

// column of byte plane p (0 = least significant) of weight e
__host__ __device__ constexpr int mm_col(int e, int p) { return e * (e + 1) / 2 + p; }

// 128 * sum_{d=0..511} C(d + e, e) = 128 * C(512 + e, e + 1): what the low sample plane, fed as
// (byte ^ 0x80) = byte - 128, leaves out of every moment
__host__ __device__ constexpr unsigned long long mm_low_plane_bias(int e)
{
    return 128ull * (unsigned long long)binom_gen(kBlk + e, e + 1);
}

// weight table for the B operand: [kMmTabRows][16 columns][16 sample bytes]
inline bool build_mm_table(uint8_t *tab /* kMmTabRows * 256 bytes */)
{
    for (int i = 0; i < kMmTabRows * 256; i++) tab[i] = 0;
    for (int kk = 0; kk < kMmSteps; kk++)
        for (int t = 0; t < 16; t++) {
            const int s = 16 * kk + t;
            if (s > kMmLastD) continue; // the newest five samples belong to the next window
            const int d = kMmLastD - s;
            for (int e = 0; e < 5; e++) {
                long long v = binom_gen(d + e, e);
                for (int p = 0; p <= e; p++) {
                    const int dig = (int)(((v + 128) & 255) - 128); // signed digit base 256
                    v = (v - dig) / 256;
                    tab[(kk * 16 + mm_col(e, p)) * 16 + t] = (uint8_t)(int8_t)dig;
                }
                if (v != 0) return false;
            }
        }
    // last row: the five samples before the block sit in byte slots 11..15 (d = 511..507)
    for (int t = 11; t < 16; t++) {
        const int d = kBlk - 1 - (t - 11);
        for (int e = 0; e < 5; e++) {
            long long v = binom_gen(d + e, e);
            for (int p = 0; p <= e; p++) {
                const int dig = (int)(((v + 128) & 255) - 128);
                v = (v - dig) / 256;
                tab[(kMmSteps * 16 + mm_col(e, p)) * 16 + t] = (uint8_t)(int8_t)dig;
            }
            if (v != 0) return false;
        }
    }
    return true;
}

template <int G>
__device__ __forceinline__ void group8m(const uint2 *tab, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t &acc32,
                                        uint32_t phi32, unsigned (&pi)[8], unsigned (&pq)[8])
{
    int a[8];
    unpack8(d0, d1, d2, a);
    uint2 e[8];
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nco_fetch(tab, acc32 + (uint32_t)k * phi32, e[k], w[k]);
    acc32 += 8u * phi32;
    asm volatile("" : "+v"(acc32)); // as in group8: keeps the multiples of phi32 out of registers
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int pi0, pq0, pi1, pq1;
        nco_finish(e[2 * k], w[2 * k], a[2 * k], pi0, pq0);
        nco_finish(e[2 * k + 1], w[2 * k + 1], a[2 * k + 1], pi1, pq1);
        pi[4 * G + k] = pack_x2(pi0, pi1);
        pq[4 * G + k] = pack_x2(pq0, pq1);
    }
}

// 16 samples (8 packed pairs of 2 x) of every window of the wave into the accumulators
__device__ __forceinline__ void mm_step(const unsigned (&pk)[8], const v4i bw, v16i &acc_hi, v16i &acc_lo)
{
    v4i hi, lo;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        hi[m] = (int)__builtin_amdgcn_perm(pk[2 * m + 1], pk[2 * m], 0x07050301u);
        lo[m] = (int)(__builtin_amdgcn_perm(pk[2 * m + 1], pk[2 * m], 0x06040200u) ^ 0x80808080u);
    }
    acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, bw, acc_hi, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, bw, acc_lo, 0, 0, 0);
}

__global__ __launch_bounds__(kMmThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void ddc_rx_front_mm(
    const uint8_t *__restrict__ adc, size_t row_bytes, uint32_t n_blk, uint32_t total_blocks,
    const uint32_t *__restrict__ acc0, const uint32_t *__restrict__ phi32v,
    const uint32_t *__restrict__ carry_in, uint32_t *__restrict__ carry_out,
    unsigned long long *__restrict__ mom, const uint2 *__restrict__ nco_tab, const uint4 *__restrict__ mm_tab)
{
    __shared__ uint2 tab[2048];
    __shared__ uint4 lines[8 * kMmThreads];
    // weight rows of the current and of the next line period (at most 6 steps each), see below
    __shared__ uint4 wbuf[2][6 * 16];
    __shared__ uint4 whead[16]; // the row of the five samples before the block

    const uint32_t tid = threadIdx.x;
    const uint32_t g0 = blockIdx.x * kMmThreads;
    for (uint32_t i = tid; i < 2048; i += kMmThreads) tab[i] = nco_tab[i];
    if (tid < 4 * 16) wbuf[0][tid] = mm_tab[tid]; // steps 0..3
    else if (tid < 5 * 16) whead[tid - 4 * 16] = mm_tab[kMmSteps * 16 + tid - 4 * 16];

    // staging sources, as in ddc_rx_front
    const uint32_t lane = tid & 63u, wave_base = tid & ~63u;
    // (kept as 32-bit offsets from the channel row of the wave's first block: the 64 blocks of a
    // wave lie in that row and the few behind it, which the host bounds; the row pitch may be 0)
    uint32_t src[8];
    const uint8_t *wave_src;
    {
        uint32_t go = g0 + wave_base;
        go = go < total_blocks ? go : total_blocks - 1;
        const uint32_t ch0 = go / n_blk;
        {
            // the same value in every lane of the wave: tell the compiler, so that it stays scalar
            const uint64_t o = (uint64_t)ch0 * row_bytes;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)o), hi = __builtin_amdgcn_readfirstlane((uint32_t)(o >> 32));
            wave_src = adc + (((uint64_t)hi << 32) | lo);
        }
        go = g0 + wave_base + (lane >> 3);
        go = go < total_blocks ? go : total_blocks - 1;
        uint32_t cho = go / n_blk, bo = go - cho * n_blk;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = wave_base + 8u * j + (lane >> 3);
            const uint32_t piece = (lane & 7u) ^ line_swz(row);
            src[j] = (uint32_t)((size_t)(cho - ch0) * row_bytes + (size_t)bo * kBlkBytes) + piece * 16u;
            if (go + 8u < total_blocks) {
                go += 8u;
                bo += 8u;
                while (bo >= n_blk) { bo -= n_blk; cho++; }
            }
        }
    }

    const uint32_t g = g0 + tid;
    const bool active = g < total_blocks;
    const uint32_t gc = active ? g : total_blocks - 1;
    const uint32_t ch = gc / n_blk, b = gc - ch * n_blk;
    const uint32_t phi_in = phi32v[ch];
    const uint32_t phi32 = phi_in >> 5;
    uint32_t acc32 = (acc0[ch] + (b * (uint32_t)kBlk) * phi_in) >> 5;

#pragma unroll
    for (int j = 0; j < 8; j++) lds_dma16(wave_src + src[j], &lines[(wave_base + 8 * j) * 8]);

    // B operand: lanes 0..15 and 48..63 carry column (lane & 15) of the step's weight row, the
    // others column 15, which is zero in every row.  The rows come from LDS: the table (8 KB) does
    // not fit beside three workgroups' buffers, so the workgroup keeps only the rows of the current
    // line period there and wave 0 requests those of the next period together with the next line
    // (same wait, same barrier).  Periods: a half block is three lines holding 2, 3 and 3 chunks,
    // i.e. 4, 6 and 6 steps.
    const uint32_t wlane = ((lane < 16u || lane >= 48u) ? (lane & 15u) : 15u) * 16u;
    const char *wbase = reinterpret_cast<const char *>(wbuf);

    v16i aih = {0}, ail = {0}, aqh = {0}, aql = {0};

    __syncthreads(); // tab is ready

    const uint32_t mine = tid * 128u + (line_swz(tid) << 4);
    const char *lbase = reinterpret_cast<const char *>(lines);
#define DDC_PIECE(p) (*reinterpret_cast<const uint4 *>(lbase + (mine ^ ((p) << 4))))

    unsigned pi[8], pq[8];
    uint4 left0 = make_uint4(0, 0, 0, 0), left1 = make_uint4(0, 0, 0, 0);
#pragma unroll 1
    for (int a = 0; a < kChunksPerBlk; a++) {
        const int c = a & 7;
        if (c == 0 || c == 2 || c == 5) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // period of this chunk and the row of its first step within the period
        const int period = (a >> 3) * 3 + (c < 2 ? 0 : c < 5 ? 1 : 2);
        const int prow = c < 2 ? 2 * c : c < 5 ? 2 * (c - 2) : 2 * (c - 5);
        const uint32_t woff = (uint32_t)((period & 1) * 6 * 256 + prow * 256) + wlane;
        const uint4 bfirst = *reinterpret_cast<const uint4 *>(wbase + woff);
        const uint4 bsecond = *reinterpret_cast<const uint4 *>(wbase + woff + 256);
        uint4 v0, v1, v2;
        switch (c) {
        case 0: v0 = DDC_PIECE(0); v1 = DDC_PIECE(1); v2 = DDC_PIECE(2); break;
        case 1: v0 = DDC_PIECE(3); v1 = DDC_PIECE(4); v2 = DDC_PIECE(5); left0 = DDC_PIECE(6); left1 = DDC_PIECE(7); break;
        case 2: v0 = left0; v1 = left1; v2 = DDC_PIECE(0); break;
        case 3: v0 = DDC_PIECE(1); v1 = DDC_PIECE(2); v2 = DDC_PIECE(3); break;
        case 4: v0 = DDC_PIECE(4); v1 = DDC_PIECE(5); v2 = DDC_PIECE(6); left0 = DDC_PIECE(7); break;
        case 5: v0 = left0; v1 = DDC_PIECE(0); v2 = DDC_PIECE(1); break;
        case 6: v0 = DDC_PIECE(2); v1 = DDC_PIECE(3); v2 = DDC_PIECE(4); break;
        default: v0 = DDC_PIECE(5); v1 = DDC_PIECE(6); v2 = DDC_PIECE(7); break;
        }
        const bool line_done = c == 1 || c == 4 || c == 7;
        if (line_done) { // last chunk of a line: free the buffer, start the next line
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (a + 1 < kChunksPerBlk) {
                const int next_line = (a >> 3) * 3 + (c == 1 ? 1 : c == 4 ? 2 : 3);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    lds_dma16(wave_src + (src[j] + next_line * 128), &lines[(wave_base + 8 * j) * 8]);
                if (wave_base == 0) { // the next period's weight rows: 4 or 6 rows of 256 bytes
                    const int np = period + 1;
                    const int first_row = (np / 3) * 16 + (np % 3 == 0 ? 0 : np % 3 == 1 ? 4 : 10);
                    const uint8_t *wsrc = reinterpret_cast<const uint8_t *>(mm_tab + first_row * 16 + lane);
                    lds_dma16(wsrc, &wbuf[np & 1][0]);
                    if (np % 3 != 0 && lane < 32) lds_dma16(wsrc + 1024, &wbuf[np & 1][64]);
                }
            }
        }
        group8m<0>(tab, v0.x, v0.y, v0.z, acc32, phi32, pi, pq);
        group8m<1>(tab, v0.w, v1.x, v1.y, acc32, phi32, pi, pq);
        {
            const v4i bw = { (int)bfirst.x, (int)bfirst.y, (int)bfirst.z, (int)bfirst.w };
            mm_step(pi, bw, aih, ail);
            mm_step(pq, bw, aqh, aql);
        }
        group8m<0>(tab, v1.z, v1.w, v2.x, acc32, phi32, pi, pq);
        group8m<1>(tab, v2.y, v2.z, v2.w, acc32, phi32, pi, pq);
        {
            const v4i bw = { (int)bsecond.x, (int)bsecond.y, (int)bsecond.z, (int)bsecond.w };
            mm_step(pi, bw, aih, ail);
            mm_step(pq, bw, aqh, aql);
        }
    }
#undef DDC_PIECE
    // mixer outputs of the block's last six samples (pairs 5..7 of the last step)
    const unsigned last_i[3] = { pi[5], pi[6], pi[7] }, last_q[3] = { pq[5], pq[6], pq[7] };

    // The five samples before this block open its window: handed over by the thread
    // of the block to the left through LDS (the line buffer is free now); the first
    // thread of the workgroup mixes them itself, the first block of a call takes
    // them from the previous call.
    unsigned pki[3], pkq[3];
    {
        uint32_t *xch = reinterpret_cast<uint32_t *>(lines);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            xch[k * kMmThreads + tid] = last_i[k];
            xch[(3 + k) * kMmThreads + tid] = last_q[k];
        }
        __syncthreads();
        if (b == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pki[k] = carry_in[ch * 6 + k];
                pkq[k] = carry_in[ch * 6 + 3 + k];
            }
        } else if (tid == 0) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(
                adc + (size_t)ch * row_bytes + (size_t)b * kBlkBytes - 12);
            int a[8];
            unpack8(p[0], p[1], p[2], a);
            uint32_t ph = (acc0[ch] + (b * (uint32_t)kBlk - 6u) * phi_in) >> 5;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int pi0, pq0, pi1, pq1;
                nco_mix(tab, ph, a[2 + 2 * k], pi0, pq0);
                ph += phi32;
                nco_mix(tab, ph, a[3 + 2 * k], pi1, pq1);
                ph += phi32;
                pki[k] = pack_x2(pi0, pi1);
                pkq[k] = pack_x2(pq0, pq1);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pki[k] = xch[k * kMmThreads + tid - 1];
                pkq[k] = xch[(3 + k) * kMmThreads + tid - 1];
            }
        }
        __syncthreads(); // the exchange area is reused below
    }
    {
        // one more step for them: byte slots 10..15 (slot 10, the sixth sample back, has weight zero)
        const uint4 hw = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(whead) + wlane);
        const v4i bw = { (int)hw.x, (int)hw.y, (int)hw.z, (int)hw.w };
        const unsigned hi_[8] = { 0, 0, 0, 0, 0, pki[0], pki[1], pki[2] }, hq_[8] = { 0, 0, 0, 0, 0, pkq[0], pkq[1], pkq[2] };
        mm_step(hi_, bw, aih, ail);
        mm_step(hq_, bw, aqh, aql);
    }

    // Accumulator register r of lane l is row i = 8 (r / 4) + r % 4 + 4 (l / 32) of column
    // j = l % 32, i.e. column j % 16 of window i + 32 (j / 16).  Every wave transposes its two
    // planes through its own 8 KB of the line buffer: T[plane][window][16 columns].
    uint32_t *T = reinterpret_cast<uint32_t *>(lines) + (wave_base >> 6) * 2048u;
    const uint32_t jcol = lane & 31u;
    const uint32_t tw = (4u * (lane >> 5) + 32u * (jcol >> 4)) * 16u + (jcol & 15u);
    unsigned long long Bi[5], Bq[5];
#pragma unroll
    for (int br = 0; br < 2; br++) {
        const v16i &hi = br ? aqh : aih, &lo = br ? aql : ail;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            T[tw + (8 * (r / 4) + r % 4) * 16] = (uint32_t)hi[r];
            T[1024 + tw + (8 * (r / 4) + r % 4) * 16] = (uint32_t)lo[r];
        }
        __syncthreads();
        int chi[16], clo[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 h4 = reinterpret_cast<const uint4 *>(T + lane * 16u)[k];
            const uint4 l4 = reinterpret_cast<const uint4 *>(T + 1024 + lane * 16u)[k];
            chi[4 * k] = (int)h4.x; chi[4 * k + 1] = (int)h4.y; chi[4 * k + 2] = (int)h4.z; chi[4 * k + 3] = (int)h4.w;
            clo[4 * k] = (int)l4.x; clo[4 * k + 1] = (int)l4.y; clo[4 * k + 2] = (int)l4.z; clo[4 * k + 3] = (int)l4.w;
        }
        __syncthreads();
        unsigned long long (&B)[5] = br ? Bq : Bi;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            unsigned long long acc = mm_low_plane_bias(e);
#pragma unroll
            for (int p = 0; p <= e; p++) {
                const long long t = (long long)chi[mm_col(e, p)] * 256 + (long long)clo[mm_col(e, p)];
                acc += (unsigned long long)t << (8 * p);
            }
            B[e] = acc;
        }
    }

    if (!active) return;
    unsigned long long *rec = mom + ((size_t)ch * (kMomHist + n_blk) + kMomHist + b) * 10;
#pragma unroll
    for (int e = 0; e < 5; e++) {
        rec[e] = Bi[e];
        rec[5 + e] = Bq[e];
    }
    if (b == n_blk - 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            carry_out[ch * 6 + k] = last_i[k];
            carry_out[ch * 6 + 3 + k] = last_q[k];
        }
    }
}

} // namespace ddc
