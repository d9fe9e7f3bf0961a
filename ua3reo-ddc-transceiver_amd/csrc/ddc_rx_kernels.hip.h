/*
 * RX kernels of the MI355X DDC: packed 12-bit ADC words -> SPEC / VOICE I/Q.
 *
 * Replaces, bit for bit, the FPGA RX datapath of the reference:
 *   NCO            FPGA/nco.qsys:56-85, nco/synthesis/submodules/nco_nco_ii_0.v:299-420
 *   nco_shift      FPGA/nco_shift.v:9
 *   mixer (x2)     FPGA/mixer.v:55-71, FPGA/rx_mixer_shift.v:9
 *   rx_cic (x2)    FPGA/rx_cic.vhd:147-404
 *   rx_ciccomp(x2) FPGA/rx_ciccomp.vhd:81-146,339-361,425-616
 *   rx_hilb (I)    FPGA/rx_hilb.vhd:77-332,902-935
 *   data_delay (Q) FPGA/data_delay.v:16-28 with delay_length=130 (FPGA/UA3REO.bdf:1692-1696)
 *
 * The hardware is serial in time.  Here time is cut into windows of 512 samples
 * and every window is reduced independently to five exact 64-bit moments from
 * which the CIC outputs follow by a fixed 5x5 integer combination (algebra in
 * tools/cic_math.py), so one tuned channel parallelises over time and many
 * channels parallelise over both.
 *
 * Kernel 1 (ddc_rx_front) is the only one that touches the ADC stream; it is
 * bound by vector-ALU issue (about 18 vector instructions per sample), reads every
 * ADC byte once (1.5 B per sample) and writes 80 B of moments per 512 samples.
 * ddc_rx_front_cic is the same kernel for block counts that are a multiple of 256:
 * it also turns the moments into rx_cic outputs.  ddc_rx_front_mx
 * (ddc_rx_mx_kernel.hip.h) does what ddc_rx_front_cic does with the moment sums on
 * the matrix cores (about 14 vector instructions per sample) and is what the
 * library runs for such block counts; it shares the NCO, mixer and unpack code and
 * the rx_cic epilogue below.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddc_cic_consts.h"
#include "ddc_coeffs.h"

namespace ddc {

typedef short short2v __attribute__((ext_vector_type(2)));

constexpr int kChunk = 32;        // samples per chunk
constexpr int kChunksPerBlk = 16; // 512 / 32
constexpr int kBlk = 512;         // CIC decimation
constexpr int kBlkBytes = 768;    // 512 samples * 1.5 B
constexpr int kChunkBytes = 48;
constexpr int kFrontThreads = 256;
constexpr int kMomHist = 5;   // windows of history a CIC output reaches back
constexpr int kCicHist = 64;  // rx_ciccomp reaches back 64 inputs
constexpr int kCompHist = 256; // rx_hilb reaches back 255, data_delay 130

constexpr int PSI[5][32] = DDC_PSI_TABLE;
constexpr long long GAM[5][5] = DDC_GAMMA_TABLE;
constexpr long long COEF[5][5] = DDC_COEF_TABLE;

__host__ __device__ constexpr long long binom_gen(int n, int k)
{
    long long num = 1;
    for (int i = 0; i < k; i++) num *= (n - i);
    long long den = 1;
    for (int i = 1; i <= k; i++) den *= i;
    return num / den;
}
// chunk-local weight of sample s for moment i: C(15 - s + i, i)
__host__ __device__ constexpr int psi_w(int i, int s) { return (int)binom_gen(15 - s + i, i); }
__host__ __device__ constexpr unsigned pack_w(int lo, int hi)
{
    return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
}
constexpr bool psi_matches_header()
{
    for (int i = 0; i < 5; i++)
        for (int s = 0; s < 32; s++)
            if (psi_w(i, s) != PSI[i][s]) return false;
    return true;
}
static_assert(psi_matches_header(), "local moment weights differ from tools/cic_math.py");

__device__ __forceinline__ int dot2(unsigned a, unsigned w, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, w), c, false);
}

// v_dot2_i32_i16 with the addend in a scalar register (the compiler's own
// choice, the accumulate-in-place VOP2 form, costs a v_mov per call to seed it).
__device__ __forceinline__ int dot2_sconst(unsigned a, unsigned w, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(w), "s"(c));
    return r;
}

// Fine sine of the NCO in one float multiply-add.  The fine phase f (11 bits),
// read as a float, is the denormal f * 2^-149.  The float D with bit pattern
// 0x1FFF0000 has ulp 2^-87, so fma(f * 2^-149, 32 K * 2^62, D) = D + 32 f K ulp,
// rounded to nearest, carries the bit pattern 0x1FFF0000 + round(32 f K): the
// pair {fine sine, 8191} of 16-bit halves the NCO's dot products want.  With
// K = 6433 / 2^24 that is the fine ROM round(8191 sin(2 pi f / 2^22)) for every
// f (checked on the host when a ctx is created).  Needs denormals kept and
// round-to-nearest, the mode HIP kernels run in.
#define DDC_SF_FMA_C 0x1.921p55f          /* 6433 * 2^43 = 32 * 6433 / 2^24 * 2^62 */
#define DDC_SF_FMA_D_BITS 0x1FFF0000u
__host__ __device__ __forceinline__ uint32_t nco_fine_pair(uint32_t f)
{
    const float x = __builtin_bit_cast(float, f);
    return __builtin_bit_cast(uint32_t, __builtin_fmaf(x, DDC_SF_FMA_C, __builtin_bit_cast(float, DDC_SF_FMA_D_BITS)));
}

// signed bit-field extract kept as one instruction (the compiler otherwise folds
// it into a shift of the multiplicand and pays for that in the 12-bit unpack)
template <int OFF, int W>
__device__ __forceinline__ int sbfe_c(int v)
{
    int r;
    asm("v_bfe_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "n"(OFF), "n"(W));
    return r;
}

// NCO, first half: coarse-table fetch and the packed {fine sine, 8191} operand.
// tab: 2048 coarse entries {2*cc | 2*sc<<16, -2*sc | 2*cc<<16}.  acc30 holds
// the 22-bit phase accumulator in bits 29..8 (bits above are ignored, bits below
// are zero): its high half is the coarse phase times eight, the byte offset of
// the table entry, apart from the two bits on top, so one instruction that reads
// the high half and masks makes the address; the fine phase is bits 18..8.
__device__ __forceinline__ void nco_fetch(const uint2 *tab, uint32_t acc30, uint2 &e, uint32_t &w)
{
    uint32_t off, f;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(off) : "v"(acc30), "v"(0x3FF8u));
    asm("v_bfe_u32 %0, %1, 8, 11" : "=v"(f) : "v"(acc30));
    e = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(tab) + off);
    w = nco_fine_pair(f);
}

// The fine-sine operands of two phases (the even / odd pair of one 64-bit phase sum) as one
// packed multiply-add; v_pk_fma_f32 keeps denormal operands like v_fma_f32 does (ddc_create
// has the device evaluate both forms for every fine phase).
typedef float nco_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nco_fine_pair2(uint32_t f0, uint32_t f1, uint32_t &w0, uint32_t &w1)
{
    const nco_v2f x = { __builtin_bit_cast(float, f0), __builtin_bit_cast(float, f1) };
    const float d = __builtin_bit_cast(float, DDC_SF_FMA_D_BITS);
    const nco_v2f r = __builtin_elementwise_fma(x, (nco_v2f){ DDC_SF_FMA_C, DDC_SF_FMA_C }, (nco_v2f){ d, d });
    const float r0 = r.x, r1 = r.y; // (a bit_cast of a vector element reads element 0 whichever is named)
    w0 = __builtin_bit_cast(uint32_t, r0);
    w1 = __builtin_bit_cast(uint32_t, r1);
}

// nco_fetch for both phases of a pair (low dword: even sample, high dword: odd sample).
__device__ __forceinline__ void nco_fetch_pair(const uint2 *tab, unsigned long long p, uint2 &e0, uint2 &e1, uint32_t &w0,
                                               uint32_t &w1)
{
    const uint32_t a0 = (uint32_t)p, a1 = (uint32_t)(p >> 32);
    uint32_t off0, off1, f0, f1;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(off0) : "v"(a0), "v"(0x3FF8u));
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(off1) : "v"(a1), "v"(0x3FF8u));
    asm("v_bfe_u32 %0, %1, 8, 11" : "=v"(f0) : "v"(a0));
    asm("v_bfe_u32 %0, %1, 8, 11" : "=v"(f1) : "v"(a1));
    e0 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(tab) + off0);
    e1 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(tab) + off1);
    nco_fine_pair2(f0, f1, w0, w1);
}

// NCO second half + mixer: 2*(cc*sf + sc*8191 + 4096) has the 12-bit mixer
// operand (14-bit output >> 2) in bits 31..16.  Returns the two 24-bit mixer
// products adc * nco12; the CIC input is x = wrap23(product) >> 8.
__device__ __forceinline__ void nco_finish(const uint2 e, uint32_t w, int adc, int &pi, int &pq)
{
    const int s = dot2_sconst(e.x, w, 8192);
    const int c = dot2_sconst(e.y, w, 8192);
    pi = __mul24(adc, s >> 16);
    pq = __mul24(adc, c >> 16);
}

__device__ __forceinline__ void nco_mix(const uint2 *tab, uint32_t acc32, int adc, int &pi, int &pq)
{
    uint2 e;
    uint32_t w;
    nco_fetch(tab, acc32, e, w);
    nco_finish(e, w, adc, pi, pq);
}

// 16 bytes per lane from global memory straight into LDS (no VGPR round trip):
// lane l of the wave lands at lds_wave_base + 16 l.
__device__ __forceinline__ void lds_dma16(const uint8_t *g, uint4 *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)g,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Two mixer products -> the packed pair {2 x_lo, 2 x_hi} of 16-bit halves.
// Bits 23..8 of a product are x except that the one product that overflows 23
// bits (2^22 = -2048 * -2048) must wrap to -2^14; doubling the 16-bit field in
// 16-bit arithmetic drops exactly that bit, and 2 x always fits.  Everything
// downstream of the mixer therefore carries twice the CIC input; the last
// kernel of the CIC takes its output bits one place higher.
__device__ __forceinline__ unsigned pack_x2(int p_lo, int p_hi)
{
    const unsigned y = __builtin_amdgcn_perm((unsigned)p_hi, (unsigned)p_lo, 0x06050201u);
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %1" : "=v"(r) : "v"(y));
    return r;
}

// eight 12-bit samples from three dwords
__device__ __forceinline__ void unpack8(uint32_t d0, uint32_t d1, uint32_t d2, int (&a)[8])
{
    a[0] = sbfe_c<0, 12>((int)d0);
    a[1] = sbfe_c<12, 12>((int)d0);
    a[2] = sbfe_c<0, 12>((int)__builtin_amdgcn_alignbit(d1, d0, 24));
    a[3] = sbfe_c<4, 12>((int)d1);
    a[4] = sbfe_c<16, 12>((int)d1);
    a[5] = sbfe_c<0, 12>((int)__builtin_amdgcn_alignbit(d2, d1, 28));
    a[6] = sbfe_c<8, 12>((int)d2);
    a[7] = (int)d2 >> 20;
}

// acc += v for a 64-bit sum and a 32-bit signed addend as one instruction (the
// compiler's choice is a shift for the sign word and a 64-bit add).
__device__ __forceinline__ void add_sext(long long &acc, int v)
{
    unsigned long long carry;
    asm("v_mad_i64_i32 %0, %1, %2, 1, %0" : "+v"(acc), "=s"(carry) : "v"(v));
}

// Running sums over the chunk steps of one window.  T[i][q] ends as
// sum_a C(a'+q, q) mu_i(a); the widths are the smallest that provably hold the
// value for doubled 15-bit inputs (cascade_widths() in tools/cic_math.py).
struct Cascade {
    int t00, t01, t02, t10, t11, t20;
    long long t03, t04, t12, t13, t21, t22, t30, t31, t40;
    __device__ __forceinline__ void clear()
    {
        t00 = t01 = t02 = t10 = t11 = t20 = 0;
        t03 = t04 = t12 = t13 = t21 = t22 = t30 = t31 = t40 = 0;
    }
    __device__ __forceinline__ void step(const int (&mu)[5])
    {
        t00 += mu[0]; t01 += t00; t02 += t01; add_sext(t03, t02); t04 += t03;
        t10 += mu[1]; t11 += t10; add_sext(t12, t11); t13 += t12;
        t20 += mu[2]; add_sext(t21, t20); t22 += t21;
        add_sext(t30, mu[3]); t31 += t30;
        add_sext(t40, mu[4]);
    }
    // Adds what a step with local moments mu, taken before the sixteen steps this
    // cascade has run, would have left behind: C(16 + q, q) * mu[i] in T[i][q].
    __device__ __forceinline__ void inject_first(const int (&mu)[5])
    {
        t00 += mu[0]; t01 += 17 * mu[0]; t02 += 153 * mu[0]; t03 += 969ll * mu[0]; t04 += 4845ll * mu[0];
        t10 += mu[1]; t11 += 17 * mu[1]; t12 += 153ll * mu[1]; t13 += 969ll * mu[1];
        t20 += mu[2]; t21 += 17ll * mu[2]; t22 += 153ll * mu[2];
        t30 += mu[3]; t31 += 17ll * mu[3];
        t40 += mu[4];
    }
    // Adds local moments mu to the last step taken: every T[i][q] of a step sees mu[i] once.
    __device__ __forceinline__ void inject_last(const int (&mu)[5])
    {
        t00 += mu[0]; t01 += mu[0]; t02 += mu[0]; t03 += mu[0]; t04 += mu[0];
        t10 += mu[1]; t11 += mu[1]; t12 += mu[1]; t13 += mu[1];
        t20 += mu[2]; t21 += mu[2]; t22 += mu[2];
        t30 += mu[3]; t31 += mu[3];
        t40 += mu[4];
    }
    // B[e] = sum_{i<=e} sum_{q<=e-i} GAM[e-i][q] * T[i][q]   (mod 2^64)
    __device__ __forceinline__ void moments(unsigned long long (&B)[5]) const
    {
        const long long T[5][5] = {
            { t00, t01, t02, t03, t04 }, { t10, t11, t12, t13, 0 }, { t20, t21, t22, 0, 0 },
            { t30, t31, 0, 0, 0 },       { t40, 0, 0, 0, 0 } };
#pragma unroll
        for (int e = 0; e < 5; e++) {
            unsigned long long acc = 0;
#pragma unroll
            for (int i = 0; i <= e; i++)
#pragma unroll
                for (int q = 0; q <= e - i; q++)
                    acc += (unsigned long long)GAM[e - i][q] * (unsigned long long)T[i][q];
            B[e] = acc;
        }
    }
};

// v_dot2_i32_i16 with a zero addend: starts an accumulator without a move to clear it
__device__ __forceinline__ int dot2_first(unsigned a, unsigned w)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(w));
    return r;
}

template <int P>
__device__ __forceinline__ void pair_acc(unsigned pk_i, unsigned pk_q, int (&mi)[5], int (&mq)[5])
{
    constexpr unsigned w0 = pack_w(1, 1);
    constexpr unsigned w1 = pack_w(psi_w(1, 2 * P), psi_w(1, 2 * P + 1));
    constexpr unsigned w2 = pack_w(psi_w(2, 2 * P), psi_w(2, 2 * P + 1));
    constexpr unsigned w3 = pack_w(psi_w(3, 2 * P), psi_w(3, 2 * P + 1));
    constexpr unsigned w4 = pack_w(psi_w(4, 2 * P), psi_w(4, 2 * P + 1));
    if (P == 0) { // first pair of a chunk
        mi[0] = dot2_first(pk_i, w0); mq[0] = dot2_first(pk_q, w0);
        mi[1] = dot2_first(pk_i, w1); mq[1] = dot2_first(pk_q, w1);
        mi[2] = dot2_first(pk_i, w2); mq[2] = dot2_first(pk_q, w2);
        mi[3] = dot2_first(pk_i, w3); mq[3] = dot2_first(pk_q, w3);
        mi[4] = dot2_first(pk_i, w4); mq[4] = dot2_first(pk_q, w4);
        return;
    }
    mi[0] = dot2(pk_i, w0, mi[0]); mq[0] = dot2(pk_q, w0, mq[0]);
    mi[1] = dot2(pk_i, w1, mi[1]); mq[1] = dot2(pk_q, w1, mq[1]);
    mi[2] = dot2(pk_i, w2, mi[2]); mq[2] = dot2(pk_q, w2, mq[2]);
    mi[3] = dot2(pk_i, w3, mi[3]); mq[3] = dot2(pk_q, w3, mq[3]);
    mi[4] = dot2(pk_i, w4, mi[4]); mq[4] = dot2(pk_q, w4, mq[4]);
}

// The five samples s = 27..31 of a chunk (pairs 13 (high half only), 14, 15),
// added with sign SGN: the part of a natural chunk that belongs to the next
// window.
template <int SGN>
__device__ __forceinline__ void tail5_acc(const unsigned (&pk)[3], int (&m)[5])
{
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const unsigned wa = pack_w(0, SGN * psi_w(i, 27));
        const unsigned wb = pack_w(SGN * psi_w(i, 28), SGN * psi_w(i, 29));
        const unsigned wc = pack_w(SGN * psi_w(i, 30), SGN * psi_w(i, 31));
        m[i] = dot2(pk[0], wa, m[i]);
        m[i] = dot2(pk[1], wb, m[i]);
        m[i] = dot2(pk[2], wc, m[i]);
    }
}

// The phases of two neighbouring samples advance together as one 64-bit sum
// (even sample in the low dword, odd in the high): a carry out of the low dword
// lands in bit 0 of the high one, below the phase field (bits 29..8), and a block
// of 512 samples produces fewer than 2^7 of them.
__device__ __forceinline__ unsigned long long pair_add(unsigned long long a, unsigned long long b)
{
    unsigned long long r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int G>
__device__ __forceinline__ void group8(const uint2 *tab, uint32_t d0, uint32_t d1, uint32_t d2,
                                       unsigned long long &accp, unsigned long long phi2, int (&mi)[5], int (&mq)[5],
                                       unsigned (&last_i)[3], unsigned (&last_q)[3])
{
    int a[8];
    unpack8(d0, d1, d2, a);
    uint2 e[8];
    uint32_t w[8];
    unsigned long long p = accp;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nco_fetch_pair(tab, p, e[2 * k], e[2 * k + 1], w[2 * k], w[2 * k + 1]);
        p = pair_add(p, phi2);
    }
    accp = p;
    unsigned pi[4], pq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int pi0, pq0, pi1, pq1;
        nco_finish(e[2 * k], w[2 * k], a[2 * k], pi0, pq0);
        nco_finish(e[2 * k + 1], w[2 * k + 1], a[2 * k + 1], pi1, pq1);
        pi[k] = pack_x2(pi0, pi1);
        pq[k] = pack_x2(pq0, pq1);
    }
    pair_acc<4 * G + 0>(pi[0], pq[0], mi, mq);
    pair_acc<4 * G + 1>(pi[1], pq[1], mi, mq);
    pair_acc<4 * G + 2>(pi[2], pq[2], mi, mq);
    pair_acc<4 * G + 3>(pi[3], pq[3], mi, mq);
    if (G == 3) {
        last_i[0] = pi[1]; last_i[1] = pi[2]; last_i[2] = pi[3];
        last_q[0] = pq[1]; last_q[1] = pq[2]; last_q[2] = pq[3];
    }
}

// COEF[j][e] * b mod 2^64 with the constant split into sign, an odd factor below
// 2^32 and a power of two: a 32 x 64 bit product and a shift instead of a full
// 64 x 64 bit multiply (the small odd factors fold into shifts and adds).
__host__ __device__ constexpr unsigned long long coef_abs(int j, int e)
{
    return COEF[j][e] < 0 ? (unsigned long long)(-COEF[j][e]) : (unsigned long long)COEF[j][e];
}
__host__ __device__ constexpr int coef_tz(int j, int e)
{
    unsigned long long m = coef_abs(j, e);
    int k = 0;
    while (m && !(m & 1)) { m >>= 1; k++; }
    return k;
}
constexpr bool coef_odd_parts_fit()
{
    for (int j = 0; j < 5; j++)
        for (int e = 0; e < 5; e++)
            if ((coef_abs(j, e) >> coef_tz(j, e)) >> 32) return false;
    return true;
}
static_assert(coef_odd_parts_fit(), "odd part of a CIC combination constant exceeds 32 bits");

__device__ __forceinline__ unsigned long long coef_mul(int j, int e, unsigned long long b)
{
    const unsigned long long mag = coef_abs(j, e);
    if (mag == 0) return 0;
    const int tz = coef_tz(j, e);
    const uint32_t odd = (uint32_t)(mag >> tz);
    const unsigned long long lo = (unsigned long long)odd * (uint32_t)b;
    const unsigned long long hi = (unsigned long long)(odd * (uint32_t)(b >> 32)) << 32;
    const unsigned long long t = (lo + hi) << tz;
    return COEF[j][e] < 0 ? 0ull - t : t;
}

// convergent rounding as the generated HDL codes it, on the low `bits` bits
__device__ __forceinline__ int cround32(uint32_t v, int s, int bits)
{
    const int sh = 32 - bits;
    const int t = (int)(v << sh) >> sh;
    const uint32_t u = (uint32_t)t + ((1u << (s - 1)) - 1u) + (((uint32_t)t >> s) & 1u);
    return ((int)(u << sh) >> sh) >> s;
}

__constant__ int16_t c_rx_ciccomp_h[65];
__constant__ int16_t c_rx_hilb_c[256];
// rx_ciccomp taps as the kernel multiplies them: pair m = {h[64-2m] | h[63-2m] << 16}, m = 0..31, and {h[0] | 0}
__constant__ uint32_t c_ciccomp_pairs[33];

constexpr int kTailTile = 1024; // outputs per workgroup of the two 48 kHz kernels (256 threads x 4)

__constant__ uint32_t c_hilb_pairs[128]; // {c[255-2m] | c[254-2m] << 16}
__constant__ uint32_t c_hilb_pairs_pad[144]; // the same at [3 + m], zeros around: the sliding window needs no bounds
__constant__ uint32_t c_hilb_odd[8];     // bit j of the 256-bit mask: c[255-j] & 1
__constant__ uint32_t c_hilb_bit1[8];    // bit j: (c[255-j] >> 1) & 1

/*
 * The two 48 kHz stages as functions over LDS arrays, shared by the stand-alone
 * kernels further down and by ddc_rx_tail (rx_tile_finish).
 *
 * rx_ciccomp: e[n] = wrap16(cround(sum_{i=0..64} h[i] * c[2n-i], 15)) for I and Q
 * (FPGA/rx_ciccomp.vhd:616; only the low 31 bits of the sum matter, so the sums
 * run mod 2^32).  The rx_cic samples are split into an I and a Q array of int16,
 * so that a dword holds two consecutive samples of one branch and one v_dot2 does
 * two taps; a thread computes four consecutive outputs from a sliding window of
 * 36 pairs (nine 16-byte LDS reads per branch, contiguous across the lanes):
 * outputs 4 q + k, k = 0..3, use pairs (4 q + k) + m, m = 0..32; pi / pq point at
 * quad q of the two arrays.  Returns {I | Q << 16} per output.
 */
__device__ __forceinline__ void ciccomp_quad(const uint4 *pi, const uint4 *pq, uint32_t (&packed)[4])
{
    int ai[4] = {0, 0, 0, 0}, aq[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const uint4 vi = pi[q], vq = pq[q];
        const uint32_t wi[4] = { vi.x, vi.y, vi.z, vi.w }, wq[4] = { vq.x, vq.y, vq.z, vq.w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int m = 4 * q + e - k;
                if (m >= 0 && m < 33) {
                    ai[k] = dot2(wi[e], c_ciccomp_pairs[m], ai[k]);
                    aq[k] = dot2(wq[e], c_ciccomp_pairs[m], aq[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int16_t ei = (int16_t)cround32((uint32_t)ai[k], 15, 31), eq = (int16_t)cround32((uint32_t)aq[k], 15, 31);
        packed[k] = (uint32_t)(uint16_t)ei | ((uint32_t)(uint16_t)eq << 16);
    }
}

/*
 * rx_hilb: wrap16(cround(wrap30(sum_k cround(c[k]*x[n-k], 1)), 14)) (FPGA/rx_hilb.vhd:902-935).
 *
 * The per-product rounding is taken out of the tap loop.  With p = c*x,
 * cround(p, 1) = floor(p / 2) + [p = 3 mod 4], so
 *     sum_k cround(p_k, 1) = (S - N1) / 2 + N3,
 * S = sum p_k (a plain dot product: v_dot2 on sample pairs), N1 = number of odd
 * products, N3 = number of products = 3 mod 4.  N1 and N3 depend only on the two
 * low bits of taps and samples: with the samples' bit 0 / bit 1 as bit planes and
 * the taps' as constant masks they are population counts, 8 dwords per output.
 * Only the low 30 bits of the sum matter, so everything runs mod 2^32.
 *
 * A tile is 1024 consecutive outputs of one channel computed from the 1279
 * samples they reach (tile sample u = the sample 255 - u before the tile's first
 * output), staged in LDS twice (xs[0]: pairs (2m, 2m+1); xs[1]: pairs (2m+1, 2m+2),
 * so that odd outputs also read aligned pairs) plus the two bit planes.  A thread
 * computes four outputs of one parity, two samples apart, from a sliding window of
 * 131 pairs (33 16-byte LDS reads, contiguous across the lanes): thread (par, g)
 * has outputs o_k = 8 g + par + 2 k, k = 0..3, output k uses pairs (4 g + k) + m,
 * m = 0..127, of array par.
 */
constexpr int kHilbIn = kTailTile + 256;         // 1280 staged samples (1279 used)

// taps: c_hilb_pairs_pad copied to LDS (hilbert_stage_taps), so that the tap loop waits for no scalar load
__device__ __forceinline__ void hilbert_stage_taps(uint32_t *taps, uint32_t t)
{
    if (t < 144) taps[t] = c_hilb_pairs_pad[t];
}

__device__ __forceinline__ void hilbert_quad_sums(const uint4 *src, const uint32_t *taps, int (&S)[4])
{
    S[0] = S[1] = S[2] = S[3] = 0;
    // quad q holds pairs 4 q + e; output k multiplies pair j with tap pair j - k, i.e. taps
    // 4 q - 3 .. 4 q + 3 (zero outside 0..127) = padded entries 4 q .. 4 q + 6: two 16-byte
    // reads at an address all lanes share
    const uint4 *tp = reinterpret_cast<const uint4 *>(taps);
#pragma unroll 3
    for (int q = 0; q < 33; q++) {
        const uint4 v = src[q];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        const uint4 c0 = tp[q], c1 = tp[q + 1];
        const uint32_t c[7] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z };
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int k = 0; k < 4; k++) S[k] = dot2(w[e], c[3 + e - k], S[k]);
        }
    }
}

// The four outputs o0, o0 + 2, o0 + 4, o0 + 6 of a thread (o0 = 8 g + par) from their plain dot
// products S and the bit planes: their 256-bit windows start in the same dword (o0 mod 32 is at most
// 25), so nine dwords of each plane serve all four.
__device__ __forceinline__ void hilbert_round4(const int (&S)[4], uint32_t o0, const uint32_t *p0, const uint32_t *p1,
                                               int16_t (&y)[4])
{
    const uint32_t w0 = o0 >> 5, sh0 = o0 & 31u;
    uint32_t a[9], b[9];
#pragma unroll
    for (int d = 0; d < 9; d++) { a[d] = p0[w0 + d]; b[d] = p1[w0 + d]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t sh = sh0 + 2u * k;
        uint32_t n1 = 0, n3 = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const uint32_t a0 = __builtin_amdgcn_alignbit(a[d + 1], a[d], sh);
            const uint32_t a1 = __builtin_amdgcn_alignbit(b[d + 1], b[d], sh);
            const uint32_t both = a0 & c_hilb_odd[d];
            n1 += __popc(both);
            n3 += __popc((a1 ^ c_hilb_bit1[d]) & both);
        }
        const uint32_t acc = (((uint32_t)S[k] - n1) >> 1) + n3;
        y[k] = (int16_t)cround32(acc, 14, 30);
    }
}

// Bit planes of the staged samples sa[0 .. kHilbIn): all 256 threads, after the samples are in LDS.
__device__ __forceinline__ void hilbert_planes(const int16_t *sa, uint32_t *p0, uint32_t *p1, uint32_t t)
{
#pragma unroll
    for (uint32_t u = t; u < (uint32_t)kHilbIn; u += 256) {
        const int v = sa[u];
        const unsigned long long b0 = __ballot(v & 1), b1 = __ballot(v & 2);
        if ((t & 63u) == 0) {
            const uint32_t w = u >> 5;
            p0[w] = (uint32_t)b0; p0[w + 1] = (uint32_t)(b0 >> 32);
            p1[w] = (uint32_t)b1; p1[w + 1] = (uint32_t)(b1 >> 32);
        }
    }
    if (t == 0) { p0[kHilbIn / 32] = 0; p0[kHilbIn / 32 + 1] = 0; p1[kHilbIn / 32] = 0; p1[kHilbIn / 32 + 1] = 0; }
}

/*
 * The two-kernel RX call (ddc_rx_front_cic, ddc_rx_tail).  Between the kernels:
 *
 *   cic       [channel][kCicHist + n_blk][2] int16: rx_cic outputs of this call at
 *             kCicHist + m (the kCicHist slots in front are unused here); a
 *             workgroup of the first kernel writes the 251 outputs whose five windows
 *             it owns itself
 *   edge      [channel][n_wg][kEdgeRecs][10] u64: the moments of a workgroup's first
 *             four and last five windows, from which the second kernel makes up the five
 *             outputs at every workgroup boundary
 *
 * Tile j of a channel (frames 1024 j .. 1024 j + 1023 of the call) reads the rx_cic
 * outputs 2048 j - 576 .. 2048 j + 2047: its own 2048, 64 the compensator reaches back,
 * 512 more to recompute the 256 compensator outputs the Hilbert filter reaches back, so
 * no tile waits for another.  Tile 0 takes what lies before the call from the saved
 * histories.  Inside the second kernel, with several tiles per channel:
 *
 *   comp_tail [channel][kCompHist][2]: the last 256 compensator outputs of the call,
 *             stored write-through (sc1) by the last tile's workgroup
 *   chan_cnt  [channel]: arrival counter, zero between calls; every tile's workgroup
 *             waits for its stores and adds one, the one whose add completes the count
 *             runs an agent-scope acquire and saves the channel's new histories, mixer
 *             carries and NCO phase.  No workgroup ever waits for another one.
 */
constexpr int kEdgeRecs = 9;
constexpr int kWgPerTile = 2 * kTailTile / kFrontThreads;  // 8 workgroups of 256 windows per 1024 frames

struct RxFused {
    int16_t *cic;
    unsigned long long *edge;
    uint32_t *comp_tail;
    uint32_t *chan_cnt;
    unsigned long long *hist_mom;    // [channel][kMomHist][10]
    int16_t *hist_cic, *hist_comp;   // [channel][kCicHist][2], [channel][kCompHist][2]
    uint32_t *carry_saved, *carry_work, *acc0;
    const uint32_t *phi32v;
    int16_t *spec_iq, *voice_iq;
    uint32_t n_wg, n_tiles, samples_lo;
    uint32_t out_pitch; // frames per channel row of spec_iq / voice_iq (at least the call's own frame count)
};

// byte offsets into the 32 KB line buffer while a tile runs
constexpr int kTailIn = 2 * (kTailTile + 256) + 64;        // rx_cic samples per branch a tile stages
constexpr uint32_t kLdsSi = 0, kLdsSq = 5376;              // int16[kTailIn + 16] each
constexpr uint32_t kLdsXs = 10752;                         // uint32[2][kHilbIn / 2]
constexpr uint32_t kLdsP0 = 15872, kLdsP1 = 16128;         // uint32[kHilbIn / 32 + 2] each
constexpr uint32_t kLdsCq = 16384;                         // int16[kHilbIn]: compensator Q of frames n0 - 256 ..
constexpr uint32_t kLdsTaps = 19456;                       // uint32[144]: the padded Hilbert tap pairs
constexpr uint32_t kLdsFlag = 20480;                       // uint32[4]
static_assert((kTailIn + 16) * 2 <= (int)(kLdsSq - kLdsSi) && kLdsXs - kLdsSq == kLdsSq - kLdsSi, "tile LDS layout");
static_assert(kLdsXs + 4 * kHilbIn <= kLdsP0 && kLdsCq + 2 * kHilbIn <= kLdsTaps && kLdsTaps + 4 * 144 <= kLdsFlag && kLdsFlag >= 10 * 256 * 8, "tile LDS layout");

// Write-through stores (global_store ... sc1): visible to every other workgroup of the launch
// once the storing wave's s_waitcnt vmcnt(0) has passed, with no L2 write-back.
__device__ __forceinline__ void store_wt(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One relaxed add to an arrival counter by one lane, after every wave of the workgroup
// has waited for its write-through stores and passed a barrier; `true` when this add
// completed the count (the counter is then put back to zero for the next call).
__device__ __forceinline__ bool count_in(uint32_t *cnt, uint32_t need)
{
    const uint32_t old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u != need) return false;
    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// The five rx_cic outputs at the front of workgroup kk (m = 256 kk + r, r = 0..4) reach
// into the previous workgroup's windows: made from the edge records (or, for kk = 0, the
// saved moments of the previous call).
__device__ __forceinline__ uint32_t cic_edge_output(const unsigned long long *edge_ch, const unsigned long long *hist_ch,
                                                    uint32_t kk, uint32_t r)
{
    unsigned long long acc[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int p = (int)r - 1 - j;   // window m - 1 - j relative to the workgroup's first
        const unsigned long long *rec;
        if (p >= 0) rec = edge_ch + ((size_t)kk * kEdgeRecs + p) * 10;
        else if (kk > 0) rec = edge_ch + ((size_t)(kk - 1) * kEdgeRecs + 9 + p) * 10;
        else rec = hist_ch + (5 + p) * 10;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            acc[0] += coef_mul(j, e, rec[e]);
            acc[1] += coef_mul(j, e, rec[5 + e]);
        }
    }
    const uint32_t yi = (uint32_t)(uint16_t)(int16_t)((long long)(acc[0] << 3) >> 48);
    const uint32_t yq = (uint32_t)(uint16_t)(int16_t)((long long)(acc[1] << 3) >> 48);
    return yi | (yq << 16);
}

// rx_ciccomp, rx_hilb and data_delay for frames 1024 j .. of channel ch; all 256 threads.
__device__ __forceinline__ void rx_tile_finish(const RxFused &fa, uint32_t ch, uint32_t j, uint32_t n_blk, char *lds,
                                               uint32_t t)
{
    const uint32_t nf = n_blk / 2, n0 = j * (uint32_t)kTailTile;
    const uint32_t n1 = n0 + kTailTile < nf ? n0 + kTailTile : nf;
    const bool halo = j != 0;
    int16_t *si = reinterpret_cast<int16_t *>(lds + kLdsSi), *sq = reinterpret_cast<int16_t *>(lds + kLdsSq);
    uint32_t *xa = reinterpret_cast<uint32_t *>(lds + kLdsXs);
    int16_t *sa = reinterpret_cast<int16_t *>(xa), *sb = sa + kHilbIn;
    uint32_t *p0 = reinterpret_cast<uint32_t *>(lds + kLdsP0), *p1 = reinterpret_cast<uint32_t *>(lds + kLdsP1);
    int16_t *cq = reinterpret_cast<int16_t *>(lds + kLdsCq);

    // rx_cic samples c_lo .. 2 n1 - 1 of the call (negative: the saved history)
    const int c_lo = 2 * (int)n0 - kCicHist - (halo ? 512 : 0);
    const uint32_t count = (uint32_t)(2 * (int)n1 - c_lo);
    uint32_t *cic32 = reinterpret_cast<uint32_t *>(fa.cic) + (size_t)ch * (kCicHist + n_blk) + kCicHist;
    const uint32_t *hc32 = reinterpret_cast<const uint32_t *>(fa.hist_cic) + (size_t)ch * kCicHist;
    // four samples per 16-byte load, all loads of a thread in flight together
    {
        constexpr int kQuads = (kTailIn + 16) / 4;
        uint4 v[(kQuads + 255) / 256];
#pragma unroll
        for (int i = 0; i < (kQuads + 255) / 256; i++) {
            const uint32_t s = 4u * (t + 256u * i);
            const int c = c_lo + (int)s;
            const uint32_t *src = c < 0 ? hc32 + (kCicHist + c) : cic32 + c;
            if (s >= count) src = cic32; // past the tile: any readable address, the value is dropped
            v[i] = *reinterpret_cast<const uint4 *>(src);
        }
#pragma unroll
        for (int i = 0; i < (kQuads + 255) / 256; i++) {
            const uint32_t s = 4u * (t + 256u * i);
            const int c = c_lo + (int)s;
            if (s >= (uint32_t)kTailIn + 16) continue;
            if (s >= count) v[i] = make_uint4(0, 0, 0, 0);
            // outputs 0..4 of every workgroup of the first kernel are made up just below
            const uint32_t r = c >= 0 ? (uint32_t)c & 255u : 8u;
            if (r == 0) continue;
            const uint32_t i01 = __builtin_amdgcn_perm(v[i].y, v[i].x, 0x05040100u), i23 = __builtin_amdgcn_perm(v[i].w, v[i].z, 0x05040100u);
            const uint32_t q01 = __builtin_amdgcn_perm(v[i].y, v[i].x, 0x07060302u), q23 = __builtin_amdgcn_perm(v[i].w, v[i].z, 0x07060302u);
            if (r == 4) {
                si[s + 1] = (int16_t)(i01 >> 16);
                sq[s + 1] = (int16_t)(q01 >> 16);
                *reinterpret_cast<uint32_t *>(si + s + 2) = i23;
                *reinterpret_cast<uint32_t *>(sq + s + 2) = q23;
            } else {
                *reinterpret_cast<uint2 *>(si + s) = make_uint2(i01, i23);
                *reinterpret_cast<uint2 *>(sq + s) = make_uint2(q01, q23);
            }
        }
    }
    hilbert_stage_taps(reinterpret_cast<uint32_t *>(lds + kLdsTaps), t);
    {
        const uint32_t k_first = halo ? j * kWgPerTile - 2 : 0;
        const uint32_t k_end = (j + 1) * kWgPerTile < fa.n_wg ? (j + 1) * kWgPerTile : fa.n_wg;
        const uint32_t kk = k_first + t / 5, r = t % 5;
        if (t < 5 * (kWgPerTile + 2) && kk < k_end) {
            const uint32_t v = cic_edge_output(fa.edge + (size_t)ch * fa.n_wg * kEdgeRecs * 10,
                                               fa.hist_mom + (size_t)ch * kMomHist * 10, kk, r);
            const uint32_t m = kk * 256u + r;
            cic32[m] = v; // for the debug tap; a neighbouring tile may write the same value
            const uint32_t s = (uint32_t)((int)m - c_lo);
            si[s] = (int16_t)(v & 0xFFFFu);
            sq[s] = (int16_t)(v >> 16);
        }
    }
    // tile 0: the 256 compensator outputs before the call come from the saved history
    if (!halo) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(fa.hist_comp)[(size_t)ch * kCompHist + t];
        if (t >= 1) sa[t - 1] = (int16_t)(v & 0xFFFFu);
        if (t >= 2) sb[t - 2] = (int16_t)(v & 0xFFFFu);
        cq[t] = (int16_t)(v >> 16);
    }
    __syncthreads();

    // compensator outputs of frames f_lo .. n1 - 1
    const uint32_t f_lo = n0 - (halo ? 256u : 0u), n_out = n1 - f_lo;
    uint32_t *spec32 = reinterpret_cast<uint32_t *>(fa.spec_iq) + (size_t)ch * fa.out_pitch;
#pragma unroll 1
    for (uint32_t o = 4 * t; o < n_out; o += 1024) {
        uint32_t packed[4];
        ciccomp_quad(reinterpret_cast<const uint4 *>(si) + o / 4, reinterpret_cast<const uint4 *>(sq) + o / 4, packed);
        const uint32_t f = f_lo + o;
        const int u0 = (int)(f + 255u) - (int)n0; // Hilbert tile sample of frame f
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = u0 + k;
            if (u >= 0) sa[u] = (int16_t)(packed[k] & 0xFFFFu);
            if (u >= 1) sb[u - 1] = (int16_t)(packed[k] & 0xFFFFu);
            cq[u + 1] = (int16_t)(packed[k] >> 16);
        }
        if (fa.n_tiles > 1 && f + kCompHist >= nf) { // the last 256 outputs of the call: what rx_hilb reaches back to next time
#pragma unroll
            for (int k = 0; k < 4; k++) store_wt(fa.comp_tail + (size_t)ch * kCompHist + (f + k + kCompHist - nf), packed[k]);
        }
        if (f >= n0) { // frame counts are multiples of 128: a quad is never cut
            if (((uintptr_t)(spec32 + f) & 15u) == 0) *reinterpret_cast<uint4 *>(spec32 + f) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            else { spec32[f] = packed[0]; spec32[f + 1] = packed[1]; spec32[f + 2] = packed[2]; spec32[f + 3] = packed[3]; }
        }
    }
    __syncthreads();
    {
        // samples beyond the last frame of a short tile read as zero
        const uint32_t n_valid = 255u + (n1 - n0);
        for (uint32_t u = n_valid + t; u < (uint32_t)kHilbIn; u += 256) { sa[u] = 0; sb[u - 1] = 0; }
        if (t == 0) sb[kHilbIn - 1] = 0;
    }
    __syncthreads();
    hilbert_planes(sa, p0, p1, t);
    __syncthreads();

    const uint32_t par = t >> 7, g = t & 127u;
    int S[4];
    hilbert_quad_sums(reinterpret_cast<const uint4 *>(xa) + par * (kHilbIn / 8) + g,
                      reinterpret_cast<const uint32_t *>(lds + kLdsTaps), S);
    uint32_t *voice32 = reinterpret_cast<uint32_t *>(fa.voice_iq) + (size_t)ch * fa.out_pitch;
    int16_t vi[4];
    hilbert_round4(S, 8 * g + par, p0, p1, vi);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = 8 * g + par + 2 * k, n = n0 + o;
        if (n >= n1) continue;
        const int16_t vq = cq[o + 126]; // frame n - 130 (FPGA/data_delay.v, delay_length = 130)
        voice32[n] = (uint32_t)(uint16_t)vi[k] | ((uint32_t)(uint16_t)vq << 16);
    }
}

// What one channel carries into the next call, beside the compensator outputs: read from what the
// first kernel left (any time in the second kernel), stored after every tile of the channel is done.
struct ChannelCarry {
    unsigned long long mom; // t < 50: the moments of the call's last five windows
    uint32_t cic;           // t < 64: its last 64 rx_cic outputs
    uint32_t carry;         // t < 6: mixer outputs of its last six samples
    uint32_t phase;         // t = 0: NCO phase after the call
};
__device__ __forceinline__ void rx_channel_load(const RxFused &fa, uint32_t ch, uint32_t n_blk, uint32_t t, ChannelCarry &c)
{
    c.mom = 0; c.cic = 0; c.carry = 0; c.phase = 0;
    if (t < kMomHist * 10) c.mom = fa.edge[(((size_t)ch * fa.n_wg + fa.n_wg - 1) * kEdgeRecs + 4) * 10 + t];
    if (t < kCicHist)
        c.cic = reinterpret_cast<const uint32_t *>(fa.cic)[(size_t)ch * (kCicHist + n_blk) + kCicHist + n_blk - kCicHist + t];
    if (t < 6) c.carry = fa.carry_work[ch * 6 + t];
    if (t == 0) c.phase = fa.acc0[ch] + fa.samples_lo * fa.phi32v[ch];
}
// comp: entry t of the new compensator history (frame nf - 256 + t of the call, or of what came before it)
__device__ __forceinline__ void rx_channel_store(const RxFused &fa, uint32_t ch, uint32_t t, const ChannelCarry &c, uint32_t comp)
{
    if (t < kMomHist * 10) fa.hist_mom[(size_t)ch * kMomHist * 10 + t] = c.mom;
    if (t < kCicHist) reinterpret_cast<uint32_t *>(fa.hist_cic)[(size_t)ch * kCicHist + t] = c.cic;
    reinterpret_cast<uint32_t *>(fa.hist_comp)[(size_t)ch * kCompHist + t] = comp;
    if (t < 6) fa.carry_saved[ch * 6 + t] = c.carry;
    if (t == 0) fa.acc0[ch] = c.phase;
}

// After the sample loop of ddc_rx_front_cic: thread t holds the moments of window b = 256 k + t of
// channel ch.  rx_cic output m needs windows m-5 .. m-1 (FPGA/rx_cic.vhd:291-404 as the 5 x 5
// combination of tools/cic_math.py): every thread applies the five rows of COEF to its own
// moments, P[j] = sum_e COEF[j][e] B[e] = what its window adds to output b + 1 + j, the P's go
// through LDS, and thread t sums the five that make output b (threads 0..4 lack windows of the
// workgroup before: their outputs are made up from the edge records by ddc_rx_tail).
__device__ __forceinline__ void rx_cic_epilogue(const RxFused &fa, uint32_t ch, uint32_t b, uint32_t n_blk,
                                                const unsigned long long (&Bi)[5], const unsigned long long (&Bq)[5],
                                                char *lds, uint32_t t)
{
    const uint32_t k = b >> 8;
    unsigned long long *px = reinterpret_cast<unsigned long long *>(lds); // [10][256]
    __syncthreads(); // the hand-over of the mixer outputs has been read
    px[t] = Bi[4];        // COEF[0] = (0, 0, 0, 0, 1)
    px[5 * 256 + t] = Bq[4];
#pragma unroll
    for (int j = 1; j < 5; j++) {
        unsigned long long pi = 0, pq = 0;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            pi += coef_mul(j, e, Bi[e]);
            pq += coef_mul(j, e, Bq[e]);
        }
        px[j * 256 + t] = pi;
        px[(5 + j) * 256 + t] = pq;
    }
    if (t < 4 || t >= 251) {
        unsigned long long *rec = fa.edge + (((size_t)ch * fa.n_wg + k) * kEdgeRecs + (t < 4 ? t : t - 247)) * 10;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            rec[e] = Bi[e];
            rec[5 + e] = Bq[e];
        }
    }
    __syncthreads();
    if (t >= 5) {
        unsigned long long yi = 0, yq = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            yi += px[j * 256 + t - 1 - j];
            yq += px[(5 + j) * 256 + t - 1 - j];
        }
        // bits 59..44 of the sum (FPGA/rx_cic.vhd:391); the moments are of 2 x, so one place up
        const uint32_t oi = (uint32_t)(uint16_t)(int16_t)((long long)(yi << 3) >> 48);
        const uint32_t oq = (uint32_t)(uint16_t)(int16_t)((long long)(yq << 3) >> 48);
        reinterpret_cast<uint32_t *>(fa.cic)[(size_t)ch * (kCicHist + n_blk) + kCicHist + b] = oi | (oq << 16);
    }
}

/*
 * Kernel 1.  One thread owns one natural block of 512 samples of one channel
 * (global block id g = channel * n_blk + block); a workgroup owns 256
 * consecutive ids.  A block is 768 bytes = six 128-byte lines.  The workgroup
 * stages one whole line per thread at a time through LDS (8 lanes fetch the
 * eight 16-byte pieces of one line, so HBM / L2 see every line exactly once and
 * as one full-line request), and every thread walks its 48-byte chunks across
 * the lines; the 32 / 16 bytes of a line that belong to the next chunk wait in
 * registers while the next line lands.  The load of the next line is issued
 * before the last chunk of the current one is computed, which hides its latency.
 *
 * LDS line buffer: row r (= thread r) holds its line at r * 128, piece p in slot
 * p ^ swz(r); the swizzle makes the 16-byte reads of 16 consecutive threads hit
 * 16 different bank groups.
 *
 *   mom  [channel][kMomHist + n_blk][2][5] u64 : window moments (of 2 x), I then
 *        Q; this kernel writes records kMomHist + block
 *   carry_in / carry_out [channel][6] : packed mixer outputs of the last six
 *        samples of the previous / this call (I pairs 13..15, Q pairs 13..15)
 */
__device__ __forceinline__ uint32_t line_swz(uint32_t row) { return ((row >> 1) & 3u) ^ (((row >> 3) & 1u) << 2); }

template <bool FUSED>
__device__ __forceinline__ void rx_front_body(
    const uint8_t *__restrict__ adc, size_t row_bytes, uint32_t n_blk, uint32_t total_blocks,
    const uint32_t *acc0, const uint32_t *__restrict__ phi32v,
    const uint32_t *carry_in, uint32_t *carry_out,
    unsigned long long *__restrict__ mom, const uint2 *__restrict__ nco_tab, const RxFused &fa)
{
    __shared__ uint2 tab[2048];
    __shared__ uint4 lines[8 * kFrontThreads];

    const uint32_t tid = threadIdx.x;
    const uint32_t g0 = blockIdx.x * kFrontThreads;
    const uint32_t lane = tid & 63u, wave_base = tid & ~63u;
    // the coarse NCO table: 16 KB straight into LDS, 1 KB per wave and load (waited for with the first line)
#pragma unroll
    for (int i = 0; i < 4; i++)
        lds_dma16(reinterpret_cast<const uint8_t *>(nco_tab) + (i * kFrontThreads + tid) * 16u,
                  reinterpret_cast<uint4 *>(tab) + i * kFrontThreads + wave_base);

    // staging sources: load j of a wave covers rows wave_base + 8 j .. + 7, lane l
    // fetching for row 8 j + (l >> 3) the piece that lives in slot l & 7
    const uint8_t *src[8];
    uint32_t ch, b;
    bool active = true;
    if (FUSED) {
        // the 256 windows of the workgroup lie one behind the other in one channel's row
        const uint32_t ch_u = blockIdx.x / fa.n_wg, k_u = blockIdx.x - ch_u * fa.n_wg;
        const uint8_t *wg_rows = adc + (size_t)ch_u * row_bytes + (size_t)k_u * (kFrontThreads * kBlkBytes);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = wave_base + 8u * j + (lane >> 3);
            const uint32_t piece = (lane & 7u) ^ line_swz(row);
            src[j] = wg_rows + (row * (uint32_t)kBlkBytes + piece * 16u);
        }
        ch = ch_u;
        b = k_u * kFrontThreads + tid;
    } else {
        uint32_t go = g0 + wave_base + (lane >> 3);
        go = go < total_blocks ? go : total_blocks - 1;
        uint32_t cho = go / n_blk, bo = go - cho * n_blk;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = wave_base + 8u * j + (lane >> 3);
            const uint32_t piece = (lane & 7u) ^ line_swz(row);
            src[j] = adc + (size_t)cho * row_bytes + (size_t)bo * kBlkBytes + piece * 16u;
            // eight blocks on, unless that runs past the last block of the launch
            if (go + 8u < total_blocks) {
                go += 8u;
                bo += 8u;
                while (bo >= n_blk) { bo -= n_blk; cho++; }
            }
        }
        const uint32_t g = g0 + tid;
        active = g < total_blocks;
        const uint32_t gc = active ? g : total_blocks - 1;
        ch = gc / n_blk;
        b = gc - ch * n_blk;
    }
    // phase and increment come in shifted left by 10 (32-bit wrap = accumulator
    // wrap); the sample loop runs on the << 8 form nco_fetch wants
    const uint32_t phi_in = phi32v[ch];
    const uint32_t phi32 = phi_in >> 2;
    const uint32_t acc32 = (acc0[ch] + (b * (uint32_t)kBlk) * phi_in) >> 2;
    unsigned long long accp = (unsigned long long)acc32 | ((unsigned long long)(acc32 + phi32) << 32);
    const unsigned long long phi2 = (unsigned long long)(2u * phi32) | ((unsigned long long)(2u * phi32) << 32);

#pragma unroll
    for (int j = 0; j < 8; j++) lds_dma16(src[j], &lines[(wave_base + 8 * j) * 8]);

    Cascade ci, cq;
    ci.clear();
    cq.clear();

    // own row in the line buffer: piece p sits at mine ^ (p << 4) (byte address)
    const uint32_t mine = tid * 128u + (line_swz(tid) << 4);
    const char *lbase = reinterpret_cast<const char *>(lines);
#define DDC_PIECE(p) (*reinterpret_cast<const uint4 *>(lbase + (mine ^ ((p) << 4))))

    unsigned last_i[3] = {0, 0, 0}, last_q[3] = {0, 0, 0};
    uint4 left0 = make_uint4(0, 0, 0, 0), left1 = make_uint4(0, 0, 0, 0);
#pragma unroll 1
    for (int a = 0; a < kChunksPerBlk; a++) {
        const int c = a & 7; // chunk within the half block of three lines
        if (c == 0 || c == 2 || c == 5) { // first chunk that needs a new line
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
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
        if (c == 1 || c == 4 || c == 7) { // last chunk of a line: free the buffer, start the next line
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (a + 1 < kChunksPerBlk) {
                const int next_line = (a >> 3) * 3 + (c == 1 ? 1 : c == 4 ? 2 : 3);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    lds_dma16(src[j] + next_line * 128, &lines[(wave_base + 8 * j) * 8]);
            }
        }
        int mi[5], mq[5];
        group8<0>(tab, v0.x, v0.y, v0.z, accp, phi2, mi, mq, last_i, last_q);
        group8<1>(tab, v0.w, v1.x, v1.y, accp, phi2, mi, mq, last_i, last_q);
        group8<2>(tab, v1.z, v1.w, v2.x, accp, phi2, mi, mq, last_i, last_q);
        group8<3>(tab, v2.y, v2.z, v2.w, accp, phi2, mi, mq, last_i, last_q);
        ci.step(mi);
        cq.step(mq);
    }
#undef DDC_PIECE
    {
        // the newest five samples of the block open the next window: take them back
        // out of the last step
        int mi[5] = {0, 0, 0, 0, 0}, mq[5] = {0, 0, 0, 0, 0};
        tail5_acc<-1>(last_i, mi);
        tail5_acc<-1>(last_q, mq);
        ci.inject_last(mi);
        cq.inject_last(mq);
    }

    // The five samples before this block open its window.  They are the last
    // samples of the block to the left: its thread hands their mixer outputs over
    // through LDS (the line buffer is free now); only the first thread of the
    // workgroup has to fetch and mix them itself, and the first block of a call
    // takes them from the previous call.
    {
        uint32_t *xch = reinterpret_cast<uint32_t *>(lines);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            xch[k * kFrontThreads + tid] = last_i[k];
            xch[(3 + k) * kFrontThreads + tid] = last_q[k];
        }
        __syncthreads();
        unsigned pki[3], pkq[3];
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
            uint32_t ph = (acc0[ch] + (b * (uint32_t)kBlk - 6u) * phi_in) >> 2;
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
                pki[k] = xch[k * kFrontThreads + tid - 1];
                pkq[k] = xch[(3 + k) * kFrontThreads + tid - 1];
            }
        }
        int mi[5] = {0, 0, 0, 0, 0}, mq[5] = {0, 0, 0, 0, 0};
        tail5_acc<1>(pki, mi);
        tail5_acc<1>(pkq, mq);
        ci.inject_first(mi);
        cq.inject_first(mq);
    }

    if (!FUSED && !active) return;
    unsigned long long Bi[5], Bq[5];
    ci.moments(Bi);
    cq.moments(Bq);
    if (b == n_blk - 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            carry_out[ch * 6 + k] = last_i[k];
            carry_out[ch * 6 + 3 + k] = last_q[k];
        }
    }
    if (!FUSED) {
        unsigned long long *rec = mom + ((size_t)ch * (kMomHist + n_blk) + kMomHist + b) * 10;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            rec[e] = Bi[e];
            rec[5 + e] = Bq[e];
        }
        return;
    }
    rx_cic_epilogue(fa, ch, b, n_blk, Bi, Bq, reinterpret_cast<char *>(lines), tid);
}

/* The ingest kernel of the call plan with separate kernels (any block count). */
__global__ __launch_bounds__(kFrontThreads) void ddc_rx_front(
    const uint8_t *__restrict__ adc, size_t row_bytes, uint32_t n_blk, uint32_t total_blocks,
    const uint32_t *__restrict__ acc0, const uint32_t *__restrict__ phi32v,
    const uint32_t *__restrict__ carry_in, uint32_t *__restrict__ carry_out,
    unsigned long long *__restrict__ mom, const uint2 *__restrict__ nco_tab)
{
    rx_front_body<false>(adc, row_bytes, n_blk, total_blocks, acc0, phi32v, carry_in, carry_out, mom, nco_tab, RxFused{});
}

/*
 * Kernel 1 of the two-kernel call plan, for block counts that are a multiple of 256 (a
 * workgroup then owns 256 consecutive windows of ONE channel): ingest as above, then
 * the workgroup turns its windows' moments into rx_cic outputs through LDS
 * (rx_cic_epilogue) and leaves only nine edge records per workgroup in memory.
 */
__global__ __launch_bounds__(kFrontThreads, 3) void ddc_rx_front_cic(
    const uint8_t *__restrict__ adc, size_t row_bytes, uint32_t n_blk, uint32_t total_blocks,
    const uint32_t *__restrict__ phi32v, const uint2 *__restrict__ nco_tab, const RxFused fa)
{
    rx_front_body<true>(adc, row_bytes, n_blk, total_blocks, fa.acc0, phi32v, fa.carry_saved, fa.carry_work, nullptr, nco_tab, fa);
}

/*
 * Kernel 2 of the two-kernel call plan: one workgroup per (channel, tile of 1024 frames)
 * runs rx_ciccomp, rx_hilb and data_delay for the tile (rx_tile_finish); the workgroup
 * that finishes a channel's last tile then saves what the channel carries into the
 * next call (rx_channel_finish).  With several tiles per channel that workgroup is
 * found by an arrival counter: tile 0 reads the old histories which the save overwrites.
 */
__global__ __launch_bounds__(256) void ddc_rx_tail(uint32_t n_blk, const RxFused fa)
{
    __shared__ __align__(16) char lds[kLdsFlag + 16];
    const uint32_t t = threadIdx.x, ch = blockIdx.x / fa.n_tiles, j = blockIdx.x - ch * fa.n_tiles;
    const uint32_t nf = n_blk / 2;
    ChannelCarry carry;
    if (fa.n_tiles == 1) {
        // the only tile of its channel: it saves the state itself, everything it needs for that is at hand
        rx_channel_load(fa, ch, n_blk, t, carry);
        rx_tile_finish(fa, ch, j, n_blk, lds, t);
        // frame nf - 256 + t is Hilbert tile sample nf - 1 + t (history included when the call is shorter than 256 frames)
        const uint32_t ci = (uint16_t)reinterpret_cast<const int16_t *>(lds + kLdsXs)[nf - 1 + t];
        const uint32_t cq = (uint16_t)reinterpret_cast<const int16_t *>(lds + kLdsCq)[nf + t];
        rx_channel_store(fa, ch, t, carry, ci | (cq << 16));
        return;
    }
    rx_tile_finish(fa, ch, j, n_blk, lds, t);
    uint32_t *flag = reinterpret_cast<uint32_t *>(lds + kLdsFlag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores of comp_tail have landed
    __syncthreads();
    if (t == 0) {
        const uint32_t last = count_in(fa.chan_cnt + ch, fa.n_tiles) ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        flag[0] = last;
    }
    __syncthreads();
    if (!flag[0]) return;
    rx_channel_load(fa, ch, n_blk, t, carry);
    rx_channel_store(fa, ch, t, carry, fa.comp_tail[(size_t)ch * kCompHist + t]);
}

/*
 * Kernel 2.  rx_cic output m of this call from the moments of windows m-5..m-1
 * (records m..m+4): y = wrap60(sum_j sum_e COEF[j][e] * B[m-1-j][e]) >> 44
 * (FPGA/rx_cic.vhd:391); the records hold 2 B, so the bits sit one place up.
 * cic: [channel][kCicHist + n_blk][2] int16.
 */
__global__ __launch_bounds__(256) void ddc_rx_cic_combine(const unsigned long long *__restrict__ mom, uint32_t n_blk,
                                                         uint32_t tiles, int16_t *__restrict__ cic)
{
    // 256 outputs of one channel need 260 consecutive records: staged through LDS
    // so that the 80-byte records are read from memory in whole 16-byte lanes
    __shared__ uint4 stage[260 * 5];
    const uint32_t ch = blockIdx.x / tiles, m0 = (blockIdx.x - ch * tiles) * 256u, t = threadIdx.x;
    const uint32_t have = kMomHist + n_blk - m0;
    const uint32_t count = (have < 260u ? have : 260u) * 5u;
    const uint4 *src = reinterpret_cast<const uint4 *>(mom + ((size_t)ch * (kMomHist + n_blk) + m0) * 10);
    for (uint32_t i = t; i < count; i += 256) stage[i] = src[i];
    __syncthreads();
    const uint32_t m = m0 + t;
    if (m >= n_blk) return;
    const unsigned long long *base = reinterpret_cast<const unsigned long long *>(stage) + (size_t)t * 10;
    unsigned long long acc[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const unsigned long long *rec = base + (size_t)(4 - j) * 10;
#pragma unroll
        for (int e = 0; e < 5; e++) {
            acc[0] += coef_mul(j, e, rec[e]);
            acc[1] += coef_mul(j, e, rec[5 + e]);
        }
    }
    int16_t *o = cic + ((size_t)ch * (kCicHist + n_blk) + kCicHist + m) * 2;
    o[0] = (int16_t)((long long)(acc[0] << 3) >> 48);
    o[1] = (int16_t)((long long)(acc[1] << 3) >> 48);
}

/*
 * Optional pass: running minimum / maximum of the ADC words of every channel,
 * the ADC_MIN / ADC_MAX registers of FPGA/stm32_interface.v:384-397 that the
 * status frame reports.  grid (x = slices of a row, y = channel); every thread
 * walks 48-byte pieces (32 samples) of its channel's row.
 */
__global__ __launch_bounds__(256) void ddc_adc_minmax(const uint8_t *__restrict__ adc, size_t row_bytes,
                                                      uint32_t pieces, int *__restrict__ mn, int *__restrict__ mx)
{
    __shared__ int s_mn, s_mx;
    if (threadIdx.x == 0) { s_mn = 4095; s_mx = -4096; }
    __syncthreads();
    const uint32_t ch = blockIdx.y;
    const uint4 *row = reinterpret_cast<const uint4 *>(adc + (size_t)ch * row_bytes);
    int lo = 4095, hi = -4096;
    for (uint32_t pc = blockIdx.x * blockDim.x + threadIdx.x; pc < pieces; pc += gridDim.x * blockDim.x) {
        const uint4 v0 = row[3 * (size_t)pc], v1 = row[3 * (size_t)pc + 1], v2 = row[3 * (size_t)pc + 2];
        const uint32_t d[12] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w };
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int a[8];
            unpack8(d[3 * g], d[3 * g + 1], d[3 * g + 2], a);
#pragma unroll
            for (int k = 0; k < 8; k++) { lo = min(lo, a[k]); hi = max(hi, a[k]); }
        }
    }
    atomicMin(&s_mn, lo);
    atomicMax(&s_mx, hi);
    __syncthreads();
    if (threadIdx.x == 0) { atomicMin(&mn[ch], s_mn); atomicMax(&mx[ch], s_mx); }
}


/*
 * Kernel 3 of the call plan with separate kernels.  rx_ciccomp (ciccomp_quad) for
 * 1024 consecutive outputs of one channel per workgroup; writes SPEC I/Q to the
 * caller (rows out_pitch frames apart) and to comp: [channel][kCompHist + nf][2].
 */
__global__ __launch_bounds__(256) void ddc_rx_ciccomp(const int16_t *__restrict__ cic, uint32_t n_blk, uint32_t tiles,
                                                     int16_t *__restrict__ comp, int16_t *__restrict__ spec_iq, uint32_t out_pitch)
{
    constexpr int kIn = 2 * kTailTile + 64;          // samples staged per branch: 2 n0 - 64 .. 2 n0 + 2047
    __shared__ __align__(16) int16_t si[kIn + 8], sq[kIn + 8];
    const uint32_t nf = n_blk / 2;
    const uint32_t ch = blockIdx.x / tiles, n0 = (blockIdx.x - ch * tiles) * (uint32_t)kTailTile, t = threadIdx.x;
    // staged sample j is entry kCicHist + 2 n0 - 64 + j of the channel (kCicHist = 64: entry 2 n0 + j)
    const uint32_t *x = reinterpret_cast<const uint32_t *>(cic) + (size_t)ch * (kCicHist + n_blk);
    const uint32_t last = kCicHist + n_blk - 1;
    for (uint32_t j = t; j < (uint32_t)kIn + 8; j += 256) {
        uint32_t e = 2 * n0 + j;
        e = e < last ? e : last;
        const uint32_t v = x[e];
        si[j] = (int16_t)(v & 0xFFFFu);
        sq[j] = (int16_t)(v >> 16);
    }
    __syncthreads();
    uint32_t packed[4];
    ciccomp_quad(reinterpret_cast<const uint4 *>(si) + t, reinterpret_cast<const uint4 *>(sq) + t, packed);
    const uint32_t n = n0 + 4 * t;
    if (n >= nf) return;
    uint32_t *oc = reinterpret_cast<uint32_t *>(comp) + (size_t)ch * (kCompHist + nf) + kCompHist + n;
    uint32_t *os = reinterpret_cast<uint32_t *>(spec_iq) + (size_t)ch * out_pitch + n;
    if (n + 3 < nf && (nf & 3u) == 0 && ((uintptr_t)os & 15u) == 0) { // whole quad, 16-byte aligned rows
        *reinterpret_cast<uint4 *>(oc) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        *reinterpret_cast<uint4 *>(os) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (n + k < nf) { oc[k] = packed[k]; os[k] = packed[k]; }
    }
}

/*
 * Kernel 4 of the call plan with separate kernels.  VOICE_I = rx_hilb(e_I)
 * (hilbert_quad_sums / hilbert_round); VOICE_Q = e_Q delayed by 130
 * (FPGA/data_delay.v).  A workgroup computes 1024 consecutive outputs of one
 * channel: waves 0-1 take the even outputs, waves 2-3 the odd ones.
 */
__global__ __launch_bounds__(256) void ddc_rx_hilbert(const int16_t *__restrict__ comp, uint32_t nf, uint32_t tiles,
                                                     int16_t *__restrict__ voice_iq, uint32_t out_pitch)
{
    __shared__ __align__(16) uint32_t xs[2][kHilbIn / 2]; // [0]: pairs (2m, 2m+1), [1]: pairs (2m+1, 2m+2)
    __shared__ uint32_t p0[kHilbIn / 32 + 2], p1[kHilbIn / 32 + 2];
    __shared__ __align__(16) uint32_t taps[144];
    const uint32_t ch = blockIdx.x / tiles, n0 = (blockIdx.x - ch * tiles) * (uint32_t)kTailTile, t = threadIdx.x;
    hilbert_stage_taps(taps, t);
    const uint32_t *x = reinterpret_cast<const uint32_t *>(comp) + (size_t)ch * (kCompHist + nf);
    const uint32_t last = kCompHist + nf - 1;
    int16_t *sa = reinterpret_cast<int16_t *>(xs[0]), *sb = reinterpret_cast<int16_t *>(xs[1]);
    // tile sample j is entry n0 + 1 + j of the channel (sample n0 - 255 + j)
#pragma unroll
    for (uint32_t j = t; j < (uint32_t)kHilbIn; j += 256) {
        uint32_t e = n0 + 1 + j;
        e = e < last ? e : last;
        const int16_t v = (int16_t)(x[e] & 0xFFFFu);
        sa[j] = v;
        if (j) sb[j - 1] = v;
    }
    if (t == 0) sb[kHilbIn - 1] = 0;
    __syncthreads();
    hilbert_planes(sa, p0, p1, t);
    __syncthreads();
    const uint32_t par = t >> 7, g = t & 127u;
    int S[4];
    hilbert_quad_sums(reinterpret_cast<const uint4 *>(&xs[0][0]) + par * (kHilbIn / 8) + g, taps, S);
    int16_t vi[4];
    hilbert_round4(S, 8 * g + par, p0, p1, vi);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = 8 * g + par + 2 * k, n = n0 + o;
        if (n >= nf) continue;
        const int16_t vq = (int16_t)((int)x[kCompHist + n - 130] >> 16);
        reinterpret_cast<uint32_t *>(voice_iq)[(size_t)ch * out_pitch + n] =
            (uint32_t)(uint16_t)vi[k] | ((uint32_t)(uint16_t)vq << 16);
    }
}

} // namespace ddc
