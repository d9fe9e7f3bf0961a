/*
 * ddc_rx_front_mx: kernel 1 of the two-kernel RX call (NCO + mixer + rx_cic integrate /
 * decimate: FPGA/nco.qsys:56-85, FPGA/mixer.v:55-71, FPGA/rx_mixer_shift.v:9,
 * FPGA/rx_cic.vhd:147-404) with the window moments formed by the matrix cores.
 *
 * ddc_rx_front_cic spends 6 of its 18 vector instructions per sample on the five moments
 *     B[e] = sum_d C(d + e, e) * 2 x[d],   d = distance of a sample from the window end,
 * and sits on the vector-ALU issue roof.  The moments are a product of the 512 samples of
 * a window with a constant 512 x 5 matrix, which is what v_mfma_i32_16x16x64_i8 is for:
 * the weights (up to 2^32) are cut into 1 + 2 + 3 + 4 + 5 = 15 signed bytes (the rows of
 * the first operand), the doubled 15-bit mixer output 2 x into a signed high byte and the
 * 7 bits below it (two second operands per branch).  32-bit accumulators hold a whole
 * window (512 * 128 * 128 = 2^23).  The matrix pipe runs beside the vector ALU, which is
 * left with the NCO, the mixer and 3.5 byte shuffles per sample.
 *
 * Work split: a wave takes 16 windows at a time (the 16 columns); lane (n, g) = (lane % 16,
 * lane / 16) mixes samples 16 g .. 16 g + 15 of every 64-sample step of window n, which are
 * exactly the 16 K slots the instruction reads from that lane.  A wave makes four such
 * passes, a workgroup (4 waves) 256 windows of one channel, as ddc_rx_front_cic does, so
 * rx_cic_epilogue and ddc_rx_tail are shared with it unchanged.
 *
 * A window is the last five samples of the block before it and the first 507 of its own.
 * The five trailing samples of a block therefore carry zero weight in the block's own
 * rows; their byte planes are moved one lane up (DPP row_shr: lane n holds window n) and
 * multiplied with a ninth weight row into the same accumulators.  The first window of a
 * pass takes them from the pass before (kept in lane 0 of the row), the first window of a
 * wave from the wave before through LDS and the first window of the workgroup by mixing
 * the six samples before it again (or from the previous call's carries): wave 0 makes
 * those four up at the end with one more step of the matrix instruction.
 *
 * Staging: wave-private.  Half a window (three 128-byte lines) of each of the wave's 16
 * windows at a time, straight into LDS (global_load_lds, 16 bytes a lane, eight lanes per
 * line), rows 400 bytes apart so that the lanes' 8-byte reads fall into different banks.
 * The lines of the next half are requested when the last two steps of the current one
 * are in registers.  No workgroup barrier inside the sample loop.
 */
#pragma once
#include "ddc_rx_kernels.hip.h"

namespace ddc {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kMxRows = 9;               // weight rows: the eight 64-sample steps of a block, the five samples before it
constexpr int kMxTabBytes = kMxRows * 1024;
constexpr int kMxStride = 400;           // bytes between the rows (windows) of a wave's staging buffer
constexpr int kMxStage = 16 * kMxStride; // per wave
constexpr int kMxDmas = 7;               // 16 rows x 25 slots of 16 bytes = 400 slots, 64 per load
constexpr int kMxLastD = 506;            // distance of a block's first sample from its window's end

// matrix row of byte plane p (0 = least significant) of weight e.  Lane group g holds rows
// 4 g .. 4 g + 3 of the result: g = 0: (4,0..3); g = 1: (4,4), (3,0..2); g = 2: (3,3),
// (2,0..2); g = 3: (1,0), (1,1), (0,0), unused.
__host__ __device__ constexpr int mx_row(int e, int p)
{
    return e == 4 ? p : e == 3 ? 5 + p : e == 2 ? 9 + p : e == 1 ? 12 + p : 14;
}

// Weight table: [kMxRows][lane 16 g + row][16 bytes]: byte j of lane (row, g) in step c is the
// byte plane `row` of the weight of block sample 64 c + 16 g + j.
inline bool build_mx_table(uint8_t *tab /* kMxTabBytes */)
{
    for (int i = 0; i < kMxTabBytes; i++) tab[i] = 0;
    auto put = [&](int step, int g, int j, int d) {
        for (int e = 0; e < 5; e++) {
            long long v = binom_gen(d + e, e);
            for (int p = 0; p <= e; p++) {
                const int dig = (int)(((v + 128) & 255) - 128); // signed digit base 256
                v = (v - dig) / 256;
                tab[step * 1024 + (16 * g + mx_row(e, p)) * 16 + j] = (uint8_t)(int8_t)dig;
            }
            if (v != 0) return false;
        }
        return true;
    };
    for (int c = 0; c < 8; c++)
        for (int g = 0; g < 4; g++)
            for (int j = 0; j < 16; j++) {
                const int s = 64 * c + 16 * g + j;
                if (s > kMxLastD) continue; // the newest five samples belong to the next window
                if (!put(c, g, j, kMxLastD - s)) return false;
            }
    // the five samples before the block: K slots 11..15 of lane group 3, d = 511..507
    for (int j = 11; j < 16; j++)
        if (!put(8, 3, j, kBlk - 1 - (j - 11))) return false;
    return true;
}

// Four samples at a time, in two halves, so that the table fetches of one group of four are under
// way while the group before is being mixed (a wave would otherwise sit out an LDS round trip per
// group): mx_quad_fetch makes the table addresses and fine-sine operands and issues the fetches,
// mx_quad_finish does the angle sums, the products and the packed pairs {2 x_even | 2 x_odd << 16}
// (bit 0 of either half undefined, see mx_pack4).
struct MxQuad {
    uint2 e[4];
    uint32_t w[4];
};
// (phi2_b: what the phase pair advances by behind the group's last two samples)
__device__ __forceinline__ void mx_quad_fetch(const uint2 *tab, unsigned long long &accp, unsigned long long phi2,
                                              unsigned long long phi2_b, MxQuad &q)
{
    unsigned long long p = accp;
    nco_fetch_pair(tab, p, q.e[0], q.e[1], q.w[0], q.w[1]);
    p = pair_add(p, phi2);
    nco_fetch_pair(tab, p, q.e[2], q.e[3], q.w[2], q.w[3]);
    accp = pair_add(p, phi2_b);
}
// Four mixer products of one branch -> two packed pairs.  The doubling that takes the 23-bit wrap
// (see pack_x2) is one 64-bit shift for all four: what crosses a 16-bit boundary is the bit the
// doubling drops, and it lands in bit 0 of the field above, which is zero in 2 x and which
// mx_planes does not read.  A pair that leaves the kernel is cleaned with mx_pair_clean.
__device__ __forceinline__ void mx_pack4(int p0, int p1, int p2, int p3, unsigned &pk0, unsigned &pk1)
{
    const unsigned y0 = __builtin_amdgcn_perm((unsigned)p1, (unsigned)p0, 0x06050201u);
    const unsigned y1 = __builtin_amdgcn_perm((unsigned)p3, (unsigned)p2, 0x06050201u);
    const unsigned long long y = (unsigned long long)y0 | ((unsigned long long)y1 << 32);
    unsigned long long r;
    asm("v_lshl_add_u64 %0, %1, 1, 0" : "=v"(r) : "v"(y));
    pk0 = (unsigned)r;
    pk1 = (unsigned)(r >> 32);
}
__device__ __forceinline__ unsigned mx_pair_clean(unsigned pk) { return pk & 0xFFFEFFFEu; }

__device__ __forceinline__ void mx_quad_finish(const MxQuad &q, const int *a, unsigned *pi, unsigned *pq)
{
    int i[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) nco_finish(q.e[k], q.w[k], a[k], i[k], v[k]);
    mx_pack4(i[0], i[1], i[2], i[3], pi[0], pi[1]);
    mx_pack4(v[0], v[1], v[2], v[3], pq[0], pq[1]);
}

// Two packed pairs (four samples) -> their high bytes and the seven bits below:
// 2 x = 256 hi + 2 lo7 with hi a signed byte (the 23-bit wrap was taken by the doubling) and
// lo7 = (low byte) >> 1 in 0..127 (v_lerp_u8 with zeros is a per-byte shift right; bit 0 of
// the low byte is not read).
__device__ __forceinline__ void mx_planes(unsigned pk0, unsigned pk1, int &hi, int &lo)
{
    hi = (int)__builtin_amdgcn_perm(pk1, pk0, 0x07050301u);
    lo = (int)__builtin_amdgcn_lerp(__builtin_amdgcn_perm(pk1, pk0, 0x06040200u), 0u, 0u);
}

// What one lane keeps of a pass: lane group g holds moment 4 - g of window n for both
// branches (group 3: moments 1 and 0).
struct MxPart {
    long long vi, vq;
    int bi, bq;
};

// Accumulators of a pass -> 64-bit moments (of 2 x).  Row r of lane group g is
// t[r] = 128 hi + lo7 of byte plane mx_row^-1(4 g + r); the top plane of moments 4 and 3
// sits in row 0 of the group above and comes down by ds_bpermute.
__device__ __forceinline__ void mx_recombine(const v4i &ih, const v4i &il, const v4i &qh, const v4i &ql, uint32_t lane,
                                             MxPart &o)
{
    const uint32_t g = lane >> 4;
    int ti[4], tq[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        ti[r] = (int)(((uint32_t)ih[r] << 7) + (uint32_t)il[r]);
        tq[r] = (int)(((uint32_t)qh[r] << 7) + (uint32_t)ql[r]);
    }
    const int up = (int)(((lane + 16u) & 63u) * 4u);
    const int ui = __builtin_amdgcn_ds_bpermute(up, ti[0]), uq = __builtin_amdgcn_ds_bpermute(up, tq[0]);
    const bool ends = g == 0 || g == 3;
    const int m0 = ends ? 2 : 0;
    const int m1 = ends ? (1 << 9) : 2;
    const int m2 = g == 0 ? (1 << 17) : g == 3 ? 0 : (1 << 9);
    const int m3 = g == 0 ? (1 << 25) : g == 3 ? 0 : (1 << 17);
    const int mu = g == 1 ? (1 << 25) : 0;
    const uint32_t hmask = g == 0 ? 0xFFFFFFFFu : 0u;
    long long vi = (long long)ti[0] * m0 + (long long)ti[1] * m1 + (long long)ti[2] * m2 + (long long)ti[3] * m3 +
                   (long long)ui * mu;
    long long vq = (long long)tq[0] * m0 + (long long)tq[1] * m1 + (long long)tq[2] * m2 + (long long)tq[3] * m3 +
                   (long long)uq * mu;
    vi += (long long)((unsigned long long)((((uint32_t)ui) << 1) & hmask) << 32);
    vq += (long long)((unsigned long long)((((uint32_t)uq) << 1) & hmask) << 32);
    o.vi = vi;
    o.vq = vq;
    o.bi = 2 * ti[2];
    o.bq = 2 * tq[2];
}

__global__ __launch_bounds__(kFrontThreads, 3) void ddc_rx_front_mx(
    const uint8_t *__restrict__ adc, size_t row_bytes, uint32_t n_blk, uint32_t n_ch, const uint32_t *__restrict__ phi32v,
    const uint2 *__restrict__ nco_tab, const uint4 *__restrict__ mx_tab, const RxFused fa)
{
    __shared__ uint2 tab[2048];
    __shared__ __align__(16) char stage[4 * kMxStage]; // after the sample loop: moments [10][256], then rx_cic_epilogue's
    __shared__ uint4 wtab[kMxRows * 64];
    __shared__ uint32_t xch[4][8];
    __shared__ unsigned long long tfix[4][10];
    static_assert(4 * kMxStage >= 10 * kFrontThreads * 8, "the staging buffer also holds the moments of 256 windows");

    // wv is the same in every lane of a wave: through readfirstlane, so that what depends on it alone
    // (the wave's rows, its staging buffer) stays in scalar registers
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = __builtin_amdgcn_readfirstlane(tid >> 6), wave_base = wv * 64u;
    const uint32_t n = lane & 15u, g = lane >> 4;

    // grid: x = workgroup within the channel's row, (y, z) = channel: no division, everything scalar
    const uint32_t ch = blockIdx.z * gridDim.y + blockIdx.y, kw = blockIdx.x;
    if (ch >= n_ch) return;
    // the coarse NCO table and the weight rows: straight into LDS, waited for with the first lines
#pragma unroll
    for (int i = 0; i < 4; i++)
        lds_dma16(reinterpret_cast<const uint8_t *>(nco_tab) + (i * kFrontThreads + tid) * 16u,
                  reinterpret_cast<uint4 *>(tab) + i * kFrontThreads + wave_base);
#pragma unroll
    for (int i = 0; i < 2; i++)
        lds_dma16(reinterpret_cast<const uint8_t *>(mx_tab) + (i * kFrontThreads + tid) * 16u, wtab + i * kFrontThreads + wave_base);
    if (wv == 0) lds_dma16(reinterpret_cast<const uint8_t *>(mx_tab) + (2 * kFrontThreads + tid) * 16u, wtab + 2 * kFrontThreads);

    // the 64 windows of the wave lie one behind the other in the channel's row
    const uint8_t *wave_rows = adc + (size_t)ch * row_bytes + ((size_t)kw * kFrontThreads + wv * 64u) * kBlkBytes;

    // staging: load i of a half puts slot 64 i + lane; slot 25 r + q holds piece q of row r (slot 24 of a row is padding)
    uint32_t doff[kMxDmas];
#pragma unroll
    for (int i = 0; i < kMxDmas; i++) {
        const uint32_t s = 64u * i + lane, r = s / 25u, q = s - 25u * r;
        doff[i] = (s < 400u && q < 24u) ? r * (uint32_t)kBlkBytes + q * 16u : 0xFFFFFFFFu;
    }
    char *wstage = stage + wv * kMxStage;
#define DDC_MX_ISSUE(it_)                                                                                   \
    do {                                                                                                    \
        /* the same address in every lane: through readfirstlane, so that one scalar base serves the loads */ \
        const uint64_t ha_ = (uint64_t)(wave_rows + (((it_) >> 1) * (16u * kBlkBytes) + ((it_) & 1u) * 384u)); \
        /* (the builtin returns int: without the casts the low half would be sign-extended over the high) */  \
        const uint32_t hlo_ = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ha_);                      \
        const uint32_t hhi_ = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ha_ >> 32));              \
        const uint8_t *half_ = reinterpret_cast<const uint8_t *>(((uint64_t)hhi_ << 32) | (uint64_t)hlo_);  \
        _Pragma("unroll") for (int i_ = 0; i_ < kMxDmas; i_++)                                              \
            if (doff[i_] != 0xFFFFFFFFu)                                                                    \
                lds_dma16(half_ + doff[i_], reinterpret_cast<uint4 *>(wstage + 1024 * i_));                 \
    } while (0)
    DDC_MX_ISSUE(0u);

    const uint32_t phi_in = phi32v[ch], phi32 = phi_in >> 2, acc_ch = fa.acc0[ch];
    const unsigned long long phi2 = (unsigned long long)(2u * phi32) | ((unsigned long long)(2u * phi32) << 32);
    // behind the 16 samples of a step a lane skips the 48 that the other three lane groups mix
    const unsigned long long phi50 = (unsigned long long)(50u * phi32) | ((unsigned long long)(50u * phi32) << 32);
    unsigned long long accp = 0;

    // planes of the five samples before a window, as the upper two dwords of a second operand each
    // (the lower two meet zero weights and may hold anything)
    v4i zih = {0, 0, 0, 0}, zil = {0, 0, 0, 0}, zqh = {0, 0, 0, 0}, zql = {0, 0, 0, 0};
    MxPart s0{}, s1{}, s2{}, s3{};
    unsigned last_i[3] = {0, 0, 0}, last_q[3] = {0, 0, 0};

    const char *rd = wstage + n * kMxStride + 24u * g;
    const char *wrow = reinterpret_cast<const char *>(wtab) + lane * 16u;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // the tables of all four waves are in place

#pragma unroll 1
    for (uint32_t p = 0; p < 4; p++) {
        {
            const uint32_t b = kw * kFrontThreads + wv * 64u + p * 16u + n;
            const uint32_t acc32 = (acc_ch + (b * (uint32_t)kBlk + 16u * g) * phi_in) >> 2;
            accp = (unsigned long long)acc32 | ((unsigned long long)(acc32 + phi32) << 32);
        }
        v4i aih, ail, aqh, aql;
        int hi_i[4], lo_i[4], hi_q[4], lo_q[4];
        uint2 nx0 = make_uint2(0, 0), nx1 = make_uint2(0, 0), nx2 = make_uint2(0, 0);
        MxQuad qa, qb;
        mx_quad_fetch(tab, accp, phi2, phi2, qa);
        mx_quad_fetch(tab, accp, phi2, phi2, qb);
#pragma unroll
        for (int hc = 0; hc < 8; hc++) {
            const int h = hc >> 2, c = hc & 3;
            if (c == 0 && (h || p)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's lines of the half have landed
            uint2 v0, v1, v2;
            if (c < 3) {
                v0 = *reinterpret_cast<const uint2 *>(rd + 96 * c);
                v1 = *reinterpret_cast<const uint2 *>(rd + 96 * c + 8);
                v2 = *reinterpret_cast<const uint2 *>(rd + 96 * c + 16);
            } else {
                v0 = nx0; v1 = nx1; v2 = nx2;
            }
            if (c == 2) { // the last step of the half too, then the buffer is free for the next half
                nx0 = *reinterpret_cast<const uint2 *>(rd + 96 * 3);
                nx1 = *reinterpret_cast<const uint2 *>(rd + 96 * 3 + 8);
                nx2 = *reinterpret_cast<const uint2 *>(rd + 96 * 3 + 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (h == 0) DDC_MX_ISSUE(2u * p + 1u);
                else if (p < 3) DDC_MX_ISSUE(2u * p + 2u);
            }
            const uint4 wa = *reinterpret_cast<const uint4 *>(wrow + 1024 * hc);
            const v4i A = { (int)wa.x, (int)wa.y, (int)wa.z, (int)wa.w };
            unsigned pi[8], pq[8];
            {
                // two groups of four are always under way: the step's first two were fetched behind
                // the step before (its phase does not need the samples), the next step's go out here
                int a[8], a2[8];
                unpack8(v0.x, v0.y, v1.x, a);
                unpack8(v1.y, v2.x, v2.y, a2);
                __builtin_amdgcn_sched_barrier(0);
                mx_quad_finish(qa, a, pi, pq);
                mx_quad_fetch(tab, accp, phi2, phi2, qa);
                __builtin_amdgcn_sched_barrier(0);
                mx_quad_finish(qb, a + 4, pi + 2, pq + 2);
                mx_quad_fetch(tab, accp, phi2, phi50, qb);
                __builtin_amdgcn_sched_barrier(0);
                mx_quad_finish(qa, a2, pi + 4, pq + 4);
                if (hc < 7) mx_quad_fetch(tab, accp, phi2, phi2, qa);
                __builtin_amdgcn_sched_barrier(0);
                mx_quad_finish(qb, a2 + 4, pi + 6, pq + 6);
                if (hc < 7) mx_quad_fetch(tab, accp, phi2, phi2, qb);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                mx_planes(pi[2 * m], pi[2 * m + 1], hi_i[m], lo_i[m]);
                mx_planes(pq[2 * m], pq[2 * m + 1], hi_q[m], lo_q[m]);
            }
            const v4i zero = {0, 0, 0, 0};
            aih = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, v4i{hi_i[0], hi_i[1], hi_i[2], hi_i[3]}, hc ? aih : zero, 0, 0, 0);
            ail = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, v4i{lo_i[0], lo_i[1], lo_i[2], lo_i[3]}, hc ? ail : zero, 0, 0, 0);
            aqh = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, v4i{hi_q[0], hi_q[1], hi_q[2], hi_q[3]}, hc ? aqh : zero, 0, 0, 0);
            aql = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, v4i{lo_q[0], lo_q[1], lo_q[2], lo_q[3]}, hc ? aql : zero, 0, 0, 0);
            if (hc == 7) {
                last_i[0] = pi[5]; last_i[1] = pi[6]; last_i[2] = pi[7];
                last_q[0] = pq[5]; last_q[1] = pq[6]; last_q[2] = pq[7];
            }
        }
        // The block's last five samples open the next window: their planes (K slots 11..15 = the
        // upper two dwords) go one lane up; lane 0 of a row keeps what the pass before left it.
#define DDC_MX_UP(z_, src_)                                                                                            \
        z_[2] = __builtin_amdgcn_update_dpp(z_[2], src_[2], 0x111 /* row_shr:1 */, 0xf, 0xf, false);                   \
        z_[3] = __builtin_amdgcn_update_dpp(z_[3], src_[3], 0x111, 0xf, 0xf, false)
#define DDC_MX_KEEP(z_, src_)                                                                                          \
        z_[2] = __builtin_amdgcn_update_dpp(0, src_[2], 0x121 /* row_ror:1 */, 0xf, 0xf, true);                        \
        z_[3] = __builtin_amdgcn_update_dpp(0, src_[3], 0x121, 0xf, 0xf, true)
        DDC_MX_UP(zih, hi_i); DDC_MX_UP(zil, lo_i); DDC_MX_UP(zqh, hi_q); DDC_MX_UP(zql, lo_q);
        const uint4 wt = *reinterpret_cast<const uint4 *>(wrow + 8 * 1024);
        const v4i At = { (int)wt.x, (int)wt.y, (int)wt.z, (int)wt.w };
        aih = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, zih, aih, 0, 0, 0);
        ail = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, zil, ail, 0, 0, 0);
        aqh = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, zqh, aqh, 0, 0, 0);
        aql = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, zql, aql, 0, 0, 0);
        DDC_MX_KEEP(zih, hi_i); DDC_MX_KEEP(zil, lo_i); DDC_MX_KEEP(zqh, hi_q); DDC_MX_KEEP(zql, lo_q);
#undef DDC_MX_UP
#undef DDC_MX_KEEP
        MxPart part;
        mx_recombine(aih, ail, aqh, aql, lane, part);
        if (p == 0) s0 = part;
        else if (p == 1) s1 = part;
        else if (p == 2) s2 = part;
        else s3 = part;
    }
#undef DDC_MX_ISSUE

    // what the call's last block hands to the next call: the mixer outputs of its last six samples
    if (kw == fa.n_wg - 1 && tid == kFrontThreads - 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            fa.carry_work[ch * 6 + k] = mx_pair_clean(last_i[k]);
            fa.carry_work[ch * 6 + 3 + k] = mx_pair_clean(last_q[k]);
        }
    }
    // lane 0 of row 3 now holds the planes of the wave's last five samples
    if (lane == 48) {
        const int zs[8] = { zih[2], zih[3], zil[2], zil[3], zqh[2], zqh[3], zql[2], zql[3] };
#pragma unroll
        for (int j = 0; j < 8; j++) xch[wv][j] = (uint32_t)zs[j];
    }
    __syncthreads();
    if (wv == 0) {
        // the first windows of the four waves (columns 0..3) still lack the five samples before them
        uint32_t zf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (g == 3 && n >= 1 && n < 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) zf[j] = xch[n - 1][j];
        }
        if (lane == 48) {
            unsigned pki[3], pkq[3];
            const uint32_t b = kw * kFrontThreads;
            if (kw == 0) { // first block of the call: from the call before
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    pki[k] = fa.carry_saved[ch * 6 + k];
                    pkq[k] = fa.carry_saved[ch * 6 + 3 + k];
                }
            } else { // they belong to another workgroup: mix them again
                const uint32_t *pw = reinterpret_cast<const uint32_t *>(adc + (size_t)ch * row_bytes + (size_t)b * kBlkBytes - 12);
                int a[8];
                unpack8(pw[0], pw[1], pw[2], a);
                uint32_t ph = (acc_ch + (b * (uint32_t)kBlk - 6u) * phi_in) >> 2;
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
            }
            int hh, ll;
            mx_planes(0u, pki[0], hh, ll); zf[0] = (uint32_t)hh; zf[2] = (uint32_t)ll;
            mx_planes(pki[1], pki[2], hh, ll); zf[1] = (uint32_t)hh; zf[3] = (uint32_t)ll;
            mx_planes(0u, pkq[0], hh, ll); zf[4] = (uint32_t)hh; zf[6] = (uint32_t)ll;
            mx_planes(pkq[1], pkq[2], hh, ll); zf[5] = (uint32_t)hh; zf[7] = (uint32_t)ll;
        }
        const uint4 wt = *reinterpret_cast<const uint4 *>(wrow + 8 * 1024);
        const v4i At = { (int)wt.x, (int)wt.y, (int)wt.z, (int)wt.w };
        const v4i zero = {0, 0, 0, 0};
        const v4i fih = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, v4i{0, 0, (int)zf[0], (int)zf[1]}, zero, 0, 0, 0);
        const v4i fil = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, v4i{0, 0, (int)zf[2], (int)zf[3]}, zero, 0, 0, 0);
        const v4i fqh = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, v4i{0, 0, (int)zf[4], (int)zf[5]}, zero, 0, 0, 0);
        const v4i fql = __builtin_amdgcn_mfma_i32_16x16x64_i8(At, v4i{0, 0, (int)zf[6], (int)zf[7]}, zero, 0, 0, 0);
        MxPart f;
        mx_recombine(fih, fil, fqh, fql, lane, f);
        if (n < 4) {
            tfix[n][4 - g] = (unsigned long long)f.vi;
            tfix[n][9 - g] = (unsigned long long)f.vq;
            if (g == 3) {
                tfix[n][0] = (unsigned long long)(long long)f.bi;
                tfix[n][5] = (unsigned long long)(long long)f.bq;
            }
        }
    }
    // every wave's passes -> moments [10][256] in the staging buffer (all lines have been consumed)
    unsigned long long *momv = reinterpret_cast<unsigned long long *>(stage);
    {
        const MxPart *parts[4] = { &s0, &s1, &s2, &s3 };
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t w = wv * 64u + p * 16u + n;
            momv[(4u - g) * kFrontThreads + w] = (unsigned long long)parts[p]->vi;
            momv[(9u - g) * kFrontThreads + w] = (unsigned long long)parts[p]->vq;
            if (g == 3) {
                momv[w] = (unsigned long long)(long long)parts[p]->bi;
                momv[5 * kFrontThreads + w] = (unsigned long long)(long long)parts[p]->bq;
            }
        }
    }
    __syncthreads();
    unsigned long long Bi[5], Bq[5];
#pragma unroll
    for (int e = 0; e < 5; e++) {
        Bi[e] = momv[e * kFrontThreads + tid];
        Bq[e] = momv[(5 + e) * kFrontThreads + tid];
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 5; e++) {
            Bi[e] += tfix[wv][e];
            Bq[e] += tfix[wv][5 + e];
        }
    }
    rx_cic_epilogue(fa, ch, kw * kFrontThreads + tid, n_blk, Bi, Bq, stage, tid);
}

} // namespace ddc
