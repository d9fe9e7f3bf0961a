// Issue-rate probe for the integer vector instructions the DDC kernels are made of.
//
// For every instruction a stream of independent copies (8 destination registers,
// each depending only on itself 8 instructions back) is run by 256-thread
// workgroups, W of them resident per CU (= W waves per SIMD; residency is set by
// the dynamic LDS size, 160 KiB / W), in a grid of 8 W workgroups per CU.
// Two figures per case, both "cycles per wave-instruction per SIMD", i.e. the
// reciprocal issue throughput of one SIMD (2.0 = a wave64 instruction leaves every
// 2 cycles = 32 lanes a cycle; 4.0 = 16 lanes a cycle):
//
//   wall : launch time (HIP events) x in-kernel clock x SIMDs / wave-instructions
//   wave : median over waves of (s_memtime span) / (instructions x W)
//
// The in-kernel clock is the median of s_memtime span / s_memrealtime span
// (100 MHz ticks) over the waves of the same launch.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/valu_issue tools/probe/valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { uint64_t cycles, realtime; };

#define BLOCK8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BLOCK32(OP) BLOCK8(OP) BLOCK8(OP) BLOCK8(OP) BLOCK8(OP)

extern __shared__ uint32_t dyn_lds[];

// destinations %0..%7 (TYPE wide), sources %8 (TYPE) %9 (32 bit) vector, %10 scalar
#define DEFB(NAME, BODY, TYPE)                                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(Stamp *out, int iters, uint32_t seed)             \
    {                                                                                                \
        TYPE r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 9u,        \
             r5 = r0 * 11u, r6 = r0 * 13u, r7 = r0 * 17u;                                            \
        TYPE a = (TYPE)seed * 2654435761u + threadIdx.x;                                             \
        uint32_t b = seed ^ 0x12345u;                                                                \
        uint32_t sc = seed & 0xFFFFu;                                                                \
        if (seed == 78u) dyn_lds[threadIdx.x] = seed;                                                \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                            \
        const uint64_t q0 = __builtin_amdgcn_s_memrealtime();                                        \
        for (int i = 0; i < iters; i++) {                                                            \
            asm volatile(BODY                                                                        \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                         : "v"(a), "v"(b), "s"(sc)                                                   \
                         : "vcc");                                                                   \
        }                                                                                            \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                            \
        const uint64_t q1 = __builtin_amdgcn_s_memrealtime();                                        \
        if ((threadIdx.x & 63u) == 0) {                                                              \
            Stamp s;                                                                                 \
            s.cycles = t1 - t0;                                                                      \
            s.realtime = q1 - q0;                                                                    \
            out[blockIdx.x * 4 + threadIdx.x / 64] = s;                                              \
        }                                                                                            \
        if ((uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x9E3779B9u && seed == 77u)          \
            out[0].cycles = (uint64_t)b;                                                             \
    }
#define DEFK(NAME, OP, TYPE) DEFB(NAME, BLOCK32(OP), TYPE)
#define DEF32(NAME, OP) DEFK(NAME, OP, uint32_t)
#define DEF64(NAME, OP) DEFK(NAME, OP, uint64_t)

#define S(x) #x
#define D(d) "%" S(d)
// the instructions of ddc_rx_front / ddc_tx_duc
#define OP_ADD(d)      "v_add_u32 " D(d) ", %8, " D(d) "\n"
#define OP_LSHR(d)     "v_lshrrev_b32 " D(d) ", 13, " D(d) "\n"
#define OP_AND(d)      "v_and_b32 " D(d) ", 0x3ff8, " D(d) "\n"
#define OP_ASHR(d)     "v_ashrrev_i32 " D(d) ", 16, " D(d) "\n"
#define OP_DOT2(d)     "v_dot2_i32_i16 " D(d) ", %8, %9, " D(d) "\n"
#define OP_DOT2S(d)    "v_dot2_i32_i16 " D(d) ", " D(d) ", %9, %10\n"
#define OP_DOT2C(d)    "v_dot2c_i32_i16 " D(d) ", %8, %9\n"
#define OP_MUL24(d)    "v_mul_i32_i24 " D(d) ", %8, " D(d) "\n"
#define OP_MUL24W(d)   "v_mul_i32_i24_sdwa " D(d) ", %8, " D(d) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define OP_PERM(d)     "v_perm_b32 " D(d) ", " D(d) ", %8, %9\n"
#define OP_PKADD(d)    "v_pk_add_u16 " D(d) ", " D(d) ", %8\n"
#define OP_MADU16(d)   "v_mad_u32_u16 " D(d) ", " D(d) ", %10, %9\n"
#define OP_BFE(d)      "v_bfe_i32 " D(d) ", " D(d) ", 4, 12\n"
#define OP_ALIGNBIT(d) "v_alignbit_b32 " D(d) ", " D(d) ", %8, 24\n"
#define OP_ADD64(d)    "v_add_co_u32 " D(d) ", vcc, " D(d) ", %8\n v_addc_co_u32 " D(d) ", vcc, " D(d) ", %9, vcc\n"
// candidates for replacing them
#define OP_FMA(d)      "v_fma_f32 " D(d) ", %8, %9, " D(d) "\n"
#define OP_FMAC(d)     "v_fmac_f32 " D(d) ", %8, %9\n"
#define OP_MULF(d)     "v_mul_f32 " D(d) ", %8, " D(d) "\n"
#define OP_ADDF(d)     "v_add_f32 " D(d) ", %8, " D(d) "\n"
#define OP_SUB(d)      "v_sub_u32 " D(d) ", %8, " D(d) "\n"
#define OP_OR(d)       "v_or_b32 " D(d) ", %8, " D(d) "\n"
#define OP_XOR(d)      "v_xor_b32 " D(d) ", %8, " D(d) "\n"
#define OP_LSHL(d)     "v_lshlrev_b32 " D(d) ", 5, " D(d) "\n"
#define OP_MOV(d)      "v_mov_b32 " D(d) ", %8\n"
#define OP_MINI(d)     "v_min_i32 " D(d) ", %8, " D(d) "\n"
#define OP_BFEU(d)     "v_bfe_u32 " D(d) ", " D(d) ", 16, 11\n"
#define OP_BFI(d)      "v_bfi_b32 " D(d) ", %9, " D(d) ", %8\n"
#define OP_ALIGNBYTE(d) "v_alignbyte_b32 " D(d) ", " D(d) ", %8, 3\n"
#define OP_LSHLOR(d)   "v_lshl_or_b32 " D(d) ", " D(d) ", 3, %9\n"
#define OP_ANDOR(d)    "v_and_or_b32 " D(d) ", " D(d) ", %8, %9\n"
#define OP_LSHLADD(d)  "v_lshl_add_u32 " D(d) ", " D(d) ", 3, %9\n"
#define OP_ADD3(d)     "v_add3_u32 " D(d) ", " D(d) ", %8, %9\n"
#define OP_MULU24(d)   "v_mul_u32_u24 " D(d) ", %8, " D(d) "\n"
#define OP_MAD24(d)    "v_mad_i32_i24 " D(d) ", %8, %9, " D(d) "\n"
#define OP_MULLO(d)    "v_mul_lo_u32 " D(d) ", " D(d) ", %8\n"
#define OP_MULHI(d)    "v_mul_hi_u32 " D(d) ", " D(d) ", %8\n"
#define OP_MULLO16(d)  "v_mul_lo_u16 " D(d) ", %8, " D(d) "\n"
#define OP_ADDU16(d)   "v_add_u16 " D(d) ", %8, " D(d) "\n"
#define OP_MADU16L(d)  "v_mad_legacy_u16 " D(d) ", " D(d) ", %8, %9\n"
#define OP_DOT4(d)     "v_dot4_i32_i8 " D(d) ", %8, %9, " D(d) "\n"
#define OP_DOT8(d)     "v_dot8_i32_i4 " D(d) ", %8, %9, " D(d) "\n"
#define OP_PKMUL(d)    "v_pk_mul_lo_u16 " D(d) ", " D(d) ", %8\n"
#define OP_PKMAD(d)    "v_pk_mad_i16 " D(d) ", %8, %9, " D(d) "\n"
#define OP_PKASHR(d)   "v_pk_ashrrev_i16 " D(d) ", 3, " D(d) "\n"
#define OP_PKLSHL(d)   "v_pk_lshlrev_b16 " D(d) ", 1, " D(d) "\n"
#define OP_PKFMAH(d)   "v_pk_fma_f16 " D(d) ", %8, %9, " D(d) "\n"
#define OP_CVTFI(d)    "v_cvt_f32_i32 " D(d) ", " D(d) "\n"
#define OP_CVTIF(d)    "v_cvt_i32_f32 " D(d) ", " D(d) "\n"
#define OP_CVTUB(d)    "v_cvt_f32_ubyte1 " D(d) ", " D(d) "\n"
#define OP_CVTPKI16(d) "v_cvt_pk_i16_i32 " D(d) ", " D(d) ", %8\n"
#define OP_MOVDPP(d)   "v_mov_b32_dpp " D(d) ", " D(d) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPP(d)   "v_add_u32_dpp " D(d) ", %8, " D(d) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADDSDWA(d)  "v_add_u32_sdwa " D(d) ", %8, " D(d) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_LSHLSDWA(d) "v_lshlrev_b32_sdwa " D(d) ", %9, " D(d) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
#define OP_CNDMASK(d)  "v_cndmask_b32 " D(d) ", " D(d) ", %8, vcc\n"
#define OP_SAD(d)      "v_sad_u32 " D(d) ", %8, %9, " D(d) "\n"
#define OP_XAD(d)      "v_xad_u32 " D(d) ", %8, %9, " D(d) "\n"
// pairs: does a slow instruction hide a fast one from the same wave?
#define OP_DOT2_ADD(d) "v_dot2_i32_i16 " D(d) ", %8, %9, " D(d) "\n v_add_u32 " D(d) ", %8, " D(d) "\n"
#define OP_DOT2_FMA(d) "v_dot2_i32_i16 " D(d) ", %8, %9, " D(d) "\n v_fma_f32 " D(d) ", %8, %9, " D(d) "\n"
#define OP_PERM_ADD(d) "v_perm_b32 " D(d) ", " D(d) ", %8, %9\n v_add_u32 " D(d) ", %8, " D(d) "\n"

DEF32(add_u32, OP_ADD)
DEF32(lshrrev_b32, OP_LSHR)
DEF32(and_b32_literal, OP_AND)
DEF32(ashrrev_i32, OP_ASHR)
DEF32(dot2_i32_i16, OP_DOT2)
DEF32(dot2_i32_i16_sgpr_addend, OP_DOT2S)
DEF32(dot2c_i32_i16, OP_DOT2C)
DEF32(mul_i32_i24, OP_MUL24)
DEF32(mul_i32_i24_sdwa_word1, OP_MUL24W)
DEF32(perm_b32, OP_PERM)
DEF32(pk_add_u16, OP_PKADD)
DEF32(mad_u32_u16, OP_MADU16)
DEF32(bfe_i32, OP_BFE)
DEF32(alignbit_b32, OP_ALIGNBIT)
DEF32(add64_pair, OP_ADD64)
DEF32(fma_f32, OP_FMA)
DEF32(fmac_f32, OP_FMAC)
DEF32(mul_f32, OP_MULF)
DEF32(add_f32, OP_ADDF)
DEF32(sub_u32, OP_SUB)
DEF32(or_b32, OP_OR)
DEF32(xor_b32, OP_XOR)
DEF32(lshlrev_b32, OP_LSHL)
DEF32(mov_b32, OP_MOV)
DEF32(min_i32, OP_MINI)
DEF32(bfe_u32, OP_BFEU)
DEF32(bfi_b32, OP_BFI)
DEF32(alignbyte_b32, OP_ALIGNBYTE)
DEF32(lshl_or_b32, OP_LSHLOR)
DEF32(and_or_b32, OP_ANDOR)
DEF32(lshl_add_u32, OP_LSHLADD)
DEF32(add3_u32, OP_ADD3)
DEF32(mul_u32_u24, OP_MULU24)
DEF32(mad_i32_i24, OP_MAD24)
DEF32(mul_lo_u32, OP_MULLO)
DEF32(mul_hi_u32, OP_MULHI)
DEF32(mul_lo_u16, OP_MULLO16)
DEF32(add_u16, OP_ADDU16)
DEF32(mad_legacy_u16, OP_MADU16L)
DEF32(dot4_i32_i8, OP_DOT4)
DEF32(dot8_i32_i4, OP_DOT8)
DEF32(pk_mul_lo_u16, OP_PKMUL)
DEF32(pk_mad_i16, OP_PKMAD)
DEF32(pk_ashrrev_i16, OP_PKASHR)
DEF32(pk_lshlrev_b16, OP_PKLSHL)
DEF32(pk_fma_f16, OP_PKFMAH)
DEF32(cvt_f32_i32, OP_CVTFI)
DEF32(cvt_i32_f32, OP_CVTIF)
DEF32(cvt_f32_ubyte1, OP_CVTUB)
DEF32(cvt_pk_i16_i32, OP_CVTPKI16)
DEF32(mov_b32_dpp, OP_MOVDPP)
DEF32(add_u32_dpp, OP_ADDDPP)
DEF32(add_u32_sdwa, OP_ADDSDWA)
DEF32(lshlrev_b32_sdwa, OP_LSHLSDWA)
DEF32(cndmask_b32, OP_CNDMASK)
DEF32(sad_u32, OP_SAD)
DEF32(xad_u32, OP_XAD)
DEF32(dot2_then_add, OP_DOT2_ADD)
DEF32(dot2_then_fma, OP_DOT2_FMA)
DEF32(perm_then_add, OP_PERM_ADD)
// the class mix of the ddc_rx_front loop (22 of the 4-cycle kind, 10 of the 2-cycle kind in 32),
// all independent: once spread evenly, once with the 2-cycle instructions in one run
#define OP_FMAMK(d)    "v_fmamk_f32 " D(d) ", " D(d) ", 0x58c90800, %8\n"
#define MIX_SPREAD "v_dot2_i32_i16 %0, %8, %9, %0\n v_mul_i32_i24 %1, %8, %1\n v_add_u32 %2, %8, %2\n" \
    "v_perm_b32 %3, %3, %8, %9\n v_dot2c_i32_i16 %4, %8, %9\n v_lshrrev_b32 %5, 13, %5\n" \
    "v_pk_add_u16 %6, %6, %8\n v_bfe_i32 %7, %7, 4, 12\n v_and_b32 %0, 0x3ff8, %0\n" \
    "v_dot2_i32_i16 %1, %8, %9, %1\n v_dot2c_i32_i16 %2, %8, %9\n v_fma_f32 %3, %8, %9, %3\n" \
    "v_mul_i32_i24 %4, %8, %4\n v_dot2c_i32_i16 %5, %8, %9\n v_add_u32 %6, %8, %6\n" \
    "v_perm_b32 %7, %7, %8, %9\n v_dot2c_i32_i16 %0, %8, %9\n v_and_b32 %1, 0x3ff8, %1\n" \
    "v_dot2c_i32_i16 %2, %8, %9\n v_dot2_i32_i16 %3, %8, %9, %3\n v_lshrrev_b32 %4, 13, %4\n" \
    "v_dot2c_i32_i16 %5, %8, %9\n v_mul_i32_i24 %6, %8, %6\n v_ashrrev_i32 %7, 16, %7\n" \
    "v_dot2c_i32_i16 %0, %8, %9\n v_pk_add_u16 %1, %1, %8\n v_add_u32 %2, %8, %2\n" \
    "v_dot2c_i32_i16 %3, %8, %9\n v_bfe_i32 %4, %4, 4, 12\n v_dot2c_i32_i16 %5, %8, %9\n" \
    "v_fma_f32 %6, %8, %9, %6\n v_dot2c_i32_i16 %7, %8, %9\n"
#define MIX_RUNS "v_add_u32 %0, %8, %0\n v_lshrrev_b32 %1, 13, %1\n v_and_b32 %2, 0x3ff8, %2\n" \
    "v_and_b32 %3, 0x3ff8, %3\n v_fma_f32 %4, %8, %9, %4\n v_add_u32 %5, %8, %5\n v_lshrrev_b32 %6, 13, %6\n" \
    "v_and_b32 %7, 0x3ff8, %7\n v_fma_f32 %0, %8, %9, %0\n v_ashrrev_i32 %1, 16, %1\n" \
    "v_dot2_i32_i16 %2, %8, %9, %2\n v_dot2_i32_i16 %3, %8, %9, %3\n v_mul_i32_i24 %4, %8, %4\n" \
    "v_mul_i32_i24 %5, %8, %5\n v_perm_b32 %6, %6, %8, %9\n v_pk_add_u16 %7, %7, %8\n" \
    "v_bfe_i32 %0, %0, 4, 12\n v_bfe_i32 %1, %1, 4, 12\n v_dot2_i32_i16 %2, %8, %9, %2\n" \
    "v_perm_b32 %3, %3, %8, %9\n v_pk_add_u16 %4, %4, %8\n v_mul_i32_i24 %5, %8, %5\n" \
    "v_dot2c_i32_i16 %6, %8, %9\n v_dot2c_i32_i16 %7, %8, %9\n v_dot2c_i32_i16 %0, %8, %9\n" \
    "v_dot2c_i32_i16 %1, %8, %9\n v_dot2c_i32_i16 %2, %8, %9\n v_dot2c_i32_i16 %3, %8, %9\n" \
    "v_dot2c_i32_i16 %4, %8, %9\n v_dot2c_i32_i16 %5, %8, %9\n v_dot2c_i32_i16 %6, %8, %9\n" \
    "v_dot2c_i32_i16 %7, %8, %9\n"
DEF32(fmamk_f32_literal, OP_FMAMK)
DEFB(front_mix_spread, MIX_SPREAD, uint32_t)
DEFB(front_mix_runs, MIX_RUNS, uint32_t)

#define OP64_ADDF64(d)  "v_add_f64 " D(d) ", " D(d) ", %8\n"
#define OP64_FMAF64(d)  "v_fma_f64 " D(d) ", " D(d) ", %8, %8\n"
#define OP64_LSHL(d)    "v_lshlrev_b64 " D(d) ", 3, " D(d) "\n"
#define OP64_MAD(d)     "v_mad_u64_u32 " D(d) ", vcc, %9, %9, " D(d) "\n"
#define OP64_PKFMA(d)   "v_pk_fma_f32 " D(d) ", " D(d) ", %8, %8\n"
#define OP64_PKADDF(d)  "v_pk_add_f32 " D(d) ", " D(d) ", %8\n"
DEF64(add_f64, OP64_ADDF64)
DEF64(fma_f64, OP64_FMAF64)
DEF64(lshlrev_b64, OP64_LSHL)
DEF64(mad_u64_u32, OP64_MAD)
DEF64(pk_fma_f32, OP64_PKFMA)
DEF64(pk_add_f32, OP64_PKADDF)

// LDS reads of the NCO-table kind (8 bytes per lane, per-lane address, table in
// the first 16 KiB of the dynamic LDS) beside arithmetic: do the two pipes issue
// side by side?  stride = table entries between neighbouring lanes.
template <int DOT_PER_READ>
__global__ __launch_bounds__(256) void k_lds_mix(Stamp *out, int iters, uint32_t seed, uint32_t stride)
{
    uint2 *lds = reinterpret_cast<uint2 *>(dyn_lds);
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_uint2(i * seed, i);
    uint32_t acc = 0, addr = (threadIdx.x * stride * 8u) & 0x3FF8u;
    uint32_t r0 = seed, r1 = seed * 3u, r2 = seed * 5u, r3 = seed * 7u;
    const uint32_t a = seed * 2654435761u + threadIdx.x, b = seed ^ 0x12345u;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint2 e = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lds) + addr);
            addr = (addr + stride * 8u * 7u) & 0x3FF8u; // two vector instructions per read
            acc ^= e.x + e.y;                            // and two to consume it
#pragma unroll
            for (int k = 0; k < DOT_PER_READ; k += 4) {
                asm volatile("v_dot2_i32_i16 %0, %4, %5, %0\n v_dot2_i32_i16 %1, %4, %5, %1\n"
                             "v_dot2_i32_i16 %2, %4, %5, %2\n v_dot2_i32_i16 %3, %4, %5, %3\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t q1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cycles = t1 - t0;
        s.realtime = q1 - q0;
        out[blockIdx.x * 4 + threadIdx.x / 64] = s;
    }
    if ((acc ^ r0 ^ r1 ^ r2 ^ r3) == 0x9E3779B9u && seed == 77u) out[0].cycles = 1;
}

typedef void (*Kern)(Stamp *, int, uint32_t);
typedef void (*KernLds)(Stamp *, int, uint32_t, uint32_t);

struct Result { double wall, wave, ghz; };

static const int kRounds = 8;
static int g_cu = 256;
static Stamp *g_out;
static hipEvent_t g_e0, g_e1;

static size_t lds_for(int w) { return ((160u * 1024u / w) / 1024u) * 1024u; }

template <typename F>
static Result measure(const void *fn, F launch, int w, double insts_per_wave)
{
    const size_t lds = lds_for(w);
    CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = g_cu * w * kRounds, n_waves = grid * 4;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) { // the first pass warms the clocks
        CHECK(hipEventRecord(g_e0, 0));
        launch(grid, lds, rep);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(g_e1, 0));
        CHECK(hipEventSynchronize(g_e1));
        CHECK(hipEventElapsedTime(&ms, g_e0, g_e1));
    }
    std::vector<Stamp> h(n_waves);
    CHECK(hipMemcpy(h.data(), g_out, n_waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> c(n_waves), f(n_waves);
    for (int i = 0; i < n_waves; i++) {
        c[i] = (double)h[i].cycles / (insts_per_wave * w);
        f[i] = h[i].realtime ? (double)h[i].cycles / (double)h[i].realtime * 0.1 : 0.0; // GHz
    }
    std::sort(c.begin(), c.end());
    std::sort(f.begin(), f.end());
    Result r;
    r.ghz = f[n_waves / 2];
    r.wave = c[n_waves / 2];
    r.wall = (double)ms * 1e-3 * r.ghz * 1e9 * (g_cu * 4.0) / ((double)n_waves * insts_per_wave);
    return r;
}

static const int kW[] = { 1, 2, 3, 4, 5, 6, 8 };

static void run_plain(const char *name, Kern k, int per_iter, int iters)
{
    printf("%-28s", name);
    double ghz = 0;
    char tail[256] = "";
    for (int w : kW) {
        const Result r = measure((const void *)k, [&](int grid, size_t lds, int rep) {
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, g_out, iters, 1234u + rep); }, w,
            (double)iters * per_iter);
        printf(" %6.2f", r.wall);
        char t[32];
        snprintf(t, sizeof t, " %6.2f", r.wave);
        strcat(tail, t);
        if (w == 4) ghz = r.ghz;
    }
    printf("  |%s  | %.2f\n", tail, ghz);
    fflush(stdout);
}

template <int DPR>
static void run_lds(const char *name, int iters)
{
    const uint32_t strides[4] = { 1, 2, 6, 16 };
    KernLds k = k_lds_mix<DPR>;
    for (uint32_t st : strides) {
        printf("%-18s stride %2u", name, st);
        char tail[256] = "";
        for (int w : kW) {
            // one loop body = 1 ds_read_b64 + 4 address / consume instructions + DPR dot2
            const Result r = measure((const void *)k, [&](int grid, size_t lds, int rep) {
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, g_out, iters, 99u + rep, st); }, w,
                (double)iters * 8);
            printf(" %6.1f", r.wall);
            char t[32];
            snprintf(t, sizeof t, " %6.1f", r.wave);
            strcat(tail, t);
        }
        printf("  |%s\n", tail);
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, %d iterations x 32 instructions per wave, %d W workgroups of 256 threads per CU\n",
           prop.gcnArchName, g_cu, iters, kRounds);
    CHECK(hipMalloc(&g_out, sizeof(Stamp) * g_cu * 8 * kRounds * 4));
    CHECK(hipEventCreate(&g_e0));
    CHECK(hipEventCreate(&g_e1));
    struct { const char *name; Kern k; int per_iter; } tests[] = {
#define T(n) { #n, k_##n, 32 }
#define T2(n) { #n " (x2)", k_##n, 64 }
        T(add_u32), T(lshrrev_b32), T(and_b32_literal), T(ashrrev_i32), T(dot2_i32_i16), T(dot2_i32_i16_sgpr_addend),
        T(dot2c_i32_i16), T(mul_i32_i24), T(mul_i32_i24_sdwa_word1), T(perm_b32), T(pk_add_u16), T(mad_u32_u16),
        T(bfe_i32), T(alignbit_b32), T2(add64_pair),
        T(fma_f32), T(fmac_f32), T(mul_f32), T(add_f32), T(sub_u32), T(or_b32), T(xor_b32), T(lshlrev_b32), T(mov_b32),
        T(min_i32), T(bfe_u32), T(bfi_b32), T(alignbyte_b32), T(lshl_or_b32), T(and_or_b32), T(lshl_add_u32),
        T(add3_u32), T(mul_u32_u24), T(mad_i32_i24), T(mul_lo_u32), T(mul_hi_u32), T(mul_lo_u16), T(add_u16),
        T(mad_legacy_u16), T(dot4_i32_i8), T(dot8_i32_i4), T(pk_mul_lo_u16), T(pk_mad_i16), T(pk_ashrrev_i16),
        T(pk_lshlrev_b16), T(pk_fma_f16), T(cvt_f32_i32), T(cvt_i32_f32), T(cvt_f32_ubyte1), T(cvt_pk_i16_i32),
        T(mov_b32_dpp), T(add_u32_dpp), T(add_u32_sdwa), T(lshlrev_b32_sdwa), T(cndmask_b32), T(sad_u32), T(xad_u32),
        T2(dot2_then_add), T2(dot2_then_fma), T2(perm_then_add), T(fmamk_f32_literal), T(front_mix_spread), T(front_mix_runs),
        T(add_f64), T(fma_f64), T(lshlrev_b64), T(mad_u64_u32), T(pk_fma_f32), T(pk_add_f32),
#undef T
#undef T2
    };
    printf("cycles per wave-instruction per SIMD; W = waves per SIMD; left: from wall time, right: median wave span\n");
    printf("%-28s", "instruction");
    for (int w : kW) printf("  W=%d  ", w);
    printf(" |");
    for (int w : kW) printf("  W=%d  ", w);
    printf(" | GHz at W=4\n");
    const char *only = argc > 2 ? argv[2] : nullptr; // optional: only the cases whose name contains this
    for (auto &t : tests)
        if (!only || strstr(t.name, only)) run_plain(t.name, t.k, t.per_iter, iters);
    if (only) return 0;
    printf("cycles per {1 ds_read_b64 (NCO-table fetch) + 4 plain VALU + N dot2} per SIMD, lanes `stride` entries apart\n");
    run_lds<0>("ds_read + 0 dot2", iters / 4);
    run_lds<8>("ds_read + 8 dot2", iters / 4);
    run_lds<16>("ds_read + 16 dot2", iters / 4);
    CHECK(hipFree(g_out));
    return 0;
}
