/*
 * Host side of the C ABI declared in include/ua3reo_ddc.h: context life cycle,
 * tuning, kernel launches, bus-frame helpers.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ua3reo_ddc.h"
#include "ddc_rx_kernels.hip.h"
#include "ddc_rx_mx_kernel.hip.h"
#include "ddc_tx_kernels.hip.h"
#include "ddc_audio_kernels.hip.h"
#include <array>

/* sinTable_f32 of CMSIS-DSP: sin(2 pi i / 512), i = 0..512, as the float of its 8-decimal literal
 * (the form the library's table source is printed in; the last entry reads -0.00000000f). */
static const std::array<float, 513> &arm_sin_table()
{
    static const std::array<float, 513> table = [] {
        std::array<float, 513> t{};
        for (int i = 0; i <= 512; i++) {
            char lit[32];
            snprintf(lit, sizeof lit, "%.8f", std::sin(2.0 * 3.14159265358979323846 * i / 512.0));
            t[i] = strtof(lit, nullptr);
        }
        return t;
    }();
    return table;
}


struct ddc_ctx {
    ddc_config cfg{};
    hipStream_t stream = nullptr;      // where the calls enqueue: own_stream unless ddc_set_stream chose another
    hipStream_t own_stream = nullptr;
    std::string err;
    // device state, per channel
    uint32_t *d_acc = nullptr;   // NCO phase << 10
    uint32_t *d_phi = nullptr;   // tuning word << 10
    uint32_t *d_carry[2] = {nullptr, nullptr}; // [0] saved between calls, [1] the copy one call reads
    uint2 *d_nco_tab = nullptr;
    uint4 *d_mx_tab = nullptr;   // byte planes of the window-moment weights (ddc_rx_front_mx)
    // work buffers, sized for cap_blk blocks of 512 samples per channel
    uint32_t cap_blk = 0;
    // rx_cic outputs of the last RX call (work buffer or static area): one piece per part the call ran as
    uint32_t last_blk = 0;             // blocks of the whole call
    uint32_t last_part_blk[2] = {0, 0};
    const int16_t *last_cic = nullptr, *last_cic2 = nullptr;
    unsigned char *scratch_base = nullptr; // this device's copy of the static work area
    unsigned long long *d_mom = nullptr;
    int16_t *d_cic = nullptr;
    int16_t *d_comp = nullptr;
    // work buffers of the two-kernel call plan (block counts that are a multiple of 256)
    uint32_t cap_fused_blk = 0;
    int16_t *d_cicf = nullptr;           // rx_cic outputs
    unsigned long long *d_edge = nullptr; // workgroup-edge moment records
    uint32_t *d_cnt = nullptr;            // arrival counters, zero between calls
    uint32_t *d_comp_tail = nullptr;      // last 256 compensator outputs of a call, on their way into the history
    size_t cnt_words = 0;
    int rx_plan = 0;                      // 0: by shape; 1: always the separate kernels; 2: by shape, vector-ALU ingest (ddc_debug_rx_plan)
    // history saved between calls (layout independent of the call size)
    unsigned long long *d_hist_mom = nullptr; // [ch][5][10]
    int16_t *d_hist_cic = nullptr;            // [ch][64][2]
    int16_t *d_hist_comp = nullptr;           // [ch][256][2]
    // staging for the host-pointer entry points
    uint8_t *d_in = nullptr;
    size_t in_cap = 0;
    int16_t *d_out = nullptr;
    size_t out_cap = 0;
    std::vector<uint32_t> h_phi;
    std::vector<uint8_t> tx_dir;       // per channel: last direction a command-1 frame set (0 = RX, the power-on state)
    // ADC_MIN / ADC_MAX registers of the status frame, gathered only when enabled
    bool adc_stats = false;
    int *d_adc_min = nullptr, *d_adc_max = nullptr;
    ddc_tx_state tx;
    // 48 kHz audio path / panadapter: per-channel state and FFT work buffers
    ddc::AudioState *d_audio = nullptr;
    float2 *d_audio_filt = nullptr;    // [ch][frames] {I, Q} behind the RX audio filters
    size_t audio_filt_cap = 0;         // frames per channel it holds
    ddc::ZoomState *d_zoom = nullptr;
    ddc::TxAudioState *d_audio_tx = nullptr;
    float *d_tx_xs = nullptr;          // [ch][200 + frames] gain-scaled TX audio ahead of the Hilbert FIR pair
    size_t tx_xs_cap = 0;              // frames per channel it holds
    uint32_t fft_zoom_all = 1;         // common zoom factor of all channels (0 = mixed)
    std::vector<uint8_t> audio_ready;  // per channel: ddc_audio_configure seen
    std::vector<uint8_t> zoom_of;      // per channel: configured FFT zoom
    std::vector<ddc_audio_params> audio_cfg; // per channel: what ddc_audio_configure was given
    float2 *d_fft_work = nullptr;      // [ch][I, Q][pitch] floats: prefiltered SPEC, planar
    float *d_fft_mag = nullptr;        // [ch][frames][256]
    uint16_t *d_fft_rows = nullptr;    // [ch][frames][256] colour rows when the caller wants none
    uint16_t *d_wtf = nullptr;         // [ch][50][256] waterfall, row 0 newest
    size_t fft_cap = 0;                // frames per channel d_fft_mag holds
    size_t fft_work_cap = 0;           // floats per channel and branch d_fft_work holds
    void *d_stage[3] = {nullptr, nullptr, nullptr}; // host-pointer entry points
    size_t stage_cap[3] = {0, 0, 0};
    // multi-GPU: the RCCL communicator of this rank and the all-gather of the decimated outputs, which runs on
    // a stream of its own, ordered against the ctx stream by events (two slots: double buffering)
    ncclComm_t comm = nullptr;
    bool comm_owned = false;
    uint32_t comm_world = 0, comm_rank = 0;
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_computed[2] = {}, ev_gather0[2] = {}, ev_gathered[2] = {};
    bool gather_pending[2] = {false, false};
    // optional per-call timing: HIP events on the ctx stream around the front
    // kernel and around the whole RX call
    bool profile = false;
    static constexpr int kProfRing = 256;
    hipEvent_t ev[kProfRing][3] = {};
    int prof_head = 0, prof_count = 0;
};

namespace {

thread_local std::string g_err;

int fail(ddc_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, -EIO, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

long rnd_half_away(double x) { return (long)(x >= 0 ? std::floor(x + 0.5) : -std::floor(-x + 0.5)); }

// {fine sine, 8191} as 16-bit halves for fine phase f, in integers: what the kernels' float multiply-add on a
// denormal (ddc::nco_fine_pair) must produce.  The host never evaluates the float form: a host thread that
// flushes denormals would get it wrong while the GPU is fine.
uint32_t nco_fine_pair_host(uint32_t f) { return (8191u << 16) | (uint32_t)((f * DDC_SF_K + DDC_SF_R) >> 24); }

// Coarse NCO table for the kernels and a check that the arithmetic fine-sine
// shortcut reproduces the 2048-entry fine ROM.
int build_nco_table(std::vector<uint2> &tab)
{
    const double two_pi = 6.283185307179586476925286766559;
    tab.resize(2048);
    for (int i = 0; i < 2048; i++) {
        const int sc = (int)rnd_half_away(8191.0 * std::sin(two_pi * i / 2048.0));
        const int cc = (int)rnd_half_away(8191.0 * std::cos(two_pi * i / 2048.0));
        tab[i].x = ddc::pack_w(2 * cc, 2 * sc);
        tab[i].y = ddc::pack_w(-2 * sc, 2 * cc);
    }
    for (uint32_t fi = 0; fi < 2048; fi++) {
        const long sf = rnd_half_away(8191.0 * std::sin(two_pi * fi / 4194304.0));
        const long cf = rnd_half_away(8191.0 * std::cos(two_pi * fi / 4194304.0));
        if ((long)((fi * DDC_SF_K + DDC_SF_R) >> 24) != sf || cf != 8191) return -1;
        if (nco_fine_pair_host(fi) != ((8191u << 16) | (uint32_t)sf)) return -1;
    }
    return 0;
}

int check_channel(ddc_ctx *ctx, int32_t ch)
{
    if (!ctx) return -EINVAL;
    if (ch < -1 || ch >= (int32_t)ctx->cfg.n_channels)
        return fail(ctx, -EINVAL, "channel out of range");
    return 0;
}

int ensure_capacity(ddc_ctx *ctx, uint32_t n_blk)
{
    if (n_blk <= ctx->cap_blk) return 0;
    const size_t nch = ctx->cfg.n_channels;
    if (ctx->d_mom) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_mom) (void)hipFree(ctx->d_mom);
    if (ctx->d_cic) (void)hipFree(ctx->d_cic);
    if (ctx->d_comp) (void)hipFree(ctx->d_comp);
    ctx->d_mom = nullptr; ctx->d_cic = nullptr; ctx->d_comp = nullptr; ctx->cap_blk = 0;
    ctx->last_cic = ctx->last_cic2 = nullptr; ctx->last_blk = 0; // the debug tap must not read a freed buffer if the call fails later
    HIP_TRY(ctx, hipMalloc(&ctx->d_mom, nch * (ddc::kMomHist + (size_t)n_blk) * 10 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_cic, nch * (ddc::kCicHist + (size_t)n_blk) * 2 * sizeof(int16_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_comp, nch * (ddc::kCompHist + (size_t)n_blk / 2) * 2 * sizeof(int16_t)));
    ctx->cap_blk = n_blk;
    return 0;
}

// Work buffers of the two-kernel call plan for n_blk blocks per channel (n_blk a multiple of 256).
struct FusedLayout { size_t cic, edge, tail, cnt_words, n_wg, n_tiles; };
FusedLayout fused_layout(size_t nch, size_t n_blk)
{
    FusedLayout f;
    f.n_wg = n_blk / ddc::kFrontThreads;
    f.n_tiles = (n_blk / 2 + ddc::kTailTile - 1) / ddc::kTailTile;
    f.cic = nch * (ddc::kCicHist + n_blk) * 2 * sizeof(int16_t);
    f.edge = nch * f.n_wg * ddc::kEdgeRecs * 10 * sizeof(unsigned long long);
    f.tail = nch * ddc::kCompHist * 2 * sizeof(int16_t);
    f.cnt_words = nch;
    return f;
}

int ensure_fused_capacity(ddc_ctx *ctx, uint32_t n_blk)
{
    if (n_blk <= ctx->cap_fused_blk) return 0;
    const FusedLayout f = fused_layout(ctx->cfg.n_channels, n_blk);
    if (ctx->d_cicf) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_cicf) (void)hipFree(ctx->d_cicf);
    if (ctx->d_edge) (void)hipFree(ctx->d_edge);
    if (ctx->d_cnt) (void)hipFree(ctx->d_cnt);
    if (ctx->d_comp_tail) (void)hipFree(ctx->d_comp_tail);
    ctx->d_comp_tail = nullptr;
    ctx->d_cicf = nullptr; ctx->d_edge = nullptr; ctx->d_cnt = nullptr; ctx->cap_fused_blk = 0; ctx->cnt_words = 0;
    ctx->last_cic = ctx->last_cic2 = nullptr; ctx->last_blk = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_cicf, f.cic));
    HIP_TRY(ctx, hipMalloc(&ctx->d_edge, f.edge));
    HIP_TRY(ctx, hipMalloc(&ctx->d_comp_tail, f.tail));
    // the counters of a smaller call sit at other offsets: every word a call may use starts at zero
    // and every call leaves its own at zero
    HIP_TRY(ctx, hipMalloc(&ctx->d_cnt, (f.cnt_words + 3) / 4 * 16));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_cnt, 0, (f.cnt_words + 3) / 4 * 16, ctx->stream));
    ctx->cnt_words = f.cnt_words;
    ctx->cap_fused_blk = n_blk;
    return 0;
}

// Zoom-FFT coefficient sets (STM32/Src/fft.c:76-186): 4 biquads {b0,b1,b2,a1,a2}
// (a's already negated) and a 4-tap decimation FIR per zoom factor 2, 4, 8, 16.
// The firmware writes them as double literals narrowed to float32.
const double kZoomBiquad[4][20] = {
    { 0.228454526413293696, 0.077639329099949764, 0.228454526413293696, 0.635534925142242080, -0.170083307068779194,
      0.436788292542003964, 0.232307972937606161, 0.436788292542003964, 0.365885230717786780, -0.471769788739400842,
      0.535974654742658707, 0.557035600464780845, 0.535974654742658707, 0.125740787233286133, -0.754725697183384336,
      0.501116342273565607, 0.914877831284765408, 0.501116342273565607, 0.013862536615004284, -0.930973052446900984 },
    { 0.182208761527446556, -0.222492493114674145, 0.182208761527446556, 1.326111070880959810, -0.468036100821178802,
      0.337123762652097259, -0.366352718812586853, 0.337123762652097259, 1.337053579516321200, -0.644948386007929031,
      0.336163175380826074, -0.199246162162897811, 0.336163175380826074, 1.354952684569386670, -0.828032873168141115,
      0.178588201750411041, 0.207271695028067304, 0.178588201750411041, 1.386486967455699220, -0.950935065984588657 },
    { 0.185643392652478922, -0.332064345389014803, 0.185643392652478922, 1.654637402827731090, -0.693859842743674182,
      0.327519300813245984, -0.571358085216950418, 0.327519300813245984, 1.715375037176782860, -0.799055553586324407,
      0.283656142708241688, -0.441088976843048652, 0.283656142708241688, 1.778230635987093860, -0.904453944560528522,
      0.079685368654848945, -0.011231810140649204, 0.079685368654848945, 1.825046003243238070, -0.973184930412286708 },
    { 0.194769868656866380, -0.379098413160710079, 0.194769868656866380, 1.824436402073870810, -0.834877726226893380,
      0.333973874901496770, -0.646106479315673776, 0.333973874901496770, 1.871892825636887640, -0.893734096124207178,
      0.272903880596429671, -0.513507745397738469, 0.272903880596429671, 1.918161772571113750, -0.950461788366234739,
      0.053535383722369843, -0.069683422367188122, 0.053535383722369843, 1.948900719896301760, -0.986288064973853129 } };
const double kZoomFir[4][4] = {
    { 475.1179397144384210E-6, 0.503905202786044337, 0.503905202786044337, 475.1179397144384210E-6 },
    { 0.198273254218889416, 0.298085149879260325, 0.298085149879260325, 0.198273254218889416 },
    { 0.199820836596682871, 0.272777397353925699, 0.272777397353925699, 0.199820836596682871 },
    { 0.199820836596682871, 0.272777397353925699, 0.272777397353925699, 0.199820836596682871 } };

} // namespace

// Work area of ddc_rx_process_dev_static: the window moments, rx_cic outputs and
// compensator outputs of one call (what ensure_capacity allocates for the other
// entry points).  One copy per GPU the library runs on, held until the process
// ends.  On the host this name is only a handle for hipGetSymbolAddress: the
// device address it yields (per device, kept in g_scratch_ptr) is what kernels,
// memsets and copies are given.
#define DDC_STATIC_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scratch[DDC_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// Address of this device's copy of the work area; `device` must be current.
int scratch_address(ddc_ctx *ctx, int device, unsigned char **out)
{
    if (device < 0 || device >= kMaxDevices) return fail(ctx, -EINVAL, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p % 256)) return fail(ctx, -EFAULT, "static work area has no usable device address");
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[device];
    return 0;
}

bool g_tables_ready[kMaxDevices] = {};
bool device_tables_ready(int device)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    return device >= 0 && device < kMaxDevices && g_tables_ready[device];
}
void device_tables_mark_ready(int device)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (device >= 0 && device < kMaxDevices) g_tables_ready[device] = true;
}

// Byte offsets of the work buffers of one call inside the static area, each 256-byte aligned.
// The area starts with the arrival counters of the two-kernel plan: they are zero between
// calls whatever plan or shape the calls before had, because nothing else is ever put there.
constexpr size_t kStaticCntBytes = 256u << 10;
struct WorkLayout { bool fused; size_t mom, cic, comp, edge, tail, cnt_words, total; };
bool plan_is_fused(int rx_plan, size_t n_blk) { return rx_plan != 1 && n_blk % ddc::kFrontThreads == 0; }
// (first: bytes behind the counters that an earlier part of the same call occupies)
WorkLayout work_layout(size_t nch, size_t n_blk, bool fused, size_t first = 0)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    WorkLayout w{};
    w.fused = fused;
    if (fused) {
        const FusedLayout f = fused_layout(nch, n_blk);
        w.cnt_words = f.cnt_words;
        w.cic = kStaticCntBytes + first;
        w.edge = w.cic + up(f.cic);
        w.tail = w.edge + up(f.edge);
        w.total = w.tail + up(f.tail);
        if (f.cnt_words * 4 > kStaticCntBytes) w.total = (size_t)-1; // more counters than the reserved block holds
        return w;
    }
    w.mom = kStaticCntBytes + first;
    w.cic = w.mom + up(nch * (ddc::kMomHist + n_blk) * 10 * sizeof(unsigned long long));
    w.comp = w.cic + up(nch * (ddc::kCicHist + n_blk) * 2 * sizeof(int16_t));
    w.total = w.comp + up(nch * (ddc::kCompHist + n_blk / 2) * 2 * sizeof(int16_t));
    return w;
}

// Device buffers one RX call works in (ctx-owned or carved from the static area).
struct RxWork { unsigned long long *mom; int16_t *cic; int16_t *comp; unsigned long long *edge; uint32_t *tail; uint32_t *cnt; };
} // namespace

namespace ddc {
// history <-> work buffer moves: the work buffers are laid out for the block
// count of the current call, the saved history is not.
__global__ void ddc_rx_load_history(const unsigned long long *hm, const int16_t *hc, const int16_t *he,
                                    unsigned long long *mom, int16_t *cic, int16_t *comp, uint32_t n_blk,
                                    const uint32_t *carry_saved, uint32_t *carry_work)
{
    const uint32_t ch = blockIdx.x, t = threadIdx.x, nf = n_blk / 2;
    // the ingest kernel reads the previous call's carries from the copy while its
    // last block of the channel overwrites the saved ones
    if (t < 6) carry_work[ch * 6 + t] = carry_saved[ch * 6 + t];
    if (t < kMomHist * 10) mom[(size_t)ch * (kMomHist + n_blk) * 10 + t] = hm[(size_t)ch * kMomHist * 10 + t];
    if (t < kCicHist)
        reinterpret_cast<uint32_t *>(cic)[(size_t)ch * (kCicHist + n_blk) + t] =
            reinterpret_cast<const uint32_t *>(hc)[(size_t)ch * kCicHist + t];
    if (t < kCompHist)
        reinterpret_cast<uint32_t *>(comp)[(size_t)ch * (kCompHist + nf) + t] =
            reinterpret_cast<const uint32_t *>(he)[(size_t)ch * kCompHist + t];
}

__global__ void ddc_rx_save_history(unsigned long long *hm, int16_t *hc, int16_t *he,
                                    const unsigned long long *mom, const int16_t *cic, const int16_t *comp,
                                    uint32_t *acc0, const uint32_t *phi32v, uint32_t n_blk, uint32_t samples_lo)
{
    const uint32_t ch = blockIdx.x, t = threadIdx.x, nf = n_blk / 2;
    if (t < kMomHist * 10)
        hm[(size_t)ch * kMomHist * 10 + t] = mom[((size_t)ch * (kMomHist + n_blk) + n_blk) * 10 + t];
    if (t < kCicHist)
        reinterpret_cast<uint32_t *>(hc)[(size_t)ch * kCicHist + t] =
            reinterpret_cast<const uint32_t *>(cic)[(size_t)ch * (kCicHist + n_blk) + n_blk + t];
    if (t < kCompHist)
        reinterpret_cast<uint32_t *>(he)[(size_t)ch * kCompHist + t] =
            reinterpret_cast<const uint32_t *>(comp)[(size_t)ch * (kCompHist + nf) + nf + t];
    if (t == 0) acc0[ch] += samples_lo * phi32v[ch];
}

// The device's own evaluation of the fine-sine multiply-add for every fine phase
// (a device or build that flushes denormals would get it wrong silently).
__global__ void ddc_nco_fine_selfcheck(uint32_t *out)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < 2048) {
        out[f] = nco_fine_pair(f);
        uint32_t w0, w1; // the packed form, both of its halves
        nco_fine_pair2(f, f, w0, w1);
        out[2048 + f] = w0 == w1 ? w0 : ~0u;
    }
}

// The kernels' NCO (coarse table in LDS, fine sine, angle sums) evaluated for a run
// of consecutive 22-bit phases, in both forms the kernels use: the TX form (14-bit
// sine and cosine) and the RX form (the 12-bit mixer operands).
__global__ __launch_bounds__(256) void ddc_nco_sweep(const uint2 *__restrict__ nco_tab, uint32_t first_phase,
                                                     uint32_t count, int16_t *__restrict__ out)
{
    __shared__ uint2 tab[2048], tab2[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) {
        tab[i] = nco_tab[i];
        tab2[i] = nco_entry_doubled(nco_tab[i]);
    }
    __syncthreads();
    // two consecutive phases per thread, through the pair forms the sample loops use
    for (uint32_t i = 2 * (blockIdx.x * blockDim.x + threadIdx.x); i < count; i += 2 * gridDim.x * blockDim.x) {
        const uint32_t a0 = ((first_phase + i) & 0x3FFFFFu) << 8, a1 = ((first_phase + i + 1) & 0x3FFFFFu) << 8;
        const unsigned long long php = (unsigned long long)a0 | ((unsigned long long)a1 << 32);
        int s[2], c[2];
        nco14x_pair(tab2, php, s, c);
        // RX: the mixer multiplies the ADC sample by the 12-bit operands and returns the products
        uint2 e[2];
        uint32_t w[2];
        nco_fetch_pair(tab, php, e[0], e[1], w[0], w[1]);
        for (uint32_t k = 0; k < 2 && i + k < count; k++) {
            int pi, pq;
            nco_finish(e[k], w[k], 1, pi, pq);
            out[i + k] = (int16_t)(s[k] >> 16);
            out[(size_t)count + i + k] = (int16_t)(c[k] >> 16);
            out[2 * (size_t)count + i + k] = (int16_t)pi;
            out[3 * (size_t)count + i + k] = (int16_t)pq;
        }
    }
}

// The synthetic bench signal of ddc_synth_adc on the device.  The noise generator
// (xorshift64*) is a serial recurrence per channel: pass 1 walks it once per channel
// and keeps the state at the start of every 512-sample segment, pass 2 fills the
// segments in parallel (tones in double precision, as the host version).
constexpr uint32_t kSynthSeg = 512;

__global__ void ddc_synth_states(uint32_t first_channel, uint32_t n_channels, uint32_t n_seg, uint64_t *states)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) return;
    uint64_t s = 0x5EED0000ull + first_channel + c;
    for (uint32_t g = 0; g < n_seg; g++) {
        states[(size_t)c * n_seg + g] = s;
        for (uint32_t i = 0; i < kSynthSeg; i++) { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; }
    }
}

__global__ void ddc_synth_rows(uint32_t first_channel, uint32_t n_channels, uint32_t n_seg, double fs_adc,
                               const uint64_t *__restrict__ states, uint8_t *__restrict__ rows, size_t row_bytes)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n_channels * n_seg) return;
    const uint32_t c = (uint32_t)(gid / n_seg), g = (uint32_t)(gid - (uint64_t)c * n_seg);
    uint64_t s = states[gid];
    const double two_pi = 6.283185307179586476925286766559;
    const double f1 = 1.0e6 + 48.0e3 * (double)(first_channel + c), f2 = f1 + 1.0e3;
    const double w1 = two_pi * f1 / fs_adc, w2 = two_pi * f2 / fs_adc;
    uint8_t *out = rows + (size_t)c * row_bytes + (size_t)g * (kSynthSeg / 2 * 3);
    int pair[2];
    for (uint32_t i = 0; i < kSynthSeg; i++) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const uint64_t r = s * 0x2545F4914F6CDD1Dull;
        const int noise = (int)(r >> 57) - 64;
        const double t = (double)((uint64_t)g * kSynthSeg + i);
        long v = lround(1400.0 * sin(w1 * t) + 300.0 * sin(w2 * t)) + noise;
        v = v > 2047 ? 2047 : (v < -2047 ? -2047 : v);
        pair[i & 1] = (int)v;
        if (i & 1) {
            const unsigned s0 = (unsigned)pair[0] & 0xFFF, s1 = (unsigned)pair[1] & 0xFFF;
            uint8_t *b = out + (i / 2) * 3;
            b[0] = (uint8_t)(s0 & 0xFF);
            b[1] = (uint8_t)((s0 >> 8) | ((s1 & 0xF) << 4));
            b[2] = (uint8_t)(s1 >> 4);
        }
    }
}
} // namespace ddc

extern "C" {

const char *ddc_version(void) { return "ua3reo-ddc-amd 0.1 (gfx950)"; }

const char *ddc_last_error(const ddc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int ddc_create(const ddc_config *cfg, ddc_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, -EINVAL, "null argument");
    *out = nullptr;
    if (cfg->n_channels == 0 || cfg->n_channels > (1u << 20))
        return fail(nullptr, -EINVAL, "n_channels must be 1..2^20");
    ddc_ctx *ctx = new ddc_ctx;
    ctx->cfg = *cfg;
    if (ctx->cfg.fs_adc == 0) ctx->cfg.fs_adc = DDC_ADC_CLOCK_DEFAULT;
    auto bail = [&](int code) { std::string m = ctx->err; ddc_destroy(ctx); g_err = m; return code; };
#define CREATE_TRY(call)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return bail(-EIO);                                                               \
        }                                                                                    \
    } while (0)
    CREATE_TRY(hipSetDevice(cfg->device));
    // the __constant__ tables are the same for every ctx: written once per device, by the first
    // ctx created on it (a later ctx must not rewrite them under a running kernel of an earlier one)
    const bool upload_tables = !device_tables_ready(cfg->device);
#define CONST_TRY(call) do { if (upload_tables) CREATE_TRY(call); } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    if (scratch_address(ctx, cfg->device, &ctx->scratch_base) != 0) return bail(-EIO);
    const size_t nch = cfg->n_channels;
    CREATE_TRY(hipMalloc(&ctx->d_acc, nch * 4));
    CREATE_TRY(hipMalloc(&ctx->d_phi, nch * 4));
    CREATE_TRY(hipMalloc(&ctx->d_carry[0], nch * 24));
    CREATE_TRY(hipMalloc(&ctx->d_carry[1], nch * 24));
    CREATE_TRY(hipMalloc(&ctx->d_nco_tab, 2048 * sizeof(uint2)));
    CREATE_TRY(hipMalloc(&ctx->d_hist_mom, nch * ddc::kMomHist * 10 * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(&ctx->d_hist_cic, nch * ddc::kCicHist * 4));
    CREATE_TRY(hipMalloc(&ctx->d_hist_comp, nch * ddc::kCompHist * 4));
    CREATE_TRY(hipMemsetAsync(ctx->d_acc, 0, nch * 4, ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_phi, 0, nch * 4, ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_carry[0], 0, nch * 24, ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_carry[1], 0, nch * 24, ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_hist_mom, 0, nch * ddc::kMomHist * 10 * sizeof(unsigned long long), ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_hist_cic, 0, nch * ddc::kCicHist * 4, ctx->stream));
    CREATE_TRY(hipMemsetAsync(ctx->d_hist_comp, 0, nch * ddc::kCompHist * 4, ctx->stream));
    std::vector<uint2> tab;
    if (build_nco_table(tab) != 0) {
        ctx->err = "fine-sine shortcut does not reproduce the NCO fine ROM";
        return bail(-EFAULT);
    }
    {
        // the kernels compute the fine sine with a float multiply-add on a denormal: have the
        // device evaluate all 2048 phases once and compare with the host's (d_nco_tab is free until
        // the table is uploaded just below)
        std::vector<uint32_t> got(4096);
        ddc::ddc_nco_fine_selfcheck<<<8, 256, 0, ctx->stream>>>(reinterpret_cast<uint32_t *>(ctx->d_nco_tab));
        CREATE_TRY(hipGetLastError());
        CREATE_TRY(hipMemcpyAsync(got.data(), ctx->d_nco_tab, 4096 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        CREATE_TRY(hipStreamSynchronize(ctx->stream));
        for (uint32_t f = 0; f < 2048; f++)
            if (got[f] != nco_fine_pair_host(f) || got[2048 + f] != got[f]) {
                ctx->err = "this device evaluates the NCO fine sine differently from the host (denormals flushed?)";
                return bail(-EFAULT);
            }
    }
    CREATE_TRY(hipMemcpyAsync(ctx->d_nco_tab, tab.data(), 2048 * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint8_t> mx(ddc::kMxTabBytes);
    if (!ddc::build_mx_table(mx.data())) {
        ctx->err = "a window-moment weight does not fit its byte planes";
        return bail(-EFAULT);
    }
    CREATE_TRY(hipMalloc(&ctx->d_mx_tab, ddc::kMxTabBytes));
    CREATE_TRY(hipMemcpyAsync(ctx->d_mx_tab, mx.data(), ddc::kMxTabBytes, hipMemcpyHostToDevice, ctx->stream));
    CONST_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(ddc::c_rx_ciccomp_h), rx_ciccomp_h, sizeof rx_ciccomp_h, 0,
                                      hipMemcpyHostToDevice, ctx->stream));
    CONST_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(ddc::c_rx_hilb_c), rx_hilb_c, sizeof rx_hilb_c, 0,
                                      hipMemcpyHostToDevice, ctx->stream));
    {
        // compensator taps as the kernel multiplies them: pair m covers samples 2n - 64 + 2m, + 1
        uint32_t cp[33];
        for (int m = 0; m < 32; m++)
            cp[m] = (uint32_t)(uint16_t)rx_ciccomp_h[64 - 2 * m] | ((uint32_t)(uint16_t)rx_ciccomp_h[63 - 2 * m] << 16);
        cp[32] = (uint32_t)(uint16_t)rx_ciccomp_h[0];
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_ciccomp_pairs), cp, sizeof cp));
    }
    {
        // Hilbert taps as the kernel wants them: reversed pairs and the two low bit planes
        uint32_t pairs[128], odd[8] = {0}, bit1[8] = {0};
        for (int m = 0; m < 128; m++)
            pairs[m] = (uint32_t)(uint16_t)rx_hilb_c[255 - 2 * m] | ((uint32_t)(uint16_t)rx_hilb_c[254 - 2 * m] << 16);
        for (int j = 0; j < 256; j++) {
            const unsigned c = (unsigned)(int)rx_hilb_c[255 - j];
            odd[j >> 5] |= (c & 1u) << (j & 31);
            bit1[j >> 5] |= ((c >> 1) & 1u) << (j & 31);
        }
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_hilb_pairs), pairs, sizeof pairs));
        uint32_t padded[144] = {0};
        memcpy(padded + 3, pairs, sizeof pairs);
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_hilb_pairs_pad), padded, sizeof padded));
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_hilb_odd), odd, sizeof odd));
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_hilb_bit1), bit1, sizeof bit1));
    }
    ctx->h_phi.assign(nch, 0);
    ctx->tx_dir.assign(nch, 0);
    {
        // power-on values of the registers (FPGA/stm32_interface.v:385-388)
        std::vector<int> lo(nch, 2000), hi(nch, -2000);
        CREATE_TRY(hipMalloc(&ctx->d_adc_min, nch * 4));
        CREATE_TRY(hipMalloc(&ctx->d_adc_max, nch * 4));
        CREATE_TRY(hipMemcpy(ctx->d_adc_min, lo.data(), nch * 4, hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(ctx->d_adc_max, hi.data(), nch * 4, hipMemcpyHostToDevice));
    }
    ctx->audio_ready.assign(nch, 0);
    ctx->zoom_of.assign(nch, 1);
    ctx->audio_cfg.assign(nch, ddc_audio_params{});
    CREATE_TRY(hipMalloc(&ctx->d_audio, nch * sizeof(ddc::AudioState)));
    CREATE_TRY(hipMemsetAsync(ctx->d_audio, 0, nch * sizeof(ddc::AudioState), ctx->stream));
    CREATE_TRY(hipMalloc(&ctx->d_wtf, nch * 50 * 256 * sizeof(uint16_t)));
    CREATE_TRY(hipMemsetAsync(ctx->d_wtf, 0, nch * 50 * 256 * sizeof(uint16_t), ctx->stream));
    CREATE_TRY(hipMalloc(&ctx->d_zoom, nch * sizeof(ddc::ZoomState)));
    CREATE_TRY(hipMemsetAsync(ctx->d_zoom, 0, nch * sizeof(ddc::ZoomState), ctx->stream));
    CREATE_TRY(hipMalloc(&ctx->d_audio_tx, nch * sizeof(ddc::TxAudioState)));
    CREATE_TRY(hipMemsetAsync(ctx->d_audio_tx, 0, nch * sizeof(ddc::TxAudioState), ctx->stream));
    CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_tx_hilbert), tx_hilbert_i, sizeof tx_hilbert_i, 0));
    CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_tx_hilbert), tx_hilbert_q, sizeof tx_hilbert_q, sizeof tx_hilbert_i));
    {
        // Hamming window as the firmware evaluates it (fft.c:274-281, with CMSIS's table cosine) and the
        // 512-point twiddles; the sine table itself for the FM modulator
        std::vector<float> win(512);
        std::vector<float2> tw(256);
        for (int i = 0; i < 512; i++)
            win[i] = 0.54f - 0.46f * ddc::arm_cos_with(arm_sin_table(), (2.0f * ddc::kPiF * i) / (512.0f - 1.0f));
        for (int k = 0; k < 256; k++) {
            const double a = -2.0 * 3.14159265358979323846 * k / 512.0;
            tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::g_arm_sin_table), arm_sin_table().data(), 513 * sizeof(float)));
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_fft_window), win.data(), 512 * sizeof(float)));
        CONST_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ddc::c_fft_twiddle), tw.data(), 256 * sizeof(float2)));
    }
    {
        int rc = ddc_tx_state_create(&ctx->tx, ctx->cfg.n_channels, ctx->stream, upload_tables);
        if (rc != 0) { ctx->err = "TX state allocation failed"; return bail(rc); }
    }
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
#undef CONST_TRY
#undef CREATE_TRY
    if (upload_tables) device_tables_mark_ready(cfg->device);
    *out = ctx;
    return 0;
}

void ddc_destroy(ddc_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *ptrs[] = { ctx->d_acc, ctx->d_phi, ctx->d_carry[0], ctx->d_carry[1], ctx->d_nco_tab, ctx->d_mom,
                     ctx->d_cic, ctx->d_comp, ctx->d_hist_mom, ctx->d_hist_cic, ctx->d_hist_comp, ctx->d_in,
                     ctx->d_out, ctx->d_tx_xs, ctx->d_fft_rows, ctx->d_wtf, ctx->d_adc_min, ctx->d_adc_max, ctx->d_audio, ctx->d_zoom, ctx->d_audio_tx, ctx->d_fft_work, ctx->d_fft_mag, ctx->d_stage[0], ctx->d_stage[1],
                     ctx->d_stage[2], ctx->d_cicf, ctx->d_edge, ctx->d_cnt, ctx->d_comp_tail, ctx->d_audio_filt, ctx->d_mx_tab };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    ddc_tx_state_destroy(&ctx->tx);
    (void)ddc_comm_destroy(ctx);
    for (auto &e3 : ctx->ev)
        for (auto &e : e3)
            if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

uint32_t ddc_phrase_from_frequency(uint32_t freq_hz, uint32_t fs_adc, int *iq_swap)
{
    if (fs_adc == 0) fs_adc = DDC_ADC_CLOCK_DEFAULT;
    bool inverted = false;
    uint32_t f = freq_hz;
    const uint32_t nyq = fs_adc / 2;
    if (f > nyq) {
        while (f > nyq) { f -= nyq; inverted = !inverted; }
        if (inverted) f = nyq - f;
    }
    if (iq_swap) *iq_swap = inverted ? 1 : 0;
    return (uint32_t)std::round(((double)f / (double)fs_adc) * 4194304.0);
}

uint32_t ddc_frequency_from_phrase(uint32_t phrase, uint32_t fs_adc)
{
    if (fs_adc == 0) fs_adc = DDC_ADC_CLOCK_DEFAULT;
    return (uint32_t)(std::ceil(((double)phrase / 4194304.0) * (double)fs_adc / 100.0) * 100.0);
}

int ddc_tune(ddc_ctx *ctx, int32_t channel, uint32_t phi_inc22)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (phi_inc22 >> 22) return fail(ctx, -EINVAL, "tuning word wider than 22 bits");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const uint32_t lo = channel < 0 ? 0 : (uint32_t)channel;
    const uint32_t hi = channel < 0 ? ctx->cfg.n_channels : (uint32_t)channel + 1;
    for (uint32_t c = lo; c < hi; c++) ctx->h_phi[c] = phi_inc22 << 10;
    // the staging vector must outlive the async copy: wait for it
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_phi + lo, ctx->h_phi.data() + lo, (hi - lo) * 4, hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_tune_many(ddc_ctx *ctx, const uint32_t *phi_inc22, uint32_t n)
{
    if (!ctx || !phi_inc22) return -EINVAL;
    if (n != ctx->cfg.n_channels) return fail(ctx, -EINVAL, "ddc_tune_many: need one word per channel");
    for (uint32_t c = 0; c < n; c++) {
        if (phi_inc22[c] >> 22) return fail(ctx, -EINVAL, "tuning word wider than 22 bits");
        ctx->h_phi[c] = phi_inc22[c] << 10;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_phi, ctx->h_phi.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_tune_hz(ddc_ctx *ctx, int32_t channel, uint32_t freq_hz, int *iq_swap)
{
    if (!ctx) return -EINVAL;
    return ddc_tune(ctx, channel, ddc_phrase_from_frequency(freq_hz, ctx->cfg.fs_adc, iq_swap) & 0x3FFFFF);
}

static int rx_reset_range(ddc_ctx *ctx, size_t lo, size_t n, bool all);

int ddc_rx_reset(ddc_ctx *ctx, int32_t channel)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    const size_t lo = channel < 0 ? 0 : (size_t)channel;
    const size_t n = channel < 0 ? ctx->cfg.n_channels : 1;
    return rx_reset_range(ctx, lo, n, channel < 0);
}

static int rx_reset_range(ddc_ctx *ctx, size_t lo, size_t n, bool all)
{
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc + lo, 0, n * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_carry[0] + lo * 6, 0, n * 24, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_carry[1] + lo * 6, 0, n * 24, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_hist_mom + lo * ddc::kMomHist * 10, 0,
                                n * ddc::kMomHist * 10 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_hist_cic + lo * ddc::kCicHist * 2, 0, n * ddc::kCicHist * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_hist_comp + lo * ddc::kCompHist * 2, 0, n * ddc::kCompHist * 4, ctx->stream));
    if (all && ctx->d_cnt) HIP_TRY(ctx, hipMemsetAsync(ctx->d_cnt, 0, (ctx->cnt_words + 3) / 4 * 16, ctx->stream));
    return 0;
}

int ddc_rx_set_phase(ddc_ctx *ctx, int32_t channel, uint32_t phase22)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0) return fail(ctx, -EINVAL, "ddc_rx_set_phase needs one channel");
    const uint32_t v = (phase22 & 0x3FFFFF) << 10;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_acc + channel, &v, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_rx_get_phase(ddc_ctx *ctx, int32_t channel, uint32_t *phase22)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !phase22) return fail(ctx, -EINVAL, "ddc_rx_get_phase needs one channel");
    uint32_t v = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(&v, ctx->d_acc + channel, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *phase22 = v >> 10;
    return 0;
}

// Argument checks shared by the RX entry points; fills the block counts.
static int rx_check_args(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples, const int16_t *spec_iq,
                         const int16_t *voice_iq, uint32_t *n_blk_out)
{
    if (!adc12 || !spec_iq || !voice_iq) return fail(ctx, -EINVAL, "null buffer");
    if (samples % DDC_DECIMATION) return fail(ctx, -EINVAL, "samples must be a multiple of 1024");
    if (row_bytes % 16 || ((uintptr_t)adc12 % 16)) return fail(ctx, -EINVAL, "ADC rows must be 16-byte aligned");
    // row_bytes = 0: one ADC stream shared by every channel (all rows are the same row)
    if (row_bytes && row_bytes < samples / 2 * 3) return fail(ctx, -EINVAL, "row_bytes smaller than one row of samples");
    const uint64_t n_blk64 = samples / 512;
    const uint64_t total64 = n_blk64 * ctx->cfg.n_channels;
    if (total64 >= (1ull << 31)) return fail(ctx, -EINVAL, "more than 2^31 blocks of 512 samples in one call");
    if (ctx->adc_stats && ctx->cfg.n_channels > 65535u)
        return fail(ctx, -EINVAL, "ADC statistics are limited to 65535 channels per ctx");
    *n_blk_out = (uint32_t)n_blk64;
    return 0;
}

// Queues one RX call on the ctx stream with the given work buffers.  pev: three
// events to record around the ingest kernel and the rest, or null.
//
// Two call plans, same arithmetic: block counts that are a multiple of 256 run as two
// kernels (ddc_rx_front_mx, or ddc_rx_front_cic under plan 2: a workgroup owns 256 windows of
// one channel and combines them into rx_cic outputs itself; ddc_rx_tail: the 48 kHz stages and
// the state hand-over, one workgroup per 1024 frames); any other block count runs the six
// separate kernels.  rx_split cuts a larger call of any size into a part of the first kind and a
// remainder of the second: the carried state is common to both plans, so the two parts are two
// calls on one stream of samples.
// out_pitch: frames per channel row of spec_iq / voice_iq (a call that runs in two parts writes both
// into the rows of the whole call).
static int rx_launch(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples, uint32_t n_blk,
                     int16_t *spec_iq, int16_t *voice_iq, uint32_t out_pitch, bool fused, const RxWork &w, hipEvent_t *pev)
{
    const uint32_t nch = ctx->cfg.n_channels, total = n_blk * nch, nf = n_blk / 2;
    hipStream_t st = ctx->stream;
    if (fused) {
        ddc::RxFused fa{};
        const FusedLayout f = fused_layout(nch, n_blk);
        fa.cic = w.cic; fa.edge = w.edge; fa.comp_tail = w.tail;
        fa.chan_cnt = w.cnt;
        fa.hist_mom = ctx->d_hist_mom; fa.hist_cic = ctx->d_hist_cic; fa.hist_comp = ctx->d_hist_comp;
        fa.carry_saved = ctx->d_carry[0]; fa.carry_work = ctx->d_carry[1]; fa.acc0 = ctx->d_acc;
        fa.phi32v = ctx->d_phi;
        fa.spec_iq = spec_iq; fa.voice_iq = voice_iq;
        fa.n_wg = (uint32_t)f.n_wg; fa.n_tiles = (uint32_t)f.n_tiles; fa.samples_lo = (uint32_t)samples;
        fa.out_pitch = out_pitch;
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[0], st));
        if (ctx->rx_plan == 2)
            ddc::ddc_rx_front_cic<<<total / ddc::kFrontThreads, ddc::kFrontThreads, 0, st>>>(adc12, row_bytes, n_blk, total,
                                                                                           ctx->d_phi, ctx->d_nco_tab, fa);
        else
        {
            // x: the channel's workgroups, (y, z): the channel
            const uint32_t gy = std::min<uint32_t>(nch, 32768u), gz = (nch + gy - 1) / gy;
            ddc::ddc_rx_front_mx<<<dim3(fa.n_wg, gy, gz), ddc::kFrontThreads, 0, st>>>(adc12, row_bytes, n_blk, nch, ctx->d_phi,
                                                                                     ctx->d_nco_tab, ctx->d_mx_tab, fa);
        }
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[1], st));
        ddc::ddc_rx_tail<<<nch * fa.n_tiles, 256, 0, st>>>(n_blk, fa);
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[2], st));
    } else {
        ddc::ddc_rx_load_history<<<nch, 256, 0, st>>>(ctx->d_hist_mom, ctx->d_hist_cic, ctx->d_hist_comp, w.mom, w.cic,
                                                     w.comp, n_blk, ctx->d_carry[0], ctx->d_carry[1]);
        const uint32_t *cin = ctx->d_carry[1];
        uint32_t *cout = ctx->d_carry[0];
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[0], st));
        ddc::ddc_rx_front<<<(total + ddc::kFrontThreads - 1) / ddc::kFrontThreads, ddc::kFrontThreads, 0, st>>>(
            adc12, row_bytes, n_blk, total, ctx->d_acc, ctx->d_phi, cin, cout, w.mom, ctx->d_nco_tab);
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[1], st));
        const uint32_t ctiles = (n_blk + 255) / 256;
        ddc::ddc_rx_cic_combine<<<nch * ctiles, 256, 0, st>>>(w.mom, n_blk, ctiles, w.cic);
        const uint32_t tiles = (nf + ddc::kTailTile - 1) / ddc::kTailTile;
        ddc::ddc_rx_ciccomp<<<nch * tiles, 256, 0, st>>>(w.cic, n_blk, tiles, w.comp, spec_iq, out_pitch);
        ddc::ddc_rx_hilbert<<<nch * tiles, 256, 0, st>>>(w.comp, nf, tiles, voice_iq, out_pitch);
        ddc::ddc_rx_save_history<<<nch, 256, 0, st>>>(ctx->d_hist_mom, ctx->d_hist_cic, ctx->d_hist_comp, w.mom, w.cic,
                                                     w.comp, ctx->d_acc, ctx->d_phi, n_blk, (uint32_t)samples);
        if (pev) HIP_TRY(ctx, hipEventRecord(pev[2], st));
    }
    if (ctx->adc_stats) {
        const uint32_t pieces = (uint32_t)(samples / 32);
        const uint32_t gx = std::min<uint32_t>((pieces + 255) / 256, std::max<uint32_t>(1, 4096 / nch));
        ddc::ddc_adc_minmax<<<dim3(gx, nch), 256, 0, st>>>(adc12, row_bytes, pieces, ctx->d_adc_min, ctx->d_adc_max);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, -EIO, std::string("RX launch: ") + hipGetErrorString(e));
    return 0;
}

// The parts a call of n_blk blocks runs as: one, or (any size above 256 blocks that is not a multiple
// of 256, unless plan 1 is forced) the largest multiple of 256 blocks as two kernels and the rest as six.
struct RxPart { uint32_t n_blk; bool fused; };
static int rx_split(int rx_plan, uint32_t n_blk, RxPart part[2])
{
    if (rx_plan != 1 && n_blk > (uint32_t)ddc::kFrontThreads && n_blk % ddc::kFrontThreads) {
        const uint32_t main_blk = n_blk / ddc::kFrontThreads * ddc::kFrontThreads;
        part[0] = RxPart{ main_blk, true };
        part[1] = RxPart{ n_blk - main_blk, false };
        return 2;
    }
    part[0] = RxPart{ n_blk, plan_is_fused(rx_plan, n_blk) };
    part[1] = RxPart{ 0, false };
    return 1;
}

// Queues the parts of one call, each on its stretch of the input rows and of the output rows.
static int rx_run(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, uint32_t n_blk, int16_t *spec_iq, int16_t *voice_iq,
                  const RxPart *part, int n_parts, const RxWork *w, hipEvent_t *pev)
{
    size_t blk0 = 0;
    for (int k = 0; k < n_parts; k++) {
        // blk0 blocks = blk0 / 2 frames of two int16 each
        const int rc = rx_launch(ctx, adc12 + blk0 * ddc::kBlkBytes, row_bytes, (size_t)part[k].n_blk * 512, part[k].n_blk,
                                 spec_iq + blk0, voice_iq + blk0, n_blk / 2, part[k].fused, w[k], k == 0 ? pev : nullptr);
        if (rc) return rc;
        blk0 += part[k].n_blk;
    }
    if (pev && n_parts > 1) HIP_TRY(ctx, hipEventRecord(pev[2], ctx->stream)); // the rest of the call ends behind the last part
    ctx->last_blk = n_blk;
    ctx->last_part_blk[0] = part[0].n_blk;
    ctx->last_part_blk[1] = n_parts > 1 ? part[1].n_blk : 0;
    ctx->last_cic = w[0].cic;
    ctx->last_cic2 = n_parts > 1 ? w[1].cic : nullptr;
    return 0;
}

int ddc_rx_process_dev(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                       int16_t *spec_iq, int16_t *voice_iq)
{
    if (!ctx) return -EINVAL;
    if (samples == 0) return 0;
    uint32_t n_blk = 0;
    int rc = rx_check_args(ctx, adc12, row_bytes, samples, spec_iq, voice_iq, &n_blk);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    RxPart part[2];
    const int n_parts = rx_split(ctx->rx_plan, n_blk, part);
    for (int k = 0; k < n_parts; k++) {
        rc = part[k].fused ? ensure_fused_capacity(ctx, part[k].n_blk) : ensure_capacity(ctx, part[k].n_blk);
        if (rc) return rc;
    }
    hipEvent_t *pev = nullptr;
    if (ctx->profile) {
        pev = ctx->ev[ctx->prof_head];
        if (!pev[0])
            for (int k = 0; k < 3; k++) HIP_TRY(ctx, hipEventCreate(&pev[k]));
        ctx->prof_head = (ctx->prof_head + 1) % ddc_ctx::kProfRing;
        if (ctx->prof_count < ddc_ctx::kProfRing) ctx->prof_count++;
    }
    RxWork w[2];
    for (int k = 0; k < n_parts; k++)
        w[k] = part[k].fused ? RxWork{ nullptr, ctx->d_cicf, nullptr, ctx->d_edge, ctx->d_comp_tail, ctx->d_cnt }
                             : RxWork{ ctx->d_mom, ctx->d_cic, ctx->d_comp, nullptr, nullptr, nullptr };
    return rx_run(ctx, adc12, row_bytes, n_blk, spec_iq, voice_iq, part, n_parts, w, pev);
}

size_t ddc_rx_static_scratch_bytes(void) { return DDC_STATIC_SCRATCH_BYTES; }

// Work areas of the parts of a call inside the static area, one behind the other (the arrival
// counters at the start of the area are only ever the two-kernel part's).
static size_t static_layout(int rx_plan, uint32_t n_channels, uint32_t n_blk, RxPart part[2], int *n_parts, WorkLayout lay[2])
{
    *n_parts = rx_split(rx_plan, n_blk, part);
    size_t off = 0;
    for (int k = 0; k < *n_parts; k++) {
        lay[k] = work_layout(n_channels, part[k].n_blk, part[k].fused, off);
        if (lay[k].total == (size_t)-1) return (size_t)-1;
        off = lay[k].total - kStaticCntBytes;
    }
    return lay[*n_parts - 1].total;
}

size_t ddc_rx_static_scratch_need(uint32_t n_channels, size_t samples)
{
    RxPart part[2];
    WorkLayout lay[2];
    int n_parts = 0;
    return static_layout(0, n_channels, (uint32_t)(samples / 512), part, &n_parts, lay);
}

int ddc_rx_process_dev_static(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                              int16_t *spec_iq, int16_t *voice_iq)
{
    if (!ctx) return -EINVAL;
    if (samples == 0) return 0;
    uint32_t n_blk = 0;
    int rc = rx_check_args(ctx, adc12, row_bytes, samples, spec_iq, voice_iq, &n_blk);
    if (rc) return rc;
    RxPart part[2];
    WorkLayout lay[2];
    int n_parts = 0;
    const size_t need = static_layout(ctx->rx_plan, ctx->cfg.n_channels, n_blk, part, &n_parts, lay);
    if (need > DDC_STATIC_SCRATCH_BYTES) {
        char msg[200];
        if (need == (size_t)-1)
            snprintf(msg, sizeof msg, "ddc_rx_process_dev_static: %u channels x %zu samples need more arrival counters "
                     "than the static work area reserves (%zu bytes)", ctx->cfg.n_channels, samples, kStaticCntBytes);
        else
            snprintf(msg, sizeof msg, "ddc_rx_process_dev_static: %u channels x %zu samples need %zu bytes of work area, "
                     "the static one holds %u", ctx->cfg.n_channels, samples, need, DDC_STATIC_SCRATCH_BYTES);
        return fail(ctx, -E2BIG, msg);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    unsigned char *base = ctx->scratch_base; // resolved by ddc_create on this ctx's device
    if (!base) return fail(ctx, -EFAULT, "static work area was not resolved when the ctx was created");
    RxWork rw[2];
    for (int k = 0; k < n_parts; k++) {
        const WorkLayout &l = lay[k];
        rw[k] = part[k].fused ? RxWork{ nullptr, reinterpret_cast<int16_t *>(base + l.cic), nullptr,
                                        reinterpret_cast<unsigned long long *>(base + l.edge),
                                        reinterpret_cast<uint32_t *>(base + l.tail), reinterpret_cast<uint32_t *>(base) }
                              : RxWork{ reinterpret_cast<unsigned long long *>(base + l.mom),
                                        reinterpret_cast<int16_t *>(base + l.cic),
                                        reinterpret_cast<int16_t *>(base + l.comp), nullptr, nullptr, nullptr };
    }
    return rx_run(ctx, adc12, row_bytes, n_blk, spec_iq, voice_iq, part, n_parts, rw, nullptr);
}

/* Debug switch: plan 0 picks the call plan by shape (two kernels for every multiple of 256 blocks
 * in a call, six for what is left), plan 1 always runs the six kernels, plan 2 is plan 0 with the
 * vector-ALU ingest kernel in the two-kernel parts.  Same results every way. */
int ddc_debug_rx_plan(ddc_ctx *ctx, int plan)
{
    if (!ctx) return -EINVAL;
    if (plan < 0 || plan > 2) return fail(ctx, -EINVAL, "ddc_debug_rx_plan: plan is 0, 1 or 2");
    ctx->rx_plan = plan;
    return 0;
}

int ddc_rx_process(ddc_ctx *ctx, const uint8_t *adc12, size_t row_bytes, size_t samples,
                   int16_t *spec_iq, int16_t *voice_iq)
{
    if (!ctx) return -EINVAL;
    if (samples == 0) return 0;
    if (!adc12 || !spec_iq || !voice_iq) return fail(ctx, -EINVAL, "null buffer");
    if (samples % DDC_DECIMATION) return fail(ctx, -EINVAL, "samples must be a multiple of 1024");
    const size_t nch = ctx->cfg.n_channels;
    const size_t need_row = samples / 2 * 3;
    if (row_bytes && row_bytes < need_row) return fail(ctx, -EINVAL, "row_bytes smaller than one row of samples");
    const size_t drow = (need_row + 255) / 256 * 256;
    const size_t in_rows = row_bytes ? nch : 1; // row_bytes = 0: one shared ADC stream
    const size_t in_bytes = drow * in_rows;
    const size_t out_elems = nch * (samples / DDC_DECIMATION) * 2;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (in_bytes > ctx->in_cap) {
        if (ctx->d_in) (void)hipFree(ctx->d_in);
        ctx->d_in = nullptr; ctx->in_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_in, in_bytes));
        ctx->in_cap = in_bytes;
    }
    if (out_elems * 2 > ctx->out_cap) {
        if (ctx->d_out) (void)hipFree(ctx->d_out);
        ctx->d_out = nullptr; ctx->out_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_out, out_elems * 2 * sizeof(int16_t)));
        ctx->out_cap = out_elems * 2;
    }
    HIP_TRY(ctx, hipMemcpy2DAsync(ctx->d_in, drow, adc12, row_bytes ? row_bytes : need_row, need_row, in_rows,
                                  hipMemcpyHostToDevice, ctx->stream));
    int rc = ddc_rx_process_dev(ctx, ctx->d_in, row_bytes ? drow : 0, samples, ctx->d_out, ctx->d_out + out_elems);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(spec_iq, ctx->d_out, out_elems * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(voice_iq, ctx->d_out + out_elems, out_elems * sizeof(int16_t), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_rx_read_cic(ddc_ctx *ctx, int16_t *cic_iq, size_t samples)
{
    if (!ctx || !cic_iq) return -EINVAL;
    const uint32_t n_blk = (uint32_t)(samples / 512);
    if (n_blk == 0 || n_blk != ctx->last_blk) return fail(ctx, -EINVAL, "ddc_rx_read_cic: size differs from the last call");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const int16_t *src[2] = { ctx->last_cic, ctx->last_cic2 };
    size_t blk0 = 0;
    for (int k = 0; k < 2; k++) { // the parts the call ran as, each with its own work buffer
        const uint32_t pb = ctx->last_part_blk[k];
        if (!pb || !src[k]) continue;
        HIP_TRY(ctx, hipMemcpy2DAsync(cic_iq + blk0 * 2, (size_t)n_blk * 4, src[k] + (size_t)ddc::kCicHist * 2,
                                      (size_t)(ddc::kCicHist + pb) * 4, (size_t)pb * 4, ctx->cfg.n_channels,
                                      hipMemcpyDeviceToHost, ctx->stream));
        blk0 += pb;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ---- multi-GPU: channel shards and the all-gather of the decimated outputs ---- */

void ddc_shard_range(uint32_t rank, uint32_t world, uint32_t total_channels, uint32_t *lo, uint32_t *hi)
{
    uint32_t a = 0, b = 0;
    if (world && rank < world) {
        const uint32_t base = total_channels / world, extra = total_channels % world;
        a = rank * base + std::min(rank, extra);
        b = a + base + (rank < extra ? 1u : 0u);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}

extern "C++" {
namespace {
// RCCL is bound at first use, by soname: a process that already has an RCCL loaded (PyTorch brings its own)
// gets that instance, so that communicators and the HIP runtime under them are the process's one set.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

RcclApi *rccl_api(ddc_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.AllGather) return &g_rccl;
    if (!g_rccl.handle) {
        for (const char *name : { "librccl.so.1", "librccl.so" }) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) {
            const char *e = dlerror();
            fail(ctx, -ENOSYS, std::string("RCCL is not loadable: ") + (e ? e : "dlopen failed"));
            return nullptr;
        }
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
    auto ag = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.handle, "ncclAllGather"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GetErrorString || !ag) {
        fail(ctx, -ENOSYS, "the RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
        return nullptr;
    }
    g_rccl.AllGather = ag;
    return &g_rccl;
}

int gather_resources(ddc_ctx *ctx)
{
    if (ctx->gather_stream) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->gather_stream, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_computed[s], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_gather0[s]));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_gathered[s]));
    }
    return 0;
}
} // namespace
} // extern "C++"

#define RCCL_TRY(ctx, api, call)                                                                   \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(ctx, -EIO, std::string(#call) + ": " + (api)->GetErrorString(r_));         \
    } while (0)

int ddc_comm_unique_id(uint8_t id[DDC_COMM_ID_BYTES])
{
    static_assert(DDC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "communicator id size");
    if (!id) return -EINVAL;
    RcclApi *api = rccl_api(nullptr);
    if (!api) return -ENOSYS;
    ncclUniqueId u;
    RCCL_TRY(nullptr, api, api->GetUniqueId(&u));
    memcpy(id, u.internal, DDC_COMM_ID_BYTES);
    return 0;
}

int ddc_comm_init(ddc_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t id[DDC_COMM_ID_BYTES])
{
    if (!ctx || !id) return -EINVAL;
    if (world == 0 || rank >= world) return fail(ctx, -EINVAL, "ddc_comm_init: rank must be below world");
    if (ctx->comm) return fail(ctx, -EBUSY, "this ctx already has a communicator");
    RcclApi *api = rccl_api(ctx);
    if (!api) return -ENOSYS;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ncclUniqueId u;
    memcpy(u.internal, id, DDC_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    RCCL_TRY(ctx, api, api->CommInitRank(&c, (int)world, u, (int)rank));
    ctx->comm = c;
    ctx->comm_owned = true;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return gather_resources(ctx);
}

int ddc_comm_attach(ddc_ctx *ctx, void *rccl_comm, uint32_t world, uint32_t rank)
{
    if (!ctx || !rccl_comm) return -EINVAL;
    if (world == 0 || rank >= world) return fail(ctx, -EINVAL, "ddc_comm_attach: rank must be below world");
    if (ctx->comm) return fail(ctx, -EBUSY, "this ctx already has a communicator");
    if (!rccl_api(ctx)) return -ENOSYS;
    ctx->comm = static_cast<ncclComm_t>(rccl_comm);
    ctx->comm_owned = false;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return gather_resources(ctx);
}

int ddc_comm_destroy(ddc_ctx *ctx)
{
    if (!ctx) return -EINVAL;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->gather_stream) (void)hipStreamSynchronize(ctx->gather_stream);
    if (ctx->comm && ctx->comm_owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_owned = false;
    ctx->comm_world = ctx->comm_rank = 0;
    for (int s = 0; s < 2; s++) {
        for (hipEvent_t *e : { &ctx->ev_computed[s], &ctx->ev_gather0[s], &ctx->ev_gathered[s] })
            if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        ctx->gather_pending[s] = false;
    }
    if (ctx->gather_stream) { (void)hipStreamDestroy(ctx->gather_stream); ctx->gather_stream = nullptr; }
    return 0;
}

int ddc_allgather_decimated_dev(ddc_ctx *ctx, int slot, const int16_t *local, size_t n_frames, int16_t *gathered)
{
    if (!ctx) return -EINVAL;
    if (slot != 0 && slot != 1) return fail(ctx, -EINVAL, "ddc_allgather_decimated_dev: slot is 0 or 1");
    if (!local || !gathered) return fail(ctx, -EINVAL, "null buffer");
    if (!ctx->comm) return fail(ctx, -ENOTCONN, "no communicator: call ddc_comm_init or ddc_comm_attach first");
    RcclApi *api = rccl_api(ctx);
    if (!api) return -ENOSYS;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    // SPEC block then VOICE block of this rank, [2][n_channels][n_frames][2] int16, sent as bytes
    const size_t bytes = (size_t)2 * ctx->cfg.n_channels * n_frames * 2 * sizeof(int16_t);
    HIP_TRY(ctx, hipEventRecord(ctx->ev_computed[slot], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->gather_stream, ctx->ev_computed[slot], 0));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_gather0[slot], ctx->gather_stream));
    RCCL_TRY(ctx, api, api->AllGather(local, gathered, bytes, ncclUint8, ctx->comm, ctx->gather_stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_gathered[slot], ctx->gather_stream));
    ctx->gather_pending[slot] = true;
    return 0;
}

int ddc_allgather_reuse(ddc_ctx *ctx, int slot)
{
    if (!ctx) return -EINVAL;
    if (slot != 0 && slot != 1) return fail(ctx, -EINVAL, "ddc_allgather_reuse: slot is 0 or 1");
    if (!ctx->gather_pending[slot]) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[slot], 0));
    return 0;
}

int ddc_allgather_wait(ddc_ctx *ctx, int slot, float *gather_ms)
{
    if (!ctx) return -EINVAL;
    if (slot < -1 || slot > 1) return fail(ctx, -EINVAL, "ddc_allgather_wait: slot is 0, 1 or -1");
    if (gather_ms) *gather_ms = 0.0f;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    for (int s = 0; s < 2; s++) {
        if ((slot >= 0 && s != slot) || !ctx->gather_pending[s]) continue;
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_gathered[s]));
        if (gather_ms && s == (slot < 0 ? 1 : slot)) HIP_TRY(ctx, hipEventElapsedTime(gather_ms, ctx->ev_gather0[s], ctx->ev_gathered[s]));
    }
    if (gather_ms && slot < 0 && !ctx->gather_pending[1] && ctx->gather_pending[0])
        HIP_TRY(ctx, hipEventElapsedTime(gather_ms, ctx->ev_gather0[0], ctx->ev_gathered[0]));
    return 0;
}

/* ---- TX ---------------------------------------------------------------- */

static int tx_reset_range(ddc_ctx *ctx, size_t lo, size_t n);

int ddc_tx_reset(ddc_ctx *ctx, int32_t channel)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    const size_t lo = channel < 0 ? 0 : (size_t)channel;
    const size_t n = channel < 0 ? ctx->cfg.n_channels : 1;
    return tx_reset_range(ctx, lo, n);
}

static int tx_reset_range(ddc_ctx *ctx, size_t lo, size_t n)
{
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->tx.d_acc + lo, 0, n * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->tx.d_hist_x + lo * ddc::kTxXHist * 2, 0, n * ddc::kTxXHist * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->tx.d_hist_y + lo * ddc::kTxYHist * 2, 0, n * ddc::kTxYHist * 4, ctx->stream));
    return 0;
}

int ddc_tx_set_phase(ddc_ctx *ctx, int32_t channel, uint32_t phase22)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0) return fail(ctx, -EINVAL, "ddc_tx_set_phase needs one channel");
    const uint32_t v = (phase22 & 0x3FFFFF) << 10;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->tx.d_acc + channel, &v, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_tx_process_dev(ddc_ctx *ctx, const int16_t *iq48k, size_t n_iq, uint16_t *dac14)
{
    if (!ctx) return -EINVAL;
    if (n_iq == 0) return 0;
    if (!iq48k || !dac14) return fail(ctx, -EINVAL, "null buffer");
    if (((uintptr_t)dac14 % 16) || ((uintptr_t)iq48k % 4)) return fail(ctx, -EINVAL, "TX buffers must be 16-byte (DAC) / 4-byte (I/Q) aligned");
    const uint64_t nch = ctx->cfg.n_channels;
    if (n_iq > (1u << 21) || nch * (2 * n_iq + 1) >= (1ull << 31))
        return fail(ctx, -EINVAL, "too many TX samples in one call");
    const uint32_t n48 = (uint32_t)n_iq, n96 = 2 * n48;
    if ((unsigned long long)nch * n96 >= (1ull << 26)) return fail(ctx, -EINVAL, "more than 2^26 blocks of 512 DAC samples in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ddc_tx_state &tx = ctx->tx;
    if (n48 > tx.cap48) {
        if (tx.d_ybuf) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(tx.d_ybuf); }
        tx.d_ybuf = nullptr; tx.cap48 = 0;
        HIP_TRY(ctx, hipMalloc(&tx.d_ybuf, nch * (ddc::kTxYHist + (size_t)n96) * 4));
        tx.cap48 = n48;
    }
    hipStream_t st = ctx->stream;
    ddc::ddc_tx_load_history<<<(uint32_t)nch, 64, 0, st>>>(tx.d_hist_y, tx.d_ybuf, n96);
    const uint32_t nfr = (uint32_t)(nch * n48);
    ddc::ddc_tx_ciccomp<<<(nfr + 255) / 256, 256, 0, st>>>(iq48k, tx.d_hist_x, n48, (uint32_t)nch, tx.d_ybuf);
    const uint32_t total = (uint32_t)(nch * n96);
    ddc::ddc_tx_duc<<<(total + 255) / 256, 256, 0, st>>>(tx.d_ybuf, n96, total, tx.d_acc, ctx->d_phi, dac14,
                                                        ctx->d_nco_tab);
    ddc::ddc_tx_save_history<<<(uint32_t)nch, 64, 0, st>>>(tx.d_hist_x, tx.d_hist_y, iq48k, tx.d_ybuf, tx.d_acc,
                                                        ctx->d_phi, n48);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ddc_tx_process(ddc_ctx *ctx, const int16_t *iq48k, size_t n_iq, uint16_t *dac14)
{
    if (!ctx) return -EINVAL;
    if (n_iq == 0) return 0;
    if (!iq48k || !dac14) return fail(ctx, -EINVAL, "null buffer");
    const size_t nch = ctx->cfg.n_channels;
    const size_t in_elems = nch * n_iq * 2, out_elems = nch * n_iq * 1024;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ddc_tx_state &tx = ctx->tx;
    if (in_elems > tx.in_cap) {
        if (tx.d_in) (void)hipFree(tx.d_in);
        tx.d_in = nullptr; tx.in_cap = 0;
        HIP_TRY(ctx, hipMalloc(&tx.d_in, in_elems * sizeof(int16_t)));
        tx.in_cap = in_elems;
    }
    if (out_elems > tx.out_cap) {
        if (tx.d_out) (void)hipFree(tx.d_out);
        tx.d_out = nullptr; tx.out_cap = 0;
        HIP_TRY(ctx, hipMalloc(&tx.d_out, out_elems * sizeof(uint16_t)));
        tx.out_cap = out_elems;
    }
    HIP_TRY(ctx, hipMemcpyAsync(tx.d_in, iq48k, in_elems * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = ddc_tx_process_dev(ctx, tx.d_in, n_iq, tx.d_out);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dac14, tx.d_out, out_elems * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_sync(ddc_ctx *ctx)
{
    if (!ctx) return -EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

void *ddc_stream(ddc_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int ddc_set_stream(ddc_ctx *ctx, void *hip_stream)
{
    if (!ctx) return -EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

int ddc_profile_enable(ddc_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    ctx->profile = on != 0;
    ctx->prof_head = 0;
    ctx->prof_count = 0;
    return 0;
}

int ddc_profile_collect(ddc_ctx *ctx, double *front_ms, double *tail_ms, uint32_t *n_calls)
{
    if (!ctx || !front_ms || !tail_ms || !n_calls) return -EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double f = 0, t = 0;
    for (int i = 0; i < ctx->prof_count; i++) {
        float a = 0, b = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev[i][0], ctx->ev[i][1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev[i][1], ctx->ev[i][2]));
        f += a;
        t += b;
    }
    *front_ms = f;
    *tail_ms = t;
    *n_calls = (uint32_t)ctx->prof_count;
    ctx->prof_head = 0;
    ctx->prof_count = 0;
    return 0;
}

/* ---- audio path and panadapter ---------------------------------------------- */

int ddc_audio_params_for_mode(uint8_t mode, uint16_t ssb_filter, uint16_t cw_filter, uint16_t fm_filter,
                              ddc_audio_params *p)
{
    if (!p) return -EINVAL;
    if (mode > DDC_MODE_LOOPBACK) return fail(nullptr, -EINVAL, "no such mode");
    p->mode = mode;
    switch (mode) { // trx_manager.c:197-216
    case DDC_MODE_LSB: case DDC_MODE_USB: case DDC_MODE_DIGI_L: case DDC_MODE_DIGI_U: case DDC_MODE_AM:
        p->filter_width = ssb_filter;
        break;
    case DDC_MODE_CW_L: case DDC_MODE_CW_U:
        p->filter_width = cw_filter;
        break;
    case DDC_MODE_NFM:
        p->filter_width = fm_filter;
        break;
    case DDC_MODE_WFM:
        p->filter_width = 0;
        break;
    default: // IQ, NO_TX, LOOPBACK: the firmware leaves the VFO's width as it is
        break;
    }
    return 0;
}

int ddc_audio_configure(ddc_ctx *ctx, int32_t channel, const ddc_audio_params *p)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (!p) return fail(ctx, -EINVAL, "null params");
    if (p->mode > DDC_MODE_LOOPBACK) return fail(ctx, -EINVAL, "unknown mode");
    if (p->agc_speed == 0 || p->fft_averaging == 0) return fail(ctx, -EINVAL, "agc_speed and fft_averaging must be >= 1");
    ddc::AudioState st;
    memset(&st, 0, sizeof st);
    st.p = *p;
    st.lpf_n = 7;
    if (p->filter_width) {
        int li = -1;
        for (int i = 0; i < DDC_N_LPF; i++) if (ddc_lpf_hz[i] == p->filter_width) li = i;
        if (li < 0) return fail(ctx, -EINVAL, "filter_width names no LPF table of the firmware");
        st.lpf_idx = li;
        st.lpf_n = ddc_lpf_stages[li];
    }
    if (p->mode == DDC_MODE_USB || p->mode == DDC_MODE_LSB) { // audio_filters.c:306-338
        int hi = -1;
        for (int i = 1; i < DDC_N_HPF; i++) if (ddc_hpf_hz[i] == p->hpf_pass) hi = i;
        if (hi < 0) return fail(ctx, -EINVAL, "hpf_pass names no HPF table of the firmware");
        st.hpf_idx = hi;
    }
    { // calcBiquad(BIQUAD_notch, fc, 48000, 0.5, 0), audio_filters.c:375-494
        const uint32_t fs = 48000, fc = p->notch_fc;
        const float q = 0.5f;
        const float k = (float)tan(ddc::kPiF * fc / fs);
        const float norm = (float)(1.0 / (1.0 + k / q + k * k));
        const float a0 = (float)((1.0 + k * k) * norm);
        const float a1 = (float)(2.0 * (k * k - 1.0) * norm);
        const float b2 = (float)((1.0 - k / q + k * k) * norm);
        st.notch_c[0] = a0; st.notch_c[1] = a1; st.notch_c[2] = a0; st.notch_c[3] = -a1; st.notch_c[4] = -b2;
    }
    st.agc_gain = 0.0f;
    st.agc_gain_old = 1.0f;
    st.agc_up = 500.0f / (float)p->agc_speed; // agc.c:9-19
    st.agc_down = st.agc_up / 10.0f;
    { // CWDecoder_Init, cw_decoder.c:41-54: bin k = (int)(0.5 + 192 * 350 / 48000) = 1
        const float omega = (float)((2.0 * ddc::kPiF * 1) / 192);
        st.cw_coeff = (float)(2.0 * ddc::arm_cos_with(arm_sin_table(), omega));
        st.cw_limit = 50;
        st.cw_limit_low = 50;
        memset(st.cw_text, ' ', 15);
    }
    static thread_local ddc::ZoomState zs; // 4 KB: not on the stack of a small-stack caller; per thread, ctxs may be configured concurrently
    memset(&zs, 0, sizeof zs);
    zs.zoom = p->fft_zoom > 1 ? p->fft_zoom : 1;
    if (zs.zoom > 1) { // FFT_Init, fft.c:188-209
        int zi;
        switch (zs.zoom) { case 2: zi = 0; break; case 4: zi = 1; break; case 8: zi = 2; break; case 16: zi = 3; break;
                           default: return fail(ctx, -EINVAL, "fft_zoom must be 1, 2, 4, 8 or 16"); }
        for (int i = 0; i < 20; i++) zs.c[i] = (float)kZoomBiquad[zi][i];
        for (int i = 0; i < 4; i++) zs.fir[i] = (float)kZoomFir[zi][i];
    }
    static thread_local ddc::TxAudioState ts;
    memset(&ts, 0, sizeof ts);
    ts.alc_gain = 1.0f;
    ts.fm_index = 2.0f; // ModulateFM, audio_processor.c:597-606
    switch (p->filter_width) {
    case 5000: ts.fm_index = 2.0f; break; case 6000: ts.fm_index = 3.0f; break; case 7000: ts.fm_index = 4.0f; break;
    case 8000: ts.fm_index = 5.0f; break; case 9000: ts.fm_index = 6.0f; break; case 9500: ts.fm_index = 7.0f; break;
    case 10000: ts.fm_index = 8.0f; break; case 15000: ts.fm_index = 9.0f; break; case 0: ts.fm_index = 10.0f; break;
    default: break; }
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t lo = channel < 0 ? 0 : (size_t)channel;
    const size_t hi = channel < 0 ? ctx->cfg.n_channels : lo + 1;
    {
        std::vector<ddc::AudioState> all(hi - lo, st);
        HIP_TRY(ctx, hipMemcpy(ctx->d_audio + lo, all.data(), all.size() * sizeof st, hipMemcpyHostToDevice));
    }
    {
        std::vector<ddc::ZoomState> all(hi - lo, zs);
        HIP_TRY(ctx, hipMemcpy(ctx->d_zoom + lo, all.data(), all.size() * sizeof zs, hipMemcpyHostToDevice));
    }
    {
        std::vector<ddc::TxAudioState> all(hi - lo, ts);
        HIP_TRY(ctx, hipMemcpy(ctx->d_audio_tx + lo, all.data(), all.size() * sizeof ts, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMemset(ctx->d_wtf + lo * 50 * 256, 0, (hi - lo) * 50 * 256 * sizeof(uint16_t)));
    for (size_t c = lo; c < hi; c++) { ctx->audio_ready[c] = 1; ctx->zoom_of[c] = (uint8_t)zs.zoom; ctx->audio_cfg[c] = *p; }
    ctx->fft_zoom_all = ctx->zoom_of[0];
    for (uint8_t zv : ctx->zoom_of) if (zv != ctx->zoom_of[0]) ctx->fft_zoom_all = 0;
    return 0;
}

int ddc_audio_update(ddc_ctx *ctx, int32_t channel, const ddc_audio_params *p)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (!p) return fail(ctx, -EINVAL, "null params");
    if (p->agc_speed == 0 || p->fft_averaging == 0) return fail(ctx, -EINVAL, "agc_speed and fft_averaging must be >= 1");
    const uint32_t lo = channel < 0 ? 0 : (uint32_t)channel;
    const uint32_t hi = channel < 0 ? ctx->cfg.n_channels : lo + 1;
    for (uint32_t c = lo; c < hi; c++) {
        const ddc_audio_params &q = ctx->audio_cfg[c];
        if (!ctx->audio_ready[c]) return fail(ctx, -EINVAL, "ddc_audio_configure has not been called for this channel");
        if (q.mode != p->mode || q.filter_width != p->filter_width || q.hpf_pass != p->hpf_pass ||
            q.notch_fc != p->notch_fc || (q.fft_zoom > 1 ? q.fft_zoom : 1) != (p->fft_zoom > 1 ? p->fft_zoom : 1))
            return fail(ctx, -EINVAL, "ddc_audio_update cannot change mode, filter_width, hpf_pass, notch_fc or fft_zoom");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(ddc::ddc_audio_update_k, dim3((hi - lo + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_audio, lo, hi, *p);
    HIP_TRY(ctx, hipGetLastError());
    for (uint32_t c = lo; c < hi; c++) ctx->audio_cfg[c] = *p;
    return 0;
}

static int audio_all_ready(ddc_ctx *ctx)
{
    for (uint8_t r : ctx->audio_ready)
        if (!r) return fail(ctx, -EINVAL, "ddc_audio_configure has not been called for every channel");
    return 0;
}

int ddc_audio_rx_process_dev(ddc_ctx *ctx, const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames,
                             int32_t *audio_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!spec_iq || !voice_iq || !audio_lr) return fail(ctx, -EINVAL, "null buffer");
    if (n_frames % ddc::kABlock) return fail(ctx, -EINVAL, "n_frames must be a multiple of 192");
    if (n_frames > 0xFFFFFF00u) return fail(ctx, -EINVAL, "n_frames too large");
    if (((uintptr_t)spec_iq | (uintptr_t)voice_iq | (uintptr_t)audio_lr) % 16)
        return fail(ctx, -EINVAL, "audio buffers must be 16-byte aligned");
    int rc = audio_all_ready(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const uint32_t nch = ctx->cfg.n_channels;
    if (n_frames > ctx->audio_filt_cap) {
        if (ctx->d_audio_filt) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_audio_filt); }
        ctx->d_audio_filt = nullptr; ctx->audio_filt_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_audio_filt, (size_t)nch * n_frames * sizeof(float2)));
        ctx->audio_filt_cap = n_frames;
    }
    // filters as a systolic pipeline (20 lanes per channel), then demodulator and block-wise stages (lane per channel)
    hipLaunchKernelGGL(ddc::ddc_audio_rx_filters, dim3((nch + ddc::kFiltPerWave - 1) / ddc::kFiltPerWave), dim3(64), 0,
                       ctx->stream, ctx->d_audio, nch, spec_iq, voice_iq, (uint32_t)n_frames, ctx->d_audio_filt);
    hipLaunchKernelGGL(ddc::ddc_audio_rx_demod, dim3(nch), dim3(64), 0, ctx->stream, ctx->d_audio, nch,
                       ctx->d_audio_filt, (uint32_t)n_frames, audio_lr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ddc_audio_usb_frames_dev(ddc_ctx *ctx, const int32_t *audio_lr, size_t n_frames, int16_t *usb_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!audio_lr || !usb_lr) return fail(ctx, -EINVAL, "null buffer");
    if (n_frames % 2) return fail(ctx, -EINVAL, "n_frames must be even");
    if (n_frames > 0xFFFFFF00u) return fail(ctx, -EINVAL, "n_frames too large");
    if (((uintptr_t)audio_lr % 16) || ((uintptr_t)usb_lr % 8))
        return fail(ctx, -EINVAL, "audio_lr must be 16-byte aligned, usb_lr 8-byte aligned");
    int rc = audio_all_ready(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const uint64_t items = (uint64_t)ctx->cfg.n_channels * (n_frames / 2);
    if (items >= (1ull << 39)) return fail(ctx, -EINVAL, "too many frames in one call");
    hipLaunchKernelGGL(ddc::ddc_audio_usb_frames, dim3((uint32_t)((items + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_audio, ctx->cfg.n_channels, audio_lr, (uint32_t)n_frames, usb_lr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

static int ensure_stage(ddc_ctx *ctx, int slot, size_t bytes)
{
    if (bytes <= ctx->stage_cap[slot]) return 0;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage[slot]) (void)hipFree(ctx->d_stage[slot]);
    ctx->d_stage[slot] = nullptr;
    ctx->stage_cap[slot] = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_stage[slot], bytes));
    ctx->stage_cap[slot] = bytes;
    return 0;
}

int ddc_audio_rx_process(ddc_ctx *ctx, const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames,
                         int32_t *audio_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!spec_iq || !voice_iq || !audio_lr) return fail(ctx, -EINVAL, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t in_b = (size_t)ctx->cfg.n_channels * n_frames * 4, out_b = in_b * 2;
    int rc;
    if ((rc = ensure_stage(ctx, 0, in_b)) || (rc = ensure_stage(ctx, 1, in_b)) || (rc = ensure_stage(ctx, 2, out_b))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], spec_iq, in_b, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[1], voice_iq, in_b, hipMemcpyHostToDevice, ctx->stream));
    rc = ddc_audio_rx_process_dev(ctx, (const int16_t *)ctx->d_stage[0], (const int16_t *)ctx->d_stage[1], n_frames,
                                  (int32_t *)ctx->d_stage[2]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(audio_lr, ctx->d_stage[2], out_b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_audio_usb_frames(ddc_ctx *ctx, const int32_t *audio_lr, size_t n_frames, int16_t *usb_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!audio_lr || !usb_lr) return fail(ctx, -EINVAL, "null buffer");
    const size_t n = (size_t)ctx->cfg.n_channels * n_frames * 2;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int rc;
    if ((rc = ensure_stage(ctx, 0, n * sizeof(int32_t))) || (rc = ensure_stage(ctx, 1, n * sizeof(int16_t)))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], audio_lr, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = ddc_audio_usb_frames_dev(ctx, (const int32_t *)ctx->d_stage[0], n_frames, (int16_t *)ctx->d_stage[1]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(usb_lr, ctx->d_stage[1], n * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_fft_process_dev(ddc_ctx *ctx, const int16_t *spec_iq, size_t n_frames, float *bins, uint16_t *rows)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!spec_iq) return fail(ctx, -EINVAL, "null buffer");
    if (n_frames % 512) return fail(ctx, -EINVAL, "n_frames must be a multiple of 512");
    if ((uintptr_t)spec_iq % 16) return fail(ctx, -EINVAL, "spec_iq must be 16-byte aligned");
    const size_t fpc = n_frames / 512;
    const size_t nch = ctx->cfg.n_channels;
    if (nch * fpc > 0x7FFFFFFFu) return fail(ctx, -EINVAL, "too many FFT frames in one call");
    int rc = audio_all_ready(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    // one launch handles one zoom factor: the frame step in the work buffer depends on it
    if (ctx->fft_zoom_all == 0) return fail(ctx, -EINVAL, "ddc_fft_process needs the same fft_zoom on every channel");
    const uint32_t M = ctx->fft_zoom_all, zw = 512 / M;
    const size_t pitch = (512 - zw) + fpc * zw; // floats per channel and branch
    if (fpc > ctx->fft_cap || pitch > ctx->fft_work_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_fft_work) (void)hipFree(ctx->d_fft_work);
        if (ctx->d_fft_mag) (void)hipFree(ctx->d_fft_mag);
        if (ctx->d_fft_rows) (void)hipFree(ctx->d_fft_rows);
        ctx->d_fft_work = nullptr; ctx->d_fft_mag = nullptr; ctx->d_fft_rows = nullptr; ctx->fft_cap = 0; ctx->fft_work_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_fft_work, nch * pitch * sizeof(float2)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_fft_mag, nch * fpc * 256 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_fft_rows, nch * fpc * 256 * sizeof(uint16_t)));
        ctx->fft_cap = fpc;
        ctx->fft_work_cap = pitch;
    }
    {
        const dim3 grid((unsigned)((2 * nch + 63) / 64)), block(64);
        float *work = reinterpret_cast<float *>(ctx->d_fft_work);
#define DDC_PREFILTER(Z) hipLaunchKernelGGL(ddc::ddc_fft_prefilter<Z>, grid, block, 0, ctx->stream, ctx->d_audio, \
                                            ctx->d_zoom, (uint32_t)nch, spec_iq, (uint32_t)n_frames, work, (uint32_t)pitch)
        switch (M) {
        case 1: // no zoom: the tiled form (recursion lanes fed through LDS by loader and writer waves)
            hipLaunchKernelGGL(ddc::ddc_fft_prefilter_tiled, dim3((unsigned)((nch + ddc::kPfCh - 1) / ddc::kPfCh)), dim3(ddc::kPfThreads), 0,
                               ctx->stream, ctx->d_audio, (uint32_t)nch, spec_iq, (uint32_t)n_frames, work, (uint32_t)pitch);
            break;
        case 2: DDC_PREFILTER(2); break;
        case 4: DDC_PREFILTER(4); break;
        case 8: DDC_PREFILTER(8); break;
        case 16: DDC_PREFILTER(16); break;
        default: return fail(ctx, -EINVAL, "fft_zoom must be 1, 2, 4, 8 or 16");
        }
#undef DDC_PREFILTER
    }
    hipLaunchKernelGGL(ddc::ddc_fft_frames, dim3((unsigned)(nch * fpc)), dim3(64), 0, ctx->stream,
                       reinterpret_cast<const float *>(ctx->d_fft_work),
                       (uint32_t)fpc, (uint32_t)pitch, zw, ctx->d_fft_mag);
    hipLaunchKernelGGL(ddc::ddc_fft_level, dim3((unsigned)nch), dim3(64), 0, ctx->stream, ctx->d_audio, ctx->d_fft_mag,
                       (uint32_t)fpc, bins, rows ? rows : ctx->d_fft_rows, ctx->d_wtf);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ddc_fft_process(ddc_ctx *ctx, const int16_t *spec_iq, size_t n_frames, float *bins, uint16_t *rows)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!spec_iq) return fail(ctx, -EINVAL, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t nch = ctx->cfg.n_channels;
    const size_t in_b = nch * n_frames * 4, nb = nch * (n_frames / 512) * 256;
    int rc;
    if ((rc = ensure_stage(ctx, 0, in_b)) || (rc = ensure_stage(ctx, 1, nb * 4)) || (rc = ensure_stage(ctx, 2, nb * 2))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], spec_iq, in_b, hipMemcpyHostToDevice, ctx->stream));
    rc = ddc_fft_process_dev(ctx, (const int16_t *)ctx->d_stage[0], n_frames, (float *)ctx->d_stage[1],
                             (uint16_t *)ctx->d_stage[2]);
    if (rc) return rc;
    if (bins) HIP_TRY(ctx, hipMemcpyAsync(bins, ctx->d_stage[1], nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (rows) HIP_TRY(ctx, hipMemcpyAsync(rows, ctx->d_stage[2], nb * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_audio_tx_process_dev(ddc_ctx *ctx, const int16_t *mic_lr, size_t n_frames, int16_t *iq48k, int32_t *loop_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!mic_lr || !iq48k) return fail(ctx, -EINVAL, "null buffer");
    if (n_frames % ddc::kABlock) return fail(ctx, -EINVAL, "n_frames must be a multiple of 192");
    if (n_frames > 0xFFFFFF00u) return fail(ctx, -EINVAL, "n_frames too large");
    if ((uintptr_t)mic_lr % 16) return fail(ctx, -EINVAL, "mic_lr must be 16-byte aligned");
    int rc = audio_all_ready(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const uint32_t nch = ctx->cfg.n_channels;
    if (n_frames > ctx->tx_xs_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_tx_xs) (void)hipFree(ctx->d_tx_xs);
        ctx->d_tx_xs = nullptr; ctx->tx_xs_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_tx_xs, (size_t)nch * (200 + n_frames) * sizeof(float)));
        ctx->tx_xs_cap = n_frames;
    }
    hipLaunchKernelGGL(ddc::ddc_audio_tx, dim3((nch + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_audio, ctx->d_audio_tx,
                       nch, mic_lr, (uint32_t)n_frames, iq48k, loop_lr, ctx->d_tx_xs);
    const uint32_t ftiles = (uint32_t)((n_frames + 255) / 256);
    hipLaunchKernelGGL(ddc::ddc_audio_tx_fir, dim3(nch * ftiles), dim3(256), 0, ctx->stream, ctx->d_audio, ftiles,
                       ctx->d_tx_xs, (uint32_t)n_frames, iq48k, loop_lr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ddc_audio_tx_process(ddc_ctx *ctx, const int16_t *mic_lr, size_t n_frames, int16_t *iq48k, int32_t *loop_lr)
{
    if (!ctx) return -EINVAL;
    if (n_frames == 0) return 0;
    if (!mic_lr || !iq48k) return fail(ctx, -EINVAL, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t b4 = (size_t)ctx->cfg.n_channels * n_frames * 4;
    int rc;
    if ((rc = ensure_stage(ctx, 0, b4)) || (rc = ensure_stage(ctx, 1, b4)) || (rc = ensure_stage(ctx, 2, b4 * 2))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], mic_lr, b4, hipMemcpyHostToDevice, ctx->stream));
    rc = ddc_audio_tx_process_dev(ctx, (const int16_t *)ctx->d_stage[0], n_frames, (int16_t *)ctx->d_stage[1],
                                  (int32_t *)ctx->d_stage[2]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(iq48k, ctx->d_stage[1], b4, hipMemcpyDeviceToHost, ctx->stream));
    if (loop_lr) HIP_TRY(ctx, hipMemcpyAsync(loop_lr, ctx->d_stage[2], b4 * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_fft_get_waterfall(ddc_ctx *ctx, int32_t channel, uint16_t *rows)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !rows) return fail(ctx, -EINVAL, "ddc_fft_get_waterfall needs one channel and an output");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(rows, ctx->d_wtf + (size_t)channel * 50 * 256, 50 * 256 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return 0;
}

int ddc_fft_move_waterfall(ddc_ctx *ctx, int32_t channel, int32_t freq_diff_hz)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0) return fail(ctx, -EINVAL, "ddc_fft_move_waterfall needs one channel");
    if (!ctx->audio_ready[channel]) return fail(ctx, -EINVAL, "ddc_audio_configure has not been called for this channel");
    if (freq_diff_hz < -32768 || freq_diff_hz > 32767) return fail(ctx, -EINVAL, "freq_diff_hz is an int16 in the firmware");
    // (freq_diff / FFT_HZ_IN_PIXEL) * TRX.FFT_Zoom, fft.c:457, in the firmware's integer arithmetic
    const int shift = (int)(int16_t)((freq_diff_hz / (48000 / 256)) * (int)ctx->zoom_of[channel]);
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(ddc::ddc_fft_move_waterfall_k, dim3(1), dim3(64), 0, ctx->stream, ctx->d_audio, ctx->d_wtf,
                       (uint32_t)channel, shift);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// log10f_fast of the firmware (functions.c:247-260): a cubic in the mantissa
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

int ddc_audio_smeter_dbm(ddc_ctx *ctx, int32_t channel, float *dbm)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !dbm) return fail(ctx, -EINVAL, "ddc_audio_smeter_dbm needs one channel and an output");
    if (!ctx->audio_ready[channel]) return fail(ctx, -EINVAL, "ddc_audio_configure has not been called for this channel");
    ddc_audio_status st;
    rc = ddc_audio_get_status(ctx, channel, &st);
    if (rc) return rc;
    const float rf_gain = ctx->audio_cfg[channel].rf_gain;
    // stm32f4xx_it.c:401-409: 16-bit bus samples -> 12-bit ADC scale -> volts (ADC_VREF 1.0) -> RMS ->
    // antenna side of the 1:4 transformer with the S9 calibration factor 0.2 -> dBm into 50 ohm
    float vpp = (st.smeter_max / rf_gain) - (st.smeter_min / rf_gain);
    for (int i = 0; i < (16 - 12); i++) vpp = vpp / 2;
    const float adc_vpp = vpp * 1.0f / ((float)pow(2.0, 12) - 1);
    const float vrms = adc_vpp * 0.3535f;
    float rf_in = (vrms / 4) * 0.2f;
    if (rf_in < 0.0000001f) rf_in = 0.0000001f;
    *dbm = 10 * log10f_fast((rf_in * rf_in) / (50.0f * 0.001f));
    hipLaunchKernelGGL(ddc::ddc_audio_smeter_clear_k, dim3(1), dim3(64), 0, ctx->stream, ctx->d_audio, (uint32_t)channel);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ddc_autogain_step(ddc_autogain *st, int16_t adc_max_amplitude)
{
    if (!st) return -EINVAL;
    // trx_manager.c:268-356, the TRX.AutoGain branch; the thresholds are evaluated in double as there
    const double att = pow(10.0, 12 / 20.0), pre = pow(10.0, 20 / 20.0); // db2rateV, functions.c:262-265
    const double amp = (double)adc_max_amplitude, limit = 1100.0;
    const uint8_t before_p = st->preamp, before_a = st->att;
    auto quiet = [&](bool ok) {
        if (ok) st->wait_reaction++;
        else st->wait_reaction = 0;
        if (st->wait_reaction >= 7) { st->stage++; st->wait_reaction = 0; }
    };
    switch (st->stage) {
    case 0: st->preamp = 0; st->att = 1; st->stage++; st->wait_reaction = 0; break;
    case 1: quiet(amp * att <= limit); break;
    case 2: st->preamp = 0; st->att = 0; st->stage++; st->wait_reaction = 0; break;
    case 3:
        if (adc_max_amplitude > 1100) st->stage -= 3; // too much gain: one setting back (and still counts below)
        quiet(amp * pre / att <= limit);
        break;
    case 4: st->preamp = 1; st->att = 1; st->stage++; st->wait_reaction = 0; break;
    case 5:
        if (adc_max_amplitude > 1100) st->stage -= 3;
        quiet(amp * att <= limit);
        break;
    case 6: st->preamp = 1; st->att = 0; st->stage++; st->wait_reaction = 0; break;
    case 7:
        if (adc_max_amplitude > 1100) st->stage -= 3;
        break;
    default: st->stage = 0; break;
    }
    return st->preamp != before_p || st->att != before_a;
}

int ddc_audio_stage(ddc_ctx *ctx, int32_t channel, int stage, int arg, float *buf, int n)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !buf) return fail(ctx, -EINVAL, "ddc_audio_stage needs one channel and a buffer");
    if (!ctx->audio_ready[channel]) return fail(ctx, -EINVAL, "ddc_audio_configure has not been called for this channel");
    switch (stage) {
    case DDC_STAGE_DC: if (n < 0 || arg < 0 || arg > 5) return fail(ctx, -EINVAL, "dc stage: slot 0..5"); break;
    case DDC_STAGE_AGC: case DDC_STAGE_CW: if (n != 192) return fail(ctx, -EINVAL, "stage works on 192 samples"); break;
    case DDC_STAGE_DNR: if (n != 64) return fail(ctx, -EINVAL, "noise reduction works on 64 samples"); break;
    default: return fail(ctx, -EINVAL, "unknown stage");
    }
    if (n == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    rc = ensure_stage(ctx, 0, (size_t)n * sizeof(float));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], buf, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ddc::ddc_audio_stage_k, dim3(1), dim3(64), 0, ctx->stream, ctx->d_audio, (uint32_t)channel, stage,
                       arg, (float *)ctx->d_stage[0], n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(buf, ctx->d_stage[0], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ddc_audio_get_status(ddc_ctx *ctx, int32_t channel, ddc_audio_status *out)
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !out) return fail(ctx, -EINVAL, "ddc_audio_get_status needs one channel and an output");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    static thread_local ddc::AudioState st;
    HIP_TRY(ctx, hipMemcpy(&st, ctx->d_audio + channel, sizeof st, hipMemcpyDeviceToHost));
    out->smeter_max = st.smeter_max;
    out->smeter_min = st.smeter_min;
    out->agc_gain = st.agc_gain;
    out->fft_max_value = st.fft_max_value;
    out->cw_wpm = st.cw_wpm;
    out->squelched = st.squelched;
    out->blocks = st.blocks;
    ddc::TxAudioState *ts = nullptr;
    {
        static thread_local ddc::TxAudioState tsv;
        HIP_TRY(ctx, hipMemcpy(&tsv, ctx->d_audio_tx + channel, sizeof tsv, hipMemcpyDeviceToHost));
        ts = &tsv;
    }
    out->alc_gain = ts->alc_gain;
    out->tx_max_amplitude = ts->max_amplitude;
    memcpy(out->cw_text, st.cw_text, 16);
    return 0;
}

/* ---- status frame and state blob ----------------------------------------------- */

int ddc_adc_stats_enable(ddc_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    ctx->adc_stats = on != 0;
    return 0;
}

int ddc_get_status(ddc_ctx *ctx, int32_t channel, uint8_t out[5])
{
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    if (channel < 0 || !out) return fail(ctx, -EINVAL, "ddc_get_status needs one channel and an output");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int lo = 0, hi = 0;
    HIP_TRY(ctx, hipMemcpy(&lo, ctx->d_adc_min + channel, 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&hi, ctx->d_adc_max + channel, 4, hipMemcpyDeviceToHost));
    // The ADC's over-range pin is not part of the sample stream; full-scale codes stand in for it.
    const int otr = ctx->adc_stats && (hi >= 2047 || lo <= -2048);
    out[0] = (uint8_t)otr;                                   // bit1 DAC_OTR can never rise, bits 2-5: front-panel keys
    out[1] = (uint8_t)((((unsigned)lo >> 8) & 0xF) << 4 | (((unsigned)hi >> 8) & 0xF));
    out[2] = (uint8_t)(lo & 0xFF);
    out[3] = (uint8_t)(hi & 0xFF);
    out[4] = 0;                                              // encoder
    const int rlo = 2000, rhi = -2000;                       // ADC_MINMAX_RESET
    HIP_TRY(ctx, hipMemcpy(ctx->d_adc_min + channel, &rlo, 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_adc_max + channel, &rhi, 4, hipMemcpyHostToDevice));
    return 0;
}

void ddc_parse_status(const uint8_t in[5], int *adc_otr, int *dac_otr, int *adc_min, int *adc_max)
{
    if (adc_otr) *adc_otr = in[0] & 1;
    if (dac_otr) *dac_otr = (in[0] >> 1) & 1;
    // FPGA_fpgadata_getparam (fpga.c:222-284) sign-extends the minimum
    // (<< 4 then / 16 on an int16) and takes the maximum as an unsigned 12-bit value
    int lo = ((in[1] >> 4) << 8) | in[2];
    const int hi = ((in[1] & 0xF) << 8) | in[3];
    if (lo & 0x800) lo -= 0x1000;
    if (adc_min) *adc_min = lo;
    if (adc_max) *adc_max = hi;
}

extern "C++" {
namespace {
struct StateSeg { void *dev; size_t bytes; };
struct StateHeader { uint32_t magic, version, n_channels, n_seg; uint64_t total; };
constexpr uint32_t kStateMagic = 0x43444455u; // "UDDC"
constexpr uint32_t kStateVersion = 3;          // 2: window moments and mixer carries hold twice the CIC input; 3: + RX/TX direction

std::vector<StateSeg> state_segments(ddc_ctx *ctx)
{
    const size_t n = ctx->cfg.n_channels;
    return {
        { ctx->d_acc, n * 4 }, { ctx->d_phi, n * 4 }, { ctx->d_carry[0], n * 24 },
        { ctx->d_hist_mom, n * ddc::kMomHist * 10 * sizeof(unsigned long long) },
        { ctx->d_hist_cic, n * ddc::kCicHist * 4 }, { ctx->d_hist_comp, n * ddc::kCompHist * 4 },
        { ctx->tx.d_acc, n * 4 }, { ctx->tx.d_hist_x, n * ddc::kTxXHist * 4 }, { ctx->tx.d_hist_y, n * ddc::kTxYHist * 4 },
        { ctx->d_adc_min, n * 4 }, { ctx->d_adc_max, n * 4 },
        { ctx->d_audio, n * sizeof(ddc::AudioState) }, { ctx->d_zoom, n * sizeof(ddc::ZoomState) },
        { ctx->d_audio_tx, n * sizeof(ddc::TxAudioState) }, { ctx->d_wtf, n * 50 * 256 * sizeof(uint16_t) },
    };
}
} // namespace
} // extern "C++"

size_t ddc_state_size(ddc_ctx *ctx)
{
    if (!ctx) return 0;
    size_t t = sizeof(StateHeader) + ctx->cfg.n_channels * (3 + sizeof(ddc_audio_params)); // + audio_ready, zoom_of, tx_dir, audio_cfg
    for (const StateSeg &g : state_segments(ctx)) t += g.bytes;
    return t;
}

int ddc_state_save(ddc_ctx *ctx, void *buf, size_t size)
{
    if (!ctx || !buf) return -EINVAL;
    const size_t need = ddc_state_size(ctx);
    if (size < need) return fail(ctx, -ENOSPC, "state buffer too small");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint8_t *p = static_cast<uint8_t *>(buf);
    const auto segs = state_segments(ctx);
    StateHeader h = { kStateMagic, kStateVersion, ctx->cfg.n_channels, (uint32_t)segs.size(), need };
    memcpy(p, &h, sizeof h);
    p += sizeof h;
    for (const StateSeg &g : segs) {
        HIP_TRY(ctx, hipMemcpy(p, g.dev, g.bytes, hipMemcpyDeviceToHost));
        p += g.bytes;
    }
    memcpy(p, ctx->audio_ready.data(), ctx->cfg.n_channels);
    memcpy(p + ctx->cfg.n_channels, ctx->zoom_of.data(), ctx->cfg.n_channels);
    memcpy(p + 2 * (size_t)ctx->cfg.n_channels, ctx->tx_dir.data(), ctx->cfg.n_channels);
    memcpy(p + 3 * (size_t)ctx->cfg.n_channels, ctx->audio_cfg.data(), ctx->cfg.n_channels * sizeof(ddc_audio_params));
    return 0;
}

int ddc_state_restore(ddc_ctx *ctx, const void *buf, size_t size)
{
    if (!ctx || !buf) return -EINVAL;
    const size_t need = ddc_state_size(ctx);
    StateHeader h;
    if (size < sizeof h) return fail(ctx, -EINVAL, "state blob truncated");
    memcpy(&h, buf, sizeof h);
    const auto segs = state_segments(ctx);
    if (h.magic != kStateMagic || h.version != kStateVersion || h.n_channels != ctx->cfg.n_channels || h.n_seg != segs.size() ||
        h.total != need || size < need)
        return fail(ctx, -EINVAL, "state blob does not match this ctx (magic / version / channel count / size)");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint8_t *p = static_cast<const uint8_t *>(buf) + sizeof h;
    for (const StateSeg &g : segs) {
        HIP_TRY(ctx, hipMemcpy(g.dev, p, g.bytes, hipMemcpyHostToDevice));
        p += g.bytes;
    }
    memcpy(ctx->audio_ready.data(), p, ctx->cfg.n_channels);
    memcpy(ctx->zoom_of.data(), p + ctx->cfg.n_channels, ctx->cfg.n_channels);
    memcpy(ctx->tx_dir.data(), p + 2 * (size_t)ctx->cfg.n_channels, ctx->cfg.n_channels);
    memcpy(ctx->audio_cfg.data(), p + 3 * (size_t)ctx->cfg.n_channels, ctx->cfg.n_channels * sizeof(ddc_audio_params));
    ctx->fft_zoom_all = ctx->zoom_of[0];
    for (uint8_t zv : ctx->zoom_of) if (zv != ctx->zoom_of[0]) ctx->fft_zoom_all = 0;
    // the host mirror of the tuning words
    std::vector<uint32_t> phi(ctx->cfg.n_channels);
    HIP_TRY(ctx, hipMemcpy(phi.data(), ctx->d_phi, phi.size() * 4, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < phi.size(); c++) ctx->h_phi[c] = phi[c]; // device format: tuning word << 10
    return 0;
}

/* ---- bus frames -------------------------------------------------------- */

void ddc_pack_rx_frames(const int16_t *spec_iq, const int16_t *voice_iq, size_t n_frames, uint8_t *out)
{
    for (size_t n = 0; n < n_frames; n++) {
        const uint16_t si = (uint16_t)spec_iq[2 * n], sq = (uint16_t)spec_iq[2 * n + 1];
        const uint16_t vi = (uint16_t)voice_iq[2 * n], vq = (uint16_t)voice_iq[2 * n + 1];
        uint8_t *o = out + 8 * n;
        o[0] = (uint8_t)(sq >> 8); o[1] = (uint8_t)sq; o[2] = (uint8_t)(si >> 8); o[3] = (uint8_t)si;
        o[4] = (uint8_t)(vq >> 8); o[5] = (uint8_t)vq; o[6] = (uint8_t)(vi >> 8); o[7] = (uint8_t)vi;
    }
}

void ddc_unpack_tx_frames(const uint8_t *in, size_t n_frames, int16_t *tx_iq)
{
    for (size_t n = 0; n < n_frames; n++) {
        const uint8_t *p = in + 4 * n;
        tx_iq[2 * n + 1] = (int16_t)(((uint16_t)p[0] << 8) | p[1]); /* Q first on the bus */
        tx_iq[2 * n] = (int16_t)(((uint16_t)p[2] << 8) | p[3]);
    }
}

void ddc_parse_params(const uint8_t in[4], int *tx, int *preamp, uint32_t *phi_inc22)
{
    if (preamp) *preamp = (in[0] >> 2) & 1;
    if (tx) *tx = (in[0] >> 3) & 1;
    if (phi_inc22) *phi_inc22 = ((uint32_t)(in[1] & 0x3F) << 16) | ((uint32_t)in[2] << 8) | in[3];
}

static int reset_range(ddc_ctx *ctx, bool tx, uint32_t lo, uint32_t n)
{
    return tx ? tx_reset_range(ctx, lo, n) : rx_reset_range(ctx, lo, n, n == ctx->cfg.n_channels);
}

int ddc_set_params(ddc_ctx *ctx, int32_t channel, const uint8_t in[4])
{
    int tx = 0, pre = 0;
    uint32_t phi = 0;
    ddc_parse_params(in, &tx, &pre, &phi);
    (void)pre; // the preamp relay is the RF unit's business
    int rc = check_channel(ctx, channel);
    if (rc) return rc;
    // rx/tx flag (FPGA/stm32_interface.v:142-171): clk_enable of one direction, reset of
    // the other, so the direction that is switched on starts from zero state
    const uint32_t lo = channel < 0 ? 0 : (uint32_t)channel;
    const uint32_t hi = channel < 0 ? ctx->cfg.n_channels : (uint32_t)channel + 1;
    // channels whose flag flips, reset run by run (one set of memsets per run of neighbours)
    uint32_t c = lo;
    while (c < hi) {
        if ((ctx->tx_dir[c] != 0) == (tx != 0)) { c++; continue; }
        uint32_t e = c;
        while (e < hi && (ctx->tx_dir[e] != 0) != (tx != 0)) e++;
        rc = reset_range(ctx, tx != 0, c, e - c);
        if (rc) return rc;
        for (uint32_t k = c; k < e; k++) ctx->tx_dir[k] = tx ? 1 : 0;
        c = e;
    }
    return ddc_tune(ctx, channel, phi);
}

/* ---- the NCO as the kernels evaluate it ----------------------------------- */

int ddc_debug_nco(ddc_ctx *ctx, uint32_t first_phase, uint32_t count, int16_t *sin14, int16_t *cos14,
                  int16_t *sin12, int16_t *cos12)
{
    if (!ctx) return -EINVAL;
    if (count == 0) return 0;
    if (!sin14 || !cos14) return fail(ctx, -EINVAL, "null buffer");
    if (count > (1u << 22)) return fail(ctx, -EINVAL, "ddc_debug_nco: at most 2^22 phases (one turn of the accumulator) per call");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int16_t *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, (size_t)count * 4 * sizeof(int16_t)));
    const uint32_t grid = std::min<uint32_t>((count + 255) / 256, 4096);
    ddc::ddc_nco_sweep<<<grid, 256, 0, ctx->stream>>>(ctx->d_nco_tab, first_phase, count, d);
    hipError_t e = hipGetLastError();
    int16_t *dst[4] = { sin14, cos14, sin12, cos12 };
    for (int k = 0; k < 4 && e == hipSuccess; k++)
        if (dst[k])
            e = hipMemcpyAsync(dst[k], d + (size_t)k * count, (size_t)count * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, -EIO, std::string("ddc_debug_nco: ") + hipGetErrorString(e));
    return 0;
}

/* ---- the constant tables compiled into this library ---------------------- */


int ddc_debug_table(const char *name, void *out, size_t cap, size_t *bytes)
{
    if (!name || !bytes) return -EINVAL;
    std::vector<uint2> nco;
    std::vector<int16_t> fine;
    std::vector<uint8_t> mx;
    const void *src = nullptr;
    size_t n = 0;
    const std::string k(name);
#define TAB(key, obj) if (k == key) { src = obj; n = sizeof obj; }
    TAB("rx_ciccomp_h", rx_ciccomp_h)
    TAB("rx_hilb_c", rx_hilb_c)
    TAB("tx_ciccomp_p1", tx_ciccomp_p1)
    TAB("tx_ciccomp_p2", tx_ciccomp_p2)
    TAB("tx_hilbert_i", tx_hilbert_i)
    TAB("tx_hilbert_q", tx_hilbert_q)
    TAB("hpf_hz", ddc_hpf_hz)
    TAB("lpf_hz", ddc_lpf_hz)
    TAB("lpf_stages", ddc_lpf_stages)
    TAB("hpf_k", ddc_hpf_k)
    TAB("hpf_v", ddc_hpf_v)
    TAB("lpf_k", ddc_lpf_k)
    TAB("lpf_v", ddc_lpf_v)
    TAB("sql_hpf_k", ddc_sql_hpf_k)
    TAB("sql_hpf_v", ddc_sql_hpf_v)
    TAB("zoom_biquad", kZoomBiquad)
    TAB("zoom_fir", kZoomFir)
#undef TAB
    if (k == "arm_sin") { src = arm_sin_table().data(); n = 513 * sizeof(float); }
    if (k == "nco_coarse") {          // as uploaded for the kernels: {2 cc | 2 sc << 16, -2 sc | 2 cc << 16}
        if (build_nco_table(nco) != 0) return fail(nullptr, -EFAULT, "fine-sine shortcut does not reproduce the NCO fine ROM");
        src = nco.data();
        n = nco.size() * sizeof(uint2);
    } else if (k == "nco_fine_sine") { // the kernels' arithmetic fine sine for every 11-bit fine phase
        fine.resize(2048);
        for (uint32_t f = 0; f < 2048; f++) fine[f] = (int16_t)(nco_fine_pair_host(f) & 0xFFFFu);
        src = fine.data();
        n = fine.size() * sizeof(int16_t);
    }
    if (k == "rx_moment_weights") {    // byte planes of the window-moment weights as ddc_rx_front_mx multiplies them
        mx.resize(ddc::kMxTabBytes);
        if (!ddc::build_mx_table(mx.data())) return fail(nullptr, -EFAULT, "a window-moment weight does not fit its byte planes");
        src = mx.data();
        n = mx.size();
    }
    if (!src) return fail(nullptr, -ENOENT, "no such table: " + k);
    *bytes = n;
    if (!out) return 0;
    if (cap < n) return fail(nullptr, -ENOSPC, "buffer too small for table " + k);
    memcpy(out, src, n);
    return 0;
}

/* ---- packed ADC words and the synthetic bench signal ------------------- */

void ddc_pack_adc12(const int16_t *samples, size_t n, uint8_t *packed)
{
    for (size_t i = 0; i < n; i += 2) {
        const unsigned s0 = (unsigned)samples[i] & 0xFFF;
        const unsigned s1 = (i + 1 < n ? (unsigned)samples[i + 1] : 0u) & 0xFFF;
        uint8_t *b = packed + (i / 2) * 3;
        b[0] = (uint8_t)(s0 & 0xFF);
        b[1] = (uint8_t)((s0 >> 8) | ((s1 & 0xF) << 4));
        b[2] = (uint8_t)(s1 >> 4);
    }
}

void ddc_unpack_adc12(const uint8_t *packed, size_t n, int16_t *samples)
{
    for (size_t i = 0; i < n; i += 2) {
        const uint8_t *b = packed + (i / 2) * 3;
        const int s0 = b[0] | ((b[1] & 0x0F) << 8);
        const int s1 = (b[1] >> 4) | (b[2] << 4);
        samples[i] = (int16_t)((s0 ^ 0x800) - 0x800);
        if (i + 1 < n) samples[i + 1] = (int16_t)((s1 ^ 0x800) - 0x800);
    }
}

void ddc_synth_adc(uint32_t channel, uint32_t fs_adc, size_t n, int16_t *samples)
{
    if (fs_adc == 0) fs_adc = DDC_ADC_CLOCK_DEFAULT;
    uint64_t s = 0x5EED0000ull + channel;
    const double two_pi = 6.283185307179586476925286766559;
    const double f1 = 1.0e6 + 48.0e3 * channel, f2 = f1 + 1.0e3;
    const double w1 = two_pi * f1 / fs_adc, w2 = two_pi * f2 / fs_adc;
    for (size_t i = 0; i < n; i++) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const uint64_t r = s * 0x2545F4914F6CDD1Dull;
        const int noise = (int)(r >> 57) - 64; /* uniform in [-64, 63] */
        long v = std::lround(1400.0 * std::sin(w1 * (double)i) + 300.0 * std::sin(w2 * (double)i)) + noise;
        if (v > 2047) v = 2047;
        if (v < -2047) v = -2047;
        samples[i] = (int16_t)v;
    }
}


int ddc_synth_adc_dev(uint32_t first_channel, uint32_t n_channels, uint32_t fs_adc, size_t samples, uint8_t *rows,
                      size_t row_bytes, void *hip_stream)
{
    if (!rows || n_channels == 0) return fail(nullptr, -EINVAL, "null buffer");
    if (samples == 0 || samples % ddc::kSynthSeg) return fail(nullptr, -EINVAL, "samples must be a multiple of 512");
    if (row_bytes < samples / 2 * 3) return fail(nullptr, -EINVAL, "row_bytes smaller than one row of samples");
    // one thread per 512-sample segment, 256 per workgroup, and the workgroup count is a 32-bit number
    if (samples / ddc::kSynthSeg > 0xFFFFFFFFull || (unsigned long long)n_channels * (samples / ddc::kSynthSeg) >= (1ull << 39))
        return fail(nullptr, -EINVAL, "too many samples (at most 2^39 segments of 512 samples per call)");
    if (fs_adc == 0) fs_adc = DDC_ADC_CLOCK_DEFAULT;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint32_t n_seg = (uint32_t)(samples / ddc::kSynthSeg);
    uint64_t *states = nullptr;
    HIP_TRY(nullptr, hipMalloc(&states, (size_t)n_channels * n_seg * sizeof(uint64_t)));
    ddc::ddc_synth_states<<<(n_channels + 63) / 64, 64, 0, st>>>(first_channel, n_channels, n_seg, states);
    const uint64_t items = (uint64_t)n_channels * n_seg;
    ddc::ddc_synth_rows<<<(uint32_t)((items + 255) / 256), 256, 0, st>>>(first_channel, n_channels, n_seg, (double)fs_adc,
                                                                       states, rows, row_bytes);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(states);
    if (e != hipSuccess) return fail(nullptr, -EIO, std::string("ddc_synth_adc_dev: ") + hipGetErrorString(e));
    return 0;
}

} /* extern "C" */
