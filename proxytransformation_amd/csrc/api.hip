// extern "C" surface of libproxyt_hip.so (include/proxyt.h) and the forward driver.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace ptx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- optional per-kernel timing (bench.py's roofline leg) -----------------------------------------
// One launch site of the forward can be bracketed by HIP events recorded on the stream the kernel
// is launched on; ptx_timing_read() returns launches and summed milliseconds.  Off by default.
// img_pass2 / img_pass3 are the launch sites after the mean pass: k_img_pool (no third launch) for bf16 / fp16
// features of the path's shape, k_img_scores / k_img_gather for fp32, k_img_scores16 / k_img_gather16 otherwise.
// (k_gemm_nt[we]: from ~4000 images per call on and at head_dim 64; k_ln_rows[norm_img]: debug / stage API; k_attn32[...]: head_dim 64,
//  more than 256 proxies, few (scene, head) pairs -- every site is live on some shape; the gate sites are the one-wave fork / join
//  launches of the stream gates.)
static const char *const kKernelNames[] = {
    "k_minmax", "k_cluster", "k_select", "k_tags",
    "k_slot_net<pointnet>", "k_img_mean", "k_gemm_nt[qkv0]",
    "k_gemm_nt[we]", "img_pass2", "img_pass3", "k_gemm_nt[o]", "k_gemm_nt[c_proj]",
    "k_ln_rows[norm_img]", "k_gemm_nt[qkv+proxy_proj]", "k_gemm_nt[pp_img]", "k_attn32[proxy_as_query]",
    "k_attn32[proxy_as_key]", "k_gemm_nt[proj]", "k_gemm_nt[fc1]", "k_gemm_nt[fc2]",
    "k_heads", "k_affine<compact>", "k_proxy_attn[fused]", "k_mlp[fc1+gelu+fc2]",
    "k_gate[fork]", "k_signal[join]", "k_gate[join]", "k_gate[tags]", "k_signal[tags]"};
enum Kid : int {
    KID_MINMAX = 0, KID_CLUSTER, KID_SELECT, KID_TAGS, KID_POINTNET,
    KID_IMG_MEAN, KID_IMG_QKV0, KID_IMG_WE, KID_IMG_SCORES, KID_IMG_GATHER, KID_IMG_O, KID_IMG_C,
    KID_IMG_LN, KID_BLK_QKV, KID_BLK_PP, KID_BLK_ATTN_A, KID_BLK_ATTN_B, KID_BLK_PROJ, KID_BLK_FC1,
    KID_BLK_FC2, KID_BLK_HEADS, KID_AFFINE, KID_BLK_ATTN_F, KID_BLK_MLP,
    KID_GATE_FORK, KID_SIGNAL_JOIN, KID_GATE_JOIN, KID_GATE_TAGS, KID_SIGNAL_TAGS, KID_COUNT};
static_assert(sizeof(kKernelNames) / sizeof(kKernelNames[0]) == KID_COUNT, "kernel name table");

struct TimingRec { int kid; hipEvent_t a, b; };
struct TimingState {
    std::mutex mu;
    uint64_t mask = 0;                  // bit k = launch site k is bracketed by events
    int every = 1;                      // ... on every `every`-th launch of the site (ptx_timing_every)
    unsigned seen[64] = {};
    std::vector<TimingRec> pool;
    size_t used = 0;
};
static TimingState g_timing;

// Events attached to the kernel's own dispatch packet (hipExtLaunchKernel): begin / end of the kernel itself, what a kernel trace
// reports.  Events recorded around a launch are packets of their own: they add the dispatch latency of the kernel behind the first
// record and the second record's own turn (3-5 us on a 55 us kernel).  A launcher that supports it takes the armed pair.
static thread_local hipEvent_t g_ext_a = nullptr, g_ext_b = nullptr;
bool timing_ext_take(hipEvent_t *a, hipEvent_t *b)
{
    if (g_ext_a == nullptr) return false;
    *a = g_ext_a; *b = g_ext_b;
    g_ext_a = g_ext_b = nullptr;
    return true;
}

struct Timed {
    hipEvent_t stop = nullptr; hipStream_t st;
    Timed(int kid, hipStream_t s, bool ext = false) : st(s) {
        if (!((g_timing.mask >> kid) & 1u)) return;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;        // timing events have no place in a captured graph
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        std::lock_guard<std::mutex> lk(g_timing.mu);
        if (g_timing.seen[kid]++ % (unsigned)g_timing.every != 0) return;
        if (g_timing.used == g_timing.pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            g_timing.pool.push_back(TimingRec{kid, a, b});
        }
        TimingRec &pr = g_timing.pool[g_timing.used++];
        pr.kid = kid;
        if (ext) { g_ext_a = pr.a; g_ext_b = pr.b; return; }        // the launcher attaches both to the kernel's packet
        (void)hipEventRecord(pr.a, st);
        stop = pr.b;
    }
    ~Timed() {
        if (stop) (void)hipEventRecord(stop, st);
        if (g_ext_a != nullptr) {       // armed but not taken (a launcher without support): fall back to records around nothing
            (void)hipEventRecord(g_ext_a, st); (void)hipEventRecord(g_ext_b, st);
            g_ext_a = g_ext_b = nullptr;
        }
    }
};
#define PTX_TIMED(kid, st, call) do { ::ptx::Timed t_(kid, st); PTX_TRY(call); } while (0)
#define PTX_TIMED_EXT(kid, st, call) do { ::ptx::Timed t_(kid, st, true); PTX_TRY(call); } while (0)

static inline float attn_scale(int hd) { return (float)(1.0 / std::sqrt((double)hd)); }   // head_dim ** -0.5

int validate_shape(const PtxShape &s)
{
    const long M = (long)s.grid_size * s.grid_size * s.grid_size;
    PTX_REQUIRE(s.B >= 1 && s.N >= 1, "shape: B=%d N=%d", s.B, s.N);
    PTX_REQUIRE(s.grid_size >= 1 && M <= (1 << 20), "shape: grid_size=%d", s.grid_size);
    PTX_REQUIRE(s.K >= 1 && s.K <= 63, "shape: num_sub K=%d must be in [1,63]", s.K);
    PTX_REQUIRE(s.Mk >= 1 && s.Mk <= s.Mt && s.Mt <= M, "shape: need 1 <= Mk=%d <= Mt=%d <= M=%ld", s.Mk, s.Mt, M);
    PTX_REQUIRE(s.C == 256 || s.C == 512, "shape: embed_dim=%d; supported: 256 (the reference, PRE:302) and 512", s.C);
    PTX_REQUIRE((s.heads == 4 || s.heads == 8 || s.heads == 16) && (s.C / s.heads == 32 || s.C / s.heads == 64),
                "shape: num_heads=%d with embed_dim=%d; the image-pool and attention kernels are built for 4, 8 or 16 heads of "
                "head_dim 32 or 64", s.heads, s.C);
    PTX_REQUIRE(s.Mt <= 4096, "shape: Mt=%d clusters after the empty-drop; the farthest point sampling holds at most 4096", s.Mt);
    PTX_REQUIRE(s.hidden >= 4 && s.hidden % 4 == 0, "shape: hidden=%d", s.hidden);
    PTX_REQUIRE(s.in_dim >= 64 && s.in_dim % 64 == 0 && s.in_dim <= 2048,
                "shape: in_dim=%d must be a multiple of 64, at most 2048", s.in_dim);
    PTX_REQUIRE(s.L >= 1 && s.V >= 1, "shape: L=%d V=%d", s.L, s.V);
    PTX_REQUIRE(s.img_dtype >= 0 && s.img_dtype <= 2, "shape: img_dtype=%d (0 fp32, 1 bf16, 2 fp16)", s.img_dtype);
    if (s.img_dtype == 0) PTX_REQUIRE(s.hw >= 4 && s.hw <= 256, "shape: H*W=%d (fp32 image features: 4..256 pixels)", s.hw);
    else PTX_REQUIRE(s.hw >= 8 && s.hw <= 256, "shape: H*W=%d (16-bit image features: 8..256 pixels)", s.hw);
    PTX_REQUIRE((long)s.Mk * s.K < (1l << 31) - 1, "shape: Mk*K overflows the ownership tag");
    return PTX_OK;
}

PrepLayout prep_layout(const PtxShape &s)
{
    PrepLayout P{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += align_up(n, 64); return r; };   // 256-B granules
    P.hd = s.C / s.heads;
    P.KT1 = s.in_dim + s.hw + 1;
    P.KT2p = (int)align_up((size_t)s.in_dim + s.hw + 1, 4);
    P.off_ab = take(2 * kSlotHidden); P.enc_ab = take(2 * (size_t)s.C);
    P.ttn_ab = take(6); P.itn_ab = take(18);
    P.posb_t = take((size_t)s.Mk * s.C); P.posb_i = take((size_t)s.Mk * s.C);
    P.w3 = take((size_t)3 * s.C * s.in_dim); P.b3 = take((size_t)3 * s.C);
    P.t1 = take((size_t)s.heads * P.KT1 * P.hd);
    P.t2 = take((size_t)s.heads * P.hd * P.KT2p);
    P.ppg_w = take((size_t)s.C * s.C); P.ppg_s = take(s.C); P.ppg_c = take(s.C);
    for (int i = 0; i < 2; ++i) {
        P.fc1g_w[i] = take((size_t)s.hidden * s.C); P.fc1g_s[i] = take(s.hidden); P.fc1g_c[i] = take(s.hidden);
        const size_t planes = mlp_fused_supported(s.C, s.hidden, 1, 0) ? (size_t)s.hidden * s.C * 3 / 2 : 0;   // bf16 x 3, in floats
        P.mlp_w1p[i] = take(planes); P.mlp_w2p[i] = take(planes);
    }
    for (int i = 0; i < 2; ++i) P.qkvb[i] = take((size_t)s.Mk * 3 * s.C);
    P.total = o;
    return P;
}

// ---- the per-shape layout decisions of ptx_forward (described there): pure functions of the shape and of PTX_LAYOUT, so that the
// workspace can leave out what a layout never touches
struct LayoutForce { int v[4]; };
static LayoutForce layout_force()
{
    LayoutForce f{{-1, -1, -1, -1}};
    if (const char *e = getenv("PTX_LAYOUT"))
        for (int i = 0; i < 4 && e[i] != '\0'; ++i) f.v[i] = e[i] == '0' ? 0 : (e[i] == '1' ? 1 : -1);
    return f;
}
struct LayoutChoice { double est_cluster, est_image; bool cluster_on_caller, img_late, early; LayoutForce lf; };
static LayoutChoice choose_layout(const PtxShape &S)
{
    static const LayoutForce lf = layout_force();
    LayoutChoice c{};
    c.lf = lf;
    const int B = S.B, Kd = S.Mt - S.Mk;
    const double img_mb = (double)B * S.V * S.in_dim * S.hw * (S.img_dtype == 0 ? 4.0 : 2.0) * 1e-6;
    c.est_cluster = 80.0 + 0.42 * Kd;
    c.est_image = 40.0 + (S.img_dtype == 0 ? 0.98 : 0.78) * img_mb;
    c.cluster_on_caller = lf.v[0] >= 0 ? lf.v[0] != 0 : c.est_cluster > c.est_image + 60.0;
    c.img_late = c.cluster_on_caller && (lf.v[2] >= 0 ? lf.v[2] != 0 : c.est_image + 60.0 < 0.8 * c.est_cluster);
    const double est_all = 1.8 * 12.0 * (double)B * S.Mt * S.C * S.C / 70e6;        // us (cfg4 at 6 scenes: 147; measured ~150)
    c.early = c.cluster_on_caller && (lf.v[1] >= 0 ? lf.v[1] != 0 : (Kd >= 128 && (long)B * S.Mk >= 1024 && est_all < 3.0 * 0.42 * Kd));
    return c;
}

WsLayout ws_layout(const PtxShape &s)
{
    WsLayout L{};
    const PrepLayout P = prep_layout(s);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    const size_t B = s.B, N = s.N, K = s.K, C = s.C;
    const size_t M = (size_t)s.grid_size * s.grid_size * s.grid_size;
    const size_t Mk = s.Mk, Mt = s.Mt, Kd = s.Mt - s.Mk;
    const size_t nimg = B * s.V, Lp = s.L > s.V ? s.L : s.V, R = B * Mk;
    L.zero_begin = o;
    L.mm_enc = take(B * 6 * 4);
    L.scene_acc = take(B * 2 * 4);
    L.tag = take(B * N * 4);
    L.fa_ticket = take(2 * B * s.heads * 4);
    L.mlp_ticket = take(mlp_ticket_bytes((int)R));
    L.mm_ticket = take(4);
    L.zero_bytes = o - L.zero_begin;
    L.minmax = take(B * 6 * 4);
    L.centers0 = take(B * M * 3 * 4); L.cluster1 = take(B * M * K * 3 * 4);
    L.offsets = take(B * M * 3 * 4);  L.centers = take(B * M * 3 * 4);
    L.idx2 = take(B * M * K * 4);     L.cluster2 = take(B * M * K * 3 * 4);
    L.pad_count = take(B * M * 4);
    L.order = take(B * Mt * 4); L.picks = take(B * (Kd ? Kd : 1) * 4); L.keep = take(B * Mk * 4); L.ksrc = take(B * Mk * 4);
    L.kcenter = take(B * Mk * 3 * 4); L.kcluster = take(B * Mk * K * 3 * 4); L.kidx = take(B * Mk * K * 4);
    L.drop_idx = take(B * (Kd ? Kd : 1) * K * 4);
    L.tile_counts = take(B * (size_t)cdiv(s.N, kTilePts) * 4);
    L.point_proxy = take(R * C * 4);
    for (int i = 0; i < 2; ++i) L.x_in[i] = take(R * C * 4);
    // the early-proxy tables (all Mt clusters through the slot network, LayerNorm1 and qkv): only where that layout can be chosen --
    // for ANY storage type of the features (the choice depends on it, the module's workspace key does not): 3.3 of 28 MB per scene
    // at the benchmark shape, where the image chain owns the caller's stream
    bool may_early = false;
    for (int dt = 0; dt < 3; ++dt) { PtxShape t = s; t.img_dtype = dt; may_early = may_early || choose_layout(t).early; }
    const size_t em = may_early ? 1 : 0;
    L.pp_all = take(em * B * Mt * C * 4); L.order_e = take(em * B * Mt * 4);
    for (int i = 0; i < 2; ++i) { L.xln_all[i] = take(em * B * Mt * C * 4); L.g_all[i] = take(em * B * Mt * 3 * C * 4); }
    L.fm = take(nimg * s.in_dim * 4); L.qkv0 = take(nimg * 3 * C * 4);
    L.we = take(nimg * s.heads * (size_t)P.KT1 * 4);
    // the pooling kernels' partials (k_img_pool / k_img_pool32) and the generic score / gather kernels' [g_h | a_h] rows are never
    // both in use for a shape: ONE region (r05: 4.6 of 35 MB per scene at the benchmark shape)
    {
        const size_t pool_b = img_pool_bytes((int)nimg, s.in_dim, P.KT2p - s.in_dim), gbuf_b = nimg * s.heads * (size_t)P.KT2p * 4;
        L.pool = L.gbuf = take(pool_b > gbuf_b ? pool_b : gbuf_b);
    }
    L.obuf = take(nimg * C * 4); L.cbuf = take(nimg * C * 4); L.img_proxy = take(nimg * C * 4);
    for (int i = 0; i < 2; ++i) {
        L.qkv[i] = take(R * 3 * C * 4); L.pt[i] = take(B * Lp * C * 4); L.pv[i] = take(B * Lp * C * 4);
        L.ao[i] = take(R * C * 4); L.x1[i] = take(R * C * 4); L.xn2[i] = take(0);          // (xn2: unused since the LayerNorm folds of r02)
        // (the hidden activations leave the CU only where the fused Mlp launch does not apply: 1 MB per scene and branch otherwise unused)
        L.hbuf[i] = take(mlp_fused_supported(s.C, s.hidden, (int)R, 0) ? 0 : R * (size_t)s.hidden * 4);
        L.x2[i] = take(R * C * 4); L.guide[i] = take(R * C * 4);
        L.head[i] = take(R * 9 * 4);
        L.lnp_x1[i] = take(R * (C / 32) * 2 * 4);
    }
    L.lnp_img = take(nimg * (C / 32) * 2 * 4);
    const size_t fsp = fattn_split_for(s.B, s.heads);
    L.fa_part = take(fsp > 1 ? 2 * B * s.heads * fsp * Mk * kFaPartRow * 4 : 0);
    L.mlp_part = take(mlp_fused_supported(s.C, s.hidden, (int)R, 0) ? mlp_part_bytes((int)R) : 0);
    L.total = o;
    return L;
}

// ---- library-owned streams / events of one caller context ------------------------------------------
// Every module instance owns one PtxContext (ptx_context_create): its side streams and fork / join events are
// private, so two instances driven from two host threads on one device never wait on each other's events.
// A NULL context in ptx_forward selects a process-wide per-device default whose enqueue section is serialised
// by a mutex (the ABI-3 behaviour made safe).
}  // namespace ptx
struct PtxContext {
    int dev = -1;
    hipStream_t st = nullptr, lo = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, aux = nullptr, tags = nullptr, early_a = nullptr, early_b = nullptr;
    // device words of the in-kernel fork / join ("gates", below); null or !gates_on: events.  Word 0: fork, 32: join,
    // 40 / 44: the two probe words, 48: poison (read by k_affine), the rest spare
    uint32_t *gate = nullptr; uint32_t gate_seq = 0;
    uint32_t *gate_err = nullptr;        // [host, pinned, device-mapped] sticky error word written by a waiter that timed out
    bool gates_on = false;               // cleared for good by the first gate that times out or by a failed probe
    bool probed = false; hipStream_t probed_st = nullptr;   // the caller stream the concurrency probe was run against
    bool lo_ok = false;                  // ... and the low-priority stream runs beside both (the slot tags may leave the chain)
    hipStream_t last_st = nullptr;       // caller stream of the latest forward (drained before the poison word is cleared)
    bool have_last = false;              // ... valid only once a forward has run (a null handle would mean the default stream)
    uint64_t gate_ticks = 0, probe_ticks = 0; int gate_trap = 0;
    std::mutex mu;                       // held for the whole enqueue section of a forward
};
namespace ptx {

// The device-word gates (k_gate below) need the two streams' kernels to run CONCURRENTLY: a waiting wave on one queue is released
// by a kernel on the other.  Anything that serialises kernel execution across queues -- counter collection (rocprofv3 --pmc sets
// ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS), thread trace, PC sampling, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING -- would leave
// the waiting wave alone on the device until its bound runs out (seconds per forward): there the events are used.  PTX_GATE=0
// forces the events, PTX_GATE=1 the gates.
static bool env_on(const char *name)
{
    const char *v = getenv(name);
    return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}
static bool gates_allowed()
{
    if (const char *g = getenv("PTX_GATE")) return atoi(g) != 0;
    for (const char *name : {"ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "ROCPROF_COUNTER_GROUPS", "ROCPROF_ADVANCED_THREAD_TRACE",
                             "ROCPROF_PC_SAMPLING_UNIT", "ROCPROF_PC_SAMPLING_METHOD", "AMD_SERIALIZE_KERNEL", "AMD_SERIALIZE_COPY",
                             "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING", "ROCPROFILER_METRICS_PATH", "ROCP_METRICS"})
        if (env_on(name)) return false;
    return true;
}

static int context_init(PtxContext *c)
{
    PTX_HIP(hipGetDevice(&c->dev));
    // highest priority: the latency-bound clustering chain runs here next to the long,
    // bandwidth-bound image passes on the caller's stream and must win work-group dispatch
    int lo = 0, hi = 0;
    PTX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PTX_HIP(hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, hi));
    PTX_HIP(hipStreamCreateWithPriority(&c->lo, hipStreamNonBlocking, lo));
    PTX_HIP(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    PTX_HIP(hipEventCreateWithFlags(&c->join, hipEventDisableTiming));
    PTX_HIP(hipEventCreateWithFlags(&c->aux, hipEventDisableTiming));
    PTX_HIP(hipEventCreateWithFlags(&c->tags, hipEventDisableTiming));
    PTX_HIP(hipEventCreateWithFlags(&c->early_a, hipEventDisableTiming));
    PTX_HIP(hipEventCreateWithFlags(&c->early_b, hipEventDisableTiming));
    if (gates_allowed()) {
        PTX_HIP(hipMalloc(reinterpret_cast<void **>(&c->gate), 256));
        PTX_HIP(hipMemset(c->gate, 0, 256));
        PTX_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->gate_err), 64, hipHostMallocMapped));
        c->gate_err[0] = 0u;
        int khz = 0;                    // s_memrealtime ticks per millisecond (100 MHz on gfx950)
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->dev) != hipSuccess || khz <= 0) khz = 100000;
        // The bound of a waiting wave.  The fork legitimately waits for everything the caller queued ahead of the forward on its
        // stream (seconds, in a training step), so the default is generous; PTX_GATE_TIMEOUT_MS overrides (tests: a few ms)
        const char *ms_env = getenv("PTX_GATE_TIMEOUT_MS");
        const long ms = ms_env ? atol(ms_env) : 30000;
        c->gate_ticks = (uint64_t)(ms > 0 ? ms : 1) * (uint64_t)khz;
        c->probe_ticks = (uint64_t)20 * (uint64_t)khz;          // probe: 20 ms on an idle pair of streams
#ifdef PTX_TEST_HOOKS
        c->gate_trap = env_on("PTX_GATE_TRAP") ? 1 : 0;    // test-hooks build only: a waiter that runs out of time also traps
#endif
        c->gates_on = true;
    }
    return PTX_OK;
}

static void context_release(PtxContext *c)
{
    if (c->gate) (void)hipFree(c->gate);
    if (c->gate_err) (void)hipHostFree(c->gate_err);
    c->gate = nullptr; c->gate_err = nullptr; c->gates_on = false;
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->aux) (void)hipEventDestroy(c->aux);
    if (c->tags) (void)hipEventDestroy(c->tags);
    if (c->early_a) (void)hipEventDestroy(c->early_a);
    if (c->early_b) (void)hipEventDestroy(c->early_b);
    if (c->st) (void)hipStreamDestroy(c->st);
    if (c->lo) (void)hipStreamDestroy(c->lo);
    c->fork = c->join = c->aux = c->tags = nullptr; c->st = c->lo = nullptr;
}

static std::mutex g_default_mu;
static PtxContext *g_default_ctx[16] = {};

static int default_context(PtxContext **out)
{
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    PTX_REQUIRE(dev >= 0 && dev < 16, "device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default_ctx[dev] == nullptr) {
        PtxContext *c = new PtxContext();
        const int rc = context_init(c);
        if (rc != PTX_OK) { context_release(c); delete c; return rc; }
        g_default_ctx[dev] = c;
    }
    *out = g_default_ctx[dev];
    return PTX_OK;
}

int make_scene_pts(const float *stacked, const float *const *list, int B, int N, ScenePts *out)
{
    PTX_REQUIRE(B >= 1 && B <= kMaxScenes, "at most %d scenes per call (got %d): split the batch", kMaxScenes, B);
    PTX_REQUIRE(stacked || list, "null points");
    for (int b = 0; b < kMaxScenes; ++b) out->p[b] = nullptr;
    for (int b = 0; b < B; ++b) {
        out->p[b] = list ? list[b] : stacked + (size_t)b * N * 3;
        PTX_REQUIRE(out->p[b] != nullptr, "null point cloud for scene %d", b);
    }
    return PTX_OK;
}

template <typename T>
static inline T *at(void *base, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(base) + off); }

// ---- image chain ---------------------------------------------------------------------------------
// Image chain (PRE:335-342).  phase 0: everything; 1: only the first streaming pass (image means);
// 2: everything after it.
// need_ln: also write the normalised proxies (stage API, debug); the forward consumes c_proj's raw rows through
// the LayerNorm fold of the image block's proxy_proj GEMM (partials L.lnp_img) and skips that launch.
// ---- gates: the fork and the join of the two chains through device words instead of event record + wait (r03) ---------------
// An event record is a packet of its own on the recording stream and the wait a barrier packet on the other: each leaves ~5 us
// of idle between the two kernels around it -- on the caller's stream once per forward in front of the mean pass (fork) and once
// in front of the proxy blocks (join).  Instead: the FIRST kernel of the image chain (k_img_mean16 / k_img_mean) stores the
// forward's sequence number into a word when its first thread starts -- which, the stream being in order, is when everything
// the caller enqueued before the forward has completed -- and the clustering stream starts with k_gate, ONE wave that polls the
// word (s_sleep between polls) and ends when it has arrived: the kernels behind it start exactly when they would have behind
// the event wait.  The join is the mirror image: k_signal behind the clustering stream's last kernel (so its stores have been
// released), k_gate on the caller's stream in front of the proxy blocks.  Kernel boundaries do the releasing and acquiring as
// before; the words only order the two queues.  The waiting wave is bounded (~4 s, then it lets go) so that a forward whose
// first kernel never runs cannot wedge the device, and the gates are not used where kernels are serialised (gates_allowed()).  Interleaved A/B on one box, 4 scenes per GPU: 17.71k (events) -> 17.97k (fork)
// -> 18.39k (fork + join) scenes/s, 0.2258 -> 0.2176 ms per step; neutral at 32.  Used when the image chain owns the caller's
// stream (the benchmark shapes); PTX_GATE=0 restores the events everywhere.
__global__ void k_signal(uint32_t *flag, uint32_t seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave that ends when the word has arrived -- or, after the bound, reports the failure (gate_wait, common.h): it never lets go silently
__global__ void k_gate(GateRef g)
{
    if (threadIdx.x == 0) gate_wait(g);
}

static GateRef gate_ref(const PtxContext *c, int word, uint32_t seq, uint32_t site, bool probe = false)
{
    // A JOIN (sites 2, 7) gets twice the bound of a fork (1, 6): it may have to sit through a fork that runs into its own bound and
    // then through the chain behind it -- with equal bounds the join of a forward whose fork word never arrives lets go FIRST and the
    // post-join kernels consume a workspace the other chain has not written yet
    const uint64_t ticks = probe ? c->probe_ticks : ((site == 2 || site == 7) ? 2 * c->gate_ticks : c->gate_ticks);
    return GateRef{c->gate + word, seq, c->gate_err, probe ? nullptr : c->gate + 48, ticks, site, probe ? 0 : c->gate_trap};
}

// A gate of a previous forward timed out (or the probe failed): report it ONCE as PTX_EGATE, switch the context to events for good.
// The forward whose gate failed had its outputs turned into NaN by k_affine (poison word), so nothing plausible-looking survives.
static int gate_check(PtxContext *c)
{
    if (c->gate_err == nullptr) return PTX_OK;
    const uint32_t e = *reinterpret_cast<volatile uint32_t *>(c->gate_err);
    if (e == 0u) return PTX_OK;
    static const char *const site_name[] = {"?", "fork (clustering stream waiting for the caller's stream)",
                                            "join (caller's stream waiting for the other chain)", "probe, fork direction",
                                            "probe, join direction", "?", "slot tags' fork (tag stream waiting for the selection)",
                                            "slot tags' join (caller's stream waiting for the tags)"};
    const unsigned site = (e >> 24) & 0x7fu;
    c->gates_on = false;
    *reinterpret_cast<volatile uint32_t *>(c->gate_err) = 0u;
    // the streams may still be running the poisoned forward (its k_affine has to see the word): drain them, then clear
    (void)hipStreamSynchronize(c->st);
    if (c->have_last) (void)hipStreamSynchronize(c->last_st);
    (void)hipMemset(c->gate + 48, 0, 4);
    set_error("stream gate timed out at the %s of forward #%u: the two chains of that forward were not ordered and its outputs "
              "were set to NaN; this context now orders its streams with events (PTX_GATE=0 selects them from the start)",
              site_name[site < 8 ? site : 0], e & 0xffffffu);
    return PTX_EGATE;
}

// The gates need the caller's stream and the context's side stream to make progress side by side.  HIP maps streams onto a small
// pool of hardware queues: two streams that share one run their kernels in enqueue order, and a waiting wave enqueued in front of
// the kernel that releases it would sit there for its whole bound.  Checked once per (context, caller stream), in both directions,
// with the WAITER enqueued first: if either wave reports a timeout (20 ms; the streams are drained first so nothing else is in
// the way) the context uses events.
static int gate_probe_pair(PtxContext *c, hipStream_t waiter, hipStream_t releaser, int word, uint32_t site, bool *ok)
{
    const uint32_t seq = ++c->gate_seq;
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, waiter, gate_ref(c, word, seq, site, true));
    PTX_LAUNCHED("k_gate[probe]");
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, releaser, c->gate + word, seq);
    PTX_LAUNCHED("k_signal[probe]");
    PTX_HIP(hipStreamSynchronize(waiter));
    PTX_HIP(hipStreamSynchronize(releaser));
    *ok = *reinterpret_cast<volatile uint32_t *>(c->gate_err) == 0u;
    *reinterpret_cast<volatile uint32_t *>(c->gate_err) = 0u;
    return PTX_OK;
}
static int gate_probe(PtxContext *c, hipStream_t st)
{
    c->probed = true; c->probed_st = st; c->lo_ok = false;
    if (!c->gates_on) return PTX_OK;
    PTX_HIP(hipStreamSynchronize(st));
    PTX_HIP(hipStreamSynchronize(c->st));
    PTX_HIP(hipStreamSynchronize(c->lo));
    bool fork_ok = false, join_ok = false, t1 = false, t2 = false;
    PTX_TRY(gate_probe_pair(c, c->st, st, 40, 3, &fork_ok));        // side waits for the caller
    PTX_TRY(gate_probe_pair(c, st, c->st, 44, 4, &join_ok));        // caller waits for the side
    if (!fork_ok || !join_ok) { c->gates_on = false; return PTX_OK; }      // silent by design: events are the correct fallback, not an error
    PTX_TRY(gate_probe_pair(c, c->lo, c->st, 40, 3, &t1));          // tag stream waits for the side stream
    PTX_TRY(gate_probe_pair(c, st, c->lo, 44, 4, &t2));             // caller waits for the tag stream
    c->lo_ok = t1 && t2;
    return PTX_OK;
}

static int run_img_proxy(const PtxShape &s, const PtxWeights &w, const float *prep, const void *img_any,
                         float *img_proxy, void *ws, hipStream_t st, int phase = 0, bool need_ln = true,
                         int i0 = 0, int ni = -1, uint32_t *gate = nullptr, uint32_t gate_seq = 0)
{
    // images [i0, i0 + ni) of the B * V of this call (default: all): every buffer of the chain is per image
    const PrepLayout P = prep_layout(s);
    const WsLayout L = ws_layout(s);
    const int C = s.C, hd = P.hd, nall = s.B * s.V, nimg = ni < 0 ? nall : ni;
    float *fm = at<float>(ws, L.fm) + (size_t)i0 * s.in_dim, *qkv0 = at<float>(ws, L.qkv0) + (size_t)i0 * 3 * C;
    float *we = at<float>(ws, L.we) + (size_t)i0 * s.heads * P.KT1, *gbuf = at<float>(ws, L.gbuf) + (size_t)i0 * s.heads * P.KT2p;
    float *obuf = at<float>(ws, L.obuf) + (size_t)i0 * C, *cbuf = at<float>(ws, L.cbuf) + (size_t)i0 * C;
    if (img_proxy) img_proxy += (size_t)i0 * C;
    const int dt = s.img_dtype;
    img_any = static_cast<const char *>(img_any) + (size_t)i0 * s.in_dim * s.hw * (dt == 0 ? 4 : 2);
    const float *img = static_cast<const float *>(img_any);
    float *Gs = nullptr, *E = nullptr, *ML = nullptr;
    const int EW = P.KT2p - s.in_dim;
    const bool pooled32 = img_pool32_supported(dt, s.in_dim, s.hw, s.heads);        // r05: fp32 features in two passes, not three
    const bool pooled = pooled32 || img_pool_supported(dt, s.in_dim, s.hw, s.heads);
    if (pooled) {
        img_pool_layout(at<float>(ws, L.pool), nall, s.in_dim, EW, &Gs, &E, &ML);
        Gs += (size_t)i0 * 2 * s.heads * s.in_dim; E += (size_t)i0 * s.heads * EW; ML += (size_t)i0 * s.heads * 5;
    }
    if (phase != 2) {
        if (dt == 0) PTX_TIMED(KID_IMG_MEAN, st, launch_img_mean(img, nimg, s.in_dim, s.hw, fm, st, gate, gate_seq));
        else PTX_TIMED(KID_IMG_MEAN, st, launch_img_mean16(img_any, dt, nimg, s.in_dim, s.hw, fm, st, gate, gate_seq));
    }
    if (phase == 1) return PTX_OK;
    // head_dim 32: a 32-column tile of the qkv0 GEMM IS one head's q, and the work-group that finishes it goes on to that
    // head's [w_h | e_h] = q_h T1_h^T (GemmProb::w2): one launch (and one boundary) less on the image chain
    // (from ~4000 images per call on -- 32 scenes of 196 views -- the two products as launches of their own, the first on the
    //  64 x 64 split-operand kernel, are faster than the latency-regime chained form: 80 + 72 vs 185 us)
    const bool chained = hd == 32 && nimg < 4096;
    {   // [q | k0 | v0] of token 0 = W3 mean(f) + b3
        GemmBatch g{}; g.n = 1;
        g.p[0] = GemmProb{fm, prep + P.w3, qkv0, prep + P.b3, nullptr, nullptr, nullptr,
                          nimg, 3 * C, s.in_dim, s.in_dim, s.in_dim, 3 * C, 0, 0, 0, EPI_NONE};
        if (chained) {
            g.p[0].w2 = prep + P.t1; g.p[0].c2 = we; g.p[0].n2 = P.KT1; g.p[0].ldc2 = s.heads * P.KT1;
            g.p[0].chain_tiles = s.heads; g.p[0].w2_stride = (long)P.KT1 * hd; g.p[0].c2_stride = P.KT1;
        }
        PTX_TIMED(KID_IMG_QKV0, st, launch_gemm(g, st));
    }
    // (a GEMM batch holds up to kMaxGroups = 8 problems: 16 heads go as two launches)
    if (!chained) {   // per head: [w_h | e_h] = q_h T1_h^T
        for (int h0 = 0; h0 < s.heads; h0 += kMaxGroups) {
            GemmBatch g{}; g.n = std::min(kMaxGroups, s.heads - h0);
            for (int h = h0; h < h0 + g.n; ++h)
                g.p[h - h0] = GemmProb{qkv0 + h * hd, prep + P.t1 + (size_t)h * P.KT1 * hd, we + (size_t)h * P.KT1,
                                       nullptr, nullptr, nullptr, nullptr, nimg, P.KT1, hd, 3 * C, hd,
                                       s.heads * P.KT1, 0, 0, 0, EPI_NONE};
            PTX_TIMED(KID_IMG_WE, st, launch_gemm(g, st));
        }
    }
    if (pooled32) {
        PTX_TIMED_EXT(KID_IMG_SCORES, st, launch_img_pool32(img, we, qkv0, nimg, s.in_dim, s.hw, C, P.KT1, EW, attn_scale(hd), Gs, E, ML, st));
    } else if (dt == 0) {
        PTX_TIMED(KID_IMG_SCORES, st, launch_img_scores(img, we, qkv0, nimg, s.in_dim, s.hw, s.heads, C, P.KT1,
                                                        P.KT2p, attn_scale(hd), gbuf, st));
        PTX_TIMED(KID_IMG_GATHER, st, launch_img_gather(img, nimg, s.in_dim, s.hw, s.heads, P.KT2p, gbuf, st));
    } else if (pooled) {
        PTX_TIMED_EXT(KID_IMG_SCORES, st, launch_img_pool(img_any, dt, we, qkv0, nimg, s.in_dim, s.hw, C, P.KT1,
                                                      EW, attn_scale(hd), Gs, E, ML, st));
    } else {
        PTX_TIMED(KID_IMG_SCORES, st, launch_img_scores16(img_any, dt, we, qkv0, nimg, s.in_dim, s.hw, s.heads, C,
                                                          P.KT1, P.KT2p, attn_scale(hd), gbuf, st));
        PTX_TIMED(KID_IMG_GATHER, st, launch_img_gather16(img_any, dt, nimg, s.in_dim, s.hw, s.heads, P.KT2p, gbuf, st));
    }
    for (int h0 = 0; h0 < s.heads; h0 += kMaxGroups) {   // per head: o_h = [g_h | a_h] T2_h^T + a_h(0) v0_h + bv_h
        GemmBatch g{}; g.n = std::min(kMaxGroups, s.heads - h0);
        for (int h = h0; h < h0 + g.n; ++h) {
            GemmProb &gp = g.p[h - h0];
            gp = GemmProb{gbuf + (size_t)h * P.KT2p, prep + P.t2 + (size_t)h * hd * P.KT2p, obuf + h * hd,
                          w.v_b + h * hd, nullptr, gbuf + (size_t)h * P.KT2p + s.in_dim,
                          qkv0 + 2 * C + h * hd, nimg, hd, P.KT2p, s.heads * P.KT2p, P.KT2p, C,
                          0, s.heads * P.KT2p, 3 * C, EPI_NONE};
            if (pooled) {       // [g_h | a_h] is merged from the pooling tiles while it is loaded
                gp.A = nullptr; gp.rs = nullptr;
                gp.pg = Gs + (size_t)h * s.in_dim; gp.ldg = 2 * s.heads * s.in_dim; gp.gslab = s.heads * s.in_dim;
                gp.pe = E + (size_t)h * EW; gp.lde = s.heads * EW;
                gp.pml = ML + (size_t)h * 5; gp.ldml = s.heads * 5;
                gp.kg = s.in_dim;
            }
        }
        PTX_TIMED(KID_IMG_O, st, launch_gemm(g, st));
    }
    {   // c_proj
        GemmBatch g{}; g.n = 1;
        g.p[0] = GemmProb{obuf, w.c_w, cbuf, w.c_b, nullptr, nullptr, nullptr,
                          nimg, C, C, C, C, C, 0, 0, 0, EPI_NONE};
        g.p[0].lnp_out = at<float>(ws, L.lnp_img) + (size_t)i0 * (C / 32) * 2;
        PTX_TIMED(KID_IMG_C, st, launch_gemm(g, st));
    }
    if (need_ln) {
        LnBatch lb{}; lb.n = 1; lb.C = C; lb.eps = s.ln_eps;
        lb.p[0] = LnProb{cbuf, img_proxy, w.norm_img_w, w.norm_img_b, nullptr, nimg, 1};
        PTX_TIMED(KID_IMG_LN, st, launch_ln_rows(lb, st));
    }
    return PTX_OK;
}

// ---- proxy blocks (one or both branches in the same launches) --------------------------------------
struct Branch {
    const PtxBlock *blk; const float *x_in; const float *proxy; int Lp; const uint8_t *mask;
    const float *head_w, *head_b, *head_ab; int nout; float *head_out; float *guide; int slot;
    bool late_proxy;     // the proxies of this branch are produced on the side stream (image proxies)
    // proxies given as RAW rows + LayerNorm partials (the forward's image branch): proxy_proj folds the LayerNorm
    const float *proxy_lnp, *pp_gw, *pp_gs, *pp_gc;
    const float *fc1_gw, *fc1_gs, *fc1_gc;      // norm2 folded into fc1
    const float *prep;                          // the parameter tables (weight planes of the fused Mlp)
};

// One work-group per (scene, head, branch) must be enough parallelism: the two-launch form spreads the query tiles of the
// second contraction over the chip, which wins when a call has few scenes and many tokens (measured).
static bool fused_attn_pays(const PtxShape &s, const Branch *br, int nb)
{
    (void)br;
    const long wgs = (long)s.B * s.heads * nb;
    return s.Mk <= 256 || wgs >= 96;
}

// phase 0: everything; phase 1: only the projections that do not wait for late proxies
// (qkv of every branch + proxy_proj of the early ones); phase 2: the rest.
// join (phase 2 only, may be null): the cross-stream join folded into the first launch of the phase -- the proxy_proj GEMM of the
// late branch needs nothing from the other stream, its work-group (0,0,0) ends with the wait, and the attention kernel behind
// it starts when both the GEMM and the other stream are done (no k_gate launch on the critical path: -4.8 us in the r03 trace).
// *join is cleared when it has been attached.
static int run_blocks(const PtxShape &s, const Branch *br, int nb, const float *point_proxy, void *ws,
                      hipStream_t st, int phase = 0, int cd = 0, GateRef *join = nullptr, GateRef *tags_join = nullptr,
                      bool skip_qkv = false)
{
    const WsLayout L = ws_layout(s);
    const int C = s.C, R = s.B * s.Mk;
    {   // qkv = Linear(C,3C)(LN1(x)+bias) (PRE:221);  proxy_tokens = proxy_proj(proxy) (PRE:223)
        GemmBatch g{}; g.n = 0;
        for (int i = 0; i < nb; ++i) {
            const int sl = br[i].slot;
            if (phase != 2 && phase != 3 && phase != 4 && !skip_qkv)
                g.p[g.n++] = GemmProb{br[i].x_in, br[i].blk->qkv_w, at<float>(ws, L.qkv[sl]), br[i].blk->qkv_b,
                                      nullptr, nullptr, nullptr, R, 3 * C, C, C, C, 3 * C, 0, 0, 0, EPI_NONE};
            // phase 3: only the late proxies' projection, then return; phase 4: every input projection has been enqueued already
            const bool now = phase == 0 || (phase == 1 && !br[i].late_proxy) || ((phase == 2 || phase == 3) && br[i].late_proxy);
            if (now && phase != 4) {
                GemmProb &q = g.p[g.n++];
                q = GemmProb{br[i].proxy, br[i].blk->pp_w, at<float>(ws, L.pt[sl]), br[i].blk->pp_b,
                             nullptr, nullptr, nullptr, s.B * br[i].Lp, C, C, C, C, C, 0, 0, 0, EPI_NONE};
                if (br[i].proxy_lnp != nullptr) {       // LayerNorm(norm_img) folded into this GEMM
                    q.W = br[i].pp_gw; q.bias = nullptr;
                    q.lnp_in = br[i].proxy_lnp; q.ln_s = br[i].pp_gs; q.ln_c = br[i].pp_gc;
                    q.ln_parts = C / 32; q.ln_C = C; q.ln_eps = s.ln_eps;
                }
            }
        }
        if (g.n > 0 && (phase == 2 || phase == 1) && join != nullptr && join->flag != nullptr) { g.tail_gate = *join; join->flag = nullptr; }
        if (g.n > 0) PTX_TIMED(phase == 2 || phase == 3 ? KID_BLK_PP : KID_BLK_QKV, st, launch_gemm(g, st, cd));
        if (phase == 1 || phase == 3) return PTX_OK;
    }
    FAttnBatch fa{}; fa.nb = nb; fa.B = s.B; fa.heads = s.heads; fa.hd = C / s.heads; fa.n = s.Mk; fa.C = C;
    fa.scale = attn_scale(C / s.heads); fa.compute_dtype = cd;
    fa.split = fattn_split_for(s.B, s.heads); fa.part = at<float>(ws, L.fa_part); fa.tickets = at<int>(ws, L.fa_ticket);
    for (int i = 0; i < nb; ++i) {
        const int sl = br[i].slot;
        fa.p[i] = FAttnProb{at<float>(ws, L.qkv[sl]), at<float>(ws, L.pt[sl]), br[i].mask, at<float>(ws, L.ao[sl]), br[i].Lp};
    }
    // head_dim 32: both contractions of a (scene, head, branch) in one work-group, PV never leaves the CU (fattn.hip)
    // (reduced-precision mode: always the fused kernel where it exists -- the two-launch form is fp32 only)
    const bool fused = fused_attn_supported(fa) && (cd == 1 || fused_attn_pays(s, br, nb));
    if (fused) PTX_TIMED(KID_BLK_ATTN_F, st, launch_proxy_attn(fa, st));
    AttnBatch a{}; a.n = nb; a.B = s.B; a.heads = s.heads; a.hd = C / s.heads; a.scale = attn_scale(C / s.heads);
    for (int i = 0; i < nb && !fused; ++i) {   // proxy as query (PRE:232-238): no mask
        const int sl = br[i].slot;
        float *qkv = at<float>(ws, L.qkv[sl]);
        a.p[i] = AttnProb{at<float>(ws, L.pt[sl]), qkv + C, qkv + 2 * C, at<float>(ws, L.pv[sl]), nullptr,
                          br[i].Lp, s.Mk, C, 3 * C, 3 * C, C,
                          (long)br[i].Lp * C, (long)s.Mk * 3 * C, (long)s.Mk * 3 * C, (long)br[i].Lp * C};
    }
    if (!fused) PTX_TIMED(KID_BLK_ATTN_A, st, launch_attn32(a, st));
    for (int i = 0; i < nb && !fused; ++i) {   // proxy as key (PRE:241-250): padded text tokens masked
        const int sl = br[i].slot;
        float *qkv = at<float>(ws, L.qkv[sl]);
        a.p[i] = AttnProb{qkv, at<float>(ws, L.pt[sl]), at<float>(ws, L.pv[sl]), at<float>(ws, L.ao[sl]),
                          br[i].mask, s.Mk, br[i].Lp, 3 * C, C, C, C,
                          (long)s.Mk * 3 * C, (long)br[i].Lp * C, (long)br[i].Lp * C, (long)s.Mk * C};
    }
    if (!fused) PTX_TIMED(KID_BLK_ATTN_B, st, launch_attn32(a, st));
    {   // x1 = x + proj(attn) (PRE:255, 274); every tile also leaves the LayerNorm partials of its rows for fc1
        GemmBatch g{}; g.n = nb;
        for (int i = 0; i < nb; ++i) {
            const int sl = br[i].slot;
            g.p[i] = GemmProb{at<float>(ws, L.ao[sl]), br[i].blk->proj_w, at<float>(ws, L.x1[sl]),
                              br[i].blk->proj_b, point_proxy, nullptr, nullptr, R, C, C, C, C, C, C, 0, 0, EPI_NONE};
            g.p[i].lnp_out = at<float>(ws, L.lnp_x1[sl]);
        }
        // the slot tags (needed by k_affine only) come from a third stream: their join rides on this launch
        if (tags_join != nullptr && tags_join->flag != nullptr) { g.tail_gate = *tags_join; tags_join->flag = nullptr; }
        PTX_TIMED(KID_BLK_PROJ, st, launch_gemm(g, st, cd));
    }
    if (mlp_fused_supported(C, s.hidden, R, cd)) {
        // x2 = x1 + fc2(GELU(fc1(norm2(x1)))) in one launch: 32 rows x a slice of 256 hidden units per work-group, the hidden
        // activations stay in LDS, the four slices of a row tile are summed by the last to arrive (mlp.hip)
        const PrepLayout P = prep_layout(s);
        MlpBatch mb{}; mb.n = nb; mb.ln_eps = s.ln_eps; mb.compute_dtype = cd;
        mb.part = at<float>(ws, L.mlp_part); mb.tickets = at<int>(ws, L.mlp_ticket);
        for (int i = 0; i < nb; ++i) {
            const int sl = br[i].slot;
            mb.p[i] = MlpProb{at<float>(ws, L.x1[sl]), at<float>(ws, L.lnp_x1[sl]), br[i].prep + P.mlp_w1p[sl],
                              br[i].prep + P.mlp_w2p[sl], br[i].fc1_gs, br[i].fc1_gc, br[i].blk->fc2_b,
                              at<float>(ws, L.x2[sl]), R,
                              br[i].blk->out_norm_w, br[i].blk->out_norm_b, br[i].head_w, br[i].head_b, br[i].head_ab,
                              br[i].head_out, br[i].guide, br[i].nout};
        }
        PTX_TIMED(KID_BLK_MLP, st, launch_mlp(mb, st));
        return PTX_OK;                  // the output heads ran on the finished rows inside that launch
    } else {
    {   // h = GELU(fc1(norm2(x1))) (PRE:275): norm2 folded into the GEMM (W1 diag(gamma), row statistics in the epilogue)
        GemmBatch g{}; g.n = nb;
        for (int i = 0; i < nb; ++i) {
            const int sl = br[i].slot;
            g.p[i] = GemmProb{at<float>(ws, L.x1[sl]), br[i].fc1_gw, at<float>(ws, L.hbuf[sl]),
                              nullptr, nullptr, nullptr, nullptr, R, s.hidden, C, C, C, s.hidden,
                              0, 0, 0, EPI_GELU};
            g.p[i].lnp_in = at<float>(ws, L.lnp_x1[sl]); g.p[i].ln_s = br[i].fc1_gs; g.p[i].ln_c = br[i].fc1_gc;
            g.p[i].ln_parts = C / 32; g.p[i].ln_C = C; g.p[i].ln_eps = s.ln_eps;
        }
        PTX_TIMED(KID_BLK_FC1, st, launch_gemm(g, st, cd));
    }
    {   // x2 = x1 + fc2(h)
        GemmBatch g{}; g.n = nb;
        for (int i = 0; i < nb; ++i) {
            const int sl = br[i].slot;
            g.p[i] = GemmProb{at<float>(ws, L.hbuf[sl]), br[i].blk->fc2_w, at<float>(ws, L.x2[sl]),
                              br[i].blk->fc2_b, at<float>(ws, L.x1[sl]), nullptr, nullptr, R, C, s.hidden,
                              s.hidden, s.hidden, C, C, 0, 0, EPI_NONE};
        }
        PTX_TIMED(KID_BLK_FC2, st, launch_gemm(g, st, cd));
    }
    }
    HeadBatch hb{}; hb.n = nb; hb.C = C; hb.eps = s.ln_eps;
    for (int i = 0; i < nb; ++i) {
        const int sl = br[i].slot;
        hb.p[i] = HeadProb{at<float>(ws, L.x2[sl]), br[i].blk->out_norm_w, br[i].blk->out_norm_b,
                           br[i].head_w, br[i].head_b, br[i].head_ab, br[i].head_out, br[i].guide, R, br[i].nout};
    }
    PTX_TIMED(KID_BLK_HEADS, st, launch_heads(hb, st));
    return PTX_OK;
}

static Branch make_branch(const PtxShape &s, const PtxWeights &w, const float *prep, int which,
                          const float *x_in, const float *proxy, int Lp, const uint8_t *mask,
                          float *head_out, float *guide)
{
    const PrepLayout P = prep_layout(s);
    Branch b{};
    b.blk = which == 0 ? &w.text : &w.img;
    b.x_in = x_in; b.proxy = proxy; b.Lp = Lp; b.mask = mask;
    b.head_w = which == 0 ? w.text_trans_w : w.img_trans_w;
    b.head_b = which == 0 ? w.text_trans_b : w.img_trans_b;
    b.head_ab = prep + (which == 0 ? P.ttn_ab : P.itn_ab);
    b.nout = which == 0 ? 3 : 9;
    b.head_out = head_out; b.guide = guide; b.slot = which; b.late_proxy = which == 1;
    b.proxy_lnp = nullptr; b.pp_gw = prep + P.ppg_w; b.pp_gs = prep + P.ppg_s; b.pp_gc = prep + P.ppg_c;
    b.fc1_gw = prep + P.fc1g_w[which]; b.fc1_gs = prep + P.fc1g_s[which]; b.fc1_gc = prep + P.fc1g_c[which];
    b.prep = prep;
    return b;
}

static int check_bufs(const PtxShape *s, const void *ws, size_t ws_bytes)
{
    PTX_REQUIRE(s != nullptr, "null shape");
    PTX_TRY(validate_shape(*s));
    PTX_REQUIRE(ws != nullptr, "null workspace");
    const size_t need = ws_layout(*s).total;
    if (ws_bytes < need) { set_error("workspace too small: %zu < %zu bytes", ws_bytes, need); return PTX_ENOSPACE; }
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

int ptx_abi_version(void) { return PTX_ABI_VERSION; }

int ptx_kernel_count(void) { return KID_COUNT; }
const char *ptx_kernel_name(int kid) { return kid >= 0 && kid < KID_COUNT ? kKernelNames[kid] : nullptr; }

int ptx_timing_select(int kid)
{
    PTX_REQUIRE(kid >= -1 && kid < KID_COUNT, "ptx_timing_select: kid=%d", kid);
    std::lock_guard<std::mutex> lk(g_timing.mu);
    g_timing.mask = kid < 0 ? 0 : (1ull << kid);
    g_timing.used = 0;
    for (unsigned &s : g_timing.seen) s = 0;
    return PTX_OK;
}

// An event record is a packet of its own in the queue and costs ~6 us of idle between the two kernels around it (r03
// timelines: 12 us per step for the one site bench.py times inside the measured steps): sample instead of timing every launch.
int ptx_timing_every(int n)
{
    PTX_REQUIRE(n >= 1, "ptx_timing_every: n=%d", n);
    std::lock_guard<std::mutex> lk(g_timing.mu);
    g_timing.every = n;
    return PTX_OK;
}

int ptx_timing_select_mask(uint64_t mask)
{
    static_assert(KID_COUNT <= 64, "timing mask is 64 bits");
    std::lock_guard<std::mutex> lk(g_timing.mu);
    g_timing.mask = mask & ((KID_COUNT >= 64) ? ~0ull : ((1ull << KID_COUNT) - 1));
    g_timing.used = 0;
    for (unsigned &s : g_timing.seen) s = 0;
    return PTX_OK;
}

static int timing_collect(int *launches, float *total_ms, int nsites)
{
    for (int k = 0; k < nsites; ++k) { launches[k] = 0; total_ms[k] = 0.0f; }
    for (size_t i = 0; i < g_timing.used; ++i) {
        const TimingRec &r = g_timing.pool[i];
        PTX_HIP(hipEventSynchronize(r.b));
        float ms = 0.0f;
        PTX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        const int k = nsites == 1 ? 0 : r.kid;
        if (k >= 0 && k < nsites) { launches[k] += 1; total_ms[k] += ms; }
    }
    g_timing.used = 0;
    return PTX_OK;
}

int ptx_timing_read(int *launches, float *total_ms)
{
    PTX_REQUIRE(launches && total_ms, "ptx_timing_read: null argument");
    std::lock_guard<std::mutex> lk(g_timing.mu);
    return timing_collect(launches, total_ms, 1);
}

int ptx_timing_read_sites(int *launches, float *total_ms, int n)
{
    PTX_REQUIRE(launches && total_ms && n == KID_COUNT, "ptx_timing_read_sites: need arrays of ptx_kernel_count() = %d entries", KID_COUNT);
    std::lock_guard<std::mutex> lk(g_timing.mu);
    return timing_collect(launches, total_ms, n);
}
const char *ptx_last_error(void) { return g_err; }

int ptx_context_create(PtxContext **ctx)
{
    PTX_REQUIRE(ctx != nullptr, "ptx_context_create: null argument");
    PtxContext *c = new PtxContext();
    const int rc = context_init(c);
    if (rc != PTX_OK) { context_release(c); delete c; *ctx = nullptr; return rc; }
    *ctx = c;
    return PTX_OK;
}

int ptx_context_destroy(PtxContext *ctx)
{
    if (ctx == nullptr) return PTX_OK;
    // the caller guarantees that no forward of this context is still being enqueued; work already
    // enqueued on the side streams is drained before they are destroyed
    if (ctx->st) (void)hipStreamSynchronize(ctx->st);
    if (ctx->lo) (void)hipStreamSynchronize(ctx->lo);
    context_release(ctx);
    delete ctx;
    return PTX_OK;
}

int ptx_context_check(PtxContext *ctx)
{
    PTX_REQUIRE(ctx != nullptr, "ptx_context_check: null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return gate_check(ctx);
}

// ptx_context_check behind a drain of the context's streams and of the caller stream of its latest forward: a JOIN (or slot-tag) gate
// that runs out of time does so after the survivor counts have been published, i.e. after the host has the outputs' lengths, so
// ptx_context_check right behind ptx_wait_counts cannot see it yet.  This is the call a host makes when it is about to TRUST the
// outputs of forwards it did not synchronise on (end of a loop, before handing results on, at teardown).
int ptx_context_sync_check(PtxContext *ctx)
{
    PTX_REQUIRE(ctx != nullptr, "ptx_context_sync_check: null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    // the caller's stream of the latest forward must still be alive (proxyt.h); no forward yet: nothing of the caller's to drain
    if (ctx->have_last) PTX_HIP(hipStreamSynchronize(ctx->last_st));
    PTX_HIP(hipStreamSynchronize(ctx->st));
    PTX_HIP(hipStreamSynchronize(ctx->lo));
    return gate_check(ctx);
}

int ptx_context_gates(const PtxContext *ctx) { return ctx != nullptr && ctx->gates_on ? (ctx->lo_ok ? 3 : 1) : 0; }

int ptx_wait_counts(const int32_t *counts_host, int B, int64_t timeout_us)
{
    PTX_REQUIRE(counts_host && B > 0, "ptx_wait_counts: null argument");
    const volatile int32_t *c = counts_host;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        int b = 0;
        while (b < B && c[b] >= 0) ++b;
        if (b == B) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return PTX_OK;
        }
        __builtin_ia32_pause();
        if ((spin & 1023u) == 1023u &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >
                timeout_us)
            return PTX_ETIMEOUT;
    }
}

size_t ptx_prep_bytes(const PtxShape *s)
{
    if (s == nullptr || validate_shape(*s) != PTX_OK) return 0;
    return prep_layout(*s).total * sizeof(float);
}

size_t ptx_workspace_bytes(const PtxShape *s)
{
    if (s == nullptr || validate_shape(*s) != PTX_OK) return 0;
    return ws_layout(*s).total;
}

int ptx_workspace_init(const PtxShape *s, void *workspace, size_t ws_bytes, void *stream)
{
    PTX_TRY(check_bufs(s, workspace, ws_bytes));
    const WsLayout L = ws_layout(*s);
    PTX_HIP(hipMemsetAsync(at<char>(workspace, L.zero_begin), 0, L.zero_bytes, static_cast<hipStream_t>(stream)));
    return PTX_OK;
}

int ptx_prepare(const PtxShape *s, const PtxWeights *w, const float *lin, void *prep, size_t prep_bytes,
                void *stream)
{
    (void)lin;
    PTX_REQUIRE(s && w && prep, "ptx_prepare: null argument");
    PTX_TRY(validate_shape(*s));
    if (prep_bytes < prep_layout(*s).total * sizeof(float)) { set_error("prep buffer too small"); return PTX_ENOSPACE; }
    return run_prepare(*s, *w, static_cast<float *>(prep), static_cast<hipStream_t>(stream));
}

int ptx_grid_centers(const float *points, int B, int N, const float *lin, int gs, float margin,
                     float *minmax, float *centers, void *workspace, size_t ws_bytes, void *stream)
{
    // stage API for PRE:37-48 only: the ball query is skipped by asking for K = 0 hits.
    PTX_REQUIRE(points && lin && minmax && centers && workspace, "ptx_grid_centers: null argument");
    PTX_REQUIRE(ws_bytes >= (size_t)B * 6 * 4, "ptx_grid_centers: workspace needs %d bytes", B * 24);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t *enc = static_cast<uint32_t *>(workspace);
    PTX_HIP(hipMemsetAsync(enc, 0, (size_t)B * 6 * 4, st));
    ScenePts sp;
    PTX_TRY(make_scene_pts(points, nullptr, B, N, &sp));
    PTX_TRY(launch_minmax(sp, B, N, enc, st));
    return launch_ball_query(nullptr, enc, lin, gs, margin, minmax, centers, sp, B, gs * gs * gs, N, 0,
                             0.0f, nullptr, nullptr, nullptr, st);
}

int ptx_ball_query(const float *centers, const float *points, int B, int M, int N, int K, float radius,
                   int32_t *idx, float *cluster, int32_t *pad_count, void *stream)
{
    PTX_REQUIRE(centers && points && idx && cluster, "ptx_ball_query: null argument");
    PTX_REQUIRE(B >= 1 && M >= 1 && N >= 1 && K >= 1, "ptx_ball_query: B=%d M=%d N=%d K=%d", B, M, N, K);
    ScenePts sp;
    PTX_TRY(make_scene_pts(points, nullptr, B, N, &sp));
    return launch_ball_query(centers, nullptr, nullptr, 0, 0.0f, nullptr, nullptr, sp, B, M, N, K, radius,
                             idx, cluster, pad_count, static_cast<hipStream_t>(stream));
}

int ptx_linear(const float *x, const float *w, const float *bias, const float *residual, float *y,
               int rows, int n_out, int n_in, int gelu, void *stream)
{
    PTX_REQUIRE(x && w && y, "ptx_linear: null argument");
    PTX_REQUIRE(rows >= 1 && n_out >= 1 && n_in >= 4, "ptx_linear: rows=%d n_out=%d n_in=%d", rows, n_out, n_in);
    GemmBatch g{}; g.n = 1;
    g.p[0] = GemmProb{x, w, y, bias, residual, nullptr, nullptr, rows, n_out, n_in, n_in, n_in, n_out, n_out, 0, 0,
                      gelu ? EPI_GELU : EPI_NONE};
    return launch_gemm(g, static_cast<hipStream_t>(stream));
}

int ptx_gemm_policy(int min_tiles_128) { return gemm_policy(min_tiles_128); }

int ptx_offset_net(const PtxShape *s, const PtxWeights *w, const void *prep, const float *centers_in,
                   const float *cluster, const float *minmax, float *centers_out, float *offsets_out,
                   void *stream)
{
    PTX_REQUIRE(s && w && prep && centers_in && cluster && minmax && centers_out, "ptx_offset_net: null argument");
    PTX_TRY(validate_shape(*s));
    const PrepLayout P = prep_layout(*s);
    const int M = s->grid_size * s->grid_size * s->grid_size;
    return launch_offset_net(static_cast<const float *>(prep) + P.off_ab, w->offset, w->offset_map_w, centers_in,
                             cluster, minmax, s->B * M, M, s->K, s->margin, centers_out, offsets_out,
                             static_cast<hipStream_t>(stream));
}

int ptx_select_clusters(const PtxShape *s, const int32_t *idx, const float *centers, const float *cluster,
                        const int32_t *pad_count, const int32_t *order_override, int32_t *order,
                        int32_t *picks, int32_t *keep, float *kcenter, float *kcluster, int32_t *kidx,
                        int32_t *drop_idx, uint32_t *tag, void *stream)
{
    PTX_REQUIRE(s && idx && centers && cluster && pad_count && order && picks && keep && kcenter && kcluster &&
                kidx && drop_idx, "ptx_select_clusters: null argument");
    PTX_TRY(validate_shape(*s));
    return launch_select(*s, idx, centers, cluster, pad_count, order_override, order, picks, keep, kcenter,
                         kcluster, kidx, drop_idx, tag, static_cast<hipStream_t>(stream));
}

int ptx_pointnet(const PtxShape *s, const PtxWeights *w, const void *prep, const float *kcenter,
                 const float *kcluster, float *point_proxy, void *stream)
{
    PTX_REQUIRE(s && w && prep && kcenter && kcluster && point_proxy, "ptx_pointnet: null argument");
    PTX_TRY(validate_shape(*s));
    const PrepLayout P = prep_layout(*s);
    return launch_pointnet(static_cast<const float *>(prep) + P.enc_ab, w->encoder, kcenter, kcluster,
                           s->B * s->Mk, s->Mk, s->K, s->C, point_proxy, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, s->ln_eps, nullptr, 0, static_cast<hipStream_t>(stream));
}

int ptx_img_proxy(const PtxShape *s, const PtxWeights *w, const void *prep, const void *img_feat,
                  float *img_proxy, void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(w && prep && img_feat && img_proxy, "ptx_img_proxy: null argument");
    PTX_TRY(check_bufs(s, workspace, ws_bytes));
    return run_img_proxy(*s, *w, static_cast<const float *>(prep), img_feat, img_proxy, workspace,
                         static_cast<hipStream_t>(stream));
}

int ptx_proxy_block(const PtxShape *s, const PtxWeights *w, const void *prep, int which,
                    const float *point_proxy, const float *proxy, int Lp, const uint8_t *mask,
                    float *head_out, float *guide, void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(w && prep && point_proxy && proxy && head_out, "ptx_proxy_block: null argument");
    PTX_REQUIRE(which == 0 || which == 1, "ptx_proxy_block: which=%d", which);
    PTX_TRY(check_bufs(s, workspace, ws_bytes));
    PTX_REQUIRE(Lp >= 1 && Lp <= (s->L > s->V ? s->L : s->V), "ptx_proxy_block: Lp=%d exceeds max(L,V)", Lp);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PrepLayout P = prep_layout(*s);
    const WsLayout L = ws_layout(*s);
    const float *pf = static_cast<const float *>(prep);
    const PtxBlock &blk = which == 0 ? w->text : w->img;
    float *x_in = at<float>(workspace, L.x_in[which]);
    LnBatch lb{}; lb.n = 1; lb.C = s->C; lb.eps = s->ln_eps;     // norm1 + per-slot bias (PRE:274, 215-217)
    lb.p[0] = LnProb{point_proxy, x_in, blk.norm1_w, blk.norm1_b, pf + (which == 0 ? P.posb_t : P.posb_i),
                     s->B * s->Mk, s->Mk};
    PTX_TRY(launch_ln_rows(lb, st));
    Branch br = make_branch(*s, *w, pf, which, x_in, proxy, Lp, mask, head_out, guide);
    return run_blocks(*s, &br, 1, point_proxy, workspace, st);
}

size_t ptx_proxy_attention_scratch_bytes(int B, int n, int Lp, int heads, int C, int impl)
{
    if (B < 1 || n < 1 || Lp < 1 || heads < 1 || C < heads) return 0;
    if (impl == 3) return 256 + (size_t)B * heads * 4 + (size_t)B * heads * kFaMaxSplit * n * kFaPartRow * sizeof(float);
    return (size_t)B * Lp * C * sizeof(float);
}

int ptx_proxy_attention(const float *qkv, const float *pt, const uint8_t *mask, float *out, float *scratch, int B, int n,
                        int Lp, int heads, int C, int impl, void *stream)
{
    PTX_REQUIRE(qkv && pt && out, "ptx_proxy_attention: null argument");
    PTX_REQUIRE(B >= 1 && n >= 1 && Lp >= 1 && heads >= 1 && C >= heads && C % heads == 0, "ptx_proxy_attention: B=%d n=%d Lp=%d heads=%d C=%d", B, n, Lp, heads, C);
    PTX_REQUIRE(impl >= 0 && impl <= 3, "ptx_proxy_attention: impl=%d", impl);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int hd = C / heads;
    FAttnBatch fa{}; fa.nb = 1; fa.B = B; fa.heads = heads; fa.hd = hd; fa.n = n; fa.C = C; fa.scale = attn_scale(hd);
    fa.p[0] = FAttnProb{qkv, pt, mask, out, Lp};
    fa.split = 1;
    const bool can = fused_attn_supported(fa);
    PTX_REQUIRE((impl != 1 && impl != 3) || can, "ptx_proxy_attention: the fused kernel does not support head_dim=%d, Lp=%d", hd, Lp);
    if (impl == 3) {        // the fused kernel with the proxies in four slices: tickets (cleared here) + partials in scratch
        PTX_REQUIRE(scratch != nullptr, "ptx_proxy_attention: impl 3 needs ptx_proxy_attention_scratch_bytes() of scratch");
        const size_t tb = align_up((size_t)B * heads * 4, 256);
        PTX_HIP(hipMemsetAsync(scratch, 0, tb, st));
        fa.split = kFaMaxSplit; fa.tickets = reinterpret_cast<int *>(scratch);
        fa.part = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + tb);
        return launch_proxy_attn(fa, st);
    }
    if (impl == 1 || (impl == 0 && can && (n <= 256 || (long)B * heads >= 96))) return launch_proxy_attn(fa, st);
    PTX_REQUIRE(scratch != nullptr, "ptx_proxy_attention: the two-launch form needs scratch for pv");
    AttnBatch a{}; a.n = 1; a.B = B; a.heads = heads; a.hd = hd; a.scale = attn_scale(hd);
    a.p[0] = AttnProb{pt, qkv + C, qkv + 2 * C, scratch, nullptr, Lp, n, C, 3 * C, 3 * C, C,
                      (long)Lp * C, (long)n * 3 * C, (long)n * 3 * C, (long)Lp * C};
    PTX_TRY(launch_attn32(a, st));
    a.p[0] = AttnProb{qkv, pt, scratch, out, mask, n, Lp, 3 * C, C, C, C, (long)n * 3 * C, (long)Lp * C, (long)Lp * C, (long)n * C};
    return launch_attn32(a, st);
}

int ptx_affine_scatter(const PtxShape *s, const float *points, const uint32_t *tag, const float *kcenter,
                       const float *translate, const float *transform, float *new_points, void *stream)
{
    PTX_REQUIRE(s && points && tag && kcenter && translate && transform && new_points, "ptx_affine_scatter: null argument");
    PTX_TRY(validate_shape(*s));
    ScenePts sp;
    PTX_TRY(make_scene_pts(points, nullptr, s->B, s->N, &sp));
    return launch_affine(*s, sp, const_cast<uint32_t *>(tag), kcenter, translate, transform, new_points, nullptr, nullptr,
                         false, false, static_cast<hipStream_t>(stream));
}

int ptx_affine_compact(const PtxShape *s, const float *points, const uint32_t *tag, const float *kcenter,
                       const float *translate, const float *transform, float *out, int32_t *counts,
                       void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(points && tag && kcenter && translate && transform && out && counts, "ptx_affine_compact: null argument");
    PTX_TRY(check_bufs(s, workspace, ws_bytes));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t *tc = at<int32_t>(workspace, ws_layout(*s).tile_counts);
    PTX_TRY(launch_tile_count(tag, s->B, s->N, tc, counts, nullptr, st));
    ScenePts sp;
    PTX_TRY(make_scene_pts(points, nullptr, s->B, s->N, &sp));
    return launch_affine(*s, sp, const_cast<uint32_t *>(tag), kcenter, translate, transform, out, counts, tc, true, false, st);
}

#define PTX_DBG(field, src, bytes)                                                                         do {                                                                                                       if (debug && debug->field)                                                                                 PTX_HIP(hipMemcpyAsync(debug->field, src, bytes, hipMemcpyDeviceToDevice, st));                } while (0)

int ptx_forward(PtxContext *ctx, const PtxShape *s, const PtxWeights *w, const void *prep, const float *lin,
                const float *points, const float *const *points_list, const float *text_feats,
                const uint8_t *text_mask, const void *img_feat, const int32_t *order_override,
                const float *centers_override,
                float *out, int32_t *counts, void *workspace, size_t ws_bytes, const PtxDebug *debug,
                void *stream)
{
    return ptx_forward_ex(ctx, s, w, prep, lin, points, points_list, text_feats, text_mask, img_feat, order_override,
                          centers_override, out, counts, workspace, ws_bytes, debug, nullptr, stream);
}

int ptx_forward_ex(PtxContext *ctx, const PtxShape *s, const PtxWeights *w, const void *prep, const float *lin,
                   const float *points, const float *const *points_list, const float *text_feats,
                   const uint8_t *text_mask, const void *img_feat, const int32_t *order_override,
                   const float *centers_override,
                   float *out, int32_t *counts, void *workspace, size_t ws_bytes, const PtxDebug *debug,
                   const PtxForwardOpts *opts, void *stream)
{
    PTX_REQUIRE(w && prep && lin && (points || points_list) && text_feats && img_feat && out && counts,
                "ptx_forward: null argument");
    const uint32_t *bbox_in = opts ? opts->bbox_enc : nullptr;
    const int compute_dtype = opts ? opts->compute_dtype : 0;
    PTX_REQUIRE(compute_dtype == 0 || compute_dtype == 1, "ptx_forward: compute_dtype=%d (0 fp32-equivalent, 1 bf16)", compute_dtype);
    PTX_TRY(check_bufs(s, workspace, ws_bytes));
    ScenePts sp;
    PTX_TRY(make_scene_pts(points, points_list, s->B, s->N, &sp));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PtxShape &S = *s;
    const PrepLayout P = prep_layout(S);
    const WsLayout L = ws_layout(S);
    const float *pf = static_cast<const float *>(prep);
    void *ws = workspace;
    const int M = S.grid_size * S.grid_size * S.grid_size, B = S.B, K = S.K, Kd = S.Mt - S.Mk;

    // ---- two streams.  The image branch (PRE:449) only depends on img_feat and is the critical
    // path (two or three streaming passes over 45 / 90 MB per scene): it stays on the CALLER's stream together
    // with everything after the join, so no cross-stream hop sits on the critical path.  The
    // clustering chain (PRE:430-437) runs on the library's high-priority stream `cs` and finishes at about
    // the same time as the image branch (cfg2 shape, bf16 features: both ~185 us).  (Running the image
    // branch as two slices of scenes on two streams, to hide its small table GEMMs behind the other
    // slice's streaming, was measured slower: 7.85k vs 8.1k scenes/s at cfg2, B = 4, at the time.)
    PtxContext *side = ctx;
    if (side == nullptr) PTX_TRY(default_context(&side));
    {
        int dev = 0;
        PTX_HIP(hipGetDevice(&dev));
        PTX_REQUIRE(dev == side->dev, "ptx_forward: context belongs to device %d, current device is %d", side->dev, dev);
    }
    std::lock_guard<std::mutex> enqueue_lock(side->mu);
    // Stream capture (hipGraph, torch.cuda.graph): the forward is capturable as it is -- the side streams fork from and join into
    // the capturing stream through events, which become graph dependencies -- with three adjustments: the device-word gates stay
    // out (the branches of a graph need not run concurrently: a waiting wave could sit in front of the kernel that releases
    // it), nothing that synchronises runs (gate check / probe), and k_select's completion event is a plain record.
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    PTX_HIP(hipStreamIsCapturing(st, &cap_status));
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    if (!capturing) {
        PTX_TRY(gate_check(side));                              // a gate of an EARLIER forward timed out: reported here, once
        side->last_st = st; side->have_last = true;
        if (side->gates_on && (!side->probed || side->probed_st != st)) PTX_TRY(gate_probe(side, st));
    }
    // Which chain stays on the caller's stream (no cross-queue hop on the path the step waits for) depends on the shape.  Estimates in
    // us -- functions of what the chains move and do, not of a named configuration:
    //   image chain      40 + (bytes of img_feat: B V in_dim hw esz) x 0.78 us / MB (16-bit) | 0.98 us / MB (fp32): two / three streaming
    //                    passes at ~5 / ~4 TB/s plus the table GEMMs (cfg2 bf16: 35 us per scene of 196 views; cfg4: 22.5 us per 50 views)
    //   clustering chain 80 + 0.42 us per farthest-point pick (Kd): the picks are sequential and one work-group per scene, so the chain
    //                    does not grow with the batch; k_minmax / k_cluster (~50 us at 100k points) run beside the mean pass
    // The image chain on the caller's stream is the cheaper arrangement by itself (fork and join ride in kernels that exist anyway),
    // so the clustering chain takes the caller's stream only when it is longer BY A MARGIN.  r05 validation, interleaved A/B of the
    // forced layouts (PTX_LAYOUT, profiles/r05_layout_rule_ab.txt): cfg2 bf16 at 1 / 2 / 3 / 4 scenes image chain +2.5 / +1.5 / +9 /
    // +8 % (the r04 rule, est_cluster > est_image, put 1 and 2 scenes on the other side), cfg2 fp32 features at 1 / 2 / 3 scenes +6 / +8 /
    // +6 %; cfg4 (519 picks) at 6 / 8 / 10 / 12 / 16 scenes clustering chain +13 / +10 / +1 / -2 / -11 %; cfg1 at 1 / 4 scenes +10 / +9 %;
    // cfg5 (1 844 picks) at 1 / 8 scenes +2 / +10 %.  The margin of 60 us decides every one of these correctly except cfg4 at 10 scenes (1 %).
    const LayoutChoice lc = choose_layout(S);
    const double est_cluster = lc.est_cluster, est_image = lc.est_image;
    // PTX_LAYOUT (the ONE layout override, for A/B runs and the parity test that drives every arrangement): four characters,
    // '0' / '1' force, anything else leaves the rule -- [0] clustering chain on the caller's stream, [1] early proxies,
    // [2] image chain forked behind k_cluster, [3] slot tags behind a gate on the third stream.  E.g. PTX_LAYOUT=1-0-
    const LayoutForce &lf = lc.lf;
    const bool cluster_on_caller = lc.cluster_on_caller;
    hipStream_t cs = cluster_on_caller ? st : side->st, is = cluster_on_caller ? side->st : st;
    const bool gated = side->gates_on && !cluster_on_caller && !capturing;        // gates instead of events (see k_gate)
    // r04: where the clustering chain is the long one AND the image chain has the slack for it, the image chain forks BEHIND k_cluster
    // instead of at the top, so that the clustering kernel does not share the chip with the mean pass (k_cluster 64 -> ~45 us at
    // cfg4 / 6 scenes: 13.04k -> 13.26k scenes/s, one scene +1 %, 3 scenes +0.8 %; 8 scenes, where the image chain is nearly as
    // long as the sampling, -0.9 %: hence the slack rule; profiles/r04_early_proxies_ab.txt; r05, the same shape in the room regime:
    // 13.9k -> 14.3k, profiles/r05_room_layout_ab.txt).
    const bool img_late = lc.img_late;
    if (!gated && !img_late) {
        PTX_HIP(hipEventRecord(side->fork, st));
        PTX_HIP(hipStreamWaitEvent(side->st, side->fork, 0));
    }
    float *img_proxy = at<float>(ws, L.img_proxy);
    // (Two staggered slices of images on two streams -- slice B streaming its means under slice A's table GEMMs, A
    // pooling under B's tables -- measured again in r02 with the fused / folded chain: 0.319 vs 0.298 ms per step.
    // Twice the launches, and the half-size pooling launches each pay their own partial last round.)
    // test aid (tests/test_gpu_host.py), compiled into libproxyt_hip_testhooks.so only: PTX_GATE_FAULT=fork / join / tags drops the
    // releasing store of that gate, so that the waiter runs into its bound and the failure path (error word, NaN outputs,
    // PTX_EGATE, fall-back to events) can be exercised.  The product library has no such hook.
#ifdef PTX_TEST_HOOKS
    static const char *const fault = getenv("PTX_GATE_FAULT");
#else
    constexpr const char *fault = nullptr;
#endif
    const bool fault_fork = gated && fault && fault[0] == 'f', fault_join = gated && fault && fault[0] == 'j';
    auto launch_gate = [&](int kid, hipStream_t s_, const GateRef &g) -> int {
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s_, g);
        PTX_LAUNCHED("k_gate");
        (void)kid;
        return PTX_OK;
    };
    if (gated) {
        const uint32_t seq = ++side->gate_seq;
        PTX_TRY(run_img_proxy(S, *w, pf, img_feat, img_proxy, ws, is, 1, true, 0, -1, fault_fork ? nullptr : side->gate, seq));
        PTX_TIMED(KID_GATE_FORK, cs, launch_gate(KID_GATE_FORK, cs, gate_ref(side, 0, seq, 1)));
    } else
    if (!img_late) PTX_TRY(run_img_proxy(S, *w, pf, img_feat, img_proxy, ws, is, 1));      // first pass starts at once

    // ---- clustering (PRE:430): bounding boxes, then everything per centre in one launch
    // The words of the workspace's zero region (encoded boxes, tags, count accumulators) are clean on entry and are
    // re-zeroed by their last readers (k_select, k_affine, k_tile_count): no memset launch per call.
    // bounding boxes: computed here (k_minmax into the workspace's clean words, re-zeroed by k_select), or handed over
    // by the ingest (ptx_ingest_gather reduced them while it wrote the points: no pass over the cloud here)
    uint32_t *mm_ws = at<uint32_t>(ws, L.mm_enc), *tag = at<uint32_t>(ws, L.tag);
    const uint32_t *mm_enc = bbox_in ? bbox_in : mm_ws;
    const bool dbg = debug != nullptr;
    float *centers0 = dbg && debug->centers0 ? at<float>(ws, L.centers0) : nullptr;
    float *cluster1 = dbg && debug->cluster1 ? at<float>(ws, L.cluster1) : nullptr;
    float *offsets = dbg && debug->offsets ? at<float>(ws, L.offsets) : nullptr;
    float *centers = at<float>(ws, L.centers), *cluster2 = at<float>(ws, L.cluster2);
    int32_t *idx2 = at<int32_t>(ws, L.idx2), *pad_count = at<int32_t>(ws, L.pad_count);
    if (!bbox_in) PTX_TIMED(KID_MINMAX, cs, launch_minmax(sp, B, S.N, mm_ws, cs));
    // (the `early` decision, needed here for k_cluster's completion event; its description is below)
    const bool early = lc.early;
    // r04: the streams that fork off behind the clusters (the late image chain, the early proxies) wait for k_cluster's own
    // completion signal instead of for one event record each on the caller's stream -- two packets (~5 us each) between k_cluster and
    // k_select on the chain the step waits for
    const bool fork_ext = !capturing && (img_late || early);
    // r04, "cgate": where the clustering chain owns the caller's stream its two cross-stream waits -- for the image chain / the early
    // proxies in front of the attention, for the slot tags in front of k_affine -- are device words as well, each folded into the launch
    // in FRONT of the wait (its first work-group ends with the poll, as at the benchmark shape's join): no barrier packet (~6 us of
    // idle each) on the chain the step waits for.  Needs the probed side-by-side progress of the streams (gates_on, lo_ok); the
    // failure path is the gates' (error word, NaN outputs, PTX_EGATE, events from then on).  (PTX_GATE=0: the events.)
    const bool cgate = cluster_on_caller && side->gates_on && side->lo_ok && !capturing;
    if (cgate) ++side->gate_seq;
    const bool fault_cjoin = cgate && fault && fault[0] == 'j', fault_ctags = cgate && fault && fault[0] == 't';
    // the join's word (52): stored by the third stream behind the early proxies AND the image chain (early), or by the image stream
    // behind its last launch (otherwise); waited for at the END of k_select (early: the gather is next) / of the qkv GEMM (otherwise)
    const bool cg_join_select = cgate && early;
    const bool cg_join_qkv = cgate && !early;
    GateRef cjoin = (cg_join_select || cg_join_qkv) ? gate_ref(side, 52, side->gate_seq, 2) : GateRef{};
    PTX_TIMED(KID_CLUSTER, cs, launch_cluster(S, mm_enc, lin, sp, pf + P.off_ab, w->offset, w->offset_map_w,
                                              centers_override, nullptr, centers0, cluster1, offsets, centers, idx2,
                                              cluster2, pad_count, cs, fork_ext ? side->early_a : nullptr));

    if (img_late) {
        if (!fork_ext) PTX_HIP(hipEventRecord(side->fork, st));
        PTX_HIP(hipStreamWaitEvent(side->st, fork_ext ? side->early_a : side->fork, 0));
        PTX_TRY(run_img_proxy(S, *w, pf, img_feat, img_proxy, ws, is, 1));
    }
    // ---- early proxies (r04).  Where the clustering chain is the long one, most of it is the farthest point sampling: one
    // work-group per scene for hundreds of dependent picks (cfg4: 519 picks, 0.2 ms), after which the point proxies, LayerNorm1 and
    // the qkv projection of the KEPT clusters were still to come (15 + 40 us at 6 scenes).  None of the three needs the selection
    // except for WHICH rows: so they are computed for all Mt clusters that enter the sampling, on the third stream beside it (1.75x the
    // kept rows at the shipped configuration, on a chip that is otherwise waiting), with (LN1(x) + posb_j) W^T + b = LN1(x) W^T +
    // [posb_j W^T + b] and the bracket a parameter-only table (prep.hip); behind the sampling the kept rows are a gather + that table.
    float *point_proxy = at<float>(ws, L.point_proxy);
    float *xin_t = at<float>(ws, L.x_in[0]), *xin_i = at<float>(ws, L.x_in[1]);
    float *translate = at<float>(ws, L.head[0]), *transform = at<float>(ws, L.head[1]);
    float *guide_t = (debug && debug->text_guide) ? at<float>(ws, L.guide[0]) : nullptr;
    float *guide_i = (debug && debug->img_guide) ? at<float>(ws, L.guide[1]) : nullptr;
    Branch br[2] = {make_branch(S, *w, pf, 0, xin_t, text_feats, S.L, text_mask, translate, guide_t),
                    make_branch(S, *w, pf, 1, xin_i, at<float>(ws, L.cbuf), S.V, nullptr, transform, guide_i)};
    br[1].proxy_lnp = at<float>(ws, L.lnp_img);       // norm_img is applied inside the image block's proxy_proj
    // (`early`, decided above: worth it from ~1000 kept rows per call: cfg4 at 6 scenes +5 %, one scene neutral, cfg1 -- 64 kept rows --
    // -6 %) ... and only while the early work (est_all: ~12 B Mt C^2 flop at ~70 TFLOP/s for the two qkv products, the point proxies
    // ~0.4x that) is less than ~3x the sampling (0.42 us per pick): the early rows are 1 / 0.57 of the kept ones, so what does not fit
    // beside the sampling still costs less than the kept-row kernels behind it until then.  Measured at the end of r04
    // (profiles/r04_early_rule_ab.txt): cfg4 at 7 / 8 / 10 / 11 scenes +8.4 / +8.6 / +6.3 / +6.7 % (ratio 0.8 ... 1.25), cfg5 at
    // 3 / 4 / 6 / 8 / 10 / 12 scenes +3.5 / +4.6 / +7 / +6.5 / +3.2 / -2.5 % (ratio 0.9 ... 3.6); cfg5 at 16 scenes with ALL grid clusters
    // as early rows (before k_order): 4 ms beside 0.8 ms of picks, 4.37k -> 3.36k scenes/s.
    if (early) {
        if (!fork_ext) PTX_HIP(hipEventRecord(side->early_a, cs));          // the clusters exist
        PTX_HIP(hipStreamWaitEvent(side->lo, side->early_a, 0));
        // only the Mt clusters that enter the sampling (the least padded ones, PRE:372-385: 70 % of the grid), in the sampling's
        // order: k_order repeats step 1 of k_select on this stream, the rows below are indexed by position in that order
        float *pp_all = at<float>(ws, L.pp_all), *xa_t = at<float>(ws, L.xln_all[0]), *xa_i = at<float>(ws, L.xln_all[1]);
        int32_t *order_e = at<int32_t>(ws, L.order_e);
        PTX_TRY(launch_order(S, pad_count, order_override, order_e, side->lo));
        PTX_TRY(launch_pointnet(pf + P.enc_ab, w->encoder, centers, cluster2, B * S.Mt, S.Mt, K, S.C, pp_all, &w->text, &w->img, nullptr,
                                nullptr, xa_t, xa_i, S.ln_eps, order_e, M, side->lo, nullptr, 0, true));
        GemmBatch g{}; g.n = 2;
        g.p[0] = GemmProb{xa_t, w->text.qkv_w, at<float>(ws, L.g_all[0]), nullptr, nullptr, nullptr, nullptr, B * S.Mt, 3 * S.C, S.C, S.C, S.C,
                          3 * S.C, 0, 0, 0, EPI_NONE};
        g.p[1] = GemmProb{xa_i, w->img.qkv_w, at<float>(ws, L.g_all[1]), nullptr, nullptr, nullptr, nullptr, B * S.Mt, 3 * S.C, S.C, S.C, S.C,
                          3 * S.C, 0, 0, 0, EPI_NONE};
        PTX_TRY(launch_gemm(g, side->lo, compute_dtype));
        PTX_TRY(run_blocks(S, br, 2, point_proxy, ws, side->lo, 1, compute_dtype, nullptr, nullptr, true));    // proxy_proj of the text block
        // (early_b is recorded where the caller's stream is told to wait for it, below)
    }

    // ---- dynamic cluster dropout (PRE:433): ordering + FPS + keep list on the chain; the slot tags and the survivor
    // counts (only k_affine and the host need them) on the low-priority stream next to it
    int32_t *order = at<int32_t>(ws, L.order), *picks = at<int32_t>(ws, L.picks), *keep = at<int32_t>(ws, L.keep);
    float *kcenter = at<float>(ws, L.kcenter);
    float *kcluster = dbg && debug->kcluster ? at<float>(ws, L.kcluster) : nullptr;
    int32_t *kidx = dbg && debug->kidx ? at<int32_t>(ws, L.kidx) : nullptr;
    int32_t *drop_idx = dbg && debug->drop_idx ? at<int32_t>(ws, L.drop_idx) : nullptr;
    int32_t *ksrc = at<int32_t>(ws, L.ksrc);
    const bool ext_event = !capturing;
    // r04 (PTX_LAYOUT[3] forces it): with the gates the tags can leave the chain altogether -- a third (low-priority) stream waits
    // for the first thread of the point-proxy kernel (k_select in front of it has completed), runs k_tags beside the point proxies /
    // qkv GEMM and signals a word that the proj GEMM's work-group 0 waits for in front of k_affine: no event record or wait on
    // either chain.  Measured (profiles/r04_tags_off_chain_ab.txt): 18.44-18.75k against 18.68-19.09k scenes/s at 4 scenes per GPU,
    // neutral at 8 and 32 -- k_tags (16 work-groups of 1024 threads, 128 KB of LDS each) now runs beside the qkv GEMM of the same
    // stream's tail, which takes 22 instead of 13 us: the clustering stream ends where it did.
    // End of r04: on by default where the two chains are about as long as each other (the benchmark shape: est_image < 1.6 est_cluster),
    // i.e. where the 12 us of k_tags at the tail of the clustering stream decide the join: 18.41k -> 18.58k scenes/s at 4 scenes per GPU
    // over six interleaved pairs, +0.3 ... +1.2 % in three sessions; 6 scenes -0.6 %, 2 and 8 neutral (profiles/r04_tags_gated_rule_ab.txt).
    const bool tags_gated = gated && side->lo_ok && (lf.v[3] >= 0 ? lf.v[3] != 0 : est_image < 1.6 * est_cluster);
    const bool tags_tail = !cluster_on_caller && !tags_gated;
    PTX_TIMED(KID_SELECT, cs, launch_select_order(S, centers, pad_count, order_override, order, picks, keep, kcenter,
                                                  ksrc, bbox_in ? nullptr : mm_ws, cs, cluster_on_caller, ext_event && !tags_tail ? side->aux : nullptr,
                                                  cg_join_select ? &cjoin : nullptr));
    int32_t *tile_counts = at<int32_t>(ws, L.tile_counts);
    // The slot tags / survivor counts: only k_affine and the host need them.  When the clustering chain is the long one
    // (it owns the caller's stream) they run on the low-priority stream next to it, after the point proxies.  When the
    // image chain is the long one they go to the END of the clustering stream, behind its last kernel and in front of the
    // join: one cross-queue wait on the caller's stream instead of two (each costs 6-7 us of idle between two dependent
    // kernels wherever it is placed; r03).  (The mirror image -- the tags behind the
    // image chain when the clustering chain owns the caller's stream -- puts them on the critical path at one scene per
    // call, where only the short point-proxy kernels separate k_select from the join: cfg4 at 6 scenes +1 %, cfg1 -6 %,
    // cfg5 -2 %: not done.)
    // (r03: the fork and the join as stream memory operations -- hipStreamWriteValue32 on the producing stream, hipStreamWaitValue32
    // on the consuming one, signal memory from hipExtMallocWithFlags -- instead of event record + wait: parity-green, the join
    // alone 17.63k / 17.53k / 17.06k vs 17.50k / 17.43k / 17.52k scenes/s, fork + join 17.47k / 17.47k / 16.63k vs 17.49k / 17.56k /
    // 17.49k: the ~6 us between the two kernels around a cross-queue wait are not the event's.)
    // (r03, many scenes per call, where the clustering chain has slack: starting it only after the mean pass -- which runs 40 %
    // slower next to it than alone -- moves the cost to the pooling pass: 24.4k vs 25.2k scenes/s at 32 scenes; the clustering
    // stream at low instead of high priority: 25.4k vs 25.6k at 32 scenes, 21.9k vs 21.4k at 12: no shape rule worth having.)
    hipStream_t ts = tags_tail ? cs : side->lo;
    const bool slots_first = !cluster_on_caller && !tags_tail;
    auto enqueue_tags = [&]() -> int {
        if (tags_gated) {
            PTX_TIMED(KID_GATE_TAGS, ts, launch_gate(KID_GATE_TAGS, ts, gate_ref(side, 36, side->gate_seq, 6)));
        } else if (!tags_tail) {
            if (!ext_event) PTX_HIP(hipEventRecord(side->aux, cs));     // else: recorded by k_select's own completion
            PTX_HIP(hipStreamWaitEvent(ts, side->aux, 0));
        }
        // gathered copies of the kept clusters / the drop list: debug outputs only
        if (kcluster || kidx || drop_idx)
            PTX_TRY(launch_select_slots(S, idx2, cluster2, order, picks, keep, kcluster, kidx, drop_idx, nullptr, ts));
        // ownership / drop tags + survivor counts (published early) in one launch, LDS atomics only
        PTX_TIMED(KID_TAGS, ts, launch_tags(S, idx2, order, picks, ksrc, tag, tile_counts, counts, at<int32_t>(ws, L.scene_acc), ts));
        if (tags_gated) {
            auto sig = [&]() -> int {
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, ts, side->gate + 38, side->gate_seq);
                PTX_LAUNCHED("k_signal[tags]");
                return PTX_OK;
            };
            PTX_TIMED(KID_SIGNAL_TAGS, ts, sig());
        } else if (cgate) {
            if (!fault_ctags) {
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, ts, side->gate + 56, side->gate_seq);
                PTX_LAUNCHED("k_signal[ctags]");
            }
        } else if (!tags_tail) PTX_HIP(hipEventRecord(side->tags, ts));
        return PTX_OK;
    };
    if (slots_first) PTX_TRY(enqueue_tags());

    // ---- the rest of the image chain is ENQUEUED here, in front of the clustering stream's later launches (host order only: the two
    // chains are different streams).  Its first GEMM is wanted ~30 us after the mean pass started; behind the point proxies, the qkv
    // GEMM, the tags and the join's signal -- eight launches of ~3 us of host time -- it could arrive late on a slow host (under
    // rocprofv3 the caller's stream sat idle for 7 us between the mean pass and that GEMM: profiles/r04_timeline_tags_gated.txt)
    const bool want_img_proxy = debug != nullptr && debug->img_proxy != nullptr;
    PTX_TRY(run_img_proxy(S, *w, pf, img_feat, img_proxy, ws, is, 2, want_img_proxy));

    // ---- point proxies (PRE:437) + norm1 / slot bias of both blocks; the kept clusters are read through the selection
    bool joined_early = false;
    if (early) {
        const float *const gsrc[2] = {at<float>(ws, L.g_all[0]), at<float>(ws, L.g_all[1])};
        const float *const tb[2] = {pf + P.qkvb[0], pf + P.qkvb[1]};
        float *const qk[2] = {at<float>(ws, L.qkv[0]), at<float>(ws, L.qkv[1])};
        {
            // both cross-stream waits of the caller's chain in ONE place, in front of the gather (each costs ~6 us of idle between
            // the two kernels around it): the image chain finished long before the sampling does.  r04: and as ONE wait -- the third
            // stream (early proxies done) waits for the image chain, the caller's stream for the third stream
            PTX_TRY(run_blocks(S, br, 2, point_proxy, ws, is, 3, compute_dtype));
            PTX_HIP(hipEventRecord(side->join, is));
            PTX_HIP(hipStreamWaitEvent(side->lo, side->join, 0));
            joined_early = true;
        }
        if (cg_join_select) {               // k_select ends with the wait for this word: no packet between it and the gather
            if (!fault_cjoin) {
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, side->lo, side->gate + 52, side->gate_seq);
                PTX_LAUNCHED("k_signal[cjoin]");
            }
        } else {
            PTX_HIP(hipEventRecord(side->early_b, side->lo));
            PTX_HIP(hipStreamWaitEvent(cs, side->early_b, 0));
        }
        // (kept cluster j of a scene = position keep[j] of the sampling's order = row keep[j] of the early tables)
        PTX_TIMED(KID_POINTNET, cs, launch_qkv_gather(at<float>(ws, L.pp_all), gsrc, tb, keep, B, S.Mt, S.Mk, S.C, point_proxy, qk, cs));
    } else
    PTX_TIMED(KID_POINTNET, cs, launch_pointnet(pf + P.enc_ab, w->encoder, kcenter, cluster2, B * S.Mk, S.Mk, K,
                                                S.C, point_proxy, &w->text, &w->img, pf + P.posb_t, pf + P.posb_i, xin_t,
                                                xin_i, S.ln_eps, ksrc, M, cs, tags_gated ? side->gate + 36 : nullptr, side->gate_seq));
    if (!slots_first && !tags_tail) PTX_TRY(enqueue_tags());

    // ---- both proxy blocks + heads in shared launches (PRE:440-455); the input projections that
    // do not need the image proxies still run on the clustering stream
    // (The whole text branch on a third stream while the image chain finishes, leaving only the image
    // branch after the join, was measured slower: 12.6k vs 13.7k scenes/s -- eight more launches, and its
    // small kernels take CUs from the image passes that are on the critical path.)
    if (!early) PTX_TRY(run_blocks(S, br, 2, point_proxy, ws, cs, 1, compute_dtype, cg_join_qkv ? &cjoin : nullptr));
    if (cg_join_qkv && cjoin.flag != nullptr) {
        set_error("ptx_forward: the join gate was not attached to the qkv launch");
        return PTX_ELAUNCH;
    }
    if (tags_tail) PTX_TRY(enqueue_tags());
    const bool pp_early = cluster_on_caller;                                 // (see below, at the join)
    GateRef join{};                     // the join, folded into the first launch behind it (run_blocks)
    if (gated) {                        // the join through the second gate word: signalled behind the clustering stream's last kernel
        auto launch_signal = [&]() -> int {
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, cs, side->gate + 32, side->gate_seq);
            PTX_LAUNCHED("k_signal");
            return PTX_OK;
        };
        if (!fault_join) PTX_TIMED(KID_SIGNAL_JOIN, cs, launch_signal());
        join = gate_ref(side, 32, side->gate_seq, 2);
    } else {
    if (!cluster_on_caller) PTX_HIP(hipEventRecord(side->join, cs));
    // where the clustering chain owns the caller's stream the image block's proxy_proj runs behind the image chain on ITS stream, in
    // front of the join, instead of behind it on the caller's (7 us + a launch gap off the long chain: cfg4 one scene +3 %)
    if (pp_early && !joined_early) PTX_TRY(run_blocks(S, br, 2, point_proxy, ws, is, 3, compute_dtype));
    if (cg_join_qkv) {                      // the qkv GEMM on the caller's stream ends with the wait for this word
        if (!fault_cjoin) {
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, is, side->gate + 52, side->gate_seq);
            PTX_LAUNCHED("k_signal[cjoin]");
        }
    } else {
    if (cluster_on_caller && !joined_early) PTX_HIP(hipEventRecord(side->join, is));
    if (!joined_early) PTX_HIP(hipStreamWaitEvent(st, side->join, 0));
    }
    }
    GateRef tags_join{};
    if (tags_gated) tags_join = gate_ref(side, 38, side->gate_seq, 7);
    if (cgate) tags_join = gate_ref(side, 56, side->gate_seq, 7);      // the proj GEMM ends with the wait for the tags' word
    PTX_TRY(run_blocks(S, br, 2, point_proxy, ws, st, pp_early ? 4 : 2, compute_dtype, &join, &tags_join));
    if (join.flag != nullptr) {         // (not attached: no launch in front of the attention took it)
        set_error("ptx_forward: the join gate was not attached to a launch");
        return PTX_ELAUNCH;
    }

    // ---- submanifold reshape + scatter + drop (PRE:459-467); k_affine is the last reader of the tags and clears them
    // (r03: waiting for the tags next to the join instead -- they are final long before it at the benchmark shape -- does not
    //  shorten the heads -> affine boundary: 0.276 / 0.283 vs 0.271 / 0.275 ms per step)
    if (cgate && tags_join.flag != nullptr) {
        set_error("ptx_forward: the tags gate was not attached to the proj launch");
        return PTX_ELAUNCH;
    }
    if (!tags_tail && !tags_gated && !cgate) PTX_HIP(hipStreamWaitEvent(st, side->tags, 0));
    PTX_DBG(tag, tag, (size_t)B * S.N * 4);
    PTX_TIMED(KID_AFFINE, st, launch_affine(S, sp, tag, kcenter, translate, transform, out, counts, tile_counts,
                                            true, true, st, (gated || cgate) ? side->gate + 48 : nullptr));

    PTX_DBG(centers0, centers0, (size_t)B * M * 3 * 4);
    PTX_DBG(cluster1, cluster1, (size_t)B * M * K * 3 * 4);
    PTX_DBG(offsets, offsets, (size_t)B * M * 3 * 4);
    PTX_DBG(centers, centers, (size_t)B * M * 3 * 4);
    PTX_DBG(cluster2, cluster2, (size_t)B * M * K * 3 * 4);
    PTX_DBG(idx2, idx2, (size_t)B * M * K * 4);
    PTX_DBG(pad_count, pad_count, (size_t)B * M * 4);
    PTX_DBG(order, order, (size_t)B * S.Mt * 4);
    PTX_DBG(picks, picks, (size_t)B * Kd * 4);
    PTX_DBG(keep, keep, (size_t)B * S.Mk * 4);
    PTX_DBG(kidx, kidx, (size_t)B * S.Mk * K * 4);
    PTX_DBG(drop_idx, drop_idx, (size_t)B * Kd * K * 4);
    PTX_DBG(kcenter, kcenter, (size_t)B * S.Mk * 3 * 4);
    PTX_DBG(kcluster, kcluster, (size_t)B * S.Mk * K * 3 * 4);
    PTX_DBG(point_proxy, point_proxy, (size_t)B * S.Mk * S.C * 4);
    PTX_DBG(img_proxy, img_proxy, (size_t)B * S.V * S.C * 4);
    PTX_DBG(text_guide, guide_t, (size_t)B * S.Mk * S.C * 4);
    PTX_DBG(img_guide, guide_i, (size_t)B * S.Mk * S.C * 4);
    PTX_DBG(translate, translate, (size_t)B * S.Mk * 3 * 4);
    PTX_DBG(transform, transform, (size_t)B * S.Mk * 9 * 4);
    return PTX_OK;
}

}  // extern "C"
