// C-ABI of libcomorag_hip.so (include/comorag_hip.h): index lifetime, append, search, scores,
// re-score, shard merge, encoder tail, profiling.  Host-side C++; every numeric step is a HIP
// kernel from scan_kernels.hip / aux_kernels.hip.  There is no CPU fallback in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/comorag_hip.h"
#include "cmr_kernels.h"
#include "cmr_internal.h"

#define CMR_DT_F32 0
#define CMR_PANEL_ROWS 32
#define CMR_SCAN_WAVES 8
#define CMR_CORPUS_SLACK (128 * 1024)

bool cmr_ring_audit_ok(int dtype, int nqt, int cap, int ring, int mode);  // ring_audit.cpp (generated at build); mode: 0 top-k, 1 scores, 2 top-k with the finishing stage
static const long long kMaxMergeLists = 4096;                    // merge_query_kernel: W <= 16 * MERGE_THREADS

namespace {
thread_local std::string g_err;
}

// sets the thread's error message; also used by comm.hip / multi.hip
int cmr_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define fail cmr_fail

namespace {

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            const int code_ = (e_ == hipErrorOutOfMemory) ? CMR_ERR_OOM                           \
                              : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? CMR_ERR_NO_DEVICE \
                                                                                        : CMR_ERR_HIP; \
            return fail(code_, "%s failed: %s", #expr, hipGetErrorString(e_));                    \
        }                                                                                         \
    } while (0)

int elem_size(int dtype) { return dtype == CMR_F32 ? 4 : 2; }
int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t need) {
        if (need <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = std::max(need, cap * 2);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Scratch of one in-flight search.  One per stream (searches on a stream are serialised by it).
struct Workspace {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf qfrag, lists, cnt, mm, flag, tau, s_lists, s_cnt, s_mm, arrive;
    DevBuf fin_ctl, fin_pmax, fin_tau, fin_dense, fin_mm;
    bool fin_ctl_armed = false;      // scan with the finishing stage (cmr_launch_scan_fin)
    // host-API staging
    DevBuf d_q, d_ids, d_scores, d_min, d_max, d_cand, d_out;
    // synchronous search: queries in, (ids | scores | min | max | non-finite flag) out through ONE pinned host buffer and
    // one copy each way — five pageable D2H copies cost more than the search of a small corpus
    DevBuf d_pack;
    int* flag_ptr = nullptr;     // the non-finite-query flag the kernels set: flag.p, or the head of d_pack for the host API
    // Synchronous host API with mapped results: the word (device view of the pinned buffer, bytes 4..7) that the search's LAST kernel sets
    // once its results are written — 1: final, 2: the finishing stage overflowed and the merge recorded in `lazy` is still due.  Offered by
    // cmr_index_search_begin; a route that can honour it (single-launch search, scan with the finishing stage) sets done_used, and
    // cmr_index_search_finish then polls the word instead of waiting for the stream (5.5 us per call: tools/probe/poll_probe.hip).
    int* done_ptr = nullptr;
    bool done_used = false;
    struct LazyMerge {
        bool due = false;
        const u64* lists = nullptr; const int* cnt = nullptr; int W = 0, NQ = 0, cap = 0, nqp = 0, k = 0; const float2* mm = nullptr; long long id_base = 0;
        int64_t* ids = nullptr; float* scores = nullptr; float* mn = nullptr; float* mx = nullptr; const int* state = nullptr;
    } lazy;
    void* h_pin = nullptr;
    void* h_pin_dev = nullptr;   // the same buffer as the device sees it (mapped, fine-grained)
    size_t h_pin_cap = 0;
    hipError_t ensure_pin(size_t need) {
        if (need <= h_pin_cap) return hipSuccess;
        if (h_pin) { hipError_t e = hipHostFree(h_pin); if (e != hipSuccess) return e; h_pin = nullptr; h_pin_dev = nullptr; h_pin_cap = 0; }
        const size_t want = std::max(need, h_pin_cap * 2);
        hipError_t e = hipHostMalloc(&h_pin, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(&h_pin_dev, h_pin, 0);
        if (e != hipSuccess) { (void)hipHostFree(h_pin); h_pin = nullptr; return e; }
        h_pin_cap = want;
        return hipSuccess;
    }
    void release() {
        qfrag.release(); lists.release(); cnt.release(); mm.release(); flag.release(); tau.release();
        s_lists.release(); s_cnt.release(); s_mm.release(); arrive.release();
        fin_ctl.release(); fin_pmax.release(); fin_tau.release(); fin_dense.release(); fin_mm.release();
        d_q.release(); d_ids.release(); d_scores.release(); d_min.release(); d_max.release(); d_cand.release(); d_out.release();
        d_pack.release();
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr; h_pin_dev = nullptr; h_pin_cap = 0;
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

struct ProfEvent { hipEvent_t a, b; };

constexpr size_t kMappedAppendMax = 128 * 1024;  // appends up to this many bytes of fp32 rows are read by the convert kernel from mapped host memory (BASELINE config 4 appends 25 rows x 768 = 75 KiB per cycle)
constexpr size_t kZeroCopyMax = 256 * 1024;   // synchronous host API: queries / results up to this size are mapped, not copied

struct PipeSlot { Workspace ws; hipEvent_t pre_done = nullptr, scan_done = nullptr, main_done = nullptr; bool used = false; };
#define CMR_PIPE_SLOTS 4
// sp / sm / sm2: pre-phase and main scans of batches of <= 64 queries (with CU masks: sm, sm2 on n_cu - 64 CUs, sp on the other
// 64); wp / wm: the same for wide batches (no masks: the wide kernel is matrix-pipe-bound and wants every CU); sq: candidate
// merges and whatever the caller appends behind a batch (no mask: its small workgroups fit beside a scan workgroup on any CU)
// usp / usm / uwp / uwm: unmasked twins of sp / sm / wp / wm, created with them — scans of a millisecond and longer run there
// (pipe_cu_mask = -1): the masks pay where ramp, tail and packet gaps are a visible share of a step, and cost a long scan CUs.
struct Pipe { hipStream_t sp = nullptr, sm = nullptr, sm2 = nullptr, wp = nullptr, wm = nullptr, wm2 = nullptr, sq = nullptr, usp = nullptr, usm = nullptr, uwp = nullptr, uwm = nullptr; PipeSlot slot[CMR_PIPE_SLOTS]; unsigned next = 0; int nslots = 2; unsigned nscan = 0, nwscan = 0; int scan_cus = 0, wide_cus = 0; int last_masked = 0; };

}  // namespace

struct cmr_index {
    int device = 0;
    int dim = 0, dpad = 0, dtype = 0;
    uint32_t flags = 0;
    int n_cu = 256;
    long long n = 0;             // rows
    long long cap_panels = 0;    // allocated panels
    void* corpus = nullptr;      // panel-major blocks (+ slack)
    float* shadow = nullptr;     // optional fp32 row-major [cap_rows, dim]
    std::shared_mutex mu;        // searches shared, append/destroy exclusive
    std::mutex ws_mu;
    std::vector<Workspace*> free_ws;            // for the synchronous host API
    std::map<hipStream_t, Workspace*> stream_ws;  // for the _dev API
    DevBuf stage;                // append staging
    void* h_pin = nullptr;       // small appends: pinned, device-mapped rows + flag (exclusive lock held)
    void* h_pin_dev = nullptr;
    size_t h_pin_cap = 0;
    int* d_flag = nullptr;       // non-finite flag for appends
    // profiling
    std::mutex prof_mu;
    bool prof_on = false;
    int prof_every = 1;          // time every prof_every-th main scan (two event packets on the scan stream cost ~25 us between scans)
    unsigned prof_seq = 0;
    std::vector<ProfEvent> prof_events;
    double prof_bytes = 0.0;
    // route selectors (cmr_index_set_option; results never depend on them)
    int force_ring = 0;      // scan_ring = 8 | 16
    int force_asm = -1;      // scan_asm_ring = 0 | 1
    int force_grid = 0;      // scan_grid
    int no_sample = 0;       // scan_no_sample = 1 disables the sampling pass
    int no_wide = 0;         // scan_no_wide = 1 disables the wide-batch (register-resident query) kernel
    int no_tiny = 0;         // scan_no_tiny = 1 disables the single-launch paths (search and all-scores) altogether
    long long single_level_max = 320000;   // sample_single_max: ONE sampling level while queries x panels stays at or below this
    int single_level = 1;    // sample_single = 0: small batches on mid-size corpora sample in two levels like everything else
    int tiny_multi = 1;      // tiny_multi = 0: the single-launch path always runs as one workgroup (<= 1024 rows only)
    int small_max_panels = 6144;   // small_max_panels: largest corpus (in 32-row panels) the single-launch path takes
    int no_small = 0;        // scan_no_small = 1: corpora of 1025 rows .. 64 K rows take the general path also for few queries
    int zero_copy = 1;       // zero_copy = 0: the synchronous host API copies queries / results instead of mapping them
    int wide_waves = 0;      // wide_waves = 4 | 8: waves per workgroup of the wide kernel at 768-d (0 = the measured default)
    int dual_scan = -1;      // pipe_dual_scan: -1 (default: with the masks, for scans shorter than 1 ms) | 1 (always) | 0 (never): main scans alternate between two streams, so the next scan's workgroups take over the CUs this
                             // scan's workgroups leave (no idle gap between two scans); needs pipe_cu_mask, else the next scan would simply
                             // occupy the CUs left free for the pre-phase
    int cu_mask = -1;        // pipe_cu_mask = -1 (default on a 256-CU device: scans shorter than 1 ms) | 1 | 2 (every scan) | 0 (off): scan stream(s) with a CU mask of n_cu - 64 CUs, the pre-phase streams
                             // with the other 64 (1: mask bits interleave the XCDs — the amdgpu driver's enumeration; 2: 32 consecutive bits per XCD)
    int wide_abl = 0;        // development builds only
    int stream_nt = -1;      // stream_nt: -1 default (non-temporal corpus loads, default policy for the query-split grid) | 0 | 1: force
    int wide_mode = 0;       // wide_mode: batches of more than one narrow pass — 1: the register-resident wide kernel, 2: the query-split grid of the
                             // narrow kernel (up to 4 query tiles walk the same panel ranges on CUs of one XCD; any dim / dtype), 0: the measured default
    int tau_in_scan = 1;     // sample_tau_in_scan = 0: the single sampling level of a small batch is merged by a launch of its own again
    int sync_poll = 1;       // sync_poll = 0: the synchronous host API waits for the stream instead of polling the done word of its mapped result buffer
    int scan_fin = 1;        // scan_fin = 0: small synchronous batches on corpora beyond the single-launch path run the sampling / scan / merge chain
                             // instead of the scan with the finishing stage (thresholds and final selection inside the scan launch)
    int fin_dense = 16384;   // scan_fin_dense: keys per query of the finishing stage's dense candidate lists (~k x panels / 1024 beat a threshold taken
                             // from 1024 first panels: 600 at 1 M rows, 6 K at 10 M; a list that overflows hands the selection to the merge launch)
    int fin_cap = 0;         // scan_fin_cap: keys per (wave, query) list of the scan with the finishing stage (128 | 256; 0: the geometry's, by k).  256 measured: 1 M rows, 8 queries 304-309 us against 314-317, everything else level (2 M rows 548-552 against 537-548)
    int fin_suppliers = 0;   // scan_fin_suppliers: workgroups whose first panels make the threshold sample (0: 64, 128 from 4 M rows up; <= 128)
    int fin_spin = 0;        // scan_fin_spin: rounds of ~1.5 us the workgroups that do not supply thresholds wait for them before they scan without (0: they look once)
    int fin_max_q = 8;       // scan_fin_queries: largest batch the finishing stage takes (<= 16).  Measured at 768-d bf16, per call, stage / chain:
                             // 1 M rows — 1 / 2 / 4 / 8 / 16 queries 287 / 292 / 297 / 322 / 392 us against 300 / 313 / 333 / 336 / 372;
                             // 2 M rows — 520 / 523 / 525 / 538 / 589 against 563 / 560 / 577 / 588 / 615
    int dual_wide_active = 0;   // read-only ("pipe_dual_scan_wide_active"): the same for the last wide pass
    int dual_active = 0;     // read-only ("pipe_dual_scan_active"): did the last pipelined <= 64-query pass alternate between the two scan streams
    long long id_base = 0;   // added to every returned row id (global ids of a row shard)
    // A row shard that took incremental appends holds several runs of consecutive global ids (cmr_index_set_id_blocks): the
    // kernels then run with base 0 and a remap launch translates their ids; candidate / row ids coming IN are translated
    // on the host.  One block = plain id_base.
    std::vector<long long> blk_local, blk_global;
    long long* d_blk = nullptr;          // [local0[nb] | global0[nb]] on the device
    std::vector<void*> blk_retired;      // earlier tables: in-flight searches may still read them (a few bytes each, freed at destroy)
    int sample_maxmul = 0;   // sample_maxmul: level-1 sample <= sample_maxmul x level 0 (0 = 128 narrow / 512 wide)
    int sample_div = 32;     // sample_div: level-1 sample = 1/sample_div of the panels (clamped to [8, 128] x level 0)
    int pipe_slots = 3;      // pipe_slots (2..4): batches in the pipeline.  A third slot lets the pre-phase of batch i+2 start before
                             // scan i has ended: 1 M x 768 bf16, B = 64 step 0.279 -> 0.264 ms; nothing at 10 M rows
    int reserve_cus = -1;    // pipe_reserve_cus: CUs the pipelined main scan leaves free (-1 = by corpus size, see enqueue_pass)
    std::mutex pipe_mu;
    Pipe pipe;
    size_t panel_bytes() const { return (size_t)CMR_PANEL_ROWS * dpad * elem_size(dtype); }
};

namespace {

// Route selectors of an index (cmr_index_set_option).  Every one of them picks between implementations that return the
// SAME results; they exist so that tests can hold the routes against each other and tools can A/B a kernel decision.
int set_option(cmr_index* idx, const char* name, long long v) {
    const std::string n(name ? name : "");
    if (n == "scan_ring") idx->force_ring = (int)v;
    else if (n == "scan_asm_ring") idx->force_asm = (int)v;
    else if (n == "scan_grid") idx->force_grid = (int)v;
    else if (n == "scan_no_sample") idx->no_sample = (int)v;
    else if (n == "scan_no_wide") idx->no_wide = (int)v;
    else if (n == "scan_no_tiny") idx->no_tiny = (int)v;
    else if (n == "scan_no_small") idx->no_small = (int)v;
    else if (n == "small_max_panels") idx->small_max_panels = (int)std::max<long long>(32, std::min<long long>(v, 6144));
    else if (n == "tiny_multi") idx->tiny_multi = (int)v;
    else if (n == "zero_copy") idx->zero_copy = (int)v;
    else if (n == "sample_single") idx->single_level = (int)v;
    else if (n == "sample_single_max") idx->single_level_max = std::max<long long>(0, v);
    else if (n == "sample_tau_in_scan") idx->tau_in_scan = (int)v;
    else if (n == "scan_fin") idx->scan_fin = (int)v;
    else if (n == "sync_poll") idx->sync_poll = (int)v;
    else if (n == "scan_fin_dense") idx->fin_dense = (int)std::max<long long>(1, std::min<long long>(v, 1 << 16));
    else if (n == "scan_fin_cap") { if (v != 0 && v != 128 && v != 256) return fail(CMR_ERR_INVALID, "scan_fin_cap must be 0, 128 or 256"); idx->fin_cap = (int)v; }
    else if (n == "scan_fin_suppliers") idx->fin_suppliers = (int)std::max<long long>(0, std::min<long long>(v, CMR_FIN_SLOTS / CMR_SCAN_WAVES));
    else if (n == "scan_fin_spin") idx->fin_spin = (int)std::max<long long>(0, std::min<long long>(v, 1000));
    else if (n == "scan_fin_queries") idx->fin_max_q = (int)std::max<long long>(1, std::min<long long>(v, CMR_FIN_MAX_QUERIES));
    else if (n == "sample_div") idx->sample_div = (int)std::max<long long>(2, v);
    else if (n == "sample_maxmul") idx->sample_maxmul = (int)std::max<long long>(0, v);
    else if (n == "pipe_reserve_cus") idx->reserve_cus = (int)v;
    else if (n == "pipe_slots") idx->pipe_slots = (int)v;
    else if (n == "wide_waves") { if (v != 0 && v != 4 && v != 8) return fail(CMR_ERR_INVALID, "wide_waves must be 0 (default), 4 or 8"); idx->wide_waves = (int)v; }
    else if (n == "wide_mode") { if (v < 0 || v > 2) return fail(CMR_ERR_INVALID, "wide_mode must be 0 (default), 1 (register-resident kernel) or 2 (query-split grid)"); idx->wide_mode = (int)v; }
    else if (n == "stream_nt") idx->stream_nt = (int)v;
    else if (n == "pipe_dual_scan") idx->dual_scan = (int)v;
    else if (n == "pipe_cu_mask") idx->cu_mask = (int)v;
#ifdef CMR_DEV_KNOBS
    else if (n == "wide_abl") idx->wide_abl = (int)v;      // ablation kernels: results are WRONG by design (development builds only)
#endif
    else return fail(CMR_ERR_INVALID, "unknown option '%s'", n.c_str());
    return CMR_OK;
}

#ifdef CMR_DEV_KNOBS
// development builds (-DCMR_DEV_KNOBS, tools/): the same options from the environment, CMR_<OPTION NAME IN CAPITALS>
void options_from_env(cmr_index* idx) {
    static const char* names[] = {"scan_ring", "scan_asm_ring", "scan_grid", "scan_no_sample", "scan_no_wide", "scan_no_tiny", "scan_no_small",
                                  "small_max_panels", "tiny_multi", "zero_copy", "sample_single", "sample_single_max", "sample_tau_in_scan", "scan_fin", "sync_poll", "scan_fin_queries", "scan_fin_dense", "scan_fin_spin", "scan_fin_suppliers", "scan_fin_cap", "sample_div", "sample_maxmul", "pipe_reserve_cus",
                                  "pipe_slots", "wide_waves", "wide_mode", "stream_nt", "pipe_dual_scan", "pipe_cu_mask", "wide_abl"};
    for (const char* nm : names) {
        std::string env = "CMR_";
        for (const char* c = nm; *c; ++c) env += (char)toupper(*c);
        const char* v = getenv(env.c_str());
        if (v && *v) (void)set_option(idx, nm, atoll(v));
    }
}
#endif

// id base the kernels add themselves (0 when a block table translates afterwards)
long long kernel_id_base(const cmr_index* idx) { return idx->blk_local.size() > 1 ? 0 : idx->id_base; }
// shard-local ids -> global ids, on the stream that produced them (no-op for a single block)
int remap_ids_enqueue(cmr_index* idx, int64_t* ids_dev, long long n, hipStream_t s);
// global id -> shard-local row, -1 when this shard does not hold it
long long to_local_row(const cmr_index* idx, long long gid) {
    if (gid < 0) return -1;
    if (idx->blk_local.size() <= 1) { const long long r = gid - idx->id_base; return (r >= 0 && r < idx->n) ? r : -1; }
    const size_t nb = idx->blk_local.size();
    size_t b = std::upper_bound(idx->blk_global.begin(), idx->blk_global.end(), gid) - idx->blk_global.begin();
    if (b == 0) return -1;
    --b;
    const long long len = (b + 1 < nb ? idx->blk_local[b + 1] : idx->n) - idx->blk_local[b];
    const long long off = gid - idx->blk_global[b];
    return off < len ? idx->blk_local[b] + off : -1;
}

// would appending n rows push the last block's global ids past what the packed exchange can carry?
bool global_id_overflow(const cmr_index* idx, long long n) {
    const long long l0 = idx->blk_local.size() > 1 ? idx->blk_local.back() : 0;
    const long long g0 = idx->blk_local.size() > 1 ? idx->blk_global.back() : idx->id_base;
    return g0 + (idx->n + n - l0) - 1 > 0xFFFFFFFEll;
}

int set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return CMR_OK;
}

int check_device(int device_id) {
    static std::mutex mu;
    static std::vector<char> ok;                  // devices already validated (hipGetDeviceProperties is slow)
    {
        std::lock_guard<std::mutex> g(mu);
        if (device_id >= 0 && (size_t)device_id < ok.size() && ok[device_id]) return CMR_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CMR_ERR_NO_DEVICE, "no HIP device visible (%s); libcomorag_hip has no CPU fallback",
                                               e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(CMR_ERR_NO_DEVICE, "device_id %d out of range [0,%d)", device_id, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CMR_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    {
        std::lock_guard<std::mutex> g(mu);
        if (ok.size() <= (size_t)device_id) ok.resize((size_t)device_id + 1, 0);
        ok[device_id] = 1;
    }
    return CMR_OK;
}

// Synchronous host API, results in the workspace's pinned, device-mapped buffer: the search's last kernel stores a word (bytes 4..7 of the
// buffer, zeroed by the caller before the launch) behind its results — both by system-scope stores, the device's writes arrive in order —
// and the host polls it: seen ~5.5 us before hipStreamSynchronize returns (tools/probe/poll_probe.hip).  Bounded: a launch that never
// reports (a fault) is left to the stream, whose error comes back.
int wait_done_word(Workspace* ws, int* state = nullptr) {
    volatile int* const w = (volatile int*)((char*)ws->h_pin + 4);
    int st = *w;
    if (!st) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 1; !(st = *w); ++spin) {
            __builtin_ia32_pause();
            if ((spin & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!st) { HIP_TRY(hipStreamSynchronize(ws->stream)); st = *w; }
    if (state) *state = st;
    return CMR_OK;
}

// the workspace's non-finite flag, allocated and zeroed on first use
int arm_flag(Workspace* ws, hipStream_t s) {
    if (ws->flag_ptr) return CMR_OK;
    HIP_TRY(ws->flag.ensure(sizeof(int)));
    HIP_TRY(hipMemsetAsync(ws->flag.p, 0, sizeof(int), s));
    ws->flag_ptr = (int*)ws->flag.p;
    return CMR_OK;
}

int remap_ids_enqueue(cmr_index* idx, int64_t* ids_dev, long long n, hipStream_t s) {
    if (idx->blk_local.size() <= 1) return CMR_OK;
    HIP_TRY(cmr_launch_remap_ids(ids_dev, n, idx->d_blk, (int)idx->blk_local.size(), s));
    return CMR_OK;
}

Workspace* acquire_ws(cmr_index* idx, hipStream_t user_stream, bool dev_api) {
    std::lock_guard<std::mutex> g(idx->ws_mu);
    if (dev_api && user_stream) {
        auto it = idx->stream_ws.find(user_stream);
        if (it != idx->stream_ws.end()) return it->second;
        Workspace* w = new Workspace();
        w->stream = user_stream;
        idx->stream_ws[user_stream] = w;
        return w;
    }
    if (dev_api) {  // NULL on the dev API is the legacy default stream itself (what torch's default stream is): work enqueued
                    // there is ordered with the caller's kernels on that stream, exactly as on any other stream handle
        auto it = idx->stream_ws.find(nullptr);
        if (it != idx->stream_ws.end()) return it->second;
        Workspace* w = new Workspace();
        w->stream = nullptr;
        idx->stream_ws[nullptr] = w;
        return w;
    }
    if (!idx->free_ws.empty()) { Workspace* w = idx->free_ws.back(); idx->free_ws.pop_back(); return w; }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
    w->own_stream = true;
    return w;
}
void release_ws(cmr_index* idx, Workspace* w) {
    std::lock_guard<std::mutex> g(idx->ws_mu);
    idx->free_ws.push_back(w);
}

// Scan geometry for a pass of `nq` queries with top-`k`.
int make_geom(cmr_index* idx, int nq, int k, bool topk, CmrScanGeom* g) {
    const int max_nqt = cmr_scan_max_nqt(idx->dtype, idx->dpad);
    if (max_nqt == 0) return fail(CMR_ERR_UNSUPPORTED, "dim %d too large for the LDS-resident query tile (dtype %d)", idx->dim, idx->dtype);
    g->dtype = idx->dtype;
    g->dpad = idx->dpad;
    g->nqt = (nq > 32 && max_nqt >= 2) ? 2 : 1;
    g->cap = (topk && k > 32) ? 256 : 128;
    const int ks = idx->dtype == CMR_F32 ? idx->dpad / 8 : idx->dpad / 16;
    g->ring = (ks % 16 == 0) ? 16 : 8;
    if (idx->force_ring == 8 || (idx->force_ring == 16 && ks % 16 == 0)) g->ring = idx->force_ring;
    const int mode = topk ? 0 : 1;
    g->asm_ring = cmr_ring_audit_ok(g->dtype, g->nqt, g->cap, g->ring, mode) ? 1 : 0;
    if (idx->force_asm == 0) g->asm_ring = 0;
    if (idx->force_asm == 1 && !cmr_ring_audit_ok(g->dtype, g->nqt, g->cap, g->ring, mode))
        return fail(CMR_ERR_UNSUPPORTED, "CMR_SCAN_ASM_RING=1 but variant (dtype %d nqt %d cap %d ring %d) failed the ISA audit", g->dtype, g->nqt, g->cap, g->ring);
    if (!cmr_scan_geom(g)) return fail(CMR_ERR_UNSUPPORTED, "scan geometry does not fit LDS (dpad %d nqt %d)", idx->dpad, g->nqt);
    const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    const int bpc = g->lds <= 80 * 1024 ? 2 : 1;
    long long grid = (npanels + CMR_SCAN_WAVES - 1) / CMR_SCAN_WAVES;  // >= 1 panel per wave
    grid = std::min<long long>(grid, (long long)idx->n_cu * bpc);
    if (idx->force_grid > 0) grid = std::min<long long>(grid, idx->force_grid);
    g->grid = (int)std::max<long long>(grid, 1);
    return CMR_OK;
}

double algorithmic_bytes(const cmr_index* idx, int nq, int k) {
    const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    return (double)npanels * CMR_PANEL_ROWS * idx->dpad * elem_size(idx->dtype) + (double)nq * idx->dim * 4 + (double)nq * k * 12;
}

// Enqueue a full search (all passes) on ws->stream.  Device pointers in, device pointers out.
int scores_enqueue(cmr_index* idx, Workspace* ws, const float* q_dev, int nq, float* out_dev, long long ld);

// k above CMR_MAX_K (retrieve_knn's synonymy_edge_topk = 2047): materialise the scores of a block of
// queries on the device, then select per row (radix select + ordered compaction + bitonic sort).
int search_large_k_enqueue(cmr_index* idx, Workspace* ws, const float* q_dev, int nq, int k, int64_t* ids_dev,
                           float* scores_dev, float* min_dev, float* max_dev) {
    hipStream_t s = ws->stream;
    const long long n = idx->n;
    const long long ld = (n + 3) / 4 * 4;
    const int blockq = (int)std::max<long long>(1, std::min<long long>(nq, (1ll << 31) / std::max<long long>(ld * 4, 1)));  // <= 2 GiB
    HIP_TRY(ws->d_out.ensure((size_t)blockq * ld * 4));
    for (int q0 = 0; q0 < nq; q0 += blockq) {
        const int nb = std::min(blockq, nq - q0);
        int rc = scores_enqueue(idx, ws, q_dev + (size_t)q0 * idx->dim, nb, (float*)ws->d_out.p, ld);
        if (rc) return rc;
        HIP_TRY(cmr_launch_topk_rows((const float*)ws->d_out.p, ld, (int)n, nb, k, kernel_id_base(idx), ids_dev + (size_t)q0 * k,
                                     scores_dev + (size_t)q0 * k, min_dev ? min_dev + q0 : nullptr,
                                     max_dev ? max_dev + q0 : nullptr, s));
    }
    return remap_ids_enqueue(idx, ids_dev, (long long)nq * k, s);
}

// Every wave (workgroup, for the wide kernel) scans a contiguous range of floor/ceil(npanels / W) panels and
// the kernel ends with the longest range: at 1 M rows on 256 CUs that is 16 panels against a mean of 15.3,
// a 5 % tail during which HBM idles.  The scan is bandwidth-bound, not CU-bound, so giving up a few workgroups
// (<= 1/8) for the W that minimises the padded panel count ceil(npanels / W) * W is free.  (Not for the wide
// kernel: it is MFMA-bound and wants every CU.)
int balanced_grid(long long npanels, int grid, int lists_per_wg) {
    // a dropped workgroup is not quite free (64 of 256 CUs cost ~3 % of the bandwidth): charge 0.05 % each
    int best = grid;
    double best_cost = -1.0;
    for (int g = grid; g >= std::max(1, grid - grid / 8); --g) {
        const long long W = (long long)g * lists_per_wg;
        const double cost = (double)((npanels + W - 1) / W * W) * (1.0 + 0.0005 * (grid - g));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
}

// One pass (<= 64 queries) of the fused search.  The pre-phase (query packing + sampling levels)
// goes to `sp`, the main scan + candidate merge to `sm`; when the two differ (pipelined mode) an
// event orders them, so the pre-phase of the NEXT pass/batch can overlap this pass's main scan.
int enqueue_pass(cmr_index* idx, Workspace* ws, hipStream_t sp, hipStream_t sm, hipStream_t sq, hipEvent_t ev_pre, hipEvent_t ev_scan,
                 hipEvent_t ev_lists_free, const float* q_dev, int nqp,
                 int k, int reserve_cus, int64_t* ids_dev, float* scores_dev, float* min_dev, float* max_dev, bool wide = false,
                 const float* min_score = nullptr, bool quad = false) {
    // quad: a batch of more than one narrow pass on the query-split grid of the NARROW kernel (scan_kernel, qgroups): the
    // caller picks the streams as for a wide pass; geometry, lists and sampling are the narrow kernel's, one set per group
    CmrScanGeom g{};
    const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    int rc = arm_flag(ws, sp);   // zeroed once; the reader re-arms it after reporting
    if (rc) return rc;
    if (quad) wide = false;
    const int narrow_cap = cmr_scan_max_nqt(idx->dtype, idx->dpad) >= 2 ? 64 : 32;
    rc = make_geom(idx, quad ? std::min(nqp, narrow_cap) : nqp, k, true, &g);
    if (rc) return rc;
    g.wide_waves = idx->wide_waves;
    g.wide_abl = idx->wide_abl;
    const int G = quad ? (nqp + g.nqt * 32 - 1) / (g.nqt * 32) : 1;      // query groups
    // the groups' twins re-read each corpus block from L2: default cache policy for them, non-temporal for single-group scans
    g.stream_default_policy = idx->stream_nt < 0 ? (G > 1 ? 1 : 0) : (idx->stream_nt ? 0 : 1);
    const int lists_per_wg = wide ? 1 : CMR_SCAN_WAVES;
    // A synchronous caller's handful of queries (everything on ONE stream) on a corpus beyond the single-launch path: the scan
    // with the finishing stage — no sampling launches, no merge of its own (scan_kernel MODE_FIN; 2 M x 768 bf16 rows, one
    // query: pack 5 + sample 15 + scan 471 + merge 46 us before, pack + scan with ~10 us of finishing after)
    bool fin = idx->scan_fin && !idx->no_sample && !wide && G == 1 && !min_score && sp == sm && sm == sq && g.nqt == 1 && nqp <= idx->fin_max_q &&
               k <= 64 && npanels >= 4096 && npanels >= (long long)idx->n_cu * 2 * CMR_SCAN_WAVES;      // (every wave of the grid has a first panel)
    if (fin) {
        // scan_fin_cap = 256: longer lists than k asks for — the panels a wave scans before the thresholds arrive go to its lists whole (32 keys
        // per query and panel, two to three panels), and a 128-key list is then "nearly full" at the first real candidate (a compaction)
        if ((idx->fin_cap == 256 || idx->fin_cap == 128) && idx->fin_cap > g.cap) {
            CmrScanGeom g2 = g;
            g2.cap = idx->fin_cap;
            if (cmr_scan_geom(&g2) && g2.lds + CMR_FIN_LDS <= 160 * 1024) g = g2;      // (else: the geometry's own lists)
        }
        const bool ok = cmr_ring_audit_ok(g.dtype, 1, g.cap, g.ring, 2);
        if (idx->force_asm == 1 && !ok) fin = false;       // the caller insists on the hand-counted ring: only audited variants
        if (g.lds + CMR_FIN_LDS > 160 * 1024 || g.grid > 512) fin = false;
        else g.asm_ring = (ok && idx->force_asm != 0) ? 1 : 0;
    }
    // Sampling passes (large corpora).  Level i scans S_i strided panels and takes the exact k-th
    // best of that sample per query as the threshold of the next level / of the main scan.  Any
    // subset's k-th best is a valid lower bound of the global k-th best, so results are unchanged;
    // what changes is that only ~S_{i+1}*k/S_i scores per query ever reach the candidate lists
    // (instead of k*ln(rows/k) per wave and query), which keeps every merge at a few thousand keys.
    //   S0 = max(512, 32k) rows, S1 = clamp(N/32, 8*S0, 128*S0) rows (only when N >= 128 Ki rows)
    // Narrow kernel: one sampled panel per wave = one candidate list per panel, and merge_query_kernel takes at
    // most 4096 lists, so S1 is capped there (k > 32 on multi-million-row shards would otherwise overrun it).
    // Wide kernel: the sampling workgroups split the sampled panels among them (one list per workgroup and query).
    long long level_panels[2] = {0, 0};
    int n_levels = 0;
    bool single_level = false;
    // threshold search (min_score): the caller's bound is the initial threshold of every query — already selective, so no
    // sampling passes
    if (!idx->no_sample && npanels >= 256 && !min_score && !fin) {
        const long long s0 = std::max<long long>(16, k);                       // panels
        // A handful of queries on a mid-size corpus (what a synchronous caller issues) is a chain of dependent launches
        // around a short scan: ONE sampling level of 128 panels instead of two saves a scan + merge pair (~45 us of a
        // 0.4 ms call at 1 M rows).  Its threshold lets ~k * npanels / 128 scores per query through — a few slow-path
        // entries per wave as long as queries x panels stays small.
        single_level = !wide && idx->single_level && k <= 32 && nqp <= 8 && npanels >= 4096 && (long long)nqp * npanels <= idx->single_level_max;
        level_panels[n_levels++] = single_level ? 128 : s0;
        if (npanels >= 4096 && !single_level) {
            // wide kernel: 256 queries share a workgroup, so ANY of 8 tiles beating its threshold stalls all four waves at
            // the next barrier — a 4x larger level-1 sample (N/32 rows up to 512 x level 0) took the main pass from 4.09 to
            // 3.76 ms at 10 M rows; the sample itself is cheap there (256 queries per pass over it)
            const long long maxmul = idx->sample_maxmul > 0 ? idx->sample_maxmul : (wide ? 512 : 128);
            long long s1 = std::min<long long>(std::max<long long>(npanels / idx->sample_div, 8 * s0), maxmul * s0);
            if (!wide) s1 = std::min<long long>(s1, kMaxMergeLists);
            if (s1 < npanels / 2) level_panels[n_levels++] = s1;
        }
    }
    const long long max_sample = std::max(level_panels[0], level_panels[1]);
    const bool pipelined = sp != sm;
    if (reserve_cus < 0) {
        if (!max_sample) reserve_cus = 0;
        else if (wide) {
            // Pipelined mode, wide kernel: a sampling workgroup owns a CU (512 registers per wave), so the next batch's
            // pre-phase runs on reserved CUs.  Workgroups are bound to a shader engine (8 CUs) at dispatch and then wait for
            // a free CU THERE: with 240 + 16 workgroups in flight some engines were full and a 16-workgroup sampling pass
            // waited 4.4 ms for the main scan to end (kernel trace), while 224 + 32 — one free CU in every one of the 32
            // engines — flows.  So the reserve is one CU per shader engine; the main pass is matrix-pipe-bound and pays for
            // them in proportion (4.2 -> 4.6 ms at 10 M rows), about what a serialised pre-phase would cost.
            reserve_cus = 32;
        } else {
            // Pipelined mode, reserve chosen by size: the next batch's pre-phase has to fit under this scan (~6 TB/s).
            // It is ~200 us of dependent small kernels plus ~200 us per round of its largest sampling pass on the
            // reserved CUs (every sampling workgroup stages the 96 KiB query tile and its loads crawl while the scan
            // saturates HBM: at 10 M rows 320 workgroups on 26 CUs took 2.3 ms and overran the scan by 90 us, on 34
            // CUs they fit).  Reserved CUs cost the scan bandwidth only at short scans (64 of 256: ~3 % at 1 M rows,
            // nothing measurable at 10 M).  Measured at 1 / 1.25 / 2.5 / 5 / 10 M rows x 768 bf16.
            const double scan_us = (double)npanels * idx->panel_bytes() / 6.0e6;
            const long long rounds = (long long)((0.7 * scan_us - 200.0) / 200.0);
            const long long wgs = std::max<long long>(1, (max_sample + lists_per_wg - 1) / lists_per_wg) * G;
            reserve_cus = rounds >= 1 ? (int)std::min<long long>(64, std::max<long long>(8, (wgs + rounds - 1) / rounds)) : 64;
        }
    }
    // workgroups of a sampling pass over spn panels, and candidate lists it produces
    auto sample_grid = [&](long long spn) -> int {
        if (!wide) return (int)((spn + lists_per_wg - 1) / lists_per_wg);
        // at most 48 sampled panels per workgroup: a level-1 pass pushes ~1 key per query and panel, and a candidate list
        // that fills up (CAP - 32 keys) costs a compaction, whose global loads drain the workgroup's DMA ring.  More
        // workgroups than (reserved) CUs simply run in rounds.
        const int cap_wgs = pipelined && reserve_cus > 0 ? reserve_cus : idx->n_cu;
        return (int)std::min<long long>(spn, std::max<long long>(cap_wgs, (spn + 47) / 48));
    };
    const int Ws = max_sample ? std::max(sample_grid(level_panels[0]), n_levels > 1 ? sample_grid(level_panels[1]) : 0) * lists_per_wg : 0;
    int NQ, W, tiles;
    if (wide) {
        // register-resident queries: 4 waves x 2 (768-d) or 1 (1024-d) tiles of 32, one list row per
        // (workgroup, query); one workgroup per CU
        const int nqb = cmr_wide_queries(idx->dtype, idx->dpad);
        g.nqt = 1;
        g.grid = (int)std::max<long long>(1, std::min<long long>(npanels, idx->n_cu));
        if (reserve_cus > 0 && g.grid > idx->n_cu / 2) g.grid = std::max(g.grid - reserve_cus, idx->n_cu / 2);
        NQ = nqb; W = g.grid; tiles = nqb / 32;
    } else if (G > 1) {
        // query-split grid: G workgroups (one per query tile) share every virtual workgroup's panel ranges; the virtual grid is
        // a multiple of 8 so that the G twins land on one XCD (scan_kernel) and G x virtual grid fills the CUs once
        if (reserve_cus > 0 && g.grid > idx->n_cu / 2) g.grid = std::max(g.grid - reserve_cus, idx->n_cu / 2);
        int vg = g.grid / G;
        if (vg >= 8) vg &= ~7;
        g.grid = std::max(vg, 1);
        NQ = g.nqt * 32; W = g.grid * CMR_SCAN_WAVES; tiles = G * g.nqt;
    } else {
        if (reserve_cus > 0 && g.grid > idx->n_cu / 2) g.grid = std::max(g.grid - reserve_cus, idx->n_cu / 2);
        if (!idx->force_grid) g.grid = balanced_grid(npanels, g.grid, CMR_SCAN_WAVES);
        NQ = g.nqt * 32; W = g.grid * CMR_SCAN_WAVES; tiles = g.nqt;
    }
    const int NQA = G * NQ;          // query slots of the pass over all groups
    // sample passes and the main pass use separate list buffers: in pipelined mode the next batch's
    // sampling runs while this batch's main scan still owns `lists`
    HIP_TRY(ws->qfrag.ensure((size_t)tiles * g.ks * 1024));
    HIP_TRY(ws->lists.ensure((size_t)W * NQA * g.cap * 8));
    HIP_TRY(ws->cnt.ensure((size_t)W * NQA * 4));
    HIP_TRY(ws->mm.ensure((size_t)W * NQA * 8));
    if (Ws) {
        HIP_TRY(ws->s_lists.ensure((size_t)Ws * NQA * g.cap * 8));
        HIP_TRY(ws->s_cnt.ensure((size_t)Ws * NQA * 4));
        HIP_TRY(ws->s_mm.ensure((size_t)Ws * NQA * 8));
    }
    HIP_TRY(ws->tau.ensure((size_t)2 * NQA * 8));
    HIP_TRY(cmr_launch_prep_queries(idx->dtype, q_dev, nqp, idx->dim, idx->dpad, tiles, ws->qfrag.p, ws->flag_ptr, sp));
    CmrScanArgs a{};
    a.corpus = idx->corpus; a.qfrag = ws->qfrag.p; a.nrows = idx->n; a.npanels = (int)npanels; a.k = k;
    a.nq = nqp;
    a.qgroups = G;
    if (min_score) {
        // key > tau  <=>  score >= *min_score: tau = (smallest key with that score) - 1
        HIP_TRY(cmr_launch_fill_threshold(*min_score, NQA, (u64*)ws->tau.p, sp));
        a.tau_init = (u64*)ws->tau.p;
    }
    for (int lv = 0; lv < n_levels; ++lv) {
        const long long spn = level_panels[lv];
        CmrScanGeom gs = g;
        gs.grid = sample_grid(spn);
        const int Wl = gs.grid * lists_per_wg;
        gs.grid *= G;                               // (query-split grid: every group samples the same panels)
        CmrScanArgs as = a;
        as.lists = (u64*)ws->s_lists.p; as.cnt = (int*)ws->s_cnt.p; as.mm = (float2*)ws->s_mm.p;
        // chunks of 8 consecutive panels (384 KiB at 768-d bf16), chunk starts spread evenly over the corpus: a sampling
        // workgroup that hops one panel at a time pays a TLB miss / DRAM page run per 48 KiB (measured 11-14 us per
        // panel under a saturating main scan)
        const int clog = spn >= 64 ? 3 : 0;
        const long long nchunks = (spn + (1 << clog) - 1) >> clog;
        as.sample_waves = (int)spn; as.sample_chunk_log2 = clog; as.sample_stride = (int)(npanels / nchunks);
        u64* tau_out = (u64*)ws->tau.p + (size_t)(lv & 1) * NQA;
        HIP_TRY(wide ? cmr_launch_scan_wide(gs, as, sp) : cmr_launch_scan_topk(gs, as, sp));
        if (single_level && idx->tau_in_scan && NQ == 32 && nqp <= 2 && k <= 64) {
            // the one sampling level of ONE or TWO queries (what a synchronous caller issues): the main scan's workgroups derive
            // the thresholds from these lists themselves (scan_kernel) — no merge launch between the two scans.  Every
            // workgroup reads the whole sample of its queries (32 KiB each): with 8 queries that costs more than the merge
            // launch it saves (1 M rows: 371 -> 405 us per call), with one it wins (344 -> 330 us)
            a.sample_lists = (const u64*)ws->s_lists.p; a.sample_cnt = (const int*)ws->s_cnt.p; a.sample_W = Wl;
            a.tau_init = nullptr;
            continue;
        }
        HIP_TRY(cmr_launch_merge_query((const u64*)ws->s_lists.p, (const int*)ws->s_cnt.p, Wl, NQ, g.cap, nqp, k, nullptr, 0, nullptr,
                                       nullptr, nullptr, nullptr, tau_out, sp, G > 1));
        a.tau_init = tau_out;
    }
    if (sp != sm) {
        if (ev_lists_free) HIP_TRY(hipStreamWaitEvent(sp, ev_lists_free, 0));   // previous merge of this slot's lists
        HIP_TRY(hipEventRecord(ev_pre, sp));
        HIP_TRY(hipStreamWaitEvent(sm, ev_pre, 0));
    }
    a.lists = (u64*)ws->lists.p; a.cnt = (int*)ws->cnt.p; a.mm = (float2*)ws->mm.p;
    ProfEvent pe{};
    bool prof = false;
    {
        std::lock_guard<std::mutex> pg(idx->prof_mu);
        prof = idx->prof_on && (idx->prof_seq++ % (unsigned)idx->prof_every) == 0;
    }
    if (prof) {
        HIP_TRY(hipEventCreate(&pe.a));
        HIP_TRY(hipEventCreate(&pe.b));
        HIP_TRY(hipEventRecord(pe.a, sm));
    }
    if (G > 1) g.grid *= G;
    const int* fin_state = nullptr;
    if (fin) {
        const int kFinDenseCap = idx->fin_dense;
        if (!ws->fin_ctl_armed) {               // zeroed once; the kernel's last workgroup re-arms the words
            HIP_TRY(ws->fin_ctl.ensure(CMR_FIN_CTL * sizeof(int)));
            HIP_TRY(hipMemsetAsync(ws->fin_ctl.p, 0, CMR_FIN_CTL * sizeof(int), sm));
            ws->fin_ctl_armed = true;           // (only once the zeroing is in the stream: garbage counters would end in garbage results)
        }
        HIP_TRY(ws->fin_pmax.ensure((size_t)32 * CMR_FIN_SLOTS * 8));
        HIP_TRY(ws->fin_tau.ensure(32 * 8));
        HIP_TRY(ws->fin_mm.ensure((size_t)32 * 512 * 8));
        a.fin_mm = (u64*)ws->fin_mm.p;
        HIP_TRY(ws->fin_dense.ensure((size_t)32 * kFinDenseCap * 8));
        a.fin = (int*)ws->fin_ctl.p; a.fin_pmax = (u64*)ws->fin_pmax.p; a.fin_tau = (u64*)ws->fin_tau.p; a.fin_dense = (u64*)ws->fin_dense.p;
        // the first workgroups through their first panels supply the thresholds: 64 of them (512 maxima; the 64th of 256 is through at a
        // third of the time the slowest of 128 takes), 128 from 4 M rows up (a looser threshold lets k x panels / maxima keys per query through)
        a.fin_first = std::min(g.grid, idx->n_cu);      // (181 registers: one workgroup per CU)
        a.fin_wgs = std::min(std::min(idx->fin_suppliers > 0 ? idx->fin_suppliers : (npanels >= 131072 ? 128 : 64), CMR_FIN_SLOTS / CMR_SCAN_WAVES), a.fin_first);
        // the golden-ratio multiple of the grid, moved to the next value coprime with it: the first fin_wgs workgroups' ranges spread evenly
        a.fin_mul = 1;
        for (int m = std::max(1, (int)(g.grid * 0.6180339887)); m < g.grid; ++m)
            if (std::gcd(m, g.grid) == 1) { a.fin_mul = m; break; }
        a.fin_dcap = kFinDenseCap;
        a.fin_spin = idx->fin_spin;
        a.out_ids = ids_dev; a.out_scores = scores_dev; a.out_min = min_dev; a.out_max = max_dev; a.id_base = kernel_id_base(idx);
        fin_state = (const int*)ws->fin_ctl.p + CMR_FIN_STATE;
        // the synchronous host API (results in its mapped buffer, nothing to remap behind the scan): the kernel reports its state in the
        // caller's done word and the merge launch — which would return in its first instruction on state 1 — is issued only on state 2
        if (ws->done_ptr && idx->blk_local.size() <= 1) a.fin_done = ws->done_ptr;
    }
    HIP_TRY(wide ? cmr_launch_scan_wide(g, a, sm) : fin ? cmr_launch_scan_fin(g, a, sm) : cmr_launch_scan_topk(g, a, sm));
    if (prof) {
        HIP_TRY(hipEventRecord(pe.b, sm));
        std::lock_guard<std::mutex> pg(idx->prof_mu);
        idx->prof_events.push_back(pe);
        idx->prof_bytes = algorithmic_bytes(idx, nqp, k);
    }
    if (a.fin_done) {
        ws->lazy.due = true;
        ws->lazy.lists = (const u64*)ws->lists.p; ws->lazy.cnt = (const int*)ws->cnt.p; ws->lazy.W = W; ws->lazy.NQ = NQ; ws->lazy.cap = g.cap; ws->lazy.nqp = nqp; ws->lazy.k = k;
        ws->lazy.mm = (const float2*)ws->mm.p; ws->lazy.id_base = kernel_id_base(idx); ws->lazy.ids = ids_dev; ws->lazy.scores = scores_dev; ws->lazy.mn = min_dev; ws->lazy.mx = max_dev;
        ws->lazy.state = fin_state;
        ws->done_used = true;
        return CMR_OK;
    }
    if (sq != sm) {   // pipelined: the candidate merge leaves the scan stream so the next main scan starts at once
        HIP_TRY(hipEventRecord(ev_scan, sm));
        HIP_TRY(hipStreamWaitEvent(sq, ev_scan, 0));
    }
    HIP_TRY(cmr_launch_merge_query((const u64*)ws->lists.p, (const int*)ws->cnt.p, W, NQ, g.cap, nqp, k, (const float2*)ws->mm.p,
                                   kernel_id_base(idx), ids_dev, scores_dev, min_dev, max_dev, nullptr, sq, G > 1, fin_state));
    return remap_ids_enqueue(idx, ids_dev, (long long)nqp * k, sq);
}

// Batches of more than one narrow pass: which kernel runs them, and how many queries it takes per corpus pass (0 = narrow passes).
// wide_mode 1: the register-resident wide kernel only (768-d: 256 queries, 1024-d: 128; 16-bit indexes) — other shapes run
// narrow passes; 2: always the query-split grid of the narrow kernel (4 tiles of 64 — or of 32 where the LDS holds one tile —
// in one pass, any dim and dtype); 0 (default): the wide kernel where it exists, the query-split grid everywhere else.
// Measured (profiles/r4_measurements.md, MI355X, 768-d bf16): at 10 M rows the wide kernel runs B = 256 in 3.79 ms, the grid in
// 6.24 (its twins stay in step only partly: 43 % L2 hits of an ideal 75 %, the rest comes over the fabric), four narrow passes in
// 9.4; at B = 128 the grid is level with the wide kernel (3.47 vs 3.26 ms; 0.39 vs 0.47 on a 1.25 M-row shard).
bool wide_pass_is_quad(const cmr_index* idx) {
    if (idx->wide_mode == 2) return true;
    if (idx->wide_mode == 1) return false;
    return cmr_wide_queries(idx->dtype, idx->dpad) == 0;
}
// A pass of 65 .. 128 queries over a SHORT scan (< 1 ms at the streaming rate: shards up to ~4 M x 768 bf16 rows) runs on the
// query-split grid although the shape has a wide kernel: two query tiles per corpus block are within what an XCD's L2 hands on
// (0.39 vs 0.47 ms at 1.25 M rows, B = 128; at 10 M rows the wide kernel wins, 3.26 vs 3.47 — profiles/r4_wide_routes_ab.txt)
bool short_two_tile_pass(const cmr_index* idx, int left) {
    if (idx->wide_mode != 0 || idx->no_wide || cmr_wide_queries(idx->dtype, idx->dpad) == 0) return false;
    if (cmr_scan_max_nqt(idx->dtype, idx->dpad) < 2 || left <= 64 || left > 128) return false;
    const double scan_us = (double)((idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS) * idx->panel_bytes() / 6.0e6;
    return scan_us < 1000.0;
}
int wide_pass_queries(const cmr_index* idx) {
    if (idx->no_wide) return 0;
    if (!wide_pass_is_quad(idx)) return cmr_wide_queries(idx->dtype, idx->dpad);
    return 4 * (cmr_scan_max_nqt(idx->dtype, idx->dpad) >= 2 ? 64 : 32);
}

// 0: the general pack / [sample] / scan / merge chain; 1: single launch, <= 1024 rows; 2: single launch, hierarchical
// selection (<= 16 queries, k <= 64, up to 64 K rows while workgroups x k <= 1024) — see tiny_search_kernel
int small_path_kind(const cmr_index* idx, int nq, int k, bool threshold_search) {
    if (idx->no_tiny || threshold_search || idx->n <= 0 || nq > 16 || k > CMR_MAX_K) return 0;
    const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    if (npanels > idx->small_max_panels) return 0;
    const int ks = idx->dtype == CMR_F32 ? idx->dpad / 8 : idx->dpad / 16;
    if ((size_t)ks * 1024 > 143 * 1024) return 0;                    // the packed operands of one query tile (+ 17 KiB of static LDS) must fit
    const int kind = cmr_tiny_kind(nq, (int)npanels, k, idx->tiny_multi, idx->small_max_panels);
    return (kind == 2 && idx->no_small) ? 0 : kind;
}

int search_enqueue(cmr_index* idx, Workspace* ws, const float* q_dev, int nq, int k, int64_t* ids_dev, float* scores_dev,
                   float* min_dev, float* max_dev, const float* min_score = nullptr) {
    if (k > CMR_MAX_K) {
        if (min_score) return fail(CMR_ERR_UNSUPPORTED, "threshold search supports k <= %d", CMR_MAX_K);
        return search_large_k_enqueue(idx, ws, q_dev, nq, k, ids_dev, scores_dev, min_dev, max_dev);
    }
    const int max_nqt = cmr_scan_max_nqt(idx->dtype, idx->dpad);
    if (small_path_kind(idx, nq, k, min_score != nullptr)) {   // small corpus, few queries: ONE launch does packing, scan, selection and min/max
        const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
        { int rc_ = arm_flag(ws, ws->stream); if (rc_) return rc_; }
        HIP_TRY(ws->d_out.ensure(cmr_tiny_scratch_bytes(nq, (int)npanels, k, idx->tiny_multi, idx->small_max_panels)));
        if (!ws->arrive.p) {          // arrival counter of the multi-workgroup search: zeroed once, re-armed by the kernel
            HIP_TRY(ws->arrive.ensure(sizeof(int)));
            HIP_TRY(hipMemsetAsync(ws->arrive.p, 0, sizeof(int), ws->stream));
        }
        HIP_TRY(cmr_launch_tiny_search(idx->dtype, idx->corpus, q_dev, nq, idx->dim, idx->dpad, idx->n, k, kernel_id_base(idx), ws->d_out.p,
                                       ids_dev, scores_dev, min_dev, max_dev, ws->flag_ptr, idx->tiny_multi ? (int*)ws->arrive.p : nullptr, idx->small_max_panels, ws->stream,
                                       (ws->done_ptr && idx->blk_local.size() <= 1) ? ws->done_ptr : nullptr));
        if (ws->done_ptr && idx->blk_local.size() <= 1) ws->done_used = true;
        return remap_ids_enqueue(idx, ids_dev, (long long)nq * k, ws->stream);
    }
    const int narrow = (nq > 32 && max_nqt >= 2) ? 64 : 32;
    const int wideq = wide_pass_queries(idx);
    const bool quad = wide_pass_is_quad(idx);
    for (int q0 = 0; q0 < nq;) {
        const int left = nq - q0;
        const bool wide = wideq > 0 && left > narrow;          // more than one narrow pass left: go wide
        const int nqp = std::min(wide ? wideq : narrow, left);
        const bool qp = wide && (quad || short_two_tile_pass(idx, left));
        int rc = enqueue_pass(idx, ws, ws->stream, ws->stream, ws->stream, nullptr, nullptr, nullptr, q_dev + (size_t)q0 * idx->dim, nqp, k, 0,
                              ids_dev + (size_t)q0 * k, scores_dev + (size_t)q0 * k, min_dev ? min_dev + q0 : nullptr,
                              max_dev ? max_dev + q0 : nullptr, wide && !qp, min_score, qp);
        if (rc) return rc;
        q0 += nqp;
    }
    return CMR_OK;
}

// Streams and per-slot events of the pipelined search, created on first use (idx->pipe_mu held).  Everything is built into
// locals and committed to idx->pipe only when ALL of it exists: a failure half way leaves the index without a pipeline (the next
// call tries again), never with a half-built one that a later call would take for complete.  A device / driver that refuses
// CU-masked streams gets plain streams (the masks buy time, not results).
void destroy_streams(std::initializer_list<hipStream_t*> sts) {
    for (hipStream_t* st : sts) if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
}
int create_pipe_streams(cmr_index* idx, Pipe& T, int mask, bool dual) {
    HIP_TRY(hipStreamCreateWithFlags(&T.sq, hipStreamNonBlocking));
    if (mask) {
        // wide batches: the matrix-pipe-bound kernel wants CUs — n_cu - 32 for its scans (28 per XCD), the 32 it used to leave
        // free by trimming its grid for the pre-phase; two scan streams for short scans as below
        uint32_t wscan[8], wrest[8];
        for (int w = 0; w < 8; ++w) {
            wscan[w] = mask == 2 ? 0x0FFFFFFFu : (w < 7 ? 0xFFFFFFFFu : 0u);
            wrest[w] = ~wscan[w];
        }
        HIP_TRY(hipExtStreamCreateWithCUMask(&T.wm, 8, wscan));
        if (dual) HIP_TRY(hipExtStreamCreateWithCUMask(&T.wm2, 8, wscan));
        HIP_TRY(hipExtStreamCreateWithCUMask(&T.wp, 8, wrest));
        T.wide_cus = idx->n_cu - 32;
        // Scans of the narrow kernel on n_cu - 64 CUs, their pre-phases on the other 64: the reservation that trimming the
        // grid only approximates, made explicit — and the precondition for TWO scan streams: the next scan's workgroups then
        // start on whatever CU of the scan set falls free (no idle gap, the tail of one scan under the ramp of the next),
        // never on the pre-phase's CUs.  Measured (profiles/r3_pipe_cu_mask_dual_scan.txt): 1 M rows 0.270 -> 0.250 ms per
        // step, 1.25 M 0.334 -> 0.308, 10 M 2.396 -> 2.331; masks alone <= 2 %, two streams without masks slower.
        // Mask bit i is CU i of the driver's enumeration, which interleaves the XCDs: the first 192 bits are 24 CUs of each.
        uint32_t scan[8], rest[8];
        for (int w = 0; w < 8; ++w) {
            scan[w] = mask == 2 ? 0x00FFFFFFu : (w < 6 ? 0xFFFFFFFFu : 0u);      // 2: 24 of every 32 bits (same split if 32 consecutive bits were one XCD)
            rest[w] = ~scan[w];
        }
        HIP_TRY(hipExtStreamCreateWithCUMask(&T.sm, 8, scan));
        if (dual) HIP_TRY(hipExtStreamCreateWithCUMask(&T.sm2, 8, scan));
        HIP_TRY(hipExtStreamCreateWithCUMask(&T.sp, 8, rest));
        T.scan_cus = idx->n_cu - 64;
        for (hipStream_t* st : {&T.usp, &T.usm, &T.uwp, &T.uwm}) HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&T.wp, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&T.wm, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&T.sp, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&T.sm, hipStreamNonBlocking));
        if (dual) HIP_TRY(hipStreamCreateWithFlags(&T.sm2, hipStreamNonBlocking));
    }
    return CMR_OK;
}
int ensure_pipe(cmr_index* idx) {
    Pipe& P = idx->pipe;
    if (P.sq) return CMR_OK;
    int mask = idx->cu_mask < 0 ? (idx->n_cu == 256 ? 1 : 0) : (idx->n_cu == 256 ? idx->cu_mask : 0);
    hipEvent_t ev[CMR_PIPE_SLOTS][3] = {};
    Pipe T;
    auto undo = [&]() {
        destroy_streams({&T.sp, &T.sm, &T.sm2, &T.wp, &T.wm, &T.wm2, &T.sq, &T.usp, &T.usm, &T.uwp, &T.uwm});
        for (auto& slot : ev) for (hipEvent_t& e : slot) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        T.scan_cus = T.wide_cus = 0;
    };
    int rc = create_pipe_streams(idx, T, mask, idx->dual_scan < 0 ? mask != 0 : idx->dual_scan != 0);
    if (rc && mask) {                     // no CU-masked streams here: plain ones
        undo();
        mask = 0;
        rc = create_pipe_streams(idx, T, 0, idx->dual_scan > 0);
    }
    if (rc) { undo(); return rc; }
    auto make_events = [&]() -> int {
        for (auto& slot : ev) for (hipEvent_t& e : slot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return CMR_OK;
    };
    rc = make_events();
    if (rc) { undo(); return rc; }
    // commit (slots keep their workspaces: none exists before the first pipelined call)
    P.sp = T.sp; P.sm = T.sm; P.sm2 = T.sm2; P.wp = T.wp; P.wm = T.wm; P.wm2 = T.wm2; P.usp = T.usp; P.usm = T.usm; P.uwp = T.uwp; P.uwm = T.uwm;
    P.scan_cus = T.scan_cus; P.wide_cus = T.wide_cus;
    for (int i = 0; i < CMR_PIPE_SLOTS; ++i) {
        P.slot[i].pre_done = ev[i][0]; P.slot[i].main_done = ev[i][1]; P.slot[i].scan_done = ev[i][2];
        P.slot[i].ws.stream = P.sm;
    }
    P.sq = T.sq;                          // the "pipeline exists" marker: last
    return CMR_OK;
}

// Pipelined search: three internal streams.  Pre-phases run on `sp` back to back, candidate merges
// on `sq`; main scans are serialised on `sm` (two HBM-bound scans at once only slow each other down) and leave
// `reserve_cus` CUs free, on which the next pass's sampling scans and the merges run concurrently.
int search_pipelined_enqueue(cmr_index* idx, const float* q_dev, int nq, int k, int64_t* ids_dev, float* scores_dev, float* min_dev,
                             float* max_dev, hipEvent_t wait_event, hipEvent_t* done_event, const float* min_score = nullptr) {
    std::lock_guard<std::mutex> pl(idx->pipe_mu);
    Pipe& P = idx->pipe;
    { int rc_ = ensure_pipe(idx); if (rc_) return rc_; }
    P.nslots = std::min(CMR_PIPE_SLOTS, std::max(2, idx->pipe_slots));
    if (k > CMR_MAX_K) return fail(CMR_ERR_UNSUPPORTED, "pipelined search supports k <= %d", CMR_MAX_K);
    const int max_nqt = cmr_scan_max_nqt(idx->dtype, idx->dpad);
    const int narrow = (nq > 32 && max_nqt >= 2) ? 64 : 32;
    const int wideq = wide_pass_queries(idx);
    const bool quad = wide_pass_is_quad(idx);
    // Scans shorter than 1 ms at the streaming rate (shards up to ~4 M x 768 bf16 rows) run on the CU-masked streams — and
    // alternate between two of them; longer ones on the unmasked twins with the trimmed grid: in bench.py's flow the masks cost
    // the 10 M-row scans CUs (same-box A/B: B = 64 step 2.441 vs 2.396 ms, B = 256 4.444 vs 4.069) where they bought the
    // short ones 7-8 %.  pipe_cu_mask = 1 | 2 forces the masks for every scan, 0 creates none.
    const double scan_us = (double)((idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS) * idx->panel_bytes() / 6.0e6;
    const bool masked = P.scan_cus != 0 && (idx->cu_mask > 0 || scan_us < 1000.0);
    const bool twins = P.scan_cus != 0 && !masked;      // masked streams exist but this call's scans use the unmasked ones
    P.last_masked = masked ? 1 : 0;
    hipStream_t const nsp = twins ? P.usp : P.sp, wsp = twins ? P.uwp : P.wp;
    if (wait_event) {      // inputs ready: both pre-phase streams may read them
        HIP_TRY(hipStreamWaitEvent(nsp, wait_event, 0));
        if (wideq > 0 && nq > narrow) HIP_TRY(hipStreamWaitEvent(wsp, wait_event, 0));
    }
    PipeSlot* last = nullptr;
    for (int q0 = 0; q0 < nq;) {
        const int left = nq - q0;
        const bool wide = wideq > 0 && left > narrow;
        const int nqp = std::min(wide ? wideq : narrow, left);
        const bool qp = wide && (quad || short_two_tile_pass(idx, left));
        PipeSlot* sl = &P.slot[P.next++ % (unsigned)P.nslots];
        hipStream_t sp = wide ? wsp : nsp;
        if (sl->used) {
            // The pre-phase rewrites the slot's query fragments / thresholds: free once the slot's previous
            // main scan is over.  Its candidate lists are still being merged (on sq) at that point, so only the
            // new MAIN scan waits for that merge — a full scan period later, i.e. never in practice; that wait
            // sits at the END of the pre-phase (enqueue_pass, before pre_done is recorded): every wait packet on
            // the scan stream itself costs ~10 us between two main scans.
            HIP_TRY(hipStreamWaitEvent(sp, sl->scan_done, 0));
        }
        // Two scan streams for the narrow kernel: consecutive main scans are not ordered by a stream any more — the second
        // one's workgroups start as the first one's retire.  A slot's own scans stay ordered through its events (pre-phase ->
        // scan -> merge -> next pre-phase), and nothing else is shared between two batches.
        // Only short scans (< 1 ms at the streaming rate: shards up to ~4 M x 768 bf16 rows) alternate: there the ramp / tail /
        // packet gap is 7-8 % of a step (1 M rows 0.270 -> 0.250 ms), at 10 M rows 2.7 % — and overlapping launches have no
        // per-launch duration any more (a kernel's begin-to-end then includes the wait for the previous scan's CUs), which
        // is what the roofline of the long headline scan is measured with.
        const bool dual = !twins && (wide ? P.wm2 != nullptr : P.sm2 != nullptr) && (idx->dual_scan > 0 || scan_us < 1000.0);
        if (!wide) idx->dual_active = dual ? 1 : 0;
        else idx->dual_wide_active = dual ? 1 : 0;
        hipStream_t sm = twins ? (wide ? P.uwm : P.usm)
                               : wide ? ((dual && (P.nwscan++ & 1)) ? P.wm2 : P.wm) : ((dual && (P.nscan++ & 1)) ? P.sm2 : P.sm);
        int rc = enqueue_pass(idx, &sl->ws, sp, sm, P.sq, sl->pre_done, sl->scan_done, sl->used ? sl->main_done : nullptr, q_dev + (size_t)q0 * idx->dim, nqp, k,
                              (masked && !wide) ? idx->n_cu - P.scan_cus : (masked && wide) ? idx->n_cu - P.wide_cus : idx->reserve_cus,
                              ids_dev + (size_t)q0 * k, scores_dev + (size_t)q0 * k, min_dev ? min_dev + q0 : nullptr,
                              max_dev ? max_dev + q0 : nullptr, wide && !qp, min_score, qp);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(sl->main_done, P.sq));
        sl->used = true;
        last = sl;
        q0 += nqp;
    }
    if (done_event) *done_event = last ? last->main_done : nullptr;
    return CMR_OK;
}

int scores_enqueue(cmr_index* idx, Workspace* ws, const float* q_dev, int nq, float* out_dev, long long ld) {
    hipStream_t s = ws->stream;
    CmrScanGeom g{};
    const int max_nqt = cmr_scan_max_nqt(idx->dtype, idx->dpad);
    const int per_pass = (nq > 32 && max_nqt >= 2) ? 64 : 32;
    const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    { int rc_ = arm_flag(ws, s); if (rc_) return rc_; }
    for (int q0 = 0; q0 < nq; q0 += per_pass) {
        const int nqp = std::min(per_pass, nq - q0);
        int rc = make_geom(idx, nqp, 1, false, &g);
        if (rc) return rc;
        HIP_TRY(ws->qfrag.ensure((size_t)g.nqt * g.ks * 1024));
        HIP_TRY(cmr_launch_prep_queries(idx->dtype, q_dev + (size_t)q0 * idx->dim, nqp, idx->dim, idx->dpad, g.nqt, ws->qfrag.p,
                                        ws->flag_ptr, s));
        CmrScanArgs a{};
        a.corpus = idx->corpus; a.qfrag = ws->qfrag.p; a.nrows = idx->n; a.npanels = (int)npanels; a.k = 1;
        a.scores = out_dev + (size_t)q0 * ld; a.ld = ld; a.nq = nqp;
        HIP_TRY(cmr_launch_scan_scores(g, a, s));
    }
    return CMR_OK;
}

int check_query_flag(Workspace* ws) {
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, ws->flag_ptr, sizeof(int), hipMemcpyDeviceToHost, ws->stream));
    HIP_TRY(hipStreamSynchronize(ws->stream));
    if (h) {
        HIP_TRY(hipMemsetAsync(ws->flag_ptr, 0, sizeof(int), ws->stream));
        return fail(CMR_ERR_NONFINITE, "query contains NaN/Inf");
    }
    return CMR_OK;
}

int grow(cmr_index* idx, long long need_panels) {
    if (need_panels <= idx->cap_panels) return CMR_OK;
    long long new_cap = std::max(need_panels, idx->cap_panels * 2);
    new_cap = std::max<long long>(new_cap, 8);
    const size_t pb = idx->panel_bytes();
    // outstanding async searches may still read the old buffer
    HIP_TRY(hipDeviceSynchronize());
    void* nc = nullptr;
    HIP_TRY(hipMalloc(&nc, (size_t)new_cap * pb + CMR_CORPUS_SLACK));
    HIP_TRY(hipMemsetAsync(nc, 0, (size_t)new_cap * pb + CMR_CORPUS_SLACK, nullptr));
    const long long used_panels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
    if (idx->corpus && used_panels)
        HIP_TRY(hipMemcpyAsync(nc, idx->corpus, (size_t)used_panels * pb, hipMemcpyDeviceToDevice, nullptr));
    float* ns = nullptr;
    if ((idx->flags & CMR_FLAG_KEEP_F32) && idx->dtype != CMR_F32) {
        HIP_TRY(hipMalloc((void**)&ns, (size_t)new_cap * CMR_PANEL_ROWS * idx->dim * sizeof(float)));
        if (idx->shadow && idx->n)
            HIP_TRY(hipMemcpyAsync(ns, idx->shadow, (size_t)idx->n * idx->dim * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    }
    HIP_TRY(hipDeviceSynchronize());
    if (idx->corpus) HIP_TRY(hipFree(idx->corpus));
    if (idx->shadow) HIP_TRY(hipFree(idx->shadow));
    idx->corpus = nc;
    idx->shadow = ns;
    idx->cap_panels = new_cap;
    return CMR_OK;
}

// rows_dev: fp32 [n, dim] on the device; converts on `s`, checks finiteness, bumps n.
int append_from_device(cmr_index* idx, const float* rows_dev, long long n, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(idx->d_flag, 0, sizeof(int), s));
    HIP_TRY(cmr_launch_convert_rows(idx->dtype, rows_dev, n, idx->dim, idx->dpad, idx->n, idx->corpus, idx->shadow, idx->d_flag, s));
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, idx->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h) return fail(CMR_ERR_NONFINITE, "appended rows contain NaN/Inf (index unchanged)");
    idx->n += n;
    return CMR_OK;
}

}  // namespace

// ---- for ppr.hip: one host query -> N raw scores in the workspace's device buffer, index lock + workspace kept until release
namespace { thread_local Workspace* tl_scores_ws = nullptr; }

int cmr_index_scores_to_device(cmr_index_t* idx, const float* q_host, float** scores_dev, long long* n, void** stream) {
    if (!idx || !q_host || !scores_dev || !n || !stream) return fail(CMR_ERR_INVALID, "NULL argument");
    if (tl_scores_ws) return fail(CMR_ERR_INVALID, "nested cmr_index_scores_to_device on one thread");
    idx->mu.lock_shared();
    int rc = set_device(idx->device);
    Workspace* ws = rc ? nullptr : acquire_ws(idx, nullptr, false);
    if (!rc && !ws) rc = fail(CMR_ERR_HIP, "could not create a workspace stream");
    if (!rc) {
        hipStream_t s = ws->stream;
        auto body = [&]() -> int {
            const size_t q_bytes = (size_t)idx->dim * 4;
            HIP_TRY(ws->d_q.ensure(q_bytes));
            HIP_TRY(ws->d_out.ensure(std::max<size_t>((size_t)idx->n * 4, 8)));
            HIP_TRY(ws->ensure_pin(q_bytes));
            memcpy(ws->h_pin, q_host, q_bytes);
            HIP_TRY(hipMemcpyAsync(ws->d_q.p, ws->h_pin, q_bytes, hipMemcpyHostToDevice, s));
            if (idx->n == 0) return CMR_OK;
            return scores_enqueue(idx, ws, (const float*)ws->d_q.p, 1, (float*)ws->d_out.p, idx->n);
        };
        rc = body();
        if (rc) (void)hipStreamSynchronize(s);
    }
    if (rc) {
        if (ws) release_ws(idx, ws);
        idx->mu.unlock_shared();
        return rc;
    }
    tl_scores_ws = ws;
    *scores_dev = (float*)ws->d_out.p; *n = idx->n; *stream = (void*)ws->stream;
    return CMR_OK;
}

int cmr_index_scores_release(cmr_index_t* idx) {
    Workspace* ws = tl_scores_ws;
    if (!idx || !ws) return CMR_OK;
    tl_scores_ws = nullptr;
    int rc = ws->flag_ptr ? check_query_flag(ws) : CMR_OK;
    release_ws(idx, ws);
    idx->mu.unlock_shared();
    return rc;
}

// ============================================================================================
extern "C" {

int32_t cmr_abi_version(void) { return CMR_ABI_VERSION; }
const char* cmr_last_error(void) { return g_err.c_str(); }

int32_t cmr_device_count(int32_t* n) {
    if (!n) return fail(CMR_ERR_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
    return CMR_OK;
}

int32_t cmr_device_info(int32_t device_id, char* name, int32_t name_len, int32_t* n_cu, int64_t* hbm_bytes) {
    int rc = check_device(device_id);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return CMR_OK;
}

int32_t cmr_index_create(int32_t device_id, int32_t dim, int32_t dtype, int64_t capacity_hint, uint32_t flags, cmr_index_t** out) {
    if (!out) return fail(CMR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dim <= 0 || dim > 16384) return fail(CMR_ERR_INVALID, "dim %d out of range", dim);
    if (dtype != CMR_F32 && dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "unknown dtype %d", dtype);
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    cmr_index* idx = new cmr_index();
    idx->device = device_id;
    idx->dim = dim;
    idx->dpad = round_up(dim, 128);
    idx->dtype = dtype;
    idx->flags = flags;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) idx->n_cu = prop.multiProcessorCount;
#ifdef CMR_DEV_KNOBS
    options_from_env(idx);
#endif
    if (cmr_scan_max_nqt(dtype, idx->dpad) == 0) {
        delete idx;
        return fail(CMR_ERR_UNSUPPORTED, "dim %d (padded %d) exceeds the LDS-resident query tile for dtype %d", dim, round_up(dim, 128), dtype);
    }
    if (hipMalloc((void**)&idx->d_flag, sizeof(int)) != hipSuccess) { delete idx; return fail(CMR_ERR_OOM, "hipMalloc flag"); }
    const long long hint_panels = std::max<long long>((capacity_hint + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS, 8);
    rc = grow(idx, hint_panels);
    if (rc) { (void)hipFree(idx->d_flag); delete idx; return rc; }
    *out = idx;
    return CMR_OK;
}

int32_t cmr_index_destroy(cmr_index_t* idx) {
    if (!idx) return CMR_OK;
    {
        std::unique_lock<std::shared_mutex> lk(idx->mu);
        (void)hipSetDevice(idx->device);
        (void)hipDeviceSynchronize();
        for (Workspace* w : idx->free_ws) { w->release(); delete w; }
        for (auto& kv : idx->stream_ws) { kv.second->release(); delete kv.second; }
        for (ProfEvent& pe : idx->prof_events) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
        for (int i = 0; i < CMR_PIPE_SLOTS; ++i) {
            idx->pipe.slot[i].ws.stream = nullptr;
            idx->pipe.slot[i].ws.release();
            if (idx->pipe.slot[i].pre_done) (void)hipEventDestroy(idx->pipe.slot[i].pre_done);
            if (idx->pipe.slot[i].main_done) (void)hipEventDestroy(idx->pipe.slot[i].main_done);
            if (idx->pipe.slot[i].scan_done) (void)hipEventDestroy(idx->pipe.slot[i].scan_done);
        }
        if (idx->pipe.sp) (void)hipStreamDestroy(idx->pipe.sp);
        if (idx->pipe.sm) (void)hipStreamDestroy(idx->pipe.sm);
        if (idx->pipe.sm2) (void)hipStreamDestroy(idx->pipe.sm2);
        if (idx->pipe.wp) (void)hipStreamDestroy(idx->pipe.wp);
        if (idx->pipe.wm) (void)hipStreamDestroy(idx->pipe.wm);
        if (idx->pipe.wm2) (void)hipStreamDestroy(idx->pipe.wm2);
        for (hipStream_t st : {idx->pipe.usp, idx->pipe.usm, idx->pipe.uwp, idx->pipe.uwm}) if (st) (void)hipStreamDestroy(st);
        if (idx->pipe.sq) (void)hipStreamDestroy(idx->pipe.sq);
        idx->stage.release();
        if (idx->h_pin) { (void)hipHostFree(idx->h_pin); idx->h_pin = nullptr; idx->h_pin_cap = 0; }
        if (idx->corpus) (void)hipFree(idx->corpus);
        if (idx->shadow) (void)hipFree(idx->shadow);
        if (idx->d_flag) (void)hipFree(idx->d_flag);
        if (idx->d_blk) (void)hipFree(idx->d_blk);
        for (void* p : idx->blk_retired) (void)hipFree(p);
    }
    delete idx;
    return CMR_OK;
}

int32_t cmr_index_size(cmr_index_t* idx, int64_t* n_rows) {
    if (!idx || !n_rows) return fail(CMR_ERR_INVALID, "NULL argument");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    *n_rows = idx->n;
    return CMR_OK;
}

int32_t cmr_index_info(cmr_index_t* idx, int32_t* dim, int32_t* dtype, int64_t* capacity_rows, int64_t* device_bytes) {
    if (!idx) return fail(CMR_ERR_INVALID, "NULL index");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (dim) *dim = idx->dim;
    if (dtype) *dtype = idx->dtype;
    if (capacity_rows) *capacity_rows = idx->cap_panels * CMR_PANEL_ROWS;
    if (device_bytes)
        *device_bytes = (int64_t)(idx->cap_panels * idx->panel_bytes() + CMR_CORPUS_SLACK +
                                  (idx->shadow ? (size_t)idx->cap_panels * CMR_PANEL_ROWS * idx->dim * 4 : 0));
    return CMR_OK;
}

int32_t cmr_index_append(cmr_index_t* idx, const float* rows, int64_t n) {
    if (!idx || (n > 0 && !rows)) return fail(CMR_ERR_INVALID, "NULL argument");
    if (n < 0) return fail(CMR_ERR_INVALID, "n < 0");
    if (n == 0) return CMR_OK;
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (idx->n + n >= 0xFFFFFFF0ll) return fail(CMR_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
    if (global_id_overflow(idx, n)) return fail(CMR_ERR_UNSUPPORTED, "appending %lld rows takes this shard's global ids beyond the 32-bit row of the packed candidate exchange", (long long)n);
    rc = grow(idx, (idx->n + n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS);
    if (rc) return rc;
    if (idx->zero_copy && (size_t)n * idx->dim * 4 <= kMappedAppendMax) {
        // A handful of rows (MemoryPool's per-cycle nodes, a store's freshly inserted strings): rows and the non-finite flag
        // in a pinned, device-mapped buffer that the convert kernel reads / writes itself — one launch and one
        // synchronisation instead of a pageable H2D, a memset, the launch, the flag's D2H and the synchronisation.
        const size_t bytes = (size_t)n * idx->dim * 4;
        if (bytes + 256 > idx->h_pin_cap) {
            if (idx->h_pin) { HIP_TRY(hipHostFree(idx->h_pin)); idx->h_pin = nullptr; idx->h_pin_cap = 0; }
            HIP_TRY(hipHostMalloc(&idx->h_pin, kMappedAppendMax + 256, hipHostMallocDefault));
            HIP_TRY(hipHostGetDevicePointer(&idx->h_pin_dev, idx->h_pin, 0));
            idx->h_pin_cap = kMappedAppendMax + 256;
        }
        char* h = (char*)idx->h_pin;
        char* d = (char*)idx->h_pin_dev;
        memset(h, 0, 8);
        memcpy(h + 256, rows, bytes);
        HIP_TRY(cmr_launch_convert_rows(idx->dtype, (const float*)(d + 256), n, idx->dim, idx->dpad, idx->n, idx->corpus, idx->shadow, (int*)d, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        int flagged = 0;
        memcpy(&flagged, h, sizeof(int));
        if (flagged) return fail(CMR_ERR_NONFINITE, "appended rows contain NaN/Inf (index unchanged)");
        idx->n += n;
        return CMR_OK;
    }
    // stage in chunks of <= 256 MiB of fp32
    const long long chunk_rows = std::max<long long>(1, (256ll << 20) / ((long long)idx->dim * 4));
    const long long n0 = idx->n;
    for (long long r0 = 0; r0 < n; r0 += chunk_rows) {
        const long long nr = std::min<long long>(chunk_rows, n - r0);
        const size_t bytes = (size_t)nr * idx->dim * 4;
        hipError_t e = idx->stage.ensure(bytes);
        if (e != hipSuccess) { idx->n = n0; return fail(CMR_ERR_OOM, "append staging: %s", hipGetErrorString(e)); }
        e = hipMemcpyAsync(idx->stage.p, rows + (size_t)r0 * idx->dim, bytes, hipMemcpyHostToDevice, nullptr);
        if (e != hipSuccess) { idx->n = n0; return fail(CMR_ERR_HIP, "H2D rows: %s", hipGetErrorString(e)); }
        rc = append_from_device(idx, (const float*)idx->stage.p, nr, nullptr);
        if (rc) { idx->n = n0; return rc; }
    }
    return CMR_OK;
}

int32_t cmr_index_append_dev(cmr_index_t* idx, const float* rows_dev, int64_t n, void* stream) {
    if (!idx || (n > 0 && !rows_dev)) return fail(CMR_ERR_INVALID, "NULL argument");
    if (n <= 0) return n == 0 ? CMR_OK : fail(CMR_ERR_INVALID, "n < 0");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (idx->n + n >= 0xFFFFFFF0ll) return fail(CMR_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
    if (global_id_overflow(idx, n)) return fail(CMR_ERR_UNSUPPORTED, "appending %lld rows takes this shard's global ids beyond the 32-bit row of the packed candidate exchange", (long long)n);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(s));  // rows_dev producer done before a possible grow() reallocates
    rc = grow(idx, (idx->n + n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS);
    if (rc) return rc;
    return append_from_device(idx, rows_dev, n, s);
}

int32_t cmr_index_search_dev(cmr_index_t* idx, const float* q_dev, int32_t nq, int32_t k, int64_t* ids_dev, float* scores_dev,
                             float* min_dev, float* max_dev, void* stream) {
    if (!idx || !q_dev || !ids_dev || !scores_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K_2PASS) return fail(CMR_ERR_UNSUPPORTED, "k = %d outside [1, %d]", k, CMR_MAX_K_2PASS);
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, (hipStream_t)stream, true);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    return search_enqueue(idx, ws, q_dev, nq, k, ids_dev, scores_dev, min_dev, max_dev);
}

int32_t cmr_index_search_min_score_dev(cmr_index_t* idx, const float* q_dev, int32_t nq, int32_t k, float min_score, int64_t* ids_dev,
                                       float* scores_dev, void* stream) {
    if (!idx || !q_dev || !ids_dev || !scores_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K) return fail(CMR_ERR_UNSUPPORTED, "threshold search supports k in [1, %d]", CMR_MAX_K);
    if (!(min_score == min_score)) return fail(CMR_ERR_INVALID, "min_score is NaN");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, (hipStream_t)stream, true);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    return search_enqueue(idx, ws, q_dev, nq, k, ids_dev, scores_dev, nullptr, nullptr, &min_score);
}

int32_t cmr_index_search_pipelined(cmr_index_t* idx, const float* q_dev, int32_t nq, int32_t k, int64_t* ids_dev, float* scores_dev,
                                   float* min_dev, float* max_dev, void* wait_event, void** done_event) {
    if (!idx || !q_dev || !ids_dev || !scores_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K) return fail(CMR_ERR_UNSUPPORTED, "k = %d outside [1, %d]", k, CMR_MAX_K);
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    hipEvent_t done = nullptr;
    rc = search_pipelined_enqueue(idx, q_dev, nq, k, ids_dev, scores_dev, min_dev, max_dev, (hipEvent_t)wait_event, &done);
    if (done_event) *done_event = (void*)done;
    return rc;
}

int32_t cmr_index_search_min_score_pipelined(cmr_index_t* idx, const float* q_dev, int32_t nq, int32_t k, float min_score, int64_t* ids_dev,
                                             float* scores_dev, void* wait_event, void** done_event) {
    if (!idx || !q_dev || !ids_dev || !scores_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K) return fail(CMR_ERR_UNSUPPORTED, "threshold search supports k in [1, %d]", CMR_MAX_K);
    if (!(min_score == min_score)) return fail(CMR_ERR_INVALID, "min_score is NaN");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    hipEvent_t done = nullptr;
    rc = search_pipelined_enqueue(idx, q_dev, nq, k, ids_dev, scores_dev, nullptr, nullptr, (hipEvent_t)wait_event, &done, &min_score);
    if (done_event) *done_event = (void*)done;
    return rc;
}

// the packed candidate exchange (comm.hip) carries the global row in 32 bits: 0xFFFFFFFF - row, key 0 = empty
static const long long kMaxGlobalId = 0xFFFFFFFEll;

int32_t cmr_index_set_id_base(cmr_index_t* idx, int64_t base) {
    if (!idx) return fail(CMR_ERR_INVALID, "NULL index");
    if (base < 0) return fail(CMR_ERR_INVALID, "id base %lld < 0", (long long)base);
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    if (base + idx->n - 1 > kMaxGlobalId)
        return fail(CMR_ERR_UNSUPPORTED, "id base %lld + %lld rows exceeds the 32-bit global row of the packed candidate exchange", (long long)base, idx->n);
    idx->id_base = base;
    idx->blk_local.clear(); idx->blk_global.clear();
    return CMR_OK;
}

int32_t cmr_index_set_id_blocks(cmr_index_t* idx, int32_t n_blocks, const int64_t* local_start, const int64_t* global_start) {
    if (!idx || n_blocks <= 0 || !local_start || !global_start) return fail(CMR_ERR_INVALID, "bad argument");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    if (local_start[0] != 0) return fail(CMR_ERR_INVALID, "the first block must start at local row 0");
    for (int b = 0; b < n_blocks; ++b) {
        const long long len = (b + 1 < n_blocks ? local_start[b + 1] : std::max<long long>(idx->n, local_start[b])) - local_start[b];
        if (global_start[b] < 0 || len < 0 || (b > 0 && (local_start[b] <= local_start[b - 1] || global_start[b] < global_start[b - 1] + (local_start[b] - local_start[b - 1]))))
            return fail(CMR_ERR_INVALID, "block %d: local starts must ascend and the global id runs must ascend without overlap", b);
        if (global_start[b] + len - 1 > kMaxGlobalId)
            return fail(CMR_ERR_UNSUPPORTED, "block %d reaches global id %lld: the packed candidate exchange carries 32-bit rows", b, (long long)(global_start[b] + len - 1));
    }
    if (n_blocks == 1) {
        idx->id_base = global_start[0];
        idx->blk_local.clear(); idx->blk_global.clear();
        return CMR_OK;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    std::vector<long long> tab((size_t)2 * n_blocks);
    for (int b = 0; b < n_blocks; ++b) { tab[b] = local_start[b]; tab[n_blocks + b] = global_start[b]; }
    long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, tab.size() * 8));
    HIP_TRY(hipMemcpy(d, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    if (idx->d_blk) idx->blk_retired.push_back(idx->d_blk);    // searches already enqueued keep reading the table they were given
    if (idx->blk_retired.size() >= 8) {                        // ... until the device has drained: then the old tables go
        HIP_TRY(hipDeviceSynchronize());
        for (void* p : idx->blk_retired) (void)hipFree(p);
        idx->blk_retired.clear();
    }
    idx->d_blk = d;
    idx->blk_local.assign(local_start, local_start + n_blocks);
    idx->blk_global.assign(global_start, global_start + n_blocks);
    idx->id_base = global_start[0];
    return CMR_OK;
}

int32_t cmr_index_set_option(cmr_index_t* idx, const char* name, int64_t value) {
    if (!idx || !name) return fail(CMR_ERR_INVALID, "NULL argument");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    return set_option(idx, name, value);
}

int32_t cmr_index_get_option(cmr_index_t* idx, const char* name, int64_t* value) {
    if (!idx || !name || !value) return fail(CMR_ERR_INVALID, "NULL argument");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const std::string n(name);
    if (n == "pipe_dual_scan_active") *value = idx->dual_active;
    else if (n == "pipe_dual_scan_wide_active") *value = idx->dual_wide_active;
    else if (n == "pipe_cu_mask_active") *value = idx->pipe.last_masked;
    else if (n == "pipe_scan_cus") *value = idx->pipe.last_masked ? idx->pipe.scan_cus : idx->n_cu;
    else return fail(CMR_ERR_INVALID, "unknown readable option '%s'", name);
    return CMR_OK;
}

int32_t cmr_index_pipeline_stream(cmr_index_t* idx, int32_t which, void** stream) {
    if (!idx || !stream) return fail(CMR_ERR_INVALID, "NULL argument");
    if (which < 0 || which > 2) return fail(CMR_ERR_INVALID, "which must be 0 (pre), 1 (scan) or 2 (post)");
    int rc = set_device(idx->device);
    if (rc) return rc;
    std::lock_guard<std::mutex> pl(idx->pipe_mu);
    Pipe& P = idx->pipe;
    rc = ensure_pipe(idx);
    if (rc) return rc;
    const bool twins = P.scan_cus != 0 && !P.last_masked && P.next != 0;      // the set the last narrow batch ran on
    *stream = which == 0 ? (void*)(twins ? P.usp : P.sp) : which == 1 ? (void*)(twins ? P.usm : P.sm) : (void*)P.sq;
    return CMR_OK;
}

int32_t cmr_index_query_status(cmr_index_t* idx, int32_t* nonfinite) {
    if (!idx || !nonfinite) return fail(CMR_ERR_INVALID, "NULL argument");
    *nonfinite = 0;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    std::vector<Workspace*> wss;
    {
        std::lock_guard<std::mutex> pl(idx->pipe_mu);
        for (int i = 0; i < CMR_PIPE_SLOTS; ++i) if (idx->pipe.slot[i].used) wss.push_back(&idx->pipe.slot[i].ws);
        if (idx->pipe.sq) {
            for (hipStream_t st : {idx->pipe.sp, idx->pipe.sm, idx->pipe.sm2, idx->pipe.wp, idx->pipe.wm, idx->pipe.wm2, idx->pipe.usp, idx->pipe.usm, idx->pipe.uwp, idx->pipe.uwm, idx->pipe.sq})
                if (st) HIP_TRY(hipStreamSynchronize(st));
        }
    }
    {
        std::lock_guard<std::mutex> g(idx->ws_mu);
        for (auto& kv : idx->stream_ws) wss.push_back(kv.second);
    }
    for (Workspace* ws : wss) {
        if (!ws->flag_ptr) continue;
        if (ws->stream || !ws->own_stream) HIP_TRY(hipStreamSynchronize(ws->stream));
        int h = 0;
        HIP_TRY(hipMemcpy(&h, ws->flag_ptr, sizeof(int), hipMemcpyDeviceToHost));
        if (h) {
            *nonfinite = 1;
            HIP_TRY(hipMemset(ws->flag_ptr, 0, sizeof(int)));
        }
    }
    return CMR_OK;
}

int32_t cmr_stream_wait_event(void* stream, void* event) {
    if (!event) return CMR_OK;
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return CMR_OK;
}

int32_t cmr_event_synchronize(void* event) {
    if (!event) return CMR_OK;
    HIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return CMR_OK;
}

}  // extern "C"

// ---- synchronous host-buffer search in two halves (cmr_internal.h): `begin` enqueues everything on the workspace's stream and
// returns without waiting, `finish` waits, checks the non-finite flag and copies the results out.  cmr_index_search is begin +
// finish; the multi-device index (multi.hip) begins on every shard before it finishes on any, so all devices scan at once.
struct CmrPending {
    cmr_index* idx = nullptr;
    Workspace* ws = nullptr;
    bool locked = false;          // holds idx->mu shared (released by finish / abandon ON THE SAME THREAD)
    bool mapped = false;          // results land in the pinned buffer by themselves (zero-copy) / by the enqueued D2H copy
    bool poll = false;            // the search's last kernel sets the done word of the pinned buffer (Workspace::done_ptr): finish polls it
    int nq = 0, k = 0;
    size_t o_ids = 0, o_sc = 0, o_min = 0, o_max = 0;
};

static void pending_release(CmrPending* p) {
    if (p->ws) release_ws(p->idx, p->ws);
    if (p->locked) p->idx->mu.unlock_shared();
    delete p;
}

int cmr_index_search_begin(cmr_index_t* idx, const float* q, int nq, int k, const float* min_score, bool take_lock, CmrPending** out) {
    if (!idx || !q || !out) return fail(CMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K_2PASS) return fail(CMR_ERR_UNSUPPORTED, "k = %d outside [1, %d]", k, CMR_MAX_K_2PASS);
    CmrPending* P = new CmrPending();
    P->idx = idx; P->nq = nq; P->k = k;
    if (take_lock) { idx->mu.lock_shared(); P->locked = true; }
    int rc = set_device(idx->device);
    if (rc) { pending_release(P); return rc; }
    Workspace* ws = acquire_ws(idx, nullptr, false);
    if (!ws) { pending_release(P); return fail(CMR_ERR_HIP, "could not create a workspace stream"); }
    P->ws = ws;
    hipStream_t s = ws->stream;
    // packed result buffer [flag (8 B) | ids nq*k i64 | scores nq*k f32 | min nq | max nq]
    const size_t o_ids = 8, o_sc = o_ids + (size_t)nq * k * 8, o_min = o_sc + (size_t)nq * k * 4, o_max = o_min + (size_t)nq * 4,
                 out_bytes = o_max + (size_t)nq * 4, q_bytes = (size_t)nq * idx->dim * 4;
    P->o_ids = o_ids; P->o_sc = o_sc; P->o_min = o_min; P->o_max = o_max;
    auto body = [&]() -> int {
        if (idx->zero_copy && k <= CMR_MAX_K && q_bytes <= kZeroCopyMax && out_bytes <= kZeroCopyMax) {
            // (the hierarchical single-launch path packs the queries in up to 64 workgroups: they read a device copy, not
            // 64 times across the link)
            const bool map_in = small_path_kind(idx, nq, k, min_score != nullptr) != 2;
            // Small calls (what ComoRAG issues: one query, a few hundred rows) are all latency.  A copy each way costs two more
            // submissions in front of / behind the kernels (36 us per call at 6 rows, of which the search itself is ~8); so
            // there are none: queries, results and the non-finite flag live in ONE pinned, device-mapped host buffer
            // (fine-grained: kernel stores are visible once the stream has been synchronised) that the kernels read and
            // write over PCIe themselves — a few KiB either way.
            const size_t o_q = (out_bytes + 255) & ~(size_t)255;
            HIP_TRY(ws->ensure_pin(o_q + q_bytes));
            char* h = (char*)ws->h_pin;
            char* d = (char*)ws->h_pin_dev;
            memcpy(h + o_q, q, q_bytes);
            memset(h, 0, 8);
            const float* q_in = (const float*)(d + o_q);
            if (!map_in) {
                HIP_TRY(ws->d_q.ensure(q_bytes));
                HIP_TRY(hipMemcpyAsync(ws->d_q.p, h + o_q, q_bytes, hipMemcpyHostToDevice, s));
                q_in = (const float*)ws->d_q.p;
            }
            int* const dev_flag = ws->flag_ptr;
            ws->flag_ptr = (int*)d;
            ws->done_ptr = idx->sync_poll ? (int*)(d + 4) : nullptr; ws->done_used = false; ws->lazy.due = false;      // (bytes 4..7 of the header were zeroed above)
            const int rc_ = search_enqueue(idx, ws, q_in, nq, k, (int64_t*)(d + o_ids), (float*)(d + o_sc), (float*)(d + o_min), (float*)(d + o_max), min_score);
            ws->flag_ptr = dev_flag;
            ws->done_ptr = nullptr;
            P->mapped = true;
            P->poll = rc_ == CMR_OK && ws->done_used;
            return rc_;
        }
        // packed device buffer and its pinned host twin, one copy each way; the queries go through the pinned buffer too (a
        // pageable H2D is staged by the runtime anyway)
        HIP_TRY(ws->d_q.ensure(q_bytes));
        if (out_bytes > ws->d_pack.cap || !ws->d_pack.p) {        // (re)allocation moves the flag: re-arm it at the new place
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(ws->d_pack.ensure(std::max<size_t>(out_bytes, 4096)));
            HIP_TRY(hipMemsetAsync(ws->d_pack.p, 0, 8, s));
        }
        int* const dev_flag = ws->flag_ptr;
        ws->flag_ptr = (int*)ws->d_pack.p;
        hipError_t e = ws->ensure_pin(std::max(out_bytes, q_bytes));
        if (e != hipSuccess) { ws->flag_ptr = dev_flag; HIP_TRY(e); }
        memcpy(ws->h_pin, q, q_bytes);
        e = hipMemcpyAsync(ws->d_q.p, ws->h_pin, q_bytes, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { ws->flag_ptr = dev_flag; HIP_TRY(e); }
        char* pk = (char*)ws->d_pack.p;
        const int rc_ = search_enqueue(idx, ws, (const float*)ws->d_q.p, nq, k, (int64_t*)(pk + o_ids), (float*)(pk + o_sc), (float*)(pk + o_min), (float*)(pk + o_max), min_score);
        ws->flag_ptr = dev_flag;
        if (rc_) return rc_;
        // (the pinned buffer still holds the queries the H2D copy reads: stream order puts the D2H copy behind it)
        HIP_TRY(hipMemcpyAsync(ws->h_pin, pk, out_bytes, hipMemcpyDeviceToHost, s));
        return CMR_OK;
    };
    rc = body();
    if (rc) { (void)hipStreamSynchronize(s); pending_release(P); return rc; }
    *out = P;
    return CMR_OK;
}

int cmr_index_search_finish(CmrPending* P, int64_t* out_ids, float* out_scores, float* out_min, float* out_max) {
    if (!P) return fail(CMR_ERR_INVALID, "NULL pending search");
    struct Rel { CmrPending* p; ~Rel() { pending_release(p); } } rel{P};
    int rc = set_device(P->idx->device);
    if (rc) return rc;
    Workspace* ws = P->ws;
    if (P->poll) {
        int st = 0;
        { int rc_ = wait_done_word(ws, &st); if (rc_) return rc_; }
        if (ws->lazy.due && st != 1) {      // a dense list or a staging area of the finishing stage overflowed: the merge works from the per-wave lists
            const Workspace::LazyMerge& L = ws->lazy;
            ws->lazy.due = false;
            HIP_TRY(cmr_launch_merge_query(L.lists, L.cnt, L.W, L.NQ, L.cap, L.nqp, L.k, L.mm, L.id_base, L.ids, L.scores, L.mn, L.mx, nullptr, ws->stream, false, L.state));
            HIP_TRY(hipStreamSynchronize(ws->stream));
        }
        ws->lazy.due = false;
    } else {
        HIP_TRY(hipStreamSynchronize(ws->stream));
    }
    const char* hp = (const char*)ws->h_pin;
    int flagged = 0;
    memcpy(&flagged, hp, sizeof(int));
    if (flagged) {
        if (!P->mapped) HIP_TRY(hipMemsetAsync(ws->d_pack.p, 0, sizeof(int), ws->stream));
        return fail(CMR_ERR_NONFINITE, "query contains NaN/Inf");
    }
    const size_t nk = (size_t)P->nq * P->k;
    if (out_ids) memcpy(out_ids, hp + P->o_ids, nk * 8);
    if (out_scores) memcpy(out_scores, hp + P->o_sc, nk * 4);
    if (out_min) memcpy(out_min, hp + P->o_min, (size_t)P->nq * 4);
    if (out_max) memcpy(out_max, hp + P->o_max, (size_t)P->nq * 4);
    return CMR_OK;
}

void cmr_index_search_abandon(CmrPending* P) {
    if (!P) return;
    (void)hipSetDevice(P->idx->device);
    (void)hipStreamSynchronize(P->ws->stream);
    P->ws->lazy.due = false;
    // a query flagged non-finite leaves its mark in the packed DEVICE buffer of the copy path: clear it for the next call
    if (!P->mapped && P->ws->d_pack.p) (void)hipMemsetAsync(P->ws->d_pack.p, 0, sizeof(int), P->ws->stream);
    pending_release(P);
}

// roll a shard back to n_rows (multi.hip: an append that failed on a later shard).  Slots beyond n_rows keep stale data:
// every kernel masks by row index, and the next append rewrites them.
int cmr_index_truncate(cmr_index_t* idx, long long n_rows) {
    if (!idx) return fail(CMR_ERR_INVALID, "NULL index");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    if (n_rows < 0 || n_rows > idx->n) return fail(CMR_ERR_INVALID, "truncate to %lld rows of %lld", n_rows, idx->n);
    idx->n = n_rows;
    return CMR_OK;
}

extern "C" {

static int32_t host_search(cmr_index_t* idx, const float* q, int32_t nq, int32_t k, int64_t* out_ids, float* out_scores,
                           float* out_min, float* out_max, const float* min_score) {
    if (!idx || !q || !out_ids || !out_scores) return fail(CMR_ERR_INVALID, "NULL argument");
    CmrPending* p = nullptr;
    int rc = cmr_index_search_begin(idx, q, nq, k, min_score, true, &p);
    if (rc) return rc;
    return cmr_index_search_finish(p, out_ids, out_scores, out_min, out_max);
}

int32_t cmr_index_search(cmr_index_t* idx, const float* q, int32_t nq, int32_t k, int64_t* out_ids, float* out_scores,
                         float* out_min, float* out_max) {
    return host_search(idx, q, nq, k, out_ids, out_scores, out_min, out_max, nullptr);
}

int32_t cmr_index_search_min_score(cmr_index_t* idx, const float* q, int32_t nq, int32_t k, float min_score, int64_t* out_ids,
                                   float* out_scores) {
    if (!(min_score == min_score)) return fail(CMR_ERR_INVALID, "min_score is NaN");
    return host_search(idx, q, nq, k, out_ids, out_scores, nullptr, nullptr, &min_score);
}

int32_t cmr_index_scores_dev(cmr_index_t* idx, const float* q_dev, int32_t nq, float* out_dev, int64_t ld, void* stream) {
    if (!idx || !q_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (ld == 0) ld = idx->n;
    if (ld < idx->n) return fail(CMR_ERR_INVALID, "ld %lld < rows %lld", (long long)ld, idx->n);
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, (hipStream_t)stream, true);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    if (idx->n == 0) return CMR_OK;
    return scores_enqueue(idx, ws, q_dev, nq, out_dev, ld);
}

int32_t cmr_index_scores(cmr_index_t* idx, const float* q, int32_t nq, float* out, int64_t ld) {
    if (!idx || !q || !out) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (ld == 0) ld = idx->n;
    if (ld < idx->n) return fail(CMR_ERR_INVALID, "ld %lld < rows %lld", (long long)ld, idx->n);
    if (idx->n == 0) return CMR_OK;
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, nullptr, false);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    struct Rel { cmr_index* i; Workspace* w; ~Rel() { release_ws(i, w); } } rel{idx, ws};
    hipStream_t s = ws->stream;
    {   // Small corpus, few queries (dense_passage_retrieval / get_fact_scores on a few thousand rows, one query per call):
        // ONE launch packs, scans and writes the scores straight into a pinned, device-mapped host buffer — no pack
        // launch, no copies, one synchronisation (the general path: pageable H2D, pack, scan, 2-D D2H, flag D2H, two syncs).
        const long long npanels = (idx->n + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS;
        const int ks = idx->dtype == CMR_F32 ? idx->dpad / 8 : idx->dpad / 16;
        const size_t sc_bytes = (size_t)nq * idx->n * 4, q_bytes = (size_t)nq * idx->dim * 4;
        if (idx->zero_copy && !idx->no_tiny && !idx->no_small && nq <= 16 && npanels <= idx->small_max_panels && (size_t)ks * 1024 <= 143 * 1024 &&
            sc_bytes <= 4 * kZeroCopyMax) {
            const size_t o_sc = 256, o_q = (o_sc + sc_bytes + 255) & ~(size_t)255;
            HIP_TRY(ws->ensure_pin(o_q + q_bytes));
            char* h = (char*)ws->h_pin;
            char* d = (char*)ws->h_pin_dev;
            memcpy(h + o_q, q, q_bytes);
            memset(h, 0, 8);
            const float* q_in = (const float*)(d + o_q);
            if (npanels > 32) {               // many workgroups pack the queries: from a device copy, not across the link once each
                HIP_TRY(ws->d_q.ensure(q_bytes));
                HIP_TRY(hipMemcpyAsync(ws->d_q.p, h + o_q, q_bytes, hipMemcpyHostToDevice, s));
                q_in = (const float*)ws->d_q.p;
            }
            HIP_TRY(ws->d_out.ensure((size_t)nq * npanels * CMR_PANEL_ROWS * 4));
            if (idx->sync_poll && !ws->arrive.p) {          // arrival counter: zeroed once, re-armed by the kernel
                HIP_TRY(ws->arrive.ensure(sizeof(int)));
                HIP_TRY(hipMemsetAsync(ws->arrive.p, 0, sizeof(int), s));
            }
            HIP_TRY(cmr_launch_tiny_scores(idx->dtype, idx->corpus, q_in, nq, idx->dim, idx->dpad, idx->n, ws->d_out.p, (float*)(d + o_sc), idx->n,
                                           (int*)d, s, idx->sync_poll ? (int*)ws->arrive.p : nullptr, idx->sync_poll ? (int*)(d + 4) : nullptr));
            if (idx->sync_poll) { int rc_ = wait_done_word(ws); if (rc_) return rc_; }      // the last workgroup's word behind everybody's rows (bytes 4..7, zeroed above)
            else HIP_TRY(hipStreamSynchronize(s));
            int flagged = 0;
            memcpy(&flagged, h, sizeof(int));
            if (flagged) return fail(CMR_ERR_NONFINITE, "query contains NaN/Inf");
            for (int qi = 0; qi < nq; ++qi) memcpy(out + (size_t)qi * ld, h + o_sc + (size_t)qi * idx->n * 4, (size_t)idx->n * 4);
            return CMR_OK;
        }
    }
    HIP_TRY(ws->d_q.ensure((size_t)nq * idx->dim * 4));
    HIP_TRY(ws->d_out.ensure((size_t)nq * idx->n * 4));
    HIP_TRY(hipMemcpyAsync(ws->d_q.p, q, (size_t)nq * idx->dim * 4, hipMemcpyHostToDevice, s));
    rc = scores_enqueue(idx, ws, (const float*)ws->d_q.p, nq, (float*)ws->d_out.p, idx->n);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * 4, ws->d_out.p, (size_t)idx->n * 4, (size_t)idx->n * 4, (size_t)nq,
                             hipMemcpyDeviceToHost, s));
    return check_query_flag(ws);
}

int32_t cmr_index_sorted_scores(cmr_index_t* idx, const float* q, int32_t nq, int64_t* out_ids, float* out_scores, float* out_min,
                                float* out_max) {
    if (!idx || !q || !out_ids || !out_scores) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return fail(CMR_ERR_INVALID, "nq must be > 0");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const long long n = idx->n;
    if (n == 0) return CMR_OK;
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, nullptr, false);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    struct Rel { cmr_index* i; Workspace* w; ~Rel() { release_ws(i, w); } } rel{idx, ws};
    hipStream_t s = ws->stream;
    HIP_TRY(ws->d_q.ensure((size_t)nq * idx->dim * 4));
    HIP_TRY(ws->d_out.ensure((size_t)n * 4));
    HIP_TRY(ws->d_cand.ensure(cmr_sort_workspace_bytes(n)));
    HIP_TRY(hipMemcpyAsync(ws->d_q.p, q, (size_t)nq * idx->dim * 4, hipMemcpyHostToDevice, s));
    // Several queries: scan + sort of query i + 1 run while the 12 N bytes of query i cross the link on a second stream
    // (two result sets, two event pairs; one query: no second stream, no events).  Nothing synchronises per query.
    const int nset = nq > 1 ? 2 : 1;
    Workspace* wc = nset > 1 ? acquire_ws(idx, nullptr, false) : nullptr;      // its stream carries the copies
    if (nset > 1 && !wc) return fail(CMR_ERR_HIP, "could not create a copy stream");
    struct Rel2 { cmr_index* i; Workspace* w; ~Rel2() { if (w) release_ws(i, w); } } rel2{idx, wc};
    DevBuf* ids_buf[2] = {&ws->d_ids, wc ? &wc->d_ids : nullptr};
    DevBuf* sc_buf[2] = {&ws->d_scores, wc ? &wc->d_scores : nullptr};
    hipEvent_t sorted[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    struct Ev { hipEvent_t* a; hipEvent_t* b; ~Ev() { for (int i = 0; i < 2; ++i) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); } } } evs{sorted, copied};
    for (int i = 0; i < nset; ++i) {
        HIP_TRY(ids_buf[i]->ensure((size_t)n * 8));
        HIP_TRY(sc_buf[i]->ensure((size_t)n * 4));
        if (nset > 1) {
            HIP_TRY(hipEventCreateWithFlags(&sorted[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
        }
    }
    hipStream_t sc = wc ? wc->stream : s;
    auto copy_out = [&](int qi) -> int {         // a pageable destination may hold the calling thread until the copy is done:
        const int b = qi % nset;                  // issued only AFTER the next query's scan + sort are in the queue
        if (nset > 1) HIP_TRY(hipStreamWaitEvent(sc, sorted[b], 0));
        HIP_TRY(hipMemcpyAsync(out_ids + (size_t)qi * n, ids_buf[b]->p, (size_t)n * 8, hipMemcpyDeviceToHost, sc));
        HIP_TRY(hipMemcpyAsync(out_scores + (size_t)qi * n, sc_buf[b]->p, (size_t)n * 4, hipMemcpyDeviceToHost, sc));
        if (nset > 1) HIP_TRY(hipEventRecord(copied[b], sc));
        return CMR_OK;
    };
    auto body = [&]() -> int {
        for (int qi = 0; qi < nq; ++qi) {
            const int b = qi % nset;
            if (nset > 1 && qi >= nset) HIP_TRY(hipStreamWaitEvent(s, copied[b], 0));      // the set's previous contents are on the host
            int rc_ = scores_enqueue(idx, ws, (const float*)ws->d_q.p + (size_t)qi * idx->dim, 1, (float*)ws->d_out.p, n);
            if (rc_) return rc_;
            HIP_TRY(cmr_launch_sort_scores((const float*)ws->d_out.p, n, kernel_id_base(idx), ws->d_cand.p, (int64_t*)ids_buf[b]->p, (float*)sc_buf[b]->p, s));
            rc_ = remap_ids_enqueue(idx, (int64_t*)ids_buf[b]->p, n, s);
            if (rc_) return rc_;
            if (nset > 1) HIP_TRY(hipEventRecord(sorted[b], s));
            if (qi > 0) { rc_ = copy_out(qi - 1); if (rc_) return rc_; }
        }
        return copy_out(nq - 1);
    };
    rc = body();
    const hipError_t e1 = hipStreamSynchronize(s);
    const hipError_t e2 = sc != s ? hipStreamSynchronize(sc) : hipSuccess;
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(CMR_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    for (int qi = 0; qi < nq; ++qi) {
        if (out_max) out_max[qi] = out_scores[(size_t)qi * n];
        if (out_min) out_min[qi] = out_scores[(size_t)qi * n + n - 1];
    }
    return check_query_flag(ws);
}

int32_t cmr_index_rescore(cmr_index_t* idx, const float* q, int32_t nq, const int64_t* cand, int32_t n_cand, int32_t k,
                          int64_t* out_ids, float* out_scores) {
    if (!idx || !q || !cand || !out_ids || !out_scores) return fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0 || n_cand <= 0 || k <= 0) return fail(CMR_ERR_INVALID, "nq, n_cand, k must be > 0");
    if (n_cand > 4096) return fail(CMR_ERR_UNSUPPORTED, "n_cand %d > 4096", n_cand);
    if (k > n_cand) k = n_cand;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, nullptr, false);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    struct Rel { cmr_index* i; Workspace* w; ~Rel() { release_ws(i, w); } } rel{idx, ws};
    hipStream_t s = ws->stream;
    // a shard with a block table: candidates come in as global ids, the kernel works on local rows (base 0), its output
    // ids are translated back on the stream
    std::vector<int64_t> local_cand;
    const bool blocks = idx->blk_local.size() > 1;
    if (blocks) {
        local_cand.resize((size_t)nq * n_cand);
        for (size_t i = 0; i < local_cand.size(); ++i) local_cand[i] = to_local_row(idx, cand[i]);
        cand = local_cand.data();
    }
    const long long base = kernel_id_base(idx);
    {   // few queries (the exact re-scorer behind a top-100 search): candidates + queries go down in ONE copy from the pinned
        // buffer (the kernel re-reads each query once per candidate: not across the link), the results are written straight
        // into its mapped half — one copy, one launch, one sync instead of two copies each way
        const size_t q_bytes = (size_t)nq * idx->dim * 4, c_bytes = ((size_t)nq * n_cand * 8 + 255) & ~(size_t)255, i_bytes = (size_t)nq * k * 8,
                     s_bytes = (size_t)nq * k * 4;
        if (idx->zero_copy && i_bytes + s_bytes <= kZeroCopyMax && c_bytes + q_bytes <= 16 * kZeroCopyMax) {
            const size_t o_ids = 0, o_sc = o_ids + i_bytes, o_in = (o_sc + s_bytes + 255) & ~(size_t)255;
            HIP_TRY(ws->ensure_pin(o_in + c_bytes + q_bytes));
            HIP_TRY(ws->d_cand.ensure(c_bytes + q_bytes));
            char* h = (char*)ws->h_pin;
            char* d = (char*)ws->h_pin_dev;
            memcpy(h + o_in, cand, (size_t)nq * n_cand * 8);
            memcpy(h + o_in + c_bytes, q, q_bytes);
            HIP_TRY(hipMemcpyAsync(ws->d_cand.p, h + o_in, c_bytes + q_bytes, hipMemcpyHostToDevice, s));
            HIP_TRY(cmr_launch_rescore(idx->dtype, idx->corpus, idx->shadow, idx->dim, idx->dpad, idx->n, base,
                                       (const float*)((const char*)ws->d_cand.p + c_bytes), nq, (const int64_t*)ws->d_cand.p, n_cand, k,
                                       (int64_t*)(d + o_ids), (float*)(d + o_sc), s));
            { int rc_ = remap_ids_enqueue(idx, (int64_t*)(d + o_ids), (long long)nq * k, s); if (rc_) { (void)hipStreamSynchronize(s); return rc_; } }
            HIP_TRY(hipStreamSynchronize(s));
            memcpy(out_ids, h + o_ids, i_bytes);
            memcpy(out_scores, h + o_sc, s_bytes);
            return CMR_OK;
        }
    }
    HIP_TRY(ws->d_q.ensure((size_t)nq * idx->dim * 4));
    HIP_TRY(ws->d_cand.ensure((size_t)nq * n_cand * 8));
    HIP_TRY(ws->d_ids.ensure((size_t)nq * k * 8));
    HIP_TRY(ws->d_scores.ensure((size_t)nq * k * 4));
    HIP_TRY(hipMemcpyAsync(ws->d_q.p, q, (size_t)nq * idx->dim * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ws->d_cand.p, cand, (size_t)nq * n_cand * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(cmr_launch_rescore(idx->dtype, idx->corpus, idx->shadow, idx->dim, idx->dpad, idx->n, base, (const float*)ws->d_q.p, nq,
                               (const int64_t*)ws->d_cand.p, n_cand, k, (int64_t*)ws->d_ids.p, (float*)ws->d_scores.p, s));
    { int rc_ = remap_ids_enqueue(idx, (int64_t*)ws->d_ids.p, (long long)nq * k, s); if (rc_) { (void)hipStreamSynchronize(s); return rc_; } }
    HIP_TRY(hipMemcpyAsync(out_ids, ws->d_ids.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_scores, ws->d_scores.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return CMR_OK;
}

int32_t cmr_index_get_rows(cmr_index_t* idx, const int64_t* ids, int64_t n, float* out) {
    if (!idx || (n > 0 && (!ids || !out))) return fail(CMR_ERR_INVALID, "NULL argument");
    if (n <= 0) return CMR_OK;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    Workspace* ws = acquire_ws(idx, nullptr, false);
    if (!ws) return fail(CMR_ERR_HIP, "could not create a workspace stream");
    struct Rel { cmr_index* i; Workspace* w; ~Rel() { release_ws(i, w); } } rel{idx, ws};
    hipStream_t s = ws->stream;
    HIP_TRY(ws->d_cand.ensure((size_t)n * 8));
    HIP_TRY(ws->d_out.ensure((size_t)n * idx->dim * 4));
    std::vector<int64_t> local_ids;
    if (idx->blk_local.size() > 1) {      // global ids -> local rows through the block table (a row this shard does not hold: -1, as an id outside [base, base + n))
        local_ids.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) local_ids[i] = to_local_row(idx, ids[i]);
        ids = local_ids.data();
    }
    HIP_TRY(hipMemcpyAsync(ws->d_cand.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(cmr_launch_gather_rows(idx->dtype, idx->corpus, idx->dim, idx->dpad, idx->n, kernel_id_base(idx), (const int64_t*)ws->d_cand.p, n,
                                   (float*)ws->d_out.p, s));
    HIP_TRY(hipMemcpyAsync(out, ws->d_out.p, (size_t)n * idx->dim * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return CMR_OK;
}

int32_t cmr_merge_topk(const int64_t* ids, const float* scores, int32_t S, int32_t nq, int32_t k, int64_t* out_ids,
                       float* out_scores) {
    // Host-side final merge (north_star: "host-side final merge" after the RCCL all-gather).
    // Pure index/compare work on S*k <= a few hundred candidates per query; same tie rule.
    if (!ids || !scores || !out_ids || !out_scores) return fail(CMR_ERR_INVALID, "NULL argument");
    if (S <= 0 || nq <= 0 || k <= 0) return fail(CMR_ERR_INVALID, "S, nq, k must be > 0");
    std::vector<std::pair<float, int64_t>> v;
    v.reserve((size_t)S * k);
    for (int q = 0; q < nq; ++q) {
        v.clear();
        for (int s = 0; s < S; ++s)
            for (int j = 0; j < k; ++j) {
                const size_t o = ((size_t)s * nq + q) * k + j;
                if (ids[o] >= 0) v.emplace_back(scores[o] + 0.0f, ids[o]);
            }
        std::sort(v.begin(), v.end(), [](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
            return a.first > b.first || (a.first == b.first && a.second < b.second);
        });
        for (int j = 0; j < k; ++j) {
            const bool have = (size_t)j < v.size();
            out_ids[(size_t)q * k + j] = have ? v[j].second : -1;
            out_scores[(size_t)q * k + j] = have ? v[j].first : -INFINITY;
        }
    }
    return CMR_OK;
}

int32_t cmr_merge_topk_dev(int32_t device_id, const int64_t* ids_dev, const float* scores_dev, int32_t S, int32_t nq, int32_t k,
                           int64_t* out_ids_dev, float* out_scores_dev, void* stream) {
    if (!ids_dev || !scores_dev || !out_ids_dev || !out_scores_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (S <= 0 || nq <= 0 || k <= 0) return fail(CMR_ERR_INVALID, "S, nq, k must be > 0");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_merge_shards(ids_dev, scores_dev, S, nq, k, out_ids_dev, out_scores_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_pool_l2norm(int32_t device_id, const void* hidden_dev, int32_t hidden_dtype, const int64_t* mask_dev, int32_t b,
                        int32_t l, int32_t d, int32_t normalize, float* out_dev, void* stream) {
    if (!hidden_dev || !mask_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (b <= 0 || l <= 0 || d <= 0) return fail(CMR_ERR_INVALID, "b, l, d must be > 0");
    if (hidden_dtype != CMR_F32 && hidden_dtype != CMR_BF16 && hidden_dtype != CMR_F16) return fail(CMR_ERR_INVALID, "unknown dtype");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    const int splits = cmr_pool_splits(b, l, d);
    // partials live in a per-device, per-stream scratch that grows on demand
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, DevBuf> scratch;
    float* partial = nullptr;
    {
        std::lock_guard<std::mutex> g(mu);
        DevBuf& bf = scratch[{device_id, (hipStream_t)stream}];
        if (bf.cap < (size_t)b * splits * d * 4) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(bf.ensure((size_t)b * splits * d * 4));
        partial = (float*)bf.p;
    }
    HIP_TRY(cmr_launch_pool(hidden_dev, hidden_dtype, mask_dev, b, l, d, normalize, partial, out_dev, splits, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_encoder_attention(int32_t device_id, const void* qkv_dev, int32_t dtype, const int32_t* lens_dev, int32_t b, int32_t l,
                              int32_t n_heads, int32_t head_dim, void* out_dev, void* stream) {
    if (!qkv_dev || !lens_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (b <= 0 || l <= 0 || n_heads <= 0) return fail(CMR_ERR_INVALID, "b, l, n_heads must be > 0");
    if (head_dim != 64) return fail(CMR_ERR_INVALID, "cmr_encoder_attention: head_dim must be 64 (BERT-base / BERT-large heads)");
    if (dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "cmr_encoder_attention: dtype must be bf16 or f16");
    if (((uintptr_t)qkv_dev | (uintptr_t)out_dev) & 15) return fail(CMR_ERR_INVALID, "cmr_encoder_attention: buffers must be 16-byte aligned");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_attention(qkv_dev, dtype, lens_dev, b, l, n_heads, out_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_encoder_add_layernorm(int32_t device_id, const void* y_dev, const void* bias_dev, const void* residual_dev, const void* gamma_dev,
                                  const void* beta_dev, float eps, int64_t rows, int32_t d, int32_t dtype, void* out_dev, void* stream) {
    if (!y_dev || !gamma_dev || !beta_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (rows <= 0 || d <= 0 || d % 4 || d > 2048) return fail(CMR_ERR_INVALID, "cmr_encoder_add_layernorm: rows > 0, d a multiple of 4, d <= 2048");
    if (dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "cmr_encoder_add_layernorm: dtype must be bf16 or f16");
    if (((uintptr_t)y_dev | (uintptr_t)bias_dev | (uintptr_t)residual_dev | (uintptr_t)gamma_dev | (uintptr_t)beta_dev | (uintptr_t)out_dev) & 7)
        return fail(CMR_ERR_INVALID, "cmr_encoder_add_layernorm: buffers must be 8-byte aligned");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_add_layernorm(y_dev, bias_dev, residual_dev, gamma_dev, beta_dev, eps, rows, d, dtype, out_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_encoder_add_layernorm_pool(int32_t device_id, const void* y_dev, const void* bias_dev, const void* residual_dev, const void* gamma_dev,
                                       const void* beta_dev, float eps, int32_t b, int32_t l, int32_t d, int32_t dtype, const int32_t* lens_dev,
                                       int32_t normalize, float* partial_dev, float* out_dev, void* stream) {
    if (!y_dev || !gamma_dev || !beta_dev || !lens_dev || !partial_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (b <= 0 || l <= 0 || d <= 0) return fail(CMR_ERR_INVALID, "b, l, d must be > 0");
    if (l % 16 || d % 8 || d > 2048) return fail(CMR_ERR_UNSUPPORTED, "cmr_encoder_add_layernorm_pool: l must be a multiple of 16, d a multiple of 8 and <= 2048");
    if (dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "cmr_encoder_add_layernorm_pool: dtype must be bf16 or f16");
    if (((uintptr_t)y_dev | (uintptr_t)bias_dev | (uintptr_t)residual_dev | (uintptr_t)gamma_dev | (uintptr_t)beta_dev | (uintptr_t)partial_dev | (uintptr_t)out_dev) & 15)
        return fail(CMR_ERR_UNSUPPORTED, "cmr_encoder_add_layernorm_pool: buffers must be 16-byte aligned");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_add_layernorm_pool(y_dev, bias_dev, residual_dev, gamma_dev, beta_dev, eps, b, l, d, dtype, (const int*)lens_dev, normalize, partial_dev,
                                          out_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_encoder_embed_layernorm(int32_t device_id, const int64_t* ids_dev, const int64_t* token_type_dev, const void* word_dev, const void* pos_dev,
                                    const void* type_dev, const void* gamma_dev, const void* beta_dev, float eps, int64_t rows, int32_t l, int32_t d,
                                    int32_t vocab, int32_t n_positions, int32_t n_types, int32_t position_offset, int32_t dtype, void* out_dev, void* stream) {
    if (!ids_dev || !word_dev || !pos_dev || !type_dev || !gamma_dev || !beta_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (rows <= 0 || l <= 0 || d <= 0 || d % 4 || d > 2048) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm: rows, l > 0, d a multiple of 4, d <= 2048");
    if (vocab <= 0 || n_positions <= 0 || n_types <= 0 || position_offset < 0) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm: empty embedding table / negative position offset");
    if (dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm: dtype must be bf16 or f16");
    if (((uintptr_t)word_dev | (uintptr_t)pos_dev | (uintptr_t)type_dev | (uintptr_t)gamma_dev | (uintptr_t)beta_dev | (uintptr_t)out_dev) & 7)
        return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm: buffers must be 8-byte aligned");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_embed_layernorm((const long long*)ids_dev, (const long long*)token_type_dev, word_dev, pos_dev, type_dev, gamma_dev, beta_dev, eps,
                                       rows, l, d, vocab, n_positions, n_types, position_offset, dtype, out_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_encoder_embed_layernorm_ragged(int32_t device_id, const int32_t* ids32_dev, const int32_t* offsets_dev, const void* word_dev, const void* pos_dev,
                                           const void* type_dev, const void* gamma_dev, const void* beta_dev, float eps, int32_t b, int32_t l, int32_t d,
                                           int32_t vocab, int32_t n_positions, int32_t position_offset, int32_t dtype, void* out_dev, void* stream) {
    if (!ids32_dev || !offsets_dev || !word_dev || !pos_dev || !type_dev || !gamma_dev || !beta_dev || !out_dev) return fail(CMR_ERR_INVALID, "NULL argument");
    if (b <= 0 || l <= 0 || d <= 0 || d % 4 || d > 2048) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm_ragged: b, l > 0, d a multiple of 4, d <= 2048");
    if (vocab <= 0 || n_positions <= 0 || position_offset < 0) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm_ragged: empty embedding table / negative position offset");
    if (dtype != CMR_BF16 && dtype != CMR_F16) return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm_ragged: dtype must be bf16 or f16");
    if (((uintptr_t)word_dev | (uintptr_t)pos_dev | (uintptr_t)type_dev | (uintptr_t)gamma_dev | (uintptr_t)beta_dev | (uintptr_t)out_dev) & 7)
        return fail(CMR_ERR_INVALID, "cmr_encoder_embed_layernorm_ragged: buffers must be 8-byte aligned");
    int rc = check_device(device_id);
    if (rc) return rc;
    rc = set_device(device_id);
    if (rc) return rc;
    HIP_TRY(cmr_launch_embed_layernorm_ragged((const int*)ids32_dev, (const int*)offsets_dev, word_dev, pos_dev, type_dev, gamma_dev, beta_dev, eps, (long long)b * l, l,
                                              d, vocab, n_positions, position_offset, dtype, out_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_profile_enable(cmr_index_t* idx, int32_t on) {
    if (!idx) return fail(CMR_ERR_INVALID, "NULL index");
    std::lock_guard<std::mutex> g(idx->prof_mu);
    idx->prof_on = on != 0;
    idx->prof_every = on > 1 ? on : 1;
    idx->prof_seq = 0;
    return CMR_OK;
}

int32_t cmr_profile_collect(cmr_index_t* idx, int64_t* n_launches, double* total_ms, double* bytes_per_launch) {
    if (!idx) return fail(CMR_ERR_INVALID, "NULL index");
    int rc = set_device(idx->device);
    if (rc) return rc;
    std::vector<ProfEvent> ev;
    double bytes = 0;
    {
        std::lock_guard<std::mutex> g(idx->prof_mu);
        ev.swap(idx->prof_events);
        bytes = idx->prof_bytes;
    }
    double ms = 0;
    for (ProfEvent& pe : ev) {
        HIP_TRY(hipEventSynchronize(pe.b));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, pe.a, pe.b));
        ms += t;
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    if (n_launches) *n_launches = (int64_t)ev.size();
    if (total_ms) *total_ms = ms;
    if (bytes_per_launch) *bytes_per_launch = bytes;
    return CMR_OK;
}

}  // extern "C"
