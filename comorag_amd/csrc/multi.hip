// Multi-device index behind the C-ABI (include/comorag_hip.h, "one process, several devices"): S row shards — each a
// cmr_index_t on its own GPU of the node — owned by ONE process, so that ComoRAG's single-process, 16-thread
// probe -> retrieve -> consolidate loop (src/comorag/ComoRAG.py:432-453, tri_retrieve :456-554) can sit on an index that is
// sharded over the node's GPUs.  (The one-process-per-GPU twin of this is comorag_amd/sharded.py + comm.hip.)
//
//  * layout: global row ids are dense in append order — what EmbeddingStore._upsert (embedding_store.py:122-128) and
//    MemoryPool.add_node (utils/memory_utils.py:294-300) rely on; appended rows go to the shards in blocks (a block opens on
//    the currently shortest shard; a bulk append opens blocks of ceil(m / S) rows, i.e. contiguous row blocks), each shard
//    translates its rows through its block table (cmr_index_set_id_blocks), so every shard already answers in global ids;
//  * a search begins on every shard before it finishes on any (cmr_index_search_begin / _finish): all devices scan at once;
//    with more than two shards the per-shard enqueues are issued by per-shard worker threads, because ONE host thread issuing
//    8 shards' launches back to back (~30-50 us each) would be on the critical path of a 0.3 ms shard scan;
//  * candidate exchange: every shard's merge kernel writes its [nq, k] candidates straight into pinned, device-mapped HOST
//    memory (nq * k * 12 bytes per shard over PCIe — no peer access, no collective, no extra launch), the host then does the
//    final S-way merge of the sorted lists with the exported tie rule (north_star: "host-side final merge"), so the result
//    equals a single index's by construction.
// The reference has no multi-device path; nothing here mirrors reference code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "cmr_internal.h"

namespace {

#define M_HIP_TRY(expr)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return cmr_fail(e_ == hipErrorOutOfMemory ? CMR_ERR_OOM : CMR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    int left = 0;
    void arm(int n) { std::lock_guard<std::mutex> g(mu); left = n; }
    void done() { std::lock_guard<std::mutex> g(mu); if (--left == 0) cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return left == 0; }); }
};

struct Job {
    std::function<int()> fn;
    int* rc = nullptr;
    std::string* err = nullptr;
    Latch* latch = nullptr;
};

// one thread per shard: issues that shard's enqueues (HIP's current device is per thread: set once per job by the shard's
// own entry points)
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool stop = false;
    void loop() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = std::move(q.front());
                q.pop_front();
            }
            const int rc = j.fn();
            if (j.rc) *j.rc = rc;
            if (rc && j.err) { const char* e = cmr_last_error(); *j.err = e ? e : ""; }
            if (j.latch) j.latch->done();
        }
    }
    void post(Job j) {
        { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(j)); }
        cv.notify_one();
    }
};

struct Chunk { int shard; long long n; };

// Where do m appended rows go?  Rows keep filling the open block (`room` rows left on shard `cur`); a new block opens on the
// currently shortest shard (lowest index on ties) with max(block_rows, ceil(m / S)) rows — a bulk append therefore lands as
// contiguous row blocks of equal size, a memory pool's 25-row appends as blocks of block_rows rows that go round the shards
// (the routing of comorag_amd/sharded.py:_route for m <= S * block_rows).  Deterministic from (sizes, cur, room).
void plan_append(const long long* sizes, int S, int& cur, long long& room, long long m, long long block_rows, std::vector<Chunk>& out) {
    std::vector<long long> sz(sizes, sizes + S);
    const long long big = std::max(block_rows, (m + S - 1) / S);
    while (m > 0) {
        if (cur < 0 || room <= 0) {
            cur = 0;
            for (int s = 1; s < S; ++s) if (sz[s] < sz[cur]) cur = s;
            room = big;
        }
        const long long take = std::min(m, room);
        if (!out.empty() && out.back().shard == cur) out.back().n += take;
        else out.push_back({cur, take});
        sz[cur] += take;
        room -= take;
        m -= take;
    }
}

// better(a, b): candidate a precedes candidate b in the exported order (score descending, then id ascending)
inline bool better(float sa, int64_t ia, float sb, int64_t ib) { return sa > sb || (sa == sb && ia < ib); }

// per-shard candidate lists [A][nq][k] (each sorted in the exported order, -1 padded at the tail) -> [nq][k]
void merge_sorted_lists(const int64_t* ids, const float* sc, int A, int nq, int k, int64_t* out_ids, float* out_sc) {
    std::vector<int> head((size_t)A);
    for (int q = 0; q < nq; ++q) {
        std::fill(head.begin(), head.end(), 0);
        for (int j = 0; j < k; ++j) {
            int best = -1;
            float bs = 0.0f;
            int64_t bi = 0;
            for (int a = 0; a < A; ++a) {
                if (head[a] >= k) continue;
                const size_t o = ((size_t)a * nq + q) * k + head[a];
                if (ids[o] < 0) { head[a] = k; continue; }
                const float s = sc[o] + 0.0f;
                if (best < 0 || better(s, ids[o], bs, bi)) { best = a; bs = s; bi = ids[o]; }
            }
            if (best < 0) { out_ids[(size_t)q * k + j] = -1; out_sc[(size_t)q * k + j] = -INFINITY; continue; }
            out_ids[(size_t)q * k + j] = bi;
            out_sc[(size_t)q * k + j] = bs;
            ++head[best];
        }
    }
}

constexpr int kSlots = 4;           // pipelined batches whose results may be uncollected at once

struct PipeTicket {
    std::atomic<bool> busy{false};
    int nq = 0, k = 0;
    std::vector<int> active;        // shards that took part
    std::vector<void*> done;        // their hipEvent_t (owned by the shard indexes)
    std::vector<int> rc;
    std::vector<std::string> err;
    Latch latch;                    // the shards' enqueues have been issued
    // pinned, device-mapped host memory the shards' kernels write into: [A][nq][k] ids | [A][nq][k] scores | [A][nq] min | [A][nq] max
    void* h = nullptr;
    size_t cap = 0;
};

}  // namespace

struct cmr_mindex {
    int S = 0, dim = 0, dtype = 0;
    uint32_t flags = 0;
    std::vector<cmr_index_t*> shard;
    std::vector<int> device;
    std::shared_mutex mu;                       // layout: searches shared, append / destroy exclusive
    std::vector<std::vector<long long>> blk_local, blk_global;      // per shard: its runs of consecutive global ids
    std::vector<long long> rows;                // per shard
    long long total = 0;
    long long block_rows = 65536;                 // append_block_rows: a corpus of a few thousand rows (what ComoRAG indexes) stays on ONE shard and
                                                // is searched without any multi-shard overhead; shards fill up block by block
    int cur = -1;                               // shard of the open block
    long long room = 0;                         // rows left in it
    int parallel_min_shards = 3;                // per-shard worker threads issue the enqueues from this many active shards on
    bool force_peer_staging = false;            // option "force_peer_staging": append_dev always stages through hipMemcpyPeer (see cmr_mindex_set_option)
    std::mutex wk_mu;
    std::vector<Worker*> workers;
    std::mutex pipe_mu;
    PipeTicket ticket[kSlots];
    unsigned next_ticket = 0;
    // Shards that SHARE a device (logical shards: the 1-GPU rehearsal of the layout) run their throughput-mode batches on one
    // plain stream each (cmr_index_search_dev) instead of the index's own pipeline: S pipelines of 11 streams each on one device
    // alias its few hardware queues and serialise on their event waits (8 shards: 8.6 ms per step against 1.06 ms for the same
    // rows in one index).  One shard per device — the layout this type exists for — keeps the pipelined path.
    std::vector<char> colocated;                // per shard: does its device hold another shard too?
    std::vector<hipStream_t> colo_stream;       // per shard (lazily, on its device)
    std::vector<hipEvent_t> colo_event;         // [kSlots][S] (lazily)
    // host-side cost of the throughput mode (cmr_mindex_profile): ns spent in the shards' enqueue jobs, in collect's waits for the
    // shards' events, and in the host merge
    std::atomic<long long> prof_jobs{0}, prof_enqueue_ns{0}, prof_batches{0}, prof_wait_ns{0}, prof_merge_ns{0};
};

namespace {

int ensure_workers(cmr_mindex* m) {
    std::lock_guard<std::mutex> g(m->wk_mu);
    if (!m->workers.empty()) return CMR_OK;
    for (int s = 0; s < m->S; ++s) {
        Worker* w = new Worker();
        w->th = std::thread([w] { w->loop(); });
        m->workers.push_back(w);
    }
    return CMR_OK;
}

// run fn(a, shard) for every active shard: on the shards' worker threads when `parallel`, else here; first error wins
int for_active(cmr_mindex* m, const std::vector<int>& act, bool parallel, const std::function<int(int, int)>& fn) {
    const int A = (int)act.size();
    if (!parallel || A <= 1) {
        for (int a = 0; a < A; ++a) { const int rc = fn(a, act[a]); if (rc) return rc; }
        return CMR_OK;
    }
    int rc0 = ensure_workers(m);
    if (rc0) return rc0;
    std::vector<int> rc((size_t)A, 0);
    std::vector<std::string> err((size_t)A);
    Latch latch;
    latch.arm(A);
    for (int a = 0; a < A; ++a) {
        Job j;
        const int s = act[a];
        j.fn = [&fn, a, s] { return fn(a, s); };
        j.rc = &rc[a]; j.err = &err[a]; j.latch = &latch;
        m->workers[s]->post(std::move(j));
    }
    latch.wait();
    for (int a = 0; a < A; ++a) if (rc[a]) return cmr_fail(rc[a], "shard %d: %s", act[a], err[a].c_str());
    return CMR_OK;
}

std::vector<int> active_shards(const cmr_mindex* m) {
    std::vector<int> a;
    for (int s = 0; s < m->S; ++s) if (m->rows[s] > 0) a.push_back(s);
    return a;
}

int push_blocks(cmr_mindex* m, int s) {
    const auto& l = m->blk_local[s];
    const auto& g = m->blk_global[s];
    if (l.empty()) return cmr_index_set_id_base(m->shard[s], 0);
    return cmr_index_set_id_blocks(m->shard[s], (int32_t)l.size(), (const int64_t*)l.data(), (const int64_t*)g.data());
}

// (shard, local row) of a global id, or shard -1
struct Where { int shard; long long local; };
Where locate(const cmr_mindex* m, long long gid) {
    if (gid < 0 || gid >= m->total) return {-1, -1};
    for (int s = 0; s < m->S; ++s) {
        const auto& g = m->blk_global[s];
        const auto& l = m->blk_local[s];
        if (g.empty()) continue;
        size_t b = std::upper_bound(g.begin(), g.end(), gid) - g.begin();
        if (b == 0) continue;
        --b;
        const long long len = (b + 1 < l.size() ? l[b + 1] : m->rows[s]) - l[b];
        if (gid - g[b] < len) return {s, l[b] + (gid - g[b])};
    }
    return {-1, -1};
}

}  // namespace

extern "C" {

int32_t cmr_mindex_plan_append(const int64_t* shard_rows, int32_t n_shards, int32_t cur_shard, int64_t cur_room, int64_t m, int64_t block_rows,
                               int32_t max_chunks, int32_t* out_shard, int64_t* out_count, int32_t* n_chunks, int32_t* new_cur, int64_t* new_room) {
    if (!shard_rows || n_shards <= 0 || m < 0 || block_rows <= 0 || !n_chunks) return cmr_fail(CMR_ERR_INVALID, "bad argument");
    std::vector<long long> sz(shard_rows, shard_rows + n_shards);
    std::vector<Chunk> ch;
    int cur = cur_shard;
    long long room = cur_room;
    plan_append(sz.data(), n_shards, cur, room, m, block_rows, ch);
    *n_chunks = (int32_t)ch.size();
    if ((int)ch.size() > max_chunks && (out_shard || out_count)) return cmr_fail(CMR_ERR_INVALID, "%d chunks, room for %d", (int)ch.size(), max_chunks);
    for (size_t i = 0; i < ch.size(); ++i) {
        if (out_shard) out_shard[i] = ch[i].shard;
        if (out_count) out_count[i] = ch[i].n;
    }
    if (new_cur) *new_cur = cur;
    if (new_room) *new_room = room;
    return CMR_OK;
}

int32_t cmr_mindex_create(int32_t n_shards, const int32_t* device_ids, int32_t dim, int32_t dtype, int64_t capacity_hint, uint32_t flags,
                          cmr_mindex_t** out) {
    if (!out) return cmr_fail(CMR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_shards <= 0 || n_shards > 64) return cmr_fail(CMR_ERR_INVALID, "n_shards %d outside [1, 64]", n_shards);
    cmr_mindex* m = new cmr_mindex();
    m->S = n_shards; m->dim = dim; m->dtype = dtype; m->flags = flags;
    m->rows.assign((size_t)n_shards, 0);
    m->blk_local.resize((size_t)n_shards);
    m->blk_global.resize((size_t)n_shards);
    for (int s = 0; s < n_shards; ++s) {
        const int dev = device_ids ? device_ids[s] : 0;
        cmr_index_t* idx = nullptr;
        const int rc = cmr_index_create(dev, dim, dtype, capacity_hint > 0 ? (capacity_hint + n_shards - 1) / n_shards : 0, flags, &idx);
        if (rc) {
            for (cmr_index_t* i : m->shard) (void)cmr_index_destroy(i);
            delete m;
            return rc;
        }
        m->shard.push_back(idx);
        m->device.push_back(dev);
    }
    m->colocated.assign((size_t)n_shards, 0);
    for (int s = 0; s < n_shards; ++s)
        for (int o = 0; o < n_shards; ++o) if (o != s && m->device[o] == m->device[s]) m->colocated[s] = 1;
    m->colo_stream.assign((size_t)n_shards, nullptr);
    m->colo_event.assign((size_t)kSlots * n_shards, nullptr);
    *out = m;
    return CMR_OK;
}

int32_t cmr_mindex_destroy(cmr_mindex_t* m) {
    if (!m) return CMR_OK;
    {
        std::unique_lock<std::shared_mutex> lk(m->mu);
        for (Worker* w : m->workers) {
            { std::lock_guard<std::mutex> g(w->mu); w->stop = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
            delete w;
        }
        m->workers.clear();
        for (PipeTicket& t : m->ticket) if (t.h) { (void)hipHostFree(t.h); t.h = nullptr; }
        for (int s = 0; s < m->S; ++s) {
            (void)hipSetDevice(m->device[s]);
            if (m->colo_stream[s]) { (void)hipStreamSynchronize(m->colo_stream[s]); (void)hipStreamDestroy(m->colo_stream[s]); }
            for (int sl = 0; sl < kSlots; ++sl) if (m->colo_event[(size_t)sl * m->S + s]) (void)hipEventDestroy(m->colo_event[(size_t)sl * m->S + s]);
        }
        for (cmr_index_t* i : m->shard) (void)cmr_index_destroy(i);
    }
    delete m;
    return CMR_OK;
}

int32_t cmr_mindex_size(cmr_mindex_t* m, int64_t* n_rows) {
    if (!m || !n_rows) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    std::shared_lock<std::shared_mutex> lk(m->mu);
    *n_rows = m->total;
    return CMR_OK;
}

int32_t cmr_mindex_info(cmr_mindex_t* m, int32_t* n_shards, int32_t* device_ids, int64_t* shard_rows, int64_t* device_bytes) {
    if (!m) return cmr_fail(CMR_ERR_INVALID, "NULL index");
    std::shared_lock<std::shared_mutex> lk(m->mu);
    if (n_shards) *n_shards = m->S;
    int64_t bytes = 0;
    for (int s = 0; s < m->S; ++s) {
        if (device_ids) device_ids[s] = m->device[s];
        if (shard_rows) shard_rows[s] = m->rows[s];
        int64_t b = 0;
        const int rc = cmr_index_info(m->shard[s], nullptr, nullptr, nullptr, &b);
        if (rc) return rc;
        bytes += b;
    }
    if (device_bytes) *device_bytes = bytes;
    return CMR_OK;
}

int32_t cmr_mindex_shard(cmr_mindex_t* m, int32_t s, cmr_index_t** out) {
    if (!m || !out) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (s < 0 || s >= m->S) return cmr_fail(CMR_ERR_INVALID, "shard %d outside [0, %d)", s, m->S);
    *out = m->shard[s];
    return CMR_OK;
}

int32_t cmr_mindex_set_option(cmr_mindex_t* m, const char* name, int64_t value) {
    if (!m || !name) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    const std::string n(name);
    if (n == "append_block_rows") {
        if (value < 1) return cmr_fail(CMR_ERR_INVALID, "append_block_rows must be >= 1");
        std::unique_lock<std::shared_mutex> lk(m->mu);
        m->block_rows = value;
        return CMR_OK;
    }
    if (n == "parallel_min_shards") {
        std::unique_lock<std::shared_mutex> lk(m->mu);
        m->parallel_min_shards = (int)std::max<int64_t>(1, value);
        return CMR_OK;
    }
    if (n == "force_peer_staging") {
        // cmr_mindex_append_dev: take the staging-buffer + hipMemcpyPeer branch even when source and shard share a device (legal: same
        // device on both ends) — the only way a one-GPU box executes the cross-device append path (tests/test_multi_device_gpu.py)
        std::unique_lock<std::shared_mutex> lk(m->mu);
        m->force_peer_staging = value != 0;
        return CMR_OK;
    }
    // Route selectors go to every shard.  Multi-shard searches call cmr_index_search_begin(take_lock = false) under m->mu SHARED and read
    // the shards' option fields while they enqueue: the forwarding loop therefore holds m->mu EXCLUSIVELY (no search in flight while an
    // option changes; a routing decision is never split inside one call).  The borrowed handles of cmr_mindex_shard are for reading
    // (get_option, profile_collect) while searches run — set options through THIS function.
    std::unique_lock<std::shared_mutex> lk(m->mu);
    {   // throughput-mode batches whose enqueue jobs are still with the worker threads: let them finish issuing first
        std::lock_guard<std::mutex> pg(m->pipe_mu);
        for (PipeTicket& t : m->ticket) if (t.busy) t.latch.wait();
    }
    for (cmr_index_t* i : m->shard) { const int rc = cmr_index_set_option(i, name, value); if (rc) return rc; }
    return CMR_OK;
}

// the routed, all-or-nothing append: `put(shard, first row of the chunk, rows in it)` places one chunk on one shard
static int32_t mindex_append(cmr_mindex_t* m, int64_t n, const std::function<int(int, long long, long long)>& put) {
    std::unique_lock<std::shared_mutex> lk(m->mu);
    if (m->total + n - 1 > 0xFFFFFFFEll) return cmr_fail(CMR_ERR_UNSUPPORTED, "global row ids must stay below 2^32 - 1");
    // state to roll back to
    const std::vector<long long> rows0 = m->rows;
    const auto bl0 = m->blk_local, bg0 = m->blk_global;
    const int cur0 = m->cur;
    const long long room0 = m->room, total0 = m->total;
    std::vector<Chunk> plan;
    int cur = m->cur;
    long long room = m->room;
    plan_append(m->rows.data(), m->S, cur, room, n, m->block_rows, plan);
    std::vector<char> touched((size_t)m->S, 0);
    long long at = 0;
    int rc = CMR_OK;
    std::string err;
    for (const Chunk& c : plan) {
        const int s = c.shard;
        const long long local_at = m->rows[s], gid = m->total;
        rc = put(s, at, c.n);
        if (rc) { const char* e = cmr_last_error(); err = e ? e : ""; break; }
        touched[s] = 1;
        auto& l = m->blk_local[s];
        auto& g = m->blk_global[s];
        bool changed = false;
        if (l.empty()) { l.push_back(0); g.push_back(gid); changed = true; }
        else if (g.back() + (local_at - l.back()) != gid) {           // not a continuation of the shard's last run: a new block
            if (local_at == l.back()) g.back() = gid;                  // (the last run is still empty: it simply starts elsewhere)
            else { l.push_back(local_at); g.push_back(gid); }
            changed = true;
        }
        m->rows[s] += c.n;
        m->total += c.n;
        at += c.n;
        if (changed) {
            rc = push_blocks(m, s);
            if (rc) { const char* e = cmr_last_error(); err = e ? e : ""; break; }
        }
    }
    if (rc) {      // all or nothing: shards that already took their chunk go back to where they were
        m->blk_local = bl0; m->blk_global = bg0; m->cur = cur0; m->room = room0; m->total = total0;
        for (int s = 0; s < m->S; ++s) {
            if (!touched[s]) continue;
            (void)cmr_index_truncate(m->shard[s], rows0[s]);
            m->rows[s] = rows0[s];
            (void)push_blocks(m, s);
        }
        m->rows = rows0;
        return cmr_fail(rc, "%s", err.c_str());
    }
    m->cur = cur;
    m->room = room;
    return CMR_OK;
}

int32_t cmr_mindex_append(cmr_mindex_t* m, const float* rows, int64_t n) {
    if (!m || (n > 0 && !rows)) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (n < 0) return cmr_fail(CMR_ERR_INVALID, "n < 0");
    if (n == 0) return CMR_OK;
    return mindex_append(m, n, [&](int s, long long at, long long cnt) { return cmr_index_append(m->shard[s], rows + (size_t)at * m->dim, cnt); });
}

int32_t cmr_mindex_append_dev(cmr_mindex_t* m, const float* rows_dev, int64_t n, int32_t src_device, void* stream) {
    if (!m || (n > 0 && !rows_dev)) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (n < 0) return cmr_fail(CMR_ERR_INVALID, "n < 0");
    if (n == 0) return CMR_OK;
    bool synced = false;
    return mindex_append(m, n, [&](int s, long long at, long long cnt) -> int {
        const float* src = rows_dev + (size_t)at * m->dim;
        if (m->device[s] == src_device && !m->force_peer_staging) return cmr_index_append_dev(m->shard[s], src, cnt, stream);
        // the chunk's shard lives on another GPU: device-to-device copy into a staging buffer there (hipMemcpyPeer goes over
        // xGMI where the devices are peers and through the host where they are not), then a local append
        if (!synced) { M_HIP_TRY(hipSetDevice(src_device)); M_HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); synced = true; }
        M_HIP_TRY(hipSetDevice(m->device[s]));
        void* stage = nullptr;
        const size_t bytes = (size_t)cnt * m->dim * 4;
        M_HIP_TRY(hipMalloc(&stage, bytes));
        hipError_t e = hipMemcpyPeer(stage, m->device[s], src, src_device, bytes);
        int rc = e == hipSuccess ? cmr_index_append_dev(m->shard[s], (const float*)stage, cnt, nullptr)
                                 : cmr_fail(CMR_ERR_HIP, "hipMemcpyPeer %d -> %d: %s", src_device, m->device[s], hipGetErrorString(e));
        (void)hipSetDevice(m->device[s]);
        (void)hipFree(stage);
        return rc;
    });
}

static int32_t mindex_search(cmr_mindex_t* m, const float* q, int32_t nq, int32_t k, const float* min_score, int64_t* out_ids, float* out_scores,
                             float* out_min, float* out_max) {
    if (!m || !q || !out_ids || !out_scores) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return cmr_fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K_2PASS) return cmr_fail(CMR_ERR_UNSUPPORTED, "k = %d outside [1, %d]", k, CMR_MAX_K_2PASS);
    std::shared_lock<std::shared_mutex> lk(m->mu);
    const std::vector<int> act = active_shards(m);
    const int A = (int)act.size();
    const size_t nk = (size_t)nq * k;
    if (A == 0) {
        for (size_t i = 0; i < nk; ++i) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
        for (int i = 0; i < nq; ++i) { if (out_min) out_min[i] = INFINITY; if (out_max) out_max[i] = -INFINITY; }
        return CMR_OK;
    }
    std::vector<CmrPending*> pend((size_t)A, nullptr);
    int rc = for_active(m, act, A >= m->parallel_min_shards, [&](int a, int s) {
        return cmr_index_search_begin(m->shard[s], q, nq, k, min_score, false, &pend[a]);
    });
    if (rc) {
        const std::string keep = cmr_last_error();
        for (CmrPending* p : pend) if (p) cmr_index_search_abandon(p);
        return cmr_fail(rc, "%s", keep.c_str());
    }
    if (A == 1) return cmr_index_search_finish(pend[0], out_ids, out_scores, out_min, out_max);
    std::vector<int64_t> ids((size_t)A * nk);
    std::vector<float> sc((size_t)A * nk), mn((size_t)A * nq), mx((size_t)A * nq);
    std::string err;
    for (int a = 0; a < A; ++a) {
        const int r = cmr_index_search_finish(pend[a], ids.data() + (size_t)a * nk, sc.data() + (size_t)a * nk, mn.data() + (size_t)a * nq, mx.data() + (size_t)a * nq);
        pend[a] = nullptr;
        if (r && !rc) { rc = r; err = cmr_last_error(); }
    }
    if (rc) return cmr_fail(rc, "%s", err.c_str());
    merge_sorted_lists(ids.data(), sc.data(), A, nq, k, out_ids, out_scores);
    for (int i = 0; i < nq; ++i) {
        float lo = INFINITY, hi = -INFINITY;
        for (int a = 0; a < A; ++a) { lo = std::fmin(lo, mn[(size_t)a * nq + i]); hi = std::fmax(hi, mx[(size_t)a * nq + i]); }
        if (out_min) out_min[i] = lo;
        if (out_max) out_max[i] = hi;
    }
    return CMR_OK;
}

int32_t cmr_mindex_search(cmr_mindex_t* m, const float* q, int32_t nq, int32_t k, int64_t* out_ids, float* out_scores, float* out_min,
                          float* out_max) {
    return mindex_search(m, q, nq, k, nullptr, out_ids, out_scores, out_min, out_max);
}

int32_t cmr_mindex_search_min_score(cmr_mindex_t* m, const float* q, int32_t nq, int32_t k, float min_score, int64_t* out_ids, float* out_scores) {
    if (!(min_score == min_score)) return cmr_fail(CMR_ERR_INVALID, "min_score is NaN");
    if (k > CMR_MAX_K) return cmr_fail(CMR_ERR_UNSUPPORTED, "threshold search supports k in [1, %d]", CMR_MAX_K);
    return mindex_search(m, q, nq, k, &min_score, out_ids, out_scores, nullptr, nullptr);
}

// big enough that a thread hand-off (~10 us) per shard is noise
static bool big_job(const cmr_mindex* m, long long work_rows) { return work_rows >= 262144; }

int32_t cmr_mindex_scores(cmr_mindex_t* m, const float* q, int32_t nq, float* out, int64_t ld) {
    if (!m || !q || !out) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return cmr_fail(CMR_ERR_INVALID, "nq must be > 0");
    std::shared_lock<std::shared_mutex> lk(m->mu);
    if (ld == 0) ld = m->total;
    if (ld < m->total) return cmr_fail(CMR_ERR_INVALID, "ld %lld < rows %lld", (long long)ld, m->total);
    const std::vector<int> act = active_shards(m);
    return for_active(m, act, big_job(m, m->total) && (int)act.size() >= 2, [&](int, int s) -> int {
        const auto& l = m->blk_local[s];
        const auto& g = m->blk_global[s];
        if (l.size() == 1)      // one run of consecutive global ids: the shard's scores go straight to their place
            return cmr_index_scores(m->shard[s], q, nq, out + g[0], ld);
        const long long ns = m->rows[s];
        std::vector<float> tmp((size_t)nq * ns);
        const int rc = cmr_index_scores(m->shard[s], q, nq, tmp.data(), ns);
        if (rc) return rc;
        for (int qi = 0; qi < nq; ++qi)
            for (size_t b = 0; b < l.size(); ++b) {
                const long long len = (b + 1 < l.size() ? l[b + 1] : ns) - l[b];
                memcpy(out + (size_t)qi * ld + g[b], tmp.data() + (size_t)qi * ns + l[b], (size_t)len * 4);
            }
        return CMR_OK;
    });
}

int32_t cmr_mindex_sorted_scores(cmr_mindex_t* m, const float* q, int32_t nq, int64_t* out_ids, float* out_scores, float* out_min, float* out_max) {
    if (!m || !q || !out_ids || !out_scores) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0) return cmr_fail(CMR_ERR_INVALID, "nq must be > 0");
    std::shared_lock<std::shared_mutex> lk(m->mu);
    const long long n = m->total;
    if (n == 0) return CMR_OK;
    const std::vector<int> act = active_shards(m);
    const int A = (int)act.size();
    if (A == 1) return cmr_index_sorted_scores(m->shard[act[0]], q, nq, out_ids, out_scores, out_min, out_max);
    // every shard ranks its rows (ids already global); the host merges the A sorted runs of every query pairwise
    std::vector<std::vector<int64_t>> ids((size_t)A);
    std::vector<std::vector<float>> sc((size_t)A);
    int rc = for_active(m, act, big_job(m, n), [&](int a, int s) -> int {
        const long long ns = m->rows[s];
        ids[a].resize((size_t)nq * ns);
        sc[a].resize((size_t)nq * ns);
        return cmr_index_sorted_scores(m->shard[s], q, nq, ids[a].data(), sc[a].data(), nullptr, nullptr);
    });
    if (rc) return rc;
    struct E { float s; int64_t id; };
    std::vector<E> x((size_t)n), y((size_t)n);
    for (int qi = 0; qi < nq; ++qi) {
        // runs laid out back to back in x, then merged pairwise (x -> y -> x ...) until one run is left
        std::vector<long long> start{0};
        for (int a = 0; a < A; ++a) {
            const long long ns = m->rows[act[a]];
            E* dst = x.data() + start.back();
            const int64_t* ia = ids[a].data() + (size_t)qi * ns;
            const float* sa = sc[a].data() + (size_t)qi * ns;
            for (long long i = 0; i < ns; ++i) dst[i] = {sa[i] + 0.0f, ia[i]};
            start.push_back(start.back() + ns);
        }
        E* src = x.data();
        E* dst = y.data();
        while (start.size() > 2) {
            std::vector<long long> ns{0};
            for (size_t r = 0; r + 1 < start.size(); r += 2) {
                const long long a0 = start[r], a1 = start[r + 1], b1 = r + 2 < start.size() ? start[r + 2] : a1;
                std::merge(src + a0, src + a1, src + a1, src + b1, dst + a0, [](const E& u, const E& v) { return better(u.s, u.id, v.s, v.id); });
                ns.push_back(b1);
            }
            start.swap(ns);
            std::swap(src, dst);
        }
        for (long long i = 0; i < n; ++i) { out_ids[(size_t)qi * n + i] = src[i].id; out_scores[(size_t)qi * n + i] = src[i].s; }
        if (out_max) out_max[qi] = src[0].s;
        if (out_min) out_min[qi] = src[n - 1].s;
    }
    return CMR_OK;
}

int32_t cmr_mindex_rescore(cmr_mindex_t* m, const float* q, int32_t nq, const int64_t* cand, int32_t n_cand, int32_t k, int64_t* out_ids,
                           float* out_scores) {
    if (!m || !q || !cand || !out_ids || !out_scores) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0 || n_cand <= 0 || k <= 0) return cmr_fail(CMR_ERR_INVALID, "nq, n_cand, k must be > 0");
    if (k > n_cand) k = n_cand;
    std::shared_lock<std::shared_mutex> lk(m->mu);
    const std::vector<int> act = active_shards(m);
    const int A = (int)act.size();
    const size_t nk = (size_t)nq * k;
    if (A == 0) {
        for (size_t i = 0; i < nk; ++i) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
        return CMR_OK;
    }
    if (A == 1) return cmr_index_rescore(m->shard[act[0]], q, nq, cand, n_cand, k, out_ids, out_scores);
    // every shard scores the candidates it holds (ids it does not hold are skipped by the kernel), the host merges
    std::vector<int64_t> ids((size_t)A * nk);
    std::vector<float> sc((size_t)A * nk);
    int rc = for_active(m, act, false, [&](int a, int s) {
        return cmr_index_rescore(m->shard[s], q, nq, cand, n_cand, k, ids.data() + (size_t)a * nk, sc.data() + (size_t)a * nk);
    });
    if (rc) return rc;
    merge_sorted_lists(ids.data(), sc.data(), A, nq, k, out_ids, out_scores);
    return CMR_OK;
}

int32_t cmr_mindex_get_rows(cmr_mindex_t* m, const int64_t* ids, int64_t n, float* out) {
    if (!m || (n > 0 && (!ids || !out))) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (n <= 0) return CMR_OK;
    std::shared_lock<std::shared_mutex> lk(m->mu);
    std::vector<std::vector<int64_t>> want((size_t)m->S);
    std::vector<std::vector<int64_t>> pos((size_t)m->S);
    for (int64_t i = 0; i < n; ++i) {
        const Where w = locate(m, ids[i]);
        if (w.shard < 0) { memset(out + (size_t)i * m->dim, 0, (size_t)m->dim * 4); continue; }   // as a single index answers an id it does not hold
        want[w.shard].push_back(ids[i]);
        pos[w.shard].push_back(i);
    }
    std::vector<float> tmp;
    for (int s = 0; s < m->S; ++s) {
        if (want[s].empty()) continue;
        tmp.resize(want[s].size() * (size_t)m->dim);
        const int rc = cmr_index_get_rows(m->shard[s], want[s].data(), (int64_t)want[s].size(), tmp.data());
        if (rc) return rc;
        for (size_t j = 0; j < want[s].size(); ++j) memcpy(out + (size_t)pos[s][j] * m->dim, tmp.data() + j * (size_t)m->dim, (size_t)m->dim * 4);
    }
    return CMR_OK;
}

// ---- throughput mode -------------------------------------------------------------------------------------------------------
int32_t cmr_mindex_search_pipelined(cmr_mindex_t* m, const float* const* q_dev, int32_t nq, int32_t k, void** ticket) {
    if (!m || !q_dev || !ticket) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    *ticket = nullptr;
    if (nq <= 0) return cmr_fail(CMR_ERR_INVALID, "nq must be > 0");
    if (k <= 0 || k > CMR_MAX_K) return cmr_fail(CMR_ERR_UNSUPPORTED, "pipelined search supports k <= %d", CMR_MAX_K);
    std::shared_lock<std::shared_mutex> lk(m->mu);
    std::lock_guard<std::mutex> pg(m->pipe_mu);
    PipeTicket& t = m->ticket[m->next_ticket % kSlots];
    if (t.busy) return cmr_fail(CMR_ERR_INVALID, "%d pipelined batches are uncollected: call cmr_mindex_collect on the oldest ticket first", kSlots);
    const std::vector<int> act = active_shards(m);
    const int A = (int)act.size();
    for (int a = 0; a < A; ++a) if (!q_dev[act[a]]) return cmr_fail(CMR_ERR_INVALID, "q_dev[%d] is NULL", act[a]);
    const size_t nk = (size_t)nq * k;
    const size_t need = (size_t)std::max(A, 1) * (nk * 12 + (size_t)nq * 8);
    if (need > t.cap) {
        if (t.h) { M_HIP_TRY(hipHostFree(t.h)); t.h = nullptr; t.cap = 0; }
        // portable + mapped: every device of the process may write it (unified addressing: host pointer == device pointer)
        M_HIP_TRY(hipHostMalloc(&t.h, need, hipHostMallocPortable | hipHostMallocMapped));
        t.cap = need;
    }
    // everything that can fail comes BEFORE the slot is claimed: on an error the caller's *ticket stays as it was and no ring position is used up
    if (A > 0) { const int rcw = ensure_workers(m); if (rcw) return rcw; }
    t.nq = nq; t.k = k; t.active = act;
    t.done.assign((size_t)A, nullptr);
    t.rc.assign((size_t)A, 0);
    t.err.assign((size_t)A, std::string());
    t.busy = true;
    ++m->next_ticket;
    *ticket = &t;
    if (A == 0) { t.latch.arm(0); return CMR_OK; }
    void* hbase = t.h;
    const size_t o_sc = (size_t)A * nk * 8, o_mn = (size_t)A * nk * 12, o_mx = o_mn + (size_t)A * nq * 4;
    t.latch.arm(A);
    for (int a = 0; a < A; ++a) {
        const int s = act[a];
        Job j;
        cmr_index_t* idx = m->shard[s];
        const float* qd = q_dev[s];
        void** done = &t.done[a];
        const int dev = m->device[s];
        const int slot = (int)(&t - m->ticket);
        j.fn = [=]() -> int {
            const auto t0 = std::chrono::steady_clock::now();
            struct Tick { cmr_mindex* m; std::chrono::steady_clock::time_point t0; ~Tick() { m->prof_enqueue_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++m->prof_jobs; } } tick{m, t0};
            // the device's view of the pinned buffer (the same address under unified addressing; asked for, not assumed)
            hipError_t e = hipSetDevice(dev);
            if (e != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
            void* dbase = nullptr;
            e = hipHostGetDevicePointer(&dbase, hbase, 0);
            if (e != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipHostGetDevicePointer on device %d: %s", dev, hipGetErrorString(e));
            char* d = (char*)dbase;
            if (m->colocated[s]) {      // a device shared with other shards: one plain stream per shard (see cmr_mindex::colocated)
                if (!m->colo_stream[s]) {
                    e = hipStreamCreateWithFlags(&m->colo_stream[s], hipStreamNonBlocking);
                    if (e != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
                }
                hipEvent_t& ev = m->colo_event[(size_t)slot * m->S + s];
                if (!ev) {
                    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                    if (e != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipEventCreateWithFlags: %s", hipGetErrorString(e));
                }
                const int rc = cmr_index_search_dev(idx, qd, nq, k, (int64_t*)d + (size_t)a * nk, (float*)(d + o_sc) + (size_t)a * nk,
                                                    (float*)(d + o_mn) + (size_t)a * nq, (float*)(d + o_mx) + (size_t)a * nq, m->colo_stream[s]);
                if (rc) return rc;
                e = hipEventRecord(ev, m->colo_stream[s]);
                if (e != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
                *done = (void*)ev;
                return CMR_OK;
            }
            return cmr_index_search_pipelined(idx, qd, nq, k, (int64_t*)d + (size_t)a * nk, (float*)(d + o_sc) + (size_t)a * nk,
                                              (float*)(d + o_mn) + (size_t)a * nq, (float*)(d + o_mx) + (size_t)a * nq, nullptr, done);
        };
        j.rc = &t.rc[a]; j.err = &t.err[a]; j.latch = &t.latch;
        m->workers[s]->post(std::move(j));
    }
    return CMR_OK;
}

int32_t cmr_mindex_collect(cmr_mindex_t* m, void* ticket, int64_t* out_ids, float* out_scores, float* out_min, float* out_max) {
    if (!m || !ticket || !out_ids || !out_scores) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    PipeTicket* t = (PipeTicket*)ticket;
    if (t < m->ticket || t >= m->ticket + kSlots || !t->busy) return cmr_fail(CMR_ERR_INVALID, "not an outstanding ticket of this index");
    const auto tw0 = std::chrono::steady_clock::now();
    t->latch.wait();                 // the shards' enqueues have been issued
    struct Free { PipeTicket* t; ~Free() { t->busy = false; } } fr{t};
    const int A = (int)t->active.size(), nq = t->nq, k = t->k;
    const size_t nk = (size_t)nq * k;
    int rc = CMR_OK;
    std::string err;
    for (int a = 0; a < A; ++a) {
        if (t->rc[a]) { if (!rc) { rc = t->rc[a]; err = "shard " + std::to_string(t->active[a]) + ": " + t->err[a]; } continue; }
        if (t->done[a] && hipEventSynchronize((hipEvent_t)t->done[a]) != hipSuccess && !rc) { rc = CMR_ERR_HIP; err = "hipEventSynchronize failed"; }
    }
    if (rc) return cmr_fail(rc, "%s", err.c_str());
    if (A == 0) {
        for (size_t i = 0; i < nk; ++i) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
        for (int i = 0; i < nq; ++i) { if (out_min) out_min[i] = INFINITY; if (out_max) out_max[i] = -INFINITY; }
        return CMR_OK;
    }
    const auto tm0 = std::chrono::steady_clock::now();
    struct Tock { cmr_mindex* m; std::chrono::steady_clock::time_point tw0, tm0; ~Tock() {
        const auto now = std::chrono::steady_clock::now();
        m->prof_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(tm0 - tw0).count();
        m->prof_merge_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(now - tm0).count();
        ++m->prof_batches; } } tock{m, tw0, tm0};
    const char* base = (const char*)t->h;
    const int64_t* ids = (const int64_t*)base;
    const float* sc = (const float*)(base + (size_t)A * nk * 8);
    const float* mn = (const float*)(base + (size_t)A * nk * 12);
    const float* mx = mn + (size_t)A * nq;
    merge_sorted_lists(ids, sc, A, nq, k, out_ids, out_scores);
    for (int i = 0; i < nq; ++i) {
        float lo = INFINITY, hi = -INFINITY;
        for (int a = 0; a < A; ++a) { lo = std::fmin(lo, mn[(size_t)a * nq + i]); hi = std::fmax(hi, mx[(size_t)a * nq + i]); }
        if (out_min) out_min[i] = lo;
        if (out_max) out_max[i] = hi;
    }
    return CMR_OK;
}

int32_t cmr_mindex_profile(cmr_mindex_t* m, int32_t reset, int64_t* n_batches, double* enqueue_us_per_shard, double* wait_us_per_batch,
                           double* merge_us_per_batch) {
    if (!m) return cmr_fail(CMR_ERR_INVALID, "NULL index");
    const long long jobs = m->prof_jobs.load(), b = m->prof_batches.load();
    if (n_batches) *n_batches = b;
    if (enqueue_us_per_shard) *enqueue_us_per_shard = jobs ? m->prof_enqueue_ns.load() / 1e3 / jobs : 0.0;
    if (wait_us_per_batch) *wait_us_per_batch = b ? m->prof_wait_ns.load() / 1e3 / b : 0.0;
    if (merge_us_per_batch) *merge_us_per_batch = b ? m->prof_merge_ns.load() / 1e3 / b : 0.0;
    if (reset) { m->prof_jobs = 0; m->prof_enqueue_ns = 0; m->prof_batches = 0; m->prof_wait_ns = 0; m->prof_merge_ns = 0; }
    return CMR_OK;
}

}  // extern "C"
