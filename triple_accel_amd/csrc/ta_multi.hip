// ta_multi.hip -- the device set: the multi-GPU split BEHIND the C ABI (BASELINE.json north_star: "large batches of independent string
// pairs -- and for levenshtein_search the haystack itself -- are partitioned across the 8 GPUs of one node"), for the callers the
// reference has: host slices, one process (src/levenshtein.rs:714-720, 1911-1918, 2508-2511; src/hamming.rs:454-475).
//
// One host worker thread per entry of the device set, each with hipSetDevice(its device), its own stream, its own pinned staging ring and --
// because every piece of library state is thread-local (ta_api.hip: Scratch, CallCtx, PinBox) -- its own device scratch.  Pairs shard as
// contiguous ranges (no data-path collective; the host is where the answers meet), a haystack as contiguous byte ranges behind
// needle_len + unit_k + 2 bytes of left context (Levenshtein: a DP started fresh that far to the left is exact for every cost <= k) or in
// front of needle_len - 1 bytes of right overlap (Hamming: the windows that start in the shard).  No RCCL: the match lists and the 4 bytes
// per pair come back over each device's own PCIe link and are concatenated in shard order, which IS the global order.
// A device id may be listed more than once (that many workers share the device): what the one-GPU test box uses to run the N = 2, 3, 8 logic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ta_internal.h"

namespace ta {

// ---------------------------------------------------------------- pinned staging ring
// Host bytes travel through a ring of pinned slots: memcpy into a slot (the worker thread), hipMemcpyAsync out of it (the DMA engine), an
// event per slot says when it may be overwritten.  Downloads use the same ring the other way round: the copy into the caller's memory
// happens when the slot comes round again (or at drain()).
struct Ring {
    static constexpr int NS = 4;
    size_t piece = 0;
    uint8_t *pin[NS] = {};
    hipEvent_t ev[NS] = {};
    struct Pending { void *dst = nullptr; size_t bytes = 0; bool busy = false; } pend[NS];
    int next = 0;

    int ensure(size_t piece_bytes) {
        if (piece >= piece_bytes && pin[0]) return TA_OK;
        int rc = drain();
        if (rc) return rc;
        release();
        for (int i = 0; i < NS; i++) {
            void *p = nullptr;
            TA_HIP(hipHostMalloc(&p, piece_bytes, hipHostMallocDefault));
            pin[i] = (uint8_t *)p;
            TA_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        piece = piece_bytes;
        return TA_OK;
    }
    // the next slot, free to be written: waits for its last transfer; a finished download is delivered to its destination first
    int acquire(int *slot) {
        const int s = next;
        next = (next + 1) % NS;
        if (pend[s].busy) {
            TA_HIP(hipEventSynchronize(ev[s]));
            if (pend[s].dst) memcpy(pend[s].dst, pin[s], pend[s].bytes);
            pend[s] = Pending{};
        }
        *slot = s;
        return TA_OK;
    }
    int drain() {
        for (int i = 0; i < NS; i++) {
            const int s = (next + i) % NS;
            if (!pend[s].busy) continue;
            TA_HIP(hipEventSynchronize(ev[s]));
            if (pend[s].dst) memcpy(pend[s].dst, pin[s], pend[s].bytes);
            pend[s] = Pending{};
        }
        return TA_OK;
    }
    void release() {
        for (int i = 0; i < NS; i++) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            pin[i] = nullptr; ev[i] = nullptr; pend[i] = Pending{};
        }
        piece = 0;
    }
};

// grow-only device buffer owned by a worker (or by a resident handle, freed on that worker)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return TA_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }      // (hipFree waits for the device: nothing still reads the old buffer)
        const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        TA_HIP(hipMalloc(&p, want));
        cap = want;
        return TA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Worker {
    int device = 0, index = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool stop = false;
    // owned by the worker thread
    int init_rc = TA_OK;
    std::string init_err;
    hipStream_t st = nullptr;
    Ring up, down;
    DevBuf buf[9];      // 0/1: a blob / offsets, 2/3: b blob / offsets, 4: results, 5: hits, 6: a haystack shard of a host-entry search, 7/8: packed scripts / run counts (host batch tracebacks)

    void post(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(f)); }
        cv.notify_one();
    }
    void main_loop() {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            init_rc = TA_ERR_HIP;
            init_err = "device set: hipSetDevice(" + std::to_string(device) + ") failed";
        }
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) break;                         // (stop: only after the queue has drained)
                f = std::move(jobs.front());
                jobs.pop_front();
            }
            f();
        }
        if (init_rc == TA_OK) {                                   // leave nothing behind on the device
            (void)hipStreamSynchronize(st);
            up.release(); down.release();
            for (DevBuf &b : buf) b.release();
            ta_thread_release();
            (void)hipStreamDestroy(st);
        }
    }
};

// Worker threads are created on demand and NEVER exit: the k-th worker of device d serves the k-th occurrence of d in any device set.  A set
// (Pool) is a list of borrowed workers.  (Round 6's first form joined a set's threads when the set changed; a fuzz run that switched sets
// ~10,000 times died three times in five with "double free or corruption" inside that tear-down -- a worker's exit racing with the runtime's own
// per-thread state -- and a thread that never exits has no tear-down.  A worker that leaves the set frees what it holds: trim().)  At process
// exit the workers are blocked on their condition variables and die with the process.
struct Registry {
    std::mutex mu;
    std::vector<std::vector<Worker *>> by_device;            // leaked with the process
    Worker *get(int device, size_t occurrence) {
        std::lock_guard<std::mutex> lk(mu);
        if (by_device.size() <= (size_t)device) by_device.resize((size_t)device + 1);
        auto &v = by_device[(size_t)device];
        while (v.size() <= occurrence) {
            Worker *w = new Worker();
            w->device = device;
            w->index = (int)v.size();
            w->th = std::thread([w] { w->main_loop(); });
            w->th.detach();
            v.push_back(w);
        }
        return v[occurrence];
    }
    std::vector<Worker *> all() {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<Worker *> r;
        for (auto &v : by_device) r.insert(r.end(), v.begin(), v.end());
        return r;
    }
};
static Registry &registry() { static Registry *r = new Registry(); return *r; }

struct Pool {
    std::vector<Worker *> w;                                   // borrowed from the registry
    std::vector<Worker *> w2;                                  // a SECOND stager per entry, for big host batches: two threads feeding a device's link (two contiguous half slices, two streams) --
                                                               // 1M x 256-byte pairs end to end: 9.9 ms with one stager per device, 9.4 with two (through the ring: 17.9 / 11.8)
    std::vector<int> devices;
    explicit Pool(const std::vector<int> &devs) : devices(devs) {
        std::vector<size_t> total, seen;
        for (int d : devs) {
            if (total.size() <= (size_t)d) { total.resize((size_t)d + 1, 0); seen.resize((size_t)d + 1, 0); }
            total[(size_t)d]++;
        }
        for (int d : devs) {
            const size_t c = seen[(size_t)d]++;
            w.push_back(registry().get(d, c));
            w2.push_back(registry().get(d, total[(size_t)d] + c));
        }
    }
};

// The process's device set.
static std::mutex g_pool_mu;
static std::shared_ptr<Pool> *g_pool = nullptr;

static std::vector<int> default_devices() {
    std::vector<int> d;
    if (const char *e = env_str("TA_DEVICES")) {                  // TA_TUNING only: "0,0,0" = three workers on device 0
        for (const char *p = e; *p;) {
            char *end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p) break;
            d.push_back((int)v);
            p = (*end == ',') ? end + 1 : end;
        }
        if (!d.empty()) return d;
    }
    const int n = ta_device_count();
    for (int i = 0; i < n; i++) d.push_back(i);
    return d;
}

static std::shared_ptr<Pool> pool() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool) g_pool = new std::shared_ptr<Pool>();
    if (!*g_pool) {
        std::vector<int> d = default_devices();
        if (d.empty()) return nullptr;
        *g_pool = std::make_shared<Pool>(d);
    }
    return *g_pool;
}

// Runs fn(r, worker) on the first `n_use` workers at once and waits for all of them; the first failure's status and message are the call's.
// The calling thread's options (ta_set_option) travel with the jobs.
static int run_on(Pool &P, size_t n_use, const std::function<int(size_t, Worker &)> &fn) {
    // The latch and the result slots live on the HEAP and every job holds a reference: a worker's last touch of them (the notify) may come after
    // the caller has seen left == 0 and returned -- on the caller's stack that was a use after scope (found by scripts/r06/fuzz_r06.py: "double
    // free or corruption" once in ~11,000 rounds).  `fn` itself is only used before the job counts down.
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        size_t left;
        std::vector<int> rcs;
        std::vector<std::string> errs;
        explicit Shared(size_t n) : left(n), rcs(n, TA_OK), errs(n) {}
    };
    auto sh = std::make_shared<Shared>(n_use);
    const bool eo = early_out_enabled(), pf = unit_prefilter_enabled();
    for (size_t r = 0; r < n_use; r++) {
        Worker *w = P.w[r];
        w->post([sh, &fn, eo, pf, r, w] {
            int rc;
            std::string err;
            if (w->init_rc) { rc = w->init_rc; err = w->init_err; }
            else {
                ta_set_option(TA_OPT_EARLY_OUT, eo);
                ta_set_option(TA_OPT_UNIT_PREFILTER, pf);
                rc = fn(r, *w);
                if (rc) {
                    err = ta_last_error();
                    (void)hipStreamSynchronize(w->st);            // a failed job leaves nothing in flight behind it
                    (void)w->up.drain(); (void)w->down.drain();
                }
            }
            std::lock_guard<std::mutex> lk(sh->mu);                // (the notify under the lock: the caller cannot run ahead of it)
            sh->rcs[r] = rc; sh->errs[r] = std::move(err);
            sh->left--;
            sh->cv.notify_one();
        });
    }
    std::unique_lock<std::mutex> lk(sh->mu);
    sh->cv.wait(lk, [&] { return sh->left == 0; });
    for (size_t r = 0; r < n_use; r++)
        if (sh->rcs[r]) { set_last_error_msg(sh->errs[r].c_str()); return sh->rcs[r]; }
    return TA_OK;
}

static size_t tuning_size(const char *name, size_t dflt) {
    if (const char *e = env_str(name)) { long long v = atoll(e); if (v > 0) return (size_t)v; }
    return dflt;
}
static size_t piece_bytes() { return tuning_size("TA_MULTI_PIECE", 4u << 20); }

static void shard_range(size_t n, size_t r, size_t world, size_t *lo, size_t *hi) {   // dist.py: shard_range
    const size_t base = n / world, rem = n % world;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

// ---------------------------------------------------------------- transfers (worker thread)
static int upload(Worker &w, void *dst_dev, const uint8_t *src, size_t bytes) {
    // Big pieces of the caller's (pageable) memory go through the runtime's own path -- it pins the pages in place and lets the DMA engine read them:
    // 1M x 256-byte pairs from host memory on one GPU 9.9 ms end to end against 17.9 through the ring with one stager and 11.8 with two
    // (scripts/r06/ab_direct_copy.sh).  The ring stays for what is transformed on the way (CSR offsets), for small pieces and for downloads.
    // TA_MULTI_DIRECT_FROM=n: the threshold in bytes (65,536; the tests raise it so that the ring carries everything).
    if (bytes >= tuning_size("TA_MULTI_DIRECT_FROM", 65536)) {
        TA_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, w.st));
        return TA_OK;
    }
    int rc = w.up.ensure(piece_bytes());
    if (rc) return rc;
    for (size_t o = 0; o < bytes; o += w.up.piece) {
        const size_t nb = bytes - o < w.up.piece ? bytes - o : w.up.piece;
        int s;
        if ((rc = w.up.acquire(&s))) return rc;
        memcpy(w.up.pin[s], src + o, nb);
        TA_HIP(hipMemcpyAsync((uint8_t *)dst_dev + o, w.up.pin[s], nb, hipMemcpyHostToDevice, w.st));
        TA_HIP(hipEventRecord(w.up.ev[s], w.st));
        w.up.pend[s].busy = true;
    }
    return TA_OK;
}
// CSR offsets off[0..cnt] of a slice, rebased to 0 while they are staged; *max_len = the slice's longest string
static int upload_offsets(Worker &w, void *dst_dev, const uint64_t *off, size_t cnt, uint64_t *max_len) {
    int rc = w.up.ensure(piece_bytes());
    if (rc) return rc;
    const uint64_t base = off[0];
    uint64_t mx = 0;
    const size_t per = w.up.piece / 8, total = cnt + 1;
    for (size_t o = 0; o < total; o += per) {
        const size_t ne = total - o < per ? total - o : per;
        int s;
        if ((rc = w.up.acquire(&s))) return rc;
        uint64_t *p = (uint64_t *)w.up.pin[s];
        for (size_t i = 0; i < ne; i++) {
            p[i] = off[o + i] - base;
            if (o + i < cnt) { const uint64_t l = off[o + i + 1] - off[o + i]; if (l > mx) mx = l; }
        }
        TA_HIP(hipMemcpyAsync((uint8_t *)dst_dev + o * 8, p, ne * 8, hipMemcpyHostToDevice, w.st));
        TA_HIP(hipEventRecord(w.up.ev[s], w.st));
        w.up.pend[s].busy = true;
    }
    *max_len = mx;
    return TA_OK;
}
static int download(Worker &w, void *dst_host, const void *src_dev, size_t bytes) {
    int rc = w.down.ensure(1u << 20);
    if (rc) return rc;
    for (size_t o = 0; o < bytes; o += w.down.piece) {
        const size_t nb = bytes - o < w.down.piece ? bytes - o : w.down.piece;
        int s;
        if ((rc = w.down.acquire(&s))) return rc;
        TA_HIP(hipMemcpyAsync(w.down.pin[s], (const uint8_t *)src_dev + o, nb, hipMemcpyDeviceToHost, w.st));
        TA_HIP(hipEventRecord(w.down.ev[s], w.st));
        w.down.pend[s] = Ring::Pending{(uint8_t *)dst_host + o, nb, true};
    }
    return TA_OK;
}

// One side of the pairs [lo, hi) of a HOST batch onto the device: -> a ta_strings over device memory (CSR offsets rebased; max_len known)
static int stage_side(Worker &w, const ta_strings *h, size_t lo, size_t hi, DevBuf &blob, DevBuf &offs, ta_strings *out) {
    const size_t cnt = hi - lo;
    int rc;
    if (h->off) {
        const uint64_t b0 = h->off[lo], bytes = h->off[hi] - b0;
        if ((rc = blob.ensure((size_t)bytes + TA_BLOB_SLACK + 64)) || (rc = offs.ensure((cnt + 1) * 8))) return rc;
        uint64_t mx = 0;
        if ((rc = upload(w, blob.p, h->blob + b0, (size_t)bytes)) || (rc = upload_offsets(w, offs.p, h->off + lo, cnt, &mx))) return rc;
        *out = ta_strings{(const uint8_t *)blob.p, (const uint64_t *)offs.p, 0, 0, mx ? mx : 1};
        return TA_OK;
    }
    const uint64_t bytes = cnt ? (uint64_t)(cnt - 1) * h->stride + h->len : 0;
    if ((rc = blob.ensure((size_t)bytes + TA_BLOB_SLACK + 64))) return rc;
    if (bytes && (rc = upload(w, blob.p, h->blob + (uint64_t)lo * h->stride, (size_t)bytes))) return rc;
    *out = ta_strings{(const uint8_t *)blob.p, nullptr, h->stride, h->len, h->len};
    return TA_OK;
}

// end of the chunk that starts at pair lo: as many pairs as fit `max_bytes` of strings (both sides) and `max_pairs`, at least one
static size_t chunk_end(const ta_strings *a, const ta_strings *b, size_t lo, size_t hi, uint64_t max_bytes, size_t max_pairs) {
    auto bytes_upto = [&](size_t e) -> uint64_t {
        uint64_t t = 0;
        for (const ta_strings *s : {a, b}) t += s->off ? s->off[e] - s->off[lo] : (uint64_t)(e - lo) * (s->stride ? s->stride : 1);
        return t;
    };
    size_t l = lo + 1, r = hi - lo > max_pairs ? lo + max_pairs : hi;         // the answer lies in [l, r]
    if (bytes_upto(r) <= max_bytes) return r;
    while (l < r) {                                                          // largest e with bytes_upto(e) <= max_bytes (monotone)
        const size_t m = l + (r - l + 1) / 2;
        if (bytes_upto(m) <= max_bytes) l = m; else r = m - 1;
    }
    return l;
}

enum PairOp { OP_LEV_K = 0, OP_LEV_EXP = 1, OP_HAMMING = 2, OP_TRACE = 3 };
static int pair_op(PairOp op, const ta_strings *A, const ta_strings *B, size_t n, uint32_t k, const ta_edit_costs *costs, uint32_t *out_dev, hipStream_t st) {
    switch (op) {
        case OP_LEV_K: return ta_levenshtein_k_batch(A, B, n, k, costs, out_dev, st);
        case OP_LEV_EXP: return ta_levenshtein_exp_batch(A, B, n, costs, out_dev, st);
        case OP_HAMMING: return ta_hamming_batch(A, B, n, out_dev, st);
        default: return TA_ERR_ARG;
    }
}

static bool host_strings_ok(const ta_strings *s, size_t n) {
    if (!s) return false;
    if (n && !s->blob && (s->off ? s->off[n] != s->off[0] : s->len != 0)) return false;
    return true;
}

// how many workers a batch of n pairs is spread over: no device gets fewer than TA_MULTI_MIN_PAIRS (4,096: a pass of fewer pairs costs a
// wavefront's lifetime whatever their number)
static size_t pair_shards(const Pool &P, size_t n) {
    const size_t min_pairs = tuning_size("TA_MULTI_MIN_PAIRS", 4096);
    size_t use = n / min_pairs;
    if (use < 1) use = 1;
    if (use > P.w.size()) use = P.w.size();
    return use;
}

// OP_TRACE: additionally packed (n x cap words, scripts right-aligned as ta_levenshtein_trace_batch_packed leaves them) and n_edits (n), host memory
static int pairs_host(PairOp op, const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs, uint32_t *out,
                      uint32_t *packed = nullptr, uint32_t *n_edits = nullptr, size_t cap = 0) {
    if (!host_strings_ok(a, n) || !host_strings_ok(b, n) || (!out && n) || n > 0xFFFFFFF0ull) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    if (op == OP_TRACE && n && (!packed || !n_edits || cap == 0 || cap > 0xFFFFFFFFull)) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    if (op != OP_HAMMING && (!costs || ta_edit_costs_new(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose, costs->transpose_cost, nullptr) != TA_OK))
        return TA_ERR_BAD_COSTS;
    if (!device_ready()) return TA_ERR_HIP;
    if (n == 0) return TA_OK;
    std::shared_ptr<Pool> P = pool();
    if (!P) return TA_ERR_HIP;
    const size_t devs_used = pair_shards(*P, n);
    const uint64_t chunk_bytes = tuning_size("TA_MULTI_CHUNK_BYTES", 64u << 20);
    size_t chunk_pairs = tuning_size("TA_MULTI_CHUNK_PAIRS", 1u << 20);
    if (op == OP_TRACE && chunk_pairs * cap * 4 > (256u << 20)) chunk_pairs = (256u << 20) / (cap * 4) ? (256u << 20) / (cap * 4) : 1;   // (a chunk's scripts: <= 256 MiB)
    const ta_edit_costs c = costs ? *costs : ta_edit_costs{1, 1, 0, 0, 0};
    // a device whose slice holds >= 16 MiB of strings gets two stagers (two contiguous half slices, two streams on the device): TA_MULTI_STAGERS=1 keeps one
    uint64_t total_bytes = 0;
    for (const ta_strings *sd : {a, b}) total_bytes += sd->off ? sd->off[n] - sd->off[0] : (uint64_t)n * (sd->stride ? sd->stride : sd->len);
    const bool two = tuning_size("TA_MULTI_STAGERS", 2) >= 2 && total_bytes / devs_used >= tuning_size("TA_MULTI_STAGERS_FROM", 16u << 20) && n / devs_used >= 2;
    Pool stagers(std::vector<int>{});
    for (size_t r = 0; r < devs_used; r++) {
        stagers.w.push_back(P->w[r]);
        if (two) stagers.w.push_back(P->w2[r]);
    }
    const size_t use = stagers.w.size();
    return run_on(stagers, use, [&](size_t r, Worker &w) -> int {
        size_t lo, hi;
        shard_range(n, r, use, &lo, &hi);
        int rc;
        for (size_t c_lo = lo; c_lo < hi;) {
            const size_t c_hi = chunk_end(a, b, c_lo, hi, chunk_bytes, chunk_pairs), cnt = c_hi - c_lo;
            ta_strings A, B;
            if ((rc = stage_side(w, a, c_lo, c_hi, w.buf[0], w.buf[1], &A)) || (rc = stage_side(w, b, c_lo, c_hi, w.buf[2], w.buf[3], &B)) ||
                (rc = w.buf[4].ensure(cnt * 4)))
                return rc;
            if (op == OP_TRACE) {
                // the scripts as packed runs (4 bytes each): cap words per pair come back -- words in front of a script are whatever the buffer held
                if ((rc = w.buf[7].ensure(cnt * cap * 4)) || (rc = w.buf[8].ensure(cnt * 4))) return rc;
                if ((rc = ta_levenshtein_trace_batch_packed(&A, &B, cnt, k, &c, (uint32_t *)w.buf[4].p, (uint32_t *)w.buf[7].p, (uint32_t *)w.buf[8].p, cap, w.st))) return rc;
                if ((rc = download(w, packed + c_lo * cap, w.buf[7].p, cnt * cap * 4)) || (rc = download(w, n_edits + c_lo, w.buf[8].p, cnt * 4))) return rc;
            } else if ((rc = pair_op(op, &A, &B, cnt, k, &c, (uint32_t *)w.buf[4].p, w.st))) return rc;
            if ((rc = download(w, out + c_lo, w.buf[4].p, cnt * 4))) return rc;
            c_lo = c_hi;
        }
        return w.down.drain();
    });
}

}  // namespace ta

using namespace ta;

// ---------------------------------------------------------------- resident handles
struct ta_sharded_pairs {
    std::shared_ptr<Pool> pool;
    size_t n = 0;
    struct Shard { size_t lo = 0, hi = 0; DevBuf blob[2], offs[2], out; ta_strings A = {}, B = {}; };
    std::vector<Shard> shards;
};

struct ta_sharded_haystack {
    std::shared_ptr<Pool> pool;
    size_t len = 0, overlap = 0;
    // shard r holds the bytes [lo - ctx, hi + tail) of the haystack
    struct Shard { size_t lo = 0, hi = 0, ctx = 0, tail = 0; DevBuf bytes; };
    std::vector<Shard> shards;
};

namespace ta {

struct HayView { const uint8_t *dev; size_t lo, hi, ctx, tail; };

// All-mode hits (or, best: the hits with the shard's smallest k -- the only ones the fold can keep) of the end positions (lo, hi], sorted by end
static int lev_search_shard(Worker &w, const HayView &v, const uint8_t *needle, size_t n, uint32_t k, const ta_edit_costs *costs, bool best,
                            std::vector<ta_match> &hits) {
    hits.clear();
    const size_t len = v.ctx + (v.hi - v.lo);
    if (v.hi == v.lo) return TA_OK;
    size_t cap = len + 2;
    if (cap > (1u << 22)) cap = 1u << 22;
    int rc;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = w.buf[5].ensure(cap * sizeof(ta_match)))) return rc;
        uint64_t count = 0;
        if (best) {
            ta_match *sel = nullptr;
            size_t n_sel = 0;
            rc = ta_levenshtein_search_best_dev(needle, n, v.dev, len, k, costs, v.lo - v.ctx, v.lo, (ta_match *)w.buf[5].p, cap, &count, &sel, &n_sel, w.st);
            if (rc == TA_OK) { hits.assign(sel, sel + n_sel); free(sel); return TA_OK; }
        } else {
            rc = ta_levenshtein_search_dev(needle, n, v.dev, len, k, costs, 0, v.lo - v.ctx, v.lo, (ta_match *)w.buf[5].p, cap, &count, w.st);
            if (rc == TA_OK) {
                hits.resize((size_t)count);
                if (count) {
                    if ((rc = download(w, hits.data(), w.buf[5].p, (size_t)count * sizeof(ta_match))) || (rc = w.down.drain())) return rc;
                    std::sort(hits.begin(), hits.end(), [](const ta_match &x, const ta_match &y) { return x.end != y.end ? x.end < y.end : x.start < y.start; });
                }
                return TA_OK;
            }
        }
        if (rc != TA_ERR_CAPACITY || count <= cap || count > (uint64_t)len + 2) return rc;
        cap = (size_t)count;                                       // a denser result: once more with room for exactly that
    }
    return rc;
}

int hamming_search_dev_nocheck(const uint8_t *needle_host, size_t needle_len, const uint8_t *haystack_dev, size_t haystack_len, uint32_t k,
                               uint64_t base, ta_match *hits_dev, size_t cap, uint64_t *count_host, void *stream);

// the windows that START in [lo, hi): the shard's bytes and needle_len - 1 more
static int ham_search_shard(Worker &w, const HayView &v, size_t hay_len, const uint8_t *needle, size_t n, uint32_t k, bool check_nul,
                            std::vector<ta_match> &hits) {
    hits.clear();
    size_t end = v.hi + n - 1;
    if (end > hay_len) end = hay_len;
    if (end > v.hi + v.tail) { set_last_error_msg("sharded haystack: the needle is longer than the overlap the shards were uploaded with"); return TA_ERR_ARG; }
    const size_t len = end - v.lo;
    if (v.hi == v.lo) return TA_OK;
    // fewer bytes than the needle: no window starts here, and the NUL rule (src/hamming.rs:463) has seen these bytes already -- len < n means
    // hay_len <= lo + n - 1, which is where the overlap of the nearest earlier shard that does hold a window ends
    if (len < n) return TA_OK;
    size_t cap = len + 2;
    if (cap > (1u << 22)) cap = 1u << 22;
    int rc = TA_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = w.buf[5].ensure(cap * sizeof(ta_match)))) return rc;
        uint64_t count = 0;
        rc = check_nul ? ta_hamming_search_dev(needle, n, v.dev + v.ctx, len, k, v.lo, (ta_match *)w.buf[5].p, cap, &count, w.st)
                       : hamming_search_dev_nocheck(needle, n, v.dev + v.ctx, len, k, v.lo, (ta_match *)w.buf[5].p, cap, &count, w.st);
        if (rc == TA_OK) {
            hits.resize((size_t)count);
            if (count) {
                if ((rc = download(w, hits.data(), w.buf[5].p, (size_t)count * sizeof(ta_match))) || (rc = w.down.drain())) return rc;
                std::sort(hits.begin(), hits.end(), [](const ta_match &x, const ta_match &y) { return x.end != y.end ? x.end < y.end : x.start < y.start; });
            }
            return TA_OK;
        }
        if (rc != TA_ERR_CAPACITY || count <= cap || count > (uint64_t)len + 2) return rc;
        cap = (size_t)count;
    }
    return rc;
}

// how many workers a haystack of h bytes is spread over (host entries: no device gets less than TA_MULTI_MIN_HAY bytes, 4 MiB)
size_t multi_search_shards(size_t h) {
    if (h < tuning_size("TA_MULTI_MIN_HAY", 4u << 20) * 2) return 1;
    std::shared_ptr<Pool> P = pool();
    if (!P || P->w.size() < 2) return 1;
    size_t use = h / tuning_size("TA_MULTI_MIN_HAY", 4u << 20);
    if (use > P->w.size()) use = P->w.size();
    return use < 1 ? 1 : use;
}

static uint32_t lev_halo(size_t needle_len, uint32_t k, const ta_edit_costs *c) {
    const uint32_t unit_k = lev_sat_sub(k, c->start_gap_cost) / c->gap_cost;
    const uint64_t h = (uint64_t)needle_len + unit_k + 2;
    return h > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)h;
}

// host entries: upload shard by shard (each worker its own slice of the caller's haystack, behind / in front of its overlap), search, concatenate
int multi_levenshtein_search_host(const uint8_t *needle, size_t n, const uint8_t *hay, size_t h, uint32_t k, bool best, const ta_edit_costs *costs,
                                  std::vector<ta_match> &hits) {
    std::shared_ptr<Pool> P = pool();
    if (!P) return TA_ERR_HIP;
    const size_t use = multi_search_shards(h);
    const size_t halo = lev_halo(n, k, costs);
    std::vector<std::vector<ta_match>> part(use);
    int rc = run_on(*P, use, [&](size_t r, Worker &w) -> int {
        HayView v;
        shard_range(h, r, use, &v.lo, &v.hi);
        v.ctx = v.lo < halo ? v.lo : halo;
        v.tail = 0;
        const size_t bytes = v.ctx + (v.hi - v.lo);
        int rc2;
        if ((rc2 = w.buf[6].ensure(bytes + TA_BLOB_SLACK + 64)) || (rc2 = upload(w, w.buf[6].p, hay + (v.lo - v.ctx), bytes))) return rc2;
        v.dev = (const uint8_t *)w.buf[6].p;
        return lev_search_shard(w, v, needle, n, k, costs, best, part[r]);
    });
    if (rc) return rc;
    hits.clear();
    for (auto &p : part) hits.insert(hits.end(), p.begin(), p.end());
    return TA_OK;
}

int multi_hamming_search_host(const uint8_t *needle, size_t n, const uint8_t *hay, size_t h, uint32_t k, bool check_nul, std::vector<ta_match> &hits) {
    std::shared_ptr<Pool> P = pool();
    if (!P) return TA_ERR_HIP;
    const size_t use = multi_search_shards(h);
    std::vector<std::vector<ta_match>> part(use);
    int rc = run_on(*P, use, [&](size_t r, Worker &w) -> int {
        HayView v;
        shard_range(h, r, use, &v.lo, &v.hi);
        v.ctx = 0;
        v.tail = h - v.hi < n - 1 ? h - v.hi : n - 1;
        const size_t bytes = (v.hi - v.lo) + v.tail;
        int rc2;
        if ((rc2 = w.buf[6].ensure(bytes + TA_BLOB_SLACK + 64)) || (rc2 = upload(w, w.buf[6].p, hay + v.lo, bytes))) return rc2;
        v.dev = (const uint8_t *)w.buf[6].p;
        return ham_search_shard(w, v, h, needle, n, k, check_nul, part[r]);
    });
    if (rc) return rc;
    hits.clear();
    for (auto &p : part) hits.insert(hits.end(), p.begin(), p.end());
    return TA_OK;
}

size_t multi_pair_shards(size_t n) {
    std::shared_ptr<Pool> P = pool();
    return P ? pair_shards(*P, n) : 1;
}

}  // namespace ta

static int give_matches(std::vector<ta_match> &v, ta_match **out, size_t *n_out) {
    *n_out = v.size();
    *out = nullptr;
    if (!v.empty()) {
        *out = (ta_match *)malloc(v.size() * sizeof(ta_match));
        if (!*out) return TA_ERR_ARG;
        memcpy(*out, v.data(), v.size() * sizeof(ta_match));
    }
    return TA_OK;
}

extern "C" {

// (debugging aid, not part of the ABI header: a native backtrace of the thread that aborts -- glibc's heap checks, an assert -- on stderr)
static void ta_abort_backtrace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "[triple_accel_amd] SIGABRT backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
void ta_debug_install_abort_backtrace(void) { signal(SIGABRT, ta_abort_backtrace); }

int ta_set_devices(const int *devices, size_t n) {
    std::vector<int> d;
    const int visible = ta_device_count();
    if (n == 0 || !devices) {
        for (int i = 0; i < visible; i++) d.push_back(i);
    } else {
        if (n > 1024) { set_last_error_msg("device set: more than 1024 entries"); return TA_ERR_ARG; }
        for (size_t i = 0; i < n; i++) {
            if (devices[i] < 0 || devices[i] >= visible) { set_last_error_msg("device set: no such device"); return TA_ERR_ARG; }
            d.push_back(devices[i]);
        }
    }
    if (d.empty()) { (void)device_ready(); return TA_ERR_HIP; }
    std::shared_ptr<Pool> old;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool) g_pool = new std::shared_ptr<Pool>();
        if (*g_pool && (*g_pool)->devices == d) return TA_OK;
        old = *g_pool;
        *g_pool = std::make_shared<Pool>(d);
    }
    // workers that are not part of the new set give back what they hold (pinned rings, device buffers, thread-local scratch); they stay alive
    // -- a resident handle created on them keeps working (its buffers are its own, the rings come back on demand)
    std::shared_ptr<Pool> now = pool();
    std::vector<Worker *> idle;
    for (Worker *w : registry().all())
        if (std::find(now->w.begin(), now->w.end(), w) == now->w.end() && std::find(now->w2.begin(), now->w2.end(), w) == now->w2.end()) idle.push_back(w);
    if (!idle.empty()) {
        Pool tmp(std::vector<int>{});
        tmp.w = idle;
        (void)run_on(tmp, idle.size(), [](size_t, Worker &w) -> int {
            (void)hipStreamSynchronize(w.st);
            w.up.release(); w.down.release();
            for (DevBuf &b : w.buf) b.release();
            ta_thread_release();
            return TA_OK;
        });
    }
    return TA_OK;
}

int ta_get_devices(int *out, size_t cap, size_t *n_out) {
    if (!n_out) return TA_ERR_ARG;
    std::shared_ptr<Pool> P = pool();
    *n_out = P ? P->devices.size() : 0;
    if (P && out)
        for (size_t i = 0; i < P->devices.size() && i < cap; i++) out[i] = P->devices[i];
    return TA_OK;
}

int ta_levenshtein_k_batch_host(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs, uint32_t *out) {
    return pairs_host(OP_LEV_K, a, b, n, k, costs, out);
}
int ta_levenshtein_exp_batch_host(const ta_strings *a, const ta_strings *b, size_t n, const ta_edit_costs *costs, uint32_t *out) {
    return pairs_host(OP_LEV_EXP, a, b, n, 0, costs, out);
}
int ta_hamming_batch_host(const ta_strings *a, const ta_strings *b, size_t n, uint32_t *out) {
    return pairs_host(OP_HAMMING, a, b, n, 0, nullptr, out);
}
int ta_levenshtein_trace_batch_host(const ta_strings *a, const ta_strings *b, size_t n, uint32_t k, const ta_edit_costs *costs,
                                    uint32_t *out, uint32_t *packed, uint32_t *n_edits, size_t cap) {
    return pairs_host(OP_TRACE, a, b, n, k, costs, out, packed, n_edits, cap);
}

/* ---- a pair batch kept RESIDENT, sharded over the device set */
int ta_sharded_pairs_upload(const ta_strings *a, const ta_strings *b, size_t n, size_t n_shards, ta_sharded_pairs **out) {
    if (!out) return TA_ERR_ARG;
    *out = nullptr;
    if (!host_strings_ok(a, n) || !host_strings_ok(b, n) || n > 0xFFFFFFF0ull) { set_last_error_msg("bad batch arguments"); return TA_ERR_ARG; }
    if (!device_ready()) return TA_ERR_HIP;
    std::shared_ptr<Pool> P = pool();
    if (!P) return TA_ERR_HIP;
    size_t use = n_shards ? n_shards : pair_shards(*P, n);
    if (use > P->w.size()) use = P->w.size();
    ta_sharded_pairs *S = new ta_sharded_pairs();
    S->pool = P; S->n = n; S->shards.resize(use);
    int rc = run_on(*P, use, [&](size_t r, Worker &w) -> int {
        ta_sharded_pairs::Shard &sh = S->shards[r];
        shard_range(n, r, use, &sh.lo, &sh.hi);
        int rc2;
        if ((rc2 = stage_side(w, a, sh.lo, sh.hi, sh.blob[0], sh.offs[0], &sh.A)) || (rc2 = stage_side(w, b, sh.lo, sh.hi, sh.blob[1], sh.offs[1], &sh.B)) ||
            (rc2 = sh.out.ensure((sh.hi - sh.lo) * 4 + 4)))
            return rc2;
        TA_HIP(hipStreamSynchronize(w.st));
        return TA_OK;
    });
    if (rc) { ta_sharded_pairs_free(S); return rc; }
    *out = S;
    return TA_OK;
}

static int sharded_pairs_run(ta_sharded_pairs *S, PairOp op, uint32_t k, const ta_edit_costs *costs, uint32_t *out, int download_results) {
    if (!S || (!out && download_results && S->n)) return TA_ERR_ARG;
    if (op != OP_HAMMING && (!costs || ta_edit_costs_new(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose, costs->transpose_cost, nullptr) != TA_OK))
        return TA_ERR_BAD_COSTS;
    const ta_edit_costs c = costs ? *costs : ta_edit_costs{1, 1, 0, 0, 0};
    return run_on(*S->pool, S->shards.size(), [&](size_t r, Worker &w) -> int {
        ta_sharded_pairs::Shard &sh = S->shards[r];
        const size_t cnt = sh.hi - sh.lo;
        if (!cnt) return TA_OK;
        int rc = pair_op(op, &sh.A, &sh.B, cnt, k, &c, (uint32_t *)sh.out.p, w.st);
        if (rc) return rc;
        if (download_results) {
            if ((rc = download(w, out + sh.lo, sh.out.p, cnt * 4))) return rc;
            return w.down.drain();
        }
        TA_HIP(hipStreamSynchronize(w.st));
        return TA_OK;
    });
}
int ta_sharded_pairs_levenshtein_k(ta_sharded_pairs *s, uint32_t k, const ta_edit_costs *costs, uint32_t *out) { return sharded_pairs_run(s, OP_LEV_K, k, costs, out, 1); }
int ta_sharded_pairs_levenshtein_exp(ta_sharded_pairs *s, const ta_edit_costs *costs, uint32_t *out) { return sharded_pairs_run(s, OP_LEV_EXP, 0, costs, out, 1); }
int ta_sharded_pairs_hamming(ta_sharded_pairs *s, uint32_t *out) { return sharded_pairs_run(s, OP_HAMMING, 0, nullptr, out, 1); }

/* `steps` passes of levenshtein_simd_k_with_opts over the resident batch, back to back on every device, the answers left on the devices:
 * *device_ms = the slowest shard's device time (HIP events on its worker's stream) -- what bench.py --single-process reports */
int ta_sharded_pairs_time_levenshtein_k(ta_sharded_pairs *S, uint32_t k, const ta_edit_costs *costs, int steps, float *device_ms) {
    if (!S || !device_ms || steps < 1) return TA_ERR_ARG;
    if (!costs) return TA_ERR_BAD_COSTS;
    std::vector<float> ms(S->shards.size(), 0.f);
    int rc = run_on(*S->pool, S->shards.size(), [&](size_t r, Worker &w) -> int {
        ta_sharded_pairs::Shard &sh = S->shards[r];
        const size_t cnt = sh.hi - sh.lo;
        if (!cnt) return TA_OK;
        hipEvent_t e0, e1;
        TA_HIP(hipEventCreate(&e0));
        TA_HIP(hipEventCreate(&e1));
        TA_HIP(hipEventRecord(e0, w.st));
        int rc2 = TA_OK;
        for (int s = 0; s < steps && !rc2; s++) rc2 = ta_levenshtein_k_batch(&sh.A, &sh.B, cnt, k, costs, (uint32_t *)sh.out.p, w.st);
        if (!rc2) {
            TA_HIP(hipEventRecord(e1, w.st));
            TA_HIP(hipEventSynchronize(e1));
            TA_HIP(hipEventElapsedTime(&ms[r], e0, e1));
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return rc2;
    });
    if (rc) return rc;
    *device_ms = *std::max_element(ms.begin(), ms.end());
    return TA_OK;
}

int ta_sharded_pairs_shards(const ta_sharded_pairs *s, size_t *n_shards, size_t *n_pairs) {
    if (!s) return TA_ERR_ARG;
    if (n_shards) *n_shards = s->shards.size();
    if (n_pairs) *n_pairs = s->n;
    return TA_OK;
}

void ta_sharded_pairs_free(ta_sharded_pairs *S) {
    if (!S) return;
    (void)run_on(*S->pool, S->shards.size(), [&](size_t r, Worker &w) -> int {
        (void)hipStreamSynchronize(w.st);
        ta_sharded_pairs::Shard &sh = S->shards[r];
        for (int i = 0; i < 2; i++) { sh.blob[i].release(); sh.offs[i].release(); }
        sh.out.release();
        return TA_OK;
    });
    delete S;
}

/* ---- a haystack kept RESIDENT, sharded over the device set (BASELINE config 5: one shard per GPU) */
int ta_sharded_haystack_upload(const uint8_t *haystack, size_t len, size_t overlap, size_t n_shards, ta_sharded_haystack **out) {
    if (!out || (!haystack && len)) return TA_ERR_ARG;
    *out = nullptr;
    if (!device_ready()) return TA_ERR_HIP;
    std::shared_ptr<Pool> P = pool();
    if (!P) return TA_ERR_HIP;
    size_t use = n_shards ? n_shards : P->w.size();
    if (use > P->w.size()) use = P->w.size();
    if (use < 1) use = 1;
    ta_sharded_haystack *H = new ta_sharded_haystack();
    H->pool = P; H->len = len; H->overlap = overlap; H->shards.resize(use);
    int rc = run_on(*P, use, [&](size_t r, Worker &w) -> int {
        ta_sharded_haystack::Shard &sh = H->shards[r];
        shard_range(len, r, use, &sh.lo, &sh.hi);
        sh.ctx = sh.lo < overlap ? sh.lo : overlap;
        sh.tail = len - sh.hi < overlap ? len - sh.hi : overlap;
        const size_t bytes = sh.ctx + (sh.hi - sh.lo) + sh.tail;
        int rc2;
        if ((rc2 = sh.bytes.ensure(bytes + TA_BLOB_SLACK + 64))) return rc2;
        if (bytes && (rc2 = upload(w, sh.bytes.p, haystack + (sh.lo - sh.ctx), bytes))) return rc2;
        TA_HIP(hipStreamSynchronize(w.st));
        return TA_OK;
    });
    if (rc) { ta_sharded_haystack_free(H); return rc; }
    *out = H;
    return TA_OK;
}

int ta_sharded_haystack_levenshtein_search(ta_sharded_haystack *H, const uint8_t *needle, size_t needle_len, uint32_t k, int search_type,
                                           const ta_edit_costs *costs, ta_match **out, size_t *n_out) {
    if (!H || !out || !n_out || (!needle && needle_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (!costs || ta_edit_costs_new(costs->mismatch_cost, costs->gap_cost, costs->start_gap_cost, costs->has_transpose, costs->transpose_cost, nullptr) != TA_OK)
        return TA_ERR_BAD_COSTS;
    if (needle_len == 0) return TA_OK;                                          // unanchored: src/levenshtein.rs:1641-1643
    if (ta_edit_costs_check_search(costs) != TA_OK) return TA_ERR_BAD_COSTS;     // :1965
    if (needle_len > 65535) { set_last_error_msg("needle longer than 65535 bytes"); return TA_ERR_ARG; }
    const size_t halo = lev_halo(needle_len, k, costs);
    if (halo > H->overlap && H->shards.size() > 1) {
        set_last_error_msg("sharded haystack: needle_len + unit_k + 2 exceeds the overlap the shards were uploaded with");
        return TA_ERR_ARG;
    }
    const bool best = search_type == TA_SEARCH_BEST;
    std::vector<std::vector<ta_match>> part(H->shards.size());
    int rc = run_on(*H->pool, H->shards.size(), [&](size_t r, Worker &w) -> int {
        const ta_sharded_haystack::Shard &sh = H->shards[r];
        // the shard's own bytes behind `halo` bytes of left context (of the sh.ctx uploaded)
        const size_t ctx = sh.ctx < halo ? sh.ctx : halo;
        HayView v{(const uint8_t *)sh.bytes.p + (sh.ctx - ctx), sh.lo, sh.hi, ctx, 0};
        return lev_search_shard(w, v, needle, needle_len, k, costs, best, part[r]);
    });
    if (rc) return rc;
    std::vector<ta_match> res;
    const uint32_t whole_gap = (uint32_t)needle_len * costs->gap_cost + costs->start_gap_cost;   // the end == 0 match (:1693-1706)
    if (whole_gap <= k) res.push_back(ta_match{0, 0, whole_gap, 0});
    for (auto &p : part) res.insert(res.end(), p.begin(), p.end());
    if (best) res.resize(ta_search_fold_best(res.data(), res.size(), k, 1));
    return give_matches(res, out, n_out);
}

int ta_sharded_haystack_hamming_search(ta_sharded_haystack *H, const uint8_t *needle, size_t needle_len, uint32_t k, int search_type,
                                       ta_match **out, size_t *n_out) {
    if (!H || !out || !n_out || (!needle && needle_len)) return TA_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (needle_len > H->len || needle_len == 0) return TA_OK;                   // src/hamming.rs:455-461
    std::vector<std::vector<ta_match>> part(H->shards.size());
    int rc = run_on(*H->pool, H->shards.size(), [&](size_t r, Worker &w) -> int {
        const ta_sharded_haystack::Shard &sh = H->shards[r];
        HayView v{(const uint8_t *)sh.bytes.p, sh.lo, sh.hi, sh.ctx, sh.tail};
        return ham_search_shard(w, v, H->len, needle, needle_len, k, true, part[r]);
    });
    if (rc) return rc;
    std::vector<ta_match> res;
    for (auto &p : part) res.insert(res.end(), p.begin(), p.end());
    if (search_type == TA_SEARCH_BEST) res.resize(ta_search_fold_best(res.data(), res.size(), k, 0));
    return give_matches(res, out, n_out);
}

int ta_sharded_haystack_shards(const ta_sharded_haystack *h, size_t *n_shards, size_t *len) {
    if (!h) return TA_ERR_ARG;
    if (n_shards) *n_shards = h->shards.size();
    if (len) *len = h->len;
    return TA_OK;
}

void ta_sharded_haystack_free(ta_sharded_haystack *H) {
    if (!H) return;
    (void)run_on(*H->pool, H->shards.size(), [&](size_t r, Worker &w) -> int {
        (void)hipStreamSynchronize(w.st);
        H->shards[r].bytes.release();
        return TA_OK;
    });
    delete H;
}

}  // extern "C"
