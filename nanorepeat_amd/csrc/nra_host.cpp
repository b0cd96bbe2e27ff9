// nra_host.cpp -- host side of the C ABI (include/nanorepeat_amd.h): encoding, 2-bit
// packing, task bucketing, kernel sequencing on a private HIP stream, HIP-event timing.
//
// Replaces, in the reference, the per-read `pymm2.main(...)` loop of round3_align
// (nanoRepeat_bam.py:452-500) + its selector (:408-450), and the per-grid-cell loop +
// selector of the joint rounds (nanoRepeat_joint.py:315-347, 397-421, 427-478).
// No CPU fallback exists: without a HIP device every compute entry point fails.
#include "nanorepeat_amd.h"
#include "nra_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define NRA_VERSION_STR "nanorepeat_amd 0.1.0 (gfx950)"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? NRA_E_NOMEM : NRA_E_DEVICE,                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// NRA_DEBUG=1 in the environment: synchronise after every launch and trace it on stderr
static const bool g_debug = getenv("NRA_DEBUG") != nullptr;
// NRA_DEBUG_PHASES=1: the phase marks alone (no synchronisation: the times are the host's)
static const bool g_debug_phases = g_debug || getenv("NRA_DEBUG_PHASES") != nullptr;

#define LAUNCH_TRY(expr)                                                                         \
    do {                                                                                         \
        if (g_debug) fprintf(stderr, "[nra] launch %.60s ...\n", #expr);                         \
        int e_ = (expr);                                                                         \
        if (e_ != 0)                                                                             \
            return fail(NRA_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_)); \
        if (g_debug) {                                                                           \
            hipError_t s_ = hipDeviceSynchronize();                                              \
            fprintf(stderr, "[nra]   done: %s\n", hipGetErrorString(s_));                        \
        }                                                                                        \
    } while (0)

// NRA_DEBUG: wall time of the host phases of a create call
struct PhaseClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!g_debug_phases) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nra]   %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

const int kRList[] = {
#define X(r) r,
    NRA_R_LIST(X)
#undef X
};
const int kNumR = (int)(sizeof(kRList) / sizeof(kRList[0]));

int rows_for_qlen(int qlen)   // index into kRList, -1 if too long
{
    for (int i = 0; i < kNumR; ++i)
        if (64 * kRList[i] >= qlen) return i;
    return -1;
}

// ASCII -> code table for the bulk encoder (same mapping as encode_base)
struct BaseCodeTable {
    uint8_t v[256];
    BaseCodeTable()
    {
        for (int i = 0; i < 256; ++i) v[i] = NRA_CODE_N;
        v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2;
        v['T'] = v['t'] = v['U'] = v['u'] = 3;
    }
    uint8_t operator[](uint8_t c) const { return v[c]; }
};
static const BaseCodeTable kBaseCode;

inline uint8_t encode_base(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return NRA_CODE_N;
    }
}

// Device memory of one call or batch: hipMalloc costs ~0.3 ms per call, a batch has ~25 buffers, so
// they are carved out of a few large chunks.  While an Arena is "current" on this thread, DevBuf
// allocations come from it (and are released with it); otherwise a DevBuf owns its allocation.
// chunks come from / go back to a small per-device cache (HandlePool below)
hipError_t device_chunk_get(int device, size_t bytes, char** out, size_t* got);
void device_chunk_put(int device, char* p, size_t bytes);
// Copies between pageable host memory and the device (defined behind HandlePool): from kStagedFrom bytes on they go
// through a pinned buffer of the calling thread instead of straight into hipMemcpy.
hipError_t copy_h2d(void* dst, const void* src, size_t bytes);
hipError_t copy_d2h(void* dst, const void* src, size_t bytes);
// Between upload_batch_begin() and upload_batch_end() the calling thread's copy_h2d calls of any size only copy into the
// pinned stage and enqueue the transfer; _end waits for all of them once.  (A create call makes some sixteen uploads: as
// synchronous copies ~15 us each, a sixth of a small call's time.)  Nothing may use the destinations before _end.
void upload_batch_begin();
hipError_t upload_batch_end();
struct UploadBatchScope {            // ends the batch on every way out of a function; the normal path calls upload_batch_end() itself
    UploadBatchScope() { upload_batch_begin(); }
    ~UploadBatchScope() { (void)upload_batch_end(); }
};

struct Arena {
    struct Chunk { char* p; size_t size, used; };
    std::vector<Chunk> chunks;
    size_t next_chunk = 1 << 20;
    int device = 0;            // set before the first allocation; the work using the chunks is over when it dies
    ~Arena() { for (Chunk& c : chunks) device_chunk_put(device, c.p, c.size); }
    void expect(size_t bytes) { next_chunk = std::max(next_chunk, bytes); }
    void reset() { for (Chunk& c : chunks) c.used = 0; }        // keep the memory, hand it out again
    void release() { for (Chunk& c : chunks) device_chunk_put(device, c.p, c.size); chunks.clear(); }    // give it back
    size_t capacity() const { size_t n = 0; for (const Chunk& c : chunks) n += c.size; return n; }
    hipError_t alloc(size_t bytes, void** out)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        for (Chunk& c : chunks)
            if (c.size - c.used >= bytes) { *out = c.p + c.used; c.used += bytes; return hipSuccess; }
        Chunk c{nullptr, std::max(bytes, next_chunk), 0};
        hipError_t e = device_chunk_get(device, c.size, &c.p, &c.size);
        if (e != hipSuccess) return e;
        c.used = bytes;
        *out = c.p;
        chunks.push_back(c);
        return hipSuccess;
    }
};
thread_local Arena* g_arena = nullptr;
struct ArenaScope {
    Arena* prev;
    explicit ArenaScope(Arena* a) : prev(g_arena) { g_arena = a; }
    ~ArenaScope() { g_arena = prev; }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = false;
    ~DevBuf() { if (p && owned) (void)hipFree(p); }
    hipError_t alloc(size_t count)
    {
        n = count;
        if (count == 0) count = 1;
        if (g_arena) { owned = false; return g_arena->alloc(count * sizeof(T), (void**)&p); }
        owned = true;
        return hipMalloc((void**)&p, count * sizeof(T));      // (outside an arena: small, short-lived buffers only)
    }
    hipError_t upload(const std::vector<T>& v)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return copy_h2d(p, v.data(), v.size() * sizeof(T));
    }
};

// Streams and events are expensive to create (~2 ms a stream) and a batch needs a handful: they
// are handed back to a per-device pool when a batch is destroyed and reused by the next one.
struct HandlePool {
    std::mutex mu;
    std::vector<std::vector<hipStream_t>> streams;      // [device]
    std::vector<std::vector<hipEvent_t>> timed, untimed;
    template <class V> static V& at(std::vector<V>& v, int device)
    {
        if ((int)v.size() <= device) v.resize((size_t)device + 1);
        return v[(size_t)device];
    }
    hipError_t stream(int device, hipStream_t* out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = at(streams, device);
            if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        }
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    hipError_t event(int device, bool timing, hipEvent_t* out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = at(timing ? timed : untimed, device);
            if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        }
        return timing ? hipEventCreate(out) : hipEventCreateWithFlags(out, hipEventDisableTiming);
    }
    // device chunks of destroyed arenas (hipMalloc + hipFree cost ~0.5 ms per one-shot call): a few are kept,
    // per device at most kMaxCachedChunks / kMaxCachedBytes; when an allocation fails they are given back first
    std::vector<std::vector<std::pair<char*, size_t>>> chunks;      // [device]
    std::vector<size_t> cached_bytes;                                // [device]
    static constexpr size_t kMaxCachedBytes = (size_t)2 << 30;
    static constexpr size_t kMaxCachedChunks = 6;
    void chunks_trim(int device)        // give every cached chunk of the device back to the runtime
    {
        std::vector<std::pair<char*, size_t>> mine;
        {
            std::lock_guard<std::mutex> lk(mu);
            mine.swap(at(chunks, device));
            at(cached_bytes, device) = 0;
        }
        for (auto& c : mine) (void)hipFree(c.first);
    }
    hipError_t chunk_get(int device, size_t bytes, char** out, size_t* got)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = at(chunks, device);
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].second >= bytes && v[i].second <= 2 * bytes + ((size_t)16 << 20)) {
                    *out = v[i].first; *got = v[i].second;
                    at(cached_bytes, device) -= v[i].second;
                    v.erase(v.begin() + (long)i);
                    return hipSuccess;
                }
        }
        *got = bytes;
        hipError_t e = hipMalloc((void**)out, bytes);
        if (e == hipErrorOutOfMemory) {          // idle cached chunks may be what is missing: release them, try once more
            (void)hipGetLastError();
            chunks_trim(device);
            e = hipMalloc((void**)out, bytes);
        }
        return e;
    }
    void chunk_put(int device, char* p, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = at(chunks, device);
            size_t& held = at(cached_bytes, device);
            if (v.size() < kMaxCachedChunks && held + bytes <= kMaxCachedBytes) {
                v.push_back({p, bytes});
                held += bytes;
                return;
            }
        }
        (void)hipFree(p);
    }
    // pinned staging buffers (hipHostMalloc costs ~0.2 ms): reused like the streams
    std::vector<std::pair<void*, size_t>> pinned;
    hipError_t pinned_get(size_t bytes, void** out, size_t* got)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < pinned.size(); ++i)
                if (pinned[i].second >= bytes && pinned[i].second <= 4 * bytes + (1u << 20)) {
                    *out = pinned[i].first; *got = pinned[i].second;
                    pinned.erase(pinned.begin() + (long)i);
                    return hipSuccess;
                }
        }
        *got = bytes;
        return hipHostMalloc(out, bytes, hipHostMallocDefault);
    }
    void pinned_trim()
    {
        std::vector<std::pair<void*, size_t>> mine;
        { std::lock_guard<std::mutex> lk(mu); mine.swap(pinned); }
        for (auto& c : mine) (void)hipHostFree(c.first);
    }
    void pinned_put(void* p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (pinned.size() < 8) pinned.push_back({p, bytes});
        else (void)hipHostFree(p);
    }
    void put_stream(int device, hipStream_t q)
    {
        std::lock_guard<std::mutex> lk(mu);
        at(streams, device).push_back(q);
    }
    void put_event(int device, bool timing, hipEvent_t e)
    {
        std::lock_guard<std::mutex> lk(mu);
        at(timing ? timed : untimed, device).push_back(e);
    }
};
HandlePool g_handles;

// (A batch runs every rows-per-lane bucket's kernels on streams of its own, the joint mode two batches at a time:
// with HIP's default of 4 hardware queues, copies and short kernels wait behind another stream's long sweeps.
// GPU_MAX_HW_QUEUES=8 helps -- config 3: 25.6 -> 19.2 ms per run -- but the environment belongs to the host
// process: the library does not touch it; see nanorepeat_amd.apply_recommended_env and INTEGRATION.md.)
hipError_t device_chunk_get(int device, size_t bytes, char** out, size_t* got) { return g_handles.chunk_get(device, bytes, out, got); }
void device_chunk_put(int device, char* p, size_t bytes) { g_handles.chunk_put(device, p, bytes); }

// hipMemcpy on pageable memory of 1 MB or more pins the caller's pages for the transfer and unpins them after it
// (the runtime's GPU_PINNED_MIN_XFER_SIZE).  With kernels of the same process running meanwhile that is not harmless on
// this platform: measured on the joint path with >= 29 000 reads (the first task array to reach 1 MB), the device drops
// into a state in which every running kernel takes 2-3 x as long (GPU busy 95 -> 55 %, 32 -> 60 ms per step, for
// seconds; 5 of 9 runs against 1 of 9 with the runtime's threshold lifted -- DESIGN.md 9(2b), profiles/r04_keep_cliff*).
// So the library never hands the runtime a large pageable buffer: it copies through a pinned stage of its own, in
// pieces, synchronously as before.
// (from 896 KB: below the runtime's 1 MB it stages the copy itself, faster than this file can -- config 3 at 5000 reads, whose
// task arrays are 300-400 KB: 6.15 ms per step, 6.55 with everything from 256 KB on going through the stage below)
constexpr size_t kStagedFrom = 896u << 10, kStageBytes = 4u << 20;
struct CopyStage {
    void* p = nullptr;
    size_t cap = 0;
    bool deferred = false;              // upload_batch_begin .. _end: transfers are enqueued, not waited for
    size_t used = 0;                    // ... bytes of the stage they hold
    ~CopyStage() { if (p) g_handles.pinned_put(p, cap); }
    hipError_t ensure() { return p ? hipSuccess : g_handles.pinned_get(kStageBytes, &p, &cap); }
};
thread_local CopyStage g_copy_stage;
// (The transfers go to the null stream: a stream of the stage's own would be one more for the runtime to map onto its
// GPU_MAX_HW_QUEUES hardware queues, and two of a batch's streams that end up sharing a queue run one behind the other --
// measured: config 3 6.15 -> 6.6 ms per step with a stream held by the stage.  The batches' streams are non-blocking, so
// the null stream orders nothing against them.)
void upload_batch_begin() { g_copy_stage.deferred = true; g_copy_stage.used = 0; }
hipError_t upload_batch_end()
{
    CopyStage& st = g_copy_stage;
    if (!st.deferred) return hipSuccess;
    st.deferred = false;
    st.used = 0;
    return st.p ? hipStreamSynchronize(nullptr) : hipSuccess;
}
hipError_t copy_h2d(void* dst, const void* src, size_t bytes)
{
    CopyStage& st = g_copy_stage;
    if (st.deferred) {
        if (bytes == 0) return hipSuccess;
        hipError_t e = st.ensure();
        if (e != hipSuccess) return e;
        for (size_t off = 0; off < bytes;) {
            if (st.cap - st.used < std::min<size_t>(bytes - off, 64u << 10)) {      // the stage is full: wait for what it holds
                if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return e;
                st.used = 0;
            }
            const size_t n = std::min(bytes - off, st.cap - st.used);
            memcpy((char*)st.p + st.used, (const char*)src + off, n);
            if ((e = hipMemcpyAsync((char*)dst + off, (char*)st.p + st.used, n, hipMemcpyHostToDevice, nullptr)) != hipSuccess) return e;
            st.used += (n + 255) & ~(size_t)255;
            off += n;
        }
        return hipSuccess;
    }
    if (bytes < kStagedFrom) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    hipError_t e = st.ensure();
    if (e != hipSuccess) return e;
    for (size_t off = 0; off < bytes; off += st.cap) {
        const size_t n = std::min(st.cap, bytes - off);
        memcpy(st.p, (const char*)src + off, n);
        if ((e = hipMemcpy((char*)dst + off, st.p, n, hipMemcpyHostToDevice)) != hipSuccess) return e;      // (pinned source: no pinning by the runtime)
    }
    return hipSuccess;
}
hipError_t copy_d2h(void* dst, const void* src, size_t bytes)
{
    if (bytes < kStagedFrom) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    CopyStage& st = g_copy_stage;
    hipError_t e = st.ensure();
    if (e != hipSuccess) return e;
    for (size_t off = 0; off < bytes; off += st.cap) {
        const size_t n = std::min(st.cap, bytes - off);
        if ((e = hipMemcpy(st.p, (const char*)src + off, n, hipMemcpyDeviceToHost)) != hipSuccess) return e;
        memcpy((char*)dst + off, st.p, n);
    }
    return hipSuccess;
}

// 2D decomposition: reads of one bucket whose wave states fit the state buffer together; the
// prefix sweeps of a group run, then its tail sweeps, then the next group reuses the buffer.
struct JointGroup {
    int R = 0;
    int bucket = 0;
    int n_pre = 0, n_tail = 0;
    size_t pre_off = 0, tail_off = 0;
};

struct Bucket {
    int R = 0;
    bool chain = false;        // reads longer than one register block: chained row blocks
    bool wide = false;         // chained: int32 cells, one read per wave (else packed int16, two reads per wave)
    bool half = false;         // k_sweep_ring32: two read pairs per wave, 32 lanes each (R = rows per lane of a half)
    int payload_R = 0;         // chained: row block of the extents kernel (NRA_CHAIN_R / NRA_CHAIN_R_TEST)
    size_t strip_off = 0;      // chained: this bucket's scratch strips in chain_sweep (int32 index)
    int n_strips = 0;
    bool mt = false;           // chained LDS-ring sweeps with the row blocks of a read as concurrent waves (k_sweep_ringmt)
    std::vector<std::pair<size_t, int>> mt_groups;   // launches: (first entry of chain_blocks, number of blocks)
    int n_pair = 0;            // pk16 tasks (1D score / 2D strand probe)
    size_t pair_off = 0;       // offset into pair_tasks
    int n_queue = 0;           // payload tasks prebuilt on the host (ALL_EXTENTS / 2D cells)
    size_t queue_off = 0;      // offset into queue_tasks (also the base of the tie queue)
    size_t queue_cap = 0;      // capacity of this bucket's queue region
    int n_sweep = 0;           // junction-decomposition tasks (pairs of reads)
    size_t sweep_off = 0;
    bool ring = false;         // k_sweep_ring (LDS hand-off) instead of k_sweep_pk16 (DPP hand-off)
    bool quanta = false;       // ... and its reverse and forward sweeps as ONE launch of quanta taken by ticket (k_sweep_ringq)
    size_t q_off = 0, q_task_off = 0, q_state_off = 0;      // into q_list, q_words' arrivals (2 a task), q_state (int32, 2 slots a task)
    int n_quanta = 0, q_steps = 0;                            // entries of q_list; steps of a part
    int n_jbwd = 0;            // 2D decomposition: reverse sweeps (one per read)
    size_t jbwd_off = 0;
    int n_jlpk = 0, n_jrpk = 0;    // ... packed sweeps of the payload-free columns of L / rev(R) (one per pair of reads)
    size_t jlpk_off = 0, jrpk_off = 0;
    int n_comb = 0;            // 2D, junction at the end of mid: combine tasks (one per read)
    size_t comb_off = 0;
    int n_probe = 0;           // 2D, chained reads: strand-probe payload tasks (two per read)
    size_t probe_off = 0;
    int64_t cells_pair = 0;    // executed cells per run, pk16
    int64_t cells_queue = 0;   // executed cells per run, prebuilt payload queue
    int64_t cells_sweep = 0;   // executed cells per run, both sweeps
};

NraScoreParams to_params(const nra_scoring_t& sc)
{
    NraScoreParams p;
    p.match = sc.match; p.mismatch = sc.mismatch;
    p.open1 = sc.gap_open1 + sc.gap_ext1; p.ext1 = sc.gap_ext1;
    p.open2 = sc.gap_open2 + sc.gap_ext2; p.ext2 = sc.gap_ext2;
    p.ambi = sc.sc_ambi; p.min_score = sc.min_dp_score;
    return p;
}

bool scoring_ok(const nra_scoring_t* sc)
{
    if (!sc) return false;
    if (sc->match <= 0 || sc->match > 64 || sc->mismatch < 0 || sc->mismatch > 64) return false;
    if (sc->gap_ext1 <= 0 || sc->gap_ext2 <= 0 || sc->gap_open1 < 0 || sc->gap_open2 < 0) return false;
    if (sc->gap_open1 + sc->gap_ext1 > 1024 || sc->gap_open2 + sc->gap_ext2 > 1024) return false;
    if (sc->sc_ambi < 0 || sc->sc_ambi > 64) return false;
    return true;
}

// Largest alignment score a read of qlen bases can reach, against what a cell format holds:
// int32 cells keep the score in their upper 16 bits; the packed int16 kernels keep value + 8192
// (brute force), two biased values added (chained sweep: 2 x 8192 + score), or the same with
// doubled scores (origin-bit sweep).
const int64_t kScoreCapI32 = 32000, kScoreCapPk16 = 24000, kScoreCapBit = 13500;
inline int64_t max_score(const nra_scoring_t* sc, int64_t qlen) { return (int64_t)sc->match * qlen; }

// A rows-per-lane bucket with few reads would run as its own under-filled launches: fold it into the
// next non-empty instantiation within `span` more rows per lane (the extra rows are padding).
void fold_small_buckets(std::vector<std::vector<int32_t>>& by_bucket, size_t min_reads, int span)
{
    for (int bi = 0; bi + 1 < kNumR; ++bi) {
        if (by_bucket[bi].empty() || by_bucket[bi].size() >= min_reads) continue;
        int up = -1;
        for (int bj = bi + 1; bj < kNumR && kRList[bj] <= kRList[bi] + span; ++bj)
            if (!by_bucket[bj].empty()) { up = bj; break; }
        if (up < 0) continue;
        by_bucket[up].insert(by_bucket[up].end(), by_bucket[bi].begin(), by_bucket[bi].end());
        by_bucket[bi].clear();
    }
}

// SIMDs of a device (4 per compute unit; 1024 on MI355X)
int device_simds(int device)
{
    static std::mutex mu;
    static std::vector<int> cache;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)cache.size() <= device) cache.resize((size_t)device + 1, 0);
    if (cache[(size_t)device] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        cache[(size_t)device] = 4 * cus;
    }
    return cache[(size_t)device];
}

// Every launch of the concurrent-block sweeps (k_sweep_ringmt) tags the granules it publishes with an epoch of its own:
// process-wide, never 0 (a zeroed strip carries none).
uint32_t next_epoch()
{
    static std::atomic<uint32_t> counter{0};
    uint32_t e = ++counter;
    if (e == 0) e = ++counter;
    return e;
}

// Scratch strips of a chained launch: one per wave, `bytes_per_strip` each (a few values per template column).  As
// many as there are tasks, at most `most`, and no more than fit the budget: a very wide template (up to
// NRA_MAX_TLEN_WIDE columns: 160 - 190 MB per strip) gets fewer waves, never none.
int chain_strips(size_t n_tasks, int most, size_t bytes_per_strip, size_t budget = NRA_CHAIN_SCRATCH_BUDGET)
{
    const size_t fit = std::max<size_t>(1, budget / std::max<size_t>(bytes_per_strip, 1));
    return (int)std::max<size_t>(1, std::min(std::min<size_t>(n_tasks, (size_t)most), fit));
}

// executed cells of one wave sweep with 64 cells in flight (k_score_pk16, payload, joint kernels):
// 64*R rows x (ceil((tlen+63)/64)*64) columns
int64_t sweep_cells(int R, int tlen) { return (int64_t)64 * R * (((int64_t)tlen + 126) / 64 * 64); }
// k_joint_sweep (reverse and tail sweeps): the step loop stops when the lane of the read's last row has taken the last column
int64_t joint_cells(int R, int ncols, int qlen) { return (int64_t)64 * R * ((int64_t)ncols + std::min(63, std::max(qlen - 1, 0) / R)); }
// k_sweep_pk16: two virtual cells per lane, 128 in flight, the step loop stops with the last column
int64_t sweep128_cells(int R, int ncols) { return (int64_t)64 * R * ((int64_t)ncols + 127); }

}  // namespace

struct nra_batch {
    Arena arena;               // declared first: released after every DevBuf below
    Arena cell_arena;          // 2D: buffers that depend on the cell list; reset by nra_batch2d_set_cells
    Arena keep_arena;          // 2D: column states kept from one routed grid for the next
    // 2D: what nra_batch2d_set_cells needs from the reads part
    std::vector<NraDevRead> host_reads;
    std::vector<uint8_t> chained_reads;
    std::string jr_left, jr_unit1, jr_mid, jr_unit2, jr_right;
    nra_scoring_t scoring{};
    size_t n_q2bit_words = 0;
    bool reads_have_n = false;
    // 2D: the reverse sweep of a read (the R side of the junction, A) depends on the read, R and the strand
    // only: a later cell list of the same batch reuses it (rev_strand: strand it was made for, 0 = not made)
    std::vector<int8_t> rev_strand;
    std::vector<std::pair<int32_t, int8_t>> rev_pending;    // made by the next run
    // 2D: rows-per-lane bucket and partner of every read, fixed when the reads are packed: the packed sweeps
    // (k_joint_pk16) take two reads per wave, and the state they leave at the end of L depends on the read and
    // its strand only, so it is kept like the reverse sweeps (lst_strand / lst_pending)
    std::vector<int32_t> jbucket, jpair_of;
    std::vector<NraJointPairTask> jpairs;
    bool jpack_l = false, jpack_r = false;
    std::vector<int8_t> lst_strand;
    std::vector<std::pair<int32_t, int8_t>> lst_pending;
    DevBuf<int32_t> jlstate, jrstate;            // packed wave states at the end of L / of rev(R) outside the window
    DevBuf<NraJointPairTask> jlpk_tasks, jrpk_tasks;
    int kind = 0;              // 1 = 1D, 2 = 2D
    int device = 0;
    int flags = 0;
    int has_n = 0;
    hipStream_t stream = nullptr;
    NraScoreParams sp{};
    int n_reads = 0, n_regions = 0;
    int64_t n_cands = 0;       // 1D candidates or 2D cells
    std::vector<Bucket> buckets;

    DevBuf<uint8_t> pool;
    DevBuf<uint32_t> q2bit, qnmask;
    DevBuf<NraDevRegion> regions;
    DevBuf<NraDevRead> reads, reads_init;
    DevBuf<NraPairTask> pair_tasks;
    DevBuf<NraSweepTask> sweep_tasks;
    DevBuf<int32_t> snap;                      // R side of the junction: per sweep task 3 planes of R x 64 (both reads packed)
    DevBuf<int32_t> read_a1d;                  // A per read: best alignment inside R (origin-bit scheme)
    DevBuf<uint8_t> cand_flag;                 // flank verdict per candidate
    bool brute = false;                        // K independent alignments instead of the sweeps
    DevBuf<int32_t> chain_sweep;                // scratch strips of the chained row blocks (int32 sweep cells)
    DevBuf<int64_t> chain_payload;              // ... and of the chained payload kernels (int32 or int64 cells)
    DevBuf<NraJointTask> jbwd_tasks, jpre_tasks, jtail_tasks; // 2D decomposition: per read / per read / per (read, k1) run
    DevBuf<int32_t> jsnap, jread_a;             // R side of the junction (3 x int32 per base), A per read
    DevBuf<int32_t> jk1list, jstate;            // k1 values per read; wave states of the prefix sweeps
    // routed grids, junction at the end of mid: column states of the MID sweeps / the extended reverse sweeps,
    // B(k1) / A(k2), one combine task per read
    DevBuf<int32_t> jfs, jrs, jfb, jra;
    DevBuf<NraJointCombineTask> jcomb_tasks;
    bool joint_v2 = false, joint_v2_prev = false;
    bool joint_chain = false;                   // ... with the MID part as column-parallel scans on column states (k_joint_midscan)
    std::vector<JointGroup> jgroups;
    bool all_strands_given = false;             // 2D: every read came with its strand, no probe needed
    int chain_cap = 0;
    int payload_strips = 0;                     // waves of a chained payload launch (strips in chain_payload)
    DevBuf<NraChainBlock> chain_blocks;         // k_sweep_ringmt: (task, row block) lists, launch group after launch group
    DevBuf<uint64_t> mt_strips;                 // ... the granule strips between consecutive blocks (zeroed at create)
    DevBuf<int32_t> mt_words;                   // ... [0] launch-wide give-up word (kMtGiveUp), [1] unused, [2 + bucket] that bucket's ticket
    // the sweeps in quanta (k_sweep_ringq): ticket order, the words of a run ([0] give-up word, [1 + bucket] that bucket's
    // ticket, then one arrival counter per task -- zeroed by one memset per run), the wave states at the cut
    DevBuf<uint32_t> q_list;
    DevBuf<int32_t> q_words, q_state;
    size_t q_words_n = 0, q_arrivals_off = 0;
    bool q_checked = true;
    DevBuf<NraTask> queue_tasks;
    DevBuf<int32_t> queue_count;   // per bucket: prebuilt queue length (constant)
    DevBuf<int32_t> tie_count;     // per bucket: tie queue length (device-written)
    DevBuf<uint32_t> bucket_task_base;
    // 1D
    DevBuf<int32_t> kmin, kmax, read_bucket;
    DevBuf<uint32_t> coff;
    DevBuf<int32_t> cand_score, cand_tstart, cand_tend, best_score, n_ties;
    DevBuf<int64_t> sum_k, sum_k2;
    DevBuf<uint8_t> status;
    // the per-read results above are views into ONE block [sum_k | sum_k2 | best_score | n_ties | status(+strand)],
    // so that a fetch is one D2H copy into a pinned staging buffer instead of four or six pageable ones
    DevBuf<uint8_t> result_block;
    void* result_stage = nullptr;      // pinned host mirror (from the handle pool)
    size_t result_bytes = 0, result_stage_bytes = 0;
    // 2D
    DevBuf<int32_t> probe_score, probe_dummy, cell_k1, cell_k2;
    DevBuf<NraGridRow> grid_rows;               // a routed grid instead of cell_k1 / cell_k2 (grid_step1 > 0)
    int32_t grid_step1 = 0, grid_step2 = 0;
    DevBuf<NraTask> probe_tasks;                // chained 2D reads: (read, first cell) in both orientations
    DevBuf<int32_t> probe_count;
    DevBuf<uint32_t> cell_first, cell_cnt;
    DevBuf<int8_t> strand_in, strand_out;
    bool have_strand_in = false;
    // 2D routed grids: the column states the prefix sweep (at every k1 of keep_rows) and the extended reverse sweep (at
    // every k2) left, kept for a later grid whose cells lie inside (nra_batch2d_invalidate / other strands drop them)
    bool keep_valid = false, keep_pending = false;
    std::vector<NraGridRow> keep_rows;
    std::vector<NraGridRow> cur_rows;           // the current routed grid's rows (nra_batch2d_refine checks what a refinement can ask for)
    int32_t cur_step1 = 0, cur_step2 = 0;
    std::vector<int8_t> keep_strand;
    std::vector<uint64_t> keep_state_off, keep_rs_off;
    std::vector<int32_t> keep_ra_off;
    bool keep_off = false;                      // set for one retry: this cell list keeps nothing (the kept states did not fit)
    int joint_keep = 0;                         // this cell list: 0 = nothing kept, 1 = sweeps that keep, 2 = no sweeps, kept states
    // 2D: the packed sweep of rev(R) up to the window is valid for this strand of the read (like lst_strand for L)
    std::vector<int8_t> rpk_strand;
    std::vector<std::pair<int32_t, int8_t>> rpk_pending;
    // 2D: flank sweeps enqueued ahead of the cell list (nra_batch2d_sweep_flanks): buffers of their own, alive as long as
    // the batch (the cell list's pool / reads / task arrays are rebuilt in place while these kernels run)
    DevBuf<uint8_t> warm_pool;
    DevBuf<NraDevRegion> warm_region;
    DevBuf<NraDevRead> warm_reads;
    DevBuf<NraJointPairTask> warm_tasks;
    std::vector<hipEvent_t> warm_ev;            // end of the ahead-of-time sweeps: [2 i] rev(R) side, [2 i + 1] L side of part i
    std::vector<int> warm_R;                    // rows per lane of part i
    int warm_has_n = 0;
    int64_t warm_cells = 0;                     // cells those sweeps execute (added to the statistics of the run they belong to)
    bool warm_built = false, warm_pending = false;   // pending: enqueued, the next run's streams have not waited for them yet
    // 2D: a refinement enqueued behind the run of a routed grid (nra_batch2d_refine): its routing happens on the device
    DevBuf<double> rf_bounds;                   // lo1 | hi1 | lo2 | hi2, n_reads each
    DevBuf<NraGridRow> rf_rows, rf_keep;        // the refinement's rows (device-written); the kept ranges
    DevBuf<uint32_t> rf_first, rf_cnt;          // cells of a read: at read * cap; how many (device-written)
    DevBuf<int64_t> rf_words;                   // [0] cells, [1] rows outside the kept ranges, [2] executed, [3] algorithmic cells
    DevBuf<int32_t> rf_rowspad;                 // 64 * rows per lane of a read's bucket (0: not swept)
    DevBuf<NraJointTask> rf_mid_tasks;
    DevBuf<NraJointCombineTask> rf_comb_tasks;
    DevBuf<int32_t> rf_fs, rf_fb;
    DevBuf<int32_t> rf_cell_score, rf_cell_wscore;   // the refinement's cells (the grid's own stay where they are: the grid may run again)
    int64_t rf_n_cands = 0;
    bool refined = false, refine_read = false;  // this run carries a refinement; its counters have been read back
    int64_t refine_bad_rows = 0;
    bool cells_need_clear = true;               // 2D: some cells are written by no kernel unless found

    std::vector<hipStream_t> bstreams;   // one per bucket: the sweep chains of different buckets overlap
    hipEvent_t mt_rev_ev = nullptr;      // end of the row blocks' reverse launches: buckets of quanta may wait for it (run_1d)
    std::vector<hipEvent_t> bdone;       // bucket chain finished
    hipEvent_t fork_ev = nullptr, fork2_ev = nullptr;
    hipEvent_t phase_ev[2] = {nullptr, nullptr};   // scoring phase start / end (timing)
    std::vector<hipEvent_t> ev;    // [0]=run start, [1]=run end, then pairs per dominant launch
    int n_score_ev = 0, n_ext_ev = 0;
    bool ran = false, accounted = false;
    bool have_cells = false;   // 2D: nra_batch2d_set_cells has run
    bool mt_checked = true;         // 1D: the give-up word of the concurrent-block sweeps has been read after this run
    bool flanks_enqueued = false;   // 2D: the strand-only kernels of this cell list are on their streams already
    int ev_next = 2;                // 2D: next free timing-event pair (the run is enqueued in two parts)
    nra_stats_t stats{};

    ~nra_batch()
    {
        // all work of the batch has to be over before its handles serve another batch
        if (stream) (void)hipStreamSynchronize(stream);
        for (hipStream_t q : bstreams) if (q) (void)hipStreamSynchronize(q);
        for (hipEvent_t e : ev) if (e) g_handles.put_event(device, true, e);
        for (hipEvent_t e : phase_ev) if (e) g_handles.put_event(device, true, e);
        for (hipEvent_t e : bdone) if (e) g_handles.put_event(device, false, e);
        for (hipEvent_t e : warm_ev) if (e) g_handles.put_event(device, false, e);
        if (fork_ev) g_handles.put_event(device, false, fork_ev);
        if (mt_rev_ev) g_handles.put_event(device, false, mt_rev_ev);
        if (fork2_ev) g_handles.put_event(device, false, fork2_ev);
        for (hipStream_t q : bstreams) if (q) g_handles.put_stream(device, q);
        if (stream) g_handles.put_stream(device, stream);
        if (result_stage) g_handles.pinned_put(result_stage, result_stage_bytes);
    }
};

namespace {

// ---- host worker threads --------------------------------------------------------------
// Starting a thread costs ~50 us and a one-shot call has ~1 ms of packing to spread: the workers are
// started once and woken per job.  run(n, fn) runs fn(0..n-1) on the caller and n-1 workers and returns
// when all are done.  One job at a time (callers from several threads take turns).
class HostWorkers {
    std::mutex job_mu, mu;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    const std::function<void(int)>* fn = nullptr;
    int n_jobs = 0, next = 0, pending = 0;
    uint64_t generation = 0;

    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            wake.wait(lk, [&] { return generation != seen; });
            seen = generation;
            while (next < n_jobs) {
                const int i = next++;
                lk.unlock();
                (*fn)(i);
                lk.lock();
                if (--pending == 0) done.notify_all();
            }
        }
    }

public:
    void run(int n, const std::function<void(int)>& f)
    {
        if (n <= 1) { if (n == 1) f(0); return; }
        std::lock_guard<std::mutex> one_job(job_mu);
        std::unique_lock<std::mutex> lk(mu);
        while ((int)threads.size() < n - 1) { threads.emplace_back([this] { loop(); }); threads.back().detach(); }
        fn = &f; n_jobs = n; next = 0; pending = n;
        ++generation;
        wake.notify_all();
        while (next < n_jobs) {            // the caller works too
            const int i = next++;
            lk.unlock();
            f(i);
            lk.lock();
            --pending;
        }
        done.wait(lk, [&] { return pending == 0; });
        fn = nullptr;
    }

    // The same job in two halves: start() hands the n pieces to the workers and returns, finish() -- same thread --
    // takes what is left of them and waits for the rest.  `f` has to stay alive in between.
    void start(int n, const std::function<void(int)>& f)
    {
        job_mu.lock();
        std::unique_lock<std::mutex> lk(mu);
        while ((int)threads.size() < n) { threads.emplace_back([this] { loop(); }); threads.back().detach(); }
        fn = &f; n_jobs = n; next = 0; pending = n;
        ++generation;
        wake.notify_all();
    }
    void finish()
    {
        std::unique_lock<std::mutex> lk(mu);
        while (next < n_jobs) {
            const int i = next++;
            lk.unlock();
            (*fn)(i);
            lk.lock();
            --pending;
        }
        done.wait(lk, [&] { return pending == 0; });
        fn = nullptr;
        lk.unlock();
        job_mu.unlock();
    }
};
// never destroyed: the workers wait on its condition variable for the life of the process, and destroying
// a condition variable that has waiters blocks (glibc) -- at exit that is a hang
HostWorkers& g_workers = *new HostWorkers;

// ---- sequence packing ---------------------------------------------------------------
struct PackedReads {
    std::vector<uint32_t> q2bit, nmask;
    std::vector<NraDevRead> reads;
    bool has_n = false;
};

// Layout of the packed reads (offsets, lengths, regions) and the zeroed word arrays; the words themselves are
// written by PackJob below.
int pack_layout(int32_t n_reads, const int64_t* seq_off, const int32_t* read_region,
                int32_t n_regions, PackedReads& out, int64_t max_len = NRA_MAX_QLEN_1BLOCK)
{
    out.reads.resize((size_t)n_reads);
    uint64_t base = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        int64_t len = seq_off[r + 1] - seq_off[r];
        if (len < 0) return fail(NRA_E_ARG, "seq_off must be non-decreasing");
        if (len > max_len)
            return fail(NRA_E_RANGE, "read " + std::to_string(r) + " has " + std::to_string(len) +
                                         " bases; this entry point holds at most " + std::to_string(max_len));
        int32_t g = read_region ? read_region[r] : 0;
        if (g < 0 || g >= n_regions) return fail(NRA_E_ARG, "read_region out of range");
        out.reads[r].qoff = (uint32_t)base;
        out.reads[r].qlen = (int32_t)len;
        out.reads[r].region = g;
        out.reads[r].rc = 0;
        base += ((uint64_t)len + 31) / 32 * 32;
        if (base > 0xfff00000ull) return fail(NRA_E_RANGE, "read pool exceeds 4G bases");
    }
    out.q2bit.assign((size_t)(base / 16) + 1, 0);
    out.nmask.assign((size_t)(base / 32) + 1, 0);
    return NRA_OK;
}

// The 2-bit packing itself, on the worker pool.  Every read starts on a 32-base boundary, so reads write disjoint
// words: one piece per ~256 k bases, each a contiguous run of reads.  The constructor starts the workers and
// returns -- the caller goes on with whatever needs the layout only (buckets, tasks) -- and join() (or the
// destructor, on an early return) takes the pieces that are left and waits; `has_n` is known after that.
class PackJob {
    PackedReads& out;
    const char* seqs;
    const int64_t* seq_off;
    int32_t n_reads;
    int nthreads = 1;
    std::vector<uint8_t> saw_n;
    std::function<void(int)> work;
    bool running = false;

public:
    PackJob(int32_t n_reads_, const char* seqs_, const int64_t* seq_off_, PackedReads& out_)
        : out(out_), seqs(seqs_), seq_off(seq_off_), n_reads(n_reads_)
    {
        const uint64_t base = (uint64_t)(out.q2bit.size() - 1) * 16;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        nthreads = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw, 16), base >> 18));
        saw_n.assign((size_t)nthreads, 0);
        work = [this](int t) {
            uint32_t any_n = 0;
            const int32_t r0 = (int32_t)((int64_t)n_reads * t / nthreads), r1 = (int32_t)((int64_t)n_reads * (t + 1) / nthreads);
            for (int32_t r = r0; r < r1; ++r) {
                const uint8_t* s = reinterpret_cast<const uint8_t*>(seqs + seq_off[r]);
                const int32_t len = out.reads[r].qlen;
                uint32_t* q2 = out.q2bit.data() + (out.reads[r].qoff >> 4);
                uint32_t* nm = out.nmask.data() + (out.reads[r].qoff >> 5);
                for (int32_t i = 0; i < len; i += 32) {
                    const int32_t n = std::min(32, len - i);
                    uint64_t w = 0;                    // 32 bases x 2 bits; an N keeps code 0 and sets its mask bit
                    uint32_t mask = 0;
                    for (int32_t j = 0; j < n; ++j) {  // branch-free: code 4 (N) = 0b100 -> bits 0..1 = 0, bit 2 = the mask
                        const uint32_t c = kBaseCode[s[i + j]];
                        w |= (uint64_t)(c & 3u) << (2 * j);
                        mask |= (c >> 2) << j;
                    }
                    q2[(i >> 4)] = (uint32_t)w;
                    if (n > 16) q2[(i >> 4) + 1] = (uint32_t)(w >> 32);
                    nm[i >> 5] = mask;
                    any_n |= mask;
                }
            }
            saw_n[(size_t)t] = any_n ? 1 : 0;
        };
        if (nthreads > 1) { g_workers.start(nthreads, work); running = true; }
        else work(0);
    }
    void join()
    {
        if (running) { g_workers.finish(); running = false; }
        for (uint8_t v : saw_n) out.has_n |= v != 0;
    }
    ~PackJob() { if (running) g_workers.finish(); }
    PackJob(const PackJob&) = delete;
    PackJob& operator=(const PackJob&) = delete;
};

int pack_reads(int32_t n_reads, const char* seqs, const int64_t* seq_off, const int32_t* read_region,
               int32_t n_regions, PackedReads& out, int64_t max_len = NRA_MAX_QLEN_1BLOCK)
{
    const int rc = pack_layout(n_reads, seq_off, read_region, n_regions, out, max_len);
    if (rc) return rc;
    PackJob job(n_reads, seqs, seq_off, out);
    job.join();
    return NRA_OK;
}

uint32_t pool_append(std::vector<uint8_t>& pool, const char* s, int32_t len, const char* unit,
                     int32_t ulen, int32_t reps, bool& has_n)
{
    uint32_t off = (uint32_t)pool.size();
    for (int32_t i = 0; i < len; ++i) { uint8_t c = encode_base(s[i]); has_n |= c >= 4; pool.push_back(c); }
    for (int32_t k = 0; k < reps; ++k)
        for (int32_t i = 0; i < ulen; ++i) { uint8_t c = encode_base(unit[i]); has_n |= c >= 4; pool.push_back(c); }
    return off;
}

int common_init(nra_batch* b, int device, const nra_scoring_t* sc, int flags)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NRA_E_DEVICE, "no HIP device: nanorepeat_amd has no CPU path");
    if (device < 0 || device >= ndev) return fail(NRA_E_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    b->device = device;
    b->arena.device = device;
    b->cell_arena.device = device;
    b->keep_arena.device = device;
    b->flags = flags;
    b->sp = to_params(*sc);
    HIP_TRY(g_handles.stream(device, &b->stream));
    return NRA_OK;
}

// one device block (+ pinned host mirror) for the per-read results; n8 = int64 arrays, strand: 2D only
int alloc_results(nra_batch* b, size_t n, bool two_d)
{
    const size_t n8 = two_d ? 2 : 1;
    const size_t np = (n + 7) & ~(size_t)7;                 // keeps every sub-array 8-byte aligned
    b->result_bytes = np * (8 * n8 + 4 + 4 + 1 + (two_d ? 1 : 0));
    HIP_TRY(b->result_block.alloc(b->result_bytes));
    HIP_TRY(g_handles.pinned_get(std::max<size_t>(b->result_bytes, 8), &b->result_stage, &b->result_stage_bytes));
    uint8_t* p = b->result_block.p;
    auto view = [&](auto& buf, size_t bytes) {
        buf.p = reinterpret_cast<decltype(buf.p)>(p); buf.n = n; buf.owned = false; p += bytes;
    };
    view(b->sum_k, np * 8);
    if (two_d) view(b->sum_k2, np * 8);
    view(b->best_score, np * 4);
    view(b->n_ties, np * 4);
    view(b->status, np);
    if (two_d) view(b->strand_out, np);
    return NRA_OK;
}

// the results block -> the pinned mirror (one copy); returns host pointers through `at`
int fetch_results(nra_batch* b)
{
    if (b->result_bytes)
        HIP_TRY(hipMemcpy(b->result_stage, b->result_block.p, b->result_bytes, hipMemcpyDeviceToHost));
    return NRA_OK;
}
template <class T> const T* staged(const nra_batch* b, const DevBuf<T>& buf)
{
    return reinterpret_cast<const T*>(static_cast<const uint8_t*>(b->result_stage) +
                                      (reinterpret_cast<const uint8_t*>(buf.p) - b->result_block.p));
}

int make_events(nra_batch* b, int n)
{
    b->ev.assign((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) HIP_TRY(g_handles.event(b->device, true, &b->ev[i]));
    for (int i = 0; i < 2; ++i) HIP_TRY(g_handles.event(b->device, true, &b->phase_ev[i]));
    return NRA_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

int nra_abi_version(void) { return NRA_ABI_VERSION; }
const char* nra_version(void) { return NRA_VERSION_STR; }
const char* nra_last_error(void) { return g_err.c_str(); }

int nra_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_err = hipGetErrorString(e); return NRA_E_DEVICE; }
    return n;
}

int nra_release_cached_memory(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NRA_E_DEVICE, "no HIP device");
    if (device >= ndev) return fail(NRA_E_ARG, "device index out of range");
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int d = device < 0 ? 0 : device; d < (device < 0 ? ndev : device + 1); ++d) {
        (void)hipSetDevice(d);
        g_handles.chunks_trim(d);
    }
    g_handles.pinned_trim();
    (void)hipSetDevice(prev);
    return NRA_OK;
}

void nra_default_scoring(nra_scoring_t* sc)
{
    if (!sc) return;
    sc->match = 2; sc->mismatch = 4;
    sc->gap_open1 = 4; sc->gap_ext1 = 2;
    sc->gap_open2 = 24; sc->gap_ext2 = 1;
    sc->sc_ambi = 1; sc->min_dp_score = 80;
}

// ---- 1D -----------------------------------------------------------------------------
int nra_batch1d_create(int device, const nra_region_t* regions, int32_t n_regions,
                       int32_t n_reads, const char* seqs, const int64_t* seq_off,
                       const int32_t* read_region, const int32_t* kmin, const int32_t* kmax,
                       const nra_scoring_t* sc, int32_t flags, nra_batch_t** out)
{
    if (!out) return fail(NRA_E_ARG, "out is NULL");
    *out = nullptr;
    if (!regions || n_regions <= 0 || n_reads < 0) return fail(NRA_E_ARG, "bad region / read count");
    if (n_reads > 0 && (!seqs || !seq_off || !kmin || !kmax)) return fail(NRA_E_ARG, "NULL input array");
    if (n_regions > 1 && n_reads > 0 && !read_region) return fail(NRA_E_ARG, "read_region required when n_regions > 1");
    if (!scoring_ok(sc)) return fail(NRA_E_ARG, "scoring parameters out of range");
    for (int32_t g = 0; g < n_regions; ++g) {
        const nra_region_t& rg = regions[g];
        if (rg.left_len < 0 || rg.right_len < 0 || rg.unit_len <= 0 || !rg.unit ||
            (rg.left_len > 0 && !rg.left) || (rg.right_len > 0 && !rg.right))
            return fail(NRA_E_ARG, "bad region " + std::to_string(g));
    }

    nra_batch* b = new nra_batch();
    std::unique_ptr<nra_batch> guard(b);
    ArenaScope arena_scope(&b->arena);
    b->kind = 1; b->n_reads = n_reads; b->n_regions = n_regions;
    int rc = common_init(b, device, sc, flags);
    if (rc) return rc;

    PhaseClock clk;
    clk.mark("device, stream");
    // the workers pack the reads while this thread builds templates, buckets and tasks from the layout
    PackedReads pr;
    rc = pack_layout(n_reads, seq_off, read_region, n_regions, pr, NRA_MAX_QLEN);
    if (rc) return rc;
    PackJob packing(n_reads, seqs, seq_off, pr);
    clk.mark("read layout, packing started");

    // per-region largest k, candidate offsets
    std::vector<int32_t> region_kmax((size_t)n_regions, 0);
    std::vector<uint32_t> coff((size_t)n_reads + 1, 0);
    std::vector<int32_t> read_bucket((size_t)n_reads, -1);
    int64_t total = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        coff[r] = (uint32_t)total;
        if (kmin[r] > kmax[r]) continue;
        if (kmin[r] < 0) return fail(NRA_E_ARG, "kmin < 0");
        const int g = pr.reads[r].region;
        const int64_t tl = (int64_t)regions[g].left_len + (int64_t)regions[g].unit_len * kmax[r] + regions[g].right_len;
        if (tl > NRA_MAX_TLEN_WIDE) return fail(NRA_E_RANGE, "template longer than " + std::to_string(NRA_MAX_TLEN_WIDE));
        region_kmax[g] = std::max(region_kmax[g], kmax[r]);
        total += (int64_t)kmax[r] - kmin[r] + 1;
        if (total > 0x7ff00000ll) return fail(NRA_E_RANGE, "more than 2^31 candidates in one batch");
    }
    coff[n_reads] = (uint32_t)total;
    b->n_cands = total;

    // code pool + region table
    std::vector<uint8_t> pool;
    std::vector<NraDevRegion> dregs((size_t)n_regions);
    bool has_n = false;                          // templates; the reads' share is known when the packing is over
    for (int32_t g = 0; g < n_regions; ++g) {
        const nra_region_t& rg = regions[g];
        NraDevRegion d{};
        d.p1_off = pool_append(pool, rg.left, rg.left_len, rg.unit, rg.unit_len, region_kmax[g], has_n);
        d.p2_off = (uint32_t)pool.size();
        d.p3_off = pool_append(pool, rg.right, rg.right_len, nullptr, 0, 0, has_n);
        {   // rev(R) + rev(unit)^kmax for the reverse sweep
            std::string rr(rg.right, rg.right + rg.right_len), ru(rg.unit, rg.unit + rg.unit_len);
            std::reverse(rr.begin(), rr.end());
            std::reverse(ru.begin(), ru.end());
            d.pr_off = pool_append(pool, rr.data(), rg.right_len, ru.data(), rg.unit_len, region_kmax[g], has_n);
        }
        d.l1 = rg.left_len; d.m1 = rg.unit_len; d.l2 = 0; d.m2 = 0; d.l3 = rg.right_len;
        dregs[g] = d;
        if (pool.size() > 0xfff00000ull) return fail(NRA_E_RANGE, "template pool exceeds 4 GB");
    }
    pool.push_back(0);

    // buckets by rows-per-lane; tasks
    const bool all_ext = (flags & NRA_F_ALL_EXTENTS) != 0;
    bool brute = all_ext || (flags & NRA_F_BRUTE_FORCE) != 0;
    for (int32_t g = 0; g < n_regions; ++g)       // the junction needs a base on either side
        if (regions[g].left_len < 1 || regions[g].right_len < 1) brute = true;
    {   // the sweep kernels keep (substitution score + gap-open cost) in unsigned table bytes
        // (doubled: the low bit of every state is the origin bit)
        const int o1 = sc->gap_open1 + sc->gap_ext1;
        if (o1 < sc->mismatch || o1 < sc->sc_ambi || 2 * (sc->match + o1) > 127) brute = true;
        // and every state in [0x0400, 0x7bff] (nra_sweep.hip: bias 2048, "minus infinity" 1280)
        const int o2 = sc->gap_open2 + sc->gap_ext2;
        if (2 * sc->gap_ext1 > 256 || 2 * sc->gap_ext2 > 256 || 2 * (o2 + sc->mismatch + sc->sc_ambi) > 700) brute = true;
    }
    // Which reads leave the packed int16 sweeps for the chained int32 ones (one read per wave, any length):
    // more rows than one register block, scores beyond the doubled int16 range, or a template whose
    // extents do not fit the 16-bit payload of the int32 extents kernel.
    const bool test_chain = (flags & NRA_F_TEST_CHAIN) != 0;
    // reads of more rows than this leave the one-block kernels for the chained ones (NRA_CHAIN_FROM: experiments)
    int chain_from = NRA_RING_MT_FROM;
    if (const char* e = getenv("NRA_CHAIN_FROM")) chain_from = std::max(64, std::min(atoi(e), NRA_MAX_QLEN_1BLOCK));
    const bool ring_units = [&] { for (int32_t g = 0; g < n_regions; ++g) if (regions[g].unit_len > NRA_SWEEP_RING_MAX_M) return false; return true; }();
    if (!ring_units || brute || (flags & (NRA_F_DPP_SWEEP | NRA_F_SERIAL_CHAIN))) chain_from = NRA_MAX_QLEN_1BLOCK;
    if (chain_from == NRA_RING_MT_FROM && !getenv("NRA_CHAIN_FROM")) {
        // The reads of (NRA_RING_MT_FROM, 3072] rows can go either way: one register block of 28 - 48 rows per lane -- one
        // wave per SIMD, padding in steps of 256 / 512 rows, 4.3 T cells/s x the fill of the launch's last round of the
        // SIMDs -- or row blocks of 12 .. 15 rows per lane, three waves per SIMD: 3.7 - 4.0 T cells/s (4.1 - 4.4 with two
        // blocks per read) once there are two rounds of them, 1.9 + 1.2 x rounds - 0.25 x (blocks per read - 2) below that
        // (the blocks of a read lag one another).  The model is fitted to tools/gpu_block_rows.py: on 1000 - 10 000 reads
        // of 1.6 - 3 kb it picks the faster form, or one within 3 %, in 54 of 54 cases
        // (profiles/r03_row_blocks_or_one_block_54_cases.txt); such reads run up to 37 % faster as blocks, config 5
        // 20.4 -> 17.3 ms.
        int64_t n_class = 0, rows_single = 0, rows_blocks[NRA_RING_MT_R + 1] = {0}, rows_class[NRA_RING_MT_R + 1] = {0};
        for (int32_t r = 0; r < n_reads; ++r) {
            const int q = pr.reads[r].qlen;
            if (kmin[r] > kmax[r] || q <= NRA_RING_MT_FROM || max_score(sc, q) > kScoreCapBit) continue;
            for (int R = NRA_RING_MT_R_MIN; R <= NRA_RING_MT_R; ++R) {
                const int64_t rows = ((int64_t)q + 64 * R - 1) / (64 * R) * (64 * R);
                rows_blocks[R] += rows;
                if (q <= NRA_MAX_QLEN_1BLOCK) rows_class[R] += rows;
            }
            if (q <= NRA_MAX_QLEN_1BLOCK) { ++n_class; rows_single += 64 * (int64_t)kRList[rows_for_qlen(q)]; }
        }
        if (n_class > 0) {
            int R = NRA_RING_MT_R;
            for (int r2 = NRA_RING_MT_R - 1; r2 >= NRA_RING_MT_R_MIN; --r2) if (rows_blocks[r2] < rows_blocks[R]) R = r2;
            const double simds = (double)device_simds(b->device);
            const double waves = (double)((n_class + 1) / 2), nblk = (double)rows_class[R] / (64.0 * R) / (double)n_class;
            const double r1 = waves / simds, e1 = r1 / std::ceil(r1);
            const double r2 = waves * nblk / (3.0 * simds);
            const double t_single = (double)rows_single / (4.3 * e1);
            const double cap = 3.7 + 0.1 * (R - NRA_RING_MT_R_MIN) + (nblk < 2.5 ? 0.4 : 0.0);
            const double t_blocks = (double)rows_class[R] / std::min(cap, 1.9 + 1.2 * r2 - 0.25 * (nblk - 2.0));
            if (t_single <= t_blocks) chain_from = NRA_MAX_QLEN_1BLOCK;
        }
    }
    std::vector<uint8_t> chained((size_t)n_reads, 0);
    for (int32_t r = 0; r < n_reads; ++r) {
        if (kmin[r] > kmax[r]) continue;
        const nra_region_t& rg = regions[pr.reads[r].region];
        const int64_t ms = max_score(sc, pr.reads[r].qlen);
        const int64_t tl = (int64_t)rg.left_len + (int64_t)rg.unit_len * kmax[r] + rg.right_len;
        chained[r] = test_chain || pr.reads[r].qlen > chain_from || ms > kScoreCapBit || tl > NRA_MAX_TLEN;
        if (brute && chained[r] && !test_chain) {
            // without the sweeps (flank-less region, unusual scoring, brute force on request) only what the
            // packed brute-force kernel holds can be scored
            if (pr.reads[r].qlen > NRA_MAX_QLEN_1BLOCK || ms > kScoreCapPk16 || tl > NRA_MAX_TLEN)
                return fail(NRA_E_RANGE, "read " + std::to_string(r) + " (" + std::to_string(pr.reads[r].qlen) +
                                             " bases) needs the junction decomposition: no brute force / ALL_EXTENTS, "
                                             "flanks >= 1, default-like scoring");
            chained[r] = 0;
        }
    }
    b->brute = brute;
    clk.mark("  candidate offsets, templates, chained flags");
    std::vector<NraSweepTask> sweep_tasks;
    uint64_t snap_total = 0;
    // the chained reads (every read with NRA_F_TEST_CHAIN): bucket kNumR = int32 cells, one read per wave;
    // bucket kNumR + 1 = packed int16, two reads per wave -- doubled scores up to 27000 (reads of up to 6750
    // bases with the default scoring) in the LDS-ring kernels, which hold units of up to NRA_SWEEP_RING_MAX_M
    // buckets kNumR + 2 + i: half-wave sweeps (k_sweep_ring32), i = index of the half's rows per lane in kRList
    std::vector<std::vector<int32_t>> by_bucket((size_t)2 * kNumR + 2);
    const bool ring_ok = (flags & NRA_F_DPP_SWEEP) == 0 && !brute;
    int chain_cols = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        if (kmin[r] > kmax[r] || pr.reads[r].qlen == 0) continue;
        int bi = rows_for_qlen(pr.reads[r].qlen);
        if (ring_ok && !chained[r] && pr.reads[r].qlen <= 32 * NRA_RING32_MAX_R &&
            regions[pr.reads[r].region].unit_len <= NRA_SWEEP_RING_MAX_M && (flags & NRA_F_NO_HALF_WAVE) == 0)
            bi = kNumR + 2 + rows_for_qlen(2 * pr.reads[r].qlen);        // 32 * R >= qlen
        if (chained[r]) {
            const nra_region_t& rg = regions[pr.reads[r].region];
            const bool packed = max_score(sc, pr.reads[r].qlen) <= kScoreCapBit && rg.unit_len <= NRA_SWEEP_RING_MAX_M &&
                                (flags & NRA_F_DPP_SWEEP) == 0;
            bi = packed ? kNumR + 1 : kNumR;
            chain_cols = std::max(chain_cols, rg.left_len + rg.unit_len * kmax[r] + rg.right_len);
        }
        read_bucket[r] = bi;
        by_bucket[bi].push_back(r);
    }
    if ((!by_bucket[kNumR].empty() || !by_bucket[kNumR + 1].empty()) && brute)
        return fail(NRA_E_RANGE, "NRA_F_TEST_CHAIN needs the junction decomposition (no brute force / ALL_EXTENTS, flanks >= 1)");
    clk.mark("  reads to buckets");
    {   // small buckets fold into the next instantiation of their own kind
        std::vector<std::vector<int32_t>> full(by_bucket.begin(), by_bucket.begin() + kNumR);
        std::vector<std::vector<int32_t>> halfb(by_bucket.begin() + kNumR + 2, by_bucket.end());
        fold_small_buckets(full, 1024, 2);    // wider folding (up to 16384 reads / 4 rows) changes nothing in 1D
        fold_small_buckets(halfb, 2048, 2);
        for (int i = 0; i < kNumR; ++i) { by_bucket[i] = std::move(full[i]); by_bucket[kNumR + 2 + i] = std::move(halfb[i]); }
    }
    if (chain_from < NRA_MAX_QLEN_1BLOCK && !test_chain && !by_bucket[kNumR + 1].empty()) {
        // ... and a handful of reads left in a one-block bucket above one row block (961 - 3072 bases, fewer than 256 of a
        // kind: 128 waves) joins the row blocks, where there are any: a length distribution that straddles
        // NRA_RING_MT_FROM would otherwise leave them a launch of their own (1000 reads of 1528 - 1582 bases, 24 of them
        // below the line: 4.2 ms against 2.7).  Larger buckets stay: two blocks pad a 1000-base read by half.
        for (int bi = 0; bi < kNumR; ++bi) {
            if (kRList[bi] <= NRA_RING_MT_R || by_bucket[bi].empty() || by_bucket[bi].size() >= 256) continue;
            for (int32_t r : by_bucket[bi]) {
                const nra_region_t& rg = regions[pr.reads[r].region];
                chain_cols = std::max(chain_cols, rg.left_len + rg.unit_len * kmax[r] + rg.right_len);
                chained[r] = 1;
                read_bucket[r] = kNumR + 1;
                by_bucket[kNumR + 1].push_back(r);
            }
            by_bucket[bi].clear();
        }
    }
    b->chain_cap = (chain_cols + 127) / 64 * 64 + 64;
    clk.mark("  small buckets folded");
    size_t strip_total = 0, chain_queue_cap = 0, mt_strip_total = 0;
    std::vector<NraChainBlock> chain_blocks;
    std::vector<NraPairTask> pair_tasks;
    std::vector<NraTask> queue_tasks;       // ALL_EXTENTS only; otherwise just capacity
    std::vector<int32_t> queue_count;
    std::vector<uint32_t> task_base;
    size_t queue_total = 0;
    int64_t alg_cells = 0;
    // longest reads first: int32 chain, packed chain, full-wave R = 48 ... 1, then the half-wave buckets 16 ... 1
    std::vector<int> bucket_order{kNumR, kNumR + 1};
    for (int i = kNumR - 1; i >= 0; --i) bucket_order.push_back(i);
    for (int i = kNumR - 1; i >= 0; --i) bucket_order.push_back(kNumR + 2 + i);
    for (const int bi : bucket_order) {
        if (by_bucket[bi].empty()) continue;
        Bucket bk;
        bk.chain = bi == kNumR || bi == kNumR + 1;
        bk.wide = bi == kNumR;
        bk.half = bi >= kNumR + 2;
        if (bk.chain) {
            // LDS-ring chain unless a unit is too long for the ring (then the DPP chain, int32 only)
            bk.ring = (flags & NRA_F_DPP_SWEEP) == 0;
            for (int32_t r : by_bucket[bi])
                if (regions[pr.reads[r].region].unit_len > NRA_SWEEP_RING_MAX_M) bk.ring = false;
            bk.payload_R = test_chain ? NRA_CHAIN_R_TEST : NRA_CHAIN_R;
            // the row blocks of a read as concurrent waves (k_sweep_ringmt), unless one read alone would need more
            // strips than the scratch budget holds (then: one wave per read, block after block, two strips)
            bk.mt = bk.ring && (flags & NRA_F_SERIAL_CHAIN) == 0;
            if (bk.mt) {
                const int rows = 64 * (test_chain ? NRA_CHAIN_R_TEST : NRA_RING_MT_R_MIN);      // (the shortest block there is)
                int qmax = 0;
                for (int32_t r : by_bucket[bi]) qmax = std::max(qmax, pr.reads[r].qlen);
                const size_t fit = (size_t)(NRA_CHAIN_SCRATCH_BUDGET / 2) / ((size_t)5 * 8 * (size_t)b->chain_cap);
                if ((size_t)((qmax + rows - 1) / rows) > fit + 1) bk.mt = false;
            }
            bk.R = test_chain ? NRA_CHAIN_R_TEST : (bk.mt ? NRA_RING_MT_R : bk.ring ? NRA_RING_CHAIN_R : NRA_CHAIN_R);
            if (bk.mt && !test_chain && !bk.wide) {
                // the block height that pads the bucket's reads least (ties: the taller block, fewer hand-offs; the int32
                // blocks of the longest reads stay at 15: 7.7 kb cores 96 ms against 102 ms at 14 with 0.5 % fewer cells)
                int64_t best = -1;
                for (int R = NRA_RING_MT_R; R >= NRA_RING_MT_R_MIN; --R) {
                    int64_t rows = 0;
                    for (int32_t r : by_bucket[bi]) rows += ((int64_t)pr.reads[r].qlen + 64 * R - 1) / (64 * R) * (64 * R);
                    if (best < 0 || rows < best) { best = rows; bk.R = R; }
                }
            }
        } else {
            bk.R = kRList[bk.half ? bi - kNumR - 2 : bi];
        }
        bk.pair_off = pair_tasks.size();
        bk.queue_off = queue_total;
        for (int32_t r : by_bucket[bi]) {
            read_bucket[r] = (int32_t)b->buckets.size();
            const NraDevRegion& d = dregs[pr.reads[r].region];
            {   // algorithmic cells: q x sum over k of (|L| + m k + |R|)
                const int64_t K = (int64_t)kmax[r] - kmin[r] + 1;
                const int64_t sum_k = ((int64_t)kmin[r] + kmax[r]) * K / 2;
                alg_cells += (int64_t)pr.reads[r].qlen * (K * ((int64_t)d.l1 + d.l3) + (int64_t)d.m1 * sum_k);
            }
            if (all_ext) {
                for (int32_t k = kmin[r]; k <= kmax[r]; ++k) {
                    queue_tasks.push_back(NraTask{r, k, 0, (int32_t)(coff[r] + (uint32_t)(k - kmin[r]))});
                    bk.cells_queue += sweep_cells(bk.R, d.l1 + d.m1 * k + d.l3);
                }
            }
            if (!all_ext && brute) {
                for (int32_t k = kmin[r]; k <= kmax[r]; k += 2) {
                    NraPairTask t{};
                    t.read = r; t.k1a = k; t.k2a = 0; t.out_a = (int32_t)(coff[r] + (uint32_t)(k - kmin[r]));
                    if (k + 1 <= kmax[r]) { t.k1b = k + 1; t.k2b = 0; t.out_b = t.out_a + 1; }
                    else { t.k1b = k; t.k2b = 0; t.out_b = -1; }
                    t.flags = 0;
                    pair_tasks.push_back(t);
                    const int tl = d.l1 + d.m1 * (t.out_b >= 0 ? k + 1 : k) + d.l3;
                    bk.cells_pair += 2 * sweep_cells(bk.R, tl);
                }
            }
            bk.queue_cap += (size_t)(kmax[r] - kmin[r] + 1);
        }
        if (!brute) {
            // pair reads of one region by length: two reads share a wave (int16 halves)
            std::vector<int32_t> order(by_bucket[bi]);
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                if (pr.reads[x].region != pr.reads[y].region) return pr.reads[x].region < pr.reads[y].region;
                return pr.reads[x].qlen > pr.reads[y].qlen;
            });
            bk.sweep_off = sweep_tasks.size();
            // the LDS-ring sweeps hold unit lengths up to NRA_SWEEP_RING_MAX_M; a bucket with a longer unit
            // keeps the DPP sweeps
            if (!bk.chain) {
                bk.ring = (flags & NRA_F_DPP_SWEEP) == 0;
                for (int32_t r : order)
                    if (dregs[pr.reads[r].region].m1 > NRA_SWEEP_RING_MAX_M) bk.ring = false;
            }
            if (bk.half) {
                // four reads of a region per wave: two pairs, one per half, the union of their windows
                for (size_t i = 0; i < order.size();) {
                    NraSweepTask t{};
                    const int32_t g = pr.reads[order[i]].region;
                    int32_t quad[4] = {-1, -1, -1, -1};
                    int nq = 0;
                    while (nq < 4 && i < order.size() && pr.reads[order[i]].region == g) quad[nq++] = order[i++];
                    t.read_a = quad[0]; t.read_b = quad[1]; t.read_c = quad[2]; t.read_d = quad[3];
                    t.kmin = kmin[quad[0]]; t.kmax = kmax[quad[0]];
                    for (int j = 1; j < nq; ++j) { t.kmin = std::min(t.kmin, kmin[quad[j]]); t.kmax = std::max(t.kmax, kmax[quad[j]]); }
                    const NraDevRegion& d = dregs[g];
                    bk.cells_sweep += (int64_t)2 * 64 * bk.R * ((d.l1 + d.m1 * t.kmax + 31 * d.m1) + (d.l3 + 31));
                    t.snap_off = snap_total;
                    snap_total += (uint64_t)NRA_SNAP_LANE_STRIDE(bk.R) * 64;       // lane-major in the half-wave kernel
                    sweep_tasks.push_back(t);
                }
            } else
            for (size_t i = 0; i < order.size();) {
                NraSweepTask t{};
                t.read_a = order[i]; t.read_b = -1; t.read_c = -1; t.read_d = -1;
                t.kmin = kmin[t.read_a]; t.kmax = kmax[t.read_a];
                // (the int32 chained sweeps take one read per wave)
                if (!bk.wide && i + 1 < order.size() && pr.reads[order[i + 1]].region == pr.reads[order[i]].region) {
                    t.read_b = order[i + 1];
                    t.kmin = std::min(t.kmin, kmin[t.read_b]);
                    t.kmax = std::max(t.kmax, kmax[t.read_b]);
                    i += 2;
                } else {
                    i += 1;
                }
                const NraDevRegion& d = dregs[pr.reads[t.read_a].region];
                int qmax = pr.reads[t.read_a].qlen;
                if (t.read_b >= 0) qmax = std::max(qmax, pr.reads[t.read_b].qlen);
                const int nblk = bk.chain ? (qmax + 64 * bk.R - 1) / (64 * bk.R) : 1;
                if (bk.ring)    // pipelines 64*m (forward) and 64 (reverse) columns deep, per row block
                    bk.cells_sweep += (int64_t)nblk * (bk.wide ? 1 : 2) * 64 * bk.R *
                                      ((d.l1 + d.m1 * t.kmax + 63 * d.m1) + (d.l3 + 63));
                else
                    bk.cells_sweep += (int64_t)nblk * (bk.wide ? 1 : 2) * (sweep128_cells(bk.R, d.l1 + d.m1 * t.kmax) +
                                                                           sweep128_cells(bk.R, d.l3));
                t.snap_off = snap_total;
                snap_total += (uint64_t)nblk * 3 * bk.R * 64;
                sweep_tasks.push_back(t);
            }
            bk.n_sweep = (int)(sweep_tasks.size() - bk.sweep_off);
            if (bk.chain && bk.mt) {
                // launch groups: as many tasks as the strip budget holds (one strip between consecutive blocks of a
                // task), each group's (task, block) list block-major: all blocks 0, then all blocks 1, ...
                const size_t fit = std::max<size_t>(1, (size_t)(NRA_CHAIN_SCRATCH_BUDGET / 2) / ((size_t)5 * 8 * (size_t)b->chain_cap));
                const int rows = 64 * bk.R;
                auto nblk_of = [&](const NraSweepTask& t) {
                    int qmax = pr.reads[t.read_a].qlen;
                    if (t.read_b >= 0) qmax = std::max(qmax, pr.reads[t.read_b].qlen);
                    return std::max(1, (qmax + rows - 1) / rows);
                };
                size_t most_strips = 0;
                for (int t0 = 0; t0 < bk.n_sweep;) {
                    size_t used = 0;
                    int t1 = t0, max_blk = 0;
                    std::vector<size_t> base;
                    while (t1 < bk.n_sweep) {
                        const int nb2 = nblk_of(sweep_tasks[bk.sweep_off + (size_t)t1]);
                        if (t1 > t0 && used + (size_t)(nb2 - 1) > fit) break;
                        base.push_back(used);
                        used += (size_t)(nb2 - 1);
                        max_blk = std::max(max_blk, nb2);
                        ++t1;
                    }
                    const size_t first_block = chain_blocks.size();
                    for (int blk = 0; blk < max_blk; ++blk)
                        for (int t = t0; t < t1; ++t) {
                            const int nb2 = nblk_of(sweep_tasks[bk.sweep_off + (size_t)t]);
                            if (blk >= nb2) continue;
                            const int32_t sb = (int32_t)base[(size_t)(t - t0)];
                            chain_blocks.push_back(NraChainBlock{t, blk, nb2, blk > 0 ? sb + blk - 1 : -1,
                                                                 blk < nb2 - 1 ? sb + blk : -1});
                        }
                    bk.mt_groups.push_back({first_block, (int)(chain_blocks.size() - first_block)});
                    most_strips = std::max(most_strips, used);
                    t0 = t1;
                }
                bk.n_strips = (int)most_strips;
                bk.strip_off = mt_strip_total;                     // (in strips of 5 * chain_cap granules)
                mt_strip_total += most_strips;
                chain_queue_cap = std::max(chain_queue_cap, bk.queue_cap);
            } else if (bk.chain) {
                // (two chained buckets at most -- packed and int32 -- share the budget)
                bk.n_strips = chain_strips((size_t)bk.n_sweep, bk.ring ? NRA_RING_CHAIN_STRIPS : NRA_CHAIN_STRIPS,
                                           (size_t)10 * 4 * (size_t)b->chain_cap, NRA_CHAIN_SCRATCH_BUDGET / 2);
                bk.strip_off = strip_total;
                strip_total += (size_t)bk.n_strips * 10 * (size_t)b->chain_cap;
                chain_queue_cap = std::max(chain_queue_cap, bk.queue_cap);
            }
        }
        bk.n_pair = (int)(pair_tasks.size() - bk.pair_off);
        bk.n_queue = all_ext ? (int)bk.queue_cap : 0;
        queue_total += bk.queue_cap;
        queue_count.push_back(bk.n_queue);
        task_base.push_back((uint32_t)bk.queue_off);
        b->buckets.push_back(bk);
    }
    const size_t nb = b->buckets.size();
    clk.mark("templates, buckets, tasks");
    packing.join();
    b->has_n = (has_n || pr.has_n) ? 1 : 0;
    clk.mark("2-bit packing (rest)");

    // one chunk for everything: ~7 B per read base, ~40 B per candidate, the tasks, the chain strips
    b->arena.expect(pr.q2bit.size() * 16 * 2 + (size_t)snap_total * 4 + (size_t)total * 40 + pool.size() + sweep_tasks.size() * sizeof(NraSweepTask) +
                    pair_tasks.size() * sizeof(NraPairTask) + (size_t)n_reads * 64 + (4u << 20));
    // (the chain strips get chunks of their own: they can be GBs)
    UploadBatchScope uploads;            // every upload below is enqueued; one wait behind the last of them
    HIP_TRY(b->pool.upload(pool));
    HIP_TRY(b->q2bit.upload(pr.q2bit));
    HIP_TRY(b->qnmask.upload(pr.nmask));
    HIP_TRY(b->regions.upload(dregs));
    HIP_TRY(b->reads.upload(pr.reads));
    HIP_TRY(b->pair_tasks.upload(pair_tasks));
    HIP_TRY(b->sweep_tasks.upload(sweep_tasks));
    if (!brute) {
        HIP_TRY(b->snap.alloc((size_t)snap_total));
        HIP_TRY(b->read_a1d.alloc((size_t)n_reads));
        // Quanta (k_sweep_ringq): for the unchained LDS-ring buckets of a batch whose tasks are few against the wave slots --
        // kernel-length tasks then end a launch with SIMDs holding 3 or 4 of them while others hold 2 or 3; with tens of
        // tasks per slot (BASELINE config 4) two launches already run at 0.98 of the issue ceiling and the states at the
        // cut (NRA_QSTATE_INTS(R) x 256 B a task) would only be traffic.
        std::vector<uint32_t> qlist;
        size_t q_tasks = 0, q_state = 0, n_q_tasks_all = 0;
        for (const Bucket& bk : b->buckets) if (bk.ring && !bk.chain && !bk.mt) n_q_tasks_all += (size_t)bk.n_sweep;
        // Parts of NRA_Q_STEPS = 384 steps.  Measured on one box, ms per step (config 2 | 5000 reads | 20 000 reads of it):
        // no quanta 5.76 | 4.21 | 10.74; parts of 2048 (whole sweeps) 6.03; 1024 5.72; 768 5.32 | 3.39 | 9.93; 512 5.35 | 3.35 |
        // 9.95; 448 5.25; 384 5.22 | 3.20 | 9.93; 320 5.33; 256 5.46 | 3.58 | 10.43; 192 5.55; 128 5.54.  With more than
        // 8 tasks per SIMD the parts lose to two launches (40 000 reads 19.9 -> 20.2 ms, config 4 279 -> 285): the rule above.
        // (NRA_TEST_QSTEPS overrides the part size for such measurements and for the tests.)
        // Fewer tasks than SIMDs: a call's time is one task's chain of parts, every cut ~20 us of it (1000 reads: 1.99 -> 2.19 ms
        // per step with parts of 384): a reverse sweep, a forward sweep to its first cut and the rest then, as in this round's
        // first form.
        int q_steps_wanted = n_q_tasks_all < (size_t)device_simds(device) ? NRA_Q_WHOLE : NRA_Q_STEPS;
        if (const char* e = getenv("NRA_TEST_QSTEPS")) q_steps_wanted = std::max(64, atoi(e) / 64 * 64);
        const bool want_quanta = (flags & (NRA_F_NO_QUANTA | NRA_F_DPP_SWEEP)) == 0 && n_q_tasks_all > 0 &&
                                 n_q_tasks_all <= (size_t)8 * (size_t)device_simds(device);
        for (Bucket& bk : b->buckets) {
            if (!want_quanta || !bk.ring || bk.chain || bk.mt || bk.n_sweep <= 0) continue;
            bk.quanta = true;
            bk.q_off = qlist.size(); bk.q_task_off = q_tasks; bk.q_state_off = q_state;
            // Parts of `q_steps` steps (a multiple of 64; longer for sweeps that would make more than NRA_Q_MAX_PARTS of them).
            // Ticket order: every reverse sweep's part 0, task by task, then every part 1, ..., then the forward sweeps' parts
            // the same way -- a part's producers (the part before it; for a forward part at or behind the first boundary step
            // the task's whole reverse sweep) always hold smaller tickets.
            const NraSweepTask* bt = sweep_tasks.data() + bk.sweep_off;
            int max_steps = 0;
            std::vector<int> steps_rev((size_t)bk.n_sweep), steps_fwd((size_t)bk.n_sweep), cut_fwd((size_t)bk.n_sweep);
            for (int t = 0; t < bk.n_sweep; ++t) {
                const NraDevRegion& d = dregs[pr.reads[bt[t].read_a].region];
                steps_rev[(size_t)t] = NRA_Q_STEPS_REV(d.l3, bk.half);
                steps_fwd[(size_t)t] = NRA_Q_STEPS_FWD(d.l1, d.m1, bt[t].kmax, bk.half);
                cut_fwd[(size_t)t] = NRA_Q_CUT(d.l1 + d.m1 * bt[t].kmin - 1);
                max_steps = std::max(max_steps, std::max(steps_rev[(size_t)t], steps_fwd[(size_t)t]));
            }
            bk.q_steps = std::max(q_steps_wanted, ((max_steps + NRA_Q_MAX_PARTS - 2) / (NRA_Q_MAX_PARTS - 1) + 63) / 64 * 64);
            for (int dir = 0; dir < 2; ++dir)
                for (int part = 0; part < NRA_Q_MAX_PARTS; ++part) {
                    bool any = false;
                    for (int t = 0; t < bk.n_sweep; ++t) {
                        // parts of this sweep: before the first cut (forward only), and from it on
                        const int cut = dir ? cut_fwd[(size_t)t] : 0, steps = dir ? steps_fwd[(size_t)t] : steps_rev[(size_t)t];
                        const int n_parts = (cut + bk.q_steps - 1) / bk.q_steps + std::max(1, (steps - cut + bk.q_steps - 1) / bk.q_steps);
                        if (part < n_parts) {
                            qlist.push_back(((uint32_t)dir << 31) | ((uint32_t)part << NRA_Q_PART_SHIFT) | (uint32_t)t);
                            any = true;
                        }
                    }
                    if (!any) break;
                }
            bk.n_quanta = (int)(qlist.size() - bk.q_off);
            q_tasks += 2 * (size_t)bk.n_sweep;
            int max_rev = 0;
            for (int v : steps_rev) max_rev = std::max(max_rev, v);
            q_state += (max_rev > bk.q_steps ? 2 : 1) * (size_t)bk.n_sweep * NRA_QSTATE_INTS(bk.R) * 64;
        }
        if (!qlist.empty()) {
            HIP_TRY(b->q_list.upload(qlist));
            b->q_arrivals_off = (1 + b->buckets.size() + 3) / 4 * 4;
            b->q_words_n = b->q_arrivals_off + q_tasks;
            HIP_TRY(b->q_words.alloc(b->q_words_n));
            HIP_TRY(b->q_state.alloc(q_state));
        }
    }
    HIP_TRY(b->cand_flag.alloc((size_t)total));
    if (!chain_blocks.empty()) {
        HIP_TRY(b->chain_blocks.upload(chain_blocks));
        const size_t granules = std::max<size_t>(mt_strip_total, 1) * 5 * (size_t)b->chain_cap;
        HIP_TRY(b->mt_strips.alloc(granules));
        // (on the batch's own stream: the bucket streams are ordered behind it by fork_ev; a null-stream memset is not
        // ordered with these non-blocking streams at all)
        HIP_TRY(hipMemsetAsync(b->mt_strips.p, 0, granules * 8, b->stream));       // no granule carries an epoch yet (epochs are never 0)
        HIP_TRY(b->mt_words.alloc(2 + b->buckets.size()));
        HIP_TRY(hipMemsetAsync(b->mt_words.p, 0, (2 + b->buckets.size()) * 4, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    if (strip_total || !chain_blocks.empty()) {
        if (strip_total) HIP_TRY(b->chain_sweep.alloc(strip_total));
        // the chained buckets' extents launches run in turn and share these strips (int64 cells)
        b->payload_strips = chain_strips(chain_queue_cap, NRA_CHAIN_STRIPS, (size_t)6 * 8 * (size_t)b->chain_cap);
        HIP_TRY(b->chain_payload.alloc((size_t)b->payload_strips * 6 * (size_t)b->chain_cap));
    }
    if (all_ext) HIP_TRY(b->queue_tasks.upload(queue_tasks));
    else HIP_TRY(b->queue_tasks.alloc(queue_total));
    HIP_TRY(b->queue_count.upload(queue_count));
    HIP_TRY(b->tie_count.alloc(nb));
    HIP_TRY(b->bucket_task_base.upload(task_base));
    {
        std::vector<int32_t> v(kmin, kmin + n_reads); HIP_TRY(b->kmin.upload(v));
        std::vector<int32_t> w(kmax, kmax + n_reads); HIP_TRY(b->kmax.upload(w));
    }
    HIP_TRY(b->read_bucket.upload(read_bucket));
    HIP_TRY(b->coff.upload(coff));
    HIP_TRY(b->cand_score.alloc((size_t)total));
    HIP_TRY(b->cand_tstart.alloc((size_t)total));
    HIP_TRY(b->cand_tend.alloc((size_t)total));
    rc = alloc_results(b, (size_t)n_reads, false);
    if (rc) return rc;
    HIP_TRY(upload_batch_end());
    clk.mark("device buffers, H2D");
    rc = make_events(b, 2 + 6 * (int)nb + 2);
    if (rc) return rc;
    if (!brute) {
        b->bstreams.assign(nb, nullptr); b->bdone.assign(nb, nullptr);
        for (size_t i = 0; i < nb; ++i) {
            HIP_TRY(g_handles.stream(b->device, &b->bstreams[i]));
            HIP_TRY(g_handles.event(b->device, false, &b->bdone[i]));
        }
        HIP_TRY(g_handles.event(b->device, false, &b->fork_ev));
    }

    clk.mark("events, streams");
    b->stats.n_alignments = total;
    b->stats.algorithmic_cells = alg_cells;
    int64_t ex = 0;
    for (const Bucket& bk : b->buckets) ex += bk.cells_pair + bk.cells_queue + bk.cells_sweep;
    b->stats.executed_cells = ex;
    b->stats.algorithmic_bytes = (int64_t)pr.q2bit.size() * 4 + (int64_t)pool.size() + total * 4 + (int64_t)n_reads * 17;
    b->stats.intermediate_bytes = brute ? 0 : 2 * ((int64_t)snap_total + (int64_t)b->q_state.n) * 4;   // the junction snapshot and the wave states at the cut: written, then read
    *out = guard.release();
    return NRA_OK;
}

static int run_1d(nra_batch* b)
{
    hipStream_t st = b->stream;
    const size_t nb = b->buckets.size();
    const bool all_ext = (b->flags & NRA_F_ALL_EXTENTS) != 0;
    const size_t nc = (size_t)b->n_cands;
    HIP_TRY(hipEventRecord(b->ev[0], st));
    HIP_TRY(hipMemsetAsync(b->cand_score.p, 0xff, std::max<size_t>(nc, 1) * 4, st));
    HIP_TRY(hipMemsetAsync(b->cand_tstart.p, 0xff, std::max<size_t>(nc, 1) * 4, st));
    HIP_TRY(hipMemsetAsync(b->cand_tend.p, 0xff, std::max<size_t>(nc, 1) * 4, st));
    HIP_TRY(hipMemsetAsync(b->tie_count.p, 0, std::max<size_t>(nb, 1) * 4, st));
    HIP_TRY(hipMemsetAsync(b->cand_flag.p, 2, std::max<size_t>(nc, 1), st));   // 2 = "needs the extents DP"
    if (b->chain_blocks.n > 0) {
        // concurrent row blocks: every block adds its maximum to the read's A (atomicMax; A >= 0); the give-up word
        HIP_TRY(hipMemsetAsync(b->read_a1d.p, 0, std::max<size_t>((size_t)b->n_reads, 1) * 4, st));
        // (NRA_TEST_MT_GIVEUP in the environment, tests only: the run starts with the give-up word set, as if a wave had
        // timed out -- every waiting wave leaves at its next look and the fetch reports NRA_E_DEVICE)
        HIP_TRY(hipMemsetAsync(b->mt_words.p, getenv("NRA_TEST_MT_GIVEUP") ? 1 : 0, 4, st));
        b->mt_checked = false;
    }
    if (b->q_words_n > 0) {
        // the sweeps in quanta: tickets and arrival counters start at 0; the give-up word too (NRA_TEST_MT_GIVEUP: set)
        HIP_TRY(hipMemsetAsync(b->q_words.p, 0, b->q_words_n * 4, st));
        if (getenv("NRA_TEST_MT_GIVEUP")) HIP_TRY(hipMemsetAsync(b->q_words.p, 1, 4, st));
        b->q_checked = false;
    }
    int ev = 2;
    b->n_score_ev = 0; b->n_ext_ev = 0;
    const int max_waves = 256 * 16;
    const bool tie_ext = (b->flags & NRA_F_TIE_EXTENTS) != 0;
    HIP_TRY(hipEventRecord(b->phase_ev[0], st));
    if (!b->brute) {
        // junction decomposition: per bucket a chain reverse sweep -> forward sweep, each chain on
        // its own stream so that short buckets fill the SIMDs a long bucket's tail leaves idle
        HIP_TRY(hipEventRecord(b->fork_ev, st));
        // (the order the buckets' chains are launched in makes no difference: 5.50 - 5.61 ms either way on config 2;
        // a bucket's tasks as 2 / 3 / 4 groups with chains of their own are slower: 5.65 -> 6.0 / 5.9 / 7.6 ms)
        std::vector<size_t> order;
        for (size_t i = 0; i < nb; ++i) if (b->buckets[i].mt) order.push_back(i);
        for (size_t i = 0; i < nb; ++i) if (!b->buckets[i].mt) order.push_back(i);
        bool mt_rev_recorded = false;
        // Row-block buckets beside quanta: the launch that is being dispatched keeps the slots its own waves free, so the
        // quanta, started together with the row blocks' reverse launch, kept the forward launch of the row blocks off the
        // device until they were through (config 5: it began at 5.7 of 17.7 ms and then ran alone).  Where the row blocks
        // are the larger share of the work the quanta wait for the reverse launch instead and share the device with the
        // forward one: 17.9 -> 17.35 ms per step.
        int64_t cells_mt = 0, cells_q = 0;
        for (const Bucket& bk : b->buckets) (bk.mt ? cells_mt : cells_q) += bk.quanta || bk.mt ? bk.cells_sweep : 0;
        const bool mt_first = cells_mt > 0 && cells_q > 0 && cells_mt >= cells_q;
        for (size_t oi = 0; oi < nb; ++oi) {
            const size_t i = order[oi];
            const Bucket& bk = b->buckets[i];
            hipStream_t q = b->bstreams[i];
            HIP_TRY(hipStreamWaitEvent(q, b->fork_ev, 0));
            if (mt_first && bk.quanta && mt_rev_recorded) HIP_TRY(hipStreamWaitEvent(q, b->mt_rev_ev, 0));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            int32_t* strips = bk.chain ? b->chain_sweep.p + bk.strip_off : nullptr;
            if (bk.quanta) {
                // one launch: the sweeps' parts, taken by ticket (two timing pairs like the two launches it replaces: the
                // second one is empty)
                auto launch = bk.half ? nra_launch_sweep_ringq32 : nra_launch_sweep_ringq;
                LAUNCH_TRY(launch(bk.R, b->has_n, q, bk.n_quanta, b->q_list.p + bk.q_off, bk.q_steps, bk.n_sweep, b->q_words.p + 1 + i,
                                  b->q_words.p + b->q_arrivals_off + bk.q_task_off, b->q_words.p, b->q_state.p + bk.q_state_off,
                                  b->sweep_tasks.p + bk.sweep_off, b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                  b->sp, b->kmin.p, b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p, b->cand_score.p,
                                  b->cand_flag.p));
                HIP_TRY(hipEventRecord(b->ev[ev++], q));
                HIP_TRY(hipEventRecord(b->ev[ev++], q));
                HIP_TRY(hipEventRecord(b->ev[ev++], q));
                b->n_score_ev += 2;
                HIP_TRY(hipEventRecord(b->bdone[i], q));
                HIP_TRY(hipStreamWaitEvent(st, b->bdone[i], 0));
                continue;
            }
            if (bk.mt) {
                for (const auto& g : bk.mt_groups)
                    LAUNCH_TRY(nra_launch_sweep_ringmt_bwd(bk.R, b->has_n, bk.wide ? 1 : 0, q, g.second, b->chain_blocks.p + g.first,
                                                           b->mt_words.p + 2 + i, b->sweep_tasks.p + bk.sweep_off, b->reads.p,
                                                           b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp, b->kmin.p,
                                                           b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p,
                                                           b->mt_strips.p + bk.strip_off * 5 * (size_t)b->chain_cap, b->chain_cap,
                                                           next_epoch(), b->mt_words.p));
            } else if (bk.ring && bk.chain)
                LAUNCH_TRY(nra_launch_sweep_ringchain_bwd(bk.R, b->has_n, bk.wide ? 1 : 0, q, bk.n_sweep,
                                                          b->sweep_tasks.p + bk.sweep_off, b->reads.p, b->regions.p, b->pool.p,
                                                          b->q2bit.p, b->qnmask.p, b->sp, b->kmin.p, b->kmax.p, b->coff.p,
                                                          b->snap.p, b->read_a1d.p, strips, b->chain_cap, bk.n_strips));
            else if (bk.half)
                LAUNCH_TRY(nra_launch_sweep_ring32_bwd(bk.R, b->has_n, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                       b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                       b->kmin.p, b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p));
            else if (bk.ring)
                LAUNCH_TRY(nra_launch_sweep_ring_bwd(bk.R, b->has_n, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                     b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                     b->kmin.p, b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p));
            else
                LAUNCH_TRY(nra_launch_sweep_bwd(bk.R, b->has_n, bk.chain ? 1 : 0, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                b->kmin.p, b->kmax.p, b->coff.p, b->snap.p,
                                                b->read_a1d.p, strips, b->chain_cap, bk.n_strips));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            if (bk.mt && mt_first) {
                if (!b->mt_rev_ev) HIP_TRY(g_handles.event(b->device, false, &b->mt_rev_ev));
                HIP_TRY(hipEventRecord(b->mt_rev_ev, q));
                mt_rev_recorded = true;
            }
            if (bk.mt) {
                for (const auto& g : bk.mt_groups)
                    LAUNCH_TRY(nra_launch_sweep_ringmt_fwd(bk.R, b->has_n, bk.wide ? 1 : 0, q, g.second, b->chain_blocks.p + g.first,
                                                           b->mt_words.p + 2 + i, b->sweep_tasks.p + bk.sweep_off, b->reads.p,
                                                           b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp, b->kmin.p,
                                                           b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p, b->cand_score.p,
                                                           b->cand_flag.p,
                                                           b->mt_strips.p + bk.strip_off * 5 * (size_t)b->chain_cap, b->chain_cap,
                                                           next_epoch(), b->mt_words.p));
            } else if (bk.ring && bk.chain)
                LAUNCH_TRY(nra_launch_sweep_ringchain_fwd(bk.R, b->has_n, bk.wide ? 1 : 0, q, bk.n_sweep,
                                                          b->sweep_tasks.p + bk.sweep_off, b->reads.p, b->regions.p, b->pool.p,
                                                          b->q2bit.p, b->qnmask.p, b->sp, b->kmin.p, b->kmax.p, b->coff.p,
                                                          b->snap.p, b->read_a1d.p, b->cand_score.p, b->cand_flag.p,
                                                          strips, b->chain_cap, bk.n_strips));
            else if (bk.half)
                LAUNCH_TRY(nra_launch_sweep_ring32_fwd(bk.R, b->has_n, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                       b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                       b->kmin.p, b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p,
                                                       b->cand_score.p, b->cand_flag.p));
            else if (bk.ring)
                LAUNCH_TRY(nra_launch_sweep_ring_fwd(bk.R, b->has_n, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                     b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                     b->kmin.p, b->kmax.p, b->coff.p, b->snap.p, b->read_a1d.p,
                                                     b->cand_score.p, b->cand_flag.p));
            else
                LAUNCH_TRY(nra_launch_sweep_fwd(bk.R, b->has_n, bk.chain ? 1 : 0, q, bk.n_sweep, b->sweep_tasks.p + bk.sweep_off,
                                                b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp,
                                                b->kmin.p, b->kmax.p, b->coff.p, b->snap.p,
                                                b->read_a1d.p, b->cand_score.p, b->cand_flag.p,
                                                strips, b->chain_cap, bk.n_strips));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            b->n_score_ev += 2;
            HIP_TRY(hipEventRecord(b->bdone[i], q));
            HIP_TRY(hipStreamWaitEvent(st, b->bdone[i], 0));
        }
    } else {
        for (size_t i = 0; i < nb; ++i) {
            const Bucket& bk = b->buckets[i];
            HIP_TRY(hipEventRecord(b->ev[ev++], st));
            if (!all_ext) {
                LAUNCH_TRY(nra_launch_score_pk16(bk.R, b->has_n, st, bk.n_pair, b->pair_tasks.p + bk.pair_off,
                                                 b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                 b->sp, b->cand_score.p));
            } else {
                LAUNCH_TRY(nra_launch_payload_origin(bk.R, b->has_n, st, std::min(bk.n_queue, max_waves),
                                                  b->queue_tasks.p + bk.queue_off, b->queue_count.p + i,
                                                  b->reads.p, b->regions.p, b->pool.p,
                                                  b->q2bit.p, b->qnmask.p, b->sp, b->cand_score.p,
                                                  b->cand_tstart.p, b->cand_tend.p, nullptr, 0, 0));
            }
            HIP_TRY(hipEventRecord(b->ev[ev++], st));
            b->n_score_ev++;
        }
    }
    HIP_TRY(hipEventRecord(b->phase_ev[1], st));
    // ties that need the explicit extents DP: all of them (brute force / TIE_EXTENTS), or only those
    // whose three-score flank verdict is ambiguous
    const int append_mode = all_ext ? 0 : ((b->brute || tie_ext) ? 2 : 1);
    LAUNCH_TRY(nra_launch_select_best_1d(st, b->n_reads, b->kmin.p, b->kmax.p, b->coff.p, b->cand_score.p,
                                         b->cand_flag.p, b->read_bucket.p, b->bucket_task_base.p, append_mode,
                                         b->queue_tasks.p, b->tie_count.p, b->best_score.p));
    if (!all_ext) {
        for (size_t i = 0; i < nb; ++i) {
            const Bucket& bk = b->buckets[i];
            HIP_TRY(hipEventRecord(b->ev[ev++], st));
            // chained reads: int64 cells (scores and extents of any size)
            LAUNCH_TRY(nra_launch_payload_origin(bk.chain ? bk.payload_R : bk.R, b->has_n, st,
                                              (int)std::min<size_t>(bk.queue_cap, bk.chain ? (size_t)b->payload_strips : (size_t)max_waves),
                                              b->queue_tasks.p + bk.queue_off, b->tie_count.p + i,
                                              b->reads.p, b->regions.p, b->pool.p,
                                              b->q2bit.p, b->qnmask.p, b->sp, b->cand_score.p,
                                              b->cand_tstart.p, b->cand_tend.p,
                                              bk.chain ? b->chain_payload.p : nullptr, b->chain_cap, bk.chain ? 1 : 0));
            HIP_TRY(hipEventRecord(b->ev[ev++], st));
            b->n_ext_ev++;
        }
    }
    LAUNCH_TRY(nra_launch_select_final_1d(st, b->n_reads, b->kmin.p, b->kmax.p, b->coff.p, b->reads.p,
                                          b->regions.p, b->cand_score.p, b->cand_flag.p, b->cand_tstart.p,
                                          b->cand_tend.p, b->best_score.p, b->sum_k.p, b->n_ties.p, b->status.p));
    HIP_TRY(hipEventRecord(b->ev[1], st));
    return NRA_OK;
}

// After a run with concurrent-block sweeps: did a wave give up waiting for the block above it?  (A hang guard that
// should never fire; if it does the results of this run are not to be used.)
static int check_mt(nra_batch* b)
{
    if (!b->q_checked && b->q_words_n > 0) {
        int32_t gave_up = 0;
        HIP_TRY(hipMemcpy(&gave_up, b->q_words.p, 4, hipMemcpyDeviceToHost));
        b->q_checked = true;
        if (gave_up) return fail(NRA_E_DEVICE, "sweep in quanta: a second part timed out waiting for its reverse sweep / first part");
    }
    if (b->mt_checked || b->chain_blocks.n == 0) return NRA_OK;
    int32_t failed = 0;
    HIP_TRY(hipMemcpy(&failed, b->mt_words.p, 4, hipMemcpyDeviceToHost));
    b->mt_checked = true;
    if (failed) return fail(NRA_E_DEVICE, "chained sweep: a row block timed out waiting for the block above it");
    return NRA_OK;
}

int nra_batch1d_fetch(nra_batch_t* b, int32_t* best_score, int64_t* sum_k, int32_t* n_ties,
                      uint8_t* status, int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend)
{
    if (!b || b->kind != 1) return fail(NRA_E_ARG, "not a 1D batch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    {
        const int rcm = check_mt(b);
        if (rcm) return rcm;
    }
    // (after a refinement the per-cell arrays are the refinement's: (2 buf1)(2 buf2) entries a read)
    const size_t n = (size_t)b->n_reads, nc = (size_t)(b->refined ? b->rf_n_cands : b->n_cands);
    const int32_t* src_score = b->refined ? b->rf_cell_score.p : b->cand_score.p;
    const int32_t* src_wscore = b->refined ? b->rf_cell_wscore.p : b->cand_tstart.p;
    if (n) {
        int rc = fetch_results(b);
        if (rc) return rc;
        if (best_score) memcpy(best_score, staged(b, b->best_score), n * 4);
        if (sum_k) memcpy(sum_k, staged(b, b->sum_k), n * 8);
        if (n_ties) memcpy(n_ties, staged(b, b->n_ties), n * 4);
        if (status) memcpy(status, staged(b, b->status), n);
        if (best_score)
            for (size_t i = 0; i < n; ++i) if (best_score[i] < 0) best_score[i] = 0;
    }
    if (nc) {
        if (cand_score) HIP_TRY(copy_d2h(cand_score, b->cand_score.p, nc * 4));
        if (cand_tstart) HIP_TRY(copy_d2h(cand_tstart, b->cand_tstart.p, nc * 4));
        if (cand_tend) HIP_TRY(copy_d2h(cand_tend, b->cand_tend.p, nc * 4));
    }
    return NRA_OK;
}

// A large many-region call in region blocks: block i + 1 is packed, bucketed and uploaded (host work + H2D) while
// the kernels of block i run, so that the host's share of the call (config 4: ~90 of ~345 ms) hides behind the
// device's.  Reads of a region stay in one block (the sweeps pair two or four reads of a region per wave); needs
// the reads grouped by region (read_region non-decreasing), which is how the host mirror lists them.
#define NRA_STREAM_MIN_READS 131072
static int round3_1d_streamed(int device, const nra_region_t* regions, int32_t n_regions, int32_t n_reads,
                              const char* seqs, const int64_t* seq_off, const int32_t* read_region,
                              const int32_t* kmin, const int32_t* kmax, const nra_scoring_t* sc, int32_t flags,
                              int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                              int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend)
{
    // Up to 8 blocks of equal size.  Measured on config 4 (1 M reads, one box, ms per call): 16 blocks 350 -- every launch a
    // sixteenth of its bucket --, 8 blocks 304-314, 4 blocks 300-312, 2 blocks 330-337 (the first block's host work is not
    // hidden), blocks of doubling size 353-359 (the last block's device chunk is too large for the chunk cache: a
    // hipMalloc / hipFree of several GB per call); the resident batch's step is 281.
    const int n_blocks = (int)std::min<int64_t>(8, std::max<int64_t>(2, n_reads / 120000));
    struct Block { int32_t r0, r1, g0, g1; int64_t c0; };
    std::vector<Block> blocks;
    {
        int64_t cands = 0;
        int32_t r = 0;
        for (int i = 0; i < n_blocks && r < n_reads; ++i) {
            Block bl{r, r, read_region[r], 0, cands};
            const int32_t want = (int32_t)((int64_t)n_reads * (i + 1) / n_blocks);
            while (r < n_reads && (r < want || read_region[r] == read_region[r - 1])) {      // whole regions only
                cands += std::max<int64_t>(0, (int64_t)kmax[r] - kmin[r] + 1);
                ++r;
            }
            if (i == n_blocks - 1) for (; r < n_reads; ++r) cands += std::max<int64_t>(0, (int64_t)kmax[r] - kmin[r] + 1);
            bl.r1 = r; bl.g1 = read_region[r - 1] + 1;
            if (bl.r1 > bl.r0) blocks.push_back(bl);
        }
    }
    std::vector<nra_batch_t*> live(blocks.size(), nullptr);
    std::vector<std::vector<int32_t>> local_region(blocks.size());
    int rc = NRA_OK;
    auto finish = [&](size_t i) {
        if (!live[i]) return;
        const Block& bl = blocks[i];
        if (!rc) rc = nra_batch_sync(live[i]);
        if (!rc) rc = nra_batch1d_fetch(live[i], best_score + bl.r0, sum_k + bl.r0, n_ties + bl.r0, status + bl.r0,
                                        cand_score ? cand_score + bl.c0 : nullptr, cand_tstart ? cand_tstart + bl.c0 : nullptr,
                                        cand_tend ? cand_tend + bl.c0 : nullptr);
        nra_batch_destroy(live[i]);
        live[i] = nullptr;
    };
    for (size_t i = 0; i < blocks.size() && !rc; ++i) {
        const Block& bl = blocks[i];
        std::vector<int32_t>& lr = local_region[i];
        lr.resize((size_t)(bl.r1 - bl.r0));
        for (int32_t r = bl.r0; r < bl.r1; ++r) lr[(size_t)(r - bl.r0)] = read_region[r] - bl.g0;
        rc = nra_batch1d_create(device, regions + bl.g0, bl.g1 - bl.g0, bl.r1 - bl.r0, seqs, seq_off + bl.r0, lr.data(),
                                kmin + bl.r0, kmax + bl.r0, sc, flags, &live[i]);
        if (!rc) rc = nra_batch_run(live[i]);
        if (i >= 2) finish(i - 2);              // at most three blocks alive: running, queued, being built
    }
    for (size_t i = 0; i < blocks.size(); ++i) finish(i);
    return rc;
}

int nra_round3_1d(int device, const nra_region_t* regions, int32_t n_regions, int32_t n_reads,
                  const char* seqs, const int64_t* seq_off, const int32_t* read_region,
                  const int32_t* kmin, const int32_t* kmax, const nra_scoring_t* sc, int32_t flags,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend)
{
    if (n_reads > 0 && (!best_score || !sum_k || !n_ties || !status)) return fail(NRA_E_ARG, "NULL output array");
    nra_batch_t* b = nullptr;
    if (cand_tstart || cand_tend) flags |= NRA_F_TIE_EXTENTS;
    if (n_reads >= NRA_STREAM_MIN_READS && n_regions > 1 && read_region && regions && seqs && seq_off && kmin && kmax) {
        bool grouped = true;
        for (int32_t r = 0; r < n_reads && grouped; ++r)
            grouped = read_region[r] >= 0 && read_region[r] < n_regions && (r == 0 || read_region[r] >= read_region[r - 1]);
        if (grouped)
            return round3_1d_streamed(device, regions, n_regions, n_reads, seqs, seq_off, read_region, kmin, kmax, sc,
                                      flags, best_score, sum_k, n_ties, status, cand_score, cand_tstart, cand_tend);
    }
    int rc = nra_batch1d_create(device, regions, n_regions, n_reads, seqs, seq_off, read_region, kmin,
                                kmax, sc, flags, &b);
    if (rc) return rc;
    rc = nra_batch_run(b);
    if (!rc) rc = nra_batch_sync(b);
    if (!rc) rc = nra_batch1d_fetch(b, best_score, sum_k, n_ties, status, cand_score, cand_tstart, cand_tend);
    nra_batch_destroy(b);
    return rc;
}

// ---- 2D -----------------------------------------------------------------------------
static int account_run(nra_batch* b);
static int account_run_fwd(nra_batch* b) { return account_run(b); }

int nra_batch2d_create_reads(int device, const nra_joint_region_t* reg, int32_t n_reads, const char* seqs,
                             const int64_t* seq_off, const nra_scoring_t* sc, int32_t flags, nra_batch_t** out)
{
    if (!out) return fail(NRA_E_ARG, "out is NULL");
    *out = nullptr;
    if (!reg || n_reads < 0) return fail(NRA_E_ARG, "bad region / counts");
    if (n_reads > 0 && (!seqs || !seq_off)) return fail(NRA_E_ARG, "NULL input array");
    if (!scoring_ok(sc)) return fail(NRA_E_ARG, "scoring parameters out of range");
    if (reg->left_len < 0 || reg->right_len < 0 || reg->mid_len < 0 || reg->unit1_len <= 0 ||
        reg->unit2_len <= 0 || !reg->unit1 || !reg->unit2)
        return fail(NRA_E_ARG, "bad joint region");

    nra_batch* b = new nra_batch();
    std::unique_ptr<nra_batch> guard(b);
    ArenaScope arena_scope(&b->arena);
    b->kind = 2; b->n_reads = n_reads; b->n_regions = 1; b->n_cands = 0;
    int rc = common_init(b, device, sc, flags);
    if (rc) return rc;
    b->scoring = *sc;
    b->jr_left.assign(reg->left ? reg->left : "", (size_t)reg->left_len);
    b->jr_unit1.assign(reg->unit1, (size_t)reg->unit1_len);
    b->jr_mid.assign(reg->mid ? reg->mid : "", (size_t)reg->mid_len);
    b->jr_unit2.assign(reg->unit2, (size_t)reg->unit2_len);
    b->jr_right.assign(reg->right ? reg->right : "", (size_t)reg->right_len);

    PackedReads pr;
    rc = pack_reads(n_reads, seqs, seq_off, nullptr, 1, pr, NRA_MAX_QLEN);
    if (rc) return rc;
    // Reads longer than one register block (3072 bases), or whose scores do not fit the int32 cells of the
    // joint sweeps, are scored cell by cell in chained row blocks with int64 cells (rare: a long amplicon).
    b->chained_reads.assign((size_t)n_reads, 0);
    for (int32_t r = 0; r < n_reads; ++r)
        b->chained_reads[r] = pr.reads[r].qlen > NRA_MAX_QLEN_1BLOCK || max_score(sc, pr.reads[r].qlen) > kScoreCapPk16;
    b->lst_strand.assign((size_t)n_reads, 0);
    uint64_t jpack_ints = 0;
    {
        // rows-per-lane bucket of every read (measured on config 3, 5000 reads, R = 13..28: folding buckets of
        // fewer than 2048 reads into the next within 4 rows beats (1024, 2) and wider spans), and its partner
        std::vector<std::vector<int32_t>> by_bucket((size_t)kNumR);
        for (int32_t r = 0; r < n_reads; ++r)
            if (pr.reads[r].qlen > 0 && !b->chained_reads[r]) by_bucket[rows_for_qlen(pr.reads[r].qlen)].push_back(r);
        fold_small_buckets(by_bucket, 2048, 4);
        b->jbucket.assign((size_t)n_reads, -1);
        b->jpair_of.assign((size_t)n_reads, -1);
        uint64_t ints = 0;
        for (int bi = 0; bi < kNumR; ++bi) {
            const auto& v = by_bucket[bi];
            for (size_t i = 0; i < v.size(); i += 2) {
                NraJointPairTask t{};
                t.read_a = v[i]; t.read_b = i + 1 < v.size() ? v[i + 1] : -1;
                t.state = ints;
                ints += (uint64_t)NRA_JOINT_NPSTATE(kRList[bi]) * 64;
                b->jpair_of[t.read_a] = (int32_t)b->jpairs.size();
                if (t.read_b >= 0) b->jpair_of[t.read_b] = (int32_t)b->jpairs.size();
                b->jpairs.push_back(t);
            }
            for (int32_t r : v) b->jbucket[r] = bi;
        }
        // the packed cells hold what the 1D sweeps hold (nra_pk16.h), without the doubling of the origin bit
        const int o1 = sc->gap_open1 + sc->gap_ext1, o2 = sc->gap_open2 + sc->gap_ext2;
        const bool fits = o1 >= sc->mismatch && o1 >= sc->sc_ambi && sc->match + o1 <= 127 && sc->gap_ext1 <= 256 &&
                          sc->gap_ext2 <= 256 && o2 + sc->mismatch + sc->sc_ambi <= 700 &&
                          (flags & (NRA_F_BRUTE_FORCE | NRA_F_NO_JOINT_PACK)) == 0 && ints * 4 <= (4ull << 30);
        auto cols_ok = [&](int64_t flank) {
            const int64_t c = NRA_JOINT_PACKED_COLS(flank);
            return c >= NRA_JOINT_PACKED_MIN_COLS && max_score(sc, std::min<int64_t>(c, NRA_MAX_QLEN_1BLOCK)) <= 27000;
        };
        b->jpack_l = fits && ints > 0 && cols_ok(reg->left_len);
        b->jpack_r = fits && ints > 0 && cols_ok(reg->right_len);
        jpack_ints = ints;
    }
    b->arena.expect(pr.q2bit.size() * 6 + (size_t)n_reads * 96 + (size_t)reg->left_len + (size_t)reg->right_len + (1u << 20) +
                    ((b->jpack_l ? jpack_ints : 0) + (b->jpack_r ? jpack_ints : 0)) * 4);
    HIP_TRY(b->q2bit.upload(pr.q2bit));
    HIP_TRY(b->qnmask.upload(pr.nmask));
    b->n_q2bit_words = pr.q2bit.size();
    b->reads_have_n = pr.has_n;
    b->host_reads = std::move(pr.reads);
    b->rev_strand.assign((size_t)n_reads, 0);
    b->rpk_strand.assign((size_t)n_reads, 0);
    if (b->jpack_l) HIP_TRY(b->jlstate.alloc((size_t)jpack_ints));
    if (b->jpack_r) HIP_TRY(b->jrstate.alloc((size_t)jpack_ints));
    if ((b->jpack_l || b->jpack_r) && reg->left_len >= 1 && reg->right_len >= 2 && n_reads > 0) {
        // what flank sweeps ahead of a cell list read (nra_batch2d_sweep_flanks): L and rev(R) alone, the oriented reads,
        // the pair tasks -- buffers of the batch, not of a cell list
        std::vector<uint8_t> pool;
        bool has_n = b->reads_have_n;
        NraDevRegion d{};
        d.p1_off = pool_append(pool, b->jr_left.data(), reg->left_len, nullptr, 0, 0, has_n);
        std::string rr(b->jr_right);
        std::reverse(rr.begin(), rr.end());
        d.pr_off = pool_append(pool, rr.data(), reg->right_len, nullptr, 0, 0, has_n);
        d.p2_off = d.p3_off = (uint32_t)pool.size();
        d.l1 = reg->left_len; d.m1 = reg->unit1_len; d.l2 = reg->mid_len; d.m2 = reg->unit2_len; d.l3 = reg->right_len;
        pool.push_back(0);
        b->warm_has_n = has_n ? 1 : 0;
        HIP_TRY(b->warm_pool.upload(pool));
        HIP_TRY(b->warm_region.upload(std::vector<NraDevRegion>(1, d)));
        HIP_TRY(b->warm_reads.alloc((size_t)n_reads));
        HIP_TRY(b->warm_tasks.alloc(2 * b->jpairs.size()));
        b->warm_built = true;
    }
    HIP_TRY(b->jsnap.alloc(b->n_q2bit_words * 16 * 3));    // R side of the junction per read base: kept across cell lists
    HIP_TRY(b->jread_a.alloc((size_t)n_reads));
    rc = alloc_results(b, (size_t)n_reads, true);      // best_wscore, n_ties, sum_k, sum_k2, status, strand_out
    if (rc) return rc;
    for (int i = 0; i < 2; ++i) HIP_TRY(g_handles.event(b->device, true, &b->phase_ev[i]));
    HIP_TRY(g_handles.event(b->device, false, &b->fork_ev));
    HIP_TRY(g_handles.event(b->device, false, &b->fork2_ev));
    *out = guard.release();
    return NRA_OK;
}

}  // extern "C"

namespace {

using GridRow = NraGridRow;
// rows: the cells of every read.  keep: the repeat counts of every read the sweeps leave column states at, when these
// are kept for a later, finer grid (same layout: k1lo, n1 / k2lo, n2 at step 1; a superset of the read's cells).
struct JointGrid { int32_t step1, step2; const GridRow* rows; const GridRow* keep; };

// first index i in [0, count] with start + i * step >= x  (numpy.searchsorted(grid, x, side="left") on the grid values)
int32_t grid_lower_bound(int32_t start, int32_t step, int32_t count, double x)
{
    if (count <= 0) return 0;
    double f = std::ceil((x - (double)start) / (double)step);
    int64_t i = f < 0 ? 0 : (f > (double)count ? count : (int64_t)f);
    while (i > 0 && (double)start + (double)(i - 1) * step >= x) --i;           // rounding of the quotient
    while (i < count && (double)start + (double)i * step < x) ++i;
    return (int32_t)i;
}

// The reference's routing of reads to the cells of one grid round (nanoRepeat_joint.py:397-409 round 2, :315-330
// round 3): read r takes the grid values g of axis a with lo_a[r] <= g < hi_a[r]; none on one axis = no cells.
int route_grid(int32_t n_reads, int32_t start1, int32_t step1, int32_t count1, const double* lo1, const double* hi1,
               int32_t start2, int32_t step2, int32_t count2, const double* lo2, const double* hi2,
               std::vector<GridRow>& rows, int64_t& n_cells, std::vector<GridRow>* keep = nullptr)
{
    if (n_reads < 0 || step1 <= 0 || step2 <= 0 || count1 < 0 || count2 < 0 || start1 < 0 || start2 < 0)
        return fail(NRA_E_ARG, "bad grid (steps > 0, starts and counts >= 0)");
    if (n_reads > 0 && (!lo1 || !hi1 || !lo2 || !hi2)) return fail(NRA_E_ARG, "NULL grid bound array");
    if ((int64_t)start1 + (int64_t)step1 * count1 > 0x3fffffff || (int64_t)start2 + (int64_t)step2 * count2 > 0x3fffffff)
        return fail(NRA_E_RANGE, "grid values too large");
    rows.assign((size_t)n_reads, GridRow{0, 0, 0, 0});
    n_cells = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        if (!(lo1[r] < hi1[r]) || !(lo2[r] < hi2[r])) continue;         // (also drops NaN bounds)
        const int32_t a1 = grid_lower_bound(start1, step1, count1, lo1[r]), b1 = grid_lower_bound(start1, step1, count1, hi1[r]);
        const int32_t a2 = grid_lower_bound(start2, step2, count2, lo2[r]), b2 = grid_lower_bound(start2, step2, count2, hi2[r]);
        if (b1 <= a1 || b2 <= a2) continue;
        rows[(size_t)r] = GridRow{start1 + a1 * step1, b1 - a1, start2 + a2 * step2, b2 - a2};
        n_cells += (int64_t)(b1 - a1) * (b2 - a2);
    }
    if (keep) {
        // What a finer grid inside the same bounds can ask of a read: every count k with lo <= k < hi -- and, the way the
        // reference refines (within one coarse step of a size between the coarse grid's first and last value), nothing
        // more than one step below the first or above the last of the read's values.  A unit-step axis keeps its own.
        keep->assign((size_t)n_reads, GridRow{0, 0, 0, 0});
        auto span = [](int32_t first, int32_t n, int32_t step, double lo, double hi, int32_t& a, int32_t& count) {
            const int32_t last = first + (n - 1) * step;
            int64_t ka = first, kb = last;
            if (step > 1) {
                const double cl = std::ceil(lo), ch = std::ceil(hi);
                const int64_t klo = cl < 0 ? 0 : (cl > 1e9 ? (int64_t)1e9 : (int64_t)cl);
                const int64_t khi = ch > 1e9 ? (int64_t)1e9 : (int64_t)ch - 1;
                ka = std::max<int64_t>(std::max<int64_t>(klo, (int64_t)first - step), 0);
                kb = std::min<int64_t>(khi, (int64_t)last + step - 1);
                ka = std::min<int64_t>(ka, first); kb = std::max<int64_t>(kb, last);
            }
            a = (int32_t)ka; count = (int32_t)(kb - ka + 1);
        };
        for (int32_t r = 0; r < n_reads; ++r) {
            const GridRow& g = rows[(size_t)r];
            if (g.n1 <= 0 || g.n2 <= 0) continue;
            GridRow& k = (*keep)[(size_t)r];
            span(g.k1lo, g.n1, step1, lo1[r], hi1[r], k.k1lo, k.n1);
            span(g.k2lo, g.n2, step2, lo2[r], hi2[r], k.k2lo, k.n2);
        }
    }
    return NRA_OK;
}

int set_cells_common(nra_batch* b, const int8_t* read_strand, int64_t n_cells, const int32_t* cell_read,
                     const int32_t* cell_k1, const int32_t* cell_k2, const JointGrid* grid);
int run_2d_flanks(nra_batch* b);

}  // namespace

extern "C" {

int64_t nra_joint_grid_cells(int32_t n_reads, int32_t start1, int32_t step1, int32_t count1, const double* lo1,
                             const double* hi1, int32_t start2, int32_t step2, int32_t count2, const double* lo2,
                             const double* hi2, int64_t cap, int32_t* cell_read, int32_t* cell_k1, int32_t* cell_k2)
{
    std::vector<GridRow> rows;
    int64_t n = 0;
    const int rc = route_grid(n_reads, start1, step1, count1, lo1, hi1, start2, step2, count2, lo2, hi2, rows, n);
    if (rc) return rc;
    if (!cell_read && !cell_k1 && !cell_k2) return n;
    if (!cell_read || !cell_k1 || !cell_k2) return fail(NRA_E_ARG, "all three cell arrays or none");
    if (cap < n) return fail(NRA_E_ARG, "cell arrays too small: " + std::to_string(n) + " cells");
    int64_t c = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        const GridRow& g = rows[(size_t)r];
        for (int32_t i = 0; i < g.n1; ++i)
            for (int32_t j = 0; j < g.n2; ++j, ++c) {
                cell_read[c] = r; cell_k1[c] = g.k1lo + i * step1; cell_k2[c] = g.k2lo + j * step2;
            }
    }
    return n;
}

// The cell list of one grid round.  May be called again on the same batch (the reads stay packed on the
// device; the buffers of the previous list are handed out again).
int nra_batch2d_set_cells(nra_batch_t* b, const int8_t* read_strand, int64_t n_cells, const int32_t* cell_read,
                          const int32_t* cell_k1, const int32_t* cell_k2)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    if (n_cells < 0) return fail(NRA_E_ARG, "bad cell count");
    if (n_cells > 0 && (!cell_read || !cell_k1 || !cell_k2)) return fail(NRA_E_ARG, "NULL cell array");
    return set_cells_common(b, read_strand, n_cells, cell_read, cell_k1, cell_k2, nullptr);
}

// The same for a whole routed grid: the library lists the cells itself (no per-cell arrays cross the boundary).
int nra_batch2d_set_grid(nra_batch_t* b, const int8_t* read_strand,
                         int32_t start1, int32_t step1, int32_t count1, const double* lo1, const double* hi1,
                         int32_t start2, int32_t step2, int32_t count2, const double* lo2, const double* hi2,
                         int64_t* n_cells_out)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    std::vector<GridRow> rows, keep;
    int64_t n = 0;
    int rc = route_grid(b->n_reads, start1, step1, count1, lo1, hi1, start2, step2, count2, lo2, hi2, rows, n, &keep);
    if (rc) return rc;
    if (n_cells_out) *n_cells_out = n;
    const JointGrid grid{step1, step2, rows.data(), keep.data()};
    // no per-cell arrays at all: the tasks come from the rows, the selector gets the rows (16 bytes a read)
    rc = set_cells_common(b, read_strand, n, nullptr, nullptr, nullptr, &grid);
    if (rc == NRA_E_NOMEM && b->joint_keep == 1) {
        // The kept column states did not fit after all (the check against the free memory races with other batches, ranks
        // and libraries on the device): give the arena back and set the grid again without keeping -- a later, finer grid
        // then sweeps again, nra_batch2d_refine says NRA_E_STATE, the results are the same.
        b->keep_arena.release();
        b->keep_valid = false;
        b->keep_off = true;
        rc = set_cells_common(b, read_strand, n, nullptr, nullptr, nullptr, &grid);
        b->keep_off = false;
    }
    return rc;
}

}  // extern "C"

namespace {

// A read's MID scans as several tasks of at most kMidscanChunk k1 values: a wave's scans are one dependent chain (load a
// column state, 1 + |mid| columns of two lane scans each, store), so more, shorter waves hide it better than one wave per
// read (config 3, round 3: 5 k1 values per read).  The pieces differ in where their lists, outputs and slots begin only.
void push_midscan(std::vector<NraJointTask>& jtail, const NraJointTask& whole, uint64_t ints_per_k1)
{
    const int kMidscanChunk = 3;            // config 3, device ms per step: whole reads 6.23, 1: 6.23, 2: 6.18, 3: 6.15, 4: 6.16, 6: 6.40
    for (int32_t c = 0; c < whole.nk1; c += kMidscanChunk) {
        NraJointTask t = whole;
        t.k1_off = whole.k1_off + c; t.nk1 = std::min<int32_t>(kMidscanChunk, whole.nk1 - c);
        t.out = whole.out + c; t.pstate = whole.pstate + (uint64_t)c * ints_per_k1;
        jtail.push_back(t);
    }
}

int set_cells_common(nra_batch* b, const int8_t* read_strand, int64_t n_cells, const int32_t* cell_read,
                     const int32_t* cell_k1, const int32_t* cell_k2, const JointGrid* grid)
{
    if (n_cells > 0x7ff00000ll) return fail(NRA_E_RANGE, "too many cells");
    HIP_TRY(hipSetDevice(b->device));
    if (b->ran) HIP_TRY(hipStreamSynchronize(b->stream));      // the previous list's kernels own the buffers
    if (b->flanks_enqueued) {                                  // ... also when only its first part was enqueued
        for (hipStream_t q : b->bstreams) HIP_TRY(hipStreamSynchronize(q));
        HIP_TRY(hipStreamSynchronize(b->stream));
        b->flanks_enqueued = false;
    }
    {
        int rc0 = account_run_fwd(b);
        if (rc0) return rc0;
    }
    b->have_cells = false;
    b->ran = false;
    const bool had_warm = b->warm_pending;                     // flank sweeps enqueued ahead of this list (nra_batch2d_sweep_flanks)
    b->rev_pending.clear();
    b->lst_pending.clear();
    b->rpk_pending.clear();
    b->cell_arena.reset();
    ArenaScope arena_scope(&b->cell_arena);
    b->buckets.clear();
    b->jgroups.clear();
    b->n_cands = n_cells;
    const int32_t n_reads = b->n_reads;
    const int32_t flags = b->flags;
    std::vector<NraDevRead> reads(b->host_reads);              // + the shadows of chained reads, below
    PhaseClock clk;

    std::vector<uint32_t> first((size_t)n_reads, 0), cnt((size_t)n_reads, 0);
    int32_t k1max = 0, k2max = 0;
    if (grid) {
        uint32_t c = 0;
        for (int32_t r = 0; r < n_reads; ++r) {
            const GridRow& g = grid->rows[(size_t)r];
            first[r] = c; cnt[r] = (uint32_t)g.n1 * (uint32_t)g.n2;
            c += cnt[r];
            if (cnt[r]) {
                k1max = std::max(k1max, g.k1lo + (g.n1 - 1) * grid->step1);
                k2max = std::max(k2max, g.k2lo + (g.n2 - 1) * grid->step2);
            }
        }
    } else
    for (int64_t c = 0; c < n_cells; ++c) {
        const int32_t r = cell_read[c];
        if (r < 0 || r >= n_reads || (c > 0 && r < cell_read[c - 1]) || cell_k1[c] < 0 || cell_k2[c] < 0)
            return fail(NRA_E_ARG, "cells must be grouped by read, k >= 0");
        if (cnt[r] == 0) first[r] = (uint32_t)c;
        cnt[r]++;
        k1max = std::max(k1max, cell_k1[c]);
        k2max = std::max(k2max, cell_k2[c]);
    }
    // k1 / k2 of a read's i-th listed cell
    auto k1_of = [&](int32_t r, uint32_t i) -> int32_t {
        if (!grid) return cell_k1[first[r] + i];
        const GridRow& g = grid->rows[(size_t)r];
        return g.k1lo + (int32_t)(i / (uint32_t)g.n2) * grid->step1;
    };
    auto k2_of = [&](int32_t r, uint32_t i) -> int32_t {
        if (!grid) return cell_k2[first[r] + i];
        const GridRow& g = grid->rows[(size_t)r];
        return g.k2lo + (int32_t)(i % (uint32_t)g.n2) * grid->step2;
    };
    const int32_t left_len = (int32_t)b->jr_left.size(), unit1_len = (int32_t)b->jr_unit1.size(),
                  mid_len = (int32_t)b->jr_mid.size(), unit2_len = (int32_t)b->jr_unit2.size(),
                  right_len = (int32_t)b->jr_right.size();
    const int64_t win = (int64_t)unit1_len * k1max + mid_len + (int64_t)unit2_len * k2max + 20;
    if (6 * win >= 32768) return fail(NRA_E_RANGE, "repeat window too long for the 16-bit window score");
    const int64_t tlmax = (int64_t)left_len + win - 20 + right_len;
    if (tlmax > NRA_MAX_TLEN) return fail(NRA_E_RANGE, "template too long");

    // ---- Kept column states (routed grids with every strand given, MID scans).  A grid's sweeps leave a column state at
    // every repeat count a finer grid inside the same bounds can ask for (grid->keep), not only at its own values: such a
    // later grid -- the reference's round 3 after round 2 -- then needs no flank, prefix or reverse sweep at all, only
    // the MID scans and the combine.  joint_keep: 0 = nothing kept, 1 = this list sweeps and keeps, 2 = no sweeps.
    int32_t k1max_pool = k1max, k2max_pool = k2max;
    int keep = 0;
    uint64_t keep_bytes = 0;
    {
        const bool brute_now = (flags & NRA_F_BRUTE_FORCE) != 0 || left_len < 1 || right_len < 2;
        bool strands_all = read_strand != nullptr && n_reads > 0;
        if (read_strand)
            for (int32_t r = 0; r < n_reads; ++r) if (cnt[r] > 0 && read_strand[r] == 0) strands_all = false;
        auto swept = [&](int32_t r) { return cnt[r] > 0 && reads[r].qlen > 0 && !b->chained_reads[r]; };
        if (grid && grid->keep && !brute_now && strands_all && n_cells > 0 &&
            (flags & (NRA_F_JOINT_TAILS | NRA_F_JOINT_NO_CHAIN | NRA_F_JOINT_NO_KEEP)) == 0) {
            bool reuse = b->keep_valid && b->keep_rows.size() == (size_t)n_reads;
            for (int32_t r = 0; reuse && r < n_reads; ++r) {
                if (!swept(r)) continue;
                const GridRow& g = grid->rows[(size_t)r];
                const NraGridRow& k = b->keep_rows[(size_t)r];
                if (b->keep_strand[(size_t)r] != read_strand[r] || k.n1 <= 0 || k.n2 <= 0 || g.k1lo < k.k1lo ||
                    g.k1lo + (g.n1 - 1) * grid->step1 > k.k1lo + k.n1 - 1 || g.k2lo < k.k2lo ||
                    g.k2lo + (g.n2 - 1) * grid->step2 > k.k2lo + k.n2 - 1)
                    reuse = false;
            }
            if (reuse) {
                keep = 2;
                // a refinement may follow (nra_batch2d_refine): its MID scans read the template up to the last kept count
                for (int32_t r = 0; r < n_reads; ++r)
                    if (swept(r)) {
                        const NraGridRow& k = b->keep_rows[(size_t)r];
                        k1max_pool = std::max(k1max_pool, k.k1lo + k.n1 - 1); k2max_pool = std::max(k2max_pool, k.k2lo + k.n2 - 1);
                    }
            } else {
                uint64_t bytes = 0;
                int32_t k1hi = k1max, k2hi = k2max;
                for (int32_t r = 0; r < n_reads; ++r) {
                    if (!swept(r)) continue;
                    const GridRow& k = grid->keep[(size_t)r];
                    bytes += ((uint64_t)k.n1 * NRA_JOINT_NSTATE(kRList[b->jbucket[r]]) + (uint64_t)k.n2 * 3 * kRList[b->jbucket[r]]) * 256;
                    k1hi = std::max(k1hi, k.k1lo + k.n1 - 1); k2hi = std::max(k2hi, k.k2lo + k.n2 - 1);
                }
                const int64_t tl_keep = (int64_t)left_len + (int64_t)unit1_len * k1hi + mid_len + (int64_t)unit2_len * k2hi + right_len;
                // ... within the budget, and within half of what the device has left (what the arena holds already counts as free)
                // (NRA_JOINT_KEEP_BUDGET_GB in the environment overrides the built-in budget; 0 keeps nothing)
                const char* budget_env = getenv("NRA_JOINT_KEEP_BUDGET_GB");
                const uint64_t budget = budget_env ? (uint64_t)(atof(budget_env) * (double)(1ull << 30)) : (uint64_t)NRA_JOINT_KEEP_BUDGET;
                bool fits = bytes <= budget && tl_keep <= NRA_MAX_TLEN && !b->keep_off;
                keep_bytes = bytes;
                if (fits) {
                    size_t held = 0, free_b = 0, total_b = 0;
                    for (const Arena::Chunk& c : b->keep_arena.chunks) held += c.size;
                    if (bytes > held) fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes - held <= free_b / 2;
                }
                if (fits) { keep = 1; k1max_pool = k1hi; k2max_pool = k2hi; }
            }
        }
        if (keep != 2) b->keep_valid = false;       // what was kept is replaced (1) or not looked at again (0)
        b->keep_pending = keep == 1;
        b->joint_keep = keep;
        if (keep == 1) {
            // what the arena holds is handed out again; if it is too small it goes back BEFORE the larger chunk is asked for
            // (old and new together may not fit)
            if (g_debug_phases)
                fprintf(stderr, "[nra] kept column states: %.3f GB wanted, the arena holds %.3f GB in %zu chunks -> %s\n", (double)keep_bytes / 1073741824.0,
                        (double)b->keep_arena.capacity() / 1073741824.0, b->keep_arena.chunks.size(),
                        b->keep_arena.capacity() < keep_bytes + (2u << 20) ? "given back, asked for again" : "handed out again");
            if (b->keep_arena.capacity() < keep_bytes + (2u << 20)) b->keep_arena.release();
            else b->keep_arena.reset();
            b->keep_rows.assign(grid->keep, grid->keep + n_reads);
            b->keep_strand.assign(read_strand, read_strand + n_reads);
            b->keep_state_off.assign((size_t)n_reads, 0);
            b->keep_rs_off.assign((size_t)n_reads, 0);
            b->keep_ra_off.assign((size_t)n_reads, 0);
        }
    }

    clk.mark("2D cells: first/count");
    std::vector<uint8_t> pool;
    bool has_n = b->reads_have_n;
    NraDevRegion d{};
    d.p1_off = pool_append(pool, b->jr_left.data(), left_len, b->jr_unit1.data(), unit1_len, k1max_pool, has_n);
    d.p2_off = pool_append(pool, b->jr_mid.data(), mid_len, b->jr_unit2.data(), unit2_len, k2max_pool, has_n);
    d.p3_off = pool_append(pool, b->jr_right.data(), right_len, nullptr, 0, 0, has_n);
    {   // rev(R) + rev(u2)^k2max: the reverse sweeps (the extended ones run on into the second repeat)
        std::string rr(b->jr_right), ru(b->jr_unit2);
        std::reverse(rr.begin(), rr.end());
        std::reverse(ru.begin(), ru.end());
        d.pr_off = pool_append(pool, rr.data(), right_len, ru.data(), unit2_len, k2max_pool, has_n);
    }
    d.l1 = left_len; d.m1 = unit1_len; d.l2 = mid_len; d.m2 = unit2_len; d.l3 = right_len;
    pool.push_back(0);
    b->has_n = has_n ? 1 : 0;
    std::vector<NraDevRegion> dregs(1, d);

    // (the rows-per-lane bucket of a read was fixed when the reads were packed)
    // (measured and dropped: a bucket as 2 - 4 sub-buckets with kernel chains of their own -- the short kernels at the
    // end of one chain beside the long ones of the next -- is slower, 7.1 -> 7.7 / 8.7 / 10.5 ms of device time on config 3)
    const int kChainBucket = kNumR;
    std::vector<std::vector<int32_t>> by_bucket((size_t)kChainBucket + 1);
    bool empty_reads_listed = false;
    for (int32_t r = 0; r < n_reads; ++r) {
        if (cnt[r] > 0 && reads[r].qlen == 0) empty_reads_listed = true;
        if (cnt[r] == 0 || reads[r].qlen == 0) continue;
        by_bucket[b->chained_reads[r] ? kChainBucket : b->jbucket[r]].push_back(r);
    }
    const std::vector<int32_t>& chain_reads = by_bucket[kChainBucket];
    std::vector<NraPairTask> pair_tasks;
    std::vector<NraTask> queue_tasks, probe_tasks;
    std::vector<int32_t> queue_count, probe_count;
    std::vector<NraJointTask> jbwd, jpre, jtail;
    std::vector<NraJointPairTask> jlpk, jrpk;
    std::vector<uint8_t> pair_l((size_t)b->jpairs.size(), 0), pair_r((size_t)b->jpairs.size(), 0);
    const int colsL = NRA_JOINT_PACKED_COLS(left_len), colsR = NRA_JOINT_PACKED_COLS(right_len);
    std::vector<int32_t> k1list;
    // junction decomposition needs a base left of the window and two bases of R
    b->brute = (flags & NRA_F_BRUTE_FORCE) != 0 || left_len < 1 || right_len < 2;
    // a routed grid: every read's cells are a full product (k1 values) x (one k2 progression) -> junction at the end
    // of mid.  The state kept from earlier cell lists differs between the two forms: switching drops it.
    b->joint_v2 = grid != nullptr && !b->brute && (flags & NRA_F_JOINT_TAILS) == 0;
    b->joint_chain = b->joint_v2 && (flags & NRA_F_JOINT_NO_CHAIN) == 0;
    // k_joint_combine writes every cell of its reads, found or not: the cell arrays need no clearing unless some
    // cells belong to no combine task (empty reads, reads scored cell by cell)
    b->cells_need_clear = !b->joint_v2 || empty_reads_listed || !chain_reads.empty();
    if (b->joint_v2 != b->joint_v2_prev) {
        std::fill(b->rev_strand.begin(), b->rev_strand.end(), (int8_t)0);
        b->joint_v2_prev = b->joint_v2;
    }
    std::vector<NraJointCombineTask> jcomb;
    std::vector<uint64_t> rs_off((size_t)n_reads, 0);
    std::vector<int32_t> ra_off((size_t)n_reads, 0);
    uint64_t rs_total = 0, fs_total = 0;
    int64_t ra_total = 0, fb_total = 0;
    // wave states of one group of reads (NRA_F_TEST_CHAIN: one read per group, to exercise the reuse)
    // The buckets run concurrently, each in its own part of the state buffer.
    size_t n_nonempty = 0;
    for (const auto& v : by_bucket) n_nonempty += v.empty() ? 0 : 1;
    const uint64_t state_cap = (flags & NRA_F_TEST_CHAIN) ? 1 : (uint64_t)NRA_JOINT_STATE_CAP_INTS / std::max<size_t>(n_nonempty, 1);
    uint64_t state_base = 0;
    b->all_strands_given = read_strand != nullptr && n_reads > 0;
    if (read_strand)
        for (int32_t r = 0; r < n_reads; ++r) if (cnt[r] > 0 && read_strand[r] == 0) b->all_strands_given = false;
    int64_t alg_cells = 0;
    std::vector<int32_t> ks;
    std::vector<int> bucket_ids;                     // by_bucket index of every entry of b->buckets
    if (!b->brute) {
        jbwd.reserve((size_t)n_reads); jpre.reserve((size_t)n_reads); jcomb.reserve(b->joint_v2 ? (size_t)n_reads : 0);
        jtail.reserve((size_t)n_reads * 2);
    }

    // ---- part 1: what depends on the reads and their strands only -- the reverse sweeps over R and the packed
    // sweeps of the flanks outside the scoring window -- plus the strand probes.  With every strand given these
    // kernels are enqueued right here, before the (longer) list of prefix and tail sweeps is built and uploaded:
    // the host's share of a round then runs beside the device's first third of it.
    for (int bi = kChainBucket; bi >= 0; --bi) {
        if (by_bucket[bi].empty()) continue;
        Bucket bk;
        bk.chain = bi == kChainBucket;
        bk.R = bk.chain ? NRA_CHAIN_R : kRList[bi];
        bk.pair_off = pair_tasks.size();
        bk.jbwd_off = jbwd.size();
        bk.jlpk_off = jlpk.size(); bk.jrpk_off = jrpk.size();
        bk.probe_off = probe_tasks.size();
        const bool per_cell = b->brute || bk.chain;      // one DP per (read, cell) instead of the joint sweeps
        for (int32_t r : by_bucket[bi]) {
            if (!per_cell) {
                // one reverse sweep over R per read and strand, kept for later cell lists of the batch
                const int8_t given = read_strand ? read_strand[r] : 0;
                const int32_t pi = b->jpair_of[r];
                const NraJointPairTask& pair = b->jpairs[(size_t)pi];
                const bool stale = given == 0 || b->rev_strand[r] != given;     // nothing kept for this read and strand
                // the packed sweep of rev(R) up to the window: once per pair and strand, kept for later cell lists
                auto packed_r = [&]() {
                    if (pair_r[pi]) return;
                    bool need = false;
                    for (int32_t q : {pair.read_a, pair.read_b})
                        if (q >= 0 && (read_strand == nullptr || read_strand[q] == 0 || b->rpk_strand[q] != read_strand[q])) need = true;
                    if (!need) return;
                    pair_r[pi] = 1; jrpk.push_back(pair); bk.cells_sweep += (int64_t)2 * 64 * bk.R * colsR;
                    for (int32_t q : {pair.read_a, pair.read_b}) {
                        if (q < 0) continue;
                        b->rpk_strand[q] = 0;
                        b->rpk_pending.push_back({q, read_strand ? read_strand[q] : (int8_t)0});
                    }
                };
                if (keep == 2) {
                    // every column state this list needs was kept: no sweep on either side
                    rs_off[(size_t)r] = b->keep_rs_off[(size_t)r]; ra_off[(size_t)r] = b->keep_ra_off[(size_t)r];
                } else if (b->joint_v2) {
                    // the reverse sweep runs on over rev(u2)^k2hi and leaves a column state per k2 of the read (of its
                    // kept range, keep == 1); the packed sweep of rev(R) up to the window is kept in any case
                    const GridRow& gr = grid->rows[(size_t)r];
                    const int32_t k2lo = keep == 1 ? grid->keep[(size_t)r].k2lo : gr.k2lo, k2n = keep == 1 ? grid->keep[(size_t)r].n2 : gr.n2,
                                  k2step = keep == 1 ? 1 : grid->step2;
                    NraJointTask tb{}; tb.read = r; tb.k2lo = k2lo; tb.k2step = k2step; tb.n2 = k2n;
                    tb.state = rs_total; tb.out = (int32_t)ra_total;
                    rs_off[(size_t)r] = rs_total; ra_off[(size_t)r] = (int32_t)ra_total;
                    if (keep == 1) { b->keep_rs_off[(size_t)r] = rs_total; b->keep_ra_off[(size_t)r] = (int32_t)ra_total; }
                    rs_total += (uint64_t)k2n * 3 * 64 * (uint64_t)bk.R; ra_total += k2n;
                    if (b->jpack_r) {
                        tb.resume = 1; tb.pstate = pair.state; tb.phalf = pair.read_b == r ? 1 : 0;
                        packed_r();
                    }
                    jbwd.push_back(tb);
                    bk.cells_sweep += joint_cells(bk.R, d.l3 - (b->jpack_r ? colsR : 0) + d.m2 * (k2lo + k2step * (k2n - 1)), reads[r].qlen);
                    if (stale) { b->rev_strand[r] = 0; b->rev_pending.push_back({r, given}); }
                } else if (stale) {
                    NraJointTask tb{}; tb.read = r; tb.k2step = 1; tb.n2 = 1;
                    if (b->jpack_r) {
                        tb.resume = 1; tb.pstate = pair.state; tb.phalf = pair.read_b == r ? 1 : 0;
                        packed_r();
                    }
                    jbwd.push_back(tb);
                    bk.cells_sweep += joint_cells(bk.R, d.l3 - (b->jpack_r ? colsR : 0), reads[r].qlen);
                    b->rev_strand[r] = 0;
                    b->rev_pending.push_back({r, given});
                }
                // the L side up to the window: swept once per pair and strand, kept for later cell lists
                if (keep != 2 && b->jpack_l && (given == 0 || b->lst_strand[r] != given) && !pair_l[pi]) {
                    pair_l[pi] = 1; jlpk.push_back(pair); bk.cells_sweep += (int64_t)2 * 64 * bk.R * colsL;
                    for (int32_t q : {pair.read_a, pair.read_b}) {
                        if (q < 0) continue;
                        b->lst_strand[q] = 0;
                        b->lst_pending.push_back({q, read_strand ? read_strand[q] : (int8_t)0});
                    }
                }
            }
            // strand probe against the read's first listed cell: half A = template, half B = its revcomp
            // (no probe runs when every strand is given)
            if (!bk.chain) {
                if (!b->all_strands_given) {
                    NraPairTask t{};
                    t.read = r; t.k1a = t.k1b = k1_of(r, 0); t.k2a = t.k2b = k2_of(r, 0);
                    t.out_a = 2 * r; t.out_b = 2 * r + 1; t.flags = 3;     // B = reverse complement; raw scores
                    pair_tasks.push_back(t);
                    bk.cells_pair += 2 * sweep_cells(bk.R, d.l1 + d.m1 * t.k1a + d.l2 + d.m2 * t.k2a + d.l3);
                }
            } else {
                // chained: the read and a reverse-complemented shadow of it against the same template
                NraDevRead shadow = reads[r];
                shadow.rc = 1;
                const int32_t sh = (int32_t)reads.size();
                reads.push_back(shadow);
                probe_tasks.push_back(NraTask{r, k1_of(r, 0), k2_of(r, 0), 2 * r});
                probe_tasks.push_back(NraTask{sh, k1_of(r, 0), k2_of(r, 0), 2 * r + 1});
            }
        }
        bk.n_pair = (int)(pair_tasks.size() - bk.pair_off);
        bk.n_jbwd = (int)(jbwd.size() - bk.jbwd_off);
        bk.n_jlpk = (int)(jlpk.size() - bk.jlpk_off);
        bk.n_jrpk = (int)(jrpk.size() - bk.jrpk_off);
        bk.n_probe = (int)(probe_tasks.size() - bk.probe_off);
        probe_count.push_back(bk.n_probe);
        b->buckets.push_back(bk);
        bucket_ids.push_back(bi);
    }
    const size_t nb = b->buckets.size();
    clk.mark("2D cells: pool, strand-only tasks built");
    // one chunk for everything but the wave states, which get their own (all of it reused by the next cell list)
    b->cell_arena.expect(b->n_q2bit_words * 16 * 13 + (size_t)n_cells * 72 + pool.size() + (size_t)n_reads * 256 + (8u << 20));
    HIP_TRY(b->pool.upload(pool));
    HIP_TRY(b->regions.upload(dregs));
    HIP_TRY(b->reads.upload(reads));
    HIP_TRY(b->reads_init.upload(reads));
    HIP_TRY(b->pair_tasks.upload(pair_tasks));
    HIP_TRY(b->probe_tasks.upload(probe_tasks));
    HIP_TRY(b->probe_count.upload(probe_count));
    HIP_TRY(b->probe_dummy.alloc(2 * (size_t)n_reads));
    HIP_TRY(b->probe_score.alloc(2 * (size_t)n_reads));
    b->have_strand_in = read_strand != nullptr;
    if (read_strand) {
        std::vector<int8_t> v(read_strand, read_strand + n_reads);
        HIP_TRY(b->strand_in.upload(v));
    }
    if (!chain_reads.empty()) {
        size_t chain_cells = 0;
        for (int32_t r : chain_reads) chain_cells += cnt[r];
        b->chain_cap = (int)((tlmax + 127) / 64 * 64 + 64);
        b->payload_strips = chain_strips(std::max(chain_cells, probe_tasks.size()), NRA_CHAIN_STRIPS,
                                         (size_t)6 * 8 * (size_t)b->chain_cap);
        HIP_TRY(b->chain_payload.alloc((size_t)b->payload_strips * 6 * (size_t)b->chain_cap));
    }
    if (!b->brute) {
        HIP_TRY(b->jbwd_tasks.upload(jbwd));
        HIP_TRY(b->jlpk_tasks.upload(jlpk));
        HIP_TRY(b->jrpk_tasks.upload(jrpk));
        if (b->joint_v2 && keep != 2) {
            ArenaScope kept(keep == 1 ? &b->keep_arena : &b->cell_arena);
            if (keep == 1) b->keep_arena.expect((size_t)rs_total * 4 + (size_t)ra_total * 4 + (1u << 20));
            HIP_TRY(b->jrs.alloc((size_t)rs_total));
            HIP_TRY(b->jra.alloc((size_t)ra_total));
        }
    }
    // events and streams: kept from one cell list to the next, more taken from the pool when needed
    auto ensure_handles = [&](size_t n_ev) -> int {
        while (b->ev.size() < n_ev) { hipEvent_t e; HIP_TRY(g_handles.event(b->device, true, &e)); b->ev.push_back(e); }
        // two streams per bucket: the buckets' chains overlap, and so do a bucket's reverse and prefix sweeps
        while (b->bstreams.size() < 2 * nb) { hipStream_t q; HIP_TRY(g_handles.stream(b->device, &q)); b->bstreams.push_back(q); }
        while (b->bdone.size() < 3 * nb) { hipEvent_t e; HIP_TRY(g_handles.event(b->device, false, &e)); b->bdone.push_back(e); }
        return NRA_OK;
    };
    {
        const int rc1 = ensure_handles((b->warm_pending ? (size_t)b->ev_next : 2) + 12 * nb + 2);
        if (rc1) return rc1;
    }
    clk.mark("2D cells: strand-only uploads, handles");
    b->flanks_enqueued = false;
    if (b->all_strands_given && !b->brute && n_cells > 0) {
        const int rc1 = run_2d_flanks(b);
        if (rc1) return rc1;
    }
    clk.mark("2D cells: strand-only kernels enqueued");

    // ---- part 2: the sweeps that depend on the cell list
    for (size_t bidx = 0; bidx < nb; ++bidx) {
        const int bi = bucket_ids[bidx];
        Bucket& bk = b->buckets[bidx];
        bk.queue_off = queue_tasks.size();
        bk.comb_off = jcomb.size();
        const bool per_cell = b->brute || bk.chain;
        const uint64_t slot = (uint64_t)NRA_JOINT_NSTATE(bk.R) * 64;
        JointGroup g; g.R = bk.R; g.bucket = (int)bidx; g.pre_off = jpre.size(); g.tail_off = jtail.size();
        uint64_t used = 0, state_max = 0;
        const size_t bucket_pre0 = jpre.size(), bucket_tail0 = jtail.size();
        for (int32_t r : by_bucket[bi]) {
            if (!per_cell) {
                // one prefix sweep over L + u1^k1max per read that leaves the wave state at each of the read's k1
                // values; one tail sweep per run of cells with the same k1 and k2 in arithmetic progression (how
                // the grid rounds list them)
                const NraJointPairTask& pair = b->jpairs[(size_t)b->jpair_of[r]];
                ks.clear();                                      // the read's distinct k1 values, ascending
                if (grid) {
                    const GridRow& gr = grid->rows[(size_t)r];
                    for (int32_t i = 0; i < gr.n1; ++i) ks.push_back(gr.k1lo + i * grid->step1);
                    if (keep != 0) {
                        // kept column states: the prefix sweep (keep == 1; none at all, keep == 2) leaves one at EVERY count
                        // of the read's kept range, the MID scans take the grid's own from among them
                        const NraGridRow kr = keep == 1 ? grid->keep[(size_t)r] : b->keep_rows[(size_t)r];
                        const uint64_t q3 = (uint64_t)3 * (uint64_t)reads[r].qlen;
                        NraJointTask t{}; t.read = r; t.k1_off = (int32_t)k1list.size(); t.nk1 = gr.n1;
                        t.k1 = kr.k1lo; t.k2step = 1;
                        t.out = (int32_t)fb_total; t.pstate = fs_total;
                        k1list.insert(k1list.end(), ks.begin(), ks.end());
                        bk.cells_sweep += (int64_t)64 * bk.R * gr.n1 * (1 + d.l2);
                        if (keep == 1) {
                            NraJointTask tp{}; tp.read = r; tp.k1_off = (int32_t)k1list.size(); tp.nk1 = kr.n1;
                            tp.state = used; tp.k2step = 1;
                            if (b->jpack_l) { tp.resume = 1; tp.pstate = pair.state; tp.phalf = pair.read_b == r ? 1 : 0; }
                            jpre.push_back(tp);
                            for (int32_t i = 0; i < kr.n1; ++i) k1list.push_back(kr.k1lo + i);
                            bk.cells_sweep += (int64_t)64 * bk.R * (d.l1 + d.m1 * (kr.k1lo + kr.n1 - 1) - 1 - (b->jpack_l ? colsL : 0));
                            bk.cells_sweep += (int64_t)64 * bk.R * (1 + std::min(63, std::max(reads[r].qlen - 1, 0) / bk.R));   // its drain
                            t.state = used;                         // (bucket-relative like tp.state: both move to the bucket's base below)
                            used += slot * (uint64_t)kr.n1;
                            state_max = std::max(state_max, used);
                        } else
                            t.state = b->keep_state_off[(size_t)r] | (1ull << 63);      // absolute already (marked; unmarked below)
                        push_midscan(jtail, t, q3);
                        NraJointCombineTask ct{};
                        ct.read = r; ct.n1 = gr.n1; ct.n2 = gr.n2; ct.out = (int32_t)first[r];
                        ct.fb = (int32_t)fb_total; ct.ra = ra_off[(size_t)r]; ct.fs = fs_total; ct.rs = rs_off[(size_t)r];
                        ct.rs_first = gr.k2lo - kr.k2lo; ct.rs_stride = grid->step2; ct.rs_plane = 64 * bk.R;
                        jcomb.push_back(ct);
                        fs_total += (uint64_t)gr.n1 * q3; fb_total += gr.n1;
                        // algorithmic cells (closed form, as below)
                        {
                            const int64_t n1 = gr.n1, n2 = gr.n2;
                            const int64_t sum_k1 = n1 * gr.k1lo + (int64_t)grid->step1 * n1 * (n1 - 1) / 2;
                            const int64_t sum_k2 = n2 * gr.k2lo + (int64_t)grid->step2 * n2 * (n2 - 1) / 2;
                            alg_cells += (int64_t)reads[r].qlen * (n1 * n2 * ((int64_t)d.l1 + d.l2 + d.l3) + d.m1 * sum_k1 * n2 + d.m2 * sum_k2 * n1);
                        }
                        continue;
                    }
                } else {
                    ks.assign(cell_k1 + first[r], cell_k1 + first[r] + cnt[r]);
                    if (!std::is_sorted(ks.begin(), ks.end())) std::sort(ks.begin(), ks.end());
                    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
                }
                if (used > 0 && used + slot * ks.size() > state_cap) {      // close the group (never with kept states: budgeted above)
                    g.n_pre = (int)(jpre.size() - g.pre_off); g.n_tail = (int)(jtail.size() - g.tail_off);
                    b->jgroups.push_back(g);
                    g.pre_off = jpre.size(); g.tail_off = jtail.size();
                    used = 0;
                }
                NraJointTask tp{}; tp.read = r; tp.k1_off = (int32_t)k1list.size(); tp.nk1 = (int32_t)ks.size();
                tp.state = used; tp.k2step = 1;
                if (b->jpack_l) { tp.resume = 1; tp.pstate = pair.state; tp.phalf = pair.read_b == r ? 1 : 0; }
                jpre.push_back(tp);
                bk.cells_sweep += (int64_t)64 * bk.R * (d.l1 + d.m1 * ks.back() - 1 - (b->jpack_l ? colsL : 0));
                if (b->joint_v2) {
                    // one MID sweep per k1 (the last prefix column + mid; leaves its column state and B(k1)), and the
                    // read's combine task: n1 x n2 cells from the column states on either side of the junction
                    const GridRow& gr = grid->rows[(size_t)r];
                    const uint64_t q3 = (uint64_t)3 * (uint64_t)reads[r].qlen;
                    if (b->joint_chain) {
                        // the MID part of all the read's k1 in one wave, column by column (k_joint_midscan)
                        NraJointTask t{}; t.read = r; t.k1_off = tp.k1_off; t.nk1 = tp.nk1;
                        t.k1 = gr.k1lo; t.k2step = grid->step1;          // slot i of `state` holds k1lo + i * step1
                        t.out = (int32_t)fb_total; t.state = used; t.pstate = fs_total;
                        push_midscan(jtail, t, q3);
                        bk.cells_sweep += (int64_t)64 * bk.R * gr.n1 * (1 + d.l2);
                        bk.cells_sweep += (int64_t)64 * bk.R * (1 + std::min(63, std::max(reads[r].qlen - 1, 0) / bk.R));   // the prefix sweep's own drain
                    } else
                    for (int32_t i = 0; i < gr.n1; ++i) {
                        NraJointTask t{}; t.read = r; t.k1 = ks[(size_t)i]; t.k2step = 1;
                        t.out = (int32_t)(fb_total + i);
                        t.state = used + slot * (uint64_t)i;
                        t.pstate = fs_total + (uint64_t)i * q3;
                        jtail.push_back(t);
                        bk.cells_sweep += joint_cells(bk.R, 1 + d.l2, reads[r].qlen);
                    }
                    NraJointCombineTask ct{};
                    ct.read = r; ct.n1 = gr.n1; ct.n2 = gr.n2; ct.out = (int32_t)first[r];
                    ct.fb = (int32_t)fb_total; ct.ra = ra_off[(size_t)r]; ct.fs = fs_total; ct.rs = rs_off[(size_t)r];
                    ct.rs_first = 0; ct.rs_stride = 1; ct.rs_plane = 64 * bk.R;
                    jcomb.push_back(ct);
                    fs_total += (uint64_t)gr.n1 * q3; fb_total += gr.n1;
                } else if (grid) {
                    // one tail sweep per k1: the read's k2 values are one arithmetic progression
                    const GridRow& gr = grid->rows[(size_t)r];
                    for (int32_t i = 0; i < gr.n1; ++i) {
                        NraJointTask t{}; t.read = r; t.k1 = ks[(size_t)i]; t.k2lo = gr.k2lo; t.k2step = grid->step2; t.n2 = gr.n2;
                        t.out = (int32_t)(first[r] + (uint32_t)i * (uint32_t)gr.n2);
                        t.state = used + slot * (uint64_t)i;
                        jtail.push_back(t);
                        bk.cells_sweep += joint_cells(bk.R, 1 + d.l2 + d.m2 * (t.k2lo + t.k2step * (t.n2 - 1)), reads[r].qlen);
                    }
                } else
                for (uint32_t c = first[r]; c < first[r] + cnt[r];) {
                    NraJointTask t{}; t.read = r; t.k1 = cell_k1[c]; t.k2lo = cell_k2[c]; t.k2step = 1; t.n2 = 1;
                    t.out = (int32_t)c;
                    uint32_t e = c + 1;
                    if (e < first[r] + cnt[r] && cell_k1[e] == t.k1 && cell_k2[e] > t.k2lo) {
                        t.k2step = cell_k2[e] - t.k2lo;
                        while (e < first[r] + cnt[r] && cell_k1[e] == t.k1 &&
                               cell_k2[e] == t.k2lo + t.k2step * (int32_t)(e - c)) ++e;
                    }
                    t.n2 = (int32_t)(e - c);
                    t.state = used + slot * (uint64_t)(std::lower_bound(ks.begin(), ks.end(), t.k1) - ks.begin());
                    jtail.push_back(t);
                    bk.cells_sweep += joint_cells(bk.R, 1 + d.l2 + d.m2 * (t.k2lo + t.k2step * (t.n2 - 1)), reads[r].qlen);
                    c = e;
                }
                k1list.insert(k1list.end(), ks.begin(), ks.end());
                used += slot * ks.size();
                state_max = std::max(state_max, used);
            }
            // algorithmic cells: the rectangle of every (read, cell) alignment
            if (grid && !per_cell) {
                // a product grid: the sum of the template lengths in closed form
                const GridRow& gr = grid->rows[(size_t)r];
                const int64_t n1 = gr.n1, n2 = gr.n2;
                const int64_t sum_k1 = n1 * gr.k1lo + (int64_t)grid->step1 * n1 * (n1 - 1) / 2;
                const int64_t sum_k2 = n2 * gr.k2lo + (int64_t)grid->step2 * n2 * (n2 - 1) / 2;
                alg_cells += (int64_t)reads[r].qlen * (n1 * n2 * ((int64_t)d.l1 + d.l2 + d.l3) + d.m1 * sum_k1 * n2 + d.m2 * sum_k2 * n1);
            } else {
                int64_t tl_sum = 0;
                for (uint32_t i = 0; i < cnt[r]; ++i) {
                    const int32_t k1 = k1_of(r, i), k2 = k2_of(r, i);
                    const int tl = d.l1 + d.m1 * k1 + d.l2 + d.m2 * k2 + d.l3;
                    tl_sum += tl;
                    if (per_cell) {
                        queue_tasks.push_back(NraTask{r, k1, k2, (int32_t)(first[r] + i)});
                        bk.cells_queue += sweep_cells(bk.R, tl);
                    }
                }
                alg_cells += (int64_t)reads[r].qlen * tl_sum;
            }
        }
        if (!per_cell) {
            g.n_pre = (int)(jpre.size() - g.pre_off); g.n_tail = (int)(jtail.size() - g.tail_off);
            if (g.n_pre > 0 || g.n_tail > 0) b->jgroups.push_back(g);
            for (size_t i = bucket_pre0; i < jpre.size(); ++i) {
                jpre[i].state += state_base;
                if (keep == 1) b->keep_state_off[(size_t)jpre[i].read] = jpre[i].state;
            }
            for (size_t i = bucket_tail0; i < jtail.size(); ++i) {
                if (jtail[i].state >> 63) jtail[i].state &= ~(1ull << 63);      // a kept state: absolute
                else jtail[i].state += state_base;
            }
            state_base += state_max;
        }
        bk.n_queue = (int)(queue_tasks.size() - bk.queue_off);
        bk.queue_cap = (size_t)bk.n_queue;
        bk.n_comb = (int)(jcomb.size() - bk.comb_off);
        queue_count.push_back(bk.n_queue);
    }

    clk.mark("2D cells: prefix and tail sweeps built");
    HIP_TRY(b->queue_tasks.upload(queue_tasks));
    HIP_TRY(b->queue_count.upload(queue_count));
    if (!b->brute) {
        HIP_TRY(b->jpre_tasks.upload(jpre));
        HIP_TRY(b->jtail_tasks.upload(jtail));
        HIP_TRY(b->jk1list.upload(k1list));
        if (keep != 2) {
            ArenaScope kept(keep == 1 ? &b->keep_arena : &b->cell_arena);
            if (keep == 1) b->keep_arena.expect((size_t)state_base * 4 + (1u << 20));
            HIP_TRY(b->jstate.alloc((size_t)state_base));
        }
        if (b->joint_v2) {
            HIP_TRY(b->jfs.alloc((size_t)fs_total));
            HIP_TRY(b->jfb.alloc((size_t)fb_total));
            HIP_TRY(b->jcomb_tasks.upload(jcomb));
        }
    }
    b->cur_rows.clear();
    if (grid) { b->cur_rows.assign(grid->rows, grid->rows + n_reads); b->cur_step1 = grid->step1; b->cur_step2 = grid->step2; }
    b->grid_step1 = b->grid_step2 = 0;
    if (grid) {                                        // the selector computes k1 / k2 of a cell from the read's row
        std::vector<GridRow> rows(grid->rows, grid->rows + n_reads);
        HIP_TRY(b->grid_rows.upload(rows));
        b->grid_step1 = grid->step1; b->grid_step2 = grid->step2;
    } else {
        std::vector<int32_t> v(cell_k1, cell_k1 + n_cells); HIP_TRY(b->cell_k1.upload(v));
        std::vector<int32_t> w(cell_k2, cell_k2 + n_cells); HIP_TRY(b->cell_k2.upload(w));
    }
    HIP_TRY(b->cell_first.upload(first));
    HIP_TRY(b->cell_cnt.upload(cnt));
    HIP_TRY(b->cand_score.alloc((size_t)n_cells));     // cell_score
    HIP_TRY(b->cand_tstart.alloc((size_t)n_cells));    // cell_wscore
    clk.mark("2D cells: device buffers, H2D");
    {
        const int rc1 = ensure_handles((b->warm_pending || b->flanks_enqueued ? (size_t)b->ev_next : 2) + 12 * nb + 4 * b->jgroups.size() + 2);
        if (rc1) return rc1;
    }

    b->stats.n_alignments = n_cells;
    b->stats.algorithmic_cells = alg_cells;
    int64_t ex = 0;
    for (const Bucket& bk : b->buckets) ex += bk.cells_pair + bk.cells_queue + bk.cells_sweep;
    if (had_warm) ex += b->warm_cells;
    b->stats.executed_cells = ex;
    b->stats.algorithmic_bytes = (int64_t)b->n_q2bit_words * 4 + (int64_t)pool.size() + n_cells * 8 + (int64_t)n_reads * 25;
    int64_t packed_ints = 0;
    for (const Bucket& bk : b->buckets) packed_ints += (int64_t)(bk.n_jlpk + bk.n_jrpk) * NRA_JOINT_NPSTATE(bk.R) * 64;
    b->stats.intermediate_bytes = b->brute ? 0 : 2 * ((int64_t)state_base * 4 + packed_ints * 4 +
                                                      (b->joint_v2 ? (int64_t)(rs_total + fs_total) * 4 : (int64_t)b->n_q2bit_words * 16 * 3 * 4));
    b->have_cells = true;
    return NRA_OK;
}

}  // namespace

extern "C" {

// Forget what earlier cell lists left for later ones (the reverse sweeps): the next list starts like the first.
int nra_batch2d_invalidate(nra_batch_t* b)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    std::fill(b->rev_strand.begin(), b->rev_strand.end(), (int8_t)0);
    std::fill(b->lst_strand.begin(), b->lst_strand.end(), (int8_t)0);
    std::fill(b->rpk_strand.begin(), b->rpk_strand.end(), (int8_t)0);
    b->keep_valid = false;
    return NRA_OK;
}

// The flank sweeps of a joint run ahead of its cell list.  The packed sweeps of L and rev(R) outside the scoring window
// (k_joint_pk16) depend on the reads and their strands only -- not on ranges, grids or cells -- and are the first third
// of a round's device time: a caller that knows the strands (round 1 does) enqueues them HERE, before it derives step
// sizes, bounds and grids on the host, and the device works through that host time instead of idling.  The cell lists
// that follow find the flank states valid (like a later list of the same batch) and sweep no flank.
int nra_batch2d_sweep_flanks(nra_batch_t* b, const int8_t* read_strand)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    const int32_t n_reads = b->n_reads;
    if (n_reads > 0 && !read_strand) return fail(NRA_E_ARG, "NULL strand array");
    const int32_t left_len = (int32_t)b->jr_left.size(), right_len = (int32_t)b->jr_right.size();
    if ((b->flags & NRA_F_BRUTE_FORCE) != 0 || left_len < 1 || right_len < 2 || (!b->jpack_l && !b->jpack_r) || n_reads == 0)
        return NRA_OK;                                         // nothing this batch sweeps ahead of time
    HIP_TRY(hipSetDevice(b->device));
    if (b->ran) HIP_TRY(hipStreamSynchronize(b->stream));      // an earlier list's kernels read the flank states
    if (b->flanks_enqueued || b->warm_pending) {
        for (hipStream_t q : b->bstreams) HIP_TRY(hipStreamSynchronize(q));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    {
        int rc0 = account_run_fwd(b);
        if (rc0) return rc0;
    }
    ArenaScope arena_scope(&b->arena);
    if (!b->warm_built) return NRA_OK;
    // the reads as the kernels orient them (what k_pick_strand leaves in the cell list's own copy)
    std::vector<NraDevRead> reads(b->host_reads.begin(), b->host_reads.begin() + n_reads);
    for (int32_t r = 0; r < n_reads; ++r) reads[(size_t)r].rc = read_strand[r] < 0 ? 1 : 0;
    // per rows-per-lane bucket (pairs are listed bucket by bucket): the pairs whose reads all come with a strand and
    // whose state is not valid for it
    struct Part { int R; size_t off, n_l, n_r; };
    std::vector<Part> parts;
    std::vector<NraJointPairTask> mt;              // L- and R-side tasks of a pair next to each other; L marked in bit 63 of `state`
    std::vector<std::pair<int32_t, int8_t>> l_done, r_done;
    for (size_t i = 0; i < b->jpairs.size();) {
        const int bi = b->jbucket[(size_t)b->jpairs[i].read_a];
        Part part{kRList[bi], mt.size(), 0, 0};
        for (; i < b->jpairs.size() && b->jbucket[(size_t)b->jpairs[i].read_a] == bi; ++i) {
            const NraJointPairTask& pair = b->jpairs[i];
            bool all = true, stale_l = false, stale_r = false;
            for (int32_t q : {pair.read_a, pair.read_b}) {
                if (q < 0) continue;
                if (read_strand[q] == 0) all = false;
                if (b->lst_strand[(size_t)q] != read_strand[q]) stale_l = true;
                if (b->rpk_strand[(size_t)q] != read_strand[q]) stale_r = true;
            }
            if (!all) continue;
            for (int32_t q : {pair.read_a, pair.read_b}) {
                if (q < 0) continue;
                if (b->jpack_l && stale_l) l_done.push_back({q, read_strand[q]});
                if (b->jpack_r && stale_r) r_done.push_back({q, read_strand[q]});
            }
            if (b->jpack_l && stale_l) { NraJointPairTask t = pair; t.state |= 1ull << 63; mt.push_back(t); part.n_l++; }
            if (b->jpack_r && stale_r) { mt.push_back(pair); part.n_r++; }
        }
        if (part.n_l || part.n_r) parts.push_back(part);
    }
    if (parts.empty()) return NRA_OK;
    std::reverse(parts.begin(), parts.end());                  // the longest reads first, like the cell lists' buckets
    HIP_TRY(copy_h2d(b->warm_reads.p, reads.data(), reads.size() * sizeof(NraDevRead)));
    HIP_TRY(copy_h2d(b->warm_tasks.p, mt.data(), mt.size() * sizeof(NraJointPairTask)));
    const size_t np = parts.size();
    while (b->bstreams.size() < 2 * np) { hipStream_t q; HIP_TRY(g_handles.stream(b->device, &q)); b->bstreams.push_back(q); }
    while (b->warm_ev.size() < 2 * np) { hipEvent_t e; HIP_TRY(g_handles.event(b->device, false, &e)); b->warm_ev.push_back(e); }
    while (b->ev.size() < 2 + 4 * np + 2) { hipEvent_t e; HIP_TRY(g_handles.event(b->device, true, &e)); b->ev.push_back(e); }
    hipStream_t st = b->stream;
    HIP_TRY(hipEventRecord(b->ev[0], st));
    HIP_TRY(hipEventRecord(b->phase_ev[0], st));
    HIP_TRY(hipEventRecord(b->fork2_ev, st));
    int ev = 2;
    b->n_score_ev = 0; b->n_ext_ev = 0;
    const int colsL = NRA_JOINT_PACKED_COLS(left_len), colsR = NRA_JOINT_PACKED_COLS(right_len);
    int64_t cells = 0;
    for (size_t i = 0; i < np; ++i) {
        const Part& pt = parts[i];
        hipStream_t qa = b->bstreams[2 * i], qb = b->bstreams[2 * i + 1];
        HIP_TRY(hipStreamWaitEvent(qa, b->fork2_ev, 0));
        HIP_TRY(hipStreamWaitEvent(qb, b->fork2_ev, 0));
        // both sides of a rows-per-lane class in ONE launch, L- and R-side tasks alternating: the two chains behind them
        // (prefix sweeps; extended reverse sweeps) carry about the same work and meet again in the cells, so neither side's
        // waves should be the older ones (two launches: the first one's waves issue first, its side finishes 1.4 ms ahead)
        HIP_TRY(hipEventRecord(b->ev[ev++], qa));
        LAUNCH_TRY(nra_launch_joint_pk16(pt.R, b->warm_has_n, qa, (int)(pt.n_l + pt.n_r), b->warm_tasks.p + pt.off, b->warm_reads.p,
                                         b->warm_region.p, b->warm_pool.p, b->q2bit.p, b->qnmask.p, b->sp, -1, b->jrstate.p, b->jlstate.p));
        HIP_TRY(hipEventRecord(b->ev[ev++], qa));
        b->n_score_ev++;
        cells += (int64_t)pt.n_l * 2 * 64 * pt.R * colsL + (int64_t)pt.n_r * 2 * 64 * pt.R * colsR;
        HIP_TRY(hipEventRecord(b->warm_ev[2 * i], qa));
        HIP_TRY(hipEventRecord(b->warm_ev[2 * i + 1], qa));
    }
    b->warm_R.clear();
    for (const Part& pt : parts) b->warm_R.push_back(pt.R);
    b->ev_next = ev;
    b->warm_pending = true;
    b->warm_cells = cells;
    for (const auto& pr2 : l_done) b->lst_strand[(size_t)pr2.first] = pr2.second;
    for (const auto& pr2 : r_done) b->rpk_strand[(size_t)pr2.first] = pr2.second;
    return NRA_OK;
}

int nra_batch2d_create(int device, const nra_joint_region_t* reg, int32_t n_reads, const char* seqs,
                       const int64_t* seq_off, const int8_t* read_strand, int64_t n_cells,
                       const int32_t* cell_read, const int32_t* cell_k1, const int32_t* cell_k2,
                       const nra_scoring_t* sc, int32_t flags, nra_batch_t** out)
{
    if (!out) return fail(NRA_E_ARG, "out is NULL");
    *out = nullptr;
    if (n_cells < 0) return fail(NRA_E_ARG, "bad region / counts");
    if (n_cells > 0 && (!cell_read || !cell_k1 || !cell_k2)) return fail(NRA_E_ARG, "NULL cell array");
    nra_batch_t* b = nullptr;
    int rc = nra_batch2d_create_reads(device, reg, n_reads, seqs, seq_off, sc, flags, &b);
    if (rc) return rc;
    rc = nra_batch2d_set_cells(b, read_strand, n_cells, cell_read, cell_k1, cell_k2);
    if (rc) { nra_batch_destroy(b); return rc; }
    *out = b;
    return NRA_OK;
}

}  // extern "C"

namespace {

// The kernels of a 2D run come in two parts.  Part 1 -- strand probes, strand choice, and per bucket the packed
// sweeps of the flanks outside the scoring window and the reverse sweeps over R -- depends on the reads and their
// strands only; nra_batch2d_set_cells / _set_grid enqueue it themselves when every strand is given, before they
// build the rest of the cell list's tasks.  Part 2 -- prefix sweeps, tail sweeps, per-cell payload launches,
// the selector -- is enqueued by nra_batch_run.
int run_2d_flanks(nra_batch* b)
{
    hipStream_t st = b->stream;
    const size_t nb = b->buckets.size();
    const size_t nr = std::max<size_t>((size_t)b->n_reads, 1);
    int ev = 2;
    const bool warm = b->warm_pending;           // flank sweeps of this run went out ahead of the cell list: the run began there
    if (warm) {
        // a bucket's two chains wait for the sweeps of its own rows-per-lane class only (the R side's for the reverse
        // chain, the L side's for the prefix chain): the classes overlap as they do within one run
        ev = b->ev_next;
        for (size_t i = 0; i < nb; ++i)
            for (size_t w = 0; w < b->warm_R.size(); ++w)
                if (b->warm_R[w] == b->buckets[i].R && !b->buckets[i].chain) {
                    HIP_TRY(hipStreamWaitEvent(b->bstreams[2 * i], b->warm_ev[2 * w], 0));
                    HIP_TRY(hipStreamWaitEvent(b->bstreams[2 * i + 1], b->warm_ev[2 * w + 1], 0));
                }
        b->warm_pending = false;
    } else {
        HIP_TRY(hipEventRecord(b->ev[0], st));
        b->n_score_ev = 0;
    }
    b->n_ext_ev = 0;
    HIP_TRY(hipMemcpyAsync(b->reads.p, b->reads_init.p, nr * sizeof(NraDevRead), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(b->probe_score.p, 0xff, 2 * nr * 4, st));
    if (!b->all_strands_given) {
        HIP_TRY(hipEventRecord(b->fork_ev, st));
        for (size_t i = 0; i < nb; ++i) {
            const Bucket& bk = b->buckets[i];
            hipStream_t q = b->bstreams[2 * i];
            HIP_TRY(hipStreamWaitEvent(q, b->fork_ev, 0));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            if (bk.chain) {
                NraScoreParams raw = b->sp;
                raw.min_score = 0;               // the probe compares raw scores
                LAUNCH_TRY(nra_launch_payload_origin(bk.R, b->has_n, q, std::min(bk.n_probe, b->payload_strips),
                                                     b->probe_tasks.p + bk.probe_off, b->probe_count.p + i,
                                                     b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                     raw, b->probe_score.p, b->probe_dummy.p, nullptr,
                                                     b->chain_payload.p, b->chain_cap, 1));
            } else
            LAUNCH_TRY(nra_launch_score_pk16(bk.R, b->has_n, q, bk.n_pair, b->pair_tasks.p + bk.pair_off,
                                             b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                             b->sp, b->probe_score.p));
            HIP_TRY(hipEventRecord(b->ev[ev++], q));
            b->n_ext_ev++;
            HIP_TRY(hipEventRecord(b->bdone[3 * i + 2], q));
            HIP_TRY(hipStreamWaitEvent(st, b->bdone[3 * i + 2], 0));
        }
    }
    LAUNCH_TRY(nra_launch_pick_strand(st, b->n_reads, b->probe_score.p,
                                      b->have_strand_in ? b->strand_in.p : nullptr, b->strand_out.p, b->reads.p));
    if (!warm) HIP_TRY(hipEventRecord(b->phase_ev[0], st));
    if (!b->brute) {
        // reverse sweeps over R (one per read) and the packed sweeps of both flanks (one per pair of reads)
        HIP_TRY(hipEventRecord(b->fork2_ev, st));
        for (size_t i = 0; i < nb; ++i) {
            const Bucket& bk = b->buckets[i];
            if (bk.chain) continue;
            hipStream_t qa = b->bstreams[2 * i], qb = b->bstreams[2 * i + 1];
            HIP_TRY(hipStreamWaitEvent(qa, b->fork2_ev, 0));
            HIP_TRY(hipStreamWaitEvent(qb, b->fork2_ev, 0));
            // the L side first: the older kernel's waves issue first, and the chain behind it (prefix sweep, MID scans,
            // combine) is the longer one (config 3: 7.3 -> 7.1 ms of device time against the R side first)
            if (bk.n_jlpk > 0) {
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                LAUNCH_TRY(nra_launch_joint_pk16(bk.R, b->has_n, qb, bk.n_jlpk, b->jlpk_tasks.p + bk.jlpk_off, b->reads.p,
                                                 b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp, 1, b->jlstate.p, nullptr));
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                b->n_score_ev++;
            }
            if (bk.n_jrpk > 0) {
                HIP_TRY(hipEventRecord(b->ev[ev++], qa));
                LAUNCH_TRY(nra_launch_joint_pk16(bk.R, b->has_n, qa, bk.n_jrpk, b->jrpk_tasks.p + bk.jrpk_off, b->reads.p,
                                                 b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p, b->sp, 0, b->jrstate.p, nullptr));
                HIP_TRY(hipEventRecord(b->ev[ev++], qa));
                b->n_score_ev++;
            }
            HIP_TRY(hipEventRecord(b->ev[ev++], qa));
            if (b->joint_v2)
                LAUNCH_TRY(nra_launch_joint_bwd_ext(bk.R, b->has_n, qa, bk.n_jbwd, b->jbwd_tasks.p + bk.jbwd_off,
                                                    b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                    b->sp, b->jrs.p, b->jra.p, b->jrstate.p));
            else
            LAUNCH_TRY(nra_launch_joint_bwd(bk.R, b->has_n, qa, bk.n_jbwd, b->jbwd_tasks.p + bk.jbwd_off,
                                            b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                            b->sp, b->jsnap.p, b->jread_a.p, b->jrstate.p));
            HIP_TRY(hipEventRecord(b->ev[ev++], qa));
            b->n_score_ev++;
            HIP_TRY(hipEventRecord(b->bdone[3 * i], qa));
        }
    }
    b->ev_next = ev;
    b->flanks_enqueued = true;
    return NRA_OK;
}

int run_2d(nra_batch* b)
{
    if (!b->flanks_enqueued) {
        const int rc = run_2d_flanks(b);
        if (rc) return rc;
    }
    b->flanks_enqueued = false;          // (a second nra_batch_run of the same cell list starts over)
    hipStream_t st = b->stream;
    const size_t nb = b->buckets.size();
    const size_t nc = std::max<size_t>((size_t)b->n_cands, 1);
    int ev = b->ev_next;
    const int max_waves = 256 * 16;
    if (b->cells_need_clear) {
        HIP_TRY(hipMemsetAsync(b->cand_score.p, 0xff, nc * 4, st));
        HIP_TRY(hipMemsetAsync(b->cand_tstart.p, 0, nc * 4, st));
        HIP_TRY(hipEventRecord(b->fork_ev, st));           // the cell arrays are cleared (the tails write them)
    }
    // buckets scored cell by cell: all of them in brute-force mode, else the chained (long) reads only
    for (size_t i = 0; i < nb; ++i) {
        const Bucket& bk = b->buckets[i];
        if (!b->brute && !bk.chain) continue;
        HIP_TRY(hipEventRecord(b->ev[ev++], st));
        LAUNCH_TRY(nra_launch_payload_window(bk.R, b->has_n, st, std::min(bk.n_queue, bk.chain ? b->payload_strips : max_waves),
                                          b->queue_tasks.p + bk.queue_off, b->queue_count.p + i,
                                          b->reads.p, b->regions.p, b->pool.p,
                                          b->q2bit.p, b->qnmask.p, b->sp, b->cand_score.p,
                                          b->cand_tstart.p, nullptr, bk.chain ? b->chain_payload.p : nullptr,
                                          bk.chain ? b->chain_cap : 0, bk.chain ? 1 : 0));
        HIP_TRY(hipEventRecord(b->ev[ev++], st));
        b->n_score_ev++;
    }
    if (!b->brute) {
        // group by group, the prefix sweeps (one per read) that leave the wave states and the tail sweeps (one
        // per (read, k1) run of cells) that resume from them
        for (size_t i = 0; i < nb; ++i) {
            const Bucket& bk = b->buckets[i];
            if (bk.chain) continue;
            hipStream_t qb = b->bstreams[2 * i + 1];
            HIP_TRY(hipStreamWaitEvent(st, b->bdone[3 * i], 0));
            if (b->cells_need_clear) HIP_TRY(hipStreamWaitEvent(qb, b->fork_ev, 0));
            bool first = true;
            for (const JointGroup& g : b->jgroups) {
                if (g.bucket != (int)i) continue;
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                if (b->joint_chain)
                    LAUNCH_TRY(nra_launch_joint_prefix_cols(g.R, b->has_n, qb, g.n_pre, b->jpre_tasks.p + g.pre_off,
                                                            b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                            b->sp, b->jk1list.p, b->jstate.p, b->jlstate.p));
                else
                LAUNCH_TRY(nra_launch_joint_prefix(g.R, b->has_n, qb, g.n_pre, b->jpre_tasks.p + g.pre_off,
                                                   b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                   b->sp, b->jk1list.p, b->jstate.p, b->jlstate.p));
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                b->n_score_ev++;
                if (first && !b->joint_v2) HIP_TRY(hipStreamWaitEvent(qb, b->bdone[3 * i], 0));     // the tails read the R side
                first = false;
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                if (b->joint_chain)
                    LAUNCH_TRY(nra_launch_joint_midscan(g.R, b->has_n, qb, g.n_tail, b->jtail_tasks.p + g.tail_off,
                                                         b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                         b->sp, b->jk1list.p, b->jstate.p, b->jfs.p, b->jfb.p, nullptr));
                else if (b->joint_v2)
                    LAUNCH_TRY(nra_launch_joint_mid(g.R, b->has_n, qb, g.n_tail, b->jtail_tasks.p + g.tail_off,
                                                    b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                    b->sp, b->jstate.p, b->jfs.p, b->jfb.p));
                else
                LAUNCH_TRY(nra_launch_joint_tail(g.R, b->has_n, qb, g.n_tail, b->jtail_tasks.p + g.tail_off,
                                                 b->reads.p, b->regions.p, b->pool.p, b->q2bit.p, b->qnmask.p,
                                                 b->sp, b->jstate.p, b->jsnap.p, b->jread_a.p, b->cand_score.p,
                                                 b->cand_tstart.p));
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                b->n_score_ev++;
            }
            if (b->joint_v2 && bk.n_comb > 0) {
                // every cell of the bucket's reads from the column states on either side of the junction
                HIP_TRY(hipStreamWaitEvent(qb, b->bdone[3 * i], 0));                // ... the extended reverse sweeps'
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                LAUNCH_TRY(nra_launch_joint_combine(qb, bk.n_comb, b->jcomb_tasks.p + bk.comb_off, b->reads.p, b->sp,
                                                    b->jfs.p, b->jrs.p, b->jfb.p, b->jra.p, b->cand_score.p, b->cand_tstart.p, nullptr));
                HIP_TRY(hipEventRecord(b->ev[ev++], qb));
                b->n_score_ev++;
            }
            HIP_TRY(hipEventRecord(b->bdone[3 * i + 1], qb));
            HIP_TRY(hipStreamWaitEvent(st, b->bdone[3 * i + 1], 0));
        }
    }
    // the reverse sweeps enqueued above stay valid for later cell lists of this batch (given strands only:
    // a probed strand is not known on the host)
    for (const auto& pr2 : b->rev_pending) b->rev_strand[(size_t)pr2.first] = pr2.second;
    b->rev_pending.clear();
    for (const auto& pr2 : b->lst_pending) b->lst_strand[(size_t)pr2.first] = pr2.second;
    b->lst_pending.clear();
    for (const auto& pr2 : b->rpk_pending) b->rpk_strand[(size_t)pr2.first] = pr2.second;
    b->rpk_pending.clear();
    if (b->keep_pending) { b->keep_valid = true; b->keep_pending = false; }      // ... and so do the column states
    HIP_TRY(hipEventRecord(b->phase_ev[1], st));
    LAUNCH_TRY(nra_launch_select_2d(st, b->n_reads, b->cell_first.p, b->cell_cnt.p, b->cell_k1.p, b->cell_k2.p,
                                    b->grid_step1 > 0 ? b->grid_rows.p : nullptr, b->grid_step1, b->grid_step2,
                                    b->cand_score.p, b->cand_tstart.p, b->best_score.p, b->sum_k.p,
                                    b->sum_k2.p, b->n_ties.p, b->status.p));
    HIP_TRY(hipEventRecord(b->ev[1], st));
    b->ev_next = ev;
    b->refined = false; b->refine_read = false; b->refine_bad_rows = 0;
    return NRA_OK;
}

// the counters a refinement left on the device (after its stream is idle): cells scored, rows that were not routable
int finish_refine(nra_batch* b)
{
    if (!b->refined || b->refine_read) return NRA_OK;
    int64_t w[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(w, b->rf_words.p, sizeof(w), hipMemcpyDeviceToHost));
    b->stats.n_alignments += w[0];
    b->stats.executed_cells += w[2];
    b->stats.algorithmic_cells += w[3];
    b->refine_bad_rows = w[1];
    b->refine_read = true;
    return NRA_OK;
}

}  // namespace

extern "C" {

// The reference's round 3 (nanoRepeat_joint.py:275-349) as a refinement of the routed grid whose run has just been
// enqueued: routed on the device from that grid's per-read results, scored from the column states its sweeps kept.
int nra_batch2d_refine(nra_batch_t* b, int32_t buf1, int32_t buf2, const double* lo1, const double* hi1,
                       const double* lo2, const double* hi2)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    if (buf1 <= 0 || buf2 <= 0 || buf1 > 4096 || buf2 > 4096) return fail(NRA_E_ARG, "bad refinement buffers");
    const int32_t n_reads = b->n_reads;
    if (n_reads > 0 && (!lo1 || !hi1 || !lo2 || !hi2)) return fail(NRA_E_ARG, "NULL bound array");
    if (!b->ran || b->accounted || b->refined || !b->joint_v2 || !b->joint_chain || b->joint_keep == 0 || !b->keep_valid ||
        b->brute || !b->all_strands_given || b->keep_rows.size() != (size_t)n_reads)
        return fail(NRA_E_STATE, "no refinement: it follows nra_batch_run of a routed grid (nra_batch2d_set_grid, every strand "
                                 "given) whose column states are kept, before anything waits for that run");
    for (const Bucket& bk : b->buckets)
        if (bk.chain) return fail(NRA_E_STATE, "no refinement: reads beyond one register block are scored cell by cell");
    // Every count a read's refinement can ask for -- within buf of a size between its first and last grid value, inside
    // [lo, hi) -- has to be among the counts column states were kept at.  True by construction behind the grid that kept
    // them when buf = its steps and [lo, hi) = its bounds; not necessarily behind a grid that itself ran from states an
    // EARLIER grid kept, or with other buffers / bounds: then the caller takes the two-call path.
    if (b->cur_rows.size() != (size_t)n_reads) return fail(NRA_E_STATE, "no refinement: the current cell list is not a routed grid");
    for (int32_t r = 0; r < n_reads; ++r) {
        const NraGridRow& g = b->cur_rows[(size_t)r];
        const NraGridRow& k = b->keep_rows[(size_t)r];
        if (g.n1 <= 0 || g.n2 <= 0 || k.n1 <= 0 || k.n2 <= 0) continue;
        auto fits = [](int32_t first, int32_t n, int32_t step, int32_t buf, double lo, double hi, int32_t klo, int32_t kn) {
            if (!(lo < hi)) return true;
            const int64_t last = (int64_t)first + (int64_t)(n - 1) * step;
            const double cl = std::ceil(lo), ch = std::ceil(hi);
            const int64_t need_lo = std::max<int64_t>({cl < 0 ? 0 : (cl > 1e9 ? (int64_t)1e9 : (int64_t)cl), (int64_t)first - buf, 0});
            const int64_t need_hi = std::min<int64_t>(ch > 1e9 ? (int64_t)1e9 : (int64_t)ch - 1, last + buf - 1);
            return need_hi < need_lo || (need_lo >= klo && need_hi <= (int64_t)klo + kn - 1);
        };
        if (!fits(g.k1lo, g.n1, b->cur_step1, buf1, lo1[r], hi1[r], k.k1lo, k.n1) ||
            !fits(g.k2lo, g.n2, b->cur_step2, buf2, lo2[r], hi2[r], k.k2lo, k.n2))
            return fail(NRA_E_STATE, "no refinement: a read's refinement could ask for repeat counts no column state was kept at");
    }
    HIP_TRY(hipSetDevice(b->device));
    ArenaScope arena_scope(&b->cell_arena);
    const int cap1 = 2 * buf1, cap2 = 2 * buf2;
    const int64_t cap = (int64_t)cap1 * cap2;
    if (cap * n_reads > 0x7ff00000ll) return fail(NRA_E_RANGE, "too many cells");
    const size_t nb = b->buckets.size();
    const int kChunk = 3;                       // k1 values per MID-scan task (push_midscan)

    // static tasks: what a read's refinement needs whatever its round-2 size turns out to be
    std::vector<std::vector<NraJointTask>> mid(nb);
    std::vector<std::vector<NraJointCombineTask>> comb(nb);
    std::vector<int> bucket_of_R((size_t)NRA_MAX_R + 1, -1);
    for (size_t i = 0; i < nb; ++i) bucket_of_R[(size_t)b->buckets[i].R] = (int)i;
    std::vector<uint32_t> first((size_t)n_reads);
    std::vector<int32_t> rowspad((size_t)n_reads, 0);
    uint64_t fs_total = 0; int64_t fb_total = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        first[(size_t)r] = (uint32_t)((int64_t)r * cap);
        const NraGridRow& k = b->keep_rows[(size_t)r];
        if (k.n1 <= 0 || k.n2 <= 0 || b->host_reads[(size_t)r].qlen <= 0 || b->chained_reads[(size_t)r]) continue;
        const int R = kRList[b->jbucket[(size_t)r]];
        const int bi = bucket_of_R[(size_t)R];
        if (bi < 0) return fail(NRA_E_STATE, "no refinement: a read's bucket did not run");
        rowspad[(size_t)r] = 64 * R;
        const uint64_t q3 = (uint64_t)3 * (uint64_t)b->host_reads[(size_t)r].qlen;
        for (int c = 0; c < cap1; c += kChunk) {
            NraJointTask t{}; t.read = r; t.k1_off = c; t.nk1 = std::min(kChunk, cap1 - c);
            t.k1 = k.k1lo; t.k2step = 1; t.state = b->keep_state_off[(size_t)r];
            t.pstate = fs_total + (uint64_t)c * q3; t.out = (int32_t)(fb_total + c);
            mid[(size_t)bi].push_back(t);
        }
        NraJointCombineTask ct{};
        ct.read = r; ct.out = (int32_t)first[(size_t)r]; ct.fb = (int32_t)fb_total; ct.ra = b->keep_ra_off[(size_t)r];
        ct.fs = fs_total; ct.rs = b->keep_rs_off[(size_t)r]; ct.rs_first = k.k2lo; ct.rs_stride = 1; ct.rs_plane = 64 * R;
        comb[(size_t)bi].push_back(ct);
        fs_total += (uint64_t)cap1 * q3; fb_total += cap1;
    }
    std::vector<NraJointTask> mid_all; std::vector<NraJointCombineTask> comb_all;
    std::vector<size_t> mid_off(nb), comb_off(nb);
    for (size_t i = 0; i < nb; ++i) {
        mid_off[i] = mid_all.size(); comb_off[i] = comb_all.size();
        mid_all.insert(mid_all.end(), mid[i].begin(), mid[i].end());
        comb_all.insert(comb_all.end(), comb[i].begin(), comb[i].end());
    }
    std::vector<double> bounds((size_t)4 * (size_t)std::max(n_reads, 1), 0.0);
    for (int32_t r = 0; r < n_reads; ++r) {
        bounds[(size_t)r] = lo1[r]; bounds[(size_t)n_reads + r] = hi1[r];
        bounds[(size_t)2 * n_reads + r] = lo2[r]; bounds[(size_t)3 * n_reads + r] = hi2[r];
    }
    b->cell_arena.expect((size_t)fs_total * 4 + (size_t)cap * n_reads * 8 + (size_t)n_reads * 128 + (4u << 20));
    HIP_TRY(b->rf_bounds.upload(bounds));
    HIP_TRY(b->rf_keep.upload(b->keep_rows));
    HIP_TRY(b->rf_first.upload(first));
    HIP_TRY(b->rf_rowspad.upload(rowspad));
    HIP_TRY(b->rf_mid_tasks.upload(mid_all));
    HIP_TRY(b->rf_comb_tasks.upload(comb_all));
    HIP_TRY(b->rf_rows.alloc((size_t)std::max(n_reads, 1)));
    HIP_TRY(b->rf_cnt.alloc((size_t)std::max(n_reads, 1)));
    HIP_TRY(b->rf_words.alloc(4));
    HIP_TRY(b->rf_fs.alloc((size_t)fs_total));
    HIP_TRY(b->rf_fb.alloc((size_t)fb_total));
    // the refinement's cells: read-major, `cap` entries a read (its n1 x n2 cells first, k1-major)
    HIP_TRY(b->rf_cell_score.alloc((size_t)(cap * n_reads)));
    HIP_TRY(b->rf_cell_wscore.alloc((size_t)(cap * n_reads)));
    b->rf_n_cands = cap * n_reads;
    {
        const size_t need = (size_t)b->ev_next + 4 * nb + 4;
        while (b->ev.size() < need) { hipEvent_t e; HIP_TRY(g_handles.event(b->device, true, &e)); b->ev.push_back(e); }
    }

    hipStream_t st = b->stream;
    int ev = b->ev_next;
    const size_t nc = (size_t)std::max<int64_t>(cap * n_reads, 1);
    HIP_TRY(hipMemsetAsync(b->rf_words.p, 0, 4 * sizeof(int64_t), st));
    HIP_TRY(hipMemsetAsync(b->rf_cell_score.p, 0xff, nc * 4, st));
    HIP_TRY(hipMemsetAsync(b->rf_cell_wscore.p, 0, nc * 4, st));
    const double* bd = b->rf_bounds.p;
    LAUNCH_TRY(nra_launch_joint_refine_route(st, n_reads, b->status.p, b->n_ties.p, b->sum_k.p, b->sum_k2.p, bd, bd + n_reads,
                                             bd + 2 * (size_t)n_reads, bd + 3 * (size_t)n_reads, buf1, buf2, b->rf_keep.p,
                                             b->rf_rows.p, b->rf_cnt.p, b->rf_rowspad.p, b->reads.p, b->regions.p,
                                             reinterpret_cast<unsigned long long*>(b->rf_words.p)));
    HIP_TRY(hipEventRecord(b->fork2_ev, st));
    for (size_t i = 0; i < nb; ++i) {
        const Bucket& bk = b->buckets[i];
        const int n_mid = (int)mid[i].size(), n_comb = (int)comb[i].size();
        if (n_mid == 0) continue;
        hipStream_t qb = b->bstreams[2 * i + 1];
        HIP_TRY(hipStreamWaitEvent(qb, b->fork2_ev, 0));
        HIP_TRY(hipEventRecord(b->ev[ev++], qb));
        LAUNCH_TRY(nra_launch_joint_midscan(bk.R, b->has_n, qb, n_mid, b->rf_mid_tasks.p + mid_off[i], b->reads.p, b->regions.p,
                                            b->pool.p, b->q2bit.p, b->qnmask.p, b->sp, nullptr, b->jstate.p, b->rf_fs.p,
                                            b->rf_fb.p, b->rf_rows.p));
        HIP_TRY(hipEventRecord(b->ev[ev++], qb));
        b->n_score_ev++;
        HIP_TRY(hipEventRecord(b->ev[ev++], qb));
        LAUNCH_TRY(nra_launch_joint_combine(qb, n_comb, b->rf_comb_tasks.p + comb_off[i], b->reads.p, b->sp, b->rf_fs.p, b->jrs.p,
                                            b->rf_fb.p, b->jra.p, b->rf_cell_score.p, b->rf_cell_wscore.p, b->rf_rows.p));
        HIP_TRY(hipEventRecord(b->ev[ev++], qb));
        b->n_score_ev++;
        HIP_TRY(hipEventRecord(b->bdone[3 * i + 1], qb));
        HIP_TRY(hipStreamWaitEvent(st, b->bdone[3 * i + 1], 0));
    }
    HIP_TRY(hipEventRecord(b->phase_ev[1], st));
    LAUNCH_TRY(nra_launch_select_2d(st, n_reads, b->rf_first.p, b->rf_cnt.p, nullptr, nullptr, b->rf_rows.p, 1, 1,
                                    b->rf_cell_score.p, b->rf_cell_wscore.p, b->best_score.p, b->sum_k.p, b->sum_k2.p, b->n_ties.p,
                                    b->status.p));
    HIP_TRY(hipEventRecord(b->ev[1], st));
    b->ev_next = ev;
    b->refined = true; b->refine_read = false;
    return NRA_OK;
}

int nra_batch2d_fetch(nra_batch_t* b, int8_t* read_strand, int32_t* cell_score, int32_t* cell_wscore,
                      int32_t* best_wscore, int64_t* sum_k1, int64_t* sum_k2, int32_t* n_ties, uint8_t* status)
{
    if (!b || b->kind != 2) return fail(NRA_E_ARG, "not a 2D batch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    {
        int rc = finish_refine(b);
        if (rc) return rc;
        if (b->refine_bad_rows > 0)
            return fail(NRA_E_RANGE, std::to_string(b->refine_bad_rows) + " reads of the refinement ask for repeat counts outside the "
                                     "kept column states (bounds other than the grid's?)");
    }
    // (after a refinement the per-cell arrays are the refinement's: (2 buf1)(2 buf2) entries a read)
    const size_t n = (size_t)b->n_reads, nc = (size_t)(b->refined ? b->rf_n_cands : b->n_cands);
    const int32_t* src_score = b->refined ? b->rf_cell_score.p : b->cand_score.p;
    const int32_t* src_wscore = b->refined ? b->rf_cell_wscore.p : b->cand_tstart.p;
    if (n) {
        int rc = fetch_results(b);
        if (rc) return rc;
        if (read_strand) memcpy(read_strand, staged(b, b->strand_out), n);
        if (best_wscore) memcpy(best_wscore, staged(b, b->best_score), n * 4);
        if (sum_k1) memcpy(sum_k1, staged(b, b->sum_k), n * 8);
        if (sum_k2) memcpy(sum_k2, staged(b, b->sum_k2), n * 8);
        if (n_ties) memcpy(n_ties, staged(b, b->n_ties), n * 4);
        if (status) memcpy(status, staged(b, b->status), n);
    }
    if (nc) {
        if (cell_score) HIP_TRY(copy_d2h(cell_score, src_score, nc * 4));
        if (cell_wscore) HIP_TRY(copy_d2h(cell_wscore, src_wscore, nc * 4));
    }
    return NRA_OK;
}

int nra_joint_2d(int device, const nra_joint_region_t* region, int32_t n_reads, const char* seqs,
                 const int64_t* seq_off, int8_t* read_strand, int64_t n_cells, const int32_t* cell_read,
                 const int32_t* cell_k1, const int32_t* cell_k2, const nra_scoring_t* sc, int32_t flags,
                 int32_t* cell_score, int32_t* cell_wscore, int32_t* best_wscore, int64_t* sum_k1,
                 int64_t* sum_k2, int32_t* n_ties, uint8_t* status)
{
    if (n_reads > 0 && (!best_wscore || !sum_k1 || !sum_k2 || !n_ties || !status)) return fail(NRA_E_ARG, "NULL output array");
    nra_batch_t* b = nullptr;
    int rc = nra_batch2d_create(device, region, n_reads, seqs, seq_off, read_strand, n_cells, cell_read,
                                cell_k1, cell_k2, sc, flags, &b);
    if (rc) return rc;
    rc = nra_batch_run(b);
    if (!rc) rc = nra_batch_sync(b);
    if (!rc) rc = nra_batch2d_fetch(b, read_strand, cell_score, cell_wscore, best_wscore, sum_k1, sum_k2, n_ties, status);
    nra_batch_destroy(b);
    return rc;
}

// ---- generic pairs --------------------------------------------------------------------
namespace {

struct PairSetup {
    std::vector<uint8_t> pool;
    std::vector<NraDevRegion> dregs;
    std::vector<NraDevRead> dreads;
    std::vector<uint32_t> q2bit, nmask;
    std::vector<int32_t> as_query, as_target;   // sequence -> read / region index
    bool has_n = false;
};

int prepare_pairs(int device, int32_t n_seqs, const char* seqs, const int64_t* seq_off, int64_t n_pairs,
                  const int32_t* pair_query, const int32_t* pair_target, const nra_scoring_t* sc, PairSetup& ps,
                  int64_t max_query, int64_t max_target, int64_t max_query_score)
{
    if (n_seqs < 0 || n_pairs < 0) return fail(NRA_E_ARG, "negative count");
    if (n_seqs > 0 && (!seqs || !seq_off)) return fail(NRA_E_ARG, "NULL sequence array");
    if (n_pairs > 0 && (!pair_query || !pair_target)) return fail(NRA_E_ARG, "NULL pair array");
    if (n_pairs > 0x7ff00000ll) return fail(NRA_E_RANGE, "too many pairs");
    if (!scoring_ok(sc)) return fail(NRA_E_ARG, "scoring parameters out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NRA_E_DEVICE, "no HIP device: nanorepeat_amd has no CPU path");
    if (device < 0 || device >= ndev) return fail(NRA_E_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    // which sequences are queries (2-bit pool, one register block) / targets (byte pool, <= 65000)
    ps.as_query.assign((size_t)n_seqs, -1); ps.as_target.assign((size_t)n_seqs, -1);
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int32_t q = pair_query[i], t = pair_target[i];
        if (q < 0 || q >= n_seqs || t < 0 || t >= n_seqs) return fail(NRA_E_ARG, "pair index out of range");
        ps.as_query[q] = 0; ps.as_target[t] = 0;
    }
    uint64_t base = 0;
    for (int32_t s = 0; s < n_seqs; ++s) {
        const int64_t len = seq_off[s + 1] - seq_off[s];
        if (len < 0) return fail(NRA_E_ARG, "seq_off must be non-decreasing");
        if (ps.as_target[s] == 0) {
            if (len > max_target) return fail(NRA_E_RANGE, "target longer than " + std::to_string(max_target));
            NraDevRegion d{};
            d.p1_off = pool_append(ps.pool, seqs + seq_off[s], (int32_t)len, nullptr, 0, 0, ps.has_n);
            d.p2_off = d.p3_off = d.pr_off = (uint32_t)ps.pool.size();
            d.l1 = (int32_t)len; d.m1 = 0; d.l2 = 0; d.m2 = 0; d.l3 = 0;
            ps.as_target[s] = (int32_t)ps.dregs.size();
            ps.dregs.push_back(d);
            if (ps.pool.size() > 0xfff00000ull) return fail(NRA_E_RANGE, "target pool exceeds 4 GB");
        }
        if (ps.as_query[s] == 0) {
            if (len > max_query) return fail(NRA_E_RANGE, "query longer than " + std::to_string(max_query));
            if (max_score(sc, len) > max_query_score)
                return fail(NRA_E_RANGE, "query " + std::to_string(s) + ": match score x length does not fit 16 bits");
            NraDevRead r{};
            r.qoff = (uint32_t)base; r.qlen = (int32_t)len; r.region = 0; r.rc = 0;
            ps.as_query[s] = (int32_t)ps.dreads.size();
            ps.dreads.push_back(r);
            base += ((uint64_t)len + 31) / 32 * 32;
            if (base > 0xfff00000ull) return fail(NRA_E_RANGE, "query pool exceeds 4G bases");
        }
    }
    ps.pool.push_back(0);
    ps.q2bit.assign((size_t)(base / 16) + 1, 0);
    ps.nmask.assign((size_t)(base / 32) + 1, 0);
    for (int32_t s = 0; s < n_seqs; ++s) {
        if (ps.as_query[s] < 0) continue;
        const NraDevRead& r = ps.dreads[ps.as_query[s]];
        const char* p = seqs + seq_off[s];
        for (int32_t i = 0; i < r.qlen; ++i) {
            uint8_t c = encode_base(p[i]);
            const uint32_t b = r.qoff + (uint32_t)i;
            if (c >= 4) { ps.nmask[b >> 5] |= 1u << (b & 31); ps.has_n = true; c = 0; }
            ps.q2bit[b >> 4] |= (uint32_t)c << ((b & 15) * 2);
        }
    }
    return NRA_OK;
}

}  // namespace

int nra_align_pairs(int device, int32_t n_seqs, const char* seqs, const int64_t* seq_off, int64_t n_pairs,
                    const int32_t* pair_query, const int32_t* pair_target, const nra_scoring_t* sc,
                    int32_t flags, int32_t* score, int32_t* tstart, int32_t* tend)
{
    (void)flags;
    if (n_pairs > 0 && (!score || !tstart || !tend)) return fail(NRA_E_ARG, "NULL output array");
    PairSetup ps;
    int rc = prepare_pairs(device, n_seqs, seqs, seq_off, n_pairs, pair_query, pair_target, sc, ps, NRA_MAX_QLEN,
                           NRA_MAX_TLEN_WIDE, (int64_t)1 << 30);
    if (rc || n_pairs == 0) return rc;
    // Launch groups: int32 cells (score < 32768, target <= 65000 columns) in the rows-per-lane bucket of the
    // query, or -- queries longer than one register block -- chained row blocks; everything beyond that
    // (a long core's score, a whole long read as the target) in int64 cells, three row counts.
    struct Group { int R; bool chain, wide; std::vector<NraTask> tasks; };
    std::vector<Group> groups;
    auto group_of = [&](int R, bool chain, bool wide) -> Group& {
        for (Group& g : groups) if (g.R == R && g.chain == chain && g.wide == wide) return g;
        groups.push_back(Group{R, chain, wide, {}});
        return groups.back();
    };
    int chain_cols = 0;
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int32_t qi = ps.as_query[pair_query[i]];
        const int32_t ti = ps.as_target[pair_target[i]];
        const int qlen = ps.dreads[qi].qlen, tlen = ps.dregs[ti].l1;
        if (qlen == 0) continue;
        const bool wide = tlen > NRA_MAX_TLEN || max_score(sc, qlen) > kScoreCapI32;
        const bool chain = qlen > NRA_MAX_QLEN_1BLOCK;
        if (chain) chain_cols = std::max(chain_cols, tlen);
        const int R = chain ? NRA_CHAIN_R
                            : wide ? (qlen <= 64 * NRA_WIDE_R_SMALL ? NRA_WIDE_R_SMALL : NRA_WIDE_R_LARGE)
                                   : kRList[rows_for_qlen(qlen)];
        group_of(R, chain, wide).tasks.push_back(NraTask{qi, ti, -1, (int32_t)i});
    }
    std::sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) {      // longest first
        return (int64_t)a.R * (a.chain ? 1000 : 1) > (int64_t)b.R * (b.chain ? 1000 : 1);
    });
    const int chain_cap = (chain_cols + 127) / 64 * 64 + 64;
    std::vector<NraTask> tasks;
    std::vector<int32_t> counts;
    std::vector<size_t> offs;
    bool any_chain = false;
    for (const Group& g : groups) {
        offs.push_back(tasks.size());
        counts.push_back((int32_t)g.tasks.size());
        tasks.insert(tasks.end(), g.tasks.begin(), g.tasks.end());
        any_chain |= g.chain;
    }
    Arena arena;                       // before the buffers: released after them
    arena.device = device;
    ArenaScope arena_scope(&arena);
    arena.expect(ps.pool.size() + ps.q2bit.size() * 6 + (size_t)n_pairs * 32 + (2u << 20));
    DevBuf<uint8_t> d_pool; DevBuf<uint32_t> d_q2, d_nm; DevBuf<NraDevRegion> d_regs; DevBuf<NraDevRead> d_reads;
    DevBuf<NraTask> d_tasks; DevBuf<int32_t> d_counts, d_score, d_ts, d_te; DevBuf<int64_t> d_chain;
    HIP_TRY(d_pool.upload(ps.pool)); HIP_TRY(d_q2.upload(ps.q2bit)); HIP_TRY(d_nm.upload(ps.nmask));
    HIP_TRY(d_regs.upload(ps.dregs)); HIP_TRY(d_reads.upload(ps.dreads)); HIP_TRY(d_tasks.upload(tasks));
    HIP_TRY(d_counts.upload(counts));
    HIP_TRY(d_score.alloc((size_t)n_pairs)); HIP_TRY(d_ts.alloc((size_t)n_pairs)); HIP_TRY(d_te.alloc((size_t)n_pairs));
    size_t chain_tasks = 0;
    for (size_t i = 0; i < groups.size(); ++i) if (groups[i].chain) chain_tasks = std::max(chain_tasks, groups[i].tasks.size());
    const int pair_strips = chain_strips(chain_tasks, NRA_CHAIN_STRIPS, (size_t)6 * 8 * (size_t)chain_cap);
    if (any_chain) HIP_TRY(d_chain.alloc((size_t)pair_strips * 6 * (size_t)chain_cap));
    HIP_TRY(hipMemset(d_score.p, 0xff, (size_t)n_pairs * 4));
    HIP_TRY(hipMemset(d_ts.p, 0xff, (size_t)n_pairs * 4));
    HIP_TRY(hipMemset(d_te.p, 0xff, (size_t)n_pairs * 4));
    const NraScoreParams sp = to_params(*sc);
    for (size_t i = 0; i < groups.size(); ++i) {
        const Group& g = groups[i];
        LAUNCH_TRY(nra_launch_payload_origin(g.R, ps.has_n ? 1 : 0, nullptr,
                                             std::min(counts[i], g.chain ? pair_strips : 256 * 16),
                                             d_tasks.p + offs[i], d_counts.p + i, d_reads.p, d_regs.p,
                                             d_pool.p, d_q2.p, d_nm.p, sp, d_score.p, d_ts.p, d_te.p,
                                             g.chain ? d_chain.p : nullptr, g.chain ? chain_cap : 0, g.wide ? 1 : 0));
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(copy_d2h(score, d_score.p, (size_t)n_pairs * 4));
    HIP_TRY(copy_d2h(tstart, d_ts.p, (size_t)n_pairs * 4));
    HIP_TRY(copy_d2h(tend, d_te.p, (size_t)n_pairs * 4));
    return NRA_OK;
}

int nra_align_pairs_cigar(int device, int32_t n_seqs, const char* seqs, const int64_t* seq_off, int64_t n_pairs,
                          const int32_t* pair_query, const int32_t* pair_target, const nra_scoring_t* sc,
                          int32_t flags, int32_t* score, int32_t* tstart, int32_t* tend, int32_t* qstart,
                          int32_t* qend, char* cigar, int64_t cigar_cap, int64_t* cigar_off)
{
    (void)flags;
    if (n_pairs > 0 && (!score || !tstart || !tend || !qstart || !qend || !cigar || !cigar_off || cigar_cap < 1))
        return fail(NRA_E_ARG, "NULL output array");
    PairSetup ps;
    int rc = prepare_pairs(device, n_seqs, seqs, seq_off, n_pairs, pair_query, pair_target, sc, ps,
                           NRA_MAX_QLEN_1BLOCK, NRA_MAX_TLEN, kScoreCapI32);
    if (rc) return rc;
    if (cigar_off) cigar_off[0] = 0;
    if (n_pairs == 0) return NRA_OK;

    // one launch per rows-per-lane bucket; tasks keep their pair index through `order`
    std::vector<std::vector<int64_t>> by_bucket((size_t)kNumR);
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int32_t qi = ps.as_query[pair_query[i]];
        if (ps.dreads[qi].qlen == 0 || ps.dregs[ps.as_target[pair_target[i]]].l1 == 0) continue;
        by_bucket[rows_for_qlen(ps.dreads[qi].qlen)].push_back(i);
    }
    std::vector<NraTraceTask> tasks;
    std::vector<int64_t> order;
    std::vector<std::pair<int, size_t>> launches;
    std::vector<int> counts;
    uint64_t trace_bytes = 0, ops_bytes = 0;
    for (int bi = kNumR - 1; bi >= 0; --bi) {
        if (by_bucket[bi].empty()) continue;
        launches.push_back({kRList[bi], tasks.size()});
        counts.push_back((int)by_bucket[bi].size());
        for (int64_t i : by_bucket[bi]) {
            NraTraceTask t{};
            t.read = ps.as_query[pair_query[i]];
            t.region = ps.as_target[pair_target[i]];
            const uint64_t ql = (uint64_t)ps.dreads[t.read].qlen, tl = (uint64_t)ps.dregs[t.region].l1;
            t.ops_cap = (int32_t)(ql + tl);
            t.trace_off = trace_bytes; t.ops_off = ops_bytes;
            trace_bytes += ql * tl; ops_bytes += ql + tl;
            tasks.push_back(t); order.push_back(i);
        }
    }
    if (trace_bytes > (8ull << 30)) return fail(NRA_E_RANGE, "trace needs more than 8 GiB: split the call");
    const size_t nt = tasks.size();
    Arena arena;                       // before the buffers: released after them
    arena.device = device;
    ArenaScope arena_scope(&arena);
    arena.expect(ps.pool.size() + ps.q2bit.size() * 6 + nt * 128 + (size_t)ops_bytes + (2u << 20));
    DevBuf<uint8_t> d_pool, d_trace, d_ops; DevBuf<uint32_t> d_q2, d_nm; DevBuf<NraDevRegion> d_regs;
    DevBuf<NraDevRead> d_reads; DevBuf<NraTraceTask> d_tasks; DevBuf<int32_t> d_fill, d_back;
    HIP_TRY(d_pool.upload(ps.pool)); HIP_TRY(d_q2.upload(ps.q2bit)); HIP_TRY(d_nm.upload(ps.nmask));
    HIP_TRY(d_regs.upload(ps.dregs)); HIP_TRY(d_reads.upload(ps.dreads)); HIP_TRY(d_tasks.upload(tasks));
    HIP_TRY(d_trace.alloc((size_t)trace_bytes)); HIP_TRY(d_ops.alloc((size_t)ops_bytes));
    HIP_TRY(d_fill.alloc(nt * 5)); HIP_TRY(d_back.alloc(nt * 3));
    const NraScoreParams sp = to_params(*sc);
    for (size_t i = 0; i < launches.size(); ++i) {
        const size_t off = launches[i].second;
        LAUNCH_TRY(nra_launch_trace_fill(launches[i].first, ps.has_n ? 1 : 0, nullptr, counts[i], d_tasks.p + off,
                                         d_reads.p, d_regs.p, d_pool.p, d_q2.p, d_nm.p, sp, d_trace.p, d_fill.p + off * 5));
        LAUNCH_TRY(nra_launch_trace_back(nullptr, counts[i], d_tasks.p + off, d_reads.p, d_regs.p, d_trace.p,
                                         d_fill.p + off * 5, d_ops.p, d_back.p + off * 3));
    }
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int32_t> fill(nt * 5), back(nt * 3);
    std::vector<uint8_t> ops((size_t)ops_bytes);
    if (nt) {
        HIP_TRY(copy_d2h(fill.data(), d_fill.p, nt * 5 * 4));
        HIP_TRY(copy_d2h(back.data(), d_back.p, nt * 3 * 4));
        HIP_TRY(copy_d2h(ops.data(), d_ops.p, (size_t)ops_bytes));
    }
    // per pair: extents + run-length encoded CIGAR
    std::vector<std::string> cg((size_t)n_pairs);
    for (int64_t i = 0; i < n_pairs; ++i) { score[i] = -1; tstart[i] = -1; tend[i] = -1; qstart[i] = -1; qend[i] = -1; }
    for (size_t t = 0; t < nt; ++t) {
        const int64_t i = order[t];
        const int32_t* f = &fill[t * 5];
        if (f[0] < 0) continue;
        score[i] = f[0]; tstart[i] = back[t * 3 + 2]; tend[i] = f[2]; qstart[i] = back[t * 3 + 1]; qend[i] = f[3] + 1;
        const int n = back[t * 3];
        const uint8_t* op = ops.data() + tasks[t].ops_off + tasks[t].ops_cap - n;
        std::string& out = cg[i];
        for (int k = 0; k < n;) {
            int e = k;
            while (e < n && op[e] == op[k]) ++e;
            out += std::to_string(e - k);
            out += (char)op[k];
            k = e;
        }
    }
    int64_t pos = 0;
    for (int64_t i = 0; i < n_pairs; ++i) {
        cigar_off[i] = pos;
        if (pos + (int64_t)cg[i].size() + 1 > cigar_cap)
            return fail(NRA_E_ARG, "cigar buffer too small");
        memcpy(cigar + pos, cg[i].c_str(), cg[i].size() + 1);
        pos += (int64_t)cg[i].size() + 1;
    }
    cigar_off[n_pairs] = pos;
    return NRA_OK;
}

// ---- common -------------------------------------------------------------------------
static int account_run(nra_batch* b);
namespace { int finish_refine(nra_batch* b); }

int nra_batch_run(nra_batch_t* b)
{
    if (!b) return fail(NRA_E_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    // the events are about to be re-recorded: a finished run nobody synchronised on is accounted first
    if (b->ran && !b->accounted) {
        HIP_TRY(hipStreamSynchronize(b->stream));
        int rc0 = account_run(b);
        if (rc0) return rc0;
    }
    if (b->kind == 2 && !b->have_cells) return fail(NRA_E_ARG, "2D batch without a cell list: call nra_batch2d_set_cells");
    int rc = b->kind == 1 ? run_1d(b) : run_2d(b);
    if (!rc) { b->ran = true; b->accounted = false; }
    return rc;
}

// Reads the HIP events of the run that has just completed (the stream is idle) into the batch's
// statistics: the last run's timings and their sums over all runs, so that a caller timing many runs
// gets averages that belong to the same runs as its wall clock.
static int account_run(nra_batch* b)
{
    if (!b->ran || b->accounted) return NRA_OK;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
    b->stats.total_ms = ms;
    HIP_TRY(hipEventElapsedTime(&ms, b->phase_ev[0], b->phase_ev[1]));
    b->stats.score_phase_ms = ms;
    // event pairs were recorded in launch order: 1D: score..., extents...; 2D: probe..., window...
    double first = 0, second = 0;
    const int n_first = b->kind == 1 ? b->n_score_ev : b->n_ext_ev;
    const int n_second = b->kind == 1 ? b->n_ext_ev : b->n_score_ev;
    int ev = 2;
    for (int i = 0; i < n_first; ++i, ev += 2) { HIP_TRY(hipEventElapsedTime(&ms, b->ev[ev], b->ev[ev + 1])); first += ms; }
    for (int i = 0; i < n_second; ++i, ev += 2) { HIP_TRY(hipEventElapsedTime(&ms, b->ev[ev], b->ev[ev + 1])); second += ms; }
    b->stats.score_kernel_ms = b->kind == 1 ? first : second;
    b->stats.extent_kernel_ms = b->kind == 1 ? second : first;
    b->stats.n_score_launches = b->n_score_ev;
    b->stats.n_runs += 1;
    b->stats.sum_score_kernel_ms += b->stats.score_kernel_ms;
    b->stats.sum_extent_kernel_ms += b->stats.extent_kernel_ms;
    b->stats.sum_total_ms += b->stats.total_ms;
    b->stats.sum_score_phase_ms += b->stats.score_phase_ms;
    b->accounted = true;
    return b->kind == 2 ? finish_refine(b) : NRA_OK;
}

int nra_batch_sync(nra_batch_t* b)
{
    if (!b) return fail(NRA_E_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    const int rc = account_run(b);
    if (rc || b->kind != 1) return rc;
    return check_mt(b);
}

int nra_batch_stats(nra_batch_t* b, nra_stats_t* st)
{
    if (!b || !st) return fail(NRA_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(b->device));
    if (b->ran) {
        HIP_TRY(hipStreamSynchronize(b->stream));
        int rc = account_run(b);
        if (rc) return rc;
        if (b->kind == 1 && !(b->flags & NRA_F_ALL_EXTENTS) && !b->buckets.empty()) {
            std::vector<int32_t> tc(b->buckets.size());
            HIP_TRY(hipMemcpy(tc.data(), b->tie_count.p, tc.size() * 4, hipMemcpyDeviceToHost));
            int64_t n = 0;
            for (int32_t v : tc) n += v;
            b->stats.n_extent_tasks = n;
        }
    }
    *st = b->stats;
    return NRA_OK;
}

void nra_batch_destroy(nra_batch_t* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    delete b;
}

}  // extern "C"
