// orcgpu.hip -- host side of liborcgpu.so: context, staging, per-stripe planning and kernel
// launches behind the C ABI of include/orcgpu.h.
//
// The planner restates, for flat columns, what array_decoder_factory wires up per ORC type
// (src/array_decoder/mod.rs:390-511; stream kinds per type: SURVEY.md Appendix C) and what
// NaiveStripeDecoder iterates batch by batch (:513-564) -- except that a whole stripe (or
// several) is decoded by a fixed sequence of kernel launches on one HIP stream:
//
//   H2D scalars/jobs -> [block decompress] -> RLE block walk (4 rounds + verify + repair)
//   -> tile/job scans -> expand PRESENT -> validity/rank -> expand data streams -> finishers
//   -> one D2H of the small summary (errors, null counts, string totals).
//
// There is NO CPU decode path in this library: every entry point that needs the GPU fails with
// ORCGPU_HIP_ERROR when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orcgpu.h"
#include "device/rle_kernels.h"
#include "device/rle_parse.h"
#include "device/rle_scan.hip"
#include "device/rle_expand.hip"
#include "device/multi_job.h"
#include "device/column_kernels.hip"
#include "device/string_kernels.hip"
#include "device/decompress_kernels.hip"
#include "device/select_kernels.hip"
#include "device/rle_encode.hip"

// Arrow C Data Interface structs (public, stable ABI)
extern "C" {
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
}

namespace {

constexpr uint64_t kAlign = 256;
constexpr int kWalkRounds = 4;
inline uint64_t align_up(uint64_t v, uint64_t a = kAlign) { return (v + a - 1) / a * a; }

struct DevBuf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + (1u << 20);
    if (hipMalloc((void**)&p, want) != hipSuccess) {
      if (hipMalloc((void**)&p, n) != hipSuccess) return false;
      want = n;
    }
    cap = want;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Bump {
  uint64_t off = 0;
  uint64_t take(uint64_t n, uint64_t a = kAlign) {
    off = align_up(off, a);
    uint64_t r = off;
    off += n;
    return r;
  }
};

struct ChunkInfo {
  uint64_t src_off;  // offset of the chunk payload inside the stream
  uint32_t len;
  uint32_t original;
  uint32_t plain_cap;  // bytes the chunk can expand to (exact for original and Snappy chunks, and for Zstandard frames that state their size)
  int32_t zparse = -1; // Zstandard: index into StagedStream::zchunks
};


}  // namespace

#include "orcgpu_zstd_host.inc"
#include "orcgpu_tz.inc"

namespace {
struct StagedStream {
  uint32_t column_id;
  int32_t kind;
  uint64_t off;  // offset inside the staged arena
  uint64_t len;
  std::vector<ChunkInfo> chunks;  // only for compressed stripes
  std::vector<ZChunkParse> zchunks;  // Zstandard: frame / block headers of every compressed chunk
  uint32_t skip_bytes = 0, skip_values = 0, skip_bits = 0;  // entry point (orcgpu_stream): where in the plain bytes the decoder starts, values it drops, bits of a bit stream's first byte that come before
  std::vector<std::pair<uint32_t, uint32_t>> hints;  // verified run starts (orcgpu_stream::entries): (chunk index or ~0, byte in its plain bytes), the stream's start first
  bool framing_error = false;     // truncated chunk header / payload (compression.rs:253-261 panics)
  uint64_t framed_len = 0;        // bytes covered by well-formed chunks
};
}  // namespace

constexpr int kMaxLanes = 4;
// Zstandard: sequences of a call from which on the FSE chains go one lane per block (zstd_lanes.h) instead of one wavefront per
// block: below it the call lasts as long as its longest chain either way (SF 3: 32 against 22 ms with the lanes forced on)
constexpr uint64_t kZstdLanesMinSequences = 40000000ull;
constexpr size_t kStagePiece = 16u << 20;  // bytes per pinned staging piece
constexpr int kCopyThreads = 6;

// A handful of host threads that copy slices of caller memory into the pinned staging piece (one thread's memcpy does
// about 10 GB/s: less than the link takes).
struct CopyPool {
  struct Task {
    uint8_t* dst;
    const uint8_t* src;
    size_t n;
  };
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv, done;
  std::vector<Task> tasks;
  size_t next = 0, pending = 0;
  bool stop = false;
  CopyPool() {
    // ORCGPU_STAGE_THREADS: helper threads of a context's staging copies (kCopyThreads by default; 0: the caller alone) -- eight
    // processes on one host, one per GPU, share its cores: bench.py gives every rank its share
    int n = kCopyThreads;
    if (const char* e = getenv("ORCGPU_STAGE_THREADS")) n = std::max(0, std::min(64, atoi(e)));
    for (int i = 0; i < n; i++) threads.emplace_back([this] { run(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> g(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : threads) t.join();
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [this] { return stop || next < tasks.size(); });
        if (stop) return;
        t = tasks[next++];
      }
      if (t.src) memcpy(t.dst, t.src, t.n);
      else memset(t.dst, 0, t.n);
      {
        std::lock_guard<std::mutex> g(m);
        if (--pending == 0) done.notify_all();
      }
    }
  }
  // copies (src == nullptr: zero fill) the calling thread takes part in; returns when all are done
  void copy_all(std::vector<Task>& ts) {
    size_t total = 0;
    for (auto& t : ts) total += t.n;
    if (total < (1u << 20)) {  // not worth waking anybody
      for (auto& t : ts) {
        if (t.src) memcpy(t.dst, t.src, t.n);
        else memset(t.dst, 0, t.n);
      }
      return;
    }
    // slices of at most 1 MiB so that the threads share large streams
    std::vector<Task> sl;
    for (auto& t : ts)
      for (size_t o = 0; o < t.n; o += 1u << 20) sl.push_back(Task{t.dst + o, t.src ? t.src + o : nullptr, std::min<size_t>(1u << 20, t.n - o)});
    {
      std::lock_guard<std::mutex> g(m);
      tasks.swap(sl);
      next = 0;
      pending = tasks.size();
    }
    cv.notify_all();
    for (;;) {  // the caller works too
      Task t;
      {
        std::lock_guard<std::mutex> g(m);
        if (next >= tasks.size()) break;
        t = tasks[next++];
      }
      if (t.src) memcpy(t.dst, t.src, t.n);
      else memset(t.dst, 0, t.n);
      std::lock_guard<std::mutex> g(m);
      if (--pending == 0) done.notify_all();
    }
    std::unique_lock<std::mutex> g(m);
    done.wait(g, [this] { return pending == 0; });
    tasks.clear();
    next = 0;
  }
};

// A context decodes the columns of a call on up to kMaxLanes LANES at once: a lane is a HIP stream with its own workspace,
// pinned summary buffer and events (lane 0 = the context itself, the others are child contexts driven by their own host
// thread for the duration of a call).  Columns are independent (stripe.rs:154-165), so lanes share nothing but the
// staged stripes (read only) and write disjoint columns of the results.
struct orcgpu_ctx {
  int device = 0;
  int lane_id = 0;
  orcgpu_ctx* lanes[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};  // [0] = this
  hipStream_t stream = nullptr;
  std::string err;
  DevBuf scratch;
  DevBuf enc_a, enc_b, enc_in, enc_tmp, enc_out[3];  // the encoder's own (orcgpu_encode.inc): chain tables, run tables, inputs brought to the device, gathered values, streams
  uint8_t* pinned = nullptr;
  size_t pinned_cap = 0;
  uint8_t* fin_pinned = nullptr;       // staging of the finishers' job table
  size_t fin_pinned_cap = 0;
  hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // start, before / after expansion, end, after decompression, after the walk, after the decompressors' first stage, behind the Zstandard table kernel, behind the sequences kernel (one lane per block), [9] in front of it (behind the wait for the other lanes' table kernels)
  uint32_t n_cus = 0;
  hipEvent_t kev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // around rle_walk_short_kernel [0] and dict_emit_kernel [1] (orcgpu_last_lane_stats)
  bool kev_used[2] = {false, false};
  hipStream_t aux_stream = nullptr;   // the Zstandard execution kernel runs here, beside the entropy kernel on `stream`
  hipEvent_t aux_ev[2] = {nullptr, nullptr};
  bool exact_on = false;  // the last decode call of this lane met a stream for the exact parallel walk (rle_exact_*): the next one sends it
  hipEvent_t lit_ev[2] = {nullptr, nullptr};  // around zstd_literals_kernel on the stream it runs on (orcgpu_last_lane_stats)
  bool lit_ev_used = false;
  float last_total_ms = 0, last_expand_ms = 0;
  float last_phase_ms[ORCGPU_N_PHASES] = {0, 0, 0, 0, 0, 0, 0};
  orcgpu_lane_stats last_stats{};     // this lane's part of the last call (orcgpu_last_lane_stats)
  uint32_t last_n_lanes = 1;          // lane 0: lanes the last call ran
  double call_t0 = 0;                 // host clock (us) when the call that drives this lane was entered
  // Zstandard at table scale, several lanes: a lane's sequences kernel takes every CU's LDS (three wavefronts of 52.5 KB); launched
  // before another lane's FSE table kernel (15 KB a wavefront) it keeps that one out until its own chains end -- the other lane then
  // starts 4 to 25 ms late (the "straggler" of round 4).  So every lane's table kernel first: a lane launches its sequences kernel
  // behind the table kernels of ALL lanes.  tables_gate: 0 = this lane has not recorded ev[7] yet, 1 = it has, 2 = it has no such kernel
  orcgpu_ctx* gate_peers[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  int n_gate_peers = 0;
int call_z_lanes = -1;  // a call split over column lanes: its Zstandard path decided for the call as a whole (1 table scale, 0 one wavefront per block), -1: every lane by its own count
    std::atomic<int> tables_gate{0};
  uint32_t last_expand_launches = 0;
  // ---- staging pipeline (lane 0 only): a copy stream, two pinned pieces filled by a few host threads while the other one
  // is on its way to HBM, a pool of stripe arenas ----
  hipStream_t copy_stream = nullptr;
  hipStream_t d2h_stream = nullptr;    // results on their way back to the host (orcgpu_result_fetch_async), beside the next decode
  hipEvent_t d2h_gate = nullptr;       // "everything enqueued on `stream` so far" for the copies on d2h_stream
  uint8_t* piece[2] = {nullptr, nullptr};
  hipEvent_t piece_ev[2] = {nullptr, nullptr};
  bool piece_used[2] = {false, false};
  int piece_next = 0;
  std::vector<std::pair<uint8_t*, size_t>> arena_pool;  // freed staged arenas (device pointer, bytes)
  // Staging (orcgpu_stage_stripe: copy stream, pinned pieces, zone tables) and decoding (everything else) may run in two
  // threads at once -- the read-ahead reader stages stripe k + 1 while stripe k is decoded; what the two share is guarded here:
  std::mutex pool_m;   // arena_pool (taken from by staging, given back to by orcgpu_staged_free)
  std::mutex err_m;    // `err`
  struct CopyPool* copiers = nullptr;
  // writer time zones seen so far (lane 0 only): host table + its copy in HBM ({at[n] i64}{offs[n] i32})
  struct Zone {
    TzTable table;
    uint8_t* dev = nullptr;
    int64_t epoch = 1420070400;
  };
  std::map<std::string, std::shared_ptr<Zone>> zones;
  // Job tables on their way to HBM go through pinned staging: hipMemcpyAsync from pageable memory makes the host wait for the
  // stream, which then runs dry while the host enqueues what follows.  Bump-allocated; taken back when a call starts (behind a
  // stream synchronisation: nothing is in flight from it any more).
  std::vector<std::pair<uint8_t*, size_t>> up_blocks;
  size_t up_block = 0, up_off = 0;
  void upload_reset() { up_block = up_off = 0; }
  hipError_t upload(void* dst, const void* src, size_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    const size_t need = (n + 63) & ~(size_t)63;
    while (up_block < up_blocks.size() && up_off + need > up_blocks[up_block].second) up_block++, up_off = 0;
    if (up_block == up_blocks.size()) {
      uint8_t* b = nullptr;
      const size_t cap = std::max<size_t>(2 * need, 1u << 20);
      hipError_t e = hipHostMalloc((void**)&b, cap, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      up_blocks.push_back({b, cap});
      up_off = 0;
    }
    uint8_t* h = up_blocks[up_block].first + up_off;
    up_off += need;
    memcpy(h, src, n);
    return hipMemcpyAsync(dst, h, n, hipMemcpyHostToDevice, st);
  }
  bool ensure_pinned(size_t n) {
    if (n <= pinned_cap) return true;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_cap = 0;
    size_t want = n + n / 4 + (1u << 16);
    if (hipHostMalloc((void**)&pinned, want, hipHostMallocDefault) != hipSuccess) return false;
    pinned_cap = want;
    return true;
  }
};

struct orcgpu_staged {
  orcgpu_ctx* ctx = nullptr;
  orcgpu_stripe_desc desc{};
  std::vector<orcgpu_column> cols;
  std::vector<StagedStream> streams;
  uint8_t* dev = nullptr;
  size_t dev_bytes = 0;     // bytes in use
  size_t dev_cap = 0;       // bytes of the arena (it may come from the pool)
  hipEvent_t ready = nullptr;  // recorded on the copy stream behind the stripe's last piece
  std::shared_ptr<orcgpu_ctx::Zone> zone;  // writer time zone of the stripe (null: none given)
  uint64_t stream_bytes = 0;
  // the file reader's row selection on these rows, as the batches it yields (rows counted from the first row staged)
  bool has_sel = false;
  std::vector<SelBatch> sel;
  bool piece = false;  // some row groups of a stripe (what a selection leaves of it), not the stripe
  const StagedStream* find(uint32_t col, int kind) const {
    for (auto& s : streams)
      if (s.column_id == col && s.kind == kind) return &s;
    return nullptr;
  }
};

namespace {

// What one column contributes to a result
struct ColumnOut {
  int32_t orc_type = 0;
  uint32_t column_id = 0;
  uint32_t width = 0;        // fixed width in bytes (0 for strings / boolean)
  bool is_string = false, is_bool = false;
  bool is_struct = false;    // validity only; its fields are the columns whose `parent` names it
  bool is_union = false;     // values = the type ids (int8 per row, 0 where the Union is null); its arms' children are the columns whose `parent` names it
  bool is_list = false;      // List / Map: per-batch int32 offsets restarting at 0 (list.rs:63-87, map.rs:74-104); char_total / char_base count
                             // ELEMENTS; the elements themselves are the columns of orcgpu_result::subs[sub]
  bool is_map = false;
  int32_t sub = -1;
  bool elem = false;         // below a List / Map: decoded as a column of that one's sub-result, nothing of it lives in this result
  int32_t parent = -1;       // index in orcgpu_result::cols of the Struct this column is a field of
  int32_t ts_unit = 3;
  uint32_t precision = 0, scale = 0;
  bool has_present = false;
  uint64_t values_off = 0, values_bytes = 0;      // in the result arena (or the chars arena)
  int lane = 0;                                   // which lane's arenas hold the column
  bool values_in_chars = false;                   // dictionary strings: values live in orcgpu_result::chars
  uint64_t offsets_off = 0;                       // strings: n_batches * (B+1) int32
  uint64_t validity_off = 0;                      // n_batches * words_per_batch u64
  std::vector<uint64_t> null_counts;              // host copy, per batch
  std::vector<uint64_t> char_base, char_total;    // strings: per batch, host copies
  // row selection (orcgpu_result_select): per selected batch
  uint64_t sel_validity_off = 0, sel_offsets_off = 0, sel_bool_off = 0;  // in orcgpu_result::sel_arena
  std::vector<uint64_t> sel_nulls, sel_char_start, sel_char_total;
  // error bookkeeping (resolved after the summary copy)
  int status = 0;
  uint32_t err_batch = 0;
};

}  // namespace

// Host mirror of a result's arenas: ONE device-to-host copy per arena into pinned memory; exported batches are views into
// it and keep it alive (reference count) after the result itself is gone.
// Pinned host memory of the result copies (orcgpu_result_fetch): pinning costs about 0.2 ms per MB, so buffers the last batch
// has let go of are kept for the next result -- of this reader or the next (a scan over many files) -- instead of being
// unpinned: process-wide, at most ORCGPU_PINNED_POOL_MB (default 4096) idle.
struct PinnedPool {
  std::mutex m;
  std::vector<std::pair<uint8_t*, size_t>> free_list;
  size_t idle = 0;
  static PinnedPool& get() {
    static PinnedPool* p = new PinnedPool();  // (never destroyed: batches may outlive every context)
    return *p;
  }
  static size_t limit() {
    static const size_t v = (size_t)(getenv("ORCGPU_PINNED_POOL_MB") ? atoll(getenv("ORCGPU_PINNED_POOL_MB")) : 4096) << 20;
    return v;
  }
  bool take(size_t want, uint8_t*& ptr, size_t& cap) {
    {
      std::lock_guard<std::mutex> g(m);
      int best = -1;
      for (size_t k = 0; k < free_list.size(); k++)
        if (free_list[k].second >= want && free_list[k].second <= 2 * want + (1u << 20) && (best < 0 || free_list[k].second < free_list[best].second)) best = (int)k;
      if (best >= 0) {
        ptr = free_list[best].first;
        cap = free_list[best].second;
        idle -= cap;
        free_list.erase(free_list.begin() + best);
        return true;
      }
    }
    if (hipHostMalloc((void**)&ptr, want, hipHostMallocDefault) != hipSuccess) {
      // make room: unpin what is idle, then once more
      std::vector<std::pair<uint8_t*, size_t>> drop;
      {
        std::lock_guard<std::mutex> g(m);
        drop.swap(free_list);
        idle = 0;
      }
      for (auto& d : drop) (void)hipHostFree(d.first);
      if (hipHostMalloc((void**)&ptr, want, hipHostMallocDefault) != hipSuccess) return false;
    }
    cap = want;
    return true;
  }
  void give(uint8_t* ptr, size_t cap) {
    if (!ptr) return;
    {
      std::lock_guard<std::mutex> g(m);
      if (idle + cap <= limit() && free_list.size() < 256) {
        free_list.push_back({ptr, cap});
        idle += cap;
        return;
      }
    }
    (void)hipHostFree(ptr);
  }
};

struct HostMirror {
  std::atomic<int> refs{1};
  uint8_t* arena[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  uint8_t* chars[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  size_t arena_cap[kMaxLanes] = {0, 0, 0, 0}, chars_cap[kMaxLanes] = {0, 0, 0, 0};
  uint8_t* sel = nullptr;  // per-batch buffers of a row selection
  size_t sel_cap = 0;
  hipEvent_t done = nullptr;  // recorded behind the copies of orcgpu_result_fetch_async
  bool pending = false;       // ... and not waited for yet
  void wait() {
    if (pending && done) (void)hipEventSynchronize(done);
    pending = false;
  }
  void unref() {
    if (refs.fetch_sub(1) == 1) {
      wait();
      if (done) (void)hipEventDestroy(done);
      PinnedPool::get().give(sel, sel_cap);
      for (int l = 0; l < kMaxLanes; l++) {
        PinnedPool::get().give(arena[l], arena_cap[l]);
        PinnedPool::get().give(chars[l], chars_cap[l]);
      }
      delete this;
    }
  }
};

struct orcgpu_result {
  orcgpu_ctx* ctx = nullptr;
  uint64_t n_rows = 0;
  uint32_t batch = 8192, n_batches = 0, words_per_batch = 0;
  DevBuf arena[kMaxLanes];  // per lane: the Arrow buffers of the columns that lane decoded
  DevBuf chars[kMaxLanes];  // dictionary -> Utf8 materialised value bytes (sized after the lengths are known)
  size_t arena_used[kMaxLanes] = {0, 0, 0, 0}, chars_used[kMaxLanes] = {0, 0, 0, 0};  // bytes of the last decode
  HostMirror* mirror = nullptr;  // filled by orcgpu_result_fetch
  bool mirror_valid = false;
  // row selection: once orcgpu_result_select has run, batch b is rows [sel[b].start, sel[b].start + sel[b].len)
  bool selected = false;
  std::vector<SelBatch> sel;
  DevBuf sel_arena;
  size_t sel_used = 0;
  uint32_t full_batches = 0;     // batches of the underlying uniform decode
  int full_status = 0;           // status of the underlying decode (before the selection re-mapped the failing batch)
  uint32_t full_err_batch = 0, full_err_col = 0;
  std::vector<ColumnOut> cols;
  std::vector<std::string> field_names;  // per column: the name of a Struct's field (set by the file reader; empty: "f<position>")
  // Elements of the List / Map columns: one result each over a "stripe" whose rows are the column's elements (all of the
  // stripe's, ONE batch), whose root columns are the element column(s) -- a Map's key and value -- with everything below them
  std::vector<orcgpu_result*> subs;
  std::vector<uint32_t> src_col;         // sub-results: per column, its index in the columns of the staged stripe
  int status = 0;
  uint32_t err_batch = 0, err_col = 0;
  uint64_t arrow_bytes = 0;
};

namespace {

void set_err(orcgpu_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) {
    std::lock_guard<std::mutex> g(c->err_m);
    c->err = buf;
  }
}

#define HIP_TRY(ctx, expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      set_err(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return ORCGPU_HIP_ERROR;                                                          \
    }                                                                                   \
  } while (0)

// Chunk framing scan on the host while the bytes are at hand (compression.rs:113-123, :244-267): one ChunkInfo per
// 3-byte header; Snappy blocks and Zstandard frames also say how large their output is.
void scan_chunks(const uint8_t* ptr, uint64_t len_total, int compression, uint64_t block_size, StagedStream& st) {
  uint64_t p = 0;
  while (p < len_total) {
    if (p + 3 > len_total) {
      st.framing_error = true;
      break;
    }
    uint32_t h = (uint32_t)ptr[p] | ((uint32_t)ptr[p + 1] << 8) | ((uint32_t)ptr[p + 2] << 16);
    uint32_t len = h >> 1;
    if (p + 3 + len > len_total) {
      st.framing_error = true;
      break;
    }
    uint32_t cap = (uint32_t)block_size;
    if (h & 1) {
      cap = len;
    } else if (compression == ORCGPU_COMP_SNAPPY) {
      // Snappy blocks start with their uncompressed length (snap::raw::decompress_len, compression.rs:163-164)
      uint64_t u = 0;
      int shift = 0;
      for (uint32_t k = 0; k < len && k < 5; k++) {
        uint8_t c = ptr[p + 3 + k];
        u |= (uint64_t)(c & 0x7f) << shift;
        shift += 7;
        if (!(c & 0x80)) break;
      }
      // (a block that claims more than the compression block size -- no writer emits one -- gets the block size: the
      // decoder then rejects it as it would a block whose output overruns its slot)
      cap = (uint32_t)std::min<uint64_t>(u, std::max<uint64_t>(block_size, 1u << 22));
    }
    ChunkInfo ci{p + 3, len, h & 1, cap, -1};
    if (!(h & 1) && compression == ORCGPU_COMP_ZSTD) {
      // frame and block headers (RFC 8878 3.1.1): what lets the device decode every block on its own
      ZChunkParse zp;
      zstd_parse_chunk(ptr + p + 3, len, zp);
      // the zstd crate grows its output as needed; the oracle gives it max(block size, 4 MiB)
      const uint64_t limit = std::max<uint64_t>(block_size, 1u << 22);
      const uint64_t bound = zp.size_known ? zp.plain_size : limit;
      uint64_t need_seq = 0, need_lit = 0;
      for (auto& it : zp.items) {
        need_seq += it.kind == 2 ? 3ull * it.nseq : 0;  // a sequence yields at least 3 bytes
        need_lit += it.kind == 2 ? it.lit_regen : 0;    // every literal is output
      }
      if (!zp.bad && (bound > limit || need_seq > bound || need_lit > bound)) zp.bad = true;
      if (zp.bad) {
        zp.items.clear();
        ci.plain_cap = 0;
      } else if (zp.size_known) {
        ci.plain_cap = (uint32_t)zp.plain_size;
      }
      ci.zparse = (int32_t)st.zchunks.size();
      st.zchunks.push_back(std::move(zp));
    }
    st.chunks.push_back(ci);
    p += 3 + (uint64_t)len;
  }
  st.framed_len = p;
}

uint32_t type_width(int t) {
  switch (t) {
    case ORCGPU_T_BYTE: return 1;
    case ORCGPU_T_SHORT: return 2;
    case ORCGPU_T_INT:
    case ORCGPU_T_DATE:
    case ORCGPU_T_FLOAT: return 4;
    case ORCGPU_T_LONG:
    case ORCGPU_T_DOUBLE:
    case ORCGPU_T_TIMESTAMP:
    case ORCGPU_T_TIMESTAMP_INSTANT: return 8;
    case ORCGPU_T_DECIMAL: return 16;
    default: return 0;
  }
}
bool is_string_type(int t) { return t == ORCGPU_T_STRING || t == ORCGPU_T_VARCHAR || t == ORCGPU_T_CHAR || t == ORCGPU_T_BINARY; }

// ---- per-call plan ------------------------------------------------------------------------------
// (JC_PRESENT1 + d - 1: PRESENT streams of the fields of Structs / arms of Unions at depth d: they are as long as their parent has non-null rows)
// kMaxStructDepth classes of them: the reference recurses without a limit (array_decoder/mod.rs:464-505); here the depth is bounded
// by a number no schema reaches (the type tree of a file is walked recursively on the host: kMaxTypeDepth below), and the PRESENT
// phase loops over the depths a call really has
constexpr int kMaxStructDepth = 256;
constexpr int kMaxTypeDepth = 256;  // nesting of any kind (Struct / List / Map / Union) the file reader follows below a root column
enum JobClass { JC_PRESENT = 0, JC_RLE2 = 1, JC_RLE1 = 2, JC_BYTE = 3, JC_PRESENT1 = 4, JC_COUNT = JC_PRESENT1 + kMaxStructDepth };

struct JobPlan {
  int cls;
  uint8_t codec, is_signed, nbits, out_bytes;
  const uint8_t* data;      // device pointer to the plain stream (may be assigned later)
  uint64_t data_scratch_off = ~0ull;  // when the plain stream lives in scratch (decompressed)
  uint64_t len_upper;       // upper bound of the plain length
  uint32_t len_idx, needed_idx, total_idx;
  uint64_t out_scratch_off = ~0ull;   // dense output in scratch ...
  uint64_t out_result_off = ~0ull;    // ... or directly in the result arena
  int result_index = 0;
  uint64_t expect_values;   // host estimate of values (for group sizing)
  // filled while laying out
  uint32_t block0 = 0, nblocks = 0, group0 = 0, ngroups = 0, group_size = 64;
  uint32_t skip_values = 0;            // of the stream's entry point: decoded in front of the output (RleJob::skip)
  const StagedStream* hint_src = nullptr;  // verified run starts, if the stream has any
  uint32_t chunk0 = 0;
  uint32_t uniform_idx = 0, uniform_value = 0;  // RleJob::uniform_idx / uniform_value (a Decimal column's scales)
  int stripe = 0, col = 0, role = 0;  // role: stream kind the job decodes
  int final_index = -1;
};

struct PlainStream {
  const uint8_t* dev = nullptr;  // device pointer when uncompressed (inside the staged arena)
  uint64_t scratch_off = ~0ull;  // scratch offset when decompressed
  uint64_t len_upper = 0;
  uint32_t len_idx = 0;          // scalar holding the actual plain length
  uint32_t err_idx = 0;          // scalar holding a codec error flag (compressed streams)
  uint32_t skip_values = 0;      // entered at a row group: values of the first run that come before the column's (orcgpu_stream)
  uint32_t skip_bits = 0;        // ... and, of a bit stream, bits of its first byte
  const StagedStream* src = nullptr;  // (for its verified run starts)
  uint32_t chunk0 = 0;           // compressed: index of the stream's first chunk in the call's chunk table
  bool exists = false;
  bool in_result = false;        // decompressed into the column's value buffer (DecompStream::result_index): no plain copy in the workspace
};

constexpr int ORCGPU_RETRY_LARGER_SLOTS = 1000;  // internal: decode_lane asks for a second run (never leaves the library)

struct DecompStream {
  const orcgpu_staged* stripe;
  const StagedStream* st;
  uint64_t scratch_off;
  uint32_t len_idx, err_idx;
  // the DATA stream of a direct string column IS the column's Arrow value buffer: its chunks are decompressed straight into the
  // result arena (results[result_index], at result_off) instead of a slot of the workspace that a finisher then copies there
  int result_index = -1;
  uint64_t result_off = 0;
};

struct ColPlan {
  int stripe, col;
  orcgpu_column c;
  uint64_t n_rows;
  bool has_present = false;   // the column has validity: a PRESENT stream of its own, or a Struct above it has
  bool own_present = false;   // ... of its own
  int parent_plan = -1;       // index in Plan::cols of the Struct this column is a field of (only when that one has validity)
  int depth = 0;              // Structs above it
  uint32_t ceil8_idx = 0;     // Struct: scalar holding ceil(non-null rows / 8), the length of its fields' PRESENT streams
  uint32_t nonascii_idx = 0;  // direct Utf8: scalar the validation pass sets when the text holds a byte >= 0x80 (else no offset check is needed)
  uint32_t uniform_idx = 0;   // Decimal without nulls: scalar that says "every value's scale is the column's" (rle2_uniform_kernel); 0: none
  uint32_t child_bits = 0;    // ... entered at a row group in mid-byte: the bits of their first byte that belong to the rows before
  bool child_bits_set = false;
  // Union (union.rs:69-136): behind the Union's own plan follow its ARMS, one per child: plans without a column of their own
  // (col = -1) that stand where a Struct stands above a field -- valid where the Union is present and its tag names the arm
  int arm0 = -1, n_arms = 0;  // the Union: its first arm in Plan::cols
  int arm_of = -1, arm_tag = 0;  // an arm: the Union's plan, the tag
  uint64_t arm_validity_off = 0;  // an arm: scratch for the per-batch bitmaps nobody reads
  bool tz = false;            // TIMESTAMP of a stripe with a writer time zone: re-labelled to UTC, which can yield nulls
  PlainStream present, data, length, secondary, dict;
  // scratch
  uint64_t pbytes_off = 0, vbits_off = 0, wpop_off = 0, rank_off = 0, rank_tiles_off = 0;
  uint64_t n_words = 0, n_rank_tiles = 0;
  uint32_t nonnull_idx = 0;   // scalar: number of non-null rows (= needed of the data streams)
  uint32_t nullcount_off = 0; // index into the summary null-count array (per batch)
  uint32_t err_idx = 0;       // scalar: finisher error word
  int job_present = -1, job_data = -1, job_length = -1, job_secondary = -1;
  bool is_dict = false;
  uint32_t key_bytes = 4;             // dictionary keys as the expansion stores them (plan_column_ext)
  uint32_t dictn_idx = 0, dicttotal_idx = 0, dicterr_idx = 0, utf8err_idx = 0;
  uint64_t dictlens_off = 0;
  uint64_t n_term_words = 0, tmask_off = 0, tpop_off = 0, trank_off = 0, ttiles_off = 0;
  uint64_t dense_off = 0, dense2_off = 0;  // dense temporaries in scratch
  // strings
  uint64_t lens_off = 0;       // spaced int32 lengths per row (scratch)
  uint32_t chartot_off = 0;    // index into the summary per-batch char totals
  uint64_t dictoff_off = 0;    // dictionary offsets (scratch, int32[dict+1])
  uint32_t dictbytes_idx = 0;
};

struct Plan {
  int lane_id = 0;
  int call_z_lanes = -1;  // (orcgpu_ctx::call_z_lanes of the lane that plans)
  std::vector<orcgpu_staged*> stripes;
  std::vector<orcgpu_result*> results;
  std::vector<ColPlan> cols;
  std::vector<JobPlan> jobs;
  std::vector<uint64_t> scalars;
  Bump scratch;
  std::vector<Bump> result_bump;
  uint64_t n_nullcount_slots = 0, n_chartot_slots = 0;
  std::vector<int> pending_gathers;
  uint64_t dictjobs_off = 0;           // device copy of the DictJob table (scratch offset)
  uint64_t presjobs_off = 0;           // device copy of the PresJob table
  std::vector<PresJob> presjobs;
  uint64_t finjobs_off = 0;            // device copy of the finishers' job table (FinBatch)
  uint64_t finjobs_cap = 0;
  uint64_t spacejobs_off = 0;          // device copy of the SpaceJob table
  std::vector<SpaceJob> spacejobs;
  std::vector<DictJob> dictjobs;       // host copy, same order as pending_gathers
  uint64_t dictplaces_off = 0;         // device copy of the DictPlace table
  std::vector<DictPlace> dictplaces;   // per result with dictionary columns: its arena as pass 2 was launched (empty: the host placed them itself)
  std::vector<uint32_t> dictplace_result;
  uint32_t decomp_chunks = 0;        // chunks of the streams in `decomp` so far
  std::vector<DecompStream> decomp;  // compressed streams to expand before anything else  // indices into cols: dictionary string columns waiting for their gather
  uint32_t new_scalar(uint64_t v) {
    scalars.push_back(v);
    return (uint32_t)scalars.size() - 1;
  }
};

}  // namespace

// GPU_MAX_HW_QUEUES: the decoder uses several HIP streams per context (column lanes, the literals stream, the copy-back stream); the
// HIP runtime maps streams onto 4 hardware queues unless told otherwise, and kernels that share a queue run one after the other
// whatever their streams (measured: the literals kernel ran beside the sequences kernel only with more queues).  The setting is
// the HOST'S to make, before its first HIP call (`GPU_MAX_HW_QUEUES=8`; bench.py and the ctypes binding do it) --
// round 5's library constructor that called setenv() changed the queue configuration of every other HIP user of the process and
// raced with getenv() in other threads; it is gone.  Results never depend on the setting, only how much of a call runs side by side.

// =================================================================================================
extern "C" {

const char* orcgpu_version(void) { return "orcgpu 0.3 (gfx950)"; }
int orcgpu_abi_version(void) { return ORCGPU_ABI_VERSION; }

// `lane` > 0: the context of a column lane beside the caller's own (orcgpu_decode.inc).  The caller's own stream -- lane 0, which takes
// the columns with the longest Zstandard chains -- is created at the highest stream priority: its sequences kernel lasts as long as
// its longest chain, and beside another lane's sequences kernel (both hold three wavefronts' worth of LDS per CU) it got half the
// slots and started its long chains late (its span 14.5 -> 9.1 ms, the step 41.3 -> 40.6 ms).  ORCGPU_LANE_PRIORITY (development):
// bit 0 -- the later lanes' streams at the lowest priority, bit 1 -- their second streams too, bit 2 -- lane 0's stream at the
// highest (the default: 4), bit 3 -- its second stream too
static orcgpu_ctx* open_ctx(int device, const orcgpu_opts* opts, int lane) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  orcgpu_ctx* c = new orcgpu_ctx();
  c->device = device;
  constexpr int lane_prio = 4;  // (lane 0's stream at the highest priority)
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const bool low = lane > 0 && (lane_prio & 1), low_aux = lane > 0 && (lane_prio & 2);
  const bool high = lane == 0 && (lane_prio & 4), high_aux = lane == 0 && (lane_prio & 8);
  if ((low || high ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, low ? least : greatest) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return nullptr;
  }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return nullptr;
    }
  for (auto& pr : c->kev)
    for (auto& e : pr)
      if (hipEventCreate(&e) != hipSuccess) {
        delete c;
        return nullptr;
      }
  for (auto& e : c->lit_ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return nullptr;
    }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cus = (uint32_t)prop.multiProcessorCount;
  }
  // (without these the Zstandard stages simply run one after the other)
  if ((low_aux || high_aux ? hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, low_aux ? least : greatest) : hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking)) != hipSuccess) c->aux_stream = nullptr;
  for (auto& e : c->aux_ev)
    if (c->aux_stream && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(c->aux_stream);
      c->aux_stream = nullptr;
    }
  if (opts && opts->workspace_bytes) c->scratch.ensure(opts->workspace_bytes);
  c->lanes[0] = c;
  return c;
}
orcgpu_ctx* orcgpu_open(int device, const orcgpu_opts* opts) { return open_ctx(device, opts, 0); }

void orcgpu_close(orcgpu_ctx* c) {
  if (!c) return;
  for (int k = 1; k < kMaxLanes; k++)
    if (c->lanes[k]) {
      orcgpu_close(c->lanes[k]);
      c->lanes[k] = nullptr;
    }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->scratch.release();
  c->enc_a.release();
  c->enc_b.release();
  c->enc_in.release();
  c->enc_tmp.release();
  for (auto& b : c->enc_out) b.release();
  delete c->copiers;
  for (int k = 0; k < 2; k++) {
    if (c->piece[k]) (void)hipHostFree(c->piece[k]);
    if (c->piece_ev[k]) (void)hipEventDestroy(c->piece_ev[k]);
  }
  for (auto& a : c->arena_pool) (void)hipFree(a.first);
  for (auto& z : c->zones)
    if (z.second->dev) (void)hipFree(z.second->dev);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  if (c->d2h_gate) (void)hipEventDestroy(c->d2h_gate);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->fin_pinned) (void)hipHostFree(c->fin_pinned);
  for (auto& b : c->up_blocks) (void)hipHostFree(b.first);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->aux_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->lit_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& pr : c->kev)
    for (auto& e : pr)
      if (e) (void)hipEventDestroy(e);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* orcgpu_last_error(const orcgpu_ctx* c) { return c ? c->err.c_str() : "no context (no usable HIP device)"; }

// ---- staging --------------------------------------------------------------------------------------
int orcgpu_stage_stripe(orcgpu_ctx* ctx, const orcgpu_stripe_desc* d, orcgpu_staged** out) {
  if (!ctx || !d || !out) return ORCGPU_INVALID_ARGUMENT;
  // Row counts and block sizes come from file footers: every buffer of a decode is sized from them (and ranks / value
  // offsets are 32-bit), so absurd values are refused here instead of wrapping a size computation later.
  if (d->n_rows >= (1ull << 31)) {
    set_err(ctx, "stripe with %llu rows: more than 2^31 - 1 rows per stripe are not supported", (unsigned long long)d->n_rows);
    return ORCGPU_INVALID_ARGUMENT;
  }
  if (d->block_size > (1ull << 30)) {
    set_err(ctx, "compression block size %llu is larger than 1 GiB", (unsigned long long)d->block_size);
    return ORCGPU_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  orcgpu_staged* s = new orcgpu_staged();
  s->ctx = ctx;
  s->desc = *d;
  if (!s->desc.block_size) s->desc.block_size = 262144;
  if (!s->desc.batch_size) s->desc.batch_size = 8192;
  if (!s->desc.ts_base_seconds) s->desc.ts_base_seconds = 1420070400;
  s->desc.writer_timezone = nullptr;
  if (d->writer_timezone && d->writer_timezone[0]) {
    // Stripe::writer_tz (stripe.rs:167-171): the ORC epoch is midnight 2015-01-01 in that zone (timestamp.rs:133-147)
    const std::string name = d->writer_timezone;
    auto it = ctx->zones.find(name);
    bool have = it != ctx->zones.end();
    if (!have) {
      auto z = std::make_shared<orcgpu_ctx::Zone>();
      if (!find_timezone(name, z->table)) {
        // only a TIMESTAMP column needs the zone (timestamp.rs:128-147); a stripe without one stages like a zone-less stripe
        bool needs_zone = false;
        for (uint32_t k = 0; k < d->n_columns; k++) needs_zone = needs_zone || d->columns[k].orc_type == ORCGPU_T_TIMESTAMP;
        if (needs_zone) {
          set_err(ctx, "writer timezone '%s': not in the tz database, or its daylight-saving rule is in a form this reader does not handle "
                       "(looked in $TZDIR, /usr/share/zoneinfo, /usr/lib/zoneinfo, /usr/share/lib/zoneinfo, /etc/zoneinfo; set TZDIR to "
                       "the directory of a tz database, e.g. the one of Python's tzdata package)", name.c_str());
          delete s;
          return ORCGPU_UNSUPPORTED;
        }
      } else {
        z->epoch = tz_orc_epoch(z->table);
        const size_t n = z->table.at.size();
        hipError_t he = hipMalloc((void**)&z->dev, n * 12 + 16);
        if (he == hipSuccess && n) he = hipMemcpy(z->dev, z->table.at.data(), n * 8, hipMemcpyHostToDevice);
        if (he == hipSuccess && n) he = hipMemcpy(z->dev + n * 8, z->table.offs.data(), n * 4, hipMemcpyHostToDevice);
        if (he != hipSuccess) {
          set_err(ctx, "table of time zone '%s' (%zu entries): %s", name.c_str(), n, hipGetErrorString(he));
          if (z->dev) (void)hipFree(z->dev);
          delete s;
          return ORCGPU_HIP_ERROR;
        }
        it = ctx->zones.emplace(name, z).first;
        have = true;
      }
    }
    if (have) {
      s->zone = it->second;
      s->desc.ts_base_seconds = s->zone->epoch;
    }
  }
  for (uint32_t i = 0; i < d->n_streams; i++) {
    // entry points (orcgpu_stream): a run holds at most 512 values, a chunk at most block_size bytes; dictionaries come whole
    const orcgpu_stream& in = d->streams[i];
    if (!in.skip_bytes && !in.skip_values && !in.skip_bits) continue;
    bool dict_stream = in.kind == ORCGPU_S_DICTIONARY_DATA;
    for (uint32_t k = 0; k < d->n_columns; k++)
      if (d->columns[k].column_id == in.column_id && in.kind == ORCGPU_S_LENGTH && is_string_type(d->columns[k].orc_type) &&
          d->columns[k].orc_type != ORCGPU_T_BINARY &&
          (d->columns[k].encoding == ORCGPU_ENC_DICTIONARY || d->columns[k].encoding == ORCGPU_ENC_DICTIONARY_V2))
        dict_stream = true;
    // (skip_bits: bits of a BIT stream's first byte -- a PRESENT stream, the DATA stream of a Boolean column; on any other stream a
    // caller built against the header of round 4, which had padding there, would pass garbage: rejected, not ignored)
    bool bit_stream = in.kind == ORCGPU_S_PRESENT;
    for (uint32_t k = 0; k < d->n_columns && !bit_stream; k++)
      bit_stream = d->columns[k].column_id == in.column_id && in.kind == ORCGPU_S_DATA && d->columns[k].orc_type == ORCGPU_T_BOOLEAN;
    if (in.skip_values > 512 || in.skip_bytes > s->desc.block_size || dict_stream || in.skip_bits > 7 || (in.skip_bits && !bit_stream)) {
      set_err(ctx, "stream (column %u, kind %d): entry point {%u bytes, %u values} is not one a ROW_INDEX position can name", in.column_id, in.kind,
              in.skip_bytes, in.skip_values);
      delete s;
      return ORCGPU_INVALID_ARGUMENT;
    }
  }
  s->cols.assign(d->columns, d->columns + d->n_columns);
  s->desc.columns = nullptr;
  s->desc.streams = nullptr;
  Bump b;
  for (uint32_t i = 0; i < d->n_streams; i++) {
    const orcgpu_stream& in = d->streams[i];
    StagedStream st;
    st.column_id = in.column_id;
    st.kind = in.kind;
    st.len = in.len;
    st.skip_bytes = in.skip_bytes;
    st.skip_values = in.skip_values;
    st.skip_bits = in.skip_bits;
    st.off = b.take(in.len + ORC_PAD);
    s->stream_bytes += in.len;
    if (d->compression != ORCGPU_COMP_NONE) scan_chunks(in.ptr, in.len, d->compression, s->desc.block_size, st);
    if (in.entries && in.n_entries && in.n_entries < (1u << 24)) {
      // verified run starts: the stream's first byte, then the entries (each named by the chunk it lies in: the chunk's place in
      // the plain stream is only known on the device).  Anything odd drops them all: they are a hint
      bool ok = true;
      st.hints.push_back({d->compression == ORCGPU_COMP_NONE ? 0xffffffffu : 0u, st.skip_bytes});
      size_t ck = 0;
      for (uint32_t e = 0; e < in.n_entries && ok; e++) {
        const orcgpu_stream_entry& en = in.entries[e];
        if (d->compression == ORCGPU_COMP_NONE) {
          ok = en.chunk_offset + en.skip_bytes <= in.len && en.chunk_offset + en.skip_bytes <= 0xffffffffull;
          st.hints.push_back({0xffffffffu, (uint32_t)(en.chunk_offset + en.skip_bytes)});
        } else {
          while (ck < st.chunks.size() && st.chunks[ck].src_off < en.chunk_offset + 3) ck++;
          ok = ck < st.chunks.size() && st.chunks[ck].src_off == en.chunk_offset + 3;
          st.hints.push_back({(uint32_t)ck, en.skip_bytes});
        }
        if (ok && st.hints.size() >= 2) {
          const auto &a = st.hints[st.hints.size() - 2], &b = st.hints.back();
          ok = a.first < b.first || (a.first == b.first && a.second <= b.second);
        }
      }
      if (!ok) st.hints.clear();
      else if (st.hints.size() >= 2 && st.hints[0] == st.hints[1]) st.hints.erase(st.hints.begin());
    }
    s->streams.push_back(std::move(st));
  }
  s->dev_bytes = align_up(b.off + ORC_PAD);
  // ---- arena: from the pool (smallest that fits, at most twice the size) or a new allocation ----
  {
    int best = -1;
    std::unique_lock<std::mutex> pool_guard(ctx->pool_m);
    for (size_t k = 0; k < ctx->arena_pool.size(); k++)
      if (ctx->arena_pool[k].second >= s->dev_bytes && ctx->arena_pool[k].second <= 2 * s->dev_bytes + (1u << 20) &&
          (best < 0 || ctx->arena_pool[k].second < ctx->arena_pool[best].second))
        best = (int)k;
    if (best >= 0) {
      s->dev = ctx->arena_pool[best].first;
      s->dev_cap = ctx->arena_pool[best].second;
      ctx->arena_pool.erase(ctx->arena_pool.begin() + best);
    } else {
      if (hipMalloc((void**)&s->dev, s->dev_bytes) != hipSuccess) {
        // give the pool back to the device and try once more
        for (auto& a : ctx->arena_pool) (void)hipFree(a.first);
        ctx->arena_pool.clear();
        if (hipMalloc((void**)&s->dev, s->dev_bytes) != hipSuccess) {
          set_err(ctx, "hipMalloc(%zu) for the staged stripe failed", s->dev_bytes);
          delete s;
          return ORCGPU_HIP_ERROR;
        }
      }
      s->dev_cap = s->dev_bytes;
    }
  }
  // ---- pipeline: pieces of the arena go through two pinned buffers; while one is on its way to HBM (copy stream) the
  // host threads fill the other.  Nothing waits for the last piece: `ready` is recorded behind it and the decode
  // streams wait for that event, so staging stripe k + 1 overlaps decoding stripe k. ----
  auto fail = [&](const char* what, hipError_t e) {
    set_err(ctx, "%s failed: %s", what, hipGetErrorString(e));
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipFree(s->dev);
    if (s->ready) (void)hipEventDestroy(s->ready);
    delete s;
    return ORCGPU_HIP_ERROR;
  };
  hipError_t e = hipSuccess;
  if (!ctx->copy_stream) {
    e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
      e = hipHostMalloc((void**)&ctx->piece[k], kStagePiece + 4096, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->piece_ev[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) ctx->copiers = new CopyPool();
    if (e != hipSuccess) {
      ctx->copy_stream = nullptr;
      return fail("setting up the staging pipeline", e);
    }
  }
  e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
  if (e != hipSuccess) return fail("hipEventCreate", e);
  {
    // the arena as a list of segments: stream bytes, then ORC_PAD zero bytes behind each
    struct Seg {
      uint64_t off;
      const uint8_t* src;
      uint64_t n;
    };
    std::vector<Seg> segs;
    for (uint32_t i = 0; i < d->n_streams; i++) {
      if (d->streams[i].len) segs.push_back(Seg{s->streams[i].off, d->streams[i].ptr, d->streams[i].len});
      segs.push_back(Seg{s->streams[i].off + d->streams[i].len, nullptr, ORC_PAD});
    }
    size_t si = 0;
    uint64_t within = 0;  // bytes of segs[si] already sent
    uint64_t pos = 0;     // arena offset where the next piece starts
    std::vector<CopyPool::Task> tasks;
    while (si < segs.size()) {
      pos = segs[si].off + within;
      const int k = ctx->piece_next;
      ctx->piece_next ^= 1;
      if (ctx->piece_used[k]) {
        e = hipEventSynchronize(ctx->piece_ev[k]);  // its previous content has left
        if (e != hipSuccess) return fail("hipEventSynchronize", e);
      }
      tasks.clear();
      uint64_t end = pos;
      while (si < segs.size()) {
        const uint64_t so = segs[si].off + within;
        if (so - pos >= kStagePiece) break;
        const uint64_t take = std::min<uint64_t>(segs[si].n - within, kStagePiece - (so - pos));
        // (alignment gaps between segments are left as they are: nobody reads them)
        tasks.push_back(CopyPool::Task{ctx->piece[k] + (so - pos), segs[si].src ? segs[si].src + within : nullptr, (size_t)take});
        end = so + take;
        within += take;
        if (within == segs[si].n) {
          si++;
          within = 0;
        } else {
          break;  // the piece is full
        }
      }
      ctx->copiers->copy_all(tasks);
      e = hipMemcpyAsync(s->dev + pos, ctx->piece[k], end - pos, hipMemcpyHostToDevice, ctx->copy_stream);
      if (e == hipSuccess) e = hipEventRecord(ctx->piece_ev[k], ctx->copy_stream);
      if (e != hipSuccess) return fail("hipMemcpyAsync", e);
      ctx->piece_used[k] = true;
    }
  }
  e = hipEventRecord(s->ready, ctx->copy_stream);
  if (e != hipSuccess) return fail("hipEventRecord", e);
  *out = s;
  return ORCGPU_OK;
}

void orcgpu_staged_free(orcgpu_staged* s) {
  if (!s) return;
  if (s->ready) {
    (void)hipEventSynchronize(s->ready);  // a copy may still be writing the arena
    (void)hipEventDestroy(s->ready);
  }
  if (s->dev) {
    // back to the pool (decodes that used the stripe have returned: they end with a stream synchronisation)
    orcgpu_ctx* c = s->ctx;
    std::lock_guard<std::mutex> pool_guard(c->pool_m);
    size_t pooled = 0;
    for (auto& a : c->arena_pool) pooled += a.second;
    if (c->arena_pool.size() < 64 && pooled + s->dev_cap <= (8ull << 30)) c->arena_pool.push_back({s->dev, s->dev_cap});
    else (void)hipFree(s->dev);
  }
  delete s;
}
uint64_t orcgpu_staged_bytes(const orcgpu_staged* s) { return s ? s->stream_bytes : 0; }

}  // extern "C"

// =================================================================================================
namespace {

// Resolve one stream of a column into a PlainStream (device pointer or scratch slot to decompress into)
PlainStream plan_stream(Plan& P, orcgpu_staged* s, uint32_t col, int kind) {
  PlainStream ps;
  const StagedStream* st = s->find(col, kind);
  if (!st) {
    // StreamMap::get: a missing stream is an empty stream (stripe.rs:319-326)
    ps.exists = false;
    ps.dev = s->dev;  // any valid pointer; length 0
    ps.len_upper = 0;
    ps.len_idx = P.new_scalar(0);
    return ps;
  }
  ps.exists = true;
  ps.skip_values = st->skip_values;
  ps.skip_bits = st->skip_bits;
  ps.src = st;
  if (s->desc.compression == ORCGPU_COMP_NONE) {
    const uint64_t skip = std::min<uint64_t>(st->skip_bytes, st->len);
    ps.dev = s->dev + st->off + skip;
    ps.len_upper = st->len - skip;
    ps.len_idx = P.new_scalar(st->len - skip);
  } else {
    uint64_t upper = 0;
    for (auto& c : st->chunks) upper += c.plain_cap;
    ps.len_upper = upper;
    ps.scratch_off = P.scratch.take(upper + ORC_PAD);
    ps.len_idx = P.new_scalar(0);  // written by the decompress finalize kernel
    ps.err_idx = P.new_scalar(0);
    ps.chunk0 = P.decomp_chunks;
    P.decomp_chunks += (uint32_t)st->chunks.size();
    P.decomp.push_back(DecompStream{s, st, ps.scratch_off, ps.len_idx, ps.err_idx});
    // (the decoders start behind the bytes of the rows before the entry point; the finalize kernel publishes what is left)
    const uint64_t skip = std::min<uint64_t>(st->skip_bytes, upper);
    ps.scratch_off += skip;
    ps.len_upper -= skip;
  }
  return ps;
}

int add_job(Plan& P, int cls, uint8_t codec, bool is_signed, uint8_t nbits, uint8_t out_bytes, const PlainStream& ps, uint32_t needed_idx,
            uint64_t expect_values, int stripe, int col, int role) {
  JobPlan j{};
  j.cls = cls;
  j.codec = codec;
  j.is_signed = is_signed;
  j.nbits = nbits;
  j.out_bytes = out_bytes;
  j.data = ps.dev;
  j.data_scratch_off = ps.scratch_off;
  j.len_upper = ps.len_upper;
  j.len_idx = ps.len_idx;
  j.needed_idx = needed_idx;
  j.total_idx = P.new_scalar(0);
  j.expect_values = expect_values + ps.skip_values;
  j.skip_values = ps.skip_values;
  if (ps.src && !ps.src->hints.empty() && ps.len_upper < 0xffffffffull) j.hint_src = ps.src;
  j.chunk0 = ps.chunk0;
  j.stripe = stripe;
  j.col = col;
  j.role = role;
  P.jobs.push_back(j);
  return (int)P.jobs.size() - 1;
}

template <typename... Args>
hipError_t launch(void (*kernel)(Args...), uint64_t nthreads_or_blocks, bool is_blocks, uint32_t bs, hipStream_t st, Args... args) {
  uint64_t grid = is_blocks ? nthreads_or_blocks : (nthreads_or_blocks + bs - 1) / bs;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(bs), 0, st, args...);
  return hipGetLastError();
}

// The finishers' small launches, filed per (pipeline stage, kernel) and sent as one grid each (device/multi_job.h).
struct FinBatch {
  struct Entry {
    int stage;
    void (*launcher)(const MJob*, uint32_t, uint32_t, hipStream_t);
    std::vector<MJob> jobs;
    uint32_t max_blocks = 0;
  };
  std::vector<Entry> entries;
  size_t n_jobs = 0;
  template <auto Body, int BS>
  static void launcher(const MJob* d, uint32_t n, uint32_t max_blocks, hipStream_t st) {
    for (uint32_t o = 0; o < n; o += 65535u)
      hipLaunchKernelGGL((mj_kernel<Body, BS>), dim3(max_blocks, std::min<uint32_t>(65535u, n - o)), dim3(BS), 0, st, d + o);
  }
  // same arguments as launch(): a thread (or block) count, then the kernel's own arguments
  template <auto Body, int BS, typename... B>
  void add(int stage, uint64_t nthreads_or_blocks, bool is_blocks, B... args) {
    const uint64_t grid = is_blocks ? nthreads_or_blocks : (nthreads_or_blocks + BS - 1) / BS;
    if (!grid) return;
    auto fn = &launcher<Body, BS>;
    Entry* e = nullptr;
    for (auto& x : entries)
      if (x.stage == stage && x.launcher == fn) e = &x;
    if (!e) {
      entries.push_back(Entry{stage, fn, {}, 0});
      e = &entries.back();
    }
    MJob j{};
    using S = MjSig<decltype(Body)>;
    S::pack(j, std::make_index_sequence<S::n>{}, args...);
    j.nblocks = (uint32_t)grid;
    e->jobs.push_back(j);
    e->max_blocks = std::max(e->max_blocks, j.nblocks);
    n_jobs++;
  }
  void clear() {
    entries.clear();
    n_jobs = 0;
  }
};

struct SummaryLayout {
  uint64_t scalars_off = 0, jobs_off = 0, nullc_off = 0, chartot_off = 0, bytes = 0;
};

}  // namespace

#include "orcgpu_ext.inc"
#include "orcgpu_decode.inc"
#include "orcgpu_export.inc"
#include "orcgpu_select.inc"
#include "orcgpu_reader.inc"
#include "orcgpu_encode.inc"
