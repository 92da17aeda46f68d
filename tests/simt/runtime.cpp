// SIMT emulation runtime (see shim/hip/hip_runtime.h): fibers, wave rendezvous, workgroup barriers, launches.
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifdef __SANITIZE_ADDRESS__
#include <sanitizer/asan_interface.h>
#endif

#include <atomic>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch,.-simt_switch
)");

namespace simt {

constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 256 * 1024;

struct WaveSync {
  uint64_t val[2][64];
  uint64_t part[2];  // lanes that deposited (a lane that has left the kernel takes no part)
  int arrived, live;
  unsigned phase;
};
struct Workgroup;
struct Fiber {
  void *sp;
  char *stack;
  Workgroup *wg;
  Idx3 tid;
  int linear, wave, lane;
  int pos;  // place in the workgroup's schedule ring
  bool done;
};
struct Workgroup {
  Fiber f[MAX_THREADS];
  WaveSync wave[MAX_THREADS / 64];
  int n, live, bar_arrived, bar_or[2];
  unsigned bar_phase;
  unsigned long progress;
  void *main_sp;
  int ring[MAX_THREADS];  // the order the fibers take turns in (SIMT_ORDER: forward, reverse or a seeded shuffle)
  void *dyn_lds;
  size_t dyn_cap;
  Idx3 bid, bdim, gdim;
  const std::function<void()> *body;
};

thread_local Fiber *cur = nullptr;

const Idx3 &thread_idx() { return cur->tid; }
const Idx3 &block_idx() { return cur->wg->bid; }
const Idx3 &block_dim() { return cur->wg->bdim; }
const Idx3 &grid_dim() { return cur->wg->gdim; }
int lane_id() { return cur->lane; }
void *dynamic_lds() { return cur->wg->dyn_lds; }

static void yield_from(Fiber *me) {
  Workgroup *g = me->wg;
  int p = me->pos;
  for (int k = 0; k < g->n; ++k) {
    p = p + 1 == g->n ? 0 : p + 1;
    Fiber *f = &g->f[g->ring[p]];
    if (!f->done) {
      if (f == me) return;
      cur = f;
      simt_switch(&me->sp, f->sp);
      return;
    }
  }
  // nobody else is live
  if (me->done) {
    cur = nullptr;
    simt_switch(&me->sp, g->main_sp);
  }
}

static void wait_until(Fiber *me, const unsigned *phase, unsigned ph, const char *what) {
  Workgroup *g = me->wg;
  unsigned long seen = g->progress;
  int idle = 0;
  while (*(volatile const unsigned *)phase == ph) {
    yield_from(me);
    if (g->progress != seen) seen = g->progress, idle = 0;
    else if (++idle > 4) {
      fprintf(stderr, "simt: deadlock at a %s (block %u,%u thread %u): not every live lane reaches it\n", what,
              g->bid.x, g->bid.y, me->tid.x);
      abort();
    }
  }
}

// Two value buffers alternate: the lanes of operation k read buffer k & 1 after the rendezvous while the first of them
// may already deposit for k + 1; buffer k & 1 is written again by operation k + 2 only, which cannot begin before every
// lane has arrived at k + 1, i.e. has read its values of k.
static const uint64_t *exchange(uint64_t v, uint64_t *participants) {
  Fiber *me = cur;
  WaveSync &w = me->wg->wave[me->wave];
  const unsigned ph = w.phase;
  w.val[ph & 1][me->lane] = v;
  w.part[ph & 1] |= 1ull << me->lane;
  me->wg->progress++;
  if (++w.arrived == w.live) {
    w.arrived = 0;
    w.part[(ph + 1) & 1] = 0;
    w.phase = ph + 1;
  } else {
    wait_until(me, &w.phase, ph, "wave operation");
  }
  if (participants) *participants = w.part[ph & 1];
  return w.val[ph & 1];
}
const uint64_t *wave_exchange(uint64_t v) { return exchange(v, nullptr); }
uint64_t wave_ballot(bool pred) {
  uint64_t part = 0, m = 0;
  const uint64_t *v = exchange(pred ? 1u : 0u, &part);
  for (int l = 0; l < 64; ++l) m |= (v[l] & 1u) << l;
  return m & part;
}

// the barrier with a vote: returns how many of the arriving threads passed a nonzero v
int wg_barrier_count(int v) {
  Fiber *me = cur;
  Workgroup *g = me->wg;
  const unsigned ph = g->bar_phase;
  if (v) g->bar_or[ph & 1] += 1;
  g->progress++;
  if (++g->bar_arrived == g->live) {
    g->bar_arrived = 0;
    g->bar_or[(ph + 1) & 1] = 0;
    g->bar_phase = ph + 1;
  } else {
    wait_until(me, &g->bar_phase, ph, "workgroup barrier");
  }
  return g->bar_or[ph & 1];
}
int wg_barrier_or(int v) { return wg_barrier_count(v) != 0; }
void wg_barrier() { (void)wg_barrier_count(0); }

static void fiber_main() {
  Fiber *me = cur;
  (*me->wg->body)();
  me = cur;
  Workgroup *g = me->wg;
  me->done = true;
  g->progress++;
  // a lane that has left no longer takes part in rendezvous (what it deposited stays readable)
  WaveSync &w = g->wave[me->wave];
  --w.live;
  if (w.live > 0 && w.arrived == w.live) w.arrived = 0, w.part[(w.phase + 1) & 1] = 0, w.phase++;
  --g->live;
  if (g->live > 0 && g->bar_arrived == g->live) {
    g->bar_arrived = 0;
    g->bar_or[(g->bar_phase + 1) & 1] = 0;
    g->bar_phase++;
  }
  yield_from(me);
  // not reached when another fiber or the main context took over
  cur = nullptr;
  simt_switch(&me->sp, g->main_sp);
  abort();
}

static void run_workgroup(Workgroup *g, Idx3 bid, Idx3 bdim, Idx3 gdim, const std::function<void()> *body) {
  const int n = (int)(bdim.x * bdim.y * bdim.z);
  if (n > MAX_THREADS) abort();
  g->n = n, g->live = n, g->bar_arrived = 0, g->bar_phase = 0, g->bar_or[0] = g->bar_or[1] = 0, g->progress = 0;
  g->bid = bid, g->bdim = bdim, g->gdim = gdim, g->body = body;
  const int nw = (n + 63) / 64;
  for (int w = 0; w < nw; ++w) {
    g->wave[w].arrived = 0, g->wave[w].phase = 0;
    g->wave[w].live = w == nw - 1 ? n - 64 * w : 64;
    memset(g->wave[w].val, 0, sizeof(g->wave[w].val));
    g->wave[w].part[0] = g->wave[w].part[1] = 0;
  }
  for (int i = 0; i < n; ++i) {
    Fiber &f = g->f[i];
    if (!f.stack) f.stack = (char *)aligned_alloc(64, STACK_BYTES);
#ifdef __SANITIZE_ADDRESS__
    // (the frames a fiber left when it switched away for the last time never unpoisoned their redzones)
    __asan_unpoison_memory_region(f.stack, STACK_BYTES);
#endif
    f.wg = g, f.linear = i, f.wave = i >> 6, f.lane = i & 63, f.done = false;
    f.tid.x = i % bdim.x, f.tid.y = (i / bdim.x) % bdim.y, f.tid.z = i / (bdim.x * bdim.y);
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)63;
    void **s = (void **)top;
    s[-1] = nullptr;              // (a return address nobody returns to)
    s[-2] = (void *)&fiber_main;  // popped by simt_switch's ret
    for (int r = 3; r <= 8; ++r) s[-r] = nullptr;
    f.sp = (void *)(s - 8);
  }
  // The order the fibers take turns in.  Between two rendezvous a fiber runs undisturbed, so the schedule decides
  // which lane's plain LDS / global accesses come first: "forward" lets thread 0 write before anybody reads -- what a
  // missing barrier needs to go unnoticed --, "reverse" and a seeded shuffle do not.
  const char *order = getenv("SIMT_ORDER");  // forward (default) | reverse | random[:seed]
  for (int i = 0; i < n; ++i) g->ring[i] = i;
  if (order && !strncmp(order, "reverse", 7)) {
    for (int i = 0; i < n; ++i) g->ring[i] = n - 1 - i;
  } else if (order && !strncmp(order, "random", 6)) {
    uint64_t x = (order[6] == ':' ? strtoull(order + 7, nullptr, 10) : 1u) * 0x9E3779B97F4A7C15ull + bid.x * 0x100000001B3ull + bid.y * 7919u + 1u;
    for (int i = n - 1; i > 0; --i) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      const int j = (int)(x % (uint64_t)(i + 1));
      const int t = g->ring[i];
      g->ring[i] = g->ring[j], g->ring[j] = t;
    }
  }
  for (int i = 0; i < n; ++i) g->f[g->ring[i]].pos = i;
  cur = &g->f[g->ring[0]];
  simt_switch(&g->main_sp, cur->sp);
  cur = nullptr;
}

static std::mutex g_free_mutex;
static std::vector<Workgroup *> g_free;
static Workgroup *acquire_workgroup() {
  std::lock_guard<std::mutex> lock(g_free_mutex);
  if (g_free.empty()) return new Workgroup();
  Workgroup *g = g_free.back();
  g_free.pop_back();
  return g;
}
static void release_workgroup(Workgroup *g) {
  std::lock_guard<std::mutex> lock(g_free_mutex);
  g_free.push_back(g);
}

static int pool_size() {
  const char *e = getenv("SIMT_THREADS");
  int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

void launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()> &body) {
  const size_t total = (size_t)grid.x * grid.y * grid.z;
  if (total == 0) return;
  std::atomic<size_t> next{0};
  const Idx3 bdim{block.x, block.y, block.z}, gdim{grid.x, grid.y, grid.z};
  auto worker = [&]() {
    Workgroup *g = acquire_workgroup();  // (fiber stacks are kept for the next launches)
    if (g->dyn_cap < dynamic_lds_bytes) {
      free(g->dyn_lds);
      g->dyn_lds = aligned_alloc(64, (dynamic_lds_bytes + 63) / 64 * 64), g->dyn_cap = dynamic_lds_bytes;
    }
    if (dynamic_lds_bytes) memset(g->dyn_lds, 0xCD, dynamic_lds_bytes);  // (LDS is not zero on the device either)
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= total) break;
      const Idx3 bid{(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y))};
      run_workgroup(g, bid, bdim, gdim, &body);
    }
    release_workgroup(g);
  };
  const int nt = (int)std::min<size_t>((size_t)pool_size(), total);
  if (nt <= 1) {
    worker();
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back(worker);
  for (auto &t : th) t.join();
}

// ---------------------------------------------------------------------------------------------- streams and events
// (see shim/hip/hip_runtime.h).  Host calls come from one thread (the test's).
struct Event {
  uint64_t recorded = 0;  // records enqueued so far
  uint64_t done = 0;      // ... executed
};
struct Op {
  std::function<void()> run;            // what it does when it executes (may be empty)
  std::function<bool()> ready;          // empty: always
};
struct Stream {
  std::deque<Op> q;
  int id;
};
static Stream g_null_stream{{}, 0};
static std::vector<Stream *> &all_streams() {
  static std::vector<Stream *> v{&g_null_stream};
  return v;
}
static Stream *stream_of(hipStream_t s) { return s ? static_cast<Stream *>(s) : &g_null_stream; }
static size_t g_queued = 0;

// SIMT_STREAMS: unset / "immediate" | "deferred[:seed]" | "deferred:lifo" | "deferred:fifo" (read at every call, so a
// test can switch; switching drains)
static bool deferred_mode() {
  const char *m = getenv("SIMT_STREAMS");
  return m && !strncmp(m, "deferred", 8);
}

void synchronize() {
  if (g_queued == 0) return;
  const char *m = getenv("SIMT_STREAMS");
  const char *arg = (m && !strncmp(m, "deferred:", 9)) ? m + 9 : "1";
  const bool lifo = !strcmp(arg, "lifo"), fifo = !strcmp(arg, "fifo");
  uint64_t x = strtoull(arg, nullptr, 10) * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
  std::vector<Stream *> &S = all_streams();
  while (g_queued) {
    std::vector<Stream *> ready;
    for (Stream *s : S)
      if (!s->q.empty() && (!s->q.front().ready || s->q.front().ready())) ready.push_back(s);
    if (ready.empty()) {
      fprintf(stderr, "simt: stream deadlock -- %zu operations queued, every stream's head waits for something that "
                      "is not going to happen\n", g_queued);
      abort();
    }
    Stream *pick;
    if (lifo) pick = ready.back();
    else if (fifo) pick = ready.front();
    else {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      pick = ready[x % ready.size()];
    }
    Op op = std::move(pick->q.front());
    pick->q.pop_front();
    --g_queued;
    if (op.run) op.run();
  }
}

static void enqueue(hipStream_t s, Op op) {
  if (!deferred_mode()) {
    synchronize();  // (whatever a deferred phase left)
    if (op.ready && !op.ready()) {
      fprintf(stderr, "simt: a stream waits for something that has not been enqueued before it (immediate mode)\n");
      abort();
    }
    if (op.run) op.run();
    return;
  }
  stream_of(s)->q.push_back(std::move(op));
  ++g_queued;
}

void submit(hipStream_t stream, std::function<void()> op) { enqueue(stream, Op{std::move(op), nullptr}); }
hipStream_t stream_create() {
  Stream *s = new Stream();
  s->id = (int)all_streams().size();
  all_streams().push_back(s);
  return s;
}
hipEvent_t event_create() { return new Event(); }
void event_destroy(hipEvent_t e) {
  synchronize();
  delete static_cast<Event *>(e);
}
void event_record(hipEvent_t e_, hipStream_t s) {
  Event *e = static_cast<Event *>(e_);
  const uint64_t n = ++e->recorded;
  enqueue(s, Op{[e, n]() { if (e->done < n) e->done = n; }, nullptr});
}
// waits for the LATEST record enqueued before this call (none: nothing to wait for)
void stream_wait_event(hipStream_t s, hipEvent_t e_) {
  Event *e = static_cast<Event *>(e_);
  const uint64_t n = e->recorded;
  if (n == 0) return;
  enqueue(s, Op{nullptr, [e, n]() { return e->done >= n; }});
}
void stream_write_value(hipStream_t s, uint32_t *p, uint32_t v) {
  enqueue(s, Op{[p, v]() { __atomic_store_n(p, v, __ATOMIC_RELEASE); }, nullptr});
}
bool stream_wait_value(hipStream_t s, const uint32_t *p, uint32_t v, uint32_t mask) {
  if (!deferred_mode()) {
    synchronize();
    return (__atomic_load_n(p, __ATOMIC_ACQUIRE) & mask) >= v;
  }
  enqueue(s, Op{nullptr, [p, v, mask]() { return (__atomic_load_n(p, __ATOMIC_ACQUIRE) & mask) >= v; }});
  return true;
}


}  // namespace simt

// for the tests: run everything that is queued (a no-op unless SIMT_STREAMS=deferred...)
extern "C" void simt_synchronize() { simt::synchronize(); }
