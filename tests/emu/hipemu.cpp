// TEST INFRASTRUCTURE ONLY -- runtime of the HIP-on-CPU simulator declared in
// tests/emu/include/hip/hip_runtime.h (see the header comment there).
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>

#if !defined(__x86_64__)
#error "the HIP-on-CPU simulator's context switch is written for x86-64"
#endif
// void hipemu_switch(void** save_sp, void* const* load_sp): push the callee-saved registers, park the
// stack pointer in *save_sp, continue on the stack *load_sp (whose top holds the same six registers
// and a return address).
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

thread_local Worker* tl_worker = nullptr;

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static void fiber_entry() {
  Worker& w = W();
  (*w.body)();
  w.cur->state = 2;
  w.live--;
  hipemu_switch(&w.cur->sp, &w.sched_sp);    // never resumed
  abort();
}

static void run_block(Worker& w, unsigned bx, unsigned by, unsigned bz, dim3 grid, dim3 block) {
  const int T = (int)(block.x * block.y * block.z);
  if ((int)w.fibers.size() < T) {
    size_t old = w.fibers.size();
    w.fibers.resize(T);
    for (size_t i = old; i < (size_t)T; ++i) w.fibers[i].stack = (char*)malloc(kStack);
  }
  w.waves.assign((T + kWave - 1) / kWave, WaveX());
  w.block_idx = dim3(bx, by, bz);
  w.block_dim = block;
  w.grid_dim = grid;
  w.at_barrier = 0;
  w.live = T;
  for (int i = 0; i < T; ++i) {
    Fiber& f = w.fibers[i];
    f.linear = i;
    f.tid.x = i % block.x;
    f.tid.y = (i / block.x) % block.y;
    f.tid.z = i / (block.x * block.y);
    f.state = 0;
    f.wseq = 0;
    // initial frame: six zeroed callee-saved registers, fiber_entry as the return address of the first
    // switch, a null return address above it (fiber_entry never returns); rsp % 16 == 8 at its entry
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8 * sizeof(void*));
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    sp[6] = (void*)&fiber_entry;
    sp[7] = nullptr;
    f.sp = (void*)sp;
  }
  while (w.live > 0) {
    bool progressed = false;
    for (int i = 0; i < T; ++i) {
      Fiber& f = w.fibers[i];
      if (f.state != 0) continue;
      w.cur = &f;
      hipemu_switch(&w.sched_sp, &f.sp);
      progressed = true;
    }
    if (w.live > 0 && w.at_barrier == w.live) {
      for (int i = 0; i < T; ++i)
        if (w.fibers[i].state == 1) w.fibers[i].state = 0;
      w.at_barrier = 0;
    } else if (!progressed && w.live > 0) {
      fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live, %d at barrier\n", bx, by, bz,
              w.live, w.at_barrier);
      abort();
    }
  }
  w.cur = nullptr;
}

static std::mutex g_launch_mu;
static std::vector<std::unique_ptr<Worker>> g_workers;

void launch(const std::function<void()>& body, dim3 grid, dim3 block) {
  std::lock_guard<std::mutex> lk(g_launch_mu);
  const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("HIPEMU_THREADS")) nthreads = (unsigned)atoi(e);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nblocks) nthreads = (unsigned)nblocks;
  while (g_workers.size() < nthreads) g_workers.emplace_back(new Worker());
  std::atomic<unsigned long long> next{0};
  auto work = [&](unsigned wi) {
    Worker& w = *g_workers[wi];
    w.body = &body;
    tl_worker = &w;
    for (;;) {
      unsigned long long b = next.fetch_add(1);
      if (b >= nblocks) break;
      unsigned bx = (unsigned)(b % grid.x);
      unsigned by = (unsigned)((b / grid.x) % grid.y);
      unsigned bz = (unsigned)(b / ((unsigned long long)grid.x * grid.y));
      run_block(w, bx, by, bz, grid, block);
    }
    tl_worker = nullptr;
  };
  if (nthreads == 1) {
    Worker* saved = tl_worker;
    work(0);
    tl_worker = saved;
  } else {
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < nthreads; ++i) ts.emplace_back(work, i);
    for (auto& t : ts) t.join();
  }
}

}  // namespace hipemu
