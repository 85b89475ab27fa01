// hipemu scheduler: one workgroup at a time, one ucontext fiber per work-item.
// TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

namespace hipemu {
static State g_state;
State& st() { return g_state; }

static void fiber_entry() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

static void set_tid(State& s, int t) {
  s.cur = t;
  s.tidx.x = t % s.block.x;
  s.tidx.y = (t / s.block.x) % s.block.y;
  s.tidx.z = t / (s.block.x * s.block.y);
}

static void yield_wait(int kind, unsigned long long gen) {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  f.wait_kind = kind;
  f.wait_gen = gen;
  int me = s.cur;
  swapcontext(&f.ctx, &s.sched);
  set_tid(s, me);
}

void block_barrier() {
  State& s = st();
  unsigned long long gen = s.blk_gen;
  if (++s.blk_arrived == s.nthreads) { s.blk_arrived = 0; ++s.blk_gen; return; }
  yield_wait(1, gen);
}

void wave_barrier() {
  State& s = st();
  int w = wave();
  unsigned long long gen = s.wave_gen[w];
  if (++s.wave_arrived[w] == wave_size_here()) { s.wave_arrived[w] = 0; ++s.wave_gen[w]; return; }
  yield_wait(2, gen);
}

void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
  State& s = st();
  s.grid = grid; s.block = block; s.body = body;
  s.nthreads = (int)(block.x * block.y * block.z);
  int nw = (s.nthreads + 63) / 64;
  s.dyn_smem.assign(smem + 64, 0);
  const size_t STK = 128 * 1024;
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned bx = 0; bx < grid.x; ++bx) {
    s.bidx = dim3(bx, by, bz);
    if ((int)s.fibers.size() < s.nthreads) s.fibers.resize(s.nthreads);
    s.blk_arrived = 0; s.blk_gen = 0;
    s.wave_arrived.assign(nw, 0); s.wave_gen.assign(nw, 0);
    s.xch_f.assign((size_t)nw * 64 * 2, 0.f); s.xch_u.assign((size_t)nw * 64, 0ull);
    s.xch_q.assign((size_t)nw * 64 * 2, 0.f); s.par_q.assign((size_t)nw * 64, 0);
    s.xch_m.assign((size_t)nw * 64 * 8, 0u);
    for (int t = 0; t < s.nthreads; ++t) {
      Fiber& f = s.fibers[t];
      if (f.stack.size() != STK) f.stack.resize(STK);
      f.done = false; f.wait_kind = 0; f.wait_gen = 0;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack.data();
      f.ctx.uc_stack.ss_size = STK;
      f.ctx.uc_link = &s.sched;
      makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int remaining = s.nthreads;
    while (remaining > 0) {
      bool progressed = false;
      for (int t = 0; t < s.nthreads; ++t) {
        Fiber& f = s.fibers[t];
        if (f.done) continue;
        if (f.wait_kind == 1 && s.blk_gen == f.wait_gen) continue;
        if (f.wait_kind == 2 && s.wave_gen[t >> 6] == f.wait_gen) continue;
        f.wait_kind = 0;
        set_tid(s, t);
        swapcontext(&s.sched, &f.ctx);
        progressed = true;
        if (f.done) --remaining;
      }
      if (!progressed) { fprintf(stderr, "hipemu: deadlock (divergent barrier?)\n"); abort(); }
    }
  }
}
}  // namespace hipemu
