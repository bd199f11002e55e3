// tests/emu/emu_rt.cpp - fiber runtime of the CPU wave emulator (TEST INFRASTRUCTURE).
// 64 fibers per "workgroup", strictly round-robin, hand-rolled x86-64 context switch (no
// syscalls), one workgroup at a time. See wave_rt.h for the contract.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <functional>

#include "wave_rt.h"

#if !defined(__x86_64__)
#error "the wave emulator's context switch is written for x86-64"
#endif

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

namespace emu {

const void* g_kernargs = nullptr;
static const int W = 64;
static const size_t STACK = 512 * 1024;

struct Wave {
  void* sp[W];
  void* main_sp;
  uint8_t* stacks;
  bool done[W];
  int cur;
  int env;
  int n_done;
  uint64_t slots[2][W];
  int ops[2][W];
  uint64_t gen[W];
  const std::function<void()>* body;
};
static Wave g;

int lane() { return g.cur; }
int env() { return g.env; }

static void die(const char* msg) {
  fprintf(stderr, "[wave-emu] env %d lane %d: %s\n", g.env, g.cur, msg);
  abort();
}

// switch from the current lane to the next runnable one (or back to the launcher)
static void yield_next() {
  int from = g.cur;
  if (g.n_done == W) {
    void* dummy;
    emu_switch(g.done[from] ? &dummy : &g.sp[from], g.main_sp);
    return;
  }
  int to = from;
  for (int k = 0; k < W; k++) {
    to = (to + 1) % W;
    if (!g.done[to]) break;
  }
  if (to == from && !g.done[from]) return;  // only runnable lane
  g.cur = to;
  void* dummy;
  emu_switch(g.done[from] ? &dummy : &g.sp[from], g.sp[to]);
}

static void fiber_main() {
  (*g.body)();
  g.done[g.cur] = true;
  g.n_done++;
  if (g.n_done != W) {
    // every other lane must also be finishing: if one still waits in a collective the kernel
    // has divergent collectives
    for (int i = 0; i < W; i++)
      if (!g.done[i] && g.gen[i] != g.gen[g.cur]) die("a lane left the kernel while others wait in a collective");
  }
  yield_next();
  die("resumed a finished fiber");
}

void collective(int op, uint64_t value) {
  int me = g.cur;
  if (g.n_done) die("collective after some lanes already left the kernel");
  int b = (int)(g.gen[me] & 1);
  g.slots[b][me] = value;
  g.ops[b][me] = op;
  g.gen[me]++;
  yield_next();
  // resumed: every lane has deposited generation gen[me]-1
  for (int i = 0; i < W; i++) {
    if (g.gen[i] < g.gen[me]) die("lane resumed before all lanes arrived (scheduler bug)");
    if (g.ops[b][i] != op) {
      fprintf(stderr, "[wave-emu] gen %llu: lane %d executes op %d, lane %d op %d\n", (unsigned long long)g.gen[me], me, op, i, g.ops[b][i]);
      die("lanes disagree on the collective they execute (non-uniform control flow)");
    }
  }
}

uint64_t slot(int l) {
  int b = (int)((g.gen[g.cur] - 1) & 1);
  return g.slots[b][l];
}

void launch(int grid, const std::function<void()>& body) {
  if (!g.stacks) {
    g.stacks = (uint8_t*)mmap(nullptr, STACK * W, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g.stacks == MAP_FAILED) die("mmap failed");
  }
  for (int blk = 0; blk < grid; blk++) {
    g.env = blk;
    g.body = &body;
    g.n_done = 0;
    for (int i = 0; i < W; i++) {
      g.done[i] = false;
      g.gen[i] = 0;
      // initial frame: six callee-saved registers, then the return address = fiber_main
      uint64_t* top = (uint64_t*)(g.stacks + STACK * (size_t)(i + 1));
      top -= 1;  // after `ret` pops the entry address rsp % 16 == 8, as at any function entry
      *top = 0;
      *--top = (uint64_t)(uintptr_t)&fiber_main;
      for (int r = 0; r < 6; r++) *--top = 0;
      g.sp[i] = top;
    }
    g.cur = 0;
    emu_switch(&g.main_sp, g.sp[0]);
    if (g.n_done != W) die("launcher resumed before all lanes finished");
  }
}

}  // namespace emu
