// Latency microbenchmarks for one wave on gfx950 (debug tool): cycles per repetition of dependent
// instruction patterns that cross between the vector and the scalar unit. One wave per CU, s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define T0() uint64_t t0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T1(i) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[i] = t1 - t0
__global__ void k(uint64_t* out, uint32_t* sink, int one) {
  uint32_t v = threadIdx.x + one, acc = 0;
  { T0(); REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(acc));) T1(0); }                      // VALU -> VALU
  { uint32_t s = one; T0(); REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));) T1(1); acc += s; }       // SALU -> SALU
  { uint64_t m; uint32_t s;                                                                                 // v_cmp -> s_ff1 -> v_add (VALU->SGPR->SALU->VALU)
    T0(); REP64(asm volatile("v_cmp_eq_u32 %1, %0, %0\n s_ff1_i32_b64 %2, %1\n v_add_u32 %0, %0, %2" : "+v"(v), "=s"(m), "=s"(s));) T1(2); }
  { uint32_t s, idx = one;                                                                                  // v_readlane (SGPR lane select) -> v_add
    T0(); REP64(asm volatile("v_readlane_b32 %1, %0, %2\n v_add_u32 %0, %0, %1" : "+v"(v), "=s"(s) : "s"(idx));) T1(3); }
  { uint32_t s, idx = one;                                                                                  // SALU -> lane select of v_readlane -> SALU
    T0(); REP64(asm volatile("v_readlane_b32 %1, %0, %2\n s_and_b32 %2, %1, 31" : "+v"(v), "=s"(s), "+s"(idx));) T1(4); }
  { uint64_t m; uint32_t s;                                                                                 // v_cmp -> s_bcnt1 -> v_add
    T0(); REP64(asm volatile("v_cmp_le_u32 %1, %0, %0\n s_bcnt1_i32_b64 %2, %1\n v_add_u32 %0, %0, %2" : "+v"(v), "=s"(m), "=s"(s));) T1(5); }
  { uint64_t m = 1;                                                                                         // v_cmp -> v_cndmask with that SGPR mask (VALU->SGPR->VALU)
    T0(); REP64(asm volatile("v_cmp_le_u32 %1, %0, %0\n v_cndmask_b32 %0, %0, %0, %1" : "+v"(v), "+s"(m));) T1(6); }
  { double d = v; T0(); REP64(asm volatile("v_add_f64 %0, %0, %0" : "+v"(d));) T1(7); acc += (uint32_t)d; } // f64 add chain
  { uint64_t x = v; T0(); REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(v) : "vcc");) T1(8); acc += (uint32_t)x; }
  { T0(); REP64(asm volatile("s_branch 1f\n s_nop 0\n 1:\n" ::: "memory");) T1(9); }                        // taken branches
  { T0(); REP64(asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));) T1(10); }  // DPP chain (hazard handled by hw? needs nops: insert)
  { T0(); REP64(asm volatile("s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));) T1(11); }
  { uint32_t s; T0(); REP64(asm volatile("v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %0, %1" : "+v"(v), "=s"(s));) T1(12); }
  { uint32_t s = one; T0(); REP64(asm volatile("v_mov_b32 %0, %1\n v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1" : "+v"(v), "+s"(s));) T1(13); }  // SALU->VALU->SALU
  { T0(); REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(v) : "v"(one), "v"(acc));) T1(14); }  // 2 independent VALU per rep
  { uint64_t m; T0(); REP64(asm volatile("v_cmp_eq_u32 %1, %0, %0\n s_and_b64 %1, %1, exec\n v_cndmask_b32 %0, %0, %0, %1" : "+v"(v), "=s"(m));) T1(15); }
  sink[threadIdx.x] = v + acc;
}
int main() {
  uint64_t* out; uint32_t* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 64 * 4);
  hipMemset(out, 0, 64 * 8);
  for (int it = 0; it < 3; it++) k<<<1, 64>>>(out, sink, 1);
  uint64_t h[64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"v_add chain", "s_add chain", "v_cmp->s_ff1->v_add", "v_readlane(s idx)->v_add", "v_readlane->s_and->lane select", "v_cmp->s_bcnt1->v_add",
                         "v_cmp->v_cndmask(sgpr mask)", "v_add_f64 chain", "v_mad_u64_u32 chain", "taken s_branch", "v_min_dpp chain (no nop)", "s_nop1 + v_min_dpp chain",
                         "v_readfirstlane->v_add", "v_mov(s)->v_readfirstlane->s_add", "2 indep v_add", "v_cmp->s_and->v_cndmask"};
  for (int i = 0; i < 16; i++) printf("%-36s %6.1f cycles/rep\n", names[i], h[i] / 64.0);
  return 0;
}
