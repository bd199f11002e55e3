// Dependent-load latency of one wave per "env" (debug tool): every wave chases K dependent 8-byte loads inside its own region of a
// large buffer - does it matter whether an env's touched lines are spread over the whole region (1.2 MB at config 3: the pool
// overflow area sized for the worst case) or sit in a compact window of it? usage: memlat [envs] [env_stride_bytes]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void chase(const uint64_t* buf, uint64_t stride_words, uint64_t span_words, int K, uint64_t* out_ticks, uint64_t* sink) {
  const uint64_t* base = buf + (uint64_t)blockIdx.x * stride_words;
  uint64_t x = blockIdx.x * 977 + 13, acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < K; i++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint64_t at = ((x >> 20) % span_words) & ~7ull;  // a 64-byte line of the window
    const uint64_t v = base[at];
    x ^= v;  // the next address depends on the value
    acc += v;
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out_ticks[blockIdx.x] = t1 - t0, sink[blockIdx.x] = acc;
}
int main(int argc, char** argv) {
  const int envs = argc > 1 ? atoi(argv[1]) : 4096;
  const uint64_t stride = argc > 2 ? strtoull(argv[2], 0, 10) : 1200000;
  const uint64_t stride_words = (stride + 63) / 64 * 8;
  uint64_t *buf, *ticks, *sink;
  hipMalloc(&buf, stride_words * 8 * envs), hipMalloc(&ticks, envs * 8), hipMalloc(&sink, envs * 8);
  hipMemset(buf, 0, stride_words * 8 * envs);
  uint64_t* h = (uint64_t*)malloc(envs * 8);
  const int K = 256;
  for (uint64_t span : {(uint64_t)4096, (uint64_t)32768, (uint64_t)262144, stride_words * 8}) {
    for (int rep = 0; rep < 3; rep++) chase<<<envs, 64>>>(buf, stride_words, span / 8, K, ticks, sink);
    hipMemcpy(h, ticks, envs * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < envs; i++) s += (double)h[i];
    printf("envs %d, env stride %llu B, window %8llu B: %7.0f ticks per dependent load (s_memtime, %d loads per wave, all waves at once)\n", envs,
           (unsigned long long)(stride_words * 8), (unsigned long long)span, s / envs / K, K);
  }
  return 0;
}
