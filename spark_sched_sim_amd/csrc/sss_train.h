// sss_train.h - kernels of the PPO update (reference trainers/ppo.py:73-138 runs
// loss.backward() on the accelerator, schedulers/scheduler.py:37-54).
//
// The weight gradient of a Linear layer over a minibatch of K rows,
//   gw[n][m] = sum_k dy[k][n] * x[k][m],   gb[n] = sum_k dy[k][n]        (M, N <= 64, K = 10^5 .. 10^7),
// is a [N x K] x [K x M] product with a tiny output and an enormous reduction dimension: the shape the
// BLAS library handles worst (profiles/r03_ppo.md: 68 % of a PPO update's device time at 0.87 ms per call).
// Here: the rows are dealt to the waves of the grid in groups of four, each wave keeps the whole [N x M] result
// in MFMA accumulators (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), reads x and dy exactly
// once; the four waves of a workgroup add their tiles in LDS and write one partial; a second small kernel adds the
// partials in a fixed order (no atomics: the result does not depend on scheduling).
#pragma once
#include <stdint.h>

// one MLP of the architecture, forward or backward (kernels: sss_train16.h; arithmetic stated there)
struct SssMlpArgs {
  int64_t rows;
  int32_t in_dim, h1, h2, out_dim;
  int32_t act;       // 0: LeakyReLU(slope), 1: Tanh
  float slope;
  const float* w;    // packed parameters [W1 (H1 x IN), b1, W2^T (H1 x H2), b2, W3 (OUT x H2), b3] (sss_gnn.h)
  const float* x;    // f32[rows, IN]
  float* a1;         // f32[rows, H1]   forward: written; backward: read
  float* a2;         // f32[rows, H2]
  float* y;          // forward: f32[rows, OUT]
  const float* dy;   // backward: f32[rows, OUT]
  float* g1;         // backward: f32[rows, H1]
  float* g2;         // backward: f32[rows, H2]
  float* dx;         // backward: f32[rows, IN], or null (the input needs no gradient)
  // a row of the input in two pieces (the DAG encoder's cat([x, h_node], -1) never built): columns 0 .. IN - 17 from x (rows of
  // IN - 16 floats), columns IN - 16 .. IN - 1 from x2 (rows of 16 floats); dx2: the gradient of the x2 piece only (dx unused)
  const float* x2;
  float* dx2;
};


struct SssWgradArgs {
  const float* x;   // [K][ldx], M columns used
  const float* dy;  // [K][ldy], N columns used
  int64_t K, ldx, ldy;
  int M, N;
  float* partial;   // [n_partials][N * M + N]
  int n_partials;   // workgroups of the grid (4 waves each)
  float* gw;        // [N][M]
  float* gb;        // [N] (nullable)
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef float sss_v4f __attribute__((ext_vector_type(4)));

// operand layout of v_mfma_f32_16x16x4_f32 (A: 16 x 4, B: 4 x 16, D: 16 x 16): lane l supplies A[l % 16][l / 16] and
// B[l / 16][l % 16]; it receives D[4 * (l / 16) + v][l % 16] in element v of the accumulator. Here A = dy^T (row = output
// neuron n, k = one of four consecutive minibatch rows), B = x (column = input feature m).
template <int NT, int MT>
__global__ __launch_bounds__(256) void sss_wgrad_partial_kernel(SssWgradArgs a) {
  const int lane = threadIdx.x & 63, wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), n_waves = 4 * a.n_partials;
  const int c16 = lane & 15, r4 = lane >> 4;
  sss_v4f acc[NT][MT];
  float bsum[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    bsum[t] = 0.0f;
#pragma unroll
    for (int u = 0; u < MT; u++) acc[t][u] = (sss_v4f){0.0f, 0.0f, 0.0f, 0.0f};
  }
  const int64_t n_groups = (a.K + 3) / 4;  // groups of four rows, dealt round-robin to the waves
  for (int64_t g = wave; g < n_groups; g += n_waves) {
    const int64_t row = g * 4 + r4;
    const bool ok = row < a.K;
    float av[NT], bv[MT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int n = t * 16 + c16;
      av[t] = (ok && n < a.N) ? a.dy[row * a.ldy + n] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < MT; u++) {
      const int m = u * 16 + c16;
      bv[u] = (ok && m < a.M) ? a.x[row * a.ldx + m] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
      bsum[t] += av[t];
#pragma unroll
      for (int u = 0; u < MT; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
    }
  }
  // the four waves of the workgroup add their tiles up in LDS (in wave order), the workgroup writes ONE partial
  __shared__ float red[NT * MT * 256 + NT * 16];
  const int w = threadIdx.x >> 6;
  for (int turn = 0; turn < 4; turn++) {
    if (w == turn) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int u = 0; u < MT; u++)
#pragma unroll
          for (int v = 0; v < 4; v++) {
            float* q = &red[((t * MT + u) * 4 + v) * 64 + lane];
            *q = turn == 0 ? acc[t][u][v] : *q + acc[t][u][v];
          }
        // the bias gradient: this lane summed dy[.][n] over the rows of its quarter; the four quarters of a column meet here
        float sb = bsum[t];
        sb += __shfl_xor(sb, 16, 64);
        sb += __shfl_xor(sb, 32, 64);
        if (r4 == 0) {
          float* q = &red[NT * MT * 256 + t * 16 + c16];
          *q = turn == 0 ? sb : *q + sb;
        }
      }
    }
    __syncthreads();
  }
  float* out = a.partial + (size_t)blockIdx.x * (size_t)(a.N * a.M + a.N);
  for (int i = threadIdx.x; i < NT * MT * 256; i += 256) {
    const int tile = i >> 8, v = (i >> 6) & 3, l = i & 63;  // red[((tile * 4) + v) * 64 + lane] = D[4 * (lane / 16) + v][lane % 16] of the tile
    const int n = (tile / MT) * 16 + 4 * (l >> 4) + v, m = (tile % MT) * 16 + (l & 15);
    if (n < a.N && m < a.M) out[n * a.M + m] = red[i];
  }
  if (threadIdx.x < NT * 16 && (int)threadIdx.x < a.N) out[a.N * a.M + threadIdx.x] = red[NT * MT * 256 + threadIdx.x];
}

// adds the per-workgroup partials in a fixed order: 16 consecutive outputs x 16 interleaved subsets of the partials per
// workgroup (64-byte rows stay coalesced), then a tree over the 16 subsets in LDS
__global__ __launch_bounds__(256) void sss_wgrad_reduce_kernel(SssWgradArgs a) {
  __shared__ float part[16][17];
  const int j = threadIdx.x & 15, q = threadIdx.x >> 4, n_out = a.N * a.M + a.N;
  const int i = (int)blockIdx.x * 16 + j;
  float s = 0.0f;
  if (i < n_out)
    for (int p = q; p < a.n_partials; p += 16) s += a.partial[(size_t)p * n_out + i];
  part[q][j] = s;
  __syncthreads();
  if (q == 0 && i < n_out) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += part[k][j];
    if (i < a.N * a.M)
      a.gw[i] = t;
    else if (a.gb)
      a.gb[i - a.N * a.M] = t;
  }
}
#endif
