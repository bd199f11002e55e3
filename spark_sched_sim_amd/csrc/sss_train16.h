// sss_train16.h - the MLPs of the PPO update (trainers/ppo.py:104-138 through
// schedulers/decima/scheduler.py:101-139) as two kernels per MLP instead of ~14 tensor operations:
//
//   forward   a1 = act(W1 x + b1), a2 = act(W2 a1 + b2), y = W3 a2 + b3           rows x (IN | H1 | H2 | OUT), row-major
//   backward  g2 = (W3^T dy) * act'(a2), g1 = (W2^T g2) * act'(a1), dx = W1^T g1   (the gradients w.r.t. the pre-activations)
//
// and the six parameter gradients are sss_linear_wgrad(dy, a2), (g2, a1), (g1, x) (sss_train.h). Sixteen lanes per row as
// in sss_gnn16.h: lane g owns neurons g, g + 16, .. of every layer, so a row's hidden vector leaves the wave as 64-byte
// segments (one thread per row would store one float per 128-byte line and lane); values cross lanes with
// `v_mov_b32_dpp row_newbcast`, the weights sit in LDS in the order the lanes read them (forward: the images of
// sss_gnn16.h; backward: the transposes below). fp32 FMAs, fixed order: the same inputs give the same bits.
// Bound: HBM - a row moves 4 (IN + H1 + H2 + OUT) bytes forward and 4 (OUT + 2 H1 + 2 H2 + IN) bytes backward against
// 2 (IN H1 + H1 H2 + H2 OUT) flops each way (node MLPs: 276 B / 1.9 kflop).
// gfx950 only (DPP); the host backend of tests/emu evaluates the same arithmetic with plain loops.
#pragma once

#define MLP16_ROWS(r) for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < a.rows; r += (int64_t)gridDim.x * 16)

template <int ACT>
SSS_DEV float act16_grad(float a, float slope) {  // derivative of the activation, from its OUTPUT (sign-preserving / 1 - tanh^2)
  if (ACT == 0) return a > 0.0f ? 1.0f : slope;
  return 1.0f - a * a;
}

// LDS image of the transposed products, for lane g of a row:
//   t3  OUT = 16: [o][g] = W3[o][g]                      OUT = 1: [g][r] = W3[0][g + 16 r]
//   t2  [mm][r][g][q] = W2[mm + 16 r][g + 16 q]          (= W2T[g + 16 q][mm + 16 r])
//   t1  [jj][q][i]    = W1[jj + 16 q][i], i < INP = IN rounded up to 16 (zero beyond IN)
template <int IN, int H1, int H2, int OUT>
struct MlpT16 {
  static constexpr int Q1 = H1 / 16, Q2 = H2 / 16, S = (IN + 15) / 16, INP = S * 16;
  static constexpr int T3 = 0, T2 = T3 + OUT * H2, T1 = T2 + H1 * H2, TOTAL = T1 + H1 * INP;
  SSS_DEV static void stage(float* lds, const float* __restrict__ w, int tid, int nthreads) {
    const float* gW1 = w;
    const float* gW2T = gW1 + H1 * IN + H1;
    const float* gW3 = gW2T + H1 * H2 + H2;
    if (OUT == 16) {
      for (int t = tid; t < 16 * H2; t += nthreads) lds[T3 + t] = gW3[t];
    } else {
      for (int t = tid; t < H2; t += nthreads) lds[T3 + t] = gW3[(t / Q2) + 16 * (t % Q2)];
    }
    for (int t = tid; t < H1 * H2; t += nthreads) {
      const int q = t % Q1, g = (t / Q1) % 16, r = (t / (Q1 * 16)) % Q2, mm = t / (Q1 * 16 * Q2);
      lds[T2 + t] = gW2T[(g + 16 * q) * H2 + mm + 16 * r];
    }
    for (int t = tid; t < H1 * INP; t += nthreads) {
      const int i = t % INP, q = (t / INP) % Q1, jj = t / (INP * Q1);
      lds[T1 + t] = i < IN ? gW1[(jj + 16 * q) * IN + i] : 0.0f;
    }
  }
};

template <int IN, int H1, int H2, int OUT, int ACT>
__global__ __launch_bounds__(256) void sss_mlp16_fwd_kernel(SssMlpArgs a) {
  using M = Mlp16<IN, H1, H2, OUT, ACT>;
  constexpr int S = (IN + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  M::stage(w_lds, a.w, threadIdx.x, 256);
  __syncthreads();
  const int g = threadIdx.x & 15;
  MLP16_ROWS(r) {
    float aa[M::Q1], hh[M::Q2];
    M::l1_bias(w_lds, aa, g);
    static_for<S>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int len = IN - 16 * s < 16 ? IN - 16 * s : 16;
      const float xs = g < len ? a.x[r * IN + 16 * s + g] : 0.0f;
      M::template l1<16 * s, len>(w_lds, aa, xs, g);
    });
    M::l2(w_lds, aa, hh, g, a.slope);  // aa: a1, hh: a2 (both after the activation)
    static_for<M::Q1>([&](auto qc) { a.a1[r * H1 + g + 16 * decltype(qc)::value] = aa[decltype(qc)::value]; });
    static_for<M::Q2>([&](auto rc) { a.a2[r * H2 + g + 16 * decltype(rc)::value] = hh[decltype(rc)::value]; });
    if (OUT == 16)
      a.y[r * 16 + g] = M::out16(w_lds, hh[0], g, 1.0f);
    else {
      const float v = M::out1(w_lds, hh, g);
      if (g == 0) a.y[r] = v;
    }
  }
}

template <int IN, int H1, int H2, int OUT, int ACT>
__global__ __launch_bounds__(256) void sss_mlp16_bwd_kernel(SssMlpArgs a) {
  using T = MlpT16<IN, H1, H2, OUT>;
  constexpr int Q1 = T::Q1, Q2 = T::Q2, S = T::S, INP = T::INP;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  T::stage(w_lds, a.w, threadIdx.x, 256);
  __syncthreads();
  const int g = threadIdx.x & 15;
  const float* t3 = w_lds + T::T3;
  const float* t2 = w_lds + T::T2;
  const float* t1 = w_lds + T::T1;
  MLP16_ROWS(r) {
    _Pragma("clang fp contract(fast)")
    float a1v[Q1], a2v[Q2], g2[Q2], g1[Q1];
    static_for<Q1>([&](auto qc) { a1v[decltype(qc)::value] = a.a1[r * H1 + g + 16 * decltype(qc)::value]; });
    static_for<Q2>([&](auto rc) { a2v[decltype(rc)::value] = a.a2[r * H2 + g + 16 * decltype(rc)::value]; });
    // ---- through the last Linear ----
    if (OUT == 16) {
      const float dv = a.dy[r * 16 + g];
      float s0 = 0.0f, s1 = 0.0f;
      static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        s0 += t3[(2 * k) * 16 + g] * row_bcast<2 * k>(dv);
        s1 += t3[(2 * k + 1) * 16 + g] * row_bcast<2 * k + 1>(dv);
      });
      g2[0] = s0 + s1;
    } else {
      const float dv = a.dy[r];
      static_for<Q2>([&](auto rc) { g2[decltype(rc)::value] = t3[g * Q2 + decltype(rc)::value] * dv; });
    }
    static_for<Q2>([&](auto rc) {
      constexpr int rr = decltype(rc)::value;
      g2[rr] *= act16_grad<ACT>(a2v[rr], a.slope);
      a.g2[r * H2 + g + 16 * rr] = g2[rr];
    });
    // ---- through the middle Linear ----
    static_for<Q1>([&](auto qc) { g1[decltype(qc)::value] = 0.0f; });
    static_for<16>([&](auto mc) {
      constexpr int mm = decltype(mc)::value;
      static_for<Q2>([&](auto rc) {
        constexpr int rr = decltype(rc)::value;
        const float v = row_bcast<mm>(g2[rr]);
        const float* wp = t2 + ((mm * Q2 + rr) * 16 + g) * Q1;
        static_for<Q1>([&](auto qc) { g1[decltype(qc)::value] += wp[decltype(qc)::value] * v; });
      });
    });
    static_for<Q1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      g1[q] *= act16_grad<ACT>(a1v[q], a.slope);
      a.g1[r * H1 + g + 16 * q] = g1[q];
    });
    // ---- through the first Linear ----
    if (a.dx) {
      float dxs[S];
      static_for<S>([&](auto sc) { dxs[decltype(sc)::value] = 0.0f; });
      static_for<16>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        static_for<Q1>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const float v = row_bcast<jj>(g1[q]);
          const float* wp = t1 + (jj * Q1 + q) * INP + g;
          static_for<S>([&](auto sc) { dxs[decltype(sc)::value] += wp[16 * decltype(sc)::value] * v; });
        });
      });
      static_for<S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (16 * s + g < IN) a.dx[r * IN + 16 * s + g] = dxs[s];
      });
    }
  }
}

template <int IN, int H1, int H2, int OUT, int ACT>
static int mlp16_launch(const SssMlpArgs& a, bool backward, void* stream) {
  if (a.rows <= 0) return 0;
  const int64_t tiles = (a.rows + 15) / 16;
  const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
  if (backward) {
    const size_t lds = (size_t)MlpT16<IN, H1, H2, OUT>::TOTAL * sizeof(float);
    hipLaunchKernelGGL((sss_mlp16_bwd_kernel<IN, H1, H2, OUT, ACT>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  } else {
    const size_t lds = (size_t)Mlp16<IN, H1, H2, OUT, ACT>::TOTAL * sizeof(float);
    hipLaunchKernelGGL((sss_mlp16_fwd_kernel<IN, H1, H2, OUT, ACT>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

// ---- the GNN-shaped MLPs (IN -> 32 -> 16 -> 16, LeakyReLU) on the matrix cores ---------------------------------------------
// Same chaining as the inference kernels (sss_gnn_mfma.h: a wave owns 16 rows, Y^T = W X^T, K-step (t, r) = features
// {16 t + 4 q + r}, an output tile's registers are the next Linear's B operands). Forward additionally stores the two hidden
// activations (a lane holds four consecutive neurons of its row: 16-byte stores); backward runs the chain the other way with
// the transposed weights as A operands: G2^T = W3^T dY^T, G1^T = W2^T G2^T, dX^T = W1^T G1^T, each followed by the
// activation's derivative taken from the stored activations - all in registers.
// The input's 16-feature segments. One matrix x: segment u = columns 16 u .. 16 u + 15. Two pieces (SPLIT: x2 given, IN > 16 - the DAG
// encoder's [x (IN - 16 wide) | x2 (16 wide)]): segment 0 = the 16 columns of x2 (one aligned 16-byte load per lane, and the only
// segment whose gradient is wanted), segment 1 = the IN - 16 columns of x. The first Linear's sum then runs over the features in
// that order (x2's first): the same numbers as the MLP on the concatenation up to the order of fp32 additions, forward and
// recomputation alike.
template <int IN, bool SPLIT>
struct MlpSeg {
  static constexpr int P = IN - 16;  // (SPLIT) width of the first piece
  // column of W1 / of the joined row for feature f of segment u, or -1 behind the end
  static SSS_DEV int col(int u, int f) { return SPLIT ? (u == 0 ? P + f : (f < P ? f : -1)) : (16 * u + f < IN ? 16 * u + f : -1); }
  // feature f of segment u of row `row` (0 behind the end; no load under a condition)
  static SSS_DEV float at(const SssMlpArgs& a, int64_t row, int u, int f) {
    if (SPLIT) {
      if (u == 0) return a.x2[row * 16 + f];
      const float l = a.x[row * P + (f < P ? f : P - 1)];
      return f < P ? l : 0.0f;
    }
    const int c = 16 * u + f;
    const float l = a.x[row * IN + (c < IN ? c : IN - 1)];
    return c < IN ? l : 0.0f;
  }
  // the lane's four features 4 q .. 4 q + 3 of segment u
  static SSS_DEV mfma_f4 at4(const SssMlpArgs& a, int64_t row, int u, int q) {
    if (SPLIT && u == 0) return *(const mfma_f4*)(a.x2 + row * 16 + 4 * q);
    if (!SPLIT && IN % 4 == 0 && 16 * u + 16 <= IN) return *(const mfma_f4*)(a.x + row * IN + 16 * u + 4 * q);  // (a whole, aligned segment)
    return mfma_f4{at(a, row, u, 4 * q), at(a, row, u, 4 * q + 1), at(a, row, u, 4 * q + 2), at(a, row, u, 4 * q + 3)};
  }
};
template <int IN, bool SPLIT = false>
__global__ __launch_bounds__(256) void sss_mlp_mfma_fwd_kernel(SssMlpArgs a) {
  constexpr int U = (IN + 15) / 16;
  using Seg = MlpSeg<IN, SPLIT>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  MfmaGnnMlp m;
  m.load(a.w, lane, IN, SPLIT ? IN - 16 : 0, IN < 16 ? IN : 16);
  float s1[2][4];  // second input segment (IN > 16)
  if (U > 1) {
    MfmaGnnMlp m2;
    m2.load(a.w, lane, IN, SPLIT ? 0 : 16, IN - 16);
    for (int t = 0; t < 2; t++)
      for (int r = 0; r < 4; r++) s1[t][r] = m2.a1[t][r];
  }
  const int64_t n_tiles = (a.rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.rows;
    mfma_f4 x0 = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f}, x1 = x0;
    {
      // (no load under a condition: a row behind the end reads the last row and is not stored, a column behind the end reads the last
      // column and is replaced by 0 - a load inside `if (valid)` is a branch with a wait of its own behind it)
      const int64_t rc = valid ? row : a.rows - 1;
      x0 = Seg::at4(a, rc, 0, q);
      if (U > 1) x1 = Seg::at4(a, rc, 1, q);
    }
    // the first Linear and its activation (kept: the backward pass needs them), then the rest of the chain
    mfma_f4 d0 = m.b1[0], d1 = m.b1[1];
    d0 = mfma16(m.a1[0][0], x0.x, d0), d1 = mfma16(m.a1[1][0], x0.x, d1);
    d0 = mfma16(m.a1[0][1], x0.y, d0), d1 = mfma16(m.a1[1][1], x0.y, d1);
    d0 = mfma16(m.a1[0][2], x0.z, d0), d1 = mfma16(m.a1[1][2], x0.z, d1);
    d0 = mfma16(m.a1[0][3], x0.w, d0), d1 = mfma16(m.a1[1][3], x0.w, d1);
    if (U > 1) {
      d0 = mfma16(s1[0][0], x1.x, d0), d1 = mfma16(s1[1][0], x1.x, d1);
      d0 = mfma16(s1[0][1], x1.y, d0), d1 = mfma16(s1[1][1], x1.y, d1);
      d0 = mfma16(s1[0][2], x1.z, d0), d1 = mfma16(s1[1][2], x1.z, d1);
      d0 = mfma16(s1[0][3], x1.w, d0), d1 = mfma16(s1[1][3], x1.w, d1);
    }
    d0 = leaky4(d0, a.slope), d1 = leaky4(d1, a.slope);
    mfma_f4 e0 = m.b2, e1 = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
    e0 = mfma16(m.a2[0][0], d0.x, e0), e1 = mfma16(m.a2[1][0], d1.x, e1);
    e0 = mfma16(m.a2[0][1], d0.y, e0), e1 = mfma16(m.a2[1][1], d1.y, e1);
    e0 = mfma16(m.a2[0][2], d0.z, e0), e1 = mfma16(m.a2[1][2], d1.z, e1);
    e0 = mfma16(m.a2[0][3], d0.w, e0), e1 = mfma16(m.a2[1][3], d1.w, e1);
    const mfma_f4 h2 = leaky4(e0 + e1, a.slope);
    const mfma_f4 y = m.out(h2, 1.0f);
    if (valid) {
      if (a.a1) {  // (NULL: the caller's backward pass recomputes the hidden activations from x - sss_mlp_mfma_bwdw_kernel<IN, true>)
        *(mfma_f4*)(a.a1 + row * 32 + 4 * q) = d0, *(mfma_f4*)(a.a1 + row * 32 + 16 + 4 * q) = d1;
        *(mfma_f4*)(a.a2 + row * 16 + 4 * q) = h2;
      }
      *(mfma_f4*)(a.y + row * 16 + 4 * q) = y;
    }
  }
}

SSS_DEV mfma_f4 leaky4_grad(mfma_f4 g, mfma_f4 act, float slope) {  // g * act'(.) from the activation's output
  return mfma_f4{g.x * (act.x > 0.0f ? 1.0f : slope), g.y * (act.y > 0.0f ? 1.0f : slope), g.z * (act.z > 0.0f ? 1.0f : slope), g.w * (act.w > 0.0f ? 1.0f : slope)};
}

template <int IN>
__global__ __launch_bounds__(256) void sss_mlp_mfma_bwd_kernel(SssMlpArgs a) {
  constexpr int U = (IN + 15) / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const float* W1 = a.w;
  const float* W2T = W1 + 32 * IN + 32;  // [j][m] = W2[m][j]
  const float* W3 = W2T + 32 * 16 + 16;
  // A operands of the transposed products, lane (i, q), register r <-> K index 4 q + r (+ 16 t)
  float t3[4], t2[2][4], t1[U][2][4];
  for (int r = 0; r < 4; r++) t3[r] = W3[(4 * q + r) * 16 + i];                                   // W3^T[i][4 q + r]
  for (int tp = 0; tp < 2; tp++)
    for (int r = 0; r < 4; r++) t2[tp][r] = W2T[(16 * tp + i) * 16 + 4 * q + r];                   // W2^T[16 t' + i][4 q + r]
  for (int u = 0; u < U; u++)
    for (int t = 0; t < 2; t++)
      for (int r = 0; r < 4; r++) t1[u][t][r] = 16 * u + i < IN ? W1[(16 * t + 4 * q + r) * IN + 16 * u + i] : 0.0f;  // W1^T[16 u + i][16 t + 4 q + r]
  const int j = i;
  const int64_t n_tiles = (a.rows + 15) / 16;
  const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.rows;
    const mfma_f4 dy = valid ? *(const mfma_f4*)(a.dy + row * 16 + 4 * q) : zero;
    const mfma_f4 a2 = valid ? *(const mfma_f4*)(a.a2 + row * 16 + 4 * q) : zero;
    const mfma_f4 a10 = valid ? *(const mfma_f4*)(a.a1 + row * 32 + 4 * q) : zero;
    const mfma_f4 a11 = valid ? *(const mfma_f4*)(a.a1 + row * 32 + 16 + 4 * q) : zero;
    mfma_f4 g2 = zero;
    g2 = mfma16(t3[0], dy.x, g2), g2 = mfma16(t3[1], dy.y, g2), g2 = mfma16(t3[2], dy.z, g2), g2 = mfma16(t3[3], dy.w, g2);
    g2 = leaky4_grad(g2, a2, a.slope);
    mfma_f4 g10 = zero, g11 = zero;
    g10 = mfma16(t2[0][0], g2.x, g10), g11 = mfma16(t2[1][0], g2.x, g11);
    g10 = mfma16(t2[0][1], g2.y, g10), g11 = mfma16(t2[1][1], g2.y, g11);
    g10 = mfma16(t2[0][2], g2.z, g10), g11 = mfma16(t2[1][2], g2.z, g11);
    g10 = mfma16(t2[0][3], g2.w, g10), g11 = mfma16(t2[1][3], g2.w, g11);
    g10 = leaky4_grad(g10, a10, a.slope), g11 = leaky4_grad(g11, a11, a.slope);
    if (valid) {
      *(mfma_f4*)(a.g2 + row * 16 + 4 * q) = g2;
      *(mfma_f4*)(a.g1 + row * 32 + 4 * q) = g10, *(mfma_f4*)(a.g1 + row * 32 + 16 + 4 * q) = g11;
    }
    if (a.dx) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        mfma_f4 dx = zero;
        dx = mfma16(t1[u][0][0], g10.x, dx), dx = mfma16(t1[u][0][1], g10.y, dx), dx = mfma16(t1[u][0][2], g10.z, dx), dx = mfma16(t1[u][0][3], g10.w, dx);
        dx = mfma16(t1[u][1][0], g11.x, dx), dx = mfma16(t1[u][1][1], g11.y, dx), dx = mfma16(t1[u][1][2], g11.z, dx), dx = mfma16(t1[u][1][3], g11.w, dx);
        if (valid) {
          float* o = a.dx + row * IN + 16 * u + 4 * q;
          if (16 * u + 4 * q + 0 < IN) o[0] = dx.x;
          if (16 * u + 4 * q + 1 < IN) o[1] = dx.y;
          if (16 * u + 4 * q + 2 < IN) o[2] = dx.z;
          if (16 * u + 4 * q + 3 < IN) o[3] = dx.w;
        }
      }
    }
  }
}

// ---- backward with the weight gradients in the same pass (GNN-shaped MLPs) -------------------------------------------------
// The separate weight-gradient launches (sss_train.h) read x, a1, a2, dy and the g1 / g2 this kernel had just written: 256 floats
// per row of a 16-32-16-16 MLP over both kernels, a fifth of an update's device time (profiles/r04_ppo.md). Here the tile's
// g2 / g1 go through a per-wave LDS tile into the weight-gradient operand layout (lane (i, q): element i of row 4 s + q), the
// stored activations and the input are read a second time in that layout (the lines are in L1 from the first read), and twenty
// more v_mfma_f32_16x16x4_f32 per tile accumulate dW3 = dY^T A2, dW2 = G2^T A1, dW1 = G1^T X and the three bias sums in
// registers: x 16 + a1 32 + a2 16 + dy 16 floats in, dx 16 out - g1 and g2 never reach HBM. The four waves of a workgroup add
// their accumulators in LDS (wave order) and the workgroup ADDS the result to its slot of the partial arrays (one slot per
// workgroup, zeroed by the caller before the first call of a group: the layers of the message passing share them); a fixed-order
// sum over the slots (sss_wgrad_reduce_kernel) finishes. Same inputs, same bits.
#define SSS_MLPW_SLOTS 2048
struct SssMlpWgradAcc {  // per-workgroup slots: [SLOTS][N * M + N] per Linear
  float* l3;  // 16 x 16 + 16
  float* l2;  // 16 x 32 + 16
  float* l1;  // 32 x IN + 32
};
// RECOMPUTE: the two hidden activations are not read but computed again from x (the forward chain of sss_mlp_mfma_fwd_kernel, the
// same instructions in the same order: the same bits) - a forward pass that did not store them wrote 16 instead of 64 floats per
// row, and this pass reads x 16 + dy 16 instead of 80 floats per row: the stored activations were two thirds of the update's MLP
// traffic, which is what bounds these kernels (profiles/r06_ppo.md), and 48 floats per MLP row of the update's peak memory. The
// recomputed tiles come out in the layout the chain below wants (lane (i, q): neurons 4 q .. 4 q + 3 of row i) and go through the
// per-wave LDS tile into the weight-gradient operand layout like g1 / g2 do.
template <int IN, bool RECOMPUTE = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void sss_mlp_mfma_bwdw_kernel(SssMlpArgs a, SssMlpWgradAcc acc) {
  constexpr int U = (IN + 15) / 16;
  using Seg = MlpSeg<IN, SPLIT>;
  constexpr int NT = 3 + 2 * U;  // accumulator tiles: dW3, dW2 (two column tiles), dW1 (two row tiles x U column tiles)
  constexpr int TS = RECOMPUTE ? 112 : 48;  // floats per row of the LDS tile (112 = 16 mod 32: rows r, r + 1 of a half-wave's read sit on different banks)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  MfmaGnnMlp fw;       // (RECOMPUTE only) the forward chain's operands
  float fs1[2][4];     // ... second input segment (IN > 16)
  if (RECOMPUTE) {
    fw.load(a.w, lane, IN, SPLIT ? IN - 16 : 0, IN < 16 ? IN : 16);
    if (U > 1) {
      MfmaGnnMlp m2;
      m2.load(a.w, lane, IN, SPLIT ? 0 : 16, IN - 16);
      for (int t = 0; t < 2; t++)
        for (int r = 0; r < 4; r++) fs1[t][r] = m2.a1[t][r];
    }
  }
  const float* W1 = a.w;
  const float* W2T = W1 + 32 * IN + 32;
  const float* W3 = W2T + 32 * 16 + 16;
  float t3[4], t2[2][4], t1[U][2][4];
  for (int r = 0; r < 4; r++) t3[r] = W3[(4 * q + r) * 16 + i];
  for (int tp = 0; tp < 2; tp++)
    for (int r = 0; r < 4; r++) t2[tp][r] = W2T[(16 * tp + i) * 16 + 4 * q + r];
  for (int u = 0; u < U; u++)
    for (int t = 0; t < 2; t++)
      for (int r = 0; r < 4; r++) t1[u][t][r] = Seg::col(u, i) >= 0 ? W1[(16 * t + 4 * q + r) * IN + Seg::col(u, i)] : 0.0f;
  __shared__ __attribute__((aligned(16))) float tr[4][16 * TS];  // per wave: [row][g2 (16) | g1 (32) | RECOMPUTE: a2 (16) | a1 (32)] of the tile
  __shared__ float red[NT * 256 + 64];
  float* T = tr[wave];
  const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  mfma_f4 w3a = zero, w2a[2] = {zero, zero}, w1a[2][U];
  for (int t = 0; t < 2; t++)
    for (int u = 0; u < U; u++) w1a[t][u] = zero;
  float b3s = 0.0f, b2s = 0.0f, b1s[2] = {0.0f, 0.0f};
  const int64_t n_tiles = (a.rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + i;
    const bool valid = row < a.rows;
    // (no load under a condition - each would be a branch with a wait of its own behind it, one exposed round trip after the other: a row
    // behind the end reads the last row, and dy = 0 there makes everything it contributes zero)
    const int64_t rc = valid ? row : a.rows - 1;
    const mfma_f4 dyl = *(const mfma_f4*)(a.dy + rc * 16 + 4 * q);
    const mfma_f4 dy = valid ? dyl : zero;
    mfma_f4 a2, a10, a11;
    if (RECOMPUTE) {
      // the forward pass again, exactly as sss_mlp_mfma_fwd_kernel runs it
      mfma_f4 x0 = zero, x1 = zero;
      x0 = Seg::at4(a, rc, 0, q);
      if (U > 1) x1 = Seg::at4(a, rc, 1, q);
      mfma_f4 d0 = fw.b1[0], d1 = fw.b1[1];
      d0 = mfma16(fw.a1[0][0], x0.x, d0), d1 = mfma16(fw.a1[1][0], x0.x, d1);
      d0 = mfma16(fw.a1[0][1], x0.y, d0), d1 = mfma16(fw.a1[1][1], x0.y, d1);
      d0 = mfma16(fw.a1[0][2], x0.z, d0), d1 = mfma16(fw.a1[1][2], x0.z, d1);
      d0 = mfma16(fw.a1[0][3], x0.w, d0), d1 = mfma16(fw.a1[1][3], x0.w, d1);
      if (U > 1) {
        d0 = mfma16(fs1[0][0], x1.x, d0), d1 = mfma16(fs1[1][0], x1.x, d1);
        d0 = mfma16(fs1[0][1], x1.y, d0), d1 = mfma16(fs1[1][1], x1.y, d1);
        d0 = mfma16(fs1[0][2], x1.z, d0), d1 = mfma16(fs1[1][2], x1.z, d1);
        d0 = mfma16(fs1[0][3], x1.w, d0), d1 = mfma16(fs1[1][3], x1.w, d1);
      }
      d0 = leaky4(d0, a.slope), d1 = leaky4(d1, a.slope);
      mfma_f4 e0 = fw.b2, e1 = zero;
      e0 = mfma16(fw.a2[0][0], d0.x, e0), e1 = mfma16(fw.a2[1][0], d1.x, e1);
      e0 = mfma16(fw.a2[0][1], d0.y, e0), e1 = mfma16(fw.a2[1][1], d1.y, e1);
      e0 = mfma16(fw.a2[0][2], d0.z, e0), e1 = mfma16(fw.a2[1][2], d1.z, e1);
      e0 = mfma16(fw.a2[0][3], d0.w, e0), e1 = mfma16(fw.a2[1][3], d1.w, e1);
      a2 = leaky4(e0 + e1, a.slope), a10 = d0, a11 = d1;
    } else {
      a2 = *(const mfma_f4*)(a.a2 + rc * 16 + 4 * q);
      a10 = *(const mfma_f4*)(a.a1 + rc * 32 + 4 * q);
      a11 = *(const mfma_f4*)(a.a1 + rc * 32 + 16 + 4 * q);
    }
    mfma_f4 g2 = zero;
    g2 = mfma16(t3[0], dy.x, g2), g2 = mfma16(t3[1], dy.y, g2), g2 = mfma16(t3[2], dy.z, g2), g2 = mfma16(t3[3], dy.w, g2);
    g2 = leaky4_grad(g2, a2, a.slope);
    mfma_f4 g10 = zero, g11 = zero;
    g10 = mfma16(t2[0][0], g2.x, g10), g11 = mfma16(t2[1][0], g2.x, g11);
    g10 = mfma16(t2[0][1], g2.y, g10), g11 = mfma16(t2[1][1], g2.y, g11);
    g10 = mfma16(t2[0][2], g2.z, g10), g11 = mfma16(t2[1][2], g2.z, g11);
    g10 = mfma16(t2[0][3], g2.w, g10), g11 = mfma16(t2[1][3], g2.w, g11);
    g10 = leaky4_grad(g10, a10, a.slope), g11 = leaky4_grad(g11, a11, a.slope);
    // (rows behind the end: dy = 0 -> g2 = g1 = 0, they add nothing below)
    *(mfma_f4*)(T + i * TS + 4 * q) = g2, *(mfma_f4*)(T + i * TS + 16 + 4 * q) = g10, *(mfma_f4*)(T + i * TS + 32 + 4 * q) = g11;
    if (RECOMPUTE) *(mfma_f4*)(T + i * TS + 48 + 4 * q) = a2, *(mfma_f4*)(T + i * TS + 64 + 4 * q) = a10, *(mfma_f4*)(T + i * TS + 80 + 4 * q) = a11;
    if (SPLIT ? a.dx2 != nullptr : a.dx != nullptr) {
#pragma unroll
      for (int u = 0; u < (SPLIT ? 1 : U); u++) {  // (SPLIT: the gradient of the x2 piece = segment 0 only)
        mfma_f4 dx = zero;
        dx = mfma16(t1[u][0][0], g10.x, dx), dx = mfma16(t1[u][0][1], g10.y, dx), dx = mfma16(t1[u][0][2], g10.z, dx), dx = mfma16(t1[u][0][3], g10.w, dx);
        dx = mfma16(t1[u][1][0], g11.x, dx), dx = mfma16(t1[u][1][1], g11.y, dx), dx = mfma16(t1[u][1][2], g11.z, dx), dx = mfma16(t1[u][1][3], g11.w, dx);
        if (SPLIT) {
          if (valid) *(mfma_f4*)(a.dx2 + row * 16 + 4 * q) = dx;
        } else if (valid) {
          float* o = a.dx + row * IN + 16 * u + 4 * q;
          if (16 * u + 4 * q + 0 < IN) o[0] = dx.x;
          if (16 * u + 4 * q + 1 < IN) o[1] = dx.y;
          if (16 * u + 4 * q + 2 < IN) o[2] = dx.z;
          if (16 * u + 4 * q + 3 < IN) o[3] = dx.w;
        }
      }
    }
    wave_sync_local();
    // the weight gradients: K-step s = rows 4 s + q of the tile; A operand = the gradient's element i of that row, B = the input's
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int r = 4 * s4 + q;
      const bool ok = tile * 16 + r < a.rows;
      const int64_t grow = ok ? tile * 16 + r : a.rows - 1;  // (a row behind the end: the last row's finite values times its own zero gradients)
      const float dyl3 = a.dy[grow * 16 + i];
      const float A3 = ok ? dyl3 : 0.0f, B3 = RECOMPUTE ? T[r * TS + 48 + i] : a.a2[grow * 16 + i];
      w3a = __builtin_amdgcn_mfma_f32_16x16x4f32(A3, B3, w3a, 0, 0, 0), b3s += A3;
      const float A2 = T[r * TS + i];
      const float B20 = RECOMPUTE ? T[r * TS + 64 + i] : a.a1[grow * 32 + i], B21 = RECOMPUTE ? T[r * TS + 80 + i] : a.a1[grow * 32 + 16 + i];
      w2a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2, B20, w2a[0], 0, 0, 0), w2a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2, B21, w2a[1], 0, 0, 0), b2s += A2;
      const float A10 = T[r * TS + 16 + i], A11 = T[r * TS + 32 + i];
      b1s[0] += A10, b1s[1] += A11;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float B1 = Seg::at(a, grow, u, i);
        w1a[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A10, B1, w1a[0][u], 0, 0, 0), w1a[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A11, B1, w1a[1][u], 0, 0, 0);
      }
    }
    wave_sync_local();  // (the next tile's stores to T come after these reads)
  }
  // the four waves add their accumulators in LDS (wave order); tiles 0: dW3, 1..2: dW2, 3..: dW1[t][u]; then the bias sums
  for (int turn = 0; turn < 4; turn++) {
    if (wave == turn) {
      auto put = [&](int tile_id, const mfma_f4& v) {
        float* p0 = &red[(tile_id * 4) * 64 + lane];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) p0[k * 64] = turn == 0 ? vv[k] : p0[k * 64] + vv[k];
      };
      put(0, w3a), put(1, w2a[0]), put(2, w2a[1]);
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < U; u++) put(3 + t * U + u, w1a[t][u]);
      float sb[4] = {b3s, b2s, b1s[0], b1s[1]};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sb[k] += __shfl_xor(sb[k], 16, 64);
        sb[k] += __shfl_xor(sb[k], 32, 64);
        if (q == 0) {
          float* p0 = &red[NT * 256 + k * 16 + i];
          *p0 = turn == 0 ? sb[k] : *p0 + sb[k];
        }
      }
    }
    __syncthreads();
  }
  // red[(tile * 4 + v) * 64 + l] = D[4 (l / 16) + v][l % 16] of the tile -> the slot's [N][M] arrays
  float* o3 = acc.l3 + (size_t)blockIdx.x * (16 * 16 + 16);
  float* o2 = acc.l2 + (size_t)blockIdx.x * (16 * 32 + 16);
  float* o1 = acc.l1 + (size_t)blockIdx.x * (32 * IN + 32);
  for (int e = threadIdx.x; e < NT * 256; e += 256) {
    const int tile_id = e >> 8, v = (e >> 6) & 3, l = e & 63;
    const int n = 4 * (l >> 4) + v, m = l & 15;
    if (tile_id == 0) o3[n * 16 + m] += red[e];
    else if (tile_id < 3) o2[n * 32 + 16 * (tile_id - 1) + m] += red[e];
    else {
      const int t = (tile_id - 3) / U, u = (tile_id - 3) % U;
      if (Seg::col(u, m) >= 0) o1[(16 * t + n) * IN + Seg::col(u, m)] += red[e];
    }
  }
  if (threadIdx.x < 16) o3[16 * 16 + threadIdx.x] += red[NT * 256 + threadIdx.x];
  else if (threadIdx.x < 32) o2[16 * 32 + threadIdx.x - 16] += red[NT * 256 + threadIdx.x];
  else if (threadIdx.x < 64) o1[32 * IN + threadIdx.x - 32] += red[NT * 256 + threadIdx.x];
}

template <int IN>
static int mlp_mfma_bwdw_launch(const SssMlpArgs& a, const SssMlpWgradAcc& acc, void* stream) {
  if (a.rows <= 0) return 0;
  const int64_t wgs = (a.rows + 63) / 64;
  const unsigned grid = (unsigned)(wgs < SSS_MLPW_SLOTS ? wgs : SSS_MLPW_SLOTS);
  if (IN > 16 && a.x2)  // (the input rows in two pieces: always without stored activations - sss_host.h sss_mlp_check_split)
    hipLaunchKernelGGL((sss_mlp_mfma_bwdw_kernel<IN, true, (IN > 16)>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, acc);
  else if (!a.a1)  // the forward pass did not store the hidden activations: recompute them from x
    hipLaunchKernelGGL((sss_mlp_mfma_bwdw_kernel<IN, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, acc);
  else
    hipLaunchKernelGGL((sss_mlp_mfma_bwdw_kernel<IN, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, acc);
  return (int)hipGetLastError();
}

template <int IN>
static int mlp_mfma_launch(const SssMlpArgs& a, bool backward, void* stream) {
  if (a.rows <= 0) return 0;
  const int64_t wgs = (a.rows + 63) / 64;
  // (one resident set of workgroups striding over the tiles, as the inference launches: sss_gnn_mfma.h gnn_resident_workgroups)
  static GnnGridCap cache_f, cache_b;
  const int64_t cap_f = gnn_resident_workgroups(cache_f, (const void*)sss_mlp_mfma_fwd_kernel<IN, false>, 256, 0);
  const int64_t cap_b = gnn_resident_workgroups(cache_b, (const void*)sss_mlp_mfma_bwd_kernel<IN>, 256, 0);
  const int64_t cap = backward ? cap_b : cap_f;
  const unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
  if (backward)
    hipLaunchKernelGGL(sss_mlp_mfma_bwd_kernel<IN>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (IN > 16 && a.x2)
    hipLaunchKernelGGL((sss_mlp_mfma_fwd_kernel<IN, (IN > 16)>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((sss_mlp_mfma_fwd_kernel<IN, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ---- the two policy heads (IN -> 64 -> 64 -> 1, Tanh) of the update on the matrix cores ---------------------------------
// Forward: the inference chain (sss_gnn_mfma.h MfmaHead; the input is a dense row here: segment u = columns 16 u .. 16 u + 15)
// plus 16-byte stores of the two 64-wide activations. Backward: g2 = w3 * dy * (1 - a2^2) on the vector unit (one output),
// then G1^T = W2^T G2^T and dX^T = W1^T G1^T with the transposed weights as LDS A-operand images, (1 - a1^2) in between.
template <int IN>
__global__ __launch_bounds__(256) void sss_mlp_head_mfma_fwd_kernel(SssMlpArgs a) {
  constexpr int U = (IN + 15) / 16;
  using H = MfmaHead<U>;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  H::stage(w_lds, a.w, IN, threadIdx.x, 256, [](int u, int f) { return 16 * u + f < IN ? 16 * u + f : -1; });
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* W3 = a.w + 64 * IN + 64 + 64 * 64 + 64;
  float w3[16];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) w3[4 * t + r] = W3[16 * t + 4 * q + r];
  const float b3 = W3[64];
  const int64_t n_tiles = (a.rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.rows;
    mfma_f4 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {  // (unconditional loads: see sss_mlp_mfma_fwd_kernel)
        const int c = 16 * u + 4 * q + r;
        const float l = a.x[(valid ? row : a.rows - 1) * IN + (c < IN ? c : IN - 1)];
        v[r] = c < IN ? l : 0.0f;
      }
      x[u] = mfma_f4{v[0], v[1], v[2], v[3]};
    }
    mfma_f4 d[4], e[4];
    H::hidden(w_lds, x, lane, d, e);
    const float y = H::reduce(e, w3, b3);
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; t++) *(mfma_f4*)(a.a1 + row * 64 + 16 * t + 4 * q) = d[t], *(mfma_f4*)(a.a2 + row * 64 + 16 * t + 4 * q) = e[t];
      if (q == 0) a.y[row] = y;
    }
  }
}

template <int IN>
__global__ __launch_bounds__(256) void sss_mlp_head_mfma_bwd_kernel(SssMlpArgs a) {
  constexpr int U = (IN + 15) / 16;
  constexpr int T2 = 0, T1 = T2 + 4 * 16 * 64;  // (+ U * 16 * 64 floats of T1: the launch sizes the LDS)
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  {
    const float* W1 = a.w;
    const float* W2T = W1 + 64 * IN + 64;  // [n][m] = W2[m][n]
    for (int t = threadIdx.x; t < 4 * 16 * 64; t += 256) {  // T2[(tp, s)][lane] = W2^T[16 tp + i][16 (s >> 2) + 4 q + (s & 3)]
      const int lane = t & 63, s = (t >> 6) & 15, tp = t >> 10;
      w_lds[T2 + t] = W2T[(16 * tp + (lane & 15)) * 64 + 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3)];
    }
    for (int t = threadIdx.x; t < U * 16 * 64; t += 256) {  // T1[(u, s)][lane] = W1^T[16 u + i][16 (s >> 2) + 4 q + (s & 3)]
      const int lane = t & 63, s = (t >> 6) & 15, u = t >> 10;
      const int c = 16 * u + (lane & 15);
      w_lds[T1 + t] = c < IN ? W1[(16 * (s >> 2) + 4 * (lane >> 4) + (s & 3)) * IN + c] : 0.0f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* W3 = a.w + 64 * IN + 64 + 64 * 64 + 64;
  float w3[16];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) w3[4 * t + r] = W3[16 * t + 4 * q + r];
  const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  const int64_t n_tiles = (a.rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.rows;
    const float dy = valid ? a.dy[row] : 0.0f;
    mfma_f4 g2[4], g1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const mfma_f4 a2 = valid ? *(const mfma_f4*)(a.a2 + row * 64 + 16 * t + 4 * q) : zero;
      g2[t] = mfma_f4{w3[4 * t] * dy * (1.0f - a2.x * a2.x), w3[4 * t + 1] * dy * (1.0f - a2.y * a2.y), w3[4 * t + 2] * dy * (1.0f - a2.z * a2.z),
                      w3[4 * t + 3] * dy * (1.0f - a2.w * a2.w)};
      if (valid) *(mfma_f4*)(a.g2 + row * 64 + 16 * t + 4 * q) = g2[t];
    }
#pragma unroll
    for (int tp = 0; tp < 4; tp++) {
      mfma_f4 acc = zero;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const float* ap = w_lds + T2 + ((tp * 16 + 4 * t) * 64) + lane;
        acc = mfma16(ap[0], g2[t].x, acc), acc = mfma16(ap[64], g2[t].y, acc), acc = mfma16(ap[128], g2[t].z, acc), acc = mfma16(ap[192], g2[t].w, acc);
      }
      const mfma_f4 a1 = valid ? *(const mfma_f4*)(a.a1 + row * 64 + 16 * tp + 4 * q) : zero;
      g1[tp] = mfma_f4{acc.x * (1.0f - a1.x * a1.x), acc.y * (1.0f - a1.y * a1.y), acc.z * (1.0f - a1.z * a1.z), acc.w * (1.0f - a1.w * a1.w)};
      if (valid) *(mfma_f4*)(a.g1 + row * 64 + 16 * tp + 4 * q) = g1[tp];
    }
    if (a.dx) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        mfma_f4 dx = zero;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float* ap = w_lds + T1 + ((u * 16 + 4 * t) * 64) + lane;
          dx = mfma16(ap[0], g1[t].x, dx), dx = mfma16(ap[64], g1[t].y, dx), dx = mfma16(ap[128], g1[t].z, dx), dx = mfma16(ap[192], g1[t].w, dx);
        }
        if (valid) {
          float* o = a.dx + row * IN + 16 * u + 4 * q;
          if (16 * u + 4 * q + 0 < IN) o[0] = dx.x;
          if (16 * u + 4 * q + 1 < IN) o[1] = dx.y;
          if (16 * u + 4 * q + 2 < IN) o[2] = dx.z;
          if (16 * u + 4 * q + 3 < IN) o[3] = dx.w;
        }
      }
    }
  }
}

// ---- the heads' backward pass with the weight gradients in the same pass (round 6) --------------------------------------------
// sss_mlp_head_mfma_bwd_kernel writes g1 / g2 (128 floats per row) for three sss_linear_wgrad launches that read them back together
// with x, a1, a2, dy: 620 floats of traffic per row over four launches, a fifth of an update's device time (profiles/r06_ppo.md).
// Here a workgroup takes 64 rows at a time: each of its four waves runs the chain of the kernel above on 16 of them (dx to memory,
// g2 / g1 into an LDS tile, its rows' a1 and x copied to LDS as well), and after a barrier each wave accumulates ITS sixteen output
// neurons of dW2 = G2^T A1 and dW1 = G1^T X over all 64 rows (K-step = rows {4 s + q} of a tile; A operand = the gradient's element
// 16 wave + i of that row, B = the input's; a1 comes as ONE 16-byte read per lane: component c of it is column 4 i + c, so output
// tile c holds the columns {4 i + c}) - 4 + U accumulator tiles per wave, no reduction between waves. dW3 = sum dy a2 and db3 are
// summed in the chain's layout on the vector unit, db2 / db1 from the A operands. x 53 + a1 64 + a2 64 + dy 1 floats in, dx 53 out -
// g1 and g2 never reach HBM, and every input is read from HBM once: the next block's rows are asked for right after the barrier and
// arrive while the matrix cores work through the weight-gradient phase, which itself reads LDS only. (A first form re-read a1 / x /
// a2 from global memory in the second phase, two workgroups per CU: every tile's inputs were an exposed round trip, 2.5 ms for 5 M
// rows against 3.6 ms for the four launches; loads under `valid ? ... : 0` - a branch and a wait each - made it 5.0 ms.)
// The workgroup adds its sums to its slot (SSS_MLPW_HEAD_SLOTS of them, zeroed by the caller); sss_wgrad_reduce_kernel adds the
// slots in a fixed order.
#define SSS_MLPW_HEAD_SLOTS 512
template <int IN>
struct MlpHeadW {
  static constexpr int U = (IN + 15) / 16;
  static constexpr int TS = 144;  // floats per row of a gradient tile: g2 (64) | g1 (64) | pad (144 = 16 mod 32)
  static constexpr int AS = 68;   // ... of the a1 copy (68 = 4 mod 32: the four rows of a K-step's 16-byte reads start on different banks)
  static constexpr int XN = (16 * IN + 63) / 64;  // floats of a tile's x per lane
  static constexpr int T2 = 0, T1 = T2 + 4 * 16 * 64, TT = T1 + U * 16 * 64, XA = TT + 4 * 16 * TS, XX = XA + 64 * AS, TOTAL = XX + 64 * IN + 64;
};
template <int IN>
__global__ __launch_bounds__(256) void sss_mlp_head_mfma_bwdw_kernel(SssMlpArgs a, SssMlpWgradAcc acc) {
  using L = MlpHeadW<IN>;
  constexpr int U = L::U, TS = L::TS, AS = L::AS, XN = L::XN;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  {
    const float* W1 = a.w;
    const float* W2T = W1 + 64 * IN + 64;  // [n][m] = W2[m][n]
    for (int t = threadIdx.x; t < 4 * 16 * 64; t += 256) {  // T2[(tp, s)][lane] = W2^T[16 tp + i][16 (s >> 2) + 4 q + (s & 3)]
      const int lane = t & 63, s = (t >> 6) & 15, tp = t >> 10;
      w_lds[L::T2 + t] = W2T[(16 * tp + (lane & 15)) * 64 + 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3)];
    }
    for (int t = threadIdx.x; t < U * 16 * 64; t += 256) {  // T1[(u, s)][lane] = W1^T[16 u + i][16 (s >> 2) + 4 q + (s & 3)]
      const int lane = t & 63, s = (t >> 6) & 15, u = t >> 10;
      const int c = 16 * u + (lane & 15);
      w_lds[L::T1 + t] = c < IN ? W1[(16 * (s >> 2) + 4 * (lane >> 4) + (s & 3)) * IN + c] : 0.0f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* W3 = a.w + 64 * IN + 64 + 64 * 64 + 64;
  float w3[16];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) w3[4 * t + r] = W3[16 * t + 4 * q + r];
  float* Tw = w_lds + L::TT + wave * 16 * TS;
  float* Aw = w_lds + L::XA + wave * 16 * AS;
  float* Xw = w_lds + L::XX + wave * 16 * IN;
  const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  mfma_f4 w2a[4] = {zero, zero, zero, zero}, w1a[U], w3a[4] = {zero, zero, zero, zero};
#pragma unroll
  for (int u = 0; u < U; u++) w1a[u] = zero;
  float b2s = 0.0f, b1s = 0.0f, b3s = 0.0f;
  const int64_t n_blocks = (a.rows + 63) / 64;
  const int64_t x_last = a.rows * IN - 1;
  // (every load is unconditional - what lies behind the end reads the last row / element instead, and dy = 0 there makes g2 / g1 zero)
  float n_dy, n_x[XN];
  mfma_f4 n_a2[4], n_a1[4];
  auto fetch = [&](int64_t block) {
    const int64_t r0 = (block * 4 + wave) * 16;
    const int64_t rc = r0 + j < a.rows ? r0 + j : a.rows - 1;
    n_dy = a.dy[rc];
#pragma unroll
    for (int t = 0; t < 4; t++) n_a2[t] = *(const mfma_f4*)(a.a2 + rc * 64 + 16 * t + 4 * q), n_a1[t] = *(const mfma_f4*)(a.a1 + rc * 64 + 16 * t + 4 * q);
#pragma unroll
    for (int k = 0; k < XN; k++) {  // the tile's 16 x IN floats of x are contiguous: lane l takes the elements l, l + 64, ...
      const int64_t e = r0 * IN + lane + 64 * k;
      n_x[k] = a.x[e < x_last ? e : x_last];
    }
  };
  if ((int64_t)blockIdx.x < n_blocks) fetch(blockIdx.x);
  for (int64_t block = blockIdx.x; block < n_blocks; block += gridDim.x) {
    {  // the chain of sss_mlp_head_mfma_bwd_kernel on this wave's 16 rows
      const int64_t row = (block * 4 + wave) * 16 + j;
      const bool valid = row < a.rows;
      const float dy = valid ? n_dy : 0.0f;
      mfma_f4 g2[4], g1[4], a1[4];
#pragma unroll
      for (int k = 0; k < XN; k++)
        if (lane + 64 * k < 16 * IN) Xw[lane + 64 * k] = n_x[k];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const mfma_f4 a2 = n_a2[t];
        a1[t] = n_a1[t];
        *(mfma_f4*)(Aw + j * AS + 16 * t + 4 * q) = a1[t];
        w3a[t] += mfma_f4{dy * a2.x, dy * a2.y, dy * a2.z, dy * a2.w};
        g2[t] = mfma_f4{w3[4 * t] * dy * (1.0f - a2.x * a2.x), w3[4 * t + 1] * dy * (1.0f - a2.y * a2.y), w3[4 * t + 2] * dy * (1.0f - a2.z * a2.z),
                        w3[4 * t + 3] * dy * (1.0f - a2.w * a2.w)};
        *(mfma_f4*)(Tw + j * TS + 16 * t + 4 * q) = g2[t];
      }
      if (q == 0) b3s += dy;
#pragma unroll
      for (int tp = 0; tp < 4; tp++) {
        mfma_f4 s = zero;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float* ap = w_lds + L::T2 + ((tp * 16 + 4 * t) * 64) + lane;
          s = mfma16(ap[0], g2[t].x, s), s = mfma16(ap[64], g2[t].y, s), s = mfma16(ap[128], g2[t].z, s), s = mfma16(ap[192], g2[t].w, s);
        }
        g1[tp] = mfma_f4{s.x * (1.0f - a1[tp].x * a1[tp].x), s.y * (1.0f - a1[tp].y * a1[tp].y), s.z * (1.0f - a1[tp].z * a1[tp].z), s.w * (1.0f - a1[tp].w * a1[tp].w)};
        *(mfma_f4*)(Tw + j * TS + 64 + 16 * tp + 4 * q) = g1[tp];
      }
      if (a.dx) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          mfma_f4 dx = zero;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const float* ap = w_lds + L::T1 + ((u * 16 + 4 * t) * 64) + lane;
            dx = mfma16(ap[0], g1[t].x, dx), dx = mfma16(ap[64], g1[t].y, dx), dx = mfma16(ap[128], g1[t].z, dx), dx = mfma16(ap[192], g1[t].w, dx);
          }
          if (valid) {
            float* o = a.dx + row * IN + 16 * u + 4 * q;
            if (16 * u + 4 * q + 0 < IN) o[0] = dx.x;
            if (16 * u + 4 * q + 1 < IN) o[1] = dx.y;
            if (16 * u + 4 * q + 2 < IN) o[2] = dx.z;
            if (16 * u + 4 * q + 3 < IN) o[3] = dx.w;
          }
        }
      }
    }
    __syncthreads();
    fetch(block + gridDim.x);  // (behind the last block: the last row again, unused)
    // this wave's sixteen neurons (16 wave + i) of the weight gradients over the workgroup's 64 rows, from LDS
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      const float* Tt = w_lds + L::TT + tt * 16 * TS;
      const float* At = w_lds + L::XA + tt * 16 * AS;
      const float* Xt = w_lds + L::XX + tt * 16 * IN;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const int r = 4 * s4 + q;
        const float A2 = Tt[r * TS + 16 * wave + j], A1 = Tt[r * TS + 64 + 16 * wave + j];
        const mfma_f4 B2 = *(const mfma_f4*)(At + r * AS + 4 * j);  // columns 4 i .. 4 i + 3: one per output tile
        b2s += A2, b1s += A1;
        w2a[0] = mfma16(A2, B2.x, w2a[0]), w2a[1] = mfma16(A2, B2.y, w2a[1]), w2a[2] = mfma16(A2, B2.z, w2a[2]), w2a[3] = mfma16(A2, B2.w, w2a[3]);
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int col = 16 * u + j;
          const float xv = Xt[r * IN + (col < IN ? col : IN - 1)];
          w1a[u] = mfma16(A1, col < IN ? xv : 0.0f, w1a[u]);
        }
      }
    }
    __syncthreads();  // (the next block's stores to the tiles come after these reads)
  }
  // register v of lane l of an accumulator tile = D[4 (l / 16) + v][l % 16]: neuron 16 wave + 4 q + v, input column per the tile's mapping
  float* o3 = acc.l3 + (size_t)blockIdx.x * (64 + 1);
  float* o2 = acc.l2 + (size_t)blockIdx.x * (64 * 64 + 64);
  float* o1 = acc.l1 + (size_t)blockIdx.x * (64 * IN + 64);
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float vv[4] = {w2a[c].x, w2a[c].y, w2a[c].z, w2a[c].w};
#pragma unroll
    for (int v = 0; v < 4; v++) o2[(16 * wave + 4 * q + v) * 64 + 4 * j + c] += vv[v];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const float vv[4] = {w1a[u].x, w1a[u].y, w1a[u].z, w1a[u].w};
#pragma unroll
    for (int v = 0; v < 4; v++)
      if (16 * u + j < IN) o1[(16 * wave + 4 * q + v) * IN + 16 * u + j] += vv[v];
  }
  float sb[2] = {b2s, b1s};
#pragma unroll
  for (int k = 0; k < 2; k++) {
    sb[k] += __shfl_xor(sb[k], 16, 64);
    sb[k] += __shfl_xor(sb[k], 32, 64);
  }
  if (q == 0) o2[64 * 64 + 16 * wave + j] += sb[0], o1[64 * IN + 16 * wave + j] += sb[1];
  // dW3 / db3: a lane holds the sums over its row (j) of the four waves' tiles - over the 16 rows of a quarter-wave, then the waves
  // one after the other through LDS (the tiles are free now: the loop's last barrier is behind every wave)
  float* red = w_lds + L::TT;  // [wave][64 + 1]
#pragma unroll
  for (int t = 0; t < 4; t++) {
    float vv[4] = {w3a[t].x, w3a[t].y, w3a[t].z, w3a[t].w};
#pragma unroll
    for (int v = 0; v < 4; v++) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) vv[v] += __shfl_xor(vv[v], m, 64);
      if (j == 0) red[wave * 65 + 16 * t + 4 * q + v] = vv[v];
    }
  }
  {
    float v = b3s;  // (lanes q == 0 hold their row's share)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) red[wave * 65 + 64] = v;
  }
  __syncthreads();
  if (threadIdx.x < 65) o3[threadIdx.x] += ((red[threadIdx.x] + red[65 + threadIdx.x]) + red[130 + threadIdx.x]) + red[195 + threadIdx.x];
}

template <int IN>
static int mlp_head_mfma_bwdw_launch(const SssMlpArgs& a, const SssMlpWgradAcc& acc, void* stream) {
  if (a.rows <= 0) return 0;
  const size_t lds = (size_t)MlpHeadW<IN>::TOTAL * sizeof(float);
  static GnnGridCap cache, attr;  // (per device, like the grid cap: a process may drive more than one GPU)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  if (dev < 0 || dev >= 16 || !__atomic_load_n(&attr.per_device[dev], __ATOMIC_RELAXED)) {  // more than 64 KB of dynamic LDS per workgroup: asked for once per device
    if (hipFuncSetAttribute((const void*)sss_mlp_head_mfma_bwdw_kernel<IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError();
    if (dev >= 0 && dev < 16) __atomic_store_n(&attr.per_device[dev], 1, __ATOMIC_RELAXED);
  }
  const int64_t blocks = (a.rows + 63) / 64;
  int64_t cap = gnn_resident_workgroups(cache, (const void*)sss_mlp_head_mfma_bwdw_kernel<IN>, 256, lds, 256);
  if (cap > SSS_MLPW_HEAD_SLOTS) cap = SSS_MLPW_HEAD_SLOTS;
  const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
  hipLaunchKernelGGL(sss_mlp_head_mfma_bwdw_kernel<IN>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, acc);
  return (int)hipGetLastError();
}

template <int IN>
static int mlp_head_mfma_launch(const SssMlpArgs& a, bool backward, void* stream) {
  if (a.rows <= 0) return 0;
  constexpr int U = (IN + 15) / 16;
  const int64_t wgs = (a.rows + 63) / 64;
  const unsigned grid = (unsigned)(wgs < 512 ? wgs : 512);  // (32 KB of LDS images per workgroup, built once and reused over its tiles)
  if (backward)
    hipLaunchKernelGGL(sss_mlp_head_mfma_bwd_kernel<IN>, dim3(grid), dim3(256), (size_t)(4 * 16 * 64 + U * 16 * 64) * sizeof(float), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(sss_mlp_head_mfma_fwd_kernel<IN>, dim3(grid), dim3(256), (size_t)MfmaHead<U>::TOTAL * sizeof(float), (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// the MLP shapes of the published architecture (config/decima_tpch.yaml:66-78); anything else: -1 (the caller keeps autograd)
// backward + weight gradients of a GNN-shaped MLP (acc: SSS_MLPW_SLOTS slots per Linear, see sss_mlp_mfma_bwdw_kernel)
static int be_launch_mlp_bwdw(const SssMlpArgs& a, float* acc, void* stream) {
  SssMlpWgradAcc w;
#ifndef SSS_TEST_VECTOR_FORMS
  if (a.h1 == 64) {  // the two policy heads (SSS_MLPW_HEAD_SLOTS slots per Linear)
    w.l3 = acc, w.l2 = w.l3 + (size_t)SSS_MLPW_HEAD_SLOTS * (64 + 1), w.l1 = w.l2 + (size_t)SSS_MLPW_HEAD_SLOTS * (64 * 64 + 64);
    if (a.in_dim == GNN_NF + 48) return mlp_head_mfma_bwdw_launch<GNN_NF + 48>(a, w, stream);
    if (a.in_dim == GNN_DF + 33) return mlp_head_mfma_bwdw_launch<GNN_DF + 33>(a, w, stream);
    return -1;
  }
#endif
  w.l3 = acc, w.l2 = w.l3 + (size_t)SSS_MLPW_SLOTS * (16 * 16 + 16), w.l1 = w.l2 + (size_t)SSS_MLPW_SLOTS * (16 * 32 + 16);
  if (a.in_dim == GNN_NF) return mlp_mfma_bwdw_launch<GNN_NF>(a, w, stream);
  if (a.in_dim == 16) return mlp_mfma_bwdw_launch<16>(a, w, stream);
  if (a.in_dim == GNN_NF + 16) return mlp_mfma_bwdw_launch<GNN_NF + 16>(a, w, stream);
  return -1;
}
// whether sss_mlp_forward may skip storing the hidden activations of this GNN-shaped MLP and sss_mlp_backward_wgrad recomputes them
static int be_mlp_recompute_supported(int in_dim) {
#ifdef SSS_TEST_VECTOR_FORMS
  (void)in_dim;
  return 0;  // (the 16-lanes-per-row comparison kernels keep the stored-activation form)
#else
  return in_dim == GNN_NF || in_dim == 16 || in_dim == GNN_NF + 16;
#endif
}
// whether the (GNN_NF + 16)-wide MLP takes its input rows in two pieces (SssMlpArgs.x2 / dx2): the matrix-core kernels with recomputation
static int be_mlp_split_supported(int in_dim) {
#ifdef SSS_TEST_VECTOR_FORMS
  (void)in_dim;
  return 0;
#else
  return in_dim == GNN_NF + 16;
#endif
}
// whether sss_mlp_backward_wgrad takes the two policy heads (IN -> 64 -> 64 -> 1, Tanh; stored activations) as well
static int be_mlp_head_bwdw_supported() {
#ifdef SSS_TEST_VECTOR_FORMS
  return 0;  // (the comparison build keeps the backward + three weight-gradient launches)
#else
  return 1;
#endif
}
static int be_launch_wgrad_reduce(const SssWgradArgs& a, void* stream) {
  hipLaunchKernelGGL(sss_wgrad_reduce_kernel, dim3((unsigned)((a.N * a.M + a.N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
static int be_launch_mlp(const SssMlpArgs& a, int backward, void* stream) {
  const bool gnn = a.h1 == 32 && a.h2 == 16 && a.out_dim == 16 && a.act == 0;
  const bool head = a.h1 == 64 && a.h2 == 64 && a.out_dim == 1 && a.act == 1;
  // (a -DSSS_TEST_VECTOR_FORMS test build takes the 16-lanes-per-row kernels for every shape: comparisons)
#ifdef SSS_TEST_VECTOR_FORMS
  constexpr bool lanes16 = true;
#else
  constexpr bool lanes16 = false;
#endif
  if (gnn && !lanes16) {
    if (a.in_dim == GNN_NF) return mlp_mfma_launch<GNN_NF>(a, backward, stream);
    if (a.in_dim == 16) return mlp_mfma_launch<16>(a, backward, stream);
    if (a.in_dim == GNN_NF + 16) return mlp_mfma_launch<GNN_NF + 16>(a, backward, stream);
  }
  if (head && !lanes16) {
    if (a.in_dim == GNN_NF + 48) return mlp_head_mfma_launch<GNN_NF + 48>(a, backward, stream);
    if (a.in_dim == GNN_DF + 33) return mlp_head_mfma_launch<GNN_DF + 33>(a, backward, stream);
  }
  if (gnn && a.in_dim == GNN_NF) return mlp16_launch<GNN_NF, 32, 16, 16, 0>(a, backward, stream);
  if (gnn && a.in_dim == 16) return mlp16_launch<16, 32, 16, 16, 0>(a, backward, stream);
  if (gnn && a.in_dim == GNN_NF + 16) return mlp16_launch<GNN_NF + 16, 32, 16, 16, 0>(a, backward, stream);
  if (head && a.in_dim == GNN_NF + 48) return mlp16_launch<GNN_NF + 48, 64, 64, 1, 1>(a, backward, stream);
  if (head && a.in_dim == GNN_DF + 33) return mlp16_launch<GNN_DF + 33, 64, 64, 1, 1>(a, backward, stream);
  return -1;
}

