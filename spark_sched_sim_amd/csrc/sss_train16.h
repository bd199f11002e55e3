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

// the MLP shapes of the published architecture (config/decima_tpch.yaml:66-78); anything else: -1 (the caller keeps autograd)
static int be_launch_mlp(const SssMlpArgs& a, int backward, void* stream) {
  const bool gnn = a.h1 == 32 && a.h2 == 16 && a.out_dim == 16 && a.act == 0;
  const bool head = a.h1 == 64 && a.h2 == 64 && a.out_dim == 1 && a.act == 1;
  if (gnn && a.in_dim == GNN_NF) return mlp16_launch<GNN_NF, 32, 16, 16, 0>(a, backward, stream);
  if (gnn && a.in_dim == 16) return mlp16_launch<16, 32, 16, 16, 0>(a, backward, stream);
  if (gnn && a.in_dim == GNN_NF + 16) return mlp16_launch<GNN_NF + 16, 32, 16, 16, 0>(a, backward, stream);
  if (head && a.in_dim == GNN_NF + 48) return mlp16_launch<GNN_NF + 48, 64, 64, 1, 1>(a, backward, stream);
  if (head && a.in_dim == GNN_DF + 33) return mlp16_launch<GNN_DF + 33, 64, 64, 1, 1>(a, backward, stream);
  return -1;
}
