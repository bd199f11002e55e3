// sss_gnn16.h - the GNN's MLPs with SIXTEEN LANES PER ROW (gfx950 only; the emulator keeps sss_gnn.h's
// one-thread-per-row formulation, which is also what the one-launch policy kernel uses).
//
// Why: one thread pushing a row through Linear-act-Linear-act-Linear is a serial chain of 1 400 - 7 700
// dependent FMAs fed by LDS broadcasts - ~40 us per launch whether 20 k or 80 k rows are in flight
// (profiles/r02_decima.md); the nine DAG layers of a Decima step are nine such chains back to back. Here a row is
// a 16-lane DPP row: lane g owns output neurons g, g + 16, ... of every layer, an input vector lives one element
// per lane and reaches the other lanes through `v_mov_b32_dpp row_newbcast:k` - no LDS exchange, no barrier -,
// and the weights sit in LDS transposed so that the 16 lanes of a row read consecutive words (no bank conflicts)
// while the four rows of a wave read the same words (broadcast).
//
// Same arithmetic as sss_gnn.h (fp32 FMAs; sums of hidden vectors are pushed through the last Linear once); only
// the order of the additions inside a dot product differs - within the 2e-5 agreement the fixtures are held to.
#pragma once
#include <utility>

// value of lane K of this lane's 16-lane row
template <int K>
SSS_DEV float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xF, 0xF, false));
}
template <typename F, int... K>
SSS_DEV void static_for_impl(F&& f, std::integer_sequence<int, K...>) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, typename F>
SSS_DEV void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
// sum over the 16 lanes of a row, on every lane of it
SSS_DEV float row_sum(float v) {
  float t = 0.0f;
  static_for<16>([&](auto kc) { t += row_bcast<decltype(kc)::value>(v); });
  return t;
}

template <int ACT>
SSS_DEV float act16(float v, float slope) {
  if (ACT == 0) return v > 0.0f ? v : v * slope;
  return tanhf(v);
}

// LDS image of one MLP IN -> H1 -> H2 -> OUT (OUT = 16 with H2 = 16, or OUT = 1) for 16 lanes per row;
// Q1 = H1 / 16, Q2 = H2 / 16 neurons per lane. From the packed parameters [W1 (H1 x IN), b1, W2T (H1 x H2), b2,
// W3 (OUT x H2), b3] of sss_gnn.h:
//   w1[i][g][q]      = W1[g + 16 q][i]                      input i, lane g, q < Q1
//   b1[g][q]
//   w2[jj][q][g][r]  = W2T[jj + 16 q][g + 16 r]             hidden-1 neuron jj + 16 q (lane jj, register q), r < Q2
//   b2[g][r]
//   OUT = 16: w3[k][g] = W3[g][k], b3[g]    OUT = 1: w3[g][r] = W3[0][g + 16 r], b3[0]
template <int IN, int H1, int H2, int OUT, int ACT>
struct Mlp16 {
  static constexpr int Q1 = H1 / 16, Q2 = H2 / 16;
  static constexpr int W1 = 0, B1 = W1 + IN * H1, W2 = B1 + H1, B2 = W2 + H1 * H2, W3 = B2 + H2, B3 = W3 + OUT * H2, TOTAL = ((B3 + OUT + 3) / 4) * 4;
  static_assert(H1 % 16 == 0 && H2 % 16 == 0 && (OUT == 1 || (OUT == 16 && H2 == 16)), "layer widths");

  SSS_DEV static void stage(float* lds, const float* __restrict__ w, int tid, int nthreads) {
    const float* gW1 = w;
    const float* gb1 = gW1 + H1 * IN;
    const float* gW2T = gb1 + H1;
    const float* gb2 = gW2T + H1 * H2;
    const float* gW3 = gb2 + H2;
    const float* gb3 = gW3 + OUT * H2;
    for (int t = tid; t < IN * H1; t += nthreads) {
      const int q = t % Q1, g = (t / Q1) % 16, i = t / (Q1 * 16);
      lds[W1 + t] = gW1[(g + 16 * q) * IN + i];
    }
    for (int t = tid; t < H1; t += nthreads) lds[B1 + t] = gb1[(t / Q1) + 16 * (t % Q1)];
    for (int t = tid; t < H1 * H2; t += nthreads) {
      const int r = t % Q2, g = (t / Q2) % 16, q = (t / (Q2 * 16)) % Q1, jj = t / (Q2 * 16 * Q1);
      lds[W2 + t] = gW2T[(jj + 16 * q) * H2 + g + 16 * r];
    }
    for (int t = tid; t < H2; t += nthreads) lds[B2 + t] = gb2[(t / Q2) + 16 * (t % Q2)];
    if (OUT == 16) {
      for (int t = tid; t < 256; t += nthreads) lds[W3 + t] = gW3[(t & 15) * 16 + (t >> 4)];
      for (int t = tid; t < 16; t += nthreads) lds[B3 + t] = gb3[t];
    } else {
      for (int t = tid; t < H2; t += nthreads) lds[W3 + t] = gW3[(t / Q2) + 16 * (t % Q2)];
      if (tid == 0) lds[B3] = gb3[0];
    }
  }

  // first Linear, one input segment: inputs BASE .. BASE + LEN - 1 are elements 0 .. LEN - 1 of the row vector whose
  // element g this lane holds in `seg`
  template <int BASE, int LEN>
  SSS_DEV static void l1(const float* m, float (&a)[Q1], float seg, int g) {
    _Pragma("clang fp contract(fast)")
    static_for<LEN>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const float xk = row_bcast<k>(seg);
      const float* wp = m + W1 + ((BASE + k) * 16 + g) * Q1;
      static_for<Q1>([&](auto qc) { a[decltype(qc)::value] += wp[decltype(qc)::value] * xk; });
    });
  }
  SSS_DEV static void l1_bias(const float* m, float (&a)[Q1], int g) {
    static_for<Q1>([&](auto qc) { a[decltype(qc)::value] = m[B1 + g * Q1 + decltype(qc)::value]; });
  }
  // activation of hidden 1, second Linear, activation: hidden-2 neurons g + 16 r of the row
  SSS_DEV static void l2(const float* m, float (&a)[Q1], float (&h)[Q2], int g, float slope) {
    _Pragma("clang fp contract(fast)")
    static_for<Q1>([&](auto qc) { a[decltype(qc)::value] = act16<ACT>(a[decltype(qc)::value], slope); });
    static_for<Q2>([&](auto rc) { h[decltype(rc)::value] = m[B2 + g * Q2 + decltype(rc)::value]; });
    static_for<16>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      static_for<Q1>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float tk = row_bcast<k>(a[q]);
        const float* wp = m + W2 + ((k * Q1 + q) * 16 + g) * Q2;
        static_for<Q2>([&](auto rc) { h[decltype(rc)::value] += wp[decltype(rc)::value] * tk; });
      });
    });
    static_for<Q2>([&](auto rc) { h[decltype(rc)::value] = act16<ACT>(h[decltype(rc)::value], slope); });
  }
  // last Linear(16,16) applied to (a sum of `count`) hidden-2 vector(s): element g of W3 . v + count * b3
  SSS_DEV static float out16(const float* m, float v, int g, float count) {
    _Pragma("clang fp contract(fast)")
    float o0 = m[B3 + g] * count, o1 = 0.0f;
    static_for<8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      o0 += m[W3 + (2 * k) * 16 + g] * row_bcast<2 * k>(v);
      o1 += m[W3 + (2 * k + 1) * 16 + g] * row_bcast<2 * k + 1>(v);
    });
    return o0 + o1;
  }
  // last Linear(H2,1): the row's score, on every lane of the row
  SSS_DEV static float out1(const float* m, const float (&h)[Q2], int g) {
    _Pragma("clang fp contract(fast)")
    float p = 0.0f;
    static_for<Q2>([&](auto rc) { p += m[W3 + g * Q2 + decltype(rc)::value] * h[decltype(rc)::value]; });
    return row_sum(p) + m[B3];
  }
};

using MlpGnn = Mlp16<16, 32, 16, 16, 0>;    // message / update / SINK / GLOBSUM-side MLPs: 16 -> 32 -> 16 -> 16
using MlpPrep = Mlp16<GNN_NF, 32, 16, 16, 0>;
using MlpDag = Mlp16<GNN_NF + 16, 32, 16, 16, 0>;
using MlpStage = Mlp16<GNN_NF + 48, 64, 64, 1, 1>;
using MlpExec = Mlp16<GNN_DF + 33, 64, 64, 1, 1>;

template <typename M>
SSS_DEV float hidden16_of(const float* m, float x, int g, float slope) {  // 16 inputs, one per lane -> hidden-2 neuron g
  float a[M::Q1], h[M::Q2];
  M::l1_bias(m, a, g);
  M::template l1<0, 16>(m, a, x, g);
  M::l2(m, a, h, g, slope);
  return h[0];
}

// the pieces of a layer's list (sss_gnn.h list_q): lane s < 32 holds piece s - its length, its number of 16-row tiles, the running
// tile count up to and including it, its first row in idx0; `total` = tiles of the layer
struct GnnListPieces {
  int64_t len, start;
  int tiles, incl, total;
  SSS_DEV void load(const SssGnnArgs& a, int lane) {
    const int n_sets = (a.n_seg + a.list_q - 1) / a.list_q;
    const bool mine = lane < n_sets && lane < SSS_LIST_SETS;
    len = mine ? a.layer_totals[(int64_t)a.layer * SSS_LIST_SETS + lane] : 0;
    start = (int64_t)a.layer * a.idx0_stride + (mine ? a.seg_off[(int64_t)lane * a.list_q] : 0);
    tiles = (int)((len + 15) >> 4), incl = tiles;
    for (int s = 1; s < 32; s <<= 1) {
      const int t = __shfl_up(incl, s);
      if ((lane & 31) >= s) incl += t;
    }
    total = __shfl(incl, 31);
  }
  // the same for a tile number the whole wave agrees on: one ballot, the piece's entries through scalar registers
  SSS_DEV void tile_uniform(int t, int lane, int64_t& first, int& count) const {
    const int s = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < SSS_LIST_SETS && incl <= t));  // pieces that end before tile t
    const int local = t - (__builtin_amdgcn_readlane(incl, s) - __builtin_amdgcn_readlane(tiles, s));
    const int64_t len_s = ((int64_t)__builtin_amdgcn_readlane((int)(len >> 32), s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)len, s);
    const int64_t start_s = ((int64_t)__builtin_amdgcn_readlane((int)(start >> 32), s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)start, s);
    const int64_t left = len_s - 16 * (int64_t)local;
    first = start_s + 16 * (int64_t)local, count = left < 16 ? (int)left : 16;
  }
  // tile t of the layer: its first row in idx0 and its number of rows (any lane may ask for any tile)
  SSS_DEV void tile(int t, int64_t& first, int& count) const {
    int s = 0;
    for (int k = 0; k < SSS_LIST_SETS; k++) s += __shfl(incl, k) <= t ? 1 : 0;  // pieces that end before tile t
    const int local = t - (__shfl(incl, s) - __shfl(tiles, s));
    const int64_t left = __shfl(len, s) - 16 * (int64_t)local;
    first = __shfl(start, s) + 16 * (int64_t)local, count = left < 16 ? (int)left : 16;
  }
};

// rows of a launch: 16 per workgroup of 256 threads at a time, looping; parameters staged once per workgroup
#define GNN16_ROWS(r) for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4), r##_end = a.n_rows_dev ? *a.n_rows_dev : a.n_rows; r < r##_end; r += (int64_t)gridDim.x * 16)

template <int KIND>
__global__ __launch_bounds__(256) void sss_gnn16_kernel(SssGnnArgs a) {
  constexpr int F = GNN_EMB;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  const int g = threadIdx.x & 15;
  if (KIND == GNN_LAYER) {
    // tmp[r] = h_init[r] + update( sum over r's out-edges e in DAG layer l of msg(h[dst_e]) )
    GnnListPieces pc;  // the graph kernel's lists (sss_gnn.h list_q): a dense piece per block of observations, walked tile by tile
    if (a.list_q) {
      pc.load(a, threadIdx.x & 63);
      a.n_rows = 16 * (int64_t)pc.total;
      if ((int64_t)blockIdx.x * 16 >= a.n_rows) return;
    } else if (a.layer_totals) {  // list length and position from the device (sss_gnn_encode): most workgroups of a sparse layer leave here
      int64_t off = (int64_t)a.layer * a.idx0_stride;
      if (a.idx0_stride == 0)
        for (int l = 0; l < a.layer; l++) off += a.layer_totals[l];
      a.n_rows = a.layer_totals[a.layer], a.idx0 += off;
      if ((int64_t)blockIdx.x * 16 >= a.n_rows) return;
    }
    if (a.w16 && a.w2_16) {  // the host has the images ready: a straight copy, 16 bytes per thread and pass
      static_assert(MlpGnn::TOTAL % 4 == 0, "images are copied 16 bytes at a time");
      for (int t = threadIdx.x; t < MlpGnn::TOTAL / 4; t += 256) {
        ((float4*)w_lds)[t] = ((const float4*)a.w16)[t];
        ((float4*)(w_lds + MlpGnn::TOTAL))[t] = ((const float4*)a.w2_16)[t];
      }
    } else {
      MlpGnn::stage(w_lds, a.w, threadIdx.x, 256);
      MlpGnn::stage(w_lds + MlpGnn::TOTAL, a.w2, threadIdx.x, 256);
    }
    __syncthreads();
    const float* msg = w_lds;
    const float* upd = w_lds + MlpGnn::TOTAL;
    GNN16_ROWS(r) {
      int64_t n;
      if (a.list_q) {
        int64_t first;
        int count;
        pc.tile((int)(r >> 4), first, count);
        n = (int)(r & 15) < count ? a.idx0[first + (r & 15)] : -1;
      } else
        n = a.idx0[r];
      if (n < 0) continue;  // (whole rows: the 16 lanes of a row always agree)
      const int64_t e0 = a.out_start[n];
      const int deg = a.out_deg[n];
      float acc = 0.0f;
      int used = 0;
      // with node_recv: embeddings alternate between `h` and `tmp` per update (sss_gnn.h), no COMMIT launch
      const uint32_t above = a.layer >= 31 ? 0u : ~((2u << a.layer) - 1u);
      float* nxt = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[n] & above) & 1)) ? a.h : a.tmp;
      for (int k = 0; k < deg; k++) {
        if (!((a.edge_layers[e0 + k] >> a.layer) & 1u)) continue;
        const int64_t c = a.dst[e0 + k];
        const float* cur = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[c] & above) & 1)) ? a.tmp : a.h;
        acc += hidden16_of<MlpGnn>(msg, cur[c * F + g], g, a.slope);
        used++;
      }
      const float agg = MlpGnn::out16(msg, acc, g, (float)used);
      const float h2 = hidden16_of<MlpGnn>(upd, agg, g, a.slope);
      nxt[n * F + g] = a.h_init[n * F + g] + MlpGnn::out16(upd, h2, g, 1.0f);
    }
  } else if (KIND == GNN_PREP) {
    MlpPrep::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) {
      float aa[MlpPrep::Q1], hh[MlpPrep::Q2];
      MlpPrep::l1_bias(w_lds, aa, g);
      MlpPrep::template l1<0, GNN_NF>(w_lds, aa, g < GNN_NF ? a.x[r * GNN_NF + g] : 0.0f, g);
      MlpPrep::l2(w_lds, aa, hh, g, a.slope);
      a.out[r * F + g] = MlpPrep::out16(w_lds, hh[0], g, 1.0f);
    }
  } else if (KIND == GNN_SINK) {
    MlpGnn::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) {
      const float x = a.h_init[r * F + g];
      const bool par = a.out_deg[r] != 0;
      const bool skip = a.obs_depth != nullptr && a.obs_depth[a.node_obs[r]] == 0;  // single-layer observation: mlp_prep only
      if (skip || par) {
        a.h[r * F + g] = skip ? x : 0.0f;
        continue;
      }
      a.h[r * F + g] = MlpGnn::out16(w_lds, hidden16_of<MlpGnn>(w_lds, x, g, a.slope), g, 1.0f);
    }
  } else if (KIND == GNN_DAGHID) {
    MlpDag::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) {
      float aa[MlpDag::Q1], hh[MlpDag::Q2];
      MlpDag::l1_bias(w_lds, aa, g);
      MlpDag::template l1<0, GNN_NF>(w_lds, aa, g < GNN_NF ? a.x[r * GNN_NF + g] : 0.0f, g);
      float hv = a.h[r * F + g];
      if (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[r]) & 1)) hv = a.tmp[r * F + g], a.h[r * F + g] = hv;  // MERGE on the fly (sss_gnn.h)
      MlpDag::template l1<GNN_NF, 16>(w_lds, aa, hv, g);
      MlpDag::l2(w_lds, aa, hh, g, a.slope);
      a.tmp[r * 16 + g] = hh[0];
    }
  } else if (KIND == GNN_GLOBHID) {
    MlpGnn::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) { a.tmp[r * 16 + g] = hidden16_of<MlpGnn>(w_lds, a.h_dag[r * F + g], g, a.slope); }
  } else if (KIND == GNN_STAGE) {
    if (a.w16) {
      static_assert(MlpStage::TOTAL % 4 == 0, "images are copied 16 bytes at a time");
      for (int t = threadIdx.x; t < MlpStage::TOTAL / 4; t += 256) ((float4*)w_lds)[t] = ((const float4*)a.w16)[t];
    } else
      MlpStage::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) {
      const int64_t n = a.idx0[r];
      if (n < 0) break;  // the list is the schedulable nodes followed by padding: nothing but padding from here on
      float aa[MlpStage::Q1], hh[MlpStage::Q2];
      MlpStage::l1_bias(w_lds, aa, g);
      MlpStage::template l1<0, GNN_NF>(w_lds, aa, g < GNN_NF ? a.x[n * GNN_NF + g] : 0.0f, g);
      MlpStage::template l1<GNN_NF, 16>(w_lds, aa, a.h[n * F + g], g);
      MlpStage::template l1<GNN_NF + 16, 16>(w_lds, aa, a.h_dag[a.node_job[n] * F + g], g);
      MlpStage::template l1<GNN_NF + 32, 16>(w_lds, aa, a.h_glob[a.node_obs[n] * F + g], g);
      MlpStage::l2(w_lds, aa, hh, g, 0.0f);
      const float v = MlpStage::out1(w_lds, hh, g);
      if (g == 0) a.out[a.node_obs[n] * a.n_pad + a.node_loc[n]] = v;
    }
  } else if (KIND == GNN_EXEC) {
    if (a.w16) {
      static_assert(MlpExec::TOTAL % 4 == 0, "images are copied 16 bytes at a time");
      for (int t = threadIdx.x; t < MlpExec::TOTAL / 4; t += 256) ((float4*)w_lds)[t] = ((const float4*)a.w16)[t];
    } else
      MlpExec::stage(w_lds, a.w, threadIdx.x, 256);
    __syncthreads();
    GNN16_ROWS(r) {
      const int64_t b = r / a.E;
      const int c = (int)(r - b * a.E);
      const int64_t j = a.idx0[b];
      float aa[MlpExec::Q1], hh[MlpExec::Q2];
      MlpExec::l1_bias(w_lds, aa, g);
      MlpExec::template l1<0, GNN_DF>(w_lds, aa, g < GNN_DF ? a.x[a.job_first[j] * GNN_NF + g] : 0.0f, g);
      MlpExec::template l1<GNN_DF, 16>(w_lds, aa, a.h_dag[j * F + g], g);
      MlpExec::template l1<GNN_DF + 16, 16>(w_lds, aa, a.h_glob[a.job_obs[j] * F + g], g);
      MlpExec::template l1<GNN_DF + 32, 1>(w_lds, aa, (float)c / (float)a.E, g);
      MlpExec::l2(w_lds, aa, hh, g, 0.0f);
      const float v = MlpExec::out1(w_lds, hh, g);
      if (g == 0) a.out[r] = c < a.job_cap[j] ? v : -__builtin_inff();
    }
  }
}

// DAGSUM / GLOBSUM with 16 lanes per row: lane g sums feature g of the segment's hidden vectors (the rows of a job's nodes / an
// observation's jobs are contiguous: each step reads one 64-byte row) in the order of sss_gnn.h's loop, then takes output g of
// the last Linear with gnn_dot's four accumulators - the additions in the same order as the one-thread-per-row form (whose
// compiled multiply-adds may be fused differently: last-bit differences), eight rows of the segment in flight per step.
template <int KIND>
__global__ __launch_bounds__(256) void sss_gnn_sum16_kernel(SssGnnArgs a) {
  constexpr int IN = KIND == GNN_DAGSUM ? GNN_NF + 16 : 16;
  const int g = threadIdx.x & 15;
  const float* W3 = a.w + 32 * IN + 32 + 16 * 32 + 16;
  float w[16];
  static_for<16>([&](auto ic) { w[decltype(ic)::value] = W3[g * 16 + decltype(ic)::value]; });
  const float b3 = W3[256 + g];
  const int64_t rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  float* out = KIND == GNN_DAGSUM ? a.h_dag : a.h_glob;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
    const int64_t first = KIND == GNN_DAGSUM ? a.job_first[r] : a.obs_job_off[r], cnt = KIND == GNN_DAGSUM ? a.job_nodes[r] : a.obs_jobs[r];
    float acc = 0.0f;
    const float* __restrict__ src = a.tmp + first * 16 + g;
    int64_t n = 0;
    for (; n + 8 <= cnt; n += 8) {  // eight rows in flight, added in order
      float t[8];
      static_for<8>([&](auto uc) { t[decltype(uc)::value] = src[(n + decltype(uc)::value) * 16]; });
      static_for<8>([&](auto uc) { acc += t[decltype(uc)::value]; });
    }
    {
      float t[8];
      static_for<8>([&](auto uc) { t[decltype(uc)::value] = n + decltype(uc)::value < cnt ? src[(n + decltype(uc)::value) * 16] : 0.0f; });
      static_for<8>([&](auto uc) { if (n + decltype(uc)::value < cnt) acc += t[decltype(uc)::value]; });
    }
    float a0 = b3 * (float)cnt, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    static_for<4>([&](auto kc) {
      constexpr int i = 4 * decltype(kc)::value;
      a0 = __builtin_fmaf(w[i], row_bcast<i>(acc), a0), a1 = __builtin_fmaf(w[i + 1], row_bcast<i + 1>(acc), a1);
      a2 = __builtin_fmaf(w[i + 2], row_bcast<i + 2>(acc), a2), a3 = __builtin_fmaf(w[i + 3], row_bcast<i + 3>(acc), a3);
    });
    out[r * 16 + g] = (a0 + a1) + (a2 + a3);
  }
}
template <int KIND>
static int gnn_sum16_launch(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  const int64_t wgs = (a.n_rows + 15) / 16;
  hipLaunchKernelGGL(sss_gnn_sum16_kernel<KIND>, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

template <int KIND>
constexpr int gnn16_lds_floats() {
  return KIND == GNN_LAYER ? 2 * MlpGnn::TOTAL : KIND == GNN_PREP ? MlpPrep::TOTAL : KIND == GNN_DAGHID ? MlpDag::TOTAL
       : KIND == GNN_STAGE ? MlpStage::TOTAL : KIND == GNN_EXEC ? MlpExec::TOTAL : MlpGnn::TOTAL;
}
template <int KIND>
static int gnn16_launch(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  // a workgroup stages its MLP(s) once (up to 31 KB for the policy heads) and then loops over row tiles: enough
  // workgroups to fill the 256 CUs a few times over, not one per tile
  const int64_t tiles = (a.n_rows + 15) / 16;
  const int64_t cap = KIND == GNN_LAYER ? 2048 : 768;
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
  hipLaunchKernelGGL(sss_gnn16_kernel<KIND>, dim3(grid), dim3(256), gnn16_lds_floats<KIND>() * sizeof(float), (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
