// Weight-gradient engines shared by the backward kernels of both mappings (gfx950).
// dW = sum over rows (one row = one bus or one line of one grid) of g (x) input is the one dense contraction of the
// path: a lane writes its row record to LDS and the wave contracts 64 (or 32) records at a time - on the fp32 matrix
// pipe (v_mfma_f32_16x16x4_f32, exact fp32) or with packed-FMA register tiles.
#pragma once
#include "gns_device.h"

// ---- LDS record of one row (one bus or one line of one grid) for one LearningBlock ---------------------
template <int IN, int H, int OUT>
struct RecLay {
  static constexpr int XP = (IN + 1 + 3) / 4 * 4;   // x | 1 | 0..      (the 1 yields the bias gradient)
  static constexpr int HP = (H + 1 + 3) / 4 * 4;    // a | 1 | 0..  and  g | 0..
  static constexpr int GP = (OUT + 3) / 4 * 4;      // g3 | 0..
  static constexpr int oX = 0, oA1 = XP, oA2 = oA1 + HP, oG1 = oA2 + HP, oG2 = oG1 + HP, oG3 = oG2 + HP;
  static constexpr int raw = oG3 + GP;
  static constexpr int RS = ((raw / 4) | 1) * 4;    // RS/4 odd: 16-byte row writes of 8 lanes hit 8 different bank quads
  static constexpr int TX = XP / 4, TH = HP / 4, TG = GP / 4;
  static constexpr int T1 = TH * TX, T2 = TH * TH, T4 = TG * TH, NT = T1 + T2 + T4;   // 4x4 tiles of dW1|db1, dW2|db2, dW4|db4
  static_assert(NT <= 64, "one pass per network");
};
#define GNS_REC_ROWS 32
constexpr int gns_cmax(int a, int b) { return a > b ? a : b; }

template <int IN, int H, int OUT, int OUTP>
__device__ __forceinline__ void rec_write(float* rec, int row, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2], const f2 (&a2)[H / 2],
                                          const f2 (&g1)[H / 2], const f2 (&g2)[H / 2], const f2 (&g3)[OUTP / 2]) {
  using R = RecLay<IN, H, OUT>;
  // 8-byte stores straight from the aligned register pairs the MLPs work on (16-byte stores would first need
  // four v_mov per store to build a register quad)
  f2* dst = reinterpret_cast<f2*>(rec + row * R::RS);
  static_for<0, R::raw / 2>([&](auto q_) {
    constexpr int q = decltype(q_)::value;
    constexpr int o = 2 * q;               // even element index; fields start at multiples of 4
    f2 e;
    if constexpr (o < R::oA1) {
      if constexpr (o + 1 < IN) e = x[q];
      else if constexpr (o < IN) e = f2{x[q].x, 1.f};            // IN odd: last input then the 1 of the bias column
      else e = (o == IN) ? f2{1.f, 0.f} : f2{0.f, 0.f};
    } else if constexpr (o < R::oA2) {
      constexpr int i = o - R::oA1;
      if constexpr (i < H) e = a1[i / 2]; else e = (i == H) ? f2{1.f, 0.f} : f2{0.f, 0.f};
    } else if constexpr (o < R::oG1) {
      constexpr int i = o - R::oA2;
      if constexpr (i < H) e = a2[i / 2]; else e = (i == H) ? f2{1.f, 0.f} : f2{0.f, 0.f};
    } else if constexpr (o < R::oG2) {
      constexpr int i = o - R::oG1;
      if constexpr (i < H) e = g1[i / 2]; else e = f2{0.f, 0.f};
    } else if constexpr (o < R::oG3) {
      constexpr int i = o - R::oG2;
      if constexpr (i < H) e = g2[i / 2]; else e = f2{0.f, 0.f};
    } else {
      constexpr int i = o - R::oG3;
      if constexpr (i + 1 < OUT) e = g3[i / 2];
      else if constexpr (i < OUT) e = f2{g3[i / 2].x, 0.f};
      else e = f2{0.f, 0.f};
    }
    dst[q] = e;
  });
}

// ---- phi' (two layers): record [x|1, a1|1, g1, g2], tiles of dW1|db1 and dW2|db2 -----------------------------------
template <int IN, int H>
struct RecLay2 {
  static constexpr int XP = (IN + 1 + 3) / 4 * 4, HP = (H + 1 + 3) / 4 * 4;
  static constexpr int oX = 0, oA1 = XP, oG1 = oA1 + HP, oG2 = oG1 + HP;
  static constexpr int raw = oG2 + HP;
  static constexpr int RS = ((raw / 4) | 1) * 4;
  static constexpr int TX = XP / 4, TH = HP / 4;
  static constexpr int T1 = TH * TX, T2 = TH * TH, NT = T1 + T2;
  static_assert(NT <= 64, "one pass per network");
};

template <int IN, int H>
__device__ __forceinline__ void rec_write2(float* rec, int row, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2],
                                           const f2 (&g1)[H / 2], const f2 (&g2)[H / 2]) {
  using R = RecLay2<IN, H>;
  f2* dst = reinterpret_cast<f2*>(rec + row * R::RS);
  static_for<0, R::raw / 2>([&](auto q_) {
    constexpr int q = decltype(q_)::value;
    constexpr int o = 2 * q;
    f2 e;
    if constexpr (o < R::oA1) {
      if constexpr (o + 1 < IN) e = x[q];
      else if constexpr (o < IN) e = f2{x[q].x, 1.f};
      else e = (o == IN) ? f2{1.f, 0.f} : f2{0.f, 0.f};
    } else if constexpr (o < R::oG1) {
      constexpr int i = o - R::oA1;
      if constexpr (i < H) e = a1[i / 2]; else e = (i == H) ? f2{1.f, 0.f} : f2{0.f, 0.f};
    } else if constexpr (o < R::oG2) {
      constexpr int i = o - R::oG1;
      if constexpr (i < H) e = g1[i / 2]; else e = f2{0.f, 0.f};
    } else {
      constexpr int i = o - R::oG2;
      if constexpr (i < H) e = g2[i / 2]; else e = f2{0.f, 0.f};
    }
    dst[q] = e;
  });
}

// Weight gradient of one LearningBlock for the 64 grids of this wave: dW += sum_grids g (x) input.
// Lane t < NT owns the 4x4 tile t of [dW1|db1], [dW2|db2] or [dW4|db4]; the tile lives in registers across all
// the rows (buses / lines) a wave handles in one reverse step and is flushed once, into the wave's slab in the
// flat layout W1[H][IN] b1[H] W2[H][H] b2[H] W4[OUT][H] b4[OUT].
struct DwTile { int kind, cb, ib, woff, uoff; };

template <int IN, int H, int OUT>
__device__ __forceinline__ DwTile dw_tile(int lane) {
  using R = RecLay<IN, H, OUT>;
  DwTile T;
  int t = lane < R::NT ? lane : 0;
  if (t < R::T1) { T.kind = 0; T.cb = t / R::TX; T.ib = t % R::TX; T.woff = R::oG1 + 4 * T.cb; T.uoff = R::oX + 4 * T.ib; }
  else if (t < R::T1 + R::T2) { t -= R::T1; T.kind = 1; T.cb = t / R::TH; T.ib = t % R::TH; T.woff = R::oG2 + 4 * T.cb; T.uoff = R::oA1 + 4 * T.ib; }
  else { t -= R::T1 + R::T2; T.kind = 2; T.cb = t / R::TH; T.ib = t % R::TH; T.woff = R::oG3 + 4 * T.cb; T.uoff = R::oA2 + 4 * T.ib; }
  return T;
}

// one half-wave of records is in LDS: every lane adds 32 rank-1 updates to its 4x4 tile
template <int RS>
__device__ __forceinline__ void dw_sweep(const float* rec, const DwTile& T, f2 (&acc)[4][2]) {
  const float* rw = rec + T.woff;
  const float* ru = rec + T.uoff;
#ifndef GNS_ABLATE_ENGINE
#pragma unroll 4
  for (int r = 0; r < GNS_REC_ROWS; ++r) {
    const f4 w = *reinterpret_cast<const f4*>(rw + r * RS);
    const f4 u = *reinterpret_cast<const f4*>(ru + r * RS);
    const f2 u0 = f2{u.x, u.y}, u1 = f2{u.z, u.w};
    acc[0][0] = __builtin_elementwise_fma(splat(w.x), u0, acc[0][0]); acc[0][1] = __builtin_elementwise_fma(splat(w.x), u1, acc[0][1]);
    acc[1][0] = __builtin_elementwise_fma(splat(w.y), u0, acc[1][0]); acc[1][1] = __builtin_elementwise_fma(splat(w.y), u1, acc[1][1]);
    acc[2][0] = __builtin_elementwise_fma(splat(w.z), u0, acc[2][0]); acc[2][1] = __builtin_elementwise_fma(splat(w.z), u1, acc[2][1]);
    acc[3][0] = __builtin_elementwise_fma(splat(w.w), u0, acc[3][0]); acc[3][1] = __builtin_elementwise_fma(splat(w.w), u1, acc[3][1]);
  }
#else
  acc[0][0] += f2{rw[0], ru[0]};
#endif
}

template <int IN, int H, int OUT, int OUTP>
__device__ __forceinline__ void dw_accumulate(float* rec, int lane, const DwTile& T, f2 (&acc)[4][2], const f2 (&x)[(IN + 1) / 2],
                                              const f2 (&a1)[H / 2], const f2 (&a2)[H / 2], const f2 (&g1)[H / 2],
                                              const f2 (&g2)[H / 2], const f2 (&g3)[OUTP / 2]) {
  using R = RecLay<IN, H, OUT>;
#ifdef GNS_ABLATE_DW
  asm volatile("" :: "v"(x[0]), "v"(a1[0]), "v"(a2[0]), "v"(g1[0]), "v"(g2[0]), "v"(g3[0]));
  return;
#endif
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#ifndef GNS_ABLATE_RECWRITE
    if ((lane >> 5) == half) rec_write<IN, H, OUT, OUTP>(rec, lane & 31, x, a1, a2, g1, g2, g3);
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dw_sweep<R::RS>(rec, T, acc);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// phi' needs only 30 tiles, so the two half-waves take the same tiles and sweep one half of the 64 records each:
// all 64 lanes write their record at once (no masked halves) and the sweep is 32 rows instead of 64.
template <int IN, int H>
__device__ __forceinline__ DwTile dw_tile2(int lane) {
  using R = RecLay2<IN, H>;
  static_assert(R::NT <= 32, "two half-waves share the tile set");
  DwTile T;
  int t = (lane & 31) < R::NT ? (lane & 31) : 0;
  if (t < R::T1) { T.kind = 0; T.cb = t / R::TX; T.ib = t % R::TX; T.woff = R::oG1 + 4 * T.cb; T.uoff = R::oX + 4 * T.ib; }
  else { t -= R::T1; T.kind = 1; T.cb = t / R::TH; T.ib = t % R::TH; T.woff = R::oG2 + 4 * T.cb; T.uoff = R::oA1 + 4 * T.ib; }
  const int half_off = (lane >> 5) * GNS_REC_ROWS * R::RS;      // rows 32..63 for the upper half-wave
  T.woff += half_off; T.uoff += half_off;
  return T;
}

template <int IN, int H>
__device__ __forceinline__ void dw_accumulate2(float* rec, int lane, const DwTile& T, f2 (&acc)[4][2], const f2 (&x)[(IN + 1) / 2],
                                               const f2 (&a1)[H / 2], const f2 (&g1)[H / 2], const f2 (&g2)[H / 2]) {
  using R = RecLay2<IN, H>;
#ifdef GNS_ABLATE_DW
  asm volatile("" :: "v"(x[0]), "v"(a1[0]), "v"(g1[0]), "v"(g2[0]));
  return;
#endif
#ifndef GNS_ABLATE_RECWRITE
  rec_write2<IN, H>(rec, lane, x, a1, g1, g2);
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  dw_sweep<R::RS>(rec, T, acc);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// phi' tile -> folded-gradient block W1[H][IN] b1[H] W2[H][H] b2[H]
template <int IN, int H>
__device__ __forceinline__ void dw_flush2(int lane, const DwTile& T, const f2 (&acc)[4][2], float* slab_blk) {
  using R = RecLay2<IN, H>;
  float tot[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const float v = (bq & 1) ? acc[a][bq >> 1].y : acc[a][bq >> 1].x;
      tot[a][bq] = v + __shfl_xor(v, 32);                        // rows 0..31 + rows 32..63
    }
  if (lane < R::NT) {
    constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int c = 4 * T.cb + a;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        const int i = 4 * T.ib + bq;
        const float val = tot[a][bq];
        int idx = -1;
        if (T.kind == 0) { if (c < H) idx = (i < IN) ? c * IN + i : (i == IN ? ob1 + c : -1); }
        else { if (c < H) idx = (i < H) ? oW2 + c * H + i : (i == H ? ob2 + c : -1); }
        if (idx >= 0) slab_blk[idx] += val;
      }
    }
  }
}

template <int IN, int H, int OUT>
__device__ __forceinline__ void dw_flush(int lane, const DwTile& T, const f2 (&acc)[4][2], float* slab_blk) {
  using R = RecLay<IN, H, OUT>;
  if (lane < R::NT) {
    constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int c = 4 * T.cb + a;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        const int i = 4 * T.ib + bq;
        const float val = (bq & 1) ? acc[a][bq >> 1].y : acc[a][bq >> 1].x;
        int idx = -1;
        if (T.kind == 0) { if (c < H) idx = (i < IN) ? c * IN + i : (i == IN ? ob1 + c : -1); }
        else if (T.kind == 1) { if (c < H) idx = (i < H) ? oW2 + c * H + i : (i == H ? ob2 + c : -1); }
        else { if (c < OUT) idx = (i < H) ? oW4 + c * H + i : (i == H ? ob4 + c : -1); }
        if (idx >= 0) slab_blk[idx] += val;
      }
    }
  }
}

__device__ __forceinline__ void zero_acc(f2 (&acc)[4][2]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc[a][0] = f2{0.f, 0.f}; acc[a][1] = f2{0.f, 0.f}; }
}

// ---- weight-gradient engines: one object per (network, family sweep) ----------------------------------------------
// dW = sum over the wave's 64 grids (and its buses / lines) of g (x) input is the one dense contraction of the path.
// Two interchangeable engines read the same LDS records: MFMA = false keeps 4x4 tiles in registers and uses packed
// FMAs (GNS_DW_MFMA=0); MFMA = true (default) puts the contraction on the otherwise idle matrix pipe.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int IN, int H, int OUT, int OUTP, bool MFMA>
struct LEngine;    // three-layer block (L')
template <int IN, int H, bool MFMA>
struct PEngine;    // two-layer block (phi')

template <int IN, int H, int OUT, int OUTP>
struct LEngine<IN, H, OUT, OUTP, false> {
  DwTile T;
  f2 acc[4][2];
  __device__ __forceinline__ void init(int lane) { T = dw_tile<IN, H, OUT>(lane); zero_acc(acc); }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2], const f2 (&a2)[H / 2],
                                             const f2 (&g1)[H / 2], const f2 (&g2)[H / 2], const f2 (&g3)[OUTP / 2]) {
    dw_accumulate<IN, H, OUT, OUTP>(rec, lane, T, acc, x, a1, a2, g1, g2, g3);
  }
  __device__ __forceinline__ void flush(int lane, float* slab_blk) { dw_flush<IN, H, OUT>(lane, T, acc, slab_blk); }
};
template <int IN, int H>
struct PEngine<IN, H, false> {
  DwTile T;
  f2 acc[4][2];
  __device__ __forceinline__ void init(int lane) { T = dw_tile2<IN, H>(lane); zero_acc(acc); }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2],
                                             const f2 (&g1)[H / 2], const f2 (&g2)[H / 2]) {
    dw_accumulate2<IN, H>(rec, lane, T, acc, x, a1, g1, g2);
  }
  __device__ __forceinline__ void flush(int lane, float* slab_blk) { dw_flush2<IN, H>(lane, T, acc, slab_blk); }
};

// Matrix-pipe engine: v_mfma_f32_16x16x4_f32 is exact fp32 (no reduced-precision inputs); its k index runs over the
// grids (rows of the LDS record buffer).  Lane l feeds A[c = l&15][k = l>>4] = g[row 4kk + (l>>4)][cbase + (l&15)] and
// B[k][i = l&15] = input[row][ibase + (l&15)], and holds D[c = 4*(l>>4) + reg][i = l&15].  Columns past a field's
// width read the neighbouring field: those products only reach D entries that the flush ignores.  The MLPs themselves
// stay explicit FMA loops; measured -9..-24 % backward time against the register tiles.
template <int IN, int H, int OUT, int OUTP>
struct LEngine<IN, H, OUT, OUTP, true> {
  using R = RecLay<IN, H, OUT>;
  static constexpr int NB1 = (R::XP + 15) / 16, NA4 = (R::GP + 15) / 16, NTL = NB1 + 1 + NA4;
  f32x4 Dacc[NTL];
  __device__ __forceinline__ void init(int) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) Dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2], const f2 (&a2)[H / 2],
                                             const f2 (&g1)[H / 2], const f2 (&g2)[H / 2], const f2 (&g3)[OUTP / 2]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half) rec_write<IN, H, OUT, OUTP>(rec, lane & 31, x, a1, a2, g1, g2, g3);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // k-slot q = l>>4 of step kk reads row 8(kk>>1) + 2(kk&1) + (q>>1) + 4(q&1): the two 16-lane groups the LDS serves
      // together are 4 rows apart, and 4 RS = 16 (mod 32 banks) for every record stride (RS = 4 mod 8), so their
      // 16-column windows never share a bank (rows q, q+1 did: SQ_LDS_BANK_CONFLICT was 32 % of the LDS cycles)
      const float* base = rec + (((lane >> 5) & 1) + 4 * ((lane >> 4) & 1)) * R::RS + (lane & 15);
      // operands of step kk+1 are read while the matrix pipe works on step kk
      constexpr int NOP = 4 + NB1 + NA4;
      float op[2][NOP];
      auto fetch = [&](float (&o)[NOP], int kk) {
        const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * R::RS;
        o[0] = b[R::oG1]; o[1] = b[R::oG2]; o[2] = b[R::oA1]; o[3] = b[R::oA2];
        static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value; o[4 + t] = b[R::oX + 16 * t]; });
        static_for<0, NA4>([&](auto t_) { constexpr int t = decltype(t_)::value; o[4 + NB1 + t] = b[R::oG3 + 16 * t]; });
      };
      fetch(op[0], 0);
      static_for<0, GNS_REC_ROWS / 4>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        if constexpr (kk + 1 < GNS_REC_ROWS / 4) fetch(op[(kk + 1) & 1], kk + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float (&o)[NOP] = op[kk & 1];
        static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value; Dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[0], o[4 + t], Dacc[t], 0, 0, 0); });
        Dacc[NB1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[1], o[2], Dacc[NB1], 0, 0, 0);
        static_for<0, NA4>([&](auto t_) { constexpr int t = decltype(t_)::value; Dacc[NB1 + 1 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[4 + NB1 + t], o[3], Dacc[NB1 + 1 + t], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // value = f(idx, value) for every accumulator element that maps to a parameter of the folded block
  template <class F>
  __device__ __forceinline__ void each(int lane, F&& f) {
    constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
    const int cl = 4 * (lane >> 4), il = lane & 15;
    static_for<0, NTL>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cl + r;
        int idx = -1;
        if constexpr (t < NB1) { const int i = 16 * t + il; if (c < H) idx = (i < IN) ? c * IN + i : (i == IN ? ob1 + c : -1); }
        else if constexpr (t == NB1) { if (c < H) idx = (il < H) ? oW2 + c * H + il : (il == H ? ob2 + c : -1); }
        else { const int j = 16 * (t - NB1 - 1) + c; if (j < OUT) idx = (il < H) ? oW4 + j * H + il : (il == H ? ob4 + j : -1); }
        if (idx >= 0) Dacc[t][r] = f(idx, Dacc[t][r]);
      }
    });
  }
  __device__ __forceinline__ void flush(int lane, float* slab_blk) { each(lane, [&](int idx, float v) { slab_blk[idx] += v; return v; }); }
  // grid-per-workgroup mapping: the running sums live in the slab between uses; a use starts from them and stores them back
  __device__ __forceinline__ void init_from(int lane, const float* slab_blk) {
    init(lane);
    each(lane, [&](int idx, float) { return slab_blk[idx]; });
  }
  __device__ __forceinline__ void store(int lane, float* slab_blk) { each(lane, [&](int idx, float v) { slab_blk[idx] = v; return v; }); }
};
template <int IN, int H>
struct PEngine<IN, H, true> {
  using R = RecLay2<IN, H>;
  static constexpr int NB1 = (R::XP + 15) / 16, NTL = NB1 + 1;
  f32x4 Dacc[NTL];
  __device__ __forceinline__ void init(int) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) Dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&x)[(IN + 1) / 2], const f2 (&a1)[H / 2],
                                             const f2 (&g1)[H / 2], const f2 (&g2)[H / 2]) {
    rec_write2<IN, H>(rec, lane, x, a1, g1, g2);                 // all 64 records at once
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* base = rec + (((lane >> 5) & 1) + 4 * ((lane >> 4) & 1)) * R::RS + (lane & 15);   // row permutation: see LEngine
    constexpr int NOP = 3 + NB1;
    float op[2][NOP];
    auto fetch = [&](float (&o)[NOP], int kk) {
      const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * R::RS;
      o[0] = b[R::oG1]; o[1] = b[R::oG2]; o[2] = b[R::oA1];
      static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value; o[3 + t] = b[R::oX + 16 * t]; });
    };
    fetch(op[0], 0);
    static_for<0, 2 * GNS_REC_ROWS / 4>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      if constexpr (kk + 1 < 2 * GNS_REC_ROWS / 4) fetch(op[(kk + 1) & 1], kk + 1);
      __builtin_amdgcn_sched_barrier(0);
      const float (&o)[NOP] = op[kk & 1];
      static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value; Dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[0], o[3 + t], Dacc[t], 0, 0, 0); });
      Dacc[NB1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[1], o[2], Dacc[NB1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void flush(int lane, float* slab_blk) {
    constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H;
    const int cl = 4 * (lane >> 4), il = lane & 15;
    static_for<0, NTL>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cl + r;
        int idx = -1;
        if constexpr (t < NB1) { const int i = 16 * t + il; if (c < H) idx = (i < IN) ? c * IN + i : (i == IN ? ob1 + c : -1); }
        else { if (c < H) idx = (il < H) ? oW2 + c * H + il : (il == H ? ob2 + c : -1); }
        if (idx >= 0) slab_blk[idx] += Dacc[t][r];
      }
    });
  }
};

// ---- grid-per-workgroup mapping: phi' is differentiated in two places (gns_gridwg_bwd.hip) --------------------------
// A line's first-layer input is [m(dst) | r x b tau shift] (main.py:155): the m columns of dW1 are contracted per BUS
// (sum over the lines ending there of g1, times the bus's own latent vector), the line columns, db1, dW2, db2 per LINE.
// Both records are written by all 64 lanes at once; slab layout = the folded phi' block W1[H][IN] b1[H] W2[H][H] b2[H].
template <int IN, int H, int D>
struct GwEdgeEngine {      // record [line parameters | 1 | 0.. (8), a1 | 1 | 0 (12), g1 | 0 0 (12), g2 | 0 0 (12)]
  static constexpr int NX = IN - D;                      // 5 line parameters
  static constexpr int XP = (NX + 1 + 3) / 4 * 4, HP = (H + 1 + 3) / 4 * 4;
  static constexpr int oX = 0, oA1 = XP, oG1 = oA1 + HP, oG2 = oG1 + HP, raw = oG2 + HP;
  static constexpr int RS = ((raw / 4) | 1) * 4;
  static constexpr int RECF = 64 * RS + 32;
  static_assert(XP <= 16 && HP <= 16, "one 16-column tile per operand");
  f32x4 Dacc[2];
  template <class F>
  __device__ __forceinline__ void each(int lane, F&& f) {
    constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H;
    const int cl = 4 * (lane >> 4), il = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cl + r;
      if (c < H) {
        const int i1 = (il < NX) ? c * IN + D + il : (il == NX ? ob1 + c : -1);
        if (i1 >= 0) Dacc[0][r] = f(i1, Dacc[0][r]);
        const int i2 = (il < H) ? oW2 + c * H + il : (il == H ? ob2 + c : -1);
        if (i2 >= 0) Dacc[1][r] = f(i2, Dacc[1][r]);
      }
    }
  }
  __device__ __forceinline__ void init_from(int lane, const float* blk) {
    Dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; Dacc[1] = Dacc[0];
    each(lane, [&](int idx, float) { return blk[idx]; });
  }
  __device__ __forceinline__ void store(int lane, float* blk) { each(lane, [&](int idx, float v) { blk[idx] = v; return v; }); }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&xt)[(NX + 1) / 2], const f2 (&a1)[H / 2],
                                             const f2 (&g1)[H / 2], const f2 (&g2)[H / 2]) {
    f2* dst = reinterpret_cast<f2*>(rec + lane * RS);
    static_for<0, raw / 2>([&](auto q_) {
      constexpr int q = decltype(q_)::value, o = 2 * q;
      f2 e;
      if constexpr (o < oA1) {
        if constexpr (o + 1 < NX) e = xt[q];
        else if constexpr (o < NX) e = f2{xt[q].x, 1.f};
        else e = (o == NX) ? f2{1.f, 0.f} : f2{0.f, 0.f};
      } else if constexpr (o < oG1) {
        constexpr int i = o - oA1;
        if constexpr (i < H) e = a1[i / 2]; else e = (i == H) ? f2{1.f, 0.f} : f2{0.f, 0.f};
      } else if constexpr (o < oG2) {
        constexpr int i = o - oG1;
        if constexpr (i < H) e = g1[i / 2]; else e = f2{0.f, 0.f};
      } else {
        constexpr int i = o - oG2;
        if constexpr (i < H) e = g2[i / 2]; else e = f2{0.f, 0.f};
      }
      dst[q] = e;
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* base = rec + (((lane >> 5) & 1) + 4 * ((lane >> 4) & 1)) * RS + (lane & 15);   // row permutation: see LEngine
    static_for<0, 16>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * RS;
      const float og1 = b[oG1], og2 = b[oG2], ox = b[oX], oa = b[oA1];
      Dacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(og1, ox, Dacc[0], 0, 0, 0);
      Dacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(og2, oa, Dacc[1], 0, 0, 0);
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

template <int IN, int H, int D>
struct GwBusPhiEngine {    // record [m (D, padded to 4) | G1 | 0 0 (12)]: dW1[:, 0..D) = sum over buses G1 (x) m
  static constexpr int MP = (D + 3) / 4 * 4, HP = (H + 1 + 3) / 4 * 4;
  static constexpr int oM = 0, oG = MP, raw = oG + HP;
  static constexpr int RS = ((raw / 4) | 1) * 4;
  static constexpr int RECF = 64 * RS + 32;
  static constexpr int NB = (MP + 15) / 16;
  f32x4 Dacc[NB];
  template <class F>
  __device__ __forceinline__ void each(int lane, F&& f) {
    const int cl = 4 * (lane >> 4), il = lane & 15;
    static_for<0, NB>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cl + r, i = 16 * t + il;
        if (c < H && i < D) Dacc[t][r] = f(c * IN + i, Dacc[t][r]);
      }
    });
  }
  __device__ __forceinline__ void init_from(int lane, const float* blk) {
#pragma unroll
    for (int t = 0; t < NB; ++t) Dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    each(lane, [&](int idx, float) { return blk[idx]; });
  }
  __device__ __forceinline__ void store(int lane, float* blk) { each(lane, [&](int idx, float v) { blk[idx] = v; return v; }); }
  __device__ __forceinline__ void accumulate(float* rec, int lane, const f2 (&m)[D / 2], const f2 (&G1)[H / 2]) {
    f2* dst = reinterpret_cast<f2*>(rec + lane * RS);
    static_for<0, raw / 2>([&](auto q_) {
      constexpr int q = decltype(q_)::value, o = 2 * q;
      f2 e;
      if constexpr (o < oG) { if constexpr (o < D) e = m[q]; else e = f2{0.f, 0.f}; }
      else { constexpr int i = o - oG; if constexpr (i < H) e = G1[i / 2]; else e = f2{0.f, 0.f}; }
      dst[q] = e;
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* base = rec + (((lane >> 5) & 1) + 4 * ((lane >> 4) & 1)) * RS + (lane & 15);
    static_for<0, 16>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * RS;
      const float og = b[oG];
      static_for<0, NB>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        Dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(og, b[oM + 16 * t], Dacc[t], 0, 0, 0);
      });
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// ---- sub-record contraction (grid-per-workgroup backward) ----------------------------------------------------------------
// Instead of one wide row record per LearningBlock, a wave keeps a [64 rows][12 + 16] window in LDS: an A block (one
// adjoint vector, <= 12 columns) and a B block (16 input columns).  Every field of a row is written exactly once, by all
// 64 lanes, then the matrix pipe contracts the window over the rows: D[c][i] += sum_rows A[row][c] B[row][i].
// 7 KB per wave instead of 14-17, half the LDS store traffic of the half-wave records, same MFMA count.
struct GwSub {
  static constexpr int NA = 12, NB = 16, RS = NA + NB, RECF = 64 * RS + 32;   // RS = 28 = 4 (mod 8): see LEngine's row map
};
// The same window with a 32-column B block: a pass contracts 16 CONSECUTIVE B columns starting at any column, so fields that
// outlive a pass (the output-layer activations, the latent tail of a bus) are parked beside the columns other passes rewrite
// and small products ride along in the spare rows / columns of a pass that is issued anyway (gns_backward.hip, V2 sweep).
struct GwSubWide {
#ifdef GNS_NO_FOLDM
  static constexpr int NA = 12, NB = 32, RS = NA + NB, RECF = 64 * RS + 32;   // RS = 44 = 4 (mod 8)
#else
  static constexpr int NA = 16, NB = 36, RS = NA + NB, RECF = 64 * RS + 32;   // all 16 rows of the tile; RS = 52 = 4 (mod 8); B columns 32..35 pad
#endif
};
__device__ __forceinline__ void gws_w2r() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void gws_r2w() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#ifdef GNS_ABLATE_REC
template <class S = GwSub> __device__ __forceinline__ void gws_putA(float* rec, int lane, int pair, f2 v) { asm volatile("" :: "v"(v)); }
template <class S = GwSub> __device__ __forceinline__ void gws_putB(float* rec, int lane, int pair, f2 v) { asm volatile("" :: "v"(v)); }
#else
template <class S = GwSub> __device__ __forceinline__ void gws_putA(float* rec, int lane, int pair, f2 v) { reinterpret_cast<f2*>(rec + lane * S::RS)[pair] = v; }
template <class S = GwSub> __device__ __forceinline__ void gws_putB(float* rec, int lane, int pair, f2 v) { reinterpret_cast<f2*>(rec + lane * S::RS + S::NA)[pair] = v; }
#endif
// One contraction pass over the window, in two halves of 8 k-steps (16 operand registers in flight instead of 32)
template <class S = GwSub, int BOFF = 0>      // BOFF: first B column of the 16 this pass contracts
__device__ __forceinline__ void gws_pass(const float* rec, int lane, f32x4& Dt) {
#ifdef GNS_ABLATE_PASS
  asm volatile("" : "+v"(Dt));
  return;
#endif
  const float* base = rec + (((lane >> 5) & 1) + 4 * ((lane >> 4) & 1)) * S::RS + (lane & 15);
#ifdef GNS_PASS_HALVES          // the round-2 schedule: operands of 8 k-steps, their 8 MFMAs, twice
  static_for<0, 2>([&](auto h_) {
    constexpr int hh = decltype(h_)::value;
    float Aop[8], Bop[8];
    static_for<0, 8>([&](auto kk_) {
      constexpr int kk = 8 * hh + decltype(kk_)::value;
      const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * S::RS;
      Aop[kk - 8 * hh] = b[0];
      Bop[kk - 8 * hh] = b[S::NA + BOFF];
    });
    static_for<0, 8>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      Dt = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[kk], Bop[kk], Dt, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
#else
  // Operand ring of GWS_AHEAD k-steps: the LDS read of step kk + AHEAD is issued right behind the MFMA of step kk, so the
  // matrix pipe never waits for a batch of reads (8 operand registers instead of 16).
#ifndef GWS_AHEAD
#define GWS_AHEAD 4
#endif
  constexpr int AH = GWS_AHEAD;
  float Aop[AH], Bop[AH];
  auto rd = [&](auto kk_) {
    constexpr int kk = decltype(kk_)::value;
    const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * S::RS;
    Aop[kk % AH] = b[0];
    Bop[kk % AH] = b[S::NA + BOFF];
  };
  static_for<0, AH>([&](auto kk_) { rd(kk_); });
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, 16>([&](auto kk_) {
    constexpr int kk = decltype(kk_)::value;
    Dt = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[kk % AH], Bop[kk % AH], Dt, 0, 0, 0);
    if constexpr (kk + AH < 16) rd(std::integral_constant<int, kk + AH>{});
    __builtin_amdgcn_sched_barrier(0);
  });
#endif
}
// slab_blk[idx_of(c, i)] += D[c][i] for the entries that map to a parameter (idx_of returns -1 otherwise)
template <class F>
__device__ __forceinline__ void gws_flush(int lane, const f32x4& Dt, float* blk, F&& idx_of, bool store = false) {
  const int cl = 4 * (lane >> 4), il = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = idx_of(cl + r, il);
    if (idx >= 0) blk[idx] = store ? Dt[r] : blk[idx] + Dt[r];     // store: the slab was not zeroed and this is its first use
  }
}
// stage[idx_of(c, i)] = D[c][i]: the wave's LDS stage of a gradient block (every entry is produced exactly once per use)
template <class F>
__device__ __forceinline__ void gws_stage(int lane, const f32x4& Dt, float* stage, F&& idx_of) {
  const int cl = 4 * (lane >> 4), il = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = idx_of(cl + r, il);
    if (idx >= 0) stage[idx] = Dt[r];
  }
}


// ---- V3: weight-gradient chains drained in the background of the weight streams ---------------------------------------------
// A window [64 rows][RS] stays in LDS while the wave runs its next packed-FMA blocks; each background slot of those blocks
// (gns_device.h, stream_pairs) issues ONE MFMA of the window's chains (chains alternate, so consecutive MFMAs never wait for
// each other) and reads the operands of the slot three ahead (= the next 32-float step, whose scalar-load wait also retires
// these LDS reads).  A program P describes a window: RS, NCH chains with their A / B column offsets, LEN = 16 * NCH slots.
template <class P, int START, class ACC>
struct GwDrain {
  const float* base;                 // rec + row-map of this lane (see LEngine) + (lane & 15)
  float (&ra)[4];
  float (&rb)[4];
  ACC acc;                           // acc(ic<chain>) -> f32x4& accumulator tile of the chain
  template <int G>
  __device__ __forceinline__ void load() {
    constexpr int kk = G / P::NCH, ch = G % P::NCH;
    const float* b = base + (8 * (kk >> 1) + 2 * (kk & 1)) * P::RS;
    ra[G & 3] = b[P::aoff(ch)];
    rb[G & 3] = b[P::boff(ch)];
  }
  template <int S>
  __device__ __forceinline__ void slot() {
    constexpr int g = START + S;
    if constexpr (g + 3 < P::LEN) load<g + 3>();
    if constexpr (g < P::LEN) {
      constexpr int ch = g % P::NCH;
#ifndef GNS_ABLATE_PASS
      f32x4& T = acc(std::integral_constant<int, ch>{});
      T = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[g & 3], rb[g & 3], T, 0, 0, 0);
#endif
    }
  }
  // the first three slots' operands (call once after the window has been written and made visible)
  __device__ __forceinline__ void prologue() { load<START>(); load<START + 1>(); load<START + 2>(); }
  // issue whatever the streams did not get to (not overlapped with anything)
  template <int FROM_S>
  __device__ __forceinline__ void rest() {
    static_for<FROM_S, (P::LEN > START ? P::LEN - START : 0)>([&](auto s_) { slot<decltype(s_)::value>(); });
  }
};
template <class P, int START, class ACC>
__device__ __forceinline__ GwDrain<P, START, ACC> gw_drain(const float* base, float (&ra)[4], float (&rb)[4], ACC acc) {
  return GwDrain<P, START, ACC>{base, ra, rb, acc};
}

// window programs of the V3 lane-per-grid backward (all windows live in the wave's record buffer, one at a time):
//   L  [g1 0..11 | x,1 12..47 | g2 48..59 | a1,1 60..71]                 chains: dW2, dW1 tile 0..NB1-1
//   E  [g1 0..11 | line parameters,1 12..19 | g2 20..31 | a1,1 32..43]   chains: line columns of dW1 | db1, dW2 | db2
//   B  [G1 0..11 | m 12..]                                               chains: latent columns of dW1, 16 at a time
template <int NB1>
struct GwProgL { static constexpr int RS = 76, NCH = 1 + NB1, LEN = 16 * NCH;
                 static constexpr int aoff(int ch) { return ch == 0 ? 48 : 0; }
                 static constexpr int boff(int ch) { return ch == 0 ? 60 : 12 + 16 * (ch - 1); } };
struct GwProgE { static constexpr int RS = 44, NCH = 2, LEN = 32;
                 static constexpr int aoff(int ch) { return ch == 0 ? 0 : 20; }
                 static constexpr int boff(int ch) { return ch == 0 ? 12 : 32; } };
template <int NDM>
struct GwProgB { static constexpr int RS = 36, NCH = NDM, LEN = 16 * NDM;
                 static constexpr int aoff(int) { return 0; }
                 static constexpr int boff(int ch) { return 12 + 16 * ch; } };
