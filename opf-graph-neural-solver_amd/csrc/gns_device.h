// Device-side building blocks shared by the forward and backward kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gns_common.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
// address space 4 (constant): loads with a wave-uniform address become s_load (scalar cache -> SGPRs)
typedef const __attribute__((address_space(4))) float* cfp;
typedef const __attribute__((address_space(4))) int* cip;
typedef const __attribute__((address_space(4))) f16v* cf16p;
typedef f16v f16u __attribute__((aligned(4)));                       // a 16-float chunk that starts at any dword
typedef const __attribute__((address_space(4))) f16u* cf16up;

// ---- teams: several workgroups (on different CUs) working the buses of ONE 64-grid group ---------------------------------
// Everything the waves of a group exchange already travels through HBM rows (state, adjoints, per-line slots), so a team needs
// only (1) a barrier across its workgroups and (2) the per-wave partial sums in HBM instead of LDS.  The barrier is an arrival
// counter (one per group, zero at launch, never reset: barrier b is passed when it reaches b * size); thread 0 of each
// workgroup adds 1 and polls.  Around it the team's stores must become visible to the other members:
//   * members on the SAME XCD share its L2, and the vector L1 is write-through: waiting for the stores (s_waitcnt vmcnt(0))
//     releases them, dropping the own L1 lines (buffer_inv sc0) acquires - a few microseconds;
//   * members on different XCDs need the agent-scope fences (L2 write-back + invalidate), ~170 us per barrier measured.
// Workgroups are dealt to the XCDs round-robin, so the kernels place a team on blocks 8 apart - and CHECK it: team_setup
// exchanges the XCC ids through a full-fence barrier and only a team that really shares an XCD takes the light path.
// All workgroups of all teams are resident at once (gns_team_size caps groups * size at the CU count), so the poll always
// ends; the poll count is bounded all the same - a wave that gives up marks the team failed (NaN losses / gradients, loud)
// instead of hanging the device.
struct GnsTeam {
  unsigned* ctr;          // arrival counter of the group being worked
  float* red;             // [2 parities][GNS_MAXP][64][2] partial sums of this group
  int size, member;       // workgroups in the team, this workgroup's rank
  unsigned epoch;         // barriers passed on ctr
  int* failed;            // LDS word shared by the workgroup
  bool local;             // every member runs on the same XCD
};
#define GNS_TEAM_POLL_LIMIT (1u << 22)
__device__ __forceinline__ void team_arrive_and_wait(GnsTeam& t, unsigned* ctr, unsigned target) {
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // a failed workgroup still arrives: its partners move on
    unsigned polls = 0;
    while (!*t.failed && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > GNS_TEAM_POLL_LIMIT) { *t.failed = 1; break; }
    }
  }
}
__device__ __forceinline__ void team_barrier(GnsTeam& t) {
  if (t.size == 1) { __syncthreads(); return; }
  t.epoch += 1;
  if (t.local) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores have reached the shared L2
    __syncthreads();
    team_arrive_and_wait(t, t.ctr, t.epoch * (unsigned)t.size);
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");                     // forget what the vector L1 held of the partners' rows
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    team_arrive_and_wait(t, t.ctr, t.epoch * (unsigned)t.size);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}
// Once per kernel, on the 64-byte line of the team's first group: word 1 = setup counter, words 4.. = XCC id of each member.
__device__ __forceinline__ void team_setup(GnsTeam& t, unsigned* line) {
  t.local = false;
  if (t.size == 1) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;     // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) __hip_atomic_store(line + 4 + t.member, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  team_arrive_and_wait(t, line + 1, (unsigned)t.size);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  bool same = true;
  for (int m = 0; m < t.size; ++m) same = same && (__hip_atomic_load(line + 4 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc + 1);
  t.local = same;
}

#define GNS_LEAKY 0.01f   // torch.nn.LeakyReLU default slope (GNS/main.py:23)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Walk NF (even) consecutive floats at a wave-uniform address in pairs: f(ic<w>, {p[w],p[w+1]}), w = 0,2,...
// 16-float chunks are fetched with s_load_dwordx16 one chunk ahead of their use; the scheduling
// barriers keep hipcc from hoisting every load to the top (which spills SGPRs through v_writelane).
// A zero the optimiser cannot see through (SGPR-constrained asm output = wave-uniform): added to a weight
// pointer it pins the s_loads to the place where the weights are used.  Without it LICM hoists all ~600
// loads of a LearningBlock out of the bus / line loops and spills the SGPRs through v_writelane.
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}

// Scalar loads return out of order, so the only wait hipcc can emit for them is lgkmcnt(0) = "everything".  To let
// the next chunk's load fly during this chunk's FMAs, the wait for the CURRENT chunk has to come before the next
// load is issued: touching one of its registers forces exactly that.
__device__ __forceinline__ void touch(const f16v& v) { asm volatile("" ::"s"(v[0])); }

// Background hook of a weight stream: slot<S>() is called three times per 32-float step (before packed FMAs 0, 5 and 10 of the
// step), S = 3 * step + {0,1,2}.  The V3 backward issues one matrix-pipe instruction of a pending weight-gradient chain per
// slot: a 16x16x4 fp32 MFMA occupies the matrix pipe for 32 cycles and the vector issue for 8, so one MFMA per ~6 packed FMAs
// keeps both pipes busy from ONE wave (gns_dw.h, GwDrain).
struct NoBG { template <int S> __device__ __forceinline__ void slot() {} };

// Links between consecutive weight streams.  A stream's first 32 floats cost a full scalar-cache round trip before its first
// FMA (measured: 0.18 ms of the 2.6 ms backward, gns_device.h GNS_ABLATE_SLOAD_ALL).  A linked stream fetches the first chunk
// of the NEXT stream in its own last step - into the half of its double buffer that is idle by then, so no extra SGPRs while
// it runs - and hands it over; the next stream starts from those registers.
struct WFirst { f16v a0, a1; };
template <class VP = cf16p>
__device__ __forceinline__ void stream_first(cfp p, WFirst& w) { p += opaque_zero(); w.a0 = *(VP)(p); w.a1 = *(VP)(p + 16); }
template <bool PRE_, bool NXT_, class NVP_ = cf16p>
struct WLink {
  static constexpr bool PRE = PRE_, NXT = NXT_;
  using NVP = NVP_;
  const WFirst* pre;      // PRE: this stream's first chunk, fetched ahead
  cfp next;               // NXT: where the next stream starts
  WFirst* nxt;            //      and where its first chunk goes
};
using NoLink = WLink<false, false>;

template <int NF, class VP = cf16p, class F, class BG = NoBG, class LK = NoLink>
__device__ __forceinline__ void stream_pairs(cfp p, F&& f, BG&& bg = BG{}, LK lk = LK{}) {
  // 32-float steps: two s_load_dwordx16 are in flight while the previous 32 floats feed 16 packed FMAs
  constexpr int NST = (NF + 31) / 32;
  p += opaque_zero();
  f16v a0, a1, b0, b1;
#ifdef GNS_ABLATE_SLOAD_ALL   // diagnostic: no scalar load at all, not even a stream's first chunk (wrong numbers; what ALL weight fetches cost)
  {
    int zi = 0;
    asm volatile("" : "+s"(zi));
    const float z = __builtin_bit_cast(float, zi);
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = z; a1[i] = z; }
  }
#else
  if constexpr (LK::PRE) { a0 = lk.pre->a0; a1 = lk.pre->a1; }
  else { a0 = *(VP)(p); a1 = *(VP)(p + 16); }
#endif
  static_for<0, NST>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    f16v& c0 = (c & 1) ? b0 : a0;
    f16v& c1 = (c & 1) ? b1 : a1;
    f16v& n0 = (c & 1) ? a0 : b0;
    f16v& n1 = (c & 1) ? a1 : b1;
    touch(c0);
    touch(c1);
#if defined(GNS_ABLATE_SLOAD) || defined(GNS_ABLATE_SLOAD_ALL)   // diagnostic: the stream keeps re-using its first 32 floats (wrong numbers, no scalar-load latency)
    if constexpr (c + 1 < NST) { n0 = c0; n1 = c1; }
#else
    if constexpr (c + 1 < NST) {
      n0 = *(VP)(p + 32 * (c + 1));
      n1 = *(VP)(p + 32 * (c + 1) + 16);
    } else if constexpr (LK::NXT) {
      cfp q = lk.next + opaque_zero();
      n0 = *(typename LK::NVP)(q);
      n1 = *(typename LK::NVP)(q + 16);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int w = c * 32 + 2 * t;
      if constexpr (!std::is_same<std::decay_t<BG>, NoBG>::value && (t == 0 || t == 5 || t == 10)) {
        bg.template slot<3 * c + t / 5>();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (w < NF) {
        if constexpr (t < 8) f(std::integral_constant<int, w>{}, f2{c0[2 * t], c0[2 * t + 1]});
        else f(std::integral_constant<int, w>{}, f2{c1[2 * (t - 8)], c1[2 * (t - 8) + 1]});
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
#if !defined(GNS_ABLATE_SLOAD) && !defined(GNS_ABLATE_SLOAD_ALL)
  if constexpr (LK::NXT) {
    lk.nxt->a0 = ((NST - 1) & 1) ? a0 : b0;
    lk.nxt->a1 = ((NST - 1) & 1) ? a1 : b1;
  }
#endif
}

// Pull [p, p + nfloats) into the scalar cache: one s_load_dword per 64-byte line, all in flight at once, one wait.  Every step of
// the K loop has its own weights, so whoever starts streaming a (family, step) block finds the scalar cache cold - and a weight
// stream keeps only two lines in flight: without this the first bus of a sweep pays a full L2 round trip per line.
__device__ __forceinline__ void scalar_cache_warm(cfp p, long long nfloats) {
#ifndef GNS_NO_SCALAR_WARM
  const unsigned nbytes = (unsigned)(nfloats * 4);
  unsigned off, dummy;
  asm volatile(
      "s_mov_b32 %0, 0\n"
      "1:\n"
      "s_load_dword %1, %2, %0\n"
      "s_add_u32 %0, %0, 64\n"
      "s_cmp_lt_u32 %0, %3\n"
      "s_cbranch_scc1 1b\n"
      "s_waitcnt lgkmcnt(0)\n"
      : "=&s"(off), "=&s"(dummy)
      : "s"(p), "s"(nbytes)
      : "scc", "memory");
#endif
}

// The same in two halves, for a phase that has other work to do while the lines arrive: warm_issue puts the loads in flight and
// hands back the registers they were issued with; they stay reserved until warm_wait (an s_waitcnt lgkmcnt(0)) has consumed them.
struct WarmTok { unsigned off, dummy; };
__device__ __forceinline__ WarmTok scalar_cache_warm_issue(cfp p, long long nfloats) {
  WarmTok t{0u, 0u};
#ifndef GNS_NO_SCALAR_WARM
  const unsigned nbytes = (unsigned)(nfloats * 4);
  asm volatile(
      "s_mov_b32 %0, 0\n"
      "1:\n"
      "s_load_dword %1, %2, %0\n"
      "s_add_u32 %0, %0, 64\n"
      "s_cmp_lt_u32 %0, %3\n"
      "s_cbranch_scc1 1b\n"
      : "=&s"(t.off), "=&s"(t.dummy)
      : "s"(p), "s"(nbytes)
      : "scc", "memory");
#endif
  return t;
}
__device__ __forceinline__ void scalar_cache_warm_wait(WarmTok& t) {
#ifndef GNS_NO_SCALAR_WARM
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t.off), "+s"(t.dummy) : : "memory");
#endif
}

// Pin a value to the point where it was computed.  MachineSink otherwise moves a whole LearningBlock's FMAs down
// to the first use of its result (past the next line loop), keeping ~600 weights alive in v_writelane spills.
__device__ __forceinline__ void pin(f2& v) { asm volatile("" : "+v"(v)); }
template <int N>
__device__ __forceinline__ void pin_all(f2 (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) pin(a[i]);
}

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 lrelu2(f2 z) { return __builtin_elementwise_max(z, z * GNS_LEAKY); }
template <int I, int HP>
__device__ __forceinline__ float lane_of(const f2 (&a)[HP]) { return (I & 1) ? a[I / 2].y : a[I / 2].x; }
// derivative of LeakyReLU from its OUTPUT (sign-preserving): 1 where a > 0 else slope
#ifndef GNS_DLRELU_CLAMP
#define GNS_DLRELU_CLAMP 1
#endif
__device__ __forceinline__ f2 dlrelu2(f2 a) {
#if GNS_DLRELU_CLAMP
  // Two packed instructions instead of two compares and two selects.  t = clamp(a * 2^126) - the clamp bit of the packed multiply
  // saturates to [0, 1] - is 1 for every normal a > 0 and 0 for a <= 0; t * 0.99f + 0.01f (one rounding) is then exactly 1.0f or
  // exactly 0.01f, so every gradient keeps its bits.  (A subnormal a > 0, below 1.2e-38, would get a slope between the two.)
  f2 t;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(a), "s"(splat(0x1p126f)));
  return __builtin_elementwise_fma(t, splat(1.f - GNS_LEAKY), splat(GNS_LEAKY));
#else
  return f2{a.x > 0.f ? 1.f : GNS_LEAKY, a.y > 0.f ? 1.f : GNS_LEAKY};
#endif
}

// LearningBlock forward (GNS/main.py:25-31) from the T-stream:
//   W1t[IN][H] b1[H] W2t[H][H] b2[H] W4t[H][OUTP] b4[OUTP]
// Two output neurons of one input share an aligned SGPR pair: v_pk_fma_f32 acc2, s[pair], x_i(bcast).
template <int IN, int H, int OUTP>
struct TLay {
  static constexpr int oW1 = 0, ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + H * OUTP,
                       total = ob4 + OUTP;
};

template <int IN, int H, int OUTP>
__device__ __forceinline__ void mlp_fwd(cfp blk, const f2 (&x)[(IN + 1) / 2], f2 (&a1)[H / 2], f2 (&a2)[H / 2], f2 (&y)[OUTP / 2]) {
  using B = TLay<IN, H, OUTP>;
  static_assert(H % 2 == 0 && OUTP % 2 == 0, "pairs");
  stream_pairs<B::total>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value;
    if constexpr (w < B::ob1) {
      constexpr int i = w / H, j = (w % H) / 2;
      const f2 xi = splat(lane_of<i>(x));        // inputs arrive as aligned pairs: op_sel picks the half, no v_mov
      a1[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a1[j]);
    } else if constexpr (w < B::oW2) {
      constexpr int j = (w - B::ob1) / 2;
      a1[j] = lrelu2(a1[j] + s);
    } else if constexpr (w < B::ob2) {
      constexpr int q = w - B::oW2, i = q / H, j = (q % H) / 2;
      const f2 xi = splat(lane_of<i>(a1));
      a2[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a2[j]);
    } else if constexpr (w < B::oW4) {
      constexpr int j = (w - B::ob2) / 2;
      a2[j] = lrelu2(a2[j] + s);
    } else if constexpr (w < B::ob4) {
      constexpr int q = w - B::oW4, i = q / OUTP, j = (q % OUTP) / 2;
      const f2 xi = splat(lane_of<i>(a2));
      y[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, y[j]);
    } else {
      constexpr int j = (w - B::ob4) / 2;
      y[j] += s;
    }
  });
  pin_all(y);
}

// LearningBlock backward, data path, from the N-stream: W4n[OUTP][H] W2n[H][H] W1n[H][INP]
//   g3 = dL/dy.  Produces g2, g1 = dL/d(pre-activation of layer 2, 1) and gx = dL/dx[0..NX).
template <int IN, int H, int OUTP>
struct NLay {
  static constexpr int INP = IN + (IN & 1);
  static constexpr int oW4 = 0, oW2 = OUTP * H, oW1 = oW2 + H * H, total = oW1 + H * INP;
};

// ACC = true: gx arrives holding values to accumulate into (e.g. the running adjoint of the latent vector).
template <int IN, int H, int OUTP, int NX, bool ACC = false>
__device__ __forceinline__ void mlp_bwd(cfp blk, const f2 (&a1)[H / 2], const f2 (&a2)[H / 2], const f2 (&g3)[OUTP / 2],
                                        f2 (&g2)[H / 2], f2 (&g1)[H / 2], f2 (&gx)[NX / 2]) {
  using B = NLay<IN, H, OUTP>;
  static_assert(NX % 2 == 0 && NX <= B::INP, "NX");
  stream_pairs<B::total>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value;
    if constexpr (w < B::oW2) {
      constexpr int j = w / H, i = (w % H) / 2;
      const f2 gj = splat(lane_of<j>(g3));
      g2[i] = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, g2[i]);
    } else if constexpr (w < B::oW1) {
      constexpr int q = w - B::oW2, j = q / H, i = (q % H) / 2;
      if constexpr (q == 0) {
        static_for<0, H / 2>([&](auto u_) { constexpr int u = decltype(u_)::value; g2[u] = g2[u] * dlrelu2(a2[u]); });
      }
      const f2 gj = splat(lane_of<j>(g2));
      g1[i] = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, g1[i]);
    } else {
      constexpr int q = w - B::oW1, j = q / B::INP, i = (q % B::INP) / 2;
      if constexpr (q == 0) {
        static_for<0, H / 2>([&](auto u_) { constexpr int u = decltype(u_)::value; g1[u] = g1[u] * dlrelu2(a1[u]); });
      }
      if constexpr (2 * i < NX) {
        const f2 gj = splat(lane_of<j>(g1));
        gx[i] = (j == 0 && !ACC) ? s * gj : __builtin_elementwise_fma(s, gj, gx[i]);
      }
    }
  });
  pin_all(gx);
}

// ---- phi' : the first two layers of a LearningBlock (the third is folded into L', see gns_common.h) --------------
// T-stream: W1t[IN][H] b1[H] W2t[H][H] b2[H]
template <int IN, int H>
struct TLay2 {
  static constexpr int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, total = ob2 + H;
};

template <int IN, int H, class BG = NoBG, class LK = NoLink, class SL = decltype(nullptr)>
__device__ __forceinline__ void mlp2_fwd(cfp blk, const f2 (&x)[(IN + 1) / 2], f2 (&a1)[H / 2], f2 (&a2)[H / 2], BG&& bg = BG{}, LK lk = LK{}, SL sl = nullptr) {
  using B = TLay2<IN, H>;
  constexpr bool KEEP = !std::is_same<SL, decltype(nullptr)>::value;      // see phi_tail
  auto act = [&](f2 z, int layer, int j) {
    if constexpr (KEEP) { const f2 k = dlrelu2(z); sl[layer][j] = k; return z * k; }
    else return lrelu2(z);
  };
  stream_pairs<B::total, cf16p>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value;
    if constexpr (w < B::ob1) {
      constexpr int i = w / H, j = (w % H) / 2;
      const f2 xi = splat(lane_of<i>(x));
      a1[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a1[j]);
    } else if constexpr (w < B::oW2) {
      constexpr int j = (w - B::ob1) / 2;
      a1[j] = act(a1[j] + s, 0, j);
    } else if constexpr (w < B::ob2) {
      constexpr int q = w - B::oW2, i = q / H, j = (q % H) / 2;
      const f2 xi = splat(lane_of<i>(a1));
      a2[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a2[j]);
    } else {
      constexpr int j = (w - B::ob2) / 2;
      a2[j] = act(a2[j] + s, 1, j);
    }
  }, bg, lk);
  pin_all(a2);
}

// phi' in two parts.  Every line ending at a bus feeds phi' the SAME latent vector m[dst] (main.py:155), so the first D
// rows of W1t - 200 of the 370 streamed floats - give the same partial sums u for all of them: phi_head once per bus,
// phi_tail (the 5 line parameters, b1, layer 2) once per line.  The accumulation order is that of mlp2_fwd
// (inputs 0..D-1, then D..IN-1, then the bias), so a1, a2 are bitwise the same.
template <int D, int H, class BG = NoBG, class LK = NoLink>
__device__ __forceinline__ void phi_head(cfp blk, const f2 (&m)[D / 2], f2 (&u)[H / 2], BG&& bg = BG{}, LK lk = LK{}) {
  stream_pairs<D * H, cf16p>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, i = w / H, j = (w % H) / 2;
    const f2 xi = splat(lane_of<i>(m));
    u[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, u[j]);
  }, bg, lk);
  pin_all(u);
}
// SL (a pointer to f2[2][H / 2], or nullptr_t): also hand out LeakyReLU's slope at each hidden unit (1.0f or 0.01f), formed on the way
// (z * slope(z) is bitwise max(z, 0.01 z) for every normal z) - a backward that keeps them multiplies instead of deriving them again
template <int IN, int H, int D, class BG = NoBG, class LK = NoLink, class SL = decltype(nullptr)>
__device__ __forceinline__ void phi_tail(cfp blk, const f2 (&u)[H / 2], const f2 (&xt)[(IN - D + 1) / 2], f2 (&a1)[H / 2], f2 (&a2)[H / 2], BG&& bg = BG{}, LK lk = LK{}, SL sl = nullptr) {
  using B = TLay2<IN, H>;
  constexpr bool KEEP = !std::is_same<SL, decltype(nullptr)>::value;
  auto act = [&](f2 z, int layer, int j) {
    if constexpr (KEEP) { const f2 k = dlrelu2(z); sl[layer][j] = k; return z * k; }
    else return lrelu2(z);
  };
  constexpr int W0 = D * H;
#pragma unroll
  for (int j = 0; j < H / 2; ++j) a1[j] = u[j];
  stream_pairs<B::total - W0, cf16up>(blk + W0, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value + W0;
    if constexpr (w < B::ob1) {
      constexpr int i = w / H, j = (w % H) / 2;
      const f2 xi = splat(lane_of<i - D>(xt));
      a1[j] = __builtin_elementwise_fma(s, xi, a1[j]);
    } else if constexpr (w < B::oW2) {
      constexpr int j = (w - B::ob1) / 2;
      a1[j] = act(a1[j] + s, 0, j);
    } else if constexpr (w < B::ob2) {
      constexpr int q = w - B::oW2, i = q / H, j = (q % H) / 2;
      const f2 xi = splat(lane_of<i>(a1));
      a2[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a2[j]);
    } else {
      constexpr int j = (w - B::ob2) / 2;
      a2[j] = act(a2[j] + s, 1, j);
    }
  }, bg, lk);
  pin_all(a2);
}

// N-stream: W2n[H][H] W1n[H][INP].  gh = dL/d(a2) (post-activation).  g2, g1 = pre-activation adjoints, gx = dL/dx[0..NX).
template <int IN, int H>
struct NLay2 {
  static constexpr int INP = IN + (IN & 1);
  static constexpr int oW1 = H * H, total = oW1 + H * INP;
};

template <int IN, int H, int NX, bool ACC = false>
__device__ __forceinline__ void mlp2_bwd(cfp blk, const f2 (&a1)[H / 2], const f2 (&a2)[H / 2], const f2 (&gh)[H / 2],
                                         f2 (&g2)[H / 2], f2 (&g1)[H / 2], f2 (&gx)[NX / 2]) {
  using B = NLay2<IN, H>;
  static_assert(NX % 2 == 0 && NX <= B::INP, "NX");
#pragma unroll
  for (int u = 0; u < H / 2; ++u) g2[u] = gh[u] * dlrelu2(a2[u]);
  stream_pairs<B::total>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value;
    if constexpr (w < B::oW1) {
      constexpr int j = w / H, i = (w % H) / 2;
      const f2 gj = splat(lane_of<j>(g2));
      g1[i] = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, g1[i]);
    } else {
      constexpr int q = w - B::oW1, j = q / B::INP, i = (q % B::INP) / 2;
      if constexpr (q == 0) {
        static_for<0, H / 2>([&](auto u_) { constexpr int u = decltype(u_)::value; g1[u] = g1[u] * dlrelu2(a1[u]); });
      }
      if constexpr (2 * i < NX) {
        const f2 gj = splat(lane_of<j>(g1));
        gx[i] = (j == 0 && !ACC) ? s * gj : __builtin_elementwise_fma(s, gj, gx[i]);
      }
    }
  });
  pin_all(gx);
}

// ---- data gradients from the FORWARD layouts (split backward) ----------------------------------------------------------------------
// The backward recomputes a network's hidden activations from the forward layouts (W1t[in][out], W2t, W4t) and then needs the same
// matrices transposed for the data gradients.  Streaming a second, transposed copy (the N-stream) doubles the network's footprint in
// the 16 KB scalar cache; these read the forward layout instead: a row of W[in][out] holds all outputs of ONE input, so an (aligned)
// weight pair is two OUTPUTS and the packed FMA multiplies element-wise with the adjoint pair (no broadcast) into a two-lane partial
// sum per input; the lanes are added when the row ends.  The recomputation stays bit-identical to the forward pass (LeakyReLU's
// derivative is taken on exactly the forward's activations); the data gradients change by rounding only (another summation order).
//   gin[NO] -> sum_j W[i][j] gin[j] for i = 0..NI-1 (NI even), handed to sink(ic<i / 2>, f2{value_i, value_i+1}) as each pair completes
template <int NI, int NO, class VP = cf16up, class F>
__device__ __forceinline__ void bwd_from_fwd_layout(cfp wt, const f2 (&gin)[NO / 2], F&& sink) {
  static_assert(NI % 2 == 0 && NO % 2 == 0, "pairs");
  f2 acc = {0.f, 0.f};
  float even = 0.f;
  stream_pairs<NI * NO, VP>(wt, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, i = w / NO, jp = (w % NO) / 2;
    acc = (jp == 0) ? s * gin[0] : __builtin_elementwise_fma(s, gin[jp], acc);
    if constexpr (jp == NO / 2 - 1) {
      if constexpr (i % 2 == 0) { even = acc.x + acc.y; asm volatile("" : "+v"(even)); }
      else { f2 v = f2{even, acc.x + acc.y}; pin(v); sink(std::integral_constant<int, i / 2>{}, v); }
    }
  });
}
template <int NI, int NO, class VP = cf16up>
__device__ __forceinline__ void bwd_rows_fwd_layout(cfp wt, const f2 (&gin)[NO / 2], f2 (&gout)[NI / 2]) {
  bwd_from_fwd_layout<NI, NO, VP>(wt, gin, [&](auto ip_, f2 v) { gout[decltype(ip_)::value] = v; });
}

// ---- layer-wise data path of the backward (used where each layer's weight gradient is contracted as soon as its operands exist) ----
// gout[i] = sum_j Wn[j][i] gin[j] from an [NJ][H] stream (output layer and hidden layer of the data path)
template <int NJP, int H, class BG = NoBG, class LK = NoLink>
__device__ __forceinline__ void bwd_rows(cfp blk, const f2 (&gin)[NJP / 2], f2 (&gout)[H / 2], BG&& bg = BG{}, LK lk = LK{}) {
  stream_pairs<NJP * H, cf16p>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, j = w / H, i = (w % H) / 2;
    const f2 gj = splat(lane_of<j>(gin));
    gout[i] = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, gout[i]);
  }, bg, lk);
  pin_all(gout);
}
// Input adjoints from the input-major stream W1x[NG][H][4]: four inputs at a time, each finished pair handed to sink(ic<pair>, value)
template <int NG, int H, class F, class BG = NoBG, class LK = NoLink>
__device__ __forceinline__ void bwd_inputs(cfp blk, const f2 (&g1)[H / 2], F&& sink, BG&& bg = BG{}, LK lk = LK{}) {
  f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  stream_pairs<NG * H * 4, cf16p>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, g = w / (4 * H), r = w % (4 * H), j = r / 4, half = (r % 4) / 2;
    const f2 gj = splat(lane_of<j>(g1));
    if constexpr (half == 0) acc0 = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, acc0);
    else acc1 = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, acc1);
    if constexpr (j == H - 1 && half == 1) {
      pin(acc0); pin(acc1);
      sink(std::integral_constant<int, 2 * g>{}, acc0);
      sink(std::integral_constant<int, 2 * g + 1>{}, acc1);
    }
  }, bg, lk);
}

// float4-row addressing: [row][lane] with 16 B per lane -> every wave access is one contiguous 1 KiB
// The row base (wave-uniform) is pinned to a scalar register pair and the lane contributes one 32-bit byte offset, so that an
// access is `global_load_dwordx4 v, v_off, s[base:base+1]`: hipcc otherwise folds the lane into a 64-bit per-lane address
// (v_lshl_add_u64 / v_mad_u64_u32 per access and a VGPR pair per live address).  Measured (round 2, -DGNS_ROW_SADDR=1): the
// forward drops to 125 VGPRs with no scratch and the backward to 25 spilled VGPRs, yet both are 3 % SLOWER (1.04 vs 1.00 ms,
// 2.73 vs 2.66 ms): the scalar address chain sits in front of every load and costs 40 more spilled SGPRs.  Off by default.
#ifndef GNS_ROW_SADDR
#define GNS_ROW_SADDR 0
#endif
__device__ __forceinline__ const f4* row_ptr(const float* base, long long row, int lane) {
#if GNS_ROW_SADDR
  const char* p = reinterpret_cast<const char*>(base) + row * (GNS_LANES * 16);
  asm("" : "+s"(p));
  return reinterpret_cast<const f4*>(p + (unsigned)(lane * 16));
#else
  return reinterpret_cast<const f4*>(base) + row * GNS_LANES + lane;
#endif
}
__device__ __forceinline__ f4* row_ptr(float* base, long long row, int lane) {
#if GNS_ROW_SADDR
  char* p = reinterpret_cast<char*>(base) + row * (GNS_LANES * 16);
  asm("" : "+s"(p));
  return reinterpret_cast<f4*>(p + (unsigned)(lane * 16));
#else
  return reinterpret_cast<f4*>(base) + row * GNS_LANES + lane;
#endif
}

// per-lane rows (the input packing kernel: a thread picks its own row)
__device__ __forceinline__ f4* row_ptr_lanewise(float* base, long long row, int lane) {
  return reinterpret_cast<f4*>(base) + row * GNS_LANES + lane;
}

// D floats (D even) as pairs: ceil(D/4) consecutive float4 rows (unused tail components are written as 0)
template <int D>
__device__ __forceinline__ void load_pairs(const float* base, long long row, int lane, f2 (&m)[D / 2]) {
  static_for<0, (D + 3) / 4>([&](auto q_) {
    constexpr int q = decltype(q_)::value;
    const f4 t = *row_ptr(base, row + q, lane);
    m[2 * q] = f2{t.x, t.y};
    if constexpr (2 * q + 1 < D / 2) m[2 * q + 1] = f2{t.z, t.w};
  });
}
// streaming variant: rows written once for a later kernel (the backward) and never re-read by this one
template <int D>
__device__ __forceinline__ void store_pairs_nt(float* base, long long row, int lane, const f2 (&m)[D / 2]) {
  static_for<0, (D + 3) / 4>([&](auto q_) {
    constexpr int q = decltype(q_)::value;
    f4 t = {m[2 * q].x, m[2 * q].y, 0.f, 0.f};
    if constexpr (2 * q + 1 < D / 2) { t.z = m[2 * q + 1].x; t.w = m[2 * q + 1].y; }
    __builtin_nontemporal_store(t, row_ptr(base, row + q, lane));
  });
}
template <int D>
__device__ __forceinline__ void store_pairs(float* base, long long row, int lane, const f2 (&m)[D / 2]) {
  static_for<0, (D + 3) / 4>([&](auto q_) {
    constexpr int q = decltype(q_)::value;
    f4 t = {m[2 * q].x, m[2 * q].y, 0.f, 0.f};
    if constexpr (2 * q + 1 < D / 2) { t.z = m[2 * q + 1].x; t.w = m[2 * q + 1].y; }
    *row_ptr(base, row + q, lane) = t;
  });
}
