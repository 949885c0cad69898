// Grid-per-workgroup reverse pass of the GNS K-step loop: what autograd does for total_loss.backward()
// (GNS/main.py:288) through GNS.forward (main.py:140-202), hand-derived, for gfx950 - see gns_gridwg.h.
//
// Same ownership as the grid-per-workgroup forward: a bus lane owns one bus of one grid and keeps its ADJOINT state
// (vbar, thetabar, dpbar, mbar[d]) in registers for all K reverse steps; an edge lane owns one line.  HBM traffic is
// the saved forward state read once per step plus the inputs; everything that crosses lanes goes through LDS.
// Per reverse step k:
//   P0   bus : close d total / d dp_{k+1} (loss term, main.py:198-199), reduce the adjoint of lambda, publish (v, theta, dpbar)
//   P1   edge: adjoints of the line physics (main.py:34-104) w.r.t. v, theta of the 2 (+4 bus-id-as-line-index) buses
//   P2   bus : gather them in fixed order; then, per phi family (L_m's first - its upstream is mbar_{k+1} itself):
//     B    bus : recompute L' from the saved state, back-propagate it, publish the adjoint of the hidden-vector sum and
//                the recomputed bus share of phi' (phi_head)
//     E    edge: recompute phi' of the line (phi_tail), back-propagate to the first-layer pre-activation g1, publish g1
//     B'   bus : sum g1 over the lines ending here; d/dm += W1[:, :d]^T G1 (phi's first layer is linear in m(dst))
// Weight gradients: row records in LDS -> v_mfma_f32_16x16x4_f32 (exact fp32) -> the wave's running sums in its slab.
// delta_q carries no gradient (identically zero as a function of v, theta: main.py:64-76 vs :83,98-103).
#include "gns_device.h"
#include "gns_gridwg.h"
#include "gns_dw.h"

namespace {
struct __attribute__((packed, aligned(4))) GbU4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) GbU2 { float x, y; };
__device__ __forceinline__ f4 gb_ld4(const float* p) { const GbU4 u = *reinterpret_cast<const GbU4*>(p); return f4{u.x, u.y, u.z, u.w}; }
__device__ __forceinline__ f2 gb_ld2(const float* p) { const GbU2 u = *reinterpret_cast<const GbU2*>(p); return f2{u.x, u.y}; }
__device__ __forceinline__ float gb_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float gb_yof(float r, float x) {
  return __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(x, x))));
}
// sin / cos on |x| <= pi/4 without range reduction (Cephes single-precision kernels, < 1 ulp)
__device__ __forceinline__ void gb_sincos_small(float x, float& s, float& c) {
  const float z = x * x;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  s = __builtin_fmaf(ps * z, x, x);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}
template <int NF>
__device__ __forceinline__ void gb_read_row(const float* row, f2 (&dst)[NF / 2]) {
#pragma unroll
  for (int i = 0; i < NF / 2; ++i) dst[i] = reinterpret_cast<const f2*>(row)[i];
}
template <int NF>
__device__ __forceinline__ void gb_write_row(float* row, const f2 (&src)[NF / 2]) {
#pragma unroll
  for (int i = 0; i < NF / 2; ++i) reinterpret_cast<f2*>(row)[i] = src[i];
}

// phi' backward on a line, hidden part only: g2 = gh * lrelu'(a2), g1 = (W2^T g2) * lrelu'(a1).  N-stream: W2n[H][H] first.
template <int H>
__device__ __forceinline__ void phi_bwd_hidden(cfp blk, const f2 (&a1)[H / 2], const f2 (&a2)[H / 2], const f2 (&gh)[H / 2],
                                               f2 (&g2)[H / 2], f2 (&g1)[H / 2]) {
#pragma unroll
  for (int u = 0; u < H / 2; ++u) g2[u] = gh[u] * dlrelu2(a2[u]);
  stream_pairs<H * H>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, j = w / H, i = (w % H) / 2;
    const f2 gj = splat(lane_of<j>(g2));
    g1[i] = (j == 0) ? s * gj : __builtin_elementwise_fma(s, gj, g1[i]);
  });
#pragma unroll
  for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * dlrelu2(a1[u]);
  pin_all(g1);
}
// d/dm += W1[:, 0..D)^T G1 from the N-stream's W1n[H][INP] (columns >= D are the line parameters: skipped)
template <int IN, int H, int D>
__device__ __forceinline__ void phi_bwd_latent(cfp blk, const f2 (&G1)[H / 2], f2 (&macc)[D / 2]) {
  constexpr int INP = IN + (IN & 1);
  stream_pairs<H * INP>(blk + H * H, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value, j = w / INP, i = (w % INP) / 2;
    if constexpr (2 * i < D) {
      const f2 gj = splat(lane_of<j>(G1));
      macc[i] = __builtin_elementwise_fma(s, gj, macc[i]);
    }
  });
  pin_all(macc);
}
constexpr int gb_max(int a, int b) { return a > b ? a : b; }
}  // namespace

template <int D, int H, bool MULTI>
struct GwBwdDims {
  using C = GnsDims<D, H, MULTI>;
  using EngE = GwEdgeEngine<C::PHI_IN, H, D>;
  using EngB = GwBusPhiEngine<C::PHI_IN, H, D>;
  static constexpr int RECF = GwSub::RECF;
};

template <int D, int H, bool MULTI, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) gns_gw_backward_kernel(GnsGwBwdArgs A) {
  using C = GnsDims<D, H, MULTI>;
  using BD = GwBwdDims<D, H, MULTI>;
  using EngE = typename BD::EngE;
  using EngB = typename BD::EngB;
  constexpr int NPHI = C::NPHI, MQ = C::MQ, HQ = C::HQ, SVQ = 1 + MQ, SSQ = NPHI * HQ;
  constexpr int XL = (C::LF_IN + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = A.N, E = A.E, K = A.K, Gn = A.Gn, WPG = A.WPG, P = A.P;
  const int gslot = wv / WPG, wig = wv - gslot * WPG;
  const int li = wig * 64 + lane;
  const bool bus_wave = wig * 64 < N, edge_wave = wig * 64 < E;
  const bool is_bus = li < N, is_edge = li < E;
  cip topo_s = (cip)A.topo;
  const int* topo = A.topo;
  cfp PT = (cfp)A.pt;
  cfp PN = (cfp)A.pn;

  // ---- per-lane topology --------------------------------------------------------------------------------------------
  int n = 0, p0 = 0, p1 = 0, q0 = 0, q1 = 0, i0 = 0, i1 = 0, g0 = 0, g1_ = 0, isgen = 0;
  if (is_bus) {
    n = topo[topo_s[TH_LANE_BUS] + li];
    p0 = topo[topo_s[TH_IN_PTR] + n];    p1 = topo[topo_s[TH_IN_PTR] + n + 1];
    q0 = topo[topo_s[TH_OUT_PTR] + n];   q1 = topo[topo_s[TH_OUT_PTR] + n + 1];
    i0 = topo[topo_s[TH_INCD_PTR] + n];  i1 = topo[topo_s[TH_INCD_PTR] + n + 1];
    g0 = topo[topo_s[TH_GEN_PTR] + n];   g1_ = topo[topo_s[TH_GEN_PTR] + n + 1];
    isgen = topo[topo_s[TH_IS_GEN] + n];
  }
  const int* q2p = topo + topo_s[TH_Q2P];
  const int* incd = topo + topo_s[TH_INCD];
  const int* gen_idx = topo + topo_s[TH_GEN_IDX];
  int e_id = 0, es = 0, et = 0, ia = 0, ib = 0, ic = 0, id = 0;
  if (is_edge) {
    e_id = topo[topo_s[TH_IN_EID] + li];
    es = topo[topo_s[TH_IN_SRC] + li];  et = topo[topo_s[TH_IN_DST] + li];
    ia = topo[topo_s[TH_IN_A] + li];    ib = topo[topo_s[TH_IN_B] + li];
    const int q = topo[topo_s[TH_P2Q] + li];
    ic = topo[topo_s[TH_OUT_C] + q];    id = topo[topo_s[TH_OUT_D] + q];
  }

  extern __shared__ __attribute__((aligned(16))) float gwb_lds_mem[];
  const GwBwdLds LY = gw_bwd_lds_layout(N, E, H, WPG, BD::RECF);
  float* Lb = gwb_lds_mem + (size_t)gslot * LY.total;
  float* plane3 = Lb + LY.plane3;
  float* slots = Lb + LY.slots;
  float* gS_l = Lb + LY.gS;
  float* u_l = Lb + LY.u;
  float* g1_l = Lb + LY.g1;
  float* red = Lb + LY.red;                                 // [0..2W): [par][w] lambda-adjoint partials, [2W..6W): gsum [4][w]
  float* rec = Lb + LY.rec + wig * BD::RECF;
  float* slab = A.slab + ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * A.slab_floats;
  const float invN = 1.0f / (float)N;

  const long long npacks = (A.Bt + P - 1) / P;
  for (long long pack = blockIdx.x; pack < npacks; pack += gridDim.x) {
    long long b = pack * P + gslot;
    const bool live = b < A.Bt;
    if (!live) b = A.Bt - 1;

    // ---- per-grid constants ---------------------------------------------------------------------------------------
    float Gs = 0.f, pmin = 0.f, pset = 0.f, pmax = 0.f;
    if (bus_wave) {
      float Pd = 0.f;
      if (is_bus) {
        const f4 bq = gb_ld4(A.buses + (b * N + n) * 6 + 2);
        Pd = bq.x; Gs = bq.z;
        for (int q = g0; q < g1_; ++q) {
          const float* r = A.gens + (b * Gn + gen_idx[q]) * 7;
          const f4 ra = gb_ld4(r + 1);
          pmax += ra.x; pmin += ra.y; pset += ra.z;
        }
      }
      const float r0 = gb_wave_sum(Pd), r1 = gb_wave_sum(pset), r2 = gb_wave_sum(pmin), r3 = gb_wave_sum(pmax);
      if (lane == 0) { red[2 * WPG + wig] = r0; red[3 * WPG + wig] = r1; red[4 * WPG + wig] = r2; red[5 * WPG + wig] = r3; }
    }
    f2 xt[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
    float ys = 0.f, taus = 1.f, shs = 0.f, yt = 0.f, taut = 1.f, sht = 0.f;
    if (edge_wave) {
      const float* lb = A.lines + b * (long long)E * 7;
      const f4 ea = gb_ld4(lb + e_id * 7 + 2);
      const float she = lb[e_id * 7 + 6];
      const f4 sa = gb_ld4(lb + es * 7 + 2);
      shs = lb[es * 7 + 6];
      const f4 ta = gb_ld4(lb + et * 7 + 2);
      sht = lb[et * 7 + 6];
      xt[0] = f2{ea.x, ea.y}; xt[1] = f2{ea.z, ea.w}; xt[2] = f2{she, 0.f};
      ys = gb_yof(sa.x, sa.y); taus = sa.w;
      yt = gb_yof(ta.x, ta.y); taut = ta.w;
    }
    const float gt = (live && A.g_total) ? A.g_total[b] : 0.f;
    const float gl = (live && A.g_last) ? A.g_last[b] : 0.f;
    const float gv_up = (live && is_bus && A.g_v) ? A.g_v[b * N + n] : 0.f;
    float vbar = 0.f, thbar = (live && is_bus && A.g_theta) ? A.g_theta[b * N + n] : 0.f, dpbar_in = 0.f;
    f2 mbar[D / 2];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) mbar[i] = f2{0.f, 0.f};
    __syncthreads();
    float gs1 = 0.f, gs2 = 0.f, gs3 = 0.f;                          // sumPset, sumPmin, sumPmax
    for (int w = 0; w * 64 < N; ++w) { gs1 += red[3 * WPG + w]; gs2 += red[4 * WPG + w]; gs3 += red[5 * WPG + w]; }

    const f4* SV = reinterpret_cast<const f4*>(A.sv_state);
    const f4* SS = reinterpret_cast<const f4*>(A.sv_S);
    for (int k = K - 1; k >= 0; --k) {
      const long long koff = k;
      const int par = k & 1;
      const f2 lamv = reinterpret_cast<const f2*>(A.sv_lam)[koff * A.Bt + b];
      const int bits = (int)lamv.y;
      const bool low1 = bits & 1, low2 = bits & 2;
      const float lden = low1 ? 2.f * (gs1 - gs2) : 2.f * (gs3 - gs1);      // d lambda / d p_global = 1 / lden (main.py:47-51)
      // ================= P0 ==========================================================================================
      f4 s1 = {0.f, 0.f, 0.f, 0.f};
      float dpb = 0.f;
      if (bus_wave) {
        if (is_bus) s1 = SV[(((koff + 1) * A.Bt + b) * SVQ) * N + li];   // (v, theta, dp, dq)_{k+1}
        if (k == K - 1) vbar = (s1.x < 0.f) ? 0.f : gv_up;               // v_out = where(v < 0, 0, v) (main.py:201)
        // d total / d dp_{k+1} = g_total * gamma^(K-k) * 2 dp / N  (+ g_last * 2 dp / N after the last step)  main.py:198-199
        const float cdp = 2.f * (gt * A.gw[k] + (k == K - 1 ? gl : 0.f)) * invN;
        dpb = dpbar_in + cdp * s1.z;
        const float lb = dpb * (low2 ? 2.f * (pset - pmin) : 2.f * (pmax - pset));    // d Pg_new / d lambda (main.py:53-57)
        if (is_bus) { plane3[3 * n] = s1.x; plane3[3 * n + 1] = s1.y; plane3[3 * n + 2] = dpb; }
        const float lsum = gb_wave_sum(is_bus ? lb : 0.f);
        if (lane == 0) red[par * WPG + wig] = lsum;
      }
      __syncthreads();
      float lbar = 0.f;
      for (int w = 0; w * 64 < N; ++w) lbar += red[par * WPG + w];
      const float pgbar = lbar / lden;
      // ================= P1: adjoints of the line physics ============================================================
      if (edge_wave) {
        const float vs = plane3[3 * es], ths = plane3[3 * es + 1], Tb = plane3[3 * es + 2];     // dp[s] += p_to   (main.py:95)
        const float vt = plane3[3 * et], tht = plane3[3 * et + 1], Fb = plane3[3 * et + 2];     // dp[t] += p_from (main.py:94)
        const float tha = plane3[3 * ia + 1], thb = plane3[3 * ib + 1], thc = plane3[3 * ic + 1], thd = plane3[3 * id + 1];
        const float dl = tha - thb, dl2 = thd - thc;
        float sA, cA, sB, cB, sD, cD, sC, cC, sD2, cD2;
        const float angA = ths - tht - dl - shs, angB = tht - ths - dl + shs, angC = tht - ths - dl2 - sht;
        const float amax = fmaxf(fmaxf(fmaxf(fabsf(angA), fabsf(angB)), fmaxf(fabsf(angC), fabsf(dl))), fabsf(dl2));
        if (__builtin_amdgcn_ballot_w64(!(amax <= 0.785f)) == 0) {     // every angle of the wave within pi/4: no range reduction
          gb_sincos_small(angA, sA, cA); gb_sincos_small(angB, sB, cB); gb_sincos_small(dl, sD, cD);
          gb_sincos_small(angC, sC, cC); gb_sincos_small(dl2, sD2, cD2);
        } else {
          sincosf(angA, &sA, &cA); sincosf(angB, &sB, &cB); sincosf(dl, &sD, &cD);
          sincosf(angC, &sC, &cC); sincosf(dl2, &sD2, &cD2);
        }
        // "from" expressions: p_from (main.py:91) and |msg| of the joule loss (main.py:41)
        const float yot = ys / taus, yot2 = ys / (taus * taus);
        const float base = vs * vt * yot;
        const float kJ = vs * yot2 + vt * vt * ys;
        const float inner = base * (sA + sB) + kJ * sD;
        const float Jb = pgbar * (inner > 0.f ? 1.f : (inner < 0.f ? -1.f : 0.f));
        float dvs = Fb * (vt * yot * sA + 2.f * vs * yot2 * sD) + Jb * (vt * yot * (sA + sB) + yot2 * sD);
        float dvt = Fb * (vs * yot * sA) + Jb * (vs * yot * (sA + sB) + 2.f * vt * ys * sD);
        const float Ab = (Fb + Jb) * base * cA, Bb = Jb * base * cB;
        const float dbar = Fb * (vs * vs * yot2) * cD + Jb * kJ * cD - Ab - Bb;
        float dths = Ab - Bb, dtht = Bb - Ab;
        // "to" expression: p_to (main.py:92)
        const float yot_t = yt / taut;
        const float base2 = vt * vs * yot_t;
        dvt += Tb * (vs * yot_t * sC + 2.f * vt * yt * sD2);
        dvs += Tb * (vt * yot_t * sC);
        const float Cb = Tb * base2 * cC;
        const float dbar2 = Tb * vt * vt * yt * cD2 - Cb;
        dtht += Cb; dths -= Cb;
        if (is_edge) {
          f2* sp = reinterpret_cast<f2*>(slots + 6 * li);
          sp[0] = f2{dvs, dvt}; sp[1] = f2{dths, dtht}; sp[2] = f2{dbar, dbar2};
        }
      }
      __syncthreads();
      // ================= P2: every bus completes d/d(v, theta)_{k+1} from the per-line adjoints =======================
      f2 s0v = {0.f, 0.f}, s0d = {0.f, 0.f};                         // (v, theta)_k, (dp, dq)_k
      f2 m[D / 2];
      f2 gx[XL];                                                     // adjoint of the L' input [v theta | dp dq | m | sum_e h_e | deg]
      f2 (&macc)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(gx[2]);
      f2 (&gSr)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(gx[2 + D / 2]);
      if (bus_wave) {
        for (int p = p0; p < p1; ++p) { vbar += slots[6 * p + 1]; thbar += slots[6 * p + 3]; }                 // lines ending here
        for (int q = q0; q < q1; ++q) { const int p = q2p[q]; vbar += slots[6 * p]; thbar += slots[6 * p + 2]; }   // lines leaving here
        for (int i = i0; i < i1; ++i) {                                                                         // angle-difference incidences
          const int code = incd[i];
          const float val = slots[6 * (code >> 2) + ((code & 2) ? 5 : 4)];
          thbar += (code & 1) ? -val : val;
        }
        if (is_bus) vbar += (pgbar - dpb) * (2.f * Gs * s1.x);     // -Gs v^2 in dp (main.py:82) and +Gs v^2 in p_global (main.py:45)
        if (is_bus) {
          const f4* sp = SV + ((koff * A.Bt + b) * SVQ) * N + li;
          const f4 r0 = sp[0];
          s0v = f2{r0.x, r0.y}; s0d = f2{r0.z, r0.w};
          static_for<0, MQ>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const f4 t = sp[(long long)(1 + q) * N];
            m[2 * q] = f2{t.x, t.y};
            if constexpr (2 * q + 1 < D / 2) m[2 * q + 1] = f2{t.z, t.w};
          });
        } else {
#pragma unroll
          for (int i = 0; i < D / 2; ++i) m[i] = f2{0.f, 0.f};
        }
        gx[0] = f2{0.f, 0.f}; gx[1] = f2{0.f, 0.f}; gx[XL - 1] = f2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < D / 2; ++i) macc[i] = mbar[i];            // identity path m_{k+1} = m_k + L_m(.) (main.py:188)
#pragma unroll
        for (int j = 0; j < H / 2; ++j) gSr[j] = f2{0.f, 0.f};
      }
      // ================= per phi family: B (bus), E (edge), B' (bus) ===================================================
      static_for<0, NPHI>([&](auto r_) {
        constexpr int pf = MULTI ? 2 - decltype(r_)::value : 0;      // phi_m, phi_theta, phi_v | the single phi
        // after the last step nothing reads m_K: L_m.{K-1} / phi_m.{K-1} get no gradient (reference: .grad is None)
        const bool skip_round = MULTI && pf == 2 && k == K - 1;
        if (bus_wave && !skip_round) {
          static_for<0, 3>([&](auto o_) {
            constexpr int l = (decltype(o_)::value == 0) ? 2 : decltype(o_)::value - 1;     // L_m first
            constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
            if constexpr (fphi == pf) {
              if (!(l == 2 && k == K - 1)) {
                constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1);
                f2 x[XL];
                f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(x[2 + D / 2]);
                if (is_bus) {
                  const f4* sp = SS + ((koff * A.Bt + b) * SSQ + fphi * HQ) * N + li;
                  static_for<0, HQ>([&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    const f4 t = sp[(long long)q * N];
                    S[2 * q] = f2{t.x, t.y};
                    if constexpr (2 * q + 1 < H / 2) S[2 * q + 1] = f2{t.z, t.w};
                  });
                } else {
#pragma unroll
                  for (int j = 0; j < H / 2; ++j) S[j] = f2{0.f, 0.f};
                }
                x[0] = s0v; x[1] = s0d;
#pragma unroll
                for (int i = 0; i < D / 2; ++i) x[2 + i] = m[i];
                x[XL - 1] = f2{(float)(p1 - p0), 0.f};
                f2 a1[H / 2], a2[H / 2], g3[OUTP / 2], g2[H / 2], g1[H / 2];
                mlp2_fwd<C::LF_IN, H>(PT + A.t_off[NPHI + l] + koff * A.t_sz[NPHI + l], x, a1, a2);
                if constexpr (l == 0) g3[0] = f2{thbar, 0.f};                               // theta += L_theta (main.py:182)
                else if constexpr (l == 1) g3[0] = f2{isgen ? 0.f : vbar, 0.f};              // v moves only without a generator (main.py:184-186)
                else {
#pragma unroll
                  for (int j = 0; j < D / 2; ++j) g3[j] = macc[j];                          // m += L_m (main.py:188)
                }
                if constexpr (MULTI) {
#pragma unroll
                  for (int j = 0; j < H / 2; ++j) gSr[j] = f2{0.f, 0.f};
                }
                mlp_bwd<C::LF_IN, H, OUTP, 2 * XL, true>(PN + A.n_off[NPHI + l] + koff * A.n_sz[NPHI + l], a1, a2, g3, g2, g1, gx);
                // ---- weight gradient of L'_l: sub-record windows -> matrix pipe -> the wave's running sums (gns_dw.h) ----
                {
                  constexpr int IN = C::LF_IN, ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
                  float* blk = slab + A.g_off[NPHI + l] + koff * A.g_sz[NPHI + l];
                  float Aop[16];
                  x[XL - 1].y = 1.f;                                         // column IN of the record: the 1 that yields db1
                  // dW1 | db1 = sum g1 (x) [x | 1]: three 16-column windows of the input
                  static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g1[j]); });
                  static_for<0, (2 * XL + 15) / 16>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < XL) gws_putB(rec, lane, j, x[8 * t + j]); });
                    gws_w2r();
                    f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (t == 0) gws_pass<true>(rec, lane, Aop, Dt); else gws_pass<false>(rec, lane, Aop, Dt);
                    gws_r2w();
                    gws_flush(lane, Dt, blk, [&](int c, int il) { const int i = 16 * t + il; return c < H ? (i < IN ? c * IN + i : (i == IN ? ob1 + c : -1)) : -1; });
                  });
                  // dW2 | db2 = sum g2 (x) [a1 | 1]
                  static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g2[j]); gws_putB(rec, lane, j, a1[j]); });
                  gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
                  gws_w2r();
                  {
                    f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                    gws_pass<true>(rec, lane, Aop, Dt);
                    gws_r2w();
                    gws_flush(lane, Dt, blk, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; });
                  }
                  // dW4 | db4 = sum g3 (x) [a2 | 1], 12 output rows per window
                  static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB(rec, lane, j, a2[j]); });
                  static_for<0, (OUTP + 11) / 12>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    static_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (6 * t + j < OUTP / 2) gws_putA(rec, lane, j, g3[6 * t + j]); });
                    gws_w2r();
                    f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                    gws_pass<true>(rec, lane, Aop, Dt);
                    gws_r2w();
                    gws_flush(lane, Dt, blk, [&](int c, int il) { const int j = 12 * t + c; return (c < 12 && j < OUT) ? (il < H ? oW4 + j * H + il : (il == H ? ob4 + j : -1)) : -1; });
                  });
                }
              }
            }
          });
          if (is_bus) gb_write_row<H>(gS_l + n * H, gSr);
          f2 uh[H / 2];
          phi_head<D, H>(PT + A.t_off[pf] + koff * A.t_sz[pf], m, uh);
          if (is_bus) gb_write_row<H>(u_l + n * H, uh);
        }
        __syncthreads();
        if (edge_wave && !skip_round) {
          f2 uh[H / 2], gh[H / 2], a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
          gb_read_row<H>(u_l + et * H, uh);
          gb_read_row<H>(gS_l + et * H, gh);
          if (!is_edge) {
#pragma unroll
            for (int j = 0; j < H / 2; ++j) gh[j] = f2{0.f, 0.f};
          }
          phi_tail<C::PHI_IN, H, D>(PT + A.t_off[pf] + koff * A.t_sz[pf], uh, xt, a1, a2);
          phi_bwd_hidden<H>(PN + A.n_off[pf] + koff * A.n_sz[pf], a1, a2, gh, g2, g1);
          if (is_edge) gb_write_row<H>(g1_l + li * H, g1);
          {   // phi' line columns of dW1, db1, dW2, db2 (folded block W1[H][IN] b1[H] W2[H][H] b2[H])
            constexpr int IN = C::PHI_IN, ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H;
            float* blk = slab + A.g_off[pf] + koff * A.g_sz[pf];
            float Aop[16];
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g1[j]); });
            gws_putB(rec, lane, 0, xt[0]); gws_putB(rec, lane, 1, xt[1]); gws_putB(rec, lane, 2, f2{xt[2].x, 1.f});
            gws_w2r();
            f32x4 D1 = {0.f, 0.f, 0.f, 0.f};
            gws_pass<true>(rec, lane, Aop, D1);
            gws_r2w();
            gws_flush(lane, D1, blk, [&](int c, int il) { return c < H ? (il < IN - D ? c * IN + D + il : (il == IN - D ? ob1 + c : -1)) : -1; });
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g2[j]); gws_putB(rec, lane, j, a1[j]); });
            gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
            gws_w2r();
            f32x4 D2 = {0.f, 0.f, 0.f, 0.f};
            gws_pass<true>(rec, lane, Aop, D2);
            gws_r2w();
            gws_flush(lane, D2, blk, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; });
          }
        }
        __syncthreads();
        if (bus_wave && !skip_round) {
          f2 G1[H / 2];
#pragma unroll
          for (int j = 0; j < H / 2; ++j) G1[j] = f2{0.f, 0.f};
          for (int p = p0; p < p1; ++p) {
            const f2* gr = reinterpret_cast<const f2*>(g1_l + p * H);
#pragma unroll
            for (int j = 0; j < H / 2; ++j) G1[j] += gr[j];
          }
          phi_bwd_latent<C::PHI_IN, H, D>(PN + A.n_off[pf] + koff * A.n_sz[pf], G1, macc);
          {   // latent columns of phi's dW1: sum over buses G1 (x) m
            constexpr int IN = C::PHI_IN;
            float* blk = slab + A.g_off[pf] + koff * A.g_sz[pf];
            float Aop[16];
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, G1[j]); });
            static_for<0, (D + 15) / 16>([&](auto t_) {
              constexpr int t = decltype(t_)::value;
              static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < D / 2) gws_putB(rec, lane, j, m[8 * t + j]); });
              gws_w2r();
              f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
              if constexpr (t == 0) gws_pass<true>(rec, lane, Aop, Dt); else gws_pass<false>(rec, lane, Aop, Dt);
              gws_r2w();
              gws_flush(lane, Dt, blk, [&](int c, int il) { const int i = 16 * t + il; return (c < H && i < D) ? c * IN + i : -1; });
            });
          }
        }
      });
      // ---- the adjoints entering step k (identity paths main.py:182,186,188 + what the L' inputs collected) --------
      if (bus_wave) {
        vbar += gx[0].x; thbar += gx[0].y; dpbar_in = gx[1].x;
#pragma unroll
        for (int i = 0; i < D / 2; ++i) mbar[i] = macc[i];
      }
    }
    __syncthreads();     // the next pack's prologue rewrites the gsum partials
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
namespace {
constexpr int GWB_LDS_MAX_BYTES = 160 * 1024;
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gwb_launch_t(const GnsGwBwdArgs& A, int blocks, int threads, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((gns_gw_backward_kernel<D, H, MULTI, MAXT, MINW>), dim3(blocks), dim3(threads), lds, st, A);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gwb_attr_t() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gns_gw_backward_kernel<D, H, MULTI, MAXT, MINW>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, GWB_LDS_MAX_BYTES) == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
int recf_of(int d, int h, int multi) {
#define GNS_CASE(DD, HH) if (d == DD && h == HH) return multi ? GwBwdDims<DD, HH, true>::RECF : GwBwdDims<DD, HH, false>::RECF;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return -1;
}
int g_gwb_cus = 0;
}  // namespace

int gns_gw_backward_init_device(void) {
  int rc = GNS_OK;
#define GNS_CASE(DD, HH)                                                                                          \
  if (gwb_attr_t<DD, HH, true, 256, 2>() != GNS_OK || gwb_attr_t<DD, HH, false, 256, 2>() != GNS_OK ||              \
      gwb_attr_t<DD, HH, true, 512, 1>() != GNS_OK || gwb_attr_t<DD, HH, false, 512, 1>() != GNS_OK ||              \
      gwb_attr_t<DD, HH, true, 1024, 1>() != GNS_OK || gwb_attr_t<DD, HH, false, 1024, 1>() != GNS_OK) rc = GNS_ELAUNCH;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  g_gwb_cus = cus;
  (void)hipGetLastError();
  return rc;
}

int gns_gw_backward_supported(int N, int E, int d, int h, int multi, int P) {
  const int recf = recf_of(d, h, multi);
  if (recf < 0 || P < 1) return 0;
  const int WPG = ((N > E ? N : E) + 63) / 64;
  if (WPG * P > 16) return 0;
  const GwBwdLds L = gw_bwd_lds_layout(N, E, h, WPG, recf);
  return (size_t)L.total * 4 * P <= (size_t)GWB_LDS_MAX_BYTES ? 1 : 0;
}

int gns_gw_backward_blocks(int N, int E, int d, int h, int multi, int P, long long Bt) {
  const int recf = recf_of(d, h, multi);
  if (recf < 0) return 0;
  const int WPG = ((N > E ? N : E) + 63) / 64;
  const GwBwdLds L = gw_bwd_lds_layout(N, E, h, WPG, recf);
  const size_t lds = (size_t)L.total * 4 * P;
  int per_cu = (int)(GWB_LDS_MAX_BYTES / (lds > 0 ? lds : 1));
  const int by_waves = 8 / (P * WPG) > 0 ? 8 / (P * WPG) : 1;            // 256 VGPRs: 2 waves per SIMD
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  const long long npacks = (Bt + P - 1) / P;
  const long long cap = (long long)(g_gwb_cus > 0 ? g_gwb_cus : 256) * per_cu;
  return (int)(npacks < cap ? npacks : cap);
}

int gns_gw_launch_backward(int d, int h, int multi, const GnsGwBwdArgs& A, int blocks, hipStream_t st) {
  const int threads = A.P * A.WPG * 64;
  const int recf = recf_of(d, h, multi);
  if (recf < 0) return GNS_EUNSUPPORTED;
  const GwBwdLds L = gw_bwd_lds_layout(A.N, A.E, h, A.WPG, recf);
  const size_t lds = (size_t)L.total * 4 * A.P;
  if (threads > 1024 || lds > (size_t)GWB_LDS_MAX_BYTES || blocks < 1) return GNS_EUNSUPPORTED;
#define GNS_CASE(DD, HH)                                                                                          \
  if (d == DD && h == HH) {                                                                                       \
    if (threads <= 256) return multi ? gwb_launch_t<DD, HH, true, 256, 2>(A, blocks, threads, lds, st)             \
                                     : gwb_launch_t<DD, HH, false, 256, 2>(A, blocks, threads, lds, st);           \
    if (threads <= 512) return multi ? gwb_launch_t<DD, HH, true, 512, 1>(A, blocks, threads, lds, st)             \
                                     : gwb_launch_t<DD, HH, false, 512, 1>(A, blocks, threads, lds, st);           \
    return multi ? gwb_launch_t<DD, HH, true, 1024, 1>(A, blocks, threads, lds, st)                                \
                 : gwb_launch_t<DD, HH, false, 1024, 1>(A, blocks, threads, lds, st);                              \
  }
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return GNS_EUNSUPPORTED;
}
