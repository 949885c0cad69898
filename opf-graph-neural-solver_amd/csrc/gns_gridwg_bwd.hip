// Grid-per-workgroup reverse pass of the GNS K-step loop: what autograd does for total_loss.backward()
// (GNS/main.py:288) through GNS.forward (main.py:140-202), hand-derived, for gfx950 - see gns_gridwg.h.
//
// Ownership: a grid belongs to ceil(N/64) waves.  A lane is a BUS lane (one bus of the grid, its adjoint state
// vbar / thetabar / dpbar / mbar[d] stays in registers for all K reverse steps) and, in the edge phases, an EDGE lane for
// up to two lines (line i and line i + 64*waves): every wave is busy in every phase.  HBM traffic is the saved forward
// state read once per step plus the inputs; everything that crosses lanes goes through LDS.
// Per reverse step k:
//   P0   bus : close d total / d dp_{k+1} (loss term, main.py:198-199), reduce the adjoint of lambda, publish (v, theta, dpbar)
//   P1   edge: adjoints of the line physics (main.py:34-104) w.r.t. v, theta of the 2 (+4 bus-id-as-line-index) buses
//   P2   bus : gather them in fixed order; then, per phi family (L_m's first - its upstream is mbar_{k+1} itself):
//     B    bus : recompute L' from the saved state and back-propagate it layer by layer; each layer's weight gradient is
//                contracted as soon as its operands exist (sub-record windows, gns_dw.h), the input adjoints come out of
//                an input-major weight stream four at a time and go straight to their consumers (the latent adjoint, the
//                LDS row of the hidden-sum adjoint) - no 36-register adjoint array; the bus share of phi' is recomputed
//     E    edge: recompute phi' of the line (phi_tail), back-propagate to the first-layer pre-activation g1, publish g1
//     B'   bus : sum g1 over the lines ending here; d/dm += W1[:, :d]^T G1 (phi's first layer is linear in m(dst))
// Weight gradients: matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32) -> a per-wave LDS stage -> the wave's running sums
// in its slab.  The slab read-modify-write is split around the phase barrier: loads before it, add + store after it.
// delta_q carries no gradient (identically zero as a function of v, theta: main.py:64-76 vs :83,98-103).
#include "gns_device.h"
#include "gns_gridwg.h"
#ifndef GNS_GWB_KEEP_SLOPES
#define GNS_GWB_KEEP_SLOPES 1     // as GNS_BWDS_KEEP_SLOPES (gns_backward_split.hip): the slopes of the hidden units are kept from the recomputation
#endif
#if GNS_GWB_KEEP_SLOPES
#define GWB_SLOPES(sl) sl
#define GWB_SLOPE_OF(sl, layer, act, u) sl[layer][u]
#else
#define GWB_SLOPES(sl) nullptr
#define GWB_SLOPE_OF(sl, layer, act, u) dlrelu2(act[u])
#endif
#include "gns_dw.h"

namespace {
struct __attribute__((packed, aligned(4))) GbU4 { float x, y, z, w; };
__device__ __forceinline__ f4 gb_ld4(const float* p) { const GbU4 u = *reinterpret_cast<const GbU4*>(p); return f4{u.x, u.y, u.z, u.w}; }
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

constexpr int gb_max(int a, int b) { return a > b ? a : b; }
}  // namespace

template <int D, int H, bool MULTI>
struct GwBwdDims {
  using C = GnsDims<D, H, MULTI>;
  static constexpr int LIN = C::LF_IN;
  static_assert(LIN % 2 == 1, "the bias column of the first-layer record is the free half of the last input pair");
  static constexpr int GL_M = LIN * H + H + H * H + H + D * H + D;        // folded gradient block of L_m (the largest)
  static constexpr int GP = C::PHI_IN * H + H + H * H + H;               // folded gradient block of phi'
  static constexpr int STG_L = (gb_max(GL_M, GP) + 63) / 64 * 64;        // per-wave stage: an L' (or the line part of a phi') block
  static constexpr int STG_P = (H * D + 63) / 64 * 64;                   //                 the latent columns of phi' dW1
  static constexpr int STGF = STG_L + STG_P;
  static constexpr int RECF = GwSub::RECF;
};

template <int D, int H, bool MULTI, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) gns_gw_backward_kernel(GnsGwBwdArgs A) {
  using C = GnsDims<D, H, MULTI>;
  using BD = GwBwdDims<D, H, MULTI>;
  constexpr int NPHI = C::NPHI, MQ = C::MQ, HQ = C::HQ, SVQ = 1 + MQ, SSQ = NPHI * HQ;
  constexpr int LIN = C::LF_IN, XL = (LIN + 1) / 2, PIN = C::PHI_IN;
  constexpr int SOFF = 2 + D / 2;                       // first pair of the hidden-vector sum inside the L' input
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = A.N, E = A.E, K = A.K, Gn = A.Gn, WPG = A.WPG, P = A.P;
  const int gslot = wv / WPG, wig = wv - gslot * WPG;
  const int li = wig * 64 + lane;
  const int WL = WPG * 64;                              // lanes of one grid = lines per edge pass
  const bool bus_wave = wig * 64 < N;
  const bool is_bus = li < N;
  const bool e_wave[2] = {wig * 64 < E, wig * 64 + WL < E};      // wave-uniform: this wave holds lines in pass 0 / 1
  const bool is_e[2] = {li < E, li + WL < E};
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
  // edge lane: line p = li + pass * (lanes per grid) in destination order.  Kept in registers for the whole grid: the
  // destination bus and the line's own parameters (phi' needs them in every family round); the operands of the line
  // physics (one use per step) are re-read from the topology blob / the caller's tensor / the LDS table of y in P1.
  int e_id[2] = {0, 0}, et[2] = {0, 0};
#pragma unroll
  for (int ep = 0; ep < 2; ++ep) {
    if (is_e[ep]) {
      e_id[ep] = topo[topo_s[TH_IN_EID] + li + ep * WL];
      et[ep] = topo[topo_s[TH_IN_DST] + li + ep * WL];
    }
  }
  const int* t_src = topo + topo_s[TH_IN_SRC];
  const int* t_a = topo + topo_s[TH_IN_A];
  const int* t_b = topo + topo_s[TH_IN_B];
  const int* t_p2q = topo + topo_s[TH_P2Q];
  const int* t_c = topo + topo_s[TH_OUT_C];
  const int* t_d = topo + topo_s[TH_OUT_D];

  extern __shared__ __attribute__((aligned(16))) float gwb_lds_mem[];
  const GwBwdLds LY = gw_bwd_lds_layout(N, E, H, WPG, BD::RECF, BD::STGF);
  float* Lb = gwb_lds_mem + (size_t)gslot * LY.total;
  float* plane3 = Lb + LY.plane3;
  float* slots = Lb + LY.slots;
  float* gS_l = Lb + LY.gS;
  float* u_l = Lb + LY.u;
  float* g1_l = Lb + LY.g1;
  float* red = Lb + LY.red;                                 // [0..2W): [par][w] lambda-adjoint partials, [2W..6W): gsum [4][w]
  float* rec = Lb + LY.rec + wig * BD::RECF;
  float* stgL = Lb + LY.stage + wig * BD::STGF;             // stage of an L' block / of the line part of a phi' block
  float* ylds = Lb + LY.ylds;                               // y = 1 / sqrt(r^2 + x^2) per line NUMBER (main.py:38), once per grid
  float* stgP = stgL + BD::STG_L;                           // stage of the latent columns of phi' dW1
  float* slab = A.slab + ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * A.slab_floats;
  const float invN = 1.0f / (float)N;

  // ---- deferred read-modify-write of the slab: loads before a barrier, add + store after it ---------------------------
  constexpr int NRB = BD::STG_L / 64, NRP = BD::STG_P / 64, NRE = (BD::GP + 63) / 64;
  float Rb[NRB], Rp[NRP], Re[NRE];
  float* rb_blk = nullptr; float* rp_blk = nullptr; float* re_blk = nullptr;     // wave-uniform: pending flush target (nullptr = none)
  int rb_n = 0;
  // latent columns of phi' W1: t -> c * PIN + i, t = c * D + i;   line part: t < H*(PIN-D) -> c * PIN + D + i, then the rest of the block
  auto map_p = [&](int t) { const int c = t / D; return c * PIN + (t - c * D); };
  auto map_e = [&](int t) { constexpr int NX = PIN - D; if (t < H * NX) { const int c = t / NX; return c * PIN + D + (t - c * NX); } return PIN * H + (t - H * NX); };
  constexpr int NE_T = BD::GP - H * D;                       // entries of a phi' block the edge lanes own

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
    f2 xt[2][3];
    const float* lb = A.lines + b * (long long)E * 7;
#pragma unroll
    for (int ep = 0; ep < 2; ++ep) {
      xt[ep][0] = f2{0.f, 0.f}; xt[ep][1] = f2{0.f, 0.f}; xt[ep][2] = f2{0.f, 0.f};
      if (e_wave[ep]) {
        const f4 ea = gb_ld4(lb + e_id[ep] * 7 + 2);                    // r, x, b, tau of the lane's own line
        const float she = lb[e_id[ep] * 7 + 6];
        xt[ep][0] = f2{ea.x, ea.y}; xt[ep][1] = f2{ea.z, ea.w}; xt[ep][2] = f2{she, 0.f};
        if (is_e[ep]) ylds[e_id[ep]] = gb_yof(ea.x, ea.y);
      }
    }
    const float gt = (live && A.g_total) ? A.g_total[b] : 0.f;
    const float gl = (live && A.g_last) ? A.g_last[b] : 0.f;
    const float gv_up = (live && is_bus && A.g_v) ? A.g_v[b * N + n] : 0.f;
    float vbar = 0.f, thbar = (live && is_bus && A.g_theta) ? A.g_theta[b * N + n] : 0.f, dpbar_in = 0.f;
    f2 macc[D / 2];                                      // adjoint of the latent vector: mbar_{k+1} on entry of step k, mbar_k on exit
#pragma unroll
    for (int i = 0; i < D / 2; ++i) macc[i] = f2{0.f, 0.f};
    __syncthreads();
    float gs1 = 0.f, gs2 = 0.f, gs3 = 0.f;                          // sumPset, sumPmin, sumPmax
    for (int w = 0; w * 64 < N; ++w) { gs1 += red[3 * WPG + w]; gs2 += red[4 * WPG + w]; gs3 += red[5 * WPG + w]; }

    const f4* SV = reinterpret_cast<const f4*>(A.sv_state);
    const f4* SS = reinterpret_cast<const f4*>(A.sv_S);
    f2 xs[XL];                                           // the L' input of the step being reversed: [v theta | dp dq | m | sum_e h_e | deg, 1]
    for (int k = K - 1; k >= 0; --k) {
      const long long koff = k;
      const int par = k & 1;
      const f2 lamv = reinterpret_cast<const f2*>(A.sv_lam)[koff * A.Bt + b];
      const int bits = (int)lamv.y;
      const bool low1 = bits & 1, low2 = bits & 2;
      const float lden = low1 ? 2.f * (gs1 - gs2) : 2.f * (gs3 - gs1);      // d lambda / d p_global = 1 / lden (main.py:47-51)
      // the saved state of this step is requested now and first needed two barriers from here
      f4 s1 = {0.f, 0.f, 0.f, 0.f};
      if (bus_wave) {
        if (is_bus) {
          s1 = SV[(((koff + 1) * A.Bt + b) * SVQ) * N + li];           // (v, theta, dp, dq)_{k+1}
          const f4* sp = SV + ((koff * A.Bt + b) * SVQ) * N + li;
          const f4 r0 = sp[0];
          xs[0] = f2{r0.x, r0.y}; xs[1] = f2{r0.z, r0.w};
          static_for<0, MQ>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const f4 t = sp[(long long)(1 + q) * N];
            xs[2 + 2 * q] = f2{t.x, t.y};
            if constexpr (2 * q + 1 < D / 2) xs[2 + 2 * q + 1] = f2{t.z, t.w};
          });
        } else {
#pragma unroll
          for (int i = 0; i < 2 + D / 2; ++i) xs[i] = f2{0.f, 0.f};
        }
      }
      // ================= P0 ==========================================================================================
      float dpb = 0.f;
      if (bus_wave) {
        if (rp_blk) {                                                      // latent columns of the previous step's last phi' round
#pragma unroll
          for (int i = 0; i < NRP; ++i) { const int t = lane + 64 * i; if (t < H * D) rp_blk[map_p(t)] = Rp[i] + stgP[t]; }
          rp_blk = nullptr;
        }
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
#pragma unroll
      for (int ep = 0; ep < 2; ++ep) {
        if (!e_wave[ep]) continue;
        const int pp = is_e[ep] ? li + ep * WL : 0;
        const int S_ = t_src[pp], T_ = et[ep], q_ = t_p2q[pp];
        const int ia_ = t_a[pp], ib_ = t_b[pp], ic_ = t_c[q_], id_ = t_d[q_];
        // the reference gathers y, tau, shift at LINE NUMBER s = src[e] and t = dst[e] (bus ids used as line indices, main.py:41,70-72)
        const float ys_ = ylds[S_], yt_ = ylds[T_];
        const float taus_ = lb[S_ * 7 + 5], shs_ = lb[S_ * 7 + 6], taut_ = lb[T_ * 7 + 5], sht_ = lb[T_ * 7 + 6];
        const float vs = plane3[3 * S_], ths = plane3[3 * S_ + 1], Tb = plane3[3 * S_ + 2];     // dp[s] += p_to   (main.py:95)
        const float vt = plane3[3 * T_], tht = plane3[3 * T_ + 1], Fb = plane3[3 * T_ + 2];     // dp[t] += p_from (main.py:94)
        const float tha = plane3[3 * ia_ + 1], thb = plane3[3 * ib_ + 1], thc = plane3[3 * ic_ + 1], thd = plane3[3 * id_ + 1];
        const float dl = tha - thb, dl2 = thd - thc;
        float sA, cA, sB, cB, sD, cD, sC, cC, sD2, cD2;
        const float angA = ths - tht - dl - shs_, angB = tht - ths - dl + shs_, angC = tht - ths - dl2 - sht_;
        const float amax = fmaxf(fmaxf(fmaxf(fabsf(angA), fabsf(angB)), fmaxf(fabsf(angC), fabsf(dl))), fabsf(dl2));
        if (__builtin_amdgcn_ballot_w64(!(amax <= 0.785f)) == 0) {     // every angle of the wave within pi/4: no range reduction
          gb_sincos_small(angA, sA, cA); gb_sincos_small(angB, sB, cB); gb_sincos_small(dl, sD, cD);
          gb_sincos_small(angC, sC, cC); gb_sincos_small(dl2, sD2, cD2);
        } else {
          sincosf(angA, &sA, &cA); sincosf(angB, &sB, &cB); sincosf(dl, &sD, &cD);
          sincosf(angC, &sC, &cC); sincosf(dl2, &sD2, &cD2);
        }
        // "from" expressions: p_from (main.py:91) and |msg| of the joule loss (main.py:41)
        const float yot = ys_ / taus_, yot2 = ys_ / (taus_ * taus_);
        const float base = vs * vt * yot;
        const float kJ = vs * yot2 + vt * vt * ys_;
        const float inner = base * (sA + sB) + kJ * sD;
        const float Jb = pgbar * (inner > 0.f ? 1.f : (inner < 0.f ? -1.f : 0.f));
        float dvs = Fb * (vt * yot * sA + 2.f * vs * yot2 * sD) + Jb * (vt * yot * (sA + sB) + yot2 * sD);
        float dvt = Fb * (vs * yot * sA) + Jb * (vs * yot * (sA + sB) + 2.f * vt * ys_ * sD);
        const float Ab = (Fb + Jb) * base * cA, Bb = Jb * base * cB;
        const float dbar = Fb * (vs * vs * yot2) * cD + Jb * kJ * cD - Ab - Bb;
        float dths = Ab - Bb, dtht = Bb - Ab;
        // "to" expression: p_to (main.py:92)
        const float yot_t = yt_ / taut_;
        const float base2 = vt * vs * yot_t;
        dvt += Tb * (vs * yot_t * sC + 2.f * vt * yt_ * sD2);
        dvs += Tb * (vt * yot_t * sC);
        const float Cb = Tb * base2 * cC;
        const float dbar2 = Tb * vt * vt * yt_ * cD2 - Cb;
        dtht += Cb; dths -= Cb;
        if (is_e[ep]) {
          f2* sp = reinterpret_cast<f2*>(slots + 6 * (li + ep * WL));
          sp[0] = f2{dvs, dvt}; sp[1] = f2{dths, dtht}; sp[2] = f2{dbar, dbar2};
        }
      }
      __syncthreads();
      // ================= P2: every bus completes d/d(v, theta)_{k+1} from the per-line adjoints =======================
      float xsv = 0.f, xsth = 0.f, xsdp = 0.f;                       // adjoints of (v, theta, dp)_k collected from the L' inputs
      f2 gSacc[H / 2];                                               // single phi: adjoint of the one hidden sum over the three L nets
#pragma unroll
      for (int j = 0; j < H / 2; ++j) gSacc[j] = f2{0.f, 0.f};
      if (bus_wave) {
        for (int p = p0; p < p1; ++p) { vbar += slots[6 * p + 1]; thbar += slots[6 * p + 3]; }                 // lines ending here
        for (int q = q0; q < q1; ++q) { const int p = q2p[q]; vbar += slots[6 * p]; thbar += slots[6 * p + 2]; }   // lines leaving here
        for (int i = i0; i < i1; ++i) {                                                                         // angle-difference incidences
          const int code = incd[i];
          const float val = slots[6 * (code >> 2) + ((code & 2) ? 5 : 4)];
          thbar += (code & 1) ? -val : val;
        }
        if (is_bus) vbar += (pgbar - dpb) * (2.f * Gs * s1.x);     // -Gs v^2 in dp (main.py:82) and +Gs v^2 in p_global (main.py:45)
        xs[XL - 1] = f2{(float)(p1 - p0), 1.f};                     // deg, and the 1 whose column of dW1 is db1
      }
      // ================= per phi family: B (bus), E (edge), B' (bus) ===================================================
      static_for<0, NPHI>([&](auto r_) {
        constexpr int pf = MULTI ? 2 - decltype(r_)::value : 0;      // phi_m, phi_theta, phi_v | the single phi
        // after the last step nothing reads m_K: L_m.{K-1} / phi_m.{K-1} get no gradient (reference: .grad is None)
        const bool skip_round = MULTI && pf == 2 && k == K - 1;
        // The weights of a (family, step) are streamed exactly once by every wave, so the scalar cache is cold for all of them and a
        // stream keeps only two lines in flight: one wave of the workgroup pulls the round's four blocks (L' and phi', forward and
        // data-gradient layouts, ~7 KB) in one burst (gns_device.h, scalar_cache_warm); its partners' streams then hit lines that
        // are at least in flight.
        if (!skip_round && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
          constexpr int lw = MULTI ? (pf == 2 ? 2 : (pf == 1 ? 0 : 1)) : 0;          // the L net that reads this phi (single phi: L_theta first)
          scalar_cache_warm(PT + A.t_off[NPHI + lw] + koff * A.t_sz[NPHI + lw], A.t_sz[NPHI + lw]);
          scalar_cache_warm(PN + A.n_off[NPHI + lw] + koff * A.n_sz[NPHI + lw], A.n_sz[NPHI + lw]);
          scalar_cache_warm(PT + A.t_off[pf] + koff * A.t_sz[pf], A.t_sz[pf]);
          scalar_cache_warm(PN + A.n_off[pf] + koff * A.n_sz[pf], A.n_sz[pf]);
          if constexpr (!MULTI) {                                                      // the single phi serves all three L nets in this one round
            scalar_cache_warm(PT + A.t_off[NPHI + 1] + koff * A.t_sz[NPHI + 1], A.t_sz[NPHI + 1]);
            scalar_cache_warm(PN + A.n_off[NPHI + 1] + koff * A.n_sz[NPHI + 1], A.n_sz[NPHI + 1]);
            scalar_cache_warm(PT + A.t_off[NPHI + 2] + koff * A.t_sz[NPHI + 2], A.t_sz[NPHI + 2]);
            scalar_cache_warm(PN + A.n_off[NPHI + 2] + koff * A.n_sz[NPHI + 2], A.n_sz[NPHI + 2]);
          }
        }
        if (bus_wave && !skip_round) {
          static_for<0, 3>([&](auto o_) {
            constexpr int l = (decltype(o_)::value == 0) ? 2 : decltype(o_)::value - 1;     // L_m first
            constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
            if constexpr (fphi == pf) {
              if (!(l == 2 && k == K - 1)) {
                constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1);
                using NL = NLay<LIN, H, OUTP>;
                constexpr int ob1 = LIN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H, GSZ = ob4 + OUT;
                cfp nb = PN + A.n_off[NPHI + l] + koff * A.n_sz[NPHI + l];
                f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(xs[SOFF]);
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
                if (rb_blk) {                                           // single phi: the previous L net's block is still in flight
#pragma unroll
                  for (int i = 0; i < NRB; ++i) { const int t = lane + 64 * i; if (t < rb_n) rb_blk[t] = Rb[i] + stgL[t]; }
                  rb_blk = nullptr;
                  gws_r2w();
                }
                f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
                // only the hidden activations are needed: the T-stream of a three-layer block starts with the two-layer layout
                f2 sl[2][H / 2];       // LeakyReLU's slopes, kept from the recomputation (GNS_GWB_KEEP_SLOPES; gns_device.h, phi_tail)
                mlp2_fwd<LIN, H>(PT + A.t_off[NPHI + l] + koff * A.t_sz[NPHI + l], xs, a1, a2, NoBG{}, NoLink{}, GWB_SLOPES(sl));
                // ---- output layer: g2 = (W4^T g3) * lrelu'(a2);  dW4 | db4 = sum g3 (x) [a2 | 1]
                if constexpr (l == 2) {
                  bwd_rows<OUTP, H>(nb, macc, g2);                                          // g3 = mbar_{k+1}: m += L_m (main.py:188)
                } else {
                  const f2 g3s[1] = {f2{l == 0 ? thbar : (isgen ? 0.f : vbar), 0.f}};      // theta += L_theta (:182); v moves only without a generator (:184-186)
                  bwd_rows<2, H>(nb, g3s, g2);
                }
#pragma unroll
                for (int u = 0; u < H / 2; ++u) g2[u] = g2[u] * GWB_SLOPE_OF(sl, 1, a2, u);
                static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB(rec, lane, j, a2[j]); });
                gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
                static_for<0, (OUTP + 11) / 12>([&](auto t_) {
                  constexpr int t = decltype(t_)::value;
                  if constexpr (l == 2) {
                    static_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (6 * t + j < D / 2) gws_putA(rec, lane, j, macc[6 * t + j]); });
                  } else {
                    gws_putA(rec, lane, 0, f2{l == 0 ? thbar : (isgen ? 0.f : vbar), 0.f});
                  }
                  gws_w2r();
                  f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                  gws_pass(rec, lane, Dt);
                  gws_r2w();
                  gws_stage(lane, Dt, stgL, [&](int c, int il) { const int j = 12 * t + c; return (c < 12 && j < OUT) ? (il < H ? oW4 + j * H + il : (il == H ? ob4 + j : -1)) : -1; });
                });
                // ---- hidden layer: g1 = (W2^T g2) * lrelu'(a1);  dW2 | db2 = sum g2 (x) [a1 | 1]
                bwd_rows<H, H>(nb + NL::oW2, g2, g1);
#pragma unroll
                for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * GWB_SLOPE_OF(sl, 0, a1, u);
                static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g2[j]); gws_putB(rec, lane, j, a1[j]); });
                gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
                gws_w2r();
                {
                  f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                  gws_pass(rec, lane, Dt);
                  gws_r2w();
                  gws_stage(lane, Dt, stgL, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; });
                }
                // ---- first layer: dW1 | db1 = sum g1 (x) [x | 1] in 16-column windows of the input
                static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g1[j]); });
                static_for<0, (2 * XL + 15) / 16>([&](auto t_) {
                  constexpr int t = decltype(t_)::value;
                  static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < XL) gws_putB(rec, lane, j, xs[8 * t + j]); });
                  gws_w2r();
                  f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
                  gws_pass(rec, lane, Dt);
                  gws_r2w();
                  gws_stage(lane, Dt, stgL, [&](int c, int il) { const int i = 16 * t + il; return c < H ? (i < LIN ? c * LIN + i : (i == LIN ? ob1 + c : -1)) : -1; });
                });
                // ---- input adjoints, four at a time, straight to their consumers
                bwd_inputs<(LIN + 3) / 4, H>(nb + NL::total, g1, [&](auto ip_, f2 v) {
                  constexpr int ip = decltype(ip_)::value;
                  if constexpr (ip == 0) { xsv += v.x; xsth += v.y; }                                   // d/dv, d/dtheta
                  else if constexpr (ip == 1) xsdp += v.x;                                              // d/ddp (dq carries none)
                  else if constexpr (ip < SOFF) macc[ip - 2] += v;                                       // d/dm_k
                  else if constexpr (ip < SOFF + H / 2) {
                    if constexpr (MULTI) { if (is_bus) reinterpret_cast<f2*>(gS_l + n * H)[ip - SOFF] = v; }   // what every line ending here receives
                    else gSacc[ip - SOFF] += v;
                  }
                });
                // the block is complete in the stage: request the slab's running sums now, add after the barrier
                gws_r2w();
                rb_blk = slab + A.g_off[NPHI + l] + koff * A.g_sz[NPHI + l]; rb_n = GSZ;
#pragma unroll
                for (int i = 0; i < NRB; ++i) { const int t = lane + 64 * i; Rb[i] = (t < GSZ) ? rb_blk[t] : 0.f; }
              }
            }
          });
          if constexpr (!MULTI) { if (is_bus) gb_write_row<H>(gS_l + n * H, gSacc); }
          f2 uh[H / 2];
          f2 (&mk)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(xs[2]);
          phi_head<D, H>(PT + A.t_off[pf] + koff * A.t_sz[pf], mk, uh);
          if (is_bus) gb_write_row<H>(u_l + n * H, uh);
        }
        __syncthreads();
        if (rb_blk) {
#pragma unroll
          for (int i = 0; i < NRB; ++i) { const int t = lane + 64 * i; if (t < rb_n) rb_blk[t] = Rb[i] + stgL[t]; }
          rb_blk = nullptr;
          gws_r2w();
        }
        if ((e_wave[0] || e_wave[1]) && !skip_round) {
          constexpr int ob1 = PIN * H, oW2 = ob1 + H, ob2 = oW2 + H * H;
          f32x4 D1 = {0.f, 0.f, 0.f, 0.f}, D2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ep = 0; ep < 2; ++ep) {
            if (!e_wave[ep]) continue;
            f2 uh[H / 2], gh[H / 2], a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
            gb_read_row<H>(u_l + et[ep] * H, uh);
            gb_read_row<H>(gS_l + et[ep] * H, gh);
            if (!is_e[ep]) {
#pragma unroll
              for (int j = 0; j < H / 2; ++j) gh[j] = f2{0.f, 0.f};
            }
            f2 sl[2][H / 2];
            phi_tail<PIN, H, D>(PT + A.t_off[pf] + koff * A.t_sz[pf], uh, xt[ep], a1, a2, NoBG{}, NoLink{}, GWB_SLOPES(sl));
#pragma unroll
            for (int u = 0; u < H / 2; ++u) g2[u] = gh[u] * GWB_SLOPE_OF(sl, 1, a2, u);
            bwd_rows<H, H>(PN + A.n_off[pf] + koff * A.n_sz[pf], g2, g1);
#pragma unroll
            for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * GWB_SLOPE_OF(sl, 0, a1, u);
            if (is_e[ep]) gb_write_row<H>(g1_l + (li + ep * WL) * H, g1);
            // phi' line columns of dW1, db1 and dW2, db2
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g1[j]); });
            gws_putB(rec, lane, 0, xt[ep][0]); gws_putB(rec, lane, 1, xt[ep][1]); gws_putB(rec, lane, 2, f2{xt[ep][2].x, 1.f});
            gws_w2r();
            gws_pass(rec, lane, D1);
            gws_r2w();
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, g2[j]); gws_putB(rec, lane, j, a1[j]); });
            gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
            gws_w2r();
            gws_pass(rec, lane, D2);
            gws_r2w();
          }
          gws_stage(lane, D1, stgL, [&](int c, int il) { return c < H ? (il < PIN - D ? c * PIN + D + il : (il == PIN - D ? ob1 + c : -1)) : -1; });
          gws_stage(lane, D2, stgL, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; });
          gws_r2w();
          re_blk = slab + A.g_off[pf] + koff * A.g_sz[pf];
#pragma unroll
          for (int i = 0; i < NRE; ++i) { const int t = lane + 64 * i; Re[i] = (t < NE_T) ? re_blk[map_e(t)] : 0.f; }
        }
        __syncthreads();
        if (re_blk) {
#pragma unroll
          for (int i = 0; i < NRE; ++i) { const int t = lane + 64 * i; if (t < NE_T) { const int ix = map_e(t); re_blk[ix] = Re[i] + stgL[ix]; } }
          re_blk = nullptr;
          gws_r2w();
        }
        if (bus_wave && !skip_round) {
          if (rp_blk) {                                                    // the previous round's latent columns
#pragma unroll
            for (int i = 0; i < NRP; ++i) { const int t = lane + 64 * i; if (t < H * D) rp_blk[map_p(t)] = Rp[i] + stgP[t]; }
            rp_blk = nullptr;
            gws_r2w();
          }
          f2 G1[H / 2];
#pragma unroll
          for (int j = 0; j < H / 2; ++j) G1[j] = f2{0.f, 0.f};
          for (int p = p0; p < p1; ++p) {
            const f2* gr = reinterpret_cast<const f2*>(g1_l + p * H);
#pragma unroll
            for (int j = 0; j < H / 2; ++j) G1[j] += gr[j];
          }
          // d/dm += W1[:, :d]^T G1, and the latent columns of phi's dW1: sum over buses G1 (x) m
          bwd_inputs<(D + 3) / 4, H>(PN + A.n_off[pf] + koff * A.n_sz[pf] + NLay2<PIN, H>::total, G1, [&](auto ip_, f2 v) {
            constexpr int ip = decltype(ip_)::value;
            if constexpr (ip < D / 2) macc[ip] += v;
          });
          static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA(rec, lane, j, G1[j]); });
          static_for<0, (D + 15) / 16>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < D / 2) gws_putB(rec, lane, j, xs[2 + 8 * t + j]); });
            gws_w2r();
            f32x4 Dt = {0.f, 0.f, 0.f, 0.f};
            gws_pass(rec, lane, Dt);
            gws_r2w();
            gws_stage(lane, Dt, stgP, [&](int c, int il) { const int i = 16 * t + il; return (c < H && i < D) ? c * D + i : -1; });
          });
          gws_r2w();
          rp_blk = slab + A.g_off[pf] + koff * A.g_sz[pf];
#pragma unroll
          for (int i = 0; i < NRP; ++i) { const int t = lane + 64 * i; Rp[i] = (t < H * D) ? rp_blk[map_p(t)] : 0.f; }
        }
      });
      // ---- the adjoints entering step k (identity paths main.py:182,186,188 + what the L' inputs collected) --------
      if (bus_wave) { vbar += xsv; thbar += xsth; dpbar_in = xsdp; }
    }
    if (rp_blk) {
#pragma unroll
      for (int i = 0; i < NRP; ++i) { const int t = lane + 64 * i; if (t < H * D) rp_blk[map_p(t)] = Rp[i] + stgP[t]; }
      rp_blk = nullptr;
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
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gwb_resident_t(int threads, size_t lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&gns_gw_backward_kernel<D, H, MULTI, MAXT, MINW>), threads, lds) != hipSuccess || nb < 1) {
    (void)hipGetLastError();
    nb = 1;
  }
  return nb;
}
struct GwbShape { int recf, stgf; };
GwbShape shape_of(int d, int h, int multi) {
#define GNS_CASE(DD, HH) if (d == DD && h == HH) return multi ? GwbShape{GwBwdDims<DD, HH, true>::RECF, GwBwdDims<DD, HH, true>::STGF} : GwbShape{GwBwdDims<DD, HH, false>::RECF, GwBwdDims<DD, HH, false>::STGF};
  GNS_FOR_EACH_DIMS_GWB(GNS_CASE)
#undef GNS_CASE
  return GwbShape{-1, -1};
}
int g_gwb_cus = 0;
}  // namespace

int gns_gw_backward_init_device(void) {
  int rc = GNS_OK;
#define GNS_CASE(DD, HH)                                                                                          \
  if (gwb_attr_t<DD, HH, true, 256, 2>() != GNS_OK || gwb_attr_t<DD, HH, false, 256, 2>() != GNS_OK ||              \
      gwb_attr_t<DD, HH, true, 512, 1>() != GNS_OK || gwb_attr_t<DD, HH, false, 512, 1>() != GNS_OK ||              \
      gwb_attr_t<DD, HH, true, 1024, 1>() != GNS_OK || gwb_attr_t<DD, HH, false, 1024, 1>() != GNS_OK) rc = GNS_ELAUNCH;
  GNS_FOR_EACH_DIMS_GWB(GNS_CASE)
#undef GNS_CASE
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  g_gwb_cus = cus;
  (void)hipGetLastError();
  return rc;
}

int gns_gw_backward_wpg(int N) { return (N + 63) / 64; }

int gns_gw_backward_supported(int N, int E, int d, int h, int multi, int P) {
  const GwbShape S = shape_of(d, h, multi);
  if (S.recf < 0 || P < 1) return 0;
  const int WPG = gns_gw_backward_wpg(N);
  if (WPG * P > 16 || E > 2 * 64 * WPG) return 0;                       // at most two lines per lane
  const GwBwdLds L = gw_bwd_lds_layout(N, E, h, WPG, S.recf, S.stgf);
  return (size_t)L.total * 4 * P <= (size_t)GWB_LDS_MAX_BYTES ? 1 : 0;
}

int gns_gw_backward_blocks(int N, int E, int d, int h, int multi, int P, long long Bt) {
  const GwbShape S = shape_of(d, h, multi);
  if (S.recf < 0) return 0;
  const int WPG = gns_gw_backward_wpg(N);
  const GwBwdLds L = gw_bwd_lds_layout(N, E, h, WPG, S.recf, S.stgf);
  const size_t lds = (size_t)L.total * 4 * P;
  const int threads = P * WPG * 64;
  int per_cu = 1;
#define GNS_CASE(DD, HH)                                                                                          \
  if (d == DD && h == HH) {                                                                                       \
    if (threads <= 256) per_cu = multi ? gwb_resident_t<DD, HH, true, 256, 2>(threads, lds) : gwb_resident_t<DD, HH, false, 256, 2>(threads, lds);        \
    else if (threads <= 512) per_cu = multi ? gwb_resident_t<DD, HH, true, 512, 1>(threads, lds) : gwb_resident_t<DD, HH, false, 512, 1>(threads, lds);   \
    else per_cu = multi ? gwb_resident_t<DD, HH, true, 1024, 1>(threads, lds) : gwb_resident_t<DD, HH, false, 1024, 1>(threads, lds);                    \
  }
  GNS_FOR_EACH_DIMS_GWB(GNS_CASE)
#undef GNS_CASE
  const long long npacks = (Bt + P - 1) / P;
  const long long cap = (long long)(g_gwb_cus > 0 ? g_gwb_cus : 256) * per_cu;
  return (int)(npacks < cap ? npacks : cap);
}

int gns_gw_launch_backward(int d, int h, int multi, const GnsGwBwdArgs& A, int blocks, hipStream_t st) {
  const int threads = A.P * A.WPG * 64;
  const GwbShape S = shape_of(d, h, multi);
  if (S.recf < 0) return GNS_EUNSUPPORTED;
  const GwBwdLds L = gw_bwd_lds_layout(A.N, A.E, h, A.WPG, S.recf, S.stgf);
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
  GNS_FOR_EACH_DIMS_GWB(GNS_CASE)
#undef GNS_CASE
  return GNS_EUNSUPPORTED;
}
