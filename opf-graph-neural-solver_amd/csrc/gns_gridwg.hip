// Grid-per-workgroup forward of the GNS K-step loop (GNS/main.py:140-202) for gfx950 - see gns_gridwg.h.
//
// Per step k, two workgroup barriers:
//   BUS-1  bus lanes: sum the hidden vectors of the lines ending at the bus (LDS, line order = the reference's
//          scatter_add order), run L'_theta, L'_v, L'_m on registers, update theta / v / m (main.py:165-188), publish
//          (v, theta) to the LDS plane, and already compute the bus share of phi' for step k+1 (phi_head)
//   ---- barrier
//   EDGE   edge lanes: line physics of step k from the plane (main.py:34-104, 8 sin/cos shared by all terms), then the
//          line share of phi' for step k+1 (phi_tail) -> hidden vector to LDS
//   ---- barrier
//   BUS-2  bus lanes: gather the per-line terms of their own lines, lambda (main.py:45-57), delta_p / delta_q, loss;
//          falls straight through into BUS-1 of step k+1.
// No atomics; every sum has a fixed order, so results are bitwise reproducible.
#include "gns_device.h"
#include "gns_gridwg.h"

namespace {

struct __attribute__((packed, aligned(4))) GwU4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) GwU2 { float x, y; };
__device__ __forceinline__ f4 gw_ld4(const float* p) { const GwU4 u = *reinterpret_cast<const GwU4*>(p); return f4{u.x, u.y, u.z, u.w}; }
__device__ __forceinline__ f2 gw_ld2(const float* p) { const GwU2 u = *reinterpret_cast<const GwU2*>(p); return f2{u.x, u.y}; }

__device__ __forceinline__ float gw_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// main.py:38: 1 / sqrt(r^2 + x^2), each op rounded like torch does
__device__ __forceinline__ float gw_yof(float r, float x) {
  return __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(x, x))));
}

// sin / cos on |x| <= pi/4 without range reduction (Cephes single-precision kernels, < 1 ulp); the caller votes
// wave-wide that every angle is in range and otherwise takes the library path.
__device__ __forceinline__ void gw_sincos_small(float x, float& s, float& c) {
  const float z = x * x;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  s = __builtin_fmaf(ps * z, x, x);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}
__device__ __forceinline__ float gw_sin_small(float x) {
  const float z = x * x;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  return __builtin_fmaf(ps * z, x, x);
}

template <int NF>
__device__ __forceinline__ void lds_read_row(const float* row, f2 (&dst)[NF / 2]) {
#pragma unroll
  for (int i = 0; i < NF / 2; ++i) dst[i] = reinterpret_cast<const f2*>(row)[i];
}
template <int NF>
__device__ __forceinline__ void lds_write_row(float* row, const f2 (&src)[NF / 2]) {
#pragma unroll
  for (int i = 0; i < NF / 2; ++i) reinterpret_cast<f2*>(row)[i] = src[i];
}

}  // namespace

template <int D, int H, bool MULTI, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) gns_gw_forward_kernel(GnsGwFwdArgs A) {
  using C = GnsDims<D, H, MULTI>;
  constexpr int NPHI = C::NPHI, UW = NPHI * H, MQ = C::MQ;
  constexpr int SVQ = 1 + MQ, HQ = C::HQ, SSQ = NPHI * HQ;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = A.N, E = A.E, K = A.K, Gn = A.Gn, WPG = A.WPG, P = A.P;
  const int gslot = wv / WPG, wig = wv - gslot * WPG;          // grid slot of this wave, wave index inside the grid
  const int li = wig * 64 + lane;
  const bool bus_wave = wig * 64 < N, edge_wave = wig * 64 < E; // wave-uniform
  const bool is_bus = li < N, is_edge = li < E;
  cip topo_s = (cip)A.topo;                                      // header words: wave-uniform -> s_load
  const int* topo = A.topo;
  cfp PT = (cfp)A.pt;

  // ---- per-lane topology (the same for every grid of the batch): loaded once ------------------------------------
  int n = 0, p0 = 0, p1 = 0, q0 = 0, q1 = 0, g0 = 0, g1 = 0, isgen = 0;
  if (is_bus) {
    n = topo[topo_s[TH_LANE_BUS] + li];
    p0 = topo[topo_s[TH_IN_PTR] + n];   p1 = topo[topo_s[TH_IN_PTR] + n + 1];
    q0 = topo[topo_s[TH_OUT_PTR] + n];  q1 = topo[topo_s[TH_OUT_PTR] + n + 1];
    g0 = topo[topo_s[TH_GEN_PTR] + n];  g1 = topo[topo_s[TH_GEN_PTR] + n + 1];
    isgen = topo[topo_s[TH_IS_GEN] + n];
  }
  const int* q2p = topo + topo_s[TH_Q2P];
  const int* gen_idx = topo + topo_s[TH_GEN_IDX];
  // edge lane p = position in the destination-sorted line list (the lines ending at a bus are consecutive rows of h)
  int e_id = 0, es = 0, et = 0, ia = 0, ib = 0, ic = 0, id = 0;
  if (is_edge) {
    e_id = topo[topo_s[TH_IN_EID] + li];
    es = topo[topo_s[TH_IN_SRC] + li];  et = topo[topo_s[TH_IN_DST] + li];
    ia = topo[topo_s[TH_IN_A] + li];    ib = topo[topo_s[TH_IN_B] + li];      // line NUMBER s = src[e] (main.py:41)
    const int q = topo[topo_s[TH_P2Q] + li];
    ic = topo[topo_s[TH_OUT_C] + q];    id = topo[topo_s[TH_OUT_D] + q];      // line NUMBER t = dst[e] (main.py:70-72)
  }

  extern __shared__ __attribute__((aligned(16))) float gw_lds_mem[];
  const GwLds LY = gw_lds_layout(N, E, UW, WPG);
  float* Lb = gw_lds_mem + (size_t)gslot * LY.total;
  float* u_l = Lb + LY.u;
  float* h_l = Lb + LY.h;
  f2* plane = reinterpret_cast<f2*>(Lb + LY.plane);
  f4* phys = reinterpret_cast<f4*>(Lb + LY.phys);
  float* red = Lb + LY.red;                 // [0..4W): [par][kind][w]   [4W..8W): gsum [4][w]   [8W..10W): epilogue [2][w]
  const float invN = 1.0f / (float)N;

  const long long npacks = (A.Bt + P - 1) / P;
  for (long long pack = blockIdx.x; pack < npacks; pack += gridDim.x) {
    long long b = pack * P + gslot;
    const bool live = b < A.Bt;
    if (!live) b = A.Bt - 1;               // a dead slot replays the last grid; nothing of it is stored

    // ================= prologue (main.py:141-152) =================================================================
    float Pd = 0.f, Qd = 0.f, Gs = 0.f, Bs = 0.f, pmin = 0.f, pset = 0.f, pmax = 0.f;
    float sv = 1.f, sth = 0.f, sdp = 0.f, sdq = 0.f;
    f2 m[D / 2];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) m[i] = f2{0.f, 0.f};
    if (bus_wave) {
      float vg = 0.f, pg = 0.f, qg = 0.f;
      if (is_bus) {
        const f4 bq = gw_ld4(A.buses + (b * N + n) * 6 + 2);      // Pd, Qd, Gs, Bs
        Pd = bq.x; Qd = bq.y; Gs = bq.z; Bs = bq.w;
        for (int q = g0; q < g1; ++q) {                           // generators on this bus, in generator order (main.py:146-151)
          const float* r = A.gens + (b * Gn + gen_idx[q]) * 7;    // (bus_i,Pmax,Pmin,Pg_set,vg,qg,Pg) utils.py:9
          const f4 ra = gw_ld4(r + 1);
          const f2 rb = gw_ld2(r + 5);
          pmax += ra.x; pmin += ra.y; pset += ra.z; vg += ra.w; qg += rb.x; pg += rb.y;
        }
      }
      sv = (vg == 0.f) ? 1.f : vg;                                // main.py:146-147
      sdp = __fsub_rn(__fsub_rn(pg, Pd), __fmul_rn(Gs, __fmul_rn(sv, sv)));     // main.py:150
      sdq = __fadd_rn(__fsub_rn(qg, Qd), __fmul_rn(Bs, __fmul_rn(sv, sv)));     // main.py:152
      // per-grid sums of main.py:45,47-51
      const float r0 = gw_wave_sum(Pd), r1 = gw_wave_sum(pset), r2 = gw_wave_sum(pmin), r3 = gw_wave_sum(pmax);
      if (lane == 0) { red[4 * WPG + 0 * WPG + wig] = r0; red[4 * WPG + 1 * WPG + wig] = r1; red[4 * WPG + 2 * WPG + wig] = r2; red[4 * WPG + 3 * WPG + wig] = r3; }
    }
    // line parameters: own line e, and - the reference's bus-id-as-line-index quirk - line NUMBER s and line NUMBER t
    f2 xt[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
    float ys = 0.f, taus = 1.f, shs = 0.f, bs = 0.f, yt = 0.f, taut = 1.f, sht = 0.f, bt = 0.f;
    if (edge_wave) {
      const float* lb = A.lines + b * (long long)E * 7;
      const f4 ea = gw_ld4(lb + e_id * 7 + 2);                     // r, x, b, tau
      const float she = lb[e_id * 7 + 6];
      const f4 sa = gw_ld4(lb + es * 7 + 2);
      shs = lb[es * 7 + 6];
      const f4 ta = gw_ld4(lb + et * 7 + 2);
      sht = lb[et * 7 + 6];
      xt[0] = f2{ea.x, ea.y}; xt[1] = f2{ea.z, ea.w}; xt[2] = f2{she, 0.f};
      ys = gw_yof(sa.x, sa.y); taus = sa.w; bs = sa.z;
      yt = gw_yof(ta.x, ta.y); taut = ta.w; bt = ta.z;
      // phi' of step 0: m = 0, so the bus share of the first layer is zero (main.py:141,155)
      f2 uz[H / 2];
#pragma unroll
      for (int j = 0; j < H / 2; ++j) uz[j] = f2{0.f, 0.f};
      static_for<0, NPHI>([&](auto f_) {
        constexpr int f = decltype(f_)::value;
        f2 a1[H / 2], a2[H / 2];
        phi_tail<C::PHI_IN, H, D>(PT + A.t_off[f], uz, xt, a1, a2);
        if (is_edge) lds_write_row<H>(h_l + li * UW + f * H, a2);
      });
    }
    __syncthreads();
    float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gs3 = 0.f;               // sumPd, sumPset, sumPmin, sumPmax
    for (int w = 0; w * 64 < N; ++w) { gs0 += red[4 * WPG + w]; gs1 += red[5 * WPG + w]; gs2 += red[6 * WPG + w]; gs3 += red[7 * WPG + w]; }
    float tot_lane = 0.f, last_lane = 0.f;

    for (int k = 0; k < K; ++k) {
      const long long koff = (long long)k;
      const int par = k & 1;
      // ================= BUS-1: message sums, L', state update (main.py:161-188) ===================================
      if (bus_wave) {
        if (A.save && is_bus && live) {
          f4* ss = reinterpret_cast<f4*>(A.sv_state) + ((koff * A.Bt + b) * SVQ) * N + li;
          __builtin_nontemporal_store(f4{sv, sth, sdp, sdq}, ss);
          static_for<0, MQ>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            f4 t = {m[2 * q].x, m[2 * q].y, 0.f, 0.f};
            if constexpr (2 * q + 1 < D / 2) { t.z = m[2 * q + 1].x; t.w = m[2 * q + 1].y; }
            __builtin_nontemporal_store(t, ss + (long long)(1 + q) * N);
          });
        }
        f2 S[UW / 2];
#pragma unroll
        for (int i = 0; i < UW / 2; ++i) S[i] = f2{0.f, 0.f};
        for (int p = p0; p < p1; ++p) {                             // lines ending here, in line order
          const f2* hr = reinterpret_cast<const f2*>(h_l + p * UW);
#pragma unroll
          for (int i = 0; i < UW / 2; ++i) S[i] += hr[i];
        }
        if (A.save && is_bus && live) {
          f4* ss = reinterpret_cast<f4*>(A.sv_S) + ((koff * A.Bt + b) * SSQ) * N + li;
          static_for<0, SSQ>([&](auto q_) {
            constexpr int q = decltype(q_)::value, f = q / HQ, r = q % HQ, o = f * (H / 2) + 2 * r;   // family f, float4 row r of its H floats
            f4 t = {S[o].x, S[o].y, 0.f, 0.f};
            if constexpr (2 * r + 1 < H / 2) { t.z = S[o + 1].x; t.w = S[o + 1].y; }
            __builtin_nontemporal_store(t, ss + (long long)q * N);
          });
        }
        const float deg = (float)(p1 - p0);
        float th_new = sth, v_new = sv;
        f2 m_new[D / 2];
        // the L' input [v theta | dp dq | m | sum h | deg] is assembled ONCE per step; only the hidden-sum slots change per family
        f2 x[(C::LF_IN + 1) / 2];
        x[0] = f2{sv, sth}; x[1] = f2{sdp, sdq};
#pragma unroll
        for (int i = 0; i < D / 2; ++i) x[2 + i] = m[i];
        x[2 + D / 2 + H / 2] = f2{deg, 0.f};
        static_for<0, 3>([&](auto l_) {
          constexpr int l = decltype(l_)::value;                    // L_theta, L_v, L_m (main.py:173-180)
          constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;   // phi_theta, phi_v, phi_m
          if constexpr (MULTI || l == 0) {
#pragma unroll
            for (int q = 0; q < H / 2; ++q) x[2 + D / 2 + q] = S[fphi * (H / 2) + q];
          }
          f2 a1[H / 2], a2[H / 2];
          if constexpr (l < 2) {
            f2 y[1];
            mlp_fwd<C::LF_IN, H, 2>(PT + A.t_off[NPHI + l] + koff * A.t_sz[NPHI + l], x, a1, a2, y);
            if constexpr (l == 0) th_new = sth + y[0].x;                        // main.py:182
            else v_new = isgen ? sv : sv + y[0].x;                             // main.py:184-186
          } else if (k + 1 < K) {
            f2 y[D / 2];
            mlp_fwd<C::LF_IN, H, D>(PT + A.t_off[NPHI + 2] + koff * A.t_sz[NPHI + 2], x, a1, a2, y);
#pragma unroll
            for (int i = 0; i < D / 2; ++i) m_new[i] = x[2 + i] + y[i];          // main.py:188
          } else {                                                  // m_K feeds nothing: the last step's L_m is dead work
#pragma unroll
            for (int i = 0; i < D / 2; ++i) m_new[i] = x[2 + i];
          }
        });
        sv = v_new; sth = th_new;
#pragma unroll
        for (int i = 0; i < D / 2; ++i) m[i] = m_new[i];
        if (is_bus) plane[n] = f2{sv, sth};
        const float vs_part = gw_wave_sum(is_bus ? (sv * sv) * Gs : 0.f);         // sum v^2 Gs (main.py:45)
        if (lane == 0) red[(par * 2 + 0) * WPG + wig] = vs_part;
        if (k + 1 < K) {                                              // bus share of phi' for the next step
          static_for<0, NPHI>([&](auto f_) {
            constexpr int f = decltype(f_)::value;
            if (MULTI && f == 2 && k + 2 >= K) return;              // phi_m of the last step only feeds the dead L_m
            f2 uh[H / 2];
            phi_head<D, H>(PT + A.t_off[f] + (koff + 1) * A.t_sz[f], m, uh);
            if (is_bus) lds_write_row<H>(u_l + n * UW + f * H, uh);
          });
        }
      }
      __syncthreads();
      // ================= EDGE: line physics of step k (main.py:34-104), phi' of step k+1 ============================
      if (edge_wave) {
        const f2 ss = plane[es], st = plane[et];
        const float tha = plane[ia].y, thb = plane[ib].y, thc = plane[ic].y, thd = plane[id].y;
        const float vs = ss.x, ths = ss.y, vt = st.x, tht = st.y;
        const float dl = tha - thb;                                   // delta_ij[src]  (bus id used as line index)
        const float dl2 = thd - thc;                                  // delta_ji[dst]
        const float angA = ths - tht - dl - shs, angB = tht - ths - dl + shs, angC = tht - ths - dl2 - sht;
        float sA, cA, sB, sD, cD, sC, cC, sD2;
        const float amax = fmaxf(fmaxf(fmaxf(fabsf(angA), fabsf(angB)), fmaxf(fabsf(angC), fabsf(dl))), fabsf(dl2));
        if (__builtin_amdgcn_ballot_w64(!(amax <= 0.785f)) == 0) {
          gw_sincos_small(angA, sA, cA); sB = gw_sin_small(angB); gw_sincos_small(dl, sD, cD);
          gw_sincos_small(angC, sC, cC); sD2 = gw_sin_small(dl2);
        } else {
          sincosf(angA, &sA, &cA); sB = sinf(angB); sincosf(dl, &sD, &cD);
          sincosf(angC, &sC, &cC); sD2 = sinf(dl2);
        }
        const float base = vs * vt * ys / taus;
        const float msg = fabsf(base * (sA + sB) + (vs / (taus * taus)) * ys * sD + (vt * vt) * ys * sD);   // main.py:41
        const float vst = vs / taus, vst2 = vst * vst;
        const float p_from = base * sA + vst2 * ys * sD;                        // main.py:91
        const float q_from = -base * cA + vst2 * (ys * cD - bs / 2.f);          // main.py:68-69 == :98
        const float base2 = vt * vs * yt / taut;
        const float p_to = base2 * sC + (vt * vt) * yt * sD2;                   // main.py:92
        const float q_to = -base2 * cC + (vt * vt) * (yt * sD2 - bt / 2.f);     // main.py:70-72 == :99
        if (is_edge) phys[li] = f4{p_from, q_from, p_to, q_to};
        const float j_part = gw_wave_sum(is_edge ? msg : 0.f);                   // main.py:42-43
        if (lane == 0) red[(par * 2 + 1) * WPG + wig] = j_part;
        if (k + 1 < K) {
          static_for<0, NPHI>([&](auto f_) {
            constexpr int f = decltype(f_)::value;
            if (MULTI && f == 2 && k + 2 >= K) return;              // (see BUS-1)
            f2 uh[H / 2], a1[H / 2], a2[H / 2];
            lds_read_row<H>(u_l + et * UW + f * H, uh);
            phi_tail<C::PHI_IN, H, D>(PT + A.t_off[f] + (koff + 1) * A.t_sz[f], uh, xt, a1, a2);
            if (is_edge) lds_write_row<H>(h_l + li * UW + f * H, a2);
          });
        }
      }
      __syncthreads();
      // ================= BUS-2: delta_p / delta_q of step k, lambda, loss (main.py:45-57, 81-103, 198) ==============
      // Every step has its own weights, and a wave streams each of them exactly once: the scalar cache is cold for all of them and
      // a stream keeps two lines in flight.  One wave per workgroup puts the next step's L blocks (and the phi blocks of the step
      // after) in flight here, all lines at once, while this phase - which streams nothing - does its sums.
      WarmTok wt[6] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
      const bool warmer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0 && k + 1 < K;     // (wave-uniform: the loads go through the scalar unit)
      if (warmer) {
#pragma unroll
        for (int f = 0; f < 3; ++f) wt[f] = scalar_cache_warm_issue(PT + A.t_off[NPHI + f] + (koff + 1) * A.t_sz[NPHI + f], A.t_sz[NPHI + f]);
        if (k + 2 < K) {
#pragma unroll
          for (int f = 0; f < NPHI; ++f) wt[3 + f] = scalar_cache_warm_issue(PT + A.t_off[f] + (koff + 2) * A.t_sz[f], A.t_sz[f]);
        }
      }
      if (bus_wave) {
        float sum_pf = 0.f, sum_qf = 0.f, sum_pt = 0.f, sum_qt = 0.f;
        for (int p = p0; p < p1; ++p) { const f4 t = phys[p]; sum_pf += t.x; sum_qf += t.y; }            // lines with dst == n
        for (int q = q0; q < q1; ++q) { const f4 t = phys[q2p[q]]; sum_pt += t.z; sum_qt += t.w; }       // lines with src == n
        float jsum = 0.f, vsum = 0.f;
        for (int w = 0; w * 64 < N; ++w) vsum += red[(par * 2 + 0) * WPG + w];
        for (int w = 0; w * 64 < E; ++w) jsum += red[(par * 2 + 1) * WPG + w];
        const float p_global = (gs0 + vsum) + jsum;
        float lam;
        const bool low1 = p_global < gs1;
        if (low1) lam = (p_global - gs2) / (2.f * (gs1 - gs2));
        else lam = (p_global - 2.f * gs1 + gs3) / (2.f * (gs3 - gs1));
        const bool low2 = lam < 0.5f;
        if (A.save && live && wig == 0 && lane == 0)
          reinterpret_cast<f2*>(A.sv_lam)[koff * A.Bt + b] = f2{lam, (low1 ? 1.f : 0.f) + (low2 ? 2.f : 0.f)};
        const float v2 = sv * sv;
        const float dp_pre = ((0.f - Pd) - Gs * v2) + sum_pf + sum_pt;           // main.py:82,96 without the generator term
        const float qg_new = ((Qd - Bs * v2) - sum_qf) - sum_qt;                 // main.py:64,76
        sdq = ((qg_new - Qd) + Bs * v2) + sum_qf + sum_qt;                       // main.py:83,103 (cancels to rounding noise)
        const float pg = low2 ? pmin + 2.f * (pset - pmin) * lam
                              : 2.f * pset - pmax + 2.f * (pmax - pset) * lam;   // main.py:53-57, summed over the bus's generators
        sdp = pg + dp_pre;                                                       // main.py:81-82,96
        const float sq = is_bus ? sdp * sdp + sdq * sdq : 0.f;
        tot_lane += A.gw[k] * sq;                                                // main.py:198 (mean over buses applied below)
        last_lane = sq;                                                          // main.py:199
      }
      if (warmer) {
#pragma unroll
        for (int f = 0; f < 6; ++f) scalar_cache_warm_wait(wt[f]);
      }
    }

    // ================= epilogue: outputs (main.py:199-202) ===========================================================
    if (bus_wave) {
      if (A.save && is_bus && live)      // (v, theta, dp)_K for the reverse pass of the last step
        __builtin_nontemporal_store(f4{sv, sth, sdp, sdq}, reinterpret_cast<f4*>(A.sv_state) + (((long long)K * A.Bt + b) * SVQ) * N + li);
      if (is_bus && live) {
        A.v_out[b * N + n] = (sv < 0.f) ? 0.f : sv;                              // main.py:201
        A.theta_out[b * N + n] = sth;
      }
      const float t = gw_wave_sum(tot_lane), l = gw_wave_sum(last_lane);
      if (lane == 0) { red[8 * WPG + wig] = t; red[9 * WPG + wig] = l; }
    }
    __syncthreads();
    if (wig == 0 && lane == 0 && live) {
      float t = 0.f, l = 0.f;
      for (int w = 0; w * 64 < N; ++w) { t += red[8 * WPG + w]; l += red[9 * WPG + w]; }
      A.total_out[b] = t * invN;
      A.last_out[b] = l * invN;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
namespace {
constexpr int GW_LDS_MAX_BYTES = 160 * 1024;

template <int D, int H, bool MULTI, int MAXT, int MINW>
int gw_launch_t(const GnsGwFwdArgs& A, int blocks, int threads, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((gns_gw_forward_kernel<D, H, MULTI, MAXT, MINW>), dim3(blocks), dim3(threads), lds, st, A);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gw_attr_t() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gns_gw_forward_kernel<D, H, MULTI, MAXT, MINW>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS_MAX_BYTES) == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
int g_gw_cus = 0;
}  // namespace

int gns_gw_supported(int N, int E, int d, int h, int multi, int P) {
  bool dims = false;
#define GNS_CASE(DD, HH) if (d == DD && h == HH) dims = true;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  if (!dims || P < 1) return 0;
  const int WPG = ((N > E ? N : E) + 63) / 64;
  if (WPG * P > 16) return 0;
  const GwLds L = gw_lds_layout(N, E, (multi ? 3 : 1) * h, WPG);
  return (size_t)L.total * 4 * P <= (size_t)GW_LDS_MAX_BYTES ? 1 : 0;
}

int gns_gw_init_device(void) {
  int rc = GNS_OK;
#define GNS_CASE(DD, HH)                                                                                       \
  if (gw_attr_t<DD, HH, true, 256, 3>() != GNS_OK || gw_attr_t<DD, HH, false, 256, 3>() != GNS_OK ||              \
      gw_attr_t<DD, HH, true, 768, 1>() != GNS_OK || gw_attr_t<DD, HH, false, 768, 1>() != GNS_OK ||              \
      gw_attr_t<DD, HH, true, 1024, 1>() != GNS_OK || gw_attr_t<DD, HH, false, 1024, 1>() != GNS_OK) rc = GNS_ELAUNCH;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  g_gw_cus = cus;
  (void)hipGetLastError();
  return rc;
}

namespace {
// resident workgroups per CU for one (instantiation, threads, LDS) combination: asked once from the runtime, then cached
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gw_resident_t(int threads, size_t lds) {
  static int cache_threads = 0, cache_val = 0;
  static size_t cache_lds = 0;
  if (cache_val > 0 && cache_threads == threads && cache_lds == lds) return cache_val;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&gns_gw_forward_kernel<D, H, MULTI, MAXT, MINW>), threads, lds) != hipSuccess || nb < 1) {
    (void)hipGetLastError();
    nb = 1;
  }
  cache_threads = threads; cache_lds = lds; cache_val = nb;
  return nb;
}
template <int D, int H, bool MULTI, int MAXT, int MINW>
int gw_go(const GnsGwFwdArgs& A, int threads, size_t lds, hipStream_t st) {
  // A persistent grid: exactly the workgroups that are resident at once (a second, partial round of workgroups was
  // measured to leave the waves alive for only 54-75 % of the kernel), each looping over its share of the packs.
  int per_cu = gw_resident_t<D, H, MULTI, MAXT, MINW>(threads, lds);
  const int waves = threads / 64;
  (void)waves;
  const long long npacks = (A.Bt + A.P - 1) / A.P;
  const long long cap = (long long)(g_gw_cus > 0 ? g_gw_cus : 256) * per_cu;
  const int blocks = (int)(npacks < cap ? npacks : cap);
  return gw_launch_t<D, H, MULTI, MAXT, MINW>(A, blocks, threads, lds, st);
}
}  // namespace

int gns_gw_launch_forward(int d, int h, int multi, const GnsGwFwdArgs& A, hipStream_t st) {
  const int threads = A.P * A.WPG * 64;
  const GwLds L = gw_lds_layout(A.N, A.E, (multi ? 3 : 1) * h, A.WPG);
  const size_t lds = (size_t)L.total * 4 * A.P;
  if (threads > 1024 || lds > (size_t)GW_LDS_MAX_BYTES) return GNS_EUNSUPPORTED;
#define GNS_CASE(DD, HH)                                                                                       \
  if (d == DD && h == HH) {                                                                                    \
    if (threads <= 256) return multi ? gw_go<DD, HH, true, 256, 3>(A, threads, lds, st) : gw_go<DD, HH, false, 256, 3>(A, threads, lds, st);   \
    if (threads <= 768) return multi ? gw_go<DD, HH, true, 768, 1>(A, threads, lds, st) : gw_go<DD, HH, false, 768, 1>(A, threads, lds, st);   \
    return multi ? gw_go<DD, HH, true, 1024, 1>(A, threads, lds, st) : gw_go<DD, HH, false, 1024, 1>(A, threads, lds, st);                      \
  }
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return GNS_EUNSUPPORTED;
}
