// Split reverse pass of the GNS K-step loop (bwd_variant 4): what total_loss.backward() does (GNS/main.py:288) through
// GNS.forward (main.py:140-202), as a SEQUENCE of kernels per reverse step instead of one persistent kernel.
//
//   gns_bwds_phys_kernel   one 16-wave workgroup per 64-grid group (lane = grid, 128-VGPR budget, 4 waves per SIMD):
//                          Pb-0 closes step k+1 (identity paths main.py:182,186 + the input-adjoint parts of its sweeps), adds
//                          the loss term (main.py:198-199), reduces the adjoint of lambda (main.py:47-57); the line phase
//                          reverses global_active_compensation / local_power_imbalance (main.py:34-104) per line; the gather
//                          sums every bus's incidence lists in fixed order.  The phases that were latency-bound at the two
//                          waves per SIMD of the 256-register sweep now run at four.
//   gns_bwds_sweep_kernel  ONE-WAVE workgroups, one per (family, bus chunk, block of groups), no barrier at all: the update
//                          step (main.py:155-188) is recomputed and back-propagated for the buses of the chunk exactly as in
//                          the V2 sweep of gns_backward.hip (layer-wise data path, sub-record windows on the matrix pipe).
//                          The three families of a step write separate parts of the latent adjoint (gns_common.h), so they are
//                          independent of each other and any number of CUs can work one 64-grid group: no teams, no counters
//                          in HBM, no spin-waits.  Every (family, step) block of a slab is stored exactly once.
// delta_q carries no gradient (identically zero as a function of (v, theta): main.py:64-76 vs :83,98-103).
#include "gns_device.h"
#include "gns_kernels.h"
#include "gns_dw.h"

typedef int gns_i8v __attribute__((ext_vector_type(8)));   // one line record of TH_EREC
#ifndef GNS_BWDS_ONE_LAYOUT
#define GNS_BWDS_ONE_LAYOUT 0  // 1: recomputation AND data gradients stream the forward layouts (gns_device.h, bwd_from_fwd_layout): one copy of every
                               //    matrix in the scalar cache (11 KB instead of 20 KB for three families).  0 (default): data gradients from the
                               //    transposed N-stream copies, like the persistent kernels.  Measured equal within noise in every mode and
                               //    configuration (profiles/r03/ablation_split.txt E): the option stays for kernels that run more families at once
#endif
#ifndef GNS_BWDS_WPE
#define GNS_BWDS_WPE 2          // waves per SIMD the sweep kernels are compiled for.  Measured (case118 x 16384): 3 (168 registers, 12 bus
                                // chunks per group) is 8 % SLOWER than 2 - the sweeps are bound by the rows they stream, not by latency
#endif
#ifndef GNS_BWDS_KEEP_SLOPES
#define GNS_BWDS_KEEP_SLOPES 1    // the recomputation keeps LeakyReLU's slope of every hidden unit (two registers per pair and layer) and the
#endif                            // backward multiplies by it instead of deriving it again from the activation: two packed instructions
                                  // less per pair and layer on the dependent chain; 0 (diagnostic): derive again
#if GNS_BWDS_KEEP_SLOPES
#define GNS_SLOPES(sl) sl
#define GNS_SLOPE_OF(sl, layer, act, u) sl[layer][u]
#else
#define GNS_SLOPES(sl) nullptr
#define GNS_SLOPE_OF(sl, layer, act, u) dlrelu2(act[u])
#endif
#ifndef GNS_BWDS_FRESH_INPUTS
#define GNS_BWDS_FRESH_INPUTS 1   // 0 (diagnostic): the nets of a bus share their broadcast input pairs (see gns_bwds_sweep_kernel)
#endif
#define GNS_BWDS_PHYS_THREADS (GNS_BWDS_PHYS_WAVES * 64)

// ------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GNS_BWDS_PHYS_THREADS) gns_bwds_phys_kernel(GnsBwdsArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int W = GNS_BWDS_PHYS_WAVES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = A.N, E = A.E, K = A.K, k = A.k;
  cip topo = (cip)A.topo;
  constexpr int pidx = 4;                                              // gns_part_index(16): the partition tables for 16 waves
  static_assert(GNS_BWDS_PHYS_WAVES == 16, "pidx");
  const cip in_ptr = topo + topo[TH_IN_PTR], out_ptr = topo + topo[TH_OUT_PTR], q2p = topo + topo[TH_Q2P], is_gen = topo + topo[TH_IS_GEN],
            incd_ptr = topo + topo[TH_INCD_PTR], incd = topo + topo[TH_INCD],
            part = topo + topo[TH_PART] + pidx * (GNS_MAXP + 1), epart = topo + topo[TH_EPART] + pidx * (GNS_MAXP + 1);
  const int n0 = part[wave], n1 = part[wave + 1];
  const int e0 = epart[wave], e1 = epart[wave + 1];
  // LDS planes of the line phase: 2 = (v, theta) and the adjoint of delta_p of every bus, 1 = (v, theta) only (case300: 154 KB),
  // 0 = none (the line phase gathers rows from L2 / HBM)
  const bool use_plane = A.use_plane != 0, plane_dp = A.use_plane == 2;
  float* const red = lds;                                            // [W][64] partial sums of the lambda adjoint
  float* const pl_v = lds + W * GNS_LANES;                           // [N][64] planes (when they fit)
  float* const pl_th = pl_v + (use_plane ? N * GNS_LANES : 0);
  float* const pl_dp = pl_th + (plane_dp ? N * GNS_LANES : 0);
  const long long g = blockIdx.x;
  const long long R = gns_in_rows(N, E);
  const float* IN = A.in;
  const long long in_base = g * R, row_ein = in_base + 3LL * N, row_eout = row_ein + 3LL * E, row_grid = row_eout + E;
  const long long b = g * GNS_LANES + lane;
  const bool live = b < A.Bt;
  const float gt = (live && A.g_total) ? A.g_total[b] : 0.f;
  const float gl = (live && A.g_last) ? A.g_last[b] : 0.f;
  const int RB = A.RB, RBA = A.RBA;
  auto state_row = [&](int slot, int n) { return (((long long)slot * A.G + g) * N + n) * RB; };
  auto adj_row = [&](int n) { return (g * N + n) * RBA; };
  auto slot_ptr = [&](int j, int p) { return A.slots + ((g * 6 + j) * E + p) * GNS_LANES + lane; };
  const f4 gsum = *row_ptr(IN, row_grid, lane);
  const float invN = 1.0f / (float)N;
  const bool last = k == K - 1;
  const int mode = A.mode;                    // which kernels the sweeps of a step are (see gns_bwds_sweep_kernel)
  const bool x2_live = mode == 2 || k + 1 < K - 1;   // slot 2 of step k+1 exists (modes 0, 1: the last step runs no L_m sweep - no gradient reaches L_m.{K-1})
  // d total / d dp_{k+1}[n] = g_total * gamma^(K-k) * 2 dp / N  (+ g_last * 2 dp / N after the last step)  main.py:198-199
  const float cdp = 2.f * (gt * A.gwk + (last ? gl : 0.f)) * invN;
  const f2 lamv = reinterpret_cast<const f2*>(A.lam)[((long long)k * A.G + g) * GNS_LANES + lane];
  const int bits = (int)lamv.y;
  const bool low1 = bits & 1, low2 = bits & 2;

  // ---------------- Pb-0: closes step k+1 (identity paths + the input adjoints its sweeps collected), loss term ----------
  float lb = 0.f;
  constexpr int PB = 4;                        // buses per round: their row loads fly together
  for (int nb = n0; nb < n1; nb += PB) {
    f4 s1[PB], a0[PB], x0[PB], x1[PB], x2[PB], b1[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int n = min(nb + j, n1 - 1);
      const long long ar = adj_row(n);
      s1[j] = *row_ptr(A.state, state_row(k + 1, n), lane);
      b1[j] = *row_ptr(IN, in_base + 3LL * n + 1, lane);                  // Pmin,Pset,Pmax,Gs per bus
      if (!last) {
        // the X rows the sweeps of step k+1 wrote (slot 2: the kernel that ran L_m, 0: L_theta or L_theta + L_v, 1: L_v alone);
        // a slot that was not written is replaced by a dead load of a row that was
        a0[j] = *row_ptr(A.adj, ar, lane);
        x0[j] = *row_ptr(A.adj, ar + (mode == 2 ? 3 : 1), lane);
        x1[j] = *row_ptr(A.adj, ar + (mode == 0 ? 2 : (mode == 2 ? 3 : 1)), lane);
        x2[j] = *row_ptr(A.adj, ar + (x2_live ? 3 : 1), lane);
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int n = nb + j;
      if (n < n1) {
        f4 a;
        if (last) {
          // adjoints of the outputs: v_out = where(v < 0, 0, v) (main.py:201), theta_out = theta
          const float vb = (live && A.g_v) ? ((s1[j].x < 0.f) ? 0.f : A.g_v[b * N + n]) : 0.f;
          const float tb = (live && A.g_theta) ? A.g_theta[b * N + n] : 0.f;
          a = f4{vb, tb, 0.f, 0.f};
        } else {
          // the order of the old in-place accumulation: L_m first, then L_theta, then L_v
          f4 xs;
          if (mode == 2) xs = x2[j];
          else if (mode == 1) xs = x2_live ? f4{x2[j].x + x0[j].x, x2[j].y + x0[j].y, x2[j].z + x0[j].z, 0.f} : x0[j];
          else if (x2_live) xs = f4{(x2[j].x + x0[j].x) + x1[j].x, (x2[j].y + x0[j].y) + x1[j].y, (x2[j].z + x0[j].z) + x1[j].z, 0.f};
          else xs = f4{x0[j].x + x1[j].x, x0[j].y + x1[j].y, x0[j].z + x1[j].z, 0.f};
          a = f4{a0[j].x + xs.x, a0[j].y + xs.y, xs.z, 0.f};             // main.py:182,186 identity paths
        }
        a.z = a.z + cdp * s1[j].z;
        a.w = 2.f * b1[j].w * s1[j].x;        // 2 Gs v for the gather, which then needs neither the state row nor the input row
        *row_ptr(A.adj, adj_row(n), lane) = a;
        if (use_plane) { pl_v[n * GNS_LANES + lane] = s1[j].x; pl_th[n * GNS_LANES + lane] = s1[j].y; }
        if (plane_dp) pl_dp[n * GNS_LANES + lane] = a.z;
        lb += a.z * (low2 ? 2.f * (b1[j].y - b1[j].x) : 2.f * (b1[j].z - b1[j].y));    // d Pg_new / d lambda  (main.py:53-57)
      }
    }
  }
  red[wave * GNS_LANES + lane] = lb;
  __syncthreads();
  float lbar = 0.f;
#pragma unroll
  for (int w = 0; w < W; ++w) lbar += red[w * GNS_LANES + lane];
  const float pgbar = lbar / (low1 ? 2.f * (gsum.y - gsum.z) : 2.f * (gsum.w - gsum.y));   // d lambda / d p_global (main.py:47-51)

  // ---------------- line phase: per line, the adjoints of the line physics w.r.t. v, theta of its 2 (+4) buses ------------
  // Two lines per iteration: the row / LDS loads of both are issued before either is evaluated (the phase waits on loads).
  {
    const cip erec = topo + topo[TH_EREC];
    // (the adjoint of v on a generator bus feeds nothing - v is an input there and L_v never runs, main.py:184-186 - so the line phase
    //  does not store it and the gather does not collect it)
    struct EdgeIn { f4 e1v, o0; float vs, ths, vt, tht, tha, thb, thc, thd, Fb, Tb; bool gs, gt; };
    auto edge_load = [&](int p, EdgeIn& L) {
      // (s, t, a, b, q, c, d) of the line in one 32-byte scalar load (gns_topology.cpp)
      const gns_i8v r = *reinterpret_cast<const __attribute__((address_space(4))) gns_i8v*>(erec + 8 * p);
      const int s = r[0], t = r[1], ia = r[2], ib = r[3], q = r[4], ic = r[5], id = r[6];
      L.gs = is_gen[s] != 0; L.gt = is_gen[t] != 0;
      L.e1v = *row_ptr(IN, row_ein + 3LL * p + 1, lane);                   // shift_e, y_s, tau_s, sh_s
      L.o0 = *row_ptr(IN, row_eout + q, lane);                              // y_t, tau_t, sh_t, b_t
      if (use_plane) {
        L.vs = pl_v[s * GNS_LANES + lane]; L.ths = pl_th[s * GNS_LANES + lane];
        L.vt = pl_v[t * GNS_LANES + lane]; L.tht = pl_th[t * GNS_LANES + lane];
        L.tha = pl_th[ia * GNS_LANES + lane]; L.thb = pl_th[ib * GNS_LANES + lane];
        L.thc = pl_th[ic * GNS_LANES + lane]; L.thd = pl_th[id * GNS_LANES + lane];
        if (plane_dp) { L.Fb = pl_dp[t * GNS_LANES + lane]; L.Tb = pl_dp[s * GNS_LANES + lane]; }
        else { L.Fb = row_ptr(A.adj, adj_row(t), lane)->z; L.Tb = row_ptr(A.adj, adj_row(s), lane)->z; }
      } else {
        const f4 ss = *row_ptr(A.state, state_row(k + 1, s), lane), st = *row_ptr(A.state, state_row(k + 1, t), lane);
        L.vs = ss.x; L.ths = ss.y; L.vt = st.x; L.tht = st.y;
        L.tha = row_ptr(A.state, state_row(k + 1, ia), lane)->y; L.thb = row_ptr(A.state, state_row(k + 1, ib), lane)->y;
        L.thc = row_ptr(A.state, state_row(k + 1, ic), lane)->y; L.thd = row_ptr(A.state, state_row(k + 1, id), lane)->y;
        L.Fb = row_ptr(A.adj, adj_row(t), lane)->z;                          // dp[t] += p_from   (main.py:94)
        L.Tb = row_ptr(A.adj, adj_row(s), lane)->z;                          // dp[s] += p_to     (main.py:95)
      }
    };
    auto edge_adjoint = [&](int p, const EdgeIn& L) {
      const f4 e1v = L.e1v, o0 = L.o0;
      const float vs = L.vs, ths = L.ths, vt = L.vt, tht = L.tht, Fb = L.Fb, Tb = L.Tb;
      const float ys = e1v.y, taus = e1v.z, shs = e1v.w;
      const float dl = L.tha - L.thb, dl2 = L.thd - L.thc;
      float sA, cA, sB, cB, sD, cD, sC, cC, sD2, cD2;
      sincosf(ths - tht - dl - shs, &sA, &cA);
      sincosf(tht - ths - dl + shs, &sB, &cB);
      sincosf(dl, &sD, &cD);
      sincosf(tht - ths - dl2 - o0.z, &sC, &cC);
      sincosf(dl2, &sD2, &cD2);
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
      float dths = Ab - Bb;
      // "to" expression: p_to (main.py:92)
      const float yot_t = o0.x / o0.y;
      const float base2 = vt * vs * yot_t;
      dvt += Tb * (vs * yot_t * sC + 2.f * vt * o0.x * sD2);
      dvs += Tb * (vt * yot_t * sC);
      const float Cb = Tb * base2 * cC;
      const float dbar2 = Tb * vt * vt * o0.x * cD2 - Cb;
      dths -= Cb;
      // dtht == -dths bit for bit (Bb - Ab + Cb against Ab - Bb - Cb): plane 3 is not written, the gather negates plane 2
      if (!L.gs) *slot_ptr(0, p) = dvs;
      if (!L.gt) *slot_ptr(1, p) = dvt;
      *slot_ptr(2, p) = dths;
      *slot_ptr(4, p) = dbar; *slot_ptr(5, p) = dbar2;
    };
    for (int p = e0; p < e1; p += 2) {
      EdgeIn L0, L1;
      edge_load(p, L0);
      edge_load(min(p + 1, e1 - 1), L1);
      edge_adjoint(p, L0);
      if (p + 1 < e1) edge_adjoint(p + 1, L1);
    }
  }
  __syncthreads();

  // ---------------- gather: every bus completes d/d(v,theta)_{k+1} from the per-line adjoints, fixed order -----------------
  for (int n = n0; n < n1; ++n) {
    const long long ar = adj_row(n);
    const f4 a0 = *row_ptr(A.adj, ar, lane);
    float vbar = a0.x, thbar = a0.y;
    const float dpb = a0.z;
    const int p0 = in_ptr[n], p1 = in_ptr[n + 1], q0 = out_ptr[n], q1 = out_ptr[n + 1], i0 = incd_ptr[n], i1 = incd_ptr[n + 1];
    const bool gen = is_gen[n] != 0;               // vbar of a generator bus is never read: its two planes are neither stored nor gathered
    {
      float ai[4], bi[4], ao[4], bo[4], ci[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int pp = min(p0 + j, max(p1 - 1, p0)); ai[j] = gen ? 0.f : *slot_ptr(1, min(pp, E - 1)); bi[j] = -*slot_ptr(2, min(pp, E - 1)); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int p = q2p[min(min(q0 + j, max(q1 - 1, q0)), E - 1)]; ao[j] = gen ? 0.f : *slot_ptr(0, p); bo[j] = *slot_ptr(2, p); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int code = incd[min(min(i0 + j, max(i1 - 1, i0)), 4 * E - 1)];
        const float val = *slot_ptr((code & 2) ? 5 : 4, code >> 2);
        ci[j] = (code & 1) ? -val : val;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) if (p0 + j < p1) { vbar += ai[j]; thbar += bi[j]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) if (q0 + j < q1) { vbar += ao[j]; thbar += bo[j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (i0 + j < i1) thbar += ci[j];
    }
    for (int p = p0 + 4; p < p1; p += 4) {
      float a[4], bb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int pp = min(p + j, p1 - 1); a[j] = gen ? 0.f : *slot_ptr(1, pp); bb[j] = -*slot_ptr(2, pp); }
#pragma unroll
      for (int j = 0; j < 4; ++j) if (p + j < p1) { vbar += a[j]; thbar += bb[j]; }
    }
    for (int q = q0 + 4; q < q1; q += 4) {
      float a[4], bb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int p = q2p[min(q + j, q1 - 1)]; a[j] = gen ? 0.f : *slot_ptr(0, p); bb[j] = *slot_ptr(2, p); }
#pragma unroll
      for (int j = 0; j < 4; ++j) if (q + j < q1) { vbar += a[j]; thbar += bb[j]; }
    }
    for (int i = i0 + 8; i < i1; i += 8) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int code = incd[min(i + j, i1 - 1)];
        const float val = *slot_ptr((code & 2) ? 5 : 4, code >> 2);
        a[j] = (code & 1) ? -val : val;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (i + j < i1) thbar += a[j];
    }
    vbar += (pgbar - dpb) * a0.w;                  // a0.w = 2 Gs v (Pb-0): -Gs v^2 in dp (main.py:82) and +Gs v^2 in p_global (main.py:45)
    *row_ptr(A.adj, ar, lane) = f4{vbar, thbar, dpb, 0.f};
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The reverse update step of one bus, main.py:155-188 recomputed + reversed, in two objects: BwdsL = one bus net L_l (recompute, layer-wise
// backward, input adjoints), BwdsPhi = one phi net over the lines ending at the bus (head / tail recompute, backward, latent adjoint).
// Each holds its weight-gradient tiles and weight-stream heads, so that a workgroup can run one, two or all three families of a bus back
// to back on the rows it has loaded once.  The code is the V2 sweep of gns_backward.hip (see there for the window layout and the
// products that ride in spare rows / columns of a pass).  MULTI: every L net reads its own phi (main.py:157-167); single phi: the three
// L nets read the one hidden sum and its adjoint is their sum (main.py:169-171).
template <int D, int H, bool MULTI>
struct BwdsShape {
  using C = GnsDims<D, H, MULTI>;
  static constexpr int LIN = C::LF_IN, XL = (LIN + 1) / 2, PIN = C::PHI_IN, SOFF = 2 + D / 2;
  static constexpr int NB1 = (2 * XL + 15) / 16, NDM = (D + 15) / 16;
  static constexpr bool WIDE = D > 16;
  using SW = std::conditional_t<WIDE, GwSubWide, GwSub>;
  static constexpr int XT = 2 * XL - 16 * (NB1 - 1);                                          // x columns of the last dW1 window
  static constexpr int W2OFF = WIDE ? 16 - XT : 0;                                            // its first B column
  static constexpr int NDMF = WIDE ? 1 : NDM;                                                 // latent tiles left after the fold
  static_assert(!WIDE || ((D - 16) + (PIN - D) + 1 <= 16 && (PIN - D) % 2 == 1), "[line parameters | 1 | latent tail] in one window, the tail on a pair boundary");
  using L0 = WLink<false, true>;            // first of a group of linked weight streams (gns_device.h)
  using L1 = WLink<true, true>;             // middle
  using L2 = WLink<true, false>;            // last
};

template <int D, int H, bool MULTI, int l>
struct BwdsL : BwdsShape<D, H, MULTI> {
  using B = BwdsShape<D, H, MULTI>;
  using C = typename B::C;
  using SW = typename B::SW;
  using L0 = typename B::L0; using L1 = typename B::L1; using L2 = typename B::L2;
  static constexpr int LIN = B::LIN, XL = B::XL, SOFF = B::SOFF, NB1 = B::NB1, XT = B::XT, W2OFF = B::W2OFF;
  static constexpr bool WIDE = B::WIDE;
  static constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
  static constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1), NA4 = (OUTP + 11) / 12;
  using NL = NLay<LIN, H, OUTP>;
  static constexpr bool FOLD4 = WIDE && l < 2 && XT + H + 1 <= 16;                           // [x tail | a2 | 1] fits one window
  static constexpr bool FOLDM = GwSubWide::NA >= 16 && WIDE && l == 2 && XT + H + 1 <= 16 && H + (OUT - 16) <= 16;

  static constexpr int NT4 = FOLD4 ? 1 : (FOLDM ? 1 : NA4);      // tiles the output layer really uses (folded: none / one)
  f32x4 T1[NB1], T2, T4[NT4];
  cfp nb, ptl;

  __device__ __forceinline__ void init(const GnsBwdsArgs& A) {
    const long long koff = A.k;
    nb = (cfp)A.pn + A.n_off[C::NPHI + l] + koff * A.n_sz[C::NPHI + l];
    ptl = (cfp)A.pt + A.t_off[C::NPHI + l] + koff * A.t_sz[C::NPHI + l];
    scalar_cache_warm(ptl, A.t_sz[C::NPHI + l]);
#if !GNS_BWDS_ONE_LAYOUT
    scalar_cache_warm(nb, A.n_sz[C::NPHI + l]);
#endif
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NB1; ++t) T1[t] = z4;
#pragma unroll
    for (int t = 0; t < NT4; ++t) T4[t] = z4;
    T2 = z4;
  }

  // xs = [v theta | dp dq | m | (the hidden sum this net reads goes here) | deg, 1]; g3s = the upstream of the scalar output (L_theta:
  // thbar, L_v: vbar or 0 on a generator bus); L_m takes macc = d/dm_{k+1} as its upstream.  xsum (d/dv, d/dtheta, d/ddp of the L
  // inputs) and macc (d/dm) accumulate; gS = the adjoint of the hidden sum: assigned (ACC_GS false) or accumulated (the single phi).
  template <bool ACC_GS, bool STEP0>
  __device__ __forceinline__ void bus(const GnsBwdsArgs& A, float* rec, int lane, long long g, int n, float g3s, f2 (&xs)[XL],
                                      f2 (&macc)[D / 2], f4& xsum, f2 (&gS)[H / 2]) {
    f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(xs[SOFF]);
    load_pairs<H>(A.msg, ((((long long)A.k * A.G + g) * A.N + n) * C::NPHI + fphi) * C::HQ, lane, S);
    WFirst wf;
    f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
    f2 sl[2][H / 2];              // LeakyReLU's slopes at the hidden units, kept from the recomputation (GNS_BWDS_KEEP_SLOPES)
    using TL [[maybe_unused]] = TLay<LIN, H, OUTP>;   // the forward layout of the block: W1t b1 W2t b2 W4t b4
#if GNS_BWDS_ONE_LAYOUT
    (void)wf;
    mlp2_fwd<LIN, H>(ptl, xs, a1, a2, NoBG{}, NoLink{}, GNS_SLOPES(sl));
#else
    using LKm = std::conditional_t<STEP0, NoLink, L2>; using LKs = std::conditional_t<STEP0, std::conditional_t<FOLD4, L0, NoLink>, std::conditional_t<FOLD4, L1, L2>>;
    if constexpr (STEP0) {
      // m_0 = 0 (main.py:141): the latent rows of W1t multiply zeros - exact zeros, so leaving them out changes no bit.  The first
      // layer is the four state inputs (phi_head's shape) + the hidden sum, deg, the bias and the second layer (phi_tail's shape).
      f2 u4[H / 2];
      const f2 (&st)[2] = reinterpret_cast<const f2 (&)[2]>(xs[0]);
      const f2 (&tl)[(LIN - 4 - D + 1) / 2] = reinterpret_cast<const f2 (&)[(LIN - 4 - D + 1) / 2]>(xs[SOFF]);
      phi_head<4, H>(ptl, st, u4);
      phi_tail<LIN, H, 4 + D>(ptl, u4, tl, a1, a2, NoBG{}, NoLink{}, GNS_SLOPES(sl));
    } else {
      mlp2_fwd<LIN, H>(ptl, xs, a1, a2, NoBG{}, L0{nullptr, nb, &wf}, GNS_SLOPES(sl));
    }
#endif
    // output layer: g2 = (W4^T g3) * lrelu'(a2);  dW4 | db4 += g3 (x) [a2 | 1]
#if GNS_BWDS_ONE_LAYOUT
    if constexpr (l == 2) {
      bwd_rows_fwd_layout<H, OUTP>(ptl + TL::oW4, macc, g2);                      // m += L_m (main.py:188): g3 = mbar_{k+1}
    } else {
      const f2 g3v[1] = {f2{g3s, 0.f}};                                           // theta += L_theta (:182); v only without a generator (:184-186)
      bwd_rows_fwd_layout<H, 2>(ptl + TL::oW4, g3v, g2);
    }
#else
    if constexpr (l == 2) {
      bwd_rows<OUTP, H>(nb, macc, g2, NoBG{}, LKm{&wf, nullptr, nullptr});      // m += L_m (main.py:188); a pass follows
    } else {
      const f2 g3v[1] = {f2{g3s, 0.f}};                                           // theta += L_theta (:182); v only without a generator (:184-186)
      bwd_rows<2, H>(nb, g3v, g2, NoBG{}, LKs{&wf, nb + NL::oW2, &wf});
    }
#endif
#pragma unroll
    for (int u = 0; u < H / 2; ++u) g2[u] = g2[u] * GNS_SLOPE_OF(sl, 1, a2, u);
    if constexpr (FOLD4 || FOLDM) {                   // parked at columns 16..26 until the last dW1 window contracts them
      static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB<SW>(rec, lane, 8 + j, a2[j]); });
      gws_putB<SW>(rec, lane, 8 + H / 2, f2{1.f, 0.f});
      if constexpr (FOLDM) {
        static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, macc[j]); });
        gws_w2r(); gws_pass<SW, 16>(rec, lane, T4[0]); gws_r2w();
      }
    } else {
      static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB<SW>(rec, lane, j, a2[j]); });
      gws_putB<SW>(rec, lane, H / 2, f2{1.f, 0.f});
      static_for<0, NA4>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        if constexpr (l == 2) {
          static_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (6 * t + j < D / 2) gws_putA<SW>(rec, lane, j, macc[6 * t + j]); });
        } else {
          gws_putA<SW>(rec, lane, 0, f2{g3s, 0.f});
        }
        gws_w2r(); gws_pass<SW>(rec, lane, T4[t]); gws_r2w();
      });
    }
    // hidden layer: g1 = (W2^T g2) * lrelu'(a1);  dW2 | db2 += g2 (x) [a1 | 1]
#if GNS_BWDS_ONE_LAYOUT
    bwd_rows_fwd_layout<H, H>(ptl + TL::oW2, g2, g1);
#else
    bwd_rows<H, H>(nb + NL::oW2, g2, g1, NoBG{}, std::conditional_t<FOLD4, L2, NoLink>{&wf, nullptr, nullptr});
#endif
#pragma unroll
    for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * GNS_SLOPE_OF(sl, 0, a1, u);
    static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g2[j]); gws_putB<SW>(rec, lane, j, a1[j]); });
    gws_putB<SW>(rec, lane, H / 2, f2{1.f, 0.f});
    gws_w2r(); gws_pass<SW>(rec, lane, T2); gws_r2w();
    // first layer: dW1 | db1 += g1 (x) [x | 1] in 16-column windows
    static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g1[j]); });
    if constexpr (FOLD4) gws_putA<SW>(rec, lane, H / 2, f2{g3s, 0.f});            // row 10: g3
    if constexpr (FOLDM) static_for<0, (OUT - 16) / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, H / 2 + j, macc[8 + j]); });   // rows 10..: upstream of outputs 16..d-1 (macc is still mbar_{k+1} here)
    static_for<0, NB1>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int po = (WIDE && t == NB1 - 1) ? W2OFF / 2 : 0;                   // the last window sits right below column 16
      static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < XL) gws_putB<SW>(rec, lane, po + j, xs[8 * t + j]); });
      gws_w2r(); gws_pass<SW, (WIDE && t == NB1 - 1) ? W2OFF : 0>(rec, lane, T1[t]); gws_r2w();
    });
    // input adjoints, four at a time, straight to their consumers
    auto to_consumers = [&](auto ip_, f2 v) {
      constexpr int ip = decltype(ip_)::value;
      if constexpr (ip == 0) { xsum.x += v.x; xsum.y += v.y; }
      else if constexpr (ip == 1) xsum.z += v.x;
      else if constexpr (ip < SOFF) macc[ip - 2] += v;
      else if constexpr (ip < SOFF + H / 2) { if constexpr (ACC_GS) gS[ip - SOFF] += v; else gS[ip - SOFF] = v; }
    };
#if GNS_BWDS_ONE_LAYOUT
    bwd_from_fwd_layout<2 * (SOFF + H / 2), H, cf16p>(ptl, g1, to_consumers);      // the rows of [v theta dp dq | m | sum h]: deg carries no adjoint
#else
    if constexpr (STEP0) {
      // step 0: the adjoints of (v, theta, dp, m)_0 feed nothing (the inputs carry no gradient); only the hidden-sum adjoint is
      // needed, i.e. the groups of four inputs that hold columns SOFF*2 .. SOFF*2 + H - 1 of W1x
      constexpr int G0 = (2 * SOFF) / 4, G1 = (2 * SOFF + H - 1) / 4 + 1;
      bwd_inputs<G1 - G0, H>(nb + NL::total + G0 * H * 4, g1, [&](auto ip_, f2 v) {
        constexpr int ip = decltype(ip_)::value + 2 * G0;
        if constexpr (ip >= SOFF && ip < SOFF + H / 2) { if constexpr (ACC_GS) gS[ip - SOFF] += v; else gS[ip - SOFF] = v; }
      });
    } else {
      bwd_inputs<(LIN + 3) / 4, H>(nb + NL::total, g1, to_consumers);
    }
#endif
  }

  // the tiles -> the (family, step) block of the workgroup's slab (folded block: W1[H][IN] b1 W2 b2 W4 b4); plain stores
  __device__ __forceinline__ void flush(const GnsBwdsArgs& A, int lane, float* slab) {
    const long long koff = A.k;
    constexpr int ob1 = LIN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
    float* lb_ = slab + A.g_off[C::NPHI + l] + koff * A.g_sz[C::NPHI + l];
    static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value;
      gws_flush(lane, T1[t], lb_, [&](int c, int il) {
        const bool lastw = WIDE && t == NB1 - 1;                       // [x tail | a2 | 1]: the x columns end at XT
        const int i = 16 * t + il;
        if (c < H) return (lastw && il >= XT) ? -1 : (i < LIN ? c * LIN + i : (i == LIN ? ob1 + c : -1));
        if (FOLD4 && lastw && c == H && il >= XT && il <= XT + H) return il - XT < H ? oW4 + (il - XT) : ob4;   // row 10 = g3: dW4 | db4 of the scalar output
        if (FOLDM && lastw && c >= H && c < H + (OUT - 16) && il >= XT && il <= XT + H) { const int j = 16 + c - H; return il - XT < H ? oW4 + j * H + (il - XT) : ob4 + j; }
        return -1; }, true); });
    gws_flush(lane, T2, lb_, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; }, true);
    if constexpr (FOLDM)
      gws_flush(lane, T4[0], lb_, [&](int c, int il) { return il < H ? oW4 + c * H + il : (il == H ? ob4 + c : -1); }, true);   // outputs 0..15
    else if constexpr (!FOLD4)
    static_for<0, NA4>([&](auto t_) { constexpr int t = decltype(t_)::value;
      gws_flush(lane, T4[t], lb_, [&](int c, int il) { const int j = 12 * t + c; return (c < 12 && j < OUT) ? (il < H ? oW4 + j * H + il : (il == H ? ob4 + j : -1)) : -1; }, true); });
  }
};

template <int D, int H, bool MULTI, int PF>
struct BwdsPhi : BwdsShape<D, H, MULTI> {
  using B = BwdsShape<D, H, MULTI>;
  using C = typename B::C;
  using SW = typename B::SW;
  using L0 = typename B::L0; using L2 = typename B::L2;
  static constexpr int XL = B::XL, PIN = B::PIN, NDM = B::NDM, NDMF = B::NDMF;
  static constexpr bool WIDE = B::WIDE;

  f32x4 TP1, TP2, TPm[NDMF];
  cfp pnb, ptb;

  __device__ __forceinline__ void init(const GnsBwdsArgs& A) {
    const long long koff = A.k;
    pnb = (cfp)A.pn + A.n_off[PF] + koff * A.n_sz[PF];
    ptb = (cfp)A.pt + A.t_off[PF] + koff * A.t_sz[PF];
    scalar_cache_warm(ptb, A.t_sz[PF]);
#if !GNS_BWDS_ONE_LAYOUT
    scalar_cache_warm(pnb, A.n_sz[PF]);
#endif
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NDMF; ++t) TPm[t] = z4;
    TP1 = z4; TP2 = z4;
  }

  // back through the hidden vectors of the lines p0..p1 ending at the bus: gS = the adjoint of their sum
  template <bool STEP0>
  __device__ __forceinline__ void bus(const GnsBwdsArgs& A, float* rec, int lane, const f2 (&xs)[XL], const f2 (&gS)[H / 2],
                                      f2 (&macc)[D / 2], int p0, int p1, long long row_ein) {
    if (p0 >= p1) return;
    const float* IN = A.in;
    const f2 (&m)[D / 2] = reinterpret_cast<const f2 (&)[D / 2]>(xs[2]);
    WFirst wf;
    (void)wf;
    f2 uh[H / 2], G1[H / 2];
    if constexpr (WIDE)                               // the latent tail, parked behind [line parameters | 1] at columns 16..21
      static_for<0, (D - 16) / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB<SW>(rec, lane, 8 + (PIN - D + 2) / 2 + j, m[8 + j]); });
    // STEP0: m_0 = 0 (main.py:141): the bus share of phi's first layer is exactly zero, and so are d/dm_0's consumers
    if constexpr (STEP0) {
#pragma unroll
      for (int j = 0; j < H / 2; ++j) uh[j] = f2{0.f, 0.f};
    } else {
      phi_head<D, H>(ptb, m, uh);
    }
#pragma unroll
    for (int j = 0; j < H / 2; ++j) G1[j] = f2{0.f, 0.f};
    for (int p = p0; p < p1; ++p) {
      const f4 ea = *row_ptr(IN, row_ein + 3LL * p, lane), eb = *row_ptr(IN, row_ein + 3LL * p + 1, lane);
      const f2 xt[3] = {f2{ea.x, ea.y}, f2{ea.z, ea.w}, f2{eb.x, 0.f}};
      f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
      f2 sl[2][H / 2];
#if GNS_BWDS_ONE_LAYOUT
      phi_tail<PIN, H, D>(ptb, uh, xt, a1, a2, NoBG{}, NoLink{}, GNS_SLOPES(sl));
#pragma unroll
      for (int u = 0; u < H / 2; ++u) g2[u] = gS[u] * GNS_SLOPE_OF(sl, 1, a2, u);
      bwd_rows_fwd_layout<H, H>(ptb + TLay2<PIN, H>::oW2, g2, g1);
#else
      phi_tail<PIN, H, D>(ptb, uh, xt, a1, a2, NoBG{}, L0{nullptr, pnb, &wf}, GNS_SLOPES(sl));
#pragma unroll
      for (int u = 0; u < H / 2; ++u) g2[u] = gS[u] * GNS_SLOPE_OF(sl, 1, a2, u);
      bwd_rows<H, H>(pnb, g2, g1, NoBG{}, L2{&wf, nullptr, nullptr});
#endif
#pragma unroll
      for (int u = 0; u < H / 2; ++u) { g1[u] = g1[u] * GNS_SLOPE_OF(sl, 0, a1, u); G1[u] += g1[u]; }
      constexpr int pw = WIDE ? 8 : 0;                // wide window: the line's pass contracts columns 16..31
      static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g1[j]); });
      gws_putB<SW>(rec, lane, pw + 0, xt[0]); gws_putB<SW>(rec, lane, pw + 1, xt[1]); gws_putB<SW>(rec, lane, pw + 2, f2{xt[2].x, 1.f});
      gws_w2r(); gws_pass<SW, 2 * pw>(rec, lane, TP1); gws_r2w();
      static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g2[j]); gws_putB<SW>(rec, lane, j, a1[j]); });
      gws_putB<SW>(rec, lane, H / 2, f2{1.f, 0.f});
      gws_w2r(); gws_pass<SW>(rec, lane, TP2); gws_r2w();
    }
    // x = [m(dst) | ...] (main.py:155): d/dm += W1[:, :d]^T G1 and the latent columns of dW1 += G1 (x) m, once per bus
    if constexpr (STEP0) return;       // ... both zero at step 0: d/dm_0 is never read and m_0 = 0
#if GNS_BWDS_ONE_LAYOUT
    bwd_from_fwd_layout<D, H, cf16p>(ptb, G1, [&](auto ip_, f2 v) { macc[decltype(ip_)::value] += v; });   // the latent rows of W1t
#else
    bwd_inputs<(D + 3) / 4, H>(pnb + NLay2<PIN, H>::total, G1, [&](auto ip_, f2 v) {
      constexpr int ip = decltype(ip_)::value;
      if constexpr (ip < D / 2) macc[ip] += v;
    });
#endif
    static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, G1[j]); });
    static_for<0, NDMF>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < D / 2) gws_putB<SW>(rec, lane, j, m[8 * t + j]); });
      gws_w2r(); gws_pass<SW>(rec, lane, TPm[t]); gws_r2w();
    });
  }

  // the tiles -> the (family, step) block of the workgroup's slab (folded block: W1[H][IN] b1 W2 b2); plain stores
  __device__ __forceinline__ void flush(const GnsBwdsArgs& A, int lane, float* slab) {
    const long long koff = A.k;
    constexpr int pb1 = PIN * H, pW2 = pb1 + H, pb2 = pW2 + H * H;
    float* pb_ = slab + A.g_off[PF] + koff * A.g_sz[PF];
    gws_flush(lane, TP1, pb_, [&](int c, int il) {
      if (c >= H) return -1;
      if (il < PIN - D) return c * PIN + D + il;
      if (il == PIN - D) return pb1 + c;
      if (WIDE && il >= PIN - D + 1 && il < PIN - D + 1 + (D - 16)) return c * PIN + 16 + (il - (PIN - D + 1));   // latent tail m[16..D)
      return -1; }, true);
    gws_flush(lane, TP2, pb_, [&](int c, int il) { return c < H ? (il < H ? pW2 + c * H + il : (il == H ? pb2 + c : -1)) : -1; }, true);
    static_for<0, NDMF>([&](auto t_) { constexpr int t = decltype(t_)::value;
      gws_flush(lane, TPm[t], pb_, [&](int c, int il) { const int i = 16 * t + il; return (c < H && i < D) ? c * PIN + i : -1; }, true); });
  }
};

// FAMS: bit 0 L_theta, bit 1 L_v, bit 2 L_m - the families this kernel runs for every bus of its chunk, in the order L_m,
// L_theta, L_v of the old in-place accumulation.  The sweep modes (GnsBwdsArgs::mode) launch per step
//   0  three kernels {m} {theta} {v}     every family streams the bus rows on its own (72.6 rows of 1 KiB per bus and step)
//   1  two kernels   {m} {theta, v}      theta and v share one read of the bus's state, latent and line rows and one part of the
//                                         latent adjoint (51.4 rows)
//   2  one kernel    {m, theta, v}       bus-major: every row once, the latent adjoint accumulates in place (30.2 rows); the three
//                                         families' weights (20 KB) cycle through the 16 KB scalar cache
// A single-phi model (MULTI false) always runs mode 2: its three L nets share the one hidden sum and the adjoint of that sum is the
// sum over them (main.py:169-171), so the one phi net is reversed after all three - and its weights (14 KB) fit the scalar cache.
// A kernel writes ONE X row and ONE latent-adjoint part per bus: slot 2 when it runs L_m, else slot 0 (theta, or theta + v), else 1.
// Readers (Pb-0 for X, the L_m sweep for the latent adjoint) sum the slots of the mode in the order 2, 0, 1; slot 2 of step K-1
// exists only in mode 2 (no gradient reaches L_m.{K-1}: in modes 0 and 1 no kernel writes it).
template <int D, int H, bool MULTI, int FAMS, bool STEP0>      // STEP0: the instantiation that reverses step 0 (its dead work compiled out)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GNS_BWDS_WPE))) gns_bwds_sweep_kernel(GnsBwdsArgs A) {
  using C = GnsDims<D, H, MULTI>;
  static_assert(MULTI || FAMS == 7, "the single phi is reversed after all three L nets");
  constexpr int RB = C::RB, MQ = C::MQ, RBA = 4 + 6 * MQ;
  constexpr bool HAS_T = FAMS & 1, HAS_V = FAMS & 2, HAS_M = FAMS & 4;
  constexpr int SLOT = HAS_M ? 2 : (HAS_T ? 0 : 1);
  constexpr int LIN = C::LF_IN, XL = (LIN + 1) / 2;
  // the window alone (no pad behind the last row: every pass reads inside a row): 64 x 52 floats = 26 x 512 B, twelve per CU
  constexpr int RECF = 64 * (D > 16 ? GwSubWide::RS : GwSub::RS);
  __shared__ __attribute__((aligned(16))) float rec[RECF];
  const int lane = threadIdx.x;
  // blockIdx = (bus chunk, group block), the group block fastest: blocks 8 apart (same group block modulo 8) share an XCD
  const long long GB = (A.G + A.R - 1) / A.R;
  const long long bid = blockIdx.x;
  const long long gb = bid % GB;
  const int c = (int)(bid / GB);
  const int N = A.N, E = A.E, K = A.K, k = A.k;
  cip topo = (cip)A.topo;
  // (a partition of its own for the {L_theta, L_v} kernel, with generator buses at half weight, measured no better: 2.02 vs 2.00 ms)
  const cip part = topo + topo[TH_PART] + A.part_idx * (GNS_MAXP + 1), in_ptr = topo + topo[TH_IN_PTR], is_gen = topo + topo[TH_IS_GEN];
  const int n0 = part[c], n1 = part[c + 1];
  const long long g0 = gb * A.R, g1 = (g0 + A.R < A.G) ? g0 + A.R : A.G;
  float* slab = A.slab + (gb * A.C + c) * A.slab_floats;
  const bool lastk = k == K - 1;           // nothing reads m_K: L_m.{K-1} / phi_m.{K-1} get no gradient (reference: .grad is None)
  if (HAS_M && lastk) {                    // ... but their blocks of the slab must read as zero
    float* z0 = slab + A.g_off[C::NPHI + 2] + (long long)(K - 1) * A.g_sz[C::NPHI + 2];
    for (int i = lane; i < (int)A.g_sz[C::NPHI + 2]; i += 64) z0[i] = 0.f;
    if constexpr (MULTI) {
      float* z1 = slab + A.g_off[2] + (long long)(K - 1) * A.g_sz[2];
      for (int i = lane; i < (int)A.g_sz[2]; i += 64) z1[i] = 0.f;
    }
    if (!HAS_T && !HAS_V) return;
  }
  BwdsL<D, H, MULTI, 2> lm;
  BwdsL<D, H, MULTI, 0> lt;
  BwdsL<D, H, MULTI, 1> lv;
  BwdsPhi<D, H, MULTI, MULTI ? 2 : 0> pm;      // phi_m | the single phi
  BwdsPhi<D, H, MULTI, MULTI ? 1 : 0> pt;      // phi_theta (registration order phi_v, phi_theta, phi_m: main.py:113-116)
  BwdsPhi<D, H, MULTI, 0> pv;                  // phi_v
  if constexpr (HAS_M) { lm.init(A); pm.init(A); }
  if constexpr (HAS_T) { lt.init(A); if constexpr (MULTI) pt.init(A); }
  if constexpr (HAS_V) { lv.init(A); if constexpr (MULTI) pv.init(A); }
  // Step 0 reads m_0 = 0 and produces adjoints of (v, theta, dp, m)_0 that nothing reads (the inputs carry no gradient)
  constexpr bool step0 = STEP0;
  const int par = k & 1, parn = par ^ 1;
  const bool m2_live = (A.mode == 2) || k + 1 < K - 1;      // slot 2 of step k+1 was written
  const long long R = gns_in_rows(N, E);

  for (long long g = g0; g < g1; ++g) {
    const long long row_ein = g * R + 3LL * N;
    auto state_row = [&](int slot, int n) { return (((long long)slot * A.G + g) * N + n) * RB; };
    for (int n = n0; n < n1; ++n) {
      const long long ar = (g * N + n) * RBA, rr = state_row(k, n);
      const f4 a0 = *row_ptr(A.adj, ar, lane);
      const f4 s0 = *row_ptr(A.state, rr, lane);
      f2 xs[XL];                                          // [v theta | dp dq | m | sum_e h_e | deg, 1]
      f2 (&m)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(xs[2]);
      load_pairs<D>(A.state, (step0 ? state_row(0, n0) : rr) + 1, lane, m);   // m_0 = 0 for every bus: step 0 reads one (cached) bus's zero rows
      f2 macc[D / 2];                                     // d/dm: L_m's upstream (main.py:188) and / or this kernel's own terms
      if (HAS_M && !lastk) {
        // what the sweeps of step k+1 left, summed in the order of the old in-place accumulation (L_m, L_theta, L_v)
        const long long mr = ar + 4 + (long long)parn * 3 * MQ;
        if (A.mode == 2) {
          load_pairs<D>(A.adj, mr + 2 * MQ, lane, macc);
        } else {
          f2 p0v[D / 2];
          load_pairs<D>(A.adj, mr, lane, p0v);
          load_pairs<D>(A.adj, mr + (m2_live ? 2 : 0) * MQ, lane, macc);
          if (A.mode == 0) {
            f2 p1v[D / 2];
            load_pairs<D>(A.adj, mr + MQ, lane, p1v);
#pragma unroll
            for (int i = 0; i < D / 2; ++i) macc[i] = m2_live ? (macc[i] + p0v[i]) + p1v[i] : p0v[i] + p1v[i];
          } else {
#pragma unroll
            for (int i = 0; i < D / 2; ++i) macc[i] = m2_live ? macc[i] + p0v[i] : p0v[i];
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) macc[i] = f2{0.f, 0.f};
      }
      f4 xsum = f4{0.f, 0.f, 0.f, 0.f};                  // d/dv, d/dtheta, d/ddp of the L inputs of this kernel's families
      const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
      xs[0] = f2{s0.x, s0.y}; xs[1] = f2{s0.z, s0.w};
      xs[XL - 1] = f2{(float)(p1 - p0), 1.f};             // deg, and the 1 whose column of dW1 is db1
      f2 gS[H / 2];                                       // adjoint of the hidden-vector sum: what every line ending at n receives
      const float g3v = is_gen[n] ? 0.f : a0.x;
      // Every net of this bus broadcasts the same inputs into its first layer.  Seen as ONE value, a broadcast pair is built once, in
      // registers of its own, and kept for the whole bus (80 v_mov and 40 VGPRs for the latent vector alone); re-defined before each
      // net (an empty asm, no instruction), it is local to the net and folds into the op_sel bits of its packed FMAs.
      auto fresh = [&]() {
#if GNS_BWDS_FRESH_INPUTS
        pin_all(xs);
#endif
      };
      if constexpr (MULTI) {
        if constexpr (HAS_M) { if (!lastk) { fresh(); lm.template bus<false, STEP0>(A, rec, lane, g, n, 0.f, xs, macc, xsum, gS); fresh(); pm.template bus<STEP0>(A, rec, lane, xs, gS, macc, p0, p1, row_ein); } }
        if constexpr (HAS_T) { fresh(); lt.template bus<false, STEP0>(A, rec, lane, g, n, a0.y, xs, macc, xsum, gS); fresh(); pt.template bus<STEP0>(A, rec, lane, xs, gS, macc, p0, p1, row_ein); }
        // v moves only on buses without a generator (main.py:184-186): on a generator bus the upstream of L_v is exactly zero, and with
        // it every adjoint and every weight-gradient term of L_v and of phi_v over the lines ending there - the bus is skipped (the
        // topology is the same for all 64 grids of the wave, so the branch is uniform).  46 % of case118's buses carry a generator.
        if constexpr (HAS_V) { if (!is_gen[n]) { fresh(); lv.template bus<false, STEP0>(A, rec, lane, g, n, g3v, xs, macc, xsum, gS); fresh(); pv.template bus<STEP0>(A, rec, lane, xs, gS, macc, p0, p1, row_ein); } }
      } else {
#pragma unroll
        for (int j = 0; j < H / 2; ++j) gS[j] = f2{0.f, 0.f};
        if (!lastk) { fresh(); lm.template bus<true, STEP0>(A, rec, lane, g, n, 0.f, xs, macc, xsum, gS); }
        fresh(); lt.template bus<true, STEP0>(A, rec, lane, g, n, a0.y, xs, macc, xsum, gS);
        if (!is_gen[n]) { fresh(); lv.template bus<true, STEP0>(A, rec, lane, g, n, g3v, xs, macc, xsum, gS); }     // (a generator bus: L_v's upstream is zero, see above)
        fresh(); pm.template bus<STEP0>(A, rec, lane, xs, gS, macc, p0, p1, row_ein);
      }
      if constexpr (!step0) {
        *row_ptr(A.adj, ar + 1 + SLOT, lane) = xsum;
        store_pairs<D>(A.adj, ar + 4 + (par * 3 + SLOT) * MQ, lane, macc);
      }
    }
  }
  if constexpr (HAS_M) { if (!lastk) lm.flush(A, lane, slab); if (MULTI ? !lastk : true) pm.flush(A, lane, slab); }
  if constexpr (HAS_T) { lt.flush(A, lane, slab); if constexpr (MULTI) pt.flush(A, lane, slab); }
  if constexpr (HAS_V) { lv.flush(A, lane, slab); if constexpr (MULTI) pv.flush(A, lane, slab); }
}

int gns_bwds_supported(int d, int h, int multi) {
  (void)multi;                                  // three phi nets: modes 0-2; the single phi: the bus-major kernel
#define GNS_CASE(DD, HH) if (d == DD && h == HH) return 1;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return 0;
}

size_t gns_bwds_phys_lds(int N, int* use_plane) {
  const size_t red = (size_t)GNS_BWDS_PHYS_WAVES * GNS_LANES * 4;
  const size_t plane = (size_t)N * GNS_LANES * 4;
  const int level = red + 3 * plane <= (size_t)160 * 1024 ? 2 : (red + 2 * plane <= (size_t)160 * 1024 ? 1 : 0);
  if (use_plane) *use_plane = level;
  return red + (level == 2 ? 3 : (level == 1 ? 2 : 0)) * plane;
}

int gns_launch_bwds_phys(const GnsBwdsArgs& A, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL(gns_bwds_phys_kernel, dim3((unsigned)A.G), dim3(GNS_BWDS_PHYS_THREADS), lds, st, A);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}

int gns_launch_bwds_sweep(int d, int h, int multi, const GnsBwdsArgs& A, hipStream_t st) {
  const long long GB = (A.G + A.R - 1) / A.R;
  const unsigned blocks = (unsigned)(GB * A.C);
#define GNS_SWEEP(DD, HH, MM, FAMS) do { if (A.k == 0) hipLaunchKernelGGL((gns_bwds_sweep_kernel<DD, HH, MM, FAMS, true>), dim3(blocks), dim3(64), 0, st, A); \
                                         else hipLaunchKernelGGL((gns_bwds_sweep_kernel<DD, HH, MM, FAMS, false>), dim3(blocks), dim3(64), 0, st, A); } while (0)
#define GNS_CASE(DD, HH)                                                                                                    \
  if (d == DD && h == HH) {                                                                                                 \
    if (!multi) { if (A.mode != 2) return GNS_EINVAL; GNS_SWEEP(DD, HH, false, 7); }                                        \
    else if (A.mode == 2) GNS_SWEEP(DD, HH, true, 7);                                                                       \
    else if (A.mode == 1) { GNS_SWEEP(DD, HH, true, 4); GNS_SWEEP(DD, HH, true, 3); }                                       \
    else { GNS_SWEEP(DD, HH, true, 4); GNS_SWEEP(DD, HH, true, 1); GNS_SWEEP(DD, HH, true, 2); }                            \
    return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;                                                          \
  }
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
#undef GNS_SWEEP
  return GNS_EUNSUPPORTED;
}

// dynamic LDS above 64 KiB needs the attribute once per device (gns_api.hip calls this from its one-time initialisation)
int gns_bwds_init_device() {
  return hipFuncSetAttribute((const void*)gns_bwds_phys_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
             ? GNS_OK : GNS_ELAUNCH;
}
