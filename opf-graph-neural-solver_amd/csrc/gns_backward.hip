// Fused reverse pass of the GNS K-step loop: what autograd does for total_loss.backward()
// (GNS/main.py:288) through GNS.forward (main.py:140-202), hand-derived, for gfx950.
//
// Same ownership as the forward kernel: lane = grid, wave = 64 grids, the 8 waves of a workgroup split
// the buses.  Per reverse step k:
//   Pb-0    adjoint of delta_p_{k+1} is completed (loss term) and the scalar adjoint of lambda is reduced
//   Pb-edge per line: adjoints of the line physics w.r.t. v, theta of its 2 (+4 bus-id-as-line-index) buses,
//           stored per line (no atomics); every bus later gathers its own lists in a fixed order
//   Ub      per bus: gather, then the update step is recomputed and back-propagated; the weight gradient
//           (a contraction over the 64 grids of the wave) goes through an LDS transpose into register-resident
//           accumulators - fp32 MFMA tiles by default, 4x4 packed-FMA tiles with GNS_DW_MFMA=0 - that are
//           flushed once per (family, step) into a per-wave slab.
// delta_q carries no gradient: it is qg_new - Qd + Bs v^2 + (the very sums qg_new was built from), i.e.
// identically zero as a function of (v, theta) (main.py:64-76 vs :83,98-103).
#include "gns_device.h"
#include "gns_kernels.h"

#include "gns_dw.h"

#ifndef GNS_BWD_PLANES
#define GNS_BWD_PLANES 1          // 0 (diagnostic build): the line phase gathers its neighbour values from HBM / L2 rows
#endif
typedef int gns_i8v __attribute__((ext_vector_type(8)));   // one line record of TH_EREC

// ------------------------------------------------------------------------------------------------
// V2 (three phi nets, matrix-pipe engine): the family sweep runs the layer-wise data path - each layer's weight gradient is
// contracted through a 7 KB sub-record window as soon as its operands exist (half the LDS stores of the wide half-wave
// records), the input adjoints stream out four at a time into their consumers (no 36-register adjoint array), phi' is
// recomputed as head (once per bus) + tail (per line), and the latent columns of phi's dW1 and d/dm are taken once per bus
// on the sum of the lines' first-layer adjoints (phi's first layer is linear in m(dst)).
template <int D, int H, bool MULTI, bool MFMA, int VAR>
__global__ void __launch_bounds__(GNS_BWD_THREADS) gns_backward_kernel(GnsBwdArgs A) {
  constexpr bool V2 = VAR >= 2;          // 1: wide half-wave records   2: layer-wise sweep, sub-record windows   3: 2 + contraction chains drained behind the weight streams
  using C = GnsDims<D, H, MULTI>;
  constexpr int RB = C::RB;
  constexpr int W = GNS_BWD_WAVES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = A.N, E = A.E, K = A.K;
  cip topo = (cip)A.topo;
  cfp PT = (cfp)A.pt;
  cfp PN = (cfp)A.pn;
  const cip in_ptr = topo + topo[TH_IN_PTR], out_ptr = topo + topo[TH_OUT_PTR], is_gen = topo + topo[TH_IS_GEN],
            q2p = topo + topo[TH_Q2P], incd_ptr = topo + topo[TH_INCD_PTR],
            incd = topo + topo[TH_INCD], part = topo + topo[TH_PART] + A.part_idx * (GNS_MAXP + 1),
            epart = topo + topo[TH_EPART] + A.part_idx * (GNS_MAXP + 1);     // (the per-line indices come as one record: TH_EREC)
  // team of A.team workgroups per 64-grid group (gns_device.h, "teams"); blocks 8 apart share an XCD when the count allows it
  const int tsize = A.team, nteams = gridDim.x / tsize;
  int team_id = blockIdx.x, member = 0;
  if (tsize > 1) {
    if ((nteams & 7) == 0) { team_id = (blockIdx.x / (8 * tsize)) * 8 + (blockIdx.x & 7); member = (blockIdx.x >> 3) % tsize; }
    else { team_id = blockIdx.x / tsize; member = blockIdx.x % tsize; }
  }
  const int cw = member * W + wave, tw = tsize * W;
  const int n0 = part[cw], n1 = part[cw + 1];
  const int e0 = epart[cw], e1 = epart[cw + 1];
  const long long R = gns_in_rows(N, E);
  const float* IN = A.in;

  constexpr int RECF = VAR == 3 ? 64 * 76 + 32 : V2 ? (D > 16 ? GwSubWide::RECF : GwSub::RECF)
                          : gns_cmax(GNS_REC_ROWS * gns_cmax(RecLay<C::LF_IN, H, D>::RS, RecLay<C::LF_IN, H, 1>::RS),
                                     2 * GNS_REC_ROWS * RecLay2<C::PHI_IN, H>::RS) + 32;   // +32: the MFMA variant reads 16-wide column blocks
  __shared__ __attribute__((aligned(16))) float rec_all[W][RECF];
  __shared__ float red[2][W][GNS_LANES];
  __shared__ int team_failed;
  if (threadIdx.x == 0) team_failed = 0;
  GnsTeam team;
  team.size = tsize; team.member = member; team.epoch = 0; team.failed = &team_failed;
  __syncthreads();                                   // team_failed is initialised
  team_setup(team, reinterpret_cast<unsigned*>(A.team_ws + (long long)team_id * GNS_TEAM_CTR_BYTES));
  float* rec = rec_all[wave];
  // Line phase (Pb-edge): v, theta of the step being reversed and the adjoint of delta_p for every bus of the 64 grids, as
  // three [N][64] planes over the record windows (idle outside the family sweeps).  A line gathers 6 buses' theta, 2 buses'
  // v and 2 buses' adjoint: 10 LDS words instead of 8 HBM/L2 rows of 1 KiB of which 4-8 bytes per lane were used - those
  // gathers were 2/3 of the phase's traffic and the phase is bound by exactly that traffic (profiles/r02/ablation_backward_v2.txt).
  // One workgroup per group only (a team member sees only its own buses) and N <= 140 for 8 windows of 13 KiB.
  constexpr bool PLANES = GNS_BWD_PLANES != 0;
  const bool use_plane = PLANES && tsize == 1 && 3LL * N * GNS_LANES <= (long long)W * RECF;
  float* const pl_v = &rec_all[0][0];
  float* const pl_th = pl_v + (use_plane ? N * GNS_LANES : 0);
  float* const pl_dp = pl_th + (use_plane ? N * GNS_LANES : 0);
  float* slab = A.slab + ((long long)blockIdx.x * W + wave) * A.slab_floats;
  const float invN = 1.0f / (float)N;

#ifdef GNS_STAMPS
  long long tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#define STAMP(i) { const long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
  // finer segments inside the V2 family sweep; every stamp first drains the wave's outstanding memory operations, so a
  // segment is charged with its own latencies (this serialises what normally overlaps: shares, not absolute times)
  long long tsw[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tl2 = clock64();
#define SSTAMP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long tn = clock64(); tsw[i] += tn - tl2; tl2 = tn; }
#else
#define STAMP(i)
#define SSTAMP(i)
#endif
  if constexpr (V2) {
    if (A.slab_dirty) {     // the two blocks no sweep touches (L_m.{K-1}, phi_m.{K-1}: no gradient in the reference) must still read as zero
      float* z0 = slab + A.g_off[C::NPHI + 2] + (long long)(K - 1) * A.g_sz[C::NPHI + 2];
      for (int i = lane; i < (int)A.g_sz[C::NPHI + 2]; i += 64) z0[i] = 0.f;
      float* z1 = slab + A.g_off[2] + (long long)(K - 1) * A.g_sz[2];
      for (int i = lane; i < (int)A.g_sz[2]; i += 64) z1[i] = 0.f;
    }
  }
  for (long long g = team_id; g < A.G; g += nteams) {
    const bool first_store = V2 && A.slab_dirty && g == (long long)team_id;
    team.epoch = 0;
    team.ctr = reinterpret_cast<unsigned*>(A.team_ws + g * GNS_TEAM_CTR_BYTES);
    team.red = reinterpret_cast<float*>(A.team_ws + A.G * GNS_TEAM_CTR_BYTES) + g * GNS_TEAM_RED_FLOATS;
    const long long in_base = g * R, row_ein = in_base + 3LL * N, row_eout = row_ein + 3LL * E, row_grid = row_eout + E;
    const long long b = g * GNS_LANES + lane;
    const bool live = b < A.Bt;
    const float gt = (live && A.g_total) ? A.g_total[b] : 0.f;
    const float gl = (live && A.g_last) ? A.g_last[b] : 0.f;
#ifdef GNS_ABLATE_HBM     // diagnostic: every workgroup re-reads / re-writes the rows of 4 buses of one state slot (cache resident): what the HBM stream costs
    auto state_row = [&](int slot, int n) { return (((long long)0 * A.G + g) * N + (n & 3)) * RB; };
#else
    auto state_row = [&](int slot, int n) { return (((long long)slot * A.G + g) * N + n) * RB; };
#endif
    // rows per bus: (vbar, thbar, dpbar, -) | input-adjoint sums of this step | [single phi: adjoint of the hidden sum] | mbar
    constexpr int RHB = MULTI ? 0 : C::HQ;
    constexpr int RBA = RB + 1 + RHB, RM = 2 + RHB;
#ifdef GNS_ABLATE_HBM
    auto adj_row = [&](int n) { return (g * N + (n & 3)) * RBA; };
#else
    auto adj_row = [&](int n) { return (g * N + n) * RBA; };
#endif
    auto slot_ptr = [&](int j, int p) { return A.slots + ((g * 6 + j) * E + p) * GNS_LANES + lane; };
    const f4 gsum = *row_ptr(IN, row_grid, lane);

    // adjoints of the outputs: v_out = where(v < 0, 0, v) (main.py:201), theta_out = theta
    for (int n = n0; n < n1; ++n) {
      const f4 sK = *row_ptr(A.state, state_row(K, n), lane);
      const float vb = (live && A.g_v) ? ((sK.x < 0.f) ? 0.f : A.g_v[b * N + n]) : 0.f;
      const float tb = (live && A.g_theta) ? A.g_theta[b * N + n] : 0.f;
      const long long ar = adj_row(n);
      *row_ptr(A.adj, ar, lane) = f4{vb, tb, 0.f, 0.f};
      // (VAR 2: nothing else to clear - the first sweep of a step starts the input-adjoint row itself, and the latent adjoint is
      //  known to be zero where it is first read, in the L_theta sweep of the last step)
      if constexpr (VAR != 2) {
#pragma unroll
        for (int q = 0; q < RBA - 1; ++q) *row_ptr(A.adj, ar + 1 + q, lane) = f4{0.f, 0.f, 0.f, 0.f};
      }
    }

    for (int k = K - 1; k >= 0; --k) {
      const long long koff = k;
      // d total / d dp_{k+1}[n] = g_total * gamma^(K-k) * 2 dp / N  (+ g_last * 2 dp / N after the last step)  main.py:198-199
      const float cdp = 2.f * (gt * A.gw[k] + (k == K - 1 ? gl : 0.f)) * invN;
      const f2 lamv = reinterpret_cast<const f2*>(A.lam)[((long long)k * A.G + g) * GNS_LANES + lane];
      const int bits = (int)lamv.y;
      const bool low1 = bits & 1, low2 = bits & 2;

      // ---------------- Pb-0 (also closes step k+1: identity paths + the input adjoints its sweeps collected) --------
      if (use_plane) __syncthreads();                      // every wave has left the sweeps: the record windows become the planes
      float lb = 0.f;
      for (int nb = n0; nb < n1; nb += 4) {                                  // four buses per round: 16 independent row loads in flight
        f4 s1[4], a0[4], xs[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = min(nb + j, n1 - 1);
          const long long ar = adj_row(n);
          s1[j] = *row_ptr(A.state, state_row(k + 1, n), lane);
          a0[j] = *row_ptr(A.adj, ar, lane);
          xs[j] = *row_ptr(A.adj, ar + 1, lane);
          b1[j] = *row_ptr(IN, in_base + 3LL * n + 1, lane);                // Pmin,Pset,Pmax,Gs per bus
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nb + j;
          if (n < n1) {
            const long long ar = adj_row(n);
            f4 a = a0[j];
            if (k < K - 1) {
              a = f4{a.x + xs[j].x, a.y + xs[j].y, xs[j].z, 0.f};           // main.py:182,186 identity paths
#pragma unroll
              for (int q = (VAR == 2 ? 1 : 0); q < 1 + RHB; ++q) *row_ptr(A.adj, ar + 1 + q, lane) = f4{0.f, 0.f, 0.f, 0.f};   // (VAR 2: the first sweep of a step starts the input-adjoint row itself)
            }
            a.z = a.z + cdp * s1[j].z;
            a.w = 2.f * b1[j].w * s1[j].x;        // 2 Gs v for pass G, which then needs neither the state row nor the input row of the bus
            *row_ptr(A.adj, ar, lane) = a;
            if (use_plane) { pl_v[n * GNS_LANES + lane] = s1[j].x; pl_th[n * GNS_LANES + lane] = s1[j].y; pl_dp[n * GNS_LANES + lane] = a.z; }
            lb += a.z * (low2 ? 2.f * (b1[j].y - b1[j].x) : 2.f * (b1[j].z - b1[j].y));    // d Pg_new / d lambda  (main.py:53-57)
          }
        }
      }
      STAMP(0)
      if (tsize == 1) red[k & 1][wave][lane] = lb;
      else team.red[((k & 1) * GNS_MAXP + cw) * GNS_LANES + lane] = lb;
      team_barrier(team);
      STAMP(1)
      float lbar = 0.f;
      if (tsize == 1) {
#pragma unroll
        for (int w = 0; w < W; ++w) lbar += red[k & 1][w][lane];
      } else {
        for (int w = 0; w < tw; ++w) lbar += team.red[((k & 1) * GNS_MAXP + w) * GNS_LANES + lane];
      }
      const float pgbar = lbar / (low1 ? 2.f * (gsum.y - gsum.z) : 2.f * (gsum.w - gsum.y));   // d lambda / d p_global (main.py:47-51)

      // ---------------- Pb-edge ----------------
      // Two lines per iteration: the 24 loads of both are issued before either is used (the phase waits on loads, not
      // on the VALU, and a wave has only one partner on its SIMD to hide them).
      struct EdgeIn { f4 e1v, o0, ss, st; float tha, thb, thc, thd, Fb, Tb; };
      const cip erec = topo + topo[TH_EREC];
      auto edge_load = [&](int p, EdgeIn& L) {
        // (s, t, a, b, q, c, d) of the line in one 32-byte scalar load (gns_topology.cpp)
        const gns_i8v r = *reinterpret_cast<const __attribute__((address_space(4))) gns_i8v*>(erec + 8 * p);
        const int s = r[0], t = r[1], ia = r[2], ib = r[3], q = r[4], ic = r[5], id = r[6];
        L.e1v = *row_ptr(IN, row_ein + 3LL * p + 1, lane);                  // shift_e, y_s, tau_s, sh_s
        L.o0 = *row_ptr(IN, row_eout + q, lane);                             // y_t, tau_t, sh_t, b_t
        if (use_plane) {
          L.ss = f4{pl_v[s * GNS_LANES + lane], pl_th[s * GNS_LANES + lane], 0.f, 0.f};
          L.st = f4{pl_v[t * GNS_LANES + lane], pl_th[t * GNS_LANES + lane], 0.f, 0.f};
          L.tha = pl_th[ia * GNS_LANES + lane]; L.thb = pl_th[ib * GNS_LANES + lane];
          L.thc = pl_th[ic * GNS_LANES + lane]; L.thd = pl_th[id * GNS_LANES + lane];
          L.Fb = pl_dp[t * GNS_LANES + lane]; L.Tb = pl_dp[s * GNS_LANES + lane];
          return;
        }
        L.ss = *row_ptr(A.state, state_row(k + 1, s), lane); L.st = *row_ptr(A.state, state_row(k + 1, t), lane);
        L.tha = row_ptr(A.state, state_row(k + 1, ia), lane)->y; L.thb = row_ptr(A.state, state_row(k + 1, ib), lane)->y;
        L.thc = row_ptr(A.state, state_row(k + 1, ic), lane)->y; L.thd = row_ptr(A.state, state_row(k + 1, id), lane)->y;
        L.Fb = row_ptr(A.adj, adj_row(t), lane)->z;                          // dp[t] += p_from   (main.py:94)
        L.Tb = row_ptr(A.adj, adj_row(s), lane)->z;                          // dp[s] += p_to     (main.py:95)
      };
      struct EdgeOut { float dvs, dvt, dths, dbar, dbar2; };
      auto edge_store = [&](int p, const EdgeOut& R) {
        // dtht == -dths bit for bit (Bb - Ab + Cb against Ab - Bb - Cb): plane 3 is not written, the gather negates plane 2
        *slot_ptr(0, p) = R.dvs; *slot_ptr(1, p) = R.dvt; *slot_ptr(2, p) = R.dths;
        *slot_ptr(4, p) = R.dbar; *slot_ptr(5, p) = R.dbar2;
      };
      auto edge_adjoint = [&](int p, const EdgeIn& L, EdgeOut& R) {
        const f4 e1v = L.e1v, o0 = L.o0;
        const float Fb = L.Fb, Tb = L.Tb;
        const float vs = L.ss.x, ths = L.ss.y, vt = L.st.x, tht = L.st.y;
        const float ys = e1v.y, taus = e1v.z, shs = e1v.w;
        const float dl = L.tha - L.thb, dl2 = L.thd - L.thc;
        float sA, cA, sB, cB, sD, cD, sC, cC, sD2, cD2;
#ifdef GNS_ABLATE_PHYS_TRIG       // diagnostic: no trigonometric evaluation (results wrong by design)
        sA = ths - tht - dl - shs; cA = 1.f - sA; sB = tht - ths - dl + shs; cB = 1.f - sB; sD = dl; cD = 1.f - dl;
        sC = tht - ths - dl2 - o0.z; cC = 1.f - sC; sD2 = dl2; cD2 = 1.f - dl2;
#else
        sincosf(ths - tht - dl - shs, &sA, &cA);
        sincosf(tht - ths - dl + shs, &sB, &cB);
        sincosf(dl, &sD, &cD);
        sincosf(tht - ths - dl2 - o0.z, &sC, &cC);
        sincosf(dl2, &sD2, &cD2);
#endif
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
        const float yot_t = o0.x / o0.y;
        const float base2 = vt * vs * yot_t;
        dvt += Tb * (vs * yot_t * sC + 2.f * vt * o0.x * sD2);
        dvs += Tb * (vt * yot_t * sC);
        const float Cb = Tb * base2 * cC;
        const float dbar2 = Tb * vt * vt * o0.x * cD2 - Cb;
        dtht += Cb; dths -= Cb;
        (void)dtht;
#ifdef GNS_ABLATE_PHYS_STORE      // diagnostic: the results are computed but not stored
        asm volatile("" :: "v"(dvs), "v"(dvt), "v"(dths), "v"(dbar), "v"(dbar2));
        R = EdgeOut{0.f, 0.f, 0.f, 0.f, 0.f};
#else
        R = EdgeOut{dvs, dvt, dths, dbar, dbar2};
#endif
      };
#ifdef GNS_ABLATE_PHYS
      for (int p = e0; p < e0; p += 2) {
#else
      for (int p = e0; p < e1; p += 2) {
#endif
        EdgeIn L0, L1;
        EdgeOut R0, R1;
        edge_load(p, L0);
        edge_load(min(p + 1, e1 - 1), L1);
        edge_adjoint(p, L0, R0);
        edge_store(p, R0);
        if (p + 1 < e1) { edge_adjoint(p + 1, L1, R1); edge_store(p + 1, R1); }
      }
      STAMP(2)
      team_barrier(team);
      STAMP(3)

      // ---------------- Ub, pass G: every bus completes d/d(v,theta)_{k+1} from the per-line adjoints --------------
#ifdef GNS_ABLATE_GATHER
      for (int n = n0; n < n0; ++n) {
#else
      for (int n = n0; n < n1; ++n) {
#endif
        const long long ar = adj_row(n);
        const f4 a0 = *row_ptr(A.adj, ar, lane);
        float vbar = a0.x, thbar = a0.y;
        const float dpb = a0.z;
        // The three lists (lines ending here, lines leaving here, angle-difference incidences) are read with clamped
        // indices, the first 4 + 4 + 8 entries in ONE round of independent loads (most buses need no second round):
        // one entry per iteration would pay the full (scalar index -> vector load) latency per entry and per list.
        const int p0 = in_ptr[n], p1 = in_ptr[n + 1], q0 = out_ptr[n], q1 = out_ptr[n + 1], i0 = incd_ptr[n], i1 = incd_ptr[n + 1];
        {
          float ai[4], bi[4], ao[4], bo[4], ci[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int pp = min(p0 + j, max(p1 - 1, p0)); ai[j] = *slot_ptr(1, min(pp, E - 1)); bi[j] = -*slot_ptr(2, min(pp, E - 1)); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int p = q2p[min(min(q0 + j, max(q1 - 1, q0)), E - 1)]; ao[j] = *slot_ptr(0, p); bo[j] = *slot_ptr(2, p); }
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
          float a[4], b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int pp = min(p + j, p1 - 1); a[j] = *slot_ptr(1, pp); b[j] = -*slot_ptr(2, pp); }
#pragma unroll
          for (int j = 0; j < 4; ++j) if (p + j < p1) { vbar += a[j]; thbar += b[j]; }
        }
        for (int q = q0 + 4; q < q1; q += 4) {
          float a[4], b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int p = q2p[min(q + j, q1 - 1)]; a[j] = *slot_ptr(0, p); b[j] = *slot_ptr(2, p); }
#pragma unroll
          for (int j = 0; j < 4; ++j) if (q + j < q1) { vbar += a[j]; thbar += b[j]; }
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

      STAMP(4)
      // ---------------- Ub, one pass over the buses per network family (main.py:155-188 recomputed + reversed) ------
      // Family-outer order keeps the 4x4 weight-gradient tiles of (L_l, phi_f) in registers for the whole pass.
      auto edge_input = [&](int p, const f2 (&m)[D / 2], f2 (&x)[(C::PHI_IN + 1) / 2]) {
        const f4 ea = *row_ptr(IN, row_ein + 3LL * p, lane), eb = *row_ptr(IN, row_ein + 3LL * p + 1, lane);
#pragma unroll
        for (int i = 0; i < D / 2; ++i) x[i] = m[i];
        x[D / 2] = f2{ea.x, ea.y}; x[D / 2 + 1] = f2{ea.z, ea.w}; x[D / 2 + 2] = f2{eb.x, 0.f};
      };
      if constexpr (VAR == 3) {
      // ---------------- V3: the V2 sweep with the weight-gradient chains issued in the background of the NEXT weight streams ----
      static_for<0, 3>([&](auto o_) {
        constexpr int l = (decltype(o_)::value == 0) ? 2 : decltype(o_)::value - 1;
        constexpr int fphi = l == 0 ? 1 : (l == 1 ? 0 : 2);
        constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1);
        constexpr int LIN = C::LF_IN, XL = (LIN + 1) / 2, PIN = C::PHI_IN, SOFF = 2 + D / 2;
        constexpr int NB1 = (2 * XL + 15) / 16, NA4 = (OUTP + 11) / 12, NDM = (D + 15) / 16;
        using NL = NLay<LIN, H, OUTP>;
        if (l == 2 && k == K - 1) return;
        // windows (all in the wave's record buffer, one at a time):
        //   L  [g1 0..11 | x,1 12..47 | g2 48..59 | a1,1 60..71]   RS 76   chains: dW2 (T2), dW1 tile t (T1[t])
        //   E  [g1 0..11 | line parameters,1 12..19 | g2 20..31 | a1,1 32..43]   RS 44   chains: TP1, TP2
        //   B  [G1 0..11 | m 12..31]   RS 36   chains: TPm[t]
        using PL = GwProgL<NB1>;
        using PE = GwProgE;
        using PB = GwProgB<NDM>;
        static_assert(PB::LEN <= 45, "the bus window must drain inside the next bus's recomputation");
        cfp nb = PN + A.n_off[C::NPHI + l] + koff * A.n_sz[C::NPHI + l];
        cfp pnb = PN + A.n_off[fphi] + koff * A.n_sz[fphi];
        cfp ptb = PT + A.t_off[fphi] + koff * A.t_sz[fphi];
        f32x4 T1[NB1], T2, T4[NA4], TP1, TP2, TPm[NDM];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NB1; ++t) T1[t] = z4;
#pragma unroll
        for (int t = 0; t < NA4; ++t) T4[t] = z4;
#pragma unroll
        for (int t = 0; t < NDM; ++t) TPm[t] = z4;
        T2 = z4; TP1 = z4; TP2 = z4;
        float ra[4], rb[4];                                   // operand ring of the background chains
        const int rowmap = ((lane >> 5) & 1) + 4 * ((lane >> 4) & 1);
        const float* baseL = rec + rowmap * PL::RS + (lane & 15);
        const float* baseE = rec + rowmap * PE::RS + (lane & 15);
        const float* baseB = rec + rowmap * PB::RS + (lane & 15);
        auto accL = [&](auto ch_) -> f32x4& { constexpr int ch = decltype(ch_)::value; if constexpr (ch == 0) return T2; else return T1[ch - 1]; };
        auto accE = [&](auto ch_) -> f32x4& { constexpr int ch = decltype(ch_)::value; if constexpr (ch == 0) return TP1; else return TP2; };
        auto accB = [&](auto ch_) -> f32x4& { constexpr int ch = decltype(ch_)::value; return TPm[ch]; };
        auto putw = [&](int rs, int off, f2 v) { *reinterpret_cast<f2*>(rec + lane * rs + off) = v; };
        // invariant at the top of every bus: a B window is pending (before the first bus: an all-zero one)
        static_for<0, PB::RS / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PB::RS, 2 * j, f2{0.f, 0.f}); });
        gws_w2r();
        for (int n = n0; n < n1; ++n) {
          const long long ar = adj_row(n), rr = state_row(k, n);
          const f4 a0 = *row_ptr(A.adj, ar, lane);
          f4 xsum = *row_ptr(A.adj, ar + 1, lane);
          const f4 s0 = *row_ptr(A.state, rr, lane);
          f2 xs[XL];
          f2 (&m)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(xs[2]);
          f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(xs[SOFF]);
          load_pairs<D>(A.state, rr + 1, lane, m);
          f2 macc[D / 2];
          load_pairs<D>(A.adj, ar + RM, lane, macc);
          load_pairs<H>(A.msg, ((((long long)k * A.G + g) * N + n) * C::NPHI + fphi) * C::HQ, lane, S);
          const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
          xs[0] = f2{s0.x, s0.y}; xs[1] = f2{s0.z, s0.w};
          xs[XL - 1] = f2{(float)(p1 - p0), 1.f};
          f2 gS[H / 2];
          f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
          {   // recomputation of L' with the previous bus's B window draining behind it
            auto dr = gw_drain<PB, 0>(baseB, ra, rb, accB);
            dr.prologue();
            mlp2_fwd<LIN, H>(PT + A.t_off[C::NPHI + l] + koff * A.t_sz[C::NPHI + l], xs, a1, a2, dr);
            gws_r2w();
          }
          // output layer: g2 = (W4^T g3) * lrelu'(a2);  dW4 | db4 += g3 (x) [a2 | 1]   (not overlapped)
          if constexpr (l == 2) {
            bwd_rows<OUTP, H>(nb, macc, g2);
          } else {
            const f2 g3s[1] = {f2{l == 0 ? a0.y : (is_gen[n] ? 0.f : a0.x), 0.f}};
            bwd_rows<2, H>(nb, g3s, g2);
          }
#pragma unroll
          for (int u = 0; u < H / 2; ++u) g2[u] = g2[u] * dlrelu2(a2[u]);
          static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB(rec, lane, j, a2[j]); });
          gws_putB(rec, lane, H / 2, f2{1.f, 0.f});
          static_for<0, NA4>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            if constexpr (l == 2) {
              static_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (6 * t + j < D / 2) gws_putA(rec, lane, j, macc[6 * t + j]); });
            } else {
              gws_putA(rec, lane, 0, f2{l == 0 ? a0.y : (is_gen[n] ? 0.f : a0.x), 0.f});
            }
            gws_w2r(); gws_pass(rec, lane, T4[t]); gws_r2w();
          });
          // hidden layer
          bwd_rows<H, H>(nb + NL::oW2, g2, g1);
#pragma unroll
          for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * dlrelu2(a1[u]);
          // L window: dW2 | db2 and dW1 | db1 drain behind the input-adjoint stream (and the phi head of this bus)
          static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PL::RS, 2 * j, g1[j]); putw(PL::RS, 48 + 2 * j, g2[j]); putw(PL::RS, 60 + 2 * j, a1[j]); });
          putw(PL::RS, 60 + H, f2{1.f, 0.f});
          static_for<0, XL>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PL::RS, 12 + 2 * j, xs[j]); });
          gws_w2r();
          constexpr int SG = 3 * (((LIN + 3) / 4 * H * 4 + 31) / 32);          // background slots of the input-adjoint stream
          {
            auto dr = gw_drain<PL, 0>(baseL, ra, rb, accL);
            dr.prologue();
            bwd_inputs<(LIN + 3) / 4, H>(nb + NL::total, g1, [&](auto ip_, f2 v) {
              constexpr int ip = decltype(ip_)::value;
              if constexpr (ip == 0) { xsum.x += v.x; xsum.y += v.y; }
              else if constexpr (ip == 1) xsum.z += v.x;
              else if constexpr (ip < SOFF) macc[ip - 2] += v;
              else if constexpr (ip < SOFF + H / 2) gS[ip - SOFF] = v;
            }, dr);
          }
          f2 G1[H / 2];
#pragma unroll
          for (int j = 0; j < H / 2; ++j) G1[j] = f2{0.f, 0.f};
          if (p0 < p1) {
            f2 uh[H / 2];
            constexpr int SH = 3 * ((D * H + 31) / 32);                        // slots of the phi head
            {
              auto dr = gw_drain<PL, SG>(baseL, ra, rb, accL);
              phi_head<D, H>(ptb, m, uh, dr);
              dr.template rest<SH>();
              gws_r2w();
            }
            constexpr int ST = 3 * ((TLay2<PIN, H>::total - D * H + 31) / 32), SB = 3 * ((H * H + 31) / 32);   // slots of a line's tail / hidden backward
            auto edge_window = [&](const f2 (&xt)[3], const f2 (&ea1)[H / 2], const f2 (&eg1)[H / 2], const f2 (&eg2)[H / 2]) {
              static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PE::RS, 2 * j, eg1[j]); putw(PE::RS, 20 + 2 * j, eg2[j]); putw(PE::RS, 32 + 2 * j, ea1[j]); });
              putw(PE::RS, 12, xt[0]); putw(PE::RS, 14, xt[1]); putw(PE::RS, 16, f2{xt[2].x, 1.f}); putw(PE::RS, 18, f2{0.f, 0.f});
              putw(PE::RS, 32 + H, f2{1.f, 0.f});
              gws_w2r();
            };
            {   // first line: nothing is pending
              const f4 ea = *row_ptr(IN, row_ein + 3LL * p0, lane), eb = *row_ptr(IN, row_ein + 3LL * p0 + 1, lane);
              const f2 xt[3] = {f2{ea.x, ea.y}, f2{ea.z, ea.w}, f2{eb.x, 0.f}};
              f2 ea1[H / 2], ea2[H / 2], eg2[H / 2], eg1[H / 2];
              phi_tail<PIN, H, D>(ptb, uh, xt, ea1, ea2);
#pragma unroll
              for (int u = 0; u < H / 2; ++u) eg2[u] = gS[u] * dlrelu2(ea2[u]);
              bwd_rows<H, H>(pnb, eg2, eg1);
#pragma unroll
              for (int u = 0; u < H / 2; ++u) { eg1[u] = eg1[u] * dlrelu2(ea1[u]); G1[u] += eg1[u]; }
              edge_window(xt, ea1, eg1, eg2);
            }
            for (int p = p0 + 1; p < p1; ++p) {       // every further line drains its predecessor's window behind its own streams
              const f4 ea = *row_ptr(IN, row_ein + 3LL * p, lane), eb = *row_ptr(IN, row_ein + 3LL * p + 1, lane);
              const f2 xt[3] = {f2{ea.x, ea.y}, f2{ea.z, ea.w}, f2{eb.x, 0.f}};
              f2 ea1[H / 2], ea2[H / 2], eg2[H / 2], eg1[H / 2];
              auto dr0 = gw_drain<PE, 0>(baseE, ra, rb, accE);
              dr0.prologue();
              phi_tail<PIN, H, D>(ptb, uh, xt, ea1, ea2, dr0);
#pragma unroll
              for (int u = 0; u < H / 2; ++u) eg2[u] = gS[u] * dlrelu2(ea2[u]);
              auto dr1 = gw_drain<PE, ST>(baseE, ra, rb, accE);
              bwd_rows<H, H>(pnb, eg2, eg1, dr1);
              dr1.template rest<SB>();
              gws_r2w();
#pragma unroll
              for (int u = 0; u < H / 2; ++u) { eg1[u] = eg1[u] * dlrelu2(ea1[u]); G1[u] += eg1[u]; }
              edge_window(xt, ea1, eg1, eg2);
            }
            {   // d/dm += W1[:, :d]^T G1 with the last line's window draining behind it
              constexpr int SP = 3 * (((D + 3) / 4 * H * 4 + 31) / 32);
              auto dr = gw_drain<PE, 0>(baseE, ra, rb, accE);
              dr.prologue();
              bwd_inputs<(D + 3) / 4, H>(pnb + NLay2<PIN, H>::total, G1, [&](auto ip_, f2 v) {
                constexpr int ip = decltype(ip_)::value;
                if constexpr (ip < D / 2) macc[ip] += v;
              }, dr);
              dr.template rest<SP>();
              gws_r2w();
            }
          } else {
            auto dr = gw_drain<PL, SG>(baseL, ra, rb, accL);
            dr.template rest<0>();
            gws_r2w();
          }
          // B window of this bus (G1 = 0 when no line ends here): drains behind the next bus's recomputation
          static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PB::RS, 2 * j, G1[j]); });
          putw(PB::RS, H, f2{0.f, 0.f});
          static_for<0, D / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; putw(PB::RS, 12 + 2 * j, m[j]); });
          gws_w2r();
          *row_ptr(A.adj, ar + 1, lane) = xsum;
          store_pairs<D>(A.adj, ar + RM, lane, macc);
        }
        {   // the last bus's B window
          auto dr = gw_drain<PB, 0>(baseB, ra, rb, accB);
          dr.prologue();
          dr.template rest<0>();
          gws_r2w();
        }
        {   // flush the family's tiles into the wave's slab
          constexpr int ob1 = LIN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
          float* lb_ = slab + A.g_off[C::NPHI + l] + koff * A.g_sz[C::NPHI + l];
          static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, T1[t], lb_, [&](int c, int il) { const int i = 16 * t + il; return c < H ? (i < LIN ? c * LIN + i : (i == LIN ? ob1 + c : -1)) : -1; }, first_store); });
          gws_flush(lane, T2, lb_, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; }, first_store);
          static_for<0, NA4>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, T4[t], lb_, [&](int c, int il) { const int j = 12 * t + c; return (c < 12 && j < OUT) ? (il < H ? oW4 + j * H + il : (il == H ? ob4 + j : -1)) : -1; }, first_store); });
          constexpr int pb1 = PIN * H, pW2 = pb1 + H, pb2 = pW2 + H * H;
          float* pb_ = slab + A.g_off[fphi] + koff * A.g_sz[fphi];
          gws_flush(lane, TP1, pb_, [&](int c, int il) { return c < H ? (il < PIN - D ? c * PIN + D + il : (il == PIN - D ? pb1 + c : -1)) : -1; }, first_store);
          gws_flush(lane, TP2, pb_, [&](int c, int il) { return c < H ? (il < H ? pW2 + c * H + il : (il == H ? pb2 + c : -1)) : -1; }, first_store);
          static_for<0, NDM>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, TPm[t], pb_, [&](int c, int il) { const int i = 16 * t + il; return (c < H && i < D) ? c * PIN + i : -1; }, first_store); });
        }
        STAMP(5 + l)
      });
      } else
      if constexpr (V2) {
      static_for<0, 3>([&](auto o_) {
        constexpr int l = (decltype(o_)::value == 0) ? 2 : decltype(o_)::value - 1;   // L_m first: its upstream is mbar_{k+1} itself
        constexpr int fphi = l == 0 ? 1 : (l == 1 ? 0 : 2);
        constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1);
        constexpr int LIN = C::LF_IN, XL = (LIN + 1) / 2, PIN = C::PHI_IN, SOFF = 2 + D / 2;
        constexpr int NB1 = (2 * XL + 15) / 16, NA4 = (OUTP + 11) / 12, NDM = (D + 15) / 16;
        using NL = NLay<LIN, H, OUTP>;
        // D > 16: 32-column window (GwSubWide).  Two products ride in passes that are issued anyway: the scalar output
        // layer of L_theta / L_v (g3 (x) [a2 | 1], 11 entries) in row 10 of the last dW1 window, whose B columns 4..14 are
        // spare, and the latent tail m[16..D) of phi's dW1 in the spare columns of every line's dW1 pass (the sum over the
        // lines of g1 (x) m IS G1 (x) m: m belongs to the destination bus).  26.6 instead of 31.6 passes per bus and step.
        constexpr bool WIDE = D > 16;
        using SW = std::conditional_t<WIDE, GwSubWide, GwSub>;
        constexpr bool FOLD4 = WIDE && l < 2 && (2 * XL - 16 * (NB1 - 1)) + H + 1 <= 16;   // [x tail | a2 | 1] fits one window
        // L_m (d outputs > 16): one output-layer pass takes rows macc[0..15], the other d - 16 ride in rows 10.. of the same window
        constexpr bool FOLDM = GwSubWide::NA >= 16 && WIDE && l == 2 && (2 * XL - 16 * (NB1 - 1)) + H + 1 <= 16 && H + (OUT - 16) <= 16;
        constexpr int XT = 2 * XL - 16 * (NB1 - 1);                                          // x columns of the last dW1 window
        constexpr int W2OFF = WIDE ? 16 - XT : 0;                                            // its first B column: the tail sits right below column 16
        constexpr int NDMF = WIDE ? 1 : NDM;                                                 // latent tiles left after the fold
        static_assert(!WIDE || ((D - 16) + (PIN - D) + 1 <= 16 && (PIN - D) % 2 == 1), "[line parameters | 1 | latent tail] in one window, the tail on a pair boundary");
        if (l == 2 && k == K - 1) return;
        cfp nb = PN + A.n_off[C::NPHI + l] + koff * A.n_sz[C::NPHI + l];
        cfp pnb = PN + A.n_off[fphi] + koff * A.n_sz[fphi];
        cfp ptb = PT + A.t_off[fphi] + koff * A.t_sz[fphi];
        f32x4 T1[NB1], T2, T4[NA4], TP1, TP2, TPm[NDM];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NB1; ++t) T1[t] = z4;
#pragma unroll
        for (int t = 0; t < NA4; ++t) T4[t] = z4;
#pragma unroll
        for (int t = 0; t < NDM; ++t) TPm[t] = z4;
        T2 = z4; TP1 = z4; TP2 = z4;
        // Weight streams that follow each other closely are linked (gns_device.h, WLink): a stream fetches the first chunk of
        // the next one in its own last step.  Only where nothing but a few vector instructions sits between them: held across a
        // contraction pass the 32 scalar registers of a fetched chunk cost more in spills than the hidden latency is worth
        // (all streams of a bus linked: 577 instead of 416 scalar spills, 2.70 instead of 2.62 ms).
        cfp ptl = PT + A.t_off[C::NPHI + l] + koff * A.t_sz[C::NPHI + l];
        WFirst wf;
        using L0 = WLink<false, true>;            // first of a group
        using L1 = WLink<true, true>;             // middle
        using L2 = WLink<true, false>;            // last
        SSTAMP(11)
        const bool first_family = (l == 2) || (l == 0 && k == K - 1);
        // Step 0 reads m_0 = 0 and produces adjoints of (v, theta, dp, m)_0 that nothing reads (the inputs carry no gradient):
        // the latent rows are not loaded, the input-adjoint row and the latent adjoint are neither read (only L_m needs the
        // latter, as its upstream) nor written.  A third of the sweeps' rows at K = 4.
        const bool step0 = k == 0;
        for (int n = n0; n < n1; ++n) {
          const long long ar = adj_row(n), rr = state_row(k, n);
          const f4 a0 = *row_ptr(A.adj, ar, lane);
          // d/dv, d/dtheta, d/ddp of the L inputs so far: the first sweep of a step (L_m, or L_theta at the last step where L_m
          // has no gradient) starts from zero without reading the row, and Pb-0 does not have to clear it
          f4 xsum = f4{0.f, 0.f, 0.f, 0.f};
          // (step 0: the value is dead - the load goes to the first bus's row, which stays in cache, instead of branching around it:
          //  a branch splits the round of loads at the top of a bus into two memory round trips)
          const long long ar_ld = step0 ? adj_row(n0) : ar;
          if (!first_family) xsum = *row_ptr(A.adj, ar_ld + 1, lane);
          const f4 s0 = *row_ptr(A.state, rr, lane);
          f2 xs[XL];                                          // [v theta | dp dq | m | sum_e h_e | deg, 1]
          f2 (&m)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(xs[2]);
          f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(xs[SOFF]);
          load_pairs<D>(A.state, (step0 ? state_row(0, n0) : rr) + 1, lane, m);   // m_0 = 0 for every bus: step 0 reads one (cached) bus's zero rows
          f2 macc[D / 2];                                     // d/dm_{k+1} (identity path main.py:188), keeps accumulating
          if (l == 0 && k == K - 1) {                         // nothing has touched it yet: zero, and never written before this sweep
#pragma unroll
            for (int i = 0; i < D / 2; ++i) macc[i] = f2{0.f, 0.f};
          } else {
            load_pairs<D>(A.adj, (l != 2 ? ar_ld : ar) + RM, lane, macc);   // (L_theta / L_v at step 0: write-only and dead)
          }
#ifdef GNS_ABLATE_HBM
          load_pairs<H>(A.msg, ((((long long)0 * A.G + g) * N + (n & 3)) * C::NPHI + fphi) * C::HQ, lane, S);
#else
          load_pairs<H>(A.msg, ((((long long)k * A.G + g) * N + n) * C::NPHI + fphi) * C::HQ, lane, S);
#endif
          const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
          xs[0] = f2{s0.x, s0.y}; xs[1] = f2{s0.z, s0.w};
          xs[XL - 1] = f2{(float)(p1 - p0), 1.f};             // deg, and the 1 whose column of dW1 is db1
          f2 gS[H / 2];                                       // adjoint of the hidden-vector sum: what every line ending at n receives
          SSTAMP(0)
          {
            f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
            mlp2_fwd<LIN, H>(ptl, xs, a1, a2, NoBG{}, L0{nullptr, nb, &wf});
            SSTAMP(1)
            // output layer: g2 = (W4^T g3) * lrelu'(a2);  dW4 | db4 += g3 (x) [a2 | 1]
            if constexpr (l == 2) {
              bwd_rows<OUTP, H>(nb, macc, g2, NoBG{}, L2{&wf, nullptr, nullptr});       // m += L_m (main.py:188); a pass follows
            } else {
              const f2 g3s[1] = {f2{l == 0 ? a0.y : (is_gen[n] ? 0.f : a0.x), 0.f}};   // theta += L_theta (:182); v only without a generator (:184-186)
              bwd_rows<2, H>(nb, g3s, g2, NoBG{}, std::conditional_t<FOLD4, L1, L2>{&wf, nb + NL::oW2, &wf});
            }
#pragma unroll
            for (int u = 0; u < H / 2; ++u) g2[u] = g2[u] * dlrelu2(a2[u]);
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
                  gws_putA<SW>(rec, lane, 0, f2{l == 0 ? a0.y : (is_gen[n] ? 0.f : a0.x), 0.f});
                }
                gws_w2r(); gws_pass<SW>(rec, lane, T4[t]); gws_r2w();
              });
            }
            SSTAMP(2)
            // hidden layer: g1 = (W2^T g2) * lrelu'(a1);  dW2 | db2 += g2 (x) [a1 | 1]
            bwd_rows<H, H>(nb + NL::oW2, g2, g1, NoBG{}, std::conditional_t<FOLD4, L2, NoLink>{&wf, nullptr, nullptr});
#pragma unroll
            for (int u = 0; u < H / 2; ++u) g1[u] = g1[u] * dlrelu2(a1[u]);
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g2[j]); gws_putB<SW>(rec, lane, j, a1[j]); });
            gws_putB<SW>(rec, lane, H / 2, f2{1.f, 0.f});
            gws_w2r(); gws_pass<SW>(rec, lane, T2); gws_r2w();
            SSTAMP(3)
            // first layer: dW1 | db1 += g1 (x) [x | 1] in 16-column windows
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g1[j]); });
            if constexpr (FOLD4) gws_putA<SW>(rec, lane, H / 2, f2{l == 0 ? a0.y : (is_gen[n] ? 0.f : a0.x), 0.f});   // row 10: g3 (theta += L_theta :182; v only without a generator :184-186)
            if constexpr (FOLDM) static_for<0, (OUT - 16) / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, H / 2 + j, macc[8 + j]); });   // rows 10..: upstream of outputs 16..d-1 (macc is still mbar_{k+1} here)
            static_for<0, NB1>([&](auto t_) {
              constexpr int t = decltype(t_)::value;
              constexpr int po = (WIDE && t == NB1 - 1) ? W2OFF / 2 : 0;                   // the last window sits right below column 16
              static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < XL) gws_putB<SW>(rec, lane, po + j, xs[8 * t + j]); });
              gws_w2r(); gws_pass<SW, (WIDE && t == NB1 - 1) ? W2OFF : 0>(rec, lane, T1[t]); gws_r2w();
            });
            SSTAMP(4)
            // input adjoints, four at a time, straight to their consumers
            bwd_inputs<(LIN + 3) / 4, H>(nb + NL::total, g1, [&](auto ip_, f2 v) {
              constexpr int ip = decltype(ip_)::value;
              if constexpr (ip == 0) { xsum.x += v.x; xsum.y += v.y; }
              else if constexpr (ip == 1) xsum.z += v.x;
              else if constexpr (ip < SOFF) macc[ip - 2] += v;
              else if constexpr (ip < SOFF + H / 2) gS[ip - SOFF] = v;
            });
          }
          SSTAMP(5)
          if (p0 < p1) {                                      // back through the hidden vectors of the lines ending at n
            f2 uh[H / 2], G1[H / 2];
            if constexpr (WIDE)                               // the latent tail, parked behind [line parameters | 1] at columns 16..21
              static_for<0, (D - 16) / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putB<SW>(rec, lane, 8 + (PIN - D + 2) / 2 + j, m[8 + j]); });
            phi_head<D, H>(ptb, m, uh);
            SSTAMP(6)
#pragma unroll
            for (int j = 0; j < H / 2; ++j) G1[j] = f2{0.f, 0.f};
            for (int p = p0; p < p1; ++p) {
#ifdef GNS_ABLATE_HBM
              const f4 ea = *row_ptr(IN, row_ein + 3LL * (p & 3), lane), eb = *row_ptr(IN, row_ein + 3LL * (p & 3) + 1, lane);
#else
              const f4 ea = *row_ptr(IN, row_ein + 3LL * p, lane), eb = *row_ptr(IN, row_ein + 3LL * p + 1, lane);
#endif
              const f2 xt[3] = {f2{ea.x, ea.y}, f2{ea.z, ea.w}, f2{eb.x, 0.f}};
              f2 a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
              phi_tail<PIN, H, D>(ptb, uh, xt, a1, a2, NoBG{}, L0{nullptr, pnb, &wf});
#pragma unroll
              for (int u = 0; u < H / 2; ++u) g2[u] = gS[u] * dlrelu2(a2[u]);
              bwd_rows<H, H>(pnb, g2, g1, NoBG{}, L2{&wf, nullptr, nullptr});
#pragma unroll
              for (int u = 0; u < H / 2; ++u) { g1[u] = g1[u] * dlrelu2(a1[u]); G1[u] += g1[u]; }
              SSTAMP(7)
              constexpr int pw = WIDE ? 8 : 0;                // wide window: the line's pass contracts columns 16..31
              static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g1[j]); });
              gws_putB<SW>(rec, lane, pw + 0, xt[0]); gws_putB<SW>(rec, lane, pw + 1, xt[1]); gws_putB<SW>(rec, lane, pw + 2, f2{xt[2].x, 1.f});
              gws_w2r(); gws_pass<SW, 2 * pw>(rec, lane, TP1); gws_r2w();
              static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, g2[j]); gws_putB<SW>(rec, lane, j, a1[j]); });
              gws_putB<SW>(rec, lane, H / 2, f2{1.f, 0.f});
              gws_w2r(); gws_pass<SW>(rec, lane, TP2); gws_r2w();
              SSTAMP(8)
            }
            // x = [m(dst) | ...] (main.py:155): d/dm += W1[:, :d]^T G1 and the latent columns of dW1 += G1 (x) m, once per bus
            bwd_inputs<(D + 3) / 4, H>(pnb + NLay2<PIN, H>::total, G1, [&](auto ip_, f2 v) {
              constexpr int ip = decltype(ip_)::value;
              if constexpr (ip < D / 2) macc[ip] += v;
            });
            static_for<0, H / 2>([&](auto j_) { constexpr int j = decltype(j_)::value; gws_putA<SW>(rec, lane, j, G1[j]); });
            static_for<0, NDMF>([&](auto t_) {
              constexpr int t = decltype(t_)::value;
              static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if constexpr (8 * t + j < D / 2) gws_putB<SW>(rec, lane, j, m[8 * t + j]); });
              gws_w2r(); gws_pass<SW>(rec, lane, TPm[t]); gws_r2w();
            });
          }
          SSTAMP(9)
          if (!step0) {
            *row_ptr(A.adj, ar + 1, lane) = xsum;
            store_pairs<D>(A.adj, ar + RM, lane, macc);
          }
          SSTAMP(10)
        }
        {   // flush the family's tiles into the wave's slab (folded blocks: W1[H][IN] b1 W2 b2 [W4 b4])
          constexpr int ob1 = LIN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H;
          float* lb_ = slab + A.g_off[C::NPHI + l] + koff * A.g_sz[C::NPHI + l];
          static_for<0, NB1>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, T1[t], lb_, [&](int c, int il) {
              const bool last = WIDE && t == NB1 - 1;                       // [x tail | a2 | 1]: the x columns end at XT
              const int i = 16 * t + il;
              if (c < H) return (last && il >= XT) ? -1 : (i < LIN ? c * LIN + i : (i == LIN ? ob1 + c : -1));
              if (FOLD4 && last && c == H && il >= XT && il <= XT + H) return il - XT < H ? oW4 + (il - XT) : ob4;   // row 10 = g3: dW4 | db4 of the scalar output
              if (FOLDM && last && c >= H && c < H + (OUT - 16) && il >= XT && il <= XT + H) { const int j = 16 + c - H; return il - XT < H ? oW4 + j * H + (il - XT) : ob4 + j; }
              return -1; }, first_store); });
          gws_flush(lane, T2, lb_, [&](int c, int il) { return c < H ? (il < H ? oW2 + c * H + il : (il == H ? ob2 + c : -1)) : -1; }, first_store);
          if constexpr (FOLDM)
            gws_flush(lane, T4[0], lb_, [&](int c, int il) { return il < H ? oW4 + c * H + il : (il == H ? ob4 + c : -1); }, first_store);   // outputs 0..15
          else if constexpr (!FOLD4)
          static_for<0, NA4>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, T4[t], lb_, [&](int c, int il) { const int j = 12 * t + c; return (c < 12 && j < OUT) ? (il < H ? oW4 + j * H + il : (il == H ? ob4 + j : -1)) : -1; }, first_store); });
          constexpr int pb1 = PIN * H, pW2 = pb1 + H, pb2 = pW2 + H * H;
          float* pb_ = slab + A.g_off[fphi] + koff * A.g_sz[fphi];
          gws_flush(lane, TP1, pb_, [&](int c, int il) {
            if (c >= H) return -1;
            if (il < PIN - D) return c * PIN + D + il;
            if (il == PIN - D) return pb1 + c;
            if (WIDE && il >= PIN - D + 1 && il < PIN - D + 1 + (D - 16)) return c * PIN + 16 + (il - (PIN - D + 1));   // latent tail m[16..D)
            return -1; }, first_store);
          gws_flush(lane, TP2, pb_, [&](int c, int il) { return c < H ? (il < H ? pW2 + c * H + il : (il == H ? pb2 + c : -1)) : -1; }, first_store);
          static_for<0, NDMF>([&](auto t_) { constexpr int t = decltype(t_)::value;
            gws_flush(lane, TPm[t], pb_, [&](int c, int il) { const int i = 16 * t + il; return (c < H && i < D) ? c * PIN + i : -1; }, first_store); });
        }
        STAMP(5 + l)
      });
      } else {
      static_for<0, 3>([&](auto o_) {
        constexpr int l = (decltype(o_)::value == 0) ? 2 : decltype(o_)::value - 1;   // L_m first: its upstream is mbar_{k+1} itself
        constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
        constexpr int OUT = (l == 2) ? D : 1, OUTP = OUT + (OUT & 1);
        // after the last step nothing reads m_K: L_m.{K-1} / phi_m.{K-1} get no gradient (reference: .grad is None)
        if (l == 2 && k == K - 1) return;
        LEngine<C::LF_IN, H, OUT, OUTP, MFMA> engL;
        PEngine<C::PHI_IN, H, MFMA> engP;
        engL.init(lane); engP.init(lane);
        for (int n = n0; n < n1; ++n) {
          const long long ar = adj_row(n), rr = state_row(k, n);
          const f4 a0 = *row_ptr(A.adj, ar, lane);
          f4 xsum = *row_ptr(A.adj, ar + 1, lane);            // d/dv, d/dtheta, d/ddp of the L inputs so far
          const f4 s0 = *row_ptr(A.state, rr, lane);
          f2 m[D / 2];
          load_pairs<D>(A.state, rr + 1, lane, m);
          // gx = adjoint of the L' input [v theta | dp dq | m | sum_e h_e | deg].  Its m part starts as d/dm_{k+1}
          // (identity path main.py:188) and keeps accumulating; its hidden-sum part is what every line ending at n receives.
          constexpr int XL = (C::LF_IN + 1) / 2;
          f2 gx[XL];
          f2 (&macc)[D / 2] = reinterpret_cast<f2 (&)[D / 2]>(gx[2]);
          f2 (&gS)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(gx[2 + D / 2]);
          gx[0] = f2{0.f, 0.f}; gx[1] = f2{0.f, 0.f}; gx[XL - 1] = f2{0.f, 0.f};
          load_pairs<D>(A.adj, ar + RM, lane, macc);
          if constexpr (MULTI) {
#pragma unroll
            for (int j = 0; j < H / 2; ++j) gS[j] = f2{0.f, 0.f};
          } else {
            load_pairs<H>(A.adj, ar + 2, lane, gS);             // the single phi: summed over the three L nets across passes
          }
          const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
          {
            f2 x[XL];
            f2 (&S)[H / 2] = reinterpret_cast<f2 (&)[H / 2]>(x[2 + D / 2]);
            // the hidden-vector sum of family fphi (main.py:155-163, folded) was saved by the forward pass
            load_pairs<H>(A.msg, ((((long long)k * A.G + g) * N + n) * C::NPHI + fphi) * C::HQ, lane, S);
            x[0] = f2{s0.x, s0.y}; x[1] = f2{s0.z, s0.w};
#pragma unroll
            for (int i = 0; i < D / 2; ++i) x[2 + i] = m[i];
            x[XL - 1] = f2{(float)(p1 - p0), 0.f};
            f2 a1[H / 2], a2[H / 2], g3[OUTP / 2], g2[H / 2], g1[H / 2];
            // only the hidden activations are needed here: the output layer of L' (a third of L_m's MACs) is not recomputed;
            // the T-stream of a three-layer block starts with exactly the two-layer layout
            mlp2_fwd<C::LF_IN, H>(PT + A.t_off[C::NPHI + l] + koff * A.t_sz[C::NPHI + l], x, a1, a2);
            if constexpr (l == 0) g3[0] = f2{a0.y, 0.f};                                  // theta += L_theta (main.py:182)
            else if constexpr (l == 1) g3[0] = f2{is_gen[n] ? 0.f : a0.x, 0.f};           // v moves only without a generator (main.py:184-186)
            else {
#pragma unroll
              for (int j = 0; j < D / 2; ++j) g3[j] = macc[j];                            // m += L_m (main.py:188)
            }
            mlp_bwd<C::LF_IN, H, OUTP, 2 * XL, true>(PN + A.n_off[C::NPHI + l] + koff * A.n_sz[C::NPHI + l], a1, a2, g3, g2, g1, gx);
            xsum.x += gx[0].x; xsum.y += gx[0].y; xsum.z += gx[1].x;
            engL.accumulate(rec, lane, x, a1, a2, g1, g2, g3);
          }
          if constexpr (MULTI) {
            if (p0 < p1) load_pairs<D>(A.state, rr + 1, lane, m);   // re-read (cache-hot) instead of keeping 20 registers live across L'
#ifdef GNS_ABLATE_MSGBWD
            for (int p = p0; p < p0; ++p) {
#else
            for (int p = p0; p < p1; ++p) {                     // back through the hidden vectors of the lines ending at n
#endif
              f2 xe[(C::PHI_IN + 1) / 2], a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
              edge_input(p, m, xe);
              mlp2_fwd<C::PHI_IN, H>(PT + A.t_off[fphi] + koff * A.t_sz[fphi], xe, a1, a2);
              // x = [m(dst) | ...] (main.py:155): the m part of the input adjoint goes straight into macc
              mlp2_bwd<C::PHI_IN, H, D, true>(PN + A.n_off[fphi] + koff * A.n_sz[fphi], a1, a2, gS, g2, g1, macc);
              engP.accumulate(rec, lane, xe, a1, g1, g2);
            }
          } else {
            store_pairs<H>(A.adj, ar + 2, lane, gS);
          }
          *row_ptr(A.adj, ar + 1, lane) = xsum;
          store_pairs<D>(A.adj, ar + RM, lane, macc);
        }
        engL.flush(lane, slab + A.g_off[C::NPHI + l] + koff * A.g_sz[C::NPHI + l]);
        if constexpr (MULTI) engP.flush(lane, slab + A.g_off[fphi] + koff * A.g_sz[fphi]);
        STAMP(5 + l)
      });
      if constexpr (!MULTI) {                                  // the single phi: its hidden-sum adjoint is the sum over the three L nets
        PEngine<C::PHI_IN, H, MFMA> engP;
        engP.init(lane);
        for (int n = n0; n < n1; ++n) {
          const long long ar = adj_row(n), rr = state_row(k, n);
          f2 m[D / 2], macc[D / 2], gS[H / 2];
          load_pairs<D>(A.state, rr + 1, lane, m);
          load_pairs<D>(A.adj, ar + RM, lane, macc);
          load_pairs<H>(A.adj, ar + 2, lane, gS);
          const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
          for (int p = p0; p < p1; ++p) {
            f2 xe[(C::PHI_IN + 1) / 2], a1[H / 2], a2[H / 2], g2[H / 2], g1[H / 2];
            edge_input(p, m, xe);
            mlp2_fwd<C::PHI_IN, H>(PT + A.t_off[0] + koff * A.t_sz[0], xe, a1, a2);
            mlp2_bwd<C::PHI_IN, H, D, true>(PN + A.n_off[0] + koff * A.n_sz[0], a1, a2, gS, g2, g1, macc);
            engP.accumulate(rec, lane, xe, a1, g1, g2);
          }
          store_pairs<D>(A.adj, ar + RM, lane, macc);
        }
        engP.flush(lane, slab + A.g_off[0] + koff * A.g_sz[0]);
      }
      }   // !V2
      STAMP(8)
    }
    team_barrier(team);
  }
  if (team_failed && lane == 0) slab[0] = __builtin_nanf("");          // a team barrier gave up: the gradients must not look valid
  if (tsize > 1 && team_failed && threadIdx.x == 0)                    // ... and the host can ask (gns_team_status)
    __hip_atomic_store(reinterpret_cast<unsigned*>(A.team_ws) + GNS_TEAM_STATUS_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The eight per-wave slabs of this workgroup are summed here, in wave order, into the first one (they are still in this
  // CU's L1 / the XCD's L2): the reduction kernels then read one slab per workgroup instead of eight (121 MB -> 15 MB).
  __syncthreads();
  {
    float* base = A.slab + (long long)blockIdx.x * W * A.slab_floats;
    for (long long i = 4LL * threadIdx.x; i < A.slab_floats; i += 4LL * GNS_BWD_THREADS) {
      f4 acc = *reinterpret_cast<const f4*>(base + i);
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const f4 t = *reinterpret_cast<const f4*>(base + w * A.slab_floats + i);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      *reinterpret_cast<f4*>(base + i) = acc;
    }
  }
#ifdef GNS_STAMPS
  if (lane == 0) for (int i = 0; i < 10; ++i) A.slots[((long long)blockIdx.x * W + wave) * 10 + i] = (float)tph[i];   // diagnostic build only: slots are dead by now
  if (lane == 0) for (int i = 0; i < 12; ++i) A.slots[(long long)gridDim.x * W * 10 + ((long long)blockIdx.x * W + wave) * 12 + i] = (float)tsw[i];
#endif
}

// ---- slab reduction: grad[i] += sum over slabs, two fixed-order stages (bitwise reproducible) ----------
__global__ void gns_reduce_stage1(const float* __restrict__ slab, float* __restrict__ part, long long nslab, long long sf, long long stride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sf) return;
  const long long j = blockIdx.y, s0 = nslab * j / GNS_RED_PARTS, s1 = nslab * (j + 1) / GNS_RED_PARTS;
  float acc = 0.f;
#pragma unroll 8
  for (long long s = s0; s < s1; ++s) acc += slab[s * stride + i];
  part[j * sf + i] = acc;
}
__global__ void gns_reduce_stage2(const float* __restrict__ part, float* __restrict__ out, long long sf, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int j = 0; j < GNS_RED_PARTS; ++j) acc += part[j * sf + i];
  out[i] = acc;
}

// Folded gradients -> gradients of the reference's parameters (gns_common.h, "FOLDED form"):
//   C = A W4, c0 = A b4 with A = W1L[:, 4+d:]  =>  dA = dC W4^T + dc0 b4^T,  dW4 = sum_L A^T dC,  db4 = sum_L A^T dc0.
// One block per (family, k); a phi block also collects dW4/db4 from every L family that reads it, in a fixed order.
__global__ void gns_unfold_kernel(const float* __restrict__ gf, const float* __restrict__ flat, float* __restrict__ grad,
                                  GnsFamilies fam, int K, int D, int H) {
  // D, H = the kernel's dims (layout of the folded gradient gf); Dr, Hr <= them = the model's (layout of flat / grad): a narrower
  // model ran zero-padded (gns_common.h, GnsFamilies) and the gradient of its padding is dropped here
  const int blk = blockIdx.x, f = blk / K, k = blk % K;
  const bool is_phi = f < fam.nphi;
  const int Dr = fam.dr, Hr = fam.hr;
  const int IN = fam.in[f], OUT = fam.out[f];                     // the model's
  const int HEAD = 4 + D, INF = HEAD + H + 1, HEADr = 4 + Dr;
  const float* g = gf + fam.g_off[f] + (int64_t)k * fam.g_sz[f];
  float* dst = grad + fam.flat_off[f] + (int64_t)k * fam.flat_sz[f];
  if (is_phi) {
    const int INk = D + 5;
    const int gb1 = INk * H, gW2 = gb1 + H, gb2 = gW2 + H * H;    // folded gradient: W1[H][INk] b1[H] W2[H][H] b2[H]
    const int oW1 = IN * Hr, ob1 = oW1 + Hr, oW2 = ob1 + Hr * Hr, n12 = oW2 + Hr;
    for (int e = threadIdx.x; e < n12; e += blockDim.x) {
      float v;
      if (e < oW1) { const int c = e / IN, i = e % IN; v = g[c * INk + (i < Dr ? i : D + (i - Dr))]; }
      else if (e < ob1) v = g[gb1 + (e - oW1)];
      else if (e < oW2) { const int q = e - ob1; v = g[gW2 + (q / Hr) * H + (q % Hr)]; }
      else v = g[gb2 + (e - oW2)];
      dst[e] += v;
    }
    float* dW4 = dst + n12;                                       // [OUT][Hr]
    float* db4 = dW4 + OUT * Hr;
    for (int e = threadIdx.x; e < OUT * Hr + OUT; e += blockDim.x) {
      float acc = 0.f;
      for (int lf = fam.nphi; lf < fam.nfam; ++lf) {
        if (fam.phi_of[lf] != f) continue;
        const float* gl = gf + fam.g_off[lf] + (int64_t)k * fam.g_sz[lf];             // dW1'[H][INF]
        const float* Wl = flat + fam.flat_off[lf] + (int64_t)k * fam.flat_sz[lf];     // W1L [Hr][L_IN]
        const int LIN = fam.in[lf];
        if (e < OUT * Hr) { const int q = e / Hr, j = e % Hr; for (int c = 0; c < Hr; ++c) acc += Wl[c * LIN + HEADr + q] * gl[c * INF + HEAD + j]; }
        else { const int q = e - OUT * Hr; for (int c = 0; c < Hr; ++c) acc += Wl[c * LIN + HEADr + q] * gl[c * INF + HEAD + H]; }
      }
      if (e < OUT * Hr) dW4[e] += acc; else db4[e - OUT * Hr] += acc;
    }
    return;
  }
  const int fp = fam.phi_of[f], PO = fam.out[fp];
  const float* ps = flat + fam.flat_off[fp] + (int64_t)k * fam.flat_sz[fp];
  const float* pW4 = ps + fam.in[fp] * Hr + Hr + Hr * Hr + Hr;
  const float* pb4 = pW4 + PO * Hr;
  for (int e = threadIdx.x; e < Hr * IN; e += blockDim.x) {       // dW1L [Hr][IN]
    const int c = e / IN, i = e % IN;
    float v = 0.f;
    if (i < HEADr) v = g[c * INF + i];                            // (4 + latent index: the same column in both layouts)
    else if (i - HEADr < PO) {
      const int q = i - HEADr;
      for (int j = 0; j < Hr; ++j) v += g[c * INF + HEAD + j] * pW4[q * Hr + j];
      v += g[c * INF + HEAD + H] * pb4[q];
    }                                                             // single phi: columns 4+d+1.. multiply zeros -> gradient 0
    dst[e] += v;
  }
  // b1 W2 b2 W4 b4: folded gradient [H] [H][H] [H] [OUTk][H] [OUTk] behind W1'[H][INF]
  const int OUTk = fam.outk[f];
  const int gb1 = H * INF, gW2 = gb1 + H, gb2 = gW2 + H * H, gW4 = gb2 + H, gb4 = gW4 + OUTk * H;
  const int ob1 = Hr * IN, oW2 = ob1 + Hr, ob2 = oW2 + Hr * Hr, oW4 = ob2 + Hr, ob4 = oW4 + OUT * Hr, oend = ob4 + OUT;
  for (int e = ob1 + threadIdx.x; e < oend; e += blockDim.x) {
    float v;
    if (e < oW2) v = g[gb1 + (e - ob1)];
    else if (e < ob2) { const int q = e - oW2; v = g[gW2 + (q / Hr) * H + (q % Hr)]; }
    else if (e < oW4) v = g[gb2 + (e - ob2)];
    else if (e < ob4) { const int q = e - oW4; v = g[gW4 + (q / Hr) * H + (q % Hr)]; }
    else v = g[gb4 + (e - ob4)];
    dst[e] += v;
  }
}

template <int D, int H, bool MULTI, bool MFMA, int VAR>
static int launch_backward_t(const GnsBwdArgs& A, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((gns_backward_kernel<D, H, MULTI, MFMA, VAR>), dim3(blocks), dim3(GNS_BWD_THREADS), 0, st, A);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}

int gns_backward_persistent_supported(int d, int h) {
#define GNS_CASE(DD, HH) if (d == DD && h == HH) return 1;
  GNS_FOR_EACH_DIMS_PERSISTENT(GNS_CASE)
#undef GNS_CASE
  return 0;
}

int gns_launch_backward(int d, int h, int multi, int mfma, int variant, const GnsBwdArgs& A, int blocks, hipStream_t st) {
#define GNS_CASE(DD, HH)                                                                                      \
  if (d == DD && h == HH) {                                                                                   \
    if (mfma && multi && variant == 3) return launch_backward_t<DD, HH, true, true, 3>(A, blocks, st);        \
    if (mfma && multi && variant == 2) return launch_backward_t<DD, HH, true, true, 2>(A, blocks, st);        \
    if (mfma) return multi ? launch_backward_t<DD, HH, true, true, 1>(A, blocks, st) : launch_backward_t<DD, HH, false, true, 1>(A, blocks, st);   \
    return multi ? launch_backward_t<DD, HH, true, false, 1>(A, blocks, st) : launch_backward_t<DD, HH, false, false, 1>(A, blocks, st);          \
  }
  GNS_FOR_EACH_DIMS_PERSISTENT(GNS_CASE)
#undef GNS_CASE
  return GNS_EUNSUPPORTED;
}

int gns_launch_reduce(const float* slab, float* part, float* tmp, const float* flat, float* grad, long long nslab, long long sf,
                      const GnsFamilies& fam, int K, int D, int H, hipStream_t st, long long stride) {
  if (stride <= 0) stride = sf;
  hipLaunchKernelGGL(gns_reduce_stage1, dim3((unsigned)((sf + 255) / 256), GNS_RED_PARTS), dim3(256), 0, st, slab, part, nslab, sf, stride);
  hipLaunchKernelGGL(gns_reduce_stage2, dim3((unsigned)((fam.g_total + 255) / 256)), dim3(256), 0, st, part, tmp, sf, (long long)fam.g_total);
  hipLaunchKernelGGL(gns_unfold_kernel, dim3(fam.nfam * K), dim3(256), 0, st, tmp, flat, grad, fam, K, D, H);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
